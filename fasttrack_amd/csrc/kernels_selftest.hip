// ft_selftest_libm: does the DEVICE evaluation of libm_f32.h equal the libm THIS PROCESS is linked with?
//
// k_orient_desc's rotation and k_frustum's PredictScale reproduce glibc's cosf / sinf / logf (libm_f32.h) because the
// reference evaluates those on the host (/root/reference/src/ORBextractor.cc:73-74, src/MapPoint.cc:539).  Bit-exact
// agreement with a reference binary therefore depends on the host libm being the glibc algorithm; a deployment on another
// libc (musl, a vendor libm) can run this sweep once to find out.  The device evaluates the functions over a range of float
// bit patterns, the host compares with its own cosf / sinf / logf.  The same for atanf / atan2f of the KannalaBrandt8
// projection (src/CameraModels/KannalaBrandt8.cpp:67-84); atan2f(y, x) sweeps y and pairs every y with one x derived from
// its bits (pair_x: either sign, 2^-9 <= |x| < 2^7 - ray coordinates); tanf (KannalaBrandt8::unproject, :114-143) is restated
// for |x| < 120.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <thread>
#include <vector>

#include "ft_host.h"
#include "libm_f32.h"

namespace {

__host__ __device__ inline float pair_x(uint32_t ybits) {
    uint32_t h = ybits * 0x9E3779B1u;
    h ^= h >> 15;
    h *= 0x85EBCA77u;
    h ^= h >> 13;
    const uint32_t bits = (h & 0x807fffffu) | ((118u + ((h >> 24) & 15u)) << 23);
    return __builtin_bit_cast(float, bits);
}

__device__ inline float tan_or_nan(float x) {
    bool exact;
    const float t = ft_libm::tanf_glibc(x, &exact);
    return exact ? t : __builtin_nanf("");  // outside the restated range: reported as a mismatch unless the host says NaN too
}

__global__ __launch_bounds__(256) void k_libm_sweep(int func, uint32_t first, uint32_t stride, uint32_t n, float *out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float x = __uint_as_float(first + i * stride);
    out[i] = func == 0   ? ft_libm::cosf_glibc(x)
             : func == 1 ? ft_libm::sinf_glibc(x)
             : func == 2 ? ft_libm::logf_glibc(x)
             : func == 3 ? ft_libm::atanf_glibc(x)
             : func == 4 ? ft_libm::atan2f_glibc(x, pair_x(first + i * stride))
                         : tan_or_nan(x);
}

float host_eval(int func, float x) {
    // through volatile pointers: the compiler must call the process's libm, not fold or substitute
    static float (*volatile fc)(float) = cosf;
    static float (*volatile fs)(float) = sinf;
    static float (*volatile fl)(float) = logf;
    static float (*volatile fa)(float) = atanf;
    static float (*volatile fa2)(float, float) = atan2f;
    static float (*volatile ft)(float) = tanf;
    return func == 0   ? fc(x)
           : func == 1 ? fs(x)
           : func == 2 ? fl(x)
           : func == 3 ? fa(x)
           : func == 4 ? fa2(x, pair_x(__builtin_bit_cast(uint32_t, x)))
                       : ft(x);
}

}  // namespace

extern "C" FT_API int ft_selftest_libm(ft_context *ctx, int func, uint32_t first_bits, uint32_t last_bits, uint32_t stride,
                                       unsigned long long *checked, unsigned long long *mismatches, uint32_t *first_bad) {
    if (!ctx || func < 0 || func > 5 || stride == 0 || last_bits < first_bits || !checked || !mismatches) {
        ft_set_error("ft_selftest_libm: bad arguments");
        return FT_ERR_INVALID;
    }
    if (int rc = ft_set_device(ctx)) return rc;
    const uint32_t chunk = 1u << 24;
    float *d = nullptr;
    FT_HIP(hipMalloc(&d, (size_t)chunk * 4));
    std::vector<float> h(chunk);
    const int nt = std::max(1, std::min(ft_usable_cpus(), 32));
    unsigned long long total = 0, bad = 0;
    uint32_t firstBad = 0;
    bool haveBad = false;
    int rc = FT_OK;
    for (uint64_t b = first_bits; b <= last_bits; b += (uint64_t)chunk * stride) {
        const uint32_t n = (uint32_t)std::min<uint64_t>(chunk, ((uint64_t)last_bits - b) / stride + 1);
        k_libm_sweep<<<(n + 255) / 256, 256, 0, ctx->stream>>>(func, (uint32_t)b, stride, n, d);
        hipError_t e = hipMemcpyAsync(h.data(), d, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            rc = ft_hip_fail(e, "ft_selftest_libm", __FILE__, __LINE__);
            break;
        }
        std::vector<unsigned long long> badT(nt, 0);
        std::vector<uint32_t> firstT(nt, 0xffffffffu);
        std::vector<std::thread> pool;
        for (int t = 0; t < nt; t++)
            pool.emplace_back([&, t] {
                for (uint32_t i = (uint32_t)t; i < n; i += (uint32_t)nt) {
                    const uint32_t bits = (uint32_t)b + i * stride;
                    float x;
                    memcpy(&x, &bits, 4);
                    const float r = host_eval(func, x);
                    // NaNs compare by kind, everything else by bits
                    if (memcmp(&r, &h[i], 4) != 0 && !(r != r && h[i] != h[i])) {
                        badT[t]++;
                        firstT[t] = std::min(firstT[t], bits);
                    }
                }
            });
        for (auto &th : pool) th.join();
        for (int t = 0; t < nt; t++) {
            bad += badT[t];
            if (badT[t] && (!haveBad || firstT[t] < firstBad)) {
                firstBad = firstT[t];
                haveBad = true;
            }
        }
        total += n;
    }
    hipFree(d);
    *checked = total;
    *mismatches = bad;
    if (first_bad) *first_bad = firstBad;
    return rc;
}
