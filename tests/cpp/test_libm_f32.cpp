// Host check of fasttrack_amd/csrc/libm_f32.h against the libm this process is linked with:
//   * cosf_glibc / sinf_glibc == cosf / sinf for every float of [-6.2831860, 6.2831860] (the rBRIEF rotation angle of
//     /root/reference/src/ORBextractor.cc:72-74 is fastAtan2 degrees * (float)(pi/180); KB8's psi is in [-pi, pi]),
//   * logf_glibc == logf for every positive float up to 1e4 (PredictScale's ratio, /root/reference/src/MapPoint.cc:539)
//     plus every float above it at a coarse stride.
// Build with -DFT_LIBM_CONTRACT=1 (glibc's *_fma ifunc variant, the one an AVX2+FMA host runs) or =0 (the *_sse2
// variant); -ffp-contract=off in both so that only the explicit fma calls fuse.
// usage: test_libm_f32 [stride] [threads]     stride 1 = exhaustive (default 1), threads default 8
// Prints one line per function: checked / mismatches, and the first mismatching arguments.  Exit code 1 on mismatch.
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include "../../fasttrack_amd/csrc/libm_f32.h"

static float from_bits(uint32_t u) {
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static uint32_t to_bits(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}

template <class Mine, class Host>
static unsigned long long sweep(const char *name, uint32_t lo, uint32_t hi, uint32_t stride, int threads, Mine mine,
                                Host host) {
    std::atomic<unsigned long long> bad{0}, checked{0};
    std::atomic<int> shown{0};
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++)
        pool.emplace_back([&, t] {
            unsigned long long b = 0, c = 0;
            for (uint64_t u = (uint64_t)lo + (uint64_t)t * stride; u <= hi; u += (uint64_t)stride * threads) {
                const float x = from_bits((uint32_t)u);
                const float m = mine(x), h = host(x);
                if (to_bits(m) != to_bits(h)) {
                    b++;
                    if (shown.fetch_add(1) < 5) printf("  %s(%.9g = 0x%08x): mine %.9g host %.9g\n", name, x, (uint32_t)u, m, h);
                }
                c++;
            }
            bad += b;
            checked += c;
        });
    for (auto &th : pool) th.join();
    printf("%s: checked %llu mismatches %llu\n", name, checked.load(), bad.load());
    return bad.load();
}

int main(int argc, char **argv) {
    const uint32_t stride = argc > 1 ? (uint32_t)atoi(argv[1]) : 1u;
    const int threads = argc > 2 ? atoi(argv[2]) : 8;
    printf("FT_LIBM_CONTRACT=%d stride %u\n", FT_LIBM_CONTRACT, stride);
    unsigned long long bad = 0;
    const uint32_t top = to_bits(6.2831860f);  // just above 360 * (float)(pi / 180)
    // volatile function pointers: the compiler must call libm, not fold or substitute a builtin
    float (*volatile hcos)(float) = cosf;
    float (*volatile hsin)(float) = sinf;
    float (*volatile hlog)(float) = logf;
    bad += sweep("cosf", 0, top, stride, threads, [](float x) { return ft_libm::cosf_glibc(x); }, [&](float x) { return hcos(x); });
    bad += sweep("sinf", 0, top, stride, threads, [](float x) { return ft_libm::sinf_glibc(x); }, [&](float x) { return hsin(x); });
    // negative arguments: KannalaBrandt8::project takes cos / sin of psi = atan2f(y, x) in [-pi, pi]
    // (/root/reference/src/CameraModels/KannalaBrandt8.cpp:74-75)
    bad += sweep("cosf(neg)", 0x80000000u, 0x80000000u | top, stride, threads, [](float x) { return ft_libm::cosf_glibc(x); },
                 [&](float x) { return hcos(x); });
    bad += sweep("sinf(neg)", 0x80000000u, 0x80000000u | top, stride, threads, [](float x) { return ft_libm::sinf_glibc(x); },
                 [&](float x) { return hsin(x); });
    // every positive float (subnormals included) up to 1e4, then the rest of the finite range at 64x the stride
    bad += sweep("logf(0,1e4]", 1, to_bits(1e4f), stride, threads, [](float x) { return ft_libm::logf_glibc(x); },
                 [&](float x) { return hlog(x); });
    bad += sweep("logf(1e4,max]", to_bits(1e4f), 0x7f7fffffu, stride * 64u, threads,
                 [](float x) { return ft_libm::logf_glibc(x); }, [&](float x) { return hlog(x); });
    return bad ? 1 : 0;
}
