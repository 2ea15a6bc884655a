// Host-to-device bandwidth of pinned memory on the GPU box, for the host-in leg of bench.py: one linear copy, the same bytes on
// 2 / 4 streams, and the strided form the front end uses (rows = frames of 1280x720 into slots 2.9 MB apart).
// build + run: hipcc --offload-arch=gfx950 -O2 tools/h2d_probe.hip -o /tmp/h2d_probe && /tmp/h2d_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <vector>

__global__ void spin(unsigned *out, int iters) {
    unsigned a = threadIdx.x, b = blockIdx.x + 1;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 64; k++) a = a * 1664525u + b;
    }
    if (a == 0x12345u) out[0] = a;
}

int main() {
    const size_t frame = 1280 * 720, n = 1024, bytes = frame * n, slot = 2916352;
    char *h, *d;
    hipHostMalloc((void **)&h, bytes, hipHostMallocDefault);
    hipMalloc((void **)&d, slot * n);
    for (size_t i = 0; i < bytes; i += 4096) h[i] = (char)i;
    std::vector<hipStream_t> st(8);
    for (auto &s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    auto timeit = [&](const char *name, int streams, bool strided) {
        for (int rep = 0; rep < 2; rep++) {
            hipDeviceSynchronize();
            const auto t0 = std::chrono::steady_clock::now();
            const size_t per = n / streams;
            for (int s = 0; s < streams; s++) {
                if (strided) hipMemcpy2DAsync(d + s * per * slot, slot, h + s * per * frame, frame, frame, per, hipMemcpyHostToDevice, st[s]);
                else hipMemcpyAsync(d + s * per * frame, h + s * per * frame, per * frame, hipMemcpyHostToDevice, st[s]);
            }
            hipDeviceSynchronize();
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (rep) printf("%-34s %6.1f GB/s\n", name, bytes / dt / 1e9);
        }
    };
    timeit("1 linear copy", 1, false);
    timeit("2 streams, linear", 2, false);
    timeit("4 streams, linear", 4, false);
    timeit("8 streams, linear", 8, false);
    timeit("1 strided copy (1024 frames)", 1, true);
    timeit("2 streams, strided", 2, true);
    timeit("4 streams, strided", 4, true);
    timeit("8 streams, strided", 8, true);
    // does a copy overlap with a kernel that keeps every CU busy?  (the host-in leg needs uploads of batch k + 1 to run
    // beside the kernels of batch k)
    unsigned *out;
    hipMalloc((void **)&out, 64);
    auto overlap = [&](const char *name, bool strided) {
        for (int rep = 0; rep < 2; rep++) {
            hipDeviceSynchronize();
            auto t0 = std::chrono::steady_clock::now();
            spin<<<256 * 32, 256, 0, st[0]>>>(out, 4000);
            hipDeviceSynchronize();
            const double tk = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            t0 = std::chrono::steady_clock::now();
            if (strided) hipMemcpy2DAsync(d, slot, h, frame, frame, n, hipMemcpyHostToDevice, st[1]);
            else hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st[1]);
            hipDeviceSynchronize();
            const double tc = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            t0 = std::chrono::steady_clock::now();
            spin<<<256 * 32, 256, 0, st[0]>>>(out, 4000);
            if (strided) hipMemcpy2DAsync(d, slot, h, frame, frame, n, hipMemcpyHostToDevice, st[1]);
            else hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st[1]);
            hipDeviceSynchronize();
            const double tb = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (rep) printf("%-34s kernel %5.1f ms, copy %5.1f ms, both at once %5.1f ms\n", name, tk * 1e3, tc * 1e3, tb * 1e3);
        }
    };
    overlap("overlap: linear copy", false);
    overlap("overlap: strided copy", true);
    return 0;
}
