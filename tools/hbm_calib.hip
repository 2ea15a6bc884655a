// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths this repo uses
// (MI355X_MICROARCH.md: FETCH_SIZE reads exactly half of a 16-B/lane coalesced stream; other widths and
// WRITE_SIZE are uncalibrated).  Each kernel moves a known number of bytes once; compare with the counters.
// build: hipcc --offload-arch=gfx950 -O3 tools/hbm_calib.hip -o /tmp/hbm_calib
#include <hip/hip_runtime.h>

#include <cstdio>

__global__ void read_b32(const unsigned *p, size_t n, unsigned *out) {
    unsigned acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void read_b128(const uint4 *p, size_t n, unsigned *out) {
    unsigned acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint4 v = p[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void read_rows48(const unsigned char *p, size_t rows, size_t pitch, unsigned *out) {
    // the FAST tile pattern: 12 aligned dwords (48 B) out of every `pitch` bytes
    unsigned acc = 0;
    const size_t n = rows * 12;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        acc += *(const unsigned *)(p + (i / 12) * pitch + 4 * (i % 12));
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void write_b32(unsigned *p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (unsigned)i;
}
__global__ void write_b8(unsigned char *p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (unsigned char)i;
}

int main() {
    const size_t bytes = 1ull << 30;  // 1 GiB: far beyond the 256 MiB Infinity Cache
    unsigned char *buf;
    unsigned *out;
    hipMalloc(&buf, bytes);
    hipMalloc(&out, 64);
    hipMemset(buf, 1, bytes);
    hipDeviceSynchronize();
    read_b32<<<4096, 256>>>((const unsigned *)buf, bytes / 4, out);
    read_b128<<<4096, 256>>>((const uint4 *)buf, bytes / 16, out);
    read_rows48<<<4096, 256>>>(buf, bytes / 1280, 1280, out);
    write_b32<<<4096, 256>>>((unsigned *)buf, bytes / 4);
    write_b8<<<4096, 256>>>(buf, bytes / 4);
    hipDeviceSynchronize();
    // 1280 = 10 x 128: every row's 48-byte segment starts a 128-byte line, so a row touches one 64-B line and one 128-B line
    printf("expected_KB read_b32 %zu read_b128 %zu read_rows48 %zu (48 of every 1280 B; whole 64-B lines touched: %zu; whole 128-B "
           "lines touched: %zu) write_b32 %zu write_b8 %zu\n",
           bytes / 1024, bytes / 1024, bytes / 1280 * 48 / 1024, bytes / 1280 * 64 / 1024, bytes / 1280 * 128 / 1024, bytes / 1024,
           bytes / 4 / 1024);
    return 0;
}
