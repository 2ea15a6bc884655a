// Probe (runs on the GPU box): does v_pk_minimum3_f16 / v_pk_maximum3_f16 of gfx950 order the bit patterns 0x0000 .. 0x00ff
// (f16 denormals: n * 2^-24) like the integers they are - i.e. no flush to zero, no canonicalisation - in both halves?
// If so, one instruction takes the minimum (maximum) of three ring pixels of TWO FAST candidates at once.
// Also checks the biased form (0x0400 | n: normal numbers) as a fallback.
// build: hipcc --offload-arch=gfx950 -O2 tools/pkmin3_probe.hip -o /tmp/pkmin3_probe
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void probe(unsigned *bad, unsigned bias) {
    const unsigned a = blockIdx.x, b = threadIdx.x;
    unsigned nbad = 0;
    for (unsigned c = 0; c < 256; c++) {
        const unsigned x = (a | ((255u - a) << 16)) | bias, y = (b | ((c ^ b) << 16)) | bias, z = (c | ((a ^ c ^ 0x55u) << 16)) | bias;
        unsigned mn, mx;
        asm volatile("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(mn) : "v"(x), "v"(y), "v"(z));
        asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(mx) : "v"(x), "v"(y), "v"(z));
        const unsigned xl = x & 0xffffu, yl = y & 0xffffu, zl = z & 0xffffu, xh = x >> 16, yh = y >> 16, zh = z >> 16;
        const unsigned emn = min(min(xl, yl), zl) | (min(min(xh, yh), zh) << 16);
        const unsigned emx = max(max(xl, yl), zl) | (max(max(xh, yh), zh) << 16);
        nbad += (mn != emn) + (mx != emx);
    }
    if (nbad) atomicAdd(bad, nbad);
}

int main() {
    unsigned *d, h[2] = {0, 0};
    hipMalloc(&d, 8);
    hipMemset(d, 0, 8);
    probe<<<256, 256>>>(d, 0u);
    probe<<<256, 256>>>(d + 1, 0x04000400u);
    hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("v_pk_minimum3_f16 / v_pk_maximum3_f16 on u8 bit patterns: %u mismatches of %u (denormal patterns), %u (biased 0x0400)\n", h[0],
           2u * 256 * 256 * 256, h[1]);
    return h[0] ? 1 : 0;
}
