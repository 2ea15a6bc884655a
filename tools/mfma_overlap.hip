// Does the matrix pipe of gfx950 take work OFF the vector issue of a SIMD?  (round 6, the experiment of VERDICT r5 item 7:
// k_orient_desc's horizontal blur as v_mfma_i32_16x16x32_i8.)  Streams of independent instructions, bracketed by s_memtime as in
// tools/valu_rate.hip, W waves per SIMD on every CU:
//   mfma        v_mfma_i32_16x16x32_i8 only (four independent accumulators)
//   valu        v_dot4_u32_u8 only (the 4.3-cycle class, what the matrix products replaced)
//   mixed k     one wave alternating 1 matrix product with k v_dot4 (independent registers)
//   split       half of the waves of a SIMD run the mfma stream, the other half the valu stream (W >= 2)
// Reported: shader cycles per SIMD for one "group" (1 matrix product + k vector instructions) - if the pipes overlap, a group costs
// max(matrix, k x vector); if the matrix product holds the SIMD's issue for its passes, the sum.
// build + run: hipcc --offload-arch=gfx950 -O3 tools/mfma_overlap.hip -o /tmp/mfma_overlap && /tmp/mfma_overlap
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define ITER 1024
typedef int v4i __attribute__((ext_vector_type(4)));

#define T_BEGIN                                                                                 \
    unsigned long long t0, t1;                                                                  \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_barrier\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#define T_END                                                                                   \
    asm volatile("s_nop 15\n s_nop 15\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory"); \
    if ((threadIdx.x & 63) == 0) out[1 + blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = (unsigned)(t1 - t0);

#define MFMA(acc) asm volatile("v_mfma_i32_16x16x32_i8 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define DOT(x) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(x) : "v"(p), "v"(q))

// K vector instructions per matrix product (K = -1: no matrix products, 8 vector instructions per step; K = 0: matrix only)
template <int K>
__global__ __launch_bounds__(512) void k_stream(unsigned *out, unsigned seed, int split) {
    long a = threadIdx.x * 0x0101010101010101ll + seed, b = a ^ 0x0102030405060708ll;
    unsigned p = threadIdx.x * 2654435761u + seed, q = p ^ 0x5bd1e995u;
    v4i c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    unsigned x0 = p, x1 = q, x2 = p + 1, x3 = q + 1, x4 = p + 2, x5 = q + 2, x6 = p + 3, x7 = q + 3;
    // split: blocks of eight waves (two per SIMD: wave w runs on SIMD w % 4) - waves 0 .. 3 the matrix stream, 4 .. 7 the vector one
    const int mode = split ? ((threadIdx.x >> 8) ? -1 : 0) : K;
    T_BEGIN
    if (mode == 0) {
        for (int it = 0; it < ITER; it++) { MFMA(c0); MFMA(c1); MFMA(c2); MFMA(c3); }
    } else if (mode < 0) {
        for (int it = 0; it < ITER; it++) { DOT(x0); DOT(x1); DOT(x2); DOT(x3); DOT(x4); DOT(x5); DOT(x6); DOT(x7); }
    } else {
        for (int it = 0; it < ITER; it++) {
#define GROUP(acc)                                                                              \
    MFMA(acc);                                                                                  \
    if (K >= 1) DOT(x0);                                                                        \
    if (K >= 2) DOT(x1);                                                                        \
    if (K >= 3) DOT(x2);                                                                        \
    if (K >= 4) DOT(x3);                                                                        \
    if (K >= 5) DOT(x4);                                                                        \
    if (K >= 6) DOT(x5);                                                                        \
    if (K >= 7) DOT(x6);                                                                        \
    if (K >= 8) DOT(x7);
            GROUP(c0) GROUP(c1) GROUP(c2) GROUP(c3)
        }
    }
    T_END
    const unsigned s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + c0.x + c1.y + c2.z + c3.w;
    if (s == 0x12345u) out[0] = s;
}

static int nCU = 256;
template <int K>
static void run(const char *label, int W, int split, unsigned *dOut, std::vector<unsigned> &h) {
    const int blocks = split ? nCU * W / 2 : nCU * W, lanes = split ? 512 : 256;
    hipMemset(dOut, 0, (1 + 4 * nCU * W) * sizeof(unsigned));
    hipLaunchKernelGGL(k_stream<K>, dim3(blocks), dim3(lanes), 0, 0, dOut, 12345u, split);  // warm
    hipLaunchKernelGGL(k_stream<K>, dim3(blocks), dim3(lanes), 0, 0, dOut, 12345u, split);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), dOut, (1 + 4 * nCU * W) * sizeof(unsigned), hipMemcpyDeviceToHost);
    if (!split) {
        std::vector<unsigned> t(h.begin() + 1, h.begin() + 1 + 4 * blocks);
        std::nth_element(t.begin(), t.begin() + t.size() / 2, t.end());
        const double med = t[t.size() / 2];
        const double groups = K < 0 ? ITER : 4.0 * ITER;  // (valu: a "group" = 8 vector instructions)
        printf("%-34s W %d  median wave time %9.0f cycles  per group per SIMD %7.2f", label, W, med, med / (groups * W));
        if (K < 0) printf("  = %.2f per vector instruction", med / (groups * W * 8));
        printf("\n");
    } else {
        std::vector<unsigned> tm, tv;
        for (int b = 0; b < blocks; b++)
            for (int w = 0; w < 8; w++) (w >= 4 ? tv : tm).push_back(h[1 + 8 * b + w]);
        std::nth_element(tm.begin(), tm.begin() + tm.size() / 2, tm.end());
        std::nth_element(tv.begin(), tv.begin() + tv.size() / 2, tv.end());
        // W/2 matrix waves and W/2 vector waves per SIMD
        printf("%-34s W %d  matrix waves %9u cycles (%.2f per product per SIMD)  vector waves %9u cycles (%.2f per instruction per SIMD)\n", label, W,
               tm[tm.size() / 2], tm[tm.size() / 2] / (4.0 * ITER * (W / 2)), tv[tv.size() / 2], tv[tv.size() / 2] / (8.0 * ITER * (W / 2)));
    }
}

int main() {
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    nCU = pr.multiProcessorCount;
    unsigned *dOut;
    hipMalloc(&dOut, (1 + 4 * nCU * 8) * sizeof(unsigned));
    std::vector<unsigned> h(1 + 4 * nCU * 8);
    printf("%s, %d CUs; 16x16x32 i8: 4 passes = 16 cycles on the matrix pipe (SQ_VALU_MFMA_BUSY_CYCLES / SQ_INSTS_VALU_MFMA_I8 of k_orient_desc)\n", pr.gcnArchName, nCU);
    for (int W : {1, 2, 4}) {
        run<0>("mfma only", W, 0, dOut, h);
        run<-1>("v_dot4 only", W, 0, dOut, h);
        run<1>("1 mfma + 1 v_dot4 (one wave)", W, 0, dOut, h);
        run<2>("1 mfma + 2 v_dot4 (one wave)", W, 0, dOut, h);
        run<4>("1 mfma + 4 v_dot4 (one wave)", W, 0, dOut, h);
        run<8>("1 mfma + 8 v_dot4 (one wave)", W, 0, dOut, h);
        if (W >= 2) run<0>("split: matrix waves | vector waves", W, 1, dOut, h);
    }
    return 0;
}
