// Issue rate of the VALU instructions the kernels lean on, measured on the GPU box: every kernel runs the same number of
// wave-level instructions (8 independent chains per lane, so latency does not bind) on 8 waves per SIMD of every CU; the
// time per instruction per SIMD in cycles (at the nominal 2.4 GHz) is printed.  4 = full rate for wave64 on a 16-lane SIMD.
// build + run: hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>

#include <cstdio>

#define ITER 256
#define CHAINS 8
#define OPS_PER_WAVE (ITER * CHAINS * 8)

#define KERNEL(name, ASM)                                                                     \
    __global__ __launch_bounds__(256) void name(unsigned *out, unsigned seed) {               \
        unsigned a[CHAINS], b = threadIdx.x * 2654435761u + seed, c = b ^ 0x5bd1e995u;        \
        for (int k = 0; k < CHAINS; k++) a[k] = b + k;                                        \
        for (int it = 0; it < ITER; it++) {                                                   \
            _Pragma("unroll") for (int r = 0; r < 8; r++) {                                   \
                _Pragma("unroll") for (int k = 0; k < CHAINS; k++) { asm volatile(ASM : "+v"(a[k]) : "v"(b), "v"(c)); } \
            }                                                                                 \
        }                                                                                     \
        unsigned s = 0;                                                                       \
        for (int k = 0; k < CHAINS; k++) s += a[k];                                           \
        if (s == 0x12345u) out[0] = s;                                                        \
    }

KERNEL(k_add, "v_add_u32 %0, %0, %1")
KERNEL(k_sad, "v_sad_u8 %0, %0, %1, %2")
KERNEL(k_sadhi, "v_sad_hi_u8 %0, %0, %1, %2")
KERNEL(k_pkmax, "v_pk_max_u16 %0, %0, %1")
KERNEL(k_pkmin, "v_pk_min_u16 %0, %0, %1")
KERNEL(k_alignbit, "v_alignbit_b32 %0, %0, %1, 16")
KERNEL(k_min3, "v_min3_u32 %0, %0, %1, %2")
KERNEL(k_max3, "v_max3_i32 %0, %0, %1, %2")
KERNEL(k_dot4, "v_dot4_u32_u8 %0, %0, %1, %2")
KERNEL(k_dot2, "v_dot2_u32_u16 %0, %0, %1, %2")
KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %2")
KERNEL(k_mad24, "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL(k_mul24, "v_mul_u32_u24 %0, %0, %1")
KERNEL(k_mullo, "v_mul_lo_u32 %0, %0, %1")
KERNEL(k_addsdwa, "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1")
KERNEL(k_lshl_or, "v_lshl_or_b32 %0, %0, 6, %1")
KERNEL(k_mbcnt, "v_mbcnt_lo_u32_b32 %0, %1, %0")
KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL(k_cndmask64, "v_cndmask_b32_e64 %0, %0, %1, s[20:21]")
KERNEL(k_cndmask64v, "v_cndmask_b32_e64 %0, %0, %1, vcc")
KERNEL(k_addc, "v_addc_co_u32 %0, vcc, %0, %1, vcc")
KERNEL(k_cndmask_afterw, "v_cmp_gt_u32 vcc, %1, %2\n v_cndmask_b32 %0, %0, %1, vcc")
KERNEL(k_and, "v_and_b32 %0, %0, %1")
KERNEL(k_max, "v_max_u32 %0, %0, %1")
KERNEL(k_lshl, "v_lshlrev_b32 %0, 3, %0")
KERNEL(k_add3, "v_add3_u32 %0, %0, %1, %2")
KERNEL(k_mov, "v_mov_b32 %0, %1")
KERNEL(k_movdpp, "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
KERNEL(k_adddpp, "v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL(k_cmp, "v_cmp_gt_u32 vcc, %0, %1")
KERNEL(k_cmp64, "v_cmp_gt_u32_e64 s[22:23], %0, %1")
KERNEL(k_cmpsdwa, "v_cmp_gt_u32_sdwa s[22:23], %0, %1 src0_sel:WORD_1 src1_sel:DWORD")
KERNEL(k_bfe, "v_bfe_u32 %0, %0, 4, 8")
KERNEL(k_fma, "v_fma_f32 %0, %0, %1, %2")
KERNEL(k_mulf, "v_mul_f32 %0, %0, %1")
KERNEL(k_cvt, "v_cvt_i32_f32 %0, %0")
KERNEL(k_rndne, "v_rndne_f32 %0, %0")

template <class F>
static void run(const char *name, F kern, unsigned *out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int blocks = 256 * 8;  // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    kern<<<blocks, 256>>>(out, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; r++) kern<<<blocks, 256>>>(out, 2u + r);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double instPerSimd = 5.0 * (double)blocks * 4 / (256.0 * 4) * OPS_PER_WAVE;
    printf("%-14s %7.3f ms  %5.2f cycles per wave instruction per SIMD (2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / instPerSimd);
}

int main() {
    unsigned *out;
    hipMalloc(&out, 64);
    run("v_add_u32", k_add, out);
    run("v_sad_u8", k_sad, out);
    run("v_sad_hi_u8", k_sadhi, out);
    run("v_pk_max_u16", k_pkmax, out);
    run("v_pk_min_u16", k_pkmin, out);
    run("v_alignbit", k_alignbit, out);
    run("v_min3_u32", k_min3, out);
    run("v_max3_i32", k_max3, out);
    run("v_dot4_u32_u8", k_dot4, out);
    run("v_dot2_u32_u16", k_dot2, out);
    run("v_perm_b32", k_perm, out);
    run("v_mad_u32_u24", k_mad24, out);
    run("v_mul_u32_u24", k_mul24, out);
    run("v_mul_lo_u32", k_mullo, out);
    run("v_add_sdwa", k_addsdwa, out);
    run("v_lshl_or", k_lshl_or, out);
    run("v_mbcnt_lo", k_mbcnt, out);
    run("v_cndmask vcc", k_cndmask, out);
    run("v_cndmask e64", k_cndmask64, out);
    run("v_cndmask e64 vcc", k_cndmask64v, out);
    run("v_addc_co vcc", k_addc, out);
    run("cmp+cndmask vcc", k_cndmask_afterw, out);
    run("v_and_b32", k_and, out);
    run("v_max_u32", k_max, out);
    run("v_lshlrev", k_lshl, out);
    run("v_add3_u32", k_add3, out);
    run("v_mov_b32", k_mov, out);
    run("v_mov_dpp", k_movdpp, out);
    run("v_add_dpp", k_adddpp, out);
    run("v_cmp vcc", k_cmp, out);
    run("v_cmp e64", k_cmp64, out);
    run("v_cmp sdwa", k_cmpsdwa, out);
    run("v_bfe_u32", k_bfe, out);
    run("v_fma_f32", k_fma, out);
    run("v_mul_f32", k_mulf, out);
    run("v_cvt_i32_f32", k_cvt, out);
    run("v_rndne_f32", k_rndne, out);
    return 0;
}
