// Issue cost of every VALU opcode the hot kernels are built from, measured on the GPU box (round 6; the first table, round 2, held
// 37 opcodes and quoted cycles at a nominal 2.4 GHz - this one reads the shader clock itself).
//
// Method.  One kernel per opcode: a wave runs ITER x 8 x CHAINS copies of the instruction (inline asm, CHAINS independent
// dependency chains per lane so that latency does not bind), bracketed by s_memtime.  W waves per SIMD on every CU of the chip
// (blocks of 256 lanes = one wave per SIMD, W blocks per CU), so a SIMD issues W x OPS instructions while a wave's clock runs:
//     cycles per wave-instruction per SIMD = (t1 - t0) / (OPS x W)          [shader cycles, s_memtime]
// and, from the HIP events around the launch and the same instruction count, the same figure "at 2.4 GHz" (the form of the
// round-2 table) and the shader clock the chip actually ran at (memtime ticks / event time).  Pairs: two opcodes alternating in
// one stream (do a cheap and a dear opcode overlap, or add?); the figure is per INSTRUCTION of the pair.
// build + run: hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate [waves-per-simd ...]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define ITER 2048
#define CHAINS 8
#define REP 8
#define OPS_PER_WAVE (ITER * CHAINS * REP)

#define TIMED_BEGIN                                                                            \
    unsigned long long t0, t1;                                                                 \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_barrier\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#define TIMED_END(sum)                                                                         \
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");                   \
    if ((threadIdx.x & 63) == 0) out[1 + blockIdx.x * 4 + (threadIdx.x >> 6)] = (unsigned)(t1 - t0); \
    if ((sum) == 0x12345u) out[0] = (sum);

// 32-bit destination chained through itself; %1, %2 = two more VGPR sources
#define K32(name, ASM, N)                                                                      \
    __global__ __launch_bounds__(256) void name(unsigned *out, unsigned seed) {                \
        unsigned a[CHAINS], b = threadIdx.x * 2654435761u + seed, c = b ^ 0x5bd1e995u;         \
        for (int k = 0; k < CHAINS; k++) a[k] = b + k * 0x01010101u;                           \
        TIMED_BEGIN                                                                            \
        for (int it = 0; it < ITER; it++) {                                                    \
            _Pragma("unroll") for (int r = 0; r < REP; r++) {                                  \
                _Pragma("unroll") for (int k = 0; k < CHAINS; k++) {                           \
                    asm volatile(ASM : "+v"(a[k]) : "v"(b), "v"(c) : "vcc", "scc", "s20", "s21", "s22", "s23"); \
                }                                                                              \
            }                                                                                  \
        }                                                                                      \
        unsigned s = 0;                                                                        \
        for (int k = 0; k < CHAINS; k++) s += a[k];                                            \
        TIMED_END(s)                                                                           \
    }

// 64-bit destination chained through itself; %1, %2 = two more 64-bit VGPR sources
#define K64(name, ASM, N)                                                                      \
    __global__ __launch_bounds__(256) void name(unsigned *out, unsigned seed) {                \
        unsigned long long a[CHAINS], b = (threadIdx.x * 2654435761ull + seed) | 0x3ff0000000000000ull, c = b ^ 0x5bd1e995ull; \
        for (int k = 0; k < CHAINS; k++) a[k] = b + k;                                         \
        TIMED_BEGIN                                                                            \
        for (int it = 0; it < ITER; it++) {                                                    \
            _Pragma("unroll") for (int r = 0; r < REP; r++) {                                  \
                _Pragma("unroll") for (int k = 0; k < CHAINS; k++) {                           \
                    asm volatile(ASM : "+v"(a[k]) : "v"(b), "v"(c) : "vcc", "scc", "s20", "s21", "s22", "s23"); \
                }                                                                              \
            }                                                                                  \
        }                                                                                      \
        unsigned s = 0;                                                                        \
        for (int k = 0; k < CHAINS; k++) s += (unsigned)a[k] + (unsigned)(a[k] >> 32);         \
        TIMED_END(s)                                                                           \
    }

// X(kernel, "label", "asm", instructions per asm statement, 32 | 64)
#define OPS(X)                                                                                                         \
    /* VOP2 / VOP1, 32-bit encodings */                                                                                \
    X(k_add, "v_add_u32", "v_add_u32 %0, %0, %1", 1, 32)                                                               \
    X(k_sub, "v_sub_u32", "v_sub_u32 %0, %0, %1", 1, 32)                                                               \
    X(k_subrev, "v_subrev_u32", "v_subrev_u32 %0, %0, %1", 1, 32)                                                      \
    X(k_and, "v_and_b32", "v_and_b32 %0, %0, %1", 1, 32)                                                               \
    X(k_or, "v_or_b32", "v_or_b32 %0, %0, %1", 1, 32)                                                                  \
    X(k_xor, "v_xor_b32", "v_xor_b32 %0, %0, %1", 1, 32)                                                               \
    X(k_not, "v_not_b32", "v_not_b32 %0, %0", 1, 32)                                                                   \
    X(k_mov, "v_mov_b32", "v_mov_b32 %0, %1", 1, 32)                                                                   \
    X(k_lshl, "v_lshlrev_b32 const", "v_lshlrev_b32 %0, 3, %0", 1, 32)                                                 \
    X(k_lshlv, "v_lshlrev_b32 vgpr", "v_lshlrev_b32 %0, %1, %0", 1, 32)                                                \
    X(k_lshr, "v_lshrrev_b32 const", "v_lshrrev_b32 %0, 3, %0", 1, 32)                                                 \
    X(k_ashr, "v_ashrrev_i32 const", "v_ashrrev_i32 %0, 3, %0", 1, 32)                                                 \
    X(k_maxu, "v_max_u32", "v_max_u32 %0, %0, %1", 1, 32)                                                              \
    X(k_minu, "v_min_u32", "v_min_u32 %0, %0, %1", 1, 32)                                                              \
    X(k_maxi, "v_max_i32", "v_max_i32 %0, %0, %1", 1, 32)                                                              \
    X(k_mini, "v_min_i32", "v_min_i32 %0, %0, %1", 1, 32)                                                              \
    X(k_maxu16, "v_max_u16", "v_max_u16 %0, %0, %1", 1, 32)                                                            \
    X(k_addu16, "v_add_u16", "v_add_u16 %0, %0, %1", 1, 32)                                                            \
    X(k_mul24, "v_mul_u32_u24", "v_mul_u32_u24 %0, %0, %1", 1, 32)                                                     \
    X(k_muli24, "v_mul_i32_i24", "v_mul_i32_i24 %0, %0, %1", 1, 32)                                                    \
    X(k_mulf, "v_mul_f32", "v_mul_f32 %0, %0, %1", 1, 32)                                                              \
    X(k_addf, "v_add_f32", "v_add_f32 %0, %0, %1", 1, 32)                                                              \
    X(k_subf, "v_sub_f32", "v_sub_f32 %0, %0, %1", 1, 32)                                                              \
    X(k_fmac, "v_fmac_f32", "v_fmac_f32 %0, %1, %2", 1, 32)                                                            \
    X(k_maxf, "v_max_f32", "v_max_f32 %0, %0, %1", 1, 32)                                                              \
    X(k_minf, "v_min_f32", "v_min_f32 %0, %0, %1", 1, 32)                                                              \
    X(k_cndvcc, "v_cndmask_b32 vcc", "v_cndmask_b32 %0, %0, %1, vcc", 1, 32)                                           \
    X(k_addco, "v_add_co_u32 vcc", "v_add_co_u32 %0, vcc, %0, %1", 1, 32)                                              \
    X(k_addc, "v_addc_co_u32 vcc", "v_addc_co_u32 %0, vcc, %0, %1, vcc", 1, 32)                                        \
    X(k_bfrev, "v_bfrev_b32", "v_bfrev_b32 %0, %0", 1, 32)                                                             \
    X(k_ffbh, "v_ffbh_u32", "v_ffbh_u32 %0, %0", 1, 32)                                                                \
    X(k_ffbl, "v_ffbl_b32", "v_ffbl_b32 %0, %0", 1, 32)                                                                \
    X(k_cvtfu, "v_cvt_f32_u32", "v_cvt_f32_u32 %0, %0", 1, 32)                                                         \
    X(k_cvtfi, "v_cvt_f32_i32", "v_cvt_f32_i32 %0, %0", 1, 32)                                                         \
    X(k_cvtif, "v_cvt_i32_f32", "v_cvt_i32_f32 %0, %0", 1, 32)                                                         \
    X(k_cvtub, "v_cvt_f32_ubyte0", "v_cvt_f32_ubyte0 %0, %0", 1, 32)                                                   \
    X(k_rndne, "v_rndne_f32", "v_rndne_f32 %0, %0", 1, 32)                                                             \
    X(k_trunc, "v_trunc_f32", "v_trunc_f32 %0, %0", 1, 32)                                                             \
    X(k_floor, "v_floor_f32", "v_floor_f32 %0, %0", 1, 32)                                                             \
    X(k_rcp, "v_rcp_f32", "v_rcp_f32 %0, %0", 1, 32)                                                                   \
    X(k_sqrt, "v_sqrt_f32", "v_sqrt_f32 %0, %0", 1, 32)                                                                \
    X(k_rfl, "v_readfirstlane_b32", "v_readfirstlane_b32 s20, %0", 1, 32)                                              \
    X(k_lshrv, "v_lshrrev_b32 vgpr", "v_lshrrev_b32 %0, %1, %0", 1, 32)                                                \
    X(k_minu16, "v_min_u16", "v_min_u16 %0, %0, %1", 1, 32)                                                            \
    X(k_subu16, "v_sub_u16", "v_sub_u16 %0, %0, %1", 1, 32)                                                            \
    X(k_maxi16, "v_max_i16", "v_max_i16 %0, %0, %1", 1, 32)                                                            \
    X(k_lshl16, "v_lshlrev_b16", "v_lshlrev_b16 %0, 1, %0", 1, 32)                                                     \
    X(k_lshr16, "v_lshrrev_b16", "v_lshrrev_b16 %0, 1, %0", 1, 32)                                                     \
    X(k_mullo16, "v_mul_lo_u16", "v_mul_lo_u16 %0, %0, %1", 1, 32)                                                     \
    X(k_addf16, "v_add_f16", "v_add_f16 %0, %0, %1", 1, 32)                                                            \
    X(k_maxf16, "v_max_f16", "v_max_f16 %0, %0, %1", 1, 32)                                                            \
    X(k_mulf16, "v_mul_f16", "v_mul_f16 %0, %0, %1", 1, 32)                                                            \
    X(k_subco, "v_sub_co_u32 vcc", "v_sub_co_u32 %0, vcc, %0, %1", 1, 32)                                              \
    X(k_dot2c, "v_dot2c_i32_i16", "v_dot2c_i32_i16 %0, %1, %2", 1, 32)                                                 \
    X(k_dot4c, "v_dot4c_i32_i8", "v_dot4c_i32_i8 %0, %1, %2", 1, 32)                                                   \
    X(k_pkfmah, "v_pk_fma_f16", "v_pk_fma_f16 %0, %0, %1, %2", 1, 32)                                                  \
    X(k_mad16, "v_mad_u16", "v_mad_u16 %0, %0, %1, %2", 1, 32)                                                         \
    X(k_madmix, "v_fma_mix_f32", "v_fma_mix_f32 %0, %0, %1, %2", 1, 32)                                                \
    X(k_cvtpk, "v_cvt_pk_u8_f32", "v_cvt_pk_u8_f32 %0, %0, %1, %2", 1, 32)                                             \
    X(k_lerp, "v_lerp_u8", "v_lerp_u8 %0, %0, %1, %2", 1, 32)                                                          \
    /* the cheap opcodes in other encodings / with other operand kinds */                                              \
    X(k_add64e, "v_add_u32 e64", "v_add_u32_e64 %0, %0, %1", 1, 32)                                                    \
    X(k_and64e, "v_and_b32 e64", "v_and_b32_e64 %0, %0, %1", 1, 32)                                                    \
    X(k_xor64e, "v_xor_b32 e64", "v_xor_b32_e64 %0, %0, %1", 1, 32)                                                    \
    X(k_mov64e, "v_mov_b32 e64", "v_mov_b32_e64 %0, %1", 1, 32)                                                        \
    X(k_mulf64e, "v_mul_f32 e64", "v_mul_f32_e64 %0, %0, %1", 1, 32)                                                   \
    X(k_addlit, "v_add_u32 literal", "v_add_u32 %0, 0x01020304, %0", 1, 32)                                            \
    X(k_andlit, "v_and_b32 literal", "v_and_b32 %0, 0x7f7f7f7f, %0", 1, 32)                                            \
    X(k_addinl, "v_add_u32 inline", "v_add_u32 %0, 7, %0", 1, 32)                                                      \
    X(k_andsgpr, "v_and_b32 sgpr", "v_and_b32 %0, s24, %0", 1, 32)                                                     \
    X(k_addself, "v_add_u32 a,a,a", "v_add_u32 %0, %0, %0", 1, 32)                                                     \
    /* VOP3 */                                                                                                         \
    X(k_bcnt, "v_bcnt_u32_b32", "v_bcnt_u32_b32 %0, %1, %0", 1, 32)                                                    \
    X(k_bfi, "v_bfi_b32", "v_bfi_b32 %0, %1, %0, %2", 1, 32)                                                           \
    X(k_bfe, "v_bfe_u32", "v_bfe_u32 %0, %0, 4, 8", 1, 32)                                                             \
    X(k_andor, "v_and_or_b32", "v_and_or_b32 %0, %0, %1, %2", 1, 32)                                                   \
    X(k_or3, "v_or3_b32", "v_or3_b32 %0, %0, %1, %2", 1, 32)                                                           \
    X(k_add3, "v_add3_u32", "v_add3_u32 %0, %0, %1, %2", 1, 32)                                                        \
    X(k_lshladd, "v_lshl_add_u32", "v_lshl_add_u32 %0, %0, 2, %1", 1, 32)                                              \
    X(k_addlshl, "v_add_lshl_u32", "v_add_lshl_u32 %0, %0, %1, 2", 1, 32)                                              \
    X(k_lshlor, "v_lshl_or_b32", "v_lshl_or_b32 %0, %0, 6, %1", 1, 32)                                                 \
    X(k_xad, "v_xad_u32", "v_xad_u32 %0, %0, %1, %2", 1, 32)                                                           \
    X(k_bitop3, "v_bitop3_b32", "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96", 1, 32)                                      \
    X(k_min3, "v_min3_u32", "v_min3_u32 %0, %0, %1, %2", 1, 32)                                                        \
    X(k_max3, "v_max3_u32", "v_max3_u32 %0, %0, %1, %2", 1, 32)                                                        \
    X(k_max3u16, "v_max3_u16", "v_max3_u16 %0, %0, %1, %2", 1, 32)                                                     \
    X(k_med3, "v_med3_u32", "v_med3_u32 %0, %0, %1, %2", 1, 32)                                                        \
    X(k_sad, "v_sad_u8", "v_sad_u8 %0, %0, %1, %2", 1, 32)                                                             \
    X(k_sadhi, "v_sad_hi_u8", "v_sad_hi_u8 %0, %0, %1, %2", 1, 32)                                                     \
    X(k_sadu16, "v_sad_u16", "v_sad_u16 %0, %0, %1, %2", 1, 32)                                                        \
    X(k_sadu32, "v_sad_u32", "v_sad_u32 %0, %0, %1, %2", 1, 32)                                                        \
    X(k_msad, "v_msad_u8", "v_msad_u8 %0, %0, %1, %2", 1, 32)                                                          \
    X(k_perm, "v_perm_b32", "v_perm_b32 %0, %0, %1, %2", 1, 32)                                                        \
    X(k_alignbit, "v_alignbit_b32", "v_alignbit_b32 %0, %0, %1, 16", 1, 32)                                            \
    X(k_alignbyte, "v_alignbyte_b32", "v_alignbyte_b32 %0, %0, %1, 1", 1, 32)                                          \
    X(k_dot4, "v_dot4_u32_u8", "v_dot4_u32_u8 %0, %0, %1, %2", 1, 32)                                                  \
    X(k_dot2, "v_dot2_u32_u16", "v_dot2_u32_u16 %0, %0, %1, %2", 1, 32)                                                \
    X(k_mad24, "v_mad_u32_u24", "v_mad_u32_u24 %0, %0, %1, %2", 1, 32)                                                 \
    X(k_madi24, "v_mad_i32_i24", "v_mad_i32_i24 %0, %0, %1, %2", 1, 32)                                                \
    X(k_mullo, "v_mul_lo_u32", "v_mul_lo_u32 %0, %0, %1", 1, 32)                                                       \
    X(k_mulhi, "v_mul_hi_u32", "v_mul_hi_u32 %0, %0, %1", 1, 32)                                                       \
    X(k_fma, "v_fma_f32", "v_fma_f32 %0, %0, %1, %2", 1, 32)                                                           \
    X(k_mbcntlo, "v_mbcnt_lo_u32_b32", "v_mbcnt_lo_u32_b32 %0, %1, %0", 1, 32)                                         \
    X(k_mbcnthi, "v_mbcnt_hi_u32_b32", "v_mbcnt_hi_u32_b32 %0, %1, %0", 1, 32)                                         \
    X(k_cnd64, "v_cndmask_b32 e64 sgpr", "v_cndmask_b32_e64 %0, %0, %1, s[22:23]", 1, 32)                              \
    X(k_readlane, "v_readlane_b32", "v_readlane_b32 s20, %0, 5", 1, 32)                                                \
    X(k_writelane, "v_writelane_b32", "v_writelane_b32 %0, s24, 5", 1, 32)                                             \
    X(k_lshle64, "v_lshlrev_b32 e64", "v_lshlrev_b32_e64 %0, 3, %0", 1, 32)                                            \
    /* VOP3P */                                                                                                        \
    X(k_pkaddu16, "v_pk_add_u16", "v_pk_add_u16 %0, %0, %1", 1, 32)                                                    \
    X(k_pksubu16, "v_pk_sub_u16", "v_pk_sub_u16 %0, %0, %1", 1, 32)                                                    \
    X(k_pksubi16, "v_pk_sub_i16", "v_pk_sub_i16 %0, %0, %1", 1, 32)                                                    \
    X(k_pkmaxu16, "v_pk_max_u16", "v_pk_max_u16 %0, %0, %1", 1, 32)                                                    \
    X(k_pkminu16, "v_pk_min_u16", "v_pk_min_u16 %0, %0, %1", 1, 32)                                                    \
    X(k_pkmaxi16, "v_pk_max_i16", "v_pk_max_i16 %0, %0, %1", 1, 32)                                                    \
    X(k_pklshl16, "v_pk_lshlrev_b16", "v_pk_lshlrev_b16 %0, 1, %0", 1, 32)                                             \
    X(k_pkmullo, "v_pk_mul_lo_u16", "v_pk_mul_lo_u16 %0, %0, %1", 1, 32)                                               \
    X(k_pkmad, "v_pk_mad_u16", "v_pk_mad_u16 %0, %0, %1, %2", 1, 32)                                                   \
    X(k_pkmin3h, "v_pk_minimum3_f16", "v_pk_minimum3_f16 %0, %0, %1, %2", 1, 32)                                       \
    X(k_pkmax3h, "v_pk_maximum3_f16", "v_pk_maximum3_f16 %0, %0, %1, %2", 1, 32)                                       \
    X(k_pkminh, "v_pk_min_f16", "v_pk_min_f16 %0, %0, %1", 1, 32)                                                      \
    X(k_pkmaxh, "v_pk_max_f16", "v_pk_max_f16 %0, %0, %1", 1, 32)                                                      \
    X(k_pkaddh, "v_pk_add_f16", "v_pk_add_f16 %0, %0, %1", 1, 32)                                                      \
    /* VOPC */                                                                                                         \
    X(k_cmp, "v_cmp_gt_u32 vcc", "v_cmp_gt_u32 vcc, %0, %1", 1, 32)                                                    \
    X(k_cmp64, "v_cmp_gt_u32 e64", "v_cmp_gt_u32_e64 s[22:23], %0, %1", 1, 32)                                         \
    X(k_cmplti, "v_cmp_lt_i32 vcc", "v_cmp_lt_i32 vcc, %0, %1", 1, 32)                                                 \
    X(k_cmpeq, "v_cmp_eq_u32 vcc", "v_cmp_eq_u32 vcc, %0, %1", 1, 32)                                                  \
    X(k_cmpu16, "v_cmp_gt_u16 vcc", "v_cmp_gt_u16 vcc, %0, %1", 1, 32)                                                 \
    X(k_cmpf, "v_cmp_lt_f32 vcc", "v_cmp_lt_f32 vcc, %0, %1", 1, 32)                                                   \
    X(k_cmpsdwa, "v_cmp_gt_u32 sdwa", "v_cmp_gt_u32_sdwa s[22:23], %0, %1 src0_sel:WORD_1 src1_sel:DWORD", 1, 32)      \
    X(k_cmpgei_sdwa, "v_cmp_ge_i32 sdwa", "v_cmp_ge_i32_sdwa vcc, %0, %1 src0_sel:BYTE_0 src1_sel:DWORD", 1, 32)       \
    /* SDWA / DPP */                                                                                                   \
    X(k_addsdwa, "v_add_u32 sdwa", "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1", 1, 32) \
    X(k_andsdwa, "v_and_b32 sdwa", "v_and_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0", 1, 32)  \
    X(k_subsdwa, "v_sub_u32 sdwa", "v_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2", 1, 32) \
    X(k_lshlsdwa, "v_lshlrev_b32 sdwa", "v_lshlrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1", 1, 32) \
    X(k_muli24sdwa, "v_mul_i32_i24 sdwa", "v_mul_i32_i24_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1", 1, 32) \
    X(k_movdpp, "v_mov_b32 dpp quad", "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", 1, 32)    \
    X(k_movdpps, "v_mov_b32 dpp row_shr", "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf", 1, 32)          \
    X(k_movdppb, "v_mov_b32 dpp row_bcast", "v_mov_b32_dpp %0, %1 row_bcast:15 row_mask:0xa bank_mask:0xf", 1, 32)     \
    X(k_adddpp, "v_add_u32 dpp", "v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf", 1, 32)               \
    X(k_mindpp, "v_min_u32 dpp", "v_min_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf", 1, 32)               \
    X(k_ordpp, "v_or_b32 dpp", "v_or_b32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf", 1, 32)                  \
    /* 64-bit */                                                                                                       \
    X(k_mov64, "v_mov_b64", "v_mov_b64 %0, %1", 1, 64)                                                                 \
    X(k_pkmov, "v_pk_mov_b32", "v_pk_mov_b32 %0, %0, %1", 1, 64)                                                       \
    X(k_lshladd64, "v_lshl_add_u64", "v_lshl_add_u64 %0, %0, 2, %1", 1, 64)                                            \
    X(k_lshl64, "v_lshlrev_b64", "v_lshlrev_b64 %0, 1, %0", 1, 64)                                                     \
    X(k_addf64, "v_add_f64", "v_add_f64 %0, %0, %1", 1, 64)                                                            \
    X(k_mulf64, "v_mul_f64", "v_mul_f64 %0, %0, %1", 1, 64)                                                            \
    X(k_fmaf64, "v_fma_f64", "v_fma_f64 %0, %0, %1, %2", 1, 64)                                                        \
    X(k_pkaddf32, "v_pk_add_f32", "v_pk_add_f32 %0, %0, %1", 1, 64)                                                    \
    X(k_pkmulf32, "v_pk_mul_f32", "v_pk_mul_f32 %0, %0, %1", 1, 64)                                                    \
    X(k_pkfmaf32, "v_pk_fma_f32", "v_pk_fma_f32 %0, %0, %1, %2", 1, 64)                                                \
    X(k_cmpu64, "v_cmp_lt_u64 vcc", "v_cmp_lt_u64 vcc, %0, %1", 1, 64)                                                 \
    /* pairs: a cheap and a dear opcode alternating (figures are per instruction of the pair) */                        \
    X(p_add_add, "pair v_add_u32 + v_and_b32", "v_add_u32 %0, %0, %1\n v_and_b32 %0, %0, %2", 2, 32)                   \
    X(p_add_sad, "pair v_add_u32 + v_sad_u8", "v_add_u32 %0, %0, %1\n v_sad_u8 %0, %0, %1, %2", 2, 32)                 \
    X(p_and_pkmax, "pair v_and_b32 + v_pk_max_u16", "v_and_b32 %0, %0, %1\n v_pk_max_u16 %0, %0, %2", 2, 32)           \
    X(p_xor_bcnt, "pair v_xor_b32 + v_bcnt_u32_b32", "v_xor_b32 %0, %0, %1\n v_bcnt_u32_b32 %0, %0, %2", 2, 32)        \
    X(p_mov_min3, "pair v_mov_b32 + v_min3_u32", "v_mov_b32 %0, %1\n v_min3_u32 %0, %0, %1, %2", 2, 32)                \
    X(p_add_cmp, "pair v_add_u32 + v_cmp_gt_u32", "v_add_u32 %0, %0, %1\n v_cmp_gt_u32 vcc, %0, %2", 2, 32)            \
    X(p_sad_pkmax, "pair v_sad_u8 + v_pk_max_u16", "v_sad_u8 %0, %0, %1, %2\n v_pk_max_u16 %0, %0, %2", 2, 32)         \
    X(p_add_mulf, "pair v_add_u32 + v_mul_f32", "v_add_u32 %0, %0, %1\n v_mul_f32 %0, %0, %2", 2, 32)                  \
    X(p_add_fma, "pair v_add_u32 + v_fma_f32", "v_add_u32 %0, %0, %1\n v_fma_f32 %0, %0, %1, %2", 2, 32)               \
    X(p_add_salu, "pair v_add_u32 + s_add_u32", "v_add_u32 %0, %0, %1\n s_add_u32 s20, s20, 1", 2, 32)                 \
    X(p_sad_salu, "pair v_sad_u8 + s_add_u32", "v_sad_u8 %0, %0, %1, %2\n s_add_u32 s20, s20, 1", 2, 32)               \
    X(p_add3x_sad, "3 v_add_u32 + 1 v_sad_u8", "v_add_u32 %0, %0, %1\n v_and_b32 %0, %0, %2\n v_xor_b32 %0, %0, %1\n v_sad_u8 %0, %0, %1, %2", 4, 32) \
    X(p_bitop_sad, "pair v_bitop3_b32 + v_sad_u8", "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96\n v_sad_u8 %0, %0, %1, %2", 2, 32) \
    X(p_lshr_and, "pair v_lshrrev_b32 + v_and_b32", "v_lshrrev_b32 %0, 8, %0\n v_and_b32 %0, %0, %1", 2, 32)           \
    X(p_fast4_slow1, "4 cheap + 1 v_pk_max_u16", "v_add_u32 %0, %0, %1\n v_and_b32 %0, %0, %2\n v_xor_b32 %0, %0, %1\n v_sub_u32 %0, %0, %2\n v_pk_max_u16 %0, %0, %1", 5, 32) \
    X(p_fast1_slow3, "1 v_add_u32 + 3 dear", "v_add_u32 %0, %0, %1\n v_sad_u8 %0, %0, %1, %2\n v_pk_max_u16 %0, %0, %1\n v_alignbit_b32 %0, %0, %1, 16", 4, 32) \
    X(p_and_cmpcnd, "v_cmp_gt_u32 + v_cndmask vcc", "v_cmp_gt_u32 vcc, %1, %2\n v_cndmask_b32 %0, %0, %1, vcc", 2, 32)

#define DEF(k, label, ASM, n, w) DEF_##w(k, ASM, n)
#define DEF_32(k, ASM, n) K32(k, ASM, n)
#define DEF_64(k, ASM, n) K64(k, ASM, n)
OPS(DEF)

struct Op {
    const char *label;
    void (*kern)(unsigned *, unsigned);
    int perStmt;
};
#define ROW(k, label, ASM, n, w) {label, k, n},
static const Op ops[] = {OPS(ROW)};

static void run(const Op &op, unsigned *out, unsigned *host, int wavesPerSimd, int nCU) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int blocks = nCU * wavesPerSimd;  // a block = 4 waves = one per SIMD; W blocks per CU
    op.kern<<<blocks, 256>>>(out, 1u);
    hipDeviceSynchronize();
    const int reps = 3;
    double cyc = 0;
    float msTotal = 0;
    for (int r = 0; r < reps; r++) {
        hipEventRecord(e0);
        op.kern<<<blocks, 256>>>(out, 2u + r);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        msTotal += ms;
        hipMemcpy(host, out, sizeof(unsigned) * (1 + blocks * 4), hipMemcpyDeviceToHost);
        std::vector<unsigned> t(host + 1, host + 1 + blocks * 4);
        std::sort(t.begin(), t.end());
        cyc += t[t.size() / 2];  // median wave
    }
    cyc /= reps;
    const double inst = (double)OPS_PER_WAVE * op.perStmt;
    const double perInstMemtime = cyc / (inst * wavesPerSimd);
    const double sec = msTotal * 1e-3 / reps;
    const double perInstEvent = sec * 2.4e9 / (inst * wavesPerSimd);
    printf("%-34s W=%d  %6.2f cycles (s_memtime)  %6.2f at 2.4 GHz by events  wave %8.0f ticks in %7.3f ms = %5.0f MHz\n", op.label,
           wavesPerSimd, perInstMemtime, perInstEvent, cyc, sec * 1e3, cyc / sec * 1e-6);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
}

int main(int argc, char **argv) {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int nCU = prop.multiProcessorCount;
    printf("# %s, %d CUs, clockRate %d kHz; %d instructions per wave and kernel, %d chains per lane\n", prop.gcnArchName, nCU, prop.clockRate,
           OPS_PER_WAVE, CHAINS);
    unsigned *out, *host = (unsigned *)malloc(sizeof(unsigned) * (1 + 256 * 8 * 4 + 64));
    hipMalloc(&out, sizeof(unsigned) * (1 + 256 * 8 * 4 + 64));
    std::vector<int> ws;
    for (int i = 1; i < argc; i++) ws.push_back(atoi(argv[i]));
    if (ws.empty()) ws = {8};
    for (int w : ws) {
        if (w < 1 || w > 8) continue;
        for (const Op &op : ops) run(op, out, host, w, nCU);
    }
    return 0;
}
