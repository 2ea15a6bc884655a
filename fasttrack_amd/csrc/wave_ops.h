// Device helpers shared by the gfx950 kernels (wave64): explicit global loads, 24-bit multiplies, DPP reductions.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Image rows (and other read-only inputs reached through a pointer that was itself loaded from memory or selected
// between two bases) are read through explicit global-address-space loads: such a pointer is generic to the compiler,
// which then emits flat_load - and a flat load counts on the LDS counter as well, so every LDS wait would also wait
// for the image loads in flight.
template <class T>
__device__ __forceinline__ T gload(const void *p) {
    return *(const __attribute__((address_space(1))) T *)p;
}

template <class T>
__device__ __forceinline__ void gstore(void *p, T v) {
    *(__attribute__((address_space(1))) T *)p = v;
}

// The wave index of a thread, as a scalar: it is uniform by construction, which the compiler cannot see (it derives
// from threadIdx); with it everything a one-item-per-wave kernel looks up for its item becomes SALU work and scalar
// loads instead of 64 identical VALU lanes.
__device__ __forceinline__ int wave_index() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

// Sum / minimum over the 64 lanes, returned wave-uniform: four DPP steps (lane pairs, quads, half rows, rows of 16;
// all four are permutations inside a row, so no lane reads an invalid source) and one v_readlane per row - instead of
// six shuffles through the LDS crossbar with their address arithmetic.
__device__ __forceinline__ int wave_sum_i32(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);  // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);  // row_mirror
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) +
           __builtin_amdgcn_readlane(v, 48);
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true));
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true));
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true));
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true));
    return min(min((unsigned)__builtin_amdgcn_readlane((int)v, 0), (unsigned)__builtin_amdgcn_readlane((int)v, 16)),
               min((unsigned)__builtin_amdgcn_readlane((int)v, 32), (unsigned)__builtin_amdgcn_readlane((int)v, 48)));
}

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#define FT_MIN64_STEP(ctrl)                                                                                   \
    {                                                                                                         \
        const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)v, ctrl, 0xF, 0xF, true); \
        const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(v >> 32), ctrl, 0xF, 0xF, true);   \
        const unsigned long long w = ((unsigned long long)hi << 32) | lo;                                     \
        v = w < v ? w : v;                                                                                    \
    }
    FT_MIN64_STEP(0xB1) FT_MIN64_STEP(0x4E) FT_MIN64_STEP(0x141) FT_MIN64_STEP(0x140)
#undef FT_MIN64_STEP
    unsigned long long best = ~0ull;
#pragma unroll
    for (int row = 0; row < 64; row += 16) {
        const unsigned long long w = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(v >> 32), row) << 32) |
                                     (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, row);
        best = w < best ? w : best;
    }
    return best;
}

// minimum over the 16 lanes of a ROW, returned in every lane of the row (the four DPP steps above stay inside a row)
__device__ __forceinline__ unsigned long long row_min_u64(unsigned long long v) {
#define FT_MIN64_STEP(ctrl)                                                                                   \
    {                                                                                                         \
        const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)v, ctrl, 0xF, 0xF, true); \
        const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(v >> 32), ctrl, 0xF, 0xF, true);   \
        const unsigned long long w = ((unsigned long long)hi << 32) | lo;                                     \
        v = w < v ? w : v;                                                                                    \
    }
    FT_MIN64_STEP(0xB1) FT_MIN64_STEP(0x4E) FT_MIN64_STEP(0x141) FT_MIN64_STEP(0x140)
#undef FT_MIN64_STEP
    return v;
}

// maximum over the row of 16 lanes, in every lane of it
__device__ __forceinline__ int row_max_i32(int v) {
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xF, 0xF, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xF, 0xF, false));
    return v;
}

// Low 32 bits of the product of two operands that fit 24 bits, as ONE full-rate instruction.  (__mul24 is dissolved
// into a plain multiply once the optimiser has proven the operand ranges, and instruction selection then falls back
// to the quarter-rate 32-bit v_mul_lo_u32 whenever it cannot re-derive them.)
__device__ __forceinline__ int vmul24(int a, int b) {
    int r;
    asm("v_mul_i32_i24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// a * b + c with 24-bit a, b: one full-rate instruction (b wave-uniform)
__device__ __forceinline__ int vmad24(int a, int bUniform, int c) {
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(bUniform), "v"(c));
    return r;
}
