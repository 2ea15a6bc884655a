// glibc's single-precision cosf / sinf / logf, restated so that the device computes exactly what the host libm of an
// ORB-SLAM3 binary returns.
//
// Why: the reference's rBRIEF rotation is `float a = (float)cos(angle), b = (float)sin(angle)` with a float `angle`
// under `using namespace std` (/root/reference/src/ORBextractor.cc:34,73-74) - that binds to std::cos(float) = cosf,
// and MapPoint::PredictScale's `log(ratio)` (/root/reference/src/MapPoint.cc:539) to logf.  glibc's float routines
// are NOT correctly rounded (<= 0.56 ulp), so "(float)cos((double)x)" differs from cosf(x) on 2.6 % of the angles and
// a descriptor sample can land one pixel away once in ~500 k keypoints.  They are, however, short double-precision
// computations with one final narrowing, which a GPU reproduces bit for bit.
//
// Third party restated here: GNU libc 2.35 (Ubuntu 2.35-0ubuntu3.11 in this image and on the GPU box),
// sysdeps/ieee754/flt-32/{s_sincosf.h,s_sincosf_data.c,s_sinf.c,s_cosf.c,e_logf.c,e_logf_data.c} - the ARM
// "optimized routines" algorithms of Szabolcs Nagy / Wilco Dijkstra, unchanged in glibc 2.28 .. 2.40.  On x86-64 the
// symbols are IFUNCs; every CPU with AVX2+FMA (any host that can carry an MI355X) selects the `_fma` build, i.e. the
// same C source compiled with -mfma, where GCC contracts each `a + b * c` into one fused operation.  That is the
// variant written below with explicit fma (FT_LIBM_CONTRACT 1); FT_LIBM_CONTRACT 0 gives the `_sse2` build for the
// checker (tests/cpp/test_libm_f32.cpp reports how often the two differ).  x86-64 builds take the !TOINT_INTRINSICS
// branch of reduce_fast (2/pi pre-scaled by 2^24, quadrant in bits 24..31 of a truncated product).
//
// Checked exhaustively: tests/cpp/test_libm_f32.cpp compares every float of [0, 2 pi] (sin, cos) and of (0, 1e4]
// (log) with the host's cosf / sinf / logf; tests/test_libm_f32_cpu.py runs a strided version in the CPU suite and the
// same sweep on the device against the GPU box's libm under -m gpu.
#pragma once
#include <stdint.h>
#if defined(__HIPCC__)
#define FT_LM_HD __host__ __device__ __forceinline__
#else
#define FT_LM_HD inline
#endif
#ifndef FT_LIBM_CONTRACT
#define FT_LIBM_CONTRACT 1
#endif

namespace ft_libm {

// a + b * c as the selected glibc build evaluates it.  The non-contracted form goes through volatile-free helpers
// that the compiler must not fuse: the translation units including this header with FT_LIBM_CONTRACT 0 are host
// only and compiled with -ffp-contract=off.
FT_LM_HD double mad(double b, double c, double a) {
#if FT_LIBM_CONTRACT
    return __builtin_fma(b, c, a);
#else
    return a + b * c;
#endif
}

FT_LM_HD uint32_t f32_bits(float f) { return __builtin_bit_cast(uint32_t, f); }
FT_LM_HD float bits_f32(uint32_t u) { return __builtin_bit_cast(float, u); }

// __sincosf_table[0]: polynomials of s_sincosf_data.c; table[1] is the same with the cosine coefficients negated
// (used when bit 1 of the quadrant is set).
struct SinCosPoly {
    double c0, c1, c2, c3, c4, s1, s2, s3;
};
FT_LM_HD SinCosPoly sincos_poly(bool negate_cos) {
    const double k = negate_cos ? -1.0 : 1.0;
    return {k * 0x1p0,
            k * -0x1.ffffffd0c621cp-2,
            k * 0x1.55553e1068f19p-5,
            k * -0x1.6c087e89a359dp-10,
            k * 0x1.99343027bf8c3p-16,
            -0x1.555545995a603p-3,
            0x1.1107605230bc4p-7,
            -0x1.994eb3774cf24p-13};
}

// sinf_poly of s_sincosf.h: sine polynomial when the quadrant is even, cosine when odd.
FT_LM_HD float sinf_poly(double x, double x2, const SinCosPoly &p, int n) {
    if ((n & 1) == 0) {
        const double x3 = x * x2;
        const double s1 = mad(x2, p.s3, p.s2);
        const double x7 = x3 * x2;
        const double s = mad(x3, p.s1, x);
        return (float)mad(x7, s1, s);
    }
    const double x4 = x2 * x2;
    const double c2 = mad(x2, p.c4, p.c3);
    const double c1 = mad(x2, p.c1, p.c0);
    const double x6 = x4 * x2;
    const double c = mad(x4, p.c2, c1);
    return (float)mad(x6, c2, c);
}

FT_LM_HD uint32_t abstop12(float x) { return (f32_bits(x) >> 20) & 0x7ff; }

// sinf (cosine = false) / cosf (cosine = true) of s_sinf.c / s_cosf.c for |y| < 120; larger arguments (the
// reduce_large path) cannot occur: the callers pass an angle of [0, 2 pi].
FT_LM_HD float sincosf_one(float y, bool cosine) {
    double x = (double)y;
    const uint32_t top = abstop12(y);
    if (top < abstop12(0x1.921FB6p-1f)) {  // pi / 4, compared on the top 12 bits as glibc does
        if (top < abstop12(0x1p-12f)) return cosine ? 1.0f : y;
        return sinf_poly(x, x * x, sincos_poly(false), cosine ? 1 : 0);
    }
    // reduce_fast, !TOINT_INTRINSICS
    const double r = x * 0x1.45F306DC9C883p+23;
    const int n = (int)(((int32_t)r + 0x800000) >> 24);
    x = mad(-(double)n, 0x1.921FB54442D18p0, x);
    const double s = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;  // sign[] = {1, -1, -1, 1}
    return sinf_poly(x * s, x * x, sincos_poly((n & 2) != 0), cosine ? (n ^ 1) : n);
}

FT_LM_HD float cosf_glibc(float y) { return sincosf_one(y, true); }
FT_LM_HD float sinf_glibc(float y) { return sincosf_one(y, false); }

// __logf_data of e_logf_data.c: 16 (1/c, log c) pairs, ln 2 and the degree-3 log1p polynomial.
FT_LM_HD void logf_tab(int i, double &invc, double &logc) {
    constexpr double T[16][2] = {
        {0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2},
        {0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2},  {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3},
        {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3},
        {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4},
        {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5}, {0x1p+0, 0x0p+0},
        {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5},  {0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4},
        {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3},
        {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},  {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2}};
    invc = T[i][0];
    logc = T[i][1];
}

// __logf of e_logf.c.  Zero, negatives, infinities and NaN return what glibc returns (without errno / exceptions);
// subnormals are normalised as there.
FT_LM_HD float logf_glibc(float x) {
    uint32_t ix = f32_bits(x);
    if (ix == 0x3f800000u) return 0.0f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
        if (ix * 2 == 0) return -__builtin_inff();
        if (ix == 0x7f800000u) return x;
        if ((ix & 0x80000000u) || ix * 2 >= 0xff000000u) return __builtin_nanf("");
        ix = f32_bits(x * 0x1p23f);
        ix -= 23u << 23;
    }
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> (23 - 4)) % 16);
    const int k = (int32_t)tmp >> 23;
    const uint32_t iz = ix - (tmp & (0x1ffu << 23));
    double invc, logc;
    logf_tab(i, invc, logc);
    const double z = (double)bits_f32(iz);
    const double r = mad(z, invc, -1.0);
    const double y0 = mad((double)k, 0x1.62e42fefa39efp-1, logc);
    const double r2 = r * r;
    double y = mad(0x1.5575b0be00b6ap-2, r, -0x1.ffffef20a4123p-2);
    y = mad(-0x1.00ea348b88334p-2, r2, y);
    y = mad(y, r2, y0 + r);
    return (float)y;
}

}  // namespace ft_libm
