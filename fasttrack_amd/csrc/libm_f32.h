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
#include <math.h>
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

// sinf (cosine = false) / cosf (cosine = true) of s_sinf.c / s_cosf.c for |y| < 120.  Larger arguments (glibc's
// reduce_large path) cannot occur on the path - the callers pass an angle of [0, 2 pi] or an atan2f result - and are not
// restated: they, NaN and infinities take the double-precision routine narrowed once (NaN for NaN / Inf, as glibc returns;
// the correctly rounded value except at double-rounding ties for finite arguments), never the polynomial outside its range.
FT_LM_HD float sincosf_one(float y, bool cosine) {
    double x = (double)y;
    const uint32_t top = abstop12(y);
    if (top >= abstop12(120.0f)) return cosine ? (float)cos(x) : (float)sin(x);
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


// ---- atanf / atan2f --------------------------------------------------------------------------------------------------
// glibc 2.35 still ships the fdlibm single-precision routines for these (sysdeps/ieee754/flt-32/s_atanf.c, e_atan2f.c; the
// x86-64 multiarch directory has no FMA variant of them): sequences of float operations, each rounded on its own - so the
// restatement below must not be contracted (host: -ffp-contract=off; device: the __f*_rn intrinsics) and its division must be
// correctly rounded (-fhip-fp32-correctly-rounded-divide-sqrt).  Used by KannalaBrandt8::project
// (/root/reference/src/CameraModels/KannalaBrandt8.cpp:67-84: theta = atan2f(sqrtf(x^2 + y^2), z), psi = atan2f(y, x)).
// Checked against the host: tests/cpp/test_libm_f32.cpp (atanf on EVERY float, atan2f on the signs / special cases and 2^31
// random pairs), and on the device through ft_selftest_libm.
#if defined(__HIP_DEVICE_COMPILE__)
#define FT_LM_MUL(a, b) __fmul_rn(a, b)
#define FT_LM_ADD(a, b) __fadd_rn(a, b)
#define FT_LM_SUB(a, b) __fsub_rn(a, b)
#define FT_LM_DIV(a, b) __fdiv_rn(a, b)
#else
#define FT_LM_MUL(a, b) ((a) * (b))
#define FT_LM_ADD(a, b) ((a) + (b))
#define FT_LM_SUB(a, b) ((a) - (b))
#define FT_LM_DIV(a, b) ((a) / (b))
#endif

FT_LM_HD float atanf_glibc(float x) {
    const float atanhi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
    const float atanlo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
    const float aT[11] = {3.3333334327e-01f,  -2.0000000298e-01f, 1.4285714924e-01f,  -1.1111110449e-01f, 9.0908870101e-02f, -7.6918758452e-02f,
                          6.6610731184e-02f,  -5.8335702866e-02f, 4.9768779427e-02f,  -3.6531571299e-02f, 1.6285819933e-02f};
    const int32_t hx = (int32_t)f32_bits(x), ix = hx & 0x7fffffff;
    int id;
    if (ix >= 0x4c000000) {  // |x| >= 2^25
        if (ix > 0x7f800000) return FT_LM_ADD(x, x);  // NaN
        const float r = FT_LM_ADD(atanhi[3], atanlo[3]);
        return hx > 0 ? r : -r;
    }
    if (ix < 0x3ee00000) {  // |x| < 0.4375
        if (ix < 0x31000000) return x;  // |x| < 2^-29
        id = -1;
    } else {
        x = __builtin_fabsf(x);
        if (ix < 0x3f980000) {      // |x| < 1.1875
            if (ix < 0x3f300000) {  // 7/16 <= |x| < 11/16
                id = 0;
                x = FT_LM_DIV(FT_LM_SUB(FT_LM_MUL(2.0f, x), 1.0f), FT_LM_ADD(2.0f, x));
            } else {  // 11/16 <= |x| < 19/16
                id = 1;
                x = FT_LM_DIV(FT_LM_SUB(x, 1.0f), FT_LM_ADD(x, 1.0f));
            }
        } else {
            if (ix < 0x401c0000) {  // |x| < 2.4375
                id = 2;
                x = FT_LM_DIV(FT_LM_SUB(x, 1.5f), FT_LM_ADD(1.0f, FT_LM_MUL(1.5f, x)));
            } else {  // 2.4375 <= |x| < 2^25
                id = 3;
                x = FT_LM_DIV(-1.0f, x);
            }
        }
    }
    const float z = FT_LM_MUL(x, x), w = FT_LM_MUL(z, z);
    // the sum of aT[i] z^(i+1) split into its odd and even terms
    float s1 = FT_LM_ADD(aT[8], FT_LM_MUL(w, aT[10]));
    s1 = FT_LM_ADD(aT[6], FT_LM_MUL(w, s1));
    s1 = FT_LM_ADD(aT[4], FT_LM_MUL(w, s1));
    s1 = FT_LM_ADD(aT[2], FT_LM_MUL(w, s1));
    s1 = FT_LM_MUL(z, FT_LM_ADD(aT[0], FT_LM_MUL(w, s1)));
    float s2 = FT_LM_ADD(aT[7], FT_LM_MUL(w, aT[9]));
    s2 = FT_LM_ADD(aT[5], FT_LM_MUL(w, s2));
    s2 = FT_LM_ADD(aT[3], FT_LM_MUL(w, s2));
    s2 = FT_LM_MUL(w, FT_LM_ADD(aT[1], FT_LM_MUL(w, s2)));
    if (id < 0) return FT_LM_SUB(x, FT_LM_MUL(x, FT_LM_ADD(s1, s2)));
    const float zz = FT_LM_SUB(atanhi[id], FT_LM_SUB(FT_LM_SUB(FT_LM_MUL(x, FT_LM_ADD(s1, s2)), atanlo[id]), x));
    return hx < 0 ? -zz : zz;
}

FT_LM_HD float atan2f_glibc(float y, float x) {
    const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    const int32_t hx = (int32_t)f32_bits(x), hy = (int32_t)f32_bits(y), ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return FT_LM_ADD(x, y);  // NaN
    if (hx == 0x3f800000) return atanf_glibc(y);                     // x = 1.0
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);               // 2 * sign(x) + sign(y)
    if (iy == 0) {                                                   // y = 0
        if (m < 2) return y;
        return m == 2 ? FT_LM_ADD(pi, tiny) : FT_LM_SUB(-pi, tiny);
    }
    if (ix == 0) return hy < 0 ? FT_LM_SUB(-pi_o_2, tiny) : FT_LM_ADD(pi_o_2, tiny);  // x = 0
    if (ix == 0x7f800000) {                                                             // x = +-inf
        if (iy == 0x7f800000) {
            switch (m) {
                case 0: return FT_LM_ADD(pi_o_4, tiny);
                case 1: return FT_LM_SUB(-pi_o_4, tiny);
                case 2: return FT_LM_ADD(FT_LM_MUL(3.0f, pi_o_4), tiny);
                default: return FT_LM_SUB(FT_LM_MUL(-3.0f, pi_o_4), tiny);
            }
        }
        switch (m) {
            case 0: return 0.0f;
            case 1: return -0.0f;
            case 2: return FT_LM_ADD(pi, tiny);
            default: return FT_LM_SUB(-pi, tiny);
        }
    }
    if (iy == 0x7f800000) return hy < 0 ? FT_LM_SUB(-pi_o_2, tiny) : FT_LM_ADD(pi_o_2, tiny);  // y = +-inf
    const int k = (iy - ix) >> 23;
    float z;
    if (k > 60) z = FT_LM_ADD(pi_o_2, FT_LM_MUL(0.5f, pi_lo));  // |y / x| > 2^60
    else if (hx < 0 && k < -60) z = 0.0f;                       // |y| / x < -2^60
    else z = atanf_glibc(__builtin_fabsf(FT_LM_DIV(y, x)));
    switch (m) {
        case 0: return z;
        case 1: return bits_f32(f32_bits(z) ^ 0x80000000u);
        case 2: return FT_LM_SUB(pi, FT_LM_SUB(z, pi_lo));
        default: return FT_LM_SUB(FT_LM_SUB(z, pi_lo), pi);
    }
}

// ---- tanf ------------------------------------------------------------------------------------------------------------
// glibc 2.35 sysdeps/ieee754/flt-32/{s_tanf.c,k_tanf.c}: |x| <= pi/4 goes to fdlibm's float kernel directly, larger
// arguments are first reduced in double by s_sincosf.h's reduce_fast (the disassembly of this image's libm.so.6 shows the
// mulsd / subsd pair and the two constants) and handed to the kernel as a float head + tail.
// KannalaBrandt8::unproject evaluates tanf(theta) with theta in [0, pi/2] (/root/reference/src/CameraModels/KannalaBrandt8.cpp:
// 112-141: theta_d clamped to +-pi/2, then Newton steps on theta); |x| >= 120 (reduce_large) is not restated: *exact =
// false tells the caller to take another route.
FT_LM_HD float kernel_tanf(float x, float y, int iy) {
    const uint32_t Tb[13] = {0x3eaaaaabu, 0x3e088889u, 0x3d5d0dd1u, 0x3cb327a4u, 0x3c11371fu, 0x3b6b6916u, 0x3abede48u,
                             0x3a1a26c8u, 0x398137b9u, 0x38a3f445u, 0x3895c07au, 0xb79bae5fu, 0x37d95384u};
    const float pio4 = bits_f32(0x3f490fdau), pio4lo = bits_f32(0x33222168u);
    const int32_t hx = (int32_t)f32_bits(x), ix = hx & 0x7fffffff;
    if (ix < 0x39000000) {  // |x| < 2^-13
        if ((int)x == 0) {
            if ((ix | (iy + 1)) == 0) return FT_LM_DIV(1.0f, __builtin_fabsf(x));
            if (iy == 1) return x;
            return FT_LM_DIV(-1.0f, x);
        }
    }
    const float sgn = (float)(1 - ((hx >> 30) & 2));
    if (ix >= 0x3f2ca140) {  // |x| >= 0.6744
        if (hx < 0) {
            x = -x;
            y = -y;
        }
        const float z0 = FT_LM_SUB(pio4, x), w0 = FT_LM_SUB(pio4lo, y);
        x = FT_LM_ADD(z0, w0);
        y = 0.0f;
        if (__builtin_fabsf(x) < 0x1p-13f) return FT_LM_MUL(FT_LM_MUL(sgn, (float)iy), FT_LM_SUB(1.0f, FT_LM_MUL((float)(2 * iy), x)));
    }
    float z = FT_LM_MUL(x, x), w = FT_LM_MUL(z, z);
    // x^5 (T[1] + x^2 T[2] + ...) split into the terms of even and odd index
    float r = FT_LM_ADD(bits_f32(Tb[9]), FT_LM_MUL(w, bits_f32(Tb[11])));
    r = FT_LM_ADD(bits_f32(Tb[7]), FT_LM_MUL(w, r));
    r = FT_LM_ADD(bits_f32(Tb[5]), FT_LM_MUL(w, r));
    r = FT_LM_ADD(bits_f32(Tb[3]), FT_LM_MUL(w, r));
    r = FT_LM_ADD(bits_f32(Tb[1]), FT_LM_MUL(w, r));
    float v = FT_LM_ADD(bits_f32(Tb[10]), FT_LM_MUL(w, bits_f32(Tb[12])));
    v = FT_LM_ADD(bits_f32(Tb[8]), FT_LM_MUL(w, v));
    v = FT_LM_ADD(bits_f32(Tb[6]), FT_LM_MUL(w, v));
    v = FT_LM_ADD(bits_f32(Tb[4]), FT_LM_MUL(w, v));
    v = FT_LM_MUL(z, FT_LM_ADD(bits_f32(Tb[2]), FT_LM_MUL(w, v)));
    float s = FT_LM_MUL(z, x);
    r = FT_LM_ADD(y, FT_LM_MUL(z, FT_LM_ADD(FT_LM_MUL(s, FT_LM_ADD(r, v)), y)));
    r = FT_LM_ADD(r, FT_LM_MUL(bits_f32(Tb[0]), s));
    w = FT_LM_ADD(x, r);
    if (ix >= 0x3f2ca140) {
        v = (float)iy;
        const float q = FT_LM_DIV(FT_LM_MUL(w, w), FT_LM_ADD(w, v));
        return FT_LM_MUL(sgn, FT_LM_SUB(v, FT_LM_MUL(2.0f, FT_LM_SUB(x, FT_LM_SUB(q, r)))));
    }
    if (iy == 1) return w;
    // -1 / (x + r), accurately
    z = bits_f32(f32_bits(w) & 0xfffff000u);
    v = FT_LM_SUB(r, FT_LM_SUB(z, x));  // z + v = r + x
    const float a = FT_LM_DIV(-1.0f, w);
    const float t = bits_f32(f32_bits(a) & 0xfffff000u);
    s = FT_LM_ADD(1.0f, FT_LM_MUL(t, z));
    return FT_LM_ADD(t, FT_LM_MUL(a, FT_LM_ADD(s, FT_LM_MUL(t, v))));
}

FT_LM_HD float tanf_glibc(float x, bool *exact) {
    const int32_t hx = (int32_t)f32_bits(x), ix = hx & 0x7fffffff;
    *exact = true;
    if (ix <= 0x3f490fda) return kernel_tanf(x, 0.0f, 1);  // |x| <= pi/4
    if (abstop12(x) >= abstop12(120.0f)) {                 // reduce_large (and inf / NaN): not restated
        *exact = false;
        return 0.0f;
    }
    // reduce_fast of s_sincosf.h in double - a plain multiply and subtract here: tanf has no FMA variant - then the reduced
    // argument split into a float head and tail for the fdlibm kernel
#if defined(__HIP_DEVICE_COMPILE__)
    const double r = __dmul_rn((double)x, 0x1.45F306DC9C883p+23);
    const int n = (int)(((int32_t)r + 0x800000) >> 24);
    const double xr = __dsub_rn((double)x, __dmul_rn((double)n, 0x1.921FB54442D18p0));
    const float y0 = (float)xr;
    const float y1 = (float)__dsub_rn(xr, (double)y0);
#else
    const double r = (double)x * 0x1.45F306DC9C883p+23;
    const int n = (int)(((int32_t)r + 0x800000) >> 24);
    const double nh = (double)n * 0x1.921FB54442D18p0;
    const double xr = (double)x - nh;
    const float y0 = (float)xr;
    const float y1 = (float)(xr - (double)y0);
#endif
    return kernel_tanf(y0, y1, 1 - ((n & 1) << 1));
}

}  // namespace ft_libm
