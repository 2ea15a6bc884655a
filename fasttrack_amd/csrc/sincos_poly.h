// cos and sin of a double in [0, 2 pi] for computeOrbDescriptor's rotation (ORBextractor.cc:121-123: the angle in
// degrees times pi/180 as float, libm's cos/sin on it, both narrowed to float).  Shared by k_orient_desc and by the
// host test that compares the narrowed results with libm's over every float of the interval (tests/cpp/test_sincos.cpp).
//
// Quadrant by Cody-Waite reduction with a two-part pi/2 (the 33-bit head makes n * head exact for n <= 4), then the
// classic odd / even minimax polynomials on [-pi/4, pi/4] (fdlibm's kernel coefficients), each within one ulp of a
// double.  The generic library routine spends twice the instructions on argument ranges and special values that
// cannot occur here.
#pragma once
#if defined(__HIPCC__)
#define FT_HD __host__ __device__ __forceinline__
#else
#define FT_HD inline
#endif

FT_HD void ft_sincos_0_2pi(double x, double &sn, double &cs) {
    const double n = __builtin_rint(x * 6.36619772367581382433e-01);  // x * 2/pi: n in 0 .. 4
    double r = __builtin_fma(n, -1.57079632673412561417e+00, x);     // pi/2 head
    r = __builtin_fma(n, -6.07710050650619224932e-11, r);            // pi/2 tail
    const double z = r * r;
    const double ps = __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z,
                          1.58969099521155010221e-10, -2.50507602534068634195e-08), 2.75573137070700676789e-06),
                          -1.98412698298579493134e-04), 8.33333333332248946124e-03), -1.66666666666666324348e-01);
    const double pc = __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z,
                          -1.13596475577881948265e-11, 2.08757232129817482790e-09), -2.75573143513906633035e-07),
                          2.48015872894767294178e-05), -1.38888888888741095749e-03), 4.16666666666666019037e-02);
    const double s0 = __builtin_fma(r * z, ps, r);                            // sin r
    const double c0 = __builtin_fma(z * z, pc, __builtin_fma(z, -0.5, 1.0));  // cos r
    const int q = (int)n;
    const double sq = (q & 1) ? c0 : s0, cq = (q & 1) ? s0 : c0;
    sn = (q & 2) ? -sq : sq;
    cs = ((q + 1) & 2) ? -cq : cq;
}
