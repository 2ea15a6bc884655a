// Transcendentals of the KannalaBrandt8 camera model (reference src/CameraModels/KannalaBrandt8.cpp:67-84,306-372:
// atan2f, cos / sin / tan of float arguments, all evaluated by the reference with the HOST libm).
//   cos / sin of psi in [-pi, pi] and atan2f: glibc's routines reproduced bit for bit (libm_f32.h; cosf / sinf / atanf checked
//   on every argument, atan2f on 2^31 pairs) - KannalaBrandt8::project, hence isInFrustum and both projection searches on
//   two-camera frames, equal the host's to the bit.
//   tanf (unproject, used by the triangulation filter only): glibc 2.35 ships the fdlibm float routine (float operations with
//   their own rounding at every step, a Cody-Waite reduction by pi/2); here it is evaluated in DOUBLE and narrowed once = the
//   correctly rounded float except for double-rounding ties, whereas glibc's result is within 1 ulp of that, NOT equal to it.
//   It feeds TriangulateMatches' SVD, which is compared within a tolerance anyway (Eigen's JacobiSVD on the host).
#pragma once
#include <hip/hip_runtime.h>

#include "libm_f32.h"

__device__ __forceinline__ float ft_atan2_f(float y, float x) { return ft_libm::atan2f_glibc(y, x); }
__device__ __forceinline__ float ft_cos_f(float a) { return ft_libm::cosf_glibc(a); }
__device__ __forceinline__ float ft_sin_f(float a) { return ft_libm::sinf_glibc(a); }
__device__ __forceinline__ float ft_tan_f(float a) { return (float)tan((double)a); }
