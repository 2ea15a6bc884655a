// Transcendentals of the KannalaBrandt8 camera model (reference src/CameraModels/KannalaBrandt8.cpp:67-84,306-372:
// atan2f, cos / sin / tan of float arguments, all evaluated by the reference with the HOST libm).
//   cos / sin of psi in [-pi, pi]: glibc's cosf / sinf reproduced bit for bit (libm_f32.h, exhaustively checked).
//   atan2f, tanf: glibc 2.35 still ships the fdlibm float routines for these (sequences of float operations with their
//   own rounding at every step); here they are evaluated in DOUBLE and narrowed once = the correctly rounded float
//   except for double-rounding ties, whereas glibc's results are within 1 ulp of that, NOT equal to it.  Device and
//   host therefore differ in the last bit on a few per cent of arguments; what reaches a projection stays below
//   1e-4 px (north_star's tolerance for floats) and the KB8 tests state that tolerance and the share of in-view / match
//   flags that sit on a decision boundary.
#pragma once
#include <hip/hip_runtime.h>

#include "libm_f32.h"

__device__ __forceinline__ float ft_atan2_f(float y, float x) { return (float)atan2((double)y, (double)x); }
__device__ __forceinline__ float ft_cos_f(float a) { return ft_libm::cosf_glibc(a); }
__device__ __forceinline__ float ft_sin_f(float a) { return ft_libm::sinf_glibc(a); }
__device__ __forceinline__ float ft_tan_f(float a) { return (float)tan((double)a); }
