// Transcendentals of the KannalaBrandt8 camera model (reference src/CameraModels/KannalaBrandt8.cpp:67-84,114-143,306-372:
// atan2f, cos / sin / tan of float arguments, all evaluated by the reference with the HOST libm).  All four are glibc's
// routines reproduced bit for bit (libm_f32.h: cosf / sinf / atanf / tanf checked against the host on every argument of their
// domain, atan2f on 2^31 pairs, and on the device through ft_selftest_libm) - KannalaBrandt8::project and ::unproject, hence
// isInFrustum, both projection searches and the triangulation filter of two-camera frames, see the host's floats.
#pragma once
#include <hip/hip_runtime.h>

#include "libm_f32.h"

__device__ __forceinline__ float ft_atan2_f(float y, float x) { return ft_libm::atan2f_glibc(y, x); }
__device__ __forceinline__ float ft_cos_f(float a) { return ft_libm::cosf_glibc(a); }
__device__ __forceinline__ float ft_sin_f(float a) { return ft_libm::sinf_glibc(a); }
__device__ __forceinline__ float ft_tan_f(float a) {
    bool exact;
    const float t = ft_libm::tanf_glibc(a, &exact);
    return exact ? t : (float)tan((double)a);  // |a| >= 120: cannot come out of unproject (theta within a few steps of [0, pi/2])
}
