// Transcendentals of the KannalaBrandt8 camera model (reference src/CameraModels/KannalaBrandt8.cpp:67-84,306-372:
// atan2f, cos / sin / tan of float arguments).  The reference's results come from the host libm, whose float functions
// are correctly rounded for practically every argument; OCML's float versions are 1-2 ulp routines, which showed as up
// to 1e-3 px in projections.  Here every call is evaluated in DOUBLE and narrowed once: the result is the correctly
// rounded float except for double-rounding ties (~1e-9 of arguments), so device and host agree to the last bit almost
// everywhere and always within 1e-4 px (north_star tolerance for floats).
#pragma once
#include <hip/hip_runtime.h>

__device__ __forceinline__ float ft_atan2_f(float y, float x) { return (float)atan2((double)y, (double)x); }
__device__ __forceinline__ float ft_cos_f(float a) { return (float)cos((double)a); }
__device__ __forceinline__ float ft_sin_f(float a) { return (float)sin((double)a); }
__device__ __forceinline__ float ft_tan_f(float a) { return (float)tan((double)a); }
