// float32 arctangent for the slope / aspect / Sx epilogues (device code only).
// libm's atanf costs ~40 instructions; this is Abramowitz & Stegun 4.4.49 on [0, 1] (|err| <= 2e-8) with the
// reciprocal for arguments above 1.  atan(0) = 0 exactly, NaN propagates, atan(inf) = pi/2.
#pragma once
#include <hip/hip_runtime.h>

namespace topo {

__device__ __forceinline__ float atan_unit(float t) {  // t in [0, 1]
    const float z = t * t;
    float p = 0.0028662257f;
    p = fmaf(p, z, -0.0161657367f);
    p = fmaf(p, z, 0.0429096138f);
    p = fmaf(p, z, -0.0752896400f);
    p = fmaf(p, z, 0.1065626393f);
    p = fmaf(p, z, -0.1420889944f);
    p = fmaf(p, z, 0.1999355085f);
    p = fmaf(p, z, -0.3333314528f);
    return fmaf(p * z, t, t);
}

__device__ __forceinline__ float atan_pos(float s) {  // s >= 0 (or NaN)
    const bool big = s > 1.0f;
    const float r = atan_unit(big ? __builtin_amdgcn_rcpf(s) : s);
    return big ? 1.5707963267948966f - r : r;
}

__device__ __forceinline__ float atan_signed(float s) { return copysignf(atan_pos(fabsf(s)), s); }

}  // namespace topo
