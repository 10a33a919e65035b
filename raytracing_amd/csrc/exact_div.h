// Correctly rounded fp32 quotients in fewer instructions than the compiler's general sequence (gfx950, device only).
//
// The compiler expands every a / b into v_div_scale x2, v_rcp, five FMAs, v_div_fmas, v_div_fixup and a pair of
// s_setreg that switch denormals on and off around them: about 13 instructions, because it must be right for every
// pair (a, b). The two special cases below need less and return the same bits. exact_div_check.hip (built next to the
// library) compares each with the compiler's division over all 2^32 inputs on the GPU; the GPU tests run it.
// Everything here assumes what the kernels are compiled with: denormals flushed, no implicit contraction.
#pragma once

#include <hip/hip_runtime.h>
#include <math.h>

#ifndef RT_SHORT_DIV
#define RT_SHORT_DIV 1
#endif

namespace rtd {

// 1 / x: hardware estimate (1 ulp), one Newton step in FMA, v_div_fixup for zeros, infinities and NaNs.
// Equal to 1.0f / x for every one of the 2^32 inputs.
__device__ __forceinline__ float rcp_exact(float x) {
#if RT_SHORT_DIV
    float y = __builtin_amdgcn_rcpf(x);
    float e = __builtin_fmaf(-x, y, 1.0f);
    y = __builtin_fmaf(e, y, y);
    return __builtin_amdgcn_div_fixupf(y, x, 1.0f);
#else
    return 1.0f / x;
#endif
}

// a / c where c is the same in every lane and rc = 1.0f / c was computed once: product, then two residual
// corrections in FMA (the second one rounds correctly by Markstein's theorem once the first has made the
// quotient faithful; c's significand must not be all ones, true of the integers below 2^24 used here).
// Equal to a / c for a = 0 and for 2^-100 <= |a| < inf: outside that range the residual underflows. The
// numerators in the tracer are sums of integers and k/255 fractions, hence zero or at least 1/255 in magnitude.
__device__ __forceinline__ float div_by_uniform(float a, float c, float rc) {
#if RT_SHORT_DIV
    float q = a * rc;
    float r = __builtin_fmaf(-c, q, a);
    q = __builtin_fmaf(r, rc, q);
    r = __builtin_fmaf(-c, q, a);
    return __builtin_fmaf(r, rc, q);
#else
    return a / c;
#endif
}

// sqrt(x): hardware reciprocal square root (1 ulp), product, one residual correction in FMA: 5 instructions in a
// chain instead of the compiler's 17. Correctly rounded for every x in [2^-100, inf); everything else (zero, tiny and
// negative values, infinities, NaNs) takes the ordinary sqrtf in a branch that is skipped when no lane needs it.
// Equal to sqrtf(x) for every one of the 2^32 inputs.
__device__ __forceinline__ float sqrt_exact(float x) {
#if RT_SHORT_DIV
    float r = __builtin_amdgcn_rsqf(x);
    float s = x * r;
    float h = 0.5f * r;
    float e = __builtin_fmaf(-s, s, x);
    float y = __builtin_fmaf(e, h, s);
    if (__builtin_expect(!(x >= 0x1p-100f && x < __builtin_inff()), 0)) y = sqrtf(x);
    return y;
#else
    return sqrtf(x);
#endif
}

}  // namespace rtd
