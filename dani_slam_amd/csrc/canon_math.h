// canon_math.h -- device-side pieces of the canonical f32 evaluation order.
// The oracle carries its own restatement of the same specification (oracle/sp_oracle.cpp); the two are
// deliberately not shared code.
#pragma once
#include <hip/hip_runtime.h>

namespace dani {

// exp(x) for x <= 0 built only from correctly rounded f32 operations (fma, rint, ldexp), so that the
// result is bit-identical on gfx950 and on the host.  Spec:
//   x  = max(x, -87)                       (result stays a normal float)
//   n  = rint(x * log2(e))                 (round half to even)
//   r  = fma(n, -LN2_HI, x); r = fma(n, -LN2_LO, r)
//   p  = Horner(1/5040, 1/720, 1/120, 1/24, 1/6, 1/2, 1, 1) in r, every step one fma
//   exp = ldexp(p, n)
__device__ __forceinline__ float dani_expf(float x) {
  x = fmaxf(x, -87.0f);
  const float n = __builtin_rintf(x * 1.44269504088896341f);
  float r = __builtin_fmaf(n, -0.693145751953125f, x);
  r = __builtin_fmaf(n, -1.42860682030941723212e-6f, r);
  float p = 1.0f / 5040.0f;
  p = __builtin_fmaf(p, r, 1.0f / 720.0f);
  p = __builtin_fmaf(p, r, 1.0f / 120.0f);
  p = __builtin_fmaf(p, r, 1.0f / 24.0f);
  p = __builtin_fmaf(p, r, 1.0f / 6.0f);
  p = __builtin_fmaf(p, r, 0.5f);
  p = __builtin_fmaf(p, r, 1.0f);
  p = __builtin_fmaf(p, r, 1.0f);
  return ldexpf(p, (int)n);
}

}  // namespace dani
