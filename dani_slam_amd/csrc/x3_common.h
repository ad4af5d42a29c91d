// x3_common.h -- device helpers of the bf16x3 kernels (lg_x3.hip: linear layers; lg_kernels.hip: attn_x3_kernel).
// An f32 value is split into three bf16 terms  x = hi + mid + lo, each the round-to-nearest-even bf16 of what the
// previous ones left (the subtractions are exact in f32), so the three carry the 24 significand bits of x.
#pragma once
#include <hip/hip_runtime.h>

namespace dani {

typedef float x3_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// two f32 -> packed (hi | mid | lo) bf16 pairs (element 0 in the low half)
__device__ __forceinline__ void x3_split2(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
  x3_f32x2 v = {a, b};
  bf16x2 h = __builtin_convertvector(v, bf16x2);
  x3_f32x2 r1 = v - __builtin_convertvector(h, x3_f32x2);      // exact
  bf16x2 m = __builtin_convertvector(r1, bf16x2);
  x3_f32x2 r2 = r1 - __builtin_convertvector(m, x3_f32x2);     // exact
  bf16x2 l = __builtin_convertvector(r2, bf16x2);
  hi = __builtin_bit_cast(unsigned, h); mid = __builtin_bit_cast(unsigned, m); lo = __builtin_bit_cast(unsigned, l);
}

// acc += (ah + am + al) x (bh + bm + bl) without the three terms below 2^-16, smallest first
template <typename Acc>
__device__ __forceinline__ void x3_mfma6(Acc& acc, const bf16x8& ah, const bf16x8& am, const bf16x8& al, const bf16x8& bh,
                                         const bf16x8& bm, const bf16x8& bl) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
}

}  // namespace dani
