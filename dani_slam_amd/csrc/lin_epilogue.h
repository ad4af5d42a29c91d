// lin_epilogue.h -- epilogue of the LightGlue linear kernels (linear_fm_kernel, linear_x3_kernel): rotary embedding,
// scale, residual add, store of one 32x32 accumulator tile.
//
// The arithmetic per element is  v = acc [+ bias];  rotary pair (v0, v1) -> (v0 c - v1 s, v1 c + v0 s);  v *= alpha;
// v += res;  in exactly this order (the order the oracle's linear follows).  What this header is about is the ADDRESS
// arithmetic: a tile's 16 rows are wave-uniform, so each row's base pointer is formed on the scalar unit (64-bit) and
// the lane contributes one 32-bit offset (4 (lane/32) ld + token) shared by all 16 rows -- a store is then
// `global_store saddr + voffset` with no vector address math.  Written per element as y[(size_t)row * ld + tok] the
// same epilogue cost ~25 VALU instructions per output (64-bit multiply-adds, one exec-mask branch per store):
// 400 per wave, 12 % of the f32 kernel's issue cycles and half of the bf16x3 kernel's (PMC, profiles/README.md).
//
// Contract (checked by the host launchers): OUT is a multiple of 64 (every tile row is a real output row), rot_rows
// is a multiple of 64 (rotary or not is uniform per tile), OUT * ld < 2^30 (lane offsets fit 32 bits).
#pragma once
#include <hip/hip_runtime.h>

namespace dani {

struct LinEpi {
  float* y; const float* res; const float* cosb; const float* sinb; const float* bias;
  int ld, rot_rows;
  float alpha;
};

// acc: the 16 accumulators of this lane (column = token `tok`, register r = row (r&3) + 8 (r>>2) + 4 h of the tile).
// row0: first output row of the tile, WAVE-UNIFORM (callers derive it from blockIdx and a readfirstlane'd wave id).
template <bool ADD_BIAS, typename Acc>
__device__ __forceinline__ void lin_epilogue(const Acc& acc, int row0, int h, int tok, bool ok, const LinEpi& e) {
  if (!ok) return;
  // byte offsets, 32 bits: base (SGPR pair) + zero-extended VGPR is the `saddr` form of global loads/stores
  const unsigned ldb = 4u * (unsigned)e.ld;
  const unsigned lane_off = 4u * (unsigned)h * ldb + 4u * (unsigned)tok;  // rows differ by 4 h
  const unsigned rot_off = 2u * (unsigned)h * ldb + 4u * (unsigned)tok;   // rotary pair index differs by 2 h
  auto at = [](const float* base, unsigned off) -> const float& { return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + off); };
  auto at_w = [](float* base, unsigned off) -> float& { return *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + off); };
  const bool rot = row0 < e.rot_rows;
  // Loads first, all of them, each kind behind ONE wave-uniform branch: written as "if (rot) load; ... if (res) load" inside
  // the row loop, hipcc branches around every pair of loads and waits vmcnt(0) after it -- 16 dependent L2 round trips per
  // wave at the very end of a workgroup, where nothing else on the CU can cover them.
  float cs[8], sn[8], rv[16];
  if (rot) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int rr = row0 + ((2 * i) & 3) + 8 * ((2 * i) >> 2);
      const size_t ip = (size_t)((rr & 63) >> 1) * e.ld;                 // uniform
      cs[i] = at(e.cosb + ip, rot_off); sn[i] = at(e.sinb + ip, rot_off);
    }
  }
  if (e.res) {
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const int rr = row0 + (r & 3) + 8 * (r >> 2);
      const float* rp = e.res + (size_t)rr * e.ld;
      rv[r] = at(rp, lane_off); rv[r + 1] = at(rp, lane_off + ldb);
    }
  }
#pragma unroll
  for (int r = 0; r < 16; r += 2) {
    const int rr = row0 + (r & 3) + 8 * (r >> 2);                        // uniform; this lane's rows are rr + 4h, +1
    float v0 = acc[r], v1 = acc[r + 1];
    if (ADD_BIAS) {
      const float* b = e.bias + rr;
      v0 += b[4 * h]; v1 += b[4 * h + 1];
    }
    if (rot) {
      const float c = cs[r >> 1], s = sn[r >> 1];
      const float r0 = v0 * c - v1 * s, r1 = v1 * c + v0 * s;            // t*cos + rotate_half(t)*sin
      v0 = r0; v1 = r1;
    }
    v0 *= e.alpha; v1 *= e.alpha;
    if (e.res) { v0 += rv[r]; v1 += rv[r + 1]; }
    float* yp = e.y + (size_t)rr * e.ld;                                 // uniform
    at_w(yp, lane_off) = v0; at_w(yp, lane_off + ldb) = v1;
  }
}

}  // namespace dani
