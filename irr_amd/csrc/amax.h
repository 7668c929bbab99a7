// Folding max |v| of what a launch stores into an "amax slot" (include/irr_hip.h, section "h2"): shared by the conv epilogues
// (conv_x3.hip), the magnitude pass (irr_amax_f32) and the producers that fold it on the way (misc.hip, conv_wgrad.hip, conv_small.hip).
#pragma once
#include "common.h"

namespace {

// max |v| over a wave -> at most one atomic max on the slot (non-negative floats order like their bit patterns; a NaN pattern
// wins).  Thousands of waves fold into ONE address and same-address atomics serialise at the memory side (~10 ns each: 6k of them
// were the whole run time of the amax pass over a small tensor), so a wave first LOOKS at the slot with a coherent (agent-scope)
// load: the slot only grows, and after the first few waves almost nobody has anything to add (a stale look only costs an atomic
// that changes nothing).  The atomic is the returning form and the wave consumes the result before it ends: a returned value
// means the update has been performed, whatever the hardware does with posted atomics at the end of a kernel.
// All comparisons run on the BIT PATTERNS (round 5): for non-negative floats the unsigned order of the patterns is the numeric
// order, +inf sorts above every finite value and every NaN pattern above +inf -- "a NaN wins" without a second compare.  One
// v_and_b32 + one v_max_u32 per folded value instead of |.|, two compares, an or and a select: the fold runs once per STORED value
// in the epilogues, and the epilogue waves of conv_x3s_kernel are bound by their VALU instructions.
__device__ __forceinline__ uint32_t x3_amax_bits(float v) { return __builtin_bit_cast(uint32_t, v) & 0x7fffffffu; }
__device__ __forceinline__ float x3_amax_fold(float m, float v) {      // m: a running maximum (non-negative, or a NaN pattern)
  const uint32_t a = x3_amax_bits(v), b = __builtin_bit_cast(uint32_t, m);
  return __builtin_bit_cast(float, a > b ? a : b);
}
__device__ __forceinline__ float x3_amax_wave(float m) {
  uint32_t b = __builtin_bit_cast(uint32_t, m);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const uint32_t o = (uint32_t)__shfl_xor((int)b, off, 64);
    b = o > b ? o : b;
  }
  return __builtin_bit_cast(float, b);
}
// max over the 32 lanes of each HALF-wave of a bit pattern, valid in lanes 16..31 and 48..63 afterwards: five DPP steps (lane swaps
// inside quads, the two mirrors of a 16-lane row, then lane 15 of rows 0 / 2 broadcast into rows 1 / 3) -- VALU only, no LDS crossbar
// round trips (sixteen of these per wave end the epilogue of a launch that folds channel maxima; as __shfl_xor ladders they were 80
// ds_bpermute with their waits)
template <int CTRL, int ROWS>
__device__ __forceinline__ uint32_t x3_dpp_max(uint32_t v) {
  const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROWS, 0xF, false);
  return o > v ? o : v;
}
__device__ __forceinline__ uint32_t x3_amax_half_upper(uint32_t v) {
  v = x3_dpp_max<0xB1, 0xF>(v);                             // quad_perm [1, 0, 3, 2]
  v = x3_dpp_max<0x4E, 0xF>(v);                             // quad_perm [2, 3, 0, 1]
  v = x3_dpp_max<0x141, 0xF>(v);                            // row_half_mirror
  v = x3_dpp_max<0x140, 0xF>(v);                            // row_mirror
  v = x3_dpp_max<0x142, 0xA>(v);                            // row_bcast:15 into rows 1 and 3
  return v;
}
#ifndef X3_AMAX_PRECHECK
#define X3_AMAX_PRECHECK 1     // 0 (A/B): every wave / block issues its atomic
#endif
__device__ __forceinline__ void x3_amax_commit(float m, float* slot) {       // one lane
  const uint32_t mb = __builtin_bit_cast(uint32_t, m);
  const uint32_t cur = X3_AMAX_PRECHECK ? __hip_atomic_load((unsigned int*)slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
  if (mb > cur || !X3_AMAX_PRECHECK) {
    const unsigned int old = atomicMax((unsigned int*)slot, mb);
    asm volatile("" ::"v"(old));                            // wait for the atomic's return: performed before the wave ends
  }
}
__device__ __forceinline__ void x3_amax_publish(float m, float* slot) {
  m = x3_amax_wave(m);
  if ((threadIdx.x & 63) == 0) x3_amax_commit(m, slot);
}
// the same for a whole 256-thread block (every thread must arrive): one look at the slot per block
__device__ __forceinline__ void x3_amax_publish_block256(float m, float* slot) {
  __shared__ float wm[4];
  m = x3_amax_wave(m);
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    float r = wm[0];
#pragma unroll
    for (int i = 1; i < 4; ++i) r = x3_amax_fold(r, wm[i]);
    x3_amax_commit(r, slot);
  }
}

}  // namespace
