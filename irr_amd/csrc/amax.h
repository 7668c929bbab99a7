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
__device__ __forceinline__ float x3_amax_wave(float m) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float o = __shfl_xor(m, off, 64);
    m = (o > m || o != o) ? o : m;
  }
  return m;
}
#ifndef X3_AMAX_PRECHECK
#define X3_AMAX_PRECHECK 1     // 0 (A/B): every wave / block issues its atomic
#endif
__device__ __forceinline__ void x3_amax_commit(float m, float* slot) {       // one lane
  const float cur = X3_AMAX_PRECHECK ? __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -1.f;
  if (m > cur || m != m) {
    const unsigned int old = atomicMax((unsigned int*)slot, __builtin_bit_cast(unsigned int, m));
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
    for (int i = 1; i < 4; ++i) r = (wm[i] > r || wm[i] != wm[i]) ? wm[i] : r;
    x3_amax_commit(r, slot);
  }
}
__device__ __forceinline__ float x3_amax_fold(float m, float v) {
  const float a = __builtin_fabsf(v);
  return (a > m || a != a) ? a : m;
}

}  // namespace
