// Direct (VALU) kernels for the tiny-Cout heads of IRR-PWC: conv_last 563->2 / 562->1, context tails 32->2 / 32->1,
// OccUpsampleNetwork.out_convs 32->1 (models/pwc_modules.py:161,198,221,239; models/irr_modules.py:44).
// A 32-wide MFMA tile would spend 94-97 % of its rows on padding for these layers (SURVEY.md Appendix A,
// observation (v)); here every lane owns one output pixel (coalesced along x), weights are wave-uniform scalar
// loads, and the 3x3 neighbourhood reads hit L1.  Same epilogue contract as irr_conv2d_fwd_f32.
#include "common.h"
#include "amax.h"
#include <stdlib.h>

namespace {

template <int NC, int KS>
__global__ __launch_bounds__(256) void conv_smallco_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ bias, const float* __restrict__ res,
                                                              float* __restrict__ y, int B, int Cin, int H, int W,
                                                              int dil, long x_bs, long y_bs, long res_bs, int lrelu,
                                                              float alpha, int accumulate) {
  constexpr int KK = KS * KS;
  const long hw = (long)H * W;
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= hw) return;
  const int b = blockIdx.y;
  const int oy = (int)(p / W), ox = (int)(p - (long)oy * W);
  const int pad = ((KS - 1) * dil) / 2;
  int off[KK];
  bool ok[KK];
#pragma unroll
  for (int t = 0; t < KK; ++t) {
    const int iy = oy - pad + (t / KS) * dil, ix = ox - pad + (t % KS) * dil;
    ok[t] = iy >= 0 && iy < H && ix >= 0 && ix < W;
    off[t] = ok[t] ? iy * W + ix : 0;
  }
  float acc[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) acc[c] = 0.f;
  const float* xb = x + (long)b * x_bs;
  for (int ci = 0; ci < Cin; ++ci) {
    const float* xc = xb + (long)ci * hw;
    float v[KK];
#pragma unroll
    for (int t = 0; t < KK; ++t) {
      const float l = xc[off[t]];
      v[t] = ok[t] ? l : 0.f;
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const float* wc = w + ((long)c * Cin + ci) * KK;      // wave-uniform -> scalar loads
#pragma unroll
      for (int t = 0; t < KK; ++t) acc[c] = fmaf(wc[t], v[t], acc[c]);
    }
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    float v = acc[c] + (bias ? bias[c] : 0.f);
    if (lrelu) v = irr_lrelu(v);
    float* dst = y + (long)b * y_bs + (long)c * hw + p;
    if (res) v = res[(long)b * res_bs + (long)c * hw + p] + alpha * v;
    else v *= alpha;
    if (accumulate) v += *dst;
    *dst = v;
  }
}

// The same sums for planes of a few hundred pixels with many input channels (conv_last 565 -> 2 at 6x7 and 12x14, where the
// quad kernel's W % 4 == 0 does not hold): the kernel above launches ONE block per sample there and walks the 565 channels
// serially (235 us at 6x7x64).  Here a block is 64 pixels x KSL channel slices (wave k: channels k, k + KSL, ...) whose partial
// sums meet in LDS, like conv_smallco_fwd4_kernel.
template <int NC, int KS, int KSL>
__global__ __launch_bounds__(64 * KSL) void conv_smallco_fwd_sl_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                      const float* __restrict__ bias, const float* __restrict__ res,
                                                                      float* __restrict__ y, int Cin, int H, int W, int dil,
                                                                      long x_bs, long y_bs, long res_bs, int lrelu, float alpha,
                                                                      int accumulate) {
  constexpr int KK = KS * KS;
  __shared__ float red[(KSL - 1) * NC * 64];
  const int lane = threadIdx.x & 63;
  const int ks = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hw = H * W;
  const int p = blockIdx.x * 64 + lane;
  const bool pok = p < hw;
  const int b = blockIdx.y;
  const int oy = (pok ? p : 0) / W, ox = (pok ? p : 0) - oy * W;
  const int pad = ((KS - 1) * dil) / 2;
  int off[KK];
  bool ok[KK];
#pragma unroll
  for (int t = 0; t < KK; ++t) {
    const int iy = oy - pad + (t / KS) * dil, ix = ox - pad + (t % KS) * dil;
    ok[t] = pok && iy >= 0 && iy < H && ix >= 0 && ix < W;
    off[t] = ok[t] ? iy * W + ix : 0;
  }
  float acc[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) acc[c] = 0.f;
  const float* xb = x + (long)b * x_bs;
#pragma unroll 2
  for (int ci = ks; ci < Cin; ci += KSL) {
    const float* xc = xb + (long)ci * hw;
    float v[KK];
#pragma unroll
    for (int t = 0; t < KK; ++t) {
      const float l = xc[off[t]];
      v[t] = ok[t] ? l : 0.f;
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const float* wc = w + ((long)c * Cin + ci) * KK;      // wave-uniform -> scalar loads
#pragma unroll
      for (int t = 0; t < KK; ++t) acc[c] = fmaf(wc[t], v[t], acc[c]);
    }
  }
  if (ks > 0) {
#pragma unroll
    for (int c = 0; c < NC; ++c) red[((ks - 1) * NC + c) * 64 + lane] = acc[c];
  }
  __syncthreads();
  if (ks > 0 || !pok) return;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    float v = acc[c];
#pragma unroll
    for (int k = 0; k < KSL - 1; ++k) v += red[(k * NC + c) * 64 + lane];
    v += bias ? bias[c] : 0.f;
    if (lrelu) v = irr_lrelu(v);
    float* dst = y + (long)b * y_bs + (long)c * hw + p;
    if (res) v = res[(long)b * res_bs + (long)c * hw + p] + alpha * v;
    else v *= alpha;
    if (accumulate) v += *dst;
    *dst = v;
  }
}

// ws[co][tap][ci] += sum over this block's pixels of gy[co][p] * x[ci][p + off(tap)]
template <int NC, int KS>
__global__ __launch_bounds__(256) void conv_smallco_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                                float* __restrict__ ws, float* __restrict__ gbias, float alpha,
                                                                int B, int Cin, int H, int W, int dil, long x_bs,
                                                                long gy_bs, int pix_per_block) {
  constexpr int KK = KS * KS;
  const long hw = (long)H * W;
  const int ci = blockIdx.y, b = blockIdx.z;
  const long p0 = (long)blockIdx.x * pix_per_block;
  const long p1 = min(hw, p0 + pix_per_block);
  const int pad = ((KS - 1) * dil) / 2;
  const float* xc = x + (long)b * x_bs + (long)ci * hw;
  const float* gb = gy + (long)b * gy_bs;
  float acc[NC][KK];
  float bsum[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    bsum[c] = 0.f;
#pragma unroll
    for (int t = 0; t < KK; ++t) acc[c][t] = 0.f;
  }
  for (long p = p0 + threadIdx.x; p < p1; p += 256) {
    const int oy = (int)(p / W), ox = (int)(p - (long)oy * W);
    float g[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      g[c] = gb[(long)c * hw + p];
      bsum[c] += g[c];
    }
#pragma unroll
    for (int t = 0; t < KK; ++t) {
      const int iy = oy - pad + (t / KS) * dil, ix = ox - pad + (t % KS) * dil;
      const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
      const float l = xc[ok ? (long)iy * W + ix : 0];
      const float v = ok ? l : 0.f;
#pragma unroll
      for (int c = 0; c < NC; ++c) acc[c][t] = fmaf(g[c], v, acc[c][t]);
    }
  }
  __shared__ float red[4][NC * KK];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (gbias && ci == 0) {                      // the ci == 0 blocks also carry the bias gradient
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      float sb = bsum[c];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) sb += __shfl_down(sb, o, 64);
      if (lane == 0) unsafeAtomicAdd(gbias + c, alpha * sb);
    }
  }
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int t = 0; t < KK; ++t) {
      float s = acc[c][t];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
      if (lane == 0) red[wv][c * KK + t] = s;
    }
  __syncthreads();
  if (threadIdx.x < NC * KK) {
    const int c = threadIdx.x / KK, t = threadIdx.x - c * KK;
    const float s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    unsafeAtomicAdd(ws + ((long)c * KK + t) * Cin + ci, alpha * s);
  }
}

// ---- 4-pixel variants (3x3, dilation 1, W % 4 == 0): a lane owns four consecutive pixels of a row and slides the 3x3
// window over a 3 x 6 register patch (one 16-B load + two edge dwords per row), i.e. 2.25 load instructions per pixel and
// channel instead of 9: these layers stream over 562/563-channel buffers and were instruction-bound, not HBM-bound.
typedef float f32x4s __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void load_patch6(const float* __restrict__ row, bool rok, int x0, int W, float* v) {
  // v[0..5] = row[x0-1 .. x0+4], zeros outside the row / for an invalid row
  if (rok) {
    const f32x4s m = *(const f32x4s*)(row + x0);
    v[1] = m[0]; v[2] = m[1]; v[3] = m[2]; v[4] = m[3];
    v[0] = x0 > 0 ? row[x0 - 1] : 0.f;
    v[5] = x0 + 4 < W ? row[x0 + 4] : 0.f;
  } else {
#pragma unroll
    for (int i = 0; i < 6; ++i) v[i] = 0.f;
  }
}

// Branch-free variant on a buffer resource for lanes that hold CONSECUTIVE quads of one sample: the 16-B load of a patch
// row carries voffset = SOOB (beyond num_records: returns 0, touches nothing) where the row is outside the image; the two
// side pixels come from the neighbouring lanes' quads (ds_bpermute), only lanes 0 and 63 of a wave load theirs (a strided
// dword load costs the L1 as many tag lookups as the 16-B load: three loads per row ran at 1.9 TB/s).
constexpr uint32_t SOOB = 0x80000000u;
__device__ __forceinline__ void patch6_voffsets(int iy, int x0, int H, int W, bool ok, int lane, uint32_t& vm, uint32_t& vl,
                                                uint32_t& vr) {
  const bool rok = ok && iy >= 0 && iy < H;
  const uint32_t base = (uint32_t)((iy * W + x0) * 4);
  vm = rok ? base : SOOB;
  vl = (rok && x0 > 0 && lane == 0) ? base - 4u : SOOB;
  vr = (rok && x0 + 4 < W && lane == 63) ? base + 16u : SOOB;
}
__device__ __forceinline__ void load_patch6_buf(__amdgpu_buffer_rsrc_t rs, uint32_t soff, uint32_t vm, uint32_t vl, uint32_t vr,
                                                float* v) {
  // (cast straight to a float vector: element-wise bit casts of a uint32 vector make this compiler shrink the load to a dword)
  const f32x4s m = __builtin_bit_cast(f32x4s, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)vm, (int)soff, 0));
  v[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)vl, (int)soff, 0));
  v[5] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)vr, (int)soff, 0));
  v[1] = m[0];
  v[2] = m[1];
  v[3] = m[2];
  v[4] = m[3];
}
// after the loads have landed: v[0] / v[5] of the inner lanes from the neighbours (zero at the row ends)
__device__ __forceinline__ void patch6_sides(float* v, int x0, int W, int lane) {
  const float up = __shfl_up(v[4], 1, 64), dn = __shfl_down(v[1], 1, 64);
  if (lane > 0) v[0] = x0 > 0 ? up : 0.f;
  if (lane < 63) v[5] = x0 + 4 < W ? dn : 0.f;
}

// KS waves per block share 64 pixel quads and split the input channels (wave k: channels k, k + KS, ...): a launch has
// only H*W/4 quads per sample, too few waves to hide the latency of a 500-channel loop without the split; the partial sums
// meet in LDS and wave 0 runs the epilogue.  A thread owns its quad in R vertically adjacent output rows: it loads R + 2
// input rows per channel for R output rows, so the rows re-read through the L2 drop from 3x to (R + 2) / R of the input
// (with one row per thread the kernel sat at the L2 bandwidth: 3 x 1.55 GB in 0.5 ms).
template <int NC, int KS, int R>
__global__ __launch_bounds__(64 * KS) void conv_smallco_fwd4_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                    const float* __restrict__ bias, const float* __restrict__ res,
                                                                    float* __restrict__ y, int Cin, int H, int W, long x_bs, long y_bs,
                                                                    long res_bs, int lrelu, float alpha, int accumulate) {
  __shared__ float red[(KS > 1 ? KS - 1 : 1) * NC * R * 4 * 64];
  const long hw = (long)H * W;
  const int lane = threadIdx.x & 63;
  const int ks = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int qpr = W / 4;                                               // quads per row
  const long q = (long)blockIdx.x * 64 + lane;                         // index over (row group, quad of the row)
  const long ngroups = (H + R - 1) / R;
  const bool qok = q < ngroups * qpr;
  const int b = blockIdx.y;
  const long qq = qok ? q : 0;
  const int oy0 = (int)(qq / qpr) * R, x0 = (int)(qq % qpr) * 4;
  float acc[NC][R][4];
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[c][r][i] = 0.f;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (long)b * x_bs), (short)0, (int)SOOB, 0x00020000);
  uint32_t vm[R + 2], vl[R + 2], vr[R + 2];
#pragma unroll
  for (int r = 0; r < R + 2; ++r) patch6_voffsets(oy0 - 1 + r, x0, H, W, qok, lane, vm[r], vl[r], vr[r]);
  const uint32_t hw4 = (uint32_t)(hw * 4);
#pragma unroll 2
  for (int ci = ks; ci < Cin; ci += KS) {
    float v[R + 2][6];
#pragma unroll
    for (int r = 0; r < R + 2; ++r) load_patch6_buf(xr, (uint32_t)ci * hw4, vm[r], vl[r], vr[r], v[r]);
#pragma unroll
    for (int r = 0; r < R + 2; ++r) patch6_sides(v[r], x0, W, lane);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const float* wc = w + ((long)c * Cin + ci) * 9;                   // wave-uniform -> scalar loads
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          const float ww = wc[a * 3 + t];
#pragma unroll
          for (int r = 0; r < R; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[c][r][i] = fmaf(ww, v[r + a][i + t], acc[c][r][i]);
        }
    }
  }
  if (KS > 1) {
    if (ks > 0) {
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
          for (int i = 0; i < 4; ++i) red[((((ks - 1) * NC + c) * R + r) * 4 + i) * 64 + lane] = acc[c][r][i];
    }
    __syncthreads();
    if (ks > 0) return;
#pragma unroll
    for (int k = 0; k < KS - 1; ++k)
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[c][r][i] += red[(((k * NC + c) * R + r) * 4 + i) * 64 + lane];
  }
  if (!qok) return;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const float bsv = bias ? bias[c] : 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (oy0 + r >= H) continue;
      const long p = (long)(oy0 + r) * W + x0;
      float* dst = y + (long)b * y_bs + (long)c * hw + p;
      f32x4s o;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float vv = acc[c][r][i] + bsv;
        if (lrelu) vv = irr_lrelu(vv);
        o[i] = vv;
      }
      if (res) {
        const f32x4s rr = *(const f32x4s*)(res + (long)b * res_bs + (long)c * hw + p);
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = rr[i] + alpha * o[i];
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] *= alpha;
      }
      if (accumulate) {
        const f32x4s d0 = *(const f32x4s*)dst;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] += d0[i];
      }
      *(f32x4s*)dst = o;
    }
  }
}

// R vertically adjacent output rows per thread (R + 2 input rows are loaded for R rows of products: the L2 re-read factor
// drops from 3 to (R + 2) / R, as in the forward kernel).
template <int NC, int R>
__global__ __launch_bounds__(256) void conv_smallco_wgrad4_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                                 float* __restrict__ ws, float* __restrict__ gbias, float alpha,
                                                                 int Cin, int H, int W, long x_bs, long gy_bs,
                                                                 int quads_per_block, int B, int bpb) {
  const long hw = (long)H * W;
  const int qpr = W / 4;
  const long nq = (long)((H + R - 1) / R) * qpr;            // "tall quads": 4 pixels x R rows
  const int ci = blockIdx.y;
  const long q0 = (long)blockIdx.x * quads_per_block;
  const long q1 = min(nq, q0 + quads_per_block);
  float acc[NC][9];
  float bsum[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    bsum[c] = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[c][t] = 0.f;
  }
  const uint32_t hw4 = (uint32_t)(hw * 4);
  const int lane = threadIdx.x & 63;
  // bpb samples per block (small pyramid levels: one sample has too few quads to amortise the block reduction)
  for (int b = blockIdx.z * bpb; b < min(B, (int)(blockIdx.z + 1) * bpb); ++b) {
  const __amdgpu_buffer_rsrc_t xr =
      __builtin_amdgcn_make_buffer_rsrc((void*)(x + (long)b * x_bs + (long)ci * hw), (short)0, (int)SOOB, 0x00020000);
  const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc((void*)(gy + (long)b * gy_bs), (short)0, (int)SOOB, 0x00020000);
  for (long qb = q0; qb < q1; qb += 256) {                 // wave-uniform trip count: the side pixels travel between lanes
    const long q = qb + threadIdx.x;
    const bool act = q < q1, inimg = q < nq;
    const long qq = inimg ? q : 0;
    const int oy0 = (int)(qq / qpr) * R, x0 = (int)(qq % qpr) * 4;
    float g[NC][R][4];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const bool ok = act && oy0 + r < H;
        const f32x4s gg = __builtin_bit_cast(f32x4s, __builtin_amdgcn_raw_buffer_load_b128(
                                                         gr, (int)(ok ? (uint32_t)(((oy0 + r) * W + x0) * 4) : SOOB), (int)((uint32_t)c * hw4), 0));
#pragma unroll
        for (int i = 0; i < 4; ++i) { g[c][r][i] = gg[i]; bsum[c] += gg[i]; }
      }
    float vv[R + 2][6];
#pragma unroll
    for (int r = 0; r < R + 2; ++r) {
      uint32_t vm, vl, vr;
      patch6_voffsets(oy0 - 1 + r, x0, H, W, inimg, lane, vm, vl, vr);
      load_patch6_buf(xr, 0u, vm, vl, vr, vv[r]);
    }
#pragma unroll
    for (int r = 0; r < R + 2; ++r) patch6_sides(vv[r], x0, W, lane);
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int r = 0; r < R; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[c][a * 3 + t] = fmaf(g[c][r][i], vv[r + a][i + t], acc[c][a * 3 + t]);
  }
  }
  __shared__ float red[4][NC * 9];
  const int wv = threadIdx.x >> 6;
  if (gbias && ci == 0) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      float sb = bsum[c];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) sb += __shfl_down(sb, o, 64);
      if (lane == 0) unsafeAtomicAdd(gbias + c, alpha * sb);
    }
  }
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      float sv = acc[c][t];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) sv += __shfl_down(sv, o, 64);
      if (lane == 0) red[wv][c * 9 + t] = sv;
    }
  __syncthreads();
  if (threadIdx.x < NC * 9) {
    const int c = threadIdx.x / 9, t = threadIdx.x - c * 9;
    const float sv = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    unsafeAtomicAdd(ws + ((long)c * 9 + t) * Cin + ci, alpha * sv);
  }
}

__global__ __launch_bounds__(256) void smallco_unpack_kernel(const float* __restrict__ ws, float* __restrict__ gw, int Cin,
                                                            int KK, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int tap = (int)(i % KK);
  const long r = i / KK;
  const int ci = (int)(r % Cin);
  const long co = r / Cin;
  gw[i] += ws[(co * KK + tap) * Cin + ci];
}

template <int KS>
int fwd_dispatch(int NC, dim3 grid, hipStream_t st, const float* x, const float* w, const float* bias, const float* res,
                 float* y, int B, int Cin, int H, int W, int dil, long x_bs, long y_bs, long res_bs, int lrelu,
                 float alpha, int accumulate) {
#define IRR_SMALL_FWD(N)                                                                                              \
  hipLaunchKernelGGL((conv_smallco_fwd_kernel<N, KS>), grid, dim3(256), 0, st, x, w, bias, res, y, B, Cin, H, W, dil, \
                     x_bs, y_bs, res_bs, lrelu, alpha, accumulate)
  switch (NC) {
    case 1: IRR_SMALL_FWD(1); break;
    case 2: IRR_SMALL_FWD(2); break;
    case 3: IRR_SMALL_FWD(3); break;
    case 4: IRR_SMALL_FWD(4); break;
    default: return IRR_EINVAL;
  }
#undef IRR_SMALL_FWD
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

}  // namespace

namespace {

// Data gradient of the tiny-Cout heads (stride 1): gx[b,ci,p] (+)= sum_{co<NC} sum_t gy[b,co,p - d(t)] * w[co][ci][t], then
// optionally *= LeakyReLU'(mask) for ci < nmask.  The MFMA kernels would run this with K = 9*NC <= 36 (94 % padding); it
// is a pure HBM stream over the Cin-channel gradient buffer (563 channels for conv_last).  A lane owns one pixel, keeps
// the 9*NC neighbourhood values of gy in registers for its whole channel range and streams over ci with wave-uniform
// (scalar) weight loads.
template <int NC>
__global__ __launch_bounds__(256) void conv_smallco_dgrad_kernel(const float* __restrict__ gy, const float* __restrict__ w,
                                                                float* __restrict__ gx, const float* __restrict__ mask, int Cin,
                                                                int H, int W, int dil, long gy_bs, long gx_bs, long mask_bs,
                                                                int nmask, int accumulate, int ci_per_block,
                                                                float* __restrict__ amax, int amax_channels) {
  const long hw = (long)H * W;
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  float vmax = 0.f;                                          // max |stored value| over channels < amax_channels (amax slot of the fp16x2 consumers)
  const int b = blockIdx.z;
  const int c0 = blockIdx.y * ci_per_block;
  const int c1 = min(Cin, c0 + ci_per_block);
  const bool pv = p < hw;
  const long pp = pv ? p : 0;
  const int oy = (int)(pp / W), ox = (int)(pp - (long)oy * W);
  float g[NC][9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    // gx[p] takes gy[p - d(t)]: the tap that maps p - d(t) onto p
    const int iy = oy + dil - (t / 3) * dil, ix = ox + dil - (t % 3) * dil;
    const bool ok = pv && iy >= 0 && iy < H && ix >= 0 && ix < W;
#pragma unroll
    for (int c = 0; c < NC; ++c) g[c][t] = ok ? gy[(long)b * gy_bs + (long)c * hw + (long)iy * W + ix] : 0.f;
  }
  for (int ci = c0; ci < c1; ++ci) {
    float v = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const float* wc = w + ((long)c * Cin + ci) * 9;          // wave-uniform -> scalar loads
#pragma unroll
      for (int t = 0; t < 9; ++t) v = fmaf(wc[t], g[c][t], v);
    }
    if (!pv) continue;
    float* dst = gx + (long)b * gx_bs + (long)ci * hw + p;
    if (accumulate) v += *dst;
    if (mask && ci < nmask) v *= irr_lrelu_grad(mask[(long)b * mask_bs + (long)ci * hw + p]);
    if (ci < amax_channels) vmax = x3_amax_fold(vmax, v);
    *dst = v;
  }
  if (amax && c0 < amax_channels) x3_amax_publish_block256(vmax, amax);      // (block-uniform condition; every thread arrives)
}

// Quad version (dilation 1, W % 4 == 0): a thread owns four adjacent pixels, keeps their 3 x 6 gy neighbourhood in
// registers and streams over the channels in batches of four: the read-modify-write operands of a batch (16-B buffer
// loads, out-of-range voffset for channels past the end) are all in flight before the first store.
// DUAL: the value BEFORE the LeakyReLU' mask is stored as well (gx_raw, same shape): OccUpsampleNetwork's backward needs the
// gradient of x2 = x_init + e both raw (the skip into x_init) and masked by e (into res_end_conv) -- one pass instead of this
// launch + an elementwise pass over two 32-channel full-resolution maps.
template <int NC, bool DUAL = false>
__global__ __launch_bounds__(256) void conv_smallco_dgrad4_kernel(const float* __restrict__ gy, const float* __restrict__ w,
                                                                 float* __restrict__ gx, const float* __restrict__ mask, int Cin,
                                                                 int H, int W, long gy_bs, long gx_bs, long mask_bs, int nmask,
                                                                 int accumulate, int ci_per_block, float* __restrict__ amax,
                                                                 int amax_channels, float* __restrict__ gx_raw = nullptr,
                                                                 long raw_bs = 0, float* __restrict__ chmax = nullptr) {
  constexpr int U = 4;
  // chmax (nullable, DUAL only; end of round 6): chmax[ci] = max(chmax[ci], max |stored gx[:, ci]|) -- per wave and channel one shuffle
  // reduction and one LDS maximum, per block and channel one look-then-atomic (the kernel is bound by its 12 bytes per element)
  __shared__ uint32_t chl[DUAL ? 64 : 1];
  if (DUAL && chmax) {
    if (threadIdx.x < 64) chl[threadIdx.x] = 0u;
    __syncthreads();
  }
  float vmax = 0.f;                                          // max |stored gx| over channels < amax_channels
  const long hw = (long)H * W;
  const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.z;
  const int c0 = blockIdx.y * ci_per_block;
  const int c1 = min(Cin, c0 + ci_per_block);
  const bool qok = q * 4 < hw;
  const long p = qok ? q * 4 : 0;
  const int oy = (int)(p / W), x0 = (int)(p - (long)oy * W);
  float gp[NC][3][6];
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int iy = oy - 1 + r;
      load_patch6(gy + (long)b * gy_bs + (long)c * hw + (long)iy * W, qok && iy >= 0 && iy < H, x0, W, gp[c][r]);
    }
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)(gx + (long)b * gx_bs), (short)0, (int)SOOB, 0x00020000);
  const __amdgpu_buffer_rsrc_t mr =
      __builtin_amdgcn_make_buffer_rsrc((void*)(mask ? mask + (long)b * mask_bs : gx), (short)0, (int)SOOB, 0x00020000);
  const __amdgpu_buffer_rsrc_t rr =
      __builtin_amdgcn_make_buffer_rsrc((void*)(DUAL ? gx_raw + (long)b * raw_bs : gx), (short)0, (int)SOOB, 0x00020000);
  const uint32_t vp = qok ? (uint32_t)(p * 4) : SOOB;
  const uint32_t hw4 = (uint32_t)(hw * 4);
  const int nm = mask ? nmask : 0;
  for (int ci = c0; ci < c1; ci += U) {
    f32x4s d[U], mk[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t so = (uint32_t)(ci + u) * hw4;
      d[u] = __builtin_bit_cast(f32x4s, __builtin_amdgcn_raw_buffer_load_b128(xr, (int)((accumulate && ci + u < c1) ? vp : SOOB), (int)so, 0));
      mk[u] = __builtin_bit_cast(f32x4s, __builtin_amdgcn_raw_buffer_load_b128(mr, (int)((ci + u < nm && ci + u < c1) ? vp : SOOB), (int)so, 0));
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int cc = min(ci + u, c1 - 1);                       // (clamped: the result of a channel past the end is dropped)
      f32x4s v = d[u];
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const float* wc = w + ((long)c * Cin + cc) * 9;          // wave-uniform -> scalar loads
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int t = 0; t < 3; ++t) {
            const float ww = wc[a * 3 + t];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = fmaf(ww, gp[c][2 - a][i + 2 - t], v[i]);
          }
      }
      if (DUAL)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int, v), rr,
                                               (int)(ci + u < c1 ? vp : SOOB), (int)((uint32_t)(ci + u) * hw4), 0);
      if (ci + u < nm) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] *= irr_lrelu_grad(mk[u][i]);
      }
      if (qok && ci + u < c1 && ci + u < amax_channels)
        vmax = x3_amax_fold(x3_amax_fold(x3_amax_fold(x3_amax_fold(vmax, v[0]), v[1]), v[2]), v[3]);
      if (DUAL && chmax) {                                       // (uniform branch)
        float m = (qok && ci + u < c1) ? x3_amax_fold(x3_amax_fold(x3_amax_fold(x3_amax_fold(0.f, v[0]), v[1]), v[2]), v[3]) : 0.f;
        m = x3_amax_wave(m);
        if ((threadIdx.x & 63) == 0 && ci + u < c1) atomicMax(&chl[(ci + u - c0) & 63], __builtin_bit_cast(uint32_t, m));
      }
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int, v), xr,
                                             (int)(ci + u < c1 ? vp : SOOB), (int)((uint32_t)(ci + u) * hw4), 0);
    }
  }
  if (DUAL && chmax) {
    __syncthreads();
    if ((int)threadIdx.x < c1 - c0 && threadIdx.x < 64) x3_amax_commit(__builtin_bit_cast(float, chl[threadIdx.x]), chmax + c0 + threadIdx.x);
  }
  if (amax && c0 < amax_channels) x3_amax_publish_block256(vmax, amax);      // (block-uniform condition; every thread arrives)
}

// Weight gradient of a layer with a TINY input-channel count (the first pyramid conv 3 -> 16, stride 2: a 32-wide MFMA
// ci-tile would be 91 % padding and ran at 2 TFLOP/s).  A thread owns output pixels, keeps the NCI x 9 input window of a
// pixel in registers (buffer loads, out-of-image taps answered with 0) and accumulates it against FOUR output channels
// (blockIdx.y = channel group); block reduction, then atomics into the [co][tap][ci] scratch image.
template <int NCI>
__global__ __launch_bounds__(256) void conv_smallci_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                                float* __restrict__ ws, float* __restrict__ gbias, float alpha,
                                                                int H, int W, int Cout, int OH, int OW, int stride, int dil,
                                                                int pad, long x_bs, long gy_bs, int pix_per_block) {
  constexpr int NW = NCI * 9;
  const long ohw = (long)OH * OW, hw = (long)H * W;
  const int co0 = blockIdx.y * 4, b = blockIdx.z;
  const long p0 = (long)blockIdx.x * pix_per_block, p1 = min(ohw, p0 + pix_per_block);
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (long)b * x_bs), (short)0, (int)SOOB, 0x00020000);
  const float* gb = gy + (long)b * gy_bs;
  float acc[4][NW];
  float bsum[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    bsum[c] = 0.f;
#pragma unroll
    for (int k = 0; k < NW; ++k) acc[c][k] = 0.f;
  }
  for (long p = p0 + threadIdx.x; p < p1; p += 256) {
    const int oy = (int)(p / OW), ox = (int)(p - (long)oy * OW);
    float xv[NW];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int iy = oy * stride - pad + (t / 3) * dil, ix = ox * stride - pad + (t % 3) * dil;
      const uint32_t vo = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? (uint32_t)((iy * W + ix) * 4) : SOOB;
#pragma unroll
      for (int ci = 0; ci < NCI; ++ci)
        xv[ci * 9 + t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, (int)vo, (int)((uint32_t)ci * (uint32_t)(hw * 4)), 0));
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float g = (co0 + c < Cout) ? gb[(long)(co0 + c) * ohw + p] : 0.f;
      bsum[c] += g;
#pragma unroll
      for (int k = 0; k < NW; ++k) acc[c][k] = fmaf(g, xv[k], acc[c][k]);
    }
  }
  __shared__ float red[4][4 * NW + 4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
#pragma unroll
    for (int k = 0; k < NW; ++k) {
      float v = acc[c][k];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
      if (lane == 0) red[wv][c * NW + k] = v;
    }
    float sb = bsum[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sb += __shfl_down(sb, o, 64);
    if (lane == 0) red[wv][4 * NW + c] = sb;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 4 * NW + 4; i += 256) {
    const float v = red[0][i] + red[1][i] + red[2][i] + red[3][i];
    if (i < 4 * NW) {
      const int c = i / NW, k = i - c * NW, ci = k / 9, t = k - ci * 9;
      if (co0 + c < Cout) unsafeAtomicAdd(ws + ((long)(co0 + c) * 9 + t) * NCI + ci, alpha * v);
    } else if (gbias && co0 + (i - 4 * NW) < Cout) {
      unsafeAtomicAdd(gbias + co0 + (i - 4 * NW), alpha * v);
    }
  }
}

}  // namespace

extern "C" int irr_conv2d_smallco_dgrad_f32(const float* gy, const float* w, float* gx, const float* mask, int B, int Cin,
                                            int H, int W, int Cout, int dil, long gy_bs, long gx_bs, long mask_bs, int nmask,
                                            int accumulate, float* amax, int amax_channels, void* stream) {
  if (!gy || !w || !gx || B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout < 1 || Cout > 2 || dil < 1 || B > 65535) return IRR_EINVAL;
  if (!amax) amax_channels = 0;
  const long hw = (long)H * W;
  const int pblocks = irr_cdiv(hw, 256);
  // enough blocks to fill the chip; every block re-reads the (tiny) gy neighbourhood once
  int split = (int)((4096 + (long)pblocks * B - 1) / ((long)pblocks * B));
  if (split < 1) split = 1;
  if (split > Cin) split = Cin;
  const int cpb = (Cin + split - 1) / split;
  dim3 grid(pblocks, irr_cdiv(Cin, cpb), B);
  hipStream_t st = (hipStream_t)stream;
  if (dil == 1 && (W & 3) == 0 && ((gy_bs | gx_bs | mask_bs) & 3) == 0 && !IRR_ENV_FLAG("IRR_SMALLCO_SCALAR")) {
    const int qblocks = irr_cdiv(hw / 4, 256);
    int split4 = (int)((2048 + (long)qblocks * B - 1) / ((long)qblocks * B));
    if (split4 < 1) split4 = 1;
    if (split4 > Cin) split4 = Cin;
    const int cpb4 = ((Cin + split4 - 1) / split4 + 3) / 4 * 4;
    dim3 grid4(qblocks, irr_cdiv(Cin, cpb4), B);
    if (Cout == 1)
      hipLaunchKernelGGL((conv_smallco_dgrad4_kernel<1>), grid4, dim3(256), 0, st, gy, w, gx, mask, Cin, H, W, gy_bs, gx_bs, mask_bs,
                         nmask, accumulate, cpb4, amax, amax_channels);
    else
      hipLaunchKernelGGL((conv_smallco_dgrad4_kernel<2>), grid4, dim3(256), 0, st, gy, w, gx, mask, Cin, H, W, gy_bs, gx_bs, mask_bs,
                         nmask, accumulate, cpb4, amax, amax_channels);
    IRR_LAUNCH_CHECK();
    return 0;
  }
  if (Cout == 1)
    hipLaunchKernelGGL((conv_smallco_dgrad_kernel<1>), grid, dim3(256), 0, st, gy, w, gx, mask, Cin, H, W, dil, gy_bs, gx_bs, mask_bs,
                       nmask, accumulate, cpb, amax, amax_channels);
  else
    hipLaunchKernelGGL((conv_smallco_dgrad_kernel<2>), grid, dim3(256), 0, st, gy, w, gx, mask, Cin, H, W, dil, gy_bs, gx_bs, mask_bs,
                       nmask, accumulate, cpb, amax, amax_channels);
  IRR_LAUNCH_CHECK();
  return 0;
}

// gx = LeakyReLU'(mask) * conv_transpose(gy, w) and gx_raw = conv_transpose(gy, w) from ONE pass (Cout = 1, dilation 1, W % 4 == 0,
// batch strides multiples of 4: the quad kernel); IRR_EINVAL otherwise -- the caller then runs irr_conv2d_smallco_dgrad_f32 +
// irr_lrelu_bwd_bias_f32.
static int smallco_dgrad_dual_impl(const float* gy, const float* w, float* gx, float* gx_raw, const float* mask, int B,
                                   int Cin, int H, int W, int Cout, long gy_bs, long gx_bs, long raw_bs, long mask_bs,
                                   float* amax, float* chmax, void* stream) {
  if (!gy || !w || !gx || !gx_raw || !mask || B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout != 1 || B > 65535) return IRR_EINVAL;
  if ((W & 3) || ((gy_bs | gx_bs | raw_bs | mask_bs) & 3)) return IRR_EINVAL;
  const long hw = (long)H * W;
  const int qblocks = irr_cdiv(hw / 4, 256);
  int split4 = (int)((2048 + (long)qblocks * B - 1) / ((long)qblocks * B));
  if (split4 < 1) split4 = 1;
  if (split4 > Cin) split4 = Cin;
  int cpb4 = ((Cin + split4 - 1) / split4 + 3) / 4 * 4;
  if (chmax && cpb4 > 64) {                                  // (the block's LDS maxima hold 64 channels)
    cpb4 = 64;
  }
  dim3 grid4(qblocks, irr_cdiv(Cin, cpb4), B);
  hipLaunchKernelGGL((conv_smallco_dgrad4_kernel<1, true>), grid4, dim3(256), 0, (hipStream_t)stream, gy, w, gx, mask, Cin, H, W, gy_bs,
                     gx_bs, mask_bs, Cin, 0, cpb4, amax, amax ? Cin : 0, gx_raw, raw_bs, chmax);
  IRR_LAUNCH_CHECK();
  return 0;
}

extern "C" int irr_conv2d_smallco_dgrad_dual_f32(const float* gy, const float* w, float* gx, float* gx_raw, const float* mask, int B,
                                                 int Cin, int H, int W, int Cout, long gy_bs, long gx_bs, long raw_bs, long mask_bs,
                                                 float* amax, void* stream) {
  return smallco_dgrad_dual_impl(gy, w, gx, gx_raw, mask, B, Cin, H, W, Cout, gy_bs, gx_bs, raw_bs, mask_bs, amax, nullptr, stream);
}

// (ABI 12) the same with chmax (nullable): chmax[ci] = max(chmax[ci], max |gx[:, ci]|) per channel of the masked form, Cin floats
extern "C" int irr_conv2d_smallco_dgrad_dual_ch_f32(const float* gy, const float* w, float* gx, float* gx_raw, const float* mask, int B,
                                                    int Cin, int H, int W, int Cout, long gy_bs, long gx_bs, long raw_bs, long mask_bs,
                                                    float* amax, float* chmax, void* stream) {
  return smallco_dgrad_dual_impl(gy, w, gx, gx_raw, mask, B, Cin, H, W, Cout, gy_bs, gx_bs, raw_bs, mask_bs, amax, chmax, stream);
}

extern "C" int irr_conv2d_smallco_fwd_f32(const float* x, const float* w, const float* bias, const float* res, float* y,
                                          int B, int Cin, int H, int W, int Cout, int k, int dil, long x_bs, long y_bs,
                                          long res_bs, int lrelu, float alpha, int accumulate, void* stream) {
  if (!x || !w || !y || B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout < 1 || Cout > 4 || B > 65535) return IRR_EINVAL;
  if ((k != 1 && k != 3) || dil < 1) return IRR_EINVAL;
  if (k == 3 && dil == 1 && (W & 3) == 0 && Cout <= 2 && ((x_bs | y_bs | res_bs) & 3) == 0 && !IRR_ENV_FLAG("IRR_SMALLCO_SCALAR")) {
    hipStream_t st = (hipStream_t)stream;
    static const int ks_env = getenv("IRR_SMALLCO_KS") ? atoi(getenv("IRR_SMALLCO_KS")) : 0;      // experiment switches
    static const int r_env = getenv("IRR_SMALLCO_R") ? atoi(getenv("IRR_SMALLCO_R")) : 0;
    // measured (563 -> 2, 64 samples): 96x112 R=4/KS=4 0.37 ms (R=1: 0.52), 48x56 R=1/KS=8 0.13 (KS=4: 0.20), 24x28 R=1/KS=8 0.08
    const int rr = r_env ? r_env : (((long)B * H * W >= 400000 && H % 4 == 0) ? 4 : 1);
    const int ks = ks_env ? ks_env : (Cin >= 64 ? (rr >= 4 ? 4 : 8) : (Cin >= 16 ? 4 : 1));
    dim3 grid4(irr_cdiv((long)((H + rr - 1) / rr) * (W / 4), 64), B, 1);
#define IRR_FWD4(NC, KS, RR)                                                                                              \
  hipLaunchKernelGGL((conv_smallco_fwd4_kernel<NC, KS, RR>), grid4, dim3(64 * KS), 0, st, x, w, bias, res, y, Cin, H, W, x_bs, y_bs, \
                     res_bs, lrelu, alpha, accumulate)
#define IRR_FWD4_R(NC, KS)                                                        \
  do {                                                                            \
    if (rr >= 4) IRR_FWD4(NC, KS, 4); else IRR_FWD4(NC, KS, 1); \
  } while (0)
    if (Cout == 1) {
      if (ks >= 8) IRR_FWD4_R(1, 8); else if (ks >= 4) IRR_FWD4_R(1, 4); else IRR_FWD4_R(1, 1);
    } else {
      if (ks >= 8) IRR_FWD4_R(2, 8); else if (ks >= 4) IRR_FWD4_R(2, 4); else IRR_FWD4_R(2, 1);
    }
#undef IRR_FWD4_R
#undef IRR_FWD4
    IRR_LAUNCH_CHECK();
    return 0;
  }
  if (k == 3 && Cout <= 2 && Cin >= 64 && (long)H * W <= 4096 && (long)irr_cdiv((long)H * W, 256) * B < 512 &&
      !IRR_ENV_FLAG("IRR_SMALLCO_NO_SLICES")) {
    // few pixels, many channels: 16 channel slices per 64 pixels
    const dim3 gs(irr_cdiv((long)H * W, 64), B, 1);
    if (Cout == 1)
      hipLaunchKernelGGL((conv_smallco_fwd_sl_kernel<1, 3, 16>), gs, dim3(1024), 0, (hipStream_t)stream, x, w, bias, res, y, Cin, H, W,
                         dil, x_bs, y_bs, res_bs, lrelu, alpha, accumulate);
    else
      hipLaunchKernelGGL((conv_smallco_fwd_sl_kernel<2, 3, 16>), gs, dim3(1024), 0, (hipStream_t)stream, x, w, bias, res, y, Cin, H, W,
                         dil, x_bs, y_bs, res_bs, lrelu, alpha, accumulate);
    IRR_LAUNCH_CHECK();
    return 0;
  }
  dim3 grid(irr_cdiv((long)H * W, 256), B, 1);
  return k == 3 ? fwd_dispatch<3>(Cout, grid, (hipStream_t)stream, x, w, bias, res, y, B, Cin, H, W, dil, x_bs, y_bs, res_bs,
                                  lrelu, alpha, accumulate)
                : fwd_dispatch<1>(Cout, grid, (hipStream_t)stream, x, w, bias, res, y, B, Cin, H, W, dil, x_bs, y_bs, res_bs,
                                  lrelu, alpha, accumulate);
}

extern "C" int irr_conv2d_smallco_wgrad_f32(const float* x, const float* gy, float* gw, float* ws, float* gbias, float alpha,
                                            int B, int Cin, int H, int W, int Cout, int k, int dil, long x_bs, long gy_bs,
                                            void* stream) {
  if (!x || !gy || !gw || !ws || B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout < 1 || Cout > 4) return IRR_EINVAL;
  if ((k != 1 && k != 3) || dil < 1 || B > 65535 || Cin > 65535) return IRR_EINVAL;
  const long n = (long)Cout * Cin * k * k;
  hipStream_t st = (hipStream_t)stream;
  IRR_HIP_TRY(irr_zero_async(ws, sizeof(float) * (size_t)n, st));
  const long hw = (long)H * W;
  // enough blocks to fill the chip, at least 4 pixels per thread
  long chunks = (2048 + (long)Cin * B - 1) / ((long)Cin * B);
  if (chunks < 1) chunks = 1;
  long ppb = (hw + chunks - 1) / chunks;
  if (ppb < 1024) ppb = 1024;
  if (k == 3 && dil == 1 && (W & 3) == 0 && Cout <= 2 && ((x_bs | gy_bs) & 3) == 0 && !IRR_ENV_FLAG("IRR_SMALLCO_SCALAR")) {
    static const int r_env = getenv("IRR_SMALLCO_WR") ? atoi(getenv("IRR_SMALLCO_WR")) : 0;      // experiment switch
    const int rr = r_env ? r_env : (((long)B * hw >= 400000 && H % 4 == 0) ? 4 : 1);
    const long ntq = (long)((H + rr - 1) / rr) * (W / 4);                 // tall quads per sample
    long qpb = ((ppb + 3) / 4 + rr - 1) / rr;
    if (qpb < 256) qpb = 256;
    // at least ~8 quads per thread and block: several samples per block at the small pyramid levels
    int bpb = (int)((8 * 256 + ntq - 1) / ntq);
    if (bpb < 1) bpb = 1;
    while (bpb > 1 && (long)Cin * irr_cdiv(B, bpb) < 1024) --bpb;          // keep the chip filled
    dim3 grid4(irr_cdiv(ntq, qpb), Cin, irr_cdiv(B, bpb));
#define IRR_WG4(NC, RR)                                                                                                         \
  hipLaunchKernelGGL((conv_smallco_wgrad4_kernel<NC, RR>), grid4, dim3(256), 0, st, x, gy, ws, gbias, alpha, Cin, H, W, x_bs, gy_bs, \
                     (int)qpb, B, bpb)
    if (Cout == 1) {
      if (rr >= 4) IRR_WG4(1, 4); else IRR_WG4(1, 1);
    } else {
      if (rr >= 4) IRR_WG4(2, 4); else IRR_WG4(2, 1);
    }
#undef IRR_WG4
    IRR_LAUNCH_CHECK();
    hipLaunchKernelGGL(smallco_unpack_kernel, dim3(irr_cdiv(n, 256)), dim3(256), 0, st, ws, gw, Cin, 9, n);
    IRR_LAUNCH_CHECK();
    return 0;
  }
  dim3 grid(irr_cdiv(hw, ppb), Cin, B);
#define IRR_SMALL_WG(N, K)                                                                                        \
  hipLaunchKernelGGL((conv_smallco_wgrad_kernel<N, K>), grid, dim3(256), 0, st, x, gy, ws, gbias, alpha, B, Cin, H, W, \
                     dil, x_bs, gy_bs, (int)ppb)
  if (k == 3) {
    switch (Cout) { case 1: IRR_SMALL_WG(1, 3); break; case 2: IRR_SMALL_WG(2, 3); break;
                    case 3: IRR_SMALL_WG(3, 3); break; default: IRR_SMALL_WG(4, 3); break; }
  } else {
    switch (Cout) { case 1: IRR_SMALL_WG(1, 1); break; case 2: IRR_SMALL_WG(2, 1); break;
                    case 3: IRR_SMALL_WG(3, 1); break; default: IRR_SMALL_WG(4, 1); break; }
  }
#undef IRR_SMALL_WG
  IRR_LAUNCH_CHECK();
  hipLaunchKernelGGL(smallco_unpack_kernel, dim3(irr_cdiv(n, 256)), dim3(256), 0, st, ws, gw, Cin, k * k, n);
  IRR_LAUNCH_CHECK();
  return 0;
}

extern "C" int irr_conv2d_smallci_wgrad_f32(const float* x, const float* gy, float* gw, float* ws, float* gbias, float alpha,
                                            int B, int Cin, int H, int W, int Cout, int OH, int OW, int stride, int dil,
                                            long x_bs, long gy_bs, void* stream) {
  if (!x || !gy || !gw || !ws || B <= 0 || Cin != 3 || H <= 0 || W <= 0 || Cout <= 0 || OH <= 0 || OW <= 0 || stride < 1 || dil < 1 ||
      B > 65535 || (long)Cin * H * W >= (1L << 29))
    return IRR_EINVAL;
  const long n = (long)Cout * Cin * 9;
  hipStream_t st = (hipStream_t)stream;
  IRR_HIP_TRY(irr_zero_async(ws, sizeof(float) * (size_t)n, st));
  const long ohw = (long)OH * OW;
  long ppb = 256 * 32;                                       // 32 pixels per thread amortise the block reduction
  while (ppb > 256 * 4 && irr_cdiv(ohw, ppb) * irr_cdiv(Cout, 4) * B < 1024) ppb /= 2;
  dim3 grid(irr_cdiv(ohw, ppb), irr_cdiv(Cout, 4), B);
  hipLaunchKernelGGL((conv_smallci_wgrad_kernel<3>), grid, dim3(256), 0, st, x, gy, ws, gbias, alpha, H, W, Cout, OH, OW, stride, dil,
                     dil, x_bs, gy_bs, (int)ppb);
  IRR_LAUNCH_CHECK();
  hipLaunchKernelGGL(smallco_unpack_kernel, dim3(irr_cdiv(n, 256)), dim3(256), 0, st, ws, gw, Cin, 9, n);
  IRR_LAUNCH_CHECK();
  return 0;
}
