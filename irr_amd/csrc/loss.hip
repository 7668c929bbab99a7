// Per-pixel parts of MultiScaleEPE_PWC_Bi_Occ_upsample (losses.py:515-577):
//   * target pyramid: adaptive_avg_pool2d to each level (losses.py:16-18) == s x s mean for the integer ratios that occur
//   * flow term:  sum_p || avgpool(target)_p - flow_p ||_2                     (losses.py:8-10, 544-549)
//   * occ  term:  per-sample sums of f1_score_bal_loss on sigmoid(logits)      (losses.py:39-48, 551-558)
// Forward kernels reduce in a FIXED order (round 4): every block stores its partial sum(s) in a caller-owned scratch array
// (one slot per block) and one finishing block adds the slots in index order -- no atomics, so flow_loss / occ_loss (and the
// balancing weights derived from them) are bit-reproducible run to run like the weight gradients.  Backward kernels are pure
// elementwise.  The scalar algebra (level weights, flow/occ balancing) stays on the host side.
#include "common.h"

namespace {

__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wv] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// out[b,c,y,x] = scale * mean_{s x s} in[b,c,y*s+i,x*s+j]
__global__ __launch_bounds__(256) void avgpool_kernel(const float* __restrict__ in, float* __restrict__ out, int h, int w,
                                                     int s, float scale, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int x = (int)(i % w);
  const long r = i / w;
  const int y = (int)(r % h);
  const long bc = r / h;
  const float* p = in + (bc * h * s + (long)y * s) * ((long)w * s) + (long)x * s;
  float acc = 0.f;
  for (int a = 0; a < s; ++a)
    for (int b = 0; b < s; ++b) acc += p[(long)a * w * s + b];
  out[i] = scale * acc / (float)(s * s);
}

// general adaptive_avg_pool2d (ATen: window [floor(o*H/h), ceil((o+1)*H/h)) per axis) for level sizes that do not divide the
// target size (odd pyramid sizes: 436x1024 -> 218x512, 109x256, 55x128, ...)
__global__ __launch_bounds__(256) void adaptive_avgpool_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W,
                                                              int h, int w, float scale, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int x = (int)(i % w);
  const long r = i / w;
  const int y = (int)(r % h);
  const long bc = r / h;
  const int y0 = (int)(((long)y * H) / h), y1 = (int)(((long)(y + 1) * H + h - 1) / h);
  const int x0 = (int)(((long)x * W) / w), x1 = (int)(((long)(x + 1) * W + w - 1) / w);
  const float* p = in + bc * (long)H * W;
  float acc = 0.f;
  for (int a = y0; a < y1; ++a)
    for (int b = x0; b < x1; ++b) acc += p[(long)a * W + b];
  out[i] = scale * (acc / (float)((y1 - y0) * (x1 - x0)));
}

// flow, tgt: (B,2,h,w) ; part[block] = weight * sum_p sqrt(du^2 + dv^2) over the block's pixels
__global__ __launch_bounds__(256) void epe_fwd_kernel(const float* __restrict__ flow, const float* __restrict__ tgt,
                                                     float* __restrict__ part, long hw, long flow_bs, long tgt_bs, float weight) {
  __shared__ float red[4];
  const int b = blockIdx.y;
  const float* f = flow + (long)b * flow_bs;
  const float* t = tgt + (long)b * tgt_bs;
  float s = 0.f;
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < hw; p += (long)gridDim.x * blockDim.x) {
    const float du = t[p] - f[p], dv = t[hw + p] - f[hw + p];
    s += sqrtf(du * du + dv * dv);
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) part[(long)b * gridDim.x + blockIdx.x] = weight * s;
}

// out[0] += part[0] + part[1] + ... + part[n-1], always added in the same order: thread t sums the slots t, t+256, ...,
// then the fixed shuffle / LDS tree of block_sum
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ part, long n, float* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.f;
  for (long i = threadIdx.x; i < n; i += 256) s += part[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) out[0] += s;
}

// gflow = gscale[0] * weight * (flow - tgt) / epe      (0 where epe == 0, as torch.norm's backward)
__global__ __launch_bounds__(256) void epe_bwd_kernel(const float* __restrict__ flow, const float* __restrict__ tgt,
                                                     const float* __restrict__ gscale, float* __restrict__ gflow, long hw,
                                                     long flow_bs, long tgt_bs, long g_bs, float weight) {
  const int b = blockIdx.y;
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= hw) return;
  const float* f = flow + (long)b * flow_bs;
  const float* t = tgt + (long)b * tgt_bs;
  float* g = gflow + (long)b * g_bs;
  const float du = f[p] - t[p], dv = f[hw + p] - t[hw + p];
  const float n = sqrtf(du * du + dv * dv);
  const float k = n > 0.f ? gscale[0] * weight / n : 0.f;
  g[p] = k * du;
  g[hw + p] = k * dv;
}

// part[b][bx][0..3] = { -sum t log(s+eps), -sum (1-t) log(1-s+eps), sum t, sum s } over the block's pixels, s = sigmoid(logit)
__global__ __launch_bounds__(256) void f1_sums_kernel(const float* __restrict__ logit, const float* __restrict__ tgt,
                                                     float* __restrict__ part, long hw, long l_bs, long t_bs) {
  __shared__ float red[4];
  const int b = blockIdx.y;
  const float* l = logit + (long)b * l_bs;
  const float* t = tgt + (long)b * t_bs;
  const float eps = 1e-8f;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < hw; p += (long)gridDim.x * blockDim.x) {
    const float s = 1.f / (1.f + expf(-l[p]));
    const float tt = t[p];
    a0 -= tt * logf(s + eps);
    a1 -= (1.f - tt) * logf((1.f - s) + eps);
    a2 += tt;
    a3 += s;
  }
  a0 = block_sum(a0, red);
  a1 = block_sum(a1, red);
  a2 = block_sum(a2, red);
  a3 = block_sum(a3, red);
  if (threadIdx.x == 0) *(float4*)(part + ((long)b * gridDim.x + blockIdx.x) * 4) = make_float4(a0, a1, a2, a3);
}

// sums[b][0..3] = part[b][0][.] + part[b][1][.] + ... (fixed order), one thread per sample
__global__ __launch_bounds__(64) void f1_fold_kernel(const float* __restrict__ part, float* __restrict__ sums, int B, int nbx) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4* p = (const float4*)part + (long)b * nbx;
  for (int i = 0; i < nbx; ++i) {
    const float4 v = p[i];
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  *(float4*)(sums + (long)b * 4) = a;
}

// loss_b = c * ( tp/D1 + fn/D2 ), D1 = st+sp+eps, D2 = 2N-st-sp+eps ; glogit = gscale[0]*weight * dloss/ds * s(1-s)
__global__ __launch_bounds__(256) void f1_bwd_kernel(const float* __restrict__ logit, const float* __restrict__ tgt,
                                                    const float* __restrict__ sums, const float* __restrict__ gscale,
                                                    float* __restrict__ glogit, long hw, long l_bs, long t_bs, long g_bs,
                                                    float weight) {
  const int b = blockIdx.y;
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= hw) return;
  const float eps = 1e-8f;
  const float tp = sums[b * 4 + 0], fn = sums[b * 4 + 1], st = sums[b * 4 + 2], sp = sums[b * 4 + 3];
  const float N = (float)hw;
  const float D1 = st + sp + eps, D2 = (N - st) + (N - sp) + eps;
  const float s = 1.f / (1.f + expf(-logit[(long)b * l_bs + p]));
  const float tt = tgt[(long)b * t_bs + p];
  const float dls = -tt / ((s + eps) * D1) - tp / (D1 * D1) + (1.f - tt) / (((1.f - s) + eps) * D2) + fn / (D2 * D2);
  glogit[(long)b * g_bs + p] = gscale[0] * weight * dls * s * (1.f - s);
}


// ---- all terms of one kind in ONE launch ----------------------------------------------------------------------------------
// The loss has 24 EPE terms and 24 balanced-F1 terms (7 pyramid levels x 2 or 4 outputs x 2 directions); each used to be
// 1 + 1 (EPE) or 2 + 1 (F1) launches of a few blocks -- ~150 launches of a step's ~2 000, issued right after the step's only
// host sync, where the GPU waits for the host.  The term table travels BY VALUE in the kernel arguments (no device copy).
struct LossTerms {
  IrrLossTerm t[IRR_LOSS_MAX_TERMS];
  int n;
};

__device__ __forceinline__ int find_term(const LossTerms& T, int blk) {
  int i = 0;
  while (i + 1 < T.n && T.t[i + 1].block0 <= blk) ++i;
  return i;
}

__global__ __launch_bounds__(256) void epe_multi_fwd_kernel(const LossTerms T, float* __restrict__ part) {
  __shared__ float red[4];
  const int ti = find_term(T, blockIdx.x);
  const IrrLossTerm& m = T.t[ti];
  const int lb = blockIdx.x - m.block0, nbx = m.nbx;
  const int b = lb / nbx, bx = lb - b * nbx;
  const long hw = m.hw;
  const float* f = m.pred + (long)b * m.pred_bs;
  const float* t = m.tgt + (long)b * m.tgt_bs;
  float s = 0.f;
  for (long p = (long)bx * 256 + threadIdx.x; p < hw; p += (long)nbx * 256) {
    const float du = t[p] - f[p], dv = t[hw + p] - f[hw + p];
    s += sqrtf(du * du + dv * dv);
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = m.weight * s;
}

__global__ __launch_bounds__(256) void epe_multi_bwd_kernel(const LossTerms T, const float* __restrict__ gscale) {
  const int ti = find_term(T, blockIdx.x);
  const IrrLossTerm& m = T.t[ti];
  const int lb = blockIdx.x - m.block0, nbx = m.nbx;
  const int b = lb / nbx;
  const long p = (long)(lb - b * nbx) * 256 + threadIdx.x;
  const long hw = m.hw;
  if (p >= hw) return;
  const float* f = m.pred + (long)b * m.pred_bs;
  const float* t = m.tgt + (long)b * m.tgt_bs;
  float* g = m.grad + (long)b * m.grad_bs;
  const float du = f[p] - t[p], dv = f[hw + p] - t[hw + p];
  const float n = sqrtf(du * du + dv * dv);
  const float k = n > 0.f ? gscale[0] * m.weight / n : 0.f;
  g[p] = k * du;
  g[hw + p] = k * dv;
}

__global__ __launch_bounds__(256) void f1_multi_sums_kernel(const LossTerms T, float* __restrict__ part) {
  __shared__ float red[4];
  const int ti = find_term(T, blockIdx.x);
  const IrrLossTerm& m = T.t[ti];
  const int lb = blockIdx.x - m.block0, nbx = m.nbx;
  const int b = lb / nbx, bx = lb - b * nbx;
  const long hw = m.hw;
  const float* l = m.pred + (long)b * m.pred_bs;
  const float* t = m.tgt + (long)b * m.tgt_bs;
  const float eps = 1e-8f;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  for (long p = (long)bx * 256 + threadIdx.x; p < hw; p += (long)nbx * 256) {
    const float s = 1.f / (1.f + expf(-l[p]));
    const float tt = t[p];
    a0 -= tt * logf(s + eps);
    a1 -= (1.f - tt) * logf((1.f - s) + eps);
    a2 += tt;
    a3 += s;
  }
  a0 = block_sum(a0, red);
  a1 = block_sum(a1, red);
  a2 = block_sum(a2, red);
  a3 = block_sum(a3, red);
  if (threadIdx.x == 0) *(float4*)(part + (long)blockIdx.x * 4) = make_float4(a0, a1, a2, a3);
}

// ONE block finishes all terms in a fixed order: wave w takes the terms w, w+4, ...; a lane folds the block partials of its
// samples in index order into aux[b][0..3] (kept for the backward pass), the per-sample algebra
// weight * ( sum_b tp/(st+sp+eps) + sum_b fn/((N-st)+(N-sp)+eps) ) is reduced by the fixed shuffle tree, and thread 0 adds
// the term values to out[0] in term order
__global__ __launch_bounds__(256) void f1_multi_value_kernel(const LossTerms T, const float* __restrict__ part,
                                                            float* __restrict__ out) {
  __shared__ float termval[IRR_LOSS_MAX_TERMS];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int ti = wv; ti < T.n; ti += 4) {
    const IrrLossTerm& m = T.t[ti];
    const float eps = 1e-8f, n = (float)m.hw;
    float s1 = 0.f, s2 = 0.f;
    for (int b = lane; b < m.B; b += 64) {
      const float4* p = (const float4*)part + (long)m.block0 + (long)b * m.nbx;
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int i = 0; i < m.nbx; ++i) {
        const float4 v = p[i];
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
      }
      *(float4*)(m.aux + (long)b * 4) = a;
      s1 += a.x / (a.z + a.w + eps);
      s2 += a.y / ((n - a.z) + (n - a.w) + eps);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      s1 += __shfl_down(s1, o, 64);
      s2 += __shfl_down(s2, o, 64);
    }
    if (lane == 0) termval[ti] = (s1 + s2) * m.weight;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < T.n; ++i) s += termval[i];
    out[0] += s;
  }
}

__global__ __launch_bounds__(256) void f1_multi_bwd_kernel(const LossTerms T, const float* __restrict__ gscale) {
  const int ti = find_term(T, blockIdx.x);
  const IrrLossTerm& m = T.t[ti];
  const int lb = blockIdx.x - m.block0, nbx = m.nbx;
  const int b = lb / nbx;
  const long p = (long)(lb - b * nbx) * 256 + threadIdx.x;
  const long hw = m.hw;
  if (p >= hw) return;
  const float eps = 1e-8f;
  const float* sums = m.aux + b * 4;
  const float tp = sums[0], fn = sums[1], st = sums[2], sp = sums[3];
  const float N = (float)hw;
  const float D1 = st + sp + eps, D2 = (N - st) + (N - sp) + eps;
  const float s = 1.f / (1.f + expf(-m.pred[(long)b * m.pred_bs + p]));
  const float tt = m.tgt[(long)b * m.tgt_bs + p];
  const float dls = -tt / ((s + eps) * D1) - tp / (D1 * D1) + (1.f - tt) / (((1.f - s) + eps) * D2) + fn / (D2 * D2);
  m.grad[(long)b * m.grad_bs + p] = gscale[0] * m.weight * dls * s * (1.f - s);
}

// grid-stride blocks per sample of the forward reductions (= partial-sum slots per sample)
static int reduce_blocks_per_sample(long hw) {
  int bx = irr_cdiv(hw, 256 * 8);
  return bx > 256 ? 256 : bx;
}

// lays the terms out over the blocks of one launch; mode 0: a few grid-stride blocks per sample (reductions), 1: one thread per pixel
static long layout_terms(const IrrLossTerm* in, int n, int mode, LossTerms* T) {
  if (!in || n <= 0 || n > IRR_LOSS_MAX_TERMS) return -1;
  long blk = 0;
  for (int i = 0; i < n; ++i) {
    IrrLossTerm t = in[i];
    if (!t.pred || !t.tgt || t.B <= 0 || t.hw <= 0) return -1;
    const int nbx = mode == 0 ? reduce_blocks_per_sample(t.hw) : irr_cdiv(t.hw, 256);
    t.nbx = nbx;
    t.block0 = (int)blk;
    blk += (long)nbx * t.B;
    T->t[i] = t;
  }
  T->n = n;
  return blk < 0x7fffffffL ? blk : -1;
}

}  // namespace

extern "C" int irr_avgpool_f32(const float* in, float* out, int BC, int h, int w, int s, float scale, void* stream) {
  if (!in || !out || BC <= 0 || h <= 0 || w <= 0 || s <= 0) return IRR_EINVAL;
  const long n = (long)BC * h * w;
  hipLaunchKernelGGL(avgpool_kernel, dim3(irr_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, in, out, h, w, s, scale, n);
  IRR_LAUNCH_CHECK();
  return 0;
}

extern "C" int irr_adaptive_avgpool_f32(const float* in, float* out, int BC, int H, int W, int h, int w, float scale, void* stream) {
  if (!in || !out || BC <= 0 || H <= 0 || W <= 0 || h <= 0 || w <= 0) return IRR_EINVAL;
  const long n = (long)BC * h * w;
  hipLaunchKernelGGL(adaptive_avgpool_kernel, dim3(irr_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, in, out, H, W, h, w, scale, n);
  IRR_LAUNCH_CHECK();
  return 0;
}

// partial-sum slots (floats) the forward reductions below need in `scratch`: one per block (EPE), four per block (F1)
extern "C" long irr_loss_partial_blocks(int B, long HW) {
  if (B <= 0 || HW <= 0) return IRR_EINVAL;
  return (long)B * reduce_blocks_per_sample(HW);
}

extern "C" int irr_epe_sum_fwd_f32(const float* flow, const float* tgt, float* out, int B, int HW, long flow_bs,
                                   long tgt_bs, float weight, float* scratch, long scratch_elems, void* stream) {
  if (!flow || !tgt || !out || !scratch || B <= 0 || HW <= 0 || B > 65535) return IRR_EINVAL;
  const int bx = reduce_blocks_per_sample(HW);
  if (scratch_elems < (long)bx * B) return IRR_EINVAL;
  hipLaunchKernelGGL(epe_fwd_kernel, dim3(bx, B), dim3(256), 0, (hipStream_t)stream, flow, tgt, scratch, (long)HW, flow_bs,
                     tgt_bs, weight);
  IRR_LAUNCH_CHECK();
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, scratch, (long)bx * B, out);
  IRR_LAUNCH_CHECK();
  return 0;
}

extern "C" int irr_epe_sum_bwd_f32(const float* flow, const float* tgt, const float* gscale, float* gflow, int B, int HW,
                                   long flow_bs, long tgt_bs, long g_bs, float weight, void* stream) {
  if (!flow || !tgt || !gscale || !gflow || B <= 0 || HW <= 0 || B > 65535) return IRR_EINVAL;
  hipLaunchKernelGGL(epe_bwd_kernel, dim3(irr_cdiv(HW, 256), B), dim3(256), 0, (hipStream_t)stream, flow, tgt, gscale,
                     gflow, (long)HW, flow_bs, tgt_bs, g_bs, weight);
  IRR_LAUNCH_CHECK();
  return 0;
}

extern "C" int irr_f1bal_sums_f32(const float* logit, const float* tgt, float* sums, int B, int HW, long l_bs, long t_bs,
                                  float* scratch, long scratch_elems, void* stream) {
  if (!logit || !tgt || !sums || !scratch || B <= 0 || HW <= 0 || B > 65535) return IRR_EINVAL;
  const int bx = reduce_blocks_per_sample(HW);
  if (scratch_elems < 4L * bx * B || ((size_t)scratch & 15) || ((size_t)sums & 15)) return IRR_EINVAL;
  hipLaunchKernelGGL(f1_sums_kernel, dim3(bx, B), dim3(256), 0, (hipStream_t)stream, logit, tgt, scratch, (long)HW, l_bs, t_bs);
  IRR_LAUNCH_CHECK();
  hipLaunchKernelGGL(f1_fold_kernel, dim3(irr_cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, scratch, sums, B, bx);
  IRR_LAUNCH_CHECK();
  return 0;
}

// out[0] = scale * ( sum_b tp/(st+sp+eps) + sum_b fn/((N-st)+(N-sp)+eps) ): the per-sample algebra of f1_score_bal_loss
// (losses.py:39-48) on the four sums per sample, one wave, fixed order
__global__ __launch_bounds__(64) void f1_value_kernel(const float* __restrict__ sums, float* __restrict__ out, int B, float n,
                                                     float scale) {
  const float eps = 1e-8f;
  float s1 = 0.f, s2 = 0.f;
  for (int b = threadIdx.x; b < B; b += 64) {
    const float tp = sums[b * 4 + 0], fn = sums[b * 4 + 1], st = sums[b * 4 + 2], sp = sums[b * 4 + 3];
    s1 += tp / (st + sp + eps);
    s2 += fn / ((n - st) + (n - sp) + eps);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s1 += __shfl_down(s1, o, 64);
    s2 += __shfl_down(s2, o, 64);
  }
  if (threadIdx.x == 0) out[0] = (s1 + s2) * scale;
}

extern "C" int irr_f1bal_value_f32(const float* sums, float* out, int B, int HW, float scale, void* stream) {
  if (!sums || !out || B <= 0 || HW <= 0) return IRR_EINVAL;
  hipLaunchKernelGGL(f1_value_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sums, out, B, (float)HW, scale);
  IRR_LAUNCH_CHECK();
  return 0;
}

extern "C" int irr_f1bal_bwd_f32(const float* logit, const float* tgt, const float* sums, const float* gscale,
                                 float* glogit, int B, int HW, long l_bs, long t_bs, long g_bs, float weight,
                                 void* stream) {
  if (!logit || !tgt || !sums || !gscale || !glogit || B <= 0 || HW <= 0 || B > 65535) return IRR_EINVAL;
  hipLaunchKernelGGL(f1_bwd_kernel, dim3(irr_cdiv(HW, 256), B), dim3(256), 0, (hipStream_t)stream, logit, tgt, sums,
                     gscale, glogit, (long)HW, l_bs, t_bs, g_bs, weight);
  IRR_LAUNCH_CHECK();
  return 0;
}

// ---- multi-term entry points: `terms` is a HOST array of nterms IrrLossTerm records (include/irr_hip.h) ----
extern "C" int irr_epe_sum_multi_fwd_f32(const void* terms, int nterms, float* out, float* scratch, long scratch_elems,
                                         void* stream) {
  LossTerms T;
  const long nb = layout_terms((const IrrLossTerm*)terms, nterms, 0, &T);
  if (nb <= 0 || !out || !scratch || scratch_elems < nb) return IRR_EINVAL;
  hipLaunchKernelGGL(epe_multi_fwd_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, T, scratch);
  IRR_LAUNCH_CHECK();
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, scratch, nb, out);
  IRR_LAUNCH_CHECK();
  return 0;
}

extern "C" int irr_epe_sum_multi_bwd_f32(const void* terms, int nterms, const float* gscale, void* stream) {
  LossTerms T;
  const long nb = layout_terms((const IrrLossTerm*)terms, nterms, 1, &T);
  if (nb <= 0 || !gscale) return IRR_EINVAL;
  for (int i = 0; i < nterms; ++i)
    if (!T.t[i].grad) return IRR_EINVAL;
  hipLaunchKernelGGL(epe_multi_bwd_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, T, gscale);
  IRR_LAUNCH_CHECK();
  return 0;
}

// per-sample sums of every term into its aux[B][4] (overwritten), then out[0] += sum over terms of weight * per-sample algebra
extern "C" int irr_f1bal_multi_fwd_f32(const void* terms, int nterms, float* out, float* scratch, long scratch_elems,
                                       void* stream) {
  LossTerms T;
  const long nb = layout_terms((const IrrLossTerm*)terms, nterms, 0, &T);
  if (nb <= 0 || !out || !scratch || scratch_elems < 4 * nb || ((size_t)scratch & 15)) return IRR_EINVAL;
  for (int i = 0; i < nterms; ++i)
    if (!T.t[i].aux || ((size_t)T.t[i].aux & 15)) return IRR_EINVAL;
  hipLaunchKernelGGL(f1_multi_sums_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, T, scratch);
  IRR_LAUNCH_CHECK();
  hipLaunchKernelGGL(f1_multi_value_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, T, (const float*)scratch, out);
  IRR_LAUNCH_CHECK();
  return 0;
}

extern "C" int irr_f1bal_multi_bwd_f32(const void* terms, int nterms, const float* gscale, void* stream) {
  LossTerms T;
  const long nb = layout_terms((const IrrLossTerm*)terms, nterms, 1, &T);
  if (nb <= 0 || !gscale) return IRR_EINVAL;
  for (int i = 0; i < nterms; ++i)
    if (!T.t[i].grad || !T.t[i].aux) return IRR_EINVAL;
  hipLaunchKernelGGL(f1_multi_bwd_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, T, gscale);
  IRR_LAUNCH_CHECK();
  return 0;
}
