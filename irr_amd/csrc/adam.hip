// Fused Adam over one flat fp32 arena (all 124 IRR-PWC tensors, 6.36 M elements): a single HBM-bound pass
// (read p, g, m, v; write p, m, v = 28 B/element = 178 MB/step) instead of torch.optim.Adam's per-tensor
// kernel chains.  Semantics = torch.optim.Adam (runtime.py:189) with L2 weight decay added to the gradient
// (scripts/IRR-PWC_flyingChairsOcc.sh:29-31: lr 1e-4, weight_decay 4e-4), amsgrad off.
#include "common.h"

namespace {
__global__ __launch_bounds__(256) void adam_kernel(float4* __restrict__ p, const float4* __restrict__ g,
                                                  float4* __restrict__ m, float4* __restrict__ v, long n4, long n,
                                                  float lr, float b1, float b2, float eps, float wd, float bc1,
                                                  float bc2_sqrt, float gscale, const float* __restrict__ step_dev) {
  if (step_dev) {                    // step count kept on the device (hipGraph replays: nothing in the arguments changes per step)
    const float t = step_dev[0];
    bc1 = 1.f - powf(b1, t);
    bc2_sqrt = sqrtf(1.f - powf(b2, t));
  }
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 pp = p[i], gg = g[i], mm = m[i], vv = v[i];
    float* P = &pp.x; float* G = &gg.x; float* M = &mm.x; float* V = &vv.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gr = G[k] * gscale + wd * P[k];
      M[k] = M[k] + (1.f - b1) * (gr - M[k]);                 // lerp
      V[k] = b2 * V[k] + (1.f - b2) * gr * gr;
      const float denom = sqrtf(V[k]) / bc2_sqrt + eps;
      P[k] = P[k] - (lr / bc1) * (M[k] / denom);
    }
    p[i] = pp; m[i] = mm; v[i] = vv;
  }
  // scalar tail
  if (blockIdx.x == 0) {
    float* ps = (float*)p; const float* gs = (const float*)g; float* ms = (float*)m; float* vs = (float*)v;
    for (long i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) {
      const float gr = gs[i] * gscale + wd * ps[i];
      ms[i] = ms[i] + (1.f - b1) * (gr - ms[i]);
      vs[i] = b2 * vs[i] + (1.f - b2) * gr * gr;
      ps[i] = ps[i] - (lr / bc1) * (ms[i] / (sqrtf(vs[i]) / bc2_sqrt + eps));
    }
  }
}
}  // namespace

extern "C" int irr_adam_step_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n, float lr,
                                 float beta1, float beta2, float eps, float weight_decay, float bias_corr1,
                                 float bias_corr2, float grad_scale, const float* step_dev, void* stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || n <= 0) return IRR_EINVAL;
  if (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) return IRR_EINVAL;
  const long n4 = n / 4;
  int blocks = irr_cdiv(n4 > 0 ? n4 : 1, 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (float4*)param, (const float4*)grad,
                     (float4*)exp_avg, (float4*)exp_avg_sq, n4, n, lr, beta1, beta2, eps, weight_decay, bias_corr1,
                     sqrtf(bias_corr2), grad_scale, step_dev);
  IRR_LAUNCH_CHECK();
  return 0;
}
