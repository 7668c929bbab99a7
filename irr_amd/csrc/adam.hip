// Fused Adam over one flat fp32 arena (all 124 IRR-PWC tensors, 6.36 M elements): a single HBM-bound pass
// (read p, g, m, v; write p, m, v = 28 B/element = 178 MB/step) instead of torch.optim.Adam's per-tensor
// kernel chains.  Semantics = torch.optim.Adam (runtime.py:189) with L2 weight decay added to the gradient
// (scripts/IRR-PWC_flyingChairsOcc.sh:29-31: lr 1e-4, weight_decay 4e-4), amsgrad off.
#include "common.h"

namespace {
__global__ __launch_bounds__(256) void adam_kernel(float4* __restrict__ p, const float4* __restrict__ g,
                                                  float4* __restrict__ m, float4* __restrict__ v, long n4, long n,
                                                  double lr_d, double b1_d, double b2_d, float eps, float wd, float step_size,
                                                  float bc2_sqrt, float gscale, const float* __restrict__ step_dev) {
  // The scalars follow torch.optim.Adam's: 1 - beta, lr / (1 - beta1^t) and sqrt(1 - beta2^t) are formed in DOUBLE and then
  // rounded to fp32 (1.f - 0.999f is off by 1.3e-5 relative: the second moment came out that much too small).
  const float b2 = (float)b2_d, omb1 = (float)(1.0 - b1_d), omb2 = (float)(1.0 - b2_d);
  if (step_dev) {                    // step count kept on the device (hipGraph replays: nothing in the arguments changes per step)
    const double t = (double)step_dev[0];
    step_size = (float)(lr_d / (1.0 - pow(b1_d, t)));
    bc2_sqrt = (float)sqrt(1.0 - pow(b2_d, t));
  }
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 pp = p[i], gg = g[i], mm = m[i], vv = v[i];
    float* P = &pp.x; float* G = &gg.x; float* M = &mm.x; float* V = &vv.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gr = G[k] * gscale + wd * P[k];
      M[k] = M[k] + omb1 * (gr - M[k]);                       // lerp
      V[k] = b2 * V[k] + omb2 * gr * gr;
      const float denom = sqrtf(V[k]) / bc2_sqrt + eps;
      P[k] = P[k] - step_size * (M[k] / denom);
    }
    p[i] = pp; m[i] = mm; v[i] = vv;
  }
  // scalar tail
  if (blockIdx.x == 0) {
    float* ps = (float*)p; const float* gs = (const float*)g; float* ms = (float*)m; float* vs = (float*)v;
    for (long i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) {
      const float gr = gs[i] * gscale + wd * ps[i];
      ms[i] = ms[i] + omb1 * (gr - ms[i]);
      vs[i] = b2 * vs[i] + omb2 * gr * gr;
      ps[i] = ps[i] - step_size * (ms[i] / (sqrtf(vs[i]) / bc2_sqrt + eps));
    }
  }
}
}  // namespace

extern "C" int irr_adam_step_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n, double lr,
                                 double beta1, double beta2, double eps, double weight_decay, double bias_corr1,
                                 double bias_corr2, double grad_scale, const float* step_dev, void* stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || n <= 0) return IRR_EINVAL;
  if (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) return IRR_EINVAL;
  const long n4 = n / 4;
  int blocks = irr_cdiv(n4 > 0 ? n4 : 1, 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (float4*)param, (const float4*)grad,
                     (float4*)exp_avg, (float4*)exp_avg_sq, n4, n, lr, beta1, beta2, (float)eps, (float)weight_decay,
                     (float)(lr / (bias_corr1 > 0 ? bias_corr1 : 1.0)), (float)sqrt(bias_corr2 > 0 ? bias_corr2 : 1.0), (float)grad_scale,
                     step_dev);
  IRR_LAUNCH_CHECK();
  return 0;
}
