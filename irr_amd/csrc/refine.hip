// Bilateral refinement tail of RefineFlow / RefineOcc (models/irr_modules.py:92-104, 130-139) and the
// nearest x2 upsampling of OccUpsampleNetwork (models/irr_modules.py:21-27).
//
// The reference builds softmax(-f^2), ReplicationPad2d, two Unfold tensors (9x the data) and a product+sum;
// here one thread per pixel keeps the 9 logits in registers, normalises them and gathers the clamped 3x3
// neighbourhood of every value channel directly: HBM sees f once, v once (neighbour reads hit L1/L2) and
// the output once.
#include "common.h"

namespace {

__device__ __forceinline__ void softmax9(const float* __restrict__ f, long plane, long p, float (&s)[9], float (&fv)[9]) {
  float mx = -INFINITY;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    fv[t] = f[(long)t * plane + p];
    s[t] = -fv[t] * fv[t];
    mx = fmaxf(mx, s[t]);
  }
  float den = 0.f;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    s[t] = expf(s[t] - mx);
    den += s[t];
  }
  const float inv = 1.f / den;
#pragma unroll
  for (int t = 0; t < 9; ++t) s[t] *= inv;
}

__global__ __launch_bounds__(256) void refine_tail_fwd_kernel(const float* __restrict__ f, const float* __restrict__ v,
                                                             float* __restrict__ out, int C, int H, int W, long f_bs,
                                                             long v_bs, long out_bs, float scale0, float scale1) {
  const long plane = (long)H * W;
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= plane) return;
  const int b = blockIdx.z;
  const int y = (int)(p / W), x = (int)(p - (long)y * W);
  float s[9], fv[9];
  softmax9(f + (long)b * f_bs, plane, p, s, fv);
  long nb[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int yy = min(max(y + t / 3 - 1, 0), H - 1), xx = min(max(x + t % 3 - 1, 0), W - 1);
    nb[t] = (long)yy * W + xx;
  }
  for (int c = 0; c < C; ++c) {
    const float* vc = v + (long)b * v_bs + (long)c * plane;
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) acc += s[t] * vc[nb[t]];
    out[(long)b * out_bs + (long)c * plane + p] = acc * (c == 0 ? scale0 : scale1);
  }
}

__global__ __launch_bounds__(256) void refine_tail_bwd_kernel(const float* __restrict__ f, const float* __restrict__ v,
                                                             const float* __restrict__ gout, float* __restrict__ gf,
                                                             float* __restrict__ gv, int C, int H, int W, long f_bs,
                                                             long v_bs, long gout_bs, long gf_bs, long gv_bs,
                                                             float scale0, float scale1) {
  const long plane = (long)H * W;
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= plane) return;
  const int b = blockIdx.z;
  const int y = (int)(p / W), x = (int)(p - (long)y * W);
  float s[9], fv[9], ds[9];
  softmax9(f + (long)b * f_bs, plane, p, s, fv);
  long nb[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int yy = min(max(y + t / 3 - 1, 0), H - 1), xx = min(max(x + t % 3 - 1, 0), W - 1);
    nb[t] = (long)yy * W + xx;
    ds[t] = 0.f;
  }
  for (int c = 0; c < C; ++c) {
    const float g = gout[(long)b * gout_bs + (long)c * plane + p] * (c == 0 ? scale0 : scale1);
    const float* vc = v + (long)b * v_bs + (long)c * plane;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      ds[t] += g * vc[nb[t]];
      if (gv) unsafeAtomicAdd(gv + (long)b * gv_bs + (long)c * plane + nb[t], g * s[t]);
    }
  }
  float dot = 0.f;
#pragma unroll
  for (int t = 0; t < 9; ++t) dot += s[t] * ds[t];
  if (gf) {
#pragma unroll
    for (int t = 0; t < 9; ++t) gf[(long)b * gf_bs + (long)t * plane + p] = s[t] * (ds[t] - dot) * (-2.f * fv[t]);
  }
}

__global__ __launch_bounds__(256) void nearest2x_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, int C,
                                                           int H, int W, long x_bs, long out_bs) {
  const int OW = 2 * W;
  const long oplane = 4L * H * W;
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= oplane) return;
  const int c = blockIdx.y, b = blockIdx.z;
  const int oy = (int)(p / OW), ox = (int)(p - (long)oy * OW);
  out[(long)b * out_bs + (long)c * oplane + p] = x[(long)b * x_bs + (long)c * H * W + (long)(oy >> 1) * W + (ox >> 1)];
}

__global__ __launch_bounds__(256) void nearest2x_bwd_kernel(const float* __restrict__ gout, float* __restrict__ gx, int C,
                                                           int H, int W, long gout_bs, long gx_bs) {
  const int OW = 2 * W;
  const long plane = (long)H * W;
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= plane) return;
  const int c = blockIdx.y, b = blockIdx.z;
  const int y = (int)(p / W), x = (int)(p - (long)y * W);
  const float* g = gout + (long)b * gout_bs + (long)c * 4 * plane + (long)(2 * y) * OW + 2 * x;
  gx[(long)b * gx_bs + (long)c * plane + p] = (g[0] + g[1]) + (g[OW] + g[OW + 1]);
}

}  // namespace

extern "C" int irr_refine_tail_fwd_f32(const float* f, const float* v, float* out, int B, int C, int H, int W, long f_bs,
                                       long v_bs, long out_bs, float scale0, float scale1, void* stream) {
  if (!f || !v || !out || B <= 0 || C <= 0 || H <= 0 || W <= 0 || B > 65535) return IRR_EINVAL;
  dim3 grid(irr_cdiv((long)H * W, 256), 1, B);
  hipLaunchKernelGGL(refine_tail_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, f, v, out, C, H, W, f_bs, v_bs,
                     out_bs, scale0, scale1);
  IRR_LAUNCH_CHECK();
  return 0;
}

extern "C" int irr_refine_tail_bwd_f32(const float* f, const float* v, const float* gout, float* gf, float* gv, int B,
                                       int C, int H, int W, long f_bs, long v_bs, long gout_bs, long gf_bs, long gv_bs,
                                       float scale0, float scale1, void* stream) {
  if (!f || !v || !gout || B <= 0 || C <= 0 || H <= 0 || W <= 0 || B > 65535) return IRR_EINVAL;
  if (!gf && !gv) return 0;
  if (gv) {
    const long plane = (long)H * W;
    if (gv_bs == (long)C * plane) {
      IRR_HIP_TRY(irr_zero_async(gv, sizeof(float) * (size_t)B * C * plane, (hipStream_t)stream));
    } else {
      for (int b = 0; b < B; ++b)
        IRR_HIP_TRY(irr_zero_async(gv + (long)b * gv_bs, sizeof(float) * (size_t)C * plane, (hipStream_t)stream));
    }
  }
  dim3 grid(irr_cdiv((long)H * W, 256), 1, B);
  hipLaunchKernelGGL(refine_tail_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, f, v, gout, gf, gv, C, H, W, f_bs,
                     v_bs, gout_bs, gf_bs, gv_bs, scale0, scale1);
  IRR_LAUNCH_CHECK();
  return 0;
}

extern "C" int irr_upsample_nearest2x_fwd_f32(const float* x, float* out, int B, int C, int H, int W, long x_bs,
                                              long out_bs, void* stream) {
  if (!x || !out || B <= 0 || C <= 0 || H <= 0 || W <= 0 || B > 65535 || C > 65535) return IRR_EINVAL;
  dim3 grid(irr_cdiv(4L * H * W, 256), C, B);
  hipLaunchKernelGGL(nearest2x_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, out, C, H, W, x_bs, out_bs);
  IRR_LAUNCH_CHECK();
  return 0;
}

extern "C" int irr_upsample_nearest2x_bwd_f32(const float* gout, float* gx, int B, int C, int H, int W, long gout_bs,
                                              long gx_bs, void* stream) {
  if (!gout || !gx || B <= 0 || C <= 0 || H <= 0 || W <= 0 || B > 65535 || C > 65535) return IRR_EINVAL;
  dim3 grid(irr_cdiv((long)H * W, 256), C, B);
  hipLaunchKernelGGL(nearest2x_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, gout, gx, C, H, W, gout_bs, gx_bs);
  IRR_LAUNCH_CHECK();
  return 0;
}
