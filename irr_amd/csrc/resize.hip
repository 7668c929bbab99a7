// Bilinear resize with align_corners=True -- upsample2d_as (models/pwc_modules.py:65-67), used for the
// x2 flow/occ upsampling between pyramid levels, the raw-image downsizing per level
// (models/IRR_PWC.py:82-85,126-127,151-152) and the eval output (:176-177).
//
// ATen semantics: scale = (in-1)/(out-1) (0 when out==1); src = scale*dst; i0 = floor(src);
// i1 = min(i0+1, in-1); l1 = src-i0; l0 = 1-l1;  out = l0y*(l0x*v00 + l1x*v01) + l1y*(l0x*v10 + l1x*v11).
// Backward is written as a gather over the (few) output pixels that touch an input pixel, so it is
// deterministic and needs no atomics.
//
// HP = true: align_corners=False ("half-pixel") -- the fallback of upsample_factor2 for odd pyramid sizes
// (models/irr_modules.py:21-27: nearest x2, then bilinear to the guide's size).  ATen: scale = in/out;
// src = max(scale*(dst+0.5)-0.5, 0); i0 = floor(src); i1 = i0 + (i0 < in-1); same blend.
#include "common.h"

namespace {

template <bool HP>
__device__ __forceinline__ float rs_scale(int in, int out) {
  if (HP) return (float)in / (float)out;
  return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
}
template <bool HP>
__device__ __forceinline__ float rs_src(float scale, int dst) {
  if (HP) return fmaxf(scale * ((float)dst + 0.5f) - 0.5f, 0.f);
  return scale * dst;
}

template <bool HP>
__global__ __launch_bounds__(256) void resize_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, int C,
                                                        int H, int W, int OH, int OW, long x_bs, long out_bs,
                                                        float alpha) {
  const long oplane = (long)OH * OW;
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= oplane) return;
  const int b = blockIdx.z;
  const int oy = (int)(p / OW), ox = (int)(p - (long)oy * OW);
  const float sy = rs_src<HP>(rs_scale<HP>(H, OH), oy), sx = rs_src<HP>(rs_scale<HP>(W, OW), ox);
  const int y0 = (int)sy, x0 = (int)sx;
  const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
  const float ly1 = sy - y0, lx1 = sx - x0, ly0 = 1.f - ly1, lx0 = 1.f - lx1;
  const float* xb = x + (long)b * x_bs;
  float* ob = out + (long)b * out_bs + p;
  const long plane = (long)H * W;
  for (int c = blockIdx.y; c < C; c += gridDim.y) {
    const float* xc = xb + (long)c * plane;
    const float v = ly0 * (lx0 * xc[(long)y0 * W + x0] + lx1 * xc[(long)y0 * W + x1]) +
                    ly1 * (lx0 * xc[(long)y1 * W + x0] + lx1 * xc[(long)y1 * W + x1]);
    ob[(long)c * oplane] = alpha * v;
  }
}

// 1-D contribution of output index o to input index i (hat function sampled on the output lattice)
template <bool HP>
__device__ __forceinline__ float tap_weight(int i, int o, int in, float scale) {
  const float s = rs_src<HP>(scale, o);
  const int i0 = (int)s;
  const int i1 = min(i0 + 1, in - 1);
  const float l1 = s - i0;
  float w = 0.f;
  if (i0 == i) w += 1.f - l1;
  if (i1 == i) w += l1;
  return w;
}

template <bool HP, bool ACC = false>
__global__ __launch_bounds__(256) void resize_bwd_kernel(const float* __restrict__ gout, float* __restrict__ gx, int C,
                                                        int H, int W, int OH, int OW, long gout_bs, long gx_bs,
                                                        float alpha) {
  const long plane = (long)H * W;
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= plane) return;
  const int b = blockIdx.z;
  const int iy = (int)(p / W), ix = (int)(p - (long)iy * W);
  const float scy = rs_scale<HP>(H, OH), scx = rs_scale<HP>(W, OW);
  // candidate output range touching input row iy: src in (iy-1, iy+1), i.e. o in ((iy-1)/s, (iy+1)/s) for src = s*o and
  // o in ((iy-0.5)/s - 0.5, (iy+1.5)/s - 0.5) for the half-pixel lattice src = s*(o+0.5) - 0.5 (outputs clamped to src = 0
  // only touch iy = 0, whose range starts at 0 anyway); one spare candidate each side, tap_weight() decides
  int oy_lo, oy_hi, ox_lo, ox_hi;
  const float ha = HP ? 0.5f : 1.f, hb = HP ? 1.5f : 1.f, hc = HP ? 0.5f : 0.f;
  if (scy > 0.f) { oy_lo = max(0, (int)floorf((iy - ha) / scy - hc) - 1); oy_hi = min(OH - 1, (int)ceilf((iy + hb) / scy - hc) + 1); }
  else { oy_lo = 0; oy_hi = OH - 1; }
  if (scx > 0.f) { ox_lo = max(0, (int)floorf((ix - ha) / scx - hc) - 1); ox_hi = min(OW - 1, (int)ceilf((ix + hb) / scx - hc) + 1); }
  else { ox_lo = 0; ox_hi = OW - 1; }
  const long oplane = (long)OH * OW;
  for (int c = blockIdx.y; c < C; c += gridDim.y) {
    const float* gc = gout + (long)b * gout_bs + (long)c * oplane;
    float acc = 0.f;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      const float wy = tap_weight<HP>(iy, oy, H, scy);
      if (wy == 0.f) continue;
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        const float wx = tap_weight<HP>(ix, ox, W, scx);
        if (wx != 0.f) acc += wy * wx * gc[(long)oy * OW + ox];
      }
    }
    float* dst = gx + (long)b * gx_bs + (long)c * plane + p;
    *dst = ACC ? *dst + alpha * acc : alpha * acc;
  }
}

// Backward of a DOWNSAMPLING by >= 2 per axis (align_corners=True; the raw images resized to every pyramid level,
// models/IRR_PWC.py:126-127 -- their gradient exists because the reference's step makes the inputs require grad, runtime.py:158-162):
// neighbouring output pixels are >= 2 input pixels apart, so their 2 x 2 footprints are disjoint and gx is zero everywhere else.  The
// gather form above evaluates ~9 candidate taps for every INPUT pixel (215 us for 384x448 -> 6x7 at 64 x 3 planes, five such calls
// per step); here gx is zero-filled and one thread per OUTPUT pixel stores its (at most) four contributions -- plain stores, no atomics.
// ACC: gx += ... (the zero fill is the caller's, or an earlier call's: irr_resize_bilinear_ac_bwd_acc_f32) -- footprints of ONE call are
// disjoint, calls on one stream are ordered, so the read-modify-write needs no atomics either.
template <bool ACC = false>
__global__ __launch_bounds__(256) void resize_bwd_sparse_kernel(const float* __restrict__ gout, float* __restrict__ gx, int C, int H,
                                                               int W, int OH, int OW, long gout_bs, long gx_bs, float alpha) {
  const long oplane = (long)OH * OW;
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= oplane) return;
  const int b = blockIdx.z;
  const int oy = (int)(p / OW), ox = (int)(p - (long)oy * OW);
  const float sy = rs_src<false>(rs_scale<false>(H, OH), oy), sx = rs_src<false>(rs_scale<false>(W, OW), ox);
  const int y0 = (int)sy, x0 = (int)sx;
  const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
  const float ly1 = sy - y0, lx1 = sx - x0;
  // per-axis weights exactly as tap_weight() forms them (a clamped second tap falls onto the first: the weights add up)
  const float wy0 = y1 == y0 ? (1.f - ly1) + ly1 : 1.f - ly1, wx0 = x1 == x0 ? (1.f - lx1) + lx1 : 1.f - lx1;
  const long plane = (long)H * W;
  for (int c = blockIdx.y; c < C; c += gridDim.y) {
    const float g = gout[(long)b * gout_bs + (long)c * oplane + p];
    float* gc = gx + (long)b * gx_bs + (long)c * plane;
    auto put = [&](long o, float v) { gc[o] = ACC ? gc[o] + v : v; };
    put((long)y0 * W + x0, alpha * (wy0 * wx0 * g));
    if (x1 != x0) put((long)y0 * W + x1, alpha * (wy0 * lx1 * g));
    if (y1 != y0) {
      put((long)y1 * W + x0, alpha * (ly1 * wx0 * g));
      if (x1 != x0) put((long)y1 * W + x1, alpha * (ly1 * lx1 * g));
    }
  }
}

}  // namespace

extern "C" int irr_resize_bilinear_ac_fwd_f32(const float* x, float* out, int B, int C, int H, int W, int OH, int OW,
                                              long x_bs, long out_bs, float alpha, void* stream) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0 || !x || !out || B > 65535) return IRR_EINVAL;
  dim3 grid(irr_cdiv((long)OH * OW, 256), C < 8 ? C : 8, B);
  hipLaunchKernelGGL(resize_fwd_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, x, out, C, H, W, OH, OW, x_bs, out_bs,
                     alpha);
  IRR_LAUNCH_CHECK();
  return 0;
}

static int resize_ac_bwd_impl(const float* gout, float* gx, int B, int C, int H, int W, int OH, int OW,
                              long gout_bs, long gx_bs, float alpha, int accumulate, void* stream) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0 || !gout || !gx || B > 65535) return IRR_EINVAL;
  if (OH > 1 && OW > 1 && (H - 1) >= 2 * (OH - 1) && (W - 1) >= 2 * (OW - 1) && !IRR_ENV_FLAG("IRR_RESIZE_BWD_GATHER")) {
    // downsampling by >= 2 per axis: disjoint 2 x 2 footprints (see resize_bwd_sparse_kernel)
    const size_t per = sizeof(float) * (size_t)C * H * W;
    if (!accumulate) {
      if (gx_bs == (long)C * H * W) {
        IRR_HIP_TRY(irr_zero_async(gx, per * (size_t)B, (hipStream_t)stream));
      } else {
        for (int b = 0; b < B; ++b) IRR_HIP_TRY(irr_zero_async(gx + (long)b * gx_bs, per, (hipStream_t)stream));
      }
    }
    dim3 gs(irr_cdiv((long)OH * OW, 256), C < 8 ? C : 8, B);
    if (accumulate)
      hipLaunchKernelGGL(resize_bwd_sparse_kernel<true>, gs, dim3(256), 0, (hipStream_t)stream, gout, gx, C, H, W, OH, OW, gout_bs, gx_bs, alpha);
    else
      hipLaunchKernelGGL(resize_bwd_sparse_kernel<false>, gs, dim3(256), 0, (hipStream_t)stream, gout, gx, C, H, W, OH, OW, gout_bs, gx_bs, alpha);
    IRR_LAUNCH_CHECK();
    return 0;
  }
  dim3 grid(irr_cdiv((long)H * W, 256), C < 8 ? C : 8, B);
  if (accumulate)
    hipLaunchKernelGGL((resize_bwd_kernel<false, true>), grid, dim3(256), 0, (hipStream_t)stream, gout, gx, C, H, W, OH, OW, gout_bs, gx_bs, alpha);
  else
    hipLaunchKernelGGL((resize_bwd_kernel<false, false>), grid, dim3(256), 0, (hipStream_t)stream, gout, gx, C, H, W, OH, OW, gout_bs, gx_bs, alpha);
  IRR_LAUNCH_CHECK();
  return 0;
}

extern "C" int irr_resize_bilinear_ac_bwd_f32(const float* gout, float* gx, int B, int C, int H, int W, int OH, int OW,
                                              long gout_bs, long gx_bs, float alpha, void* stream) {
  return resize_ac_bwd_impl(gout, gx, B, C, H, W, OH, OW, gout_bs, gx_bs, alpha, 0, stream);
}

// gx += alpha * resize^T(gout) (accumulate = 1; 0: irr_resize_bilinear_ac_bwd_f32): the gradients of ONE tensor resized to several sizes
// meet in one buffer -- the raw images at the five refinement levels (models/IRR_PWC.py:126-127): one zero fill and five sparse
// read-modify-write launches instead of five fills and four dense additions of maps that are 75-99.9 % zeros
extern "C" int irr_resize_bilinear_ac_bwd_acc_f32(const float* gout, float* gx, int B, int C, int H, int W, int OH, int OW,
                                                  long gout_bs, long gx_bs, float alpha, int accumulate, void* stream) {
  return resize_ac_bwd_impl(gout, gx, B, C, H, W, OH, OW, gout_bs, gx_bs, alpha, accumulate ? 1 : 0, stream);
}

extern "C" int irr_resize_bilinear_hp_fwd_f32(const float* x, float* out, int B, int C, int H, int W, int OH, int OW,
                                              long x_bs, long out_bs, float alpha, void* stream) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0 || !x || !out || B > 65535) return IRR_EINVAL;
  dim3 grid(irr_cdiv((long)OH * OW, 256), C < 8 ? C : 8, B);
  hipLaunchKernelGGL(resize_fwd_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, x, out, C, H, W, OH, OW, x_bs, out_bs,
                     alpha);
  IRR_LAUNCH_CHECK();
  return 0;
}

extern "C" int irr_resize_bilinear_hp_bwd_f32(const float* gout, float* gx, int B, int C, int H, int W, int OH, int OW,
                                              long gout_bs, long gx_bs, float alpha, void* stream) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0 || !gout || !gx || B > 65535) return IRR_EINVAL;
  dim3 grid(irr_cdiv((long)H * W, 256), C < 8 ? C : 8, B);
  hipLaunchKernelGGL((resize_bwd_kernel<true, false>), grid, dim3(256), 0, (hipStream_t)stream, gout, gx, C, H, W, OH, OW, gout_bs,
                     gx_bs, alpha);
  IRR_LAUNCH_CHECK();
  return 0;
}
