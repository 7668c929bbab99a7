// Flow warping with validity mask -- WarpingLayer.forward (models/pwc_modules.py:107-133).
//
// The reference runs two grid_sample calls (data + a freshly allocated ones tensor), a compare and a
// multiply; here one pass computes the four bilinear taps once per pixel, derives the mask from the
// in-bounds tap weights and reuses the taps for every channel.  Lanes run along x: flow/grid loads and
// the output stores are coalesced, the gathers hit neighbouring cache lines because flow is smooth.
//
// Bit-level contract with ATen's CPU grid_sampler (align_corners=True, zeros padding):
//   ix = (gx + 1) * ((W-1)/2) ; x0 = floor(ix) ; w = ix - x0 ; e = 1 - w ; (same for y: n, s)
//   nw = s*e, ne = s*w, sw = n*e, se = n*w ;   ones-sample = ((nw+ne)+sw)+se over in-bounds taps only
// and gx = linspace[x] + ((flow*2)/max(W_im-1,1))/div_flow with IEEE divisions (no reciprocal tricks).
// This file must be compiled WITHOUT fast-math / fp-contract so the mask test reproduces.
#include "common.h"

namespace {

struct Taps {
  int x0, y0;
  float nw, ne, sw, se;     // weights (already zeroed for out-of-bounds taps? no: raw weights)
  bool in_nw, in_ne, in_sw, in_se;
  float mask;
  float w, e, n, s;
};

__device__ __forceinline__ Taps make_taps(float fu, float fv, float gxb, float gyb, int H, int W, float den_w,
                                          float den_h, float div_flow, float mask_thr) {
  Taps t;
  // flow[:,0]*2/max(width_im-1,1)/div_flow   (models/pwc_modules.py:121-122)
  const float ox = __fdiv_rn(__fdiv_rn(__fmul_rn(fu, 2.f), den_w), div_flow);
  const float oy = __fdiv_rn(__fdiv_rn(__fmul_rn(fv, 2.f), den_h), div_flow);
  const float gx = __fadd_rn(gxb, ox);
  const float gy = __fadd_rn(gyb, oy);
  const float sx = __fdiv_rn((float)(W - 1), 2.f);
  const float sy = __fdiv_rn((float)(H - 1), 2.f);
  const float ix = __fmul_rn(__fadd_rn(gx, 1.f), sx);
  const float iy = __fmul_rn(__fadd_rn(gy, 1.f), sy);
  const float fx0 = floorf(ix), fy0 = floorf(iy);
  t.w = __fsub_rn(ix, fx0);
  t.e = __fsub_rn(1.f, t.w);
  t.n = __fsub_rn(iy, fy0);
  t.s = __fsub_rn(1.f, t.n);
  t.nw = __fmul_rn(t.s, t.e);
  t.ne = __fmul_rn(t.s, t.w);
  t.sw = __fmul_rn(t.n, t.e);
  t.se = __fmul_rn(t.n, t.w);
  // clamp before the int conversion so wild flows cannot overflow
  const float cx = fminf(fmaxf(fx0, -2.f), (float)W + 1.f);
  const float cy = fminf(fmaxf(fy0, -2.f), (float)H + 1.f);
  t.x0 = (int)cx;
  t.y0 = (int)cy;
  const bool xin0 = (t.x0 >= 0) && (t.x0 < W), xin1 = (t.x0 + 1 >= 0) && (t.x0 + 1 < W);
  const bool yin0 = (t.y0 >= 0) && (t.y0 < H), yin1 = (t.y0 + 1 >= 0) && (t.y0 + 1 < H);
  t.in_nw = xin0 && yin0;
  t.in_ne = xin1 && yin0;
  t.in_sw = xin0 && yin1;
  t.in_se = xin1 && yin1;
  float m = 0.f;                                    // sequential nw -> ne -> sw -> se, one rounding each
  m = __fadd_rn(m, t.in_nw ? t.nw : 0.f);
  m = __fadd_rn(m, t.in_ne ? t.ne : 0.f);
  m = __fadd_rn(m, t.in_sw ? t.sw : 0.f);
  m = __fadd_rn(m, t.in_se ? t.se : 0.f);
  // NaN flow -> comparison false -> masked out, like (mask >= thr).float()
  t.mask = (m >= mask_thr) ? 1.f : 0.f;
  return t;
}

__global__ __launch_bounds__(256) void warp_fwd_kernel(const float* __restrict__ x, const float* __restrict__ flow,
                                                      const float* __restrict__ gridx, const float* __restrict__ gridy,
                                                      float* __restrict__ out, int C, int H, int W, long x_bs,
                                                      long flow_bs, long out_bs, float den_w, float den_h,
                                                      float div_flow, float mask_thr, int cchunk, int xshift) {
  const long plane = (long)H * W;
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= plane) return;
  const int b = blockIdx.z;
  const int c_begin = blockIdx.y * cchunk, c_end = min(C, c_begin + cchunk);
  const int yy = (int)(p / W), xx = (int)(p - (long)yy * W);
  const float* fl = flow + (long)b * flow_bs;
  const Taps t = make_taps(fl[p], fl[plane + p], gridx[xx], gridy[yy], H, W, den_w, den_h, div_flow, mask_thr);
  const long o_nw = (long)t.y0 * W + t.x0;
  const int bx = (b + xshift) % (int)gridDim.z;             // swap_halves: sample b reads the OTHER half of x
  const float* xb = x + (long)bx * x_bs;
  float* ob = out + (long)b * out_bs + p;
  for (int c = c_begin; c < c_end; ++c) {
    const float* xc = xb + (long)c * plane;
    float v = 0.f;
    if (t.mask != 0.f) {
      const float a = t.in_nw ? xc[o_nw] : 0.f;
      const float bq = t.in_ne ? xc[o_nw + 1] : 0.f;
      const float cq = t.in_sw ? xc[o_nw + W] : 0.f;
      const float dq = t.in_se ? xc[o_nw + W + 1] : 0.f;
      v = a * t.nw + bq * t.ne + cq * t.sw + dq * t.se;
    }
    ob[(long)c * plane] = v;
  }
}

// Backward: gather for gflow, atomic scatter for gx.  The scatter is bound by the device-scope atomic rate (~33 G/s measured), so
// neighbouring pixels share their work.  A wave covers 16 columns x 4 rows of the image (a block 32 x 8); for a smooth flow the
// four bilinear targets of a pixel coincide with targets of its right / lower / lower-right neighbours, so a lane COLLECTS, for
// its nw target, the ne contribution of the lane to its left, the sw contribution of the lane above and the se contribution of
// the lane above-left (three lane shuffles of the output gradient per channel) and issues ONE atomic; only the last row / column
// of a wave and flow discontinuities issue more (~1.3 atomics per pixel and channel; a row-only version of this scheme issued 2,
// the plain scatter 4).  Measured at 96x112x64, 32 channels: 1.38 -> 1.23 ms for a smooth flow, 2.47 -> 1.82 ms for a rough one
// (tools/warp_bench.py): ~23 G atomics/s -- the device-scope atomic path itself is the limit now, not the lane work.
__global__ __launch_bounds__(256) void warp_bwd_kernel(const float* __restrict__ x, const float* __restrict__ flow,
                                                      const float* __restrict__ gridx, const float* __restrict__ gridy,
                                                      const float* __restrict__ gout, float* __restrict__ gx,
                                                      float* __restrict__ gflow, int C, int H, int W, long x_bs,
                                                      long flow_bs, long gout_bs, long gx_bs, long gflow_bs,
                                                      float den_w, float den_h, float div_flow, float mask_thr, int xshift) {
  const long plane = (long)H * W;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int c = lane & 15, r = lane >> 4;
  const int xx = blockIdx.x * 32 + (wv & 1) * 16 + c, yy = blockIdx.y * 8 + (wv >> 1) * 4 + r;
  const bool act = xx < W && yy < H;
  const int xc = act ? xx : 0, yc = act ? yy : 0;
  const long pp = (long)yc * W + xc;
  const int b = blockIdx.z;
  const float* fl = flow + (long)b * flow_bs;
  const Taps t = make_taps(fl[pp], fl[plane + pp], gridx[xc], gridy[yc], H, W, den_w, den_h, div_flow, mask_thr);
  const bool on = act && t.mask != 0.f;
  const long o_nw = (long)t.y0 * W + t.x0;
  // effective scatter weights (0 = nothing to add)
  const float wnw = (on && t.in_nw) ? t.nw : 0.f, wne = (on && t.in_ne) ? t.ne : 0.f;
  const float wsw = (on && t.in_sw) ? t.sw : 0.f, wse = (on && t.in_se) ? t.se : 0.f;
  // neighbours inside the wave: L = left, U = up, UL = up-left (sources I may collect from); R, D, DR (receivers of my own)
  const bool hasL = c > 0, hasU = r > 0, hasR = c < 15, hasD = r < 3;
  const int ia = act ? 1 : 0;
  const int Lx = __shfl_up(t.x0, 1, 64), Ly = __shfl_up(t.y0, 1, 64);
  const int Ux = __shfl_up(t.x0, 16, 64), Uy = __shfl_up(t.y0, 16, 64);
  const int ULx = __shfl_up(t.x0, 17, 64), ULy = __shfl_up(t.y0, 17, 64);
  const int Rx = __shfl_down(t.x0, 1, 64), Ry = __shfl_down(t.y0, 1, 64), Ra = __shfl_down(ia, 1, 64);
  const int Dx = __shfl_down(t.x0, 16, 64), Dy = __shfl_down(t.y0, 16, 64), Da = __shfl_down(ia, 16, 64);
  const int DRx = __shfl_down(t.x0, 17, 64), DRy = __shfl_down(t.y0, 17, 64), DRa = __shfl_down(ia, 17, 64);
  const float Lwne = __shfl_up(wne, 1, 64), Lwse = __shfl_up(wse, 1, 64);
  const float Uwsw = __shfl_up(wsw, 16, 64), ULwse = __shfl_up(wse, 17, 64);
  // what I collect (the sources' targets are inside the image by their own flags: zero weights otherwise)
  const bool mL = act && hasL && Lx + 1 == t.x0 && Ly == t.y0;
  const bool mU = act && hasU && Ux == t.x0 && Uy + 1 == t.y0;
  const bool mUL = act && hasL && hasU && ULx + 1 == t.x0 && ULy + 1 == t.y0;
  // the left lane's se goes to ITS lower-right lane (= the lane below me) when that one matches; otherwise to my sw slot
  const bool L_se_down = hasD && Da && Dx == Lx + 1 && Dy == Ly + 1;
  const float cL_nw = mL ? Lwne : 0.f, cU_nw = mU ? Uwsw : 0.f, cUL_nw = mUL ? ULwse : 0.f;
  const float cL_sw = (mL && !L_se_down) ? Lwse : 0.f;
  // what I hand over
  const bool toR = hasR && Ra && Rx == t.x0 + 1 && Ry == t.y0;
  const bool toD = hasD && Da && Dx == t.x0 && Dy == t.y0 + 1;
  const bool toDR = hasR && hasD && DRa && DRx == t.x0 + 1 && DRy == t.y0 + 1;
  const bool do_nw = wnw != 0.f || cL_nw != 0.f || cU_nw != 0.f || cUL_nw != 0.f;
  const bool do_ne = wne != 0.f && !toR;
  const bool do_sw = (wsw != 0.f && !toD) || cL_sw != 0.f;
  const float own_sw = toD ? 0.f : wsw;
  const bool do_se = wse != 0.f && !toDR && !toR;
  float gix = 0.f, giy = 0.f;
  const int bx = (b + xshift) % (int)gridDim.z;             // swap_halves: x (and its gradient) of the other batch half
  const float* xb = x + (long)bx * x_bs;
  const float* gb = gout + (long)b * gout_bs + pp;
  float* gxb = gx ? gx + (long)bx * gx_bs : nullptr;
  for (int ch = 0; ch < C; ++ch) {
    const float g = act ? gb[(long)ch * plane] : 0.f;
    if (gxb) {
      const float gL = __shfl_up(g, 1, 64), gU = __shfl_up(g, 16, 64), gUL = __shfl_up(g, 17, 64);
      float* gc = gxb + (long)ch * plane;
      if (do_nw) unsafeAtomicAdd(gc + o_nw, g * wnw + gL * cL_nw + gU * cU_nw + gUL * cUL_nw);
      if (do_ne) unsafeAtomicAdd(gc + o_nw + 1, g * wne);
      if (do_sw) unsafeAtomicAdd(gc + o_nw + W, g * own_sw + gL * cL_sw);
      if (do_se) unsafeAtomicAdd(gc + o_nw + W + 1, g * wse);
    }
    if (gflow && on) {
      const float* xcp = xb + (long)ch * plane;
      const float a = t.in_nw ? xcp[o_nw] : 0.f;
      const float bq = t.in_ne ? xcp[o_nw + 1] : 0.f;
      const float cq = t.in_sw ? xcp[o_nw + W] : 0.f;
      const float dq = t.in_se ? xcp[o_nw + W + 1] : 0.f;
      // d/d ix : -nw_val*s + ne_val*s - sw_val*n + se_val*n ;  d/d iy : -nw_val*e - ne_val*w + sw_val*e + se_val*w
      gix += g * ((bq - a) * t.s + (dq - cq) * t.n);
      giy += g * ((cq - a) * t.e + (dq - bq) * t.w);
    }
  }
  if (gflow && act) {
    // d ix / d gx = (W-1)/2 ; d gx / d flow_u = 2 / max(W_im-1,1) / div_flow
    float* gf = gflow + (long)b * gflow_bs;
    gf[pp] = gix * (0.5f * (float)(W - 1)) * (2.f / den_w / div_flow);
    gf[plane + pp] = giy * (0.5f * (float)(H - 1)) * (2.f / den_h / div_flow);
  }
}

}  // namespace

extern "C" int irr_warp_fwd_f32(const float* x, const float* flow, const float* gridx, const float* gridy, float* out,
                                int B, int C, int H, int W, long x_bs, long flow_bs, long out_bs, int height_im,
                                int width_im, float div_flow, float mask_thr, int swap_halves, void* stream) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || !x || !flow || !gridx || !gridy || !out || B > 65535) return IRR_EINVAL;
  if (swap_halves && (B & 1)) return IRR_EINVAL;
  const long plane = (long)H * W;
  // split channels over blockIdx.y only when the pixel grid alone cannot fill 256 CUs
  int cchunk = C;
  const long pix_blocks = (long)irr_cdiv(plane, 256) * B;
  if (pix_blocks < 1024 && C > 8) cchunk = 8;
  dim3 grid(irr_cdiv(plane, 256), irr_cdiv(C, cchunk), B);
  const float den_w = (float)(width_im - 1 > 1 ? width_im - 1 : 1), den_h = (float)(height_im - 1 > 1 ? height_im - 1 : 1);
  hipLaunchKernelGGL(warp_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, flow, gridx, gridy, out, C, H, W, x_bs,
                     flow_bs, out_bs, den_w, den_h, div_flow, mask_thr, cchunk, swap_halves ? B / 2 : 0);
  IRR_LAUNCH_CHECK();
  return 0;
}

extern "C" int irr_warp_bwd_f32(const float* x, const float* flow, const float* gridx, const float* gridy,
                                const float* gout, float* gx, float* gflow, int B, int C, int H, int W, long x_bs,
                                long flow_bs, long gout_bs, long gx_bs, long gflow_bs, int height_im, int width_im,
                                float div_flow, float mask_thr, int swap_halves, void* stream) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || !x || !flow || !gridx || !gridy || !gout || B > 65535) return IRR_EINVAL;
  if (swap_halves && (B & 1)) return IRR_EINVAL;
  if (!gx && !gflow) return 0;
  const long plane = (long)H * W;
  if (gx) {
    // gx is a scatter target: zero exactly the region this call owns (dense when gx_bs == C*plane)
    if (gx_bs == (long)C * plane) {
      IRR_HIP_TRY(hipMemsetAsync(gx, 0, sizeof(float) * (size_t)B * C * plane, (hipStream_t)stream));
    } else {
      for (int b = 0; b < B; ++b)
        IRR_HIP_TRY(hipMemsetAsync(gx + (long)b * gx_bs, 0, sizeof(float) * (size_t)C * plane, (hipStream_t)stream));
    }
  }
  dim3 grid(irr_cdiv(W, 32), irr_cdiv(H, 8), B);
  const float den_w = (float)(width_im - 1 > 1 ? width_im - 1 : 1), den_h = (float)(height_im - 1 > 1 ? height_im - 1 : 1);
  hipLaunchKernelGGL(warp_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, flow, gridx, gridy, gout, gx, gflow, C,
                     H, W, x_bs, flow_bs, gout_bs, gx_bs, gflow_bs, den_w, den_h, div_flow, mask_thr, swap_halves ? B / 2 : 0);
  IRR_LAUNCH_CHECK();
  return 0;
}
