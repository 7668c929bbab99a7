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

// (owner-computes backward, see below)
#define GMAX 16         // largest per-sample margin the gather handles
#define GK 4            // contributors per round of the channel loop
#define GKMAX 24        // list slots per owned pixel (a pixel has 4 contributors on average)
#define GMS 32          // margin slots per sample (the sample's margin is their maximum)

__device__ __forceinline__ int sample_margin(const int* __restrict__ margin, int b) {
  int m = 0;
#pragma unroll
  for (int k = 0; k < GMS; ++k) m = max(m, margin[b * GMS + k]);
  return m;
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
                                                      float den_w, float den_h, float div_flow, float mask_thr, int xshift,
                                                      const int* __restrict__ gx_only_if, const int* __restrict__ gx_flags) {
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
  // gx_only_if (owner-computes route): the scatter only runs for the samples whose targets left the gather window
  float* gxb = (gx && (!gx_only_if || gx_flags[b] != 0 || sample_margin(gx_only_if, b) > GMAX)) ? gx + (long)bx * gx_bs : nullptr;
  if (!gxb && !gflow) return;
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


// ---- owner-computes gradient w.r.t. x (round 4) ---------------------------------------------------------------------------
// The scatter above is bound by the DEVICE-SCOPE atomic rate (~23 G/s: 1.2 ms at 96x112x64x32 for 90 us worth of HBM traffic).
// Here every pixel s of gx is OWNED by one thread, which gathers what the output pixels scatter into it:
//   1. warp_margin_kernel: per sample the largest distance M_b (in pixels, per axis) between an output pixel and any of its
//      bilinear targets;
//   2. warp_gather_kernel: a block owns an 8 x 32 tile of gx.  BINNING, once per block (not per channel): the threads walk the
//      output pixels within M_b of the tile, compute their taps (same arithmetic as everywhere else) and append, for each of the
//      four targets that lies in the tile, (offset of the output pixel, weight) to that target's list in LDS (an LDS integer
//      atomic hands out the slot: <= GKMAX entries per pixel).  GATHER: a thread takes its pixel's list four entries at a time
//      and runs the channel loop as a 4-tap gather like the forward pass -- plain loads and stores, no atomics on data, no zero
//      fill; lists longer than four (compressive flows) take further rounds (read-modify-write of the OWN pixel).
//      A sample with M_b > GMAX is left alone; a pixel with more than GKMAX contributors flags its SAMPLE; then
//   3. warp_zero_flagged_kernel + warp_bwd_kernel (above): the flagged samples are zero-filled and take the device-scope atomic
//      scatter.
// All decisions are taken on the device: nothing synchronises.  (History of this round, 96x112x64x32, smooth flow: device-scope
// scatter 1.23 ms; LDS float atomics per owned tile 0.59 ms -- ds_add_f32 retires a few lanes per cycle; per-pixel window scan
// 0.05 ms, but (2 M + 1)^2 tests per pixel made it 0.76 ms on the noisy flows of a freshly initialised network.)
__global__ __launch_bounds__(256) void warp_margin_kernel(const float* __restrict__ flow, const float* __restrict__ gridx,
                                                         const float* __restrict__ gridy, int* __restrict__ margin, int H, int W,
                                                         long flow_bs, float den_w, float den_h, float div_flow, float mask_thr) {
  __shared__ int red[4];
  const long plane = (long)H * W;
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.z;
  int m = 0;
  if (p < plane) {
    const int yy = (int)(p / W), xx = (int)(p - (long)yy * W);
    const float* fl = flow + (long)b * flow_bs;
    const Taps t = make_taps(fl[p], fl[plane + p], gridx[xx], gridy[yy], H, W, den_w, den_h, div_flow, mask_thr);
    if (t.mask != 0.f) {
      if (t.in_nw || t.in_sw) m = max(m, abs(t.x0 - xx));
      if (t.in_ne || t.in_se) m = max(m, abs(t.x0 + 1 - xx));
      if (t.in_nw || t.in_ne) m = max(m, abs(t.y0 - yy));
      if (t.in_sw || t.in_se) m = max(m, abs(t.y0 + 1 - yy));
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_down(m, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = max(max(red[0], red[1]), max(red[2], red[3]));
    // (hundreds of blocks per sample: GMS slots per sample spread the same-address atomics -- 53 -> 17 us at 192x224x64)
    if (m > 0) atomicMax(margin + b * GMS + (blockIdx.x % GMS), m);
  }
}

// flags[b] (the second half of the workspace): 1 = some pixel of sample b overflowed its list -> the atomic route redoes the sample
__global__ __launch_bounds__(256) void warp_gather_kernel(const float* __restrict__ flow, const float* __restrict__ gridx,
                                                         const float* __restrict__ gridy, const float* __restrict__ gout,
                                                         float* __restrict__ gx, const int* __restrict__ margin,
                                                         int* __restrict__ flags, int C, int H, int W, long flow_bs, long gout_bs,
                                                         long gx_bs, float den_w, float den_h, float div_flow, float mask_thr,
                                                         int xshift) {
  __shared__ int cnt[256];
  __shared__ int loff[GKMAX][256];
  __shared__ float lwt[GKMAX][256];
  const long plane = (long)H * W;
  const int c = threadIdx.x & 31, r = threadIdx.x >> 5;
  const int nb = gridDim.z;
  const int bs = blockIdx.z;                                           // the owned tile of gx belongs to sample bs
  const int b = (bs - xshift + nb) % nb;                               // the output sample that scatters into sample bs
  const int M = sample_margin(margin, b);                              // (uniform)
  if (M > GMAX) return;                                                // the atomic route handles this sample
  const int X0 = blockIdx.x * 32, Y0 = blockIdx.y * 8;
  const int ox = X0 + c, oy = Y0 + r;
  const bool own = ox < W && oy < H;
  cnt[threadIdx.x] = 0;
  __syncthreads();
  // ---- binning: output pixels [Y0 - M, Y0 + 7 + M] x [X0 - M, X0 + 31 + M] ----
  const int RW = 32 + 2 * M, RH = 8 + 2 * M;
  const float* fl = flow + (long)b * flow_bs;
  for (int i = threadIdx.x; i < RW * RH; i += 256) {
    const int ry = i / RW, rx = i - ry * RW;
    const int yy = Y0 - M + ry, xx = X0 - M + rx;
    if (xx < 0 || xx >= W || yy < 0 || yy >= H) continue;
    const long pp = (long)yy * W + xx;
    const Taps t = make_taps(fl[pp], fl[plane + pp], gridx[xx], gridy[yy], H, W, den_w, den_h, div_flow, mask_thr);
    if (t.mask == 0.f) continue;
    const int lx = t.x0 - X0, ly = t.y0 - Y0;                          // nw target relative to the owned tile
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int tx = lx + (k & 1), ty = ly + (k >> 1);
      const float w = k == 0 ? (t.in_nw ? t.nw : 0.f) : k == 1 ? (t.in_ne ? t.ne : 0.f) : k == 2 ? (t.in_sw ? t.sw : 0.f) : (t.in_se ? t.se : 0.f);
      if (w == 0.f || (unsigned)tx >= 32u || (unsigned)ty >= 8u) continue;
      const int ti = ty * 32 + tx;
      const int slot = atomicAdd(&cnt[ti], 1);
      if (slot < GKMAX) {
        loff[slot][ti] = (int)pp;
        lwt[slot][ti] = w;
      }
    }
  }
  __syncthreads();
  const int n_all = cnt[threadIdx.x];
  if (__syncthreads_or(n_all > GKMAX)) {                               // (block-uniform) the atomic route redoes this sample
    if (threadIdx.x == 0) atomicOr(flags + b, 1);
    return;
  }
  if (!own) return;
  const int n = n_all;
  const float* gb = gout + (long)b * gout_bs;
  const int self = oy * W + ox;
  float* gdst = gx + (long)bs * gx_bs + self;
  // ---- gather: GK list entries per round ----
  for (int k0 = 0; k0 == 0 || k0 < n; k0 += GK) {
    int off[GK];
    float wt[GK];
#pragma unroll
    for (int k = 0; k < GK; ++k) {
      const bool ok = k0 + k < n;
      off[k] = ok ? loff[k0 + k][threadIdx.x] : self;
      wt[k] = ok ? lwt[k0 + k][threadIdx.x] : 0.f;
    }
    if (k0 == 0) {
#pragma unroll 4
      for (int ch = 0; ch < C; ++ch) {
        const float* gc = gb + (long)ch * plane;
        gdst[(long)ch * plane] = gc[off[0]] * wt[0] + gc[off[1]] * wt[1] + gc[off[2]] * wt[2] + gc[off[3]] * wt[3];
      }
    } else {
#pragma unroll 4
      for (int ch = 0; ch < C; ++ch) {
        const float* gc = gb + (long)ch * plane;
        gdst[(long)ch * plane] += gc[off[0]] * wt[0] + gc[off[1]] * wt[1] + gc[off[2]] * wt[2] + gc[off[3]] * wt[3];
      }
    }
  }
}

// zero fill of the samples that take the atomic route (margin beyond the gather window, or a list overflow)
__global__ __launch_bounds__(256) void warp_zero_flagged_kernel(float* __restrict__ gx, const int* __restrict__ margin,
                                                               const int* __restrict__ flags, long n_per_sample, long gx_bs,
                                                               int xshift) {
  const int nb = gridDim.z, bs = blockIdx.z;
  const int b = (bs - xshift + nb) % nb;
  if (sample_margin(margin, b) <= GMAX && flags[b] == 0) return;
  float* g = gx + (long)bs * gx_bs;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n_per_sample; i += (long)gridDim.x * blockDim.x) g[i] = 0.f;
}

// gradient w.r.t. the flow alone (owner-computes route: the scatter kernel above spends its time in the 16 x 4 lane layout its
// atomic merging needs -- 4 cache lines per load instruction).  Lanes run along x: gout and the four taps of x are read with
// (nearly) coalesced loads, four channels in flight per thread.
__global__ __launch_bounds__(256) void warp_gflow_kernel(const float* __restrict__ x, const float* __restrict__ flow,
                                                        const float* __restrict__ gridx, const float* __restrict__ gridy,
                                                        const float* __restrict__ gout, float* __restrict__ gflow, int C,
                                                        int H, int W, long x_bs, long flow_bs, long gout_bs, long gflow_bs,
                                                        float den_w, float den_h, float div_flow, float mask_thr, int xshift) {
  const long plane = (long)H * W;
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= plane) return;
  const int b = blockIdx.z;
  const int yy = (int)(p / W), xx = (int)(p - (long)yy * W);
  const float* fl = flow + (long)b * flow_bs;
  const Taps t = make_taps(fl[p], fl[plane + p], gridx[xx], gridy[yy], H, W, den_w, den_h, div_flow, mask_thr);
  float* gf = gflow + (long)b * gflow_bs;
  float gix = 0.f, giy = 0.f;
  if (t.mask != 0.f) {
    const int bx = (b + xshift) % (int)gridDim.z;
    // clamped tap offsets: an out-of-image tap reads a valid address and is multiplied by 0
    const long o_nw = t.in_nw ? (long)t.y0 * W + t.x0 : 0, o_ne = t.in_ne ? (long)t.y0 * W + t.x0 + 1 : 0;
    const long o_sw = t.in_sw ? (long)(t.y0 + 1) * W + t.x0 : 0, o_se = t.in_se ? (long)(t.y0 + 1) * W + t.x0 + 1 : 0;
    const float m_nw = t.in_nw ? 1.f : 0.f, m_ne = t.in_ne ? 1.f : 0.f, m_sw = t.in_sw ? 1.f : 0.f, m_se = t.in_se ? 1.f : 0.f;
    const float* xb = x + (long)bx * x_bs;
    const float* gb = gout + (long)b * gout_bs + p;
#pragma unroll 4
    for (int ch = 0; ch < C; ++ch) {
      const float* xcp = xb + (long)ch * plane;
      const float g = gb[(long)ch * plane];
      const float a = xcp[o_nw] * m_nw, bq = xcp[o_ne] * m_ne, cq = xcp[o_sw] * m_sw, dq = xcp[o_se] * m_se;
      gix += g * ((bq - a) * t.s + (dq - cq) * t.n);
      giy += g * ((cq - a) * t.e + (dq - bq) * t.w);
    }
  }
  gf[p] = gix * (0.5f * (float)(W - 1)) * (2.f / den_w / div_flow);
  gf[plane + p] = giy * (0.5f * (float)(H - 1)) * (2.f / den_h / div_flow);
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
      IRR_HIP_TRY(irr_zero_async(gx, sizeof(float) * (size_t)B * C * plane, (hipStream_t)stream));
    } else {
      for (int b = 0; b < B; ++b)
        IRR_HIP_TRY(irr_zero_async(gx + (long)b * gx_bs, sizeof(float) * (size_t)C * plane, (hipStream_t)stream));
    }
  }
  dim3 grid(irr_cdiv(W, 32), irr_cdiv(H, 8), B);
  const float den_w = (float)(width_im - 1 > 1 ? width_im - 1 : 1), den_h = (float)(height_im - 1 > 1 ? height_im - 1 : 1);
  hipLaunchKernelGGL(warp_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, flow, gridx, gridy, gout, gx, gflow, C,
                     H, W, x_bs, flow_bs, gout_bs, gx_bs, gflow_bs, den_w, den_h, div_flow, mask_thr, swap_halves ? B / 2 : 0,
                     (const int*)nullptr, (const int*)nullptr);
  IRR_LAUNCH_CHECK();
  return 0;
}

// workspace (ints) of irr_warp_bwd_gather_f32: one margin and one overflow flag per sample
extern "C" long irr_warp_bwd_ws_elems(int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return IRR_EINVAL;
  return (long)B * 32 + ((B + 3) / 4 * 4);                   // GMS margin slots per sample + one overflow flag per sample
}

// Same contract as irr_warp_bwd_f32; the gradient w.r.t. x is gathered per owned pixel (no atomics, no zero fill) for every
// sample whose bilinear targets stay within GMAX = 16 pixels of their output pixel (per axis) and whose pixels have at most 24
// contributors each, and falls back to the atomic scatter per SAMPLE otherwise (decided on the device).  ws: irr_warp_bwd_ws_elems(B, H, W) ints, any contents.
extern "C" int irr_warp_bwd_gather_f32(const float* x, const float* flow, const float* gridx, const float* gridy,
                                       const float* gout, float* gx, float* gflow, int B, int C, int H, int W, long x_bs,
                                       long flow_bs, long gout_bs, long gx_bs, long gflow_bs, int height_im, int width_im,
                                       float div_flow, float mask_thr, int swap_halves, int* ws, long ws_elems, void* stream) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || !x || !flow || !gridx || !gridy || !gout || B > 65535) return IRR_EINVAL;
  if (swap_halves && (B & 1)) return IRR_EINVAL;
  if (!gx && !gflow) return 0;
  const float den_w = (float)(width_im - 1 > 1 ? width_im - 1 : 1), den_h = (float)(height_im - 1 > 1 ? height_im - 1 : 1);
  dim3 grid(irr_cdiv(W, 32), irr_cdiv(H, 8), B);
  const long plane = (long)H * W;
  if (gx) {
    if (!ws || ws_elems < irr_warp_bwd_ws_elems(B, H, W)) return IRR_EINVAL;
    const int BP = (B + 3) / 4 * 4;
    int* flags = ws + (long)B * GMS;
    IRR_HIP_TRY(irr_zero_async(ws, sizeof(int) * ((size_t)B * GMS + BP), (hipStream_t)stream));
    hipLaunchKernelGGL(warp_margin_kernel, dim3(irr_cdiv(plane, 256), 1, B), dim3(256), 0, (hipStream_t)stream, flow, gridx, gridy, ws,
                       H, W, flow_bs, den_w, den_h, div_flow, mask_thr);
    IRR_LAUNCH_CHECK();
    hipLaunchKernelGGL(warp_gather_kernel, grid, dim3(256), 0, (hipStream_t)stream, flow, gridx, gridy, gout, gx,
                       (const int*)ws, flags, C, H, W, flow_bs, gout_bs, gx_bs, den_w, den_h, div_flow, mask_thr,
                       swap_halves ? B / 2 : 0);
    IRR_LAUNCH_CHECK();
    // samples whose targets left the gather window or overflowed a list: zero fill + the device-scope atomic scatter (both exit at
    // once for all the other samples)
    const long nps = (long)C * plane;
    int zb = irr_cdiv(nps, 256 * 8);
    if (zb > 1024) zb = 1024;
    hipLaunchKernelGGL(warp_zero_flagged_kernel, dim3(zb, 1, B), dim3(256), 0, (hipStream_t)stream, gx, (const int*)ws,
                       (const int*)flags, nps, gx_bs, swap_halves ? B / 2 : 0);
    IRR_LAUNCH_CHECK();
    hipLaunchKernelGGL(warp_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, flow, gridx, gridy, gout, gx, (float*)nullptr, C,
                       H, W, x_bs, flow_bs, gout_bs, gx_bs, gflow_bs, den_w, den_h, div_flow, mask_thr, swap_halves ? B / 2 : 0,
                       (const int*)ws, (const int*)flags);
    IRR_LAUNCH_CHECK();
  }
  if (gflow) {
    hipLaunchKernelGGL(warp_gflow_kernel, dim3(irr_cdiv(plane, 256), 1, B), dim3(256), 0, (hipStream_t)stream, x, flow, gridx, gridy,
                       gout, gflow, C, H, W, x_bs, flow_bs, gout_bs, gflow_bs, den_w, den_h, div_flow, mask_thr,
                       swap_halves ? B / 2 : 0);
    IRR_LAUNCH_CHECK();
  }
  return 0;
}
