// On-GPU training augmentation: affine resampling of images / occlusion maps / flow fields.
// One thread per output pixel; the four bilinear taps of neighbouring lanes coalesce in L2 (the affine map is
// within ~11 degrees of axis-aligned and zooms in by 1..1.5, so a wave's taps span 2-3 source rows).
// Built with -ffp-contract=off: the coordinate arithmetic follows the reference's fp32 op order exactly.
#include "common.h"

namespace {

struct Coord {          // normalised <-> pixel coordinates, reference augmentations.py:10-21
  float nx, ny, dx, dy;
  float W, H;
  __host__ Coord(int w, int h)
      : nx((float)(2.0 / (w - 1.0))), ny((float)(2.0 / (h - 1.0))), dx((float)(0.5 * (w - 1.0))),
        dy((float)(0.5 * (h - 1.0))), W((float)w), H((float)h) {}
  __device__ float normx(float x) const { return nx * x - 1.0f; }
  __device__ float normy(float y) const { return ny * y - 1.0f; }
  __device__ float denx(float x) const { return dx * (x + 1.0f); }
  __device__ float deny(float y) const { return dy * (y + 1.0f); }
};

struct Taps {
  int i00, i01, i10, i11;
  float w00, w01, w10, w11;
  bool valid;
};

// transform_coords (augmentations.py:415-440) + the index/weight part of Interp2 (utils/interpolation.py:82-128).
// inv = (b1, b2, b4, b5, a3, a6) of the inverted affine map.
__device__ __forceinline__ Taps make_taps(const Coord& c, const float* __restrict__ inv, int x, int y, int W, int H) {
  const float xh = c.normx((float)x) - inv[4];
  const float yh = c.normy((float)y) - inv[5];
  const float xq = c.denx(inv[0] * xh + inv[1] * yh);
  const float yq = c.deny(inv[2] * xh + inv[3] * yh);
  Taps t;
  t.valid = !((xq < 0.f) | (xq >= c.W) | (yq < 0.f) | (yq >= c.H));
  // NaN / inf coordinates are invalid by the test above only when comparisons are true; clamp through float first
  float fx = floorf(xq), fy = floorf(yq);
  fx = fminf(fmaxf(fx, 0.f), c.W - 1.f);
  fy = fminf(fmaxf(fy, 0.f), c.H - 1.f);
  const int x0 = (int)fx, y0 = (int)fy;
  const int x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1);
  const float lx = xq - fx, ly = yq - fy;
  t.w00 = (1.0f - ly) * (1.0f - lx);
  t.w01 = (1.0f - ly) * lx;
  t.w10 = ly * (1.0f - lx);
  t.w11 = ly * lx;
  t.i00 = y0 * W + x0; t.i01 = y0 * W + x1; t.i10 = y1 * W + x0; t.i11 = y1 * W + x1;
  return t;
}

__device__ __forceinline__ float blend(const Taps& t, float v00, float v01, float v10, float v11) {
  return ((v00 * t.w00 + v01 * t.w01) + v10 * t.w10) + v11 * t.w11;
}

// ---- images / occlusion maps ------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
affine_warp_kernel(const float* __restrict__ src, float* __restrict__ dst, const float* __restrict__ inv,
                   const float* __restrict__ noise, float noise_std, int C, int H, int W, int OH, int OW,
                   int y0, int x0, long src_bs, long dst_bs, Coord c) {
  const int b = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= OH * OW) return;
  const int oy = p / OW, ox = p - oy * OW;
  const Taps t = make_taps(c, inv + b * 6, ox + x0, oy + y0, W, H);
  const float* s = src + b * src_bs;
  float* d = dst + b * dst_bs + p;
  const float* nz = noise ? noise + b * (long)C * OH * OW + p : nullptr;
  const long hw = (long)H * W, ohw = (long)OH * OW;
  for (int ch = 0; ch < C; ++ch) {
    float v = 0.f;
    if (t.valid) v = blend(t, s[t.i00], s[t.i01], s[t.i10], s[t.i11]);
    if (nz) {
      v = v + noise_std * nz[ch * ohw];
      v = fminf(fmaxf(v, 0.f), 1.f);
    }
    d[ch * ohw] = v;
    s += hw;
  }
}

// ---- flow (+ its occlusion map) -----------------------------------------------------------------------
// transform_flow (augmentations.py:525-548): the field that is resampled is
//   new(xs,ys) = T_b(xs + u, ys + v) - T_a(xs, ys),   T = inverse_transform_coords (augmentations.py:391-413)
// evaluated at the four taps; then check_out_of_bound (augmentations.py:550-563) folds "flow leaves the (cropped)
// frame" into the resampled occlusion map.
__device__ __forceinline__ void new_flow_at(const Coord& c, const float* __restrict__ ta, const float* __restrict__ tb,
                                            const float* __restrict__ fu, const float* __restrict__ fv, int idx, int W,
                                            float& nu, float& nv) {
  const int ys = idx / W, xs = idx - ys * W;
  const float xa = c.normx((float)xs), ya = c.normy((float)ys);
  const float ax = c.denx((ta[0] * xa + ta[1] * ya) + ta[2]);
  const float ay = c.deny((ta[3] * xa + ta[4] * ya) + ta[5]);
  const float xb = c.normx((float)xs + fu[idx]), yb = c.normy((float)ys + fv[idx]);
  const float bx = c.denx((tb[0] * xb + tb[1] * yb) + tb[2]);
  const float by = c.deny((tb[3] * xb + tb[4] * yb) + tb[5]);
  nu = bx - ax;
  nv = by - ay;
}

__global__ void __launch_bounds__(256)
affine_flow_occ_kernel(const float* __restrict__ flow, const float* __restrict__ occ, float* __restrict__ flow_out,
                       float* __restrict__ occ_out, const float* __restrict__ inv_a, const float* __restrict__ theta_a,
                       const float* __restrict__ theta_b, int H, int W, int OH, int OW, int y0, int x0,
                       long flow_bs, long occ_bs, long fo_bs, long oo_bs, Coord c) {
  const int b = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= OH * OW) return;
  const int oy = p / OW, ox = p - oy * OW;
  const Taps t = make_taps(c, inv_a + b * 6, ox + x0, oy + y0, W, H);
  const float* fu = flow + b * flow_bs;
  const float* fv = fu + (long)H * W;
  const float* ta = theta_a + b * 6;
  const float* tb = theta_b + b * 6;
  float u = 0.f, v = 0.f, o = 0.f;
  if (t.valid) {
    float u00, v00, u01, v01, u10, v10, u11, v11;
    new_flow_at(c, ta, tb, fu, fv, t.i00, W, u00, v00);
    new_flow_at(c, ta, tb, fu, fv, t.i01, W, u01, v01);
    new_flow_at(c, ta, tb, fu, fv, t.i10, W, u10, v10);
    new_flow_at(c, ta, tb, fu, fv, t.i11, W, u11, v11);
    u = blend(t, u00, u01, u10, u11);
    v = blend(t, v00, v01, v10, v11);
    if (occ) {
      const float* oc = occ + b * occ_bs;
      o = blend(t, oc[t.i00], oc[t.i01], oc[t.i10], oc[t.i11]);
    }
  }
  float* fo = flow_out + b * fo_bs + p;
  fo[0] = u;
  fo[(long)OH * OW] = v;
  if (occ_out) {
    const float xx = (float)ox + u, yy = (float)oy + v;     // coordinates in the CROPPED frame
    const float oob = ((xx < 0.f) | (yy < 0.f) | (xx >= (float)OW) | (yy >= (float)OH)) ? 1.f : 0.f;
    occ_out[b * oo_bs + p] = fminf(fmaxf(oob + o, 0.f), 1.f);
  }
}

}  // namespace

extern "C" int irr_affine_warp_f32(const float* src, float* dst, const float* inv, const float* noise, float noise_std,
                                   int B, int C, int H, int W, int OH, int OW, int y0, int x0, long src_bs, long dst_bs,
                                   void* stream) {
  if (B <= 0 || C <= 0 || H < 2 || W < 2 || OH <= 0 || OW <= 0 || y0 < 0 || x0 < 0 || y0 + OH > H || x0 + OW > W ||
      B > 65535 || (long)H * W >= (1L << 31))
    return IRR_EINVAL;
  dim3 grid(irr_cdiv((long)OH * OW, 256), B);
  affine_warp_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(src, dst, inv, noise, noise_std, C, H, W, OH, OW, y0, x0, src_bs,
                                                            dst_bs, Coord(W, H));
  IRR_LAUNCH_CHECK();
  return 0;
}

extern "C" int irr_affine_flow_occ_f32(const float* flow, const float* occ, float* flow_out, float* occ_out,
                                       const float* inv_a, const float* theta_a, const float* theta_b, int B, int H, int W,
                                       int OH, int OW, int y0, int x0, long flow_bs, long occ_bs, long flow_out_bs,
                                       long occ_out_bs, void* stream) {
  if (B <= 0 || H < 2 || W < 2 || OH <= 0 || OW <= 0 || y0 < 0 || x0 < 0 || y0 + OH > H || x0 + OW > W || B > 65535 ||
      (long)H * W >= (1L << 31) || ((occ == nullptr) != (occ_out == nullptr)))
    return IRR_EINVAL;
  dim3 grid(irr_cdiv((long)OH * OW, 256), B);
  affine_flow_occ_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(flow, occ, flow_out, occ_out, inv_a, theta_a, theta_b, H, W, OH,
                                                                OW, y0, x0, flow_bs, occ_bs, flow_out_bs, occ_out_bs,
                                                                Coord(W, H));
  IRR_LAUNCH_CHECK();
  return 0;
}
