// Weight gradient of the conv() blocks on fp32 MFMA:
//   dW[co][ci][tap] += sum_p gy[co][p] * x[ci][p + off(tap)]        (p = linear (b,oy,ox) output pixel)
//
// Here the reduction (K) dimension is the PIXEL axis, but MFMA fragments spread lanes over the
// non-reduced axes (32 rows x 2 k-values), while NCHW memory is contiguous along pixels.  So the operands
// are staged through LDS: global loads are coalesced along pixels (a lane = a pixel, 128 B runs per
// channel row), written as [channel][pixel] with an odd row pitch (33), and read back with lanes along the
// channel axis (stride 33 words -> conflict-free ds_read_b32).  One 32-pixel stage feeds 16 MFMA k-steps.
//
// Work split: a block owns MTB co-tiles x NTB ci-tiles (one wave per (co-tile, ci-tile) pair, all 9 taps:
// 9 accumulator tiles = 144 VGPRs) and a contiguous slice of the pixel range (split-K); partial results are
// added to dW with fp32 atomics (dW is shared by every pyramid level and both flow directions anyway).
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int KP = 32;          // pixels per LDS stage
constexpr int PITCH = KP + 1;

struct WgArgs {
  const float* x;
  const float* gy;
  float* gw;
  int B, Cin, H, W, Cout, OH, OW;
  int stride, dil, pad;
  long x_bs, gy_bs;
  int chunks_per_block;
};

template <int MTB, int NTB, int KS>
__global__ __launch_bounds__(MTB* NTB * 64, 2) void conv_wgrad_kernel(const WgArgs a) {
  constexpr int KK = KS * KS;
  constexpr int NW = MTB * NTB;
  constexpr int NT = NW * 64;
  constexpr int AROWS = MTB * 32, BROWS = NTB * 32;
  __shared__ float As[AROWS][PITCH];
  __shared__ float Bs[KK][BROWS][PITCH];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / NTB, wn = wave - wm * NTB;
  const int j = lane & 31, half = lane >> 5;
  const int co0 = blockIdx.z * AROWS, ci0 = blockIdx.y * BROWS;
  const long ohw = (long)a.OH * a.OW;
  const long hw = (long)a.H * a.W;
  const long total = (long)a.B * ohw;
  const long nchunks = (total + KP - 1) / KP;
  const long c_begin = (long)blockIdx.x * a.chunks_per_block;
  const long c_end = min(nchunks, c_begin + a.chunks_per_block);

  f32x16 acc[KK];
#pragma unroll
  for (int t = 0; t < KK; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // staging role of this thread: pixel column `px` of the stage, rows rg, rg+RG, ...
  const int px = tid & (KP - 1);
  const int rg = tid / KP;
  constexpr int RG = NT / KP;

  for (long c = c_begin; c < c_end; ++c) {
    const long p = c * KP + px;
    const bool pv = p < total;
    const long pp = pv ? p : total - 1;
    const int b = (int)(pp / ohw);
    const int r = (int)(pp - (long)b * ohw);
    const int oy = r / a.OW, ox = r - oy * a.OW;
    __syncthreads();                         // previous stage fully consumed
    {
      const float* g = a.gy + (long)b * a.gy_bs + r;
#pragma unroll 4
      for (int row = rg; row < AROWS; row += RG) {
        const int co = co0 + row;
        As[row][px] = (pv && co < a.Cout) ? g[(long)co * ohw] : 0.f;
      }
    }
#pragma unroll
    for (int t = 0; t < KK; ++t) {
      const int ty = t / KS, tx = t - ty * KS;
      const int iy = oy * a.stride - a.pad + ty * a.dil;
      const int ix = ox * a.stride - a.pad + tx * a.dil;
      const bool ok = pv && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      const float* xs = a.x + (long)b * a.x_bs + (long)(ok ? iy : 0) * a.W + (ok ? ix : 0);
#pragma unroll 4
      for (int row = rg; row < BROWS; row += RG) {
        const int ci = ci0 + row;
        Bs[t][row][px] = (ok && ci < a.Cin) ? xs[(long)ci * hw] : 0.f;
      }
    }
    __syncthreads();
    const float* arow = &As[wm * 32 + j][half];
#pragma unroll 4
    for (int k = 0; k < KP / 2; ++k) {
      const float av = arow[2 * k];
#pragma unroll
      for (int t = 0; t < KK; ++t) {
        const float bv = Bs[t][wn * 32 + j][2 * k + half];
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
      }
    }
  }

  // D[i][jj]: i = co_local = (r&3) + 8*(r>>2) + 4*half, jj = ci_local = lane&31
  const int ci = ci0 + wn * 32 + j;
  if (ci < a.Cin) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (co >= a.Cout) continue;
      float* dst = a.gw + ((long)co * a.Cin + ci) * KK;
#pragma unroll
      for (int t = 0; t < KK; ++t) unsafeAtomicAdd(dst + t, acc[t][r]);
    }
  }
}

template <int MTB, int NTB, int KS>
int launch(WgArgs a, hipStream_t st) {
  const long total = (long)a.B * a.OH * a.OW;
  const long nchunks = (total + KP - 1) / KP;
  const int gy_ = irr_cdiv(a.Cin, NTB * 32), gz_ = irr_cdiv(a.Cout, MTB * 32);
  // aim for ~4 blocks per CU overall, at least 4 stages per block
  long want = (1024 + (long)gy_ * gz_ - 1) / ((long)gy_ * gz_);
  if (want < 1) want = 1;
  long cpb = (nchunks + want - 1) / want;
  if (cpb < 4) cpb = 4;
  a.chunks_per_block = (int)cpb;
  dim3 grid(irr_cdiv(nchunks, cpb), gy_, gz_);
  hipLaunchKernelGGL((conv_wgrad_kernel<MTB, NTB, KS>), grid, dim3(MTB * NTB * 64), 0, st, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

template <int KS>
int dispatch(const WgArgs& a, hipStream_t st) {
  const int cot = (a.Cout + 31) / 32;
  if (cot == 1) return launch<1, 4, KS>(a, st);
  if (cot == 2) return launch<2, 2, KS>(a, st);
  if (cot == 3) return launch<3, 1, KS>(a, st);
  return launch<4, 1, KS>(a, st);
}

// gpre = gy * lrelu'(y) ; gbias[c] += sum_p gpre
__global__ __launch_bounds__(256) void lrelu_bwd_bias_kernel(const float* __restrict__ gy, const float* __restrict__ y,
                                                            float* __restrict__ gpre, float* __restrict__ gbias,
                                                            int C, long HW, long gy_bs, long y_bs, long gpre_bs,
                                                            int lrelu, int chunk) {
  const int c = blockIdx.y, b = blockIdx.z;
  const long p0 = (long)blockIdx.x * chunk;
  const long p1 = min(HW, p0 + chunk);
  const float* g = gy + (long)b * gy_bs + (long)c * HW;
  const float* yy = y ? y + (long)b * y_bs + (long)c * HW : nullptr;
  float* o = gpre ? gpre + (long)b * gpre_bs + (long)c * HW : nullptr;
  float s = 0.f;
  for (long p = p0 + threadIdx.x; p < p1; p += 256) {
    float v = g[p];
    if (lrelu) v *= irr_lrelu_grad(yy[p]);
    if (o) o[p] = v;
    s += v;
  }
  if (!gbias) return;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) unsafeAtomicAdd(gbias + c, red[0] + red[1] + red[2] + red[3]);
}

// gather-form data gradient for strided convs (pyramid stride-2 layers; < 1 % of the FLOPs)
__global__ __launch_bounds__(256) void dgrad_strided_kernel(const float* __restrict__ gy, const float* __restrict__ w,
                                                           float* __restrict__ gx, int Cin, int H, int W, int Cout,
                                                           int OH, int OW, int KS, int stride, int dil, int pad,
                                                           long gy_bs, long gx_bs) {
  const long hw = (long)H * W;
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= hw) return;
  const int ci = blockIdx.y, b = blockIdx.z;
  const int iy = (int)(p / W), ix = (int)(p - (long)iy * W);
  const long ohw = (long)OH * OW;
  const float* g = gy + (long)b * gy_bs;
  float s = 0.f;
  for (int ty = 0; ty < KS; ++ty) {
    const int ny = iy + pad - ty * dil;
    if (ny < 0 || ny % stride) continue;
    const int oy = ny / stride;
    if (oy >= OH) continue;
    for (int tx = 0; tx < KS; ++tx) {
      const int nx = ix + pad - tx * dil;
      if (nx < 0 || nx % stride) continue;
      const int ox = nx / stride;
      if (ox >= OW) continue;
      const float* gp = g + (long)oy * OW + ox;
      const float* wp = w + (long)ci * KS * KS + ty * KS + tx;
      for (int co = 0; co < Cout; ++co) s = fmaf(gp[(long)co * ohw], wp[(long)co * Cin * KS * KS], s);
    }
  }
  gx[(long)b * gx_bs + (long)ci * hw + p] = s;
}

}  // namespace

extern "C" int irr_conv2d_wgrad_f32(const float* x, const float* gy, float* gw, int B, int Cin, int H, int W, int Cout,
                                    int OH, int OW, int k, int stride, int dil, long x_bs, long gy_bs, void* stream) {
  if (!x || !gy || !gw || B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0) return IRR_EINVAL;
  if ((k != 1 && k != 3) || stride < 1 || dil < 1) return IRR_EINVAL;
  WgArgs a;
  a.x = x; a.gy = gy; a.gw = gw;
  a.B = B; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout; a.OH = OH; a.OW = OW;
  a.stride = stride; a.dil = dil; a.pad = ((k - 1) * dil) / 2;
  a.x_bs = x_bs; a.gy_bs = gy_bs; a.chunks_per_block = 0;
  return (k == 3) ? dispatch<3>(a, (hipStream_t)stream) : dispatch<1>(a, (hipStream_t)stream);
}

extern "C" int irr_lrelu_bwd_bias_f32(const float* gy, const float* y, float* gpre, float* gbias, int B, int C, int HW,
                                      long gy_bs, long y_bs, long gpre_bs, int lrelu, void* stream) {
  if (!gy || B <= 0 || C <= 0 || HW <= 0 || (lrelu && !y) || B > 65535 || C > 65535) return IRR_EINVAL;
  if (!gpre && !gbias) return 0;
  int chunk = 4096;
  dim3 grid(irr_cdiv(HW, chunk), C, B);
  hipLaunchKernelGGL(lrelu_bwd_bias_kernel, grid, dim3(256), 0, (hipStream_t)stream, gy, y, gpre, gbias, C, (long)HW,
                     gy_bs, y_bs, gpre_bs, lrelu, chunk);
  IRR_LAUNCH_CHECK();
  return 0;
}

extern "C" int irr_conv2d_dgrad_strided_f32(const float* gy, const float* w, float* gx, int B, int Cin, int H, int W,
                                            int Cout, int OH, int OW, int k, int stride, int dil, long gy_bs,
                                            long gx_bs, void* stream) {
  if (!gy || !w || !gx || B <= 0 || Cin <= 0 || Cout <= 0 || B > 65535 || Cin > 65535) return IRR_EINVAL;
  if ((k != 1 && k != 3) || stride < 1 || dil < 1) return IRR_EINVAL;
  dim3 grid(irr_cdiv((long)H * W, 256), Cin, B);
  hipLaunchKernelGGL(dgrad_strided_kernel, grid, dim3(256), 0, (hipStream_t)stream, gy, w, gx, Cin, H, W, Cout, OH, OW,
                     k, stride, dil, ((k - 1) * dil) / 2, gy_bs, gx_bs);
  IRR_LAUNCH_CHECK();
  return 0;
}
