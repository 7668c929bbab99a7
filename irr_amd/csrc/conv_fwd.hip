// fp32 MFMA direct convolution for the IRR-PWC conv() blocks (models/pwc_modules.py:8-19,
// models/irr_modules.py:7-18): k in {1,3}, stride in {1,2}, dilation d, "same" padding, bias, LeakyReLU(0.1),
// with the epilogue variants the decoders need (residual add, scaling, accumulate-into-output).
//
// Formulation (per wave, no LDS, no barriers):  D[co][px] += W[co][k] * X[k][px],  k = (ci, tap)
//   * v_mfma_f32_32x32x2_f32: rows i = 32 output channels, cols j = 32 output pixels, 2 k-values per
//     instruction = two consecutive input channels of one tap (lanes 0-31 -> ci=2cp, lanes 32-63 -> ci=2cp+1).
//   * pixels are a LINEAR index over (b, oy, ox); a lane owns one pixel per 32-pixel sub-tile and keeps
//     its 9 tap offsets (+ validity) for the whole K loop, so tiles may wrap rows and samples freely and the
//     X operand is one coalesced global_load_dword per (sub-tile, tap, channel pair) straight into the MFMA
//     source VGPR -- the activations are NCHW, i.e. already "K-major, pixel-contiguous", which is exactly
//     the B-fragment layout; an LDS round trip would buy nothing at the fp32 MFMA rate (64 cyc / instr).
//   * weights come from a packed [cp][tap][half][CoP] array (irr_conv_pack_weights_f32) so the A fragment
//     is one coalesced 256 B load per (tap, pair, co-tile); all waves of a CU share it through L1/L2.
//   * a wave owns MT co-tiles x NT pixel sub-tiles (MT*NT*16 accumulator VGPRs); one block = 4 independent
//     waves; grid.x walks the pixel range, grid.y the co-tile groups.
//   * epilogue writes D rows as dense 128 B pixel runs per output channel (NCHW), fusing bias, LeakyReLU,
//     residual/scale and the "+=" used for DenseNet gradient accumulation.
// The same kernel computes the stride-1 data gradient when fed transposed+flipped packed weights.
#include "common.h"
#include "pack.h"

#ifndef CONV_ORDER
#define CONV_ORDER 3        // 3 (default): one refill load issued behind each MFMA of the step; 0/1/2: block orders kept for A/B runs
#endif
#ifndef CONV_ABL
#define CONV_ABL 0          // ablation builds only: 1 = no activation loads, 2 = no activation and no weight loads
#endif
#ifndef CONV_PREFETCH_D
#define CONV_PREFETCH_D 3   // must divide 9
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvArgs {
  const float* x;
  const float* wp;
  const float* bias;
  const float* res;
  float* y;
  int B, Cin, H, W, Cout, OH, OW;
  int stride, dil, pad;
  int CoP;               // padded Cout (multiple of 32)
  long x_bs, y_bs, res_bs;
  int lrelu, accumulate;
  float alpha;
  const float* mask;     // optional: after everything else, y *= LeakyReLU'(mask[b,co,p]) for co < nmask
  long mask_bs;
  int nmask;
};

__device__ __forceinline__ float buf_load(__amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)voff, (int)soff, 0));
}

// SPLITK (tiny pyramid levels only, MT = NT = 1): the four waves of a block share ONE output tile and each walks
// a quarter of the channel pairs; partial tiles meet in LDS and wave 0 runs the epilogue.  At 6x7 / 12x14
// resolution there are too few tiles to fill 1024 SIMDs and a single wave would otherwise run the whole
// K loop (up to 2547 dependent steps) alone.
template <int MT, int NT, int KS, bool SPLITK = false>
__global__ __launch_bounds__(256, 2) void conv_fwd_kernel(const ConvArgs a) {
  constexpr int KK = KS * KS;
  // Ring refill placement.  Default (CONV_ORDER 3): the refill loads of a k-step are threaded BETWEEN its MFMAs
  // (MFMA, load, MFMA, load, ...): a wave is in-order, so a load that waits for a free slot in the memory
  // pipeline only delays one MFMA instead of the whole group.  Measured on the 565->128 level-4 conv:
  // loads-then-MFMAs 122 TFLOP/s, MFMAs-then-loads 132, interleaved 138.5 (88 % of the fp32 MFMA peak).
  constexpr bool REFILL_AFTER = (CONV_ORDER == 1) || (CONV_ORDER == 0 && MT >= 2);
  constexpr int D = (KK == 9) ? CONV_PREFETCH_D : 1;          // prefetch distance in k-steps (ring slots)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 31, half = lane >> 5;
  const long ohw = (long)a.OH * a.OW;
  const long hw = (long)a.H * a.W;
  const long total = (long)a.B * ohw;
  // XCD-major block order (common.h): blocks of one XCD take consecutive pixel ranges (shared halo rows in that XCD's L2),
  // the co-tile groups of one pixel range stay together
  const unsigned xpos = irr_xcd_order(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
  const unsigned bxp = xpos / gridDim.y;
  const long pbase = SPLITK ? (long)bxp * (NT * 32) : ((long)bxp * 4 + wave) * (NT * 32);
  if (pbase >= total) return;
  const int cog = (int)(xpos % gridDim.y);

  uint32_t voff[NT][KK];
  bool valid[NT][KK];
  uint32_t ooff[NT], roff[NT], moff[NT];
  bool pvalid[NT];
#pragma unroll
  for (int s = 0; s < NT; ++s) {
    const long p = pbase + s * 32 + j;
    pvalid[s] = p < total;
    const long pp = pvalid[s] ? p : total - 1;
    const int b = (int)(pp / ohw);
    const int r = (int)(pp - (long)b * ohw);
    const int oy = r / a.OW, ox = r - oy * a.OW;
    ooff[s] = (uint32_t)((long)b * a.y_bs + r);
    roff[s] = (uint32_t)((long)b * a.res_bs + r);
    moff[s] = (uint32_t)((long)b * a.mask_bs + r);
#pragma unroll
    for (int t = 0; t < KK; ++t) {
      const int ty = t / KS, tx = t - ty * KS;
      const int iy = oy * a.stride - a.pad + ty * a.dil;
      const int ix = ox * a.stride - a.pad + tx * a.dil;
      const bool ok = pvalid[s] && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      const int cy = min(max(iy, 0), a.H - 1), cx = min(max(ix, 0), a.W - 1);
      voff[s][t] = (uint32_t)(((long)b * a.x_bs + (long)cy * a.W + cx + (long)half * hw) * 4);
      valid[s][t] = ok;
    }
  }
  const uint32_t aoff = (uint32_t)((half * a.CoP + cog * MT * 32 + j) * 4);
  const uint32_t cop_bytes2 = (uint32_t)a.CoP * 8u;       // two k-rows per tap
  const uint32_t wstep = (uint32_t)KK * cop_bytes2;       // packed-weight bytes per channel pair
  const uint32_t xstep = (uint32_t)(hw * 8);              // activation bytes per channel pair
  const int ncp = (a.Cin + 1) >> 1;
  // odd Cin: the last pair is (Cin-2, Cin-1) with a zero weight row for Cin-2 (see pack_weights_kernel),
  // so every activation read stays inside the tensor.
  const uint32_t x_last = (uint32_t)((long)(a.Cin - 2) * hw * 4);

  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, (short)0, (int)0xffffffffu, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)a.wp, (short)0, (int)0xffffffffu, 0x00020000);

  f32x16 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int s = 0; s < NT; ++s)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][s][r] = 0.f;

  float ra[D][MT], rb[D][NT];      // operand ring, statically indexed

  // issue the loads of k-step (cp, tap) into ring slot `slot`
  auto issue = [&](int slot, int cp, int tap) {
    const int cpc = min(cp, ncp - 1);                       // past-the-end prefetches re-read the last pair
    const uint32_t xs = (cpc == ncp - 1 && (a.Cin & 1)) ? x_last : (uint32_t)cpc * xstep;
    const uint32_t ws = (uint32_t)cpc * wstep + (uint32_t)tap * cop_bytes2;
#pragma unroll
    for (int m = 0; m < MT; ++m) ra[slot][m] = (CONV_ABL >= 2) ? 0.5f + m : buf_load(wr, aoff + m * 128, ws);
#pragma unroll
    for (int s = 0; s < NT; ++s) rb[slot][s] = (CONV_ABL >= 1) ? 0.25f + s : buf_load(xr, voff[s][tap], xs);
  };

  const int cp_begin = SPLITK ? (ncp * wave) / 4 : 0;
  const int cp_end = SPLITK ? (ncp * (wave + 1)) / 4 : ncp;
#pragma unroll
  for (int d = 0; d < D; ++d) issue(d, cp_begin + d / KK, d % KK);

  for (int cp = cp_begin; cp < cp_end; ++cp) {
#pragma unroll
    for (int tap = 0; tap < KK; ++tap) {
      const int slot = tap % D;
      float av[MT], bv[NT];
#pragma unroll
      for (int m = 0; m < MT; ++m) av[m] = ra[slot][m];
#pragma unroll
      for (int s = 0; s < NT; ++s) bv[s] = valid[s][tap] ? rb[slot][s] : 0.f;
      const int ntap = tap + D;
      __builtin_amdgcn_sched_barrier(0);      // keep the software pipeline: consume slot -> refill slot -> MFMAs
      if (!REFILL_AFTER && CONV_ORDER != 3) {
        issue(slot, ntap >= KK ? cp + 1 : cp, ntap >= KK ? ntap - KK : ntap);
        __builtin_amdgcn_sched_barrier(0);
      }
#if CONV_ORDER == 3
      {                                        // one refill load behind each MFMA
        const int ncp2 = ntap >= KK ? cp + 1 : cp, ntap2 = ntap >= KK ? ntap - KK : ntap;
        const int cpc = min(ncp2, ncp - 1);
        const uint32_t xs = (cpc == ncp - 1 && (a.Cin & 1)) ? x_last : (uint32_t)cpc * xstep;
        const uint32_t ws = (uint32_t)cpc * wstep + (uint32_t)ntap2 * cop_bytes2;
#pragma unroll
        for (int q = 0; q < MT * NT; ++q) {
          const int m = q / NT, s2 = q - m * NT;
          acc[m][s2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv[s2], acc[m][s2], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (q < MT) ra[slot][q] = buf_load(wr, aoff + q * 128, ws);
          else if (q < MT + NT) rb[slot][q - MT] = buf_load(xr, voff[q - MT][ntap2], xs);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int q = MT * NT; q < MT + NT; ++q) {            // variants with more operands than MFMAs per step
          if (q < MT) ra[slot][q] = buf_load(wr, aoff + q * 128, ws);
          else rb[slot][q - MT] = buf_load(xr, voff[q - MT][ntap2], xs);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#else
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int s = 0; s < NT; ++s)
          acc[m][s] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv[s], acc[m][s], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (REFILL_AFTER) {
        issue(slot, ntap >= KK ? cp + 1 : cp, ntap >= KK ? ntap - KK : ntap);
        __builtin_amdgcn_sched_barrier(0);
      }
#endif
    }
  }

  if (SPLITK) {
    __shared__ float part[3][16][64];
    if (wave > 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) part[wave - 1][r][lane] = acc[0][0][r];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] += (part[0][r][lane] + part[1][r][lane]) + part[2][r][lane];
  }

  // ---- epilogue: D[i][j], i = (r&3) + 8*(r>>2) + 4*half, j = lane&31 ----
  // Read-modify-write operands (residual, "+=", LeakyReLU'-mask) of RB rows x NT sub-tiles are loaded as one batch before the
  // first store of the batch (loads inside the store loop each waited for their own round trip to memory).
  constexpr int RB = (MT * NT >= 8) ? 2 : 4;
#pragma unroll
  for (int m = 0; m < MT; ++m) {
#pragma unroll
    for (int r0 = 0; r0 < 16; r0 += RB) {
      float rv[RB][NT], dv[RB][NT], mv[RB][NT], bsv[RB];
      bool cok[RB];
#pragma unroll
      for (int k = 0; k < RB; ++k) {
        const int r = r0 + k;
        const int co = (cog * MT + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        cok[k] = co < a.Cout;
        bsv[k] = (cok[k] && a.bias) ? a.bias[co] : 0.f;
#pragma unroll
        for (int s = 0; s < NT; ++s) {
          const bool ok = cok[k] && pvalid[s];
          rv[k][s] = (ok && a.res) ? a.res[roff[s] + (long)co * ohw] : 0.f;
          dv[k][s] = (ok && a.accumulate) ? a.y[ooff[s] + (long)co * ohw] : 0.f;
          mv[k][s] = (ok && a.mask && co < a.nmask) ? a.mask[moff[s] + (long)co * ohw] : 1.f;
        }
      }
#pragma unroll
      for (int k = 0; k < RB; ++k) {
        const int r = r0 + k;
        const int co = (cog * MT + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (!cok[k]) continue;
#pragma unroll
        for (int s = 0; s < NT; ++s) {
          if (!pvalid[s]) continue;
          float v = acc[m][s][r] + bsv[k];
          if (a.lrelu) v = irr_lrelu(v);
          v = a.res ? rv[k][s] + a.alpha * v : v * a.alpha;
          v += dv[k][s];
          if (a.mask && co < a.nmask) v *= irr_lrelu_grad(mv[k][s]);
          a.y[ooff[s] + (long)co * ohw] = v;
        }
      }
    }
  }
}

// wp[((cp*KK + tap)*2 + half)*CoP + co]
__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cin, int Cout, int KK,
                                    int CoP, int transpose, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) pack_f32_elem(w, wp, Cin, Cout, KK, CoP, transpose, n, i);
}

// (combined data-gradient matrices of the DenseNet backward: see pack_sub_elem in pack.h)
__global__ void pack_weights_sub_kernel(const float* __restrict__ w, float* __restrict__ wp, int w_cin, int w_cout, int KK,
                                        int chan0, int nchan, int CoP, int row_offset, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) pack_sub_elem(w, wp, w_cin, KK, chan0, nchan, CoP, row_offset, i);
}

template <int MT, int NT, int KS>
int launch(const ConvArgs& a, hipStream_t st) {
  const long total = (long)a.B * a.OH * a.OW;
  const int cot = a.CoP / 32;
  dim3 grid(irr_cdiv(total, 4L * NT * 32), irr_cdiv(cot, MT), 1);
  hipLaunchKernelGGL((conv_fwd_kernel<MT, NT, KS>), grid, dim3(256), 0, st, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

template <int KS>
int launch_splitk(const ConvArgs& a, hipStream_t st) {
  const long total = (long)a.B * a.OH * a.OW;
  dim3 grid(irr_cdiv(total, 32), a.CoP / 32, 1);
  hipLaunchKernelGGL((conv_fwd_kernel<1, 1, KS, true>), grid, dim3(256), 0, st, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

// (MT, NT) tiling choice, shared by the launcher and by irr_conv2d_fwd_variant (bench / profile labelling)
// *mt == 0 means: split-K variant (<1,1> tile, 4 waves over the channel pairs)
static void pick_variant(int cot, long total, int* mt, int* nt, int cin = 0) {
  const long ptiles = (total + 31) / 32;
  if (ptiles * cot < 2048 && cin >= 64) { *mt = 0; *nt = 1; return; }
  if (ptiles * cot < 4096) {            // small problems: favour many waves over register blocking
    if (cot >= 2 && ptiles * ((cot + 1) / 2) >= 1024) { *mt = 2; *nt = 1; return; }
    *mt = 1; *nt = 1; return;
  }
  if (cot == 1) { *mt = 1; *nt = 4; return; }
  if (cot == 2) { *mt = 2; *nt = 4; return; }
  if (cot == 3) { *mt = 3; *nt = 2; return; }
  *mt = 4; *nt = 2;
}

template <int KS>
int dispatch(const ConvArgs& a, hipStream_t st) {
  int mt, nt;
  pick_variant(a.CoP / 32, (long)a.B * a.OH * a.OW, &mt, &nt, a.Cin);
  if (mt == 0) return launch_splitk<KS>(a, st);
  switch (mt * 10 + nt) {
    case 11: return launch<1, 1, KS>(a, st);
    case 21: return launch<2, 1, KS>(a, st);
    case 14: return launch<1, 4, KS>(a, st);
    case 24: return launch<2, 4, KS>(a, st);
    case 32: return launch<3, 2, KS>(a, st);
    default: return launch<4, 2, KS>(a, st);
  }
}

}  // namespace

extern "C" int irr_conv2d_fwd_variant(int B, int Cout, int OH, int OW, int k) {
  int mt, nt;
  pick_variant((Cout + 31) / 32, (long)B * OH * OW, &mt, &nt);
  return mt * 100 + nt * 10 + k;
}

extern "C" long irr_conv_packed_weight_elems(int Cin, int Cout, int k) {
  const long CoP = (Cout + 31) / 32 * 32;
  // + slack: a wave that owns MT co-tiles may read up to 3 tiles past CoP in the LAST row (rows past Cout are
  // discarded in the epilogue, but the addresses must stay inside the allocation)
  return (long)((Cin + 1) / 2) * k * k * 2 * CoP + 4 * 32;
}

extern "C" int irr_conv_pack_weights_f32(const float* w, float* wp, int Cin, int Cout, int k, int transpose,
                                         void* stream) {
  if (!w || !wp || Cin <= 0 || Cout <= 0 || (k != 1 && k != 3)) return IRR_EINVAL;
  const int CoP = (Cout + 31) / 32 * 32;
  const long n = irr_conv_packed_weight_elems(Cin, Cout, k);
  hipLaunchKernelGGL(pack_weights_kernel, dim3(irr_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, w, wp, Cin, Cout,
                     k * k, CoP, transpose, n);
  IRR_LAUNCH_CHECK();
  return 0;
}

extern "C" int irr_conv_pack_weights_sub_f32(const float* w, float* wp, int w_cin, int w_cout, int k, int chan0, int nchan,
                                             int CoP, int row_offset, void* stream) {
  if (!w || !wp || w_cin <= 0 || w_cout <= 0 || (k != 1 && k != 3) || chan0 < 0 || nchan <= 0 || chan0 + nchan > w_cin ||
      CoP < nchan || (CoP & 31) || row_offset < 0)
    return IRR_EINVAL;
  const long n = (long)w_cout * k * k * CoP;
  hipLaunchKernelGGL(pack_weights_sub_kernel, dim3(irr_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, w, wp, w_cin,
                     w_cout, k * k, chan0, nchan, CoP, row_offset, n);
  IRR_LAUNCH_CHECK();
  return 0;
}

extern "C" int irr_conv2d_fwd_f32(const float* x, const float* wp, const float* bias, const float* res, float* y, int B,
                                  int Cin, int H, int W, int Cout, int OH, int OW, int k, int stride, int dil,
                                  long x_bs, long y_bs, long res_bs, int lrelu, float alpha, int accumulate,
                                  const float* mask, long mask_bs, int nmask, void* stream) {
  if (!x || !wp || !y || B <= 0 || Cin < 2 || Cout <= 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0) return IRR_EINVAL;
  if ((k != 1 && k != 3) || stride < 1 || dil < 1) return IRR_EINVAL;
  const int pad = ((k - 1) * dil) / 2;
  if (OH != (H + 2 * pad - dil * (k - 1) - 1) / stride + 1 || OW != (W + 2 * pad - dil * (k - 1) - 1) / stride + 1)
    return IRR_EINVAL;
  ConvArgs a;
  a.wp = wp; a.bias = bias;
  a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout; a.OH = OH; a.OW = OW;
  a.stride = stride; a.dil = dil; a.pad = pad;
  a.CoP = (Cout + 31) / 32 * 32;
  a.x_bs = x_bs; a.y_bs = y_bs; a.res_bs = res_bs;
  a.lrelu = lrelu; a.accumulate = accumulate; a.alpha = alpha;
  a.mask_bs = mask_bs; a.nmask = nmask;
  // 32-bit byte offsets inside the kernel: split the batch so every offset stays below 4 GiB
  const long lim = (1L << 30) - (long)(Cin + 2) * H * W - 1;       // elements
  long per = x_bs > 0 ? lim / (x_bs > y_bs ? (x_bs > res_bs ? x_bs : res_bs) : (y_bs > res_bs ? y_bs : res_bs)) : B;
  if (mask && mask_bs > 0 && lim / mask_bs < per) per = lim / mask_bs;
  if (per < 1) return IRR_EINVAL;
  if (per > B) per = B;
  for (int b0 = 0; b0 < B; b0 += (int)per) {
    const int nb = (B - b0) < per ? (B - b0) : (int)per;
    a.B = nb;
    a.x = x + (long)b0 * x_bs;
    a.y = y + (long)b0 * y_bs;
    a.res = res ? res + (long)b0 * res_bs : nullptr;
    a.mask = mask ? mask + (long)b0 * mask_bs : nullptr;
    const int rc = (k == 3) ? dispatch<3>(a, (hipStream_t)stream) : dispatch<1>(a, (hipStream_t)stream);
    if (rc) return rc;
  }
  return 0;
}
