// Weight gradient of the conv() blocks on fp32 MFMA:
//   dW[co][ci][tap] += sum_p gy[co][p] * x[ci][p + off(tap)]        (p = linear (b,oy,ox) output pixel)
//
// Here the reduction (K) dimension is the PIXEL axis, but MFMA fragments spread lanes over the
// non-reduced axes (32 rows x 2 k-values), while NCHW memory is contiguous along pixels.  So the operands
// are staged through LDS: global loads are coalesced along pixels (a lane = a pixel, 128 B runs per
// channel row), written as [channel][pixel] with an odd row pitch (33), and read back with lanes along the
// channel axis (stride 33 words -> conflict-free ds_read_b32).  One 32-pixel stage feeds 16 MFMA k-steps.
//
// Work split: a block owns MTB co-tiles x ONE ci-tile and a contiguous slice of the pixel range (split-K).
// Wave (m, ty) accumulates co-tile m against the three taps of kernel row ty (3 accumulator tiles = 48
// VGPRs), so a block has 3*MTB waves that share one staged x tile (9 tap-shifted copies of 32 channels) and
// MTB gy tiles.  The next stage is prefetched into VGPRs (one full stage in flight per block), its loads
// interleaved with the MFMAs of the current stage.  Split-K partials are added with COALESCED fp32 atomics
// into a [co][tap][ci] workspace (lanes = ci are contiguous) and then folded into dW[co][ci][tap].
#include "common.h"
#include "amax.h"
#include "wgrad_reduce.h"
#include <stdlib.h>

#ifndef WG_ABL
#define WG_ABL 0      // ablation builds only (tools/): 1 = no global loads, 2 = + no LDS writes, 3 = + no barriers
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int KP = 32;          // pixels per LDS stage
constexpr int PITCH = KP + 1;

struct WgArgs {
  const float* x;
  const float* gy;
  float* gw;                    // workspace, [grid.x][Cout][KK][Cin]: one partial result per block column (plain stores)
  long n;                       // Cout * KK * Cin
  float* gbias;                 // optional: gbias[co] += sum of gy over pixels (taken from the staged gy tiles)
  float alpha;                  // scales both results (residual branches: y = x + alpha*conv(...))
  int B, Cin, H, W, Cout, OH, OW;
  int stride, dil, pad;
  long x_bs, gy_bs;
  int chunks_per_block;
};

__device__ __forceinline__ float wg_buf_load(__amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)voff, (int)soff, 0));
}

// Out-of-range marker for a voffset.  The hardware bounds check of a raw buffer load covers voffset (+imm) but
// NOT soffset, so channel/row offsets are added into the voffset (one v_add_u32 per load) and the marker is
// chosen so that marker + any in-view offset (< 2 GiB, enforced by the launcher) still exceeds num_records
// without wrapping: such lanes read 0 and touch no memory.
constexpr uint32_t OOB = 0x80000000u;

template <int MTB, int KS>
__global__ __launch_bounds__(MTB* KS * 64) void conv_wgrad_kernel(const WgArgs a) {
  constexpr int KK = KS * KS;
  constexpr int NW = MTB * KS;                 // waves: (co-tile m, kernel row ty)
  constexpr int AROWS = MTB * 32, BROWS = KK * 32, ROWS = AROWS + BROWS;
  // Staging roles are per wave: NWB waves stage the x tile, NWA the gy tile (a lone wave does both).  A
  // half-wave (32 lanes = the 32 pixels of a stage) loads one channel row per instruction.  Every x stager
  // owns CPB input channels x all taps, so the tap geometry (9 voffsets) is computed once per stage; the
  // channel enters through the scalar soffset; taps that fall outside the image get an out-of-range voffset
  // and the buffer load's hardware bounds check returns 0 -- no per-load VALU work at all.
  constexpr int NWB = (NW == 1) ? 1 : (NW >= 12) ? 8 : (NW >= 6) ? 4 : (NW >= 4) ? 2 : (NW == 3) ? 2 : 1;
  constexpr int NWA = (NW == 1) ? 1 : NW - NWB;
  constexpr int NB = 2 * NWB, NA = 2 * NWA;    // half-waves per role
  constexpr int CPB = 32 / NB;                 // input channels per x-staging half-wave
  constexpr int NRB = KK * CPB;                // staged values per lane (x stagers)
  constexpr int NRA = (AROWS + NA - 1) / NA;   // staged values per lane (gy stagers)
  constexpr int NR = (NW == 1) ? NRB + NRA : (NRB > NRA ? NRB : NRA);
  __shared__ float S[ROWS][PITCH];             // rows [0,AROWS): gy tiles; then tap-major x tiles

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / KS, ty_w = wave - wm * KS;
  const int j = lane & 31, half = lane >> 5;
  const int co0 = blockIdx.z * AROWS, ci0 = blockIdx.y * 32;
  const long ohw = (long)a.OH * a.OW;
  const long hw = (long)a.H * a.W;
  const long total = (long)a.B * ohw;
  const long nchunks = (total + KP - 1) / KP;
  const long c_begin = (long)blockIdx.x * a.chunks_per_block;
  const long c_end = min(nchunks, c_begin + a.chunks_per_block);

  f32x16 acc[KS];
#pragma unroll
  for (int t = 0; t < KS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const bool do_b = (NW == 1) || (wave < NWB);            // wave-uniform roles
  const bool do_a = (NW == 1) || (wave >= NWB);
  const int wb = wave;                                     // x-stager wave index
  const int wa = (NW == 1) ? 0 : wave - NWB;               // gy-stager wave index
  const int px = lane & 31;

  // extents of the two views (bytes) for the hardware bounds check
  const uint32_t x_bytes = (uint32_t)min((long)0x7ffffffcL, ((long)(a.B - 1) * a.x_bs + (long)a.Cin * hw) * 4);
  const uint32_t g_bytes = (uint32_t)min((long)0x7ffffffcL, ((long)(a.B - 1) * a.gy_bs + (long)a.Cout * ohw) * 4);
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, (short)0, (int)x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc((void*)a.gy, (short)0, (int)g_bytes, 0x00020000);
  const uint32_t xs0 = (uint32_t)((long)(ci0 + 2 * wb) * hw * 4);     // soffset of this wave's first channel pair
  const uint32_t xsc = (uint32_t)((long)NB * hw * 4);                 // soffset step between owned channels
  const uint32_t gs0 = (uint32_t)((long)(co0 + 2 * wa) * ohw * 4);
  const uint32_t gsc = (uint32_t)((long)NA * ohw * 4);

  float stg[NR];
  const bool want_bias = do_a && a.gbias != nullptr && blockIdx.y == 0;   // one ci-tile column sums the gy tiles
  float bsum[NRA];
#pragma unroll
  for (int ia = 0; ia < NRA; ++ia) bsum[ia] = 0.f;
  uint32_t xv[KK];                              // per-tap voffset of this lane's pixel (+ half-wave channel)
  uint32_t gv = OOB;
  auto decode = [&](long c) {
    const long p = c * KP + px;
    const bool pv = p < total;
    const long pp = pv ? p : total - 1;
    const int b = (int)(pp / ohw);
    const int r = (int)(pp - (long)b * ohw);
    if (do_a) gv = pv ? (uint32_t)(((long)b * a.gy_bs + r + (long)half * ohw) * 4) : OOB;
    if (do_b) {
      const int oy = r / a.OW, ox = r - oy * a.OW;
      const long base = (long)b * a.x_bs + (long)half * hw;
#pragma unroll
      for (int t = 0; t < KK; ++t) {
        const int ty = t / KS, tx = t - ty * KS;
        const int iy = oy * a.stride - a.pad + ty * a.dil, ix = ox * a.stride - a.pad + tx * a.dil;
        const bool ok = pv && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        xv[t] = ok ? (uint32_t)((base + (long)iy * a.W + ix) * 4) : OOB;
      }
    }
  };
  // issue the global load of staged value i
  auto issue1 = [&](int i) {
    if (WG_ABL >= 1) { stg[i] = 1.f; return; }
    if (do_b && i < NRB) {
      const int t = i / CPB, cc = i - t * CPB;                 // compile-time after unrolling
      stg[i] = wg_buf_load(xr, xv[t] + xs0 + cc * xsc, 0);
    }
    if (do_a && i >= ((NW == 1) ? NRB : 0) && i < ((NW == 1) ? NRB : 0) + NRA) {
      const int ia = i - ((NW == 1) ? NRB : 0);
      if (2 * wa + ia * NA < AROWS) stg[i] = wg_buf_load(gr, gv + gs0 + ia * gsc, 0);
    }
  };
  auto store_stage = [&]() {
    if (do_b) {
#pragma unroll
      for (int i = 0; i < NRB; ++i) {
        const int t = i / CPB, cc = i - t * CPB;
        S[AROWS + t * 32 + 2 * wb + half + cc * NB][px] = stg[i];
      }
    }
    if (do_a) {
#pragma unroll
      for (int ia = 0; ia < NRA; ++ia) {
        const int row = 2 * wa + half + ia * NA;
        if (2 * wa + ia * NA < AROWS) {
          S[row][px] = stg[((NW == 1) ? NRB : 0) + ia];
          if (want_bias) bsum[ia] += stg[((NW == 1) ? NRB : 0) + ia];
        }
      }
    }
  };

  if (c_begin < c_end) {
    decode(c_begin);
#pragma unroll
    for (int i = 0; i < NR; ++i) issue1(i);
  }
  constexpr int KSTEPS = KP / 2;
  constexpr int PER = (NR + KSTEPS - 1) / KSTEPS;      // prefetch loads interleaved per k-step
  for (long c = c_begin; c < c_end; ++c) {
    if (WG_ABL < 3) __syncthreads();           // previous stage fully consumed
    if (WG_ABL < 2 || c == c_begin) store_stage();
    if (WG_ABL < 3) __syncthreads();
    const bool more = c + 1 < c_end;           // block-uniform
    if (more) decode(c + 1);
    const float* arow = &S[wm * 32 + j][half];
    const float* brow = &S[AROWS + (ty_w * KS) * 32 + j][half];
#pragma unroll
    for (int k = 0; k < KSTEPS; ++k) {
      const float av = arow[2 * k];
      float bv[KS];
#pragma unroll
      for (int t = 0; t < KS; ++t) bv[t] = brow[t * 32 * PITCH + 2 * k];
      // next stage's loads are threaded between the MFMAs of the k-step (an in-order wave that has to wait
      // for a slot in the memory pipeline then delays one MFMA, not three)
#pragma unroll
      for (int t = 0; t < KS; ++t) {
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[t], acc[t], 0, 0, 0);
        if (more) {
#pragma unroll
          for (int u = t; u < PER; u += KS)
            if (k * PER + u < NR) issue1(k * PER + u);
        }
      }
    }
  }

  if (want_bias) {
#pragma unroll
    for (int ia = 0; ia < NRA; ++ia) {
      float v = bsum[ia];
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);          // reduce over the 32 pixels of the half-wave
      const int co = co0 + 2 * wa + half + ia * NA;
      if (px == 0 && 2 * wa + ia * NA < AROWS && co < a.Cout) unsafeAtomicAdd(a.gbias + co, a.alpha * v);
    }
  }

  // D[i][jj]: i = co_local = (r&3) + 8*(r>>2) + 4*half, jj = ci_local = lane&31
  const int ci = ci0 + j;
  if (ci < a.Cin) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (co >= a.Cout) continue;
      // workspace layout [block column][co][tap][ci]: lanes (ci) are contiguous; wgrad_reduce_kernel sums the columns
      float* dst = a.gw + (long)blockIdx.x * a.n + ((long)co * KK + ty_w * KS) * a.Cin + ci;
#pragma unroll
      for (int t = 0; t < KS; ++t) {
        if (a.n) dst[(long)t * a.Cin] = a.alpha * acc[t][r];
        else unsafeAtomicAdd(dst + (long)t * a.Cin, a.alpha * acc[t][r]);      // IRR_WGRAD_ATOMIC=1: one shared, zeroed image
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Halo-tile variant for the common case k=3, stride=1, dilation=1 (every DenseNet / refinement / occlusion-
// upsampler conv; only the dilated context-network layers take the tap-copy kernel above).
// A stage is a 2-D patch of TR rows x 30 output columns; x is staged ONCE with a one-pixel halo
// ((TR+2) rows x 32 columns per channel -- exactly one half-wave load per halo row) and the nine taps read
// shifted positions of the same LDS tile, so the staged volume per MFMA drops ~3x versus nine tap copies.
// ---------------------------------------------------------------------------------------------------------
template <int MTB, int TR>
__global__ __launch_bounds__(MTB * 3 * 64) void conv_wgrad_halo_kernel(const WgArgs a) {
  constexpr int KS = 3, KK = 9;
  constexpr int NW = MTB * KS;
  constexpr int TC = 30;                         // output columns per tile (+2 halo = 32 = one half-wave)
  constexpr int AROWS = MTB * 32;
  constexpr int AP = TR * 32 + 1, BP = (TR + 2) * 32 + 1;     // odd pitches: conflict-free channel-strided reads
  constexpr int NWB = (NW >= 12) ? 4 : (NW >= 9) ? 4 : (NW >= 6) ? 2 : 2;
  constexpr int NWA = NW - NWB;
  constexpr int NB = 2 * NWB, NA = 2 * NWA;
  constexpr int CPB = 32 / NB;                   // channels per x-staging half-wave
  constexpr int NRB = CPB * (TR + 2);
  constexpr int ALOADS = AROWS * TR;             // half-wave loads of the gy tile
  constexpr int NRA = (ALOADS + NA - 1) / NA;
  constexpr int NR = NRB > NRA ? NRB : NRA;
  __shared__ float SA[AROWS][AP];
  __shared__ float SB[32][BP];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / KS, ty_w = wave - wm * KS;
  const int j = lane & 31, half = lane >> 5;
  const int co0 = blockIdx.z * AROWS, ci0 = blockIdx.y * 32;
  const long hw = (long)a.H * a.W;
  const int tiles_x = (a.W + TC - 1) / TC, tiles_y = (a.H + TR - 1) / TR;
  const long ntiles = (long)a.B * tiles_y * tiles_x;
  const long c_begin = (long)blockIdx.x * a.chunks_per_block;
  const long c_end = min(ntiles, c_begin + a.chunks_per_block);

  f32x16 acc[KS];
#pragma unroll
  for (int t = 0; t < KS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const bool do_b = wave < NWB;
  const int wb = wave, wa = wave - NWB;
  const int hl = lane & 31;                      // column inside the half-wave

  const uint32_t x_bytes = (uint32_t)min((long)0x7ffffffcL, ((long)(a.B - 1) * a.x_bs + (long)a.Cin * hw) * 4);
  const uint32_t g_bytes = (uint32_t)min((long)0x7ffffffcL, ((long)(a.B - 1) * a.gy_bs + (long)a.Cout * hw) * 4);
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, (short)0, (int)x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc((void*)a.gy, (short)0, (int)g_bytes, 0x00020000);
  const uint32_t xs0 = (uint32_t)((long)(ci0 + 2 * wb) * hw * 4);
  const uint32_t xsc = (uint32_t)((long)NB * hw * 4);

  float stg[NR];
  const bool want_bias = !do_b && a.gbias != nullptr && blockIdx.y == 0;
  float bsum[NRA];
#pragma unroll
  for (int i = 0; i < NRA; ++i) bsum[i] = 0.f;
  uint32_t xv[TR + 2];                           // x stagers: voffset of halo row hr at this lane's halo column
  uint32_t gv[TR];                               // gy stagers: voffset of tile row r at this lane's column
  auto decode = [&](long c) {
    const int tx_ = (int)(c % tiles_x);
    const long r1 = c / tiles_x;
    const int ty_ = (int)(r1 % tiles_y);
    const int b = (int)(r1 / tiles_y);
    const int y0 = ty_ * TR, x0 = tx_ * TC;
    if (do_b) {
      const int x = x0 - 1 + hl;
      const bool xok = x >= 0 && x < a.W;
      const long base = (long)b * a.x_bs + (long)half * hw + x;
#pragma unroll
      for (int hr = 0; hr < TR + 2; ++hr) {
        const int y = y0 - 1 + hr;
        xv[hr] = (xok && y >= 0 && y < a.H) ? (uint32_t)((base + (long)y * a.W) * 4) : OOB;
      }
    } else {
      const int x = x0 + hl;
      const bool xok = hl < TC && x < a.W;
      const long base = (long)b * a.gy_bs + x;
#pragma unroll
      for (int r = 0; r < TR; ++r) {
        const int y = y0 + r;
        gv[r] = (xok && y < a.H) ? (uint32_t)((base + (long)y * a.W) * 4) : OOB;
      }
    }
  };
  auto issue1 = [&](int i) {
    if (WG_ABL >= 1) { stg[i] = 1.f; return; }
    if (do_b) {
      if (i < NRB) {
        const int cc = i / (TR + 2), hr = i - cc * (TR + 2);
        stg[i] = wg_buf_load(xr, xv[hr] + xs0 + cc * xsc, 0);
      }
    } else if (i < NRA) {
      // half-wave load id q = (2*wa + half) + i*NA  ->  (row = q / TR, r = q % TR)
      const int q = 2 * wa + half + i * NA;
      const int row = q / TR, r = q - row * TR;
      const uint32_t v = (TR == 2) ? (r ? gv[1] : gv[0]) : (r == 0 ? gv[0] : r == 1 ? gv[1] : r == 2 ? gv[TR > 2 ? 2 : 0] : gv[TR > 3 ? 3 : 0]);
      const uint32_t vo = (q < ALOADS && co0 + row < a.Cout) ? v + (uint32_t)((long)(co0 + row) * hw * 4) : OOB;
      stg[i] = wg_buf_load(gr, vo, 0);
    }
  };
  auto store_stage = [&]() {
    if (do_b) {
#pragma unroll
      for (int i = 0; i < NRB; ++i) {
        const int cc = i / (TR + 2), hr = i - cc * (TR + 2);
        SB[2 * wb + half + cc * NB][hr * 32 + hl] = stg[i];
      }
    } else {
#pragma unroll
      for (int i = 0; i < NRA; ++i) {
        const int q = 2 * wa + half + i * NA;
        const int row = q / TR, r = q - row * TR;
        if (q < ALOADS) {
          SA[row][r * 32 + hl] = stg[i];
          if (want_bias) bsum[i] += stg[i];
        }
      }
    }
  };

  if (c_begin < c_end) {
    decode(c_begin);
#pragma unroll
    for (int i = 0; i < NR; ++i) issue1(i);
  }
  constexpr int KROW = TC / 2;                  // k-steps per tile row
  // next stage's loads are threaded between the MFMAs of the first LROWS tile rows (static schedule); the remaining
  // rows run a rolled loop, which keeps the register footprint of tall tiles (TR = 4) in check
  constexpr int LROWS = (TR <= 2) ? TR : 2;
  constexpr int PER = (NR + LROWS * KROW - 1) / (LROWS * KROW);
  for (long c = c_begin; c < c_end; ++c) {
    if (WG_ABL < 3) __syncthreads();
    if (WG_ABL < 2 || c == c_begin) store_stage();
    if (WG_ABL < 3) __syncthreads();
    const bool more = c + 1 < c_end;
    if (more) decode(c + 1);
    const float* arow = &SA[wm * 32 + j][half];
    const float* brow = &SB[j][ty_w * 32 + half];
#pragma unroll
    for (int k = 0; k < LROWS * KROW; ++k) {
      const int r = k / KROW, cpair = k - r * KROW;
      const float av = arow[r * 32 + 2 * cpair];
      float bv[KS];
#pragma unroll
      for (int t = 0; t < KS; ++t) bv[t] = brow[r * 32 + 2 * cpair + t];
#pragma unroll
      for (int t = 0; t < KS; ++t) {
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[t], acc[t], 0, 0, 0);
        if (more) {                            // next stage's loads threaded between the MFMAs
#pragma unroll
          for (int u = t; u < PER; u += KS)
            if (k * PER + u < NR) issue1(k * PER + u);
        }
      }
    }
    for (int r = LROWS; r < TR; ++r) {
#pragma unroll 5
      for (int cpair = 0; cpair < KROW; ++cpair) {
        const float av = arow[r * 32 + 2 * cpair];
        float bv[KS];
#pragma unroll
        for (int t = 0; t < KS; ++t) bv[t] = brow[r * 32 + 2 * cpair + t];
#pragma unroll
        for (int t = 0; t < KS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[t], acc[t], 0, 0, 0);
      }
    }
  }

  if (want_bias) {
#pragma unroll
    for (int i = 0; i < NRA; ++i) {
      float v = bsum[i];
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
      const int q = 2 * wa + half + i * NA;
      const int co = co0 + q / TR;
      if (hl == 0 && q < ALOADS && co < a.Cout) unsafeAtomicAdd(a.gbias + co, a.alpha * v);
    }
  }

  const int ci = ci0 + j;
  if (ci < a.Cin) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (co >= a.Cout) continue;
      float* dst = a.gw + (long)blockIdx.x * a.n + ((long)co * KK + ty_w * KS) * a.Cin + ci;
#pragma unroll
      for (int t = 0; t < KS; ++t) {
        if (a.n) dst[(long)t * a.Cin] = a.alpha * acc[t][r];
        else unsafeAtomicAdd(dst + (long)t * a.Cin, a.alpha * acc[t][r]);      // IRR_WGRAD_ATOMIC=1: one shared, zeroed image
      }
    }
  }
}

static int cu_count() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

// plan-only mode: the launch functions record their grid.x (= number of partial workspace images) and launch nothing
// (thread-local: the hand-over is between two statements of one host thread; concurrent callers do not see each other)
static thread_local bool g_plan_only = false;
static thread_local long g_parts = 0;

template <int MTB, int KS>
int launch(WgArgs a, hipStream_t st) {
  const long total = (long)a.B * a.OH * a.OW;
  const long nchunks = (total + KP - 1) / KP;
  const int gy_ = irr_cdiv(a.Cin, 32), gz_ = irr_cdiv(a.Cout, MTB * 32);
  // split-K so that the grid is (just under) a whole number of residency rounds: equal-sized blocks in
  // ROUNDS full waves of the chip, no ragged tail; few rounds keep the atomic epilogue small.
  static int occ = 0;
  if (!occ) {
    int o = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, conv_wgrad_kernel<MTB, KS>, MTB * KS * 64, 0) != hipSuccess || o < 1) o = 1;
    occ = o;
  }
  const long slots = (long)occ * cu_count();
  constexpr int ROUNDS = 2;
  long xs = (slots * ROUNDS) / ((long)gy_ * gz_);
  if (xs < 1) xs = 1;
  long cpb = (nchunks + xs - 1) / xs;
  if (cpb < 8) cpb = 8;
  a.chunks_per_block = (int)cpb;
  dim3 grid(irr_cdiv(nchunks, cpb), gy_, gz_);
  g_parts = grid.x;
  if (g_plan_only) return 0;
  hipLaunchKernelGGL((conv_wgrad_kernel<MTB, KS>), grid, dim3(MTB * KS * 64), 0, st, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

template <int MTB, int TR>
int launch_halo(WgArgs a, hipStream_t st) {
  const int tiles_x = (a.W + 29) / 30, tiles_y = (a.H + TR - 1) / TR;
  const long ntiles = (long)a.B * tiles_y * tiles_x;
  const int gy_ = irr_cdiv(a.Cin, 32), gz_ = irr_cdiv(a.Cout, MTB * 32);
  static int occ = 0;
  if (!occ) {
    int o = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, conv_wgrad_halo_kernel<MTB, TR>, MTB * 3 * 64, 0) != hipSuccess || o < 1) o = 1;
    occ = o;
  }
  const long slots = (long)occ * cu_count();
  constexpr int ROUNDS = 2;
  long xs = (slots * ROUNDS) / ((long)gy_ * gz_);
  if (xs < 1) xs = 1;
  long cpb = (ntiles + xs - 1) / xs;
  if (cpb < 4) cpb = 4;
  a.chunks_per_block = (int)cpb;
  dim3 grid(irr_cdiv(ntiles, cpb), gy_, gz_);
  g_parts = grid.x;
  if (g_plan_only) return 0;
  hipLaunchKernelGGL((conv_wgrad_halo_kernel<MTB, TR>), grid, dim3(MTB * 3 * 64), 0, st, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

static int dispatch_halo(const WgArgs& a, hipStream_t st) {
  const int cot = (a.Cout + 31) / 32;
#ifdef WG_TR4
  if (cot == 1) return launch_halo<1, 4>(a, st);
  if (cot == 2) return launch_halo<2, 4>(a, st);
  if (cot == 3) return launch_halo<3, 4>(a, st);
#else
  if (cot == 1) return launch_halo<1, 2>(a, st);
  if (cot == 2) return launch_halo<2, 2>(a, st);
  if (cot == 3) return launch_halo<3, 2>(a, st);
#endif
  return launch_halo<4, 2>(a, st);
}

template <int KS>
int dispatch(const WgArgs& a, hipStream_t st) {
  const int cot = (a.Cout + 31) / 32;
  if (cot == 1) return launch<1, KS>(a, st);
  if (cot == 2) return launch<2, KS>(a, st);
  if (cot == 3) return launch<3, KS>(a, st);
  return launch<4, KS>(a, st);
}

// gw[co][ci][tap] += sum over the P block columns of ws[p][co][tap][ci]: 256 threads = 64 consecutive workspace elements
// x 4 column lanes (coalesced reads), fixed summation order -> no atomics, no zeroed workspace, reproducible bits.
// (body shared with the batched fold of many launches: wgrad_reduce.h)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const IrrReduceJob J) {
  __shared__ float red[3][256];
  irr_reduce_block(J, blockIdx.x, red);
}

// gpre = gy * lrelu'(y) ; gbias[c] += sum_p gpre
__global__ __launch_bounds__(256) void lrelu_bwd_bias_kernel(const float* __restrict__ gy, const float* __restrict__ y,
                                                            float* __restrict__ gpre, float* __restrict__ gbias,
                                                            int C, long HW, long gy_bs, long y_bs, long gpre_bs,
                                                            int lrelu, int chunk, float* __restrict__ amax) {
  const int c = blockIdx.y, b = blockIdx.z;
  const long p0 = (long)blockIdx.x * chunk;
  const long p1 = min(HW, p0 + chunk);
  const float* g = gy + (long)b * gy_bs + (long)c * HW;
  const float* yy = y ? y + (long)b * y_bs + (long)c * HW : nullptr;
  float* o = gpre ? gpre + (long)b * gpre_bs + (long)c * HW : nullptr;
  float s = 0.f, m = 0.f;
  for (long p = p0 + threadIdx.x; p < p1; p += 256) {
    float v = g[p];
    if (lrelu) v *= irr_lrelu_grad(yy[p]);
    if (o) o[p] = v;
    s += v;
    m = x3_amax_fold(m, v);
  }
  if (amax) x3_amax_publish_block256(m, amax);              // max |gpre|: the amax slot of the fp16x2 launches that read it
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

// Stride-2, 3x3, padding-1 data gradient with a handful of RESULT channels (the image gradient of the first pyramid conv,
// 16 -> 3: the reference asks for it, runtime.py:158-162 sets requires_grad on the inputs).  A thread owns one 2x2 block of
// gx = the four parity classes of the transposed conv: it reads the 2x2 neighbourhood of gy once per output channel and uses
// 1 + 2 + 2 + 4 = 9 taps per (co, ci) -- no zero-interleaved copy of gy, no padded MFMA tile (the gather kernel above: 1.04 ms,
// zero-interleave + MFMA: 1.36 ms at 384x448x64).  Weights (Cout x CIN x 9) are staged in LDS and read wave-uniformly.
template <int CIN>
__global__ __launch_bounds__(256) void dgrad_s2k3_smallci_kernel(const float* __restrict__ gy, const float* __restrict__ w,
                                                                float* __restrict__ gx, int H, int W, int Cout, int OH, int OW,
                                                                long gy_bs, long gx_bs, int Cin_total) {
  // blockIdx.z = group of CIN result channels (the other stride-2 layers of the pyramid: 16 ... 128 result channels in groups
  // of four -- 21 GFLOP of useful work per step, which the zero-interleaved MFMA route spent 1.4 ms on)
  extern __shared__ float wl[];                                  // [co][ci][9] of this block's channel group
  const int ci0 = blockIdx.z * CIN;
  for (int i = threadIdx.x; i < Cout * CIN * 9; i += blockDim.x) {
    const int co = i / (CIN * 9), r = i - co * (CIN * 9), c = r / 9;
    wl[i] = ci0 + c < Cin_total ? w[((long)co * Cin_total + ci0) * 9 + r] : 0.f;
  }
  __syncthreads();
  gx += (long)ci0 * H * W;
  const int nbx = (W + 1) / 2, nby = (H + 1) / 2;
  const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= (long)nbx * nby) return;
  const int i = (int)(q / nbx), j = (int)(q - (long)i * nbx);
  const int b = blockIdx.y;
  const long ohw = (long)OH * OW, hw = (long)H * W;
  const float* g = gy + (long)b * gy_bs;
  const bool v00 = i < OH && j < OW, v01 = i < OH && j + 1 < OW, v10 = i + 1 < OH && j < OW, v11 = i + 1 < OH && j + 1 < OW;
  const long o00 = (long)i * OW + j;
  float a00[CIN], a01[CIN], a10[CIN], a11[CIN];                 // gx at (2i,2j), (2i,2j+1), (2i+1,2j), (2i+1,2j+1)
#pragma unroll
  for (int c = 0; c < CIN; ++c) a00[c] = a01[c] = a10[c] = a11[c] = 0.f;
  for (int co = 0; co < Cout; ++co) {
    const float* gc = g + (long)co * ohw + o00;
    const float g00 = v00 ? gc[0] : 0.f, g01 = v01 ? gc[1] : 0.f, g10 = v10 ? gc[OW] : 0.f, g11 = v11 ? gc[OW + 1] : 0.f;
    const float* wc = wl + co * CIN * 9;
#pragma unroll
    for (int c = 0; c < CIN; ++c) {
      const float* k = wc + c * 9;                               // k[ty*3 + tx]
      a00[c] = fmaf(g00, k[4], a00[c]);
      a01[c] = fmaf(g00, k[5], fmaf(g01, k[3], a01[c]));
      a10[c] = fmaf(g00, k[7], fmaf(g10, k[1], a10[c]));
      a11[c] = fmaf(g00, k[8], fmaf(g01, k[6], fmaf(g10, k[2], fmaf(g11, k[0], a11[c]))));
    }
  }
  const int r0 = 2 * i, c0 = 2 * j;
  float* o = gx + (long)b * gx_bs + (long)r0 * W + c0;
#pragma unroll
  for (int c = 0; c < CIN; ++c) {
    if (ci0 + c >= Cin_total) break;
    float* oc = o + (long)c * hw;
    oc[0] = a00[c];
    if (c0 + 1 < W) oc[1] = a01[c];
    if (r0 + 1 < H) {
      oc[W] = a10[c];
      if (c0 + 1 < W) oc[W + 1] = a11[c];
    }
  }
}

}  // namespace

// one batch slice through the kernel family (or, in plan-only mode, only its grid.x)
static int wgrad_slice(const WgArgs& a, int k, int stride, int dil, hipStream_t st) {
  const bool halo = (k == 3 && stride == 1 && dil == 1 && a.W >= 56 && !IRR_ENV_FLAG("IRR_WGRAD_NO_HALO"));
  return halo ? dispatch_halo(a, st) : (k == 3) ? dispatch<3>(a, st) : dispatch<1>(a, st);
}

static long wgrad_batch_per(int B, int Cin, int H, int W, int Cout, long x_bs, long gy_bs) {
  // 32-bit byte offsets inside the kernel: split the batch so both views stay below 2 GiB (see OOB)
  const long lim = (1L << 29) - 64;                                  // elements
  const long bsmax = x_bs > gy_bs ? x_bs : gy_bs;
  long per = bsmax > 0 ? (lim - (long)(Cin > Cout ? Cin : Cout) * H * W) / bsmax : B;
  if (per > B) per = B;
  return per;
}

extern "C" long irr_conv2d_wgrad_ws_elems(int B, int Cin, int H, int W, int Cout, int OH, int OW, int k, int stride, int dil,
                                          long x_bs, long gy_bs) {
  if (B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0 || (k != 1 && k != 3)) return 0;
  WgArgs a{};
  a.B = B; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout; a.OH = OH; a.OW = OW;
  a.stride = stride; a.dil = dil; a.pad = ((k - 1) * dil) / 2;
  const long per = wgrad_batch_per(B, Cin, H, W, Cout, x_bs, gy_bs);
  if (per < 1) return 0;
  // grid.x is not monotone in the slice size (ceil(nchunks / max(8, ceil(nchunks / xs)))): take the maximum over the slice
  // sizes the launch loop will really use (the full slices and the remainder)
  long parts = 0;
  const int sizes[2] = {(int)per, (int)(B % per)};
  for (int i = 0; i < 2; ++i) {
    if (sizes[i] <= 0) continue;
    a.B = sizes[i];
    g_plan_only = true;
    g_parts = 0;
    const int rc = wgrad_slice(a, k, stride, dil, nullptr);
    g_plan_only = false;
    if (rc) return 0;
    if (g_parts > parts) parts = g_parts;
  }
  return parts * (long)Cout * Cin * k * k;
}

extern "C" int irr_conv2d_wgrad_f32(const float* x, const float* gy, float* gw, float* ws, float* gbias, float alpha, int B,
                                    int Cin, int H, int W,
                                    int Cout, int OH, int OW, int k, int stride, int dil, long x_bs, long gy_bs,
                                    long ws_elems, void* stream) {
  if (!x || !gy || !gw || !ws || B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0) return IRR_EINVAL;
  if ((k != 1 && k != 3) || stride < 1 || dil < 1) return IRR_EINVAL;
  WgArgs a;
  const long n = (long)Cout * Cin * k * k;
  const bool atomic = IRR_ENV_FLAG("IRR_WGRAD_ATOMIC");           // A/B switch: the earlier atomic flush into one image
  if (atomic) IRR_HIP_TRY(irr_zero_async(ws, sizeof(float) * (size_t)n, (hipStream_t)stream));
  a.x = x; a.gy = gy; a.gw = ws; a.n = atomic ? 0 : n; a.gbias = gbias; a.alpha = alpha;
  a.B = B; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout; a.OH = OH; a.OW = OW;
  a.stride = stride; a.dil = dil; a.pad = ((k - 1) * dil) / 2;
  a.x_bs = x_bs; a.gy_bs = gy_bs; a.chunks_per_block = 0;
  const long per = wgrad_batch_per(B, Cin, H, W, Cout, x_bs, gy_bs);
  if (per < 1) return IRR_EINVAL;
  for (int b0 = 0; b0 < B; b0 += (int)per) {
    a.B = (B - b0) < per ? (B - b0) : (int)per;
    a.x = x + (long)b0 * x_bs;
    a.gy = gy + (long)b0 * gy_bs;
    if (!atomic) {                                                   // the partial images of this slice must fit the caller's scratch
      g_plan_only = true;
      g_parts = 0;
      const int prc = wgrad_slice(a, k, stride, dil, nullptr);
      g_plan_only = false;
      if (prc) return prc;
      if (g_parts * n > ws_elems) return IRR_EINVAL;
    } else if (n > ws_elems) {
      return IRR_EINVAL;
    }
    const int rc = wgrad_slice(a, k, stride, dil, (hipStream_t)stream);
    if (rc) return rc;
    if (!atomic) {
      // (the scratch is reused by the next batch slice: only a single-slice launch may defer its fold)
      if (!(per >= B && irr_reduce_defer(ws, gw, n, (int)g_parts, Cin, Cout, k * k, 0))) {
        IrrReduceJob J{};
        J.ws = ws; J.gw = gw; J.n = n; J.P = (int)g_parts; J.Cin = Cin; J.Cout = Cout; J.KK = k * k;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(irr_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, J);
        IRR_LAUNCH_CHECK();
      }
    }
  }
  if (atomic) {
    IrrReduceJob J{};
    J.ws = ws; J.gw = gw; J.n = n; J.P = 1; J.Cin = Cin; J.Cout = Cout; J.KK = k * k;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(irr_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, J);
    IRR_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int irr_lrelu_bwd_bias_f32(const float* gy, const float* y, float* gpre, float* gbias, int B, int C, int HW,
                                      long gy_bs, long y_bs, long gpre_bs, int lrelu, float* amax, void* stream) {
  if (!gy || B <= 0 || C <= 0 || HW <= 0 || (lrelu && !y) || B > 65535 || C > 65535) return IRR_EINVAL;
  if (!gpre && !gbias && !amax) return 0;
  int chunk = 4096;
  dim3 grid(irr_cdiv(HW, chunk), C, B);
  hipLaunchKernelGGL(lrelu_bwd_bias_kernel, grid, dim3(256), 0, (hipStream_t)stream, gy, y, gpre, gbias, C, (long)HW,
                     gy_bs, y_bs, gpre_bs, lrelu, chunk, amax);
  IRR_LAUNCH_CHECK();
  return 0;
}

extern "C" int irr_conv2d_dgrad_strided_f32(const float* gy, const float* w, float* gx, int B, int Cin, int H, int W,
                                            int Cout, int OH, int OW, int k, int stride, int dil, long gy_bs,
                                            long gx_bs, void* stream) {
  if (!gy || !w || !gx || B <= 0 || Cin <= 0 || Cout <= 0 || B > 65535 || Cin > 65535) return IRR_EINVAL;
  if ((k != 1 && k != 3) || stride < 1 || dil < 1) return IRR_EINVAL;
  if (k == 3 && stride == 2 && dil == 1 && Cout <= 256 && OH == (H + 1) / 2 && OW == (W + 1) / 2 && Cin <= 65535 * 4) {
    const long nblk = irr_cdiv((long)((H + 1) / 2) * ((W + 1) / 2), 256);
    if (Cin <= 3) {
      dim3 g2((unsigned)nblk, B, 1);
      const size_t lds = sizeof(float) * (size_t)Cout * Cin * 9;
      switch (Cin) {
        case 1: hipLaunchKernelGGL(dgrad_s2k3_smallci_kernel<1>, g2, dim3(256), lds, (hipStream_t)stream, gy, w, gx, H, W, Cout, OH, OW, gy_bs, gx_bs, Cin); break;
        case 2: hipLaunchKernelGGL(dgrad_s2k3_smallci_kernel<2>, g2, dim3(256), lds, (hipStream_t)stream, gy, w, gx, H, W, Cout, OH, OW, gy_bs, gx_bs, Cin); break;
        default: hipLaunchKernelGGL(dgrad_s2k3_smallci_kernel<3>, g2, dim3(256), lds, (hipStream_t)stream, gy, w, gx, H, W, Cout, OH, OW, gy_bs, gx_bs, Cin); break;
      }
    } else {
      dim3 g2((unsigned)nblk, B, irr_cdiv(Cin, 4));
      hipLaunchKernelGGL(dgrad_s2k3_smallci_kernel<4>, g2, dim3(256), sizeof(float) * (size_t)Cout * 4 * 9, (hipStream_t)stream, gy, w, gx,
                         H, W, Cout, OH, OW, gy_bs, gx_bs, Cin);
    }
    IRR_LAUNCH_CHECK();
    return 0;
  }
  dim3 grid(irr_cdiv((long)H * W, 256), Cin, B);
  hipLaunchKernelGGL(dgrad_strided_kernel, grid, dim3(256), 0, (hipStream_t)stream, gy, w, gx, Cin, H, W, Cout, OH, OW,
                     k, stride, dil, ((k - 1) * dil) / 2, gy_bs, gx_bs);
  IRR_LAUNCH_CHECK();
  return 0;
}
