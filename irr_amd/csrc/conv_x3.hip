// fp32-faithful 3x3 stride-1 convolution on the bf16 matrix pipe of gfx950 (IRR-PWC conv() blocks,
// models/pwc_modules.py:8-19, models/irr_modules.py:7-18; forward and stride-1 data gradient).
//
// Why: the fp32 MFMA (v_mfma_f32_32x32x2_f32) runs at 1/16 of the bf16 MFMA rate (157 vs 2500 TFLOP/s) and the
// fp32 direct-conv kernel of conv_fwd.hip already sits at 80-89 % of that peak.  Here every fp32 operand is split
// EXACTLY into three bf16 pieces, x = hi + mid + lo (round-to-nearest each time: |mid| <= 2^-8 |x|,
// |lo| <= 2^-17 |x|, and the 24-bit significand is covered completely), and the product a*b is accumulated in
// fp32 from the six piece products of weight >= 2^-17:
//     a*b ~= ah*bh + ah*bm + am*bh + am*bm + ah*bl + al*bh          (dropped: am*bl + al*bm + al*bl <= 2^-24 |ab|)
// Each piece product is exact in fp32 (8x8-bit significands), so the result carries the same error class as an
// fp32 FMA chain (one rounding of relative size <= 2^-24 per product) at 6/16 of the fp32-MFMA cost: an effective
// fp32 roof of 2500/6 = 417 TFLOP/s.
//
// Formulation: D[co][px] += W[co][k] * X[k][px] with v_mfma_f32_32x32x16_bf16; k = 16 input channels of one tap.
//   * block = CT co-tile waves x PG pixel-group waves; a wave owns ONE 32-channel co-tile x NT sub-tiles of 32 pixels
//     (NT*16 accumulator VGPRs).  The block's pixels form a TR x TC tile of one sample.
//   * per 16-channel chunk the (TR+2d) x (TC+2d) halo patch is loaded ONCE (coalesced dword loads, out-of-image
//     positions answered with 0 by the buffer bounds check), split into the three pieces in registers and written
//     to LDS as six planes [piece][k-group g][pixel] of 16 B (= the 8 bf16 k-values a lane feeds to the MFMA), so the
//     nine taps are nine shifted conflict-free ds_read_b128 views of the same patch: the split costs 1/9 of a
//     per-tap split and no activation is re-read from L1/L2 per tap.
//   * weights are pre-split at pack time (irr_conv_pack_weights_x3): one coalesced 1 KiB buffer_load_dwordx4 per
//     (chunk, tap, piece, co-tile) is exactly the A fragment; prefetched one tap ahead.
//   * epilogue identical to conv_fwd.hip (bias, LeakyReLU, residual/scale, "+=", LeakyReLU'-mask).
#include "x3_split.h"
#include "pack.h"
#include "amax.h"
#include <stdlib.h>
#include <atomic>

#ifdef X3S_TRACE
static unsigned long long* g_x3s_dbg = nullptr;
#endif

#ifndef X3_ABL
#define X3_ABL 0     // ablation builds (timing only, results wrong): 1 = no weight loads in the loop, 2 = no LDS reads in the loop,
#endif               // 3 = producers skip their global loads, 4 = independent accumulators (no dependent MFMA chain)

int irr_amax_channels_launch(const float* x, int B, int C, long hw, long bs, float* out, hipStream_t st, bool zero);      // misc.hip

namespace {

constexpr uint32_t OOB = 0x80000000u;     // voffset marker: beyond num_records -> the load returns 0, touches nothing

struct X3Args {
  const float* x;
  const u32x4* wq;
  const float* bias;
  const float* res;
  float* y;
  int B, Cin, H, W, Cout;
  int dil;
  int RD;                                  // row fold: 1, or dil (the rows y = r (mod dil) of a sample form an independent dil-1-in-y problem)
  int CoT, nchunk;
  int TR, TC, tiles_x, tiles_y;
  long x_bs, y_bs, res_bs;
  int lrelu, accumulate;
  float alpha;
  const float* mask;
  long mask_bs;
  int nmask;
  int ngy;                                 // co-tile groups per pixel tile (folded into the 1-D x grid, see the kernel)
  int ksplit;                              // > 1 (small pyramid levels): blockIdx.z walks a slice of the 16-channel chunks and
  float* part;                             // stores raw partial sums part[kz][b][co][pixel]; x3_splitk_epilogue_kernel finishes
  // NP == 2 (the fp16x2 "h2" form, x3_split.h) only:
  const float* x_amax;                     // max |x| of the input tensor = max over n_amax device slots -> the operand scale
  int n_amax;
  float* y_amax;                           // nullable: slot that receives max |y| of this launch's output (atomic max on the bit pattern)
  float* y_chmax;                          // nullable (round 6): Cout slots that receive max |y[:, co]| PER OUTPUT CHANNEL (irr_conv_x3_next_chmax)
  unsigned int* kcnt;                      // nullable (end of round 6, ksplit > 1): one ZEROED counter per block of the x grid -- the block that
                                           // arrives LAST at a pixel tile sums the slices' partial images in slice order and runs the epilogue
                                           // itself (no finishing launch); the counters wrap back to zero (atomicInc)
};

// Block = CT*PG symmetric waves, two blocks per CU (256 registers per wave, 128 of them accumulators): every wave
// takes part in staging a chunk's patch (load -> split -> LDS), then runs its 9 x NT x 6 MFMAs; while one block
// stages, the other block on the CU keeps the matrix pipe busy.  (A producer/consumer-specialised variant with one
// 8-wave block per CU and double-buffered LDS was built and measured: 5-20 % slower on every layer shape, because
// nothing overlaps a block's prologue/epilogue there; on random data both variants run into the same power-limited
// clock: all-zero operands run 32 % faster than random ones on the same launch.)
// NP = 3: bf16x3, six products (above).  NP = 2: fp16x2 of operands scaled by a power of two, three products (x3_split.h, "h2"):
// the activations are scaled by 2^ex from a.x_amax, the packed weights carry their own exponent in the 16-B unit behind the
// last fragment, and the accumulators are scaled back before the epilogue.
template <int CT, int PG, int NT, int PLANE_PIX, int NP = 3>
__global__ __launch_bounds__(CT* PG * 64, 2) void conv_x3_kernel(const X3Args a) {
  constexpr int NTHR = CT * PG * 64;
  constexpr int NR = (2 * PLANE_PIX + NTHR - 1) / NTHR;      // staging rounds: one (pixel, k-group) unit per thread per round
  __shared__ u32x4 lds[2 * NP * PLANE_PIX];                 // [piece][g][pixel] x 16 B

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ct = wave % CT, pg = wave / CT;
  const int j = lane & 31, g = lane >> 5;
  // XCD-major block order (common.h): blocks of one XCD take CONSECUTIVE tiles (their halos overlap in that XCD's L2), and
  // the co-tile groups of one pixel tile (blockIdx.y, layers with more co-tiles than CT) stay on one XCD: they read the
  // same input patch.  K-split slices (blockIdx.z) read different channels and are left alone.
  // (launched with a ONE-dimensional x grid of ntiles * ngy blocks, like the weight-gradient kernel: the XCD placement of a
  // workgroup is only observed for the linear id of a 1-D grid.  Open: the 128 -> 565 data gradient -- five co-tile groups per
  // pixel tile -- still fetches 2.7 GB per launch for a 0.35 GB input (3.0 GB with the groups in gridDim.y): the groups do not
  // share the patch through L2 although they are dispatched back to back on one XCD.  That launch runs at the power wall
  // (232 TFLOP/s, 0.6 TB/s), so this was left; profiles/r3_traffic_dgrad_128to565.txt.)
  const unsigned xpos = irr_xcd_order(blockIdx.x, gridDim.x);
  const int by = (int)(xpos % (unsigned)a.ngy);
  int bt = (int)(xpos / (unsigned)a.ngy);
  const int tx = bt % a.tiles_x;
  bt /= a.tiles_x;
  const int ty = bt % a.tiles_y;
  bt /= a.tiles_y;
  const int rr = bt % a.RD;                                 // row residue class of this block (0 when rows are not folded)
  const int b = bt / a.RD;
  const int y0 = ty * a.TR, x0 = tx * a.TC;                 // y0 counts rows of the residue class: image row = rr + RD * y
  const int d = a.dil;
  const int dy = a.RD > 1 ? 1 : d;                          // vertical tap distance in rows of the class
  const int Hs = (a.H - rr + a.RD - 1) / a.RD;
  const int LW = a.TC + 2 * d, LH = a.TR + 2 * dy;
  const int npix = LH * LW;
  const long hw = (long)a.H * a.W;
  const int cot = by * CT + ct;
  const bool active = cot < a.CoT;

  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, (short)0, (int)0x80000000u, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)a.wq, (short)0, (int)0xffffffffu, 0x00020000);

  // ---- staging roles ----
  uint32_t svoff[NR];
  int swidx[NR];                                            // LDS index (16-B units) of the unit's hi piece, -1 = none
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int u = r * NTHR + tid;
    const int gg = u >= npix ? 1 : 0;
    const int pix = u - gg * npix;
    const bool inr = u < 2 * npix;
    const int ly = pix / LW, lx = pix - ly * LW;
    const int iy = y0 - dy + ly, ix = x0 - d + lx;
    const bool ok = inr && iy >= 0 && iy < Hs && ix >= 0 && ix < a.W;
    svoff[r] = ok ? (uint32_t)(((long)b * a.x_bs + (long)gg * 8 * hw + (long)(rr + iy * a.RD) * a.W + ix) * 4) : OOB;
    swidx[r] = inr ? gg * PLANE_PIX + pix : -1;
  }
  const uint32_t hw4 = (uint32_t)(hw * 4);
  const int tail_base = a.Cin - 16;                        // last chunk re-reads [Cin-16, Cin) (duplicates have zero weights)

  // ---- compute roles ----
  int xidx[NT];                                            // LDS index (16-B units) of the lane's pixel, tap (0,0), piece hi
#pragma unroll
  for (int s = 0; s < NT; ++s) {
    const int t = (pg * NT + s) * 32 + j;
    const int row = t / a.TC, col = t - row * a.TC;
    xidx[s] = g * PLANE_PIX + row * LW + col;
  }
  const uint32_t wvoff = (uint32_t)((cot * 64 + lane) * 16);
  const uint32_t wpiece = (uint32_t)a.CoT * 1024u;         // bytes between pieces
  const uint32_t wtap = (uint32_t)NP * wpiece;
  float sx = 1.f, inv_x = 1.f, inv_w = 1.f;                // h2: operand scale of x, and the two factors that undo both scales
  if (NP == 2) {
    const int ex = x3_h2_exp(x3_h2_amax(a.x_amax, a.n_amax));
    const int ew = ((const int*)(a.wq + (long)a.nchunk * 9 * NP * a.CoT * 64))[0];
    sx = ldexpf(1.f, ex);
    inv_x = ldexpf(1.f, -ex);
    inv_w = ldexpf(1.f, -ew);
  }

  f32x16 acc[NT];
#pragma unroll
  for (int s = 0; s < NT; ++s)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[s][r] = 0.f;

  float raw[NR][8];
  auto issue_x = [&](int c) {
    const int ch0 = X3_ABL == 12 ? 0 : (c == a.nchunk - 1) ? tail_base : c * 16;       // ablation 12: every chunk re-reads chunk 0 (cache hits, real data)
    const uint32_t s0 = (uint32_t)ch0 * hw4;
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
      for (int e = 0; e < 8; ++e)
        raw[r][e] = (X3_ABL == 3 && c > 0) ? 1.f : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, (int)svoff[r], (int)(s0 + e * hw4), 0));
  };
  u32x4 wa[2][NP];
  auto issue_w = [&](int slot, int c, int tap) {
    const uint32_t so = ((uint32_t)c * 9u + (uint32_t)tap) * wtap;
#pragma unroll
    for (int p = 0; p < NP; ++p)
      wa[slot][p] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wr, (int)wvoff, (int)(so + p * wpiece), 0));
  };
  const int c_begin = (int)(((long)blockIdx.z * a.nchunk) / a.ksplit), c_end = (int)(((long)(blockIdx.z + 1) * a.nchunk) / a.ksplit);
  if (active) issue_w(0, c_begin, 0);

  for (int c = c_begin; c < c_end; ++c) {
    // load the chunk's halo patch, split it into its three bf16 pieces and publish it.  (No register prefetch across
    // the MFMA phase: 128 of the wave's 256 registers are accumulators; the second block on the CU covers the wait.)
    issue_x(c);
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      u32x4 h, m, l;
      if (NP == 2) split8_h2<true>(raw[r], sx, h, m);     // low piece x 2^11 (x3_split.h, "Range")
      else split8(raw[r], h, m, l);
      if (swidx[r] >= 0) {
        lds[swidx[r]] = h;
        lds[swidx[r] + 2 * PLANE_PIX] = m;
        if (NP == 3) lds[swidx[r] + 4 * PLANE_PIX] = l;
      }
    }
    __syncthreads();
    const bool more = c + 1 < c_end;
    if (active) {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int slot = tap & 1;
        // weights of the next tap (next chunk's tap 0 after the last one; clamped re-read at the very end)
        if (X3_ABL != 1) {
          const int nt = tap == 8 ? 0 : tap + 1;
          const int nc = tap == 8 ? (more ? c + 1 : c) : c;
          issue_w(slot ^ 1, nc, nt);
        }
        __builtin_amdgcn_sched_barrier(0);
        const int toff = (tap / 3) * LW * dy + (tap % 3) * d;
        u32x4 wdn;                                          // NP == 2: this tap's high weight piece times 2^-11 (pairs with the scaled-up low piece of x)
        if (NP == 2) wdn = h2_hi_down(wa[slot][0]);
        u32x4 xb[2][NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) xb[0][p] = lds[xidx[0] + toff + 2 * p * PLANE_PIX];
#pragma unroll
        for (int s = 0; s < NT; ++s) {
          const int cur = s & 1;
          if (s + 1 < NT) {
#pragma unroll
            for (int p = 0; p < NP; ++p) xb[cur ^ 1][p] = lds[xidx[s + 1] + toff + 2 * p * PLANE_PIX];
          }
          __builtin_amdgcn_sched_barrier(0);        // keep the next sub-tile's LDS reads AHEAD of this one's six MFMAs
          f32x16 t = acc[s];
          if constexpr (NP == 2) {
            t = mma_h(wa[slot][1], xb[cur][0], t);    // lo * hi
            t = mma_h(wdn, xb[cur][1], t);            // (hi * 2^-11) * (lo * 2^11)
            t = mma_h(wa[slot][0], xb[cur][0], t);    // hi * hi
          } else {
            if (X3_ABL != 20) {                       // ablation 20 (timing only): three products of two pieces
            t = mma(wa[slot][NP - 1], xb[cur][0], t); // lo * hi
            t = mma(wa[slot][0], xb[cur][NP - 1], t); // hi * lo
            t = mma(wa[slot][1], xb[cur][1], t);      // mid * mid
            }
            t = mma(wa[slot][1], xb[cur][0], t);      // mid * hi
            t = mma(wa[slot][0], xb[cur][1], t);      // hi * mid
            t = mma(wa[slot][0], xb[cur][0], t);      // hi * hi
          }
          acc[s] = t;
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // nine taps flip the slot parity: hand the prefetched (c+1, tap 0) fragments back to slot 0
#pragma unroll
      for (int p = 0; p < NP; ++p) wa[0][p] = wa[1][p];
    }
    __syncthreads();
  }
  // (the in-launch K split below exists for NT <= 4 only -- the tiles the planner gives the small pyramid levels: in the register-tight
  // NT = 7 / 8 instantiations its mere presence cost the level-4 launches 8-15 %; for them the code is what it was)
  if constexpr (NT > 4) {
    if (!active) return;
  }
  if (active && NP == 2) {                   // back to the operands' own scale (two exact power-of-two factors)
#pragma unroll
    for (int s = 0; s < NT; ++s)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[s][r] = (acc[s][r] * inv_x) * inv_w;
  }
  const long ohw = hw;
  bool split_out = a.ksplit > 1;             // this block stores a raw partial image and somebody else finishes
  if constexpr (NT <= 4)
  if (a.ksplit > 1 && a.kcnt != nullptr) {   // (uniform) K split finished IN THE LAUNCH: every wave of the block takes part in the barriers
    // The slices of a pixel tile are the blocks (blockIdx.x, z = 0 .. ksplit-1).  Each stores its partial image at agent scope (the eight
    // XCDs' L2s are not coherent with each other for plain stores), waits for the stores and counts itself; the block that counts LAST
    // replaces its accumulators by the sum of all slices IN SLICE ORDER (the finishing kernel's order: results
    // are bit-identical to the two-launch route, whichever block happens to be last) and falls through to the ordinary epilogue.
    if (active) {
#pragma unroll
      for (int s = 0; s < NT; ++s) {
        const int t = (pg * NT + s) * 32 + j;
        const int row = t / a.TC, col = t - row * a.TC;
        const int oys = y0 + row, ox = x0 + col;
        if (oys >= Hs || ox >= a.W) continue;
        float* pb = a.part + (((long)blockIdx.z * a.B + b) * a.Cout) * ohw + (long)(rr + oys * a.RD) * a.W + ox;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = cot * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
          // (agent-scope atomic store: written THROUGH this XCD's L2 -- a release FENCE instead would write back the whole L2 of the XCD,
          // and the acquire on the other side invalidate one: measured +4.8 ms per step with the lane's kernels sharing those caches)
          if (co < a.Cout) __hip_atomic_store(pb + (long)co * ohw, acc[s][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    __shared__ unsigned int s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's partial image has reached the device's coherence point
    __syncthreads();
    if (tid == 0) {
      const unsigned int old = atomicInc(a.kcnt + blockIdx.x, (unsigned int)a.ksplit - 1u);     // wraps to 0 behind the last arrival
      s_last = old == (unsigned int)a.ksplit - 1u ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last) return;
    // (the slices are read with agent-scope atomic loads below: served from the coherence point, never from a stale line of this XCD's L2)
    if (active) {
#pragma unroll
      for (int s = 0; s < NT; ++s) {
        const int t = (pg * NT + s) * 32 + j;
        const int row = t / a.TC, col = t - row * a.TC;
        const int oys = y0 + row, ox = x0 + col;
        if (oys >= Hs || ox >= a.W) continue;
        const long pofs0 = ((long)b * a.Cout) * ohw + (long)(rr + oys * a.RD) * a.W + ox;
        const long kstride = (long)a.B * a.Cout * ohw;
        // slice by slice, the sixteen rows of the sub-tile in flight together (one row after the other, slice after slice, was a chain
        // of 16 * NT * ksplit dependent round trips for the one block that finishes a tile: +2.7 ms per step)
        f32x16 sum;
#pragma unroll
        for (int r = 0; r < 16; ++r) sum[r] = 0.f;
        for (int k = 0; k < a.ksplit; ++k) {
          float tv[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int co = cot * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
            const int cc = co < a.Cout ? co : a.Cout - 1;    // (clamped: the row of a channel past the end is never stored)
            tv[r] = __hip_atomic_load(a.part + pofs0 + (long)cc * ohw + (long)k * kstride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) sum[r] += tv[r];
        }
        acc[s] = sum;
      }
    }
    split_out = false;
  }
  if (!active) return;
  float ymax = 0.f;
  const bool want_amax = NP == 2 && a.y_amax != nullptr && !split_out;
  const bool want_ch = NP == 2 && a.y_chmax != nullptr && !split_out;
  float chm[16];                                           // max |stored value| of this lane per accumulator row (= output channel)
#pragma unroll
  for (int r = 0; r < 16; ++r) chm[r] = 0.f;

  // ---- epilogue: D[i][jj], i = (r&3) + 8*(r>>2) + 4*g, jj = lane&31 ----
  // per-sample bases: the byte offsets inside a sample stay below the 2 GiB out-of-range marker for any batch
  const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)(a.y + (long)b * a.y_bs), (short)0, (int)0x80000000u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsr =
      __builtin_amdgcn_make_buffer_rsrc((void*)(a.res ? a.res + (long)b * a.res_bs : a.y), (short)0, (int)0x80000000u, 0x00020000);
  const __amdgpu_buffer_rsrc_t mkr =
      __builtin_amdgcn_make_buffer_rsrc((void*)(a.mask ? a.mask + (long)b * a.mask_bs : a.y), (short)0, (int)0x80000000u, 0x00020000);
#pragma unroll
  for (int s = 0; s < NT; ++s) {
    const int t = (pg * NT + s) * 32 + j;
    const int row = t / a.TC, col = t - row * a.TC;
    const int oys = y0 + row, ox = x0 + col;
    if (oys >= Hs || ox >= a.W) continue;
    const long pofs = (long)(rr + oys * a.RD) * a.W + ox;
    if (split_out) {                         // raw partial sums of this chunk slice; the epilogue runs in a second kernel
      float* pb = a.part + (((long)blockIdx.z * a.B + b) * a.Cout) * ohw + pofs;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = cot * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
        if (co < a.Cout) pb[(long)co * ohw] = acc[s][r];
      }
      continue;
    }
    // Read-modify-write epilogues (residual, "+=", LeakyReLU'-mask): the operands of the 16 rows of a sub-tile are loaded
    // as ONE batch of buffer loads (out-of-range voffset where an operand or a row does not exist) before the first store --
    // with pointer loads inside the row loop every row waited for its own round trip to memory (the K <= 128 data gradients
    // of the DenseNet columns ran at 55-110 TFLOP/s because of it).
    constexpr int RB = NT >= 8 ? 8 : 16;              // rows per batch (NT = 8 has no registers left for 16)
#pragma unroll
    for (int r0 = 0; r0 < 16; r0 += RB) {
      uint32_t vo[RB];
      float bv[RB];
#pragma unroll
      for (int k = 0; k < RB; ++k) {
        const int r = r0 + k;
        const int co = cot * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
        const bool ok = co < a.Cout;
        vo[k] = ok ? (uint32_t)(((long)co * ohw + pofs) * 4) : OOB;
        bv[k] = (ok && a.bias) ? a.bias[co] : 0.f;
      }
      if (!a.res && !a.accumulate && !a.mask) {
#pragma unroll
        for (int k = 0; k < RB; ++k) {
          float v = acc[s][r0 + k] + bv[k];
          if (a.lrelu) v = irr_lrelu(v);
          v *= a.alpha;
          // (one fold per stored value: with channel maxima wanted the tensor's maximum is taken from them at the end)
          if (want_ch) { if (vo[k] != OOB) chm[r0 + k] = x3_amax_fold(chm[r0 + k], v); }
          else if (want_amax && vo[k] != OOB) ymax = x3_amax_fold(ymax, v);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), yr, (int)vo[k], 0, 0);
        }
        continue;
      }
      float rv[RB], dv[RB], mv[RB];
#pragma unroll
      for (int k = 0; k < RB; ++k) {
        const int r = r0 + k;
        const int co = cot * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
        rv[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsr, (int)(a.res ? vo[k] : OOB), 0, 0));
        dv[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(yr, (int)(a.accumulate ? vo[k] : OOB), 0, 0));
        mv[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(mkr, (int)((a.mask && co < a.nmask) ? vo[k] : OOB), 0, 0));
      }
#pragma unroll
      for (int k = 0; k < RB; ++k) {
        const int r = r0 + k;
        const int co = cot * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
        float v = acc[s][r] + bv[k];
        if (a.lrelu) v = irr_lrelu(v);
        v = a.res ? rv[k] + a.alpha * v : v * a.alpha;
        v += dv[k];                                   // 0 unless accumulating
        if (a.mask && co < a.nmask) v *= irr_lrelu_grad(mv[k]);
        if (want_ch) { if (vo[k] != OOB) chm[r] = x3_amax_fold(chm[r], v); }
        else if (want_amax && vo[k] != OOB) ymax = x3_amax_fold(ymax, v);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), yr, (int)vo[k], 0, 0);
      }
    }
  }
  if (want_ch && want_amax) {
#pragma unroll
    for (int r = 0; r < 16; ++r) ymax = __builtin_bit_cast(float, max(__builtin_bit_cast(uint32_t, ymax), __builtin_bit_cast(uint32_t, chm[r])));
  }
  if (want_amax) x3_amax_publish(ymax, a.y_amax);
  if (want_ch) {
    // the 32 lanes of a half-wave hold the 32 pixels of one accumulator row each: fold them (DPP, amax.h); then lane j = 16 + r of each half
    // looks at the slot of row r and raises it when it has to -- sixteen lanes, one round trip (sixteen dependent look-then-atomic sequences at the
    // end of every wave cost these launches 8 %)
    uint32_t mine = 0u;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const uint32_t mb = x3_amax_half_upper(__builtin_bit_cast(uint32_t, chm[r]));      // (valid in lanes j >= 16 of each half)
      mine = j == 16 + r ? mb : mine;
    }
    if (j >= 16) {
      const int r = j - 16;
      const int co = cot * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
      if (co < a.Cout) x3_amax_commit(__builtin_bit_cast(float, mine), a.y_chmax + co);
    }
  }
}

static thread_local float* g_next_chmax = nullptr;   // (irr_conv_x3_next_chmax)
static std::atomic<int> g_min_blocks{384};     // (the ONE process-wide routing policy, see irr_conv_x3_set_min_blocks) launches with fewer blocks cannot fill 256 CUs x 2 blocks: they stay on the fp32 kernels

// ---------------------------------------------------------------------------------------------------------------
// Streaming variant for the 32-channel layers (OccUpsampleNetwork at 1/2 and full resolution, models/irr_modules.py:30-56:
// seven 32 -> 32 convs per call, 41 % of all conv activation traffic).  With K = 9 * 32 the MFMA work of a tile
// (2.9 us) is SHORTER than its HBM time (76-108 KiB per 256-pixel tile = 3.9-5.5 us at 5 TB/s / 256 CUs), so the
// kernel is organised around keeping loads in flight: PERSISTENT blocks of 4 consumer + 4 producer waves walk the
// tiles; the producers stage the complete 32-channel halo patch of tile n+1 (both 16-channel chunks, split to bf16x3)
// into the second LDS buffer and already have tile n+2's global loads in flight while the consumers run the
// 18 x 12 MFMAs of tile n and write it out (epilogue operands -- residual / accumulate / mask -- are prefetched at the
// start of the tile).  One s_barrier per tile.  Tile = 8 rows x 32 columns; Cout <= 32, 16 < Cin <= 32, dilation 1.
struct X3SArgs {
  const float* x;
  const u32x4* wq;
  const float* bias;
  const float* res;
  float* y;
  int B, Cin, H, W, Cout;
  int tiles_x, tiles_y;
  long ntiles;
  long x_bs, y_bs, res_bs;
  int lrelu, accumulate;
  float alpha;
  const float* mask;
  long mask_bs;
  int nmask;
  unsigned long long* dbg;      // X3S_TRACE builds only
  int wCoT, wcot;               // co-tiles in the packed weights / the one this launch computes
  float* y2;                    // EPI 3: second output, the value before the residual is added
  long y2_bs;
  const float* x_amax;          // NP == 2: as in X3Args
  int n_amax;
  float* y_amax;
  float* y_chmax;               // nullable (round 6, NP == 2): this launch's Cout slots for max |y[:, co]| PER OUTPUT CHANNEL (irr_conv_x3_next_chmax)
  // LeakyReLU' masks as BITS (round 5): one 32-bit word per epilogue thread and tile -- bit e * 4 + px = (stored value > 0) of channel
  // eq_c8 * 8 + e, pixel px of the thread's quad -- at word (bits_tile0 + tile) * 256 + ptid.  A forward launch (EPI 0) writes them
  // next to its output; the masked data gradient of the SAME map shape (EPI 4) reads one dword instead of eight 16-B loads of the
  // activation: these launches are bound by HBM bytes and by the producers' memory-instruction rate.
  const uint32_t* mask_bits;
  uint32_t* bits_out;
  long bits_tile0;
};

// s_memtime trace points (IRR_X3S_TRACE=1 builds, tools/x3s_trace.py): block 7, lane 0 of every wave
#ifdef X3S_TRACE
#define TR(slot) do { if (blockIdx.x == 7 && lane == 0 && ntr < 400) { dbgp[ntr++] = ((unsigned long long)(slot) << 56) | (__builtin_amdgcn_s_memtime() & 0xffffffffffffffull); } } while (0)
#else
#define TR(slot) do {} while (0)
#endif

#ifndef X3S_PRODUCERS_OLD
#define X3S_PRODUCERS_OLD 1   // 1: waves 0..3 produce, 4..7 run the MFMAs; 0 (A/B): the other way round
#endif
// EPI: what the epilogue has to read besides the accumulators -- 0: nothing (bias, LeakyReLU, alpha), 1: + a residual operand,
// 2: everything (residual, accumulate-into-output, LeakyReLU'-mask), 3: residual + a SECOND output y2 = the value before the
// residual is added (y = res + y2; irr_conv2d_fwd_x3_dual), 4: as 2 with the mask read as bits (X3SArgs::mask_bits).  The kernel is bound by the producer waves' VALU issue
// slots (operand split + epilogue), so the plain layers do not pay for 24 operand loads and 5 unused VALU per output.
// NP: pieces per operand -- 3: bf16x3 (six products), 2: the fp16x2 form of x3_split.h (three products; operands scaled by the
// powers of two derived from a.x_amax and the weight pack's trailer, accumulators scaled back when they are handed over).
template <int EPI, int NP = 3>
__global__ __launch_bounds__(512) void conv_x3s_kernel(const X3SArgs a) {
  constexpr int PLANE_PIX = 352;                           // >= 10 x 34 halo patch
  constexpr int CHUNK = 2 * NP * PLANE_PIX;                // 16-B units per chunk slot: [piece][g][pixel]
  constexpr int LW = 34;
  constexpr int WUNITS = 18 * NP * 64;                     // the complete pre-split weight set: [step][piece][lane] x 16 B
  extern __shared__ u32x4 lds[];
  u32x4* const wl = lds;                                   // weights (54 / 36 KiB), loaded once per block
  u32x4* const xl = lds + WUNITS;                          // two chunk slots (66 / 44 KiB)
  float* const ol = (float*)(lds + WUNITS + 2 * CHUNK);    // accumulators of the finished tile: [32 co][256 px] fp32 (32 KiB)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long hw = (long)a.H * a.W;
  const long t_begin = irr_xcd_order(blockIdx.x, gridDim.x), t_step = gridDim.x;     // blocks of one XCD walk neighbouring tiles (shared halos)
#ifdef X3S_TRACE
  int ntr = 0;
  unsigned long long* dbgp = a.dbg + (size_t)wave * 400;
#endif

  // The MFMA waves touch no global memory at all: weights live in LDS for the lifetime of the (persistent) block, and
  // finished accumulators are handed to the producer waves through LDS, which run the epilogue (bias, LeakyReLU,
  // residual, accumulate, mask, store) of tile n while the MFMA waves are already on tile n+1.
  // (the packed weights interleave the co-tiles of a layer: this launch keeps co-tile a.wcot of a.wCoT)
  for (int u = tid; u < WUNITS; u += 512) wl[u] = a.wq[((long)(u >> 6) * a.wCoT + a.wcot) * 64 + (u & 63)];
  float sx = 1.f, inv_x = 1.f, inv_w = 1.f;                // h2: operand scale of x, and the two factors that undo both scales
  if (NP == 2) {
    const int ex = x3_h2_exp(x3_h2_amax(a.x_amax, a.n_amax));
    const int ew = ((const int*)(a.wq + (long)18 * NP * a.wCoT * 64))[0];
    sx = ldexpf(1.f, ex);
    inv_x = ldexpf(1.f, -ex);                              // (applied one after the other, as in conv_x3_kernel: |ex + ew| can reach 192,
    inv_w = ldexpf(1.f, -ew);                              //  beyond the fp32 exponent range -- ONE factor 2^-(ex+ew) flushed to zero for tiny tensors)
  }
  __syncthreads();

  // Roles by wave AGE: the instruction arbiter of a SIMD prefers its older wave, and a wave that streams MFMAs leaves the
  // other one almost no issue slots (tools/wx3_trace.py shows the same in the weight-gradient kernel).  The producers are the
  // bottleneck of a tile and the MFMA waves have slack at the barriers, so the producers take the OLDER slots (waves 0..3).
  if (X3S_PRODUCERS_OLD ? wave < 4 : wave >= 4) {
    // ================= producers (+ epilogue) =================
    const int ptid = X3S_PRODUCERS_OLD ? tid : tid - 256;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, (short)0, (int)0x80000000u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc((void*)a.res, (short)0, (int)0x80000000u, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, (short)0, (int)0x80000000u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rmask = __builtin_amdgcn_make_buffer_rsrc((void*)a.mask, (short)0, (int)0x80000000u, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry2 = __builtin_amdgcn_make_buffer_rsrc((void*)(EPI == 3 ? a.y2 : a.y), (short)0, (int)0x80000000u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rbits = __builtin_amdgcn_make_buffer_rsrc((void*)(EPI == 4 ? (const void*)a.mask_bits : (const void*)a.bits_out),
                                                                            (short)0, (int)0x80000000u, 0x00020000);
    // Staging unit = (k-group gg, patch row ly, aligned pixel quad q): 8 channels x 4 pixels as eight 16-B loads (the
    // vector-memory INSTRUCTION rate, not bandwidth, limited the dword version of this kernel).  The loaded window is
    // columns x0-4 .. x0+35 (ten aligned quads; only x0-1 and x0+32 of the two margin quads are used), so every
    // quad is entirely inside or entirely outside the image.  200 units per chunk, one per producer thread.
    const int su_gg = ptid / 100, su_rest = ptid % 100;
    const int su_ly = su_rest / 10, su_q = su_rest % 10;
    const bool su_act = ptid < 200;
    const uint32_t hw4 = (uint32_t)(hw * 4);
    const uint32_t c1off = (uint32_t)(a.Cin - 16) * hw4;   // second chunk = channels [Cin-16, Cin) (duplicates: zero weights)
    f32x4 raw[2][8];
    // Tile coordinates are stepped, not decoded: three 64-bit divisions per call (twice per tile here, once more in the
    // epilogue) were 12 % of the producers' critical path.  t advances by gridDim.x = (sb, sy, sx) in mixed radix.
    struct TileAt { int b, ty, tx; };
    const int step_x = (int)(t_step % a.tiles_x), step_y = (int)((t_step / a.tiles_x) % a.tiles_y),
              step_b = (int)(t_step / ((long)a.tiles_x * a.tiles_y));
    auto advance = [&](TileAt& c) {
      c.tx += step_x;
      const int cx = c.tx >= a.tiles_x;
      c.tx -= cx ? a.tiles_x : 0;
      c.ty += step_y + cx;
      const int cy = c.ty >= a.tiles_y;
      c.ty -= cy ? a.tiles_y : 0;
      c.b += step_b + cy;
    };
    // (`valid` = false: the same eight loads with the out-of-range marker.  The loop body below has NO conditional memory
    // instruction: the compiler then counts the operations between a load and its use exactly and waits with vmcnt(16+)
    // instead of vmcnt(0) -- with branches around them it drained the whole queue, the stores just issued and the next tile's
    // prefetch included, before every split.)
    auto issue_chunk = [&](const TileAt& at, int c, bool valid) {
      const int tx = X3_ABL == 11 ? 1 : at.tx, ty = X3_ABL == 11 ? 1 : at.ty, b = X3_ABL == 11 ? 0 : at.b;   // ablation 11: every block re-reads one tile (cache hits), no stores
      const int iy = ty * 8 - 1 + su_ly, ix = tx * 32 - 4 + su_q * 4;
      const bool ok = valid && su_act && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W && X3_ABL != 10;     // ablation 10: no global loads, no stores
      const uint32_t vo = ok ? (uint32_t)(((long)b * a.x_bs + (long)su_gg * 8 * hw + (long)iy * a.W + ix) * 4) : OOB;
#pragma unroll
      for (int e = 0; e < 8; ++e)
        raw[c][e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, (int)vo, (int)((c ? c1off : 0u) + e * hw4), 0));
    };
    auto write_chunk = [&](int c) {
      u32x4* buf = xl + c * CHUNK + su_gg * PLANE_PIX + su_ly * LW;
#pragma unroll
      for (int px = 0; px < 4; ++px) {
        const int lx = su_q * 4 + px - 3;                    // patch column of this pixel
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = raw[c][e][px];
        u32x4 h, m, l;
        if (NP == 2) split8_h2<true>(v, sx, h, m);
        else split8(v, h, m, l);
        if (su_act && lx >= 0 && lx < LW) {
          buf[lx] = h;
          buf[lx + 2 * PLANE_PIX] = m;
          if (NP == 3) buf[lx + 4 * PLANE_PIX] = l;
        }
      }
    };
    // Epilogue of one tile: this thread owns an aligned quad of pixels (row eq_row, columns 4*eq_q ..) for 8 channels
    // (co = eq_c8*8 ..).  Two halves: the operand loads are issued at the START of the phase (they fly while the chunk is
    // split and written), the arithmetic and the 16-B stores come at its end -- no load ever waits behind a store.
    const int eq_q = ptid & 7, eq_row = (ptid >> 3) & 7, eq_c8 = ptid >> 6;
    float bias_r[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bias_r[e] = (a.bias && eq_c8 * 8 + e < a.Cout) ? a.bias[eq_c8 * 8 + e] : 0.f;
    f32x4 erv[8], edv[8], emv[8];
    uint32_t evd = OOB, evd2 = OOB;
    uint32_t ebits = 0, ebo = OOB;                          // EPI 4: the tile's mask word; EPI 0: where this tile's word goes
    auto epilogue_loads = [&](const TileAt& at, bool valid, long tix) {
      const int tx = at.tx, ty = at.ty, b = at.b;
      const int oy = ty * 8 + eq_row, ox = tx * 32 + eq_q * 4;
      const bool pv = valid && oy < a.H && ox < a.W;
      const long pofs = (long)oy * a.W + ox + (long)eq_c8 * 8 * hw;
      const uint32_t vr = (EPI >= 1 && pv && a.res) ? (uint32_t)(((long)b * a.res_bs + pofs) * 4) : OOB;
      evd = pv ? (uint32_t)(((long)b * a.y_bs + pofs) * 4) : OOB;
      if (EPI == 3) evd2 = pv ? (uint32_t)(((long)b * a.y2_bs + pofs) * 4) : OOB;
      const uint32_t vm = (EPI == 2 && pv && a.mask) ? (uint32_t)(((long)b * a.mask_bs + pofs) * 4) : OOB;
      if (EPI == 0 || EPI == 4) {
        const uint32_t wo = (uint32_t)(((a.bits_tile0 + tix) * 256 + ptid) * 4);
        if (EPI == 4) ebits = __builtin_amdgcn_raw_buffer_load_b32(rbits, (int)(valid ? wo : OOB), 0, 0);
        else ebo = (valid && a.bits_out) ? wo : OOB;
      }
      if (EPI >= 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const uint32_t so = (uint32_t)e * hw4;
          const int co = eq_c8 * 8 + e;
          const bool cok = co < a.Cout;
          erv[e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rres, (int)(cok ? vr : OOB), (int)so, 0));
          if (EPI == 2 || EPI == 4)
            edv[e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ry, (int)((cok && a.accumulate) ? evd : OOB), (int)so, 0));
          if (EPI == 2)
            emv[e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rmask, (int)((cok && co < a.nmask) ? vm : OOB), (int)so, 0));
        }
      }
    };
    // NP == 2 with a.y_amax / a.y_chmax: max |stored value| of this thread PER CHANNEL (its eight channels are the same for all 64 threads
    // of the wave: eq_c8 = ptid >> 6).  The tensor's maximum is the maximum of these -- the per-channel fold costs the same four VALU
    // instructions per stored 16-B unit the per-tensor fold did, and the cross-lane part runs once per wave and LAUNCH (persistent blocks).
    float chm[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) chm[e] = 0.f;
    const bool want_amax = NP == 2 && (a.y_amax != nullptr || a.y_chmax != nullptr);
    f32x4 eacc[8];                                           // the finished tile's accumulators, copied out of the LDS stage
    auto epilogue_grab = [&]() {
#pragma unroll
      for (int e = 0; e < 8; ++e) eacc[e] = *(const f32x4*)(ol + (eq_c8 * 8 + e) * 256 + eq_row * 32 + eq_q * 4);
    };
    auto epilogue_finish = [&]() {
      uint32_t obits = 0;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int co = eq_c8 * 8 + e;
        f32x4 o, o2;
#pragma unroll
        for (int px = 0; px < 4; ++px) {
          float v = eacc[e][px] + bias_r[e];
          if (a.lrelu) v = irr_lrelu(v);
          if (EPI == 0) v *= a.alpha;
          else if (EPI == 3) { v *= a.alpha; o2[px] = v; v = erv[e][px] + v; }
          else v = erv[e][px] + a.alpha * v;        // erv = 0 without a residual operand
          if (EPI == 2) {
            v += edv[e][px];                        // edv = 0 unless accumulating
            if (a.mask && co < a.nmask) v *= irr_lrelu_grad(emv[e][px]);
          }
          if (EPI == 4) {
            v += edv[e][px];
            if (co < a.nmask) v *= ((ebits >> (e * 4 + px)) & 1u) ? 1.f : 0.1f;
          }
          o[px] = v;
          if (EPI == 0) obits |= (v > 0.f ? 1u : 0u) << (e * 4 + px);     // (the same predicate irr_lrelu_grad applies to the stored value)
        }
        if (want_amax && co < a.Cout && evd != OOB) chm[e] = x3_amax_fold(x3_amax_fold(x3_amax_fold(x3_amax_fold(chm[e], o[0]), o[1]), o[2]), o[3]);
        // (irr_buffer_store_b128_guarded, common.h: these stores carry an SGPR soffset, the form behind which hipcc inserts NO wait states
        // -- and the next channel's arithmetic reuses the data registers at once: the fault of NOTES D.4 / D.5)
        irr_buffer_store_b128_guarded(__builtin_bit_cast(u32x4, o), ry, (int)((co < a.Cout && X3_ABL != 7 && X3_ABL != 10 && X3_ABL != 11) ? evd : OOB),
                                      (int)((uint32_t)e * hw4));
        if (EPI == 3)
          irr_buffer_store_b128_guarded(__builtin_bit_cast(u32x4, o2), ry2, (int)(co < a.Cout ? evd2 : OOB), (int)((uint32_t)e * hw4));
      }
      if (EPI == 0) __builtin_amdgcn_raw_buffer_store_b32(obits, rbits, (int)ebo, 0, 0);       // (out-of-range marker unless bits were asked for)
    };
    TileAt at_prev = {0, 0, 0}, at_cur, at_next;            // tiles t - t_step (epilogue), t, t + t_step (loads in flight)
    at_cur.tx = (int)(t_begin % a.tiles_x);
    at_cur.ty = (int)((t_begin / a.tiles_x) % a.tiles_y);
    at_cur.b = (int)(t_begin / ((long)a.tiles_x * a.tiles_y));
    at_next = at_cur;
    advance(at_next);
    issue_chunk(at_cur, 0, t_begin < a.ntiles);
    issue_chunk(at_cur, 1, t_begin < a.ntiles);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      eacc[e] = f32x4{0.f, 0.f, 0.f, 0.f};
      erv[e] = edv[e] = emv[e] = eacc[e];
    }
    // The two phases of a tile are balanced against the MFMA waves' two chunk phases: the epilogue of tile n-1 is
    // split into operand loads + accumulator copy (phase of chunk 0) and arithmetic + stores (phase of chunk 1).
    // Nothing in the body is conditional: without a finished tile the epilogue stores go to the out-of-range marker (evd).
    long tprev = -1;
    for (long t = t_begin; t < a.ntiles; t += t_step) {
      const bool more = t + t_step < a.ntiles;
      // phase after barrier #(2n-1): the MFMA waves are on (n-1, chunk 1); slot 0 is free
      TR(1);
      epilogue_finish();                                    // tile n-2: operands and accumulators are in registers
      TR(2);
      write_chunk(0);
      TR(3);
      issue_chunk(at_next, 0, more);
      TR(4);
      __syncthreads();
      TR(5);                                      // barrier #2n: (n, chunk 0) published; accumulators of tile n-1 published
      // phase: MFMA waves on (n, chunk 0); slot 1 is free; the accumulator stage holds tile n-1 until barrier #2n+1
      epilogue_loads(at_prev, tprev >= 0, tprev);
      epilogue_grab();
      TR(6);
      write_chunk(1);
      TR(7);
      issue_chunk(at_next, 1, more);
      TR(8);
      __syncthreads();
      TR(9);                                      // barrier #2n+1: (n, chunk 1) published; accumulator stage free again
      tprev = t;
      at_prev = at_cur;
      at_cur = at_next;
      advance(at_next);
    }
    epilogue_finish();
    __syncthreads();                                        // final barrier: accumulators of the last tile published
    epilogue_loads(at_prev, tprev >= 0, tprev);
    epilogue_grab();
    epilogue_finish();
    if (want_amax) {
      float ymax = 0.f;
      uint32_t mine = 0u;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float m = x3_amax_wave(chm[e]);                // (uniform after the fold)
        ymax = x3_amax_fold(ymax, m);
        mine = lane == e ? __builtin_bit_cast(uint32_t, m) : mine;
      }
      if (a.y_amax && lane == 0) x3_amax_commit(ymax, a.y_amax);
      if (a.y_chmax && lane < 8 && eq_c8 * 8 + lane < a.Cout) x3_amax_commit(__builtin_bit_cast(float, mine), a.y_chmax + eq_c8 * 8 + lane);   // eight lanes, one round trip
    }
    return;
  }

  // ================= MFMA waves: wave = pixel group (64 pixels = 2 rows of the tile), one 32-channel co-tile =================
  const int pg = X3S_PRODUCERS_OLD ? wave - 4 : wave;
  const int j = lane & 31, g = lane >> 5;
  const int row0 = pg * 2;                                 // sub-tile s = tile row row0 + s, column j
  const int xidx0 = g * PLANE_PIX + row0 * LW + j;
  for (long t = t_begin; t < a.ntiles; t += t_step) {
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      TR(10 + c);
      __syncthreads();                                      // barrier #2n+c: chunk c of tile n is in slot c
      TR(12 + c);
      if (c == 1) {
        // (the accumulator stage was released by the barrier that just passed: tile n-1's epilogue is done)
      }
      const u32x4* buf = xl + c * CHUNK;
      u32x4 xb[2][2][NP], wa[2][NP];
      auto read_step = [&](int sel, int tap) {
        const int base = xidx0 + (tap / 3) * LW + (tap % 3);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int p = 0; p < NP; ++p) xb[sel][s][p] = (X3_ABL == 9 && tap > 0) ? xb[sel ^ 1][s][p] : buf[base + s * LW + 2 * p * PLANE_PIX];
        if (X3_ABL == 8 && tap > 0) {                       // ablation: the weight fragments of tap 0 for every tap (no LDS reads)
#pragma unroll
          for (int p = 0; p < NP; ++p) wa[sel][p] = wa[sel ^ 1][p];
        } else {
#pragma unroll
          for (int p = 0; p < NP; ++p) wa[sel][p] = wl[((c * 9 + tap) * NP + p) * 64 + lane];
        }
      };
      read_step(0, 0);
#pragma unroll
      for (int tap = 0; tap < (X3_ABL == 5 ? 1 : 9); ++tap) {
        const int cur = tap & 1;
        if (tap + 1 < 9) read_step(cur ^ 1, tap + 1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (NP == 2) {
          const u32x4 wdn = h2_hi_down(wa[cur][0]);         // (the MFMA waves have slack: four v_pk_mul_f16 per tap)
          acc0 = mma_h(wa[cur][1], xb[cur][0][0], acc0);    // lo * hi
          acc1 = mma_h(wa[cur][1], xb[cur][1][0], acc1);
          acc0 = mma_h(wdn, xb[cur][0][1], acc0);           // (hi * 2^-11) * (lo * 2^11)
          acc1 = mma_h(wdn, xb[cur][1][1], acc1);
          acc0 = mma_h(wa[cur][0], xb[cur][0][0], acc0);    // hi * hi
          acc1 = mma_h(wa[cur][0], xb[cur][1][0], acc1);
        } else {
          acc0 = mma(wa[cur][NP - 1], xb[cur][0][0], acc0);   // lo * hi
          acc1 = mma(wa[cur][NP - 1], xb[cur][1][0], acc1);
          acc0 = mma(wa[cur][0], xb[cur][0][NP - 1], acc0);   // hi * lo
          acc1 = mma(wa[cur][0], xb[cur][1][NP - 1], acc1);
          acc0 = mma(wa[cur][1], xb[cur][0][1], acc0);        // mid * mid
          acc1 = mma(wa[cur][1], xb[cur][1][1], acc1);
          acc0 = mma(wa[cur][1], xb[cur][0][0], acc0);        // mid * hi
          acc1 = mma(wa[cur][1], xb[cur][1][0], acc1);
          acc0 = mma(wa[cur][0], xb[cur][0][1], acc0);        // hi * mid
          acc1 = mma(wa[cur][0], xb[cur][1][1], acc1);
          acc0 = mma(wa[cur][0], xb[cur][0][0], acc0);        // hi * hi
          acc1 = mma(wa[cur][0], xb[cur][1][0], acc1);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    TR(14);
    // hand the accumulators to the epilogue waves: D[i][jj], i = (r&3) + 8*(r>>2) + 4*g (channel), jj = column
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = (r & 3) + 8 * (r >> 2) + 4 * g;
        float v = s ? acc1[r] : acc0[r];
        if (NP == 2) v = (v * inv_x) * inv_w;               // back to the operands' own scale (two exact powers of two)
        ol[co * 256 + (row0 + s) * 32 + j] = v;
      }
  }
  __syncthreads();                                          // final barrier (pairs with the producers')
}

// 16-channel chunks of a layer.  Cin = 16 with one co-tile (the 16 -> 16 pyramid convs) is packed as TWO chunks, the second
// with zero weights (the kernels re-read channels [Cin-16, Cin) for it): that is the shape conv_x3s_kernel is built for.
static int x3_nchunk(int Cin, int Cout) { return (Cin == 16 && Cout <= 32) ? 2 : (Cin + 15) / 16; }

// Cout <= 32, or (two launches, one per co-tile) Cout <= 64 with 16 < Cin <= 32: the K = 32 data gradients of the 64 -> 32
// layers and the c5 -> c4 column of the DenseNet backward, read-modify-write launches that ran at 55 TFLOP/s on conv_x3_kernel
static bool x3s_ok(int B, int Cin, int H, int W, int Cout, int dil) {
  if (dil != 1 || Cout > 64 || Cin < 16 || Cin > 32 || (W & 3)) return false;
  if (Cout > 32 && (Cin == 16 || IRR_ENV_FLAG("IRR_X3S_NO_COT2"))) return false;
  const long tiles = (long)B * ((H + 7) / 8) * ((W + 31) / 32);
  const double eff = (double)H * W / ((double)((H + 7) / 8) * ((W + 31) / 32) * 256);
  return eff >= 0.8 && (tiles >= 2048 || g_min_blocks == 0);
}

// ---- weight packing: see pack_x3_unit (pack.h) for the layout and the three modes ----
template <int NP>
__global__ void pack_x3_kernel(const float* __restrict__ w, u32x4* __restrict__ wq, int Cin, int Cout, int CoT, int nchunk,
                               int mode, int w_cin, int chan0, int row_offset, int w_cout, long nunits, const float* __restrict__ amax) {
  const long u = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (u < nunits) pack_x3_unit<NP>(w, wq, Cin, Cout, CoT, nchunk, mode, w_cin, chan0, row_offset, w_cout, u, amax);
}

struct TileCfg { int nt, tr, tc; };

// choose the TR x TC tile (TR*TC = 32*NT*PG pixels) that wastes the fewest MFMA columns, then the smallest halo
static bool pick_tile(int H, int W, int dy, int dx, int PG, int plane_pix, const int* nts, int n_nts, TileCfg* out) {
  double best = 1e30;
  bool found = false;
  for (int q = 0; q < n_nts; ++q) {
    const int P = 32 * nts[q] * PG;
    for (int tc = 4; tc <= P; ++tc) {
      if (P % tc) continue;
      const int tr = P / tc;
      if ((long)(tr + 2 * dy) * (tc + 2 * dx) > plane_pix) continue;
      const double tiles = (double)((H + tr - 1) / tr) * ((W + tc - 1) / tc);
      const double cost = tiles * P * (1.0 + 0.05 * (double)(tr + 2 * dy) * (tc + 2 * dx) / P);
      if (cost < best) { best = cost; out->nt = nts[q]; out->tr = tr; out->tc = tc; found = true; }
    }
  }
  return found;
}

template <int CT, int PG, int NT, int PLANE_PIX>
int launch_x3(X3Args& a, const TileCfg& t, hipStream_t st, int np) {
  a.TR = t.tr; a.TC = t.tc;
  const int Hs = (a.H + a.RD - 1) / a.RD;
  a.tiles_x = (a.W + t.tc - 1) / t.tc;
  a.tiles_y = (Hs + t.tr - 1) / t.tr;
  a.ngy = (a.CoT + CT - 1) / CT;
  dim3 grid((unsigned)((long)a.B * a.RD * a.tiles_x * a.tiles_y * a.ngy), 1, (unsigned)a.ksplit);
  if (np == 2) hipLaunchKernelGGL((conv_x3_kernel<CT, PG, NT, PLANE_PIX, 2>), grid, dim3(CT * PG * 64), 0, st, a);
  else hipLaunchKernelGGL((conv_x3_kernel<CT, PG, NT, PLANE_PIX, 3>), grid, dim3(CT * PG * 64), 0, st, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

// block shape: CT co-tile waves (x PG pixel groups) per block
static int pick_ct(int CoT) {
  if (CoT == 1) return 1;
  if (CoT == 2) return 2;
  if (CoT % 4 == 0) return 4;
  if (CoT >= 7) return 4;            // 4 waves = one per SIMD; a partly idle last group costs less than 3-wave blocks
  if (CoT % 3 == 0) return 3;
  return 4;
}

struct Plan { int ct, pg, plane, rd, ksplit; TileCfg t; long blocks; double eff; };

// rd = 1: plain (TR + 2 dil) x (TC + 2 dil) halo patches.  rd = dil: rows folded by residue class, the patch is
// (TR + 2) x (TC + 2 dil) -- the only way a dilation-8/16 patch fits the LDS planes, and a smaller halo for 2 and 4.
static bool make_plan_rd(int B, int Cin, int H, int W, int Cout, int dil, int rd, Plan* p) {
  const int CoT = (Cout + 31) / 32;
  const int Hs = (H + rd - 1) / rd;
  const int dy = rd > 1 ? 1 : dil;
  p->rd = rd;
  p->ct = pick_ct(CoT);
  static int nts78[2] = {8, 7};
  static const int nts4[1] = {4};
  static const int force_nt = getenv("IRR_X3_NT") ? atoi(getenv("IRR_X3_NT")) : 0;       // experiment switch: 7 or 8
  if (force_nt == 7 || force_nt == 8) nts78[0] = nts78[1] = force_nt;
  bool ok;
  if (p->ct >= 3) {
    // NT = 8 or 7 (256 / 224-pixel tiles): the one with the smaller estimated time.  Blocks of a launch take equal time and
    // 512 of them are resident, so a launch of few rounds pays for its partly filled last round (2688 blocks of NT = 8 at
    // 96x112x64 = 5.25 rounds cost 6; 3072 blocks of NT = 7 cost 6 * 7/8); one sub-tile of fixed cost per block.
    p->pg = 1;
    ok = false;
    double best = 1e30;
    const int cog = (CoT + p->ct - 1) / p->ct;
    for (int q = 0; q < 2; ++q) {
      TileCfg t;
      int plane = 352;
      bool okq = pick_tile(Hs, W, dy, dil, 1, 352, nts78 + q, 1, &t);
      if (!okq) { plane = 616; okq = pick_tile(Hs, W, dy, dil, 1, 616, nts78 + q, 1, &t); }
      if (!okq) continue;
      const double blocks = (double)B * rd * ((Hs + t.tr - 1) / t.tr) * ((W + t.tc - 1) / t.tc) * cog;
      const double rounds = blocks / 512.0;
      const double est = (rounds <= 8.0 ? (double)(long)(rounds + 0.999) : rounds) * (t.nt + 1.0) * (plane == 616 ? 1.04 : 1.0);
      if (est < best) { best = est; p->t = t; p->plane = plane; ok = true; }
    }
  } else if (p->ct == 2) {
    p->pg = 2;
    p->plane = 616;
    // (measured at 96x112: quarter-size sub-tile sets, 3 blocks per CU, beat NT = 8 by 17-21 % for the 64-channel layers)
    ok = pick_tile(Hs, W, dy, dil, 2, 616, nts4, 1, &p->t);
    {
      // 640-pixel planes: the 2 x 128 tile of a row-folded dilation-16 layer ((2 + 2) x (128 + 32) halo) -- with 616 the 96 -> 64
      // forward of the context networks only found a 4 x 64 tile that wastes a third of its pixels and stayed on the fp32 kernel
      auto eff_of = [&](const TileCfg& t) { return (double)Hs * W / ((double)((Hs + t.tr - 1) / t.tr) * ((W + t.tc - 1) / t.tc) * t.tr * t.tc); };
      TileCfg t2;
      if (!IRR_ENV_FLAG("IRR_X3_NO_PLANE640") && pick_tile(Hs, W, dy, dil, 2, 640, nts4, 1, &t2) && (!ok || eff_of(t2) > eff_of(p->t) + 0.1)) {
        p->t = t2;
        p->plane = 640;
        ok = true;
      }
    }
    if (!ok) ok = pick_tile(Hs, W, dy, dil, 2, 616, nts78, 2, &p->t);
  } else {
    p->pg = 4;
    static const int nts2[1] = {2};
    p->plane = 352;                                       // 256-pixel tiles, four blocks per CU (+4 ... +18 % over NT = 4)
    ok = pick_tile(Hs, W, dy, dil, 4, 352, nts2, 1, &p->t);
    if (!ok) {
      p->plane = 616;
      ok = pick_tile(Hs, W, dy, dil, 4, 616, nts4, 1, &p->t);
    }
  }
  if (!ok) return false;
  auto ntiles = [&](const TileCfg& t) { return (long)((Hs + t.tr - 1) / t.tr) * ((W + t.tc - 1) / t.tc); };
  p->blocks = (long)B * rd * ntiles(p->t) * ((CoT + p->ct - 1) / p->ct);
  if ((p->blocks < g_min_blocks || IRR_ENV_FLAG("IRR_X3_FORCE_NT4")) && p->ct >= 2) {
    // small pyramid levels: half-size tiles (NT = 4) double the number of blocks
    TileCfg t4;
    const int plane4 = p->ct == 2 ? 616 : 352;
    if (pick_tile(Hs, W, dy, dil, p->pg, plane4, nts4, 1, &t4)) {
      const long b4 = (long)B * rd * ntiles(t4) * ((CoT + p->ct - 1) / p->ct);
      const double e4 = (double)H * W / ((double)rd * ntiles(t4) * t4.tr * t4.tc);
      if ((b4 > p->blocks || IRR_ENV_FLAG("IRR_X3_FORCE_NT4")) && e4 >= 0.70) { p->t = t4; p->plane = plane4; p->blocks = b4; }
    }
  }
  // padded work must stay close to the real work
  p->eff = (double)H * W / ((double)rd * ntiles(p->t) * p->t.tr * p->t.tc);
  // small pyramid levels: too few tiles to fill the chip -> blockIdx.z additionally splits the 16-channel chunks (K); the
  // partial sums meet in x3_splitk_epilogue_kernel.  At least two chunks per slice, about two blocks per CU in total.
  p->ksplit = 1;
  if (g_min_blocks > 0 && p->blocks < g_min_blocks && !IRR_ENV_FLAG("IRR_X3_NO_SPLITK")) {
    const int nchunk = x3_nchunk(Cin, Cout);
    long ks = (512 + p->blocks - 1) / p->blocks;
    if (ks > nchunk / 2) ks = nchunk / 2;
    if (ks > 16) ks = 16;
    if (ks >= 2) p->ksplit = (int)ks;
  }
  // (dilation >= 8: the fp32 kernel stages nine tap-shifted copies there and is the slower alternative by a wider margin -- the
  // 96 -> 64 dilation-16 forward at 48x56 has a best tile of 4 x 64 = 0.656 and runs 258 -> 145 us on this kernel;
  // IRR_X3_MIN_EFF: experiment switch, percent)
  static const double min_eff_env = getenv("IRR_X3_MIN_EFF") ? atof(getenv("IRR_X3_MIN_EFF")) / 100.0 : 0.0;
  const double min_eff = min_eff_env > 0.0 ? min_eff_env : (dil >= 8 ? 0.65 : 0.70);
  return p->eff >= min_eff;
}

static bool make_plan(int B, int Cin, int H, int W, int Cout, int dil, Plan* p) {
  if (Cin < 16 || dil < 1 || H < 8 || W < 8) return false;
  const int CoT = (Cout + 31) / 32;
  if (Cin < 64 && CoT == 1 && g_min_blocks > 0) return false;   // two or three chunks AND one co-tile: see conv_x3s_kernel
  if (dil > 1 && !IRR_ENV_FLAG("IRR_X3_NO_ROWFOLD")) {
    Plan f;
    const bool okf = make_plan_rd(B, Cin, H, W, Cout, dil, dil, &f);
    Plan u;
    const bool oku = make_plan_rd(B, Cin, H, W, Cout, dil, 1, &u);
    if (okf && (!oku || f.eff >= u.eff - 0.08)) { *p = f; return true; }
    if (oku) { *p = u; return true; }
    return false;
  }
  return make_plan_rd(B, Cin, H, W, Cout, dil, 1, p);
}

}  // namespace

#ifdef X3S_TRACE
extern "C" int irr_x3s_trace_dump(unsigned long long* host) {
  if (!g_x3s_dbg) return -1;
  hipDeviceSynchronize();
  return (int)hipMemcpy(host, g_x3s_dbg, 8 * 400 * 8, hipMemcpyDeviceToHost);
}
#endif

extern "C" int irr_conv_x3_set_min_blocks(int n) {
  return n >= 0 ? g_min_blocks.exchange(n) : g_min_blocks.load();
}

extern "C" long irr_conv_x3_packed_bytes(int Cin, int Cout) {
  const long CoT = (Cout + 31) / 32, nchunk = x3_nchunk(Cin, Cout);
  return nchunk * 9 * 3 * CoT * 64 * 16;
}

extern "C" int irr_conv_pack_weights_x3(const float* w, void* wq, int Cin, int Cout, int transpose, void* stream) {
  if (!w || !wq || Cin < 16 || Cout <= 0) return IRR_EINVAL;
  const int CoT = (Cout + 31) / 32, nchunk = x3_nchunk(Cin, Cout);
  const long nunits = (long)nchunk * 9 * CoT * 64;
  hipLaunchKernelGGL(pack_x3_kernel<3>, dim3(irr_cdiv(nunits, 256)), dim3(256), 0, (hipStream_t)stream, w, (u32x4*)wq, Cin, Cout,
                     CoT, nchunk, transpose ? 1 : 0, 0, 0, 0, 0, nunits, (const float*)nullptr);
  IRR_LAUNCH_CHECK();
  return 0;
}

// ---- the fp16x2 ("h2") form: two pieces per fragment + one 16-B unit that holds the pack's scale exponent ----
extern "C" long irr_conv_h2_packed_bytes(int Cin, int Cout) {
  const long CoT = (Cout + 31) / 32, nchunk = x3_nchunk(Cin, Cout);
  return (nchunk * 9 * 2 * CoT * 64 + 1) * 16;
}

extern "C" int irr_conv_pack_weights_h2(const float* w, void* wq, int Cin, int Cout, int transpose, const float* amax, void* stream) {
  if (!w || !wq || !amax || Cin < 16 || Cout <= 0) return IRR_EINVAL;
  const int CoT = (Cout + 31) / 32, nchunk = x3_nchunk(Cin, Cout);
  const long nunits = (long)nchunk * 9 * CoT * 64;
  hipLaunchKernelGGL(pack_x3_kernel<2>, dim3(irr_cdiv(nunits, 256)), dim3(256), 0, (hipStream_t)stream, w, (u32x4*)wq, Cin, Cout,
                     CoT, nchunk, transpose ? 1 : 0, 0, 0, 0, 0, nunits, amax);
  IRR_LAUNCH_CHECK();
  return 0;
}

extern "C" int irr_conv_pack_weights_h2_sub(const float* w, void* wq, int w_cin, int w_cout, int total_rows, int chan0,
                                            int nchan, int row_offset, const float* amax, void* stream) {
  if (!w || !wq || !amax || w_cin <= 0 || w_cout <= 0 || total_rows < 16 || (total_rows & 15) || chan0 < 0 || nchan <= 0 ||
      chan0 + nchan > w_cin || row_offset < 0 || (row_offset & 7) || (w_cout & 7) || row_offset + w_cout > total_rows)
    return IRR_EINVAL;
  const int CoT = (nchan + 31) / 32, nchunk = total_rows / 16;
  const long nunits = (long)nchunk * 9 * CoT * 64;
  hipLaunchKernelGGL(pack_x3_kernel<2>, dim3(irr_cdiv(nunits, 256)), dim3(256), 0, (hipStream_t)stream, w, (u32x4*)wq, total_rows,
                     nchan, CoT, nchunk, 2, w_cin, chan0, row_offset, w_cout, nunits, amax);
  IRR_LAUNCH_CHECK();
  return 0;
}

// slot = max(slot, max |x|) over a (B, n) block of plane-dense samples (batch stride bs): the per-tensor magnitude the h2
// kernels scale their operands by.  slot must hold a non-negative float (0 to start); several launches may fold into one slot.
__global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ x, long n4, long n, long bs, float* __restrict__ slot) {
  const float* xb = x + (long)blockIdx.y * bs;
  float m = 0.f;
  const long stride = (long)gridDim.x * 256;
  if ((((uintptr_t)xb) & 15) == 0) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
      const f32x4 v = ((const f32x4*)xb)[i];
      m = x3_amax_fold(x3_amax_fold(x3_amax_fold(x3_amax_fold(m, v[0]), v[1]), v[2]), v[3]);
    }
    for (long i = n4 * 4 + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) m = x3_amax_fold(m, xb[i]);
  } else {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) m = x3_amax_fold(m, xb[i]);
  }
  x3_amax_publish_block256(m, slot);
}

extern "C" int irr_amax_f32(const float* x, int B, long n, long bs, float* slot, void* stream) {
  if (!x || !slot || B <= 0 || n <= 0) return IRR_EINVAL;
  long blocks = irr_cdiv(n / 4 + 1, 256 * 8);             // >= eight 16-B loads per thread
  const long cap = 1024 / B > 0 ? 1024 / B : 1;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(amax_kernel, dim3((unsigned)blocks, (unsigned)B), dim3(256), 0, (hipStream_t)stream, x, n / 4, n, bs, slot);
  IRR_LAUNCH_CHECK();
  return 0;
}

extern "C" int irr_conv_pack_weights_x3_sub(const float* w, void* wq, int w_cin, int w_cout, int total_rows, int chan0,
                                            int nchan, int row_offset, void* stream) {
  if (!w || !wq || w_cin <= 0 || w_cout <= 0 || total_rows < 16 || (total_rows & 15) || chan0 < 0 || nchan <= 0 ||
      chan0 + nchan > w_cin || row_offset < 0 || (row_offset & 7) || (w_cout & 7) || row_offset + w_cout > total_rows)
    return IRR_EINVAL;
  const int CoT = (nchan + 31) / 32, nchunk = total_rows / 16;
  const long nunits = (long)nchunk * 9 * CoT * 64;
  hipLaunchKernelGGL(pack_x3_kernel<3>, dim3(irr_cdiv(nunits, 256)), dim3(256), 0, (hipStream_t)stream, w, (u32x4*)wq, total_rows,
                     nchan, CoT, nchunk, 2, w_cin, chan0, row_offset, w_cout, nunits, (const float*)nullptr);
  IRR_LAUNCH_CHECK();
  return 0;
}

extern "C" int irr_conv2d_x3_eligible(int B, int Cin, int H, int W, int Cout, int k, int stride, int dil) {
  if (k != 3 || stride != 1 || B <= 0) return 0;
  if (x3s_ok(B, Cin, H, W, Cout, dil)) return 9001;                       // streaming 32-channel kernel
  Plan p;
  if (!make_plan(B, Cin, H, W, Cout, dil, &p)) return 0;
  if (p.blocks * p.ksplit < g_min_blocks / 2) return 0;       // launches that cannot fill half the chip even with a K split: fp32 split-K kernel
  return p.ct * 1000 + p.pg * 100 + p.t.nt * 10 + (p.plane == 352 ? 1 : p.plane == 616 ? 2 : 3);
}

// the fp16x2 form takes what the bf16x3 kernels take (9001: the streaming 32-channel kernel; IRR_X3S_NO_H2: A/B switch that keeps
// that one on bf16x3)
extern "C" int irr_conv2d_h2_eligible(int B, int Cin, int H, int W, int Cout, int k, int stride, int dil) {
  const int code = irr_conv2d_x3_eligible(B, Cin, H, W, Cout, k, stride, dil);
  return (code == 9001 && IRR_ENV_FLAG("IRR_X3S_NO_H2")) ? 0 : code;
}

// finishes a K-split launch: y = epilogue(sum over the slices of part[kz][b][co][pixel])  (same order of operations as the
// epilogue of conv_x3_kernel: bias, LeakyReLU, residual / alpha, accumulate, LeakyReLU'-mask)
__global__ __launch_bounds__(256) void x3_splitk_epilogue_kernel(const float* __restrict__ part, const float* __restrict__ bias,
                                                                const float* __restrict__ res, float* __restrict__ y,
                                                                const float* __restrict__ mask, int B, int Cout, long hw, int ksplit,
                                                                long y_bs, long res_bs, long mask_bs, int nmask, int lrelu,
                                                                float alpha, int accumulate, float* __restrict__ y_amax) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long n = (long)B * Cout * hw;
  float v = 0.f;
  if (i < n) {
    const long pix = i % hw;
    const long r = i / hw;
    const int co = (int)(r % Cout);
    const long b = r / Cout;
    for (int k = 0; k < ksplit; ++k) v += part[(long)k * n + i];
    v += bias ? bias[co] : 0.f;
    if (lrelu) v = irr_lrelu(v);
    const long o = (long)co * hw + pix;
    if (res) v = res[b * res_bs + o] + alpha * v;
    else v *= alpha;
    float* dst = y + b * y_bs + o;
    if (accumulate) v += *dst;
    if (mask && co < nmask) v *= irr_lrelu_grad(mask[b * mask_bs + o]);
    *dst = v;
  }
  if (y_amax) x3_amax_publish_block256(__builtin_fabsf(v), y_amax);      // (uniform: every thread of the block arrives)
}

// The same with BLOCKS THAT STAY INSIDE ONE CHANNEL: used when the launch also owes the channel maxima of what it stores
// (irr_conv_x3_next_chmax) -- a block = the planes of channel co for spb consecutive samples (256 / hw of them at the 6x7 and 12x14
// levels), so its maximum is one channel's: one look-then-atomic per block instead of a second launch that reads the finished slice
// again (26 such passes per train step on the main stream).  (One WAVE per plane, or several, was slower than the flat kernel plus
// the pass: tens of thousands of waves each looking at the same slots.)
__global__ __launch_bounds__(256) void x3_splitk_epilogue_planes_kernel(const float* __restrict__ part, const float* __restrict__ bias,
                                                                       const float* __restrict__ res, float* __restrict__ y,
                                                                       const float* __restrict__ mask, int B, int Cout, long hw, int ksplit,
                                                                       long y_bs, long res_bs, long mask_bs, int nmask, int lrelu,
                                                                       float alpha, int accumulate, float* __restrict__ y_amax,
                                                                       float* __restrict__ y_chmax, int spb) {
  const int co = (int)(blockIdx.x % (unsigned)Cout);
  const long b0 = (long)(blockIdx.x / (unsigned)Cout) * spb;
  const long n = (long)B * Cout * hw;
  const float bv = bias ? bias[co] : 0.f;
  const bool masked = mask && co < nmask;
  float m = 0.f;
  const int ihw = (int)hw;                                   // (hw <= 4096 here)
  // spb > 1: hw < 256 and a thread owns at most one element (sample e / hw, pixel e % hw: one 32-bit division per thread);
  // spb == 1: the pixels of one plane, no division
  const int s0 = spb > 1 ? (int)threadIdx.x / ihw : 0;
  const int p0 = spb > 1 ? (int)threadIdx.x - s0 * ihw : (int)threadIdx.x;
  const int pstep = spb > 1 ? ihw : 256;                     // (spb > 1: one trip)
  for (int pix = p0; pix < ihw && s0 < spb; pix += pstep) {
    const long b = b0 + s0;
    if (b >= B) break;
    const long i = (b * Cout + co) * hw + pix;
    float v = 0.f;
#pragma unroll 4
    for (int k = 0; k < ksplit; ++k) v += part[(long)k * n + i];
    v += bv;
    if (lrelu) v = irr_lrelu(v);
    const long o = (long)co * hw + pix;
    if (res) v = res[b * res_bs + o] + alpha * v;
    else v *= alpha;
    float* dst = y + b * y_bs + o;
    if (accumulate) v += *dst;
    if (masked) v *= irr_lrelu_grad(mask[b * mask_bs + o]);
    *dst = v;
    m = x3_amax_fold(m, v);
  }
  __shared__ float wm[4];
  m = x3_amax_wave(m);
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float r = x3_amax_fold(x3_amax_fold(wm[0], wm[1]), x3_amax_fold(wm[2], wm[3]));
    x3_amax_commit(r, y_chmax + co);
    if (y_amax) x3_amax_commit(r, y_amax);
  }
}

// floats of scratch a K-split launch needs (0: the problem runs unsplit)
extern "C" long irr_conv2d_fwd_x3_ws_elems(int B, int Cin, int H, int W, int Cout, int dil) {
  if (B <= 0 || Cin < 16 || Cout <= 0 || H <= 0 || W <= 0 || dil < 1 || x3s_ok(B, Cin, H, W, Cout, dil)) return 0;
  Plan p;
  if (!make_plan(B, Cin, H, W, Cout, dil, &p) || p.ksplit <= 1) return 0;
  return (long)p.ksplit * B * Cout * H * W;
}

static int fwd_x3_impl(const float* x, const void* wq, const float* bias, const float* res, float* y, int B,
                       int Cin, int H, int W, int Cout, int dil, long x_bs, long y_bs, long res_bs, int lrelu,
                       float alpha, int accumulate, const float* mask, long mask_bs, int nmask, float* ws, long ws_elems,
                       void* stream, float* y2 = nullptr, long y2_bs = 0, int np = 3, const float* x_amax = nullptr, int n_amax = 0,
                       float* y_amax = nullptr, const uint32_t* mask_bits = nullptr, uint32_t* bits_out = nullptr,
                       unsigned int* kcnt = nullptr, long kcnt_elems = 0) {
  float* const next_chmax = g_next_chmax;                   // (irr_conv_x3_next_chmax: consumed by this launch, whatever happens to it)
  g_next_chmax = nullptr;
  if (!x || !wq || !y || B <= 0 || Cin < 16 || Cout <= 0 || H <= 0 || W <= 0 || dil < 1) return IRR_EINVAL;
  if (next_chmax && np != 2) return IRR_EINVAL;             // only the fp16x2 forms fold channel maxima
  if (y2 && (!res || accumulate || mask || !x3s_ok(B, Cin, H, W, Cout, dil))) return IRR_EINVAL;      // second output: streaming kernel only
  if (np == 2 && (!x_amax || n_amax <= 0)) return IRR_EINVAL;
  // bit masks: the fp16x2 streaming kernel with one co-tile only; a launch either writes them (plain forward) or reads them
  if ((mask_bits || bits_out) && (np != 2 || Cout > 32 || !x3s_ok(B, Cin, H, W, Cout, dil) || (mask_bits && (mask || bits_out || nmask <= 0)) ||
                                  (bits_out && (res || accumulate || mask || y2))))
    return IRR_EINVAL;
  if (x3s_ok(B, Cin, H, W, Cout, dil)) {
    X3SArgs s;
    s.wq = (const u32x4*)wq; s.bias = bias;
    s.Cin = Cin; s.H = H; s.W = W; s.Cout = Cout;
    s.tiles_x = (W + 31) / 32; s.tiles_y = (H + 7) / 8;
    s.x_bs = x_bs; s.y_bs = y_bs; s.res_bs = res_bs;
    s.lrelu = lrelu; s.accumulate = accumulate; s.alpha = alpha;
    s.mask_bs = mask_bs; s.nmask = nmask;
    s.dbg = nullptr;
#ifdef X3S_TRACE
    static unsigned long long* dbg_buf = nullptr;
    if (!dbg_buf) { hipMalloc(&dbg_buf, 8 * 400 * 8); }
    hipMemsetAsync(dbg_buf, 0, 8 * 400 * 8, (hipStream_t)stream);
    s.dbg = dbg_buf;
    g_x3s_dbg = dbg_buf;
#endif
    constexpr size_t lds3 = (18 * 3 * 64 + 2 * 6 * 352) * 16 + 32 * 256 * 4, lds2 = (18 * 2 * 64 + 2 * 4 * 352) * 16 + 32 * 256 * 4;
    const size_t lds_bytes = np == 2 ? lds2 : lds3;
    static bool attr_set = false;
    if (!attr_set) {
      IRR_HIP_TRY(hipFuncSetAttribute((const void*)conv_x3s_kernel<0, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));
      IRR_HIP_TRY(hipFuncSetAttribute((const void*)conv_x3s_kernel<1, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));
      IRR_HIP_TRY(hipFuncSetAttribute((const void*)conv_x3s_kernel<2, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));
      IRR_HIP_TRY(hipFuncSetAttribute((const void*)conv_x3s_kernel<3, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));
      IRR_HIP_TRY(hipFuncSetAttribute((const void*)conv_x3s_kernel<0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
      IRR_HIP_TRY(hipFuncSetAttribute((const void*)conv_x3s_kernel<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
      IRR_HIP_TRY(hipFuncSetAttribute((const void*)conv_x3s_kernel<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
      IRR_HIP_TRY(hipFuncSetAttribute((const void*)conv_x3s_kernel<3, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
      IRR_HIP_TRY(hipFuncSetAttribute((const void*)conv_x3s_kernel<4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
      attr_set = true;
    }
    s.x_amax = x_amax; s.n_amax = n_amax; s.y_amax = y_amax;
    s.mask_bits = mask_bits; s.bits_out = bits_out;
    // every operand is addressed through 32-bit byte voffsets below the 2 GiB out-of-range marker
    long bsmax = x_bs > y_bs ? x_bs : y_bs;
    if (res && res_bs > bsmax) bsmax = res_bs;
    if (mask && mask_bs > bsmax) bsmax = mask_bs;
    if (y2 && y2_bs > bsmax) bsmax = y2_bs;
    s.y2 = nullptr; s.y2_bs = y2_bs;
    const long lim = (1L << 29) - (long)(Cin + 48) * H * W - 64;
    if (lim <= 0) return IRR_EINVAL;
    long per = bsmax > 0 ? lim / bsmax : B;
    if (per < 1) per = 1;
    if (per > B) per = B;
    const long hw_ = (long)H * W;
    s.wCoT = (Cout + 31) / 32;
    for (int b0 = 0; b0 < B; b0 += (int)per) {
      s.B = (B - b0) < per ? (B - b0) : (int)per;
      s.x = x + (long)b0 * x_bs;
      s.ntiles = (long)s.B * s.tiles_x * s.tiles_y;
      // the bit-mask words are rebased per batch chunk like x and y: the kernel forms their byte offset in 32 bits against the 2 GiB
      // out-of-range marker, and a chunk's words (<= H * W * 4 bytes per sample) stay below it when its activations do
      const long bits_w0 = (long)b0 * s.tiles_x * s.tiles_y * 256;
      s.mask_bits = mask_bits ? mask_bits + bits_w0 : nullptr;
      s.bits_out = bits_out ? bits_out + bits_w0 : nullptr;
      s.bits_tile0 = 0;
      const long nblk = s.ntiles < 256 ? s.ntiles : 256;             // persistent: one block per CU
      for (int cot = 0; cot < s.wCoT; ++cot) {                       // one launch per 32-channel co-tile
        const long co0 = 32L * cot;
        s.wcot = cot;
        s.Cout = Cout - (int)co0 < 32 ? Cout - (int)co0 : 32;
        s.bias = bias ? bias + co0 : nullptr;
        s.y = y + (long)b0 * y_bs + co0 * hw_;
        s.res = res ? res + (long)b0 * res_bs + co0 * hw_ : nullptr;
        s.mask = (mask && nmask > co0) ? mask + (long)b0 * mask_bs + co0 * hw_ : nullptr;
        s.nmask = nmask - (int)co0 < 0 ? 0 : (nmask - (int)co0 > 32 ? 32 : nmask - (int)co0);
        s.y2 = y2 ? y2 + (long)b0 * y2_bs + co0 * hw_ : nullptr;
        s.y_chmax = (np == 2 && next_chmax) ? next_chmax + co0 : nullptr;
        const int epi = s.mask_bits ? 4 : (s.accumulate || s.mask) ? 2 : s.res ? (s.y2 ? 3 : 1) : 0;
#define X3S_GO(E) do { if (np == 2) hipLaunchKernelGGL((conv_x3s_kernel<E, 2>), dim3((unsigned)nblk), dim3(512), lds_bytes, (hipStream_t)stream, s); \
                      else hipLaunchKernelGGL((conv_x3s_kernel<E, 3>), dim3((unsigned)nblk), dim3(512), lds_bytes, (hipStream_t)stream, s); } while (0)
        if (epi == 4) hipLaunchKernelGGL((conv_x3s_kernel<4, 2>), dim3((unsigned)nblk), dim3(512), lds_bytes, (hipStream_t)stream, s);
        else if (epi == 3) X3S_GO(3);
        else if (epi == 0) X3S_GO(0);
        else if (epi == 1) X3S_GO(1);
        else X3S_GO(2);
#undef X3S_GO
        IRR_LAUNCH_CHECK();
      }
    }
    return 0;
  }
  Plan p;
  if (!make_plan(B, Cin, H, W, Cout, dil, &p)) return IRR_EINVAL;
  X3Args a;
  a.wq = (const u32x4*)wq; a.bias = bias;
  a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout; a.dil = dil; a.RD = p.rd;
  a.CoT = (Cout + 31) / 32; a.nchunk = x3_nchunk(Cin, Cout);
  a.x_bs = x_bs; a.y_bs = y_bs; a.res_bs = res_bs;
  a.lrelu = lrelu; a.accumulate = accumulate; a.alpha = alpha;
  a.mask_bs = mask_bs; a.nmask = nmask;
  a.ksplit = (ws && (long)p.ksplit * B * Cout * H * W <= ws_elems) ? p.ksplit : 1;     // no scratch: run unsplit (slower, same result class)
  a.part = ws;
  a.x_amax = x_amax; a.n_amax = n_amax; a.y_amax = y_amax;
  a.y_chmax = np == 2 ? next_chmax : nullptr;
  a.kcnt = (a.ksplit > 1 && p.t.nt <= 4 && kcnt && kcnt_elems >= p.blocks) ? kcnt : nullptr;       // (zeroed counters: the launch finishes its K split itself; NT <= 4 kernels only)
  // 32-bit byte voffsets below the 2 GiB out-of-range marker: split the batch accordingly
  const long lim = (1L << 29) - (long)(Cin + 16) * H * W - 64;        // elements
  if (lim <= 0) return IRR_EINVAL;
  long per = x_bs > 0 ? lim / x_bs : B;
  if (per < 1) per = 1;
  if (per > B) per = B;
  for (int b0 = 0; b0 < B; b0 += (int)per) {
    a.B = (B - b0) < per ? (B - b0) : (int)per;
    a.x = x + (long)b0 * x_bs;
    a.y = y + (long)b0 * y_bs;
    a.res = res ? res + (long)b0 * res_bs : nullptr;
    a.mask = mask ? mask + (long)b0 * mask_bs : nullptr;
    int rc;
    hipStream_t st = (hipStream_t)stream;
    const int key = p.ct * 1000 + p.pg * 100 + p.t.nt * 10 + (p.plane == 352 ? 1 : p.plane == 616 ? 2 : 3);
    switch (key) {
      case 4181: rc = launch_x3<4, 1, 8, 352>(a, p.t, st, np); break;
      case 4171: rc = launch_x3<4, 1, 7, 352>(a, p.t, st, np); break;
      case 4182: rc = launch_x3<4, 1, 8, 616>(a, p.t, st, np); break;
      case 4172: rc = launch_x3<4, 1, 7, 616>(a, p.t, st, np); break;
      case 3181: rc = launch_x3<3, 1, 8, 352>(a, p.t, st, np); break;
      case 3171: rc = launch_x3<3, 1, 7, 352>(a, p.t, st, np); break;
      case 3182: rc = launch_x3<3, 1, 8, 616>(a, p.t, st, np); break;
      case 3172: rc = launch_x3<3, 1, 7, 616>(a, p.t, st, np); break;
      case 2282: rc = launch_x3<2, 2, 8, 616>(a, p.t, st, np); break;
      case 2272: rc = launch_x3<2, 2, 7, 616>(a, p.t, st, np); break;
      case 1442: rc = launch_x3<1, 4, 4, 616>(a, p.t, st, np); break;
      case 1421: rc = launch_x3<1, 4, 2, 352>(a, p.t, st, np); break;
      case 4141: rc = launch_x3<4, 1, 4, 352>(a, p.t, st, np); break;
      case 3141: rc = launch_x3<3, 1, 4, 352>(a, p.t, st, np); break;
      case 2242: rc = launch_x3<2, 2, 4, 616>(a, p.t, st, np); break;
      case 2243: rc = launch_x3<2, 2, 4, 640>(a, p.t, st, np); break;
      default: return IRR_EINVAL;
    }
    if (rc) return rc;
    if (a.kcnt) {                                            // (finished inside the launch; the next batch chunk counts on its own counters)
      a.kcnt += (long)a.B * a.RD * a.tiles_x * a.tiles_y * a.ngy;
      continue;
    }
    if (a.ksplit > 1) {
      const long n = (long)a.B * Cout * H * W;
      if (a.y_chmax && (long)H * W <= 4096 && !IRR_ENV_FLAG("IRR_X3_NO_PLANES_EPILOGUE")) {       // (A/B switch: the flat kernel + a pass)               // (the small pyramid levels: waves that stay inside a plane fold the channel's maximum)
        const int spb = (long)H * W >= 256 ? 1 : (int)(256 / ((long)H * W));
        const long blocks = (long)Cout * irr_cdiv(a.B, spb);
        hipLaunchKernelGGL(x3_splitk_epilogue_planes_kernel, dim3((unsigned)blocks), dim3(256), 0, st, ws, bias, a.res, a.y, a.mask,
                           a.B, Cout, (long)H * W, a.ksplit, y_bs, res_bs, mask_bs, nmask, lrelu, alpha, accumulate,
                           np == 2 ? y_amax : nullptr, a.y_chmax, spb);
      } else
        hipLaunchKernelGGL(x3_splitk_epilogue_kernel, dim3((unsigned)irr_cdiv(n, 256)), dim3(256), 0, st, ws, bias, a.res, a.y, a.mask, a.B,
                           Cout, (long)H * W, a.ksplit, y_bs, res_bs, mask_bs, nmask, lrelu, alpha, accumulate, np == 2 ? y_amax : nullptr);
      IRR_LAUNCH_CHECK();
      // (larger planes with a K split: one pass over the finished slice of y)
      if (a.y_chmax && ((long)H * W > 4096 || IRR_ENV_FLAG("IRR_X3_NO_PLANES_EPILOGUE"))) { const int rc2 = irr_amax_channels_launch(a.y, a.B, Cout, (long)H * W, y_bs, a.y_chmax, st, false); if (rc2) return rc2; }
    }
  }
  return 0;
}

extern "C" int irr_conv2d_fwd_x3(const float* x, const void* wq, const float* bias, const float* res, float* y, int B,
                                 int Cin, int H, int W, int Cout, int dil, long x_bs, long y_bs, long res_bs, int lrelu,
                                 float alpha, int accumulate, const float* mask, long mask_bs, int nmask, void* stream) {
  return fwd_x3_impl(x, wq, bias, res, y, B, Cin, H, W, Cout, dil, x_bs, y_bs, res_bs, lrelu, alpha, accumulate, mask, mask_bs, nmask,
                     nullptr, 0, stream);
}

extern "C" int irr_conv2d_fwd_x3_dual(const float* x, const void* wq, const float* bias, const float* res, float* y, float* y2, int B,
                                      int Cin, int H, int W, int Cout, int dil, long x_bs, long y_bs, long res_bs, long y2_bs,
                                      int lrelu, float alpha, void* stream) {
  if (!y2 || !res) return IRR_EINVAL;
  return fwd_x3_impl(x, wq, bias, res, y, B, Cin, H, W, Cout, dil, x_bs, y_bs, res_bs, lrelu, alpha, 0, nullptr, 0, 0, nullptr, 0, stream,
                     y2, y2_bs);
}

extern "C" int irr_conv2d_fwd_x3_splitk(const float* x, const void* wq, const float* bias, const float* res, float* y, int B,
                                        int Cin, int H, int W, int Cout, int dil, long x_bs, long y_bs, long res_bs, int lrelu,
                                        float alpha, int accumulate, const float* mask, long mask_bs, int nmask, float* ws,
                                        long ws_elems, void* stream) {
  return fwd_x3_impl(x, wq, bias, res, y, B, Cin, H, W, Cout, dil, x_bs, y_bs, res_bs, lrelu, alpha, accumulate, mask, mask_bs, nmask,
                     ws, ws_elems, stream);
}

// irr_conv2d_fwd_x3_splitk on the fp16x2 form (ws may be null: the problem then runs unsplit): wq from irr_conv_pack_weights_h2,
// x_amax[0 .. n_amax) = device slots whose maximum bounds |x|, y_amax (nullable) receives max |y| (atomic max: pre-set to 0).
extern "C" int irr_conv2d_fwd_h2(const float* x, const void* wq, const float* bias, const float* res, float* y, int B,
                                 int Cin, int H, int W, int Cout, int dil, long x_bs, long y_bs, long res_bs, int lrelu,
                                 float alpha, int accumulate, const float* mask, long mask_bs, int nmask, float* ws,
                                 long ws_elems, const float* x_amax, int n_amax, float* y_amax, void* stream) {
  return fwd_x3_impl(x, wq, bias, res, y, B, Cin, H, W, Cout, dil, x_bs, y_bs, res_bs, lrelu, alpha, accumulate, mask, mask_bs, nmask,
                     ws, ws_elems, stream, nullptr, 0, 2, x_amax, n_amax, y_amax);
}

// (ABI 12) irr_conv2d_fwd_h2 whose K-split launches (the small pyramid levels) finish INSIDE the launch: kcnt = kcnt_elems >=
// irr_conv2d_fwd_x3_kcounters(...) ZEROED 32-bit counters (they are zero again when the launch has finished: atomicInc wraps).  The block that
// arrives last at a pixel tile sums the slices' partial images in slice order and runs the epilogue (bias, LeakyReLU, residual, accumulate,
// mask, magnitude folds): no finishing launch, results bit-identical to irr_conv2d_fwd_h2.  kcnt == NULL: irr_conv2d_fwd_h2.
extern "C" long irr_conv2d_fwd_x3_kcounters(int B, int Cin, int H, int W, int Cout, int dil) {
  if (B <= 0 || Cin < 16 || Cout <= 0 || H <= 0 || W <= 0 || dil < 1 || x3s_ok(B, Cin, H, W, Cout, dil)) return 0;
  Plan p;
  if (!make_plan(B, Cin, H, W, Cout, dil, &p) || p.ksplit <= 1 || p.t.nt > 4) return 0;
  return p.blocks;
}

extern "C" int irr_conv2d_fwd_h2_kfused(const float* x, const void* wq, const float* bias, const float* res, float* y, int B,
                                        int Cin, int H, int W, int Cout, int dil, long x_bs, long y_bs, long res_bs, int lrelu,
                                        float alpha, int accumulate, const float* mask, long mask_bs, int nmask, float* ws,
                                        long ws_elems, void* kcnt, long kcnt_elems, const float* x_amax, int n_amax, float* y_amax,
                                        void* stream) {
  if (((uintptr_t)kcnt) & 3) { g_next_chmax = nullptr; return IRR_EINVAL; }
  return fwd_x3_impl(x, wq, bias, res, y, B, Cin, H, W, Cout, dil, x_bs, y_bs, res_bs, lrelu, alpha, accumulate, mask, mask_bs, nmask,
                     ws, ws_elems, stream, nullptr, 0, 2, x_amax, n_amax, y_amax, nullptr, nullptr, (unsigned int*)kcnt, kcnt_elems);
}

// irr_conv2d_fwd_h2 for the problems of the streaming kernel (irr_conv2d_h2_eligible == 9001, Cout <= 32) with LeakyReLU' masks as bits
extern "C" long irr_conv2d_x3s_mask_words(int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return IRR_EINVAL;
  return (long)B * ((W + 31) / 32) * ((H + 7) / 8) * 256;
}

extern "C" int irr_conv2d_fwd_h2_bits(const float* x, const void* wq, const float* bias, const float* res, float* y, int B,
                                      int Cin, int H, int W, int Cout, int dil, long x_bs, long y_bs, long res_bs, int lrelu,
                                      float alpha, int accumulate, const void* mask_bits, int nmask, void* bits_out,
                                      const float* x_amax, int n_amax, float* y_amax, void* stream) {
  if ((!mask_bits && !bits_out) || (((uintptr_t)mask_bits | (uintptr_t)bits_out) & 3)) {
    g_next_chmax = nullptr;                                  // (a rejected launch consumes the one-shot too: it must not reach a later one)
    return IRR_EINVAL;
  }
  return fwd_x3_impl(x, wq, bias, res, y, B, Cin, H, W, Cout, dil, x_bs, y_bs, res_bs, lrelu, alpha, accumulate, nullptr, 0, nmask,
                     nullptr, 0, stream, nullptr, 0, 2, x_amax, n_amax, y_amax, (const uint32_t*)mask_bits, (uint32_t*)bits_out);
}

// irr_conv2d_fwd_x3_dual on the fp16x2 form (y_amax bounds y, the sum)
extern "C" int irr_conv2d_fwd_h2_dual(const float* x, const void* wq, const float* bias, const float* res, float* y, float* y2, int B,
                                      int Cin, int H, int W, int Cout, int dil, long x_bs, long y_bs, long res_bs, long y2_bs,
                                      int lrelu, float alpha, const float* x_amax, int n_amax, float* y_amax, void* stream) {
  if (!y2 || !res) {
    g_next_chmax = nullptr;                                  // (see irr_conv2d_fwd_h2_bits)
    return IRR_EINVAL;
  }
  return fwd_x3_impl(x, wq, bias, res, y, B, Cin, H, W, Cout, dil, x_bs, y_bs, res_bs, lrelu, alpha, 0, nullptr, 0, 0, nullptr, 0, stream,
                     y2, y2_bs, 2, x_amax, n_amax, y_amax);
}

// The NEXT irr_conv2d_fwd_h2 / _h2_bits / _h2_dual launch of the calling thread (forward or data gradient; since the end of round 6 the
// streaming 32-channel kernel too) additionally folds max |y[:, co]| of what it stores into chmax[co], co < Cout (atomic max on
// non-negative bit patterns: pre-set the slots to 0; a launch that accumulates into y folds the accumulated values).  One-shot:
// consumed by that launch.  The scales of the weight gradient's gy-role operand (irr_conv2d_wgrad_h2_ch) without a pass over it.
extern "C" int irr_conv_x3_next_chmax(float* chmax) {
  g_next_chmax = chmax;
  return 0;
}
