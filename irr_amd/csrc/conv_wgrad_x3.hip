// fp32-faithful weight gradient of the 3x3 / stride-1 / dilation-1 conv() blocks on the bf16 matrix pipe of gfx950
// (same exact 3-way bf16 operand split and six-product accumulation as conv_x3.hip):
//   dW[co][ci][dy][dx] += sum_{b,y,x} gy[b,co,y,x] * x[b,ci,y+dy-1,x+dx-1]
//
// GEMM view per tap: M = co, N = ci, K = pixels.  v_mfma_f32_32x32x16_bf16 wants 8 CONSECUTIVE k-values per lane, and
// NCHW planes are pixel-contiguous, so a lane's fragment is simply one "group" of 8 consecutive pixels of one channel
// row -- no transpose.  The +-1 column taps are the same group shifted by one pixel (2 bytes): built in registers
// with five v_alignbit per piece from the group and one neighbour pixel on each side.
//   * block = MW co-tile waves x NW ci-tile waves (8 waves); wave (m, n) keeps all 9 taps of its 32x32 (co, ci) tile in
//     144 accumulator registers.
//   * the pixel axis is walked in UNITS of R rows x KG groups (a column strip of the image); per unit the block stages the
//     gy patch and the R new x rows (plus one margin group left/right) ONCE: coalesced 32-B runs per lane, exact split
//     into three bf16 pieces in registers, written to LDS as [piece][channel][row][group] x 16 B with a channel pitch of
//     16 B x odd (conflict-free channel-strided ds_read_b128).  x rows live in a ring of 2R+2 rows, so each row is staged
//     once and used for the three vertical taps; out-of-image rows/margins are zeros from the buffer bounds check.
//   * the next unit's loads are in flight during the MFMAs of the current one (one barrier per unit).
//   * a block walks several columns with persistent accumulators and flushes once: coalesced fp32 atomics into the
//     [co][tap][ci] workspace that wgrad_unpack_kernel (conv_wgrad.hip) folds into dW; the bias gradient is summed by
//     the gy stagers (fixed channel per thread) and added once per block.
#include "x3_split.h"
#include "wgrad_reduce.h"
#include <stdlib.h>
#include <type_traits>

#ifndef WX3_STAGGER
#define WX3_STAGGER 0  // 1 (A/B): the two waves of a SIMD publish / compute in opposite orders (see the unit loop); 0: all compute first
#endif
#ifndef WX3_STAGE_OLD
#define WX3_STAGE_OLD 1  // 1: in eight-wave blocks only waves 0..3 stage (see SWAVES); 0 (A/B): all waves stage
#endif
#ifndef WX3_MINI
#define WX3_MINI 1     // 1: DIL == 1 stages ONE pixel per side of a strip row instead of a whole margin group (see MINI); 0 (A/B): groups
#endif
#ifndef WX3_TALL
#define WX3_TALL 1     // 1: DIL == 1 walks all samples of a strip as one tall image (see TALL); 0 (A/B): one column per sample
#endif
#ifndef WX3_STRIP_FAST
#define WX3_STRIP_FAST 1   // 1: tall columns are ordered strip-fastest (L2 sharing of x lines between neighbouring strips); 0 (A/B)
#endif
#ifndef WX3_ABL
#define WX3_ABL 0      // ablation builds (timing only): 1 = no operand split (VALU) in the staging path
#endif

#ifdef WX3_TRACE
static unsigned long long* g_wx3_dbg = nullptr;      // s_memtime trace (IRR_WX3_TRACE=1 builds, tools/wx3_trace.py)
#endif

namespace {

constexpr uint32_t OOB = 0x80000000u;

struct WX3Args {
  const float* x;
  const float* gy;
  float* ws;                     // [grid.x][Cout][9][Cin] workspace: one partial result per block column (plain stores)
  long n;                        // Cout * 9 * Cin
  float* gbias;                  // nullable: bias gradient = sum of the gy-role operand
  float* xbias;                  // nullable (NW == 1 instantiations): the same sum over the x-role operand -- launches with exchanged
                                 // roles carry the real output gradient there (it used to take a separate pass over gy per launch)
  float alpha;
  int B, Cin, H, W, Cout;
  long x_bs, gy_bs;
  int nstrips, nchunks_y, rows_per_chunk;      // column = (b, strip, row chunk)
  int tall;                                    // 1: columns = (strip, chunk of the B * (H + 1)-row tall image), see TALL
  long ncols;
  int cols_per_block;
  unsigned long long* dbg;       // WX3_TRACE builds only
  int ngx, ngy, ngz;             // logical grid: block columns x ci tiles x co tiles (launched as a 1-D grid, see the kernel)
  // H2 instantiations (fp16x2 of scaled operands, x3_split.h): device slots whose maxima bound |x| and |gy| (kernel roles)
  const float* x_amax;
  const float* g_amax;
  int nx_amax, ng_amax;
  // round 6: nullable -- max |.| of EVERY channel of the gy-role operand (a.Cout floats).  The plain fp16 pair of that operand carries
  // full precision only within 2^17 of its scale: with one scale per CHANNEL (= per output row of dW, undone per row at the flush) a
  // channel that is 1e-6 of the tensor's loudest keeps fp32 accuracy relative to its own range (Adam divides every
  // element by its own sqrt(v), so a quiet filter's gradient row matters as much as a loud one's).
  const float* g_chmax;
};


__device__ __forceinline__ uint32_t alignbit16(uint32_t hi, uint32_t lo) { return (lo >> 16) | (hi << 16); }

// KW > 1 (the 32 -> 32 layers: a single (co, ci) tile): KW wave groups share the staged unit and split its k-steps;
// their accumulators are folded through LDS before the flush.
// DIL > 1 (dilated context-network layers): the three vertical taps are DIL rows apart, so a column additionally fixes a
// row residue and walks rows res, res + DIL, res + 2 DIL, ...: in that walk the taps are again neighbouring rows and the
// ring works unchanged.  The +-DIL column taps are whole dwords (DIL = 2, 4) or whole groups (8, 16) of the neighbours.
// H2: the operands as two fp16 pieces of x * 2^ex / gy * 2^eg (three piece products per k-step instead of six); the flush scales back.
template <int MW, int NW, int KG, int R, int KW = 1, int DIL = 1, bool NARROW = false, bool H2 = false>
__global__ __launch_bounds__(MW* NW * KW * 64) void conv_wgrad_x3_kernel(const WX3Args a) {
  constexpr int NP = H2 ? 2 : 3;                            // pieces per operand
  constexpr int NTHR = MW * NW * KW * 64;
  constexpr int RING = 2 * R + 2;                          // x rows resident
  constexpr int MG = DIL > 8 ? DIL / 8 : 1;                // margin groups on each side of a staged x row
  constexpr int XG = KG + 2 * MG;                          // groups per staged x row
  constexpr int XPITCH = RING * XG + 1;                    // 16-B units per input channel (odd: conflict-free channel stride)
  constexpr int GPITCH = 2 * R * KG + 1;                   // 16-B units per output channel (two unit buffers)
  constexpr int XPLANE = 32 * NW * XPITCH;                 // 16-B units per piece
  constexpr int GPLANE = 32 * MW * GPITCH;
  // MINI (DIL == 1): the +-1 column taps need ONE pixel from each neighbour of a strip row (the high half of dword 3 of the left
  // margin group, the low half of dword 0 of the right one).  Staging whole 32-B margin groups for them doubled the x traffic and
  // the x split work at KG = 2 (the 96x112 level); a margin is now one dword load, a one-value split and three 4-B LDS stores
  // into the same slots -- the MFMA waves read exactly what they read before.
  constexpr bool MINI = WX3_MINI && DIL == 1;
  // TALL (DIL == 1): a column does not stop at the end of a sample.  The B samples of a strip are walked as ONE image of
  // B * (H + 1) rows -- sample b's row y is tall row b * (H + 1) + y, the row between two samples reads as zeros (it is the lower
  // padding row of one sample and the upper one of the next; its gy row is zero too) -- so the ring never has to be refilled:
  // one prologue per block instead of one per (sample, strip) (at 48x56 a column was 12 units + a 2-unit prologue with its loads
  // exposed), and the vertical split into chunks can balance the blocks' shares to a unit.
  // (dilated layers: the same with the rows of ONE residue class -- sample b's k-th row of the class is tall row b * (Hk + 1) + k)
  const bool TALL = WX3_TALL && a.tall;                   // (chosen per launch by the host's cost model, see launch_wx3)
  constexpr int XGS = MINI ? KG : XG;                      // groups of a row staged as full 32-B units
  constexpr int GOFS = MINI ? MG : 0;                      // ... stored from this group slot on
  constexpr int XUNITS = R * XGS * 32 * NW;                // (channel, row, group) staging units per step
  constexpr int MUNITS = MINI ? R * 2 * 32 * NW : 0;       // (channel, row, side) one-pixel margin units per step
  constexpr int GUNITS = R * KG * 32 * MW;
  // Staging waves: with two waves per SIMD the OLDER wave of a SIMD (0..3) runs its MFMAs first and then waits at the barrier
  // while the younger one (4..7), starved until then, computes (tools/wx3_trace.py) -- so the older waves do ALL the staging
  // (issue + split + publish) in that wait and the younger ones go straight from their MFMAs to the barrier.
  // (only where the doubled staging registers fit: at most four 32-B units per staging thread)
  constexpr int SWAVES = (WX3_STAGE_OLD && MW * NW * KW == 8 && (XUNITS + 255) / 256 + (GUNITS + 255) / 256 <= 4) ? 4 : MW * NW * KW;
  constexpr int SNTHR = SWAVES * 64;
  constexpr int XR = (XUNITS + SNTHR - 1) / SNTHR;
  constexpr int MR = (MUNITS + SNTHR - 1) / SNTHR;
  constexpr int GR = (GUNITS + SNTHR - 1) / SNTHR;
  constexpr int NK = R * KG / 2;                           // MFMA k-steps (16 pixels = two groups) per unit
  static_assert((R * KG) % 2 == 0, "a unit must hold an even number of 8-pixel groups");
  static_assert(NK % KW == 0, "the k-steps of a unit must divide evenly among the wave groups");
  extern __shared__ u32x4 lds[];
  u32x4* const xs = lds;                                   // [NP][32*NW][XPITCH]
  u32x4* const gs = lds + NP * XPLANE;                     // [NP][32*MW][GPITCH]
  float sx = 1.f, sg = 1.f, unscale_x = 1.f, unscale_g = 1.f;
  if (H2) {
    const int ex = x3_h2_exp(x3_h2_amax(a.x_amax, a.nx_amax)), eg = x3_h2_exp(x3_h2_amax(a.g_amax, a.ng_amax));
    sx = ldexpf(1.f, ex); unscale_x = ldexpf(1.f, -ex);
    sg = ldexpf(1.f, eg); unscale_g = ldexpf(1.f, -eg);
  }

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef WX3_TRACE
  int ntr = 0;
  unsigned long long* dbgp = a.dbg + (size_t)wave * 400;
#define WTR(slot) do { if (blockIdx.x == 16 && lane == 0 && ntr < 400) { dbgp[ntr++] = ((unsigned long long)(slot) << 56) | (__builtin_amdgcn_s_memtime() & 0xffffffffffffffull); } } while (0)
#else
#define WTR(slot) do {} while (0)
#endif
  const int wm = wave % MW, wn = (wave / MW) % NW, wk = wave / (MW * NW);
  const int j = lane & 31, g = lane >> 5;
  // Blocks with the same pixel columns (bx) and different channel tiles (by, bz) stage the same x / gy data: decode the
  // XCD-major position with the channel tiles fastest, so they share one XCD's L2 (gy used to be fetched once per ci tile
  // row by up to eight L2s: 3.2 GB of HBM reads per launch for 1.9 GB of operands on the 565 -> 128 level-4 layer).
  // (1-D launch: the XCD placement of a workgroup is only documented / observed for the linear id of a 1-D grid)
  const unsigned nyz = (unsigned)(a.ngy * a.ngz);
  const unsigned pos = irr_xcd_order(blockIdx.x, (unsigned)a.ngx * nyz);
  const unsigned bx = pos / nyz, byz = pos - bx * nyz;
  const unsigned by = byz % (unsigned)a.ngy, bz = byz / (unsigned)a.ngy;
  const int ci0 = by * 32 * NW, co0 = bz * 32 * MW;
  const long hw = (long)a.H * a.W;

  const uint32_t x_bytes = 0x80000000u, g_bytes = 0x80000000u;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, (short)0, (int)x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc((void*)a.gy, (short)0, (int)g_bytes, 0x00020000);

  // ---- staging roles (fixed per thread) ----
  // one register per role: channel | row << 8 | group << 12 (negative: no unit) -- the unit loop runs at 246-256 registers
  int xu[XR], gu[GR], mu[MR > 0 ? MR : 1];
#pragma unroll
  for (int r = 0; r < XR; ++r) {
    const int u = r * SNTHR + tid;
    xu[r] = (u < XUNITS && tid < SNTHR) ? (u / (XGS * R)) | (((u / XGS) % R) << 8) | ((u % XGS) << 12) : -1;
  }
#pragma unroll
  for (int r = 0; r < GR; ++r) {
    const int u = r * SNTHR + tid;
    gu[r] = (u < GUNITS && tid < SNTHR) ? (u / (KG * R)) | (((u / KG) % R) << 8) | ((u % KG) << 12) : -1;
  }
#pragma unroll
  for (int r = 0; r < MR; ++r) {                            // "group" = side: 0 left, 1 right
    const int u = r * SNTHR + tid;
    mu[r] = (u < MUNITS && tid < SNTHR) ? (u / (2 * R)) | (((u >> 1) % R) << 8) | ((u & 1) << 12) : -1;
  }
#define RU_CH(v) ((v) & 255)
#define RU_RR(v) (((v) >> 8) & 15)
#define RU_GRP(v) ((v) >> 12)
  float bsum[GR];
#pragma unroll
  for (int r = 0; r < GR; ++r) bsum[r] = 0.f;
  float sgc[GR];                                            // H2: the scale of each gy staging role (its channel's own, or the tensor's)
#pragma unroll
  for (int r = 0; r < GR; ++r) {
    sgc[r] = sg;
    if (H2 && a.g_chmax && gu[r] >= 0 && co0 + RU_CH(gu[r]) < a.Cout) sgc[r] = ldexpf(1.f, x3_h2_exp(a.g_chmax[co0 + RU_CH(gu[r])]));
  }
  constexpr bool XB = NW == 1;                              // (only the one-tile-wide blocks ever run with exchanged roles)
  float xsum[XB ? XR : 1];
#pragma unroll
  for (int r = 0; r < (XB ? XR : 1); ++r) xsum[r] = 0.f;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  f32x4 xraw[XR][2], graw[GR][2];
  float mraw[MR > 0 ? MR : 1];

  // issue the global loads of x rows [row0, row0+R) and (optionally) gy rows [grow0, grow0+R) of column (b, strip c0)
  // (row0 / grow0 count rows of the residue walk: image row = res + DIL * k)
  const bool stager = wave < SWAVES;
  int Hp = a.H + 1;                                        // rows per sample in the tall image (set per column when dilated)
  // NARROW is a TEMPLATE parameter on purpose: as a run-time branch around the loads it cost every variant its graded
  // vmcnt(7 .. 0) waits (the compiler cannot count loads behind a branch and waited with vmcnt(0): the two-deep prefetch of the
  // unit loop was gone -- dilation-16 layers -21 %, the Cout = 64 layers -7 %).
  constexpr bool narrow = NARROW;
  int trow = 0, tb = 0, ty = 0;                            // TALL: tall row trow is row ty of sample tb
  int col_row_lo = 0, col_row_hi = 0;                      // the current column's own rows [ya, yb) (x rows outside are halo)
  auto issue = [&](int b, int c0, int res, int row0, bool with_x, int grow0, bool with_g) {
    if (!stager) return;
    // TALL: (sample, row) of the first staged x / gy row (uniform); a thread's row is at most R - 1 further down: one wrap
    // (tracked from (tb, ty) = sample and row of tall row trow, which the unit loop steps: no division per unit)
    int xb = b, xy0 = row0, gb0 = b, gy0 = grow0;
    if (TALL) {
      xb = tb; xy0 = ty + (row0 - trow);
      while (xy0 >= Hp) { xy0 -= Hp; ++xb; }               // (uniform; one or two turns: the rows are at most 2R + 1 <= 9 ahead)
      if (xy0 < 0) xy0 = -1;                               // the row above the tracked one: a separator / the top padding row
      gb0 = tb; gy0 = ty + (grow0 - trow);
      while (gy0 >= Hp) { gy0 -= Hp; ++gb0; }
    }
#pragma unroll
    for (int r = 0; r < MR; ++r) {
      int kk = xy0 + RU_RR(mu[r]), bb = xb;
      if (TALL && kk >= Hp) { kk -= Hp; ++bb; }
      const int yy = kk < 0 ? -1 : res + DIL * kk;           // (TALL: kk = Hk, the separator, lands beyond the image)
      const int xx = RU_GRP(mu[r]) ? c0 + KG * 8 : c0 - 1;
      const int ci = ci0 + RU_CH(mu[r]);
      const bool ok = with_x && mu[r] >= 0 && ci < a.Cin && yy >= 0 && yy < a.H && bb < a.B && xx >= 0 && xx < a.W;
      const uint32_t vo = ok ? (uint32_t)(((long)bb * a.x_bs + (long)ci * hw + (long)yy * a.W + xx) * 4) : OOB;
      mraw[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, (int)vo, 0, 0));
    }
#pragma unroll
    for (int r = 0; r < XR; ++r) {
      int kk = xy0 + RU_RR(xu[r]), bb = xb;
      if (TALL && kk >= Hp) { kk -= Hp; ++bb; }
      const int yy = kk < 0 ? -1 : res + DIL * kk;
      const int xx = c0 + (RU_GRP(xu[r]) + GOFS - MG) * 8;
      const int ci = ci0 + RU_CH(xu[r]);
      const bool ok = with_x && xu[r] >= 0 && ci < a.Cin && yy >= 0 && yy < a.H && bb < a.B && xx >= 0 && xx < a.W;
      const uint32_t vo = ok ? (uint32_t)(((long)bb * a.x_bs + (long)ci * hw + (long)yy * a.W + xx) * 4) : OOB;
      if (!narrow) {
        // (W % 4 == 0: a group may straddle the end of the row -- its second half then reads zeros)
        xraw[r][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, (int)vo, 0, 0));
        xraw[r][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, (int)((ok && xx + 4 < a.W) ? vo + 16 : OOB), 0, 0));
      } else {
        // rows that are not a multiple of four pixels (the 6x7 / 12x14 pyramid levels): element-wise loads, every pixel beyond
        // the row end answered with 0 by the bounds check (and never touched: the last row may end the tensor)
#pragma unroll
        for (int e = 0; e < 8; ++e)
          xraw[r][e >> 2][e & 3] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, (int)((ok && xx + e < a.W) ? vo + 4 * e : OOB), 0, 0));
      }
    }
#pragma unroll
    for (int r = 0; r < GR; ++r) {
      int kk = gy0 + RU_RR(gu[r]), bb = gb0;
      if (TALL && kk >= Hp) { kk -= Hp; ++bb; }
      const int yy = res + DIL * kk;
      const int xx = c0 + RU_GRP(gu[r]) * 8;
      const int co = co0 + RU_CH(gu[r]);
      const bool ok = with_g && gu[r] >= 0 && co < a.Cout && yy >= 0 && yy < a.H && bb < a.B && xx < a.W;
      const uint32_t vo = ok ? (uint32_t)(((long)bb * a.gy_bs + (long)co * hw + (long)yy * a.W + xx) * 4) : OOB;
      if (!narrow) {
        graw[r][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(gr, (int)vo, 0, 0));
        graw[r][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(gr, (int)((ok && xx + 4 < a.W) ? vo + 16 : OOB), 0, 0));
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          graw[r][e >> 2][e & 3] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(gr, (int)((ok && xx + e < a.W) ? vo + 4 * e : OOB), 0, 0));
      }
    }
  };
  // split the loaded units and publish them: x rows [row0, row0+R) into their ring slots, gy rows into unit buffer gbuf
  auto publish = [&](int row0, bool with_x, int gbuf, bool with_g) {
    if (!stager) return;
    if (with_x) {
#pragma unroll
      for (int r = 0; r < MR; ++r) {
        if (mu[r] < 0) continue;
        // one value -> three bf16 pieces (same rounding sequence as split8), each into its half-dword of the margin slot
        const float v0 = mraw[r];
        uint32_t hp, mp, lp = 0;
        if (H2) {
          split1_h2<true>(v0, sx, hp, mp);
        } else {
          hp = pk_bf16(v0, 0.f);
          float r0 = v0 - lo_f(hp);
          asm volatile("" : "+v"(r0));
          mp = pk_bf16(r0, 0.f);
          float s0 = r0 - lo_f(mp);
          asm volatile("" : "+v"(s0));
          lp = pk_bf16(s0, 0.f);
        }
        const int side = RU_GRP(mu[r]);
        const int slot = (row0 + RU_RR(mu[r]) + RING) % RING;
        const int idx = RU_CH(mu[r]) * XPITCH + slot * XG + (side ? KG + MG : MG - 1);
        uint32_t* const w = (uint32_t*)xs + idx * 4 + (side ? 0 : 3);
        const int sh = side ? 0 : 16;                        // left margin: pixel c0-1 is the HIGH half of dword 3
        w[0] = (hp & 0xffffu) << sh;
        w[XPLANE * 4] = (mp & 0xffffu) << sh;
        if (NP == 3) w[2 * XPLANE * 4] = (lp & 0xffffu) << sh;
      }
#pragma unroll
      for (int r = 0; r < XR; ++r) {
        if (xu[r] < 0) continue;
        float v[8] = {xraw[r][0][0], xraw[r][0][1], xraw[r][0][2], xraw[r][0][3], xraw[r][1][0], xraw[r][1][1], xraw[r][1][2], xraw[r][1][3]};
        // (every own row of the column once: the prologue's second round and the first unit's rows overlap, halo rows belong
        // to the neighbouring chunks)
        if (XB && a.xbias && row0 + RU_RR(xu[r]) >= col_row_lo && row0 + RU_RR(xu[r]) < col_row_hi && (!MINI ? (RU_GRP(xu[r]) >= MG && RU_GRP(xu[r]) < MG + KG) : true))
          xsum[r] += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        u32x4 h, m, l;
#if WX3_ABL == 1
        h = __builtin_bit_cast(u32x4, xraw[r][0]); m = __builtin_bit_cast(u32x4, xraw[r][1]); l = h; (void)v;
#else
        if (H2) split8_h2<true>(v, sx, h, m);      // x role: low piece x 2^11 (x3_split.h, "Range")
        else split8(v, h, m, l);
#endif
        const int slot = (row0 + RU_RR(xu[r]) + RING) % RING;      // rows >= -1
        const int idx = RU_CH(xu[r]) * XPITCH + slot * XG + RU_GRP(xu[r]) + GOFS;
        xs[idx] = h;
        xs[idx + XPLANE] = m;
        if (NP == 3) xs[idx + 2 * XPLANE] = l;
      }
      if (XB && row0 + R > col_row_lo) col_row_lo = row0 + R;      // rows below are counted
    }
    if (with_g) {
#pragma unroll
      for (int r = 0; r < GR; ++r) {
        if (gu[r] < 0) continue;
        float v[8] = {graw[r][0][0], graw[r][0][1], graw[r][0][2], graw[r][0][3], graw[r][1][0], graw[r][1][1], graw[r][1][2], graw[r][1][3]};
        bsum[r] += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        u32x4 h, m, l;
#if WX3_ABL == 1
        h = __builtin_bit_cast(u32x4, graw[r][0]); m = __builtin_bit_cast(u32x4, graw[r][1]); l = h;
#else
        if (H2) split8_h2(v, sgc[r], h, m);
        else split8(v, h, m, l);
#endif
        const int idx = RU_CH(gu[r]) * GPITCH + (gbuf * R + RU_RR(gu[r])) * KG + RU_GRP(gu[r]);
        gs[idx] = h;
        gs[idx + GPLANE] = m;
        if (NP == 3) gs[idx + 2 * GPLANE] = l;
      }
    }
  };

  const int a_base = (wm * 32 + j) * GPITCH;               // this lane's gy channel row
  const int b_base = (wn * 32 + j) * XPITCH;               // this lane's x channel row

  // the 54 MFMAs x NK of one unit: gy rows of unit buffer gbuf against x rows y-1 .. y+R.
  // Flat software pipeline over stages (ks, dy, B piece q): the LDS reads of stage s+1 are issued ahead of the MFMAs of
  // stage s; sched_barriers keep the compiler from hoisting more than that (144 of the 256 registers are accumulators).
  // (Round 3 measured nine-MFMA stages with three B-piece buffers -- every read issued >= 288 cycles ahead of its use instead of
  // 96 .. 288: 2-4 % SLOWER on every shape.  LDS latency is not what holds a lone wave at 70 % MFMA density)
  auto compute = [&](int y, int gbuf) {
    constexpr int NKW = NK / KW;                           // this wave's k-steps: wk, wk + KW, ...
    u32x4 af[NP];
    // ring slot of x row (y - 1 + k), k = row + dy in [0, R + 2): ONE division per unit, then compare-and-subtract (round 6,
    // profiles/r6_pmc_wgrad.txt: 1.9-3.6 scalar instructions per MFMA -- the division-by-constant sequence used to run in every stage)
    const int ring0 = (y - 1 + RING) % RING;
    auto b_index = [&](int ki, int dy, int q) {
      const int ks = wk + KW * ki;
      const int gi = 2 * ks + g;
      const int row = gi / KG, grp = gi - row * KG;
      int slot = ring0 + row + dy;                           // < 2 RING: row + dy <= R + 1 < RING
      slot = slot >= RING ? slot - RING : slot;
      return b_base + slot * XG + grp + MG + q * XPLANE;
    };
    auto read_a1 = [&](int ki, int p) {
      const int ks = wk + KW * ki;
      const int gi = 2 * ks + g;
      const int row = gi / KG, grp = gi - row * KG;
      af[p] = gs[a_base + (gbuf * R + row) * KG + grp + p * GPLANE];
    };
    constexpr int SPK = 3 * NP;                            // stages per k-step: (dy, B piece q)
    constexpr int NS = NKW * SPK;
    u32x4 ob[2], lb[2], rb[2];                             // the group and its left / right neighbours (DIL <= 2: one dword each)
    auto read_b = [&](int sel, int st) {
      const int xi = b_index(st / SPK, (st % SPK) / NP, st % NP);
      ob[sel] = xs[xi];
      if (WX3_ABL == 2) {
      } else if (DIL <= 2) {
        lb[sel][3] = xs[xi - 1][3];
        rb[sel][0] = xs[xi + 1][0];
      } else if (DIL == 4) {
        lb[sel][2] = xs[xi - 1][2];
        lb[sel][3] = xs[xi - 1][3];
        rb[sel][0] = xs[xi + 1][0];
        rb[sel][1] = xs[xi + 1][1];
      } else {
        lb[sel] = xs[xi - MG];
        rb[sel] = xs[xi + MG];
      }
    };
    u32x4 afdn;                                            // H2: af[0] * 2^-11
    auto read_a = [&](int ki) {
#pragma unroll
      for (int p = 0; p < NP; ++p) read_a1(ki, p);
      if (H2) afdn = h2_hi_down(af[0]);
    };
    read_a(0);
    read_b(0, 0);
#pragma unroll
    for (int st = 0; st < NS; ++st) {
      const int cur = st & 1;
      const int dy = (st % SPK) / NP, q = st % NP;
      if (st + 1 < NS) read_b(cur ^ 1, st + 1);
      __builtin_amdgcn_sched_barrier(0);
      const u32x4 o = ob[cur];
      u32x4 fm, fp;                                        // dx = -DIL / +DIL fragments
      if (WX3_ABL == 2) {                                  // ablation (timing only): no shifted fragments
        fm = o;
        fp = o;
      } else if (DIL == 1) {
        fm[0] = alignbit16(o[0], lb[cur][3]);
        fm[1] = alignbit16(o[1], o[0]);
        fm[2] = alignbit16(o[2], o[1]);
        fm[3] = alignbit16(o[3], o[2]);
        fp[0] = fm[1];
        fp[1] = fm[2];
        fp[2] = fm[3];
        fp[3] = alignbit16(rb[cur][0], o[3]);
      } else if (DIL == 2) {                               // two pixels = one dword
        fm = u32x4{lb[cur][3], o[0], o[1], o[2]};
        fp = u32x4{o[1], o[2], o[3], rb[cur][0]};
      } else if (DIL == 4) {
        fm = u32x4{lb[cur][2], lb[cur][3], o[0], o[1]};
        fp = u32x4{o[2], o[3], rb[cur][0], rb[cur][1]};
      } else {                                             // 8 / 16 pixels = one / two whole groups
        fm = lb[cur];
        fp = rb[cur];
      }
      // products of weight >= 2^-17: (A piece, B piece) in {(l,h),(m,h),(h,h),(m,m),(h,m),(h,l)}
#pragma unroll
      for (int pa = NP - 1; pa >= 0; --pa) {
        if (pa + q > (WX3_ABL == 20 ? 1 : NP - 1)) continue;      // (ablation 20, timing only: three products of two pieces)
        if constexpr (H2) {
          // (q == 1: the x fragments are the scaled-up low pieces -- their partner is the gy high piece times 2^-11)
          const u32x4 ap = q == 1 ? afdn : af[pa];
          acc[dy * 3 + 0] = mma_h(ap, fm, acc[dy * 3 + 0]);
          acc[dy * 3 + 1] = mma_h(ap, o, acc[dy * 3 + 1]);
          acc[dy * 3 + 2] = mma_h(ap, fp, acc[dy * 3 + 2]);
        } else {
          acc[dy * 3 + 0] = mma(af[pa], fm, acc[dy * 3 + 0]);
          acc[dy * 3 + 1] = mma(af[pa], o, acc[dy * 3 + 1]);
          acc[dy * 3 + 2] = mma(af[pa], fp, acc[dy * 3 + 2]);
        }
      }
      if (st % SPK == SPK - 1 && st + 1 < NS) read_a(st / SPK + 1);   // next k-step's gy fragments (after their last use)
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- walk this block's columns ----
  const long col_begin = (long)bx * a.cols_per_block;
  const long col_end = min(a.ncols, col_begin + a.cols_per_block);
  // column = (b, strip, res, chunk) in mixed radix, chunk fastest: decoded once, then stepped (the 64-bit divisions of a
  // per-column decode are ~2k cycles of scalar work on every wave)
  // Columns in mixed radix with the STRIP fastest, then chunk, residue, sample (WX3_STRIP_FAST; round 2 had the chunk fastest):
  // neighbouring strips share every 128-B line of x (a strip is 32 ... 128 B wide and reads one more pixel on each side);
  // consecutive columns sit on one XCD and walk the same rows at the same time, so those lines come out of that XCD's L2
  // instead of being fetched once per strip (565 -> 128 at 96x112: 6.6 -> 3.1 GB fetched per launch for 1.9 GB of operands).
  // Decoded once, then stepped (the 64-bit divisions of a per-column decode are ~2k cycles of scalar work on every wave).
  int chunk, res, strip, b;
  {
    long t = col_begin;
    if (WX3_STRIP_FAST) {
      strip = (int)(t % a.nstrips);
      t /= a.nstrips;
      chunk = (int)(t % a.nchunks_y);
      t /= a.nchunks_y;
      res = (int)(t % DIL);
      b = (int)(t / DIL);
    } else {
      chunk = (int)(t % a.nchunks_y);
      t /= a.nchunks_y;
      res = (int)(t % DIL);
      t /= DIL;
      strip = (int)(t % a.nstrips);
      b = (int)(t / a.nstrips);
    }
  }
  for (long col = col_begin; col < col_end; ++col) {
    if (col != col_begin) {
      if (WX3_STRIP_FAST) {
        if (++strip == a.nstrips) {
          strip = 0;
          if (++chunk == a.nchunks_y) {
            chunk = 0;
            if (++res == DIL) {
              res = 0;
              ++b;
            }
          }
        }
      } else if (++chunk == a.nchunks_y) {
        chunk = 0;
        if (++res == DIL) {
          res = 0;
          if (++strip == a.nstrips) {
            strip = 0;
            ++b;
          }
        }
      }
    }
    const int c0 = strip * KG * 8;
    const int Hk = (a.H - res + DIL - 1) / DIL;              // rows of this residue class in one sample
    Hp = Hk + 1;
    const int hk = TALL ? a.B * Hp : Hk;                     // rows of this walk (TALL: all samples, b stays 0)
    const int ya = chunk * a.rows_per_chunk;
    const int yb = min(hk, ya + a.rows_per_chunk);
    if (TALL) { trow = ya; tb = ya / Hp; ty = ya - tb * Hp; }
    col_row_lo = ya; col_row_hi = yb;
    // prologue: x rows ya-1 .. ya+R and the first gy unit
    __syncthreads();
    for (int r0 = ya - 1; r0 <= ya + R; r0 += R) {
      const bool first = r0 == ya - 1;
      issue(b, c0, res, r0, true, ya, first);
      publish(r0, true, 0, first);
    }
    __syncthreads();
    // Units are software-pipelined two deep: while unit u computes, unit u+1 is split and published (its loads were issued
    // one unit earlier) and the loads of unit u+2 are issued.  publish(u+1) writes the ring slots / gy buffer that unit u-1
    // read -- free since the barrier that ended unit u-1 -- and nothing that unit u reads, so inside a unit the two steps may
    // run in EITHER order.  Measured at 96x112x64 (TFLOP/s, one-deep -> two-deep): 565->128 206 -> 214, 467->64 159 -> 171,
    // 32->32 full resolution 139 -> 148.  Letting the two waves of a SIMD take OPPOSITE orders (WX3_STAGGER=1: one feeds the
    // matrix pipe while the other splits and writes) is 1-4 % slower than all waves computing first, so it is off.
    const bool pub_first = WX3_STAGGER && ((wave >> 2) & 1);
    if (ya + R < yb) issue(b, c0, res, ya + R + 1, true, ya + R, true);
    int gbuf = 0;
    for (int y = ya; y < yb; y += R) {
      const bool more = y + R < yb, more2 = y + 2 * R < yb;
      if (pub_first) {
        if (more) publish(y + R + 1, true, gbuf ^ 1, true);
        if (more2) issue(b, c0, res, y + 2 * R + 1, true, y + 2 * R, true);
      }
      __builtin_amdgcn_sched_barrier(0);
      WTR(1);
      compute(y, gbuf);                                      // (one copy of the MFMA sequence: 144 accumulators stay in place)
      __builtin_amdgcn_sched_barrier(0);
      WTR(2);
      if (!pub_first) {
        if (more) publish(y + R + 1, true, gbuf ^ 1, true);
        WTR(3);
        if (more2) issue(b, c0, res, y + 2 * R + 1, true, y + 2 * R, true);
      }
      WTR(4);
      __syncthreads();
      WTR(5);
      gbuf ^= 1;
      if (TALL) { trow += R; ty += R; if (ty >= Hp) { ty -= Hp; ++tb; } }
    }
  }

  if (KW > 1) {
    // fold the wave groups' accumulators: the upper half of the active groups parks its tiles in LDS (free after the
    // last unit), the lower half adds them; log2(KW) rounds, then group 0 flushes
    float* red = (float*)lds;                              // [group][tap][r][lane]
#pragma unroll
    for (int half = KW / 2; half >= 1; half >>= 1) {
      __syncthreads();
      if (wk >= half && wk < 2 * half) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) red[(((wk - half) * MW * NW + wn * MW + wm) * 144 + t * 16 + r) * 64 + lane] = acc[t][r];
      }
      __syncthreads();
      if (wk < half) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[t][r] += red[((wk * MW * NW + wn * MW + wm) * 144 + t * 16 + r) * 64 + lane];
      }
    }
    if (wk > 0) {
      // (bias partial sums below are per thread and do not depend on wk)
    }
  }
  // ---- flush: ws[blockIdx.x][co][tap][ci] = alpha * acc (lanes = ci: coalesced stores; wgrad_reduce_x3_kernel sums the
  // partials in a fixed order: no atomics, no zero-fill of the workspace, bit-reproducible weight gradients) ----
  float* const wsp = a.ws + (long)bx * a.n;
  if (!(KW > 1 && wk > 0)) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
      const int ci = ci0 + wn * 32 + j;
      // (the row's own scale, once per row: with the exponent arithmetic inside the tap loop the flush of the short launches of the
      // small pyramid levels took longer than their pixel walk)
      float ug = unscale_g;
      if (H2 && a.g_chmax && co < a.Cout) ug = ldexpf(1.f, -x3_h2_exp(a.g_chmax[co]));
      const float us = H2 ? a.alpha * ug : a.alpha;
      if (co < a.Cout && ci < a.Cin) {
#pragma unroll
        for (int t = 0; t < 9; ++t) wsp[((long)co * 9 + t) * a.Cin + ci] = H2 ? (acc[t][r] * unscale_x) * us : us * acc[t][r];
      }
    }
  }
  if constexpr (XB) if (a.xbias && bz == 0) {
    // threads with the same x channel are adjacent (R * XGS of them): fold, then one atomic per channel
#pragma unroll
    for (int r = 0; r < XR; ++r) {
      float s2 = xsum[r];
      constexpr int PERX = R * XGS;
      constexpr int P2 = PERX <= 1 ? 1 : PERX <= 2 ? 2 : PERX <= 4 ? 4 : PERX <= 8 ? 8 : 16;
      static_assert(!MINI || PERX == P2, "x units per channel must be a power of two for the shuffle fold");
      if (PERX != P2) __builtin_trap();                      // (group-margin builds of the one-tile-wide blocks: not a product configuration)
#pragma unroll
      for (int off = 1; off < PERX; off <<= 1) s2 += __shfl_xor(s2, off, 64);
      const int u = r * SNTHR + tid;
      if (xu[r] >= 0 && (u % PERX) == 0 && ci0 + RU_CH(xu[r]) < a.Cin) unsafeAtomicAdd(a.xbias + ci0 + RU_CH(xu[r]), a.alpha * s2);
    }
  }
  if (a.gbias && by == 0) {
    // threads with the same gy channel are adjacent (R*KG of them): fold, then one atomic per channel
#pragma unroll
    for (int r = 0; r < GR; ++r) {
      float s = bsum[r];
      constexpr int PER = R * KG;
#pragma unroll
      for (int off = 1; off < PER; off <<= 1) s += __shfl_xor(s, off, 64);
      const int u = r * SNTHR + tid;
      if (gu[r] >= 0 && (u % PER) == 0 && co0 + RU_CH(gu[r]) < a.Cout) unsafeAtomicAdd(a.gbias + co0 + RU_CH(gu[r]), a.alpha * s);
    }
  }
}

thread_local int g_last_parts = 0;   // grid.x of this thread's last launch_wx3 (host-side hand-over to the reduce launch)

int g_cu_count = 0;
int cu_count() {
  if (!g_cu_count) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) g_cu_count = p.multiProcessorCount;
    if (g_cu_count <= 0) g_cu_count = 256;
  }
  return g_cu_count;
}

// floats the workspace must hold: one [Cout][9][Cin] partial per block column.  A block covers at most 8 tiles of
// 32 x 32 channel pairs and a launch has at most cu_count() blocks.
long ws_capacity(int Cin, int Cout) {
  const long n = (long)Cout * 9 * Cin;
  const long tiles = ((long)((Cin + 31) / 32) * ((Cout + 31) / 32) + 7) / 8;
  long parts = cu_count() / tiles;
  if (parts < 1) parts = 1;
  return parts * n;
}

template <int MW, int NW, int KG, int R, int KW = 1, int DIL = 1, bool NARROW = false, bool H2 = false>
int launch_wx3_np(WX3Args a, hipStream_t st) {
  constexpr int RING = 2 * R + 2, XG = KG + 2 * (DIL > 8 ? DIL / 8 : 1);
  constexpr size_t NPC = H2 ? 2 : 3;
  constexpr size_t lds_stage = 16 * (NPC * (size_t)(32 * NW) * (RING * XG + 1) + NPC * (size_t)(32 * MW) * (2 * R * KG + 1));
  constexpr size_t lds_red = KW > 1 ? (size_t)(KW / 2) * MW * NW * 144 * 64 * 4 : 0;
  constexpr size_t lds_bytes = lds_stage > lds_red ? lds_stage : lds_red;
  static_assert(lds_bytes <= 160 * 1024, "unit does not fit the 160 KiB LDS");
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e0 = hipFuncSetAttribute((const void*)conv_wgrad_x3_kernel<MW, NW, KG, R, KW, DIL, NARROW, H2>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e0 != hipSuccess) return (int)e0;
    attr_set = true;
  }
  a.nstrips = (a.W + KG * 8 - 1) / (KG * 8);
  const int gy_ = irr_cdiv(a.Cin, 32 * NW), gz_ = irr_cdiv(a.Cout, 32 * MW);
  // one block per CU (LDS) and ONE round of blocks over the chip, each block walking >= 1 column: every block ends with
  // a flush of 9 * 32 MW * 32 NW partial sums, a fixed cost per block that a second round doubles (measured with the
  // earlier atomic flush: one round is 2-7 % faster at 96x112, 10-25 % at 48x56, 15-45 % at 24x28)
  const long want = cu_count() / ((long)gy_ * gz_) > 0 ? cu_count() / ((long)gy_ * gz_) : 1;
  // Cost model (units of R rows): a block walks cols_per_block columns of ceil(rows / R) units + a ring-fill prologue worth ~2.
  constexpr double PROLOGUE = 2.0;
  // (a) one column per (sample, strip, residue), split vertically into 1, 2, 4 or 8 row chunks
  const int hk = (a.H + DIL - 1) / DIL;                     // rows of the longest residue walk
  int rows = hk;
  a.nchunks_y = 1;
  double best = 1e30;
  for (int c = 1; c <= 8; c *= 2) {
    const int rc = c == 1 ? hk : ((hk + c - 1) / c + R - 1) / R * R;
    if (c > 1 && rc < 8 * R) break;
    const int nch = (hk + rc - 1) / rc;
    const long ncols = (long)a.B * a.nstrips * DIL * nch;
    const long cpb = (ncols + want - 1) / want;
    const double t = (double)cpb * ((rc + R - 1) / R + PROLOGUE);
    if (t < best * 0.97) { best = t; rows = rc; a.nchunks_y = nch; }
  }
  a.rows_per_chunk = rows;
  a.ncols = (long)a.B * a.nstrips * DIL * a.nchunks_y;
  a.tall = 0;
  // (dilated layers: the kernel supports the tall walk of one residue class too, but a class has only H / d rows per sample, so the
  // separator row costs 8-17 % there: measured 3-9 % slower at 96x112, faster only at 48x56 d8 -- off unless IRR_WX3_TALL_DIL=1)
  if (WX3_TALL && (DIL == 1 || IRR_ENV_FLAG("IRR_WX3_TALL_DIL")) && !IRR_ENV_FLAG("IRR_WX3_NO_TALL")) {
    // (b) columns = (strip, chunk of the tall image of B * (H + 1) rows), any number of chunks: one extra row per sample, but a
    // block's share is ONE column (one prologue) and the shares are equal to a unit
    const long tk = (long)a.B * (hk + 1);                    // (hk: rows of the longest residue class; shorter ones leave a chunk short)
    const long cmax = 4 * want > 64 ? 4 * want : 64;
    for (long c = 1; c <= cmax; ++c) {
      const long rc = ((tk + c - 1) / c + R - 1) / R * R;
      if (c > 1 && rc < R) break;
      const long nch = (tk + rc - 1) / rc;
      const long ncols = (long)a.nstrips * DIL * nch;
      const long cpb = (ncols + want - 1) / want;
      const double t = (double)cpb * ((double)(rc / R) + PROLOGUE);
      if (t < best * 0.99) { best = t; a.tall = 1; a.rows_per_chunk = (int)rc; a.nchunks_y = (int)nch; a.ncols = ncols; }
    }
  }
  a.cols_per_block = (int)((a.ncols + want - 1) / want);
  if (a.cols_per_block < 1) a.cols_per_block = 1;
  a.ngx = irr_cdiv(a.ncols, a.cols_per_block); a.ngy = gy_; a.ngz = gz_;
  a.dbg = nullptr;
#ifdef WX3_TRACE
  if (!g_wx3_dbg) { hipMalloc(&g_wx3_dbg, 16 * 400 * 8); }
  hipMemsetAsync(g_wx3_dbg, 0, 16 * 400 * 8, st);
  a.dbg = g_wx3_dbg;
#endif
  a.n = (long)a.Cout * 9 * a.Cin;
  if ((long)a.ngx * a.n > ws_capacity(a.Cin, a.Cout)) return IRR_EINVAL;      // (cannot happen: gx <= want)
  hipLaunchKernelGGL((conv_wgrad_x3_kernel<MW, NW, KG, R, KW, DIL, NARROW, H2>), dim3((unsigned)a.ngx * gy_ * gz_), dim3(MW * NW * KW * 64),
                     lds_bytes, st, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  g_last_parts = a.ngx;
  return 0;
}

// (a.x_amax set: the fp16x2 instantiation)
template <int MW, int NW, int KG, int R, int KW = 1, int DIL = 1, bool NARROW = false>
int launch_wx3(const WX3Args& a, hipStream_t st) {
  return a.x_amax ? launch_wx3_np<MW, NW, KG, R, KW, DIL, NARROW, true>(a, st) : launch_wx3_np<MW, NW, KG, R, KW, DIL, NARROW, false>(a, st);
}

// (KG, R) by image width: strips of KG groups must tile the row without waste
static int pick_kg(int W) {
  static const int force = getenv("IRR_WX3_KG") ? atoi(getenv("IRR_WX3_KG")) : 0;     // experiment switch
  if (force == 1 || force == 2 || force == 4) return force;
  const int groups = (W + 7) / 8;                          // W % 8 == 4: the last group of a row is half empty
  if (groups % 4 == 0) return 4;
  if (groups % 2 == 0) return 2;
  return 1;
}

}  // namespace

// gw[co][ci][tap] += sum over the P partials of ws[p][co][tap][ci]   (swapped: the launch ran with the operand roles
// exchanged, see below, and produced ws[p][ci][8-tap][co]).  Body shared with the batched fold: wgrad_reduce.h.
__global__ __launch_bounds__(256) void wgrad_reduce_x3_kernel(const IrrReduceJob J) {
  __shared__ float red[3][256];
  irr_reduce_block(J, blockIdx.x, red);
}

// fold now, or -- between irr_wgrad_defer_begin / _end -- leave a job for irr_wgrad_reduce_batch
static int reduce_or_defer(const float* ws, float* gw, long n, int P, int Cin, int Cout, int swapped, bool may_defer, hipStream_t st) {
  if (may_defer && irr_reduce_defer(ws, gw, n, P, Cin, Cout, 9, swapped)) return 0;
  IrrReduceJob J{};
  J.ws = ws; J.gw = gw; J.n = n; J.P = P; J.Cin = Cin; J.Cout = Cout; J.KK = 9; J.swapped = swapped;
  hipLaunchKernelGGL(wgrad_reduce_x3_kernel, dim3(irr_cdiv(n, 256)), dim3(256), 0, st, J);
  IRR_LAUNCH_CHECK();
  return 0;
}

// gbias[c] += alpha * sum over (b, pixels) of gy (used when the gy stagers of the main kernel cannot provide it)
__global__ __launch_bounds__(256) void wx3_bias_kernel(const float* __restrict__ gy, float* __restrict__ gbias, long HW, long gy_bs,
                                                      float alpha, int chunk) {
  const int c = blockIdx.y, b = blockIdx.z;
  const long p0 = (long)blockIdx.x * chunk, p1 = min(HW, p0 + chunk);
  const float* g = gy + (long)b * gy_bs + (long)c * HW;
  float s = 0.f;
  for (long p = p0 + threadIdx.x; p < p1; p += 256) s += g[p];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) unsafeAtomicAdd(gbias + c, alpha * (red[0] + red[1] + red[2] + red[3]));
}

// dilated layers: only the block shapes the context networks need (128 -> 128 d2/d4, 128 -> 96 d8, 96 -> 64 d16)
static bool dil_ok(int Cout, int W, int dil) {
  const int cot = (Cout + 31) / 32, kg = pick_kg(W);
  if (dil == 2 || dil == 4) return cot == 4;
  if (dil == 8) return cot == 3;
  if (dil == 16) return cot == 2 && (kg != 1 || !IRR_ENV_FLAG("IRR_WX3_NO_D16_KG1"));      // (kg = 1: the (1, 2) unit, see launch_dil)
  return false;
}

#ifdef WX3_TRACE
extern "C" int irr_wx3_trace_dump(unsigned long long* host) {
  if (!g_wx3_dbg) return -1;
  hipDeviceSynchronize();
  return (int)hipMemcpy(host, g_wx3_dbg, 16 * 400 * 8, hipMemcpyDeviceToHost);
}
#endif

extern "C" long irr_conv2d_wgrad_x3_ws_elems(int Cin, int Cout) { return (Cin > 0 && Cout > 0) ? ws_capacity(Cin, Cout) : 0; }

extern "C" int irr_conv2d_wgrad_x3_eligible(int B, int Cin, int H, int W, int Cout, int k, int stride, int dil) {
  if (k != 3 || stride != 1 || B <= 0) return 0;
  // (W % 8 == 4 -- the 24x28 level: the last group of a row is half empty, as in the dilation-1 walk; round 4)
  if (dil != 1 && !(W % 4 == 0 && W >= (IRR_ENV_FLAG("IRR_WX3_DIL_W8") ? 32 : 24) && (W % 8 == 0 || !IRR_ENV_FLAG("IRR_WX3_DIL_W8")) && Cin >= 64 &&
                    dil_ok(Cout, W, dil))) return 0;
  if (dil != 1) return 5000 + dil;
  // Any width >= 7 (rows that are not a multiple of four pixels load element-wise) and, with the tall-image walk, any height:
  // the 6x7 and 12x14 pyramid levels ran on the fp32 kernel at 4-40 TFLOP/s (85 launches, 5.5 ms per step).
  // IRR_WX3_NO_SMALL=1 (A/B): the round-2 limits.
  if (IRR_ENV_FLAG("IRR_WX3_NO_SMALL") || !WX3_TALL) {
    if ((W % 4) || W < 24 || H < 8) return 0;
    if ((long)B * H * W < 40000 && irr_conv_x3_set_min_blocks(-1) > 0) return 0;
  }
  if (W < 7 || H < 3 || Cin < 8) return 0;
  const int kg = pick_kg(W);
  const int cot = (Cout + 31) / 32;
  if (cot == 1 && Cin <= 32) return W % 32 == 0 ? 1144 : 1124;             // <1,1,4,4,KW=8> / <1,1,2,4,KW=4>
  if ((cot == 1 && Cin <= 64) || (cot == 2 && Cin <= 32)) return 2124;     // <2,1,2,4,KW=4> (roles swapped when Cout <= 32)
  const int mw = cot == 1 ? 8 : cot == 2 ? 2 : cot == 3 ? 3 : 4;            // 8: operand roles swapped (see irr_conv2d_wgrad_x3)
  return mw * 100 + kg * 10 + (4 / kg);
}

template <int MW, int DIL>
static int launch_dil(const WX3Args& a, int kg, hipStream_t st) {
  if (kg == 4) return launch_wx3<MW, 2, 4, 1, 1, DIL>(a, st);
  if (kg == 2) return launch_wx3<MW, 2, 2, 2, 1, DIL>(a, st);
  if constexpr (DIL <= 8) return launch_wx3<MW, 2, 1, 4, 1, DIL>(a, st);
  // dilation 16 on rows of an odd number of groups (48x56): the (1, 4) unit with two margin groups per side exceeds the LDS; a
  // (1, 2) unit -- two rows of the residue class, one k-step, 110 KiB -- fits (round 4: the layer ran on the fp32 kernel at 47 TFLOP/s)
  return launch_wx3<MW, 2, 1, 2, 1, DIL>(a, st);
}

static int wgrad_x3_dil_impl(const float* x, const float* gy, float* gw, float* ws, float* gbias, float alpha, int B,
                             int Cin, int H, int W, int Cout, int dil, long x_bs, long gy_bs, const float* x_amax, int nx,
                             const float* g_amax, int ng, void* stream, const float* gy_chmax = nullptr) {
  if (!x || !gy || !gw || !ws || B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || (W % 4) || !dil_ok(Cout, W, dil)) return IRR_EINVAL;
  const long n = (long)Cout * Cin * 9;
  hipStream_t st = (hipStream_t)stream;
  WX3Args a;
  a.ws = ws; a.gbias = gbias; a.xbias = nullptr; a.alpha = alpha;
  a.g_chmax = nullptr;
  a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.x_bs = x_bs; a.gy_bs = gy_bs;
  a.x_amax = x_amax; a.nx_amax = nx; a.g_amax = g_amax; a.ng_amax = ng;
  a.g_chmax = gy_chmax;                                     // (dilated launches never exchange the roles)
  const long lim = (1L << 29) - 64;
  const long bsmax = x_bs > gy_bs ? x_bs : gy_bs;
  long per = bsmax > 0 ? (lim - (long)(Cin > Cout ? Cin : Cout) * H * W) / bsmax : B;
  if (per < 1) return IRR_EINVAL;
  if (per > B) per = B;
  const int kg = pick_kg(W);
  for (int b0 = 0; b0 < B; b0 += (int)per) {
    a.B = (B - b0) < per ? (B - b0) : (int)per;
    a.x = x + (long)b0 * x_bs;
    a.gy = gy + (long)b0 * gy_bs;
    int rc;
    switch (dil) {
      case 2: rc = launch_dil<4, 2>(a, kg, st); break;
      case 4: rc = launch_dil<4, 4>(a, kg, st); break;
      case 8: rc = launch_dil<3, 8>(a, kg, st); break;
      default: rc = launch_dil<2, 16>(a, kg, st); break;
    }
    if (rc) return rc;
    // (the scratch is reused by the next batch slice: only a single-slice launch may defer its fold)
    const int rr = reduce_or_defer(ws, gw, n, g_last_parts, Cin, Cout, 0, per >= B, st);
    if (rr) return rr;
  }
  return 0;
}

extern "C" int irr_conv2d_wgrad_x3_dil(const float* x, const float* gy, float* gw, float* ws, float* gbias, float alpha, int B,
                                       int Cin, int H, int W, int Cout, int dil, long x_bs, long gy_bs, void* stream) {
  return wgrad_x3_dil_impl(x, gy, gw, ws, gbias, alpha, B, Cin, H, W, Cout, dil, x_bs, gy_bs, nullptr, 0, nullptr, 0, stream);
}

// which operand runs in the kernel's x role: the launches of wgrad_x3_impl exchange the roles for some channel counts (see there)
static bool wx3_roles_exchanged(int Cin, int Cout) {
  const int cot = (Cout + 31) / 32, cit = (Cin + 31) / 32;
  const bool ksplit = cot == 1 && Cin <= 32;
  const bool k4 = (cot == 1 && Cin > 32 && Cin <= 64) || (cot == 2 && Cin <= 32);
  const bool sw4 = !IRR_ENV_FLAG("IRR_WX3_NO_SW4") && !ksplit && !k4 &&
                   ((cot == 3 && cit >= 4) || (cot == 1 && (cit + 7) / 8 * 8 * 5 >= (cit + 3) / 4 * 4 * 6));
  return (cot == 1 && !ksplit) || sw4;
}

// 1: x, 0: gy is the operand whose fp16x2 pair carries the scaled-up low piece (2^28 : 1 of its tensor's range at full
// precision, x3_split.h "Range") in irr_conv2d_wgrad_h2 for this problem -- the kernel's x role; the other operand keeps the plain
// pair (2^17 : 1, absolute 2^-25 of the tensor's scale below).  Dilated launches never exchange the roles.
extern "C" int irr_conv2d_wgrad_h2_robust_side(int B, int Cin, int H, int W, int Cout, int dil) {
  (void)B; (void)H; (void)W;
  return (dil > 1 || !wx3_roles_exchanged(Cin, Cout)) ? 1 : 0;
}

static int wgrad_x3_impl(const float* x, const float* gy, float* gw, float* ws, float* gbias, float alpha, int B,
                         int Cin, int H, int W, int Cout, long x_bs, long gy_bs, const float* x_amax, int nx,
                         const float* g_amax, int ng, void* stream, const float* x_chmax = nullptr, const float* gy_chmax = nullptr) {
  if (!x || !gy || !gw || !ws || B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return IRR_EINVAL;
  if ((W % 4) && (((uintptr_t)x | (uintptr_t)gy) & 3)) return IRR_EINVAL;
  const long n = (long)Cout * Cin * 9;
  hipStream_t st = (hipStream_t)stream;
  const int kg = pick_kg(W);
  const int cot = (Cout + 31) / 32;
  // Cout <= 32: one co-tile cannot feed eight waves.  The correlation is symmetric in its operands,
  //   dW[co][ci][t] = sum gy[p] x[p + d(t)] = sum x[p'] gy[p' - d(t)] = dW'[ci][co][8 - t],
  // so the launch runs with the roles exchanged (8 "output-channel" waves over Cin, one "input" tile = Cout) and the
  // unpack kernel transposes and flips the taps.  The bias gradient then comes from a separate pass over the small gy.
  const bool ksplit = cot == 1 && Cin <= 32;               // one (co, ci) tile: eight (four) wave groups split the pixels
  const bool k4 = (cot == 1 && Cin > 32 && Cin <= 64) || (cot == 2 && Cin <= 32);   // two tiles: 2 waves x 4 pixel groups
  // Exchanged roles with FOUR "output-channel" waves over Cin x one "input" tile of Cout x two pixel wave groups (<4,1,*,*,2>):
  //   * Cout = 96 (three co-tiles): the direct 3 x 2 block has six waves -- two SIMDs carry two of them, two carry one;
  //   * Cout <= 32 with a Cin tile count that fills the eight-wave column badly (531 channels = 17 tiles: 24 slots).
  const int cit = (Cin + 31) / 32;
  const bool sw4 = !IRR_ENV_FLAG("IRR_WX3_NO_SW4") && !ksplit && !k4 &&
                   ((cot == 3 && cit >= 4) || (cot == 1 && (cit + 7) / 8 * 8 * 5 >= (cit + 3) / 4 * 4 * 6));   // (the eight-wave column pads >= 1.2 x as much)
  const bool swapped = (cot == 1 && !ksplit) || sw4;
  if (swapped != wx3_roles_exchanged(Cin, Cout)) return IRR_EINVAL;          // (one rule, stated twice: keep them together)
  WX3Args a;
  a.g_chmax = nullptr;
  a.ws = ws; a.gbias = swapped ? nullptr : gbias; a.xbias = swapped ? gbias : nullptr; a.alpha = alpha;
  a.H = H; a.W = W;
  a.Cin = swapped ? Cout : Cin;
  a.Cout = swapped ? Cin : Cout;
  a.x_bs = swapped ? gy_bs : x_bs;
  a.gy_bs = swapped ? x_bs : gy_bs;
  a.x_amax = swapped ? g_amax : x_amax; a.nx_amax = swapped ? ng : nx;
  a.g_amax = swapped ? x_amax : g_amax; a.ng_amax = swapped ? nx : ng;
  a.g_chmax = swapped ? x_chmax : gy_chmax;                                 // the channel maxima of the tensor in the kernel's gy role
  const long lim = (1L << 29) - 64;                                        // elements: byte voffsets below the 2 GiB marker
  const long bsmax = x_bs > gy_bs ? x_bs : gy_bs;
  long per = bsmax > 0 ? (lim - (long)(Cin > Cout ? Cin : Cout) * H * W) / bsmax : B;
  if (per < 1) return IRR_EINVAL;
  if (per > B) per = B;
  for (int b0 = 0; b0 < B; b0 += (int)per) {
    a.B = (B - b0) < per ? (B - b0) : (int)per;
    a.x = (swapped ? gy : x) + (long)b0 * a.x_bs;
    a.gy = (swapped ? x : gy) + (long)b0 * a.gy_bs;
    int rc;
    auto dispatch = [&](auto narrow_tag) {
      constexpr bool NW_ = decltype(narrow_tag)::value;
    if (ksplit) {
        rc = W % 32 == 0 ? launch_wx3<1, 1, 4, 4, 8, 1, NW_>(a, st) : launch_wx3<1, 1, 2, 4, 4, 1, NW_>(a, st);
      } else if (k4) {
        rc = launch_wx3<2, 1, 2, 4, 4, 1, NW_>(a, st);
      } else if (sw4) {
        rc = kg == 4 ? launch_wx3<4, 1, 4, 1, 2, 1, NW_>(a, st) : kg == 2 ? launch_wx3<4, 1, 2, 2, 2, 1, NW_>(a, st) : launch_wx3<4, 1, 1, 4, 2, 1, NW_>(a, st);
      } else if (swapped) {
        rc = kg == 4 ? launch_wx3<8, 1, 4, 1, 1, 1, NW_>(a, st) : kg == 2 ? launch_wx3<8, 1, 2, 2, 1, 1, NW_>(a, st) : launch_wx3<8, 1, 1, 4, 1, 1, NW_>(a, st);
      } else if (cot == 2) {
        // 2 x 2 tiles = four waves would leave one wave per SIMD: two wave groups split the k-steps of a unit (eight waves)
        if (IRR_ENV_FLAG("IRR_WX3_NO_KW2"))
          rc = kg == 4 ? launch_wx3<2, 2, 4, 1, 1, 1, NW_>(a, st) : kg == 2 ? launch_wx3<2, 2, 2, 2, 1, 1, NW_>(a, st) : launch_wx3<2, 2, 1, 4, 1, 1, NW_>(a, st);
        else
          rc = kg == 4 ? launch_wx3<2, 2, 4, 1, 2, 1, NW_>(a, st) : kg == 2 ? launch_wx3<2, 2, 2, 2, 2, 1, NW_>(a, st) : launch_wx3<2, 2, 1, 4, 2, 1, NW_>(a, st);
      } else if (cot % 4 == 0 || cot > 4) {
        rc = kg == 4 ? launch_wx3<4, 2, 4, 1, 1, 1, NW_>(a, st) : kg == 2 ? launch_wx3<4, 2, 2, 2, 1, 1, NW_>(a, st) : launch_wx3<4, 2, 1, 4, 1, 1, NW_>(a, st);
      } else {
        rc = kg == 4 ? launch_wx3<3, 2, 4, 1, 1, 1, NW_>(a, st) : kg == 2 ? launch_wx3<3, 2, 2, 2, 1, 1, NW_>(a, st) : launch_wx3<3, 2, 1, 4, 1, 1, NW_>(a, st);
      }
    };
    if (W & 3) dispatch(std::true_type{});
    else dispatch(std::false_type{});
    if (rc) return rc;
    const int rr = reduce_or_defer(ws, gw, n, g_last_parts, Cin, Cout, swapped ? 1 : 0, per >= B, st);
    if (rr) return rr;
  }
  return 0;
}

extern "C" int irr_conv2d_wgrad_x3(const float* x, const float* gy, float* gw, float* ws, float* gbias, float alpha, int B,
                                   int Cin, int H, int W, int Cout, long x_bs, long gy_bs, void* stream) {
  return wgrad_x3_impl(x, gy, gw, ws, gbias, alpha, B, Cin, H, W, Cout, x_bs, gy_bs, nullptr, 0, nullptr, 0, stream);
}

// The two launches above on the fp16x2 form (dil = 1: irr_conv2d_wgrad_x3's problems, else irr_conv2d_wgrad_x3_dil's; same
// eligibility, scratch and fold): x_amax[0 .. nx) / gy_amax[0 .. ng) = device slots whose maxima bound |x| / |gy|.
extern "C" int irr_conv2d_wgrad_h2(const float* x, const float* gy, float* gw, float* ws, float* gbias, float alpha, int B,
                                   int Cin, int H, int W, int Cout, int dil, long x_bs, long gy_bs, const float* x_amax, int nx,
                                   const float* gy_amax, int ng, void* stream) {
  if (!x_amax || !gy_amax || nx <= 0 || ng <= 0 || dil < 1) return IRR_EINVAL;
  if (dil > 1) return wgrad_x3_dil_impl(x, gy, gw, ws, gbias, alpha, B, Cin, H, W, Cout, dil, x_bs, gy_bs, x_amax, nx, gy_amax, ng, stream);
  return wgrad_x3_impl(x, gy, gw, ws, gbias, alpha, B, Cin, H, W, Cout, x_bs, gy_bs, x_amax, nx, gy_amax, ng, stream);
}

// irr_conv2d_wgrad_h2 with one scale per CHANNEL of the operand in the kernel's gy role (round 6; ABI 11): x_chmax (Cin floats) /
// gy_chmax (Cout floats) = max |.| of every channel of x / gy, e.g. from irr_amax_channels_f32; only the one that
// irr_conv2d_wgrad_h2_robust_side names as NOT robust is read (robust side 1 -> gy_chmax, 0 -> x_chmax), the other may be NULL.
extern "C" int irr_conv2d_wgrad_h2_ch(const float* x, const float* gy, float* gw, float* ws, float* gbias, float alpha, int B,
                                      int Cin, int H, int W, int Cout, int dil, long x_bs, long gy_bs, const float* x_amax, int nx,
                                      const float* gy_amax, int ng, const float* x_chmax, const float* gy_chmax, void* stream) {
  if (!x_amax || !gy_amax || nx <= 0 || ng <= 0 || dil < 1) return IRR_EINVAL;
  if (dil > 1) {
    if (!gy_chmax) return IRR_EINVAL;
    return wgrad_x3_dil_impl(x, gy, gw, ws, gbias, alpha, B, Cin, H, W, Cout, dil, x_bs, gy_bs, x_amax, nx, gy_amax, ng, stream, gy_chmax);
  }
  if (!(wx3_roles_exchanged(Cin, Cout) ? x_chmax : gy_chmax)) return IRR_EINVAL;
  return wgrad_x3_impl(x, gy, gw, ws, gbias, alpha, B, Cin, H, W, Cout, x_bs, gy_bs, x_amax, nx, gy_amax, ng, stream, x_chmax, gy_chmax);
}
