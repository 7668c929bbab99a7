// 81-channel cost volume (search range 4, stride 1) -- forward and both gradients.
//
// Reference semantics: compute_cost_volume (models/pwc_modules.py:42-62) ==
// correlation_forward / correlation_backward_input{1,2} (models/correlation_package/
// correlation_cuda_kernel.cu:41-300) at (pad,k,md,s1,s2)=(4,1,4,1,1).
//
// MI355X design (not the reference's one-block-per-pixel / serial-reduction scheme):
//   * a workgroup owns a 16x16 pixel tile; lanes run along x so every global access is a
//     coalesced 64 B row segment and the 81-plane output is written as dense 64 B rows;
//   * the "other" feature map is staged in LDS with a 4-pixel halo, CC channels at a time
//     ((16+8) x (16+8) x CC floats, row pitch 25 to keep the 9 x-shifts on distinct banks);
//   * each lane keeps all 81 displacement accumulators in VGPRs -> no cross-lane reduction at
//     all, the channel sum is a private FMA chain; HBM sees f1, f2 once and the output once;
//   * both gradients are gathers with the same LDS tile (no atomics, deterministic).
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int TS = 16;            // tile side
constexpr int HALO = 4;
constexpr int TP = TS + 2 * HALO; // 24
constexpr int PITCH = TP + 1;     // 25
constexpr int CC = 16;            // channels per LDS stage

// Generic tile staging with the loads of a batch issued together (a one-load-per-iteration loop exposes the full memory
// latency on every element and dominated these kernels).
template <int NCH, int PITCH_, int NTHR>
__device__ __forceinline__ void stage_tile_t(float (*tile)[TP][PITCH_], const float* __restrict__ src, long plane,
                                             int c0, int C, int y0, int x0, int H, int W, int tid) {
  // tile[c][ty][tx] = src[c0+c][y0-4+ty][x0-4+tx], zero outside image / channel range
  constexpr int N = NCH * TP * TP;
  constexpr int UN = 12;
  for (int i0 = tid; i0 < N; i0 += NTHR * UN) {
    float v[UN];
    int cc[UN], ty[UN], tx[UN];
#pragma unroll
    for (int k = 0; k < UN; ++k) {
      const int i = i0 + k * NTHR;
      const int c = i / (TP * TP);
      const int r = i - c * (TP * TP);
      ty[k] = r / TP;
      tx[k] = r - ty[k] * TP;
      cc[k] = c;
      const int y = y0 - HALO + ty[k], x = x0 - HALO + tx[k];
      v[k] = 0.f;
      if (i < N && c0 + c < C && y >= 0 && y < H && x >= 0 && x < W) v[k] = src[(long)(c0 + c) * plane + (long)y * W + x];
    }
#pragma unroll
    for (int k = 0; k < UN; ++k)
      if (i0 + k * NTHR < N) tile[cc[k]][ty[k]][tx[k]] = v[k];
  }
}

__device__ __forceinline__ void stage_tile(float (*tile)[TP][PITCH], const float* __restrict__ src, long plane,
                                           int c0, int C, int y0, int x0, int H, int W, int tid) {
  stage_tile_t<CC, PITCH, TS * TS>(tile, src, plane, c0, C, y0, x0, H, W, tid);
}

__global__ __launch_bounds__(256) void corr81_fwd_kernel(const float* __restrict__ f1, const float* __restrict__ f2,
                                                        float* __restrict__ out, int C, int H, int W, long f1_bs,
                                                        long f2_bs, long out_bs, int fuse_lrelu) {
  __shared__ __attribute__((aligned(16))) float tile[CC][TP][PITCH];
  const int tid = threadIdx.x;
  const int tx = tid & (TS - 1), ty = tid >> 4;
  const int x0 = blockIdx.x * TS, y0 = blockIdx.y * TS, b = blockIdx.z;
  const int x = x0 + tx, y = y0 + ty;
  const bool inside = (x < W) && (y < H);
  const long plane = (long)H * W;
  const float* f1b = f1 + (long)b * f1_bs;
  const float* f2b = f2 + (long)b * f2_bs;

  float acc[81];
#pragma unroll
  for (int d = 0; d < 81; ++d) acc[d] = 0.f;

  for (int c0 = 0; c0 < C; c0 += CC) {
    __syncthreads();
    stage_tile(tile, f2b, plane, c0, C, y0, x0, H, W, tid);
    __syncthreads();
    const int cn = min(CC, C - c0);
    for (int c = 0; c < cn; ++c) {
      const float a = inside ? f1b[(long)(c0 + c) * plane + (long)y * W + x] : 0.f;
#pragma unroll
      for (int dy = 0; dy < 9; ++dy)
#pragma unroll
        for (int dx = 0; dx < 9; ++dx) acc[dy * 9 + dx] = fmaf(a, tile[c][ty + dy][tx + dx], acc[dy * 9 + dx]);
    }
  }
  if (!inside) return;
  float* o = out + (long)b * out_bs + (long)y * W + x;
  const float cf = (float)C;
#pragma unroll
  for (int d = 0; d < 81; ++d) {
    float v = acc[d] / cf;     // mean over channels (torch.mean = sum / C)
    if (fuse_lrelu) v = irr_lrelu(v);
    o[(long)d * plane] = v;
  }
}

// ---- tiny planes whose width is not a multiple of 4 (the 6x7 and 12x14 pyramid levels) ----------------------------------------
// The 16 x 16-tile kernels launch ONE block per sample there (64 blocks, 42 or 168 of 256 lanes busy) and walk the channels
// serially: 185 us for 86 MFLOP at 6x7x64x196.  These variants spread the same sums over the chip.
//   forward : a block = 64 pixels x 4 channel slices of ONE displacement (81 x chunks x B blocks), slices meet in LDS;
//   gradient: a block = 64 pixels x 4 waves, each wave a few channels (grid.y splits the channels until ~512 blocks exist); the 81
//             weights of a pixel (output gradient x LeakyReLU') live in registers, summed in the same order as corr81_bwd_kernel.
__global__ __launch_bounds__(256) void corr81_small_fwd_kernel(const float* __restrict__ f1, const float* __restrict__ f2,
                                                              float* __restrict__ out, int C, int H, int W, long f1_bs,
                                                              long f2_bs, long out_bs, int fuse_lrelu) {
  __shared__ float red[3][64];
  const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int d = blockIdx.x, b = blockIdx.z;
  const int dy = d / 9 - HALO, dx = d % 9 - HALO;
  const int plane = H * W;
  const int p = blockIdx.y * 64 + lane;
  const bool pok = p < plane;
  const int y = p / W, x = p - y * W;
  const int y2 = y + dy, x2 = x + dx;
  const bool ok = pok && y2 >= 0 && y2 < H && x2 >= 0 && x2 < W;
  const float* a = f1 + (long)b * f1_bs + (ok ? p : 0);
  const float* o = f2 + (long)b * f2_bs + (ok ? y2 * W + x2 : 0);
  float s = 0.f;
#pragma unroll 4
  for (int c = sl; c < C; c += 4) s = fmaf(a[(long)c * plane], o[(long)c * plane], s);
  if (!ok) s = 0.f;
  if (sl) red[sl - 1][lane] = s;
  __syncthreads();
  if (sl || !pok) return;
  s = (s + red[0][lane]) + (red[1][lane] + red[2][lane]);
  float v = s / (float)C;
  if (fuse_lrelu) v = irr_lrelu(v);
  out[(long)b * out_bs + (long)d * plane + p] = v;
}

template <bool SECOND>
__global__ __launch_bounds__(256) void corr81_small_bwd_kernel(const float* __restrict__ other, const float* __restrict__ gout,
                                                              const float* __restrict__ fwd_out, float* __restrict__ gin,
                                                              int C, int H, int W, long other_bs, long gout_bs, long out_bs,
                                                              long gin_bs, int cpb) {
  const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int b = blockIdx.z;
  const int plane = H * W;
  const int p = blockIdx.x * 64 + lane;
  const bool pok = p < plane;
  const int y = p / W, x = p - y * W;
  const float* gb = gout + (long)b * gout_bs;
  const float* fb = fwd_out ? fwd_out + (long)b * out_bs : nullptr;
  float wgt[81];
#pragma unroll
  for (int d = 0; d < 81; ++d) {
    const int dy = d / 9 - HALO, dx = d % 9 - HALO;
    const int yo = SECOND ? y - dy : y + dy, xo = SECOND ? x - dx : x + dx;      // position read in the other map
    const bool ok = pok && yo >= 0 && yo < H && xo >= 0 && xo < W;
    const int go = ok ? (SECOND ? yo * W + xo : p) : 0;                            // position of the output gradient
    float g = gb[(long)d * plane + go];
    if (fb) g *= irr_lrelu_grad(fb[(long)d * plane + go]);
    wgt[d] = ok ? g : 0.f;
  }
  // The other map is read through a buffer resource over this sample's C planes at p +- d WITHOUT a per-displacement offset
  // register: positions outside the image carry weight 0 and land in a neighbouring row / plane or, beyond the sample, in the
  // hardware bounds check (returns 0) -- 81 weights are all the kernel keeps per lane (333 -> ~110 VGPRs: four waves per SIMD).
  const uint32_t obytes = (uint32_t)min((long)0x7ffffffcL, (long)C * plane * 4);
  const __amdgpu_buffer_rsrc_t orr = __builtin_amdgcn_make_buffer_rsrc((void*)(other + (long)b * other_bs), (short)0, (int)obytes, 0x00020000);
  const float inv_c = 1.f / (float)C;
  const int c_end = min(C, ((int)blockIdx.y + 1) * cpb);
  for (int c = blockIdx.y * cpb + sl; c < c_end; c += 4) {
    const uint32_t vo = (uint32_t)(c * plane + p) * 4u;
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < 81; ++d) {
      const int dy = d / 9 - HALO, dx = d % 9 - HALO;
      const int sh = (SECOND ? -(dy * W + dx) : dy * W + dx) * 4;                  // wave-uniform
      s = fmaf(wgt[d], __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(orr, (int)(vo + (uint32_t)sh), 0, 0)), s);
    }
    if (pok) gin[(long)b * gin_bs + (long)c * plane + p] = s * inv_c;
  }
}

// the 16 x 16-tile launch would leave most of the chip empty
static inline bool corr_small(int B, int H, int W) {
  static const int off = IRR_ENV_FLAG("IRR_CORR_NO_SMALL");
  return !off && (long)irr_cdiv(W, TS) * irr_cdiv(H, TS) * B < 512 && (long)H * W <= 65536;
}

// ---- forward, quad variant (W % 4 == 0) ----------------------------------------------------------------------------
// The kernel above issues one LDS read per FMA (81 per channel and pixel), stages its tile one dword at a time and
// writes 64-B row segments: 317 us at 96x112x64 (1.3 TB/s of algorithmic traffic).  Here
//   * the block owns 8 rows x 32 columns, so every output store instruction writes full 128-B lines;
//   * a lane owns FOUR consecutive pixels of a row and THREE of the nine vertical displacements (108 accumulators): per
//     channel it reads its f1 quad and, per vertical displacement, the 12-float window of the staged f2 row as three aligned
//     ds_read_b128 -- 10 LDS instructions per 108 FMAs;
//   * both operands are staged per 8-channel stage as aligned 16-B units with all loads of a thread issued together
//     (f2: (8+8) x (32+8) halo tile, f1: the block's own pixels), so the FMA loop contains no global load.
// 8 rows x 8 quads x 3 displacement groups = 192 threads.
// Tile shape QX x QY (256 pixels; a template parameter since round 3): 32 x 8 where the width is a multiple of 32, 16 x 16 where
// it is only a multiple of 16 -- at 96x112 (the heaviest level) the fourth 32-wide tile of a row was half empty: 12.5 % of the
// lanes of every block idle.  Halo tile (QX + 8) x (QY + 8).
constexpr int QC = 8;                              // channels per LDS stage
// Forward tile pitch: a 16-lane pass of ds_read_b128 covers two tile rows of a 32-wide tile (8 quads = 128 B each); with the
// natural 160-B pitch the second row's window wraps onto the first row's banks (2-way conflict on every window read).  384 B =
// 128 B (mod 256 B) puts consecutive rows on disjoint bank halves.  16-wide tiles: four rows per pass, 192 B = 64 B x 3 (mod
// 256 B) puts them on the four bank quarters.
template <int QX> struct QTile { static constexpr int FP = QX == 32 ? 96 : 48, BP = QX == 32 ? QX + 2 * HALO : 48; };

#ifndef CORR_NT_STORE
#define CORR_NT_STORE 1   // 1: an 81-plane output too large for the caches (56 % of the forward's bytes) is stored non-temporally
                          // (96x112x64: 124 -> 110 us; at 48x56x64, 56 MB, the consumer still finds it in L2 / MALL and nt stores
                          // cost 12 %: the launcher decides by size); 0: A/B
#endif
#ifndef CORR_ABL
#define CORR_ABL 0      // ablation builds (timing only): 1 = no output stores, 2 = no FMA loop, 3 = no global loads in the staging
#endif
typedef float f32x4c __attribute__((ext_vector_type(4)));
typedef float f32x2c __attribute__((ext_vector_type(2)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));      // dword-aligned 16-B global access

template <int QX, int QY>
__global__ __launch_bounds__(192) void corr81_fwd4_kernel(const float* __restrict__ f1, const float* __restrict__ f2,
                                                         float* __restrict__ out, int C, int H, int W, long f1_bs,
                                                         long f2_bs, long out_bs, int fuse_lrelu, int nt_store) {
  static_assert(QX * QY == 256 && (QX == 32 || QX == 16), "256-pixel tiles of 32 x 8 or 16 x 16");
  constexpr int QTX = QX + 2 * HALO, QTY = QY + 2 * HALO, FP = QTile<QX>::FP, NQ = QX / 4;
  __shared__ __attribute__((aligned(16))) float tile[QC][QTY][FP];
  __shared__ __attribute__((aligned(16))) float t1[QC][QY][QX];      // the block's own f1 pixels
  const int tid = threadIdx.x;
  const int grp = tid / 64;                       // vertical displacements 3*grp .. 3*grp + 2
  const int t64 = tid - grp * 64;
  const int q = t64 % NQ, ty = t64 / NQ;          // quad column (4 pixels), tile row
  // XCD-major tile order (common.h): the tiles of one XCD are neighbours, their 2.5x halo overlap is served by that XCD's L2
  const unsigned tpos = irr_xcd_order(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), gridDim.x * gridDim.y * gridDim.z);
  const int x0 = (int)(tpos % gridDim.x) * QX, y0 = (int)((tpos / gridDim.x) % gridDim.y) * QY, b = (int)(tpos / (gridDim.x * gridDim.y));
  const int x = x0 + 4 * q, y = y0 + ty;
  const bool inside = (x < W) && (y < H);         // W % 4 == 0: a quad is inside or outside as a whole
  const long plane = (long)H * W;
  const float* f1b = f1 + (long)b * f1_bs;
  const float* f2b = f2 + (long)b * f2_bs;

  // 108 accumulators per lane, held as 4 aligned PAIRS + 1 single per (displacement row r, pixel i): with the window value
  // index i + d even, (d, d + 1) is an aligned register pair of the 12-float window, so two FMAs are one v_pk_fma_f32 with
  // the f1 value broadcast by op_sel -- even i: pairs d = (0,1) (2,3) (4,5) (6,7) + single d = 8; odd i: single d = 0 +
  // pairs d = (1,2) (3,4) (5,6) (7,8).  20 VALU instructions per row instead of 36.
  f32x2c accp[3][4][4];
  float accs[3][4];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      accs[r][i] = 0.f;
#pragma unroll
      for (int p_ = 0; p_ < 4; ++p_) accp[r][i][p_] = f32x2c{0.f, 0.f};
    }

  // Staging is software-pipelined: the global loads of stage s+1 are issued (into registers) BEFORE the FMA loop of stage
  // s and written to LDS after it, so a block's memory latency hides behind its own arithmetic.  (Measured: three waves per
  // SIMD / four blocks per CU instead of this register prefetch is 40 % SLOWER -- the kernel is bound by the CU's LDS
  // pipe, which more resident blocks only contend for.)
  constexpr int RU = QTX / 4;                               // 16-B units per halo row
  constexpr int N2 = QC * QTY * RU, K2 = (N2 + 191) / 192;
  constexpr int N1 = QC * QY * (QX / 4), K1 = (N1 + 191) / 192;
  f32x4c v2[K2], v1[K1];
  // per-thread staging roles do not depend on the stage: precompute offsets (elements; -1 = outside the image / no unit)
  int o2[K2], o1[K1];                                      // element offsets inside the sample (< 2^31)
#pragma unroll
  for (int k = 0; k < K2; ++k) {
    const int i = tid + k * 192;
    const int c = i / (QTY * RU), r = i - c * (QTY * RU);
    const int sy = r / RU, sx = (r - sy * RU) * 4;
    const int yy = y0 - HALO + sy, xx = x0 - HALO + sx;
    o2[k] = (i < N2 && yy >= 0 && yy < H && xx >= 0 && xx < W) ? (int)(c * plane + (long)yy * W + xx) : -1;
  }
#pragma unroll
  for (int k = 0; k < K1; ++k) {
    const int i = tid + k * 192;
    const int c = i / (QY * (QX / 4)), r = i - c * (QY * (QX / 4));
    const int yy = y0 + r / (QX / 4), xx = x0 + (r % (QX / 4)) * 4;
    o1[k] = (i < N1 && yy < H && xx < W) ? (int)(c * plane + (long)yy * W + xx) : -1;
  }
  auto issue = [&](int c0) {
    const float* p2 = f2b + (long)c0 * plane;
    const float* p1 = f1b + (long)c0 * plane;
#pragma unroll
    for (int k = 0; k < K2; ++k) {
      v2[k] = f32x4c{0.f, 0.f, 0.f, 0.f};
      if (CORR_ABL != 3 && o2[k] >= 0 && c0 + (tid + k * 192) / (QTY * RU) < C) v2[k] = *(const f32x4c*)(p2 + o2[k]);
    }
#pragma unroll
    for (int k = 0; k < K1; ++k) {
      v1[k] = f32x4c{0.f, 0.f, 0.f, 0.f};
      if (CORR_ABL != 3 && o1[k] >= 0 && c0 + (tid + k * 192) / (QY * (QX / 4)) < C) v1[k] = *(const f32x4c*)(p1 + o1[k]);
    }
  };
  auto publish = [&]() {
#pragma unroll
    for (int k = 0; k < K2; ++k) {
      const int i = tid + k * 192;
      if (i < N2) *(f32x4c*)(&tile[0][0][0] + (i / RU) * FP + (i % RU) * 4) = v2[k];      // unit i = (row i / RU of [c][sy], 16-B column i % RU)
    }
#pragma unroll
    for (int k = 0; k < K1; ++k) {
      const int i = tid + k * 192;
      if (i < N1) *(f32x4c*)(&t1[0][0][0] + 4 * i) = v1[k];
    }
  };
  issue(0);
  for (int c0 = 0; c0 < C; c0 += QC) {
    __syncthreads();                                        // the previous stage's FMAs have read the tiles
    publish();
    __syncthreads();
    if (c0 + QC < C) issue(c0 + QC);                        // in flight during the FMA loop below
    // FMA loop over (channel, displacement row) steps with the NEXT step's LDS window read ahead of the current step's 36
    // FMAs (the reads used to be issued right before their use: ~24 exposed LDS round trips per stage, the loop was bound by
    // LDS latency, not by LDS or VALU throughput)
    const int cn = min(QC, C - c0);
    const float* const trow = &tile[0][ty + 3 * grp][4 * q];
    f32x4c a = *(const f32x4c*)(&t1[0][ty][4 * q]);
    f32x4c w0 = *(const f32x4c*)(trow), w1 = *(const f32x4c*)(trow + 4), w2 = *(const f32x4c*)(trow + 8);
#pragma unroll
    for (int c = 0; c < QC; ++c) {
      if (c >= cn || (CORR_ABL == 2 && a[0] != 12345.f)) break;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int cnx = r == 2 ? c + 1 : c, rnx = r == 2 ? 0 : r + 1;
        const int cl = cnx < QC ? cnx : QC - 1;            // (the step after the last one re-reads a valid window, unused)
        const float* row = trow + (cl * QTY + rnx) * FP;
        const f32x4c n0 = *(const f32x4c*)(row), n1 = *(const f32x4c*)(row + 4), n2 = *(const f32x4c*)(row + 8);
        const f32x4c an = r == 2 ? *(const f32x4c*)(&t1[cl][ty][4 * q]) : a;
        const f32x2c wp[6] = {{w0[0], w0[1]}, {w0[2], w0[3]}, {w1[0], w1[1]}, {w1[2], w1[3]}, {w2[0], w2[1]}, {w2[2], w2[3]}};
        const float wv[12] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3], w2[0], w2[1], w2[2], w2[3]};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const f32x2c aa = {a[i], a[i]};
#pragma unroll
          for (int p_ = 0; p_ < 4; ++p_) accp[r][i][p_] = __builtin_elementwise_fma(aa, wp[(i + (i & 1)) / 2 + p_], accp[r][i][p_]);
          accs[r][i] = fmaf(a[i], wv[(i & 1) ? i : i + 8], accs[r][i]);
        }
        w0 = n0; w1 = n1; w2 = n2; a = an;
      }
    }
  }
  if (!inside) return;
  float* o = out + (long)b * out_bs + (long)y * W + x;
  // mean over channels (torch.mean = sum / C): for a power-of-two C the multiplication by 1/C is the same rounding as the
  // division (108 IEEE divisions per lane were a quarter of the kernel's VALU instructions); other C divide
  const float cf = (float)C;
  const bool pow2 = (C & (C - 1)) == 0;
  const float rc = 1.f / cf;
  auto store_all = [&](auto mean) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int d = 0; d < 9; ++d) {
        f32x4c v;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int e = d - (i & 1);                        // position among the paired displacements of pixel i
          float t = mean((i & 1) ? (d == 0 ? accs[r][i] : accp[r][i][e / 2][e & 1]) : (d == 8 ? accs[r][i] : accp[r][i][d / 2][d & 1]));
          if (fuse_lrelu) t = irr_lrelu(t);
          v[i] = t;
        }
        if (CORR_ABL != 1 || v[0] == 12345.f) {
          if (CORR_NT_STORE && nt_store) __builtin_nontemporal_store(v, (f32x4c*)(o + (long)((3 * grp + r) * 9 + d) * plane));
          else *(f32x4c*)(o + (long)((3 * grp + r) * 9 + d) * plane) = v;
        }
      }
  };
  if (pow2) store_all([&](float v) { return v * rc; });
  else store_all([&](float v) { return v / cf; });
}

// SECOND == false: g1[c,p] = (1/C) sum_d g[d][p]     * f2[c][p+d]   (tile = f2, shift +d)
// SECOND == true : g2[c,p] = (1/C) sum_d g[d][p-d]   * f1[c][p-d]   (tile = f1, shift -d)
template <bool SECOND>
__global__ __launch_bounds__(256) void corr81_bwd_kernel(const float* __restrict__ other, const float* __restrict__ gout,
                                                        const float* __restrict__ fwd_out, float* __restrict__ gin,
                                                        int C, int H, int W, long other_bs, long gout_bs, long out_bs,
                                                        long gin_bs) {
  __shared__ __attribute__((aligned(16))) float tile[CC][TP][PITCH];
  const int tid = threadIdx.x;
  const int tx = tid & (TS - 1), ty = tid >> 4;
  const int x0 = blockIdx.x * TS, y0 = blockIdx.y * TS, b = blockIdx.z;
  const int x = x0 + tx, y = y0 + ty;
  const bool inside = (x < W) && (y < H);
  const long plane = (long)H * W;
  const float* ob = other + (long)b * other_bs;
  const float* gb = gout + (long)b * gout_bs;
  const float* fb = fwd_out ? fwd_out + (long)b * out_bs : nullptr;
  const float inv_c = 1.f / (float)C;

  float wgt[81];
#pragma unroll
  for (int dy = 0; dy < 9; ++dy)
#pragma unroll
    for (int dx = 0; dx < 9; ++dx) {
      const int d = dy * 9 + dx;
      int yy = y, xx = x;
      if (SECOND) { yy = y - (dy - 4); xx = x - (dx - 4); }
      float g = 0.f;
      if (inside && yy >= 0 && yy < H && xx >= 0 && xx < W) {
        const long off = (long)d * plane + (long)yy * W + xx;
        g = gb[off];
        if (fb) g *= irr_lrelu_grad(fb[off]);
      }
      wgt[d] = g;
    }

  for (int c0 = 0; c0 < C; c0 += CC) {
    __syncthreads();
    stage_tile(tile, ob, plane, c0, C, y0, x0, H, W, tid);
    __syncthreads();
    const int cn = min(CC, C - c0);
    for (int c = 0; c < cn; ++c) {
      float s = 0.f;
#pragma unroll
      for (int dy = 0; dy < 9; ++dy)
#pragma unroll
        for (int dx = 0; dx < 9; ++dx) {
          const float t = SECOND ? tile[c][ty + 8 - dy][tx + 8 - dx] : tile[c][ty + dy][tx + dx];
          s = fmaf(wgt[dy * 9 + dx], t, s);
        }
      if (inside) gin[(long)b * gin_bs + (long)(c0 + c) * plane + (long)y * W + x] = s * inv_c;
    }
  }
}

// ---- gradients, quad variant (W % 4 == 0) --------------------------------------------------------------------------
// Same decomposition as corr81_fwd4_kernel: 8 x 32 pixel tile, a lane owns four consecutive pixels and three of the nine
// vertical displacements.  Its 108 weights (the output gradient at its pixels, times LeakyReLU' of the forward output when
// the activation was fused) live in registers for the whole kernel; per channel it forms the partial sum over its three
// displacement rows from three aligned 12-float windows of the staged "other" map, and the three partial sums of a pixel
// meet in LDS once per 8-channel stage (one 16-B write per channel and lane, 16-B global stores of full 128-B lines).
//   SECOND == false: g1[c,p] = (1/C) sum_d g[d][p]   * f2[c][p+d]     (tile = f2)
//   SECOND == true : g2[c,p] = (1/C) sum_d g[d][p-d] * f1[c][p-d]     (tile = f1, window mirrored)
template <bool SECOND, int QX, int QY>
__device__ __forceinline__ void corr81_bwd4_body(const float* __restrict__ other, const float* __restrict__ gout,
                                                 const float* __restrict__ fwd_out, float* __restrict__ gin, int C, int H, int W,
                                                 long other_bs, long gout_bs, long out_bs, long gin_bs, unsigned tpos, unsigned ntx,
                                                 unsigned nty, float (*tile)[QY + 2 * HALO][QTile<QX>::BP], float (*red)[QC][QY][QX]) {
  static_assert(QX * QY == 256 && (QX == 32 || QX == 16), "256-pixel tiles of 32 x 8 or 16 x 16");
  constexpr int QTX = QX + 2 * HALO, QTY = QY + 2 * HALO, QP = QTile<QX>::BP, NQ = QX / 4;
  const int tid = threadIdx.x;
  const int grp = tid / 64;
  const int t64 = tid - grp * 64;
  const int q = t64 % NQ, ty = t64 / NQ;
  const int x0 = (int)(tpos % ntx) * QX, y0 = (int)((tpos / ntx) % nty) * QY, b = (int)(tpos / (ntx * nty));
  const int x = x0 + 4 * q, y = y0 + ty;
  const bool inside = (x < W) && (y < H);
  const long plane = (long)H * W;
  const float* ob = other + (long)b * other_bs;
  const float* gb = gout + (long)b * gout_bs;
  const float* fb = fwd_out ? fwd_out + (long)b * out_bs : nullptr;
  const float inv_c = 1.f / (float)C;

  float wgt[3][4][9];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const int dy = 3 * grp + r;
#pragma unroll
    for (int dx = 0; dx < 9; ++dx) {
      const int d = dy * 9 + dx;
      if (!SECOND) {
        f32x4c g = {0.f, 0.f, 0.f, 0.f};
        if (inside) {
          const long off = (long)d * plane + (long)y * W + x;
          g = *(const f32x4c*)(gb + off);
          if (fb) {
            const f32x4c f = *(const f32x4c*)(fb + off);
#pragma unroll
            for (int i = 0; i < 4; ++i) g[i] *= irr_lrelu_grad(f[i]);
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) wgt[r][i][dx] = g[i];
      } else {
        // the quad's four sources are consecutive pixels of row y - (dy - 4) starting at x - (dx - 4): one dword-aligned
        // 16-B load when the whole quad lies inside the row (it used to be four scalar loads with four bounds checks --
        // 216 load instructions per lane before the first FMA); element-wise only at the image borders
        const int yy = y - (dy - 4), xx0 = x - (dx - 4);
        f32x4c g = {0.f, 0.f, 0.f, 0.f};
        if (inside && yy >= 0 && yy < H) {
          const long off = (long)d * plane + (long)yy * W + xx0;
          if (xx0 >= 0 && xx0 + 3 < W) {
            g = *(const f32x4u*)(gb + off);
            if (fb) {
              const f32x4c f = *(const f32x4u*)(fb + off);
#pragma unroll
              for (int i = 0; i < 4; ++i) g[i] *= irr_lrelu_grad(f[i]);
            }
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (xx0 + i >= 0 && xx0 + i < W) {
                g[i] = gb[off + i];
                if (fb) g[i] *= irr_lrelu_grad(fb[off + i]);
              }
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) wgt[r][i][dx] = g[i];
      }
    }
  }

  // software-pipelined staging as in corr81_fwd4_kernel: stage s+1 is loaded into registers during the FMAs of stage s
  constexpr int RU = QTX / 4;
  constexpr int N2 = QC * QTY * RU, K2 = (N2 + 191) / 192;
  f32x4c v2[K2];
  long o2[K2];
  int c2[K2];
#pragma unroll
  for (int k = 0; k < K2; ++k) {
    const int i = tid + k * 192;
    const int c = i / (QTY * RU), r = i - c * (QTY * RU);
    const int sy = r / RU, sx = (r - sy * RU) * 4;
    const int yy = y0 - HALO + sy, xx = x0 - HALO + sx;
    c2[k] = c;
    o2[k] = (i < N2 && yy >= 0 && yy < H && xx >= 0 && xx < W) ? (long)c * plane + (long)yy * W + xx : -1;
  }
  auto issue = [&](int c0) {
    const float* p2 = ob + (long)c0 * plane;
#pragma unroll
    for (int k = 0; k < K2; ++k) {
      v2[k] = f32x4c{0.f, 0.f, 0.f, 0.f};
      if (o2[k] >= 0 && c0 + c2[k] < C) v2[k] = *(const f32x4c*)(p2 + o2[k]);
    }
  };
  issue(0);
  for (int c0 = 0; c0 < C; c0 += QC) {
    __syncthreads();                                        // previous stage's reduction has read `red`, its FMAs `tile`
#pragma unroll
    for (int k = 0; k < K2; ++k) {
      const int i = tid + k * 192;
      if (i < N2) *(f32x4c*)(&tile[0][0][0] + (i / RU) * QP + (i % RU) * 4) = v2[k];      // (row i / RU of [c][sy], 16-B column i % RU)
    }
    __syncthreads();
    if (c0 + QC < C) issue(c0 + QC);
#pragma unroll
    for (int c = 0; c < QC; ++c) {
      f32x4c sacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int dy = 3 * grp + r;
        const float* row = &tile[c][SECOND ? ty + 8 - dy : ty + dy][4 * q];
        const f32x4c w0 = *(const f32x4c*)(row), w1 = *(const f32x4c*)(row + 4), w2 = *(const f32x4c*)(row + 8);
        const float wv[12] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3], w2[0], w2[1], w2[2], w2[3]};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int dx = 0; dx < 9; ++dx) sacc[i] = fmaf(wgt[r][i][dx], wv[SECOND ? i + 8 - dx : i + dx], sacc[i]);
      }
      *(f32x4c*)(&red[grp][c][ty][4 * q]) = sacc;
    }
    __syncthreads();
    {
      constexpr int N = QC * QY * (QX / 4), K = (N + 191) / 192;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int i = tid + k * 192;
        if (i >= N) continue;
        const int c = i / (QY * (QX / 4)), r = i - c * (QY * (QX / 4));
        const int ry = r / (QX / 4), rq = r - ry * (QX / 4);
        const int yy = y0 + ry, xx = x0 + 4 * rq;
        if (c0 + c >= C || yy >= H || xx >= W) continue;
        const f32x4c p0 = *(const f32x4c*)(&red[0][c][ry][4 * rq]), p1 = *(const f32x4c*)(&red[1][c][ry][4 * rq]),
                     p2 = *(const f32x4c*)(&red[2][c][ry][4 * rq]);
        f32x4c o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = ((p0[j] + p1[j]) + p2[j]) * inv_c;
        *(f32x4c*)(gin + (long)b * gin_bs + (long)(c0 + c) * plane + (long)yy * W + xx) = o;
      }
    }
  }
}

template <bool SECOND, int QX, int QY>
__global__ __launch_bounds__(192) void corr81_bwd4_kernel(const float* __restrict__ other, const float* __restrict__ gout,
                                                         const float* __restrict__ fwd_out, float* __restrict__ gin,
                                                         int C, int H, int W, long other_bs, long gout_bs, long out_bs,
                                                         long gin_bs) {
  __shared__ __attribute__((aligned(16))) float tile[QC][QY + 2 * HALO][QTile<QX>::BP];
  __shared__ __attribute__((aligned(16))) float red[3][QC][QY][QX];
  // XCD-major tile order (common.h): the tiles of one XCD are neighbours, their 2.5x halo overlap is served by that XCD's L2
  const unsigned tpos = irr_xcd_order(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), gridDim.x * gridDim.y * gridDim.z);
  corr81_bwd4_body<SECOND, QX, QY>(other, gout, fwd_out, gin, C, H, W, other_bs, gout_bs, out_bs, gin_bs, tpos, gridDim.x, gridDim.y,
                                   tile, red);
}

// BOTH gradients from one launch (round 4): a 1-D grid of 2 x tiles blocks in XCD-major order, position 2t computes g1 of tile t and
// position 2t + 1 its g2.  The two blocks run side by side on one XCD and read the same 81 planes of the output gradient around the
// same tile -- the second read comes out of that XCD's L2 instead of HBM (two launches: gout fetched twice, 0.45 of the 0.8 GB they
// move at 96x112x64).
template <int QX, int QY>
__global__ __launch_bounds__(192) void corr81_bwd4_pair_kernel(const float* __restrict__ f1, const float* __restrict__ f2,
                                                              const float* __restrict__ gout, const float* __restrict__ fwd_out,
                                                              float* __restrict__ g1, float* __restrict__ g2, int C, int H, int W,
                                                              long f1_bs, long f2_bs, long gout_bs, long out_bs, long g1_bs,
                                                              long g2_bs, unsigned ntx, unsigned nty) {
  __shared__ __attribute__((aligned(16))) float tile[QC][QY + 2 * HALO][QTile<QX>::BP];
  __shared__ __attribute__((aligned(16))) float red[3][QC][QY][QX];
  const unsigned pos = irr_xcd_order(blockIdx.x, gridDim.x);
  if (pos & 1u)
    corr81_bwd4_body<true, QX, QY>(f1, gout, fwd_out, g2, C, H, W, f1_bs, gout_bs, out_bs, g2_bs, pos >> 1, ntx, nty, tile, red);
  else
    corr81_bwd4_body<false, QX, QY>(f2, gout, fwd_out, g1, C, H, W, f2_bs, gout_bs, out_bs, g1_bs, pos >> 1, ntx, nty, tile, red);
}

// 16 x 16 tiles where they waste fewer lanes than 32 x 8 ones (96x112: 7 full tiles per row instead of 3.5); IRR_CORR_TILE32=1: A/B
static bool corr_tile16(int W) {
  if (IRR_ENV_FLAG("IRR_CORR_TILE32")) return false;
  return irr_cdiv(W, 16) * 16 < irr_cdiv(W, 32) * 32;
}

}  // namespace

extern "C" int irr_corr81_fwd_f32(const float* f1, const float* f2, float* out, int B, int C, int H, int W, long f1_bs,
                                  long f2_bs, long out_bs, int fuse_lrelu, void* stream) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || !f1 || !f2 || !out) return IRR_EINVAL;
  if (B > 65535) return IRR_EINVAL;
  dim3 grid(irr_cdiv(W, TS), irr_cdiv(H, TS), B);
  if ((W & 3) == 0 && ((f1_bs | out_bs) & 3) == 0 && (((uintptr_t)f1 | (uintptr_t)f2 | (uintptr_t)out) & 15) == 0 && ((f2_bs & 3) == 0) && !IRR_ENV_FLAG("IRR_CORR_SCALAR")) {
    const int nt = (long)B * 81 * H * W * 4 > (128L << 20) ? 1 : 0;      // output beyond the L2s + most of the MALL: stream it out
    if (corr_tile16(W)) {
      hipLaunchKernelGGL((corr81_fwd4_kernel<16, 16>), dim3(irr_cdiv(W, 16), irr_cdiv(H, 16), B), dim3(192), 0, (hipStream_t)stream, f1, f2,
                         out, C, H, W, f1_bs, f2_bs, out_bs, fuse_lrelu, nt);
    } else {
      hipLaunchKernelGGL((corr81_fwd4_kernel<32, 8>), dim3(irr_cdiv(W, 32), irr_cdiv(H, 8), B), dim3(192), 0, (hipStream_t)stream, f1, f2,
                         out, C, H, W, f1_bs, f2_bs, out_bs, fuse_lrelu, nt);
    }
    IRR_LAUNCH_CHECK();
    return 0;
  }
  if (corr_small(B, H, W))
    hipLaunchKernelGGL(corr81_small_fwd_kernel, dim3(81, irr_cdiv(H * W, 64), B), dim3(256), 0, (hipStream_t)stream, f1, f2, out, C, H,
                       W, f1_bs, f2_bs, out_bs, fuse_lrelu);
  else
    hipLaunchKernelGGL(corr81_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, f1, f2, out, C, H, W, f1_bs, f2_bs,
                       out_bs, fuse_lrelu);
  IRR_LAUNCH_CHECK();
  return 0;
}

extern "C" int irr_corr81_bwd_f32(const float* f1, const float* f2, const float* gout, const float* out, float* g1,
                                  float* g2, int B, int C, int H, int W, long f1_bs, long f2_bs, long gout_bs,
                                  long out_bs, long g1_bs, long g2_bs, void* stream) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || !f1 || !f2 || !gout) return IRR_EINVAL;
  if (B > 65535) return IRR_EINVAL;
  dim3 grid(irr_cdiv(W, TS), irr_cdiv(H, TS), B);
  const bool al16 = (((uintptr_t)f1 | (uintptr_t)f2 | (uintptr_t)gout | (uintptr_t)out | (uintptr_t)g1 | (uintptr_t)g2) & 15) == 0 &&
                    ((f1_bs | f2_bs | gout_bs | out_bs | g1_bs | g2_bs) & 3) == 0;
  if ((W & 3) == 0 && al16 && !IRR_ENV_FLAG("IRR_CORR_SCALAR")) {
    const bool t16 = corr_tile16(W);
    const dim3 grid4 = t16 ? dim3(irr_cdiv(W, 16), irr_cdiv(H, 16), B) : dim3(irr_cdiv(W, 32), irr_cdiv(H, 8), B);
    // both gradients of a LARGE problem: one launch, the pair of a tile shares gout through L2 (measured, 64 samples x 32 channels:
    // 96x112 307 -> 263 us; 48x56 87 -> 108 us -- with 768 tiles the launch is a single round of blocks either way and the pair
    // kernel's register count is the larger of the two bodies)
    if (g1 && g2 && (long)grid4.x * grid4.y * grid4.z >= 2048 && !IRR_ENV_FLAG("IRR_CORR_NO_PAIR")) {
      const unsigned nblk = 2u * grid4.x * grid4.y * grid4.z;
      if (t16) hipLaunchKernelGGL((corr81_bwd4_pair_kernel<16, 16>), dim3(nblk), dim3(192), 0, (hipStream_t)stream, f1, f2, gout, out, g1, g2,
                                  C, H, W, f1_bs, f2_bs, gout_bs, out_bs, g1_bs, g2_bs, grid4.x, grid4.y);
      else hipLaunchKernelGGL((corr81_bwd4_pair_kernel<32, 8>), dim3(nblk), dim3(192), 0, (hipStream_t)stream, f1, f2, gout, out, g1, g2,
                              C, H, W, f1_bs, f2_bs, gout_bs, out_bs, g1_bs, g2_bs, grid4.x, grid4.y);
      IRR_LAUNCH_CHECK();
      return 0;
    }
    if (g1) {
      if (t16) hipLaunchKernelGGL((corr81_bwd4_kernel<false, 16, 16>), grid4, dim3(192), 0, (hipStream_t)stream, f2, gout, out, g1, C, H, W,
                                  f2_bs, gout_bs, out_bs, g1_bs);
      else hipLaunchKernelGGL((corr81_bwd4_kernel<false, 32, 8>), grid4, dim3(192), 0, (hipStream_t)stream, f2, gout, out, g1, C, H, W,
                              f2_bs, gout_bs, out_bs, g1_bs);
      IRR_LAUNCH_CHECK();
    }
    if (g2) {
      if (t16) hipLaunchKernelGGL((corr81_bwd4_kernel<true, 16, 16>), grid4, dim3(192), 0, (hipStream_t)stream, f1, gout, out, g2, C, H, W,
                                  f1_bs, gout_bs, out_bs, g2_bs);
      else hipLaunchKernelGGL((corr81_bwd4_kernel<true, 32, 8>), grid4, dim3(192), 0, (hipStream_t)stream, f1, gout, out, g2, C, H, W,
                              f1_bs, gout_bs, out_bs, g2_bs);
      IRR_LAUNCH_CHECK();
    }
    return 0;
  }
  if (corr_small(B, H, W)) {
    const int chunks = irr_cdiv(H * W, 64);
    int cs = irr_cdiv(512, chunks * B);                            // channel splits: ~512 blocks
    cs = cs < 1 ? 1 : (cs > irr_cdiv(C, 4) ? irr_cdiv(C, 4) : cs);
    const int cpb = irr_cdiv(irr_cdiv(C, cs), 4) * 4;               // channels per block, a multiple of the four waves
    const dim3 gs(chunks, irr_cdiv(C, cpb), B);
    if (g1) {
      hipLaunchKernelGGL(corr81_small_bwd_kernel<false>, gs, dim3(256), 0, (hipStream_t)stream, f2, gout, out, g1, C, H, W, f2_bs,
                         gout_bs, out_bs, g1_bs, cpb);
      IRR_LAUNCH_CHECK();
    }
    if (g2) {
      hipLaunchKernelGGL(corr81_small_bwd_kernel<true>, gs, dim3(256), 0, (hipStream_t)stream, f1, gout, out, g2, C, H, W, f1_bs,
                         gout_bs, out_bs, g2_bs, cpb);
      IRR_LAUNCH_CHECK();
    }
    return 0;
  }
  if (g1) {
    hipLaunchKernelGGL(corr81_bwd_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, f2, gout, out, g1, C, H, W,
                       f2_bs, gout_bs, out_bs, g1_bs);
    IRR_LAUNCH_CHECK();
  }
  if (g2) {
    hipLaunchKernelGGL(corr81_bwd_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, f1, gout, out, g2, C, H, W,
                       f1_bs, gout_bs, out_bs, g2_bs);
    IRR_LAUNCH_CHECK();
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------
// The legacy operator at ANY parameter point (pad_size, kernel_size, max_displacement, stride1, stride2)
// (models/correlation_package/correlation.py:47-61; forward arithmetic: correlation_cuda_kernel.cu:41-114, output shape:
// correlation_cuda.cc:23-32).  No model of the reference leaves the point (4, 1, 4, 1, 1) that the kernels above serve; this pair
// exists so that the drop-in module accepts what the reference module accepts.  With P1, P2 = the inputs zero-padded by pad_size,
// kr = (k - 1) / 2, dr = md / s2, D = 2 dr + 1, (y1, x1) = (oy s1 + md, ox s1 + md) in padded coordinates:
//     out[n, (tj + dr) D + (ti + dr), oy, ox] = 1 / (k k C) * sum_{j, i in [-kr, kr]} sum_c P1[n, c, y1 + j, x1 + i] * P2[n, c, y1 + tj s2 + j, x1 + ti s2 + i]
// (positions outside the padded arrays -- possible only for kr > md -- read as zero, where the reference would read out of bounds).
// Backward = the exact adjoint of this forward (for k = 1, s1 = 1 that IS correlation_cuda_kernel.cu:116-300; for the other points
// the reference's own backward is an approximation that cannot be run or pinned here: DESIGN.md 8).
// One thread per output / input element, lanes along x: a general-purpose pair, not a tuned one.
namespace {

struct CorrGen { int C, H, W, pad, k, md, s1, s2, OH, OW, kr, dr, D; long f1_bs, f2_bs, out_bs; };

__device__ __forceinline__ float corrg_at(const float* __restrict__ f, const CorrGen& p, int c, int py, int px) {   // padded coords
  const int y = py - p.pad, x = px - p.pad;
  return (y >= 0 && y < p.H && x >= 0 && x < p.W) ? f[(long)c * p.H * p.W + (long)y * p.W + x] : 0.f;
}

__global__ __launch_bounds__(256) void corr_general_fwd_kernel(const float* __restrict__ f1, const float* __restrict__ f2,
                                                              float* __restrict__ out, const CorrGen p) {
  // (rows of the launch = (displacement channel, output row) pairs, folded into grid.x: D * D * OH exceeds the 65535 of grid.y for
  // FlowNetC-sized volumes at full resolution)
  const int nxb = (p.OW + 255) / 256;
  const int ox = (int)(blockIdx.x % nxb) * 256 + threadIdx.x;
  const int row = blockIdx.x / nxb;
  const int oy = row % p.OH, tc = row / p.OH, n = blockIdx.y;
  if (ox >= p.OW) return;
  const int tj = tc / p.D - p.dr, ti = tc % p.D - p.dr;
  const int y1 = oy * p.s1 + p.md, x1 = ox * p.s1 + p.md;
  const float* a = f1 + (long)n * p.f1_bs;
  const float* b = f2 + (long)n * p.f2_bs;
  float s = 0.f;
  for (int c = 0; c < p.C; ++c)
    for (int j = -p.kr; j <= p.kr; ++j)
      for (int i = -p.kr; i <= p.kr; ++i)
        s += corrg_at(a, p, c, y1 + j, x1 + i) * corrg_at(b, p, c, y1 + tj * p.s2 + j, x1 + ti * p.s2 + i);
  out[(long)n * p.out_bs + ((long)tc * p.OH + oy) * p.OW + ox] = s / (float)(p.k * p.k * p.C);
}

// SECOND == false: g1[n, c, y, x] = 1/(k k C) sum over (tj, ti, j, i) with (y + pad - j - md, x + pad - i - md) = (oy s1, ox s1):
//                                   gout[n, tc, oy, ox] * P2[n, c, y + pad + tj s2, x + pad + ti s2]
// SECOND == true : g2[n, c, y, x]: the same with P1 read at (y + pad - tj s2, x + pad - ti s2) and the window centre moved by (tj s2, ti s2)
template <bool SECOND>
__global__ __launch_bounds__(256) void corr_general_bwd_kernel(const float* __restrict__ other, const float* __restrict__ gout,
                                                              float* __restrict__ gin, const CorrGen p, long other_bs, long gin_bs) {
  const int nxb = (p.W + 255) / 256;
  const int x = (int)(blockIdx.x % nxb) * 256 + threadIdx.x;
  const int row = blockIdx.x / nxb;
  const int y = row % p.H, c = row / p.H, n = blockIdx.y;
  if (x >= p.W) return;
  const float* o = other + (long)n * other_bs;
  const float* g = gout + (long)n * p.out_bs;
  float s = 0.f;
  for (int tj = -p.dr; tj <= p.dr; ++tj)
    for (int ti = -p.dr; ti <= p.dr; ++ti) {
      const int tc = (tj + p.dr) * p.D + (ti + p.dr);
      for (int j = -p.kr; j <= p.kr; ++j)
        for (int i = -p.kr; i <= p.kr; ++i) {
          // padded position of THIS element inside the window: first operand at (y1 + j, x1 + i), second at (y1 + tj s2 + j, x1 + ti s2 + i)
          const int cy = y + p.pad - j - (SECOND ? tj * p.s2 : 0) - p.md, cx = x + p.pad - i - (SECOND ? ti * p.s2 : 0) - p.md;
          if (cy < 0 || cx < 0 || cy % p.s1 || cx % p.s1) continue;
          const int oy = cy / p.s1, ox = cx / p.s1;
          if (oy >= p.OH || ox >= p.OW) continue;
          const int py = y + p.pad + (SECOND ? -tj * p.s2 : tj * p.s2), px = x + p.pad + (SECOND ? -ti * p.s2 : ti * p.s2);
          s += g[((long)tc * p.OH + oy) * p.OW + ox] * corrg_at(o, p, c, py, px);
        }
    }
  gin[(long)n * gin_bs + ((long)c * p.H + y) * p.W + x] = s / (float)(p.k * p.k * p.C);
}

static int corrg_setup(CorrGen* p, int B, int C, int H, int W, int pad, int k, int md, int s1, int s2) {
  if (B <= 0 || B > 65535 || C <= 0 || H <= 0 || W <= 0 || pad < 0 || k < 1 || !(k & 1) || md < 0 || s1 < 1 || s2 < 1) return IRR_EINVAL;
  p->C = C; p->H = H; p->W = W; p->pad = pad; p->k = k; p->md = md; p->s1 = s1; p->s2 = s2;
  p->kr = (k - 1) / 2; p->dr = md / s2; p->D = 2 * p->dr + 1;
  const int border = p->kr + md;
  const int ph = H + 2 * pad - 2 * border, pw = W + 2 * pad - 2 * border;
  if (ph <= 0 || pw <= 0) return IRR_EINVAL;
  p->OH = (ph + s1 - 1) / s1; p->OW = (pw + s1 - 1) / s1;                     // ceil, correlation_cuda.cc:31-32
  return 0;
}

}  // namespace

extern "C" int irr_corr_general_out_shape(int H, int W, int pad, int k, int md, int s1, int s2, int* channels, int* OH, int* OW) {
  CorrGen p;
  const int rc = corrg_setup(&p, 1, 1, H, W, pad, k, md, s1, s2);
  if (rc) return rc;
  if (channels) *channels = p.D * p.D;
  if (OH) *OH = p.OH;
  if (OW) *OW = p.OW;
  return 0;
}

extern "C" int irr_corr_general_fwd_f32(const float* f1, const float* f2, float* out, int B, int C, int H, int W, int pad, int k,
                                        int md, int s1, int s2, long f1_bs, long f2_bs, long out_bs, void* stream) {
  CorrGen p;
  if (!f1 || !f2 || !out) return IRR_EINVAL;
  const int rc = corrg_setup(&p, B, C, H, W, pad, k, md, s1, s2);
  if (rc) return rc;
  const long nblk = (long)irr_cdiv(p.OW, 256) * p.D * p.D * p.OH;
  if (nblk > 0x7fffffffL) return IRR_EINVAL;
  p.f1_bs = f1_bs; p.f2_bs = f2_bs; p.out_bs = out_bs;
  hipLaunchKernelGGL(corr_general_fwd_kernel, dim3((unsigned)nblk, B), dim3(256), 0, (hipStream_t)stream, f1, f2, out, p);
  IRR_LAUNCH_CHECK();
  return 0;
}

extern "C" int irr_corr_general_bwd_f32(const float* f1, const float* f2, const float* gout, float* g1, float* g2, int B, int C, int H,
                                        int W, int pad, int k, int md, int s1, int s2, long f1_bs, long f2_bs, long gout_bs, long g1_bs,
                                        long g2_bs, void* stream) {
  CorrGen p;
  if (!f1 || !f2 || !gout) return IRR_EINVAL;
  const int rc = corrg_setup(&p, B, C, H, W, pad, k, md, s1, s2);
  if (rc) return rc;
  const long nblk = (long)irr_cdiv(W, 256) * C * H;
  if (nblk > 0x7fffffffL) return IRR_EINVAL;
  p.f1_bs = f1_bs; p.f2_bs = f2_bs; p.out_bs = gout_bs;
  const dim3 grid((unsigned)nblk, B);
  if (g1) {
    hipLaunchKernelGGL(corr_general_bwd_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, f2, gout, g1, p, f2_bs, g1_bs);
    IRR_LAUNCH_CHECK();
  }
  if (g2) {
    hipLaunchKernelGGL(corr_general_bwd_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, f1, gout, g2, p, f1_bs, g2_bs);
    IRR_LAUNCH_CHECK();
  }
  return 0;
}
