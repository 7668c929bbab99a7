// Winograd F(2x2, 3x3) on the fp16x2 split-operand arithmetic (round 6, VERDICT r5 next #1: a GATED experiment).
//
// The 3x3 / stride-1 / dilation-1 conv() blocks with Cin >= 64 (models/pwc_modules.py:153-243, models/irr_modules.py:63-139) run
// conv_x3_kernel at the power wall: 73 % matrix-pipe busy at 1.42 GHz (profiles/r6_pmc_wgrad.txt).  What is left is the NUMBER of matrix
// instructions.  F(2x2, 3x3) needs 16 instead of 36 products per 2x2 output tile:
//     U = G g G^T                (4x4 per (co, ci); at pack time, fp32, then the plain fp16 pair  uh + ul  of U * 2^ew)
//     V = B^T d B                (4x4 per (ci, tile); adds only, |V| <= 4 max |x|: two more bits of head room in the x scale;
//                                 pair  vh + 2^-11 vl'  -- the scaled-up low piece of x3_split.h, element-wise range 2^29)
//     M[xi] = sum_ci U[xi] V[xi] (16 GEMMs, v_mfma_f32_32x32x16_f16:  acc += ul vh + (uh 2^-11) vl' + uh vh)
//     Y = A^T M A                (2x2 per (co, tile), fp32)
// Host emulation of this arithmetic (tools/wino_emulate.py): 1.1-3.2x the error of an fp32 convolution in every operand range and
// regional case of tests/test_h2_gpu.py (bar: 4x).
//
// Layout of the work.  ONE block of eight waves per CU computes 64 output channels x 64 tiles (16 x 16 pixels) with all 16 xi:
// 64 * 64 * 16 fp32 accumulators = 256 KiB = half the CU's register file; wave w owns xi = 2w, 2w + 1 for both 32-channel co-tiles
// and both 32-tile groups (128 accumulator registers), so no packed U fragment is ever needed by two waves: U streams from L2
// straight into registers (1 KiB per fragment, 64 KiB per 16-channel chunk and block), V goes through LDS:
//   per chunk   raw 18 x 18 x 16 patch (global -> registers one chunk ahead -> LDS, scaled by 2^ex)
//               transform: thread = (tile, channel pair): 16 ds_read_b64, 64 adds, 16 x (pair split), 32 ds_write_b32 into
//                          V[buf][xi][piece][k-group][tile] x 16 B (= the B fragments: conflict-free ds_read_b128)
//               MFMA: per xi 4 A fragments (global), 4 B fragments (LDS), 12 MFMAs
//   V is double-buffered: transform(c + 1) and MFMA(c) sit between the same two barriers, and the two waves of a SIMD run them in
//   OPPOSITE orders (waves 0-3: transform first; 4-7: MFMA first), so one wave's VALU / LDS work runs under the other's MFMAs.
//   Epilogue: the accumulators of one co-tile at a time go through LDS ([xi][co][tile] fp32 = 128 KiB), every thread output-
//   transforms four (co, tile) pairs: bias, LeakyReLU, alpha, 2x2 pixels as two 8-byte stores, max |y| folded into y_amax.
#include "x3_split.h"
#include "amax.h"

#ifndef WINO_TR_FIRST_OLD
#define WINO_TR_FIRST_OLD 1   // 1: the older waves (0-3) transform first and the younger run their MFMAs first; 0 (A/B): the other way round
#endif
#ifndef WINO_PRIO
#define WINO_PRIO 1     // s_setprio of the MFMA phase (0: A/B)
#endif
#ifndef WINO_ABL
#define WINO_ABL 0      // ablation builds (timing only, results wrong): 1 = no U loads in the loop, 2 = no transform (V stale),
#endif                  // 3 = no raw global loads, 4 = no MFMAs, 5 = same order in all waves (no stagger)

#ifdef WINO_TRACE
static unsigned long long* g_wino_dbg = nullptr;     // s_memtime trace (tag builds with -DWINO_TRACE=1, tools/wino_trace.py): block 7, lane 0 of every wave
#endif

namespace {

constexpr uint32_t WOOB = 0x80000000u;
typedef unsigned int u32x2v __attribute__((__vector_size__(2 * sizeof(unsigned int))));
constexpr int RAWP = 20;                       // dwords per raw patch row (18 + 2: conflict-free ds_read_b64 over four channel pairs)
constexpr int RAWCH = 18 * RAWP;               // dwords per raw channel
constexpr int VUNITS = 16 * 2 * 2 * 64;        // 16-B units per V buffer: [xi][piece][k-group][tile]

struct WinoArgs {
  const float* x;
  const u32x4* uq;
  const float* bias;
  float* y;
  int B, Cin, H, W, Cout;
  int nchunk, CoT;                             // 16-channel chunks; 32-channel co-tiles of the pack (even)
  int tiles_x, tiles_y, ngy;
  long x_bs, y_bs;
  int lrelu;
  float alpha;
  const float* x_amax;
  int n_amax;
  float* y_amax;
  unsigned long long* dbg;                     // WINO_TRACE builds only
};

#if defined(WINO_TRACE) && WINO_TRACE >= 2
#define WTR2(slot) WTR(slot)
#define WTR2_WAIT_LDS() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#else
#define WTR2(slot) do {} while (0)
#define WTR2_WAIT_LDS() do {} while (0)
#endif
#ifdef WINO_TRACE
#define WTR(slot) do { if (blockIdx.x == 7 && lane == 0 && ntr < 1024) { dbgp[ntr++] = ((unsigned long long)(slot) << 56) | (__builtin_amdgcn_s_memtime() & 0xffffffffffffffull); } } while (0)
#else
#define WTR(slot) do {} while (0)
#endif

__global__ __launch_bounds__(512) void conv_wino_kernel(const WinoArgs a) {
  extern __shared__ u32x4 lds[];
  u32x4* const vl = lds;                                    // two V buffers (128 KiB); the epilogue's exchange area afterwards
  float* const rawl = (float*)(lds + 2 * VUNITS);           // raw patch [16][18][RAWP] fp32, already scaled

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef WINO_TRACE
  int ntr = 0;
  unsigned long long* dbgp = a.dbg + (size_t)wave * 1024;
#endif
  const unsigned xpos = irr_xcd_order(blockIdx.x, gridDim.x);
  const int by = (int)(xpos % (unsigned)a.ngy);
  int bt = (int)(xpos / (unsigned)a.ngy);
  const int tx = bt % a.tiles_x;
  bt /= a.tiles_x;
  const int ty = bt % a.tiles_y;
  const int b = bt / a.tiles_y;
  const int y0 = ty * 16, x0 = tx * 16;
  const long hw = (long)a.H * a.W;
  const uint32_t hw4 = (uint32_t)(hw * 4);

  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (long)b * a.x_bs), (short)0, (int)0x80000000u, 0x00020000);
  const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc((void*)a.uq, (short)0, (int)0xffffffffu, 0x00020000);

  // ---- operand scales ----
  const int ex = x3_h2_exp(x3_h2_amax(a.x_amax, a.n_amax)) - 2;       // |V| <= 4 max |x|
  const int ew = ((const int*)(a.uq + (long)a.nchunk * 16 * 2 * a.CoT * 64))[0];
  const float sx = ldexpf(1.f, ex), inv_x = ldexpf(1.f, -ex), inv_w = ldexpf(1.f, -ew);

  // ---- staging role ----
  // 16-byte loads: the vector-memory pipe takes ~16 cycles per wave instruction whatever its width, and eight waves x 16 dword loads
  // were 2 000 cycles per chunk of pure issue time (s_memtime trace: 1 750 cycles between the barrier and the next stamp in waves that
  // did nothing else).  A unit = four pixels x0 - 4 + 4 quad .. + 3 of one patch row (quads 0 and 5 carry one patch column each); thread =
  // (row, quad) of channels 4 kb .. 4 kb + 3, kb = tid / 108: 432 threads, four loads each.  Needs W % 4 == 0 and 16-byte aligned planes.
  const bool stager = tid < 432;
  const int kb = tid / 108, spos = tid - kb * 108;
  const int srow = spos / 6, squad = spos - srow * 6;
  uint32_t svoff = WOOB;
  {
    const int iy = y0 - 1 + srow, ix = x0 - 4 + 4 * squad;
    if (stager && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) svoff = (uint32_t)(((long)(kb * 4) * hw + (long)iy * a.W + ix) * 4);
  }
  int soff[4];                                               // LDS dword index of the unit's four values (pad columns 18 / 19 take the unused ones)
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int pc = 4 * squad - 3 + e;
    soff[e] = (kb * 4) * RAWCH + srow * RAWP + ((pc < 0 || pc > 17) ? 18 + (e & 1) : pc);
  }
  const int tail_base = a.Cin - 16;                          // the last chunk re-reads [Cin-16, Cin) (duplicates have zero weights)
  f32x4 raw[4];
  auto issue_raw = [&](int c) {
    const int ch0 = (c == a.nchunk - 1) ? tail_base : c * 16;
    const uint32_t s0 = (uint32_t)ch0 * hw4;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      raw[k] = (WINO_ABL == 3 && c > 0) ? f32x4{1.f, 1.f, 1.f, 1.f} : __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, (int)svoff, (int)(s0 + k * hw4), 0));
  };
  auto publish_raw = [&]() {
    if (stager) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) rawl[k * RAWCH + soff[e]] = raw[k][e] * sx;
    }
  };

  // ---- transform role: (tile, channel pair) ----
  const int ttile = 16 * (wave >> 1) + (lane >> 2);          // 0..63: tile row ttile >> 3, tile column ttile & 7
  const int tq = 4 * (wave & 1) + (lane & 3);                // channel pair: channels 2 tq, 2 tq + 1
  const int traw = (2 * tq) * RAWCH + (2 * (ttile >> 3)) * RAWP + 2 * (ttile & 7);
  const int tvw = (((wave & 1) * 64 + ttile) * 4) + (lane & 3);        // dword index inside one [k-group][tile] plane pair (+ (xi * 2 + piece) * 512)
  auto transform = [&](int buf) {
    if (WINO_ABL == 2) return;
    // rows 1 and 2 of the 4 x 4 patch feed every xi row; row 0 only r = 0, row 3 only r = 3: at most three rows are live (the
    // kernel runs at 256 registers with 128 accumulators: every sched_barrier below keeps the scheduler from widening this)
    float dm[2][2][4], de[2][4];
    auto load_row = [&](int ch, int i, float* o) {
      const f32x2 p0 = *(const f32x2*)(rawl + traw + ch * RAWCH + i * RAWP);
      const f32x2 p1 = *(const f32x2*)(rawl + traw + ch * RAWCH + i * RAWP + 2);
      o[0] = p0[0]; o[1] = p0[1]; o[2] = p1[0]; o[3] = p1[1];
    };
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
      load_row(ch, 1, dm[ch][0]);
      load_row(ch, 2, dm[ch][1]);
      load_row(ch, 0, de[ch]);
    }
    uint32_t* const vw = (uint32_t*)(vl + buf * VUNITS) + tvw;
    WTR2(10);
    WTR2_WAIT_LDS();
    WTR2(11);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float t[2][4];
#pragma unroll
      for (int ch = 0; ch < 2; ++ch)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          t[ch][j] = r == 0 ? de[ch][j] - dm[ch][1][j] : r == 1 ? dm[ch][0][j] + dm[ch][1][j] : r == 2 ? dm[ch][1][j] - dm[ch][0][j] : dm[ch][0][j] - de[ch][j];
      if (r == 0) {                                          // row 0 is dead: fetch row 3 into its registers
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) load_row(ch, 3, de[ch]);
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float v0, v1;
        if (c == 0) { v0 = t[0][0] - t[0][2]; v1 = t[1][0] - t[1][2]; }
        else if (c == 1) { v0 = t[0][1] + t[0][2]; v1 = t[1][1] + t[1][2]; }
        else if (c == 2) { v0 = t[0][2] - t[0][1]; v1 = t[1][2] - t[1][1]; }
        else { v0 = t[0][1] - t[0][3]; v1 = t[1][1] - t[1][3]; }
        const f32x2 vv = {v0, v1};
        const f16x2 hp = __builtin_convertvector(vv, f16x2);
        const f32x2 vu = vv * H2_LO_UP;                       // (the transform phase never runs beside this wave's own MFMAs)
        const float r0 = __builtin_fmaf((float)hp[0], -H2_LO_UP, vu[0]);
        const float r1 = __builtin_fmaf((float)hp[1], -H2_LO_UP, vu[1]);
        const f32x2 rr = {r0, r1};
        const int xi = r * 4 + c;
        if (WINO_ABL == 6) {                                 // ablation 6: the arithmetic without the LDS writes (kept alive by an empty asm)
          uint32_t k0 = __builtin_bit_cast(uint32_t, hp), k1 = __builtin_bit_cast(uint32_t, __builtin_convertvector(rr, f16x2));
          asm volatile("" ::"v"(k0), "v"(k1));
          continue;
        }
        if (WINO_ABL == 7) {                                 // ablation 7: the LDS writes without the arithmetic
          vw[(xi * 2 + 0) * 512] = __builtin_bit_cast(uint32_t, v0);
          vw[(xi * 2 + 1) * 512] = __builtin_bit_cast(uint32_t, v1);
          continue;
        }
        vw[(xi * 2 + 0) * 512] = __builtin_bit_cast(uint32_t, hp);
        vw[(xi * 2 + 1) * 512] = __builtin_bit_cast(uint32_t, __builtin_convertvector(rr, f16x2));
      }
      __builtin_amdgcn_sched_barrier(0);
      WTR2(12 + r);
    }
    WTR2_WAIT_LDS();
    WTR2(16);
  };

  // ---- MFMA role: xi = 2 wave + {0, 1}; co-tiles by * 2 + {0, 1}; tile groups {0, 1} ----
  const int j = lane & 31, g = lane >> 5;
  const int cot0 = by * 2;
  const uint32_t uvoff = (uint32_t)(lane * 16);
  const uint32_t upiece = (uint32_t)a.CoT * 1024u;           // bytes between the two pieces of one xi
  u32x4 ua[2][2][2];                                        // [xi slot][piece][co-tile]
  auto issue_u = [&](int s, int c) {
    if (WINO_ABL == 1 && c > 0) return;
    const uint32_t so = ((uint32_t)(c * 16 + 2 * wave + s) * 2u) * upiece + (uint32_t)cot0 * 1024u;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int q = 0; q < 2; ++q)
        ua[s][p][q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ur, (int)uvoff, (int)(so + p * upiece + q * 1024u), 0));
  };
  f32x16 acc[2][2][2];                                      // [xi slot][co-tile][tile group]
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[s][q][t][r] = 0.f;
  const int bidx = g * 64 + j;                              // 16-B unit inside one [piece] plane pair: + (xi * 2 + piece) * 128 + tg * 32
  auto mfma = [&](int buf, int cnext) {
    const u32x4* const vb = vl + buf * VUNITS + bidx;
    // The MFMA phase runs at raised priority: the partner wave of this SIMD is in its transform phase (VALU-dense), and VALU-class
    // issue is arbitrated by priority, then AGE -- without this the younger waves' MFMAs starved behind the older waves' transform
    // (s_memtime trace, profiles/r6_wino_trace.txt: 3 300 cycles for 24 MFMAs beside a transforming partner, 1 370 alone).
    __builtin_amdgcn_s_setprio(WINO_PRIO);
    WTR2(20);
    // four stages (xi slot s, tile group t): the two B fragments of stage i + 1 are read before the MFMAs of stage i
    u32x4 vbuf[2][2];
    vbuf[0][0] = vb[((2 * wave) * 2 + 0) * 128];
    vbuf[0][1] = vb[((2 * wave) * 2 + 1) * 128];
    u32x4 udn[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int s = i >> 1, t = i & 1, cur = i & 1;
      if (i + 1 < 4) {
        const int s1 = (i + 1) >> 1, t1 = (i + 1) & 1;
        vbuf[cur ^ 1][0] = vb[((2 * wave + s1) * 2 + 0) * 128 + t1 * 32];
        vbuf[cur ^ 1][1] = vb[((2 * wave + s1) * 2 + 1) * 128 + t1 * 32];
      }
      if (t == 0) {
#pragma unroll
        for (int q = 0; q < 2; ++q) udn[q] = h2_hi_down(ua[s][0][q]);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (WINO_ABL != 4) {
        // the two co-tiles' accumulators alternate: no MFMA waits for the one issued right before it
        f32x16 m0 = acc[s][0][t], m1 = acc[s][1][t];
        m0 = mma_h(ua[s][1][0], vbuf[cur][0], m0);            // lo * hi
        m1 = mma_h(ua[s][1][1], vbuf[cur][0], m1);
        m0 = mma_h(udn[0], vbuf[cur][1], m0);                 // (hi * 2^-11) * (lo * 2^11)
        m1 = mma_h(udn[1], vbuf[cur][1], m1);
        m0 = mma_h(ua[s][0][0], vbuf[cur][0], m0);            // hi * hi
        m1 = mma_h(ua[s][0][1], vbuf[cur][0], m1);
        acc[s][0][t] = m0;
        acc[s][1][t] = m1;
      }
      __builtin_amdgcn_sched_barrier(0);
      WTR2(21 + i);
      if (t == 1) {
        issue_u(s, cnext);                                  // the next chunk's fragments of this slot (clamped re-read at the end)
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_s_setprio(0);
  };

  // ---- main loop over the 16-channel chunks ----
  const int c_begin = 0, c_end = a.nchunk;
  const bool tr_first = WINO_ABL == 5 || (WINO_TR_FIRST_OLD ? wave < 4 : wave >= 4);          // (wave-uniform)
  issue_raw(c_begin);
  issue_u(0, c_begin);
  issue_u(1, c_begin);
  publish_raw();
  __syncthreads();
  if (c_begin + 1 < c_end) issue_raw(c_begin + 1);
  transform(0);
  __syncthreads();
  for (int c = c_begin; c < c_end; ++c) {
    const int nb = (c - c_begin) & 1;
    const bool more = c + 1 < c_end;
    WTR(0);
    if (more) publish_raw();                                // chunk c + 1 (loaded one iteration ago)
    WTR(1);
    __syncthreads();
    WTR(2);
    issue_raw(c + 2 < c_end ? c + 2 : c_end - 1);            // (unconditional: a load behind a branch turns the graded vmcnt waits into vmcnt(0))
    const int cnext = more ? c + 1 : c;
    // ONE copy of the MFMA phase at a fixed place (two copies in the arms of a branch cost 277 spilled registers: the accumulators
    // did not stay in place; a two-trip loop with the order as a run-time choice hid from the compiler that the phase runs exactly
    // once, and its vmcnt waits for the U fragments then also waited for the raw loads issued just before -- 2 800 cycles per chunk in
    // the waves that run their MFMAs first, profiles/r6_wino_trace.txt); the cheap transform phase has two guarded copies around it.
    if (tr_first && more) transform(nb ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    WTR(3);
    mfma(nb, cnext);
    __builtin_amdgcn_sched_barrier(0);
    WTR(4);
    if (!tr_first && more) transform(nb ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    WTR(7);
    __syncthreads();
    WTR(5);
  }

  // ---- epilogue: back to the operands' scale, exchange through LDS, output transform ----
  float* const el = (float*)lds;                            // [xi][co 32][tile 64]
  float ymax = 0.f;
  const bool want_amax = a.y_amax != nullptr;
  const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)(a.y + (long)b * a.y_bs), (short)0, (int)0x80000000u, 0x00020000);
  const bool w_even = (a.W & 1) == 0;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    if (q == 1) __syncthreads();
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = (r & 3) + 8 * (r >> 2) + 4 * g;
          el[((2 * wave + s) * 32 + i) * 64 + t * 32 + j] = (acc[s][q][t][r] * inv_x) * inv_w;
        }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int pi = it * 512 + tid;
      const int tile = pi & 63, col = pi >> 6;
      const int co = (cot0 + q) * 32 + col;
      float m[16];
#pragma unroll
      for (int xi = 0; xi < 16; ++xi) m[xi] = el[(xi * 32 + col) * 64 + tile];
      float sr[2][4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        sr[0][c] = (m[c] + m[4 + c]) + m[8 + c];
        sr[1][c] = (m[4 + c] - m[8 + c]) - m[12 + c];
      }
      const float bv = (a.bias && co < a.Cout) ? a.bias[co] : 0.f;
      const int oy = y0 + 2 * (tile >> 3), ox = x0 + 2 * (tile & 7);
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        float o0 = (sr[p][0] + sr[p][1]) + sr[p][2] + bv;
        float o1 = (sr[p][1] - sr[p][2]) - sr[p][3] + bv;
        if (a.lrelu) { o0 = irr_lrelu(o0); o1 = irr_lrelu(o1); }
        o0 *= a.alpha; o1 *= a.alpha;
        const bool okr = co < a.Cout && oy + p < a.H;
        const bool ok0 = okr && ox < a.W, ok1 = okr && ox + 1 < a.W;
        if (want_amax) { if (ok0) ymax = x3_amax_fold(ymax, o0); if (ok1) ymax = x3_amax_fold(ymax, o1); }
        const uint32_t vo = (uint32_t)(((long)co * hw + (long)(oy + p) * a.W + ox) * 4);
        if (w_even) {
          const f32x2 ov = {o0, o1};
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2v, ov), yr, (int)(ok0 ? vo : WOOB), 0, 0);
        } else {
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, o0), yr, (int)(ok0 ? vo : WOOB), 0, 0);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, o1), yr, (int)(ok1 ? vo + 4 : WOOB), 0, 0);
        }
      }
    }
  }
  if (want_amax) x3_amax_publish(ymax, a.y_amax);
  WTR(6);
#ifdef WINO_TRACE
  if (blockIdx.x == 7 && lane == 0) a.dbg[8 * 1024 + wave] = (unsigned long long)ntr;
#endif
}

// ---- weight pack: uq[(((chunk * 16 + xi) * 2 + piece) * CoT + cot) * 64 + lane] = 8 fp16 of U[xi] * 2^ew (k-group lane >> 5, row lane & 31) --
// mode 0: w is (Cout, Cin, 3, 3) -> forward; mode 1: w is (Cin, Cout, 3, 3), used transposed + flipped -> stride-1 data gradient.
// amax[0] >= max |w|: |U| <= 2.25 max |w| bounds the one scale of the packed matrix.
__global__ __launch_bounds__(256) void pack_wino_h2_kernel(const float* __restrict__ w, u32x4* __restrict__ uq, int Cin, int Cout, int CoT,
                                                           int nchunk, int mode, const float* __restrict__ amax, long nunits) {
  const long u = (long)blockIdx.x * 256 + threadIdx.x;
  if (u >= nunits) return;
  const int lane = (int)(u & 63);
  long r = u >> 6;
  const int cot = (int)(r % CoT);
  r /= CoT;
  const int xi = (int)(r % 16);
  const int chunk = (int)(r / 16);
  const int g = lane >> 5, i = lane & 31;
  const int co = cot * 32 + i;
  const bool tail = (chunk == nchunk - 1) && (Cin & 15);
  const int ch0 = (tail ? Cin - 16 : chunk * 16) + 8 * g;
  const int rr = xi >> 2, cc = xi & 3;
  const float G[4][3] = {{1.f, 0.f, 0.f}, {.5f, .5f, .5f}, {.5f, -.5f, .5f}, {0.f, 0.f, 1.f}};
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int ci = ch0 + e;
    float val = 0.f;
    const bool dup = tail && ci < (nchunk - 1) * 16;
    if (!dup && ci >= 0 && ci < Cin && co < Cout) {
      float gk[3][3];
#pragma unroll
      for (int t = 0; t < 9; ++t)
        gk[t / 3][t % 3] = mode == 0 ? w[((long)co * Cin + ci) * 9 + t] : w[((long)ci * Cout + co) * 9 + (8 - t)];
      float tmp[3];                                          // (G g)[rr][:]
#pragma unroll
      for (int bb = 0; bb < 3; ++bb) tmp[bb] = (G[rr][0] * gk[0][bb] + G[rr][1] * gk[1][bb]) + G[rr][2] * gk[2][bb];
      val = (tmp[0] * G[cc][0] + tmp[1] * G[cc][1]) + tmp[2] * G[cc][2];
    }
    v[e] = val;
  }
  const int ew = x3_h2_exp(2.25f * amax[0]);
  if (u == 0) uq[(long)nchunk * 16 * 2 * CoT * 64] = u32x4{(uint32_t)ew, 0u, 0u, 0u};
  u32x4 h, m;
  split8_h2(v, ldexpf(1.f, ew), h, m);
  const long base = (((long)chunk * 16 + xi) * 2 * CoT + cot) * 64 + lane;
  uq[base] = h;
  uq[base + (long)CoT * 64] = m;
}

inline int wino_cot(int Cout) { return 2 * ((Cout + 63) / 64); }
inline int wino_nchunk(int Cin) { return (Cin + 15) / 16; }

}  // namespace

extern "C" long irr_conv_wino_packed_bytes(int Cin, int Cout) {
  if (Cin < 16 || Cout <= 0) return IRR_EINVAL;
  return ((long)wino_nchunk(Cin) * 16 * 2 * wino_cot(Cout) * 64 + 1) * 16;
}

extern "C" int irr_conv_pack_weights_wino_h2(const float* w, void* uq, int Cin, int Cout, int transpose, const float* amax, void* stream) {
  if (!w || !uq || !amax || Cin < 16 || Cout <= 0) return IRR_EINVAL;
  const int CoT = wino_cot(Cout), nchunk = wino_nchunk(Cin);
  const long nunits = (long)nchunk * 16 * CoT * 64;
  hipLaunchKernelGGL(pack_wino_h2_kernel, dim3((unsigned)((nunits + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, (u32x4*)uq, Cin, Cout,
                     CoT, nchunk, transpose ? 1 : 0, amax, nunits);
  IRR_LAUNCH_CHECK();
  return 0;
}

#ifdef WINO_TRACE
extern "C" int irr_wino_trace_read(unsigned long long* host) {      // (trace builds only; not part of include/irr_hip.h)
  if (!g_wino_dbg) return IRR_EINVAL;
  IRR_HIP_TRY(hipDeviceSynchronize());
  IRR_HIP_TRY(hipMemcpy(host, g_wino_dbg, (8 * 1024 + 8) * 8, hipMemcpyDeviceToHost));
  return 0;
}
#endif

extern "C" int irr_conv2d_wino_eligible(int B, int Cin, int H, int W, int Cout) {
  if (B <= 0 || Cin < 16 || Cout <= 0 || H <= 0 || W <= 0 || (W & 3)) return 0;          // (16-byte row loads)
  if ((long)Cin * H * W * 4 >= (1L << 31) || (long)Cout * H * W * 4 >= (1L << 31)) return 0;
  return 1;
}

extern "C" int irr_conv2d_wino_fwd_h2(const float* x, const void* uq, const float* bias, float* y, int B, int Cin, int H, int W, int Cout,
                                      long x_bs, long y_bs, int lrelu, float alpha, const float* x_amax, int n_amax, float* y_amax,
                                      void* stream) {
  if (!x || !uq || !y || !x_amax || n_amax <= 0 || !irr_conv2d_wino_eligible(B, Cin, H, W, Cout)) return IRR_EINVAL;
  if (((uintptr_t)x & 15) || (x_bs & 3) || (((long)H * W) & 3)) return IRR_EINVAL;
  WinoArgs a;
  a.x = x; a.uq = (const u32x4*)uq; a.bias = bias; a.y = y;
  a.B = B; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout;
  a.nchunk = wino_nchunk(Cin); a.CoT = wino_cot(Cout);
  a.tiles_x = (W + 15) / 16; a.tiles_y = (H + 15) / 16; a.ngy = a.CoT / 2;
  a.x_bs = x_bs; a.y_bs = y_bs; a.lrelu = lrelu; a.alpha = alpha;
  a.x_amax = x_amax; a.n_amax = n_amax; a.y_amax = y_amax;
  a.dbg = nullptr;
#ifdef WINO_TRACE
  if (!g_wino_dbg) { IRR_HIP_TRY(hipMalloc((void**)&g_wino_dbg, (8 * 1024 + 8) * 8)); }
  a.dbg = g_wino_dbg;
#endif
  constexpr size_t lds_bytes = (size_t)2 * VUNITS * 16 + (size_t)16 * RAWCH * 4;
  static_assert(lds_bytes <= 160 * 1024, "V double buffer + raw patch must fit the 160 KiB LDS");
  static bool attr_set = false;
  if (!attr_set) {
    IRR_HIP_TRY(hipFuncSetAttribute((const void*)conv_wino_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    attr_set = true;
  }
  const long nblk = (long)B * a.tiles_x * a.tiles_y * a.ngy;
  if (nblk <= 0 || nblk >= (1L << 31)) return IRR_EINVAL;
  hipLaunchKernelGGL(conv_wino_kernel, dim3((unsigned)nblk), dim3(512), lds_bytes, (hipStream_t)stream, a);
  IRR_LAUNCH_CHECK();
  return 0;
}
