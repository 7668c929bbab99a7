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
// Layout of the work (design 2; design 1 -- V through LDS, transform and MFMA phases staggered between the two waves of a SIMD -- was
// correct and 1.07x the direct kernel: every chunk pushed 91 KiB through the 64 B/clk LDS write port next to 91 KiB through the
// 64 B/clk vector-memory path, with two block-wide barriers; profiles/r6_wino_v1_*).
// ONE block of eight waves per CU computes 64 output channels x 64 tiles (16 x 16 pixels) with all 16 xi: 64 * 64 * 16 fp32
// accumulators = 256 KiB = half the CU's register file.  Wave w owns xi = (r, c0), (r, c0 + 1) with r = w >> 1, c0 = 2 (w & 1), for both
// 32-channel co-tiles and both 32-tile groups (128 accumulator registers): no U fragment is needed by two waves, and -- the point of
// this design -- no V element either, IF every wave builds its own B fragments:
//   * the raw 18 x 24 x 16 patch of a chunk (columns x0 - 4 .. x0 + 19: whole 16-byte quads) goes global -> LDS by LDS-DMA
//     (buffer_load_dwordx4 ... lds, 27 + 5 wave instructions per chunk, no registers, no store pass), three buffers deep;
//   * a lane's B fragment = (tile lane & 31, channels 8 (lane >> 5) .. + 7).  Its two xi share the row operator (rows ra, rb:
//     t = d[ra] +- d[rb]) and need three neighbouring columns: per channel one ds_read_b64 + one ds_read_b32 per row, three FMAs, two
//     adds, then the pair split of two channels at a time (v_cvt_pk_f16_f32 packs them);
//   * U streams from L2 straight into registers (1 KiB per fragment, 64 KiB per chunk and block), as before.
// One barrier per chunk (raw buffer hand-over), all waves symmetric.  Every vector-memory instruction of the loop is inline assembly
// with hand-counted s_waitcnt vmcnt: hipcc does not count LDS-DMA operations and would wait vmcnt(0) at every use of a loaded
// register while one is in flight (cdna_hip_programming.md, "Pipelining across barriers").
//   Epilogue: the accumulators of one co-tile at a time go through LDS ([xi][co][tile] fp32 = 128 KiB), every thread output-
//   transforms four (co, tile) pairs: bias, LeakyReLU, alpha, 2x2 pixels as two 8-byte stores, max |y| folded into y_amax.
#include "x3_split.h"
#include "amax.h"

#ifndef WINO_M0_NOPS
#define WINO_M0_NOPS "7"
#endif
#ifndef WINO_PARANOID
#define WINO_PARANOID 0 // (diagnosis) bit 0: drain right behind every LDS-DMA group; bit 1: behind every group of U loads; bit 2: sleep at the end of an iteration
#endif
#ifndef WINO_ABL
#define WINO_ABL 0      // ablation builds (timing only, results wrong): 1 = U fragments loaded once, 2 = B fragments built once,
#endif                  // 3 = no LDS-DMA after the prologue, 4 = no MFMAs

#ifdef WINO_TRACE
static unsigned long long* g_wino_dbg = nullptr;     // s_memtime trace (tag builds with -DWINO_TRACE=1, tools/wino_trace.py): block 7, lane 0 of every wave
#endif

namespace {

constexpr uint32_t WOOB = 0x80000000u;
typedef unsigned int u32x2v __attribute__((__vector_size__(2 * sizeof(unsigned int))));
constexpr int RP = 24;                         // dwords per raw patch row: columns x0 - 4 .. x0 + 19 (patch column pc = x - (x0 - 1) is stored at pc + 3)
constexpr int RCH = 18 * RP;                   // dwords per raw channel
constexpr int RBUF = 8192;                     // dwords per raw buffer (16 * RCH = 6912 used; 32 DMA instructions x 1 KiB)

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

#ifdef WINO_TRACE
#define WTR(slot) do { if (blockIdx.x == 7 && lane == 0 && ntr < 1024) { dbgp[ntr++] = ((unsigned long long)(slot) << 56) | (__builtin_amdgcn_s_memtime() & 0xffffffffffffffull); } } while (0)
#else
#define WTR(slot) do {} while (0)
#endif

// 128-bit buffer resource as four SGPR words (inline assembly takes it as one "s" operand)
__device__ __forceinline__ u32x4 wino_rsrc(const void* p, uint32_t bytes) {
  const uint64_t a = (uint64_t)p;
  u32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((uint32_t)a);
  r[1] = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32) & 0xffffu);
  r[2] = bytes;
  r[3] = 0x00020000u;
  return r;
}
// LDS-DMA: 64 lanes x 16 bytes from per-lane global offsets to the wave-uniform LDS address m0v (+ lane * 16); not counted by hipcc
__device__ __forceinline__ void wino_dma16(uint32_t voff, u32x4 rs, uint32_t soff, uint32_t m0v) {
  uint32_t keep;
  soff = __builtin_amdgcn_readfirstlane(soff);              // (an "s" operand the compiler holds in a VGPR is printed as one: assembler error)
  m0v = __builtin_amdgcn_readfirstlane(m0v);
  // (wait states between the instruction and the next write of M0: with M0 restored in the very next instruction, lanes 12..15 / 8..15 of
  // some 16-lane rows went to the OLD address on a loaded memory system -- the instruction reads M0 late, the way a wide buffer store
  // reads its data registers late, common.h)
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_nop " WINO_M0_NOPS "\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rs), "s"(m0v), "s"(soff) : "memory");
}
__device__ __forceinline__ u32x4 wino_load16(uint32_t voff, u32x4 rs, uint32_t soff) {
  u32x4 d;
  soff = __builtin_amdgcn_readfirstlane(soff);
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(d) : "v"(voff), "s"(rs), "s"(soff) : "memory");
  return d;
}

__global__ __launch_bounds__(512) void conv_wino_kernel(const WinoArgs a) {
  extern __shared__ u32x4 lds[];
  float* const rawl = (float*)lds;                          // three raw buffers of RBUF dwords; the epilogue's exchange area afterwards

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

  const u32x4 xrs = wino_rsrc(a.x + (long)b * a.x_bs, 0x80000000u);
  const u32x4 urs = wino_rsrc(a.uq, 0xffffffffu);

  // ---- operand scales ----
  const int ex = x3_h2_exp(x3_h2_amax(a.x_amax, a.n_amax)) - 2;       // |V| <= 4 max |x|
  const int ew = ((const int*)(a.uq + (long)a.nchunk * 16 * 2 * a.CoT * 64))[0];
  const float sx = ldexpf(1.f, ex), sxu = sx * H2_LO_UP, inv_x = ldexpf(1.f, -ex), inv_w = ldexpf(1.f, -ew);

  // ---- LDS-DMA role: instruction i = wave + 8 r (r = 0..3, i < 27) moves the 16-byte units 64 i .. 64 i + 63 of the buffer; unit
  // L = (channel * 18 + row) * 6 + quad.  Positions outside the image fetch nothing and STORE nothing (the buffers are zero-filled once
  // per block: they are the same positions in every chunk).
  uint32_t dvoff[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int L = 64 * (wave + 8 * r) + lane;
    const int ch = L / 108, rem = L - ch * 108;
    const int row = rem / 6, quad = rem - row * 6;
    const int iy = y0 - 1 + row, ix = x0 - 4 + 4 * quad;
    dvoff[r] = (L < 1728 && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) ? (uint32_t)(((long)ch * hw + (long)iy * a.W + ix) * 4) : WOOB;
  }
  const uint32_t lds0 = (uint32_t)(uintptr_t)rawl;
  const int tail_base = a.Cin - 16;                          // the last chunk re-reads [Cin-16, Cin) (duplicates have zero weights)
  auto issue_dma = [&](int c, int buf) {                     // three or four operations per wave (27 instructions per chunk)
    const int ch0 = (c == a.nchunk - 1) ? tail_base : c * 16;
    const uint32_t s0 = (uint32_t)ch0 * hw4;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (wave + 8 * r < 27) wino_dma16(dvoff[r], xrs, s0, lds0 + (uint32_t)(buf * RBUF * 4 + (wave + 8 * r) * 1024));
  };

  // ---- fragment roles ----
  const int j = lane & 31, g = lane >> 5;
  const int wr = wave >> 1, wc0 = 2 * (wave & 1);            // xi = (wr, wc0), (wr, wc0 + 1) = 2 wave, 2 wave + 1
  // row operator of xi row wr: t = d[ra] + sgn * d[rb]
  const int ra = wr == 0 ? 0 : wr == 2 ? 2 : 1, rb = wr == 0 ? 2 : wr == 1 ? 2 : wr == 2 ? 1 : 3;
  const float sgn = wr == 1 ? 1.f : -1.f;
  // the lane's tile of tile group t: tile = 32 t + j (tile row tile >> 3, tile column tile & 7); patch column 2 tcol is stored at 2 tcol + 3
  const int fbase = (8 * g) * RCH + (2 * (j >> 3)) * RP + 2 * (j & 7) + 4;     // + t * 8 * RP; the aligned pair (patch columns 1, 2 of the tile)
  const int foa = ra * RP, fob = rb * RP;
  const int f32o = wc0 == 0 ? -1 : 2;                        // the third column: patch column 0 (left of the pair) or 3 (right of it)

  const int cot0 = by * 2;
  const uint32_t uvoff = (uint32_t)(lane * 16);
  const uint32_t upiece = (uint32_t)a.CoT * 1024u;           // bytes between the two pieces of one xi
  u32x4 ua[2][2][2];                                        // [xi slot][piece][co-tile]
  auto issue_u = [&](int s, int c) {                         // four operations
    const uint32_t so = ((uint32_t)(c * 16 + 2 * wave + s) * 2u) * upiece + (uint32_t)cot0 * 1024u;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int q = 0; q < 2; ++q) ua[s][p][q] = wino_load16(uvoff, urs, so + p * upiece + q * 1024u);
  };
#define WINO_WAIT_U(s, n) asm volatile("s_waitcnt vmcnt(" #n ")" : "+v"(ua[s][0][0]), "+v"(ua[s][0][1]), "+v"(ua[s][1][0]), "+v"(ua[s][1][1]) :: "memory")

  f32x16 acc[2][2][2];                                      // [xi slot][co-tile][tile group]
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[s][q][t][r] = 0.f;

  // B fragments of one tile group: [xi slot][piece], dword p = channels 2 p, 2 p + 1 of the lane's k-group
  u32x4 fr[2][2][2];                                        // [buffer][xi slot][piece]
  // one channel pair of a fragment set: rows ra / rb, the aligned column pair + the third column, row operator, column operators, split
  auto build_pair = [&](const float* rb_, int fb, int p, int t) {
    float va[2], vb[2];
#if WINO_PARANOID & 64
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");      // (diagnosis: distance between the preceding MFMAs and this pair's register writes)
#endif
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const float* q_ = rb_ + fbase + t * 8 * RP + (2 * p + e) * RCH;
      const f32x2 A2 = *(const f32x2*)(q_ + foa), B2 = *(const f32x2*)(q_ + fob);
      const float A1 = q_[foa + f32o], B1 = q_[fob + f32o];
#if WINO_PARANOID & 16
      asm volatile("s_nop 3" ::: "memory");                  // (diagnosis: wait states behind every group of LDS reads)
#endif
      const float tp = __builtin_fmaf(B2[0], sgn, A2[0]), tq = __builtin_fmaf(B2[1], sgn, A2[1]), tw = __builtin_fmaf(B1, sgn, A1);
      // wc0 == 0: patch columns (0, 1, 2) = (tw, tp, tq): V[c = 0] = t0 - t2, V[1] = t1 + t2
      // wc0 == 2: patch columns (1, 2, 3) = (tp, tq, tw): V[c = 2] = t2 - t1, V[3] = t1 - t3
      va[e] = wc0 == 0 ? tw - tq : tq - tp;
      vb[e] = wc0 == 0 ? tp + tq : tp - tw;
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const f32x2 vv = {s == 0 ? va[0] : vb[0], s == 0 ? va[1] : vb[1]};
      const f32x2 vs = vv * sx, vu = vv * sxu;
      const f16x2 hp = __builtin_convertvector(vs, f16x2);
      const f32x2 rr = {__builtin_fmaf((float)hp[0], -H2_LO_UP, vu[0]), __builtin_fmaf((float)hp[1], -H2_LO_UP, vu[1])};
      fr[fb][s][0][p] = __builtin_bit_cast(uint32_t, hp);
      fr[fb][s][1][p] = __builtin_bit_cast(uint32_t, __builtin_convertvector(rr, f16x2));
    }
  };
  // three MFMAs of one (xi slot, co-tile) chain against fragment buffer fb, tile group t
  auto mfma3 = [&](int s, int q, int t, int fb) {
    if (WINO_ABL == 4) return;
    const u32x4 udn = h2_hi_down(ua[s][0][q]);
    f32x16 m = acc[s][q][t];
    m = mma_h(ua[s][1][q], fr[fb][s][0], m);                // lo * hi
    m = mma_h(udn, fr[fb][s][1], m);                        // (hi * 2^-11) * (lo * 2^11)
    m = mma_h(ua[s][0][q], fr[fb][s][0], m);                // hi * hi
    acc[s][q][t] = m;
  };

  // ---- main loop over the 16-channel chunks ----
  // Vector-memory operations per wave, in program order: prologue DMA(0) DMA(1) U(0)s0 U(0)s1, drained; iteration c:
  //   ... wait W0 ... wait W1, A = DMA(c + 2) ... B = U(c + 1) slot 0 [4] ... C = U(c + 1) slot 1 [4].
  // The waits rely on in-order completion AMONG the register loads only, never between them and the LDS-DMA operations (a first
  // version counted on one common order -- vmcnt(8) with DMA, B, C, DMA outstanding -- and used fragments that had not arrived, in a
  // few blocks per launch and only on a loaded memory system; an instruction whose 64 lanes are all out of range also dropped out of
  // that arithmetic):
  //   W0 (first use of slot 0): outstanding A(c-1), B(c-1), C(c-1).  vmcnt(4): of 8 + k operations all but four have completed; C
  //       cannot complete before B, so at most 3 + k completions leave B unfinished -- B is complete whatever the DMA operations did.
  //   W1 (first use of slot 1): vmcnt(0) -- C(c-1), and A(c-1) = DMA(c + 1), issued more than half an iteration earlier and due before
  //       the next barrier anyway.  DMA(c + 2) is issued right behind it, into the buffer chunk c - 1 was read from.
  const int n = a.nchunk;
  // Positions outside the image fetch nothing -- and an out-of-range LDS-DMA lane also STORES nothing.  They are the same positions in
  // every chunk, so the three buffers are zero-filled once per block (96 KiB of ds_write_b128: ~1 % of a block's time).
#pragma unroll
  for (int k = 0; k < 3 * RBUF / 4 / 512; ++k) lds[k * 512 + tid] = u32x4{0u, 0u, 0u, 0u};
  __syncthreads();
  issue_dma(0, 0);
  if (n > 1) issue_dma(1, 1);
  issue_u(0, 0);
  issue_u(1, 0);
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(ua[0][0][0]), "+v"(ua[0][0][1]), "+v"(ua[0][1][0]), "+v"(ua[0][1][1]), "+v"(ua[1][0][0]),
               "+v"(ua[1][0][1]), "+v"(ua[1][1][0]), "+v"(ua[1][1][1]) :: "memory");
  for (int c = 0; c < n; ++c) {
    WTR(0);
    __syncthreads();                                        // DMA(c) of every wave is in LDS (drained at W1 of iteration c - 1 / in the prologue)
    WTR(1);
    const float* const rbuf = rawl + (c % 3) * RBUF;
    const int cn = c + 1 < n ? c + 1 : c;
    // fragments of tile group 0
    if (!(WINO_ABL == 2 && c > 0)) {
#pragma unroll
      for (int p = 0; p < 4; ++p) build_pair(rbuf, 0, p, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
#if WINO_PARANOID & 32
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");        // (diagnosis: wait states between the last fragment write and the first MFMA)
#endif
    WTR(2);
#if WINO_PARANOID & 8
    WINO_WAIT_U(0, 0);
#else
    WINO_WAIT_U(0, 4);
#endif
    // tile group 0: twelve MFMAs in four chains, the channel pairs of tile group 1 built in between
    mfma3(0, 0, 0, 0);
    if (!(WINO_ABL == 2 && c > 0)) build_pair(rbuf, 1, 0, 1);
    __builtin_amdgcn_sched_barrier(0);
    mfma3(0, 1, 0, 0);
    if (!(WINO_ABL == 2 && c > 0)) build_pair(rbuf, 1, 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    WINO_WAIT_U(1, 0);
    if (c + 2 < n && WINO_ABL != 3) issue_dma(c + 2, (c + 2) % 3);
#if WINO_PARANOID & 1
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    mfma3(1, 0, 0, 0);
    if (!(WINO_ABL == 2 && c > 0)) build_pair(rbuf, 1, 2, 1);
    __builtin_amdgcn_sched_barrier(0);
    mfma3(1, 1, 0, 0);
    if (!(WINO_ABL == 2 && c > 0)) build_pair(rbuf, 1, 3, 1);
    __builtin_amdgcn_sched_barrier(0);
#if WINO_PARANOID & 32
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
#endif
    WTR(3);
    // tile group 1
    mfma3(0, 0, 1, 1);
    mfma3(0, 1, 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    if (!(WINO_ABL == 1 && c > 0)) issue_u(0, cn);
#if WINO_PARANOID & 2
    WINO_WAIT_U(0, 0);
#endif
    __builtin_amdgcn_sched_barrier(0);
    mfma3(1, 0, 1, 1);
    mfma3(1, 1, 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    if (!(WINO_ABL == 1 && c > 0)) issue_u(1, cn);
#if WINO_PARANOID & 2
    WINO_WAIT_U(1, 0);
#endif
#if WINO_PARANOID & 4
    __builtin_amdgcn_s_sleep(40);
#endif
    __builtin_amdgcn_sched_barrier(0);
    WTR(4);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the clamped re-reads of U in the last iteration
  __syncthreads();

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
  constexpr size_t lds_bytes = (size_t)16 * 32 * 64 * 4;      // the epilogue's exchange area; the three raw buffers (96 KiB) lie inside it
  static_assert(lds_bytes >= (size_t)3 * RBUF * 4 && lds_bytes <= 160 * 1024, "LDS plan");
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
