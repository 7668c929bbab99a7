// Weight packing, per-element device functions shared by the single-job launchers (conv_fwd.hip, conv_x3.hip) and the
// batched launch that repacks every registered weight of a model in ONE dispatch after an optimizer step
// (pack_batch.hip; a train step used to spend ~250 launches / 3.3 ms on them).
#pragma once
#include "x3_split.h"

struct IrrPackJob {            // opaque to the host: size and the block0 offset are exported (irr_conv_pack_job_bytes / _block0_offset); filled by irr_conv_pack_job_*
  const float* w;             // source weights (device)
  void* dst;                  // packed destination (device)
  long n;                     // elements (kind 0/1) or 16-B units x lanes (kind 2) of the job
  long block0;                // first 256-thread block of the job inside the batched launch
  int kind;                   // 0 = fp32 pack, 1 = fp32 combined-matrix sub-block, 2 = bf16x3 pack (modes 0 / 1 / 2), 3 = fp16x2 pack (same modes)
  int p[9];
  const float* amax;          // kind 3: device scalar >= max |w| over EVERY weight that goes into dst (the pack's one scale)
};

// fp32 packed layout wp[cp][tap][half][CoP]  (conv_fwd.hip)
__device__ __forceinline__ void pack_f32_elem(const float* __restrict__ w, float* __restrict__ wp, int Cin, int Cout, int KK, int CoP,
                                              int transpose, long n, long i) {
  if (i >= n - 4 * 32) { wp[i] = 0.f; return; }       // tail slack
  const int co = (int)(i % CoP);
  long r = i / CoP;
  const int half = (int)(r & 1);
  r >>= 1;
  const int tap = (int)(r % KK);
  const int cp = (int)(r / KK);
  int ci = 2 * cp + half;
  bool zero_row = false;
  if ((Cin & 1) && cp == (Cin >> 1)) {        // odd tail: lanes read channels (Cin-2, Cin-1); Cin-2 was already consumed
    ci = Cin - 2 + half;
    zero_row = (half == 0);
  }
  float v = 0.f;
  if (!transpose) {
    if (!zero_row && ci < Cin && co < Cout) v = w[((long)co * Cin + ci) * KK + tap];
  } else {
    // logical conv': Cin' = Cout(orig), Cout' = Cin(orig): here (Cin, Cout) are ALREADY the swapped sizes;
    // w is the original (Cout_orig = Cin, Cin_orig = Cout) tensor: w[ci][co][KK-1-tap]
    if (!zero_row && ci < Cin && co < Cout) v = w[((long)ci * Cout + co) * KK + (KK - 1 - tap)];
  }
  wp[i] = v;
}

// Sub-block pack for COMBINED data-gradient weights (DenseNet backward): rows [row_offset, row_offset + w_cout) of the
// packed matrix take the transposed+flipped weights of one layer restricted to its input channels
// [chan0, chan0 + nchan):  wp[((r/2*KK + tap)*2 + (r&1))*CoP + c] = w[r - row_offset][chan0 + c][KK-1-tap].
__device__ __forceinline__ void pack_sub_elem(const float* __restrict__ w, float* __restrict__ wp, int w_cin, int KK, int chan0, int nchan,
                                              int CoP, int row_offset, long i) {
  const int c = (int)(i % CoP);
  long r1 = i / CoP;
  const int tap = (int)(r1 % KK);
  const int rl = (int)(r1 / KK);                 // local row = output channel of the layer
  const int r = row_offset + rl;
  float v = 0.f;
  if (c < nchan) v = w[((long)rl * w_cin + chan0 + c) * KK + (KK - 1 - tap)];
  wp[(((long)(r >> 1) * KK + tap) * 2 + (r & 1)) * CoP + c] = v;
}

// ---- bf16x3 packing: wq[(((chunk*9 + tap)*3 + piece)*CoT + cot)*64 + lane] = 8 bf16 (k-group g = lane>>5, row i = lane&31) ----
// mode 0: w is (Cout, Cin, 3, 3)                    -> forward
// mode 1: w is (Cin, Cout, 3, 3) = original layout, used transposed + flipped -> stride-1 data gradient
// mode 2: sub-block of a COMBINED data-gradient matrix (DenseNet backward): rows [row_offset, row_offset + w_cout)
//         take layer weights w (w_cout, w_cin, 3, 3) transposed+flipped, restricted to input channels [chan0, chan0+Cout)
// NP = 2 (kind 3, "h2"): two fp16 pieces of w * 2^ew, ew from *amax (x3_h2_exp); piece stride as for NP = 3; the 16-B unit behind the
// last fragment holds ew (every job of a combined matrix writes the same value there).
template <int NP = 3>
__device__ __forceinline__ void pack_x3_unit(const float* __restrict__ w, u32x4* __restrict__ wq, int Cin, int Cout, int CoT, int nchunk,
                                             int mode, int w_cin, int chan0, int row_offset, int w_cout, long u,
                                             const float* __restrict__ amax = nullptr) {
  const int lane = (int)(u & 63);
  long r = u >> 6;
  const int cot = (int)(r % CoT);
  r /= CoT;
  const int tap = (int)(r % 9);
  const int chunk = (int)(r / 9);
  const int g = lane >> 5, i = lane & 31;
  const int co = cot * 32 + i;
  const bool tail = (chunk == nchunk - 1) && (Cin & 15);
  const int ch0 = (tail ? Cin - 16 : chunk * 16) + 8 * g;
  float v[8];
  bool any = false;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int ci = ch0 + e;
    float val = 0.f;
    const bool dup = tail && ci < (nchunk - 1) * 16;
    if (!dup && ci >= 0 && ci < Cin && co < Cout) {
      if (mode == 0) val = w[((long)co * Cin + ci) * 9 + tap];
      else if (mode == 1) val = w[((long)ci * Cout + co) * 9 + (8 - tap)];
      else if (ci >= row_offset && ci < row_offset + w_cout) {
        val = w[((long)(ci - row_offset) * w_cin + chan0 + co) * 9 + (8 - tap)];
        any = true;
      }
    }
    v[e] = val;
  }
  int ew = 0;
  if (NP == 2) {
    ew = x3_h2_exp(amax[0]);
    if (u == 0) wq[(long)nchunk * 9 * NP * CoT * 64] = u32x4{(uint32_t)ew, 0u, 0u, 0u};
  }
  if (mode == 2 && !any) return;                   // rows of other layers: leave untouched
  u32x4 h, m, l;
  if (NP == 2) split8_h2(v, ldexpf(1.f, ew), h, m);
  else split8(v, h, m, l);
  const long base = (((long)chunk * 9 + tap) * NP * CoT + cot) * 64 + lane;
  wq[base] = h;
  wq[base + (long)CoT * 64] = m;
  if (NP == 3) wq[base + 2L * CoT * 64] = l;
}
