// Deferred reduction of the weight-gradient kernels' partial images (conv_wgrad.hip, conv_wgrad_x3.hip).
// Every weight-gradient launch leaves P partial [c1][tap][c2] images in its scratch and a tiny second kernel folds them into
// dW in a fixed order (bit-reproducible, no atomics).  A train step has ~210 of those folds, 10-20 us each, on the
// asynchronous weight-gradient lane where they queue behind millisecond kernels.  Between irr_wgrad_defer_begin() and
// irr_wgrad_defer_end() on the calling thread the launchers append a job record instead, and ONE launch of
// irr_wgrad_reduce_batch() folds up to IRR_REDUCE_BATCH_MAX of them (jobs travel by value in the kernel arguments).
#pragma once
#include "common.h"

struct IrrReduceJob {         // opaque to the host (irr_wgrad_job_bytes())
  const float* ws;            // P partial images of n floats each
  float* gw;                  // (Cout, Cin, k, k) gradient, accumulated into
  long n;                     // Cout * k*k * Cin
  int P;
  int Cin, Cout, KK;
  int swapped;                // the launch ran with the operand roles exchanged: ws[p][ci][8 - tap][co]
  int block0;                 // first 256-thread block of the job inside the batched launch
};

#define IRR_REDUCE_BATCH_MAX 40

struct IrrReduceCollector {
  IrrReduceJob* jobs;         // host memory owned by the caller of irr_wgrad_defer_begin
  int capacity, count;
};
extern thread_local IrrReduceCollector g_irr_reduce_collector;

// true: the job was recorded and the launcher must NOT run its own reduce kernel
static inline bool irr_reduce_defer(const float* ws, float* gw, long n, int P, int Cin, int Cout, int KK, int swapped) {
  IrrReduceCollector& c = g_irr_reduce_collector;
  if (!c.jobs || c.count >= c.capacity) return false;
  IrrReduceJob j{};
  j.ws = ws; j.gw = gw; j.n = n; j.P = P; j.Cin = Cin; j.Cout = Cout; j.KK = KK; j.swapped = swapped; j.block0 = 0;
  c.jobs[c.count++] = j;
  return true;
}
static inline bool irr_reduce_deferring() { return g_irr_reduce_collector.jobs != nullptr; }

// one (job, 256 consecutive workspace elements): 256 threads = 64 quads of elements x 4 partial lanes; a lane walks the partials
// pl, pl + 4, ... with FOUR 16-B loads in flight (the first version walked them one dependent 4-B load at a time over 64 elements
// per block: 2.8 ms of folds per train step for ~1.5 GB of partials).  Fixed summation order: a lane keeps four running sums
// (partials pl + 4k with k = 0, 1, 2, 3 mod 4), folds them ((s0 + s1) + (s2 + s3)), and the four lanes meet in LDS in lane order.
__device__ __forceinline__ void irr_reduce_block(const IrrReduceJob& J, long blk, float (*red)[256]) {
  const int jl = threadIdx.x & 63, pl = threadIdx.x >> 6;
  const long j0 = blk * 256 + 4 * jl;
  typedef float f4 __attribute__((ext_vector_type(4)));
  f4 s = {0.f, 0.f, 0.f, 0.f};
  if ((J.n & 3) == 0) {
    if (j0 < J.n) {
      // (component-wise adds on purpose: vector-typed arithmetic becomes v_pk_add_f32, and packed fp32 instructions gave wrong
      // results beside MFMA waves of another stream -- this kernel runs on the lane beside the main stream's convolutions; build.py)
      f4 a0 = s, a1 = s, a2 = s, a3 = s;
      auto add4 = [](f4& a, const f4 v) {
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] += v[e];
      };
      int p = pl;
      for (; p + 12 < J.P; p += 16) {
        const f4 v0 = *(const f4*)(J.ws + (long)p * J.n + j0), v1 = *(const f4*)(J.ws + (long)(p + 4) * J.n + j0),
                 v2 = *(const f4*)(J.ws + (long)(p + 8) * J.n + j0), v3 = *(const f4*)(J.ws + (long)(p + 12) * J.n + j0);
        add4(a0, v0); add4(a1, v1); add4(a2, v2); add4(a3, v3);
      }
      if (p < J.P) add4(a0, *(const f4*)(J.ws + (long)p * J.n + j0));
      if (p + 4 < J.P) add4(a1, *(const f4*)(J.ws + (long)(p + 4) * J.n + j0));
      if (p + 8 < J.P) add4(a2, *(const f4*)(J.ws + (long)(p + 8) * J.n + j0));
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] = (a0[e] + a1[e]) + (a2[e] + a3[e]);
    }
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float t = 0.f;
      if (j0 + e < J.n)
        for (int p = pl; p < J.P; p += 4) t += J.ws[(long)p * J.n + j0 + e];
      s[e] = t;
    }
  }
  if (pl > 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) red[pl - 1][4 * jl + e] = s[e];
  }
  __syncthreads();
  if (pl > 0) return;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const long j = j0 + e;
    if (j >= J.n) break;
    const float v = ((s[e] + red[0][4 * jl + e]) + red[1][4 * jl + e]) + red[2][4 * jl + e];
    // workspace element j = [c1][t][c2] with c2 the fastest (the "input channel" role of the launch)
    const int d2 = J.swapped ? J.Cout : J.Cin;
    const int c2 = (int)(j % d2);
    const long r = j / d2;
    const int t = (int)(r % J.KK);
    const long c1 = r / J.KK;
    const long dst = J.swapped ? ((long)c2 * J.Cin + c1) * J.KK + (J.KK - 1 - t) : (c1 * J.Cin + c2) * J.KK + t;
    J.gw[dst] += v;
  }
}
