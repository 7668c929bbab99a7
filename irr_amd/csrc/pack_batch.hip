// One dispatch that repacks EVERY registered conv weight of a model after an optimizer step: fp32 packs (forward and
// transposed), the combined DenseNet data-gradient matrices and the pre-split bf16x3 packs.  The host keeps a table of
// IrrPackJob records on the device (built once; sources and destinations are persistent buffers); block b finds its job
// by binary search over the jobs' first-block prefix and runs the job's per-element function (pack.h).
#include <cstddef>
#include "pack.h"

namespace {

__global__ __launch_bounds__(256) void pack_batch_kernel(const IrrPackJob* __restrict__ jobs, int njobs) {
  const long b = blockIdx.x;
  int lo = 0, hi = njobs - 1;                       // last job with block0 <= b
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].block0 <= b) lo = mid;
    else hi = mid - 1;
  }
  const IrrPackJob j = jobs[lo];
  const long i = (b - j.block0) * 256 + threadIdx.x;
  if (i >= j.n) return;
  if (j.kind == 0) pack_f32_elem(j.w, (float*)j.dst, j.p[0], j.p[1], j.p[2], j.p[3], j.p[4], j.n, i);
  else if (j.kind == 1) pack_sub_elem(j.w, (float*)j.dst, j.p[0], j.p[2], j.p[3], j.p[4], j.p[5], j.p[6], i);
  else if (j.kind == 2) pack_x3_unit<3>(j.w, (u32x4*)j.dst, j.p[0], j.p[1], j.p[2], j.p[3], j.p[4], j.p[5], j.p[6], j.p[7], j.p[8], i);
  else pack_x3_unit<2>(j.w, (u32x4*)j.dst, j.p[0], j.p[1], j.p[2], j.p[3], j.p[4], j.p[5], j.p[6], j.p[7], j.p[8], i, j.amax);
}

static int x3_nchunk_(int Cin, int Cout) { return (Cin == 16 && Cout <= 32) ? 2 : (Cin + 15) / 16; }     // = conv_x3.hip

}  // namespace

extern "C" int irr_conv_pack_job_bytes(void) { return (int)sizeof(IrrPackJob); }
extern "C" int irr_conv_pack_job_block0_offset(void) { return (int)offsetof(IrrPackJob, block0); }

// The four job builders take the arguments of the single-job launchers (irr_conv_pack_weights_f32 / _sub_f32 / _x3 / _x3_sub)
// and write the record into HOST memory `job`; they return the number of 256-thread blocks the job needs (< 0: IRR_EINVAL).
// block0 is left 0: the caller lays the jobs out one after another.
extern "C" long irr_conv_pack_job_f32(void* job, const float* w, float* wp, int Cin, int Cout, int k, int transpose) {
  if (!job || !w || !wp || Cin <= 0 || Cout <= 0 || (k != 1 && k != 3)) return IRR_EINVAL;
  IrrPackJob j{};
  j.w = w; j.dst = wp; j.kind = 0;
  j.n = irr_conv_packed_weight_elems(Cin, Cout, k);
  j.p[0] = Cin; j.p[1] = Cout; j.p[2] = k * k; j.p[3] = (Cout + 31) / 32 * 32; j.p[4] = transpose ? 1 : 0;
  *(IrrPackJob*)job = j;
  return (j.n + 255) / 256;
}

extern "C" long irr_conv_pack_job_sub_f32(void* job, const float* w, float* wp, int w_cin, int w_cout, int k, int chan0, int nchan,
                                          int CoP, int row_offset) {
  if (!job || !w || !wp || w_cin <= 0 || w_cout <= 0 || (k != 1 && k != 3) || chan0 < 0 || nchan <= 0 || chan0 + nchan > w_cin ||
      CoP < nchan || (CoP & 31) || row_offset < 0)
    return IRR_EINVAL;
  IrrPackJob j{};
  j.w = w; j.dst = wp; j.kind = 1;
  j.n = (long)w_cout * k * k * CoP;
  j.p[0] = w_cin; j.p[1] = w_cout; j.p[2] = k * k; j.p[3] = chan0; j.p[4] = nchan; j.p[5] = CoP; j.p[6] = row_offset;
  *(IrrPackJob*)job = j;
  return (j.n + 255) / 256;
}

extern "C" long irr_conv_pack_job_x3(void* job, const float* w, void* wq, int Cin, int Cout, int transpose) {
  if (!job || !w || !wq || Cin < 16 || Cout <= 0) return IRR_EINVAL;
  IrrPackJob j{};
  const int CoT = (Cout + 31) / 32, nchunk = x3_nchunk_(Cin, Cout);
  j.w = w; j.dst = wq; j.kind = 2;
  j.n = (long)nchunk * 9 * CoT * 64;
  j.p[0] = Cin; j.p[1] = Cout; j.p[2] = CoT; j.p[3] = nchunk; j.p[4] = transpose ? 1 : 0;
  *(IrrPackJob*)job = j;
  return (j.n + 255) / 256;
}

extern "C" long irr_conv_pack_job_x3_sub(void* job, const float* w, void* wq, int w_cin, int w_cout, int total_rows, int chan0,
                                         int nchan, int row_offset) {
  if (!job || !w || !wq || w_cin <= 0 || w_cout <= 0 || total_rows < 16 || (total_rows & 15) || chan0 < 0 || nchan <= 0 ||
      chan0 + nchan > w_cin || row_offset < 0 || (row_offset & 7) || (w_cout & 7) || row_offset + w_cout > total_rows)
    return IRR_EINVAL;
  IrrPackJob j{};
  const int CoT = (nchan + 31) / 32, nchunk = total_rows / 16;
  j.w = w; j.dst = wq; j.kind = 2;
  j.n = (long)nchunk * 9 * CoT * 64;
  j.p[0] = total_rows; j.p[1] = nchan; j.p[2] = CoT; j.p[3] = nchunk; j.p[4] = 2; j.p[5] = w_cin; j.p[6] = chan0;
  j.p[7] = row_offset; j.p[8] = w_cout;
  *(IrrPackJob*)job = j;
  return (j.n + 255) / 256;
}

// fp16x2 ("h2") forms of the two builders above: amax = device scalar >= max |w| over every weight that goes into wq
extern "C" long irr_conv_pack_job_h2(void* job, const float* w, void* wq, int Cin, int Cout, int transpose, const float* amax) {
  if (!amax) return IRR_EINVAL;
  const long nb = irr_conv_pack_job_x3(job, w, wq, Cin, Cout, transpose);
  if (nb < 0) return nb;
  ((IrrPackJob*)job)->kind = 3;
  ((IrrPackJob*)job)->amax = amax;
  return nb;
}

extern "C" long irr_conv_pack_job_h2_sub(void* job, const float* w, void* wq, int w_cin, int w_cout, int total_rows, int chan0,
                                         int nchan, int row_offset, const float* amax) {
  if (!amax) return IRR_EINVAL;
  const long nb = irr_conv_pack_job_x3_sub(job, w, wq, w_cin, w_cout, total_rows, chan0, nchan, row_offset);
  if (nb < 0) return nb;
  ((IrrPackJob*)job)->kind = 3;
  ((IrrPackJob*)job)->amax = amax;
  return nb;
}

// jobs: DEVICE array of njobs records whose block0 fields are the exclusive prefix sums of the jobs' block counts
// (nblocks = their total).  Jobs that write the same destination (sub-blocks of one combined matrix) touch disjoint elements.
extern "C" int irr_conv_pack_batch(const void* jobs, int njobs, long nblocks, void* stream) {
  if (!jobs || njobs <= 0 || nblocks <= 0 || nblocks > 0x7fffffffL) return IRR_EINVAL;
  hipLaunchKernelGGL(pack_batch_kernel, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream, (const IrrPackJob*)jobs, njobs);
  IRR_LAUNCH_CHECK();
  return 0;
}
