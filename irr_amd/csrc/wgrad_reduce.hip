// Batched fold of weight-gradient partial images (see wgrad_reduce.h).
#include "wgrad_reduce.h"

thread_local IrrReduceCollector g_irr_reduce_collector = {nullptr, 0, 0};

namespace {
struct ReduceBatch {
  IrrReduceJob job[IRR_REDUCE_BATCH_MAX];
  int njobs;
};

__global__ __launch_bounds__(256) void wgrad_reduce_batch_kernel(const ReduceBatch T) {
  __shared__ float red[3][256];
  const int blk = blockIdx.x;
  int i = 0;
  while (i + 1 < T.njobs && T.job[i + 1].block0 <= blk) ++i;       // (uniform: <= 40 scalar compares)
  irr_reduce_block(T.job[i], blk - T.job[i].block0, red);
}
}  // namespace

extern "C" int irr_wgrad_job_bytes(void) { return (int)sizeof(IrrReduceJob); }
extern "C" int irr_wgrad_reduce_batch_max(void) { return IRR_REDUCE_BATCH_MAX; }

extern "C" int irr_wgrad_defer_begin(void* jobs, int capacity) {
  if (!jobs || capacity <= 0) return IRR_EINVAL;
  g_irr_reduce_collector = {(IrrReduceJob*)jobs, capacity, 0};
  return 0;
}

extern "C" int irr_wgrad_defer_end(void) {
  const int n = g_irr_reduce_collector.count;
  g_irr_reduce_collector = {nullptr, 0, 0};
  return n;
}

extern "C" int irr_wgrad_reduce_batch(const void* jobs, int njobs, void* stream) {
  if (!jobs || njobs <= 0 || njobs > IRR_REDUCE_BATCH_MAX) return IRR_EINVAL;
  ReduceBatch T;
  const IrrReduceJob* src = (const IrrReduceJob*)jobs;
  long blk = 0;
  for (int i = 0; i < njobs; ++i) {
    T.job[i] = src[i];
    if (!src[i].ws || !src[i].gw || src[i].n <= 0 || src[i].P <= 0) return IRR_EINVAL;
    for (int k = 0; k < i; ++k)
      if (src[k].gw == src[i].gw) return IRR_EINVAL;               // two folds into one gradient would race inside a launch
    T.job[i].block0 = (int)blk;
    blk += irr_cdiv(src[i].n, 256);
  }
  T.njobs = njobs;
  hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3((unsigned)blk), dim3(256), 0, (hipStream_t)stream, T);
  IRR_LAUNCH_CHECK();
  return 0;
}
