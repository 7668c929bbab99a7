#include "common.h"
#include "amax.h"
extern "C" int irr_abi_version(void) { return 12; }   // 12: channel maxima out of the dual small-Cout data gradient (irr_conv2d_smallco_dgrad_dual_ch_f32) and the streaming kernel (irr_conv_x3_next_chmax on every fp16x2 launch); 11: one operand scale per channel for the weight gradient's gy-role operand (irr_conv2d_wgrad_h2_ch, irr_amax_channels_f32); 10: Winograd F(2x2,3x3) forward on the fp16x2 arithmetic (irr_conv2d_wino_fwd_h2, experimental); 9: bit masks for the streaming kernel (irr_conv2d_fwd_h2_bits); 8: the legacy Correlation operator at any parameter point (irr_corr_general_*); 7: the fp16x2 pairs of activation-side operands carry a scaled-up low piece (range 2^17 -> 2^29 per element; irr_conv2d_wgrad_h2_robust_side); 6: the streaming 32-channel kernel takes the fp16x2 form too (irr_conv2d_fwd_h2_dual); 5: fp16x2 ("h2") conv entry points + amax slots; 4: loss reductions take a partial-sum scratch (fixed summation order); 3: Adam scalars are doubles

// ---- channel concatenation of up to IRR_CAT_MAX_PARTS tensors in ONE launch (include/irr_hip.h) ----------------------------
// The decoder input of a level is cat([cost volume, projected features, flow, occlusion]) (models/IRR_PWC.py:104-107) and the
// upsampler's is cat([occ, img1, img2 warped, flow, flow warped]) (:166-167): the parts are written straight into their channel
// slices of the consumer's buffer -- one dispatch instead of one strided copy per part; a part without a source is zero fill
// (the padding channels of the 16-channel upsampler input).
namespace {

struct CatArgs {
  const float* src[IRR_CAT_MAX_PARTS];
  long bs[IRR_CAT_MAX_PARTS];
  int cend[IRR_CAT_MAX_PARTS];      // exclusive prefix end of the part's channels in dst
  int n;
};

// AMAX: max |value written| (zero-fill parts included) folded into *amax -- the consumer of the buffer runs on the fp16x2 conv
// kernels and needs the magnitude of its input: a separate pass over the buffer costs as much as this copy (round 5)
template <bool VEC, bool AMAX>
__global__ __launch_bounds__(256) void cat_channels_kernel(float* __restrict__ dst, long dst_bs, const CatArgs a, long hw, float* __restrict__ amax,
                                                           float* __restrict__ chmax) {
  const int c = blockIdx.y, b = blockIdx.z;
  int part = 0;
#pragma unroll
  for (int i = 0; i < IRR_CAT_MAX_PARTS - 1; ++i) part += (i < a.n - 1 && c >= a.cend[i]) ? 1 : 0;
  const int c_local = c - (part ? a.cend[part - 1] : 0);
  const float* s = a.src[part] ? a.src[part] + (long)b * a.bs[part] + (long)c_local * hw : nullptr;
  float* d = dst + (long)b * dst_bs + (long)c * hw;
  const long p0 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  typedef float f4 __attribute__((ext_vector_type(4)));
  float m = 0.f;
  if (p0 < hw) {
    if (VEC) {
      const f4 v = s ? *(const f4*)(s + p0) : f4{0.f, 0.f, 0.f, 0.f};
      *(f4*)(d + p0) = v;
      if (AMAX) m = x3_amax_fold(x3_amax_fold(x3_amax_fold(x3_amax_fold(m, v[0]), v[1]), v[2]), v[3]);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (p0 + i < hw) {
          const float v = s ? s[p0 + i] : 0.f;
          d[p0 + i] = v;
          if (AMAX) m = x3_amax_fold(m, v);
        }
    }
  }
  if (AMAX) {                                               // (every thread of the block arrives)
    // a block copies pixels of ONE destination channel: its maximum also serves chmax[c] (ABI 12: the channel maxima of the assembled
    // input, for the weight gradients that take it as the operand in their gy role) -- one more look-then-atomic per block
    __shared__ float wm[4];
    m = x3_amax_wave(m);
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
      const float r = x3_amax_fold(x3_amax_fold(wm[0], wm[1]), x3_amax_fold(wm[2], wm[3]));
      if (amax) x3_amax_commit(r, amax);
      if (chmax) x3_amax_commit(r, chmax + c);
    }
  }
}

}  // namespace

static int cat_channels_impl(float* dst, long dst_bs, const void* parts, int nparts, int B, long hw, float* amax, void* stream,
                             float* chmax = nullptr) {
  if (!dst || !parts || nparts < 1 || nparts > IRR_CAT_MAX_PARTS || B <= 0 || B > 65535 || hw <= 0) return IRR_EINVAL;
  const IrrCatPart* p = (const IrrCatPart*)parts;
  CatArgs a;
  int c = 0;
  bool vec = (hw & 3) == 0 && (dst_bs & 3) == 0 && ((uintptr_t)dst & 15) == 0;
  for (int i = 0; i < IRR_CAT_MAX_PARTS; ++i) {
    if (i < nparts) {
      if (p[i].channels <= 0) return IRR_EINVAL;
      c += p[i].channels;
      a.src[i] = p[i].src;
      a.bs[i] = p[i].src_bs;
      if (p[i].src) vec = vec && (p[i].src_bs & 3) == 0 && ((uintptr_t)p[i].src & 15) == 0;
    } else {
      a.src[i] = nullptr;
      a.bs[i] = 0;
    }
    a.cend[i] = c;
  }
  a.n = nparts;
  if (c > 65535) return IRR_EINVAL;
  const dim3 grid(irr_cdiv(hw, 1024), c, B);
  hipStream_t st = (hipStream_t)stream;
  if (amax || chmax) {
    if (vec) hipLaunchKernelGGL((cat_channels_kernel<true, true>), grid, dim3(256), 0, st, dst, dst_bs, a, hw, amax, chmax);
    else hipLaunchKernelGGL((cat_channels_kernel<false, true>), grid, dim3(256), 0, st, dst, dst_bs, a, hw, amax, chmax);
  } else {
    if (vec) hipLaunchKernelGGL((cat_channels_kernel<true, false>), grid, dim3(256), 0, st, dst, dst_bs, a, hw, amax, chmax);
    else hipLaunchKernelGGL((cat_channels_kernel<false, false>), grid, dim3(256), 0, st, dst, dst_bs, a, hw, amax, chmax);
  }
  IRR_LAUNCH_CHECK();
  return 0;
}

extern "C" int irr_cat_channels_f32(float* dst, long dst_bs, const void* parts, int nparts, int B, long hw, void* stream) {
  return cat_channels_impl(dst, dst_bs, parts, nparts, B, hw, nullptr, stream);
}

// the same, and *amax = max(*amax, max |value written|) (an amax slot of the fp16x2 conv kernels: non-negative float, starts at 0)
extern "C" int irr_cat_channels_amax_f32(float* dst, long dst_bs, const void* parts, int nparts, int B, long hw, float* amax, void* stream) {
  if (!amax) return IRR_EINVAL;
  return cat_channels_impl(dst, dst_bs, parts, nparts, B, hw, amax, stream);
}

// (ABI 12) the same, and chmax[c] = max(chmax[c], max |dst[:, c]| as written) for every destination channel c (amax nullable here)
extern "C" int irr_cat_channels_amax_ch_f32(float* dst, long dst_bs, const void* parts, int nparts, int B, long hw, float* amax,
                                            float* chmax, void* stream) {
  if (!chmax) return IRR_EINVAL;
  return cat_channels_impl(dst, dst_bs, parts, nparts, B, hw, amax, stream, chmax);
}

// ---- out[b, :] = x[b, :] + y[b, :] for B samples of n plane-dense floats with independent batch strides ------------------------
// The backward passes add a channel-slice VIEW of one tensor to a dense tensor in a few places (the occlusion channel of the
// upsampler's input gradient + the skip gradient, the estimate slot of a DenseNet gradient buffer + the head's own gradient): as
// ATen ops on a (B, n) view with a batch stride these run through the generic strided-iterator kernel at ~0.25 TB/s (0.55 ms for a
// 44 MB full-resolution plane set; profiles/r4_op_census.txt); here they are one coalesced 16-B pass.  out may alias x or y.
namespace {
template <bool VEC>
__global__ __launch_bounds__(256) void add_planes_kernel(float* __restrict__ out, const float* __restrict__ x, const float* __restrict__ y,
                                                        long n, long out_bs, long x_bs, long y_bs) {
  const int b = blockIdx.y;
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  const float* xb = x + (long)b * x_bs;
  const float* yb = y + (long)b * y_bs;
  float* ob = out + (long)b * out_bs;
  typedef float f4 __attribute__((ext_vector_type(4)));
  if (VEC) {
    const f4 u = *(const f4*)(xb + i), v = *(const f4*)(yb + i);
    f4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = u[e] + v[e];
    *(f4*)(ob + i) = r;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (i + e < n) ob[i + e] = xb[i + e] + yb[i + e];
  }
}
}  // namespace

extern "C" int irr_add_planes_f32(float* out, const float* x, const float* y, int B, long n, long out_bs, long x_bs, long y_bs, void* stream) {
  if (!out || !x || !y || B <= 0 || B > 65535 || n <= 0) return IRR_EINVAL;
  const bool vec = (n & 3) == 0 && ((out_bs | x_bs | y_bs) & 3) == 0 && ((((uintptr_t)out) | ((uintptr_t)x) | ((uintptr_t)y)) & 15) == 0;
  const dim3 grid(irr_cdiv(n, 1024), B);
  if (vec) hipLaunchKernelGGL(add_planes_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, out, x, y, n, out_bs, x_bs, y_bs);
  else hipLaunchKernelGGL(add_planes_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, out, x, y, n, out_bs, x_bs, y_bs);
  IRR_LAUNCH_CHECK();
  return 0;
}

// ---- max |x| per CHANNEL (round 6): out[c] = max over (b, pixels) of |x[b, c, :]| for B plane-dense samples -- the scales of the weight
// gradient's gy-role operand (irr_conv2d_wgrad_h2_ch).  A block takes one channel and about 16K floats of it: a chunk of one plane, or
// the whole planes of several samples at the small pyramid levels (one block per (sample, channel) plane of 42 floats was launch-bound:
// 36 000 blocks for a 6 MB tensor); 16-byte loads where a plane allows; the bit-pattern fold of amax.h, one look at the slot + at most
// one atomic max per block.  HBM-bound on the large maps.
namespace {
__global__ __launch_bounds__(256) void amax_channels_kernel(const float* __restrict__ x, int B, long hw, long bs, float* __restrict__ out, long chunk,
                                                            int spb, int vec) {
  const int c = blockIdx.y;
  float m = 0.f;
  if (spb > 1) {                                            // small planes: samples [b0, b0 + spb) of channel c, whole planes
    const int b0 = blockIdx.x * spb, b1 = min(B, b0 + spb);
    const long n = (long)(b1 - b0) * hw;
    for (long i = threadIdx.x; i < n; i += 256) {
      const long bb = i / hw, p = i - bb * hw;
      m = x3_amax_fold(m, x[(b0 + bb) * bs + (long)c * hw + p]);
    }
  } else {
    const int b = blockIdx.z;
    const long p0 = (long)blockIdx.x * chunk, p1 = min(hw, p0 + chunk);
    const float* xp = x + (long)b * bs + (long)c * hw;
    if (vec) {
      typedef float f4 __attribute__((ext_vector_type(4)));
      for (long p = p0 + 4L * threadIdx.x; p < p1; p += 1024) {
        const f4 v = *(const f4*)(xp + p);
        m = x3_amax_fold(x3_amax_fold(x3_amax_fold(x3_amax_fold(m, v[0]), v[1]), v[2]), v[3]);
      }
    } else {
      for (long p = p0 + threadIdx.x; p < p1; p += 256) m = x3_amax_fold(m, xp[p]);
    }
  }
  x3_amax_publish_block256(m, out + c);
}
}  // namespace

// (also called by conv_x3.hip behind a K-split launch that was asked for channel maxima: accumulate = the slots already hold folds)
int irr_amax_channels_launch(const float* x, int B, int C, long hw, long bs, float* out, hipStream_t st, bool zero) {
  if (zero) IRR_HIP_TRY(irr_zero_async(out, (size_t)C * 4, st));
  const int vec = ((hw & 3) == 0 && (bs & 3) == 0 && (((uintptr_t)x) & 15) == 0) ? 1 : 0;
  const long chunk = 16384;                                 // floats per block
  if (hw * 2 <= chunk) {
    const int spb = (int)(chunk / hw);
    hipLaunchKernelGGL(amax_channels_kernel, dim3((unsigned)((B + spb - 1) / spb), (unsigned)C, 1), dim3(256), 0, st, x, B, hw, bs, out, chunk, spb, 0);
  } else {
    const long nchunks = (hw + chunk - 1) / chunk;
    hipLaunchKernelGGL(amax_channels_kernel, dim3((unsigned)nchunks, (unsigned)C, (unsigned)B), dim3(256), 0, st, x, B, hw, bs, out, chunk, 1, vec);
  }
  IRR_LAUNCH_CHECK();
  return 0;
}

extern "C" int irr_amax_channels_f32(const float* x, int B, int C, long hw, long bs, float* out, int accumulate, void* stream) {
  if (!x || !out || B <= 0 || C <= 0 || hw <= 0 || B > 65535 || C > 65535) return IRR_EINVAL;
  return irr_amax_channels_launch(x, B, C, hw, bs, out, (hipStream_t)stream, !accumulate);
}
