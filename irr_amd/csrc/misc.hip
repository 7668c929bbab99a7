#include "common.h"
extern "C" int irr_abi_version(void) { return 7; }   // 7: the fp16x2 pairs of activation-side operands carry a scaled-up low piece (range 2^17 -> 2^29 per element; irr_conv2d_wgrad_h2_robust_side); 6: the streaming 32-channel kernel takes the fp16x2 form too (irr_conv2d_fwd_h2_dual); 5: fp16x2 ("h2") conv entry points + amax slots; 4: loss reductions take a partial-sum scratch (fixed summation order); 3: Adam scalars are doubles

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

template <bool VEC>
__global__ __launch_bounds__(256) void cat_channels_kernel(float* __restrict__ dst, long dst_bs, const CatArgs a, long hw) {
  const int c = blockIdx.y, b = blockIdx.z;
  int part = 0;
#pragma unroll
  for (int i = 0; i < IRR_CAT_MAX_PARTS - 1; ++i) part += (i < a.n - 1 && c >= a.cend[i]) ? 1 : 0;
  const int c_local = c - (part ? a.cend[part - 1] : 0);
  const float* s = a.src[part] ? a.src[part] + (long)b * a.bs[part] + (long)c_local * hw : nullptr;
  float* d = dst + (long)b * dst_bs + (long)c * hw;
  const long p0 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (p0 >= hw) return;
  typedef float f4 __attribute__((ext_vector_type(4)));
  if (VEC) {
    *(f4*)(d + p0) = s ? *(const f4*)(s + p0) : f4{0.f, 0.f, 0.f, 0.f};
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (p0 + i < hw) d[p0 + i] = s ? s[p0 + i] : 0.f;
  }
}

}  // namespace

extern "C" int irr_cat_channels_f32(float* dst, long dst_bs, const void* parts, int nparts, int B, long hw, void* stream) {
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
  if (vec) hipLaunchKernelGGL(cat_channels_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, dst, dst_bs, a, hw);
  else hipLaunchKernelGGL(cat_channels_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, dst, dst_bs, a, hw);
  IRR_LAUNCH_CHECK();
  return 0;
}
