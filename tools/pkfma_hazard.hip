// Stand-alone check of round 4's finding (profiles/NOTES.md C.3): does a kernel of packed fp32 FMAs (v_pk_fma_f32) return the same
// bits when waves of ANOTHER kernel that streams MFMAs share its SIMDs?
//   hipcc --offload-arch=gfx950 -O3 tools/pkfma_hazard.hip -o /tmp/pkfma && /tmp/pkfma
// Victim: 256-thread blocks, few registers, a dependent chain of FMAs on data it loaded (packed: float2 arithmetic; scalar: the same
// arithmetic component by component behind an optimisation barrier).  Aggressor: FOUR-wave blocks (one wave per SIMD, 96 KiB of LDS so
// that one block owns a CU) streaming v_mfma_f32_32x32x16_f16 -- the block shape of the dilation-16 weight gradient.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// second victim: the operand form of the kernel the finding comes from -- packed FMAs whose multiplier is a WAVE-UNIFORM value from
// a scalar load, broadcast to both halves (v_pk_fma_f32 v[a:b], s[c:d], v[e:f], v[a:b] op_sel_hi:[0,1,1])
template <bool PACKED>
__global__ __launch_bounds__(256) void victim_sgpr(const float* __restrict__ in, const float* __restrict__ w, float* __restrict__ out, int iters) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  f2 p0 = {in[4 * i], in[4 * i + 1]}, p1 = {in[4 * i + 2], in[4 * i + 3]}, p2 = p0 + p1, p3 = p0 - p1;
  f2 acc0 = {0.f, 0.f}, acc1 = acc0;
  for (int k = 0; k < iters; ++k) {
    const float* wc = w + (k & 31) * 9;                     // wave-uniform: scalar loads
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const float ww = wc[t];
      if (PACKED) {
        const f2 wv = {ww, ww};
        acc0 = wv * (t & 1 ? p0 : p2) + acc0;
        acc1 = wv * (t & 1 ? p1 : p3) + acc1;
      } else {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          float t0 = __builtin_fmaf(ww, (t & 1 ? p0 : p2)[e], acc0[e]);
          asm volatile("" : "+v"(t0));
          float t1 = __builtin_fmaf(ww, (t & 1 ? p1 : p3)[e], acc1[e]);
          asm volatile("" : "+v"(t1));
          acc0[e] = t0; acc1[e] = t1;
        }
      }
    }
    acc0 = acc0 * 0.5f; acc1 = acc1 * 0.5f;
  }
  out[4 * i] = acc0[0]; out[4 * i + 1] = acc0[1]; out[4 * i + 2] = acc1[0]; out[4 * i + 3] = acc1[1];
}

template <bool PACKED>
__global__ __launch_bounds__(256) void victim(const float* __restrict__ in, float* __restrict__ out, int iters) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  f2 a = {in[4 * i], in[4 * i + 1]}, b = {in[4 * i + 2], in[4 * i + 3]};
  f2 acc0 = a, acc1 = b, acc2 = a + b, acc3 = a - b;
  for (int k = 0; k < iters; ++k) {
    if (PACKED) {
      acc0 = acc0 * a + b;
      acc1 = acc1 * b + a;
      acc2 = acc2 * a + acc0;
      acc3 = acc3 * b + acc1;
    } else {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        float t0 = __builtin_fmaf(acc0[e], a[e], b[e]);
        asm volatile("" : "+v"(t0));
        float t1 = __builtin_fmaf(acc1[e], b[e], a[e]);
        asm volatile("" : "+v"(t1));
        float t2 = __builtin_fmaf(acc2[e], a[e], t0);
        asm volatile("" : "+v"(t2));
        float t3 = __builtin_fmaf(acc3[e], b[e], t1);
        asm volatile("" : "+v"(t3));
        acc0[e] = t0; acc1[e] = t1; acc2[e] = t2; acc3[e] = t3;
      }
    }
    // keep the values bounded
    acc0 = acc0 * 0.5f; acc1 = acc1 * 0.5f; acc2 = acc2 * 0.25f; acc3 = acc3 * 0.25f;
  }
  out[4 * i] = acc0[0] + acc1[0];
  out[4 * i + 1] = acc0[1] + acc1[1];
  out[4 * i + 2] = acc2[0] + acc3[0];
  out[4 * i + 3] = acc2[1] + acc3[1];
}

__global__ __launch_bounds__(256) void aggressor(float* __restrict__ sink, int iters) {
  extern __shared__ unsigned int lds[];
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  h8 av, bv;
  for (int e = 0; e < 8; ++e) { av[e] = (_Float16)(0.001f * (float)((threadIdx.x + e) & 15)); bv[e] = (_Float16)(0.002f * (float)((threadIdx.x * 3 + e) & 7)); }
  f16v acc[6];
  for (int t = 0; t < 6; ++t)
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  for (int k = 0; k < iters; ++k) {
#pragma unroll
    for (int t = 0; t < 6; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc[t], 0, 0, 0);
  }
  float s = 0.f;
  for (int t = 0; t < 6; ++t)
    for (int r = 0; r < 16; ++r) s += acc[t][r];
  if (s == 12345.f) sink[blockIdx.x] = s + (float)lds[(threadIdx.x * 7) & 255];
}

int main() {
  const int nblk = 4096, n = nblk * 256 * 4, iters = 400;
  std::vector<float> h(n);
  for (int i = 0; i < n; ++i) h[i] = 0.5f + 0.4f * (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f;
  float *in, *out, *sink;
  hipMalloc(&in, n * 4); hipMalloc(&out, n * 4); hipMalloc(&sink, 4096 * 4);
  hipMemcpy(in, h.data(), n * 4, hipMemcpyHostToDevice);
  hipStream_t s1, s2;
  hipStreamCreate(&s1); hipStreamCreate(&s2);
  hipFuncSetAttribute((const void*)aggressor, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  std::vector<float> ref(n), cur(n);
  float* w;
  hipMalloc(&w, 32 * 9 * 4);
  std::vector<float> hw(32 * 9);
  for (int i = 0; i < 32 * 9; ++i) hw[i] = 0.1f + 0.01f * (float)(i % 17);
  hipMemcpy(w, hw.data(), 32 * 9 * 4, hipMemcpyHostToDevice);
  for (int form = 0; form < 2; ++form)
  for (int packed = 1; packed >= 0; --packed) {
    auto launch_victim = [&]() {
      if (form == 0) {
        if (packed) hipLaunchKernelGGL(victim<true>, dim3(nblk), dim3(256), 0, s1, in, out, iters);
        else hipLaunchKernelGGL(victim<false>, dim3(nblk), dim3(256), 0, s1, in, out, iters);
      } else {
        if (packed) hipLaunchKernelGGL(victim_sgpr<true>, dim3(nblk), dim3(256), 0, s1, in, w, out, iters);
        else hipLaunchKernelGGL(victim_sgpr<false>, dim3(nblk), dim3(256), 0, s1, in, w, out, iters);
      }
    };
    launch_victim();
    hipDeviceSynchronize();
    hipMemcpy(ref.data(), out, n * 4, hipMemcpyDeviceToHost);
    for (int beside = 0; beside <= 1; ++beside) {
      int bad = 0;
      double worst = 0;
      for (int rep = 0; rep < 30; ++rep) {
        if (beside)
          for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(aggressor, dim3(256), dim3(256), 96 * 1024, s2, sink, 4000);
        launch_victim();
        hipDeviceSynchronize();
        hipMemcpy(cur.data(), out, n * 4, hipMemcpyDeviceToHost);
        if (memcmp(cur.data(), ref.data(), n * 4)) {
          ++bad;
          for (int i = 0; i < n; ++i) {
            const double d = fabs((double)cur[i] - ref[i]) / (fabs((double)ref[i]) + 1e-30);
            if (d > worst) worst = d;
          }
        }
      }
      printf("victim (%s operands) with %s FMAs, %s: %d of 30 runs differ from the lone launch (worst relative difference %.1e)\n",
             form ? "SGPR multiplier" : "VGPR", packed ? "PACKED (v_pk_fma_f32)" : "scalar (v_fma_f32)", beside ? "beside the MFMA kernel" : "alone", bad, worst);
    }
  }
  return 0;
}
