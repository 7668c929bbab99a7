// Round 5, NOTES D.5: stand-alone victim for the wide-store hazard of gfx950.  Every thread stores (1, 1, 1, 1) to its own 16-byte slot
// with `buffer_store_dwordx4 v[10:13], voff, rsrc, <soffset> offen` and then, after WS wait states, overwrites the store's first data
// register (v10) with 2.0 -- the pattern hipcc produced in the epilogue of conv_x3s_kernel (an SGPR soffset, no wait states).  A slot whose
// first component reads 2.0 afterwards was stored from the REWRITTEN register.
//   FORM 0: soffset in an SGPR (the form the ISA manuals exempt from the wait states and hipcc leaves unguarded)
//   FORM 1: soffset = 0 literal (the form behind which hipcc inserts two wait states)
//   WS    : s_nop count between the store and the overwrite (0 = none)
// Built and driven by tools/store_hazard.py.
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int FORM, int WS>
__global__ __launch_bounds__(256) void store_victim(float* out, int iters, long stride_bytes) {
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)out, (short)0, (int)0x7fffffff, 0x00020000);
  const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
  const float good = 1.0f, bad = 2.0f;
  for (int it = 0; it < iters; ++it) {
    const uint32_t voff = tid * 16u;
    const uint32_t soff = (uint32_t)(it * stride_bytes);
    if (FORM == 0) {
      if (WS == 0)
        asm volatile("v_mov_b32 v10, %0\n\tv_mov_b32 v11, %0\n\tv_mov_b32 v12, %0\n\tv_mov_b32 v13, %0\n\ts_nop 4\n\t"
                     "buffer_store_dwordx4 v[10:13], %2, %3, %4 offen\n\t"
                     "v_mov_b32 v10, %1\n\t"
                     :: "v"(good), "v"(bad), "v"(voff), "s"(rsrc), "s"(soff) : "v10", "v11", "v12", "v13", "memory");
      else
        asm volatile("v_mov_b32 v10, %0\n\tv_mov_b32 v11, %0\n\tv_mov_b32 v12, %0\n\tv_mov_b32 v13, %0\n\ts_nop 4\n\t"
                     "buffer_store_dwordx4 v[10:13], %2, %3, %4 offen\n\t"
                     "s_nop %5\n\t"
                     "v_mov_b32 v10, %1\n\t"
                     :: "v"(good), "v"(bad), "v"(voff), "s"(rsrc), "s"(soff), "n"(WS - 1) : "v10", "v11", "v12", "v13", "memory");
    } else {
      const uint32_t vo2 = voff + soff;
      if (WS == 0)
        asm volatile("v_mov_b32 v10, %0\n\tv_mov_b32 v11, %0\n\tv_mov_b32 v12, %0\n\tv_mov_b32 v13, %0\n\ts_nop 4\n\t"
                     "buffer_store_dwordx4 v[10:13], %2, %3, 0 offen\n\t"
                     "v_mov_b32 v10, %1\n\t"
                     :: "v"(good), "v"(bad), "v"(vo2), "s"(rsrc) : "v10", "v11", "v12", "v13", "memory");
      else
        asm volatile("v_mov_b32 v10, %0\n\tv_mov_b32 v11, %0\n\tv_mov_b32 v12, %0\n\tv_mov_b32 v13, %0\n\ts_nop 4\n\t"
                     "buffer_store_dwordx4 v[10:13], %2, %3, 0 offen\n\t"
                     "s_nop %4\n\t"
                     "v_mov_b32 v10, %1\n\t"
                     :: "v"(good), "v"(bad), "v"(vo2), "s"(rsrc), "n"(WS - 1) : "v10", "v11", "v12", "v13", "memory");
    }
  }
}

// aggressor: a kernel that streams reads and writes through the memory pipeline (grid-stride copy with a little arithmetic)
__global__ __launch_bounds__(256) void mem_aggressor(const float4* __restrict__ a, float4* __restrict__ b, long n, int rounds) {
  for (int r = 0; r < rounds; ++r)
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
      float4 v = a[i];
      v.x += 1.f; v.y *= 0.5f;
      b[i] = v;
    }
}

// narrower stores (the hazard rule covers only stores of MORE than 64 bits): dwordx2 / dword with an SGPR soffset, overwrite at once
template <int WIDTH>
__global__ __launch_bounds__(256) void store_victim_narrow(float* out, int iters, long stride_bytes) {
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)out, (short)0, (int)0x7fffffff, 0x00020000);
  const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
  const float good = 1.0f, bad = 2.0f;
  for (int it = 0; it < iters; ++it) {
    const uint32_t voff = tid * 16u;
    const uint32_t soff = (uint32_t)(it * stride_bytes);
    if (WIDTH == 2)
      asm volatile("v_mov_b32 v10, %0\n\tv_mov_b32 v11, %0\n\ts_nop 4\n\t"
                   "buffer_store_dwordx2 v[10:11], %2, %3, %4 offen\n\t"
                   "v_mov_b32 v10, %1\n\t"
                   :: "v"(good), "v"(bad), "v"(voff), "s"(rsrc), "s"(soff) : "v10", "v11", "memory");
    else if (WIDTH == 3)
      asm volatile("v_mov_b32 v10, %0\n\tv_mov_b32 v11, %0\n\tv_mov_b32 v12, %0\n\ts_nop 4\n\t"
                   "buffer_store_dwordx3 v[10:12], %2, %3, %4 offen\n\t"
                   "v_mov_b32 v10, %1\n\t"
                   :: "v"(good), "v"(bad), "v"(voff), "s"(rsrc), "s"(soff) : "v10", "v11", "v12", "memory");
    else
      asm volatile("v_mov_b32 v10, %0\n\ts_nop 4\n\t"
                   "buffer_store_dword v10, %2, %3, %4 offen\n\t"
                   "v_mov_b32 v10, %1\n\t"
                   :: "v"(good), "v"(bad), "v"(voff), "s"(rsrc), "s"(soff) : "v10", "memory");
  }
}
extern "C" int launch_store_victim_narrow(float* out, int nblk, int iters, long stride_bytes, int width, void* stream) {
  if (width == 1) hipLaunchKernelGGL(store_victim_narrow<1>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, out, iters, stride_bytes);
  else if (width == 2) hipLaunchKernelGGL(store_victim_narrow<2>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, out, iters, stride_bytes);
  else if (width == 3) hipLaunchKernelGGL(store_victim_narrow<3>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, out, iters, stride_bytes);
  else return -1;
  return (int)hipGetLastError();
}

// the same question for LDS: ds_write_b128 v[10:13] followed at once by a VALU write of v10 (no hazard is documented)
__global__ __launch_bounds__(256) void lds_victim(float* out, int iters) {
  __shared__ float4 buf[256];
  const float good = 1.0f, bad = 2.0f;
  const uint32_t laddr = (uint32_t)(uintptr_t)(&buf[threadIdx.x]) ;
  float4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
    asm volatile("v_mov_b32 v10, %0\n\tv_mov_b32 v11, %0\n\tv_mov_b32 v12, %0\n\tv_mov_b32 v13, %0\n\ts_nop 4\n\t"
                 "ds_write_b128 %2, v[10:13]\n\t"
                 "v_mov_b32 v10, %1\n\t"
                 "s_waitcnt lgkmcnt(0)\n\t"
                 :: "v"(good), "v"(bad), "v"(laddr) : "v10", "v11", "v12", "v13", "memory");
    const float4 r = buf[threadIdx.x];
    acc.x += (r.x == 2.0f) ? 1.f : 0.f;                      // count slots that hold the rewritten value
    acc.y += (r.y != 1.0f || r.z != 1.0f || r.w != 1.0f || (r.x != 1.0f && r.x != 2.0f)) ? 1.f : 0.f;
    __syncthreads();
  }
  ((float4*)out)[blockIdx.x * 256 + threadIdx.x] = acc;
}
extern "C" int launch_lds_victim(float* out, int nblk, int iters, void* stream) {
  hipLaunchKernelGGL(lds_victim, dim3(nblk), dim3(256), 0, (hipStream_t)stream, out, iters);
  return (int)hipGetLastError();
}

#define GO(F, W) hipLaunchKernelGGL((store_victim<F, W>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, out, iters, stride_bytes)
extern "C" int launch_store_victim(float* out, int nblk, int iters, long stride_bytes, int form, int ws, void* stream) {
  if (form == 0) { switch (ws) { case 0: GO(0, 0); break; case 1: GO(0, 1); break; case 2: GO(0, 2); break; case 4: GO(0, 4); break; case 8: GO(0, 8); break; default: return -1; } }
  else if (form == 1) { switch (ws) { case 0: GO(1, 0); break; case 1: GO(1, 1); break; case 2: GO(1, 2); break; case 4: GO(1, 4); break; case 8: GO(1, 8); break; default: return -1; } }
  else return -1;
  return (int)hipGetLastError();
}
extern "C" int launch_mem_aggressor(const void* a, void* b, long n16, int rounds, int nblk, void* stream) {
  hipLaunchKernelGGL(mem_aggressor, dim3(nblk), dim3(256), 0, (hipStream_t)stream, (const float4*)a, (float4*)b, n16, rounds);
  return (int)hipGetLastError();
}
