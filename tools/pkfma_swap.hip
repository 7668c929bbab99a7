// Round 4, NOTES C.3: a stand-alone victim that replays the instruction sequence at which conv_smallco_dgrad4_kernel (SLP build)
// deviates beside the dilation-16 weight gradient.  What tools/pair_diff.py established about the real kernel: every wrong output is
// the fourth pixel of the fourth channel of a loop trip, lanes 48..63 of a wave (the last quarter pass), and is bit-equal to the FMA
// chain whose accumulator reads as ZERO at the ninth product -- the one v_pk_fma_f32 whose accumulator halves are swapped
// (op_sel:[0,0,1] op_sel_hi:[0,1,0]), issued right after an s_load_dword and right before s_waitcnt lgkmcnt(0).
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/pkfma_swap.hip -o tools/_pkfma_swap.so     (driven by tools/pkfma_swap.py)
// Forms: see the SEQ(...) lines of swap_victim.
#include <hip/hip_runtime.h>
#include <cstdint>

typedef float f2 __attribute__((ext_vector_type(2)));

#define PK_HEAD "v_pk_fma_f32 %[a], %[p0], %[x], %[a] op_sel_hi:[0,1,1]\n" "v_pk_fma_f32 %[b], %[p0], %[y], %[b] op_sel_hi:[0,1,1]\n"
#define PK_A "v_pk_fma_f32 %[a], %[p1], %[y], %[a] op_sel_hi:[0,1,1]\n"
#define PK_SWAP "v_pk_fma_f32 %[b], %[p1], %[x], %[b] op_sel:[0,0,1] op_sel_hi:[0,1,0]\n"
#define PK_PLAIN "v_pk_fma_f32 %[b], %[p1], %[x], %[b] op_sel_hi:[0,1,1]\n"
#define PK_SWAP1 "v_pk_fma_f32 %[b], %[p1], %[x], %[b] op_sel:[0,1,0] op_sel_hi:[0,0,1]\n"
#define SLOAD "s_load_dword %[t], %[wp], 0x20\n"
#define SWAIT "s_waitcnt lgkmcnt(0)\n"
#define SEQ(body) asm volatile(body : [a] "+v"(a), [b] "+v"(b), [t] "=&s"(t), [d] "=&v"(dmy) : [p0] "s"(p0), [p1] "s"(p1), [x] "v"(x), [y] "v"(y), [wp] "s"(wc) : "memory")

template <int FORM>
__global__ __launch_bounds__(256) void swap_victim(const float* __restrict__ in, const float* __restrict__ w, float* __restrict__ out, int iters) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  f2 x = {in[4 * i], in[4 * i + 1]}, y = {in[4 * i + 2], in[4 * i + 3]};
  f2 a = {0.f, 0.f}, b = {0.f, 0.f};
  float tsum = 0.f;
  for (int k = 0; k < iters; ++k) {
    const float* wc = w + (k & 31) * 12;                                  // wave-uniform
    const uint32_t w0 = __builtin_amdgcn_readfirstlane(__float_as_uint(wc[0])), w1 = __builtin_amdgcn_readfirstlane(__float_as_uint(wc[1]));
    const uint64_t p0 = ((uint64_t)0x00007f5au << 32) | w0, p1 = ((uint64_t)0x00007f5au << 32) | w1;   // (high halves: junk, as in the kernel)
    uint32_t t = 0, dmy;
    if (FORM == 0) SEQ(PK_HEAD SLOAD PK_A PK_SWAP SWAIT);                                  // as in the kernel
    else if (FORM == 1) SEQ(PK_HEAD SLOAD PK_A PK_PLAIN SWAIT);                            // no accumulator swap
    else if (FORM == 2) SEQ(PK_HEAD "s_mov_b32 %[t], 0\n" PK_A PK_SWAP);                   // swap, no scalar load / wait
    else if (FORM == 3) SEQ(PK_HEAD SLOAD PK_A PK_SWAP "s_nop 7\n" SWAIT);                 // 8 idle cycles before the wait
    else if (FORM == 4) SEQ(SLOAD PK_HEAD PK_A PK_SWAP SWAIT);                             // scalar load issued earlier
    else if (FORM == 5) SEQ(PK_HEAD SLOAD PK_A PK_SWAP1 SWAIT);                            // halves of the VECTOR MULTIPLICAND swapped instead
    else if (FORM == 6) SEQ(PK_HEAD "s_mov_b32 %[t], 0\n" PK_A PK_SWAP SWAIT);             // swap + wait with nothing outstanding
    else if (FORM == 7) SEQ(PK_HEAD SLOAD PK_A PK_SWAP "v_mov_b32 %[d], 0\n" SWAIT);       // one independent VALU instruction before the wait
    else if (FORM == 8) SEQ(PK_HEAD SLOAD PK_SWAP PK_A SWAIT);                             // swap is the SECOND to last VALU instruction
    else if (FORM == 9) SEQ(PK_HEAD SLOAD PK_A PK_SWAP "s_nop 0\n" SWAIT);                 // one idle cycle before the wait
    else if (FORM == 10) SEQ(PK_HEAD PK_A "v_pk_add_f32 %[b], %[x], %[b] op_sel:[0,1] op_sel_hi:[1,0]\n");              // packed ADD, addend halves swapped
    else if (FORM == 11) SEQ(PK_HEAD PK_A "v_pk_mul_f32 %[b], %[x], %[b] op_sel:[0,1] op_sel_hi:[1,0]\n");              // packed MUL, second factor swapped
    else if (FORM == 12) SEQ(PK_HEAD PK_A "v_pk_fma_f32 %[b], %[p1], %[x], %[b] op_sel:[0,0,1] op_sel_hi:[0,1,1]\n");   // accumulator: high half broadcast
    else if (FORM == 13) SEQ(PK_HEAD PK_A "v_pk_fma_f32 %[b], %[y], %[x], %[b] op_sel:[0,0,1] op_sel_hi:[1,1,0]\n");    // swap, all three sources in VGPRs
    else if (FORM == 14) SEQ(PK_HEAD PK_A "v_pk_fma_f32 %[b], %[p1], %[x], %[b] op_sel:[0,0,0] op_sel_hi:[0,1,0]\n");   // accumulator: low half broadcast
    // Round 6: is it a missing WAIT STATE, like the wide-store finding of round 5?  n wait states between the last
    // writer of the destination pair (the second packed FMA of PK_HEAD, one instruction -- PK_A -- earlier) and the swapped read.
    else if (FORM == 15) SEQ(PK_HEAD PK_A "s_nop 0\n" PK_SWAP);                            // 1 wait state before the swap
    else if (FORM == 16) SEQ(PK_HEAD PK_A "s_nop 1\n" PK_SWAP);                            // 2
    else if (FORM == 17) SEQ(PK_HEAD PK_A "s_nop 3\n" PK_SWAP);                            // 4
    else if (FORM == 18) SEQ(PK_HEAD PK_A "s_nop 7\n" PK_SWAP);                            // 8
    else if (FORM == 19) SEQ(PK_HEAD "s_nop 3\n" PK_A PK_SWAP);                            // 4 wait states right behind the writer
    else if (FORM == 20) SEQ("v_pk_fma_f32 %[a], %[p0], %[x], %[a] op_sel_hi:[0,1,1]\n" PK_A "v_pk_fma_f32 %[b], %[p0], %[y], %[b] op_sel_hi:[0,1,1]\n" PK_SWAP);   // writer DIRECTLY before the swap
    else if (FORM == 21) SEQ("v_pk_fma_f32 %[a], %[p0], %[x], %[a] op_sel_hi:[0,1,1]\n" PK_A "v_pk_fma_f32 %[b], %[p0], %[y], %[b] op_sel_hi:[0,1,1]\n" "s_nop 3\n" PK_SWAP);   // ... with 4 wait states
    else if (FORM == 22) SEQ(PK_HEAD PK_A "s_nop 7\n" "s_nop 7\n" PK_SWAP);               // 16
    else if (FORM == 23) SEQ(PK_HEAD PK_A "v_mov_b32 %[d], 0\n" "v_mov_b32 %[d], 0\n" "v_mov_b32 %[d], 0\n" "v_mov_b32 %[d], 0\n" PK_SWAP);   // four independent VALU instructions instead of wait states
    tsum += __uint_as_float(t);
    a = a * 0.5f;
    b = b * 0.5f;
  }
  out[4 * i] = a[0]; out[4 * i + 1] = a[1]; out[4 * i + 2] = b[0]; out[4 * i + 3] = b[1] + tsum * 1e-30f;
}

typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
// synthetic aggressor: four-wave blocks (96 KiB of LDS: one block per CU) streaming v_mfma_f32_32x32x16_f16
__global__ __launch_bounds__(256) void mfma_aggressor(float* __restrict__ sink, int iters) {
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

// the same block shape without matrix instructions: dependent scalar FMAs
__global__ __launch_bounds__(256) void valu_aggressor(float* __restrict__ sink, int iters) {
  extern __shared__ unsigned int lds[];
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  float acc[8];
  for (int t = 0; t < 8; ++t) acc[t] = 0.001f * (float)(threadIdx.x + t);
  for (int k = 0; k < iters * 8; ++k) {
#pragma unroll
    for (int t = 0; t < 8; ++t) { acc[t] = __builtin_fmaf(acc[t], 0.999f, 0.001f); asm volatile("" : "+v"(acc[t])); }
  }
  float s = 0.f;
  for (int t = 0; t < 8; ++t) s += acc[t];
  if (s == 12345.f) sink[blockIdx.x] = s + (float)lds[(threadIdx.x * 7) & 255];
}

extern "C" int launch_valu_aggressor(float* sink, int nblk, int iters, void* stream) {
  (void)hipFuncSetAttribute((const void*)valu_aggressor, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  hipLaunchKernelGGL(valu_aggressor, dim3(nblk), dim3(256), 96 * 1024, (hipStream_t)stream, sink, iters);
  return (int)hipGetLastError();
}

extern "C" int launch_mfma_aggressor(float* sink, int nblk, int iters, void* stream) {
  (void)hipFuncSetAttribute((const void*)mfma_aggressor, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  hipLaunchKernelGGL(mfma_aggressor, dim3(nblk), dim3(256), 96 * 1024, (hipStream_t)stream, sink, iters);
  return (int)hipGetLastError();
}

extern "C" int launch_swap_victim(const float* in, const float* w, float* out, int nblk, int iters, int form, void* stream) {
  hipStream_t st = (hipStream_t)stream;
#define CASE(F) case F: hipLaunchKernelGGL(swap_victim<F>, dim3(nblk), dim3(256), 0, st, in, w, out, iters); break;
  switch (form) { CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13) CASE(14) CASE(15) CASE(16) CASE(17) CASE(18) CASE(19) CASE(20) CASE(21) CASE(22) CASE(23) default: return -1; }
  return (int)hipGetLastError();
}
