// Shared helpers for the gfx950 kernels of libirr_hip.so.  wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/irr_hip.h"

#define IRR_LAUNCH_CHECK()                      \
  do {                                          \
    hipError_t e__ = hipGetLastError();         \
    if (e__ != hipSuccess) return (int)e__;     \
  } while (0)

#define IRR_HIP_TRY(expr)                       \
  do {                                          \
    hipError_t e__ = (expr);                    \
    if (e__ != hipSuccess) return (int)e__;     \
  } while (0)

static inline int irr_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// A/B switches read from the environment ONCE per call site (the launchers run thousands of times per step).
#include <stdlib.h>
#define IRR_ENV_FLAG(name)                                   \
  ([]() -> bool {                                            \
    static const bool v = getenv(name) != nullptr;           \
    return v;                                                \
  }())

// LeakyReLU(0.1): max(v, 0.1 v) is the same value for every input incl. -0 and NaN (two VALU instead of compare + select + multiply)
__device__ __forceinline__ float irr_lrelu(float v) { return fmaxf(v, 0.1f * v); }
__device__ __forceinline__ float irr_lrelu_grad(float y) { return y > 0.f ? 1.f : 0.1f; }

// XCD-aware block order (speed only, never correctness): the dispatcher is observed to place workgroup b on XCD b % 8, each XCD
// with its own 4 MiB L2 (MI355X_MICROARCH.md, "Workgroup dispatch").  irr_xcd_order maps the linear workgroup id to its
// position in "XCD-major" order: ids of one XCD get CONSECUTIVE positions, so a kernel that decodes (position -> tile) with
// the operand-sharing index fastest keeps the blocks that read the same data on one XCD's L2.  Bijection on [0, total).
__device__ __forceinline__ unsigned irr_xcd_order(unsigned lin, unsigned total) {
  const unsigned per = total >> 3, rem = total & 7u, r = lin & 7u, q = lin >> 3;
  return r * per + (r < rem ? r : rem) + q;
}
