// Shared helpers for libflownet2_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/flownet2_hip.h"

namespace fn2 {

// thread-local last error (fn2_last_error)
char* err_buf();
int fail(int code, const char* fmt, ...);

#define FN2_REQUIRE(cond, ...)                                        \
  do {                                                                \
    if (!(cond)) return fn2::fail(FN2_ERR_INVALID_ARGUMENT, __VA_ARGS__); \
  } while (0)

#define FN2_CHECK_LAUNCH(what)                                                        \
  do {                                                                                \
    hipError_t e_ = hipGetLastError();                                                \
    if (e_ != hipSuccess)                                                             \
      return fn2::fail(FN2_ERR_HIP, "%s: launch failed: %s", what, hipGetErrorString(e_)); \
  } while (0)

#define FN2_HIP(call)                                                                  \
  do {                                                                                 \
    hipError_t e_ = (call);                                                            \
    if (e_ != hipSuccess)                                                              \
      return fn2::fail(FN2_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_));    \
  } while (0)

typedef __bf16 bf16_t;

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return (float)v; }
template <typename T>
__device__ __forceinline__ T from_f32(float v);
template <>
__device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <>
__device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }  // RNE, NaN-safe

// LeakyReLU exactly as the reference writes it (utils.py:401-405): f1*x + f2*|x|
__device__ __forceinline__ float leaky(float x) { return 0.55f * x + 0.45f * fabsf(x); }

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace fn2
