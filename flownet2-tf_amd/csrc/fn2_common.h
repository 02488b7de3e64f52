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
typedef _Float16 f16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return (float)v; }
__device__ __forceinline__ float to_f32(f16_t v) { return (float)v; }
template <typename T>
__device__ __forceinline__ T from_f32(float v);
template <>
__device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <>
__device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }  // RNE, NaN-safe
template <>
__device__ __forceinline__ f16_t from_f32<f16_t>(float v) { return (f16_t)v; }

// N consecutive elements converted from fp32 and written with one (N*sizeof(T))-byte store
template <typename T, int N>
__device__ __forceinline__ void store_vec(T* p, const float* v) {
  typedef T vt __attribute__((ext_vector_type(N)));
  vt t;
#pragma unroll
  for (int j = 0; j < N; ++j) t[j] = (T)v[j];
  *reinterpret_cast<vt*>(p) = t;
}
// 16 consecutive elements as 16-byte stores
template <typename T>
__device__ __forceinline__ void store16(T* p, const float* v) {
  constexpr int EPC = 16 / (int)sizeof(T);
#pragma unroll
  for (int q = 0; q < 16 / EPC; ++q) store_vec<T, EPC>(p + q * EPC, v + q * EPC);
}

// 16-bit matrix-core products on raw 16-byte operand chunks (8 elements of T)
template <typename T>
__device__ __forceinline__ f32x4 mfma_16x16x32(const uint4& a, const uint4& b, const f32x4& c);
template <>
__device__ __forceinline__ f32x4 mfma_16x16x32<bf16_t>(const uint4& a, const uint4& b, const f32x4& c) {
  typedef __attribute__((ext_vector_type(8))) __bf16 v8;
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8, a), __builtin_bit_cast(v8, b), c, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x4 mfma_16x16x32<f16_t>(const uint4& a, const uint4& b, const f32x4& c) {
  typedef __attribute__((ext_vector_type(8))) _Float16 v8;
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v8, a), __builtin_bit_cast(v8, b), c, 0, 0, 0);
}
template <typename T>
__device__ __forceinline__ f32x16 mfma_32x32x16(const uint4& a, const uint4& b, const f32x16& c);
template <>
__device__ __forceinline__ f32x16 mfma_32x32x16<bf16_t>(const uint4& a, const uint4& b, const f32x16& c) {
  typedef __attribute__((ext_vector_type(8))) __bf16 v8;
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8, a), __builtin_bit_cast(v8, b), c, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x16 mfma_32x32x16<f16_t>(const uint4& a, const uint4& b, const f32x16& c) {
  typedef __attribute__((ext_vector_type(8))) _Float16 v8;
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8, a), __builtin_bit_cast(v8, b), c, 0, 0, 0);
}
// float specialisations exist only so that `if constexpr (sizeof(T) == 2)` branches parse
template <>
__device__ __forceinline__ f32x4 mfma_16x16x32<float>(const uint4&, const uint4&, const f32x4& c) { return c; }
template <>
__device__ __forceinline__ f32x16 mfma_32x32x16<float>(const uint4&, const uint4&, const f32x16& c) { return c; }

static inline bool is_16bit(int dtype) { return dtype == FN2_BF16 || dtype == FN2_F16; }
static inline int dtype_size(int dtype) { return dtype == FN2_F32 ? 4 : 2; }


// LeakyReLU exactly as the reference writes it (utils.py:401-405): f1*x + f2*|x|
__device__ __forceinline__ float leaky(float x) { return 0.55f * x + 0.45f * fabsf(x); }

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace fn2
