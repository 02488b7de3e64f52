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
// scalar store of one output element (x2: hi/lo halves of its group; the pointer is element-addressed)
template <typename T>
__device__ __forceinline__ void store_elem(T* p, float v) { *p = from_f32<T>(v); }

// Three consecutive floats (an RGB pixel of an NHWC image) as ONE 12-byte access: the 4-byte-aligned packed struct
// compiles to global_load/store_dwordx3.  The 3-channel passes are bound by vector-memory instruction issue, and this
// form needs a third of the instructions of per-channel accesses (flow_warp at batch 64: 0.271 -> 0.182 ms).
struct __attribute__((packed, aligned(4))) rgb3_t { float r, g, b; };
__device__ __forceinline__ rgb3_t load_rgb(const float* p) { return *reinterpret_cast<const rgb3_t*>(p); }
__device__ __forceinline__ void store_rgb(float* p, float r, float g, float b) {
  *reinterpret_cast<rgb3_t*>(p) = rgb3_t{r, g, b};
}

// Split-fp16 storage element (FN2_F16X2): 4 bytes per logical channel; a group of 8 channels is 8 fp16 hi
// parts (16 B) followed by 8 fp16 lo parts (16 B).  sizeof == 4 so that all address arithmetic is that of fp32.
struct x2_t { unsigned raw; };
template <typename T> struct is_x2 { static constexpr bool value = false; };
template <> struct is_x2<x2_t> { static constexpr bool value = true; };

// 8 channels -> (hi, lo) fp16 vectors
__device__ __forceinline__ void split8(const float* v, uint4& hi, uint4& lo) {
  typedef __attribute__((ext_vector_type(8))) _Float16 h8;
  h8 h, l;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    h[j] = (_Float16)v[j];
    l[j] = (_Float16)(v[j] - (float)h[j]);
  }
  hi = __builtin_bit_cast(uint4, h);
  lo = __builtin_bit_cast(uint4, l);
}
// One two-channel pixel of tf.image.resize_bilinear(align_corners=True) times `scale` from its four neighbours.  Shared by
// the resize kernel (ops.hip) and the ops that interpolate their own flow vector (elem.hip: flow_at), so that
// both produce the same bits whatever the surrounding code is.
__device__ __forceinline__ float2 bilerp_c2(const float2 tl, const float2 tr, const float2 bl, const float2 br, float lx,
                                            float ly, float scale) {
  // every multiply-add is an EXPLICIT fma: nothing is left for the compiler to contract one way here and another way there
  const float top0 = __builtin_fmaf(tr.x - tl.x, lx, tl.x), bot0 = __builtin_fmaf(br.x - bl.x, lx, bl.x);
  const float top1 = __builtin_fmaf(tr.y - tl.y, lx, tl.y), bot1 = __builtin_fmaf(br.y - bl.y, lx, bl.y);
  return make_float2(__builtin_fmaf(bot0 - top0, ly, top0) * scale, __builtin_fmaf(bot1 - top1, ly, top1) * scale);
}

__device__ __forceinline__ void join8(const uint4& hi, const uint4& lo, float* v) {
  typedef __attribute__((ext_vector_type(8))) _Float16 h8;
  const h8 h = __builtin_bit_cast(h8, hi), l = __builtin_bit_cast(h8, lo);
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (float)h[j] + (float)l[j];
}

template <>
__device__ __forceinline__ void store_elem<x2_t>(x2_t* p, float v) {
  const size_t addr = reinterpret_cast<size_t>(p);  // element-addressed: 4 bytes per logical channel
  _Float16* g = reinterpret_cast<_Float16*>(addr & ~size_t(31));
  const int j = (int)((addr & 31) >> 2);
  const _Float16 h = (_Float16)v;
  g[j] = h;
  g[8 + j] = (_Float16)(v - (float)h);
}

// scalar load of one element as fp32 (x2: hi + lo of its group; element-addressed pointer)
template <typename T>
__device__ __forceinline__ float load_elem(const T* p) { return (float)*p; }
template <>
__device__ __forceinline__ float load_elem<x2_t>(const x2_t* p) {
  const size_t addr = reinterpret_cast<size_t>(p);
  const _Float16* g = reinterpret_cast<const _Float16*>(addr & ~size_t(31));
  const int j = (int)((addr & 31) >> 2);
  return (float)g[j] + (float)g[8 + j];
}

// 4 consecutive channels starting at a multiple of 4, as fp32
template <typename T>
__device__ __forceinline__ void load4(const T* p, float* v) {
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = (float)p[j];
}
template <>
__device__ __forceinline__ void load4<x2_t>(const x2_t* p, float* v) {
  typedef __attribute__((ext_vector_type(4))) _Float16 h4;
  const size_t addr = reinterpret_cast<size_t>(p);
  const _Float16* g = reinterpret_cast<const _Float16*>(addr & ~size_t(31));
  const int j0 = (int)((addr & 31) >> 2);
  const h4 h = *reinterpret_cast<const h4*>(g + j0), l = *reinterpret_cast<const h4*>(g + 8 + j0);
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = (float)h[j] + (float)l[j];
}

// N consecutive elements converted from fp32 and written with one (N*sizeof(T))-byte store
template <typename T, int N>
__device__ __forceinline__ void store_vec(T* p, const float* v) {
  typedef T vt __attribute__((ext_vector_type(N)));
  vt t;
#pragma unroll
  for (int j = 0; j < N; ++j) t[j] = (T)v[j];
  *reinterpret_cast<vt*>(p) = t;
}
// 4 consecutive channels starting at a multiple of 4 (split fp16: two 8-byte stores, hi and lo halves)
template <>
__device__ __forceinline__ void store_vec<x2_t, 4>(x2_t* p, const float* v) {
  typedef __attribute__((ext_vector_type(4))) _Float16 h4;
  h4 h, l;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    h[j] = (_Float16)v[j];
    l[j] = (_Float16)(v[j] - (float)h[j]);
  }
  // p points at logical channel c (c % 4 == 0); the group starts at p - (c & 7) channels = 4 bytes each
  const size_t addr = reinterpret_cast<size_t>(p);
  _Float16* g = reinterpret_cast<_Float16*>(addr & ~size_t(31));  // group base (32-byte aligned)
  const int j0 = (int)((addr & 31) >> 2);                         // 0 or 4
  *reinterpret_cast<h4*>(g + j0) = h;
  *reinterpret_cast<h4*>(g + 8 + j0) = l;
}
// 16 consecutive elements as 16-byte stores
template <typename T>
__device__ __forceinline__ void store16(T* p, const float* v) {
  if constexpr (is_x2<T>::value) {  // two groups of 8: hi, lo, hi, lo
    uint4* q = reinterpret_cast<uint4*>(p);
    split8(v, q[0], q[1]);
    split8(v + 8, q[2], q[3]);
  } else {
    constexpr int EPC = 16 / (int)sizeof(T);
#pragma unroll
    for (int q = 0; q < 16 / EPC; ++q) store_vec<T, EPC>(p + q * EPC, v + q * EPC);
  }
}

// 16 consecutive elements as 16-byte loads (fp32 and split fp16: the types gradient buffers have)
template <typename T>
__device__ __forceinline__ void load16(const T* p, float* v) {
  static_assert(sizeof(T) == 4, "load16: fp32 / split-fp16 tensors");
  const uint4* q = reinterpret_cast<const uint4*>(p);
  const uint4 a = q[0], b = q[1], c = q[2], d = q[3];
  if constexpr (is_x2<T>::value) {
    join8(a, b, v);
    join8(c, d, v + 8);
  } else {
    const uint4 r[4] = {a, b, c, d};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[4 * i] = __uint_as_float(r[i].x); v[4 * i + 1] = __uint_as_float(r[i].y);
      v[4 * i + 2] = __uint_as_float(r[i].z); v[4 * i + 3] = __uint_as_float(r[i].w);
    }
  }
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

template <>
__device__ __forceinline__ f32x4 mfma_16x16x32<x2_t>(const uint4&, const uint4&, const f32x4& c) { return c; }
template <>
__device__ __forceinline__ f32x16 mfma_32x32x16<x2_t>(const uint4&, const uint4&, const f32x16& c) { return c; }

static inline bool is_16bit(int dtype) { return dtype == FN2_BF16 || dtype == FN2_F16; }
static inline int dtype_size(int dtype) { return (dtype == FN2_F32 || dtype == FN2_F16X2) ? 4 : 2; }


// LeakyReLU exactly as the reference writes it (utils.py:401-405): f1*x + f2*|x|
__device__ __forceinline__ float leaky(float x) { return 0.55f * x + 0.45f * fabsf(x); }

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace fn2
