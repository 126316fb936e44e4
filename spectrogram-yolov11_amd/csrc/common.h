// common.h — shared device/host helpers for libsy11 (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/sy11.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// ------------------------------------------------------------------------------------------------ errors
void sy11_set_error(const char* fmt, ...);
#define SY11_FAIL(code, ...)      \
  do {                            \
    sy11_set_error(__VA_ARGS__);  \
    return (code);                \
  } while (0)
#define SY11_REQUIRE(cond, ...) \
  do {                          \
    if (!(cond)) SY11_FAIL(SY11_EINVAL, __VA_ARGS__); \
  } while (0)
#define SY11_LAUNCH_CHECK(name)                                                             \
  do {                                                                                      \
    hipError_t e_ = hipGetLastError();                                                      \
    if (e_ != hipSuccess) SY11_FAIL(SY11_ELAUNCH, "%s: launch failed: %s", name, hipGetErrorString(e_)); \
  } while (0)

static inline int dtype_size(int dt) { return dt == SY11_F32 ? 4 : 2; }
static inline bool dtype_ok(int dt) { return dt == SY11_F32 || dt == SY11_F16 || dt == SY11_BF16; }
static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ------------------------------------------------------------------------------------------------ scalar type traits
struct bf16_t { uint16_t v; };

template <typename T> struct ElemTraits;
template <> struct ElemTraits<float> {
  static constexpr int code = SY11_F32;
  static __device__ __forceinline__ float to_f(float x) { return x; }
  static __device__ __forceinline__ float from_f(float x) { return x; }
};
template <> struct ElemTraits<_Float16> {
  static constexpr int code = SY11_F16;
  static __device__ __forceinline__ float to_f(_Float16 x) { return (float)x; }
  static __device__ __forceinline__ _Float16 from_f(float x) { return (_Float16)x; }
};
template <> struct ElemTraits<__bf16> {
  static constexpr int code = SY11_BF16;
  static __device__ __forceinline__ float to_f(__bf16 x) { return (float)x; }
  static __device__ __forceinline__ __bf16 from_f(float x) { return (__bf16)x; }
};

__device__ __forceinline__ float silu_f(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }   // 1-ulp rcp
// d/dv silu(v) = s + v*s*(1-s),  s = sigmoid(v)
__device__ __forceinline__ float dsilu_f(float v) {
  float s = __builtin_amdgcn_rcpf(1.0f + __expf(-v));
  return s * (1.0f + v * (1.0f - s));
}

// dispatch a templated launcher on the runtime dtype code
#define SY11_DISPATCH_DTYPE(dt, T, ...)                         \
  do {                                                          \
    if ((dt) == SY11_F32) { typedef float T; __VA_ARGS__; }     \
    else if ((dt) == SY11_F16) { typedef _Float16 T; __VA_ARGS__; } \
    else { typedef __bf16 T; __VA_ARGS__; }                     \
  } while (0)
