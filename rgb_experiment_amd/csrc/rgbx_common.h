// Shared helpers for the librgbx_hip translation units (gfx950 only).
#pragma once
#include <cstring>  // rocPRIM's texture iterator needs host memset declared before hip_runtime
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <initializer_list>

#include "../../include/rgbx_hip.h"

namespace rgbx {

constexpr int kWave = 64;  // CDNA wavefront

char* err_buf();  // thread-local, 256 bytes

inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 256, fmt, ap);
  va_end(ap);
  return code;
}

inline int hip_fail(hipError_t e, const char* what) {
  snprintf(err_buf(), 256, "%s: %s", what, hipGetErrorString(e));
  return (int)e;
}

// Launch errors surface through hipGetLastError; nothing here synchronises.
#define RGBX_CHECK_LAUNCH(what)                               \
  do {                                                        \
    hipError_t e_ = hipGetLastError();                        \
    if (e_ != hipSuccess) return ::rgbx::hip_fail(e_, what);  \
  } while (0)

#define RGBX_HIP(call)                                         \
  do {                                                         \
    hipError_t e_ = (call);                                    \
    if (e_ != hipSuccess) return ::rgbx::hip_fail(e_, #call);  \
  } while (0)

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// VEC-wide (16/8/4-byte) row-fragment loads and stores.
template <int VEC> struct VecT;
template <> struct VecT<4> { using type = float4; };
template <> struct VecT<2> { using type = float2; };
template <> struct VecT<1> { using type = float; };

template <int VEC>
__device__ __forceinline__ void load_vec(float (&v)[VEC], const float* p) {
  using V = typename VecT<VEC>::type;
  const V t = *reinterpret_cast<const V*>(p);
  if constexpr (VEC == 4) { v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
  else if constexpr (VEC == 2) { v[0] = t.x; v[1] = t.y; }
  else { v[0] = t; }
}

template <int VEC>
__device__ __forceinline__ void store_vec(float* p, const float (&v)[VEC]) {
  using V = typename VecT<VEC>::type;
  V t;
  if constexpr (VEC == 4) { t.x = v[0]; t.y = v[1]; t.z = v[2]; t.w = v[3]; }
  else if constexpr (VEC == 2) { t.x = v[0]; t.y = v[1]; }
  else { t = v[0]; }
  *reinterpret_cast<V*>(p) = t;
}

// Memory-bound grid cap: 256 CUs x 8 blocks of 256 threads, x4 so the tail is short.
constexpr int kMaxGrid = 256 * 8 * 4;

}  // namespace rgbx
