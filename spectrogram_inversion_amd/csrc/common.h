// Shared device/host helpers of libspecinv (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#include "../../include/specinv.h"

namespace specinv {

// ---------------------------------------------------------------------------------------
// error plumbing (nothing throws across the C ABI)
// ---------------------------------------------------------------------------------------
std::string& last_error();
int fail(int code, const char* fmt, ...);

#define SI_HIP(expr)                                                                         \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess)                                                                    \
      return ::specinv::fail(SPECINV_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                             __FILE__, __LINE__);                                            \
  } while (0)

// Wait for a stream whose remaining work is a few microseconds (a scalar read-back): poll instead of sleeping on the
// completion interrupt, which costs ~20 us of wake-up latency per call on this stack; falls back to the blocking wait.
inline hipError_t si_stream_wait_short(hipStream_t stream) {
  for (int spin = 0; spin < 200000; ++spin) {
    const hipError_t q = hipStreamQuery(stream);
    if (q == hipSuccess) return hipSuccess;
    if (q != hipErrorNotReady) return q;
  }
  return hipStreamSynchronize(stream);
}

// Device-memory accounting: an allocation made inside an ABI call is charged to the plan that call entered
// (specinv_plan_device_bytes; the host layer caps its plan cache by bytes).
inline int64_t*& bytes_sink() {
  static thread_local int64_t* sink = nullptr;
  return sink;
}
inline void account_bytes(int64_t delta) {
  if (bytes_sink()) *bytes_sink() += delta;
}

#define SI_CHECK(cond, code, ...)                       \
  do {                                                  \
    if (!(cond)) return ::specinv::fail(code, __VA_ARGS__); \
  } while (0)

#define SI_TRY(expr)          \
  do {                        \
    int _r = (expr);          \
    if (_r != SPECINV_OK) return _r; \
  } while (0)

// ---------------------------------------------------------------------------------------
// complex value type (interleaved re, im - the layout torch uses for complex64/128)
// ---------------------------------------------------------------------------------------
template <typename T>
struct alignas(2 * sizeof(T)) cplx {
  T x, y;
};

template <typename T>
__host__ __device__ inline cplx<T> mk(T a, T b) {
  cplx<T> r;
  r.x = a;
  r.y = b;
  return r;
}
template <typename T>
__host__ __device__ inline cplx<T> operator+(cplx<T> a, cplx<T> b) { return mk<T>(a.x + b.x, a.y + b.y); }
template <typename T>
__host__ __device__ inline cplx<T> operator-(cplx<T> a, cplx<T> b) { return mk<T>(a.x - b.x, a.y - b.y); }
template <typename T>
__host__ __device__ inline cplx<T> cmul(cplx<T> a, cplx<T> b) {
  return mk<T>(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
template <typename T>
__host__ __device__ inline cplx<T> conj(cplx<T> a) { return mk<T>(a.x, -a.y); }

__device__ inline float si_hypot(float a, float b) { return hypotf(a, b); }
__device__ inline double si_hypot(double a, double b) { return hypot(a, b); }

template <typename T>
struct eps16;
template <>
struct eps16<float> {
  static constexpr float value = 1e-16f;
};
template <>
struct eps16<double> {
  static constexpr double value = 1e-16;
};

// wave-wide (64 lanes) sum in double
__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

constexpr int kWave = 64;
constexpr int kMaxStages = 16;

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace specinv
