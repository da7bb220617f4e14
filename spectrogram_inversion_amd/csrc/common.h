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

// 1 / (|s| + 1e-16), the factor of the reference's projection (methods.py:246-247, :472: s m / (|s| + 1e-16) - ATen multiplies by
// the rounded reciprocal), without the library's hypot and an IEEE division (~45 / ~60 instructions per bin):
//   float32: the wave-level kernels' chain (fast_core.h: ref_rcp_abs, SPECINV_REFCHAIN 2) - t = |s|^2 + 1e-32 by two fma, y =
//     v_rsq_f32(t), one Newton step to the correctly rounded t^-1/2; the guard is below float32's resolution above |s| = 3e-9;
//     below that, and where |s|^2 leaves float32's range, hypot and the division (the float64 overload's rule);
//   float64: y = v_rsq_f64(t) refined twice, h = t y corrected to the rounded square root, + 1e-16 (visible in float64: 1e-13
//     relative at |s| = 1e-3), one Newton step from y to 1 / (h + 1e-16) and a second for the last bits; tiny |s| (< 1e-6: the
//     seed 1 / h is no longer near 1 / (h + 1e-16)), huge or non-finite values take hypot and the division.
__device__ __forceinline__ float proj_inv(float x, float y) {
  const float t = fmaf(y, y, fmaf(x, x, 1e-32f));
  // (|s| above 1.8e19 - its square beyond float32 - or not finite, and |s| below 3e-9, where the squared floor and the reference's
  // additive guard part ways: the reference's own operations)
  if (!(t > 1e-17f && t < 1e37f)) return 1.0f / (hypotf(x, y) + 1e-16f);
  const float r = __builtin_amdgcn_rsqf(t);
  return fmaf(r * 0.5f, fmaf(-(t * r), r, 1.0f), r);
}
__device__ __forceinline__ double proj_inv(double x, double y) {
  const double t = fma(y, y, x * x);
  if (!(t > 1e-12 && t < 1e280)) return 1.0 / (hypot(x, y) + 1e-16);
  double r = __builtin_amdgcn_rsq(t);
  r = fma(0.5 * r, fma(-(t * r), r, 1.0), r);
  r = fma(0.5 * r, fma(-(t * r), r, 1.0), r);
  double h = t * r;
  h = fma(fma(-h, h, t), 0.5 * r, h);
  const double den = h + 1e-16;
  double q = fma(fma(-den, r, 1.0), r, r);
  q = fma(fma(-den, q, 1.0), q, q);
  return q;
}

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

// sin and cos of a phase (methods.py:612, exp(1j * phi)) in float64, for rounding to the phase's own type.
// A float32 phase (up to ~1.5e6 rad at 1024 frames, far more on long signals) does not need the library's general float64 sincos -
// its argument reduction alone is several hundred instructions at the float64 rate, and phase_init was bound by it: quadrant
// k = rint(phi * 2/pi), remainder r = phi - k * pi/2 by two fused multiply-adds on a two-term pi/2 (each rounds the exact
// difference once: |error| < 1e-16 + k * 1e-33), then the float64 kernels' polynomials on |r| <= pi/4 (the classic fdlibm minimax
// coefficients, 2^-58): the float64 results are good to ~1e-16, rounded to float32 they equal the library's except where the
// exact value sits within 1e-16 of a rounding boundary.  Phases beyond 1e9 rad (and NaN / inf) take the library path.
__device__ inline void sincos_phase(double phi, double* sn, double* cs) { sincos(phi, sn, cs); }
__device__ inline void sincos_phase(float phi_f, double* sn, double* cs) {
  const double phi = (double)phi_f;
  if (!(fabs(phi) < 1.0e9)) {
    sincos(phi, sn, cs);
    return;
  }
  const double k = rint(phi * 6.36619772367581382433e-01);
  double r = fma(-k, 1.57079632679489655800e+00, phi);
  r = fma(-k, 6.12323399573676603587e-17, r);
  const double z = r * r;
  double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  ps = fma(z, ps, 2.75573137070700676789e-06);
  ps = fma(z, ps, -1.98412698298579493134e-04);
  ps = fma(z, ps, 8.33333333332248946124e-03);
  ps = fma(z, ps, -1.66666666666666324348e-01);
  const double s = fma(r * z, ps, r);
  double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  pc = fma(z, pc, -2.75573143513906633035e-07);
  pc = fma(z, pc, 2.48015872894767294178e-05);
  pc = fma(z, pc, -1.38888888888741095749e-03);
  pc = fma(z, pc, 4.16666666666666019037e-02);
  const double c = fma(z * z, pc, fma(z, -0.5, 1.0));
  const int q = (int)k & 3;               // (|k| < 2^30: the conversion is exact, two's complement gives k mod 4)
  const double s1 = (q & 1) ? c : s, c1 = (q & 1) ? s : c;
  *sn = (q & 2) ? -s1 : s1;
  *cs = ((q + 1) & 2) ? -c1 : c1;
}

// wave-wide (64 lanes) sum in double
__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// inclusive prefix sum over the 64 lanes in double: shifts inside the rows of 16 lanes, then lane 15 of rows 0 / 2 into rows 1 / 3
// and lane 31 into the upper half - data-parallel-primitive moves on the vector unit (two per step) instead of the twelve
// ds_bpermute round trips of a shuffle ladder
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_scan_inclusive(double v) {
  v += dpp_move_f64<0x111, 0xf>(v);   // row_shr:1
  v += dpp_move_f64<0x112, 0xf>(v);   // row_shr:2
  v += dpp_move_f64<0x114, 0xf>(v);   // row_shr:4
  v += dpp_move_f64<0x118, 0xf>(v);   // row_shr:8
  v += dpp_move_f64<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
  v += dpp_move_f64<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3
  return v;
}

constexpr int kWave = 64;
constexpr int kMaxStages = 16;

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace specinv
