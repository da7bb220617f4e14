// Template definitions of the wave-level coverage kernel's host entry points (wave_api.h).  The kernels of kernels_wave.h are
// instantiated in seven units that build side by side: tu_wave_f32.hip / tu_wave_f64.hip (FAM 0: n_fft 128 ... 2048, and the public
// entry points), tu_wave_f32s.hip / tu_wave_f64s.hip (FAM 1: n_fft 400 / 800 / 1000) and tu_wave_f32b.hip / tu_wave_f64b.hip (FAM 2:
// n_fft 4096 / 8192, a frame on a team of two to eight waves; tu_wave_f32c.hip, FAM 3: float32 n_fft 16384).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "kernels_wave.h"

namespace specinv {

inline bool wave_smooth_size(int n_fft) { return n_fft == 400 || n_fft == 800 || n_fft == 1000; }
inline int wave_family(int n_fft) { return wave_smooth_size(n_fft) ? 1 : n_fft == 16384 ? 3 : n_fft >= 4096 ? 2 : 0; }

// f(size tag) for the family's size n_fft, `dflt` for any other
template <int FAM, typename T, typename F, typename R>
R wave_by_size(int n_fft, F&& f, R dflt) {
  if constexpr (FAM == 0) {
    switch (n_fft) {
      case 128: return f(std::integral_constant<int, 6>{});
      case 256: return f(std::integral_constant<int, 7>{});
      case 512: return f(std::integral_constant<int, 8>{});
      case 1024: return f(std::integral_constant<int, 9>{});
      case 2048: return f(std::integral_constant<int, 10>{});
      default: return dflt;
    }
  } else if constexpr (FAM == 1) {
    switch (n_fft) {
      case 400: return f(std::integral_constant<int, 200>{});
      case 800: return f(std::integral_constant<int, 400>{});
      case 1000: return f(std::integral_constant<int, 500>{});
      default: return dflt;
    }
  } else if constexpr (FAM == 2) {
    return n_fft == 4096 ? f(std::integral_constant<int, 11>{}) : n_fft == 8192 ? f(std::integral_constant<int, 12>{}) : dflt;
  } else {
    if constexpr (sizeof(T) == 4) {
      if (n_fft == 16384) return f(std::integral_constant<int, 13>{});
    }
    return dflt;
  }
}

template <typename T, int FAM>
int wave_iter_waves_f(int n_fft, int64_t frames_total, int* waves_per_workgroup) {
  // (the two modes of a size share their launch shape unless their register counts part them: the evaluation's partial sums are
  // sized for the larger)
  const wave::Launch l = wave_by_size<FAM, T>(n_fft, [&](auto tag) {
    constexpr int LOGM = decltype(tag)::value;
    wave::Launch best{};
    for (int mode = 4; mode < 8; ++mode) {       // (the evaluating instantiations: theirs are the partial sums)
      const wave::Launch a = wave::shape<T, LOGM>(frames_total, mode, 0);
      if (a.wgs * a.waves_per_wg > best.wgs * best.waves_per_wg) best = a;
    }
    return best;
  }, wave::Launch{});
  if (waves_per_workgroup) *waves_per_workgroup = l.waves_per_wg;
  return l.wgs * l.waves_per_wg;
}

template <typename T, int FAM>
int wave_iter_launch_f(const WaveIterArgs<T>& a, hipStream_t stream, int* waves_out) {
  const int rc = wave_by_size<FAM, T>(a.c.n_fft, [&](auto tag) { return wave::launch_one<T, decltype(tag)::value>(a, stream, waves_out); }, -1);
  SI_CHECK(rc != -1, SPECINV_EUNSUPPORTED, "k_wave_iter does not cover n_fft=%d", a.c.n_fft);
  return rc;
}

template <typename T, int FAM>
int wave_iter_ola_chunks_f(int n_fft, int hop, int n_frames, int batch, bool onesided, int* ov_out) {
  if (ov_out) *ov_out = 0;
  if (hop <= 0 || hop >= n_fft) return 0;
  // registers where hop = n_fft / 2, / 4, / 8 of a one-sided spectrogram and the partial sums fit; the LDS ring for every other hop
  // below n_fft and for two-sided spectrograms (SPECINV_WAVE_RING=0: frames + k_ola there)
  return wave_by_size<FAM, T>(n_fft, [&](auto tag) -> int {
    constexpr int LOGM = decltype(tag)::value;
    const int ovd = n_fft % hop == 0 ? n_fft / hop : 0;
    int ov = onesided && wave::ola_registers<T, LOGM>(ovd) ? ovd : 1;
    if (ov == 1) {
      // (float64 at n_fft 4096, both dtypes at 8192: frame buffer + ring are 69 - 135 KB per team, half the teams per CU - measured
      // behind frames + k_ola: float64 4096 / 3000 / 1000 0.463 against 0.394 ms, two-sided 0.466 against 0.414; float32 8192 / 6000 /
      // 1500 ADMM 0.310 against 0.247, two-sided 0.325 against 0.260)
      if (LOGM == 12 || LOGM == 13 || (sizeof(T) == 8 && LOGM == 11)) return 0;
      if (const char* e = getenv("SPECINV_WAVE_RING")) {
        if (e[0] == '0') return 0;
      }
    }
    const int nch = wave::ola_chunks<T, LOGM>(ov, (n_fft + hop - 1) / hop, n_frames, batch, onesided ? 0 : 2);
    if (nch > 0 && ov_out) *ov_out = ov;
    return nch;
  }, 0);
}

template <typename T, int FAM>
void wave_iter_geometry_f(int n_fft, int hop, int n_frames, int batch, bool onesided, int out[4]) {
  int ov = 0;
  const int nch = wave_iter_ola_chunks_f<T, FAM>(n_fft, hop, n_frames, batch, onesided, &ov);
  const int mode = onesided ? 0 : 2;
  const int64_t work = nch > 0 ? (int64_t)batch * nch : (int64_t)batch * n_frames;
  const wave::Launch l = wave_by_size<FAM, T>(n_fft, [&](auto tag) { return wave::shape<T, decltype(tag)::value>(work, mode, ov); }, wave::Launch{});
  out[0] = l.waves_per_wg;
  out[1] = nch > 0 ? nch : n_frames;
  out[2] = l.wgs * l.waves_per_wg;
  out[3] = ov == 1 ? 9 : 8;                   // (specinv_plan_launch_geometry's kernel codes)
}

// what a unit instantiates for its element type and family ...
#define SPECINV_WAVE_FAMILY(T, FAM)                                                                       \
  template int wave_iter_waves_f<T, FAM>(int, int64_t, int*);                                             \
  template int wave_iter_launch_f<T, FAM>(const WaveIterArgs<T>&, hipStream_t, int*);                     \
  template int wave_iter_ola_chunks_f<T, FAM>(int, int, int, int, bool, int*);                            \
  template void wave_iter_geometry_f<T, FAM>(int, int, int, int, bool, int*);
// ... and the public entry points of an element type (in its FAM 0 unit; the other families are other units')
#define SPECINV_WAVE_EXTERN(T, FAM)                                                                       \
  extern template int wave_iter_waves_f<T, FAM>(int, int64_t, int*);                                      \
  extern template int wave_iter_launch_f<T, FAM>(const WaveIterArgs<T>&, hipStream_t, int*);              \
  extern template int wave_iter_ola_chunks_f<T, FAM>(int, int, int, int, bool, int*);                     \
  extern template void wave_iter_geometry_f<T, FAM>(int, int, int, int, bool, int*);
#define SPECINV_WAVE_DISPATCH(T, n_fft, fn, ...)                                                          \
  (wave_family(n_fft) == 1 ? fn<T, 1>(__VA_ARGS__) : wave_family(n_fft) == 2 ? fn<T, 2>(__VA_ARGS__) :    \
   wave_family(n_fft) == 3 ? fn<T, 3>(__VA_ARGS__) : fn<T, 0>(__VA_ARGS__))
#define SPECINV_WAVE_PUBLIC(T)                                                                            \
  SPECINV_WAVE_EXTERN(T, 1)                                                                               \
  SPECINV_WAVE_EXTERN(T, 2)                                                                               \
  SPECINV_WAVE_EXTERN(T, 3)                                                                               \
  template <>                                                                                             \
  int wave_iter_waves<T>(int n_fft, int64_t frames_total, int* wpw) {                                     \
    return SPECINV_WAVE_DISPATCH(T, n_fft, wave_iter_waves_f, n_fft, frames_total, wpw);                  \
  }                                                                                                       \
  template <>                                                                                             \
  int wave_iter_launch<T>(const WaveIterArgs<T>& a, hipStream_t stream, int* waves_out) {                 \
    return SPECINV_WAVE_DISPATCH(T, a.c.n_fft, wave_iter_launch_f, a, stream, waves_out);                 \
  }                                                                                                       \
  template <>                                                                                             \
  int wave_iter_ola_chunks<T>(int n_fft, int hop, int n_frames, int batch, bool onesided, int* ov_out) {  \
    return SPECINV_WAVE_DISPATCH(T, n_fft, wave_iter_ola_chunks_f, n_fft, hop, n_frames, batch, onesided, ov_out); \
  }                                                                                                       \
  template <>                                                                                             \
  void wave_iter_geometry<T>(int n_fft, int hop, int n_frames, int batch, bool onesided, int out[4]) {    \
    SPECINV_WAVE_DISPATCH(T, n_fft, wave_iter_geometry_f, n_fft, hop, n_frames, batch, onesided, out);    \
  }

}  // namespace specinv
