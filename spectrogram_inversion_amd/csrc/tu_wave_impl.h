// Template definitions of the wave-level coverage kernel's host entry points (wave_api.h); tu_wave_f32.hip / tu_wave_f64.hip
// instantiate them - and with them the kernels of kernels_wave.h - for one element type each (two units: they build side by side).
#pragma once
#include <hip/hip_runtime.h>

#include "kernels_wave.h"

namespace specinv {

template <typename T>
int wave_iter_waves(int n_fft, int64_t frames_total, int* waves_per_workgroup) {
  wave::Launch l{};
  // (the two modes of a size share their launch shape unless their register counts part them: the evaluation's partial sums are
  // sized for the larger)
  auto both = [&](auto tag) {
    constexpr int LOGM = decltype(tag)::value;
    wave::Launch best{};
    for (int mode = 4; mode < 8; ++mode) {       // (the evaluating instantiations: theirs are the partial sums)
      const wave::Launch a = wave::shape<T, LOGM>(frames_total, mode, 0);
      if (a.wgs * a.waves_per_wg > best.wgs * best.waves_per_wg) best = a;
    }
    return best;
  };
  switch (n_fft) {
    case 128: l = both(std::integral_constant<int, 6>{}); break;
    case 256: l = both(std::integral_constant<int, 7>{}); break;
    case 512: l = both(std::integral_constant<int, 8>{}); break;
    case 1024: l = both(std::integral_constant<int, 9>{}); break;
    case 2048: l = both(std::integral_constant<int, 10>{}); break;
    default: return 0;
  }
  if (waves_per_workgroup) *waves_per_workgroup = l.waves_per_wg;
  return l.wgs * l.waves_per_wg;
}

template <typename T>
int wave_iter_launch(const WaveIterArgs<T>& a, hipStream_t stream, int* waves_out) {
  switch (a.c.n_fft) {
    case 128: return wave::launch_one<T, 6>(a, stream, waves_out);
    case 256: return wave::launch_one<T, 7>(a, stream, waves_out);
    case 512: return wave::launch_one<T, 8>(a, stream, waves_out);
    case 1024: return wave::launch_one<T, 9>(a, stream, waves_out);
    case 2048: return wave::launch_one<T, 10>(a, stream, waves_out);
    default: break;
  }
  SI_CHECK(false, SPECINV_EUNSUPPORTED, "k_wave_iter does not cover n_fft=%d", a.c.n_fft);
  return SPECINV_OK;
}

template <typename T>
int wave_iter_ola_chunks(int n_fft, int hop, int n_frames, int batch, bool onesided, int* ov_out) {
  if (ov_out) *ov_out = 0;
  if (hop <= 0 || hop >= n_fft) return 0;
  // registers where hop = n_fft / 2, / 4, / 8 of a one-sided spectrogram and the partial sums fit; the LDS ring for every other hop
  // below n_fft and for two-sided spectrograms (SPECINV_WAVE_RING=0: frames + k_ola there)
  auto pick = [&](auto tag) -> int {
    constexpr int LOGM = decltype(tag)::value;
    const int ovd = n_fft % hop == 0 ? n_fft / hop : 0;
    int ov = onesided && wave::ola_registers<T, LOGM>(ovd) ? ovd : 1;
    if (ov == 1) {
      if (const char* e = getenv("SPECINV_WAVE_RING")) {
        if (e[0] == '0') return 0;
      }
    }
    const int nch = wave::ola_chunks<T, LOGM>(ov, (n_fft + hop - 1) / hop, n_frames, batch, onesided ? 0 : 2);
    if (nch > 0 && ov_out) *ov_out = ov;
    return nch;
  };
  switch (n_fft) {
    case 128: return pick(std::integral_constant<int, 6>{});
    case 256: return pick(std::integral_constant<int, 7>{});
    case 512: return pick(std::integral_constant<int, 8>{});
    case 1024: return pick(std::integral_constant<int, 9>{});
    case 2048: return pick(std::integral_constant<int, 10>{});
    default: return 0;
  }
}

template <typename T>
void wave_iter_geometry(int n_fft, int hop, int n_frames, int batch, bool onesided, int out[3]) {
  int ov = 0;
  const int nch = wave_iter_ola_chunks<T>(n_fft, hop, n_frames, batch, onesided, &ov);
  const int mode = onesided ? 0 : 2;
  const int64_t work = nch > 0 ? (int64_t)batch * nch : (int64_t)batch * n_frames;
  wave::Launch l{};
  switch (n_fft) {
    case 128: l = wave::shape<T, 6>(work, mode, ov); break;
    case 256: l = wave::shape<T, 7>(work, mode, ov); break;
    case 512: l = wave::shape<T, 8>(work, mode, ov); break;
    case 1024: l = wave::shape<T, 9>(work, mode, ov); break;
    case 2048: l = wave::shape<T, 10>(work, mode, ov); break;
    default: break;
  }
  out[0] = l.waves_per_wg;
  out[1] = nch > 0 ? nch : n_frames;
  out[2] = l.wgs * l.waves_per_wg;
}



}  // namespace specinv
