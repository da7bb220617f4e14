// The wave-level coverage kernel (kernels_wave.h): float32 / float64 x n_fft 128 ... 2048 x {Griffin-Lim, ADMM}, and its host entry
// points (wave_api.h).
#include <hip/hip_runtime.h>

#include "kernels_wave.h"

namespace specinv {

bool wave_iter_covers(int n_fft) { return n_fft == 128 || n_fft == 256 || n_fft == 512 || n_fft == 1024 || n_fft == 2048; }

template <typename T>
int wave_iter_waves(int n_fft, int64_t frames_total, int* waves_per_workgroup) {
  wave::Launch l{};
  // (the two modes of a size share their launch shape unless their register counts part them: the evaluation's partial sums are
  // sized for the larger)
  auto both = [&](auto tag) {
    constexpr int LOGM = decltype(tag)::value;
    wave::Launch best{};
    for (int mode = 4; mode < 8; ++mode) {       // (the evaluating instantiations: theirs are the partial sums)
      const wave::Launch a = wave::shape<T, LOGM>(frames_total, mode);
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
int wave_iter_launch(const WaveIterArgs<T>& a, hipStream_t stream) {
  switch (a.c.n_fft) {
    case 128: return wave::launch_one<T, 6>(a, stream);
    case 256: return wave::launch_one<T, 7>(a, stream);
    case 512: return wave::launch_one<T, 8>(a, stream);
    case 1024: return wave::launch_one<T, 9>(a, stream);
    case 2048: return wave::launch_one<T, 10>(a, stream);
    default: break;
  }
  SI_CHECK(false, SPECINV_EUNSUPPORTED, "k_wave_iter does not cover n_fft=%d", a.c.n_fft);
  return SPECINV_OK;
}

template int wave_iter_waves<float>(int, int64_t, int*);
template int wave_iter_waves<double>(int, int64_t, int*);
template int wave_iter_launch<float>(const WaveIterArgs<float>&, hipStream_t);
template int wave_iter_launch<double>(const WaveIterArgs<double>&, hipStream_t);

}  // namespace specinv
