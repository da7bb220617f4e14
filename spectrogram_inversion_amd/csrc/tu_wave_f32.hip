// The wave-level coverage kernel (kernels_wave.h) and its host entry points for float: n_fft 128 ... 2048 x {Griffin-Lim, ADMM} x
// {one-sided, two-sided} x {plain, evaluating} x overlap-add {frames buffer, LDS ring, registers at hop = n_fft / 2, / 4, / 8}.
#include "tu_wave_impl.h"

namespace specinv {

bool wave_iter_covers(int n_fft, int elem_size) {
  if (n_fft == 16384) return elem_size == 4;
  return n_fft == 128 || n_fft == 256 || n_fft == 512 || n_fft == 1024 || n_fft == 2048 || n_fft == 4096 || n_fft == 8192 || wave_smooth_size(n_fft);
}

SPECINV_WAVE_FAMILY(float, 0)
SPECINV_WAVE_PUBLIC(float)

}  // namespace specinv
