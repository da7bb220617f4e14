// The wave-level coverage kernel (kernels_wave.h) and its host entry points for double: n_fft 128 ... 2048 x {Griffin-Lim, ADMM} x
// {one-sided, two-sided} x {plain, evaluating} x overlap-add {frames buffer, LDS ring, registers at hop = n_fft / 2, / 4, / 8}.
#include "tu_wave_impl.h"

namespace specinv {

SPECINV_WAVE_FAMILY(double, 0)
SPECINV_WAVE_PUBLIC(double)

}  // namespace specinv
