// The wave-level coverage kernel (kernels_wave.h) for double at n_fft 4096 / 8192: a frame on a team of four / eight waves.
#include "tu_wave_impl.h"

namespace specinv {

SPECINV_WAVE_FAMILY(double, 2)

}  // namespace specinv
