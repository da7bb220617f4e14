// The wave-level coverage kernel (kernels_wave.h) for double at n_fft 400 / 800 / 1000 (radix 5 / 10 / 20 passes; frames buffer or LDS ring).
#include "tu_wave_impl.h"

namespace specinv {

SPECINV_WAVE_FAMILY(double, 1)

}  // namespace specinv
