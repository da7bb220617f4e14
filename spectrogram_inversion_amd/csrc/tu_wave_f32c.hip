// The wave-level coverage kernel (kernels_wave.h) for float at n_fft 16384 (a frame on a team of eight waves), and the double
// family's empty stand-in: float64 frames of that size are kernels_big.h's.
#include "tu_wave_impl.h"

namespace specinv {

SPECINV_WAVE_FAMILY(float, 3)
SPECINV_WAVE_FAMILY(double, 3)

}  // namespace specinv
