// The wave-level coverage kernel (kernels_wave.h) for float at n_fft 4096 / 8192: a frame on a team of two / four waves (16384: tu_wave_f32c.hip).
#include "tu_wave_impl.h"

namespace specinv {

SPECINV_WAVE_FAMILY(float, 2)

}  // namespace specinv
