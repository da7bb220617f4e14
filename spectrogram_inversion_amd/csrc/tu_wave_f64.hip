// The wave-level coverage kernel (kernels_wave.h) and its host entry points for double: n_fft 128 ... 2048 x {Griffin-Lim, ADMM} x
// {one-sided, two-sided} x {plain, evaluating} x overlap-add {frames buffer, registers at hop = n_fft / 2, / 4, / 8}.
#include "tu_wave_impl.h"

namespace specinv {

template int wave_iter_waves<double>(int, int64_t, int*);
template int wave_iter_launch<double>(const WaveIterArgs<double>&, hipStream_t, int*);
template int wave_iter_ola_chunks<double>(int, int, int, int, bool, int*);
template void wave_iter_geometry<double>(int, int, int, int, bool, int*);

}  // namespace specinv
