// The wave-level coverage kernel (kernels_wave.h) and its host entry points for float: n_fft 128 ... 2048 x {Griffin-Lim, ADMM} x
// {one-sided, two-sided} x {plain, evaluating} x overlap-add {frames buffer, registers at hop = n_fft / 2, / 4, / 8}.
#include "tu_wave_impl.h"

namespace specinv {

bool wave_iter_covers(int n_fft) { return n_fft == 128 || n_fft == 256 || n_fft == 512 || n_fft == 1024 || n_fft == 2048; }

template int wave_iter_waves<float>(int, int64_t, int*);
template int wave_iter_launch<float>(const WaveIterArgs<float>&, hipStream_t, int*);
template int wave_iter_ola_chunks<float>(int, int, int, int, bool, int*);
template void wave_iter_geometry<float>(int, int, int, int, bool, int*);

}  // namespace specinv
