// Host interface of the wave-level coverage kernel (kernels_wave.h, compiled in tu_wave.hip): one Griffin-Lim / ADMM iteration's
// frame part - torch_specinv/methods.py:241-248, :464-477 - for float32 AND float64 at power-of-two n_fft 128 ... 2048, any hop,
// centring, pad mode, sidedness and `normalized`, on the coverage path's buffers (x, the (B, T, F) state and target, the frames
// that k_ola overlap-adds): what k_iter_pair / k_iter_pair_dr compute, on a transform that lives in a wave instead of a workgroup.
#pragma once
#include "kernels_generic.h"

namespace specinv {

template <typename T>
struct WaveIterArgs {
  FrameCfg<T> c;            // n_fft, n_freq, n_frames, hop, pad, pad_mode, onesided, length, scales, tw (n_fft entries), window
  const T* x;               // (B, length)
  cplx<T>* S0;              // (B, T, F): pre_spec (Griffin-Lim) / X (ADMM)
  cplx<T>* S1;              //            U (ADMM)
  const T* mag;             // (B, T, F)
  T coef, inv1p;
  T* frames;                // (B, T, n_fft) out: windowed synthesis frames
  double* partials;         // [waves][2] evaluation sums (eval != 0)
  int batch, mode, eval;    // mode 0 Griffin-Lim, 1 ADMM
  // overlap-add in registers (hop = n_fft / 2, / 4 or / 8; wave_iter_ola_chunks() > 0): no frames buffer - the new signal goes to
  // x_out (another buffer than x: other waves still read x), the blocks at chunk boundaries through seamL / seamR and k_wave_seams
  T* x_out = nullptr;
  const T* env = nullptr;   // (length) window-square envelope
  T* seamL = nullptr;       // [batch * nch][keep] each, keep = (ov - 1) hop (registers) or n_fft - hop (ring)
  T* seamR = nullptr;
  int nch = 0, ov = 0;      // chunks of frames per item; wave_iter_ola_chunks' ov_out (0: frames buffer + k_ola)
};

// n_fft the kernel covers (a power of two, 128 ... 2048)
bool wave_iter_covers(int n_fft, int elem_size);
// ... and the frame counts its 32-bit frame indices and row offsets take (`chunks`: with the register overlap-add, whose lane groups
// walk chunks up to n_frames apart); beyond them the plan uses the frames form / the workgroup-level kernels
inline bool wave_iter_fits(int n_fft, int n_frames, int batch, bool chunks) {
  const int64_t total = (int64_t)batch * n_frames, lim = (int64_t)1 << 31;
  return total * 1 < lim && (int64_t)8 * (n_fft + 2) < lim && (!chunks || (int64_t)8 * n_frames * (n_fft + 2) < lim);
}
// workgroups x waves the launch will use for `frames_total` frames: the number of partial-sum pairs an evaluating launch leaves
template <typename T>
int wave_iter_waves(int n_fft, int64_t frames_total, int* waves_per_workgroup = nullptr);
// `waves_out`: waves the launch used (an evaluating launch leaves that many pairs of partial sums)
template <typename T>
int wave_iter_launch(const WaveIterArgs<T>& a, hipStream_t stream, int* waves_out = nullptr);
// Chunks of frames per item for the register overlap-add, 0 where it does not apply (hop is not n_fft / 2, / 4, / 8; a two-sided
// spectrogram; a float64 frame of 16 points per lane, whose partial sums would not fit the registers; fewer than 2 OV frames).
// wave_iter_launch with nch > 0 also runs k_wave_seams.
// `ov_out`: how - 2 / 4 / 8 = n_fft / hop, the partial sums in registers; 1: in an LDS ring (any other hop < n_fft, two-sided)
template <typename T>
int wave_iter_ola_chunks(int n_fft, int hop, int n_frames, int batch, bool onesided, int* ov_out = nullptr);
// diagnostics (specinv_plan_launch_geometry): out = {waves per workgroup, chunks of frames per item (overlap-add in the kernel) or
// the frame count, waves of a plain launch, kernel code: 9 with the LDS ring, else 8}
template <typename T>
void wave_iter_geometry(int n_fft, int hop, int n_frames, int batch, bool onesided, int out[4]);

}  // namespace specinv
