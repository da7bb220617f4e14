// Frames beyond what one workgroup's LDS holds (n_fft > 16384 in float32, > 8192 in float64): the transform in FOUR STEPS through
// device memory.  The reference derives n_fft from the spectrogram without a bound (torch_specinv/methods.py:65-68) and hands the
// frame to torch.stft / torch.fft (:241, :142-146); here n_fft = N1 * N2 with N1 in {2, 4, 8} and N2 a size the LDS kernels of
// kernels_generic.h take:
//
//     X[k1 + N1 k2] = sum_n2 W_N2^(n2 k2) [ W_N^(n2 k1) sum_n1 z[n1 N2 + n2] W_N1^(n1 k1) ]
//
//   k_big_pass1   windowed frame -> radix-N1 butterflies across n1 (registers), times W_N^(n2 k1) -> y[item][k1][n2]
//   k_big_rows    the N1 rows of an item: an N2-point transform each, in LDS (lds_fft with the sub-transform's stages), in place
//   k_big_update  the bin update of the iteration (update_core: methods.py:243-247 / :467-475) on bins (f, N - f); bin k sits at
//                 row k mod N1, column k div N1
//   k_big_rows    ... inverse
//   k_big_pass4   times conj W_N^(n2 k1), inverse butterflies across k1, scale, synthesis window -> frames (then k_ola as usual)
//   k_big_split_out / k_big_merge_in   the spectrum rows of specinv_stft / the inverse frames of specinv_istft and the adjoints
// One frame per item (no frame pairing: a non-finite sample stays inside its own frame by construction).  This is a coverage path
// - five passes over an item's n_fft complex points per iteration - for sizes no audio front end uses; it exists so that no
// n_fft the reference accepts is refused below 65536 (float32) / 32768 (float64).
#pragma once
#include "kernels_generic.h"

namespace specinv {

template <typename T>
struct BigCfg {
  FrameCfg<T> f;       // the frame: n_fft = N, window, hop, pad, scales; f.tw = W_N^n, n < N
  FrameCfg<T> s;       // the sub-transform: n_fft = N2, its stages, its own twiddle table
  int n1, n2, n_frames;
  cplx<T>* y;          // [batch * frames][N]: row k1 of an item at k1 * N2
};

template <typename T>
__device__ __forceinline__ int big_pos(const BigCfg<T>& c, int k) {     // where bin k of an item lives after the forward rows
  const int k2 = k / c.n1;
  return (k - k2 * c.n1) * c.n2 + k2;
}

template <typename T, int N1>
__global__ __launch_bounds__(256) void k_big_pass1(BigCfg<T> c, const T* __restrict__ x) {
  const int n2 = blockIdx.x * blockDim.x + threadIdx.x, t = blockIdx.y, bi = blockIdx.z;
  if (n2 >= c.n2) return;
  const T* xr = x + (int64_t)bi * c.f.length;
  const int64_t start = (int64_t)t * c.f.hop - c.f.pad;
  cplx<T> v[N1];
#pragma unroll
  for (int q = 0; q < N1; ++q) {
    const int n = q * c.n2 + n2;
    v[q] = mk<T>(load_padded(xr, c.f.length, start + n, c.f.pad_mode) * c.f.window[n], T(0));
  }
  Butterfly<T, N1, false>::run(v, c.f.tw, c.f.n_fft);
  cplx<T>* out = c.y + ((int64_t)bi * c.n_frames + t) * c.f.n_fft + n2;
  out[0] = v[0];
#pragma unroll
  for (int k1 = 1; k1 < N1; ++k1) out[(int64_t)k1 * c.n2] = cmul(v[k1], c.f.tw[n2 * k1]);
}

template <typename T, int N1>
__global__ __launch_bounds__(256) void k_big_pass4(BigCfg<T> c, T* __restrict__ frames) {
  const int n2 = blockIdx.x * blockDim.x + threadIdx.x, t = blockIdx.y, bi = blockIdx.z;
  if (n2 >= c.n2) return;
  const int64_t item = (int64_t)bi * c.n_frames + t;
  const cplx<T>* in = c.y + item * c.f.n_fft + n2;
  cplx<T> v[N1];
  v[0] = in[0];
#pragma unroll
  for (int k1 = 1; k1 < N1; ++k1) v[k1] = cmul(in[(int64_t)k1 * c.n2], conj(c.f.tw[n2 * k1]));
  Butterfly<T, N1, true>::run(v, c.f.tw, c.f.n_fft);
  T* fr = frames + item * c.f.n_fft;
#pragma unroll
  for (int q = 0; q < N1; ++q) {
    const int n = q * c.n2 + n2;
    fr[n] = (v[q].x * c.f.inv_scale) * c.f.window[n];
  }
}

// one row (k1) of one item: N2 points through LDS, back to where they came from
template <typename T, bool IP, bool INV>
__global__ void k_big_rows(BigCfg<T> c) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cplx<T>* a = reinterpret_cast<cplx<T>*>(smem);
  cplx<T>* b = IP ? a : a + c.n2;
  cplx<T>* row = c.y + ((int64_t)blockIdx.z * c.n_frames + blockIdx.y) * c.f.n_fft + (int64_t)blockIdx.x * c.n2;
  for (int i = threadIdx.x; i < c.n2; i += blockDim.x) a[i] = row[i];
  __syncthreads();
  lds_fft<T, IP>(a, b, c.s, INV);
  for (int i = threadIdx.x; i < c.n2; i += blockDim.x) row[i] = a[i];
}

// the iteration's bin update (k_iter_pair's, one frame per item): bins f <= N / 2 and their mirror images
template <typename T, int MODE, bool EVAL>
__global__ __launch_bounds__(256) void k_big_update(BigCfg<T> c, cplx<T>* __restrict__ S0, cplx<T>* __restrict__ S1,
                                                    const T* __restrict__ mag, T coef, T inv1p, double* __restrict__ partials) {
  __shared__ double red[16];
  const int N = c.f.n_fft, F = c.f.n_freq;
  const int f = blockIdx.x * blockDim.x + threadIdx.x, t = blockIdx.y, bi = blockIdx.z;
  const int64_t item = (int64_t)bi * c.n_frames + t, base = item * F;
  cplx<T>* z = c.y + item * N;
  double s_d = 0, s_o = 0;
  if (f <= N / 2) {
    const int g = f ? N - f : 0;
    const int pf = big_pos(c, f), pg = g == f ? pf : big_pos(c, g);
    const cplx<T> zf = z[pf], zg = z[pg];
    const T hs = T(0.5) * c.f.fwd_scale;
    const cplx<T> ra = mk<T>((zf.x + zg.x) * hs, (zf.y - zg.y) * hs);       // the real frame's bin f (its mirror image: the conjugate)
    const cplx<T> zero = mk<T>(T(0), T(0));
    cplx<T> ha;
    if (c.f.onesided) {
      cplx<T> n0, n1;
      ha = update_core<T, MODE>(ra, mag[base + f], S0[base + f], MODE == 1 ? S1[base + f] : zero, coef, inv1p, EVAL, s_d, s_o, n0, n1);
      S0[base + f] = n0;
      if (MODE == 1) S1[base + f] = n1;
      if (g == f || 2 * f == N) ha.y = T(0);                                // irfft ignores the imaginary parts of DC / Nyquist
    } else {
      const cplx<T> yaf = update_one<T, MODE>(ra, S0, S1, mag, base + f, coef, inv1p, EVAL, s_d, s_o);
      cplx<T> yag = yaf;
      if (g != f) yag = update_one<T, MODE>(conj(ra), S0, S1, mag, base + g, coef, inv1p, EVAL, s_d, s_o);
      ha = mk<T>(T(0.5) * (yaf.x + yag.x), T(0.5) * (yaf.y - yag.y));        // Hermitian part: what ifft(.).real sees
    }
    z[pf] = ha;
    if (g != f) z[pg] = conj(ha);
  }
  if (EVAL) {
    const double d = block_sum(s_d, red);
    const double o = block_sum(s_o, red);
    if (threadIdx.x == 0) {
      const int64_t pi = ((int64_t)bi * gridDim.y + t) * gridDim.x + blockIdx.x;
      partials[2 * pi] = d;
      partials[2 * pi + 1] = o;
    }
  }
}

// spectrum rows (B, T, F) of the forward transform, scaled
template <typename T>
__global__ __launch_bounds__(256) void k_big_split_out(BigCfg<T> c, cplx<T>* __restrict__ spec, T scale) {
  const int N = c.f.n_fft, F = c.f.n_freq;
  const int f = blockIdx.x * blockDim.x + threadIdx.x, t = blockIdx.y, bi = blockIdx.z;
  if (f >= F) return;
  const int64_t item = (int64_t)bi * c.n_frames + t;
  const cplx<T> v = c.y[item * N + big_pos(c, f)];
  spec[item * F + f] = mk<T>(v.x * scale, v.y * scale);
}

// the spectrum an inverse real transform sees (irfft: Hermitian extension; ifft(.).real: Hermitian part) -> y, ready for the rows
template <typename T>
__global__ __launch_bounds__(256) void k_big_merge_in(BigCfg<T> c, const cplx<T>* __restrict__ spec) {
  const int N = c.f.n_fft, F = c.f.n_freq;
  const int f = blockIdx.x * blockDim.x + threadIdx.x, t = blockIdx.y, bi = blockIdx.z;
  if (f > N / 2) return;
  const int64_t item = (int64_t)bi * c.n_frames + t;
  const cplx<T>* in = spec + item * F;
  cplx<T>* z = c.y + item * N;
  const int g = f ? N - f : 0;
  cplx<T> ha = in[f];
  if (c.f.onesided) {
    if (g == f || 2 * f == N) ha.y = T(0);
  } else {
    const cplx<T> vg = in[g];
    ha = mk<T>(T(0.5) * (ha.x + vg.x), T(0.5) * (ha.y - vg.y));
  }
  z[big_pos(c, f)] = ha;
  if (g != f) z[big_pos(c, g)] = conj(ha);
}

}  // namespace specinv
