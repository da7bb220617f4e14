// RTISI-LA (reference: torch_specinv/methods.py:273-412) as ONE persistent launch.
//
// The algorithm is frame-serial per batch item ((T + LA) * max_iter strictly dependent inner steps) and
// batch-parallel, so one workgroup owns one batch item for the whole recursion: no host round trip and no
// kernel boundary between the 25 000+ dependent steps of BASELINE config 3.  Per step (methods.py:365-398):
// overlap-add of the K kept + LA+1 look-ahead frames, STFT of the LA+1 look-ahead frames (the newest one
// optionally with the asymmetric analysis window), momentum, magnitude projection, inverse FFT.  The FFTs run
// in LDS (same Stockham code as the generic path, any n_fft / sidedness / dtype); the frame ring and the
// previous spectra live in a per-item global scratch that stays L2-resident.
#pragma once
#include <vector>

#include <type_traits>

#include "common.h"
#include "kernels_generic.h"
#include "kernels_rtisi_fast.h"

namespace specinv {

template <typename T>
struct RtisiArgs {
  FrameCfg<T> c;
  const T* mag;      // (B, T, F) frame-major target
  T* ring;           // (B, K+LA+1, N) frame ring: K kept frames then LA+1 frames being updated
  cplx<T>* pre;      // (B, 2, LA+1, F) pre_spec, double-buffered (frames are updated concurrently)
  T* frames_out;     // (B, T, N) committed frames times the synthesis window (input of the final overlap-add)
  const T* wsyn;     // window * hop / (w.w)             (methods.py:318, :367)
  const T* asym1;    // asym_window1                     (methods.py:325-329)
  const T* asym2;    // asym_window2                     (methods.py:331-335)
  int keep, la, steps, max_iter, asym;
  int groups;        // look-ahead frames transformed concurrently (each thread group owns LDS FFT buffers)
  T lr;
};

template <typename T>
__global__ void k_rtisi(RtisiArgs<T> r) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const FrameCfg<T>& c = r.c;
  const int N = c.n_fft, F = c.n_freq, hop = c.hop;
  const int G = r.groups;
  const int gsz = blockDim.x / G;                  // threads per group
  const int g = threadIdx.x / gsz, gt = threadIdx.x - g * gsz;
  cplx<T>* a = reinterpret_cast<cplx<T>*>(smem) + (size_t)g * 2 * N;   // this group's FFT buffers
  cplx<T>* b = a + N;
  T* xbuf = reinterpret_cast<T*>(reinterpret_cast<cplx<T>*>(smem) + (size_t)G * 2 * N);   // la*hop + N samples
  const int bi = blockIdx.x;
  const int nslots = r.keep + r.la + 1;
  T* ring = r.ring + (int64_t)bi * nslots * N;
  cplx<T>* pre_base = r.pre + (int64_t)bi * 2 * (r.la + 1) * F;
  const T* mag = r.mag + (int64_t)bi * r.steps * F;
  T* fout = r.frames_out + (int64_t)bi * r.steps * N;
  const int xlen = r.la * hop + N;

  // ---- initial state (methods.py:353-358): zero frames, newest slot = irfft(first target frame, zero phase)
  for (int i = threadIdx.x; i < (nslots - 1) * N; i += blockDim.x) ring[i] = T(0);
  {
    cplx<T>* pa = a;
    cplx<T>* pb = b;
    if (g == 0)
      for (int f = gt; f < F; f += gsz) pa[f] = mk<T>(mag[f], T(0));
    __syncthreads();
    if (c.onesided) {
      if (g == 0)
        for (int f = gt; f <= N / 2; f += gsz) {
          const cplx<T> v = pa[f];
          if (f == 0 || 2 * f == N) pa[f] = mk<T>(v.x, T(0));
          else pa[N - f] = conj(v);
        }
      __syncthreads();
    }
    lds_fft(pa, pb, c, true, gt, gsz);
    if (g == 0) {
      T* dst = ring + (int64_t)(nslots - 1) * N;
      for (int k = gt; k < N; k += gsz) dst[k] = pa[k].x * c.inv_scale;
    }
    __syncthreads();
  }

  int base = 0;   // ring slot of frame 0 (oldest kept frame)
  int pcur = 0;   // pre_spec buffer read in this inner step (the other one is written)
  for (int i = 0; i < r.steps + r.la; ++i) {
    for (int j = 0; j < r.max_iter; ++j) {
      // ---- overlap-add of all K+LA+1 frames with the synthesis window, samples [K*hop, (K+LA)*hop + N)
      for (int np = threadIdx.x; np < xlen; np += blockDim.x) {
        const int n = np + r.keep * hop;
        int f_hi = n / hop;
        if (f_hi > nslots - 1) f_hi = nslots - 1;
        const int f_lo = n - N + 1 <= 0 ? 0 : (n - N + hop) / hop;
        T acc = 0;
        for (int f = f_lo; f <= f_hi; ++f) {
          int slot = base + f;
          if (slot >= nslots) slot -= nslots;
          const int k = n - f * hop;
          acc += ring[(int64_t)slot * N + k] * r.wsyn[k];
        }
        xbuf[np] = acc;
      }
      __syncthreads();
      const cplx<T>* pre_in = pre_base + (int64_t)pcur * (r.la + 1) * F;
      cplx<T>* pre_out = pre_base + (int64_t)(pcur ^ 1) * (r.la + 1) * F;
      for (int q0 = 0; q0 <= r.la; q0 += G) {
        const int q = q0 + g;
        const bool active = q <= r.la;          // idle groups still walk through every barrier
        cplx<T>* pa = a;
        cplx<T>* pb = b;
        if (active) {
          const T* win = (r.asym && q == r.la) ? (j ? r.asym2 : r.asym1) : c.window;   // methods.py:371-383
          for (int k = gt; k < N; k += gsz) pa[k] = mk<T>(xbuf[q * hop + k] * win[k], T(0));
        }
        __syncthreads();
        lds_fft(pa, pb, c, false, gt, gsz);
        if (active) {
          const int tt = i + q - r.la;          // target frame of look-ahead slot q (methods.py:339, :395)
          const bool valid = tt >= 0 && tt < r.steps;
          for (int f = gt; f < F; f += gsz) {
            cplx<T> s = mk<T>(pa[f].x * c.fwd_scale, pa[f].y * c.fwd_scale);
            if (j) {                            // methods.py:387-388
              const cplx<T> p = pre_in[(int64_t)q * F + f];
              s = mk<T>(s.x - r.lr * p.x, s.y - r.lr * p.y);
            } else if (i && q < r.la) {         // methods.py:389-391: frame-shifted momentum
              const cplx<T> p = pre_in[(int64_t)(q + 1) * F + f];
              s = mk<T>(s.x - r.lr * p.x, s.y - r.lr * p.y);
            }
            pre_out[(int64_t)q * F + f] = s;    // :392
            const T m = valid ? mag[(int64_t)tt * F + f] : T(0);
            const T inv = T(1) / (si_hypot(s.x, s.y) + eps16<T>::value);     // :394
            pa[f] = mk<T>((s.x * m) * inv, (s.y * m) * inv);                  // :395-396
          }
        }
        __syncthreads();
        if (c.onesided) {
          if (active)
            for (int f = gt; f <= N / 2; f += gsz) {
              const cplx<T> v = pa[f];
              if (f == 0 || 2 * f == N) pa[f] = mk<T>(v.x, T(0));
              else pa[N - f] = conj(v);
            }
          __syncthreads();
        }
        lds_fft(pa, pb, c, true, gt, gsz);
        if (active) {
          int slot = base + r.keep + q;
          if (slot >= nslots) slot -= nslots;
          T* dst = ring + (int64_t)slot * N;
          for (int k = gt; k < N; k += gsz) dst[k] = pa[k].x * c.inv_scale;    // :398
        }
        __syncthreads();
      }
      pcur ^= 1;
    }
    // ---- commit look-ahead slot 0 (methods.py:401-404) and slide the ring
    int s0 = base + r.keep;
    if (s0 >= nslots) s0 -= nslots;
    if (i >= r.la) {
      const T* src = ring + (int64_t)s0 * N;
      T* dst = fout + (int64_t)(i - r.la) * N;
      for (int k = threadIdx.x; k < N; k += blockDim.x) dst[k] = src[k] * c.window[k];
    }
    __syncthreads();
    T* fresh = ring + (int64_t)base * N;        // the oldest frame's slot becomes the new (zero) newest frame
    for (int k = threadIdx.x; k < N; k += blockDim.x) fresh[k] = T(0);
    base = base + 1 == nslots ? 0 : base + 1;
    __syncthreads();
  }
}

template <typename P, typename T>
int rtisi_launch(P& pl, const T* mag_user, int look_ahead, int asym, int max_iter, double alpha, T* x_out) {
  SI_CHECK(mag_user && x_out, SPECINV_EINVAL, "null pointer");
  SI_CHECK(max_iter > 0, SPECINV_EINVAL, "max_iter must be > 0");     // methods.py:295
  SI_CHECK(alpha >= 0, SPECINV_EINVAL, "alpha must be >= 0");        // methods.py:296
  const int N = pl.N(), hop = pl.cfg.hop_length, F = pl.n_freq, Tn = pl.Tn(), Bn = pl.B();
  const int keep = (N - 1) / hop;                                     // methods.py:322
  const int la = look_ahead < 0 ? keep : look_ahead;                  // :323-324
  const int nslots = keep + la + 1;

  // windows (methods.py:318-336), evaluated in T like the reference does
  const std::vector<T>& w = pl.h_window;
  T dot = 0;
  for (int k = 0; k < N; ++k) dot += w[k] * w[k];
  const T coeff = (T)hop / dot;
  std::vector<T> wsyn(N), a1(N, T(0)), a2(N, T(0));
  for (int k = 0; k < N; ++k) wsyn[k] = w[k] * coeff;
  for (int i = 0; i < keep; ++i) {
    const int s = (i + 1) * hop;
    for (int k = s; k < N; ++k) a1[k] += w[N - 1 - (k - s)];
  }
  for (int i = 0; i <= keep; ++i) {
    const int s = i * hop;
    for (int k = s; k < N; ++k) a2[k] += w[N - 1 - (k - s)];
  }
  for (int k = 0; k < N; ++k) {
    a1[k] *= coeff;
    a2[k] *= coeff;
  }
  const size_t ring_elems = (size_t)Bn * nslots * N;
  const size_t pre_elems = (size_t)Bn * 2 * (la + 1) * F;
  SI_TRY(pl.rt_state.reserve(ring_elems * sizeof(T) + pre_elems * sizeof(cplx<T>) + 3 * (size_t)N * sizeof(T) + 64));
  char* basep = static_cast<char*>(pl.rt_state.p);
  cplx<T>* d_pre = reinterpret_cast<cplx<T>*>(basep);
  T* d_ring = reinterpret_cast<T*>(basep + pre_elems * sizeof(cplx<T>));
  T* d_wsyn = d_ring + ring_elems;
  T* d_a1 = d_wsyn + N;
  T* d_a2 = d_a1 + N;
  SI_HIP(hipMemcpyAsync(d_wsyn, wsyn.data(), N * sizeof(T), hipMemcpyHostToDevice, pl.stream));
  SI_HIP(hipMemcpyAsync(d_a1, a1.data(), N * sizeof(T), hipMemcpyHostToDevice, pl.stream));
  SI_HIP(hipMemcpyAsync(d_a2, a2.data(), N * sizeof(T), hipMemcpyHostToDevice, pl.stream));
  SI_HIP(hipStreamSynchronize(pl.stream));    // the host vectors go out of scope

  if constexpr (std::is_same<T, float>::value) {
    bool used = false;
    SI_TRY(rtisi_fast_launch(pl, mag_user, la, asym, max_iter, alpha, x_out, d_wsyn, d_a1, d_a2, &used));
    if (used) return SPECINV_OK;
  }

  SI_TRY(pl.mag.reserve(pl.nspec() * sizeof(T)));
  SI_TRY((pl.template transpose<T>(mag_user, pl.mag.template as<T>(), F, Tn)));
  SI_TRY(pl.frames_needed());

  RtisiArgs<T> r;
  r.c = pl.fc;
  r.mag = pl.mag.template as<T>();
  r.ring = d_ring;
  r.pre = d_pre;
  r.frames_out = pl.frames.template as<T>();
  r.wsyn = d_wsyn;
  r.asym1 = d_a1;
  r.asym2 = d_a2;
  r.keep = keep;
  r.la = la;
  r.steps = Tn;
  r.max_iter = max_iter;
  r.asym = asym ? 1 : 0;
  r.lr = (T)(alpha / (1.0 + alpha));                                  // methods.py:360
  const int threads = N >= 2048 ? 1024 : (N >= 1024 ? 512 : 256);
  // as many look-ahead frames in flight as LDS (2 FFT buffers per group) and threads (>= 128 per group) allow
  const size_t xbytes = ((size_t)la * hop + N) * sizeof(T);
  int groups = 1;
  while (groups * 2 <= la + 1 && threads / (groups * 2) >= 128 &&
         (size_t)(groups * 2) * 2 * N * sizeof(cplx<T>) + xbytes <= 150 * 1024)
    groups *= 2;
  r.groups = groups;
  const size_t lds = (size_t)groups * 2 * N * sizeof(cplx<T>) + xbytes;
  SI_CHECK(lds <= 160 * 1024 - 512, SPECINV_EUNSUPPORTED, "RTISI_LA: n_fft=%d look_ahead=%d needs %zu bytes of LDS", N, la,
           lds);
  SI_HIP(hipFuncSetAttribute((const void*)k_rtisi<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL((k_rtisi<T>), dim3(Bn), dim3(threads), lds, pl.stream, r);
  SI_HIP(hipGetLastError());
  return pl.launch_ola(pl.frames.template as<T>(), x_out, true);     // methods.py:406-408
}

}  // namespace specinv
