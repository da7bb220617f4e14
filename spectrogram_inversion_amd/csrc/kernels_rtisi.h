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
#include "rtisi_fast_host.h"

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
  // step range of this launch.  A whole-signal run is [0, steps + la) from scratch; a stream resumes from the
  // state the previous launch left in `ring` / `pre` and addresses `mag` / `frames_out` as rings of frames.
  int i_begin, i_end, resume;
  int n_valid;       // target frames that exist so far (look-ahead slots beyond them see a zero target, methods.py:339)
  int mag_ring;      // 0: mag is (B, steps, F); else (B, mag_ring, F) indexed by frame % mag_ring
  int out_ring;      // 0: frames_out is (B, steps, N); else (B, out_ring, N) indexed by frame % out_ring
  cplx<T>* rec;      // NULL, or (B, steps+la, max_iter, la+1, F): every pre-projection spectrum, for the adjoint
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
  const T* mag = r.mag + (int64_t)bi * (r.mag_ring ? r.mag_ring : r.steps) * F;
  T* fout = r.frames_out + (int64_t)bi * (r.out_ring ? r.out_ring : r.steps) * N;
  const int xlen = r.la * hop + N;

  // ---- initial state (methods.py:353-358): zero frames, newest slot = irfft(first target frame, zero phase)
  if (!r.resume) {
    for (int i = threadIdx.x; i < (nslots - 1) * N; i += blockDim.x) ring[i] = T(0);
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

  int base = r.i_begin % nslots;                                  // ring slot of the oldest kept frame
  int pcur = (int)(((int64_t)r.i_begin * r.max_iter) & 1);        // pre_spec buffer read in this inner step
  for (int i = r.i_begin; i < r.i_end; ++i) {
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
          const bool valid = tt >= 0 && tt < r.n_valid;
          const int mrow = r.mag_ring ? tt % r.mag_ring : tt;
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
            if (r.rec)
              r.rec[((((int64_t)bi * (r.steps + r.la) + i) * r.max_iter + j) * (r.la + 1) + q) * F + f] = s;
            const T m = valid ? mag[(int64_t)mrow * F + f] : T(0);
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
      const int cf = i - r.la;
      T* dst = fout + (int64_t)(r.out_ring ? cf % r.out_ring : cf) * N;
      for (int k = threadIdx.x; k < N; k += blockDim.x) dst[k] = src[k] * c.window[k];
    }
    __syncthreads();
    T* fresh = ring + (int64_t)base * N;        // the oldest frame's slot becomes the new (zero) newest frame
    for (int k = threadIdx.x; k < N; k += blockDim.x) fresh[k] = T(0);
    base = base + 1 == nslots ? 0 : base + 1;
    __syncthreads();
  }
}

// windows (methods.py:318-336, evaluated in T like the reference does) and the per-item state of one recursion
template <typename T>
struct RtisiLayout {
  int keep = 0, la = 0, nslots = 0;
  cplx<T>* pre = nullptr;
  T *ring = nullptr, *wsyn = nullptr, *asym1 = nullptr, *asym2 = nullptr;
  char* extra = nullptr;   // `extra_bytes` more, 16-byte aligned
};

template <typename P, typename T, typename Buf>
int rtisi_prepare(P& pl, int look_ahead, Buf& buf, size_t extra_bytes, RtisiLayout<T>& lay) {
  const int N = pl.N(), hop = pl.cfg.hop_length, F = pl.n_freq, Bn = pl.B();
  const int keep = (N - 1) / hop;                                     // methods.py:322
  const int la = look_ahead < 0 ? keep : look_ahead;                  // :323-324
  const int nslots = keep + la + 1;
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
  const size_t fixed = ring_elems * sizeof(T) + pre_elems * sizeof(cplx<T>) + 3 * (size_t)N * sizeof(T);
  const size_t fixed_al = (fixed + 63) & ~(size_t)63;
  SI_TRY(buf.reserve(fixed_al + extra_bytes + 64));
  char* basep = static_cast<char*>(buf.p);
  lay.keep = keep;
  lay.la = la;
  lay.nslots = nslots;
  lay.pre = reinterpret_cast<cplx<T>*>(basep);
  lay.ring = reinterpret_cast<T*>(basep + pre_elems * sizeof(cplx<T>));
  lay.wsyn = lay.ring + ring_elems;
  lay.asym1 = lay.wsyn + N;
  lay.asym2 = lay.asym1 + N;
  lay.extra = basep + fixed_al;
  SI_HIP(hipMemcpyAsync(lay.wsyn, wsyn.data(), N * sizeof(T), hipMemcpyHostToDevice, pl.stream));
  SI_HIP(hipMemcpyAsync(lay.asym1, a1.data(), N * sizeof(T), hipMemcpyHostToDevice, pl.stream));
  SI_HIP(hipMemcpyAsync(lay.asym2, a2.data(), N * sizeof(T), hipMemcpyHostToDevice, pl.stream));
  SI_HIP(hipStreamSynchronize(pl.stream));    // the host vectors go out of scope
  return SPECINV_OK;
}

// launch geometry of k_rtisi: as many look-ahead frames in flight as LDS (2 FFT buffers per group) and threads
// (>= 128 per group) allow
template <typename P, typename T>
int rtisi_generic_launch(P& pl, RtisiArgs<T>& r) {
  const int N = pl.N(), hop = pl.cfg.hop_length, la = r.la;
  const int threads = N >= 2048 ? 1024 : (N >= 1024 ? 512 : 256);
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
  hipLaunchKernelGGL((k_rtisi<T>), dim3(pl.B()), dim3(threads), lds, pl.stream, r);
  SI_HIP(hipGetLastError());
  return SPECINV_OK;
}

template <typename P, typename T>
int rtisi_launch(P& pl, const T* mag_user, int look_ahead, int asym, int max_iter, double alpha, T* x_out) {
  SI_CHECK(mag_user && x_out, SPECINV_EINVAL, "null pointer");
  SI_CHECK(max_iter > 0, SPECINV_EINVAL, "max_iter must be > 0");     // methods.py:295
  SI_CHECK(alpha >= 0, SPECINV_EINVAL, "alpha must be >= 0");        // methods.py:296
  const int F = pl.n_freq, Tn = pl.Tn();
  RtisiLayout<T> lay;
  SI_TRY(rtisi_prepare(pl, look_ahead, pl.rt_state, 0, lay));
  const int la = lay.la;

  if constexpr (std::is_same<T, float>::value) {
    bool used = false;
    SI_TRY(rtisi_fast_launch(pl, mag_user, la, asym, max_iter, alpha, x_out, lay.wsyn, lay.asym1, lay.asym2, &used));
    if (used) return SPECINV_OK;
  }

  SI_TRY(pl.mag.reserve(pl.nspec() * sizeof(T)));
  SI_TRY((pl.template transpose<T>(mag_user, pl.mag.template as<T>(), F, Tn)));
  SI_TRY(pl.frames_needed());

  RtisiArgs<T> r{};
  r.c = pl.fc;
  r.mag = pl.mag.template as<T>();
  r.ring = lay.ring;
  r.pre = lay.pre;
  r.frames_out = pl.frames.template as<T>();
  r.wsyn = lay.wsyn;
  r.asym1 = lay.asym1;
  r.asym2 = lay.asym2;
  r.keep = lay.keep;
  r.la = la;
  r.steps = Tn;
  r.max_iter = max_iter;
  r.asym = asym ? 1 : 0;
  r.lr = (T)(alpha / (1.0 + alpha));                                  // methods.py:360
  r.i_begin = 0;
  r.i_end = Tn + la;
  r.resume = 0;
  r.n_valid = Tn;
  SI_TRY(rtisi_generic_launch(pl, r));
  return pl.launch_ola(pl.frames.template as<T>(), x_out, true);     // methods.py:406-408
}

// ---- gradient w.r.t. the magnitudes (the reference's result is differentiable: test/test_rtisila.py:58-70) ---------
// Reverse sweep over the recorded run: one workgroup per item walks the (steps+la)*max_iter inner steps backwards.
// Cotangents: g_ring (every ring frame), g_pre (double-buffered like pre), gx (the overlap-added look-ahead span, LDS).
template <typename T>
struct RtisiAdjArgs {
  FrameCfg<T> c;
  const T* mag;          // (B, T, F)
  const cplx<T>* rec;    // recorded pre-projection spectra
  const T* g_x;          // (B, L) cotangent of the waveform
  const T* env;          // (L,)
  T* g_ring;             // (B, K+LA+1, N)
  cplx<T>* g_pre;        // (B, 2, LA+1, F)
  T* gmag;               // (B, T, F), zero on entry
  const T *wsyn, *asym1, *asym2;
  int keep, la, steps, max_iter, asym;
  T lr;
};

template <typename T>
__global__ void k_rtisi_adjoint(RtisiAdjArgs<T> r) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const FrameCfg<T>& c = r.c;
  const int N = c.n_fft, F = c.n_freq, hop = c.hop;
  cplx<T>* pa = reinterpret_cast<cplx<T>*>(smem);
  cplx<T>* pb = pa + N;
  T* gxb = reinterpret_cast<T*>(pb + N);            // la*hop + N
  const int bi = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const int nslots = r.keep + r.la + 1, xlen = r.la * hop + N;
  T* g_ring = r.g_ring + (int64_t)bi * nslots * N;
  cplx<T>* g_pre = r.g_pre + (int64_t)bi * 2 * (r.la + 1) * F;
  const T* mag = r.mag + (int64_t)bi * r.steps * F;
  T* gmag = r.gmag + (int64_t)bi * r.steps * F;
  const T* g_x = r.g_x + (int64_t)bi * c.length;
  const cplx<T>* rec = r.rec + (int64_t)bi * (r.steps + r.la) * r.max_iter * (r.la + 1) * F;

  for (int i = tid; i < nslots * N; i += nt) g_ring[i] = T(0);
  for (int i = tid; i < 2 * (r.la + 1) * F; i += nt) g_pre[i] = mk<T>(T(0), T(0));
  __syncthreads();

  for (int i = r.steps + r.la - 1; i >= 0; --i) {
    const int base = i % nslots;
    // the slot that became the next step's (zero) newest frame was this step's oldest kept frame: nothing flows back
    for (int k = tid; k < N; k += nt) g_ring[(int64_t)base * N + k] = T(0);
    if (i >= r.la) {                                 // commit: frames_out[i-la] = ring[keep] * window, then k_ola / env
      int s0 = base + r.keep;
      if (s0 >= nslots) s0 -= nslots;
      const int64_t off = (int64_t)(i - r.la) * hop - c.pad;
      for (int k = tid; k < N; k += nt) {
        const int64_t n = off + k;
        if (n >= 0 && n < c.length) g_ring[(int64_t)s0 * N + k] += g_x[n] / r.env[n] * c.window[k];
      }
    }
    __syncthreads();
    for (int j = r.max_iter - 1; j >= 0; --j) {
      const int64_t sidx = (int64_t)i * r.max_iter + j;
      const int pin = (int)(sidx & 1);
      cplx<T>* gp_out = g_pre + (int64_t)(pin ^ 1) * (r.la + 1) * F;   // cotangent of what this inner step wrote
      cplx<T>* gp_in = g_pre + (int64_t)pin * (r.la + 1) * F;          // ... of what it read
      for (int n = tid; n < xlen; n += nt) gxb[n] = T(0);
      if (j == 0)
        for (int f = tid; f < F; f += nt) gp_in[f] = mk<T>(T(0), T(0));   // slot 0 of the shifted momentum is unused
      __syncthreads();
      for (int q = 0; q <= r.la; ++q) {
        int slot = base + r.keep + q;
        if (slot >= nslots) slot -= nslots;
        T* gfr = g_ring + (int64_t)slot * N;
        // ring[slot] = inv_scale * Re IDFT(Hermitian(Y))   =>   gY = inv_scale * (interior ? 2 : Re) DFT(g)
        for (int k = tid; k < N; k += nt) {
          pa[k] = mk<T>(gfr[k], T(0));
          gfr[k] = T(0);                              // the frame was overwritten: its old value only fed the overlap-add
        }
        __syncthreads();
        lds_fft(pa, pb, c, false);
        const int tt = i + q - r.la;
        const bool valid = tt >= 0 && tt < r.steps;
        const cplx<T>* srow = rec + (sidx * (r.la + 1) + q) * F;
        for (int f = tid; f < F; f += nt) {
          cplx<T> gy = pa[f];
          if (c.onesided) {
            if (f == 0 || 2 * f == N) gy = mk<T>(gy.x * c.inv_scale, T(0));
            else gy = mk<T>(gy.x * (2 * c.inv_scale), gy.y * (2 * c.inv_scale));
          } else {
            gy = mk<T>(gy.x * c.inv_scale, gy.y * c.inv_scale);
          }
          // Y = S m / (|S| + eps)   (methods.py:394-396)
          const cplx<T> sv = srow[f];
          const T m = valid ? mag[(int64_t)tt * F + f] : T(0);
          const T ab = si_hypot(sv.x, sv.y);
          const T d = ab + eps16<T>::value;
          const T dot = gy.x * sv.x + gy.y * sv.y;
          const T c1 = m / d;
          const T c2 = ab > T(0) ? dot * m / (d * d * ab) : T(0);
          cplx<T> gs = mk<T>(gy.x * c1 - sv.x * c2, gy.y * c1 - sv.y * c2);
          if (valid) gmag[(int64_t)tt * F + f] += dot / d;
          // pre_out[q] = S                                  (:392)
          const cplx<T> gpo = gp_out[(int64_t)q * F + f];
          gs = mk<T>(gs.x + gpo.x, gs.y + gpo.y);
          // S = R - lr * pre_in[q] (j > 0)  |  R - lr * pre_in[q+1] (j == 0, i > 0, q < la)   (:387-391)
          if (j) gp_in[(int64_t)q * F + f] = mk<T>(-r.lr * gs.x, -r.lr * gs.y);
          else if (q < r.la) gp_in[(int64_t)(q + 1) * F + f] = i ? mk<T>(-r.lr * gs.x, -r.lr * gs.y) : mk<T>(T(0), T(0));
          // R = fwd_scale * DFT(frame): stored bins only -> halve the interior ones before the Hermitian inverse
          if (c.onesided && f != 0 && 2 * f != N) gs = mk<T>(gs.x * T(0.5), gs.y * T(0.5));
          pa[f] = gs;
        }
        __syncthreads();
        if (c.onesided) {
          for (int f = tid; f <= N / 2; f += nt) {
            const cplx<T> v = pa[f];
            if (f == 0 || 2 * f == N) pa[f] = mk<T>(v.x, T(0));
            else pa[N - f] = conj(v);
          }
          __syncthreads();
        }
        lds_fft(pa, pb, c, true);
        const T* win = (r.asym && q == r.la) ? (j ? r.asym2 : r.asym1) : c.window;
        for (int k = tid; k < N; k += nt) gxb[q * hop + k] += pa[k].x * c.fwd_scale * win[k];
        __syncthreads();
      }
      // x[n] = sum_f ring[f][n - f*hop] * wsyn[...]  over every slot, n in [keep*hop, keep*hop + xlen)
      for (int e = tid; e < nslots * N; e += nt) {
        const int f = e / N, k = e - f * N;
        const int np = f * hop + k - r.keep * hop;
        if (np >= 0 && np < xlen) {
          int slot = base + f;
          if (slot >= nslots) slot -= nslots;
          g_ring[(int64_t)slot * N + k] += gxb[np] * r.wsyn[k];
        }
      }
      __syncthreads();
    }
  }
  // initial state: newest slot = inv_scale * Re IDFT(Hermitian(mag[0] + 0j))   (methods.py:353-358)
  {
    const T* gfr = g_ring + (int64_t)(nslots - 1) * N;
    for (int k = tid; k < N; k += nt) pa[k] = mk<T>(gfr[k], T(0));
    __syncthreads();
    lds_fft(pa, pb, c, false);
    for (int f = tid; f < F; f += nt) {
      const T sc = (c.onesided && f != 0 && 2 * f != N) ? 2 * c.inv_scale : c.inv_scale;
      gmag[f] += pa[f].x * sc;
    }
  }
}

template <typename T>
inline int64_t rtisi_record_elems(int batch, int n_frames, int n_freq, int n_fft, int hop, int look_ahead, int max_iter) {
  const int keep = (n_fft - 1) / hop;
  const int la = look_ahead < 0 ? keep : look_ahead;
  return (int64_t)batch * (n_frames + la) * max_iter * (la + 1) * n_freq;
}

// whole-signal run on the generic kernel that also records what the adjoint needs
template <typename P, typename T>
int rtisi_launch_recorded(P& pl, const T* mag_user, int look_ahead, int asym, int max_iter, double alpha, T* x_out,
                          cplx<T>* rec) {
  SI_CHECK(mag_user && x_out && rec, SPECINV_EINVAL, "null pointer");
  SI_CHECK(max_iter > 0, SPECINV_EINVAL, "max_iter must be > 0");
  SI_CHECK(alpha >= 0, SPECINV_EINVAL, "alpha must be >= 0");
  const int F = pl.n_freq, Tn = pl.Tn();
  RtisiLayout<T> lay;
  SI_TRY(rtisi_prepare(pl, look_ahead, pl.rt_state, 0, lay));
  SI_TRY(pl.mag.reserve(pl.nspec() * sizeof(T)));
  SI_TRY((pl.template transpose<T>(mag_user, pl.mag.template as<T>(), F, Tn)));
  SI_TRY(pl.frames_needed());
  RtisiArgs<T> r{};
  r.c = pl.fc;
  r.mag = pl.mag.template as<T>();
  r.ring = lay.ring;
  r.pre = lay.pre;
  r.frames_out = pl.frames.template as<T>();
  r.wsyn = lay.wsyn;
  r.asym1 = lay.asym1;
  r.asym2 = lay.asym2;
  r.keep = lay.keep;
  r.la = lay.la;
  r.steps = Tn;
  r.max_iter = max_iter;
  r.asym = asym ? 1 : 0;
  r.lr = (T)(alpha / (1.0 + alpha));
  r.i_begin = 0;
  r.i_end = Tn + lay.la;
  r.n_valid = Tn;
  r.rec = rec;
  SI_TRY(rtisi_generic_launch(pl, r));
  return pl.launch_ola(pl.frames.template as<T>(), x_out, true);
}

template <typename P, typename T>
int rtisi_adjoint_launch(P& pl, const T* mag_user, const cplx<T>* rec, const T* g_x, int look_ahead, int asym, int max_iter,
                         double alpha, T* gmag_user) {
  SI_CHECK(mag_user && rec && g_x && gmag_user, SPECINV_EINVAL, "null pointer");
  SI_CHECK(max_iter > 0 && alpha >= 0, SPECINV_EINVAL, "bad max_iter / alpha");
  const int N = pl.N(), hop = pl.cfg.hop_length, F = pl.n_freq, Tn = pl.Tn();
  RtisiLayout<T> lay;     // same layout as the forward: ring -> g_ring, pre -> g_pre
  const size_t gm_bytes = (size_t)pl.nspec() * sizeof(T);
  SI_TRY(rtisi_prepare(pl, look_ahead, pl.rt_state, gm_bytes, lay));
  SI_TRY(pl.mag.reserve(pl.nspec() * sizeof(T)));
  SI_TRY((pl.template transpose<T>(mag_user, pl.mag.template as<T>(), F, Tn)));
  T* gm = reinterpret_cast<T*>(lay.extra);
  SI_HIP(hipMemsetAsync(gm, 0, gm_bytes, pl.stream));
  RtisiAdjArgs<T> r{};
  r.c = pl.fc;
  r.mag = pl.mag.template as<T>();
  r.rec = rec;
  r.g_x = g_x;
  r.env = pl.env.template as<T>();
  r.g_ring = lay.ring;
  r.g_pre = lay.pre;
  r.gmag = gm;
  r.wsyn = lay.wsyn;
  r.asym1 = lay.asym1;
  r.asym2 = lay.asym2;
  r.keep = lay.keep;
  r.la = lay.la;
  r.steps = Tn;
  r.max_iter = max_iter;
  r.asym = asym ? 1 : 0;
  r.lr = (T)(alpha / (1.0 + alpha));
  const int threads = N >= 2048 ? 1024 : (N >= 1024 ? 512 : 256);
  const size_t lds = 2 * (size_t)N * sizeof(cplx<T>) + ((size_t)lay.la * hop + N) * sizeof(T);
  SI_CHECK(lds <= 160 * 1024 - 512, SPECINV_EUNSUPPORTED, "RTISI_LA adjoint: n_fft=%d look_ahead=%d needs %zu bytes of LDS", N,
           lay.la, lds);
  SI_HIP(hipFuncSetAttribute((const void*)k_rtisi_adjoint<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL((k_rtisi_adjoint<T>), dim3(pl.B()), dim3(threads), lds, pl.stream, r);
  SI_HIP(hipGetLastError());
  return pl.template transpose<T>(gm, gmag_user, Tn, F);
}

// ---- streaming: the same recursion fed a few frames at a time -------------------------------------------
// Frame c is committed at step c + la; once it is, the padded samples [c*hop, (c+1)*hop) are final (frame c+1
// starts at (c+1)*hop).  `commits` holds the last keep+1+cap committed frames (times the window).  The sums run
// in the order of k_ola and the envelope like PlanT::setup (products in T, sum in double), so the concatenated
// output equals the whole-signal run of the same kernel bit for bit.
template <typename T>
__global__ void k_rtisi_emit(const T* __restrict__ commits, int out_ring, const T* __restrict__ window, int n_fft,
                             int hop, int64_t np0, int64_t np1, int64_t c_last, T* __restrict__ out, int64_t out_stride) {
#pragma clang fp contract(off)   // the envelope sums rounded products, like the host loop of PlanT::setup
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= np1 - np0) return;
  const int bi = blockIdx.y;
  const int64_t np = np0 + i;
  int64_t t_hi = np / hop;
  if (t_hi > c_last) t_hi = c_last;
  const int64_t t_lo = np - n_fft + 1 <= 0 ? 0 : (np - n_fft + hop) / hop;
  const T* fr = commits + (int64_t)bi * out_ring * n_fft;
  T acc = 0;
  double env = 0;
  for (int64_t t = t_lo; t <= t_hi; ++t) {
    const int k = (int)(np - t * hop);
    acc += fr[(t % out_ring) * n_fft + k];
    const T w2 = window[k] * window[k];
    env += (double)w2;
  }
  out[(int64_t)bi * out_stride + i] = acc / (T)env;
}

// scatter (B, F, k) user frames into the (B, mag_ring, F) ring at frames t0 .. t0+k-1
template <typename T>
__global__ void k_rtisi_store_mag(const T* __restrict__ user, int F, int k, T* __restrict__ ring, int mag_ring, int64_t t0) {
  __shared__ T tile[32][33];
  const int bi = blockIdx.z;
  const int f0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
  for (int r = threadIdx.y; r < 32; r += blockDim.y) {
    const int f = f0 + r, j = j0 + threadIdx.x;
    if (f < F && j < k) tile[r][threadIdx.x] = user[((int64_t)bi * F + f) * k + j];
  }
  __syncthreads();
  for (int r = threadIdx.y; r < 32; r += blockDim.y) {
    const int j = j0 + r, f = f0 + threadIdx.x;
    if (f < F && j < k) ring[((int64_t)bi * mag_ring + (t0 + j) % mag_ring) * F + f] = tile[threadIdx.x][r];
  }
}

template <typename T>
struct RtisiStream {
  bool active = false, flushed = false;
  RtisiLayout<T> lay;
  int asym = 0, max_iter = 0, mag_ring = 0, out_ring = 0;
  T lr = 0;
  int64_t steps_done = 0, n_pushed = 0, emitted = 0;   // emitted: next padded sample to hand out
  T *mags = nullptr, *commits = nullptr;
  // wave-level kernel (float32 shapes it covers): targets in pair layout, its frame ring / registers saved between launches
  bool fast = false;
  RtisiFastPick pick;
  float* fstate = nullptr;
  fast::v4f* mpairs = nullptr;
  float* mmid = nullptr;
};

template <typename P, typename T>
int rtisi_stream_begin(P& pl, RtisiStream<T>& st, int look_ahead, int asym, int max_iter, double alpha) {
  SI_CHECK(max_iter > 0, SPECINV_EINVAL, "max_iter must be > 0");
  SI_CHECK(alpha >= 0, SPECINV_EINVAL, "alpha must be >= 0");
  const int N = pl.N(), hop = pl.cfg.hop_length, F = pl.n_freq, cap = pl.Tn(), Bn = pl.B();
  const int keep = (N - 1) / hop;
  const int la = look_ahead < 0 ? keep : look_ahead;
  st.mag_ring = la + 1 + cap;
  st.out_ring = keep + 1 + cap;
  const size_t mag_bytes = ((size_t)Bn * st.mag_ring * F * sizeof(T) + 63) & ~(size_t)63;
  const size_t out_bytes = ((size_t)Bn * st.out_ring * N * sizeof(T) + 63) & ~(size_t)63;
  st.fast = false;
  size_t fast_bytes = 0, state_bytes = 0, pairs_bytes = 0, mid_bytes = 0;
  if constexpr (std::is_same<T, float>::value) {
    st.pick = rtisi_fast_pick(pl, la);
    if (st.pick.fn != nullptr) {
      st.fast = true;
      const int Rr = st.pick.R, nslots = keep + la + 1, Hh = Rr / 2;
      const size_t per_item = (size_t)nslots * (64 * Rr) + (size_t)(la + 1) * (2 * (Hh * 2 * 64) + 2 * 64);   // rtisi_state_v2f
      state_bytes = ((size_t)Bn * per_item * sizeof(fast::v2f) + 63) & ~(size_t)63;
      pairs_bytes = ((size_t)Bn * st.mag_ring * (Hh / 2) * 64 * sizeof(fast::v4f) + 63) & ~(size_t)63;
      mid_bytes = ((size_t)Bn * st.mag_ring * sizeof(float) + 63) & ~(size_t)63;
      fast_bytes = state_bytes + pairs_bytes + mid_bytes;
    }
  }
  SI_TRY(rtisi_prepare(pl, look_ahead, pl.rs_state, mag_bytes + out_bytes + fast_bytes, st.lay));
  st.mags = reinterpret_cast<T*>(st.lay.extra);
  st.commits = reinterpret_cast<T*>(st.lay.extra + mag_bytes);
  if (st.fast) {
    char* fb = st.lay.extra + mag_bytes + out_bytes;
    st.fstate = reinterpret_cast<float*>(fb);
    st.mpairs = reinterpret_cast<fast::v4f*>(fb + state_bytes);
    st.mmid = reinterpret_cast<float*>(fb + state_bytes + pairs_bytes);
  }
  st.asym = asym ? 1 : 0;
  st.max_iter = max_iter;
  st.lr = (T)(alpha / (1.0 + alpha));
  st.steps_done = st.n_pushed = 0;
  st.emitted = pl.pad;                          // centre trimming: the first pad samples are never handed out
  st.active = true;
  st.flushed = false;
  return SPECINV_OK;
}

template <typename P, typename T>
int rtisi_stream_steps(P& pl, RtisiStream<T>& st, int n_steps) {
  if constexpr (std::is_same<T, float>::value) {
    if (st.fast) {
      fast::RtisiFastArgs a{};
      a.m_pairs = st.mpairs;
      a.m_mid = st.mmid;
      a.frames_out = st.commits;
      a.window = pl.window.template as<float>();
      a.wsyn = st.lay.wsyn;
      a.asym1 = st.lay.asym1;
      a.asym2 = st.lay.asym2;
      a.T = 0;
      a.la = st.lay.la;
      a.max_iter = st.max_iter;
      a.asym = st.asym;
      a.lr = st.lr;
      a.fwd_scale = pl.fc.fwd_scale;
      a.inv_scale = pl.fc.inv_scale;
      a.i_begin = (int)st.steps_done;
      a.i_end = (int)st.steps_done + n_steps;
      a.resume = st.steps_done > 0;
      a.n_valid = (int)st.n_pushed;
      a.mag_ring = st.mag_ring;
      a.out_ring = st.out_ring;
      a.state = st.fstate;
      SI_HIP(hipFuncSetAttribute(st.pick.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)st.pick.lds));
      void* kargs[] = {&a};
      SI_HIP(hipLaunchKernel(st.pick.fn, dim3(pl.B()), dim3(st.pick.threads), kargs, st.pick.lds, pl.stream));
      st.steps_done += n_steps;
      return SPECINV_OK;
    }
  }
  RtisiArgs<T> r{};
  r.c = pl.fc;
  r.mag = st.mags;
  r.ring = st.lay.ring;
  r.pre = st.lay.pre;
  r.frames_out = st.commits;
  r.wsyn = st.lay.wsyn;
  r.asym1 = st.lay.asym1;
  r.asym2 = st.lay.asym2;
  r.keep = st.lay.keep;
  r.la = st.lay.la;
  r.steps = 0;
  r.max_iter = st.max_iter;
  r.asym = st.asym;
  r.lr = st.lr;
  r.i_begin = (int)st.steps_done;
  r.i_end = (int)st.steps_done + n_steps;
  r.resume = st.steps_done > 0;
  r.n_valid = (int)st.n_pushed;
  r.mag_ring = st.mag_ring;
  r.out_ring = st.out_ring;
  SI_TRY(rtisi_generic_launch(pl, r));
  st.steps_done += n_steps;
  return SPECINV_OK;
}

// hand out every padded sample in [st.emitted, upto), appended at column *n_out of x_out
template <typename P, typename T>
int rtisi_stream_emit(P& pl, RtisiStream<T>& st, int64_t upto, T* x_out, int64_t out_stride, int64_t* n_out) {
  const int64_t c_last = st.steps_done - st.lay.la - 1;      // newest committed frame
  const int64_t n = upto - st.emitted;
  if (n <= 0) return SPECINV_OK;
  SI_CHECK(out_stride >= *n_out + n, SPECINV_EINVAL, "out_stride %lld < %lld samples of this call", (long long)out_stride,
           (long long)(*n_out + n));
  hipLaunchKernelGGL((k_rtisi_emit<T>), dim3((unsigned)ceil_div(n, 256), pl.B()), dim3(256), 0, pl.stream, st.commits,
                     st.out_ring, pl.window.template as<T>(), pl.N(), pl.cfg.hop_length, st.emitted, upto, c_last,
                     x_out + *n_out, out_stride);
  SI_HIP(hipGetLastError());
  st.emitted = upto;
  *n_out += n;
  return SPECINV_OK;
}

// samples that are final once `commits` frames are committed and `n_pushed` frames are known to exist
template <typename P, typename T>
int64_t rtisi_stream_final(const P& pl, const RtisiStream<T>& st) {
  const int64_t commits = st.steps_done - st.lay.la;                                 // frames 0 .. commits-1
  if (commits <= 0) return 0;
  const int64_t end = (st.n_pushed - 1) * pl.cfg.hop_length + pl.N() - pl.pad;       // right trim if the signal ended here
  return std::min<int64_t>(commits * pl.cfg.hop_length, end);
}

template <typename P, typename T>
int rtisi_stream_push(P& pl, RtisiStream<T>& st, const T* mag_user, int k, T* x_out, int64_t out_stride, int64_t* n_out) {
  SI_CHECK(st.active && !st.flushed, SPECINV_ESTATE, "rtisi_stream_push without rtisi_stream_begin (or after flush)");
  SI_CHECK(mag_user && x_out && n_out, SPECINV_EINVAL, "null pointer");
  SI_CHECK(k >= 1 && k <= pl.Tn(), SPECINV_EINVAL, "push of %d frames, the plan was made for at most %d", k, pl.Tn());
  SI_CHECK(st.steps_done + k < (int64_t)1 << 30, SPECINV_EUNSUPPORTED, "stream too long");
  *n_out = 0;
  const int F = pl.n_freq;
  bool stored = false;
  if constexpr (std::is_same<T, float>::value) {
    if (st.fast) {
      // (B, F, k) -> (B, k, F) -> pair layout at the ring rows of these frames
      SI_TRY(pl.mag.reserve((size_t)pl.B() * pl.Tn() * F * sizeof(float)));
      {
        dim3 grid((k + 31) / 32, (F + 31) / 32, pl.B());
        hipLaunchKernelGGL((k_transpose<float>), grid, dim3(32, 8), 0, pl.stream, mag_user, pl.mag.template as<float>(), F, k);
        SI_HIP(hipGetLastError());
      }
      const long long total = (long long)pl.B() * k * (st.pick.R / 4) * 64;
      SPECINV_R_SWITCH(st.pick.R, hipLaunchKernelGGL((fast::k_mag_to_pairs_ring<RR>), dim3((unsigned)ceil_div(total, 256)),
                                                     dim3(256), 0, pl.stream, pl.mag.template as<float>(), st.mpairs, st.mmid,
                                                     k, st.mag_ring, (long long)st.n_pushed, total));
      SI_HIP(hipGetLastError());
      stored = true;
    }
  }
  if (!stored) {
    hipLaunchKernelGGL((k_rtisi_store_mag<T>), dim3((k + 31) / 32, (F + 31) / 32, pl.B()), dim3(32, 8), 0, pl.stream, mag_user,
                       F, k, st.mags, st.mag_ring, st.n_pushed);
    SI_HIP(hipGetLastError());
  }
  st.n_pushed += k;
  SI_TRY(rtisi_stream_steps(pl, st, k));
  return rtisi_stream_emit(pl, st, rtisi_stream_final(pl, st), x_out, out_stride, n_out);
}

// end of the signal: la more steps (their new look-ahead slots see zero targets) and the tail of the last frames
template <typename P, typename T>
int rtisi_stream_flush(P& pl, RtisiStream<T>& st, T* x_out, int64_t out_stride, int64_t* n_out) {
  SI_CHECK(st.active && !st.flushed, SPECINV_ESTATE, "rtisi_stream_flush without rtisi_stream_begin (or twice)");
  SI_CHECK(x_out && n_out, SPECINV_EINVAL, "null pointer");
  SI_CHECK(st.n_pushed >= 1, SPECINV_ESTATE, "rtisi_stream_flush before any frame was pushed");
  *n_out = 0;
  for (int left = st.lay.la; left > 0;) {       // the ring of committed frames takes Tn() new frames per launch
    const int n = std::min(left, pl.Tn());
    SI_TRY(rtisi_stream_steps(pl, st, n));
    SI_TRY(rtisi_stream_emit(pl, st, rtisi_stream_final(pl, st), x_out, out_stride, n_out));
    left -= n;
  }
  st.flushed = true;
  const int64_t total = (st.n_pushed - 1) * pl.cfg.hop_length + pl.N() - pl.pad;
  return rtisi_stream_emit(pl, st, total, x_out, out_stride, n_out);
}

}  // namespace specinv
