// Generic (any n_fft / hop / pad mode / sidedness / f32+f64) kernels of libspecinv.
//
// One workgroup owns one frame: the frame's n_fft samples are staged in LDS, transformed by
// a Stockham autosort FFT whose radix list is a factorisation of n_fft (any prime factor is
// handled by a direct small DFT), updated in the frequency domain, transformed back and
// written out windowed; a second kernel overlap-adds the frames (gather form, so the sum
// order is fixed and the result is bitwise reproducible) and divides by the envelope.
// This is the coverage path; the measured path for the headline shapes is kernels_fast_td.h / kernels_fused.h on fast_core.h.
#pragma once
#include "common.h"

namespace specinv {

template <typename T>
struct FrameCfg {
  int n_fft, n_freq, n_frames, hop, pad, pad_mode, onesided;
  int64_t length;  // samples per batch row of the signal being read / written
  T fwd_scale;     // 1 or n_fft^-1/2 (torch.stft normalized=True)
  T inv_scale;     // 1/n_fft or n_fft^-1/2 (irfft norm backward / ortho)
  int n_stages;
  int radix[kMaxStages];
  // per stage s (ns = product of the radices before it): ceil(2^32 / ns) and ceil(2^32 / (ns * radix)) for the
  // division-free index split (exact for dividends < 2^16), and the twiddle stride n_fft / (ns * radix)
  unsigned ns_magic[kMaxStages], m_magic[kMaxStages];
  int tw_step[kMaxStages];
  const cplx<T>* tw;  // tw[n] = exp(-2 pi i n / n_fft)
  const T* window;    // n_fft
  // In-place transforms (k_stft, k_iter_pair, k_grad_frames): the kernel has ONE buffer of n_fft complex points in LDS instead of
  // two; a stage's threads hold their butterflies (IpMB<T, R> each at most) in registers across a barrier between the last read
  // and the first write.  Twice the workgroups per CU where LDS is what limits them (float64 at n_fft 2048: 2 -> 4), and n_fft up to
  // 16384 (float32) / 8192 (float64) in 128 KiB.  Radices 2, 3, 4, 5, 7, 8 only (a larger prime factor keeps the two-buffer form).
  // A compile-time property of the kernel (template parameter IP); these fields only record the plan's choice.
  int inplace, maxb;
  // Power-of-two n_fft (k_iter_pair_dr): decimation in frequency forward, decimation in time back, both in place with ONE barrier
  // per stage - a butterfly reads and writes the same positions - the spectrum staying in digit-reversed order in between.
  // Stage i (radix dr_radix[i] = 1 << dr_bits[i]) works on blocks of dr_m[i] * radix points; bin f sits at
  // sum_i digit_i(f) * dr_m[i], digit_i the i-th least significant digit of f in the mixed radix dr_radix[0], dr_radix[1], ...
  int dr_stages, dr_lg8;                             // stages; log2(n_fft / 8): the octant of the twiddle table (dr_tw)
  int tw_lds;                                        // k_iter_pair: a copy of `tw` (n_fft entries) sits behind its LDS buffers
  int dr_bits[kMaxStages], dr_shift[kMaxStages];     // log2 radix, log2 dr_m
};

// Padded LDS layout of the digit-reversed transform: one element of slack after every 32 - the strided accesses of the small-block
// stages (thread j at j * radix + q) and of the bin update (consecutive bins sit n_fft / radix apart) then spread over all banks
// (complex float64: 16 lanes per pass, the minimum; complex float32: 32)
__device__ __host__ __forceinline__ int dr_phys(int p) { return p + (p >> 5); }
template <typename T>
__device__ __forceinline__ int dr_pos(const FrameCfg<T>& c, int f) {
  int pos = 0;
#pragma unroll 1
  for (int i = 0; i < c.dr_stages; ++i) {
    pos += (f & ((1 << c.dr_bits[i]) - 1)) << c.dr_shift[i];
    f >>= c.dr_bits[i];
  }
  return pos;
}

// ---- signal access with torch.stft's centre padding (methods.py:241 -> F.pad) -----------
template <typename T>
__device__ inline T load_padded(const T* __restrict__ x, int64_t L, int64_t n, int pad_mode) {
  if (n >= 0 && n < L) return x[n];
  switch (pad_mode) {
    case SPECINV_PAD_REFLECT:
      n = n < 0 ? -n : 2 * (L - 1) - n;
      n = n < 0 ? 0 : (n >= L ? L - 1 : n);
      return x[n];
    case SPECINV_PAD_REPLICATE:
      return x[n < 0 ? 0 : L - 1];
    case SPECINV_PAD_CIRCULAR:
      n %= L;
      if (n < 0) n += L;
      return x[n];
    default:
      return T(0);
  }
}

// ---- Stockham FFT of n_fft complex points held in LDS ------------------------------------
// On entry `a` holds the input (all threads synchronised); on exit `a` points at the result
// (natural order, unscaled) and all threads are synchronised.
// `tid` / `nthr`: the calling thread's rank inside, and the size of, the thread group that owns the
// transform (default: the whole workgroup).  Every thread of the WORKGROUP must still reach the barriers.
//
// One thread owns one radix-R butterfly of a stage: R strided loads, R-1 twiddles W_m^(k q) from the
// table, an in-register DFT_R, R stores.  Radices 2, 3, 4, 5, 7, 8 are unrolled; a larger prime
// factor falls back to one output per thread (direct DFT_R from LDS).
// times -i (forward) / +i (inverse)
template <typename T, bool INV>
__device__ __forceinline__ cplx<T> rot_mi(cplx<T> a) {
  return INV ? mk<T>(-a.y, a.x) : mk<T>(a.y, -a.x);
}

template <typename T, bool INV>
__device__ __forceinline__ void bf4(cplx<T>& v0, cplx<T>& v1, cplx<T>& v2, cplx<T>& v3) {
  const cplx<T> s02 = v0 + v2, d02 = v0 - v2, s13 = v1 + v3, d13 = rot_mi<T, INV>(v1 - v3);
  v0 = s02 + s13;
  v2 = s02 - s13;
  v1 = d02 + d13;
  v3 = d02 - d13;
}

template <typename T, int R, bool INV>
struct Butterfly {   // generic small prime: X_r = sum_q v_q W_R^(q r), W_R^e = tw[e * N / R]
  static __device__ __forceinline__ void run(cplx<T> (&v)[R], const cplx<T>* __restrict__ tw, int N) {
    cplx<T> w[R];
#pragma unroll
    for (int e = 0; e < R; ++e) {
      w[e] = tw[e * (N / R)];
      if (INV) w[e].y = -w[e].y;
    }
    cplx<T> o[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      cplx<T> acc = v[0];
#pragma unroll
      for (int q = 1; q < R; ++q) acc = acc + cmul(v[q], w[(q * r) % R]);
      o[r] = acc;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = o[r];
  }
};
template <typename T, bool INV>
struct Butterfly<T, 2, INV> {
  static __device__ __forceinline__ void run(cplx<T> (&v)[2], const cplx<T>*, int) {
    const cplx<T> a = v[0] + v[1], b = v[0] - v[1];
    v[0] = a;
    v[1] = b;
  }
};
template <typename T, bool INV>
struct Butterfly<T, 3, INV> {
  static __device__ __forceinline__ void run(cplx<T> (&v)[3], const cplx<T>*, int) {
    const T h = T(0.86602540378443864676372317075294);   // sin(pi/3)
    const cplx<T> t = v[1] + v[2];
    const cplx<T> m = mk<T>(v[0].x - T(0.5) * t.x, v[0].y - T(0.5) * t.y);
    const cplx<T> d = v[1] - v[2];
    const cplx<T> s = rot_mi<T, INV>(mk<T>(h * d.x, h * d.y));
    v[0] = v[0] + t;
    v[1] = m + s;
    v[2] = m - s;
  }
};
template <typename T, bool INV>
struct Butterfly<T, 4, INV> {
  static __device__ __forceinline__ void run(cplx<T> (&v)[4], const cplx<T>*, int) { bf4<T, INV>(v[0], v[1], v[2], v[3]); }
};
template <typename T, bool INV>
struct Butterfly<T, 8, INV> {
  static __device__ __forceinline__ void run(cplx<T> (&v)[8], const cplx<T>*, int) {
    const T h = T(0.70710678118654752440084436210485);
    bf4<T, INV>(v[0], v[2], v[4], v[6]);   // even samples -> E0..E3 in v0, v2, v4, v6
    bf4<T, INV>(v[1], v[3], v[5], v[7]);   // odd samples  -> O0..O3 in v1, v3, v5, v7
    // O_k *= W8^k (conjugated for the inverse)
    const cplx<T> o1 = INV ? mk<T>(h * (v[3].x - v[3].y), h * (v[3].x + v[3].y)) : mk<T>(h * (v[3].x + v[3].y), h * (v[3].y - v[3].x));
    const cplx<T> o2 = rot_mi<T, INV>(v[5]);
    const cplx<T> o3 = INV ? mk<T>(-h * (v[7].x + v[7].y), h * (v[7].x - v[7].y)) : mk<T>(h * (v[7].y - v[7].x), -h * (v[7].x + v[7].y));
    const cplx<T> e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6], o0 = v[1];
    v[0] = e0 + o0;
    v[4] = e0 - o0;
    v[1] = e1 + o1;
    v[5] = e1 - o1;
    v[2] = e2 + o2;
    v[6] = e2 - o2;
    v[3] = e3 + o3;
    v[7] = e3 - o3;
  }
};

// j / d for 0 <= j < 2^16, d >= 1, magic = ceil(2^32 / d) (d = 1: magic wraps to 0 and is not used)
__device__ __forceinline__ int div_magic(int j, int d, unsigned magic) {
  return d == 1 ? j : (int)__umulhi((unsigned)j, magic);
}

template <typename T, int R, bool INV>
__device__ __forceinline__ void fft_stage(const cplx<T>* __restrict__ a, cplx<T>* __restrict__ b, const FrameCfg<T>& c,
                                          int s, int ns, int tid, int nthr, const cplx<T>* __restrict__ tw) {
  const int N = c.n_fft, nb = (unsigned)N / (unsigned)R, tws = c.tw_step[s];
  const unsigned magic = c.ns_magic[s];
  for (int j = tid; j < nb; j += nthr) {
    const int blk = div_magic(j, ns, magic), k = j - blk * ns;
    cplx<T> v[R];
#pragma unroll
    for (int q = 0; q < R; ++q) v[q] = a[j + q * nb];
    if (ns > 1) {
      const int kt = k * tws;                 // W_m^(k q) = tw[k q N/m], k q < m
#pragma unroll
      for (int q = 1; q < R; ++q) {
        cplx<T> w = tw[kt * q];
        if (INV) w.y = -w.y;
        v[q] = cmul(v[q], w);
      }
    }
    Butterfly<T, R, INV>::run(v, tw, N);
    const int base = blk * ns * R + k;
#pragma unroll
    for (int r = 0; r < R; ++r) b[base + r * ns] = v[r];
  }
}

// any radix: one output per thread, direct DFT_R
template <typename T>
__device__ inline void fft_stage_any(const cplx<T>* __restrict__ a, cplx<T>* __restrict__ b, const FrameCfg<T>& c, int R,
                                     int s, int ns, bool inverse, int tid, int nthr, const cplx<T>* __restrict__ tw) {
  const int N = c.n_fft, m = ns * R, stride = N / R, twstep = c.tw_step[s];
  const unsigned magic_m = c.m_magic[s], magic_ns = c.ns_magic[s];
  for (int o = tid; o < N; o += nthr) {
    const int block = div_magic(o, m, magic_m);
    const int within = o - block * m;
    const int r = div_magic(within, ns, magic_ns);
    const int k = within - r * ns;
    const int j = block * ns + k;
    const int e1 = k + r * ns;  // < m
    T accx = 0, accy = 0;
    int e = 0;
    for (int q = 0; q < R; ++q) {
      const cplx<T> v = a[j + q * stride];
      cplx<T> w = tw[e * twstep];
      if (inverse) w.y = -w.y;
      accx += v.x * w.x - v.y * w.y;
      accy += v.x * w.y + v.y * w.x;
      e += e1;
      if (e >= m) e -= m;
    }
    b[o] = mk<T>(accx, accy);
  }
}

// One stage of the transform IN PLACE: every butterfly of the stage is read and computed, then - after a barrier - written to the
// Stockham positions of the same buffer.  Thread `tid` owns butterflies tid, tid + nthr, ... (at most MB of them).
template <typename T, int R, bool INV, int MB>
__device__ __forceinline__ void fft_stage_inplace(cplx<T>* __restrict__ a, const FrameCfg<T>& c, int s, int ns, int tid, int nthr,
                                                  const cplx<T>* __restrict__ tw) {
  const int N = c.n_fft, nb = (unsigned)N / (unsigned)R, tws = c.tw_step[s];
  const unsigned magic = c.ns_magic[s];
  cplx<T> v[MB][R];
  int base[MB];
#pragma unroll
  for (int it = 0; it < MB; ++it) {
    const int j = tid + it * nthr;
    base[it] = -1;
    if (j < nb) {
      const int blk = div_magic(j, ns, magic), k = j - blk * ns;
#pragma unroll
      for (int q = 0; q < R; ++q) v[it][q] = a[j + q * nb];
      if (ns > 1) {
        const int kt = k * tws;
#pragma unroll
        for (int q = 1; q < R; ++q) {
          cplx<T> w = tw[kt * q];
          if (INV) w.y = -w.y;
          v[it][q] = cmul(v[it][q], w);
        }
      }
      Butterfly<T, R, INV>::run(v[it], tw, N);
      base[it] = blk * ns * R + k;
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < MB; ++it) {
    if (base[it] >= 0) {
#pragma unroll
      for (int r = 0; r < R; ++r) a[base[it] + r * ns] = v[it][r];
    }
  }
}

// Butterflies a thread may hold across the barrier, by radix: 16 complex values in float32, 8 in float64 (their registers), at
// most 4.  The host sizes the workgroup so that every stage fits (ip_butterflies_per_thread, plan_impl.h).
template <typename T, int R>
struct IpMB {
  static constexpr int budget = sizeof(T) == 8 ? 8 : 16;
  static constexpr int value = (R == 5 || R == 7) ? 1 : (budget / R >= 4 ? 4 : (budget / R >= 1 ? budget / R : 1));
};
inline int ip_butterflies_per_thread(int radix, bool f64) {          // (the same rule for the host, plan_impl.h)
  const int budget = f64 ? 8 : 16;
  if (radix == 5 || radix == 7) return 1;                            // (the odd radices' butterflies carry their own tables)
  return budget / radix >= 4 ? 4 : (budget / radix >= 1 ? budget / radix : 1);
}

template <typename T, bool INV>
__device__ inline void lds_fft_inplace(cplx<T>* a, const FrameCfg<T>& c, int tid, int nthr, const cplx<T>* tw) {
  int ns = 1;
  for (int s = 0; s < c.n_stages; ++s) {
    const int R = c.radix[s];
    switch (R) {
      case 2: fft_stage_inplace<T, 2, INV, IpMB<T, 2>::value>(a, c, s, ns, tid, nthr, tw); break;
      case 3: fft_stage_inplace<T, 3, INV, IpMB<T, 3>::value>(a, c, s, ns, tid, nthr, tw); break;
      case 4: fft_stage_inplace<T, 4, INV, IpMB<T, 4>::value>(a, c, s, ns, tid, nthr, tw); break;
      case 5: fft_stage_inplace<T, 5, INV, IpMB<T, 5>::value>(a, c, s, ns, tid, nthr, tw); break;
      case 7: fft_stage_inplace<T, 7, INV, IpMB<T, 7>::value>(a, c, s, ns, tid, nthr, tw); break;
      default: fft_stage_inplace<T, 8, INV, IpMB<T, 8>::value>(a, c, s, ns, tid, nthr, tw); break;
    }
    __syncthreads();
    ns *= R;
  }
}

template <typename T, bool INV>
__device__ inline void lds_fft_dir(cplx<T>*& a, cplx<T>*& b, const FrameCfg<T>& c, int tid, int nthr, const cplx<T>* tw) {
  int ns = 1;
  for (int s = 0; s < c.n_stages; ++s) {
    const int R = c.radix[s];
    switch (R) {
      case 2: fft_stage<T, 2, INV>(a, b, c, s, ns, tid, nthr, tw); break;
      case 3: fft_stage<T, 3, INV>(a, b, c, s, ns, tid, nthr, tw); break;
      case 4: fft_stage<T, 4, INV>(a, b, c, s, ns, tid, nthr, tw); break;
      case 5: fft_stage<T, 5, INV>(a, b, c, s, ns, tid, nthr, tw); break;
      case 7: fft_stage<T, 7, INV>(a, b, c, s, ns, tid, nthr, tw); break;
      case 8: fft_stage<T, 8, INV>(a, b, c, s, ns, tid, nthr, tw); break;
      default: fft_stage_any<T>(a, b, c, R, s, ns, INV, tid, nthr, tw); break;
    }
    __syncthreads();
    cplx<T>* tmp = a;
    a = b;
    b = tmp;
    ns *= R;
  }
}

// IP: the in-place form (one buffer: b == a, the result stays where the input was) - a compile-time choice, so that a kernel
// carries the registers of one form only
template <typename T, bool IP = false>
__device__ inline void lds_fft(cplx<T>*& a, cplx<T>*& b, const FrameCfg<T>& c, bool inverse, int tid = -1,
                               int nthr = 0, const cplx<T>* tw = nullptr) {      // tw: the twiddle table to read (default: c.tw)
  if (tid < 0) {
    tid = threadIdx.x;
    nthr = blockDim.x;
  }
  if (tw == nullptr) tw = c.tw;
  if constexpr (IP) {
    if (inverse) lds_fft_inplace<T, true>(a, c, tid, nthr, tw);
    else lds_fft_inplace<T, false>(a, c, tid, nthr, tw);
  } else {
    if (inverse) lds_fft_dir<T, true>(a, b, c, tid, nthr, tw);
    else lds_fft_dir<T, false>(a, b, c, tid, nthr, tw);
  }
}

// ---- the digit-reversed in-place transform (power-of-two n_fft; FrameCfg::dr_*) -----------------------------------------------------
// Twiddles of the digit-reversed transform from an OCTANT table in LDS: tab[j] = (cos, sin)(2 pi j / N), 0 <= j <= N / 8, the
// other seven octants by swaps and signs.  (From the global table every stage waited an L2 round trip for its R - 1 twiddles:
// float64 n_fft 2048 0.416 -> 0.355 ms per iteration with the loads stubbed out - round 5, tools/log/EXPERIMENTS.md.)
template <typename T>
__device__ __forceinline__ cplx<T> dr_tw(const cplx<T>* __restrict__ tab, int lg8, int i) {
  const int n8 = 1 << lg8, o = i >> lg8, jp = i & (n8 - 1);
  const cplx<T> t = tab[(o & 1) ? n8 - jp : jp];
  const bool sw = ((o + 1) >> 1) & 1;                  // octants 1, 2, 5, 6: cosine and sine trade places
  const T ca = sw ? t.y : t.x, sa = sw ? t.x : t.y;    // |cos|, |sin| of the angle
  return mk<T>((((o + 2) >> 2) & 1) ? -ca : ca, (o >> 2) ? sa : -sa);     // exp(-i angle) = (cos, -sin)
}
// forward stage: v = DFT_R of the block's R points m apart, output r times W_L^(k r) (L = R m), written where it was read.  A
// thread takes MB butterflies per trip, their reads all requested first (nothing else hides the LDS and twiddle latencies of a
// workgroup that has one wave per SIMD).
template <typename T, int R>
struct DrMB {
  // (float64: one - two radix-8 butterflies and their twiddles spill ~90 registers at four waves per SIMD, 0.42 -> 0.68 ms per
  // iteration at n_fft 2048; float32: two)
  static constexpr int value = sizeof(T) == 8 ? 1 : (R >= 8 ? 1 : 2);
};
template <typename T, int R>
__device__ __forceinline__ void dr_stage_fwd(cplx<T>* __restrict__ a, const cplx<T>* __restrict__ tab, int lg8, const FrameCfg<T>& c, int lm) {
  constexpr int MB = DrMB<T, R>::value;
  const int N = c.n_fft, nb = N / R, m = 1 << lm, tws = N / (R * m);
  for (int j0 = threadIdx.x; j0 < nb; j0 += MB * blockDim.x) {
    cplx<T> v[MB][R], w[MB][R];
    int base[MB];
#pragma unroll
    for (int it = 0; it < MB; ++it) {
      const int j = j0 + it * blockDim.x;
      base[it] = -1;
      if (j < nb) {
        const int blk = j >> lm, k = j & (m - 1);
        base[it] = blk * (R * m) + k;
#pragma unroll
        for (int q = 0; q < R; ++q) v[it][q] = a[dr_phys(base[it] + q * m)];
        if (m > 1) {
          const int kt = k * tws;
#pragma unroll
          for (int r = 1; r < R; ++r) w[it][r] = dr_tw<T>(tab, lg8, kt * r);
        }
      }
    }
#pragma unroll
    for (int it = 0; it < MB; ++it) {
      if (base[it] >= 0) {
        Butterfly<T, R, false>::run(v[it], c.tw, N);
        if (m > 1) {
#pragma unroll
          for (int r = 1; r < R; ++r) v[it][r] = cmul(v[it][r], w[it][r]);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) a[dr_phys(base[it] + r * m)] = v[it][r];
      }
    }
  }
}
// inverse stage: the adjoint - input q times conj W_L^(k q), inverse DFT_R, in place
template <typename T, int R>
__device__ __forceinline__ void dr_stage_inv(cplx<T>* __restrict__ a, const cplx<T>* __restrict__ tab, int lg8, const FrameCfg<T>& c, int lm) {
  constexpr int MB = DrMB<T, R>::value;
  const int N = c.n_fft, nb = N / R, m = 1 << lm, tws = N / (R * m);
  for (int j0 = threadIdx.x; j0 < nb; j0 += MB * blockDim.x) {
    cplx<T> v[MB][R], w[MB][R];
    int base[MB];
#pragma unroll
    for (int it = 0; it < MB; ++it) {
      const int j = j0 + it * blockDim.x;
      base[it] = -1;
      if (j < nb) {
        const int blk = j >> lm, k = j & (m - 1);
        base[it] = blk * (R * m) + k;
#pragma unroll
        for (int q = 0; q < R; ++q) v[it][q] = a[dr_phys(base[it] + q * m)];
        if (m > 1) {
          const int kt = k * tws;
#pragma unroll
          for (int q = 1; q < R; ++q) w[it][q] = dr_tw<T>(tab, lg8, kt * q);
        }
      }
    }
#pragma unroll
    for (int it = 0; it < MB; ++it) {
      if (base[it] >= 0) {
        if (m > 1) {
#pragma unroll
          for (int q = 1; q < R; ++q) v[it][q] = cmul(v[it][q], conj(w[it][q]));
        }
        Butterfly<T, R, true>::run(v[it], c.tw, N);
#pragma unroll
        for (int r = 0; r < R; ++r) a[dr_phys(base[it] + r * m)] = v[it][r];
      }
    }
  }
}
// natural order in -> digit-reversed out / digit-reversed in -> natural out; every stage ends with the workgroup synchronised
template <typename T>
__device__ inline void dr_fft_fwd(cplx<T>* a, const cplx<T>* tab, int lg8, const FrameCfg<T>& c) {
  for (int i = 0; i < c.dr_stages; ++i) {
    switch (c.dr_bits[i]) {
      case 1: dr_stage_fwd<T, 2>(a, tab, lg8, c, c.dr_shift[i]); break;
      case 2: dr_stage_fwd<T, 4>(a, tab, lg8, c, c.dr_shift[i]); break;
      default: dr_stage_fwd<T, 8>(a, tab, lg8, c, c.dr_shift[i]); break;
    }
    __syncthreads();
  }
}
template <typename T>
__device__ inline void dr_fft_inv(cplx<T>* a, const cplx<T>* tab, int lg8, const FrameCfg<T>& c) {
  for (int i = c.dr_stages - 1; i >= 0; --i) {
    switch (c.dr_bits[i]) {
      case 1: dr_stage_inv<T, 2>(a, tab, lg8, c, c.dr_shift[i]); break;
      case 2: dr_stage_inv<T, 4>(a, tab, lg8, c, c.dr_shift[i]); break;
      default: dr_stage_inv<T, 8>(a, tab, lg8, c, c.dr_shift[i]); break;
    }
    __syncthreads();
  }
}

// windowed frame t of row `x` -> LDS (imaginary part zero); ends synchronised
template <typename T>
__device__ inline void load_frame(const FrameCfg<T>& c, const T* __restrict__ x, int t, cplx<T>* a,
                                  const T* __restrict__ window) {
  const int64_t start = (int64_t)t * c.hop - c.pad;
  for (int n = threadIdx.x; n < c.n_fft; n += blockDim.x) {
    T v = load_padded(x, c.length, start + n, c.pad_mode);
    a[n] = mk<T>(v * window[n], T(0));
  }
  __syncthreads();
}

// LDS bins [0, n_freq) -> real frame (inverse FFT, scale, synthesis window) -> out[n_fft]
template <typename T, bool IP = false>
__device__ inline void spectrum_to_frame(const FrameCfg<T>& c, cplx<T>* a, cplx<T>* b, T* __restrict__ out,
                                         const T* __restrict__ window) {
  const int N = c.n_fft;
  if (c.onesided) {
    // irfft semantics: Hermitian extension, imaginary parts of DC / Nyquist ignored
    for (int f = threadIdx.x; f <= N / 2; f += blockDim.x) {
      cplx<T> v = a[f];
      if (f == 0 || 2 * f == N) {
        a[f] = mk<T>(v.x, T(0));
      } else {
        a[N - f] = conj(v);
      }
    }
    __syncthreads();
  }
  lds_fft<T, IP>(a, b, c, true);
  for (int n = threadIdx.x; n < N; n += blockDim.x) out[n] = (a[n].x * c.inv_scale) * window[n];
}

// sum over the workgroup; result valid on thread 0
__device__ inline double block_sum(double v, double* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  double tot = 0;
  if (threadIdx.x == 0) {
    const int nw = (blockDim.x + 63) >> 6;
    for (int w = 0; w < nw; ++w) tot += red[w];
  }
  return tot;
}

// ---- forward STFT: x (B, length) -> spec (B, T, F) [frame-major internal layout] -----------
template <typename T, bool IP = false>
__global__ void k_stft(FrameCfg<T> c, const T* __restrict__ x, cplx<T>* __restrict__ spec) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cplx<T>* a = reinterpret_cast<cplx<T>*>(smem);
  cplx<T>* b = IP ? a : a + c.n_fft;
  const int t = blockIdx.x, bi = blockIdx.y;
  load_frame(c, x + (int64_t)bi * c.length, t, a, c.window);
  lds_fft<T, IP>(a, b, c, false);
  cplx<T>* out = spec + ((int64_t)bi * c.n_frames + t) * c.n_freq;
  for (int f = threadIdx.x; f < c.n_freq; f += blockDim.x) out[f] = mk<T>(a[f].x * c.fwd_scale, a[f].y * c.fwd_scale);
}

// ---- overlap-add + envelope division (methods.py:127,132) ---------------------------------
// x[b, n] = (sum_t frames[b, t, n + pad - t*hop]) / env[n], t ascending.
template <typename T>
__global__ void k_ola(const T* __restrict__ frames, const T* __restrict__ env, T* __restrict__ x, int n_fft,
                      int hop, int pad, int n_frames, int64_t length, int64_t total, int use_env) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int64_t bi = i / length;
  const int64_t n = i - bi * length;
  const int64_t np = n + pad;
  int64_t t_hi = np / hop;
  if (t_hi > n_frames - 1) t_hi = n_frames - 1;
  int64_t t_lo = np - n_fft + 1 <= 0 ? 0 : (np - n_fft + hop) / hop;
  const T* fr = frames + bi * n_frames * n_fft;
  T acc = 0;
  for (int64_t t = t_lo; t <= t_hi; ++t) acc += fr[t * n_fft + (np - t * hop)];
  x[i] = use_env ? acc / env[n] : acc;
}

// four consecutive samples per thread (float, hop / n_fft / pad / length all multiples of 4): the four samples
// share their frame range and sit contiguously in every frame, so each term is one 16-byte load.  Same sums in
// the same order as k_ola.
static __global__ void k_ola_f4(const float* __restrict__ frames, const float* __restrict__ env, float* __restrict__ x, int n_fft,
                         int hop, int pad, int n_frames, int64_t length, int64_t total4, int use_env) {
  using f4 = float __attribute__((ext_vector_type(4)));
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total4) return;
  const int64_t l4 = length / 4;
  const int64_t bi = i / l4;
  const int64_t n = (i - bi * l4) * 4;
  const int64_t np = n + pad;
  int64_t t_hi = np / hop;
  if (t_hi > n_frames - 1) t_hi = n_frames - 1;
  const int64_t t_lo = np - n_fft + 1 <= 0 ? 0 : (np - n_fft + hop) / hop;
  const float* fr = frames + bi * n_frames * n_fft;
  f4 acc = f4{0.0f, 0.0f, 0.0f, 0.0f};
  for (int64_t t = t_lo; t <= t_hi; ++t) acc += *reinterpret_cast<const f4*>(fr + t * n_fft + (np - t * hop));
  if (use_env) acc = acc / *reinterpret_cast<const f4*>(env + n);
  *reinterpret_cast<f4*>(x + bi * length + n) = acc;
}

// ... two consecutive samples per thread in float64 (hop / n_fft / pad / length even): 16-byte loads as in k_ola_f4
static __global__ void k_ola_d2(const double* __restrict__ frames, const double* __restrict__ env, double* __restrict__ x, int n_fft,
                                int hop, int pad, int n_frames, int64_t length, int64_t total2, int use_env) {
  using d2 = double __attribute__((ext_vector_type(2)));
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total2) return;
  const int64_t l2 = length / 2;
  const int64_t bi = i / l2;
  const int64_t n = (i - bi * l2) * 2;
  const int64_t np = n + pad;
  int64_t t_hi = np / hop;
  if (t_hi > n_frames - 1) t_hi = n_frames - 1;
  const int64_t t_lo = np - n_fft + 1 <= 0 ? 0 : (np - n_fft + hop) / hop;
  const double* fr = frames + bi * n_frames * n_fft;
  d2 acc = d2{0.0, 0.0};
  for (int64_t t = t_lo; t <= t_hi; ++t) acc += *reinterpret_cast<const d2*>(fr + t * n_fft + (np - t * hop));
  if (use_env) acc = acc / *reinterpret_cast<const d2*>(env + n);
  *reinterpret_cast<d2*>(x + bi * length + n) = acc;
}

// ---- Griffin-Lim / ADMM iteration, frame part (methods.py:237-248, :458-477) -----------------
// ---- two frames per complex FFT ---------------------------------------------------------------
// The frames are real, so frames t and t+1 ride one complex transform: z = a + i b gives
// A_f = (Z_f + conj Z_{N-f})/2, B_f = (Z_f - conj Z_{N-f})/(2i); on the way back the spectra that
// irfft / ifft(.).real actually see are the Hermitian parts H(Y)_f = (Y_f + conj Y_{N-f})/2 (for a
// one-sided spectrum: the Hermitian extension, DC / Nyquist imaginary parts ignored), and
// Z'_f = H(Y_A)_f + i H(Y_B)_f comes back as a = Re z', b = Im z'.  Works for any n_fft and any radix.
// One thread owns the bin pair (f, N-f) of both frames.
template <typename T, int MODE>   // 0: Griffin-Lim (S0 = pre_spec), 1: ADMM (S0 = X, S1 = U)
__device__ __forceinline__ cplx<T> update_core(cplx<T> r, T m, cplx<T> s0, cplx<T> s1, T coef, T inv1p, bool eval,
                                               double& s_d, double& s_o, cplx<T>& n0, cplx<T>& n1) {
  if (eval) {
    const T o = si_hypot(r.x, r.y);
    const double d = (double)o - (double)m;
    s_d += d * d;
    s_o += (double)o * (double)o;
  }
  if (MODE == 0) {                                             // methods.py:243-247
    const cplx<T> sv = mk<T>(r.x - s0.x * coef, r.y - s0.y * coef);
    n0 = sv;
    const T inv = proj_inv(sv.x, sv.y);
    return mk<T>((sv.x * m) * inv, (sv.y * m) * inv);
  } else {                                                     // methods.py:467-475
    const cplx<T> xo = s0, uo = s1;
    const cplx<T> y = xo + uo;
    const cplx<T> z = mk<T>((coef * y.x + r.x) * inv1p, (coef * y.y + r.y) * inv1p);
    const cplx<T> un = (uo + xo) - z;
    cplx<T> xn = z - un;
    const T inv = proj_inv(xn.x, xn.y);
    xn = mk<T>((xn.x * m) * inv, (xn.y * m) * inv);
    n0 = xn;
    n1 = un;
    return xn + un;
  }
}

template <typename T, int MODE>
__device__ __forceinline__ cplx<T> update_one(cplx<T> r, cplx<T>* __restrict__ S0, cplx<T>* __restrict__ S1,
                                              const T* __restrict__ mag, int64_t idx, T coef, T inv1p, bool eval,
                                              double& s_d, double& s_o) {
  cplx<T> n0, n1 = mk<T>(T(0), T(0));
  const cplx<T> s1 = MODE == 1 ? S1[idx] : mk<T>(T(0), T(0));
  const cplx<T> y = update_core<T, MODE>(r, mag[idx], S0[idx], s1, coef, inv1p, eval, s_d, s_o, n0, n1);
  S0[idx] = n0;
  if (MODE == 1) S1[idx] = n1;
  return y;
}

// MAXT: the largest workgroup the instantiation is launched with (1024: the register allocator keeps to 128 registers per lane -
// the in-place float64 form then spills 16 ... 24 of them; workgroups of at most 256 threads take the 256-thread instantiation)
template <typename T, int MODE, bool EVAL, bool IP = false, int MAXT = 1024>
__global__ __launch_bounds__(MAXT) void k_iter_pair(FrameCfg<T> c, const T* __restrict__ x, cplx<T>* __restrict__ S0, cplx<T>* __restrict__ S1,
                            const T* __restrict__ mag, T coef, T inv1p, T* __restrict__ frames,
                            double* __restrict__ partials) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ double red[16];
  cplx<T>* a = reinterpret_cast<cplx<T>*>(smem);
  cplx<T>* b = IP ? a : a + c.n_fft;     // (in place: a bin pair (f, N - f) is read and rewritten by one thread)
  // Small transforms keep a copy of the twiddle table in LDS (FrameCfg::tw_lds): every stage otherwise waits an L2 round trip for
  // its radix - 1 twiddles, with one wave per SIMD and nothing else to run (round 5: 15 - 20 % of an iteration).  The stages read
  // it through a pointer handed down to them (a flat address into LDS; a modified copy of the argument record would live in
  // scratch memory: 3 x slower); the first barrier of the frame load makes it visible.
  const cplx<T>* twp = c.tw;
  if (c.tw_lds) {
    cplx<T>* twl = a + (IP ? 1 : 2) * c.n_fft;
    for (int i = threadIdx.x; i < c.n_fft; i += blockDim.x) twl[i] = c.tw[i];
    twp = twl;
  }
  const int N = c.n_fft, F = c.n_freq;
  const int t0 = 2 * blockIdx.x, bi = blockIdx.y;
  const T* xr = x + (int64_t)bi * c.length;
  const T hs = T(0.5) * c.fwd_scale;
  double s_d = 0, s_o = 0;
  // Normally one pass with (ta, tb) = (t0, t0+1).  A non-finite sample (the reference's 0/0 where the envelope
  // vanishes, methods.py:132) must stay inside its own frame, so such a pair is done as two single-frame passes.
  int ta = t0, tb = t0 + 1 < c.n_frames ? t0 + 1 : -1;
  for (int pass = 0; pass < 2; ++pass) {
    int bad = 0;
    {
      const int64_t sa = (int64_t)ta * c.hop - c.pad, sb = (int64_t)tb * c.hop - c.pad;
      if (sa >= 0 && tb >= 0 && sb + N <= c.length) {          // both frames inside the signal: no padding tests
        const T *pa = xr + sa, *pb = xr + sb;
        for (int n = threadIdx.x; n < N; n += blockDim.x) {
          const T w = c.window[n];
          const T va = pa[n] * w, vb = pb[n] * w;
          bad |= !__builtin_isfinite(va) || !__builtin_isfinite(vb);
          a[n] = mk<T>(va, vb);
        }
      } else {
        for (int n = threadIdx.x; n < N; n += blockDim.x) {
          const T w = c.window[n];
          const T va = load_padded(xr, c.length, sa + n, c.pad_mode) * w;
          const T vb = tb >= 0 ? load_padded(xr, c.length, sb + n, c.pad_mode) * w : T(0);
          bad |= !__builtin_isfinite(va) || !__builtin_isfinite(vb);
          a[n] = mk<T>(va, vb);
        }
      }
    }
    const bool split = __syncthreads_or(bad) && tb >= 0;   // (also the barrier after the load)
    if (split) {
      tb = -1;
      for (int n = threadIdx.x; n < N; n += blockDim.x) a[n].y = T(0);
      __syncthreads();
    }
    lds_fft<T, IP>(a, b, c, false, -1, 0, twp);
    const int64_t base_a = ((int64_t)bi * c.n_frames + ta) * F, base_b = ((int64_t)bi * c.n_frames + tb) * F;
    const bool has_b = tb >= 0;
    if (c.onesided) {
      // spectra of the two frames at bin f (bin g = N - f holds their conjugates) -> updated Hermitian parts
      auto bin = [&](int f, T ma, cplx<T> a0, cplx<T> a1, T mb, cplx<T> b0, cplx<T> b1) {
        const int g = f ? N - f : 0;
        const cplx<T> zf = a[f], zg = a[g];
        const cplx<T> ra = mk<T>((zf.x + zg.x) * hs, (zf.y - zg.y) * hs);
        const cplx<T> rb = mk<T>((zf.y + zg.y) * hs, (zg.x - zf.x) * hs);
        cplx<T> n0, n1, hb = mk<T>(T(0), T(0));
        cplx<T> ha = update_core<T, MODE>(ra, ma, a0, a1, coef, inv1p, EVAL, s_d, s_o, n0, n1);
        S0[base_a + f] = n0;
        if (MODE == 1) S1[base_a + f] = n1;
        if (has_b) {
          hb = update_core<T, MODE>(rb, mb, b0, b1, coef, inv1p, EVAL, s_d, s_o, n0, n1);
          S0[base_b + f] = n0;
          if (MODE == 1) S1[base_b + f] = n1;
        }
        if (g == f) {
          ha.y = T(0);
          hb.y = T(0);
        }
        b[f] = mk<T>(ha.x - hb.y, ha.y + hb.x);
        if (g != f) b[g] = mk<T>(ha.x + hb.y, hb.x - ha.y);
      };
      const cplx<T> zero = mk<T>(T(0), T(0));
      for (int f = threadIdx.x; f <= N / 2; f += blockDim.x)
        bin(f, mag[base_a + f], S0[base_a + f], MODE == 1 ? S1[base_a + f] : zero, has_b ? mag[base_b + f] : T(0),
            has_b ? S0[base_b + f] : zero, (MODE == 1 && has_b) ? S1[base_b + f] : zero);
    } else {
      for (int f = threadIdx.x; f <= N / 2; f += blockDim.x) {
        const int g = f ? N - f : 0;
        const cplx<T> zf = a[f], zg = a[g];
        const cplx<T> ra = mk<T>((zf.x + zg.x) * hs, (zf.y - zg.y) * hs);
        const cplx<T> rb = mk<T>((zf.y + zg.y) * hs, (zg.x - zf.x) * hs);
        const cplx<T> yaf = update_one<T, MODE>(ra, S0, S1, mag, base_a + f, coef, inv1p, EVAL, s_d, s_o);
        const cplx<T> ybf = has_b ? update_one<T, MODE>(rb, S0, S1, mag, base_b + f, coef, inv1p, EVAL, s_d, s_o)
                                  : mk<T>(T(0), T(0));
        cplx<T> yag = yaf, ybg = ybf;
        if (g != f) {
          yag = update_one<T, MODE>(conj(ra), S0, S1, mag, base_a + g, coef, inv1p, EVAL, s_d, s_o);
          if (has_b) ybg = update_one<T, MODE>(conj(rb), S0, S1, mag, base_b + g, coef, inv1p, EVAL, s_d, s_o);
        }
        // Hermitian parts of the updated spectra at bin f
        const cplx<T> ha = mk<T>(T(0.5) * (yaf.x + yag.x), T(0.5) * (yaf.y - yag.y));
        const cplx<T> hb = mk<T>(T(0.5) * (ybf.x + ybg.x), T(0.5) * (ybf.y - ybg.y));
        b[f] = mk<T>(ha.x - hb.y, ha.y + hb.x);
        if (g != f) b[g] = mk<T>(ha.x + hb.y, hb.x - ha.y);
      }
    }
    __syncthreads();
    {
      cplx<T>* tmp = a;
      a = b;
      b = tmp;
    }
    lds_fft<T, IP>(a, b, c, true, -1, 0, twp);
    T* fa = frames + ((int64_t)bi * c.n_frames + ta) * N;
    T* fb = frames + ((int64_t)bi * c.n_frames + tb) * N;
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
      const T w = c.window[n];
      fa[n] = (a[n].x * c.inv_scale) * w;
      if (has_b) fb[n] = (a[n].y * c.inv_scale) * w;
    }
    if (!split) break;
    ta = t0 + 1;          // second pass: the other frame on its own
    __syncthreads();      // everyone is done reading `a` before the next load overwrites it
  }
  if (EVAL) {
    const double d = block_sum(s_d, red);
    const double o = block_sum(s_o, red);
    if (threadIdx.x == 0) {
      const int64_t pi = (int64_t)bi * gridDim.x + blockIdx.x;
      partials[2 * pi] = d;
      partials[2 * pi + 1] = o;
    }
  }
}

// k_iter_pair on the digit-reversed in-place transform (power-of-two n_fft, any dtype / sidedness / padding): the same arithmetic
// on the same values - a frame pair per complex transform, update_core per bin - with half the stage barriers of the in-place
// Stockham form and LDS accesses that spread over the banks (round 4's counters on k_iter_pair<double>: 77 % of a wave's life at
// stage barriers, SQ_LDS_BANK_CONFLICT 36 % of the LDS cycles).  Butterflies of radix 8 / 4 / 2; only their ORDER differs from the
// Stockham kernel's, so results agree to rounding, not bit for bit.
template <typename T, int MODE, bool EVAL>
__global__ void k_iter_pair_dr(FrameCfg<T> c, const T* __restrict__ x, cplx<T>* __restrict__ S0, cplx<T>* __restrict__ S1,
                               const T* __restrict__ mag, T coef, T inv1p, T* __restrict__ frames, double* __restrict__ partials) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ double red[16];
  cplx<T>* a = reinterpret_cast<cplx<T>*>(smem);
  const int N = c.n_fft, F = c.n_freq;
  // the octant twiddle table behind the frame buffer (dr_tw): N / 8 + 1 entries (cos, sin) from the plan's table of exp(-i angle);
  // the load's barrier below makes it visible
  cplx<T>* tab = a + dr_phys(N) + 1;
  const int lg8 = c.dr_lg8;
  for (int i = threadIdx.x; i <= N / 8; i += blockDim.x) {
    const cplx<T> w = c.tw[i];
    tab[i] = mk<T>(w.x, -w.y);
  }
  const int t0 = 2 * blockIdx.x, bi = blockIdx.y;
  const T* xr = x + (int64_t)bi * c.length;
  const T hs = T(0.5) * c.fwd_scale;
  double s_d = 0, s_o = 0;
  int ta = t0, tb = t0 + 1 < c.n_frames ? t0 + 1 : -1;
  for (int pass = 0; pass < 2; ++pass) {
    int bad = 0;
    {
      const int64_t sa = (int64_t)ta * c.hop - c.pad, sb = (int64_t)tb * c.hop - c.pad;
      if (sa >= 0 && tb >= 0 && sb + N <= c.length) {          // both frames inside the signal: no padding tests
        const T *pa = xr + sa, *pb = xr + sb;
        for (int n = threadIdx.x; n < N; n += blockDim.x) {
          const T w = c.window[n];
          const T va = pa[n] * w, vb = pb[n] * w;
          bad |= !__builtin_isfinite(va) || !__builtin_isfinite(vb);
          a[dr_phys(n)] = mk<T>(va, vb);
        }
      } else {
        for (int n = threadIdx.x; n < N; n += blockDim.x) {
          const T w = c.window[n];
          const T va = load_padded(xr, c.length, sa + n, c.pad_mode) * w;
          const T vb = tb >= 0 ? load_padded(xr, c.length, sb + n, c.pad_mode) * w : T(0);
          bad |= !__builtin_isfinite(va) || !__builtin_isfinite(vb);
          a[dr_phys(n)] = mk<T>(va, vb);
        }
      }
    }
    const bool split = __syncthreads_or(bad) && tb >= 0;   // (also the barrier after the load)
    if (split) {                                           // a non-finite sample stays inside its own frame (k_iter_pair)
      tb = -1;
      for (int n = threadIdx.x; n < N; n += blockDim.x) a[dr_phys(n)].y = T(0);
      __syncthreads();
    }
    dr_fft_fwd<T>(a, tab, lg8, c);
    const int64_t base_a = ((int64_t)bi * c.n_frames + ta) * F, base_b = ((int64_t)bi * c.n_frames + tb) * F;
    const bool has_b = tb >= 0;
    const cplx<T> zero = mk<T>(T(0), T(0));
    // one thread owns the bin pair (f, N - f) of both frames: it reads and rewrites their two positions
    for (int f = threadIdx.x; f <= N / 2; f += blockDim.x) {
      const int g = f ? N - f : 0;
      const int pf = dr_phys(dr_pos(c, f)), pg = g == f ? pf : dr_phys(dr_pos(c, g));
      const cplx<T> zf = a[pf], zg = a[pg];
      const cplx<T> ra = mk<T>((zf.x + zg.x) * hs, (zf.y - zg.y) * hs);
      const cplx<T> rb = mk<T>((zf.y + zg.y) * hs, (zg.x - zf.x) * hs);
      if (c.onesided) {
        cplx<T> n0, n1, hb = zero;
        cplx<T> ha = update_core<T, MODE>(ra, mag[base_a + f], S0[base_a + f], MODE == 1 ? S1[base_a + f] : zero, coef, inv1p, EVAL,
                                          s_d, s_o, n0, n1);
        S0[base_a + f] = n0;
        if (MODE == 1) S1[base_a + f] = n1;
        if (has_b) {
          hb = update_core<T, MODE>(rb, mag[base_b + f], S0[base_b + f], MODE == 1 ? S1[base_b + f] : zero, coef, inv1p, EVAL, s_d,
                                    s_o, n0, n1);
          S0[base_b + f] = n0;
          if (MODE == 1) S1[base_b + f] = n1;
        }
        if (g == f) {
          ha.y = T(0);
          hb.y = T(0);
        }
        a[pf] = mk<T>(ha.x - hb.y, ha.y + hb.x);
        if (g != f) a[pg] = mk<T>(ha.x + hb.y, hb.x - ha.y);
      } else {
        const cplx<T> yaf = update_one<T, MODE>(ra, S0, S1, mag, base_a + f, coef, inv1p, EVAL, s_d, s_o);
        const cplx<T> ybf = has_b ? update_one<T, MODE>(rb, S0, S1, mag, base_b + f, coef, inv1p, EVAL, s_d, s_o) : zero;
        cplx<T> yag = yaf, ybg = ybf;
        if (g != f) {
          yag = update_one<T, MODE>(conj(ra), S0, S1, mag, base_a + g, coef, inv1p, EVAL, s_d, s_o);
          if (has_b) ybg = update_one<T, MODE>(conj(rb), S0, S1, mag, base_b + g, coef, inv1p, EVAL, s_d, s_o);
        }
        // Hermitian parts of the updated spectra at bin f
        const cplx<T> ha = mk<T>(T(0.5) * (yaf.x + yag.x), T(0.5) * (yaf.y - yag.y));
        const cplx<T> hb = mk<T>(T(0.5) * (ybf.x + ybg.x), T(0.5) * (ybf.y - ybg.y));
        a[pf] = mk<T>(ha.x - hb.y, ha.y + hb.x);
        if (g != f) a[pg] = mk<T>(ha.x + hb.y, hb.x - ha.y);
      }
    }
    __syncthreads();
    dr_fft_inv<T>(a, tab, lg8, c);
    T* fa = frames + ((int64_t)bi * c.n_frames + ta) * N;
    T* fb = frames + ((int64_t)bi * c.n_frames + tb) * N;
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
      const T w = c.window[n];
      const cplx<T> v = a[dr_phys(n)];
      fa[n] = (v.x * c.inv_scale) * w;
      if (has_b) fb[n] = (v.y * c.inv_scale) * w;
    }
    if (!split) break;
    ta = t0 + 1;          // second pass: the other frame on its own
    __syncthreads();      // everyone is done reading `a` before the next load overwrites it
  }
  if (EVAL) {
    const double d = block_sum(s_d, red);
    const double o = block_sum(s_o, red);
    if (threadIdx.x == 0) {
      const int64_t pi = (int64_t)bi * gridDim.x + blockIdx.x;
      partials[2 * pi] = d;
      partials[2 * pi + 1] = o;
    }
  }
}

// ---- deterministic reductions ----------------------------------------------------------------
// sums[k] = sum_i partials[n_comp*i + k] for k < n_comp; single workgroup, fixed order.
static __global__ void k_finish_partials(const double* __restrict__ partials, int64_t n, int n_comp,
                                  double* __restrict__ sums) {
  __shared__ double red[16];
  for (int k = 0; k < n_comp; ++k) {
    double v = 0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) v += partials[n_comp * i + k];
    const double tot = block_sum(v, red);
    if (threadIdx.x == 0) sums[k] = tot;
  }
}

static __global__ void k_store2(double* __restrict__ out, double a, double b) {
  out[0] = a;
  out[1] = b;
}

// per-block partial sums of (a-b)^2, a^2, b^2 (b may be null: only a^2 is meaningful)
template <typename T>
__global__ void k_metric_partials(const T* __restrict__ a, const T* __restrict__ b, int64_t n,
                                  double* __restrict__ partials) {
  __shared__ double red[16];
  double sd = 0, sa = 0, sb = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double av = (double)a[i];
    const double bv = b ? (double)b[i] : 0.0;
    sd += (av - bv) * (av - bv);
    sa += av * av;
    sb += bv * bv;
  }
  const double d = block_sum(sd, red), aa = block_sum(sa, red), bb = block_sum(sb, red);
  if (threadIdx.x == 0) {
    partials[3 * blockIdx.x] = d;
    partials[3 * blockIdx.x + 1] = aa;
    partials[3 * blockIdx.x + 2] = bb;
  }
}

// ---- batched 2-D transpose (B, R, C) -> (B, C, R), any element type ---------------------------
template <typename E>
__global__ void k_transpose(const E* __restrict__ in, E* __restrict__ out, int R, int C) {
  __shared__ E tile[32][33];
  const int64_t boff = (int64_t)blockIdx.z * R * C;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    const int r = r0 + i, cc = c0 + threadIdx.x;
    if (r < R && cc < C) tile[i][threadIdx.x] = in[boff + (int64_t)r * C + cc];
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    const int cc = c0 + i, r = r0 + threadIdx.x;
    if (r < R && cc < C) out[boff + (int64_t)cc * R + r] = tile[threadIdx.x][i];
  }
}

// ---- phase_init (methods.py:572-615) -----------------------------------------------------------
// One wave per (b, f) row of the (B, F, T) magnitude: peak test / omega in the input dtype in
// the reference's operation order, float64 running sum over time with each partial sum
// rounded to the input dtype (ATen's CPU cumsum), cos/sin evaluated in float64 of the rounded
// phase and rounded once.
template <typename T>
__device__ inline bool peak_omega(const T* __restrict__ col, int64_t fstride, int g, int F, T two_pi, T n_fft,
                                  T hop, T& w) {
#pragma clang fp contract(off)
  if (g < 1 || g > F - 2) return false;
  const T a = col[(int64_t)(g - 1) * fstride];
  const T bb = col[(int64_t)g * fstride];
  const T r = col[(int64_t)(g + 1) * fstride];
  if (!(bb > r && bb > a)) return false;                // :597
  const T p = T(0.5) * (a - r) / (a - T(2) * bb + r);   // :604
  w = two_pi * (T(g) + p) / n_fft * hop;                // :605
  return true;
}

// omega of bin f after the scatter of :607-609 - its own peak, else the k+1 write of a peak below, else the k-1 write of a peak
// above (later statements overwrite earlier ones), else 0 - from the five magnitudes f-2 .. f+2 of one time step.  The three
// peak tests are comparisons; the two divisions of :604-605 run ONCE, on the operands of whichever peak won (the same
// operations on the same values as peak_omega: bit-identical), instead of once per candidate.
template <typename T>
__device__ inline T scatter_omega(const T (&cur)[5], int f, int F, T two_pi, T n_fft, T hop) {
#pragma clang fp contract(off)
  const bool own = f >= 1 && f <= F - 2 && cur[2] > cur[3] && cur[2] > cur[1];             // :597 at g = f
  const bool below = f - 1 >= 1 && f - 1 <= F - 2 && cur[1] > cur[2] && cur[1] > cur[0];   //      at g = f - 1
  const bool above = f + 1 >= 1 && f + 1 <= F - 2 && cur[3] > cur[4] && cur[3] > cur[2];   //      at g = f + 1
  const int g = own ? f : below ? f - 1 : f + 1;
  const T a = own ? cur[1] : below ? cur[0] : cur[2];
  const T bb = own ? cur[2] : below ? cur[1] : cur[3];
  const T r = own ? cur[3] : below ? cur[2] : cur[4];
  const T p = T(0.5) * (a - r) / (a - T(2) * bb + r);   // :604
  const T w = two_pi * (T(g) + p) / n_fft * hop;        // :605
  return (own || below || above) ? w : T(0);
}

template <typename T>
__global__ void k_phase_init(const T* __restrict__ mag, cplx<T>* __restrict__ out, int B, int F, int Tn,
                             int n_fft, int hop) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= B * F) return;
  const int bi = row / F, f = row - bi * F;
  const T* base = mag + (int64_t)bi * F * Tn;
  cplx<T>* orow = out + ((int64_t)bi * F + f) * Tn;
  const T two_pi = T(6.283185307179586476925286766559);
  // the five rows f-2 .. f+2 of one time step; the next 64 time steps are fetched while the current ones are scanned
  auto fetch = [&](int t, T (&v)[5]) {
#pragma unroll
    for (int d = 0; d < 5; ++d) {
      const int g = f + d - 2;
      v[d] = (t < Tn && g >= 0 && g < F) ? base[(int64_t)g * Tn + t] : T(0);
    }
  };
  T cur[5], nxt[5];
  fetch(lane, cur);
  double carry = 0;
  for (int t0 = 0; t0 < Tn; t0 += 64) {
    const int t = t0 + lane;
    fetch(t + 64, nxt);
    T om = 0;
    const T m0 = cur[2];
    if (t < Tn) om = scatter_omega<T>(cur, f, F, two_pi, T(n_fft), T(hop));
    double v = (double)om;
    v = wave_scan_inclusive(v);
    v += carry;
    carry = __shfl(v, 63, 64);
    if (t < Tn) {
      const T phi = (T)v;                                  // :611
      double s, cs;
      sincos_phase(phi, &s, &cs);                          // :612
      orow[t] = mk<T>(m0 * (T)cs, m0 * (T)s);              // :614
    }
#pragma unroll
    for (int d = 0; d < 5; ++d) cur[d] = nxt[d];
  }
}

// |z| of a complex array (target_spec = spec.abs(), methods.py:110)
template <typename T>
__global__ void k_cabs(const cplx<T>* __restrict__ in, T* __restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = si_hypot(in[i].x, in[i].y);
}

}  // namespace specinv
