// Building blocks for differentiating Griffin-Lim with respect to the input spectrogram (the reference's
// results are differentiable through torch autograd: test/test_griffin.py:54,65-66).  All element-wise kernels
// work on (B, F, T) arrays in the caller's layout; the linear operators' adjoints reuse the transform kernels.
//
//   forward step (methods.py:243-247):  S = R - lr P ; Q = S m / (|S| + 1e-16)
//   adjoint:  gS = gQ m/d - S/|S| * Re(conj(gQ) S) m/d^2 + gP_next ,  d = |S| + 1e-16
//             gR = gS ; gP = -lr gS ; gm = Re(conj(gQ) S)/d
//   ISTFT adjoint: u = g/env ; Y = window-weighted forward DFT of the zero-padded frames of u ;
//             gQ = inv_scale * (interior ? 2 Y : Re Y)   (onesided Hermitian inverse, DC/Nyquist real)
//   STFT adjoint: the L_BFGS gradient path (halve interior, Hermitian inverse with the forward scale, fold).
//   phase_init adjoint (methods.py:597-614): through m e^{i phi}, the time cumsum and the parabolic peak offset.
#pragma once
#include "common.h"

namespace specinv {

template <typename T>
__global__ void k_gla_update(const cplx<T>* __restrict__ R, const cplx<T>* __restrict__ P, const T* __restrict__ m, T lr,
                             cplx<T>* __restrict__ S_out, cplx<T>* __restrict__ Q_out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const cplx<T> r = R[i], p = P[i];
  const cplx<T> s = mk<T>(r.x - p.x * lr, r.y - p.y * lr);
  S_out[i] = s;
  const T inv = T(1) / (si_hypot(s.x, s.y) + eps16<T>::value);
  Q_out[i] = mk<T>((s.x * m[i]) * inv, (s.y * m[i]) * inv);
}

template <typename T>
__global__ void k_gla_update_adjoint(const cplx<T>* __restrict__ gQ, const cplx<T>* __restrict__ gPn,
                                     const cplx<T>* __restrict__ S, const T* __restrict__ m, T lr,
                                     cplx<T>* __restrict__ gR, cplx<T>* __restrict__ gP, T* __restrict__ gm, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const cplx<T> g = gQ[i], s = S[i];
  const T mag = si_hypot(s.x, s.y);
  const T d = mag + eps16<T>::value;
  const T dot = g.x * s.x + g.y * s.y;                     // Re(conj(gQ) S)
  const T c1 = m[i] / d;
  const T c2 = mag > T(0) ? dot * m[i] / (d * d * mag) : T(0);
  cplx<T> gs = mk<T>(g.x * c1 - s.x * c2, g.y * c1 - s.y * c2);
  if (gPn) {
    gs.x += gPn[i].x;
    gs.y += gPn[i].y;
  }
  gR[i] = gs;
  gP[i] = mk<T>(-lr * gs.x, -lr * gs.y);
  gm[i] += dot / d;
}

// ADMM closure without the transforms (methods.py:467-475) and its adjoint.
//   Y = X + U ; Z = (rho Y + R)/(1+rho) ; U' = U + X - Z ; V = Z - U' ; X' = V m/(|V| + 1e-16) ; Y' = X' + U'
template <typename T>
__global__ void k_admm_update(const cplx<T>* __restrict__ R, const cplx<T>* __restrict__ X, const cplx<T>* __restrict__ U,
                              const T* __restrict__ m, T rho, T inv1p, cplx<T>* __restrict__ Xn, cplx<T>* __restrict__ Un,
                              cplx<T>* __restrict__ V_out, cplx<T>* __restrict__ Yn, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const cplx<T> r = R[i], xo = X[i], uo = U[i];
  const cplx<T> y = xo + uo;
  const cplx<T> z = mk<T>((rho * y.x + r.x) * inv1p, (rho * y.y + r.y) * inv1p);
  const cplx<T> un = (uo + xo) - z;
  const cplx<T> v = z - un;
  const T inv = T(1) / (si_hypot(v.x, v.y) + eps16<T>::value);
  const cplx<T> xn = mk<T>((v.x * m[i]) * inv, (v.y * m[i]) * inv);
  Xn[i] = xn;
  Un[i] = un;
  V_out[i] = v;
  Yn[i] = xn + un;
}

// cotangents of (Y' via the ISTFT, X', U') -> cotangents of (R, X, U), gm += d/dm.  gXn / gUn may be NULL.
template <typename T>
__global__ void k_admm_update_adjoint(const cplx<T>* __restrict__ gYn, const cplx<T>* __restrict__ gXn,
                                      const cplx<T>* __restrict__ gUn, const cplx<T>* __restrict__ V,
                                      const T* __restrict__ m, T rho, T inv1p, cplx<T>* __restrict__ gR,
                                      cplx<T>* __restrict__ gX, cplx<T>* __restrict__ gU, T* __restrict__ gm, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  cplx<T> gx = gYn[i], gu = gYn[i];                       // Y' = X' + U'
  if (gXn) gx = gx + gXn[i];
  if (gUn) gu = gu + gUn[i];
  // X' = proj(V, m)
  const cplx<T> v = V[i];
  const T mag = si_hypot(v.x, v.y);
  const T d = mag + eps16<T>::value;
  const T dot = gx.x * v.x + gx.y * v.y;
  const T c1 = m[i] / d;
  const T c2 = mag > T(0) ? dot * m[i] / (d * d * mag) : T(0);
  const cplx<T> gv = mk<T>(gx.x * c1 - v.x * c2, gx.y * c1 - v.y * c2);
  gm[i] += dot / d;
  gu = gu - gv;                                           // V = Z - U'
  cplx<T> gz = gv - gu;                                   // ... and U' = U + X - Z
  const cplx<T> gy = mk<T>(gz.x * (rho * inv1p), gz.y * (rho * inv1p));   // Z = (rho Y + R)/(1+rho)
  gR[i] = mk<T>(gz.x * inv1p, gz.y * inv1p);
  gX[i] = gu + gy;                                        // U' = U + X - Z ; Y = X + U
  gU[i] = gu + gy;
}

template <typename T>
__global__ void k_div_env(const T* __restrict__ g, const T* __restrict__ env, T* __restrict__ u, int64_t L, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total) u[i] = g[i] / env[i % L];
}

// frame-major (BT, F): gQ = inv_scale * (interior ? 2 Y : Re Y) for the onesided Hermitian inverse; inv_scale * Y two-sided
template <typename T>
__global__ void k_istft_adjoint_scale(cplx<T>* __restrict__ y, int F, int n_fft, int onesided, T inv_scale, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int f = i % F;
  cplx<T> v = y[i];
  if (onesided) {
    if (f == 0 || 2 * f == n_fft) v = mk<T>(v.x * inv_scale, T(0));
    else v = mk<T>(v.x * (2 * inv_scale), v.y * (2 * inv_scale));
  } else {
    v = mk<T>(v.x * inv_scale, v.y * inv_scale);
  }
  y[i] = v;
}

template <typename T>
__global__ void k_halve_interior(cplx<T>* __restrict__ g, int F, int n_fft, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int f = i % F;
  if (f != 0 && 2 * f != n_fft) g[i] = mk<T>(g[i].x * T(0.5), g[i].y * T(0.5));
}

// ---- phase_init adjoint ---------------------------------------------------------------------------------
// pass 1 (one wave per (b, f) row): gm_direct = Re(conj(e^{i phi}) gC) and gomega = reverse cumsum over time of
// gphi = Re(conj(i C) gC); phi is recomputed exactly like the forward kernel does.
template <typename T>
__device__ inline bool peak_omega_adj(const T* __restrict__ col, int64_t fstride, int g, int F, T two_pi, T n_fft, T hop,
                                      T& w) {
#pragma clang fp contract(off)
  if (g < 1 || g > F - 2) return false;
  const T a = col[(int64_t)(g - 1) * fstride], bb = col[(int64_t)g * fstride], r = col[(int64_t)(g + 1) * fstride];
  if (!(bb > r && bb > a)) return false;
  const T p = T(0.5) * (a - r) / (a - T(2) * bb + r);
  w = two_pi * (T(g) + p) / n_fft * hop;
  return true;
}

template <typename T>
__global__ void k_phase_init_adjoint_rows(const T* __restrict__ mag, const cplx<T>* __restrict__ gC, T* __restrict__ gm,
                                          T* __restrict__ gomega, int B, int F, int Tn, int n_fft, int hop) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= B * F) return;
  const int bi = row / F, f = row - bi * F;
  const T* base = mag + (int64_t)bi * F * Tn;
  const int64_t roff = ((int64_t)bi * F + f) * Tn;
  const T two_pi = T(6.283185307179586476925286766559);
  // forward scan to rebuild phi (same arithmetic as k_phase_init), storing gphi in gomega temporarily
  double carry = 0;
  for (int t0 = 0; t0 < Tn; t0 += 64) {
    const int t = t0 + lane;
    T om = 0;
    if (t < Tn) {
      const T* col = base + t;
      T w;
      if (peak_omega_adj<T>(col, Tn, f, F, two_pi, T(n_fft), T(hop), w)) om = w;
      else if (peak_omega_adj<T>(col, Tn, f - 1, F, two_pi, T(n_fft), T(hop), w)) om = w;
      else if (peak_omega_adj<T>(col, Tn, f + 1, F, two_pi, T(n_fft), T(hop), w)) om = w;
    }
    double v = (double)om;
    v = wave_scan_inclusive(v);
    v += carry;
    carry = __shfl(v, 63, 64);
    if (t < Tn) {
      const T phi = (T)v;
      double s, c;
      sincos_phase(phi, &s, &c);
      const cplx<T> g = gC[roff + t];
      const T m0 = base[(int64_t)f * Tn + t];
      gm[roff + t] += (T)c * g.x + (T)s * g.y;
      gomega[roff + t] = m0 * ((T)c * g.y - (T)s * g.x);       // gphi
    }
  }
  // reverse cumulative sum over time
  double rc = 0;
  const int nchunk = (Tn + 63) / 64;
  for (int ch = nchunk - 1; ch >= 0; --ch) {
    const int t = ch * 64 + (63 - lane);                       // lane 0 holds the latest time of the chunk
    double v = t < Tn ? (double)gomega[roff + t] : 0.0;
    v = wave_scan_inclusive(v);
    v += rc;
    rc = __shfl(v, 63, 64);
    if (t < Tn) gomega[roff + t] = (T)v;
  }
}

// pass 2 (thread per (b, f, t)): gather the peak-offset gradients: bin f is the 'a' of peak f+1, the 'b' of peak f,
// the 'r' of peak f-1; a peak's gp is the sum of gomega over the bins that finally hold its omega.
template <typename T>
__global__ void k_phase_init_adjoint_peaks(const T* __restrict__ mag, const T* __restrict__ gomega, T* __restrict__ gm,
                                           int B, int F, int Tn, int n_fft, int hop) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)B * F * Tn) return;
  const int t = i % Tn;
  const int f = (i / Tn) % F;
  const int64_t b = i / ((int64_t)Tn * F);
  const T* col = mag + b * F * Tn + t;
  const T* go = gomega + b * F * Tn + t;
  auto M = [&](int g) { return col[(int64_t)g * Tn]; };
  auto is_peak = [&](int g) { return g >= 1 && g <= F - 2 && M(g) > M(g + 1) && M(g) > M(g - 1); };
  // gradient w.r.t. p of peak g: omega_g sits in bin g, in bin g+1 (always: the k+1 write is last), and in bin g-1
  // unless g-2 is a peak too (whose k+1 write overwrites it)
  auto gp_of = [&](int g) -> T {
    T s = go[(int64_t)g * Tn] + go[(int64_t)(g + 1) * Tn];
    if (!is_peak(g - 2)) s += go[(int64_t)(g - 1) * Tn];
    return s * (T(6.283185307179586476925286766559) * T(hop) / T(n_fft));
  };
  T acc = 0;
  for (int role = 0; role < 3; ++role) {
    const int g = role == 0 ? f + 1 : (role == 1 ? f : f - 1);   // f is a / b / r of peak g
    if (!is_peak(g)) continue;
    const T a = M(g - 1), bb = M(g), r = M(g + 1);
    const T D = a - T(2) * bb + r, num = a - r;
    const T gp = gp_of(g);
    if (role == 0) acc += gp * (T(0.5) / D - T(0.5) * num / (D * D));
    else if (role == 1) acc += gp * (num / (D * D));
    else acc += gp * (-T(0.5) / D - T(0.5) * num / (D * D));
  }
  gm[i] += acc;
}

}  // namespace specinv
