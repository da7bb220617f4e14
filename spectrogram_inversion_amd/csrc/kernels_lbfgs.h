// L_BFGS building blocks (reference: torch_specinv/methods.py:509-569 + torch.optim.LBFGS).
//
//   * transform forward  V = |STFT(x)|  or  V = log1p(Mel @ |STFT(x)|)
//   * loss = mean((V - target)^2) and its analytic gradient w.r.t. x (verified against
//     autograd): dV = 2(V-T)/numel ; dMel = dV/(1 + Mel|S|) ; dA = Mel^T dMel ; G = dA * S/|S|
//     (0 where |S| = 0) ; frame gradient = Re sum_k G[k] e^{+2 pi i k n/N} (onesided: interior bins
//     halved, then Hermitian inverse) * window ; overlap-add without envelope ; padded margins folded
//     back according to the pad mode.
//   * the two mel contractions are dense GEMMs and run on the matrix cores with the exact-float32 MFMA
//     (v_mfma_f32_32x32x2_f32, bitwise an fmaf chain) - the only MFMA use in the library.
//   * flat-vector reductions / updates of the two-loop recursion, accumulated in float64.
#pragma once
#include <type_traits>

#include "common.h"
#include "kernels_generic.h"
#include "fast_state.h"
#include "objective_args.h"
#include "lbfgs_state.h"

namespace specinv {

// ---- flat vector kernels ------------------------------------------------------------------------------
template <typename T>
__global__ void k_dot_partials(const T* __restrict__ a, const T* __restrict__ b, int64_t n, double* __restrict__ part) {
  __shared__ double red[16];
  double s = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    s += (double)a[i] * (double)b[i];
  const double t = block_sum(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = t;
}

template <typename T>
__global__ void k_abs_partials(const T* __restrict__ a, int64_t n, double* __restrict__ part) {
  __shared__ double red[16];
  double s = 0, m = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double v = fabs((double)a[i]);
    s += v;
    m = v > m ? v : m;
  }
  // max over the block
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const double o = __shfl_xor(m, off, 64);
    m = o > m ? o : m;
  }
  __shared__ double mx[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) mx[wave] = m;
  const double t = block_sum(s, red);
  if (threadIdx.x == 0) {
    double mm = 0;
    for (int w = 0; w < (int)((blockDim.x + 63) >> 6); ++w) mm = mx[w] > mm ? mx[w] : mm;
    part[2 * blockIdx.x] = mm;
    part[2 * blockIdx.x + 1] = t;
  }
}

// out[0] = max_i part[2i], out[1] = sum_i part[2i+1]
static __global__ void k_finish_absmax(const double* __restrict__ part, int n, double* __restrict__ out) {
  __shared__ double red[16];
  double s = 0, m = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    m = part[2 * i] > m ? part[2 * i] : m;
    s += part[2 * i + 1];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const double o = __shfl_xor(m, off, 64);
    m = o > m ? o : m;
  }
  __shared__ double mx[16];
  if ((threadIdx.x & 63) == 0) mx[threadIdx.x >> 6] = m;
  const double t = block_sum(s, red);
  if (threadIdx.x == 0) {
    double mm = 0;
    for (int w = 0; w < (int)((blockDim.x + 63) >> 6); ++w) mm = mx[w] > mm ? mx[w] : mm;
    out[0] = mm;
    out[1] = t;
  }
}

template <typename T>
__global__ void k_axpy(T alpha, const T* __restrict__ x, T* __restrict__ y, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = y[i] + alpha * x[i];
}

template <typename T>
__global__ void k_scale(T alpha, const T* __restrict__ x, T* __restrict__ y, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = alpha * x[i];
}

// One pass for the curvature pair of an L-BFGS iteration: y = g - g_prev, s = t * d, partial sums of y.s, y.y, g.g and
// g.g_prev (the last two give the new pair's products with g by linearity: y.g = g.g - g_prev.g)
template <typename T>
__global__ void k_lbfgs_pair(const T* __restrict__ g, const T* __restrict__ gp, const T* __restrict__ d, T t,
                             T* __restrict__ y, T* __restrict__ sv, int64_t n, double* __restrict__ part) {
  __shared__ double red[16];
  double ys = 0, yy = 0, gg = 0, ggp = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const T gi = g[i], pi = gp[i];
    const T yi = gi - pi;
    const T si = t * d[i];
    y[i] = yi;
    sv[i] = si;
    ys += (double)yi * (double)si;
    yy += (double)yi * (double)yi;
    gg += (double)gi * (double)gi;
    ggp += (double)gi * (double)pi;
  }
  const double a = block_sum(ys, red), b = block_sum(yy, red), c = block_sum(gg, red), e = block_sum(ggp, red);
  if (threadIdx.x == 0) {
    part[4 * blockIdx.x] = a;
    part[4 * blockIdx.x + 1] = b;
    part[4 * blockIdx.x + 2] = c;
    part[4 * blockIdx.x + 3] = e;
  }
}

// One pass for what a step needs to know about g and d: g.d, max|g|, sum|g|, max|d|
template <typename T>
__global__ void k_lbfgs_stats(const T* __restrict__ g, const T* __restrict__ d, int64_t n, double* __restrict__ part) {
  __shared__ double red[16];
  __shared__ double mx[2][16];
  double gd = 0, sg = 0, mg = 0, md = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double gi = (double)g[i], di = (double)d[i];
    gd += gi * di;
    const double ag = fabs(gi), ad = fabs(di);
    sg += ag;
    mg = ag > mg ? ag : mg;
    md = ad > md ? ad : md;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const double o1 = __shfl_xor(mg, off, 64), o2 = __shfl_xor(md, off, 64);
    mg = o1 > mg ? o1 : mg;
    md = o2 > md ? o2 : md;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    mx[0][wave] = mg;
    mx[1][wave] = md;
  }
  const double a = block_sum(gd, red), b = block_sum(sg, red);
  if (threadIdx.x == 0) {
    double m0 = 0, m1 = 0;
    for (int w = 0; w < (int)((blockDim.x + 63) >> 6); ++w) {
      m0 = mx[0][w] > m0 ? mx[0][w] : m0;
      m1 = mx[1][w] > m1 ? mx[1][w] : m1;
    }
    part[4 * blockIdx.x] = a;
    part[4 * blockIdx.x + 1] = b;
    part[4 * blockIdx.x + 2] = m0;
    part[4 * blockIdx.x + 3] = m1;
  }
}

// out[0..1] = sums of part[4i], part[4i+1]; out[2..3] = maxima of part[4i+2], part[4i+3]   (one workgroup, fixed order)
static __global__ void k_finish_stats(const double* __restrict__ part, int n, double* __restrict__ out) {
  __shared__ double red[16];
  __shared__ double mx[2][16];
  double a = 0, b = 0, m0 = 0, m1 = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    a += part[4 * i];
    b += part[4 * i + 1];
    m0 = part[4 * i + 2] > m0 ? part[4 * i + 2] : m0;
    m1 = part[4 * i + 3] > m1 ? part[4 * i + 3] : m1;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const double o0 = __shfl_xor(m0, off, 64), o1 = __shfl_xor(m1, off, 64);
    m0 = o0 > m0 ? o0 : m0;
    m1 = o1 > m1 ? o1 : m1;
  }
  if ((threadIdx.x & 63) == 0) {
    mx[0][threadIdx.x >> 6] = m0;
    mx[1][threadIdx.x >> 6] = m1;
  }
  const double ta = block_sum(a, red), tb = block_sum(b, red);
  if (threadIdx.x == 0) {
    double r0 = 0, r1 = 0;
    for (int w = 0; w < (int)((blockDim.x + 63) >> 6); ++w) {
      r0 = mx[0][w] > r0 ? mx[0][w] : r0;
      r1 = mx[1][w] > r1 ? mx[1][w] : r1;
    }
    out[0] = ta;
    out[1] = tb;
    out[2] = r0;
    out[3] = r1;
  }
}

// The curvature pair and the step statistics in ONE pass over g, g_prev, d (what an L-BFGS iteration needs to know about
// the new gradient): partials per block = {g.d, sum|g|, y.s, y.y, g.g, g.g_prev, max|g|, max|d|}
template <typename T>
__global__ void k_lbfgs_pair_stats(const T* __restrict__ g, const T* __restrict__ gp, const T* __restrict__ d, T t,
                                   T* __restrict__ y, T* __restrict__ sv, int64_t n, double* __restrict__ part) {
  __shared__ double red[16];
  __shared__ double mx[2][16];
  double s[6] = {0, 0, 0, 0, 0, 0}, mg = 0, md = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const T gi = g[i], pi = gp[i], di = d[i];
    const T yi = gi - pi;
    const T si = t * di;
    y[i] = yi;
    sv[i] = si;
    const double g64 = (double)gi, d64 = (double)di, ag = fabs(g64), ad = fabs(d64);
    s[0] += g64 * d64;
    s[1] += ag;
    s[2] += (double)yi * (double)si;
    s[3] += (double)yi * (double)yi;
    s[4] += g64 * g64;
    s[5] += g64 * (double)pi;
    mg = ag > mg ? ag : mg;
    md = ad > md ? ad : md;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const double o1 = __shfl_xor(mg, off, 64), o2 = __shfl_xor(md, off, 64);
    mg = o1 > mg ? o1 : mg;
    md = o2 > md ? o2 : md;
  }
  if ((threadIdx.x & 63) == 0) {
    mx[0][threadIdx.x >> 6] = mg;
    mx[1][threadIdx.x >> 6] = md;
  }
  double tot[6];
#pragma unroll
  for (int c = 0; c < 6; ++c) tot[c] = block_sum(s[c], red);
  if (threadIdx.x == 0) {
    double m0 = 0, m1 = 0;
    for (int w = 0; w < (int)((blockDim.x + 63) >> 6); ++w) {
      m0 = mx[0][w] > m0 ? mx[0][w] : m0;
      m1 = mx[1][w] > m1 ? mx[1][w] : m1;
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) part[8 * blockIdx.x + c] = tot[c];
    part[8 * blockIdx.x + 6] = m0;
    part[8 * blockIdx.x + 7] = m1;
  }
}

// out = {g.d, sum|g|, max|g|, max|d|, y.s, y.y, g.g, g.g_prev}   (one workgroup, fixed order)
static __global__ void k_finish_pair_stats(const double* __restrict__ part, int n, double* __restrict__ out) {
  __shared__ double red[16];
  __shared__ double mx[2][16];
  double s[6] = {0, 0, 0, 0, 0, 0}, m0 = 0, m1 = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
#pragma unroll
    for (int c = 0; c < 6; ++c) s[c] += part[8 * i + c];
    m0 = part[8 * i + 6] > m0 ? part[8 * i + 6] : m0;
    m1 = part[8 * i + 7] > m1 ? part[8 * i + 7] : m1;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const double o0 = __shfl_xor(m0, off, 64), o1 = __shfl_xor(m1, off, 64);
    m0 = o0 > m0 ? o0 : m0;
    m1 = o1 > m1 ? o1 : m1;
  }
  if ((threadIdx.x & 63) == 0) {
    mx[0][threadIdx.x >> 6] = m0;
    mx[1][threadIdx.x >> 6] = m1;
  }
  double tot[6];
#pragma unroll
  for (int c = 0; c < 6; ++c) tot[c] = block_sum(s[c], red);
  if (threadIdx.x == 0) {
    double r0 = 0, r1 = 0;
    for (int w = 0; w < (int)((blockDim.x + 63) >> 6); ++w) {
      r0 = mx[0][w] > r0 ? mx[0][w] : r0;
      r1 = mx[1][w] > r1 ? mx[1][w] : r1;
    }
    out[0] = tot[0];
    out[1] = tot[1];
    out[2] = r0;
    out[3] = r1;
    out[4] = tot[2];
    out[5] = tot[3];
    out[6] = tot[4];
    out[7] = tot[5];
  }
}

template <typename P, typename T>
int lb_pair_stats(P& pl, const T* g, const T* gp, const T* d, double t, T* y, T* sv, int64_t n, double* out8_dev) {
  SI_CHECK(g && gp && d && y && sv && out8_dev && n > 0, SPECINV_EINVAL, "bad arguments");
  const int nb = (int)std::min<int64_t>(1024, std::max<int64_t>(1, ceil_div(n, 256 * 8)));
  SI_TRY(pl.lb_part.reserve((size_t)4 * 4 * 1024 * sizeof(double)));
  double* part = pl.lb_part.template as<double>() + 2 * 4 * 1024;       // slots 2-3 of 4
  hipLaunchKernelGGL((k_lbfgs_pair_stats<T>), dim3(nb), dim3(256), 0, pl.stream, g, gp, d, (T)t, y, sv, n, part);
  SI_HIP(hipGetLastError());
  hipLaunchKernelGGL(k_finish_pair_stats, dim3(1), dim3(256), 0, pl.stream, part, nb, out8_dev);
  SI_HIP(hipGetLastError());
  return SPECINV_OK;
}

// Results go to the host (`out_host`: the call synchronises) or to device memory (`out_dev`: nothing waits; the partial
// sums then live in a scratch of their own, `pl.lb_part`, slot `part_slot`, so that back-to-back passes do not share one).
template <typename P, typename T>
int lb_pair(P& pl, const T* g, const T* gp, const T* d, double t, T* y, T* sv, int64_t n, double* out2_host,
            double* out4_dev = nullptr) {
  SI_CHECK(g && gp && d && y && sv && (out2_host || out4_dev) && n > 0, SPECINV_EINVAL, "bad arguments");
  const int nb = (int)std::min<int64_t>(1024, std::max<int64_t>(1, ceil_div(n, 256 * 8)));
  SI_TRY(pl.lb_part.reserve((size_t)4 * 4 * 1024 * sizeof(double)));
  double* part = pl.lb_part.template as<double>();                      // slot 0 of 4
  hipLaunchKernelGGL((k_lbfgs_pair<T>), dim3(nb), dim3(256), 0, pl.stream, g, gp, d, (T)t, y, sv, n, part);
  SI_HIP(hipGetLastError());
  hipLaunchKernelGGL(k_finish_partials, dim3(1), dim3(256), 0, pl.stream, part, (int64_t)nb, 4,
                     out4_dev ? out4_dev : pl.sums.template as<double>());
  SI_HIP(hipGetLastError());
  if (out4_dev) return SPECINV_OK;
  SI_HIP(hipMemcpyAsync(out2_host, pl.sums.p, 2 * sizeof(double), hipMemcpyDeviceToHost, pl.stream));
  SI_HIP(si_stream_wait_short(pl.stream));
  return SPECINV_OK;
}

template <typename P, typename T>
int lb_stats(P& pl, const T* g, const T* d, int64_t n, double* out4_host, double* out4_dev = nullptr) {
  SI_CHECK(g && d && (out4_host || out4_dev) && n > 0, SPECINV_EINVAL, "bad arguments");
  const int nb = (int)std::min<int64_t>(1024, std::max<int64_t>(1, ceil_div(n, 256 * 8)));
  SI_TRY(pl.lb_part.reserve((size_t)4 * 4 * 1024 * sizeof(double)));
  double* part = pl.lb_part.template as<double>() + 4 * 1024;           // slot 1 of 4
  hipLaunchKernelGGL((k_lbfgs_stats<T>), dim3(nb), dim3(256), 0, pl.stream, g, d, n, part);
  SI_HIP(hipGetLastError());
  hipLaunchKernelGGL(k_finish_stats, dim3(1), dim3(256), 0, pl.stream, part, nb, out4_dev ? out4_dev : pl.sums.template as<double>());
  SI_HIP(hipGetLastError());
  if (out4_dev) return SPECINV_OK;
  SI_HIP(hipMemcpyAsync(out4_host, pl.sums.p, 4 * sizeof(double), hipMemcpyDeviceToHost, pl.stream));
  SI_HIP(si_stream_wait_short(pl.stream));
  return SPECINV_OK;
}

template <typename P, typename T>
int lb_dot(P& pl, const T* a, const T* b, int64_t n, double* out) {
  SI_CHECK(a && b && out && n > 0, SPECINV_EINVAL, "bad arguments");
  const int nb = (int)std::min<int64_t>(1024, std::max<int64_t>(1, ceil_div(n, 256 * 8)));
  SI_TRY(pl.partials.reserve(std::max<size_t>((size_t)nb, 3 * 1024) * sizeof(double)));
  hipLaunchKernelGGL((k_dot_partials<T>), dim3(nb), dim3(256), 0, pl.stream, a, b, n, pl.partials.template as<double>());
  SI_HIP(hipGetLastError());
  hipLaunchKernelGGL(k_finish_partials, dim3(1), dim3(256), 0, pl.stream, pl.partials.template as<double>(), (int64_t)nb, 1,
                     pl.sums.template as<double>());
  SI_HIP(hipGetLastError());
  SI_HIP(hipMemcpyAsync(out, pl.sums.p, sizeof(double), hipMemcpyDeviceToHost, pl.stream));
  SI_HIP(si_stream_wait_short(pl.stream));
  return SPECINV_OK;
}

template <typename P, typename T>
int lb_axpy(P& pl, T alpha, const T* x, T* y, int64_t n) {
  SI_CHECK(x && y && n > 0, SPECINV_EINVAL, "bad arguments");
  hipLaunchKernelGGL((k_axpy<T>), dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, pl.stream, alpha, x, y, n);
  SI_HIP(hipGetLastError());
  return SPECINV_OK;
}

template <typename P, typename T>
int lb_scale(P& pl, T alpha, const T* x, T* y, int64_t n) {
  SI_CHECK(x && y && n > 0, SPECINV_EINVAL, "bad arguments");
  hipLaunchKernelGGL((k_scale<T>), dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, pl.stream, alpha, x, y, n);
  SI_HIP(hipGetLastError());
  return SPECINV_OK;
}

template <typename P, typename T>
int lb_absmax_abssum(P& pl, const T* x, int64_t n, double* out) {
  SI_CHECK(x && out && n > 0, SPECINV_EINVAL, "bad arguments");
  const int nb = (int)std::min<int64_t>(1024, std::max<int64_t>(1, ceil_div(n, 256 * 8)));
  SI_TRY(pl.partials.reserve(std::max<size_t>((size_t)nb * 2, 3 * 1024) * sizeof(double)));
  hipLaunchKernelGGL((k_abs_partials<T>), dim3(nb), dim3(256), 0, pl.stream, x, n, pl.partials.template as<double>());
  SI_HIP(hipGetLastError());
  hipLaunchKernelGGL(k_finish_absmax, dim3(1), dim3(256), 0, pl.stream, pl.partials.template as<double>(), nb,
                     pl.sums.template as<double>());
  SI_HIP(hipGetLastError());
  SI_HIP(hipMemcpyAsync(out, pl.sums.p, 2 * sizeof(double), hipMemcpyDeviceToHost, pl.stream));
  SI_HIP(si_stream_wait_short(pl.stream));
  return SPECINV_OK;
}

// ---- many vectors in one pass (the L-BFGS recursion written on Gram matrices needs g . v_j for the whole memory and one
// linear combination of it: two passes over the 2 m vectors instead of four dependent ones per pair) ----------------
constexpr int kMultiVec = 64;   // vectors per launch (their addresses travel as kernel arguments)
template <typename T>
struct MultiVecArgs {
  const T* v[kMultiVec];
  double c[kMultiVec];
  int k;
};

// part[j * gridDim.x + block] = sum over the block's elements of g[e] * v_j[e]  (float64 accumulation, fixed order).
// A block keeps kSlab elements per thread of g in registers and streams the k vectors past them, 16 bytes per lane and
// load; one wave reduction per (block pass, j).
template <typename T>
__global__ __launch_bounds__(256) void k_multi_dot(const T* __restrict__ g, MultiVecArgs<T> a, int64_t n,
                                                   double* __restrict__ part) {
  constexpr int W = 16 / sizeof(T);                 // elements per 16-byte load
  constexpr int Q = 8;                              // 16-byte pieces per thread and pass
  typedef T VT __attribute__((ext_vector_type(W)));
  __shared__ double acc[4][kMultiVec];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int j = lane; j < a.k; j += 64) acc[wv][j] = 0.0;
  const int64_t nv = n / W;                         // whole 16-byte pieces; the tail is handled by block 0 below
  const int64_t pass = (int64_t)blockDim.x * Q;
  for (int64_t base = (int64_t)blockIdx.x * pass; base < nv; base += (int64_t)gridDim.x * pass) {
    VT gv[Q];
#pragma unroll
    for (int e = 0; e < Q; ++e) {
      const int64_t i = base + (int64_t)e * blockDim.x + threadIdx.x;
      if (i < nv) gv[e] = reinterpret_cast<const VT*>(g)[i];
      else
        for (int c = 0; c < W; ++c) gv[e][c] = T(0);
    }
    for (int j = 0; j < a.k; ++j) {
      const VT* __restrict__ v = reinterpret_cast<const VT*>(a.v[j]);
      VT vv[Q];
#pragma unroll
      for (int e = 0; e < Q; ++e) {
        const int64_t i = base + (int64_t)e * blockDim.x + threadIdx.x;
        if (i < nv) vv[e] = v[i];
        else
          for (int c = 0; c < W; ++c) vv[e][c] = T(0);
      }
      double s = 0.0;
#pragma unroll
      for (int e = 0; e < Q; ++e)
#pragma unroll
        for (int c = 0; c < W; ++c) s += (double)gv[e][c] * (double)vv[e][c];
      s = wave_sum(s);
      if (lane == 0) acc[wv][j] += s;               // wave-private slot: no atomics, fixed order
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {        // the n % W trailing elements
    for (int j = 0; j < a.k; ++j) {
      double s = 0.0;
      for (int64_t i = nv * W; i < n; ++i) s += (double)g[i] * (double)a.v[j][i];
      acc[0][j] += s;
    }
  }
  __syncthreads();
  for (int j = threadIdx.x; j < a.k; j += blockDim.x)
    part[(int64_t)j * gridDim.x + blockIdx.x] = ((acc[0][j] + acc[1][j]) + acc[2][j]) + acc[3][j];
}

// out[j] = sum_b part[j * nb + b]
static __global__ void k_multi_finish(const double* __restrict__ part, int nb, double* __restrict__ out) {
  __shared__ double red[16];
  const int j = blockIdx.x;
  double s = 0;
  for (int i = threadIdx.x; i < nb; i += blockDim.x) s += part[(int64_t)j * nb + i];
  const double t = block_sum(s, red);
  if (threadIdx.x == 0) out[j] = t;
}

// out[e] = (accumulate ? out[e] : 0) + sum_j c_j * v_j[e], summed in float64 in the order of j, rounded once; one
// 16-byte piece per thread (the trailing n % W elements by the last thread)
// With `xs` given the same pass also takes the step xs += t * out (the rounded direction: what k_axpy would read back).
template <typename T>
__global__ __launch_bounds__(256) void k_lincomb(MultiVecArgs<T> a, int accumulate, T* __restrict__ out, int64_t n,
                                                 T t = T(0), T* __restrict__ xs = nullptr) {
  constexpr int W = 16 / sizeof(T);
  typedef T VT __attribute__((ext_vector_type(W)));
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t nv = n / W;
  if (i < nv) {
    double s[W];
    const VT o = accumulate ? reinterpret_cast<const VT*>(out)[i] : VT(0);
#pragma unroll
    for (int c = 0; c < W; ++c) s[c] = (double)o[c];
#pragma unroll 4
    for (int j = 0; j < a.k; ++j) {
      const VT v = reinterpret_cast<const VT*>(a.v[j])[i];
#pragma unroll
      for (int c = 0; c < W; ++c) s[c] += a.c[j] * (double)v[c];
    }
    VT r;
#pragma unroll
    for (int c = 0; c < W; ++c) r[c] = (T)s[c];
    reinterpret_cast<VT*>(out)[i] = r;
    if (xs != nullptr) {
      VT xv = reinterpret_cast<const VT*>(xs)[i];
#pragma unroll
      for (int c = 0; c < W; ++c) xv[c] = fma(t, r[c], xv[c]);
      reinterpret_cast<VT*>(xs)[i] = xv;
    }
  } else if (i == nv) {
    for (int64_t e = nv * W; e < n; ++e) {
      double s = accumulate ? (double)out[e] : 0.0;
      for (int j = 0; j < a.k; ++j) s += a.c[j] * (double)a.v[j][e];
      out[e] = (T)s;
      if (xs != nullptr) xs[e] = fma(t, (T)s, xs[e]);
    }
  }
}

template <typename P, typename T>
int lb_multi_dot(P& pl, const T* g, const void* const* vecs, int k, int64_t n, double* out, double* out_dev = nullptr) {
  SI_CHECK(g && vecs && (out || out_dev) && k > 0 && n > 0, SPECINV_EINVAL, "bad arguments");
  SI_CHECK(((uintptr_t)g & 15) == 0, SPECINV_EINVAL, "g is not 16-byte aligned");
  const int nb = (int)std::min<int64_t>(1024, std::max<int64_t>(1, ceil_div(n, 256 * 8 * 4)));
  SI_TRY(pl.partials.reserve(std::max<size_t>((size_t)nb * kMultiVec, 3 * 1024) * sizeof(double)));
  SI_TRY(pl.lb_scal.reserve((size_t)std::max(k, 4) * sizeof(double)));
  double* dots = out_dev ? out_dev : pl.lb_scal.template as<double>();
  for (int j0 = 0; j0 < k; j0 += kMultiVec) {
    MultiVecArgs<T> a{};
    a.k = std::min(kMultiVec, k - j0);
    for (int j = 0; j < a.k; ++j) {
      SI_CHECK(vecs[j0 + j] != nullptr && ((uintptr_t)vecs[j0 + j] & 15) == 0, SPECINV_EINVAL,
               "vector %d is NULL or not 16-byte aligned", j0 + j);
      a.v[j] = static_cast<const T*>(vecs[j0 + j]);
    }
    hipLaunchKernelGGL((k_multi_dot<T>), dim3(nb), dim3(256), 0, pl.stream, g, a, n, pl.partials.template as<double>());
    hipLaunchKernelGGL(k_multi_finish, dim3(a.k), dim3(256), 0, pl.stream, pl.partials.template as<double>(), nb, dots + j0);
    SI_HIP(hipGetLastError());
  }
  if (out_dev) return SPECINV_OK;
  SI_HIP(hipMemcpyAsync(out, dots, (size_t)k * sizeof(double), hipMemcpyDeviceToHost, pl.stream));
  SI_HIP(si_stream_wait_short(pl.stream));
  return SPECINV_OK;
}

template <typename P, typename T>
int lb_lincomb(P& pl, const void* const* vecs, const double* coef, int k, int64_t n, T* out, double t = 0.0, T* xs = nullptr) {
  SI_CHECK(vecs && coef && out && k > 0 && n > 0, SPECINV_EINVAL, "bad arguments");
  SI_CHECK(((uintptr_t)out & 15) == 0, SPECINV_EINVAL, "out is not 16-byte aligned");
  SI_CHECK(xs == nullptr || ((uintptr_t)xs & 15) == 0, SPECINV_EINVAL, "x is not 16-byte aligned");
  for (int j0 = 0; j0 < k; j0 += kMultiVec) {
    MultiVecArgs<T> a{};
    a.k = std::min(kMultiVec, k - j0);
    for (int j = 0; j < a.k; ++j) {
      SI_CHECK(vecs[j0 + j] != nullptr && ((uintptr_t)vecs[j0 + j] & 15) == 0, SPECINV_EINVAL,
               "vector %d is NULL or not 16-byte aligned", j0 + j);
      a.v[j] = static_cast<const T*>(vecs[j0 + j]);
      a.c[j] = coef[j0 + j];
    }
    const int64_t pieces = n / (16 / (int64_t)sizeof(T)) + 1;      // + the thread that takes the trailing elements
    const bool last = j0 + kMultiVec >= k;                         // the step rides on the launch that completes the sum
    hipLaunchKernelGGL((k_lincomb<T>), dim3((unsigned)ceil_div(pieces, 256)), dim3(256), 0, pl.stream, a, j0 > 0 ? 1 : 0, out, n,
                       (T)t, last ? xs : nullptr);
    SI_HIP(hipGetLastError());
  }
  return SPECINV_OK;
}

// ---- two-loop recursion with device-resident scalars -----------------------------------------------------
// slot = scale * sum(partials)      (al_i = rho_i * (s_i . q))
static __global__ void k_finish_scaled(const double* __restrict__ part, int n, double scale, double* __restrict__ slot) {
  __shared__ double red[16];
  double s = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += part[i];
  const double t = block_sum(s, red);
  if (threadIdx.x == 0) *slot = scale * t;
}

// y += (sa * a[0] + sb * (b ? b[0] : 0)) * x    with the coefficient read from device memory
template <typename T>
__global__ void k_axpy_dev(const double* __restrict__ a, double sa, const double* __restrict__ b, double sb,
                           const T* __restrict__ x, T* __restrict__ y, int64_t n) {
  const T c = (T)(sa * a[0] + (b ? sb * b[0] : 0.0));
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = y[i] + c * x[i];
}

template <typename P, typename T>
int lb_direction(P& pl, const T* g, const void* const* s_list, const void* const* y_list, const double* rho, int m,
                 double h_diag, T* d, int64_t n) {
  SI_CHECK(g && d && n > 0 && m >= 0, SPECINV_EINVAL, "bad arguments");
  SI_CHECK(m == 0 || (s_list && y_list && rho), SPECINV_EINVAL, "history arrays are NULL");
  const int nb = (int)std::min<int64_t>(1024, std::max<int64_t>(1, ceil_div(n, 256 * 8)));
  SI_TRY(pl.partials.reserve(std::max<size_t>((size_t)nb, 3 * 1024) * sizeof(double)));
  SI_TRY(pl.lb_scal.reserve((size_t)(m + 2) * sizeof(double)));
  double* al = pl.lb_scal.template as<double>();      // al[0..m-1], be at al[m]
  double* part = pl.partials.template as<double>();
  const dim3 ge((unsigned)ceil_div(n, 256)), blk(256);
  hipLaunchKernelGGL((k_scale<T>), ge, blk, 0, pl.stream, T(-1), g, d, n);                       // q = -g
  for (int i = m - 1; i >= 0; --i) {
    const T* si = static_cast<const T*>(s_list[i]);
    const T* yi = static_cast<const T*>(y_list[i]);
    hipLaunchKernelGGL((k_dot_partials<T>), dim3(nb), blk, 0, pl.stream, si, static_cast<const T*>(d), n, part);
    hipLaunchKernelGGL(k_finish_scaled, dim3(1), blk, 0, pl.stream, part, nb, rho[i], al + i);   // al_i
    hipLaunchKernelGGL((k_axpy_dev<T>), ge, blk, 0, pl.stream, al + i, -1.0, (const double*)nullptr, 0.0, yi, d, n);
  }
  hipLaunchKernelGGL((k_scale<T>), ge, blk, 0, pl.stream, (T)h_diag, static_cast<const T*>(d), d, n);   // r = H0 q
  for (int i = 0; i < m; ++i) {
    const T* si = static_cast<const T*>(s_list[i]);
    const T* yi = static_cast<const T*>(y_list[i]);
    hipLaunchKernelGGL((k_dot_partials<T>), dim3(nb), blk, 0, pl.stream, yi, static_cast<const T*>(d), n, part);
    hipLaunchKernelGGL(k_finish_scaled, dim3(1), blk, 0, pl.stream, part, nb, rho[i], al + m);   // be_i
    hipLaunchKernelGGL((k_axpy_dev<T>), ge, blk, 0, pl.stream, al + i, 1.0, al + m, -1.0, si, d, n);
  }
  SI_HIP(hipGetLastError());
  return SPECINV_OK;
}

// ---- mel contractions on the matrix cores ----------------------------------------------------------------
// One wave owns a 32 x 32 output tile and feeds v_mfma_f32_32x32x2_f32 (A: lane l holds A[l&31][l>>5],
// B: lane l holds B[l>>5][l&31]; C/D: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5)).  Operands are staged
// through LDS in 32 x 32 tiles with a one-dword row pad (conflict-free column reads).
using f32x16 = float __attribute__((ext_vector_type(16)));

__device__ inline f32x16 mfma_32x32x2(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

// mm[bt, m] = sum_f Mel[m, f] * |S[bt, f]|  ->  V = log1p(mm).   S: (BT, F) complex, Mel: (n_mels, F).
// grid (ceil(BT/32), ceil(n_mels/32)), one wave per block.  Float32 only.
static __global__ __launch_bounds__(64) void k_mel_forward_mfma(const cplx<float>* __restrict__ spec, const float* __restrict__ mel,
                                                         float* __restrict__ mm_out, int64_t BT, int F, int n_mels) {
  __shared__ float sa[32][33];   // Mel tile  [m][k]
  __shared__ float sb[32][33];   // |S| tile  [bt][k]
  const int lane = threadIdx.x;
  const int64_t bt0 = (int64_t)blockIdx.x * 32;
  const int m0 = blockIdx.y * 32;
  f32x16 acc = {0};
  for (int k0 = 0; k0 < F; k0 += 32) {
    // 32 x 32 tiles, 16 elements per lane, rows contiguous in k (coalesced 128-byte rows)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int r = i * 2 + (lane >> 5), c = lane & 31;
      const int k = k0 + c;
      float va = 0.f, vb = 0.f;
      if (k < F) {
        if (m0 + r < n_mels) va = mel[(int64_t)(m0 + r) * F + k];
        if (bt0 + r < BT) {
          const cplx<float> s = spec[(bt0 + r) * F + k];
          vb = hypotf(s.x, s.y);
        }
      }
      sa[r][c] = va;
      sb[r][c] = vb;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 32; kk += 2) {
      const float a = sa[lane & 31][kk + (lane >> 5)];
      const float b = sb[lane & 31][kk + (lane >> 5)];
      acc = mfma_32x32x2(a, b, acc);
    }
    __syncthreads();
  }
  // D[row = m][col = bt]
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    const int col = lane & 31;
    if (m0 + row < n_mels && bt0 + col < BT) mm_out[(bt0 + col) * n_mels + m0 + row] = acc[r];
  }
}

// Same contraction, one workgroup per 32 frames and ALL mel rows (MT tiles of 32): the spectrum - the large operand -
// is read once.  The four waves split K (wave w takes the k-steps w, w+4, ...), each with wave-private LDS tiles
// (no workgroup barrier inside the K loop: the LDS queue of a wave is in order), and add their accumulators through
// LDS at the end.  The filterbank is read from a copy tiled per k-step as [k][m] (zero padded to 32 k x MT*32 m,
// k_mel_tile): one k-step is MT*4 KB of contiguous, 16-byte aligned data - 16-byte loads, 16-byte LDS stores, and
// conflict-free operand reads (consecutive m in consecutive lanes).  The |S| tile is kept [k][bt] with a padded row
// for the same reason.  The next k-step's spectrum values are fetched before the MFMA block of the current one.
template <int MT>
__global__ __launch_bounds__(256) void k_mel_forward_splitk(const cplx<float>* __restrict__ spec,
                                                            const float* __restrict__ mel_tiled, float* __restrict__ mm_out,
                                                            int64_t BT, int F, int n_mels) {
  using f4 = float __attribute__((ext_vector_type(4)));
  constexpr int MW = MT * 32;                      // mel rows per k
  extern __shared__ __attribute__((aligned(16))) float mel_smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* sa = mel_smem + (size_t)wave * (32 * MW + 32 * 33);   // Mel tile [k][m]
  float* sb = sa + 32 * MW;                                    // |S| tile [k][bt], row stride 33
  const int64_t bt0 = (int64_t)blockIdx.x * 32;
  f32x16 acc[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) acc[t] = f32x16{0};
  const int ksteps = (F + 31) / 32;
  const int c = lane & 31, rh = lane >> 5;

  cplx<float> sv[16];
  auto fetch = [&](int ks) {
    const int k = ks * 32 + c;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int r = i * 2 + rh;
      sv[i] = (k < F && bt0 + r < BT) ? spec[(bt0 + r) * F + k] : mk<float>(0.f, 0.f);
    }
  };
  if (wave < ksteps) fetch(wave);
  for (int ks = wave; ks < ksteps; ks += 4) {
    const f4* mg = reinterpret_cast<const f4*>(mel_tiled + (size_t)ks * 32 * MW);
    f4 mv[MT * 4];
#pragma unroll
    for (int i = 0; i < MT * 4; ++i) mv[i] = mg[i * 64 + lane];
#pragma unroll
    for (int i = 0; i < 16; ++i) sb[c * 33 + i * 2 + rh] = __builtin_amdgcn_sqrtf(fmaf(sv[i].x, sv[i].x, sv[i].y * sv[i].y));
#pragma unroll
    for (int i = 0; i < MT * 4; ++i) reinterpret_cast<f4*>(sa)[i * 64 + lane] = mv[i];
    if (ks + 4 < ksteps) fetch(ks + 4);                  // in flight during the MFMA block
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's tiles are in LDS (and the compiler keeps the order)
#pragma unroll
    for (int kk = 0; kk < 32; kk += 2) {
      const float b = sb[(kk + rh) * 33 + c];
#pragma unroll
      for (int t = 0; t < MT; ++t) acc[t] = mfma_32x32x2(sa[(kk + rh) * MW + t * 32 + c], b, acc[t]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // tile reads done before the next k-step overwrites them
  }
  __syncthreads();
  float* red = mel_smem;                                 // [wave][MT][16][64]
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[((wave * MT + t) * 16 + r) * 64 + lane] = acc[t][r];
  __syncthreads();
  for (int idx = threadIdx.x; idx < MT * 16 * 64; idx += 256) {
    const int l = idx & 63, r = (idx >> 6) & 15, t = idx >> 10;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) v += red[((w * MT + t) * 16 + r) * 64 + l];
    const int m = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    const int64_t bt = bt0 + (l & 31);
    if (m < n_mels && bt < BT) mm_out[bt * n_mels + m] = v;
  }
}

// mel (n_mels, F) -> tiled[ks][k][m] with m padded to mw and k to 32 * ksteps (zeros)
static __global__ void k_mel_tile(const float* __restrict__ mel, float* __restrict__ tiled, int F, int n_mels, int mw, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int m = i % mw;
  const int64_t k = i / mw;
  tiled[i] = (m < n_mels && k < F) ? mel[(int64_t)m * F + k] : 0.f;
}

// dA[bt, f] = sum_m dM[bt, m] * Mel[m, f] ; G[bt, f] = dA * S/|S| * (interior ? 1/2 : 1)   (in place over S)
// grid (ceil(BT/32), ceil(F/32)), one wave per block.
static __global__ __launch_bounds__(64) void k_mel_backward_mfma(cplx<float>* __restrict__ spec, const float* __restrict__ mel,
                                                          const float* __restrict__ dM, int64_t BT, int F, int n_mels,
                                                          int n_fft, int onesided) {
  __shared__ float sa[32][33];   // dM tile  [bt][m]
  __shared__ float sb[32][33];   // Mel tile [m][f]
  const int lane = threadIdx.x;
  const int64_t bt0 = (int64_t)blockIdx.x * 32;
  const int f0 = blockIdx.y * 32;
  f32x16 acc = {0};
  for (int k0 = 0; k0 < n_mels; k0 += 32) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int r = i * 2 + (lane >> 5), c = lane & 31;
      float va = 0.f, vb = 0.f;
      if (bt0 + r < BT && k0 + c < n_mels) va = dM[(bt0 + r) * n_mels + k0 + c];
      if (k0 + r < n_mels && f0 + c < F) vb = mel[(int64_t)(k0 + r) * F + f0 + c];
      sa[r][c] = va;
      sb[r][c] = vb;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 32; kk += 2) {
      const float a = sa[lane & 31][kk + (lane >> 5)];   // A[i = bt][k = m]
      const float b = sb[kk + (lane >> 5)][lane & 31];   // B[k = m][j = f]
      acc = mfma_32x32x2(a, b, acc);
    }
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);   // bt
    const int f = f0 + (lane & 31);
    if (bt0 + row < BT && f < F) {
      const int64_t idx = (bt0 + row) * F + f;
      const cplx<float> s = spec[idx];
      const float mag = hypotf(s.x, s.y);
      float g = mag > 0.f ? acc[r] / mag : 0.f;
      if (onesided && f != 0 && 2 * f != n_fft) g *= 0.5f;
      spec[idx] = mk<float>(s.x * g, s.y * g);
    }
  }
}

// Same contraction per workgroup of 32 frames: the dM tile (32 frames x all mel rows) is staged once as [m][bt]; each
// wave then walks frequency tiles (wave w takes tiles w, w+4, ...), reading the filterbank from a copy tiled per
// frequency tile as [m][f] (k_mel_tile_t) and the spectrum values of the tile before the MFMA block.
template <int MT>
__global__ __launch_bounds__(256) void k_mel_backward_tiles(cplx<float>* __restrict__ spec, const float* __restrict__ mel_tiled_t,
                                                            const float* __restrict__ dM, int64_t BT, int F, int n_mels,
                                                            int n_fft, int onesided) {
  using f4 = float __attribute__((ext_vector_type(4)));
  constexpr int MW = MT * 32;
  extern __shared__ __attribute__((aligned(16))) float mel_smem[];
  float* sd = mel_smem;                                   // dM tile [m][bt], row stride 33
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* sm = mel_smem + MW * 33 + (size_t)wave * MW * 32;   // this wave's Mel tile [m][f]
  const int64_t bt0 = (int64_t)blockIdx.x * 32;
  const int c = lane & 31, rh = lane >> 5;
  for (int e = threadIdx.x; e < 32 * MW; e += 256) {
    const int bt = e / MW, m = e - bt * MW;
    sd[m * 33 + bt] = (bt0 + bt < BT && m < n_mels) ? dM[(bt0 + bt) * n_mels + m] : 0.f;
  }
  __syncthreads();
  const int ftiles = (F + 31) / 32;
  cplx<float> sv[16], sn[16];
  f4 mv[MT * 4], mn[MT * 4];
  auto fetch = [&](int ft, cplx<float>(&svv)[16], f4(&mvv)[MT * 4]) {
    const int f = ft * 32 + c;
    const f4* mg = reinterpret_cast<const f4*>(mel_tiled_t + (size_t)ft * MW * 32);
#pragma unroll
    for (int i = 0; i < MT * 4; ++i) mvv[i] = mg[i * 64 + lane];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * rh;
      svv[r] = (bt0 + row < BT && f < F) ? spec[(bt0 + row) * F + f] : mk<float>(0.f, 0.f);
    }
  };
  if (wave < ftiles) fetch(wave, sv, mv);
  for (int ft = wave; ft < ftiles; ft += 4) {
    const int f = ft * 32 + c;
#pragma unroll
    for (int i = 0; i < MT * 4; ++i) reinterpret_cast<f4*>(sm)[i * 64 + lane] = mv[i];
    if (ft + 4 < ftiles) fetch(ft + 4, sn, mn);          // the next tile's operands fly during this tile's MFMA block
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    f32x16 acc = {0};
#pragma unroll
    for (int kk = 0; kk < MW; kk += 2) acc = mfma_32x32x2(sd[(kk + rh) * 33 + c], sm[(kk + rh) * 32 + c], acc);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * rh;
      if (bt0 + row < BT && f < F) {
        const float mag = __builtin_amdgcn_sqrtf(fmaf(sv[r].x, sv[r].x, sv[r].y * sv[r].y));
        float g = mag > 0.f ? acc[r] / mag : 0.f;
        if (onesided && f != 0 && 2 * f != n_fft) g *= 0.5f;
        spec[(bt0 + row) * F + f] = mk<float>(sv[r].x * g, sv[r].y * g);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) sv[r] = sn[r];
#pragma unroll
    for (int i = 0; i < MT * 4; ++i) mv[i] = mn[i];
  }
}

// mel (n_mels, F) -> tiled_t[ft][m][f] with m padded to mw and f to 32 * ftiles (zeros)
static __global__ void k_mel_tile_t(const float* __restrict__ mel, float* __restrict__ tiled, int F, int n_mels, int mw, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int fl = i & 31;
  const int m = (i >> 5) % mw;
  const int64_t ft = i / ((int64_t)32 * mw);
  const int64_t f = ft * 32 + fl;
  tiled[i] = (m < n_mels && f < F) ? mel[(int64_t)m * F + f] : 0.f;
}

// ---- elementwise pieces -----------------------------------------------------------------------------------------
// V = |S| in user layout (B, F, T) from S (B, T, F)
template <typename T>
__global__ void k_mag_to_user(const cplx<T>* __restrict__ spec, T* __restrict__ v, int Bn, int Tn, int F) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over (b, f, t)
  if (i >= (int64_t)Bn * F * Tn) return;
  const int t = i % Tn;
  const int f = (i / Tn) % F;
  const int64_t b = i / ((int64_t)Tn * F);
  const cplx<T> s = spec[(b * Tn + t) * F + f];
  v[i] = si_hypot(s.x, s.y);
}

// V = log1p(mm) in user layout (B, n_mels, T) from mm (B*T, n_mels)
template <typename T>
__global__ void k_log1p_to_user(const T* __restrict__ mm, T* __restrict__ v, int Bn, int Tn, int n_mels) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over (b, m, t)
  if (i >= (int64_t)Bn * n_mels * Tn) return;
  const int t = i % Tn;
  const int m = (i / Tn) % n_mels;
  const int64_t b = i / ((int64_t)Tn * n_mels);
  v[i] = log1p(mm[(b * Tn + t) * n_mels + m]);
}

// MAG transform: loss partials and G = (2/numel) (|S| - T) S/|S| * (interior ? 1/2 : 1) in place over S
template <typename T>
__global__ void k_mag_loss_grad(cplx<T>* __restrict__ spec, const T* __restrict__ target, int target_btf, int Bn, int Tn, int F,
                                int n_fft, int onesided, double inv_numel, double* __restrict__ part) {
  __shared__ double red[16];
  double s2 = 0;
  const int64_t total = (int64_t)Bn * Tn * F;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int f = i % F;
    const int t = (i / F) % Tn;
    const int64_t b = i / ((int64_t)F * Tn);
    const cplx<T> s = spec[i];
    const T mag = si_hypot(s.x, s.y);
    const T d = mag - (target_btf ? target[i] : target[(b * F + f) * Tn + t]);
    s2 += (double)d * (double)d;
    T g = mag > T(0) ? (T)(2.0 * inv_numel) * d / mag : T(0);
    if (onesided && f != 0 && 2 * f != n_fft) g *= T(0.5);
    spec[i] = mk<T>(s.x * g, s.y * g);
  }
  const double t = block_sum(s2, red);
  if (threadIdx.x == 0) part[blockIdx.x] = t;
}

// LOGMEL: loss partials and dM = (2/numel) (log1p(mm) - T) / (1 + mm), in place over mm (BT, n_mels)
template <typename T>
__global__ void k_logmel_loss_dm(T* __restrict__ mm, const T* __restrict__ target, int Bn, int Tn, int n_mels,
                                 double inv_numel, double* __restrict__ part) {
  __shared__ double red[16];
  double s2 = 0;
  const int64_t total = (int64_t)Bn * Tn * n_mels;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int m = i % n_mels;
    const int t = (i / n_mels) % Tn;
    const int64_t b = i / ((int64_t)n_mels * Tn);
    const T v = mm[i];
    const T d = log1p(v) - target[(b * n_mels + m) * Tn + t];
    s2 += (double)d * (double)d;
    mm[i] = (T)(2.0 * inv_numel) * d / (T(1) + v);
  }
  const double t = block_sum(s2, red);
  if (threadIdx.x == 0) part[blockIdx.x] = t;
}

// float64 / generic mel contractions (VALU): thread per output
template <typename T>
__global__ void k_mel_forward_valu(const cplx<T>* __restrict__ spec, const T* __restrict__ mel, T* __restrict__ mm,
                                   int64_t BT, int F, int n_mels) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= BT * n_mels) return;
  const int m = i % n_mels;
  const int64_t bt = i / n_mels;
  T acc = 0;
  for (int f = 0; f < F; ++f) {
    const cplx<T> s = spec[bt * F + f];
    acc += mel[(int64_t)m * F + f] * si_hypot(s.x, s.y);
  }
  mm[i] = acc;
}

template <typename T>
__global__ void k_mel_backward_valu(cplx<T>* __restrict__ spec, const T* __restrict__ mel, const T* __restrict__ dM,
                                    int64_t BT, int F, int n_mels, int n_fft, int onesided) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= BT * F) return;
  const int f = i % F;
  const int64_t bt = i / F;
  T acc = 0;
  for (int m = 0; m < n_mels; ++m) acc += dM[bt * n_mels + m] * mel[(int64_t)m * F + f];
  const cplx<T> s = spec[i];
  const T mag = si_hypot(s.x, s.y);
  T g = mag > T(0) ? acc / mag : T(0);
  if (onesided && f != 0 && 2 * f != n_fft) g *= T(0.5);
  spec[i] = mk<T>(s.x * g, s.y * g);
}

// inverse frames (generic): windowed Hermitian inverse transform of a (B, T, F) spectrum, scale = c.inv_scale.
// Used by _istft (scale 1/N) and by the STFT adjoint of the L_BFGS gradient (scale = forward scale).
template <typename T, bool IP = false>
__global__ void k_grad_frames(FrameCfg<T> c, const cplx<T>* __restrict__ g, T* __restrict__ frames) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cplx<T>* a = reinterpret_cast<cplx<T>*>(smem);
  cplx<T>* b = IP ? a : a + c.n_fft;
  const int t = blockIdx.x, bi = blockIdx.y;
  const cplx<T>* in = g + ((int64_t)bi * c.n_frames + t) * c.n_freq;
  for (int f = threadIdx.x; f < c.n_freq; f += blockDim.x) a[f] = in[f];
  __syncthreads();
  spectrum_to_frame<T, IP>(c, a, b, frames + ((int64_t)bi * c.n_frames + t) * c.n_fft, c.window);
}

// Fold of the padded margins onto the signal: grad already holds the plain overlap-add of the gradient frames over
// the signal's own positions (k_ola / k_ola_f4 without the envelope); every sample within `pad` of an edge also
// receives what the padding copied from it (reflect / replicate / circular).  One thread per margin sample.  The
// gradient w.r.t. a padded sample is gathered from `frames`, or read from `margins` (B, 2, pad: the pad samples left and
// right of the signal, written by k_hop_inverse) when that is given.
template <typename T>
__global__ void k_grad_fold_margins(const T* __restrict__ frames, T* __restrict__ grad, int n_fft, int hop, int pad,
                                    int pad_mode, int n_frames, int64_t len, int64_t rows, const T* __restrict__ margins) {
  const int64_t per_row = 2 * ((int64_t)pad + 1);
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * per_row) return;
  const int64_t bi = i / per_row, j = i - bi * per_row;
  int64_t n;
  if (j <= pad) {
    n = j;                                            // left stretch 0 .. pad
    if (n >= len) return;
  } else {
    n = len - 1 - pad + (j - pad - 1);                // right stretch len-1-pad .. len-1
    if (n <= pad || n >= len) return;                 // (short signals: already covered by the left stretch)
  }
  const T* fr = frames + bi * n_frames * n_fft;
  const int64_t covered = (int64_t)(n_frames - 1) * hop + n_fft;   // padded positions that receive any frame
  auto at = [&](int64_t np) -> T {                                 // gradient w.r.t. padded sample np
    if (np < 0 || np >= covered) return T(0);
    if (margins != nullptr) {
      const T* mg = margins + bi * 2 * pad;
      if (np < pad) return mg[np];
      const int64_t r = np - pad - len;
      return (r >= 0 && r < pad) ? mg[pad + r] : T(0);
    }
    int64_t t_hi = np / hop;
    if (t_hi > n_frames - 1) t_hi = n_frames - 1;
    const int64_t t_lo = np - n_fft + 1 <= 0 ? 0 : (np - n_fft + hop) / hop;
    T acc = 0;
    for (int64_t t = t_lo; t <= t_hi; ++t) acc += fr[t * n_fft + (np - t * hop)];
    return acc;
  };
  T g = 0;
  switch (pad_mode) {
    case SPECINV_PAD_REFLECT:
      if (n >= 1 && n <= pad) g += at(pad - n);                               // left margin i = pad - n
      if (n <= len - 2 && n >= len - 1 - pad) g += at(pad + len + (len - 2 - n));
      break;
    case SPECINV_PAD_REPLICATE:
      if (n == 0)
        for (int64_t q = 0; q < pad; ++q) g += at(q);
      if (n == len - 1)
        for (int64_t q = 0; q < pad; ++q) g += at(pad + len + q);
      break;
    case SPECINV_PAD_CIRCULAR:
      if (n >= len - pad) g += at(n - (len - pad));
      if (n < pad) g += at(pad + len + n);
      break;
    default:
      break;
  }
  grad[bi * len + n] += g;
}

// ---- host side ------------------------------------------------------------------------------------------------
// 16-row mel tiles the one-launch objective is instantiated for (0: not covered)
// (n_fft 1024 takes NINE tiles for 81 ... 128 bands, the last one empty: k_objective_logmel<8, 8> is the one instantiation the
// register allocator loses - 256 registers and 978 spilled, 0.204 ms per evaluation at B16 x T1024 where <8, 5> takes 0.113 -
// while <8, 7>, <8, 9> and <8, 10> allocate 180 ... 193 and spill nothing)
inline int obj_mel_tiles(int n_mels, int R) {
  const int mt = (n_mels + 15) / 16;
  return mt <= 3 ? 3 : mt <= 4 ? 4 : mt <= 5 ? 5 : mt <= 8 ? (R == 8 ? 9 : 8) : 0;
}

// What follows k_objective_logmel, in ONE launch of kObjRows blocks (grid-stride over the work):
//   * chunk seams: grad[n] += the previous tile's tail over the first n_fft - hop samples of tiles 1.. (k_hop_tails_raw),
//   * fold of the padded margins onto the signal (k_grad_fold_margins with `margins` given),
//   * the statistics of the gradient (fast::ObjStatReq; none: st.rows == nullptr): the pass over the seams becomes a pass over the
//     WHOLE gradient with g_prev and d beside it - every sample counted once, by the thread that finishes it - and leaves the
//     first level of the reduction tree: block j adds the squared-error sums of the tiles [j n_tiles / kObjRows, (j + 1) n_tiles /
//     kObjRows) of the objective kernel to its own figures and writes ONE row of kObjStatRow doubles - {g.d, sum|g|, y.s, y.y,
//     g.g, g.g_prev, max|g|, max|d|, squared error}, stored component-major [kObjStatRow][kObjRows]; whoever needs the figures (k_objective_finish_rows, the optimiser's
//     decision kernels) reads kObjRows rows.  (Round 4: k_lbd_pair_stats, a pass of its own after this kernel.  Taken inside the
//     objective kernel's gather instead, the statistics cost more than that pass: 0.156 against 0.124 ms per evaluation at C5 -
//     with one workgroup per CU nothing hides the loads of g_prev and d.)
//   * without statistics: loss = scale * sum(partials) by one extra block (block kObjRows), as k_finish_scaled would.
// A margin sample that also lies in a seam region is finished by its margin thread (own + tail first, then the fold: the order
// of the separate launches); the seam threads leave those samples alone, so no two threads touch the same sample.
#ifndef SPECINV_EPI_ABL              // (timing experiments, wrong results: 1 no margins, 2 no statistics arithmetic, 4 no g_prev loads, 8 no seams)
#define SPECINV_EPI_ABL 0
#endif
#ifndef SPECINV_EPI_GROUP            // threads that walk one tile together in the epilogue's statistics pass
#define SPECINV_EPI_GROUP 128      // (C5: 512 / 256 / 128 -> 122.7 / 121.2 / 120.8 ms per step)
#endif
constexpr int kObjEpiThreads = 1024;   // a streaming pass wants waves in flight: 16 per CU at one block per CU
// the tail's hand-over of the rows by write-through stores + vmcnt(0) instead of a release (see the tail below): only where the
// ISA documents it
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx942__) && !defined(__gfx950__)
constexpr bool kTailSc1Protocol = false;
#else
constexpr bool kTailSc1Protocol = true;
#endif
static __global__ __launch_bounds__(kObjEpiThreads) void k_objective_epilogue(float* __restrict__ grad, const float* __restrict__ xtail,
                                                                   const float* __restrict__ margins, const double* __restrict__ part,
                                                                   double* __restrict__ slot, int T, int nchunks, int n_fft, int hop,
                                                                   int keep, int pad, int pad_mode, int64_t len, int64_t rows,
                                                                   int64_t n_tail, int64_t n_margin, int n_part, double scale,
                                                                   fast::ObjCtl ctl, fast::ObjStatReq st, int vec_ok, int skew,
                                                                   fast::ObjDecide dec) {
  // (chunks of frames: k_objective_logmel's tiles - even, skew 0 - or k_objective_walk's, which may be skewed in pairs)
  auto hop_chunk_begin_ = [&](int c_, int T_, int n_) { return fast::chunk_begin(c_, T_, n_, skew); };
  float* grad_other = ctl.grad_alt;
  if (ctl.do_eval != nullptr) {                  // device-resident optimiser: gate and gradient ping-pong (lbfgs_dev.h)
    if (*ctl.do_eval == 0) {
      if (dec.ticket != nullptr && blockIdx.x == 0) lbd_tail_pass(dec);
      return;
    }
    if ((*ctl.cur ^ 1) != 0) {
      grad_other = grad;
      grad = ctl.grad_alt;
    }
  }
  __shared__ double red[16];
  __shared__ LbdState lbd_r;
  if (dec.ticket != nullptr) lbd_tail_preload(dec, lbd_r);
  if (st.rows == nullptr && blockIdx.x == fast::kObjRows) {
    double sacc = 0;
    for (int i = threadIdx.x; i < n_part; i += blockDim.x) sacc += part[i];
    const double t = block_sum(sacc, red);
    if (threadIdx.x == 0) *slot = scale * t;
    return;
  }
  const bool st_on = st.rows != nullptr;
  const float* st_d = st.d;
  const float* st_gp = st.gp;
  float st_t = st.t;
  bool d_impl = false;
  double c0_d = 0.0;
  if (st_on && st.have != nullptr) {
    const bool have = *st.have != 0;
    st_t = (float)*st.t_dev;
    st_gp = have ? grad_other : nullptr;
    st_d = have ? st.d : nullptr;
    if (have && st.d_implicit != nullptr && *st.d_implicit != 0) {    // d = (float)(c0 (double)g_prev): recomputed, not read
      d_impl = true;
      c0_d = *st.c0_d;
      st_d = nullptr;
    }
  }
  auto d_of = [&](float gpv) { return (float)(c0_d * (double)gpv); };
  fast::ObjStatAcc sta;
  const bool fold = pad > 0 && pad_mode != SPECINV_PAD_CONSTANT;
  const bool seams = keep > 0 && nchunks > 1;
  const int64_t covered = (int64_t)(T - 1) * hop + n_fft;                // padded positions that receive any frame
  const int64_t stride = (int64_t)fast::kObjRows * blockDim.x, first = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  // the seam sample n may lie in: the first n_fft - hop padded positions of the tile that holds frame (n + pad) / hop (at most one
  // seam: tiles are longer than a frame); returns the index into xtail or -1
  const bool small = covered < ((int64_t)1 << 31) && (int64_t)T * nchunks < ((int64_t)1 << 31);   // 32-bit divisions (a 64-bit one is ~200 instructions)
  auto seam_of = [&](int64_t b, int64_t n) -> int64_t {
    if (!seams) return -1;
    int64_t f0 = small ? (int64_t)((unsigned)(n + pad) / (unsigned)hop) : (n + pad) / hop;
    if (f0 > T - 1) f0 = T - 1;
    int c = small ? (int)(((unsigned)(f0 + 1) * (unsigned)nchunks - 1u) / (unsigned)T) : (int)(((f0 + 1) * nchunks - 1) / T);   // largest c with c * T / nchunks <= f0, up to rounding: corrected below
    while (c + 1 < nchunks && hop_chunk_begin_(c + 1, T, nchunks) <= f0) ++c;
    while (c > 0 && hop_chunk_begin_(c, T, nchunks) > f0) --c;
    if (c < 1) return -1;
    const int64_t j0 = n + pad - (int64_t)hop_chunk_begin_(c, T, nchunks) * hop;
    return (j0 >= 0 && j0 < keep) ? (b * nchunks + (c - 1)) * keep + j0 : -1;
  };
  if (st_on) {
    // ---- the whole gradient: seams finished on the way, every sample outside the folded stretches counted
    if (vec_ok) {
      // 16-byte pieces (keep, hop, pad, len multiples of 4), tile by tile: the samples between the first frame of tile c and the
      // first frame of tile c + 1 (the first tile from 0, the last to the end), whose first n_fft - hop are the seam with tile c - 1
      // - no division per piece; a piece lies in a seam whole or not at all
      // Two groups of 512 threads walk tiles of their own, three pieces per thread and trip with their loads all requested first:
      // 9 - 12 loads of 16 bytes in flight per thread (four pieces: 46 registers spilled at 16 waves per CU).
      const int64_t n_pairs = rows * nchunks;
      constexpr int GT = SPECINV_EPI_GROUP, NG = kObjEpiThreads / GT;
      const int grp = threadIdx.x / GT, tig = threadIdx.x - grp * GT;
      // (b, c) of a group's tiles advance by additions: an integer division per tile - and the plain form had three, one of them
      // 64-bit - is a few hundred vector instructions that every wave of the group repeats (round 5: 15 of the pass's 29 us went
      // into index arithmetic)
      const unsigned pr0 = (unsigned)(NG * (int)blockIdx.x + grp), pstep = (unsigned)(NG * (int)gridDim.x);
      unsigned bq = pr0 / (unsigned)nchunks, cq = pr0 - bq * (unsigned)nchunks;
      const unsigned bstep = pstep / (unsigned)nchunks, cstep = pstep - bstep * (unsigned)nchunks;
      for (int64_t pr = pr0; pr < n_pairs; pr += pstep, bq += bstep, cq += cstep) {
        if (cq >= (unsigned)nchunks) {
          cq -= (unsigned)nchunks;
          ++bq;
        }
        const int64_t b = bq;
        const int c = (int)cq;
        const int64_t t_lo = (int64_t)hop_chunk_begin_(c, T, nchunks) * hop - pad;       // span[0] of tile c (may be < 0)
        const int64_t t_hi = (int64_t)hop_chunk_begin_(c + 1, T, nchunks) * hop - pad;
        const int64_t n_lo = c == 0 ? 0 : (t_lo < 0 ? 0 : (t_lo > len ? len : t_lo));
        const int64_t n_hi = c == nchunks - 1 ? len : (t_hi < 0 ? 0 : (t_hi > len ? len : t_hi));
        // (offsets inside a tile are ints: pointers to the tile's first sample, the fold stretches as offsets too)
        const int span = (int)(n_hi - n_lo), seam_end = (seams && c >= 1) ? (int)(t_lo + keep - n_lo) : 0;   // pieces below seam_end: + tail
        const float* tl = xtail + (b * nchunks + (c - 1)) * keep + (n_lo - t_lo);
        float* gb = grad + b * len + n_lo;
        const float* pb = st_gp ? st_gp + b * len + n_lo : nullptr;
        const float* db = st_d ? st_d + b * len + n_lo : nullptr;
        // offsets o with n_lo + o <= pad or n_lo + o >= len - 1 - pad lie in a folded stretch
        const int64_t fl = fold ? pad - n_lo : -1, fh = fold ? len - 1 - pad - n_lo : (int64_t)1 << 40;
        const int f_lo = fl < -1 ? -1 : (fl > span ? span : (int)fl), f_hi = fh > span ? span : (fh < 0 ? 0 : (int)fh);
        constexpr int U = 3;
        for (int o0 = 4 * tig; o0 < span; o0 += U * 4 * GT) {
          int of[U];
          fast::v4f gv[U], pv[U], dv[U], tv[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            of[u] = o0 + u * 4 * GT < span ? o0 + u * 4 * GT : o0;
            gv[u] = (SPECINV_EPI_ABL & 16) ? fast::v4f{1.0f, 2.0f, 3.0f, (float)of[u]} : *reinterpret_cast<const fast::v4f*>(gb + of[u]);
            pv[u] = (pb && !(SPECINV_EPI_ABL & 4)) ? *reinterpret_cast<const fast::v4f*>(pb + of[u]) : gv[u];
            dv[u] = db ? *reinterpret_cast<const fast::v4f*>(db + of[u]) : gv[u];
            if (d_impl) {
#pragma unroll
              for (int e = 0; e < 4; ++e) dv[u][e] = d_of(pv[u][e]);
            }
            tv[u] = (of[u] < seam_end && !(SPECINV_EPI_ABL & 8)) ? *reinterpret_cast<const fast::v4f*>(tl + of[u]) : fast::v4f{0.0f, 0.0f, 0.0f, 0.0f};
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            if (u > 0 && o0 + u * 4 * GT >= span) continue;
            const int o = of[u];
            const bool sm = o < seam_end && !(SPECINV_EPI_ABL & 8);
            if (o > f_lo && o + 3 < f_hi) {
              if (sm) {
                gv[u] = gv[u] + tv[u];
                *reinterpret_cast<fast::v4f*>(gb + o) = gv[u];
              }
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                if (SPECINV_EPI_ABL & 2) sta.mg = fmaxf(sta.mg, gv[u][e] + pv[u][e] + dv[u][e]);
                else sta.add(gv[u][e], pb ? pv[u][e] : gv[u][e], (db || d_impl) ? dv[u][e] : gv[u][e], st_t);
              }
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                if (o + e <= f_lo || o + e >= f_hi) continue;             // finished - and counted - by the margin thread of this sample
                const float g1 = sm ? gv[u][e] + tv[u][e] : gv[u][e];
                if (sm) gb[o + e] = g1;
                sta.add(g1, pb ? pv[u][e] : g1, (db || d_impl) ? dv[u][e] : g1, st_t);
              }
            }
          }
        }
      }
    } else {
      const int64_t items = rows * len;
      for (int64_t i = first; i < items; i += stride) {
        const int64_t b = i / len, n = i - b * len;
        if (fold && (n <= pad || n >= len - 1 - pad)) continue;
        float g1 = grad[i];
        const int64_t sx = seam_of(b, n);
        if (sx >= 0) {
          g1 += xtail[sx];
          grad[i] = g1;
        }
        sta.add(g1, st_gp ? st_gp[i] : g1, d_impl ? d_of(st_gp[i]) : st_d ? st_d[i] : g1, st_t);
      }
    }
  } else if (vec_ok) {
    // ---- seams only
    const int n4 = keep / 4;
    const int64_t items = n_tail / 4;
    for (int64_t i = first; i < items; i += stride) {
      const int q = (int)(i / n4), j = 4 * (int)(i - (int64_t)q * n4);
      const int b = q / (nchunks - 1), c = q - b * (nchunks - 1) + 1;
      const int64_t n = (int64_t)hop_chunk_begin_(c, T, nchunks) * hop + j - pad;
      if (n < 0 || n >= len) continue;
      const fast::v4f tv = *reinterpret_cast<const fast::v4f*>(xtail + ((int64_t)b * nchunks + (c - 1)) * keep + j);
      float* gq = grad + (int64_t)b * len + n;
      const fast::v4f gv = *reinterpret_cast<const fast::v4f*>(gq);
      if (!(fold && (n <= pad || n + 3 >= len - 1 - pad))) {
        *reinterpret_cast<fast::v4f*>(gq) = gv + tv;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int64_t ne = n + e;
          if (ne <= pad || ne >= len - 1 - pad) continue;                 // finished by the margin thread of this sample
          gq[e] = gv[e] + tv[e];
        }
      }
    }
  } else {
    for (int64_t i = first; i < n_tail; i += stride) {                   // (b, c - 1, j)
      const int j = (int)(i % keep);
      const int c = (int)((i / keep) % (nchunks - 1)) + 1;
      const int64_t b = i / ((int64_t)keep * (nchunks - 1));
      const int64_t n = (int64_t)hop_chunk_begin_(c, T, nchunks) * hop + j - pad;
      if (n < 0 || n >= len) continue;
      if (fold && (n <= pad || n >= len - 1 - pad)) continue;             // finished by the margin thread of this sample
      grad[b * len + n] += xtail[(b * nchunks + (c - 1)) * keep + j];
    }
  }
  // ---- margins
  for (int64_t i = first; i < ((SPECINV_EPI_ABL & 1) ? 0 : n_margin); i += stride) {
    const int64_t per_row = 2 * ((int64_t)pad + 1);
    const int64_t bi = n_margin < ((int64_t)1 << 31) ? (int64_t)((unsigned)i / (unsigned)per_row) : i / per_row, j = i - bi * per_row;
    int64_t n;
    if (j <= pad) {
      n = j;                                            // left stretch 0 .. pad
      if (n >= len) continue;
    } else {
      n = len - 1 - pad + (j - pad - 1);                // right stretch len-1-pad .. len-1
      if (n <= pad || n >= len) continue;               // (short signals: already covered by the left stretch)
    }
    const float* mg = margins + bi * 2 * pad;
    auto at = [&](int64_t np) -> float {                // gradient w.r.t. padded sample np
      if (np < 0 || np >= covered) return 0.0f;
      if (np < pad) return mg[np];
      const int64_t r = np - pad - len;
      return (r >= 0 && r < pad) ? mg[pad + r] : 0.0f;
    };
    float g = 0;
    switch (pad_mode) {
      case SPECINV_PAD_REFLECT:
        if (n >= 1 && n <= pad) g += at(pad - n);
        if (n <= len - 2 && n >= len - 1 - pad) g += at(pad + len + (len - 2 - n));
        break;
      case SPECINV_PAD_REPLICATE:
        if (n == 0)
          for (int64_t q = 0; q < pad; ++q) g += at(q);
        if (n == len - 1)
          for (int64_t q = 0; q < pad; ++q) g += at(pad + len + q);
        break;
      case SPECINV_PAD_CIRCULAR:
        if (n >= len - pad) g += at(n - (len - pad));
        if (n < pad) g += at(pad + len + n);
        break;
      default:
        break;
    }
    float v = grad[bi * len + n];
    const int64_t sx = seam_of(bi, n);
    if (sx >= 0) v += xtail[sx];
    const float g1 = v + g;
    grad[bi * len + n] = g1;
    if (st_on) sta.add(g1, st_gp ? st_gp[bi * len + n] : g1, d_impl ? d_of(st_gp[bi * len + n]) : st_d ? st_d[bi * len + n] : g1, st_t);
  }
  if (!st_on) return;
  // ---- this block's row: its own figures + the squared-error sums of its share of the objective kernel's tiles
  __shared__ double red9[kObjEpiThreads / 64][9];
  double v[9] = {sta.s[0], sta.s[1], sta.s[2], sta.s[3], sta.s[4], sta.s[5], 0.0, (double)sta.mg, (double)sta.md};
  const int lo = (int)((int64_t)n_part * blockIdx.x / fast::kObjRows), hi = (int)((int64_t)n_part * (blockIdx.x + 1) / fast::kObjRows);
  for (int tl = lo + threadIdx.x; tl < hi; tl += blockDim.x) v[6] += part[tl];
  if (!(SPECINV_EPI_ABL & 32)) block_reduce9(v, red9);
  const bool tail = dec.ticket != nullptr;
  __shared__ int last;
  if (threadIdx.x == 0) {                       // component-major: readers take one component of every row with one coalesced load
    double* row = st.rows + blockIdx.x;
    auto w = [&](int c) { return c < 6 ? v[c] : c == 6 ? v[7] : c == 7 ? v[8] : v[6]; };
    if (!tail) {
#pragma unroll
      for (int c = 0; c < 9; ++c) row[c * fast::kObjRows] = w(c);
    } else {
      // ---- the optimiser's decisions ride along, taken by the workgroup that finishes last.  Two forms of the hand-over:
      // (a) dec.fence (the default since round 6): the HIP memory model's own - the ticket is an acquire-release
      // read-modify-write at agent scope drawn by the ONE thread that wrote the row, the last workgroup's row loads follow an
      // acquire fence.  Measured at C5 against (b): 112.4 / 114.0 / 112.8 against 111.7 / 113.7 / 111.9 ms per step (+0.5 %;
      // round 5 had measured a release FENCE executed by every thread of every workgroup: +11 us per launch).
      // (b) SPECINV_LBFGS_TAIL_FENCE=0, gfx942 / gfx950 only - round 5's ISA-level protocol: the row goes out with agent-scope
      // atomic stores (`sc1`: written THROUGH this XCD's L2), `s_waitcnt vmcnt(0)` retires them (a write-through store is
      // acknowledged by the memory side: the wait the release sequence `buffer_wbl2 sc1; s_waitcnt vmcnt(0)` itself relies on)
      // before a RELAXED ticket is drawn; the reader's row loads are agent-scope atomic loads issued after its ticket has
      // returned.  Not a synchronises-with edge of the language's model; kept for the A/B and pinned against (a) bit for bit at
      // full size (tests/test_gpu_bench_sizes.py::test_c5_optimiser_step_full_size).
#pragma unroll
      for (int c = 0; c < 9; ++c) __hip_atomic_store(row + c * fast::kObjRows, w(c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (kTailSc1Protocol && !dec.fence) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        last = __hip_atomic_fetch_add(dec.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1 : 0;
      } else {
        last = __hip_atomic_fetch_add(dec.ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1 : 0;
      }
    }
  }
  if (!tail) return;
  __syncthreads();
  if (!last) return;
  if (!kTailSc1Protocol || dec.fence) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  if (threadIdx.x == 0) __hip_atomic_store(dec.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  lbd_tail_decide(dec, scale, red9, lbd_r);
}

// the second level of the tree: out = {loss, g.d, sum|g|, max|g|, max|d|, y.s, y.y, g.g, g.g_prev} from the epilogue's rows (one
// workgroup, fixed order; the layout lbfgs.py:_batch reads from its board)
static __global__ __launch_bounds__(256) void k_objective_finish_rows(const double* __restrict__ rows, double scale, double* __restrict__ out,
                                                                      int n_out) {
  __shared__ double red9[4][9];
  double v[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int r = threadIdx.x; r < fast::kObjRows; r += blockDim.x) {
    const double* q = rows + r;
#pragma unroll
    for (int c = 0; c < 6; ++c) v[c] += q[c * fast::kObjRows];
    v[6] += q[8 * fast::kObjRows];
    v[7] = fmax(v[7], q[6 * fast::kObjRows]);
    v[8] = fmax(v[8], q[7 * fast::kObjRows]);
  }
  block_reduce9(v, red9);
  if (threadIdx.x == 0) {
    out[0] = scale * v[6];
    out[1] = v[0];
    out[2] = v[1];
    out[3] = v[7];
    out[4] = v[8];
    if (n_out > 5) {
      out[5] = v[2];
      out[6] = v[3];
      out[7] = v[4];
      out[8] = v[5];
    }
  }
}

// loss = sum(partials) / numel, to the host (synchronises) or to a device scalar (nothing waits)
template <typename P>
int tf_finish_loss(P& pl, int64_t n_part, double numel, double* loss_host, double* loss_dev) {
  hipLaunchKernelGGL(k_finish_scaled, dim3(1), dim3(256), 0, pl.stream, pl.partials.template as<double>(), (int)n_part, 1.0 / numel,
                     loss_dev ? loss_dev : pl.sums.template as<double>());
  SI_HIP(hipGetLastError());
  if (loss_dev) return SPECINV_OK;
  SI_HIP(hipMemcpyAsync(loss_host, pl.sums.p, sizeof(double), hipMemcpyDeviceToHost, pl.stream));
  SI_HIP(si_stream_wait_short(pl.stream));
  return SPECINV_OK;
}

// will the frame walk (kernels_objective_walk.h) serve this plan's objective?  hop = n_fft / {2, 4, 8}, centred, a signal of whole
// hops, a filterbank with at most two rows per bin (the device-resident optimiser defers its steps into the walk only then)
template <typename P>
bool tf_walk_serves(P& pl, int64_t len) {
  if (pl.force_generic || !pl.cfg.onesided || !pl.fast.xform_ok || (pl.fast.xform_R != 8 && pl.fast.xform_R != 16)) return false;
  if (pl.tf_kind != SPECINV_TF_LOGMEL || !pl.tf_sp_ok) return false;
  for (const char* name : {"SPECINV_OBJ_SPARSE", "SPECINV_OBJ_WALK"})
    if (const char* e = getenv(name)) {
      if (e[0] == '0') return false;
    }
  if (const char* e = getenv("SPECINV_DISABLE_FUSED_OBJECTIVE")) {
    if (e[0] == '1') return false;
  }
  const int N = pl.N(), hop = pl.cfg.hop_length, T = pl.Tn(), pad = pl.pad, R = pl.fast.xform_R;
  if (hop > N || hop < 2 || pad >= len) return false;
  const int ov = N % hop == 0 ? N / hop : 0;
  if (!(pl.tf_walk_ok && (ov == 2 || ov == 4 || ov == 8) && 2 * pad == N && len == (int64_t)(T - 1) * hop && T >= (ov == 8 ? 16 : 8))) return false;
  const size_t lds = R == 16 ? fast::obj_walk_lds_bytes<16>(pl.tf_walk.total) : fast::obj_walk_lds_bytes<8>(pl.tf_walk.total);
  return lds <= 160 * 1024 - 512;
}

// loss and gradient of the log-mel objective in one launch (+ the seam / padding passes of the unfused path).
// `*used` stays false when the configuration is not covered: the caller then runs the kernel chain.
template <typename P>
int tf_loss_grad_fused(P& pl, const float* x, int64_t len, const float* target, double* loss, float* grad, bool* used,
                       double* loss_dev = nullptr, const fast::ObjCtl* ctl = nullptr, const fast::ObjStatReq* st = nullptr,
                       const fast::ObjDecide* dec = nullptr) {
  *used = false;
  const bool mag = pl.tf_kind == SPECINV_TF_MAG;
  if (pl.force_generic || !pl.cfg.onesided || !pl.fast.xform_ok || (pl.fast.xform_R != 8 && pl.fast.xform_R != 16)) return SPECINV_OK;
  bool sparse = !mag && pl.tf_sp_ok;     // the filterbank in band form: contractions on the vector units (SPECINV_OBJ_SPARSE=0: matrix cores)
  if (const char* e = getenv("SPECINV_OBJ_SPARSE")) {
    if (e[0] == '0') sparse = false;
  }
  if (!mag && (pl.tf_kind != SPECINV_TF_LOGMEL || (pl.tf_obj_mt == 0 && !sparse))) return SPECINV_OK;
  if (const char* e = getenv("SPECINV_DISABLE_FUSED_OBJECTIVE")) {
    if (e[0] == '1') return SPECINV_OK;
  }
  const int N = pl.N(), hop = pl.cfg.hop_length, T = pl.Tn(), B = pl.B(), pad = pl.pad, R = pl.fast.xform_R;
  const int MT = mag ? 3 : sparse ? 9 : pl.tf_obj_mt;
  if (hop > N || hop < 2 || pad >= len) return SPECINV_OK;
  // ---- the frame walk (kernels_objective_walk.h): hop = n_fft / 4, centred, a filterbank with at most two rows per bin
  {
    const int ov = hop > 0 && N % hop == 0 ? N / hop : 0;
    const bool walk = tf_walk_serves(pl, len);              // (SPECINV_OBJ_WALK=0 / SPECINV_OBJ_SPARSE=0: the tile kernel)
    SI_CHECK(walk || !ctl || !ctl->x_sel, SPECINV_ESTATE, "a deferred step without the frame walk");
    if (walk) {
      // chunks: one round of two waves per SIMD where the frames allow it (>= 8 frames per wave), an even count so that the two
      // waves of a SIMD can take a long and a short chunk (the older wave runs faster: kernels_fast_td.h)
      // (hop = n_fft / 8: seven of a chunk's hop-blocks are its seam with the chunk before - chunks of >= 16 frames there)
      const int floor_ch = ov == 8 ? 16 : 8;
      int nch = (int)std::max<int64_t>(1, std::min<int64_t>(T / floor_ch, 2048 / std::max(1, B)));
      if (nch > 1 && (nch & 1)) --nch;
      if (const char* e = getenv("SPECINV_OBJ_WALK_CHUNKS")) nch = std::max(1, std::min(T / floor_ch, atoi(e)));
      const int len_ch = T / nch;
      int skew = 0;
      // (C5, chunks of 8 frames, one box: skew 0 / 1 / 2 / 3 -> 126.1 / 122.0 / 124.9 / 128.0 ms per step: an eighth of the chunk -
      // the contractions' LDS waits leave the younger wave more of the SIMD than the Griffin-Lim kernel's pure transforms do)
      if ((nch & 1) == 0 && (int64_t)B * nch > 1024 && len_ch >= 8) skew = std::max(1, len_ch / 8);
      if (const char* e = getenv("SPECINV_OBJ_WALK_SKEW")) skew = (nch & 1) == 0 ? std::max(0, atoi(e)) : 0;
      if (len_ch - skew < std::max(ov, 6)) skew = 0;            // (the shorter chunk of a pair still holds its seam and a frame more)
      const int keep = N - hop;
      const int64_t n_waves = (int64_t)B * nch;
      SI_TRY(pl.fast.hop_inv_tail.reserve((size_t)n_waves * keep * sizeof(float) + 16));
      SI_TRY(pl.fast.hop_inv_margins.reserve((size_t)B * 2 * pad * sizeof(float)));
      SI_TRY(pl.partials.reserve(std::max<size_t>((size_t)n_waves, 3 * 1024) * sizeof(double)));
      const double numel = (double)B * T * pl.tf_mels;
      fast::ObjWalkArgs a{};
      a.x = x;
      a.grad = grad;
      a.margins = pl.fast.hop_inv_margins.template as<float>();
      a.xtail = pl.fast.hop_inv_tail.template as<float>();
      a.target = target;
      a.blob = pl.tf_walk_blob.template as<fast::f32x4>();
      a.window = pl.window.template as<float>();
      a.partials = pl.partials.template as<double>();
      a.len = len;
      a.T = T;
      a.nchunks = nch;
      a.n_waves = (int)n_waves;
      a.skew = skew;
      a.pad_mode = pl.cfg.pad_mode;
      a.n_mels = pl.tf_mels;
      a.fwd_scale = pl.fc.fwd_scale;
      a.dscale = (float)(2.0 / numel);
      a.w = pl.tf_walk;
      if (ctl) {
        a.ctl_eval = ctl->do_eval;
        a.ctl_cur = ctl->cur;
        a.grad_alt = ctl->grad_alt;
        a.x_alt = ctl->x_alt;
        a.px_sel = ctl->x_sel;
        a.px_pending = ctl->x_pending;
        a.pt_pend = ctl->t_pend;
        a.pc0_pend = ctl->c0_pend;
      }
      const void* fn = R == 16 ? (ov == 2 ? (const void*)fast::k_objective_walk<16, 2> : ov == 4 ? (const void*)fast::k_objective_walk<16, 4>
                                                                                                 : (const void*)fast::k_objective_walk<16, 8>)
                               : (ov == 2 ? (const void*)fast::k_objective_walk<8, 2> : ov == 4 ? (const void*)fast::k_objective_walk<8, 4>
                                                                                                : (const void*)fast::k_objective_walk<8, 8>);
      const size_t lds = R == 16 ? fast::obj_walk_lds_bytes<16>(pl.tf_walk.total) : fast::obj_walk_lds_bytes<8>(pl.tf_walk.total);
      if (lds <= 160 * 1024 - 512) {
        SI_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        void* kargs[] = {&a};
        SI_HIP(hipLaunchKernel(fn, dim3((unsigned)ceil_div(n_waves, fast::kWalkWaves)), dim3(64 * fast::kWalkWaves), kargs, lds, pl.stream));
        *used = true;
        pl.objective_kind = 3;
        const bool fold = pl.cfg.pad_mode != SPECINV_PAD_CONSTANT;
        const int64_t n_tail = nch > 1 ? (int64_t)B * (nch - 1) * keep : 0;
        const int64_t n_margin = fold ? (int64_t)B * 2 * (pad + 1) : 0;
        const bool with_rows = st && st->rows;
        const int blocks = fast::kObjRows + (with_rows ? 0 : 1);
        double* slot = loss_dev ? loss_dev : pl.sums.template as<double>();
        const int vec_ok = (len & 3) == 0 && ((uintptr_t)grad & 15) == 0 && (!ctl || ((uintptr_t)ctl->grad_alt & 15) == 0) &&
                           (!with_rows || ((((uintptr_t)st->d | (uintptr_t)st->gp) & 15) == 0));
        hipLaunchKernelGGL(k_objective_epilogue, dim3((unsigned)blocks), dim3(kObjEpiThreads), 0, pl.stream, grad, (const float*)a.xtail,
                           (const float*)a.margins, (const double*)pl.partials.template as<double>(), slot, T, nch, N, hop, keep,
                           pad, pl.cfg.pad_mode, (int64_t)len, (int64_t)B, n_tail, n_margin, (int)n_waves, 1.0 / numel,
                           ctl ? *ctl : fast::ObjCtl{}, with_rows ? *st : fast::ObjStatReq{}, vec_ok, skew,
                           with_rows && dec ? *dec : fast::ObjDecide{});
        SI_HIP(hipGetLastError());
        if (with_rows || loss_dev) return SPECINV_OK;
        SI_HIP(hipMemcpyAsync(loss, pl.sums.p, sizeof(double), hipMemcpyDeviceToHost, pl.stream));
        SI_HIP(si_stream_wait_short(pl.stream));
        return SPECINV_OK;
      }
    }
  }
  const int nch = (T + fast::kObjTile - 1) / fast::kObjTile;
  if (nch > 1 && T / nch < (N - 1) / hop + 1) return SPECINV_OK;                // a seam must not reach a tile's own tail
  const int keep = N - hop;
  const int64_t n_tiles = (int64_t)B * nch;
  SI_TRY(pl.fast.hop_inv_tail.reserve((size_t)n_tiles * std::max(1, keep) * sizeof(float) + 16));
  SI_TRY(pl.fast.hop_inv_margins.reserve((size_t)B * 2 * std::max(1, pad) * sizeof(float)));
  SI_TRY(pl.partials.reserve(std::max<size_t>((size_t)n_tiles, 3 * 1024) * sizeof(double)));
  const double numel = (double)B * T * (mag ? pl.n_freq : pl.tf_mels);
  fast::ObjArgs a{};
  a.x = x;
  a.grad = grad;
  a.margins = pl.fast.hop_inv_margins.template as<float>();
  a.xtail = pl.fast.hop_inv_tail.template as<float>();
  a.target = target;
  a.melA = pl.tf_mel_a.template as<fast::f32x4>();
  a.melB = pl.tf_mel_b.template as<fast::f32x4>();
  a.tab = pl.tf_obj_tab.template as<int>();
  if (sparse) {
    a.melA = pl.tf_sp_blob.template as<fast::f32x4>();
    a.melB = nullptr;
    a.tab = pl.tf_sp_tab.template as<int>();
    a.sp_rm = pl.tf_sp.rm;
    a.sp_cm = pl.tf_sp.cm;
    a.sp_cw = pl.tf_sp.cw;
    a.sp_total = pl.tf_sp.total;
    a.sp_cmax = pl.tf_sp.cmax;
    a.sp_rows = pl.tf_sp.rows;
  }
  a.window = pl.window.template as<float>();
  a.partials = pl.partials.template as<double>();
  a.len = len;
  a.T = T;
  a.nchunks = nch;
  a.hop = hop;
  a.pad = pad;
  a.pad_mode = pl.cfg.pad_mode;
  a.n_mels = pl.tf_mels;
  a.fwd_scale = pl.fc.fwd_scale;
  a.dscale = (float)(2.0 / numel);
  a.hop_magic = (unsigned)(((1ull << 32) + hop - 1) / hop);
  if (ctl) {
    a.ctl_eval = ctl->do_eval;
    a.ctl_cur = ctl->cur;
    a.grad_alt = ctl->grad_alt;
  }
  const void* fn = nullptr;
  size_t lds = 0;
#define SPECINV_OBJ_CASE(RR, MM)                                   \
  if (R == RR && MT == MM) {                                       \
    fn = (const void*)fast::k_objective_logmel<RR, MM>;            \
    lds = fast::ObjGeo<RR, MM>::lds_bytes();                       \
  }
  if (mag) {
    if (R == 16) fn = (const void*)fast::k_objective_logmel<16, 3, true>;
    else fn = (const void*)fast::k_objective_logmel<8, 3, true>;
    lds = R == 16 ? fast::ObjGeo<16, 3>::lds_bytes() : fast::ObjGeo<8, 3>::lds_bytes();
  } else if (sparse) {
    if (R == 16) fn = (const void*)fast::k_objective_logmel<16, 9, false, true>;
    else fn = (const void*)fast::k_objective_logmel<8, 9, false, true>;
    lds = R == 16 ? fast::ObjGeo<16, 9>::lds_bytes() : fast::ObjGeo<8, 9>::lds_bytes();
  } else {
    SPECINV_OBJ_CASE(16, 3) SPECINV_OBJ_CASE(16, 4) SPECINV_OBJ_CASE(16, 5) SPECINV_OBJ_CASE(16, 8)
    SPECINV_OBJ_CASE(8, 3) SPECINV_OBJ_CASE(8, 4) SPECINV_OBJ_CASE(8, 5) SPECINV_OBJ_CASE(8, 9)
  }
#undef SPECINV_OBJ_CASE
  if (fn == nullptr || lds > 160 * 1024) return SPECINV_OK;
  SI_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
#if SPECINV_OBJ_STAMPS
  static unsigned long long* d_stamps = nullptr;
  if (!d_stamps) SI_HIP(hipMalloc(&d_stamps, (size_t)n_tiles * 16 * sizeof(unsigned long long)));
  a.stamps = d_stamps;
#endif
  void* kargs[] = {&a};
  SI_HIP(hipLaunchKernel(fn, dim3((unsigned)n_tiles), dim3(64 * fast::kObjWaves), kargs, lds, pl.stream));
#if SPECINV_OBJ_STAMPS
  {
    std::vector<unsigned long long> h((size_t)n_tiles * 16);
    SI_HIP(hipMemcpy(h.data(), d_stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    static int calls = 0;
    if (++calls == 5) {
      // (band form: "forward GEMM" = staging the bands, its "wait" = the band forward with step 3, "reduce" = the barrier behind it)
      const char* names[12] = {"tables + twiddle regs", "analysis (2 frames)", "  wait", "forward GEMM", "  wait", "reduce+log1p+dM",
                               "backward GEMM", "  wait", "synthesis (2 frames)", "  wait", "frames to LDS", "overlap-add"};
      double tot[13] = {0};
      for (int64_t t = 0; t < n_tiles; ++t)
        for (int i = 1; i <= 12; ++i) tot[i] += (double)(h[t * 16 + i] - h[t * 16 + i - 1]);
      fprintf(stderr, "k_objective_logmel phase cycles (s_memtime; mean over %lld tiles, wave %d of the workgroup; 'wait' = at the barrier):\n", (long long)n_tiles, SPECINV_OBJ_STAMP_WAVE);
      for (int i = 1; i <= 11; ++i) fprintf(stderr, "  %-22s %9.0f\n", names[i - 1], tot[i] / n_tiles);
      fprintf(stderr, "  %-22s %9.0f\n", "write-out", tot[12] / n_tiles);
      double all = 0;
      for (int64_t t = 0; t < n_tiles; ++t) all += (double)(h[t * 16 + 12] - h[t * 16]);
      fprintf(stderr, "  %-22s %9.0f\n", "whole tile", all / n_tiles);
    }
  }
#endif
  *used = true;
  pl.objective_kind = sparse ? 2 : 1;
  {
    const bool fold = pad > 0 && pl.cfg.pad_mode != SPECINV_PAD_CONSTANT;
    const int64_t n_tail = (nch > 1 && keep > 0) ? (int64_t)B * (nch - 1) * keep : 0;
    const int64_t n_margin = fold ? (int64_t)B * 2 * (pad + 1) : 0;
    const bool with_rows = st && st->rows;
    const int blocks = fast::kObjRows + (with_rows ? 0 : 1);
    double* slot = loss_dev ? loss_dev : pl.sums.template as<double>();
    const int vec_ok = ((keep | hop | pad) & 3) == 0 && (len & 3) == 0 && ((uintptr_t)grad & 15) == 0 &&
                       (!ctl || ((uintptr_t)ctl->grad_alt & 15) == 0) && n_tail / 4 / std::max(1, keep / 4) < (1ll << 31) &&
                       (!with_rows || ((((uintptr_t)st->d | (uintptr_t)st->gp) & 15) == 0));
    hipLaunchKernelGGL(k_objective_epilogue, dim3((unsigned)blocks), dim3(kObjEpiThreads), 0, pl.stream, grad, (const float*)a.xtail,
                       (const float*)a.margins, (const double*)pl.partials.template as<double>(), slot, T, nch, N, hop, keep,
                       pad, pl.cfg.pad_mode, (int64_t)len, (int64_t)B, n_tail, n_margin, (int)n_tiles, 1.0 / numel,
                       ctl ? *ctl : fast::ObjCtl{}, with_rows ? *st : fast::ObjStatReq{}, vec_ok, 0, fast::ObjDecide{});
    SI_HIP(hipGetLastError());
    if (with_rows) return SPECINV_OK;      // (loss and figures are in the rows: the caller finishes them)
  }
  if (loss_dev) return SPECINV_OK;
  SI_HIP(hipMemcpyAsync(loss, pl.sums.p, sizeof(double), hipMemcpyDeviceToHost, pl.stream));
  SI_HIP(si_stream_wait_short(pl.stream));
  return SPECINV_OK;
}

template <typename P, typename T>
int tf_setup(P& pl, int kind, const T* mel_fb, int n_mels) {
  SI_CHECK(kind == SPECINV_TF_MAG || kind == SPECINV_TF_LOGMEL, SPECINV_EINVAL, "unknown transform kind %d", kind);
  if (kind == SPECINV_TF_LOGMEL) {
    SI_CHECK(mel_fb && n_mels > 0, SPECINV_EINVAL, "log-mel transform needs a filterbank");
    SI_TRY(pl.tf_mel.reserve((size_t)n_mels * pl.n_freq * sizeof(T)));
    SI_HIP(hipMemcpyAsync(pl.tf_mel.p, mel_fb, (size_t)n_mels * pl.n_freq * sizeof(T), hipMemcpyDeviceToDevice, pl.stream));
    pl.tf_mels = n_mels;
    if constexpr (std::is_same<T, float>::value) {
      const int mt = (n_mels + 31) / 32;
      if (mt <= 4) {
        const int ksteps = (pl.n_freq + 31) / 32;
        const int64_t total = (int64_t)ksteps * 32 * mt * 32;
        SI_TRY(pl.tf_mel_tiled.reserve((size_t)total * sizeof(float)));
        hipLaunchKernelGGL(k_mel_tile, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, pl.stream,
                           pl.tf_mel.template as<float>(), pl.tf_mel_tiled.template as<float>(), pl.n_freq, n_mels, mt * 32,
                           total);
        SI_HIP(hipGetLastError());
        SI_TRY(pl.tf_mel_tiled_t.reserve((size_t)total * sizeof(float)));
        hipLaunchKernelGGL(k_mel_tile_t, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, pl.stream,
                           pl.tf_mel.template as<float>(), pl.tf_mel_tiled_t.template as<float>(), pl.n_freq, n_mels, mt * 32,
                           total);
        SI_HIP(hipGetLastError());
      }
      // operand tiles of the one-launch objective (kernels_objective.h): one-sided spectra on the wave-level FFT
      pl.tf_obj_mt = 0;
      const int mt16 = obj_mel_tiles(n_mels, pl.fast.xform_R);
      if (mt16 > 0 && pl.cfg.onesided && pl.fast.xform_ok && (pl.fast.xform_R == 8 || pl.fast.xform_R == 16)) {
        std::vector<float> h_mel((size_t)n_mels * pl.n_freq), hA, hB;
        std::vector<int> h_tab;
        SI_HIP(hipMemcpyAsync(h_mel.data(), pl.tf_mel.p, h_mel.size() * sizeof(float), hipMemcpyDeviceToHost, pl.stream));
        SI_HIP(hipStreamSynchronize(pl.stream));
        fast::obj_build_blocks(h_mel.data(), pl.n_freq, n_mels, mt16, hA, hB, h_tab);
        SI_TRY(pl.tf_mel_a.reserve(hA.size() * sizeof(float)));
        SI_TRY(pl.tf_mel_b.reserve(hB.size() * sizeof(float)));
        SI_TRY(pl.tf_obj_tab.reserve(h_tab.size() * sizeof(int)));
        SI_HIP(hipMemcpy(pl.tf_mel_a.p, hA.data(), hA.size() * sizeof(float), hipMemcpyHostToDevice));
        SI_HIP(hipMemcpy(pl.tf_mel_b.p, hB.data(), hB.size() * sizeof(float), hipMemcpyHostToDevice));
        SI_HIP(hipMemcpy(pl.tf_obj_tab.p, h_tab.data(), h_tab.size() * sizeof(int), hipMemcpyHostToDevice));
        pl.tf_obj_mt = mt16;
      }
      // ... and its band form, when the filterbank is sparse enough (a mel filterbank is): k_objective_logmel<R, 9, false, true>
      pl.tf_sp_ok = false;
      pl.tf_walk_ok = false;
      if (n_mels <= 16 * 9 && pl.cfg.onesided && pl.fast.xform_ok && (pl.fast.xform_R == 8 || pl.fast.xform_R == 16)) {
        std::vector<float> h_mel((size_t)n_mels * pl.n_freq), blob;
        std::vector<int> h_tab;
        SI_HIP(hipMemcpyAsync(h_mel.data(), pl.tf_mel.p, h_mel.size() * sizeof(float), hipMemcpyDeviceToHost, pl.stream));
        SI_HIP(hipStreamSynchronize(pl.stream));
        const int uni = pl.fast.xform_R == 16 ? fast::ObjGeo<16, 9>::UNI : fast::ObjGeo<8, 9>::UNI;
        fast::ObjSparseInfo inf;
        if (fast::obj_build_sparse(h_mel.data(), pl.n_freq, n_mels, uni, blob, h_tab, inf)) {
          SI_TRY(pl.tf_sp_blob.reserve(blob.size() * sizeof(float)));
          SI_TRY(pl.tf_sp_tab.reserve(h_tab.size() * sizeof(int)));
          SI_HIP(hipMemcpy(pl.tf_sp_blob.p, blob.data(), blob.size() * sizeof(float), hipMemcpyHostToDevice));
          SI_HIP(hipMemcpy(pl.tf_sp_tab.p, h_tab.data(), h_tab.size() * sizeof(int), hipMemcpyHostToDevice));
          pl.tf_sp = inf;
          pl.tf_sp_ok = true;
        }
        // ... and the tables of the frame walk (kernels_objective_walk.h)
        pl.tf_walk_ok = false;
        std::vector<float> wblob;
        fast::ObjWalkInfo winf;
        if (fast::obj_build_walk(h_mel.data(), pl.n_freq, n_mels, pl.fast.xform_R, wblob, winf)) {
          SI_TRY(pl.tf_walk_blob.reserve(wblob.size() * sizeof(float)));
          SI_HIP(hipMemcpy(pl.tf_walk_blob.p, wblob.data(), wblob.size() * sizeof(float), hipMemcpyHostToDevice));
          pl.tf_walk = winf;
          pl.tf_walk_ok = true;
        }
      }
    }
  } else {
    pl.tf_mels = 0;
    pl.tf_obj_mt = 0;
    pl.tf_sp_ok = false;
    pl.tf_walk_ok = false;
  }
  pl.tf_kind = kind;
  return SPECINV_OK;
}

template <typename P, typename T>
int tf_mel_forward(P& pl) {
  const int64_t BT = (int64_t)pl.B() * pl.Tn();
  SI_TRY(pl.tf_v.reserve((size_t)BT * pl.tf_mels * sizeof(T)));
  if constexpr (std::is_same<T, float>::value) {
    const int mt = (pl.tf_mels + 31) / 32;
    if (mt <= 4) {
      const void* fn = mt == 1 ? (const void*)k_mel_forward_splitk<1> : mt == 2 ? (const void*)k_mel_forward_splitk<2>
                       : mt == 3 ? (const void*)k_mel_forward_splitk<3> : (const void*)k_mel_forward_splitk<4>;
      const size_t lds = (size_t)4 * (32 * mt * 32 + 32 * 33) * sizeof(float);
      SI_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      const cplx<float>* sp = pl.tf_spec.template as<cplx<float>>();
      const float* ml = pl.tf_mel_tiled.template as<float>();
      float* out = pl.tf_v.template as<float>();
      int64_t bt = BT;
      int F = pl.n_freq, nm = pl.tf_mels;
      void* kargs[] = {&sp, &ml, &out, &bt, &F, &nm};
      SI_HIP(hipLaunchKernel(fn, dim3((unsigned)ceil_div(BT, 32)), dim3(256), kargs, lds, pl.stream));
    } else {
      hipLaunchKernelGGL(k_mel_forward_mfma, dim3((unsigned)ceil_div(BT, 32), (unsigned)ceil_div(pl.tf_mels, 32)), dim3(64), 0,
                         pl.stream, pl.tf_spec.template as<cplx<float>>(), pl.tf_mel.template as<float>(),
                         pl.tf_v.template as<float>(), BT, pl.n_freq, pl.tf_mels);
    }
  } else {
    hipLaunchKernelGGL((k_mel_forward_valu<T>), dim3((unsigned)ceil_div(BT * pl.tf_mels, 256)), dim3(256), 0, pl.stream,
                       pl.tf_spec.template as<cplx<T>>(), pl.tf_mel.template as<T>(), pl.tf_v.template as<T>(), BT, pl.n_freq,
                       pl.tf_mels);
  }
  SI_HIP(hipGetLastError());
  return SPECINV_OK;
}

template <typename P, typename T>
int tf_forward(P& pl, const T* x, int64_t len, T* v_out) {
  SI_CHECK(pl.tf_kind >= 0, SPECINV_ESTATE, "specinv_transform_setup has not been called");
  SI_CHECK(x && v_out, SPECINV_EINVAL, "null pointer");
  using C = cplx<T>;
  SI_TRY(pl.tf_spec.reserve(pl.nspec() * sizeof(C)));
  SI_TRY(pl.stft_internal(x, len, pl.tf_spec.template as<C>()));
  if (pl.tf_kind == SPECINV_TF_MAG) {
    const int64_t n = pl.nspec();
    hipLaunchKernelGGL((k_mag_to_user<T>), dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, pl.stream,
                       pl.tf_spec.template as<C>(), v_out, pl.B(), pl.Tn(), pl.n_freq);
  } else {
    SI_TRY((tf_mel_forward<P, T>(pl)));
    const int64_t n = (int64_t)pl.B() * pl.Tn() * pl.tf_mels;
    hipLaunchKernelGGL((k_log1p_to_user<T>), dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, pl.stream,
                       pl.tf_v.template as<T>(), v_out, pl.B(), pl.Tn(), pl.tf_mels);
  }
  SI_HIP(hipGetLastError());
  return SPECINV_OK;
}

template <typename P, typename T>
int tf_loss_grad(P& pl, const T* x, int64_t len, const T* target, double* loss, T* grad, double* loss_dev = nullptr,
                 bool with_stats = false, const T* stat_d = nullptr) {
  // with_stats: loss_dev[0 .. 4] = {loss, g.d, sum|g|, max|g|, max|d|} (stat_d == nullptr: d = g) - from the objective's own launch
  // where the one-launch kernel serves the configuration, by a pass over g and d otherwise
  SI_CHECK(pl.tf_kind >= 0, SPECINV_ESTATE, "specinv_transform_setup has not been called");
  SI_CHECK(x && target && (loss || loss_dev) && grad, SPECINV_EINVAL, "null pointer");
  SI_CHECK(!with_stats || loss_dev, SPECINV_EINVAL, "statistics go to device memory");
  using C = cplx<T>;
  const int64_t BT = (int64_t)pl.B() * pl.Tn();
  {
    const int64_t tcheck = 1 + (len + 2 * pl.pad - pl.N()) / pl.cfg.hop_length;
    SI_CHECK(len + 2 * pl.pad >= pl.N() && tcheck == pl.Tn(), SPECINV_EINVAL,
             "signal length %lld gives %lld frames, plan has %d", (long long)len, (long long)tcheck, pl.Tn());
  }
  if constexpr (std::is_same<T, float>::value) {
    bool used = false;
    if (with_stats && (((uintptr_t)stat_d | (uintptr_t)grad) & 15) == 0) {
      SI_TRY(pl.tf_rows.reserve((size_t)fast::kObjRows * fast::kObjStatRow * sizeof(double)));
      fast::ObjStatReq sr{};
      sr.d = stat_d;
      sr.rows = pl.tf_rows.template as<double>();
      SI_TRY(tf_loss_grad_fused(pl, x, len, target, nullptr, grad, &used, loss_dev, nullptr, &sr));
      if (used) {
        const double numel = (double)BT * (pl.tf_kind == SPECINV_TF_MAG ? pl.n_freq : pl.tf_mels);
        hipLaunchKernelGGL(k_objective_finish_rows, dim3(1), dim3(256), 0, pl.stream, (const double*)sr.rows, 1.0 / numel, loss_dev, 5);
        SI_HIP(hipGetLastError());
        return SPECINV_OK;
      }
    } else {
      SI_TRY(tf_loss_grad_fused(pl, x, len, target, loss, grad, &used, loss_dev));
      if (used && with_stats) return lb_stats(pl, (const T*)grad, stat_d ? stat_d : (const T*)grad, (int64_t)pl.B() * len, nullptr, loss_dev + 1);
      if (used) return SPECINV_OK;
    }
  }
  if (with_stats) {
    SI_TRY((tf_loss_grad<P, T>(pl, x, len, target, loss, grad, loss_dev)));
    return lb_stats(pl, (const T*)grad, stat_d ? stat_d : (const T*)grad, (int64_t)pl.B() * len, nullptr, loss_dev + 1);
  }
  if (const char* e = getenv("SPECINV_REQUIRE_FUSED_OBJECTIVE")) {     // tests: the shape must be on the one-launch kernel
    SI_CHECK(e[0] != '1', SPECINV_EUNSUPPORTED, "the one-launch objective does not cover this configuration");
  }
  pl.objective_kind = 0;
  SI_TRY(pl.tf_spec.reserve(pl.nspec() * sizeof(C)));
  SI_TRY(pl.stft_internal(x, len, pl.tf_spec.template as<C>()));
  const int nb = 1024;
  SI_TRY(pl.partials.reserve(std::max<size_t>((size_t)nb, 3 * 1024) * sizeof(double)));
  double numel;
  if (pl.tf_kind == SPECINV_TF_MAG) {
    numel = (double)pl.nspec();
    // the target comes in the caller's (B, F, T) layout; one tiled transpose makes the loss kernel's reads contiguous
    SI_TRY(pl.tf_v.reserve((size_t)pl.nspec() * sizeof(T)));
    SI_TRY((pl.template transpose<T>(target, pl.tf_v.template as<T>(), pl.n_freq, pl.Tn())));
    hipLaunchKernelGGL((k_mag_loss_grad<T>), dim3(nb), dim3(256), 0, pl.stream, pl.tf_spec.template as<C>(),
                       pl.tf_v.template as<T>(), 1, pl.B(), pl.Tn(), pl.n_freq, pl.N(), pl.cfg.onesided, 1.0 / numel,
                       pl.partials.template as<double>());
    SI_HIP(hipGetLastError());
  } else {
    numel = (double)BT * pl.tf_mels;
    SI_TRY((tf_mel_forward<P, T>(pl)));
    hipLaunchKernelGGL((k_logmel_loss_dm<T>), dim3(nb), dim3(256), 0, pl.stream, pl.tf_v.template as<T>(), target, pl.B(),
                       pl.Tn(), pl.tf_mels, 1.0 / numel, pl.partials.template as<double>());
    SI_HIP(hipGetLastError());
    if constexpr (std::is_same<T, float>::value) {
      const int mt = (pl.tf_mels + 31) / 32;
      if (mt <= 4) {
        const void* fn = mt == 1 ? (const void*)k_mel_backward_tiles<1> : mt == 2 ? (const void*)k_mel_backward_tiles<2>
                         : mt == 3 ? (const void*)k_mel_backward_tiles<3> : (const void*)k_mel_backward_tiles<4>;
        const size_t lds = (size_t)(mt * 32 * 33 + 4 * mt * 32 * 32) * sizeof(float);
        SI_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        cplx<float>* sp = pl.tf_spec.template as<cplx<float>>();
        const float* ml = pl.tf_mel_tiled_t.template as<float>();
        const float* dm = pl.tf_v.template as<float>();
        int64_t bt = BT;
        int F = pl.n_freq, nm = pl.tf_mels, nf = pl.N(), os = pl.cfg.onesided;
        void* kargs[] = {&sp, &ml, &dm, &bt, &F, &nm, &nf, &os};
        SI_HIP(hipLaunchKernel(fn, dim3((unsigned)ceil_div(BT, 32)), dim3(256), kargs, lds, pl.stream));
      } else {
        hipLaunchKernelGGL(k_mel_backward_mfma, dim3((unsigned)ceil_div(BT, 32), (unsigned)ceil_div(pl.n_freq, 32)), dim3(64), 0,
                           pl.stream, pl.tf_spec.template as<cplx<float>>(), pl.tf_mel.template as<float>(),
                           pl.tf_v.template as<float>(), BT, pl.n_freq, pl.tf_mels, pl.N(), pl.cfg.onesided);
      }
    } else {
      hipLaunchKernelGGL((k_mel_backward_valu<T>), dim3((unsigned)ceil_div(BT * pl.n_freq, 256)), dim3(256), 0, pl.stream,
                         pl.tf_spec.template as<C>(), pl.tf_mel.template as<T>(), pl.tf_v.template as<T>(), BT, pl.n_freq,
                         pl.tf_mels, pl.N(), pl.cfg.onesided);
    }
    SI_HIP(hipGetLastError());
  }
  // (the sum is finished before the next kernel that uses the partials scratch)
  hipLaunchKernelGGL(k_finish_scaled, dim3(1), dim3(256), 0, pl.stream, pl.partials.template as<double>(), nb, 1.0 / numel,
                     loss_dev ? loss_dev : pl.sums.template as<double>());
  SI_HIP(hipGetLastError());
  // frames of the gradient: irfft-style inverse with the forward scale
  SI_TRY(pl.grad_from_spec(pl.tf_spec.template as<C>(), grad, pl.fc.fwd_scale, len));
  if (loss_dev) return SPECINV_OK;
  SI_HIP(hipMemcpyAsync(loss, pl.sums.p, sizeof(double), hipMemcpyDeviceToHost, pl.stream));
  SI_HIP(si_stream_wait_short(pl.stream));
  return SPECINV_OK;
}

}  // namespace specinv
