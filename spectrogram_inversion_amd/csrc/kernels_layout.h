// Light kernels around the fused / frame kernels: layout conversions into the conjugate-pair order, phase_init in pair order, chunk-seam passes.  Compiled with the plan (specinv.hip).
#pragma once
#include "fast_core.h"

namespace specinv {
namespace fast {

// User layout (B, F, T) -> pair layout in one pass (32 x 32 tile transposed through LDS): reads are contiguous in
// time, writes are 8-byte (spectrum) / 4-byte (magnitude) pieces of the 16-byte pair records, contiguous in k.
// Bin f goes to pair k = f (first half) for f < M/2, to pair k = M - f (second half) for f > M/2, to `mid` for M/2.
// Two-sided spectrograms (rows = N = 2 M bins per item): `mirror` = 0 takes the rows 0 .. M as above, 1 the MIRROR rows - bin f's
// slot receives row N - f (f = 1 .. M - 1; the slots of bins 0 and M, which are their own mirror images, are left zero).
template <int R>
__global__ void k_user_spec_to_pairs(const v2f* __restrict__ in, v2f* __restrict__ pairs /* v4f records as 2 x v2f */,
                                     v2f* __restrict__ mid, int T, int rows = Geo<R>::M + 1, int mirror = 0) {
  using G = Geo<R>;
  constexpr int F = G::M + 1;
  __shared__ v2f tile[32][33];
  const int b = blockIdx.z, f0 = blockIdx.y * 32, t0 = blockIdx.x * 32;
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    const int f = f0 + i, t = t0 + threadIdx.x;
    if (f < F && t < T) {
      const bool none = mirror && (f == 0 || f == G::M);
      tile[i][threadIdx.x] = none ? v2f{0.0f, 0.0f} : in[((long long)b * rows + (mirror ? rows - f : f)) * T + t];
    }
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    const int t = t0 + i, f = f0 + threadIdx.x;
    if (f >= F || t >= T) continue;
    const long long fr = (long long)b * T + t;
    const v2f v = tile[threadIdx.x][i];
    if (2 * f == G::M) {
      mid[fr] = v;
    } else {
      const int kk = f < G::M / 2 ? f : G::M - f, half = f < G::M / 2 ? 0 : 1;
      pairs[(((fr * G::H) + (kk >> 6)) * 64 + (kk & 63)) * 2 + half] = v;
    }
  }
}

// same for the target magnitude; also per-block partial sums of m^2 (for the metrics)
template <int R>
__global__ void k_user_mag_to_pairs(const float* __restrict__ in, float* __restrict__ pairs /* v4f records */,
                                    float* __restrict__ mid, int T, double* __restrict__ partials, int rows = Geo<R>::M + 1,
                                    int mirror = 0) {
  using G = Geo<R>;
  constexpr int F = G::M + 1;
  __shared__ float tile[32][33];
  __shared__ double red[4];
  const int b = blockIdx.z, f0 = blockIdx.y * 32, t0 = blockIdx.x * 32;
  double s2 = 0.0;
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    const int f = f0 + i, t = t0 + threadIdx.x;
    if (f < F && t < T) {
      const bool none = mirror && (f == 0 || f == G::M);
      const float v = none ? 0.0f : in[((long long)b * rows + (mirror ? rows - f : f)) * T + t];
      tile[i][threadIdx.x] = v;
      s2 += (double)v * (double)v;
    }
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    const int t = t0 + i, f = f0 + threadIdx.x;
    if (f >= F || t >= T) continue;
    const long long fr = (long long)b * T + t;
    const float v = tile[threadIdx.x][i];
    if (2 * f == G::M) {
      mid[fr] = v;
    } else {
      const int kk = f < G::M / 2 ? f : G::M - f, second = f < G::M / 2 ? 0 : 1;
      const int j = kk >> 6;
      pairs[(((fr * (G::H / 2)) + (j >> 1)) * 64 + (kk & 63)) * 4 + (j & 1) * 2 + second] = v;
    }
  }
  s2 = wave_sum(s2);
  const int tid = threadIdx.y * blockDim.x + threadIdx.x;
  if ((tid & 63) == 0) red[tid >> 6] = s2;
  __syncthreads();
  if (tid == 0) {
    double tot = 0.0;
    for (int w = 0; w < (int)((blockDim.x * blockDim.y + 63) >> 6); ++w) tot += red[w];
    partials[((long long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = tot;
  }
}

// phase_init (methods.py:572-615) for the fused path: the starting spectrum and the target magnitude written straight in
// conjugate-pair order (what k_phase_init + k_user_spec_to_pairs + k_user_mag_to_pairs produce in three passes and one
// (B, F, T) complex round trip).  One workgroup per (item, pair row j): its 128 spectrogram rows - bins 64 j + l and
// M - (64 j + l) - are scanned over time exactly like k_phase_init does it (a wave per row, lanes = 64 consecutive time
// steps, float64 wave scan with each partial sum rounded to float32, the same operation order), 64 time steps at a time;
// the 128 x 64 block is transposed through LDS and leaves as 64 records of 1 KiB.  The bin M/2 rides with the last pair row.
template <int R>
__global__ __launch_bounds__(1024) void k_phase_init_pairs(const float* __restrict__ mag, v4f* __restrict__ P, v2f* __restrict__ Pmid,
                                                          float* __restrict__ mpairs, float* __restrict__ mmid,
                                                          double* __restrict__ partials, int T, int hop) {
  using G = Geo<R>;
  constexpr int M = G::M, F = M + 1, H = G::H, NFFT = G::N, LD = 129, WAVES = 16, RPW = 128 / WAVES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  v2f* ct = reinterpret_cast<v2f*>(smem);                   // [64 time steps][LD] complex values, column = row slot
  float* mt = reinterpret_cast<float*>(ct + 64 * LD);       // [64][LD] magnitudes
  __shared__ double red[16];
  __shared__ double carry_s[WAVES][RPW + 1];                // running phase of every row (the row loop is not unrolled)
  const int b = blockIdx.x / H, j = blockIdx.x - b * H;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float* base = mag + (long long)b * F * T;
  const float two_pi = 6.283185307179586476925286766559f;
  const bool has_mid = j == H - 1 && wave == 0;
  double* carry = carry_s[wave];
  if (lane <= RPW) carry[lane] = 0.0;
  double s2 = 0.0;

  // one row, 64 time steps: returns the complex value and the magnitude of (f, t0 + lane)
  auto row_step = [&](int f, int t, double* cr, v2f& val, float& m0) {
    float cur[5];
#pragma unroll
    for (int d = 0; d < 5; ++d) {
      const int g = f + d - 2;
      cur[d] = (t < T && g >= 0 && g < F) ? base[(long long)g * T + t] : 0.0f;
    }
    float om = 0.0f;
    m0 = cur[2];
    if (t < T) om = scatter_omega<float>(cur, f, F, two_pi, (float)NFFT, (float)hop);   // (:597-609)
    double v = (double)om;
    v = wave_scan_inclusive(v);
    v += *cr;                                               // (wave-private LDS slot: in-order within the wave)
    if (lane == 63) *cr = v;
    const float phi = (float)v;                              // :611
    double sn, cs;
    sincos_phase(phi, &sn, &cs);                             // :612
    val = v2f{m0 * (float)cs, m0 * (float)sn};               // :614
  };

  for (int t0 = 0; t0 < T; t0 += 64) {
    const int t = t0 + lane;
#pragma unroll 1
    for (int i = 0; i < RPW; ++i) {
      const int s = wave * RPW + i;                          // row slot: 0..63 bins 64 j + s, 64..127 bins M - (64 j + s - 64)
      const int f = s < 64 ? 64 * j + s : M - (64 * j + s - 64);
      v2f val;
      float m0;
      row_step(f, t, carry + i, val, m0);
      ct[lane * LD + s] = val;
      mt[lane * LD + s] = m0;
      if (t < T) s2 += (double)m0 * (double)m0;
    }
    if (has_mid) {                                           // bin M/2: time-major arrays, written as they come
      v2f val;
      float m0;
      row_step(M / 2, t, carry + RPW, val, m0);
      if (t < T) {
        Pmid[(long long)b * T + t] = val;
        mmid[(long long)b * T + t] = m0;
        s2 += (double)m0 * (double)m0;
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 64 / WAVES; ++q) {
      const int tl = wave * (64 / WAVES) + q, tt = t0 + tl;
      if (tt < T) {
        const long long fr = (long long)b * T + tt;
        const v2f a = ct[tl * LD + lane], bb = ct[tl * LD + 64 + lane];
        P[(fr * H + j) * 64 + lane] = v4f{a.x, a.y, bb.x, bb.y};
        // target record c = j / 2 holds (m[k_2c], m[M - k_2c], m[k_2c+1], m[M - k_2c+1]): this row fills one half of it
        *reinterpret_cast<v2f*>(mpairs + ((fr * (H / 2) + (j >> 1)) * 64 + lane) * 4 + (j & 1) * 2) =
            v2f{mt[tl * LD + lane], mt[tl * LD + 64 + lane]};
      }
    }
    __syncthreads();
  }
  const double tot = block_sum(s2, red);
  if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

// x += the tail partial sums (final waveform for get_wave)
template <int R, int OV>
__global__ void k_add_tails(float* __restrict__ x, const float* __restrict__ xtail, int T, int nchunks, long long L,
                            long long total, int skew) {
  constexpr int HOP = Ovl<R, OV>::HOP, NB = Ovl<R, OV>::NB, PB = Ovl<R, OV>::PB;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // (b, c, q, sample) over tails
  if (i >= total) return;
  const int smp = i % HOP;
  const int q = (i / HOP) % NB;
  const int c = (i / (NB * HOP)) % nchunks;
  const long long b = i / ((long long)NB * HOP * nchunks);
  if (c >= nchunks - 1) return;                       // the last chunk has no successor
  const int blk = chunk_begin(c + 1, T, nchunks, skew) + q; // padded-signal hop-block
  // register layout of a block: element (reg i2, lane l, comp e) <-> sample 128*i2 + 2*l + e
  const int i2 = smp / 128, rem = smp % 128;
  x[b * L + (long long)(blk - PB) * HOP + smp] += xtail[((b * nchunks + c) * NB + q) * HOP + (i2 * 64 + rem / 2) * 2 + (rem & 1)];
}

// ---- layout conversion between the frame-major (B*T, F) spectra and the pair layout ---------------
template <int R>
__global__ void k_spec_to_pairs(const v2f* __restrict__ spec, v4f* __restrict__ pairs, v2f* __restrict__ mid,
                                long long n_frames) {
  using G = Geo<R>;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // (frame, j, lane)
  if (i >= n_frames * G::H * 64) return;
  const int lane = i & 63;
  const int j = (i >> 6) % G::H;
  const long long f = i / (64 * G::H);
  const v2f* s = spec + f * (G::M + 1);
  const int kk = lane + 64 * j;
  const v2f a = s[kk], bb = s[G::M - kk];
  pairs[i] = v4f{a.x, a.y, bb.x, bb.y};
  if (lane == 0 && j == 0) mid[f] = s[G::M / 2];
}

template <int R>
__global__ void k_pairs_to_spec(const v4f* __restrict__ pairs, const v2f* __restrict__ mid, v2f* __restrict__ spec,
                                long long n_frames, int rows = Geo<R>::M + 1, int mirror = 0) {
  using G = Geo<R>;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_frames * G::H * 64) return;
  const int lane = i & 63;
  const int j = (i >> 6) % G::H;
  const long long f = i / (64 * G::H);
  v2f* s = spec + f * rows;
  const int kk = lane + 64 * j;
  const v4f p = pairs[i];
  if (!mirror) {
    s[kk] = v2f{p.x, p.y};
    s[G::M - kk] = v2f{p.z, p.w};
    if (lane == 0 && j == 0) s[G::M / 2] = mid[f];
  } else {                                           // (two-sided: the mirror rows N - kk and M + kk; none for kk = 0)
    if (kk != 0) {
      s[rows - kk] = v2f{p.x, p.y};
      s[G::M + kk] = v2f{p.z, p.w};
    }
    if (lane == 0 && j == 0) s[rows - G::M / 2] = mid[f];
  }
}

template <int R>
__global__ void k_mag_to_pairs(const float* __restrict__ mag, v4f* __restrict__ pairs, float* __restrict__ mid,
                               long long n_frames) {
  using G = Geo<R>;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // (frame, c, lane)
  if (i >= n_frames * (G::H / 2) * 64) return;
  const int lane = i & 63;
  const int c = (i >> 6) % (G::H / 2);
  const long long f = i / (64 * (G::H / 2));
  const float* s = mag + f * (G::M + 1);
  const int k0 = lane + 64 * (2 * c), k1 = lane + 64 * (2 * c + 1);
  pairs[i] = v4f{s[k0], s[G::M - k0], s[k1], s[G::M - k1]};
  if (lane == 0 && c == 0) mid[f] = s[G::M / 2];
}

// the envelope table of the wave-level kernels: 1 / envelope (they multiply), or - exact projection - the envelope itself (they
// divide, like methods.py:132)
static __global__ void k_reciprocal(const float* __restrict__ in, float* __restrict__ out, long long n, int exact) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = exact ? in[i] : 1.0f / in[i];
}


// x[n] = (own partial sum + the previous chunk's tail) * (1 / envelope), z_out[n] = x[n] - lr z_in[n] over the seam samples
static __global__ void k_hop_tails_td(float* __restrict__ x, float* __restrict__ z_out, const float* __restrict__ z_in,
                               const float* __restrict__ xtail, const float* __restrict__ env, float lr, int T, int nchunks,
                               int hop, int keep, int pad, long long L, long long total, int exact) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // (b, c - 1, j)
  if (i >= total) return;
  const int j = (int)(i % keep);
  const int c = (int)((i / keep) % (nchunks - 1)) + 1;
  const long long b = i / ((long long)keep * (nchunks - 1));
  const long long n = (long long)hop_chunk_begin(c, T, nchunks) * hop + j - pad;
  if (n < 0 || n >= L) return;
  const float sum = x[b * L + n] + xtail[(b * nchunks + (c - 1)) * keep + j];
  const float xv = exact ? __fdiv_rn(sum, env[n]) : sum * env[n];
  x[b * L + n] = xv;
  z_out[b * L + n] = fmaf(-lr, z_in[b * L + n], xv);
}

// out[n] += the previous chunk's tail over the first n_fft - hop samples of chunks 1.. (all inside the signal: a chunk
// is at least (n_fft - 1) / hop + 1 frames long)
static __global__ void k_hop_tails_raw(float* __restrict__ x, const float* __restrict__ xtail, int T, int nchunks, int hop, int keep,
                                int pad, long long L, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // (b, c - 1, j)
  if (i >= total) return;
  const int j = (int)(i % keep);
  const int c = (int)((i / keep) % (nchunks - 1)) + 1;
  const long long b = i / ((long long)keep * (nchunks - 1));
  const long long n = (long long)hop_chunk_begin(c, T, nchunks) * hop + j - pad;
  if (n < 0 || n >= L) return;
  x[b * L + n] += xtail[(b * nchunks + (c - 1)) * keep + j];
}

// x[n] = (own partial sum + the previous chunk's tail) * (1 / envelope) over the first n_fft - hop samples of chunks 1..
static __global__ void k_hop_tails(float* __restrict__ x, const float* __restrict__ xtail, const float* __restrict__ env, int T,
                            int nchunks, int hop, int keep, int pad, long long L, long long total, int exact) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // (b, c - 1, j)
  if (i >= total) return;
  const int j = (int)(i % keep);
  const int c = (int)((i / keep) % (nchunks - 1)) + 1;
  const long long b = i / ((long long)keep * (nchunks - 1));
  const long long n = (long long)hop_chunk_begin(c, T, nchunks) * hop + j - pad;
  if (n < 0 || n >= L) return;
  float* px = x + b * L + n;
  const float sum = *px + xtail[(b * nchunks + (c - 1)) * keep + j];
  *px = exact ? __fdiv_rn(sum, env[n]) : sum * env[n];
}

}  // namespace fast
}  // namespace specinv
