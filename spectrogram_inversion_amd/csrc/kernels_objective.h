// The L-BFGS objective of the log-mel path in ONE launch (reference: the closure of torch_specinv/methods.py:545-550 with
// transform_fn = log1p(mel_fb @ |stft(x)|), BASELINE.json configs[4]):
//
//     loss = mean((log1p(Mel |STFT(x)|) - target)^2)   and   d loss / d x
//
// One 8-wave workgroup owns a tile of up to 16 consecutive frames of one batch item; the spectrum never leaves the chip.
//   1. each wave transforms two frames with the wave-level FFT of kernels_fast.h, writes |S| into an LDS tile
//      [bin][frame] and keeps the unit phases S/|S| in registers;
//   2. forward mel contraction on the matrix cores (v_mfma_f32_16x16x4_f32: exact float32, an fmaf chain): the eight
//      waves split the 1025 bins (K), partial accumulators are added through LDS in a fixed order;
//   3. log1p, squared error (float64 partial sum per tile), dM = 2/numel (V - T) / (1 + Mel|S|) -> LDS;
//   4. backward contraction dA = Mel^T dM on the matrix cores, the waves split the bin tiles; dA overwrites the |S| tile;
//   5. each wave forms G = dA S/|S| (interior bins halved: the Hermitian extension) for its two frames, runs the inverse
//      FFT and applies the window;
//   6. the frames are overlap-added into one LDS span of the tile in a fixed order (waves whose frames overlap take
//      turns), the span's first frames*hop samples go to the gradient, the remaining n_fft - hop (what the tile adds
//      to its successor's samples) to `xtail`; k_hop_tails_raw and k_grad_fold_margins of the unfused path finish
//      the seams and the padding.
// HBM traffic per frame: 4*hop read + 4*hop written + 4*n_mels target (+ the seams, + the filterbank, which every
// workgroup streams from L2): SURVEY 8d's 8h + 4 n_mels.  The filterbank is read from two copies tiled in operand order
// (k_mel_tile16): one 16-byte load per lane feeds four MFMAs.
#pragma once

namespace specinv {
namespace fast {

using f32x4 = float __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma_16x16x4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

constexpr int kObjWaves = 8;      // waves per workgroup
constexpr int kObjTile = 16;      // frames per tile (the N of the MFMA), two per wave
constexpr int kObjRow = 17;       // LDS row stride of the [bin][frame] tiles (odd: column reads are conflict-free)

struct ObjArgs {
  const float* x;          // (B, len)
  float* grad;             // (B, len)
  float* margins;          // (B, 2, pad): gradient w.r.t. the padded samples either side of the signal
  float* xtail;            // (B, nchunks, n_fft - hop)
  const float* target;     // (B, n_mels, T), the caller's layout
  const f32x4* melA;       // forward operand tiles  [KQ][MT][64] x 4 k-steps
  const f32x4* melB;       // backward operand tiles [KQ][MT][64] x 4 k-steps
  const float* window;
  double* partials;        // [B * nchunks] squared-error sums
  long long len;
  int T, nchunks, hop, pad, pad_mode, n_mels;
  float fwd_scale;
  float dscale;            // 2 / numel
};

template <int R, int MT>
struct ObjGeo {
  using G = Geo<R>;
  static constexpr int F = G::M + 1;
  static constexpr int KQ = (F + 15) / 16;            // groups of 16 bins
  static constexpr int FP = 16 * KQ;                  // rows of the |S| / dA tile
  static constexpr int UNI_TR = kObjWaves * G::TR * 2;            // floats: FFT transpose scratch of the waves
  static constexpr int UNI_RED = kObjWaves * MT * 4 * 64;         // floats: partial accumulators of the forward contraction
  static constexpr int UNI = UNI_TR > UNI_RED ? UNI_TR : UNI_RED; // (the output span must fit too: checked on the host)
  static constexpr size_t lds_bytes() {
    return sizeof(v2f) * G::M + sizeof(float) * ((size_t)FP * kObjRow + 16 * MT * kObjRow + UNI);
  }
};

template <int R, int MT>
__global__ __launch_bounds__(64 * kObjWaves, 2) void k_objective_logmel(ObjArgs a) {
  using G = Geo<R>;
  using OG = ObjGeo<R, MT>;
  constexpr int H = G::H, M = G::M, N = G::N, F = OG::F, KQ = OG::KQ, FP = OG::FP, RS = kObjRow;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  v2f* lds_win = reinterpret_cast<v2f*>(smem);
  float* tile = reinterpret_cast<float*>(lds_win + M);
  float* dmt = tile + FP * RS;
  float* uni = dmt + 16 * MT * RS;
  __shared__ double lsum[kObjWaves];
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  v2f* tr = reinterpret_cast<v2f*>(uni) + wib * G::TR;
  const LaneConst<R> k = lane_consts<R>();
  const int lane = k.lane;

  for (int i = threadIdx.x; i < M; i += blockDim.x) lds_win[i] = v2f{a.window[2 * i], a.window[2 * i + 1]};
  for (int i = threadIdx.x; i < (FP - F) * RS; i += blockDim.x) tile[F * RS + i] = 0.0f;   // rows the zero-padded filterbank meets
  TwRegs<R> twr;
#pragma unroll
  for (int k1 = 1; k1 < R; ++k1) twr.w[k1 - 1] = unit(2.0f * (float)((lane * k1) % M) / (float)M);   // W_M^(lane*k1)

  const int b = blockIdx.x / a.nchunks, c = blockIdx.x - b * a.nchunks;
  const int t0 = hop_chunk_begin(c, a.T, a.nchunks), t1 = hop_chunk_begin(c + 1, a.T, a.nchunks);
  const int nfr = t1 - t0;                                  // <= 16
  const float* xrow = a.x + (long long)b * a.len;
  const float hs = 0.5f * a.fwd_scale;
  __syncthreads();

  // ---- 1. analysis of this wave's two frames ---------------------------------------------------------------------------
  v2f un[2][H], um[2][H], umid[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int n = 2 * wib + i;
    v2f z[R];
    if (n < nfr) {
      load_frame_raw<R>(xrow, a.len, (long long)(t0 + n) * a.hop - a.pad, lane, a.pad_mode, z);
    } else {
#pragma unroll
      for (int u = 0; u < R; ++u) z[u] = v2f{0.0f, 0.0f};   // a frame beyond the tile: zeros all the way through
    }
#pragma unroll
    for (int u = 0; u < R; ++u) z[u] = z[u] * lds_win[64 * u + lane];
    fft_forward_t<R>(z, k, twr, tr);
    v2f rc[H];
#pragma unroll
    for (int m = H; m < R; ++m) {
      const v2f got = shfl2(z[m], k.partner);
      const v2f own = z[(m + 1) % R];
      rc[m - H] = v2f{lane == 0 ? own.x : got.x, lane == 0 ? own.y : got.y};
    }
#pragma unroll
    for (int j = 0; j < H; ++j) {
      const v2f wk = j == 0 ? k.wn : cmul(k.wn, w64(j * (32 / R)));
      const v2f zk = z[j], zm = rc[R - 1 - j - H];
      const v2f e2 = add_conj(zk, zm);
      const v2f tw = cmul(mul_mi(wk), sub_conj(zk, zm));
      const v2f xk = (e2 + tw) * hs;
      const v2f xm = (e2 - tw) * v2f{hs, -hs};
      const float ak = fast_abs(xk), am = fast_abs(xm);
      const int kk = lane + 64 * j;
      tile[kk * RS + n] = ak;
      tile[(M - kk) * RS + n] = am;
      un[i][j] = ak > 0.0f ? v2f{xk.x / ak, xk.y / ak} : v2f{0.0f, 0.0f};   // G = dA * S/|S|, 0 where |S| = 0
      um[i][j] = am > 0.0f ? v2f{xm.x / am, xm.y / am} : v2f{0.0f, 0.0f};
    }
    const v2f xmid = z[H] * v2f{a.fwd_scale, -a.fwd_scale};                  // bin M/2 (lane 0)
    const float amid = fast_abs(xmid);
    if (lane == 0) tile[(M / 2) * RS + n] = amid;
    umid[i] = amid > 0.0f ? v2f{xmid.x / amid, xmid.y / amid} : v2f{0.0f, 0.0f};
  }
  __syncthreads();

  // ---- 2. forward contraction mm[m, n] = sum_f Mel[m, f] |S|[f, n]: the waves split K ----------------------------------
  {
    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    const int q0 = (KQ * wib) / kObjWaves, q1 = (KQ * (wib + 1)) / kObjWaves;
    f32x4 av[MT], an[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) av[mt] = a.melA[((long long)q0 * MT + mt) * 64 + lane];
    for (int kq = q0; kq < q1; ++kq) {
      const int kn = kq + 1 < q1 ? kq + 1 : kq;                // next group's operands fly during this group's MFMAs
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) an[mt] = a.melA[((long long)kn * MT + mt) * 64 + lane];
      float bv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) bv[j] = tile[(16 * kq + 4 * j + (lane >> 4)) * RS + (lane & 15)];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = mfma_16x16x4(av[mt][j], bv[j], acc[mt]);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) av[mt] = an[mt];
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) uni[((wib * MT + mt) * 4 + r) * 64 + lane] = acc[mt][r];
  }
  __syncthreads();

  // ---- 3. V = log1p(mm), squared error, dM = 2/numel (V - T) / (1 + mm) ------------------------------------------------
  {
    double s2 = 0.0;
    for (int q = wib; q < 4 * MT; q += kObjWaves) {
      const int mt = q >> 2, r = q & 3;
      float v = 0.0f;
#pragma unroll
      for (int w = 0; w < kObjWaves; ++w) v += uni[((w * MT + mt) * 4 + r) * 64 + lane];   // fixed order
      const int m = 16 * mt + 4 * (lane >> 4) + r, n = lane & 15;   // D[i = m][j = n]: col = lane & 15, row = 4 (lane >> 4) + r
      float dm = 0.0f;
      if (m < a.n_mels && n < nfr) {
        const float d = log1pf(v) - a.target[((long long)b * a.n_mels + m) * a.T + t0 + n];
        s2 += (double)d * (double)d;
        dm = a.dscale * d / (1.0f + v);
      }
      dmt[m * RS + n] = dm;
    }
    s2 = wave_sum(s2);
    if (lane == 0) lsum[wib] = s2;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = 0.0;
    for (int w = 0; w < kObjWaves; ++w) tot += lsum[w];
    a.partials[blockIdx.x] = tot;
  }

  // ---- 4. backward contraction dA[f, n] = sum_m Mel[m, f] dM[m, n]: the waves split the bin tiles ------------------------
  {
    float bb[MT][4];
#pragma unroll
    for (int kq = 0; kq < MT; ++kq)
#pragma unroll
      for (int j = 0; j < 4; ++j) bb[kq][j] = dmt[(16 * kq + 4 * j + (lane >> 4)) * RS + (lane & 15)];
    for (int ft = wib; ft < KQ; ft += 2 * kObjWaves) {
      const int f2 = ft + kObjWaves < KQ ? ft + kObjWaves : ft;   // two tiles in flight: independent accumulator chains
      f32x4 a0[MT], a1[MT];
#pragma unroll
      for (int kq = 0; kq < MT; ++kq) {
        a0[kq] = a.melB[((long long)ft * MT + kq) * 64 + lane];
        a1[kq] = a.melB[((long long)f2 * MT + kq) * 64 + lane];
      }
      f32x4 c0 = f32x4{0.0f, 0.0f, 0.0f, 0.0f}, c1 = c0;
#pragma unroll
      for (int kq = 0; kq < MT; ++kq)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          c0 = mfma_16x16x4(a0[kq][j], bb[kq][j], c0);
          c1 = mfma_16x16x4(a1[kq][j], bb[kq][j], c1);
        }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        tile[(16 * ft + 4 * (lane >> 4) + r) * RS + (lane & 15)] = c0[r];
        if (f2 != ft) tile[(16 * f2 + 4 * (lane >> 4) + r) * RS + (lane & 15)] = c1[r];
      }
    }
  }
  __syncthreads();

  // ---- 5. gradient frames: G = dA S/|S| (Hermitian weights), inverse FFT, window --------------------------------------------
  v2f fr[2][R];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int n = 2 * wib + i;
    v2f z[R], back[H];
#pragma unroll
    for (int j = 0; j < H; ++j) {
      const v2f wk = j == 0 ? k.wn : cmul(k.wn, w64(j * (32 / R)));
      const int kk = lane + 64 * j;
      // interior bins of the one-sided spectrum count half (their mirror images carry the other half); bins 0 and M do not
      const float hw = (kk == 0 ? 1.0f : 0.5f) * a.fwd_scale;
      v2f ak = un[i][j] * (tile[kk * RS + n] * hw);
      v2f am = um[i][j] * (tile[(M - kk) * RS + n] * hw);
      if (j == 0 && lane == 0) {
        ak.y = 0.0f;
        am.y = 0.0f;
      }
      const v2f e2i = add_conj(ak, am);
      const v2f o2i = cmulc(sub_conj(ak, am), wk);
      z[j] = add_i(e2i, o2i);
      back[j] = conj_sub_i(e2i, o2i);
    }
    const float dmid = tile[(M / 2) * RS + n] * a.fwd_scale;
    const v2f zmid = umid[i] * v2f{dmid, -dmid};
#pragma unroll
    for (int m = H; m < R; ++m) {
      const v2f got = shfl2(back[R - 1 - m], k.partner);
      const v2f l0 = (m == H) ? zmid : back[(R - m) % H];
      z[m] = v2f{lane == 0 ? l0.x : got.x, lane == 0 ? l0.y : got.y};
    }
    fft_inverse_t<R>(z, k, twr, tr);
#pragma unroll
    for (int u = 0; u < R; ++u) fr[i][u] = z[u] * lds_win[64 * u + lane];
  }
  __syncthreads();                                            // every inverse transform is done: the scratch becomes the span

  // ---- 6. overlap-add in LDS, fixed order -------------------------------------------------------------------------------------
  float* span = uni;
  const int span_len = (nfr - 1) * a.hop + N;
  for (int s = threadIdx.x; s < span_len; s += blockDim.x) span[s] = 0.0f;
  __syncthreads();
  // wave w's frames reach into the frames of waves w+1 .. w+P-1: P groups of waves take turns
  int P = (a.hop + N - 1) / (2 * a.hop) + 1;
  if (P > kObjWaves) P = kObjWaves;
  for (int ph = 0; ph < P; ++ph) {
    if (wib % P == ph) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int n = 2 * wib + i;
        if (n < nfr) {
          const int off = n * a.hop;
          if ((off & 1) == 0) {
            v2f* s2 = reinterpret_cast<v2f*>(span + off);
#pragma unroll
            for (int u = 0; u < R; ++u) s2[64 * u + lane] = s2[64 * u + lane] + fr[i][u];
          } else {
#pragma unroll
            for (int u = 0; u < R; ++u) {
              span[off + 128 * u + 2 * lane] += fr[i][u].x;
              span[off + 128 * u + 2 * lane + 1] += fr[i][u].y;
            }
          }
        }
      }
    }
    __syncthreads();
  }
  // the tile's own frames*hop samples are final up to the previous tile's tail; the rest of the span is this tile's tail
  const long long p0 = (long long)t0 * a.hop;
  const int owned = nfr * a.hop, keep = N - a.hop;
  const bool last = c == a.nchunks - 1;
  float* go = a.grad + (long long)b * a.len;
  float* mg = a.margins + (long long)b * 2 * a.pad;
  float* tl = a.xtail + ((long long)b * a.nchunks + c) * keep;
  for (int s = threadIdx.x; s < span_len; s += blockDim.x) {
    const float v = span[s];
    if (s < owned || last) {
      const long long nn = p0 + s - a.pad;
      if (nn >= 0 && nn < a.len) go[nn] = v;
      else if (nn < 0) mg[p0 + s] = v;
      else if (nn - a.len < a.pad) mg[a.pad + (nn - a.len)] = v;
    } else {
      tl[s - owned] = v;
    }
  }
}

// filterbank (n_mels, F) -> the two operand-ordered copies, zero padded:
//   A[((kq * MT + mt) * 64 + lane) * 4 + j] = Mel[16 mt + (lane & 15)][16 kq + 4 j + (lane >> 4)]     (forward: A[i = m][k = f])
//   B[((ft * MT + kq) * 64 + lane) * 4 + j] = Mel[16 kq + 4 j + (lane >> 4)][16 ft + (lane & 15)]     (backward: A[i = f][k = m])
__global__ void k_mel_tile16(const float* __restrict__ mel, float* __restrict__ A, float* __restrict__ B, int F, int n_mels,
                             int KQ, int MT) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)KQ * MT * 256) return;
  const int j = i & 3, lane = (i >> 2) & 63;
  const int inner = (int)((i >> 8) % MT), outer = (int)(i / (256LL * MT));
  {
    const int m = 16 * inner + (lane & 15), f = 16 * outer + 4 * j + (lane >> 4);
    A[i] = (m < n_mels && f < F) ? mel[(long long)m * F + f] : 0.0f;
  }
  {
    const int m = 16 * inner + 4 * j + (lane >> 4), f = 16 * outer + (lane & 15);
    B[i] = (m < n_mels && f < F) ? mel[(long long)m * F + f] : 0.0f;
  }
}

}  // namespace fast
}  // namespace specinv
