// The log-mel objective (reference: the closure of torch_specinv/methods.py:545-550 with transform_fn = log1p(mel_fb @ |stft(x)|),
// BASELINE.json configs[4]) as a FRAME WALK - the form of the Griffin-Lim kernel (kernels_fast_td.h) instead of k_objective_logmel's
// tiles (kernels_objective.h):
//
//     loss = mean((log1p(Mel |STFT(x)|) - target)^2)   and   d loss / d x
//
// One 64-lane wave owns a chunk of consecutive frames of one item and takes each frame from samples to gradient on its own:
//   1. the sample window: three hop-blocks carried in registers from frame to frame, the fourth requested a frame ahead;
//   2. analysis on the wave-level FFT; |S| goes to a wave-private LDS column (the FFT's transpose scratch, idle by then), the unit
//      phases S/|S| stay in registers;
//   3. forward contraction mm = Mel |S| over the rows' bands: a lane per band segment, the segments of a row summed with xor
//      shuffles (objective_args.h: obj_build_walk);
//   4. V = log1p(mm), squared error, dM = 2/numel (V - T) / (1 + mm), a lane per row;
//   5. backward contraction dA = Mel^T dM in the lanes' own conjugate-pair order (two rows per bin);
//   6. G = dA S/|S| (Hermitian weights), inverse FFT, window, overlap-add in three register accumulators; the finished hop-block
//      goes to the gradient (or, in the padded margins, to `margins`); a chunk's last three accumulators go to `xtail`.
// k_objective_epilogue finishes seams, margins, loss and statistics exactly as it does for the tile kernel.
// No workgroup barrier after the tables are staged: the two waves of a SIMD drift apart and one contracts (LDS) while the other
// transforms (vector units).  The tile kernel ran its two FFT phases at the pace of the younger wave of each SIMD and could overlap
// nothing with them (0.37 of the issue slots); per frame and SIMD it took 6.7 us where the Griffin-Lim kernel - two
// transforms, projection, overlap-add - takes 2.6.
// Shapes: float32, one-sided, centred, hop = n_fft / 2, / 4, / 8, n_fft 1024 / 2048, len = (T - 1) hop, a filterbank obj_build_walk accepts;
// everything else stays on k_objective_logmel / the kernel chain.
#pragma once
#include "objective_args.h"

namespace specinv {
namespace fast {

// log1p, squared error and dM of one output (kernels_objective.h: obj_point)
__device__ __forceinline__ float walk_point(float v, float target, float dscale, double& s2) {
  const float u = 1.0f + v;
  float ru = fast_rcp(u);
  ru = fmaf(fmaf(-u, ru, 1.0f), ru, ru);
  const float d = (logf(u) + (v - (u - 1.0f)) * ru) - target;
  s2 += (double)d * (double)d;
  return (dscale * d) * ru;
}

// one hop-block of the read-only signal (the loaders of the Griffin-Lim kernels, told that the wave walks the whole item: no seams)
template <int R, int OV>
__device__ __forceinline__ void walk_load_block(const float* __restrict__ xrow, long long L, int T, int j, int lane, int pad_mode,
                                                v2f (&q)[R / OV]) {
  if constexpr (OV == 4) load_block4<R>(xrow, nullptr, L, T, 0, 0, T, j, lane, pad_mode, q);
  else load_block<R, OV>(xrow, nullptr, L, T, 0, 0, T, j, lane, pad_mode, q);
}

// ... of the iterate with a step pending: x_new = fma(t, (float)(c0 (double)g_prev), x_old) sample by sample (the float operations of
// k_lbd_direction_lean / k_lbd_lincomb_step), the previous gradient read through the same index map as the samples (padding included);
// a hop-block the chunk owns - signal blocks [own_lo, own_hi) - goes to the new iterate's buffer on the way
template <int R, int OV>
__device__ __forceinline__ void walk_load_step(const float* __restrict__ xrow, const float* __restrict__ gprow, float* __restrict__ xnew,
                                               float t, double c0, long long L, int T, int j, int own_lo, int own_hi, int lane,
                                               int pad_mode, v2f (&q)[R / OV]) {
  using O = Ovl<R, OV>;
  walk_load_block<R, OV>(xrow, L, T, j, lane, pad_mode, q);
  v2f gq[R / OV];
  walk_load_block<R, OV>(gprow, L, T, j, lane, pad_mode, gq);
#pragma unroll
  for (int i = 0; i < R / OV; ++i) {
    q[i].x = fmaf(t, (float)(c0 * (double)gq[i].x), q[i].x);
    q[i].y = fmaf(t, (float)(c0 * (double)gq[i].y), q[i].y);
  }
  if (j >= own_lo && j < own_hi) {
    v2f* dst = reinterpret_cast<v2f*>(xnew + (long long)(j - O::PB) * O::HOP);
#pragma unroll
    for (int i = 0; i < R / OV; ++i) dst[64u * i + (unsigned)lane] = q[i];
  }
}

template <int R, int OV>
__global__ __launch_bounds__(64 * kWalkWaves, 2) void k_objective_walk(ObjWalkArgs a) {
  using G = Geo<R>;
  using O = Ovl<R, OV>;
  constexpr int H = G::H, M = G::M, QU = O::QU, HOP = O::HOP, NB = O::NB, PB = O::PB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (a.ctl_eval != nullptr && *a.ctl_eval == 0) return;     // (the optimiser has stopped: the rest of its enqueued step is no-ops)
  float* grad_base = a.grad;
  if (a.ctl_cur != nullptr && (*a.ctl_cur ^ 1) != 0) grad_base = a.grad_alt;
  v2f* lds_win = reinterpret_cast<v2f*>(smem);
  v2f* lds_tw1 = lds_win + M;
  f32x4* blob = reinterpret_cast<f32x4*>(lds_tw1 + (R - 1) * 64);
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* priv = reinterpret_cast<char*>(blob + a.w.total) + (size_t)wib * (sizeof(v2f) * G::TR + sizeof(float) * kWalkMM);
  v2f* tr = reinterpret_cast<v2f*>(priv);                     // the FFT's transpose scratch ...
  float* col = reinterpret_cast<float*>(priv);                // ... and, between the transforms, the frame's |S| column
  float* mmv = reinterpret_cast<float*>(priv + sizeof(v2f) * G::TR);   // mm, then dM, of the frame

  const LaneConst<R> k = lane_consts<R>();
  const int lane = k.lane;
  const unsigned ulane = (unsigned)lane;
  // chunk of this wave: waves i and i + 4 of a workgroup share a SIMD, the older one (i < 4) takes the longer chunk of a skewed
  // pair (kernels_fast_td.h)
  int w = __builtin_amdgcn_readfirstlane(blockIdx.x * kWalkWaves + wib);
  if (a.skew != 0) w = 2 * ((int)blockIdx.x * 4 + (wib & 3)) + (wib >> 2);
  const bool idle = w >= a.n_waves;            // (a wave past the last chunk still helps staging the tables)
  if (idle) w = a.n_waves - 1;
  const int b = w / a.nchunks, c = w - b * a.nchunks;
  const int t_begin = chunk_begin(c, a.T, a.nchunks, a.skew), t_end = chunk_begin(c + 1, a.T, a.nchunks, a.skew);
  // the iterate: plain, or one of the optimiser's two buffers with a step still to be applied (ObjWalkArgs: the deferred step)
  const float* xbase = a.x;
  float* xnew = nullptr;
  bool pend = false;
  float t_p = 0.0f;
  double c0_p = 0.0;
  if (a.px_sel != nullptr) {
    const int sel = *a.px_sel;
    pend = *a.px_pending != 0;
    float* cur_buf = sel ? a.x_alt : const_cast<float*>(a.x);
    float* oth_buf = sel ? const_cast<float*>(a.x) : a.x_alt;
    xbase = pend ? oth_buf : cur_buf;              // (a pending step: read the old iterate, write the new one)
    if (pend) {
      xnew = cur_buf + (long long)b * a.len;
      t_p = (float)*a.pt_pend;
      c0_p = *a.pc0_pend;
    }
  }
  const float* xrow = xbase + (long long)b * a.len;
  const float* gprow = (grad_base == a.grad ? a.grad_alt : a.grad) + (long long)(pend ? b : 0) * a.len;   // the previous gradient (pending step only)
  float* go = grad_base + (long long)b * a.len;
  float* mgn = a.margins + (long long)b * 2 * (PB * HOP);
  const float* tg = a.target + (long long)b * a.n_mels * a.T;
  // the sample window (the signal is read-only here: no chunk seams to resolve - the loader is told it walks the whole item),
  // requested before the tables are staged: the chunk's first samples fly while the workgroup builds them
  v2f xq[NB][QU], xn[QU];
  // (signal blocks this chunk owns when it has to write the new iterate: the blocks its frames begin with, the last chunk the rest)
  const int own_lo = t_begin < PB ? PB : t_begin, own_hi = idle ? 0 : (t_end == a.T ? a.T + PB - 1 : t_end);
  auto load_x = [&](int j, v2f (&q)[QU]) {
    if (pend) walk_load_step<R, OV>(xrow, gprow, xnew, t_p, c0_p, a.len, a.T, j, own_lo, own_hi, lane, a.pad_mode, q);
    else walk_load_block<R, OV>(xrow, a.len, a.T, j, lane, a.pad_mode, q);
  };
#pragma unroll
  for (int q = 0; q < NB; ++q) load_x(t_begin + q, xq[q]);
  load_x(t_begin + NB, xn);

  // ---- tables: window, pass-1 twiddles, the filterbank (once per workgroup: the only barrier)
  for (int i = threadIdx.x; i < M; i += blockDim.x) lds_win[i] = v2f{a.window[2 * i], a.window[2 * i + 1]};
  for (int i = threadIdx.x; i < (R - 1) * 64; i += blockDim.x) {
    const int k1 = i / 64 + 1, l = i & 63;
    lds_tw1[i] = unit(2.0f * (float)((l * k1) % M) / (float)M);
  }
  for (int i = threadIdx.x; i < a.w.total; i += blockDim.x) blob[i] = a.blob[i];
  __syncthreads();

  if (idle) return;
  const int rows = a.w.rows;
  for (int i = lane; i < kWalkMM; i += 64) mmv[i] = 0.0f;     // (the entry behind the last row stays zero: dM[m0 + 1] of the top row)

  TwRegs<R> twr;
#pragma unroll
  for (int k1 = 1; k1 < R; ++k1) twr.w[k1 - 1] = lds_tw1[(k1 - 1) * 64 + lane];
  const float hs = 0.5f * a.fwd_scale;
  const int4* tasks = reinterpret_cast<const int4*>(blob + a.w.task_off);
  const f32x4* bw = blob + a.w.bw_off;
  const int* bm = reinterpret_cast<const int*>(blob + a.w.bm_off);
  const float* bmid = reinterpret_cast<const float*>(blob + a.w.mid_off);

  // a finished hop-block (padded block index jb) goes to the gradient, or - in the padding - to the margins the epilogue folds back
  auto emit = [&](int jb, const v2f (&v)[QU]) {
    v2f* dst;
    if (jb < PB) dst = reinterpret_cast<v2f*>(mgn + (long long)jb * HOP);
    else if (jb <= a.T + PB - 2) dst = reinterpret_cast<v2f*>(go + (long long)(jb - PB) * HOP);
    else dst = reinterpret_cast<v2f*>(mgn + (long long)(PB + jb - (a.T + PB - 1)) * HOP);
#pragma unroll
    for (int i = 0; i < QU; ++i) dst[64u * i + ulane] = v[i];
  };

  v2f acc[NB * QU];
#pragma unroll
  for (int i = 0; i < NB * QU; ++i) acc[i] = v2f{0.0f, 0.0f};
  double s2 = 0.0;
  for (int t = t_begin; t < t_end; ++t) {
    asm volatile("" ::: "memory");     // (window / table reads stay inside the loop)
    // ---- targets of this frame: a row per lane (rows 64 .. in a second register), requested now, used after the contraction
    float tg0 = 0.0f, tg1 = 0.0f;
    if (lane < rows) tg0 = tg[(long long)lane * a.T + t];
    if (lane + 64 < rows) tg1 = tg[(long long)(lane + 64) * a.T + t];

    // ---- 1 + 2. analysis
    v2f z[R];
#pragma unroll
    for (int i = 0; i < QU; ++i) {
#pragma unroll
      for (int q = 0; q < NB; ++q) z[q * QU + i] = xq[q][i] * lds_win[64 * (q * QU + i) + lane];
      z[NB * QU + i] = xn[i] * lds_win[64 * (NB * QU + i) + lane];
#pragma unroll
      for (int q = 0; q + 1 < NB; ++q) xq[q][i] = xq[q + 1][i];
      xq[NB - 1][i] = xn[i];
    }
    if (t + 1 < t_end) load_x(t + OV, xn);
    fft_forward_t<R>(z, k, twr, tr);
    v2f un[H], um[H], umid;
    {
      v2f rc[H];
#pragma unroll
      for (int m = H; m < R; ++m) {
        const v2f got = shfl2(z[m], k.partner);
        const v2f own = z[(m + 1) % R];
        rc[m - H] = v2f{lane == 0 ? own.x : got.x, lane == 0 ? own.y : got.y};
      }
      if (lane >= 1 && lane <= 3) col[M + lane] = 0.0f;       // the last bin quad's three bins beyond the spectrum
#pragma unroll
      for (int j = 0; j < H; ++j) {
        const v2f wk = pair_twiddle<R>(k.wn, j);
        const v2f zk = z[j], zm = rc[R - 1 - j - H];
        const v2f e2 = add_conj(zk, zm);
        const v2f tw = cmul_mi(wk, sub_conj(zk, zm));
        const v2f xk = (e2 + tw) * hs;
        const v2f xm = (e2 - tw) * v2f{hs, -hs};
        const float ak = fast_abs(xk), am = fast_abs(xm);
        col[lane + 64 * j] = ak;
        col[M - lane - 64 * j] = am;
        const float ik = ak > 0.0f ? fast_rcp(ak) : 0.0f, im = am > 0.0f ? fast_rcp(am) : 0.0f;
        un[j] = xk * ik;                                       // G = dA * S/|S|, 0 where |S| = 0
        um[j] = xm * im;
      }
      const v2f xmid = z[H] * v2f{a.fwd_scale, -a.fwd_scale};    // bin M/2 (lane 0)
      const float amid = fast_abs(xmid);
      if (lane == 0) col[M / 2] = amid;
      umid = xmid * (amid > 0.0f ? fast_rcp(amid) : 0.0f);
    }

    // ---- 3. forward contraction: a lane per band segment
    {
      const f32x4* c4 = reinterpret_cast<const f32x4*>(col);
      for (int p = 0; p < a.w.n_pass; ++p) {
        const int4 tk = tasks[p * 64 + lane];
        float v = 0.0f;
        for (int i = 0; i < tk.z; ++i) {
          const f32x4 wv = blob[tk.x + i], sv = c4[tk.y + i];
          v = fmaf(wv[0], sv[0], v);
          v = fmaf(wv[1], sv[1], v);
          v = fmaf(wv[2], sv[2], v);
          v = fmaf(wv[3], sv[3], v);
        }
        const int gs = a.w.pass_gs[p];
        for (int sh = 0; sh < gs; ++sh) v += __shfl_xor(v, 1 << sh, 64);
        if (tk.w & 0x10000) mmv[tk.w & 0xffff] = v;
      }
    }
    // ---- 4. log1p, squared error, dM (in place)
    {
      if (lane < rows) mmv[lane] = walk_point(mmv[lane], tg0, a.dscale, s2);
      if (lane + 64 < rows) mmv[lane + 64] = walk_point(mmv[lane + 64], tg1, a.dscale, s2);
    }
    // ---- 5 + 6. backward contraction in pair order, gradient frame
    v2f back[H];
#pragma unroll
    for (int j = 0; j < H; ++j) {
      const f32x4 wv = bw[j * 64 + lane];
      const int mi = bm[j * 64 + lane];
      const float* dk = mmv + (mi & 0xffff);
      const float* dq = mmv + ((unsigned)mi >> 16);
      const float dak = fmaf(wv[1], dk[1], wv[0] * dk[0]);
      const float daq = fmaf(wv[3], dq[1], wv[2] * dq[0]);
      const v2f wk = pair_twiddle<R>(k.wn, j);
      // interior bins of the one-sided spectrum count half (their mirror images carry the other half); bins 0 and M do not
      const float hw = ((lane + 64 * j) == 0 ? 1.0f : 0.5f) * a.fwd_scale;
      v2f ak = un[j] * (dak * hw);
      v2f am = um[j] * (daq * hw);
      if (j == 0 && lane == 0) {
        ak.y = 0.0f;
        am.y = 0.0f;
      }
      const v2f e2i = add_conj(ak, am);
      const v2f o2i = cmulc(sub_conj(ak, am), wk);
      z[j] = add_i(e2i, o2i);
      back[j] = conj_sub_i(e2i, o2i);
    }
    {
      const int mmid = reinterpret_cast<const int*>(bmid)[4];
      const float dmid = fmaf(bmid[1], mmv[mmid + 1], bmid[0] * mmv[mmid]) * a.fwd_scale;
      const v2f zmid = umid * v2f{dmid, -dmid};
#pragma unroll
      for (int m = H; m < R; ++m) {
        const v2f got = shfl2(back[R - 1 - m], k.partner);
        const v2f l0 = (m == H) ? zmid : back[(R - m) % H];
        z[m] = v2f{lane == 0 ? l0.x : got.x, lane == 0 ? l0.y : got.y};
      }
    }
    fft_inverse_t<R>(z, k, twr, tr);
#pragma unroll
    for (int u = 0; u < R; ++u) z[u] = z[u] * lds_win[64 * u + lane];
    // the frame's oldest hop-block is finished (up to the previous chunk's share of this chunk's first three: the seam)
    {
      v2f out[QU];
#pragma unroll
      for (int i = 0; i < QU; ++i) out[i] = acc[i] + z[i];
      emit(t, out);
    }
#pragma unroll
    for (int i = 0; i < QU; ++i) {
#pragma unroll
      for (int q = 0; q + 1 < NB; ++q) acc[q * QU + i] = acc[(q + 1) * QU + i] + z[(q + 1) * QU + i];
      acc[(NB - 1) * QU + i] = z[NB * QU + i];
    }
  }
  if (t_end == a.T) {
    // the chunk that holds the last frame finishes the NB blocks behind it as well
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      v2f out[QU];
#pragma unroll
      for (int i = 0; i < QU; ++i) out[i] = acc[q * QU + i];
      emit(a.T + q, out);
    }
  } else {
    // what this chunk's last three frames contribute to the next chunk's first three hop-blocks
    v2f* tl = reinterpret_cast<v2f*>(a.xtail + ((long long)b * a.nchunks + c) * NB * HOP);
#pragma unroll
    for (int q = 0; q < NB; ++q)
#pragma unroll
      for (int i = 0; i < QU; ++i) tl[(q * QU + i) * 64u + ulane] = acc[q * QU + i];
  }
  s2 = wave_sum(s2);
  if (lane == 0) a.partials[w] = s2;
}

}  // namespace fast
}  // namespace specinv
