// Griffin-Lim on the fused shapes with the momentum carried as a signal: fused_td_body and its two launch forms (k_fused4_td for the
// headline shapes, k_fused_td<R, OV> for every other overlap).  Built on fast_core.h (the wave-level FFT, the block loaders load_block /
// load_block4, FastArgs); compiled in tu_td_*.hip; the host side is FastState<float>::launch_td (fast_state.h).
#pragma once
#include "fast_core.h"

namespace specinv {
namespace SI_FAST_NS {

// ---- Griffin-Lim with the momentum carried in the time domain (every fused shape: hop = n_fft/2, /4, /8) ---------------------
// methods.py:243-244 keep pre_t = STFT(x_t) - lr * pre_{t-1}, a (B, F, T) complex array read and written every iteration
// (16 F of the 8 hop + 20 F bytes a frame-iteration moves).  The STFT (padding included) is linear, so
//     pre_t = STFT(z_t) + (-lr)^t * c0,     z_t = x_t - lr * z_{t-1},  z_0 = 0,
// with c0 the starting spectrum: the recursion can run on the (B, L) signal z instead.  This kernel transforms z_t's
// frames (one forward FFT, as before), adds the geometrically vanishing c0 term while it is above 2^-30 (EARLY: c0 is read,
// never written), projects, synthesises x_{t+1} and writes z_{t+1} = x_{t+1} - lr * z_t next to it: the hop-block of z_t that
// an output block needs is the oldest block of the frame just analysed, still in registers.  Per frame-iteration: 4 hop (z in)
// + 4 F (target) + 8 hop (z, x out) instead of 8 hop + 20 F bytes; the same arithmetic up to the rounding of where the linear
// combination is taken (time domain here, frequency domain in the reference).  The evaluating variant transforms x_t's frames
// as well (|STFT(x_t)| is what the metric wants, methods.py:242).
//   a.x_in / a.x_out  : z_t / z_{t+1}          a.x2_in / a.x2_out : x_t (EVAL only) / x_{t+1} (nullptr: not wanted)
//   a.xtail_in / _out : chunk seams, shared by x and z (the - lr * z_t term goes to the block's owner)
//   a.P_in, a.Pmid_in : c0 pairs (EARLY)        a.tds : (-lr)^t
#ifndef SPECINV_TD_STAMPS          // diagnostic build: per-phase s_memtime sums of every wave of k_fused4_td (tools/td_stamps.py)
#define SPECINV_TD_STAMPS 0
#endif
#if SPECINV_TD_STAMPS
#define TD_STAMP(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); stamp_sum[i] += now_ - stamp_prev; stamp_prev = now_; } while (0)
#else
#define TD_STAMP(i) do { } while (0)
#endif

template <bool FIRST, typename A, typename B>
__device__ __forceinline__ const auto& td_pick(const A& a, const B& b) {
  if constexpr (FIRST) return a;
  else return b;
}


// one hop-block of the sample window (the tuned n_fft/4 copy of the loader where it applies)
template <int R, int OV>
__device__ __forceinline__ void td_load_block(const float* __restrict__ xrow, const float* __restrict__ tailrow, long long L,
                                              int T, int c, int t_begin, int t_end, int j, int lane, int pad_mode,
                                              v2f (&q)[R / OV]) {
  if constexpr (OV == 4) load_block4<R>(xrow, tailrow, L, T, c, t_begin, t_end, j, lane, pad_mode, q);
  else load_block<R, OV>(xrow, tailrow, L, T, c, t_begin, t_end, j, lane, pad_mode, q);
}

#ifndef SPECINV_TD_WKREG
#define SPECINV_TD_WKREG 1
#endif
#ifndef SPECINV_TD_MINWAVES
#define SPECINV_TD_MINWAVES 2
#endif
#ifndef SPECINV_TD_ENVREG          // 1: one block of the (hop-periodic) envelope reciprocal stays in registers (8 at n_fft 2048)
#define SPECINV_TD_ENVREG 1
#endif
#ifndef SPECINV_TD_ABLATE          // timing-only builds (wrong results): 1 target from frame 0 (L2-resident), 2 no output stores,
#define SPECINV_TD_ABLATE 0        // 4 z samples from the first hop-blocks (L2-resident)
#endif
template <int R, int OV, bool EARLY, bool EVAL>
__device__ __forceinline__ void fused_td_body(const FastArgs& a) {
  using G = Geo<R>;
  using O = Ovl<R, OV>;
  constexpr int H = G::H, M = G::M, QU = O::QU, HOP = O::HOP, NB = O::NB, PB = O::PB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // The synthesis window carries the inverse transform's scale (a second table): one multiplication less per bin pair in the
  // projection; 1 / n_fft is a power of two, so nothing changes in the result (normalized=True never takes this kernel when the
  // reference's operation chain is asked for: FastState::begin_t).
  constexpr bool WSCALE = SPECINV_IEEE || SPECINV_RSQ;
  v2f* lds_win = reinterpret_cast<v2f*>(smem);
  v2f* lds_wins = WSCALE ? lds_win + M : lds_win;
  v2f* lds_tw1 = lds_win + 2 * M;      // (the host reserves both tables for every build: Geo::lds_bytes_td)
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  v2f* tr = lds_tw1 + (R - 1) * 64 + wib * G::TR;

  for (int i = threadIdx.x; i < M; i += blockDim.x) {
    const v2f wv = v2f{a.window[2 * i], a.window[2 * i + 1]};
    lds_win[i] = wv;
    if (WSCALE) lds_wins[i] = wv * a.inv_scale;
  }
  for (int i = threadIdx.x; i < (R - 1) * 64; i += blockDim.x) {
    const int k1 = i / 64 + 1, l = i & 63;
    lds_tw1[i] = unit(2.0f * (float)((l * k1) % M) / (float)M);   // W_M^(l*k1)
  }
  __syncthreads();

  int w = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + wib);   // (scalar: the frame loop and its padding tests branch on the scalar unit; left to the compiler the work-group size may arrive in a vector register and make all of it per-lane)
  if (a.skew == 0 && w >= a.n_waves) return;     // (skewed chunks: the test follows the mapping, which permutes inside a workgroup)
  const LaneConst<R> k = lane_consts<R>();
  const int lane = k.lane;
  const unsigned ulane = (unsigned)lane;
  // Even chunks: wave w walks chunk w mod nchunks of item w / nchunks.  Skewed chunks (a launch of exactly two waves per SIMD):
  // the older wave of a SIMD, which the arbiter serves first - it runs 1.7 x as fast as its neighbour while both are there -
  // takes the even (longer) chunk of a pair, the younger the odd one.  In 8-wave workgroups (the headline shape) the role is the
  // wave's index in the workgroup / 4.  In 4-wave workgroups (the other overlaps) it is inferred from the dispatch order: the
  // dispatcher hands every CU its first workgroup before any gets a second, so the first half of the waves are the older ones
  // (1024 of 1024 in every dump of tools/td_waves.py; a launch placed differently is merely less balanced).  (Drawing the chunk
  // by the slot actually occupied, with two atomic counters, balanced the waves to 4 % - and cost more in 1024 same-address
  // atomics per counter than it won.)
  int b = w / a.nchunks, c = w - b * a.nchunks;
  if (a.skew >= 0x10000) {       // three waves per SIMD, 12-wave workgroups: the chunk triples of k_fused4 (kernels_fused.h)
    const int cg = 3 * ((int)blockIdx.x * 4 + (wib & 3)) + (wib >> 2);
    b = cg / a.nchunks;
    c = cg - b * a.nchunks;
    w = cg;
  } else if (a.skew != 0 && (blockDim.x >> 6) == 8) {
    // 8-wave workgroups, one per CU: the two waves of a SIMD are waves i and i + 4 of one workgroup, dispatched in that order -
    // the hardware slot is the wave's index in the workgroup / 4 whatever else runs on the chip
    const int cg = 2 * ((int)blockIdx.x * 4 + (wib & 3)) + (wib >> 2);
    b = cg / a.nchunks;
    c = cg - b * a.nchunks;
    w = cg;
  } else if (a.skew != 0) {
    const int half = a.n_waves >> 1, second = w >= half ? 1 : 0, wl = w - second * half, pairs = a.nchunks >> 1;
    b = wl / pairs;
    c = 2 * (wl - b * pairs) + second;
    w = b * a.nchunks + c;      // (the index of the chunk walked: partial sums and stamps go by it)
  }
  if (a.skew != 0 && w >= a.n_waves) return;
  const int t_begin = chunk_begin(c, a.T, a.nchunks, a.skew);
  const int t_end = chunk_begin(c + 1, a.T, a.nchunks, a.skew);
  const float* zrow = a.x_in + (long long)b * a.L;
  const float* tailrow = a.xtail_in + (long long)b * a.nchunks * NB * HOP;
  float* zorow = a.x_out + (long long)b * a.L;
  // x_{t+1} is only written when somebody will read it (a.x2_out given): the launch before an evaluating one and the last one of
  // a call; the recursion itself runs on z
  const bool write_x = a.x2_out != nullptr;
  float* xorow = a.x2_out + (long long)b * a.L;
  const float half_scale = 0.5f * a.fwd_scale;
  const float tds_raw = a.tds / half_scale;
  const float nlr = -a.coef;

  v2f acc[NB * QU];
#pragma unroll
  for (int i = 0; i < NB * QU; ++i) acc[i] = v2f{0.0f, 0.0f};
  double sd = 0.0, so = 0.0;
  // pass-1 twiddles in registers - except in the evaluating variants (one launch in ten), which have no room for them at
  // n_fft 2048 and read the LDS table instead (measured: the same step time, no spills in the late one)
  constexpr bool TWLDS = (EVAL && R >= 16) || SPECINV_TD_MINWAVES == 3;
  TwRegs<TWLDS ? 2 : R> twr_regs;
  if (!TWLDS) {
#pragma unroll
    for (int k1 = 1; k1 < R; ++k1) twr_regs.w[(TWLDS ? 1 : k1) - 1] = lds_tw1[(k1 - 1) * 64 + lane];
  }
  const TwLds twr_lds{lds_tw1, lane};
  const auto& twr = td_pick<TWLDS>(twr_lds, twr_regs);

  // samples of z_t: three hop-blocks carried from frame to frame plus the new one, fetched one frame ahead
  v2f xq[NB][QU], xn[QU];
#pragma unroll
  for (int q = 0; q < NB; ++q)
    td_load_block<R, OV>(zrow, tailrow, a.L, a.T, c, t_begin, t_end, t_begin + q, lane, a.pad_mode, xq[q]);
  td_load_block<R, OV>(zrow, tailrow, a.L, a.T, c, t_begin, t_end, t_begin + NB, lane, a.pad_mode, xn);

  // The overlap-add envelope is periodic in the hop wherever all n_fft / hop frames that cover a sample exist (hop-blocks NB .. T-1:
  // the same summands in the same order, plan_impl.h): one block of its reciprocal is kept in registers instead of being loaded for
  // every frame (the evaluating variant has no registers to spare and keeps loading)
  constexpr bool ENVREG = !EVAL && SPECINV_TD_ENVREG;
  v2f envc[ENVREG ? QU : 1];
  v2f envr[(ENVREG && SPECINV_IEEE) ? QU : 1];     // (reference chain: the envelope block and its correctly rounded reciprocal)
  if (ENVREG) {
    const v2f* e0 = reinterpret_cast<const v2f*>(a.inv_env + (long long)(NB - PB) * HOP);
#pragma unroll
    for (int i = 0; i < QU; ++i) {
      envc[i] = e0[64u * i + ulane];
      if (SPECINV_IEEE) envr[SPECINV_IEEE ? i : 0] = env_rcp(envc[i]);
    }
  }
#if SPECINV_TD_STAMPS
  unsigned long long stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long stamp_prev = __builtin_amdgcn_s_memtime();
  const unsigned long long stamp_begin = stamp_prev;
#endif
  // the real-FFT twiddles of the pairs, W_N^(lane + 64 j): the plain launches have the registers to keep all of them (two waves
  // per SIMD leave 256 each), the others rebuild them from W_N^lane every frame
  constexpr bool WKREG = !EVAL && SPECINV_TD_MINWAVES == 2 && SPECINV_TD_WKREG;
  v2f wkr[WKREG ? H : 1];
  if (WKREG) {
#pragma unroll
    for (int j = 0; j < H; ++j) wkr[j] = pair_twiddle<R>(k.wn, j);
  }
  for (int t = t_begin; t < t_end; ++t) {
    TD_STAMP(5);                       // (loop overhead / nothing on the first pass)
    asm volatile("" ::: "memory");     // (window / twiddle reads stay inside the loop: hoisted they pin ~80 VGPRs)
    v2f wn = k.wn;
    asm volatile("" : "+v"(wn));
    const long long fi = (long long)b * a.T + t;
    v4f mm[H / 2];
    v4f pp[EARLY ? H : 1];
    v2f pmid = v2f{0.0f, 0.0f};
    float mmid = 0.0f;
    __builtin_amdgcn_s_setprio(1);
    {
      const v4f* min_ = a.m_pairs + ((SPECINV_TD_ABLATE & 1) ? (long long)(fi & 255) : fi) * (H / 2 * 64);
#pragma unroll
      for (int j = 0; j < H / 2; ++j) mm[j] = ld_stream(&min_[j * 64u + ulane]);
      if (lane == 0) {
        mmid = a.m_mid[fi];
        if (EARLY) pmid = a.Pmid_in[fi];
      }
    }
#define SPECINV_TD_C0_LOADS()                                                              \
    if (EARLY) {                                                                           \
      const v4f* pin_ = a.P_in + fi * (H * 64);                                            \
      _Pragma("unroll") for (int j = 0; j < H; ++j) pp[j] = ld_stream(&pin_[j * 64u + ulane]); \
    }
    if (!EVAL) SPECINV_TD_C0_LOADS();      // (in flight during the forward FFT; the evaluating variant has no registers for
                                           // them before its first transform is done)

    v2f z[R];
    if (EVAL) {
      // |STFT(x_t)| against the target (methods.py:242): x_t's frame, transformed and dropped
      const float* xrow = a.x2_in + (long long)b * a.L;
#pragma unroll
      for (int qq = 0; qq < OV; ++qq) {
        v2f q[QU];
        td_load_block<R, OV>(xrow, tailrow, a.L, a.T, c, t_begin, t_end, t + qq, lane, a.pad_mode, q);
#pragma unroll
        for (int i = 0; i < QU; ++i) z[qq * QU + i] = q[i] * lds_win[64 * (qq * QU + i) + lane];
      }
      fft_forward_t<R>(z, k, twr, tr);
      v2f rc[H];
#pragma unroll
      for (int m = H; m < R; ++m) {
        const v2f got = shfl2(z[m], k.partner);
        const v2f own = z[(m + 1) % R];
        rc[m - H] = v2f{lane == 0 ? own.x : got.x, lane == 0 ? own.y : got.y};
      }
      // a frame's 2 H + 1 terms per lane are summed in float32 (relative error 1e-7: the metric is compared to 1e-5), the
      // frames in float64 - this kernel is bound by its vector instructions, float64 ones cost several each
      float fd = 0.0f, fo = 0.0f;
#pragma unroll
      for (int j = 0; j < H; ++j) {
        const v2f wk = pair_twiddle<R>(wn, j);
        v2f xk, xm;
        td_split<R>(z[j], rc[R - 1 - j - H], wk, half_scale, xk, xm);
        const float mk = (j & 1) ? mm[j / 2].z : mm[j / 2].x;
        const float mq = (j & 1) ? mm[j / 2].w : mm[j / 2].y;
        const float ok = fast_abs(xk), om = fast_abs(xm);
        const float dk = ok - mk, dm = om - mq;
        fd = fmaf(dk, dk, fmaf(dm, dm, fd));
        fo = fmaf(ok, ok, fmaf(om, om, fo));
      }
      if (lane == 0) {
        const float o = fast_abs(z[H] * v2f{a.fwd_scale, -a.fwd_scale});
        const float d = o - mmid;
        fd = fmaf(d, d, fd);
        fo = fmaf(o, o, fo);
      }
      sd += (double)fd;
      so += (double)fo;
      asm volatile("" ::: "memory");       // (keeps the scheduler from hoisting the loads into the transform above)
      __builtin_amdgcn_sched_barrier(0);
      SPECINV_TD_C0_LOADS();
    }
#undef SPECINV_TD_C0_LOADS

    // ---- analysis of z_t's frame; its oldest hop-block is the one this frame's output block needs
    v2f zold[QU];
#pragma unroll
    for (int i = 0; i < QU; ++i) {
      zold[i] = xq[0][i];
#pragma unroll
      for (int q = 0; q < NB; ++q) z[q * QU + i] = xq[q][i] * lds_win[64 * (q * QU + i) + lane];
      z[NB * QU + i] = xn[i] * lds_win[64 * (NB * QU + i) + lane];
#pragma unroll
      for (int q = 0; q + 1 < NB; ++q) xq[q][i] = xq[q + 1][i];
      xq[NB - 1][i] = xn[i];
    }
    if (t + 1 < t_end)
      td_load_block<R, OV>(zrow, tailrow, a.L, a.T, c, t_begin, t_end, (SPECINV_TD_ABLATE & 4) ? 8 + (t & 3) : t + OV, lane, a.pad_mode, xn);
    __builtin_amdgcn_s_setprio(0);
    TD_STAMP(0);

    fft_forward_t<R>(z, k, twr, tr);
    TD_STAMP(1);

    v2f rc[H];
#pragma unroll
    for (int m = H; m < R; ++m) {
      const v2f got = shfl2(z[m], k.partner);
      const v2f own = z[(m + 1) % R];              // lane 0 is its own partner, shifted by one register
      rc[m - H] = v2f{lane == 0 ? own.x : got.x, lane == 0 ? own.y : got.y};
    }

    // ---- per pair: split -> (+ c0 term) -> projection -> fold back
    v2f back[H];
#if SPECINV_IEEE && !SPECINV_REFBREADTH
#pragma unroll
    for (int j = 0; j < H; ++j) {
      const v2f wk = WKREG ? wkr[WKREG ? j : 0] : pair_twiddle<R>(wn, j);
      v2f sk, sm;
      td_split_raw<R>(z[j], rc[R - 1 - j - H], wk, sk, sm);
      if (EARLY) {
        sk = __builtin_elementwise_fma(v2f{pp[j].x, pp[j].y}, v2f{tds_raw, tds_raw}, sk);
        sm = __builtin_elementwise_fma(v2f{pp[j].z, pp[j].w}, v2f{tds_raw, tds_raw}, sm);
      }
      const v2f rr = ref_rcp_abs2(v2f{ref_norm2(sk, 4.0f * kRefFloor), ref_norm2(sm, 4.0f * kRefFloor)}, 2e-16f);
      const v2f mp = (j & 1) ? v2f{mm[j / 2].z, mm[j / 2].w} : v2f{mm[j / 2].x, mm[j / 2].y};
      v2f ak = scale_lo(scale_lo(sk, mp), rr);
      v2f am = scale_hi(scale_hi(sm, mp), rr);
      if (j == 0 && lane == 0) {   // bins 0 and M: irfft ignores their imaginary parts
        ak.y = 0.0f;
        am.y = 0.0f;
      }
      const v2f e2i = add_conj(ak, am);
      const v2f o2i = cmulc(sub_conj(ak, am), wk);
      z[j] = add_i(e2i, o2i);
      back[j] = conj_sub_i(e2i, o2i);
    }
#elif SPECINV_IEEE
    // The reference's operation order (ref_rcp_abs2, fast_core.h), written breadth-first over the frame's H pairs: every step of
    // the chain of dependent packed operations is issued for all pairs before the next one, so that no instruction waits on the
    // one just before it (the compiler pads such pairs with s_nop: 44 in the frame loop instead of 106).
    {
      v2f sk[H], sm[H], tt[H], yy[H], hh[H];
#pragma unroll
      for (int j = 0; j < H; ++j) {
        const v2f wk = WKREG ? wkr[WKREG ? j : 0] : pair_twiddle<R>(wn, j);
        td_split_raw<R>(z[j], rc[R - 1 - j - H], wk, sk[j], sm[j]);
        if (EARLY) {
          sk[j] = __builtin_elementwise_fma(v2f{pp[j].x, pp[j].y}, v2f{tds_raw, tds_raw}, sk[j]);
          sm[j] = __builtin_elementwise_fma(v2f{pp[j].z, pp[j].w}, v2f{tds_raw, tds_raw}, sm[j]);
        }
        tt[j] = v2f{ref_norm2(sk[j], 4.0f * kRefFloor), ref_norm2(sm[j], 4.0f * kRefFloor)};
      }
#pragma unroll
      for (int j = 0; j < H; ++j) yy[j] = v2f{__builtin_amdgcn_rsqf(tt[j].x), __builtin_amdgcn_rsqf(tt[j].y)};
#if SPECINV_REFCHAIN == 1
#pragma unroll
      for (int j = 0; j < H; ++j) hh[j] = tt[j] * yy[j];
#pragma unroll
      for (int j = 0; j < H; ++j) tt[j] = __builtin_elementwise_fma(-hh[j], hh[j], tt[j]);        // residual t - h^2
#pragma unroll
      for (int j = 0; j < H; ++j) hh[j] = __builtin_elementwise_fma(tt[j], yy[j] * 0.5f, hh[j]);  // RN(sqrt t)
#pragma unroll
      for (int j = 0; j < H; ++j) hh[j] = hh[j] + 2e-16f;                                           // (+ 1e-16 at the true scale)
#pragma unroll
      for (int j = 0; j < H; ++j) tt[j] = __builtin_elementwise_fma(-hh[j], yy[j], v2f{1.0f, 1.0f});
#pragma unroll
      for (int j = 0; j < H; ++j) yy[j] = __builtin_elementwise_fma(tt[j], yy[j], yy[j]);          // RN(1 / (|s| + 1e-16))
#else
      // one Newton step on y ~ t^-1/2: the correctly rounded 1 / |s| in all but ~1e-6 of the cases (ONE rounding of the exact
      // value where the reference rounds |s| and then its reciprocal); the guard 1e-16 only matters below |s| = 3e-9
#pragma unroll
      for (int j = 0; j < H; ++j) hh[j] = tt[j] * yy[j];
#pragma unroll
      for (int j = 0; j < H; ++j) tt[j] = __builtin_elementwise_fma(-hh[j], yy[j], v2f{1.0f, 1.0f});
#pragma unroll
      for (int j = 0; j < H; ++j) yy[j] = __builtin_elementwise_fma(yy[j] * 0.5f, tt[j], yy[j]);
#endif
#pragma unroll
      for (int j = 0; j < H; ++j) {
        const v2f mp = (j & 1) ? v2f{mm[j / 2].z, mm[j / 2].w} : v2f{mm[j / 2].x, mm[j / 2].y};
        sk[j] = scale_lo(sk[j], mp);
        sm[j] = scale_hi(sm[j], mp);
      }
#pragma unroll
      for (int j = 0; j < H; ++j) {
        const v2f wk = WKREG ? wkr[WKREG ? j : 0] : pair_twiddle<R>(wn, j);
        v2f ak = scale_lo(sk[j], yy[j]);
        v2f am = scale_hi(sm[j], yy[j]);
        if (j == 0 && lane == 0) {   // bins 0 and M: irfft ignores their imaginary parts
          ak.y = 0.0f;
          am.y = 0.0f;
        }
        const v2f e2i = add_conj(ak, am);
        const v2f o2i = cmulc(sub_conj(ak, am), wk);
        z[j] = add_i(e2i, o2i);
        back[j] = conj_sub_i(e2i, o2i);
      }
    }
#else
#pragma unroll
    for (int j = 0; j < H; ++j) {
      const v2f wk = WKREG ? wkr[WKREG ? j : 0] : pair_twiddle<R>(wn, j);
      v2f sk, sm;
      if (!SPECINV_RSQ) {
        td_split<R>(z[j], rc[R - 1 - j - H], wk, half_scale, sk, sm);
        if (EARLY) {
          sk = v2f{fmaf(a.tds, pp[j].x, sk.x), fmaf(a.tds, pp[j].y, sk.y)};
          sm = v2f{fmaf(a.tds, pp[j].z, sm.x), fmaf(a.tds, pp[j].w, sm.y)};
        }
      } else {
        // nothing but the projection reads the bins, and it divides by their magnitude: the 1/2 fwd_scale of the split is left
        // out (late launches) or moved onto the c0 term's factor (early launches: S / s = raw + (tds / s) c0, one packed
        // multiply-add per bin).  With the reference's operation chain (SPECINV_IEEE) this is exact, not approximate: the
        // kernel is only taken for fwd_scale = 1, the factor 2 scales every intermediate result without a rounding, and the
        // guard / floor constants are doubled / quadrupled with it.
        td_split_raw<R>(z[j], rc[R - 1 - j - H], wk, sk, sm);
        if (EARLY) {
          sk = __builtin_elementwise_fma(v2f{pp[j].x, pp[j].y}, v2f{tds_raw, tds_raw}, sk);
          sm = __builtin_elementwise_fma(v2f{pp[j].z, pp[j].w}, v2f{tds_raw, tds_raw}, sm);
        }
      }
      const float mk = (j & 1) ? mm[j / 2].z : mm[j / 2].x;
      const float mq = (j & 1) ? mm[j / 2].w : mm[j / 2].y;
#if SPECINV_RSQ
      // the projection's factors (proj_rsq, fast_core.h) of the pair's two bins, formed and applied as packed operations
      const v2f inv = v2f{proj_rsq(sk), proj_rsq(sm)};
      const v2f mi = v2f{mk, mq} * inv;                       // (the inverse scale rides on the synthesis window)
      v2f ak = scale_lo(sk, mi);
      v2f am = scale_hi(sm, mi);
#else
      const float ik = fast_rcp(fast_abs(sk) + 1e-16f) * a.inv_scale, iq = fast_rcp(fast_abs(sm) + 1e-16f) * a.inv_scale;
      v2f ak = v2f{(sk.x * mk) * ik, (sk.y * mk) * ik};
      v2f am = v2f{(sm.x * mq) * iq, (sm.y * mq) * iq};
#endif
      if (j == 0 && lane == 0) {   // bins 0 and M: irfft ignores their imaginary parts
        ak.y = 0.0f;
        am.y = 0.0f;
      }
      const v2f e2i = add_conj(ak, am);
      const v2f o2i = cmulc(sub_conj(ak, am), wk);
      z[j] = add_i(e2i, o2i);
      back[j] = conj_sub_i(e2i, o2i);
    }
#endif
    v2f zmid;
    {
      v2f smid = z[H] * v2f{a.fwd_scale, -a.fwd_scale};
      if (EARLY) smid = v2f{fmaf(a.tds, pmid.x, smid.x), fmaf(a.tds, pmid.y, smid.y)};
#if SPECINV_IEEE
      const v2f am = (smid * mmid) * ref_rcp_abs(ref_norm2(smid));
#elif SPECINV_RSQ
      const v2f am = smid * (mmid * proj_rsq(smid));
#else
      const float inv = fast_rcp(fast_abs(smid) + 1e-16f) * a.inv_scale;
      const v2f am = v2f{(smid.x * mmid) * inv, (smid.y * mmid) * inv};
#endif
      zmid = am * v2f{2.0f, -2.0f};
    }
#pragma unroll
    for (int m = H; m < R; ++m) {
      const v2f got = shfl2(back[R - 1 - m], k.partner);
      const v2f l0 = (m == H) ? zmid : back[(R - m) % H];
      z[m] = v2f{lane == 0 ? l0.x : got.x, lane == 0 ? l0.y : got.y};
    }
    TD_STAMP(2);

    fft_inverse_t<R>(z, k, twr, tr);
    TD_STAMP(3);

    // ---- synthesis window, register overlap-add, one finished hop-block of x_{t+1} and of z_{t+1} out
#pragma unroll
    for (int u = 0; u < R; ++u) z[u] = z[u] * lds_wins[64 * u + lane];
    if (t >= PB) {
      const long long o0 = (long long)(t - PB) * HOP;
      const v2f* envp = reinterpret_cast<const v2f*>(a.inv_env + o0);   // uniform
      v2f* xo = reinterpret_cast<v2f*>(xorow + o0);
      v2f* zo = reinterpret_cast<v2f*>(zorow + o0);
      // The envelope block: the register copy wherever it is valid.  The first frames of an item load theirs - in an arm of
      // its own that also waits for it (the empty asm reads the registers): left to a select per value the compiler puts each
      // load behind a branch and an unconditional s_waitcnt vmcnt(0) after the join, which on gfx950 also waits for the stores
      // just issued - three store round trips per frame in the steady state.
      v2f ev[QU], er[QU];
      if (!ENVREG) {
#pragma unroll
        for (int i = 0; i < QU; ++i) {
          ev[i] = envp[64u * i + ulane];
          er[i] = env_rcp(ev[i]);
        }
      } else if (t >= NB) {
#pragma unroll
        for (int i = 0; i < QU; ++i) {
          ev[i] = envc[ENVREG ? i : 0];
          er[i] = envr[(ENVREG && SPECINV_IEEE) ? i : 0];
        }
      } else {
#pragma unroll
        for (int i = 0; i < QU; ++i) ev[i] = envp[64u * i + ulane];
#pragma unroll
        for (int i = 0; i < QU; ++i) asm volatile("" : "+v"(ev[i]));
#pragma unroll
        for (int i = 0; i < QU; ++i) er[i] = env_rcp(ev[i]);
      }
#pragma unroll
      for (int i = 0; i < QU; ++i) {
        const v2f xv = env_apply_r(acc[i] + z[i], ev[i], er[i]);
        const v2f zv = v2f{fmaf(nlr, zold[i].x, xv.x), fmaf(nlr, zold[i].y, xv.y)};
        if (!(SPECINV_TD_ABLATE & 2) || zv.x == 1.2345e30f) {
          if (write_x) xo[64u * i + ulane] = xv;
          zo[64u * i + ulane] = zv;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < QU; ++i) {
#pragma unroll
      for (int q = 0; q + 1 < NB; ++q) acc[q * QU + i] = acc[(q + 1) * QU + i] + z[(q + 1) * QU + i];
      acc[(NB - 1) * QU + i] = z[NB * QU + i];
    }
    TD_STAMP(4);
  }
#if SPECINV_TD_STAMPS
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) a.stamps[(long long)w * 8 + i] = stamp_sum[i];
    a.stamps[(long long)w * 8 + 6] = (unsigned long long)(t_end - t_begin);
    // where and when the wave ran: HW_ID (wave / SIMD / CU / SH / SE) with the XCC id above it, begin and end in the low / high
    // halves of one word (differences to the launch's earliest begin fit 32 bits)
    const unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));    // HW_REG_HW_ID, 32 bits
    const unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));  // HW_REG_XCC_ID
    a.stamps[(long long)a.n_waves * 8 + 2 * (long long)w] = ((unsigned long long)xcc << 32) | hw;
    a.stamps[(long long)a.n_waves * 8 + 2 * (long long)w + 1] = stamp_begin;
    a.stamps[(long long)a.n_waves * 10 + (long long)w] = __builtin_amdgcn_s_memtime();
  }
#endif
  if (t_end == a.T) {
    // the chunk that holds the last frame also finishes hop-blocks T .. T + PB - 2 (the frames that reach them are done);
    // xq[q] is z_t's block T + q by now
#pragma unroll
    for (int q = 0; q < PB - 1; ++q) {
      const long long o0 = (long long)(a.T + q - PB) * HOP;
      const v2f* envp = reinterpret_cast<const v2f*>(a.inv_env + o0);
      v2f* xo = reinterpret_cast<v2f*>(xorow + o0);
      v2f* zo = reinterpret_cast<v2f*>(zorow + o0);
#pragma unroll
      for (int i = 0; i < QU; ++i) {
        const v2f xv = env_apply(acc[q * QU + i], envp[64u * i + ulane]);
        if (write_x) xo[64u * i + ulane] = xv;
        zo[64u * i + ulane] = v2f{fmaf(nlr, xq[q][i].x, xv.x), fmaf(nlr, xq[q][i].y, xv.y)};
      }
    }
  } else {
    // what this chunk's last NB frames contribute to the next chunk's first NB hop-blocks (of x and of z alike)
    v2f* tl = reinterpret_cast<v2f*>(a.xtail_out + ((long long)b * a.nchunks + c) * NB * HOP);
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      const v2f* envp = reinterpret_cast<const v2f*>(a.inv_env + (long long)(t_end + q - PB) * HOP);
#pragma unroll
      for (int i = 0; i < QU; ++i) tl[(q * QU + i) * 64u + ulane] = env_apply(acc[q * QU + i], envp[64u * i + ulane]);
    }
  }
  if (EVAL) {
    const double d = wave_sum(sd), o = wave_sum(so);
    if (lane == 0) {
      a.partials[2 * (long long)w] = d;
      a.partials[2 * (long long)w + 1] = o;
    }
  }
}

// the headline shapes (hop = n_fft/4 at n_fft 1024 / 2048) launch 8-wave workgroups, one per CU, like k_fused4
template <int R, bool EARLY, bool EVAL>
__global__ __launch_bounds__((SPECINV_TD_MINWAVES == 3 || (SPECINV_R8_W3 && R == 8)) ? 768 : 64 * SPECINV_WGW,
                            (SPECINV_R8_W3 && R == 8) ? 3 : SPECINV_TD_MINWAVES) void k_fused4_td(FastArgs a) {
  fused_td_body<R, 4, EARLY, EVAL>(a);
}
// every other fused shape: 4-wave workgroups like k_fused<R, OV>
template <int R, int OV, bool EARLY, bool EVAL>
__global__ __launch_bounds__(256, R >= 32 ? 1 : (SPECINV_R8_W3 && R == 8) ? 3 : SPECINV_MINWAVES) void k_fused_td(FastArgs a) {
  fused_td_body<R, OV, EARLY, EVAL>(a);
}

// ---- the evaluation of x_t as a kernel of its own ----------------------------------------------------------------------------
// |STFT(x_t)| against the target (methods.py:180-182, 242) does not depend on iteration t's update, which reads z_t.  Inside
// the iteration kernel the second transform runs in a 256-register wave, two to a SIMD, its frame loaded block by block through
// the seam logic four times over: + 0.16 ms per launch at BASELINE C2.  On its own - launched behind the plain iteration kernel
// on the same stream - it is a forward transform and a sum with the sample window carried from frame to frame like the
// iteration's: 0.12 ms.  A wave walks one of kEvalSub pieces of a chunk of the iteration's chunking (whose seams the block
// loader resolves).  Same operations in the same order as the EVAL block of fused_td_body: the same sums, bit for bit.
#ifndef SPECINV_EVAL_TWREGS
#define SPECINV_EVAL_TWREGS 1
#endif
constexpr int kEvalSub = kEvalPieces;
template <int R, int OV>
__global__ __launch_bounds__(256, SPECINV_EVAL_WAVES) void k_eval_td(FastArgs a) {
  using G = Geo<R>;
  using O = Ovl<R, OV>;
  constexpr int H = G::H, M = G::M, QU = O::QU, HOP = O::HOP, NB = O::NB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  v2f* lds_win = reinterpret_cast<v2f*>(smem);
  v2f* lds_tw1 = lds_win + M;
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  v2f* tr = lds_tw1 + (R - 1) * 64 + wib * G::TR;
  for (int i = threadIdx.x; i < M; i += blockDim.x) lds_win[i] = v2f{a.window[2 * i], a.window[2 * i + 1]};
  for (int i = threadIdx.x; i < (R - 1) * 64; i += blockDim.x) {
    const int k1 = i / 64 + 1, l = i & 63;
    lds_tw1[i] = unit(2.0f * (float)((l * k1) % M) / (float)M);
  }
  __syncthreads();
  const int w = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wib);
  if (w >= a.n_waves * kEvalSub) return;
  const LaneConst<R> k = lane_consts<R>();
  const int lane = k.lane;
  const unsigned ulane = (unsigned)lane;
  const int cw = w / kEvalSub, sub = w - cw * kEvalSub;
  const int b = cw / a.nchunks, c = cw - b * a.nchunks;
  const int t_begin = chunk_begin(c, a.T, a.nchunks, a.skew), t_end = chunk_begin(c + 1, a.T, a.nchunks, a.skew);
  const int s_begin = t_begin + (sub * (t_end - t_begin)) / kEvalSub, s_end = t_begin + ((sub + 1) * (t_end - t_begin)) / kEvalSub;
  const float* xrow = a.x2_in + (long long)b * a.L;
  const float* tailrow = a.xtail_in + (long long)b * a.nchunks * NB * HOP;
  const float half_scale = 0.5f * a.fwd_scale;
#if SPECINV_EVAL_TWREGS
  TwRegs<R> twr;
#pragma unroll
  for (int k1 = 1; k1 < R; ++k1) twr.w[k1 - 1] = lds_tw1[(k1 - 1) * 64 + lane];
#else
  const TwLds twr{lds_tw1, lane};
#endif
  double sd = 0.0, so = 0.0;
  v2f xq[NB][QU], xn[QU];
#pragma unroll
  for (int q = 0; q < NB; ++q)
    td_load_block<R, OV>(xrow, tailrow, a.L, a.T, c, t_begin, t_end, s_begin + q, lane, a.pad_mode, xq[q]);
  td_load_block<R, OV>(xrow, tailrow, a.L, a.T, c, t_begin, t_end, s_begin + NB, lane, a.pad_mode, xn);
  for (int t = s_begin; t < s_end; ++t) {
    asm volatile("" ::: "memory");
    v2f wn = k.wn;
    asm volatile("" : "+v"(wn));
    const long long fi = (long long)b * a.T + t;
    v4f mm[H / 2];
    float mmid = 0.0f;
    {
      const v4f* min_ = a.m_pairs + fi * (H / 2 * 64);
#pragma unroll
      for (int j = 0; j < H / 2; ++j) mm[j] = ld_stream(&min_[j * 64u + ulane]);
      if (lane == 0) mmid = a.m_mid[fi];
    }
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
    if (t + 1 < s_end) td_load_block<R, OV>(xrow, tailrow, a.L, a.T, c, t_begin, t_end, t + OV, lane, a.pad_mode, xn);
    fft_forward_t<R>(z, k, twr, tr);
    v2f rc[H];
#pragma unroll
    for (int m = H; m < R; ++m) {
      const v2f got = shfl2(z[m], k.partner);
      const v2f own = z[(m + 1) % R];
      rc[m - H] = v2f{lane == 0 ? own.x : got.x, lane == 0 ? own.y : got.y};
    }
    float fd = 0.0f, fo = 0.0f;
#pragma unroll
    for (int j = 0; j < H; ++j) {
      const v2f wk = pair_twiddle<R>(wn, j);
      v2f xk, xm;
      td_split<R>(z[j], rc[R - 1 - j - H], wk, half_scale, xk, xm);
      const float mk = (j & 1) ? mm[j / 2].z : mm[j / 2].x;
      const float mq = (j & 1) ? mm[j / 2].w : mm[j / 2].y;
      const float ok = fast_abs(xk), om = fast_abs(xm);
      const float dk = ok - mk, dm = om - mq;
      fd = fmaf(dk, dk, fmaf(dm, dm, fd));
      fo = fmaf(ok, ok, fmaf(om, om, fo));
    }
    if (lane == 0) {
      const float o = fast_abs(z[H] * v2f{a.fwd_scale, -a.fwd_scale});
      const float d = o - mmid;
      fd = fmaf(d, d, fd);
      fo = fmaf(o, o, fo);
    }
    sd += (double)fd;
    so += (double)fo;
  }
  const double d = wave_sum(sd), o = wave_sum(so);
  if (lane == 0) {
    a.partials[2 * (long long)w] = d;
    a.partials[2 * (long long)w + 1] = o;
  }
}

}  // namespace SI_FAST_NS (fast, or fast_approx in the approximate-projection units)
}  // namespace specinv
