// The fused iteration kernels on the spectral state: k_fused4 (hop = n_fft/4 at n_fft 1024 / 2048, the tuned copy), k_fused<R, OV> (every fused shape) and the fused initial ISTFT.  Compiled in tu_fused_*.hip.
#pragma once
#include "fast_core.h"

#ifndef SPECINV_K4_STAMPS
#define SPECINV_K4_STAMPS 0
#endif

namespace specinv {
namespace SI_FAST_NS {

template <int R, int MODE, bool EVAL>
__global__ __launch_bounds__((SPECINV_R8_W3 && R == 8) ? 768 : 64 * SPECINV_WGW, (SPECINV_R8_W3 && R == 8) ? 3 : SPECINV_MINWAVES) void k_fused4(FastArgs a) {
  using G = Geo<R>;
  constexpr int H = G::H, QU = G::QU, M = G::M, HOP = G::HOP;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  v2f* lds_win = reinterpret_cast<v2f*>(smem);
  v2f* lds_tw1 = lds_win + M;
  // wave-uniform values are forced into SGPRs: every global address below is then
  // "scalar base + 32-bit lane offset" instead of one 64-bit VGPR pointer per access
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  v2f* tr = lds_tw1 + (R - 1) * 64 + wib * G::TR;

  for (int i = threadIdx.x; i < M; i += blockDim.x) lds_win[i] = v2f{a.window[2 * i], a.window[2 * i + 1]};
  for (int i = threadIdx.x; i < (R - 1) * 64; i += blockDim.x) {
    const int k1 = i / 64 + 1, l = i & 63;
    lds_tw1[i] = unit(2.0f * (float)((l * k1) % M) / (float)M);   // W_M^(l*k1)
  }
  __syncthreads();

  const int w = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + wib);   // (scalar: the frame loop and its padding tests branch on the scalar unit; left to the compiler the work-group size may arrive in a vector register and make all of it per-lane)
  if (w >= a.n_waves) return;
#if SPECINV_K4_STAMPS
  const unsigned long long k4_begin = __builtin_amdgcn_s_memtime();
#endif
  const LaneConst<R> k = lane_consts<R>();
  const int lane = k.lane;
  const unsigned ulane = (unsigned)lane;
  // Even chunks: wave w walks chunk w mod nchunks of item w / nchunks.  With three waves per SIMD (12-wave workgroups, one per
  // CU) the hardware slot of a wave is its index in the workgroup / 4, and the arbiter serves the oldest wave first: at BASELINE
  // C4 the slot-0 waves finished after 193 k ticks, slot 1 after 226 k, slot 2 after 271 k (1024 each).  The plan then skews the
  // chunks in threes (chunk_begin) and the three waves of a SIMD walk one triple: the oldest the longest.
  int b = w / a.nchunks, c = w - b * a.nchunks;
  if (a.skew >= 0x10000) {
    const int cg = 3 * ((int)blockIdx.x * 4 + (wib & 3)) + (wib >> 2);
    b = cg / a.nchunks;
    c = cg - b * a.nchunks;
  }
  const int t_begin = chunk_begin(c, a.T, a.nchunks, a.skew);
  const int t_end = chunk_begin(c + 1, a.T, a.nchunks, a.skew);
  const int t_start = t_begin;   // no halo: the previous chunk's share of the first three hop-blocks comes via xtail
  const float* xrow = a.x_in + (long long)b * a.L;
  const float* tailrow = a.xtail_in + (long long)b * a.nchunks * 3 * HOP;
  float* orow = a.x_out + (long long)b * a.L;
  const float half_scale = 0.5f * a.fwd_scale;

  v2f acc[3 * QU];
#pragma unroll
  for (int i = 0; i < 3 * QU; ++i) acc[i] = v2f{0.0f, 0.0f};
  double sd = 0.0, so = 0.0;
  // one block of the envelope reciprocal, periodic in the hop from hop-block 3 on (kernels_fast_td.h), kept in registers
  v2f envc[SPECINV_K4_ENVREG ? QU : 1];
  v2f envr[(SPECINV_K4_ENVREG && SPECINV_IEEE) ? QU : 1];   // (reference chain: the envelope and its correctly rounded reciprocal)
  if (SPECINV_K4_ENVREG) {
    const v2f* e0 = reinterpret_cast<const v2f*>(a.inv_env + (long long)HOP);
#pragma unroll
    for (int i = 0; i < QU; ++i) {
      envc[i] = e0[64u * i + ulane];
      if (SPECINV_IEEE) envr[SPECINV_IEEE ? i : 0] = env_rcp(envc[i]);
    }
  }
#if SPECINV_TW_REGS
  TwRegs<R> twr;
#pragma unroll
  for (int k1 = 1; k1 < R; ++k1) twr.w[k1 - 1] = lds_tw1[(k1 - 1) * 64 + lane];
#endif

  // raw samples of the current frame: three hop-blocks carried from frame to frame plus the
  // new one, which is fetched one frame ahead so that its latency hides behind a whole frame
#if SPECINV_XPREF == 2
  v2f znext[R];
#pragma unroll
  for (int qq = 0; qq < 4; ++qq) {
    v2f q[QU];
    load_block4<R>(xrow, tailrow, a.L, a.T, c, t_begin, t_end, t_start + qq, lane, a.pad_mode, q);
#pragma unroll
    for (int i = 0; i < QU; ++i) znext[qq * QU + i] = q[i];
  }
#elif SPECINV_XPREF == 1
  v2f xq[3][QU], xn[QU];
  load_block4<R>(xrow, tailrow, a.L, a.T, c, t_begin, t_end, t_start, lane, a.pad_mode, xq[0]);
  load_block4<R>(xrow, tailrow, a.L, a.T, c, t_begin, t_end, t_start + 1, lane, a.pad_mode, xq[1]);
  load_block4<R>(xrow, tailrow, a.L, a.T, c, t_begin, t_end, t_start + 2, lane, a.pad_mode, xq[2]);
  load_block4<R>(xrow, tailrow, a.L, a.T, c, t_begin, t_end, t_start + 3, lane, a.pad_mode, xn);
#endif

  // state of frame `FI`: uniform bases (SGPR) + unsigned 32-bit lane offsets -> "saddr + voffset" addressing
#define SPECINV_STATE_LOADS4(FI)                                                            \
  do {                                                                                     \
    const long long fl_ = (FI);                                                            \
    const v4f* pin_ = a.P_in + fl_ * (H * 64);                                             \
    const v4f* min_ = a.m_pairs + fl_ * (H / 2 * 64);                                      \
    _Pragma("unroll") for (int j = 0; j < H; ++j) pp[j] = ld_stream(&pin_[j * 64u + ulane]); \
    _Pragma("unroll") for (int j = 0; j < H / 2; ++j) mm[j] = ld_stream(&min_[j * 64u + ulane]); \
    if (lane == 0) {                                                                       \
      pmid = a.Pmid_in[fl_];                                                               \
      mmid = a.m_mid[fl_];                                                                 \
    }                                                                                      \
  } while (0)
#if SPECINV_PLATE == 2
  v4f pp[H], mm[H / 2];
  v2f pmid = v2f{0.0f, 0.0f}, umid = v2f{0.0f, 0.0f};
  float mmid = 0.0f;
  SPECINV_STATE_LOADS4((long long)b * a.T + t_start);
#endif

  for (int t = t_start; t < t_end; ++t) {
    // Keep the loop-invariant table reads (window, twiddles) and products inside the loop: hoisted out
    // of it they pin ~80 VGPRs for the whole kernel and cost a wave of occupancy.
    asm volatile("" ::: "memory");
    v2f wn = k.wn;
    asm volatile("" : "+v"(wn));
    constexpr bool live = true;
    const long long fi = (long long)b * a.T + t;
    v4f* pout = a.P_out + fi * (H * 64);
    const bool keep_xu = MODE == MODE_ADMM && a.U_out != nullptr;   // (uniform: a kernel argument)
#if SPECINV_PLATE != 2
    v4f pp[H], mm[H / 2];
    v2f pmid = v2f{0.0f, 0.0f}, umid = v2f{0.0f, 0.0f};
    float mmid = 0.0f;
#endif
#if SPECINV_PLATE == 0
#if SPECINV_PRIO
    __builtin_amdgcn_s_setprio(SPECINV_PRIO & 3);
#endif
    SPECINV_STATE_LOADS4(fi);   // early: the loads fly during the forward FFT
#endif

    // ---- analysis: windowed frame -> registers
    v2f z[R];
#if SPECINV_XPREF == 2
#pragma unroll
    for (int u = 0; u < R; ++u) z[u] = znext[u] * lds_win[64 * u + lane];
#elif SPECINV_XPREF == 1
    // slide the sample window and prefetch the next hop-block
#pragma unroll
    for (int i = 0; i < QU; ++i) {
      z[i] = xq[0][i] * lds_win[64 * i + lane];
      z[QU + i] = xq[1][i] * lds_win[64 * (QU + i) + lane];
      z[2 * QU + i] = xq[2][i] * lds_win[64 * (2 * QU + i) + lane];
      z[3 * QU + i] = xn[i] * lds_win[64 * (3 * QU + i) + lane];
      xq[0][i] = xq[1][i];
      xq[1][i] = xq[2][i];
      xq[2][i] = xn[i];
    }
    if (t + 1 < t_end) load_block4<R>(xrow, tailrow, a.L, a.T, c, t_begin, t_end, t + 4, lane, a.pad_mode, xn);
#if SPECINV_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
#else
    {
      v2f q[QU];
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        load_block4<R>(xrow, tailrow, a.L, a.T, c, t_begin, t_end, t + qq, lane, a.pad_mode, q);
#pragma unroll
        for (int i = 0; i < QU; ++i) z[qq * QU + i] = q[i];
      }
#pragma unroll
      for (int u = 0; u < R; ++u) z[u] = z[u] * lds_win[64 * u + lane];
    }
#endif

#if SPECINV_ABLATE & 4
#elif SPECINV_TW_REGS
    fft_forward_t<R>(z, k, twr, tr);
#else
    fft_forward<R>(z, k, lds_tw1, tr);
#endif
#if SPECINV_PLATE == 1
    SPECINV_STATE_LOADS4(fi);
#endif

    // ---- conjugate partners: upper half of lane (64 - r)
    v2f rc[H];   // rc[i] pairs with own register H-1-i ... see below: rc[m-H] = Z[M - (lane + 64*(R-1-m))]
    // (the lane-0 special case is patched AFTER the shuffle: selecting between two elements of
    // one register array before it makes the compiler index the array dynamically)
#pragma unroll
    for (int m = H; m < R; ++m) {
      const v2f got = shfl2(z[m], k.partner);
      const v2f own = z[(m + 1) % R];              // lane 0 is its own partner, shifted by one register
      rc[m - H] = v2f{lane == 0 ? own.x : got.x, lane == 0 ? own.y : got.y};
    }

    // ---- per pair: split -> update -> fold back
    v2f back[H];   // back[j] = Z''[M - k_j], to be returned to the partner lane
#pragma unroll
    for (int j = 0; j < H; ++j) {
      // W_N^(lane + 64 j) = W_N^lane * W_{2R}^j
      const v2f wk = pair_twiddle<R>(wn, j);
      const v2f zk = z[j], zm = rc[R - 1 - j - H];
      const v2f e2 = add_conj(zk, zm);
      const v2f dd = sub_conj(zk, zm);
      const v2f tw = cmul_mi(wk, dd);                 // W * (-i (Zk - conj Zm))
      v2f xk = (e2 + tw) * half_scale;
      v2f xm = (e2 - tw) * v2f{half_scale, -half_scale};   // conj(...)
      v2f pk = v2f{pp[j].x, pp[j].y}, pm = v2f{pp[j].z, pp[j].w};
      v2f uk = v2f{0.0f, 0.0f}, um = v2f{0.0f, 0.0f}, sk = v2f{0.0f, 0.0f}, sm = v2f{0.0f, 0.0f};
      const float mk = (j & 1) ? mm[j / 2].z : mm[j / 2].x;
      const float mq = (j & 1) ? mm[j / 2].w : mm[j / 2].y;
      v2f ak = update_bin<MODE, EVAL>(xk, pk, uk, sk, mk, a, live, sd, so);
      v2f am = update_bin<MODE, EVAL>(xm, pm, um, sm, mq, a, live, sd, so);
      if (live) {
        st_stream(&pout[j * 64u + ulane], v4f{pk.x, pk.y, pm.x, pm.y});
        if (keep_xu) {
          st_stream(&a.X_out[fi * (H * 64) + j * 64u + ulane], v4f{sk.x, sk.y, sm.x, sm.y});
          st_stream(&a.U_out[fi * (H * 64) + j * 64u + ulane], v4f{uk.x, uk.y, um.x, um.y});
        }
      }
      if (j == 0 && lane == 0) {   // bins 0 and M: irfft ignores their imaginary parts
        ak.y = 0.0f;
        am.y = 0.0f;
      }
      const v2f e2i = add_conj(ak, am);
      const v2f o2i = cmulc(sub_conj(ak, am), wk);
      z[j] = add_i(e2i, o2i);
      back[j] = conj_sub_i(e2i, o2i);
    }
    // ---- bin M/2 (lane 0): X = conj(Z), Z'' = 2 conj(X')
    v2f zmid;
    {
      v2f xmid = z[H] * v2f{a.fwd_scale, -a.fwd_scale};
      const bool live0 = live && lane == 0;
      v2f smid = v2f{0.0f, 0.0f};
      const v2f am = update_bin<MODE, EVAL>(xmid, pmid, umid, smid, mmid, a, live0, sd, so);
      if (live0) {
        a.Pmid_out[fi] = pmid;
        if (keep_xu) {
          a.Xmid_out[fi] = smid;
          a.Umid_out[fi] = umid;
        }
      }
      zmid = am * v2f{2.0f, -2.0f};
    }
    // ---- return the mirrored halves
#pragma unroll
    for (int m = H; m < R; ++m) {
      const v2f got = shfl2(back[R - 1 - m], k.partner);
      const v2f l0 = (m == H) ? zmid : back[(R - m) % H];
      z[m] = v2f{lane == 0 ? l0.x : got.x, lane == 0 ? l0.y : got.y};
    }
#if SPECINV_PLATE == 2
    // the state registers are free again: fetch the next frame's state now, it flies through the inverse FFT,
    // the overlap-add and the next forward FFT
    if (t + 1 < t_end) SPECINV_STATE_LOADS4(fi + 1);
#endif

#if SPECINV_XPREF == 2
    if (t + 1 < t_end) {
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        v2f q[QU];
        load_block4<R>(xrow, tailrow, a.L, a.T, c, t_begin, t_end, t + 1 + qq, lane, a.pad_mode, q);
#pragma unroll
        for (int i = 0; i < QU; ++i) znext[qq * QU + i] = q[i];
      }
    }
#endif

#if SPECINV_ABLATE & 4
#elif SPECINV_TW_REGS
    fft_inverse_t<R>(z, k, twr, tr);
#else
    asm volatile("" ::: "memory");   // re-read the twiddles instead of keeping them live since the forward FFT
    fft_inverse<R>(z, k, lds_tw1, tr);
#endif

    // ---- synthesis window, register overlap-add, one finished hop-block out
#pragma unroll
    for (int u = 0; u < R; ++u) z[u] = z[u] * lds_win[64 * u + lane];
#if SPECINV_PRIO & 4
    __builtin_amdgcn_s_setprio(3);
#endif
    if (live && t >= 2) {
      const long long o0 = (long long)(t - 2) * HOP;
      const v2f* envp = reinterpret_cast<const v2f*>(a.inv_env + o0);   // uniform
      v2f* outp = reinterpret_cast<v2f*>(orow + o0);
      // (the envelope block: register copy, or - first frames of an item - loaded and waited for in an arm of its own; see
      // fused_td_body, kernels_fast_td.h)
      v2f ev[QU], er[QU];
      if (!SPECINV_K4_ENVREG) {
#pragma unroll
        for (int i = 0; i < QU; ++i) {
          ev[i] = envp[64u * i + ulane];
          er[i] = env_rcp(ev[i]);
        }
      } else if (t >= 3) {
#pragma unroll
        for (int i = 0; i < QU; ++i) {
          ev[i] = envc[SPECINV_K4_ENVREG ? i : 0];
          er[i] = envr[(SPECINV_K4_ENVREG && SPECINV_IEEE) ? i : 0];
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
      for (int i = 0; i < QU; ++i) outp[64u * i + ulane] = env_apply_r(acc[i] + z[i], ev[i], er[i]);
    }
#if SPECINV_PRIO & 4
    __builtin_amdgcn_s_setprio(0);
#endif
#pragma unroll
    for (int i = 0; i < QU; ++i) {
      acc[i] = acc[QU + i] + z[QU + i];
      acc[QU + i] = acc[2 * QU + i] + z[2 * QU + i];
      acc[2 * QU + i] = z[3 * QU + i];
    }
  }
  if (t_end == a.T) {
    // the chunk that holds the last frame also finishes hop-block T (frames T-3 .. T-1)
    const long long o0 = (long long)(a.T - 2) * HOP;
    const v2f* envp = reinterpret_cast<const v2f*>(a.inv_env + o0);
    v2f* outp = reinterpret_cast<v2f*>(orow + o0);
#pragma unroll
    for (int i = 0; i < QU; ++i) outp[64u * i + ulane] = env_apply(acc[i], envp[64u * i + ulane]);
  } else {
    // what this chunk's last three frames contribute to the next chunk's first three hop-blocks
    v2f* tl = reinterpret_cast<v2f*>(a.xtail_out + ((long long)b * a.nchunks + c) * 3 * HOP);
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const v2f* envp = reinterpret_cast<const v2f*>(a.inv_env + (long long)(t_end + q - 2) * HOP);
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
#if SPECINV_K4_STAMPS   // diagnostic build: where and when the wave ran (tools/td_waves.py reads the dump)
  if (lane == 0 && a.stamps != nullptr) {
    const unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
    const unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));
    a.stamps[4 * (long long)w] = ((unsigned long long)xcc << 32) | hw;
    a.stamps[4 * (long long)w + 1] = k4_begin;
    a.stamps[4 * (long long)w + 2] = __builtin_amdgcn_s_memtime();
    a.stamps[4 * (long long)w + 3] = (unsigned long long)(t_end - t_begin);
  }
#endif
}

template <int R, int OV, int MODE, bool EVAL>
__global__ __launch_bounds__(256, R >= 32 ? 1 : (SPECINV_R8_W3 && R == 8) ? 3 : SPECINV_MINWAVES) void k_fused(FastArgs a) {
  using G = Geo<R>;
  using O = Ovl<R, OV>;
  constexpr int H = G::H, M = G::M, QU = O::QU, HOP = O::HOP, NB = O::NB, PB = O::PB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  v2f* lds_win = reinterpret_cast<v2f*>(smem);
  v2f* lds_tw1 = lds_win + M;
  // wave-uniform values are forced into SGPRs: every global address below is then
  // "scalar base + 32-bit lane offset" instead of one 64-bit VGPR pointer per access
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  v2f* tr = lds_tw1 + (R - 1) * 64 + wib * G::TR;

  for (int i = threadIdx.x; i < M; i += blockDim.x) lds_win[i] = v2f{a.window[2 * i], a.window[2 * i + 1]};
  for (int i = threadIdx.x; i < (R - 1) * 64; i += blockDim.x) {
    const int k1 = i / 64 + 1, l = i & 63;
    lds_tw1[i] = unit(2.0f * (float)((l * k1) % M) / (float)M);   // W_M^(l*k1)
  }
  __syncthreads();

  const int w = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + wib);   // (scalar: the frame loop and its padding tests branch on the scalar unit; left to the compiler the work-group size may arrive in a vector register and make all of it per-lane)
  if (w >= a.n_waves) return;
  const LaneConst<R> k = lane_consts<R>();
  const int lane = k.lane;
  const unsigned ulane = (unsigned)lane;
  const int b = w / a.nchunks, c = w - b * a.nchunks;
  const int t_begin = chunk_begin(c, a.T, a.nchunks, a.skew);
  const int t_end = chunk_begin(c + 1, a.T, a.nchunks, a.skew);
  const int t_start = t_begin;   // no halo: the previous chunk's share of the first NB hop-blocks comes via xtail
  const float* xrow = a.x_in + (long long)b * a.L;
  const float* tailrow = a.xtail_in + (long long)b * a.nchunks * NB * HOP;
  float* orow = a.x_out + (long long)b * a.L;
  const float half_scale = 0.5f * a.fwd_scale;

  v2f acc[NB * QU];
#pragma unroll
  for (int i = 0; i < NB * QU; ++i) acc[i] = v2f{0.0f, 0.0f};
  double sd = 0.0, so = 0.0;
#if SPECINV_TW_REGS
  TwRegs<R> twr;
#pragma unroll
  for (int k1 = 1; k1 < R; ++k1) twr.w[k1 - 1] = lds_tw1[(k1 - 1) * 64 + lane];
#endif

  // raw samples of the current frame: NB hop-blocks carried from frame to frame plus the
  // new one, which is fetched one frame ahead so that its latency hides behind a whole frame
  v2f xq[NB][QU], xn[QU];
#pragma unroll
  for (int q = 0; q < NB; ++q)
    load_block<R, OV>(xrow, tailrow, a.L, a.T, c, t_begin, t_end, t_start + q, lane, a.pad_mode, xq[q]);
  load_block<R, OV>(xrow, tailrow, a.L, a.T, c, t_begin, t_end, t_start + NB, lane, a.pad_mode, xn);

  // state of frame `FI`: uniform bases (SGPR) + unsigned 32-bit lane offsets -> "saddr + voffset" addressing
#define SPECINV_STATE_LOADS(FI)                                                            \
  do {                                                                                     \
    const long long fl_ = (FI);                                                            \
    const v4f* pin_ = a.P_in + fl_ * (H * 64);                                             \
    const v4f* min_ = a.m_pairs + fl_ * (H / 2 * 64);                                      \
    _Pragma("unroll") for (int j = 0; j < H; ++j) pp[j] = ld_stream(&pin_[j * 64u + ulane]); \
    _Pragma("unroll") for (int j = 0; j < H / 2; ++j) mm[j] = ld_stream(&min_[j * 64u + ulane]); \
    if (lane == 0) {                                                                       \
      pmid = a.Pmid_in[fl_];                                                               \
      mmid = a.m_mid[fl_];                                                                 \
    }                                                                                      \
  } while (0)

  for (int t = t_start; t < t_end; ++t) {
    // Keep the loop-invariant table reads (window, twiddles) and products inside the loop: hoisted out
    // of it they pin ~80 VGPRs for the whole kernel and cost a wave of occupancy.
    asm volatile("" ::: "memory");
    v2f wn = k.wn;
    asm volatile("" : "+v"(wn));
    constexpr bool live = true;
    const long long fi = (long long)b * a.T + t;
    v4f* pout = a.P_out + fi * (H * 64);
    const bool keep_xu = MODE == MODE_ADMM && a.U_out != nullptr;   // (uniform: a kernel argument)
    v4f pp[H], mm[H / 2];
    v2f pmid = v2f{0.0f, 0.0f}, umid = v2f{0.0f, 0.0f};
    float mmid = 0.0f;
#if SPECINV_PRIO
    if (R < 32) __builtin_amdgcn_s_setprio(SPECINV_PRIO & 3);   // (one wave per SIMD at R = 32: nothing to outrank)
#endif
    SPECINV_STATE_LOADS(fi);   // early: the loads fly during the forward FFT

    // ---- analysis: windowed frame -> registers; slide the sample window and prefetch the next hop-block
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
    if (t + 1 < t_end) load_block<R, OV>(xrow, tailrow, a.L, a.T, c, t_begin, t_end, t + OV, lane, a.pad_mode, xn);
#if SPECINV_PRIO
    if (R < 32) __builtin_amdgcn_s_setprio(0);
#endif

#if SPECINV_ABLATE & 4
#elif SPECINV_TW_REGS
    fft_forward_t<R>(z, k, twr, tr);
#else
    fft_forward<R>(z, k, lds_tw1, tr);
#endif

    // ---- conjugate partners: upper half of lane (64 - r)
    v2f rc[H];   // rc[i] pairs with own register H-1-i ... see below: rc[m-H] = Z[M - (lane + 64*(R-1-m))]
    // (the lane-0 special case is patched AFTER the shuffle: selecting between two elements of
    // one register array before it makes the compiler index the array dynamically)
#pragma unroll
    for (int m = H; m < R; ++m) {
      const v2f got = shfl2(z[m], k.partner);
      const v2f own = z[(m + 1) % R];              // lane 0 is its own partner, shifted by one register
      rc[m - H] = v2f{lane == 0 ? own.x : got.x, lane == 0 ? own.y : got.y};
    }

    // ---- per pair: split -> update -> fold back
    v2f back[H];   // back[j] = Z''[M - k_j], to be returned to the partner lane
#pragma unroll
    for (int j = 0; j < H; ++j) {
      // W_N^(lane + 64 j) = W_N^lane * W_{2R}^j
      const v2f wk = pair_twiddle<R>(wn, j);
      const v2f zk = z[j], zm = rc[R - 1 - j - H];
      const v2f e2 = add_conj(zk, zm);
      const v2f dd = sub_conj(zk, zm);
      const v2f tw = cmul_mi(wk, dd);                 // W * (-i (Zk - conj Zm))
      v2f xk = (e2 + tw) * half_scale;
      v2f xm = (e2 - tw) * v2f{half_scale, -half_scale};   // conj(...)
      v2f pk = v2f{pp[j].x, pp[j].y}, pm = v2f{pp[j].z, pp[j].w};
      v2f uk = v2f{0.0f, 0.0f}, um = v2f{0.0f, 0.0f}, sk = v2f{0.0f, 0.0f}, sm = v2f{0.0f, 0.0f};
      const float mk = (j & 1) ? mm[j / 2].z : mm[j / 2].x;
      const float mq = (j & 1) ? mm[j / 2].w : mm[j / 2].y;
      v2f ak = update_bin<MODE, EVAL>(xk, pk, uk, sk, mk, a, live, sd, so);
      v2f am = update_bin<MODE, EVAL>(xm, pm, um, sm, mq, a, live, sd, so);
      if (live) {
        st_stream(&pout[j * 64u + ulane], v4f{pk.x, pk.y, pm.x, pm.y});
        if (keep_xu) {
          st_stream(&a.X_out[fi * (H * 64) + j * 64u + ulane], v4f{sk.x, sk.y, sm.x, sm.y});
          st_stream(&a.U_out[fi * (H * 64) + j * 64u + ulane], v4f{uk.x, uk.y, um.x, um.y});
        }
      }
      if (j == 0 && lane == 0) {   // bins 0 and M: irfft ignores their imaginary parts
        ak.y = 0.0f;
        am.y = 0.0f;
      }
      const v2f e2i = add_conj(ak, am);
      const v2f o2i = cmulc(sub_conj(ak, am), wk);
      z[j] = add_i(e2i, o2i);
      back[j] = conj_sub_i(e2i, o2i);
    }
    // ---- bin M/2 (lane 0): X = conj(Z), Z'' = 2 conj(X')
    v2f zmid;
    {
      v2f xmid = z[H] * v2f{a.fwd_scale, -a.fwd_scale};
      const bool live0 = live && lane == 0;
      v2f smid = v2f{0.0f, 0.0f};
      const v2f am = update_bin<MODE, EVAL>(xmid, pmid, umid, smid, mmid, a, live0, sd, so);
      if (live0) {
        a.Pmid_out[fi] = pmid;
        if (keep_xu) {
          a.Xmid_out[fi] = smid;
          a.Umid_out[fi] = umid;
        }
      }
      zmid = am * v2f{2.0f, -2.0f};
    }
    // ---- return the mirrored halves
#pragma unroll
    for (int m = H; m < R; ++m) {
      const v2f got = shfl2(back[R - 1 - m], k.partner);
      const v2f l0 = (m == H) ? zmid : back[(R - m) % H];
      z[m] = v2f{lane == 0 ? l0.x : got.x, lane == 0 ? l0.y : got.y};
    }

#if SPECINV_ABLATE & 4
#elif SPECINV_TW_REGS
    fft_inverse_t<R>(z, k, twr, tr);
#else
    asm volatile("" ::: "memory");   // re-read the twiddles instead of keeping them live since the forward FFT
    fft_inverse<R>(z, k, lds_tw1, tr);
#endif

    // ---- synthesis window, register overlap-add, one finished hop-block out
#pragma unroll
    for (int u = 0; u < R; ++u) z[u] = z[u] * lds_win[64 * u + lane];
    if (live && t >= PB) {
      const long long o0 = (long long)(t - PB) * HOP;
      const v2f* envp = reinterpret_cast<const v2f*>(a.inv_env + o0);   // uniform
      v2f* outp = reinterpret_cast<v2f*>(orow + o0);
#pragma unroll
      for (int i = 0; i < QU; ++i) outp[64u * i + ulane] = env_apply(acc[i] + z[i], envp[64u * i + ulane]);
    }
#pragma unroll
    for (int i = 0; i < QU; ++i) {
#pragma unroll
      for (int q = 0; q + 1 < NB; ++q) acc[q * QU + i] = acc[(q + 1) * QU + i] + z[(q + 1) * QU + i];
      acc[(NB - 1) * QU + i] = z[NB * QU + i];
    }
  }
  if (t_end == a.T) {
    // the chunk that holds the last frame also finishes hop-blocks T .. T + PB - 2 (the frames that reach them are done)
#pragma unroll
    for (int q = 0; q < PB - 1; ++q) {
      const long long o0 = (long long)(a.T + q - PB) * HOP;
      const v2f* envp = reinterpret_cast<const v2f*>(a.inv_env + o0);
      v2f* outp = reinterpret_cast<v2f*>(orow + o0);
#pragma unroll
      for (int i = 0; i < QU; ++i) outp[64u * i + ulane] = env_apply(acc[q * QU + i], envp[64u * i + ulane]);
    }
  } else {
    // what this chunk's last NB frames contribute to the next chunk's first NB hop-blocks
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

// ISTFT of a spectrum held in pair layout: x = overlap-add(w * irfft(S)) / envelope  (methods.py:233: the
// initial signal of griffin_lim / ADMM).  Same wave-per-chunk walk as k_fused, without the analysis half.
template <int R, int OV>
__global__ __launch_bounds__(256, R >= 32 ? 1 : SPECINV_MINWAVES) void k_fused_istft(FastArgs a) {
  using G = Geo<R>;
  using O = Ovl<R, OV>;
  constexpr int H = G::H, M = G::M, QU = O::QU, HOP = O::HOP, NB = O::NB, PB = O::PB;
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
  const int w = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + wib);   // (scalar: the frame loop and its padding tests branch on the scalar unit; left to the compiler the work-group size may arrive in a vector register and make all of it per-lane)
  if (w >= a.n_waves) return;
  const LaneConst<R> k = lane_consts<R>();
  const int lane = k.lane;
  const unsigned ulane = (unsigned)lane;
  int b = w / a.nchunks, c = w - b * a.nchunks;
  if (a.skew != 0 && a.skew < 0x10000) {         // skewed chunk pairs: the first half of the waves the long ones (fused_td_body)
    const int half = a.n_waves >> 1, second = w >= half ? 1 : 0, wl = w - second * half, pairs = a.nchunks >> 1;
    b = wl / pairs;
    c = 2 * (wl - b * pairs) + second;
  }
  const int t_begin = chunk_begin(c, a.T, a.nchunks, a.skew);
  const int t_end = chunk_begin(c + 1, a.T, a.nchunks, a.skew);
  const int t_start = max(0, t_begin - NB);      // one-off kernel: recompute the NB-frame halo, write whole blocks
  float* orow = a.x_out + (long long)b * a.L;
  v2f acc[NB * QU];
#pragma unroll
  for (int i = 0; i < NB * QU; ++i) acc[i] = v2f{0.0f, 0.0f};
  for (int t = t_start; t < t_end; ++t) {
    asm volatile("" ::: "memory");
    v2f wn = k.wn;
    asm volatile("" : "+v"(wn));
    const bool live = t >= t_begin;
    const long long fi = (long long)b * a.T + t;
    const v4f* pin = a.P_in + fi * (H * 64);
    v4f pp[H];
#pragma unroll
    for (int j = 0; j < H; ++j) pp[j] = pin[j * 64u + ulane];
    v2f pmid = v2f{0.0f, 0.0f};
    if (lane == 0) pmid = a.Pmid_in[fi];
    v2f z[R], back[H];
#pragma unroll
    for (int j = 0; j < H; ++j) {
      const v2f wk = pair_twiddle<R>(wn, j);
      v2f ak = v2f{pp[j].x, pp[j].y} * a.inv_scale, am = v2f{pp[j].z, pp[j].w} * a.inv_scale;
      if (j == 0 && lane == 0) {
        ak.y = 0.0f;
        am.y = 0.0f;
      }
      const v2f e2i = add_conj(ak, am);
      const v2f o2i = cmulc(sub_conj(ak, am), wk);
      z[j] = add_i(e2i, o2i);
      back[j] = conj_sub_i(e2i, o2i);
    }
    const v2f zmid = pmid * v2f{2.0f * a.inv_scale, -2.0f * a.inv_scale};
#pragma unroll
    for (int m = H; m < R; ++m) {
      const v2f got = shfl2(back[R - 1 - m], k.partner);
      const v2f l0 = (m == H) ? zmid : back[(R - m) % H];
      z[m] = v2f{lane == 0 ? l0.x : got.x, lane == 0 ? l0.y : got.y};
    }
    fft_inverse<R>(z, k, lds_tw1, tr);
#pragma unroll
    for (int u = 0; u < R; ++u) z[u] = z[u] * lds_win[64 * u + lane];
    if (live && t >= PB) {
      const long long o0 = (long long)(t - PB) * HOP;
      const v2f* envp = reinterpret_cast<const v2f*>(a.inv_env + o0);
      v2f* outp = reinterpret_cast<v2f*>(orow + o0);
#pragma unroll
      for (int i = 0; i < QU; ++i) outp[64u * i + ulane] = env_apply(acc[i] + z[i], envp[64u * i + ulane]);
    }
#pragma unroll
    for (int i = 0; i < QU; ++i) {
#pragma unroll
      for (int q = 0; q + 1 < NB; ++q) acc[q * QU + i] = acc[(q + 1) * QU + i] + z[(q + 1) * QU + i];
      acc[(NB - 1) * QU + i] = z[NB * QU + i];
    }
  }
  if (t_end == a.T) {
#pragma unroll
    for (int q = 0; q < PB - 1; ++q) {
      const long long o0 = (long long)(a.T + q - PB) * HOP;
      const v2f* envp = reinterpret_cast<const v2f*>(a.inv_env + o0);
      v2f* outp = reinterpret_cast<v2f*>(orow + o0);
#pragma unroll
      for (int i = 0; i < QU; ++i) outp[64u * i + ulane] = env_apply(acc[q * QU + i], envp[64u * i + ulane]);
    }
  }
}


}  // namespace SI_FAST_NS (fast, or fast_approx in the approximate-projection units)
}  // namespace specinv
