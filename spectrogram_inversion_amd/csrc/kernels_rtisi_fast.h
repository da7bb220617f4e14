// RTISI-LA on the wave-level FFT of fast_core.h (float32, onesided, hop = n_fft/2, /4, /8, n_fft in {512, 1024, 2048}).
//
// The recursion (reference: torch_specinv/methods.py:363-404) is serial per batch item, so what matters is
// the latency of ONE inner step.  One workgroup owns one item; wave q owns look-ahead frame q for the whole
// run.  Everything an inner step touches stays on the CU:
//   * the frame ring (K kept + LA+1 look-ahead frames) lives in LDS, each frame already multiplied by the synthesis
//     window, so the overlap-add is plain additions;
//   * each wave keeps its frame's previous spectrum (pre_spec, conjugate-pair order), its target magnitudes, its
//     analysis window and the part of the overlap-add that the kept frames contribute (constant during the inner
//     iterations of a step) in REGISTERS; its own frame never leaves the registers between two inner iterations;
//   * per inner iteration a wave adds the hop-blocks of the other look-ahead frames that cover its frame, runs the
//     packed real FFT (in-register radix-R, lane-swap radix-4, one LDS transpose), applies momentum and the
//     magnitude projection to the pairs, transforms back and writes the frame to the ring: 2 workgroup
//     barriers per iteration, no global memory traffic except the once-per-frame target load and commit.
// Step latency drops from ~115 us (generic k_rtisi) to a few us.
#pragma once
#include "rtisi_fast_args.h"

#ifndef SPECINV_RTISI_PK      // complex products of the step loop's FFTs: 0 scalar (default: a lone wave per SIMD, see fft_forward_t), 1 packed
#define SPECINV_RTISI_PK 0
#endif

#if SPECINV_IEEE
#define RTISI_MSCALE a.inv_scale     // the target carries the inverse transform's scale (a power of two: exact)
#else
#define RTISI_MSCALE 1.0f
#endif

namespace specinv {
namespace fast {


// MAXT: 256 (look_ahead <= 3: one wave per SIMD, the full 512-register file per lane) or 512
template <int R, int MAXT, int OV>
__global__ __launch_bounds__(MAXT, 1) void k_rtisi_fast(RtisiFastArgs a) {
  using G = Geo<R>;
  constexpr int H = G::H, QU = R / OV, M = G::M, K = RtisiGeo<R, OV>::K;
  static_assert(R % OV == 0, "a hop-block must be whole registers");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int la = a.la, nslots = K + la + 1, nw = la + 1;
  v2f* ring = reinterpret_cast<v2f*>(smem);                 // [nslots][M]
  v2f* lds_tw1 = ring + (size_t)nslots * M;                 // [(R-1)*64]
  const int q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // this wave's look-ahead slot
  v2f* lds_wsyn = lds_tw1 + (R - 1) * 64;                   // [M] synthesis window w * hop / (w.w)
  v2f* lds_win = lds_wsyn + M;                              // [3][M] analysis window w, asym1, asym2
  v2f* lds_zero = lds_win + 3 * M;                          // [QU*64] what a frame outside the ring contributes
  v2f* xch = lds_zero + QU * 64;                            // [nw][TR] pre_spec exchange at the end of a step, and
  v2f* tr = xch + q * G::TR;                                // this wave's transpose scratch inside a step
  const int bi = blockIdx.x;
  const LaneConst<R> k = lane_consts<R>();
  const int lane = k.lane;
  const unsigned ulane = (unsigned)lane;

  for (int i = threadIdx.x; i < (R - 1) * 64; i += blockDim.x) {
    const int k1 = i / 64 + 1, l = i & 63;
    lds_tw1[i] = unit(2.0f * (float)((l * k1) % M) / (float)M);
  }
  for (int i = threadIdx.x; i < M; i += blockDim.x) {
    lds_wsyn[i] = reinterpret_cast<const v2f*>(a.wsyn)[i];
    lds_win[i] = reinterpret_cast<const v2f*>(a.window)[i];
    if (a.asym) {
      lds_win[M + i] = reinterpret_cast<const v2f*>(a.asym1)[i];
      lds_win[2 * M + i] = reinterpret_cast<const v2f*>(a.asym2)[i];
    }
    if (i < QU * 64) lds_zero[i] = v2f{0.0f, 0.0f};
  }
  v2f* st_ring = nullptr;
  v2f* st_wave = nullptr;
  if (a.state) {
    st_ring = reinterpret_cast<v2f*>(a.state) + (size_t)bi * rtisi_state_v2f<R>(nslots, nw);
    st_wave = st_ring + (size_t)nslots * M + (size_t)q * (2 * (H * 2 * 64) + 2 * 64);
  }
  if (a.resume) {
    for (int i = threadIdx.x; i < nslots * M; i += blockDim.x) ring[i] = st_ring[i];
  } else {
    for (int i = threadIdx.x; i < (nslots - 1) * M; i += blockDim.x) ring[i] = v2f{0.0f, 0.0f};
  }

  // analysis window of this wave's frame: the newest frame of an asymmetric run takes asym1 in the first inner
  // iteration and asym2 afterwards (methods.py:371-383)
  const bool newest = a.asym && q == la;

  // ---- first frame (methods.py:353-358): irfft of the zero-phase first target frame into the newest slot
  const long long f0 = (long long)bi * (a.mag_ring ? a.mag_ring : a.T);
  __syncthreads();
  if (q == 0 && !a.resume) {
    v2f z[R], back[H];
    const v4f* mp = a.m_pairs + f0 * (H / 2 * 64);
#pragma unroll
    for (int j = 0; j < H; ++j) {
      const v4f mm = mp[(j / 2) * 64u + ulane];
      const float mk = (j & 1) ? mm.z : mm.x, mq = (j & 1) ? mm.w : mm.y;
      const v2f wk = j == 0 ? k.wn : cmul_k(k.wn, w64(j * (32 / R)));
      const v2f ak = v2f{mk * a.inv_scale, 0.0f}, am = v2f{mq * a.inv_scale, 0.0f};
      const v2f e2i = add_conj(ak, am);
      const v2f o2i = cmulc_k(sub_conj(ak, am), wk);
      z[j] = add_i(e2i, o2i);
      back[j] = conj_sub_i(e2i, o2i);
    }
    float mmid = 0.0f;
    if (lane == 0) mmid = a.m_mid[f0];
    const v2f zmid = v2f{2.0f * mmid * a.inv_scale, 0.0f};
#pragma unroll
    for (int m = H; m < R; ++m) {
      const v2f got = shfl2(back[R - 1 - m], k.partner);
      const v2f l0 = (m == H) ? zmid : back[(R - m) % H];
      z[m] = v2f{lane == 0 ? l0.x : got.x, lane == 0 ? l0.y : got.y};
    }
    fft_inverse_t<R, true, false>(z, k, TwLds{lds_tw1, k.lane}, tr);   // (constant twiddles in the scalar form, like the step loop)
    v2f* dst = ring + (size_t)(nslots - 1) * M;
#pragma unroll
    for (int u = 0; u < R; ++u) dst[64 * u + lane] = z[u] * lds_wsyn[64 * u + lane];
  }
  __syncthreads();

  // pre_spec pairs (Re k, Im k, Re M-k, Im M-k) the momentum term of the next inner iteration uses: this frame's own
  // during a step; between steps the one of frame q+1, whose place this wave's frame takes (methods.py:389-392)
  v4f pre[H];
  v2f premid = v2f{0.0f, 0.0f};
#pragma unroll
  for (int j = 0; j < H; ++j) pre[j] = v4f{0.0f, 0.0f, 0.0f, 0.0f};
  if (a.resume) {
#pragma unroll
    for (int j = 0; j < H; ++j) {
      const v2f p0 = st_wave[(2 * j) * 64 + lane], p1 = st_wave[(2 * j + 1) * 64 + lane];
      pre[j] = v4f{p0.x, p0.y, p1.x, p1.y};
    }
    premid = st_wave[H * 2 * 64 + lane];
  }
  int base = a.i_begin % nslots;   // ring slot of the oldest kept frame
  const float half_scale = 0.5f * a.fwd_scale;
  // pass-1 twiddles of the FFT in registers (one wave per SIMD: the register file is not the constraint, and every LDS
  // read of a lone wave is exposed latency)
  TwRegs<R> twr;
#pragma unroll
  for (int k1 = 1; k1 < R; ++k1) twr.w[k1 - 1] = lds_tw1[(k1 - 1) * 64 + lane];
  // ... and so are the two windows of a symmetric-window run, and the wave's own frame between two inner iterations (round 5: 48
  // of an inner iteration's ~210 LDS instructions; only where a wave has the whole register file: look_ahead <= 3)
  constexpr bool kRegs = MAXT == 256;
  v2f wreg[kRegs ? R : 1], sreg[kRegs ? R : 1], zown[kRegs ? R : 1];
  if (kRegs) {
#pragma unroll
    for (int u = 0; u < R; ++u) {
      wreg[kRegs ? u : 0] = lds_win[64 * u + lane];
      sreg[kRegs ? u : 0] = lds_wsyn[64 * u + lane];
    }
  }

  // target of this wave's frame at outer step i (zero outside the spectrogram: methods.py:339); requested a step ahead - the only
  // global load of a step would otherwise sit, exposed, in front of its first inner iteration
  v4f mnext[H / 2];
  float mnext_mid = 0.0f;
  auto request_target = [&](int i) {
    const int tt = i + q - la;
#pragma unroll
    for (int j = 0; j < H / 2; ++j) mnext[j] = v4f{0.0f, 0.0f, 0.0f, 0.0f};
    mnext_mid = 0.0f;
    if (i < a.i_end && tt >= 0 && tt < a.n_valid) {
      const long long mrow = f0 + (a.mag_ring ? tt % a.mag_ring : tt);
      const v4f* mp = a.m_pairs + mrow * (H / 2 * 64);
#pragma unroll
      for (int j = 0; j < H / 2; ++j) mnext[j] = mp[j * 64u + ulane];
      if (lane == 0) mnext_mid = a.m_mid[mrow];
    }
  };
  // (one wave per SIMD - look_ahead <= 3 - has the registers for it; the two-wave form at n_fft 2048 would spill: it loads where it
  // needs them, as before)
  constexpr bool kAhead = MAXT == 256;
  if (kAhead) request_target(a.i_begin);
  for (int i = a.i_begin; i < a.i_end; ++i) {
    if (!kAhead) request_target(i);
    v4f mm[H / 2];
#pragma unroll
    for (int j = 0; j < H / 2; ++j) mm[j] = mnext[j] * RTISI_MSCALE;
    float mmid = mnext_mid * RTISI_MSCALE;
    if (kAhead) request_target(i + 1);

    // ---- overlap-add (methods.py:365-370), the part that does not change during the inner iterations: the kept
    // frames (ring frames 0..K-1) that reach into this frame.  Frame kf covers hop-block qi of this frame with its own
    // hop-block d = K - kf + q + qi if that is < OV; otherwise the block of zeros is added (no branches: a conditional
    // update of z[] costs whole-array register copies)
    v2f zc[R];
#pragma unroll
    for (int u = 0; u < R; ++u) zc[u] = v2f{0.0f, 0.0f};
#pragma unroll
    for (int kf = 0; kf < K; ++kf) {
      int slot = base + kf;
      if (slot >= nslots) slot -= nslots;
#pragma unroll
      for (int qi = 0; qi <= kf; ++qi) {
        const int d = K - kf + q + qi;
        const v2f* fr = d < OV ? ring + (size_t)slot * M + d * QU * 64 : lds_zero;
#pragma unroll
        for (int i2 = 0; i2 < QU; ++i2) zc[qi * QU + i2] += fr[64 * i2 + lane];
      }
    }
    int so = base + K + q;
    if (so >= nslots) so -= nslots;
    v2f* own = ring + (size_t)so * M;                  // this wave's frame in the ring
    const bool shift = i > 0 && q < la;                // pre holds frame q+1's spectrum of the previous step (:389-391)

    for (int it = 0; it < a.max_iter; ++it) {
      // ---- kept part + own frame + the hop-blocks of the other look-ahead frames q + dl that cover it
      v2f z[R];
      if (kRegs && it > 0) {
#pragma unroll
        for (int u = 0; u < R; ++u) z[u] = zc[u] + zown[kRegs ? u : 0];
      } else {
#pragma unroll
        for (int u = 0; u < R; ++u) z[u] = zc[u] + own[64 * u + lane];
      }
#pragma unroll
      for (int dl = 1 - OV; dl < OV; ++dl) {
        if (dl == 0) continue;
        const int qo = q + dl;
        const bool ok = qo >= 0 && qo <= la;
        int slot = base + K + qo;
        if (slot >= nslots) slot -= nslots;
        const v2f* fo = ring + (size_t)(ok ? slot : 0) * M;
#pragma unroll
        for (int qi = (dl > 0 ? dl : 0); qi < (dl > 0 ? OV : OV + dl); ++qi) {
          const v2f* fr = ok ? fo + (qi - dl) * QU * 64 : lds_zero;   // its hop-block d = qi - dl
#pragma unroll
          for (int i2 = 0; i2 < QU; ++i2) z[qi * QU + i2] += fr[64 * i2 + lane];
        }
      }
      __syncthreads();                               // every wave has read the ring; slots may be rewritten
      if (kRegs && !newest) {
#pragma unroll
        for (int u = 0; u < R; ++u) z[u] = z[u] * wreg[kRegs ? u : 0];
      } else {
        const v2f* win = newest ? lds_win + (it == 0 ? M : 2 * M) : lds_win;
#pragma unroll
        for (int u = 0; u < R; ++u) z[u] = z[u] * win[64 * u + lane];
      }

      fft_forward_t<R, SPECINV_RTISI_PK != 0>(z, k, twr, tr);

      v2f rc[H];
#pragma unroll
      for (int m = H; m < R; ++m) {
        const v2f got = shfl2(z[m], k.partner);
        const v2f own = z[(m + 1) % R];
        rc[m - H] = v2f{lane == 0 ? own.x : got.x, lane == 0 ? own.y : got.y};
      }
      const float lr = (it > 0 || shift) ? a.lr : 0.0f;   // methods.py:387-391
      v2f back[H];
#pragma unroll
      for (int j = 0; j < H; ++j) {
        const v2f wk = j == 0 ? k.wn : cmul_k(k.wn, w64(j * (32 / R)));
        const v2f zk = z[j], zm = rc[R - 1 - j - H];
        const v2f e2 = add_conj(zk, zm);
        const v2f dd = sub_conj(zk, zm);
        const v2f tw = cmul_k(mul_mi(wk), dd);
        const v2f xk = (e2 + tw) * half_scale;
        const v2f xm = (e2 - tw) * v2f{half_scale, -half_scale};
        const v4f p = pre[j];
        const v2f sk = v2f{fmaf(-lr, p.x, xk.x), fmaf(-lr, p.y, xk.y)};
        const v2f sm = v2f{fmaf(-lr, p.z, xm.x), fmaf(-lr, p.w, xm.y)};
        pre[j] = v4f{sk.x, sk.y, sm.x, sm.y};        // :392
        const float mk = (j & 1) ? mm[j / 2].z : mm[j / 2].x;
        const float mq = (j & 1) ? mm[j / 2].w : mm[j / 2].y;
#if SPECINV_IEEE
        // :394-396 in the reference's operation order, (s m) r with r the correctly rounded 1 / |s| (fast_core.h: ref_rcp_abs, as in
        // the Griffin-Lim / ADMM kernels since round 4); the inverse transform's 1 / n_fft - a power of two - rides on m
        const v2f rr = ref_rcp_abs2(v2f{ref_norm2(sk), ref_norm2(sm)});
        const float ik = rr.x, im = rr.y;
#else
        const float ik = __builtin_amdgcn_rcpf(fast_abs(sk) + 1e-16f) * a.inv_scale;   // :394-396
        const float im = __builtin_amdgcn_rcpf(fast_abs(sm) + 1e-16f) * a.inv_scale;
#endif
        v2f ak = v2f{(sk.x * mk) * ik, (sk.y * mk) * ik};
        v2f am = v2f{(sm.x * mq) * im, (sm.y * mq) * im};
        if (j == 0 && lane == 0) {
          ak.y = 0.0f;
          am.y = 0.0f;
        }
        const v2f e2i = add_conj(ak, am);
        const v2f o2i = cmulc_k(sub_conj(ak, am), wk);
        z[j] = add_i(e2i, o2i);
        back[j] = conj_sub_i(e2i, o2i);
      }
      v2f zmid;
      {
        const v2f xmid = z[H] * v2f{a.fwd_scale, -a.fwd_scale};
        const v2f p = premid;
        const v2f s = v2f{fmaf(-lr, p.x, xmid.x), fmaf(-lr, p.y, xmid.y)};
        premid = s;
#if SPECINV_IEEE
        const float inv = ref_rcp_abs(ref_norm2(s));
#else
        const float inv = __builtin_amdgcn_rcpf(fast_abs(s) + 1e-16f) * a.inv_scale;
#endif
        zmid = v2f{(s.x * mmid) * inv, (s.y * mmid) * inv} * v2f{2.0f, -2.0f};
      }
#pragma unroll
      for (int m = H; m < R; ++m) {
        const v2f got = shfl2(back[R - 1 - m], k.partner);
        const v2f l0 = (m == H) ? zmid : back[(R - m) % H];
        z[m] = v2f{lane == 0 ? l0.x : got.x, lane == 0 ? l0.y : got.y};
      }
      fft_inverse_t<R, SPECINV_RTISI_PK != 0>(z, k, twr, tr);

      if (q == 0 && i >= la && it == a.max_iter - 1) {   // commit look-ahead slot 0 (methods.py:401-404)
        const v2f* w = reinterpret_cast<const v2f*>(a.window);
        const long long orow = a.out_ring ? (long long)bi * a.out_ring + (i - la) % a.out_ring : (long long)bi * a.T + (i - la);
        v2f* out = reinterpret_cast<v2f*>(a.frames_out + orow * (2 * M));
#pragma unroll
        for (int u = 0; u < R; ++u) out[64u * u + ulane] = z[u] * w[64u * u + ulane];
      }
      if (kRegs) {
#pragma unroll
        for (int u = 0; u < R; ++u) {
          zown[kRegs ? u : 0] = z[u] * sreg[kRegs ? u : 0];
          own[64 * u + lane] = zown[kRegs ? u : 0];   // :398
        }
      } else {
#pragma unroll
        for (int u = 0; u < R; ++u) own[64 * u + lane] = z[u] * lds_wsyn[64 * u + lane];   // :398
      }
      __syncthreads();                               // new frames visible before the next overlap-add
    }

    // ---- slide: publish pre_spec, take over the one of the frame that moves into this wave's place
    {
      v2f* mine = xch + (size_t)q * (H * 2 * 64 + 64);
#pragma unroll
      for (int j = 0; j < H; ++j) {
        mine[(2 * j) * 64 + lane] = v2f{pre[j].x, pre[j].y};
        mine[(2 * j + 1) * 64 + lane] = v2f{pre[j].z, pre[j].w};
      }
      if (lane == 0) mine[H * 2 * 64] = premid;
    }
    __syncthreads();
    if (q < la) {
      const v2f* nxt = xch + (size_t)(q + 1) * (H * 2 * 64 + 64);
#pragma unroll
      for (int j = 0; j < H; ++j) {
        const v2f a0 = nxt[(2 * j) * 64 + lane], a1 = nxt[(2 * j + 1) * 64 + lane];
        pre[j] = v4f{a0.x, a0.y, a1.x, a1.y};
      }
      premid = nxt[H * 2 * 64];
    }
    if (q == la) {   // the oldest frame's slot becomes the new (zero) newest frame
      v2f* fresh = ring + (size_t)base * M;
#pragma unroll
      for (int u = 0; u < R; ++u) fresh[64 * u + lane] = v2f{0.0f, 0.0f};
    }
    base = base + 1 == nslots ? 0 : base + 1;
    __syncthreads();
  }
  if (a.state) {   // what a later launch needs to carry on from step i_end
    for (int i = threadIdx.x; i < nslots * M; i += blockDim.x) st_ring[i] = ring[i];
#pragma unroll
    for (int j = 0; j < H; ++j) {
      st_wave[(2 * j) * 64 + lane] = v2f{pre[j].x, pre[j].y};
      st_wave[(2 * j + 1) * 64 + lane] = v2f{pre[j].z, pre[j].w};
    }
    st_wave[H * 2 * 64 + lane] = premid;
  }
}


}  // namespace fast
}  // namespace specinv
