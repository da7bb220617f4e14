// Frame kernels on the wave-level FFT of fast_core.h (float32, one-sided, n_fft 512 ... 4096, ANY hop / centring /
// pad mode): the stand-alone transforms (k_fast_stft, k_fast_inverse_frames), one iteration a frame at a time
// (k_semi + the gather overlap-add of kernels_generic.h), the same over chunks of frames with the overlap-add in an LDS
// ring (k_hop + k_hop_tails), and the adjoint of the analysis in that structure (k_hop_inverse + k_hop_tails_raw).
// Built on fast_core.h; compiled in tu_frame_*.hip; the host side is FastState<float> (fast_state.h).
#pragma once
#include "fast_core.h"

namespace specinv {
namespace SI_FAST_NS {

// ---- stand-alone transforms on the wave-level FFT (any hop; used by specinv_stft and the L_BFGS objective) ----



// torch.stft (center, reflect, onesided): one wave per frame, spectrum written in natural bin order
// (register j of lane l is bin l + 64 j: every store instruction covers 64 consecutive bins)
template <int R>
__global__ __launch_bounds__(256) void k_fast_stft(FastXformArgs a) {
  using G = Geo<R>;
  constexpr int H = G::H, M = G::M;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  v2f* lds_win = reinterpret_cast<v2f*>(smem);
  v2f* lds_tw1 = lds_win + M;
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  v2f* tr = lds_tw1 + (R - 1) * 64 + wib * G::TR;
  xform_tables<R>(a.window, lds_win, lds_tw1);
  const LaneConst<R> k = lane_consts<R>();
  const int lane = k.lane;
  for (long long fi = (long long)blockIdx.x * 4 + wib; fi < a.n_frames_total; fi += (long long)gridDim.x * 4) {
    const long long b = fi / a.T;
    const int t = (int)(fi - b * a.T);
    v2f z[R];
    load_frame_regs<R>(a.x + b * a.len, a.len, (long long)t * a.hop - a.pad, lane, a.pad_mode, lds_win, z);
    fft_forward<R>(z, k, lds_tw1, tr);
    v2f rc[H];
#pragma unroll
    for (int m = H; m < R; ++m) {
      const v2f got = shfl2(z[m], k.partner);
      const v2f own = z[(m + 1) % R];
      rc[m - H] = v2f{lane == 0 ? own.x : got.x, lane == 0 ? own.y : got.y};
    }
    v2f* out = a.spec + fi * (M + 1);
    const float hs = 0.5f * a.scale;
#pragma unroll
    for (int j = 0; j < H; ++j) {
      const v2f wk = pair_twiddle<R>(k.wn, j);
      const v2f zk = z[j], zm = rc[R - 1 - j - H];
      const v2f e2 = add_conj(zk, zm);
      const v2f tw = cmul_mi(wk, sub_conj(zk, zm));
      const int kk = lane + 64 * j;
      out[kk] = (e2 + tw) * hs;
      out[M - kk] = (e2 - tw) * v2f{hs, -hs};
    }
    if (lane == 0) out[M / 2] = z[H] * v2f{a.scale, -a.scale};
  }
}

// frames[n] = window[n] * scale * Re sum_k Xfull[k] e^{+2 pi i k n / N} for the Hermitian extension Xfull of the
// stored onesided spectrum (imaginary parts of DC / Nyquist ignored): the synthesis half of an ISTFT and the
// adjoint of the STFT (L_BFGS gradient).  Natural bin order in, one wave per frame.
template <int R>
__global__ __launch_bounds__(256) void k_fast_inverse_frames(FastXformArgs a) {
  using G = Geo<R>;
  constexpr int H = G::H, M = G::M;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  v2f* lds_win = reinterpret_cast<v2f*>(smem);
  v2f* lds_tw1 = lds_win + M;
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  v2f* tr = lds_tw1 + (R - 1) * 64 + wib * G::TR;
  xform_tables<R>(a.window, lds_win, lds_tw1);
  const LaneConst<R> k = lane_consts<R>();
  const int lane = k.lane;
  for (long long fi = (long long)blockIdx.x * 4 + wib; fi < a.n_frames_total; fi += (long long)gridDim.x * 4) {
    const v2f* in = a.spec + fi * (M + 1);
    v2f z[R], back[H];
#pragma unroll
    for (int j = 0; j < H; ++j) {
      const v2f wk = pair_twiddle<R>(k.wn, j);
      const int kk = lane + 64 * j;
      v2f ak = in[kk] * a.scale, am = in[M - kk] * a.scale;
      if (j == 0 && lane == 0) {
        ak.y = 0.0f;
        am.y = 0.0f;
      }
      const v2f e2i = add_conj(ak, am);
      const v2f o2i = cmulc(sub_conj(ak, am), wk);
      z[j] = add_i(e2i, o2i);
      back[j] = conj_sub_i(e2i, o2i);
    }
    v2f zmid = v2f{0.0f, 0.0f};
    if (lane == 0) zmid = in[M / 2] * v2f{2.0f * a.scale, -2.0f * a.scale};
#pragma unroll
    for (int m = H; m < R; ++m) {
      const v2f got = shfl2(back[R - 1 - m], k.partner);
      const v2f l0 = (m == H) ? zmid : back[(R - m) % H];
      z[m] = v2f{lane == 0 ? l0.x : got.x, lane == 0 ? l0.y : got.y};
    }
    fft_inverse<R>(z, k, lds_tw1, tr);
    v2f* out = reinterpret_cast<v2f*>(a.frames + fi * (2 * M));
#pragma unroll
    for (int u = 0; u < R; ++u) out[64u * u + (unsigned)lane] = z[u] * lds_win[64 * u + lane];
  }
}

// ---- one iteration for any hop / centring (n_fft 1024 / 2048): frame kernel + gather overlap-add ---------------
// Same wave-level FFT, pair-layout state and update as k_fused, but a wave takes whole frames one at a time (samples
// straight from x, any hop, any pad mode) and writes the windowed synthesis frame to `frames`; k_ola then does the
// overlap-add / envelope division.  Costs one frame round trip (8 N bytes per frame) more than k_fused; used when
// hop != n_fft/4 or centre = False.  MODE_INIT synthesises the stored spectrum as it is (the initial ISTFT).

// One frame of the frame kernels: state in, (samples -> spectrum -> update) unless MODE_INIT, state out, inverse
// transform.  On return z holds the synthesis frame before its window (register u <-> samples 128u + 2 lane, +1).
// PRE: z already holds the frame's samples (unwindowed), fetched by the caller one frame ahead.
// TWO: a two-sided spectrogram (onesided=False: target and state hold all N bins).  The frame is real, so its spectrum at the
// mirror bin N - f is the conjugate of bin f; the reference updates both bins on their own - each with its own target and state
// (methods.py:243-247 / :467-475 on the full spectrum) - and `ifft(.).real` (:142-146) sees the Hermitian part
// (Y_f + conj Y_{N-f}) / 2 of the result: the update runs a second time on the conjugate with the mirror arrays (FastArgs::P2_out),
// the two results are averaged, the rest of the frame is the one-sided kernel's.
template <int R, int MODE, bool EVAL, bool PRE = false, bool TWO = false>
__device__ __forceinline__ void semi_frame(const FastArgs& a, long long fi, long long b, int t, int hop, int pad,
                                           const LaneConst<R>& k, const v2f* lds_win, const v2f* lds_tw1, v2f* tr,
                                           v2f (&z)[R], double& sd, double& so) {
  using G = Geo<R>;
  constexpr int H = G::H;
  constexpr int UMODE = MODE == MODE_INIT ? MODE_GLA : MODE;
  const int lane = k.lane;
  const unsigned ulane = (unsigned)lane;
  const float half_scale = 0.5f * a.fwd_scale;
  v4f pp[H], mm[H / 2];
  const bool keep_xu = MODE == MODE_ADMM && a.U_out != nullptr;   // (uniform: a kernel argument)
  v2f pmid = v2f{0.0f, 0.0f}, umid = v2f{0.0f, 0.0f};
  float mmid = 0.0f;
  v4f pp2[TWO ? H : 1], mm2[TWO ? (H / 2) : 1];
  v2f pmid2 = v2f{0.0f, 0.0f};
  float mmid2 = 0.0f;
  auto conjv = [](v2f v) { return v2f{v.x, -v.y}; };
  if (TWO) {
    const v4f* pin2 = a.P2_out + fi * (H * 64);
#pragma unroll
    for (int j = 0; j < H; ++j) pp2[TWO ? j : 0] = ld_stream(&pin2[j * 64u + ulane]);
    if (MODE != MODE_INIT) {
      const v4f* min2 = a.m2_pairs + fi * (H / 2 * 64);
#pragma unroll
      for (int j = 0; j < H / 2; ++j) mm2[TWO ? j : 0] = ld_stream(&min2[j * 64u + ulane]);
    }
    if (lane == 0) {
      pmid2 = a.Pmid2_out[fi];
      if (MODE != MODE_INIT) mmid2 = a.m2_mid[fi];
    }
  }
  {
    v4f* pin = a.P_out + fi * (H * 64);
#pragma unroll
    for (int j = 0; j < H; ++j) pp[j] = ld_stream(&pin[j * 64u + ulane]);
    if (MODE != MODE_INIT) {
      const v4f* min = a.m_pairs + fi * (H / 2 * 64);
#pragma unroll
      for (int j = 0; j < H / 2; ++j) mm[j] = ld_stream(&min[j * 64u + ulane]);
    }
    if (lane == 0) {
      pmid = a.Pmid_out[fi];
      if (MODE != MODE_INIT) mmid = a.m_mid[fi];
    }
  }
  v2f rc[H];
  if (MODE != MODE_INIT) {
    if (PRE) {
#pragma unroll
      for (int u = 0; u < R; ++u) z[u] = z[u] * lds_win[64 * u + lane];
    } else {
      load_frame_regs<R>(a.x_in + b * a.L, a.L, (long long)t * hop - pad, lane, a.pad_mode, lds_win, z);
    }
    fft_forward<R>(z, k, lds_tw1, tr);
#pragma unroll
    for (int m = H; m < R; ++m) {
      const v2f got = shfl2(z[m], k.partner);
      const v2f own = z[(m + 1) % R];
      rc[m - H] = v2f{lane == 0 ? own.x : got.x, lane == 0 ? own.y : got.y};
    }
  }
  v2f back[H];
#pragma unroll
  for (int j = 0; j < H; ++j) {
    const v2f wk = pair_twiddle<R>(k.wn, j);
    v2f pk = v2f{pp[j].x, pp[j].y}, pm = v2f{pp[j].z, pp[j].w};
    v2f ak, am;
    if (MODE != MODE_INIT) {
      const v2f zk = z[j], zm = rc[R - 1 - j - H];
      const v2f e2 = add_conj(zk, zm);
      const v2f tw = cmul_mi(wk, sub_conj(zk, zm));
      const v2f xk = (e2 + tw) * half_scale;
      const v2f xm = (e2 - tw) * v2f{half_scale, -half_scale};
      v2f uk = v2f{0.0f, 0.0f}, um = v2f{0.0f, 0.0f}, sk = v2f{0.0f, 0.0f}, sm = v2f{0.0f, 0.0f};
      const float mk = (j & 1) ? mm[j / 2].z : mm[j / 2].x;
      const float mq = (j & 1) ? mm[j / 2].w : mm[j / 2].y;
      ak = update_bin<UMODE, EVAL>(xk, pk, uk, sk, mk, a, true, sd, so);
      am = update_bin<UMODE, EVAL>(xm, pm, um, sm, mq, a, true, sd, so);
      st_stream(&a.P_out[fi * (H * 64) + j * 64u + ulane], v4f{pk.x, pk.y, pm.x, pm.y});
      if (keep_xu) {
        st_stream(&a.X_out[fi * (H * 64) + j * 64u + ulane], v4f{sk.x, sk.y, sm.x, sm.y});
        st_stream(&a.U_out[fi * (H * 64) + j * 64u + ulane], v4f{uk.x, uk.y, um.x, um.y});
      }
      if (TWO) {
        const int j2 = TWO ? j : 0;
        v2f pk2 = v2f{pp2[j2].x, pp2[j2].y}, pm2 = v2f{pp2[j2].z, pp2[j2].w};
        const float mk2 = (j & 1) ? mm2[j2 / 2].z : mm2[j2 / 2].x;
        const float mq2 = (j & 1) ? mm2[j2 / 2].w : mm2[j2 / 2].y;
        const bool live2 = !(j == 0 && lane == 0);          // bins 0 and M are their own mirror images
        v2f u2 = v2f{0.0f, 0.0f}, s2 = v2f{0.0f, 0.0f};
        const v2f ak2 = update_bin<UMODE, EVAL>(conjv(xk), pk2, u2, s2, mk2, a, live2, sd, so);
        const v2f am2 = update_bin<UMODE, EVAL>(conjv(xm), pm2, u2, s2, mq2, a, live2, sd, so);
        st_stream(&a.P2_out[fi * (H * 64) + j * 64u + ulane], v4f{pk2.x, pk2.y, pm2.x, pm2.y});
        if (live2) {
          ak = (ak + conjv(ak2)) * 0.5f;
          am = (am + conjv(am2)) * 0.5f;
        }
      }
    } else {
      ak = pk * a.inv_scale;
      am = pm * a.inv_scale;
      if (TWO && !(j == 0 && lane == 0)) {
        const int j2 = TWO ? j : 0;
        ak = (ak + conjv(v2f{pp2[j2].x, pp2[j2].y} * a.inv_scale)) * 0.5f;
        am = (am + conjv(v2f{pp2[j2].z, pp2[j2].w} * a.inv_scale)) * 0.5f;
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
  v2f zmid;
  if (MODE != MODE_INIT) {
    const v2f xmid = z[H] * v2f{a.fwd_scale, -a.fwd_scale};
    const bool live0 = lane == 0;
    v2f smid = v2f{0.0f, 0.0f};
    v2f am = update_bin<UMODE, EVAL>(xmid, pmid, umid, smid, mmid, a, live0, sd, so);
    if (live0) {
      a.Pmid_out[fi] = pmid;
      if (keep_xu) {
        a.Xmid_out[fi] = smid;
        a.Umid_out[fi] = umid;
      }
    }
    if (TWO) {
      v2f u2 = v2f{0.0f, 0.0f}, s2 = v2f{0.0f, 0.0f};
      const v2f am2 = update_bin<UMODE, EVAL>(conjv(xmid), pmid2, u2, s2, mmid2, a, live0, sd, so);
      if (live0) a.Pmid2_out[fi] = pmid2;
      am = (am + conjv(am2)) * 0.5f;
    }
    zmid = am * v2f{2.0f, -2.0f};
  } else {
    v2f pm_ = pmid;
    if (TWO) pm_ = (pmid + conjv(pmid2)) * 0.5f;
    zmid = pm_ * v2f{2.0f * a.inv_scale, -2.0f * a.inv_scale};
  }
#pragma unroll
  for (int m = H; m < R; ++m) {
    const v2f got = shfl2(back[R - 1 - m], k.partner);
    const v2f l0 = (m == H) ? zmid : back[(R - m) % H];
    z[m] = v2f{lane == 0 ? l0.x : got.x, lane == 0 ? l0.y : got.y};
  }
  fft_inverse<R>(z, k, lds_tw1, tr);
}

template <int R, int MODE, bool EVAL>
__global__ __launch_bounds__(256) void k_semi(SemiArgs s) {
  using G = Geo<R>;
  constexpr int M = G::M;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  v2f* lds_win = reinterpret_cast<v2f*>(smem);
  v2f* lds_tw1 = lds_win + M;
  const FastArgs& a = s.f;
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  v2f* tr = lds_tw1 + (R - 1) * 64 + wib * G::TR;
  xform_tables<R>(a.window, lds_win, lds_tw1);
  const LaneConst<R> k = lane_consts<R>();
  const int lane = k.lane;
  const unsigned ulane = (unsigned)lane;
  double sd = 0.0, so = 0.0;
  for (long long fi = (long long)blockIdx.x * 4 + wib; fi < s.n_frames_total; fi += (long long)gridDim.x * 4) {
    const long long b = fi / a.T;
    const int t = (int)(fi - b * a.T);
    v2f z[R];
    semi_frame<R, MODE, EVAL>(a, fi, b, t, s.hop, s.pad, k, lds_win, lds_tw1, tr, z, sd, so);
    v2f* out = reinterpret_cast<v2f*>(s.frames + fi * (2 * M));
#pragma unroll
    for (int u = 0; u < R; ++u) out[64u * u + ulane] = z[u] * lds_win[64 * u + lane];
  }
  if (EVAL) {
    const double d = wave_sum(sd), o = wave_sum(so);
    if (lane == 0) {
      const long long w = (long long)blockIdx.x * 4 + wib;
      a.partials[2 * w] = d;
      a.partials[2 * w + 1] = o;
    }
  }
}

// k_semi for a two-sided spectrogram (semi_frame<..., TWO>): float32, onesided=False - what two thirds of the reference's own
// parametrisations ask for (test/test_griffin.py:24-32) - ran on the coverage kernels until round 5.
template <int R, int MODE, bool EVAL>
__global__ __launch_bounds__(256) void k_semi2(SemiArgs s) {
  using G = Geo<R>;
  constexpr int M = G::M;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  v2f* lds_win = reinterpret_cast<v2f*>(smem);
  v2f* lds_tw1 = lds_win + M;
  const FastArgs& a = s.f;
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  v2f* tr = lds_tw1 + (R - 1) * 64 + wib * G::TR;
  xform_tables<R>(a.window, lds_win, lds_tw1);
  const LaneConst<R> k = lane_consts<R>();
  const int lane = k.lane;
  const unsigned ulane = (unsigned)lane;
  double sd = 0.0, so = 0.0;
  for (long long fi = (long long)blockIdx.x * 4 + wib; fi < s.n_frames_total; fi += (long long)gridDim.x * 4) {
    const long long b = fi / a.T;
    const int t = (int)(fi - b * a.T);
    v2f z[R];
    semi_frame<R, MODE, EVAL, false, true>(a, fi, b, t, s.hop, s.pad, k, lds_win, lds_tw1, tr, z, sd, so);
    v2f* out = reinterpret_cast<v2f*>(s.frames + fi * (2 * M));
#pragma unroll
    for (int u = 0; u < R; ++u) out[64u * u + ulane] = z[u] * lds_win[64 * u + lane];
  }
  if (EVAL) {
    const double d = wave_sum(sd), o = wave_sum(so);
    if (lane == 0) {
      const long long w = (long long)blockIdx.x * 4 + wib;
      a.partials[2 * w] = d;
      a.partials[2 * w + 1] = o;
    }
  }
}

// ---- the same, with the overlap-add on the chip (any hop <= n_fft, n_fft 512 ... 2048) ---------------------------------
// A wave walks a chunk of consecutive frames of one item (like k_fused) and adds every synthesis frame into a private
// ring of n_fft samples in LDS, in frame order - the order of k_ola's gather, so the sums round identically.  After
// frame t the hop samples [t hop, (t+1) hop) are final: they are multiplied by the envelope's reciprocal and stored, and their ring
// slots cleared.  No frame round trip through HBM.  At a chunk boundary the first n_fft - hop samples of the later
// chunk lack what the earlier chunk's last frames add: the later chunk stores its own partial sums undivided, the
// earlier one leaves the rest of its ring in `xtail`, and k_hop_tails adds the two and divides (a few MB per
// iteration).  x ping-pongs between two buffers, the spectral state is updated in place.

template <int R, int MODE, bool EVAL, bool TWO>
__device__ __forceinline__ void hop_body(const HopArgs& s) {
  using G = Geo<R>;
  constexpr int M = G::M, N = G::N;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  v2f* lds_win = reinterpret_cast<v2f*>(smem);
  v2f* lds_tw1 = lds_win + M;
  const FastArgs& a = s.f;
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwib = __builtin_amdgcn_readfirstlane(blockDim.x >> 6);
  v2f* tr = lds_tw1 + (R - 1) * 64 + wib * G::TR;
  float* ring = reinterpret_cast<float*>(lds_tw1 + (R - 1) * 64 + nwib * G::TR) + wib * N;
  xform_tables<R>(a.window, lds_win, lds_tw1);
  const LaneConst<R> k = lane_consts<R>();
  const int lane = k.lane;
  const int w = __builtin_amdgcn_readfirstlane(blockIdx.x * nwib + wib);
  double sd = 0.0, so = 0.0;
  if (w < a.n_waves) {
    const int b = w / a.nchunks, c = w - b * a.nchunks;
    const int t0 = hop_chunk_begin(c, a.T, a.nchunks), t1 = hop_chunk_begin(c + 1, a.T, a.nchunks);
    const int hop = s.hop, keep = N - hop;
    const float* env = s.env;
    float* xo = a.x_out + (long long)b * a.L;
#pragma unroll
    for (int u = 0; u < R; ++u) reinterpret_cast<v2f*>(ring)[64 * u + lane] = v2f{0.0f, 0.0f};
    // padded-signal position p <-> ring slot p mod N; samples below `raw_end` of a later chunk stay undivided
    const long long raw_end = c > 0 ? (long long)t0 * hop + keep : -1;
    int slot0 = (int)(((long long)t0 * hop) % N);          // ring slot of the frame's first sample
    // even hop, padding and length: ring slots, signal positions and row starts of the emitted samples are all even
    const bool pairs = ((hop | s.pad) & 1) == 0 && (a.L & 1) == 0;
    // the samples of frame t + 1 are requested before frame t is transformed (they come from L2 / HBM with a latency
    // that two waves per SIMD do not hide)
    constexpr bool PRE = MODE != MODE_INIT;
    v2f zn[R];
    if (PRE) load_frame_raw<R>(a.x_in + (long long)b * a.L, a.L, (long long)t0 * hop - s.pad, lane, a.pad_mode, zn);
    for (int t = t0; t < t1; ++t) {
      const long long fi = (long long)b * a.T + t;
      v2f z[R];
      if (PRE) {
#pragma unroll
        for (int u = 0; u < R; ++u) z[u] = zn[u];
        if (t + 1 < t1) load_frame_raw<R>(a.x_in + (long long)b * a.L, a.L, (long long)(t + 1) * hop - s.pad, lane, a.pad_mode, zn);
      }
      semi_frame<R, MODE, EVAL, PRE, TWO>(a, fi, b, t, hop, s.pad, k, lds_win, lds_tw1, tr, z, sd, so);
      if ((slot0 & 1) == 0) {                              // register pairs stay aligned in the ring
        v2f* r2 = reinterpret_cast<v2f*>(ring);
        const int h0 = slot0 >> 1;
#pragma unroll
        for (int u = 0; u < R; ++u) {
          int i = h0 + 64 * u + lane;
          if (i >= M) i -= M;
          r2[i] = r2[i] + z[u] * lds_win[64 * u + lane];
        }
      } else {
#pragma unroll
        for (int u = 0; u < R; ++u) {
          const v2f v = z[u] * lds_win[64 * u + lane];
          int i = slot0 + 128 * u + 2 * lane;
          if (i >= N) i -= N;
          const int i1 = i + 1 == N ? 0 : i + 1;
          ring[i] += v.x;
          ring[i1] += v.y;
        }
      }
      // the hop samples no later frame reaches (two at a time when every index involved is even)
      const long long p0 = (long long)t * hop;
      if (pairs) {
        for (int j = 2 * lane; j < hop; j += 128) {
          int i = slot0 + j;
          if (i >= N) i -= N;
          v2f* rp = reinterpret_cast<v2f*>(ring + i);
          const v2f v = *rp;
          *rp = v2f{0.0f, 0.0f};
          const long long p = p0 + j, n = p - s.pad;
          if (n >= 0 && n < a.L)
            *reinterpret_cast<v2f*>(xo + n) = p < raw_end ? v : env_apply(v, *reinterpret_cast<const v2f*>(env + n));
        }
      } else {
        for (int j = lane; j < hop; j += 64) {
          int i = slot0 + j;
          if (i >= N) i -= N;
          const float v = ring[i];
          ring[i] = 0.0f;
          const long long p = p0 + j, n = p - s.pad;
          if (n >= 0 && n < a.L) xo[n] = p < raw_end ? v : env_apply(v, env[n]);
        }
      }
      slot0 += hop;
      if (slot0 >= N) slot0 -= N;
    }
    // what is left in the ring: the end of the signal (last chunk), or the share of the next chunk's first samples
    const long long p0 = (long long)t1 * hop;
    if (c == a.nchunks - 1) {
      for (int j = lane; j < keep; j += 64) {
        int i = slot0 + j;
        if (i >= N) i -= N;
        const long long p = p0 + j, n = p - s.pad;
        if (n >= 0 && n < a.L) xo[n] = p < raw_end ? ring[i] : env_apply(ring[i], env[n]);
      }
    } else {
      float* tl = s.xtail + ((long long)b * a.nchunks + c) * keep;
      for (int j = lane; j < keep; j += 64) {
        int i = slot0 + j;
        if (i >= N) i -= N;
        tl[j] = ring[i];
      }
    }
  }
  if (EVAL) {
    const double d = wave_sum(sd), o = wave_sum(so);
    if (lane == 0 && w < a.n_waves) {
      a.partials[2 * w] = d;
      a.partials[2 * w + 1] = o;
    }
  }
}
template <int R, int MODE, bool EVAL>
__global__ __launch_bounds__(512, (SPECINV_HOP_R8_W2 && R == 8) ? 4 : 1) void k_hop(HopArgs s) {
  hop_body<R, MODE, EVAL, false>(s);
}
// ... of a two-sided spectrogram (semi_frame<..., TWO>: the mirror bins' state and target beside the lower half's)
template <int R, int MODE, bool EVAL>
__global__ __launch_bounds__(512, 1) void k_hop2(HopArgs s) {
  hop_body<R, MODE, EVAL, true>(s);
}

// ---- k_hop with Griffin-Lim's momentum carried as a signal (kernels_fast_td.h has the derivation) ---------------
// pre_t = STFT(z_t) + (-lr)^t c0, z_{t+1} = x_{t+1} - lr z_t.  The wave transforms z_t's frames (a.x_in), an evaluating launch
// x_t's as well (a.x2_in); the samples that become final after a frame go out as x_{t+1} (a.x2_out) and z_{t+1} (a.x_out), z_t's
// value at those positions re-read from L2.  Seam samples (first n_fft - hop of a later chunk) leave the kernel as undivided
// partial sums of x, as in k_hop; k_hop_tails_td finishes x and z there.
template <int R, bool EARLY, bool EVAL>
__device__ __forceinline__ void semi_frame_td(const FastArgs& a, long long fi, const LaneConst<R>& k, const v2f* lds_win,
                                              const v2f* lds_tw1, v2f* tr, v2f (&z)[R], const v2f (&xf)[EVAL ? R : 1],
                                              double& sd, double& so) {
  using G = Geo<R>;
  constexpr int H = G::H;
  const int lane = k.lane;
  const unsigned ulane = (unsigned)lane;
  const float half_scale = 0.5f * a.fwd_scale;
  v4f mm[H / 2];
  float mmid = 0.0f;
  v2f pmid = v2f{0.0f, 0.0f};
  {
    const v4f* min = a.m_pairs + fi * (H / 2 * 64);
#pragma unroll
    for (int j = 0; j < H / 2; ++j) mm[j] = ld_stream(&min[j * 64u + ulane]);
    if (lane == 0) {
      mmid = a.m_mid[fi];
      if (EARLY) pmid = a.Pmid_in[fi];
    }
  }
  if (EVAL) {
    // |STFT(x_t)| against the target (methods.py:242)
    v2f y[R];
#pragma unroll
    for (int u = 0; u < R; ++u) y[u] = xf[u] * lds_win[64 * u + lane];
    fft_forward<R>(y, k, lds_tw1, tr);
    v2f rc[H];
#pragma unroll
    for (int m = H; m < R; ++m) {
      const v2f got = shfl2(y[m], k.partner);
      const v2f own = y[(m + 1) % R];
      rc[m - H] = v2f{lane == 0 ? own.x : got.x, lane == 0 ? own.y : got.y};
    }
    float fd = 0.0f, fo = 0.0f;
#pragma unroll
    for (int j = 0; j < H; ++j) {
      const v2f wk = pair_twiddle<R>(k.wn, j);
      v2f xk, xm;
      td_split<R>(y[j], rc[R - 1 - j - H], wk, half_scale, xk, xm);
      const float mk = (j & 1) ? mm[j / 2].z : mm[j / 2].x;
      const float mq = (j & 1) ? mm[j / 2].w : mm[j / 2].y;
      const float ok = fast_abs(xk), om = fast_abs(xm);
      const float dk = ok - mk, dm = om - mq;
      fd = fmaf(dk, dk, fmaf(dm, dm, fd));
      fo = fmaf(ok, ok, fmaf(om, om, fo));
    }
    if (lane == 0) {
      const float o = fast_abs(y[H] * v2f{a.fwd_scale, -a.fwd_scale});
      const float d = o - mmid;
      fd = fmaf(d, d, fd);
      fo = fmaf(o, o, fo);
    }
    sd += (double)fd;
    so += (double)fo;
  }
  v4f pp[EARLY ? H : 1];
  if (EARLY) {
    const v4f* pin = a.P_in + fi * (H * 64);
#pragma unroll
    for (int j = 0; j < H; ++j) pp[j] = ld_stream(&pin[j * 64u + ulane]);
  }
#pragma unroll
  for (int u = 0; u < R; ++u) z[u] = z[u] * lds_win[64 * u + lane];
  fft_forward<R>(z, k, lds_tw1, tr);
  v2f rc[H];
#pragma unroll
  for (int m = H; m < R; ++m) {
    const v2f got = shfl2(z[m], k.partner);
    const v2f own = z[(m + 1) % R];
    rc[m - H] = v2f{lane == 0 ? own.x : got.x, lane == 0 ? own.y : got.y};
  }
  v2f back[H];
#pragma unroll
  for (int j = 0; j < H; ++j) {
    const v2f wk = pair_twiddle<R>(k.wn, j);
    v2f sk, sm;
    td_split<R>(z[j], rc[R - 1 - j - H], wk, half_scale, sk, sm);
    if (EARLY) {
      sk = v2f{fmaf(a.tds, pp[j].x, sk.x), fmaf(a.tds, pp[j].y, sk.y)};
      sm = v2f{fmaf(a.tds, pp[j].z, sm.x), fmaf(a.tds, pp[j].w, sm.y)};
    }
    const float mk = (j & 1) ? mm[j / 2].z : mm[j / 2].x;
    const float mq = (j & 1) ? mm[j / 2].w : mm[j / 2].y;
#if SPECINV_IEEE
    // (s m) r with r the correctly rounded 1 / |s|: the reference's operation order (ref_rcp_abs2, fast_core.h)
    const v2f rr = ref_rcp_abs2(v2f{ref_norm2(sk), ref_norm2(sm)});
    const v2f mp = v2f{mk, mq};
    v2f ak = scale_lo(scale_lo(sk, mp), rr) * a.inv_scale;
    v2f am = scale_hi(scale_hi(sm, mp), rr) * a.inv_scale;
#elif SPECINV_RSQ
    const v2f mi = (v2f{mk, mq} * v2f{proj_rsq(sk), proj_rsq(sm)}) * a.inv_scale;
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
  v2f zmid;
  {
    v2f smid = z[H] * v2f{a.fwd_scale, -a.fwd_scale};
    if (EARLY) smid = v2f{fmaf(a.tds, pmid.x, smid.x), fmaf(a.tds, pmid.y, smid.y)};
#if SPECINV_IEEE
    const v2f am = ((smid * mmid) * ref_rcp_abs(ref_norm2(smid))) * a.inv_scale;
#elif SPECINV_RSQ
    const v2f am = smid * ((mmid * proj_rsq(smid)) * a.inv_scale);
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
  fft_inverse<R>(z, k, lds_tw1, tr);
}

template <int R, bool EARLY, bool EVAL>
__global__ __launch_bounds__(512, 1) void k_hop_td(HopArgs s) {
  using G = Geo<R>;
  constexpr int M = G::M, N = G::N;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  v2f* lds_win = reinterpret_cast<v2f*>(smem);
  v2f* lds_tw1 = lds_win + M;
  const FastArgs& a = s.f;
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwib = __builtin_amdgcn_readfirstlane(blockDim.x >> 6);
  v2f* tr = lds_tw1 + (R - 1) * 64 + wib * G::TR;
  float* ring = reinterpret_cast<float*>(lds_tw1 + (R - 1) * 64 + nwib * G::TR) + wib * N;
  xform_tables<R>(a.window, lds_win, lds_tw1);
  const LaneConst<R> k = lane_consts<R>();
  const int lane = k.lane;
  const int w = __builtin_amdgcn_readfirstlane(blockIdx.x * nwib + wib);
  double sd = 0.0, so = 0.0;
  if (w < a.n_waves) {
    const int b = w / a.nchunks, c = w - b * a.nchunks;
    const int t0 = hop_chunk_begin(c, a.T, a.nchunks), t1 = hop_chunk_begin(c + 1, a.T, a.nchunks);
    const int hop = s.hop, keep = N - hop;
    const float* env = s.env;
    const float nlr = -a.coef;
    const float* zrow = a.x_in + (long long)b * a.L;
    const float* xrow = a.x2_in + (long long)b * a.L;
    float* zo = a.x_out + (long long)b * a.L;
    float* xo = a.x2_out + (long long)b * a.L;
#pragma unroll
    for (int u = 0; u < R; ++u) reinterpret_cast<v2f*>(ring)[64 * u + lane] = v2f{0.0f, 0.0f};
    const long long raw_end = c > 0 ? (long long)t0 * hop + keep : -1;
    int slot0 = (int)(((long long)t0 * hop) % N);
    const bool pairs = ((hop | s.pad) & 1) == 0 && (a.L & 1) == 0;     // (as in k_hop)
    v2f zn[R];
    load_frame_raw<R>(zrow, a.L, (long long)t0 * hop - s.pad, lane, a.pad_mode, zn);
    for (int t = t0; t < t1; ++t) {
      const long long fi = (long long)b * a.T + t;
      v2f z[R], xf[EVAL ? R : 1];
#pragma unroll
      for (int u = 0; u < R; ++u) z[u] = zn[u];
      if constexpr (EVAL) load_frame_raw<R>(xrow, a.L, (long long)t * hop - s.pad, lane, a.pad_mode, xf);
      if (t + 1 < t1) load_frame_raw<R>(zrow, a.L, (long long)(t + 1) * hop - s.pad, lane, a.pad_mode, zn);
      semi_frame_td<R, EARLY, EVAL>(a, fi, k, lds_win, lds_tw1, tr, z, xf, sd, so);
      if ((slot0 & 1) == 0) {                              // register pairs stay aligned in the ring
        v2f* r2 = reinterpret_cast<v2f*>(ring);
        const int h0 = slot0 >> 1;
#pragma unroll
        for (int u = 0; u < R; ++u) {
          int i = h0 + 64 * u + lane;
          if (i >= M) i -= M;
          r2[i] = r2[i] + z[u] * lds_win[64 * u + lane];
        }
      } else {
#pragma unroll
        for (int u = 0; u < R; ++u) {
          const v2f v = z[u] * lds_win[64 * u + lane];
          int i = slot0 + 128 * u + 2 * lane;
          if (i >= N) i -= N;
          const int i1 = i + 1 == N ? 0 : i + 1;
          ring[i] += v.x;
          ring[i1] += v.y;
        }
      }
      // the hop samples no later frame reaches: x_{t+1} and z_{t+1} = x_{t+1} - lr z_t
      const long long p0 = (long long)t * hop;
      if (pairs) {
        for (int j = 2 * lane; j < hop; j += 128) {
          int i = slot0 + j;
          if (i >= N) i -= N;
          v2f* rp = reinterpret_cast<v2f*>(ring + i);
          const v2f v = *rp;
          *rp = v2f{0.0f, 0.0f};
          const long long p = p0 + j, n = p - s.pad;
          if (n >= 0 && n < a.L) {
            if (p < raw_end) {
              *reinterpret_cast<v2f*>(xo + n) = v;         // undivided partial sums: k_hop_tails_td finishes x and z
            } else {
              const v2f xv = env_apply(v, *reinterpret_cast<const v2f*>(env + n));
              const v2f zv = *reinterpret_cast<const v2f*>(zrow + n);
              if (s.write_x) *reinterpret_cast<v2f*>(xo + n) = xv;
              *reinterpret_cast<v2f*>(zo + n) = v2f{fmaf(nlr, zv.x, xv.x), fmaf(nlr, zv.y, xv.y)};
            }
          }
        }
      } else {
        for (int j = lane; j < hop; j += 64) {
          int i = slot0 + j;
          if (i >= N) i -= N;
          const float v = ring[i];
          ring[i] = 0.0f;
          const long long p = p0 + j, n = p - s.pad;
          if (n >= 0 && n < a.L) {
            if (p < raw_end) {
              xo[n] = v;                                   // undivided partial sum: k_hop_tails_td finishes x and z
            } else {
              const float xv = env_apply(v, env[n]);
              if (s.write_x) xo[n] = xv;
              zo[n] = fmaf(nlr, zrow[n], xv);
            }
          }
        }
      }
      slot0 += hop;
      if (slot0 >= N) slot0 -= N;
    }
    const long long p0 = (long long)t1 * hop;
    if (c == a.nchunks - 1) {
      for (int j = lane; j < keep; j += 64) {
        int i = slot0 + j;
        if (i >= N) i -= N;
        const long long p = p0 + j, n = p - s.pad;
        if (n >= 0 && n < a.L) {
          if (p < raw_end) {
            xo[n] = ring[i];
          } else {
            const float xv = env_apply(ring[i], env[n]);
            if (s.write_x) xo[n] = xv;
            zo[n] = fmaf(nlr, zrow[n], xv);
          }
        }
      }
    } else {
      float* tl = s.xtail + ((long long)b * a.nchunks + c) * keep;
      for (int j = lane; j < keep; j += 64) {
        int i = slot0 + j;
        if (i >= N) i -= N;
        tl[j] = ring[i];
      }
    }
  }
  if (EVAL) {
    const double d = wave_sum(sd), o = wave_sum(so);
    if (lane == 0 && w < a.n_waves) {
      a.partials[2 * w] = d;
      a.partials[2 * w + 1] = o;
    }
  }
}


// ---- adjoint of the STFT without the frame round trip: inverse frames + plain overlap-add over the padded signal --------
// k_fast_inverse_frames' body in k_hop's chunk / ring structure (no envelope): samples inside the signal go to `out`
// (B, len), the `pad` samples on either side of it to `margins` (B, 2, pad) for the fold of the padding, chunk seams
// through `xtail` + k_hop_tails_raw.

template <int R>
__global__ __launch_bounds__(512, 1) void k_hop_inverse(HopInvArgs a) {
  using G = Geo<R>;
  constexpr int H = G::H, M = G::M, N = G::N;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  v2f* lds_win = reinterpret_cast<v2f*>(smem);
  v2f* lds_tw1 = lds_win + M;
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwib = __builtin_amdgcn_readfirstlane(blockDim.x >> 6);
  v2f* tr = lds_tw1 + (R - 1) * 64 + wib * G::TR;
  float* ring = reinterpret_cast<float*>(lds_tw1 + (R - 1) * 64 + nwib * G::TR) + wib * N;
  xform_tables<R>(a.window, lds_win, lds_tw1);
  const LaneConst<R> k = lane_consts<R>();
  const int lane = k.lane;
  const int w = __builtin_amdgcn_readfirstlane(blockIdx.x * nwib + wib);
  if (w >= a.n_waves) return;
  const int b = w / a.nchunks, c = w - b * a.nchunks;
  const int t0 = hop_chunk_begin(c, a.T, a.nchunks), t1 = hop_chunk_begin(c + 1, a.T, a.nchunks);
  const int hop = a.hop, keep = N - hop;
  float* xo = a.out + (long long)b * a.len;
  float* mg = a.margins + (long long)b * 2 * a.pad;
  auto emit = [&](long long p, float v) {
    const long long n = p - a.pad;
    if (n >= 0 && n < a.len) xo[n] = v;
    else if (n < 0) mg[p] = v;
    else if (n - a.len < a.pad) mg[a.pad + (n - a.len)] = v;
  };
#pragma unroll
  for (int u = 0; u < R; ++u) reinterpret_cast<v2f*>(ring)[64 * u + lane] = v2f{0.0f, 0.0f};
  int slot0 = (int)(((long long)t0 * hop) % N);
  for (int t = t0; t < t1; ++t) {
    const long long fi = (long long)b * a.T + t;
    const v2f* in = a.spec + fi * (M + 1);
    v2f z[R], back[H];
#pragma unroll
    for (int j = 0; j < H; ++j) {
      const v2f wk = pair_twiddle<R>(k.wn, j);
      const int kk = lane + 64 * j;
      v2f ak = in[kk] * a.scale, am = in[M - kk] * a.scale;
      if (j == 0 && lane == 0) {
        ak.y = 0.0f;
        am.y = 0.0f;
      }
      const v2f e2i = add_conj(ak, am);
      const v2f o2i = cmulc(sub_conj(ak, am), wk);
      z[j] = add_i(e2i, o2i);
      back[j] = conj_sub_i(e2i, o2i);
    }
    v2f zmid = v2f{0.0f, 0.0f};
    if (lane == 0) zmid = in[M / 2] * v2f{2.0f * a.scale, -2.0f * a.scale};
#pragma unroll
    for (int m = H; m < R; ++m) {
      const v2f got = shfl2(back[R - 1 - m], k.partner);
      const v2f l0 = (m == H) ? zmid : back[(R - m) % H];
      z[m] = v2f{lane == 0 ? l0.x : got.x, lane == 0 ? l0.y : got.y};
    }
    fft_inverse<R>(z, k, lds_tw1, tr);
    if ((slot0 & 1) == 0) {
      v2f* r2 = reinterpret_cast<v2f*>(ring);
      const int h0 = slot0 >> 1;
#pragma unroll
      for (int u = 0; u < R; ++u) {
        int i = h0 + 64 * u + lane;
        if (i >= M) i -= M;
        r2[i] = r2[i] + z[u] * lds_win[64 * u + lane];
      }
    } else {
#pragma unroll
      for (int u = 0; u < R; ++u) {
        const v2f v = z[u] * lds_win[64 * u + lane];
        int i = slot0 + 128 * u + 2 * lane;
        if (i >= N) i -= N;
        const int i1 = i + 1 == N ? 0 : i + 1;
        ring[i] += v.x;
        ring[i1] += v.y;
      }
    }
    const long long p0 = (long long)t * hop;
    for (int j = lane; j < hop; j += 64) {
      int i = slot0 + j;
      if (i >= N) i -= N;
      const float v = ring[i];
      ring[i] = 0.0f;
      emit(p0 + j, v);
    }
    slot0 += hop;
    if (slot0 >= N) slot0 -= N;
  }
  const long long p0 = (long long)t1 * hop;
  if (c == a.nchunks - 1) {
    for (int j = lane; j < keep; j += 64) {
      int i = slot0 + j;
      if (i >= N) i -= N;
      emit(p0 + j, ring[i]);
    }
  } else {
    float* tl = a.xtail + ((long long)b * a.nchunks + c) * keep;
    for (int j = lane; j < keep; j += 64) {
      int i = slot0 + j;
      if (i >= N) i -= N;
      tl[j] = ring[i];
    }
  }
}



}  // namespace SI_FAST_NS (fast, or fast_approx in the approximate-projection units)
}  // namespace specinv
