// RTISI-LA on the wave-level FFT: the host side (which instantiation covers a plan, the launch) and the ring layout kernel.
// The kernel itself is kernels_rtisi_fast.h.
#pragma once
#include "fast_state.h"
#include "rtisi_fast_args.h"

namespace specinv {
namespace fast {

// (B, k, F) frame-major magnitudes of a push -> pair layout at ring rows (t0 + j) % mag_ring of every item
template <int R>
__global__ void k_mag_to_pairs_ring(const float* __restrict__ mag, v4f* __restrict__ pairs, float* __restrict__ mid, int k,
                                    int mag_ring, long long t0, long long total) {
  using G = Geo<R>;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // (b, j, c, lane)
  if (i >= total) return;
  const int lane = i & 63;
  const int c = (i >> 6) % (G::H / 2);
  const long long fj = i / (64 * (G::H / 2));
  const long long b = fj / k, j = fj - b * k;
  const float* sp = mag + fj * (G::M + 1);
  const long long row = b * mag_ring + (t0 + j) % mag_ring;
  const int k0 = lane + 64 * (2 * c), k1 = lane + 64 * (2 * c + 1);
  pairs[(row * (G::H / 2) + c) * 64 + lane] = v4f{sp[k0], sp[G::M - k0], sp[k1], sp[G::M - k1]};
  if (lane == 0 && c == 0) mid[row] = sp[G::M / 2];
}

}  // namespace fast

// which instantiation (if any) covers the plan at this look-ahead: kernel, its LDS bytes and geometry
struct RtisiFastPick {
  const void* fn = nullptr;
  size_t lds = 0;
  int R = 0, OV = 0, threads = 0;
};

template <typename P>
RtisiFastPick rtisi_fast_pick(P& pl, int la) {
  RtisiFastPick k;
  const auto& cfg = pl.cfg;
  if (cfg.dtype != SPECINV_F32 || !cfg.onesided) return k;
  if (cfg.n_fft != 2048 && cfg.n_fft != 1024 && cfg.n_fft != 512) return k;   // (4096: the per-wave tables would not
                                                                              // fit the registers)
  const int R = cfg.n_fft / 128;
  int OV = 0;
  for (int o : {2, 4, 8})
    if (cfg.hop_length * o == cfg.n_fft && R % o == 0) OV = o;
  if (OV == 0 || la > 7 || pl.force_generic) return k;
  if (const char* e = getenv("SPECINV_DISABLE_FAST")) {
    if (e[0] == '1') return k;
  }
  size_t lds = 0;
  const void* fn = nullptr;
  const bool small = 64 * (la + 1) <= 256;
  SPECINV_R_SWITCH(R, if constexpr (RR <= 16) {
    if constexpr (RR % 8 == 0) {
      if (OV == 8) {
        lds = fast::RtisiGeo<RR, 8>::lds_bytes(la);
        fn = small ? (const void*)fast::k_rtisi_fast<RR, 256, 8> : (const void*)fast::k_rtisi_fast<RR, 512, 8>;
      }
    }
    if (OV == 4) {
      lds = fast::RtisiGeo<RR, 4>::lds_bytes(la);
      fn = small ? (const void*)fast::k_rtisi_fast<RR, 256, 4> : (const void*)fast::k_rtisi_fast<RR, 512, 4>;
    }
    if (OV == 2) {
      lds = fast::RtisiGeo<RR, 2>::lds_bytes(la);
      fn = small ? (const void*)fast::k_rtisi_fast<RR, 256, 2> : (const void*)fast::k_rtisi_fast<RR, 512, 2>;
    }
  });
  if (fn == nullptr || lds > 160 * 1024 - 1024) return k;
  k.fn = fn;
  k.lds = lds;
  k.R = R;
  k.OV = OV;
  k.threads = 64 * (la + 1);
  return k;
}

// `*used` stays false (and nothing is launched) when the configuration is not covered
template <typename P>
int rtisi_fast_launch(P& pl, const float* mag_user, int la, int asym, int max_iter, double alpha, float* x_out,
                      const float* d_wsyn, const float* d_a1, const float* d_a2, bool* used) {
  *used = false;
  const RtisiFastPick pick = rtisi_fast_pick(pl, la);
  if (pick.fn == nullptr) return SPECINV_OK;
  const int R = pick.R;
  const size_t lds = pick.lds;
  using v4f = fast::v4f;
  const long long nf = (long long)pl.B() * pl.Tn();
  const int H = R / 2;
  SI_TRY(pl.mag.reserve(pl.nspec() * sizeof(float)));
  SI_TRY((pl.template transpose<float>(mag_user, pl.mag.template as<float>(), pl.n_freq, pl.Tn())));
  SI_TRY(pl.fast.mpairs.reserve((size_t)nf * (H / 2) * 64 * sizeof(v4f)));
  SI_TRY(pl.fast.mmid.reserve(nf * sizeof(float)));
  const long long nm = nf * (H / 2) * 64;
  SPECINV_R_SWITCH(R, hipLaunchKernelGGL((fast::k_mag_to_pairs<RR>), dim3((unsigned)ceil_div(nm, 256)), dim3(256), 0, pl.stream,
                                         pl.mag.template as<float>(), pl.fast.mpairs.template as<v4f>(),
                                         pl.fast.mmid.template as<float>(), nf));
  SI_HIP(hipGetLastError());
  SI_TRY(pl.frames_needed());
  fast::RtisiFastArgs a{};
  a.m_pairs = pl.fast.mpairs.template as<v4f>();
  a.m_mid = pl.fast.mmid.template as<float>();
  a.frames_out = pl.frames.template as<float>();
  a.window = pl.window.template as<float>();
  a.wsyn = d_wsyn;
  a.asym1 = d_a1;
  a.asym2 = d_a2;
  a.T = pl.Tn();
  a.la = la;
  a.max_iter = max_iter;
  a.asym = asym ? 1 : 0;
  a.lr = (float)(alpha / (1.0 + alpha));
  a.fwd_scale = pl.fc.fwd_scale;
  a.inv_scale = pl.fc.inv_scale;
  a.i_begin = 0;
  a.i_end = pl.Tn() + la;
  a.resume = 0;
  a.n_valid = pl.Tn();
  a.mag_ring = 0;
  a.out_ring = 0;
  a.state = nullptr;
  const int threads = pick.threads;
  const void* fn = pick.fn;
  SI_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  void* kargs[] = {&a};
  SI_HIP(hipLaunchKernel(fn, dim3(pl.B()), dim3(threads), kargs, lds, pl.stream));
  SI_HIP(hipGetLastError());
  SI_TRY(pl.launch_ola(pl.frames.template as<float>(), x_out, true));
  *used = true;
  return SPECINV_OK;
}

}  // namespace specinv
