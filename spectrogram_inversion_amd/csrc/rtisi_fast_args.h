// RTISI-LA on the wave-level FFT: argument record and LDS geometry of k_rtisi_fast (kernels_rtisi_fast.h), and its declaration.
// No host code here: a host-side mention of k_rtisi_fast<...> ahead of the definition would instantiate the declaration
// without the definition's __launch_bounds__ (the kernel then compiles for 128 registers and spills).
#pragma once
#include "fast_core.h"

namespace specinv {
namespace fast {

struct RtisiFastArgs {
  const v4f* m_pairs;   // [B*T][H/2][64] target magnitude, pair order
  const float* m_mid;   // [B*T]
  float* frames_out;    // (B, T, N) committed frames times the synthesis window
  const float* window;  // N   analysis/synthesis window w
  const float* wsyn;    // N   w * hop / (w.w)
  const float* asym1;   // N
  const float* asym2;   // N
  int T, la, max_iter, asym;
  float lr, fwd_scale, inv_scale;
  // step range [i_begin, i_end) of this launch; a stream resumes from `state` (what the previous launch left: the
  // frame ring and each wave's pre_spec registers) and addresses targets / committed frames as rings of frames
  int i_begin, i_end, resume, n_valid;
  int mag_ring;          // 0: m_pairs is [B*T]; else [B*mag_ring], frame t at t % mag_ring
  int out_ring;          // 0: frames_out is (B, T, N); else (B, out_ring, N)
  float* state;          // NULL, or per item: ring | per wave (pre pairs, pre mid; the rest of the record is unused)
};

template <int R>
constexpr size_t rtisi_state_v2f(int nslots, int waves) {
  return (size_t)nslots * Geo<R>::M + (size_t)waves * (2 * (Geo<R>::H * 2 * 64) + 2 * 64);
}

template <int R, int OV = 4>
struct RtisiGeo {
  using G = Geo<R>;
  static constexpr int K = OV - 1;   // kept frames: (n_fft - 1) / hop with hop = n_fft / OV
  // LDS (v2f units): ring | tw1 | synthesis window, analysis window, the two asymmetric windows | one hop-block of
  // zeros | per-wave transpose scratch, which doubles as the pre_spec exchange between two steps (pairs as 2 x v2f + mid)
  static_assert(G::TR >= G::H * 2 * 64 + 64, "the pre_spec exchange must fit the transpose scratch");
  static constexpr size_t lds_bytes(int la) {
    const size_t waves = la + 1, nslots = K + la + 1;
    return sizeof(v2f) * (nslots * G::M + (R - 1) * 64 + 4 * G::M + (R / OV) * 64 + waves * G::TR);
  }
};

// the persistent kernel: defined in kernels_rtisi_fast.h, compiled in tu_rtisi_fast.hip
template <int R, int MAXT, int OV>
__global__ __launch_bounds__(MAXT, 1) void k_rtisi_fast(RtisiFastArgs a);

}  // namespace fast
}  // namespace specinv
