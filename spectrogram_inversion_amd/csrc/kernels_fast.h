// Fused gfx950 fast path (placeholder until the wave-per-frame kernels land).
#pragma once
#include <vector>

#include "common.h"

namespace specinv {

template <typename T>
struct FastState {
  bool supported = false;
  int setup(const specinv_stft_cfg&, const std::vector<T>&, int64_t, int) {
    supported = false;
    return SPECINV_OK;
  }
  template <typename P>
  int gla_begin(P&) { return fail(SPECINV_EUNSUPPORTED, "fast path not built"); }
  template <typename P>
  int admm_begin(P&) { return fail(SPECINV_EUNSUPPORTED, "fast path not built"); }
  template <typename P>
  int iterate(P&, int, bool) { return fail(SPECINV_EUNSUPPORTED, "fast path not built"); }
  template <typename P>
  int get_wave(P&, T*) { return fail(SPECINV_EUNSUPPORTED, "fast path not built"); }
  template <typename P>
  int get_state_spec(P&, int, cplx<T>*) { return fail(SPECINV_EUNSUPPORTED, "fast path not built"); }
};

}  // namespace specinv
