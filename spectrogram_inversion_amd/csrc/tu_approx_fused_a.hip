// k_fused4 and k_fused<R, OV> / k_fused_istft at n_fft 512 / 1024 with the approximate projection (opt-in).
// The second copy of these kernels (fast_core.h, SPECINV_IEEE=0): m * v_rsq_f32(|s|^2 + 1e-32) in the projection and a multiplication
// by 1 / envelope instead of the reference's operation order with correctly rounded factors (the default build), in namespace
// specinv::fast_approx.  The host side takes the kernels' addresses from the table function below (specinv_plan_set_exact(plan, 0)).
#define SPECINV_IEEE 0
#define SI_FAST_NS fast_approx
#include "kernels_fused.h"

namespace specinv {
namespace fast_approx {

template __global__ void k_fused4<8, MODE_GLA, false>(FastArgs);
template __global__ void k_fused4<8, MODE_GLA, true>(FastArgs);
template __global__ void k_fused4<8, MODE_ADMM, false>(FastArgs);
template __global__ void k_fused4<8, MODE_ADMM, true>(FastArgs);
template __global__ void k_fused4<16, MODE_GLA, false>(FastArgs);
template __global__ void k_fused4<16, MODE_GLA, true>(FastArgs);
template __global__ void k_fused4<16, MODE_ADMM, false>(FastArgs);
template __global__ void k_fused4<16, MODE_ADMM, true>(FastArgs);
template __global__ void k_fused<4, 4, MODE_GLA, false>(FastArgs);
template __global__ void k_fused<4, 4, MODE_GLA, true>(FastArgs);
template __global__ void k_fused<4, 4, MODE_ADMM, false>(FastArgs);
template __global__ void k_fused<4, 4, MODE_ADMM, true>(FastArgs);
template __global__ void k_fused_istft<4, 4>(FastArgs);
template __global__ void k_fused<4, 2, MODE_GLA, false>(FastArgs);
template __global__ void k_fused<4, 2, MODE_GLA, true>(FastArgs);
template __global__ void k_fused<4, 2, MODE_ADMM, false>(FastArgs);
template __global__ void k_fused<4, 2, MODE_ADMM, true>(FastArgs);
template __global__ void k_fused_istft<4, 2>(FastArgs);
template __global__ void k_fused<8, 8, MODE_GLA, false>(FastArgs);
template __global__ void k_fused<8, 8, MODE_GLA, true>(FastArgs);
template __global__ void k_fused<8, 8, MODE_ADMM, false>(FastArgs);
template __global__ void k_fused<8, 8, MODE_ADMM, true>(FastArgs);
template __global__ void k_fused_istft<8, 8>(FastArgs);
template __global__ void k_fused<8, 4, MODE_GLA, false>(FastArgs);
template __global__ void k_fused<8, 4, MODE_GLA, true>(FastArgs);
template __global__ void k_fused<8, 4, MODE_ADMM, false>(FastArgs);
template __global__ void k_fused<8, 4, MODE_ADMM, true>(FastArgs);
template __global__ void k_fused_istft<8, 4>(FastArgs);
template __global__ void k_fused<8, 2, MODE_GLA, false>(FastArgs);
template __global__ void k_fused<8, 2, MODE_GLA, true>(FastArgs);
template __global__ void k_fused<8, 2, MODE_ADMM, false>(FastArgs);
template __global__ void k_fused<8, 2, MODE_ADMM, true>(FastArgs);
template __global__ void k_fused_istft<8, 2>(FastArgs);

}  // namespace fast_approx
}  // namespace specinv

extern "C" __attribute__((visibility("hidden"))) const void* specinv_approx_fused_a(int R, int OV, int mode /* 0 GLA, 1 ADMM, 2 initial ISTFT */, int eval, int tuned4) {
  using namespace specinv::fast_approx;
  if (tuned4 && R == 8 && OV == 4 && mode == 0 && eval == 0) return (const void*)k_fused4<8, MODE_GLA, false>;
  if (tuned4 && R == 8 && OV == 4 && mode == 0 && eval == 1) return (const void*)k_fused4<8, MODE_GLA, true>;
  if (tuned4 && R == 8 && OV == 4 && mode == 1 && eval == 0) return (const void*)k_fused4<8, MODE_ADMM, false>;
  if (tuned4 && R == 8 && OV == 4 && mode == 1 && eval == 1) return (const void*)k_fused4<8, MODE_ADMM, true>;
  if (tuned4 && R == 16 && OV == 4 && mode == 0 && eval == 0) return (const void*)k_fused4<16, MODE_GLA, false>;
  if (tuned4 && R == 16 && OV == 4 && mode == 0 && eval == 1) return (const void*)k_fused4<16, MODE_GLA, true>;
  if (tuned4 && R == 16 && OV == 4 && mode == 1 && eval == 0) return (const void*)k_fused4<16, MODE_ADMM, false>;
  if (tuned4 && R == 16 && OV == 4 && mode == 1 && eval == 1) return (const void*)k_fused4<16, MODE_ADMM, true>;
  if (!tuned4 && R == 4 && OV == 4 && mode == 0 && eval == 0) return (const void*)k_fused<4, 4, MODE_GLA, false>;
  if (!tuned4 && R == 4 && OV == 4 && mode == 0 && eval == 1) return (const void*)k_fused<4, 4, MODE_GLA, true>;
  if (!tuned4 && R == 4 && OV == 4 && mode == 1 && eval == 0) return (const void*)k_fused<4, 4, MODE_ADMM, false>;
  if (!tuned4 && R == 4 && OV == 4 && mode == 1 && eval == 1) return (const void*)k_fused<4, 4, MODE_ADMM, true>;
  if (mode == 2 && R == 4 && OV == 4) return (const void*)k_fused_istft<4, 4>;
  if (!tuned4 && R == 4 && OV == 2 && mode == 0 && eval == 0) return (const void*)k_fused<4, 2, MODE_GLA, false>;
  if (!tuned4 && R == 4 && OV == 2 && mode == 0 && eval == 1) return (const void*)k_fused<4, 2, MODE_GLA, true>;
  if (!tuned4 && R == 4 && OV == 2 && mode == 1 && eval == 0) return (const void*)k_fused<4, 2, MODE_ADMM, false>;
  if (!tuned4 && R == 4 && OV == 2 && mode == 1 && eval == 1) return (const void*)k_fused<4, 2, MODE_ADMM, true>;
  if (mode == 2 && R == 4 && OV == 2) return (const void*)k_fused_istft<4, 2>;
  if (!tuned4 && R == 8 && OV == 8 && mode == 0 && eval == 0) return (const void*)k_fused<8, 8, MODE_GLA, false>;
  if (!tuned4 && R == 8 && OV == 8 && mode == 0 && eval == 1) return (const void*)k_fused<8, 8, MODE_GLA, true>;
  if (!tuned4 && R == 8 && OV == 8 && mode == 1 && eval == 0) return (const void*)k_fused<8, 8, MODE_ADMM, false>;
  if (!tuned4 && R == 8 && OV == 8 && mode == 1 && eval == 1) return (const void*)k_fused<8, 8, MODE_ADMM, true>;
  if (mode == 2 && R == 8 && OV == 8) return (const void*)k_fused_istft<8, 8>;
  if (!tuned4 && R == 8 && OV == 4 && mode == 0 && eval == 0) return (const void*)k_fused<8, 4, MODE_GLA, false>;
  if (!tuned4 && R == 8 && OV == 4 && mode == 0 && eval == 1) return (const void*)k_fused<8, 4, MODE_GLA, true>;
  if (!tuned4 && R == 8 && OV == 4 && mode == 1 && eval == 0) return (const void*)k_fused<8, 4, MODE_ADMM, false>;
  if (!tuned4 && R == 8 && OV == 4 && mode == 1 && eval == 1) return (const void*)k_fused<8, 4, MODE_ADMM, true>;
  if (mode == 2 && R == 8 && OV == 4) return (const void*)k_fused_istft<8, 4>;
  if (!tuned4 && R == 8 && OV == 2 && mode == 0 && eval == 0) return (const void*)k_fused<8, 2, MODE_GLA, false>;
  if (!tuned4 && R == 8 && OV == 2 && mode == 0 && eval == 1) return (const void*)k_fused<8, 2, MODE_GLA, true>;
  if (!tuned4 && R == 8 && OV == 2 && mode == 1 && eval == 0) return (const void*)k_fused<8, 2, MODE_ADMM, false>;
  if (!tuned4 && R == 8 && OV == 2 && mode == 1 && eval == 1) return (const void*)k_fused<8, 2, MODE_ADMM, true>;
  if (mode == 2 && R == 8 && OV == 2) return (const void*)k_fused_istft<8, 2>;
  return nullptr;
}
