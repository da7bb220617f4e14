// k_fused4_td / k_fused_td<R, OV> with the approximate projection (opt-in).
// The second copy of these kernels (fast_core.h, SPECINV_IEEE=0): m * v_rsq_f32(|s|^2 + 1e-32) in the projection and a multiplication
// by 1 / envelope instead of the reference's operation order with correctly rounded factors (the default build), in namespace
// specinv::fast_approx.  The host side takes the kernels' addresses from the table function below (specinv_plan_set_exact(plan, 0)).
#define SPECINV_IEEE 0
#define SI_FAST_NS fast_approx
#include "kernels_fast_td.h"

namespace specinv {
namespace fast_approx {

template __global__ void k_fused4_td<8, false, false>(FastArgs);
template __global__ void k_fused4_td<8, false, true>(FastArgs);
template __global__ void k_fused4_td<8, true, false>(FastArgs);
template __global__ void k_fused4_td<8, true, true>(FastArgs);
template __global__ void k_fused4_td<16, false, false>(FastArgs);
template __global__ void k_fused4_td<16, false, true>(FastArgs);
template __global__ void k_fused4_td<16, true, false>(FastArgs);
template __global__ void k_fused4_td<16, true, true>(FastArgs);
template __global__ void k_fused_td<4, 4, false, false>(FastArgs);
template __global__ void k_fused_td<4, 4, false, true>(FastArgs);
template __global__ void k_fused_td<4, 4, true, false>(FastArgs);
template __global__ void k_fused_td<4, 4, true, true>(FastArgs);
template __global__ void k_fused_td<4, 2, false, false>(FastArgs);
template __global__ void k_fused_td<4, 2, false, true>(FastArgs);
template __global__ void k_fused_td<4, 2, true, false>(FastArgs);
template __global__ void k_fused_td<4, 2, true, true>(FastArgs);
template __global__ void k_fused_td<8, 8, false, false>(FastArgs);
template __global__ void k_fused_td<8, 8, false, true>(FastArgs);
template __global__ void k_fused_td<8, 8, true, false>(FastArgs);
template __global__ void k_fused_td<8, 8, true, true>(FastArgs);
template __global__ void k_fused_td<8, 2, false, false>(FastArgs);
template __global__ void k_fused_td<8, 2, false, true>(FastArgs);
template __global__ void k_fused_td<8, 2, true, false>(FastArgs);
template __global__ void k_fused_td<8, 2, true, true>(FastArgs);
template __global__ void k_fused_td<16, 8, false, false>(FastArgs);
template __global__ void k_fused_td<16, 8, false, true>(FastArgs);
template __global__ void k_fused_td<16, 8, true, false>(FastArgs);
template __global__ void k_fused_td<16, 8, true, true>(FastArgs);
template __global__ void k_fused_td<16, 2, false, false>(FastArgs);
template __global__ void k_fused_td<16, 2, false, true>(FastArgs);
template __global__ void k_fused_td<16, 2, true, false>(FastArgs);
template __global__ void k_fused_td<16, 2, true, true>(FastArgs);

}  // namespace fast_approx
}  // namespace specinv

extern "C" __attribute__((visibility("hidden"))) const void* specinv_approx_td(int R, int OV, int early, int eval, int tuned4) {
  using namespace specinv::fast_approx;
  if (tuned4 && R == 8 && OV == 4 && early == 0 && eval == 0) return (const void*)k_fused4_td<8, false, false>;
  if (tuned4 && R == 8 && OV == 4 && early == 0 && eval == 1) return (const void*)k_fused4_td<8, false, true>;
  if (tuned4 && R == 8 && OV == 4 && early == 1 && eval == 0) return (const void*)k_fused4_td<8, true, false>;
  if (tuned4 && R == 8 && OV == 4 && early == 1 && eval == 1) return (const void*)k_fused4_td<8, true, true>;
  if (tuned4 && R == 16 && OV == 4 && early == 0 && eval == 0) return (const void*)k_fused4_td<16, false, false>;
  if (tuned4 && R == 16 && OV == 4 && early == 0 && eval == 1) return (const void*)k_fused4_td<16, false, true>;
  if (tuned4 && R == 16 && OV == 4 && early == 1 && eval == 0) return (const void*)k_fused4_td<16, true, false>;
  if (tuned4 && R == 16 && OV == 4 && early == 1 && eval == 1) return (const void*)k_fused4_td<16, true, true>;
  if (!tuned4 && R == 4 && OV == 4 && early == 0 && eval == 0) return (const void*)k_fused_td<4, 4, false, false>;
  if (!tuned4 && R == 4 && OV == 4 && early == 0 && eval == 1) return (const void*)k_fused_td<4, 4, false, true>;
  if (!tuned4 && R == 4 && OV == 4 && early == 1 && eval == 0) return (const void*)k_fused_td<4, 4, true, false>;
  if (!tuned4 && R == 4 && OV == 4 && early == 1 && eval == 1) return (const void*)k_fused_td<4, 4, true, true>;
  if (!tuned4 && R == 4 && OV == 2 && early == 0 && eval == 0) return (const void*)k_fused_td<4, 2, false, false>;
  if (!tuned4 && R == 4 && OV == 2 && early == 0 && eval == 1) return (const void*)k_fused_td<4, 2, false, true>;
  if (!tuned4 && R == 4 && OV == 2 && early == 1 && eval == 0) return (const void*)k_fused_td<4, 2, true, false>;
  if (!tuned4 && R == 4 && OV == 2 && early == 1 && eval == 1) return (const void*)k_fused_td<4, 2, true, true>;
  if (!tuned4 && R == 8 && OV == 8 && early == 0 && eval == 0) return (const void*)k_fused_td<8, 8, false, false>;
  if (!tuned4 && R == 8 && OV == 8 && early == 0 && eval == 1) return (const void*)k_fused_td<8, 8, false, true>;
  if (!tuned4 && R == 8 && OV == 8 && early == 1 && eval == 0) return (const void*)k_fused_td<8, 8, true, false>;
  if (!tuned4 && R == 8 && OV == 8 && early == 1 && eval == 1) return (const void*)k_fused_td<8, 8, true, true>;
  if (!tuned4 && R == 8 && OV == 2 && early == 0 && eval == 0) return (const void*)k_fused_td<8, 2, false, false>;
  if (!tuned4 && R == 8 && OV == 2 && early == 0 && eval == 1) return (const void*)k_fused_td<8, 2, false, true>;
  if (!tuned4 && R == 8 && OV == 2 && early == 1 && eval == 0) return (const void*)k_fused_td<8, 2, true, false>;
  if (!tuned4 && R == 8 && OV == 2 && early == 1 && eval == 1) return (const void*)k_fused_td<8, 2, true, true>;
  if (!tuned4 && R == 16 && OV == 8 && early == 0 && eval == 0) return (const void*)k_fused_td<16, 8, false, false>;
  if (!tuned4 && R == 16 && OV == 8 && early == 0 && eval == 1) return (const void*)k_fused_td<16, 8, false, true>;
  if (!tuned4 && R == 16 && OV == 8 && early == 1 && eval == 0) return (const void*)k_fused_td<16, 8, true, false>;
  if (!tuned4 && R == 16 && OV == 8 && early == 1 && eval == 1) return (const void*)k_fused_td<16, 8, true, true>;
  if (!tuned4 && R == 16 && OV == 2 && early == 0 && eval == 0) return (const void*)k_fused_td<16, 2, false, false>;
  if (!tuned4 && R == 16 && OV == 2 && early == 0 && eval == 1) return (const void*)k_fused_td<16, 2, false, true>;
  if (!tuned4 && R == 16 && OV == 2 && early == 1 && eval == 0) return (const void*)k_fused_td<16, 2, true, false>;
  if (!tuned4 && R == 16 && OV == 2 && early == 1 && eval == 1) return (const void*)k_fused_td<16, 2, true, true>;
  return nullptr;
}
