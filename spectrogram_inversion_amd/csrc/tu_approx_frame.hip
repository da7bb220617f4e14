// k_semi / k_hop / k_hop_td with the approximate projection (opt-in).
// The second copy of these kernels (fast_core.h, SPECINV_IEEE=0): m * v_rsq_f32(|s|^2 + 1e-32) in the projection and a multiplication
// by 1 / envelope instead of the reference's operation order with correctly rounded factors (the default build), in namespace
// specinv::fast_approx.  The host side takes the kernels' addresses from the table function below (specinv_plan_set_exact(plan, 0)).
#define SPECINV_IEEE 0
#define SI_FAST_NS fast_approx
#include "kernels_frame.h"

namespace specinv {
namespace fast_approx {

template __global__ void k_semi<4, MODE_GLA, false>(SemiArgs);
template __global__ void k_semi<4, MODE_GLA, true>(SemiArgs);
template __global__ void k_semi<4, MODE_ADMM, false>(SemiArgs);
template __global__ void k_semi<4, MODE_ADMM, true>(SemiArgs);
template __global__ void k_semi<4, MODE_INIT, false>(SemiArgs);
template __global__ void k_semi<8, MODE_GLA, false>(SemiArgs);
template __global__ void k_semi<8, MODE_GLA, true>(SemiArgs);
template __global__ void k_semi<8, MODE_ADMM, false>(SemiArgs);
template __global__ void k_semi<8, MODE_ADMM, true>(SemiArgs);
template __global__ void k_semi<8, MODE_INIT, false>(SemiArgs);
template __global__ void k_semi<16, MODE_GLA, false>(SemiArgs);
template __global__ void k_semi<16, MODE_GLA, true>(SemiArgs);
template __global__ void k_semi<16, MODE_ADMM, false>(SemiArgs);
template __global__ void k_semi<16, MODE_ADMM, true>(SemiArgs);
template __global__ void k_semi<16, MODE_INIT, false>(SemiArgs);
template __global__ void k_semi<32, MODE_GLA, false>(SemiArgs);
template __global__ void k_semi<32, MODE_GLA, true>(SemiArgs);
template __global__ void k_semi<32, MODE_ADMM, false>(SemiArgs);
template __global__ void k_semi<32, MODE_ADMM, true>(SemiArgs);
template __global__ void k_semi<32, MODE_INIT, false>(SemiArgs);
template __global__ void k_hop<4, MODE_GLA, false>(HopArgs);
template __global__ void k_hop<4, MODE_GLA, true>(HopArgs);
template __global__ void k_hop<4, MODE_ADMM, false>(HopArgs);
template __global__ void k_hop<4, MODE_ADMM, true>(HopArgs);
template __global__ void k_hop<4, MODE_INIT, false>(HopArgs);
template __global__ void k_hop_td<4, false, false>(HopArgs);
template __global__ void k_hop_td<4, false, true>(HopArgs);
template __global__ void k_hop_td<4, true, false>(HopArgs);
template __global__ void k_hop_td<4, true, true>(HopArgs);
template __global__ void k_hop<8, MODE_GLA, false>(HopArgs);
template __global__ void k_hop<8, MODE_GLA, true>(HopArgs);
template __global__ void k_hop<8, MODE_ADMM, false>(HopArgs);
template __global__ void k_hop<8, MODE_ADMM, true>(HopArgs);
template __global__ void k_hop<8, MODE_INIT, false>(HopArgs);
template __global__ void k_hop_td<8, false, false>(HopArgs);
template __global__ void k_hop_td<8, false, true>(HopArgs);
template __global__ void k_hop_td<8, true, false>(HopArgs);
template __global__ void k_hop_td<8, true, true>(HopArgs);
template __global__ void k_hop<16, MODE_GLA, false>(HopArgs);
template __global__ void k_hop<16, MODE_GLA, true>(HopArgs);
template __global__ void k_hop<16, MODE_ADMM, false>(HopArgs);
template __global__ void k_hop<16, MODE_ADMM, true>(HopArgs);
template __global__ void k_hop<16, MODE_INIT, false>(HopArgs);
template __global__ void k_hop_td<16, false, false>(HopArgs);
template __global__ void k_hop_td<16, false, true>(HopArgs);
template __global__ void k_hop_td<16, true, false>(HopArgs);
template __global__ void k_hop_td<16, true, true>(HopArgs);

}  // namespace fast_approx
}  // namespace specinv

extern "C" __attribute__((visibility("hidden"))) int specinv_approx_units_built(void) { return 1; }   // (tu_noapprox.hip: 0)

extern "C" __attribute__((visibility("hidden"))) const void* specinv_approx_frame(int family /* 0 k_semi, 1 k_hop, 2 k_hop_td */, int R, int a /* mode, or early */, int b /* eval */) {
  using namespace specinv::fast_approx;
  if (family == 0 && R == 4 && a == 0 && b == 0) return (const void*)k_semi<4, MODE_GLA, false>;
  if (family == 0 && R == 4 && a == 0 && b == 1) return (const void*)k_semi<4, MODE_GLA, true>;
  if (family == 0 && R == 4 && a == 1 && b == 0) return (const void*)k_semi<4, MODE_ADMM, false>;
  if (family == 0 && R == 4 && a == 1 && b == 1) return (const void*)k_semi<4, MODE_ADMM, true>;
  if (family == 0 && R == 4 && a == 2 && b == 0) return (const void*)k_semi<4, MODE_INIT, false>;
  if (family == 0 && R == 8 && a == 0 && b == 0) return (const void*)k_semi<8, MODE_GLA, false>;
  if (family == 0 && R == 8 && a == 0 && b == 1) return (const void*)k_semi<8, MODE_GLA, true>;
  if (family == 0 && R == 8 && a == 1 && b == 0) return (const void*)k_semi<8, MODE_ADMM, false>;
  if (family == 0 && R == 8 && a == 1 && b == 1) return (const void*)k_semi<8, MODE_ADMM, true>;
  if (family == 0 && R == 8 && a == 2 && b == 0) return (const void*)k_semi<8, MODE_INIT, false>;
  if (family == 0 && R == 16 && a == 0 && b == 0) return (const void*)k_semi<16, MODE_GLA, false>;
  if (family == 0 && R == 16 && a == 0 && b == 1) return (const void*)k_semi<16, MODE_GLA, true>;
  if (family == 0 && R == 16 && a == 1 && b == 0) return (const void*)k_semi<16, MODE_ADMM, false>;
  if (family == 0 && R == 16 && a == 1 && b == 1) return (const void*)k_semi<16, MODE_ADMM, true>;
  if (family == 0 && R == 16 && a == 2 && b == 0) return (const void*)k_semi<16, MODE_INIT, false>;
  if (family == 0 && R == 32 && a == 0 && b == 0) return (const void*)k_semi<32, MODE_GLA, false>;
  if (family == 0 && R == 32 && a == 0 && b == 1) return (const void*)k_semi<32, MODE_GLA, true>;
  if (family == 0 && R == 32 && a == 1 && b == 0) return (const void*)k_semi<32, MODE_ADMM, false>;
  if (family == 0 && R == 32 && a == 1 && b == 1) return (const void*)k_semi<32, MODE_ADMM, true>;
  if (family == 0 && R == 32 && a == 2 && b == 0) return (const void*)k_semi<32, MODE_INIT, false>;
  if (family == 1 && R == 4 && a == 0 && b == 0) return (const void*)k_hop<4, MODE_GLA, false>;
  if (family == 1 && R == 4 && a == 0 && b == 1) return (const void*)k_hop<4, MODE_GLA, true>;
  if (family == 1 && R == 4 && a == 1 && b == 0) return (const void*)k_hop<4, MODE_ADMM, false>;
  if (family == 1 && R == 4 && a == 1 && b == 1) return (const void*)k_hop<4, MODE_ADMM, true>;
  if (family == 1 && R == 4 && a == 2 && b == 0) return (const void*)k_hop<4, MODE_INIT, false>;
  if (family == 2 && R == 4 && a == 0 && b == 0) return (const void*)k_hop_td<4, false, false>;
  if (family == 2 && R == 4 && a == 0 && b == 1) return (const void*)k_hop_td<4, false, true>;
  if (family == 2 && R == 4 && a == 1 && b == 0) return (const void*)k_hop_td<4, true, false>;
  if (family == 2 && R == 4 && a == 1 && b == 1) return (const void*)k_hop_td<4, true, true>;
  if (family == 1 && R == 8 && a == 0 && b == 0) return (const void*)k_hop<8, MODE_GLA, false>;
  if (family == 1 && R == 8 && a == 0 && b == 1) return (const void*)k_hop<8, MODE_GLA, true>;
  if (family == 1 && R == 8 && a == 1 && b == 0) return (const void*)k_hop<8, MODE_ADMM, false>;
  if (family == 1 && R == 8 && a == 1 && b == 1) return (const void*)k_hop<8, MODE_ADMM, true>;
  if (family == 1 && R == 8 && a == 2 && b == 0) return (const void*)k_hop<8, MODE_INIT, false>;
  if (family == 2 && R == 8 && a == 0 && b == 0) return (const void*)k_hop_td<8, false, false>;
  if (family == 2 && R == 8 && a == 0 && b == 1) return (const void*)k_hop_td<8, false, true>;
  if (family == 2 && R == 8 && a == 1 && b == 0) return (const void*)k_hop_td<8, true, false>;
  if (family == 2 && R == 8 && a == 1 && b == 1) return (const void*)k_hop_td<8, true, true>;
  if (family == 1 && R == 16 && a == 0 && b == 0) return (const void*)k_hop<16, MODE_GLA, false>;
  if (family == 1 && R == 16 && a == 0 && b == 1) return (const void*)k_hop<16, MODE_GLA, true>;
  if (family == 1 && R == 16 && a == 1 && b == 0) return (const void*)k_hop<16, MODE_ADMM, false>;
  if (family == 1 && R == 16 && a == 1 && b == 1) return (const void*)k_hop<16, MODE_ADMM, true>;
  if (family == 1 && R == 16 && a == 2 && b == 0) return (const void*)k_hop<16, MODE_INIT, false>;
  if (family == 2 && R == 16 && a == 0 && b == 0) return (const void*)k_hop_td<16, false, false>;
  if (family == 2 && R == 16 && a == 0 && b == 1) return (const void*)k_hop_td<16, false, true>;
  if (family == 2 && R == 16 && a == 1 && b == 0) return (const void*)k_hop_td<16, true, false>;
  if (family == 2 && R == 16 && a == 1 && b == 1) return (const void*)k_hop_td<16, true, true>;
  return nullptr;
}
