// k_fused4_td / k_fused_td<R, OV> with the exact projection.
// The exact-projection copy of these kernels (fast_core.h): correctly rounded sqrt / divisions and a true division by the envelope,
// the reference's own operations (torch_specinv/methods.py:132,246-247), in namespace specinv::fast_exact.  The host side takes
// the kernels' addresses from the table function below (specinv_plan_set_exact).
#define SPECINV_IEEE 1
#define SI_FAST_NS fast_exact
#include "kernels_fast_td.h"

namespace specinv {
namespace fast_exact {

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

}  // namespace fast_exact
}  // namespace specinv

extern "C" __attribute__((visibility("hidden"))) const void* specinv_exact_td(int R, int OV, int early, int eval, int tuned4) {
  using namespace specinv::fast_exact;
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
