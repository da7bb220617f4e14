// k_fused4_td: Griffin-Lim with the momentum carried as a signal at the headline shapes (BASELINE C2 runs k_fused4_td<16>).
// Explicit instantiations: the host side (fast_state.h / rtisi_fast_host.h / kernels_lbfgs.h) takes these kernels' addresses from
// the declarations in fast_core.h / rtisi_fast_host.h / objective_args.h; a kernel missing here is an undefined symbol at link time.
#include "kernels_fast_td.h"

namespace specinv {
namespace fast {

template __global__ void k_fused4_td<8, false, false>(FastArgs);
template __global__ void k_fused4_td<8, false, true>(FastArgs);
template __global__ void k_fused4_td<8, true, false>(FastArgs);
template __global__ void k_fused4_td<8, true, true>(FastArgs);
template __global__ void k_fused4_td<16, false, false>(FastArgs);
template __global__ void k_fused4_td<16, false, true>(FastArgs);
template __global__ void k_fused4_td<16, true, false>(FastArgs);
template __global__ void k_fused4_td<16, true, true>(FastArgs);
template __global__ void k_eval_td<8, 4>(FastArgs);
template __global__ void k_eval_td<16, 4>(FastArgs);

}  // namespace fast
}  // namespace specinv
