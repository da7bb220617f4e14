// k_fused_td<R, OV>: the signal-form Griffin-Lim kernel at the other overlaps, n_fft 512 / 1024.
// Explicit instantiations: the host side (fast_state.h / rtisi_fast_host.h / kernels_lbfgs.h) takes these kernels' addresses from
// the declarations in fast_core.h / rtisi_fast_host.h / objective_args.h; a kernel missing here is an undefined symbol at link time.
#include "kernels_fast_td.h"

namespace specinv {
namespace fast {

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

}  // namespace fast
}  // namespace specinv
