// k_fused<4, OV> and the fused initial ISTFT at n_fft 512.
// Explicit instantiations: the host side (fast_state.h / rtisi_fast_host.h / kernels_lbfgs.h) takes these kernels' addresses from
// the declarations in fast_core.h / rtisi_fast_host.h / objective_args.h; a kernel missing here is an undefined symbol at link time.
#include "kernels_fused.h"

namespace specinv {
namespace fast {

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

}  // namespace fast
}  // namespace specinv
