// k_fused4: the spectral-state iteration kernel of the headline shapes (hop = n_fft/4 at n_fft 1024 / 2048; ADMM, and Griffin-Lim on request).
// Explicit instantiations: the host side (fast_state.h / rtisi_fast_host.h / kernels_lbfgs.h) takes these kernels' addresses from
// the declarations in fast_core.h / rtisi_fast_host.h / objective_args.h; a kernel missing here is an undefined symbol at link time.
#include "kernels_fused.h"

namespace specinv {
namespace fast {

template __global__ void k_fused4<8, MODE_GLA, false>(FastArgs);
template __global__ void k_fused4<8, MODE_GLA, true>(FastArgs);
template __global__ void k_fused4<8, MODE_ADMM, false>(FastArgs);
template __global__ void k_fused4<8, MODE_ADMM, true>(FastArgs);
template __global__ void k_fused4<16, MODE_GLA, false>(FastArgs);
template __global__ void k_fused4<16, MODE_GLA, true>(FastArgs);
template __global__ void k_fused4<16, MODE_ADMM, false>(FastArgs);
template __global__ void k_fused4<16, MODE_ADMM, true>(FastArgs);

}  // namespace fast
}  // namespace specinv
