// k_hop (chunks of frames, overlap-add in an LDS ring) and the adjoint of the analysis in that structure.
// Explicit instantiations: the host side (fast_state.h / rtisi_fast_host.h / kernels_lbfgs.h) takes these kernels' addresses from
// the declarations in fast_core.h / rtisi_fast_host.h / objective_args.h; a kernel missing here is an undefined symbol at link time.
#include "kernels_frame.h"

namespace specinv {
namespace fast {

template __global__ void k_hop<4, MODE_GLA, false>(HopArgs);
template __global__ void k_hop<4, MODE_GLA, true>(HopArgs);
template __global__ void k_hop<4, MODE_ADMM, false>(HopArgs);
template __global__ void k_hop<4, MODE_ADMM, true>(HopArgs);
template __global__ void k_hop<4, MODE_INIT, false>(HopArgs);
template __global__ void k_hop_inverse<4>(HopInvArgs);
template __global__ void k_hop<8, MODE_GLA, false>(HopArgs);
template __global__ void k_hop<8, MODE_GLA, true>(HopArgs);
template __global__ void k_hop<8, MODE_ADMM, false>(HopArgs);
template __global__ void k_hop<8, MODE_ADMM, true>(HopArgs);
template __global__ void k_hop<8, MODE_INIT, false>(HopArgs);
template __global__ void k_hop_inverse<8>(HopInvArgs);
template __global__ void k_hop<16, MODE_GLA, false>(HopArgs);
template __global__ void k_hop<16, MODE_GLA, true>(HopArgs);
template __global__ void k_hop<16, MODE_ADMM, false>(HopArgs);
template __global__ void k_hop<16, MODE_ADMM, true>(HopArgs);
template __global__ void k_hop<16, MODE_INIT, false>(HopArgs);
template __global__ void k_hop_inverse<16>(HopInvArgs);

}  // namespace fast
}  // namespace specinv
