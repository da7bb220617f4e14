// k_hop_td: the chunked frame kernel with the momentum carried as a signal.
// Explicit instantiations: the host side (fast_state.h / rtisi_fast_host.h / kernels_lbfgs.h) takes these kernels' addresses from
// the declarations in fast_core.h / rtisi_fast_host.h / objective_args.h; a kernel missing here is an undefined symbol at link time.
#include "kernels_frame.h"

namespace specinv {
namespace fast {

template __global__ void k_hop_td<4, false, false>(HopArgs);
template __global__ void k_hop_td<4, false, true>(HopArgs);
template __global__ void k_hop_td<4, true, false>(HopArgs);
template __global__ void k_hop_td<4, true, true>(HopArgs);
template __global__ void k_hop_td<8, false, false>(HopArgs);
template __global__ void k_hop_td<8, false, true>(HopArgs);
template __global__ void k_hop_td<8, true, false>(HopArgs);
template __global__ void k_hop_td<8, true, true>(HopArgs);
template __global__ void k_hop_td<16, false, false>(HopArgs);
template __global__ void k_hop_td<16, false, true>(HopArgs);
template __global__ void k_hop_td<16, true, false>(HopArgs);
template __global__ void k_hop_td<16, true, true>(HopArgs);

}  // namespace fast
}  // namespace specinv
