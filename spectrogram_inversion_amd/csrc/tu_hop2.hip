// k_hop2: the chunked frame kernel of two-sided spectrograms (explicit instantiations, see tu_hop.hip).
#include "kernels_frame.h"

namespace specinv {
namespace fast {

template __global__ void k_hop2<4, MODE_GLA, false>(HopArgs);
template __global__ void k_hop2<4, MODE_GLA, true>(HopArgs);
template __global__ void k_hop2<4, MODE_ADMM, false>(HopArgs);
template __global__ void k_hop2<4, MODE_ADMM, true>(HopArgs);
template __global__ void k_hop2<4, MODE_INIT, false>(HopArgs);
template __global__ void k_hop2<8, MODE_GLA, false>(HopArgs);
template __global__ void k_hop2<8, MODE_GLA, true>(HopArgs);
template __global__ void k_hop2<8, MODE_ADMM, false>(HopArgs);
template __global__ void k_hop2<8, MODE_ADMM, true>(HopArgs);
template __global__ void k_hop2<8, MODE_INIT, false>(HopArgs);
template __global__ void k_hop2<16, MODE_GLA, false>(HopArgs);
template __global__ void k_hop2<16, MODE_GLA, true>(HopArgs);
template __global__ void k_hop2<16, MODE_ADMM, false>(HopArgs);
template __global__ void k_hop2<16, MODE_ADMM, true>(HopArgs);
template __global__ void k_hop2<16, MODE_INIT, false>(HopArgs);

}  // namespace fast
}  // namespace specinv
