// k_semi2: the frame kernel of two-sided spectrograms (explicit instantiations, see tu_semi.hip).
#include "kernels_frame.h"

namespace specinv {
namespace fast {

template __global__ void k_semi2<4, MODE_GLA, false>(SemiArgs);
template __global__ void k_semi2<4, MODE_GLA, true>(SemiArgs);
template __global__ void k_semi2<4, MODE_ADMM, false>(SemiArgs);
template __global__ void k_semi2<4, MODE_ADMM, true>(SemiArgs);
template __global__ void k_semi2<4, MODE_INIT, false>(SemiArgs);
template __global__ void k_semi2<8, MODE_GLA, false>(SemiArgs);
template __global__ void k_semi2<8, MODE_GLA, true>(SemiArgs);
template __global__ void k_semi2<8, MODE_ADMM, false>(SemiArgs);
template __global__ void k_semi2<8, MODE_ADMM, true>(SemiArgs);
template __global__ void k_semi2<8, MODE_INIT, false>(SemiArgs);
template __global__ void k_semi2<16, MODE_GLA, false>(SemiArgs);
template __global__ void k_semi2<16, MODE_GLA, true>(SemiArgs);
template __global__ void k_semi2<16, MODE_ADMM, false>(SemiArgs);
template __global__ void k_semi2<16, MODE_ADMM, true>(SemiArgs);
template __global__ void k_semi2<16, MODE_INIT, false>(SemiArgs);
template __global__ void k_semi2<32, MODE_GLA, false>(SemiArgs);
template __global__ void k_semi2<32, MODE_GLA, true>(SemiArgs);
template __global__ void k_semi2<32, MODE_ADMM, false>(SemiArgs);
template __global__ void k_semi2<32, MODE_ADMM, true>(SemiArgs);
template __global__ void k_semi2<32, MODE_INIT, false>(SemiArgs);

}  // namespace fast
}  // namespace specinv
