// k_semi (one frame per wave, any hop) and the stand-alone transforms.
// Explicit instantiations: the host side (fast_state.h / rtisi_fast_host.h / kernels_lbfgs.h) takes these kernels' addresses from
// the declarations in fast_core.h / rtisi_fast_host.h / objective_args.h; a kernel missing here is an undefined symbol at link time.
#include "kernels_frame.h"

namespace specinv {
namespace fast {

template __global__ void k_semi<4, MODE_GLA, false>(SemiArgs);
template __global__ void k_semi<4, MODE_GLA, true>(SemiArgs);
template __global__ void k_semi<4, MODE_ADMM, false>(SemiArgs);
template __global__ void k_semi<4, MODE_ADMM, true>(SemiArgs);
template __global__ void k_semi<4, MODE_INIT, false>(SemiArgs);
template __global__ void k_fast_stft<4>(FastXformArgs);
template __global__ void k_fast_inverse_frames<4>(FastXformArgs);
template __global__ void k_semi<8, MODE_GLA, false>(SemiArgs);
template __global__ void k_semi<8, MODE_GLA, true>(SemiArgs);
template __global__ void k_semi<8, MODE_ADMM, false>(SemiArgs);
template __global__ void k_semi<8, MODE_ADMM, true>(SemiArgs);
template __global__ void k_semi<8, MODE_INIT, false>(SemiArgs);
template __global__ void k_fast_stft<8>(FastXformArgs);
template __global__ void k_fast_inverse_frames<8>(FastXformArgs);
template __global__ void k_semi<16, MODE_GLA, false>(SemiArgs);
template __global__ void k_semi<16, MODE_GLA, true>(SemiArgs);
template __global__ void k_semi<16, MODE_ADMM, false>(SemiArgs);
template __global__ void k_semi<16, MODE_ADMM, true>(SemiArgs);
template __global__ void k_semi<16, MODE_INIT, false>(SemiArgs);
template __global__ void k_fast_stft<16>(FastXformArgs);
template __global__ void k_fast_inverse_frames<16>(FastXformArgs);
template __global__ void k_semi<32, MODE_GLA, false>(SemiArgs);
template __global__ void k_semi<32, MODE_GLA, true>(SemiArgs);
template __global__ void k_semi<32, MODE_ADMM, false>(SemiArgs);
template __global__ void k_semi<32, MODE_ADMM, true>(SemiArgs);
template __global__ void k_semi<32, MODE_INIT, false>(SemiArgs);
template __global__ void k_fast_stft<32>(FastXformArgs);
template __global__ void k_fast_inverse_frames<32>(FastXformArgs);

}  // namespace fast
}  // namespace specinv
