// k_rtisi_fast: the persistent RTISI-LA kernel on the wave-level FFT.
// Explicit instantiations: the host side (fast_state.h / rtisi_fast_host.h / kernels_lbfgs.h) takes these kernels' addresses from
// the declarations in fast_core.h / rtisi_fast_host.h / objective_args.h; a kernel missing here is an undefined symbol at link time.
#include "kernels_rtisi_fast.h"

namespace specinv {
namespace fast {

template __global__ void k_rtisi_fast<4, 256, 4>(RtisiFastArgs);
template __global__ void k_rtisi_fast<4, 512, 4>(RtisiFastArgs);
template __global__ void k_rtisi_fast<4, 256, 2>(RtisiFastArgs);
template __global__ void k_rtisi_fast<4, 512, 2>(RtisiFastArgs);
template __global__ void k_rtisi_fast<8, 256, 8>(RtisiFastArgs);
template __global__ void k_rtisi_fast<8, 512, 8>(RtisiFastArgs);
template __global__ void k_rtisi_fast<8, 256, 4>(RtisiFastArgs);
template __global__ void k_rtisi_fast<8, 512, 4>(RtisiFastArgs);
template __global__ void k_rtisi_fast<8, 256, 2>(RtisiFastArgs);
template __global__ void k_rtisi_fast<8, 512, 2>(RtisiFastArgs);
template __global__ void k_rtisi_fast<16, 256, 8>(RtisiFastArgs);
template __global__ void k_rtisi_fast<16, 512, 8>(RtisiFastArgs);
template __global__ void k_rtisi_fast<16, 256, 4>(RtisiFastArgs);
template __global__ void k_rtisi_fast<16, 512, 4>(RtisiFastArgs);
template __global__ void k_rtisi_fast<16, 256, 2>(RtisiFastArgs);
template __global__ void k_rtisi_fast<16, 512, 2>(RtisiFastArgs);

}  // namespace fast
}  // namespace specinv
