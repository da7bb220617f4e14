// k_objective_logmel: loss and gradient of the log-mel / magnitude objective in one launch.
// Explicit instantiations: the host side (fast_state.h / rtisi_fast_host.h / kernels_lbfgs.h) takes these kernels' addresses from
// the declarations in fast_core.h / rtisi_fast_host.h / objective_args.h; a kernel missing here is an undefined symbol at link time.
#include "kernels_objective.h"

namespace specinv {
namespace fast {

template __global__ void k_objective_logmel<8, 3, false>(ObjArgs);
template __global__ void k_objective_logmel<8, 4, false>(ObjArgs);
template __global__ void k_objective_logmel<8, 5, false>(ObjArgs);
template __global__ void k_objective_logmel<8, 9, false>(ObjArgs);
template __global__ void k_objective_logmel<16, 3, false>(ObjArgs);
template __global__ void k_objective_logmel<16, 4, false>(ObjArgs);
template __global__ void k_objective_logmel<16, 5, false>(ObjArgs);
template __global__ void k_objective_logmel<16, 8, false>(ObjArgs);
template __global__ void k_objective_logmel<8, 9, false, true>(ObjArgs);
template __global__ void k_objective_logmel<16, 9, false, true>(ObjArgs);
template __global__ void k_objective_logmel<8, 3, true>(ObjArgs);
template __global__ void k_objective_logmel<16, 3, true>(ObjArgs);

}  // namespace fast
}  // namespace specinv
