// k_objective_walk: the log-mel objective as a frame walk (kernels_objective_walk.h), hop = n_fft / 4.  Explicit instantiations: the
// host side (kernels_lbfgs.h) takes the kernels' addresses from the declaration in objective_args.h.
#include "kernels_objective_walk.h"

namespace specinv {
namespace fast {

template __global__ void k_objective_walk<8, 4>(ObjWalkArgs);
template __global__ void k_objective_walk<16, 4>(ObjWalkArgs);

}  // namespace fast
}  // namespace specinv
