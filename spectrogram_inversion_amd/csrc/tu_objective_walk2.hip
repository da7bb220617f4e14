// k_objective_walk at hop = n_fft / 2 and n_fft / 8 (kernels_objective_walk.h).
#include "kernels_objective_walk.h"

namespace specinv {
namespace fast {

template __global__ void k_objective_walk<8, 2>(ObjWalkArgs);
template __global__ void k_objective_walk<16, 2>(ObjWalkArgs);
template __global__ void k_objective_walk<8, 8>(ObjWalkArgs);
template __global__ void k_objective_walk<16, 8>(ObjWalkArgs);

}  // namespace fast
}  // namespace specinv
