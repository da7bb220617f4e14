// L_BFGS building blocks (methods.py:509-569): fused transform forward / loss+gradient and the
// flat-vector kernels of the two-loop recursion.
#pragma once
#include "common.h"

namespace specinv {

template <typename P, typename T>
int tf_setup(P&, int, const T*, int) { return fail(SPECINV_EUNSUPPORTED, "transform path not built yet"); }
template <typename P, typename T>
int tf_forward(P&, const T*, int64_t, T*) { return fail(SPECINV_EUNSUPPORTED, "transform path not built yet"); }
template <typename P, typename T>
int tf_loss_grad(P&, const T*, int64_t, const T*, double*, T*) {
  return fail(SPECINV_EUNSUPPORTED, "transform path not built yet");
}
template <typename P, typename T>
int lb_dot(P&, const T*, const T*, int64_t, double*) { return fail(SPECINV_EUNSUPPORTED, "not built yet"); }
template <typename P, typename T>
int lb_axpy(P&, T, const T*, T*, int64_t) { return fail(SPECINV_EUNSUPPORTED, "not built yet"); }
template <typename P, typename T>
int lb_scale(P&, T, const T*, T*, int64_t) { return fail(SPECINV_EUNSUPPORTED, "not built yet"); }
template <typename P, typename T>
int lb_absmax_abssum(P&, const T*, int64_t, double*) { return fail(SPECINV_EUNSUPPORTED, "not built yet"); }

}  // namespace specinv
