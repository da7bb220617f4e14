// RTISI-LA device path (methods.py:273-412).
#pragma once
#include "common.h"

namespace specinv {

template <typename P, typename T>
int rtisi_launch(P&, const T*, int, int, int, double, T*) {
  return fail(SPECINV_EUNSUPPORTED, "RTISI_LA device path not built yet");
}

}  // namespace specinv
