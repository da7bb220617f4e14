// The strong-Wolfe line search of L-BFGS with the decisions on the device (reference: torch_specinv/methods.py:553 ->
// torch.optim.LBFGS.step with line_search_fn='strong_wolfe' -> torch.optim.lbfgs._strong_wolfe, third-party; restated on the host in
// lbfgs.py:_wolfe_packed / _step_wolfe_packed, which this follows decision for decision).
//
// One `optimizer.step` is enqueued as a sequence of identical SLOTS and the host synchronises once:
//     k_lbd_update_x   x = x0 + t d (a trial point), or: the new direction d = sum coef_j v_j, x0 = x, x = x0 + t d (an iteration begins)
//     objective        loss and gradient at x, into the gradient buffer the state names (one of four)
//     k_lbd_pair_stats the evaluation's eight sums against the line search's starting gradient and s = t d
//     k_lbd_multi_dot  products of a gradient with the memory (only for a gradient an iteration begins at)
//     k_lbd_decide_ls  one workgroup: what the evaluation means - bracket phase, zoom phase (cubic interpolation, the stall
//                      guards), the tests that end a line search, an iteration, the step - and what the next slot has to do
// Every kernel reads its orders from the state record; what is not ordered returns at once.  A line search ends on the best
// point of its bracket, not necessarily the last one evaluated: the record keeps (loss, g.d, max|g|, the eight sums, the gradient
// buffer) of the bracket's points, and x is put back on x0 + t d by the next k_lbd_update_x.  With curvature pairs in the memory
// the products of the accepted gradient with them take one slot of their own (no evaluation) before the iteration's decisions.
#pragma once
#include "lbfgs_dev.h"

namespace specinv {

static __global__ void k_lbd_begin_ls(LbdState* st) {
  st->active = 1;
  st->mode = LBD_ENTRY;
  st->do_eval = 1;
  st->do_mdot = 0;                        // (the entry gradient's products with the memory wait for the step to get past its first test)
  st->do_trial = 0;
  st->need_fix = 0;
  st->do_lincomb = 0;
  st->do_step = 0;
  st->n_iter = 0;
  st->evals = 0;
  st->g_eval = (st->g_cur + 1) & 3;       // (no line search is open: only prev_flat_grad's buffer is taken)
  st->g_md = st->g_eval;
  st->eval_slots = 0ull;
}

// after the update an ended step left orders for: nothing is pending any more
static __global__ void k_lbd_clear_orders(LbdState* st) {
  st->do_trial = 0;
  st->need_fix = 0;
  st->do_lincomb = 0;
  st->do_step = 0;
}

// torch.optim.lbfgs._cubic_interpolate (lbfgs.py:_cubic_step): the minimiser of the cubic through (x1, f1, g1), (x2, f2, g2),
// clipped to [lo, hi] (the two abscissae unless bounds are given).  Plain IEEE double operations, no contraction: the decisions
// that hang on it are the host loop's.
__device__ inline double lbd_cubic(double x1, double f1, double g1, double x2, double f2, double g2, bool bounded, double blo, double bhi) {
#pragma clang fp contract(off)
  const double lo = bounded ? blo : (x1 <= x2 ? x1 : x2), hi = bounded ? bhi : (x1 <= x2 ? x2 : x1);
  const double d1 = g1 + g2 - 3.0 * (f1 - f2) / (x1 - x2);
  const double disc = d1 * d1 - g1 * g2;
  if (disc < 0) return 0.5 * (lo + hi);
  const double d2 = sqrt(disc);
  double pos;
  if (x1 <= x2) pos = x2 - (x2 - x1) * ((g2 + d2 - d1) / (g2 - g1 + 2.0 * d2));
  else pos = x1 - (x1 - x2) * ((g1 + d2 - d1) / (g1 - g2 + 2.0 * d2));
  return fmin(fmax(pos, lo), hi);
}

// a gradient buffer (of four) none of the given ones uses
__device__ inline int lbd_free_buffer(int a, int b, int c) {
  for (int i = 0; i < 4; ++i)
    if (i != a && i != b && i != c) return i;
  return 0;
}

// k_lbd_lincomb_step with the line search's orders: a trial point, or the direction of a new iteration with its first trial
template <typename T>
__global__ __launch_bounds__(256) void k_lbd_update_x(LbdPtrs<T> p, T* __restrict__ xs, int64_t n) {
  const LbdState& S = *p.st;
  const bool lin = S.do_lincomb != 0, trial = S.do_trial != 0, fix = S.need_fix != 0;
  if (!lin && !trial) return;
  constexpr int W = 16 / sizeof(T);
  typedef T VT __attribute__((ext_vector_type(W)));
  const int k = S.k_lin;
  const bool step = S.do_step != 0;
  const T t = (T)S.t, tf = (T)S.t_fix;
  const int jy = lin ? S.pair_y : -1, js = lin ? S.pair_s : -1;
  const T* __restrict__ g = p.g4[S.g_cur];
  const T* __restrict__ gp = p.g4[S.g_old];
  const T tp = (T)S.t_pair;
  T* __restrict__ out = p.d;
  T* __restrict__ x0 = p.x0;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t nv = n / W;
  if (i < nv) {
    const VT dv = reinterpret_cast<const VT*>(out)[i];
    if (!lin) {                                   // a trial point (or the accepted one again): x0 + t d, the sum axpy rounds
      const VT x0v = reinterpret_cast<const VT*>(x0)[i];
      const T tt = fix ? tf : t;
      VT xv;
#pragma unroll
      for (int c = 0; c < W; ++c) xv[c] = fma(tt, dv[c], x0v[c]);
      reinterpret_cast<VT*>(xs)[i] = xv;
      return;
    }
    VT xv;
    if (fix) {
      const VT x0v = reinterpret_cast<const VT*>(x0)[i];
#pragma unroll
      for (int c = 0; c < W; ++c) xv[c] = fma(tf, dv[c], x0v[c]);
    } else {
      xv = reinterpret_cast<const VT*>(xs)[i];
    }
    VT yv, sv;
    if (jy >= 0) {
      const VT gv = reinterpret_cast<const VT*>(g)[i], pv = reinterpret_cast<const VT*>(gp)[i];
#pragma unroll
      for (int c = 0; c < W; ++c) {
        yv[c] = gv[c] - pv[c];
        sv[c] = tp * dv[c];
      }
      reinterpret_cast<VT*>(const_cast<T*>(p.lin_ptr[jy]))[i] = yv;
      reinterpret_cast<VT*>(const_cast<T*>(p.lin_ptr[js]))[i] = sv;
    }
    double s[W];
#pragma unroll
    for (int c = 0; c < W; ++c) s[c] = 0.0;
#pragma unroll 4
    for (int j = 0; j < k; ++j) {
      VT v;
      if (j == jy) v = yv;
      else if (j == js) v = sv;
      else v = reinterpret_cast<const VT*>(p.lin_ptr[j])[i];
      const double cj = p.coef[j];
#pragma unroll
      for (int c = 0; c < W; ++c) s[c] += cj * (double)v[c];
    }
    VT r;
#pragma unroll
    for (int c = 0; c < W; ++c) r[c] = (T)s[c];
    reinterpret_cast<VT*>(out)[i] = r;
    reinterpret_cast<VT*>(x0)[i] = xv;
    if (step) {
#pragma unroll
      for (int c = 0; c < W; ++c) xv[c] = fma(t, r[c], xv[c]);
    }
    if (step || fix) reinterpret_cast<VT*>(xs)[i] = xv;
  } else if (i == nv) {
    for (int64_t e = nv * W; e < n; ++e) {
      const T de = out[e];
      if (!lin) {
        xs[e] = fma(fix ? tf : t, de, x0[e]);
        continue;
      }
      T xe = fix ? fma(tf, de, x0[e]) : xs[e];
      T ye = T(0), se = T(0);
      if (jy >= 0) {
        ye = g[e] - gp[e];
        se = tp * de;
        const_cast<T*>(p.lin_ptr[jy])[e] = ye;
        const_cast<T*>(p.lin_ptr[js])[e] = se;
      }
      double s = 0.0;
      for (int j = 0; j < k; ++j) s += p.coef[j] * (double)(j == jy ? ye : j == js ? se : p.lin_ptr[j][e]);
      out[e] = (T)s;
      x0[e] = xe;
      if (step) xe = fma(t, (T)s, xe);
      if (step || fix) xs[e] = xe;
    }
  }
}

// What the slot's evaluation (or, in a LBD_POST slot, the accepted gradient's memory products) means.  256 threads finish the sums;
// the scalar decisions are taken by every thread alike on the same values (no broadcast), thread 0 writes the record.
template <typename T>
__global__ __launch_bounds__(256) void k_lbd_decide_ls(LbdPtrs<T> p, int slot, const double* __restrict__ part_pair, int nb,
                                                       const double* __restrict__ part_dot, const double* __restrict__ loss_slot) {
#pragma clang fp contract(off)
  extern __shared__ double lds_sy[];              // [m * m]
  __shared__ LbdShared sh;
  const LbdState& R = sh.R;
  LbdState& S = *p.st;
  const int tid = threadIdx.x;
  lbd_load_state(sh, p.st);
  if (tid == 0) {                                 // slots decided so far: the host keeps a few slots ahead of it, no more
    p.board[1] = (double)(slot + 1);
    __threadfence_system();
  }
  // The step has ended: what the host needs of it goes to the pinned board (thread 0 reads its own writes of the record back), then
  // the flag the host waits for - no copy of the record, no synchronisation of the stream.
  auto report = [&]() {
    if (tid == 0) {
      double* b = p.board;
      b[kLbdInfo + 0] = S.first_loss;
      b[kLbdInfo + 1] = S.loss;
      b[kLbdInfo + 2] = S.t;
      b[kLbdInfo + 3] = (double)S.total_iters;
      b[kLbdInfo + 4] = (double)S.func_evals;
      b[kLbdInfo + 5] = (double)S.n_iter;
      b[kLbdInfo + 6] = (double)S.m;
      b[kLbdInfo + 7] = (double)S.pairs_accepted;
      b[kLbdInfo + 8] = (double)S.pairs_rejected;
      b[kLbdInfo + 9] = (double)S.evals;
      b[kLbdInfo + 10] = (double)(unsigned)(S.eval_slots & 0xffffffffull);
      b[kLbdInfo + 11] = (double)(unsigned)(S.eval_slots >> 32);
      b[kLbdInfo + 12] = (double)((S.do_trial ? 1 : 0) | (S.need_fix ? 2 : 0) | (S.do_lincomb ? 4 : 0));   // orders the next update still carries out
      __threadfence_system();
      b[0] = 0.0;
      __threadfence_system();
    }
  };
  auto stop = [&](int fix, double t_fix) {        // the step ends here (x may still have to be put on the accepted point)
    if (tid == 0) {
      S.active = 0;
      S.mode = LBD_IDLE;
      S.do_lincomb = 0;
      S.do_step = 0;
      S.do_eval = 0;
      S.do_mdot = 0;
      S.do_trial = fix;
      S.need_fix = fix;
      S.t_fix = t_fix;
    }
    report();
  };
  if (!R.active) {
    if (tid == 0) {                               // (the update an ended step left orders for has run: nothing is pending)
      S.do_trial = 0;
      S.need_fix = 0;
      S.do_lincomb = 0;
      S.do_step = 0;
    }
    return;
  }
  const int mode = R.mode;
  lbd_finish_sums(sh, part_pair, nb, part_dot, R.do_mdot ? R.k_dot : 0);
  const double loss = *loss_slot;
  if (tid == 0 && mode != LBD_POST && slot < 64) S.eval_slots = R.eval_slots | (1ull << slot);
  LbdPoint A;                                     // the point the next iteration begins at
  int fix_pending = 0, evals_now = R.evals;
  if (mode == LBD_ENTRY) {
    // ---- the step's entry evaluation (torch.optim.LBFGS.step: orig_loss, opt_cond)
    evals_now = 1;
    if (tid == 0) {
      S.first_loss = loss;
      S.loss = loss;
      S.evals = 1;
      S.func_evals = R.func_evals + 1;
    }
    if (sh.bps[2] <= R.tol_grad) {
      stop(0, 0.0);
      return;
    }
    A.t = R.t;
    A.f = loss;
    A.gtd = sh.bps[0];
    A.gmax = sh.bps[2];
#pragma unroll
    for (int c = 0; c < 8; ++c) A.ps[c] = sh.bps[c];
    A.g = R.g_eval;
    A.pad_ = 0;
    if (R.m > 0) {                                // its products with the memory: a slot without an evaluation (many a step ends above)
      if (tid == 0) {
        S.ls_acc = A;
        S.mode = LBD_POST;
        S.do_mdot = 1;
        S.g_md = A.g;
        S.do_eval = 0;
        S.do_trial = 0;
        S.need_fix = 0;
        S.do_lincomb = 0;
        S.do_step = 0;
      }
      return;
    }
  } else if (mode == LBD_TRIAL) {
    // ---- a trial point of the line search has been evaluated (lbfgs.py:_wolfe_packed)
    evals_now = R.evals + 1;
    LbdPoint nw;
    nw.t = R.t;
    nw.f = loss;
    nw.gtd = sh.bps[0];
    nw.gmax = sh.bps[2];
#pragma unroll
    for (int c = 0; c < 8; ++c) nw.ps[c] = sh.bps[c];
    nw.g = R.g_eval;
    const double dnorm = R.ls_first ? sh.bps[3] : R.ls_dnorm;
    const double f0 = R.ls_f0, gtd0 = R.ls_gtd0, c1 = 1e-4, c2 = 0.9;
    const int max_ls = R.ls_max;
    int it = R.ls_it, phase = R.ls_phase, nbr = R.ls_nbr, lo = R.ls_lo, hi = R.ls_hi, stalled = R.ls_stalled;
    LbdPoint prev = R.ls_prev, br[2] = {R.ls_br[0], R.ls_br[1]};
    bool done = false, want_next = false;
    double t_next = 0.0;
    if (phase == 1) {                             // bracket phase
      if (!R.ls_first) it += 1;
      bool to_zoom = true;
      if (it >= max_ls) {
        br[0] = R.ls_start;
        br[1] = nw;
        nbr = 2;
      } else if (nw.f > f0 + c1 * nw.t * gtd0 || (it > 1 && nw.f >= prev.f)) {
        br[0] = prev;
        br[1] = nw;
        nbr = 2;
      } else if (fabs(nw.gtd) <= -c2 * gtd0) {
        br[0] = nw;
        nbr = 1;
        done = true;
      } else if (nw.gtd >= 0) {
        br[0] = prev;
        br[1] = nw;
        nbr = 2;
      } else {
        t_next = lbd_cubic(prev.t, prev.f, prev.gtd, nw.t, nw.f, nw.gtd, true, nw.t + 0.01 * (nw.t - prev.t), nw.t * 10.0);
        prev = nw;
        want_next = true;
        to_zoom = false;
      }
      if (to_zoom) {
        phase = 2;
        stalled = 0;
        const double f_last = nbr == 2 ? br[1].f : br[0].f;
        lo = br[0].f <= f_last ? 0 : 1;
        hi = 1 - lo;
      }
    } else {                                      // zoom phase: the point interpolated inside the bracket
      it += 1;
      if (nw.f > f0 + c1 * nw.t * gtd0 || nw.f >= br[lo].f) {
        br[hi] = nw;
        lo = br[0].f <= br[1].f ? 0 : 1;
        hi = 1 - lo;
      } else {
        if (fabs(nw.gtd) <= -c2 * gtd0) done = true;
        else if (nw.gtd * (br[hi].t - br[lo].t) >= 0) br[hi] = br[lo];
        br[lo] = nw;
      }
    }
    if (phase == 2 && !want_next && !done && it < max_ls && !(fabs(br[1].t - br[0].t) * dnorm < 1e-9)) {
      double t = lbd_cubic(br[0].t, br[0].f, br[0].gtd, br[1].t, br[1].f, br[1].gtd, false, 0.0, 0.0);
      const double bmax = fmax(br[0].t, br[1].t), bmin = fmin(br[0].t, br[1].t), margin = 0.1 * (bmax - bmin);
      if (fmin(bmax - t, t - bmin) < margin) {
        if (stalled || t >= bmax || t <= bmin) {
          t = fabs(t - bmax) < fabs(t - bmin) ? bmax - margin : bmin + margin;
          stalled = 0;
        } else {
          stalled = 1;
        }
      } else {
        stalled = 0;
      }
      t_next = t;
      want_next = true;
    }
    if (want_next) {
      if (tid == 0) {
        S.func_evals = R.func_evals + 1;
        S.evals = evals_now;
        S.ls_it = it;
        S.ls_phase = phase;
        S.ls_nbr = nbr;
        S.ls_lo = lo;
        S.ls_hi = hi;
        S.ls_stalled = stalled;
        S.ls_first = 0;
        S.ls_dnorm = dnorm;
        S.ls_prev = prev;
        S.ls_br[0] = br[0];
        S.ls_br[1] = br[1];
        S.t = t_next;
        S.do_trial = 1;
        S.need_fix = 0;
        S.do_eval = 1;
        S.do_mdot = 0;
        S.do_lincomb = 0;
        S.do_step = 0;
        S.mode = LBD_TRIAL;
        // a buffer no live point of the search sits in
        S.g_eval = phase == 1 ? lbd_free_buffer(R.g_cur, prev.g, prev.g) : lbd_free_buffer(R.g_cur, br[0].g, br[1].g);
      }
      return;
    }
    // ---- the line search ends on the lower point of its bracket (torch.optim.LBFGS.step after _strong_wolfe)
    const LbdPoint best = nbr == 1 ? br[0] : br[lo];
    fix_pending = best.g != nw.g ? 1 : 0;
    if (tid == 0) {
      S.func_evals = R.func_evals + 1;
      S.evals = evals_now;
      S.loss = best.f;
      S.t = best.t;
      S.ls_phase = 0;
    }
    const bool end = R.n_iter == R.max_iter || evals_now >= R.max_eval || best.gmax <= R.tol_grad ||
                     fabs(best.t) * dnorm <= R.tol_change || fabs(best.f - R.prev_loss) < R.tol_change;
    if (end) {
      stop(fix_pending, best.t);                  // (prev_flat_grad stays the gradient the search started from)
      return;
    }
    A = best;
    if (R.m > 0) {                                // its products with the memory first: a slot without an evaluation
      if (tid == 0) {
        S.ls_acc = best;
        S.mode = LBD_POST;
        S.do_mdot = 1;
        S.g_md = best.g;
        S.do_eval = 0;
        S.do_trial = fix_pending;
        S.need_fix = fix_pending;
        S.t_fix = best.t;
        S.do_lincomb = 0;
        S.do_step = 0;
      }
      return;
    }
  } else {
    A = R.ls_acc;                                 // LBD_POST: sh.dotv holds its products with the memory
  }
  // ---- an iteration begins at A (its eight sums were taken against prev_flat_grad and s = A.t d)
  __syncthreads();
  if (tid < 8) sh.bps[tid] = A.ps[tid];
  __syncthreads();
  const LbdDirection dir = lbd_iteration<T>(p, sh, lds_sy, p.g4[A.g], A.f, A.t, R.n_iter + 1);
  if (tid == 0) {
    S.g_old = R.g_cur;
    S.g_cur = A.g;
    S.loss = A.f;
    S.do_lincomb = 1;
    S.need_fix = fix_pending;
    S.t_fix = A.t;
    S.do_trial = 0;
    S.do_mdot = 0;
    if (dir.gtd > -R.tol_change) {                // no descent left: the direction is formed, no step, the loop ends
      S.do_step = 0;
      S.do_eval = 0;
      S.active = 0;
      S.mode = LBD_IDLE;
    } else {                                      // the line search starts: its first trial is x0 + t d
      S.do_step = 1;
      S.do_eval = 1;
      S.mode = LBD_TRIAL;
      S.ls_phase = 1;
      S.ls_it = 0;
      S.ls_max = R.max_eval - evals_now;
      S.ls_first = 1;
      S.ls_stalled = 0;
      S.ls_nbr = 0;
      S.ls_f0 = A.f;
      S.ls_gtd0 = dir.gtd;
      LbdPoint st0;
      st0.t = 0.0;
      st0.f = A.f;
      st0.gtd = dir.gtd;
      st0.gmax = A.gmax;
#pragma unroll
      for (int c = 0; c < 8; ++c) st0.ps[c] = A.ps[c];
      st0.g = A.g;
      st0.pad_ = 0;
      S.ls_start = st0;
      S.ls_prev = st0;
      S.g_eval = lbd_free_buffer(A.g, R.g_cur, R.g_cur);
    }
  }
  if (dir.gtd > -R.tol_change) report();
}

// one optimizer.step with the line search: slots enqueued until the device reports the end, one synchronisation per batch of slots
template <typename P>
int lbd_step_ls(P& pl, LbfgsDev<float>& L, float* x, int64_t len, const float* target, specinv_lbfgs_info* info) {
  SI_CHECK(x && target && info, SPECINV_EINVAL, "null pointer");
  SI_CHECK((int64_t)pl.B() * len == L.n, SPECINV_EINVAL, "signal size does not match the optimiser's parameter vector");
  SI_CHECK(((uintptr_t)x & 15) == 0, SPECINV_EINVAL, "x is not 16-byte aligned");
  SI_TRY(lbd_grow(pl, L, L.h.max_iter));
  LbdPtrs<float> p = L.ptrs();
  if (L.pairs_y.size() > 0 && L.h.total_iters == 0) {
    void* c[2] = {L.pairs_y[0], L.pairs_s[0]};
    SI_HIP(hipMemcpy(L.cand.p, c, sizeof(c), hipMemcpyHostToDevice));
  }
  const int64_t n = L.n;
  const int nb = (int)std::min<int64_t>(1024, std::max<int64_t>(1, ceil_div(n, 256 * 8)));
  const int hist = L.h.hist;
  L.board_host[0] = 1.0;
  hipLaunchKernelGGL(k_lbd_begin_ls, dim3(1), dim3(1), 0, pl.stream, p.st);
  SI_HIP(hipGetLastError());
  fast::ObjCtl ctl{};
  ctl.do_eval = &p.st->do_eval;
  ctl.cur = &p.st->cur;
  ctl.grad_alt = p.g4[1];
  ctl.sel = &p.st->g_eval;
  ctl.tab = L.gtab.template as<float*>();
  const size_t lds = (size_t)hist * hist * sizeof(double);
  SI_HIP(hipFuncSetAttribute((const void*)k_lbd_decide_ls<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int64_t pieces = n / 4 + 1;
  const dim3 grid_x((unsigned)ceil_div(pieces, 256));
  // Slots are enqueued until the device reports the end of the step on the pinned board - at most kAhead slots beyond the one it
  // is deciding (the decision kernel reports its slot there): a step's length is not known in advance - one evaluation or forty -
  // and every slot enqueued past its end is six launches of no-ops.  Two slots ahead keep the device busy (a slot runs ~0.2 ms,
  // enqueueing one takes ~0.03); the entry evaluation is waited for: many a step ends on it (max|g| <= tolerance_grad).  The end of
  // the step costs no synchronisation of the stream either: the device has written what the host needs next to the flag.
  volatile double* seen = L.board_host;                      // (written by the device: read it every time)
  seen[1] = 0.0;
  int slot = 0;
  for (;; ++slot) {
    SI_CHECK(slot < 4096, SPECINV_ESTATE, "the line search did not end");
    const int ahead = slot == 1 ? 1 : 2;                     // (slot s goes out once slot s - ahead is decided)
    bool waited_long = false;
    for (long spin = 0; seen[0] != 0.0 && (int)seen[1] + ahead < slot + 1; ++spin) {
      __builtin_ia32_pause();
      if (spin > (1L << 34)) {                               // (minutes: the device is gone)
        waited_long = true;
        break;
      }
    }
    SI_CHECK(!waited_long, SPECINV_ESTATE, "the device did not report the decision of slot %d", slot - ahead);
    if (slot > 0 && seen[0] == 0.0) break;                   // the device has ended the step
    hipLaunchKernelGGL((k_lbd_update_x<float>), grid_x, dim3(256), 0, pl.stream, p, x, n);
    bool used = false;
    const bool timed = L.time_objective > 0 && slot < 64 && slot % L.time_objective == 0 && (size_t)(2 * slot + 1) < L.ev.size();
    if (timed) SI_HIP(hipEventRecord(L.ev[2 * slot], pl.stream));
    SI_TRY(tf_loss_grad_fused(pl, x, len, target, nullptr, p.g4[0], &used, L.loss_slot.template as<double>(), &ctl));
    SI_CHECK(used, SPECINV_EUNSUPPORTED, "the one-launch objective does not cover this configuration");
    if (timed) SI_HIP(hipEventRecord(L.ev[2 * slot + 1], pl.stream));
    hipLaunchKernelGGL((k_lbd_pair_stats<float>), dim3(nb), dim3(256), 0, pl.stream, p, n, L.part.template as<double>());
    hipLaunchKernelGGL((k_lbd_multi_dot<float>), dim3(nb), dim3(256), 0, pl.stream, p, n, L.mpart.template as<double>());
    hipLaunchKernelGGL((k_lbd_decide_ls<float>), dim3(1), dim3(256), lds, pl.stream, p, slot, (const double*)L.part.template as<double>(), nb,
                       (const double*)L.mpart.template as<double>(), (const double*)L.loss_slot.template as<double>());
    SI_HIP(hipGetLastError());
  }
  // what the step leaves behind (the board's record was complete before the flag fell)
  const double* b = L.board_host + kLbdInfo;
  L.h.first_loss = b[0];
  L.h.loss = b[1];
  L.h.t = b[2];
  L.h.total_iters = (int)b[3];
  L.h.func_evals = (int)b[4];
  L.h.n_iter = (int)b[5];
  L.h.m = (int)b[6];
  L.h.pairs_accepted = (int)b[7];
  L.h.pairs_rejected = (int)b[8];
  L.h.evals = (int)b[9];
  L.h.eval_slots = (unsigned long long)b[10] | ((unsigned long long)b[11] << 32);
  L.h.active = 0;
  if ((int)b[12] != 0) {
    // an accepted point that was not the last trial, or a direction formed without a step: one more update, unless a slot enqueued
    // past the end has carried it out already (then its decision kernel has cleared the orders, and this one finds none)
    hipLaunchKernelGGL((k_lbd_update_x<float>), grid_x, dim3(256), 0, pl.stream, p, x, n);
    hipLaunchKernelGGL(k_lbd_clear_orders, dim3(1), dim3(1), 0, pl.stream, p.st);
    SI_HIP(hipGetLastError());
  }
  if (L.time_objective) SI_HIP(hipStreamSynchronize(pl.stream));   // (benchmarks: the events are read below)
  L.accepted_seen = L.h.pairs_accepted;
  info->first_loss = L.h.first_loss;
  info->loss = L.h.loss;
  info->t = L.h.t;
  info->total_iters = L.h.total_iters;
  info->func_evals = L.h.func_evals;
  info->n_iter = L.h.n_iter;
  info->history_len = L.h.m;
  info->pairs_accepted = L.h.pairs_accepted;
  info->pairs_rejected = L.h.pairs_rejected;
  info->objective_launches = L.h.evals;
  info->objective_timed = 0;
  info->objective_ms = 0.0;
  if (L.time_objective) {
    for (int s = 0; s < std::min(slot, 64); ++s)
      if (((L.h.eval_slots >> s) & 1ull) && (size_t)(2 * s + 1) < L.ev.size() && s % L.time_objective == 0) {
        float ms = 0.0f;
        SI_HIP(hipEventElapsedTime(&ms, L.ev[2 * s], L.ev[2 * s + 1]));
        info->objective_ms += ms;
        info->objective_timed += 1;
      }
  }
  return SPECINV_OK;
}

}  // namespace specinv
