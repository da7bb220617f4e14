// The device-resident optimiser's state record and the scalar decisions of its LEAN iteration (lbfgs_dev.h has the kernels and the
// host side): what the objective's epilogue (kernels_lbfgs.h) needs to take an iteration's decisions itself, in the workgroup that
// finishes last - reference: torch.optim.LBFGS.step (third-party; called from torch_specinv/methods.py:553).
#pragma once
#include "common.h"
#include "kernels_generic.h"
#include "objective_args.h"

namespace specinv {

// nine figures over the 256 threads of a workgroup, lanes then waves in a fixed order: v[0 .. 6] sums, v[7], v[8] maxima; every
// thread leaves with the totals
__device__ inline void block_reduce9(double (&v)[9], double (*red)[9]) {       // red: [waves of the workgroup][9]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 7; ++i) v[i] = wave_sum(v[i]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    v[7] = fmax(v[7], __shfl_xor(v[7], off, 64));
    v[8] = fmax(v[8], __shfl_xor(v[8], off, 64));
  }
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < 9; ++i) red[wave][i] = v[i];
  }
  __syncthreads();
  // (nine threads add the waves' rows and publish the totals: every thread walking all rows cost 144 LDS reads each - 4 us of a
  // 1024-thread workgroup)
  const int nw = (int)(blockDim.x >> 6);
  if (threadIdx.x < 9) {
    const int i = threadIdx.x;
    double t = red[0][i];
    for (int w = 1; w < nw; ++w) t = i < 7 ? t + red[w][i] : fmax(t, red[w][i]);
    red[0][i] = t;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 9; ++i) v[i] = red[0][i];
  __syncthreads();
}

constexpr int kLbdMaxHist = 120;          // history_size the device path takes (Gram matrix in LDS: hist^2 doubles)
constexpr int kLbdInfo = 2;               // pinned board: [0] the step is live, [1] slots decided, [kLbdInfo ..] what the step leaves
constexpr int kLbdBoard = 16;             // doubles of the board

struct LbdState {                         // device-resident (two buffers, see above); copied to the host at the end of a step
  // options
  double lr, tol_grad, tol_change;
  int max_iter, max_eval, hist;
  // torch.optim.LBFGS's state
  int total_iters, func_evals, m, seq0, cur, pairs_accepted, pairs_rejected, n_prev;
  double t, h_diag, prev_loss, loss;
  // control of the step being executed
  int active, do_lincomb, do_step, do_eval, n_iter, evals, have_prev, k_lin, k_dot;
  int suspended;                          // a lean chain met a non-empty memory at iteration resume_k: the host continues in the full form
  int resume_k;
  // lean iterations that accept no pair form d = (float)(c0 (double)g) and x += t d in registers and do not STORE d: whoever needs it
  // (the next evaluation's statistics, the pair s = t d of the iteration that does accept) recomputes it from the gradient it was
  // formed from - gbuf[cur], the previous gradient by then - bit for bit.  Cleared by the lean iteration that accepts a pair.
  int d_implicit;
  double c0_d;
  // The DEFERRED STEP (round 5; frame-walk objective only): the iterate lives in one of two buffers, x_sel names the one the next
  // evaluation is taken at; x_pending != 0: that buffer is not written yet - it is the other one advanced by
  // x_new = fma(t_pend, (float)(c0_pend (double)g), x_old), g = gbuf[cur] - and the next evaluation's walk forms it while it
  // loads its samples (kernels_objective_walk.h), or k_lbd_settle_x at the end of the step.  A lean iteration that accepts no pair
  // then streams nothing at all.
  int x_sel, x_pending;
  double t_pend, c0_pend;
  // the pair accepted by the last decision is FORMED by the direction kernel (y = g - g_prev, s = t_pair d_old, written to the
  // ring and used from registers): positions of y_new / s_new in the list of the linear combination, -1: no new pair
  int pair_y, pair_s;
  double t_pair;
  double first_loss, gtd;
  // reductions of the last evaluation: loss; {g.d, sum|g|, max|g|, max|d|, y.s, y.y, g.g, g.g_prev}
  double b_loss, b_ps[8];
};

template <typename T>
struct LbdPtrs {                          // kernel argument: where everything lives
  LbdState* st;        // the state record this launch reads (and, but for the lean direction kernel, writes)
  LbdState* st_next;   // lean direction kernel: the record it writes
  double* sgp;         // [hist]    s_i . g of the previous direction (lbfgs.py: _sg)
  double* ygp;         // [hist]
  double* rho;         // [hist]
  double* sy;          // [hist * hist]  s_i . y_j (i <= j)
  double* yy;          // [hist * hist]  y_i . y_j
  double* coef;        // [1 + 2 hist]   coefficients of d over [g] + ys + ss
  const T** lin_ptr;   // [1 + 2 hist]
  const T** dot_ptr;   // [2 hist]       ss then ys
  T** ys_slot;         // [hist + 1]     ring of vector slots (slot of pair number q: q mod (hist + 1))
  T** ss_slot;         // [hist + 1]
  T* gbuf[2];          // gradient ping-pong: the evaluation writes gbuf[cur ^ 1], reads gbuf[cur] as the previous gradient
  T* xbuf[2];          // the iterate's two buffers: [0] the caller's, [1] the optimiser's (deferred step; [1] == nullptr: not in use)
  T* d;
  double* board;       // pinned host memory: [0] = active (a peek, not a synchronisation)
};

// start of a step: the loop is live, the entry evaluation runs
static __global__ void k_lbd_begin(LbdState* st) {
  st->active = 1;
  st->do_eval = 1;
  st->do_lincomb = 0;
  st->do_step = 0;
  st->n_iter = 0;
  st->evals = 0;
}

__device__ inline double lbd_wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ inline void lbd_load_state(LbdState& R, const LbdState* st) {
  static_assert(sizeof(LbdState) % 8 == 0 && sizeof(LbdState) / 8 <= 256, "LbdState is copied by one pass of doubles");
  const int tid = threadIdx.x;
  if (tid < (int)(sizeof(LbdState) / 8)) reinterpret_cast<double*>(&R)[tid] = reinterpret_cast<const double*>(st)[tid];
  __syncthreads();
}

// The second level of the evaluation's reduction tree (the first: k_objective_epilogue's rows): every thread of the workgroup
// leaves with bps = {g.d, sum|g|, max|g|, max|d|, y.s, y.y, g.g, g.g_prev} and the loss, summed in a fixed order.
// SAME_LAUNCH: the rows were written by other workgroups of the launch that reads them (lbd_tail_decide) - device-scope loads,
// past the reader's own L2.
template <bool SAME_LAUNCH = false>
__device__ inline void lbd_finish_rows(double (&bps)[8], double& loss, double (*red9)[9], const double* rows, double scale) {
  double v[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int r = threadIdx.x; r < fast::kObjRows; r += blockDim.x) {
    const double* q = rows + r;                     // component-major rows (k_objective_epilogue)
    double w[9];
#pragma unroll
    for (int c = 0; c < 9; ++c)
      w[c] = SAME_LAUNCH ? __hip_atomic_load(q + c * fast::kObjRows, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : q[c * fast::kObjRows];
#pragma unroll
    for (int c = 0; c < 6; ++c) v[c] += w[c];
    v[6] += w[8];
    v[7] = fmax(v[7], w[6]);
    v[8] = fmax(v[8], w[7]);
  }
  block_reduce9(v, red9);
  bps[0] = v[0];
  bps[1] = v[1];
  bps[2] = v[7];
  bps[3] = v[8];
  bps[4] = v[2];
  bps[5] = v[3];
  bps[6] = v[4];
  bps[7] = v[5];
  loss = scale * v[6];
}

// What follows an evaluation in torch.optim.LBFGS.step - opt_cond at the entry evaluation (k == 1); max_eval, opt_cond, the step
// and loss tolerances at the end of an iteration - on the finished sums: does the step end here?  `evals`: evaluations of this
// step including this one.
__device__ inline bool lbd_step_ends(const LbdState& R, const double (&bps)[8], double loss, int k, int& evals) {
  const double gmax = bps[2];
  if (k == 1) {
    evals = 1;
    return gmax <= R.tol_grad;
  }
  evals = R.evals + 1;
  return evals >= R.max_eval || gmax <= R.tol_grad || fabs(R.t) * bps[3] <= R.tol_change || fabs(loss - R.prev_loss) < R.tol_change;
}

// ---- the lean iteration: decisions and direction in one kernel (file header) ------------------------------------------------
// What one iteration decides while the memory holds no pair when it begins: plain scalars, computed by every thread alike.
struct LbdLean {
  int stop, suspend, accept;      // the step ends at this evaluation / a memory to multiply with: not for this kernel / pair accepted
  int evals, total, m, n_prev, cur, do_step, do_eval, active;
  double loss, c0, cy, cs, t, gtd, h_diag, rho0, sg0, yg0, ys, yy;
};

// `accept_suspends`: the caller forms no pairs (the two-launch iteration, k_objective_epilogue's tail) - the iteration that would
// accept one hands over to the full form instead of the one after it.
__device__ inline LbdLean lbd_lean_decide(const LbdState& R, const double (&bps)[8], double loss, int k, bool accept_suspends = false) {
  LbdLean o{};
  o.loss = loss;
  o.stop = lbd_step_ends(R, bps, loss, k, o.evals) ? 1 : 0;
  if (o.stop) return o;
  if (R.total_iters >= 1 && R.m > 0) {            // (the iteration after this chain's first accepted pair)
    o.suspend = 1;
    return o;
  }
  o.cur = R.cur ^ 1;
  o.total = R.total_iters + 1;
  o.h_diag = R.h_diag;
  if (o.total == 1) {                             // (lbfgs.py:_forget) the statistics were taken with d = g
    o.h_diag = 1.0;
    o.c0 = -1.0;
    o.gtd = -bps[0];
    o.n_prev = -1;
    o.t = fmin(1.0, 1.0 / bps[1]) * R.lr;
  } else {
    const double gd = bps[0], ys = bps[4], yyn = bps[5], gg = bps[6], ggp = bps[7];
    o.accept = ys > 1e-10 ? 1 : 0;
    if (o.accept && accept_suspends) {            // (nothing of this iteration is committed: lbd_lean_commit)
      o.accept = 0;
      o.suspend = 1;
      return o;
    }
    double part = 0.0;
    if (o.accept) {                               // lbd_iteration with m = 1: the recursion on the one pair
      o.ys = ys;
      o.yy = yyn;
      o.rho0 = 1.0 / ys;
      o.sg0 = R.t * gd;                           // s_new . g = t_prev (d . g)
      o.yg0 = gg - ggp;                           // y_new . g = g . g - g_prev . g
      o.h_diag = ys / yyn;
      const double al0 = o.rho0 * o.sg0;
      const double yq0 = o.yg0 - yyn * al0;
      const double cc0 = al0 - o.rho0 * (o.h_diag * yq0);
      o.cy = o.h_diag * al0;
      o.cs = -cc0;
      part = o.cy * o.yg0 + o.cs * o.sg0;
      o.m = 1;
    }
    o.n_prev = o.m;
    o.c0 = -o.h_diag;
    o.gtd = -o.h_diag * gg + part;
    o.t = R.lr;
  }
  if (o.gtd > -R.tol_change) {                    // no descent left: the direction is formed, no step, the loop ends
    o.do_step = 0;
    o.do_eval = 0;
    o.active = 0;
  } else {
    o.do_step = 1;
    o.do_eval = k != R.max_iter ? 1 : 0;
    o.active = k != R.max_iter ? 1 : 0;
  }
  return o;
}


// The state record after a lean decision: N = R with this iteration's changes (ONE thread, after N has been filled with a copy of
// R).  Returns false when nothing is left to do for the caller (the step ended or is suspended here).  `defer`: a step of an
// iteration that accepts no pair is left to the next evaluation's frame walk (LbdState::x_pending).
__device__ inline bool lbd_lean_commit(LbdState& N, const LbdState& R, const LbdLean& o, const double (&bps)[8], double loss, int k,
                                       double* board, bool defer) {
  N.x_pending = 0;                                // (whatever was pending, the evaluation that brought us here has applied)
  auto off = [&]() {
    N.active = 0;
    N.do_lincomb = 0;
    N.do_step = 0;
    N.do_eval = 0;
    board[0] = 0.0;
  };
  if (o.suspend) {
    N.suspended = 1;
    N.resume_k = k;
    off();
    return false;
  }
  if (k == 1) N.first_loss = loss;
  N.loss = loss;
  N.evals = o.evals;
  N.func_evals = R.func_evals + 1;
  if (o.stop) {                                   // (the gradient just evaluated is dropped: prev_flat_grad stays gbuf[cur])
    off();
    return false;
  }
#pragma unroll
  for (int c = 0; c < 8; ++c) N.b_ps[c] = bps[c];
  N.b_loss = loss;
  N.total_iters = o.total;
  N.n_iter = k;
  N.m = o.m;
  if (o.total == 1) N.seq0 = 0;
  N.h_diag = o.h_diag;
  N.prev_loss = loss;
  N.t = o.t;
  N.gtd = o.gtd;
  N.k_lin = 1 + 2 * o.m;
  N.k_dot = 2 * o.m;
  N.have_prev = 1;
  N.n_prev = o.n_prev;
  N.cur = o.cur;
  N.pair_y = o.accept ? 1 : -1;
  N.pair_s = o.accept ? 2 : -1;
  N.t_pair = R.t;
  N.do_lincomb = 1;
  N.do_step = o.do_step;
  N.do_eval = o.do_eval;
  N.active = o.active;
  N.d_implicit = o.accept ? 0 : 1;
  N.c0_d = o.c0;
  if (o.total > 1) {
    N.pairs_accepted = R.pairs_accepted + o.accept;
    N.pairs_rejected = R.pairs_rejected + (o.accept ? 0 : 1);
  }
  if (!o.active) board[0] = 0.0;
  if (defer && !o.accept && o.do_step) {
    N.x_pending = 1;
    N.x_sel = R.x_sel ^ 1;
    N.t_pend = o.t;
    N.c0_pend = o.c0;
  }
  return true;
}

// The two-launch lean iteration: the decisions of iteration k taken by the LAST workgroup of the evaluation's epilogue to finish
// (fast::ObjDecide; the rows of all workgroups are in memory by then: write-through stores retired before the ticket, or - the
// portable form - an acquire-release ticket and an acquire fence: k_objective_epilogue's tail has the protocol).  All threads of the
// workgroup call; blockDim.x >= 256.  red9: [blockDim.x / 64][9].  The record is read by every workgroup when the launch begins
// (lbd_tail_preload: the last one has it in LDS when it needs it).
__device__ inline void lbd_tail_decide(const fast::ObjDecide& q, double scale, double (*red9)[9], LbdState& R /* shared */) {
  LbdState& N = *static_cast<LbdState*>(q.st_next);            // (R: read when the launch began, lbd_tail_preload)
  double bps[8], loss;
  lbd_finish_rows<true>(bps, loss, red9, q.rows, scale);
  const LbdLean o = lbd_lean_decide(R, bps, loss, q.k, true);
  if (threadIdx.x < (int)(sizeof(LbdState) / 8)) reinterpret_cast<double*>(&N)[threadIdx.x] = reinterpret_cast<const double*>(&R)[threadIdx.x];
  __threadfence_block();
  __syncthreads();
  if (threadIdx.x == 0) lbd_lean_commit(N, R, o, bps, loss, q.k, q.board, true);
}

__device__ inline void lbd_tail_preload(const fast::ObjDecide& q, LbdState& R /* shared; a barrier before it is read */) {
  if (threadIdx.x < (int)(sizeof(LbdState) / 8)) reinterpret_cast<double*>(&R)[threadIdx.x] = reinterpret_cast<const double*>(q.st)[threadIdx.x];
}

// ... the evaluation was gated off (the step is over): the record is handed on unchanged
__device__ inline void lbd_tail_pass(const fast::ObjDecide& q) {
  if (threadIdx.x < (int)(sizeof(LbdState) / 8))
    reinterpret_cast<double*>(q.st_next)[threadIdx.x] = reinterpret_cast<const double*>(q.st)[threadIdx.x];
}

}  // namespace specinv
