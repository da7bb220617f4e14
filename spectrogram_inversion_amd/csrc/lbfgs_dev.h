// L-BFGS with the decisions on the device: one `optimizer.step` (reference: torch_specinv/methods.py:553 ->
// torch.optim.LBFGS.step, third-party; this file: no line search - strong Wolfe: lbfgs_dev_ls.h, on the state record, the decision
// functions and the reduction kernels defined here) is ENQUEUED as a whole - objective, curvature pair + statistics, memory
// products, a one-wave decision kernel, direction + step - and the host synchronises once per step instead of once per inner
// iteration (kernels_lbfgs.h / lbfgs.py:_step_packed: ~0.05 ms of host turnaround around a 0.15 ms objective).
//
// What the host used to decide after reading the iteration's scalars now happens in `k_lbd_decide` (a single wave), in the order of
// torch.optim.LBFGS.step: the tolerance tests that end a step, the curvature guard y.s > 1e-10 and the memory update (ring of
// history_size + 1 vector slots: the candidate pair is written to the spare slot, so a rejected pair costs nothing), the Gram
// matrices s_i.y_j / y_i.y_j and the two-loop recursion on scalars (lbfgs.py:_gram_append / _gram_coefficients), t, g.d.  The
// kernels that follow read what they need from the state record in device memory: the coefficient / pointer lists of the linear
// combination, the step length, which of the two gradient buffers is current, and whether they are to run at all (after a
// break the rest of the enqueued step is a chain of no-ops).  Objective: the one-launch kernel (kernels_objective.h) only.
#pragma once
#include <cstring>
#include <memory>
#include <vector>

#include "kernels_lbfgs.h"

namespace specinv {

constexpr int kLbdMaxHist = 120;          // history_size the device path takes (Gram matrix in LDS: hist^2 doubles)
constexpr int kLbdInfo = 2;               // pinned board: [0] the step is live, [1] slots decided, [kLbdInfo ..] what the step leaves (lbfgs_dev_ls.h)
constexpr int kLbdBoard = 16;             // doubles of the board

struct LbdPoint {                         // a point of the line search: step length, loss, g.d, max|g|, the evaluation's eight sums, its gradient buffer
  double t, f, gtd, gmax, ps[8];
  int g, pad_;
};

struct LbdState {                         // device-resident; copied to the host at the end of a step
  // options
  double lr, tol_grad, tol_change;
  int max_iter, max_eval, hist;
  // torch.optim.LBFGS's state
  int total_iters, func_evals, m, seq0, cur, pairs_accepted, pairs_rejected, n_prev;
  double t, h_diag, prev_loss, loss;
  // control of the step being executed
  int active, do_lincomb, do_step, do_eval, n_iter, evals, have_prev, k_lin, k_dot, pad_;
  // the pair accepted by the last decision is FORMED by the direction kernel (y = g - g_prev, s = t_pair d_old, written to the
  // ring and used from registers): positions of y_new / s_new in the list of the linear combination, -1: no new pair
  int pair_y, pair_s;
  double t_pair;
  double first_loss, gtd;
  // reductions of the last evaluation: loss; {g.d, sum|g|, max|g|, max|d|, y.s, y.y, g.g, g.g_prev}
  double b_loss, b_ps[8];
  // ---- strong-Wolfe line search on the device (lbfgs_dev_ls.h) ------------------------------------------------------------
  int ls;                                 // option: line_search_fn = 'strong_wolfe'
  int mode;                               // what the slot that ends with the next decision carries (LbdMode)
  int do_trial, do_mdot, need_fix;        // x = x0 + t d;  memory products of gradient g_md;  ... with t_fix (the accepted point was not the last trial)
  int g_cur, g_old, g_eval, g_md;         // gradient buffers (of four): prev_flat_grad, the one before it, where the next evaluation writes, whose products
  int ls_phase, ls_it, ls_max, ls_lo, ls_hi, ls_nbr, ls_stalled, ls_first, pad3_;
  unsigned long long eval_slots;          // bit s: slot s of the step executed an evaluation (benchmarks: which event pairs count)
  double t_fix, ls_f0, ls_gtd0, ls_dnorm;
  LbdPoint ls_prev, ls_start, ls_br[2], ls_acc;
};
enum LbdMode { LBD_IDLE = 0, LBD_ENTRY = 1, LBD_TRIAL = 2, LBD_POST = 3 };

template <typename T>
struct LbdPtrs {                          // kernel argument: where everything lives
  LbdState* st;
  double* dots;        // [2 hist]  g . (ss then ys) of the last evaluation
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
  T** cand;            // [2]            where the next evaluation's y and s go
  T* gbuf[2];          // gradient ping-pong: the evaluation writes gbuf[cur ^ 1], reads gbuf[cur] as the previous gradient
  T* g4[4];            // line search: four gradient buffers (g4[0], g4[1] = gbuf), chosen by the state's g_* indices
  T* const* g4_dev;    // ... the same table in device memory (the objective kernel's ObjCtl.tab)
  T* x0;               // ... the point the line search started from
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

// Scratch of the decision kernels (one workgroup of 256): the state as it was at entry (one coalesced read instead of a chain of
// dependent global loads), the finished sums of the evaluation, the vectors of the two triangular recursions.
struct LbdShared {
  double al[kLbdMaxHist], cc[kLbdMaxHist], yq[kLbdMaxHist], sgv[kLbdMaxHist], ygv[kLbdMaxHist];
  double dotv[2 * kLbdMaxHist], bps[8], red[16], mx[2][4];
  LbdState R;
};

__device__ inline void lbd_load_state(LbdShared& sh, const LbdState* st) {
  static_assert(sizeof(LbdState) % 8 == 0 && sizeof(LbdState) / 8 <= 256, "LbdState is copied by one pass of doubles");
  const int tid = threadIdx.x;
  if (tid < (int)(sizeof(LbdState) / 8)) reinterpret_cast<double*>(&sh.R)[tid] = reinterpret_cast<const double*>(st)[tid];
  __syncthreads();
}

// The reductions an evaluation left unfinished (its kernels write per-block partial sums; finishing them here saves two
// one-workgroup launches per evaluation): sh.bps = {g.d, sum|g|, max|g|, max|d|, y.s, y.y, g.g, g.g_prev} over the nb blocks of
// k_lbd_pair_stats (fixed order), sh.dotv = the kd products of g with the memory.
__device__ inline void lbd_finish_sums(LbdShared& sh, const double* __restrict__ part_pair, int nb, const double* __restrict__ part_dot,
                                       int kd) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  double(&bps)[8] = sh.bps;
  double(&dotv)[2 * kLbdMaxHist] = sh.dotv;
  double(&red)[16] = sh.red;
  double(&mx)[2][4] = sh.mx;
  // ---- finish the evaluation's sums: {g.d, sum|g|, y.s, y.y, g.g, g.g_prev} + max|g|, max|d| over the nb blocks of
  // k_lbd_pair_dots (fixed order), and the products of g with the memory
  {
    double s6[6] = {0, 0, 0, 0, 0, 0}, m0 = 0, m1 = 0;
    for (int i = tid; i < nb; i += 256) {
#pragma unroll
      for (int c = 0; c < 6; ++c) s6[c] += part_pair[8 * i + c];
      m0 = part_pair[8 * i + 6] > m0 ? part_pair[8 * i + 6] : m0;
      m1 = part_pair[8 * i + 7] > m1 ? part_pair[8 * i + 7] : m1;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const double o0 = __shfl_xor(m0, off, 64), o1 = __shfl_xor(m1, off, 64);
      m0 = o0 > m0 ? o0 : m0;
      m1 = o1 > m1 ? o1 : m1;
    }
    if (lane == 0) {
      mx[0][wv] = m0;
      mx[1][wv] = m1;
    }
    double tot[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) tot[c] = block_sum(s6[c], red);
    if (tid == 0) {
      double r0 = 0, r1 = 0;
      for (int w = 0; w < 4; ++w) {
        r0 = mx[0][w] > r0 ? mx[0][w] : r0;
        r1 = mx[1][w] > r1 ? mx[1][w] : r1;
      }
      bps[0] = tot[0];
      bps[1] = tot[1];
      bps[2] = r0;
      bps[3] = r1;
      bps[4] = tot[2];
      bps[5] = tot[3];
      bps[6] = tot[4];
      bps[7] = tot[5];
    }
    for (int j = wv; j < kd; j += 4) {            // one wave per product
      double sj = 0.0;
      for (int i = lane; i < nb; i += 64) sj += part_dot[(int64_t)j * nb + i];
      sj = lbd_wave_sum(sj);
      if (lane == 0) dotv[j] = sj;
    }
    __syncthreads();
  }
}

struct LbdDirection {
  double gtd, t;
  int m;
};

// Iteration `total` of the optimiser begins at the gradient g (its sums in sh.bps - for total > 1 taken against the previous
// gradient and s = t_prev d - and its products with the memory in sh.dotv): the curvature guard and the memory update, the Gram
// matrices, the two-loop recursion on scalars, the coefficient / pointer lists of the direction, the step length and g.d
// (torch.optim.LBFGS.step between "compute gradient descent direction" and "compute step length").  Writes what the direction
// kernel and the next evaluation's reductions read; the caller sets the control flags.  Every branch is uniform over the
// workgroup.
template <typename T>
__device__ inline LbdDirection lbd_iteration(const LbdPtrs<T>& p, LbdShared& sh, double* lds_sy, const T* g, double loss, double t_prev,
                                             int k) {
  const LbdState& R = sh.R;
  LbdState& S = *p.st;
  double(&al)[kLbdMaxHist] = sh.al;
  double(&cc)[kLbdMaxHist] = sh.cc;
  double(&yq)[kLbdMaxHist] = sh.yq;
  double(&sgv)[kLbdMaxHist] = sh.sgv;
  double(&ygv)[kLbdMaxHist] = sh.ygv;
  double(&dotv)[2 * kLbdMaxHist] = sh.dotv;
  double(&bps)[8] = sh.bps;
  double(&red)[16] = sh.red;
  const int tid = threadIdx.x, lane = tid & 63;
  const bool w0 = (tid >> 6) == 0;
  const int hist = R.hist;
  const int total = R.total_iters + 1;
  const int m_board = R.m;                        // the evaluation's products: ss[0..m_board) then ys[0..m_board)
  int m = m_board, seq0 = R.seq0;
  double gtd, h_diag = R.h_diag;
  const double g_abssum = bps[1];
  __syncthreads();
  if (total == 1) {
    m = 0;                                        // (lbfgs.py:_forget)
    seq0 = 0;
    h_diag = 1.0;
    gtd = -bps[0];                                // the statistics were taken with d = g
    if (tid == 0) {
      p.coef[0] = -1.0;
      p.lin_ptr[0] = g;
      S.pair_y = -1;
      S.pair_s = -1;
      S.n_prev = -1;
    }
  } else {
    const double gd = bps[0], ys = bps[4], yyn = bps[5], gg = bps[6], ggp = bps[7];
    int off = 0;                                  // first product of the board that still belongs to the memory
    const bool accept = ys > 1e-10;
    int n_prev = R.n_prev;
    if (accept && m == hist) {                    // drop the oldest pair: matrices up-left, vectors of products by one
      // every thread of the workgroup moves its share, staged through the LDS scratch (lds_sy holds hist * hist doubles; the
      // recursion below fills it afterwards): two barriers per matrix instead of ~220 dependent load -> store round trips of
      // one wave at history 100 (round-3 advice)
      const int h1 = hist - 1;
      for (int pass = 0; pass < 2; ++pass) {
        double* mat = pass == 0 ? p.sy : p.yy;
        for (int e = tid; e < h1 * h1; e += 256) {
          const int r = e / h1, c = e - r * h1;
          lds_sy[e] = mat[(r + 1) * hist + c + 1];
        }
        __syncthreads();
        for (int e = tid; e < h1 * h1; e += 256) {
          const int r = e / h1, c = e - r * h1;
          mat[r * hist + c] = lds_sy[e];
        }
        __syncthreads();
      }
      for (int i = tid; i < h1; i += 256) {
        al[i] = p.rho[i + 1];
        cc[i] = p.sgp[i + 1];
        yq[i] = p.ygp[i + 1];
      }
      __syncthreads();
      for (int i = tid; i < h1; i += 256) {
        p.rho[i] = al[i];
        p.sgp[i] = cc[i];
        p.ygp[i] = yq[i];
      }
      if (n_prev > 0) n_prev -= 1;
      seq0 += 1;
      m -= 1;
      off = 1;
      __threadfence_block();
      __syncthreads();
    }
    // products of g with the memory as it stands (+ the new pair's, by linearity: s.g = t (d.g), y.g = g.g - g_prev.g)
    for (int i = tid; i < m; i += 256) {
      sgv[i] = dotv[off + i];
      ygv[i] = dotv[m_board + off + i];
    }
    __syncthreads();
    if (accept) {
      if (tid == 0) {
        p.rho[m] = 1.0 / ys;
        sgv[m] = t_prev * gd;
        ygv[m] = gg - ggp;
      }
      h_diag = ys / yyn;
      // the new Gram column (lbfgs.py:_gram_append): v . y_new = v . g - v . g_prev
      for (int i = tid; i < m; i += 256) {
        p.sy[i * hist + m] = sgv[i] - p.sgp[i];
        const double v = ygv[i] - p.ygp[i];
        p.yy[i * hist + m] = v;
        p.yy[m * hist + i] = v;
      }
      if (tid == 0) {
        p.sy[m * hist + m] = ys;
        p.yy[m * hist + m] = yyn;
      }
      m += 1;
    }
    __threadfence_block();
    __syncthreads();
    for (int i = tid; i < m; i += 256) {
      p.sgp[i] = sgv[i];
      p.ygp[i] = ygv[i];
    }
    n_prev = m;
    // ---- two-loop recursion on scalars (lbfgs.py:_gram_coefficients)
    for (int e = tid; e < m * m; e += 256) {
      const int r = e / m, c = e - r * m;
      lds_sy[e] = p.sy[r * hist + c];
    }
    __syncthreads();
    for (int i = m - 1; i >= 0; --i) {            // al_i = rho_i (s_i.g - sum_{j > i} al_j s_i.y_j)
      if (w0) {
        double part = 0.0;
        for (int j = i + 1 + lane; j < m; j += 64) part += al[j] * lds_sy[i * m + j];
        part = lbd_wave_sum(part);
        if (lane == 0) al[i] = p.rho[i] * (sgv[i] - part);
      }
      __syncthreads();
    }
    for (int i = tid; i < m; i += 256) {          // y_i . q = y_i.g - sum_j y_i.y_j al_j
      double sq = ygv[i];
      for (int j = 0; j < m; ++j) sq -= p.yy[i * hist + j] * al[j];
      yq[i] = sq;
    }
    __syncthreads();
    for (int i = 0; i < m; ++i) {                 // c_i = al_i - rho_i (gamma y_i.q + sum_{j < i} c_j s_j.y_i)
      if (w0) {
        double part = 0.0;
        for (int j = lane; j < i; j += 64) part += cc[j] * lds_sy[j * m + i];
        part = lbd_wave_sum(part);
        if (lane == 0) cc[i] = al[i] - p.rho[i] * (h_diag * yq[i] + part);
      }
      __syncthreads();
    }
    // d = -(gamma (g - sum al_j y_j) + sum c_i s_i) over [g] + ys + ss; g . d from the known products
    double part = 0.0;
    if (w0) {
      for (int i = lane; i < m; i += 64) {
        const double cy = h_diag * al[i], cs = -cc[i];
        p.coef[1 + i] = cy;
        p.coef[1 + m + i] = cs;
        p.lin_ptr[1 + i] = p.ys_slot[(seq0 + i) % (hist + 1)];
        p.lin_ptr[1 + m + i] = p.ss_slot[(seq0 + i) % (hist + 1)];
        part += cy * ygv[i] + cs * sgv[i];
      }
      part = lbd_wave_sum(part);
    }
    if (tid == 0) red[0] = part;
    __syncthreads();
    gtd = -h_diag * gg + red[0];
    if (tid == 0) {
      p.coef[0] = -h_diag;
      p.lin_ptr[0] = g;
      // an accepted pair is element m - 1 of the memory: the direction kernel forms it on the way (its slot - the spare one of
      // the ring - has not been written)
      S.pair_y = accept ? 1 + (m - 1) : -1;
      S.pair_s = accept ? 1 + m + (m - 1) : -1;
      S.t_pair = t_prev;
      S.n_prev = n_prev;
      S.pairs_accepted = R.pairs_accepted + (accept ? 1 : 0);
      S.pairs_rejected = R.pairs_rejected + (accept ? 0 : 1);
    }
  }
  // memory products of the next evaluation, and where its candidate pair goes (the spare slot of the ring)
  for (int i = tid; i < m; i += 256) {
    p.dot_ptr[i] = p.ss_slot[(seq0 + i) % (hist + 1)];
    p.dot_ptr[m + i] = p.ys_slot[(seq0 + i) % (hist + 1)];
  }
  const double t = total == 1 ? fmin(1.0, 1.0 / g_abssum) * R.lr : R.lr;
  if (tid == 0) {
#pragma unroll
    for (int c = 0; c < 8; ++c) S.b_ps[c] = bps[c];
    S.b_loss = loss;
    S.total_iters = total;
    S.n_iter = k;
    S.m = m;
    S.seq0 = seq0;
    S.h_diag = h_diag;
    S.prev_loss = loss;
    S.t = t;
    S.gtd = gtd;
    S.k_lin = 1 + 2 * m;
    S.k_dot = 2 * m;
    S.have_prev = 1;
    p.cand[0] = p.ys_slot[(seq0 + m) % (hist + 1)];
    p.cand[1] = p.ss_slot[(seq0 + m) % (hist + 1)];
  }
  return LbdDirection{gtd, t, m};
}

// The decisions of iteration k (1-based) of a step without a line search, preceded by the reductions of the evaluation: 256
// threads finish the sums, wave 0 decides; the Gram matrix s_i.y_j is staged in LDS for the two triangular recursions.  Every
// branch below is uniform over the workgroup, so all four waves reach every barrier.
template <typename T>
__global__ __launch_bounds__(256) void k_lbd_decide(LbdPtrs<T> p, int k, const double* __restrict__ part_pair, int nb,
                                                    const double* __restrict__ part_dot, const double* __restrict__ loss_slot) {
  extern __shared__ double lds_sy[];              // [m * m]
  __shared__ LbdShared sh;
  const LbdState& R = sh.R;
  LbdState& S = *p.st;                            // writes go to S
  const int tid = threadIdx.x;
  lbd_load_state(sh, p.st);
  auto stop = [&]() {                             // the step ends here: everything still enqueued for it is a no-op
    if (tid == 0) {
      S.active = 0;
      S.do_lincomb = 0;
      S.do_step = 0;
      S.do_eval = 0;
      p.board[0] = 0.0;
    }
  };
  if (!R.active) {
    stop();
    return;
  }
  lbd_finish_sums(sh, part_pair, nb, part_dot, R.k_dot);
  // ---- the tests that follow an evaluation (torch.optim.LBFGS.step: opt_cond at entry; max_eval, opt_cond, step and loss
  // tolerances at the end of an iteration)
  const double loss = *loss_slot, gmax = sh.bps[2];
  if (k == 1) {
    if (tid == 0) {
      S.first_loss = loss;
      S.loss = loss;
      S.evals = 1;
      S.func_evals = R.func_evals + 1;
    }
    if (gmax <= R.tol_grad) {
      stop();
      return;
    }
  } else {
    const int evals = R.evals + 1;
    const bool end = evals >= R.max_eval || gmax <= R.tol_grad || fabs(R.t) * sh.bps[3] <= R.tol_change ||
                     fabs(loss - R.prev_loss) < R.tol_change;
    __syncthreads();                              // (everybody has read R.evals / S.t / R.prev_loss)
    if (tid == 0) {
      S.loss = loss;
      S.evals = evals;
      S.func_evals = R.func_evals + 1;
    }
    if (end) {
      stop();                                     // (the gradient just evaluated is dropped: prev_flat_grad stays gbuf[cur])
      return;
    }
  }
  // ---- iteration k begins: the evaluated gradient is the current one
  const int cur = R.cur ^ 1;
  const LbdDirection dir = lbd_iteration<T>(p, sh, lds_sy, p.gbuf[cur], loss, R.t, k);
  if (tid == 0) {
    S.cur = cur;
    S.do_lincomb = 1;
    if (dir.gtd > -R.tol_change) {                // no descent left: the direction is formed, no step, the loop ends
      S.do_step = 0;
      S.do_eval = 0;
      S.active = 0;
      p.board[0] = 0.0;
    } else {
      S.do_step = 1;
      S.do_eval = k != R.max_iter ? 1 : 0;
      if (k == R.max_iter) {
        S.active = 0;
        p.board[0] = 0.0;
      }
    }
  }
}

// d = sum_j coef_j v_j over the device-resident list (float64 accumulation in list order, rounded once), then x += t d:
// k_lincomb with its arguments read from the state record
template <typename T>
__global__ __launch_bounds__(256) void k_lbd_lincomb_step(LbdPtrs<T> p, T* __restrict__ xs, int64_t n) {
  const LbdState& S = *p.st;
  if (!S.do_lincomb) return;
  constexpr int W = 16 / sizeof(T);
  typedef T VT __attribute__((ext_vector_type(W)));
  const int k = S.k_lin;
  const bool step = S.do_step != 0;
  const T t = (T)S.t;
  // a pair accepted by the decision just taken: y = g - g_prev and s = t_pair d_old are formed here (the same float operations
  // k_lbd_pair_stats sums over), stored to their ring slots and used from registers - the evaluation's reductions no longer write
  // a candidate pair that most decisions (BASELINE C5: every one) throw away: 67 MB per iteration less
  const int jy = S.pair_y, js = S.pair_s;
  const T* __restrict__ g = p.gbuf[S.cur];
  const T* __restrict__ gp = p.gbuf[S.cur ^ 1];
  const T tp = (T)S.t_pair;
  T* __restrict__ out = p.d;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t nv = n / W;
  if (i < nv) {
    VT yv, sv;
    if (jy >= 0) {
      const VT gv = reinterpret_cast<const VT*>(g)[i], pv = reinterpret_cast<const VT*>(gp)[i], dv = reinterpret_cast<const VT*>(out)[i];
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
    if (step) {
      VT xv = reinterpret_cast<const VT*>(xs)[i];
#pragma unroll
      for (int c = 0; c < W; ++c) xv[c] = fma(t, r[c], xv[c]);
      reinterpret_cast<VT*>(xs)[i] = xv;
    }
  } else if (i == nv) {
    for (int64_t e = nv * W; e < n; ++e) {
      T ye = T(0), se = T(0);
      if (jy >= 0) {
        ye = g[e] - gp[e];
        se = tp * out[e];
        const_cast<T*>(p.lin_ptr[jy])[e] = ye;
        const_cast<T*>(p.lin_ptr[js])[e] = se;
      }
      double s = 0.0;
      for (int j = 0; j < k; ++j) s += p.coef[j] * (double)(j == jy ? ye : j == js ? se : p.lin_ptr[j][e]);
      out[e] = (T)s;
      if (step) xs[e] = fma(t, (T)s, xs[e]);
    }
  }
}

// The evaluation's reductions: k_lbfgs_pair_stats on the state's buffers - g = gbuf[cur ^ 1] (just evaluated), g_prev = gbuf[cur],
// the sums over y = g - g_prev and s = t d without writing them; before the first iteration there is no previous gradient: statistics with d = g
// (lbfgs.py:_batch), the pair unused - and k_multi_dot over the device-resident list of memory vectors (ss then ys; it returns at
// once while the memory is empty).  Two kernels: fused into one they share its 158 registers and the streaming pass over g, g_prev,
// d runs at three waves per SIMD instead of eight (41 against 30 us at C5).  Per-block partial sums; k_lbd_decide finishes them.
constexpr int kLbdMaxVec = 2 * kLbdMaxHist;
template <typename T>
__global__ __launch_bounds__(256) void k_lbd_pair_stats(LbdPtrs<T> p, int64_t n, double* __restrict__ part) {
  const LbdState& S = *p.st;
  if (!S.do_eval) return;
  const bool have_prev = S.have_prev != 0;
  const T* __restrict__ g = S.ls ? p.g4[S.g_eval] : p.gbuf[S.cur ^ 1];
  {
    const T* __restrict__ gp = have_prev ? (S.ls ? p.g4[S.g_cur] : p.gbuf[S.cur]) : g;
    const T* __restrict__ d = have_prev ? p.d : g;
    const T t = (T)S.t;
    __shared__ double red[16];
    __shared__ double mx[2][16];
    double s[6] = {0, 0, 0, 0, 0, 0}, mg = 0, md = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
      const T gi = g[i], pi = gp[i], di = d[i];
      const T yi = gi - pi;                     // (the pair itself is formed by k_lbd_lincomb_step if the decision accepts it)
      const T si = t * di;
      const double g64 = (double)gi, d64 = (double)di, ag = fabs(g64), ad = fabs(d64);
      s[0] += g64 * d64;
      s[1] += ag;
      s[2] += (double)yi * (double)si;
      s[3] += (double)yi * (double)yi;
      s[4] += g64 * g64;
      s[5] += g64 * (double)pi;
      mg = ag > mg ? ag : mg;
      md = ad > md ? ad : md;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const double o1 = __shfl_xor(mg, off, 64), o2 = __shfl_xor(md, off, 64);
      mg = o1 > mg ? o1 : mg;
      md = o2 > md ? o2 : md;
    }
    if ((threadIdx.x & 63) == 0) {
      mx[0][threadIdx.x >> 6] = mg;
      mx[1][threadIdx.x >> 6] = md;
    }
    double tot[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) tot[c] = block_sum(s[c], red);
    if (threadIdx.x == 0) {
      double m0 = 0, m1 = 0;
      for (int w = 0; w < (int)((blockDim.x + 63) >> 6); ++w) {
        m0 = mx[0][w] > m0 ? mx[0][w] : m0;
        m1 = mx[1][w] > m1 ? mx[1][w] : m1;
      }
#pragma unroll
      for (int c = 0; c < 6; ++c) part[8 * blockIdx.x + c] = tot[c];
      part[8 * blockIdx.x + 6] = m0;
      part[8 * blockIdx.x + 7] = m1;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void k_lbd_multi_dot(LbdPtrs<T> p, int64_t n, double* __restrict__ part_dot) {
  const LbdState& S = *p.st;
  const int kk = S.k_dot;
  if (!(S.ls ? S.do_mdot : S.do_eval) || kk == 0) return;
  const T* __restrict__ g = S.ls ? p.g4[S.g_md] : p.gbuf[S.cur ^ 1];
  constexpr int W = 16 / sizeof(T);
  constexpr int Q = 8;
  typedef T VT __attribute__((ext_vector_type(W)));
  __shared__ double acc[4][kLbdMaxVec];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int j = lane; j < kk; j += 64) acc[wv][j] = 0.0;
  const int64_t nv = n / W;
  const int64_t pass = (int64_t)blockDim.x * Q;
  for (int64_t base = (int64_t)blockIdx.x * pass; base < nv; base += (int64_t)gridDim.x * pass) {
    VT gv[Q];
#pragma unroll
    for (int e = 0; e < Q; ++e) {
      const int64_t i = base + (int64_t)e * blockDim.x + threadIdx.x;
      if (i < nv) gv[e] = reinterpret_cast<const VT*>(g)[i];
      else
        for (int c = 0; c < W; ++c) gv[e][c] = T(0);
    }
    for (int j = 0; j < kk; ++j) {
      const VT* __restrict__ v = reinterpret_cast<const VT*>(p.dot_ptr[j]);
      VT vv[Q];
#pragma unroll
      for (int e = 0; e < Q; ++e) {
        const int64_t i = base + (int64_t)e * blockDim.x + threadIdx.x;
        if (i < nv) vv[e] = v[i];
        else
          for (int c = 0; c < W; ++c) vv[e][c] = T(0);
      }
      double sj = 0.0;
#pragma unroll
      for (int e = 0; e < Q; ++e)
#pragma unroll
        for (int c = 0; c < W; ++c) sj += (double)gv[e][c] * (double)vv[e][c];
      sj = wave_sum(sj);
      if (lane == 0) acc[wv][j] += sj;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    for (int j = 0; j < kk; ++j) {
      double sj = 0.0;
      for (int64_t i = nv * W; i < n; ++i) sj += (double)g[i] * (double)p.dot_ptr[j][i];
      acc[0][j] += sj;
    }
  }
  __syncthreads();
  for (int j = threadIdx.x; j < kk; j += blockDim.x)
    part_dot[(int64_t)j * gridDim.x + blockIdx.x] = ((acc[0][j] + acc[1][j]) + acc[2][j]) + acc[3][j];
}

// ---- host side ---------------------------------------------------------------------------------------------------------
template <typename T>
struct LbfgsDev {
  LbdState h{};                           // host mirror (valid after a step)
  int64_t n = 0;                          // elements of x
  FastBuf st, dots, sgp, ygp, rho, sy, yy, coef, lin_ptr, dot_ptr, ys_slot, ss_slot, cand, part, mpart, loss_slot;
  // the parameter-sized vectors (two gradients, the direction, the ring of curvature pairs) come from - and go back to - a pool
  // the plan keeps: an optimiser is created per L_BFGS call, and 2 (history + 1) + 3 hipMallocs of the parameter's size per call
  // would cost more than the step they serve
  std::vector<std::unique_ptr<FastBuf>> vecs;
  T* g0 = nullptr;
  T* g1 = nullptr;
  T* g2 = nullptr;                        // (line search: two more gradients and the starting point)
  T* g3 = nullptr;
  T* x0 = nullptr;
  FastBuf gtab;
  T* d = nullptr;
  std::vector<T*> pairs_y, pairs_s;       // host mirror of the slot tables (allocated so far)
  double* board_host = nullptr;          // pinned, device-mapped: owned here (a plan outlives many optimisers)
  double* board_dev = nullptr;
  int64_t accepted_seen = 0;
  int time_objective = 0;                 // > 0: HIP events around every time_objective-th evaluation (benchmarks)
  std::vector<hipEvent_t> ev;             // 2 per evaluation of a step
  LbfgsDev() = default;
  LbfgsDev(const LbfgsDev&) = delete;
  LbfgsDev& operator=(const LbfgsDev&) = delete;
  ~LbfgsDev() {
    if (board_host) (void)hipHostFree(board_host);
    for (hipEvent_t e : ev) (void)hipEventDestroy(e);
  }

  LbdPtrs<T> ptrs() const {
    LbdPtrs<T> p{};
    p.st = st.as<LbdState>();
    p.dots = dots.as<double>();
    p.sgp = sgp.as<double>();
    p.ygp = ygp.as<double>();
    p.rho = rho.as<double>();
    p.sy = sy.as<double>();
    p.yy = yy.as<double>();
    p.coef = coef.as<double>();
    p.lin_ptr = lin_ptr.as<const T*>();
    p.dot_ptr = dot_ptr.as<const T*>();
    p.ys_slot = ys_slot.as<T*>();
    p.ss_slot = ss_slot.as<T*>();
    p.cand = cand.as<T*>();
    p.gbuf[0] = g0;
    p.gbuf[1] = g1;
    p.g4[0] = g0;
    p.g4[1] = g1;
    p.g4[2] = g2;
    p.g4[3] = g3;
    p.g4_dev = gtab.as<T*>();
    p.x0 = x0;
    p.d = d;
    p.board = board_dev;
    return p;
  }
};

// a parameter-sized vector from the plan's pool (or a new allocation)
template <typename P>
int lbd_take(P& pl, LbfgsDev<float>& L, float** out) {
  const size_t bytes = (size_t)L.n * sizeof(float);
  std::unique_ptr<FastBuf> b;
  for (size_t i = 0; i < pl.lbd_pool.size(); ++i)
    if (pl.lbd_pool[i]->bytes >= bytes) {
      b = std::move(pl.lbd_pool[i]);
      pl.lbd_pool.erase(pl.lbd_pool.begin() + i);
      break;
    }
  if (!b) {
    b.reset(new FastBuf());
    SI_TRY(b->reserve(bytes));
  }
  *out = b->template as<float>();
  L.vecs.push_back(std::move(b));
  return SPECINV_OK;
}

template <typename P>
int lbd_create(P& pl, LbfgsDev<float>& L, int64_t n, const specinv_lbfgs_opts& o) {
  SI_CHECK(n > 0, SPECINV_EINVAL, "empty parameter vector");
  SI_CHECK(o.history_size >= 1 && o.history_size <= kLbdMaxHist, SPECINV_EUNSUPPORTED,
           "history_size %d: the device-resident optimiser takes 1 .. %d", o.history_size, kLbdMaxHist);
  SI_CHECK(o.max_iter >= 1, SPECINV_EINVAL, "max_iter must be >= 1");
  const int hist = o.history_size;
  L.n = n;
  std::memset(&L.h, 0, sizeof(L.h));
  L.h.lr = o.lr;
  L.h.tol_grad = o.tolerance_grad;
  L.h.tol_change = o.tolerance_change;
  L.h.max_iter = o.max_iter;
  L.h.max_eval = o.max_eval > 0 ? o.max_eval : o.max_iter * 5 / 4;
  L.h.hist = hist;
  L.h.h_diag = 1.0;
  L.h.n_prev = -1;
  L.h.ls = o.line_search != 0 ? 1 : 0;
  SI_TRY(L.st.reserve(sizeof(LbdState)));
  SI_TRY(L.dots.reserve((size_t)2 * hist * sizeof(double)));
  SI_TRY(L.sgp.reserve((size_t)hist * sizeof(double)));
  SI_TRY(L.ygp.reserve((size_t)hist * sizeof(double)));
  SI_TRY(L.rho.reserve((size_t)hist * sizeof(double)));
  SI_TRY(L.sy.reserve((size_t)hist * hist * sizeof(double)));
  SI_TRY(L.yy.reserve((size_t)hist * hist * sizeof(double)));
  SI_TRY(L.coef.reserve((size_t)(1 + 2 * hist) * sizeof(double)));
  SI_TRY(L.lin_ptr.reserve((size_t)(1 + 2 * hist) * sizeof(void*)));
  SI_TRY(L.dot_ptr.reserve((size_t)2 * hist * sizeof(void*)));
  SI_TRY(L.ys_slot.reserve((size_t)(hist + 1) * sizeof(void*)));
  SI_TRY(L.ss_slot.reserve((size_t)(hist + 1) * sizeof(void*)));
  SI_TRY(L.cand.reserve(2 * sizeof(void*)));
  SI_TRY(lbd_take(pl, L, &L.g0));
  SI_TRY(lbd_take(pl, L, &L.g1));
  SI_TRY(lbd_take(pl, L, &L.d));
  if (L.h.ls) {
    SI_TRY(lbd_take(pl, L, &L.g2));
    SI_TRY(lbd_take(pl, L, &L.g3));
    SI_TRY(lbd_take(pl, L, &L.x0));
    SI_TRY(L.gtab.reserve(4 * sizeof(void*)));
    float* tab[4] = {L.g0, L.g1, L.g2, L.g3};
    SI_HIP(hipMemcpy(L.gtab.p, tab, sizeof(tab), hipMemcpyHostToDevice));
  }
  SI_TRY(L.part.reserve((size_t)8 * 1024 * sizeof(double)));
  SI_TRY(L.mpart.reserve((size_t)kLbdMaxVec * 1024 * sizeof(double)));
  SI_TRY(L.loss_slot.reserve(sizeof(double)));
  SI_HIP(hipMemsetAsync(L.sy.p, 0, (size_t)hist * hist * sizeof(double), pl.stream));
  SI_HIP(hipMemsetAsync(L.yy.p, 0, (size_t)hist * hist * sizeof(double), pl.stream));
  SI_HIP(hipMemsetAsync(L.ys_slot.p, 0, (size_t)(hist + 1) * sizeof(void*), pl.stream));
  SI_HIP(hipMemsetAsync(L.ss_slot.p, 0, (size_t)(hist + 1) * sizeof(void*), pl.stream));
  SI_HIP(hipMemcpyAsync(L.st.p, &L.h, sizeof(LbdState), hipMemcpyHostToDevice, pl.stream));
  {
    void* hp = nullptr;
    void* dp = nullptr;
    // (coherent: the device's writes - "the step has ended", "slot s is decided" - are to be seen by the host while later kernels of
    // the stream still run, not when the queue drains)
    SI_HIP(hipHostMalloc(&hp, kLbdBoard * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
    L.board_host = static_cast<double*>(hp);
    std::memset(hp, 0, kLbdBoard * sizeof(double));
    SI_HIP(hipHostGetDevicePointer(&dp, hp, 0));
    L.board_dev = static_cast<double*>(dp);
  }
  L.time_objective = o.time_objective > 0 ? o.time_objective : 0;
  if (L.time_objective) {
    L.ev.resize((size_t)2 * (L.h.ls ? 64 : o.max_iter));
    for (auto& e : L.ev) SI_HIP(hipEventCreate(&e));
  }
  SI_HIP(hipStreamSynchronize(pl.stream));
  return SPECINV_OK;
}

// vector slots of the ring: the pairs accepted so far + those the step about to be enqueued can add + the spare one
template <typename P>
int lbd_grow(P& pl, LbfgsDev<float>& L, int iterations_ahead) {
  const int hist = L.h.hist;
  const size_t want = (size_t)std::min<int64_t>(hist + 1, L.accepted_seen + iterations_ahead + 1);
  const size_t have = L.pairs_y.size();
  if (have >= want) return SPECINV_OK;
  while (L.pairs_y.size() < want) {
    float* y = nullptr;
    float* sv = nullptr;
    SI_TRY(lbd_take(pl, L, &y));
    SI_TRY(lbd_take(pl, L, &sv));
    L.pairs_y.push_back(y);
    L.pairs_s.push_back(sv);
  }
  // the new table entries in one copy each (blocking: the sources are host vectors)
  SI_HIP(hipMemcpy(L.ys_slot.template as<float*>() + have, L.pairs_y.data() + have, (want - have) * sizeof(float*), hipMemcpyHostToDevice));
  SI_HIP(hipMemcpy(L.ss_slot.template as<float*>() + have, L.pairs_s.data() + have, (want - have) * sizeof(float*), hipMemcpyHostToDevice));
  return SPECINV_OK;
}

// one optimizer.step: everything enqueued, one synchronisation at the end
template <typename P>
int lbd_step(P& pl, LbfgsDev<float>& L, float* x, int64_t len, const float* target, specinv_lbfgs_info* info) {
  SI_CHECK(x && target && info, SPECINV_EINVAL, "null pointer");
  SI_CHECK((int64_t)pl.B() * len == L.n, SPECINV_EINVAL, "signal size does not match the optimiser's parameter vector");
  SI_CHECK(((uintptr_t)x & 15) == 0, SPECINV_EINVAL, "x is not 16-byte aligned");
  SI_TRY(lbd_grow(pl, L, L.h.max_iter));
  LbdPtrs<float> p = L.ptrs();
  if (L.pairs_y.size() > 0 && L.h.total_iters == 0) {
    // before the first decision the candidate slots are not set: point them at the first slot (unused until a pair exists)
    void* c[2] = {L.pairs_y[0], L.pairs_s[0]};
    SI_HIP(hipMemcpy(L.cand.p, c, sizeof(c), hipMemcpyHostToDevice));
  }
  const int64_t n = L.n;
  const int nb = (int)std::min<int64_t>(1024, std::max<int64_t>(1, ceil_div(n, 256 * 8)));
  const int hist = L.h.hist;
  L.board_host[0] = 1.0;
  hipLaunchKernelGGL(k_lbd_begin, dim3(1), dim3(1), 0, pl.stream, p.st);
  SI_HIP(hipGetLastError());
  fast::ObjCtl ctl{};
  ctl.do_eval = &p.st->do_eval;
  ctl.cur = &p.st->cur;
  ctl.grad_alt = p.gbuf[1];
  int n_eval_launched = 0;
  auto evaluate = [&]() -> int {
    bool used = false;
    const bool timed = L.time_objective > 0 && n_eval_launched % L.time_objective == 0;
    if (timed) SI_HIP(hipEventRecord(L.ev[2 * n_eval_launched], pl.stream));
    SI_TRY(tf_loss_grad_fused(pl, x, len, target, nullptr, p.gbuf[0], &used, L.loss_slot.template as<double>(), &ctl));
    SI_CHECK(used, SPECINV_EUNSUPPORTED, "the one-launch objective does not cover this configuration");
    if (timed) SI_HIP(hipEventRecord(L.ev[2 * n_eval_launched + 1], pl.stream));
    ++n_eval_launched;
    hipLaunchKernelGGL((k_lbd_pair_stats<float>), dim3(nb), dim3(256), 0, pl.stream, p, n, L.part.template as<double>());
    hipLaunchKernelGGL((k_lbd_multi_dot<float>), dim3(nb), dim3(256), 0, pl.stream, p, n, L.mpart.template as<double>());
    SI_HIP(hipGetLastError());
    return SPECINV_OK;
  };
  SI_TRY(evaluate());
  const size_t lds = (size_t)hist * hist * sizeof(double);
  SI_HIP(hipFuncSetAttribute((const void*)k_lbd_decide<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int64_t pieces = n / 4 + 1;
  for (int k = 1; k <= L.h.max_iter; ++k) {
    if (k > 1 && L.board_host[0] == 0.0) break;     // a peek at what the device has decided so far (may lag: only saves no-ops)
    hipLaunchKernelGGL((k_lbd_decide<float>), dim3(1), dim3(256), lds, pl.stream, p, k, (const double*)L.part.template as<double>(), nb,
                       (const double*)L.mpart.template as<double>(), (const double*)L.loss_slot.template as<double>());
    hipLaunchKernelGGL((k_lbd_lincomb_step<float>), dim3((unsigned)ceil_div(pieces, 256)), dim3(256), 0, pl.stream, p, x, n);
    SI_HIP(hipGetLastError());
    if (k < L.h.max_iter) SI_TRY(evaluate());
  }
  // a step cut short by the peek leaves `active` set on the device only if the device had not stopped: it had (the peek read 0)
  SI_HIP(hipMemcpyAsync(&L.h, L.st.p, sizeof(LbdState), hipMemcpyDeviceToHost, pl.stream));
  SI_HIP(hipStreamSynchronize(pl.stream));
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
    for (int i = 0; i < std::min(L.h.evals, n_eval_launched); i += L.time_objective) {   // (executed ones; gated launches are no-ops)
      float ms = 0.0f;
      SI_HIP(hipEventElapsedTime(&ms, L.ev[2 * i], L.ev[2 * i + 1]));
      info->objective_ms += ms;
      info->objective_timed += 1;
    }
  }
  return SPECINV_OK;
}

}  // namespace specinv
