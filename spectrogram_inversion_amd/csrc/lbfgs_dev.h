// L-BFGS with the decisions on the device: one `optimizer.step` (reference: torch_specinv/methods.py:553 ->
// torch.optim.LBFGS.step, third-party; no line search - strong Wolfe runs on the host-driven packed loop, lbfgs.py) is ENQUEUED as a
// whole and the host synchronises once per step instead of once per inner iteration (kernels_lbfgs.h / lbfgs.py:_step_packed:
// ~0.05 ms of host turnaround around a 0.15 ms objective).
//
// An inner iteration is TWO launches while the memory is empty and the frame walk (kernels_objective_walk.h) serves the objective:
//   k_objective_walk      loss + gradient; a step x += t d still pending from the iteration before is applied on the way - the walk
//                         forms x_new = fma(t, (float)(c0 (double)g_prev), x_old) as it loads its samples and writes it to the
//                         iterate's other buffer (LbdState::x_pending; the step left pending at the end of a step: k_lbd_settle_x)
//   k_objective_epilogue  seams, margins, statistics, rows - and its workgroup that finishes LAST takes the iteration's decisions
//                         (lbfgs_state.h: lbd_tail_decide) and writes the other state record.  An iteration that accepts a pair
//                         suspends the chain instead: the full form takes it from there.
// Elsewhere (tile kernel, magnitude objective; SPECINV_LBFGS_DEFER=0 / SPECINV_LBFGS_LEAN2=0) THREE launches (six in round 4):
//   k_objective_logmel    loss + gradient, and the statistics of the new gradient - {g.d, sum|g|, y.s, y.y, g.g, g.g_prev, max|g|,
//                         max|d|} - taken where each sample becomes final (ObjArgs::st_*), not by a pass of their own
//   k_objective_epilogue  seams, margins, their share of the statistics, and the first level of the reduction: kObjRows rows
//   k_lbd_direction_lean  every workgroup finishes the rows and takes the iteration's decisions itself (scalars: the tolerance
//                         tests that end a step, the curvature guard y.s > 1e-10, a memory of at most the ONE pair this iteration
//                         accepts), then forms its share of d and x += t d; workgroup 0 writes the state record - into the OTHER of
//                         two buffers, so that no workgroup reads what another has already replaced.
// With pairs in the memory (BASELINE C5 never gets there: every pair fails the guard) an iteration needs the products of g with
// them and the two-loop recursion, and takes the full form: k_lbd_multi_dot, k_lbd_decide (one workgroup: Gram matrices s_i.y_j /
// y_i.y_j and the recursion on scalars, lbfgs.py:_gram_append / _gram_coefficients), k_lbd_lincomb_step.  The host picks the form
// at the start of a step from the memory length it last read; a lean chain that meets a non-empty memory (the iteration after
// its first accepted pair) SUSPENDS the step - the rest of the chain runs as no-ops - and the host resumes it in the full form.
// The kernels read what they need from the state record in device memory: coefficient / pointer lists, step length, which of the
// two gradient buffers is current, whether they are to run at all (after a break the rest of the enqueued step is a chain of
// no-ops).  Objective: the one-launch kernel (kernels_objective.h) only.
#pragma once
#include <cstring>
#include <memory>
#include <vector>

#include "kernels_lbfgs.h"

namespace specinv {

// Scratch of the decision kernels (one workgroup of 256): the state as it was at entry (one coalesced read instead of a chain of
// dependent global loads), the finished sums of the evaluation, the vectors of the two triangular recursions.
struct LbdShared {
  double al[kLbdMaxHist], cc[kLbdMaxHist], yq[kLbdMaxHist], sgv[kLbdMaxHist], ygv[kLbdMaxHist];
  double dotv[2 * kLbdMaxHist], bps[8], red[16], red9[4][9], loss;
  LbdState R;
};

// ... and the products of g with the memory, from k_lbd_multi_dot's per-block partial sums: one wave per product
__device__ inline void lbd_finish_dots(LbdShared& sh, const double* __restrict__ part_dot, int nb, int kd) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (int j = wv; j < kd; j += 4) {
    double sj = 0.0;
    for (int i = lane; i < nb; i += 64) sj += part_dot[(int64_t)j * nb + i];
    sj = lbd_wave_sum(sj);
    if (lane == 0) sh.dotv[j] = sj;
  }
  __syncthreads();
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
  // memory products of the next evaluation
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
  }
  return LbdDirection{gtd, t, m};
}

// The decisions of iteration k (1-based) of a step with pairs in the memory, preceded by the second level of the evaluation's
// reductions: 256 threads finish the sums, wave 0 decides; the Gram matrix s_i.y_j is staged in LDS for the two triangular
// recursions.  Every branch below is uniform over the workgroup, so all four waves reach every barrier.
template <typename T>
__global__ __launch_bounds__(256) void k_lbd_decide(LbdPtrs<T> p, int k, const double* __restrict__ rows, double scale,
                                                    const double* __restrict__ part_dot, int nb) {
  extern __shared__ double lds_sy[];              // [m * m]
  __shared__ LbdShared sh;
  const LbdState& R = sh.R;
  LbdState& S = *p.st;                            // writes go to S
  const int tid = threadIdx.x;
  lbd_load_state(sh.R, p.st);
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
  lbd_finish_rows(sh.bps, sh.loss, sh.red9, rows, scale);
  lbd_finish_dots(sh, part_dot, nb, R.k_dot);
  const double loss = sh.loss;
  int evals;
  const bool end = lbd_step_ends(R, sh.bps, loss, k, evals);
  __syncthreads();                                // (everybody has read R.evals / R.t / R.prev_loss)
  if (tid == 0) {
    if (k == 1) S.first_loss = loss;
    S.loss = loss;
    S.evals = evals;
    S.func_evals = R.func_evals + 1;
  }
  if (end) {
    stop();                                       // (the gradient just evaluated is dropped: prev_flat_grad stays gbuf[cur])
    return;
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
__global__ __launch_bounds__(256) void k_lbd_lincomb_step(LbdPtrs<T> p, int64_t n) {
  const LbdState& S = *p.st;
  if (!S.do_lincomb) return;
  T* __restrict__ xs = p.xbuf[p.xbuf[1] ? S.x_sel : 0];
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

constexpr int kLbdLeanItems = 2;                  // 16-byte pieces per thread and trip of the lean direction kernel

template <typename T>
__global__ __launch_bounds__(256) void k_lbd_direction_lean(LbdPtrs<T> p, int k, int64_t n, const double* __restrict__ rows,
                                                            double scale, int defer_ok) {
  __shared__ LbdState R;
  __shared__ double red9[4][9];
  const int tid = threadIdx.x;
  const bool writer = blockIdx.x == 0 && tid == 0;
  lbd_load_state(R, p.st);
  LbdState& N = *p.st_next;
  if (!R.active) {                                // the step is over (or suspended): hand the record on unchanged
    if (blockIdx.x == 0 && tid < (int)(sizeof(LbdState) / 8))
      reinterpret_cast<double*>(&N)[tid] = reinterpret_cast<const double*>(&R)[tid];
    return;
  }
  // the new gradient and x of this thread's first trip are requested BEFORE the rows are finished and the decisions taken: they
  // do not depend on them (if the step ends here they were read for nothing), and their latency covers the head's
  constexpr int W = 16 / sizeof(T);
  typedef T VT __attribute__((ext_vector_type(W)));
  const T* __restrict__ g = p.gbuf[R.cur ^ 1];
  const T* __restrict__ gp = p.gbuf[R.cur];
  T* __restrict__ out = p.d;
  // the iterate this iteration's evaluation was taken at (a pending step has been applied by that evaluation's walk)
  T* __restrict__ xs = p.xbuf[R.x_sel];
  const int64_t nv = n / W;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  VT ng[kLbdLeanItems], nx[kLbdLeanItems];
  auto request = [&](int64_t base) {
#pragma unroll
    for (int e = 0; e < kLbdLeanItems; ++e) {
      const int64_t i = base + e * stride;
      if (i < nv) {
        ng[e] = reinterpret_cast<const VT*>(g)[i];
        nx[e] = reinterpret_cast<const VT*>(xs)[i];
      }
    }
  };
  const int64_t base0 = (int64_t)blockIdx.x * blockDim.x + tid;
  if (!defer_ok) request(base0);                  // (with the deferred step most iterations stream nothing: no speculative loads)
  double bps[8], loss;
  lbd_finish_rows(bps, loss, red9, rows, scale);
  const LbdLean o = lbd_lean_decide(R, bps, loss, k);
  if (blockIdx.x == 0) {
    // the record of the next launches: a copy with this iteration's changes (one thread; everybody has read R long before any
    // later kernel reads N)
    __syncthreads();
    if (tid < (int)(sizeof(LbdState) / 8)) reinterpret_cast<double*>(&N)[tid] = reinterpret_cast<const double*>(&R)[tid];
    __threadfence_block();
    __syncthreads();
  }
  bool go = true;
  if (writer) go = lbd_lean_commit(N, R, o, bps, loss, k, p.board, defer_ok != 0);
  if (o.suspend || o.stop) return;
  const int hist = R.hist;
  const int slot = R.seq0 % (hist + 1);           // (the memory is empty: the new pair is number seq0 of the ring)
  if (writer && o.accept) {                       // what the full form finds when it takes over: Gram entries, products, lists
    p.rho[0] = o.rho0;
    p.sgp[0] = o.sg0;
    p.ygp[0] = o.yg0;
    p.sy[0] = o.ys;
    p.yy[0] = o.yy;
    p.dot_ptr[0] = p.ss_slot[slot];
    p.dot_ptr[1] = p.ys_slot[slot];
  }
  (void)go;
  const bool acc = o.accept != 0, step = o.do_step != 0;
  // ---- the deferred step: no pair to form and store - d = (float)(c0 g) stays implicit - so the only thing left to stream would be
  // x += t d, and the next evaluation's walk does that on its way (LbdState::x_pending)
  if (defer_ok && !acc) return;                   // (lbd_lean_commit has recorded the pending step)
  if (defer_ok) request(base0);
  // ---- d = c0 g (+ cy y + cs s of the pair just accepted, formed here and stored to its ring slot), x += t d: the float
  // operations of k_lbd_lincomb_step over the list [g, y, s]
  const T tp = (T)R.t, t = (T)o.t;
  T* ysl = acc ? p.ys_slot[slot] : nullptr;
  T* ssl = acc ? p.ss_slot[slot] : nullptr;
  for (int64_t base = base0; base < nv; base += kLbdLeanItems * stride) {
    VT gv[kLbdLeanItems], xv[kLbdLeanItems], pv[kLbdLeanItems], dv[kLbdLeanItems];
#pragma unroll
    for (int e = 0; e < kLbdLeanItems; ++e) {
      gv[e] = ng[e];
      xv[e] = nx[e];
      const int64_t i = base + e * stride;
      if (acc && i < nv) {
        pv[e] = reinterpret_cast<const VT*>(gp)[i];
        if (R.d_implicit) {
#pragma unroll
          for (int c = 0; c < W; ++c) dv[e][c] = (T)(R.c0_d * (double)pv[e][c]);
        } else {
          dv[e] = reinterpret_cast<const VT*>(out)[i];
        }
      }
    }
    if (base + kLbdLeanItems * stride < nv) request(base + kLbdLeanItems * stride);     // the next trip's, a trip ahead
#pragma unroll
    for (int e = 0; e < kLbdLeanItems; ++e) {
      const int64_t i = base + e * stride;
      if (i < nv) {
        VT r;
        if (acc) {
          VT yv, sv;
#pragma unroll
          for (int c = 0; c < W; ++c) {
            yv[c] = gv[e][c] - pv[e][c];
            sv[c] = tp * dv[e][c];
            double s = 0.0;
            s += o.c0 * (double)gv[e][c];
            s += o.cy * (double)yv[c];
            s += o.cs * (double)sv[c];
            r[c] = (T)s;
          }
          reinterpret_cast<VT*>(ysl)[i] = yv;
          reinterpret_cast<VT*>(ssl)[i] = sv;
        } else {
#pragma unroll
          for (int c = 0; c < W; ++c) {
            double s = 0.0;
            s += o.c0 * (double)gv[e][c];
            r[c] = (T)s;
          }
        }
        if (acc) reinterpret_cast<VT*>(out)[i] = r;     // (otherwise d stays implicit: LbdState::d_implicit)
        if (step) {
#pragma unroll
          for (int c = 0; c < W; ++c) xv[e][c] = fma(t, r[c], xv[e][c]);
          reinterpret_cast<VT*>(xs)[i] = xv[e];
        }
      }
    }
  }
  if (blockIdx.x == 0 && tid == 0) {
    for (int64_t e = nv * W; e < n; ++e) {
      T ye = T(0), se = T(0);
      if (acc) {
        ye = g[e] - gp[e];
        se = tp * (R.d_implicit ? (T)(R.c0_d * (double)gp[e]) : out[e]);
        ysl[e] = ye;
        ssl[e] = se;
      }
      double s = 0.0;
      s += o.c0 * (double)g[e];
      if (acc) {
        s += o.cy * (double)ye;
        s += o.cs * (double)se;
      }
      if (acc) out[e] = (T)s;
      if (step) xs[e] = fma(t, (T)s, xs[e]);
    }
  }
}

// The products of the evaluated gradient g = gbuf[cur ^ 1] with the memory (ss then ys, the device-resident list): per-block
// partial sums, finished by k_lbd_decide.  Full form only (the lean chain never has a memory to multiply with).
constexpr int kLbdMaxVec = 2 * kLbdMaxHist;
template <typename T>
__global__ __launch_bounds__(256) void k_lbd_multi_dot(LbdPtrs<T> p, int64_t n, double* __restrict__ part_dot) {
  const LbdState& S = *p.st;
  const int kk = S.k_dot;
  if (!S.active || kk == 0) return;
  const T* __restrict__ g = p.gbuf[S.cur ^ 1];
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

// an implicit direction made explicit (the full form reads d from memory): d = (float)(c0_d (double)gbuf[cur])
template <typename T>
__global__ __launch_bounds__(256) void k_lbd_materialise_d(LbdPtrs<T> p, int64_t n) {
  LbdState& S = *p.st;
  if (!S.d_implicit) return;
  const T* __restrict__ g = p.gbuf[S.cur];
  const double c0 = S.c0_d;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p.d[i] = (T)(c0 * (double)g[i]);
}
static __global__ void k_lbd_clear_implicit(LbdState* st) { st->d_implicit = 0; }

// End of a step: the iterate back in the caller's buffer - a pending step applied (xbuf[0] = fma(t, (float)(c0 g), old), the old
// iterate being xbuf[x_sel ^ 1], possibly xbuf[0] itself), or a plain copy if the current iterate sits in the optimiser's buffer.
template <typename T>
__global__ __launch_bounds__(256) void k_lbd_settle_x(LbdPtrs<T> p, int64_t n) {
  const LbdState& S = *p.st;
  const int pending = S.x_pending, sel = S.x_sel;
  if (!pending && sel == 0) return;
  const T* __restrict__ src = p.xbuf[pending ? sel ^ 1 : sel];
  T* __restrict__ dst = p.xbuf[0];
  const T* __restrict__ g = p.gbuf[S.cur];
  const T t = (T)S.t_pend;
  const double c0 = S.c0_pend;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = pending ? fma(t, (T)(c0 * (double)g[i]), src[i]) : src[i];
}
static __global__ void k_lbd_settled(LbdState* st) {
  st->x_pending = 0;
  st->x_sel = 0;
}

// ---- host side ---------------------------------------------------------------------------------------------------------
template <typename T>
struct LbfgsDev {
  LbdState h{};                           // host mirror (valid after a step)
  int64_t n = 0;                          // elements of x
  int par = 0;                            // which of the two state records is current
  FastBuf st, sgp, ygp, rho, sy, yy, coef, lin_ptr, dot_ptr, ys_slot, ss_slot, rows, mpart;
  // the parameter-sized vectors (two gradients, the direction, the ring of curvature pairs) come from - and go back to - a pool
  // the plan keeps: an optimiser is created per L_BFGS call, and 2 (history + 1) + 3 hipMallocs of the parameter's size per call
  // would cost more than the step they serve
  std::vector<std::unique_ptr<FastBuf>> vecs;
  T* g0 = nullptr;
  T* g1 = nullptr;
  T* d = nullptr;
  FastBuf ticket;                         // the two-launch lean iteration: workgroups of the epilogue done (fast::ObjDecide)
  T* x_user = nullptr;                    // the caller's iterate (set per step) and the optimiser's second buffer (deferred step)
  T* x_alt = nullptr;
  std::vector<T*> pairs_y, pairs_s;       // host mirror of the slot tables (allocated so far)
  double* board_host = nullptr;          // pinned, device-mapped: owned here (a plan outlives many optimisers)
  double* board_dev = nullptr;
  int64_t accepted_seen = 0;
  int time_objective = 0;                 // > 0: HIP events around every time_objective-th evaluation (benchmarks)
  std::vector<hipEvent_t> ev;             // 2 per evaluation of a step
  int lean_launches = 0, full_launches = 0, suspensions = 0;   // diagnostics (tests: which form ran)
  bool defer_live = false;                // the step being enqueued defers its x += t d (the iterate may sit in x_alt until it is settled)
  bool broken = false;                    // a step failed half-way: the record on the device no longer matches the caller's x
  LbfgsDev() = default;
  LbfgsDev(const LbfgsDev&) = delete;
  LbfgsDev& operator=(const LbfgsDev&) = delete;
  ~LbfgsDev() {
    if (board_host) (void)hipHostFree(board_host);
    for (hipEvent_t e : ev) (void)hipEventDestroy(e);
  }

  LbdState* state(int which) const { return st.as<LbdState>() + which; }
  LbdPtrs<T> ptrs() const {
    LbdPtrs<T> p{};
    p.st = state(par);
    p.st_next = state(par ^ 1);
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
    p.gbuf[0] = g0;
    p.gbuf[1] = g1;
    p.xbuf[0] = x_user;
    p.xbuf[1] = x_alt;
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
  L.par = 0;
  std::memset(&L.h, 0, sizeof(L.h));
  L.h.lr = o.lr;
  L.h.tol_grad = o.tolerance_grad;
  L.h.tol_change = o.tolerance_change;
  L.h.max_iter = o.max_iter;
  L.h.max_eval = o.max_eval > 0 ? o.max_eval : o.max_iter * 5 / 4;
  L.h.hist = hist;
  L.h.h_diag = 1.0;
  L.h.n_prev = -1;
  L.h.pair_y = -1;
  L.h.pair_s = -1;
  SI_TRY(L.st.reserve(2 * sizeof(LbdState)));
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
  SI_TRY(lbd_take(pl, L, &L.g0));
  SI_TRY(lbd_take(pl, L, &L.g1));
  SI_TRY(lbd_take(pl, L, &L.d));
  SI_TRY(lbd_take(pl, L, &L.x_alt));
  SI_TRY(L.rows.reserve((size_t)fast::kObjRows * fast::kObjStatRow * sizeof(double)));
  SI_TRY(L.mpart.reserve((size_t)kLbdMaxVec * 1024 * sizeof(double)));
  SI_TRY(L.ticket.reserve(16));
  SI_HIP(hipMemsetAsync(L.ticket.p, 0, 16, pl.stream));
  SI_HIP(hipMemsetAsync(L.sy.p, 0, (size_t)hist * hist * sizeof(double), pl.stream));
  SI_HIP(hipMemsetAsync(L.yy.p, 0, (size_t)hist * hist * sizeof(double), pl.stream));
  SI_HIP(hipMemsetAsync(L.ys_slot.p, 0, (size_t)(hist + 1) * sizeof(void*), pl.stream));
  SI_HIP(hipMemsetAsync(L.ss_slot.p, 0, (size_t)(hist + 1) * sizeof(void*), pl.stream));
  SI_HIP(hipMemcpyAsync(L.state(0), &L.h, sizeof(LbdState), hipMemcpyHostToDevice, pl.stream));
  SI_HIP(hipMemcpyAsync(L.state(1), &L.h, sizeof(LbdState), hipMemcpyHostToDevice, pl.stream));
  {
    void* hp = nullptr;
    void* dp = nullptr;
    // (coherent: the device's write - "the step has ended" - is to be seen by the host while later kernels of the stream still
    // run, not when the queue drains)
    SI_HIP(hipHostMalloc(&hp, kLbdBoard * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
    L.board_host = static_cast<double*>(hp);
    std::memset(hp, 0, kLbdBoard * sizeof(double));
    SI_HIP(hipHostGetDevicePointer(&dp, hp, 0));
    L.board_dev = static_cast<double*>(dp);
  }
  L.time_objective = o.time_objective > 0 ? o.time_objective : 0;
  if (L.time_objective) {
    L.ev.resize((size_t)2 * (2 * o.max_iter + 2));     // (a suspended chain's evaluations are launched twice: lean no-ops, then full)
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

// one optimizer.step: everything enqueued, one synchronisation at the end (one more if a lean chain is suspended)
template <typename P>
int lbd_step_run(P& pl, LbfgsDev<float>& L, float* x, int64_t len, const float* target, specinv_lbfgs_info* info);

// ... and what a failure in the middle of it leaves behind: with the step deferred the current iterate may sit in the optimiser's
// own buffer (LbdState::x_sel / x_pending) - it is brought back to the caller's x, and the optimiser refuses further steps (its
// record and the enqueued chain no longer describe one consistent iteration).
template <typename P>
int lbd_step(P& pl, LbfgsDev<float>& L, float* x, int64_t len, const float* target, specinv_lbfgs_info* info) {
  SI_CHECK(!L.broken, SPECINV_ESTATE, "an earlier step of this optimiser failed: create a new one");
  L.defer_live = false;
  const int rc = lbd_step_run(pl, L, x, len, target, info);
  if (rc != SPECINV_OK && L.defer_live) {
    const std::string why = last_error();            // (the settle's own HIP calls must not replace the message)
    (void)hipGetLastError();
    hipLaunchKernelGGL((k_lbd_settle_x<float>), dim3(1024), dim3(256), 0, pl.stream, L.ptrs(), L.n);
    hipLaunchKernelGGL(k_lbd_settled, dim3(1), dim3(1), 0, pl.stream, L.state(L.par));
    (void)hipStreamSynchronize(pl.stream);
    (void)hipGetLastError();
    last_error() = why;
  }
  if (rc != SPECINV_OK) L.broken = true;
  L.defer_live = false;
  return rc;
}

template <typename P>
int lbd_step_run(P& pl, LbfgsDev<float>& L, float* x, int64_t len, const float* target, specinv_lbfgs_info* info) {
  SI_CHECK(x && target && info, SPECINV_EINVAL, "null pointer");
  SI_CHECK((int64_t)pl.B() * len == L.n, SPECINV_EINVAL, "signal size does not match the optimiser's parameter vector");
  SI_CHECK(((uintptr_t)x & 15) == 0, SPECINV_EINVAL, "x is not 16-byte aligned");
  SI_TRY(lbd_grow(pl, L, L.h.max_iter));
  const int64_t n = L.n;
  const int nb = (int)std::min<int64_t>(1024, std::max<int64_t>(1, ceil_div(n, 256 * 8)));
  const int hist = L.h.hist;
  const double scale = 1.0 / ((double)pl.B() * pl.Tn() * (pl.tf_kind == SPECINV_TF_MAG ? pl.n_freq : pl.tf_mels));
  L.board_host[0] = 1.0;
  hipLaunchKernelGGL(k_lbd_begin, dim3(1), dim3(1), 0, pl.stream, L.state(L.par));
  SI_HIP(hipGetLastError());
  // the deferred step: only where the frame walk serves the objective (it is the kernel that applies the step)
  bool defer = tf_walk_serves(pl, len) && ((uintptr_t)x & 7) == 0;
  if (const char* e = getenv("SPECINV_LBFGS_DEFER")) {
    if (e[0] == '0') defer = false;                 // (tests / A-B runs)
  }
  L.x_user = x;
  L.defer_live = defer;
  // evaluate() launches that EXECUTED, as index ranges [lo, hi): after a suspended lean chain the no-op launches of its tail sit
  // between the chain's executed evaluations and the resumed ones (only executed launches are summed into objective_ms)
  std::vector<std::pair<int, int>> exec_ranges;
  int exec_lo = 0, exec_counted = 0;
  int tail_fence = 1;                               // the epilogue's rows handed over by release / acquire (0: round 5's write-through protocol)
  if (const char* e = getenv("SPECINV_LBFGS_TAIL_FENCE")) tail_fence = e[0] != '0';
  int n_eval_launched = 0;
  // objective + epilogue on the record that is current NOW; k_decide > 0: the epilogue also takes iteration k_decide's decisions
  // (the two-launch lean iteration) and writes the OTHER record
  auto evaluate = [&](int k_decide) -> int {
    LbdState* st = L.state(L.par);
    fast::ObjCtl ctl{};
    ctl.do_eval = &st->do_eval;
    ctl.cur = &st->cur;
    ctl.grad_alt = L.g1;
    if (defer) {
      ctl.x_alt = L.x_alt;
      ctl.x_sel = &st->x_sel;
      ctl.x_pending = &st->x_pending;
      ctl.t_pend = &st->t_pend;
      ctl.c0_pend = &st->c0_pend;
    }
    fast::ObjStatReq sr{};
    sr.d = L.d;
    sr.have = &st->have_prev;
    sr.t_dev = &st->t;
    sr.d_implicit = &st->d_implicit;
    sr.c0_d = &st->c0_d;
    sr.rows = L.rows.template as<double>();
    bool used = false;
    const bool timed = L.time_objective > 0 && n_eval_launched % L.time_objective == 0 && (size_t)(2 * n_eval_launched + 1) < L.ev.size();
    if (timed) SI_HIP(hipEventRecord(L.ev[2 * n_eval_launched], pl.stream));
    fast::ObjDecide dec{};
    if (k_decide > 0) {
      dec.ticket = L.ticket.template as<unsigned>();
      dec.st = st;
      dec.st_next = L.state(L.par ^ 1);
      dec.board = L.board_dev;
      dec.rows = sr.rows;
      dec.k = k_decide;
      dec.fence = tail_fence;
    }
    SI_TRY(tf_loss_grad_fused(pl, x, len, target, nullptr, L.g0, &used, nullptr, &ctl, &sr, k_decide > 0 ? &dec : nullptr));
    SI_CHECK(used, SPECINV_EUNSUPPORTED, "the one-launch objective does not cover this configuration");
    if (timed) SI_HIP(hipEventRecord(L.ev[2 * n_eval_launched + 1], pl.stream));
    ++n_eval_launched;
    return SPECINV_OK;
  };
  const size_t lds = (size_t)hist * hist * sizeof(double);
  SI_HIP(hipFuncSetAttribute((const void*)k_lbd_decide<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int64_t pieces = n / 4 + 1;
  const int lean_grid = (int)std::min<int64_t>(1024, std::max<int64_t>(1, ceil_div(pieces, 256 * kLbdLeanItems)));
  const double* rows = L.rows.template as<double>();
  bool lean = L.h.m == 0;                           // the memory as the host last saw it (a fresh optimiser: empty)
  if (const char* e = getenv("SPECINV_LBFGS_LEAN")) {
    if (e[0] == '0') lean = false;                  // (tests / A-B runs: the full form from the first iteration)
  }
  if (!lean && L.h.d_implicit) {                    // (only when the form is forced: a lean chain clears the flag before it hands over)
    hipLaunchKernelGGL((k_lbd_materialise_d<float>), dim3(1024), dim3(256), 0, pl.stream, L.ptrs(), n);
    hipLaunchKernelGGL(k_lbd_clear_implicit, dim3(1), dim3(1), 0, pl.stream, L.state(L.par));
    SI_HIP(hipGetLastError());
  }
  // the lean iteration in two launches: with the step deferred into the frame walk nothing is left to stream unless a pair is
  // accepted, and that iteration hands over to the full form - the decisions ride in the epilogue (lbd_tail_decide)
  bool lean2 = lean && defer;
  if (const char* e = getenv("SPECINV_LBFGS_LEAN2")) {
    if (e[0] == '0') lean2 = false;
  }
  if (!lean2) SI_TRY(evaluate(0));                  // (the entry evaluation; lean2: iteration 1's evaluation carries its decision)
  int k = 1, k_first = 1;
  for (;;) {
    for (; k <= L.h.max_iter; ++k) {
      if (k > k_first && L.board_host[0] == 0.0) break;   // a peek at what the device has decided so far (may lag: only saves no-ops)
      LbdPtrs<float> p = L.ptrs();
      if (lean && lean2) {
        SI_TRY(evaluate(k));
        L.par ^= 1;
        ++L.lean_launches;
        continue;
      }
      if (lean) {
        hipLaunchKernelGGL((k_lbd_direction_lean<float>), dim3(lean_grid), dim3(256), 0, pl.stream, p, k, n, rows, scale,
                           defer ? 1 : 0);
        L.par ^= 1;
        ++L.lean_launches;
      } else {
        hipLaunchKernelGGL((k_lbd_multi_dot<float>), dim3(nb), dim3(256), 0, pl.stream, p, n, L.mpart.template as<double>());
        hipLaunchKernelGGL((k_lbd_decide<float>), dim3(1), dim3(256), lds, pl.stream, p, k, rows, scale,
                           (const double*)L.mpart.template as<double>(), nb);
        hipLaunchKernelGGL((k_lbd_lincomb_step<float>), dim3((unsigned)ceil_div(pieces, 256)), dim3(256), 0, pl.stream, p, n);
        ++L.full_launches;
      }
      SI_HIP(hipGetLastError());
      if (k < L.h.max_iter) SI_TRY(evaluate(0));
    }
    // a step cut short by the peek leaves `active` set on the device only if the device had not stopped: it had (the peek read 0)
    if (defer) {                                    // the iterate back in the caller's buffer (a suspended chain: its last one)
      hipLaunchKernelGGL((k_lbd_settle_x<float>), dim3(1024), dim3(256), 0, pl.stream, L.ptrs(), n);
      hipLaunchKernelGGL(k_lbd_settled, dim3(1), dim3(1), 0, pl.stream, L.state(L.par));
      SI_HIP(hipGetLastError());
    }
    SI_HIP(hipMemcpyAsync(&L.h, L.state(L.par), sizeof(LbdState), hipMemcpyDeviceToHost, pl.stream));
    SI_HIP(hipStreamSynchronize(pl.stream));
    if (!L.h.suspended) break;
    // the lean chain met a memory to multiply with at iteration resume_k (whose evaluation is done: gradient, rows and record
    // are as it left them - everything enqueued behind it ran as no-ops): the full form takes over from that decision
    ++L.suspensions;
    exec_ranges.emplace_back(exec_lo, std::min(n_eval_launched, exec_lo + (L.h.evals - exec_counted) + 1));   // (+ the suspended iteration's own)
    exec_counted = L.h.evals + 1;
    exec_lo = n_eval_launched;
    lean = false;
    if (L.h.d_implicit) {                           // (a two-launch chain hands over at the iteration that accepts its first pair:
      hipLaunchKernelGGL((k_lbd_materialise_d<float>), dim3(1024), dim3(256), 0, pl.stream, L.ptrs(), n);   // s = t d is read from d)
      L.h.d_implicit = 0;
      SI_HIP(hipGetLastError());
    }
    k = k_first = L.h.resume_k;
    L.h.suspended = 0;
    L.h.active = 1;
    L.h.do_eval = 1;
    SI_HIP(hipMemcpyAsync(L.state(L.par), &L.h, sizeof(LbdState), hipMemcpyHostToDevice, pl.stream));
    L.board_host[0] = 1.0;
  }
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
  info->lean_iterations = L.lean_launches;
  info->full_iterations = L.full_launches;
  info->suspensions = L.suspensions;
  info->reserved_ = 0;
  if (L.time_objective) {
    exec_ranges.emplace_back(exec_lo, std::min(n_eval_launched, exec_lo + std::max(0, L.h.evals - exec_counted)));
    for (const auto& r : exec_ranges)                 // (executed ones; gated launches are no-ops)
      for (int i = r.first; i < r.second; ++i) {
        if (i % L.time_objective != 0 || (size_t)(2 * i + 1) >= L.ev.size()) continue;
        float ms = 0.0f;
        SI_HIP(hipEventElapsedTime(&ms, L.ev[2 * i], L.ev[2 * i + 1]));
        info->objective_ms += ms;
        info->objective_timed += 1;
      }
  }
  L.defer_live = false;
  return SPECINV_OK;
}

}  // namespace specinv
