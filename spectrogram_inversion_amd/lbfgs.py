"""L-BFGS optimiser used by `L_BFGS` (reference: torch_specinv/methods.py:543,553 hand the
problem to the third-party `torch.optim.LBFGS`).

This is an independent implementation of that optimiser's published algorithm with its
option names and defaults (lr, max_iter, max_eval, tolerance_grad, tolerance_change,
history_size, line_search_fn): limited-memory BFGS two-loop recursion, first step length
min(1, 1/|g|_1)*lr, curvature guard y.s > 1e-10, initial Hessian scale y.s/y.y, optional
strong-Wolfe line search by cubic interpolation.  The host only runs the control flow;
all vector arithmetic (dot, axpy, scale, |.|max / |.|1) is done by libspecinv's HIP
reduction kernels through a `VecOps` backend, with dot products accumulated in float64.
"""
from __future__ import annotations

import math
import os

import numpy as np
import torch

from .plan import StftArgs, get_plan


class _HostBoard:
    """`specinv_board_alloc`: doubles in pinned host memory, mapped into the device's address space."""

    def __init__(self, plan, n):
        self.plan, self.n = plan, n                        # (the plan owns the memory: keep it alive)
        self.host, self.dev = plan.board_alloc(n)

    def data_ptr(self):
        return self.dev

    def numel(self):
        return self.n

    def put(self, slot, value):
        self.host[slot] = value

    def read(self, n):
        self.plan.stream_wait()
        return self.host[:n]


class HipVecOps:
    """Vector primitives on the HIP device (libspecinv vec_* entry points)."""

    def __init__(self, dtype, device):
        args = StftArgs(2, 2, 1, torch.ones(2, dtype=dtype), False, "reflect", False, True)
        self.plan = get_plan(args, 1, 1, dtype, device)

    def dot(self, a, b):
        return self.plan.vec_dot(a, b)

    def axpy(self, alpha, x, y):
        self.plan.vec_axpy(alpha, x, y)

    def scaled(self, alpha, x):
        out = torch.empty_like(x)
        self.plan.vec_scale(alpha, x, out)
        return out

    def absmax_abssum(self, x):
        return self.plan.vec_absmax_abssum(x)

    def direction(self, g, ss, ys, rho, h_diag):
        return self.plan.lbfgs_direction(g, ss, ys, rho, h_diag)

    def multi_dot(self, g, vecs):
        return self.plan.vec_multi_dot(g, vecs)

    def lincomb(self, vecs, coefs):
        return self.plan.vec_lincomb(vecs, coefs)

    def lincomb_step(self, vecs, coefs, t, x):
        if x.data_ptr() % 16 != 0 or not x.is_contiguous():      # (a view at an odd offset: two passes)
            d = self.plan.vec_lincomb(vecs, coefs)
            self.axpy(t, d, x)
            return d
        return self.plan.vec_lincomb_step(vecs, coefs, t, x)

    def pair(self, g, g_prev, d, t):
        return self.plan.lbfgs_pair(g, g_prev, d, t)

    def stats(self, g, d):
        return self.plan.lbfgs_stats(g, d)

    # -- packed form: results stay in a device "board" of doubles; `read` is the only synchronisation -----------------
    packed = True

    def board(self, n):
        """scalar results of an iteration: pinned host memory written by the kernels themselves (no copy on the way back).
        ONE board per plan, shared by every optimiser that runs on it (the plan - cached by `get_plan` - owns the pinned
        memory until it is destroyed, so a board per `LBFGS` object would pin another region on every `L_BFGS` call): an
        iteration writes and reads its slots within one `_batch` call, and a plan serves one thread at a time."""
        cached = getattr(self.plan, "_lbfgs_board", None)
        if cached is None or cached.numel() < n:
            cached = _HostBoard(self.plan, max(n, 9 + 2 * 100))      # (room for torch.optim.LBFGS's default history)
            self.plan._lbfgs_board = cached
        return cached

    def eval_into(self, fg, x, board, slot):
        """gradient of the objective at x; the loss goes to board[slot]"""
        if hasattr(fg, "dev"):
            return fg.dev(x, board.data_ptr() + 8 * slot)
        loss, g = fg(x)                                    # a generic callable: its loss is already on the host
        board.put(slot, float(loss))
        return g

    def stats_into(self, g, d, board, slot):
        self.plan.lbfgs_stats_dev(g, d, board.data_ptr() + 8 * slot)

    def pair_into(self, g, g_prev, d, t, board, slot):
        return self.plan.lbfgs_pair_dev(g, g_prev, d, t, board.data_ptr() + 8 * slot)

    def pair_stats_into(self, g, g_prev, d, t, board, slot):
        """`stats_into` (4 values) followed by `pair_into` (4 values) in one pass over the vectors"""
        return self.plan.lbfgs_pair_stats_dev(g, g_prev, d, t, board.data_ptr() + 8 * slot)

    def multi_dot_into(self, g, vecs, board, slot):
        self.plan.vec_multi_dot_dev(g, vecs, board.data_ptr() + 8 * slot)

    def read(self, board, n):
        return board.read(n)


def _cubic_step(xa, fa, ga, xb, fb, gb, bounds=None):
    """Minimiser of the cubic interpolating (xa, fa, ga) and (xb, fb, gb), clipped to bounds."""
    lo, hi = bounds if bounds is not None else ((xa, xb) if xa <= xb else (xb, xa))
    d1 = ga + gb - 3.0 * (fa - fb) / (xa - xb)
    disc = d1 * d1 - ga * gb
    if disc < 0:
        return 0.5 * (lo + hi)
    d2 = math.sqrt(disc)
    if xa <= xb:
        pos = xb - (xb - xa) * ((gb + d2 - d1) / (gb - ga + 2.0 * d2))
    else:
        pos = xa - (xa - xb) * ((ga + d2 - d1) / (ga - gb + 2.0 * d2))
    return min(max(pos, lo), hi)


class LBFGS:
    """Optimises the flat tensor `x` in place.  `step(fg)` performs one `optimizer.step`:
    `fg(x)` must return (loss: float, grad: tensor like x)."""

    def __init__(self, x, device=None, lr=1, max_iter=20, max_eval=None, tolerance_grad=1e-7,
                 tolerance_change=1e-9, history_size=100, line_search_fn=None, vec_ops=None):
        if not 0.0 <= lr:
            raise ValueError(f"Invalid learning rate: {lr}")
        if line_search_fn not in (None, "strong_wolfe"):
            raise RuntimeError("only 'strong_wolfe' is supported")
        self.x = x
        self.lr, self.max_iter = float(lr), int(max_iter)
        self.max_eval = int(max_eval) if max_eval is not None else self.max_iter * 5 // 4
        self.tol_grad, self.tol_change = tolerance_grad, tolerance_change
        self.history_size, self.line_search = int(history_size), line_search_fn
        self.ops = vec_ops if vec_ops is not None else HipVecOps(x.dtype, x.device if device is None else device)
        self.total_iters = 0
        self.func_evals = 0
        self.pairs_accepted = 0                # curvature pairs that passed `y.s > 1e-10` ...
        self.pairs_rejected = 0                # ... and those that did not (the memory stays as it was)
        self.d = None
        self.t = None
        # the recursion on Gram matrices (two passes over the memory) needs the one-pass vector kernels
        self.gram = hasattr(self.ops, "multi_dot") and hasattr(self.ops, "lincomb") and os.environ.get("SPECINV_LBFGS_GRAM", "1") != "0"
        self._forget()
        self.prev_grad = None
        self.prev_loss = None
        self._board = None
        self._dev = None                       # device-resident optimiser: None undecided, False not taken, else its handle
        self._dev_history = 0
        self.time_objective = 0                # benchmarks: HIP events around every k-th objective evaluation of the device path
        self.objective_ms, self.objective_timed, self.objective_launches = 0.0, 0, 0

    @property
    def history_len(self):
        """curvature pairs in the memory (<= history_size)"""
        return self._dev_history if self._dev else len(self.ss)

    # ---- pieces -----------------------------------------------------------------------------
    def _forget(self):
        self.ys, self.ss, self.rho, self.h_diag = [], [], [], 1.0
        self._sy = np.zeros((0, 0))            # s_i . y_j (used for i <= j)
        self._yy = np.zeros((0, 0))            # y_i . y_j
        self._sg = self._yg = None             # s_i . g, y_i . g of the previous direction
        self._pushed = None                    # (y.s, y.y) of a pair appended since the previous direction

    def _drop_oldest(self):
        self.ys.pop(0)
        self.ss.pop(0)
        self.rho.pop(0)
        if self.gram:
            self._sy, self._yy = self._sy[1:, 1:], self._yy[1:, 1:]
            if self._sg is not None:
                self._sg, self._yg = self._sg[1:], self._yg[1:]

    def _direction_gram(self, g):
        """-H g with two passes over the memory.  The two-loop recursion only needs the scalars s_i . q, y_i . r; with
        q = g - sum_j al_j y_j and r = gamma q + sum_j (al_j - be_j) s_j these follow from s_i . g, y_i . g (one pass:
        `multi_dot`) and the Gram matrices s_i . y_j, y_i . y_j.  A new pair's Gram column needs no pass of its own:
        y_new = g - g_prev, so v . y_new = v . g - v . g_prev, both already known.  The direction is one linear
        combination of g and the memory (`lincomb`).  Same algebra as `_direction`, different rounding."""
        ops, m = self.ops, len(self.ss)
        dots = np.asarray(ops.multi_dot(g, self.ss + self.ys), dtype=np.float64)
        sg, yg = dots[:m], dots[m:]
        self._gram_append(sg, yg)
        # d = -(gamma (g - sum al_j y_j) + sum c_i s_i)
        return ops.lincomb([g] + self.ys + self.ss, self._gram_coefficients(sg, yg))

    def _gram_append(self, sg, yg):
        """Bring the Gram matrices s_i.y_j, y_i.y_j up to date with the memory (a pair may have been appended since the
        previous direction) from the products `sg`, `yg` of the current gradient with the memory."""
        m = len(self.ss)
        if self._pushed is not None:
            ys_new, yy_new = self._pushed
            sy, yy = np.zeros((m, m)), np.zeros((m, m))
            sy[:m - 1, :m - 1], yy[:m - 1, :m - 1] = self._sy, self._yy
            if m > 1:
                sy[:m - 1, m - 1] = sg[:m - 1] - self._sg          # s_i . y_new
                yy[:m - 1, m - 1] = yy[m - 1, :m - 1] = yg[:m - 1] - self._yg
            sy[m - 1, m - 1], yy[m - 1, m - 1] = ys_new, yy_new
            self._sy, self._yy, self._pushed = sy, yy, None
        self._sg, self._yg = sg, yg

    def _gram_coefficients(self, sg, yg):
        """Coefficients of d = -H g as a linear combination of [g] + ys + ss (the two-loop recursion on scalars)."""
        m = len(self.ss)
        rho, gamma = np.asarray(self.rho, dtype=np.float64), float(self.h_diag)
        al = np.zeros(m)
        for i in range(m - 1, -1, -1):
            al[i] = rho[i] * (sg[i] - np.dot(al[i + 1:], self._sy[i, i + 1:]))
        yq = yg - self._yy @ al                                  # y_i . q
        c = np.zeros(m)                                          # al_i - be_i
        for i in range(m):
            c[i] = al[i] - rho[i] * (gamma * yq[i] + np.dot(c[:i], self._sy[:i, i]))
        return [-gamma] + list(gamma * al) + list(-c)

    def _direction(self, g):
        """Two-loop recursion: returns -H g for the current memory."""
        ops = self.ops
        if self.gram and self.ss:
            return self._direction_gram(g)
        if hasattr(ops, "direction"):          # device-resident scalars: no host sync inside the recursion
            return ops.direction(g, self.ss, self.ys, self.rho, self.h_diag)
        m = len(self.ys)
        q = ops.scaled(-1.0, g)
        al = [0.0] * m
        for i in range(m - 1, -1, -1):
            al[i] = ops.dot(self.ss[i], q) * self.rho[i]
            ops.axpy(-al[i], self.ys[i], q)
        r = ops.scaled(self.h_diag, q)
        for i in range(m):
            be = ops.dot(self.ys[i], r) * self.rho[i]
            ops.axpy(al[i] - be, self.ss[i], r)
        return r

    def _wolfe(self, fg, x0, t, d, f0, g0, gtd0, max_ls, c1=1e-4, c2=0.9):
        ops = self.ops

        def phi(step):
            trial = x0.clone()
            ops.axpy(step, d, trial)
            f, g = fg(trial)
            return f, g, ops.dot(g, d)

        d_norm = ops.absmax_abssum(d)[0]
        f_new, g_new, gtd_new = phi(t)
        evals, it = 1, 0
        t_prev, f_prev, g_prev, gtd_prev = 0.0, f0, g0, gtd0
        done, br = False, None
        while it < max_ls:
            if f_new > f0 + c1 * t * gtd0 or (it > 1 and f_new >= f_prev):
                br = [[t_prev, f_prev, g_prev, gtd_prev], [t, f_new, g_new, gtd_new]]
                break
            if abs(gtd_new) <= -c2 * gtd0:
                br, done = [[t, f_new, g_new, gtd_new]], True
                break
            if gtd_new >= 0:
                br = [[t_prev, f_prev, g_prev, gtd_prev], [t, f_new, g_new, gtd_new]]
                break
            nxt = _cubic_step(t_prev, f_prev, gtd_prev, t, f_new, gtd_new, (t + 0.01 * (t - t_prev), t * 10))
            t_prev, f_prev, g_prev, gtd_prev = t, f_new, g_new, gtd_new
            t = nxt
            f_new, g_new, gtd_new = phi(t)
            evals += 1
            it += 1
        if it == max_ls:
            br = [[0.0, f0, g0, gtd0], [t, f_new, g_new, gtd_new]]

        stalled = False
        lo, hi = (0, 1) if br[0][1] <= br[-1][1] else (1, 0)
        while not done and it < max_ls:
            # torch.optim.LBFGS.step does not forward its tolerance_change to the line search: the bracket test always
            # uses _strong_wolfe's own default of 1e-9
            if abs(br[1][0] - br[0][0]) * d_norm < 1e-9:
                break
            t = _cubic_step(br[0][0], br[0][1], br[0][3], br[1][0], br[1][1], br[1][3])
            bmax, bmin = max(br[0][0], br[1][0]), min(br[0][0], br[1][0])
            margin = 0.1 * (bmax - bmin)
            if min(bmax - t, t - bmin) < margin:
                if stalled or t >= bmax or t <= bmin:
                    t = bmax - margin if abs(t - bmax) < abs(t - bmin) else bmin + margin
                    stalled = False
                else:
                    stalled = True
            else:
                stalled = False
            f_new, g_new, gtd_new = phi(t)
            evals += 1
            it += 1
            if f_new > f0 + c1 * t * gtd0 or f_new >= br[lo][1]:
                br[hi] = [t, f_new, g_new, gtd_new]
                lo, hi = (0, 1) if br[0][1] <= br[1][1] else (1, 0)
            else:
                if abs(gtd_new) <= -c2 * gtd0:
                    done = True
                elif gtd_new * (br[hi][0] - br[lo][0]) >= 0:
                    br[hi] = list(br[lo])
                br[lo] = [t, f_new, g_new, gtd_new]
        if len(br) == 1:
            lo = 0
        return br[lo][1], br[lo][2], br[lo][0], evals

    # ---- one optimizer.step, one host synchronisation per inner iteration ------------------------
    def _batch(self, fg, x, d, t):
        """Everything the next decisions need, enqueued back to back and fetched with ONE read: the objective at x (loss,
        gradient g), the step statistics {g.d, sum|g|, max|g|, max|d|} and - once there is a previous gradient - the
        curvature pair y = g - g_prev, s = t d with {y.s, y.y, g.g, g.g_prev} and the products of g with the memory."""
        ops = self.ops
        m = len(self.ss)
        have_prev = self.total_iters >= 1
        if self._board is None or self._board.numel() < 9 + 2 * self.history_size:
            self._board = ops.board(9 + 2 * self.history_size)
        bd = self._board
        g = ops.eval_into(fg, x, bd, 0)
        y = s = None
        if not have_prev:
            ops.stats_into(g, g, bd, 1)
        else:
            if hasattr(ops, "pair_stats_into"):
                y, s = ops.pair_stats_into(g, self.prev_grad, d, t, bd, 1)
            else:
                ops.stats_into(g, d, bd, 1)
                y, s = ops.pair_into(g, self.prev_grad, d, t, bd, 5)
            if m:
                ops.multi_dot_into(g, self.ss + self.ys, bd, 9)
        v = ops.read(bd, 9 + 2 * m if have_prev else 5)
        out = dict(g=g, loss=v[0], gd=v[1], g_abssum=v[2], g_absmax=v[3], d_absmax=v[4], have_prev=have_prev, m=m)
        if have_prev:
            out.update(y=y, s=s, ys=v[5], yy=v[6], gg=v[7], ggp=v[8], sg=np.asarray(v[9:9 + m], dtype=np.float64),
                       yg=np.asarray(v[9 + m:9 + 2 * m], dtype=np.float64))
        return out

    def _step_packed(self, fg):
        """`step` for a backend with device-resident results (no line search): the same decisions in the same order as
        torch.optim.LBFGS.step, taken from one packed read-back per inner iteration.  The products of the new pair with
        the gradient follow by linearity (s.g = t d.g, y.g = g.g - g_prev.g), and so does g.d of the new direction, a
        linear combination of vectors whose products with g are all known."""
        ops, x = self.ops, self.x
        b = self._batch(fg, x, self.d, self.t)
        loss = first_loss = b["loss"]
        evals = 1
        self.func_evals += 1
        if b["g_absmax"] <= self.tol_grad:
            return first_loss
        d, t = self.d, self.t
        n_iter = 0
        while n_iter < self.max_iter:
            n_iter += 1
            self.total_iters += 1
            g = b["g"]
            if self.total_iters == 1:
                self._forget()
                gtd = -b["gd"]                                 # statistics were taken with d = g: g.g
                vecs, coefs = None, None
            else:
                sg, yg = b["sg"], b["yg"]
                self.pairs_accepted += b["ys"] > 1e-10
                self.pairs_rejected += not b["ys"] > 1e-10
                if b["ys"] > 1e-10:
                    if len(self.ys) == self.history_size:
                        self._drop_oldest()
                        if b["m"] == self.history_size:
                            sg, yg = sg[1:], yg[1:]
                    self.ys.append(b["y"])
                    self.ss.append(b["s"])
                    self.rho.append(1.0 / b["ys"])
                    self.h_diag = b["ys"] / b["yy"]
                    self._pushed = (b["ys"], b["yy"])
                    sg = np.append(sg, t * b["gd"])            # s_new . g = t (d . g)
                    yg = np.append(yg, b["gg"] - b["ggp"])     # y_new . g = g . g - g_prev . g
                self._gram_append(sg, yg)
                coefs = self._gram_coefficients(sg, yg)
                vecs = [g] + self.ys + self.ss
                m = len(self.ss)
                gtd = coefs[0] * b["gg"] + float(np.dot(coefs[1:1 + m], yg)) + float(np.dot(coefs[1 + m:], sg))
            self.prev_grad = g
            self.prev_loss = loss
            t = min(1.0, 1.0 / b["g_abssum"]) * self.lr if self.total_iters == 1 else self.lr
            if gtd > -self.tol_change:
                d = ops.scaled(-1.0, g) if vecs is None else ops.lincomb(vecs, coefs)
                break
            if hasattr(ops, "lincomb_step"):                   # direction and step in one pass over the vectors
                d = ops.lincomb_step([g] if vecs is None else vecs, [-1.0] if vecs is None else coefs, t, x)
            else:
                d = ops.scaled(-1.0, g) if vecs is None else ops.lincomb(vecs, coefs)
                ops.axpy(t, d, x)
            ls_evals = 0
            opt = False
            if n_iter != self.max_iter:
                b = self._batch(fg, x, d, t)
                loss = b["loss"]
                opt = b["g_absmax"] <= self.tol_grad
                ls_evals = 1
            evals += ls_evals
            self.func_evals += ls_evals
            if n_iter == self.max_iter or evals >= self.max_eval or opt:
                break
            if abs(t) * b["d_absmax"] <= self.tol_change:
                break
            if abs(loss - self.prev_loss) < self.tol_change:
                break
        self.d, self.t = d, t
        return first_loss

    # ---- one optimizer.step with the strong-Wolfe line search, one host synchronisation per EVALUATION -----------------------
    def _eval_point(self, fg, x, d):
        """The objective at x and the statistics a decision needs, enqueued back to back and fetched with ONE read:
        (g, loss, g.d, sum|g|, max|g|, max|d|).  `d` None: the statistics are taken with d = g."""
        ops = self.ops
        if self._board is None or self._board.numel() < 9 + 2 * self.history_size:
            self._board = ops.board(9 + 2 * self.history_size)
        bd = self._board
        if hasattr(fg, "dev_stats") and os.environ.get("SPECINV_LBFGS_FUSED_STATS", "1") != "0":
            g = fg.dev_stats(x, d, bd.data_ptr())          # the statistics ride on the objective's own launches
        else:
            g = ops.eval_into(fg, x, bd, 0)
            ops.stats_into(g, g if d is None else d, bd, 1)
        v = ops.read(bd, 5)
        return g, v[0], v[1], v[2], v[3], v[4]

    def _pair_products(self, g, g_prev, d, t):
        """The curvature pair y = g - g_prev, s = t d with {y.s, y.y, g.g, g.g_prev} and the products of g with the memory:
        one pass each, ONE read - no evaluation (the line search has produced g already)."""
        ops, bd, m = self.ops, self._board, len(self.ss)
        y, s = ops.pair_into(g, g_prev, d, t, bd, 5)
        if m:
            ops.multi_dot_into(g, self.ss + self.ys, bd, 9)
        v = ops.read(bd, 9 + 2 * m)
        return y, s, v[5], v[6], v[7], v[8], np.asarray(v[9:9 + m], dtype=np.float64), np.asarray(v[9 + m:9 + 2 * m], dtype=np.float64)

    def _wolfe_packed(self, fg, x0, t, d, f0, g0, gtd0, max_ls, c1=1e-4, c2=0.9):
        """`_wolfe` (torch.optim.lbfgs._strong_wolfe: bracket, cubic interpolation, zoom) with every trial point evaluated by
        `_eval_point`: the step, the objective, g.d and max|g| enqueued together, one read-back per trial instead of three.
        Same decisions in the same order on the same values.  Returns (loss, g, t, evaluations, g.d, max|g|) of the accepted point and max|d|."""
        ops = self.ops

        trial = torch.empty_like(x0)                       # (the trial point is dropped after its evaluation: one buffer)

        def phi(step):
            torch.add(x0, d, alpha=step, out=trial)        # x0 + step d in one launch (a fused multiply-add like `axpy`'s)
            g, f, gd, _, gmax, dmax = self._eval_point(fg, trial, d)
            return [step, f, g, gd, gmax], dmax

        new, d_norm = phi(t)
        evals, it = 1, 0
        prev = [0.0, f0, g0, gtd0, None]
        done, br = False, None
        while it < max_ls:
            if new[1] > f0 + c1 * new[0] * gtd0 or (it > 1 and new[1] >= prev[1]):
                br = [prev, new]
                break
            if abs(new[3]) <= -c2 * gtd0:
                br, done = [new], True
                break
            if new[3] >= 0:
                br = [prev, new]
                break
            nxt = _cubic_step(prev[0], prev[1], prev[3], new[0], new[1], new[3], (new[0] + 0.01 * (new[0] - prev[0]), new[0] * 10))
            prev = new
            new, _ = phi(nxt)
            evals += 1
            it += 1
        if it == max_ls:
            br = [[0.0, f0, g0, gtd0, None], new]
        stalled = False
        lo, hi = (0, 1) if br[0][1] <= br[-1][1] else (1, 0)
        while not done and it < max_ls:
            if abs(br[1][0] - br[0][0]) * d_norm < 1e-9:         # (_strong_wolfe's own tolerance_change: torch does not forward its)
                break
            t = _cubic_step(br[0][0], br[0][1], br[0][3], br[1][0], br[1][1], br[1][3])
            bmax, bmin = max(br[0][0], br[1][0]), min(br[0][0], br[1][0])
            margin = 0.1 * (bmax - bmin)
            if min(bmax - t, t - bmin) < margin:
                if stalled or t >= bmax or t <= bmin:
                    t = bmax - margin if abs(t - bmax) < abs(t - bmin) else bmin + margin
                    stalled = False
                else:
                    stalled = True
            else:
                stalled = False
            new, _ = phi(t)
            evals += 1
            it += 1
            if new[1] > f0 + c1 * t * gtd0 or new[1] >= br[lo][1]:
                br[hi] = new
                lo, hi = (0, 1) if br[0][1] <= br[1][1] else (1, 0)
            else:
                if abs(new[3]) <= -c2 * gtd0:
                    done = True
                elif new[3] * (br[hi][0] - br[lo][0]) >= 0:
                    br[hi] = list(br[lo])
                br[lo] = new
        if len(br) == 1:
            lo = 0
        best = br[lo]
        return best[1], best[2], best[0], evals, best[3], best[4], d_norm

    def _step_wolfe_packed(self, fg):
        """`step` with line_search_fn='strong_wolfe' for an objective that can leave its loss on the device (`fg.dev`): the
        decisions of torch.optim.LBFGS.step in the same order, taken from ONE packed read-back per evaluation (entry point, every
        trial of the line search) and one per curvature pair, where the general path below reads three scalars one by one around
        every evaluation.  The direction is a linear combination of g and the memory with coefficients from the Gram recursion
        (`_gram_coefficients`); g.d of the new direction follows from the known products, as in `_step_packed`."""
        ops, x = self.ops, self.x
        # the entry evaluation; its statistics are taken against the direction of the previous step's last iteration (g . d is
        # needed for the curvature pair: s_new . g = t (d . g)) or, before the first iteration, against g itself (g . g)
        g, loss, gd_old, g_abssum, g_absmax, d_norm = self._eval_point(fg, x, self.d if self.total_iters >= 1 else None)
        first_loss = loss
        evals = 1
        self.func_evals += 1
        if g_absmax <= self.tol_grad:
            return first_loss
        d, t = self.d, self.t
        n_iter = 0
        while n_iter < self.max_iter:
            n_iter += 1
            self.total_iters += 1
            if self.total_iters == 1:
                self._forget()
                d = ops.scaled(-1.0, g)
                gtd = -gd_old                                  # (the entry statistics were taken with d = g)
            else:
                y, s, ys, yy, gg, ggp, sg, yg = self._pair_products(g, self.prev_grad, d, t)
                self.pairs_accepted += ys > 1e-10
                self.pairs_rejected += not ys > 1e-10
                if ys > 1e-10:
                    if len(self.ys) == self.history_size:
                        self._drop_oldest()
                        sg, yg = sg[1:], yg[1:]
                    self.ys.append(y)
                    self.ss.append(s)
                    self.rho.append(1.0 / ys)
                    self.h_diag = ys / yy
                    self._pushed = (ys, yy)
                    sg = np.append(sg, t * gd_old)             # s_new . g = t (d . g)
                    yg = np.append(yg, gg - ggp)               # y_new . g = g . g - g_prev . g
                self._gram_append(sg, yg)
                coefs = self._gram_coefficients(sg, yg)
                m = len(self.ss)
                d = ops.lincomb([g] + self.ys + self.ss, coefs)
                gtd = coefs[0] * gg + float(np.dot(coefs[1:1 + m], yg)) + float(np.dot(coefs[1 + m:], sg))
            self.prev_grad = g
            self.prev_loss = loss
            t = min(1.0, 1.0 / g_abssum) * self.lr if self.total_iters == 1 else self.lr
            if gtd > -self.tol_change:
                break
            loss, g, t, ls_evals, gd_old, ls_gmax, d_norm = self._wolfe_packed(fg, x.clone(), t, d, loss, g, gtd, self.max_eval - evals)
            ops.axpy(t, d, x)
            if ls_gmax is not None:                        # (None: the search ended on its starting point - t = 0, g is the gradient
                g_absmax = ls_gmax                         # it started from, whose max|g| is known)
            opt = g_absmax <= self.tol_grad
            evals += ls_evals
            self.func_evals += ls_evals
            if n_iter == self.max_iter or evals >= self.max_eval or opt:
                break
            if abs(t) * d_norm <= self.tol_change:
                break
            if abs(loss - self.prev_loss) < self.tol_change:
                break
        self.d, self.t = d, t
        return first_loss

    # ---- one optimizer.step with the decisions on the device: one host synchronisation per STEP -------------------
    def _device_ok(self, fg):
        """The device-resident optimiser (csrc/lbfgs_dev.h) serves float32 parameters on the one-launch objective, without a line
        search, history_size <= 120.  Decided at the first step; an optimiser never changes paths afterwards (its state
        lives where its path keeps it)."""
        if self._dev is not None:
            return self._dev is not False
        self._dev = False
        obj = getattr(fg, "device_objective", None)
        if obj is None or not self.gram or os.environ.get("SPECINV_LBFGS_DEVICE", "1") == "0":
            return False
        if self.line_search is not None:
            # strong Wolfe is driven from the host (_step_wolfe_packed: one read-back per evaluation).  Round 4 also built its state
            # machine as a decision kernel; measured slower (47 M against 77 M evaluations*frames/s at C5: six gated launches per
            # slot cost what the read-backs did) and removed in round 5
            return False
        plan, target, shape = obj
        x = self.x
        if self.total_iters != 0 or x.dtype != torch.float32 or not x.is_contiguous() or x.data_ptr() % 16 != 0 or \
                x.numel() != shape[0] * shape[1] or not 1 <= self.history_size <= 120:
            return False
        from . import _lib
        try:
            handle = plan.lbfgs_dev_create(x.numel(), self.lr, self.max_iter, self.max_eval, self.tol_grad, self.tol_change,
                                           self.history_size, self.time_objective)
        except (_lib.SpecinvError, NotImplementedError):
            return False
        self._dev = (plan, handle, target, shape)
        # the device keeps its own copy of the options from here on: a later change of these attributes would be ignored
        self._dev_opts = (self.lr, self.max_iter, self.max_eval, self.tol_grad, self.tol_change, self.history_size)
        return True

    def _step_device(self, fg):
        from . import _lib
        plan, handle, target, shape = self._dev
        obj = getattr(fg, "device_objective", None)     # the closure of THIS step: same objective kernel, possibly another target
        if obj is None or obj[0] is not plan or tuple(obj[2]) != tuple(shape):
            raise RuntimeError("this optimiser's state lives on the device with the objective it was first stepped with; "
                               "use a new LBFGS for another transform / shape (SPECINV_LBFGS_DEVICE=0 keeps the state on the host)")
        target = obj[1]
        if (self.lr, self.max_iter, self.max_eval, self.tol_grad, self.tol_change, self.history_size) != self._dev_opts:
            raise RuntimeError("this optimiser's options were copied to the device at its first step and cannot change afterwards "
                               "(lr, max_iter, max_eval, tolerances, history_size); create a new LBFGS, or run with "
                               "SPECINV_LBFGS_DEVICE=0 to keep the state on the host")
        if getattr(self, "_dev_poisoned", False):
            raise RuntimeError("an earlier step of this optimiser failed after part of it had been enqueued: its device state is "
                               "undefined; create a new LBFGS")
        try:
            info = plan.lbfgs_dev_step(handle, self.x.view(shape), target)
        except NotImplementedError:
            if self.total_iters != 0:
                self._dev_poisoned = True
                raise
            plan.lbfgs_dev_destroy(handle)              # a configuration the one-launch objective does not cover: host-driven loop
            self._dev = False
            return self.step(fg)
        except Exception:
            self._dev_poisoned = True                   # (kernels of the step may have run: x and the state record may disagree)
            raise
        self.total_iters, self.func_evals = info.total_iters, info.func_evals
        self.pairs_accepted, self.pairs_rejected = info.pairs_accepted, info.pairs_rejected
        self._dev_history, self.t, self.prev_loss = info.history_len, info.t, info.loss
        self.objective_ms += info.objective_ms
        self.objective_timed += info.objective_timed
        self.objective_launches += info.objective_launches
        self.dev_iterations = (info.lean_iterations, info.full_iterations, info.suspensions)
        return info.first_loss

    def __del__(self):
        dev = getattr(self, "_dev", None)
        if dev:
            try:
                dev[0].lbfgs_dev_destroy(dev[1])
            except Exception:                           # (interpreter shutdown: the plan may be gone already)
                pass

    # ---- one optimizer.step ------------------------------------------------------------------
    def step(self, fg):
        if self._device_ok(fg):
            return self._step_device(fg)
        if self.line_search is None and self.gram and getattr(self.ops, "packed", False) and \
                os.environ.get("SPECINV_LBFGS_PACKED", "1") != "0":
            return self._step_packed(fg)
        if self.line_search is not None and self.gram and getattr(self.ops, "packed", False) and hasattr(fg, "dev") and \
                os.environ.get("SPECINV_LBFGS_PACKED", "1") != "0":
            return self._step_wolfe_packed(fg)
        ops, x = self.ops, self.x
        fused = hasattr(ops, "pair") and hasattr(ops, "stats")     # one pass per group of vector operations
        loss, g = fg(x)
        first_loss = loss
        evals = 1
        self.func_evals += 1
        g_absmax = ops.absmax_abssum(g)[0]
        if g_absmax <= self.tol_grad:
            return first_loss
        d, t = self.d, self.t
        n_iter = 0
        while n_iter < self.max_iter:
            n_iter += 1
            self.total_iters += 1
            if self.total_iters == 1:
                d = ops.scaled(-1.0, g)
                self._forget()
            else:
                if fused:
                    y, s, ys, yy = ops.pair(g, self.prev_grad, d, t)
                else:
                    y = g.clone()
                    ops.axpy(-1.0, self.prev_grad, y)
                    s = ops.scaled(t, d)
                    ys, yy = ops.dot(y, s), None
                self.pairs_accepted += ys > 1e-10
                self.pairs_rejected += not ys > 1e-10
                if ys > 1e-10:
                    if len(self.ys) == self.history_size:
                        self._drop_oldest()
                    self.ys.append(y)
                    self.ss.append(s)
                    self.rho.append(1.0 / ys)
                    yy = yy if yy is not None else ops.dot(y, y)
                    self.h_diag = ys / yy
                    self._pushed = (ys, yy)
                d = self._direction(g)
            self.prev_grad = g if fused else g.clone()     # fg returns a fresh tensor: nothing writes into it later
            self.prev_loss = loss
            if fused:
                gtd, g_abssum, g_absmax, d_absmax = ops.stats(g, d)
            else:
                gtd, d_absmax = ops.dot(g, d), None
                g_abssum = ops.absmax_abssum(g)[1] if self.total_iters == 1 else None
            if self.total_iters == 1:
                t = min(1.0, 1.0 / g_abssum) * self.lr
            else:
                t = self.lr
            if gtd > -self.tol_change:
                break
            ls_evals = 0
            if self.line_search is not None:
                loss, g, t, ls_evals = self._wolfe(fg, x.clone(), t, d, loss, g, gtd, self.max_eval - evals)
                ops.axpy(t, d, x)
                opt = ops.absmax_abssum(g)[0] <= self.tol_grad
            else:
                ops.axpy(t, d, x)
                opt = False
                if n_iter != self.max_iter:
                    loss, g = fg(x)
                    opt = ops.absmax_abssum(g)[0] <= self.tol_grad
                    ls_evals = 1
            evals += ls_evals
            self.func_evals += ls_evals
            if n_iter == self.max_iter or evals >= self.max_eval or opt:
                break
            if d_absmax is None:
                d_absmax = ops.absmax_abssum(d)[0]
            if abs(t) * d_absmax <= self.tol_change:
                break
            if abs(loss - self.prev_loss) < self.tol_change:
                break
        self.d, self.t = d, t
        return first_loss
