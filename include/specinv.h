/*
 * specinv.h - C ABI of libspecinv.so, the MI355X (gfx950) spectrogram-inversion engine.
 *
 * The reference (yoyololicon/spectrogram-inversion, package torch_specinv) has no FFI of
 * its own: its hot path is a set of plain Python functions exported at
 * torch_specinv/__init__.py:6.  This header is the boundary a binding for that path would
 * use; every entry point names the reference code it replaces (paths relative to the
 * reference checkout).  The Python host layer in spectrogram_inversion_amd/ binds it with
 * ctypes and mirrors the reference's function signatures on top.
 *
 * Conventions
 *   - plain C: opaque plan handle, raw pointers, ints/doubles; no exceptions cross the ABI.
 *   - every function returns 0 on success or a negative SPECINV_E* code; the message for the
 *     calling thread's last failure is available from specinv_last_error().
 *   - all data pointers are DEVICE pointers (HIP, the plan's device) unless the parameter
 *     name ends in _host.  The caller owns them; the plan owns its internal state buffers.
 *   - spectrogram arguments use the reference's layout: row-major (batch, n_freq, n_frames),
 *     real (float/double) or complex interleaved (re, im).  Waveforms are (batch, length).
 *   - work is enqueued on the plan's stream (specinv_plan_set_stream; default: the null
 *     stream).  Functions that return scalars to the host synchronise that stream.
 *   - a plan is not thread-safe; different plans may be used from different threads.
 */
#ifndef SPECINV_H
#define SPECINV_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPECINV_ABI_VERSION 1

enum {
  SPECINV_OK = 0,
  SPECINV_EINVAL = -1,       /* bad argument (the host layer raises AssertionError / ValueError) */
  SPECINV_EHIP = -2,         /* a HIP runtime call failed */
  SPECINV_EUNSUPPORTED = -3, /* configuration not implemented on the device path */
  SPECINV_ENOMEM = -4,
  SPECINV_ESTATE = -5        /* call sequence error (e.g. iterate before init) */
};

enum { SPECINV_F32 = 0, SPECINV_F64 = 1 };
enum { SPECINV_PAD_REFLECT = 0, SPECINV_PAD_CONSTANT = 1, SPECINV_PAD_REPLICATE = 2, SPECINV_PAD_CIRCULAR = 3 };
enum { SPECINV_METRIC_SC = 0, SPECINV_METRIC_SNR = 1, SPECINV_METRIC_SER = 2 };

/* Normalised STFT arguments: what torch_specinv/methods.py:21-91 (_args_helper) returns,
 * plus the problem shape.  `window_host` has n_fft elements of `dtype` and is already
 * centre-padded (methods.py:80-83). */
typedef struct specinv_stft_cfg {
  int32_t n_fft;
  int32_t hop_length;
  int32_t n_frames;      /* T */
  int32_t batch;         /* B, at most 65535 (the batch is a grid dimension of the layout kernels) */
  int32_t center;        /* bool */
  int32_t pad_mode;      /* SPECINV_PAD_* ; only read by the forward STFT when center != 0 */
  int32_t normalized;    /* bool: 'ortho' scaling both ways (methods.py:142-146) */
  int32_t onesided;      /* bool */
  int32_t dtype;         /* SPECINV_F32 / SPECINV_F64 */
  int32_t device;        /* HIP device ordinal */
  const void* window_host;
} specinv_stft_cfg;

typedef struct specinv_plan specinv_plan;

/* One evaluation of _training_loop (methods.py:180-184): metric value and F.mse_loss. */
typedef struct specinv_eval {
  int32_t iteration;     /* 0-based index of the iteration that was evaluated */
  double metric;
  double loss;
} specinv_eval;

/* Progress callback, called on the host after each evaluation; return non-zero to abort. */
typedef int (*specinv_eval_cb)(const specinv_eval* ev, void* user);

/* Whole-tensor sums behind metrics.py:14,28-29,43 and F.mse_loss:
 * sums[0] = sum((a-b)^2), sums[1] = sum(a^2), sums[2] = sum(b^2), sums[3] = element count. */

const char* specinv_last_error(void);
int specinv_abi_version(void);
/* 1 if the library carries the approximate-projection copies of the float32 wave-level kernels (built with SPECINV_BUILD_APPROX=1;
 * not in a default build since round 6: five translation units, ~2 CPU-minutes, for a 3 % opt-in), else 0 - specinv_plan_set_exact(plan, 0)
 * is then accepted and the plan keeps the reference's operation order. */
int specinv_has_approx(void);

/* ---- plan ---------------------------------------------------------------------------- */
int specinv_plan_create(const specinv_stft_cfg* cfg, specinv_plan** out);
int specinv_plan_destroy(specinv_plan* plan);
int specinv_plan_set_stream(specinv_plan* plan, void* hip_stream);
/* n_freq (F), signal length L = (T-1)*hop + n_fft - 2*pad (methods.py:127), and which iteration kernels the plan
 * uses: 1 = fused (one launch per iteration: wave-level FFT, register overlap-add; float32, one-sided, centred,
 * n_fft 512 / 1024 / 2048 / 4096, hop = n_fft/2, /4 or /8, >= n_fft/hop + 2 frames), 2 = the same wave-level frame
 * kernel + gather overlap-add (those n_fft, any hop / centring; two-sided float32 spectrograms as well), 3 = that frame
 * kernel walking chunks of frames with the overlap-add in LDS (n_fft <= 2048, hop <= n_fft, >= 12288 frames in the batch at
 * n_fft 2048, >= 32768 below; one- or two-sided), 0 = generic LDS-FFT kernels (everything else: float64, other sizes, a
 * two-sided run with specinv_plan_keep_state). */
int specinv_plan_n_freq(const specinv_plan* plan);
int64_t specinv_plan_length(const specinv_plan* plan);
int specinv_plan_fast_path(const specinv_plan* plan);
/* Device memory currently held by the plan's own buffers, in bytes (state, scratch, tables; grows as entry points
 * reserve what they need).  The host layer caps its plan cache with it. */
int64_t specinv_plan_device_bytes(const specinv_plan* plan);
/* Launch geometry of the iteration kernel (diagnostics; the tests assert that the benchmark's geometry is the one they
 * cover): out = { waves per workgroup, chunks of frames per item, waves per launch, kernel }, kernel: 0 generic
 * k_iter_pair, 1 k_fused4 (hop = n_fft/4 at n_fft 1024 / 2048), 2 k_fused<R, OV>, 3 k_semi, 4 k_hop, 5 k_fused4_td and
 * 6 k_fused_td<R, OV>, 7 k_hop_td (Griffin-Lim with the momentum carried as a signal; known once specinv_gla_init has run),
 * 8 k_wave_iter (the generic path's wave-level kernel: n_fft 128 ... 8192 (float32: 16384) and 400 / 800 / 1000; frames buffer + k_ola where out[1] is
 * the frame count, else the overlap-add in its registers), 9 k_wave_iter with the overlap-add in an LDS ring. */
int specinv_plan_launch_geometry(const specinv_plan* plan, int32_t out[4]);
/* 0: allow the fast path when the configuration supports it (default); 1: force the generic kernels. */
int specinv_plan_force_generic(specinv_plan* plan, int on);
/* The arithmetic of the magnitude projection S * m / (|S| + 1e-16) and of the division by the overlap-add envelope
 * (torch_specinv/methods.py:132,246-247) on the float32 wave-level kernels (fused, frame, chunked frame).
 * on = 1 (the default since round 4): the reference's operation order - ATen executes the projection as (S * m) * r with r the rounded
 * reciprocal of |S| + 1e-16 (tools/ref_ops_probe.py) - with r the correctly rounded 1 / |S| (one Newton step on v_rsq_f32) and a
 * correctly rounded division by the envelope: 73 % of the projected bins bit-identical to the reference's chain, every one within
 * its rounding noise; + 3 % on the headline step.  on = 0: S * (m * v_rsq_f32(|S|^2 + 1e-32)) and a multiplication by 1 / envelope
 * (52 % bit-identical, the same distance from the exact value).  The generic kernels and float64 use IEEE operations in the
 * reference's order throughout.  Takes effect at the next specinv_gla_init / specinv_admm_init.  on = 0 needs a library built with
 * SPECINV_BUILD_APPROX=1 (specinv_has_approx()); a default build keeps on = 1 whatever is asked. */
int specinv_plan_set_exact(specinv_plan* plan, int on);
/* The float32 fast paths do not carry the reference's spectral state as such.
 * ADMM keeps only Y = X + U between iterations: methods.py:467-468 read the two as U + X, i.e. the Y that :475 has just
 * rounded, so the iterates are bit-identical and the state traffic halves.  Griffin-Lim on the fused shapes (hop = n_fft/2, /4, /8) and on the chunked
 * frame kernel keeps its momentum as the signal z_t = x_t - lr z_{t-1} (pre_t = STFT(z_t) + (-lr)^t c0 by linearity of the STFT):
 * pre_spec is never formed.
 * 1: ADMM - the last iteration of every specinv_admm_iterate call (and specinv_admm_init) also leaves X and U behind for
 * specinv_get_state_spec; Griffin-Lim - the iteration runs on pre_spec itself (the spectral-state kernel).
 * 0 (default): asking for X / U / pre_spec after an iteration is SPECINV_ESTATE.  Call before specinv_*_init: where the flag
 * selects the kernels (a two-sided float32 run keeps X and U on the generic kernels only) it is latched by specinv_gla_init /
 * specinv_admm_init - a later call takes effect at the next init, the running method keeps its kernels and buffers.
 * (The generic kernels keep X and U anyway; the frame-at-a-time and generic Griffin-Lim kernels keep pre_spec.) */
int specinv_plan_keep_state(specinv_plan* plan, int on);

/* ---- building blocks ------------------------------------------------------------------ */
/* torch.stft(x, n_fft, **processed_args) as called at methods.py:241.  x (B, L_in) -> spec
 * (B, F, T) complex.  L_in must give exactly the plan's T frames. */
int specinv_stft(specinv_plan* plan, const void* x, int64_t length, void* spec_out);
/* _istft + _ola, methods.py:114-150: spec (B, F, T) complex -> x (B, L) = OLA / envelope. */
int specinv_istft(specinv_plan* plan, const void* spec, void* x_out);
/* window-square envelope of methods.py:129-131, (L,) */
int specinv_envelope(specinv_plan* plan, void* env_out);
/* phase_init, methods.py:572-615: mag (B, F, T) real -> (B, F, T) complex */
int specinv_phase_init(specinv_plan* plan, const void* mag, void* spec_out);
/* metrics.py sums over n elements of two real device arrays of the plan's dtype */
int specinv_metric_sums(specinv_plan* plan, const void* a, const void* b, int64_t n, double sums_host[4]);

/* ---- griffin_lim (methods.py:193-270) -------------------------------------------------- */
/* Setup :223-235.  `init_spec` (B,F,T) complex is the starting spectrogram (cmplx_spec);
 * `mag` (B,F,T) real is target_spec.  If init_spec is NULL the plan runs phase_init(mag). */
int specinv_gla_init(specinv_plan* plan, const void* init_spec, const void* mag, double alpha);
/* n_iter closure calls (:237-250).  If eval_last != 0 the |STFT| of the last call is reduced
 * against the target into sums_host[4] (this synchronises the stream). */
int specinv_gla_iterate(specinv_plan* plan, int n_iter, int eval_last, double sums_host[4]);
/* The whole _training_loop (:153-190) on the device side: evaluation every eva_iter, early
 * stop rule :186-190.  evals_out (may be NULL) receives up to max_iter/eva_iter entries. */
int specinv_gla_run(specinv_plan* plan, int max_iter, int eva_iter, double tol, int metric,
                    specinv_eval* evals_out, int* n_evals_out, int* iters_done_out,
                    specinv_eval_cb cb, void* user);

/* ---- ADMM (methods.py:415-506) --------------------------------------------------------- */
int specinv_admm_init(specinv_plan* plan, const void* init_spec, const void* mag, double rho);
int specinv_admm_iterate(specinv_plan* plan, int n_iter, int eval_last, double sums_host[4]);
/* n_iter iterations of the running method (Griffin-Lim or ADMM), the last one evaluating (methods.py:180-182) - its four sums
 * {sum (|S| - m)^2, sum |S|^2, sum m^2, count} left in DEVICE memory (4 doubles) instead of on the host: nothing waits.  For callers
 * that reduce the sums over several plans / ranks before they look at them (`_training_loop`'s whole-batch metric across GPUs,
 * methods.py:181-190: one all-reduce on the device, one read). */
int specinv_iterate_eval_dev(specinv_plan* plan, int n_iter, void* sums_dev);
int specinv_admm_run(specinv_plan* plan, int max_iter, int eva_iter, double tol, int metric,
                     specinv_eval* evals_out, int* n_evals_out, int* iters_done_out,
                     specinv_eval_cb cb, void* user);

/* current waveform estimate status_dict['x'] (B, L) of the running GLA / ADMM state */
int specinv_get_wave(specinv_plan* plan, void* x_out);
/* running state as (B, F, T) complex: which = 0 pre_spec (GLA) or X (ADMM), 1 U (ADMM), 2 Y = X + U (ADMM, always
 * available; X and U need specinv_plan_keep_state) - for state-parity tests */
int specinv_get_state_spec(specinv_plan* plan, int which, void* spec_out);

/* ---- differentiating griffin_lim w.r.t. the spectrogram (the reference's outputs are autograd-differentiable:
 * test/test_griffin.py:54,65-66).  Element-wise pieces work on (B, F, T) arrays; "g*" are cotangents. ---------- */
/* one closure call without the transforms, methods.py:243-247: S = R - lr*P ; Q = S*mag/(|S| + 1e-16) */
int specinv_gla_update(specinv_plan* plan, const void* R, const void* P, const void* mag, double lr, void* S_out,
                       void* Q_out);
/* its adjoint: gR = gS, gP = -lr*gS, gmag += d/dmag, with gS = proj^T(gQ) + gP_next (gP_next may be NULL) */
int specinv_gla_update_adjoint(specinv_plan* plan, const void* gQ, const void* gP_next, const void* S, const void* mag,
                               double lr, void* gR_out, void* gP_out, void* gmag_accum);
/* one ADMM closure call without the transforms, methods.py:467-475 (V is the pre-projection value Z - U', kept for
 * the adjoint; Yn = X' + U' feeds the ISTFT) and its adjoint (gXn / gUn may be NULL) */
int specinv_admm_update(specinv_plan* plan, const void* R, const void* X, const void* U, const void* mag, double rho,
                        void* Xn_out, void* Un_out, void* V_out, void* Yn_out);
int specinv_admm_update_adjoint(specinv_plan* plan, const void* gYn, const void* gXn, const void* gUn, const void* V,
                                const void* mag, double rho, void* gR_out, void* gX_out, void* gU_out, void* gmag_accum);
/* adjoint of specinv_istft: cotangent of x (B, L) -> cotangent of the spectrogram (B, F, T) complex */
int specinv_istft_adjoint(specinv_plan* plan, const void* g_x, void* g_spec_out);
/* adjoint of specinv_stft: cotangent of the spectrogram -> cotangent of x (B, length) */
int specinv_stft_adjoint(specinv_plan* plan, const void* g_spec, int64_t length, void* g_x_out);
/* adjoint of specinv_phase_init: gmag += d/dmag of <g_spec, phase_init(mag)> */
int specinv_phase_init_adjoint(specinv_plan* plan, const void* mag, const void* g_spec, void* gmag_accum);

/* ---- RTISI-LA (methods.py:273-412) ------------------------------------------------------ */
int specinv_rtisi_run(specinv_plan* plan, const void* mag, int look_ahead, int asymmetric_window,
                      int max_iter, double alpha, void* x_out);

/* RTISI_LA with a gradient (the reference's result is differentiable w.r.t. spec: test/test_rtisila.py:58-70).
 * `_run_recorded` is specinv_rtisi_run on the generic kernel that also stores every pre-projection spectrum in
 * `rec` (specinv_rtisi_record_elems() complex numbers of the plan's dtype); `_adjoint` sweeps the recursion
 * backwards: g_x (B, L) -> gmag_out (B, F, T), overwritten. */
int specinv_rtisi_record_elems(specinv_plan* plan, int look_ahead, int max_iter, int64_t* n_complex_out);
int specinv_rtisi_run_recorded(specinv_plan* plan, const void* mag, int look_ahead, int asymmetric_window, int max_iter,
                               double alpha, void* x_out, void* rec_out);
int specinv_rtisi_adjoint(specinv_plan* plan, const void* mag, const void* rec, const void* g_x, int look_ahead,
                          int asymmetric_window, int max_iter, double alpha, void* gmag_out);

/* Streaming RTISI-LA: the recursion of methods.py:363-404 fed a few frames at a time (what "real-time" in its name
 * is about, :275-278).  The plan's n_frames is the largest number of frames one push may carry; the stream length
 * is open-ended.  `push` takes (B, F, k) magnitudes and appends the samples that became final to x_out (row b at
 * x_out + b*out_stride; at most k*hop of them, *n_out per row, the first pushes give fewer: a sample is final once
 * look_ahead more frames are known).  `flush` ends the signal and returns the rest (at most look_ahead*hop + n_fft).
 * Concatenated, the pieces are the (B, L) result of specinv_rtisi_run on the whole spectrogram. */
int specinv_rtisi_stream_begin(specinv_plan* plan, int look_ahead, int asymmetric_window, int max_iter, double alpha);
int specinv_rtisi_stream_push(specinv_plan* plan, const void* mag, int k, void* x_out, int64_t out_stride,
                              int64_t* n_out);
int specinv_rtisi_stream_flush(specinv_plan* plan, void* x_out, int64_t out_stride, int64_t* n_out);

/* ---- L_BFGS building blocks (methods.py:509-569 + torch.optim.LBFGS) -------------------- */
/* transform kinds for the fused forward/backward: V = |STFT(x)| or V = log1p(M |STFT(x)|) */
enum { SPECINV_TF_MAG = 0, SPECINV_TF_LOGMEL = 1 };
/* mel_fb: (n_mels, F) device array or NULL for SPECINV_TF_MAG */
int specinv_transform_setup(specinv_plan* plan, int kind, const void* mel_fb, int n_mels);
/* V = transform(x): x (B, L_x) -> (B, n_out, T) with n_out = F or n_mels */
int specinv_transform_forward(specinv_plan* plan, const void* x, int64_t length, void* v_out);
/* loss = mean((transform(x) - target)^2) and d loss / d x (methods.py:545-550) */
int specinv_transform_loss_grad(specinv_plan* plan, const void* x, int64_t length, const void* target,
                                double* loss_host, void* grad_out);
/* How the last specinv_transform_loss_grad / _dev / lbfgs_dev_step of this plan evaluated the objective (diagnostics; the tests
 * assert that every implementation is covered): 0 a chain of kernels with the spectrum in device memory (any configuration),
 * 1 one launch, the mel contractions on the matrix cores (float32, one-sided, n_fft 1024 / 2048, <= 128 bands, any matrix),
 * 2 one launch, the filterbank in band form on the vector units (a sparse matrix - at most 4 rows per bin, <= 140 bands:
 * what a mel filterbank is; SPECINV_OBJ_SPARSE=0 in the environment selects 1 instead), 3 the frame walk (a wave per chunk of
 * frames, the contractions on each frame's own spectrum: hop = n_fft / 2, / 4, / 8, centred, at most two rows per bin; SPECINV_OBJ_WALK=0
 * selects 2 instead), -1 none yet. */
int specinv_transform_objective_kind(const specinv_plan* plan);
/* flat-vector kernels of the two-loop recursion (n elements of the plan dtype) */
int specinv_vec_dot(specinv_plan* plan, const void* a, const void* b, int64_t n, double* out_host);
int specinv_vec_axpy(specinv_plan* plan, double alpha, const void* x, void* y, int64_t n);
int specinv_vec_scale(specinv_plan* plan, double alpha, const void* x, void* y, int64_t n);
int specinv_vec_absmax_abssum(specinv_plan* plan, const void* x, int64_t n, double out_host[2]);
/* Many vectors in one pass: out_host[j] = g . v_j for the k device vectors whose addresses are in the HOST array
 * vecs_host; and out = sum_j coef_host[j] * v_j (float64 accumulation, rounded once).  With the k = 2m vectors of the
 * L-BFGS memory these two passes carry the whole two-loop recursion (the recursion itself then runs on the m x m
 * Gram matrices s_i.y_j, y_i.y_j on the host, which follow from the g . v_j of consecutive iterations by linearity):
 * 2 (2m + 1) vector reads per iteration instead of 10 m (`lbfgs.py:LBFGS._direction_gram`). */
int specinv_vec_multi_dot(specinv_plan* plan, const void* g, const void* const* vecs_host, int k, int64_t n,
                          double* out_host);
int specinv_vec_lincomb(specinv_plan* plan, const void* const* vecs_host, const double* coef_host, int k, int64_t n,
                        void* out);
/* The same, and the step x += t * out in the same pass (torch.optim.LBFGS.step: "p.add_(d, alpha=t)" right after the
 * direction): out is rounded first, x then takes fma(t, out, x) - what a separate specinv_vec_axpy would compute. */
int specinv_vec_lincomb_step(specinv_plan* plan, const void* const* vecs_host, const double* coef_host, int k, int64_t n,
                             void* out, double t, void* x);
/* The whole L-BFGS two-loop recursion d = -H g on the device (torch.optim.LBFGS.step's "compute the approximate
 * inverse Hessian multiplied by the gradient"): s_list_host / y_list_host are HOST arrays of m device pointers
 * (old_stps / old_dirs, oldest first), rho_host[m] = 1 / (y_i . s_i).  All 2m dot products stay on the device
 * (no host synchronisation); d_out receives the direction. */
int specinv_lbfgs_direction(specinv_plan* plan, const void* g, const void* const* s_list_host,
                            const void* const* y_list_host, const double* rho_host, int m, double h_diag,
                            void* d_out, int64_t n);
/* fused passes of one L-BFGS iteration (torch.optim.LBFGS.step): the curvature pair y = g - g_prev, s = t*d with
 * out_host = {y.s, y.y}; and the statistics of a step, out_host = {g.d, sum|g|, max|g|, max|d|}. */
int specinv_lbfgs_pair(specinv_plan* plan, const void* g, const void* g_prev, const void* d, double t, void* y_out,
                       void* s_out, int64_t n, double* out_host);
int specinv_lbfgs_stats(specinv_plan* plan, const void* g, const void* d, int64_t n, double* out_host);

/* The same passes with their scalar results left in DEVICE memory (doubles) and no host synchronisation, so that one
 * L-BFGS iteration (objective, step statistics, curvature pair, memory products) is enqueued back to back and the host
 * reads everything it needs for its decisions (torch.optim.LBFGS.step: y.s > 1e-10, the tolerance tests) with ONE
 * synchronisation: specinv_read_doubles.  pair: out_dev[4] = {y.s, y.y, g.g, g.g_prev}; stats: out_dev[4] as above;
 * multi_dot: out_dev[k]; loss_grad: *loss_dev = the loss. */
int specinv_transform_loss_grad_dev(specinv_plan* plan, const void* x, int64_t length, const void* target,
                                    double* loss_dev, void* grad_out);
/* The objective and the step statistics of its gradient in one go: out5_dev = {loss, g.d, sum|g|, max|g|, max|d|} with g the
 * gradient written to grad_out and d a device vector of the signal's shape (NULL: d = g) - what a trial point of the strong-Wolfe
 * search needs (torch.optim.lbfgs._strong_wolfe: f_new, gtd_new; torch.optim.LBFGS.step: opt_cond, d.abs().max()).  Where the
 * one-launch objective serves the plan the statistics are taken inside its launches, as each gradient sample becomes final;
 * otherwise by specinv_lbfgs_stats_dev's pass. */
int specinv_transform_loss_grad_stats_dev(specinv_plan* plan, const void* x, int64_t length, const void* target, const void* d,
                                          double* out5_dev, void* grad_out);
int specinv_vec_multi_dot_dev(specinv_plan* plan, const void* g, const void* const* vecs_host, int k, int64_t n,
                              double* out_dev);
int specinv_lbfgs_pair_dev(specinv_plan* plan, const void* g, const void* g_prev, const void* d, double t, void* y_out,
                           void* s_out, int64_t n, double* out_dev);
int specinv_lbfgs_stats_dev(specinv_plan* plan, const void* g, const void* d, int64_t n, double* out_dev);
/* both in one pass over g, g_prev, d: out_dev[8] = {g.d, sum|g|, max|g|, max|d|, y.s, y.y, g.g, g.g_prev} */
int specinv_lbfgs_pair_stats_dev(specinv_plan* plan, const void* g, const void* g_prev, const void* d, double t,
                                 void* y_out, void* s_out, int64_t n, double* out_dev);
/* n doubles from device memory to the host, after everything enqueued on the plan's stream so far */
int specinv_read_doubles(specinv_plan* plan, const double* src_dev, int n, double* out_host);
/* A board of n doubles in pinned host memory that the device writes directly (pass *dev_out + slot to the *_dev entry points):
 * the scalars of an iteration land in host memory without a copy kernel; specinv_stream_wait (everything enqueued on the
 * plan's stream so far has finished) makes *host_out readable.  The board lives as long as the plan. */
int specinv_board_alloc(specinv_plan* plan, int n, double** host_out, double** dev_out);
int specinv_stream_wait(specinv_plan* plan);

/* ---- L-BFGS with the decisions on the device (float32, the one-launch objective of specinv_transform_setup) --------------
 * Replaces the inner loop of torch.optim.LBFGS.step as torch_specinv/methods.py:553 drives it (no line search): one call
 * enqueues a whole optimizer.step - for each inner iteration the objective (which takes the statistics of its gradient on the
 * way) and ONE kernel that decides (tolerance tests, y.s > 1e-10, step length) and forms direction + step; with pairs in the
 * memory also the products with them and a one-workgroup decision kernel (memory ring, two-loop recursion on Gram matrices) -
 * and synchronises ONCE, at the end; after a break the rest of the enqueued step runs as no-ops.  The optimiser's state (memory, d, t, previous gradient / loss, counters) lives on the device between steps.
 * Options carry torch.optim.LBFGS's names; max_eval <= 0 means max_iter * 5 / 4.  history_size <= 120.
 * SPECINV_EUNSUPPORTED when the plan's transform is not served by the one-launch objective (the caller then runs the
 * host-driven loop on the *_dev entry points above). */
typedef struct specinv_lbfgs_opts {
  double lr, tolerance_grad, tolerance_change;
  int32_t max_iter, max_eval, history_size;
  int32_t time_objective;  /* k > 0: bracket every k-th objective evaluation of a step with HIP events (benchmarks:
                              specinv_lbfgs_info.objective_ms / objective_timed; an event pair costs ~10 us of the timeline) */
} specinv_lbfgs_opts;
typedef struct specinv_lbfgs_info {
  double first_loss;      /* what optimizer.step returns: the loss at the entry evaluation */
  double loss, t;
  int32_t total_iters, func_evals, n_iter, history_len, pairs_accepted, pairs_rejected;
  int32_t objective_launches;  /* evaluations this step executed, ... */
  int32_t objective_timed;     /* ... how many of them were bracketed with events (time_objective) ... */
  double objective_ms;         /* ... and the summed duration of those (objective + epilogue launch) */
  int32_t lean_iterations;     /* iterations enqueued so far in the lean form (objective, epilogue, decision + direction: memory empty) */
  int32_t full_iterations;     /* ... in the full form (+ memory products, one-workgroup decision kernel) */
  int32_t suspensions;         /* lean chains that met a non-empty memory and were resumed in the full form */
  int32_t reserved_;
} specinv_lbfgs_info;
/* `n` = elements of the parameter (batch * length); *handle_out identifies the optimiser within the plan */
int specinv_lbfgs_dev_create(specinv_plan* plan, int64_t n, const specinv_lbfgs_opts* opts, int32_t* handle_out);
/* one optimizer.step on x (batch, length), updated in place; target as in specinv_transform_loss_grad */
int specinv_lbfgs_dev_step(specinv_plan* plan, int32_t handle, void* x, int64_t length, const void* target,
                           specinv_lbfgs_info* info_out);
int specinv_lbfgs_dev_destroy(specinv_plan* plan, int32_t handle);

#ifdef __cplusplus
}
#endif
#endif /* SPECINV_H */
