// libspecinv.so - C ABI entry points (include/specinv.h).  gfx950 only.
#include <cstdarg>
#include <cstdio>
#include <new>
#include <vector>

#include "plan_impl.h"

namespace specinv {

std::string& last_error() {
  static thread_local std::string msg;
  return msg;
}

int fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  last_error() = buf;
  return code;
}

std::unique_ptr<PlanBase> make_plan_f32() { return std::unique_ptr<PlanBase>(new PlanT<float>()); }
std::unique_ptr<PlanBase> make_plan_f64() { return std::unique_ptr<PlanBase>(new PlanT<double>()); }

// metrics.py:14 (sc, dB), :28-29 (snr), :43 (ser) from the whole-tensor sums
// sums = { sum((out-target)^2), sum(out^2), sum(target^2), count }
double metric_from_sums(int metric, const double s[4]) {
  switch (metric) {
    case SPECINV_METRIC_SC:
      return 20.0 * (std::log10(std::sqrt(s[0])) - std::log10(std::sqrt(s[2])));
    case SPECINV_METRIC_SNR:
      return -10.0 * std::log10(s[0] / s[2]);
    default:
      return 10.0 * (std::log10(s[1]) - std::log10(s[0]));
  }
}

// _training_loop, methods.py:153-190.  Iterations between evaluations are enqueued back to
// back; the host only synchronises at an evaluation.
int PlanBase::run_loop(int max_iter, int eva_iter, double tol, int metric, specinv_eval* evals, int* n_evals,
                       int* iters_done, specinv_eval_cb cb, void* user) {
  SI_CHECK(eva_iter > 0, SPECINV_EINVAL, "eva_iter must be > 0");    // :163
  SI_CHECK(max_iter > 0, SPECINV_EINVAL, "max_iter must be > 0");    // :164
  SI_CHECK(tol >= 0, SPECINV_EINVAL, "tol must be >= 0");            // :165
  SI_CHECK(metric >= 0 && metric <= 2, SPECINV_EINVAL, "unknown metric");  // :168
  double init_loss = 0, prev = 0;
  bool have_init = false;
  int done = 0, ne = 0;
  if (tol == 0.0 && cb == nullptr) {
    // The stop rule `(prev - loss)/init < 0 and prev > loss` (:188) can never fire with tol == 0, and
    // nobody watches the evaluations: enqueue everything, read all sums back once.
    const int n_slots = max_iter / eva_iter;
    SI_TRY(begin_deferred(n_slots));
    int rc = SPECINV_OK;
    while (done < max_iter && rc == SPECINV_OK) {
      const int until_eval = eva_iter - (done % eva_iter);
      if (done + until_eval > max_iter) {
        rc = iterate(max_iter - done, false, nullptr);
        done = max_iter;
        break;
      }
      deferred_slot = ne;
      rc = iterate(until_eval, true, nullptr);
      deferred_slot = -1;
      done += until_eval;
      ++ne;
    }
    deferred_slot = -1;
    SI_TRY(rc);
    std::vector<double> all((size_t)std::max(1, ne) * 4);
    SI_TRY(read_deferred(ne, all.data()));
    for (int i = 0; i < ne && evals; ++i) {
      evals[i].iteration = (i + 1) * eva_iter - 1;
      evals[i].metric = metric_from_sums(metric, &all[4 * i]);
      evals[i].loss = all[4 * i] / all[4 * i + 3];
    }
    if (n_evals) *n_evals = ne;
    if (iters_done) *iters_done = done;
    return SPECINV_OK;
  }
  while (done < max_iter) {
    const int until_eval = eva_iter - (done % eva_iter);             // next i with i % eva == eva-1
    if (done + until_eval > max_iter) {
      SI_TRY(iterate(max_iter - done, false, nullptr));
      done = max_iter;
      break;
    }
    double s[4];
    SI_TRY(iterate(until_eval, true, s));
    done += until_eval;
    specinv_eval ev;
    ev.iteration = done - 1;
    ev.metric = metric_from_sums(metric, s);
    ev.loss = s[0] / s[3];                                            // F.mse_loss, :182
    if (evals) evals[ne] = ev;
    ++ne;
    if (cb && cb(&ev, user) != 0) break;
    if (!have_init || init_loss == 0.0) {                             // `if not init_loss`, :186
      init_loss = ev.loss;
      have_init = true;
    } else if ((prev - ev.loss) / init_loss < tol && prev > ev.loss) {  // :188
      break;
    }
    prev = ev.loss;
  }
  if (n_evals) *n_evals = ne;
  if (iters_done) *iters_done = done;
  return SPECINV_OK;
}

}  // namespace specinv

using namespace specinv;

// Calls run on the plan's device; the calling thread's current device (which torch reads back with hipGetDevice) is
// restored on every return path.
struct DeviceGuard {
  int prev = -1, want = -1;
  hipError_t enter(int device, int64_t* sink = nullptr) {
    want = device;
    bytes_sink() = sink;
    hipError_t e = hipGetDevice(&prev);
    if (e != hipSuccess) {
      prev = -1;
      return e;
    }
    if (prev == want) {
      prev = -1;                // nothing to restore
      return hipSuccess;
    }
    return hipSetDevice(want);
  }
  ~DeviceGuard() {
    bytes_sink() = nullptr;
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

#define PLAN_OR_FAIL(p) SI_CHECK((p) != nullptr && (p)->impl, SPECINV_EINVAL, "plan is NULL")

extern "C" {

const char* specinv_last_error(void) { return last_error().c_str(); }
int specinv_abi_version(void) { return SPECINV_ABI_VERSION; }
int specinv_has_approx(void) { return specinv_approx_units_built(); }

int specinv_plan_create(const specinv_stft_cfg* cfg, specinv_plan** out) {
  SI_CHECK(cfg && out, SPECINV_EINVAL, "null argument");
  SI_CHECK(cfg->dtype == SPECINV_F32 || cfg->dtype == SPECINV_F64, SPECINV_EINVAL, "bad dtype %d", cfg->dtype);
  specinv_plan* p = new (std::nothrow) specinv_plan();
  SI_CHECK(p, SPECINV_ENOMEM, "out of host memory");
  p->impl = cfg->dtype == SPECINV_F32 ? make_plan_f32() : make_plan_f64();
  p->impl->cfg = *cfg;
  DeviceGuard guard_;                     // setup() selects the plan's device
  (void)guard_.enter(cfg->device, &p->impl->dev_bytes);
  const int rc = p->impl->setup();
  bytes_sink() = nullptr;                 // (p may be deleted below)
  if (rc != SPECINV_OK) {
    delete p;
    return rc;
  }
  *out = p;
  return SPECINV_OK;
}

int specinv_plan_destroy(specinv_plan* plan) {
  if (plan) {
    DeviceGuard guard_;
    if (plan->impl) (void)guard_.enter(plan->impl->cfg.device);
    delete plan;
  }
  return SPECINV_OK;
}

int specinv_plan_set_stream(specinv_plan* plan, void* hip_stream) {
  PLAN_OR_FAIL(plan);
  plan->impl->stream = static_cast<hipStream_t>(hip_stream);
  return SPECINV_OK;
}

int specinv_plan_n_freq(const specinv_plan* plan) { return plan && plan->impl ? plan->impl->n_freq : SPECINV_EINVAL; }
int64_t specinv_plan_length(const specinv_plan* plan) { return plan && plan->impl ? plan->impl->length : SPECINV_EINVAL; }
int specinv_plan_fast_path(const specinv_plan* plan) { return plan && plan->impl ? plan->impl->path_kind() : SPECINV_EINVAL; }
int specinv_transform_objective_kind(const specinv_plan* plan) { return plan && plan->impl ? plan->impl->objective_kind : SPECINV_EINVAL; }
int64_t specinv_plan_device_bytes(const specinv_plan* plan) { return plan && plan->impl ? plan->impl->dev_bytes : SPECINV_EINVAL; }
int specinv_plan_launch_geometry(const specinv_plan* plan, int32_t out[4]) {
  SI_CHECK(plan != nullptr && plan->impl && out, SPECINV_EINVAL, "null argument");
  int g[4] = {0, 0, 0, 0};
  plan->impl->launch_geometry(g);
  for (int i = 0; i < 4; ++i) out[i] = g[i];
  return SPECINV_OK;
}
int specinv_plan_keep_state(specinv_plan* plan, int on) {
  PLAN_OR_FAIL(plan);
  // one-sided plans re-read the flag on every iterate(); a two-sided float32 plan picks its KERNELS by it (the frame kernel carries
  // Y = X + U alone, keep_state takes the coverage kernels and their buffers, reserved by *_init) and latches it there: like
  // specinv_plan_set_exact it is read by the next specinv_gla_init / specinv_admm_init, a running method keeps its kernels
  plan->impl->keep_state = on != 0;
  return SPECINV_OK;
}
int specinv_plan_set_exact(specinv_plan* plan, int on) {
  PLAN_OR_FAIL(plan);
  plan->impl->exact = on != 0;          // (read by the next specinv_gla_init / specinv_admm_init: a running method keeps its kernels)
  return SPECINV_OK;
}
int specinv_plan_force_generic(specinv_plan* plan, int on) {
  PLAN_OR_FAIL(plan);
  SI_CHECK(plan->impl->method == Method::None, SPECINV_ESTATE, "cannot switch paths while a method is running");
  plan->impl->force_generic = on != 0;
  return SPECINV_OK;
}

#define ENTER(plan)                                 \
  PLAN_OR_FAIL(plan);                               \
  DeviceGuard guard_;                               \
  SI_HIP(guard_.enter((plan)->impl->cfg.device, &(plan)->impl->dev_bytes))

int specinv_stft(specinv_plan* plan, const void* x, int64_t length, void* spec_out) {
  ENTER(plan);
  return plan->impl->stft(x, length, spec_out);
}
int specinv_istft(specinv_plan* plan, const void* spec, void* x_out) {
  ENTER(plan);
  return plan->impl->istft(spec, x_out);
}
int specinv_envelope(specinv_plan* plan, void* env_out) {
  ENTER(plan);
  return plan->impl->envelope(env_out);
}
int specinv_phase_init(specinv_plan* plan, const void* mag, void* spec_out) {
  ENTER(plan);
  return plan->impl->phase_init(mag, spec_out);
}
int specinv_metric_sums(specinv_plan* plan, const void* a, const void* b, int64_t n, double sums_host[4]) {
  ENTER(plan);
  SI_CHECK(sums_host, SPECINV_EINVAL, "sums_host is NULL");
  return plan->impl->metric_sums(a, b, n, sums_host);
}

int specinv_gla_init(specinv_plan* plan, const void* init_spec, const void* mag, double alpha) {
  ENTER(plan);
  return plan->impl->gla_init(init_spec, mag, alpha);
}
int specinv_gla_iterate(specinv_plan* plan, int n_iter, int eval_last, double sums_host[4]) {
  ENTER(plan);
  SI_CHECK(plan->impl->method == Method::Gla, SPECINV_ESTATE, "specinv_gla_init has not been called");
  return plan->impl->iterate(n_iter, eval_last != 0, sums_host);
}
int specinv_gla_run(specinv_plan* plan, int max_iter, int eva_iter, double tol, int metric, specinv_eval* evals_out,
                    int* n_evals_out, int* iters_done_out, specinv_eval_cb cb, void* user) {
  ENTER(plan);
  SI_CHECK(plan->impl->method == Method::Gla, SPECINV_ESTATE, "specinv_gla_init has not been called");
  return plan->impl->run_loop(max_iter, eva_iter, tol, metric, evals_out, n_evals_out, iters_done_out, cb, user);
}

int specinv_admm_init(specinv_plan* plan, const void* init_spec, const void* mag, double rho) {
  ENTER(plan);
  return plan->impl->admm_init(init_spec, mag, rho);
}
int specinv_admm_iterate(specinv_plan* plan, int n_iter, int eval_last, double sums_host[4]) {
  ENTER(plan);
  SI_CHECK(plan->impl->method == Method::Admm, SPECINV_ESTATE, "specinv_admm_init has not been called");
  return plan->impl->iterate(n_iter, eval_last != 0, sums_host);
}
int specinv_admm_run(specinv_plan* plan, int max_iter, int eva_iter, double tol, int metric, specinv_eval* evals_out,
                     int* n_evals_out, int* iters_done_out, specinv_eval_cb cb, void* user) {
  ENTER(plan);
  SI_CHECK(plan->impl->method == Method::Admm, SPECINV_ESTATE, "specinv_admm_init has not been called");
  return plan->impl->run_loop(max_iter, eva_iter, tol, metric, evals_out, n_evals_out, iters_done_out, cb, user);
}

int specinv_iterate_eval_dev(specinv_plan* plan, int n_iter, void* sums_dev) {
  ENTER(plan);
  SI_CHECK(plan->impl->method != Method::None, SPECINV_ESTATE, "specinv_gla_init / specinv_admm_init has not been called");
  SI_CHECK(sums_dev != nullptr && n_iter >= 1, SPECINV_EINVAL, "bad arguments");
  plan->impl->eval_dev_out = static_cast<double*>(sums_dev);
  const int rc = plan->impl->iterate(n_iter, true, nullptr);
  plan->impl->eval_dev_out = nullptr;
  return rc;
}

int specinv_get_wave(specinv_plan* plan, void* x_out) {
  ENTER(plan);
  return plan->impl->get_wave(x_out);
}
int specinv_get_state_spec(specinv_plan* plan, int which, void* spec_out) {
  ENTER(plan);
  return plan->impl->get_state_spec(which, spec_out);
}

int specinv_gla_update(specinv_plan* plan, const void* R, const void* P, const void* mag, double lr, void* S_out,
                       void* Q_out) {
  ENTER(plan);
  return plan->impl->gla_update(R, P, mag, lr, S_out, Q_out);
}
int specinv_gla_update_adjoint(specinv_plan* plan, const void* gQ, const void* gP_next, const void* S, const void* mag,
                               double lr, void* gR_out, void* gP_out, void* gmag_accum) {
  ENTER(plan);
  return plan->impl->gla_update_adjoint(gQ, gP_next, S, mag, lr, gR_out, gP_out, gmag_accum);
}
int specinv_admm_update(specinv_plan* plan, const void* R, const void* X, const void* U, const void* mag, double rho,
                        void* Xn_out, void* Un_out, void* V_out, void* Yn_out) {
  ENTER(plan);
  return plan->impl->admm_update(R, X, U, mag, rho, Xn_out, Un_out, V_out, Yn_out);
}
int specinv_admm_update_adjoint(specinv_plan* plan, const void* gYn, const void* gXn, const void* gUn, const void* V,
                                const void* mag, double rho, void* gR_out, void* gX_out, void* gU_out, void* gmag_accum) {
  ENTER(plan);
  return plan->impl->admm_update_adjoint(gYn, gXn, gUn, V, mag, rho, gR_out, gX_out, gU_out, gmag_accum);
}
int specinv_istft_adjoint(specinv_plan* plan, const void* g_x, void* g_spec_out) {
  ENTER(plan);
  return plan->impl->istft_adjoint(g_x, g_spec_out);
}
int specinv_stft_adjoint(specinv_plan* plan, const void* g_spec, int64_t length, void* g_x_out) {
  ENTER(plan);
  return plan->impl->stft_adjoint(g_spec, length, g_x_out);
}
int specinv_phase_init_adjoint(specinv_plan* plan, const void* mag, const void* g_spec, void* gmag_accum) {
  ENTER(plan);
  return plan->impl->phase_init_adjoint(mag, g_spec, gmag_accum);
}

int specinv_rtisi_run(specinv_plan* plan, const void* mag, int look_ahead, int asymmetric_window, int max_iter,
                      double alpha, void* x_out) {
  ENTER(plan);
  return plan->impl->rtisi_run(mag, look_ahead, asymmetric_window, max_iter, alpha, x_out);
}

int specinv_rtisi_record_elems(specinv_plan* plan, int look_ahead, int max_iter, int64_t* n_complex_out) {
  ENTER(plan);
  return plan->impl->rtisi_record_elems(look_ahead, max_iter, n_complex_out);
}
int specinv_rtisi_run_recorded(specinv_plan* plan, const void* mag, int look_ahead, int asymmetric_window, int max_iter,
                               double alpha, void* x_out, void* rec_out) {
  ENTER(plan);
  return plan->impl->rtisi_run_recorded(mag, look_ahead, asymmetric_window, max_iter, alpha, x_out, rec_out);
}
int specinv_rtisi_adjoint(specinv_plan* plan, const void* mag, const void* rec, const void* g_x, int look_ahead,
                          int asymmetric_window, int max_iter, double alpha, void* gmag_out) {
  ENTER(plan);
  return plan->impl->rtisi_adjoint(mag, rec, g_x, look_ahead, asymmetric_window, max_iter, alpha, gmag_out);
}
int specinv_rtisi_stream_begin(specinv_plan* plan, int look_ahead, int asymmetric_window, int max_iter, double alpha) {
  ENTER(plan);
  return plan->impl->rtisi_stream_begin(look_ahead, asymmetric_window, max_iter, alpha);
}
int specinv_rtisi_stream_push(specinv_plan* plan, const void* mag, int k, void* x_out, int64_t out_stride, int64_t* n_out) {
  ENTER(plan);
  return plan->impl->rtisi_stream_push(mag, k, x_out, out_stride, n_out);
}
int specinv_rtisi_stream_flush(specinv_plan* plan, void* x_out, int64_t out_stride, int64_t* n_out) {
  ENTER(plan);
  return plan->impl->rtisi_stream_flush(x_out, out_stride, n_out);
}

int specinv_transform_setup(specinv_plan* plan, int kind, const void* mel_fb, int n_mels) {
  ENTER(plan);
  return plan->impl->transform_setup(kind, mel_fb, n_mels);
}
int specinv_transform_forward(specinv_plan* plan, const void* x, int64_t length, void* v_out) {
  ENTER(plan);
  return plan->impl->transform_forward(x, length, v_out);
}
int specinv_transform_loss_grad(specinv_plan* plan, const void* x, int64_t length, const void* target,
                                double* loss_host, void* grad_out) {
  ENTER(plan);
  return plan->impl->transform_loss_grad(x, length, target, loss_host, grad_out);
}
int specinv_transform_loss_grad_dev(specinv_plan* plan, const void* x, int64_t length, const void* target,
                                    double* loss_dev, void* grad_out) {
  ENTER(plan);
  SI_CHECK(loss_dev, SPECINV_EINVAL, "loss_dev is NULL");
  return plan->impl->transform_loss_grad(x, length, target, nullptr, grad_out, loss_dev);
}
int specinv_transform_loss_grad_stats_dev(specinv_plan* plan, const void* x, int64_t length, const void* target, const void* d,
                                          double* out5_dev, void* grad_out) {
  ENTER(plan);
  SI_CHECK(out5_dev, SPECINV_EINVAL, "out5_dev is NULL");
  return plan->impl->transform_loss_grad(x, length, target, nullptr, grad_out, out5_dev, true, d);
}
int specinv_vec_dot(specinv_plan* plan, const void* a, const void* b, int64_t n, double* out_host) {
  ENTER(plan);
  return plan->impl->vec_dot(a, b, n, out_host);
}
int specinv_vec_axpy(specinv_plan* plan, double alpha, const void* x, void* y, int64_t n) {
  ENTER(plan);
  return plan->impl->vec_axpy(alpha, x, y, n);
}
int specinv_vec_scale(specinv_plan* plan, double alpha, const void* x, void* y, int64_t n) {
  ENTER(plan);
  return plan->impl->vec_scale(alpha, x, y, n);
}
int specinv_vec_absmax_abssum(specinv_plan* plan, const void* x, int64_t n, double out_host[2]) {
  ENTER(plan);
  return plan->impl->vec_absmax_abssum(x, n, out_host);
}

int specinv_vec_multi_dot(specinv_plan* plan, const void* g, const void* const* vecs_host, int k, int64_t n,
                          double* out_host) {
  ENTER(plan);
  return plan->impl->vec_multi_dot(g, vecs_host, k, n, out_host);
}
int specinv_vec_lincomb(specinv_plan* plan, const void* const* vecs_host, const double* coef_host, int k, int64_t n,
                        void* out) {
  ENTER(plan);
  return plan->impl->vec_lincomb(vecs_host, coef_host, k, n, out);
}
int specinv_vec_lincomb_step(specinv_plan* plan, const void* const* vecs_host, const double* coef_host, int k, int64_t n,
                             void* out, double t, void* x) {
  ENTER(plan);
  return plan->impl->vec_lincomb_step(vecs_host, coef_host, k, n, out, t, x);
}

int specinv_lbfgs_direction(specinv_plan* plan, const void* g, const void* const* s_list_host,
                            const void* const* y_list_host, const double* rho_host, int m, double h_diag, void* d_out,
                            int64_t n) {
  ENTER(plan);
  return plan->impl->lbfgs_direction(g, s_list_host, y_list_host, rho_host, m, h_diag, d_out, n);
}
int specinv_lbfgs_pair(specinv_plan* plan, const void* g, const void* g_prev, const void* d, double t, void* y_out,
                       void* s_out, int64_t n, double* out_host) {
  ENTER(plan);
  return plan->impl->lbfgs_pair(g, g_prev, d, t, y_out, s_out, n, out_host);
}
int specinv_lbfgs_stats(specinv_plan* plan, const void* g, const void* d, int64_t n, double* out_host) {
  ENTER(plan);
  return plan->impl->lbfgs_stats(g, d, n, out_host);
}

int specinv_vec_multi_dot_dev(specinv_plan* plan, const void* g, const void* const* vecs_host, int k, int64_t n,
                              double* out_dev) {
  ENTER(plan);
  SI_CHECK(out_dev, SPECINV_EINVAL, "out_dev is NULL");
  return plan->impl->vec_multi_dot(g, vecs_host, k, n, nullptr, out_dev);
}
int specinv_lbfgs_pair_dev(specinv_plan* plan, const void* g, const void* g_prev, const void* d, double t, void* y_out,
                           void* s_out, int64_t n, double* out_dev) {
  ENTER(plan);
  SI_CHECK(out_dev, SPECINV_EINVAL, "out_dev is NULL");
  return plan->impl->lbfgs_pair(g, g_prev, d, t, y_out, s_out, n, nullptr, out_dev);
}
int specinv_lbfgs_stats_dev(specinv_plan* plan, const void* g, const void* d, int64_t n, double* out_dev) {
  ENTER(plan);
  SI_CHECK(out_dev, SPECINV_EINVAL, "out_dev is NULL");
  return plan->impl->lbfgs_stats(g, d, n, nullptr, out_dev);
}
int specinv_lbfgs_pair_stats_dev(specinv_plan* plan, const void* g, const void* g_prev, const void* d, double t,
                                 void* y_out, void* s_out, int64_t n, double* out_dev) {
  ENTER(plan);
  SI_CHECK(out_dev, SPECINV_EINVAL, "out_dev is NULL");
  return plan->impl->lbfgs_pair_stats(g, g_prev, d, t, y_out, s_out, n, out_dev);
}
int specinv_read_doubles(specinv_plan* plan, const double* src_dev, int n, double* out_host) {
  ENTER(plan);
  return plan->impl->read_doubles(src_dev, n, out_host);
}
int specinv_board_alloc(specinv_plan* plan, int n, double** host_out, double** dev_out) {
  ENTER(plan);
  return plan->impl->board_alloc(n, host_out, dev_out);
}
int specinv_stream_wait(specinv_plan* plan) {
  ENTER(plan);
  return plan->impl->stream_wait();
}
int specinv_lbfgs_dev_create(specinv_plan* plan, int64_t n, const specinv_lbfgs_opts* opts, int32_t* handle_out) {
  ENTER(plan);
  return plan->impl->lbfgs_dev_create(n, opts, handle_out);
}
int specinv_lbfgs_dev_step(specinv_plan* plan, int32_t handle, void* x, int64_t length, const void* target,
                           specinv_lbfgs_info* info_out) {
  ENTER(plan);
  return plan->impl->lbfgs_dev_step(handle, x, length, target, info_out);
}
int specinv_lbfgs_dev_destroy(specinv_plan* plan, int32_t handle) {
  ENTER(plan);
  return plan->impl->lbfgs_dev_destroy(handle);
}

}  // extern "C"
