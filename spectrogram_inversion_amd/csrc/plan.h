// Plan object behind the C ABI (include/specinv.h).
#pragma once
#include <memory>
#include <vector>

#include "common.h"

namespace specinv {

enum class Method { None, Gla, Admm };

// Type-erased interface; PlanT<float> / PlanT<double> implement it (plan_impl.h).
struct PlanBase {
  specinv_stft_cfg cfg{};
  int n_freq = 0;
  int pad = 0;
  int64_t length = 0;  // (T-1)*hop + n_fft - 2*pad
  hipStream_t stream = nullptr;
  Method method = Method::None;
  bool force_generic = false;
  bool exact = true;         // projection / envelope division in the reference's operation order (specinv_plan_set_exact; 0: approximations)
  bool keep_state = false;   // ADMM: the last iteration of every iterate() also writes X and U (specinv_plan_keep_state)
  bool keep_latched = false; // ... as gla_init / admm_init found it: where the flag selects the kernels (two-sided float32) a run keeps them
  int64_t dev_bytes = 0;   // device memory held by the plan's buffers (account_bytes)
  int objective_kind = -1;   // what the last transform_loss_grad ran (specinv_transform_objective_kind)

  virtual ~PlanBase() = default;
  virtual int setup() = 0;
  virtual bool fast_path() const = 0;
  virtual int path_kind() const = 0;   // 0 generic, 1 fused, 2 frame kernel + overlap-add
  virtual void launch_geometry(int out[4]) const = 0;

  virtual int stft(const void* x, int64_t len, void* spec_out) = 0;
  virtual int istft(const void* spec, void* x_out) = 0;
  virtual int envelope(void* env_out) = 0;
  virtual int phase_init(const void* mag, void* spec_out) = 0;
  virtual int metric_sums(const void* a, const void* b, int64_t n, double sums[4]) = 0;

  virtual int gla_init(const void* init_spec, const void* mag, double alpha) = 0;
  virtual int admm_init(const void* init_spec, const void* mag, double rho) = 0;
  virtual int iterate(int n_iter, bool eval_last, double sums[4]) = 0;
  // evaluations whose result cannot influence the run (tol == 0, no callback) stay on the device and are
  // read back once at the end: deferred_slot >= 0 makes iterate() park its sums in that slot
  int deferred_slot = -1;
  // ... or to device memory the caller names (specinv_iterate_eval_dev: 4 doubles, nothing waits): the multi-GPU loop all-reduces
  // them where they are and reads the result once
  double* eval_dev_out = nullptr;
  virtual int begin_deferred(int n_slots) = 0;
  virtual int read_deferred(int n_slots, double* out) = 0;
  virtual int get_wave(void* x_out) = 0;
  virtual int get_state_spec(int which, void* spec_out) = 0;

  virtual int gla_update(const void* R, const void* P, const void* mag, double lr, void* S_out, void* Q_out) = 0;
  virtual int gla_update_adjoint(const void* gQ, const void* gPn, const void* S, const void* mag, double lr, void* gR,
                                 void* gP, void* gmag) = 0;
  virtual int admm_update(const void* R, const void* X, const void* U, const void* mag, double rho, void* Xn, void* Un,
                          void* V, void* Yn) = 0;
  virtual int admm_update_adjoint(const void* gYn, const void* gXn, const void* gUn, const void* V, const void* mag,
                                  double rho, void* gR, void* gX, void* gU, void* gmag) = 0;
  virtual int istft_adjoint(const void* g_x, void* g_spec_out) = 0;
  virtual int stft_adjoint(const void* g_spec, int64_t len, void* g_x_out) = 0;
  virtual int phase_init_adjoint(const void* mag, const void* g_spec, void* gmag) = 0;

  virtual int rtisi_run(const void* mag, int look_ahead, int asym, int max_iter, double alpha, void* x_out) = 0;
  virtual int rtisi_record_elems(int look_ahead, int max_iter, int64_t* out) = 0;
  virtual int rtisi_run_recorded(const void* mag, int look_ahead, int asym, int max_iter, double alpha, void* x_out,
                                 void* rec_out) = 0;
  virtual int rtisi_adjoint(const void* mag, const void* rec, const void* g_x, int look_ahead, int asym, int max_iter,
                            double alpha, void* gmag_out) = 0;
  virtual int rtisi_stream_begin(int look_ahead, int asym, int max_iter, double alpha) = 0;
  virtual int rtisi_stream_push(const void* mag, int k, void* x_out, int64_t out_stride, int64_t* n_out) = 0;
  virtual int rtisi_stream_flush(void* x_out, int64_t out_stride, int64_t* n_out) = 0;

  virtual int transform_setup(int kind, const void* mel_fb, int n_mels) = 0;
  virtual int transform_forward(const void* x, int64_t len, void* v_out) = 0;
  virtual int transform_loss_grad(const void* x, int64_t len, const void* target, double* loss, void* grad,
                                  double* loss_dev = nullptr, bool with_stats = false, const void* stat_d = nullptr) = 0;
  virtual int vec_dot(const void* a, const void* b, int64_t n, double* out) = 0;
  virtual int vec_axpy(double alpha, const void* x, void* y, int64_t n) = 0;
  virtual int vec_scale(double alpha, const void* x, void* y, int64_t n) = 0;
  virtual int vec_absmax_abssum(const void* x, int64_t n, double out[2]) = 0;
  virtual int lbfgs_direction(const void* g, const void* const* s_list, const void* const* y_list, const double* rho,
                              int m, double h_diag, void* d_out, int64_t n) = 0;
  virtual int vec_multi_dot(const void* g, const void* const* vecs, int k, int64_t n, double* out, double* out_dev = nullptr) = 0;
  virtual int vec_lincomb(const void* const* vecs, const double* coef, int k, int64_t n, void* out) = 0;
  virtual int vec_lincomb_step(const void* const* vecs, const double* coef, int k, int64_t n, void* out, double t, void* x) = 0;
  virtual int lbfgs_pair(const void* g, const void* gp, const void* d, double t, void* y, void* sv, int64_t n, double* out,
                         double* out_dev = nullptr) = 0;
  virtual int lbfgs_stats(const void* g, const void* d, int64_t n, double* out, double* out_dev = nullptr) = 0;
  virtual int lbfgs_pair_stats(const void* g, const void* gp, const void* d, double t, void* y, void* sv, int64_t n,
                               double* out8_dev) = 0;
  virtual int read_doubles(const double* src_dev, int n, double* out_host) = 0;
  virtual int board_alloc(int n, double** host_out, double** dev_out) = 0;
  virtual int stream_wait() = 0;
  virtual int lbfgs_dev_create(int64_t n, const specinv_lbfgs_opts* opts, int32_t* handle_out) = 0;
  virtual int lbfgs_dev_step(int32_t handle, void* x, int64_t len, const void* target, specinv_lbfgs_info* info) = 0;
  virtual int lbfgs_dev_destroy(int32_t handle) = 0;

  // _training_loop (methods.py:153-190) driving `iterate`
  int run_loop(int max_iter, int eva_iter, double tol, int metric, specinv_eval* evals, int* n_evals,
               int* iters_done, specinv_eval_cb cb, void* user);
};

std::unique_ptr<PlanBase> make_plan_f32();
std::unique_ptr<PlanBase> make_plan_f64();

double metric_from_sums(int metric, const double sums[4]);

}  // namespace specinv

struct specinv_plan {
  std::unique_ptr<specinv::PlanBase> impl;
};
