// PlanT<T>: owns the device state of one (device, shape, stft-args) problem and launches the
// kernels.  T = float or double.
#pragma once
#include <cmath>
#include <cstring>
#include <memory>
#include <type_traits>
#include <atomic>
#include <vector>

#include "kernels_adjoint.h"
#include "fast_state.h"
#include "kernels_generic.h"
#include "wave_api.h"
#include "kernels_lbfgs.h"
#include "kernels_big.h"
#include "lbfgs_dev.h"
#include "kernels_rtisi.h"
#include "plan.h"

namespace specinv {

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  void release() {
    if (p) {
      (void)hipFree(p);
      account_bytes(-(int64_t)bytes);
    }
    p = nullptr;
    bytes = 0;
  }
  // grow-only allocation
  int reserve(size_t n) {
    if (n <= bytes && p) return SPECINV_OK;
    release();
    if (n == 0) n = 16;
    hipError_t e = hipMalloc(&p, n);
    if (e != hipSuccess) {
      p = nullptr;
      return fail(SPECINV_ENOMEM, "hipMalloc(%zu bytes) failed: %s", n, hipGetErrorString(e));
    }
    bytes = n;
    account_bytes((int64_t)n);
    return SPECINV_OK;
  }
  template <typename U>
  U* as() const { return static_cast<U*>(p); }
};

// radix schedule of the LDS Stockham FFT: powers of two as 8s and 4s (fewest stages), then 3, 5, 7, other primes
inline std::vector<int> factorize(int n) {
  std::vector<int> f;
  int e = 0;
  while (n % 2 == 0) {
    ++e;
    n /= 2;
  }
  if (e == 1) f.push_back(2);
  else if (e % 3 == 0) f.insert(f.end(), e / 3, 8);
  else if (e % 3 == 2) {
    f.push_back(4);                       // small first radix: its stores are the strided ones
    f.insert(f.end(), e / 3, 8);
  } else if (e >= 4) {
    f.push_back(4);
    f.push_back(4);
    f.insert(f.end(), (e - 4) / 3, 8);
  }
  const int small[] = {3, 5, 7, 11, 13};
  for (int p : small)
    while (n % p == 0) {
      f.push_back(p);
      n /= p;
    }
  for (int d = 17; n > 1; d += 2)
    while (n % d == 0) {
      f.push_back(d);
      n /= d;
    }
  return f;
}

template <typename T>
struct PlanT final : PlanBase {
  using C = cplx<T>;
  DevBuf window, tw, env;
  DevBuf x, frames, specA, specB, mag, partials, sums, tmp_spec, tmp_real;
  DevBuf rt_state;                      // RTISI per-item state
  DevBuf rs_state;                      // ... of the streaming recursion (survives between pushes)
  RtisiStream<T> rstream;
  DevBuf lb_scal, lb_part;              // device scalars / partial sums of the L-BFGS passes
  DevBuf eval_log;                      // per-evaluation sums of a run with deferred read-back
  DevBuf tf_mel, tf_mel_tiled, tf_mel_tiled_t, tf_spec, tf_v;   // transform (L_BFGS) scratch
  DevBuf tf_mel_a, tf_mel_b, tf_obj_tab;   // non-zero filterbank blocks in MFMA operand order + block table (one-launch objective)
  int tf_obj_mt = 0;                    // its 16-row mel tiles (0: the objective runs as a kernel chain)
  DevBuf tf_rows;                       // statistics of the objective's gradient: the epilogue's rows (objective_args.h)
  DevBuf tf_sp_blob, tf_sp_tab;         // a sparse filterbank in band form (objective_args.h: obj_build_sparse)
  DevBuf tf_walk_blob;                  // ... and as the frame walk's tables (obj_build_walk)
  fast::ObjWalkInfo tf_walk;
  bool tf_walk_ok = false;
  fast::ObjSparseInfo tf_sp{};
  bool tf_sp_ok = false;
  std::vector<T> h_window;
  FrameCfg<T> fc{};
  size_t lds_bytes = 0;        // LDS of k_stft / k_iter_pair / k_grad_frames (one buffer when use_inplace)
  size_t lds_bytes2 = 0;       // two buffers (the kernels that keep the two-buffer transform)
  bool use_inplace = false;
  int ip_threads = 0, ip_maxb = 0;
  // frames beyond one workgroup's LDS (kernels_big.h): n_fft = big_n1 * nf, the LDS kernels transform nf points
  bool big = false;
  int big_n1 = 1, nf = 0;
  FrameCfg<T> fc2{};           // the sub-transform's stages (n_fft = nf, tw = tw2)
  DevBuf tw2, big_y;
  bool tw_lds = false;         // k_iter_pair: the twiddle table copied to LDS (small n_fft)
  bool use_dr = false;         // k_iter_pair_dr: the digit-reversed in-place transform (power-of-two n_fft)
  bool use_wave = false;       // k_wave_iter (kernels_wave.h): the iteration's frame part on a transform that lives in one wave
  int wave_last_waves = 0;     // ... waves of its last launch (an evaluating launch leaves that many pairs of partial sums)
  DevBuf x_alt, seam_l, seam_r;   // ... with the overlap-add in registers: the other signal buffer, the chunk boundaries' partial sums
  size_t dr_lds = 0;
  int dr_threads = 0;
  double sum_m2 = 0, count = 0;
  T coef = 0;  // lr (GLA) or rho (ADMM)
  FastState<T> fast;
  int tf_kind = -1, tf_mels = 0;

  int B() const { return cfg.batch; }
  int Tn() const { return cfg.n_frames; }
  int N() const { return cfg.n_fft; }
  int64_t nspec() const { return (int64_t)B() * Tn() * n_freq; }

  // ------------------------------------------------------------------------------------
  int setup() override {
    SI_CHECK(cfg.n_fft >= 2 && cfg.hop_length >= 1 && cfg.n_frames >= 1 && cfg.batch >= 1, SPECINV_EINVAL,
             "bad shape: n_fft=%d hop=%d frames=%d batch=%d", cfg.n_fft, cfg.hop_length, cfg.n_frames, cfg.batch);
    SI_CHECK(cfg.window_host != nullptr, SPECINV_EINVAL, "window_host is NULL");
    SI_CHECK(cfg.pad_mode >= 0 && cfg.pad_mode <= 3, SPECINV_EINVAL, "bad pad_mode %d", cfg.pad_mode);
    SI_CHECK(!cfg.onesided || cfg.n_fft % 2 == 0, SPECINV_EINVAL, "onesided needs an even n_fft");
    SI_HIP(hipSetDevice(cfg.device));
    n_freq = cfg.onesided ? cfg.n_fft / 2 + 1 : cfg.n_fft;
    pad = cfg.center ? cfg.n_fft / 2 : 0;
    length = (int64_t)(cfg.n_frames - 1) * cfg.hop_length + cfg.n_fft - 2 * pad;
    SI_CHECK(length >= 1, SPECINV_EINVAL, "empty signal (length %lld)", (long long)length);
    SI_CHECK(cfg.batch <= 65535, SPECINV_EUNSUPPORTED, "batch > 65535");

    const int n = cfg.n_fft;
    h_window.assign(static_cast<const T*>(cfg.window_host), static_cast<const T*>(cfg.window_host) + n);
    cfg.window_host = nullptr;
    SI_TRY(window.reserve(n * sizeof(T)));
    SI_HIP(hipMemcpy(window.p, h_window.data(), n * sizeof(T), hipMemcpyHostToDevice));

    std::vector<C> h_tw(n);
    for (int i = 0; i < n; ++i) {
      const long double a = -2.0L * 3.141592653589793238462643383279502884L * i / n;
      h_tw[i] = mk<T>((T)cosl(a), (T)sinl(a));
    }
    SI_TRY(tw.reserve(n * sizeof(C)));
    SI_HIP(hipMemcpy(tw.p, h_tw.data(), n * sizeof(C), hipMemcpyHostToDevice));

    // window-square envelope (methods.py:129-131); products in T like the reference's
    // `weight * weight`, the <= ceil(N/hop) term sums in double, rounded once
    std::vector<double> e(length, 0.0);
    for (int t = 0; t < cfg.n_frames; ++t)
      for (int k = 0; k < n; ++k) {
        const int64_t pos = (int64_t)t * cfg.hop_length + k - pad;
        if (pos >= 0 && pos < length) e[pos] += (double)(T)(h_window[k] * h_window[k]);
      }
    std::vector<T> h_env(length);
    for (int64_t i = 0; i < length; ++i) h_env[i] = (T)e[i];
    SI_TRY(env.reserve(length * sizeof(T)));
    SI_HIP(hipMemcpy(env.p, h_env.data(), length * sizeof(T), hipMemcpyHostToDevice));

    // A frame's transform lives in one workgroup's LDS up to 16384 points in float32, 8192 in float64 (in place); beyond that it is
    // cut into big_n1 in {2, 4, 8} rows of nf points and takes four steps through device memory (kernels_big.h)
    big = false;
    big_n1 = 1;
    nf = n;
    {
      const size_t cap = (size_t)128 * 1024 / sizeof(C);          // points one in-place buffer may hold
      if ((size_t)n > cap) {
        for (int n1 = 2; n1 <= 8 && !big; n1 *= 2)
          if (n % n1 == 0 && (size_t)(n / n1) * 2 <= cap) {        // (rows of half the cap: two workgroups per CU)
            big = true;
            big_n1 = n1;
            nf = n / n1;
          }
        SI_CHECK(big, SPECINV_EUNSUPPORTED, "n_fft=%d: frames above %zu points are transformed as 2, 4 or 8 rows of at most %zu",
                 n, cap, cap / 2);
      }
    }
    std::vector<int> rad = factorize(nf);
    SI_CHECK((int)rad.size() <= kMaxStages, SPECINV_EUNSUPPORTED, "n_fft=%d has too many prime factors", n);
    fc.n_fft = n;
    fc.n_freq = n_freq;
    fc.n_frames = cfg.n_frames;
    fc.hop = cfg.hop_length;
    fc.pad = pad;
    fc.pad_mode = cfg.pad_mode;
    fc.onesided = cfg.onesided;
    fc.length = length;
    fc.fwd_scale = cfg.normalized ? (T)(1.0 / std::sqrt((double)n)) : T(1);
    fc.inv_scale = cfg.normalized ? (T)(1.0 / std::sqrt((double)n)) : (T)(1.0 / n);
    fc.n_stages = (int)rad.size();
    for (size_t i = 0, ns = 1; i < rad.size(); ++i) {
      fc.radix[i] = rad[i];
      const uint64_t m = (uint64_t)ns * rad[i];
      fc.ns_magic[i] = (unsigned)(((1ull << 32) + ns - 1) / ns);      // ns = 1: wraps to 0, never used
      fc.m_magic[i] = (unsigned)(((1ull << 32) + m - 1) / m);
      fc.tw_step[i] = (int)(nf / m);
      ns = m;
    }
    fc.tw = tw.as<C>();
    fc.window = window.as<T>();
    fc.inplace = 0;
    fc.maxb = 0;
    lds_bytes2 = 2 * (size_t)nf * sizeof(C);         // two buffers: every kernel can run (RTISI-LA and the L-BFGS chain need them)
    // In-place transforms (kernels_generic.h: FrameCfg::inplace) for k_stft / k_iter_pair / k_grad_frames: one buffer - more
    // workgroups per CU for a latency-bound kernel (measured, tools/bench_generic_r04.py: float64 2048 / 512 0.58 -> 0.44 ms per
    // iteration, float64 512 two-sided 1.40 -> 1.02, float32 8192 0.64 -> 0.31; float32 at n_fft <= 1024 2 % slower: the extra
    // barrier per stage with nothing to gain) - and the only form beyond n_fft 8192 (float32) / 4096 (float64).
    bool ip_ok = true;
    for (int r : rad) ip_ok = ip_ok && (r == 2 || r == 3 || r == 4 || r == 5 || r == 7 || r == 8);
    // the smallest workgroup (whole waves, 64 ... 1024 threads) whose threads can hold every stage's butterflies
    ip_threads = 64;
    ip_maxb = 1;
    if (ip_ok) {
      auto fits = [&](int threads) {
        for (int r : rad)
          if ((nf / r + threads - 1) / threads > ip_butterflies_per_thread(r, sizeof(T) == 8)) return false;
        return true;
      };
      // (at least a thread per butterfly of the widest stage up to 256 threads, like the two-buffer form)
      int rmin = 8;
      for (int r : rad) rmin = std::min(rmin, r);
      while (ip_threads < 256 && ip_threads < nf / std::max(2, rmin)) ip_threads *= 2;
      while (ip_threads < 1024 && !fits(ip_threads)) ip_threads *= 2;
      ip_ok = fits(ip_threads);
    }
    const char* ip_env = getenv("SPECINV_GENERIC_INPLACE");           // (experiments: 0 never, 1 whenever possible)
    use_inplace = ip_ok && (sizeof(T) == 8 || lds_bytes2 > 16 * 1024 || (ip_env && ip_env[0] == '1')) && !(ip_env && ip_env[0] == '0');
    lds_bytes = use_inplace ? lds_bytes2 / 2 : lds_bytes2;
    SI_CHECK(lds_bytes <= 160 * 1024 - 256, SPECINV_EUNSUPPORTED,
             "n_fft=%d needs %zu bytes of LDS per frame (limit 160 KiB)", n, lds_bytes);
    // small transforms: the iteration kernel keeps the twiddle table in LDS (kernels_generic.h: FrameCfg::tw_lds)
    {
      const char* e1 = getenv("SPECINV_GENERIC_TWLDS");            // (experiments: 0 never)
      // (workgroups of four waves: with one-wave workgroups - n_fft 512 - LDS is what limits the workgroups per CU, and the table
      // costs more occupancy than its latency saves: float64 512 two-sided 1.03 -> 1.26 ms per iteration; float32 1024 0.492 ->
      // 0.448, float64 1000 0.259 -> 0.245)
      tw_lds = !big && (size_t)n * sizeof(C) <= 16 * 1024 && frame_threads() >= 256 && !(e1 && e1[0] == '0');
    }
    // Power-of-two n_fft: the iteration kernel on the digit-reversed in-place transform (kernels_generic.h: k_iter_pair_dr) -
    // radix 8 stages, then 4 (or 4, 4; a lone 2): the small blocks last, where no twiddles are left
    fc.dr_stages = 0;
    use_dr = false;
    if (!big && n >= 8 && (n & (n - 1)) == 0) {
      int e = 0;
      while ((1 << e) < n) ++e;
      std::vector<int> bits;
      if (e % 3 == 0) bits.assign(e / 3, 3);
      else if (e % 3 == 2) {
        bits.assign(e / 3, 3);
        bits.push_back(2);
      } else {
        bits.assign((e - 4) / 3, 3);
        bits.push_back(2);
        bits.push_back(2);
      }
      int left = e;
      for (size_t i = 0; i < bits.size(); ++i) {
        left -= bits[i];
        fc.dr_bits[i] = bits[i];
        fc.dr_shift[i] = left;                                    // log2 of n / (R_1 ... R_i)
      }
      fc.dr_stages = (int)bits.size();
      fc.dr_lg8 = e - 3;
      dr_lds = (size_t)(n + n / 32 + 2 + n / 8 + 1) * sizeof(C);   // padded frame buffer + octant twiddle table
      {
        // threads: n_fft / 4 up to 256 (the loads and the bin update like more threads than a radix-8 stage has butterflies);
        // beyond that as many as keep a thread's butterflies per stage within two trips (kernels_generic.h: DrMB), at most 1024
        const int per_trip = sizeof(T) == 8 ? 1 : 2;
        dr_threads = std::min(256, std::max(64, n / 4 / 64 * 64));
        while (dr_threads < 1024 && n / 8 > dr_threads * per_trip * 2) dr_threads *= 2;
      }
      if (const char* e2 = getenv("SPECINV_GENERIC_DR_THREADS")) dr_threads = std::max(64, std::min(1024, atoi(e2) / 64 * 64));
      const char* dr_env = getenv("SPECINV_GENERIC_DR");          // (experiments / tests: 0 keeps the Stockham kernels)
      // where it wins (round 5, tools/bench_generic_r05.py, one box): float64 n_fft 1024 / 2048 -10 ... -13 %, float32 n_fft 2048
      // (two-sided) -32 %; float64 512 +-3 %, float64 4096 +12 %, float32 <= 1024 +2 ... +5 %, n_fft >= 8192 +2 ... +20 %: those
      // keep the Stockham kernels (SPECINV_GENERIC_DR=1 forces this one wherever it fits)
      const bool wins = sizeof(T) == 8 ? (n == 1024 || n == 2048) : n == 2048;
      use_dr = dr_lds <= 160 * 1024 - 256 && !(dr_env && dr_env[0] == '0') && (wins || (dr_env && dr_env[0] == '1'));
    }
    // Power-of-two n_fft 128 ... 2048: the wave-level coverage kernel (kernels_wave.h, round 6) - float64 at every size it covers,
    // float32 at n_fft 128 / 256 (512 ... 4096 have the packed wave-level kernels of fast_core.h; what falls through those - a
    // two-sided run that keeps X and U - stays on k_iter_pair).  SPECINV_GENERIC_WAVE=0 never, =1 wherever it covers.
    {
      const char* we = getenv("SPECINV_GENERIC_WAVE");
      // measured (tools/bench_wave.py, round 6, ms per iteration against k_iter_pair / k_iter_pair_dr): float64 one-sided 2048 / 512
      // 0.386 -> 0.326, 1024 / 256 0.429 -> 0.215, 512 / 128 0.370 -> 0.240, 256 / 64 0.857 -> 0.405; float32 128 / 32 0.284 -> 0.143,
      // 256 / 64 0.455 -> 0.295 (hop = n_fft / 2, / 4, / 8: overlap-add in registers; other hops -8 ... -25 % on frames + k_ola).
      // A two-sided frame updates four bins per conjugate pair (their state requested together): 512 / 300 / 100 two-sided float64
      // 0.971 -> 0.767, float32 0.536 -> 0.466 (float32 at n_fft >= 512 normally runs k_semi2 / k_hop2, faster still).
      // With the overlap-add inside the kernel (registers or the LDS ring: every hop < n_fft, two-sided too) it is ahead of
      // k_iter_pair at float32 512 ... 2048 as well - two-sided 512 / 300 / 100 0.560 -> 0.414, 2048 / 512 0.521 -> 0.417, one-sided
      // 1024 / 800 / 200 0.448 -> 0.260, ADMM 2048 / 1200 / 300 0.470 -> 0.400; its frames + k_ola form is not (ADMM 2048: 0.518).
      // n_fft 4096 / 8192 (teams of two to eight waves per frame): ahead in every form - float32 8192 / 2048 0.305 -> 0.161,
      // frames + k_ola 0.200; 4096 / 1024 0.546 -> 0.257.
      const bool wins = sizeof(T) == 8 || n <= 256 || n >= 4096 ||
                        (wave_iter_covers(n, (int)sizeof(T)) && wave_iter_ola_chunks<T>(n, cfg.hop_length, cfg.n_frames, cfg.batch, cfg.onesided != 0) > 0);
      use_wave = !big && wave_iter_covers(n, (int)sizeof(T)) && wave_iter_fits(n, cfg.n_frames, cfg.batch, false) && !(we && we[0] == '0') &&
                 (wins || (we && we[0] == '1'));
    }
    if (std::max(lds_bytes, use_dr ? dr_lds : (size_t)0) > 48 * 1024) {
      // (the attribute belongs to the kernel, not to the plan: never lower what another plan has asked for)
      static std::atomic<int> lds_cap{0};
      const int need = (int)std::max(lds_bytes, use_dr ? dr_lds : (size_t)0);
      int seen = lds_cap.load();
      while (seen < need && !lds_cap.compare_exchange_weak(seen, need)) {}
      const int lim = std::max(seen, need);
      const void* fns[] = {(const void*)k_stft<T, false>, (const void*)k_stft<T, true>, (const void*)k_grad_frames<T, false>,
                           (const void*)k_grad_frames<T, true>,
                           (const void*)k_iter_pair<T, 0, false, false>, (const void*)k_iter_pair<T, 0, true, false>,
                           (const void*)k_iter_pair<T, 1, false, false>, (const void*)k_iter_pair<T, 1, true, false>,
                           (const void*)k_iter_pair<T, 0, false, true>, (const void*)k_iter_pair<T, 0, true, true>,
                           (const void*)k_iter_pair<T, 1, false, true>, (const void*)k_iter_pair<T, 1, true, true>,
                           (const void*)k_iter_pair<T, 0, false, true, 256>, (const void*)k_iter_pair<T, 0, true, true, 256>,
                           (const void*)k_iter_pair<T, 1, false, true, 256>, (const void*)k_iter_pair<T, 1, true, true, 256>,
                           (const void*)k_iter_pair_dr<T, 0, false>, (const void*)k_iter_pair_dr<T, 0, true>,
                           (const void*)k_iter_pair_dr<T, 1, false>, (const void*)k_iter_pair_dr<T, 1, true>};
      for (const void* fn : fns) SI_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
    }
    if (big) {
      // the sub-transform: fc's stages with its own size and twiddle table W_nf^j = W_N^(j big_n1)
      std::vector<C> h2(nf);
      for (int i = 0; i < nf; ++i) {
        const long double a = -2.0L * 3.141592653589793238462643383279502884L * i / nf;
        h2[i] = mk<T>((T)cosl(a), (T)sinl(a));
      }
      SI_TRY(tw2.reserve((size_t)nf * sizeof(C)));
      SI_HIP(hipMemcpy(tw2.p, h2.data(), (size_t)nf * sizeof(C), hipMemcpyHostToDevice));
      fc2 = fc;
      fc2.n_fft = nf;
      fc2.tw = tw2.as<C>();
      fc2.inplace = use_inplace ? 1 : 0;
      fc2.maxb = use_inplace ? ip_maxb : 0;
      SI_TRY(big_y.reserve((size_t)B() * Tn() * n * sizeof(C)));
      const void* fns[] = {(const void*)k_big_rows<T, false, false>, (const void*)k_big_rows<T, false, true>,
                           (const void*)k_big_rows<T, true, false>, (const void*)k_big_rows<T, true, true>};
      for (const void* fn : fns) SI_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max<size_t>(lds_bytes, 64 * 1024)));
    }
    SI_TRY(sums.reserve(16 * sizeof(double)));
    SI_TRY(fast.setup(cfg, h_window, length, pad));
    return SPECINV_OK;
  }

  // (a two-sided ADMM run asked to keep X and U takes the coverage kernels: the frame kernel carries Y = X + U alone there)
  // - decided when the method starts (init_common latches the flag): a toggle in the middle of a run must not send iterate(),
  // get_wave() and get_state_spec() to buffers the other path never reserved
  bool fast_path() const override {
    return fast.supported && !force_generic && !(fast.two && (method != Method::None ? keep_latched : keep_state));
  }
  int path_kind() const override { return fast_path() ? (fast.semi ? (fast.hopk ? 3 : 2) : 1) : 0; }
  void launch_geometry(int out[4]) const override {
    if (fast_path()) {
      fast.geometry(out);
    } else if (use_wave) {                     // a lane group of one wave per frame, every wave walking its share
      wave_iter_geometry<T>(N(), cfg.hop_length, Tn(), B(), cfg.onesided != 0, out);   // (out[1]: chunks per item where the overlap-add
                                                                                         //  runs in the kernel, else the frame count)
    } else {                                   // generic: one workgroup per frame pair
      out[0] = (use_dr ? dr_threads : frame_threads()) / 64;
      out[1] = (Tn() + 1) / 2;
      out[2] = out[0] * out[1] * B();
      out[3] = 0;
    }
  }

  // ------------------------------------------------------------------------------------
  // layout helpers: user (B, F, T) <-> internal (B, T, F)
  template <typename E>
  int transpose(const E* in, E* out, int R, int Cc) {
    dim3 grid((Cc + 31) / 32, (R + 31) / 32, B());
    SI_CHECK(grid.y <= 65535, SPECINV_EUNSUPPORTED, "dimension too large for transpose");
    hipLaunchKernelGGL((k_transpose<E>), grid, dim3(32, 8), 0, stream, in, out, R, Cc);
    SI_HIP(hipGetLastError());
    return SPECINV_OK;
  }

  // `fc` describes the two-buffer transform (what RTISI-LA and the L-BFGS chain run); k_stft / k_iter_pair / k_grad_frames take
  // the in-place form where the plan chose it
  FrameCfg<T> frame_cfg(int64_t len) const {
    FrameCfg<T> c = fc;
    c.length = len;
    c.inplace = use_inplace ? 1 : 0;
    c.maxb = use_inplace ? ip_maxb : 0;
    c.tw_lds = 0;
    return c;
  }

  // `len`: samples per row of `out` (default: the plan's length; another length with the same frame count is allowed
  // for the un-normalised form only, the envelope belongs to the plan's length)
  int launch_ola(const T* fr, T* out, bool use_env, int64_t len = -1, int items = -1) {   // items: rows of `fr` / `out` (default: the batch)
    if (len < 0) len = length;
    SI_CHECK(!use_env || len == length, SPECINV_EINVAL, "envelope division needs the plan's own signal length");
    const int64_t total = (int64_t)(items < 0 ? B() : items) * len;
    if constexpr (std::is_same<T, float>::value) {
      if (cfg.hop_length % 4 == 0 && N() % 4 == 0 && pad % 4 == 0 && len % 4 == 0) {
        hipLaunchKernelGGL(k_ola_f4, dim3((unsigned)ceil_div(total / 4, 256)), dim3(256), 0, stream, fr, env.as<float>(), out,
                           N(), cfg.hop_length, pad, Tn(), len, total / 4, use_env ? 1 : 0);
        SI_HIP(hipGetLastError());
        return SPECINV_OK;
      }
    }
    if constexpr (std::is_same<T, double>::value) {
      if (cfg.hop_length % 2 == 0 && N() % 2 == 0 && pad % 2 == 0 && len % 2 == 0) {
        hipLaunchKernelGGL(k_ola_d2, dim3((unsigned)ceil_div(total / 2, 256)), dim3(256), 0, stream, fr, env.as<double>(), out,
                           N(), cfg.hop_length, pad, Tn(), len, total / 2, use_env ? 1 : 0);
        SI_HIP(hipGetLastError());
        return SPECINV_OK;
      }
    }
    hipLaunchKernelGGL((k_ola<T>), dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, stream, fr, env.as<T>(), out,
                       N(), cfg.hop_length, pad, Tn(), len, total, use_env ? 1 : 0);
    SI_HIP(hipGetLastError());
    return SPECINV_OK;
  }

  // gradient of a centre-padded analysis w.r.t. the signal: overlap-add over the signal's own positions plus the
  // fold of the padded margins (reflect / replicate / circular copies)
  int launch_grad_fold(const T* fr, T* grad, int64_t len, const T* margins = nullptr) {
    if (margins == nullptr) SI_TRY(launch_ola(fr, grad, false, len));
    if (pad > 0 && cfg.pad_mode != SPECINV_PAD_CONSTANT) {
      const int64_t margin = (int64_t)B() * 2 * (pad + 1);
      hipLaunchKernelGGL((k_grad_fold_margins<T>), dim3((unsigned)ceil_div(margin, 256)), dim3(256), 0, stream, fr, grad,
                         N(), cfg.hop_length, pad, cfg.pad_mode, Tn(), len, (int64_t)B(), margins);
      SI_HIP(hipGetLastError());
    }
    return SPECINV_OK;
  }

  // gradient w.r.t. the signal from the gradient w.r.t. its spectrogram (internal layout, Hermitian weights applied):
  // inverse frames with `scale`, overlap-add over the padded signal, fold of the margins.  Big float32 batches on the
  // wave-level FFT do all of it on the chip (k_hop_inverse), everything else goes through the frames buffer.
  int grad_from_spec(const C* spec_btf, T* grad, T scale, int64_t len) {
    if constexpr (std::is_same<T, float>::value) {
      bool used = false;
      float* margins = nullptr;
      SI_TRY(fast.launch_inverse_ola(*this, reinterpret_cast<const fast::v2f*>(spec_btf), grad, (long long)len, scale, &margins,
                                     &used));
      if (used) return launch_grad_fold(nullptr, grad, len, margins);
    }
    SI_TRY(frames_needed());
    SI_TRY(inverse_frames(spec_btf, frames.template as<T>(), scale, len));
    return launch_grad_fold(frames.template as<T>(), grad, len);
  }

  // threads per frame workgroup of the generic kernels: one per butterfly of the widest stage (n_fft / smallest radix),
  // whole waves, at most 256
  int frame_threads() const {
    if (use_inplace) return ip_threads;
    int rmin = 8;
    for (int i = 0; i < fc.n_stages; ++i) rmin = std::min(rmin, fc.radix[i]);
    const int want = (nf / std::max(2, rmin) + 63) / 64 * 64;
    return std::min(256, std::max(64, want));
  }

  int frames_needed() { return frames.reserve((size_t)B() * Tn() * N() * sizeof(T)); }

  // internal-layout spectrum -> x
  int istft_internal(const C* spec_btf, T* out) {
    SI_TRY(frames_needed());
    SI_TRY(inverse_frames(spec_btf, frames.as<T>(), fc.inv_scale, length));
    return launch_ola(frames.as<T>(), out, true);
  }

  // ---- frames in four steps (kernels_big.h) ---------------------------------------------------------------------------
  BigCfg<T> big_cfg(int64_t len) const {
    BigCfg<T> c{};
    c.f = fc;
    c.f.length = len;
    c.s = fc2;
    c.n1 = big_n1;
    c.n2 = nf;
    c.n_frames = Tn();
    c.y = big_y.as<C>();
    return c;
  }
  int big_rows(const BigCfg<T>& c, bool inverse) {
    const dim3 grid(big_n1, Tn(), B()), blk(frame_threads());
    if (use_inplace) {
      if (inverse) hipLaunchKernelGGL((k_big_rows<T, true, true>), grid, blk, lds_bytes, stream, c);
      else hipLaunchKernelGGL((k_big_rows<T, true, false>), grid, blk, lds_bytes, stream, c);
    } else {
      if (inverse) hipLaunchKernelGGL((k_big_rows<T, false, true>), grid, blk, lds_bytes, stream, c);
      else hipLaunchKernelGGL((k_big_rows<T, false, false>), grid, blk, lds_bytes, stream, c);
    }
    SI_HIP(hipGetLastError());
    return SPECINV_OK;
  }
  int big_forward(const BigCfg<T>& c, const T* xin) {            // x -> y: every frame's spectrum, bin k at big_pos(k)
    const dim3 grid((unsigned)ceil_div(nf, 256), Tn(), B());
    if (big_n1 == 2) hipLaunchKernelGGL((k_big_pass1<T, 2>), grid, dim3(256), 0, stream, c, xin);
    else if (big_n1 == 4) hipLaunchKernelGGL((k_big_pass1<T, 4>), grid, dim3(256), 0, stream, c, xin);
    else hipLaunchKernelGGL((k_big_pass1<T, 8>), grid, dim3(256), 0, stream, c, xin);
    SI_HIP(hipGetLastError());
    return big_rows(c, false);
  }
  int big_inverse(const BigCfg<T>& c, T* fr) {                   // y (what an inverse real transform sees) -> windowed frames
    SI_TRY(big_rows(c, true));
    const dim3 grid((unsigned)ceil_div(nf, 256), Tn(), B());
    if (big_n1 == 2) hipLaunchKernelGGL((k_big_pass4<T, 2>), grid, dim3(256), 0, stream, c, fr);
    else if (big_n1 == 4) hipLaunchKernelGGL((k_big_pass4<T, 4>), grid, dim3(256), 0, stream, c, fr);
    else hipLaunchKernelGGL((k_big_pass4<T, 8>), grid, dim3(256), 0, stream, c, fr);
    SI_HIP(hipGetLastError());
    return SPECINV_OK;
  }

  // pad_mode_override >= 0 / scale_override > 0 replace the plan's pad mode / forward scale (used by the adjoints)
  int stft_internal(const T* xin, int64_t len, C* spec_btf, int pad_mode_override = -1, T scale_override = T(0)) {
    const int pm = pad_mode_override >= 0 ? pad_mode_override : cfg.pad_mode;
    const T sc = scale_override > T(0) ? scale_override : fc.fwd_scale;
    const int64_t tcheck = 1 + (len + 2 * pad - N()) / cfg.hop_length;
    SI_CHECK(len + 2 * pad >= N() && tcheck == Tn(), SPECINV_EINVAL,
             "signal length %lld gives %lld frames, plan has %d", (long long)len, (long long)tcheck, Tn());
    if (cfg.center && pm == SPECINV_PAD_REFLECT)
      SI_CHECK(pad < len, SPECINV_EINVAL, "reflect padding needs n_fft/2 < length");
    if constexpr (std::is_same<T, float>::value) {
      if (fast.xform_ok && !force_generic)
        return fast.launch_xform(*this, true, xin, (long long)len, reinterpret_cast<fast::v2f*>(spec_btf), nullptr, sc,
                                 pm);
    }
    if (big) {
      BigCfg<T> bc = big_cfg(len);
      bc.f.pad_mode = pm;
      SI_TRY(big_forward(bc, xin));
      hipLaunchKernelGGL((k_big_split_out<T>), dim3((unsigned)ceil_div(n_freq, 256), Tn(), B()), dim3(256), 0, stream, bc, spec_btf, sc);
      SI_HIP(hipGetLastError());
      return SPECINV_OK;
    }
    FrameCfg<T> c = frame_cfg(len);
    c.pad_mode = pm;
    c.fwd_scale = sc;
    if (use_inplace) hipLaunchKernelGGL((k_stft<T, true>), dim3(Tn(), B()), dim3(frame_threads()), lds_bytes, stream, c, xin, spec_btf);
    else hipLaunchKernelGGL((k_stft<T, false>), dim3(Tn(), B()), dim3(frame_threads()), lds_bytes, stream, c, xin, spec_btf);
    SI_HIP(hipGetLastError());
    return SPECINV_OK;
  }

  // frames[b,t,:] = window * scale * (Hermitian inverse DFT sum of spec[b,t,:]); scale = 1/N gives _istft's frames
  int inverse_frames(const C* spec_btf, T* fr, T scale, int64_t len) {
    if constexpr (std::is_same<T, float>::value) {
      if (fast.xform_ok && !force_generic)
        return fast.launch_xform(*this, false, nullptr, (long long)len,
                                 const_cast<fast::v2f*>(reinterpret_cast<const fast::v2f*>(spec_btf)), fr, scale);
    }
    if (big) {
      BigCfg<T> bc = big_cfg(len);
      bc.f.inv_scale = scale;
      hipLaunchKernelGGL((k_big_merge_in<T>), dim3((unsigned)ceil_div(N() / 2 + 1, 256), Tn(), B()), dim3(256), 0, stream, bc, spec_btf);
      SI_HIP(hipGetLastError());
      return big_inverse(bc, fr);
    }
    FrameCfg<T> c = frame_cfg(len);
    c.inv_scale = scale;
    if (use_inplace) hipLaunchKernelGGL((k_grad_frames<T, true>), dim3(Tn(), B()), dim3(frame_threads()), lds_bytes, stream, c, spec_btf, fr);
    else hipLaunchKernelGGL((k_grad_frames<T, false>), dim3(Tn(), B()), dim3(frame_threads()), lds_bytes, stream, c, spec_btf, fr);
    SI_HIP(hipGetLastError());
    return SPECINV_OK;
  }

  int stft(const void* xin, int64_t len, void* spec_out) override {
    SI_CHECK(xin && spec_out, SPECINV_EINVAL, "null pointer");
    SI_TRY(tmp_spec.reserve(nspec() * sizeof(C)));
    SI_TRY(stft_internal(static_cast<const T*>(xin), len, tmp_spec.as<C>()));
    return transpose<C>(tmp_spec.as<C>(), static_cast<C*>(spec_out), Tn(), n_freq);
  }

  int istft(const void* spec, void* x_out) override {
    SI_CHECK(spec && x_out, SPECINV_EINVAL, "null pointer");
    SI_TRY(tmp_spec.reserve(nspec() * sizeof(C)));
    SI_TRY(transpose<C>(static_cast<const C*>(spec), tmp_spec.as<C>(), n_freq, Tn()));
    return istft_internal(tmp_spec.as<C>(), static_cast<T*>(x_out));
  }

  int envelope(void* env_out) override {
    SI_CHECK(env_out, SPECINV_EINVAL, "null pointer");
    SI_HIP(hipMemcpyAsync(env_out, env.p, length * sizeof(T), hipMemcpyDeviceToDevice, stream));
    return SPECINV_OK;
  }

  int phase_init(const void* magp, void* spec_out) override {
    SI_CHECK(magp && spec_out, SPECINV_EINVAL, "null pointer");
    const int rows = B() * n_freq;
    hipLaunchKernelGGL((k_phase_init<T>), dim3((rows + 3) / 4), dim3(256), 0, stream, static_cast<const T*>(magp),
                       static_cast<C*>(spec_out), B(), n_freq, Tn(), N(), cfg.hop_length);
    SI_HIP(hipGetLastError());
    return SPECINV_OK;
  }

  int reduce3(const T* a, const T* b, int64_t n, double out3[3]) {
    const int nb = (int)std::min<int64_t>(1024, std::max<int64_t>(1, ceil_div(n, 256 * 8)));
    SI_TRY(partials.reserve(std::max<size_t>((size_t)nb * 3, (size_t)B() * Tn() * 2) * sizeof(double)));
    hipLaunchKernelGGL((k_metric_partials<T>), dim3(nb), dim3(256), 0, stream, a, b, n, partials.as<double>());
    SI_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_finish_partials, dim3(1), dim3(256), 0, stream, partials.as<double>(), (int64_t)nb, 3,
                       sums.as<double>());
    SI_HIP(hipGetLastError());
    SI_HIP(hipMemcpyAsync(out3, sums.p, 3 * sizeof(double), hipMemcpyDeviceToHost, stream));
    SI_HIP(hipStreamSynchronize(stream));
    return SPECINV_OK;
  }

  int metric_sums(const void* a, const void* b, int64_t n, double s[4]) override {
    SI_CHECK(a && b && n > 0, SPECINV_EINVAL, "bad arguments");
    double r[3];
    SI_TRY(reduce3(static_cast<const T*>(a), static_cast<const T*>(b), n, r));
    s[0] = r[0];
    s[1] = r[1];
    s[2] = r[2];
    s[3] = (double)n;
    return SPECINV_OK;
  }

  // ------------------------------------------------------------------------------------
  // shared part of gla_init / admm_init: target + starting spectrum into internal layout,
  // x = ISTFT(start) (methods.py:233 / :453)
  int init_common(const void* init_spec, const void* magp, int fast_mode) {
    SI_CHECK(init_spec || magp, SPECINV_EINVAL, "need init_spec and/or mag");
    // torch.stft refuses reflect padding that is not smaller than the signal (methods.py:241 would raise)
    SI_CHECK(!(cfg.center && cfg.pad_mode == SPECINV_PAD_REFLECT && pad >= length), SPECINV_EINVAL,
             "reflect padding (%d) must be smaller than the signal length (%lld): too few frames", pad, (long long)length);
    const int64_t ns = nspec();
    count = (double)ns;
    const C* start_user = static_cast<const C*>(init_spec);
    keep_latched = keep_state;
    fast.keep_state = keep_state;
    fast.exact = exact;
    if constexpr (std::is_same<T, float>::value) {
      // fused kernels, magnitude input: phase_init writes the pair layout itself (methods.py:106 without the (B, F, T)
      // complex round trip and the two layout passes)
      if (!init_spec && fast_path() && !fast.semi && (fast.R == 8 || fast.R == 16) && !getenv("SPECINV_DISABLE_INIT_PAIRS"))
        return fast.begin(*this, fast_mode, nullptr, magp, &sum_m2);
    }
    if (!init_spec) {
      SI_TRY(tmp_spec.reserve(ns * sizeof(C)));
      SI_TRY(phase_init(magp, tmp_spec.p));                       // methods.py:106
      start_user = tmp_spec.as<C>();
    }
    if (fast_path()) {
      // fused path: user layout -> pair layout directly, x0 from its own ISTFT kernel
      const T* mag_user = static_cast<const T*>(magp);
      if (!magp) {                                                  // methods.py:110
        SI_TRY(tmp_real.reserve(ns * sizeof(T)));
        hipLaunchKernelGGL((k_cabs<T>), dim3((unsigned)ceil_div(ns, 256)), dim3(256), 0, stream, start_user,
                           tmp_real.as<T>(), ns);
        SI_HIP(hipGetLastError());
        mag_user = tmp_real.as<T>();
      }
      return fast.begin(*this, fast_mode, start_user, mag_user, &sum_m2);
    }
    SI_TRY(specA.reserve(ns * sizeof(C)));
    SI_TRY(mag.reserve(ns * sizeof(T)));
    SI_TRY(x.reserve((size_t)B() * length * sizeof(T)));
    SI_TRY(partials.reserve(std::max<size_t>((size_t)B() * Tn() * 2, 3 * 1024) * sizeof(double)));
    SI_TRY(transpose<C>(start_user, specA.as<C>(), n_freq, Tn()));
    if (magp) {
      SI_TRY(transpose<T>(static_cast<const T*>(magp), mag.as<T>(), n_freq, Tn()));
    } else {                                                        // methods.py:110
      hipLaunchKernelGGL((k_cabs<T>), dim3((unsigned)ceil_div(ns, 256)), dim3(256), 0, stream, specA.as<C>(),
                         mag.as<T>(), ns);
      SI_HIP(hipGetLastError());
    }
    double r[3];
    SI_TRY(reduce3(mag.as<T>(), nullptr, ns, r));
    sum_m2 = r[1];
    return istft_internal(specA.as<C>(), x.as<T>());
  }

  int gla_init(const void* init_spec, const void* magp, double alpha) override {
    SI_CHECK(alpha >= 0, SPECINV_EINVAL, "alpha must be >= 0");
    method = Method::None;
    coef = (T)(alpha / (1.0 + alpha));                              // methods.py:235
    SI_TRY(init_common(init_spec, magp, fast::MODE_GLA));
    method = Method::Gla;
    return SPECINV_OK;
  }

  int admm_init(const void* init_spec, const void* magp, double rho) override {
    method = Method::None;
    coef = (T)rho;
    SI_TRY(init_common(init_spec, magp, fast::MODE_ADMM));
    if (!fast_path()) {
      SI_TRY(specB.reserve(nspec() * sizeof(C)));
      SI_HIP(hipMemsetAsync(specB.p, 0, nspec() * sizeof(C), stream));  // U = 0, methods.py:456
    }
    method = Method::Admm;
    return SPECINV_OK;
  }

  int iterate(int n_iter, bool eval_last, double s[4]) override {
    SI_CHECK(method != Method::None, SPECINV_ESTATE, "iterate called before gla_init/admm_init");
    SI_CHECK(n_iter >= 0, SPECINV_EINVAL, "n_iter < 0");
    if (n_iter == 0) return SPECINV_OK;
    if (fast_path()) {
      fast.keep_state = fast.two ? keep_latched : keep_state;
      SI_TRY(fast.iterate(*this, n_iter, eval_last));
    } else {
      if (!(use_wave && wave_iter_ola_chunks<T>(N(), cfg.hop_length, Tn(), B(), cfg.onesided != 0) > 0)) SI_TRY(frames_needed());
      const T inv1p = T(1) / (T)(1.0 + (double)coef);
      const FrameCfg<T> fci = frame_cfg(length);
      for (int i = 0; i < n_iter; ++i) {
        const bool ev = eval_last && i == n_iter - 1;
        if (big) {                                   // the same iteration in four steps through device memory (kernels_big.h)
          const BigCfg<T> bc = big_cfg(length);
          SI_TRY(big_forward(bc, x.as<T>()));
          const dim3 ug((unsigned)ceil_div(N() / 2 + 1, 256), Tn(), B());
          const int mode = method == Method::Gla ? 0 : 1;
          C* sb = mode == 0 ? (C*)nullptr : specB.as<C>();
          if (ev) SI_TRY(partials.reserve((size_t)2 * ug.x * ug.y * ug.z * sizeof(double)));
          if (mode == 0) {
            if (ev) hipLaunchKernelGGL((k_big_update<T, 0, true>), ug, dim3(256), 0, stream, bc, specA.as<C>(), sb, mag.as<T>(), coef, inv1p, partials.as<double>());
            else hipLaunchKernelGGL((k_big_update<T, 0, false>), ug, dim3(256), 0, stream, bc, specA.as<C>(), sb, mag.as<T>(), coef, inv1p, partials.as<double>());
          } else {
            if (ev) hipLaunchKernelGGL((k_big_update<T, 1, true>), ug, dim3(256), 0, stream, bc, specA.as<C>(), sb, mag.as<T>(), coef, inv1p, partials.as<double>());
            else hipLaunchKernelGGL((k_big_update<T, 1, false>), ug, dim3(256), 0, stream, bc, specA.as<C>(), sb, mag.as<T>(), coef, inv1p, partials.as<double>());
          }
          SI_HIP(hipGetLastError());
          SI_TRY(big_inverse(bc, frames.as<T>()));
          SI_TRY(launch_ola(frames.as<T>(), x.as<T>(), true));
          continue;
        }
        if (use_wave) {
          WaveIterArgs<T> wa{};
          wa.c = fci;
          wa.x = x.as<T>();
          wa.S0 = specA.as<C>();
          wa.S1 = method == Method::Gla ? (C*)nullptr : specB.as<C>();
          wa.mag = mag.as<T>();
          wa.coef = coef;
          wa.inv1p = inv1p;
          wa.batch = B();
          wa.mode = method == Method::Gla ? 0 : 1;
          wa.eval = ev ? 1 : 0;
          // hop = n_fft / 2, / 4, / 8: the overlap-add in the kernel's registers - no frames buffer, the new signal written to the
          // plan's other signal buffer (the frames of an iteration read the old one), the chunk boundaries finished by k_wave_seams
          int ov = 0;
          const int nch = wave_iter_ola_chunks<T>(N(), cfg.hop_length, Tn(), B(), cfg.onesided != 0, &ov);
          if (nch > 0) {
            const size_t keep = ov == 1 ? (size_t)(N() - cfg.hop_length) : (size_t)(ov - 1) * cfg.hop_length;
            const size_t seam_bytes = (size_t)B() * nch * keep * sizeof(T);
            SI_TRY(x_alt.reserve((size_t)B() * length * sizeof(T)));
            SI_TRY(seam_l.reserve(seam_bytes));
            SI_TRY(seam_r.reserve(seam_bytes));
            wa.x_out = x_alt.as<T>();
            wa.env = env.as<T>();
            wa.seamL = seam_l.as<T>();
            wa.seamR = seam_r.as<T>();
            wa.nch = nch;
            wa.ov = ov;
          } else {
            wa.frames = frames.as<T>();
          }
          // (an evaluating launch leaves a pair of partial sums per wave: room for the fullest launch)
          if (ev) SI_TRY(partials.reserve((size_t)2 * std::max(wave_iter_waves<T>(N(), (int64_t)B() * Tn()), 16 * 1024) * sizeof(double)));
          wa.partials = partials.as<double>();
          SI_TRY(wave_iter_launch<T>(wa, stream, &wave_last_waves));
          if (nch > 0) {
            std::swap(x.p, x_alt.p);
            std::swap(x.bytes, x_alt.bytes);
          } else {
            SI_TRY(launch_ola(frames.as<T>(), x.as<T>(), true));
          }
          continue;
        }
        const dim3 grid((Tn() + 1) / 2, B()), blk(use_dr ? dr_threads : frame_threads());   // two frames per complex FFT
        static const bool ip_small = !(getenv("SPECINV_GENERIC_IP256") && getenv("SPECINV_GENERIC_IP256")[0] == '0');
        {
          const void* fn = nullptr;
          const int mode = method == Method::Gla ? 0 : 1;
          if (use_dr) fn = mode == 0 ? (ev ? (const void*)k_iter_pair_dr<T, 0, true> : (const void*)k_iter_pair_dr<T, 0, false>)
                                     : (ev ? (const void*)k_iter_pair_dr<T, 1, true> : (const void*)k_iter_pair_dr<T, 1, false>);
          else if (use_inplace && blk.x <= 256 && ip_small)
            fn = mode == 0 ? (ev ? (const void*)k_iter_pair<T, 0, true, true, 256> : (const void*)k_iter_pair<T, 0, false, true, 256>)
                           : (ev ? (const void*)k_iter_pair<T, 1, true, true, 256> : (const void*)k_iter_pair<T, 1, false, true, 256>);
          else if (use_inplace) fn = mode == 0 ? (ev ? (const void*)k_iter_pair<T, 0, true, true> : (const void*)k_iter_pair<T, 0, false, true>)
                                          : (ev ? (const void*)k_iter_pair<T, 1, true, true> : (const void*)k_iter_pair<T, 1, false, true>);
          else fn = mode == 0 ? (ev ? (const void*)k_iter_pair<T, 0, true, false> : (const void*)k_iter_pair<T, 0, false, false>)
                              : (ev ? (const void*)k_iter_pair<T, 1, true, false> : (const void*)k_iter_pair<T, 1, false, false>);
          FrameCfg<T> ca = fci;
          ca.tw_lds = (!use_dr && tw_lds) ? 1 : 0;
          const T* xa = x.as<T>();
          C* sa = specA.as<C>();
          C* sb = mode == 0 ? (C*)nullptr : specB.as<C>();
          const T* ma = mag.as<T>();
          T cf = coef, ip = inv1p;
          T* fra = frames.as<T>();
          double* pa = partials.as<double>();
          void* kargs[] = {&ca, &xa, &sa, &sb, &ma, &cf, &ip, &fra, &pa};
          SI_HIP(hipLaunchKernel(fn, grid, blk, kargs, use_dr ? dr_lds : lds_bytes + (tw_lds ? (size_t)N() * sizeof(C) : 0), stream));
        }
        SI_HIP(hipGetLastError());
        SI_TRY(launch_ola(frames.as<T>(), x.as<T>(), true));
      }
    }
    if (eval_last) {
      const int64_t n_part = fast_path() ? (int64_t)fast.n_partials
                             : big ? (int64_t)B() * Tn() * ceil_div(N() / 2 + 1, 256)
                             : use_wave ? (int64_t)wave_last_waves : (int64_t)B() * ((Tn() + 1) / 2);
      if (eval_dev_out != nullptr) {
        hipLaunchKernelGGL(k_finish_partials, dim3(1), dim3(256), 0, stream, partials.as<double>(), n_part, 2, eval_dev_out);
        hipLaunchKernelGGL(k_store2, dim3(1), dim3(1), 0, stream, eval_dev_out + 2, sum_m2, count);
        SI_HIP(hipGetLastError());
        return SPECINV_OK;
      }
      if (deferred_slot >= 0) {
        // deferred evaluation (run_loop with tol == 0 and no callback): keep the sums on the device
        SI_TRY(eval_log.reserve((size_t)(deferred_slot + 1) * 2 * sizeof(double)));
        hipLaunchKernelGGL(k_finish_partials, dim3(1), dim3(256), 0, stream, partials.as<double>(), n_part, 2,
                           eval_log.as<double>() + 2 * deferred_slot);
        SI_HIP(hipGetLastError());
        return SPECINV_OK;
      }
      SI_CHECK(s != nullptr, SPECINV_EINVAL, "sums_host is NULL");
      hipLaunchKernelGGL(k_finish_partials, dim3(1), dim3(256), 0, stream, partials.as<double>(), n_part, 2,
                         sums.as<double>());
      SI_HIP(hipGetLastError());
      double r[2];
      SI_HIP(hipMemcpyAsync(r, sums.p, 2 * sizeof(double), hipMemcpyDeviceToHost, stream));
      SI_HIP(hipStreamSynchronize(stream));
      s[0] = r[0];
      s[1] = r[1];
      s[2] = sum_m2;
      s[3] = count;
    }
    return SPECINV_OK;
  }

  int begin_deferred(int n_slots) override {
    SI_TRY(eval_log.reserve((size_t)std::max(1, n_slots) * 2 * sizeof(double)));
    return SPECINV_OK;
  }

  int read_deferred(int n_slots, double* out /* n_slots x 4 */) override {
    std::vector<double> h((size_t)n_slots * 2);
    if (n_slots > 0) {
      SI_HIP(hipMemcpyAsync(h.data(), eval_log.p, h.size() * sizeof(double), hipMemcpyDeviceToHost, stream));
    }
    SI_HIP(hipStreamSynchronize(stream));
    for (int i = 0; i < n_slots; ++i) {
      out[4 * i] = h[2 * i];
      out[4 * i + 1] = h[2 * i + 1];
      out[4 * i + 2] = sum_m2;
      out[4 * i + 3] = count;
    }
    return SPECINV_OK;
  }

  int get_wave(void* x_out) override {
    SI_CHECK(method != Method::None, SPECINV_ESTATE, "no running state");
    SI_CHECK(x_out, SPECINV_EINVAL, "null pointer");
    if (fast_path()) return fast.get_wave(*this, static_cast<T*>(x_out));
    SI_HIP(hipMemcpyAsync(x_out, x.p, (size_t)B() * length * sizeof(T), hipMemcpyDeviceToDevice, stream));
    return SPECINV_OK;
  }

  int get_state_spec(int which, void* spec_out) override {
    SI_CHECK(method != Method::None, SPECINV_ESTATE, "no running state");
    SI_CHECK(spec_out && (which == 0 || ((which == 1 || which == 2) && method == Method::Admm)), SPECINV_EINVAL, "bad arguments");
    if (fast_path()) return fast.get_state_spec(*this, which, static_cast<C*>(spec_out));
    if (which == 2) {                                   // Y = X + U (what the next iteration reads, methods.py:467-468)
      SI_TRY(tmp_spec.reserve(nspec() * sizeof(C)));
      SI_HIP(hipMemcpyAsync(tmp_spec.p, specA.p, nspec() * sizeof(C), hipMemcpyDeviceToDevice, stream));
      hipLaunchKernelGGL((k_axpy<T>), dim3((unsigned)ceil_div(2 * nspec(), 256)), dim3(256), 0, stream, (T)1,
                         reinterpret_cast<const T*>(specB.as<C>()), reinterpret_cast<T*>(tmp_spec.as<C>()), 2 * nspec());
      SI_HIP(hipGetLastError());
      return transpose<C>(tmp_spec.as<C>(), static_cast<C*>(spec_out), Tn(), n_freq);
    }
    return transpose<C>(which == 0 ? specA.as<C>() : specB.as<C>(), static_cast<C*>(spec_out), Tn(), n_freq);
  }

  // ------------------------------------------------------------------------------------
  // ------------------------------------------------------------------------------------
  // differentiation building blocks (kernels_adjoint.h)
  int gla_update(const void* R, const void* P, const void* m, double lr, void* S_out, void* Q_out) override {
    SI_CHECK(R && P && m && S_out && Q_out, SPECINV_EINVAL, "null pointer");
    const int64_t n = nspec();
    hipLaunchKernelGGL((k_gla_update<T>), dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, stream, static_cast<const C*>(R),
                       static_cast<const C*>(P), static_cast<const T*>(m), (T)lr, static_cast<C*>(S_out),
                       static_cast<C*>(Q_out), n);
    SI_HIP(hipGetLastError());
    return SPECINV_OK;
  }

  int gla_update_adjoint(const void* gQ, const void* gPn, const void* S, const void* m, double lr, void* gR, void* gP,
                         void* gmag) override {
    SI_CHECK(gQ && S && m && gR && gP && gmag, SPECINV_EINVAL, "null pointer");
    const int64_t n = nspec();
    hipLaunchKernelGGL((k_gla_update_adjoint<T>), dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, stream,
                       static_cast<const C*>(gQ), static_cast<const C*>(gPn), static_cast<const C*>(S),
                       static_cast<const T*>(m), (T)lr, static_cast<C*>(gR), static_cast<C*>(gP), static_cast<T*>(gmag), n);
    SI_HIP(hipGetLastError());
    return SPECINV_OK;
  }

  int admm_update(const void* R, const void* X, const void* U, const void* m, double rho, void* Xn, void* Un, void* V,
                  void* Yn) override {
    SI_CHECK(R && X && U && m && Xn && Un && V && Yn, SPECINV_EINVAL, "null pointer");
    const int64_t n = nspec();
    const T r = (T)rho, inv1p = T(1) / (T)(1.0 + (double)r);
    hipLaunchKernelGGL((k_admm_update<T>), dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, stream, static_cast<const C*>(R),
                       static_cast<const C*>(X), static_cast<const C*>(U), static_cast<const T*>(m), r, inv1p,
                       static_cast<C*>(Xn), static_cast<C*>(Un), static_cast<C*>(V), static_cast<C*>(Yn), n);
    SI_HIP(hipGetLastError());
    return SPECINV_OK;
  }

  int admm_update_adjoint(const void* gYn, const void* gXn, const void* gUn, const void* V, const void* m, double rho,
                          void* gR, void* gX, void* gU, void* gmag) override {
    SI_CHECK(gYn && V && m && gR && gX && gU && gmag, SPECINV_EINVAL, "null pointer");
    const int64_t n = nspec();
    const T r = (T)rho, inv1p = T(1) / (T)(1.0 + (double)r);
    hipLaunchKernelGGL((k_admm_update_adjoint<T>), dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, stream,
                       static_cast<const C*>(gYn), static_cast<const C*>(gXn), static_cast<const C*>(gUn),
                       static_cast<const C*>(V), static_cast<const T*>(m), r, inv1p, static_cast<C*>(gR),
                       static_cast<C*>(gX), static_cast<C*>(gU), static_cast<T*>(gmag), n);
    SI_HIP(hipGetLastError());
    return SPECINV_OK;
  }

  // x = overlap_add(w * inv_scale * IDFT_H(Q)) / env   =>   gQ = scale_k * DFT(w * zero-padded frames of g/env)
  int istft_adjoint(const void* g_x, void* g_spec_out) override {
    SI_CHECK(g_x && g_spec_out, SPECINV_EINVAL, "null pointer");
    const int64_t total = (int64_t)B() * length;
    SI_TRY(tmp_real.reserve(total * sizeof(T)));
    hipLaunchKernelGGL((k_div_env<T>), dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, stream,
                       static_cast<const T*>(g_x), env.as<T>(), tmp_real.as<T>(), length, total);
    SI_HIP(hipGetLastError());
    SI_TRY(tmp_spec.reserve(nspec() * sizeof(C)));
    SI_TRY(stft_internal(tmp_real.as<T>(), length, tmp_spec.as<C>(), SPECINV_PAD_CONSTANT, T(1)));
    const int64_t ns = nspec();
    hipLaunchKernelGGL((k_istft_adjoint_scale<T>), dim3((unsigned)ceil_div(ns, 256)), dim3(256), 0, stream,
                       tmp_spec.as<C>(), n_freq, N(), cfg.onesided, fc.inv_scale, ns);
    SI_HIP(hipGetLastError());
    return transpose<C>(tmp_spec.as<C>(), static_cast<C*>(g_spec_out), Tn(), n_freq);
  }

  // R = fwd_scale * DFT(w * padded frames of x)   =>   gx = fold(overlap_add(w * fwd_scale * sum_k gR_k e^{+})))
  int stft_adjoint(const void* g_spec, int64_t len, void* g_x_out) override {
    SI_CHECK(g_spec && g_x_out, SPECINV_EINVAL, "null pointer");
    const int64_t tcheck = 1 + (len + 2 * pad - N()) / cfg.hop_length;
    SI_CHECK(len + 2 * pad >= N() && tcheck == Tn(), SPECINV_EINVAL, "signal length %lld does not match the plan",
             (long long)len);
    const int64_t ns = nspec();
    SI_TRY(tmp_spec.reserve(ns * sizeof(C)));
    SI_TRY(transpose<C>(static_cast<const C*>(g_spec), tmp_spec.as<C>(), n_freq, Tn()));
    if (cfg.onesided) {
      hipLaunchKernelGGL((k_halve_interior<T>), dim3((unsigned)ceil_div(ns, 256)), dim3(256), 0, stream, tmp_spec.as<C>(),
                         n_freq, N(), ns);
      SI_HIP(hipGetLastError());
    }
    return grad_from_spec(tmp_spec.as<C>(), static_cast<T*>(g_x_out), fc.fwd_scale, len);
  }

  int phase_init_adjoint(const void* magp, const void* g_spec, void* gmag) override {
    SI_CHECK(magp && g_spec && gmag, SPECINV_EINVAL, "null pointer");
    const int64_t ns = nspec();
    SI_TRY(tmp_real.reserve(ns * sizeof(T)));
    const int rows = B() * n_freq;
    hipLaunchKernelGGL((k_phase_init_adjoint_rows<T>), dim3((rows + 3) / 4), dim3(256), 0, stream,
                       static_cast<const T*>(magp), static_cast<const C*>(g_spec), static_cast<T*>(gmag),
                       tmp_real.as<T>(), B(), n_freq, Tn(), N(), cfg.hop_length);
    SI_HIP(hipGetLastError());
    hipLaunchKernelGGL((k_phase_init_adjoint_peaks<T>), dim3((unsigned)ceil_div(ns, 256)), dim3(256), 0, stream,
                       static_cast<const T*>(magp), tmp_real.as<T>(), static_cast<T*>(gmag), B(), n_freq, Tn(), N(),
                       cfg.hop_length);
    SI_HIP(hipGetLastError());
    return SPECINV_OK;
  }

  // RTISI-LA stages its target and its committed frames in the buffers that hold the target / frame scratch of a
  // running griffin_lim / ADMM state: such a state ends here (a later iterate() fails with SPECINV_ESTATE instead of
  // iterating against the wrong target)
  void rtisi_takes_buffers() { method = Method::None; }

  int rtisi_run(const void* magp, int look_ahead, int asym, int max_iter, double alpha, void* x_out) override {
    rtisi_takes_buffers();
    return rtisi_launch(*this, static_cast<const T*>(magp), look_ahead, asym, max_iter, alpha, static_cast<T*>(x_out));
  }

  int rtisi_record_elems(int look_ahead, int max_iter, int64_t* out) override {
    SI_CHECK(out && max_iter > 0, SPECINV_EINVAL, "bad arguments");
    *out = specinv::rtisi_record_elems<T>(B(), Tn(), n_freq, N(), cfg.hop_length, look_ahead, max_iter);
    return SPECINV_OK;
  }
  int rtisi_run_recorded(const void* magp, int look_ahead, int asym, int max_iter, double alpha, void* x_out,
                         void* rec_out) override {
    rtisi_takes_buffers();
    return rtisi_launch_recorded(*this, static_cast<const T*>(magp), look_ahead, asym, max_iter, alpha,
                                 static_cast<T*>(x_out), static_cast<C*>(rec_out));
  }
  int rtisi_adjoint(const void* magp, const void* rec, const void* g_x, int look_ahead, int asym, int max_iter, double alpha,
                    void* gmag_out) override {
    rtisi_takes_buffers();
    return rtisi_adjoint_launch(*this, static_cast<const T*>(magp), static_cast<const C*>(rec), static_cast<const T*>(g_x),
                                look_ahead, asym, max_iter, alpha, static_cast<T*>(gmag_out));
  }

  int rtisi_stream_begin(int look_ahead, int asym, int max_iter, double alpha) override {
    rtisi_takes_buffers();
    return specinv::rtisi_stream_begin(*this, rstream, look_ahead, asym, max_iter, alpha);
  }
  int rtisi_stream_push(const void* magp, int k, void* x_out, int64_t out_stride, int64_t* n_out) override {
    rtisi_takes_buffers();
    return specinv::rtisi_stream_push(*this, rstream, static_cast<const T*>(magp), k, static_cast<T*>(x_out), out_stride, n_out);
  }
  int rtisi_stream_flush(void* x_out, int64_t out_stride, int64_t* n_out) override {
    return specinv::rtisi_stream_flush(*this, rstream, static_cast<T*>(x_out), out_stride, n_out);
  }

  // ------------------------------------------------------------------------------------
  int transform_setup(int kind, const void* mel_fb, int n_mels) override {
    return tf_setup(*this, kind, static_cast<const T*>(mel_fb), n_mels);
  }
  int transform_forward(const void* xin, int64_t len, void* v_out) override {
    return tf_forward(*this, static_cast<const T*>(xin), len, static_cast<T*>(v_out));
  }
  int transform_loss_grad(const void* xin, int64_t len, const void* target, double* loss, void* grad, double* loss_dev,
                          bool with_stats, const void* stat_d) override {
    return tf_loss_grad(*this, static_cast<const T*>(xin), len, static_cast<const T*>(target), loss,
                        static_cast<T*>(grad), loss_dev, with_stats, static_cast<const T*>(stat_d));
  }
  int vec_dot(const void* a, const void* b, int64_t n, double* out) override {
    return lb_dot(*this, static_cast<const T*>(a), static_cast<const T*>(b), n, out);
  }
  int vec_axpy(double alpha, const void* xin, void* y, int64_t n) override {
    return lb_axpy(*this, (T)alpha, static_cast<const T*>(xin), static_cast<T*>(y), n);
  }
  int vec_scale(double alpha, const void* xin, void* y, int64_t n) override {
    return lb_scale(*this, (T)alpha, static_cast<const T*>(xin), static_cast<T*>(y), n);
  }
  int vec_absmax_abssum(const void* xin, int64_t n, double out[2]) override {
    return lb_absmax_abssum(*this, static_cast<const T*>(xin), n, out);
  }
  int lbfgs_direction(const void* g, const void* const* s_list, const void* const* y_list, const double* rho, int m,
                      double h_diag, void* d_out, int64_t n) override {
    return lb_direction(*this, static_cast<const T*>(g), s_list, y_list, rho, m, h_diag, static_cast<T*>(d_out), n);
  }
  int vec_multi_dot(const void* g, const void* const* vecs, int k, int64_t n, double* out, double* out_dev) override {
    return lb_multi_dot(*this, static_cast<const T*>(g), vecs, k, n, out, out_dev);
  }
  int vec_lincomb_step(const void* const* vecs, const double* coef, int k, int64_t n, void* out, double t, void* xs) override {
    SI_CHECK(xs != nullptr, SPECINV_EINVAL, "null pointer");
    return lb_lincomb(*this, vecs, coef, k, n, static_cast<T*>(out), t, static_cast<T*>(xs));
  }
  int vec_lincomb(const void* const* vecs, const double* coef, int k, int64_t n, void* out) override {
    return lb_lincomb(*this, vecs, coef, k, n, static_cast<T*>(out));
  }
  int lbfgs_pair(const void* g, const void* gp, const void* d, double t, void* y, void* sv, int64_t n, double* out,
                 double* out_dev) override {
    return lb_pair(*this, static_cast<const T*>(g), static_cast<const T*>(gp), static_cast<const T*>(d), t, static_cast<T*>(y),
                   static_cast<T*>(sv), n, out, out_dev);
  }
  int lbfgs_stats(const void* g, const void* d, int64_t n, double* out, double* out_dev) override {
    return lb_stats(*this, static_cast<const T*>(g), static_cast<const T*>(d), n, out, out_dev);
  }
  int lbfgs_pair_stats(const void* g, const void* gp, const void* d, double t, void* y, void* sv, int64_t n,
                       double* out8_dev) override {
    return lb_pair_stats(*this, static_cast<const T*>(g), static_cast<const T*>(gp), static_cast<const T*>(d), t,
                         static_cast<T*>(y), static_cast<T*>(sv), n, out8_dev);
  }
  int read_doubles(const double* src_dev, int n, double* out_host) override {
    SI_CHECK(src_dev && out_host && n > 0, SPECINV_EINVAL, "bad arguments");
    SI_HIP(hipMemcpyAsync(out_host, src_dev, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, stream));
    SI_HIP(si_stream_wait_short(stream));
    return SPECINV_OK;
  }
  // pinned, device-mapped host memory for the scalar results of an optimiser iteration
  struct HostBoard {
    void* p = nullptr;
    HostBoard() = default;
    HostBoard(const HostBoard&) = delete;
    HostBoard& operator=(const HostBoard&) = delete;
    ~HostBoard() {
      if (p) (void)hipHostFree(p);
    }
  };
  std::vector<std::unique_ptr<HostBoard>> boards;
  int board_alloc(int n, double** host_out, double** dev_out) override {
    SI_CHECK(n > 0 && host_out && dev_out, SPECINV_EINVAL, "bad arguments");
    std::unique_ptr<HostBoard> b(new HostBoard());
    SI_HIP(hipHostMalloc(&b->p, (size_t)n * sizeof(double), hipHostMallocMapped));
    std::memset(b->p, 0, (size_t)n * sizeof(double));
    void* dp = nullptr;
    SI_HIP(hipHostGetDevicePointer(&dp, b->p, 0));
    *host_out = static_cast<double*>(b->p);
    *dev_out = static_cast<double*>(dp);
    boards.push_back(std::move(b));
    return SPECINV_OK;
  }
  int stream_wait() override {
    SI_HIP(si_stream_wait_short(stream));
    return SPECINV_OK;
  }

  // the device-resident optimiser (lbfgs_dev.h): float32 on the one-launch objective
  std::vector<std::unique_ptr<LbfgsDev<float>>> lbfgs_devs;
  std::vector<std::unique_ptr<FastBuf>> lbd_pool;     // parameter-sized vectors of optimisers that are gone, for the next one
  static constexpr size_t kLbdPoolKeep = 4 + 2 * 11;  // (what an optimiser at main.py:43's history_size = 10 needs)
  int lbfgs_dev_create(int64_t n, const specinv_lbfgs_opts* opts, int32_t* handle_out) override {
    SI_CHECK(opts && handle_out, SPECINV_EINVAL, "null pointer");
    if constexpr (std::is_same<T, float>::value) {
      SI_CHECK(tf_kind >= 0, SPECINV_ESTATE, "specinv_transform_setup has not been called");
      std::unique_ptr<LbfgsDev<float>> L(new LbfgsDev<float>());
      SI_TRY(lbd_create(*this, *L, n, *opts));
      size_t slot = 0;
      while (slot < lbfgs_devs.size() && lbfgs_devs[slot]) ++slot;
      if (slot == lbfgs_devs.size()) lbfgs_devs.emplace_back();
      lbfgs_devs[slot] = std::move(L);
      *handle_out = (int32_t)slot;
      return SPECINV_OK;
    } else {
      return fail(SPECINV_EUNSUPPORTED, "the device-resident optimiser is float32 only");
    }
  }
  int lbfgs_dev_step(int32_t handle, void* xs, int64_t len, const void* target, specinv_lbfgs_info* info) override {
    if constexpr (std::is_same<T, float>::value) {
      SI_CHECK(handle >= 0 && (size_t)handle < lbfgs_devs.size() && lbfgs_devs[handle], SPECINV_EINVAL, "bad optimiser handle");
      return lbd_step(*this, *lbfgs_devs[handle], static_cast<float*>(xs), len, static_cast<const float*>(target), info);
    } else {
      return fail(SPECINV_EUNSUPPORTED, "the device-resident optimiser is float32 only");
    }
  }
  int lbfgs_dev_destroy(int32_t handle) override {
    SI_CHECK(handle >= 0 && (size_t)handle < lbfgs_devs.size() && lbfgs_devs[handle], SPECINV_EINVAL, "bad optimiser handle");
    SI_HIP(hipStreamSynchronize(stream));
    // the optimiser's parameter-sized vectors go back to the pool for the next one - up to the two gradients, the direction and
    // a handful of curvature pairs (kLbdPoolKeep vectors): a finished optimiser with history 100 would otherwise leave ~200 of
    // them (6.9 GB at C5) on the device until the plan is dropped
    for (auto& b : lbfgs_devs[handle]->vecs)
      if (lbd_pool.size() < kLbdPoolKeep) lbd_pool.push_back(std::move(b));
    lbfgs_devs[handle].reset();
    return SPECINV_OK;
  }
};

}  // namespace specinv
