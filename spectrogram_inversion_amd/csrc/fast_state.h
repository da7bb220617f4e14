// Host side of the wave-level kernels: FastState<float> picks the kernel, sizes the launch and owns the device state.
#pragma once
#include <algorithm>
#include "fast_core.h"
#include "kernels_layout.h"

// kernel tables of the approximate-projection units (tu_approx_*.hip): nullptr where there is no such kernel
extern "C" {
__attribute__((visibility("hidden"))) const void* specinv_approx_fused_a(int R, int OV, int mode, int eval, int tuned4);
__attribute__((visibility("hidden"))) const void* specinv_approx_fused_b(int R, int OV, int mode, int eval, int tuned4);
__attribute__((visibility("hidden"))) const void* specinv_approx_fused_c(int R, int OV, int mode, int eval, int tuned4);
__attribute__((visibility("hidden"))) const void* specinv_approx_td(int R, int OV, int early, int eval, int tuned4);
__attribute__((visibility("hidden"))) const void* specinv_approx_frame(int family, int R, int a, int b);
__attribute__((visibility("hidden"))) int specinv_approx_units_built(void);   // 0: the library was built without them (tu_noapprox.hip)
}

namespace specinv {

inline const void* approx_fused(int R, int OV, int mode, int eval, int tuned4) {
  const void* fn = specinv_approx_fused_a(R, OV, mode, eval, tuned4);
  if (!fn) fn = specinv_approx_fused_b(R, OV, mode, eval, tuned4);
  if (!fn) fn = specinv_approx_fused_c(R, OV, mode, eval, tuned4);
  return fn;
}

// ---- host side ---------------------------------------------------------------------------------------
struct FastBuf {
  void* p = nullptr;
  size_t bytes = 0;
  ~FastBuf() {
    if (p) {
      (void)hipFree(p);
      account_bytes(-(int64_t)bytes);
    }
  }
  int reserve(size_t n) {
    if (p && n <= bytes) return SPECINV_OK;
    if (p) {
      (void)hipFree(p);
      account_bytes(-(int64_t)bytes);
    }
    p = nullptr;
    bytes = 0;
    hipError_t e = hipMalloc(&p, n ? n : 16);
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

template <typename T>
struct FastState {
  bool supported = false;
  bool semi = false;
  bool hopk = false;
  bool xform_ok = false;
  bool two = false;
  bool keep_state = false;
  bool exact = true;
  int n_partials = 0;
  int setup(const specinv_stft_cfg&, const std::vector<T>&, int64_t, int) { return SPECINV_OK; }
  void geometry(int out[4]) const { out[0] = out[1] = out[2] = out[3] = 0; }
  template <typename P>
  int launch_xform(P&, bool, const T*, long long, void*, T*, T, int = -1) { return fail(SPECINV_EUNSUPPORTED, "no fused path"); }
  template <typename P>
  int launch_inverse_ola(P&, const void*, T*, long long, T, T**, bool* used) {
    *used = false;
    return SPECINV_OK;
  }
  template <typename P>
  int begin(P&, int, const void*, const void*, double*) { return fail(SPECINV_EUNSUPPORTED, "no fused path for this dtype"); }
  template <typename P>
  int iterate(P&, int, bool) { return fail(SPECINV_EUNSUPPORTED, "no fused path for this dtype"); }
  template <typename P>
  int get_wave(P&, T*) { return fail(SPECINV_EUNSUPPORTED, "no fused path for this dtype"); }
  template <typename P>
  int get_state_spec(P&, int, cplx<T>*) { return fail(SPECINV_EUNSUPPORTED, "no fused path for this dtype"); }
};

// run STMT with `RR` bound to the compile-time registers-per-lane count R = n_fft / 128
#define SPECINV_R_SWITCH(RV, ...)                      \
  switch (RV) {                                        \
    case 4: { constexpr int RR = 4; __VA_ARGS__; } break;   \
    case 8: { constexpr int RR = 8; __VA_ARGS__; } break;   \
    case 16: { constexpr int RR = 16; __VA_ARGS__; } break; \
    default: { constexpr int RR = 32; __VA_ARGS__; } break; \
  }

#ifndef SPECINV_TD_SKEW_DEFAULT
#define SPECINV_TD_SKEW_DEFAULT 10
#endif
#ifndef SPECINV_K4_SKEW1            // begin shifts of the second / third chunk of a triple (frames) at BASELINE C4's launch shape
#define SPECINV_K4_SKEW1 4          // (C4 step, two runs each: "0,0" 30.38 / 30.38 ms, "3,4" 30.29 / 30.26, "4,6" 29.93 / 29.97, "5,8" 30.49 / 30.20,
#define SPECINV_K4_SKEW2 6          //  "6,9" 30.27 / 30.21, "6,11" 30.78 / 30.68, "8,12" 30.40 / 30.34: the kernel sits on the memory system, the balance buys 1.4 %)
#endif

template <>
struct FastState<float> {
  using v2f = fast::v2f;
  using v4f = fast::v4f;
  bool supported = false;
  bool semi = false;   // frame kernels instead of k_fused (hop != n_fft/2, /4, /8, centre = False, small problems)
  bool hopk = false;   // ... k_hop (overlap-add in LDS, chunks of frames) rather than k_semi + k_ola
  int semi_grid = 0;
  int R = 0;
  int OV = 0;          // n_fft / hop of the fused kernel (2, 4 or 8)
  bool state_in_place = true;
  bool use_template = false;   // tests: run k_fused<R, 4> where the tuned copy k_fused4<R> would run (SPECINV_FUSED_TEMPLATE=1)
  int chunk = 32, nchunks = 0, n_waves = 0, n_partials = 0;
  int skew = 0;        // frames every odd chunk cedes to the even chunk before it (chunk_begin; set per Griffin-Lim run in begin_t)
  int cur = 0;   // index of the buffers holding the current state
  int mode = fast::MODE_GLA;
  FastBuf xb[2], xtail[2], Pb[2], Pmid[2], mpairs, mmid, inv_env, scratch;
  // a two-sided spectrogram (onesided=False): the frame kernel k_semi2 with the mirror bins' state and target beside the lower
  // half's (FastArgs::P2_out); no fused / chunked / signal-form kernels, no stand-alone transforms
  bool two = false;
  FastBuf Pb2, Pmid2, mpairs2, mmid2;
  // ADMM carries Y = X + U in Pb (FastArgs).  X and U themselves are only written when the caller has asked for them
  // (specinv_plan_keep_state), by the last iteration of every iterate() call.
  bool keep_state = false, xu_valid = false;
  // the projection in the reference's operation order with correctly rounded factors and a true division by the envelope (the
  // default kernels); false: the approximate copies of the tu_approx_*.hip units, 3 % faster on the headline step
  // (specinv_plan_set_exact)
  bool exact = true;
  FastBuf Xb, Xmid, Ub, Umid;
  // Griffin-Lim on k_fused4_td: the momentum state is the signal z (zb), Pb keeps the starting spectrum c0
  bool td = false;
  int td_t = 0;          // closure calls so far (z_1 = x_1: the first call reads x itself)
  FastBuf zb[2];

  // the optional X / U outputs of an ADMM iteration (`last`: the last iteration of an iterate() call)
  template <typename P>
  int want_xu(P& pl, fast::FastArgs& a, bool last) {
    if (mode != fast::MODE_ADMM) return SPECINV_OK;
    xu_valid = false;
    if (!keep_state || !last) return SPECINV_OK;
    SI_TRY(reserve_xu(pl));
    a.X_out = Xb.template as<v4f>();
    a.U_out = Ub.template as<v4f>();
    a.Xmid_out = Xmid.template as<v2f>();
    a.Umid_out = Umid.template as<v2f>();
    xu_valid = true;
    return SPECINV_OK;
  }
  template <typename P>
  int reserve_xu(P& pl) {
    const long long nf = (long long)pl.B() * pl.Tn();
    const size_t pbytes = (size_t)nf * (R / 2) * 64 * sizeof(v4f);
    SI_TRY(Xb.reserve(pbytes));
    SI_TRY(Ub.reserve(pbytes));
    SI_TRY(Xmid.reserve(nf * sizeof(v2f)));
    SI_TRY(Umid.reserve(nf * sizeof(v2f)));
    return SPECINV_OK;
  }

  int setup(const specinv_stft_cfg& cfg, const std::vector<float>&, int64_t length, int pad) {
    supported = false;
    xform_ok = false;
    two = false;
    if (cfg.dtype != SPECINV_F32) return SPECINV_OK;
    if (const char* e = getenv("SPECINV_DISABLE_FAST")) {
      if (e[0] == '1') return SPECINV_OK;
    }
    const bool size_ok = cfg.n_fft == 512 || cfg.n_fft == 1024 || cfg.n_fft == 2048 || cfg.n_fft == 4096;
    if (!size_ok) return SPECINV_OK;
    if (!cfg.onesided) {
      // two-sided: the frame kernels only - k_semi2 + gather overlap-add, or k_hop2 over chunks of frames (round 5;
      // SPECINV_DISABLE_TWOSIDED=1: the coverage kernels); no fused / signal-form kernels, no stand-alone transforms
      if (const char* e = getenv("SPECINV_DISABLE_TWOSIDED")) {
        if (e[0] == '1') return SPECINV_OK;
      }
      if (pad >= length) return SPECINV_OK;
      two = true;
    } else {
      xform_ok = true;              // any hop, any pad mode, centred or not
      xform_R = cfg.n_fft / 128;
    }
    R = cfg.n_fft / 128;
    semi = false;
    state_in_place = true;     // (same speed as ping-pong buffers, measured; a third less memory)
    if (const char* e = getenv("SPECINV_STATE_INPLACE")) state_in_place = e[0] != '0';
    use_template = false;
    if (const char* e = getenv("SPECINV_FUSED_TEMPLATE")) use_template = e[0] == '1';
    // fused kernel: hop = n_fft / 2, / 4 or / 8 (whole registers per hop-block), centred, enough frames
    OV = 0;
    for (int o : {2, 4, 8})
      if (cfg.hop_length * o == cfg.n_fft && R % o == 0) OV = o;
    if (two) OV = 0;
    if (const char* e = getenv("SPECINV_DISABLE_FUSED")) {   // tests: put the shape on the frame kernel
      if (e[0] == '1') OV = 0;
    }
    // small problems are latency-bound on the fused kernel (a wave walks >= 8 frames one after the other): below
    // ~6 k frames the frame kernel, one frame per wave, finishes an iteration sooner (measured: 1 x 512 frames at
    // n_fft 1024 10 vs 27 us, 16 x 256 26 vs 32 us; 16 x 512 at n_fft 2048 69 vs 55 us)
    long long small_below = 6144;
    if (const char* e = getenv("SPECINV_SMALL_FRAMES")) small_below = atoll(e);       // (tests pin the chunked kernel with 0)
    const bool small = (long long)cfg.batch * cfg.n_frames < small_below;
    if (!cfg.center || OV == 0 || cfg.n_frames < OV + 2 || pad >= length || small) {
      // any other hop / centring: frame kernel on the wave-level FFT + gather overlap-add (k_semi)
      if (const char* e = getenv("SPECINV_DISABLE_SEMI")) {
        if (e[0] == '1') return SPECINV_OK;
      }
      semi = true;
      hopk = false;
      OV = 0;
      chunk = cfg.n_frames;
      nchunks = 1;
      const long long nf = (long long)cfg.batch * cfg.n_frames;
      semi_grid = (int)std::min<long long>((nf + 3) / 4, 256 * 8);
      n_waves = semi_grid * 4;
      supported = true;
      // Large enough problems keep the overlap-add on the chip (k_hop): a wave per chunk of frames, 8-wave workgroups,
      // one per CU (2048 wave slots).  A chunk must emit the n_fft - hop samples it shares with its predecessor with
      // its regular frames: at least (n_fft - 1) / hop + 1 frames.  n_fft 4096 does not fit (ring + scratch in LDS).
      // (a wave walks >= 8 frames one after the other there: measured against k_semi + k_ola, it pays from ~12 k frames
      // at n_fft 2048 and ~32 k frames at 1024 and 512)
      // (round 5's sweep, n_fft 2048 / hop 333, T 300: B 48 = 14 400 frames 145 M against 128 M it*frames/s for k_semi + k_ola,
      // B 32 = 9 600 frames 102 against 120: the crossover at n_fft 2048 is nearer 12 k; n_fft 1024 / hop 300 at 14 400: 260 vs 257)
      long long hop_from = R >= 16 ? 12288 : 32768;
      if (const char* e = getenv("SPECINV_SMALL_FRAMES")) hop_from = atoll(e);          // (tests: 0 pins the chunked kernels)
      bool want_hop = R <= 16 && (long long)cfg.batch * cfg.n_frames >= hop_from && !small && cfg.hop_length >= 1 &&
                      cfg.hop_length <= cfg.n_fft && pad < length;
      if (const char* e = getenv("SPECINV_DISABLE_HOP")) {
        if (e[0] == '1') want_hop = false;
      }
      if (want_hop) {
        const int floor_ch = std::max(8, (cfg.n_fft - 1) / cfg.hop_length + 1);
        // wave slots: one 8-wave workgroup per CU at n_fft 2048 (LDS), two at 1024, three at 512 (registers allow it)
        long long slots = R >= 16 ? 2048 : R == 8 ? 4096 : 6144;
        if (const char* e = getenv("SPECINV_HOP_SLOTS")) slots = atoll(e);
        int best_nch = 1;
        double best_cost = 1e300;
        for (int nch = 1; nch <= std::max(1, cfg.n_frames / floor_ch); ++nch) {
          const long long waves = (long long)cfg.batch * nch;
          const long long rounds = (waves + slots - 1) / slots;
          const int longest = (cfg.n_frames + nch - 1) / nch;
          const double cost = (double)rounds * (longest + 2.0) * (1.0 + 0.0015 * std::max(0, longest - 32));
          if (cost < best_cost - 1e-9 || (cost < best_cost + 1e-9 && nch > best_nch)) {
            best_cost = cost;
            best_nch = nch;
          }
        }
        if (const char* e = getenv("SPECINV_FAST_CHUNK")) {
          const int v = atoi(e);
          if (v >= 1) best_nch = std::max(1, cfg.n_frames / std::min(std::max(v, floor_ch), cfg.n_frames));
        }
        hopk = true;
        nchunks = best_nch;
        chunk = (cfg.n_frames + nchunks - 1) / nchunks;
        n_waves = cfg.batch * nchunks;
        if (getenv("SPECINV_DEBUG")) fprintf(stderr, "specinv: frame kernel with LDS overlap-add R=%d hop=%d chunks=%d of <=%d frames, %d waves\n", R, cfg.hop_length, nchunks, chunk, n_waves);
      }
      return SPECINV_OK;
    }
    // Frames per wave.  A launch takes about rounds x (longest chunk) frame times, rounds = ceil(waves / wave slots):
    // pick the chunk count that minimises it (the chip holds 2 waves of these kernels per SIMD, 3 at n_fft 512, 1 at
    // 4096), e.g. 64 x 1024 frames -> 32 chunks of 32 (2048 waves, one round), 96 x 1024 -> 21 chunks of 49 (2016 waves,
    // one round) instead of 32 chunks (3072 waves, two rounds).  A chunk boundary costs OV - 1 split hop-blocks, and the
    // reflected edge samples must not fall on split blocks (first / last chunk long enough): chunks of >= 8 (16) frames.
    const int floor_ch = OV == 8 ? 16 : 8;
    long long slots = 1024LL * (R >= 32 ? 1 : R <= 4 ? 3 : 2);
    if (SPECINV_R8_W3 && R == 8) slots = 3072;
    if (const char* e = getenv("SPECINV_FUSED_SLOTS")) slots = atoll(e);      // (experiments: wave slots of the chip)
    // SPECINV_CU_BUDGET=k: plan the launch for a chip of 256 - k compute units - the chunk count is chosen so that one round of
    // workgroups leaves k CUs free.  For multi-GPU runs whose RCCL gather overlaps the next step's launches: an iteration
    // workgroup takes a whole CU's registers, so RCCL's workgroups can only run beside a launch that does not fill the chip
    // (bench.py --gather-kernel-budget; C2: 30 chunks of 34 frames on 240 CUs instead of 32 x 32 on 256: +6 % per launch).
    if (const char* e = getenv("SPECINV_CU_BUDGET")) {
      const int k = atoi(e);
      if (k > 0 && k < 256) slots = slots * (256 - k) / 256;
    }
    int best_nch = 1;
    double best_cost = 1e300;
    for (int nch = 1; nch <= std::max(1, cfg.n_frames / floor_ch); ++nch) {
      const long long waves = (long long)cfg.batch * nch;
      const long long rounds = (waves + slots - 1) / slots;
      const int longest = (cfg.n_frames + nch - 1) / nch;
      // (+2: what a chunk boundary costs - split blocks, pipeline fill; measured: 2 rounds of 16-frame chunks are 8 %
      // slower than 1 round of 32.  Last factor: chunks beyond 32 frames measured ~5 % slower than two rounds of 32)
      const double cost = (double)rounds * (longest + 2.0) * (1.0 + 0.0015 * std::max(0, longest - 32));
      if (cost < best_cost - 1e-9 || (cost < best_cost + 1e-9 && nch > best_nch)) {
        best_cost = cost;
        best_nch = nch;
      }
    }
    // The waves of a SIMD do not run at the same speed (begin_t: skewed chunks): where the skew applies - hop = n_fft/4 at n_fft
    // 2048 (chunk pairs) and 1024 (chunk triples) - a chunk count that pairs / triples up is preferred when it costs at most 4 %
    // more frame times than the best one (round 5: B 65 or 100 at T 1024 took 31 / 20 chunks and ran unskewed)
    if (OV == 4 && (R == 16 || (SPECINV_R8_W3 && R == 8))) {
      const int mult = R == 16 ? 2 : 3;
      auto cost_of = [&](int nch) {
        const long long waves = (long long)cfg.batch * nch;
        const long long rounds = (waves + slots - 1) / slots;
        const int longest = (cfg.n_frames + nch - 1) / nch;
        return (double)rounds * (longest + 2.0) * (1.0 + 0.0015 * std::max(0, longest - 32));
      };
      if (best_nch % mult != 0) {
        int pick = 0;
        double pick_cost = 1e300;
        for (int nch = std::max(mult, best_nch - mult); nch <= std::min(best_nch + mult, std::max(1, cfg.n_frames / floor_ch)); ++nch)
          if (nch % mult == 0 && cost_of(nch) < pick_cost) {
            pick = nch;
            pick_cost = cost_of(nch);
          }
        if (pick > 0 && pick_cost <= 1.04 * best_cost) best_nch = pick;
      }
    }
    nchunks = best_nch;
    if (const char* e = getenv("SPECINV_FAST_CHUNK")) {
      const int v = atoi(e);
      if (v >= 4) nchunks = std::max(1, cfg.n_frames / std::min(std::max(v, OV == 8 ? 13 : 4), cfg.n_frames));
    }
    chunk = (cfg.n_frames + nchunks - 1) / nchunks;   // frames split as evenly as possible (sizes differ by at most one)
    n_waves = cfg.batch * nchunks;
    if (getenv("SPECINV_DEBUG")) fprintf(stderr, "specinv: fused R=%d OV=%d chunks=%d of <=%d frames, %d waves (%lld slots)\n", R, OV, nchunks, chunk, n_waves, slots);
    supported = true;
    return SPECINV_OK;
  }

  // `spec_user` (B, F, T) complex and `mag_user` (B, F, T) are the caller's / phase_init's arrays
  // spec_user == nullptr: the starting spectrum is phase_init(mag_user), produced in pair order by k_phase_init_pairs
  template <int RR, typename P>
  int begin_t(P& pl, int md, const v2f* spec_user, const float* mag_user, double* sum_m2_out) {
    using G = fast::Geo<RR>;
    const int hop = pl.cfg.hop_length;
    mode = md;
    // (n_fft 4096 runs one wave per SIMD: its vector latency, not the state traffic, is what bounds it there - the signal form
    // measured 0.360 against 0.340 ms per iteration and is not used)
    if (two || !specinv_approx_units_built()) exact = true;   // (no approximate copy: of the two-sided kernels; in a default build)
    td = md == fast::MODE_GLA && (!semi || hopk) && !use_template && !keep_state && RR <= 16 && !two;
    // (the reference-chain build leaves the real-FFT split unscaled, which is exact only for a power-of-two fwd_scale / 2)
    if (exact && pl.cfg.normalized && !hopk) td = false;
    // k_hop_td writes two signals and re-reads z_t where k_hop writes one: at large hops its emission loop overtakes the saved state
    // traffic.  Measured crossovers (late iterations, 65 536 frames, tools/r02_hop_td2.sh), emission two samples at a time (even hop,
    // padding and length) / one at a time: n_fft 2048: wins up to hop 768 (0.350 vs 0.367 ms), loses at 1000 / wins at 333, loses at
    // 601; n_fft 1024: wins everywhere measured (hop 800: 0.195 vs 0.240) / wins at 301; n_fft 512: wins at 300, loses at 400 / wins
    // at 100, loses at 201
    const bool emit_pairs = ((hop | pl.pad) & 1) == 0 && (pl.length & 1) == 0;
    int hop_td_max = emit_pairs ? (RR == 4 ? 320 : RR == 8 ? 1024 : 800) : (RR == 4 ? 128 : RR == 8 ? 448 : 416);
    if (const char* e = getenv("SPECINV_HOP_TD_MAX")) hop_td_max = atoi(e);          // (experiments)
    if (hopk && hop > hop_td_max) td = false;
    if (const char* e = getenv("SPECINV_DISABLE_TD")) {      // tests: the spectral-state kernel
      if (e[0] == '1') td = false;
    }
    td_t = 0;
    // Skewed chunks.  Two waves share a SIMD and the arbiter serves the OLDER one first whenever both have an instruction ready:
    // at BASELINE C2 the wave in hardware slot 0 ran 0.132 frames per kilotick against its neighbour's 0.077 and finished after
    // 69 % of the launch (237 k against 337 k ticks on every one of the 1024 SIMDs, tools/td_waves.py); the neighbour ran the last
    // third alone at 0.143 - a SIMD with two waves does 0.209.  Chunks of 42 and 22 frames instead of 32 and 32 let both finish
    // nearly together (late launch 0.172-0.176 -> 0.166-0.168 ms at 8 ... 10 frames of skew, 0.169 / 0.172 at 12 / 14; the
    // evaluating launches and the initial ISTFT like less of it: whole C2 step 20.54 / 20.18 / 20.12 / 20.14 / 20.41 ms at
    // 0 / 6 / 8 / 10 / 12, tools/log/r03_skewstep.sh; the other overlaps of n_fft 2048 gain 3.5 % (hop 256) and 2 % (hop 1024)
    // at the same 8, tools/log/r03_skew_ov.sh).  Only for
    // the launch shape this was measured on - the signal-form kernel at two waves per SIMD with exactly as many waves as the
    // chip has slots for them (BASELINE C2 per GPU: 2048).  SPECINV_TD_SKEW overrides (experiments; 0 switches it off).
    // Also tried: s_setprio by frame parity or by time slice so that the two waves take turns (-2...3 %, no better with the skew).
    // Round 5: not only at the launch shape this was measured on.  With one 8-wave workgroup per CU (fused_wgw) waves i and i + 4
    // of a workgroup share a SIMD whatever the number of workgroups or rounds, so the pairing holds for any launch that puts two
    // waves on a SIMD (more waves than SIMDs); the skew scales with the chunk: a quarter of its frames (10 of 32 at C2), the
    // shorter chunk keeping >= 8 (the floor of a chunk: its seams).
    skew = 0;
    {
      const int len_ch = pl.Tn() / std::max(1, nchunks);
      if (td && !semi && RR == 16 && OV == 4 && n_waves > 1024 && (nchunks & 1) == 0 && len_ch >= 12) {
        skew = n_waves == 2048 && len_ch == 32 ? SPECINV_TD_SKEW_DEFAULT : std::min((len_ch + 2) / 4, len_ch - 8);
        if (skew < 2) skew = 0;
      }
      // ... and the n_fft 1024 kernels (spectral state or signal form) at three waves per SIMD (12-wave workgroups, the hardware
      // slot is the wave's index in the workgroup / 4, kernels_fused.h): chunk triples, the oldest wave the longest.  BASELINE
      // C4's shard: 3072 waves, chunks of 64 frames, begin shifts 4 / 6; other chunk lengths scale them.
      if (!semi && RR == 8 && OV == 4 && !use_template && fused_wgw() == 12 && nchunks % 3 == 0 && len_ch >= 16) {
        const int s1 = len_ch == 64 ? SPECINV_K4_SKEW1 : (SPECINV_K4_SKEW1 * len_ch + 32) / 64;
        const int s2 = len_ch == 64 ? SPECINV_K4_SKEW2 : (SPECINV_K4_SKEW2 * len_ch + 32) / 64;
        if ((s1 | s2) && len_ch - s2 >= 8 && len_ch + s2 - s1 >= 8) skew = 0x10000 | (s1 << 8) | s2;
      }
    }
    if (const char* e = getenv("SPECINV_K4_SKEW")) {          // "s1,s2" (experiments; "0,0" switches it off)
      int s1 = 0, s2 = 0;
      if (sscanf(e, "%d,%d", &s1, &s2) == 2 && !semi && fused_wgw() == 12 && nchunks % 3 == 0 && n_waves % 12 == 0 && s1 >= 0 && s2 >= 0 &&
          s1 < 200 && s2 < 200 && pl.Tn() / nchunks - s2 >= 8 && pl.Tn() / nchunks + s2 - s1 >= 8)
        skew = (s1 | s2) ? (0x10000 | (s1 << 8) | s2) : 0;
    }
    if (const char* e = getenv("SPECINV_TD_SKEW")) {
      const int v = atoi(e);
      if (v == 0 || (td && !semi && (nchunks & 1) == 0 && (n_waves & 1) == 0 && pl.Tn() / nchunks - v >= 8)) skew = v;
    }
    if (td) {
      SI_TRY(zb[0].reserve((size_t)pl.B() * pl.length * sizeof(float)));
      SI_TRY(zb[1].reserve((size_t)pl.B() * pl.length * sizeof(float)));
    }
    const long long nf = (long long)pl.B() * pl.Tn();
    const size_t pbytes = (size_t)nf * G::H * 64 * sizeof(v4f);
    const size_t tail_bytes = (size_t)pl.B() * nchunks * (OV > 0 ? OV - 1 : 0) * hop * sizeof(float);
    if (hopk) SI_TRY(xtail[0].reserve((size_t)pl.B() * nchunks * (pl.N() - hop) * sizeof(float) + 16));
    for (int i = 0; i < ((semi && !hopk) ? 1 : 2); ++i) {     // x (and the chunk tails) ping-pong between iterations
      if (!semi) SI_TRY(xtail[i].reserve(tail_bytes));
      SI_TRY(xb[i].reserve((size_t)pl.B() * pl.length * sizeof(float)));
      if (i == 1 && state_in_place) continue;      // the spectral state is updated in place
      SI_TRY(Pb[i].reserve(pbytes));
      SI_TRY(Pmid[i].reserve(nf * sizeof(v2f)));
    }
    if (semi && !hopk) SI_TRY(pl.frames_needed());
    SI_TRY(mpairs.reserve((size_t)nf * (G::H / 2) * 64 * sizeof(v4f)));
    SI_TRY(mmid.reserve(nf * sizeof(float)));
    if (two) {
      SI_TRY(Pb2.reserve(pbytes));
      SI_TRY(Pmid2.reserve(nf * sizeof(v2f)));
      SI_TRY(mpairs2.reserve((size_t)nf * (G::H / 2) * 64 * sizeof(v4f)));
      SI_TRY(mmid2.reserve(nf * sizeof(float)));
    }
    SI_TRY(inv_env.reserve(pl.length * sizeof(float)));
    cur = 0;
    SI_CHECK(!two || spec_user != nullptr, SPECINV_ESTATE, "the two-sided frame kernel takes its starting spectrum in the user layout");
    if (spec_user == nullptr) {
      const int nwg = pl.B() * G::H;
      SI_TRY(pl.partials.reserve(std::max<size_t>((size_t)nwg, 3 * 1024) * sizeof(double)));
      const size_t lds = (size_t)64 * 129 * (sizeof(v2f) + sizeof(float));
      const void* fn = (const void*)fast::k_phase_init_pairs<RR>;
      SI_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      const float* mg = mag_user;
      v4f* pp = Pb[0].template as<v4f>();
      v2f* pm = Pmid[0].template as<v2f>();
      float* mp = mpairs.template as<float>();
      float* mm = mmid.template as<float>();
      double* part = pl.partials.template as<double>();
      int Tn = pl.Tn(), hp = hop;
      void* kargs[] = {&mg, &pp, &pm, &mp, &mm, &part, &Tn, &hp};
      SI_HIP(hipLaunchKernel(fn, dim3(nwg), dim3(1024), kargs, lds, pl.stream));
      hipLaunchKernelGGL(k_finish_partials, dim3(1), dim3(256), 0, pl.stream, pl.partials.template as<double>(), (int64_t)nwg, 1,
                         pl.sums.template as<double>() + 4);
      SI_HIP(hipGetLastError());
      SI_HIP(hipMemcpyAsync(sum_m2_out, pl.sums.template as<double>() + 4, sizeof(double), hipMemcpyDeviceToHost, pl.stream));
    } else {
      const int rows = pl.n_freq;                   // G::M + 1, or N for a two-sided spectrogram
      const dim3 grid((pl.Tn() + 31) / 32, (G::M + 1 + 31) / 32, pl.B()), blk(32, 8);
      hipLaunchKernelGGL((fast::k_user_spec_to_pairs<RR>), grid, blk, 0, pl.stream, spec_user, Pb[0].template as<v2f>(),
                         Pmid[0].template as<v2f>(), pl.Tn(), rows, 0);
      SI_HIP(hipGetLastError());
      const long long nblk1 = (long long)grid.x * grid.y * grid.z, nblk = two ? 2 * nblk1 : nblk1;
      SI_TRY(pl.partials.reserve(std::max<size_t>((size_t)nblk, 3 * 1024) * sizeof(double)));
      hipLaunchKernelGGL((fast::k_user_mag_to_pairs<RR>), grid, blk, 0, pl.stream, mag_user, mpairs.template as<float>(),
                         mmid.template as<float>(), pl.Tn(), pl.partials.template as<double>(), rows, 0);
      SI_HIP(hipGetLastError());
      if (two) {                                    // the mirror rows N - f into the same layout (sum of m^2: their partials behind)
        hipLaunchKernelGGL((fast::k_user_spec_to_pairs<RR>), grid, blk, 0, pl.stream, spec_user, Pb2.template as<v2f>(),
                           Pmid2.template as<v2f>(), pl.Tn(), rows, 1);
        hipLaunchKernelGGL((fast::k_user_mag_to_pairs<RR>), grid, blk, 0, pl.stream, mag_user, mpairs2.template as<float>(),
                           mmid2.template as<float>(), pl.Tn(), pl.partials.template as<double>() + nblk1, rows, 1);
        SI_HIP(hipGetLastError());
      }
      hipLaunchKernelGGL(k_finish_partials, dim3(1), dim3(256), 0, pl.stream, pl.partials.template as<double>(), nblk, 1,
                         pl.sums.template as<double>() + 4);
      SI_HIP(hipGetLastError());
      SI_HIP(hipMemcpyAsync(sum_m2_out, pl.sums.template as<double>() + 4, sizeof(double), hipMemcpyDeviceToHost, pl.stream));
    }
    xu_valid = false;
    if (md == fast::MODE_ADMM && keep_state) {      // methods.py:447-449: X = the start spectrum, U = 0 (Y = X is already in Pb)
      SI_TRY(reserve_xu(pl));
      SI_HIP(hipMemcpyAsync(Xb.p, Pb[0].p, pbytes, hipMemcpyDeviceToDevice, pl.stream));
      SI_HIP(hipMemcpyAsync(Xmid.p, Pmid[0].p, nf * sizeof(v2f), hipMemcpyDeviceToDevice, pl.stream));
      SI_HIP(hipMemsetAsync(Ub.p, 0, pbytes, pl.stream));
      SI_HIP(hipMemsetAsync(Umid.p, 0, nf * sizeof(v2f), pl.stream));
      xu_valid = true;
    }
    if (hopk) {
      hipLaunchKernelGGL(fast::k_reciprocal, dim3((unsigned)ceil_div(pl.length, 256)), dim3(256), 0, pl.stream,
                         pl.env.template as<float>(), inv_env.template as<float>(), (long long)pl.length, exact ? 1 : 0);
      SI_HIP(hipGetLastError());
    }
    if (semi) {
      // x0 = ISTFT(start spectrum): synthesis frames from the pair layout, then the overlap-add
      if constexpr (RR <= 16) {
        if (hopk) SI_TRY((launch_hop<RR, fast::MODE_INIT, false>(pl)));
      }
      if (!hopk) SI_TRY((launch_semi<RR, fast::MODE_INIT, false>(pl)));
      SI_HIP(hipStreamSynchronize(pl.stream));   // *sum_m2_out is valid from here on
      return SPECINV_OK;
    }
    hipLaunchKernelGGL(fast::k_reciprocal, dim3((unsigned)ceil_div(pl.length, 256)), dim3(256), 0, pl.stream,
                       pl.env.template as<float>(), inv_env.template as<float>(), (long long)pl.length, exact ? 1 : 0);
    SI_HIP(hipGetLastError());
    SI_HIP(hipMemsetAsync(xtail[0].p, 0, tail_bytes, pl.stream));   // x0 below is written whole
    // x0 = ISTFT(start spectrum) (methods.py:233 / :453) straight from the pair layout
    fast::FastArgs a{};
    a.x_out = xb[0].template as<float>();
    a.P_in = Pb[0].template as<v4f>();
    a.Pmid_in = Pmid[0].template as<v2f>();
    a.window = pl.window.template as<float>();
    a.inv_env = inv_env.template as<float>();
    a.T = pl.Tn();
    a.nchunks = nchunks;
    a.skew = skew;
    a.n_waves = n_waves;
    a.L = pl.length;
    a.fwd_scale = pl.fc.fwd_scale;
    a.inv_scale = pl.fc.inv_scale;
    const size_t lds = G::lds_bytes(4);
    const void* fn = nullptr;
    if constexpr (RR % 8 == 0) {
      if (OV == 8) fn = (const void*)fast::k_fused_istft<RR, 8>;
    }
    if (OV == 4) fn = (const void*)fast::k_fused_istft<RR, 4>;
    if (OV == 2) fn = (const void*)fast::k_fused_istft<RR, 2>;
    if (!exact && fn != nullptr) fn = approx_fused(RR, OV, 2, 0, 0);
    SI_CHECK(fn != nullptr, SPECINV_EUNSUPPORTED, "no fused kernel for n_fft / hop = %d", OV);
    SI_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    void* kargs[] = {&a};
    SI_HIP(hipLaunchKernel(fn, dim3((n_waves + 3) / 4), dim3(256), kargs, lds, pl.stream));
    SI_HIP(hipStreamSynchronize(pl.stream));   // *sum_m2_out is valid from here on
    return SPECINV_OK;
  }

  template <typename P>
  int begin(P& pl, int md, const void* spec_user, const void* mag_user, double* sum_m2_out) {
    const v2f* s = static_cast<const v2f*>(spec_user);
    const float* m = static_cast<const float*>(mag_user);
    int rc = SPECINV_OK;
    SPECINV_R_SWITCH(R, rc = begin_t<RR>(pl, md, s, m, sum_m2_out));
    return rc;
  }

  // wave-level FFT usable for stand-alone transforms of this plan (any hop / frame count)
  bool xform_ok = false;
  int xform_R = 0;

  template <typename P>
  int launch_xform(P& pl, bool forward, const float* x, long long len, fast::v2f* spec, float* frames, float scale,
                   int pad_mode = -1) {
    fast::FastXformArgs a{};
    a.x = x;
    a.spec = spec;
    a.frames = frames;
    a.window = pl.window.template as<float>();
    a.len = len;
    a.n_frames_total = (long long)pl.B() * pl.Tn();
    a.T = pl.Tn();
    a.hop = pl.cfg.hop_length;
    a.pad = pl.pad;
    a.pad_mode = pad_mode >= 0 ? pad_mode : pl.cfg.pad_mode;
    a.scale = scale;
    size_t lds = 0;
    const void* fn = nullptr;
    SPECINV_R_SWITCH(xform_R, lds = fast::Geo<RR>::lds_bytes(4);
                     fn = forward ? (const void*)fast::k_fast_stft<RR> : (const void*)fast::k_fast_inverse_frames<RR>);
    const unsigned grid = (unsigned)std::min<long long>((a.n_frames_total + 3) / 4, 256 * 12);
    SI_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    void* kargs[] = {&a};
    SI_HIP(hipLaunchKernel(fn, dim3(grid), dim3(256), kargs, lds, pl.stream));
    return SPECINV_OK;
  }

  // Adjoint of the analysis (gradient frames -> overlap-add over the padded signal) in one launch for big batches of
  // frames; `*used` stays false when the shape is not covered (the caller then runs inverse frames + gather).
  // `margins` receives the pad samples on either side of the signal.
  template <typename P>
  int launch_inverse_ola(P& pl, const fast::v2f* spec, float* out, long long len, float scale, float** margins, bool* used) {
    *used = false;
    if (!xform_ok || xform_R > 16 || pl.force_generic) return SPECINV_OK;
    const int N = pl.N(), hop = pl.cfg.hop_length, T = pl.Tn(), B = pl.B();
    long long from = xform_R >= 16 ? 16384 : 32768;
    if (const char* e = getenv("SPECINV_SMALL_FRAMES")) from = atoll(e);
    if (const char* e = getenv("SPECINV_DISABLE_HOP")) {
      if (e[0] == '1') return SPECINV_OK;
    }
    if ((long long)B * T < from || hop < 1 || hop > N || pl.pad >= len) return SPECINV_OK;
    // the kernel writes every sample of `out` only if the frames cover the padded signal exactly
    if ((long long)(T - 1) * hop + N != len + 2LL * pl.pad) return SPECINV_OK;
    const int floor_ch = std::max(8, (N - 1) / hop + 1);
    const int nch = (int)std::max(1LL, std::min<long long>(T / floor_ch, std::max(1, 2048 / B)));
    const int wgw = 8, keep = N - hop, n_w = B * nch;
    SI_TRY(hop_inv_tail.reserve((size_t)B * nch * keep * sizeof(float) + 16));
    SI_TRY(hop_inv_margins.reserve((size_t)B * 2 * std::max(1, pl.pad) * sizeof(float)));
    fast::HopInvArgs a{};
    a.spec = spec;
    a.out = out;
    a.margins = hop_inv_margins.template as<float>();
    a.xtail = hop_inv_tail.template as<float>();
    a.window = pl.window.template as<float>();
    a.len = len;
    a.T = T;
    a.nchunks = nch;
    a.n_waves = n_w;
    a.hop = hop;
    a.pad = pl.pad;
    a.scale = scale;
    size_t lds = 0;
    const void* fn = nullptr;
    SPECINV_R_SWITCH(xform_R, if constexpr (RR <= 16) {
      lds = fast::Geo<RR>::lds_bytes(wgw) + (size_t)wgw * fast::Geo<RR>::N * sizeof(float);
      fn = (const void*)fast::k_hop_inverse<RR>;
    });
    SI_CHECK(fn != nullptr, SPECINV_EUNSUPPORTED, "no k_hop_inverse instantiation");
    SI_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    void* kargs[] = {&a};
    SI_HIP(hipLaunchKernel(fn, dim3((n_w + wgw - 1) / wgw), dim3(64 * wgw), kargs, lds, pl.stream));
    if (nch > 1 && keep > 0) {
      const long long total = (long long)B * (nch - 1) * keep;
      hipLaunchKernelGGL(fast::k_hop_tails_raw, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, pl.stream, out,
                         (const float*)a.xtail, T, nch, hop, keep, pl.pad, len, total);
      SI_HIP(hipGetLastError());
    }
    *margins = a.margins;
    *used = true;
    return SPECINV_OK;
  }
  FastBuf hop_inv_tail, hop_inv_margins;

  // waves per workgroup of the fused iteration kernel: k_fused4 takes 8-wave workgroups (one per CU) once every wave
  // slot is filled; fewer waves than slots: smaller workgroups reach more CUs
  int fused_wgw() const {
    if (const char* e = getenv("SPECINV_FUSED_WGW")) return atoi(e);           // (experiments)
    if (SPECINV_R8_W3 && R == 8 && OV == 4 && !use_template) {
      // three waves per SIMD: a 12-wave workgroup is a whole CU, and 8-wave workgroups do not pair up there (2 + 2 waves on a SIMD
      // that holds 3: at 2048 ... 3071 waves every CU's second workgroup waited for the first - round 5's sweep, B 64 x T 300:
      // 316 M against 400 M at B 48).  Below a full chip, 4-wave workgroups: three fit a CU.
      return n_waves >= 3072 ? 12 : 4;
    }
    // (the signal-form kernel at n_fft 2048 measured 2 % faster with two 4-wave workgroups per CU than with one 8-wave one:
    // C2 25.8 vs 26.3 ms per step on one box, three runs each - the opposite of k_fused4)
    // ... except when the chunks are skewed (begin_t): one 8-wave workgroup per CU makes the hardware slot of a wave its index in
    // the workgroup / 4, whatever else runs on the chip - the 4-wave form has to infer it from the dispatch order (measured with
    // the skew: 19.89-20.09 against 20.02-20.16 ms per C2 step, tools/log/r03_wgw8.sh)
    if (td && R == 16 && OV == 4) return (skew != 0 && skew < 0x10000) ? 8 : 4;
    if ((R == 8 || R == 16) && OV == 4 && !use_template) return n_waves >= 2048 ? SPECINV_WGW : 4;
    return 4;
  }
  // {waves per workgroup, chunks per item, waves, kernel: 1 k_fused4, 2 k_fused<R, OV>, 3 k_semi, 4 k_hop, 5 k_fused_td<R, 4> at n_fft 1024 / 2048, 6 k_fused_td<R, OV> otherwise}
  void geometry(int out[4]) const {
    if (semi) {
      out[0] = hopk ? 8 : 4;
      out[3] = hopk ? (td ? 7 : 4) : 3;
    } else {
      out[0] = fused_wgw();
      out[3] = ((R == 8 || R == 16) && OV == 4 && !use_template) ? (td ? 5 : 1) : (td ? 6 : 2);
    }
    out[1] = nchunks;
    out[2] = n_waves;
  }

  template <int RR, int MODE, bool EVAL, typename P>
  int launch(P& pl, const fast::FastArgs& a) {
    using G = fast::Geo<RR>;
    const void* fn = nullptr;
    if constexpr (RR % 8 == 0) {
      if (OV == 8) fn = (const void*)fast::k_fused<RR, 8, MODE, EVAL>;
    }
    if constexpr (RR == 8 || RR == 16) {
      if (OV == 4)                                                       // the tuned copy for the headline shapes
        fn = use_template ? (const void*)fast::k_fused<RR, 4, MODE, EVAL> : (const void*)fast::k_fused4<RR, MODE, EVAL>;
    } else {
      if (OV == 4) fn = (const void*)fast::k_fused<RR, 4, MODE, EVAL>;
    }
    if (OV == 2) fn = (const void*)fast::k_fused<RR, 2, MODE, EVAL>;
    if (!exact && fn != nullptr) fn = approx_fused(RR, OV, MODE, EVAL ? 1 : 0, ((RR == 8 || RR == 16) && OV == 4 && !use_template) ? 1 : 0);
    SI_CHECK(fn != nullptr, SPECINV_EUNSUPPORTED, "no fused kernel for n_fft / hop = %d", OV);
    const int wgw = fused_wgw();
    const size_t lds_used = G::lds_bytes(wgw);
    SI_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_used));
    fast::FastArgs args = a;
#if SPECINV_K4_STAMPS      // diagnostic build (-DSPECINV_K4_STAMPS=1): the 20th launch's waves to $SPECINV_K4_STAMP_DUMP, tools/td_waves.py format
    static unsigned long long* d_k4 = nullptr;
    static int k4_launches = 0;
    if (!d_k4) SI_HIP(hipMalloc(&d_k4, (size_t)n_waves * 4 * sizeof(unsigned long long)));
    args.stamps = d_k4;
#endif
    void* kargs[] = {&args};
    SI_HIP(hipLaunchKernel(fn, dim3((n_waves + wgw - 1) / wgw), dim3(64 * wgw), kargs, lds_used, pl.stream));
#if SPECINV_K4_STAMPS
    if (++k4_launches == 20) {
      if (const char* dump = getenv("SPECINV_K4_STAMP_DUMP")) {
        std::vector<unsigned long long> h((size_t)n_waves * 4);
        SI_HIP(hipMemcpy(h.data(), d_k4, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull;
        for (int wv = 0; wv < n_waves; ++wv) t0 = std::min(t0, h[4 * (size_t)wv + 1]);
        if (FILE* f = fopen(dump, "w")) {
          for (int wv = 0; wv < n_waves; ++wv)
            fprintf(f, "40 %d %u %u %llu %llu %llu\n", wv, (unsigned)(h[4 * (size_t)wv] >> 32), (unsigned)h[4 * (size_t)wv],
                    h[4 * (size_t)wv + 1] - t0, h[4 * (size_t)wv + 2] - t0, h[4 * (size_t)wv + 3]);
          fclose(f);
        }
      }
    }
#endif
    return SPECINV_OK;
  }

  template <int RR, int OVV>
  static const void* td_kernel(bool early, bool ev) {
    return early ? (ev ? (const void*)fast::k_fused_td<RR, OVV, true, true> : (const void*)fast::k_fused_td<RR, OVV, true, false>)
                 : (ev ? (const void*)fast::k_fused_td<RR, OVV, false, true> : (const void*)fast::k_fused_td<RR, OVV, false, false>);
  }
  template <int RR>
  static const void* td_kernel4(bool early, bool ev) {
    return early ? (ev ? (const void*)fast::k_fused4_td<RR, true, true> : (const void*)fast::k_fused4_td<RR, true, false>)
                 : (ev ? (const void*)fast::k_fused4_td<RR, false, true> : (const void*)fast::k_fused4_td<RR, false, false>);
  }
  template <typename P>
  int launch_td(P& pl, const fast::FastArgs& a, bool early, bool ev) {
    const void* fn = nullptr;
    size_t lds_used = 0;
    const int wgw = fused_wgw();
    SPECINV_R_SWITCH(R, if constexpr (RR <= 16) {                      // (n_fft 4096 never takes the signal form: begin_t)
                       lds_used = fast::Geo<RR>::lds_bytes_td(wgw);
                       if constexpr (RR % 8 == 0) { if (OV == 8) fn = td_kernel<RR, 8>(early, ev); }
                       if constexpr (RR == 8 || RR == 16) { if (OV == 4) fn = td_kernel4<RR>(early, ev); }
                       else { if (OV == 4) fn = td_kernel<RR, 4>(early, ev); }
                       if (OV == 2) fn = td_kernel<RR, 2>(early, ev);
                     });
    if (!exact && fn != nullptr) fn = specinv_approx_td(R, OV, early ? 1 : 0, ev ? 1 : 0, ((R == 8 || R == 16) && OV == 4) ? 1 : 0);
    SI_CHECK(fn != nullptr, SPECINV_EUNSUPPORTED, "no fused kernel for n_fft / hop = %d", OV);
    SI_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_used));
    fast::FastArgs args = a;
#if SPECINV_TD_STAMPS
    static unsigned long long* d_stamps = nullptr;
    if (!d_stamps) SI_HIP(hipMalloc(&d_stamps, (size_t)n_waves * 11 * sizeof(unsigned long long)));
    args.stamps = d_stamps;
#endif
    void* kargs[] = {&args};
    SI_HIP(hipLaunchKernel(fn, dim3((n_waves + wgw - 1) / wgw), dim3(64 * wgw), kargs, lds_used, pl.stream));
#if SPECINV_TD_STAMPS
    if (td_t == 40 || td_t == 5) {
      std::vector<unsigned long long> h((size_t)n_waves * 11);
      SI_HIP(hipMemcpy(h.data(), d_stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      const char* names[6] = {"loads + window + slide", "forward FFT", "split / project / fold", "inverse FFT", "window + out + OLA", "loop"};
      double tot[6] = {0}, frames = 0;
      for (int wv = 0; wv < n_waves; ++wv) {
        for (int i = 0; i < 6; ++i) tot[i] += (double)h[(size_t)wv * 8 + i];
        frames += (double)h[(size_t)wv * 8 + 6];
      }
      fprintf(stderr, "k_fused4_td<%d, early=%d, eval=%d> iteration %d: cycles per frame (s_memtime, mean over %d waves)\n", R, (int)early, (int)ev, td_t, n_waves);
      double all = 0;
      for (int i = 0; i < 6; ++i) {
        fprintf(stderr, "  %-24s %8.0f\n", names[i], tot[i] / frames);
        all += tot[i] / frames;
      }
      fprintf(stderr, "  %-24s %8.0f\n", "frame", all);
      // how evenly the waves finish: every wave walks the same number of frames, a launch lasts as long as its slowest wave
      std::vector<double> per_wave((size_t)n_waves);
      for (int wv = 0; wv < n_waves; ++wv) {
        double t = 0;
        for (int i = 0; i < 6; ++i) t += (double)h[(size_t)wv * 8 + i];
        per_wave[(size_t)wv] = t;
      }
      std::sort(per_wave.begin(), per_wave.end());
      double mean = 0;
      for (double v : per_wave) mean += v / n_waves;
      if (const char* dump = getenv("SPECINV_TD_STAMP_DUMP")) {     // wave, xcc, hw_id, begin, end, frames: one line per wave
        FILE* f = fopen(dump, td_t == 5 ? "w" : "a");
        if (f) {
          unsigned long long t0 = ~0ull;
          for (int wv = 0; wv < n_waves; ++wv) t0 = std::min(t0, h[(size_t)n_waves * 8 + 2 * (size_t)wv + 1]);
          for (int wv = 0; wv < n_waves; ++wv) {
            const unsigned long long id = h[(size_t)n_waves * 8 + 2 * (size_t)wv];
            fprintf(f, "%d %d %u %u %llu %llu %llu\n", td_t, wv, (unsigned)(id >> 32), (unsigned)id,
                    h[(size_t)n_waves * 8 + 2 * (size_t)wv + 1] - t0, h[(size_t)n_waves * 10 + (size_t)wv] - t0, h[(size_t)wv * 8 + 6]);
          }
          fclose(f);
        }
      }
      fprintf(stderr, "  per-wave total: min %.0f  p10 %.0f  median %.0f  mean %.0f  p90 %.0f  p99 %.0f  max %.0f  (max / mean %.3f)\n",
              per_wave.front(), per_wave[(size_t)(0.1 * n_waves)], per_wave[(size_t)(0.5 * n_waves)], mean,
              per_wave[(size_t)(0.9 * n_waves)], per_wave[(size_t)(0.99 * n_waves)], per_wave.back(), per_wave.back() / mean);
    }
#endif
    return SPECINV_OK;
  }

  template <int RR, int MODE, bool EVAL, typename P>
  int launch_semi(P& pl, bool last = false) {
    using G = fast::Geo<RR>;
    fast::SemiArgs s{};
    fast::FastArgs& a = s.f;
    a.x_in = xb[0].template as<float>();
    a.P_out = Pb[0].template as<v4f>();
    a.Pmid_out = Pmid[0].template as<v2f>();
    if (MODE == fast::MODE_ADMM) SI_TRY(want_xu(pl, a, last));
    a.m_pairs = mpairs.template as<v4f>();
    a.m_mid = mmid.template as<float>();
    if (two) {
      a.P2_out = Pb2.template as<v4f>();
      a.Pmid2_out = Pmid2.template as<v2f>();
      a.m2_pairs = mpairs2.template as<v4f>();
      a.m2_mid = mmid2.template as<float>();
    }
    a.window = pl.window.template as<float>();
    a.partials = pl.partials.template as<double>();
    a.T = pl.Tn();
    a.pad_mode = pl.cfg.pad_mode;
    a.L = pl.length;
    a.coef = pl.coef;
    a.inv1p = 1.0f / (float)(1.0 + (double)pl.coef);
    a.fwd_scale = pl.fc.fwd_scale;
    a.inv_scale = pl.fc.inv_scale;
    s.frames = pl.frames.template as<float>();
    s.n_frames_total = (long long)pl.B() * pl.Tn();
    s.hop = pl.cfg.hop_length;
    s.pad = pl.pad;
    const size_t lds = G::lds_bytes(4);
    // (two-sided: the reference's operation order only - the approximate copy is not built for k_semi2)
    const void* fn = two ? (const void*)fast::k_semi2<RR, MODE, EVAL>
                         : !exact ? specinv_approx_frame(0, RR, MODE, EVAL ? 1 : 0) : (const void*)fast::k_semi<RR, MODE, EVAL>;
    SI_CHECK(fn != nullptr, SPECINV_EUNSUPPORTED, "no approximate-projection frame kernel for this shape");
    SI_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    void* kargs[] = {&s};
    SI_HIP(hipLaunchKernel(fn, dim3(semi_grid), dim3(256), kargs, lds, pl.stream));
    return pl.launch_ola(pl.frames.template as<float>(), xb[0].template as<float>(), true);
  }

  // one iteration (or the initial ISTFT) of k_hop: reads x from xb[cur], writes xb[cur ^ 1], then mends the chunk seams
  template <int RR, int MODE, bool EVAL, typename P>
  int launch_hop(P& pl, bool last = false) {
    using G = fast::Geo<RR>;
    const int wgw = 8, hop = pl.cfg.hop_length, keep = pl.N() - hop;
    const int nx = MODE == fast::MODE_INIT ? 0 : (cur ^ 1);
    fast::HopArgs s{};
    fast::FastArgs& a = s.f;
    a.x_in = xb[cur].template as<float>();
    a.x_out = xb[nx].template as<float>();
    a.P_out = Pb[0].template as<v4f>();
    a.Pmid_out = Pmid[0].template as<v2f>();
    if (MODE == fast::MODE_ADMM) SI_TRY(want_xu(pl, a, last));
    a.m_pairs = mpairs.template as<v4f>();
    a.m_mid = mmid.template as<float>();
    a.window = pl.window.template as<float>();
    a.partials = pl.partials.template as<double>();
    a.T = pl.Tn();
    a.nchunks = nchunks;
    a.n_waves = n_waves;
    a.pad_mode = pl.cfg.pad_mode;
    a.L = pl.length;
    a.coef = pl.coef;
    a.inv1p = 1.0f / (float)(1.0 + (double)pl.coef);
    a.fwd_scale = pl.fc.fwd_scale;
    a.inv_scale = pl.fc.inv_scale;
    if (two) {
      a.P2_out = Pb2.template as<v4f>();
      a.Pmid2_out = Pmid2.template as<v2f>();
      a.m2_pairs = mpairs2.template as<v4f>();
      a.m2_mid = mmid2.template as<float>();
    }
    s.env = inv_env.template as<float>();
    s.xtail = xtail[0].template as<float>();
    s.hop = hop;
    s.pad = pl.pad;
    const size_t lds = G::lds_bytes(wgw) + (size_t)wgw * G::N * sizeof(float);
    const void* fn = two ? (const void*)fast::k_hop2<RR, MODE, EVAL>
                         : !exact ? specinv_approx_frame(1, RR, MODE, EVAL ? 1 : 0) : (const void*)fast::k_hop<RR, MODE, EVAL>;
    SI_CHECK(fn != nullptr, SPECINV_EUNSUPPORTED, "no approximate-projection chunked frame kernel for this shape");
    SI_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    void* kargs[] = {&s};
    SI_HIP(hipLaunchKernel(fn, dim3((n_waves + wgw - 1) / wgw), dim3(64 * wgw), kargs, lds, pl.stream));
    if (nchunks > 1 && keep > 0) {
      const long long total = (long long)pl.B() * (nchunks - 1) * keep;
      hipLaunchKernelGGL(fast::k_hop_tails, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, pl.stream, a.x_out,
                         (const float*)s.xtail, s.env, pl.Tn(), nchunks, hop, keep, pl.pad, (long long)pl.length, total, exact ? 1 : 0);
      SI_HIP(hipGetLastError());
    }
    cur = nx;
    return SPECINV_OK;
  }

  // the same for Griffin-Lim with the momentum carried as a signal: z from zb[cur] (the first closure call: x itself) to
  // zb[cur ^ 1], x to xb[cur ^ 1]
  template <int RR, typename P>
  int launch_hop_td(P& pl, bool ev, bool need_x) {
    using G = fast::Geo<RR>;
    const int wgw = 8, hop = pl.cfg.hop_length, keep = pl.N() - hop;
    const int nx = cur ^ 1;
    ++td_t;
    const double tds = std::pow(-(double)pl.coef, (double)td_t);
    const bool early = std::fabs(tds) >= 9.3132257461547852e-10;       // 2^-30, as in iterate()
    fast::HopArgs s{};
    fast::FastArgs& a = s.f;
    a.x_in = td_t == 1 ? xb[cur].template as<float>() : zb[cur].template as<float>();
    a.x_out = zb[nx].template as<float>();
    a.x2_in = xb[cur].template as<float>();
    a.x2_out = xb[nx].template as<float>();
    a.P_in = Pb[0].template as<v4f>();
    a.Pmid_in = Pmid[0].template as<v2f>();
    a.tds = (float)tds;
    a.m_pairs = mpairs.template as<v4f>();
    a.m_mid = mmid.template as<float>();
    a.window = pl.window.template as<float>();
    a.partials = pl.partials.template as<double>();
    a.T = pl.Tn();
    a.nchunks = nchunks;
    a.n_waves = n_waves;
    a.pad_mode = pl.cfg.pad_mode;
    a.L = pl.length;
    a.coef = pl.coef;
    a.fwd_scale = pl.fc.fwd_scale;
    a.inv_scale = pl.fc.inv_scale;
    s.env = inv_env.template as<float>();
    s.xtail = xtail[0].template as<float>();
    s.hop = hop;
    s.pad = pl.pad;
    s.write_x = need_x ? 1 : 0;
    const size_t lds = G::lds_bytes(wgw) + (size_t)wgw * G::N * sizeof(float);
    const void* fn = early ? (ev ? (const void*)fast::k_hop_td<RR, true, true> : (const void*)fast::k_hop_td<RR, true, false>)
                           : (ev ? (const void*)fast::k_hop_td<RR, false, true> : (const void*)fast::k_hop_td<RR, false, false>);
    if (!exact) fn = specinv_approx_frame(2, RR, early ? 1 : 0, ev ? 1 : 0);
    SI_CHECK(fn != nullptr, SPECINV_EUNSUPPORTED, "no approximate-projection chunked frame kernel for this shape");
    SI_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    void* kargs[] = {&s};
    SI_HIP(hipLaunchKernel(fn, dim3((n_waves + wgw - 1) / wgw), dim3(64 * wgw), kargs, lds, pl.stream));
    if (nchunks > 1 && keep > 0) {
      const long long total = (long long)pl.B() * (nchunks - 1) * keep;
      hipLaunchKernelGGL(fast::k_hop_tails_td, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, pl.stream, a.x2_out, a.x_out,
                         a.x_in, (const float*)s.xtail, s.env, a.coef, pl.Tn(), nchunks, hop, keep, pl.pad, (long long)pl.length,
                         total, exact ? 1 : 0);
      SI_HIP(hipGetLastError());
    }
    cur = nx;
    return SPECINV_OK;
  }

  template <typename P>
  int iterate_semi(P& pl, int n_iter, bool eval_last) {
    SI_TRY(pl.partials.reserve(std::max<size_t>((size_t)n_waves * 2, 3 * 1024) * sizeof(double)));
    for (int i = 0; i < n_iter; ++i) {
      const bool last = i == n_iter - 1, ev = eval_last && last;
      int rc = SPECINV_OK;
      if (hopk) {
        SPECINV_R_SWITCH(R, if constexpr (RR <= 16) {
          if (td) rc = launch_hop_td<RR>(pl, ev, last || (eval_last && i == n_iter - 2));
          else if (mode == fast::MODE_GLA) rc = ev ? launch_hop<RR, fast::MODE_GLA, true>(pl) : launch_hop<RR, fast::MODE_GLA, false>(pl);
          else rc = ev ? launch_hop<RR, fast::MODE_ADMM, true>(pl, last) : launch_hop<RR, fast::MODE_ADMM, false>(pl, last);
        });
        SI_TRY(rc);
        continue;
      }
      SPECINV_R_SWITCH(R, if (mode == fast::MODE_GLA) rc = ev ? launch_semi<RR, fast::MODE_GLA, true>(pl)
                                                               : launch_semi<RR, fast::MODE_GLA, false>(pl);
                       else rc = ev ? launch_semi<RR, fast::MODE_ADMM, true>(pl, last) : launch_semi<RR, fast::MODE_ADMM, false>(pl, last));
      SI_TRY(rc);
    }
    n_partials = n_waves;
    return SPECINV_OK;
  }

  template <typename P>
  int iterate(P& pl, int n_iter, bool eval_last) {
    if (semi) return iterate_semi(pl, n_iter, eval_last);
    SI_TRY(pl.partials.reserve(std::max<size_t>((size_t)n_waves * 2 * fast::kEvalPieces, 3 * 1024) * sizeof(double)));
    int eval_pieces = 1;
    for (int i = 0; i < n_iter; ++i) {
      const bool ev = eval_last && i == n_iter - 1;
      const int nx = cur ^ 1;
      fast::FastArgs a{};
      a.x_in = xb[cur].template as<float>();
      a.x_out = xb[nx].template as<float>();
      a.xtail_in = xtail[cur].template as<float>();
      a.xtail_out = xtail[nx].template as<float>();
      // the spectral state of a frame is read and written by the same lane: it can live in one buffer
      const int ps = state_in_place ? 0 : cur, pn = state_in_place ? 0 : nx;
      a.P_in = Pb[ps].template as<v4f>();
      a.P_out = Pb[pn].template as<v4f>();
      a.Pmid_in = Pmid[ps].template as<v2f>();
      a.Pmid_out = Pmid[pn].template as<v2f>();
      SI_TRY(want_xu(pl, a, i == n_iter - 1));
      a.m_pairs = mpairs.template as<v4f>();
      a.m_mid = mmid.template as<float>();
      a.window = pl.window.template as<float>();
      a.inv_env = inv_env.template as<float>();
      a.partials = pl.partials.template as<double>();
      a.T = pl.Tn();
        a.nchunks = nchunks;
      a.skew = skew;
      a.n_waves = n_waves;
      a.pad_mode = pl.cfg.pad_mode;
      a.L = pl.length;
      a.coef = pl.coef;
      a.inv1p = 1.0f / (float)(1.0 + (double)pl.coef);
      a.fwd_scale = pl.fc.fwd_scale;
      a.inv_scale = pl.fc.inv_scale;
      if (td) {
        ++td_t;
        const double tds = std::pow(-(double)pl.coef, (double)td_t);
        const bool early = std::fabs(tds) >= 9.3132257461547852e-10;       // 2^-30: below float32 resolution of |pre|
        a.x_in = td_t == 1 ? xb[cur].template as<float>() : zb[cur].template as<float>();
        a.x_out = zb[nx].template as<float>();
        a.x2_in = xb[cur].template as<float>();
        // x_{t+1} has a reader only after the last iteration of a call (get_wave, the next call) and before an evaluating launch
        const bool need_x = i == n_iter - 1 || (eval_last && i == n_iter - 2);
        a.x2_out = need_x ? xb[nx].template as<float>() : nullptr;
        a.P_in = Pb[0].template as<v4f>();
        a.Pmid_in = Pmid[0].template as<v2f>();
        a.tds = (float)tds;
        // an evaluating iteration on the headline shapes: the plain kernel, then the evaluation of x_t as a kernel of its own
        // (kernels_fast_td.h: k_eval_td; SPECINV_EVAL_KERNEL=0: the fused evaluating variant)
        const char* eval_env = ev ? getenv("SPECINV_EVAL_KERNEL") : nullptr;
        const bool eval_kernel = !(eval_env && eval_env[0] == '0');
        if (ev && eval_kernel && OV == 4 && (R == 8 || R == 16)) {
          SI_TRY(launch_td(pl, a, early, false));
          const void* fn = R == 16 ? (const void*)fast::k_eval_td<16, 4> : (const void*)fast::k_eval_td<8, 4>;
          const size_t lds_used = R == 16 ? fast::Geo<16>::lds_bytes(4) : fast::Geo<8>::lds_bytes(4);
          SI_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_used));
          fast::FastArgs args = a;
          void* kargs[] = {&args};
          SI_HIP(hipLaunchKernel(fn, dim3((n_waves * fast::kEvalPieces + 3) / 4), dim3(256), kargs, lds_used, pl.stream));
          eval_pieces = fast::kEvalPieces;
          cur = nx;
          continue;
        }
        SI_TRY(launch_td(pl, a, early, ev));
        cur = nx;
        continue;
      }
      int rc = SPECINV_OK;
      SPECINV_R_SWITCH(R, if (mode == fast::MODE_GLA) rc = ev ? launch<RR, fast::MODE_GLA, true>(pl, a)
                                                               : launch<RR, fast::MODE_GLA, false>(pl, a);
                       else rc = ev ? launch<RR, fast::MODE_ADMM, true>(pl, a) : launch<RR, fast::MODE_ADMM, false>(pl, a));
      SI_TRY(rc);
      cur = nx;
    }
    n_partials = n_waves * eval_pieces;
    return SPECINV_OK;
  }

  template <typename P>
  int get_wave(P& pl, float* out) {
    SI_HIP(hipMemcpyAsync(out, xb[cur].p, (size_t)pl.B() * pl.length * sizeof(float), hipMemcpyDeviceToDevice, pl.stream));
    if (!semi && nchunks > 1) {
      const int hop = pl.cfg.hop_length;
      const long long total = (long long)pl.B() * nchunks * (OV - 1) * hop;
      const void* fn = nullptr;
      SPECINV_R_SWITCH(R, if constexpr (RR % 8 == 0) { if (OV == 8) fn = (const void*)fast::k_add_tails<RR, 8>; }
                       if (OV == 4) fn = (const void*)fast::k_add_tails<RR, 4>;
                       if (OV == 2) fn = (const void*)fast::k_add_tails<RR, 2>);
      float* xo = out;
      const float* tl = xtail[cur].template as<float>();
      int Tn = pl.Tn(), nc = nchunks;
      long long Ln = (long long)pl.length, tot = total;
      int sk = skew;
      void* kargs[] = {&xo, &tl, &Tn, &nc, &Ln, &tot, &sk};
      SI_HIP(hipLaunchKernel(fn, dim3((unsigned)ceil_div(total, 256)), dim3(256), kargs, 0, pl.stream));
    }
    return SPECINV_OK;
  }

  template <typename P>
  int get_state_spec(P& pl, int which, cplx<float>* out) {
    const long long nf = (long long)pl.B() * pl.Tn();
    SI_TRY(scratch.reserve((size_t)nf * pl.n_freq * sizeof(v2f)));
    const int ps = (semi || state_in_place) ? 0 : cur;
    const bool admm = mode == fast::MODE_ADMM;
    SI_CHECK(admm || !td || td_t == 0, SPECINV_ESTATE,
             "Griffin-Lim carries its momentum as a signal on this path (pre_spec is never formed); call "
             "specinv_plan_keep_state(plan, 1) before specinv_gla_init to iterate on pre_spec itself");
    SI_CHECK(!admm || which == 2 || xu_valid, SPECINV_ESTATE,
             "ADMM carries Y = X + U; call specinv_plan_keep_state(plan, 1) before iterating to read X and U (which = 2 reads Y)");
    const FastBuf& src = (!admm || which == 2) ? Pb[ps] : which == 0 ? Xb : Ub;
    const FastBuf& mid = (!admm || which == 2) ? Pmid[ps] : which == 0 ? Xmid : Umid;
    SI_CHECK(!two || !admm || which == 2, SPECINV_EUNSUPPORTED, "X and U of a two-sided ADMM run are not kept on this path");
    SPECINV_R_SWITCH(R, const long long np = nf * fast::Geo<RR>::H * 64;
                     hipLaunchKernelGGL((fast::k_pairs_to_spec<RR>), dim3((unsigned)ceil_div(np, 256)), dim3(256), 0, pl.stream,
                                        src.template as<v4f>(), mid.template as<v2f>(), scratch.template as<v2f>(), nf, pl.n_freq, 0);
                     if (two) hipLaunchKernelGGL((fast::k_pairs_to_spec<RR>), dim3((unsigned)ceil_div(np, 256)), dim3(256), 0,
                                                 pl.stream, Pb2.template as<v4f>(), Pmid2.template as<v2f>(),
                                                 scratch.template as<v2f>(), nf, pl.n_freq, 1));
    SI_HIP(hipGetLastError());
    return pl.template transpose<cplx<float>>(scratch.template as<cplx<float>>(), out, pl.Tn(), pl.n_freq);
  }
};

}  // namespace specinv
