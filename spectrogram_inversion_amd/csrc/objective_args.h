// The one-launch L-BFGS objective (kernels_objective.h): argument record, block table, LDS geometry and the host-side block builder.
#pragma once
#include <algorithm>
#include <vector>

#include "fast_core.h"

namespace specinv {
namespace fast {

using f32x4 = float __attribute__((ext_vector_type(4)));

#ifndef SPECINV_OBJ_STAMPS        // diagnostic build: s_memtime at the phase boundaries of one wave of every workgroup
#define SPECINV_OBJ_STAMPS 0      // (tools/obj_stamps.py; the shipped kernel executes no stamp)
#endif
#ifndef SPECINV_OBJ_STAMP_WAVE    // ... which wave (0 .. 7; 0 - 3 are the older waves of their SIMDs, 4 - 7 the younger ones)
#define SPECINV_OBJ_STAMP_WAVE 0
#endif

constexpr int kObjWaves = 8;      // waves per workgroup
constexpr int kObjTile = 16;      // frames per tile (the N of the MFMA), two per wave
constexpr int kObjRow = 16;       // floats per row of the [bin][frame] tiles; elements sit at obj_at(row, frame)

// The [rows][16 frames] float tiles in LDS are stored QUAD-MAJOR: the four rows of an aligned group of four sit side by side as
// one 16-byte unit per frame, [row >> 2][frame ^ swizzle][row & 3], with the frame index XOR-ed by the quad's low four bits.
//   * the matrix cores take their operands and leave their results in exactly these units: v_mfma_f32_16x16x4_f32's lane
//     (i = lane >> 4, c = lane & 15) supplies B[k][c] of k-step j from row 4 i + j of the 16-row group (obj_build_blocks orders the
//     filterbank operand to match) - ONE ds_read_b128 per block instead of four ds_read_b32 - and holds D rows 4 i .. 4 i + 3 of
//     column c: one ds_write_b128.  The 64 lanes cover 64 consecutive units (the XOR permutes them inside a row): no conflict;
//   * the FFT's lanes - 64 consecutive rows at one frame - write |S| and read dA one float each: lanes 4 q .. 4 q + 3 share a
//     unit, the XOR spreads the 16 quads over the 16 units of a row of 64 banks: no conflict either.
// (Round 3 used rows of 17 floats: conflict-free for the FFT's lanes, but the operand read - 4 rows x 16 frames - put six of its
// 64 lanes on three banks: SQ_LDS_BANK_CONFLICT 14 % of SQ_LDS_IDX_ACTIVE, profiles/r03_C5_pmc.json.)
__device__ __forceinline__ int obj_quad(int q, int n) { return (q << 4) + (n ^ (q & 15)); }       // in 16-byte units
__device__ __forceinline__ int obj_at(int f, int n) { return obj_quad(f >> 2, n) * 4 + (f & 3); }   // in floats

struct ObjArgs {
  const float* x;          // (B, len)
  float* grad;             // (B, len)
  float* margins;          // (B, 2, pad): gradient w.r.t. the padded samples either side of the signal
  float* xtail;            // (B, nchunks, n_fft - hop)
  const float* target;     // (B, n_mels, T), the caller's layout
  const f32x4* melA;       // forward operand blocks  [E][64] x 4 k-steps: A[i = mel row][k = bin]
  const f32x4* melB;       // backward operand blocks [E][64] x 4 k-steps: A[i = bin][k = mel row]
  const int* tab;          // block list, see ObjTab
  const float* window;
  double* partials;        // [B * nchunks] squared-error sums
  long long len;
  int T, nchunks, hop, pad, pad_mode, n_mels;
  float fwd_scale;
  float dscale;            // 2 / numel
  unsigned hop_magic;      // ceil(2^32 / hop) (hop > 1), for the division-free frame lookup of the overlap-add
  // band form of a sparse filterbank (SP kernels; obj_build_sparse): melA is the blob the workgroup stages into LDS, in 16-byte
  // units [0, sp_rm) row weights, [sp_rm, sp_cm) row records, [sp_cm, sp_cw) column records, [sp_cw, sp_total) column weights;
  // tab holds the waves' lists of row quads
  int sp_rm, sp_cm, sp_cw, sp_total, sp_cmax, sp_rows;
  // device-resident optimiser (lbfgs_dev.h): run only if *ctl_eval != 0; the gradient goes to (*ctl_cur ^ 1 ? grad_alt : grad)
  const int* ctl_eval;
  const int* ctl_cur;
  float* grad_alt;
#if SPECINV_OBJ_STAMPS
  unsigned long long* stamps;   // [tiles][16]
#endif
};

// what lbfgs_dev.h hands to the objective: the gate and the gradient ping-pong of the optimiser's state record; for the frame walk
// also the iterate's two buffers and the deferred step (ObjWalkArgs)
struct ObjCtl {
  const int* do_eval;
  const int* cur;
  float* grad_alt;
  float* x_alt = nullptr;
  const int* x_sel = nullptr;
  const int* x_pending = nullptr;
  const double* t_pend = nullptr;
  const double* c0_pend = nullptr;
};

// the two-launch lean iteration of the device-resident optimiser (lbfgs_state.h: lbd_tail_decide): the epilogue's workgroup that
// finishes last takes iteration k's decisions.  ticket == nullptr: not asked for.
struct ObjDecide {
  unsigned* ticket = nullptr;   // workgroups done (reset by the last one)
  const void* st = nullptr;     // LbdState the evaluation ran under
  void* st_next = nullptr;      // LbdState the decision writes
  double* board = nullptr;
  const double* rows = nullptr;
  int k = 0;
  int fence = 0;                // hand the rows over with the memory model's own release / acquire (k_objective_epilogue's tail)
};

// the statistics of the evaluated gradient: what the caller hands in, and where the figures go.  d / gp == nullptr: the gradient
// itself stands in (d = g before the first iteration, lbfgs.py:_batch).  Device-resident optimiser: *have == 0 -> d = g_prev = g,
// else d as given and g_prev = the gradient buffer this evaluation does NOT write; t = *t_dev.
constexpr int kObjStatRow = 9;    // a row of the epilogue's tree: the eight + the squared-error sum
constexpr int kObjRows = 256;     // rows the epilogue leaves (one per block of its grid)
struct ObjStatReq {
  const float* d = nullptr;       // direction (nullptr: d = g)
  const float* gp = nullptr;      // previous gradient (nullptr: g_prev = g)
  float t = 0.0f;
  const int* have = nullptr;      // device-resident optimiser: see above
  const double* t_dev = nullptr;
  // ... whose direction may be IMPLICIT (lbfgs_dev.h, lean iterations): *d_implicit != 0 -> d = (float)(*c0_d * (double)g_prev), the
  // float operations that formed it from the gradient it was formed from - nobody stores or reads d
  const int* d_implicit = nullptr;
  const double* c0_d = nullptr;
  double* rows = nullptr;         // [kObjRows][kObjStatRow]: out
};

// Statistics of the gradient an evaluation produces (ObjStatReq): the sums k_lbfgs_pair_stats takes over g, g_prev, d - the same
// float operations for y = g - g_prev and s = t d, products and sums in float64 - accumulated per thread by k_objective_epilogue,
// whose pass over the seams becomes a pass over the whole gradient.
struct ObjStatAcc {
  double s[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  float mg = 0.0f, md = 0.0f;
  __device__ __forceinline__ void add(float g, float gp, float d, float t) {
    const float y = g - gp, sv = t * d;
    const double g64 = (double)g;
    s[0] += g64 * (double)d;
    s[1] += fabs(g64);
    s[2] += (double)y * (double)sv;
    s[3] += (double)y * (double)y;
    s[4] += g64 * g64;
    s[5] += g64 * (double)gp;
    mg = fmaxf(mg, fabsf(g));
    md = fmaxf(md, fabsf(d));
  }
};

// Layout of the block table (ints): the non-zero blocks are sorted by bin group, then mel group.
//   [0 .. 8]            forward: wave w takes blocks [tab[w], tab[w+1])
//   [9 .. 17]           backward: wave w takes bin groups [tab[9+w], tab[9+w+1])  (balanced by block count)
//   [18 .. 18+KQ]       first block of bin group g (KQ + 1 entries)
//   [19+KQ .. +E)       mel group of block e
//   [19+KQ+E .. +E)     bin group of block e
struct ObjTab {
  static constexpr int FWD = 0, BWD = 9, BEGIN = 18;
  static constexpr int mel_group(int KQ) { return 19 + KQ; }
  static constexpr int bin_group(int KQ, int E) { return 19 + KQ + E; }
};

template <int R, int MT>
struct ObjGeo {
  using G = Geo<R>;
  static constexpr int F = G::M + 1;
  static constexpr int KQ = (F + 15) / 16;            // groups of 16 bins
  static constexpr int FP = 16 * KQ;                  // rows of the |S| / dA tile
  static constexpr int UNI_TR = kObjWaves * G::TR * 2;            // floats: FFT transpose scratch of the waves
  static constexpr int UNI_RED = kObjWaves * MT * 4 * 64;         // floats: partial accumulators of the forward contraction
  static constexpr int UNI = UNI_TR > UNI_RED ? UNI_TR : UNI_RED;
  static constexpr size_t lds_bytes() {
    return sizeof(v2f) * G::M + sizeof(float) * ((size_t)FP * kObjRow + 16 * MT * kObjRow + UNI);
  }
};

// defined in kernels_objective.h, compiled in tu_objective.hip
template <int R, int MT, bool MAG = false, bool SP = false>
__global__ void k_objective_logmel(ObjArgs a);

// Host side: cut the filterbank (n_mels x F, row-major, host copy) into 16 x 16 blocks, keep the non-zero ones in
// operand order and build the block table (ObjTab).
inline void obj_build_blocks(const float* mel, int F, int n_mels, int MT, std::vector<float>& A, std::vector<float>& B,
                             std::vector<int>& tab) {
  const int KQ = (F + 15) / 16;
  std::vector<int> begin(KQ + 1, 0), mgs, fgs;
  for (int fg = 0; fg < KQ; ++fg) {
    begin[fg] = (int)mgs.size();
    for (int mg = 0; mg < MT; ++mg) {
      bool any = false;
      for (int m = 16 * mg; m < std::min(n_mels, 16 * mg + 16) && !any; ++m)
        for (int f = 16 * fg; f < std::min(F, 16 * fg + 16); ++f)
          if (mel[(size_t)m * F + f] != 0.0f) {
            any = true;
            break;
          }
      if (any) {
        mgs.push_back(mg);
        fgs.push_back(fg);
      }
    }
  }
  if (mgs.empty()) {                       // an all-zero filterbank still needs one (zero) block to point the loads at
    mgs.push_back(0);
    fgs.push_back(0);
    for (int fg = 1; fg <= KQ; ++fg) begin[fg] = 1;
  }
  const int E = (int)mgs.size();
  begin[KQ] = E;
  auto at = [&](int m, int f) { return (m < n_mels && f < F) ? mel[(size_t)m * F + f] : 0.0f; };
  A.assign((size_t)E * 256, 0.0f);
  B.assign((size_t)E * 256, 0.0f);
  for (int e = 0; e < E; ++e)
    for (int lane = 0; lane < 64; ++lane)
      for (int j = 0; j < 4; ++j) {
        // k-step j pairs lane-row i = lane >> 4 with row 4 i + j of the 16-row group: the quad the B operand's lane reads in one piece
        // (obj_quad).  forward: A[i = mel row = lane & 15][k = bin]; backward: A[i = bin][k = mel row]
        A[((size_t)e * 64 + lane) * 4 + j] = at(16 * mgs[e] + (lane & 15), 16 * fgs[e] + 4 * (lane >> 4) + j);
        B[((size_t)e * 64 + lane) * 4 + j] = at(16 * mgs[e] + 4 * (lane >> 4) + j, 16 * fgs[e] + (lane & 15));
      }
  tab.assign(ObjTab::bin_group(KQ, E) + E, 0);
  for (int w = 0; w <= kObjWaves; ++w) tab[ObjTab::FWD + w] = (int)((long long)E * w / kObjWaves);
  for (int w = 0; w <= kObjWaves; ++w) {   // first bin group that starts at or after block E w / 8
    const int want = (int)((long long)E * w / kObjWaves);
    int g = 0;
    while (g < KQ && begin[g] < want) ++g;
    tab[ObjTab::BWD + w] = w == kObjWaves ? KQ : g;
  }
  tab[ObjTab::BWD] = 0;
  for (int g = 0; g <= KQ; ++g) tab[ObjTab::BEGIN + g] = begin[g];
  for (int e = 0; e < E; ++e) {
    tab[ObjTab::mel_group(KQ) + e] = mgs[e];
    tab[ObjTab::bin_group(KQ, E) + e] = fgs[e];
  }
}

// ---- band form of a sparse filterbank -------------------------------------------------------------------------------------
// A mel filterbank is triangles around a diagonal: ~2 non-zeros per bin (2 F of n_mels x F), 11 % of the entries of its non-zero
// 16 x 16 blocks.  The SP kernels contract it on the vector units instead, as bands:
//   forward   row m = its weights over the bin quads [q0, q0 + len) (16-byte units: four consecutive bins, zero outside the row's
//             support); lane (r, n) of a wave runs row 4 g + r of row quad g at frame n, four rows of a quad side by side, each
//             padded to the quad's longest row; the waves take row quads from lists balanced by length (ObjSp::LIST);
//   backward  bin f = the rows [m0, m0 + cmax) that meet it (zero weights beyond its own count), per bin quad one record of four
//             16-bit offsets of row m0 in the dM tile and cmax 16-byte units of weights.
// Every weight is the matrix entry itself, so the sums hold exactly the non-zero products of the dense contraction (in band
// order).  Eligible: at most 4 rows per bin, <= 255 rows, the blob fits the workgroup's staging registers and the FFT scratch.
struct ObjSp {
  static constexpr int MAXQ = 16;                       // row quads per wave
  static constexpr int STAGE = 6;                       // 16-byte units a thread stages: blob <= STAGE * 512 units
  static constexpr int COUNT = 8, LIST = 16;            // tab[COUNT + w] row quads of wave w; tab[LIST + MAXQ w + i] = g | len << 8
};
struct ObjSparseInfo {
  int rm = 0, cm = 0, cw = 0, total = 0, cmax = 0, rows = 0;
};

inline bool obj_build_sparse(const float* mel, int F, int n_mels, int uni_floats, std::vector<float>& blob, std::vector<int>& tab,
                             ObjSparseInfo& inf) {
  const int KQ = (F + 15) / 16, FP = 16 * KQ, NBQ = FP / 4, QG = (n_mels + 3) / 4, rows = 4 * QG, W = kObjWaves;
  if (rows + 3 > 16 * 9) return false;            // (the dM tile of k_objective_logmel<R, 9>; rows m0 + j beyond it meet zero weights only)
  auto at = [&](int m, int f) { return (m < n_mels && f < F) ? mel[(size_t)m * F + f] : 0.0f; };
  std::vector<int> q0(rows, 0), nq(rows, 0), m0(FP, 0), cnt(FP, 0);
  for (int m = 0; m < n_mels; ++m) {
    int lo = -1, hi = -1;
    for (int f = 0; f < F; ++f)
      if (mel[(size_t)m * F + f] != 0.0f) {
        if (lo < 0) lo = f;
        hi = f;
      }
    if (lo >= 0) {
      q0[m] = lo / 4;
      nq[m] = hi / 4 - lo / 4 + 1;
    }
  }
  int cmax = 1;
  for (int f = 0; f < F; ++f) {
    int lo = -1, hi = -1;
    for (int m = 0; m < n_mels; ++m)
      if (mel[(size_t)m * F + f] != 0.0f) {
        if (lo < 0) lo = m;
        hi = m;
      }
    if (lo >= 0) {
      m0[f] = lo;
      cnt[f] = hi - lo + 1;
      cmax = std::max(cmax, cnt[f]);
    }
  }
  if (cmax > 4) return false;
  cmax = cmax <= 2 ? 2 : 4;                   // (the kernel has these two; the padding rows carry zero weights)
  std::vector<int> glen(QG, 1);
  for (int g = 0; g < QG; ++g) {
    for (int r = 0; r < 4; ++r) glen[g] = std::max(glen[g], nq[4 * g + r]);
    glen[g] = std::min((glen[g] + 3) & ~3, NBQ);      // the kernel walks a band four quads at a time (NBQ is a multiple of 4)
  }
  for (int g = 0; g < QG; ++g)
    if (glen[g] > 0xffff) return false;
  // longest-first onto the least loaded wave (a quad costs its length + the log1p / division of its four rows)
  std::vector<int> order(QG);
  for (int g = 0; g < QG; ++g) order[g] = g;
  std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return glen[x] > glen[y]; });
  std::vector<std::vector<int>> lists(W);
  std::vector<long long> load(W, 0);
  for (int g : order) {
    int w = 0;
    for (int v = 1; v < W; ++v)
      if (load[v] < load[w]) w = v;
    lists[w].push_back(g);
    load[w] += glen[g] + 6;
  }
  for (int w = 0; w < W; ++w)
    if ((int)lists[w].size() > ObjSp::MAXQ) return false;
  // blob: row weights | row records (ptr, q0) | column records | column weights
  std::vector<int> ptr(rows, 0);
  int n_rw = 0;
  for (int g = 0; g < QG; ++g)
    for (int r = 0; r < 4; ++r) {
      ptr[4 * g + r] = n_rw;
      n_rw += glen[g];
    }
  // a band never leaves the tile: a short row of a long quad starts early instead (the weights are the matrix entries: zeros)
  for (int m = 0; m < rows; ++m) q0[m] = std::min(q0[m], NBQ - glen[m / 4]);
  inf.rm = n_rw;
  inf.cm = inf.rm + (2 * rows + 3) / 4;
  inf.cw = inf.cm + (2 * NBQ + 3) / 4;
  inf.total = inf.cw + NBQ * cmax;
  inf.cmax = cmax;
  inf.rows = rows;
  if (inf.total > ObjSp::STAGE * 64 * kObjWaves || (long long)inf.total * 4 > uni_floats) return false;
  blob.assign((size_t)inf.total * 4, 0.0f);
  for (int g = 0; g < QG; ++g)
    for (int r = 0; r < 4; ++r) {
      const int m = 4 * g + r;
      for (int t = 0; t < glen[g]; ++t)
        for (int c = 0; c < 4; ++c) blob[((size_t)ptr[m] + t) * 4 + c] = at(m, 4 * (q0[m] + t) + c);
    }
  int* rec = reinterpret_cast<int*>(blob.data() + (size_t)inf.rm * 4);
  for (int m = 0; m < rows; ++m) {
    rec[2 * m] = ptr[m];
    rec[2 * m + 1] = q0[m];
  }
  int* crec = reinterpret_cast<int*>(blob.data() + (size_t)inf.cm * 4);
  for (int q = 0; q < NBQ; ++q) {
    // byte offsets of row m0 in the plain [row][16 frames] dM tile, 16 bits each; row m0 + j: + 64 j (the instruction's offset)
    crec[2 * q] = (m0[4 * q] * 64) | ((m0[4 * q + 1] * 64) << 16);
    crec[2 * q + 1] = (m0[4 * q + 2] * 64) | ((m0[4 * q + 3] * 64) << 16);
    for (int j = 0; j < cmax; ++j)
      for (int c = 0; c < 4; ++c) {
        const int f = 4 * q + c;
        blob[((size_t)inf.cw + (size_t)q * cmax + j) * 4 + c] = j < cnt[f] ? at(m0[f] + j, f) : 0.0f;
      }
  }
  tab.assign(ObjSp::LIST + ObjSp::MAXQ * W, 0);
  for (int w = 0; w < W; ++w) {
    tab[ObjSp::COUNT + w] = (int)lists[w].size();
    for (size_t i = 0; i < lists[w].size(); ++i) tab[ObjSp::LIST + ObjSp::MAXQ * w + (int)i] = lists[w][i] | (glen[lists[w][i]] << 8);
  }
  return true;
}


// ---- the objective as a frame walk (kernels_objective_walk.h: k_objective_walk) ------------------------------------------------
// The kernel of BASELINE configs[4] rebuilt in the Griffin-Lim kernel's form (round 5): a wave walks a chunk of consecutive frames
// of one item - sample window carried in registers, analysis, the filterbank contractions on the frame's OWN spectrum, synthesis,
// overlap-add in registers - and no workgroup barrier separates anything: while one wave of a SIMD contracts (LDS) the other
// transforms (vector units).  The filterbank in two tables, both staged to LDS once per workgroup:
//   forward   tasks: the band of row m (bin quads [q0, q0 + len) of the frame's |S| column) cut into 1, 2, 4, 8 or 16 segments of at
//             most kObjWalkSeg quads, a lane per segment, the segments of a row in adjacent lanes (summed with xor shuffles);
//             rows with equal segment counts share a pass of 64 lanes.  Task = {weights (16-byte units), first quad, quads, row |
//             leader << 16}; pass p: tasks [64 p, 64 p + 64), log2(segments) in `pass_gs`.
//   backward  in the lanes' own conjugate-pair order: for pair j of lane l (bins k = l + 64 j and M - k) the two weights of each
//             bin (a mel filterbank meets a bin with at most two rows) and their first rows.
constexpr int kObjWalkSeg = 4;          // quads per forward task
constexpr int kObjWalkMaxPass = 16;
struct ObjWalkInfo {
  int rows = 0, n_pass = 0;
  int task_off = 0, bw_off = 0, bm_off = 0, mid_off = 0, total = 0;   // offsets into the blob, in 16-byte units ([0, task_off): weights)
  int pass_gs[kObjWalkMaxPass] = {};
};
struct ObjWalkArgs {
  const float* x;          // (B, len), len = (T - 1) hop
  float* grad;
  float* margins;          // (B, 2, pad)
  float* xtail;            // (B, nchunks, n_fft - hop)
  const float* target;     // (B, n_mels, T)
  const f32x4* blob;
  const float* window;
  double* partials;        // [n_waves] squared-error sums
  long long len;
  int T, nchunks, n_waves, skew, pad_mode, n_mels;
  float fwd_scale, dscale;
  ObjWalkInfo w;
  const int* ctl_eval;     // device-resident optimiser: as ObjArgs
  const int* ctl_cur;
  float* grad_alt;
  // ... and its DEFERRED STEP (lbfgs_dev.h): the iterate lives in one of two buffers, x and x_alt; *px_sel names the one this
  // evaluation is taken at.  *px_pending != 0: that buffer is not written yet - it is the other one advanced by the last
  // iteration's step, x_new = fma(t, (float)(c0 (double)g_prev), x_old), g_prev the gradient buffer this evaluation does NOT write -
  // and the walk forms it while it loads its samples and writes every hop-block it owns (the optimiser's direction kernel then
  // has nothing to stream: 100 MB per iteration less)
  float* x_alt;
  const int* px_sel;
  const int* px_pending;
  const double* pt_pend;
  const double* pc0_pend;
};
template <int R, int OV>
__global__ void k_objective_walk(ObjWalkArgs a);
constexpr int kWalkWaves = 8;     // waves per workgroup, one workgroup per CU (two waves per SIMD)
constexpr int kWalkMM = 144;      // floats of a wave's mm / dM vector (rows <= 128, + 1 zero behind the last row)

template <int R>
constexpr size_t obj_walk_lds_bytes(int blob_units) {
  return sizeof(v2f) * (size_t)(Geo<R>::M + (R - 1) * 64) + 16 * (size_t)blob_units +
         (size_t)kWalkWaves * (sizeof(v2f) * Geo<R>::TR + sizeof(float) * kWalkMM);
}


// Host side: the two tables.  false: not a filterbank this kernel takes (more than two rows on a bin, rows that are not adjacent,
// too many rows or passes) - the tile kernel serves it.
inline bool obj_build_walk(const float* mel, int F, int n_mels, int R, std::vector<float>& blob, ObjWalkInfo& inf) {
  const int M = 64 * R, H = R / 2, NBQ = (F + 3) / 4;
  if (F != M + 1 || n_mels < 1 || n_mels > 128) return false;
  auto at = [&](int m, int f) { return (m >= 0 && m < n_mels && f < F) ? mel[(size_t)m * F + f] : 0.0f; };
  // rows of a bin: at most two, adjacent
  std::vector<int> m0(F, 0);
  for (int f = 0; f < F; ++f) {
    int lo = -1, hi = -1;
    for (int m = 0; m < n_mels; ++m)
      if (mel[(size_t)m * F + f] != 0.0f) {
        if (lo < 0) lo = m;
        hi = m;
      }
    if (lo >= 0) {
      if (hi - lo > 1) return false;
      m0[f] = lo;
    }
  }
  // forward tasks
  struct Row { int m, q0, len, nseg; };
  std::vector<Row> rows;
  for (int m = 0; m < n_mels; ++m) {
    int lo = -1, hi = -1;
    for (int f = 0; f < F; ++f)
      if (mel[(size_t)m * F + f] != 0.0f) {
        if (lo < 0) lo = f;
        hi = f;
      }
    Row r{m, 0, 0, 1};
    if (lo >= 0) {
      r.q0 = lo / 4;
      r.len = hi / 4 - lo / 4 + 1;
      int need = (r.len + kObjWalkSeg - 1) / kObjWalkSeg;
      while (r.nseg < need && r.nseg < 16) r.nseg *= 2;
    }
    rows.push_back(r);
  }
  std::stable_sort(rows.begin(), rows.end(), [](const Row& x, const Row& y) { return x.nseg > y.nseg; });
  std::vector<float> weights;
  std::vector<int> tasks;          // 4 ints each
  int n_pass = 0;
  size_t i = 0;
  while (i < rows.size()) {
    const int nseg = rows[i].nseg, per = 64 / nseg;
    if (n_pass == kObjWalkMaxPass) return false;
    int gs = 0;
    while ((1 << gs) < nseg) ++gs;
    inf.pass_gs[n_pass] = gs;
    int lane = 0;
    for (int r = 0; r < per && i < rows.size() && rows[i].nseg == nseg; ++r, ++i) {
      const Row& row = rows[i];
      const int seglen = (row.len + nseg - 1) / nseg;
      for (int sg = 0; sg < nseg; ++sg, ++lane) {
        const int b = std::min(row.len, sg * seglen), e = std::min(row.len, (sg + 1) * seglen);
        tasks.push_back((int)(weights.size() / 4));
        tasks.push_back(row.q0 + b);
        tasks.push_back(e - b);
        tasks.push_back(row.m | (sg == 0 ? 0x10000 : 0));
        for (int q = row.q0 + b; q < row.q0 + e; ++q)
          for (int c = 0; c < 4; ++c) weights.push_back(at(row.m, 4 * q + c));
      }
    }
    for (; lane < 64; ++lane) {          // idle lanes of the pass
      tasks.push_back(0);
      tasks.push_back(0);
      tasks.push_back(0);
      tasks.push_back(0x7fff);           // (no row: never a leader)
    }
    ++n_pass;
  }
  (void)NBQ;
  inf.rows = n_mels;
  inf.n_pass = n_pass;
  inf.task_off = (int)(weights.size() / 4);
  inf.bw_off = inf.task_off + n_pass * 64;
  inf.bm_off = inf.bw_off + H * 64;
  inf.mid_off = inf.bm_off + (H * 64 + 3) / 4;
  inf.total = inf.mid_off + 2;
  blob.assign((size_t)inf.total * 4, 0.0f);
  std::copy(weights.begin(), weights.end(), blob.begin());
  int* tk = reinterpret_cast<int*>(blob.data() + (size_t)inf.task_off * 4);
  std::copy(tasks.begin(), tasks.end(), tk);
  float* bw = blob.data() + (size_t)inf.bw_off * 4;
  int* bm = reinterpret_cast<int*>(blob.data() + (size_t)inf.bm_off * 4);
  for (int j = 0; j < H; ++j)
    for (int l = 0; l < 64; ++l) {
      const int k = l + 64 * j, kk = M - k;
      float* w = bw + ((size_t)j * 64 + l) * 4;
      w[0] = at(m0[k], k);
      w[1] = at(m0[k] + 1, k);
      w[2] = at(m0[kk], kk);
      w[3] = at(m0[kk] + 1, kk);
      bm[j * 64 + l] = m0[k] | (m0[kk] << 16);
    }
  float* md = blob.data() + (size_t)inf.mid_off * 4;
  md[0] = at(m0[M / 2], M / 2);
  md[1] = at(m0[M / 2] + 1, M / 2);
  reinterpret_cast<int*>(md)[4] = m0[M / 2];
  return true;
}

}  // namespace fast
}  // namespace specinv
