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
  // device-resident optimiser (lbfgs_dev.h): run only if *ctl_eval != 0; the gradient goes to (*ctl_cur ^ 1 ? grad_alt : grad)
  const int* ctl_eval;
  const int* ctl_cur;
  float* grad_alt;
#if SPECINV_OBJ_STAMPS
  unsigned long long* stamps;   // [tiles][16]
#endif
};

// what lbfgs_dev.h hands to the objective: the gate and the gradient ping-pong of the optimiser's state record
struct ObjCtl {
  const int* do_eval;
  const int* cur;
  float* grad_alt;
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
template <int R, int MT, bool MAG = false>
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

}  // namespace fast
}  // namespace specinv
