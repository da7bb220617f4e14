// The L-BFGS objective of the log-mel path in ONE launch (reference: the closure of torch_specinv/methods.py:545-550 with
// transform_fn = log1p(mel_fb @ |stft(x)|), BASELINE.json configs[4]):
//
//     loss = mean((log1p(Mel |STFT(x)|) - target)^2)   and   d loss / d x
//
// One 8-wave workgroup owns a tile of up to 16 consecutive frames of one batch item; the spectrum never leaves the chip.
//   1. each wave transforms two frames with the wave-level FFT of fast_core.h, writes |S| into an LDS tile
//      [bin][frame] and keeps the unit phases S/|S| in registers;
//   2. forward mel contraction on the matrix cores (v_mfma_f32_16x16x4_f32: exact float32, an fmaf chain): the eight
//      waves split the 1025 bins (K), partial accumulators are added through LDS in a fixed order;
//   3. log1p, squared error (float64 partial sum per tile), dM = 2/numel (V - T) / (1 + Mel|S|) -> LDS;
//   4. backward contraction dA = Mel^T dM on the matrix cores, the waves split the bin tiles; dA overwrites the |S| tile;
//   5. each wave forms G = dA S/|S| (interior bins halved: the Hermitian extension) for its two frames, runs the inverse
//      FFT and applies the window;
//   6. the windowed frames are staged in LDS (the tiles are free by then; with hop = n_fft/2, /4, /8 a wave's two frames are
//      added in registers first) and every output sample gathers what covers it in frame order; the span's first frames*hop
//      samples go to the gradient, the remaining n_fft - hop (what the tile adds to its successor's samples) to `xtail`;
//      k_objective_epilogue finishes the seams and the padding.
// HBM traffic per frame: 4*hop read + 4*hop written + 4*n_mels target (+ the seams, + the filterbank, which every
// workgroup streams from L2): SURVEY 8d's 8h + 4 n_mels.
// Matrix cores (any filterbank matrix): it is cut into 16 x 16 blocks (16 mel rows x 16 bins) stored in MFMA operand order, one
// copy per contraction: one 16-byte load per lane feeds four MFMAs.  Blocks that are entirely zero are left out of the block list
// (obj_build_blocks): adding 0 * x changes nothing, so the result is bit for bit that of the dense contraction, and a mel
// filterbank - triangles around the diagonal - keeps ~1/3 of its blocks.  A dense matrix keeps all of them.
// Bands (SP, a sparse matrix - what a mel filterbank is): 2 F non-zeros are 11 % of the entries of those blocks, and float32 MFMA
// has the vector units' rate; steps 2 - 4 run on the vector units over the rows' / bins' bands instead (objective_args.h:
// obj_build_sparse), a third of the cycles.
#pragma once
#include <type_traits>

#include "objective_args.h"

namespace specinv {
namespace fast {


__device__ __forceinline__ f32x4 mfma_16x16x4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

#if SPECINV_OBJ_STAMPS
#define OBJ_STAMP(i) do { if (threadIdx.x == 64 * SPECINV_OBJ_STAMP_WAVE) a.stamps[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define OBJ_STAMP(i) do { } while (0)
#endif


// Step 3 for one output: d = log1p(v) - target, the squared error, dM = dscale d / (1 + v)  (v = (Mel |S|)[m, n] >= 0).
//   log1p(v) = log(u) + (v - (u - 1)) / u with u = fl(1 + v): the second term restores what the rounding of 1 + v lost (a third of
//   log1pf's instructions, the same result to an ulp); 1 / u: v_rcp_f32 and one Newton step.
__device__ __forceinline__ float obj_point(float v, float target, float dscale, double& s2) {
  const float u = 1.0f + v;
  float ru = fast_rcp(u);
  ru = fmaf(fmaf(-u, ru, 1.0f), ru, ru);
  const float d = (logf(u) + (v - (u - 1.0f)) * ru) - target;
  s2 += (double)d * (double)d;
  return (dscale * d) * ru;
}

// MAG: the magnitude objective mean((|STFT(x)| - target)^2) (`MagSTFT`, the reference's test / demo transform,
// test/test_lbfgs.py:17-18, main.py:21-43): the same kernel without the contractions - dA = 2/numel (|S| - T) is formed
// element by element on the |S| tile (MT only sizes the shared scratch then).
// SP: the filterbank in band form (objective_args.h: obj_build_sparse): steps 2 - 4 on the vector units, no partial sums - a lane
// owns whole rows (forward, log1p and dM in one pass) and whole bin quads (backward); the bands are staged into the FFT scratch,
// which idles between the transforms.
template <int R, int MT, bool MAG, bool SP>
__global__ __launch_bounds__(64 * kObjWaves, 2) void k_objective_logmel(ObjArgs a) {
  using G = Geo<R>;
  using OG = ObjGeo<R, MT>;
  constexpr int H = G::H, M = G::M, N = G::N, F = OG::F, KQ = OG::KQ, FP = OG::FP, RS = kObjRow;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  v2f* lds_win = reinterpret_cast<v2f*>(smem);
  float* tile = reinterpret_cast<float*>(lds_win + M);
  float* dmt = tile + FP * RS;
  float* uni = dmt + 16 * MT * RS;
  __shared__ double lsum[kObjWaves];
  if (a.ctl_eval != nullptr && *a.ctl_eval == 0) return;     // (the optimiser has stopped: the rest of its enqueued step is no-ops)
  // where the gradient goes (read here, not in front of the gather that needs it: a scalar load from device memory)
  float* grad_base = a.grad;
  if (a.ctl_cur != nullptr && (*a.ctl_cur ^ 1) != 0) grad_base = a.grad_alt;
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  v2f* tr = reinterpret_cast<v2f*>(uni) + wib * G::TR;
  const LaneConst<R> k = lane_consts<R>();
  const int lane = k.lane;

  const int b = blockIdx.x / a.nchunks, c = blockIdx.x - b * a.nchunks;
  const int t0 = hop_chunk_begin(c, a.T, a.nchunks), t1 = hop_chunk_begin(c + 1, a.T, a.nchunks);
  const int nfr = t1 - t0;                                  // <= 16
  const float* xrow = a.x + (long long)b * a.len;
  // the samples of this wave's two frames are requested first: they fly while the tables are built
  v2f zin[2][R];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int n = 2 * wib + i;
    if (n < nfr) {
      load_frame_raw<R>(xrow, a.len, (long long)(t0 + n) * a.hop - a.pad, lane, a.pad_mode, zin[i]);
    } else {
#pragma unroll
      for (int u = 0; u < R; ++u) zin[i][u] = v2f{0.0f, 0.0f};   // a frame beyond the tile: zeros all the way through
    }
  }
  // ... and so are the target values this lane will compare against in step 3 (output (m, n) of the forward contraction)
  constexpr int NQ = (4 * MT + kObjWaves - 1) / kObjWaves;
  float tgt[NQ];
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const int q = wib + kObjWaves * i, m = 16 * (q >> 2) + 4 * (lane >> 4) + (q & 3), n = lane & 15;
    tgt[i] = (!MAG && !SP && q < 4 * MT && m < a.n_mels && n < nfr) ? a.target[((long long)b * a.n_mels + m) * a.T + t0 + n] : 0.0f;
  }
  // SP: this wave's list of row quads (lane i: entry i), and the tile's targets [row][frame] - they wait in the dM tile, where
  // step 3 replaces each by its dM
  int sp_list = 0, sp_count = 0;
  constexpr int NT = (16 * MT * 16 + 64 * kObjWaves - 1) / (64 * kObjWaves);
  float sp_tgt[NT];
  if constexpr (SP) {
    sp_count = a.tab[ObjSp::COUNT + wib];
    sp_list = a.tab[ObjSp::LIST + ObjSp::MAXQ * wib + (lane & (ObjSp::MAXQ - 1))];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int e = threadIdx.x + 64 * kObjWaves * i, m = e >> 4, n = e & 15;
      sp_tgt[i] = (m < a.n_mels && n < nfr) ? a.target[((long long)b * a.n_mels + m) * a.T + t0 + n] : 0.0f;
    }
  }
  for (int i = threadIdx.x; i < M; i += blockDim.x) lds_win[i] = v2f{a.window[2 * i], a.window[2 * i + 1]};
  for (int i = threadIdx.x; i < (FP - F) * RS; i += blockDim.x) tile[obj_at(F + (i >> 4), i & 15)] = 0.0f;   // rows the zero-padded filterbank meets
  OBJ_STAMP(0);
  {
    v2f* twt = reinterpret_cast<v2f*>(uni);                   // W_M^(l*k1), (R-1) x 64 entries: built once, kept in registers
    for (int i = threadIdx.x; i < (R - 1) * 64; i += blockDim.x) {
      const int k1 = i / 64 + 1, l = i & 63;
      twt[i] = unit(2.0f * (float)((l * k1) % M) / (float)M);
    }
  }

  const float hs = 0.5f * a.fwd_scale;
  __syncthreads();
  TwRegs<R> twr;
#pragma unroll
  for (int k1 = 1; k1 < R; ++k1) twr.w[k1 - 1] = reinterpret_cast<const v2f*>(uni)[(k1 - 1) * 64 + lane];
  __syncthreads();                                            // (the staging area is the FFT scratch from here on)
  OBJ_STAMP(1);

  // ---- 1. analysis of this wave's two frames ---------------------------------------------------------------------------
  v2f un[2][H], um[2][H], umid[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int n = 2 * wib + i;
    // rows lane + 64 j and M - lane - 64 j of the quad-major tile: lane-constant bases, 1024 floats per j (objective_args.h)
    const int ak0 = obj_at(lane, n), am0 = obj_at(M - lane, n);
    v2f z[R];
#pragma unroll
    for (int u = 0; u < R; ++u) z[u] = zin[i][u] * lds_win[64 * u + lane];
    fft_forward_t<R>(z, k, twr, tr);
    v2f rc[H];
#pragma unroll
    for (int m = H; m < R; ++m) {
      const v2f got = shfl2(z[m], k.partner);
      const v2f own = z[(m + 1) % R];
      rc[m - H] = v2f{lane == 0 ? own.x : got.x, lane == 0 ? own.y : got.y};
    }
#pragma unroll
    for (int j = 0; j < H; ++j) {
      const v2f wk = pair_twiddle<R>(k.wn, j);
      const v2f zk = z[j], zm = rc[R - 1 - j - H];
      const v2f e2 = add_conj(zk, zm);
      const v2f tw = cmul_mi(wk, sub_conj(zk, zm));
      const v2f xk = (e2 + tw) * hs;
      const v2f xm = (e2 - tw) * v2f{hs, -hs};
      const float ak = fast_abs(xk), am = fast_abs(xm);
      tile[ak0 + 1024 * j] = ak;
      tile[am0 - 1024 * j] = am;
      const float ik = ak > 0.0f ? fast_rcp(ak) : 0.0f, im = am > 0.0f ? fast_rcp(am) : 0.0f;
      un[i][j] = xk * ik;                                                      // G = dA * S/|S|, 0 where |S| = 0
      um[i][j] = xm * im;
    }
    const v2f xmid = z[H] * v2f{a.fwd_scale, -a.fwd_scale};                  // bin M/2 (lane 0)
    const float amid = fast_abs(xmid);
    if (lane == 0) tile[obj_at(M / 2, n)] = amid;
    umid[i] = xmid * (amid > 0.0f ? fast_rcp(amid) : 0.0f);
  }
  OBJ_STAMP(2);
  constexpr int kRing = 8;
  int n_blk = 1, fe0 = 0, fe1 = 0;
  const int* blk_mg = nullptr;
  const int* blk_fg = nullptr;
  f32x4 av[kRing];
  int dsc_fg = 0, dsc_mg = 0;
  f32x4 stg[ObjSp::STAGE];
  if constexpr (SP) {
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int e = threadIdx.x + 64 * kObjWaves * i;
      if (e < 16 * MT * 16) dmt[e] = sp_tgt[i];
    }
    // the band blob is requested now and goes to LDS after the barrier
#pragma unroll
    for (int i = 0; i < ObjSp::STAGE; ++i) stg[i] = a.melA[min((int)threadIdx.x + 64 * kObjWaves * i, a.sp_total - 1)];
  }
  if constexpr (!MAG && !SP) {
    // operand blocks of the forward contraction: a ring of kRing blocks in flight per wave (a mel filterbank leaves a wave
    // fewer blocks than that: all of them are requested here, before the barrier, and land while the other waves finish)
    n_blk = a.tab[ObjTab::BEGIN + KQ];
    blk_mg = a.tab + ObjTab::mel_group(KQ);
    blk_fg = a.tab + ObjTab::bin_group(KQ, n_blk);
    fe0 = a.tab[ObjTab::FWD + wib];
    fe1 = a.tab[ObjTab::FWD + wib + 1];
#pragma unroll
    for (int i = 0; i < kRing; ++i) av[i] = a.melA[(long long)min(fe0 + i, n_blk - 1) * 64 + lane];
    // ... and so do the blocks' descriptors (bin group, mel group): lane i of a descriptor register holds block e + i of the
    // ring pass, read back with v_readlane - a scalar load from the table per block would sit, unprefetched, in front of every
    // block's LDS reads (~300 cycles each, nine blocks per wave)
    dsc_fg = blk_fg[min(fe0 + (lane & (kRing - 1)), n_blk - 1)];
    dsc_mg = blk_mg[min(fe0 + (lane & (kRing - 1)), n_blk - 1)];
  }
  __syncthreads();
  OBJ_STAMP(3);

  if constexpr (MAG) {
    // ---- 2'-4'. dA = 2/numel (|S| - T) on the tile, squared error; the target is read in the caller's (B, F, T) layout,
    // 16 consecutive frames (64 bytes) per bin
    double s2 = 0.0;
    for (int e = threadIdx.x; e < F * kObjTile; e += blockDim.x) {
      const int f = e >> 4, n = e & 15;
      if (n < nfr) {
        const float d = tile[obj_at(f, n)] - a.target[((long long)b * F + f) * a.T + t0 + n];
        s2 += (double)d * (double)d;
        tile[obj_at(f, n)] = a.dscale * d;
      }
    }
    s2 = wave_sum(s2);
    if (lane == 0) lsum[wib] = s2;
    __syncthreads();
    if (threadIdx.x == 0) {
      double tot = 0.0;
      for (int w = 0; w < kObjWaves; ++w) tot += lsum[w];
      a.partials[blockIdx.x] = tot;
    }
  } else if constexpr (SP) {
    // the bands go to LDS (the FFT scratch idles between the transforms): read straight from global memory - 23 KB of L1 / L2
    // hits on a path of their own - the two loops below wait longer for their operands than the LDS port costs them (measured:
    // forward 4.5 k -> 5.2 k cycles, backward 3.8 k -> 6.5 k)
    f32x4* u4 = reinterpret_cast<f32x4*>(uni);
#pragma unroll
    for (int i = 0; i < ObjSp::STAGE; ++i) {
      const int u = threadIdx.x + 64 * kObjWaves * i;
      if (u < a.sp_total) u4[u] = stg[i];
    }
    __syncthreads();
    OBJ_STAMP(4);
    // ---- 2 + 3 (bands). lane (r, n): mm = row 4 g + r of the filterbank . |S|[:, n] over the row's band, then log1p, squared
    // error and dM in place; dM tile plain [row][frame].  Four bin quads per pass (a band is a multiple of four long); the next
    // pass's operands - the NEXT row quad's first pass after the last one - are requested before this pass's products, so a wave
    // waits for its LDS reads once per list, not once per row quad.  (Bands starting on multiples of four quads would save nine
    // address instructions per pass - the swizzle then moves two bits of n by a lane constant - and cost 34 % more passes.)
    {
      const int r = lane >> 4, n = lane & 15;
      const int* rrec = reinterpret_cast<const int*>(u4 + a.sp_rm);
      const f32x4* t4s = reinterpret_cast<const f32x4*>(tile);
      double s2 = 0.0;
      int e_cur = __builtin_amdgcn_readlane(sp_list, 0);
      int m_cur = 4 * (e_cur & 255) + r;
      int ptr_cur = 0, q0_cur = 0;
      float tv_cur = 0.0f;
      f32x4 w[4], sv[4];
      if (sp_count > 0) {
        ptr_cur = rrec[2 * m_cur];
        q0_cur = rrec[2 * m_cur + 1];
        tv_cur = dmt[m_cur * 16 + n];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          w[c] = u4[ptr_cur + c];
          sv[c] = t4s[obj_quad(q0_cur + c, n)];               // (q0 + len <= FP / 4: obj_build_sparse)
        }
      }
      for (int i = 0; i < sp_count; ++i) {
        const int len = e_cur >> 8;
        const int e_nx = __builtin_amdgcn_readlane(sp_list, i + 1 < sp_count ? i + 1 : i);
        const int m_nx = 4 * (e_nx & 255) + r;
        const int ptr_nx = rrec[2 * m_nx], q0_nx = rrec[2 * m_nx + 1];
        const float tv_nx = dmt[m_nx * 16 + n];              // (this lane's own element: nobody else writes it)
        float v = 0.0f;
        for (int t = 0; t < len; t += 4) {
          const bool lastp = t + 4 >= len;
          const int pn = lastp ? ptr_nx : ptr_cur + t + 4, qn = lastp ? q0_nx : q0_cur + t + 4;
          f32x4 wn[4], sn[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            wn[c] = u4[pn + c];
            sn[c] = t4s[obj_quad(qn + c, n)];
          }
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            v = fmaf(w[c][0], sv[c][0], v);
            v = fmaf(w[c][1], sv[c][1], v);
            v = fmaf(w[c][2], sv[c][2], v);
            v = fmaf(w[c][3], sv[c][3], v);
          }
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            w[c] = wn[c];
            sv[c] = sn[c];
          }
        }
        float dm = 0.0f;
        if (m_cur < a.n_mels && n < nfr) dm = obj_point(v, tv_cur, a.dscale, s2);
        dmt[m_cur * 16 + n] = dm;
        e_cur = e_nx;
        m_cur = m_nx;
        ptr_cur = ptr_nx;
        q0_cur = q0_nx;
        tv_cur = tv_nx;
      }
      OBJ_STAMP(5);
      s2 = wave_sum(s2);
      if (lane == 0) lsum[wib] = s2;
    }
    __syncthreads();
    OBJ_STAMP(6);
    if (threadIdx.x == 0) {
      double tot = 0.0;
      for (int w = 0; w < kObjWaves; ++w) tot += lsum[w];
      a.partials[blockIdx.x] = tot;
    }
    // ---- 4 (bands). thread (bin quad q, frame n): dA[4 q + c, n] = sum_j W[m0 + j, 4 q + c] dM[m0 + j, n]; dA overwrites |S|.
    // A zero weight may meet one of the three rows past the filterbank's last quad: they hold the zeros of the staged targets
    // (obj_build_sparse keeps them inside the tile)
    {
      const int2* crec = reinterpret_cast<const int2*>(u4 + a.sp_cm);
      const int n = threadIdx.x & 15;
      const char* dmn = reinterpret_cast<const char*>(dmt + n);
      constexpr int QS = (64 * kObjWaves) >> 4, NI = (FP / 4 + QS - 1) / QS;     // bin quads per pass of the workgroup, passes
      int2 cr[NI];
#pragma unroll
      for (int i = 0; i < NI; ++i) cr[i] = crec[min((int)(threadIdx.x >> 4) + QS * i, FP / 4 - 1)];   // all records first: the dM reads hang on them
      // three bin quads per trip, their weights and dM values all requested before the first product (CM = rows per bin, padded to
      // 2 or 4 with zero weights by obj_build_sparse)
      auto backward = [&](auto cm_c) {
        constexpr int CM = decltype(cm_c)::value, BATCH = CM == 2 ? 3 : 1;
#pragma unroll
        for (int ib = 0; ib < NI; ib += BATCH) {
          f32x4 w[BATCH][CM];
          float d[BATCH][CM][4];
#pragma unroll
          for (int k = 0; k < BATCH; ++k)
            if (ib + k < NI) {
              const int q = min((int)(threadIdx.x >> 4) + QS * (ib + k), FP / 4 - 1);      // (past the end: the last quad again, not stored)
              const int2 c2 = cr[ib + k];
              const int off[4] = {c2.x & 0xffff, (int)((unsigned)c2.x >> 16), c2.y & 0xffff, (int)((unsigned)c2.y >> 16)};
#pragma unroll
              for (int j = 0; j < CM; ++j) {
                w[k][j] = u4[a.sp_cw + q * CM + j];
#pragma unroll
                for (int c = 0; c < 4; ++c) d[k][j][c] = *reinterpret_cast<const float*>(dmn + off[c] + 64 * j);
              }
            }
#pragma unroll
          for (int k = 0; k < BATCH; ++k)
            if (ib + k < NI) {
              const int q = (threadIdx.x >> 4) + QS * (ib + k);
              f32x4 out = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
              for (int j = 0; j < CM; ++j)
#pragma unroll
                for (int c = 0; c < 4; ++c) out[c] = fmaf(w[k][j][c], d[k][j][c], out[c]);
              if (q < FP / 4) reinterpret_cast<f32x4*>(tile)[obj_quad(q, n)] = out;
            }
        }
      };
      if (a.sp_cmax == 2) backward(std::integral_constant<int, 2>{});
      else backward(std::integral_constant<int, 4>{});
    }
  } else {
  // ---- 2. forward contraction mm[m, n] = sum_f Mel[m, f] |S|[f, n]: the waves split the list of non-zero blocks -------
  {
    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    for (int e = fe0; e < fe1; e += kRing) {
      const int cfg = dsc_fg, cmg = dsc_mg;
      if (e + kRing < fe1) {                        // the next pass's descriptors, a pass ahead
        dsc_fg = blk_fg[min(e + kRing + (lane & (kRing - 1)), n_blk - 1)];
        dsc_mg = blk_mg[min(e + kRing + (lane & (kRing - 1)), n_blk - 1)];
      }
#pragma unroll
      for (int i = 0; i < kRing; ++i) {
        if (e + i < fe1) {
          const int fg = __builtin_amdgcn_readlane(cfg, i), mg = __builtin_amdgcn_readlane(cmg, i);
          const f32x4 bv = reinterpret_cast<const f32x4*>(tile)[obj_quad(4 * fg + (lane >> 4), lane & 15)];
          const f32x4 cur = av[i];
          if (e + i + kRing < fe1) av[i] = a.melA[(long long)(e + i + kRing) * 64 + lane];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            if (mg == mt) {
#pragma unroll
              for (int j = 0; j < 4; ++j) acc[mt] = mfma_16x16x4(cur[j], bv[j], acc[mt]);
            }
        }
      }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) uni[((wib * MT + mt) * 4 + r) * 64 + lane] = acc[mt][r];
  }
  // the backward contraction's blocks are requested now: they land during the reduction
  const int bg0 = a.tab[ObjTab::BWD + wib], bg1 = a.tab[ObjTab::BWD + wib + 1];
  const int be0 = a.tab[ObjTab::BEGIN + bg0], be1 = a.tab[ObjTab::BEGIN + bg1];
#pragma unroll
  for (int i = 0; i < kRing; ++i) av[i] = a.melB[(long long)min(be0 + i, n_blk - 1) * 64 + lane];
  dsc_fg = blk_fg[min(be0 + (lane & (kRing - 1)), n_blk - 1)];
  dsc_mg = blk_mg[min(be0 + (lane & (kRing - 1)), n_blk - 1)];
  OBJ_STAMP(4);
  __syncthreads();
  OBJ_STAMP(5);

  // ---- 3. V = log1p(mm), squared error, dM = 2/numel (V - T) / (1 + mm) ------------------------------------------------
  {
    double s2 = 0.0;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int q = wib + kObjWaves * i;
      if (q < 4 * MT) {
        const int mt = q >> 2, r = q & 3;
        float v = 0.0f;
#pragma unroll
        for (int w = 0; w < kObjWaves; ++w) v += uni[((w * MT + mt) * 4 + r) * 64 + lane];   // fixed order
        const int m = 16 * mt + 4 * (lane >> 4) + r, n = lane & 15;   // D[i = m][j = n]: col = lane & 15, row = 4 (lane >> 4) + r
        float dm = 0.0f;
        if (m < a.n_mels && n < nfr) dm = obj_point(v, tgt[i], a.dscale, s2);
        dmt[obj_at(m, n)] = dm;
      }
    }
    s2 = wave_sum(s2);
    if (lane == 0) lsum[wib] = s2;
  }
  __syncthreads();
  OBJ_STAMP(6);
  if (threadIdx.x == 0) {
    double tot = 0.0;
    for (int w = 0; w < kObjWaves; ++w) tot += lsum[w];
    a.partials[blockIdx.x] = tot;
  }

  // ---- 4. backward contraction dA[f, n] = sum_m Mel[m, f] dM[m, n]: the waves split the bin groups ----------------------
  {
    f32x4 c = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    int g = bg0;
    auto flush = [&](int upto) {                            // bin groups [g, upto) are complete (groups without a block: zeros)
      for (; g < upto; ++g) {
        reinterpret_cast<f32x4*>(tile)[obj_quad(4 * g + (lane >> 4), lane & 15)] = c;     // D rows 4 i .. 4 i + 3 of column c: one unit
        c = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      }
    };
    for (int e = be0; e < be1; e += kRing) {
      const int cfg = dsc_fg, cmg = dsc_mg;
      if (e + kRing < be1) {
        dsc_fg = blk_fg[min(e + kRing + (lane & (kRing - 1)), n_blk - 1)];
        dsc_mg = blk_mg[min(e + kRing + (lane & (kRing - 1)), n_blk - 1)];
      }
#pragma unroll
      for (int i = 0; i < kRing; ++i) {
        if (e + i < be1) {
          const int fg = __builtin_amdgcn_readlane(cfg, i), mg = __builtin_amdgcn_readlane(cmg, i);
          flush(fg);
          const f32x4 bv = reinterpret_cast<const f32x4*>(dmt)[obj_quad(4 * mg + (lane >> 4), lane & 15)];
          const f32x4 cur = av[i];
          if (e + i + kRing < be1) av[i] = a.melB[(long long)(e + i + kRing) * 64 + lane];
#pragma unroll
          for (int j = 0; j < 4; ++j) c = mfma_16x16x4(cur[j], bv[j], c);
        }
      }
    }
    flush(bg1);
  }
  }
  OBJ_STAMP(7);
  __syncthreads();
  OBJ_STAMP(8);

  // ---- 5. gradient frames: G = dA S/|S| (Hermitian weights), inverse FFT, window --------------------------------------------
  v2f fr[2][R];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int n = 2 * wib + i;
    const int ak0 = obj_at(lane, n), am0 = obj_at(M - lane, n);
    v2f z[R], back[H];
#pragma unroll
    for (int j = 0; j < H; ++j) {
      const v2f wk = pair_twiddle<R>(k.wn, j);
      const int kk = lane + 64 * j;
      // interior bins of the one-sided spectrum count half (their mirror images carry the other half); bins 0 and M do not
      const float hw = (kk == 0 ? 1.0f : 0.5f) * a.fwd_scale;
      v2f ak = un[i][j] * (tile[ak0 + 1024 * j] * hw);
      v2f am = um[i][j] * (tile[am0 - 1024 * j] * hw);
      if (j == 0 && lane == 0) {
        ak.y = 0.0f;
        am.y = 0.0f;
      }
      const v2f e2i = add_conj(ak, am);
      const v2f o2i = cmulc(sub_conj(ak, am), wk);
      z[j] = add_i(e2i, o2i);
      back[j] = conj_sub_i(e2i, o2i);
    }
    const float dmid = tile[obj_at(M / 2, n)] * a.fwd_scale;
    const v2f zmid = umid[i] * v2f{dmid, -dmid};
#pragma unroll
    for (int m = H; m < R; ++m) {
      const v2f got = shfl2(back[R - 1 - m], k.partner);
      const v2f l0 = (m == H) ? zmid : back[(R - m) % H];
      z[m] = v2f{lane == 0 ? l0.x : got.x, lane == 0 ? l0.y : got.y};
    }
    fft_inverse_t<R>(z, k, twr, tr);
#pragma unroll
    for (int u = 0; u < R; ++u) fr[i][u] = z[u] * lds_win[64 * u + lane];
  }
  OBJ_STAMP(9);
  __syncthreads();                                            // every inverse transform is done: the scratch becomes the span
  OBJ_STAMP(10);

  // ---- 6. overlap-add: the windowed frames go to LDS (the |S| / dA tile, the dM tile and the scratch are one region of
  // >= 16 n_fft floats, all free by now), then every output sample gathers what covers it, in frame order.  With hop = n_fft/2,
  // /4 or /8 a wave's two frames are first added in registers (frame 2 w + 1 lies hop samples - a whole number of its lanes'
  // 128-sample rows - behind frame 2 w): eight spans of n_fft + hop samples instead of sixteen frames, 5/8 of the LDS writes and
  // of the gather's reads.  Any other hop: the sixteen frames, the order of k_ola's gather in the unfused path. ------------------
  float* frames = tile;
  static_assert((size_t)FP * RS + 16 * MT * RS + OG::UNI >= (size_t)kObjTile * N, "the frame buffers do not fit");
  const int hrows = a.hop >> 7;                               // the hop in rows of 128 samples (64 lanes x 2)
  const bool pairs = (a.hop & 127) == 0 && (2 * hrows == R || 4 * hrows == R || 8 * hrows == R);
  if (pairs) {
    v2f* p2 = reinterpret_cast<v2f*>(frames + wib * (N + a.hop));
    auto put = [&](auto hs_c) {
      constexpr int HS = decltype(hs_c)::value;
#pragma unroll
      for (int u = 0; u < R + HS; ++u) {
        v2f v = v2f{0.0f, 0.0f};
        if (u < R) v = fr[0][u < R ? u : 0];
        if (u >= HS) v = u < R ? v + fr[1][u - HS] : fr[1][u - HS];
        p2[64 * u + lane] = v;
      }
    };
    if (2 * hrows == R) put(std::integral_constant<int, R / 2>{});
    else if (4 * hrows == R) put(std::integral_constant<int, R / 4>{});
    else put(std::integral_constant<int, R / 8>{});
  } else {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      v2f* f2 = reinterpret_cast<v2f*>(frames + (2 * wib + i) * N);
#pragma unroll
      for (int u = 0; u < R; ++u) f2[64 * u + lane] = fr[i][u];
    }
  }
  __syncthreads();
  OBJ_STAMP(11);
  // units of the gather: frames (step hop, length n_fft) or pair sums (step 2 hop, length n_fft + hop)
  const int ulen = pairs ? N + a.hop : N, ustep = pairs ? 2 * a.hop : a.hop, ush = pairs ? 1 : 0;
  const int ulast = pairs ? (nfr - 1) >> 1 : nfr - 1;
  // the tile's own frames*hop samples are final up to the previous tile's tail; the rest of the span is this tile's tail
  const int span_len = (nfr - 1) * a.hop + N;
  const long long p0 = (long long)t0 * a.hop;
  const int owned = nfr * a.hop, keep = N - a.hop;
  const bool last = c == a.nchunks - 1;
  float* go = grad_base + (long long)b * a.len;
  float* mgn = a.margins + (long long)b * 2 * a.pad;
  float* tl = a.xtail + ((long long)b * a.nchunks + c) * keep;
  const long long n0 = p0 - a.pad;                          // signal index of span[0]
  const int lim = last ? span_len : owned;                  // span[0, lim) -> gradient / margins, span[lim, span_len) -> tail
  if (((a.hop | a.pad) & 3) == 0 && n0 >= 0 && n0 + lim <= a.len && (a.len & 3) == 0) {
    // the common case: 16-byte pieces, everything inside the signal
    v4f* g4 = reinterpret_cast<v4f*>(go + n0);
    v4f* t4 = reinterpret_cast<v4f*>(tl);
    const int dlt = ulen - ustep;                           // sample s sits at frames[n ulen + s - n ustep] = frames[s + n dlt] in unit n
    const int total = span_len / 4;
    // NU units at most cover a sample; five outputs per thread and trip, their reads all requested before the first sum: a trip
    // of one output at a time waits for its LDS reads three times over (5.2 k cycles of a tile; this form: see DESIGN 3.7)
    auto gather = [&](auto nu_c) {
      constexpr int NU = decltype(nu_c)::value, UNR = 5, NTH = 64 * kObjWaves;
      for (int base = threadIdx.x; base < total; base += UNR * NTH) {
        v4f t[UNR][NU];
        bool h[UNR][NU];
#pragma unroll
        for (int k = 0; k < UNR; ++k) {
          const int s4 = min(base + k * NTH, total - 1), s = 4 * s4;      // (past the end: the last output again, not stored)
          // units n_lo .. n_hi cover sample s (division by the hop: multiplication by ceil(2^32 / hop), exact below 2^16)
          const int n_lo = s < ulen ? 0 : (int)(__umulhi((unsigned)(s - ulen), a.hop_magic) >> ush) + 1;
          int n_hi = (int)(__umulhi((unsigned)s, a.hop_magic) >> ush);
          if (n_hi > ulast) n_hi = ulast;
          const float* q = frames + s + n_lo * dlt;
          t[k][0] = *reinterpret_cast<const v4f*>(q);
#pragma unroll
          for (int j = 1; j < NU; ++j) {
            h[k][j] = n_lo + j <= n_hi;
            q = h[k][j] ? q + dlt : q;                      // (a unit past the last one: its neighbour's address again, not added)
            t[k][j] = *reinterpret_cast<const v4f*>(q);
          }
        }
#pragma unroll
        for (int k = 0; k < UNR; ++k) {
          const int s4 = base + k * NTH, s = 4 * s4;
          v4f acc = t[k][0];
#pragma unroll
          for (int j = 1; j < NU; ++j)
            if (h[k][j]) acc = acc + t[k][j];               // unit order
          if (s4 < total) {
            if (s < lim) g4[s4] = acc;
            else t4[s4 - lim / 4] = acc;
          }
        }
      }
    };
    if (ulen <= 3 * ustep) gather(std::integral_constant<int, 3>{});
    else if (ulen <= 5 * ustep) gather(std::integral_constant<int, 5>{});
    else
    for (int s4 = threadIdx.x; s4 < total; s4 += blockDim.x) {
      const int s = 4 * s4;
      const int n_lo = s < ulen ? 0 : (int)(__umulhi((unsigned)(s - ulen), a.hop_magic) >> ush) + 1;
      int n_hi = (int)(__umulhi((unsigned)s, a.hop_magic) >> ush);
      if (n_hi > ulast) n_hi = ulast;
      v4f acc = v4f{0.0f, 0.0f, 0.0f, 0.0f};
      for (int nb = n_lo; nb <= n_hi; nb += 4) {            // four reads in flight, added in unit order
        const float* p0q = frames + s + nb * dlt;
        const bool h1 = nb + 1 <= n_hi, h2 = nb + 2 <= n_hi, h3 = nb + 3 <= n_hi;
        const float* p1q = h1 ? p0q + dlt : p0q;
        const float* p2q = h2 ? p1q + dlt : p1q;
        const float* p3q = h3 ? p2q + dlt : p2q;
        const v4f t0 = *reinterpret_cast<const v4f*>(p0q), t1 = *reinterpret_cast<const v4f*>(p1q);
        const v4f t2 = *reinterpret_cast<const v4f*>(p2q), t3 = *reinterpret_cast<const v4f*>(p3q);
        acc = acc + t0;
        if (h1) acc = acc + t1;
        if (h2) acc = acc + t2;
        if (h3) acc = acc + t3;
      }
      if (s < lim) g4[s4] = acc;
      else t4[s4 - lim / 4] = acc;
    }
  } else {
    for (int s = threadIdx.x; s < span_len; s += blockDim.x) {
      int n_lo = s < ulen ? 0 : (s - ulen) / ustep + 1, n_hi = s / ustep;
      if (n_hi > ulast) n_hi = ulast;
      float v = 0.0f;
      for (int n = n_lo; n <= n_hi; ++n) v += frames[n * ulen + (s - n * ustep)];
      if (s < lim) {
        const long long nn = n0 + s;
        if (nn >= 0 && nn < a.len) go[nn] = v;
        else if (nn < 0) mgn[p0 + s] = v;
        else if (nn - a.len < a.pad) mgn[a.pad + (nn - a.len)] = v;
      } else {
        tl[s - owned] = v;
      }
    }
  }
  if (last) {
    // samples beyond the last frame (a signal a little longer than (T - 1) hop + n_fft - 2 pad: torch.stft drops the
    // remainder) take no part in the objective: zero gradient
    for (long long nn = n0 + span_len + threadIdx.x; nn < a.len; nn += blockDim.x) go[nn] = 0.0f;
  }
  OBJ_STAMP(12);
}


}  // namespace fast
}  // namespace specinv
