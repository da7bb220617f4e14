// Fused gfx950 fast path: one Griffin-Lim / ADMM iteration = ONE kernel launch (the wave-level FFT primitives, the fused
// kernels and the host-side state; the any-hop frame kernels built on the same primitives are in kernels_frame.h).
//
// Mapping (n_fft = N = 128*R with R in {4, 8, 16, 32}, hop = N/2, N/4 or N/8, float32, onesided, centred, any pad mode;
// described for hop = N/4):
//   * one 64-lane wave owns one frame at a time and walks a chunk of consecutive frames of one
//     batch item; nothing is shared between waves except read-only tables, so the frame loop
//     has no workgroup barrier.
//   * the N real samples are packed as M = N/2 = 64*R complex points, R per lane
//     (lane l, register u  <->  z[64u + l] = x[128u + 2l] + i x[128u + 2l + 1], a coalesced
//     512-byte row per load instruction).  The M-point FFT runs as
//         in-register radix-R  ->  cross-lane radix-(64/R) butterflies on the gfx950 lane-swap
//         instructions (v_permlane32_swap / v_permlane16_swap)  ->  RxR transpose through
//         wave-private LDS  ->  in-register radix-R,
//     which leaves bin k in lane k mod 64, register k div 64.
//   * real-FFT split: lane r trades its upper R/2 registers with lane 64-r (ds_bpermute), after
//     which every lane holds R/2 conjugate pairs (k, M-k).  The spectral state (pre_spec / X, U /
//     target magnitude) lives in HBM in exactly that pair order, so each lane reads and writes
//     16-byte pieces of contiguous 1-KiB rows.  Bin M/2 is the one odd bin; lane 0 carries it.
//   * the momentum / ADMM update and the magnitude projection are applied to the pairs in
//     registers, the pairs are folded back, and the mirrored inverse FFT returns the frame in
//     the input register layout.
//   * overlap-add is done in registers: with hop = N/4 a lane's R registers split into 4
//     quarters that land on 4 consecutive hop-blocks; three quarter-accumulators are carried
//     from frame to frame and one finished hop-block is normalised by the envelope and stored
//     per frame.  No halo: the first three hop-blocks of a chunk are stored as two partial sums
//     (own frames in x, the previous chunk's last three frames in xtail) that the next
//     iteration's loader adds.
//   Algorithmic HBM traffic per frame-iteration: 8*hop + 20*F bytes (GLA), 8*hop + 36*F (ADMM).
#pragma once
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "common.h"
#include "kernels_generic.h"

// build-time tunables of the fused kernel (tools/sweep_variants.py builds variants, tools/run_variants.sh times them)
#ifndef SPECINV_XPREF      // 0: load the frame when it starts; 1: carry the samples in registers and prefetch one
#define SPECINV_XPREF 1    //    hop-block ahead (measured best: 0.336 vs 0.352 ms on C2); 2: fetch the whole next
#endif                     //    frame after the spectral update so that it flies during the inverse FFT
#ifndef SPECINV_PLATE      // 0: issue the state loads at the start of the frame; 1: after the forward FFT;
#define SPECINV_PLATE 0    // 2: one frame ahead, right after the previous frame's spectral update freed the registers
#endif
#ifndef SPECINV_TW_REGS    // 1: keep the pass-1 twiddles of the FFT in registers instead of re-reading the LDS table
#define SPECINV_TW_REGS 1   // measured on C2: 0.313 vs 0.318 ms
#endif
#ifndef SPECINV_ABLATE     // timing experiments (WRONG RESULTS): 1 no state stores, 2 no state loads, 4 no FFTs
#define SPECINV_ABLATE 0
#endif
#ifndef SPECINV_MINWAVES   // __launch_bounds__ waves per SIMD (caps the register allocation)
#define SPECINV_MINWAVES 2
#endif
#ifndef SPECINV_PRIO        // k_fused4: wave priority (bits 0-1) while a frame's state loads and the sample prefetch are being
#define SPECINV_PRIO 1      // issued, so that they are not queued behind the other wave's FFT; +4: also around the output
#endif                      // store.  Measured on two boxes (C2, ms per launch): 0 0.3023 / 0.3093, 1 0.2997, 3 0.3011 / 0.3042
#ifndef SPECINV_WGW         // most waves per workgroup of k_fused4 (they share the window / twiddle tables in LDS): 8-wave
#define SPECINV_WGW 8       // workgroups (one per CU) measured 2-3 % faster than 4-wave ones once every wave slot is filled
#endif
#ifndef SPECINV_NT
#define SPECINV_NT 1         // nontemporal state streams (keeps the re-used samples in L2)
#endif
#ifndef SPECINV_IEEE        // 1: correctly rounded sqrt / division in the projection and a true division by the envelope
#define SPECINV_IEEE 0      //    (the reference's operations, methods.py:132,246-247) instead of v_sqrt_f32 / v_rcp_f32 and
#endif                      //    a multiplication by 1/envelope: the accuracy study of tools/dbg_acc.py, profiles/r02_ieee_study.txt

namespace specinv {
namespace fast {

using v2f = float __attribute__((ext_vector_type(2)));
using v4f = float __attribute__((ext_vector_type(4)));

enum { MODE_GLA = 0, MODE_ADMM = 1 };

// complex products with a literal constant operand: four scalar operations with inline literals
__device__ __forceinline__ v2f cmul_k(v2f a, v2f b) { return v2f{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ v2f cmulc_k(v2f a, v2f b) { return v2f{a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y}; }
// ... with both operands in registers: TWO packed operations - the VOP3P source modifiers broadcast a.x / a.y over both
// halves, pick b's halves crosswise and negate one product (the compiler emits 2 v_mul + 2 v_fma for the scalar form; the
// wave-level kernels are bound by the instructions a wave can issue, a packed one counts once)
#ifndef SPECINV_ASM_CMUL
#define SPECINV_ASM_CMUL 1
#endif
__device__ __forceinline__ v2f cmul(v2f a, v2f b) {
#if SPECINV_ASM_CMUL
  v2f t, d;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b));                 // (a.x b.x, a.x b.y)
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(d) : "v"(a), "v"(b), "v"(t));
  return d;                                                                                               // (- a.y b.y, + a.y b.x)
#else
  return cmul_k(a, b);
#endif
}
// a * conj(b)
__device__ __forceinline__ v2f cmulc(v2f a, v2f b) {
#if SPECINV_ASM_CMUL
  v2f t, d;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1] neg_hi:[1,0]" : "=v"(t) : "v"(a), "v"(b));    // (a.x b.x, - a.x b.y)
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(t)); // (+ a.y b.y, + a.y b.x)
  return d;
#else
  return cmulc_k(a, b);
#endif
}
__device__ __forceinline__ v2f cconj(v2f a) { return v2f{a.x, -a.y}; }
__device__ __forceinline__ v2f mul_i(v2f a) { return v2f{-a.y, a.x}; }    // a * (+i)
__device__ __forceinline__ v2f mul_mi(v2f a) { return v2f{a.y, -a.x}; }   // a * (-i)
// four products a_i <- a_i * b_i (CONJ: a_i * conj(b_i)) in one block: the four multiplies first, then the four dependent
// multiply-adds, so that no instruction waits for the one issued just before it (the compiler has no latency model for inline
// assembly and would leave each pair back to back)
template <bool CONJ>
__device__ __forceinline__ void cmul_x4(v2f& a0, v2f& a1, v2f& a2, v2f& a3, v2f b0, v2f b1, v2f b2, v2f b3) {
#if SPECINV_ASM_CMUL
  v2f t0, t1, t2, t3;
  if (!CONJ) {
    asm("v_pk_mul_f32 %4, %0, %8 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %5, %1, %9 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %6, %2, %10 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %7, %3, %11 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %0, %8, %4 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"
        "v_pk_fma_f32 %1, %1, %9, %5 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"
        "v_pk_fma_f32 %2, %2, %10, %6 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"
        "v_pk_fma_f32 %3, %3, %11, %7 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "v"(b0), "v"(b1), "v"(b2), "v"(b3));
  } else {
    asm("v_pk_mul_f32 %4, %0, %8 op_sel:[0,0] op_sel_hi:[0,1] neg_hi:[1,0]\n\t"
        "v_pk_mul_f32 %5, %1, %9 op_sel:[0,0] op_sel_hi:[0,1] neg_hi:[1,0]\n\t"
        "v_pk_mul_f32 %6, %2, %10 op_sel:[0,0] op_sel_hi:[0,1] neg_hi:[1,0]\n\t"
        "v_pk_mul_f32 %7, %3, %11 op_sel:[0,0] op_sel_hi:[0,1] neg_hi:[1,0]\n\t"
        "v_pk_fma_f32 %0, %0, %8, %4 op_sel:[1,1,0] op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %1, %1, %9, %5 op_sel:[1,1,0] op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %2, %2, %10, %6 op_sel:[1,1,0] op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %3, %3, %11, %7 op_sel:[1,1,0] op_sel_hi:[1,0,1]"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "v"(b0), "v"(b1), "v"(b2), "v"(b3));
  }
#else
  a0 = CONJ ? cmulc_k(a0, b0) : cmul_k(a0, b0);
  a1 = CONJ ? cmulc_k(a1, b1) : cmul_k(a1, b1);
  a2 = CONJ ? cmulc_k(a2, b2) : cmul_k(a2, b2);
  a3 = CONJ ? cmulc_k(a3, b3) : cmul_k(a3, b3);
#endif
}
// z[i] <- z[i] * w(i) for i = FIRST .. R-1 (R a multiple of 4, FIRST 0 or 1)
template <int R, bool PK, bool CONJ, int FIRST, typename W>
__device__ __forceinline__ void cmul_all(v2f (&z)[R], const W& w) {
  if (PK) {
#pragma unroll
    for (int i = FIRST; i < 4; ++i) z[i] = CONJ ? cmulc(z[i], w(i)) : cmul(z[i], w(i));
#pragma unroll
    for (int g = 4; g < R; g += 4) cmul_x4<CONJ>(z[g], z[g + 1], z[g + 2], z[g + 3], w(g), w(g + 1), w(g + 2), w(g + 3));
  } else {
#pragma unroll
    for (int i = FIRST; i < R; ++i) z[i] = CONJ ? cmulc_k(z[i], w(i)) : cmul_k(z[i], w(i));
  }
}
struct SameW {
  v2f w;
  __device__ __forceinline__ v2f operator()(int) const { return w; }
};
template <bool PK>
__device__ __forceinline__ v2f cmul_p(v2f a, v2f b) { return PK ? cmul(a, b) : cmul_k(a, b); }
template <bool PK>
__device__ __forceinline__ v2f cmulc_p(v2f a, v2f b) { return PK ? cmulc(a, b) : cmulc_k(a, b); }
// (-i w) * d = (w.y d.x + w.x d.y, w.y d.y - w.x d.x) without forming -i w
__device__ __forceinline__ v2f cmul_mi(v2f w, v2f d) {
#if SPECINV_ASM_CMUL
  v2f t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(t) : "v"(w), "v"(d));                              // (w.y d.x, w.y d.y)
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,0,1] neg_hi:[1,0,0]" : "=v"(r) : "v"(w), "v"(d), "v"(t));
  return r;                                                                                                              // (+ w.x d.y, - w.x d.x)
#else
  return cmul_k(v2f{w.y, -w.x}, d);
#endif
}
template <bool INV>
__device__ __forceinline__ v2f rot(v2f a) { return INV ? mul_i(a) : mul_mi(a); }
// exp(-+ i*theta) from (cos, sin): forward uses (c, -s), inverse (c, +s)
template <bool INV>
__device__ __forceinline__ v2f twc(float c, float s) { return v2f{c, INV ? s : -s}; }
template <bool INV>
__device__ __forceinline__ v2f dirmul(v2f a, v2f w) { return INV ? cmulc_k(a, w) : cmul_k(a, w); }   // (literal twiddles)

__device__ __forceinline__ v2f shfl_xor2(v2f a, int mask) {
  return v2f{__shfl_xor(a.x, mask, 64), __shfl_xor(a.y, mask, 64)};
}
__device__ __forceinline__ v2f shfl2(v2f a, int src) { return v2f{__shfl(a.x, src, 64), __shfl(a.y, src, 64)}; }

// a + i*b and a - i*b as ONE packed add: VOP3P source modifiers pick b's halves crosswise (op_sel) and negate
// one of them, so the multiplication by +-i costs nothing (the compiler otherwise emits v_xor + v_mov for it).
#ifndef SPECINV_ASM_ROT
#define SPECINV_ASM_ROT 1
#endif
__device__ __forceinline__ v2f add_i(v2f a, v2f b) {   // (a.x - b.y, a.y + b.x)
#if SPECINV_ASM_ROT
  v2f d;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(d) : "v"(a), "v"(b));
  return d;
#else
  return v2f{a.x - b.y, a.y + b.x};
#endif
}
__device__ __forceinline__ v2f sub_i(v2f a, v2f b) {   // (a.x + b.y, a.y - b.x)
#if SPECINV_ASM_ROT
  v2f d;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
  return d;
#else
  return v2f{a.x + b.y, a.y - b.x};
#endif
}

__device__ __forceinline__ v2f add_conj(v2f a, v2f b) {   // a + conj(b)
#if SPECINV_ASM_ROT
  v2f d;
  asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
  return d;
#else
  return v2f{a.x + b.x, a.y - b.y};
#endif
}
__device__ __forceinline__ v2f sub_conj(v2f a, v2f b) {   // a - conj(b)
#if SPECINV_ASM_ROT
  v2f d;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(d) : "v"(a), "v"(b));
  return d;
#else
  return v2f{a.x - b.x, a.y + b.y};
#endif
}
__device__ __forceinline__ v2f conj_sub_i(v2f a, v2f b) {   // conj(a - i*b) = (a.x + b.y, -a.y + b.x)
#if SPECINV_ASM_ROT
  v2f d;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b));
  return d;
#else
  return v2f{a.x + b.y, b.x - a.y};
#endif
}

// ---- small in-register DFTs (natural order in, natural order out) ------------------------------
template <bool INV>
__device__ __forceinline__ void dft4(v2f& a0, v2f& a1, v2f& a2, v2f& a3) {
  const v2f t0 = a0 + a2, t1 = a0 - a2, t2 = a1 + a3, d = a1 - a3;
  a0 = t0 + t2;
  a2 = t0 - t2;
  // forward: X1 = t1 - i d, X3 = t1 + i d ; inverse: signs swapped
  a1 = INV ? add_i(t1, d) : sub_i(t1, d);
  a3 = INV ? sub_i(t1, d) : add_i(t1, d);
}

template <int R, bool INV>
struct Dft;

template <bool INV>
struct Dft<4, INV> {
  static __device__ __forceinline__ void run(v2f (&a)[4]) { dft4<INV>(a[0], a[1], a[2], a[3]); }
};

template <bool INV>
struct Dft<8, INV> {
  static __device__ __forceinline__ void run(v2f (&a)[8]) {
    constexpr float h = 0.70710678118654752440f;
    // n = 4*n1 + n0: radix-2 over n1, twiddle W8^(n0*k1), radix-4 over n0
#pragma unroll
    for (int n0 = 0; n0 < 4; ++n0) {
      const v2f s = a[n0] + a[n0 + 4], d = a[n0] - a[n0 + 4];
      a[n0] = s;
      a[n0 + 4] = d;
    }
    a[5] = dirmul<INV>(a[5], v2f{h, -h});
    a[6] = rot<INV>(a[6]);
    a[7] = dirmul<INV>(a[7], v2f{-h, -h});
    dft4<INV>(a[0], a[1], a[2], a[3]);
    dft4<INV>(a[4], a[5], a[6], a[7]);
    // a[k0 + 4*k1] = X[k1 + 2*k0]
    v2f o[8];
#pragma unroll
    for (int k0 = 0; k0 < 4; ++k0) {
      o[2 * k0] = a[k0];
      o[2 * k0 + 1] = a[k0 + 4];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = o[i];
  }
};

template <bool INV>
struct Dft<16, INV> {
  static __device__ __forceinline__ void run(v2f (&a)[16]) {
    constexpr float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, h = 0.70710678118654752440f;
    // n = 4*n1 + n0: radix-4 over n1 (in place -> slot n0 + 4*k1), twiddle W16^(n0*k1), radix-4 over n0
#pragma unroll
    for (int n0 = 0; n0 < 4; ++n0) dft4<INV>(a[n0], a[n0 + 4], a[n0 + 8], a[n0 + 12]);
    a[5] = dirmul<INV>(a[5], v2f{c1, -s1});    // W16^1
    a[6] = dirmul<INV>(a[6], v2f{h, -h});      // W16^2
    a[7] = dirmul<INV>(a[7], v2f{s1, -c1});    // W16^3
    a[9] = dirmul<INV>(a[9], v2f{h, -h});      // W16^2
    a[10] = rot<INV>(a[10]);                   // W16^4
    a[11] = dirmul<INV>(a[11], v2f{-h, -h});   // W16^6
    a[13] = dirmul<INV>(a[13], v2f{s1, -c1});  // W16^3
    a[14] = dirmul<INV>(a[14], v2f{-h, -h});   // W16^6
    a[15] = dirmul<INV>(a[15], v2f{-c1, s1});  // W16^9
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) dft4<INV>(a[4 * k1], a[4 * k1 + 1], a[4 * k1 + 2], a[4 * k1 + 3]);
    // a[k0 + 4*k1] = X[k1 + 4*k0]
    v2f o[16];
#pragma unroll
    for (int k0 = 0; k0 < 4; ++k0)
#pragma unroll
      for (int k1 = 0; k1 < 4; ++k1) o[k1 + 4 * k0] = a[k0 + 4 * k1];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = o[i];
  }
};

// W_32^e = exp(-2 pi i e / 32)
__device__ __forceinline__ v2f w32(int e) {
  constexpr float tab[32][2] = {{1.000000000e+00f, -0.000000000e+00f}, {9.807852804e-01f, -1.950903220e-01f}, {9.238795325e-01f, -3.826834324e-01f}, {8.314696123e-01f, -5.555702330e-01f}, {7.071067812e-01f, -7.071067812e-01f}, {5.555702330e-01f, -8.314696123e-01f}, {3.826834324e-01f, -9.238795325e-01f}, {1.950903220e-01f, -9.807852804e-01f}, {6.123233996e-17f, -1.000000000e+00f}, {-1.950903220e-01f, -9.807852804e-01f}, {-3.826834324e-01f, -9.238795325e-01f}, {-5.555702330e-01f, -8.314696123e-01f}, {-7.071067812e-01f, -7.071067812e-01f}, {-8.314696123e-01f, -5.555702330e-01f}, {-9.238795325e-01f, -3.826834324e-01f}, {-9.807852804e-01f, -1.950903220e-01f}, {-1.000000000e+00f, -1.224646799e-16f}, {-9.807852804e-01f, 1.950903220e-01f}, {-9.238795325e-01f, 3.826834324e-01f}, {-8.314696123e-01f, 5.555702330e-01f}, {-7.071067812e-01f, 7.071067812e-01f}, {-5.555702330e-01f, 8.314696123e-01f}, {-3.826834324e-01f, 9.238795325e-01f}, {-1.950903220e-01f, 9.807852804e-01f}, {-1.836970199e-16f, 1.000000000e+00f}, {1.950903220e-01f, 9.807852804e-01f}, {3.826834324e-01f, 9.238795325e-01f}, {5.555702330e-01f, 8.314696123e-01f}, {7.071067812e-01f, 7.071067812e-01f}, {8.314696123e-01f, 5.555702330e-01f}, {9.238795325e-01f, 3.826834324e-01f}, {9.807852804e-01f, 1.950903220e-01f}};
  return v2f{tab[e][0], tab[e][1]};
}

template <bool INV>
struct Dft<32, INV> {
  static __device__ __forceinline__ void run(v2f (&a)[32]) {
    // n = 8*n1 + n0: radix-4 over n1 (in place -> slot n0 + 8*k1), twiddle W32^(n0*k1), radix-8 over n0
#pragma unroll
    for (int n0 = 0; n0 < 8; ++n0) dft4<INV>(a[n0], a[n0 + 8], a[n0 + 16], a[n0 + 24]);
#pragma unroll
    for (int k1 = 1; k1 < 4; ++k1)
#pragma unroll
      for (int n0 = 1; n0 < 8; ++n0) {
        const int e = n0 * k1;
        if (e == 8) a[n0 + 8 * k1] = rot<INV>(a[n0 + 8 * k1]);
        else a[n0 + 8 * k1] = dirmul<INV>(a[n0 + 8 * k1], w32(e));
      }
    v2f o[32];
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) {
      v2f t[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) t[i] = a[8 * k1 + i];
      Dft<8, INV>::run(t);
      // t[k0] = X[k1 + 4*k0]
#pragma unroll
      for (int k0 = 0; k0 < 8; ++k0) o[k1 + 4 * k0] = t[k0];
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) a[i] = o[i];
  }
};

template <int R>
struct Geo {
  static constexpr int C = 64 / R;       // cross-lane radix
  static constexpr int LOGC = C == 2 ? 1 : C == 4 ? 2 : C == 8 ? 3 : 4;
  static_assert(R == 4 || R == 8 || R == 16 || R == 32, "n_fft must be 512, 1024, 2048 or 4096");
  static constexpr int M = 64 * R;       // complex points per frame
  static constexpr int N = 2 * M;        // n_fft
  static constexpr int H = R / 2;        // conjugate pairs per lane
  static constexpr int QU = R / 4;       // registers per hop-block quarter
  static constexpr int HOP = N / 4;
  static constexpr int TR = 64 * (R + 1);  // transpose scratch per wave (complex elements)
  static constexpr size_t lds_bytes(int waves) { return sizeof(v2f) * (size_t)(M + (R - 1) * 64 + waves * TR); }
};

// overlap geometry of the fused kernels: hop = N / OV, OV in {2, 4, 8}
template <int R, int OV>
struct Ovl {
  static_assert(OV == 2 || OV == 4 || OV == 8, "hop must be n_fft / 2, / 4 or / 8");
  static_assert(R % OV == 0, "a hop-block must be whole registers");
  static constexpr int HOP = Geo<R>::N / OV;
  static constexpr int QU = R / OV;      // registers per hop-block
  static constexpr int NB = OV - 1;      // hop-blocks carried from frame to frame (accumulators, sample window, tails)
  static constexpr int PB = OV / 2;      // hop-blocks of centre padding on either side
};

// per-lane constants
template <int R>
struct LaneConst {
  int lane, n2, v, kv, partner;
  v2f post;                // W_64^(n2*kv)
  v2f stage[2];            // twiddles between the cross-lane radix-4 and radix-2 steps (C == 8: one, C == 16: two)
  v2f wn;                  // W_N^lane
  int tr_a;                // transpose address for layout A: (kv*R + reg)*(R+1) + n2  -> base + reg*(R+1)
  int tr_b;                // layout B: lane*(R+1) + reg
};

__device__ __forceinline__ v2f unit(float turns_times_2) {  // exp(-i*pi*x)
  float s, c;
  sincospif(turns_times_2, &s, &c);
  return v2f{c, -s};
}

template <int R>
__device__ __forceinline__ LaneConst<R> lane_consts() {
  using G = Geo<R>;
  LaneConst<R> k;
  k.lane = threadIdx.x & 63;
  k.n2 = k.lane % R;
  k.v = k.lane / R;
  // frequency digit held at lane position v after the cross-lane transform, and the one twiddle of the
  // radix-4 x radix-2 split (C == 8 only): W8^(vl * kh) with v = 2*vh + vl, kv = kh + 4*kl, kh = vh, kl = vl
  k.stage[1] = v2f{1.0f, 0.0f};
  if (G::C == 4 || G::C == 2) {
    k.kv = k.v;
    k.stage[0] = v2f{1.0f, 0.0f};
  } else if (G::C == 8) {
    const int vh = k.v >> 1, vl = k.v & 1;
    k.kv = vh + 4 * vl;
    k.stage[0] = unit(2.0f * (float)(vl * vh) / 8.0f);
  } else {
    // C == 16, v = 4*vh + vl, vl = 2*b3 + b2 (lane bits 3, 2): radix-4 over vh, twiddle W16^(vl*kh), then the radix-4
    // over vl as two radix-2 steps (bit 3 first, twiddle W4^(b2*k3), then bit 2); digits kh = vh, kl = b3 + 2*b2
    const int vh = k.v >> 2, vl = k.v & 3, b3 = vl >> 1, b2 = vl & 1;
    k.kv = vh + 4 * (b3 + 2 * b2);
    k.stage[0] = unit(2.0f * (float)(vl * vh) / 16.0f);
    k.stage[1] = (b3 && b2) ? v2f{0.0f, -1.0f} : v2f{1.0f, 0.0f};
  }
  k.partner = (64 - k.lane) & 63;
  k.post = unit(2.0f * (float)(k.n2 * k.kv) / 64.0f);
  k.wn = unit(2.0f * (float)k.lane / (float)G::N);
  k.tr_a = (k.kv * R) * (R + 1) + k.n2;
  k.tr_b = k.lane * (R + 1);
  return k;
}

// ---- cross-lane butterflies with the gfx950 lane-swap instructions ---------------------------------
// v_permlane32_swap a, b : swaps a[32..63] with b[0..31];  v_permlane16_swap a, b : swaps the odd 16-lane
// rows of a with the even rows of b.  Two levels of swaps bring the four values that sit in lanes
// l, l+16, l+32, l+48 of ONE register into four registers of one lane (each 16-lane row ends up
// owning one of the four registers), a plain in-register radix-4 runs at full lane efficiency, and the
// same swaps in reverse order put result kv back into lane row kv.  No LDS round trip, no twiddles.
__device__ __forceinline__ void swap32(v2f& a, v2f& b) {
  const auto rx = __builtin_amdgcn_permlane32_swap(__float_as_uint(a.x), __float_as_uint(b.x), false, false);
  const auto ry = __builtin_amdgcn_permlane32_swap(__float_as_uint(a.y), __float_as_uint(b.y), false, false);
  a = v2f{__uint_as_float(rx[0]), __uint_as_float(ry[0])};
  b = v2f{__uint_as_float(rx[1]), __uint_as_float(ry[1])};
}
__device__ __forceinline__ void swap16(v2f& a, v2f& b) {
  const auto rx = __builtin_amdgcn_permlane16_swap(__float_as_uint(a.x), __float_as_uint(b.x), false, false);
  const auto ry = __builtin_amdgcn_permlane16_swap(__float_as_uint(a.y), __float_as_uint(b.y), false, false);
  a = v2f{__uint_as_float(rx[0]), __uint_as_float(ry[0])};
  b = v2f{__uint_as_float(rx[1]), __uint_as_float(ry[1])};
}
// DFT over lane bits 5,4 (lane = 16*q + r, q = 0..3) of four registers at once
template <bool INV>
__device__ __forceinline__ void xlane_dft4(v2f& A, v2f& B, v2f& C, v2f& D) {
  swap32(A, B);
  swap32(C, D);
  swap16(A, C);
  swap16(B, D);
  dft4<INV>(A, C, B, D);   // this lane's four values, q = 0..3, sit in (A, C, B, D)
  swap16(A, C);
  swap16(B, D);
  swap32(A, B);
  swap32(C, D);
}

// DFT over lane bit 5 of two registers at once: after the first swap the low half of the wave holds both halves
// of A (in A, B) and the high half both halves of B
__device__ __forceinline__ void xlane_dft2(v2f& A, v2f& B) {
  swap32(A, B);
  const v2f sum = A + B, dif = A - B;
  A = sum;
  B = dif;
  swap32(A, B);
}
// radix-2 over one lane bit (lower lane: a + partner, upper lane: partner - a)
__device__ __forceinline__ v2f xlane_bf2(v2f z, int mask, float sg) {
  const v2f p = shfl_xor2(z, mask);
  return v2f{fmaf(z.x, sg, p.x), fmaf(z.y, sg, p.y)};
}

// ---- M-point FFT across the wave ------------------------------------------------------------------
// forward: in z[u] = time sample 64u + lane; out z[j] = bin lane + 64j
// pass-1 twiddles W_M^(lane*k1): read from the LDS table (default) or from a per-lane register copy
struct TwLds {
  const v2f* t;
  int lane;
  __device__ __forceinline__ v2f operator()(int k1) const { return t[(k1 - 1) * 64 + lane]; }
};
template <int R>
struct TwRegs {
  v2f w[R - 1];
  __device__ __forceinline__ v2f operator()(int k1) const { return w[k1 - 1]; }
};

// PK: complex products as two packed operations (cmul) or four scalar ones (cmul_k).  Packed wins wherever two waves share a
// SIMD (C2 -6 %, C4 -5 %); a lone wave per SIMD (k_rtisi_fast) has nobody to cover the packed pair's dependent latency and
// measured 12 % slower with it, so that kernel asks for the scalar form.
template <int R, bool PK = true, typename TW>
__device__ __forceinline__ void fft_forward_t(v2f (&z)[R], const LaneConst<R>& k, const TW& tw, v2f* __restrict__ tr);
template <int R, bool PK = true, typename TW>
__device__ __forceinline__ void fft_inverse_t(v2f (&z)[R], const LaneConst<R>& k, const TW& tw, v2f* __restrict__ tr);

template <int R>
__device__ __forceinline__ void fft_forward(v2f (&z)[R], const LaneConst<R>& k, const v2f* __restrict__ tw1,
                                            v2f* __restrict__ tr) {
  fft_forward_t<R>(z, k, TwLds{tw1, k.lane}, tr);
}
template <int R>
__device__ __forceinline__ void fft_inverse(v2f (&z)[R], const LaneConst<R>& k, const v2f* __restrict__ tw1,
                                            v2f* __restrict__ tr) {
  fft_inverse_t<R>(z, k, TwLds{tw1, k.lane}, tr);
}

template <int R, bool PK, typename TW>
__device__ __forceinline__ void fft_forward_t(v2f (&z)[R], const LaneConst<R>& k, const TW& tw, v2f* __restrict__ tr) {
  using G = Geo<R>;
  Dft<R, false>::run(z);
  cmul_all<R, PK, false, 1>(z, tw);
  // cross-lane radix-C over v = lane / R; afterwards lane position v holds frequency digit k.kv
  if (G::C == 2) {
#pragma unroll
    for (int g = 0; g < R; g += 2) xlane_dft2(z[g], z[g + 1]);
  } else if (G::C == 4) {
#pragma unroll
    for (int g = 0; g < R; g += 4) xlane_dft4<false>(z[g], z[g + 1], z[g + 2], z[g + 3]);
  } else if (G::C == 16) {
#pragma unroll
    for (int g = 0; g < R; g += 4) xlane_dft4<false>(z[g], z[g + 1], z[g + 2], z[g + 3]);
    const float s3 = (k.lane & 8) ? -1.0f : 1.0f, s2 = (k.lane & 4) ? -1.0f : 1.0f;
#pragma unroll
    for (int i = 0; i < R; ++i) {
      z[i] = cmul_p<PK>(z[i], k.stage[0]);
      z[i] = cmul_p<PK>(xlane_bf2(z[i], 8, s3), k.stage[1]);
      z[i] = xlane_bf2(z[i], 4, s2);
    }
  } else {
    // C == 8, v = 2*vh + vl: radix-4 over vh (lane bits 5,4), twiddle W8^(vl*kh), radix-2 over vl (lane bit 3)
#pragma unroll
    for (int g = 0; g < R; g += 4) xlane_dft4<false>(z[g], z[g + 1], z[g + 2], z[g + 3]);
    const bool upper = k.v & 1;
    const float sg = upper ? -1.0f : 1.0f;
#pragma unroll
    for (int i = 0; i < R; ++i) {
      z[i] = cmul_p<PK>(z[i], k.stage[0]);
      const v2f p = shfl_xor2(z[i], R);
      z[i] = v2f{fmaf(z[i].x, sg, p.x), fmaf(z[i].y, sg, p.y)};   // lower: a + p ; upper: p - a
    }
  }
  cmul_all<R, PK, false, 0>(z, SameW{k.post});
  // transpose n2 <-> k1 inside each group of R lanes (wave-private LDS, no barrier)
#pragma unroll
  for (int i = 0; i < R; ++i) tr[k.tr_a + i * (R + 1)] = z[i];
#pragma unroll
  for (int i = 0; i < R; ++i) z[i] = tr[k.tr_b + i];
  Dft<R, false>::run(z);
}

// inverse (unnormalised): in z[j] = bin lane + 64j; out z[u] = time sample 64u + lane
template <int R, bool PK, typename TW>
__device__ __forceinline__ void fft_inverse_t(v2f (&z)[R], const LaneConst<R>& k, const TW& tw, v2f* __restrict__ tr) {
  using G = Geo<R>;
  Dft<R, true>::run(z);
#pragma unroll
  for (int i = 0; i < R; ++i) tr[k.tr_b + i] = z[i];
#pragma unroll
  for (int i = 0; i < R; ++i) z[i] = tr[k.tr_a + i * (R + 1)];
  cmul_all<R, PK, true, 0>(z, SameW{k.post});
  if (G::C == 2) {
#pragma unroll
    for (int g = 0; g < R; g += 2) xlane_dft2(z[g], z[g + 1]);
  } else if (G::C == 4) {
#pragma unroll
    for (int g = 0; g < R; g += 4) xlane_dft4<true>(z[g], z[g + 1], z[g + 2], z[g + 3]);
  } else if (G::C == 16) {
    const float s3 = (k.lane & 8) ? -1.0f : 1.0f, s2 = (k.lane & 4) ? -1.0f : 1.0f;
#pragma unroll
    for (int i = 0; i < R; ++i) {
      z[i] = cmulc_p<PK>(xlane_bf2(z[i], 4, s2), k.stage[1]);
      z[i] = cmulc_p<PK>(xlane_bf2(z[i], 8, s3), k.stage[0]);
    }
#pragma unroll
    for (int g = 0; g < R; g += 4) xlane_dft4<true>(z[g], z[g + 1], z[g + 2], z[g + 3]);
  } else {
    const bool upper = k.v & 1;
    const float sg = upper ? -1.0f : 1.0f;
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const v2f p = shfl_xor2(z[i], R);
      z[i] = v2f{fmaf(z[i].x, sg, p.x), fmaf(z[i].y, sg, p.y)};
      z[i] = cmulc_p<PK>(z[i], k.stage[0]);
    }
#pragma unroll
    for (int g = 0; g < R; g += 4) xlane_dft4<true>(z[g], z[g + 1], z[g + 2], z[g + 3]);
  }
  cmul_all<R, PK, true, 1>(z, tw);
  Dft<R, true>::run(z);
}

// W_64^j = exp(-2 pi i j / 64), j < 16: the per-pair step of the real-FFT twiddle W_N^(lane + 64 j)
__device__ __forceinline__ v2f w64(int j) {
  constexpr float tab[16][2] = {{1.000000000e+00f, -0.000000000e+00f}, {9.951847267e-01f, -9.801714033e-02f}, {9.807852804e-01f, -1.950903220e-01f}, {9.569403357e-01f, -2.902846773e-01f}, {9.238795325e-01f, -3.826834324e-01f}, {8.819212643e-01f, -4.713967368e-01f}, {8.314696123e-01f, -5.555702330e-01f}, {7.730104534e-01f, -6.343932842e-01f}, {7.071067812e-01f, -7.071067812e-01f}, {6.343932842e-01f, -7.730104534e-01f}, {5.555702330e-01f, -8.314696123e-01f}, {4.713967368e-01f, -8.819212643e-01f}, {3.826834324e-01f, -9.238795325e-01f}, {2.902846773e-01f, -9.569403357e-01f}, {1.950903220e-01f, -9.807852804e-01f}, {9.801714033e-02f, -9.951847267e-01f}};
  return v2f{tab[j][0], tab[j][1]};
}

struct FastArgs {
  const float* x_in;
  float* x_out;
  const float* xtail_in;   // [B][nchunks][3*HOP]: what chunk c's last three frames add to the first three
  float* xtail_out;        //                       hop-blocks of chunk c+1 (already times 1/envelope)
  const v4f* P_in;     // GLA: pre_spec pairs ; ADMM: Y = X + U    [B*T][H][64] x (Re k, Im k, Re M-k, Im M-k)
  v4f* P_out;
  const v2f* Pmid_in;  // bin M/2                                   [B*T]
  v2f* Pmid_out;
  v4f* X_out;          // ADMM, optional (nullptr: not wanted): X and U of this iteration, for specinv_get_state_spec.
  v4f* U_out;          // The recursion itself only needs their sum: methods.py:467-468 read X and U as U + X, which is
  v2f* Xmid_out;       // the Y = x_ + u that :475 has just rounded (float addition commutes), so carrying Y alone is
  v2f* Umid_out;       // bit-identical and halves the state traffic (16 F instead of 32 F bytes per frame and iteration)
  const v4f* m_pairs;  // target magnitude                          [B*T][H/2][64] x (k_2c, M-k_2c, k_2c+1, M-k_2c+1)
  const float* m_mid;  //                                           [B*T]
  const float* window;   // N
  const float* inv_env;  // L, 1 / envelope
  double* partials;      // [n_waves][2]
  int T, nchunks, n_waves, pad_mode;
  long long L;
  const float* x2_in;   // k_fused4_td: x_t (x_in / x_out carry z there)
  float* x2_out;
  float tds;            //              (-lr)^t
  unsigned long long* stamps;   // SPECINV_TD_STAMPS builds only: [n_waves][8]
  float coef;       // lr (GLA) or rho (ADMM)
  float inv1p;      // 1/(1+rho)
  float fwd_scale;  // 1 or N^-1/2
  float inv_scale;  // 1/N or N^-1/2
};

#if SPECINV_IEEE
__device__ __forceinline__ float fast_abs(v2f s) { return __fsqrt_rn(fmaf(s.x, s.x, s.y * s.y)); }
__device__ __forceinline__ float fast_rcp(float v) { return __fdiv_rn(1.0f, v); }
// the envelope table holds the envelope itself
__device__ __forceinline__ v2f env_apply(v2f v, v2f e) { return v2f{__fdiv_rn(v.x, e.x), __fdiv_rn(v.y, e.y)}; }
__device__ __forceinline__ float env_apply(float v, float e) { return __fdiv_rn(v, e); }
#else
__device__ __forceinline__ float fast_abs(v2f s) { return __builtin_amdgcn_sqrtf(fmaf(s.x, s.x, s.y * s.y)); }
__device__ __forceinline__ float fast_rcp(float v) { return __builtin_amdgcn_rcpf(v); }
// the envelope table holds 1 / envelope
__device__ __forceinline__ v2f env_apply(v2f v, v2f e) { return v * e; }
__device__ __forceinline__ float env_apply(float v, float e) { return v * e; }
#endif

// Frequency-domain update of one bin.  `r` is the STFT bin, `p`/`u` the stored state, `m` the target.
// Returns the bin to synthesise from (already multiplied by isc); writes the new state.
template <int MODE, bool EVAL>
__device__ __forceinline__ v2f update_bin(v2f r, v2f& p, v2f& u, v2f& xs, float m, const FastArgs& a, bool live, double& sd,
                                          double& so) {
  if (EVAL) {
    const float o = fast_abs(r);
    if (live) {
      const double d = (double)o - (double)m;
      sd += d * d;
      so += (double)o * (double)o;
    }
  }
  if (MODE == MODE_GLA) {
    // methods.py:243-247: S = R - lr*P ; P <- S ; S * m / (|S| + 1e-16)
    const v2f s = v2f{fmaf(-a.coef, p.x, r.x), fmaf(-a.coef, p.y, r.y)};
    p = s;
#if SPECINV_IEEE
    const float den = fast_abs(s) + 1e-16f;
    return v2f{__fdiv_rn(s.x * m, den) * a.inv_scale, __fdiv_rn(s.y * m, den) * a.inv_scale};
#else
    const float inv = fast_rcp(fast_abs(s) + 1e-16f) * a.inv_scale;
    return v2f{(s.x * m) * inv, (s.y * m) * inv};
#endif
  } else {
    // methods.py:467-475 with p = Y of the previous iteration (= fl(X + U), the first operation of :468)
    const v2f y = p;
    const v2f z = v2f{fmaf(a.coef, y.x, r.x) * a.inv1p, fmaf(a.coef, y.y, r.y) * a.inv1p};
    const v2f un = y - z;
    v2f xn = z - un;
#if SPECINV_IEEE
    const float den = fast_abs(xn) + 1e-16f;
    xn = v2f{__fdiv_rn(xn.x * m, den), __fdiv_rn(xn.y * m, den)};
#else
    const float inv = fast_rcp(fast_abs(xn) + 1e-16f);
    xn = v2f{(xn.x * m) * inv, (xn.y * m) * inv};
#endif
    xs = xn;
    u = un;
    p = xn + un;
    return p * a.inv_scale;
  }
}

// signal index of padded position n (n < 0 or n >= L) for torch.stft's pad modes; -1 = zero (constant padding)
__device__ __forceinline__ long long pad_index(long long n, long long L, int pad_mode) {
  if (n >= 0 && n < L) return n;
  switch (pad_mode) {
    case SPECINV_PAD_REFLECT:
      return n < 0 ? -n : 2 * (L - 1) - n;
    case SPECINV_PAD_REPLICATE:
      return n < 0 ? 0 : L - 1;
    case SPECINV_PAD_CIRCULAR:
      return n < 0 ? n + L : n - L;
    default:
      return -1;
  }
}

// Frames [chunk_begin(c), chunk_begin(c+1)) belong to wave-chunk c; sizes differ by at most one frame.
__device__ __host__ __forceinline__ int chunk_begin(int c, int T, int nchunks) {
  return (int)(((unsigned)c * (unsigned)T) / (unsigned)nchunks);   // c * T < 2^32 for any plan that fits in memory
}

// One hop-block (N/4 samples, padded-signal block index j) of row `xrow` in the register layout
// (lane l, register i <-> samples 128 i + 2 l, +1).  Blocks 2..T lie inside the signal; the two
// blocks on either side are torch.stft's reflect padding.  The first three hop-blocks of every chunk but
// the first were stored as two partial sums (the chunk's own frames in x, the previous chunk's last three
// frames in `xtail`); they are added here.  Edge (reflected) blocks never touch such blocks because the
// first and the last chunk are at least 6 frames long.
template <int R, int OV>
__device__ __forceinline__ void load_block(const float* __restrict__ xrow, const float* __restrict__ tailrow,
                                           long long L, int T, int c, int t_begin, int t_end, int j, int lane,
                                           int pad_mode, v2f (&q)[R / OV]) {
  using O = Ovl<R, OV>;
  constexpr int HOP = O::HOP, QU = O::QU, NB = O::NB, PB = O::PB;
  const long long s0 = (long long)(j - PB) * HOP;
  if (j >= PB && j <= T + PB - 2) {
    const v2f* src = reinterpret_cast<const v2f*>(xrow + s0);   // uniform
#pragma unroll
    for (int i = 0; i < QU; ++i) q[i] = src[64u * i + (unsigned)lane];
    // a wave only ever reads blocks t_begin .. t_end + NB - 1 of its own chunk c: split blocks are the chunk's
    // own first NB (other half from chunk c-1) and the next chunk's first NB (other half: this chunk's)
    int tc = -1, off = 0;
    if (c >= 1 && j - t_begin < NB) {
      tc = c - 1;
      off = j - t_begin;
    } else if (j >= t_end && j < T) {
      tc = c;
      off = j - t_end;
    }
    if (tc >= 0) {
      const v2f* tl = reinterpret_cast<const v2f*>(tailrow + ((long long)tc * NB + off) * HOP);
#pragma unroll
      for (int i = 0; i < QU; ++i) q[i] = q[i] + tl[64u * i + (unsigned)lane];
    }
  } else {
#pragma unroll
    for (int i = 0; i < QU; ++i) {
      const long long n0 = pad_index(s0 + 128 * i + 2 * lane, L, pad_mode);
      const long long n1 = pad_index(s0 + 128 * i + 2 * lane + 1, L, pad_mode);
      q[i] = v2f{n0 < 0 ? 0.0f : xrow[(unsigned)n0], n1 < 0 ? 0.0f : xrow[(unsigned)n1]};
    }
  }
}

__device__ __forceinline__ v4f ld_stream(const v4f* p) {
#if SPECINV_ABLATE & 2
  return v4f{1.0f, 0.5f, 0.25f, 2.0f} * (float)(((unsigned long long)p >> 4) & 7);
#else
#if SPECINV_NT
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
#endif
}
__device__ __forceinline__ void st_stream(v4f* p, v4f v) {
#if SPECINV_ABLATE & 1
  if (v.x == 1.2345e30f) __builtin_nontemporal_store(v, p);   // keeps the value alive, (almost) never stores
#elif SPECINV_NT
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}

// ---- hop = n_fft/4: the headline shape keeps its own hand-tuned copy of the kernel --------------------------------
// (k_fused<R, 4, ...> below is the same algorithm; the compiler's register allocation of this text is the one that
// was tuned and measured - 6 % faster at C2 - so it stays as it is.)
// One hop-block (N/4 samples, padded-signal block index j) of row `xrow` in the register layout
// (lane l, register i <-> samples 128 i + 2 l, +1).  Blocks 2..T lie inside the signal; the two
// blocks on either side are torch.stft's reflect padding.  The first three hop-blocks of every chunk but
// the first were stored as two partial sums (the chunk's own frames in x, the previous chunk's last three
// frames in `xtail`); they are added here.  Edge (reflected) blocks never touch such blocks because the
// first and the last chunk are at least 6 frames long.
template <int R>
__device__ __forceinline__ void load_block4(const float* __restrict__ xrow, const float* __restrict__ tailrow,
                                           long long L, int T, int c, int t_begin, int t_end, int j, int lane,
                                           int pad_mode, v2f (&q)[R / 4]) {
  constexpr int HOP = Geo<R>::HOP;
  const long long s0 = (long long)(j - 2) * HOP;
  if (j >= 2 && j <= T) {
    const v2f* src = reinterpret_cast<const v2f*>(xrow + s0);   // uniform
#pragma unroll
    for (int i = 0; i < R / 4; ++i) q[i] = src[64u * i + (unsigned)lane];
    // a wave only ever reads blocks t_begin .. t_end + 2 of its own chunk c: split blocks are the chunk's
    // own first three (other half from chunk c-1) and the next chunk's first three (other half: this chunk's)
    int tc = -1, off = 0;
    if (c >= 1 && j - t_begin < 3) {
      tc = c - 1;
      off = j - t_begin;
    } else if (j >= t_end && j < T) {
      tc = c;
      off = j - t_end;
    }
    if (tc >= 0) {
      const v2f* tl = reinterpret_cast<const v2f*>(tailrow + ((long long)tc * 3 + off) * HOP);
#pragma unroll
      for (int i = 0; i < R / 4; ++i) q[i] = q[i] + tl[64u * i + (unsigned)lane];
    }
  } else {
#pragma unroll
    for (int i = 0; i < R / 4; ++i) {
      const long long n0 = pad_index(s0 + 128 * i + 2 * lane, L, pad_mode);
      const long long n1 = pad_index(s0 + 128 * i + 2 * lane + 1, L, pad_mode);
      q[i] = v2f{n0 < 0 ? 0.0f : xrow[(unsigned)n0], n1 < 0 ? 0.0f : xrow[(unsigned)n1]};
    }
  }
}

template <int R, int MODE, bool EVAL>
#ifndef SPECINV_K4_ENVREG
#define SPECINV_K4_ENVREG 1
#endif
#ifndef SPECINV_R8_W3        // n_fft 1024: three waves per SIMD (3072 wave slots; 12-wave workgroups at hop 256).  The plain launches fit
#define SPECINV_R8_W3 1      // 168 registers (ADMM 2 spilled, the evaluating variants 8-31); measured against two waves per SIMD:
#endif                       // C4 34.3 -> 32.3 ms per step, Griffin-Lim 1024 / 256 0.135 -> 0.127 ms per iteration
__global__ __launch_bounds__((SPECINV_R8_W3 && R == 8) ? 768 : 64 * SPECINV_WGW, (SPECINV_R8_W3 && R == 8) ? 3 : SPECINV_MINWAVES) void k_fused4(FastArgs a) {
  using G = Geo<R>;
  constexpr int H = G::H, QU = G::QU, M = G::M, HOP = G::HOP;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  v2f* lds_win = reinterpret_cast<v2f*>(smem);
  v2f* lds_tw1 = lds_win + M;
  // wave-uniform values are forced into SGPRs: every global address below is then
  // "scalar base + 32-bit lane offset" instead of one 64-bit VGPR pointer per access
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  v2f* tr = lds_tw1 + (R - 1) * 64 + wib * G::TR;

  for (int i = threadIdx.x; i < M; i += blockDim.x) lds_win[i] = v2f{a.window[2 * i], a.window[2 * i + 1]};
  for (int i = threadIdx.x; i < (R - 1) * 64; i += blockDim.x) {
    const int k1 = i / 64 + 1, l = i & 63;
    lds_tw1[i] = unit(2.0f * (float)((l * k1) % M) / (float)M);   // W_M^(l*k1)
  }
  __syncthreads();

  const int w = blockIdx.x * (blockDim.x >> 6) + wib;
  if (w >= a.n_waves) return;
  const LaneConst<R> k = lane_consts<R>();
  const int lane = k.lane;
  const unsigned ulane = (unsigned)lane;
  const int b = w / a.nchunks, c = w - b * a.nchunks;
  const int t_begin = chunk_begin(c, a.T, a.nchunks);
  const int t_end = chunk_begin(c + 1, a.T, a.nchunks);
  const int t_start = t_begin;   // no halo: the previous chunk's share of the first three hop-blocks comes via xtail
  const float* xrow = a.x_in + (long long)b * a.L;
  const float* tailrow = a.xtail_in + (long long)b * a.nchunks * 3 * HOP;
  float* orow = a.x_out + (long long)b * a.L;
  const float half_scale = 0.5f * a.fwd_scale;

  v2f acc[3 * QU];
#pragma unroll
  for (int i = 0; i < 3 * QU; ++i) acc[i] = v2f{0.0f, 0.0f};
  double sd = 0.0, so = 0.0;
  // one block of the envelope reciprocal, periodic in the hop from hop-block 3 on (kernels_fast_td.h), kept in registers
  v2f envc[SPECINV_K4_ENVREG ? QU : 1];
  if (SPECINV_K4_ENVREG) {
    const v2f* e0 = reinterpret_cast<const v2f*>(a.inv_env + (long long)HOP);
#pragma unroll
    for (int i = 0; i < QU; ++i) envc[i] = e0[64u * i + ulane];
  }
#if SPECINV_TW_REGS
  TwRegs<R> twr;
#pragma unroll
  for (int k1 = 1; k1 < R; ++k1) twr.w[k1 - 1] = lds_tw1[(k1 - 1) * 64 + lane];
#endif

  // raw samples of the current frame: three hop-blocks carried from frame to frame plus the
  // new one, which is fetched one frame ahead so that its latency hides behind a whole frame
#if SPECINV_XPREF == 2
  v2f znext[R];
#pragma unroll
  for (int qq = 0; qq < 4; ++qq) {
    v2f q[QU];
    load_block4<R>(xrow, tailrow, a.L, a.T, c, t_begin, t_end, t_start + qq, lane, a.pad_mode, q);
#pragma unroll
    for (int i = 0; i < QU; ++i) znext[qq * QU + i] = q[i];
  }
#elif SPECINV_XPREF == 1
  v2f xq[3][QU], xn[QU];
  load_block4<R>(xrow, tailrow, a.L, a.T, c, t_begin, t_end, t_start, lane, a.pad_mode, xq[0]);
  load_block4<R>(xrow, tailrow, a.L, a.T, c, t_begin, t_end, t_start + 1, lane, a.pad_mode, xq[1]);
  load_block4<R>(xrow, tailrow, a.L, a.T, c, t_begin, t_end, t_start + 2, lane, a.pad_mode, xq[2]);
  load_block4<R>(xrow, tailrow, a.L, a.T, c, t_begin, t_end, t_start + 3, lane, a.pad_mode, xn);
#endif

  // state of frame `FI`: uniform bases (SGPR) + unsigned 32-bit lane offsets -> "saddr + voffset" addressing
#define SPECINV_STATE_LOADS4(FI)                                                            \
  do {                                                                                     \
    const long long fl_ = (FI);                                                            \
    const v4f* pin_ = a.P_in + fl_ * (H * 64);                                             \
    const v4f* min_ = a.m_pairs + fl_ * (H / 2 * 64);                                      \
    _Pragma("unroll") for (int j = 0; j < H; ++j) pp[j] = ld_stream(&pin_[j * 64u + ulane]); \
    _Pragma("unroll") for (int j = 0; j < H / 2; ++j) mm[j] = ld_stream(&min_[j * 64u + ulane]); \
    if (lane == 0) {                                                                       \
      pmid = a.Pmid_in[fl_];                                                               \
      mmid = a.m_mid[fl_];                                                                 \
    }                                                                                      \
  } while (0)
#if SPECINV_PLATE == 2
  v4f pp[H], mm[H / 2];
  v2f pmid = v2f{0.0f, 0.0f}, umid = v2f{0.0f, 0.0f};
  float mmid = 0.0f;
  SPECINV_STATE_LOADS4((long long)b * a.T + t_start);
#endif

  for (int t = t_start; t < t_end; ++t) {
    // Keep the loop-invariant table reads (window, twiddles) and products inside the loop: hoisted out
    // of it they pin ~80 VGPRs for the whole kernel and cost a wave of occupancy.
    asm volatile("" ::: "memory");
    v2f wn = k.wn;
    asm volatile("" : "+v"(wn));
    constexpr bool live = true;
    const long long fi = (long long)b * a.T + t;
    v4f* pout = a.P_out + fi * (H * 64);
    const bool keep_xu = MODE == MODE_ADMM && a.U_out != nullptr;   // (uniform: a kernel argument)
#if SPECINV_PLATE != 2
    v4f pp[H], mm[H / 2];
    v2f pmid = v2f{0.0f, 0.0f}, umid = v2f{0.0f, 0.0f};
    float mmid = 0.0f;
#endif
#if SPECINV_PLATE == 0
#if SPECINV_PRIO
    __builtin_amdgcn_s_setprio(SPECINV_PRIO & 3);
#endif
    SPECINV_STATE_LOADS4(fi);   // early: the loads fly during the forward FFT
#endif

    // ---- analysis: windowed frame -> registers
    v2f z[R];
#if SPECINV_XPREF == 2
#pragma unroll
    for (int u = 0; u < R; ++u) z[u] = znext[u] * lds_win[64 * u + lane];
#elif SPECINV_XPREF == 1
    // slide the sample window and prefetch the next hop-block
#pragma unroll
    for (int i = 0; i < QU; ++i) {
      z[i] = xq[0][i] * lds_win[64 * i + lane];
      z[QU + i] = xq[1][i] * lds_win[64 * (QU + i) + lane];
      z[2 * QU + i] = xq[2][i] * lds_win[64 * (2 * QU + i) + lane];
      z[3 * QU + i] = xn[i] * lds_win[64 * (3 * QU + i) + lane];
      xq[0][i] = xq[1][i];
      xq[1][i] = xq[2][i];
      xq[2][i] = xn[i];
    }
    if (t + 1 < t_end) load_block4<R>(xrow, tailrow, a.L, a.T, c, t_begin, t_end, t + 4, lane, a.pad_mode, xn);
#if SPECINV_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
#else
    {
      v2f q[QU];
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        load_block4<R>(xrow, tailrow, a.L, a.T, c, t_begin, t_end, t + qq, lane, a.pad_mode, q);
#pragma unroll
        for (int i = 0; i < QU; ++i) z[qq * QU + i] = q[i];
      }
#pragma unroll
      for (int u = 0; u < R; ++u) z[u] = z[u] * lds_win[64 * u + lane];
    }
#endif

#if SPECINV_ABLATE & 4
#elif SPECINV_TW_REGS
    fft_forward_t<R>(z, k, twr, tr);
#else
    fft_forward<R>(z, k, lds_tw1, tr);
#endif
#if SPECINV_PLATE == 1
    SPECINV_STATE_LOADS4(fi);
#endif

    // ---- conjugate partners: upper half of lane (64 - r)
    v2f rc[H];   // rc[i] pairs with own register H-1-i ... see below: rc[m-H] = Z[M - (lane + 64*(R-1-m))]
    // (the lane-0 special case is patched AFTER the shuffle: selecting between two elements of
    // one register array before it makes the compiler index the array dynamically)
#pragma unroll
    for (int m = H; m < R; ++m) {
      const v2f got = shfl2(z[m], k.partner);
      const v2f own = z[(m + 1) % R];              // lane 0 is its own partner, shifted by one register
      rc[m - H] = v2f{lane == 0 ? own.x : got.x, lane == 0 ? own.y : got.y};
    }

    // ---- per pair: split -> update -> fold back
    v2f back[H];   // back[j] = Z''[M - k_j], to be returned to the partner lane
#pragma unroll
    for (int j = 0; j < H; ++j) {
      // W_N^(lane + 64 j) = W_N^lane * W_{2R}^j
      const v2f wk = j == 0 ? wn : cmul_k(wn, w64(j * (32 / R)));
      const v2f zk = z[j], zm = rc[R - 1 - j - H];
      const v2f e2 = add_conj(zk, zm);
      const v2f dd = sub_conj(zk, zm);
      const v2f tw = cmul_mi(wk, dd);                 // W * (-i (Zk - conj Zm))
      v2f xk = (e2 + tw) * half_scale;
      v2f xm = (e2 - tw) * v2f{half_scale, -half_scale};   // conj(...)
      v2f pk = v2f{pp[j].x, pp[j].y}, pm = v2f{pp[j].z, pp[j].w};
      v2f uk = v2f{0.0f, 0.0f}, um = v2f{0.0f, 0.0f}, sk = v2f{0.0f, 0.0f}, sm = v2f{0.0f, 0.0f};
      const float mk = (j & 1) ? mm[j / 2].z : mm[j / 2].x;
      const float mq = (j & 1) ? mm[j / 2].w : mm[j / 2].y;
      v2f ak = update_bin<MODE, EVAL>(xk, pk, uk, sk, mk, a, live, sd, so);
      v2f am = update_bin<MODE, EVAL>(xm, pm, um, sm, mq, a, live, sd, so);
      if (live) {
        st_stream(&pout[j * 64u + ulane], v4f{pk.x, pk.y, pm.x, pm.y});
        if (keep_xu) {
          st_stream(&a.X_out[fi * (H * 64) + j * 64u + ulane], v4f{sk.x, sk.y, sm.x, sm.y});
          st_stream(&a.U_out[fi * (H * 64) + j * 64u + ulane], v4f{uk.x, uk.y, um.x, um.y});
        }
      }
      if (j == 0 && lane == 0) {   // bins 0 and M: irfft ignores their imaginary parts
        ak.y = 0.0f;
        am.y = 0.0f;
      }
      const v2f e2i = add_conj(ak, am);
      const v2f o2i = cmulc(sub_conj(ak, am), wk);
      z[j] = add_i(e2i, o2i);
      back[j] = conj_sub_i(e2i, o2i);
    }
    // ---- bin M/2 (lane 0): X = conj(Z), Z'' = 2 conj(X')
    v2f zmid;
    {
      v2f xmid = z[H] * v2f{a.fwd_scale, -a.fwd_scale};
      const bool live0 = live && lane == 0;
      v2f smid = v2f{0.0f, 0.0f};
      const v2f am = update_bin<MODE, EVAL>(xmid, pmid, umid, smid, mmid, a, live0, sd, so);
      if (live0) {
        a.Pmid_out[fi] = pmid;
        if (keep_xu) {
          a.Xmid_out[fi] = smid;
          a.Umid_out[fi] = umid;
        }
      }
      zmid = am * v2f{2.0f, -2.0f};
    }
    // ---- return the mirrored halves
#pragma unroll
    for (int m = H; m < R; ++m) {
      const v2f got = shfl2(back[R - 1 - m], k.partner);
      const v2f l0 = (m == H) ? zmid : back[(R - m) % H];
      z[m] = v2f{lane == 0 ? l0.x : got.x, lane == 0 ? l0.y : got.y};
    }
#if SPECINV_PLATE == 2
    // the state registers are free again: fetch the next frame's state now, it flies through the inverse FFT,
    // the overlap-add and the next forward FFT
    if (t + 1 < t_end) SPECINV_STATE_LOADS4(fi + 1);
#endif

#if SPECINV_XPREF == 2
    if (t + 1 < t_end) {
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        v2f q[QU];
        load_block4<R>(xrow, tailrow, a.L, a.T, c, t_begin, t_end, t + 1 + qq, lane, a.pad_mode, q);
#pragma unroll
        for (int i = 0; i < QU; ++i) znext[qq * QU + i] = q[i];
      }
    }
#endif

#if SPECINV_ABLATE & 4
#elif SPECINV_TW_REGS
    fft_inverse_t<R>(z, k, twr, tr);
#else
    asm volatile("" ::: "memory");   // re-read the twiddles instead of keeping them live since the forward FFT
    fft_inverse<R>(z, k, lds_tw1, tr);
#endif

    // ---- synthesis window, register overlap-add, one finished hop-block out
#pragma unroll
    for (int u = 0; u < R; ++u) z[u] = z[u] * lds_win[64 * u + lane];
#if SPECINV_PRIO & 4
    __builtin_amdgcn_s_setprio(3);
#endif
    if (live && t >= 2) {
      const long long o0 = (long long)(t - 2) * HOP;
      const v2f* envp = reinterpret_cast<const v2f*>(a.inv_env + o0);   // uniform
      v2f* outp = reinterpret_cast<v2f*>(orow + o0);
#pragma unroll
      for (int i = 0; i < QU; ++i)
        outp[64u * i + ulane] = env_apply(acc[i] + z[i], (SPECINV_K4_ENVREG && t >= 3) ? envc[SPECINV_K4_ENVREG ? i : 0] : envp[64u * i + ulane]);
    }
#if SPECINV_PRIO & 4
    __builtin_amdgcn_s_setprio(0);
#endif
#pragma unroll
    for (int i = 0; i < QU; ++i) {
      acc[i] = acc[QU + i] + z[QU + i];
      acc[QU + i] = acc[2 * QU + i] + z[2 * QU + i];
      acc[2 * QU + i] = z[3 * QU + i];
    }
  }
  if (t_end == a.T) {
    // the chunk that holds the last frame also finishes hop-block T (frames T-3 .. T-1)
    const long long o0 = (long long)(a.T - 2) * HOP;
    const v2f* envp = reinterpret_cast<const v2f*>(a.inv_env + o0);
    v2f* outp = reinterpret_cast<v2f*>(orow + o0);
#pragma unroll
    for (int i = 0; i < QU; ++i) outp[64u * i + ulane] = env_apply(acc[i], envp[64u * i + ulane]);
  } else {
    // what this chunk's last three frames contribute to the next chunk's first three hop-blocks
    v2f* tl = reinterpret_cast<v2f*>(a.xtail_out + ((long long)b * a.nchunks + c) * 3 * HOP);
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const v2f* envp = reinterpret_cast<const v2f*>(a.inv_env + (long long)(t_end + q - 2) * HOP);
#pragma unroll
      for (int i = 0; i < QU; ++i) tl[(q * QU + i) * 64u + ulane] = env_apply(acc[q * QU + i], envp[64u * i + ulane]);
    }
  }
  if (EVAL) {
    const double d = wave_sum(sd), o = wave_sum(so);
    if (lane == 0) {
      a.partials[2 * (long long)w] = d;
      a.partials[2 * (long long)w + 1] = o;
    }
  }
}

#include "kernels_fast_td.h"   // k_fused4_td / k_fused_td: the same iteration with the momentum carried as a signal

template <int R, int OV, int MODE, bool EVAL>
__global__ __launch_bounds__(256, R >= 32 ? 1 : (SPECINV_R8_W3 && R == 8) ? 3 : SPECINV_MINWAVES) void k_fused(FastArgs a) {
  using G = Geo<R>;
  using O = Ovl<R, OV>;
  constexpr int H = G::H, M = G::M, QU = O::QU, HOP = O::HOP, NB = O::NB, PB = O::PB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  v2f* lds_win = reinterpret_cast<v2f*>(smem);
  v2f* lds_tw1 = lds_win + M;
  // wave-uniform values are forced into SGPRs: every global address below is then
  // "scalar base + 32-bit lane offset" instead of one 64-bit VGPR pointer per access
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  v2f* tr = lds_tw1 + (R - 1) * 64 + wib * G::TR;

  for (int i = threadIdx.x; i < M; i += blockDim.x) lds_win[i] = v2f{a.window[2 * i], a.window[2 * i + 1]};
  for (int i = threadIdx.x; i < (R - 1) * 64; i += blockDim.x) {
    const int k1 = i / 64 + 1, l = i & 63;
    lds_tw1[i] = unit(2.0f * (float)((l * k1) % M) / (float)M);   // W_M^(l*k1)
  }
  __syncthreads();

  const int w = blockIdx.x * (blockDim.x >> 6) + wib;
  if (w >= a.n_waves) return;
  const LaneConst<R> k = lane_consts<R>();
  const int lane = k.lane;
  const unsigned ulane = (unsigned)lane;
  const int b = w / a.nchunks, c = w - b * a.nchunks;
  const int t_begin = chunk_begin(c, a.T, a.nchunks);
  const int t_end = chunk_begin(c + 1, a.T, a.nchunks);
  const int t_start = t_begin;   // no halo: the previous chunk's share of the first NB hop-blocks comes via xtail
  const float* xrow = a.x_in + (long long)b * a.L;
  const float* tailrow = a.xtail_in + (long long)b * a.nchunks * NB * HOP;
  float* orow = a.x_out + (long long)b * a.L;
  const float half_scale = 0.5f * a.fwd_scale;

  v2f acc[NB * QU];
#pragma unroll
  for (int i = 0; i < NB * QU; ++i) acc[i] = v2f{0.0f, 0.0f};
  double sd = 0.0, so = 0.0;
#if SPECINV_TW_REGS
  TwRegs<R> twr;
#pragma unroll
  for (int k1 = 1; k1 < R; ++k1) twr.w[k1 - 1] = lds_tw1[(k1 - 1) * 64 + lane];
#endif

  // raw samples of the current frame: NB hop-blocks carried from frame to frame plus the
  // new one, which is fetched one frame ahead so that its latency hides behind a whole frame
  v2f xq[NB][QU], xn[QU];
#pragma unroll
  for (int q = 0; q < NB; ++q)
    load_block<R, OV>(xrow, tailrow, a.L, a.T, c, t_begin, t_end, t_start + q, lane, a.pad_mode, xq[q]);
  load_block<R, OV>(xrow, tailrow, a.L, a.T, c, t_begin, t_end, t_start + NB, lane, a.pad_mode, xn);

  // state of frame `FI`: uniform bases (SGPR) + unsigned 32-bit lane offsets -> "saddr + voffset" addressing
#define SPECINV_STATE_LOADS(FI)                                                            \
  do {                                                                                     \
    const long long fl_ = (FI);                                                            \
    const v4f* pin_ = a.P_in + fl_ * (H * 64);                                             \
    const v4f* min_ = a.m_pairs + fl_ * (H / 2 * 64);                                      \
    _Pragma("unroll") for (int j = 0; j < H; ++j) pp[j] = ld_stream(&pin_[j * 64u + ulane]); \
    _Pragma("unroll") for (int j = 0; j < H / 2; ++j) mm[j] = ld_stream(&min_[j * 64u + ulane]); \
    if (lane == 0) {                                                                       \
      pmid = a.Pmid_in[fl_];                                                               \
      mmid = a.m_mid[fl_];                                                                 \
    }                                                                                      \
  } while (0)

  for (int t = t_start; t < t_end; ++t) {
    // Keep the loop-invariant table reads (window, twiddles) and products inside the loop: hoisted out
    // of it they pin ~80 VGPRs for the whole kernel and cost a wave of occupancy.
    asm volatile("" ::: "memory");
    v2f wn = k.wn;
    asm volatile("" : "+v"(wn));
    constexpr bool live = true;
    const long long fi = (long long)b * a.T + t;
    v4f* pout = a.P_out + fi * (H * 64);
    const bool keep_xu = MODE == MODE_ADMM && a.U_out != nullptr;   // (uniform: a kernel argument)
    v4f pp[H], mm[H / 2];
    v2f pmid = v2f{0.0f, 0.0f}, umid = v2f{0.0f, 0.0f};
    float mmid = 0.0f;
#if SPECINV_PRIO
    if (R < 32) __builtin_amdgcn_s_setprio(SPECINV_PRIO & 3);   // (one wave per SIMD at R = 32: nothing to outrank)
#endif
    SPECINV_STATE_LOADS(fi);   // early: the loads fly during the forward FFT

    // ---- analysis: windowed frame -> registers; slide the sample window and prefetch the next hop-block
    v2f z[R];
#pragma unroll
    for (int i = 0; i < QU; ++i) {
#pragma unroll
      for (int q = 0; q < NB; ++q) z[q * QU + i] = xq[q][i] * lds_win[64 * (q * QU + i) + lane];
      z[NB * QU + i] = xn[i] * lds_win[64 * (NB * QU + i) + lane];
#pragma unroll
      for (int q = 0; q + 1 < NB; ++q) xq[q][i] = xq[q + 1][i];
      xq[NB - 1][i] = xn[i];
    }
    if (t + 1 < t_end) load_block<R, OV>(xrow, tailrow, a.L, a.T, c, t_begin, t_end, t + OV, lane, a.pad_mode, xn);
#if SPECINV_PRIO
    if (R < 32) __builtin_amdgcn_s_setprio(0);
#endif

#if SPECINV_ABLATE & 4
#elif SPECINV_TW_REGS
    fft_forward_t<R>(z, k, twr, tr);
#else
    fft_forward<R>(z, k, lds_tw1, tr);
#endif

    // ---- conjugate partners: upper half of lane (64 - r)
    v2f rc[H];   // rc[i] pairs with own register H-1-i ... see below: rc[m-H] = Z[M - (lane + 64*(R-1-m))]
    // (the lane-0 special case is patched AFTER the shuffle: selecting between two elements of
    // one register array before it makes the compiler index the array dynamically)
#pragma unroll
    for (int m = H; m < R; ++m) {
      const v2f got = shfl2(z[m], k.partner);
      const v2f own = z[(m + 1) % R];              // lane 0 is its own partner, shifted by one register
      rc[m - H] = v2f{lane == 0 ? own.x : got.x, lane == 0 ? own.y : got.y};
    }

    // ---- per pair: split -> update -> fold back
    v2f back[H];   // back[j] = Z''[M - k_j], to be returned to the partner lane
#pragma unroll
    for (int j = 0; j < H; ++j) {
      // W_N^(lane + 64 j) = W_N^lane * W_{2R}^j
      const v2f wk = j == 0 ? wn : cmul_k(wn, w64(j * (32 / R)));
      const v2f zk = z[j], zm = rc[R - 1 - j - H];
      const v2f e2 = add_conj(zk, zm);
      const v2f dd = sub_conj(zk, zm);
      const v2f tw = cmul_mi(wk, dd);                 // W * (-i (Zk - conj Zm))
      v2f xk = (e2 + tw) * half_scale;
      v2f xm = (e2 - tw) * v2f{half_scale, -half_scale};   // conj(...)
      v2f pk = v2f{pp[j].x, pp[j].y}, pm = v2f{pp[j].z, pp[j].w};
      v2f uk = v2f{0.0f, 0.0f}, um = v2f{0.0f, 0.0f}, sk = v2f{0.0f, 0.0f}, sm = v2f{0.0f, 0.0f};
      const float mk = (j & 1) ? mm[j / 2].z : mm[j / 2].x;
      const float mq = (j & 1) ? mm[j / 2].w : mm[j / 2].y;
      v2f ak = update_bin<MODE, EVAL>(xk, pk, uk, sk, mk, a, live, sd, so);
      v2f am = update_bin<MODE, EVAL>(xm, pm, um, sm, mq, a, live, sd, so);
      if (live) {
        st_stream(&pout[j * 64u + ulane], v4f{pk.x, pk.y, pm.x, pm.y});
        if (keep_xu) {
          st_stream(&a.X_out[fi * (H * 64) + j * 64u + ulane], v4f{sk.x, sk.y, sm.x, sm.y});
          st_stream(&a.U_out[fi * (H * 64) + j * 64u + ulane], v4f{uk.x, uk.y, um.x, um.y});
        }
      }
      if (j == 0 && lane == 0) {   // bins 0 and M: irfft ignores their imaginary parts
        ak.y = 0.0f;
        am.y = 0.0f;
      }
      const v2f e2i = add_conj(ak, am);
      const v2f o2i = cmulc(sub_conj(ak, am), wk);
      z[j] = add_i(e2i, o2i);
      back[j] = conj_sub_i(e2i, o2i);
    }
    // ---- bin M/2 (lane 0): X = conj(Z), Z'' = 2 conj(X')
    v2f zmid;
    {
      v2f xmid = z[H] * v2f{a.fwd_scale, -a.fwd_scale};
      const bool live0 = live && lane == 0;
      v2f smid = v2f{0.0f, 0.0f};
      const v2f am = update_bin<MODE, EVAL>(xmid, pmid, umid, smid, mmid, a, live0, sd, so);
      if (live0) {
        a.Pmid_out[fi] = pmid;
        if (keep_xu) {
          a.Xmid_out[fi] = smid;
          a.Umid_out[fi] = umid;
        }
      }
      zmid = am * v2f{2.0f, -2.0f};
    }
    // ---- return the mirrored halves
#pragma unroll
    for (int m = H; m < R; ++m) {
      const v2f got = shfl2(back[R - 1 - m], k.partner);
      const v2f l0 = (m == H) ? zmid : back[(R - m) % H];
      z[m] = v2f{lane == 0 ? l0.x : got.x, lane == 0 ? l0.y : got.y};
    }

#if SPECINV_ABLATE & 4
#elif SPECINV_TW_REGS
    fft_inverse_t<R>(z, k, twr, tr);
#else
    asm volatile("" ::: "memory");   // re-read the twiddles instead of keeping them live since the forward FFT
    fft_inverse<R>(z, k, lds_tw1, tr);
#endif

    // ---- synthesis window, register overlap-add, one finished hop-block out
#pragma unroll
    for (int u = 0; u < R; ++u) z[u] = z[u] * lds_win[64 * u + lane];
    if (live && t >= PB) {
      const long long o0 = (long long)(t - PB) * HOP;
      const v2f* envp = reinterpret_cast<const v2f*>(a.inv_env + o0);   // uniform
      v2f* outp = reinterpret_cast<v2f*>(orow + o0);
#pragma unroll
      for (int i = 0; i < QU; ++i) outp[64u * i + ulane] = env_apply(acc[i] + z[i], envp[64u * i + ulane]);
    }
#pragma unroll
    for (int i = 0; i < QU; ++i) {
#pragma unroll
      for (int q = 0; q + 1 < NB; ++q) acc[q * QU + i] = acc[(q + 1) * QU + i] + z[(q + 1) * QU + i];
      acc[(NB - 1) * QU + i] = z[NB * QU + i];
    }
  }
  if (t_end == a.T) {
    // the chunk that holds the last frame also finishes hop-blocks T .. T + PB - 2 (the frames that reach them are done)
#pragma unroll
    for (int q = 0; q < PB - 1; ++q) {
      const long long o0 = (long long)(a.T + q - PB) * HOP;
      const v2f* envp = reinterpret_cast<const v2f*>(a.inv_env + o0);
      v2f* outp = reinterpret_cast<v2f*>(orow + o0);
#pragma unroll
      for (int i = 0; i < QU; ++i) outp[64u * i + ulane] = env_apply(acc[q * QU + i], envp[64u * i + ulane]);
    }
  } else {
    // what this chunk's last NB frames contribute to the next chunk's first NB hop-blocks
    v2f* tl = reinterpret_cast<v2f*>(a.xtail_out + ((long long)b * a.nchunks + c) * NB * HOP);
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      const v2f* envp = reinterpret_cast<const v2f*>(a.inv_env + (long long)(t_end + q - PB) * HOP);
#pragma unroll
      for (int i = 0; i < QU; ++i) tl[(q * QU + i) * 64u + ulane] = env_apply(acc[q * QU + i], envp[64u * i + ulane]);
    }
  }
  if (EVAL) {
    const double d = wave_sum(sd), o = wave_sum(so);
    if (lane == 0) {
      a.partials[2 * (long long)w] = d;
      a.partials[2 * (long long)w + 1] = o;
    }
  }
}

// ISTFT of a spectrum held in pair layout: x = overlap-add(w * irfft(S)) / envelope  (methods.py:233: the
// initial signal of griffin_lim / ADMM).  Same wave-per-chunk walk as k_fused, without the analysis half.
template <int R, int OV>
__global__ __launch_bounds__(256, R >= 32 ? 1 : SPECINV_MINWAVES) void k_fused_istft(FastArgs a) {
  using G = Geo<R>;
  using O = Ovl<R, OV>;
  constexpr int H = G::H, M = G::M, QU = O::QU, HOP = O::HOP, NB = O::NB, PB = O::PB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  v2f* lds_win = reinterpret_cast<v2f*>(smem);
  v2f* lds_tw1 = lds_win + M;
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  v2f* tr = lds_tw1 + (R - 1) * 64 + wib * G::TR;
  for (int i = threadIdx.x; i < M; i += blockDim.x) lds_win[i] = v2f{a.window[2 * i], a.window[2 * i + 1]};
  for (int i = threadIdx.x; i < (R - 1) * 64; i += blockDim.x) {
    const int k1 = i / 64 + 1, l = i & 63;
    lds_tw1[i] = unit(2.0f * (float)((l * k1) % M) / (float)M);
  }
  __syncthreads();
  const int w = blockIdx.x * (blockDim.x >> 6) + wib;
  if (w >= a.n_waves) return;
  const LaneConst<R> k = lane_consts<R>();
  const int lane = k.lane;
  const unsigned ulane = (unsigned)lane;
  const int b = w / a.nchunks, c = w - b * a.nchunks;
  const int t_begin = chunk_begin(c, a.T, a.nchunks);
  const int t_end = chunk_begin(c + 1, a.T, a.nchunks);
  const int t_start = max(0, t_begin - NB);      // one-off kernel: recompute the NB-frame halo, write whole blocks
  float* orow = a.x_out + (long long)b * a.L;
  v2f acc[NB * QU];
#pragma unroll
  for (int i = 0; i < NB * QU; ++i) acc[i] = v2f{0.0f, 0.0f};
  for (int t = t_start; t < t_end; ++t) {
    asm volatile("" ::: "memory");
    v2f wn = k.wn;
    asm volatile("" : "+v"(wn));
    const bool live = t >= t_begin;
    const long long fi = (long long)b * a.T + t;
    const v4f* pin = a.P_in + fi * (H * 64);
    v4f pp[H];
#pragma unroll
    for (int j = 0; j < H; ++j) pp[j] = pin[j * 64u + ulane];
    v2f pmid = v2f{0.0f, 0.0f};
    if (lane == 0) pmid = a.Pmid_in[fi];
    v2f z[R], back[H];
#pragma unroll
    for (int j = 0; j < H; ++j) {
      const v2f wk = j == 0 ? wn : cmul_k(wn, w64(j * (32 / R)));
      v2f ak = v2f{pp[j].x, pp[j].y} * a.inv_scale, am = v2f{pp[j].z, pp[j].w} * a.inv_scale;
      if (j == 0 && lane == 0) {
        ak.y = 0.0f;
        am.y = 0.0f;
      }
      const v2f e2i = add_conj(ak, am);
      const v2f o2i = cmulc(sub_conj(ak, am), wk);
      z[j] = add_i(e2i, o2i);
      back[j] = conj_sub_i(e2i, o2i);
    }
    const v2f zmid = pmid * v2f{2.0f * a.inv_scale, -2.0f * a.inv_scale};
#pragma unroll
    for (int m = H; m < R; ++m) {
      const v2f got = shfl2(back[R - 1 - m], k.partner);
      const v2f l0 = (m == H) ? zmid : back[(R - m) % H];
      z[m] = v2f{lane == 0 ? l0.x : got.x, lane == 0 ? l0.y : got.y};
    }
    fft_inverse<R>(z, k, lds_tw1, tr);
#pragma unroll
    for (int u = 0; u < R; ++u) z[u] = z[u] * lds_win[64 * u + lane];
    if (live && t >= PB) {
      const long long o0 = (long long)(t - PB) * HOP;
      const v2f* envp = reinterpret_cast<const v2f*>(a.inv_env + o0);
      v2f* outp = reinterpret_cast<v2f*>(orow + o0);
#pragma unroll
      for (int i = 0; i < QU; ++i) outp[64u * i + ulane] = env_apply(acc[i] + z[i], envp[64u * i + ulane]);
    }
#pragma unroll
    for (int i = 0; i < QU; ++i) {
#pragma unroll
      for (int q = 0; q + 1 < NB; ++q) acc[q * QU + i] = acc[(q + 1) * QU + i] + z[(q + 1) * QU + i];
      acc[(NB - 1) * QU + i] = z[NB * QU + i];
    }
  }
  if (t_end == a.T) {
#pragma unroll
    for (int q = 0; q < PB - 1; ++q) {
      const long long o0 = (long long)(a.T + q - PB) * HOP;
      const v2f* envp = reinterpret_cast<const v2f*>(a.inv_env + o0);
      v2f* outp = reinterpret_cast<v2f*>(orow + o0);
#pragma unroll
      for (int i = 0; i < QU; ++i) outp[64u * i + ulane] = env_apply(acc[q * QU + i], envp[64u * i + ulane]);
    }
  }
}

// User layout (B, F, T) -> pair layout in one pass (32 x 32 tile transposed through LDS): reads are contiguous in
// time, writes are 8-byte (spectrum) / 4-byte (magnitude) pieces of the 16-byte pair records, contiguous in k.
// Bin f goes to pair k = f (first half) for f < M/2, to pair k = M - f (second half) for f > M/2, to `mid` for M/2.
template <int R>
__global__ void k_user_spec_to_pairs(const v2f* __restrict__ in, v2f* __restrict__ pairs /* v4f records as 2 x v2f */,
                                     v2f* __restrict__ mid, int T) {
  using G = Geo<R>;
  constexpr int F = G::M + 1;
  __shared__ v2f tile[32][33];
  const int b = blockIdx.z, f0 = blockIdx.y * 32, t0 = blockIdx.x * 32;
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    const int f = f0 + i, t = t0 + threadIdx.x;
    if (f < F && t < T) tile[i][threadIdx.x] = in[((long long)b * F + f) * T + t];
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    const int t = t0 + i, f = f0 + threadIdx.x;
    if (f >= F || t >= T) continue;
    const long long fr = (long long)b * T + t;
    const v2f v = tile[threadIdx.x][i];
    if (2 * f == G::M) {
      mid[fr] = v;
    } else {
      const int kk = f < G::M / 2 ? f : G::M - f, half = f < G::M / 2 ? 0 : 1;
      pairs[(((fr * G::H) + (kk >> 6)) * 64 + (kk & 63)) * 2 + half] = v;
    }
  }
}

// same for the target magnitude; also per-block partial sums of m^2 (for the metrics)
template <int R>
__global__ void k_user_mag_to_pairs(const float* __restrict__ in, float* __restrict__ pairs /* v4f records */,
                                    float* __restrict__ mid, int T, double* __restrict__ partials) {
  using G = Geo<R>;
  constexpr int F = G::M + 1;
  __shared__ float tile[32][33];
  __shared__ double red[4];
  const int b = blockIdx.z, f0 = blockIdx.y * 32, t0 = blockIdx.x * 32;
  double s2 = 0.0;
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    const int f = f0 + i, t = t0 + threadIdx.x;
    if (f < F && t < T) {
      const float v = in[((long long)b * F + f) * T + t];
      tile[i][threadIdx.x] = v;
      s2 += (double)v * (double)v;
    }
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    const int t = t0 + i, f = f0 + threadIdx.x;
    if (f >= F || t >= T) continue;
    const long long fr = (long long)b * T + t;
    const float v = tile[threadIdx.x][i];
    if (2 * f == G::M) {
      mid[fr] = v;
    } else {
      const int kk = f < G::M / 2 ? f : G::M - f, second = f < G::M / 2 ? 0 : 1;
      const int j = kk >> 6;
      pairs[(((fr * (G::H / 2)) + (j >> 1)) * 64 + (kk & 63)) * 4 + (j & 1) * 2 + second] = v;
    }
  }
  s2 = wave_sum(s2);
  const int tid = threadIdx.y * blockDim.x + threadIdx.x;
  if ((tid & 63) == 0) red[tid >> 6] = s2;
  __syncthreads();
  if (tid == 0) {
    double tot = 0.0;
    for (int w = 0; w < (int)((blockDim.x * blockDim.y + 63) >> 6); ++w) tot += red[w];
    partials[((long long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = tot;
  }
}

// phase_init (methods.py:572-615) for the fused path: the starting spectrum and the target magnitude written straight in
// conjugate-pair order (what k_phase_init + k_user_spec_to_pairs + k_user_mag_to_pairs produce in three passes and one
// (B, F, T) complex round trip).  One workgroup per (item, pair row j): its 128 spectrogram rows - bins 64 j + l and
// M - (64 j + l) - are scanned over time exactly like k_phase_init does it (a wave per row, lanes = 64 consecutive time
// steps, float64 wave scan with each partial sum rounded to float32, the same operation order), 64 time steps at a time;
// the 128 x 64 block is transposed through LDS and leaves as 64 records of 1 KiB.  The bin M/2 rides with the last pair row.
template <int R>
__global__ __launch_bounds__(1024) void k_phase_init_pairs(const float* __restrict__ mag, v4f* __restrict__ P, v2f* __restrict__ Pmid,
                                                          float* __restrict__ mpairs, float* __restrict__ mmid,
                                                          double* __restrict__ partials, int T, int hop) {
  using G = Geo<R>;
  constexpr int M = G::M, F = M + 1, H = G::H, NFFT = G::N, LD = 129, WAVES = 16, RPW = 128 / WAVES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  v2f* ct = reinterpret_cast<v2f*>(smem);                   // [64 time steps][LD] complex values, column = row slot
  float* mt = reinterpret_cast<float*>(ct + 64 * LD);       // [64][LD] magnitudes
  __shared__ double red[16];
  __shared__ double carry_s[WAVES][RPW + 1];                // running phase of every row (the row loop is not unrolled)
  const int b = blockIdx.x / H, j = blockIdx.x - b * H;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float* base = mag + (long long)b * F * T;
  const float two_pi = 6.283185307179586476925286766559f;
  const bool has_mid = j == H - 1 && wave == 0;
  double* carry = carry_s[wave];
  if (lane <= RPW) carry[lane] = 0.0;
  double s2 = 0.0;

  // one row, 64 time steps: returns the complex value and the magnitude of (f, t0 + lane)
  auto row_step = [&](int f, int t, double* cr, v2f& val, float& m0) {
    float cur[5];
#pragma unroll
    for (int d = 0; d < 5; ++d) {
      const int g = f + d - 2;
      cur[d] = (t < T && g >= 0 && g < F) ? base[(long long)g * T + t] : 0.0f;
    }
    float om = 0.0f;
    m0 = cur[2];
    if (t < T) {
      float w;
      // scatter order :607-609: own bin, else the k+1 write of a peak below, else the k-1 write of a peak above
      if (peak_omega_vals<float>(cur[1], cur[2], cur[3], f, F, two_pi, (float)NFFT, (float)hop, w)) om = w;
      else if (peak_omega_vals<float>(cur[0], cur[1], cur[2], f - 1, F, two_pi, (float)NFFT, (float)hop, w)) om = w;
      else if (peak_omega_vals<float>(cur[2], cur[3], cur[4], f + 1, F, two_pi, (float)NFFT, (float)hop, w)) om = w;
    }
    double v = (double)om;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const double u = __shfl_up(v, off, 64);
      if (lane >= off) v += u;
    }
    v += *cr;                                               // (wave-private LDS slot: in-order within the wave)
    if (lane == 63) *cr = v;
    const float phi = (float)v;                              // :611
    double sn, cs;
    sincos((double)phi, &sn, &cs);                           // :612
    val = v2f{m0 * (float)cs, m0 * (float)sn};               // :614
  };

  for (int t0 = 0; t0 < T; t0 += 64) {
    const int t = t0 + lane;
#pragma unroll 1
    for (int i = 0; i < RPW; ++i) {
      const int s = wave * RPW + i;                          // row slot: 0..63 bins 64 j + s, 64..127 bins M - (64 j + s - 64)
      const int f = s < 64 ? 64 * j + s : M - (64 * j + s - 64);
      v2f val;
      float m0;
      row_step(f, t, carry + i, val, m0);
      ct[lane * LD + s] = val;
      mt[lane * LD + s] = m0;
      if (t < T) s2 += (double)m0 * (double)m0;
    }
    if (has_mid) {                                           // bin M/2: time-major arrays, written as they come
      v2f val;
      float m0;
      row_step(M / 2, t, carry + RPW, val, m0);
      if (t < T) {
        Pmid[(long long)b * T + t] = val;
        mmid[(long long)b * T + t] = m0;
        s2 += (double)m0 * (double)m0;
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 64 / WAVES; ++q) {
      const int tl = wave * (64 / WAVES) + q, tt = t0 + tl;
      if (tt < T) {
        const long long fr = (long long)b * T + tt;
        const v2f a = ct[tl * LD + lane], bb = ct[tl * LD + 64 + lane];
        P[(fr * H + j) * 64 + lane] = v4f{a.x, a.y, bb.x, bb.y};
        // target record c = j / 2 holds (m[k_2c], m[M - k_2c], m[k_2c+1], m[M - k_2c+1]): this row fills one half of it
        *reinterpret_cast<v2f*>(mpairs + ((fr * (H / 2) + (j >> 1)) * 64 + lane) * 4 + (j & 1) * 2) =
            v2f{mt[tl * LD + lane], mt[tl * LD + 64 + lane]};
      }
    }
    __syncthreads();
  }
  const double tot = block_sum(s2, red);
  if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

}  // namespace fast
}  // namespace specinv

#include "kernels_frame.h"   // stand-alone transforms and the any-hop frame kernels (k_semi, k_hop, k_hop_inverse)

namespace specinv {
namespace fast {

// x += the tail partial sums (final waveform for get_wave)
template <int R, int OV>
__global__ void k_add_tails(float* __restrict__ x, const float* __restrict__ xtail, int T, int nchunks, long long L,
                            long long total) {
  constexpr int HOP = Ovl<R, OV>::HOP, NB = Ovl<R, OV>::NB, PB = Ovl<R, OV>::PB;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // (b, c, q, sample) over tails
  if (i >= total) return;
  const int smp = i % HOP;
  const int q = (i / HOP) % NB;
  const int c = (i / (NB * HOP)) % nchunks;
  const long long b = i / ((long long)NB * HOP * nchunks);
  if (c >= nchunks - 1) return;                       // the last chunk has no successor
  const int blk = chunk_begin(c + 1, T, nchunks) + q; // padded-signal hop-block
  // register layout of a block: element (reg i2, lane l, comp e) <-> sample 128*i2 + 2*l + e
  const int i2 = smp / 128, rem = smp % 128;
  x[b * L + (long long)(blk - PB) * HOP + smp] += xtail[((b * nchunks + c) * NB + q) * HOP + (i2 * 64 + rem / 2) * 2 + (rem & 1)];
}

// ---- layout conversion between the frame-major (B*T, F) spectra and the pair layout ---------------
template <int R>
__global__ void k_spec_to_pairs(const v2f* __restrict__ spec, v4f* __restrict__ pairs, v2f* __restrict__ mid,
                                long long n_frames) {
  using G = Geo<R>;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // (frame, j, lane)
  if (i >= n_frames * G::H * 64) return;
  const int lane = i & 63;
  const int j = (i >> 6) % G::H;
  const long long f = i / (64 * G::H);
  const v2f* s = spec + f * (G::M + 1);
  const int kk = lane + 64 * j;
  const v2f a = s[kk], bb = s[G::M - kk];
  pairs[i] = v4f{a.x, a.y, bb.x, bb.y};
  if (lane == 0 && j == 0) mid[f] = s[G::M / 2];
}

template <int R>
__global__ void k_pairs_to_spec(const v4f* __restrict__ pairs, const v2f* __restrict__ mid, v2f* __restrict__ spec,
                                long long n_frames) {
  using G = Geo<R>;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_frames * G::H * 64) return;
  const int lane = i & 63;
  const int j = (i >> 6) % G::H;
  const long long f = i / (64 * G::H);
  v2f* s = spec + f * (G::M + 1);
  const int kk = lane + 64 * j;
  const v4f p = pairs[i];
  s[kk] = v2f{p.x, p.y};
  s[G::M - kk] = v2f{p.z, p.w};
  if (lane == 0 && j == 0) s[G::M / 2] = mid[f];
}

template <int R>
__global__ void k_mag_to_pairs(const float* __restrict__ mag, v4f* __restrict__ pairs, float* __restrict__ mid,
                               long long n_frames) {
  using G = Geo<R>;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // (frame, c, lane)
  if (i >= n_frames * (G::H / 2) * 64) return;
  const int lane = i & 63;
  const int c = (i >> 6) % (G::H / 2);
  const long long f = i / (64 * (G::H / 2));
  const float* s = mag + f * (G::M + 1);
  const int k0 = lane + 64 * (2 * c), k1 = lane + 64 * (2 * c + 1);
  pairs[i] = v4f{s[k0], s[G::M - k0], s[k1], s[G::M - k1]};
  if (lane == 0 && c == 0) mid[f] = s[G::M / 2];
}

__global__ void k_reciprocal(const float* __restrict__ in, float* __restrict__ out, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = SPECINV_IEEE ? in[i] : 1.0f / in[i];
}

}  // namespace fast

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
  bool keep_state = false;
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
  int cur = 0;   // index of the buffers holding the current state
  int mode = fast::MODE_GLA;
  FastBuf xb[2], xtail[2], Pb[2], Pmid[2], mpairs, mmid, inv_env, scratch;
  // ADMM carries Y = X + U in Pb (FastArgs).  X and U themselves are only written when the caller has asked for them
  // (specinv_plan_keep_state), by the last iteration of every iterate() call.
  bool keep_state = false, xu_valid = false;
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
    if (cfg.dtype != SPECINV_F32 || !cfg.onesided) return SPECINV_OK;
    if (const char* e = getenv("SPECINV_DISABLE_FAST")) {
      if (e[0] == '1') return SPECINV_OK;
    }
    if (cfg.n_fft == 512 || cfg.n_fft == 1024 || cfg.n_fft == 2048 || cfg.n_fft == 4096) {
      xform_ok = true;              // any hop, any pad mode, centred or not
      xform_R = cfg.n_fft / 128;
    }
    if (!xform_ok) return SPECINV_OK;
    R = xform_R;
    semi = false;
    state_in_place = true;     // (same speed as ping-pong buffers, measured; a third less memory)
    if (const char* e = getenv("SPECINV_STATE_INPLACE")) state_in_place = e[0] != '0';
    use_template = false;
    if (const char* e = getenv("SPECINV_FUSED_TEMPLATE")) use_template = e[0] == '1';
    // fused kernel: hop = n_fft / 2, / 4 or / 8 (whole registers per hop-block), centred, enough frames
    OV = 0;
    for (int o : {2, 4, 8})
      if (cfg.hop_length * o == cfg.n_fft && R % o == 0) OV = o;
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
      // (a wave walks >= 8 frames one after the other there: measured against k_semi + k_ola, it pays from ~16 k frames
      // at n_fft 2048 - 0.126 vs 0.142 ms per iteration - and ~32 k frames at 1024 and 512)
      long long hop_from = R >= 16 ? 16384 : 32768;
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
    td = md == fast::MODE_GLA && (!semi || hopk) && !use_template && !keep_state && RR <= 16;
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
    SI_TRY(inv_env.reserve(pl.length * sizeof(float)));
    cur = 0;
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
      const dim3 grid((pl.Tn() + 31) / 32, (pl.n_freq + 31) / 32, pl.B()), blk(32, 8);
      hipLaunchKernelGGL((fast::k_user_spec_to_pairs<RR>), grid, blk, 0, pl.stream, spec_user, Pb[0].template as<v2f>(),
                         Pmid[0].template as<v2f>(), pl.Tn());
      SI_HIP(hipGetLastError());
      const long long nblk = (long long)grid.x * grid.y * grid.z;
      SI_TRY(pl.partials.reserve(std::max<size_t>((size_t)nblk, 3 * 1024) * sizeof(double)));
      hipLaunchKernelGGL((fast::k_user_mag_to_pairs<RR>), grid, blk, 0, pl.stream, mag_user, mpairs.template as<float>(),
                         mmid.template as<float>(), pl.Tn(), pl.partials.template as<double>());
      SI_HIP(hipGetLastError());
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
                         pl.env.template as<float>(), inv_env.template as<float>(), (long long)pl.length);
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
                       pl.env.template as<float>(), inv_env.template as<float>(), (long long)pl.length);
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
    if (SPECINV_R8_W3 && R == 8 && OV == 4 && !use_template && n_waves >= 3072) return 12;
    // (the signal-form kernel at n_fft 2048 measured 2 % faster with two 4-wave workgroups per CU than with one 8-wave one:
    // C2 25.8 vs 26.3 ms per step on one box, three runs each - the opposite of k_fused4)
    if (td && R == 16 && OV == 4) return 4;
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
    SI_CHECK(fn != nullptr, SPECINV_EUNSUPPORTED, "no fused kernel for n_fft / hop = %d", OV);
    const int wgw = fused_wgw();
    const size_t lds_used = G::lds_bytes(wgw);
    SI_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_used));
    fast::FastArgs args = a;
    void* kargs[] = {&args};
    SI_HIP(hipLaunchKernel(fn, dim3((n_waves + wgw - 1) / wgw), dim3(64 * wgw), kargs, lds_used, pl.stream));
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
    SPECINV_R_SWITCH(R, lds_used = fast::Geo<RR>::lds_bytes(wgw);
                     if constexpr (RR % 8 == 0) { if (OV == 8) fn = td_kernel<RR, 8>(early, ev); }
                     if constexpr (RR == 8 || RR == 16) { if (OV == 4) fn = td_kernel4<RR>(early, ev); }
                     else { if (OV == 4) fn = td_kernel<RR, 4>(early, ev); }
                     if (OV == 2) fn = td_kernel<RR, 2>(early, ev));
    SI_CHECK(fn != nullptr, SPECINV_EUNSUPPORTED, "no fused kernel for n_fft / hop = %d", OV);
    SI_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_used));
    fast::FastArgs args = a;
#if SPECINV_TD_STAMPS
    static unsigned long long* d_stamps = nullptr;
    if (!d_stamps) SI_HIP(hipMalloc(&d_stamps, (size_t)n_waves * 8 * sizeof(unsigned long long)));
    args.stamps = d_stamps;
#endif
    void* kargs[] = {&args};
    SI_HIP(hipLaunchKernel(fn, dim3((n_waves + wgw - 1) / wgw), dim3(64 * wgw), kargs, lds_used, pl.stream));
#if SPECINV_TD_STAMPS
    if (td_t == 40 || td_t == 5) {
      std::vector<unsigned long long> h((size_t)n_waves * 8);
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
    SI_HIP(hipFuncSetAttribute((const void*)fast::k_semi<RR, MODE, EVAL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((fast::k_semi<RR, MODE, EVAL>), dim3(semi_grid), dim3(256), lds, pl.stream, s);
    SI_HIP(hipGetLastError());
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
    s.env = inv_env.template as<float>();
    s.xtail = xtail[0].template as<float>();
    s.hop = hop;
    s.pad = pl.pad;
    const size_t lds = G::lds_bytes(wgw) + (size_t)wgw * G::N * sizeof(float);
    const void* fn = (const void*)fast::k_hop<RR, MODE, EVAL>;
    SI_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    void* kargs[] = {&s};
    SI_HIP(hipLaunchKernel(fn, dim3((n_waves + wgw - 1) / wgw), dim3(64 * wgw), kargs, lds, pl.stream));
    if (nchunks > 1 && keep > 0) {
      const long long total = (long long)pl.B() * (nchunks - 1) * keep;
      hipLaunchKernelGGL(fast::k_hop_tails, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, pl.stream, a.x_out,
                         (const float*)s.xtail, s.env, pl.Tn(), nchunks, hop, keep, pl.pad, (long long)pl.length, total);
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
    SI_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    void* kargs[] = {&s};
    SI_HIP(hipLaunchKernel(fn, dim3((n_waves + wgw - 1) / wgw), dim3(64 * wgw), kargs, lds, pl.stream));
    if (nchunks > 1 && keep > 0) {
      const long long total = (long long)pl.B() * (nchunks - 1) * keep;
      hipLaunchKernelGGL(fast::k_hop_tails_td, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, pl.stream, a.x2_out, a.x_out,
                         a.x_in, (const float*)s.xtail, s.env, a.coef, pl.Tn(), nchunks, hop, keep, pl.pad, (long long)pl.length,
                         total);
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
    SI_TRY(pl.partials.reserve(std::max<size_t>((size_t)n_waves * 2, 3 * 1024) * sizeof(double)));
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
    n_partials = n_waves;
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
      void* kargs[] = {&xo, &tl, &Tn, &nc, &Ln, &tot};
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
    SPECINV_R_SWITCH(R, const long long np = nf * fast::Geo<RR>::H * 64;
                     hipLaunchKernelGGL((fast::k_pairs_to_spec<RR>), dim3((unsigned)ceil_div(np, 256)), dim3(256), 0, pl.stream,
                                        src.template as<v4f>(), mid.template as<v2f>(), scratch.template as<v2f>(), nf));
    SI_HIP(hipGetLastError());
    return pl.template transpose<cplx<float>>(scratch.template as<cplx<float>>(), out, pl.Tn(), pl.n_freq);
  }
};

}  // namespace specinv
