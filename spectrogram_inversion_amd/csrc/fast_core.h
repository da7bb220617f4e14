// Fused gfx950 fast path: one Griffin-Lim / ADMM iteration = ONE kernel launch.  This header holds what every wave-level kernel
// shares - the FFT across one wave, the conjugate-pair update, the block / frame loaders, the argument records - and the
// declarations of the kernels themselves; their bodies are kernels_fused.h (spectral state), kernels_fast_td.h (momentum as a
// signal), kernels_frame.h (any hop), kernels_rtisi_fast.h and kernels_objective.h, each compiled in its own tu_*.hip; the host
// side is fast_state.h.
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
#ifndef SPECINV_IEEE        // 1 (default): the reference's operation order in the projection, (s m) r with r the correctly rounded
#define SPECINV_IEEE 1      //    1 / |s|, and a true division by the envelope (methods.py:132,246-247; ref_rcp_abs2 below);
#endif                      // 0: s (m v_rsq_f32(|s|^2 + 1e-32)) and a multiplication by 1 / envelope (the fast_approx copy)

#ifndef SPECINV_REFCHAIN    // (experiments) 1: RN(1 / (RN(sqrt t) + 1e-16)), both roundings; 2: one Newton step to RN(t^-1/2)
#define SPECINV_REFCHAIN 2
#endif
#ifndef SPECINV_REFBREADTH  // (experiments) 1: the chain written breadth-first over a frame's pairs in k_fused_td
#define SPECINV_REFBREADTH 1
#endif
#ifndef SPECINV_K4_ENVREG
#define SPECINV_K4_ENVREG 1
#endif
#ifndef SPECINV_R8_W3        // n_fft 1024: three waves per SIMD (3072 wave slots; 12-wave workgroups at hop 256).  The plain launches fit
#define SPECINV_R8_W3 1      // 168 registers (ADMM 2 spilled, the evaluating variants 8-31); measured against two waves per SIMD:
#endif                       // C4 34.3 -> 32.3 ms per step, Griffin-Lim 1024 / 256 0.135 -> 0.127 ms per iteration

// The wave-level kernels are compiled twice: as `specinv::fast` with the reference's operation order and correctly rounded factors
// (SPECINV_IEEE=1, the default arithmetic since round 4: + 3 % on the headline step, tools/refchain_study.py), and - in the
// tu_approx_*.hip units, which define SPECINV_IEEE=0 and SI_FAST_NS=fast_approx before including the kernel headers - as
// `specinv::fast_approx` with the hardware's approximate inverse square root in the projection and a multiplication by
// 1 / envelope.  `specinv_plan_set_exact(plan, 0)` selects the second set (fast_state.h takes its kernels' addresses from the
// extern "C" tables of those units).
#ifndef SI_FAST_NS
#define SI_FAST_NS fast
#endif

namespace specinv {
namespace SI_FAST_NS {

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
  v2f t, d;   // (one asm statement: between two the compiler puts an s_nop, which costs an issue slot)
  asm("v_pk_mul_f32 %1, %2, %3 op_sel:[0,0] op_sel_hi:[0,1]\n\t"                                         // (a.x b.x, a.x b.y)
      "v_pk_fma_f32 %0, %2, %3, %1 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"                       // (- a.y b.y, + a.y b.x)
      : "=v"(d), "=&v"(t) : "v"(a), "v"(b));
  return d;
#else
  return cmul_k(a, b);
#endif
}
// a * conj(b)
__device__ __forceinline__ v2f cmulc(v2f a, v2f b) {
#if SPECINV_ASM_CMUL
  v2f t, d;
  asm("v_pk_mul_f32 %1, %2, %3 op_sel:[0,0] op_sel_hi:[0,1] neg_hi:[1,0]\n\t"                             // (a.x b.x, - a.x b.y)
      "v_pk_fma_f32 %0, %2, %3, %1 op_sel:[1,1,0] op_sel_hi:[1,0,1]"                                      // (+ a.y b.y, + a.y b.x)
      : "=v"(d), "=&v"(t) : "v"(a), "v"(b));
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
  asm("v_pk_mul_f32 %1, %2, %3 op_sel:[1,0] op_sel_hi:[1,1]\n\t"                                         // (w.y d.x, w.y d.y)
      "v_pk_fma_f32 %0, %2, %3, %1 op_sel:[0,1,0] op_sel_hi:[0,0,1] neg_hi:[1,0,0]"                       // (+ w.x d.y, - w.x d.x)
      : "=v"(r), "=&v"(t) : "v"(w), "v"(d));
  return r;
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
// a * w (INV: a * conj(w)) with a compile-time constant w: the constant sits in a scalar register pair and the product is TWO
// packed operations like cmul (VOP3P takes no literal, but one scalar source) instead of the four scalar ones with inline
// literals; the in-register DFTs' fixed twiddles and the W_64^j steps of the real-FFT split are 43 such products per frame
// and iteration at n_fft 2048 (86 of ~1240 vector instructions)
#ifndef SPECINV_PKCONST
#define SPECINV_PKCONST 1
#endif
template <bool INV>
__device__ __forceinline__ v2f cmul_sk(v2f a, v2f w) {
  v2f t, d;
  if (!INV) {
    asm("v_pk_mul_f32 %1, %2, %3 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %2, %3, %1 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"
        : "=v"(d), "=&v"(t) : "v"(a), "s"(w));
  } else {
    asm("v_pk_mul_f32 %1, %2, %3 op_sel:[0,0] op_sel_hi:[0,1] neg_hi:[1,0]\n\t"
        "v_pk_fma_f32 %0, %2, %3, %1 op_sel:[1,1,0] op_sel_hi:[1,0,1]"
        : "=v"(d), "=&v"(t) : "v"(a), "s"(w));
  }
  return d;
}
// four of them in one block: the four multiplies first, then the four dependent multiply-adds (cmul_x4)
template <bool INV>
__device__ __forceinline__ void cmul_sk_x4(v2f& a0, v2f& a1, v2f& a2, v2f& a3, v2f w0, v2f w1, v2f w2, v2f w3) {
  v2f t0, t1, t2, t3;
  if (!INV) {
    asm("v_pk_mul_f32 %4, %0, %8 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %5, %1, %9 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %6, %2, %10 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %7, %3, %11 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %0, %8, %4 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"
        "v_pk_fma_f32 %1, %1, %9, %5 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"
        "v_pk_fma_f32 %2, %2, %10, %6 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"
        "v_pk_fma_f32 %3, %3, %11, %7 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "s"(w0), "s"(w1), "s"(w2), "s"(w3));
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
        : "s"(w0), "s"(w1), "s"(w2), "s"(w3));
  }
}
template <bool INV, bool PK>
__device__ __forceinline__ void dirmul_x4(v2f& a0, v2f& a1, v2f& a2, v2f& a3, v2f w0, v2f w1, v2f w2, v2f w3) {
  if (PK && SPECINV_PKCONST) {
    cmul_sk_x4<INV>(a0, a1, a2, a3, w0, w1, w2, w3);
  } else {
    a0 = dirmul<INV>(a0, w0);
    a1 = dirmul<INV>(a1, w1);
    a2 = dirmul<INV>(a2, w2);
    a3 = dirmul<INV>(a3, w3);
  }
}
template <bool INV, bool PK>
__device__ __forceinline__ v2f dirmul_p(v2f a, v2f w) {
  if (PK && SPECINV_PKCONST) return cmul_sk<INV>(a, w);
  return dirmul<INV>(a, w);
}

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

template <int R, bool INV, bool PK = true>
struct Dft;

template <bool INV, bool PK>
struct Dft<4, INV, PK> {
  static __device__ __forceinline__ void run(v2f (&a)[4]) { dft4<INV>(a[0], a[1], a[2], a[3]); }
};

template <bool INV, bool PK>
struct Dft<8, INV, PK> {
  static __device__ __forceinline__ void run(v2f (&a)[8]) {
    constexpr float h = 0.70710678118654752440f;
    // n = 4*n1 + n0: radix-2 over n1, twiddle W8^(n0*k1), radix-4 over n0
#pragma unroll
    for (int n0 = 0; n0 < 4; ++n0) {
      const v2f s = a[n0] + a[n0 + 4], d = a[n0] - a[n0 + 4];
      a[n0] = s;
      a[n0 + 4] = d;
    }
    a[5] = dirmul_p<INV, PK>(a[5], v2f{h, -h});
    a[6] = rot<INV>(a[6]);
    a[7] = dirmul_p<INV, PK>(a[7], v2f{-h, -h});
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

template <bool INV, bool PK>
struct Dft<16, INV, PK> {
  static __device__ __forceinline__ void run(v2f (&a)[16]) {
    constexpr float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, h = 0.70710678118654752440f;
    // n = 4*n1 + n0: radix-4 over n1 (in place -> slot n0 + 4*k1), twiddle W16^(n0*k1), radix-4 over n0
#pragma unroll
    for (int n0 = 0; n0 < 4; ++n0) dft4<INV>(a[n0], a[n0 + 4], a[n0 + 8], a[n0 + 12]);
    //      W16^1, W16^2, W16^3, W16^2 ; W16^6, W16^3, W16^6, W16^9 ; W16^4
    dirmul_x4<INV, PK>(a[5], a[6], a[7], a[9], v2f{c1, -s1}, v2f{h, -h}, v2f{s1, -c1}, v2f{h, -h});
    dirmul_x4<INV, PK>(a[11], a[13], a[14], a[15], v2f{-h, -h}, v2f{s1, -c1}, v2f{-h, -h}, v2f{-c1, s1});
    a[10] = rot<INV>(a[10]);
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

template <bool INV, bool PK>
struct Dft<32, INV, PK> {
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
        else a[n0 + 8 * k1] = dirmul_p<INV, PK>(a[n0 + 8 * k1], w32(e));
      }
    v2f o[32];
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) {
      v2f t[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) t[i] = a[8 * k1 + i];
      Dft<8, INV, PK>::run(t);
      // t[k0] = X[k1 + 4*k0]
#pragma unroll
      for (int k0 = 0; k0 < 8; ++k0) o[k1 + 4 * k0] = t[k0];
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) a[i] = o[i];
  }
};

// The transpose scratch of one wave: a 64 x R matrix (rows of R + 1 elements) written in layout A - lane (v, n2) holds row
// kv(v) * R + i, column n2 for register i - and read back in layout B - lane l holds row l, column i - by the forward transform,
// the other way round by the inverse.  The 64 / R groups of R rows (one per frequency digit kv) start at kv * R * (R + 1) +
// tr_pad<R>(kv).  With the odd row stride alone, n_fft 2048 (R = 16) is free of LDS bank conflicts, but at R = 8 / R = 4 the
// groups a 32-lane read group (16-lane write group) touches in layout A fall on the same banks: 2-way conflicts on every layout-A
// access (PMC, C4: SQ_LDS_BANK_CONFLICT 14 % of SQ_LDS_IDX_ACTIVE).  The pads below make the group bases distinct mod 32
// elements for the lanes of a ds_read_b64 group and mod 16 for a ds_write_b64 group in BOTH layouts (banking rules:
// MI355X_MICROARCH.md, LDS; found by exhaustive search over the residues, smallest total size).
template <int R>
__host__ __device__ constexpr int tr_pad(int kv) {
  if (R == 8) {
    constexpr int p[8] = {0, 0, 8, 24, 24, 40, 48, 48};
    return p[kv];
  }
  if (R == 4) {
    constexpr int p[16] = {0, 0, 0, 16, 20, 20, 36, 36, 40, 56, 56, 56, 76, 76, 76, 76};
    return p[kv];
  }
  return 0;
}

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
  static constexpr int TR = 64 * (R + 1) + tr_pad<R>(64 / R - 1);  // transpose scratch per wave (complex elements)
  static constexpr size_t lds_bytes(int waves) { return sizeof(v2f) * (size_t)(M + (R - 1) * 64 + waves * TR); }
  static constexpr size_t lds_bytes_td(int waves) { return lds_bytes(waves) + sizeof(v2f) * (size_t)M; }   // + the scaled synthesis window
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
  v2f postq[3];            // C == 4 (n_fft 2048): W_64^(n2*s), s = 1, 2, 3 - the lane holds all four frequency digits there (fft_forward_t)
  int tr_q;                // ... and its transpose address in that arrangement: (0, 2, 1, 3)[lane / 16] * (R + 1) + n2
  v2f stage[2];            // twiddles between the cross-lane radix-4 and radix-2 steps (C == 8: one, C == 16: two)
  v2f wn;                  // W_N^lane
  int tr_a;                // transpose address for layout A: (kv*R + reg)*(R+1) + tr_pad(kv) + n2  -> base + reg*(R+1)
  int tr_b;                // layout B: lane*(R+1) + tr_pad(lane / R) + reg
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
#pragma unroll
  for (int sgt = 1; sgt < 4; ++sgt) k.postq[sgt - 1] = unit(2.0f * (float)(k.n2 * sgt) / 64.0f);
  k.tr_q = (((k.v & 1) << 1) | ((k.v >> 1) & 1)) * (R + 1) + k.n2;   // row q ends up owning register g + (0, 2, 1, 3)[q] (swap32 then swap16)
  k.wn = unit(2.0f * (float)k.lane / (float)G::N);
  k.tr_a = (k.kv * R) * (R + 1) + tr_pad<R>(k.kv) + k.n2;
  k.tr_b = k.lane * (R + 1) + tr_pad<R>(k.lane / R);
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

// The two halves of xlane_dft4 for transforms that meet the wave's transpose scratch on one side (n_fft 2048, C == 4): after the
// swaps each 16-lane row owns ONE of the four registers, all four of its row values - in (A, C, B, D) - and after the in-lane
// radix-4 all four of its frequency digits.  The forward transform writes them to the scratch from there (every register has its
// own constant address, the lane its own base) instead of swapping them back into rows first, the inverse reads them from the
// scratch in that arrangement instead of swapping them in: 32 of the 64 lane swaps of a transform go, and the digit-0 quarter
// of the post-twiddles (a multiplication by one) with them.  The same butterflies on the same values.  Taken with the packed
// products (PK): k_rtisi_fast - a lone wave per SIMD, scalar products - measured 2.6 % slower with it (C3 126.1 vs 122.9 ms).
template <bool INV>
__device__ __forceinline__ void xlane_dft4_in(v2f& A, v2f& B, v2f& C, v2f& D) {
  swap32(A, B);
  swap32(C, D);
  swap16(A, C);
  swap16(B, D);
  dft4<INV>(A, C, B, D);   // digit s = 0..3 of the register this row owns sits in (A, C, B, D)
}
template <bool INV>
__device__ __forceinline__ void xlane_dft4_out(v2f& A, v2f& B, v2f& C, v2f& D) {
  dft4<INV>(A, C, B, D);   // in: row value q = 0..3 of the register this row owns in (A, C, B, D)
  swap16(A, C);
  swap16(B, D);
  swap32(A, B);
  swap32(C, D);
}
#ifndef SPECINV_XLANE_HALF
#define SPECINV_XLANE_HALF 1
#endif

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
// PKC: the in-register DFTs' constant twiddles as packed products with a scalar-register operand (cmul_sk) or as four scalar
// operations with literals (k_rtisi_fast keeps the scalar form throughout).
template <int R, bool PK = true, bool PKC = PK, typename TW>
__device__ __forceinline__ void fft_forward_t(v2f (&z)[R], const LaneConst<R>& k, const TW& tw, v2f* __restrict__ tr);
template <int R, bool PK = true, bool PKC = PK, typename TW>
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

template <int R, bool PK, bool PKC, typename TW>
__device__ __forceinline__ void fft_forward_t(v2f (&z)[R], const LaneConst<R>& k, const TW& tw, v2f* __restrict__ tr) {
  using G = Geo<R>;
  Dft<R, false, PKC>::run(z);
  cmul_all<R, PK, false, 1>(z, tw);
  // cross-lane radix-C over v = lane / R; afterwards lane position v holds frequency digit k.kv
  if (G::C == 2) {
#pragma unroll
    for (int g = 0; g < R; g += 2) xlane_dft2(z[g], z[g + 1]);
  } else if (G::C == 4 && SPECINV_XLANE_HALF && PK) {
    // slot (A, C, B, D) = (z[g], z[g+2], z[g+1], z[g+3]) holds digit s = 0..3 of the register g + q' this row owns
    // (q' = (0, 2, 1, 3)[lane / 16]): post-twiddle W_64^(n2 s), scratch row s R + g + q'
#pragma unroll
    for (int g = 0; g < R; g += 4) xlane_dft4_in<false>(z[g], z[g + 1], z[g + 2], z[g + 3]);
#pragma unroll
    for (int g = 0; g < R; g += 4) {
      z[g + 2] = cmul_p<PK>(z[g + 2], k.postq[0]);
      z[g + 1] = cmul_p<PK>(z[g + 1], k.postq[1]);
      z[g + 3] = cmul_p<PK>(z[g + 3], k.postq[2]);
    }
#pragma unroll
    for (int g = 0; g < R; g += 4) {
      tr[k.tr_q + (0 * R + g) * (R + 1)] = z[g];
      tr[k.tr_q + (1 * R + g) * (R + 1)] = z[g + 2];
      tr[k.tr_q + (2 * R + g) * (R + 1)] = z[g + 1];
      tr[k.tr_q + (3 * R + g) * (R + 1)] = z[g + 3];
    }
#pragma unroll
    for (int i = 0; i < R; ++i) z[i] = tr[k.tr_b + i];
    Dft<R, false, PKC>::run(z);
    return;
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
  Dft<R, false, PKC>::run(z);
}

// inverse (unnormalised): in z[j] = bin lane + 64j; out z[u] = time sample 64u + lane
template <int R, bool PK, bool PKC, typename TW>
__device__ __forceinline__ void fft_inverse_t(v2f (&z)[R], const LaneConst<R>& k, const TW& tw, v2f* __restrict__ tr) {
  using G = Geo<R>;
  Dft<R, true, PKC>::run(z);
#pragma unroll
  for (int i = 0; i < R; ++i) tr[k.tr_b + i] = z[i];
  if (G::C == 4 && SPECINV_XLANE_HALF && PK) {
    // (the mirror image of the forward transform's short cut: the row that owns register g + lane / 16 reads its four row values)
#pragma unroll
    for (int g = 0; g < R; g += 4) {
      z[g] = tr[k.tr_q + (0 * R + g) * (R + 1)];
      z[g + 2] = tr[k.tr_q + (1 * R + g) * (R + 1)];
      z[g + 1] = tr[k.tr_q + (2 * R + g) * (R + 1)];
      z[g + 3] = tr[k.tr_q + (3 * R + g) * (R + 1)];
    }
#pragma unroll
    for (int g = 0; g < R; g += 4) {
      z[g + 2] = cmulc_p<PK>(z[g + 2], k.postq[0]);
      z[g + 1] = cmulc_p<PK>(z[g + 1], k.postq[1]);
      z[g + 3] = cmulc_p<PK>(z[g + 3], k.postq[2]);
    }
#pragma unroll
    for (int g = 0; g < R; g += 4) xlane_dft4_out<true>(z[g], z[g + 1], z[g + 2], z[g + 3]);
    cmul_all<R, PK, true, 1>(z, tw);
    Dft<R, true, PKC>::run(z);
    return;
  }
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
  Dft<R, true, PKC>::run(z);
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
  int skew;             // chunk_begin's skew (0: even chunks)
  unsigned long long* stamps;   // SPECINV_TD_STAMPS builds only: [n_waves][8]
  float coef;       // lr (GLA) or rho (ADMM)
  float inv1p;      // 1/(1+rho)
  float fwd_scale;  // 1 or N^-1/2
  float inv_scale;  // 1/N or N^-1/2
  // two-sided spectrograms (k_semi2): state and target of the MIRROR bins N - k, in the lower half's layout - record (k, M - k)
  // holds bins (N - k, M + k), `mid` bin N - M/2; bins 0 and M have no mirror image (their slots are not used)
  v4f* P2_out;
  v2f* Pmid2_out;
  const v4f* m2_pairs;
  const float* m2_mid;
};

// |s| where only the metric reads it (sums compared to 1e-5: the hardware square root is good to 1 ulp)
__device__ __forceinline__ float fast_abs(v2f s) { return __builtin_amdgcn_sqrtf(fmaf(s.x, s.x, s.y * s.y)); }
__device__ __forceinline__ float fast_rcp(float v) { return __builtin_amdgcn_rcpf(v); }
#if SPECINV_IEEE
// ---- the reference's operation order from corrected hardware approximations ------------------------------------------------
// methods.py:246-247 is `spec * target / (spec.abs() + 1e-16)`.  What ATen executes for it (torch 2.10 CPU, checked value by
// value on 2^22 random bins, tools/ref_ops_probe.py): abs = hypotf (correctly rounded), and the complex / real division is a
// multiplication by the rounded RECIPROCAL of the real divisor (c10::complex<T>::operator/= with a zero imaginary part):
//     out = (s * m) * r,   r = RN(1 / (RN(|s|) + 1e-16)).
// Here t = |s|^2 by two fused multiply-adds, y = v_rsq_f32(t) (1 ulp), and
//   SPECINV_REFCHAIN == 2 (shipped): ONE Newton step on y gives r = RN(t^-1/2), the correctly rounded 1 / |s| in all but ~1e-6 of
//     the cases (one rounding of the exact value where the reference rounds |s| and then its reciprocal): two transcendental and
//     four packed instructions per PAIR of bins, no division; 73 % of the outputs bit-identical to the reference's chain, rms
//     deviation from it 4.9e-8 relative - and from the EXACT value 4.7e-8, where the reference's own chain has 5.1e-8;
//   SPECINV_REFCHAIN == 1: h = t y corrected by one residual step to RN(sqrt t), + 1e-16, one Newton step from y to
//     RN(1 / (h + 1e-16)): both of the reference's roundings, 88 % bit-identical (the rest is hypotf rounding the exact sum of
//     squares), three more packed instructions per pair: + 5.3 % on the BASELINE C2 step against + 3.1 % (tools/refchain_study.py,
//     profiles/r04_refchain.txt).
// For comparison: the IEEE sqrt + two divisions of rounds 2-3 (x m / d: not the reference's chain) matched 65 % at + 27 %, the
// approximate path (SPECINV_IEEE=0: s (m rsq(t))) 52 %; every form sits 4.7-5.3e-8 from the exact value.
// kRefFloor keeps rsq finite at s = 0 (then out = 0 like the reference's 0 / 1e-16) and bounds r by 1e16 like the guard does; it
// is absorbed by t above |s| = 6e-13 (REFCHAIN 2 has no other guard: it only differs from the reference's below |s| = 3e-9).
constexpr float kRefFloor = 1e-32f;
__device__ __forceinline__ float ref_norm2(v2f s, float floor_ = kRefFloor) { return fmaf(s.y, s.y, fmaf(s.x, s.x, floor_)); }
__device__ __forceinline__ v2f ref_rcp_abs2(v2f t, float guard = 1e-16f) {
  const v2f y = v2f{__builtin_amdgcn_rsqf(t.x), __builtin_amdgcn_rsqf(t.y)};
  v2f h = t * y;
#if SPECINV_REFCHAIN == 1
  const v2f res = __builtin_elementwise_fma(-h, h, t);
  h = __builtin_elementwise_fma(res, y * 0.5f, h);
  const v2f den = h + guard;
  const v2f e = __builtin_elementwise_fma(-den, y, v2f{1.0f, 1.0f});
  return __builtin_elementwise_fma(e, y, y);
#else
  const v2f e = __builtin_elementwise_fma(-h, y, v2f{1.0f, 1.0f});
  return __builtin_elementwise_fma(y * 0.5f, e, y);
#endif
}
__device__ __forceinline__ float ref_rcp_abs(float t, float guard = 1e-16f) {
  const float y = __builtin_amdgcn_rsqf(t);
  float h = t * y;
#if SPECINV_REFCHAIN == 1
  const float res = fmaf(-h, h, t);
  h = fmaf(res, y * 0.5f, h);
  const float den = h + guard;
  const float e = fmaf(-den, y, 1.0f);
  return fmaf(e, y, y);
#else
  return fmaf(y * 0.5f, fmaf(-h, y, 1.0f), y);
#endif
}
// The envelope division of methods.py:132 (a true float32 division): q = v r, one residual step - correctly rounded when r is the
// correctly rounded reciprocal of e (`env_rcp`: v_rcp_f32 + one Newton step), which the kernels keep in registers wherever the
// envelope is periodic; where it is not (the first frames of an item, tails) they divide.  The envelope table holds the envelope.
__device__ __forceinline__ float env_rcp(float e) {
  const float r = __builtin_amdgcn_rcpf(e);
  return fmaf(fmaf(-e, r, 1.0f), r, r);
}
__device__ __forceinline__ v2f env_rcp(v2f e) { return v2f{env_rcp(e.x), env_rcp(e.y)}; }
__device__ __forceinline__ v2f env_apply_r(v2f v, v2f e, v2f r) {
  const v2f q = v * r;
  return __builtin_elementwise_fma(__builtin_elementwise_fma(-q, e, v), r, q);
}
__device__ __forceinline__ v2f env_apply(v2f v, v2f e) { return v2f{__fdiv_rn(v.x, e.x), __fdiv_rn(v.y, e.y)}; }
__device__ __forceinline__ float env_apply(float v, float e) { return __fdiv_rn(v, e); }
#else
// the envelope table holds 1 / envelope
__device__ __forceinline__ v2f env_rcp(v2f e) { return e; }
__device__ __forceinline__ v2f env_apply_r(v2f v, v2f e, v2f) { return v * e; }
__device__ __forceinline__ v2f env_apply(v2f v, v2f e) { return v * e; }
__device__ __forceinline__ float env_apply(float v, float e) { return v * e; }
#endif

// The projection's factor m / (|s| + 1e-16) (methods.py:246-247) on the default (approximate) path: m * rsq(|s|^2 + 1e-32) -
// one transcendental instruction instead of two (they cost two issue slots each) and no addition.  v_rsq_f32 is good to 1 ulp like
// v_sqrt_f32 and v_rcp_f32 each; the guard term only matters below |s| ~ 1e-9, where both forms tend to m * 1e16, and s = 0 gives
// 0 either way.  The exact-projection build (SPECINV_IEEE) keeps the reference's operations.
#ifndef SPECINV_RSQ
#define SPECINV_RSQ 1
#endif
__device__ __forceinline__ float proj_rsq(v2f s) { return __builtin_amdgcn_rsqf(fmaf(s.y, s.y, fmaf(s.x, s.x, 1e-32f))); }

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
    return ((s * m) * ref_rcp_abs(ref_norm2(s))) * a.inv_scale;
#elif SPECINV_RSQ
    return s * ((m * proj_rsq(s)) * a.inv_scale);
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
    {
#pragma clang fp contract(off)   // X is rounded before Y = X + U is formed (:473-475): no multiply-add across the two
      xn = (xn * m) * ref_rcp_abs(ref_norm2(xn));
    }
#elif SPECINV_RSQ
    {
#pragma clang fp contract(off)   // X is rounded before Y = X + U is formed (:473-475): no multiply-add across the two
      xn = xn * (m * proj_rsq(xn));
    }
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

// Frames [chunk_begin(c), chunk_begin(c+1)) belong to wave-chunk c; sizes differ by at most one frame - unless the plan skews
// them: `skew` frames move from every odd chunk to the even chunk before it (FastState::skew: the waves of a launch that exactly
// fills two wave slots per SIMD do not run at the same speed, kernels_fast_td.h).
// Three waves per SIMD (k_fused4<8>): skew = 0x10000 | s1 << 8 | s2 moves the begin of chunks 3q + 1 / 3q + 2 by s1 / s2 frames.
__device__ __host__ __forceinline__ int chunk_begin(int c, int T, int nchunks, int skew = 0) {
  const int even = (int)(((unsigned)c * (unsigned)T) / (unsigned)nchunks);   // c * T < 2^32 for any plan that fits in memory
  if (skew < 0x10000) return even + ((c & 1) ? skew : 0);
  const int r = c % 3;
  return even + (r == 1 ? ((skew >> 8) & 0xff) : r == 2 ? (skew & 0xff) : 0);
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

// ---- Griffin-Lim with the momentum carried as a signal: the real-FFT split of one conjugate pair (kernels_fast_td.h, k_hop_td)
template <int R>
__device__ __forceinline__ void td_split(v2f zk, v2f zm, v2f wk, float half_scale, v2f& xk, v2f& xm) {
  const v2f e2 = add_conj(zk, zm);
  const v2f tw = cmul_mi(wk, sub_conj(zk, zm));            // W * (-i (Zk - conj Zm))
  xk = (e2 + tw) * half_scale;
  xm = (e2 - tw) * v2f{half_scale, -half_scale};
}
// ... left unscaled (2 / fwd_scale times the STFT bins): for a launch whose projection is the only reader of the bins - it
// divides by their magnitude, so the scale cancels
template <int R>
__device__ __forceinline__ void td_split_raw(v2f zk, v2f zm, v2f wk, v2f& xk, v2f& xm) {
  const v2f e2 = add_conj(zk, zm);
  const v2f tw = cmul_mi(wk, sub_conj(zk, zm));
  xk = e2 + tw;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[1,0]" : "=v"(xm) : "v"(e2), "v"(tw));   // conj(e2 - tw)
}
// W_N^(lane + 64 j) = W_N^lane * W_{2R}^j: the real-FFT twiddle of pair j from the lane's own W_N^lane
template <int R, bool PK = true>
__device__ __forceinline__ v2f pair_twiddle(v2f wn, int j) {
  if (j == 0) return wn;
  return (PK && SPECINV_PKCONST) ? cmul_sk<false>(wn, w64(j * (32 / R))) : cmul_k(wn, w64(j * (32 / R)));
}
// s * p.x and s * p.y: ONE packed multiplication each (the scalar factor is picked from a register pair by op_sel)
__device__ __forceinline__ v2f scale_lo(v2f s, v2f p) {
  v2f d;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(d) : "v"(s), "v"(p));
  return d;
}
__device__ __forceinline__ v2f scale_hi(v2f s, v2f p) {
  v2f d;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(d) : "v"(s), "v"(p));
  return d;
}

// ---- whole frames from the signal (the frame kernels and the one-launch objective)
// windowed frame starting at signal index `start` (may reach into the reflect padding) -> registers
template <int R>
__device__ __forceinline__ void load_frame_raw(const float* __restrict__ xrow, long long len, long long start, int lane,
                                               int pad_mode, v2f (&z)[R]) {
  constexpr int N = Geo<R>::N;
  if (start >= 0 && start + N <= len && (start & 1) == 0) {
    const v2f* src = reinterpret_cast<const v2f*>(xrow + start);
#pragma unroll
    for (int u = 0; u < R; ++u) z[u] = src[64u * u + (unsigned)lane];
  } else if (start >= 0 && start + N <= len) {   // inside the signal at an odd offset: two 4-byte loads per register
    const float* src = xrow + start;
#pragma unroll
    for (int u = 0; u < R; ++u) z[u] = v2f{src[128u * u + 2u * (unsigned)lane], src[128u * u + 2u * (unsigned)lane + 1u]};
  } else {
#pragma unroll
    for (int u = 0; u < R; ++u) {
      const long long n0 = pad_index(start + 128 * u + 2 * lane, len, pad_mode);
      const long long n1 = pad_index(start + 128 * u + 2 * lane + 1, len, pad_mode);
      z[u] = v2f{n0 < 0 ? 0.0f : xrow[n0], n1 < 0 ? 0.0f : xrow[n1]};
    }
  }
}

template <int R>
__device__ __forceinline__ void load_frame_regs(const float* __restrict__ xrow, long long len, long long start, int lane,
                                                int pad_mode, const v2f* __restrict__ lds_win, v2f (&z)[R]) {
  load_frame_raw<R>(xrow, len, start, lane, pad_mode, z);
#pragma unroll
  for (int u = 0; u < R; ++u) z[u] = z[u] * lds_win[64 * u + lane];
}

template <int R>
__device__ __forceinline__ void xform_tables(const float* __restrict__ window, v2f* lds_win, v2f* lds_tw1) {
  constexpr int M = Geo<R>::M;
  for (int i = threadIdx.x; i < M; i += blockDim.x) lds_win[i] = v2f{window[2 * i], window[2 * i + 1]};
  for (int i = threadIdx.x; i < (R - 1) * 64; i += blockDim.x) {
    const int k1 = i / 64 + 1, l = i & 63;
    lds_tw1[i] = unit(2.0f * (float)((l * k1) % M) / (float)M);
  }
  __syncthreads();
}

// ---- argument records of the frame kernels (kernels_frame.h) ----
struct FastXformArgs {
  const float* x;        // (B, len)
  v2f* spec;             // (B*T, F) frame-major, natural bin order
  float* frames;         // (B*T, N)
  const float* window;
  long long len, n_frames_total;
  int T, hop, pad, pad_mode;
  float scale;
};

constexpr int MODE_INIT = 2;
struct SemiArgs {
  FastArgs f;              // x_in, P_in (updated in place), U_in, m_pairs, ..., L, T, pad_mode, coef, scales, partials
  float* frames;           // (B*T, N)
  long long n_frames_total;
  int hop, pad;
  int write_x;             // k_hop_td: 0 = x_{t+1} has no reader (only the seam samples, which the tails kernel needs, are stored)
};

#ifndef SPECINV_HOP_R8_W2     // k_hop at n_fft 1024: 128 registers (2 - 13 spilled), so that two 8-wave workgroups fit a CU like the host's
#define SPECINV_HOP_R8_W2 1     // 4096 wave slots assume: ADMM 1024 / 160 0.191 -> 0.182 ms, 1024 / 300 0.197 -> 0.184 (k_hop_td fits as it is and
#endif                          // measured 5 % slower under the same bound)
struct HopArgs {
  FastArgs f;              // x_in, x_out, P_out (in place), U_out, m_pairs, ..., nchunks, n_waves, L, T, pad_mode, partials
  const float* env;        // (L) reciprocal of the overlap-add envelope
  float* xtail;            // (B, nchunks, n_fft - hop)
  int hop, pad;
  int write_x;             // k_hop_td: 0 = x_{t+1} has no reader (only the seam samples, which the tails kernel needs, are stored)
};

__host__ __device__ inline int hop_chunk_begin(int c, int T, int nchunks) { return (int)((long long)c * T / nchunks); }

struct HopInvArgs {
  const v2f* spec;         // (B*T, F) frame-major, natural bin order
  float* out;              // (B, len)
  float* margins;          // (B, 2, pad)
  float* xtail;            // (B, nchunks, n_fft - hop)
  const float* window;
  long long len;
  int T, nchunks, n_waves, hop, pad;
  float scale;
};

// ---- the heavy kernels: declared here, defined in kernels_fused.h / kernels_fast_td.h / kernels_frame.h and compiled in their own
// translation units (tu_*.hip instantiate them explicitly); the host side (fast_state.h) only takes their addresses
template <int R, int MODE, bool EVAL>
__global__ void k_fused4(FastArgs a);
template <int R, int OV, int MODE, bool EVAL>
__global__ void k_fused(FastArgs a);
template <int R, int OV>
__global__ void k_fused_istft(FastArgs a);
template <int R, bool EARLY, bool EVAL>
__global__ void k_fused4_td(FastArgs a);
template <int R, int OV, bool EARLY, bool EVAL>
__global__ void k_fused_td(FastArgs a);
#ifndef SPECINV_EVAL_PIECES          // k_eval_td at BASELINE C2 (rocprofv3): 1 piece / 2 waves per SIMD 0.133 ms, 2 / 3 0.122, 4 / 4 with LDS
#define SPECINV_EVAL_PIECES 2        // twiddles 0.141 (0.129 at 1 piece) - the fused evaluating variant spends 0.16 on the same work
#endif
#ifndef SPECINV_EVAL_WAVES
#define SPECINV_EVAL_WAVES 3
#endif
constexpr int kEvalPieces = SPECINV_EVAL_PIECES;     // pieces of a chunk per wave of k_eval_td (kernels_fast_td.h: kEvalSub)
template <int R, int OV>
__global__ __launch_bounds__(256, SPECINV_EVAL_WAVES) void k_eval_td(FastArgs a);   // (bounds on the declaration too: see rtisi_fast_args.h)
template <int R>
__global__ void k_fast_stft(FastXformArgs a);
template <int R>
__global__ void k_fast_inverse_frames(FastXformArgs a);
template <int R, int MODE, bool EVAL>
__global__ void k_semi(SemiArgs s);
template <int R, int MODE, bool EVAL>
__global__ void k_semi2(SemiArgs s);     // ... of a two-sided spectrogram
template <int R, int MODE, bool EVAL>
__global__ void k_hop(HopArgs s);
template <int R, int MODE, bool EVAL>
__global__ void k_hop2(HopArgs s);       // ... of a two-sided spectrogram
template <int R, bool EARLY, bool EVAL>
__global__ void k_hop_td(HopArgs s);
template <int R>
__global__ void k_hop_inverse(HopInvArgs a);

}  // namespace SI_FAST_NS (fast, or fast_approx in the approximate-projection units)
}  // namespace specinv
