// The wave-level coverage kernel: one Griffin-Lim / ADMM iteration's frame part (torch_specinv/methods.py:241-248, :464-477) for
// float32 and float64 at power-of-two n_fft 128 ... 2048 - every stft kwarg the reference's tests sweep (test/test_griffin.py:9-32:
// both dtypes, n_fft 128 / 256 / 512, win_length, hop, centring, pad mode, `normalized`, sidedness) - on the coverage path's
// buffers.  Round 6; the generic kernels (kernels_generic.h) keep every other size.
//
// What k_iter_pair / k_iter_pair_dr do with a WORKGROUP per frame pair - stage barriers, one wave per SIMD parked between them,
// 22 - 57 % of their bytes' roofline (profiles/r05_generic.txt) - a lane group of ONE wave does here:
//   * a real frame of n_fft = 2 M samples is M complex points z[n] = x[2n] + i x[2n+1], held RG = M / LG per lane by a group of LG
//     lanes (64 / LG frames per wave: 8 frames at n_fft 128 / 256, 4 at 512, one at 1024 / 2048);
//   * the M-point transform is a Stockham autosort in two or three passes of radix 8 / 16 butterflies in registers; between passes
//     the points cross the group through a WAVE-PRIVATE piece of LDS (in place: a wave's LDS operations execute in order, all of a
//     pass's reads are issued before its first write).  No workgroup barrier after the twiddle table is staged: the waves of a CU
//     drift apart, one streams while another transforms;
//   * the first pass takes its points straight from the signal (windowed on the way, torch.stft's padding resolved per sample at
//     the edges only), the last inverse pass writes the windowed synthesis frame straight to the frames buffer;
//   * between the transforms the group walks the conjugate pairs (k, M - k) of its frame: real-FFT split, the per-bin update of
//     kernels_generic.h (update_core: the same operations as every other coverage kernel - momentum or ADMM step, projection by
//     proj_inv), the inverse split; state and target of a chunk of pairs are requested together before the first is used.  A
//     two-sided spectrogram updates the mirror bins N - k with their own state as k_iter_pair does and takes the Hermitian part.
// The overlap-add stays k_ola / k_ola_f4 / k_ola_d2 (fixed summation order), so x changes only by the transform's rounding.
// Bytes per frame and iteration as for the kernels it replaces: 8 hop + 20 F + 8 N elements (ADMM 36 F).
#pragma once
#include <mutex>

#include "wave_api.h"


#ifndef SPECINV_WAVE_OLA_ALL        // experiments: 1 the register overlap-add for float64 frames of 16 points per lane as well
#define SPECINV_WAVE_OLA_ALL 0
#endif

namespace specinv {
namespace wave {

// lanes per frame and the passes' radices, by log2 M (M = n_fft / 2)
template <int LOGM>
struct GeoF;
// (n_fft 128 / 256: sixteen lanes per frame, FOUR frames per wave - first cut as eight lanes x 8 / 16 points: the wider groups load
// 128-byte pieces instead of 64-byte ones and the shallower registers leave three waves per SIMD their room; 128 / 32 0.137 -> 0.120 ms,
// 256 / 64 0.302 -> 0.242)
template <>
struct GeoF<6> {     // n_fft 128:  4 x 4 x 4
  static constexpr int LG = 16, NPASS = 3, R0 = 4, R1 = 4, R2 = 4, R3 = 1;
};
template <>
struct GeoF<7> {     // n_fft 256:  8 x 4 x 4
  static constexpr int LG = 16, NPASS = 3, R0 = 8, R1 = 4, R2 = 4, R3 = 1;
};
template <>
struct GeoF<8> {     // n_fft 512:  16 x 16
  static constexpr int LG = 16, NPASS = 2, R0 = 16, R1 = 16, R2 = 1, R3 = 1;
};
template <>
struct GeoF<9> {     // n_fft 1024: 8 x 8 x 8
  static constexpr int LG = 64, NPASS = 3, R0 = 8, R1 = 8, R2 = 8, R3 = 1;
};
template <>
struct GeoF<10> {    // n_fft 2048: 16 x 8 x 8
  static constexpr int LG = 64, NPASS = 3, R0 = 16, R1 = 8, R2 = 8, R3 = 1;
};
// The sizes people use that are not 128 2^k - n_fft 400 / 800 / 1000 (25 / 50 / 62.5 ms at 16 kHz; torchaudio's default is 400) - by
// M itself: passes of radix 10 butterflies (2 x 5 in registers) and a short last one; ten points per lane at 400 / 1000 (three
// frames / one frame per wave on 60 / 50 of its lanes), twenty at 800 (three frames per wave).  (First cuts: radix 8 / 5 / 5 passes
// on 16 / 32 lanes without an overlap-add in the kernel - not ahead of the workgroup kernels, tools/log/EXPERIMENTS.md r06-q; two
// passes of radix 10 / 20 with twenty points per lane everywhere - 95 ... 540 registers spilled at 256.)
template <>
struct GeoF<200> {   // n_fft 400:  10 x 10 x 2
  static constexpr int LG = 20, NPASS = 3, R0 = 10, R1 = 10, R2 = 2, R3 = 1;
};
template <>
struct GeoF<400> {   // n_fft 800:  10 x 10 x 2 x 2
  static constexpr int LG = 40, NPASS = 4, R0 = 10, R1 = 10, R2 = 2, R3 = 2;
};
template <>
struct GeoF<500> {   // n_fft 1000: 10 x 10 x 5
  static constexpr int LG = 50, NPASS = 3, R0 = 10, R1 = 10, R2 = 5, R3 = 1;
};
template <typename T, int LOGM>
struct Geo : GeoF<LOGM> {};
// `LOGM` names a size: log2 M for the powers of two, M itself (>= 100) for the others
template <int LOGM>
constexpr int m_of() { return LOGM >= 100 ? LOGM : 1 << LOGM; }
// float64 at n_fft 512: FOUR points per lane on a whole wave, four radix-4 passes (sixteen points per lane on sixteen lanes: the
// partial sums of the register overlap-add were spilled, slower than frames + k_ola; eight on thirty-two: 0.253 ms at 512 / 128;
// four on sixty-four: 0.206 - wider groups load longer pieces and leave the registers to the loads in flight)
// float64 at n_fft 2048: the frame on the 128 lanes of a TWO-WAVE workgroup (a "team": LG = 128), eight points per lane, four passes.
// On one wave it is 16 points per lane - 64 registers before the first butterfly - and the register overlap-add's partial sums had
// to be spilled (measured: slower than frames + k_ola).  What was a wave-private exchange becomes a workgroup barrier of two waves.
template <>
struct Geo<double, 10> {  // 8 x 8 x 4 x 4
  static constexpr int LG = 128, NPASS = 4, R0 = 8, R1 = 8, R2 = 4, R3 = 4;
};
// n_fft 4096 (float32 normally runs the packed frame kernels; this is its coverage form): teams of two waves x 16 points / four
// waves x 8 points
template <>
struct GeoF<11> {         // 16 x 16 x 8
  static constexpr int LG = 128, NPASS = 3, R0 = 16, R1 = 16, R2 = 8, R3 = 1;
};
template <>
struct Geo<double, 11> {  // 8 x 8 x 8 x 4
  static constexpr int LG = 256, NPASS = 4, R0 = 8, R1 = 8, R2 = 8, R3 = 4;
};
// n_fft 8192: teams of four waves x 16 points (float32) / eight waves x 8 points (float64)
template <>
struct GeoF<12> {         // 16 x 16 x 16
  static constexpr int LG = 256, NPASS = 3, R0 = 16, R1 = 16, R2 = 16, R3 = 1;
};
template <>
struct Geo<double, 12> {  // 8 x 8 x 8 x 8
  static constexpr int LG = 512, NPASS = 4, R0 = 8, R1 = 8, R2 = 8, R3 = 8;
};
// n_fft 16384, float32 only (float64 frames of that size take kernels_big.h's rows): eight waves x 16 points; the radix-2 pass second,
// where its table is 16 entries (last: 64 KB of twiddles)
template <>
struct Geo<float, 13> {   // 16 x 2 x 16 x 16
  static constexpr int LG = 512, NPASS = 4, R0 = 16, R1 = 2, R2 = 16, R3 = 16;
};
template <>
struct Geo<double, 8> {   // 4 x 4 x 4 x 4
  static constexpr int LG = 64, NPASS = 4, R0 = 4, R1 = 4, R2 = 4, R3 = 4;
};
constexpr int ilog2(int v) { return v <= 1 ? 0 : 1 + ilog2(v >> 1); }

// LDS position of point p of a frame: one element of slack after every R0 - the first pass writes point j R0 + i from lane j, a
// stride of R0 elements that would put every lane of a store on the same banks
template <int R0>
__device__ __host__ __forceinline__ constexpr int phys(int p) { return p + p / R0; }

template <typename T, bool INV>
__device__ __forceinline__ void bf16(cplx<T> (&a)[16]) {
  constexpr double c1d = 0.92387953251128675613, s1d = 0.38268343236508977173, hd = 0.70710678118654752440;
  const T c1 = (T)c1d, s1 = (T)s1d, h = (T)hd;
  // n = 4 n1 + n0: radix 4 over n1 (slot n0 + 4 k1), twiddle W16^(n0 k1), radix 4 over n0 (slot 4 k1 + k0 = X[k1 + 4 k0])
#pragma unroll
  for (int n0 = 0; n0 < 4; ++n0) bf4<T, INV>(a[n0], a[n0 + 4], a[n0 + 8], a[n0 + 12]);
  auto tw = [&](cplx<T> v, T c, T s) {      // v * (c - i s), conjugated for the inverse
    return INV ? mk<T>(v.x * c - v.y * s, v.y * c + v.x * s) : mk<T>(v.x * c + v.y * s, v.y * c - v.x * s);
  };
  a[5] = tw(a[5], c1, s1);                   // W16^1
  a[6] = tw(a[6], h, h);                     // W16^2
  a[7] = tw(a[7], s1, c1);                   // W16^3
  a[9] = tw(a[9], h, h);                     // W16^2
  a[10] = rot_mi<T, INV>(a[10]);             // W16^4 = -i
  a[11] = tw(a[11], -h, h);                  // W16^6
  a[13] = tw(a[13], s1, c1);                 // W16^3
  a[14] = tw(a[14], -h, h);                  // W16^6
  a[15] = tw(a[15], -c1, -s1);               // W16^9
#pragma unroll
  for (int k1 = 0; k1 < 4; ++k1) bf4<T, INV>(a[4 * k1], a[4 * k1 + 1], a[4 * k1 + 2], a[4 * k1 + 3]);
  cplx<T> o[16];
#pragma unroll
  for (int k0 = 0; k0 < 4; ++k0)
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) o[k1 + 4 * k0] = a[k0 + 4 * k1];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = o[i];
}
template <typename T, bool INV>
__device__ __forceinline__ void bf5(cplx<T>& v0, cplx<T>& v1, cplx<T>& v2, cplx<T>& v3, cplx<T>& v4) {
  const T c1 = T(0.30901699437494742410), c2 = T(-0.80901699437494742410), s1 = T(0.95105651629515357212), s2 = T(0.58778525229247312917);
  const cplx<T> t1 = v1 + v4, t2 = v2 + v3, t3 = v1 - v4, t4 = v2 - v3;
  const cplx<T> m1 = mk<T>(v0.x + c1 * t1.x + c2 * t2.x, v0.y + c1 * t1.y + c2 * t2.y);
  const cplx<T> m2 = mk<T>(v0.x + c2 * t1.x + c1 * t2.x, v0.y + c2 * t1.y + c1 * t2.y);
  const cplx<T> r1 = rot_mi<T, INV>(mk<T>(s1 * t3.x + s2 * t4.x, s1 * t3.y + s2 * t4.y));   // -i (s1 t3 + s2 t4), +i for the inverse
  const cplx<T> r2 = rot_mi<T, INV>(mk<T>(s2 * t3.x - s1 * t4.x, s2 * t3.y - s1 * t4.y));
  v0 = v0 + t1 + t2;
  v1 = m1 + r1;
  v4 = m1 - r1;
  v2 = m2 + r2;
  v3 = m2 - r2;
}
// W_20^e = kW20C[e] - i kW20S[e]
__device__ constexpr double kW20C[20] = {1.0, 0.95105651629515357212, 0.80901699437494742410, 0.58778525229247312917, 0.30901699437494742410, 0.0,
                                         -0.30901699437494742410, -0.58778525229247312917, -0.80901699437494742410, -0.95105651629515357212, -1.0,
                                         -0.95105651629515357212, -0.80901699437494742410, -0.58778525229247312917, -0.30901699437494742410, 0.0,
                                         0.30901699437494742410, 0.58778525229247312917, 0.80901699437494742410, 0.95105651629515357212};
__device__ constexpr double kW20S[20] = {0.0, 0.30901699437494742410, 0.58778525229247312917, 0.80901699437494742410, 0.95105651629515357212, 1.0,
                                         0.95105651629515357212, 0.80901699437494742410, 0.58778525229247312917, 0.30901699437494742410, 0.0,
                                         -0.30901699437494742410, -0.58778525229247312917, -0.80901699437494742410, -0.95105651629515357212, -1.0,
                                         -0.95105651629515357212, -0.80901699437494742410, -0.58778525229247312917, -0.30901699437494742410};
// R = RA RB points in registers: n = RB n1 + n2; DFT_RA over n1 for every n2 (slot RB k1 + n2), twiddle W_R^(n2 k1), DFT_RB over n2
// for every k1, output X[k1 + RA k2]
template <typename T, int RA, int RB, bool INV>
__device__ __forceinline__ void bf_comp(cplx<T> (&a)[RA * RB]) {
  constexpr int R = RA * RB;
  static_assert(20 % R == 0 && (RA == 2 || RA == 4) && RB == 5, "2 x 5 and 4 x 5");
#pragma unroll
  for (int n2 = 0; n2 < RB; ++n2) {
    if constexpr (RA == 2) {
      const cplx<T> u = a[n2], w = a[RB + n2];
      a[n2] = u + w;
      a[RB + n2] = u - w;
    } else {
      bf4<T, INV>(a[n2], a[RB + n2], a[2 * RB + n2], a[3 * RB + n2]);
    }
  }
#pragma unroll
  for (int k1 = 1; k1 < RA; ++k1)
#pragma unroll
    for (int n2 = 1; n2 < RB; ++n2) {
      const int e = (n2 * k1 * (20 / R)) % 20;
      const T c = (T)kW20C[e], sn = (T)kW20S[e];
      const cplx<T> v = a[RB * k1 + n2];
      a[RB * k1 + n2] = INV ? mk<T>(v.x * c - v.y * sn, v.y * c + v.x * sn) : mk<T>(v.x * c + v.y * sn, v.y * c - v.x * sn);
    }
#pragma unroll
  for (int k1 = 0; k1 < RA; ++k1) bf5<T, INV>(a[RB * k1], a[RB * k1 + 1], a[RB * k1 + 2], a[RB * k1 + 3], a[RB * k1 + 4]);
  cplx<T> o[R];
#pragma unroll
  for (int k1 = 0; k1 < RA; ++k1)
#pragma unroll
    for (int k2 = 0; k2 < RB; ++k2) o[k1 + RA * k2] = a[RB * k1 + k2];
#pragma unroll
  for (int i = 0; i < R; ++i) a[i] = o[i];
}
template <typename T, int R, bool INV>
__device__ __forceinline__ void bfly(cplx<T> (&v)[R]) {
  if constexpr (R == 16) bf16<T, INV>(v);
  else if constexpr (R == 5) bf5<T, INV>(v[0], v[1], v[2], v[3], v[4]);
  else if constexpr (R == 10) bf_comp<T, 2, 5, INV>(v);
  else if constexpr (R == 20) bf_comp<T, 4, 5, INV>(v);
  else Butterfly<T, R, INV>::run(v, nullptr, 0);
}

// Twiddles of a pass with NS > 1 from the pass's OWN table in LDS, tab[(q - 1) NS + k] = W_(NS R)^(k q): lanes with consecutive k read
// consecutive entries.  (Round 6, first version: one half-circle table W_N^e for every pass, entry (k q) << shift - a stride of
// q 2^shift elements across the lanes, 16 lanes of a read on the same banks: SQ_LDS_BANK_CONFLICT 53 % of the LDS cycles at float64
// n_fft 2048.)  Conjugated for the inverse.
template <typename T, bool INV>
__device__ __forceinline__ cplx<T> tw_get(const cplx<T>* __restrict__ tab, int idx) {
  const cplx<T> w = tab[idx];
  return mk<T>(w.x, INV ? -w.y : w.y);
}
// entries of the tables of a geometry: pass 1 (NS = R0) and, with three passes, pass 2 (NS = R0 R1)
template <typename T, int LOGM>
struct Tabs {
  using G = Geo<T, LOGM>;
  static constexpr int N1 = G::R0 * (G::R1 - 1);
  static constexpr int N2 = G::NPASS >= 3 ? G::R0 * G::R1 * (G::R2 - 1) : 0;
  static constexpr int N3 = G::NPASS == 4 ? G::R0 * G::R1 * G::R2 * (G::R3 - 1) : 0;
  static constexpr int NPAIR = (m_of<LOGM>() / 2) / G::LG;
  static constexpr int TOTAL = N1 + N2 + N3 + NPAIR;           // + W_N^(i LG), the real-FFT split's step between a lane's pairs
};

// One Stockham pass on the group's LDS buffer, in place: butterfly j (of M / R) reads points j + q M / R, multiplies by W_m^(k q)
// (m = NS R, j = blk NS + k), transforms, writes points blk NS R + k + i NS.  A lane owns butterflies gl, gl + LG, ...
// what orders one lane's LDS write before another lane's read of it: the hardware executes a wave's LDS operations in order; the
// compiler is told by a wavefront-scope fence (no instruction)
// A frame on a two-wave team (LG = 128) synchronises with the workgroup's barrier instead - the workgroup IS the team, so every
// thread reaches every barrier (the unit loop's bounds are the same for both waves).
template <int LG>
__device__ __forceinline__ void wave_sync() {
  if constexpr (LG > 64) __syncthreads();
  else __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

template <typename T, int R, int NS, bool INV, int LOGM, int LG, int PS>
__device__ __forceinline__ void pass_lds(cplx<T>* buf, const cplx<T>* __restrict__ tab, int gl) {
  constexpr int M = m_of<LOGM>(), NB = M / R, PER = NB / LG;
  static_assert(PER >= 1 && NB % LG == 0, "a lane owns whole butterflies of every pass");
  cplx<T> v[PER][R];
#pragma unroll
  for (int it = 0; it < PER; ++it) {
    const int j = gl + it * LG;
#pragma unroll
    for (int q = 0; q < R; ++q) v[it][q] = buf[phys<PS>(j + q * NB)];
  }
#pragma unroll
  for (int it = 0; it < PER; ++it) {
    const int j = gl + it * LG, k = j % NS;
    if (NS > 1) {
#pragma unroll
      for (int q = 1; q < R; ++q) v[it][q] = cmul(v[it][q], tw_get<T, INV>(tab, (q - 1) * NS + k));
    }
    bfly<T, R, INV>(v[it]);
  }
  wave_sync<LG>();
#pragma unroll
  for (int it = 0; it < PER; ++it) {
    const int j = gl + it * LG, blk = j / NS, k = j % NS;
    const int base = blk * NS * R + k;
#pragma unroll
    for (int i = 0; i < R; ++i) buf[phys<PS>(base + i * NS)] = v[it][i];
  }
  wave_sync<LG>();
}

// registers: a float64 frame of 16 points per lane is 64 registers before the first butterfly - those instantiations may take 256
// (two waves per SIMD, which is also what their 17 KB of LDS per wave allow); everything else is held to 128 (four per SIMD)
template <typename T, int LOGM>
constexpr int max_threads() { return (sizeof(T) == 8 && m_of<LOGM>() / Geo<T, LOGM>::LG >= 16) ? 512 : 1024; }
// ... and with the overlap-add in registers (partial sums, the frame's outputs, envelope and signal addresses on top): three waves
// per SIMD, 168 registers (at 128 those instantiations spilled 33 - 187 registers and ran slower than frames + k_ola)
// - and a two-sided frame's four bins per conjugate pair: two (float64) or three waves
// ... and with the ring (measured at 4 / 3 / 2 waves per SIMD, tools/log/EXPERIMENTS.md r06-u: float64 512 / 300 / 100 0.499 / 0.397 /
// 0.398 ms, 1024 / 800 / 200 0.513 / 0.351 / 0.247, 256 / 200 / 50 0.754 / 0.529 / 0.471 - at 128 registers those spill 31 ... 280;
// float32 1024 / 800 / 200 0.295 / 0.260 / 0.286, 256 / 200 / 50 0.304 / 0.272 / 0.301)
#ifndef SPECINV_WAVE_RING_WPS64
#define SPECINV_WAVE_RING_WPS64 2
#endif
#ifndef SPECINV_WAVE_RING_WPS32
#define SPECINV_WAVE_RING_WPS32 3
#endif
// ... at n_fft 400 / 800 / 1000 (ten points per lane; 3 / 2 waves per SIMD: 1000 / 250 0.111 / 0.084 ms, ADMM 400 / 100 0.430 / 0.313,
// 400 / 160 0.277 / 0.272, 800 / 200 0.329 / 0.336)
#ifndef SPECINV_WAVE_RING_WPS32S
#define SPECINV_WAVE_RING_WPS32S 2
#endif
template <typename T, int LOGM, int OV, bool TWO>
constexpr int waves_per_simd() {
  if (Geo<T, LOGM>::LG > 64) return 2;            // (a team's workgroups: four to a CU by their LDS)
  // (the ring: a wave's frame buffers and rings are 16 KB of LDS where a lane carries 16 points - two waves per SIMD fit anyway)
  if (OV == 1 && sizeof(T) == 4 && m_of<LOGM>() / Geo<T, LOGM>::LG >= 16) return 2;
  if (TWO) return sizeof(T) == 8 || LOGM >= 100 ? 2 : 3;
  if (OV == 1) return sizeof(T) == 8 ? SPECINV_WAVE_RING_WPS64 : (LOGM >= 100 ? SPECINV_WAVE_RING_WPS32S : SPECINV_WAVE_RING_WPS32);
  return max_threads<T, LOGM>() == 512 ? 2 : (OV > 1 ? ((m_of<LOGM>() / Geo<T, LOGM>::LG >= 16 || sizeof(T) == 8) ? 2 : 3) : 4);
}

template <typename T, int LOGM, int MODE, bool TWO, bool EVAL, int OV>
__global__ __attribute__((amdgpu_flat_work_group_size(64, 512), amdgpu_waves_per_eu((waves_per_simd<T, LOGM, OV, TWO>()))))
void k_wave_iter(WaveIterArgs<T> a) {
  using G = Geo<T, LOGM>;
  using C = cplx<T>;
  constexpr int M = m_of<LOGM>(), N = 2 * M, LG = G::LG;
  constexpr int TEAM = LG > 64 ? LG / 64 : 1;            // waves per frame (a team is a whole workgroup)
  constexpr int FPW = LG > 64 ? 1 : 64 / LG;             // frames per wave (per workgroup for a team); lanes >= FPW LG idle
  constexpr int PS = G::R0;
  constexpr int MP = phys<PS>(M) + 1;                    // a frame's points in LDS
  constexpr bool POW2 = (N & (N - 1)) == 0;
  constexpr int NS1 = G::R0, NS2 = G::R0 * G::R1, NS3 = G::R0 * G::R1 * G::R2;   // block lengths before passes 1, 2, 3
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using TB = Tabs<T, LOGM>;
  C* tab1 = reinterpret_cast<C*>(smem);                  // pass tables, then W_N^(i LG)
  C* tab2 = tab1 + TB::N1;
  C* tab3 = tab2 + TB::N2;
  C* tabs = tab3 + TB::N3;
  // (the wave's index as a scalar: everything derived from it - the group of frames, the bases of their state, target and
  // synthesis rows - then lives in scalar registers and the loads take the base + 32-bit offset form; from threadIdx.x alone the
  // compiler would carry a 64-bit address per access in vector registers)
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const bool lane_on = TEAM > 1 || lane < FPW * LG;
  const int g = TEAM > 1 ? 0 : (lane_on ? lane / LG : 0), gl = TEAM > 1 ? (int)threadIdx.x : lane % LG;
  C* buf = tab1 + TB::TOTAL + (size_t)(TEAM > 1 ? 0 : wave * FPW + g) * MP;
  {
    constexpr int ST1 = N / (NS1 * G::R1);                             // W_(NS R)^(k q) = W_N^((k q) N / (NS R))
    for (int i = threadIdx.x; i < TB::N1; i += blockDim.x) tab1[i] = a.c.tw[((i % NS1) * (i / NS1 + 1)) * ST1];
    if constexpr (G::NPASS >= 3) {
      constexpr int ST2 = N / (NS2 * G::R2);
      for (int i = threadIdx.x; i < TB::N2; i += blockDim.x) tab2[i] = a.c.tw[((i % NS2) * (i / NS2 + 1)) * ST2];
    }
    if constexpr (G::NPASS == 4) {
      constexpr int ST3 = N / (NS3 * G::R3);
      for (int i = threadIdx.x; i < TB::N3; i += blockDim.x) tab3[i] = a.c.tw[((i % NS3) * (i / NS3 + 1)) * ST3];
    }
    for (int i = threadIdx.x; i < TB::NPAIR; i += blockDim.x) tabs[i] = a.c.tw[i * LG];
  }
  const C wlane = a.c.tw[gl];                            // W_N^gl: pair k = gl + i LG takes W_N^k = wlane W_N^(i LG)
  __syncthreads();
  const FrameCfg<T>& c = a.c;
  const int Tn = c.n_frames, F = c.n_freq;
  const int64_t total = (int64_t)a.batch * Tn;
  // (the stride of the unit loop: waves of the launch, or teams)
  const int64_t w0 = TEAM > 1 ? (int64_t)blockIdx.x : (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
  const int64_t nw = TEAM > 1 ? (int64_t)gridDim.x : (int64_t)gridDim.x * (blockDim.x >> 6);
  const T hs = T(0.5) * c.fwd_scale;
  const T coef = a.coef, inv1p = a.inv1p;
  constexpr bool eval = EVAL, two = TWO;
  const T* __restrict__ win = c.window;
  double s_d = 0, s_o = 0;
  // Work units.  OV == 0: a frame, its windowed synthesis frame written to the frames buffer for k_ola.  OV in {2, 4, 8} (hop =
  // n_fft / OV): a CHUNK of consecutive frames [cb(c), cb(c + 1)) of one item, walked in order by one lane group with the
  // overlap-add in registers - OV - 1 hop-blocks of partial sums carried from frame to frame, one finished block divided by the
  // envelope and stored to the OTHER signal buffer per frame, added in ascending frame order like k_ola.  The first OV - 1 blocks
  // of a chunk lack the previous chunk's frames and its last OV - 1 partial sums lack the next one's: both go to side buffers and
  // k_wave_seams finishes those blocks.
  // OV == 1 (any other hop < n_fft, two-sided spectrograms): the same chunk walk with the overlap-add in an LDS RING of n_fft samples
  // per lane group - a frame's samples are added at (t hop + s) mod n_fft, the first hop of them are complete, leave for the signal
  // (or, at the chunk's start, for k_wave_seams) and are zeroed; consecutive frames of a group follow each other in program order
  // (a team: behind the frame's closing barrier), so every ring entry is summed in ascending frame order like k_ola's sums.
  static_assert(OV == 0 || OV == 1 || OV == 2 || OV == 4 || OV == 8, "hop = n_fft / 2, / 4 or / 8, or the ring");
  constexpr int RL = G::NPASS == 4 ? G::R3 : G::NPASS == 3 ? G::R2 : G::R1, PERL = (M / RL) / LG;       // the last pass: radix, butterflies per lane
  constexpr int RPB = OV > 1 ? RL / (OV > 1 ? OV : 1) : 1, NBLK = OV > 1 ? PERL * RPB : 1, NACC = OV > 1 ? OV - 1 : 1;   // complex values per lane and hop-block
  static_assert(OV <= 1 || RL % OV == 0, "a hop-block is whole outputs of the last pass");
  T* ring = nullptr;                                     // OV == 1: n_fft samples per lane group, behind the frame buffers
  if constexpr (OV == 1) {
    const int nbuf = TEAM > 1 ? 1 : (int)(blockDim.x >> 6) * FPW;
    ring = reinterpret_cast<T*>(tab1 + TB::TOTAL + (size_t)nbuf * MP) + (size_t)(TEAM > 1 ? 0 : wave * FPW + g) * N;
  }
  const int nch = OV > 0 ? a.nch : 1;
  const int64_t units = OV > 0 ? (int64_t)a.batch * nch : total;
  const int trips = OV > 0 ? (Tn + nch - 1) / nch : 1;
  auto cb = [&](int cc) { return (int)(((int64_t)cc * Tn) / nch); };
  for (int64_t ur = w0; ur * FPW < units; ur += nw) {
    const int64_t u = ur * FPW + g;                 // this lane group's unit
    int fstart = 0, len = 0, ub = 0, ta = 0;
    if (u < units && lane_on) {
      if (OV > 0) {
        ub = (int)(u / nch);
        const int cc = (int)(u - (int64_t)ub * nch);
        ta = cb(cc);
        len = cb(cc + 1) - ta;
        fstart = ub * Tn + ta;
      } else {
        fstart = (int)u;
        len = 1;
      }
    }
    C acc[NACC][NBLK];
#pragma unroll
    for (int b = 0; b < NACC; ++b)
#pragma unroll
      for (int e = 0; e < NBLK; ++e) acc[b][e] = mk<T>(T(0), T(0));
    if constexpr (OV == 1) {                        // an empty ring (every lane its own sample slots)
#pragma unroll
      for (int m = 0; m < M / LG; ++m) *reinterpret_cast<C*>(ring + 2 * (gl + m * LG)) = mk<T>(T(0), T(0));
      wave_sync<LG>();
    }
   for (int s = 0; s < trips; ++s) {
    if (s >= len) continue;
    const int fi = fstart + s;                      // (host: fewer than 2^31 frames)
    const int bi = OV > 0 ? ub : (int)((unsigned)fi / (unsigned)Tn);
    const int t = OV > 0 ? ta + s : fi - bi * Tn;
    // state, target and synthesis rows: a scalar base (the wave's first active group) + this group's 32-bit offset
    const int fi0 = __builtin_amdgcn_readfirstlane(fi);
    C* S0u = a.S0 + (int64_t)fi0 * F;
    C* S1u = MODE == 1 ? a.S1 + (int64_t)fi0 * F : nullptr;
    const T* magu = a.mag + (int64_t)fi0 * F;
    const int so = (fi - fi0) * F;
    // ---- analysis, first pass: butterfly j takes points j + q M / R0 from the signal
    {
      constexpr int R = G::R0, NB = M / R, PER = NB / LG;
      static_assert(PER >= 1, "a lane owns at least one butterfly of the first pass");
      const int64_t start = (int64_t)t * c.hop - c.pad;
      const T* xr = a.x + (int64_t)bi * c.length;
      const T* xp = xr + start;
      const bool interior = start >= 0 && start + N <= c.length;
      const bool aligned = (((int64_t)bi * c.length + start) & 1) == 0;
      C v[PER][R];
      if (interior && aligned) {
#pragma unroll
        for (int it = 0; it < PER; ++it)
#pragma unroll
          for (int q = 0; q < R; ++q) v[it][q] = *reinterpret_cast<const C*>(xp + 2 * (gl + it * LG + q * NB));
      } else if (interior) {
#pragma unroll
        for (int it = 0; it < PER; ++it)
#pragma unroll
          for (int q = 0; q < R; ++q) {
            const int p = gl + it * LG + q * NB;
            v[it][q] = mk<T>(xp[2 * p], xp[2 * p + 1]);
          }
      } else {
        // a frame that reaches into the padding (the first and last n_fft / (2 hop) frames of an item): torch.stft's pad modes per
        // sample - index arithmetic with 64-bit remainders - in a ROLLED loop that parks the lane's points in its LDS buffer (idle
        // until the first pass writes it); unrolled sixteen-fold in front of every frame it cost 4 000 instructions and the
        // registers of the common path
#pragma unroll 1
        for (int m = 0; m < M / LG; ++m) {
          const int p = gl + m * LG;
          buf[phys<PS>(p)] = mk<T>(load_padded(xr, c.length, start + 2 * p, c.pad_mode), load_padded(xr, c.length, start + 2 * p + 1, c.pad_mode));
        }
#pragma unroll
        for (int it = 0; it < PER; ++it)
#pragma unroll
          for (int q = 0; q < R; ++q) v[it][q] = buf[phys<PS>(gl + it * LG + q * NB)];
        // (a team: the other wave may still be reading its parked points when this one starts writing the pass's outputs over
        // them - `interior` is the frame's, the same for every thread of the team)
        wave_sync<LG>();
      }
#pragma unroll
      for (int it = 0; it < PER; ++it) {
#pragma unroll
        for (int q = 0; q < R; ++q) {
          const C w2 = *reinterpret_cast<const C*>(win + 2 * (gl + it * LG + q * NB));
          v[it][q] = mk<T>(v[it][q].x * w2.x, v[it][q].y * w2.y);
        }
        bfly<T, R, false>(v[it]);
        const int j = gl + it * LG;
#pragma unroll
        for (int i = 0; i < R; ++i) buf[phys<PS>(j * R + i)] = v[it][i];
      }
      wave_sync<LG>();
    }
    pass_lds<T, G::R1, NS1, false, LOGM, LG, PS>(buf, tab1, gl);
    if constexpr (G::NPASS >= 3) pass_lds<T, G::R2, NS2, false, LOGM, LG, PS>(buf, tab2, gl);
    if constexpr (G::NPASS == 4) pass_lds<T, G::R3, NS3, false, LOGM, LG, PS>(buf, tab3, gl);
    // ---- the conjugate pairs (k, M - k): split, update, inverse split
    auto upd = [&](C r, int f) -> C {               // one bin of this frame
      C n0, n1;
      const C zero = mk<T>(T(0), T(0));
      const C y = update_core<T, MODE>(r, magu[so + f], S0u[so + f], MODE == 1 ? S1u[so + f] : zero, coef, inv1p, eval, s_d, s_o, n0, n1);
      S0u[so + f] = n0;
      if (MODE == 1) S1u[so + f] = n1;
      return y;
    };
    auto herm = [&](C r, int f) -> C {              // two-sided: bins f and N - f, each with its own state; the Hermitian part
      const C y1 = upd(r, f), y2 = upd(conj(r), N - f);
      return mk<T>(T(0.5) * (y1.x + y2.x), T(0.5) * (y1.y - y2.y));
    };
    constexpr int NPAIR = (M / 2) / LG;             // k = gl + i LG < M / 2
    constexpr int CH = sizeof(T) == 8 ? (NPAIR % 2 == 0 ? 2 : 1) : (NPAIR % 4 == 0 ? 4 : NPAIR % 5 == 0 ? 5 : NPAIR % 2 == 0 ? 2 : 1);   // pairs whose state is requested together
    static_assert(NPAIR % CH == 0 && (M / 2) % LG == 0, "whole chunks of pairs");
#pragma unroll 1
    for (int i0 = 0; i0 < NPAIR; i0 += CH) {
      C za[CH], zb[CH], s0a[CH], s0b[CH], s1a[CH], s1b[CH];
      T ma[CH], mb[CH];
      // a two-sided spectrogram (methods.py:142-146): the mirror bins N - k and N - (M - k) = M + k carry their own state and target
      constexpr int CT = TWO ? CH : 1;
      C s0c[CT], s0d[CT], s1c[CT], s1d[CT];
      T mc[CT], md[CT];
      // requests first
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        const int k = gl + (i0 + u) * LG;
        const int kb = k == 0 ? M : M - k;          // k = 0 pairs the real bins 0 and M
        s0a[u] = S0u[so + k];
        s0b[u] = S0u[so + kb];
        ma[u] = magu[so + k];
        mb[u] = magu[so + kb];
        if (MODE == 1) {
          s1a[u] = S1u[so + k];
          s1b[u] = S1u[so + kb];
        }
        if constexpr (TWO) {
          const int kc = k == 0 ? 0 : N - k, kd = k == 0 ? M : M + k;   // (bins 0 and M are their own mirror images: loaded, not used)
          s0c[u] = S0u[so + kc];
          s0d[u] = S0u[so + kd];
          mc[u] = magu[so + kc];
          md[u] = magu[so + kd];
          if (MODE == 1) {
            s1c[u] = S1u[so + kc];
            s1d[u] = S1u[so + kd];
          }
        }
        za[u] = buf[phys<PS>(k)];
        zb[u] = buf[phys<PS>(k == 0 ? 0 : M - k)];
      }
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        const int k = gl + (i0 + u) * LG;
        const int kb = k == 0 ? M : M - k;
        C xk, xm;                                   // the frame's bins k and kb
        const C w = (i0 + u) == 0 ? wlane : cmul(wlane, tabs[i0 + u]);   // W_N^k
        if (k == 0) {
          xk = mk<T>((za[u].x + za[u].y) * c.fwd_scale, T(0));
          xm = mk<T>((za[u].x - za[u].y) * c.fwd_scale, T(0));
        } else {
          const C bc = conj(zb[u]);
          const C e = mk<T>((za[u].x + bc.x) * hs, (za[u].y + bc.y) * hs);
          const C d = mk<T>((za[u].x - bc.x) * hs, (za[u].y - bc.y) * hs);
          const C wo = cmul(w, mk<T>(d.y, -d.x));   // W^k (-i d)
          xk = e + wo;
          xm = conj(e - wo);
        }
        C yk, ym;
        {
          C n0, n1;
          const C zero = mk<T>(T(0), T(0));
          yk = update_core<T, MODE>(xk, ma[u], s0a[u], MODE == 1 ? s1a[u] : zero, coef, inv1p, eval, s_d, s_o, n0, n1);
          S0u[so + k] = n0;
          if (MODE == 1) S1u[so + k] = n1;
          ym = update_core<T, MODE>(xm, mb[u], s0b[u], MODE == 1 ? s1b[u] : zero, coef, inv1p, eval, s_d, s_o, n0, n1);
          S0u[so + kb] = n0;
          if (MODE == 1) S1u[so + kb] = n1;
          if constexpr (TWO) {
            if (k != 0) {                           // the mirror bins see the conjugate spectrum; what ifft(.).real keeps is the Hermitian part
              const C y2 = update_core<T, MODE>(conj(xk), mc[u], s0c[u], MODE == 1 ? s1c[u] : zero, coef, inv1p, eval, s_d, s_o, n0, n1);
              S0u[so + N - k] = n0;
              if (MODE == 1) S1u[so + N - k] = n1;
              yk = mk<T>(T(0.5) * (yk.x + y2.x), T(0.5) * (yk.y - y2.y));
              const C y3 = update_core<T, MODE>(conj(xm), md[u], s0d[u], MODE == 1 ? s1d[u] : zero, coef, inv1p, eval, s_d, s_o, n0, n1);
              S0u[so + M + k] = n0;
              if (MODE == 1) S1u[so + M + k] = n1;
              ym = mk<T>(T(0.5) * (ym.x + y3.x), T(0.5) * (ym.y - y3.y));
            }
          }
        }
        if (k == 0) {                               // irfft / ifft(.).real: the imaginary parts of bins 0 and M do not count
          buf[phys<PS>(0)] = mk<T>(yk.x + ym.x, yk.x - ym.x);
        } else {
          const C p = mk<T>(yk.x + ym.x, yk.y - ym.y);                   // Y_k + conj Y_{M-k}
          const C q = cmul(mk<T>(yk.x - ym.x, yk.y + ym.y), conj(w));    // (Y_k - conj Y_{M-k}) conj W^k
          buf[phys<PS>(k)] = mk<T>(p.x - q.y, p.y + q.x);                // p + i q
          buf[phys<PS>(M - k)] = mk<T>(p.x + q.y, q.x - p.y);            // conj p + i conj q
        }
      }
    }
    if (gl == 0) {                                  // the one bin that is its own partner: M / 2
      const C z = buf[phys<PS>(M / 2)];
      const C xk = mk<T>(z.x * c.fwd_scale, -z.y * c.fwd_scale);
      const C y = two ? herm(xk, M / 2) : upd(xk, M / 2);
      buf[phys<PS>(M / 2)] = mk<T>(T(2) * y.x, T(-2) * y.y);
    }
    // ---- synthesis: the passes again with conjugated twiddles, the last one straight to the frames buffer
    wave_sync<LG>();
    pass_lds<T, G::R0, 1, true, LOGM, LG, PS>(buf, tab1, gl);
    if constexpr (G::NPASS >= 3) pass_lds<T, G::R1, NS1, true, LOGM, LG, PS>(buf, tab1, gl);
    if constexpr (G::NPASS == 4) pass_lds<T, G::R2, NS2, true, LOGM, LG, PS>(buf, tab2, gl);
    {
      constexpr int R = RL, NS = M / R, NB = M / R, PER = NB / LG;
      static_assert(NB == NS, "the last pass has one block");
      const C* tabl = G::NPASS == 4 ? tab3 : G::NPASS == 3 ? tab2 : tab1;
      C v[PER][R];
#pragma unroll
      for (int it = 0; it < PER; ++it) {
        const int j = gl + it * LG;
#pragma unroll
        for (int q = 0; q < R; ++q) v[it][q] = buf[phys<PS>(j + q * NB)];
      }
#pragma unroll
      for (int it = 0; it < PER; ++it) {
        const int j = gl + it * LG;                 // k = j, blk = 0
#pragma unroll
        for (int q = 1; q < R; ++q) v[it][q] = cmul(v[it][q], tw_get<T, true>(tabl, (q - 1) * NS + j));
        bfly<T, R, true>(v[it]);
      }
      {
      // (no contraction in this block: the frame's windowed samples are ROUNDED products - what the frames buffer would hold - before
      // they are added, or the register overlap-add would differ from k_ola by a fused multiply-add's rounding in a quarter of the
      // samples)
#pragma clang fp contract(off)
      C y[PER][R];                                  // the windowed synthesis frame: samples 2 p, 2 p + 1 at p = j + i NS
#pragma unroll
      for (int it = 0; it < PER; ++it) {
#pragma unroll
        for (int i = 0; i < R; ++i) {
          const C w2 = *reinterpret_cast<const C*>(win + 2 * (gl + it * LG + i * NS));
          y[it][i] = mk<T>((v[it][i].x * c.inv_scale) * w2.x, (v[it][i].y * c.inv_scale) * w2.y);
        }
      }
      if constexpr (OV == 0) {
        T* fru = a.frames + (int64_t)fi0 * N;
        const int fo = (fi - fi0) * N;
#pragma unroll
        for (int it = 0; it < PER; ++it)
#pragma unroll
          for (int i = 0; i < R; ++i) *reinterpret_cast<C*>(fru + fo + 2 * (gl + it * LG + i * NS)) = y[it][i];
      } else if constexpr (OV == 1) {
        const int hop = c.hop, keep = N - hop;
        const int64_t a0 = (int64_t)t * hop;        // padded position of the frame's first sample (the chunk's: ta hop)
        const int rb = POW2 ? (int)(a0 & (N - 1)) : (int)(a0 % N);
        const int64_t n0 = a0 - c.pad;              // ... its place in the signal; samples [lo, hi) of the frame are inside
        const int lo = (int)(n0 < 0 ? (-n0 < N ? -n0 : N) : 0), hi = (int)(c.length - n0 < N ? (c.length - n0 > 0 ? c.length - n0 : 0) : N);
        const int head = keep - s * hop;            // the previous chunk's last frames reach samples [0, head): k_wave_seams' part
        T* xf = a.x_out + ((int64_t)bi * c.length + n0);
        const T* ef = a.env + n0;
        T* sR = a.seamR + ((int64_t)u * keep + (int64_t)s * hop);
        T* sL = a.seamL + ((int64_t)u * keep - hop);
        const bool last = s == len - 1, even = !(hop & 1);   // (an even hop: sample pairs stay together in the ring)
        auto emit = [&](int sidx, T v) -> T {       // the new ring entry
          if (sidx < hop) {                         // complete as far as this chunk's frames go
            if (sidx < head) sR[sidx] = v;
            else if (sidx >= lo && sidx < hi) xf[sidx] = v / ef[sidx];
            return T(0);
          }
          if (last) sL[sidx] = v;                   // what the chunk leaves for the samples after its last frame's first hop
          return v;
        };
        auto put = [&](int p, C yv) {               // samples 2 p, 2 p + 1 of the frame into the ring
          const int s0 = 2 * p;
          int i0 = rb + s0, i1 = rb + s0 + 1;
          if constexpr (POW2) {
            i0 &= N - 1;
            i1 &= N - 1;
          } else {
            i0 -= i0 >= N ? N : 0;
            i1 -= i1 >= N ? N : 0;
          }
          if (even) {
            const C r = *reinterpret_cast<const C*>(ring + i0);
            *reinterpret_cast<C*>(ring + i0) = mk<T>(emit(s0, r.x + yv.x), emit(s0 + 1, r.y + yv.y));
          } else {
            ring[i0] = emit(s0, ring[i0] + yv.x);
            ring[i1] = emit(s0 + 1, ring[i1] + yv.y);
          }
        };
        // (taking the frame to the ring through its LDS buffer in a rolled loop - fewer live registers - was no faster at 2, 3 or 4
        // waves per SIMD: tools/log/EXPERIMENTS.md r06-w)
#pragma unroll
        for (int it = 0; it < PER; ++it)
#pragma unroll
          for (int i = 0; i < R; ++i) put(gl + it * LG + i * NS, y[it][i]);
      } else {
        // hop-block b of the frame is outputs i in [b RPB, (b + 1) RPB): slot e = it RPB + i mod RPB sits at complex offset
        // gl + it LG + (i mod RPB) NS of the block.  Block 0 closes the oldest partial sum (padded position t hop): stored - or,
        // for the first OV - 1 frames of the chunk, parked for k_wave_seams - then the sums move up by one block.
        const bool complete = s >= OV - 1;
        const int64_t seam = ((int64_t)u * (OV - 1)) * c.hop;
        T* xo = a.x_out + (int64_t)bi * c.length;
#pragma unroll
        for (int it = 0; it < PER; ++it)
#pragma unroll
          for (int ii = 0; ii < RPB; ++ii) {
            const int e = it * RPB + ii, o = gl + it * LG + ii * NS;
            const C val = acc[0][e] + y[it][ii];
            if (complete) {
              const int64_t n = (int64_t)t * c.hop + 2 * o - c.pad;
              if (n >= 0 && n < c.length) {
                const C ev = *reinterpret_cast<const C*>(a.env + n);
                *reinterpret_cast<C*>(xo + n) = mk<T>(val.x / ev.x, val.y / ev.y);
              }
            } else {
              *reinterpret_cast<C*>(a.seamR + seam + (int64_t)s * c.hop + 2 * o) = val;
            }
          }
#pragma unroll
        for (int b = 0; b + 1 < OV - 1; ++b)
#pragma unroll
          for (int it = 0; it < PER; ++it)
#pragma unroll
            for (int ii = 0; ii < RPB; ++ii) acc[b][it * RPB + ii] = acc[b + 1][it * RPB + ii] + y[it][(b + 1) * RPB + ii];
#pragma unroll
        for (int it = 0; it < PER; ++it)
#pragma unroll
          for (int ii = 0; ii < RPB; ++ii) acc[OV - 2][it * RPB + ii] = y[it][(OV - 1) * RPB + ii];
        if (s == len - 1) {                         // what the chunk leaves for the blocks after its last frame
#pragma unroll
          for (int b = 0; b < OV - 1; ++b)
#pragma unroll
            for (int it = 0; it < PER; ++it)
#pragma unroll
              for (int ii = 0; ii < RPB; ++ii)
                *reinterpret_cast<C*>(a.seamL + seam + (int64_t)b * c.hop + 2 * (gl + it * LG + ii * NS)) = acc[b][it * RPB + ii];
        }
      }
      }
      wave_sync<LG>();
    }
   }
  }
  if (eval) {
    const double d = wave_sum(s_d), o = wave_sum(s_o);
    if (lane == 0) {
      const int64_t wi = TEAM > 1 ? w0 * TEAM + wave : w0;     // a pair of sums per WAVE of the launch
      a.partials[2 * wi] = d;
      a.partials[2 * wi + 1] = o;
    }
  }
}

// The hop-blocks at chunk boundaries of the register overlap-add: boundary c of item b (c = 0 ... nch; frame position cb(c), T for
// the last) has OV - 1 blocks whose sums are split between the chunk before it (seamL: what its last frames left) and the chunk
// after it (seamR: what its first frames had so far); x = (left + right) / envelope.
// (`keep` samples per boundary: (OV - 1) hop of the register form, n_fft - hop of the ring)
template <typename T>
__global__ void k_wave_seams(const T* __restrict__ seamL, const T* __restrict__ seamR, const T* __restrict__ env, T* __restrict__ x_out,
                             int Tn, int nch, int keep, int hop, int pad, int64_t length, int batch) {
  const int64_t per = keep, total = (int64_t)batch * (nch + 1) * per;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t cu = idx / per;
    const int r = (int)(idx - cu * per);
    const int b = (int)(cu / (nch + 1)), cc = (int)(cu - (int64_t)b * (nch + 1));
    const int pos = cc < nch ? (int)(((int64_t)cc * Tn) / nch) : Tn;
    const int64_t n = (int64_t)pos * hop + r - pad;
    if (n < 0 || n >= length) continue;
    const T right = cc < nch ? seamR[(int64_t)(b * (int64_t)nch + cc) * keep + r] : T(0);
    const T left = cc > 0 ? seamL[(int64_t)(b * (int64_t)nch + cc - 1) * keep + r] : T(0);
    x_out[(int64_t)b * length + n] = (left + right) / env[n];
  }
}

// ---- host side ---------------------------------------------------------------------------------------------------------------
struct Launch {
  int wgs = 0, waves_per_wg = 0, capacity = 0;      // capacity: lane groups (frames in flight) the chip holds at once
  size_t lds = 0;
};
// can the partial sums of the register overlap-add stay in registers?  (a float64 frame of 16 points per lane already fills them:
// its OV - 1 blocks would be spilled - the traffic of the frames buffer in another place)
// - and a hop-block must be whole outputs of the last pass (its radix a multiple of OV)
template <typename T, int LOGM, int OV>
constexpr bool ola_fits() {
  using G = Geo<T, LOGM>;
  // (the sizes that are not powers of two: the ring only)
  // (... and float32 n_fft 16384 at hop = n_fft / 8: seven blocks of sums beside sixteen points - 0.394 ms against 0.270 on frames + k_ola)
  if (LOGM == 13 && OV == 8) return false;
  return OV <= 1 || (LOGM < 100 && (SPECINV_WAVE_OLA_ALL || !(sizeof(T) == 8 && m_of<LOGM>() / G::LG >= 16)) &&
                     (G::NPASS == 4 ? G::R3 : G::NPASS == 3 ? G::R2 : G::R1) % OV == 0);
}

template <typename T, int LOGM, int OV>
const void* kernel_ov(int mode) {            // mode: bit 0 ADMM, bit 2 evaluating (one-sided)
  if constexpr (!ola_fits<T, LOGM, OV>()) return nullptr;
  else {
    switch (mode & 5) {
      case 0: return (const void*)k_wave_iter<T, LOGM, 0, false, false, OV>;
      case 1: return (const void*)k_wave_iter<T, LOGM, 1, false, false, OV>;
      case 4: return (const void*)k_wave_iter<T, LOGM, 0, false, true, OV>;
      default: return (const void*)k_wave_iter<T, LOGM, 1, false, true, OV>;
    }
  }
}
template <typename T, int LOGM, int OV>
const void* kernel_two(int mode) {           // two-sided: frames form (OV 0) or the ring (OV 1)
  switch (mode & 5) {
    case 0: return (const void*)k_wave_iter<T, LOGM, 0, true, false, OV>;
    case 1: return (const void*)k_wave_iter<T, LOGM, 1, true, false, OV>;
    case 4: return (const void*)k_wave_iter<T, LOGM, 0, true, true, OV>;
    default: return (const void*)k_wave_iter<T, LOGM, 1, true, true, OV>;
  }
}
template <typename T, int LOGM>
const void* kernel_of(int mode, int ov) {    // mode: bit 0 ADMM, bit 1 two-sided, bit 2 evaluating
  if (mode & 2) return ov == 1 ? kernel_two<T, LOGM, 1>(mode) : kernel_two<T, LOGM, 0>(mode);
  switch (ov) {
    case 1: return kernel_ov<T, LOGM, 1>(mode);
    case 2: return kernel_ov<T, LOGM, 2>(mode);
    case 4: return kernel_ov<T, LOGM, 4>(mode);
    case 8: return kernel_ov<T, LOGM, 8>(mode);
    default: return kernel_ov<T, LOGM, 0>(mode);
  }
}

// `work`: frames (ov == 0) or chunks (ov > 0) the launch has to cover
template <typename T, int LOGM>
Launch shape(int64_t work, int mode, int ov) {
  using G = Geo<T, LOGM>;
  constexpr int M = m_of<LOGM>(), TEAM = G::LG > 64 ? G::LG / 64 : 1, FPW = G::LG > 64 ? 1 : 64 / G::LG, PS = G::R0, MP = phys<PS>(M) + 1;
  static int n_cu = 0;
  // workgroups of four or eight waves (each carries its own twiddle table), whichever puts more waves on a CU by the runtime's
  // own count of resident workgroups (registers and LDS); one launch fills the chip once and every wave walks its share of frames
  static int wpw_of[64] = {}, per_cu_of[64] = {};
  const int key = (mode & 7) | (ov == 2 ? 8 : ov == 4 ? 16 : ov == 8 ? 24 : 0) | (ov == 1 ? 32 : 0);
  // (a team - a frame on the lanes of TEAM waves - is a workgroup of its own: `w` counts its one frame;
  // + the ring of n_fft samples per lane group where the overlap-add runs in LDS)
  auto lds_of = [&](int w) {
    const size_t groups_wg = (size_t)(TEAM > 1 ? 1 : w) * FPW;
    return sizeof(cplx<T>) * ((size_t)Tabs<T, LOGM>::TOTAL + groups_wg * MP) + (ov == 1 ? groups_wg * 2 * M * sizeof(T) : 0);
  };
  const void* fn = kernel_of<T, LOGM>(mode, ov);
  static std::mutex mu;                                  // (plans of several host threads may ask at once)
  std::lock_guard<std::mutex> lock(mu);
  if (wpw_of[key] == 0) {
    int dev = 0;
    hipDeviceProp_t prop{};
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
    if (n_cu <= 0) n_cu = 256;
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::min<size_t>(lds_of(8), 160 * 1024));
    int best = 0;
    // (n_fft 400 / 800 / 1000: a wave's three float64 frames and rings are 20 KB - any size from two waves up, so that the LDS is
    // not left half empty by the granularity of four)
    for (int w : {4, 8, 7, 6, 5, 3, 2}) {
      int nb = 0;
      if (LOGM < 100 && w != 4 && w != 8) continue;
      if (TEAM > 1) w = TEAM;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 64 * w, lds_of(w)) != hipSuccess) nb = 0;
      if (const char* e = getenv("SPECINV_WAVE_WPW")) {
        if (TEAM == 1 && atoi(e) != w) continue;
      }
      if (nb * w > best) {
        best = nb * w;
        wpw_of[key] = w;
        per_cu_of[key] = nb;
      }
    }
    (void)hipGetLastError();
    if (wpw_of[key] == 0) {
      wpw_of[key] = 4;
      per_cu_of[key] = 1;
    }
  }
  const int wpw = wpw_of[key];
  const int64_t groups = (work + FPW - 1) / FPW;
  Launch l;
  l.waves_per_wg = wpw;
  const int64_t wg_work = TEAM > 1 ? groups : (groups + wpw - 1) / wpw;      // workgroups the work fills: a team takes a frame
  l.wgs = (int)std::max<int64_t>(1, std::min<int64_t>(wg_work, (int64_t)n_cu * per_cu_of[key]));
  l.capacity = TEAM > 1 ? n_cu * per_cu_of[key] : n_cu * per_cu_of[key] * wpw * FPW;
  l.lds = lds_of(wpw);
  return l;
}

// chunks per item of the register overlap-add: enough units to fill the chip once, chunks of at least max(2 OV, 8) frames
// does the register overlap-add apply to hop = n_fft / ov here?
template <typename T, int LOGM>
bool ola_registers(int ov) {
  return ov == 2 ? ola_fits<T, LOGM, 2>() : ov == 4 ? ola_fits<T, LOGM, 4>() : ov == 8 ? ola_fits<T, LOGM, 8>() : false;
}
// `ov`: 2 / 4 / 8 registers, 1 the ring (frames_over = ceil(n_fft / hop) frames cover a sample)
template <typename T, int LOGM>
int ola_chunks(int ov, int frames_over, int n_frames, int batch, int mode) {
  if (!wave_iter_fits(2 * m_of<LOGM>(), n_frames, batch, true)) return 0;
  const int min_len = std::max(2 * frames_over, 8);
  if (n_frames < min_len) return 0;
  if (const char* e = getenv("SPECINV_WAVE_OLA")) {
    if (e[0] == '0') return 0;
  }
  const Launch l = shape<T, LOGM>(1, mode, ov);
  int nch = (int)std::max<int64_t>(1, (int64_t)l.capacity / std::max(1, batch));
  if (const char* e = getenv("SPECINV_WAVE_CHUNK")) nch = std::max(1, n_frames / std::max(1, atoi(e)));
  return std::max(1, std::min(nch, n_frames / min_len));
}

template <typename T, int LOGM>
int launch_one(const WaveIterArgs<T>& a, hipStream_t stream, int* waves_out) {
  const int mode = (a.mode & 1) | (a.c.onesided ? 0 : 2) | (a.eval ? 4 : 0);
  const int ov = a.nch > 0 ? a.ov : 0;
  const void* fn = kernel_of<T, LOGM>(mode, ov);
  SI_CHECK(fn != nullptr && (ov == 0 || (a.x_out && a.env && a.seamL && a.seamR && a.x_out != a.x)) && (ov != 1 || a.c.hop < a.c.n_fft),
           SPECINV_EINVAL, "k_wave_iter: bad overlap-add arguments");
  // (frames are indexed with 32 bits; the lane groups of a wave address their rows relative to the first group's, with the register
  // overlap-add up to a chunk apart - the plan keeps such shapes on the frames form / the workgroup kernels: wave_iter_fits)
  SI_CHECK(wave_iter_fits(a.c.n_fft, a.c.n_frames, a.batch, ov > 0), SPECINV_EUNSUPPORTED, "k_wave_iter: too many frames for 32-bit frame offsets");
  const Launch l = shape<T, LOGM>(ov > 0 ? (int64_t)a.batch * a.nch : (int64_t)a.batch * a.c.n_frames, mode, ov);
  if (waves_out) *waves_out = l.wgs * l.waves_per_wg;
  WaveIterArgs<T> args = a;
  void* kargs[] = {&args};
  SI_HIP(hipLaunchKernel(fn, dim3(l.wgs), dim3(64 * l.waves_per_wg), kargs, l.lds, stream));
  if (ov > 0) {
    const int keep = ov == 1 ? a.c.n_fft - a.c.hop : (ov - 1) * a.c.hop;
    const int64_t total = (int64_t)a.batch * (a.nch + 1) * keep;
    hipLaunchKernelGGL((k_wave_seams<T>), dim3((unsigned)std::min<int64_t>(4096, (total + 255) / 256)), dim3(256), 0, stream,
                       (const T*)a.seamL, (const T*)a.seamR, a.env, a.x_out, a.c.n_frames, a.nch, keep, a.c.hop, a.c.pad, a.c.length, a.batch);
    SI_HIP(hipGetLastError());
  }
  return SPECINV_OK;
}

}  // namespace wave
}  // namespace specinv
