// mifwt_stream.h — idioms shared by the fused streaming kernels (gfx950).
#pragma once
#include "mifwt_common.h"

namespace mifwt {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
// 8/16-byte vectors that are only guaranteed 4-byte aligned (odd row pitches such as 515 floats)
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

// v_pk_fma_f32 with a PACKED TAP PAIR and a BROADCAST sample:
//   acc(.x, .y) += (tap.x, tap.y) * sample,  sample = low (…_lo) or high (…_hi) half of an even-aligned VGPR pair.
// The tap pair lives in SGPRs (one 64-bit scalar operand), the sample needs no pairing of its own.
// (Inline assembly: each FMA is opaque to the compiler, whose hazard recogniser keeps four wait states between an asm statement and the
// next one that touches its result — a round of FOUR accumulation chains costs one s_nop (1048 s_nop among the 8453 instructions of the
// 16-tap streaming kernel): callers interleave at least five independent chains.  Round 6 measured the compiler's own packed FMAs
// (-DMIFWT_BUILTIN_FMA: __builtin_elementwise_fma on the broadcast sample selects the same v_pk_fma_f32 with an SGPR tap pair, no
// s_nop): its scheduler then hoists loads and copies accumulators — the 16-tap streaming kernels go to 256 registers + scratch,
// config 4 2.37 -> 2.77 / 3.2 ms; configs 2 / 3 within 1 % (profiles/r06c_fma_builtin_ab.txt).  Assembly stays.)
#ifndef MIFWT_BUILTIN_FMA
__device__ __forceinline__ void pkfma_lo(f2& acc, const f2 tap, const f2 pair) {
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "s"(tap), "v"(pair));
}
__device__ __forceinline__ void pkfma_hi(f2& acc, const f2 tap, const f2 pair) {
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc) : "s"(tap), "v"(pair));
}
__device__ __forceinline__ f2 pkmul_lo(const f2 tap, const f2 pair) {
  f2 r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "s"(tap), "v"(pair));
  return r;
}
__device__ __forceinline__ f2 pkmul_hi(const f2 tap, const f2 pair) {
  f2 r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(r) : "s"(tap), "v"(pair));
  return r;
}
#else
__device__ __forceinline__ void pkfma_lo(f2& acc, const f2 tap, const f2 pair) { acc = __builtin_elementwise_fma(tap, (f2){pair.x, pair.x}, acc); }
__device__ __forceinline__ void pkfma_hi(f2& acc, const f2 tap, const f2 pair) { acc = __builtin_elementwise_fma(tap, (f2){pair.y, pair.y}, acc); }
__device__ __forceinline__ f2 pkmul_lo(const f2 tap, const f2 pair) { return tap * (f2){pair.x, pair.x}; }
__device__ __forceinline__ f2 pkmul_hi(const f2 tap, const f2 pair) { return tap * (f2){pair.y, pair.y}; }
#endif

// Arithmetic layer of the tile kernels: f32 / f16 storage computes in packed f32 (the four asm helpers above), f64
// storage in double (no packed f64 FMA on gfx950: two v_fma_f64 per "packed" step).  a*_lo / a*_hi have the meaning of
// pk*_lo / pk*_hi: (acc.x, acc.y) (+)= (tap.x, tap.y) * pair.x  resp.  * pair.y.
typedef double d2 __attribute__((ext_vector_type(2)));
template <typename T> struct TileArith { typedef float type; typedef f2 vec2; };
template <> struct TileArith<double> { typedef double type; typedef d2 vec2; };
__device__ __forceinline__ f2 amul_lo(const f2 tap, const f2 pair) { return pkmul_lo(tap, pair); }
__device__ __forceinline__ f2 amul_hi(const f2 tap, const f2 pair) { return pkmul_hi(tap, pair); }
__device__ __forceinline__ void afma_lo(f2& acc, const f2 tap, const f2 pair) { pkfma_lo(acc, tap, pair); }
__device__ __forceinline__ void afma_hi(f2& acc, const f2 tap, const f2 pair) { pkfma_hi(acc, tap, pair); }
__device__ __forceinline__ d2 amul_lo(const d2 tap, const d2 pair) { return (d2){tap.x * pair.x, tap.y * pair.x}; }
__device__ __forceinline__ d2 amul_hi(const d2 tap, const d2 pair) { return (d2){tap.x * pair.y, tap.y * pair.y}; }
__device__ __forceinline__ void afma_lo(d2& acc, const d2 tap, const d2 pair) {
  acc.x = __builtin_fma(tap.x, pair.x, acc.x);
  acc.y = __builtin_fma(tap.y, pair.x, acc.y);
}
__device__ __forceinline__ void afma_hi(d2& acc, const d2 tap, const d2 pair) {
  acc.x = __builtin_fma(tap.x, pair.y, acc.x);
  acc.y = __builtin_fma(tap.y, pair.y, acc.y);
}

// Ordering of one wave's own LDS traffic (different lanes write and read the same slab): DS operations of a
// wave execute in order, this only stops the compiler from moving them across.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Boundary extension for an index at most ONE period outside [0, n) (what a level's pads of L-2 / L-1 samples need once
// n >= L), branch-free and without the general fallback of ext_index_near, whose integer division — inlined at every
// use — dominated the code size and the scalar-unit time of the tile kernels.  The mode is folded into three integers
// once per workgroup:  i < 0 -> lo_add + k i,  i >= n -> hi_add(n) + k i,  k = -1 (mirror modes), 0 (constant), +1 (periodic).
// Zero mode: the caller tests (unsigned)i >= n itself and requests nothing.
struct Fold1 {
  int kneg, kpos, sym, per;  // bit masks (0 / -1): mirror, periodic; sym = 1 for the half-sample mirror
  __device__ __forceinline__ void set(int mode) {
    kneg = (mode == MIFWT_MODE_REFLECT || mode == MIFWT_MODE_SYMMETRIC || mode == MIFWT_MODE_ZERO) ? -1 : 0;
    kpos = mode == MIFWT_MODE_PERIODIC ? -1 : 0;
    sym = mode == MIFWT_MODE_SYMMETRIC ? 1 : 0;
    per = kpos;
  }
  __device__ __forceinline__ int operator()(int i, int n) const {
    const int ki = (i & kpos) - (i & kneg);                                      // k * i
    const int lo_add = (n & per) - sym;                                           // n (periodic), -1 (symmetric), 0
    const int hi_add = kneg ? 2 * n - 2 + sym : ((n - 1) & ~per) - (n & per);     // 2n-2(+1) | n-1 (constant) | -n (periodic)
    return i < 0 ? lo_add + ki : (i >= n ? hi_add + ki : i);
  }
};

// Division of a workgroup index by a launch-time constant without the ~40-instruction software divide (v_rcp_iflag +
// Newton step + fix-ups) the compiler emits for a runtime divisor — three of them opened every tile kernel.  Round-up
// method (Granlund & Montgomery): q = (mulhi(m, n) + n) >> s, m = floor(2^32 (2^s - d) / d) + 1, s = ceil(log2 d);
// exact for n < 2^31 (the sum cannot overflow), any d >= 1.
struct FastDiv {
  uint32_t mul, shift, d;
  __device__ __forceinline__ uint32_t div(uint32_t n) const { return (__umulhi(mul, n) + n) >> shift; }
  __device__ __forceinline__ uint32_t divmod(uint32_t n, uint32_t& rem) const {
    const uint32_t q = div(n);
    rem = n - q * d;
    return q;
  }
};
inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f;
  f.d = d;
  uint32_t s = 0;
  while ((uint64_t(1) << s) < d) ++s;
  f.shift = s;
  f.mul = (uint32_t)(((uint64_t(1) << 32) * ((uint64_t(1) << s) - d)) / d + 1);
  return f;
}

// XCD-aware block remap (block b runs on XCD b % 8): every XCD gets a contiguous range of logical blocks so
// that tasks sharing halo rows / columns meet in one L2.  Bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

}  // namespace mifwt
