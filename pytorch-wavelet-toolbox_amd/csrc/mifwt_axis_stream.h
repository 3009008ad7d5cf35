// mifwt_axis_stream.h — streaming single-axis analysis / synthesis kernels for gfx950 (kernel ids 3-6).
//
// One transformed axis per launch, but — unlike the catch-all of mifwt_generic.hip — organised for HBM:
//   * OUTER-axis kernels (the transformed axis is NOT the innermost one; the data under it is a dense run of
//     `inner` elements): a lane owns 16 bytes of that run, a wavefront 1 KiB, and walks down the transformed
//     axis with the filter window in a register ring — every input row is loaded once per chunk, every
//     load / store is one fully coalesced 1 KiB wave access, no LDS, no barriers.  This is the depth pass of
//     the 3-D transforms (fused 2-D plane kernel + this) and the row pass of the separable N-D composition.
//   * INNER-axis kernels (the transformed axis is the contiguous one): a lane produces 4 (f32/f16) or 2 (f64)
//     consecutive coefficients (analysis) / samples (synthesis) of one row from a window of 2*4+L-2 inputs
//     it loads itself; neighbouring lanes' windows overlap, the overlap is served by the vector L1.  Boundary
//     lanes (window leaves [0, N)) take a per-element index-mapped path.  This is the 1-D transform
//     (wavedec / waverec, reference src/ptwt/conv_transform.py:135-139, :184-199).
// Templated on the storage type T (float, double, _Float16 with float arithmetic) and the even filter
// length L; (lo, hi) taps travel in the kernel arguments.
//
// Math:
//   analysis  c[k] = sum_m h[m] * x_ext[2k + 1 - m],                              k in [0, M)
//   synthesis y[2p + r] = sum_{i < L/2} g_lo[L-2-2i+r] a[p+i] + g_hi[L-2-2i+r] d[p+i],   2p + r in [0, Nout)
#pragma once
#include "mifwt_stream.h"

namespace mifwt {

template <typename T>
struct ElemTraits;
template <>
struct ElemTraits<float> {
  using Acc = float;
  static constexpr int EV = 4;  // elements per 16-byte lane access (outer kernels)
  static constexpr int EO = 4;  // outputs per lane (inner kernels)
};
template <>
struct ElemTraits<double> {
  using Acc = double;
  static constexpr int EV = 2;
  static constexpr int EO = 2;
};
template <>
struct ElemTraits<_Float16> {
  using Acc = float;
  static constexpr int EV = 8;
  static constexpr int EO = 4;
};

// N consecutive elements of T at an address that is only guaranteed to be aligned like T itself (odd row
// pitches): the hardware takes dword-aligned wide global accesses, the typedefs say so to the compiler.
template <typename T, int N>
struct VecOf {
  typedef T type __attribute__((ext_vector_type(N), aligned(sizeof(T))));
};
template <typename T>
struct VecOf<T, 1> {
  typedef T type;
};

template <typename T, typename A, int N>
__device__ __forceinline__ void load_run(const T* __restrict__ p, A (&dst)[N]) {
  const typename VecOf<T, N>::type v = *reinterpret_cast<const typename VecOf<T, N>::type*>(p);
  if constexpr (N == 1) {
    dst[0] = (A)v;
  } else {
#pragma unroll
    for (int e = 0; e < N; ++e) dst[e] = (A)v[e];
  }
}
template <typename T, typename A, int N>
__device__ __forceinline__ void store_run(T* __restrict__ p, const A (&src)[N]) {
  if constexpr (N == 1) {
    *p = (T)src[0];
  } else {
    typename VecOf<T, N>::type v;
#pragma unroll
    for (int e = 0; e < N; ++e) v[e] = (T)src[e];
    *reinterpret_cast<typename VecOf<T, N>::type*>(p) = v;
  }
}

template <typename A, int L>
struct StreamArgs {
  StreamJob job[4];
  int njobs;
  int mode;
  int n_in;    // analysis: N (signal extent)        synthesis: M (coefficient extent)
  int n_out;   // analysis: M                        synthesis: Nout (cropped extent)
  int batch;   // outer kernels: folded batch
  int rows[3]; // inner kernels: row dims
  int64_t inner;   // outer kernels: dense run under the transformed axis (elements)
  int nstrips;     // outer: 64-lane strips over `inner`;   inner: 64-lane segments over n_out
  int nchunks;     // outer: chunks over the output rows;   inner: row groups
  int per_chunk;   // outer: output rows (analysis) / row pairs (synthesis) per chunk;  inner: rows per group
  int64_t ntasks;  // waves with work
  A lo[L], hi[L];  // taps, PyWavelets order
  DevTapArg dt;    // device-resident taps (mifwt_common.h); dt.lo == nullptr: lo / hi count
};

// the taps of a pass: by value, or (a learnable filter bank that lives on the GPU) read once from device memory
template <typename A, int L>
__device__ __forceinline__ void stream_taps(const StreamArgs<A, L>& a, A (&tlo)[L], A (&thi)[L]) {
  if (a.dt.lo) {
#pragma unroll
    for (int m = 0; m < L; ++m) tlo[m] = dtap_lo<A>(a.dt, m), thi[m] = dtap_hi<A>(a.dt, m);
  } else {
#pragma unroll
    for (int m = 0; m < L; ++m) tlo[m] = a.lo[m], thi[m] = a.hi[m];
  }
}

// ------------------------------------------------------------------------------------------------------
// OUTER axis, analysis.  Task = (strip of 64*EV inner elements) x (chunk of output rows) x batch x job.
template <typename T, int L>
__global__ void __launch_bounds__(256) outer_fwd_kernel(const StreamArgs<typename ElemTraits<T>::Acc, L> a) {
  using A = typename ElemTraits<T>::Acc;
  A tlo[L], thi[L];
  stream_taps(a, tlo, thi);
  constexpr int E = ElemTraits<T>::EV;
  constexpr int RING = L + 2, U = RING / 2;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int64_t task = (int64_t)xcd_remap(blockIdx.x, gridDim.x) * 4 + wave;
  if (task >= a.ntasks) return;
  const int strip = (int)(task % a.nstrips);
  task /= a.nstrips;
  const int chunk = (int)(task % a.nchunks);
  task /= a.nchunks;
  const int b = (int)(task % a.batch);
  const StreamJob& jb = a.job[(int)(task / a.batch)];

  const int64_t j0 = ((int64_t)strip * 64 + lane) * E;
  if (j0 >= a.inner) return;
  const bool full = j0 + E <= a.inner;
  const int nvalid = full ? E : (int)(a.inner - j0);
  const T* __restrict__ xp = static_cast<const T*>(jb.in0) + (int64_t)b * jb.in0_s[0] + j0;
  T* __restrict__ lop = static_cast<T*>(jb.out0) + (int64_t)b * jb.out0_s[0] + j0;
  T* __restrict__ hip_ = static_cast<T*>(jb.out1) + (int64_t)b * jb.out1_s[0] + j0;
  const int k0 = chunk * a.per_chunk;
  const int k1 = min(k0 + a.per_chunk, a.n_out);

  auto load_row = [&](int n, A(&dst)[E]) {
    const int src = ext_index(n, a.n_in, a.mode);  // wave-uniform
    if (src < 0) {
#pragma unroll
      for (int e = 0; e < E; ++e) dst[e] = A(0);
      return;
    }
    const T* p = xp + (int64_t)src * jb.in0_s[1];
    if (full) {
      load_run<T, A, E>(p, dst);
    } else {
#pragma unroll
      for (int e = 0; e < E; ++e) dst[e] = e < nvalid ? (A)p[e] : A(0);
    }
  };

  A ring[RING][E];
  const int n_first = 2 * k0 - (L - 2);  // extended row held in ring slot 0
#pragma unroll
  for (int t = 0; t < L; ++t) load_row(n_first + t, ring[t]);

  for (int g = 0;; ++g) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int s = g * U + u;
      const int k = k0 + s;
      if (k >= k1) return;
      // rows of the next step go into the two slots the previous step released
      load_row(n_first + 2 * s + L, ring[(2 * u + L) % RING]);
      load_row(n_first + 2 * s + L + 1, ring[(2 * u + L + 1) % RING]);
      A lo[E], hi[E];
#pragma unroll
      for (int m = 0; m < L; ++m) {
        const A(&r)[E] = ring[(2 * u + L - 1 - m) % RING];  // row 2k + 1 - m
#pragma unroll
        for (int e = 0; e < E; ++e) {
          lo[e] = m == 0 ? tlo[0] * r[e] : fma(tlo[m], r[e], lo[e]);
          hi[e] = m == 0 ? thi[0] * r[e] : fma(thi[m], r[e], hi[e]);
        }
      }
      T* lp = lop + (int64_t)k * jb.out0_s[1];
      T* hp = hip_ + (int64_t)k * jb.out1_s[1];
      if (full) {
        store_run<T, A, E>(lp, lo);
        store_run<T, A, E>(hp, hi);
      } else {
#pragma unroll
        for (int e = 0; e < E; ++e)
          if (e < nvalid) {
            lp[e] = (T)lo[e];
            hp[e] = (T)hi[e];
          }
      }
    }
  }
}

// OUTER axis, synthesis.  Task = strip x (chunk of output row PAIRS) x batch x job.
template <typename T, int L>
__global__ void __launch_bounds__(256) outer_inv_kernel(const StreamArgs<typename ElemTraits<T>::Acc, L> a) {
  using A = typename ElemTraits<T>::Acc;
  A tlo[L], thi[L];
  stream_taps(a, tlo, thi);
  constexpr int E = ElemTraits<T>::EV;
  constexpr int HL = L / 2, RING = HL + 1, U = RING;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int64_t task = (int64_t)xcd_remap(blockIdx.x, gridDim.x) * 4 + wave;
  if (task >= a.ntasks) return;
  const int strip = (int)(task % a.nstrips);
  task /= a.nstrips;
  const int chunk = (int)(task % a.nchunks);
  task /= a.nchunks;
  const int b = (int)(task % a.batch);
  const StreamJob& jb = a.job[(int)(task / a.batch)];

  const int64_t j0 = ((int64_t)strip * 64 + lane) * E;
  if (j0 >= a.inner) return;
  const bool full = j0 + E <= a.inner;
  const int nvalid = full ? E : (int)(a.inner - j0);
  const T* __restrict__ ap = static_cast<const T*>(jb.in0) + (int64_t)b * jb.in0_s[0] + j0;
  const T* __restrict__ dp = static_cast<const T*>(jb.in1) + (int64_t)b * jb.in1_s[0] + j0;
  T* __restrict__ yp = static_cast<T*>(jb.out0) + (int64_t)b * jb.out0_s[0] + j0;
  const int npairs = (a.n_out + 1) >> 1;
  const int p0 = chunk * a.per_chunk;
  const int p1 = min(p0 + a.per_chunk, npairs);

  auto load_rows = [&](int m, A(&da)[E], A(&dd)[E]) {
    m = m < a.n_in ? m : a.n_in - 1;  // rows past the band are never used; keep the load in range
    const T* pa = ap + (int64_t)m * jb.in0_s[1];
    const T* pd = dp + (int64_t)m * jb.in1_s[1];
    if (full) {
      load_run<T, A, E>(pa, da);
      load_run<T, A, E>(pd, dd);
    } else {
#pragma unroll
      for (int e = 0; e < E; ++e) {
        da[e] = e < nvalid ? (A)pa[e] : A(0);
        dd[e] = e < nvalid ? (A)pd[e] : A(0);
      }
    }
  };

  A ra[RING][E], rd[RING][E];
#pragma unroll
  for (int t = 0; t < HL; ++t) load_rows(p0 + t, ra[t], rd[t]);

  for (int g = 0;; ++g) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int s = g * U + u;
      const int p = p0 + s;
      if (p >= p1) return;
      load_rows(p + HL, ra[(u + HL) % RING], rd[(u + HL) % RING]);
      A y0[E], y1[E];
#pragma unroll
      for (int i = 0; i < HL; ++i) {
        const A(&va)[E] = ra[(u + i) % RING];
        const A(&vd)[E] = rd[(u + i) % RING];
        const A gl0 = tlo[L - 2 - 2 * i], gl1 = tlo[L - 1 - 2 * i];
        const A gh0 = thi[L - 2 - 2 * i], gh1 = thi[L - 1 - 2 * i];
#pragma unroll
        for (int e = 0; e < E; ++e) {
          y0[e] = i == 0 ? gl0 * va[e] : fma(gl0, va[e], y0[e]);
          y1[e] = i == 0 ? gl1 * va[e] : fma(gl1, va[e], y1[e]);
          y0[e] = fma(gh0, vd[e], y0[e]);
          y1[e] = fma(gh1, vd[e], y1[e]);
        }
      }
      T* r0 = yp + (int64_t)(2 * p) * jb.out0_s[1];
      T* r1 = r0 + jb.out0_s[1];
      const bool has1 = 2 * p + 1 < a.n_out;
      if (full) {
        store_run<T, A, E>(r0, y0);
        if (has1) store_run<T, A, E>(r1, y1);
      } else {
#pragma unroll
        for (int e = 0; e < E; ++e)
          if (e < nvalid) {
            r0[e] = (T)y0[e];
            if (has1) r1[e] = (T)y1[e];
          }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// INNER axis, analysis.  Task = (segment of 64*EO coefficients) x (group of rows) x job.
template <typename T, int L>
__global__ void __launch_bounds__(256) inner_fwd_kernel(const StreamArgs<typename ElemTraits<T>::Acc, L> a) {
  using A = typename ElemTraits<T>::Acc;
  A tlo[L], thi[L];
  stream_taps(a, tlo, thi);
  constexpr int EO = ElemTraits<T>::EO;
  constexpr int WN = 2 * EO + L - 2;                          // window: extended columns 2k0-(L-2) .. 2k0+2EO-1
  constexpr int LV = sizeof(T) == 8 ? 2 : 4;                  // elements per window load
  constexpr int NLD = (WN + LV - 1) / LV;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int64_t task = (int64_t)xcd_remap(blockIdx.x, gridDim.x) * 4 + wave;
  if (task >= a.ntasks) return;
  const int seg = (int)(task % a.nstrips);
  task /= a.nstrips;
  const int grp = (int)(task % a.nchunks);
  const StreamJob& jb = a.job[(int)(task / a.nchunks)];

  const int k0 = (seg * 64 + lane) * EO;
  if (k0 >= a.n_out) return;
  const bool full_out = k0 + EO <= a.n_out;
  const int c0 = 2 * k0 - (L - 2);
  const bool interior = c0 >= 0 && c0 + NLD * LV <= a.n_in;
  // per-element source columns of a boundary lane (-1: implicit zero)
  int src[WN];
  if (!interior) {
#pragma unroll
    for (int i = 0; i < WN; ++i) src[i] = ext_index(c0 + i, a.n_in, a.mode);
  }
  const int nrows = a.rows[0] * a.rows[1] * a.rows[2];
  const int r0 = grp * a.per_chunk;
  const int r1 = min(r0 + a.per_chunk, nrows);

  auto row_offsets = [&](int r, int64_t& xo, int64_t& lo_o, int64_t& hi_o) {
    const int i2 = r % a.rows[2];
    const int t = r / a.rows[2];
    const int i1 = t % a.rows[1];
    const int i0 = t / a.rows[1];
    xo = i0 * jb.in0_s[0] + i1 * jb.in0_s[1] + i2 * jb.in0_s[2];
    lo_o = i0 * jb.out0_s[0] + i1 * jb.out0_s[1] + i2 * jb.out0_s[2];
    hi_o = i0 * jb.out1_s[0] + i1 * jb.out1_s[1] + i2 * jb.out1_s[2];
  };
  auto load_window = [&](const T* __restrict__ xr, A(&w)[NLD * LV]) {
    if (interior) {
#pragma unroll
      for (int c = 0; c < NLD; ++c) {
        A t[LV];
        load_run<T, A, LV>(xr + c0 + c * LV, t);
#pragma unroll
        for (int e = 0; e < LV; ++e) w[c * LV + e] = t[e];
      }
    } else {
#pragma unroll
      for (int i = 0; i < WN; ++i) w[i] = src[i] >= 0 ? (A)xr[src[i]] : A(0);
    }
  };

  A cur[NLD * LV], nxt[NLD * LV];
  int64_t xo, lo_o, hi_o;
  row_offsets(r0, xo, lo_o, hi_o);
  load_window(static_cast<const T*>(jb.in0) + xo, cur);
  for (int r = r0; r < r1; ++r) {
    int64_t xo_n, lo_n, hi_n;
    row_offsets(r + 1 < r1 ? r + 1 : r, xo_n, lo_n, hi_n);
    load_window(static_cast<const T*>(jb.in0) + xo_n, nxt);  // next row in flight while this one is filtered
    A lo[EO], hi[EO];
#pragma unroll
    for (int e = 0; e < EO; ++e) {
#pragma unroll
      for (int m = 0; m < L; ++m) {
        const A v = cur[2 * e + L - 1 - m];  // extended column 2(k0+e) + 1 - m
        lo[e] = m == 0 ? tlo[0] * v : fma(tlo[m], v, lo[e]);
        hi[e] = m == 0 ? thi[0] * v : fma(thi[m], v, hi[e]);
      }
    }
    T* lp = static_cast<T*>(jb.out0) + lo_o + k0;
    T* hp = static_cast<T*>(jb.out1) + hi_o + k0;
    if (full_out) {
      store_run<T, A, EO>(lp, lo);
      store_run<T, A, EO>(hp, hi);
    } else {
#pragma unroll
      for (int e = 0; e < EO; ++e)
        if (k0 + e < a.n_out) {
          lp[e] = (T)lo[e];
          hp[e] = (T)hi[e];
        }
    }
#pragma unroll
    for (int i = 0; i < NLD * LV; ++i) cur[i] = nxt[i];
    lo_o = lo_n;
    hi_o = hi_n;
  }
}

// INNER axis, synthesis.  Task = (segment of 64*EO output samples) x (group of rows) x job.
template <typename T, int L>
__global__ void __launch_bounds__(256) inner_inv_kernel(const StreamArgs<typename ElemTraits<T>::Acc, L> a) {
  using A = typename ElemTraits<T>::Acc;
  A tlo[L], thi[L];
  stream_taps(a, tlo, thi);
  constexpr int EO = ElemTraits<T>::EO;   // even
  constexpr int HL = L / 2;
  constexpr int WN = HL + EO / 2 - 1;     // coefficients p0 .. p0 + WN - 1 of each band
  constexpr int LV = 2;                   // elements per window load (p0 is even for EO = 4)
  constexpr int NLD = (WN + LV - 1) / LV;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int64_t task = (int64_t)xcd_remap(blockIdx.x, gridDim.x) * 4 + wave;
  if (task >= a.ntasks) return;
  const int seg = (int)(task % a.nstrips);
  task /= a.nstrips;
  const int grp = (int)(task % a.nchunks);
  const StreamJob& jb = a.job[(int)(task / a.nchunks)];

  const int n0 = (seg * 64 + lane) * EO;
  if (n0 >= a.n_out) return;
  const bool full_out = n0 + EO <= a.n_out;
  const int p0 = n0 >> 1;
  const bool interior = p0 + NLD * LV <= a.n_in;
  const int nrows = a.rows[0] * a.rows[1] * a.rows[2];
  const int r0 = grp * a.per_chunk;
  const int r1 = min(r0 + a.per_chunk, nrows);

  auto row_offsets = [&](int r, int64_t& ao, int64_t& d_o, int64_t& yo) {
    const int i2 = r % a.rows[2];
    const int t = r / a.rows[2];
    const int i1 = t % a.rows[1];
    const int i0 = t / a.rows[1];
    ao = i0 * jb.in0_s[0] + i1 * jb.in0_s[1] + i2 * jb.in0_s[2];
    d_o = i0 * jb.in1_s[0] + i1 * jb.in1_s[1] + i2 * jb.in1_s[2];
    yo = i0 * jb.out0_s[0] + i1 * jb.out0_s[1] + i2 * jb.out0_s[2];
  };
  auto load_window = [&](const T* __restrict__ row, A(&w)[NLD * LV]) {
    if (interior) {
#pragma unroll
      for (int c = 0; c < NLD; ++c) {
        A t[LV];
        load_run<T, A, LV>(row + p0 + c * LV, t);
#pragma unroll
        for (int e = 0; e < LV; ++e) w[c * LV + e] = t[e];
      }
    } else {
#pragma unroll
      for (int i = 0; i < WN; ++i) w[i] = p0 + i < a.n_in ? (A)row[p0 + i] : A(0);
    }
  };

  A ca[NLD * LV], cd[NLD * LV], na[NLD * LV], nd[NLD * LV];
  int64_t ao, d_o, yo;
  row_offsets(r0, ao, d_o, yo);
  load_window(static_cast<const T*>(jb.in0) + ao, ca);
  load_window(static_cast<const T*>(jb.in1) + d_o, cd);
  for (int r = r0; r < r1; ++r) {
    int64_t ao_n, do_n, yo_n;
    row_offsets(r + 1 < r1 ? r + 1 : r, ao_n, do_n, yo_n);
    load_window(static_cast<const T*>(jb.in0) + ao_n, na);
    load_window(static_cast<const T*>(jb.in1) + do_n, nd);
    A y[EO];
#pragma unroll
    for (int e = 0; e < EO; ++e) {
      const int pp = e >> 1, rr = e & 1;  // sample n0 + e = 2 (p0 + pp) + rr
#pragma unroll
      for (int i = 0; i < HL; ++i) {
        const A gl = tlo[L - 2 - 2 * i + rr], gh = thi[L - 2 - 2 * i + rr];
        y[e] = i == 0 ? gl * ca[pp + i] : fma(gl, ca[pp + i], y[e]);
        y[e] = fma(gh, cd[pp + i], y[e]);
      }
    }
    T* yp = static_cast<T*>(jb.out0) + yo + n0;
    if (full_out) {
      store_run<T, A, EO>(yp, y);
    } else {
#pragma unroll
      for (int e = 0; e < EO; ++e)
        if (n0 + e < a.n_out) yp[e] = (T)y[e];
    }
#pragma unroll
    for (int i = 0; i < NLD * LV; ++i) {
      ca[i] = na[i];
      cd[i] = nd[i];
    }
    yo = yo_n;
  }
}

// ------------------------------------------------------------------------------------------------------
// host-side launchers (one translation unit per storage type instantiates them)
template <typename T, int L>
int stream_launch(int kind, const StreamCall& c) {  // kind: 0 outer fwd, 1 outer inv, 2 inner fwd, 3 inner inv
  using A = typename ElemTraits<T>::Acc;
  StreamArgs<A, L> a;
  if (c.njobs < 1 || c.njobs > 4) return MIFWT_ERR_BADARG;
  for (int i = 0; i < 4; ++i) a.job[i] = c.jobs[i < c.njobs ? i : 0];
  a.njobs = c.njobs;
  a.mode = c.mode;
  if (c.n_in > INT32_MAX / 4 || c.n_out > INT32_MAX / 4) return MIFWT_ERR_UNSUPPORTED;
  a.n_in = (int)c.n_in;
  a.n_out = (int)c.n_out;
  for (int m = 0; m < L; ++m) {
    a.lo[m] = (A)c.lo[m];
    a.hi[m] = (A)c.hi[m];
  }
  a.dt = dev_tap_arg(L);
  int64_t ntasks;
  if (kind < 2) {
    if (c.batch > INT32_MAX) return MIFWT_ERR_UNSUPPORTED;
    a.batch = (int)c.batch;
    a.rows[0] = a.rows[1] = a.rows[2] = 1;
    a.inner = c.inner;
    const int64_t per_strip = 64 * ElemTraits<T>::EV;
    const int64_t nstrips = (c.inner + per_strip - 1) / per_strip;
    const int64_t units = kind == 0 ? c.n_out : (c.n_out + 1) / 2;  // rows / row pairs
    // chunk length along the transformed axis: long enough to amortise the L-2 (L/2-1) halo rows a chunk
    // re-reads, short enough that the launch has >= ~8 waves per SIMD of parallelism
    int64_t per_chunk = kind == 0 ? 16 : 8;
    const int64_t lanes_tasks = nstrips * c.batch * c.njobs;
    while (per_chunk < units && lanes_tasks * ((units + per_chunk - 1) / per_chunk) > 8 * 4 * 256 * 4) per_chunk *= 2;
    // ... and a SMALL launch (fewer waves than one per SIMD: the deep levels of a 3-D decomposition) is a chain of row requests a wave
    // waits for one after the other — shorter chunks, more of them: 32 x 54 x (4 x 31^2) db5, 16 -> 4 output rows a chunk: 25 -> 12 us
    // (the halo rows a chunk re-reads come from L2 there)
    const int64_t min_chunk = kind == 0 ? 4 : 2;
    while (per_chunk > min_chunk && lanes_tasks * ((units + per_chunk - 1) / per_chunk) < 4 * 256 * 4) per_chunk /= 2;
    if (per_chunk > units) per_chunk = units > 0 ? units : 1;
    a.per_chunk = (int)per_chunk;
    const int64_t nchunks = (units + per_chunk - 1) / per_chunk;
    if (nstrips > INT32_MAX || nchunks > INT32_MAX) return MIFWT_ERR_UNSUPPORTED;
    a.nstrips = (int)nstrips;
    a.nchunks = (int)nchunks;
    ntasks = nstrips * nchunks * c.batch * c.njobs;
  } else {
    const int64_t nrows = c.rows[0] * c.rows[1] * c.rows[2];
    if (nrows > INT32_MAX / 2) return MIFWT_ERR_UNSUPPORTED;
    for (int i = 0; i < 3; ++i) a.rows[i] = (int)c.rows[i];
    a.batch = 1;
    a.inner = 1;
    const int64_t per_seg = 64 * ElemTraits<T>::EO;
    const int64_t nsegs = (c.n_out + per_seg - 1) / per_seg;
    int64_t per_grp = 1;
    while (per_grp < 8 && per_grp * 2 <= nrows && nsegs * ((nrows + 2 * per_grp - 1) / (2 * per_grp)) * c.njobs >= 8 * 4 * 256 * 2)
      per_grp *= 2;
    a.per_chunk = (int)per_grp;
    const int64_t ngrp = (nrows + per_grp - 1) / per_grp;
    if (nsegs > INT32_MAX || ngrp > INT32_MAX) return MIFWT_ERR_UNSUPPORTED;
    a.nstrips = (int)nsegs;
    a.nchunks = (int)ngrp;
    ntasks = nsegs * ngrp * c.njobs;
  }
  a.ntasks = ntasks;
  if (ntasks == 0) return MIFWT_OK;
  const int64_t nblk = (ntasks + 3) / 4;
  if (nblk > INT32_MAX) return MIFWT_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)nblk), block(256);
  switch (kind) {
    case 0: hipLaunchKernelGGL((outer_fwd_kernel<T, L>), grid, block, 0, c.stream, a); break;
    case 1: hipLaunchKernelGGL((outer_inv_kernel<T, L>), grid, block, 0, c.stream, a); break;
    case 2: hipLaunchKernelGGL((inner_fwd_kernel<T, L>), grid, block, 0, c.stream, a); break;
    default: hipLaunchKernelGGL((inner_inv_kernel<T, L>), grid, block, 0, c.stream, a); break;
  }
  return hipGetLastError() == hipSuccess ? MIFWT_OK : MIFWT_ERR_LAUNCH;
}

}  // namespace mifwt

// One translation unit per (storage type, filter-length group) defines stream_call_<type>_L<len> for its
// lengths, so that the extension builds in parallel (mifwt_axis_stream_*.hip; table in mifwt_compose.hip).
#define MIFWT_STREAM_DEFINE(NAME, TYPE, LEN) \
  namespace mifwt { int stream_call_##NAME##_L##LEN(int kind, const StreamCall& c) { return stream_launch<TYPE, LEN>(kind, c); } }
