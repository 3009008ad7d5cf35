// mifwt_dwt1_long.hip — SEVERAL levels of a 1-D decomposition in one launch, a chunk of a row per workgroup (gfx950), kernel id 17.
//
// Reference seam: the leading trips of wavedec's level loop (src/ptwt/conv_transform.py:133-140: _fwt_pad + F.conv1d(stride 2)
// per level, the approximation fed back).  One kernel per level writes every intermediate approximation to HBM and reads it
// straight back, and from the third level on a launch is latency, not work (the reference's own 1-D speed test, 32 x 10^6
// samples db5 level 10, examples/speed_tests/timeitconv_1d.py:10-12: level 1 ran at 0.67 of the HBM peak, the whole call at
// 0.19).  Here a workgroup owns a CHUNK of a row plus the (L - 2)(2^K - 1) halo samples its K levels consume:
//   * level 1 reads its windows straight from global memory (16-byte buffer loads, two adjacent outputs per lane, the lanes'
//     windows 16 bytes apart; the overlap is served by the vector L1) — nothing is parked, so the LDS footprint is the level-1
//     and level-2 rows only (25 KB: six workgroups per CU.  Parking the chunk first: 48 KB, three per CU, 76 us against 61);
//   * levels 2 .. K ping-pong between two LDS buffers (16-byte reads, conflict-free at that lane stride; 8-byte reads hit
//     every bank four times); detail coefficients go to HBM through a buffer resource that spans the owned outputs, so
//     nothing is branched around; the level-K approximation of the chunk is stored at the end.
// mifwt_dwt1_fwd_tail finishes the pyramid once a row fits into one workgroup — unless there are too few rows to occupy the
// chip that way: rows of >= 4096 samples are then still cut into (smaller) chunks, about one workgroup per CU.
//   c_lo/hi[k] = sum_m h_lo/hi[m] x_ext[2k + 1 - m],  k < floor((n + L - 1) / 2)     
// Boundary extension: the two ENDS of a row belong to one workgroup (two pieces, left and right) — the only place where an
// extended sample is ever read: the few outputs next to an end go through the boundary index map, one lane per (output, tap),
// and in periodic mode each piece finds the wrapped samples in the other one.  Interior chunks are map-free.
// The launch is bound by instruction issue, not by bandwidth (with loads and stores switched off: 45 of 61 us): see the
// note in front of dwt1_long_body.  Measured (MI355X, 32 x 10^6 f32 db5, tools/long1d_time.py): levels 1-6 in 61 us = 4.2 TB/s
// algorithmic (0.52 of 8 TB/s), levels 7-10 (32 rows of 15 633 samples, 288 workgroups) in 17 us, the whole level-10 call
// 73 us = 0.44 of the HBM peak on its compulsory bytes (one kernel per level + the one-workgroup-per-row tail: 165 us, 0.19).
// Bound: HBM.  Algorithmic bytes: 4 B (n_0 + n_1 + ... + n_K + n_K) per row (input once, every detail band and the last
// approximation once); the halo is re-read by the neighbouring chunk (6 % for db5, K = 6, 8 K-sample chunks).
// f32, even L <= 20, every boundary mode, rows of contiguous samples (rows that do not start on 16-byte boundaries are parked
// in LDS with scalar loads first).
#include "mifwt_pyr.h"

namespace mifwt {

namespace {

constexpr int kLongThreads = 256;
constexpr int kLongMaxLevels = 8;
constexpr int kLongCapA = 8192;              // most floats of LDS buffer A (level 0, 2, ...); buffer B holds half of it
constexpr int kLongPad = 16;                 // floats of slack behind each buffer (the pair walk may read one pair too far)

template <int L>
struct Dwt1LongArgs {
  const float* x;
  float* approx;                 // level-K approximation [rows, n[K]]
  float* det[kLongMaxLevels];    // per fused level: detail coefficients [rows, n[l + 1]]
  int64_t x_rs, approx_rs, det_rs[kLongMaxLevels];  // row strides (elements)
  int n[kLongMaxLevels + 1];     // n[0] = samples per row, n[l] = coefficients of level l
  int nlevels, mode, rows;
  int chunk, nchunks;            // level-K outputs per interior chunk, interior chunks per row
  int end_l, end_r;              // level-K outputs of the left / right end piece (workgroup `row` < rows)
  int vec;                       // rows start on 16-byte boundaries: 16-byte loads
  int cap;                       // floats of LDS buffer A
  unsigned long long* prof;      // per workgroup 12 cycle stamps (tools/long1d_prof.py), or null
  int dbg;                       // MIFWT_OPT_DEBUG: 1 = no detail stores, 2 = no loads (A/B measurements)
  f2 tap[L];                     // (dec_lo[m], dec_hi[m]), PyWavelets order
};

// Ranges of a piece that owns the level-K outputs [A, B), at level K - sh (closed forms of the top-down recurrences
// a_l = max(0, 2 a_{l+1} - (L - 2)), b_l = min(n_l, 2 b_{l+1}); 2 n_{l+1} >= n_l makes the clipping commute with the doubling):
// computed [a, b) — what the piece holds in LDS — and owned [oa, ob) — what this workgroup stores (a partition of the level).
// Interior chunks (ENDS = false) never clip: the planner keeps them L outputs away from both ends of the row.
template <bool ENDS>
struct LongPiece {
  int A, B;
  bool last;  // B == n[K]
  __device__ __forceinline__ int a(int sh, int HL) const {
    const int v = (A << sh) - HL * ((1 << sh) - 1);
    return ENDS ? max(0, v) : v;
  }
  __device__ __forceinline__ int b(int sh, int nl) const { return ENDS ? min(nl, B << sh) : (B << sh); }
  __device__ __forceinline__ int oa(int sh) const { return A << sh; }
  __device__ __forceinline__ int ob(int sh, int nl) const { return ENDS ? (last ? nl : min(nl, B << sh)) : (B << sh); }
};

// boundary map for rows at least as long as the filter (one fold), as two affine maps chosen once per level: position i < 0
// reads cl + sl i, position i >= n reads cr + sr i; zero mode: -1 = an implicit zero
struct LongExt {
  int n, cl, sl, cr, sr;
  bool zero;
  __device__ __forceinline__ void set(int mode, int n_) {
    n = n_;
    zero = mode == MIFWT_MODE_ZERO;
    switch (mode) {
      case MIFWT_MODE_PERIODIC: cl = n_; sl = 1; cr = -n_; sr = 1; break;
      case MIFWT_MODE_SYMMETRIC: cl = -1; sl = -1; cr = 2 * n_ - 1; sr = -1; break;
      case MIFWT_MODE_REFLECT: cl = 0; sl = -1; cr = 2 * (n_ - 1); sr = -1; break;
      default: cl = 0; sl = 0; cr = n_ - 1; sr = 0; break;  // constant (and zero, masked below)
    }
  }
  __device__ __forceinline__ int operator()(int i) const {
    if ((unsigned)i < (unsigned)n) return i;
    if (zero) return -1;
    return i < 0 ? cl + sl * i : cr + sr * i;
  }
};

// The kernel is bound by instruction issue, scalar instructions included (one scalar unit per CU).  With the range arithmetic
// of the general case in every workgroup the launch took 63 us with 4 waves per workgroup, 88 with 8 and 201 with 16 for the
// same samples: the per-wave cost that does not depend on the samples per lane was 46 of the 63 us.  Hence interior chunks
// (99 % of the workgroups) run a body without pieces, clipping, boundary map or origin arithmetic and with the phase of the
// walk known at compile time; only the workgroups that own the two ends of a row run the general one (with a smaller unroll:
// the whole kernel has to stay well inside the instruction cache — the first version was 80 KB of code).
template <int L, int kLongThreads, bool ENDS>
__device__ __forceinline__ void dwt1_long_body(const Dwt1LongArgs<L>& a, float* bufA, float* bufB, const int row, const LongPiece<ENDS> pc0,
                                               const LongPiece<ENDS> pc1) {
  const int tid = threadIdx.x, K = a.nlevels;
  constexpr int HL = L - 2;
  // output pairs per lane and trip (their windows are requested together); the end-piece body is 32 workgroups of a launch:
  // small code matters more than its speed (the whole kernel has to stay well inside the 64 KB instruction cache)
  constexpr int U = ENDS ? 1 : ((L <= 10 && kLongThreads <= 256) ? 4 : 2);
  constexpr int NP = ENDS ? 2 : 1;
  const bool direct = a.vec != 0;
  // (end pieces) the taps where a lane can index them, and the partial products of the outputs next to an end of the row
  f2* tapl = reinterpret_cast<f2*>(bufB + a.cap / 2 + 64 + kLongPad);
  f2* part = tapl + L;
  if constexpr (ENDS) {
    if (tid < L) tapl[tid] = a.tap[tid];
    __syncthreads();
  }
  int stamp_i = 1;
  auto stamp = [&]() {
    if (MIFWT_PROFP(a) && tid == 0 && stamp_i < 12) MIFWT_PROFP(a)[(size_t)blockIdx.x * 12 + stamp_i++] = __builtin_readcyclecounter();
  };
  int fine_i = 0;
  auto fine = [&](int l) {  // finer stamps of level 2 of the end-piece workgroups, behind the coarse ones (MIFWT_DBG(a) & 8)
    if (ENDS && MIFWT_PROFP(a) && (MIFWT_DBG(a) & 8) && tid == 0 && l == 1 && fine_i < 12) MIFWT_PROFP(a)[(size_t)(gridDim.x + blockIdx.x) * 12 + fine_i++] = __builtin_readcyclecounter();
  };
  auto piece = [&](int p) { return (ENDS && p == 1) ? pc1 : pc0; };  // (no indexed array: it would live in scratch memory)
  // LDS index of a piece's first sample at level K - sh: piece 0 starts at 0, piece 1 follows it
  auto origin = [&](int p, int sh, int nl) {
    if (!ENDS || p == 0) return 0;
    return ((pc0.b(sh, nl) - pc0.a(sh, HL) + 3) & ~3) + 4;
  };
  const float* __restrict__ xr = a.x + (int64_t)row * a.x_rs;
  if (!direct) {  // rows that do not start on 16-byte boundaries: park the level-0 samples of the piece(s)
    for (int p = 0; p < NP; ++p) {
      const LongPiece<ENDS> P = piece(p);
      const int pa = P.a(K, HL), pb = P.b(K, a.n[0]), org = origin(p, K, a.n[0]);
      for (int i = tid; pa + i < pb; i += kLongThreads) bufA[org + i] = (MIFWT_DBG(a) & 2) ? 0.f : xr[pa + i];
    }
    __syncthreads();
  }
  stamp();

  f2 tap[L];
#pragma unroll
  for (int m = 0; m < L; ++m) tap[m] = a.tap[m];
  const rsrc_t xres = pyr_rsrc(xr, (MIFWT_DBG(a) & 2) ? 0u : (uint32_t)a.n[0] * 4u);  // reads past the end of a row return 0

  float* src = bufA;
  float* dst = bufB;
  for (int l = 0; l < ((MIFWT_DBG(a) & 4) ? 0 : K); ++l) {
    const int nl = a.n[l], nk = a.n[l + 1], sh = K - l;
    float* __restrict__ dr = a.det[l] + (int64_t)row * a.det_rs[l];
    const bool glob = direct && l == 0;
#pragma unroll 1
    for (int p = 0; p < NP; ++p) {
      fine(l);
      const LongPiece<ENDS> P = piece(p), Q = piece(1 - p);
      const int sa = P.a(sh, HL), sb = P.b(sh, nl), sorg = origin(p, sh, nl);
      const int ka = P.a(sh - 1, HL), kb = P.b(sh - 1, nk), korg = origin(p, sh - 1, nk);
      const int oa = P.oa(sh - 1), ob = (MIFWT_DBG(a) & 1) ? 0 : P.ob(sh - 1, nk);  // (owned range; empty with the stores switched off)
      // outputs whose L samples all lie inside the piece: [kf0, kf1), two adjacent ones per lane and slot
      const int kf0 = ENDS ? min(kb, max(ka, (sa + HL) >> 1)) : ka, kf1 = ENDS ? max(kf0, min(kb, sb >> 1)) : kb;
      // sample 2 k - (L - 2) of the row sits at wbase[2 k].  A lane's windows start 16 bytes apart: 16-byte LDS reads are
      // conflict-free (8-byte reads at that stride hit every bank four times); the walk starts on a 16-byte boundary or 8
      // bytes behind one (PH)
      const float* wbase = src + sorg - HL - sa;
      float* obase = dst + korg - ka;
      // detail stores go through a buffer resource that spans the owned outputs: everything else drops out of range
      const rsrc_t dres = pyr_rsrc(dr + oa, ob > oa ? (uint32_t)(ob - oa) * 4u : 0u);
      const int klast = kf0 + ((kf1 - 1 - kf0) & ~1);  // the last pair of the walk (lanes past the end re-read its window)
      auto walk = [&](auto ph_tag, auto glob_tag) {
        constexpr int PH = decltype(ph_tag)::value;
        constexpr bool GLOB = decltype(glob_tag)::value;
        constexpr int NQ = (PH + L + 2 + 3) / 4;
        for (int kt0 = kf0; kt0 < kf1; kt0 += 2 * kLongThreads * U) {
          const int kt = kt0 + 2 * tid;
          f4 q[U][NQ];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            if (kt0 + 2 * kLongThreads * u < kf1) {  // (same for every lane)
              const int kw = min(kt + 2 * kLongThreads * u, klast);
              if constexpr (GLOB) {
                const uint32_t voff = 4u * (uint32_t)(2 * kw - HL - PH);
#pragma unroll
                for (int j = 0; j < NQ; ++j) q[u][j] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(xres, voff + 16u * j, 0, 0));
              } else {
                const float* w = wbase + 2 * kw - PH;
#pragma unroll
                for (int j = 0; j < NQ; ++j) q[u][j] = *reinterpret_cast<const f4*>(w + 4 * j);
              }
            }
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            if (kt0 + 2 * kLongThreads * u < kf1) {
              const int k0 = kt + 2 * kLongThreads * u;
              f2 win[L / 2 + 1];
#pragma unroll
              for (int j = 0; j < L / 2 + 1; ++j) {
                const int e = PH + 2 * j;  // float index inside the aligned reads
                win[j] = (e & 3) ? (f2){q[u][e >> 2].z, q[u][e >> 2].w} : (f2){q[u][e >> 2].x, q[u][e >> 2].y};
              }
              // (lo, hi) accumulators of the two outputs: tap pair in an SGPR pair, packed FMAs
              f2 acc0 = vmul_lo(tap[L - 1], win[0]), acc1 = vmul_lo(tap[L - 1], win[1]);
              vfma_hi(acc0, tap[L - 2], win[0]);
              vfma_hi(acc1, tap[L - 2], win[1]);
#pragma unroll
              for (int j = 1; j < L / 2; ++j) {
                vfma_lo(acc0, tap[L - 1 - 2 * j], win[j]);
                vfma_lo(acc1, tap[L - 1 - 2 * j], win[j + 1]);
                vfma_hi(acc0, tap[L - 2 - 2 * j], win[j]);
                vfma_hi(acc1, tap[L - 2 - 2 * j], win[j + 1]);
              }
              const bool v0 = k0 < kf1, v1 = k0 + 1 < kf1;
              if (v0) obase[k0] = acc0.x;
              if (v1) obase[k0 + 1] = acc1.x;
              pyr_store1(acc0.y, dres, (v0 && k0 >= oa) ? 4u * (uint32_t)(k0 - oa) : kPyrOob, 0);
              pyr_store1(acc1.y, dres, (v1 && k0 + 1 >= oa) ? 4u * (uint32_t)(k0 + 1 - oa) : kPyrOob, 0);
            }
          }
        }
      };
      fine(l);
      if constexpr (ENDS) {
        if (glob) {
          if ((2 * kf0 - HL) & 2)
            walk(std::integral_constant<int, 2>{}, std::true_type{});
          else
            walk(std::integral_constant<int, 0>{}, std::true_type{});
        } else {
          if ((sorg - HL - sa + 2 * kf0) & 2)
            walk(std::integral_constant<int, 2>{}, std::false_type{});
          else
            walk(std::integral_constant<int, 0>{}, std::false_type{});
        }
      } else {
        // interior chunks: the phase of the walk is a property of the filter.  LDS levels start at index 0 with
        // a_l = 2 a_{l+1} - (L - 2): phase 0; the global walk starts at sample a_0 = 2^K A - (L - 2)(2^K - 1), congruent to
        // L - 2 modulo 4 for K >= 2 (the planner fuses at least two levels)
        if (glob)
          walk(std::integral_constant<int, (HL & 2)>{}, std::true_type{});
        else
          walk(std::integral_constant<int, 0>{}, std::false_type{});
      }
      fine(l);
      if constexpr (ENDS) {
        // the outputs next to an end of the row, through the boundary index map: one lane per (output, tap) — one lane per
        // output with the L taps unrolled was 3 500 cycles per piece and level, the longest workgroups of a launch
        const int nslow = (kf0 - ka) + (kb - kf1);
        const int oa_sa = Q.a(sh, HL), oorg = origin(1 - p, sh, nl);
        LongExt ext;
        ext.set(a.mode, nl);
        for (int e = tid; e < nslow * L; e += kLongThreads) {
          const int idx = e / L, t = e - idx * L;
          const int k = idx < kf0 - ka ? ka + idx : kf1 + (idx - (kf0 - ka));
          const int q = ext(2 * k - HL + t);
          float v = 0.f;
          if (q >= 0) {
            if (glob)
              v = (MIFWT_DBG(a) & 2) ? 0.f : xr[q];
            else  // periodic: a wrapped sample lives in the other end piece
              v = src[(q >= sa && q < sb) ? sorg + (q - sa) : min(max(oorg + (q - oa_sa), 0), a.cap)];
          }
          const f2 h = tapl[L - 1 - t];
          part[e] = (f2){h.x * v, h.y * v};
        }
        __syncthreads();
        for (int idx = tid; idx < nslow; idx += kLongThreads) {
          const int k = idx < kf0 - ka ? ka + idx : kf1 + (idx - (kf0 - ka));
          f2 acc = {0.f, 0.f};
#pragma unroll
          for (int t = 0; t < L; ++t) acc += part[idx * L + t];
          obase[k] = acc.x;
          if (k >= oa && k < ob) dr[k] = acc.y;
        }
        __syncthreads();  // (the other piece reuses the partial products' slots)
      }
    }
    fine(l);
    __syncthreads();
    fine(l);
    stamp();
    float* tmp = src;
    src = dst;
    dst = tmp;
  }
  float* __restrict__ ar = a.approx + (int64_t)row * a.approx_rs;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const LongPiece<ENDS> P = piece(p);
    const int org = origin(p, 0, a.n[K]);
    for (int k = P.A + tid; k < P.B; k += kLongThreads) ar[k] = src[org + (k - P.A)];
  }
  stamp();
}

template <int L, int kLongThreads>
__global__ void __launch_bounds__(kLongThreads) dwt1_long_kernel(const Dwt1LongArgs<L> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char long_lds[];
  // buffer A holds the levels 0, 2, 4, ..., buffer B the odd ones.  Rows that start on 16-byte boundaries (a.vec) are never
  // parked: level 1 reads its windows straight from global memory (16-byte buffer loads, served by the vector L1 where they
  // overlap), so A only has to hold level 2 — half the LDS per workgroup, twice the workgroups per CU
  float* bufA = reinterpret_cast<float*>(long_lds);
  float* bufB = bufA + (a.vec ? a.cap / 4 + 64 : a.cap) + kLongPad;
  const int K = a.nlevels;
  if (MIFWT_PROFP(a) && threadIdx.x == 0) MIFWT_PROFP(a)[(size_t)blockIdx.x * 12] = __builtin_readcyclecounter();
  // workgroups [0, rows): the two ends of row `blockIdx.x`; then the interior chunks, row-major
  if ((int)blockIdx.x < a.rows) {
    const LongPiece<true> p0 = {0, a.end_l, a.end_l == a.n[K]}, p1 = {a.n[K] - a.end_r, a.n[K], true};
    dwt1_long_body<L, kLongThreads, true>(a, bufA, bufB, (int)blockIdx.x, p0, p1);
  } else {
    const int q = blockIdx.x - a.rows;
    const int row = q / a.nchunks;
    const int A = a.end_l + (q - row * a.nchunks) * a.chunk;
    const LongPiece<false> p0 = {A, min(A + a.chunk, a.n[K] - a.end_r), false};
    dwt1_long_body<L, kLongThreads, false>(a, bufA, bufB, row, p0, p0);
  }
}

struct LongPlan {
  int nlevels, chunk, nchunks, end_l, end_r, cap;
  int n[kLongMaxLevels + 1];
};

// how many levels one launch fuses for rows of n0 samples when `want` are asked for: as many as keep the halo below a
// twelfth of a chunk (at most 8); 0 = not this kernel's case
bool long_plan(int dtype, int L, int mode, int64_t rows, int64_t n0, int want, LongPlan* p) {
  if (g_options[MIFWT_OPT_FORCE_GENERIC] || g_options[MIFWT_OPT_PAIR_MODE] == 2) return false;
  if (dtype != MIFWT_F32 || L < 2 || L > 20 || (L & 1) || want < 1) return false;
  if (mode < 0 || mode > MIFWT_MODE_SYMMETRIC) return false;
  if (rows < 1 || rows > (int64_t(1) << 24) || n0 > (int64_t(1) << 30)) return false;
  // rows one workgroup could hold (mifwt_dwt1_fwd_tail) are still served here when they are at least 4096 samples long: with
  // few rows to occupy the chip (smaller chunks, about one workgroup per CU), and with many rows because the one-workgroup-per-
  // row launch runs its levels as one latency chain on 512 lanes (1024 rows of 16 384 samples, 8 levels: 149 us there, 65 here;
  // 4096 rows of 4096 samples — the two end pieces ARE the row — 120 against 84 us)
  const bool big = n0 > dwt1_tail_max_n(dtype);
  if (!big && n0 < 4096) return false;
  int cap = kLongCapA;
  if (!big) {
    cap = 1024;
    while (cap < kLongCapA && (int64_t)cap * 256 < rows * n0) cap *= 2;
  }
  if (g_options[MIFWT_OPT_ROWS_PER_CHUNK] >= 1024 && g_options[MIFWT_OPT_ROWS_PER_CHUNK] <= kLongCapA) cap = g_options[MIFWT_OPT_ROWS_PER_CHUNK] & ~15;
  p->cap = cap;
  int K = 0;
  int64_t n = n0;
  p->n[0] = (int)n0;
  while (K < want && K < kLongMaxLevels) {
    const int halo = (L - 2) * ((2 << K) - 1);  // of K + 1 levels
    if (halo > cap / 12 && K > 0) break;
    if (halo > cap / 4) break;
    n = (n + L - 1) / 2;
    ++K;
    p->n[K] = (int)n;
  }
  if (K < 2) return false;  // (the interior walk's alignment argument needs two levels; one level is the per-level kernels' job)
  const int halo = (L - 2) * ((1 << K) - 1);
  int chunk = ((cap - halo - 16) >> K) & ~1;
  if (chunk < 2 * L + 2) return false;
  const int nK = p->n[K];
  if (nK < 2 * L + 2) return false;
  p->nlevels = K;
  p->chunk = chunk;
  // the two end pieces share one workgroup's LDS: together at most one chunk (each at least L outputs long)
  int e = chunk / 2 - (halo >> (K + 1)) - 2;
  if (e < L + 1) return false;
  if (2 * e >= nK) {
    p->end_l = nK / 2;
    p->end_r = nK - p->end_l;
    p->nchunks = 0;
  } else {
    p->end_l = p->end_r = e;
    p->nchunks = (nK - 2 * e + chunk - 1) / chunk;
  }
  if (rows * (int64_t)(p->nchunks + 1) > (int64_t(1) << 30)) return false;
  return true;
}

template <int L, int T>
int launch_long(const LongPlan& p, int mode, int64_t rows, const void* x, int64_t x_rs, void* approx, int64_t approx_rs, void* const* details,
                const int64_t* det_rs, const double* lo, const double* hi, hipStream_t stream) {
  Dwt1LongArgs<L> a;
  a.x = static_cast<const float*>(x);
  a.approx = static_cast<float*>(approx);
  a.x_rs = x_rs;
  a.approx_rs = approx_rs;
  for (int l = 0; l < p.nlevels; ++l) {
    a.det[l] = static_cast<float*>(details[l]);
    a.det_rs[l] = det_rs[l];
  }
  for (int l = 0; l <= p.nlevels; ++l) a.n[l] = p.n[l];
  a.nlevels = p.nlevels;
  a.mode = mode;
  a.rows = (int)rows;
  a.chunk = p.chunk;
  a.nchunks = p.nchunks;
  a.end_l = p.end_l;
  a.end_r = p.end_r;
  a.vec = ((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (x_rs & 3) == 0) ? 1 : 0;
  a.cap = p.cap;
  a.prof = g_pyr_prof;
  a.dbg = g_options[MIFWT_OPT_DEBUG];
  for (int m = 0; m < L; ++m) a.tap[m] = (f2){(float)lo[m], (float)hi[m]};
  const size_t lds = (size_t)((a.vec ? p.cap / 4 + 64 : p.cap) + kLongPad + p.cap / 2 + 64 + kLongPad + 2 * (L + (L + 2) * L)) * sizeof(float);
  static DynLdsOnce lds_once;
  if (!lds_once.ensure(reinterpret_cast<const void*>(&dwt1_long_kernel<L, T>), 100 * 1024)) return MIFWT_ERR_LAUNCH;
  const unsigned grid = (unsigned)(rows * (p.nchunks + 1));
  hipLaunchKernelGGL((dwt1_long_kernel<L, T>), dim3(grid), dim3(T), lds, stream, a);
  return hipGetLastError() == hipSuccess ? MIFWT_OK : MIFWT_ERR_LAUNCH;
}


// =============================================================================================================================
// SYNTHESIS: the finest K levels of a 1-D reconstruction in one launch, a chunk of the output row per workgroup (kernel id 18).
// Reference seam: the trailing trips of waverec's level loop (src/ptwt/conv_transform.py:184-199: stack + conv_transpose1d(stride 2)
// + crop per level).  Polyphase form, cropped (the formula of mifwt_dwt1_inv_tail and of the 2-D synthesis kernels per axis):
//     y[2p + r] = sum_{i < L/2} g_lo[L-2-2i+r] a[p+i] + g_hi[L-2-2i+r] d[p+i]        (coefficients past the end read as zero)
// A workgroup owns output samples [x0, x1); step s (coarsest first) consumes coefficients [x0 >> (K-s), ...) — L/2 - 1 more per
// level at the right end.  The approximation of the coarsest fused level is parked in LDS, every step reads its detail band
// straight from global memory (buffer loads: out of range = 0), the running approximation ping-pongs between two LDS
// buffers, the last step stores 16 bytes per lane.  No boundary map on this side: every workgroup runs the same body.
template <int L>
struct Idwt1LongArgs {
  const float* approx;               // [rows, m[0]]: the approximation entering the first fused step
  const float* det[kLongMaxLevels];  // det[s]: detail coefficients of step s (coarsest first) [rows, m[s]]
  float* y;                          // [rows, m[K]]
  int64_t approx_rs, y_rs, det_rs[kLongMaxLevels];
  int m[kLongMaxLevels + 1];         // m[s] = coefficients per row entering step s; m[s + 1] = its (cropped) output length
  int nlevels, rows, chunk, nchunks, cap;  // chunk = output samples per workgroup (a multiple of 4); cap = floats of LDS buffer A
  int vec;                           // output rows start on 16-byte boundaries
  int dvec[kLongMaxLevels];          // detail rows of step s start on 8-byte boundaries: 8-byte loads where the walk starts on an even position
  f2 ga[L / 2], gd[L / 2];           // (g[L-2-2i], g[L-1-2i]) of rec_lo / rec_hi
};

template <int L, int T>
__global__ void __launch_bounds__(T) idwt1_long_kernel(const Idwt1LongArgs<L> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char long_lds[];
  __shared__ int rlo[kLongMaxLevels + 1], rhi[kLongMaxLevels + 1];
  // buffer A (a.cap floats) holds the inputs of the LAST step (half a chunk of outputs + halo), buffer B half of that: the step
  // before; the walk starts in whichever buffer makes the parities come out that way
  float* bufA = reinterpret_cast<float*>(long_lds);
  float* bufB = bufA + a.cap + kLongPad;
  constexpr int HLn = L / 2;
  constexpr int U = L <= 10 ? 2 : 1;        // position pairs per lane and trip
  constexpr int NA = HLn / 2 + 1;           // 8-byte pieces of a lane's window: HLn + 1 coefficients (two positions)
  const int tid = threadIdx.x, K = a.nlevels;
  const int row = blockIdx.x / a.nchunks;
  const int c = blockIdx.x - row * a.nchunks;
  const int x0 = c * a.chunk, x1 = min(a.m[K], x0 + a.chunk);
  if (tid == 0) {
    // coefficient range of every step: [lo >> 1, ((hi - 1) >> 1) + L/2) of the range below it, clipped to the row
    int lo = x0, hi = x1;
    rlo[K] = lo;
    rhi[K] = hi;
    for (int s = K - 1; s >= 0; --s) {
      hi = min(a.m[s], ((hi - 1) >> 1) + HLn);
      lo >>= 1;
      rlo[s] = lo;
      rhi[s] = hi;
    }
  }
  __syncthreads();
  {  // park the approximation of the first step (+ zeros behind it: what windows read past the range)
    const int lo = rlo[0], hi = rhi[0];
    const float* __restrict__ ar = a.approx + (int64_t)row * a.approx_rs;
    float* first = (a.nlevels & 1) ? bufA : bufB;  // step s reads A when nlevels - 1 - s is even
    for (int i = tid; i < hi - lo + HLn + 3; i += T) first[i] = lo + i < hi ? ar[lo + i] : 0.f;
  }
  __syncthreads();
  f2 ga[HLn], gd[HLn];
#pragma unroll
  for (int i = 0; i < HLn; ++i) {
    ga[i] = a.ga[i];
    gd[i] = a.gd[i];
  }
  float* src = (a.nlevels & 1) ? bufA : bufB;
  float* dst = (a.nlevels & 1) ? bufB : bufA;
  float* __restrict__ yr = a.y + (int64_t)row * a.y_rs;
  for (int s = 0; s < K; ++s) {
    const int ilo = rlo[s], olo = rlo[s + 1], ohi = rhi[s + 1];
    const bool last = s == K - 1;
    const rsrc_t dres = pyr_rsrc(a.det[s] + (int64_t)row * a.det_rs[s], (uint32_t)a.m[s] * 4u);
    // a lane takes two adjacent positions p, p + 1 (four outputs 2p .. 2p + 3); positions start at ilo = olo >> 1.  Outputs
    // [ohi, ohi + L/2 + 3) are written as zeros: the next step's windows read them
    const int pend = ((last ? ohi : ohi + HLn + 3) + 1) >> 1;
    // 8-byte detail loads where the rows are aligned and the walk starts on an even position (always so in the last step: x0 is
    // a multiple of 4), else 4-byte loads
    auto walk = [&](auto dv_tag) {
      constexpr bool DV = decltype(dv_tag)::value;
      for (int pt0 = ilo; pt0 < pend; pt0 += 2 * T * U) {
        f2 aw[U][NA], dw[U][NA];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (pt0 + 2 * T * u < pend) {  // (same for every lane)
            const int p = pt0 + 2 * (tid + T * u);
            const float* w = src + ((p < pend ? p : ilo) - ilo);  // (8-byte aligned: positions advance in pairs from ilo)
#pragma unroll
            for (int k = 0; k < NA; ++k) aw[u][k] = *reinterpret_cast<const f2*>(w + 2 * k);
#pragma unroll
            for (int k = 0; k < NA; ++k) {
              if constexpr (DV) {
                dw[u][k] = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(dres, 4u * (uint32_t)(p + 2 * k), 0, 0));
              } else {
                dw[u][k].x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dres, 4u * (uint32_t)(p + 2 * k), 0, 0));
                dw[u][k].y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dres, 4u * (uint32_t)(p + 2 * k + 1), 0, 0));
              }
            }
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (pt0 + 2 * T * u < pend) {
            const int p = pt0 + 2 * (tid + T * u);
            // (even, odd) output of position p (y0) and p + 1 (y1): tap pair in an SGPR pair, packed FMAs; coefficient p + i is
            // the .x / .y half of window piece i / 2
            f2 y0 = vmul_lo(ga[0], aw[u][0]), y1 = vmul_hi(ga[0], aw[u][0]);
            vfma_lo(y0, gd[0], dw[u][0]);
            vfma_hi(y1, gd[0], dw[u][0]);
#pragma unroll
            for (int i = 1; i < HLn; ++i) {
              if (i & 1) {
                vfma_hi(y0, ga[i], aw[u][i / 2]);
                vfma_hi(y0, gd[i], dw[u][i / 2]);
                vfma_lo(y1, ga[i], aw[u][(i + 1) / 2]);
                vfma_lo(y1, gd[i], dw[u][(i + 1) / 2]);
              } else {
                vfma_lo(y0, ga[i], aw[u][i / 2]);
                vfma_lo(y0, gd[i], dw[u][i / 2]);
                vfma_hi(y1, ga[i], aw[u][i / 2]);
                vfma_hi(y1, gd[i], dw[u][i / 2]);
              }
            }
            const int j = 2 * p;
            if (last) {
              if (j + 4 <= ohi && a.vec) {
                *reinterpret_cast<f4*>(yr + j) = (f4){y0.x, y0.y, y1.x, y1.y};
              } else {
                if (j < ohi) yr[j] = y0.x;
                if (j + 1 < ohi) yr[j + 1] = y0.y;
                if (j + 2 < ohi) yr[j + 2] = y1.x;
                if (j + 3 < ohi) yr[j + 3] = y1.y;
              }
            } else if (p < pend) {  // (lanes past the end of the walk must not write: the other buffer follows this one)
              float* o = dst + (j - olo);
              if (j >= olo) o[0] = j < ohi ? y0.x : 0.f;
              o[1] = j + 1 < ohi ? y0.y : 0.f;
              o[2] = j + 2 < ohi ? y1.x : 0.f;
              o[3] = j + 3 < ohi ? y1.y : 0.f;
            }
          }
        }
      }
    };
    if (a.dvec[s] && !(ilo & 1))
      walk(std::true_type{});
    else
      walk(std::false_type{});
    __syncthreads();
    float* tmp = src;
    src = dst;
    dst = tmp;
  }
}

// ---- the same launch for f64 data (the reference's second dtype): plain double FMAs, 16-byte LDS reads, 8-byte detail loads ----
template <int L>
struct Idwt1LongArgsD {
  const double* approx;
  const double* det[kLongMaxLevels];
  double* y;
  int64_t approx_rs, y_rs, det_rs[kLongMaxLevels];
  int m[kLongMaxLevels + 1];
  int nlevels, rows, chunk, nchunks, cap;  // cap = doubles of LDS buffer A
  int vec;                                 // output rows start on 16-byte boundaries
  double glo[L], ghi[L];                   // rec_lo / rec_hi, PyWavelets order
};

typedef double d2 __attribute__((ext_vector_type(2)));

template <int L, int T>
__global__ void __launch_bounds__(T) idwt1_long_kernel_f64(const Idwt1LongArgsD<L> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char long_lds[];
  __shared__ int rlo[kLongMaxLevels + 1], rhi[kLongMaxLevels + 1];
  double* bufA = reinterpret_cast<double*>(long_lds);
  double* bufB = bufA + a.cap + kLongPad;
  constexpr int HLn = L / 2;
  constexpr int NA = HLn / 2 + 1;  // 16-byte pieces of a lane's window: HLn + 1 coefficients (two positions)
  const int tid = threadIdx.x, K = a.nlevels;
  const int row = blockIdx.x / a.nchunks;
  const int c = blockIdx.x - row * a.nchunks;
  const int x0 = c * a.chunk, x1 = min(a.m[K], x0 + a.chunk);
  if (tid == 0) {
    int lo = x0, hi = x1;
    rlo[K] = lo;
    rhi[K] = hi;
    for (int s = K - 1; s >= 0; --s) {
      hi = min(a.m[s], ((hi - 1) >> 1) + HLn);
      lo >>= 1;
      rlo[s] = lo;
      rhi[s] = hi;
    }
  }
  __syncthreads();
  {
    const int lo = rlo[0], hi = rhi[0];
    const double* __restrict__ ar = a.approx + (int64_t)row * a.approx_rs;
    double* first = (a.nlevels & 1) ? bufA : bufB;
    for (int i = tid; i < hi - lo + HLn + 3; i += T) first[i] = lo + i < hi ? ar[lo + i] : 0.0;
  }
  __syncthreads();
  double glo[L], ghi[L];
#pragma unroll
  for (int i = 0; i < L; ++i) {
    glo[i] = a.glo[i];
    ghi[i] = a.ghi[i];
  }
  double* src = (a.nlevels & 1) ? bufA : bufB;
  double* dst = (a.nlevels & 1) ? bufB : bufA;
  double* __restrict__ yr = a.y + (int64_t)row * a.y_rs;
  for (int s = 0; s < K; ++s) {
    const int ilo = rlo[s], olo = rlo[s + 1], ohi = rhi[s + 1];
    const bool last = s == K - 1;
    const rsrc_t dres = pyr_rsrc(a.det[s] + (int64_t)row * a.det_rs[s], (uint32_t)a.m[s] * 8u);
    const int pend = ((last ? ohi : ohi + HLn + 3) + 1) >> 1;
    for (int pt0 = ilo; pt0 < pend; pt0 += 2 * T) {
      const int p = pt0 + 2 * tid;
      const double* w = src + ((p < pend ? p : ilo) - ilo);  // (16-byte aligned: positions advance in pairs from ilo)
      double af[2 * NA], df[HLn + 1];
#pragma unroll
      for (int k = 0; k < NA; ++k) {
        const d2 v = *reinterpret_cast<const d2*>(w + 2 * k);
        af[2 * k] = v.x;
        af[2 * k + 1] = v.y;
      }
#pragma unroll
      for (int k = 0; k < HLn + 1; ++k) df[k] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(dres, 8u * (uint32_t)(p + k), 0, 0));
      double y0e = 0.0, y0o = 0.0, y1e = 0.0, y1o = 0.0;  // outputs 2p, 2p + 1, 2p + 2, 2p + 3: the per-level kernels' summation order
#pragma unroll
      for (int i = 0; i < HLn; ++i) {
        y0e = __builtin_fma(glo[L - 2 - 2 * i], af[i], y0e);
        y0e = __builtin_fma(ghi[L - 2 - 2 * i], df[i], y0e);
        y0o = __builtin_fma(glo[L - 1 - 2 * i], af[i], y0o);
        y0o = __builtin_fma(ghi[L - 1 - 2 * i], df[i], y0o);
        y1e = __builtin_fma(glo[L - 2 - 2 * i], af[i + 1], y1e);
        y1e = __builtin_fma(ghi[L - 2 - 2 * i], df[i + 1], y1e);
        y1o = __builtin_fma(glo[L - 1 - 2 * i], af[i + 1], y1o);
        y1o = __builtin_fma(ghi[L - 1 - 2 * i], df[i + 1], y1o);
      }
      const int j = 2 * p;
      if (last) {
        if (j + 4 <= ohi && a.vec) {
          *reinterpret_cast<d2*>(yr + j) = (d2){y0e, y0o};
          *reinterpret_cast<d2*>(yr + j + 2) = (d2){y1e, y1o};
        } else {
          if (j < ohi) yr[j] = y0e;
          if (j + 1 < ohi) yr[j + 1] = y0o;
          if (j + 2 < ohi) yr[j + 2] = y1e;
          if (j + 3 < ohi) yr[j + 3] = y1o;
        }
      } else if (p < pend) {
        double* o = dst + (j - olo);
        if (j >= olo) o[0] = j < ohi ? y0e : 0.0;
        o[1] = j + 1 < ohi ? y0o : 0.0;
        o[2] = j + 2 < ohi ? y1e : 0.0;
        o[3] = j + 3 < ohi ? y1o : 0.0;
      }
    }
    __syncthreads();
    double* tmp = src;
    src = dst;
    dst = tmp;
  }
}

struct InvLongPlan {
  int nlevels, chunk, nchunks, cap;
};

// how many of the FINEST levels one launch fuses: all `nlevels` given ones if the halo rule allows (the halo of K levels is
// about (L/2 - 1) 2^K output samples: below a twelfth of a chunk), else fewer — the caller runs the coarser ones first
bool inv_long_plan(int dtype, int L, int64_t rows, int nlevels, const int* m, InvLongPlan* p) {
  if (g_options[MIFWT_OPT_FORCE_GENERIC] || g_options[MIFWT_OPT_PAIR_MODE] == 2) return false;
  if ((dtype != MIFWT_F32 && dtype != MIFWT_F64) || L < 2 || L > 20 || (L & 1) || nlevels < 2 || nlevels > kLongMaxLevels || !m) return false;
  if (rows < 1 || rows > (int64_t(1) << 24)) return false;
  for (int s = 0; s < nlevels; ++s) {
    const int64_t full = 2 * (int64_t)m[s] - L + 2;
    if (m[s] < 1 || (m[s + 1] != full && m[s + 1] != full - 1) || m[s + 1] < 1) return false;
  }
  const int n = m[nlevels];
  if (n > (1 << 30)) return false;
  // output rows one workgroup could hold (mifwt_dwt1_inv_tail) are still served here when they are at least 1024 samples long
  // (few rows: smaller chunks, about one workgroup per CU; many rows: 1024 x 16 384 142 -> 48 us, 4096 x 4096 86 -> 44 us,
  // 16 384 x 1024 111 -> 74 us against the one-workgroup-per-row launch)
  const bool big = n > dwt1_tail_max_n(dtype);
  if (!big && n < 1024) return false;
  const int cap_max = dtype == MIFWT_F64 ? kLongCapA / 2 : kLongCapA;  // elements: the same LDS bytes for either type
  int cap = cap_max;
  if (!big) {
    cap = 1024;
    while (cap < cap_max && (int64_t)cap * 256 < rows * n) cap *= 2;
  }
  const int halo = (L / 2) * (1 << nlevels);
  if (halo > cap / 12) return false;
  p->nlevels = nlevels;
  p->cap = cap / 2 + 64;              // buffer A holds the inputs of the LAST step at most: half a chunk of outputs + halo
  p->chunk = (cap - 2 * halo - 64) & ~3;
  if (p->chunk < 4 * L) return false;
  p->nchunks = (n + p->chunk - 1) / p->chunk;
  if (rows * (int64_t)p->nchunks > (int64_t(1) << 30)) return false;
  return true;
}

template <int L>
int launch_inv_long(const InvLongPlan& p, int64_t rows, const int* m, const void* approx, int64_t approx_rs, const void* const* details,
                    const int64_t* det_rs, void* y, int64_t y_rs, const double* lo, const double* hi, hipStream_t stream) {
  Idwt1LongArgs<L> a;
  a.approx = static_cast<const float*>(approx);
  a.y = static_cast<float*>(y);
  a.approx_rs = approx_rs;
  a.y_rs = y_rs;
  for (int s = 0; s < p.nlevels; ++s) {
    a.det[s] = static_cast<const float*>(details[s]);
    a.det_rs[s] = det_rs[s];
  }
  for (int s = 0; s <= p.nlevels; ++s) a.m[s] = m[s];
  a.nlevels = p.nlevels;
  a.rows = (int)rows;
  a.chunk = p.chunk;
  a.nchunks = p.nchunks;
  a.cap = p.cap;
  a.vec = ((reinterpret_cast<uintptr_t>(y) & 15) == 0 && (y_rs & 3) == 0) ? 1 : 0;
  for (int s = 0; s < p.nlevels; ++s) a.dvec[s] = ((reinterpret_cast<uintptr_t>(details[s]) & 7) == 0 && (det_rs[s] & 1) == 0) ? 1 : 0;
  for (int i = 0; i < L / 2; ++i) {
    a.ga[i] = (f2){(float)lo[L - 2 - 2 * i], (float)lo[L - 1 - 2 * i]};
    a.gd[i] = (f2){(float)hi[L - 2 - 2 * i], (float)hi[L - 1 - 2 * i]};
  }
  const size_t lds = (size_t)(p.cap + kLongPad + p.cap / 2 + 64 + kLongPad) * sizeof(float);
  const unsigned grid = (unsigned)(rows * p.nchunks);
  hipLaunchKernelGGL((idwt1_long_kernel<L, kLongThreads>), dim3(grid), dim3(kLongThreads), lds, stream, a);
  return hipGetLastError() == hipSuccess ? MIFWT_OK : MIFWT_ERR_LAUNCH;
}

template <int L>
int launch_inv_long_f64(const InvLongPlan& p, int64_t rows, const int* m, const void* approx, int64_t approx_rs, const void* const* details,
                        const int64_t* det_rs, void* y, int64_t y_rs, const double* lo, const double* hi, hipStream_t stream) {
  Idwt1LongArgsD<L> a;
  a.approx = static_cast<const double*>(approx);
  a.y = static_cast<double*>(y);
  a.approx_rs = approx_rs;
  a.y_rs = y_rs;
  for (int s = 0; s < p.nlevels; ++s) {
    a.det[s] = static_cast<const double*>(details[s]);
    a.det_rs[s] = det_rs[s];
  }
  for (int s = 0; s <= p.nlevels; ++s) a.m[s] = m[s];
  a.nlevels = p.nlevels;
  a.rows = (int)rows;
  a.chunk = p.chunk;
  a.nchunks = p.nchunks;
  a.cap = p.cap;
  a.vec = ((reinterpret_cast<uintptr_t>(y) & 15) == 0 && (y_rs & 1) == 0) ? 1 : 0;
  for (int i = 0; i < L; ++i) {
    a.glo[i] = lo[i];
    a.ghi[i] = hi[i];
  }
  const size_t lds = (size_t)(p.cap + kLongPad + p.cap / 2 + 64 + kLongPad) * sizeof(double);
  static DynLdsOnce lds_once;
  if (!lds_once.ensure(reinterpret_cast<const void*>(&idwt1_long_kernel_f64<L, kLongThreads>), 100 * 1024)) return MIFWT_ERR_LAUNCH;
  const unsigned grid = (unsigned)(rows * p.nchunks);
  hipLaunchKernelGGL((idwt1_long_kernel_f64<L, kLongThreads>), dim3(grid), dim3(kLongThreads), lds, stream, a);
  return hipGetLastError() == hipSuccess ? MIFWT_OK : MIFWT_ERR_LAUNCH;
}

}  // namespace

int dwt1_long_levels(int dtype, int filt_len, int mode, int64_t rows, int64_t n0, int want) {
  LongPlan p;
  return long_plan(dtype, filt_len, mode, rows, n0, want, &p) ? p.nlevels : 0;
}

int dwt1_long(int dtype, int filt_len, int mode, int64_t rows, int64_t n0, int nlevels, const void* x, int64_t x_row_stride, void* approx,
              int64_t approx_row_stride, void* const* details, const int64_t* detail_row_strides, const double* lo, const double* hi,
              hipStream_t stream) {
  LongPlan p;
  if (!long_plan(dtype, filt_len, mode, rows, n0, nlevels, &p) || p.nlevels != nlevels) return MIFWT_ERR_UNSUPPORTED;
#define MIFWT_LONG_CASE(LL) \
  case LL:                  \
    return launch_long<LL, kLongThreads>(p, mode, rows, x, x_row_stride, approx, approx_row_stride, details, detail_row_strides, lo, hi, stream);
  switch (filt_len) {
    MIFWT_LONG_CASE(2)
    MIFWT_LONG_CASE(4)
    MIFWT_LONG_CASE(6)
    MIFWT_LONG_CASE(8)
    MIFWT_LONG_CASE(10)
    MIFWT_LONG_CASE(12)
    MIFWT_LONG_CASE(14)
    MIFWT_LONG_CASE(16)
    MIFWT_LONG_CASE(18)
    MIFWT_LONG_CASE(20)
    default: return MIFWT_ERR_UNSUPPORTED;
  }
#undef MIFWT_LONG_CASE
}

// the launch geometry of both directions, for the host-side model tests (tests/test_long1d_model.py): out[0..5] =
// {levels, chunk, nchunks, end_l, end_r, cap} (analysis) / {levels, chunk, nchunks, 0, 0, cap} (synthesis)
int dwt1_long_plan_query(int dtype, int filt_len, int mode, int64_t rows, int64_t n0, int want, int* out) {
  LongPlan p;
  if (!out || !long_plan(dtype, filt_len, mode, rows, n0, want, &p)) return 0;
  out[0] = p.nlevels; out[1] = p.chunk; out[2] = p.nchunks; out[3] = p.end_l; out[4] = p.end_r; out[5] = p.cap;
  return 1;
}

int idwt1_long_plan_query(int dtype, int filt_len, int64_t rows, int nlevels, const int* m, int* out) {
  InvLongPlan p;
  if (!out || !inv_long_plan(dtype, filt_len, rows, nlevels, m, &p)) return 0;
  out[0] = p.nlevels; out[1] = p.chunk; out[2] = p.nchunks; out[3] = 0; out[4] = 0; out[5] = p.cap;
  return 1;
}

int idwt1_long_supported(int dtype, int filt_len, int64_t rows, int nlevels, const int* m) {
  InvLongPlan p;
  return inv_long_plan(dtype, filt_len, rows, nlevels, m, &p) ? 1 : 0;
}

int idwt1_long(int dtype, int filt_len, int64_t rows, int nlevels, const int* m, const void* approx, int64_t approx_row_stride,
               const void* const* details, const int64_t* detail_row_strides, void* y, int64_t y_row_stride, const double* lo,
               const double* hi, hipStream_t stream) {
  InvLongPlan p;
  if (!inv_long_plan(dtype, filt_len, rows, nlevels, m, &p)) return MIFWT_ERR_UNSUPPORTED;
#define MIFWT_INV_LONG_CASE(LL)                                                                                                          \
  case LL:                                                                                                                               \
    return dtype == MIFWT_F64                                                                                                            \
               ? launch_inv_long_f64<LL>(p, rows, m, approx, approx_row_stride, details, detail_row_strides, y, y_row_stride, lo, hi, stream) \
               : launch_inv_long<LL>(p, rows, m, approx, approx_row_stride, details, detail_row_strides, y, y_row_stride, lo, hi, stream);
  switch (filt_len) {
    MIFWT_INV_LONG_CASE(2)
    MIFWT_INV_LONG_CASE(4)
    MIFWT_INV_LONG_CASE(6)
    MIFWT_INV_LONG_CASE(8)
    MIFWT_INV_LONG_CASE(10)
    MIFWT_INV_LONG_CASE(12)
    MIFWT_INV_LONG_CASE(14)
    MIFWT_INV_LONG_CASE(16)
    MIFWT_INV_LONG_CASE(18)
    MIFWT_INV_LONG_CASE(20)
    default: return MIFWT_ERR_UNSUPPORTED;
  }
#undef MIFWT_INV_LONG_CASE
}

}  // namespace mifwt
