// mifwt_adjoint_border.hip — the boundary part of the ADJOINT of one analysis level with a boundary extension (gfx950).
//
// Seam: the backward of the reference's F.pad + F.conv{1,2,3}d(stride 2) (src/ptwt/conv_transform.py:135-139,
// conv_transform_2.py:142-149, conv_transform_3.py:121-129): ATen's conv backward followed by the pad's backward, which folds the
// gradient of the padded border back into the signal.
//
// An analysis level is  c = A_0 E x :  E extends the signal by (L - 2, L - 2 + N % 2) samples per axis through the boundary index map
// (ext_index, mifwt_common.h), A_0 is the zero-mode filter bank on the extended signal.  Its transpose is  g_x = E^T u  with
//     u[e] = sum_k sum_bands g_band[k] h_band[2 k + 1 - e]            (per axis; e in [-(L - 2), N + L - 2 + N % 2))
// and  (E^T u)[n] = the sum of u over every extended index that maps to n.  For e inside [0, N) u is exactly what the zero-mode adjoint
// computes — the fused SYNTHESIS kernels with the dec taps reversed (mifwt_api.hip, run_inv) — and an interior sample has no other
// preimage.  So the adjoint with ANY boundary mode = that fast launch over the whole signal + this kernel, which ADDS, to the samples
// within B = L - 1 + N % 2 of a border, the terms of their other preimages: one thread per such sample, every pad position that maps
// onto it (mirrored / wrapped / clamped) times every coefficient in reach (the sample's own term is already there).  O(preimages x (L/2)^ndim x 2^ndim) per border sample, a few
// percent of a level's samples: 64 x 1024^2 db4 reflect: 1.18 ms for the generic per-axis adjoint passes -> the synthesis kernel's
// 0.12 ms + this.  f32 (f32 sums) and f64, 1-3 axes, L <= 32, single-fold extents (N >= 2 B per axis); everything else stays on the
// generic passes (launch_axis_adj).
#include "mifwt_common.h"
#include "mifwt_stream.h"

namespace mifwt {

namespace {

constexpr int kMaxTaps = 32;

template <typename T, int ND>
struct BorderArgs {
  const T* gband[1 << ND];  // gradient of band s (bit ND-1-a set <=> axis a high-pass); [0] = the approximation
  T* gx;
  int64_t xs[ND + 1];       // strides of g_x (elements): [0] batch, then the axes
  int64_t as[ND + 1], ds[ND + 1];  // ... of the approximation's / the detail bands' gradients
  int N[ND], M[ND], B[ND], pl[ND], pr[ND];
  int L, mode;
  uint32_t per_image;        // border samples per batch element (a launch covers `images` of them: blockIdx.y)
  int64_t img0;              // first batch element of this launch
  uint32_t slabs[ND], rest_border[ND];     // decode_border: samples in the border hyperplanes of axis d / border samples of a box of the axes after d
  FastDiv dv_full[ND], dv_border[ND], dv_n[ND];  // divisions by (full hyperplane of the axes after d), rest_border[d], N[d]
  T lo[kMaxTaps], hi[kMaxTaps];
  DevTapArg dt;  // device-resident taps (mifwt_common.h); dt.lo == nullptr: lo / hi count
};

// the e-th border sample of a box of extents N with border widths B: coordinates n.  Samples are enumerated slab by slab: first the
// 2 B[d] border hyperplanes of axis d (full extent of the axes after it), then, for every interior position of axis d, the border
// samples of the remaining axes.  32-bit indices, divisions by launch-time constants (FastDiv: the 64-bit runtime divisions of the
// first version cost more than the sums once 16 lanes shared a sample).
template <typename T, int ND>
__device__ __forceinline__ void decode_border(uint32_t e, const BorderArgs<T, ND>& a, int* n) {
#pragma unroll
  for (int d = 0; d < ND; ++d) {
    if (e < a.slabs[d] || d == ND - 1) {
      uint32_t r;
      const uint32_t hb = a.dv_full[d].divmod(e, r);  // which border hyperplane of axis d, position inside it
      n[d] = (int)hb < a.B[d] ? (int)hb : a.N[d] - 2 * a.B[d] + (int)hb;
#pragma unroll
      for (int q = ND - 1; q > d; --q) {
        uint32_t rem;
        r = a.dv_n[q].divmod(r, rem);
        n[q] = (int)rem;
      }
      return;
    }
    e -= a.slabs[d];
    uint32_t rem;
    const uint32_t pos = a.dv_border[d].divmod(e, rem);
    n[d] = a.B[d] + (int)pos;
    e = rem;
  }
}

// the extended indices that map to sample n of an axis, as up to three ranges [a, b]: itself, pad positions below 0, pad positions
// from N on (empty: a > b)
__device__ __forceinline__ void preimages(int n, int N, int pl, int pr, int mode, int (&a)[3], int (&b)[3]) {
  a[0] = b[0] = n;
  a[1] = a[2] = 0;
  b[1] = b[2] = -1;
  int j;
  switch (mode) {
    case MIFWT_MODE_CONSTANT:
      if (n == 0) a[1] = -pl, b[1] = -1;
      if (n == N - 1) a[2] = N, b[2] = N + pr - 1;
      break;
    case MIFWT_MODE_PERIODIC:
      j = n - N;  // a pad position below 0
      if (j >= -pl) a[1] = b[1] = j;
      j = n + N;
      if (j < N + pr) a[2] = b[2] = j;
      break;
    case MIFWT_MODE_SYMMETRIC:
      j = -1 - n;
      if (j >= -pl) a[1] = b[1] = j;
      j = 2 * N - 1 - n;
      if (j < N + pr) a[2] = b[2] = j;
      break;
    case MIFWT_MODE_REFLECT:
      j = -n;
      if (n >= 1 && j >= -pl) a[1] = b[1] = j;
      j = 2 * (N - 1) - n;
      if (n <= N - 2 && j < N + pr) a[2] = b[2] = j;
      break;
    default: break;  // zero: no pad position carries a gradient
  }
}

// Lanes per border sample (1, 4 or 16): with more than one the lanes split the coefficient positions in reach (per axis: lane part i_d
// takes k = k_lo + i_d, k_lo + i_d + S_d, ...) and their partial sums meet in a shuffle tree.  MEASURED on config 2 (reflect, level 1 /
// 2 / 3 of the backward): one lane per sample 85 / 50 / 30 us, sixteen 215 / 105 / 55 us — the kernel is bound by its instruction
// count (~25 instructions of loop and address arithmetic per 4 loads), not by latency, and sixteen lanes run the loop nests sixteen
// times.  One lane per sample it is.
#ifndef MIFWT_BORDER_LANES
#define MIFWT_BORDER_LANES 1
#endif
constexpr int kLanesPerSample = MIFWT_BORDER_LANES;
static_assert(kLanesPerSample == 1 || kLanesPerSample == 4 || kLanesPerSample == 16, "lanes per border sample");
constexpr int kS1 = kLanesPerSample;                                                             // 1 axis: all lanes along it
constexpr int kS2a = kLanesPerSample == 16 ? 4 : (kLanesPerSample == 4 ? 2 : 1), kS2b = kS2a;   // 2 axes
constexpr int kS3a = kLanesPerSample == 16 ? 2 : 1, kS3b = kLanesPerSample >= 4 ? 2 : 1, kS3c = kLanesPerSample / (kS3a * kS3b);  // 3 axes

template <typename T, int ND>
__global__ void __launch_bounds__(256) adjoint_border_kernel(const BorderArgs<T, ND> a) {
  // the taps are indexed per lane: from LDS (from the kernel arguments every such read is a memory request)
  __shared__ T s_lo[kMaxTaps], s_hi[kMaxTaps];
  if (threadIdx.x < kMaxTaps) {
    const int m = (int)threadIdx.x;
    if (a.dt.lo) {  // (a learnable filter bank that lives on the GPU)
      const int mm = a.dt.rev ? a.L - 1 - m : m;
      s_lo[m] = m < a.L ? (T)a.dt.lo[mm] : (T)0;
      s_hi[m] = m < a.L ? (T)a.dt.hi[mm] : (T)0;
    } else {
      s_lo[m] = a.lo[m];
      s_hi[m] = a.hi[m];
    }
  }
  __syncthreads();
  const uint32_t e = (blockIdx.x * 256u + threadIdx.x) / kLanesPerSample;
  const int sub = threadIdx.x & (kLanesPerSample - 1);
  if (e >= a.per_image) return;  // (whole 16-lane groups leave together)
  const int64_t img = a.img0 + blockIdx.y;
  int n[ND];
  decode_border<T, ND>(e, a, n);
  int ra[ND][3], rb[ND][3];
#pragma unroll
  for (int d = 0; d < ND; ++d) preimages(n[d], a.N[d], a.pl[d], a.pr[d], a.mode, ra[d], rb[d]);
  const int L = a.L;
  T acc = 0;
  if constexpr (ND == 1) {
    for (int q0 = 1; q0 < 3; ++q0)  // (q0 = 0 is the sample itself: that term is what the zero-mode adjoint has written)
      for (int e0 = ra[0][q0]; e0 <= rb[0][q0]; ++e0) {
        const int k_lo = max(0, e0 >> 1), k_hi = min(a.M[0] - 1, (e0 + L - 2) >> 1);  // 0 <= 2 k + 1 - e0 < L
        for (int k0 = k_lo + sub; k0 <= k_hi; k0 += kS1) {
          const int m0 = 2 * k0 + 1 - e0;
          const int64_t oa = img * a.as[0] + k0 * a.as[1], od = img * a.ds[0] + k0 * a.ds[1];
          acc += a.gband[0][oa] * s_lo[m0] + a.gband[1][od] * s_hi[m0];
        }
      }
  } else if constexpr (ND == 2) {
    const int i0 = sub / kS2b, i1 = sub % kS2b;
    for (int q0 = 0; q0 < 3; ++q0)
      for (int e0 = ra[0][q0]; e0 <= rb[0][q0]; ++e0) {
        const int k0_lo = max(0, e0 >> 1), k0_hi = min(a.M[0] - 1, (e0 + L - 2) >> 1);
        for (int k0 = k0_lo + i0; k0 <= k0_hi; k0 += kS2a) {
          const int m0 = 2 * k0 + 1 - e0;
          const T l0 = s_lo[m0], h0 = s_hi[m0];
          for (int q1 = (q0 == 0 ? 1 : 0); q1 < 3; ++q1)  // (q0 = q1 = 0: the sample itself, already in g_x)
            for (int e1 = ra[1][q1]; e1 <= rb[1][q1]; ++e1) {
              const int k1_lo = max(0, e1 >> 1), k1_hi = min(a.M[1] - 1, (e1 + L - 2) >> 1);
              // (requesting four columns' sixteen loads before the first use changed nothing: 46 / 85 / 218 us per level of config 2's
              // backward either way — the kernel is bound by its instruction count, ~43 us per million border samples)
              for (int k1 = k1_lo + i1; k1 <= k1_hi; k1 += kS2b) {
                const int m1 = 2 * k1 + 1 - e1;
                const T l1 = s_lo[m1], h1 = s_hi[m1];
                const int64_t oa = img * a.as[0] + k0 * a.as[1] + k1 * a.as[2], od = img * a.ds[0] + k0 * a.ds[1] + k1 * a.ds[2];
                acc += l0 * (a.gband[0][oa] * l1 + a.gband[1][od] * h1) + h0 * (a.gband[2][od] * l1 + a.gband[3][od] * h1);
              }
            }
        }
      }
  } else {
    const int i0 = sub / (kS3b * kS3c), i1 = (sub / kS3c) % kS3b, i2 = sub % kS3c;
    for (int q0 = 0; q0 < 3; ++q0)
      for (int e0 = ra[0][q0]; e0 <= rb[0][q0]; ++e0) {
        const int k0_lo = max(0, e0 >> 1), k0_hi = min(a.M[0] - 1, (e0 + L - 2) >> 1);
        for (int k0 = k0_lo + i0; k0 <= k0_hi; k0 += kS3a) {
          const int m0 = 2 * k0 + 1 - e0;
          const T l0 = s_lo[m0], h0 = s_hi[m0];
          for (int q1 = 0; q1 < 3; ++q1)
            for (int e1 = ra[1][q1]; e1 <= rb[1][q1]; ++e1) {
              const int k1_lo = max(0, e1 >> 1), k1_hi = min(a.M[1] - 1, (e1 + L - 2) >> 1);
              for (int k1 = k1_lo + i1; k1 <= k1_hi; k1 += kS3b) {
                const int m1 = 2 * k1 + 1 - e1;
                const T l1 = s_lo[m1], h1 = s_hi[m1];
                const T w00 = l0 * l1, w01 = l0 * h1, w10 = h0 * l1, w11 = h0 * h1;  // (axis 0, axis 1) = (lo, lo), (lo, hi), ...
                for (int q2 = (q0 == 0 && q1 == 0 ? 1 : 0); q2 < 3; ++q2)  // (all three zero: the sample itself, already in g_x)
                  for (int e2 = ra[2][q2]; e2 <= rb[2][q2]; ++e2) {
                    const int k2_lo = max(0, e2 >> 1), k2_hi = min(a.M[2] - 1, (e2 + L - 2) >> 1);
                    for (int k2 = k2_lo + i2; k2 <= k2_hi; k2 += kS3c) {
                      const int m2 = 2 * k2 + 1 - e2;
                      const T l2 = s_lo[m2], h2 = s_hi[m2];
                      const int64_t oa = img * a.as[0] + k0 * a.as[1] + k1 * a.as[2] + k2 * a.as[3];
                      const int64_t od = img * a.ds[0] + k0 * a.ds[1] + k1 * a.ds[2] + k2 * a.ds[3];
                      acc += w00 * (a.gband[0][oa] * l2 + a.gband[1][od] * h2) + w01 * (a.gband[2][od] * l2 + a.gband[3][od] * h2) +
                             w10 * (a.gband[4][od] * l2 + a.gband[5][od] * h2) + w11 * (a.gband[6][od] * l2 + a.gband[7][od] * h2);
                    }
                  }
              }
            }
        }
      }
  }
#pragma unroll
  for (int m = kLanesPerSample / 2; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
  if (sub != 0) return;
  int64_t ox = img * a.xs[0];
#pragma unroll
  for (int d = 0; d < ND; ++d) ox += (int64_t)n[d] * a.xs[1 + d];
  a.gx[ox] += acc;  // (on top of the sample's own term, which the zero-mode adjoint launch has written)
}

// ---- two axes, one thread per border LINE ---------------------------------------------------------------------------------------------
// The kernel above spends ~25 instructions of loop and address arithmetic per 4 loads and shares nothing between the samples of a border:
// config 2's backward (64 x 1024^2 db4 reflect) paid 56 + 37 + 21 us for the three levels' borders next to 145 us of synthesis launches.
// The extended adjoint is separable, u[e0, e1] = sum_k0 h[2 k0 + 1 - e0] X[k0][e1]: here a thread owns
//   * one COLUMN n1 (N1 threads) and every border row of it, top and bottom: for each preimage e1 of n1 it synthesises the few coefficient
//     rows within reach of the two row frames along its column ONCE (X_lo, X_hi per row, kept in LDS), then folds them into the 2 B0
//     border rows (every preimage pair but (n0, n1) itself);  or
//   * one interior ROW n0 (N0 - 2 B0 threads) and its left / right border columns: the same with the axes exchanged — an interior row has
//     no other preimage, so only the pad columns are synthesised.
// Rows inside the row slabs belong to the column threads, the rest of the column slabs to the row threads: every border sample has one
// owner, one launch, no atomics.  ~750 instructions per thread, ~2 N threads per image instead of ~450 x 4 B N.
template <typename T>
struct BPair {
  T x, y;
};

constexpr int kBorder2Threads = 128;

template <typename T, int G>
__global__ void __launch_bounds__(kBorder2Threads) adjoint_border2_kernel(const BorderArgs<T, 2> a, int krmax) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* const s_lo = reinterpret_cast<T*>(smem_raw);
  T* const s_hi = s_lo + kMaxTaps;
  // G lanes of a wave share a line: each synthesises every G-th coefficient row of the frames, then folds every G-th border sample
  const int tid = (int)threadIdx.x, sub = tid % G;
  BPair<T>* const X = reinterpret_cast<BPair<T>*>(s_hi + kMaxTaps) + (tid / G) * (2 * krmax);  // [frame][row of the frame] of this line
  if (tid < kMaxTaps) {
    if (a.dt.lo) {  // (a learnable filter bank that lives on the GPU)
      const int mm = a.dt.rev ? a.L - 1 - tid : tid;
      s_lo[tid] = tid < a.L ? (T)a.dt.lo[mm] : (T)0;
      s_hi[tid] = tid < a.L ? (T)a.dt.hi[mm] : (T)0;
    } else {
      s_lo[tid] = a.lo[tid];
      s_hi[tid] = a.hi[tid];
    }
  }
  __syncthreads();
  const int N0 = a.N[0], N1 = a.N[1];
  const int inner_rows = max(0, N0 - 2 * a.B[0]);
  const int line = (int)blockIdx.x * (kBorder2Threads / G) + tid / G;
  if (line >= N1 + inner_rows) return;  // (the G lanes of a line leave together)
  const bool col = line < N1;          // a column thread (border axis 0) or a row thread (border axis 1)
  const int b = col ? 0 : 1, o = 1 - b;
  const int c = col ? line : a.B[0] + (line - N1);  // the thread's coordinate along its own axis o
  const int Nb = a.N[b], Mb = a.M[b], Bb = a.B[b], plb = a.pl[b], prb = a.pr[b], Mo = a.M[o];
  const int L = a.L;
  const int64_t img = a.img0 + blockIdx.y;
  // bands by (high along b, high along o): band index bit 1 = axis 0 high, bit 0 = axis 1 high
  const T* const pLL = a.gband[0] + img * a.as[0];
  const T* const pLH = a.gband[col ? 1 : 2] + img * a.ds[0];  // low along b, high along o
  const T* const pHL = a.gband[col ? 2 : 1] + img * a.ds[0];
  const T* const pHH = a.gband[3] + img * a.ds[0];
  const int as_b = (int)a.as[1 + b], as_o = (int)a.as[1 + o], ds_b = (int)a.ds[1 + b], ds_o = (int)a.ds[1 + o];
  T* const gx = a.gx + img * a.xs[0] + (int64_t)c * a.xs[1 + o];
  const int64_t xs_b = a.xs[1 + b];
  int oa[3], ob[3];
  if (col) {
    preimages(c, a.N[o], a.pl[o], a.pr[o], a.mode, oa, ob);
  } else {  // an interior row is its own only preimage
    oa[0] = ob[0] = c;
    oa[1] = oa[2] = 0;
    ob[1] = ob[2] = -1;
  }
  for (int qo = 0; qo < 3; ++qo)
    for (int eo = oa[qo]; eo <= ob[qo]; ++eo) {
      const bool self_o = qo == 0;
      const int ko_lo = max(0, eo >> 1), ko_hi = min(Mo - 1, (eo + L - 2) >> 1);
      // the two frames of extended positions along b whose u the border needs: pad positions, and (unless eo is the thread's own
      // coordinate: that pair is the sample itself) the slab's own rows
      int klo[2], khi[2];
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        const int e_first = f == 0 ? -plb : (self_o ? Nb : Nb - Bb);
        const int e_last = f == 0 ? (self_o ? -1 : Bb - 1) : Nb + prb - 1;
        klo[f] = max(0, e_first >> 1);
        khi[f] = e_last < e_first ? klo[f] - 1 : min(Mb - 1, (e_last + L - 2) >> 1);
        for (int kb = klo[f] + sub; kb <= khi[f]; kb += G) {
          T xl = 0, xh = 0;
          // (all requests of four coefficients before the first use: positions beyond the window are clamped and weighted zero)
          for (int k4 = ko_lo; k4 <= ko_hi; k4 += 4) {
            T va[4], vb[4], vc[4], vd[4], wl[4], wh[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int ko = min(k4 + j, ko_hi);
              const int offa = kb * as_b + ko * as_o, offd = kb * ds_b + ko * ds_o;
              va[j] = pLL[offa];
              vb[j] = pLH[offd];
              vc[j] = pHL[offd];
              vd[j] = pHH[offd];
              const int m = 2 * ko + 1 - eo;
              const bool on = k4 + j <= ko_hi;
              wl[j] = on ? s_lo[m] : T(0);
              wh[j] = on ? s_hi[m] : T(0);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              xl += va[j] * wl[j] + vb[j] * wh[j];
              xh += vc[j] * wl[j] + vd[j] * wh[j];
            }
          }
          X[f * krmax + (kb - klo[f])] = BPair<T>{xl, xh};
        }
      }
      wave_lds_fence();  // (the lanes of a line sit in one wave: DS operations of a wave execute in order)
      {
        for (int di = sub; di < 2 * Bb; di += G) {
          const int t = di < Bb ? di : Nb - 2 * Bb + di;
          int ta[3], tb[3];
          preimages(t, Nb, plb, prb, a.mode, ta, tb);
          T sum = 0;
          bool any = false;
          for (int q = self_o ? 1 : 0; q < 3; ++q)
            for (int eb = ta[q]; eb <= tb[q]; ++eb) {
              const int f = eb < Bb ? 0 : 1;
              const int kl = max(0, eb >> 1), kh = min(Mb - 1, (eb + L - 2) >> 1);
              for (int kb = kl; kb <= kh; ++kb) {
                const int m = 2 * kb + 1 - eb;
                const BPair<T> x = X[f * krmax + (kb - klo[f])];
                sum += s_lo[m] * x.x + s_hi[m] * x.y;
              }
              any = true;
            }
          if (any) gx[(int64_t)t * xs_b] += sum;  // (on top of what the zero-mode adjoint launch and earlier preimages of the column have written)
        }
      }
      wave_lds_fence();
    }
}

// rows of a frame within reach: the frame spans at most pl + B positions, its coefficients (span + L - 2) / 2 + 1
inline int border2_krmax(int L) { return (3 * L) / 2 + 2; }

template <typename T>
bool border2_fits(const mifwt_level_desc* d, int* threads, size_t* lds) {
  // from eight taps on (border part of a level's adjoint on 64 x 1024^2 / 515^2 / 261^2, us, this kernel against one thread per sample:
  // db2 27 / 16 / 10 against 24 / 10 / 9; db3 34 / 21 / 14 against 33 / 21 / 14; db4 36 / 22 / 15 against 52 / 37 / 21; db8 58 / 36 / 26
  // against 228 / 148 / 99; profiles/r05x_border_ab.txt)
  if (d->ndim != 2 || d->filt_len < 8 || (g_options[MIFWT_OPT_DEBUG] & 4096)) return false;
  for (int i = 0; i <= 2; ++i)
    if (d->approx_stride[i] < 0 || d->detail_stride[i] < 0) return false;
  // 32-bit element offsets inside one image of a band
  if ((d->coef_extent[0] - 1) * d->approx_stride[1] + (d->coef_extent[1] - 1) * d->approx_stride[2] >= (int64_t(1) << 31) ||
      (d->coef_extent[0] - 1) * d->detail_stride[1] + (d->coef_extent[1] - 1) * d->detail_stride[2] >= (int64_t(1) << 31))
    return false;
  const int kr = border2_krmax(d->filt_len);
  *threads = kBorder2Threads;
  *lds = 2 * kMaxTaps * sizeof(T) + (size_t)2 * kr * (kBorder2Threads / 8) * sizeof(BPair<T>);
  return true;
}

template <typename T, int ND>
int launch_border(const mifwt_level_desc* d, const void* g_approx, const void* const* g_details, void* g_x, const double* lo,
                  const double* hi, hipStream_t stream) {
  BorderArgs<T, ND> a;
  a.gband[0] = static_cast<const T*>(g_approx);
  for (int s = 1; s < (1 << ND); ++s) a.gband[s] = static_cast<const T*>(g_details[s - 1]);
  a.gx = static_cast<T*>(g_x);
  int64_t box = 1, inner = 1;
  for (int i = 0; i <= ND; ++i) {
    a.xs[i] = d->sig_stride[i];
    a.as[i] = d->approx_stride[i];
    a.ds[i] = d->detail_stride[i];
  }
  for (int i = 0; i < ND; ++i) {
    a.N[i] = (int)d->sig_extent[i];
    a.M[i] = (int)d->coef_extent[i];
    a.pl[i] = d->filt_len - 2;
    a.pr[i] = d->filt_len - 2 + (a.N[i] & 1);
    a.B[i] = a.pr[i] + 1;
    box *= a.N[i];
    inner *= a.N[i] - 2 * a.B[i];
  }
  a.L = d->filt_len;
  a.mode = d->mode;
  const int64_t per_image = box - inner;
  if (per_image * kLanesPerSample >= (int64_t(1) << 31)) return MIFWT_ERR_UNSUPPORTED;
  a.per_image = (uint32_t)per_image;
  for (int i = 0; i < ND; ++i) {
    int64_t rest_full = 1, rest_inner = 1;
    for (int q = i + 1; q < ND; ++q) {
      rest_full *= a.N[q];
      rest_inner *= a.N[q] - 2 * a.B[q];
    }
    a.slabs[i] = (uint32_t)(2 * (int64_t)a.B[i] * rest_full);
    a.rest_border[i] = (uint32_t)(rest_full - rest_inner);
    a.dv_full[i] = make_fastdiv((uint32_t)rest_full);
    a.dv_border[i] = make_fastdiv(a.rest_border[i] ? a.rest_border[i] : 1u);
    a.dv_n[i] = make_fastdiv((uint32_t)a.N[i]);
  }
  for (int m = 0; m < kMaxTaps; ++m) {
    a.lo[m] = m < a.L ? (T)lo[m] : (T)0;
    a.hi[m] = m < a.L ? (T)hi[m] : (T)0;
  }
  a.dt = dev_tap_arg(a.L);
  if (per_image == 0 || d->batch == 0) return MIFWT_OK;
  if constexpr (ND == 2) {
    int threads = 0;
    size_t lds = 0;
    if (border2_fits<T>(d, &threads, &lds)) {
      const int64_t lines = a.N[1] + std::max(0, a.N[0] - 2 * a.B[0]);
      constexpr int G = 8;  // lanes per line (64 x 1024^2 / 515^2 / 261^2 db4, border part alone: 4 lanes 41 / 22 / 17 us, 8: 36 / 22 / 15, 16: 50 / 24 / 22)
      const int lpb = threads / G;
      const unsigned gl = (unsigned)((lines + lpb - 1) / lpb);
      for (int64_t b0 = 0; b0 < d->batch; b0 += 32768) {
        a.img0 = b0;
        const unsigned gy = (unsigned)std::min<int64_t>(32768, d->batch - b0);
        hipLaunchKernelGGL((adjoint_border2_kernel<T, G>), dim3(gl, gy), dim3(threads), lds, stream, a, border2_krmax(a.L));
      }
      return hipGetLastError() == hipSuccess ? MIFWT_OK : MIFWT_ERR_LAUNCH;
    }
  }
  const unsigned gx = (unsigned)((per_image * kLanesPerSample + 255) / 256);
  for (int64_t b0 = 0; b0 < d->batch; b0 += 32768) {  // (grid.y is 16 bits)
    a.img0 = b0;
    const unsigned gy = (unsigned)std::min<int64_t>(32768, d->batch - b0);
    hipLaunchKernelGGL((adjoint_border_kernel<T, ND>), dim3(gx, gy), dim3(256), 0, stream, a);
  }
  return hipGetLastError() == hipSuccess ? MIFWT_OK : MIFWT_ERR_LAUNCH;
}

}  // namespace

// the fast route of the analysis adjoint applies: a boundary extension, f32 / f64, <= 32 taps, every axis long enough for a single fold
// and two disjoint borders
bool adjoint_border_supported(const mifwt_level_desc* d) {
  if (g_options[MIFWT_OPT_FORCE_GENERIC] || (g_options[MIFWT_OPT_DEBUG] & 1024)) return false;
  if (d->mode == MIFWT_MODE_ZERO || d->ndim < 1 || d->ndim > 3) return false;
  if (d->dtype != MIFWT_F32 && d->dtype != MIFWT_F64) return false;
  if (d->filt_len > kMaxTaps || d->filt_len < 2 || (d->filt_len & 1)) return false;
  int64_t box = 1, inner = 1;
  for (int i = 0; i < d->ndim; ++i) {
    const int64_t n = d->sig_extent[i];
    if (n < 2 * (d->filt_len + 1) || n > INT32_MAX / 4) return false;
    box *= n;
    inner *= n - 2 * (d->filt_len - 1 + (n & 1));
    if (box >= (int64_t(1) << 40)) return false;
  }
  return (box - inner) * kLanesPerSample < (int64_t(1) << 31);  // 32-bit sample indices inside one batch element
}

int adjoint_border(const mifwt_level_desc* d, const void* g_approx, const void* const* g_details, void* g_x, const double* lo,
                   const double* hi, hipStream_t stream) {
  if (!adjoint_border_supported(d)) return MIFWT_ERR_UNSUPPORTED;
  const bool f64 = d->dtype == MIFWT_F64;
  switch (d->ndim) {
    case 1: return f64 ? launch_border<double, 1>(d, g_approx, g_details, g_x, lo, hi, stream) : launch_border<float, 1>(d, g_approx, g_details, g_x, lo, hi, stream);
    case 2: return f64 ? launch_border<double, 2>(d, g_approx, g_details, g_x, lo, hi, stream) : launch_border<float, 2>(d, g_approx, g_details, g_x, lo, hi, stream);
    default: return f64 ? launch_border<double, 3>(d, g_approx, g_details, g_x, lo, hi, stream) : launch_border<float, 3>(d, g_approx, g_details, g_x, lo, hi, stream);
  }
}

}  // namespace mifwt
