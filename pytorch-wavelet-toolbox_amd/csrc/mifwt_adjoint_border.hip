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
    s_lo[threadIdx.x] = a.lo[threadIdx.x];
    s_hi[threadIdx.x] = a.hi[threadIdx.x];
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
  if (per_image == 0 || d->batch == 0) return MIFWT_OK;
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
