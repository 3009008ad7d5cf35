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
// preimage.  So the adjoint with ANY boundary mode = that fast launch over the whole signal + this kernel, which recomputes, from
// scratch, the samples within B = L - 1 + N % 2 of a border: one thread per such sample, every preimage of it (itself, mirrored /
// wrapped / clamped pad positions) times every coefficient in reach.  O(preimages x (L/2)^ndim x 2^ndim) per border sample, a few
// percent of a level's samples: 64 x 1024^2 db4 reflect: 1.18 ms for the generic per-axis adjoint passes -> the synthesis kernel's
// 0.12 ms + this.  f32 (f32 sums) and f64, 1-3 axes, L <= 32, single-fold extents (N >= 2 B per axis); everything else stays on the
// generic passes (launch_axis_adj).
#include "mifwt_common.h"

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
  int64_t per_image, total;  // border samples per batch element / in all
  T lo[kMaxTaps], hi[kMaxTaps];
};

// the e-th border sample of a box of extents N[d..ND) with border widths B: coordinates n[d..ND).  Samples are enumerated
// slab by slab: first the 2 B[d] border hyperplanes of axis d (full extent of the other axes), then, for every interior position of
// axis d, the border samples of the remaining axes.
template <int ND>
__device__ __forceinline__ void decode_border(int64_t e, const int* N, const int* B, int* n) {
  int64_t rest_full = 1;  // samples of a full hyperplane of the axes after d
#pragma unroll
  for (int d = 0; d < ND; ++d) {
    rest_full = 1;
#pragma unroll
    for (int q = d + 1; q < ND; ++q) rest_full *= N[q];
    const int64_t slabs = 2 * (int64_t)B[d] * rest_full;
    if (e < slabs || d == ND - 1) {
      // inside a border hyperplane of axis d: everything after d is a plain mixed-radix index
      const int64_t hb = e / rest_full;
      int64_t r = e - hb * rest_full;
      n[d] = hb < B[d] ? (int)hb : N[d] - 2 * B[d] + (int)hb;
#pragma unroll
      for (int q = ND - 1; q > d; --q) {
        n[q] = (int)(r % N[q]);
        r /= N[q];
      }
      return;
    }
    e -= slabs;
    // interior position of axis d, then recurse into the remaining axes
    int64_t rest_border = 1, rest_inner = 1;
#pragma unroll
    for (int q = d + 1; q < ND; ++q) {
      rest_border *= N[q];
      rest_inner *= N[q] - 2 * B[q];
    }
    rest_border -= rest_inner;  // border samples of a box of the remaining axes
    const int64_t pos = e / rest_border;
    n[d] = B[d] + (int)pos;
    e -= pos * rest_border;
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

template <typename T, int ND>
__global__ void __launch_bounds__(256) adjoint_border_kernel(const BorderArgs<T, ND> a) {
  // the taps are indexed per lane: from LDS (from the kernel arguments every such read is a memory request)
  __shared__ T s_lo[kMaxTaps], s_hi[kMaxTaps];
  if (threadIdx.x < kMaxTaps) {
    s_lo[threadIdx.x] = a.lo[threadIdx.x];
    s_hi[threadIdx.x] = a.hi[threadIdx.x];
  }
  __syncthreads();
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= a.total) return;
  const int64_t img = t / a.per_image;
  int n[ND];
  decode_border<ND>(t - img * a.per_image, a.N, a.B, n);
  int ra[ND][3], rb[ND][3];
#pragma unroll
  for (int d = 0; d < ND; ++d) preimages(n[d], a.N[d], a.pl[d], a.pr[d], a.mode, ra[d], rb[d]);
  const int L = a.L;
  T acc = 0;
  // nested loops over (range, extended index, coefficient index) per axis; the innermost axis is ND - 1
  if constexpr (ND == 1) {
    for (int q0 = 0; q0 < 3; ++q0)
      for (int e0 = ra[0][q0]; e0 <= rb[0][q0]; ++e0) {
        const int k_lo = max(0, (e0 - 1 + 1) >> 1), k_hi = min(a.M[0] - 1, (e0 + L - 2) >> 1);  // (e0 - 1) / 2 rounded up; e0 + L - 2 >= 0
        for (int k0 = k_lo; k0 <= k_hi; ++k0) {
          const int m0 = 2 * k0 + 1 - e0;
          if ((unsigned)m0 >= (unsigned)L) continue;
          const int64_t oa = img * a.as[0] + k0 * a.as[1], od = img * a.ds[0] + k0 * a.ds[1];
          acc += a.gband[0][oa] * s_lo[m0] + a.gband[1][od] * s_hi[m0];
        }
      }
  } else if constexpr (ND == 2) {
    for (int q0 = 0; q0 < 3; ++q0)
      for (int e0 = ra[0][q0]; e0 <= rb[0][q0]; ++e0) {
        const int k0_lo = max(0, e0 >> 1), k0_hi = min(a.M[0] - 1, (e0 + L - 2) >> 1);
        for (int k0 = k0_lo; k0 <= k0_hi; ++k0) {
          const int m0 = 2 * k0 + 1 - e0;
          if ((unsigned)m0 >= (unsigned)L) continue;
          const T l0 = s_lo[m0], h0 = s_hi[m0];
          for (int q1 = 0; q1 < 3; ++q1)
            for (int e1 = ra[1][q1]; e1 <= rb[1][q1]; ++e1) {
              const int k1_lo = max(0, e1 >> 1), k1_hi = min(a.M[1] - 1, (e1 + L - 2) >> 1);
              for (int k1 = k1_lo; k1 <= k1_hi; ++k1) {
                const int m1 = 2 * k1 + 1 - e1;
                if ((unsigned)m1 >= (unsigned)L) continue;
                const T l1 = s_lo[m1], h1 = s_hi[m1];
                const int64_t oa = img * a.as[0] + k0 * a.as[1] + k1 * a.as[2], od = img * a.ds[0] + k0 * a.ds[1] + k1 * a.ds[2];
                acc += l0 * (a.gband[0][oa] * l1 + a.gband[1][od] * h1) + h0 * (a.gband[2][od] * l1 + a.gband[3][od] * h1);
              }
            }
        }
      }
  } else {
    for (int q0 = 0; q0 < 3; ++q0)
      for (int e0 = ra[0][q0]; e0 <= rb[0][q0]; ++e0) {
        const int k0_lo = max(0, e0 >> 1), k0_hi = min(a.M[0] - 1, (e0 + L - 2) >> 1);
        for (int k0 = k0_lo; k0 <= k0_hi; ++k0) {
          const int m0 = 2 * k0 + 1 - e0;
          if ((unsigned)m0 >= (unsigned)L) continue;
          const T l0 = s_lo[m0], h0 = s_hi[m0];
          for (int q1 = 0; q1 < 3; ++q1)
            for (int e1 = ra[1][q1]; e1 <= rb[1][q1]; ++e1) {
              const int k1_lo = max(0, e1 >> 1), k1_hi = min(a.M[1] - 1, (e1 + L - 2) >> 1);
              for (int k1 = k1_lo; k1 <= k1_hi; ++k1) {
                const int m1 = 2 * k1 + 1 - e1;
                if ((unsigned)m1 >= (unsigned)L) continue;
                const T l1 = s_lo[m1], h1 = s_hi[m1];
                const T w00 = l0 * l1, w01 = l0 * h1, w10 = h0 * l1, w11 = h0 * h1;  // (axis 0, axis 1) = (lo, lo), (lo, hi), ...
                for (int q2 = 0; q2 < 3; ++q2)
                  for (int e2 = ra[2][q2]; e2 <= rb[2][q2]; ++e2) {
                    const int k2_lo = max(0, e2 >> 1), k2_hi = min(a.M[2] - 1, (e2 + L - 2) >> 1);
                    for (int k2 = k2_lo; k2 <= k2_hi; ++k2) {
                      const int m2 = 2 * k2 + 1 - e2;
                      if ((unsigned)m2 >= (unsigned)L) continue;
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
  int64_t ox = img * a.xs[0];
#pragma unroll
  for (int d = 0; d < ND; ++d) ox += (int64_t)n[d] * a.xs[1 + d];
  a.gx[ox] = acc;
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
  a.per_image = box - inner;
  a.total = a.per_image * d->batch;
  for (int m = 0; m < kMaxTaps; ++m) {
    a.lo[m] = m < a.L ? (T)lo[m] : (T)0;
    a.hi[m] = m < a.L ? (T)hi[m] : (T)0;
  }
  if (a.total == 0) return MIFWT_OK;
  const int64_t blocks = (a.total + 255) / 256;
  if (blocks > INT32_MAX) return MIFWT_ERR_UNSUPPORTED;
  hipLaunchKernelGGL((adjoint_border_kernel<T, ND>), dim3((unsigned)blocks), dim3(256), 0, stream, a);
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
  for (int i = 0; i < d->ndim; ++i) {
    const int64_t n = d->sig_extent[i];
    if (n < 2 * (d->filt_len + 1) || n > INT32_MAX / 4) return false;
  }
  return true;
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
