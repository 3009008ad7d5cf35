// mifwt_tapgrad.hip — the reduction behind the gradients w.r.t. FILTER TAPS (learnable wavelets).
//
// The reference keeps the taps in the autograd graph (torch.as_tensor in src/ptwt/_util.py:132; learnable banks in
// src/ptwt/wavelets_learnable.py, trained in examples/network_compression/wavelet_linear.py:118,150), so ATen's conv
// backward produces d loss / d taps.  Both level maps are linear in the taps; along the transformed axis
//   analysis   c[k] = sum_m h[m] z_ext[2k + 1 - m]          =>  dL/dh[m] = sum_{rows, k} g_c[k] z_ext[2k + 1 - m]
//   synthesis  y[n] = sum_k u[k] g[n + L - 2 - 2k]            =>  dL/dg[t] = sum_{rows, k} u[k]  g_y[2k + t - (L - 2)]
// (z / u = the level input / coefficients with the OTHER axes already transformed — the host layer composes that from
// ordinary level calls).  Both are one correlation
//     out[t] += sum_{rows} sum_k a[row, k] * b_ext[row, 2k + c0 + sgn * t],      t in [0, L)
// with b extended by the boundary rule (analysis) or by zeros (synthesis).  The stationary levels (swt / iswt,
// src/ptwt/stationary_transform.py:95-107, :142-156: stride 1, dilation D, periodic extension with any number of wraps) are the
// same reduction with unit stride in k and a step of -D per tap:
//     out[t] += sum_{rows} sum_k a[row, k] * b[row, (k + c0 + tstep * t) mod n]            (mifwt_tap_correlate_dilated).
// One thread block strides over (row, k),
// keeps L partial sums per thread in registers (chunks of 32 taps), reduces them across the wave with DPP shuffles and
// issues one double-precision atomic per tap and wave.
#include "mifwt_common.h"

namespace mifwt {

namespace {

template <typename T>
__global__ void __launch_bounds__(256) tap_correlate_kernel(const T* __restrict__ a, const T* __restrict__ b, double* __restrict__ out,
                                                           int64_t rows, int m_len, int n_len, int64_t a_rs, int64_t b_rs, int L, int c0,
                                                           int tstep, int kstride, int mode, int t0) {
  using A = typename std::conditional<std::is_same<T, double>::value, double, float>::type;
  constexpr int TC = 32;  // taps per pass
  A acc[TC];
#pragma unroll
  for (int t = 0; t < TC; ++t) acc[t] = A(0);
  const int64_t total = rows * (int64_t)m_len;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = idx / m_len;
    const int k = (int)(idx - row * m_len);
    const A av = (A)a[row * a_rs + k];
    const T* br = b + row * b_rs;
#pragma unroll
    for (int t = 0; t < TC; ++t) {
      if (t0 + t < L) {
        const int src = ext_index_near(kstride * k + c0 + tstep * (t0 + t), n_len, mode);
        if (src >= 0) acc[t] = fma(av, (A)br[src], acc[t]);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < TC; ++t) {
    if (t0 + t >= L) break;
    double v = (double)acc[t];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(&out[t0 + t], v);
  }
}

}  // namespace

}  // namespace mifwt

namespace mifwt {
namespace {

int tap_correlate(int dtype, int64_t rows, int64_t m_len, int64_t n_len, const void* a, int64_t a_row_stride, const void* b,
                  int64_t b_row_stride, int filt_len, int c0, int sgn, int kstride, int mode, double* out, void* stream) {
  if (!a || !b || !out || rows < 0 || m_len < 1 || n_len < 1 || filt_len < 1 || filt_len > MIFWT_MAX_FILT) return MIFWT_ERR_BADARG;
  if (mode < MIFWT_MODE_ZERO || mode > MIFWT_MODE_SYMMETRIC) return MIFWT_ERR_BADARG;
  if (m_len > INT32_MAX / 4 || n_len > INT32_MAX / 4) return MIFWT_ERR_UNSUPPORTED;
  if (rows == 0) return MIFWT_OK;
  const int64_t total = rows * m_len;
  const int64_t want = (total + 255) / 256;
  const unsigned grid = (unsigned)(want < 2048 ? want : 2048);
  hipStream_t st = static_cast<hipStream_t>(stream);
  for (int t0 = 0; t0 < filt_len; t0 += 32) {
    if (dtype == MIFWT_F32)
      hipLaunchKernelGGL(tap_correlate_kernel<float>, dim3(grid), dim3(256), 0, st, static_cast<const float*>(a),
                         static_cast<const float*>(b), out, rows, (int)m_len, (int)n_len, a_row_stride, b_row_stride, filt_len, c0,
                         sgn, kstride, mode, t0);
    else if (dtype == MIFWT_F64)
      hipLaunchKernelGGL(tap_correlate_kernel<double>, dim3(grid), dim3(256), 0, st, static_cast<const double*>(a),
                         static_cast<const double*>(b), out, rows, (int)m_len, (int)n_len, a_row_stride, b_row_stride, filt_len, c0,
                         sgn, kstride, mode, t0);
    else
      return MIFWT_ERR_UNSUPPORTED;
    if (hipGetLastError() != hipSuccess) return MIFWT_ERR_LAUNCH;
  }
  return MIFWT_OK;
}

}  // namespace
}  // namespace mifwt

extern "C" int mifwt_tap_correlate(int dtype, int64_t rows, int64_t m_len, int64_t n_len, const void* a, int64_t a_row_stride,
                                   const void* b, int64_t b_row_stride, int filt_len, int c0, int sgn, int mode, double* out,
                                   void* stream) {
  if (sgn != 1 && sgn != -1) return MIFWT_ERR_BADARG;
  return mifwt::tap_correlate(dtype, rows, m_len, n_len, a, a_row_stride, b, b_row_stride, filt_len, c0, sgn, 2, mode, out, stream);
}

extern "C" int mifwt_tap_correlate_dilated(int dtype, int64_t rows, int64_t n, const void* a, int64_t a_row_stride, const void* b,
                                           int64_t b_row_stride, int filt_len, int64_t c0, int64_t tstep, double* out, void* stream) {
  if (c0 > INT32_MAX / 8 || c0 < -(INT32_MAX / 8) || tstep * filt_len > INT32_MAX / 8 || tstep * filt_len < -(INT32_MAX / 8)) return MIFWT_ERR_UNSUPPORTED;
  return mifwt::tap_correlate(dtype, rows, n, n, a, a_row_stride, b, b_row_stride, filt_len, (int)c0, (int)tstep, 1, MIFWT_MODE_PERIODIC, out,
                              stream);
}
