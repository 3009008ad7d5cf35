// mifwt_generic.hip — generic per-axis analysis / synthesis kernels for gfx950.
//
// The catch-all behind the fused fast paths: one transformed axis per launch, any filter length up to
// 128 taps, any boundary mode, arbitrary element strides, f32 / f64 in their own precision and f16 STORAGE with f32 arithmetic (the
// intermediate passes of an N-D level are then rounded to f16 like every other f16 path of the library).  A thread owns one output
// coefficient position and produces the low- and high-pass value together (analysis), or one output
// sample (synthesis, polyphase gather: only the L/2 non-zero products of the transposed convolution are
// formed and only the cropped interior is computed).  Up to four independent (input -> lo, hi) jobs of
// identical geometry ride in one launch (blockIdx.y), which is how an N-D level is assembled from axis
// passes without extra launches.
//
// Math (reference src/ptwt/conv_transform.py:133-139 and :184-199):
//   analysis  c[k] = sum_m h[m] * x_ext[2k + 1 - m],     k in [0, M),  M = floor((N + L - 1) / 2)
//   synthesis y[n] = sum_k a[k] g_lo[n + L - 2 - 2k] + d[k] g_hi[n + L - 2 - 2k],  n in [0, 2M - L + 2 - t)
#include "mifwt_common.h"

namespace mifwt {

template <typename T>
struct Taps {
  T lo[kMaxFilt];
  T hi[kMaxFilt];
};

struct AxisGeom {
  int64_t ext[4];  // output iteration space (batch, axis0, axis1, axis2), unused dims = 1
  int64_t total;   // product of ext
  int taxis;       // transformed dim of the iteration space
  int n_src;       // source extent along taxis (analysis: N, synthesis / adjoint: M)
  int n_sig;       // adjoint only: signal extent N along taxis
  int mode;
  int filt_len;
};

struct AxisJobs {
  AxisJob job[4];
};

thread_local DeviceTaps g_dtaps = {nullptr, nullptr, 0};
thread_local int g_dtaps_taken = 0;

// tap m of the pass: from the launch arguments, or (DT) from the device arrays of a filter bank that lives on the GPU
template <bool DT, typename A>
__device__ __forceinline__ A tap_of(const A* arg, const double* dev, int rev, int L, int m) {
  if constexpr (DT) return (A)dev[rev ? L - 1 - m : m];
  else return arg[m];
}

template <typename T, typename A, bool DT>
__global__ void __launch_bounds__(256) axis_fwd_kernel(AxisJobs jobs, AxisGeom g, Taps<A> taps, DeviceTaps dt) {
  const AxisJob& jb = jobs.job[blockIdx.y];
  const T* __restrict__ in = static_cast<const T*>(jb.in0);
  T* __restrict__ lo = static_cast<T*>(jb.out0);
  T* __restrict__ hi = static_cast<T*>(jb.out1);
  const int64_t stride_t = jb.in0_stride[g.taxis];
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < g.total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t rem = idx;
    int64_t c[4];
#pragma unroll
    for (int d = 3; d >= 0; --d) {
      c[d] = rem % g.ext[d];
      rem /= g.ext[d];
    }
    int64_t ibase = 0, lbase = 0, hbase = 0;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      if (d != g.taxis) ibase += c[d] * jb.in0_stride[d];
      lbase += c[d] * jb.out0_stride[d];
      hbase += c[d] * jb.out1_stride[d];
    }
    const int k = (int)c[g.taxis];
    A acc_lo = 0, acc_hi = 0;
    for (int m = 0; m < g.filt_len; ++m) {
      const int src = ext_index(2 * k + 1 - m, g.n_src, g.mode);
      const A v = src >= 0 ? (A)in[ibase + (int64_t)src * stride_t] : A(0);
      acc_lo = fma(tap_of<DT, A>(taps.lo, dt.lo, dt.rev, g.filt_len, m), v, acc_lo);
      acc_hi = fma(tap_of<DT, A>(taps.hi, dt.hi, dt.rev, g.filt_len, m), v, acc_hi);
    }
    lo[lbase] = (T)acc_lo;
    hi[hbase] = (T)acc_hi;
  }
}

template <typename T, typename A, bool DT>
__global__ void __launch_bounds__(256) axis_inv_kernel(AxisJobs jobs, AxisGeom g, Taps<A> taps, DeviceTaps dt) {
  const AxisJob& jb = jobs.job[blockIdx.y];
  const T* __restrict__ a = static_cast<const T*>(jb.in0);
  const T* __restrict__ dd = static_cast<const T*>(jb.in1);
  T* __restrict__ y = static_cast<T*>(jb.out0);
  const int64_t sa = jb.in0_stride[g.taxis], sd = jb.in1_stride[g.taxis];
  const int L = g.filt_len;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < g.total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t rem = idx;
    int64_t c[4];
#pragma unroll
    for (int d = 3; d >= 0; --d) {
      c[d] = rem % g.ext[d];
      rem /= g.ext[d];
    }
    int64_t abase = 0, dbase = 0, ybase = 0;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      if (d != g.taxis) {
        abase += c[d] * jb.in0_stride[d];
        dbase += c[d] * jb.in1_stride[d];
      }
      ybase += c[d] * jb.out0_stride[d];
    }
    // y[n] = u[n + L - 2];  u[q] = sum_k a[k] g[q - 2k]  with 0 <= q - 2k <= L - 1
    const int q = (int)c[g.taxis] + L - 2;
    int k_lo = (q - (L - 1) + 1) >> 1;  // ceil((q - L + 1) / 2), q - L + 1 may be negative
    if (k_lo < 0) k_lo = 0;
    int k_hi = q >> 1;
    if (k_hi > g.n_src - 1) k_hi = g.n_src - 1;
    A acc = 0;
    for (int k = k_lo; k <= k_hi; ++k) {
      const int t = q - 2 * k;
      acc = fma(tap_of<DT, A>(taps.lo, dt.lo, dt.rev, L, t), (A)a[abase + (int64_t)k * sa], acc);
      acc = fma(tap_of<DT, A>(taps.hi, dt.hi, dt.rev, L, t), (A)dd[dbase + (int64_t)k * sd], acc);
    }
    y[ybase] = (T)acc;
  }
}

// Adjoint (transpose) of one analysis axis pass: (g_lo, g_hi) -> g_x.  With u[e] the full transposed
// convolution over the EXTENDED index range e in [-(L-2), N + L - 2 + N%2),
//     u[e] = sum_k g_lo[k] h_lo[2k + 1 - e] + g_hi[k] h_hi[2k + 1 - e],
// the gradient folds the halo back through the boundary index map:  g_x[i] = sum_{e : ext_index(e) = i} u[e].
// (The reference gets this from ATen autograd through F.pad / _pad_symmetric + F.conv*d.)
template <typename T, typename A, bool DT>
__global__ void __launch_bounds__(256) axis_adj_kernel(AxisJobs jobs, AxisGeom g, Taps<A> taps, DeviceTaps dt) {
  const AxisJob& jb = jobs.job[blockIdx.y];
  const T* __restrict__ a = static_cast<const T*>(jb.in0);
  const T* __restrict__ dd = static_cast<const T*>(jb.in1);
  T* __restrict__ y = static_cast<T*>(jb.out0);
  const int64_t sa = jb.in0_stride[g.taxis], sd = jb.in1_stride[g.taxis];
  const int L = g.filt_len, N = g.n_sig, M = g.n_src;
  const int pl = L - 2, pr = L - 2 + (N & 1);
  const int border = (pl > pr ? pl : pr) + 1;  // halo indices only ever map into [0, border) or [N - border, N)
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < g.total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t rem = idx;
    int64_t c[4];
#pragma unroll
    for (int d = 3; d >= 0; --d) {
      c[d] = rem % g.ext[d];
      rem /= g.ext[d];
    }
    int64_t abase = 0, dbase = 0, ybase = 0;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      if (d != g.taxis) {
        abase += c[d] * jb.in0_stride[d];
        dbase += c[d] * jb.in1_stride[d];
      }
      ybase += c[d] * jb.out0_stride[d];
    }
    const int i = (int)c[g.taxis];
    auto u = [&](int e) -> A {  // 2k + 1 - e in [0, L)  <=>  k in [ceil((e - 1) / 2), floor((e + L - 2) / 2)]
      int k_lo = e >> 1;        // ceil((e - 1) / 2) = floor(e / 2) for every integer e
      if (k_lo < 0) k_lo = 0;
      int k_hi = (e + L - 2) >> 1;
      if (k_hi > M - 1) k_hi = M - 1;
      A acc = 0;
      for (int k = k_lo; k <= k_hi; ++k) {
        const int m = 2 * k + 1 - e;
        acc = fma(tap_of<DT, A>(taps.lo, dt.lo, dt.rev, L, m), (A)a[abase + (int64_t)k * sa], acc);
        acc = fma(tap_of<DT, A>(taps.hi, dt.hi, dt.rev, L, m), (A)dd[dbase + (int64_t)k * sd], acc);
      }
      return acc;
    };
    A acc = u(i);
    if (g.mode != MIFWT_MODE_ZERO && (i < border || i >= N - border)) {
      for (int e = -pl; e < 0; ++e)
        if (ext_index(e, N, g.mode) == i) acc += u(e);
      for (int e = N; e < N + pr; ++e)
        if (ext_index(e, N, g.mode) == i) acc += u(e);
    }
    y[ybase] = (T)acc;
  }
}

template <typename T, typename A = T>
static int launch_axis(int kind, const AxisJob* jobs, int njobs, const int64_t out_ext[4], int taxis,
                       int64_t n_src, int mode, int filt_len, const double* lo, const double* hi,
                       hipStream_t stream, int64_t n_sig = 0) {  // kind: 0 analysis, 1 synthesis, 2 analysis adjoint
  if (njobs < 1 || njobs > 4 || filt_len < 1 || filt_len > kMaxFilt) return MIFWT_ERR_BADARG;
  AxisJobs js;
  for (int i = 0; i < 4; ++i) js.job[i] = jobs[i < njobs ? i : 0];
  AxisGeom g;
  g.total = 1;
  for (int d = 0; d < 4; ++d) {
    g.ext[d] = out_ext[d];
    g.total *= out_ext[d];
  }
  if (g.total == 0) return MIFWT_OK;
  g.taxis = taxis;
  g.n_src = (int)n_src;
  g.n_sig = (int)n_sig;
  g.mode = mode;
  g.filt_len = filt_len;
  const DeviceTaps dt = g_dtaps;  // (this thread's: set around the call by a mifwt_*_dtaps entry point)
  Taps<A> taps;
  for (int m = 0; m < kMaxFilt; ++m) {
    taps.lo[m] = (!dt.lo && m < filt_len) ? (A)lo[m] : A(0);
    taps.hi[m] = (!dt.lo && m < filt_len) ? (A)hi[m] : A(0);
  }
  const int64_t want = (g.total + 255) / 256;
  const unsigned gx = (unsigned)(want < 8192 ? want : 8192);  // grid-stride beyond 256 CUs x 32 blocks
  dim3 grid(gx, (unsigned)njobs), block(256);
  if (dt.lo) {
    if (kind == 2)
      hipLaunchKernelGGL((axis_adj_kernel<T, A, true>), grid, block, 0, stream, js, g, taps, dt);
    else if (kind == 1)
      hipLaunchKernelGGL((axis_inv_kernel<T, A, true>), grid, block, 0, stream, js, g, taps, dt);
    else
      hipLaunchKernelGGL((axis_fwd_kernel<T, A, true>), grid, block, 0, stream, js, g, taps, dt);
  } else if (kind == 2)
    hipLaunchKernelGGL((axis_adj_kernel<T, A, false>), grid, block, 0, stream, js, g, taps, dt);
  else if (kind == 1)
    hipLaunchKernelGGL((axis_inv_kernel<T, A, false>), grid, block, 0, stream, js, g, taps, dt);
  else
    hipLaunchKernelGGL((axis_fwd_kernel<T, A, false>), grid, block, 0, stream, js, g, taps, dt);
  return hipGetLastError() == hipSuccess ? MIFWT_OK : MIFWT_ERR_LAUNCH;
}

int launch_axis_fwd(int dtype, const AxisJob* jobs, int njobs, const int64_t out_ext[4], int taxis, int64_t n_in,
                    int mode, int filt_len, const double* lo, const double* hi, hipStream_t stream) {
  if (dtype == MIFWT_F32) return launch_axis<float>(0, jobs, njobs, out_ext, taxis, n_in, mode, filt_len, lo, hi, stream);
  if (dtype == MIFWT_F64) return launch_axis<double>(0, jobs, njobs, out_ext, taxis, n_in, mode, filt_len, lo, hi, stream);
  if (dtype == MIFWT_F16) return launch_axis<_Float16, float>(0, jobs, njobs, out_ext, taxis, n_in, mode, filt_len, lo, hi, stream);
  return MIFWT_ERR_UNSUPPORTED;
}

int launch_axis_inv(int dtype, const AxisJob* jobs, int njobs, const int64_t out_ext[4], int taxis, int64_t m_in,
                    int filt_len, const double* lo, const double* hi, hipStream_t stream) {
  if (dtype == MIFWT_F32) return launch_axis<float>(1, jobs, njobs, out_ext, taxis, m_in, 0, filt_len, lo, hi, stream);
  if (dtype == MIFWT_F64) return launch_axis<double>(1, jobs, njobs, out_ext, taxis, m_in, 0, filt_len, lo, hi, stream);
  if (dtype == MIFWT_F16) return launch_axis<_Float16, float>(1, jobs, njobs, out_ext, taxis, m_in, 0, filt_len, lo, hi, stream);
  return MIFWT_ERR_UNSUPPORTED;
}

int launch_axis_adj(int dtype, const AxisJob* jobs, int njobs, const int64_t out_ext[4], int taxis, int64_t m_in,
                    int64_t n_sig, int mode, int filt_len, const double* lo, const double* hi, hipStream_t stream) {
  if (dtype == MIFWT_F32) return launch_axis<float>(2, jobs, njobs, out_ext, taxis, m_in, mode, filt_len, lo, hi, stream, n_sig);
  if (dtype == MIFWT_F64) return launch_axis<double>(2, jobs, njobs, out_ext, taxis, m_in, mode, filt_len, lo, hi, stream, n_sig);
  if (dtype == MIFWT_F16) return launch_axis<_Float16, float>(2, jobs, njobs, out_ext, taxis, m_in, mode, filt_len, lo, hi, stream, n_sig);
  return MIFWT_ERR_UNSUPPORTED;
}

}  // namespace mifwt
