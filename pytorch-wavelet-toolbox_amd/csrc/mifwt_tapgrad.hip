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
#include "mifwt_stream.h"

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

// The decimated levels (kstride 2, one sample per tap step, L <= 32) — what a learnable-wavelet training step runs 24 times per call on
// image-sized planes.  Round 6: the kernel above maps every one of its 32 x (elements) samples through the boundary rule and divides a
// 64-bit index per element: 0.84 ms per call on config 2's levels, 82 % of a training step (torch profiler, tools/learnable_prof.py).
// Here a WAVE takes 64 consecutive k of one row: it stages the 126 + L samples of b the 64 windows span into LDS once (the boundary rule
// applied per staged sample: three per lane), then every lane reads its L consecutive samples from 2 lane on (8-byte reads, conflict-free)
// and keeps L partial sums; one wave reduction and one double-precision atomic per tap when the wave is through with its tasks.
template <typename T, int LT>
__global__ void __launch_bounds__(256) tap_correlate_rows_kernel(const T* __restrict__ a, const T* __restrict__ b, double* __restrict__ out,
                                                                uint32_t ntasks, FastDiv chunks, FastDiv rpb, int m_len, int n_len, int64_t a_bs, int64_t a_rs,
                                                                int64_t b_bs, int64_t b_rs, int L, int c0, int sgn, int mode) {
  using A = typename std::conditional<std::is_same<T, double>::value, double, float>::type;
  constexpr int KW = 64, SPAN = 2 * KW + 32;  // (126 + L <= 158 staged samples)
  __shared__ __attribute__((aligned(16))) A win[4][SPAN];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  A* const w = win[wave];
  A acc[LT];
#pragma unroll
  for (int t = 0; t < LT; ++t) acc[t] = A(0);
  const int span = 2 * KW - 2 + L;
  for (uint32_t task = blockIdx.x * 4u + (uint32_t)wave; task < ntasks; task += gridDim.x * 4u) {
    uint32_t chunk, rin;
    const uint32_t row = chunks.divmod(task, chunk);
    const uint32_t bat = rpb.divmod(row, rin);  // row = (batch element, row inside it): the two operands' planes may be strided views
    const int k0 = (int)chunk * KW;
    const int base = 2 * k0 + c0 - (sgn < 0 ? L - 1 : 0);  // the first sample of b any of the wave's windows touches
    const T* br = b + (int64_t)bat * b_bs + (int64_t)rin * b_rs;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int i = lane + 64 * j;
      if (i < span) {
        const int src = ext_index_near(base + i, n_len, mode);
        w[i] = src >= 0 ? (A)br[src] : A(0);
      }
    }
    const int k = k0 + lane;
    const A av = k < m_len ? (A)a[(int64_t)bat * a_bs + (int64_t)rin * a_rs + k] : A(0);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // tap t reads sample 2 lane + t (sgn > 0) / 2 lane + L - 1 - t (sgn < 0) of the staged span
    if (sgn > 0) {
#pragma unroll
      for (int t = 0; t < LT; t += 2) {
        if (t < L) {
          const A x0 = w[2 * lane + t], x1 = w[2 * lane + t + 1];
          acc[t] = fma(av, x0, acc[t]);
          if (t + 1 < L) acc[t + 1] = fma(av, x1, acc[t + 1]);
        }
      }
    } else {
#pragma unroll
      for (int t = 0; t < LT; ++t)
        if (t < L) acc[t] = fma(av, w[2 * lane + L - 1 - t], acc[t]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  // wave sums, then ONE atomic per tap and WORKGROUP (the four waves' sums meet in LDS): every atomic of a launch lands on the same L
  // doubles, and 2048 x 4 waves x L of them were what the first version of this kernel spent its time on
  __shared__ double part[4][LT];
#pragma unroll
  for (int t = 0; t < LT; ++t) {
    if (t < L) {
      double v = (double)acc[t];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
      if (lane == 0) part[wave][t] = v;
    }
  }
  __syncthreads();
  if (threadIdx.x < (unsigned)L) atomicAdd(&out[threadIdx.x], part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]);
}


// The same reduction along the MIDDLE axis of [batch, rows, columns] operands (columns contiguous):
//     out[t] += sum_{b, k, c} a[b, k, c] * b_ext[b, 2k + c0 + sgn t, c]
// — the tap gradients along the row axis of a 2-D level with both operands in their natural layout (round 6: the host layer used to
// transpose both in front of the row kernel: a third of a training step was torch's strided copies).  Lane = column, a wave walks a
// chunk of k: the L rows of b a k needs are a window in registers that slides by two rows per k (two coalesced loads per k, the
// boundary rule per new row on the scalar unit), L FMAs per k.
template <typename T, int LT>
__global__ void __launch_bounds__(256) tap_correlate_cols_kernel(const T* __restrict__ a, const T* __restrict__ b, double* __restrict__ out,
                                                                uint32_t ntasks, FastDiv strips, FastDiv kchunks, int kc, int m_len, int n_len, int ncols,
                                                                int64_t a_bs, int64_t a_ks, int64_t b_bs, int64_t b_ks, int L, int c0, int sgn, int mode) {
  using A = typename std::conditional<std::is_same<T, double>::value, double, float>::type;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  A acc[LT];
#pragma unroll
  for (int t = 0; t < LT; ++t) acc[t] = A(0);
  for (uint32_t task = blockIdx.x * 4u + (uint32_t)wave; task < ntasks; task += gridDim.x * 4u) {
    uint32_t strip, chunk;
    const uint32_t bk = strips.divmod(task, strip);
    const uint32_t bat = kchunks.divmod(bk, chunk);
    const int col = (int)strip * 64 + lane;
    const bool on = col < ncols;
    const int cc = on ? col : 0;
    const int k_lo = (int)chunk * kc, k_hi = min(m_len, k_lo + kc);
    const T* ap = a + (int64_t)bat * a_bs + cc;
    const T* bp = b + (int64_t)bat * b_bs + cc;
    // window: w[t] = the sample tap t reads at the current k = b_ext[2 k + c0 + sgn t]; it slides by two rows per k
    A w[LT];
    auto row_of = [&](int p) -> A {  // (p is wave-uniform)
      const int src = ext_index_near(p, n_len, mode);
      return src >= 0 ? (A)bp[(int64_t)src * b_ks] : A(0);
    };
#pragma unroll
    for (int t = 0; t < LT; ++t) w[t] = t < L ? row_of(2 * k_lo + c0 + sgn * t) : A(0);
    for (int k = k_lo; k < k_hi; ++k) {
      const A av = on ? (A)ap[(int64_t)k * a_ks] : A(0);
#pragma unroll
      for (int t = 0; t < LT; ++t)
        if (t < L) acc[t] = fma(av, w[t], acc[t]);
      if (k + 1 < k_hi) {
        const int p1 = 2 * (k + 1) + c0;
        if (sgn > 0) {  // w[t] <- w[t + 2]; the two new rows are the taps L - 2, L - 1
          const A n0 = row_of(p1 + L - 2), n1 = row_of(p1 + L - 1);
#pragma unroll
          for (int t = 0; t < LT; ++t) {
            if (t + 2 < L) w[t] = w[t + 2 < LT ? t + 2 : 0];
            else if (t + 2 == L) w[t] = n0;
            else if (t + 1 == L) w[t] = n1;
          }
        } else {  // w[t] <- w[t - 2]; the two new rows are the taps 0, 1
          const A n0 = row_of(p1), n1 = row_of(p1 - 1);
#pragma unroll
          for (int t = LT - 1; t >= 2; --t) w[t] = w[t - 2];
          w[0] = n0;
          if (LT > 1) w[1] = n1;
        }
      }
    }
  }
  __shared__ double part[4][LT];
#pragma unroll
  for (int t = 0; t < LT; ++t) {
    if (t < L) {
      double v = (double)acc[t];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
      if (lane == 0) part[wave][t] = v;
    }
  }
  __syncthreads();
  if (threadIdx.x < (unsigned)L) atomicAdd(&out[threadIdx.x], part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]);
}
}  // namespace

}  // namespace mifwt

namespace mifwt {
namespace {

template <typename T>
int tap_correlate_rows(int64_t batch, int64_t rows_per_batch, int64_t m_len, int64_t n_len, const T* a, int64_t a_bs, int64_t a_rs, const T* b,
                       int64_t b_bs, int64_t b_rs, int L, int c0, int sgn, int mode, double* out, hipStream_t st) {
  const int64_t rows = batch * rows_per_batch;
  const int64_t chunks = (m_len + 63) / 64, ntasks = rows * chunks;
  const FastDiv rpb = make_fastdiv((uint32_t)rows_per_batch);
  const int64_t want = (ntasks + 3) / 4;
  const unsigned grid = (unsigned)(want < 1024 ? want : 1024);  // (four workgroups per CU: the loads of one hide behind the sums of another)
  const FastDiv dv = make_fastdiv((uint32_t)chunks);
#define MIFWT_TC_LAUNCH(LT) \
  hipLaunchKernelGGL((tap_correlate_rows_kernel<T, LT>), dim3(grid), dim3(256), 0, st, a, b, out, (uint32_t)ntasks, dv, rpb, (int)m_len, (int)n_len, a_bs, \
                     a_rs, b_bs, b_rs, L, c0, sgn, mode)
  if (L <= 8) MIFWT_TC_LAUNCH(8);
  else if (L <= 16) MIFWT_TC_LAUNCH(16);
  else MIFWT_TC_LAUNCH(32);
#undef MIFWT_TC_LAUNCH
  return hipGetLastError() == hipSuccess ? MIFWT_OK : MIFWT_ERR_LAUNCH;
}

int tap_correlate(int dtype, int64_t rows, int64_t m_len, int64_t n_len, const void* a, int64_t a_row_stride, const void* b,
                  int64_t b_row_stride, int filt_len, int c0, int sgn, int kstride, int mode, double* out, void* stream) {
  if (!a || !b || !out || rows < 0 || m_len < 1 || n_len < 1 || filt_len < 1 || filt_len > MIFWT_MAX_FILT) return MIFWT_ERR_BADARG;
  if (mode < MIFWT_MODE_ZERO || mode > MIFWT_MODE_SYMMETRIC) return MIFWT_ERR_BADARG;
  if (m_len > INT32_MAX / 4 || n_len > INT32_MAX / 4) return MIFWT_ERR_UNSUPPORTED;
  if (rows == 0) return MIFWT_OK;
  hipStream_t st0 = static_cast<hipStream_t>(stream);
  // the decimated levels on the row kernel (every launch of a learnable-wavelet training step); dilated / long filters below
  if (kstride == 2 && (sgn == 1 || sgn == -1) && filt_len <= 32 && rows * ((m_len + 63) / 64) < (int64_t(1) << 31) && !g_options[MIFWT_OPT_FORCE_GENERIC]) {
    if (dtype == MIFWT_F32)
      return tap_correlate_rows(1, rows, m_len, n_len, static_cast<const float*>(a), 0, a_row_stride, static_cast<const float*>(b), 0, b_row_stride, filt_len, c0, sgn, mode, out, st0);
    if (dtype == MIFWT_F64)
      return tap_correlate_rows(1, rows, m_len, n_len, static_cast<const double*>(a), 0, a_row_stride, static_cast<const double*>(b), 0, b_row_stride, filt_len, c0, sgn, mode, out, st0);
  }
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

// Round 6: the decimated reduction on operands in their NATURAL layout [batch, rows, columns] (element strides; columns contiguous):
// along = 1: along the columns (the row kernel with a two-level row index: no copy of a strided band plane), along = 0: along the rows
// (the column kernel).  a: [batch, ra, ca], b: [batch, rb, cb] with ra == rb (along 1) / ca == cb (along 0).
namespace mifwt {
namespace {
template <typename T>
int tap_correlate_cols(int64_t batch, int64_t m_len, int64_t n_len, int64_t ncols, const T* a, int64_t a_bs, int64_t a_ks, const T* b, int64_t b_bs,
                       int64_t b_ks, int L, int c0, int sgn, int mode, double* out, hipStream_t st) {
  const int64_t strips = (ncols + 63) / 64;
  int kc = 32;
  while (kc > 8 && batch * strips * ((m_len + kc - 1) / kc) < 4096) kc /= 2;  // (enough waves for the chip; a chunk re-reads L - 2 rows)
  const int64_t kchunks = (m_len + kc - 1) / kc, ntasks = batch * kchunks * strips;
  if (ntasks >= (int64_t(1) << 31)) return MIFWT_ERR_UNSUPPORTED;
  const int64_t want = (ntasks + 3) / 4;
  const unsigned grid = (unsigned)(want < 2048 ? want : 2048);
  const FastDiv ds = make_fastdiv((uint32_t)strips), dk = make_fastdiv((uint32_t)kchunks);
#define MIFWT_TCC_LAUNCH(LT) \
  hipLaunchKernelGGL((tap_correlate_cols_kernel<T, LT>), dim3(grid), dim3(256), 0, st, a, b, out, (uint32_t)ntasks, ds, dk, kc, (int)m_len, (int)n_len, \
                     (int)ncols, a_bs, a_ks, b_bs, b_ks, L, c0, sgn, mode)
  if (L <= 8) MIFWT_TCC_LAUNCH(8);
  else if (L <= 16) MIFWT_TCC_LAUNCH(16);
  else MIFWT_TCC_LAUNCH(32);
#undef MIFWT_TCC_LAUNCH
  return hipGetLastError() == hipSuccess ? MIFWT_OK : MIFWT_ERR_LAUNCH;
}
}  // namespace
}  // namespace mifwt

extern "C" int mifwt_tap_correlate_planes(int dtype, int along, int64_t batch, int64_t a_rows, int64_t a_cols, int64_t b_rows, int64_t b_cols, const void* a,
                                          int64_t a_batch_stride, int64_t a_row_stride, const void* b, int64_t b_batch_stride, int64_t b_row_stride,
                                          int filt_len, int c0, int sgn, int mode, double* out, void* stream) {
  using namespace mifwt;
  if (!a || !b || !out || batch < 0 || a_rows < 1 || a_cols < 1 || b_rows < 1 || b_cols < 1 || (sgn != 1 && sgn != -1)) return MIFWT_ERR_BADARG;
  if (filt_len < 1 || filt_len > 32 || mode < MIFWT_MODE_ZERO || mode > MIFWT_MODE_SYMMETRIC) return MIFWT_ERR_BADARG;
  if ((along == 1 && a_rows != b_rows) || (along == 0 && a_cols != b_cols) || (along != 0 && along != 1)) return MIFWT_ERR_BADARG;
  if (a_rows > INT32_MAX / 4 || b_rows > INT32_MAX / 4 || a_cols > INT32_MAX / 4 || b_cols > INT32_MAX / 4) return MIFWT_ERR_UNSUPPORTED;
  if (dtype != MIFWT_F32 && dtype != MIFWT_F64) return MIFWT_ERR_UNSUPPORTED;
  if (batch == 0) return MIFWT_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (along == 1) {
    if (batch * a_rows * ((a_cols + 63) / 64) >= (int64_t(1) << 31)) return MIFWT_ERR_UNSUPPORTED;
    return dtype == MIFWT_F32
               ? tap_correlate_rows(batch, a_rows, a_cols, b_cols, static_cast<const float*>(a), a_batch_stride, a_row_stride, static_cast<const float*>(b),
                                    b_batch_stride, b_row_stride, filt_len, c0, sgn, mode, out, st)
               : tap_correlate_rows(batch, a_rows, a_cols, b_cols, static_cast<const double*>(a), a_batch_stride, a_row_stride, static_cast<const double*>(b),
                                    b_batch_stride, b_row_stride, filt_len, c0, sgn, mode, out, st);
  }
  return dtype == MIFWT_F32
             ? tap_correlate_cols(batch, a_rows, b_rows, a_cols, static_cast<const float*>(a), a_batch_stride, a_row_stride, static_cast<const float*>(b),
                                  b_batch_stride, b_row_stride, filt_len, c0, sgn, mode, out, st)
             : tap_correlate_cols(batch, a_rows, b_rows, a_cols, static_cast<const double*>(a), a_batch_stride, a_row_stride, static_cast<const double*>(b),
                                  b_batch_stride, b_row_stride, filt_len, c0, sgn, mode, out, st);
}
