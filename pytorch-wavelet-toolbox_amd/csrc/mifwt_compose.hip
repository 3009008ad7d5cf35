// mifwt_compose.hip — N-D levels composed from the fused 2-D plane kernel and the streaming axis passes.
//
// Routes (kernel ids, include/mifwt.h mifwt_kernel_id):
//   kDwt3FwdStream / kDwt3InvStream  (5, 6)  ndim 3, f32, L <= 16: the fused 2-D kernel over the (H, W) plane of
//                                            every (batch, depth) slice + ONE streaming pass along depth
//                                            (replaces F.conv3d / F.conv_transpose3d of the reference,
//                                            src/ptwt/conv_transform_3.py:127, :218); scratch = 4 planes per
//                                            slice, traffic = 2x the algorithmic bytes
//   kDwt1FwdRow / kDwt1InvRow        (3, 4)  unit innermost stride and L in the streaming set, f32 / f64 / f16:
//                                            the inner-axis pass (the whole transform in 1-D: F.conv1d /
//                                            F.conv_transpose1d, src/ptwt/conv_transform.py:137, :187) plus
//                                            one outer-axis pass per further axis through dense scratch
#include <string.h>

#include "mifwt_common.h"

namespace mifwt {

#define MIFWT_STREAM_LENGTHS(X, n) X(n, 2) X(n, 4) X(n, 6) X(n, 8) X(n, 10) X(n, 12) X(n, 14) X(n, 16) X(n, 18) X(n, 20) X(n, 24) X(n, 32)
#define MIFWT_DECL(n, L) int stream_call_##n##_L##L(int, const StreamCall&);
MIFWT_STREAM_LENGTHS(MIFWT_DECL, f32)
MIFWT_STREAM_LENGTHS(MIFWT_DECL, f64)
MIFWT_STREAM_LENGTHS(MIFWT_DECL, f16)
#undef MIFWT_DECL

bool stream_filter_supported(int L) {
  return (L >= 2 && L <= 20 && (L & 1) == 0) || L == 24 || L == 32;
}

int stream_call(int dtype, int kind, const StreamCall& c) {
#define MIFWT_CASE(n, L) case L: return stream_call_##n##_L##L(kind, c);
  switch (dtype) {
    case MIFWT_F32:
      switch (c.filt_len) { MIFWT_STREAM_LENGTHS(MIFWT_CASE, f32) default: break; }
      break;
    case MIFWT_F64:
      switch (c.filt_len) { MIFWT_STREAM_LENGTHS(MIFWT_CASE, f64) default: break; }
      break;
    case MIFWT_F16:
      switch (c.filt_len) { MIFWT_STREAM_LENGTHS(MIFWT_CASE, f16) default: break; }
      break;
    default: break;
  }
#undef MIFWT_CASE
  return MIFWT_ERR_UNSUPPORTED;
}

namespace {

inline int64_t elem_size(int dtype) { return dtype == MIFWT_F64 ? 8 : (dtype == MIFWT_F16 ? 2 : 4); }

// axes from..nd-1 form one dense run (innermost stride 1, each outer one = product of the inner extents)
inline bool dense_under(const int64_t* stride, const int64_t* ext, int nd, int from) {
  int64_t run = 1;
  for (int a = nd - 1; a >= from; --a) {
    if (stride[1 + a] != run) return false;
    run *= ext[a];
  }
  return true;
}

inline void set3(int64_t dst[3], int64_t a, int64_t b, int64_t c) {
  dst[0] = a;
  dst[1] = b;
  dst[2] = c;
}

// the 2-D problem of the depth slices of a 3-D level, writing / reading scratch [slices, 4, Ho, Wo]
mifwt_level_desc plane_desc(const mifwt_level_desc* d, int64_t depth, bool* foldable) {
  mifwt_level_desc p = *d;
  const int64_t plane = d->coef_extent[1] * d->coef_extent[2];
  p.ndim = 2;
  for (int a = 0; a < 2; ++a) {
    p.sig_extent[a] = d->sig_extent[1 + a];
    p.coef_extent[a] = d->coef_extent[1 + a];
    p.sig_stride[1 + a] = d->sig_stride[2 + a];
  }
  p.approx_stride[1] = p.detail_stride[1] = d->coef_extent[2];
  p.approx_stride[2] = p.detail_stride[2] = 1;
  p.approx_stride[0] = p.detail_stride[0] = 4 * plane;
  *foldable = d->batch == 1 || d->sig_stride[0] == depth * d->sig_stride[1];
  p.batch = *foldable ? d->batch * depth : depth;
  p.sig_stride[0] = d->sig_stride[1];
  return p;
}

}  // namespace

// ---- LDS-tile fused 2-D analysis: envelope and per-(type, length) instantiation units ---------------------------
int dwt2_fwd_tile_f32_short(const mifwt_level_desc*, const void*, void*, void* const*, const double*, const double*, hipStream_t);
int dwt2_fwd_tile_f16_short(const mifwt_level_desc*, const void*, void*, void* const*, const double*, const double*, hipStream_t);
int dwt2_fwd_tile_f64_short(const mifwt_level_desc*, const void*, void*, void* const*, const double*, const double*, hipStream_t);
int dwt2_fwd_tile_long18(const mifwt_level_desc*, const void*, void*, void* const*, const double*, const double*, hipStream_t);
int dwt2_fwd_tile_long20(const mifwt_level_desc*, const void*, void*, void* const*, const double*, const double*, hipStream_t);
int dwt2_fwd_tile_long24(const mifwt_level_desc*, const void*, void*, void* const*, const double*, const double*, hipStream_t);
int dwt2_fwd_tile_long32(const mifwt_level_desc*, const void*, void*, void* const*, const double*, const double*, hipStream_t);

bool dwt2_fwd_tile_supported(const mifwt_level_desc* d) {
  if (d->ndim != 2 || (d->dtype != MIFWT_F32 && d->dtype != MIFWT_F16 && d->dtype != MIFWT_F64)) return false;
  const int L = d->filt_len;
  if (!((L >= 2 && L <= 20 && (L & 1) == 0) || L == 24 || L == 32)) return false;
  if (d->dtype == MIFWT_F64 && L > 16) return false;  // f64 instantiations: mifwt_dwt2_fwd_tile_f64.hip
  if (d->sig_stride[2] != 1 || d->approx_stride[2] != 1 || d->detail_stride[2] != 1) return false;
  // one image must be addressable with 32-bit byte offsets below 2^31 (buffer-resource loads, out-of-range switch)
  const int64_t span = (d->sig_extent[0] - 1) * d->sig_stride[1] + d->sig_extent[1];
  if (d->sig_stride[1] < 0 || span >= (int64_t(1) << (d->dtype == MIFWT_F64 ? 28 : 29))) return false;
  for (int i = 0; i < 2; ++i)
    if (d->approx_stride[i] < 0 || d->detail_stride[i] < 0) return false;
  // the kernel maps the boundary with ONE fold (mifwt_stream.h: Fold1): every requested index lies within one period
  // of the plane once the plane is at least as long as the filter; shorter planes take the streaming / generic routes
  if (d->sig_extent[0] < L || d->sig_extent[1] < L) return false;
  // 32-bit element offsets inside one image of a band
  const int64_t lim = int64_t(1) << 31;
  if (d->coef_extent[0] * d->approx_stride[1] >= lim || d->coef_extent[0] * d->detail_stride[1] >= lim) return false;
  return true;
}

int dwt2_fwd_tile(const mifwt_level_desc* d, const void* x, void* approx, void* const* details, const double* lo,
                  const double* hi, hipStream_t stream) {
  if (d->dtype == MIFWT_F64) return dwt2_fwd_tile_f64_short(d, x, approx, details, lo, hi, stream);
  switch (d->filt_len) {
    case 18: return dwt2_fwd_tile_long18(d, x, approx, details, lo, hi, stream);
    case 20: return dwt2_fwd_tile_long20(d, x, approx, details, lo, hi, stream);
    case 24: return dwt2_fwd_tile_long24(d, x, approx, details, lo, hi, stream);
    case 32: return dwt2_fwd_tile_long32(d, x, approx, details, lo, hi, stream);
    default: break;
  }
  return d->dtype == MIFWT_F16 ? dwt2_fwd_tile_f16_short(d, x, approx, details, lo, hi, stream)
                               : dwt2_fwd_tile_f32_short(d, x, approx, details, lo, hi, stream);
}

// Two fused 2-D analysis kernels.  Measured on MI355X (64-image batches, 128^2 .. 4096^2 planes): the LDS-tile kernel
// wins for every L <= 14 (by 4-35 %) and for L = 16 up to ~1500^2 planes; only 16-tap filters on big planes favour
// the streaming kernel (its register ring re-reads no row halo).  f16 storage and 18..32 taps: tile kernel only.
int dwt2_fwd_choice(const mifwt_level_desc* d) {
  const int tm = g_options[MIFWT_OPT_TILE_MODE];  // 0 auto, 1 always tile, 2 never tile
  if (g_options[MIFWT_OPT_MFMA_MODE] != 2 && dwt2_fwd_mfma_supported(d)) return kDwt2FwdMfma;
  if (tm == 0 && g_options[MIFWT_OPT_PYRAMID_MODE] != 2 && d->sig_extent[1] >= 896 && d->sig_extent[1] <= 1280 && d->sig_extent[0] >= 256) {
    // ONE level through the streaming multi-level kernel (id 16): 64 x 1024^2 db4 83.6 against 105.3 us for the tile kernel (db2: 83.6
    // against 98.5; equal at 515^2, behind at 1400^2 and 2048^2: tools/fwd1_probe.py, profiles/r04r_fwd1_probe.txt) — what a
    // single-level mifwt_dwt_fwd call and every synthesis adjoint of such planes now take
    const mifwt_level_desc* dd[1] = {d};
    if (dwt2_fwd_pyr_supported(1, dd)) return kDwt2FwdPyr;
  }
  const bool stream_ok = dwt2_fwd_stream_supported(d), tile_ok = dwt2_fwd_tile_supported(d);
  if (stream_ok && (tm == 2 || !tile_ok)) return kDwt2FwdStream;
  if (tile_ok && tm != 2) {
    const bool big_long = d->filt_len == 16 && d->sig_extent[0] * d->sig_extent[1] >= (int64_t(1) << 21);
    return (tm == 1 || !stream_ok || !big_long) ? kDwt2FwdTile : kDwt2FwdStream;
  }
  return -1;
}

int dwt2_fwd_fused(const mifwt_level_desc* d, const void* x, void* approx, void* const* details, const double* lo,
                   const double* hi, hipStream_t stream) {
  switch (dwt2_fwd_choice(d)) {
    case kDwt2FwdTile: return dwt2_fwd_tile(d, x, approx, details, lo, hi, stream);
    case kDwt2FwdStream:
      if (g_dtaps.lo) return MIFWT_ERR_UNSUPPORTED;  // (device-resident taps: this kernel takes its taps by value only — never silently on zeros)
      return dwt2_fwd_stream(d, x, approx, details, lo, hi, stream);
    case kDwt2FwdPyr: {
      const mifwt_level_desc* dd[1] = {d};
      void* const* dp[1] = {details};
      return dwt2_fwd_pyr(1, dd, x, dp, approx, lo, hi, stream);
    }
    default: return MIFWT_ERR_UNSUPPORTED;
  }
}

// ---- LDS-tile fused 2-D synthesis -------------------------------------------------------------------------------------
int idwt2_tile_f32_short(const mifwt_level_desc*, const void*, const void* const*, void*, const double*, const double*, hipStream_t);
int idwt2_tile_f16_short(const mifwt_level_desc*, const void*, const void* const*, void*, const double*, const double*, hipStream_t);
int idwt2_tile_f64_short(const mifwt_level_desc*, const void*, const void* const*, void*, const double*, const double*, hipStream_t);
int idwt2_tile_long_a(const mifwt_level_desc*, const void*, const void* const*, void*, const double*, const double*, hipStream_t);
int idwt2_tile_long_b(const mifwt_level_desc*, const void*, const void* const*, void*, const double*, const double*, hipStream_t);

bool dwt2_inv_tile_supported(const mifwt_level_desc* d) {
  if (d->ndim != 2 || (d->dtype != MIFWT_F32 && d->dtype != MIFWT_F16 && d->dtype != MIFWT_F64)) return false;
  const int L = d->filt_len;
  if (!((L >= 2 && L <= 20 && (L & 1) == 0) || L == 24 || L == 32)) return false;
  if (d->dtype == MIFWT_F64 && L > 16) return false;  // f64 instantiations: mifwt_idwt2_tile_f64.hip
  if (d->sig_stride[2] != 1 || d->approx_stride[2] != 1 || d->detail_stride[2] != 1) return false;
  for (int i = 0; i < 2; ++i)
    if (d->approx_stride[i] < 0 || d->detail_stride[i] < 0 || d->sig_stride[i] < 0) return false;
  const int64_t span_a = (d->coef_extent[0] - 1) * d->approx_stride[1] + d->coef_extent[1];
  const int64_t span_d = (d->coef_extent[0] - 1) * d->detail_stride[1] + d->coef_extent[1];
  const int64_t lim = int64_t(1) << (d->dtype == MIFWT_F64 ? 28 : 29);  // 32-bit byte offsets below 2^31
  return span_a < lim && span_d < lim;
}

int dwt2_inv_tile(const mifwt_level_desc* d, const void* approx, const void* const* details, void* y, const double* lo,
                  const double* hi, hipStream_t stream) {
  if (d->dtype == MIFWT_F64) return idwt2_tile_f64_short(d, approx, details, y, lo, hi, stream);
  if (d->filt_len == 18 || d->filt_len == 20) return idwt2_tile_long_a(d, approx, details, y, lo, hi, stream);
  if (d->filt_len == 24 || d->filt_len == 32) return idwt2_tile_long_b(d, approx, details, y, lo, hi, stream);
  return d->dtype == MIFWT_F16 ? idwt2_tile_f16_short(d, approx, details, y, lo, hi, stream)
                               : idwt2_tile_f32_short(d, approx, details, y, lo, hi, stream);
}

int dwt2_inv_choice(const mifwt_level_desc* d) {
  const int tm = g_options[MIFWT_OPT_TILE_MODE];  // 0 auto, 1 always tile, 2 never tile
  if (dwt2_inv_mfma_supported(d)) return kDwt2InvMfma;
  if (tm == 0 && g_options[MIFWT_OPT_PYRAMID_MODE] != 2 && d->sig_extent[1] >= 700 && d->sig_extent[1] <= 1536 && d->sig_extent[0] >= 256) {
    // ONE level through the streaming multi-level kernel (id 22): measured ahead of both per-level kernels on planes of about a
    // thousand columns (64 x 1024^2 db4: 80 against 115 us; db2: 81 against 98; 32 x 1000^2 db5: 43 against 53; 16 x 1400^2 db3: 42
    // against 50; equal at 515^2, behind at 2055^2: tools/inv1_probe.py, profiles/r04q_inv1_probe.txt) — what a single-level
    // mifwt_dwt_inv call and the zero-mode part of every analysis adjoint of such planes now take
    const mifwt_level_desc* dd[1] = {d};
    if (dwt2_inv_pyr_supported(1, dd)) return kDwt2InvPyr;
  }
  const bool stream_ok = dwt2_inv_stream_supported(d), tile_ok = dwt2_inv_tile_supported(d);
  if (stream_ok && (tm == 2 || !tile_ok)) return kDwt2InvStream;
  if (tile_ok && tm != 2) {
    // measured (MI355X, 64-image batches): the tile kernel wins on planes below ~1000^2 (515^2: 33.8 vs 37.1 us) and
    // for 2- / 4-tap filters (haar 1024^2: 98 vs 109 us); from 1024^2 on the streaming kernel is equal (db4) or
    // better (db8 4096^2: 1.79 vs 2.11 ms)
    const bool big = d->sig_extent[0] * d->sig_extent[1] >= (int64_t(1) << 20) && d->filt_len > 4;
    return (tm == 1 || !stream_ok || !big) ? kDwt2InvTile : kDwt2InvStream;
  }
  return -1;
}

int dwt2_inv_fused(const mifwt_level_desc* d, const void* approx, const void* const* details, void* y, const double* lo,
                   const double* hi, hipStream_t stream) {
  switch (dwt2_inv_choice(d)) {
    case kDwt2InvTile: return dwt2_inv_tile(d, approx, details, y, lo, hi, stream);
    case kDwt2InvStream:
      if (g_dtaps.lo) return MIFWT_ERR_UNSUPPORTED;  // (device-resident taps: these kernels take their taps by value only)
      return dwt2_inv_stream(d, approx, details, y, lo, hi, stream);
    case kDwt2InvMfma:
      if (g_dtaps.lo) return MIFWT_ERR_UNSUPPORTED;
      return dwt2_inv_mfma(d, approx, details, y, lo, hi, stream);
    case kDwt2InvPyr: {
      const mifwt_level_desc* dd[1] = {d};
      const void* const* dp[1] = {details};
      return dwt2_inv_pyr(1, dd, approx, dp, y, lo, hi, stream);
    }
    default: return MIFWT_ERR_UNSUPPORTED;
  }
}

bool rows_route_ok(const mifwt_level_desc* d, int direction) {
  (void)direction;
  if (!stream_filter_supported(d->filt_len)) return false;
  const int nd = d->ndim;
  if (d->sig_stride[nd] != 1 || d->approx_stride[nd] != 1 || d->detail_stride[nd] != 1) return false;
  for (int i = 0; i <= nd; ++i)
    if (d->sig_stride[i] < 0 || d->approx_stride[i] < 0 || d->detail_stride[i] < 0) return false;
  int64_t rows = d->batch;
  for (int a = 0; a < nd - 1; ++a) rows *= d->sig_extent[a] > d->coef_extent[a] ? d->sig_extent[a] : d->coef_extent[a];
  if (rows > INT32_MAX / 4) return false;
  // the last analysis pass writes / the first synthesis pass reads the caller's band tensors as
  // [batch, axis 0, dense run of the remaining axes]
  if (nd >= 2 && (!dense_under(d->approx_stride, d->coef_extent, nd, 1) || !dense_under(d->detail_stride, d->coef_extent, nd, 1)))
    return false;
  return true;
}

bool plane3_route_ok(const mifwt_level_desc* d, int direction) {
  // (f32, and since round 5 f64: the tile kernels compute doubles, the depth pass is the streaming axis kernel of either precision —
  // two passes over the volume instead of the three single-axis passes f64 volumes took until then)
  if (d->ndim != 3 || (d->dtype != MIFWT_F32 && d->dtype != MIFWT_F64) || !rows_route_ok(d, direction)) return false;
  bool foldable;
  const mifwt_level_desc p = plane_desc(d, d->sig_extent[0], &foldable);
  return direction == 0 ? dwt2_fwd_choice(&p) >= 0 : dwt2_inv_choice(&p) >= 0;
}

size_t plane3_ws_bytes(const mifwt_level_desc* d, int direction) {
  (void)direction;  // analysis: [B, D, 4, Ho, Wo];  synthesis: [B, Dout, 4, Ho, Wo]
  return (size_t)(4 * d->batch * d->sig_extent[0] * d->coef_extent[1] * d->coef_extent[2]) * (size_t)elem_size(d->dtype);
}

size_t rows_ws_bytes(const mifwt_level_desc* d, int direction) {
  const int nd = d->ndim;
  int64_t total = 0;
  if (direction == 0) {
    // after transforming axes a..nd-1 (innermost first): 2^(nd-a) arrays [B, sig(0..a-1), coef(a..nd-1)]
    for (int a = nd - 1; a >= 1; --a) {
      int64_t e = d->batch;
      for (int i = 0; i < nd; ++i) e *= i >= a ? d->coef_extent[i] : d->sig_extent[i];
      total += e << (nd - a);
    }
  } else {
    // after expanding axes 0..a (outermost first): 2^(nd-1-a) arrays [B, sig(0..a), coef(a+1..nd-1)]
    for (int a = 0; a < nd - 1; ++a) {
      int64_t e = d->batch;
      for (int i = 0; i < nd; ++i) e *= i <= a ? d->sig_extent[i] : d->coef_extent[i];
      total += e << (nd - 1 - a);
    }
  }
  return (size_t)(total * elem_size(d->dtype));
}

// ---- ndim 3, f32 ----------------------------------------------------------------------------------------
int plane3_fwd(const mifwt_level_desc* d, const void* x, void* approx, void* const* details, const double* lo,
               const double* hi, void* ws, hipStream_t stream) {
  const int64_t D = d->sig_extent[0], plane = d->coef_extent[1] * d->coef_extent[2];
  const int64_t esz = elem_size(d->dtype);
  char* scratch = static_cast<char*>(ws);  // [B, D, 4, Ho, Wo] elements of the level's dtype (byte arithmetic below)
  bool foldable;
  const mifwt_level_desc p = plane_desc(d, D, &foldable);
  mifwt_level_desc pall = p;  // every slice of every volume in ONE launch: a two-level batch of the tile kernel's input
  pall.batch = d->batch * D;
  if (!foldable && dwt2_fwd_tile_supported(&pall) && d->batch * D < (int64_t(1) << 31)) {
    // (the input of a deeper level is plane 0 of the previous level's [B, 8, D, H, W] buffer: volumes 8 D H W apart, slices H W apart —
    // one launch per volume cost 32 launches of 5 us each on 32 x 54^3, profiles/r03h_kernel_trace_refshapes.txt)
    void* det[3] = {scratch + plane * esz, scratch + 2 * plane * esz, scratch + 3 * plane * esz};
    g_batch_split = {D, d->sig_stride[0]};
    const int rc = dwt2_fwd_tile(&pall, x, scratch, det, lo, hi, stream);
    g_batch_split = {0, 0};
    if (rc != MIFWT_OK) return rc;
  } else {
    const int64_t nb = foldable ? 1 : d->batch;
    for (int64_t b = 0; b < nb; ++b) {
      const char* xb = static_cast<const char*>(x) + b * d->sig_stride[0] * esz;
      char* sb = scratch + b * D * 4 * plane * esz;
      void* det[3] = {sb + plane * esz, sb + 2 * plane * esz, sb + 3 * plane * esz};
      const int rc = dwt2_fwd_fused(&p, xb, sb, det, lo, hi, stream);
      if (rc != MIFWT_OK) return rc;
    }
  }
  StreamJob jobs[4];
  for (int s = 0; s < 4; ++s) {  // plane band s (axes H, W) -> bands s (depth low) and 4 + s (depth high)
    StreamJob& j = jobs[s];
    memset(&j, 0, sizeof(j));
    j.in0 = scratch + s * plane * esz;
    set3(j.in0_s, D * 4 * plane, 4 * plane, 0);
    j.out0 = s == 0 ? approx : details[s - 1];
    j.out1 = details[s + 4 - 1];
    const int64_t* os = s == 0 ? d->approx_stride : d->detail_stride;
    set3(j.out0_s, os[0], os[1], 0);
    set3(j.out1_s, d->detail_stride[0], d->detail_stride[1], 0);
  }
  StreamCall c;
  memset(&c, 0, sizeof(c));
  c.filt_len = d->filt_len;
  c.mode = d->mode;
  c.jobs = jobs;
  c.njobs = 4;
  c.batch = d->batch;
  c.n_in = D;
  c.n_out = d->coef_extent[0];
  c.inner = plane;
  c.lo = lo;
  c.hi = hi;
  c.stream = stream;
  return stream_call(d->dtype, kOuterFwd, c);
}

int plane3_inv(const mifwt_level_desc* d, const void* approx, const void* const* details, void* y, const double* lo,
               const double* hi, void* ws, hipStream_t stream) {
  const int64_t Dout = d->sig_extent[0], plane = d->coef_extent[1] * d->coef_extent[2];
  const int64_t esz = elem_size(d->dtype);
  char* scratch = static_cast<char*>(ws);  // [B, Dout, 4, Ho, Wo] elements of the level's dtype
  StreamJob jobs[4];
  for (int s = 0; s < 4; ++s) {
    StreamJob& j = jobs[s];
    memset(&j, 0, sizeof(j));
    j.in0 = s == 0 ? approx : details[s - 1];
    j.in1 = details[s + 4 - 1];
    const int64_t* is = s == 0 ? d->approx_stride : d->detail_stride;
    set3(j.in0_s, is[0], is[1], 0);
    set3(j.in1_s, d->detail_stride[0], d->detail_stride[1], 0);
    j.out0 = scratch + s * plane * esz;
    set3(j.out0_s, Dout * 4 * plane, 4 * plane, 0);
  }
  StreamCall c;
  memset(&c, 0, sizeof(c));
  c.filt_len = d->filt_len;
  c.jobs = jobs;
  c.njobs = 4;
  c.batch = d->batch;
  c.n_in = d->coef_extent[0];
  c.n_out = Dout;
  c.inner = plane;
  c.lo = lo;
  c.hi = hi;
  c.stream = stream;
  int rc = stream_call(d->dtype, kOuterInv, c);
  if (rc != MIFWT_OK) return rc;
  bool foldable;
  const mifwt_level_desc p = plane_desc(d, Dout, &foldable);
  const int64_t nb = foldable ? 1 : d->batch;
  for (int64_t b = 0; b < nb; ++b) {
    char* yb = static_cast<char*>(y) + b * d->sig_stride[0] * esz;
    const char* sb = scratch + b * Dout * 4 * plane * esz;
    const void* det[3] = {sb + plane * esz, sb + 2 * plane * esz, sb + 3 * plane * esz};
    rc = dwt2_inv_fused(&p, sb, det, yb, lo, hi, stream);
    if (rc != MIFWT_OK) return rc;
  }
  return MIFWT_OK;
}

// ---- any ndim / dtype: inner-axis pass + one outer-axis pass per further axis ------------------------------
int rows_fwd(const mifwt_level_desc* d, const void* x, void* approx, void* const* details, const double* lo,
             const double* hi, void* ws, hipStream_t stream) {
  const int nd = d->ndim;
  const int64_t esz = elem_size(d->dtype);
  char* wsp = static_cast<char*>(ws);
  const void* cur[8];  // current arrays, indexed by partial band bits; strides for (batch, axis 0.., axis nd-1)
  int64_t cur_stride[8][4];
  int ncur = 1;
  cur[0] = x;
  for (int i = 0; i <= nd; ++i) cur_stride[0][i] = d->sig_stride[i];
  StreamCall c;
  memset(&c, 0, sizeof(c));
  c.filt_len = d->filt_len;
  c.mode = d->mode;
  c.lo = lo;
  c.hi = hi;
  c.stream = stream;
  for (int pass = 0; pass < nd; ++pass) {
    const int a = nd - 1 - pass;        // axis transformed in this pass (innermost first)
    const int bit = 1 << (nd - 1 - a);  // its bit in the band index
    const bool last = a == 0;
    int64_t ext[4] = {d->batch, 1, 1, 1};  // extents of the arrays this pass WRITES
    for (int i = 0; i < nd; ++i) ext[1 + i] = i >= a ? d->coef_extent[i] : d->sig_extent[i];
    int64_t dense[4], elems = 1;
    for (int i = nd; i >= 0; --i) {
      dense[i] = elems;
      elems *= ext[i];
    }
    void* nxt[8];
    int64_t nxt_stride[8][4];
    for (int s = 0; s < ncur; ++s)
      for (int h = 0; h < 2; ++h) {
        const int band = s | (h ? bit : 0);
        if (last) {
          nxt[band] = band == 0 ? approx : details[band - 1];
          for (int i = 0; i <= nd; ++i) nxt_stride[band][i] = band == 0 ? d->approx_stride[i] : d->detail_stride[i];
        } else {
          nxt[band] = wsp;
          wsp += elems * esz;
          for (int i = 0; i <= nd; ++i) nxt_stride[band][i] = dense[i];
        }
      }
    for (int s0 = 0; s0 < ncur; s0 += 4) {
      StreamJob jobs[4];
      const int nj = ncur - s0 < 4 ? ncur - s0 : 4;
      for (int j = 0; j < nj; ++j) {
        const int s = s0 + j;
        StreamJob& jb = jobs[j];
        memset(&jb, 0, sizeof(jb));
        jb.in0 = cur[s];
        jb.out0 = nxt[s];
        jb.out1 = nxt[s | bit];
        if (a == nd - 1) {  // inner pass: row dims = (batch, axes 0..nd-2), padded to three
          for (int i = 0; i < 3; ++i) {
            const bool used = i < nd;
            jb.in0_s[i] = used ? cur_stride[s][i] : 0;
            jb.out0_s[i] = used ? nxt_stride[s][i] : 0;
            jb.out1_s[i] = used ? nxt_stride[s | bit][i] : 0;
          }
        } else {
          // outer pass over axis a: batch = (batch, axes < a) folded — every array it touches is dense scratch
          // (or, for a == 0, addressed by its own batch stride); inner = dense run of axes > a
          const int hi_band = s | bit;
          set3(jb.in0_s, a == 0 ? cur_stride[s][0] : d->sig_extent[a] * cur_stride[s][1 + a], cur_stride[s][1 + a], 0);
          set3(jb.out0_s, a == 0 ? nxt_stride[s][0] : ext[1 + a] * nxt_stride[s][1 + a], nxt_stride[s][1 + a], 0);
          set3(jb.out1_s, a == 0 ? nxt_stride[hi_band][0] : ext[1 + a] * nxt_stride[hi_band][1 + a], nxt_stride[hi_band][1 + a], 0);
        }
      }
      c.jobs = jobs;
      c.njobs = nj;
      c.n_in = d->sig_extent[a];
      c.n_out = d->coef_extent[a];
      int rc;
      if (a == nd - 1) {
        for (int i = 0; i < 3; ++i) c.rows[i] = i < nd ? (i == 0 ? d->batch : d->sig_extent[i - 1]) : 1;
        rc = stream_call(d->dtype, kInnerFwd, c);
      } else {
        c.batch = d->batch;
        for (int i = 0; i < a; ++i) c.batch *= d->sig_extent[i];
        c.inner = 1;
        for (int i = a + 1; i < nd; ++i) c.inner *= d->coef_extent[i];
        rc = stream_call(d->dtype, kOuterFwd, c);
      }
      if (rc != MIFWT_OK) return rc;
    }
    ncur *= 2;
    for (int s = 0; s < ncur; ++s) {
      cur[s] = nxt[s];
      memcpy(cur_stride[s], nxt_stride[s], sizeof(int64_t) * 4);
    }
  }
  return MIFWT_OK;
}

int rows_inv(const mifwt_level_desc* d, const void* approx, const void* const* details, void* y, const double* lo,
             const double* hi, void* ws, hipStream_t stream) {
  const int nd = d->ndim;
  const int64_t esz = elem_size(d->dtype);
  const int nb = 1 << nd;
  const void* cur[8];
  int64_t cur_stride[8][4];
  for (int s = 0; s < nb; ++s) {
    cur[s] = s == 0 ? approx : details[s - 1];
    for (int i = 0; i <= nd; ++i) cur_stride[s][i] = s == 0 ? d->approx_stride[i] : d->detail_stride[i];
  }
  char* wsp = static_cast<char*>(ws);
  StreamCall c;
  memset(&c, 0, sizeof(c));
  c.filt_len = d->filt_len;
  c.lo = lo;
  c.hi = hi;
  c.stream = stream;
  int ncur = nb;
  for (int a = 0; a < nd; ++a) {  // outermost axis first: its bit is the most significant of the remaining
    const bool last = a == nd - 1;
    const int nnext = ncur / 2;            // arrays pair up as (s, s + nnext): low / high along axis a
    int64_t ext[4] = {d->batch, 1, 1, 1};  // extents of the arrays this pass WRITES
    for (int i = 0; i < nd; ++i) ext[1 + i] = i <= a ? d->sig_extent[i] : d->coef_extent[i];
    int64_t dense[4], elems = 1;
    for (int i = nd; i >= 0; --i) {
      dense[i] = elems;
      elems *= ext[i];
    }
    void* nxt[4];
    int64_t nxt_stride[4][4];
    for (int s = 0; s < nnext; ++s) {
      if (last) {
        nxt[s] = y;
        for (int i = 0; i <= nd; ++i) nxt_stride[s][i] = d->sig_stride[i];
      } else {
        nxt[s] = wsp;
        wsp += elems * esz;
        for (int i = 0; i <= nd; ++i) nxt_stride[s][i] = dense[i];
      }
    }
    StreamJob jobs[4];
    for (int s = 0; s < nnext; ++s) {
      StreamJob& jb = jobs[s];
      memset(&jb, 0, sizeof(jb));
      jb.in0 = cur[s];
      jb.in1 = cur[s + nnext];
      jb.out0 = nxt[s];
      if (last) {
        for (int i = 0; i < 3; ++i) {
          const bool used = i < nd;
          jb.in0_s[i] = used ? cur_stride[s][i] : 0;
          jb.in1_s[i] = used ? cur_stride[s + nnext][i] : 0;
          jb.out0_s[i] = used ? nxt_stride[s][i] : 0;
        }
      } else {
        // batch = (batch, axes < a) folded: for a > 0 every array is dense scratch with signal extents there
        const int64_t* s0 = cur_stride[s];
        const int64_t* s1 = cur_stride[s + nnext];
        set3(jb.in0_s, a == 0 ? s0[0] : d->coef_extent[a] * s0[1 + a], s0[1 + a], 0);
        set3(jb.in1_s, a == 0 ? s1[0] : d->coef_extent[a] * s1[1 + a], s1[1 + a], 0);
        set3(jb.out0_s, a == 0 ? nxt_stride[s][0] : ext[1 + a] * nxt_stride[s][1 + a], nxt_stride[s][1 + a], 0);
      }
    }
    c.jobs = jobs;
    c.njobs = nnext;
    c.n_in = d->coef_extent[a];
    c.n_out = d->sig_extent[a];
    int rc;
    if (last) {
      for (int i = 0; i < 3; ++i) c.rows[i] = i < nd ? (i == 0 ? d->batch : d->sig_extent[i - 1]) : 1;
      rc = stream_call(d->dtype, kInnerInv, c);
    } else {
      c.batch = d->batch;
      for (int i = 0; i < a; ++i) c.batch *= d->sig_extent[i];
      c.inner = 1;
      for (int i = a + 1; i < nd; ++i) c.inner *= d->coef_extent[i];
      rc = stream_call(d->dtype, kOuterInv, c);
    }
    if (rc != MIFWT_OK) return rc;
    ncur = nnext;
    for (int s = 0; s < ncur; ++s) {
      cur[s] = nxt[s];
      memcpy(cur_stride[s], nxt_stride[s], sizeof(int64_t) * 4);
    }
  }
  return MIFWT_OK;
}

}  // namespace mifwt
