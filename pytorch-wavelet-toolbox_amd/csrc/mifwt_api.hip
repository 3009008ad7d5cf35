// mifwt_api.hip — the C-ABI of libmifwt.so (declared in include/mifwt.h) and its dispatcher.
//
// One call = one decomposition / reconstruction level over the folded batch.  The dispatcher picks a
// fused single-launch kernel when the descriptor is inside a fast path's envelope and otherwise
// assembles the level from generic per-axis passes through caller-provided scratch.
#include <string.h>

#include "mifwt_common.h"

using namespace mifwt;

namespace {

inline int64_t elem_size(int dtype) { return dtype == MIFWT_F64 ? 8 : (dtype == MIFWT_F16 ? 2 : 4); }

int validate(const mifwt_level_desc* d, int direction) {
  if (!d) return MIFWT_ERR_BADARG;
  if (d->ndim < 1 || d->ndim > MIFWT_MAX_NDIM) return MIFWT_ERR_BADARG;
  if (d->dtype != MIFWT_F32 && d->dtype != MIFWT_F64 && d->dtype != MIFWT_F16) return MIFWT_ERR_BADARG;
  if (d->filt_len < 2 || d->filt_len > MIFWT_MAX_FILT) return MIFWT_ERR_BADARG;
  if (d->batch < 0) return MIFWT_ERR_BADARG;
  if (direction == 0 && (d->mode < MIFWT_MODE_ZERO || d->mode > MIFWT_MODE_SYMMETRIC)) return MIFWT_ERR_BADARG;
  for (int a = 0; a < d->ndim; ++a) {
    const int64_t n = d->sig_extent[a], m = d->coef_extent[a], L = d->filt_len;
    if (n < 1 || m < 1 || n > INT32_MAX / 4 || m > INT32_MAX / 4) return MIFWT_ERR_BADARG;
    if (direction == 0) {
      // conv output length of the padded signal (reference src/ptwt/_util.py:204-217)
      const int64_t pad = 2 * ((2 * L - 3) / 2) + (n % 2);
      if (m != (n + pad - L) / 2 + 1) return MIFWT_ERR_BADARG;
    } else {
      const int64_t full = 2 * m - L + 2;  // after cropping L-2 on both sides
      if (n != full && n != full - 1) return MIFWT_ERR_BADARG;
      if (n < 1) return MIFWT_ERR_BADARG;
    }
  }
  return MIFWT_OK;
}

// scratch layout of the generic N-D path: stage s holds 2^(s+1) (analysis) arrays, dense, batch-major.
struct Stage {
  int64_t ext[4];    // (batch, axis0, axis1, axis2) extents of each array of this stage
  int64_t stride[4];
  int64_t elems;     // per array
  int narrays;
};

// analysis: axes are transformed innermost first; after transforming axes a..ndim-1 the arrays have
// coefficient extents on those axes and signal extents on the rest.
int plan_fwd(const mifwt_level_desc* d, Stage st[MIFWT_MAX_NDIM]) {
  int ns = 0;
  for (int a = d->ndim - 1; a >= 1; --a) {  // the last pass (axis 0) writes the final bands
    Stage& s = st[ns++];
    s.ext[0] = d->batch;
    for (int i = 0; i < 3; ++i) s.ext[1 + i] = i < d->ndim ? (i >= a ? d->coef_extent[i] : d->sig_extent[i]) : 1;
    s.elems = 1;
    for (int i = 3; i >= 0; --i) {
      s.stride[i] = s.elems;
      s.elems *= s.ext[i];
    }
    s.narrays = 1 << (d->ndim - a);
  }
  return ns;
}

// synthesis: axes are expanded outermost first.
int plan_inv(const mifwt_level_desc* d, Stage st[MIFWT_MAX_NDIM]) {
  int ns = 0;
  for (int a = 0; a < d->ndim - 1; ++a) {  // the last pass (innermost axis) writes y
    Stage& s = st[ns++];
    s.ext[0] = d->batch;
    for (int i = 0; i < 3; ++i) s.ext[1 + i] = i < d->ndim ? (i <= a ? d->sig_extent[i] : d->coef_extent[i]) : 1;
    s.elems = 1;
    for (int i = 3; i >= 0; --i) {
      s.stride[i] = s.elems;
      s.elems *= s.ext[i];
    }
    s.narrays = 1 << (d->ndim - 1 - a);
  }
  return ns;
}

inline void pad_strides(const int64_t src[1 + MIFWT_MAX_NDIM], int ndim, int64_t dst[4]) {
  for (int i = 0; i < 4; ++i) dst[i] = i <= ndim ? src[i] : 0;
}

int generic_fwd(const mifwt_level_desc* d, const void* x, void* approx, void* const* details, const double* lo,
                const double* hi, void* ws, hipStream_t stream) {
  Stage st[MIFWT_MAX_NDIM];
  const int ns = plan_fwd(d, st);
  const int64_t esz = elem_size(d->dtype);
  const int nd = d->ndim;
  // current arrays, indexed by partial band bits
  const void* cur[8];
  int64_t cur_stride[8][4];
  int ncur = 1;
  cur[0] = x;
  pad_strides(d->sig_stride, nd, cur_stride[0]);
  char* wsp = static_cast<char*>(ws);
  for (int pass = 0; pass < nd; ++pass) {
    const int a = nd - 1 - pass;           // axis transformed in this pass
    const int bit = 1 << (nd - 1 - a);     // its bit in the band index
    const bool last = (a == 0);
    void* nxt[8];
    int64_t nxt_stride[8][4];
    int64_t out_ext[4] = {d->batch, 1, 1, 1};
    for (int i = 0; i < nd; ++i) out_ext[1 + i] = i >= a ? d->coef_extent[i] : d->sig_extent[i];
    for (int s = 0; s < ncur; ++s) {
      for (int h = 0; h < 2; ++h) {
        const int band = s | (h ? bit : 0);
        if (last) {
          nxt[band] = band == 0 ? approx : details[band - 1];
          pad_strides(band == 0 ? d->approx_stride : d->detail_stride, nd, nxt_stride[band]);
        } else {
          nxt[band] = wsp;
          wsp += st[pass].elems * esz;
          memcpy(nxt_stride[band], st[pass].stride, sizeof(int64_t) * 4);
        }
      }
    }
    for (int s0 = 0; s0 < ncur; s0 += 4) {
      AxisJob jobs[4];
      const int nj = ncur - s0 < 4 ? ncur - s0 : 4;
      for (int j = 0; j < nj; ++j) {
        const int s = s0 + j;
        AxisJob& jb = jobs[j];
        memset(&jb, 0, sizeof(jb));
        jb.in0 = cur[s];
        memcpy(jb.in0_stride, cur_stride[s], sizeof(int64_t) * 4);
        jb.out0 = nxt[s];
        jb.out1 = nxt[s | bit];
        memcpy(jb.out0_stride, nxt_stride[s], sizeof(int64_t) * 4);
        memcpy(jb.out1_stride, nxt_stride[s | bit], sizeof(int64_t) * 4);
      }
      const int rc = launch_axis_fwd(d->dtype, jobs, nj, out_ext, 1 + a, d->sig_extent[a], d->mode, d->filt_len, lo,
                                     hi, stream);
      if (rc != MIFWT_OK) return rc;
    }
    ncur *= 2;
    // band indices produced so far are all combinations of the bits of axes >= a: compact is not
    // needed because bits are assigned from the least significant end.
    for (int s = 0; s < ncur; ++s) {
      cur[s] = nxt[s];
      memcpy(cur_stride[s], nxt_stride[s], sizeof(int64_t) * 4);
    }
  }
  return MIFWT_OK;
}

// adjoint == true: the transpose of the generic ANALYSIS of this descriptor (same pass structure as the
// synthesis: band pairs in, signal-extent arrays out; lo / hi are then the DEC taps and the halo is folded back).
int generic_inv(const mifwt_level_desc* d, const void* approx, const void* const* details, void* y,
                const double* lo, const double* hi, void* ws, hipStream_t stream, bool adjoint = false) {
  Stage st[MIFWT_MAX_NDIM];
  plan_inv(d, st);
  const int64_t esz = elem_size(d->dtype);
  const int nd = d->ndim;
  const int nb = 1 << nd;
  const void* cur[8];
  int64_t cur_stride[8][4];
  for (int s = 0; s < nb; ++s) {
    cur[s] = s == 0 ? approx : details[s - 1];
    pad_strides(s == 0 ? d->approx_stride : d->detail_stride, nd, cur_stride[s]);
  }
  char* wsp = static_cast<char*>(ws);
  int ncur = nb;
  for (int a = 0; a < nd; ++a) {  // outermost axis first: its bit is the most significant of the remaining
    const bool last = (a == nd - 1);
    const int nnext = ncur / 2;   // arrays pair up as (s, s + nnext): low / high along axis a
    int64_t out_ext[4] = {d->batch, 1, 1, 1};
    for (int i = 0; i < nd; ++i) out_ext[1 + i] = i <= a ? d->sig_extent[i] : d->coef_extent[i];
    void* nxt[4];
    int64_t nxt_stride[4][4];
    for (int s = 0; s < nnext; ++s) {
      if (last) {
        nxt[s] = y;
        pad_strides(d->sig_stride, nd, nxt_stride[s]);
      } else {
        nxt[s] = wsp;
        wsp += st[a].elems * esz;
        memcpy(nxt_stride[s], st[a].stride, sizeof(int64_t) * 4);
      }
    }
    AxisJob jobs[4];
    for (int s = 0; s < nnext; ++s) {
      AxisJob& jb = jobs[s];
      memset(&jb, 0, sizeof(jb));
      jb.in0 = cur[s];
      jb.in1 = cur[s + nnext];
      memcpy(jb.in0_stride, cur_stride[s], sizeof(int64_t) * 4);
      memcpy(jb.in1_stride, cur_stride[s + nnext], sizeof(int64_t) * 4);
      jb.out0 = nxt[s];
      memcpy(jb.out0_stride, nxt_stride[s], sizeof(int64_t) * 4);
    }
    const int rc = adjoint ? launch_axis_adj(d->dtype, jobs, nnext, out_ext, 1 + a, d->coef_extent[a], d->sig_extent[a],
                                             d->mode, d->filt_len, lo, hi, stream)
                           : launch_axis_inv(d->dtype, jobs, nnext, out_ext, 1 + a, d->coef_extent[a], d->filt_len, lo, hi, stream);
    if (rc != MIFWT_OK) return rc;
    ncur = nnext;
    for (int s = 0; s < ncur; ++s) {
      cur[s] = nxt[s];
      memcpy(cur_stride[s], nxt_stride[s], sizeof(int64_t) * 4);
    }
  }
  return MIFWT_OK;
}

size_t generic_ws(const mifwt_level_desc* d, int direction) {
  Stage st[MIFWT_MAX_NDIM];
  const int ns = direction == 0 ? plan_fwd(d, st) : plan_inv(d, st);
  int64_t total = 0;
  for (int i = 0; i < ns; ++i) total += st[i].elems * st[i].narrays;
  return (size_t)(total * elem_size(d->dtype));
}

size_t route_ws(const mifwt_level_desc* d, int direction, int kid) {
  switch (kid) {
    case kDwt2FwdPyr:
    case kDwt2FwdStream:
    case kDwt2FwdTile:
    case kDwt2FwdMfma:
    case kDwt2InvTile:
    case kDwt2InvMfma:
    case kDwt3FwdTile:
    case kDwt3FwdWalk:
    case kDwt3InvWalk:
    case kDwt3InvTile:
    case kDwt2InvPyr:
    case kDwt2InvStream: return 0;
    case kDwt3FwdStream:
    case kDwt3InvStream: return plane3_ws_bytes(d, direction);
    case kDwt1FwdRow:
    case kDwt1InvRow: return rows_ws_bytes(d, direction);
    default: return generic_ws(d, direction);
  }
}

int pick_kernel(const mifwt_level_desc* d, int direction) {
  if (g_options[MIFWT_OPT_FORCE_GENERIC]) return kGeneric;
  if (direction == 0) {
    {
      const int k2 = dwt2_fwd_choice(d);
      if (k2 >= 0) return k2;
    }
    // 3-D: the depth-walking kernel on volumes from ~4 M samples on (config 3: 256^3 224 against 292 us; 129^3 and 66^3 are latency-bound
    // either way and the bricks are 1-4 us ahead), and for 8 taps (no bricks: 16 x 128^3 db4 144 against 171 us on the composed route)
    if (dwt3_fwd_walk_supported(d)) {
      const int tm = g_options[MIFWT_OPT_TILE_MODE];
      const int64_t vol = d->sig_extent[0] * d->sig_extent[1] * d->sig_extent[2];
      // (8 taps: from ~1 M samples on — the one measurement is 16 x 128^3; small 8-tap volumes such as the deep levels of a decomposition
      // are latency-bound persistent workgroups there and take the composed route instead)
      if (tm == 4 || (tm == 0 && ((d->filt_len <= 6 && vol >= (int64_t(1) << 22)) || (d->filt_len == 8 && vol >= (int64_t(1) << 20))))) return kDwt3FwdWalk;
      // (those thresholds were measured with config 3's 8 volumes; 16 volumes and more fill the chip from smaller ones on.  GPU time of a
      // level from kernel traces, 32 volumes, walk against bricks / composed, us — db2 100^3 / 51^3 / 27^3: 63.6 / 19.0 / 18.0 against 79.4 /
      // 24.5 / 10.5; db3 100^3 / 52^3 / 28^3: 90.2 / 26.2 / 23.5 against 116.4 / 33.9 / 14.1; db4 100^3 / 53^3 / 30^3: 133.7 / 55.4 / 22.3
      // against 133.9 / 31.9 / 18.6; db5: 199.7 / 75.5 / 44.0 against 134.9 / 38.6 / 20.3 — profiles/r06k_walk3_routes.txt)
      if (tm == 0 && d->dtype == MIFWT_F32 && d->batch >= 16 && d->filt_len <= 6 && vol >= 100000) return kDwt3FwdWalk;
      // (and config 3's second level, 8 x 129^3 db2: 34.4 against 40.4 us on the bricks — 48 us inside the pyramid, where its input is a
      // plane of the first level's buffer; 8 x 66^3: 17.2 against 13.8: profiles/r06y_walk3_routes.txt)
      if (tm == 0 && d->dtype == MIFWT_F32 && d->batch >= 8 && d->filt_len <= 6 && vol >= (int64_t(1) << 21)) return kDwt3FwdWalk;
      // eight / ten taps on rows of at most 128 samples: the slab form of the walk where its cost model says so
      if (tm == 0 && !(g_options[MIFWT_OPT_DEBUG] & 2097152) && dwt3_fwd_slab_pays(d)) return kDwt3FwdWalk;
      // f64 has no bricks: the walk kernel against the composed route (planes + depth pass) — 8 x 256^3 / 129^3 / 66^3 / 40^3, us: db2 444 / 69 /
      // 17 / 14 against 945 / 158 / 25 / 25; db3 547 / 91 / 22 / 20 against 947 / 166 / 34 / 18; db4 733 / 155 / 35 / 29 against 1019 / 170 /
      // 28 / 27; db5 1033 / 225 / 63 / 38 against 979 / 184 / 40 / 21 (profiles/r05w_f64_walk_vs_planes.txt)
      if (tm == 0 && d->dtype == MIFWT_F64 && d->filt_len <= 8 && vol >= (int64_t(1) << (d->filt_len <= 4 ? 15 : d->filt_len == 6 ? 17 : 19)))
        return kDwt3FwdWalk;
    }
    if (g_options[MIFWT_OPT_TILE_MODE] != 2 && dwt3_fwd_tile_supported(d)) return kDwt3FwdTile;
    if (plane3_route_ok(d, 0)) return kDwt3FwdStream;
    if (rows_route_ok(d, 0)) return kDwt1FwdRow;
  } else {
    {
      const int k2 = dwt2_inv_choice(d);
      if (k2 >= 0) return k2;
    }
    // 3-D: the depth-walking kernel from ~1 M output samples on (config 3: 256^3 197 against 238 us, 129^3 40-50 against 64 us; 66^3 24
    // against 12 us on the bricks)
    if (dwt3_inv_walk_supported(d)) {
      const int tm = g_options[MIFWT_OPT_TILE_MODE];
      const int64_t vol = d->sig_extent[0] * d->sig_extent[1] * d->sig_extent[2];
      if (tm == 4 || (tm == 0 && vol >= (int64_t(1) << 20))) return kDwt3InvWalk;
      // f64 (no bricks), same file: db2 421 / 66 / 17 / 17 against 926 / 152 / 24 / 14; db3 592 / 102 / 29 / 28 against 942 / 163 / 28 / 18;
      // db4 — / 145 / 44 / 30 against 981 / 171 / 27 / 15 (a row group's pieces of 256^3 exceed 5 KiB)
      if (tm == 0 && d->dtype == MIFWT_F64 && vol >= (int64_t(1) << (d->filt_len <= 4 ? 16 : 19))) return kDwt3InvWalk;
    }
    if (g_options[MIFWT_OPT_TILE_MODE] != 2 && dwt3_inv_tile_supported(d)) return kDwt3InvTile;
    if (plane3_route_ok(d, 1)) return kDwt3InvStream;
    if (rows_route_ok(d, 1)) return kDwt1InvRow;
  }
  return kGeneric;
}

}  // namespace

namespace mifwt {
extern unsigned long long* g_pyr_prof;
int g_options[16] = {0};
unsigned long long g_launch_counts[16] = {0};
thread_local BatchSplit g_batch_split = {0, 0};
}

extern "C" {

int mifwt_set_option(int key, int value) {
  if (key < 0 || key >= 16) return MIFWT_ERR_BADARG;
  // the measurement switches that break results and the experiment word exist in -DMIFWT_DIAG builds only (mifwt_common.h): the product
  // build says so instead of measuring something else
  if (!kDiag && ((key == MIFWT_OPT_DEBUG && (value & ~kRouteBits)) || (key == MIFWT_OPT_EXP && value != 0))) return MIFWT_ERR_UNSUPPORTED;
  g_options[key] = value;
  return MIFWT_OK;
}

int mifwt_abi_version(void) { return MIFWT_ABI_VERSION; }

unsigned long long mifwt_launch_count(int variant) {
  return variant >= 0 && variant < 16 ? __atomic_load_n(&g_launch_counts[variant], __ATOMIC_RELAXED) : 0ull;
}

// diagnostic: device buffer of 2 x uint64 per wave of every workgroup of mifwt_dwt2_fwd_pyramid launches (NULL = off)
int mifwt_pyr_profile_buffer(void* device_buffer) {
  if (!kDiag) return device_buffer ? MIFWT_ERR_UNSUPPORTED : MIFWT_OK;  // (the profiling instances are compiled in -DMIFWT_DIAG builds only)
  g_pyr_prof = static_cast<unsigned long long*>(device_buffer);
  return MIFWT_OK;
}

const char* mifwt_strerror(int code) {
  switch (code) {
    case MIFWT_OK: return "ok";
    case MIFWT_ERR_BADARG: return "bad argument (null pointer, ndim/dtype/mode/filt_len out of range or inconsistent extents)";
    case MIFWT_ERR_UNSUPPORTED: return "valid request this build has no kernel for";
    case MIFWT_ERR_WORKSPACE: return "workspace smaller than mifwt_workspace_bytes()";
    case MIFWT_ERR_LAUNCH: return "HIP kernel launch failed";
    default: return "unknown mifwt error code";
  }
}

static mifwt_level_desc as_zero_mode(const mifwt_level_desc* desc) {
  mifwt_level_desc z = *desc;
  z.mode = MIFWT_MODE_ZERO;
  return z;
}

int mifwt_kernel_id(const mifwt_level_desc* desc, int direction) {
  if (direction == 2) {
    const int rc = validate(desc, 0);
    if (rc != MIFWT_OK) return rc;
    // (boundary extensions: the same launch over the whole signal + the border kernel, mifwt_adjoint_border.hip)
    return desc->mode == MIFWT_MODE_ZERO || adjoint_border_supported(desc) ? pick_kernel(desc, 1) : kGeneric;
  }
  if (direction == 3) {
    const mifwt_level_desc z = as_zero_mode(desc);
    const int rc = validate(&z, 0);
    if (rc != MIFWT_OK) return rc;
    return pick_kernel(&z, 0);
  }
  const int rc = validate(desc, direction);
  if (rc != MIFWT_OK) return rc;
  return pick_kernel(desc, direction);
}

// direction 2 / 3 = adjoint of the analysis / synthesis level described by desc.  A zero-mode analysis adjoint IS a
// synthesis level (reversed dec taps) and every synthesis adjoint IS a zero-mode analysis level (reversed rec
// taps), so they ride on the fast kernels; other boundary modes fold their halo back in the generic adjoint passes.
size_t mifwt_workspace_bytes(const mifwt_level_desc* desc, int direction) {
  if (direction == 2) {
    if (validate(desc, 0) != MIFWT_OK) return 0;
    if (desc->mode == MIFWT_MODE_ZERO || adjoint_border_supported(desc)) return route_ws(desc, 1, pick_kernel(desc, 1));
    return generic_ws(desc, 1);
  }
  if (direction == 3) {
    const mifwt_level_desc z = as_zero_mode(desc);
    if (validate(&z, 0) != MIFWT_OK) return 0;
    return route_ws(&z, 0, pick_kernel(&z, 0));
  }
  if (validate(desc, direction) != MIFWT_OK) return 0;
  return route_ws(desc, direction, pick_kernel(desc, direction));
}

static int run_fwd(const mifwt_level_desc* desc, const void* x, void* approx, void* const* details,
                   const double* dec_lo, const double* dec_hi, void* workspace, size_t workspace_bytes,
                   void* stream) {
  int rc = validate(desc, 0);
  if (rc != MIFWT_OK) return rc;
  if (!x || !approx || !details || !dec_lo || !dec_hi) return MIFWT_ERR_BADARG;
  for (int s = 1; s < (1 << desc->ndim); ++s)
    if (!details[s - 1]) return MIFWT_ERR_BADARG;
  if (desc->batch == 0) return MIFWT_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int kid = pick_kernel(desc, 0);
  const size_t need = route_ws(desc, 0, kid);
  if (need > 0 && (!workspace || workspace_bytes < need)) return MIFWT_ERR_WORKSPACE;
  switch (kid) {
    case kDwt2FwdStream: return dwt2_fwd_stream(desc, x, approx, details, dec_lo, dec_hi, st);
    case kDwt2FwdTile: return dwt2_fwd_tile(desc, x, approx, details, dec_lo, dec_hi, st);
    case kDwt2FwdMfma: return dwt2_fwd_mfma(desc, x, approx, details, dec_lo, dec_hi, st);
    case kDwt2FwdPyr: return dwt2_fwd_fused(desc, x, approx, details, dec_lo, dec_hi, st);  // (one level through the streaming kernel)
    case kDwt3FwdTile: return dwt3_fwd_tile(desc, x, approx, details, dec_lo, dec_hi, st);
    case kDwt3FwdWalk: return dwt3_fwd_walk(desc, x, approx, details, dec_lo, dec_hi, st);
    case kDwt3FwdStream: return plane3_fwd(desc, x, approx, details, dec_lo, dec_hi, workspace, st);
    case kDwt1FwdRow: return rows_fwd(desc, x, approx, details, dec_lo, dec_hi, workspace, st);
    default: break;
  }
  return generic_fwd(desc, x, approx, details, dec_lo, dec_hi, workspace, st);
}

static int run_inv(const mifwt_level_desc* desc, const void* approx, const void* const* details, void* y,
                   const double* rec_lo, const double* rec_hi, void* workspace, size_t workspace_bytes,
                   void* stream) {
  int rc = validate(desc, 1);
  if (rc != MIFWT_OK) return rc;
  if (!y || !approx || !details || !rec_lo || !rec_hi) return MIFWT_ERR_BADARG;
  for (int s = 1; s < (1 << desc->ndim); ++s)
    if (!details[s - 1]) return MIFWT_ERR_BADARG;
  if (desc->batch == 0) return MIFWT_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int kid = pick_kernel(desc, 1);
  const size_t need = route_ws(desc, 1, kid);
  if (need > 0 && (!workspace || workspace_bytes < need)) return MIFWT_ERR_WORKSPACE;
  switch (kid) {
    case kDwt2InvStream: return dwt2_inv_stream(desc, approx, details, y, rec_lo, rec_hi, st);
    case kDwt2InvTile: return dwt2_inv_tile(desc, approx, details, y, rec_lo, rec_hi, st);
    case kDwt2InvMfma: return dwt2_inv_mfma(desc, approx, details, y, rec_lo, rec_hi, st);
    case kDwt2InvPyr: return dwt2_inv_fused(desc, approx, details, y, rec_lo, rec_hi, st);  // (one level through the streaming kernel)
    case kDwt3InvTile: return dwt3_inv_tile(desc, approx, details, y, rec_lo, rec_hi, st);
    case kDwt3InvWalk: return dwt3_inv_walk(desc, approx, details, y, rec_lo, rec_hi, st);
    case kDwt3InvStream: return plane3_inv(desc, approx, details, y, rec_lo, rec_hi, workspace, st);
    case kDwt1InvRow: return rows_inv(desc, approx, details, y, rec_lo, rec_hi, workspace, st);
    default: break;
  }
  return generic_inv(desc, approx, details, y, rec_lo, rec_hi, workspace, st);
}

int mifwt_dwt_fwd(const mifwt_level_desc* desc, const void* x, void* approx, void* const* details,
                  const double* dec_lo, const double* dec_hi, void* workspace, size_t workspace_bytes,
                  void* stream) {
  return run_fwd(desc, x, approx, details, dec_lo, dec_hi, workspace, workspace_bytes, stream);
}

int mifwt_dwt_inv(const mifwt_level_desc* desc, const void* approx, const void* const* details, void* y,
                  const double* rec_lo, const double* rec_hi, void* workspace, size_t workspace_bytes,
                  void* stream) {
  return run_inv(desc, approx, details, y, rec_lo, rec_hi, workspace, workspace_bytes, stream);
}

int mifwt_dwt_fwd_adjoint(const mifwt_level_desc* desc, const void* g_approx, const void* const* g_details, void* g_x,
                          const double* dec_lo, const double* dec_hi, void* workspace, size_t workspace_bytes,
                          void* stream) {
  int rc = validate(desc, 0);
  if (rc != MIFWT_OK) return rc;
  if (!g_x || !g_approx || !g_details || !dec_lo || !dec_hi) return MIFWT_ERR_BADARG;
  const int L = desc->filt_len;
  const bool fold_back = desc->mode != MIFWT_MODE_ZERO && adjoint_border_supported(desc);
  if (desc->mode == MIFWT_MODE_ZERO || fold_back) {
    // u[n] = sum_k a[k] h[2k + 1 - n] is the synthesis formula with g[j] = h[L - 1 - j]; its cropped interior
    // [0, 2M - L + 2 - N%2) is exactly [0, N)
    double lo[MIFWT_MAX_FILT], hi[MIFWT_MAX_FILT];
    for (int j = 0; j < L; ++j) {
      lo[j] = dec_lo[L - 1 - j];
      hi[j] = dec_hi[L - 1 - j];
    }
    if (!fold_back) return run_inv(desc, g_approx, g_details, g_x, lo, hi, workspace, workspace_bytes, stream);
    // a boundary extension: the interior of the adjoint is the zero-mode adjoint (a sample away from the borders has no pad position
    // mapped onto it); the samples near a border are recomputed with the pad positions folded back (mifwt_adjoint_border.hip)
    for (int s = 1; s < (1 << desc->ndim); ++s)
      if (!g_details[s - 1]) return MIFWT_ERR_BADARG;
    const mifwt_level_desc z = as_zero_mode(desc);
    rc = run_inv(&z, g_approx, g_details, g_x, lo, hi, workspace, workspace_bytes, stream);
    if (rc != MIFWT_OK || desc->batch == 0) return rc;
    return adjoint_border(desc, g_approx, g_details, g_x, dec_lo, dec_hi, static_cast<hipStream_t>(stream));
  }
  for (int s = 1; s < (1 << desc->ndim); ++s)
    if (!g_details[s - 1]) return MIFWT_ERR_BADARG;
  if (desc->batch == 0) return MIFWT_OK;
  const size_t need = generic_ws(desc, 1);
  if (need > 0 && (!workspace || workspace_bytes < need)) return MIFWT_ERR_WORKSPACE;
  return generic_inv(desc, g_approx, g_details, g_x, dec_lo, dec_hi, workspace, static_cast<hipStream_t>(stream), true);
}

int mifwt_dwt_inv_adjoint(const mifwt_level_desc* desc, const void* g_y, void* g_approx, void* const* g_details,
                          const double* rec_lo, const double* rec_hi, void* workspace, size_t workspace_bytes,
                          void* stream) {
  int rc = validate(desc, 1);
  if (rc != MIFWT_OK) return rc;
  if (!rec_lo || !rec_hi) return MIFWT_ERR_BADARG;
  // g_a[k] = sum_n g_y[n] g[n + L - 2 - 2k] is the zero-mode analysis formula with h[m] = g[L - 1 - m]; its
  // output extent floor((Nout + L - 1) / 2) is exactly M
  const int L = desc->filt_len;
  double lo[MIFWT_MAX_FILT], hi[MIFWT_MAX_FILT];
  for (int j = 0; j < L; ++j) {
    lo[j] = rec_lo[L - 1 - j];
    hi[j] = rec_hi[L - 1 - j];
  }
  const mifwt_level_desc z = as_zero_mode(desc);
  return run_fwd(&z, g_y, g_approx, g_details, lo, hi, workspace, workspace_bytes, stream);
}

// ---- device-resident taps (include/mifwt.h): the four level operations with the filter read from device memory by the kernels.
// Nothing here touches the taps on the host.  Round 5: the generic axis passes.  Round 6: the fused 2-D kernels that serve a
// learnable-wavelet training step on image-sized planes — LDS tiles (ids 7 / 8), one level through the streaming kernels (ids 16 / 22),
// the border kernels of the analysis adjoint — and the streaming axis passes (ids 3 / 4: 1-D levels and the per-axis maps of the tap
// gradients; ids 5 / 6: the composed 3-D route) take the same device arrays (DevTapArg, mifwt_common.h); everything else stays on the
// generic passes.
namespace {
struct DtapsScope {
  DtapsScope(const double* lo, const double* hi, int rev) {
    mifwt::g_dtaps = {lo, hi, rev};
    mifwt::g_dtaps_taken = 0;
  }
  ~DtapsScope() { mifwt::g_dtaps = {nullptr, nullptr, 0}; }
  // a fused route must have handed the device taps to every kernel it launched: one that ran on the (zero) host taps is a bug, and loud
  static int checked(int rc) { return rc == MIFWT_OK && mifwt::g_dtaps_taken == 0 ? MIFWT_ERR_LAUNCH : rc; }
};
const double kNoHostTaps[MIFWT_MAX_FILT] = {0};
// kernel ids whose kernels read DevTapArg
bool dtaps_fused(int kid) {
  return kid == kDwt2FwdTile || kid == kDwt2InvTile || kid == kDwt2FwdPyr || kid == kDwt2InvPyr || kid == kDwt1FwdRow || kid == kDwt1InvRow;
}
// which kernel serves a device-tap call: that of the host-tap call where it reads device taps (direction 2: also the border kernel),
// else the generic passes
int dtaps_kernel(const mifwt_level_desc* desc, int direction) {
  if (g_options[MIFWT_OPT_FORCE_GENERIC]) return kGeneric;
  int kid = kGeneric;
  if (direction == 0 || direction == 1) {
    kid = pick_kernel(desc, direction);
  } else if (direction == 2) {
    if (desc->mode == MIFWT_MODE_ZERO || adjoint_border_supported(desc)) kid = pick_kernel(desc, 1);
  } else {
    const mifwt_level_desc z = as_zero_mode(desc);
    kid = pick_kernel(&z, 0);
  }
  return dtaps_fused(kid) ? kid : kGeneric;
}
}  // namespace

int mifwt_kernel_id_dtaps(const mifwt_level_desc* desc, int direction) {
  if (!desc || direction < 0 || direction > 3) return MIFWT_ERR_BADARG;
  mifwt_level_desc z = *desc;
  if (direction == 3) z.mode = MIFWT_MODE_ZERO;
  const int rc = validate(&z, direction == 1 ? 1 : 0);
  if (rc != MIFWT_OK) return rc;
  return dtaps_kernel(desc, direction);
}

size_t mifwt_workspace_bytes_dtaps(const mifwt_level_desc* desc, int direction) {
  if (direction == 3) {
    const mifwt_level_desc z = as_zero_mode(desc);
    if (validate(&z, 0) != MIFWT_OK) return 0;
    const int kid = dtaps_kernel(desc, 3);
    return kid == kGeneric ? generic_ws(&z, 0) : route_ws(&z, 0, kid);
  }
  if (validate(desc, direction == 1 ? 1 : 0) != MIFWT_OK) return 0;
  const int kid = dtaps_kernel(desc, direction);
  if (kid != kGeneric) return route_ws(desc, direction == 0 ? 0 : 1, kid);
  return generic_ws(desc, direction == 0 ? 0 : 1);
}

int mifwt_dwt_fwd_dtaps(const mifwt_level_desc* desc, const void* x, void* approx, void* const* details, const double* d_dec_lo,
                        const double* d_dec_hi, void* workspace, size_t workspace_bytes, void* stream) {
  const int rc = validate(desc, 0);
  if (rc != MIFWT_OK) return rc;
  if (!x || !approx || !details || !d_dec_lo || !d_dec_hi) return MIFWT_ERR_BADARG;
  for (int s = 1; s < (1 << desc->ndim); ++s)
    if (!details[s - 1]) return MIFWT_ERR_BADARG;
  if (desc->batch == 0) return MIFWT_OK;
  DtapsScope scope(d_dec_lo, d_dec_hi, 0);
  if (dtaps_kernel(desc, 0) != kGeneric)
    return DtapsScope::checked(run_fwd(desc, x, approx, details, kNoHostTaps, kNoHostTaps, workspace, workspace_bytes, stream));
  const size_t need = generic_ws(desc, 0);
  if (need > 0 && (!workspace || workspace_bytes < need)) return MIFWT_ERR_WORKSPACE;
  return generic_fwd(desc, x, approx, details, kNoHostTaps, kNoHostTaps, workspace, static_cast<hipStream_t>(stream));
}

int mifwt_dwt_inv_dtaps(const mifwt_level_desc* desc, const void* approx, const void* const* details, void* y, const double* d_rec_lo,
                        const double* d_rec_hi, void* workspace, size_t workspace_bytes, void* stream) {
  const int rc = validate(desc, 1);
  if (rc != MIFWT_OK) return rc;
  if (!y || !approx || !details || !d_rec_lo || !d_rec_hi) return MIFWT_ERR_BADARG;
  for (int s = 1; s < (1 << desc->ndim); ++s)
    if (!details[s - 1]) return MIFWT_ERR_BADARG;
  if (desc->batch == 0) return MIFWT_OK;
  DtapsScope scope(d_rec_lo, d_rec_hi, 0);
  if (dtaps_kernel(desc, 1) != kGeneric)
    return DtapsScope::checked(run_inv(desc, approx, details, y, kNoHostTaps, kNoHostTaps, workspace, workspace_bytes, stream));
  const size_t need = generic_ws(desc, 1);
  if (need > 0 && (!workspace || workspace_bytes < need)) return MIFWT_ERR_WORKSPACE;
  return generic_inv(desc, approx, details, y, kNoHostTaps, kNoHostTaps, workspace, static_cast<hipStream_t>(stream));
}

int mifwt_dwt_fwd_adjoint_dtaps(const mifwt_level_desc* desc, const void* g_approx, const void* const* g_details, void* g_x,
                                const double* d_dec_lo, const double* d_dec_hi, void* workspace, size_t workspace_bytes, void* stream) {
  int rc = validate(desc, 0);
  if (rc != MIFWT_OK) return rc;
  if (!g_x || !g_approx || !g_details || !d_dec_lo || !d_dec_hi) return MIFWT_ERR_BADARG;
  for (int s = 1; s < (1 << desc->ndim); ++s)
    if (!g_details[s - 1]) return MIFWT_ERR_BADARG;
  if (desc->batch == 0) return MIFWT_OK;
  if (dtaps_kernel(desc, 2) != kGeneric) {
    // (mifwt_dwt_fwd_adjoint: the synthesis kernel with the dec taps REVERSED over the whole signal, then — a boundary extension — the
    // border kernel with the dec taps as they are)
    const mifwt_level_desc z = as_zero_mode(desc);
    {
      DtapsScope scope(d_dec_lo, d_dec_hi, 1);
      rc = DtapsScope::checked(run_inv(&z, g_approx, g_details, g_x, kNoHostTaps, kNoHostTaps, workspace, workspace_bytes, stream));
    }
    if (rc != MIFWT_OK || desc->mode == MIFWT_MODE_ZERO) return rc;
    DtapsScope scope(d_dec_lo, d_dec_hi, 0);
    return DtapsScope::checked(adjoint_border(desc, g_approx, g_details, g_x, kNoHostTaps, kNoHostTaps, static_cast<hipStream_t>(stream)));
  }
  const size_t need = generic_ws(desc, 1);
  if (need > 0 && (!workspace || workspace_bytes < need)) return MIFWT_ERR_WORKSPACE;
  DtapsScope scope(d_dec_lo, d_dec_hi, 0);
  return generic_inv(desc, g_approx, g_details, g_x, kNoHostTaps, kNoHostTaps, workspace, static_cast<hipStream_t>(stream), true);
}

int mifwt_dwt_inv_adjoint_dtaps(const mifwt_level_desc* desc, const void* g_y, void* g_approx, void* const* g_details,
                                const double* d_rec_lo, const double* d_rec_hi, void* workspace, size_t workspace_bytes, void* stream) {
  int rc = validate(desc, 1);
  if (rc != MIFWT_OK) return rc;
  if (!g_y || !g_approx || !g_details || !d_rec_lo || !d_rec_hi) return MIFWT_ERR_BADARG;
  // (a zero-mode analysis level with the rec taps reversed: mifwt_dwt_inv_adjoint)
  const mifwt_level_desc z = as_zero_mode(desc);
  rc = validate(&z, 0);
  if (rc != MIFWT_OK) return rc;
  for (int s = 1; s < (1 << desc->ndim); ++s)
    if (!g_details[s - 1]) return MIFWT_ERR_BADARG;
  if (desc->batch == 0) return MIFWT_OK;
  DtapsScope scope(d_rec_lo, d_rec_hi, 1);
  if (dtaps_kernel(desc, 3) != kGeneric)
    return DtapsScope::checked(run_fwd(&z, g_y, g_approx, g_details, kNoHostTaps, kNoHostTaps, workspace, workspace_bytes, stream));
  const size_t need = generic_ws(&z, 0);
  if (need > 0 && (!workspace || workspace_bytes < need)) return MIFWT_ERR_WORKSPACE;
  return generic_fwd(&z, g_y, g_approx, g_details, kNoHostTaps, kNoHostTaps, workspace, static_cast<hipStream_t>(stream));
}

// One 1-D analysis level along the MIDDLE axis of [batch, n, inner] arrays (inner contiguous) — the streaming outer-axis kernel
// (mifwt_axis_stream.h) on its own: what the tap gradients of a 2-D level along its column axis need of the input (the level along
// the rows only), in the natural layout.  d_dec_lo / d_dec_hi: device taps (non-null: the host taps are ignored).
int mifwt_dwt1_fwd_outer(int dtype, int64_t batch, int64_t n, int64_t inner, const void* x, int64_t x_batch_stride, int64_t x_axis_stride, void* lo_out,
                         void* hi_out, int64_t out_batch_stride, int64_t out_axis_stride, int mode, int filt_len, const double* dec_lo,
                         const double* dec_hi, const double* d_dec_lo, const double* d_dec_hi, void* stream) {
  if (!x || !lo_out || !hi_out || batch < 0 || n < 1 || inner < 1) return MIFWT_ERR_BADARG;
  if (!((dec_lo && dec_hi) || (d_dec_lo && d_dec_hi))) return MIFWT_ERR_BADARG;
  if (mode < MIFWT_MODE_ZERO || mode > MIFWT_MODE_SYMMETRIC) return MIFWT_ERR_BADARG;
  if ((dtype != MIFWT_F32 && dtype != MIFWT_F64) || !stream_filter_supported(filt_len)) return MIFWT_ERR_UNSUPPORTED;
  if (batch == 0) return MIFWT_OK;
  StreamJob jb;
  memset(&jb, 0, sizeof(jb));
  jb.in0 = x;
  jb.out0 = lo_out;
  jb.out1 = hi_out;
  jb.in0_s[0] = x_batch_stride, jb.in0_s[1] = x_axis_stride;
  jb.out0_s[0] = jb.out1_s[0] = out_batch_stride, jb.out0_s[1] = jb.out1_s[1] = out_axis_stride;
  StreamCall c;
  memset(&c, 0, sizeof(c));
  c.filt_len = filt_len;
  c.mode = mode;
  c.jobs = &jb;
  c.njobs = 1;
  c.batch = batch;
  c.inner = inner;
  c.n_in = n;
  c.n_out = (n + filt_len - 1) / 2;
  c.lo = d_dec_lo ? kNoHostTaps : dec_lo;
  c.hi = d_dec_lo ? kNoHostTaps : dec_hi;
  c.stream = static_cast<hipStream_t>(stream);
  if (d_dec_lo) {
    DtapsScope scope(d_dec_lo, d_dec_hi, 0);
    return DtapsScope::checked(stream_call(dtype, kOuterFwd, c));
  }
  return stream_call(dtype, kOuterFwd, c);
}

// ... and its inverse: one 1-D synthesis level along the middle axis, (lo, hi) [batch, m, inner] -> y [batch, n_out, inner] (n_out = the
// cropped extent, 2 m - L + 2 or one less): the bands of a 2-D level synthesised along the rows axis only, for the tap gradients
// along the columns.
int mifwt_dwt1_inv_outer(int dtype, int64_t batch, int64_t m, int64_t n_out, int64_t inner, const void* lo_in, int64_t lo_batch_stride,
                         int64_t lo_axis_stride, const void* hi_in, int64_t hi_batch_stride, int64_t hi_axis_stride, void* y, int64_t y_batch_stride,
                         int64_t y_axis_stride, int filt_len, const double* rec_lo, const double* rec_hi, const double* d_rec_lo,
                         const double* d_rec_hi, void* stream) {
  if (!lo_in || !hi_in || !y || batch < 0 || m < 1 || inner < 1 || n_out < 1) return MIFWT_ERR_BADARG;
  if (!((rec_lo && rec_hi) || (d_rec_lo && d_rec_hi))) return MIFWT_ERR_BADARG;
  if (n_out > 2 * m - filt_len + 2 || n_out < 2 * m - filt_len + 1) return MIFWT_ERR_BADARG;
  if ((dtype != MIFWT_F32 && dtype != MIFWT_F64) || !stream_filter_supported(filt_len)) return MIFWT_ERR_UNSUPPORTED;
  if (batch == 0) return MIFWT_OK;
  StreamJob jb;
  memset(&jb, 0, sizeof(jb));
  jb.in0 = lo_in;
  jb.in1 = hi_in;
  jb.out0 = y;
  jb.in0_s[0] = lo_batch_stride, jb.in0_s[1] = lo_axis_stride;
  jb.in1_s[0] = hi_batch_stride, jb.in1_s[1] = hi_axis_stride;
  jb.out0_s[0] = y_batch_stride, jb.out0_s[1] = y_axis_stride;
  StreamCall c;
  memset(&c, 0, sizeof(c));
  c.filt_len = filt_len;
  c.jobs = &jb;
  c.njobs = 1;
  c.batch = batch;
  c.inner = inner;
  c.n_in = m;
  c.n_out = n_out;
  c.lo = d_rec_lo ? kNoHostTaps : rec_lo;
  c.hi = d_rec_lo ? kNoHostTaps : rec_hi;
  c.stream = static_cast<hipStream_t>(stream);
  if (d_rec_lo) {
    DtapsScope scope(d_rec_lo, d_rec_hi, 0);
    return DtapsScope::checked(stream_call(dtype, kOuterInv, c));
  }
  return stream_call(dtype, kOuterInv, c);
}

// Two consecutive 2-D analysis levels in one launch (mifwt_dwt2_fwd_pair.hip); d2 describes the second level, whose
// input is the (never materialised) approximation of d1.
int mifwt_dwt2_fwd_pair_supported(const mifwt_level_desc* d1, const mifwt_level_desc* d2) {
  if (!d1 || !d2 || validate(d1, 0) != MIFWT_OK || validate(d2, 0) != MIFWT_OK) return 0;
  return dwt2_fwd_pair_supported(d1, d2) ? 1 : 0;
}

int mifwt_dwt2_fwd_pair(const mifwt_level_desc* d1, const mifwt_level_desc* d2, const void* x, void* const* details1,
                        void* approx2, void* const* details2, const double* dec_lo, const double* dec_hi, void* stream) {
  if (!d1 || !d2) return MIFWT_ERR_BADARG;
  int rc = validate(d1, 0);
  if (rc == MIFWT_OK) rc = validate(d2, 0);
  if (rc != MIFWT_OK) return rc;
  if (!x || !details1 || !approx2 || !details2 || !dec_lo || !dec_hi) return MIFWT_ERR_BADARG;
  for (int s = 0; s < 3; ++s)
    if (!details1[s] || !details2[s]) return MIFWT_ERR_BADARG;
  if (!dwt2_fwd_pair_supported(d1, d2)) return MIFWT_ERR_UNSUPPORTED;
  if (d1->batch == 0) return MIFWT_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  // rolling strips carry no row halo but more bookkeeping per row: measured ahead of the tile version only for 8 taps
  // (64 x 1024^2, 3 levels, same runs: db4 131 vs 139 us; db3 124-139 vs 119-127 us, db2 121 vs 113 us, haar 110 vs
  // 101 us), so auto mode picks by filter length
  const int pm = g_options[MIFWT_OPT_PAIR_MODE];
  if (pm != 1 && (pm == 3 || d1->filt_len >= 8) && dwt2_fwd_roll_supported(d1, d2))
    return dwt2_fwd_roll(d1, d2, x, details1, approx2, details2, dec_lo, dec_hi, st);
  return dwt2_fwd_pair(d1, d2, x, details1, approx2, details2, dec_lo, dec_hi, st);
}
// Up to three consecutive 2-D analysis levels in one launch (mifwt_dwt2_fwd_pyr.hip).
// which kernel serves a multi-level 2-D analysis call: 0 none, 1 the streaming three-level kernel (mifwt_dwt2_fwd_pyr.hip), 2 the
// whole-pyramid kernel for small planes (mifwt_dwt2_fwd_small.hip).  Auto mode prefers the small-plane kernel; with
// MIFWT_OPT_PYRAMID_MODE 1 ("the streaming kernel wherever it can run") the streaming kernel goes first.
static int pyramid_route(int nlevels, const mifwt_level_desc* const* descs) {
  const bool pyr = nlevels <= 3 && dwt2_fwd_pyr_supported(nlevels, descs);
  if (pyr && g_options[MIFWT_OPT_PYRAMID_MODE] == 1) return 1;
  // (a SINGLE level of a small plane is better off in the per-level tile kernel unless the batch is big: one workgroup per image at a
  // time — 256 x 67^2: 60.7 us for the call against 54.9, 32 x 70^2: 10.6 against ~6 us for the level; MIFWT_OPT_PYRAMID_MODE 3 lifts it)
  const bool lone = nlevels == 1 && descs[0]->batch < 1024 && g_options[MIFWT_OPT_PYRAMID_MODE] != 3;
  if (!lone && dwt2_fwd_small_supported(nlevels, descs)) return 2;
  return pyr ? 1 : 0;
}

int mifwt_dwt2_fwd_pyramid_supported(int nlevels, const mifwt_level_desc* const* descs) {
  if (!descs || nlevels < 1 || nlevels > 8) return 0;
  for (int l = 0; l < nlevels; ++l)
    if (!descs[l] || validate(descs[l], 0) != MIFWT_OK) return 0;
  return pyramid_route(nlevels, descs);
}

int mifwt_dwt2_fwd_pyramid(int nlevels, const mifwt_level_desc* const* descs, const void* x, void* const* const* details, void* approx,
                           const double* dec_lo, const double* dec_hi, void* stream) {
  if (!descs || nlevels < 1 || nlevels > 8) return MIFWT_ERR_BADARG;
  for (int l = 0; l < nlevels; ++l) {
    if (!descs[l]) return MIFWT_ERR_BADARG;
    const int rc = validate(descs[l], 0);
    if (rc != MIFWT_OK) return rc;
  }
  if (!x || !details || !approx || !dec_lo || !dec_hi) return MIFWT_ERR_BADARG;
  for (int l = 0; l < nlevels; ++l) {
    if (!details[l]) return MIFWT_ERR_BADARG;
    for (int s = 0; s < 3; ++s)
      if (!details[l][s]) return MIFWT_ERR_BADARG;
  }
  const int route = pyramid_route(nlevels, descs);
  if (route == 0) return MIFWT_ERR_UNSUPPORTED;
  if (descs[0]->batch == 0) return MIFWT_OK;
  if (route == 2) return dwt2_fwd_small(nlevels, descs, x, details, approx, dec_lo, dec_hi, static_cast<hipStream_t>(stream));
  return dwt2_fwd_pyr(nlevels, descs, x, details, approx, dec_lo, dec_hi, static_cast<hipStream_t>(stream));
}
int mifwt_dwt2_fwd_pyramid_schedule(int nlevels, const mifwt_level_desc* const* descs, unsigned int* wg_start, int capacity) {
  if (!descs || !wg_start || nlevels < 1 || nlevels > 3) return MIFWT_ERR_BADARG;
  for (int l = 0; l < nlevels; ++l) {
    if (!descs[l]) return MIFWT_ERR_BADARG;
    const int rc = validate(descs[l], 0);
    if (rc != MIFWT_OK) return rc;
  }
  if (pyramid_route(nlevels, descs) != 1) return MIFWT_ERR_UNSUPPORTED;
  return dwt2_fwd_pyr_schedule(nlevels, descs, wg_start, capacity);
}
int mifwt_dwt3_fwd_slab_plan(const mifwt_level_desc* desc, int* out, int capacity) {
  if (!desc || !out) return MIFWT_ERR_BADARG;
  const int rc = validate(desc, 0);
  if (rc != MIFWT_OK) return rc;
  return dwt3_fwd_slab_plan_query(desc, out, capacity);
}
// Every level of a 2-D reconstruction of a small plane in one launch (mifwt_dwt2_inv_small.hip); descs[0] = the coarsest level.
int mifwt_dwt2_inv_pyramid_supported(int nlevels, const mifwt_level_desc* const* descs) {
  if (!descs || nlevels < 1 || nlevels > 8) return 0;
  for (int l = 0; l < nlevels; ++l)
    if (!descs[l] || validate(descs[l], 1) != MIFWT_OK) return 0;
  if (dwt2_inv_small_supported(nlevels, descs)) return 1;
  return nlevels <= 3 && dwt2_inv_pyr_supported(nlevels, descs) ? 2 : 0;
}

int mifwt_dwt2_inv_pyramid(int nlevels, const mifwt_level_desc* const* descs, const void* approx, const void* const* const* details, void* y,
                           const double* rec_lo, const double* rec_hi, void* stream) {
  if (!descs || nlevels < 1 || nlevels > 8) return MIFWT_ERR_BADARG;
  for (int l = 0; l < nlevels; ++l) {
    if (!descs[l]) return MIFWT_ERR_BADARG;
    const int rc = validate(descs[l], 1);
    if (rc != MIFWT_OK) return rc;
  }
  if (!approx || !details || !y || !rec_lo || !rec_hi) return MIFWT_ERR_BADARG;
  for (int l = 0; l < nlevels; ++l) {
    if (!details[l]) return MIFWT_ERR_BADARG;
    for (int s = 0; s < 3; ++s)
      if (!details[l][s]) return MIFWT_ERR_BADARG;
  }
  if (dwt2_inv_small_supported(nlevels, descs)) {
    if (descs[0]->batch == 0) return MIFWT_OK;
    return dwt2_inv_small(nlevels, descs, approx, details, y, rec_lo, rec_hi, static_cast<hipStream_t>(stream));
  }
  if (nlevels > 3 || !dwt2_inv_pyr_supported(nlevels, descs)) return MIFWT_ERR_UNSUPPORTED;
  return dwt2_inv_pyr(nlevels, descs, approx, details, y, rec_lo, rec_hi, static_cast<hipStream_t>(stream));
}
// Two consecutive 2-D synthesis levels in one launch (mifwt_idwt2_pair.hip); d2 describes the coarser level, whose
// (cropped) output is the approximation of d1 and is never materialised.
int mifwt_dwt2_inv_pair_supported(const mifwt_level_desc* d2, const mifwt_level_desc* d1) {
  if (!d1 || !d2 || validate(d1, 1) != MIFWT_OK || validate(d2, 1) != MIFWT_OK) return 0;
  return dwt2_inv_pair_supported(d2, d1) ? 1 : 0;
}

int mifwt_dwt2_inv_pair(const mifwt_level_desc* d2, const mifwt_level_desc* d1, const void* approx2, const void* const* details2,
                        const void* const* details1, void* y, const double* rec_lo, const double* rec_hi, void* stream) {
  if (!d1 || !d2) return MIFWT_ERR_BADARG;
  int rc = validate(d2, 1);
  if (rc == MIFWT_OK) rc = validate(d1, 1);
  if (rc != MIFWT_OK) return rc;
  if (!approx2 || !details2 || !details1 || !y || !rec_lo || !rec_hi) return MIFWT_ERR_BADARG;
  for (int s = 0; s < 3; ++s)
    if (!details1[s] || !details2[s]) return MIFWT_ERR_BADARG;
  if (!dwt2_inv_pair_supported(d2, d1)) return MIFWT_ERR_UNSUPPORTED;
  if (d1->batch == 0) return MIFWT_OK;
  return dwt2_inv_pair(d2, d1, approx2, details2, details1, y, rec_lo, rec_hi, static_cast<hipStream_t>(stream));
}
// The deep levels of a 1-D decomposition in one launch (mifwt_dwt1_tail.hip).
int mifwt_dwt1_fwd_tail_max_n(int dtype) { return dwt1_tail_max_n(dtype); }

int mifwt_dwt1_fwd_tail(int dtype, int filt_len, int mode, int64_t rows, int64_t n, int nlevels, const void* x, int64_t x_row_stride,
                        void* approx, int64_t approx_row_stride, void* const* details, const int64_t* detail_row_strides,
                        const double* dec_lo, const double* dec_hi, void* stream) {
  if (!x || !approx || !details || !detail_row_strides || !dec_lo || !dec_hi) return MIFWT_ERR_BADARG;
  if (!dwt1_tail_supported(dtype, filt_len, mode, rows, n, nlevels)) return MIFWT_ERR_UNSUPPORTED;
  for (int l = 0; l < nlevels; ++l)
    if (!details[l]) return MIFWT_ERR_BADARG;
  return dwt1_tail(dtype, filt_len, mode, rows, n, nlevels, x, x_row_stride, approx, approx_row_stride, details, detail_row_strides,
                   dec_lo, dec_hi, static_cast<hipStream_t>(stream));
}
// Several levels of a 1-D decomposition of long rows in one launch (mifwt_dwt1_long.hip).
int mifwt_dwt1_fwd_long_levels(int dtype, int filt_len, int mode, int64_t rows, int64_t n, int want) {
  return dwt1_long_levels(dtype, filt_len, mode, rows, n, want);
}

int mifwt_dwt1_fwd_long(int dtype, int filt_len, int mode, int64_t rows, int64_t n, int nlevels, const void* x, int64_t x_row_stride,
                        void* approx, int64_t approx_row_stride, void* const* details, const int64_t* detail_row_strides,
                        const double* dec_lo, const double* dec_hi, void* stream) {
  if (!x || !approx || !details || !detail_row_strides || !dec_lo || !dec_hi) return MIFWT_ERR_BADARG;
  if (nlevels < 1 || dwt1_long_levels(dtype, filt_len, mode, rows, n, nlevels) != nlevels) return MIFWT_ERR_UNSUPPORTED;
  for (int l = 0; l < nlevels; ++l)
    if (!details[l]) return MIFWT_ERR_BADARG;
  return dwt1_long(dtype, filt_len, mode, rows, n, nlevels, x, x_row_stride, approx, approx_row_stride, details, detail_row_strides,
                   dec_lo, dec_hi, static_cast<hipStream_t>(stream));
}
// The finest levels of a 1-D reconstruction in one launch, a chunk of the output per workgroup (mifwt_dwt1_long.hip).
int mifwt_dwt1_inv_long_supported(int dtype, int filt_len, int64_t rows, int nlevels, const int32_t* m) {
  return idwt1_long_supported(dtype, filt_len, rows, nlevels, m);
}

int mifwt_dwt1_inv_long(int dtype, int filt_len, int64_t rows, int nlevels, const int32_t* m, const void* approx, int64_t approx_row_stride,
                        const void* const* details, const int64_t* detail_row_strides, void* y, int64_t y_row_stride,
                        const double* rec_lo, const double* rec_hi, void* stream) {
  if (!m || !approx || !details || !detail_row_strides || !y || !rec_lo || !rec_hi) return MIFWT_ERR_BADARG;
  if (!idwt1_long_supported(dtype, filt_len, rows, nlevels, m)) return MIFWT_ERR_UNSUPPORTED;
  for (int l = 0; l < nlevels; ++l)
    if (!details[l]) return MIFWT_ERR_BADARG;
  return idwt1_long(dtype, filt_len, rows, nlevels, m, approx, approx_row_stride, details, detail_row_strides, y, y_row_stride, rec_lo,
                    rec_hi, static_cast<hipStream_t>(stream));
}
// Launch geometry of the chunked 1-D kernels (diagnostic: the host-side model tests check coverage and LDS capacities with it).
int mifwt_dwt1_long_plan(int inverse, int dtype, int filt_len, int mode, int64_t rows, int64_t n, int nlevels, const int32_t* m, int32_t* out6) {
  if (!out6) return 0;
  return inverse ? idwt1_long_plan_query(dtype, filt_len, rows, nlevels, m, out6) : dwt1_long_plan_query(dtype, filt_len, mode, rows, n, nlevels, out6);
}
// The coarse levels of a 1-D reconstruction in one launch (mifwt_dwt1_tail.hip).
int mifwt_dwt1_inv_tail(int dtype, int filt_len, int64_t rows, int64_t m, int nlevels, const void* approx, int64_t approx_row_stride,
                        const void* const* details, const int64_t* detail_row_strides, const int32_t* out_len, void* y,
                        int64_t y_row_stride, const double* rec_lo, const double* rec_hi, void* stream) {
  if (!approx || !details || !detail_row_strides || !out_len || !y || !rec_lo || !rec_hi) return MIFWT_ERR_BADARG;
  if (!idwt1_tail_supported(dtype, filt_len, rows, m, nlevels, out_len)) return MIFWT_ERR_UNSUPPORTED;
  for (int l = 0; l < nlevels; ++l)
    if (!details[l]) return MIFWT_ERR_BADARG;
  return idwt1_tail(dtype, filt_len, rows, m, nlevels, approx, approx_row_stride, details, detail_row_strides, out_len, y, y_row_stride,
                    rec_lo, rec_hi, static_cast<hipStream_t>(stream));
}
}  // extern "C"
