// mifwt_dwt3_inv_walk.hip — fully fused 3-D synthesis level, workgroups WALK along the depth axis (gfx950), kernel id 25.
//
// Replaces, for one level of waverec3 / fswaverec3: torch.stack + F.conv_transpose3d([8,1,L,L,L], stride 2) + the crops (reference
// src/ptwt/conv_transform_3.py:205-249) — separably, in polyphase (gather) form, reading the eight sub-band volumes once and writing
// the reconstruction once.  Mirror of the depth-walking analysis kernel (mifwt_dwt3_fwd_walk.hip, id 24):
//   per axis, output index n = 2p + r:   y[2p + r] = sum_{i < L/2} g_lo[L-2-2i+r] a[p+i] + g_hi[L-2-2i+r] d[p+i]
//   * a workgroup owns 2 CY output rows x EVERY output column of one depth segment of one volume and walks down the segment's
//     coefficient slices: slice z of the eight bands contributes to the output slice pairs z - L/2 + 1 .. z, which live in rolling
//     register accumulators — no depth halo (the brick kernel, id 10, requests CZ + L/2 - 1 slices per CZ it completes and needs
//     a third column tile for the 128th output column pair of a 256-column row: NQ = 64 - (L/2 - 1) pairs per tile);
//   * two LOADER waves (four bands each) request the CY + L/2 - 1 rows of a band's slice as ONE contiguous piece (dense coefficient
//     rows) by LDS-DMA, a ring of slices ahead; their vmcnt queues hold loads only, the compute waves' queues stores only; one
//     s_barrier per coefficient slice;
//   * a COMPUTE wave owns 64 output column PAIRS (lane = pair p): W pass first, straight from the staged rows (the columns p + i of
//     any lane are one LDS read away), then the H and D passes never leave the lane; (column 2p, 2p + 1) leave as 8-byte stores,
//     512 contiguous bytes per wave, row and slice.
// Valid outputs never touch a coefficient row / column / slice beyond the bands' extents (n < 2 M - L + 2), so ragged last groups
// need no zero fill: what lands behind a piece is only read for outputs that are not stored.
// f32 and f64 (template parameter T; f64: two row pairs per workgroup, 8-byte-aligned 16-byte LDS reads, two v_fma_f64 where f32 has
// one packed FMA), even L <= 8, dense coefficient rows (row stride = row length), unit innermost strides, rows of at most 512 outputs.
// Algorithmic traffic: sizeof(T) (8 B Md Mh Mw read + B D H W written).
#include "mifwt_pyr.h"

namespace mifwt {

namespace {

constexpr int kIW3MaxStrips = 4;

typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
template <typename T> struct IW3Pair { typedef f2u type; };
template <> struct IW3Pair<double> { typedef d2u type; };

template <typename T, int L>
struct IWalk3Args {
  const T* in[8];  // band s: bit 2 = depth high, bit 1 = row high, bit 0 = column high
  T* y;
  int64_t is_b[2], ys_b;  // batch strides: [0] approximation, [1] details; output
  uint32_t is_d[2];       // slice strides of the bands (rows are dense)
  uint32_t ys_d, ys_h;
  int Md, Mh, Mw, D, H, W;
  int nstrips;        // compute waves: 64 output column pairs each
  int ngroups;        // row groups of CY output row pairs
  int nseg, seg_out;  // depth segments, output slice PAIRS per segment
  int nslots;         // staged coefficient slices
  int yvec, nt, dbg;
  int st16;  // 16-byte output stores: lanes l and l + 32 own neighbouring column pairs and exchange rows (W, strides, base: multiples of 4)
  FastDiv div_g, div_s;
  typename TileArith<T>::vec2 tlo[L / 2], thi[L / 2];  // (rec_lo[2j], rec_lo[2j+1]), (rec_hi[2j], rec_hi[2j+1])
};

template <int N>
__device__ __forceinline__ void iwalk3_wait(int later) {  // at most `later` slices of N requests each may still be in flight
  if (later >= 5) pyr_wait_vm<(5 * N > 63 ? 63 : 5 * N)>();
  else if (later == 4) pyr_wait_vm<(4 * N > 63 ? 63 : 4 * N)>();
  else if (later == 3) pyr_wait_vm<(3 * N > 63 ? 63 : 3 * N)>();
  else if (later == 2) pyr_wait_vm<(2 * N > 63 ? 63 : 2 * N)>();
  else if (later == 1) pyr_wait_vm<(N > 63 ? 63 : N)>();
  else pyr_wait_vm<0>();
}

// 64 lanes x 16 B -> LDS [lds + 16 lane); default cache policy: the row group next door asks for L/2 - 1 of these rows too
__device__ __forceinline__ void iwalk3_dma(uint32_t voff, rsrc_t rsrc, uint32_t soff, uint32_t lds) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds) : "memory");
}

template <typename T, int L, int CY, int NKB>
__global__ void __launch_bounds__(64 * (kIW3MaxStrips + 2)) idwt3_walk_kernel(const IWalk3Args<T, L> a) {
  using V2 = typename TileArith<T>::vec2;
  using V2U = typename IW3Pair<T>::type;
  constexpr int ES = (int)sizeof(T);
  constexpr int HL = L / 2, IY = CY + HL - 1;
  constexpr int BANDB = NKB * 1024, SLOTB = 8 * BANDB;  // bytes of a band's piece / of a staged slice
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint32_t ug, us;
  const int img = __builtin_amdgcn_readfirstlane((int)a.div_s.divmod(a.div_g.divmod((uint32_t)xcd_remap(blockIdx.x, gridDim.x), ug), us));
  const int py0 = __builtin_amdgcn_readfirstlane((int)ug * CY);  // first output row pair = first coefficient row of the group
  const int NP = (a.D + 1) >> 1;                                  // output slice pairs
  const int PA = __builtin_amdgcn_readfirstlane((int)us * a.seg_out), PB = min(NP, PA + a.seg_out);
  const int nsl = PB - PA + HL - 1;  // coefficient slices PA .. PB + HL - 2 = steps

  // =====================================================================================================================
  // loader waves: bands 4 l .. 4 l + 3
  if (wave >= a.nstrips) {
    const int l = wave - a.nstrips;
    const int nrows = min(IY, a.Mh - py0);
    const uint32_t piece = (uint32_t)(nrows * a.Mw) * (uint32_t)ES;  // bytes of a band's rows py0 .. py0 + nrows - 1 of one slice: contiguous
    uint32_t voff[NKB];
#pragma unroll
    for (int j = 0; j < NKB; ++j) {
      const uint32_t o = 1024u * (uint32_t)j + 16u * (uint32_t)lane;
      voff[j] = o < piece ? o : kPyrOob;
    }
    rsrc_t rs[4];
    uint32_t sd[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int s = 4 * l + b, k = s == 0 ? 0 : 1;
      const uint32_t bytes = (uint32_t)(((int64_t)(a.Md - 1) * a.is_d[k] + (int64_t)a.Mh * a.Mw) * ES);
      rs[b] = pyr_rsrc(a.in[s] + (int64_t)img * a.is_b[k] + (int64_t)py0 * a.Mw, bytes - (uint32_t)(py0 * a.Mw) * (uint32_t)ES);
      sd[b] = a.is_d[k] * (uint32_t)ES;
    }
    __builtin_amdgcn_s_setprio(3);
    constexpr int PER = 4 * NKB;
    int ib = 0;
    auto issue = [&](int t) {
      const uint32_t buf = (uint32_t)ib * (uint32_t)SLOTB + (uint32_t)(4 * l) * (uint32_t)BANDB;
      ib = ib + 1 == a.nslots ? 0 : ib + 1;
      if (MIFWT_DBG(a) & 2) return;
      const uint32_t zc = (uint32_t)(PA + t);
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const uint32_t so = __builtin_amdgcn_readfirstlane(zc * sd[b]);
#pragma unroll
        for (int j = 0; j < NKB; ++j) iwalk3_dma(voff[j], rs[b], so, __builtin_amdgcn_readfirstlane(buf + (uint32_t)(b * BANDB + j * 1024)));
      }
    };
    const int ahead = a.nslots - 1;
    for (int t = 0; t < ahead; ++t)
      if (t < nsl) issue(t);
#pragma unroll 1
    for (int t = 0; t < nsl; ++t) {
      iwalk3_wait<PER>(min(ahead - 1, nsl - 1 - t));
      __syncthreads();
      if (t + ahead < nsl) issue(t + ahead);
    }
    return;
  }

  // =====================================================================================================================
  // compute wave: output column pairs 64 wave .. 64 wave + 63
  // (16-byte stores: lane l < 32 owns pair 2 l of the wave's 64, lane l + 32 pair 2 l + 1 — after the exchange each lane holds four
  // columns of one row)
  const int p = 64 * wave + ((ES == 4 && a.st16) ? 2 * (lane & 31) + (lane >> 5) : lane);
  const int xo = 2 * p;
  const bool active = xo < a.W, both = xo + 1 < a.W;
  const int pc = min(p, a.Mw - HL);  // idle lanes read the row's last window
  uint32_t rowaddr[IY];
#pragma unroll
  for (int yy = 0; yy < IY; ++yy) rowaddr[yy] = (uint32_t)(yy * a.Mw + pc) * (uint32_t)ES;
  T* const yb = a.y + (int64_t)img * a.ys_b;
  const rsrc_t yr = pyr_rsrc(yb, a.st16 ? (uint32_t)(((int64_t)(a.D - 1) * a.ys_d + (int64_t)(a.H - 1) * a.ys_h + a.W) * 4) : 0u);
  const uint32_t c16 = 4u * (uint32_t)(128 * wave + 4 * (lane & 31));  // byte offset of the lane's four columns (16-byte path)

  V2 acc[HL][CY][2][2];  // [slot of the output slice pair][row pair q][column c][row r of the pair]: (slice 2P, slice 2P + 1)
#pragma unroll
  for (int s = 0; s < HL; ++s)
#pragma unroll
    for (int q = 0; q < CY; ++q)
#pragma unroll
      for (int c = 0; c < 2; ++c) acc[s][q][c][0] = acc[s][q][c][1] = V2{};
  int slot = 0;

  // coefficient slice -> himg[d][q][c] = (row 2q, row 2q + 1) of column c of the depth-low / depth-high image
  auto filter_slice = [&](V2 (&himg)[2][CY][2]) {
    const unsigned char* const sl = smem + slot * SLOTB;
    slot = slot + 1 == a.nslots ? 0 : slot + 1;
    if (MIFWT_DBG(a) & 4) {
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int q = 0; q < CY; ++q) himg[d][q][0] = himg[d][q][1] = (V2){T(1), T(2)};
      return;
    }
    V2 wimg[2][2][IY];  // [d][h][row]: (column 2p, 2p + 1)
#pragma unroll
    for (int dh = 0; dh < 4; ++dh) {
      const unsigned char* const lo_b = sl + (2 * dh) * BANDB;      // band (d, h, W low)
      const unsigned char* const hi_b = sl + (2 * dh + 1) * BANDB;  // band (d, h, W high)
#pragma unroll
      for (int yy = 0; yy < IY; ++yy) {
        V2 w;
#pragma unroll
        for (int i2 = 0; i2 < (HL + 1) / 2; ++i2) {
          const V2 pa = *reinterpret_cast<const V2U*>(lo_b + rowaddr[yy] + 2 * ES * i2);  // coefficients p + 2 i2, p + 2 i2 + 1
          const V2 pd = *reinterpret_cast<const V2U*>(hi_b + rowaddr[yy] + 2 * ES * i2);
          if (i2 == 0) w = amul_lo(a.tlo[HL - 1], pa);
          else afma_lo(w, a.tlo[HL - 1 - 2 * i2], pa);
          afma_lo(w, a.thi[HL - 1 - 2 * i2], pd);
          if (2 * i2 + 1 < HL) {
            afma_hi(w, a.tlo[HL - 2 - 2 * i2], pa);
            afma_hi(w, a.thi[HL - 2 - 2 * i2], pd);
          }
        }
        wimg[dh >> 1][dh & 1][yy] = w;
      }
    }
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int q = 0; q < CY; ++q) {
        V2 h0, h1;
#pragma unroll
        for (int i = 0; i < HL; ++i) {
          if (i == 0) {
            h0 = amul_lo(a.tlo[HL - 1], wimg[d][0][q]);
            h1 = amul_hi(a.tlo[HL - 1], wimg[d][0][q]);
          } else {
            afma_lo(h0, a.tlo[HL - 1 - i], wimg[d][0][q + i]);
            afma_hi(h1, a.tlo[HL - 1 - i], wimg[d][0][q + i]);
          }
          afma_lo(h0, a.thi[HL - 1 - i], wimg[d][1][q + i]);
          afma_hi(h1, a.thi[HL - 1 - i], wimg[d][1][q + i]);
        }
        himg[d][q][0] = h0;
        himg[d][q][1] = h1;
      }
  };

  // D pass: slice of relative index t (R = t mod HL) feeds the output slice pairs t - i, i < HL (slot (R - i) mod HL)
  auto feed = [&](auto r_tag, const V2 (&himg)[2][CY][2]) {
    constexpr int R = decltype(r_tag)::value;
    pyr_static_for<HL>([&](auto i_tag) {
      constexpr int i = decltype(i_tag)::value, s = (R - i + HL) % HL;
#pragma unroll
      for (int q = 0; q < CY; ++q)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          if constexpr (i == 0) {
            acc[s][q][c][0] = amul_lo(a.tlo[HL - 1], himg[0][q][c]);
            acc[s][q][c][1] = amul_hi(a.tlo[HL - 1], himg[0][q][c]);
          } else {
            afma_lo(acc[s][q][c][0], a.tlo[HL - 1 - i], himg[0][q][c]);
            afma_hi(acc[s][q][c][1], a.tlo[HL - 1 - i], himg[0][q][c]);
          }
          afma_lo(acc[s][q][c][0], a.thi[HL - 1 - i], himg[1][q][c]);
          afma_hi(acc[s][q][c][1], a.thi[HL - 1 - i], himg[1][q][c]);
        }
    });
  };

  // output slice pair P from accumulator slot S
  auto emit = [&](auto s_tag, int P) {
    constexpr int S = decltype(s_tag)::value;
    if (MIFWT_DBG(a) & 1) return;
    if constexpr (ES == 4) {
    if (a.st16) {
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int z = 2 * P + r;
        if (z >= a.D) continue;
        const uint32_t so = (uint32_t)z * a.ys_d * 4u;
#pragma unroll
        for (int q = 0; q < CY; ++q) {
          const int n = 2 * (py0 + q) + (lane >> 5);  // lanes 0-31 store row 2 q of the group, lanes 32-63 row 2 q + 1
          const f4 t = r ? pyr_swap_rows(acc[S][q][0][0].y, acc[S][q][1][0].y, acc[S][q][0][1].y, acc[S][q][1][1].y)
                         : pyr_swap_rows(acc[S][q][0][0].x, acc[S][q][1][0].x, acc[S][q][0][1].x, acc[S][q][1][1].x);
          pyr_store4(t, yr, (n < a.H && c16 < 4u * (uint32_t)a.W) ? (uint32_t)n * a.ys_h * 4u + c16 : kPyrOob, so);
        }
      }
      return;
    }
    }
    if (!active) return;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int z = 2 * P + r;
      if (z >= a.D) continue;
#pragma unroll
      for (int q = 0; q < CY; ++q)
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
          const int n = 2 * (py0 + q) + rr;
          if (n >= a.H) continue;
          T* dst = yb + ((uint32_t)z * a.ys_d + (uint32_t)n * a.ys_h + (uint32_t)xo);
          const V2 v = {r ? acc[S][q][0][rr].y : acc[S][q][0][rr].x, r ? acc[S][q][1][rr].y : acc[S][q][1][rr].x};
          if (a.yvec && both) {
            if (a.nt) __builtin_nontemporal_store(v, reinterpret_cast<V2*>(dst));
            else *reinterpret_cast<V2*>(dst) = v;
          } else {
            dst[0] = v.x;
            if (both) dst[1] = v.y;
          }
        }
    }
  };

  for (int tb = 0; tb < nsl; tb += HL) {
    bool done = false;
    pyr_static_for<HL>([&](auto r_tag) {
      constexpr int R = decltype(r_tag)::value;
      const int t = tb + R;
      if (done || t >= nsl) {
        done = true;
        return;
      }
      V2 himg[2][CY][2];
      __syncthreads();
      filter_slice(himg);
      feed(r_tag, himg);
      if (t >= HL - 1) emit(std::integral_constant<int, (R + 1) % HL>{}, PA + t - (HL - 1));
    });
  }
}

template <typename T, int L, int CY, int NKB>
int launch_iwalk3(const mifwt_level_desc* d, const void* approx, const void* const* details, void* y, const double* lo, const double* hi,
                  hipStream_t stream) {
  constexpr int HL = L / 2;
  constexpr int SLOTB = 8 * NKB * 1024;
  constexpr int ES = (int)sizeof(T);
  IWalk3Args<T, L> a;
  for (int s = 0; s < 8; ++s) a.in[s] = static_cast<const T*>(s == 0 ? approx : details[s - 1]);
  a.is_b[0] = d->approx_stride[0];
  a.is_b[1] = d->detail_stride[0];
  a.is_d[0] = (uint32_t)d->approx_stride[1];
  a.is_d[1] = (uint32_t)d->detail_stride[1];
  a.y = static_cast<T*>(y);
  a.ys_b = d->sig_stride[0];
  a.ys_d = (uint32_t)d->sig_stride[1];
  a.ys_h = (uint32_t)d->sig_stride[2];
  a.Md = (int)d->coef_extent[0];
  a.Mh = (int)d->coef_extent[1];
  a.Mw = (int)d->coef_extent[2];
  a.D = (int)d->sig_extent[0];
  a.H = (int)d->sig_extent[1];
  a.W = (int)d->sig_extent[2];
  a.nt = g_options[MIFWT_OPT_NT_STORE];
  a.dbg = g_options[MIFWT_OPT_DEBUG] & 7;
  a.yvec = (a.ys_h % 2 == 0 && a.ys_d % 2 == 0 && a.ys_b % 2 == 0 && reinterpret_cast<uintptr_t>(y) % (2 * ES) == 0) ? 1 : 0;
  // 16-byte stores where every row piece of four columns is aligned and inside the row (MIFWT_OPT_DEBUG 512: never, as for kernel 16)
  {
    const int64_t span = (int64_t)(a.D - 1) * a.ys_d + (int64_t)(a.H - 1) * a.ys_h + a.W;
    a.st16 = (ES == 4 && a.W % 4 == 0 && a.ys_h % 4 == 0 && a.ys_d % 4 == 0 && a.ys_b % 4 == 0 && reinterpret_cast<uintptr_t>(y) % 16 == 0 &&
              span < (int64_t(1) << 30) && !(g_options[MIFWT_OPT_DEBUG] & 512)) ? 1 : 0;
  }
  for (int j = 0; j < HL; ++j) {
    a.tlo[j] = (typename TileArith<T>::vec2){(T)lo[2 * j], (T)lo[2 * j + 1]};
    a.thi[j] = (typename TileArith<T>::vec2){(T)hi[2 * j], (T)hi[2 * j + 1]};
  }
  a.nstrips = ((a.W + 1) / 2 + 63) / 64;
  a.ngroups = ((a.H + 1) / 2 + CY - 1) / CY;
  // staged slices: occupancy beats depth (config 3 finest level, 1 / 2 / 3 / 4 ahead: 199 / 197 / 247 / 250 us — from three ahead on
  // only one workgroup fits a CU; its 129^3 level: 40 / 50 / 48 us)
  const int64_t out_vol = d->sig_extent[0] * d->sig_extent[1] * d->sig_extent[2];
  int nslots = out_vol >= (int64_t(1) << 22) ? 3 : 2;
  if (g_options[MIFWT_OPT_PREFETCH_PAIRS] > 0) nslots = g_options[MIFWT_OPT_PREFETCH_PAIRS] + 1;
  if (nslots < 2) nslots = 2;
  if (nslots > 6) nslots = 6;
  while (nslots > 2 && (4 * NKB * (nslots - 1) > 63 || nslots * SLOTB > 150 * 1024)) --nslots;
  a.nslots = nslots;
  const size_t lds_bytes = (size_t)nslots * SLOTB;
  int ncu = 256;
  {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ncu = v;
  }
  int wpc = (int)((size_t)(160 * 1024) / lds_bytes);
  if (wpc > 8) wpc = 8;
  if (wpc < 1) wpc = 1;
  const int NP = (a.D + 1) / 2;
  const int64_t base = (int64_t)d->batch * a.ngroups;
  // depth segments: about two workgroups per slot, at least 8 output slice pairs each (a segment re-reads L/2 - 1 slices)
  int nseg = (int)((2 * (int64_t)ncu * wpc + base - 1) / base);
  if (nseg > NP / 8) nseg = NP / 8;
  if (g_options[MIFWT_OPT_ROWS_PER_CHUNK] > 0) nseg = (NP + g_options[MIFWT_OPT_ROWS_PER_CHUNK] - 1) / g_options[MIFWT_OPT_ROWS_PER_CHUNK];
  if (nseg < 1) nseg = 1;
  a.seg_out = (NP + nseg - 1) / nseg;
  a.nseg = (NP + a.seg_out - 1) / a.seg_out;
  a.div_g = make_fastdiv((uint32_t)a.ngroups);
  a.div_s = make_fastdiv((uint32_t)a.nseg);
  const int64_t nblk = base * a.nseg;
  if (nblk > INT32_MAX / 8) return MIFWT_ERR_UNSUPPORTED;
  static DynLdsOnce lds_once;
  if (!lds_once.ensure(reinterpret_cast<const void*>(&idwt3_walk_kernel<T, L, CY, NKB>), 6 * SLOTB > 160 * 1024 ? 160 * 1024 : 6 * SLOTB))
    return MIFWT_ERR_LAUNCH;
  hipLaunchKernelGGL((idwt3_walk_kernel<T, L, CY, NKB>), dim3((unsigned)nblk), dim3(64 * (a.nstrips + 2)), lds_bytes, stream, a);
  return hipGetLastError() == hipSuccess ? MIFWT_OK : MIFWT_ERR_LAUNCH;
}

constexpr int iwalk3_cy(int L, int ES) { return (L <= 6 && ES == 4) ? 4 : 2; }  // (f64: the accumulators of four row pairs would not fit)

template <typename T, int L>
int launch_iwalk3_l(const mifwt_level_desc* d, const void* approx, const void* const* details, void* y, const double* lo, const double* hi,
                    hipStream_t stream) {
  constexpr int CY = iwalk3_cy(L, (int)sizeof(T)), IY = CY + L / 2 - 1;
  const int64_t piece = (int64_t)IY * d->coef_extent[2] * (int64_t)sizeof(T);
  if (piece <= 1024) return launch_iwalk3<T, L, CY, 1>(d, approx, details, y, lo, hi, stream);
  if (piece <= 2048) return launch_iwalk3<T, L, CY, 2>(d, approx, details, y, lo, hi, stream);
  if (piece <= 3072) return launch_iwalk3<T, L, CY, 3>(d, approx, details, y, lo, hi, stream);
  if constexpr (sizeof(T) == 8) {  // (config 3's finest level in f64: three rows of 129 coefficients = 3096 bytes)
    if (piece <= 4096) return launch_iwalk3<T, L, CY, 4>(d, approx, details, y, lo, hi, stream);
  }
  if (piece <= 5120) return launch_iwalk3<T, L, CY, 5>(d, approx, details, y, lo, hi, stream);
  return MIFWT_ERR_UNSUPPORTED;
}

}  // namespace

bool dwt3_inv_walk_supported(const mifwt_level_desc* d) {
  if (d->ndim != 3 || (d->dtype != MIFWT_F32 && d->dtype != MIFWT_F64)) return false;
  const int ES = d->dtype == MIFWT_F64 ? 8 : 4;
  const int L = d->filt_len;
  if (L < 2 || L > 8 || (L & 1)) return false;
  if (d->sig_stride[3] != 1 || d->approx_stride[3] != 1 || d->detail_stride[3] != 1) return false;
  for (int i = 0; i < 3; ++i) {
    if (d->sig_stride[i] < 0 || d->approx_stride[i] < 0 || d->detail_stride[i] < 0) return false;
    if (d->coef_extent[i] < L / 2 || d->sig_extent[i] < 1 || d->sig_extent[i] > 2 * d->coef_extent[i] - L + 2) return false;
  }
  // dense coefficient rows: the rows of a band a row group needs are one contiguous piece per slice, at most 5 KiB
  if (d->approx_stride[2] != d->coef_extent[2] || d->detail_stride[2] != d->coef_extent[2]) return false;
  const int IY = iwalk3_cy(L, ES) + L / 2 - 1;
  if ((int64_t)IY * d->coef_extent[2] * ES > 5120 || d->sig_extent[2] > 128 * kIW3MaxStrips) return false;
  for (const int64_t* st : {d->approx_stride, d->detail_stride}) {
    const int64_t span = (d->coef_extent[0] - 1) * st[1] + d->coef_extent[1] * d->coef_extent[2];
    if (span >= (int64_t(1) << (ES == 8 ? 28 : 29)) || st[1] >= (int64_t(1) << 29)) return false;
  }
  // 32-bit element offsets inside one batch element of the output
  if (d->sig_extent[0] * d->sig_stride[1] >= (int64_t(1) << 31) || d->sig_stride[2] >= (int64_t(1) << 29)) return false;
  return true;
}

int dwt3_inv_walk(const mifwt_level_desc* d, const void* approx, const void* const* details, void* y, const double* lo, const double* hi,
                  hipStream_t stream) {
  if (d->dtype == MIFWT_F64) {
    switch (d->filt_len) {
      case 2: return launch_iwalk3_l<double, 2>(d, approx, details, y, lo, hi, stream);
      case 4: return launch_iwalk3_l<double, 4>(d, approx, details, y, lo, hi, stream);
      case 6: return launch_iwalk3_l<double, 6>(d, approx, details, y, lo, hi, stream);
      case 8: return launch_iwalk3_l<double, 8>(d, approx, details, y, lo, hi, stream);
      default: return MIFWT_ERR_UNSUPPORTED;
    }
  }
  switch (d->filt_len) {
    case 2: return launch_iwalk3_l<float, 2>(d, approx, details, y, lo, hi, stream);
    case 4: return launch_iwalk3_l<float, 4>(d, approx, details, y, lo, hi, stream);
    case 6: return launch_iwalk3_l<float, 6>(d, approx, details, y, lo, hi, stream);
    case 8: return launch_iwalk3_l<float, 8>(d, approx, details, y, lo, hi, stream);
    default: return MIFWT_ERR_UNSUPPORTED;
  }
}

}  // namespace mifwt
