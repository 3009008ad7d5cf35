// ubench.hip — round-2 micro-benchmarks behind the design of the multi-level rolling kernel (standalone; hipcc -O3
// --offload-arch=gfx950 tools/ubench.hip -o tools/ubench).
//   1. LDS-DMA semantics (buffer_load_dword[x4] ... lds): lane layout, out-of-range lanes, M0 above 64 KiB, soffset
//   2. VALU issue rates: v_fma_f32 vs v_pk_fma_f32 (SGPR-pair operand with op_sel broadcast), 1..4 waves per SIMD
//   3. streaming skeleton of the solo-wave pyramid walk on config 2 (64 x 1024^2 f32, 3 levels): one wave = one
//      column strip of one row segment, rows arrive by LDS-DMA into a private ring, the three levels' planes are written
//      with the real store pattern; no arithmetic.  Gives the memory-side ceiling of that design.
//   4. plain float4 copy / read / write ceilings.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
// one LDS-DMA row piece: 64 lanes x 16 B -> LDS [lds_addr + 16 lane); global = rsrc.base + voff(lane) + soff
__device__ __forceinline__ void dma_x4(uint32_t voff, rsrc_t rsrc, uint32_t soff, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" ::"v"(voff), "s"(rsrc), "s"(lds_addr), "s"(soff) : "memory", "m0");
}
__device__ __forceinline__ void dma_x1(uint32_t voff, rsrc_t rsrc, uint32_t soff, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dword %0, %1, %3 offen lds" ::"v"(voff), "s"(rsrc), "s"(lds_addr), "s"(soff) : "memory", "m0");
}
__device__ __forceinline__ void dma_x4_nt(uint32_t voff, rsrc_t rsrc, uint32_t soff, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen nt lds" ::"v"(voff), "s"(rsrc), "s"(lds_addr), "s"(soff) : "memory", "m0");
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// ---------------------------------------------------------------------------------------------------------------------
// 1. DMA semantics
extern __shared__ __attribute__((aligned(16))) float dyn_lds[];
__global__ void k_dma_test(const float* src, uint32_t src_bytes, float* out, uint32_t lds_base, uint32_t soff) {
  const int lane = threadIdx.x;
  const rsrc_t rs = make_rsrc(src, src_bytes);
  // lanes 0..59: 16 B each at 16 * lane; lanes 60..63: out of range
  const uint32_t voff = lane < 60 ? 16u * lane : 0x80000000u;
  for (int i = lane; i < 512; i += 64) dyn_lds[lds_base / 4 + i] = -1.0f;
  __syncthreads();
  dma_x4(voff, rs, soff, lds_base);
  // dword DMA: lanes 0..15 -> floats 256.. of the slot, mirrored source (lane i reads element 15 - i)
  const uint32_t voff1 = lane < 16 ? 4u * (15 - lane) : (lane < 20 ? 0x80000000u : 4u * lane);
  dma_x1(voff1, rs, soff, lds_base + 1024);
  wait_vm<0>();
  __syncthreads();
  for (int i = lane; i < 512; i += 64) out[i] = dyn_lds[lds_base / 4 + i];
}

// ---------------------------------------------------------------------------------------------------------------------
// 2. VALU rates
template <int MODE>
__global__ void __launch_bounds__(256) k_valu(float* out, int iters, f2 tapA, f2 tapB) {
  f2 acc[8];
  float s[16];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = (f2){(float)threadIdx.x, (float)i};
#pragma unroll
  for (int i = 0; i < 16; ++i) s[i] = (float)(threadIdx.x + i);
  const f2 x = {1.0f + threadIdx.x * 1e-9f, 0.5f};
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {  // 16 independent v_fma_f32
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s[i]) : "s"(tapA.x), "v"(x.x));
    } else if (MODE == 1) {  // 8 independent v_pk_fma_f32 with an SGPR tap pair and a broadcast sample
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[i]) : "s"(tapA), "v"(x));
    } else if (MODE == 2) {  // v_pk_fma_f32 all-VGPR operands
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(x));
    } else if (MODE == 4) {  // the vertical-pass pattern: 8 accumulators x 8 distinct VGPR tap pairs x 2 samples, all distinct registers
      f2 t[8], xx[2];
#pragma unroll
      for (int i = 0; i < 8; ++i) { t[i] = (f2){tapA.x + i, tapA.y - i}; asm volatile("" : "+v"(t[i])); }
      xx[0] = x; xx[1] = (f2){x.y, x.x};
      asm volatile("" : "+v"(xx[0]), "+v"(xx[1]));
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[i]) : "v"(t[(i + r) & 7]), "v"(xx[r & 1]));
    } else if (MODE == 5) {  // the same with the taps in SGPR pairs
      f2 xx[2];
      xx[0] = x; xx[1] = (f2){x.y, x.x};
      asm volatile("" : "+v"(xx[0]), "+v"(xx[1]));
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[i]) : "s"((i + r) & 1 ? tapA : tapB), "v"(xx[r & 1]));
    } else if (MODE == 3) {  // v_fmac_f32 with DPP wave_shl:1 source
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(s[i]) : "v"(x.x), "v"(x.y));
    }
  }
  float t = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) t += acc[i].x + acc[i].y;
#pragma unroll
  for (int i = 0; i < 16; ++i) t += s[i];
  if (t == 12345.6789f) out[0] = t;
}

// ---------------------------------------------------------------------------------------------------------------------
// 3. solo-wave streaming skeleton, config 2 geometry: W0 = H0 = 1024, planes 515^2, 261^2, 134^2
struct SkelArgs {
  const float* x;
  float* d1;  // [B][3][515][515]
  float* d2;  // [B][3][261][261]
  float* o3;  // [B][4][134][134]
  int B, nseg, rows_per_seg, prologue;  // prologue: extra level-0 rows streamed before the segment
  int do_load, do_store;
};
// WG = NW waves = the NW column strips of one (image, segment); ring depth RD rows of 1 KiB (+ 128 B edge piece)
template <int NW, int RD>
__global__ void __launch_bounds__(64 * NW) k_skel(const SkelArgs a) {
  constexpr int SLOT = 1024 + 128;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int seg = blockIdx.x % a.nseg, img = blockIdx.x / a.nseg;
  const uint32_t my_lds = (uint32_t)wave * (RD * SLOT);
  const rsrc_t xr = make_rsrc(a.x + (size_t)img * 1024 * 1024, 1024u * 1024u * 4u);
  // strip k: level-3 columns [27 k, 27 k + 27) -> level-0 columns from 216 k - 42 (clamped, 16-B aligned)
  int c0 = 216 * wave - 44;
  c0 = c0 < 0 ? 0 : c0;
  c0 = c0 > 1024 - 256 ? 1024 - 256 : c0;
  const uint32_t voff = (uint32_t)(c0 + 4 * lane) * 4u;
  const uint32_t voff_e = lane < 24 ? (uint32_t)(((c0 + 256 + lane) & 1023)) * 4u : 0x80000000u;
  const int r0 = max(seg * a.rows_per_seg - a.prologue, 0), r1 = (seg + 1) * a.rows_per_seg;
  const rsrc_t r_d1[3] = {make_rsrc(a.d1 + ((size_t)img * 3 + 0) * 515 * 515, 515u * 515u * 4u), make_rsrc(a.d1 + ((size_t)img * 3 + 1) * 515 * 515, 515u * 515u * 4u),
                      make_rsrc(a.d1 + ((size_t)img * 3 + 2) * 515 * 515, 515u * 515u * 4u)};
  const rsrc_t r_d2[3] = {make_rsrc(a.d2 + ((size_t)img * 3 + 0) * 261 * 261, 261u * 261u * 4u), make_rsrc(a.d2 + ((size_t)img * 3 + 1) * 261 * 261, 261u * 261u * 4u),
                      make_rsrc(a.d2 + ((size_t)img * 3 + 2) * 261 * 261, 261u * 261u * 4u)};
  const rsrc_t r_o3[4] = {make_rsrc(a.o3 + ((size_t)img * 4 + 0) * 134 * 134, 134u * 134u * 4u), make_rsrc(a.o3 + ((size_t)img * 4 + 1) * 134 * 134, 134u * 134u * 4u),
                      make_rsrc(a.o3 + ((size_t)img * 4 + 2) * 134 * 134, 134u * 134u * 4u), make_rsrc(a.o3 + ((size_t)img * 4 + 3) * 134 * 134, 134u * 134u * 4u)};
  // owned columns: level 1 [108 k, 108 k + 108) (2 per lane, 54 lanes; the last strip 515 - 432 = 83), level 2 54 per strip, level 3 27
  const int n1 = wave == NW - 1 ? 515 - 108 * (NW - 1) : 108, n2 = wave == NW - 1 ? 261 - 54 * (NW - 1) : 54, n3 = wave == NW - 1 ? 134 - 27 * (NW - 1) : 27;
  const uint32_t so1 = 2 * lane + 1 < n1 ? (uint32_t)(108 * wave + 2 * lane) * 4u : 0x80000000u;
  const uint32_t so2 = lane < n2 ? (uint32_t)(54 * wave + lane) * 4u : 0x80000000u;
  const uint32_t so3 = lane < n3 ? (uint32_t)(27 * wave + lane) * 4u : 0x80000000u;
  const int nrows = r1 - r0;
  // prime the ring
  if (a.do_load) {
    for (int i = 0; i < RD - 1 && i < nrows; ++i) {
      dma_x4(voff, xr, (uint32_t)(r0 + i) * 4096u, my_lds + (uint32_t)(i % RD) * SLOT);
      dma_x1(voff_e, xr, (uint32_t)(r0 + i) * 4096u, my_lds + (uint32_t)(i % RD) * SLOT + 1024);
    }
  }
  f4 acc = {0, 0, 0, 0};
  for (int i = 0; i < nrows; ++i) {
    const int r = r0 + i;
    if (a.do_load) {
      if (i + RD - 1 < nrows) {
        dma_x4(voff, xr, (uint32_t)(r + RD - 1) * 4096u, my_lds + (uint32_t)((i + RD - 1) % RD) * SLOT);
        dma_x1(voff_e, xr, (uint32_t)(r + RD - 1) * 4096u, my_lds + (uint32_t)((i + RD - 1) % RD) * SLOT + 1024);
        wait_vm<2 * (RD - 1)>();
      } else {
        wait_vm<0>();
      }
      const float* row = &lds[(my_lds + (uint32_t)(i % RD) * SLOT) / 4];
      const f2 w0 = *reinterpret_cast<const f2*>(row + 4 * lane + 2);
      const f4 w1 = *reinterpret_cast<const f4*>(row + 4 * lane + 4);
      const f4 w2 = *reinterpret_cast<const f4*>(row + 4 * lane + 8);
      acc += w1 * w0.x + w2 * w0.y;
    }
    if (a.do_store && r >= seg * a.rows_per_seg) {
      if ((r & 1) == 1) {
        const uint32_t ro = (uint32_t)(r >> 1) * 515u * 4u;
        for (int b = 0; b < 3; ++b) __builtin_amdgcn_raw_buffer_store_b64((f2){acc.x + b, acc.y}, r_d1[b], so1, ro, 0);
      }
      if ((r & 3) == 3) {
        const uint32_t ro = (uint32_t)(r >> 2) * 261u * 4u;
        for (int b = 0; b < 3; ++b) __builtin_amdgcn_raw_buffer_store_b32(acc.z + b, r_d2[b], so2, ro, 0);
      }
      if ((r & 7) == 7) {
        const uint32_t ro = (uint32_t)(r >> 3) * 134u * 4u;
        for (int b = 0; b < 4; ++b) __builtin_amdgcn_raw_buffer_store_b32(acc.w + b, r_o3[b], so3, ro, 0);
      }
    }
  }
  if (acc.x == 1234.5678f) a.o3[0] = acc.y;
}

// 3b. the same traffic with a LOADER wave: wave NW issues every LDS-DMA of the workgroup (its vmcnt queue holds loads only),
// the NW compute waves read LDS and store (their vmcnt queue holds stores only and is never waited on); one barrier per
// step of 8 rows, double-buffered staging.
template <int NW, int VAR, int SUB, int NBUF>
__global__ void __launch_bounds__(64 * (NW + 1)) k_skel2(const SkelArgs a) {
  constexpr int SLOT = 1024 + 32, RD = SUB * NBUF;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int seg = blockIdx.x % a.nseg, img = blockIdx.x / a.nseg;
  const rsrc_t xr = make_rsrc(a.x + (size_t)img * 1024 * 1024, 1024u * 1024u * 4u);
  const int r0 = max(seg * a.rows_per_seg - a.prologue, 0) & ~7, r1 = (seg + 1) * a.rows_per_seg;
  const int nsteps = (r1 - r0) / SUB;
  if (wave == NW) {
    uint32_t voff[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      int c0 = w == 0 ? 0 : 256 + 208 * (w - 1) - 48;
      c0 = c0 > 1024 - 256 ? 1024 - 256 : c0;
      voff[w] = (uint32_t)(c0 + 4 * lane) * 4u;
    }
    auto issue = [&](int st) {
      const uint32_t buf = (uint32_t)(st % NBUF) * SUB * SLOT;
#pragma unroll
      for (int k = 0; k < SUB; ++k) {
        const uint32_t so = (uint32_t)(r0 + SUB * st + k) * 4096u;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
          if (VAR & 1) dma_x4_nt(voff[w], xr, so, (uint32_t)w * (RD * SLOT) + buf + (uint32_t)k * SLOT);
          else dma_x4(voff[w], xr, so, (uint32_t)w * (RD * SLOT) + buf + (uint32_t)k * SLOT);
        }
      }
    };
    for (int st = 0; st < NBUF - 1 && st < nsteps; ++st) issue(st);
    wait_vm<(NBUF - 2) * SUB * NW>();
    __syncthreads();
    for (int st = 0; st < nsteps; ++st) {
      if (st + NBUF - 1 < nsteps) {
        issue(st + NBUF - 1);  // its buffer was read in step st - 1
        wait_vm<(NBUF - 2) * SUB * NW>();
      } else {
        wait_vm<0>();
      }
      __syncthreads();
    }
    return;
  }
  const rsrc_t r_d1[3] = {make_rsrc(a.d1 + ((size_t)img * 3 + 0) * 544 * 515, 544u * 515u * 4u), make_rsrc(a.d1 + ((size_t)img * 3 + 1) * 544 * 515, 544u * 515u * 4u),
                          make_rsrc(a.d1 + ((size_t)img * 3 + 2) * 544 * 515, 544u * 515u * 4u)};
  const rsrc_t r_d2[3] = {make_rsrc(a.d2 + ((size_t)img * 3 + 0) * 261 * 261, 261u * 261u * 4u), make_rsrc(a.d2 + ((size_t)img * 3 + 1) * 261 * 261, 261u * 261u * 4u),
                          make_rsrc(a.d2 + ((size_t)img * 3 + 2) * 261 * 261, 261u * 261u * 4u)};
  const rsrc_t r_o3[4] = {make_rsrc(a.o3 + ((size_t)img * 4 + 0) * 134 * 134, 134u * 134u * 4u), make_rsrc(a.o3 + ((size_t)img * 4 + 1) * 134 * 134, 134u * 134u * 4u),
                          make_rsrc(a.o3 + ((size_t)img * 4 + 2) * 134 * 134, 134u * 134u * 4u), make_rsrc(a.o3 + ((size_t)img * 4 + 3) * 134 * 134, 134u * 134u * 4u)};
  // strip 0 owns level-3 columns [0, 32), strips k >= 1 own 26 each
  const int a3 = wave == 0 ? 0 : 32 + 26 * (wave - 1), b3 = min(134, wave == 0 ? 32 : 32 + 26 * wave);
  const int n3 = b3 - a3, n2 = min(261, 2 * b3) - 2 * a3, n1 = min(515, 4 * b3) - 4 * a3;
  constexpr int X4 = 0;
  constexpr int AUX = (VAR & 2) ? 2 : 0;
  constexpr uint32_t P1 = (VAR & 4) ? 544u : 515u;
  uint32_t so1;
  if (VAR & 4) so1 = lane < 48 ? (uint32_t)(96 * wave + 2 * lane) * 4u : 0x80000000u;
  else so1 = 2 * lane + 1 < n1 ? (uint32_t)(4 * a3 + 2 * lane) * 4u : 0x80000000u;
  const uint32_t so2 = lane < n2 ? (uint32_t)(2 * a3 + lane) * 4u : 0x80000000u;
  const uint32_t so3 = lane < n3 ? (uint32_t)(a3 + lane) * 4u : 0x80000000u;
  const uint32_t my_lds = (uint32_t)wave * (RD * SLOT);
  f4 acc = {0, 0, 0, 0};
  __syncthreads();
  for (int st = 0; st < nsteps; ++st) {
    const float* buf = &lds[(my_lds + (uint32_t)(st % NBUF) * SUB * SLOT) / 4];
#pragma unroll
    for (int kk = 0; kk < SUB; ++kk) {
      const int r = r0 + SUB * st + kk;
      const int k = r & 7;
      const int kslot = kk;
      const float* row = buf + kslot * (SLOT / 4);
      const f2 w0 = *reinterpret_cast<const f2*>(row + 4 * lane + 2);
      const f4 w1 = *reinterpret_cast<const f4*>(row + 4 * lane + 4);
      const f4 w2 = *reinterpret_cast<const f4*>(row + (4 * lane + 8 < 264 ? 4 * lane + 8 : 0));
      acc += w1 * w0.x + w2 * w0.y;
      if (a.do_store && r >= seg * a.rows_per_seg) {
        if (X4) {
          if ((k & 3) == 3) {
            const uint32_t ro = (uint32_t)((r >> 1) - 1 + (lane & 1)) * 515u * 4u;  // per-lane row: fold into the offset
            for (int b = 0; b < 3; ++b) __builtin_amdgcn_raw_buffer_store_b128(acc + (float)b, r_d1[b], so1 + ro, 0, 0);
          }
        } else if ((k & 1) == 1) {
          const uint32_t ro = (uint32_t)(r >> 1) * P1 * 4u;
          for (int b = 0; b < 3; ++b) __builtin_amdgcn_raw_buffer_store_b64((f2){acc.x + b, acc.y}, r_d1[b], so1, ro, AUX);
        }
        if ((k & 3) == 3) {
          const uint32_t ro = (uint32_t)(r >> 2) * 261u * 4u;
          for (int b = 0; b < 3; ++b) __builtin_amdgcn_raw_buffer_store_b32(acc.z + b, r_d2[b], so2, ro, AUX);
        }
        if (k == 7) {
          const uint32_t ro = (uint32_t)(r >> 3) * 134u * 4u;
          for (int b = 0; b < 4; ++b) __builtin_amdgcn_raw_buffer_store_b32(acc.w + b, r_o3[b], so3, ro, AUX);
        }
      }
    }
    __syncthreads();
  }
  if (acc.x == 1234.5678f) a.o3[0] = acc.y;
}

// ---------------------------------------------------------------------------------------------------------------------
// 4. plain ceilings
__global__ void k_copy(const f4* __restrict__ src, f4* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
template <int U>
__global__ void k_copy_u(const f4* __restrict__ src, f4* __restrict__ dst, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(&src[i + u * stride]);
#pragma unroll
    for (int u = 0; u < U; ++u) __builtin_nontemporal_store(v[u], &dst[i + u * stride]);
  }
  for (; i < n; i += stride) dst[i] = src[i];
}
__global__ void k_read(const f4* __restrict__ src, float* __restrict__ out, size_t n) {
  f4 acc = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += src[i];
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = 1.f;
}
__global__ void k_write(f4* __restrict__ dst, size_t n) {
  const f4 v = {1.f, 2.f, 3.f, 4.f};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = v;
}

template <typename F>
double time_ms(F f, int iters, double* best = nullptr) {
  hipEvent_t s, e;
  hipEventCreate(&s);
  hipEventCreate(&e);
  f();
  hipDeviceSynchronize();
  std::vector<float> ts;
  for (int rnd = 0; rnd < 7; ++rnd) {
    hipEventRecord(s);
    for (int i = 0; i < iters; ++i) f();
    hipEventRecord(e);
    hipEventSynchronize(e);
    float ms;
    hipEventElapsedTime(&ms, s, e);
    ts.push_back(ms / iters);
  }
  std::sort(ts.begin(), ts.end());
  if (best) *best = ts[0];
  return ts[ts.size() / 2];
}

template <int NW, int RD>
void run_skel(const char* name, SkelArgs a, float* const* xs, int nseg, int prologue, int do_load, int do_store) {
  a.nseg = nseg;
  a.rows_per_seg = 1024 / nseg;
  a.prologue = prologue;
  a.do_load = do_load;
  a.do_store = do_store;
  const size_t lds_bytes = (size_t)NW * RD * (1024 + 128);
  CK(hipFuncSetAttribute((const void*)k_skel<NW, RD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  int it = 0;
  double best;
  const double ms = time_ms([&] {
    a.x = xs[it++ % 3];
    hipLaunchKernelGGL((k_skel<NW, RD>), dim3(a.B * nseg), dim3(64 * NW), lds_bytes, 0, a);
  }, 20, &best);
  const double rd = 4.0 * 64 * 1024 * 1024, wr = 4.0 * 64 * (3.0 * 515 * 515 + 3.0 * 261 * 261 + 4.0 * 134 * 134);
  const double bytes = (do_load ? rd : 0) + (do_store ? wr : 0);
  printf("skel %-28s NW=%d RD=%2d nseg=%2d prol=%2d ld=%d st=%d  %.4f ms (best %.4f)  %.0f GB/s compulsory\n", name, NW, RD, nseg, prologue, do_load,
         do_store, ms, best, bytes / ms / 1e6);
  CK(hipGetLastError());
}

template <int NW, int VAR, int SUB, int NBUF>
void run_skel2(const char* name, SkelArgs a, float* const* xs, int nseg, int prologue, int do_store) {
  a.nseg = nseg;
  a.rows_per_seg = 1024 / nseg;
  a.prologue = prologue;
  a.do_load = 1;
  a.do_store = do_store;
  const size_t lds_bytes = (size_t)NW * SUB * NBUF * (1024 + 32);
  CK(hipFuncSetAttribute((const void*)k_skel2<NW, VAR, SUB, NBUF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  int it = 0;
  double best;
  const double ms = time_ms([&] {
    a.x = xs[it++ % 3];
    hipLaunchKernelGGL((k_skel2<NW, VAR, SUB, NBUF>), dim3(a.B * nseg), dim3(64 * (NW + 1)), lds_bytes, 0, a);
  }, 20, &best);
  const double rd = 4.0 * 64 * 1024 * 1024, wr = 4.0 * 64 * (3.0 * 515 * 515 + 3.0 * 261 * 261 + 4.0 * 134 * 134);
  const double bytes = rd + (do_store ? wr : 0);
  printf("skel2 (loader wave) %-16s NW=%d var=%d sub=%d nbuf=%d nseg=%2d prol=%2d st=%d  %.4f ms (best %.4f)  %.0f GB/s compulsory\n", name, NW, VAR, SUB, NBUF, nseg, prologue,
         do_store, ms, best, bytes / ms / 1e6);
  CK(hipGetLastError());
}

int main(int argc, char** argv) {
  const char* what = argc > 1 ? argv[1] : "all";
  const bool all = !strcmp(what, "all");
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s, CUs %d, clock %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);

  if (all || !strcmp(what, "dma")) {
    std::vector<float> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = (float)i;
    float *src, *out;
    CK(hipMalloc(&src, 4096 * 4));
    CK(hipMalloc(&out, 512 * 4));
    CK(hipMemcpy(src, h.data(), 4096 * 4, hipMemcpyHostToDevice));
    for (uint32_t base : {0u, 4096u, 65536u + 2048u, 131072u + 8192u}) {
      const size_t lds_bytes = base + 4096;
      CK(hipFuncSetAttribute((const void*)k_dma_test, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
      hipLaunchKernelGGL(k_dma_test, dim3(1), dim3(64), lds_bytes, 0, src, 4096u * 4u, out, base, 512u);
      CK(hipDeviceSynchronize());
      std::vector<float> o(512);
      CK(hipMemcpy(o.data(), out, 512 * 4, hipMemcpyDeviceToHost));
      // expected: floats 0..239 = 128 + i (soffset 512 B = 128 floats); 240..255: OOB lanes (0 or untouched -1);
      // 256..271 = 128 + 15 - i; 272..275 OOB; 276..319 = 128 + lane
      int bad = 0;
      for (int i = 0; i < 240; ++i) bad += o[i] != 128.0f + i;
      int bad1 = 0;
      for (int i = 0; i < 16; ++i) bad1 += o[256 + i] != 128.0f + 15 - i;
      for (int i = 20; i < 64; ++i) bad1 += o[256 + i] != 128.0f + i;
      printf("dma base=%6u: x4 mismatches %d, oob lanes -> [%g %g %g %g], x1 mismatches %d, x1 oob -> [%g %g], beyond [%g]\n", base, bad, o[240], o[241],
             o[254], o[255], bad1, o[256 + 16], o[256 + 19], o[256 + 64]);
    }
    hipFree(src);
    hipFree(out);
  }

  if (all || !strcmp(what, "valu")) {
    float* out;
    CK(hipMalloc(&out, 64));
    const int iters = 2000;
    for (int wps : {1, 2, 4}) {  // waves per SIMD = blocks of 256 threads per CU
      const int blocks = 256 * wps;
      const char* names[6] = {"v_fma_f32 (sgpr tap)", "v_pk_fma_f32 sgpr-pair op_sel", "v_pk_fma_f32 vgpr", "v_fmac_f32_dpp wave_shl", "v_pk_fma_f32 distinct vgpr taps", "v_pk_fma_f32 distinct, sgpr taps"};
      for (int mode = 0; mode < 6; ++mode) {
        double ms = 0;
        auto go = [&](auto kern) { ms = time_ms([&] { hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters, (f2){1.0f, 0.5f}, (f2){0.25f, 2.0f}); }, 3); };
        if (mode == 0) go(k_valu<0>);
        if (mode == 1) go(k_valu<1>);
        if (mode == 2) go(k_valu<2>);
        if (mode == 3) go(k_valu<3>);
        if (mode == 4) go(k_valu<4>);
        if (mode == 5) go(k_valu<5>);
        const double ninstr = 64.0 * iters;                       // per wave
        const double cyc = ms * 1e-3 * prop.clockRate * 1e3;      // at nominal clock
        printf("valu %-32s waves/SIMD=%d  %.4f ms  -> %.2f nominal cycles per wave-instruction per SIMD\n", names[mode], wps, ms, cyc / (ninstr * wps));
      }
    }
    hipFree(out);
  }

  const size_t n_in = (size_t)64 * 1024 * 1024;
  float* xs[3];
  float* dst[3];
  const size_t n_out = (size_t)64 * 4 * 560 * 560 + 64;
  if (all || !strcmp(what, "copy") || !strcmp(what, "skel")) {
    for (int i = 0; i < 3; ++i) {
      CK(hipMalloc(&xs[i], n_in * 4));
      CK(hipMalloc(&dst[i], n_in * 4));
      CK(hipMemset(xs[i], 1, n_in * 4));
      CK(hipMemset(dst[i], 0, n_in * 4));
    }
  }
  if (all || !strcmp(what, "copy")) {
    int it = 0;
    const size_t n4 = n_in / 4;
    for (int blocks : {1024, 2048, 4096, 8192, 16384}) {
      double best;
      double ms = time_ms([&] { hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, (const f4*)xs[it % 3], (f4*)dst[it % 3], n4); ++it; }, 10, &best);
      printf("copy_f4        blocks=%5d  %.4f ms (best %.4f)  %.0f GB/s (r+w)\n", blocks, ms, best, 2.0 * n_in * 4 / ms / 1e6);
    }
    for (int blocks : {1024, 2048, 4096}) {
      double best;
      double ms = time_ms([&] { hipLaunchKernelGGL(k_copy_u<4>, dim3(blocks), dim3(256), 0, 0, (const f4*)xs[it % 3], (f4*)dst[it % 3], n4); ++it; }, 10, &best);
      printf("copy_f4 nt u4  blocks=%5d  %.4f ms (best %.4f)  %.0f GB/s (r+w)\n", blocks, ms, best, 2.0 * n_in * 4 / ms / 1e6);
      ms = time_ms([&] { hipLaunchKernelGGL(k_copy_u<8>, dim3(blocks), dim3(256), 0, 0, (const f4*)xs[it % 3], (f4*)dst[it % 3], n4); ++it; }, 10, &best);
      printf("copy_f4 nt u8  blocks=%5d  %.4f ms (best %.4f)  %.0f GB/s (r+w)\n", blocks, ms, best, 2.0 * n_in * 4 / ms / 1e6);
    }
    {
      double best;
      double ms = time_ms([&] { hipLaunchKernelGGL(k_read, dim3(8192), dim3(256), 0, 0, (const f4*)xs[it % 3], dst[0], n4); ++it; }, 10, &best);
      printf("read_f4        blocks= 8192  %.4f ms (best %.4f)  %.0f GB/s\n", ms, best, 1.0 * n_in * 4 / ms / 1e6);
      ms = time_ms([&] { hipLaunchKernelGGL(k_write, dim3(8192), dim3(256), 0, 0, (f4*)dst[it % 3], n4); ++it; }, 10, &best);
      printf("write_f4       blocks= 8192  %.4f ms (best %.4f)  %.0f GB/s\n", ms, best, 1.0 * n_in * 4 / ms / 1e6);
    }
  }
  if (all || !strcmp(what, "skel")) {
    SkelArgs a;
    a.B = 64;
    a.d1 = dst[0];
    a.d2 = dst[1];
    a.o3 = dst[2];
    (void)n_out;
    run_skel2<5, 0, 4, 4>("full", a, xs, 4, 40, 1);
    run_skel2<5, 1, 4, 4>("nt-load", a, xs, 4, 40, 1);
    run_skel2<5, 2, 4, 4>("nt-store", a, xs, 4, 40, 1);
    run_skel2<5, 3, 4, 4>("nt-both", a, xs, 4, 40, 1);
    run_skel2<5, 4, 4, 4>("aligned-L1", a, xs, 4, 40, 1);
    run_skel2<5, 5, 4, 4>("aligned-L1 nt-load", a, xs, 4, 40, 1);
    run_skel2<5, 7, 4, 4>("aligned-L1 nt-both", a, xs, 4, 40, 1);
    run_skel2<5, 0, 4, 4>("loads only", a, xs, 4, 40, 0);
    run_skel2<5, 1, 4, 4>("nt loads only", a, xs, 4, 40, 0);
    return 0;
    run_skel<5, 16>("full", a, xs, 4, 42, 1, 1);
    run_skel<5, 16>("full no prologue", a, xs, 4, 0, 1, 1);
    run_skel<5, 16>("loads only", a, xs, 4, 42, 1, 0);
    run_skel<5, 16>("stores only", a, xs, 4, 42, 0, 1);
    run_skel<5, 8>("full", a, xs, 4, 42, 1, 1);
    run_skel<5, 24>("full", a, xs, 4, 42, 1, 1);
    run_skel<5, 16>("full 8 segs", a, xs, 8, 42, 1, 1);
    run_skel<5, 8>("full 8 segs", a, xs, 8, 42, 1, 1);
    run_skel<5, 8>("full 16 segs", a, xs, 16, 42, 1, 1);
    run_skel<5, 8>("full 16 segs no prologue", a, xs, 16, 0, 1, 1);
    run_skel<5, 4>("full 16 segs no prologue", a, xs, 16, 0, 1, 1);
    run_skel<5, 4>("full 32 segs no prologue", a, xs, 32, 0, 1, 1);
  }
  return 0;
}
