// dma_probe.hip — round-3 questions about LDS-DMA (buffer_load_dwordx4 ... lds) on gfx950, behind the streaming synthesis kernel:
//   A. global address only 4-byte aligned (rows of 515 floats): does a 16-byte-per-lane request still deliver the right data?
//   B. lanes switched off in EXEC: do they leave their 16 bytes of LDS alone?
//   C. LDS base (M0) only 4-byte aligned?
//   D. dword-range checking: a lane whose 16 bytes straddle the end of the resource
// hipcc -O3 --offload-arch=gfx950 tools/dma_probe.hip -o tools/dma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <string.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void dma_x4(uint32_t voff, rsrc_t rsrc, uint32_t soff, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen nt lds" ::"v"(voff), "s"(rsrc), "s"(lds_addr), "s"(soff) : "memory", "m0");
}
extern __shared__ __attribute__((aligned(16))) float lds[];

// mode 0: misaligned global (byte offset goff), all lanes; mode 1: lanes >= nact masked by EXEC; mode 2: misaligned M0 (lds_off bytes)
__global__ void k_probe(const float* src, uint32_t src_bytes, float* out, int mode, uint32_t goff, int nact, uint32_t lds_off) {
  const int lane = threadIdx.x;
  for (int i = lane; i < 1024; i += 64) lds[i] = -7.0f;
  __syncthreads();
  const rsrc_t rs = make_rsrc(src, src_bytes);
  const uint32_t voff = 16u * lane + goff;
  if (mode == 1) {
    if (lane < nact) dma_x4(voff, rs, 0, 1024 + lds_off);
  } else {
    dma_x4(voff, rs, 0, 1024 + lds_off);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = lane; i < 1024; i += 64) out[i] = lds[i];
}

int main() {
  const int N = 4096;
  std::vector<float> h(N);
  for (int i = 0; i < N; ++i) h[i] = (float)i;
  float *d, *o;
  CK(hipMalloc(&d, N * 4));
  CK(hipMalloc(&o, 1024 * 4));
  CK(hipMemcpy(d, h.data(), N * 4, hipMemcpyHostToDevice));
  std::vector<float> r(1024);
  auto run = [&](int mode, uint32_t goff, int nact, uint32_t lds_off, uint32_t bytes) {
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 4096 + 64, 0, d, bytes, o, mode, goff, nact, lds_off);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(r.data(), o, 1024 * 4, hipMemcpyDeviceToHost));
  };
  // A
  for (uint32_t goff : {0u, 4u, 8u, 12u, 2060u, 4120u}) {
    run(0, goff, 64, 0, N * 4);
    int bad = 0;
    for (int i = 0; i < 256; ++i) bad += r[256 + i] != (float)(goff / 4 + i);
    printf("A global byte offset %5u: mismatches %d (first floats %g %g %g %g, before slot %g, after slot %g)\n", goff, bad, r[256], r[257], r[258], r[259], r[255], r[512]);
  }
  // B
  for (int nact : {10, 33, 63}) {
    run(1, 0, nact, 0, N * 4);
    int bad = 0, touched = 0;
    for (int i = 0; i < 4 * nact; ++i) bad += r[256 + i] != (float)i;
    for (int i = 4 * nact; i < 256; ++i) touched += r[256 + i] != -7.0f;
    printf("B exec: %2d active lanes: mismatches in their range %d, floats of inactive lanes overwritten %d (value at first inactive %g)\n", nact, bad, touched, r[256 + 4 * nact]);
  }
  // C
  for (uint32_t lo : {4u, 8u, 12u}) {
    run(2, 0, 64, lo, N * 4);
    int bad = 0;
    for (int i = 0; i < 256; ++i) bad += r[256 + lo / 4 + i] != (float)i;
    printf("C M0 offset %2u bytes: mismatches %d (floats at slot start %g %g %g %g %g)\n", lo, bad, r[256], r[257], r[258], r[259], r[260]);
  }
  // E: global address only 2-byte aligned (rows of f16 planes with an odd pitch)
  for (uint32_t goff : {2u, 6u, 10u, 14u, 8222u}) {
    run(0, goff, 64, 0, N * 4);
    const int bad = memcmp(&r[256], reinterpret_cast<const char*>(h.data()) + goff, 1024) != 0;
    printf("E global byte offset %5u (2-byte aligned): %s\n", goff, bad ? "WRONG data" : "right data");
  }
  // D: resource ends 8 bytes into lane 5's piece
  run(0, 0, 64, 0, 5 * 16 + 8);
  printf("D resource of %d bytes: lane 5 got %g %g %g %g, lane 6 got %g %g\n", 5 * 16 + 8, r[256 + 20], r[256 + 21], r[256 + 22], r[256 + 23], r[256 + 24], r[256 + 25]);
  return 0;
}
