// Do 8-byte / 16-byte global stores (and loads) work at 2-byte aligned addresses on gfx950?  (f16 planes with an odd pitch.)
// hipcc --offload-arch=gfx950 -O2 tools/align_probe.hip -o /tmp/align_probe && /tmp/align_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef unsigned short u16;
typedef u16 u16x4 __attribute__((ext_vector_type(4)));
typedef u16 u16x8 __attribute__((ext_vector_type(8)));
struct __attribute__((packed, aligned(2))) P4 { u16x4 v; };
struct __attribute__((packed, aligned(2))) P8 { u16x8 v; };
__global__ void k_store(u16* out, int shift, int wide) {
  const int lane = threadIdx.x;
  u16* p = out + shift + lane * (wide ? 8 : 4) + 64;
  if (wide) {
    u16x8 v;
    for (int e = 0; e < 8; ++e) v[e] = (u16)(1000 + lane * 8 + e);
    asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
  } else {
    u16x4 v;
    for (int e = 0; e < 4; ++e) v[e] = (u16)(1000 + lane * 4 + e);
    asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
  }
}
__global__ void k_load(const u16* in, u16* out, int shift) {
  const int lane = threadIdx.x;
  const u16* p = in + shift + lane * 4 + 64;
  u16x4 v;
  asm volatile("global_load_dwordx2 %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  for (int e = 0; e < 4; ++e) out[lane * 4 + e] = v[e];
}
int main() {
  const int N = 4096;
  u16 *d, *d2;
  hipMalloc(&d, N * 2);
  hipMalloc(&d2, N * 2);
  std::vector<u16> h(N), h2(N);
  for (int wide = 0; wide < 2; ++wide)
    for (int shift = 0; shift < 4; ++shift) {
      hipMemset(d, 0, N * 2);
      hipLaunchKernelGGL(k_store, dim3(1), dim3(64), 0, 0, d, shift, wide);
      hipError_t e = hipDeviceSynchronize();
      hipMemcpy(h.data(), d, N * 2, hipMemcpyDeviceToHost);
      int bad = 0;
      const int per = wide ? 8 : 4;
      for (int i = 0; i < 64 * per; ++i) bad += h[64 + shift + i] != (u16)(1000 + i);
      for (int i = 0; i < 64 + shift; ++i) bad += h[i] != 0;
      printf("store x%d at +%d halfs (%d-byte aligned): %s, %d wrong\n", wide ? 4 : 2, shift, (shift * 2) % 4 == 0 ? 4 : 2, hipGetErrorString(e), bad);
    }
  for (int i = 0; i < N; ++i) h[i] = (u16)i;
  hipMemcpy(d, h.data(), N * 2, hipMemcpyHostToDevice);
  for (int shift = 0; shift < 4; ++shift) {
    hipLaunchKernelGGL(k_load, dim3(1), dim3(64), 0, 0, d, d2, shift);
    hipError_t e = hipDeviceSynchronize();
    hipMemcpy(h2.data(), d2, 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; ++i) bad += h2[i] != (u16)(64 + shift + i);
    printf("load x2 at +%d halfs: %s, %d wrong\n", shift, hipGetErrorString(e), bad);
  }
  return 0;
}
