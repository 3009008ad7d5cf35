// wbench.hip — write-path ceilings on gfx950: a streaming write of a big buffer with every cache-policy combination of the store
// instruction (sc0 / sc1 / nt), 16-byte and 8-byte stores; and the same for a streaming read and a copy.
// hipcc -O3 --offload-arch=gfx950 tools/wbench.hip -o tools/wbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int POL> __device__ __forceinline__ void st4(f4* p, f4 v) {
  if constexpr (POL == 0) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
  else if constexpr (POL == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
  else if constexpr (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
  else if constexpr (POL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  else if constexpr (POL == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
  else if constexpr (POL == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 nt" ::"v"(p), "v"(v) : "memory");
  else if constexpr (POL == 6) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
}
template <int POL> __device__ __forceinline__ f4 ld4(const f4* p) {
  f4 v;
  if constexpr (POL == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  else if constexpr (POL == 1) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
  else if constexpr (POL == 2) asm volatile("global_load_dwordx4 %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
  else if constexpr (POL == 3) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  else if constexpr (POL == 4) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
  else if constexpr (POL == 5) asm volatile("global_load_dwordx4 %0, %1, off sc0 nt" : "=v"(v) : "v"(p) : "memory");
  else if constexpr (POL == 6) asm volatile("global_load_dwordx4 %0, %1, off sc1 nt" : "=v"(v) : "v"(p) : "memory");
  else asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt" : "=v"(v) : "v"(p) : "memory");
  return v;
}
// 8-byte stores: a wave writes 512 B per instruction (what the analysis kernel's level-1 waves do per band row)
__global__ void __launch_bounds__(256) k_write8(f2* dst, size_t n_per_wg) {
  f2* p = dst + (size_t)blockIdx.x * n_per_wg + threadIdx.x;
  const f2 v = {1.f, (float)threadIdx.x};
  for (size_t i = 0; i < n_per_wg; i += 256) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p + i), "v"(v) : "memory");
}
// the analysis kernel's level-1 pattern: rows of W floats (W = 515: rows start on 4-byte boundaries), three planes, a wave writes 128
// columns of one row of one plane per instruction (8 bytes per lane), four waves side by side cover 512 columns
template <int BYTES> __global__ void __launch_bounds__(256) k_rows(float* dst, int W, int H, int planes) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // a workgroup owns rows [r0, r1) of all planes of one image
  const int img = blockIdx.x >> 2, seg = blockIdx.x & 3;
  const int r0 = seg * (H / 4), r1 = seg == 3 ? H : r0 + H / 4;
  float* base = dst + (size_t)img * planes * H * W;
  for (int r = r0; r < r1; ++r)
    for (int p = 0; p < planes; ++p) {
      float* row = base + ((size_t)p * H + r) * W;
      if (BYTES == 8) {
        const int c = 128 * wave + 2 * lane;
        if (c + 1 < W) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(row + c), "v"((f2){1.f, 2.f}) : "memory");
      } else {
        const int c = 256 * (wave & 1) + 4 * lane;  // two waves cover a row, the other two take the next plane's row
        float* rw = base + ((size_t)(p ^ (wave >> 1)) * H + r) * W;
        if (c + 3 < W) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(rw + c), "v"((f4){1.f, 2.f, 3.f, 4.f}) : "memory");
      }
    }
}
// each workgroup streams a contiguous span; per iteration a wave writes 1 KiB
template <int POL> __global__ void __launch_bounds__(256) k_write(f4* dst, size_t n_per_wg) {
  f4* p = dst + (size_t)blockIdx.x * n_per_wg + threadIdx.x;
  const f4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
  for (size_t i = 0; i < n_per_wg; i += 256) st4<POL>(p + i, v);
}
template <int POL> __global__ void __launch_bounds__(256) k_read(const f4* src, size_t n_per_wg, float* out) {
  const f4* p = src + (size_t)blockIdx.x * n_per_wg + threadIdx.x;
  f4 acc = {0, 0, 0, 0};
  for (size_t i = 0; i < n_per_wg; i += 1024) {
    f4 a = ld4<POL>(p + i), b = ld4<POL>(p + i + 256), c = ld4<POL>(p + i + 512), d = ld4<POL>(p + i + 768);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    acc += a + b + c + d;
  }
  if (acc.x == 123.456f) out[0] = acc.y;
}
template <int LP, int SP> __global__ void __launch_bounds__(256) k_copy(const f4* src, f4* dst, size_t n_per_wg) {
  const f4* p = src + (size_t)blockIdx.x * n_per_wg + threadIdx.x;
  f4* q = dst + (size_t)blockIdx.x * n_per_wg + threadIdx.x;
  for (size_t i = 0; i < n_per_wg; i += 1024) {
    f4 a = ld4<LP>(p + i), b = ld4<LP>(p + i + 256), c = ld4<LP>(p + i + 512), d = ld4<LP>(p + i + 768);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    st4<SP>(q + i, a); st4<SP>(q + i + 256, b); st4<SP>(q + i + 512, c); st4<SP>(q + i + 768, d);
  }
}
static const char* kPol[8] = {"-", "nt", "sc0", "sc1", "sc0 sc1", "sc0 nt", "sc1 nt", "sc0 sc1 nt"};

template <typename F> float timeit(F&& f, int reps = 12) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) f(i);
  CK(hipDeviceSynchronize());
  std::vector<float> t;
  for (int r = 0; r < 5; ++r) {
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) f(i);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms / reps);
  }
  std::sort(t.begin(), t.end());
  return t[2];
}

int main() {
  const size_t bytes = 256u << 20;  // one buffer: 256 MiB (config 2's input or output); three of each are rotated
  const size_t n = bytes / 16;
  f4 *src[3], *dst[3]; float* out;
  for (int i = 0; i < 3; ++i) { CK(hipMalloc(&src[i], bytes)); CK(hipMalloc(&dst[i], bytes)); CK(hipMemset(src[i], 1, bytes)); }
  CK(hipMalloc(&out, 64));
  for (int nwg : {256, 1024, 4096}) {
    const size_t per = n / nwg;
    printf("== %d workgroups of 256 threads, %zu KiB contiguous each\n", nwg, per * 16 / 1024);
#define RUNW(P) { float ms = timeit([&](int i) { hipLaunchKernelGGL(k_write<P>, dim3(nwg), dim3(256), 0, 0, dst[i % 3], per); }); printf("write  [%-10s] %.4f ms  %6.0f GB/s\n", kPol[P], ms, bytes / ms / 1e6); }
    RUNW(0) RUNW(1) RUNW(2) RUNW(3) RUNW(4) RUNW(5) RUNW(6) RUNW(7)
#define RUNR(P) { float ms = timeit([&](int i) { hipLaunchKernelGGL(k_read<P>, dim3(nwg), dim3(256), 0, 0, src[i % 3], per, out); }); printf("read   [%-10s] %.4f ms  %6.0f GB/s\n", kPol[P], ms, bytes / ms / 1e6); }
    RUNR(0) RUNR(1) RUNR(4) RUNR(7)
#define RUNC(LP, SP) { float ms = timeit([&](int i) { hipLaunchKernelGGL((k_copy<LP, SP>), dim3(nwg), dim3(256), 0, 0, src[i % 3], dst[i % 3], per); }); printf("copy   [ld %-10s st %-10s] %.4f ms  %6.0f GB/s (r+w)\n", kPol[LP], kPol[SP], ms, 2.0 * bytes / ms / 1e6); }
    RUNC(0, 0) RUNC(1, 0) RUNC(1, 1) RUNC(1, 4) RUNC(1, 7) RUNC(7, 7) RUNC(1, 3) RUNC(1, 2)
  }
  { const int nwg = 1024; const size_t per8 = bytes / 8 / nwg;
    float ms = timeit([&](int i) { hipLaunchKernelGGL(k_write8, dim3(nwg), dim3(256), 0, 0, (f2*)dst[i % 3], per8); }); printf("write 8-byte stores, 1024 workgroups: %.4f ms %6.0f GB/s\n", ms, bytes / ms / 1e6);
    // 60 images x 4 planes x 512 rows x 516 floats = 254 MB (inside the 256 MiB buffers)
    const int W = 515, H = 512, P = 4, NI = 60; const double rb = (double)NI * P * H * W * 4;
    ms = timeit([&](int i) { hipLaunchKernelGGL(k_rows<8>, dim3(4 * NI), dim3(256), 0, 0, (float*)dst[i % 3], W, H, P); }); printf("rows of 515 floats, 8-byte stores, 128 columns per wave:  %.4f ms %6.0f GB/s\n", ms, rb / ms / 1e6);
    ms = timeit([&](int i) { hipLaunchKernelGGL(k_rows<16>, dim3(4 * NI), dim3(256), 0, 0, (float*)dst[i % 3], W, H, P); }); printf("rows of 515 floats, 16-byte stores, 256 columns per wave: %.4f ms %6.0f GB/s\n", ms, rb / ms / 1e6);
    const int W2 = 516;
    ms = timeit([&](int i) { hipLaunchKernelGGL(k_rows<8>, dim3(4 * NI), dim3(256), 0, 0, (float*)dst[i % 3], W2, H, P); }); printf("rows of 516 floats, 8-byte stores:  %.4f ms %6.0f GB/s\n", ms, (double)NI * P * H * W2 * 4 / ms / 1e6);
    ms = timeit([&](int i) { hipLaunchKernelGGL(k_rows<16>, dim3(4 * NI), dim3(256), 0, 0, (float*)dst[i % 3], W2, H, P); }); printf("rows of 516 floats, 16-byte stores: %.4f ms %6.0f GB/s\n", ms, (double)NI * P * H * W2 * 4 / ms / 1e6);
  }
  // the same output buffer every time (what a benchmark loop over one allocation does)
  { const int nwg = 1024; const size_t per = n / nwg;
    float ms = timeit([&](int i) { hipLaunchKernelGGL(k_write<0>, dim3(nwg), dim3(256), 0, 0, dst[0], per); }); printf("write same buffer [-] %.4f ms %6.0f GB/s\n", ms, bytes / ms / 1e6);
    ms = timeit([&](int i) { hipLaunchKernelGGL((k_copy<1, 0>), dim3(nwg), dim3(256), 0, 0, src[i % 3], dst[0], per); }); printf("copy into the same buffer [ld nt st -] %.4f ms %6.0f GB/s (r+w)\n", ms, 2.0 * bytes / ms / 1e6);
  }
  return 0;
}
