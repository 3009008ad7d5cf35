// membench.hip — streaming ceilings of the box the FWT kernels run on (standalone; hipcc -O3 --offload-arch=gfx950).
// Prints achieved GB/s of: float4 copy, read-only, write-only, copy with 4-byte-misaligned 16-byte stores,
// and a "strip" copy that mimics the dwt2 kernel's access pattern (1 KiB row segments in, 496 B segments out).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

__global__ void k_copy(const f4* __restrict__ src, f4* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
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
__global__ void k_copy_misaligned(const f4* __restrict__ src, float* __restrict__ dst, size_t n, int shift) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    *reinterpret_cast<f4u*>(dst + 4 * i + shift) = src[i];
}
// each wave: rows of 256 floats from a [B][1024][1024] image (column strip s of 4), writes 4 planes x 124 floats per 2 rows
__global__ void k_strip(const float* __restrict__ src, float* __restrict__ dst, int B, int rows_per_task) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int task = blockIdx.x * 4 + wave;
  const int nchunks = 1024 / rows_per_task;
  const int strip = task & 3, chunk = (task >> 2) % nchunks, img = task / (4 * nchunks);
  if (img >= B) return;
  const float* s = src + ((size_t)img * 1024 + (size_t)chunk * rows_per_task) * 1024 + strip * 256 + lane * 4;
  const int hrow = lane >> 5, q = lane & 31;
  float* d = dst + (size_t)img * 4 * 515 * 515 + ((size_t)chunk * (rows_per_task / 2)) * 515 + strip * 124 + q * 4;
  for (int r = 0; r < rows_per_task; r += 4) {
    f4 a = *reinterpret_cast<const f4*>(s + (size_t)(r + 0) * 1024);
    f4 b = *reinterpret_cast<const f4*>(s + (size_t)(r + 1) * 1024);
    f4 c = *reinterpret_cast<const f4*>(s + (size_t)(r + 2) * 1024);
    f4 e = *reinterpret_cast<const f4*>(s + (size_t)(r + 3) * 1024);
    if (q < 31) {
      float* o = d + (size_t)(r / 2 + hrow) * 515;
      *reinterpret_cast<f4u*>(o) = a;
      *reinterpret_cast<f4u*>(o + 515 * 515) = b;
      *reinterpret_cast<f4u*>(o + 2 * 515 * 515) = c;
      *reinterpret_cast<f4u*>(o + 3 * 515 * 515) = e;
    }
  }
}

// variants: RD = strip-pattern reads (else linear), WR: 0 = linear aligned, 1 = 4 planes x 515-pitch rows (496 B
// unaligned segments), 2 = 4 planes x 512-pitch rows (512 B aligned segments)
template <int RD, int WR>
__global__ void k_var(const float* __restrict__ src, float* __restrict__ dst, int B, int rows_per_task) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int task = blockIdx.x * 4 + wave;
  const int nchunks = 1024 / rows_per_task;
  const int strip = task & 3, chunk = (task >> 2) % nchunks, img = task / (4 * nchunks);
  if (img >= B) return;
  const float* s;
  size_t sstride;
  if (RD) { s = src + ((size_t)img * 1024 + (size_t)chunk * rows_per_task) * 1024 + strip * 256 + lane * 4; sstride = 1024; }
  else { s = src + (size_t)task * rows_per_task * 256 + lane * 4; sstride = 256; }
  const int hrow = lane >> 5, q = lane & 31;
  const int pitch = WR == 1 ? 515 : 512, seg = WR == 1 ? 124 : 128;
  float* d = dst + (size_t)img * 4 * pitch * pitch + ((size_t)chunk * (rows_per_task / 2)) * pitch + strip * seg + q * 4;
  float* dl = dst + (size_t)task * rows_per_task * 256 + lane * 4;
  for (int r = 0; r < rows_per_task; r += 4) {
    f4 a = *reinterpret_cast<const f4*>(s + (size_t)(r + 0) * sstride);
    f4 b = *reinterpret_cast<const f4*>(s + (size_t)(r + 1) * sstride);
    f4 c = *reinterpret_cast<const f4*>(s + (size_t)(r + 2) * sstride);
    f4 e = *reinterpret_cast<const f4*>(s + (size_t)(r + 3) * sstride);
    if (WR == 0) {
      *reinterpret_cast<f4*>(dl + (size_t)(r + 0) * 256) = a;
      *reinterpret_cast<f4*>(dl + (size_t)(r + 1) * 256) = b;
      *reinterpret_cast<f4*>(dl + (size_t)(r + 2) * 256) = c;
      *reinterpret_cast<f4*>(dl + (size_t)(r + 3) * 256) = e;
    } else if (WR == 2 || q < 31) {
      float* o = d + (size_t)(r / 2 + hrow) * pitch;
      *reinterpret_cast<f4u*>(o) = a;
      *reinterpret_cast<f4u*>(o + pitch * pitch) = b;
      *reinterpret_cast<f4u*>(o + 2 * pitch * pitch) = c;
      *reinterpret_cast<f4u*>(o + 3 * pitch * pitch) = e;
    }
  }
}

// write-pattern study: every task writes, for 4 planes and rows_per_task/2 rows, a run of RUN floats at float
// offset (row * PITCH + strip * RUN); MODE 0: lane q stores 16 B at run + 16q (4-byte aligned),
// MODE 1: 16-byte aligned body + scalar head/tail, MODE 2: 64-byte aligned body + scalar head/tail
template <int PITCH, int RUN, int MODE>
__global__ void k_wr(const float* __restrict__ src, float* __restrict__ dst, int B, int rows_per_task) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int task = blockIdx.x * 4 + wave;
  const int nchunks = 1024 / rows_per_task;
  const int strip = task & 3, chunk = (task >> 2) % nchunks, img = task / (4 * nchunks);
  if (img >= B) return;
  const float* s = src + ((size_t)img * 1024 + (size_t)chunk * rows_per_task) * 1024 + strip * 256 + lane * 4;
  const int hrow = lane >> 5, q = lane & 31;
  for (int r = 0; r < rows_per_task; r += 4) {
    f4 v[4];
    for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const f4*>(s + (size_t)(r + i) * 1024);
    const size_t row = (size_t)chunk * (rows_per_task / 2) + r / 2 + hrow;
    for (int pl = 0; pl < 4; ++pl) {
      const size_t off = ((size_t)img * 4 + pl) * PITCH * PITCH + row * PITCH + strip * RUN;  // float offset of the run
      float* run = dst + off;
      if (MODE == 0) {
        if (4 * q < RUN) *reinterpret_cast<f4u*>(run + 4 * q) = v[pl];
      } else {
        const int AL = MODE == 1 ? 4 : 16;                       // alignment in floats
        const int head = (int)((AL - (off & (AL - 1))) & (AL - 1));  // floats before the aligned body
        const int nbody = (RUN - head) / 4;                       // 16-byte units in the body
        if (q < nbody) *reinterpret_cast<f4*>(run + head + 4 * q) = v[pl];
        const int tail0 = head + 4 * nbody;
        if (q < head) run[q] = v[pl].x;
        if (q >= 16 && q - 16 < RUN - tail0) run[tail0 + q - 16] = v[pl].y;
      }
    }
  }
}

template <typename F>
double time_ms(F f, int iters) {
  hipEvent_t s, e;
  hipEventCreate(&s); hipEventCreate(&e);
  f();
  hipDeviceSynchronize();
  std::vector<float> ts;
  for (int rnd = 0; rnd < 5; ++rnd) {
    hipEventRecord(s);
    for (int i = 0; i < iters; ++i) f();
    hipEventRecord(e);
    hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    ts.push_back(ms / iters);
  }
  std::sort(ts.begin(), ts.end());
  return ts[ts.size() / 2];
}

int main() {
  const size_t n_in = (size_t)64 * 1024 * 1024;           // floats
  const size_t n_out = (size_t)64 * 4 * 560 * 560 + 64;   // floats
  float *src[3], *dst[3];
  for (int i = 0; i < 3; ++i) { hipMalloc(&src[i], n_in * 4); hipMalloc(&dst[i], n_out * 4); hipMemset(src[i], 1, n_in * 4); hipMemset(dst[i], 0, n_out * 4); }
  int it = 0;
  const size_t n4 = n_in / 4;
  for (int blocks : {2048, 4096, 8192, 16384}) {
    double ms = time_ms([&] { hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, (const f4*)src[it % 3], (f4*)dst[it % 3], n4); ++it; }, 10);
    printf("copy_f4        blocks=%5d  %.4f ms  %.0f GB/s (r+w)\n", blocks, ms, 2.0 * n_in * 4 / ms / 1e6);
  }
  {
    double ms = time_ms([&] { hipLaunchKernelGGL(k_read, dim3(8192), dim3(256), 0, 0, (const f4*)src[it % 3], dst[0], n4); ++it; }, 10);
    printf("read_f4        blocks= 8192  %.4f ms  %.0f GB/s\n", ms, 1.0 * n_in * 4 / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL(k_write, dim3(8192), dim3(256), 0, 0, (f4*)dst[it % 3], n4); ++it; }, 10);
    printf("write_f4       blocks= 8192  %.4f ms  %.0f GB/s\n", ms, 1.0 * n_in * 4 / ms / 1e6);
  }
  for (int shift : {0, 1, 2, 3}) {
    double ms = time_ms([&] { hipLaunchKernelGGL(k_copy_misaligned, dim3(8192), dim3(256), 0, 0, (const f4*)src[it % 3], dst[it % 3], n4, shift); ++it; }, 10);
    printf("copy_misalign  shift=%d      %.4f ms  %.0f GB/s (r+w)\n", shift, ms, 2.0 * n_in * 4 / ms / 1e6);
  }
  for (int rpt : {16, 32, 64, 128, 256}) {
    const int ntasks = 64 * 4 * (1024 / rpt);
    double ms = time_ms([&] { hipLaunchKernelGGL(k_strip, dim3(ntasks / 4), dim3(256), 0, 0, src[it % 3], dst[it % 3], 64, rpt); ++it; }, 10);
    const double bytes = 4.0 * n_in + 4.0 * 64 * 4 * 512 * 496;
    printf("strip rows/task=%3d          %.4f ms  %.0f GB/s (r+w)\n", rpt, ms, bytes / ms / 1e6);
  }
  for (int rpt : {32, 128}) {
    const int ntasks = 64 * 4 * (1024 / rpt);
    auto run = [&](const char* name, auto kern, double bytes) {
      double ms = time_ms([&] { hipLaunchKernelGGL(kern, dim3(ntasks / 4), dim3(256), 0, 0, src[it % 3], dst[it % 3], 64, rpt); ++it; }, 10);
      printf("%-34s rows/task=%3d  %.4f ms  %.0f GB/s (r+w)\n", name, rpt, ms, bytes / ms / 1e6);
    };
    run("linear read , linear write", k_var<0, 0>, 8.0 * n_in);
    run("strip read  , linear write", k_var<1, 0>, 8.0 * n_in);
    run("linear read , 515-pitch strip write", k_var<0, 1>, 4.0 * n_in + 4.0 * 64 * 4 * 512 * 496);
    run("strip read  , 515-pitch strip write", k_var<1, 1>, 4.0 * n_in + 4.0 * 64 * 4 * 512 * 496);
    run("strip read  , 512-pitch strip write", k_var<1, 2>, 8.0 * n_in);
    run("linear read , 512-pitch strip write", k_var<0, 2>, 8.0 * n_in);
  }
  {
    const int rpt = 32;
    const int ntasks = 64 * 4 * (1024 / rpt);
    auto run = [&](const char* name, auto kern, double bytes) {
      double ms = time_ms([&] { hipLaunchKernelGGL(kern, dim3(ntasks / 4), dim3(256), 0, 0, src[it % 3], dst[it % 3], 64, rpt); ++it; }, 10);
      printf("%-44s %.4f ms  %.0f GB/s (r+w)\n", name, ms, bytes / ms / 1e6);
    };
    const double b124 = 4.0 * n_in + 4.0 * 64 * 4 * 512 * 496, b128 = 8.0 * n_in;
    run("pitch 515 run 124 lane-store 4B-aligned", k_wr<515, 124, 0>, b124);
    run("pitch 515 run 124 16B-aligned body+head/tail", k_wr<515, 124, 1>, b124);
    run("pitch 515 run 124 64B-aligned body+head/tail", k_wr<515, 124, 2>, b124);
    run("pitch 515 run 128 lane-store 4B-aligned", k_wr<515, 128, 0>, b128);
    run("pitch 515 run 128 16B-aligned body+head/tail", k_wr<515, 128, 1>, b128);
    run("pitch 516 run 124 lane-store (16B aligned)", k_wr<516, 124, 0>, b124);
    run("pitch 516 run 128 lane-store (16B aligned)", k_wr<516, 128, 0>, b128);
    run("pitch 528 run 128 lane-store (64B aligned)", k_wr<528, 128, 0>, b128);
    run("pitch 512 run 128 lane-store (512B aligned)", k_wr<512, 128, 0>, b128);
    run("pitch 544 run 128 lane-store (128B aligned)", k_wr<544, 128, 0>, b128);
  }
  return 0;
}
