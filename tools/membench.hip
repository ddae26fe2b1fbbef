// tools/membench.hip — HBM ceilings on the bench box for the access patterns this engine uses.
//   hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o /tmp/membench && /tmp/membench
// Prints GB/s for: float4 copy, read-only, write-only (plain / nontemporal), and the voice-bank
// store pattern (every lane walks rows V*16 B apart, one 16-byte store per row) without compute.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                                 \
  do                                                                          \
  {                                                                           \
    hipError_t e = (x);                                                       \
    if (e != hipSuccess)                                                      \
    {                                                                         \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
      exit(1);                                                                \
    }                                                                         \
  } while (0)

__global__ void k_copy(const f32x4* a, f32x4* b, size_t n)
{
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) b[i] = a[i];
}
__global__ void k_copy_nt(const f32x4* a, f32x4* b, size_t n)
{
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    __builtin_nontemporal_store(__builtin_nontemporal_load(&a[i]), &b[i]);
}
__global__ void k_read(const f32x4* a, float* sink, size_t n)
{
  size_t stride = (size_t)gridDim.x * blockDim.x;
  f32x4 acc = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) acc += a[i];
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) *sink = acc.x;
}
template <bool NT>
__global__ void k_write(f32x4* b, size_t n, float v)
{
  size_t stride = (size_t)gridDim.x * blockDim.x;
  f32x4 x = {v, v, v, v};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
  {
    if (NT)
      __builtin_nontemporal_store(x, &b[i]);
    else
      b[i] = x;
  }
}
// voice-bank pattern: lane = voice, rows = quads; row stride V float4; optional fake compute
template <bool NT, int WORK>
__global__ void k_bank(f32x4* out, size_t V, size_t rows, float seed)
{
  size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  float s = seed + (float)v;
  f32x4* p = out + v;
  for (size_t r = 0; r < rows; ++r)
  {
    f32x4 y;
#pragma unroll
    for (int k = 0; k < 4; ++k)
    {
#pragma unroll
      for (int w = 0; w < WORK; ++w) s = __builtin_fmaf(s, 1.0000001f, 0.5f);
      y[k] = s;
    }
    if (NT)
      __builtin_nontemporal_store(y, p + r * V);
    else
      p[r * V] = y;
  }
}

template <class F>
float timeit(F f, int reps = 10)
{
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  f();
  f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

int main()
{
  const size_t bytes = (size_t)2 << 30;  // 2 GiB per buffer: far beyond the 256 MiB Infinity Cache
  const size_t n = bytes / 16;
  f32x4 *a, *b;
  float* sink;
  CK(hipMalloc(&a, bytes));
  CK(hipMalloc(&b, bytes));
  CK(hipMalloc(&sink, 4));
  CK(hipMemset(a, 1, bytes));
  CK(hipMemset(b, 0, bytes));
  const int blocks = 256 * 8;
  float ms;
  ms = timeit([&] { hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, a, b, n); });
  printf("copy float4 (r+w)        %8.1f GB/s\n", 2.0 * bytes / ms / 1e6);
  ms = timeit([&] { hipLaunchKernelGGL(k_copy_nt, dim3(blocks), dim3(256), 0, 0, a, b, n); });
  printf("copy float4 nt (r+w)     %8.1f GB/s\n", 2.0 * bytes / ms / 1e6);
  ms = timeit([&] { hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, a, sink, n); });
  printf("read float4              %8.1f GB/s\n", 1.0 * bytes / ms / 1e6);
  ms = timeit([&] { hipLaunchKernelGGL((k_write<false>), dim3(blocks), dim3(256), 0, 0, b, n, 1.f); });
  printf("write float4             %8.1f GB/s\n", 1.0 * bytes / ms / 1e6);
  ms = timeit([&] { hipLaunchKernelGGL((k_write<true>), dim3(blocks), dim3(256), 0, 0, b, n, 1.f); });
  printf("write float4 nt          %8.1f GB/s\n", 1.0 * bytes / ms / 1e6);
  for (int blk : {64, 256})
  {
    const size_t V = 262144, rows = n / V;  // 512 rows = 32 DSPVectors
    const unsigned g = (unsigned)(V / blk);
    ms = timeit([&] { hipLaunchKernelGGL((k_bank<false, 0>), dim3(g), dim3(blk), 0, 0, b, V, rows, 1.f); });
    printf("bank store blk%-3d         %8.1f GB/s\n", blk, 1.0 * bytes / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL((k_bank<true, 0>), dim3(g), dim3(blk), 0, 0, b, V, rows, 1.f); });
    printf("bank store nt blk%-3d      %8.1f GB/s\n", blk, 1.0 * bytes / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL((k_bank<true, 8>), dim3(g), dim3(blk), 0, 0, b, V, rows, 1.f); });
    printf("bank store nt +8fma/smp   %8.1f GB/s (blk %d)\n", 1.0 * bytes / ms / 1e6, blk);
    ms = timeit([&] { hipLaunchKernelGGL((k_bank<true, 32>), dim3(g), dim3(blk), 0, 0, b, V, rows, 1.f); });
    printf("bank store nt +32fma/smp  %8.1f GB/s (blk %d)\n", 1.0 * bytes / ms / 1e6, blk);
    ms = timeit([&] { hipLaunchKernelGGL((k_bank<false, 32>), dim3(g), dim3(blk), 0, 0, b, V, rows, 1.f); });
    printf("bank store    +32fma/smp  %8.1f GB/s (blk %d)\n", 1.0 * bytes / ms / 1e6, blk);
  }
  // fewer voices per launch but more rows: does per-wave row count matter?
  {
    const size_t V = 131072, rows = n / V;
    ms = timeit([&] { hipLaunchKernelGGL((k_bank<true, 0>), dim3(V / 256), dim3(256), 0, 0, b, V, rows, 1.f); });
    printf("bank store nt V=131072    %8.1f GB/s\n", 1.0 * bytes / ms / 1e6);
    const size_t V2 = 1048576, rows2 = n / V2;
    ms = timeit([&] { hipLaunchKernelGGL((k_bank<true, 0>), dim3(V2 / 256), dim3(256), 0, 0, b, V2, rows2, 1.f); });
    printf("bank store nt V=1048576   %8.1f GB/s\n", 1.0 * bytes / ms / 1e6);
  }
  return 0;
}
