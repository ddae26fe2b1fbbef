// tools/copybench.hip — what a read+write stream can reach on this box, to price the kernels that stream a signal in and a
// signal out (config 4's cascade, the op kernels, resamplers): hipMemcpy D2D, grid-stride float4 copies (loads in flight
// 1..8, plain / nontemporal, XCD-aware or not), and the voice-bank pattern (one lane per voice walking rows V * 16 B apart,
// the QUAD layout) with 4 / 8 rows prefetched. 1 GiB in, 1 GiB out, sustained >= 40 ms each after a warm-up.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int U, bool NTL, bool NTS, bool XCD>
__global__ __launch_bounds__(256) void k_copy(const f32x4* a, f32x4* b, size_t n)
{
  size_t stride = (size_t)gridDim.x * 256, i = (size_t)blockIdx.x * 256 + threadIdx.x, end = n;
  if (XCD && (gridDim.x & 7) == 0)
  {
    const size_t per = (n / 8) & ~(size_t)255, x = blockIdx.x & 7;
    stride = (size_t)(gridDim.x >> 3) * 256;
    i = x * per + (size_t)(blockIdx.x >> 3) * 256 + threadIdx.x;
    end = (x == 7) ? n : (x + 1) * per;
  }
  for (; i + (U - 1) * stride < end; i += U * stride)
  {
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NTL ? __builtin_nontemporal_load(&a[i + u * stride]) : a[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u)
    {
      if (NTS) __builtin_nontemporal_store(v[u], &b[i + u * stride]); else b[i + u * stride] = v[u];
    }
  }
  for (; i < end; i += stride) b[i] = a[i];
}

// voice-bank pattern: lane = voice, row r of voice v at (r * V + v); PF rows of loads in flight
template <int PF, bool XCD>
__global__ __launch_bounds__(256) void k_bank(const f32x4* a, f32x4* b, size_t V, size_t rows)
{
  size_t blk = blockIdx.x;
  const size_t nbFull = (size_t)gridDim.x & ~(size_t)7;
  if (XCD && blk < nbFull) blk = (blk & 7) * (nbFull >> 3) + (blk >> 3);
  const size_t v = blk * 256 + threadIdx.x;
  if (v >= V) return;
  const f32x4* pa = a + v;
  f32x4* pb = b + v;
  f32x4 w[PF];
#pragma unroll
  for (int k = 0; k < PF; ++k) w[k] = __builtin_nontemporal_load(pa + (size_t)k * V);
  size_t r = 0;
  for (; r + 2 * PF <= rows; r += PF)
  {
    f32x4 nx[PF];
#pragma unroll
    for (int k = 0; k < PF; ++k) nx[k] = __builtin_nontemporal_load(pa + (r + PF + k) * V);
#pragma unroll
    for (int k = 0; k < PF; ++k) __builtin_nontemporal_store(w[k] * 1.0001f, pb + (r + k) * V);
#pragma unroll
    for (int k = 0; k < PF; ++k) w[k] = nx[k];
  }
#pragma unroll
  for (int k = 0; k < PF; ++k) if (r + k < rows) __builtin_nontemporal_store(w[k], pb + (r + k) * V);
}


// the same pattern with the loads and the stores in different wavefronts: a block of 256 threads takes 128 voices, its first
// two wavefronts load (one trip of PF rows ahead) and hand the rows over through LDS, its last two store. A wavefront that
// both loads and stores waits, at every s_waitcnt for a prefetched row, for its older stores to be acknowledged as well (one
// counter, in order); here the loaders never wait for a store. Twice the wavefronts for the same voices.
template <int PF, bool XCD>
__global__ __launch_bounds__(256) void k_bank_split(const f32x4* a, f32x4* b, size_t V, size_t rows)
{
  __shared__ f32x4 hand[2][PF][128];
  size_t blk = blockIdx.x;
  const size_t nbFull = (size_t)gridDim.x & ~(size_t)7;
  if (XCD && blk < nbFull) blk = (blk & 7) * (nbFull >> 3) + (blk >> 3);
  const unsigned l = threadIdx.x & 127;
  const bool loader = threadIdx.x < 128;
  const size_t v = blk * 128 + l;
  const f32x4* pa = a + v;
  f32x4* pb = b + v;
  const size_t trips = rows / PF;
  f32x4 w[PF];
  if (loader)
  {
#pragma unroll
    for (int k = 0; k < PF; ++k) w[k] = __builtin_nontemporal_load(pa + (size_t)k * V);
  }
  for (size_t t = 0; t <= trips; ++t)
  {
    if (loader)
    {
      f32x4 nx[PF];
      if (t + 1 < trips)
      {
#pragma unroll
        for (int k = 0; k < PF; ++k) nx[k] = __builtin_nontemporal_load(pa + ((t + 1) * PF + k) * V);
      }
      if (t < trips)
      {
#pragma unroll
        for (int k = 0; k < PF; ++k) hand[t & 1][k][l] = w[k] * 1.0001f;
      }
#pragma unroll
      for (int k = 0; k < PF; ++k) w[k] = nx[k];
    }
    else if (t > 0)
    {
#pragma unroll
      for (int k = 0; k < PF; ++k) __builtin_nontemporal_store(hand[(t - 1) & 1][k][l], pb + ((t - 1) * PF + k) * V);
    }
    __syncthreads();
  }
}

// one direction only: what a kernel that only writes (config 3: a generator bank) or only reads can reach. Grid-stride stream
// (U float4 per lane per trip) and the voice-bank pattern (lane = voice, rows V * 16 B apart, XCD-aware)
template <int U, bool STORE>
__global__ __launch_bounds__(256) void k_stream_one_way(f32x4* b, size_t n, float x)
{
  const size_t per = (n / 8) & ~(size_t)255, xcd = blockIdx.x & 7, stride = (size_t)(gridDim.x >> 3) * 256;
  size_t i = xcd * per + (size_t)(blockIdx.x >> 3) * 256 + threadIdx.x;
  const size_t end = (xcd == 7) ? n : (xcd + 1) * per;
  f32x4 acc = {x, x, x, x};
  for (; i + (U - 1) * stride < end; i += U * stride)
  {
#pragma unroll
    for (int u = 0; u < U; ++u)
    {
      if (STORE) __builtin_nontemporal_store(acc, &b[i + u * stride]);
      else acc += __builtin_nontemporal_load(&b[i + u * stride]);
    }
  }
  if (!STORE && acc[0] == 1234.5f) b[0] = acc;
}
template <bool STORE>
__global__ __launch_bounds__(256) void k_bank_one_way(f32x4* b, size_t V, size_t rows, float x)
{
  size_t blk = blockIdx.x;
  const size_t nbFull = (size_t)gridDim.x & ~(size_t)7;
  if (blk < nbFull) blk = (blk & 7) * (nbFull >> 3) + (blk >> 3);
  const size_t v = blk * 256 + threadIdx.x;
  if (v >= V) return;
  f32x4* pb = b + v;
  f32x4 acc = {x, x, x, x};
#pragma unroll 8
  for (size_t r = 0; r < rows; ++r)
  {
    if (STORE) __builtin_nontemporal_store(acc, pb + r * V);
    else acc += __builtin_nontemporal_load(pb + r * V);
  }
  if (!STORE && acc[0] == 1234.5f) b[0] = acc;
}

template <class F>
double sustain(F launch)
{
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch(); CK(hipDeviceSynchronize());
  int reps = 1; float ms = 0;
  for (;;)
  {
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms >= 40.f) break;
    reps *= 2;
  }
  return ms / reps;
}

int main()
{
  const size_t bytes = (size_t)1 << 30, n = bytes / 16;
  f32x4 *a, *b;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
  CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
  for (int i = 0; i < 30; ++i) hipLaunchKernelGGL((k_copy<4, true, true, true>), dim3(2048), dim3(256), 0, 0, a, b, n);
  CK(hipDeviceSynchronize());
  auto rep = [&](const char* name, double ms) { printf("%-58s %7.3f ms  %7.1f GB/s (read + write)\n", name, ms, 2.0 * bytes / ms / 1e6); fflush(stdout); };
  rep("hipMemcpyAsync device-to-device", sustain([&] { CK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0)); }));
#define RUN(U, NTL, NTS, XCD, GRID) rep("copy U=" #U " ntload=" #NTL " ntstore=" #NTS " xcd=" #XCD " grid=" #GRID, sustain([&] { hipLaunchKernelGGL((k_copy<U, NTL, NTS, XCD>), dim3(GRID), dim3(256), 0, 0, a, b, n); }))
  RUN(1, false, false, false, 2048); RUN(1, true, true, false, 2048); RUN(2, true, true, false, 2048); RUN(4, true, true, false, 2048); RUN(8, true, true, false, 2048);
  RUN(4, true, true, true, 2048); RUN(8, true, true, true, 2048); RUN(4, true, true, true, 1024); RUN(4, true, true, true, 4096); RUN(4, true, true, true, 8192);
  RUN(4, false, true, true, 2048); RUN(4, true, false, true, 2048); RUN(4, false, false, true, 2048); RUN(8, true, true, true, 4096);
  for (size_t V : {(size_t)131072, (size_t)262144, (size_t)1048576})
  {
    const size_t rows = n / V;
    char name[128];
#define BANK(PF, XCD) snprintf(name, sizeof(name), "bank pattern V=%zu rows=%zu prefetch=" #PF " xcd=" #XCD, V, rows); \
    rep(name, sustain([&] { hipLaunchKernelGGL((k_bank<PF, XCD>), dim3((unsigned)(V / 256)), dim3(256), 0, 0, a, b, V, rows); }))
    BANK(4, true); BANK(8, true); BANK(16, true); BANK(4, false);
#define SPLIT(PF) snprintf(name, sizeof(name), "bank pattern, loads / stores in separate waves V=%zu prefetch=" #PF, V); \
    rep(name, sustain([&] { hipLaunchKernelGGL((k_bank_split<PF, true>), dim3((unsigned)(V / 128)), dim3(256), 0, 0, a, b, V, rows); }))
    SPLIT(4); SPLIT(8); SPLIT(16);
  }
  // one direction: 2 GiB (a and b are not adjacent; each 1 GiB is walked on its own, two launches per measurement)
  auto rep1 = [&](const char* name, double ms) { printf("%-58s %7.3f ms  %7.1f GB/s (one direction)\n", name, ms, 2.0 * bytes / ms / 1e6); fflush(stdout); };
#define ONEWAY(U, STORE, GRID) rep1((STORE) ? "stores only, stream U=" #U " grid=" #GRID : "loads only, stream U=" #U " grid=" #GRID, sustain([&] { \
    hipLaunchKernelGGL((k_stream_one_way<U, STORE>), dim3(GRID), dim3(256), 0, 0, a, n, 1.f); hipLaunchKernelGGL((k_stream_one_way<U, STORE>), dim3(GRID), dim3(256), 0, 0, b, n, 1.f); }))
  ONEWAY(4, true, 2048); ONEWAY(8, true, 2048); ONEWAY(4, true, 4096); ONEWAY(4, false, 2048); ONEWAY(8, false, 2048); ONEWAY(8, false, 4096);
  for (size_t V : {(size_t)131072, (size_t)262144})
  {
    char name[128];
    const size_t rows = n / V;
    snprintf(name, sizeof(name), "stores only, bank pattern V=%zu rows=%zu", V, rows);
    rep1(name, sustain([&] { hipLaunchKernelGGL((k_bank_one_way<true>), dim3((unsigned)(V / 256)), dim3(256), 0, 0, a, V, rows, 1.f);
                             hipLaunchKernelGGL((k_bank_one_way<true>), dim3((unsigned)(V / 256)), dim3(256), 0, 0, b, V, rows, 1.f); }));
    snprintf(name, sizeof(name), "loads only, bank pattern V=%zu rows=%zu", V, rows);
    rep1(name, sustain([&] { hipLaunchKernelGGL((k_bank_one_way<false>), dim3((unsigned)(V / 256)), dim3(256), 0, 0, a, V, rows, 1.f);
                             hipLaunchKernelGGL((k_bank_one_way<false>), dim3((unsigned)(V / 256)), dim3(256), 0, 0, b, V, rows, 1.f); }));
  }
  return 0;
}
