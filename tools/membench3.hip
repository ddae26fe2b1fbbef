// tools/membench3.hip — does the ORDER in which a voice bank's output rows land in memory matter for the write ceiling?
// (no parity content). Layouts of a signal of V voices x `rows` quads:
//   L0  [row][V][4]                 the QUAD layout: a wavefront writes 1 KiB per row, rows 4 MiB apart
//   L1  [row/16][V/64][16][64][4]   wave-tiled vectors: a wavefront writes 16 KiB contiguous per DSPVector
//   L2  [V/64][row][64][4]          wave-major: each wavefront owns one contiguous stream
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int BLK, int LAYOUT, int WORK>
__global__ __launch_bounds__(BLK) void k_bank(f32x4* out, size_t V, size_t rows, float seed)
{
  size_t b = blockIdx.x;
  { size_t nb = gridDim.x; b = (b & 7) * (nb >> 3) + (b >> 3); }  // XCD-contiguous voices
  const size_t v = b * BLK + threadIdx.x;
  if (v >= V) return;
  float s[4];
  for (int i = 0; i < 4; ++i) s[i] = seed + (float)v + i;
  const size_t wave = v >> 6, lane = v & 63, nWaves = V >> 6;
  for (size_t r = 0; r < rows; ++r)
  {
    f32x4 y;
#pragma unroll
    for (int k = 0; k < 4; ++k)
    {
#pragma unroll
      for (int w = 0; w < WORK / 4; ++w)
#pragma unroll
        for (int i = 0; i < 4; ++i) s[i] = __builtin_fmaf(s[i], 1.0000001f, 0.5f);
      y[k] = (s[0] + s[1]) + (s[2] + s[3]);
    }
    size_t idx;
    if (LAYOUT == 0) idx = r * V + v;
    else if (LAYOUT == 1) idx = (((r >> 4) * nWaves + wave) * 16 + (r & 15)) * 64 + lane;
    else idx = (wave * rows + r) * 64 + lane;
    __builtin_nontemporal_store(y, out + idx);
  }
}
template <class F> float timeit(F f, int reps = 10)
{
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}
template <int BLK, int LAYOUT, int WORK>
void run(f32x4* b, size_t bytes, size_t V)
{
  size_t rows = bytes / 16 / V;
  float ms = timeit([&] { hipLaunchKernelGGL((k_bank<BLK, LAYOUT, WORK>), dim3(V / BLK), dim3(BLK), 0, 0, b, V, rows, 1.f); }, 25);
  printf("V=%7zu blk=%3d layout=%d work=%2d : %7.1f GB/s  (%.3f ms)\n", V, BLK, LAYOUT, WORK, bytes / ms / 1e6, ms);
}
int main()
{
  const size_t bytes = (size_t)2 << 30;
  f32x4* b; CK(hipMalloc(&b, bytes)); CK(hipMemset(b, 0, bytes));
  const size_t V = 262144;
  for (int rep = 0; rep < 2; ++rep)
  {
    run<256, 0, 0>(b, bytes, V);
    run<256, 1, 0>(b, bytes, V);
    run<256, 2, 0>(b, bytes, V);
    run<256, 0, 32>(b, bytes, V);
    run<256, 1, 32>(b, bytes, V);
    run<256, 2, 32>(b, bytes, V);
    run<64, 1, 32>(b, bytes, V);
    run<64, 2, 32>(b, bytes, V);
  }
  return 0;
}
