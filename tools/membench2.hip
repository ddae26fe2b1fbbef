// tools/membench2.hip — store-pattern sweep for the voice-bank kernel (no parity content).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// MAP 0: voice = global thread id. MAP 1: XCD-contiguous: block b runs on XCD b%8; give XCD x the
// x-th eighth of the voices so each XCD's L2 sees one contiguous segment of every row.
template <int BLK, int MAP, bool NT, int WORK, int ILP>
__global__ __launch_bounds__(BLK) void k_bank(f32x4* out, size_t V, size_t rows, float seed)
{
  size_t b = blockIdx.x;
  if (MAP == 1) { size_t nb = gridDim.x; b = (b & 7) * (nb >> 3) + (b >> 3); }
  size_t v = b * BLK + threadIdx.x;
  if (v >= V) return;
  float s[ILP];
  for (int i = 0; i < ILP; ++i) s[i] = seed + (float)v + i;
  f32x4* p = out + v;
  for (size_t r = 0; r < rows; ++r)
  {
    f32x4 y;
#pragma unroll
    for (int k = 0; k < 4; ++k)
    {
#pragma unroll
      for (int w = 0; w < WORK / ILP; ++w)
#pragma unroll
        for (int i = 0; i < ILP; ++i) s[i] = __builtin_fmaf(s[i], 1.0000001f, 0.5f);
      float t = 0; for (int i = 0; i < ILP; ++i) t += s[i];
      y[k] = t;
    }
    if (NT) __builtin_nontemporal_store(y, p + r * V); else p[r * V] = y;
  }
}
template <class F> float timeit(F f, int reps = 10)
{
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}
template <int BLK, int MAP, bool NT, int WORK, int ILP>
void run(f32x4* b, size_t bytes, size_t V)
{
  size_t rows = bytes / 16 / V;
  float ms = timeit([&] { hipLaunchKernelGGL((k_bank<BLK, MAP, NT, WORK, ILP>), dim3(V / BLK), dim3(BLK), 0, 0, b, V, rows, 1.f); });
  printf("V=%7zu blk=%3d map=%d nt=%d work=%2d ilp=%d : %7.1f GB/s  (%.3f ms)\n", V, BLK, MAP, (int)NT, WORK, ILP, bytes / ms / 1e6, ms);
}
int main()
{
  const size_t bytes = (size_t)2 << 30;
  f32x4* b; CK(hipMalloc(&b, bytes)); CK(hipMemset(b, 0, bytes));
  const size_t V = 262144;
  run<64, 0, true, 0, 1>(b, bytes, V);
  run<64, 1, true, 0, 1>(b, bytes, V);
  run<128, 0, true, 0, 1>(b, bytes, V);
  run<128, 1, true, 0, 1>(b, bytes, V);
  run<256, 0, true, 0, 1>(b, bytes, V);
  run<256, 1, true, 0, 1>(b, bytes, V);
  run<64, 0, false, 0, 1>(b, bytes, V);
  run<64, 1, false, 0, 1>(b, bytes, V);
  // with work: 40 fma per sample, dependent (ilp1) or 4 chains
  run<64, 0, true, 40, 1>(b, bytes, V);
  run<64, 0, true, 40, 4>(b, bytes, V);
  run<64, 1, true, 40, 4>(b, bytes, V);
  run<64, 0, false, 40, 4>(b, bytes, V);
  run<256, 0, true, 40, 4>(b, bytes, V);
  run<256, 1, true, 40, 4>(b, bytes, V);
  run<64, 0, true, 24, 4>(b, bytes, V);
  run<64, 0, true, 80, 4>(b, bytes, 131072);
  run<64, 1, true, 80, 4>(b, bytes, 131072);
  run<256, 0, true, 80, 4>(b, bytes, 131072);
  return 0;
}
