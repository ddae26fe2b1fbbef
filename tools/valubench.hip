// tools/valubench.hip — VALU issue rate / dependent latency on the bench box (no memory traffic).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int ILP, int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b)
{
  float s[ILP];
  for (int i = 0; i < ILP; ++i) s[i] = a + threadIdx.x + i;
  for (int it = 0; it < iters; ++it)
  {
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int i = 0; i < ILP; ++i)
      {
        if (KIND == 0) s[i] = __builtin_fmaf(s[i], a, b);
        if (KIND == 1) s[i] = s[i] * a;                       // mul
        if (KIND == 2) s[i] = s[i] + b;                       // add
        if (KIND == 3) s[i] = (s[i] > b) ? a : s[i] + 1.0f;   // cmp + cndmask + add
        if (KIND == 4) s[i] = (float)(int)(((unsigned)__float_as_uint(s[i])) >> 1) * a;  // lshr, cvt, mul
        if (KIND == 5) s[i] = __builtin_amdgcn_rcpf(s[i]);
      }
  }
  float t = 0;
  for (int i = 0; i < ILP; ++i) t += s[i];
  if (t == 1234.5f) out[0] = t;
}
template <int ILP, int KIND>
void run(float* out, int wavesPerSimd, const char* name, int opsPerStep)
{
  const int iters = 2000;
  const int blocks = 256 * wavesPerSimd;  // 256 CUs x (wavesPerSimd x 4 waves)/4 per block of 256
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<ILP, KIND>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0000001f, 0.5f);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<ILP, KIND>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0000001f, 0.5f);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double instPerWave = (double)iters * 16 * ILP * opsPerStep;
  double nsPerInstPerWave = ms * 1e6 / instPerWave;                 // time between a wave's consecutive instrs
  double nsPerInstPerSimd = nsPerInstPerWave / wavesPerSimd;          // SIMD issue interval
  printf("%-14s ilp=%d waves/SIMD=%d : %.3f ms  %.2f ns/instr/wave  %.2f ns/instr/SIMD  (%.1f Tinstr-lanes/s)\n", name, ILP,
         wavesPerSimd, ms, nsPerInstPerWave, nsPerInstPerSimd, 1024.0 * 64 / nsPerInstPerSimd / 1e3);
}
int main()
{
  float* out; CK(hipMalloc(&out, 64));
  for (int w : {1, 2, 4, 8})
  {
    run<1, 0>(out, w, "fma", 1);
    run<2, 0>(out, w, "fma", 1);
    run<4, 0>(out, w, "fma", 1);
    run<8, 0>(out, w, "fma", 1);
  }
  run<4, 1>(out, 4, "mul", 1);
  run<4, 2>(out, 4, "add", 1);
  run<4, 3>(out, 4, "cmp+cnd+add", 3);
  run<4, 4>(out, 4, "lshr+cvt+mul", 3);
  run<4, 5>(out, 4, "rcp", 1);
  run<1, 5>(out, 4, "rcp", 1);
  return 0;
}
