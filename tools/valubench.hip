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
typedef float f2 __attribute__((ext_vector_type(2)));
// packed FP32 (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32): two IEEE f32 results per lane per instruction
template <int ILP, int KIND>
__global__ __launch_bounds__(256) void kpk(float* out, int iters, float a, float b)
{
  f2 s[ILP];
  const f2 a2 = {a, a + 1e-7f}, b2 = {b, b + 1e-3f};
  for (int i = 0; i < ILP; ++i) s[i] = f2{a + threadIdx.x + i, b + threadIdx.x - i};
  for (int it = 0; it < iters; ++it)
  {
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int i = 0; i < ILP; ++i)
      {
        if (KIND == 0) s[i] = __builtin_elementwise_fma(s[i], a2, b2);
        if (KIND == 1) s[i] = s[i] * a2;
        if (KIND == 2) s[i] = s[i] + b2;
        if (KIND == 3) { s[i] = s[i] * a2; s[i] = s[i] + b2; }   // the no-contraction pair the parity build needs
      }
  }
  float t = 0;
  for (int i = 0; i < ILP; ++i) t += s[i].x + s[i].y;
  if (t == 1234.5f) out[0] = t;
}
// Every case is launched back to back for >= 40 ms after a 0.3 s warm-up (main): a single short launch after idle runs at
// whatever clock the chip is ramping through and says little (a 0.1-3 ms launch measured 36 T lane-instr/s where the
// sustained figure is 60+).
template <class F>
double timeLaunches(F launch)
{
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch(); CK(hipDeviceSynchronize());
  int reps = 1; float ms = 0;
  for (;;)
  {
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms >= 40.f) break;
    reps *= 2;
  }
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  return ms / reps;
}
template <int ILP, int KIND>
void runpk(float* out, int wavesPerSimd, const char* name, int opsPerStep)
{
  const int iters = 2000;
  const int blocks = 256 * wavesPerSimd;
  const double ms = timeLaunches([&] { hipLaunchKernelGGL((kpk<ILP, KIND>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0000001f, 0.5f); });
  double instPerWave = (double)iters * 16 * ILP * opsPerStep;
  double nsPerInstPerSimd = ms * 1e6 / instPerWave / wavesPerSimd;
  printf("%-14s ilp=%d waves/SIMD=%d : %.3f ms  %.2f ns/instr/SIMD  (%.1f T packed instr-lanes/s = %.1f T f32 results/s)\n", name, ILP,
         wavesPerSimd, ms, nsPerInstPerSimd, 1024.0 * 64 / nsPerInstPerSimd / 1e3, 2 * 1024.0 * 64 / nsPerInstPerSimd / 1e3);
}
template <int ILP, int KIND>
void run(float* out, int wavesPerSimd, const char* name, int opsPerStep)
{
  const int iters = 2000;
  const int blocks = 256 * wavesPerSimd;  // 256 CUs x (wavesPerSimd x 4 waves)/4 per block of 256
  const double ms = timeLaunches([&] { hipLaunchKernelGGL((k<ILP, KIND>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0000001f, 0.5f); });
  double instPerWave = (double)iters * 16 * ILP * opsPerStep;
  double nsPerInstPerWave = ms * 1e6 / instPerWave;                 // time between a wave's consecutive instrs
  double nsPerInstPerSimd = nsPerInstPerWave / wavesPerSimd;          // SIMD issue interval
  printf("%-14s ilp=%d waves/SIMD=%d : %.3f ms  %.2f ns/instr/wave  %.2f ns/instr/SIMD  (%.1f Tinstr-lanes/s)\n", name, ILP,
         wavesPerSimd, ms, nsPerInstPerWave, nsPerInstPerSimd, 1024.0 * 64 / nsPerInstPerSimd / 1e3);
}
int main()
{
  float* out; CK(hipMalloc(&out, 64));
  for (int i = 0; i < 100; ++i) hipLaunchKernelGGL((k<4, 0>), dim3(1024), dim3(256), 0, 0, out, 8000, 1.0000001f, 0.5f);   // ~0.3 s warm-up
  CK(hipDeviceSynchronize());
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
  for (int w : {1, 2, 4})
  {
    runpk<1, 0>(out, w, "pk_fma", 1);
    runpk<2, 0>(out, w, "pk_fma", 1);
    runpk<4, 0>(out, w, "pk_fma", 1);
    runpk<8, 0>(out, w, "pk_fma", 1);
  }
  runpk<4, 1>(out, 4, "pk_mul", 1);
  runpk<4, 2>(out, 4, "pk_add", 1);
  runpk<4, 3>(out, 4, "pk_mul+pk_add", 2);
  runpk<1, 3>(out, 4, "pk_mul+pk_add", 2);
  return 0;
}
