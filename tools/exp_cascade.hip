// tools/exp_cascade.hip — lab for the cascade (config 4) kernel structure. Developer tool.
// variant 0: product structure (Chain<Lopass x8>, one sample through all stages at a time)
// variant 1: stage-skewed evaluation: at tick i stage s works on sample i-s, so the 8 stage updates
//            of a tick are mutually independent and are emitted op-by-op across stages (ILP 8).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <functional>
#include <string>
#include <vector>

#include "../madronalib_amd/csrc/mldsp_procs.hpp"

using namespace mldev;
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int NST = 8;
struct Args
{
  const float* coeffs;  // [24][V]
  uint32_t* state;      // [16][V]
  const f32x4* in;      // QUAD
  f32x4* out;           // QUAD
  size_t V, T;
};

__device__ __forceinline__ size_t xcd_block(size_t b, size_t nb)
{
  const size_t full = nb & ~(size_t)7;
  return (b < full) ? (b & 7) * (full >> 3) + (b >> 3) : b;
}

using CH = Chain<16, 16, 16, 16, 16, 16, 16, 16>;

template <int BLK>
__global__ __launch_bounds__(BLK) void k_v0(Args a)
{
  const size_t v = xcd_block(blockIdx.x, gridDim.x) * BLK + threadIdx.x;
  if (v >= a.V) return;
  CH ch;
  VoiceMem m{a.coeffs + v, a.state + v, a.V};
  KernelTables tb{nullptr};
  ch.load(m, tb);
  const f32x4* pi = a.in + v;
  f32x4* po = a.out + v;
#pragma unroll 4
  for (size_t r = 0; r < a.T * 16; ++r)
  {
    const f32x4 x = __builtin_nontemporal_load(pi + r * a.V);
    f32x4 y;
    y.x = ch.next(x.x);
    y.y = ch.next(x.y);
    y.z = ch.next(x.z);
    y.w = ch.next(x.w);
    __builtin_nontemporal_store(y, po + r * a.V);
  }
  ch.store(m);
}

// ---- skewed cascade -------------------------------------------------------------------------
struct Casc
{
  float g0[NST], g1[NST], g2[NST], ic1[NST], ic2[NST], r[NST];

  // all stages active: one tick; returns the last stage's output (sample i - (NST-1))
  __device__ __forceinline__ float tick(float x)
  {
    float in[NST], t0[NST], a[NST], b[NST], c[NST], d[NST], t1[NST], t2[NST];
    in[0] = x;
#pragma unroll
    for (int s = 1; s < NST; ++s) in[s] = r[s - 1];
#pragma unroll
    for (int s = 0; s < NST; ++s) t0[s] = in[s] - ic2[s];
#pragma unroll
    for (int s = 0; s < NST; ++s) b[s] = g1[s] * ic1[s];
#pragma unroll
    for (int s = 0; s < NST; ++s) d[s] = g0[s] * ic1[s];
#pragma unroll
    for (int s = 0; s < NST; ++s) a[s] = g0[s] * t0[s];
#pragma unroll
    for (int s = 0; s < NST; ++s) c[s] = g2[s] * t0[s];
#pragma unroll
    for (int s = 0; s < NST; ++s) t1[s] = a[s] + b[s];
#pragma unroll
    for (int s = 0; s < NST; ++s) t2[s] = c[s] + d[s];
#pragma unroll
    for (int s = 0; s < NST; ++s) r[s] = t2[s] + ic2[s];
#pragma unroll
    for (int s = 0; s < NST; ++s) ic1[s] = __builtin_fmaf(2.0f, t1[s], ic1[s]);
#pragma unroll
    for (int s = 0; s < NST; ++s) ic2[s] = __builtin_fmaf(2.0f, t2[s], ic2[s]);
    return r[NST - 1];
  }
  // boundary tick: only stages in [sLo, sHi] are active (wave-uniform runtime bounds)
  __device__ __forceinline__ float tick_masked(float x, int sLo, int sHi)
  {
    float in[NST];
    in[0] = x;
#pragma unroll
    for (int s = 1; s < NST; ++s) in[s] = r[s - 1];
#pragma unroll
    for (int s = 0; s < NST; ++s)
    {
      if (s >= sLo && s <= sHi)
      {
        const float t0 = in[s] - ic2[s];
        const float t1 = g0[s] * t0 + g1[s] * ic1[s];
        const float t2 = g2[s] * t0 + g0[s] * ic1[s];
        r[s] = t2 + ic2[s];
        ic1[s] = __builtin_fmaf(2.0f, t1, ic1[s]);
        ic2[s] = __builtin_fmaf(2.0f, t2, ic2[s]);
      }
    }
    return r[NST - 1];
  }
};

template <int BLK>
__global__ __launch_bounds__(BLK) void k_v1(Args a)
{
  const size_t v = xcd_block(blockIdx.x, gridDim.x) * BLK + threadIdx.x;
  if (v >= a.V) return;
  Casc c;
#pragma unroll
  for (int s = 0; s < NST; ++s)
  {
    c.g0[s] = a.coeffs[(size_t)(3 * s) * a.V + v];
    c.g1[s] = a.coeffs[(size_t)(3 * s + 1) * a.V + v];
    c.g2[s] = a.coeffs[(size_t)(3 * s + 2) * a.V + v];
    c.ic1[s] = u2f(a.state[(size_t)(2 * s) * a.V + v]);
    c.ic2[s] = u2f(a.state[(size_t)(2 * s + 1) * a.V + v]);
    c.r[s] = 0.f;
  }
  constexpr int D = NST - 1;          // output lag in ticks
  constexpr int A = D / 4, B = D % 4;  // D = 4A + B
  const size_t S = a.T * 64;
  const float* pin = (const float*)(a.in + v);   // sample i at pin[(i/4)*4V + i%4]
  float* pout = (float*)(a.out + v);
  const size_t rowF = a.V * 4;        // floats per quad row
  auto inAt = [&](size_t i) { return pin[(i >> 2) * rowF + (i & 3)]; };
  auto outAt = [&](size_t n, float y) { pout[(n >> 2) * rowF + (n & 3)] = y; };

  // prologue: ticks 0..D-1, stages 0..i
  for (int i = 0; i < D; ++i) c.tick_masked(inAt(i), 0, i);
  // steady: output quads q = 0..Q-1, ticks D+4q .. D+4q+3, inputs x[4(q+A)+B+j]
  const size_t Q = (S - D) / 4;
  f32x4 cur = __builtin_nontemporal_load(a.in + v + (size_t)A * a.V);  // quad A
  for (size_t q = 0; q < Q; ++q)
  {
    const f32x4 nxt = __builtin_nontemporal_load(a.in + v + (q + A + 1) * a.V);  // quad q+A+1 (< S/4 as D+4q+3 < S)
    f32x4 y;
#pragma unroll
    for (int j = 0; j < 4; ++j)
    {
      const int e = B + j;  // element index within cur (e<4) or nxt (e-4)
      const float x = (e < 4) ? cur[e & 3] : nxt[e & 3];
      y[j] = c.tick(x);
    }
    __builtin_nontemporal_store(y, a.out + v + q * a.V);
    cur = nxt;
  }
  // tail: remaining ticks D+4Q .. S+D-1 (inputs while i < S; stages > i-S stay active)
  for (size_t i = D + 4 * Q; i < S + D; ++i)
  {
    const float x = (i < S) ? inAt(i) : 0.f;
    const int sLo = (i < S) ? 0 : (int)(i - S + 1);
    const float y = c.tick_masked(x, sLo, NST - 1);
    outAt(i - D, y);
  }
#pragma unroll
  for (int s = 0; s < NST; ++s)
  {
    a.state[(size_t)(2 * s) * a.V + v] = f2u(c.ic1[s]);
    a.state[(size_t)(2 * s + 1) * a.V + v] = f2u(c.ic2[s]);
  }
}

template <class F>
float timeit(F f, int reps)
{
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps;
}

int main(int argc, char** argv)
{
  const size_t V = 131072, T = 32, n = V * T * 64;
  const int rounds = argc > 1 ? atoi(argv[1]) : 7;
  std::vector<float> co(24 * V), x(n);
  for (int s = 0; s < 8; ++s)
  {
    const float omega = 0.02f * (s + 1), k = 0.7f;
    const float piOmega = 3.14159265f * omega, s1 = sinf(piOmega), s2 = sinf(2.f * piOmega), nrm = 1.f / (2.f + k * s2);
    for (size_t v = 0; v < V; ++v)
    {
      co[(3 * s) * V + v] = s2 * nrm; co[(3 * s + 1) * V + v] = (-2.f * s1 * s1 - k * s2) * nrm; co[(3 * s + 2) * V + v] = (2.f * s1 * s1) * nrm;
    }
  }
  uint32_t seed = 12345;
  for (size_t i = 0; i < n; ++i) { seed = seed * 0x0019660Du + 0x3C6EF35Fu; uint32_t t = ((seed >> 9) & 0x7FFFFF) | 0x3F800000; float f; memcpy(&f, &t, 4); x[i] = f * 2.f - 3.f; }
  float* dco; uint32_t* dst; f32x4 *din, *out0, *out1;
  CK(hipMalloc(&dco, 96 * V)); CK(hipMalloc(&dst, 64 * V)); CK(hipMalloc(&din, 4 * n)); CK(hipMalloc(&out0, 4 * n)); CK(hipMalloc(&out1, 4 * n));
  CK(hipMemcpy(dco, co.data(), 96 * V, hipMemcpyHostToDevice)); CK(hipMemcpy(din, x.data(), 4 * n, hipMemcpyHostToDevice));
  std::vector<uint32_t> ref(n), got(n), st0(16 * V), st1(16 * V);
  struct Var { std::string name; std::function<void(Args)> launch; std::vector<float> ms; size_t bad; };
  std::vector<Var> vars;
  auto add = [&](const char* name, std::function<void(Args)> f) { vars.push_back({name, f, {}, 0}); };
  add("v0 chain blk256", [&](Args a) { hipLaunchKernelGGL(k_v0<256>, dim3(V / 256), dim3(256), 0, 0, a); });
  add("v0 chain blk64", [&](Args a) { hipLaunchKernelGGL(k_v0<64>, dim3(V / 64), dim3(64), 0, 0, a); });
  add("v1 skewed blk256", [&](Args a) { hipLaunchKernelGGL(k_v1<256>, dim3(V / 256), dim3(256), 0, 0, a); });
  add("v1 skewed blk128", [&](Args a) { hipLaunchKernelGGL(k_v1<128>, dim3(V / 128), dim3(128), 0, 0, a); });
  add("v1 skewed blk64", [&](Args a) { hipLaunchKernelGGL(k_v1<64>, dim3(V / 64), dim3(64), 0, 0, a); });
  for (size_t i = 0; i < vars.size(); ++i)
  {
    Args a{dco, dst, din, i == 0 ? out0 : out1, V, T};
    CK(hipMemset(dst, 0, 64 * V));
    vars[i].launch(a);
    vars[i].launch(a);  // second launch continues from carried state: checks the launch boundary
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(got.data(), a.out, 4 * n, hipMemcpyDeviceToHost));
    CK(hipMemcpy(st1.data(), dst, 64 * V, hipMemcpyDeviceToHost));
    if (i == 0) { ref = got; st0 = st1; }
    else
    {
      for (size_t j = 0; j < n; ++j) vars[i].bad += (got[j] != ref[j]);
      for (size_t j = 0; j < 16 * V; ++j) vars[i].bad += (st1[j] != st0[j]);
    }
  }
  for (int r = 0; r < rounds; ++r)
    for (auto& v : vars)
    {
      int k = 0;
      Args a0{dco, dst, din, out0, V, T}, a1{dco, dst, din, out1, V, T};
      v.ms.push_back(timeit([&] { v.launch((k++ & 1) ? a1 : a0); }, 10));
    }
  for (auto& v : vars)
  {
    std::sort(v.ms.begin(), v.ms.end());
    const float mn = v.ms.front(), md = v.ms[v.ms.size() / 2];
    printf("%-20s min %.4f ms (%.0f GB/s)  median %.4f ms (%.0f GB/s, %.3e smp/s)  mismatches %zu\n", v.name.c_str(), mn, 8.0 * n / mn / 1e6,
           md, 8.0 * n / md / 1e6, n / md * 1e3, v.bad);
  }
  return 0;
}
