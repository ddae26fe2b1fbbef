// tools/exp_chain.hip — kernel-structure lab for the voice-bank kernel (developer tool, not product).
// Builds variants of the SawGen->Bandpass->gain loop from the PRODUCT device headers and times
// them on the config-3 workload; every variant's output is compared bit-for-bit with variant 0.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp tools/exp_chain.hip -o tools/bin/exp_chain
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../madronalib_amd/csrc/mldsp_procs.hpp"

using namespace mldev;
typedef float f32x4 __attribute__((ext_vector_type(4)));
// -DEXP_TURNS=1: the wavefronts of a SIMD take turns at the priority levels by the shared clock (round 4's take_turns_by_clock), every 4 quads
#ifndef EXP_TURNS
#define EXP_TURNS 0
#endif
#if EXP_TURNS
#define EXP_TURN(r) do { if (((r) & 3) == 0) take_turns_by_clock(wave_slot(), 13); } while (0)
#else
#define EXP_TURN(r) do { } while (0)
#endif
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

struct Args
{
  const float* coeffs;  // [4][V]: g0,g1,g2,gain
  uint32_t* state;      // [3][V]
  const float* freq;    // [V]
  f32x4* out;           // QUAD
  size_t V, T;
};

__device__ __forceinline__ size_t xcd_block(size_t b, size_t nb)
{
  const size_t full = nb & ~(size_t)7;
  return (b < full) ? (b & 7) * (full >> 3) + (b >> 3) : b;
}

using CH = Chain<MLGPU_PROC_SAW_GEN, MLGPU_PROC_BANDPASS, MLGPU_PROC_GAIN>;

// variant 0: product structure (1 voice per lane)
template <int BLK>
__global__ __launch_bounds__(BLK) void k_v0(Args a)
{
  const size_t v = xcd_block(blockIdx.x, gridDim.x) * BLK + threadIdx.x;
  if (v >= a.V) return;
  CH ch;
  VoiceMem m{a.coeffs + v, a.state + v, a.V};
  KernelTables tb{nullptr};
  ch.load(m, tb);
  const float xc = a.freq[v];
  f32x4* po = a.out + v;
  for (size_t r = 0; r < a.T * 16; ++r)
  {
    EXP_TURN(r);
    f32x4 y;
    y.x = ch.next_head<true>(xc);
    y.y = ch.next_head<true>(xc);
    y.z = ch.next_head<true>(xc);
    y.w = ch.next_head<true>(xc);
    __builtin_nontemporal_store(y, po + r * a.V);
  }
  ch.store(m);
}

// variant 0b: product loop shape: T loop x 16 quads with UNROLL, generic runtime strides
template <int BLK, int UNROLL, bool NT>
__global__ __launch_bounds__(BLK) void k_v0b(Args a, size_t strideT, size_t strideQ, size_t strideV)
{
  const size_t v = xcd_block(blockIdx.x, gridDim.x) * BLK + threadIdx.x;
  if (v >= a.V) return;
  CH ch;
  VoiceMem m{a.coeffs + v, a.state + v, a.V};
  KernelTables tb{nullptr};
  ch.load(m, tb);
  const float xc = a.freq[v];
  f32x4* pout = a.out + v * strideV;
  for (size_t t = 0; t < a.T; ++t)
  {
    f32x4* po = pout + t * strideT;
#pragma unroll UNROLL
    for (int q = 0; q < 16; ++q)
    {
      f32x4 y;
      y.x = ch.next_head<true>(xc);
      y.y = ch.next_head<true>(xc);
      y.z = ch.next_head<true>(xc);
      y.w = ch.next_head<true>(xc);
      if (NT) __builtin_nontemporal_store(y, po + q * strideQ); else po[q * strideQ] = y;
    }
  }
  ch.store(m);
}

// variant 1: two voices per lane (v, v+64 inside a 128-voice group), calls alternated per sample
template <int BLK>
__global__ __launch_bounds__(BLK) void k_v1(Args a)
{
  const size_t g = xcd_block(blockIdx.x, gridDim.x) * BLK + threadIdx.x;  // lane id over V/2
  const size_t va = (g >> 6) * 128 + (g & 63), vb = va + 64;
  if (vb >= a.V) return;
  CH ca, cb;
  VoiceMem ma{a.coeffs + va, a.state + va, a.V}, mb{a.coeffs + vb, a.state + vb, a.V};
  KernelTables tb{nullptr};
  ca.load(ma, tb);
  cb.load(mb, tb);
  const float xa = a.freq[va], xb = a.freq[vb];
  f32x4 *pa = a.out + va, *pb = a.out + vb;
  for (size_t r = 0; r < a.T * 16; ++r)
  {
    EXP_TURN(r);
    f32x4 ya, yb;
    ya.x = ca.next_head<true>(xa); yb.x = cb.next_head<true>(xb);
    ya.y = ca.next_head<true>(xa); yb.y = cb.next_head<true>(xb);
    ya.z = ca.next_head<true>(xa); yb.z = cb.next_head<true>(xb);
    ya.w = ca.next_head<true>(xa); yb.w = cb.next_head<true>(xb);
    __builtin_nontemporal_store(ya, pa + r * a.V);
    __builtin_nontemporal_store(yb, pb + r * a.V);
  }
  ca.store(ma);
  cb.store(mb);
}

// variant 2: N voices per lane with every arithmetic step written N-wide (explicit instruction-
// level interleave): hand-expanded SawGen->Bandpass->gain on arrays.
template <int BLK, int N>
__global__ __launch_bounds__(BLK) void k_v2(Args a)
{
  const size_t g = xcd_block(blockIdx.x, gridDim.x) * BLK + threadIdx.x;
  const size_t v0 = (g >> 6) * (64 * N) + (g & 63);
  if (v0 + 64 * (N - 1) >= a.V) return;
  uint32_t om[N], istep[N];
  float dt[N], omdt[N], r1[N], ndt[N], g0[N], g1[N], g2[N], gain[N], ic1[N], ic2[N];
#pragma unroll
  for (int i = 0; i < N; ++i)
  {
    const size_t v = v0 + 64 * i;
    g0[i] = a.coeffs[v]; g1[i] = a.coeffs[a.V + v]; g2[i] = a.coeffs[2 * a.V + v]; gain[i] = a.coeffs[3 * a.V + v];
    om[i] = a.state[v]; ic1[i] = u2f(a.state[a.V + v]); ic2[i] = u2f(a.state[2 * a.V + v]);
    dt[i] = a.freq[v];
    istep[i] = (uint32_t)sse_cvt(dt[i] * kStepsPerCycle);
    omdt[i] = 1.0f - dt[i];
    const float r0 = __builtin_amdgcn_rcpf(dt[i]);
    const float e = __builtin_fmaf(-dt[i], r0, 1.0f);
    r1[i] = __builtin_fmaf(e, r0, r0);
    ndt[i] = -dt[i];
  }
  for (size_t r = 0; r < a.T * 16; ++r)
  {
    EXP_TURN(r);
    f32x4 y[N];
#pragma unroll
    for (int k = 0; k < 4; ++k)
    {
      float p[N], num[N], q[N], rem[N], qq[N], clo[N], chi[N], c[N], saw[N], x[N], t0[N], t1[N], t2[N], m1[N], m2[N], o[N];
      bool lo[N], hi[N];
#define EACH for (int i = 0; i < N; ++i)
#pragma unroll
      EACH om[i] += istep[i];
#pragma unroll
      EACH p[i] = (float)(int32_t)(om[i] >> 1);
#pragma unroll
      EACH p[i] = p[i] * 4.656612873077392578125e-10f;
#pragma unroll
      EACH lo[i] = p[i] < dt[i];
#pragma unroll
      EACH hi[i] = p[i] > omdt[i];
#pragma unroll
      EACH num[i] = p[i] - 1.0f;
#pragma unroll
      EACH num[i] = lo[i] ? p[i] : num[i];
#pragma unroll
      EACH q[i] = num[i] * r1[i];
#pragma unroll
      EACH rem[i] = __builtin_fmaf(ndt[i], q[i], num[i]);
#pragma unroll
      EACH q[i] = __builtin_fmaf(rem[i], r1[i], q[i]);
#pragma unroll
      EACH rem[i] = __builtin_fmaf(ndt[i], q[i], num[i]);
#pragma unroll
      EACH q[i] = __builtin_fmaf(rem[i], r1[i], q[i]);
#pragma unroll
      EACH qq[i] = q[i] * q[i];
#pragma unroll
      EACH clo[i] = __builtin_fmaf(2.0f, q[i], -qq[i]);
#pragma unroll
      EACH chi[i] = qq[i] + q[i];
#pragma unroll
      EACH clo[i] = clo[i] - 1.0f;
#pragma unroll
      EACH chi[i] = chi[i] + q[i];
#pragma unroll
      EACH saw[i] = __builtin_fmaf(p[i], 2.f, -1.f);
#pragma unroll
      EACH chi[i] = chi[i] + 1.0f;
#pragma unroll
      EACH c[i] = lo[i] ? clo[i] : chi[i];
#pragma unroll
      EACH c[i] = (lo[i] || hi[i]) ? c[i] : 0.f;
#pragma unroll
      EACH x[i] = saw[i] - c[i];
      // Bandpass
#pragma unroll
      EACH t0[i] = x[i] - ic2[i];
#pragma unroll
      EACH m1[i] = g1[i] * ic1[i];
#pragma unroll
      EACH m2[i] = g0[i] * ic1[i];
#pragma unroll
      EACH t1[i] = g0[i] * t0[i];
#pragma unroll
      EACH t2[i] = g2[i] * t0[i];
#pragma unroll
      EACH t1[i] = t1[i] + m1[i];
#pragma unroll
      EACH t2[i] = t2[i] + m2[i];
#pragma unroll
      EACH o[i] = t1[i] + ic1[i];
#pragma unroll
      EACH ic1[i] = __builtin_fmaf(2.0f, t1[i], ic1[i]);
#pragma unroll
      EACH ic2[i] = __builtin_fmaf(2.0f, t2[i], ic2[i]);
#pragma unroll
      EACH y[i][k] = o[i] * gain[i];
    }
#pragma unroll
    EACH __builtin_nontemporal_store(y[i], a.out + v0 + 64 * i + r * a.V);
  }
#pragma unroll
  for (int i = 0; i < N; ++i)
  {
    const size_t v = v0 + 64 * i;
    a.state[v] = om[i]; a.state[a.V + v] = f2u(ic1[i]); a.state[2 * a.V + v] = f2u(ic2[i]);
  }
}

// variant 3: two voices per lane as PACKED FP32 pairs: every f32 add/mul/fma is one v_pk_* instruction for both voices
// (same issue cost as the scalar instruction on gfx950, tools/valubench.hip); compares, selects, the integer phase and the
// conversion stay per voice.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef int32_t i32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pkfma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
template <int BLK, int PAIRDIST>
__global__ __launch_bounds__(BLK) void k_v3(Args a)
{
  const size_t g = xcd_block(blockIdx.x, gridDim.x) * BLK + threadIdx.x;
  size_t va, vb;
  if (PAIRDIST == 64) { va = (g >> 6) * 128 + (g & 63); vb = va + 64; }
  else { va = (g / BLK) * (2 * BLK) + (g % BLK); vb = va + BLK; }
  if (vb >= a.V) return;
  u32x2 om, istep;
  f32x2 dt, omdt, r1, ndt, g0, g1, g2, gain, ic1, ic2;
  auto ld = [&](const float* p) { return f32x2{p[va], p[vb]}; };
  g0 = ld(a.coeffs); g1 = ld(a.coeffs + a.V); g2 = ld(a.coeffs + 2 * a.V); gain = ld(a.coeffs + 3 * a.V);
  om = u32x2{a.state[va], a.state[vb]};
  ic1 = f32x2{u2f(a.state[a.V + va]), u2f(a.state[a.V + vb])};
  ic2 = f32x2{u2f(a.state[2 * a.V + va]), u2f(a.state[2 * a.V + vb])};
  dt = ld(a.freq);
  istep = u32x2{(uint32_t)sse_cvt(dt.x * kStepsPerCycle), (uint32_t)sse_cvt(dt.y * kStepsPerCycle)};
  omdt = 1.0f - dt;
  {
    const f32x2 r0 = {__builtin_amdgcn_rcpf(dt.x), __builtin_amdgcn_rcpf(dt.y)};
    const f32x2 e = pkfma(-dt, r0, f32x2{1.0f, 1.0f});
    r1 = pkfma(e, r0, r0);
  }
  ndt = -dt;
  const f32x2 one = {1.0f, 1.0f}, two = {2.0f, 2.0f}, zero = {0.f, 0.f};
  f32x4 *pa = a.out + va, *pb = a.out + vb;
  for (size_t r = 0; r < a.T * 16; ++r)
  {
    EXP_TURN(r);
    f32x4 ya, yb;
#pragma unroll
    for (int k = 0; k < 4; ++k)
    {
      om += istep;
      f32x2 p = __builtin_convertvector(__builtin_convertvector(om >> 1, i32x2), f32x2);
      p = p * 4.656612873077392578125e-10f;
      const i32x2 lo = p < dt, hi = p > omdt;
      f32x2 num = p - one;
      num = lo ? p : num;
      f32x2 q = num * r1;
      f32x2 rem = pkfma(ndt, q, num);
      q = pkfma(rem, r1, q);
      rem = pkfma(ndt, q, num);
      q = pkfma(rem, r1, q);
      const f32x2 qq = q * q;
      f32x2 clo = pkfma(two, q, -qq);
      f32x2 chi = qq + q;
      clo = clo - one;
      chi = chi + q;
      const f32x2 saw = pkfma(p, two, -one);
      chi = chi + one;
      f32x2 c = lo ? clo : chi;
      c = (lo | hi) ? c : zero;
      const f32x2 x = saw - c;
      const f32x2 t0 = x - ic2;
      const f32x2 m1 = g1 * ic1, m2 = g0 * ic1;
      f32x2 t1 = g0 * t0, t2 = g2 * t0;
      t1 = t1 + m1;
      t2 = t2 + m2;
      const f32x2 o = t1 + ic1;
      ic1 = pkfma(two, t1, ic1);
      ic2 = pkfma(two, t2, ic2);
      const f32x2 y = o * gain;
      ya[k] = y.x;
      yb[k] = y.y;
    }
    __builtin_nontemporal_store(ya, pa + r * a.V);
    __builtin_nontemporal_store(yb, pb + r * a.V);
  }
  a.state[va] = om.x; a.state[vb] = om.y;
  a.state[a.V + va] = f2u(ic1.x); a.state[a.V + vb] = f2u(ic1.y);
  a.state[2 * a.V + va] = f2u(ic2.x); a.state[2 * a.V + vb] = f2u(ic2.y);
}

// variants 4..7: the product's arithmetic written out for one voice per lane, UN quads per trip, to try instruction-level
// changes one at a time (every one must give variant 0's bits):
//   SCALED  the phase stays the integer-valued float k = float(omega32 >> 1); dt, 1 - dt and the reciprocal are scaled by 2^31
//           once per launch instead (every operand of the polyBLEP scales by an exact power of two, so every rounding is the
//           same), saw = fma(k, 2^-30, -1): one multiply per sample less
//   SADDR   the store address = a wave-uniform 64-bit base (scalar adds) + a 32-bit per-lane offset, so no vector instruction
//           is spent on addresses (needs the signal to be < 4 GiB)
//   VCCSEL  the first select right behind its compare (mask in VCC: a full-rate v_cndmask_b32_e32) via inline asm
template <int BLK, int UN, bool SCALED, bool SADDR, bool VCCSEL, bool NOSTORE = false>
__global__ __launch_bounds__(BLK) void k_v4(Args a)
{
  const size_t v = xcd_block(blockIdx.x, gridDim.x) * BLK + threadIdx.x;
  if (v >= a.V) return;
  const float g0 = a.coeffs[v], g1 = a.coeffs[a.V + v], g2 = a.coeffs[2 * a.V + v], gain = a.coeffs[3 * a.V + v];
  uint32_t om = a.state[v];
  float ic1 = u2f(a.state[a.V + v]), ic2 = u2f(a.state[2 * a.V + v]);
  const float dt = a.freq[v];
  const uint32_t istep = (uint32_t)sse_cvt(dt * kStepsPerCycle);
  const float r0 = __builtin_amdgcn_rcpf(dt);
  const float e = __builtin_fmaf(-dt, r0, 1.0f);
  const float r1u = __builtin_fmaf(e, r0, r0);
  const float K = 2147483648.0f, IK = 4.656612873077392578125e-10f;      // 2^31, 2^-31
  // operands of the per-sample code, scaled or not
  const float dtS = SCALED ? dt * K : dt, omdtS = SCALED ? (1.0f - dt) * K : (1.0f - dt), ndtS = -dtS;
  const float r1S = SCALED ? r1u * IK : r1u;          // num' * r1' = num * r1
  const float oneS = SCALED ? K : 1.0f;
  const float sawMul = SCALED ? 9.31322574615478515625e-10f : 2.0f;     // 2^-30 : 2
  const uint32_t laneOff = (uint32_t)(v * 16);
  const char* base = (const char*)a.out;
  f32x4* po = a.out + v;
  for (size_t r = 0; r < a.T * 16; r += UN)
  {
    EXP_TURN(r);
#pragma unroll
    for (int u = 0; u < UN; ++u)
    {
      f32x4 y;
#pragma unroll
      for (int k = 0; k < 4; ++k)
      {
        om += istep;
        float p = (float)(int32_t)(om >> 1);
        if (!SCALED) p = p * IK;
        const bool hi = p > omdtS;
        const float pm1 = p - oneS;
        float num;
        bool lo;
        if (VCCSEL)
        {
          // v_cmp -> vcc, v_cndmask reads vcc at once (full rate); the mask is copied out for the later selects
          unsigned long long m;
          asm volatile("v_cmp_lt_f32 vcc, %2, %3\n\tv_cndmask_b32 %0, %4, %2, vcc\n\ts_mov_b64 %1, vcc" : "=v"(num), "=s"(m) : "v"(p), "v"(dtS), "v"(pm1) : "vcc");
          lo = (m >> (threadIdx.x & 63)) & 1;
        }
        else
        {
          lo = p < dtS;
          num = lo ? p : pm1;
        }
        float q = num * r1S;
        float rem = __builtin_fmaf(ndtS, q, num);
        q = __builtin_fmaf(rem, r1S, q);
        rem = __builtin_fmaf(ndtS, q, num);
        q = __builtin_fmaf(rem, r1S, q);
        const float qq = q * q;
        const float clo = __builtin_fmaf(2.0f, q, -qq) - 1.0f;
        const float chi = ((qq + q) + q) + 1.0f;
        float c = lo ? clo : chi;
        c = (lo || hi) ? c : 0.f;
        const float saw = __builtin_fmaf(p, sawMul, -1.0f);
        const float x = saw - c;
        const float t0 = x - ic2;
        const float m1 = g1 * ic1, m2 = g0 * ic1;
        const float t1 = g0 * t0 + m1, t2 = g2 * t0 + m2;
        const float o = t1 + ic1;
        ic1 = __builtin_fmaf(2.0f, t1, ic1);
        ic2 = __builtin_fmaf(2.0f, t2, ic2);
        y[k] = o * gain;
      }
      if (SADDR)
      {
        const char* rowBase = base + (r + u) * a.V * 16;      // wave-uniform: scalar arithmetic
        __builtin_nontemporal_store(y, (f32x4*)(rowBase + laneOff));
      }
      else if (!NOSTORE || y.x == 123.456f)  // NOSTORE: the arithmetic alone (the condition never holds)
        __builtin_nontemporal_store(y, po + (r + u) * a.V);
    }
  }
  a.state[v] = om;
  a.state[a.V + v] = f2u(ic1);
  a.state[2 * a.V + v] = f2u(ic2);
}

// variant 8: SPARSE polyBLEP. With a launch-constant frequency 0 < dt <= 1/16 a voice wraps at most once in N = 4 NQ <= 16 samples, so
// at most one sample of a trip lies in the zone after a step (p < dt) and at most one in the zone before (p > 1 - dt) - and those are
// the samples with the smallest / the largest phase of the trip. So: phases and zone masks for the whole trip first (min / max kept
// along), ONE division + polynomial for the lower zone and one for the upper zone per trip instead of one per sample, then
// x_i = saw_i - (lo_i ? cLo : hi_i ? cHi : 0) - the same operands through the same operations for every corrected sample, so the same
// bits. "At most one" fails only on a knife edge: the sample right after a wrap landing within a few units of phase 0 (its successor
// may round below dt too), or the one before a wrap within a few hundred units of 2^32 (its predecessor may round above 1 - dt): both
// show in the trip's min / max (bounds derived in DESIGN 3.1), and such a trip - about one in four thousand - takes the per-sample form.
template <int BLK, int NQ, bool SCALED>
__global__ __launch_bounds__(BLK) void k_v8(Args a)
{
  const size_t v = xcd_block(blockIdx.x, gridDim.x) * BLK + threadIdx.x;
  if (v >= a.V) return;
  constexpr int N = 4 * NQ;
  const float g0 = a.coeffs[v], g1 = a.coeffs[a.V + v], g2 = a.coeffs[2 * a.V + v], gain = a.coeffs[3 * a.V + v];
  uint32_t om = a.state[v];
  float ic1 = u2f(a.state[a.V + v]), ic2 = u2f(a.state[2 * a.V + v]);
  const float dt = a.freq[v];
  const uint32_t istep = (uint32_t)sse_cvt(dt * kStepsPerCycle);
  const float r0 = __builtin_amdgcn_rcpf(dt);
  const float e = __builtin_fmaf(-dt, r0, 1.0f);
  const float r1 = __builtin_fmaf(e, r0, r0);
  const float omdt = 1.0f - dt, ndt = -dt;
  const float IK = 4.656612873077392578125e-10f, K = 2147483648.0f;  // 2^-31, 2^31
  // SCALED: phases stay integer-valued floats k = float(omega32 >> 1) through the compares and min / max (scaling by 2^31 is exact on
  // both sides of every compare); only the two gathered phases and the saw are scaled back
  const float dtC = SCALED ? dt * K : dt, omdtC = SCALED ? omdt * K : omdt;
  const float tinyC = SCALED ? 32.0f : 0x1p-26f, nearOneC = SCALED ? 2147483136.0f : 0.99999976158142089844f;  // 2^-26 ; 1 - 2^-22
  const bool sparseOK = __builtin_amdgcn_ballot_w64(!(dt > 0.f && dt <= 0.0625f)) == 0;
  f32x4* po = a.out + v;
  auto svf = [&](float x) {
    const float t0 = x - ic2;
    const float m1 = g1 * ic1, m2 = g0 * ic1;
    const float t1 = g0 * t0 + m1, t2 = g2 * t0 + m2;
    const float o = t1 + ic1;
    ic1 = __builtin_fmaf(2.0f, t1, ic1);
    ic2 = __builtin_fmaf(2.0f, t2, ic2);
    return o * gain;
  };
  auto divdt = [&](float num) {
    float q = num * r1;
    float rem = __builtin_fmaf(ndt, q, num);
    q = __builtin_fmaf(rem, r1, q);
    rem = __builtin_fmaf(ndt, q, num);
    return __builtin_fmaf(rem, r1, q);
  };
  for (size_t r = 0; r < a.T * 16; r += NQ)
  {
    EXP_TURN(r);
    float p[N];
    bool lo[N], hi[N];
    float tmin = 0.f, tmax = 0.f;
    const uint32_t om0 = om;
#pragma unroll
    for (int i = 0; i < N; ++i)
    {
      om += istep;
      p[i] = (float)(int32_t)(om >> 1);
      if (!SCALED) p[i] *= IK;
      lo[i] = p[i] < dtC;
      hi[i] = p[i] > omdtC;
      tmin = i ? __builtin_fminf(tmin, p[i]) : p[i];
      tmax = i ? __builtin_fmaxf(tmax, p[i]) : p[i];
    }
    const bool suspect = (tmin < tinyC) || (tmax > nearOneC);
    f32x4 y[NQ];
    if (sparseOK && __builtin_amdgcn_ballot_w64(suspect) == 0)
    {
      const float tl = SCALED ? tmin * IK : tmin, th = SCALED ? tmax * IK : tmax;
      const float ql = divdt(tl), qh = divdt(th - 1.0f);
      const float qql = ql * ql, qqh = qh * qh;
      const float cLo = __builtin_fmaf(2.0f, ql, -qql) - 1.0f;
      const float cHi = ((qqh + qh) + qh) + 1.0f;
#pragma unroll
      for (int i = 0; i < N; ++i)
      {
        const float saw = SCALED ? __builtin_fmaf(p[i], 9.31322574615478515625e-10f, -1.0f) : __builtin_fmaf(p[i], 2.0f, -1.0f);
        const float c = lo[i] ? cLo : (hi[i] ? cHi : 0.f);
        y[i >> 2][i & 3] = svf(saw - c);
      }
    }
    else
    {
      om = om0;
#pragma unroll 1
      for (int i = 0; i < N; ++i)
      {
        om += istep;
        const float pp = (float)(int32_t)(om >> 1) * IK;
        const bool l = pp < dt, h = pp > omdt;
        const float num = l ? pp : (pp - 1.0f);
        const float q = num / dt;
        const float qq = q * q;
        const float clo = ((q + q) - qq) - 1.0f;
        const float chi = ((qq + q) + q) + 1.0f;
        float c = l ? clo : chi;
        c = (l || h) ? c : 0.f;
        const float saw = __builtin_fmaf(pp, 2.0f, -1.0f);
        const float o = svf(saw - c);
        // y[i >> 2][i & 3] with a run-time i: four selects instead of scratch
#pragma unroll
        for (int k = 0; k < N; ++k)
          if (k == i) y[k >> 2][k & 3] = o;
      }
    }
#pragma unroll
    for (int u = 0; u < NQ; ++u) __builtin_nontemporal_store(y[u], po + (r + u) * a.V);
  }
  a.state[v] = om;
  a.state[a.V + v] = f2u(ic1);
  a.state[2 * a.V + v] = f2u(ic2);
}

// variant 9: variant 8 software-pipelined by hand: the filter of trip k (a dependent chain, 2.75 independent instructions per step)
// shares a basic block with the phases / compares / min / max of trip k + 1 (all independent), so the scheduler can fill the chain's
// issue gaps the way the dense form's per-sample corrections did.
template <int BLK, int NQ, bool SCALED, bool NOSTORE = false>
__global__ __launch_bounds__(BLK) void k_v9(Args a)
{
  const size_t v = xcd_block(blockIdx.x, gridDim.x) * BLK + threadIdx.x;
  if (v >= a.V) return;
  constexpr int N = 4 * NQ;
  const float g0 = a.coeffs[v], g1 = a.coeffs[a.V + v], g2 = a.coeffs[2 * a.V + v], gain = a.coeffs[3 * a.V + v];
  uint32_t om = a.state[v];
  float ic1 = u2f(a.state[a.V + v]), ic2 = u2f(a.state[2 * a.V + v]);
  const float dt = a.freq[v];
  const uint32_t istep = (uint32_t)sse_cvt(dt * kStepsPerCycle);
  const float r0 = __builtin_amdgcn_rcpf(dt);
  const float e = __builtin_fmaf(-dt, r0, 1.0f);
  const float r1 = __builtin_fmaf(e, r0, r0);
  const float omdt = 1.0f - dt, ndt = -dt;
  const float IK = 4.656612873077392578125e-10f, K = 2147483648.0f;  // 2^-31, 2^31
  const float dtC = SCALED ? dt * K : dt, omdtC = SCALED ? omdt * K : omdt;
  const float tinyC = SCALED ? 32.0f : 0x1p-26f, nearOneC = SCALED ? 2147483136.0f : 0.99999976158142089844f;
  const bool sparseOK = __builtin_amdgcn_ballot_w64(!(dt > 0.f && dt <= 0.0625f)) == 0;
  f32x4* po = a.out + v;
  auto divdt = [&](float num) {
    float q = num * r1;
    float rem = __builtin_fmaf(ndt, q, num);
    q = __builtin_fmaf(rem, r1, q);
    rem = __builtin_fmaf(ndt, q, num);
    return __builtin_fmaf(rem, r1, q);
  };
  float x[N];  // the oscillator's samples of the trip about to be filtered
  float p[N];
  bool lo[N], hi[N];
  uint32_t tmin = 0u, tmax = 0u;  // the extremes as bit patterns: phases are >= 0, so unsigned order = float order (and no canonicalising v_max x, x)
  uint32_t om0 = om;
  // stage A of a trip: phases, zone masks, extremes
  auto stageA = [&]() {
    om0 = om;
#pragma unroll
    for (int i = 0; i < N; ++i)
    {
      om += istep;
      p[i] = (float)(int32_t)(om >> 1);
      if (!SCALED) p[i] *= IK;
      lo[i] = p[i] < dtC;
      hi[i] = p[i] > omdtC;
      tmin = i ? min(tmin, f2u(p[i])) : f2u(p[i]);
      tmax = i ? max(tmax, f2u(p[i])) : f2u(p[i]);
    }
  };
  // stage B: the two corrections of the trip and x[]
  auto stageB = [&]() {
    const float fmin_ = u2f(tmin), fmax_ = u2f(tmax);
    const bool suspect = (fmin_ < tinyC) || (fmax_ > nearOneC);
    if (sparseOK && __builtin_amdgcn_ballot_w64(suspect) == 0)
    {
      const float tl = SCALED ? fmin_ * IK : fmin_, th = SCALED ? fmax_ * IK : fmax_;
      const float ql = divdt(tl), qh = divdt(th - 1.0f);
      const float qql = ql * ql, qqh = qh * qh;
      const float cLo = __builtin_fmaf(2.0f, ql, -qql) - 1.0f;
      const float cHi = ((qqh + qh) + qh) + 1.0f;
#pragma unroll
      for (int i = 0; i < N; ++i)
      {
        const float saw = SCALED ? __builtin_fmaf(p[i], 9.31322574615478515625e-10f, -1.0f) : __builtin_fmaf(p[i], 2.0f, -1.0f);
        const float c = lo[i] ? cLo : (hi[i] ? cHi : 0.f);
        x[i] = saw - c;
      }
    }
    else
    {
      uint32_t o2 = om0;
#pragma unroll 1
      for (int i = 0; i < N; ++i)
      {
        o2 += istep;
        const float pp = (float)(int32_t)(o2 >> 1) * IK;
        const bool l = pp < dt, h = pp > omdt;
        const float num = l ? pp : (pp - 1.0f);
        const float q = num / dt;
        const float qq = q * q;
        const float clo = ((q + q) - qq) - 1.0f;
        const float chi = ((qq + q) + q) + 1.0f;
        float c = l ? clo : chi;
        c = (l || h) ? c : 0.f;
        const float xi = __builtin_fmaf(pp, 2.0f, -1.0f) - c;
#pragma unroll
        for (int k = 0; k < N; ++k)
          if (k == i) x[k] = xi;
      }
    }
  };
  stageA();
  stageB();
  const size_t R = a.T * 16;
  for (size_t r = 0; r < R; r += NQ)
  {
    EXP_TURN(r);
    // the filter of this trip next to stage A of the next one (the last trip's extra stage A is undone below)
    f32x4 y[NQ];
    float xs[N];
#pragma unroll
    for (int i = 0; i < N; ++i) xs[i] = x[i];
    om0 = om;
#pragma unroll
    for (int i = 0; i < N; ++i)
    {
      // stage A of the next trip, sample by sample between the filter's steps
      om += istep;
      p[i] = (float)(int32_t)(om >> 1);
      if (!SCALED) p[i] *= IK;
      lo[i] = p[i] < dtC;
      hi[i] = p[i] > omdtC;
      tmin = i ? min(tmin, f2u(p[i])) : f2u(p[i]);
      tmax = i ? max(tmax, f2u(p[i])) : f2u(p[i]);
      const float t0 = xs[i] - ic2;
      const float m1 = g1 * ic1, m2 = g0 * ic1;
      const float t1 = g0 * t0 + m1, t2 = g2 * t0 + m2;
      const float o = t1 + ic1;
      ic1 = __builtin_fmaf(2.0f, t1, ic1);
      ic2 = __builtin_fmaf(2.0f, t2, ic2);
      y[i >> 2][i & 3] = o * gain;
      if ((i & 3) == 3 && (!NOSTORE || y[i >> 2].x == 123.456f)) __builtin_nontemporal_store(y[i >> 2], po + (r + (i >> 2)) * a.V);
      // keep this sample's filter step and its share of stage A together: the empty statement wants both sides' values
      asm volatile("" : "+v"(tmin), "+v"(tmax), "+v"(ic1), "+v"(ic2));
      __builtin_amdgcn_sched_barrier(0);
    }
    stageB();  // (unconditional, or the compiler sinks stage A into the condition and away from the filter)
  }
  om -= (uint32_t)N * istep;  // the loop ran one trip of phases ahead
  a.state[v] = om;
  a.state[a.V + v] = f2u(ic1);
  a.state[2 * a.V + v] = f2u(ic2);
}

template <class F>
float timeit(F f, int reps = 8)
{
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps;
}

#include <algorithm>
#include <functional>
#include <string>
int main(int argc, char** argv)
{
  const size_t V = 262144, T = 32, n = V * T * 64;
  const int rounds = argc > 1 ? atoi(argv[1]) : 7;
  std::vector<float> co(4 * V), fr(V);
  for (size_t v = 0; v < V; ++v)
  {
    fr[v] = (float)(55.0 * pow(2.0, 5.0 * v / (double)V) / 48000.0);
    const float omega = fminf(0.45f, 4.f * fr[v]), k = 0.5f;
    const float piOmega = 3.14159265f * omega, s1 = sinf(piOmega), s2 = sinf(2.f * piOmega), nrm = 1.f / (2.f + k * s2);
    co[v] = s2 * nrm; co[V + v] = (-2.f * s1 * s1 - k * s2) * nrm; co[2 * V + v] = (2.f * s1 * s1) * nrm; co[3 * V + v] = 0.25f;
  }
  float *dco, *dfr; uint32_t* dst; f32x4 *out0, *out1;
  CK(hipMalloc(&dco, 16 * V)); CK(hipMalloc(&dfr, 4 * V)); CK(hipMalloc(&dst, 12 * V));
  CK(hipMalloc(&out0, 4 * n)); CK(hipMalloc(&out1, 4 * n));
  CK(hipMemcpy(dco, co.data(), 16 * V, hipMemcpyHostToDevice)); CK(hipMemcpy(dfr, fr.data(), 4 * V, hipMemcpyHostToDevice));
  std::vector<uint32_t> ref(n), got(n);
  struct Var { std::string name; std::function<void(Args)> launch; std::vector<float> ms; size_t bad; };
  std::vector<Var> vars;
  auto add = [&](const char* name, std::function<void(Args)> f) { vars.push_back({name, f, {}, 0}); };
  add("v0  1v/lane blk256", [&](Args a) { hipLaunchKernelGGL(k_v0<256>, dim3(V / 256), dim3(256), 0, 0, a); });
  add("v0  1v/lane blk64", [&](Args a) { hipLaunchKernelGGL(k_v0<64>, dim3(V / 64), dim3(64), 0, 0, a); });
  add("v0b shape unroll4 nt blk256", [&](Args a) { hipLaunchKernelGGL((k_v0b<256, 4, true>), dim3(V / 256), dim3(256), 0, 0, a, 16 * V, V, (size_t)1); });
  add("v0b shape unroll1 nt blk256", [&](Args a) { hipLaunchKernelGGL((k_v0b<256, 1, true>), dim3(V / 256), dim3(256), 0, 0, a, 16 * V, V, (size_t)1); });
  add("v0b shape unroll1 plain blk256", [&](Args a) { hipLaunchKernelGGL((k_v0b<256, 1, false>), dim3(V / 256), dim3(256), 0, 0, a, 16 * V, V, (size_t)1); });
  add("v0b shape unroll1 nt blk64", [&](Args a) { hipLaunchKernelGGL((k_v0b<64, 1, true>), dim3(V / 64), dim3(64), 0, 0, a, 16 * V, V, (size_t)1); });
  add("v0b shape unroll4 nt blk64", [&](Args a) { hipLaunchKernelGGL((k_v0b<64, 4, true>), dim3(V / 64), dim3(64), 0, 0, a, 16 * V, V, (size_t)1); });
  add("v1  2v/lane blk64", [&](Args a) { hipLaunchKernelGGL(k_v1<64>, dim3(V / 128), dim3(64), 0, 0, a); });
  add("v1  2v/lane blk128", [&](Args a) { hipLaunchKernelGGL(k_v1<128>, dim3(V / 256), dim3(128), 0, 0, a); });
  add("v1  2v/lane blk256", [&](Args a) { hipLaunchKernelGGL(k_v1<256>, dim3(V / 512), dim3(256), 0, 0, a); });
  add("v2  N=2 explicit blk64", [&](Args a) { hipLaunchKernelGGL((k_v2<64, 2>), dim3(V / 128), dim3(64), 0, 0, a); });
  add("v3  2v/lane packed blk64 d64", [&](Args a) { hipLaunchKernelGGL((k_v3<64, 64>), dim3(V / 128), dim3(64), 0, 0, a); });
  add("v3  2v/lane packed blk128 d64", [&](Args a) { hipLaunchKernelGGL((k_v3<128, 64>), dim3(V / 256), dim3(128), 0, 0, a); });
  add("v3  2v/lane packed blk256 d64", [&](Args a) { hipLaunchKernelGGL((k_v3<256, 64>), dim3(V / 512), dim3(256), 0, 0, a); });
  add("v3  2v/lane packed blk256 d256", [&](Args a) { hipLaunchKernelGGL((k_v3<256, 256>), dim3(V / 512), dim3(256), 0, 0, a); });
  add("v3  2v/lane packed blk128 d128", [&](Args a) { hipLaunchKernelGGL((k_v3<128, 128>), dim3(V / 256), dim3(128), 0, 0, a); });
  add("v4  written out, 4 quads/trip", [&](Args a) { hipLaunchKernelGGL((k_v4<256, 4, false, false, false>), dim3(V / 256), dim3(256), 0, 0, a); });
  add("v4  written out, 2 quads/trip", [&](Args a) { hipLaunchKernelGGL((k_v4<256, 2, false, false, false>), dim3(V / 256), dim3(256), 0, 0, a); });
  add("v5  + scaled phase", [&](Args a) { hipLaunchKernelGGL((k_v4<256, 4, true, false, false>), dim3(V / 256), dim3(256), 0, 0, a); });
  add("v6  + scalar store base", [&](Args a) { hipLaunchKernelGGL((k_v4<256, 4, true, true, false>), dim3(V / 256), dim3(256), 0, 0, a); });
  add("v6b scalar store base only", [&](Args a) { hipLaunchKernelGGL((k_v4<256, 4, false, true, false>), dim3(V / 256), dim3(256), 0, 0, a); });
  add("v8  sparse blep, 4 quads/trip", [&](Args a) { hipLaunchKernelGGL((k_v8<256, 4, false>), dim3(V / 256), dim3(256), 0, 0, a); });
  add("v8  sparse blep, 2 quads/trip", [&](Args a) { hipLaunchKernelGGL((k_v8<256, 2, false>), dim3(V / 256), dim3(256), 0, 0, a); });
  add("v8s sparse blep scaled, 4 quads/trip", [&](Args a) { hipLaunchKernelGGL((k_v8<256, 4, true>), dim3(V / 256), dim3(256), 0, 0, a); });
  add("v8s sparse blep scaled, 2 quads/trip", [&](Args a) { hipLaunchKernelGGL((k_v8<256, 2, true>), dim3(V / 256), dim3(256), 0, 0, a); });
  add("v9  sparse pipelined, 4 quads/trip", [&](Args a) { hipLaunchKernelGGL((k_v9<256, 4, false>), dim3(V / 256), dim3(256), 0, 0, a); });
  add("v9  sparse pipelined, 2 quads/trip", [&](Args a) { hipLaunchKernelGGL((k_v9<256, 2, false>), dim3(V / 256), dim3(256), 0, 0, a); });
  add("v9s sparse pipelined scaled, 4 quads/trip", [&](Args a) { hipLaunchKernelGGL((k_v9<256, 4, true>), dim3(V / 256), dim3(256), 0, 0, a); });
  add("v9s sparse pipelined scaled, 2 quads/trip", [&](Args a) { hipLaunchKernelGGL((k_v9<256, 2, true>), dim3(V / 256), dim3(256), 0, 0, a); });
  add("v4  arithmetic only (no stores)", [&](Args a) { hipLaunchKernelGGL((k_v4<256, 4, false, false, false, true>), dim3(V / 256), dim3(256), 0, 0, a); });
  add("v9  arithmetic only (no stores)", [&](Args a) { hipLaunchKernelGGL((k_v9<256, 4, false, true>), dim3(V / 256), dim3(256), 0, 0, a); });
  add("v7  + vcc select", [&](Args a) { hipLaunchKernelGGL((k_v4<256, 4, true, true, true>), dim3(V / 256), dim3(256), 0, 0, a); });
  // correctness of every variant against variant 0, from cleared state
  // ... and from a state with free-running phases, a quarter of the voices placed so that a sample lands within a few units of a
  // wrap (just after: phase 0 .. 23; just before: 2^32 - 16 * (0 .. 23)) some 1 .. 37 samples into the launch
  std::vector<uint32_t> st0(3 * V, 0u), st1(3 * V);
  {
    uint32_t x = 12345u;
    for (size_t v = 0; v < V; ++v)
    {
      x = x * 1664525u + 1013904223u;
      const uint32_t istep = (uint32_t)lrintf(fr[v] * 4294967296.0f), k = (uint32_t)(v % 37) + 1, j = (uint32_t)((v / 4) % 24);
      uint32_t om = x;
      if (v % 8 == 1) om = 0u - k * istep + j;
      if (v % 8 == 5) om = 0u - k * istep - 16u * j;
      st1[v] = om;
      x = x * 1664525u + 1013904223u;
      const float i1 = ((int32_t)x) * (0.3f / 2147483648.f);
      x = x * 1664525u + 1013904223u;
      const float i2 = ((int32_t)x) * (0.3f / 2147483648.f);
      memcpy(&st1[V + v], &i1, 4);
      memcpy(&st1[2 * V + v], &i2, 4);
    }
  }
  std::vector<uint32_t> refState(3 * V), gotState(3 * V);
  for (int pass = 0; pass < 2; ++pass)
  {
    for (size_t i = 0; i < vars.size(); ++i)
    {
      Args a{dco, dst, dfr, i == 0 ? out0 : out1, V, T};
      CK(hipMemcpy(dst, (pass ? st1 : st0).data(), 12 * V, hipMemcpyHostToDevice));
      vars[i].launch(a);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(got.data(), a.out, 4 * n, hipMemcpyDeviceToHost));
      CK(hipMemcpy(gotState.data(), dst, 12 * V, hipMemcpyDeviceToHost));
      if (i == 0) { ref = got; refState = gotState; }
      else
      {
        for (size_t j = 0; j < n; ++j) vars[i].bad += (got[j] != ref[j]);
        for (size_t j = 0; j < 3 * V; ++j) vars[i].bad += (gotState[j] != refState[j]);
      }
    }
  }
  CK(hipMemcpy(dst, st1.data(), 12 * V, hipMemcpyHostToDevice));  // the timed launches start from free-running phases
  // round 6: `exp_chain <rounds> steady` - every variant ALONE for 600 back-to-back launches after 100 untimed ones: the board is at its
  // power cap after a few milliseconds of this load (profiles/r06_cfg3_spread.md), and a variant timed for 10 launches between
  // other variants runs on the clock the previous one left
  if (argc > 2 && strcmp(argv[2], "steady") == 0)
  {
    for (auto& v : vars)
    {
      if (argc > 3 && v.name.find(argv[3]) == std::string::npos) continue;
      int k = 0;
      Args a0{dco, dst, dfr, out0, V, T}, a1{dco, dst, dfr, out1, V, T};
      for (int i = 0; i < 100; ++i) v.launch((k++ & 1) ? a1 : a0);
      CK(hipDeviceSynchronize());
      const float ms = timeit([&] { v.launch((k++ & 1) ? a1 : a0); }, 600);
      printf("steady %-44s %.4f ms (%.0f GB/s)  mismatches %zu\n", v.name.c_str(), ms, 4.0 * n / ms / 1e6, v.bad);
    }
    return 0;
  }
  // interleaved timing: rounds x (each variant: 10 launches alternating two output buffers)
  for (int r = 0; r < rounds; ++r)
    for (auto& v : vars)
    {
      int k = 0;
      Args a0{dco, dst, dfr, out0, V, T}, a1{dco, dst, dfr, out1, V, T};
      v.ms.push_back(timeit([&] { v.launch((k++ & 1) ? a1 : a0); }, 10));
    }
  for (auto& v : vars)
  {
    std::sort(v.ms.begin(), v.ms.end());
    const float mn = v.ms.front(), md = v.ms[v.ms.size() / 2];
    printf("%-32s min %.4f ms (%.0f GB/s)  median %.4f ms (%.0f GB/s)  mismatches %zu\n", v.name.c_str(), mn,
           4.0 * n / mn / 1e6, md, 4.0 * n / md / 1e6, v.bad);
  }
  return 0;
}
