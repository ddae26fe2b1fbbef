// mldsp_kernels.hpp — the voice-bank device kernels (templates), shared by the ahead-of-time
// instantiations in chains.hip and by kernels generated at run time (graph.hip, hiprtc).
//
// chain_kernel<Chain<K0,K1,...>, HAS_SIGNAL>: one wavefront lane per voice, DSPVectors walked
// serially: the lane loads its voice's coefficients and state (SoA, coalesced 4 B/lane), keeps
// them in VGPRs for the whole launch, produces 4 samples at a time and moves them with one 16-byte
// access per lane (1 KiB per wavefront per instruction in the QUAD layout), then writes the state
// back. Intermediate signals between processors never touch memory.
// HBM traffic per voice-sample = 4 B out (+4 B in when a signal is streamed)
// + (4*(NC+NS) read + 4*NS written)/(64*T) — DESIGN.md §3.
//
// No LDS except the 17-tap ImpulseGen table (a genuinely shared coefficient table, staged once per
// workgroup); no MFMA: the path is elementwise/recurrent, not a contraction.
// Compile with -ffp-contract=off -fno-slp-vectorize (see mldsp_math.hpp).
#pragma once
#include "mlgpu_device_args.hpp"
#include "mldsp_procs.hpp"

namespace mldev
{
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kChainBlock = 256;

// the streaming loop of one voice: T DSPVectors, 16 quads each, one 16-byte access per quad
template <class CH, bool HAS_SIGNAL, bool FAST_HEAD>
__device__ __forceinline__ void run_voice(CH& ch, const ChainArgs& a, size_t v, float xc)
{
  const f32x4* pin = HAS_SIGNAL ? (const f32x4*)a.in.base + v * a.in.strideV : nullptr;
  f32x4* pout = (f32x4*)a.out.base + v * a.out.strideV;
  const size_t inQ = a.in.strideQ, outQ = a.out.strideQ;
  for (size_t t = 0; t < a.T; ++t)
  {
    const f32x4* pi = HAS_SIGNAL ? pin + t * a.in.strideT : nullptr;
    f32x4* po = pout + t * a.out.strideT;
#pragma unroll 4
    for (int q = 0; q < 16; ++q)
    {
      f32x4 x = {xc, xc, xc, xc};
      if constexpr (HAS_SIGNAL) x = __builtin_nontemporal_load(pi + q * inQ);
      f32x4 y;
      y.x = ch.template next_head<FAST_HEAD>(x.x);
      y.y = ch.template next_head<FAST_HEAD>(x.y);
      y.z = ch.template next_head<FAST_HEAD>(x.z);
      y.w = ch.template next_head<FAST_HEAD>(x.w);
      __builtin_nontemporal_store(y, po + q * outQ);
    }
    ch.end_vector();
  }
}

template <class CH, bool HAS_SIGNAL>
__device__ __forceinline__ void chain_kernel_body(const ChainArgs& a)
{
  __shared__ float ldsTable[CH::kHasImpulse ? 32 : 1];
  if constexpr (CH::kHasImpulse)
  {
    if (threadIdx.x < Proc<MLGPU_PROC_IMPULSE_GEN>::kTableSize) ldsTable[threadIdx.x] = a.impulseTable[threadIdx.x];
    __syncthreads();
  }
  // XCD-aware workgroup -> voice mapping. Workgroup b is dispatched to XCD b % 8 (observed on
  // MI355X; used for speed only, any bijection is correct). Give XCD x the x-th contiguous eighth
  // of the voices, so each XCD's L2 writes back one contiguous segment of every signal row instead
  // of every 8th KiB: measured +37 % on the bare store pattern (tools/membench2.hip).
  size_t blk = blockIdx.x;
  const size_t nbFull = (size_t)gridDim.x & ~(size_t)7;
  if (blk < nbFull) blk = (blk & 7) * (nbFull >> 3) + (blk >> 3);
  const size_t v = blk * kChainBlock + threadIdx.x;
  if (v >= a.V) return;

  CH ch;
  const VoiceMem mem{a.coeffs + v, a.state + v, a.V};
  const KernelTables tables{ldsTable};
  ch.load(mem, tables);

  const float xc = (!HAS_SIGNAL && a.inConst) ? a.inConst[v] : 0.f;

  // A launch-constant input lets the head processor (SawGen / PulseGen) skip its per-sample
  // range test: decide once per wavefront which loop body to run.
  bool fastHead = false;
  if constexpr (!HAS_SIGNAL && CH::kHeadHasFastPath) fastHead = (__builtin_amdgcn_ballot_w64(CH::head_input_is_odd(xc)) == 0);
  if (fastHead)
    run_voice<CH, HAS_SIGNAL, true>(ch, a, v, xc);
  else
    run_voice<CH, HAS_SIGNAL, false>(ch, a, v, xc);
  ch.store(mem);
}

template <class CH, bool HAS_SIGNAL>
__global__ __launch_bounds__(kChainBlock) void chain_kernel(const ChainArgs a)
{
  chain_kernel_body<CH, HAS_SIGNAL>(a);
}

// ---------------------------------------------------------------------------------------------
// cascade_kernel — N identical SVF sections in series (config 4), evaluated STAGE-SKEWED.
//
// A cascade is a long dependent chain per sample (N x ~5 dependent VALU ops), and on gfx950 a
// wave's back-to-back dependent VALU instructions issue at half rate no matter how many other
// waves are resident (tools/valubench.hip: ILP 1 -> 31 T lane-instr/s, ILP >= 4 -> 64 T). So the
// cascade is software-pipelined across stages: at tick i, stage s works on sample i - s, reading
// the value stage s-1 produced at tick i-1. The N stage updates of one tick are mutually
// independent and are written op-by-op ACROSS stages, so consecutive instructions never depend on
// each other (ILP = N). Same arithmetic per stage as SvfCore (mldsp_procs.hpp) => same bits.
//
// The pipeline is filled and drained inside every launch (prologue: stages 0..i active; epilogue:
// stages i-S+1..N-1 active), so the state written back is exactly the state after S samples and a
// launch boundary is invisible (tests: split launches == one launch). Boundary ticks (2N-2 of
// 64*T) use a slow masked form; the steady loop handles 4 output quads per trip and keeps the next
// trip's four 16-byte input loads in flight for a whole trip (~1300 VALU instructions).
template <int KIND, int N>
struct SvfCascade
{
  static constexpr int NCK = (KIND == MLGPU_PROC_HIPASS) ? 4 : 3;
  static constexpr int NC = NCK * N, NS = 2 * N;
  float g0[N], g1[N], g2[N], kk[N], ic1[N], ic2[N], r[N];

  MLD void load(const VoiceMem& m)
  {
#pragma unroll
    for (int s = 0; s < N; ++s)
    {
      g0[s] = m.c(NCK * s);
      g1[s] = m.c(NCK * s + 1);
      g2[s] = m.c(NCK * s + 2);
      kk[s] = (KIND == MLGPU_PROC_HIPASS) ? m.c(NCK * s + 3) : 0.f;
      ic1[s] = u2f(m.s(2 * s));
      ic2[s] = u2f(m.s(2 * s + 1));
      r[s] = 0.f;
    }
  }
  MLD void store(const VoiceMem& m) const
  {
#pragma unroll
    for (int s = 0; s < N; ++s)
    {
      m.set(2 * s, f2u(ic1[s]));
      m.set(2 * s + 1, f2u(ic2[s]));
    }
  }
  // all stages active; returns stage N-1's output, i.e. the chain output for sample (tick - (N-1))
  MLD float tick(float x)
  {
    float in[N], t0[N], a[N], b[N], c[N], d[N], t1[N], t2[N], v1[N], v2[N];
    in[0] = x;
#pragma unroll
    for (int s = 1; s < N; ++s) in[s] = r[s - 1];
#pragma unroll
    for (int s = 0; s < N; ++s) t0[s] = in[s] - ic2[s];
#pragma unroll
    for (int s = 0; s < N; ++s) b[s] = g1[s] * ic1[s];
#pragma unroll
    for (int s = 0; s < N; ++s) d[s] = g0[s] * ic1[s];
#pragma unroll
    for (int s = 0; s < N; ++s) a[s] = g0[s] * t0[s];
#pragma unroll
    for (int s = 0; s < N; ++s) c[s] = g2[s] * t0[s];
#pragma unroll
    for (int s = 0; s < N; ++s) t1[s] = a[s] + b[s];
#pragma unroll
    for (int s = 0; s < N; ++s) t2[s] = c[s] + d[s];
    if (KIND == MLGPU_PROC_LOPASS)
    {
#pragma unroll
      for (int s = 0; s < N; ++s) r[s] = t2[s] + ic2[s];
    }
    else if (KIND == MLGPU_PROC_BANDPASS)
    {
#pragma unroll
      for (int s = 0; s < N; ++s) r[s] = t1[s] + ic1[s];
    }
    else
    {
#pragma unroll
      for (int s = 0; s < N; ++s) v1[s] = t1[s] + ic1[s];
#pragma unroll
      for (int s = 0; s < N; ++s) v2[s] = t2[s] + ic2[s];
#pragma unroll
      for (int s = 0; s < N; ++s) v1[s] = kk[s] * v1[s];
#pragma unroll
      for (int s = 0; s < N; ++s) v1[s] = in[s] - v1[s];
#pragma unroll
      for (int s = 0; s < N; ++s) r[s] = v1[s] - v2[s];
    }
#pragma unroll
    for (int s = 0; s < N; ++s) ic1[s] = __builtin_fmaf(2.0f, t1[s], ic1[s]);
#pragma unroll
    for (int s = 0; s < N; ++s) ic2[s] = __builtin_fmaf(2.0f, t2[s], ic2[s]);
    return r[N - 1];
  }
  // boundary tick: only stages sLo..sHi (wave-uniform) are active
  MLD float tick_masked(float x, int sLo, int sHi)
  {
    float in[N];
    in[0] = x;
#pragma unroll
    for (int s = 1; s < N; ++s) in[s] = r[s - 1];
#pragma unroll
    for (int s = 0; s < N; ++s)
    {
      if (s >= sLo && s <= sHi)
      {
        const float t0 = in[s] - ic2[s];
        const float t1 = g0[s] * t0 + g1[s] * ic1[s];
        const float t2 = g2[s] * t0 + g0[s] * ic1[s];
        if (KIND == MLGPU_PROC_LOPASS)
          r[s] = t2 + ic2[s];
        else if (KIND == MLGPU_PROC_BANDPASS)
          r[s] = t1 + ic1[s];
        else
          r[s] = in[s] - kk[s] * (t1 + ic1[s]) - (t2 + ic2[s]);
        ic1[s] = __builtin_fmaf(2.0f, t1, ic1[s]);
        ic2[s] = __builtin_fmaf(2.0f, t2, ic2[s]);
      }
    }
    return r[N - 1];
  }
};

// HEAD = Chain<...> of processors in front of the cascade (e.g. NoiseGen), evaluated unskewed.
template <class HEAD, int KIND, int N, bool HAS_SIGNAL>
__global__ __launch_bounds__(kChainBlock) void cascade_kernel(const ChainArgs a)
{
  static_assert(!HEAD::kHasImpulse, "ImpulseGen heads are not supported by the cascade kernel");
  size_t blk = blockIdx.x;
  const size_t nbFull = (size_t)gridDim.x & ~(size_t)7;
  if (blk < nbFull) blk = (blk & 7) * (nbFull >> 3) + (blk >> 3);  // XCD-aware, see chain_kernel
  const size_t v = blk * kChainBlock + threadIdx.x;
  if (v >= a.V) return;

  HEAD head;
  SvfCascade<KIND, N> c;
  const VoiceMem mh{a.coeffs + v, a.state + v, a.V};
  const VoiceMem mc{a.coeffs + (size_t)HEAD::NC * a.V + v, a.state + (size_t)HEAD::NS * a.V + v, a.V};
  const KernelTables tables{nullptr};
  head.load(mh, tables);
  c.load(mc);
  const float xc = (!HAS_SIGNAL && a.inConst) ? a.inConst[v] : 0.f;

  constexpr int D = N - 1;             // output lag in ticks
  constexpr int A = D / 4, B = D % 4;  // D = 4A + B
  const size_t S = a.T * 64;
  const f32x4* pin = HAS_SIGNAL ? (const f32x4*)a.in.base + v * a.in.strideV : nullptr;
  f32x4* pout = (f32x4*)a.out.base + v * a.out.strideV;
  // quad index qi = 16 t + q  ->  element offset
  auto inQuad = [&](size_t qi) { return pin + (qi >> 4) * a.in.strideT + (qi & 15) * a.in.strideQ; };
  auto outQuad = [&](size_t qi) { return pout + (qi >> 4) * a.out.strideT + (qi & 15) * a.out.strideQ; };
  auto inAt = [&](size_t i) -> float {
    if constexpr (HAS_SIGNAL) return ((const float*)inQuad(i >> 2))[i & 3];
    return xc;
  };

  // prologue: ticks 0..D-1, stages 0..i active
  for (int i = 0; i < D; ++i) c.tick_masked(head.next(inAt((size_t)i)), 0, i);

  // steady state: output quad q <- ticks D+4q .. D+4q+3, inputs x[4(q+A)+B+j]
  const size_t Q = (S - D) / 4;
  size_t q = 0;
  if constexpr (HAS_SIGNAL)
  {
    f32x4 w[5];  // input quads q+A .. q+A+4 of the current trip (4 output quads)
    if (Q >= 4)
    {
#pragma unroll
      for (int k = 0; k < 5; ++k) w[k] = __builtin_nontemporal_load(inQuad((size_t)(A + k)));
    }
    for (; q + 4 <= Q; q += 4)
    {
      const bool more = (q + 8 <= Q);
      f32x4 nx[4];
      if (more)
      {
#pragma unroll
        for (int k = 0; k < 4; ++k) nx[k] = __builtin_nontemporal_load(inQuad(q + A + 5 + k));
      }
#pragma unroll
      for (int g = 0; g < 4; ++g)
      {
        f32x4 y;
#pragma unroll
        for (int j = 0; j < 4; ++j)
        {
          const int e = B + j;
          const float x = (e < 4) ? w[g][e & 3] : w[g + 1][e & 3];
          y[j] = c.tick(head.next(x));
        }
        __builtin_nontemporal_store(y, outQuad(q + g));
      }
      if (more)
      {
        w[0] = w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k + 1] = nx[k];
      }
    }
  }
  else
  {
    for (; q + 4 <= Q; q += 4)
    {
#pragma unroll
      for (int g = 0; g < 4; ++g)
      {
        f32x4 y;
#pragma unroll
        for (int j = 0; j < 4; ++j) y[j] = c.tick(head.next(xc));
        __builtin_nontemporal_store(y, outQuad(q + g));
      }
    }
  }
  // tail: ticks D+4q .. S+D-1; inputs exist while i < S, after that the pipeline drains
  for (size_t i = D + 4 * q; i < S + D; ++i)
  {
    float x = 0.f;
    if (i < S) x = head.next(inAt(i));
    const int sLo = (i < S) ? 0 : (int)(i - S + 1);
    const float y = c.tick_masked(x, sLo, N - 1);
    const size_t n = i - D;
    ((float*)outQuad(n >> 2))[n & 3] = y;
  }
  head.store(mh);
  c.store(mc);
}

}  // namespace mldev
