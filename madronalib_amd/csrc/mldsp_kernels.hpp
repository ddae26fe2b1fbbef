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
// the wavefronts of a SIMD take turns at the priority levels (mldsp_math.hpp: take_turns_by_clock) - 0 for A / B builds
#ifndef MLGPU_CHAIN_TURNS
#define MLGPU_CHAIN_TURNS 1
#endif
#ifndef MLGPU_CHAIN_TURN_SHIFT
#define MLGPU_CHAIN_TURN_SHIFT 13
#endif
constexpr int kTurnClockShift = MLGPU_CHAIN_TURN_SHIFT;  // 82 us per turn (graph kernels: the best of 2^7 .. 2^18 ticks, profiles/r04_take_turns.txt)

// ---- the voices of a wavefront summed inside the voice kernel (chain_mix_kernel) ----
// mlgpu_mixdown's first stage - the balanced tree over 64 consecutive voices, a[i] += a[i + d] for d = 1, 2 ... 32 - without the
// voices' signals ever reaching memory: 16 samples (four quads) of the 64 lanes are parked in an LDS strip of the wavefront, lane
// (s = lane & 15, g = lane >> 4) adds up sample s of voices 16 g .. 16 g + 15 in tree order, and two lane exchanges add the four
// quarters as the tree does. Row of voice L at 20 L + 16 (L >> 4) floats: 16-byte aligned for the b128 writes, and the 64 column
// reads of a step fall on 64 different banks.
constexpr int kMixStrip = 64 * 20 + 3 * 16 + 16;  // floats per wavefront
template <int LO, int N>
struct MixTree16
{
  static __device__ __forceinline__ float sum(const float* col) { return MixTree16<LO, N / 2>::sum(col) + MixTree16<LO + N / 2, N / 2>::sum(col); }
};
template <int LO>
struct MixTree16<LO, 1>
{
  static __device__ __forceinline__ float sum(const float* col) { return col[LO * 20]; }
};

// a quad (four samples) of every lane into the strip, quad qq of the 16 samples a step adds up
__device__ __forceinline__ void mix64_park(float* strip, int qq, f32x4 y)
{
  const uint32_t lane = threadIdx.x & 63u;
  *(f32x4*)(strip + lane * 20 + (lane >> 4) * 16 + 4 * qq) = y;
}
// ... and after the fourth quad: the 16 sums of the wavefront's 64 voices, stored by lanes 0 .. 15 to row16[0 .. 15]
__device__ __forceinline__ void mix64_sum_store(const float* strip, float* row16)
{
  const uint32_t lane = threadIdx.x & 63u;
  // (the strip is this wavefront's own: its lanes run in lockstep, the LDS operations of a wavefront complete in order)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const uint32_t s = lane & 15u, g = lane >> 4;
  const float t16 = MixTree16<0, 16>::sum(strip + g * (16 * 20 + 16) + s);
  const float t32 = t16 + u2f((uint32_t)__builtin_amdgcn_ds_bpermute((int)(((lane + 16u) & 63u) * 4u), (int)f2u(t16)));
  const float t64 = t32 + u2f((uint32_t)__builtin_amdgcn_ds_bpermute((int)(((lane + 32u) & 63u) * 4u), (int)f2u(t32)));
  if (lane < 16) row16[s] = t64;
  __builtin_amdgcn_wave_barrier();
}

template <bool MIX>
struct MixStrips
{
  static __device__ __forceinline__ float* mine() { return nullptr; }
};
template <>
struct MixStrips<true>
{
  static __device__ __forceinline__ float* mine()
  {
    __shared__ __attribute__((aligned(16))) float lds[(256 / 64) * kMixStrip];
    return lds + (threadIdx.x >> 6) * kMixStrip;
  }
};

// the streaming loop of one voice: T DSPVectors, 16 quads each, one 16-byte access per quad
// (SCALED: the summing form with per-voice gains and / or spare lanes in the wavefront - a loop of its own, so that the plain one
// pays nothing for them)
template <class CH, bool HAS_SIGNAL, bool FAST_HEAD, bool MIX = false, bool SCALED = false>
__device__ __forceinline__ void run_voice(CH& ch, const ChainArgs& a, size_t v, float xc, float* strip = nullptr, bool live = true)
{
  const float mixGain = (SCALED && a.mixGains) ? a.mixGains[v] : 1.f;
  const f32x4* pin = HAS_SIGNAL ? (const f32x4*)a.in.base + v * a.in.strideV : nullptr;
  f32x4* pout = MIX ? nullptr : (f32x4*)a.out.base + v * a.out.strideV;
  const size_t inQ = a.in.strideQ, outQ = a.out.strideQ;
#if MLGPU_CHAIN_TURNS
  const uint32_t slot = wave_slot();
#endif
  for (size_t t = 0; t < a.T; ++t)
  {
    const f32x4* pi = HAS_SIGNAL ? pin + t * a.in.strideT : nullptr;
    f32x4* po = MIX ? nullptr : pout + t * a.out.strideT;
#pragma unroll 4
    for (int q = 0; q < 16; ++q)
    {
#if MLGPU_CHAIN_TURNS
      if ((q & 3) == 0) take_turns_by_clock(slot, kTurnClockShift);
#endif
      f32x4 x = {xc, xc, xc, xc};
      if constexpr (HAS_SIGNAL) x = __builtin_nontemporal_load(pi + q * inQ);
      f32x4 y;
      y.x = ch.template next_head<FAST_HEAD>(x.x);
      y.y = ch.template next_head<FAST_HEAD>(x.y);
      y.z = ch.template next_head<FAST_HEAD>(x.z);
      y.w = ch.template next_head<FAST_HEAD>(x.w);
      if constexpr (MIX)
      {
        if constexpr (SCALED)
        {
          if (a.mixGains)  // (wave-uniform; the first stage's `y *= g`)
          {
            y.x *= mixGain;
            y.y *= mixGain;
            y.z *= mixGain;
            y.w *= mixGain;
          }
          if (!live) y = f32x4{0.f, 0.f, 0.f, 0.f};  // (a voice the bank does not have counts as +0)
        }
        mix64_park(strip, q & 3, y);
        if ((q & 3) == 3) mix64_sum_store(strip, a.mix + ((v >> 6) * a.T + t) * 64 + (size_t)(q & ~3) * 4);
      }
      else
        __builtin_nontemporal_store(y, po + q * outQ);
    }
    ch.end_vector();
  }
}

template <class CH, bool HAS_SIGNAL, bool MIX = false>
__device__ __forceinline__ void chain_kernel_body(const ChainArgs& a)
{
  apply_fp_mode(a.flags);
  __shared__ float ldsTable[CH::kHasImpulse ? 32 : 1];
  float* const strip = MixStrips<MIX>::mine();
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
  size_t v = blk * kChainBlock + threadIdx.x;
  bool live = true;
  if constexpr (MIX)
  {
    // the last wavefront of a bank that does not fill it keeps its spare lanes: they are part of the tree. They run the bank's last
    // voice again (same inputs, same state, same stores) and put +0 into the sum.
    if ((v & ~(size_t)63) >= a.V) return;
    live = v < a.V;
    if (!live) v = a.V - 1;
  }
  else if (v >= a.V)
    return;

  CH ch;
  const VoiceMem mem{a.coeffs + v, a.state + v, a.V};
  const KernelTables tables{ldsTable};
  ch.load(mem, tables);

  const float xc = (!HAS_SIGNAL && a.inConst) ? a.inConst[v] : 0.f;

  // A launch-constant input lets the head processor (SawGen / PulseGen) skip its per-sample
  // range test: decide once per wavefront which loop body to run.
  bool fastHead = false;
  if constexpr (!HAS_SIGNAL && CH::kHeadHasFastPath) fastHead = (__builtin_amdgcn_ballot_w64(CH::head_input_is_odd(xc)) == 0);
  if constexpr (MIX)
  {
    const bool scaled = a.mixGains != nullptr || __builtin_amdgcn_ballot_w64(!live) != 0;  // (the same in all lanes)
    if (scaled)
    {
      if (fastHead)
        run_voice<CH, HAS_SIGNAL, true, true, true>(ch, a, v, xc, strip, live);
      else
        run_voice<CH, HAS_SIGNAL, false, true, true>(ch, a, v, xc, strip, live);
    }
    else if (fastHead)
      run_voice<CH, HAS_SIGNAL, true, true>(ch, a, v, xc, strip, live);
    else
      run_voice<CH, HAS_SIGNAL, false, true>(ch, a, v, xc, strip, live);
  }
  else if (fastHead)
    run_voice<CH, HAS_SIGNAL, true, MIX>(ch, a, v, xc, strip, live);
  else
    run_voice<CH, HAS_SIGNAL, false, MIX>(ch, a, v, xc, strip, live);
  ch.store(mem);
}

template <class CH, bool HAS_SIGNAL>
__global__ __launch_bounds__(kChainBlock) void chain_kernel(const ChainArgs a)
{
  chain_kernel_body<CH, HAS_SIGNAL>(a);
}

// the same voices, their sum instead of their signals: a.mix gets the group sums mlgpu_mixdown's first stage would have made of
// a.out (same bits: same tree), nothing else is written but the state.
template <class CH, bool HAS_SIGNAL>
__global__ __launch_bounds__(kChainBlock) void chain_mix_kernel(const ChainArgs a)
{
  chain_kernel_body<CH, HAS_SIGNAL, true>(a);
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
//
// PACKED FP32. The N stage updates of a tick are the same ten operations on N independent operand sets, which is what
// gfx950's packed FP32 instructions are for: v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 produce two IEEE f32 results per
// lane (each component rounded exactly like the scalar instruction: same bits) and issue at the scalar instructions' rate
// (tools/valubench.hip). Stages are paired (p, p + N/2): the pair's inputs are then the previous pair's outputs, one
// 64-bit register pair, except pair 0 = {x, r[N/2 - 1]}. A tick is 10 packed instructions per stage PAIR instead of 10
// scalar ones per stage. MLGPU_CASCADE_PACKED=0 keeps the scalar form (A/B measurements, profiles/).
#ifndef MLGPU_CASCADE_PACKED
#define MLGPU_CASCADE_PACKED 1
#endif
#ifndef MLGPU_CASCADE_QUADS_PER_TRIP
#define MLGPU_CASCADE_QUADS_PER_TRIP 4
#endif
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND, int N>
struct SvfCascade
{
  static_assert(N % 2 == 0, "stages are evaluated in pairs");
  static constexpr int NCK = (KIND == MLGPU_PROC_HIPASS) ? 4 : 3;
  static constexpr int NC = NCK * N, NS = 2 * N;
  static constexpr int H = N / 2;
  // component .x of element p belongs to stage p, .y to stage p + H
  f32x2 g0[H], g1[H], g2[H], kk[H], ic1[H], ic2[H], r[H];

  template <class ARR>
  static MLD void put(ARR& arr, int s, float value)
  {
    if (s < H)
      arr[s].x = value;
    else
      arr[s - H].y = value;
  }
  template <class ARR>
  static MLD float get(const ARR& arr, int s) { return s < H ? arr[s].x : arr[s - H].y; }

  MLD void load(const VoiceMem& m)
  {
#pragma unroll
    for (int s = 0; s < N; ++s)
    {
      put(g0, s, m.c(NCK * s));
      put(g1, s, m.c(NCK * s + 1));
      put(g2, s, m.c(NCK * s + 2));
      put(kk, s, (KIND == MLGPU_PROC_HIPASS) ? m.c(NCK * s + 3) : 0.f);
      put(ic1, s, u2f(m.s(2 * s)));
      put(ic2, s, u2f(m.s(2 * s + 1)));
      put(r, s, 0.f);
    }
  }
  MLD void store(const VoiceMem& m) const
  {
#pragma unroll
    for (int s = 0; s < N; ++s)
    {
      m.set(2 * s, f2u(get(ic1, s)));
      m.set(2 * s + 1, f2u(get(ic2, s)));
    }
  }
  // all stages active; returns stage N-1's output, i.e. the chain output for sample (tick - (N-1))
#if MLGPU_CASCADE_PACKED
  MLD float tick(float x) { return tick_in0(f32x2{x, r[H - 1].x}); }
  // in0 = {input of stage 0, input of stage H = r of stage H - 1}
  MLD float tick_in0(f32x2 in0) { return tick_common<false>(in0, in0); }
  // the first pair's t0 = in0 - ic2[0] made by the caller (the lane-group kernel folds a DPP hand-over into that
  // subtraction); not for Hipass, whose output also reads in0
  MLD float tick_t0(f32x2 t00)
  {
    static_assert(KIND != MLGPU_PROC_HIPASS, "Hipass reads its input twice");
    return tick_common<true>(t00, t00);
  }
  template <bool HAVE_T0>
  MLD float tick_common(f32x2 in0, f32x2 t00)
  {
    f32x2 in[H], t0[H], a[H], b[H], c[H], d[H], t1[H], t2[H], v1[H], v2[H];
    const f32x2 two = {2.0f, 2.0f};
    in[0] = in0;
#pragma unroll
    for (int p = 1; p < H; ++p) in[p] = r[p - 1];
#pragma unroll
    for (int p = 0; p < H; ++p) t0[p] = in[p] - ic2[p];
    if constexpr (HAVE_T0) t0[0] = t00;
#pragma unroll
    for (int p = 0; p < H; ++p) b[p] = g1[p] * ic1[p];
#pragma unroll
    for (int p = 0; p < H; ++p) d[p] = g0[p] * ic1[p];
#pragma unroll
    for (int p = 0; p < H; ++p) a[p] = g0[p] * t0[p];
#pragma unroll
    for (int p = 0; p < H; ++p) c[p] = g2[p] * t0[p];
#pragma unroll
    for (int p = 0; p < H; ++p) t1[p] = a[p] + b[p];
#pragma unroll
    for (int p = 0; p < H; ++p) t2[p] = c[p] + d[p];
    if (KIND == MLGPU_PROC_LOPASS)
    {
#pragma unroll
      for (int p = 0; p < H; ++p) r[p] = t2[p] + ic2[p];
    }
    else if (KIND == MLGPU_PROC_BANDPASS)
    {
#pragma unroll
      for (int p = 0; p < H; ++p) r[p] = t1[p] + ic1[p];
    }
    else
    {
#pragma unroll
      for (int p = 0; p < H; ++p) v1[p] = t1[p] + ic1[p];
#pragma unroll
      for (int p = 0; p < H; ++p) v2[p] = t2[p] + ic2[p];
#pragma unroll
      for (int p = 0; p < H; ++p) v1[p] = kk[p] * v1[p];
#pragma unroll
      for (int p = 0; p < H; ++p) v1[p] = in[p] - v1[p];
#pragma unroll
      for (int p = 0; p < H; ++p) r[p] = v1[p] - v2[p];
    }
    // PARITY: fma(2, t, ic) == ic + 2*t (2*t is exact short of overflow; see SvfCore)
#pragma unroll
    for (int p = 0; p < H; ++p) ic1[p] = MLGPU_SVF_STRICT ? ic1[p] + two * t1[p] : __builtin_elementwise_fma(two, t1[p], ic1[p]);
#pragma unroll
    for (int p = 0; p < H; ++p) ic2[p] = MLGPU_SVF_STRICT ? ic2[p] + two * t2[p] : __builtin_elementwise_fma(two, t2[p], ic2[p]);
    return r[H - 1].y;
  }
#else
  MLD float tick(float x)
  {
    float in[N], t0[N], a[N], b[N], c[N], d[N], t1[N], t2[N], v1[N], v2[N];
    in[0] = x;
#pragma unroll
    for (int s = 1; s < N; ++s) in[s] = get(r, s - 1);
#pragma unroll
    for (int s = 0; s < N; ++s) t0[s] = in[s] - get(ic2, s);
#pragma unroll
    for (int s = 0; s < N; ++s) b[s] = get(g1, s) * get(ic1, s);
#pragma unroll
    for (int s = 0; s < N; ++s) d[s] = get(g0, s) * get(ic1, s);
#pragma unroll
    for (int s = 0; s < N; ++s) a[s] = get(g0, s) * t0[s];
#pragma unroll
    for (int s = 0; s < N; ++s) c[s] = get(g2, s) * t0[s];
#pragma unroll
    for (int s = 0; s < N; ++s) t1[s] = a[s] + b[s];
#pragma unroll
    for (int s = 0; s < N; ++s) t2[s] = c[s] + d[s];
    if (KIND == MLGPU_PROC_LOPASS)
    {
#pragma unroll
      for (int s = 0; s < N; ++s) put(r, s, t2[s] + get(ic2, s));
    }
    else if (KIND == MLGPU_PROC_BANDPASS)
    {
#pragma unroll
      for (int s = 0; s < N; ++s) put(r, s, t1[s] + get(ic1, s));
    }
    else
    {
#pragma unroll
      for (int s = 0; s < N; ++s) v1[s] = t1[s] + get(ic1, s);
#pragma unroll
      for (int s = 0; s < N; ++s) v2[s] = t2[s] + get(ic2, s);
#pragma unroll
      for (int s = 0; s < N; ++s) v1[s] = get(kk, s) * v1[s];
#pragma unroll
      for (int s = 0; s < N; ++s) v1[s] = in[s] - v1[s];
#pragma unroll
      for (int s = 0; s < N; ++s) put(r, s, v1[s] - v2[s]);
    }
#pragma unroll
    for (int s = 0; s < N; ++s) put(ic1, s, svf_acc(get(ic1, s), t1[s]));
#pragma unroll
    for (int s = 0; s < N; ++s) put(ic2, s, svf_acc(get(ic2, s), t2[s]));
    return get(r, N - 1);
  }
#endif
  // boundary tick: only stages sLo..sHi (wave-uniform) are active
  MLD float tick_masked(float x, int sLo, int sHi)
  {
    float in[N];
    in[0] = x;
#pragma unroll
    for (int s = 1; s < N; ++s) in[s] = get(r, s - 1);
#pragma unroll
    for (int s = 0; s < N; ++s)
    {
      if (s >= sLo && s <= sHi)
      {
        const float i1 = get(ic1, s), i2 = get(ic2, s);
        const float t0 = in[s] - i2;
        const float t1 = get(g0, s) * t0 + get(g1, s) * i1;
        const float t2 = get(g2, s) * t0 + get(g0, s) * i1;
        if (KIND == MLGPU_PROC_LOPASS)
          put(r, s, t2 + i2);
        else if (KIND == MLGPU_PROC_BANDPASS)
          put(r, s, t1 + i1);
        else
          put(r, s, in[s] - get(kk, s) * (t1 + i1) - (t2 + i2));
        put(ic1, s, svf_acc(i1, t1));
        put(ic2, s, svf_acc(i2, t2));
      }
    }
    return get(r, N - 1);
  }
};

// HEAD = Chain<...> of processors in front of the cascade (e.g. NoiseGen), evaluated unskewed.
template <class HEAD, int KIND, int N, bool HAS_SIGNAL>
__global__ __launch_bounds__(kChainBlock) void cascade_kernel(const ChainArgs a)
{
  static_assert(!HEAD::kHasImpulse, "ImpulseGen heads are not supported by the cascade kernel");
  // with B == 0 the steady loop's last prefetch (inQuad(q + A + QT + 1 + k)) would read one quad past the input
  static_assert((N - 1) % 4 != 0, "cascade lengths with (N - 1) % 4 == 0 need a bounded prefetch");
  apply_fp_mode(a.flags);
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

  // steady state: output quad q <- ticks D+4q .. D+4q+3, inputs x[4(q+A)+B+j]. A trip is QT output quads; the QT input
  // loads of the NEXT trip are issued at the top of a trip and consumed a whole trip (QT * 4 ticks of arithmetic) later.
  // QT = 4 (4 KiB of loads in flight per wave) is the measured optimum on config 4: QT = 8 takes all 256 registers and runs
  // 0.570 ms against 0.477, QT = 16 0.491 (round 2, 131 072 channels x 32 DSPVectors) - the kernel is not waiting for its
  // loads (round 3's account, profiles/archive/r03_cfg4_account.md: it is bound by VALU issue, and the streams cost it clock).
  constexpr int QT = MLGPU_CASCADE_QUADS_PER_TRIP;
  const size_t Q = (S - D) / 4;
  size_t q = 0;
  if constexpr (HAS_SIGNAL)
  {
    f32x4 w[QT + 1];  // input quads q+A .. q+A+QT of the current trip
    if (Q >= QT)
    {
#pragma unroll
      for (int k = 0; k < QT + 1; ++k) w[k] = __builtin_nontemporal_load(inQuad((size_t)(A + k)));
    }
    for (; q + QT <= Q; q += QT)
    {
      const bool more = (q + 2 * QT <= Q);
      f32x4 nx[QT];
      if (more)
      {
#pragma unroll
        for (int k = 0; k < QT; ++k) nx[k] = __builtin_nontemporal_load(inQuad(q + A + QT + 1 + k));
      }
#pragma unroll
      for (int g = 0; g < QT; ++g)
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
        w[0] = w[QT];
#pragma unroll
        for (int k = 0; k < QT; ++k) w[k + 1] = nx[k];
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

// ---------------------------------------------------------------------------------------------
// cascade_lanes_kernel — the stage-skewed cascade of round 3: LPC = 1, 2 or 4 wavefront lanes per channel.
//
// What bounds this kernel (profiles/archive/r03_cfg4_account.md): VALU issue. Its cycle count (GRBM_GUI_ACTIVE) is the same with
// the HBM streams and with every row collapsed onto one cache-resident row; what the streams cost is clock (the chip holds
// ~2.1 GHz on the arithmetic alone and ~1.8 GHz with 4.7 TB/s next to it). A packed-FP32 instruction occupies the SIMD for
// 4 cycles, a plain one for 2, so a tick of 8 sections cannot cost less than 80 multiplies / adds = 160 cycles per
// wavefront; cascade_kernel spends 216, this one 196 with one lane per channel:
//   * a whole DSPVector (16 quads, 64 ticks) per trip: every quad's position and ring slot is a compile-time constant, the
//     pointers advance by launch constants, loads land in the ring slot they are used from (no copies), and the launch's
//     last vector runs through the same code (its fetches are sent back to valid addresses);
//   * the first pair's `input - ic2` as two plain subtractions that read the sample and the previous result where they
//     are, instead of a packed one that needs both moved into a register pair first.
// LPC > 1 spreads a channel over LPC lanes: lane j of a channel's group owns the N / LPC consecutive sections from j * N/LPC
// on and runs them exactly as SvfCascade does, taking as its input what lane j - 1 produced one tick earlier - handed over by
// DPP, folded into the subtraction that consumes it (v_sub_f32_dpp row_shr:4). Same arithmetic per section => same bits.
// It multiplies the wavefronts of a SMALL bank (4 096 channels: 84 us per 32 DSPVectors instead of 204); at 131 072
// channels the extra DPP slot per tick and lane costs more than four wavefronts per SIMD gain (0.49 against 0.46 ms), so
// chains.hip picks LPC from the bank's size.
//
// Lane layout inside a wavefront: lane = 16 * row + 4 * bank + cc. A DPP bank (4 consecutive lanes) is the unit the
// bank_mask of a DPP instruction can address, so the section group j has to be the bank: j = bank % LPC, and the channel is
// (lane / (4 * LPC)) * 4 + cc. "From the previous group" is then row_shr:4 under a bank mask that spares the group's first
// lane, which keeps the result computed from its own sample.
// (Measured and not kept, profiles/archive/r03_cascade_lds_handover.txt: the hand-over through LDS - ds_write2 + ds_read per tick, no
// VALU instruction - is slower than DPP, 0.505 against 0.491 ms with two lanes per channel.)
//
// Memory: 16 bytes per lane per quad as before. Every lane of a group fetches the channel's input quads (the same
// addresses: one fetch serves the group, only the first lane's copy is used - a load under a lane mask would make the whole
// ring a phi of the branch), a ring of R quads in registers, R - 2 quads ahead of their use; the group's last lane stores.
template <int LPC>
struct LaneGroup
{
  static_assert(LPC == 1 || LPC == 2 || LPC == 4, "1, 2 or 4 lanes per channel");
  static constexpr int kFirst = (LPC == 4) ? 0x1 : (LPC == 2 ? 0x5 : 0xF);  // bank mask of the lanes with j == 0
  static constexpr int kNotFirst = 0xF & ~kFirst;
  // the input of this lane's first stage: in the group's first lane its own `sample`; elsewhere what lane j - 1 holds in
  // `prevOut` (its last stage's output of the previous tick)
  static MLD float stageInput(float prevOut, float sample)
  {
    if constexpr (LPC == 1) return sample;
    else return u2f((uint32_t)__builtin_amdgcn_update_dpp((int)f2u(sample), (int)f2u(prevOut), 0x114 /* row_shr:4 */, 0xF, kNotFirst, false));
  }
};

#ifndef MLGPU_CASCADE_LANES_DPP_SUB
#define MLGPU_CASCADE_LANES_DPP_SUB 1
#endif

template <int KIND, int N, int LPC, int R, bool HAS_SIGNAL>
__device__ __forceinline__ void cascade_lanes_body(const ChainArgs& a)
{
  constexpr int SPL = N / LPC;  // stages per lane
  static_assert(N % LPC == 0 && SPL % 2 == 0, "each lane runs an even number of stages (packed pairs)");
  static_assert(R == 4 || R == 8 || R == 16, "the input ring: R quads, R divides a DSPVector's 16 so that slots are compile-time");
  using LG = LaneGroup<LPC>;
  using Core = SvfCascade<KIND, SPL>;
  constexpr int NCK = Core::NCK, H = Core::H;
  constexpr int kChannelsPerBlock = kChainBlock / LPC;
  apply_fp_mode(a.flags);
  size_t blk = blockIdx.x;
  const size_t nbFull = (size_t)gridDim.x & ~(size_t)7;
  if (blk < nbFull) blk = (blk & 7) * (nbFull >> 3) + (blk >> 3);  // XCD-aware, see chain_kernel
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = (lane >> 2) % LPC;                                   // which stage group of its channel this lane runs
  const size_t v = blk * kChannelsPerBlock + (size_t)wave * (64 / LPC) + (lane / (4 * LPC)) * 4 + (lane & 3);
  if (v >= a.V) return;  // whole groups leave together: all LPC lanes of a channel share v
  const bool last = (j == LPC - 1);

  Core c;
  const int base = j * SPL;  // first stage of this lane
  c.load(VoiceMem{a.coeffs + (size_t)(NCK * base) * a.V + v, a.state + (size_t)(2 * base) * a.V + v, a.V});
  const float xc = (!HAS_SIGNAL && a.inConst) ? a.inConst[v] : 0.f;

  constexpr int D = N - 1;             // output lag in ticks
  constexpr int A = D / 4, B = D % 4;  // D = 4A + B
  static_assert(A + R - 2 <= 15, "the ring is primed from the first DSPVector");
  const size_t S = a.T * 64;
  // every lane of a group reads the channel's input (the same addresses: one fetch serves the group; only the first
  // lane's copy is used) — a load under a lane mask would make the whole ring a phi of the branch
  const f32x4* pin = HAS_SIGNAL ? (const f32x4*)a.in.base + v * a.in.strideV : nullptr;
  f32x4* pout = (f32x4*)a.out.base + v * a.out.strideV;
  auto inQuad = [&](size_t qi) { return pin + (qi >> 4) * a.in.strideT + (qi & 15) * a.in.strideQ; };
  auto outQuad = [&](size_t qi) { return pout + (qi >> 4) * a.out.strideT + (qi & 15) * a.out.strideQ; };
  auto inAt = [&](size_t i) -> float {
    if constexpr (HAS_SIGNAL) return ((const float*)inQuad(i >> 2))[i & 3];
    return xc;
  };
  // one tick with every stage active: the lane's first stage reads `x` (group's first lane) or the previous lane's output
  auto fastTick = [&](float x) {
    const f32x2 rr = c.r[H - 1];
    if constexpr (KIND != MLGPU_PROC_HIPASS && MLGPU_CASCADE_LANES_DPP_SUB)
    {
      // t0 of the lane's first stage = input - ic2: with the group's own sample everywhere, then again in the lanes that
      // take the previous lane's output, the DPP hand-over folded into the subtraction (v_sub_f32_dpp; masked lanes keep
      // the first result). Two wait states between a VALU write and a DPP read of the same register (s_nop 1).
      float t0lo = x - c.ic2[0].x;
      if constexpr (LPC == 1) asm("" : "+v"(t0lo));
      else if constexpr (LPC == 2) asm("s_nop 1\n\tv_sub_f32_dpp %0, %1, %2 row_shr:4 row_mask:0xf bank_mask:0xa" : "+v"(t0lo) : "v"(rr.y), "v"(c.ic2[0].x));
      else asm("s_nop 1\n\tv_sub_f32_dpp %0, %1, %2 row_shr:4 row_mask:0xf bank_mask:0xe" : "+v"(t0lo) : "v"(rr.y), "v"(c.ic2[0].x));
      float t0hi = rr.x - c.ic2[0].y;
      asm("" : "+v"(t0hi));  // keeps this a plain v_sub_f32 into the pair's high half (the compiler would otherwise make it a packed subtraction plus a move)
      return c.tick_t0(f32x2{t0lo, t0hi});
    }
    else
    {
      const f32x2 rx = f32x2{rr.x, LG::stageInput(rr.y, x)};  // the DPP move is in place: the tick reads the pair swapped
      return c.tick_in0(__builtin_shufflevector(rx, rx, 1, 0));
    }
  };
  // output quad q <- ticks D + 4q + jj, inputs x[4(q + A) + B + jj]: elements of input quads q + A (wa) and q + A + 1 (wb)
  auto quadTicks = [&](const f32x4& wa, const f32x4& wb) {
    f32x4 y;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
    {
      constexpr int kB = B;
      const int e = kB + jj;
      y[jj] = fastTick(e < 4 ? wa[e & 3] : wb[e & 3]);
    }
    return y;
  };
  // boundary tick (slow form): stages sLo..sHi of the whole cascade are active
  auto slowTick = [&](float x, int sLo, int sHi) { c.tick_masked(LG::stageInput(c.r[H - 1].y, x), sLo - base, sHi - base); };

  // prologue: ticks 0..D-1, stages 0..i active
  for (int i = 0; i < D; ++i) slowTick(inAt((size_t)i), 0, i);

  // steady state: one DSPVector = 16 quads = 64 ticks per trip, so every quad's position in its vector and its ring slot
  // (input quad n lives in slot n % R) are compile-time constants and the pointers advance by launch constants. While
  // output quad q is computed, input quad q + A + R - 1 is fetched into the slot that quad q + A - 1 has just left; near
  // the end of a vector the fetches reach into the next one. The launch's last vector has no next one: there the fetch
  // pointer is sent back to the start of the same vector (valid addresses, values never used), and its last
  // kTailQuads output quads - the ones whose ticks reach past the end of the input - are left to the tail below.
  const size_t Q = (S - D) / 4;
  constexpr int kTailQuads = A + 1;  // 16 T - Q
  {
    const int64_t inStep = (int64_t)a.in.strideQ * 16, inNext = ((int64_t)a.in.strideT - 15 * (int64_t)a.in.strideQ) * 16, inRewind = -15 * inStep;
    const int64_t outStep = (int64_t)a.out.strideQ * 16, outNext = ((int64_t)a.out.strideT - 15 * (int64_t)a.out.strideQ) * 16;
    f32x4 w[R];
    const char* pf = (const char*)pin;  // the next input quad to fetch
    char* ps = (char*)pout;             // the next output quad to store
    if constexpr (HAS_SIGNAL)
    {
      pf += A * inStep;
#pragma unroll
      for (int n = 0; n < R - 1; ++n)  // quads A .. A + R - 2 of the first vector
      {
        w[(A + n) % R] = __builtin_nontemporal_load((const f32x4*)pf);
        pf += ((A + n) & 15) == 15 ? (a.T == 1 ? inRewind : inNext) : inStep;
      }
    }
    auto quadStep = [&](int s, int64_t inWrap0, int64_t inWrap1) {  // s is a constant after unrolling
      f32x4 y;
      if constexpr (HAS_SIGNAL)
      {
        constexpr int kA = A;
        const int n = s + kA + R - 1;  // the quad fetched now (relative to this vector's first)
        w[n % R] = __builtin_nontemporal_load((const f32x4*)pf);
        pf += (n & 15) == 15 ? (n < 16 ? inWrap0 : inWrap1) : inStep;  // leaving vector t (n == 15) or t + 1 (n == 31, R = 16 only)
        y = quadTicks(w[(s + kA) % R], w[(s + kA + 1) % R]);
      }
      else
      {
        const f32x4 wc = f32x4{xc, xc, xc, xc};
        y = quadTicks(wc, wc);
      }
      if (last) __builtin_nontemporal_store(y, (f32x4*)ps);
      ps += s == 15 ? outNext : outStep;
    };
#if MLGPU_CHAIN_TURNS
    const uint32_t slot = wave_slot();
#endif
    for (size_t t = 0; t < a.T; ++t)
    {
#if MLGPU_CHAIN_TURNS
      take_turns_by_clock(slot, kTurnClockShift);
#endif
      const bool lastVector = (t + 1 == a.T);
      // where the fetch pointer goes when it leaves a vector: on to the next one, or - from the launch's last vector - back
      // to that vector's start
      const int64_t inWrap0 = (t + 1 < a.T) ? inNext : inRewind, inWrap1 = (t + 2 < a.T) ? inNext : inRewind;
#pragma unroll
      for (int s = 0; s < 16 - kTailQuads; ++s) quadStep(s, inWrap0, inWrap1);
      if (!lastVector)
      {
#pragma unroll
        for (int s = 16 - kTailQuads; s < 16; ++s) quadStep(s, inWrap0, inWrap1);
      }
    }
  }
  // tail: ticks D + 4Q .. S + D - 1; inputs exist while i < S, after that the pipeline drains
  for (size_t i = D + 4 * Q; i < S + D; ++i)
  {
    const float x = (i < S) ? inAt(i) : 0.f;
    const int sLo = (i < S) ? 0 : (int)(i - S + 1);
    slowTick(x, sLo, N - 1);
    const size_t n = i - D;
    if (last) ((float*)outQuad(n >> 2))[n & 3] = c.r[H - 1].y;
  }
  c.store(VoiceMem{a.coeffs + (size_t)(NCK * base) * a.V + v, a.state + (size_t)(2 * base) * a.V + v, a.V});
}

// MINW = wavefronts per SIMD the register allocation must leave room for: 2 / 4 / 6 for 1 / 2 / 4 lanes per channel
template <int KIND, int N, int LPC, int R, int MINW, bool HAS_SIGNAL>
__global__ __launch_bounds__(kChainBlock) __attribute__((amdgpu_waves_per_eu(MINW))) void cascade_lanes_kernel(const ChainArgs a)
{
  cascade_lanes_body<KIND, N, LPC, R, HAS_SIGNAL>(a);
}

}  // namespace mldev
