// mldsp_procs.hpp — the stateful processors of mldsp.h as per-lane, per-sample device objects.
//
// Each struct restates one reference functor (source/DSP/MLDSPGens.h, MLDSPFilters.h):
//   load()   coefficients + state  HBM (SoA [slot][V]) -> registers
//   next(x)  one sample; x is the processor's audio-rate input (cyclesPerSample for
//            generators, exactly the reference's operator() argument)
//   end_vector()  bookkeeping the reference does once per 64-sample DSPVector
//   store()  state registers -> HBM
// One wavefront lane owns one voice for the whole launch, so state never leaves
// registers while the DSPVectors are walked serially.
//
// Bit-parity notes are marked PARITY. This file must be compiled with -ffp-contract=off.
#pragma once
#include "mldsp_math.hpp"
#ifdef __HIPCC_RTC__
#include "mlgpu.h"  // provided as an in-memory header by graph.hip
#else
#include "../../include/mlgpu.h"
#endif

namespace mldev
{
// coefficient / state views of one voice: element i of this processor lives at base[i*V]
struct VoiceMem
{
  const float* coeffs;  // already offset to (this processor's first slot)*V + v
  uint32_t* state;      // idem
  size_t V;
  // delay-line memory of this processor (nullptr for processors without any): rings of `memMask + 1` floats per voice,
  // SoA [sample][V] like coefficients and state (already offset to + v), so the write of one time step is one coalesced
  // 256-byte row per wavefront and the reads are coalesced whenever neighbouring voices use the same delay
  float* mem{nullptr};
  uint32_t memMask{0};
// Sample i of ring ringIdx. Rows behind 32-bit offsets (addr32, set by the generated kernel where a ring of the bank is at most
  // 4 GiB and the bank below 2^22 voices): `mem` is then the node's memory for the whole bank - wave-uniform, a scalar register pair,
  // and so is the ring's start in it - and a row's place i * 4 V + 4 v one v_mad_u32_u24 on top of that, where the 64-bit form costs
  // two v_mad_u64_u32 and a 64-bit add per access (the reference's reverb example: 170 of its 1 300 instructions per sample).
  MLD float* ringPtr(int ringIdx, uint32_t i) const
  {
    if (addr32) return (float*)((char*)(mem + (size_t)((uint32_t)ringIdx * (memMask + 1)) * V) + (__umul24(i, V4) + lane4));
    return mem + (size_t)((uint32_t)ringIdx * (memMask + 1) + i) * V;
  }
  MLD float ring(int ringIdx, uint32_t i) const { return *ringPtr(ringIdx, i); }
  MLD void ringSet(int ringIdx, uint32_t i, float x) const { *ringPtr(ringIdx, i) = x; }
  // Windowed rings (mlgpu_graph_set_delay_layout, the generated source defines MLGPU_RING_WINDOWS 1): `mem` is this lane's
  // sector of chunk 0 in the block's [chunk][lane][8] ring storage and `lds` this lane's column of the workgroup's write
  // windows, [ring][kRingWindow][256 lanes] (RingCore below).
  float* lds{nullptr};
  uint32_t ldsHist{0};  // ring layout 4: where this node's history rows start, in floats from `lds` (after its rings' held sectors)
  uint32_t lane4{0}, V4{0};  // addr32: 4 * the lane's voice, 4 * the bank's voices
  bool addr32{false};
  MLD float c(int i) const { return coeffs[(size_t)i * V]; }
  MLD uint32_t s(int i) const { return state[(size_t)i * V]; }
  MLD void set(int i, uint32_t x) const { state[(size_t)i * V] = x; }
};

struct KernelTables
{
  const float* impulse_table;  // 17 floats, built on the host with libm (MLDSPGens.h:65-78)
};

constexpr float kStepsPerCycle = 4294967296.0f;            // MLDSPGens.h:184  2^32
constexpr float kCyclesPerStep = 2.3283064365386963e-10f;  // MLDSPGens.h:185  2^-32

// PARITY: const_math::sqrt(2.0f) in the reference is itself an approximation (0x3fb50505,
// not 0x3fb504f3); these are the reference's constexpr results (MLDSPGens.h:318-327).
constexpr uint32_t kSqrt2Bits = 0x3fb50505u;
constexpr uint32_t kSineDomainBits = 0x40b50505u;
constexpr uint32_t kSineScaleBits = 0x3f87c3b6u;
constexpr uint32_t kSineFlipBits = 0x40350505u;
constexpr uint32_t kOneSixthBits = 0x3e2aaaabu;

// PhasorGen core, MLDSPGens.h:187-203.
// PARITY: unsignedIntToFloat(u)*2^-32 is (hi + hi) * 2^-32 with hi = float(int(u >> 1)); both
// steps are exact power-of-two scalings of hi (hi is 0 or >= 1, so nothing goes denormal), hence
// hi * 2^-31 is the same float with one multiply instead of an add and a multiply.
MLD float phasor_next(uint32_t& omega32, float cyclesPerSample)
{
  const float steps = cyclesPerSample * kStepsPerCycle;
  const int32_t istep = sse_cvt(steps);  // roundFloatToInt; loop-invariant when cps is
  omega32 += (uint32_t)istep;
  const float hi = (float)(int32_t)(omega32 >> 1);
  return hi * 4.656612873077392578125e-10f;  // 2^-31
}

// The same counter step, handing back `hi` itself - the phase times 2^31. Where the consumer only compares the phase, scales it or
// feeds it to one multiply-add, the constants on the other side carry the 2^-31 instead (power-of-two scalings are exact, and hi is 0
// or >= 1: nothing goes denormal), and the oscillator is one instruction per sample shorter:
//   p < a  <=>  hi < a * 2^31;    RN(p * c) = RN(hi * (c * 2^-31));    RN(p - w) = fma(hi, 2^-31, -w)   (p itself is exact)
constexpr float kPhaseUnit = 4.656612873077392578125e-10f;  // 2^-31
constexpr float kPhaseScale = 2147483648.0f;                 // 2^31
MLD float phasor_next_hi(uint32_t& omega32, float cyclesPerSample)
{
  const float steps = cyclesPerSample * kStepsPerCycle;
  const int32_t istep = sse_cvt(steps);
  omega32 += (uint32_t)istep;
  asm("" : "+v"(omega32));  // one add per sample: left to itself the compiler makes sample k's phase from base + k * step (two or three instructions)
  return (float)(int32_t)(omega32 >> 1);
}

// Correctly rounded n/d for the polyBLEP operands. This is the Newton-Raphson sequence the
// compiler itself expands an IEEE f32 division to on gfx9 (v_rcp_f32, one reciprocal refinement,
// two quotient refinements, all with exact FMA residuals) without the v_div_scale / v_div_fixup
// range handling, so it returns the same bits as `n / d` whenever no scaling is needed:
// d in [2^-64, 2^64] and |n| in [2^-31, 2] or 0 here, so quotient and residuals stay normal.
// Everything that depends only on d (rcp + 2 FMA) is loop-invariant for a per-voice constant
// frequency and is hoisted by the compiler: 5 VALU ops per sample instead of ~12.
MLD float div_nr(float n, float d)
{
  const float r0 = __builtin_amdgcn_rcpf(d);
  const float e = __builtin_fmaf(-d, r0, 1.0f);
  const float r1 = __builtin_fmaf(e, r0, r0);
  const float q0 = n * r1;
  const float rem0 = __builtin_fmaf(-d, q0, n);
  const float q1 = __builtin_fmaf(rem0, r1, q0);
  const float rem1 = __builtin_fmaf(-d, q1, n);
  return __builtin_fmaf(rem1, r1, q1);
}

// polyBLEP, MLDSPGens.h:285-311, branch-free. Exactly one of the reference's two divisions is
// ever taken for a sample, so: select the numerator, divide once, evaluate both polynomials,
// select. Lanes that need no correction compute a value that is discarded (it may be NaN for
// dt <= 0; v_cndmask does not care).
// PARITY: `t + t - t*t - 1` == fma(2, t, -(t*t)) - 1 because t + t is exact.
//
// A frequency outside [2^-64, 2^64] (absurd, but legal input) needs the full IEEE division.
// FAST = true promises the caller has excluded that for the whole wavefront (the voice-bank
// kernel tests the per-voice constant frequency once per launch); FAST = false tests per sample
// with a wave-uniform ballot (streamed, time-varying frequency).
MLD bool blep_freq_is_odd(float dt) { return (dt > 0.f) && !((dt >= 0x1p-64f) && (dt <= 0x1p+64f)); }

// The part of polyBLEP that depends on the frequency alone. With a launch-constant frequency all of it (and the
// reciprocal inside div_nr) leaves the sample loop; PulseGen's two corrections of one sample share one.
// The same question for a frequency that changes every sample (FAST = false), asked only where a correction is really evaluated:
// "is the frequency outside [2^-64, 2^64]" as ONE unsigned range test of its bit pattern - no scalar mask logic, no divergent
// short-circuit. It also answers yes for zero, negative and NaN frequencies, which blep_freq_is_odd leaves alone: such a lane is
// never inside a zone itself (t < dt and t > 1 - dt are both false), it only sends the wavefronts it sits in to the reference's own
// IEEE division - the same bits for every lane div_nr is exact for, a few instructions more.
MLD bool blep_freq_not_regular(float dt) { return (f2u(dt) - 0x1F800000u) > (0x5F800000u - 0x1F800000u); }

template <bool FAST>
struct BlepFreq
{
  float dt, omdt;
  bool otherOdd;  // this lane's other operands (an op form's phase, a pulse's shifted phase) are outside div_nr's ranges
  static MLD BlepFreq make(float dt, bool otherwiseOdd = false)
  {
    BlepFreq f;
    f.dt = dt;
    f.omdt = 1.0f - dt;
    f.otherOdd = otherwiseOdd;  // FAST vouches for the frequency, not for the caller's phase
    return f;
  }
  MLD bool lo(float t) const { return t < dt; }
  MLD bool hi(float t) const { return t > omdt; }  // only consulted when !lo (the reference's else-if)
  // wave-uniform: does any lane need the IEEE division? Asked only where a correction is really evaluated (after the
  // skip test), never on the quiet path
  MLD bool anyLaneOdd() const { return __builtin_amdgcn_ballot_w64((!FAST && blep_freq_not_regular(dt)) || otherOdd) != 0; }  // (a FAST caller with no phase of its own: constant false)
  // the correction for a phase already known to be in the lower (isLo) or upper zone; garbage (never used) elsewhere
  MLD float correction(float t, bool isLo, bool full) const
  {
    const float num = isLo ? t : (t - 1.0f);
    // PARITY: `t + t - t*t - 1` == fma(2, t, -(t*t)) - 1 while t + t is exact, i.e. short of overflow: |t| <= 1 for every
    // lane of a wavefront that is not `full` (an oscillator's phase, or an op's operands inside their ranges); a `full`
    // wavefront (some lane has absurd operands - t may be 3e38, where t + t overflows but 2t - t*t does not) takes the
    // reference's own operation order along with its division
    float q;
    if (full)
      q = num / dt;
    else
      q = div_nr(num, dt);
    const float qq = q * q;
    const float twoq_qq = full ? ((q + q) - qq) : __builtin_fmaf(2.0f, q, -qq);   // `full` is wave-uniform: a scalar branch
    const float clo = twoq_qq - 1.0f;
    const float chi = ((qq + q) + q) + 1.0f;
    return isLo ? clo : chi;
  }
};

// A step is near for a fraction 2 * dt of the samples; neighbouring voices have free-running phases, so for low and
// middle frequencies most samples find no lane of the wavefront in either zone: with SKIP one scalar branch skips the
// division and both polynomials (the skipped lanes would have selected 0 anyway). SKIP is set only where the whole
// division is per sample (a streamed frequency: +12 % on the instrument-bank pipeline); with a launch-constant frequency
// the remaining work is too short for a branch to pay (config 3: -7 %), so those paths keep straight-line code.
template <bool SKIP, bool FAST>
MLD float poly_blep(float t, const BlepFreq<FAST>& f)
{
  const bool lo = f.lo(t), hi = f.hi(t);
  if (SKIP && __builtin_amdgcn_ballot_w64(lo || hi) == 0) return 0.f;
  const float c = f.correction(t, lo, f.anyLaneOdd());
  return (lo || hi) ? c : 0.f;
}
template <bool FAST, bool SKIP = false>
MLD float poly_blep(float t, float dt)
{
  return poly_blep<SKIP>(t, BlepFreq<FAST>::make(dt));
}

// phasorToSaw, MLDSPGens.h:362-369
// ANY_PHASE: p is a caller's float, not a PhasorGen's output in [0, 1) (the op forms, mldsp_ops.hpp)
// (div_nr wants |numerator| in [2^-31, 2] or 0: a PhasorGen's phase is a multiple of 2^-31; a caller's float may be 1e-36)
MLD bool phase_is_odd(float p) { return !(((p >= 0x1p-31f) && (p <= 1.0f)) || (p == 0.f)); }

template <bool FAST, bool SKIP = false, bool ANY_PHASE = false>
MLD float phasor_to_saw(float p, float cps)
{
  // PARITY: p*2 is exact (short of overflow: then the product is inf either way), so one rounding either way
  const float saw = ANY_PHASE ? (p * 2.f - 1.f) : __builtin_fmaf(p, 2.f, -1.f);
  return saw - poly_blep<SKIP>(p, BlepFreq<FAST>::make(cps, ANY_PHASE && phase_is_odd(p)));
}

#ifndef MLGPU_PULSE_SINGLE_BLEP
#define MLGPU_PULSE_SINGLE_BLEP 1
#endif
// phasorToPulse, MLDSPGens.h:342-358: +-1 by pulse width, plus the correction of the rising step at phase 0, minus the
// correction of the falling step at phase = width (the phase shifted by 1 - width). Written out that is two polyBLEPs per
// sample - but a lane is almost never inside both zones at once (only when the width is within one sample of 0 or 1), and
// when no lane of the wavefront is, ONE evaluation serves: the phase of whichever step is near goes in, the result is
// added or subtracted. Bits: the reference computes (pulse + c_up) - c_down with the idle correction exactly 0.f, and
// x + 0 == x, x - 0 == x for the +-1 / finite values here, so pulse + c_up or pulse - c_down is the same float.
// REGULAR_W: the caller has excluded, for the whole wavefront, every pulse width outside [0, 1] (and NaN): with a PhasorGen's p
// in [0, 1) the shifted phase d = p - w + 1 is then in [0, 2], where fractionalPart's d - float(trunc(d)) (d, or d - 1: both
// exact) is what v_fract_f32 returns - one instruction instead of sse_cvtt's six, a conversion back and a subtraction. Any other
// width takes the general path.
MLD bool pulse_width_is_odd(float w) { return !(w >= 0.f && w <= 1.0f); }

template <bool FAST, bool SKIP = false, bool ANY_PHASE = false, bool REGULAR_W = false>
MLD float phasor_to_pulse(float p, float cps, float w)
{
  const float pulse = (p >= w) ? -1.f : 1.f;
  const float d = p - w + 1.0f;
  const float down = REGULAR_W ? __builtin_amdgcn_fractf(d) : d - (float)sse_cvtt(d);  // fractionalPart
  // a regular width leaves `down` inside [0, 1): div_nr's ground. Other widths can make it negative, huge, infinite or NaN, and
  // the reference's own division has to produce whatever comes out of that.
  const bool downOdd = ANY_PHASE ? phase_is_odd(down) : (!REGULAR_W && !(abs_ps(down) <= 2.0f));
  const BlepFreq<FAST> f = BlepFreq<FAST>::make(cps, (ANY_PHASE && phase_is_odd(p)) || downOdd);
  const bool loUp = f.lo(p), nearUp = loUp || f.hi(p);
  const bool loDown = f.lo(down), nearDown = loDown || f.hi(down);
  if (SKIP && __builtin_amdgcn_ballot_w64(nearUp || nearDown) == 0) return pulse;
  const bool full = f.anyLaneOdd();
  // (a `full` wavefront of the op forms may hold non-finite corrections, for which x + 0 == x does not hold: two evaluations)
  if (MLGPU_PULSE_SINGLE_BLEP && !(ANY_PHASE && full) && __builtin_amdgcn_ballot_w64(nearUp && nearDown) == 0)
  {
    const float c = f.correction(nearDown ? down : p, nearDown ? loDown : loUp, full);
    return nearDown ? (pulse - c) : (nearUp ? (pulse + c) : pulse);
  }
  const float cUp = f.correction(p, loUp, full), cDown = f.correction(down, loDown, full);
  return (pulse + (nearUp ? cUp : 0.f)) - (nearDown ? cDown : 0.f);
}

// ---- the corrections of a whole trip of N samples at once ("sparse polyBLEP") ------------------------------------------------
// With a launch-constant frequency 0 < dt <= 1 / (2 N) an oscillator wraps at most once in N consecutive samples, so at most one
// sample of the trip lies in the zone after a step (t < dt) and at most one in the zone before it (t > 1 - dt) - and they are the
// samples with the smallest / the largest phase of the trip. The division and the polynomial are then evaluated once per zone per
// trip (for the extremes) instead of once per sample, and each sample takes what its own zone tests select: the same operands
// through the same operations as the per-sample form, so the same bits. "At most one" can fail only next to a rounding knife edge,
// where a phase comes out within a few 2^-24 of 0 or 1 (the computed phase differs from the exact one, which does advance by
// the same step every sample, by at most 2^-25 for a PhasorGen's output and 2^-23 for PulseGen's shifted phase): a trip whose
// extremes come that close - about one in a thousand - is evaluated per sample instead (wave-uniform). DESIGN.md 3.4.
constexpr float kTripMaxFreq(int n) { return 0.5f / (float)n; }
constexpr float kTripTiny = 0x1p-22f, kTripNearOne = 1.0f - 0x1p-22f;          // a PhasorGen's phase
constexpr float kTripTinyShifted = 0x1p-21f, kTripNearOneShifted = 1.0f - 0x1p-21f;  // fractionalPart(p - w + 1)
MLD bool trip_freq_is_dense(float dt, int n) { return !(dt > 0.f && dt <= kTripMaxFreq(n)); }
// phases are >= 0: their order as unsigned bit patterns is their order as floats (and an integer min needs no canonicalising)
MLD uint32_t trip_umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
MLD uint32_t trip_umax(uint32_t a, uint32_t b) { return a > b ? a : b; }

MLD float phasor_to_sine(float p)  // MLDSPGens.h:316-338
{
  const float sqrt2 = u2f(kSqrt2Bits);
  const float omega = p * u2f(kSineDomainBits) + (-sqrt2);
  const float tri = (omega > sqrt2) ? (u2f(kSineFlipBits) - omega) : omega;
  return (u2f(kSineScaleBits) * tri) * (1.0f - (tri * tri) * u2f(kOneSixthBits));
}

// SineGen's own form: the phase still scaled by 2^31 (phasor_next_hi), so the domain scale carries the 2^-31 (one multiply less),
// and the triangle fold as a minimum: for omega > sqrt2 the reflection flip - omega is below sqrt2 (the smaller one), for omega <
// sqrt2 it is above, and at omega == sqrt2 both are sqrt2 (flip = 2 sqrt2 exactly) - min(omega, flip - omega) is the reference's
// select, value for value; omega is a difference of floats near sqrt2 or larger, never NaN, never denormal, so v_min_f32 returns
// one of its operands unchanged in both denormal modes.
MLD float phasor_hi_to_sine(float hi)
{
  const float sqrt2 = u2f(kSqrt2Bits);
  const float omega = hi * u2f(kSineDomainBits - (31u << 23)) + (-sqrt2);
  const float refl = u2f(kSineFlipBits) - omega;
  float tri;
  asm("v_min_f32 %0, %1, %2" : "=v"(tri) : "v"(omega), "v"(refl));
  return (u2f(kSineScaleBits) * tri) * (1.0f - (tri * tri) * u2f(kOneSixthBits));
}

template <int KIND>
struct Proc;

// ---- generators -----------------------------------------------------------------------

template <>
struct Proc<MLGPU_PROC_PHASOR_GEN>
{
  static constexpr int NC = 0, NS = 1;
  uint32_t omega32;
  MLD void load(const VoiceMem& m, const KernelTables&) { omega32 = m.s(0); }
  MLD void store(const VoiceMem& m) const { m.set(0, omega32); }
  MLD float next(float cps) { return phasor_next(omega32, cps); }
  MLD void end_vector() {}
};

template <>
struct Proc<MLGPU_PROC_SINE_GEN>  // MLDSPGens.h:373-381
{
  static constexpr int NC = 0, NS = 1;
  uint32_t omega32;
  MLD void load(const VoiceMem& m, const KernelTables&) { omega32 = m.s(0); }
  MLD void store(const VoiceMem& m) const { m.set(0, omega32); }
  MLD float next(float cps) { return phasor_hi_to_sine(phasor_next_hi(omega32, cps)); }
  MLD void end_vector() {}
};

template <>
struct Proc<MLGPU_PROC_TEST_SINE_GEN>  // TestSineGen, MLDSPGens.h:151-171: the slow, precise reference oscillator
{
  static constexpr int NC = 0, NS = 1;
  float omega;
  MLD void load(const VoiceMem& m, const KernelTables&) { omega = u2f(m.s(0)); }
  MLD void store(const VoiceMem& m) const { m.set(0, f2u(omega)); }
  MLD float next(float cps)
  {
    const float twoPi = 6.2831853071795864769252867f;  // ml::kTwoPi, MLDSPScalarMath.h:23
    const float step = twoPi * cps;
    omega += step;
    if (omega > twoPi) omega -= twoPi;
    return libm_sinf(omega);
  }
  MLD void end_vector() {}
};

template <>
struct Proc<MLGPU_PROC_SAW_GEN>  // MLDSPGens.h:395-402, phasorToSaw :362-369
{
  static constexpr int NC = 0, NS = 1;
  uint32_t omega32;
  MLD void load(const VoiceMem& m, const KernelTables&) { omega32 = m.s(0); }
  MLD void store(const VoiceMem& m) const { m.set(0, omega32); }
  template <bool FAST, bool SKIP = false>
  MLD float step(float cps)
  {
    return phasor_to_saw<FAST, SKIP>(phasor_next(omega32, cps), cps);
  }
  MLD float next(float cps) { return step<false, true>(cps); }
  MLD float next_fast(float cps) { return step<true>(cps); }
  // launch-constant cps whose range test `odd` (wave-uniform) was done once by the caller
  MLD float next_u(float cps, bool odd) { return odd ? step<false>(cps) : step<true>(cps); }
  static MLD bool input_is_odd(float cps) { return blep_freq_is_odd(cps); }
  // N samples at once for a launch-constant cps (graph kernels). `dense` (wave-uniform, tested once by the caller): some lane is
  // `odd` or its frequency is outside (0, 1 / 2N]: the trip is N per-sample evaluations.
  template <int N>
  MLD void trip_u(float cps, bool odd, bool dense, float (&out)[N])
  {
    if (!dense)
    {
      // phases times 2^31 (phasor_next_hi): the zone bounds and the saw's slope carry the scale, the two extremes are scaled back
      const BlepFreq<true> f = BlepFreq<true>::make(cps);
      const float dtS = f.dt * kPhaseScale, omdtS = f.omdt * kPhaseScale;
      const uint32_t before = omega32;
      float h[N];
      uint32_t lo = 0u, hi = 0u;
#pragma unroll
      for (int i = 0; i < N; ++i)
      {
        h[i] = phasor_next_hi(omega32, cps);
        lo = i ? trip_umin(lo, f2u(h[i])) : f2u(h[i]);
        hi = i ? trip_umax(hi, f2u(h[i])) : f2u(h[i]);
      }
      if (__builtin_amdgcn_ballot_w64(u2f(lo) < kTripTiny * kPhaseScale || u2f(hi) > kTripNearOne * kPhaseScale) == 0)
      {
        const float cLo = f.correction(u2f(lo) * kPhaseUnit, true, false), cHi = f.correction(u2f(hi) * kPhaseUnit, false, false);
#pragma unroll
        for (int i = 0; i < N; ++i)
        {
          const float saw = __builtin_fmaf(h[i], 2.f * kPhaseUnit, -1.f);
          out[i] = saw - ((h[i] < dtS) ? cLo : ((h[i] > omdtS) ? cHi : 0.f));
        }
        return;
      }
      omega32 = before;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) out[i] = next_u(cps, odd);
  }
  MLD void end_vector() {}
};

template <>
struct Proc<MLGPU_PROC_PULSE_GEN>  // MLDSPGens.h:383-393, phasorToPulse :342-358
{
  static constexpr int NC = 1, NS = 1;
  uint32_t omega32;
  float width;
  MLD void load(const VoiceMem& m, const KernelTables&)
  {
    width = m.c(0);
    omega32 = m.s(0);
  }
  MLD void store(const VoiceMem& m) const { m.set(0, omega32); }
  template <bool FAST, bool SKIP = false, bool REGULAR_W = false>
  MLD float step(float cps, float w)
  {
    return phasor_to_pulse<FAST, SKIP, false, REGULAR_W>(phasor_next(omega32, cps), cps, w);
  }
  MLD float next(float cps) { return step<false, true>(cps, width); }
  MLD float next_fast(float cps) { return step<true>(cps, width); }
  // graph form: pulse width as an audio-rate input, PulseGen::operator()(freq, width) MLDSPGens.h:390
  MLD float next2(float cps, float w) { return step<false, true>(cps, w); }
  // launch-constant frequency AND width: `odd` (wave-uniform, tested once by the caller) covers both - some lane's frequency is
  // outside the fast division's ranges or some lane's width is outside pulse_width_is_odd's
  MLD float next_u(float cps, bool odd) { return odd ? step<false>(cps, width) : step<true, false, true>(cps, width); }
  MLD float next_u(float cps, float w, bool odd) { return odd ? step<false>(cps, w) : step<true, false, true>(cps, w); }
  // launch-constant frequency, the width a signal: `odd` is about the frequency alone
  MLD float next_uw(float cps, float w, bool odd) { return odd ? step<false>(cps, w) : step<true>(cps, w); }
  // the frequency a signal, the width launch-constant: `oddW` (wave-uniform, tested once by the caller) is about the width alone
  MLD float next_sw(float cps, bool oddW) { return oddW ? step<false, true>(cps, width) : step<false, true, true>(cps, width); }
  MLD float next_sw(float cps, float w, bool oddW) { return oddW ? step<false, true>(cps, w) : step<false, true, true>(cps, w); }
  static MLD bool input_is_odd(float cps) { return blep_freq_is_odd(cps); }
  // N samples at once, launch-constant frequency and width (see SawGen::trip_u; `odd` covers the width's range as in next_u, so
  // here 0 <= w <= 1 and the shifted phase is v_fract's ground). Four zones: after / before the rising step (the phase itself),
  // after / before the falling step (the phase shifted by 1 - w); (pulse + cUp) - cDown with the idle corrections exactly 0.f is
  // the reference's own expression (phasor_to_pulse above).
  template <int N>
  MLD void trip_u(float cps, float w, bool odd, bool dense, float (&out)[N])
  {
    if (!dense)
    {
      // the phase times 2^31 (phasor_next_hi; 0 <= w <= 1 here, so w * 2^31 is exact); the shifted phase is a phase proper
      const BlepFreq<true> f = BlepFreq<true>::make(cps);
      const float dtS = f.dt * kPhaseScale, omdtS = f.omdt * kPhaseScale, wS = w * kPhaseScale;
      const uint32_t before = omega32;
      float h[N], d[N];
      uint32_t lo = 0u, hi = 0u, dlo = 0u, dhi = 0u;
#pragma unroll
      for (int i = 0; i < N; ++i)
      {
        h[i] = phasor_next_hi(omega32, cps);
        d[i] = __builtin_amdgcn_fractf(__builtin_fmaf(h[i], kPhaseUnit, -w) + 1.0f);
        lo = i ? trip_umin(lo, f2u(h[i])) : f2u(h[i]);
        hi = i ? trip_umax(hi, f2u(h[i])) : f2u(h[i]);
        dlo = i ? trip_umin(dlo, f2u(d[i])) : f2u(d[i]);
        dhi = i ? trip_umax(dhi, f2u(d[i])) : f2u(d[i]);
      }
      const bool suspect = u2f(lo) < kTripTiny * kPhaseScale || u2f(hi) > kTripNearOne * kPhaseScale || u2f(dlo) < kTripTinyShifted || u2f(dhi) > kTripNearOneShifted;
      if (__builtin_amdgcn_ballot_w64(suspect) == 0)
      {
        const float cUpLo = f.correction(u2f(lo) * kPhaseUnit, true, false), cUpHi = f.correction(u2f(hi) * kPhaseUnit, false, false);
        const float cDownLo = f.correction(u2f(dlo), true, false), cDownHi = f.correction(u2f(dhi), false, false);
#pragma unroll
        for (int i = 0; i < N; ++i)
        {
          const float pulse = (h[i] >= wS) ? -1.f : 1.f;
          const float cUp = (h[i] < dtS) ? cUpLo : ((h[i] > omdtS) ? cUpHi : 0.f);
          const float cDown = f.lo(d[i]) ? cDownLo : (f.hi(d[i]) ? cDownHi : 0.f);
          out[i] = (pulse + cUp) - cDown;
        }
        return;
      }
      omega32 = before;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) out[i] = next_u(cps, w, odd);
  }
  template <int N>
  MLD void trip_u(float cps, bool odd, bool dense, float (&out)[N])
  {
    trip_u<N>(cps, width, odd, dense, out);
  }
  MLD void end_vector() {}
};

template <>
struct Proc<MLGPU_PROC_NOISE_GEN>  // MLDSPGens.h:109-148
{
  static constexpr int NC = 0, NS = 1;
  uint32_t seed;
  MLD void load(const VoiceMem& m, const KernelTables&) { seed = m.s(0); }
  MLD void store(const VoiceMem& m) const { m.set(0, seed); }
  MLD float next(float)
  {
    seed = seed * 0x0019660Du + 0x3C6EF35Fu;
    const uint32_t temp = ((seed >> 9) & 0x007FFFFFu) | 0x3F800000u;
    return __builtin_fmaf(u2f(temp), 2.f, -3.f);  // PARITY: temp is in [1, 2): the product and the difference are both exact (:128)
  }
  MLD void end_vector() {}
};

template <>
struct Proc<MLGPU_PROC_TICK_GEN>  // MLDSPGens.h:24-47
{
  static constexpr int NC = 0, NS = 1;
  float omega;
  MLD void load(const VoiceMem& m, const KernelTables&) { omega = u2f(m.s(0)); }
  MLD void store(const VoiceMem& m) const { m.set(0, f2u(omega)); }
  MLD float next(float cps)
  {
    float y = 0.f;
    omega += cps;
    if (omega > 1.0f)
    {
      omega -= 1.0f;
      y = 1.0f;
    }
    return y;
  }
  MLD void end_vector() {}
};

template <>
struct Proc<MLGPU_PROC_IMPULSE_GEN>  // MLDSPGens.h:53-104; table staged through LDS by the kernel
{
  static constexpr int NC = 0, NS = 2;
  static constexpr int kTableSize = 17;
  float omega;
  int32_t counter;
  const float* table;  // LDS copy of the 17-tap windowed sinc
  MLD void load(const VoiceMem& m, const KernelTables& t)
  {
    omega = u2f(m.s(0));
    counter = (int32_t)m.s(1);
    table = t.impulse_table;
  }
  MLD void store(const VoiceMem& m) const
  {
    m.set(0, f2u(omega));
    m.set(1, (uint32_t)counter);
  }
  MLD float next(float cps)
  {
    float y = 0.f;
    omega += cps;
    if (omega > 1.0f)
    {
      omega -= 1.0f;
      counter = 0;
    }
    if (counter < kTableSize)
    {
      y = table[counter];
      counter++;
    }
    return y;
  }
  MLD void end_vector() {}
};

template <>
struct Proc<MLGPU_PROC_ONE_SHOT_GEN>  // MLDSPGens.h:221-282
{
  static constexpr int NC = 0, NS = 3;
  uint32_t omega32, gate, prev;
  MLD void load(const VoiceMem& m, const KernelTables&)
  {
    omega32 = m.s(0);
    gate = m.s(1);
    prev = m.s(2);
  }
  MLD void store(const VoiceMem& m) const
  {
    m.set(0, omega32);
    m.set(1, gate);
    m.set(2, prev);
  }
  MLD float next(float cps)
  {
    const float steps = cps * kStepsPerCycle;
    const int32_t istep = sse_cvt(steps);
    omega32 += (uint32_t)istep * gate;
    if (omega32 < prev)
    {
      gate = 0;
      omega32 = 0;
    }
    prev = omega32;
    return uint_to_float(omega32) * kCyclesPerStep;
  }
  MLD void end_vector() {}
};

// ---- SVF family (Simper), MLDSPFilters.h:51-442 ------------------------------------------
//
// PARITY: `ic1eq += 2.0f * t1` is evaluated as fma(2, t1, ic1eq): 2*t1 is exact in binary
// floating point while it is finite, so the fused form rounds the same real number once and is
// bit-identical to mul-then-add — and one VALU op cheaper. Likewise `2 * v1 - ic1eq` ==
// fma(2, v1, -ic1eq). The one place the two differ: |t1| > FLT_MAX / 2 (1.7e38) with ic1eq of the
// opposite sign - there 2*t1 alone overflows to inf in the reference while the exact sum is still
// finite (tools/ftz_probe.hip shows the pair 7f7fffff / ff7fffff). A filter in that state is one
// doubling away from inf either way; the public contract (mlgpu.h) names the case.

// MLGPU_SVF_STRICT 1 (kernels generated for an engine in strict mode, mlgpu_engine_set_strict_svf) spends the second
// instruction: `ic + 2 * t` and `2 * v - ic` as the reference writes them, identical in the overflow corner too.
#ifndef MLGPU_SVF_STRICT
#define MLGPU_SVF_STRICT 0
#endif
MLD float svf_acc(float ic, float t)  // ic += 2.0f * t
{
#if MLGPU_SVF_STRICT
  return ic + 2.0f * t;
#else
  return __builtin_fmaf(2.0f, t, ic);
#endif
}
MLD float svf_flip(float v, float ic)  // ic = 2 * v - ic
{
#if MLGPU_SVF_STRICT
  return 2.0f * v - ic;
#else
  return __builtin_fmaf(2.0f, v, -ic);
#endif
}

template <int KIND>
struct SvfCore
{
  static constexpr int NC = (KIND == MLGPU_PROC_HIPASS) ? 4 : 3;
  static constexpr int NS = 2;
  float g0, g1, g2, k;
  float ic1eq, ic2eq;
  MLD void load(const VoiceMem& m, const KernelTables&)
  {
    g0 = m.c(0);
    g1 = m.c(1);
    g2 = m.c(2);
    k = (KIND == MLGPU_PROC_HIPASS) ? m.c(3) : 0.f;
    ic1eq = u2f(m.s(0));
    ic2eq = u2f(m.s(1));
  }
  MLD void store(const VoiceMem& m) const
  {
    m.set(0, f2u(ic1eq));
    m.set(1, f2u(ic2eq));
  }
  MLD float next(float v0)
  {
    const float t0 = v0 - ic2eq;
    const float t1 = g0 * t0 + g1 * ic1eq;
    const float t2 = g2 * t0 + g0 * ic1eq;
    float y;
    if (KIND == MLGPU_PROC_LOPASS)
    {
      y = t2 + ic2eq;  // v2, :128-131
    }
    else if (KIND == MLGPU_PROC_BANDPASS)
    {
      y = t1 + ic1eq;  // v1, :234-237
    }
    else
    {
      const float v1 = t1 + ic1eq;
      const float v2 = t2 + ic2eq;
      y = v0 - k * v1 - v2;  // :189-193
    }
    ic1eq = svf_acc(ic1eq, t1);
    ic2eq = svf_acc(ic2eq, t2);
    return y;
  }
  MLD void end_vector() {}
};
template <>
struct Proc<MLGPU_PROC_LOPASS> : SvfCore<MLGPU_PROC_LOPASS>
{
  // Lopass::operator()(vx, omega, k), MLDSPFilters.h:136-152: coefficients made per sample by
  // makeCoeffsVec (:97-115) — omega clamped to <= 0.5, k to >= 0.01 (SSE min/max), two libm sinf
  // calls and one IEEE division per sample. The stored `coeffs` are not used by this form.
  MLD float next(float v0, float omega, float kq)
  {
    omega = sse_min_bound(omega, 0.5f);
    kq = sse_max_bound(kq, 0.01f);
    const float piOmega = 3.1415926535897932384626433832795f * omega;  // kPi, MLDSPScalarMath.h
    const float twoPiOmega = 2.0f * piOmega;
    // The two sinf in the forms proven for this ground (mldsp_math.hpp), chosen per sample for the whole wavefront with ONE
    // question on the common path - an unsigned range test of 2 pi omega's bit pattern (negative and NaN patterns are above every
    // positive float's):
    //   every lane has 2^-11 <= 2 pi omega < kSinfT1 (omega below 0.125: 6 kHz at 48 kHz) -> quadrant 0 for both arguments, the
    //     sine polynomial on the arguments themselves;
    //   every lane has pi omega >= 2^-12 (omega <= 0.5 by the clamp) -> quadrants by comparison, both polynomials, selects;
    //   otherwise (a negative, tiny or NaN omega somewhere) glibc's algorithm as it stands, for all lanes.
    // With a regular resonance too - k <= 1.98, so that 2 + k sin lies in [0.02, 3.98] - the division needs no range handling
    // (div_nr, exact there); any other k sends the wavefront to the last form with the IEEE division.
    const bool regularK = (kq <= 1.98f);
    const uint32_t kOut = regularK ? 0u : 0x80000000u;  // (an irregular k puts the lane's word out of the range below: one compare, one branch)
    const bool q0 = (((f2u(twoPiOmega) - f2u(2.f * kSinfMin))) | kOut) < (f2u(kSinfT1) - f2u(2.f * kSinfMin));
    float s1, s2, nrm;
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!q0) == 0, 1))
    {
      s1 = libm_sinf_q0(piOmega);
      s2 = libm_sinf_q0(twoPiOmega);
      nrm = div_nr(1.0f, 2.f + kq * s2);
    }
    else if (__builtin_amdgcn_ballot_w64(!((piOmega >= kSinfMin) && regularK)) == 0)
    {
      libm_sinf_pair(piOmega, twoPiOmega, s1, s2);
      nrm = div_nr(1.0f, 2.f + kq * s2);
    }
    else
    {
      float s[2];
#pragma unroll 1
      for (int i = 0; i < 2; ++i) s[i] = libm_sinf(i ? twoPiOmega : piOmega);  // (rolled: one copy of the long general code)
      s1 = s[0];
      s2 = s[1];
      nrm = 1.0f / (2.f + kq * s2);
    }
    const float c0 = s2 * nrm;
    const float c1 = (-2.f * s1 * s1 - kq * s2) * nrm;
    const float c2 = (2.0f * s1 * s1) * nrm;
    const float t0 = v0 - ic2eq;
    const float t1 = c0 * t0 + c1 * ic1eq;
    const float t2 = c2 * t0 + c0 * ic1eq;
    const float y = t2 + ic2eq;
    ic1eq = svf_acc(ic1eq, t1);
    ic2eq = svf_acc(ic2eq, t2);
    return y;
  }
  using SvfCore<MLGPU_PROC_LOPASS>::next;
};
template <>
struct Proc<MLGPU_PROC_HIPASS> : SvfCore<MLGPU_PROC_HIPASS>
{
};
template <>
struct Proc<MLGPU_PROC_BANDPASS> : SvfCore<MLGPU_PROC_BANDPASS>
{
};

template <int KIND>
struct ShelfCore  // LoShelf :288-302, HiShelf :369-383, Bell :427-441
{
  static constexpr int NC = (KIND == MLGPU_PROC_LO_SHELF) ? 5 : (KIND == MLGPU_PROC_HI_SHELF ? 6 : 4);
  static constexpr int NS = 2;
  float a1, a2, a3, m0, m1, m2;
  float ic1eq, ic2eq;
  MLD void load(const VoiceMem& m, const KernelTables&)
  {
    a1 = m.c(0);
    a2 = m.c(1);
    a3 = m.c(2);
    m0 = m1 = m2 = 0.f;
    if (KIND == MLGPU_PROC_LO_SHELF)
    {
      m1 = m.c(3);
      m2 = m.c(4);
    }
    else if (KIND == MLGPU_PROC_HI_SHELF)
    {
      m0 = m.c(3);
      m1 = m.c(4);
      m2 = m.c(5);
    }
    else
    {
      m1 = m.c(3);
    }
    ic1eq = u2f(m.s(0));
    ic2eq = u2f(m.s(1));
  }
  MLD void store(const VoiceMem& m) const
  {
    m.set(0, f2u(ic1eq));
    m.set(1, f2u(ic2eq));
  }
  MLD float next(float v0)
  {
    const float v3 = v0 - ic2eq;
    const float v1 = a1 * ic1eq + a2 * v3;
    const float v2 = ic2eq + a2 * ic1eq + a3 * v3;
    ic1eq = svf_flip(v1, ic1eq);
    ic2eq = svf_flip(v2, ic2eq);
    if (KIND == MLGPU_PROC_LO_SHELF) return v0 + m1 * v1 + m2 * v2;
    if (KIND == MLGPU_PROC_HI_SHELF) return m0 * v0 + m1 * v1 + m2 * v2;
    return v0 + m1 * v1;
  }
  // operator()(vx, vcoeffs): one coefficient SIGNAL per coefficient (rows of the reference's
  // DSPVectorArray<COEFFS_SIZE>, e.g. from interpolateCoeffsLinear), LoShelf :304-318, HiShelf :385-399
  MLD float next(float v0, float ca1, float ca2, float ca3, float cm1, float cm2)  // LoShelf {a1,a2,a3,m1,m2}
  {
    const float v3 = v0 - ic2eq;
    const float v1 = ca1 * ic1eq + ca2 * v3;
    const float v2 = ic2eq + ca2 * ic1eq + ca3 * v3;
    ic1eq = svf_flip(v1, ic1eq);
    ic2eq = svf_flip(v2, ic2eq);
    return v0 + cm1 * v1 + cm2 * v2;
  }
  MLD float next(float v0, float ca1, float ca2, float ca3, float cm0, float cm1, float cm2)  // HiShelf {a1,a2,a3,m0,m1,m2}
  {
    const float v3 = v0 - ic2eq;
    const float v1 = ca1 * ic1eq + ca2 * v3;
    const float v2 = ic2eq + ca2 * ic1eq + ca3 * v3;
    ic1eq = svf_flip(v1, ic1eq);
    ic2eq = svf_flip(v2, ic2eq);
    return cm0 * v0 + cm1 * v1 + cm2 * v2;
  }
  MLD void end_vector() {}
};
template <>
struct Proc<MLGPU_PROC_LO_SHELF> : ShelfCore<MLGPU_PROC_LO_SHELF>
{
};
template <>
struct Proc<MLGPU_PROC_HI_SHELF> : ShelfCore<MLGPU_PROC_HI_SHELF>
{
};
template <>
struct Proc<MLGPU_PROC_BELL> : ShelfCore<MLGPU_PROC_BELL>
{
};

// ---- one-state recurrences, MLDSPFilters.h:446-653 ----------------------------------------

template <>
struct Proc<MLGPU_PROC_ONE_POLE>  // :446-481
{
  static constexpr int NC = 2, NS = 1;
  float a0, b1, y1;
  MLD void load(const VoiceMem& m, const KernelTables&)
  {
    a0 = m.c(0);
    b1 = m.c(1);
    y1 = u2f(m.s(0));
  }
  MLD void store(const VoiceMem& m) const { m.set(0, f2u(y1)); }
  MLD float next(float x)
  {
    y1 = a0 * x + b1 * y1;
    return y1;
  }
  MLD void end_vector() {}
};

template <>
struct Proc<MLGPU_PROC_DC_BLOCKER>  // :489-513
{
  static constexpr int NC = 1, NS = 2;
  float c, x1, y1;
  MLD void load(const VoiceMem& m, const KernelTables&)
  {
    c = m.c(0);
    x1 = u2f(m.s(0));
    y1 = u2f(m.s(1));
  }
  MLD void store(const VoiceMem& m) const
  {
    m.set(0, f2u(x1));
    m.set(1, f2u(y1));
  }
  MLD float next(float x0)
  {
    const float y0 = x0 - x1 + c * y1;
    y1 = y0;
    x1 = x0;
    return y0;
  }
  MLD void end_vector() {}
};

template <>
struct Proc<MLGPU_PROC_DIFFERENTIATOR>  // :517-535
{
  static constexpr int NC = 0, NS = 1;
  float x1;
  MLD void load(const VoiceMem& m, const KernelTables&) { x1 = u2f(m.s(0)); }
  MLD void store(const VoiceMem& m) const { m.set(0, f2u(x1)); }
  MLD float next(float x)
  {
    const float y = x - x1;
    x1 = x;
    return y;
  }
  MLD void end_vector() {}
};

template <>
struct Proc<MLGPU_PROC_INTEGRATOR>  // :539-558
{
  static constexpr int NC = 1, NS = 1;
  float leak, y1;
  MLD void load(const VoiceMem& m, const KernelTables&)
  {
    leak = m.c(0);
    y1 = u2f(m.s(0));
  }
  MLD void store(const VoiceMem& m) const { m.set(0, f2u(y1)); }
  MLD float next(float x)
  {
    y1 -= y1 * leak;
    y1 += x;
    return y1;
  }
  MLD void end_vector() {}
};

template <>
struct Proc<MLGPU_PROC_PEAK>  // :562-615; counter decrements once per DSPVector (:607-610)
{
  static constexpr int NC = 3, NS = 2;
  float a0, b1, y1;
  int32_t hold, counter;
  MLD void load(const VoiceMem& m, const KernelTables&)
  {
    a0 = m.c(0);
    b1 = m.c(1);
    hold = (int32_t)f2u(m.c(2));
    y1 = u2f(m.s(0));
    counter = (int32_t)m.s(1);
  }
  MLD void store(const VoiceMem& m) const
  {
    m.set(0, f2u(y1));
    m.set(1, (uint32_t)counter);
  }
  MLD float next(float x)
  {
    const float xsq = x * x;
    if (xsq > y1)
    {
      y1 = xsq;
      counter = hold;
    }
    else if (counter <= 0)
    {
      y1 = a0 * xsq + b1 * y1;
    }
    return (y1 > 1e-20f) ? sqrt_approx(y1) : 0.f;
  }
  MLD void end_vector()
  {
    if (counter > 0) counter -= MLGPU_FLOATS_PER_DSPVECTOR;
  }
};

template <>
struct Proc<MLGPU_PROC_RMS>  // :619-653
{
  static constexpr int NC = 2, NS = 1;
  float a0, b1, y1;
  MLD void load(const VoiceMem& m, const KernelTables&)
  {
    a0 = m.c(0);
    b1 = m.c(1);
    y1 = u2f(m.s(0));
  }
  MLD void store(const VoiceMem& m) const { m.set(0, f2u(y1)); }
  MLD float next(float x)
  {
    const float xsq = x * x;
    y1 = a0 * xsq + b1 * y1;
    return (y1 > 1e-20f) ? sqrt_approx(y1) : 0.f;
  }
  MLD void end_vector() {}
};

template <>
struct Proc<MLGPU_PROC_ADSR>  // :657-797
{
  static constexpr int NC = 4, NS = 8;
  enum { A = 0, D = 1, S = 2, R = 3, off = 4 };
  float ka, kd, s, kr;
  float y, y1, x1, threshold, target, k, amp;
  int32_t segment;
  MLD void load(const VoiceMem& m, const KernelTables&)
  {
    ka = m.c(0);
    kd = m.c(1);
    s = m.c(2);
    kr = m.c(3);
    y = u2f(m.s(0));
    y1 = u2f(m.s(1));
    x1 = u2f(m.s(2));
    threshold = u2f(m.s(3));
    target = u2f(m.s(4));
    k = u2f(m.s(5));
    amp = u2f(m.s(6));
    segment = (int32_t)m.s(7);
    derive_masks();
  }
  MLD void store(const VoiceMem& m) const
  {
    m.set(0, f2u(y));
    m.set(1, f2u(y1));
    m.set(2, f2u(x1));
    m.set(3, f2u(threshold));
    m.set(4, f2u(target));
    m.set(5, f2u(k));
    m.set(6, f2u(amp));
    m.set(7, (uint32_t)segment);
  }
  // processSample (:704-786) is a little state machine: most of its code runs only on the sample where a voice changes
  // segment (gate on, gate off, the envelope crossing its segment's end). Evaluated as the reference writes it, every sample
  // of every wavefront walks those divergent branches and seven compares: about 27 ns per wavefront-sample, the second most
  // expensive node of the synth voice (tools/node_costs.py: 12.9 now). Here one wave-uniform test decides whether ANY lane might change segment
  // at this sample, from lane masks carried across samples (ballots; scalar instructions cost as much as vector ones on this
  // chip - tools/instbench.hip - so the test is two compares and a handful of scalar operations):
  //   gate on / gate off need x == 0 to differ from last sample's  (a superset of the reference's two edge tests),
  //   a crossing needs y > threshold to have changed at the last update (its test of y1 and y, made one sample early).
  // If it might, the reference's tests run first for every lane (change_segment) and re-parameterize the one-pole. Either
  // way the sample ends with `x1 = x; y1 = y; y += k (target - y); return y amp`, selected away on the idle lanes (segment
  // off and x == 0 on entry: the reference returns 0 and touches nothing). Same values, same state, bit for bit.
  uint64_t mXZero, mAbove, mCrossed, mOff;
  MLD void derive_masks()
  {
    mOff = __builtin_amdgcn_ballot_w64(segment == off);
    mXZero = __builtin_amdgcn_ballot_w64(x1 == 0.f);
    mAbove = __builtin_amdgcn_ballot_w64(y > threshold);
    mCrossed = __builtin_amdgcn_ballot_w64(y1 > threshold) ^ mAbove;
  }
  MLD float next(float x)
  {
    const uint64_t xz = __builtin_amdgcn_ballot_w64(x == 0.f);
    uint64_t maybe = (xz ^ mXZero) | mCrossed;
    // (kept as mask arithmetic: left alone, the compiler rewrites the test as two 64-bit equality compares, two selects and two
    // ands - nine scalar instructions per sample where these are six, and scalar issue is not hidden on this chip)
    asm("" : "+s"(maybe));
    const uint64_t idle = mOff & xz;
#ifndef MLGPU_X_ADSR_NO_SEGMENT_TEST  // (elimination experiment: wrong results, profiles/r06_synth_scalar_ceiling.txt)
    if (__builtin_expect(maybe != 0, 0)) change_segment(x);
#endif
    const float yn = y + k * (target - y);
    x1 = lane_select(idle, x1, x);
    y1 = lane_select(idle, y1, y);
    y = lane_select(idle, y, yn);
    // (an idle lane keeps y: its bit of `above` stays, its bit of mCrossed clears - a crossing only counts before off)
    const uint64_t above = __builtin_amdgcn_ballot_w64(y > threshold);
    mXZero = xz;
    mCrossed = mAbove ^ above;
    asm("" : "+s"(mCrossed));
    mAbove = above;
    return lane_select(idle, 0.f, yn * amp);
  }
  // processSample :711-775 for the lanes that are not idle at this sample: crossing / gate on / gate off -> new segment,
  // coefficient, target, threshold (and y, y1 where the reference resets them). Written with selects instead of the
  // reference's branches and switch - the same tests in the same order, the same two arithmetic expressions - so that next()
  // holds wave-uniform branches only (a divergent branch in here makes the compiler thread flags and register copies through
  // the quiet path). Rare; the whole wavefront walks it.
  MLD void change_segment(float x)
  {
    const bool active = !((segment == off) && (x == 0.f));
    const bool crossed = active && ((y1 > threshold) != (y > threshold)) && (segment < off);
    const bool trigOn = active && (x1 == 0.f) && (x > 0.f);
    const bool trigOff = active && !trigOn && (x1 > 0.f) && (x == 0.f);
    const bool recalc = crossed || trigOn || trigOff;
    int32_t seg = segment + (crossed ? 1 : 0);
    seg = trigOn ? (int32_t)A : (trigOff ? (int32_t)R : seg);
    amp = trigOn ? x : amp;
    const bool isA = (seg == A), isD = (seg == D), isS = (seg == S), isR = (seg == R);
    const float startEnv = isA ? 0.f : (isD ? 1.f : ((isS || isR) ? s : 0.f));
    const float endEnv = isA ? 1.f : ((isD || isS) ? s : 0.f);
    const float kNew = isA ? ka : (isD ? kd : (isR ? kr : 0.f));
    const bool resetY = recalc && !(isA || isD || isR);  // S, and the default case (off)
    const float yNew = isS ? s : 0.f;
    const float segmentBias = (endEnv - startEnv) * 0.1f;
    const float targetNew = endEnv + segmentBias;
    k = recalc ? kNew : k;
    threshold = recalc ? endEnv : threshold;
    target = recalc ? targetNew : target;
    y1 = resetY ? yNew : y1;
    y = resetY ? yNew : y;
    segment = seg;
    // (a lane that has just gone off is not idle at this sample: next() took its idle mask before)
    mOff = __builtin_amdgcn_ballot_w64(segment == off);
    mAbove = __builtin_amdgcn_ballot_w64(y > threshold);
  }
  MLD void end_vector() {}
};

template <>
struct Proc<MLGPU_PROC_GAIN>  // x * DSPVector(gain), MLDSPOps.h:157,345-348
{
  static constexpr int NC = 1, NS = 0;
  float gain;
  MLD void load(const VoiceMem& m, const KernelTables&) { gain = m.c(0); }
  MLD void store(const VoiceMem&) const {}
  MLD float next(float x) { return x * gain; }
  MLD void end_vector() {}
};

// ---- control-rate -> audio-rate ramps, MLDSPGens.h:404-590 -----------------------------------------
//
// Interpolator1 and LinearGlide take ONE float per DSPVector (`operator()(float f)`): they are
// vector-rate processors. The graph kernel calls begin_vector(f) once per DSPVector and next_n(n) for
// sample n of it. They exist in graphs only (a bank's input is audio-rate).
// kUnityRampVec[n] = (n + 1) / 64.f (:409-410) is exact in binary floating point.

template <>
struct Proc<MLGPU_PROC_INTERPOLATOR1>  // :412-423
{
  static constexpr int NC = 0, NS = 1;
  static constexpr bool kVectorRate = true;
  float cur, base, dydt;
  MLD void load(const VoiceMem& m, const KernelTables&) { cur = u2f(m.s(0)); base = cur; dydt = 0.f; }
  MLD void store(const VoiceMem& m) const { m.set(0, f2u(cur)); }
  MLD void begin_vector(float f)
  {
    dydt = f - cur;
    base = cur;
    cur = f;
  }
  MLD float next_n(int n) const { return base + ((float)(n + 1) * 0.015625f) * dydt; }
  MLD void end_vector() {}
};

template <>
struct Proc<MLGPU_PROC_LINEAR_GLIDE>  // :433-515
{
  // C{vectorsPerGlide:i32, dyPerVector}  S{target, step, vectorsRemaining:i32, currVec[64]}
  // mCurrVec is a 64-float member: it stays in the SoA state array in HBM (slot 3 + n, coalesced across the
  // wavefront) and is read / updated in place at sample n; mStepVec is always a broadcast, one float.
  static constexpr int NC = 2, NS = 3 + MLGPU_FLOATS_PER_DSPVECTOR;
  static constexpr bool kVectorRate = true;
  enum { kHold = 0, kEnd = 1, kStart = 2, kContinue = 3 };
  int32_t perGlide, remaining;
  float dyPerVector, target, step, startValue;
  int mode;
  VoiceMem mem;
  MLD void load(const VoiceMem& m, const KernelTables&)
  {
    mem = m;
    perGlide = (int32_t)f2u(m.c(0));
    dyPerVector = m.c(1);
    target = u2f(m.s(0));
    step = u2f(m.s(1));
    remaining = (int32_t)m.s(2);
    mode = kHold;
    startValue = 0.f;
  }
  MLD void store(const VoiceMem& m) const
  {
    m.set(0, f2u(target));
    m.set(1, f2u(step));
    m.set(2, (uint32_t)remaining);
  }
  MLD void begin_vector(float f)
  {
    if (f != target)
    {
      target = f;
      remaining = perGlide;
    }
    if (remaining < 0)
    {
      mode = kHold;
    }
    else if (remaining == 0)
    {
      mode = kEnd;
      step = 0.f;
      remaining--;
    }
    else if (remaining == perGlide)
    {
      mode = kStart;
      startValue = u2f(mem.s(3 + MLGPU_FLOATS_PER_DSPVECTOR - 1));
      step = (target - startValue) * dyPerVector;
      remaining--;
    }
    else
    {
      mode = kContinue;
      remaining--;
    }
  }
  // mCurrVec slot n is read and rewritten at sample n only: the graph kernel fetches a quad's four slots together
  // (begin_quad) instead of one load per sample in the middle of the sample loop's stores
  float cur[4];
  MLD void begin_quad(int q)
  {
    if (mode == kHold || mode == kContinue)
    {
#pragma unroll
      for (int k = 0; k < 4; ++k) cur[k] = u2f(mem.s(3 + 4 * q + k));
    }
  }
  MLD float next_n(int n) const
  {
    float c;
    if (mode == kHold) return cur[n & 3];
    if (mode == kEnd)
      c = target;
    else if (mode == kStart)
      c = startValue + ((float)(n + 1) * 0.015625f) * step;
    else
      c = cur[n & 3] + step;
    mem.set(3 + n, f2u(c));
    return c;
  }
  MLD void end_vector() {}
};

template <>
struct Proc<MLGPU_PROC_SAMPLE_ACCURATE_LINEAR_GLIDE>  // :517-590, nextSample(f) once per sample
{
  // C{samplesPerGlide:i32, dyPerSample}  S{curr, step, target, samplesRemaining:i32}
  static constexpr int NC = 2, NS = 4;
  int32_t perGlide, remaining;
  float dyPerSample, curr, step, target;
  MLD void load(const VoiceMem& m, const KernelTables&)
  {
    perGlide = (int32_t)f2u(m.c(0));
    dyPerSample = m.c(1);
    curr = u2f(m.s(0));
    step = u2f(m.s(1));
    target = u2f(m.s(2));
    remaining = (int32_t)m.s(3);
  }
  MLD void store(const VoiceMem& m) const
  {
    m.set(0, f2u(curr));
    m.set(1, f2u(step));
    m.set(2, f2u(target));
    m.set(3, (uint32_t)remaining);
  }
  MLD float next(float f)
  {
    if (f != target)
    {
      target = f;
      remaining = perGlide;
    }
    if (remaining < 0)
    {
    }
    else if (remaining == 0)
    {
      curr = target;
      step = 0.f;
      remaining--;
    }
    else if (remaining == perGlide)
    {
      step = (target - curr) * dyPerSample;
      remaining--;
    }
    else
    {
      curr += step;
      remaining--;
    }
    return curr;
  }
  MLD void end_vector() {}
};

// ---- delay lines, MLDSPFilters.h:799-1106 -----------------------------------------------------------------
//
// Per-voice rings live in HBM (VoiceMem::mem). Every form is evaluated per sample as the reference's processSample
// (:899-914): write x at the write index, read at (write - delay) & mask, advance. For delays inside the ring's
// valid range 0 <= d <= length - 64 that is identical to the block form of operator()(vx) (:834-875), which writes
// the whole vector before reading it: a read can then never land on a sample written later in the same vector.

constexpr int kRingWindow = 8;       // floats per LDS window = one 32-byte HBM sector
constexpr int kRingLdsLanes = 256;   // lanes per workgroup of a graph kernel
#ifndef MLGPU_RING_WINDOWS
#define MLGPU_RING_WINDOWS 0
#endif
constexpr bool kRingWindows = MLGPU_RING_WINDOWS != 0;  // fixed per generated kernel, so the unused form costs nothing
// Layout 2 (round 5, "transposed windows"): the ring of a 256-voice block as [chunk = sample / 16][lane][16], every global access
// a whole 64-byte piece per voice and made by FOUR NEIGHBOURING LANES (16 bytes each) - a wavefront's memory instruction then
// covers 16 voices' pieces, 8 full cache lines when the voices read the same chunk, and no lane ever moves a byte it does not use.
// All of it on a wave-uniform clock: every 16 samples (the write index is the same in every lane of a wavefront - where it is not,
// the launch falls back to plain per-sample accesses) the write window is stored, the chunk each voice asked for 16 samples ago goes
// from the loaders' registers into that voice's read window, and the chunk it will need 16 samples from now is requested - a whole
// period ahead of its first use, so no sample waits for memory (layout 1 refilled lane by lane whenever a lane crossed a sector:
// a divergent load and a memory round trip on almost every sample of a wavefront whose voices have different delay times, and its
// read window was eight registers behind a seven-select chain). Windows live in LDS, one strip per wavefront and ring:
// rows 0..7 the eight samples being written (a chunk's first eight parked in registers until its second eight are there: what
// goes to memory is whole 64-byte pieces), rows 8..39 the 32 ring positions being read - two prefetched chunks, or, for a lane
// that reads closer than three chunks behind the writer, its own last 32 samples - 64 floats per
// row (lane = column: an access with the lane as the fast index touches every bank twice, whatever the row). 10 KiB per wavefront:
// a CU holds its sixteen wavefronts of a 262 144-voice bank at once (with 12.5 KiB - a 16-sample write window - three of four
// workgroups fit and the fourth runs alone afterwards: twice the launch time, measured).
constexpr bool kRingTransposed = MLGPU_RING_WINDOWS == 2;
// Layout 4 (round 6, "sector trips"; MLGPU_RING_WINDOWS 3): layout 1's memory - [256-voice block][sample / 8][lane][8], a voice's 32-byte
// sectors - with NO LDS and no per-sample decisions, for graphs with many rings. The sample loop runs in trips of 8 samples on the
// wave-uniform write clock (w a multiple of 8 at every trip's start). Per ring and trip: the eight written samples stay in registers
// and go out as ONE sector (two 16-byte stores); the eight samples read are positions r0 .. r0 + 7 of two neighbouring sectors - the
// lower one HELD in eight registers since the previous trip, the upper one loaded in the TRIP'S PROLOGUE, where the generated kernel
// issues the loads of ALL its rings together before any arithmetic (one memory round trip per trip and wavefront, not one per ring
// and sample) - and a three-stage barrel shifter (28 selects per trip, 3.5 per sample) picks them by r0 & 7. The prologue PREDICTS
// the read position from the delay time in effect (delay times move rarely: a PitchbendableDelay's at most every 32 samples); a trip
// whose first sample finds another sector, and every sample whose delay time differs from the trip's first, loads again (waited).
// Delay times under 16 samples - a read that could land in a sector not yet in memory - send that trip of that wavefront through
// plain per-sample accesses of the same memory. Layout 1 decided per lane and sample whether to refill its register window: a
// divergent load and a round trip on almost every sample of a wavefront whose voices have different delay times.
constexpr bool kRingSectors = MLGPU_RING_WINDOWS == 3;
constexpr int32_t kSectorMinDelay = 16;
constexpr int kTChunk = 16, kTWrite = 8, kTRowPad = 64, kTRows = kTWrite + 2 * kTChunk, kTStrip = kTRows * kTRowPad;
constexpr uint32_t kTNone = 0xFFFFFFFFu;

struct RingCore  // IntegerDelay's buffer, index and mask
{
  uint32_t w;
  uint32_t rChunk;  // windowed layout: the 8-sample chunk the read window holds (0xFFFFFFFF: none)
  float rw[8];      // windowed layout: the read window
  typedef float f32x4r __attribute__((ext_vector_type(4)));

  // ---- windowed layout: 32-byte sectors per voice, staged through LDS windows ----
  // When neighbouring voices have different delay times, the [sample][V] layout makes every lane pull a whole cache line
  // for 4 bytes (measured 10.6x the algorithmic traffic). Here the ring of a 256-voice block is stored as
  // [chunk = sample / 8][lane][8]: a lane collects 8 written samples in an LDS window and stores them as one 32-byte
  // sector (the block's sectors of one chunk are contiguous: 8 KiB), and reads a sector at a time into a register window,
  // so every byte moved is used whatever the delay times are, and equal delay times still coalesce. Same values as the
  // direct form for delays within the node's maximum: a read is served from the write window, the read window or memory,
  // whichever holds that sample. m.mem points at this lane's sector of chunk 0 of the node's first ring.
  MLD float* wbuf(const VoiceMem& m, int ringIdx) const { return m.lds + (size_t)ringIdx * kRingWindow * kRingLdsLanes; }
  MLD float* chunkMem(const VoiceMem& m, int ringIdx, uint32_t i) const  // the sector holding ring position i
  {
    return m.mem + ((size_t)ringIdx * (m.memMask + 1) + (size_t)(i & ~(uint32_t)(kRingWindow - 1))) * kRingLdsLanes;
  }
  // (histWriter: ring layout 4 - this core writes the node's ring and keeps its last 16 samples in LDS; a PitchbendableDelay's
  // second core does not)
  MLD void begin(const VoiceMem& m, int ringIdx, bool histWriter = true)  // launch start: the chunk being written comes back into its window
  {
    rChunk = 0xFFFFFFFFu;
    if (kRingTransposed)
    {
      beginT(m, ringIdx);
      return;
    }
    if (kRingSectors)
    {
      beginS(m, ringIdx, histWriter);
      return;
    }
    if (!kRingWindows) return;
    const f32x4r* src = (const f32x4r*)chunkMem(m, ringIdx, w);
    float* wb = wbuf(m, ringIdx);
    const f32x4r a = src[0], b = src[1];
    wb[0 * kRingLdsLanes] = a[0]; wb[1 * kRingLdsLanes] = a[1]; wb[2 * kRingLdsLanes] = a[2]; wb[3 * kRingLdsLanes] = a[3];
    wb[4 * kRingLdsLanes] = b[0]; wb[5 * kRingLdsLanes] = b[1]; wb[6 * kRingLdsLanes] = b[2]; wb[7 * kRingLdsLanes] = b[3];
  }
  MLD void flush(const VoiceMem& m, int ringIdx, uint32_t chunkStart) const
  {
    const float* wb = wbuf(m, ringIdx);
    f32x4r* dst = (f32x4r*)chunkMem(m, ringIdx, chunkStart);
    const f32x4r a = {wb[0 * kRingLdsLanes], wb[1 * kRingLdsLanes], wb[2 * kRingLdsLanes], wb[3 * kRingLdsLanes]};
    const f32x4r b = {wb[4 * kRingLdsLanes], wb[5 * kRingLdsLanes], wb[6 * kRingLdsLanes], wb[7 * kRingLdsLanes]};
    dst[0] = a;
    dst[1] = b;
  }
  MLD void end(const VoiceMem& m, int ringIdx) const  // launch end: the partly filled write window goes back to memory
  {
    if (kRingTransposed)
    {
      if (uniformW) flushLast(m, ringIdx);
    }
    else if (kRingSectors)
    {
      // (a launch is whole DSPVectors = whole trips: nothing is pending)
    }
    else if (kRingWindows) flush(m, ringIdx, w & ~(uint32_t)(kRingWindow - 1));
  }
  MLD float sampleWindowed(const VoiceMem& m, int ringIdx, float x, int32_t d)
  {
    float* wb = wbuf(m, ringIdx);
    wb[(w & (kRingWindow - 1)) * kRingLdsLanes] = x;
    const uint32_t r = (w - (uint32_t)d) & m.memMask;
    const uint32_t rc = r >> 3, wc = w >> 3;
    float y;
    if (rc == wc)
      y = wb[(r & (kRingWindow - 1)) * kRingLdsLanes];  // still in the write window (d < 8)
    else
    {
      if (rc != rChunk)
      {
        const f32x4r* src = (const f32x4r*)chunkMem(m, ringIdx, r);
        const f32x4r a = src[0], b = src[1];
        rw[0] = a[0]; rw[1] = a[1]; rw[2] = a[2]; rw[3] = a[3];
        rw[4] = b[0]; rw[5] = b[1]; rw[6] = b[2]; rw[7] = b[3];
        rChunk = rc;
      }
      // the read window lives in registers: a select chain over values (all eight read first, so that no arm of a
      // conditional holds a load the optimizer could turn into an indexed access)
      const uint32_t slot = r & (kRingWindow - 1);
      const float r0 = rw[0], r1 = rw[1], r2 = rw[2], r3 = rw[3], r4 = rw[4], r5 = rw[5], r6 = rw[6], r7 = rw[7];
      y = r0;
      y = (slot == 1) ? r1 : y;
      y = (slot == 2) ? r2 : y;
      y = (slot == 3) ? r3 : y;
      y = (slot == 4) ? r4 : y;
      y = (slot == 5) ? r5 : y;
      y = (slot == 6) ? r6 : y;
      y = (slot == 7) ? r7 : y;
    }
    if ((w & (kRingWindow - 1)) == kRingWindow - 1)
    {
      flush(m, ringIdx, w & ~(uint32_t)(kRingWindow - 1));
      if (rChunk == wc) rChunk = 0xFFFFFFFFu;  // (cannot hold: the read window never holds the chunk being written)
    }
    w = (w + 1) & m.memMask;
    return y;
  }

  // ---- layout 2: transposed windows (see the constants above) ----
  // What the 32 read rows hold: the samples of ring positions [lo, lo + len), len <= 32, position t in row 8 + (t & 31). Whole chunks
  // fetched a period ahead for a lane that reads three or more chunks behind the writer; its own last 32 written samples (copied
  // from the write window at every flush: `hist`) for a lane that reads closer than that - those chunks cannot be asked for a period
  // ahead, they are not written yet, and a quarter of the voices of a bank of plucked strings (55 .. 880 Hz) are such lanes: served
  // from memory sample by sample they cost every wavefront a memory round trip per sample (measured: 1.41 ms per launch against
  // 1.17 with that path compiled out, profiles/r05_ring_layouts.txt).
  uint32_t lo, len;
  bool hist;
  uint32_t pend;        // the chunk this lane asked for at the last boundary: in flight in its four loader lanes' registers
  f32x4r stage[4];      // pieces this lane loaded for OTHER lanes: stage[m] = piece (lane & 3) of lane (m * 16 + lane / 4)'s pending chunk
  bool uniformW, primed;
  MLD float* strip(const VoiceMem& m, int ringIdx) const { return m.lds + (size_t)ringIdx * 4 * kTStrip; }  // this lane's column of row 0
  // lane `other`'s chunk at ring position i of ring ringIdx (m.mem is THIS lane's 16-float piece of chunk 0 of the node's first ring)
  MLD float* chunkOf(const VoiceMem& m, int ringIdx, uint32_t other, uint32_t i) const
  {
    const int32_t rel = (int32_t)other - (int32_t)(threadIdx.x & 63u);
    return m.mem + (ptrdiff_t)rel * kTChunk + ((size_t)ringIdx * (m.memMask + 1) + (size_t)(i & ~(uint32_t)(kTChunk - 1))) * kRingLdsLanes;
  }
  MLD void beginT(const VoiceMem& m, int ringIdx)
  {
    pend = kTNone;
    lo = len = 0;
    hist = false;
    primed = false;
    parked = false;
    park[0] = park[1] = f32x4r{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) stage[i] = f32x4r{0.f, 0.f, 0.f, 0.f};
    const uint32_t w0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)w);
    uniformW = __builtin_amdgcn_ballot_w64(w != w0) == 0;
    if (!uniformW) return;
    // the eight samples being written come back into the write window (all eight positions: end() stores them all)
    const f32x4r* src = (const f32x4r*)(chunkOf(m, ringIdx, threadIdx.x & 63u, w) + (w & (uint32_t)kTWrite));
    float* col = strip(m, ringIdx);
#pragma unroll
    for (int j = 0; j < 2; ++j)
    {
      const f32x4r a = src[j];
#pragma unroll
      for (int e = 0; e < 4; ++e) col[(4 * j + e) * kTRowPad] = a[e];
    }
  }
  // The write window to memory, transposed: lane L handles 16 bytes (L & 1) of the 32-byte half piece of lanes L / 2 and 32 + L / 2;
  // halfStart = the ring position of the window's first sample (a multiple of 8). A chunk's FIRST half is only read out of LDS
  // and parked in eight registers; when its second half is complete both go out in neighbouring instructions, so that what
  // reaches memory is whole 64-byte pieces: half pieces eight samples (10 us) apart are two partial writes of a line the L2 has
  // evicted in between, and cost what whole ones cost (measured: 1.25 -> 1.12 ms per launch of the strings workload with the
  // traffic pattern of this form, profiles/r05_ring_layouts.txt). `last`: the launch ends here, nothing stays parked.
  f32x4r park[2];
  bool parked;  // (the same in every lane)
  MLD void readHalf(const VoiceMem& m, int ringIdx, f32x4r (&a)[2]) const
  {
    const uint32_t lane = threadIdx.x & 63u, j = lane & 1u;
    const float* row0 = strip(m, ringIdx) - lane;  // column 0 of row 0
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int mm = 0; mm < 2; ++mm)
    {
      const float* src = row0 + (4 * j) * kTRowPad + (uint32_t)mm * 32u + (lane >> 1);
      a[mm] = f32x4r{src[0], src[kTRowPad], src[2 * kTRowPad], src[3 * kTRowPad]};
    }
    __builtin_amdgcn_wave_barrier();
  }
  // half `second` (0 / 1) of the chunk at ring position chunkStart
  MLD void storeHalf(const VoiceMem& m, int ringIdx, uint32_t chunkStart, uint32_t second, const f32x4r (&a)[2]) const
  {
    const uint32_t lane = threadIdx.x & 63u, j = lane & 1u;
#pragma unroll
    for (int mm = 0; mm < 2; ++mm)
      *((f32x4r*)(chunkOf(m, ringIdx, (uint32_t)mm * 32u + (lane >> 1), chunkStart) + second * (uint32_t)kTWrite) + j) = a[mm];
  }
  MLD void storeBoth(const VoiceMem& m, int ringIdx, uint32_t chunkStart, const f32x4r (&a0)[2], const f32x4r (&a1)[2]) const
  {
    const uint32_t lane = threadIdx.x & 63u, j = lane & 1u;
#pragma unroll
    for (int mm = 0; mm < 2; ++mm)
    {
      f32x4r* piece = (f32x4r*)chunkOf(m, ringIdx, (uint32_t)mm * 32u + (lane >> 1), chunkStart);
      piece[j] = a0[mm];
      piece[2 + j] = a1[mm];
    }
  }
  MLD void flushT(const VoiceMem& m, int ringIdx, uint32_t halfStart)
  {
#ifdef MLGPU_RING_X_NOFLUSH  // (elimination experiments: wrong results, used to find what a launch waits for - profiles/r05_ring_layouts.txt)
    return;
#endif
    f32x4r a[2];
    readHalf(m, ringIdx, a);
    if ((halfStart & (uint32_t)kTWrite) == 0)
    {
      park[0] = a[0];
      park[1] = a[1];
      parked = true;
      return;
    }
    if (parked) storeBoth(m, ringIdx, halfStart & ~(uint32_t)(kTChunk - 1), park, a);
    else storeHalf(m, ringIdx, halfStart & ~(uint32_t)(kTChunk - 1), 1u, a);
    parked = false;
  }
  // launch end: the partly filled write window (all eight positions: beginT brought the others in) and what is parked
  MLD void flushLast(const VoiceMem& m, int ringIdx) const
  {
    f32x4r a[2];
    readHalf(m, ringIdx, a);
    const uint32_t chunkStart = w & ~(uint32_t)(kTChunk - 1);
    if ((w & (uint32_t)kTWrite) == 0) storeHalf(m, ringIdx, chunkStart, 0u, a);
    else if (parked) storeBoth(m, ringIdx, chunkStart, park, a);
    else storeHalf(m, ringIdx, chunkStart, 1u, a);
  }
  // a read that goes to memory (sampleT's last resort) must find the parked half there
  MLD void unpark(const VoiceMem& m, int ringIdx)
  {
    if (!parked) return;
#ifdef MLGPU_RING_X_NOUNPARK  // (shows that the tests reach this: tests/test_gpu_delays.py fails with it)
    return;
#endif
    storeHalf(m, ringIdx, w & ~(uint32_t)(kTChunk - 1), 0u, park);
    parked = false;
  }
  // one round of transposed loads: every lane names a chunk (kTNone: none); the pieces land in `out` of the four loader lanes.
  // UNCONDITIONAL loads (a lane that names no chunk gets the piece at `spare`, a ring position whose chunk is complete in memory,
  // and nobody uses it): a load under a condition into registers that may still await an earlier load makes the compiler wait for
  // every memory operation in flight before it - a memory round trip per load instead of none.
  MLD void loadRound(const VoiceMem& m, int ringIdx, uint32_t want, uint32_t spare, f32x4r (&out)[4]) const
  {
    const uint32_t lane = threadIdx.x & 63u, j = lane & 3u;
#pragma unroll
    for (int mm = 0; mm < 4; ++mm)
    {
      const uint32_t other = (uint32_t)mm * 16u + (lane >> 2);
      const uint32_t q = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(other * 4u), (int)want);
#ifdef MLGPU_RING_X_NOLOAD
      out[mm] = f32x4r{(float)q, 0.f, 0.f, 0.f};
#else
      out[mm] = *((const f32x4r*)chunkOf(m, ringIdx, other, (q != kTNone ? q : spare) * (uint32_t)kTChunk) + j);
#endif
    }
  }
  // ... and into the owners' read windows: chunk q goes to slot q & 1 (rows 16 + 16 (q & 1) ...)
  MLD void storeRound(const VoiceMem& m, int ringIdx, uint32_t want, const f32x4r (&in)[4]) const
  {
    const uint32_t lane = threadIdx.x & 63u, j = lane & 3u;
    float* row0 = strip(m, ringIdx) - lane;
#pragma unroll
    for (int mm = 0; mm < 4; ++mm)
    {
      const uint32_t other = (uint32_t)mm * 16u + (lane >> 2);
      const uint32_t q = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(other * 4u), (int)want);
      if (q != kTNone)
      {
        float* dst = row0 + (kTWrite + kTChunk * (q & 1u) + 4 * j) * kTRowPad + other;
#pragma unroll
        for (int e = 0; e < 4; ++e) dst[e * kTRowPad] = in[mm][e];
      }
    }
  }
  MLD bool held(const VoiceMem& m, uint32_t t) const { return ((t - lo) & m.memMask) < len; }
  // n samples from ring position t on have just been put into their rows: joined to what is held when they continue it (the rows
  // they overwrote were the oldest 32 back), else they are all that is held
  MLD void installed(const VoiceMem& m, uint32_t t, uint32_t n)
  {
    if (len != 0 && ((lo + len) & m.memMask) == t)
    {
      len += n;
      if (len > 2 * kTChunk)
      {
        lo = (lo + len - 2 * kTChunk) & m.memMask;
        len = 2 * kTChunk;
      }
    }
    else
    {
      lo = t;
      len = n;
    }
  }
  // the write window's eight samples (ring positions t .. t + 7, t = the block just finished) into the read rows of the lanes that
  // read close behind the writer
  MLD void keepHistory(const VoiceMem& m, int ringIdx, uint32_t t)
  {
    if (__builtin_amdgcn_ballot_w64(hist) == 0) return;
    if (hist)
    {
      float* col = strip(m, ringIdx);
      float* dst = col + (kTWrite + (t & (2 * kTChunk - 1))) * kTRowPad;
      float v[kTWrite];
#pragma unroll
      for (int j = 0; j < kTWrite; ++j) v[j] = col[j * kTRowPad];
#pragma unroll
      for (int j = 0; j < kTWrite; ++j) dst[j * kTRowPad] = v[j];
      installed(m, t, kTWrite);
    }
  }
  static constexpr int kWaitVm0 = 0x0F70;  // s_waitcnt vmcnt(0), the other counters left alone (gfx9 encoding)
  // every 16 samples, all lanes together (w is the same in all of them and a multiple of 16, or this is the launch's first sample)
  MLD void boundaryT(const VoiceMem& m, int ringIdx, int32_t d)
  {
#ifdef MLGPU_RING_X_NOBOUNDARY
    primed = true;
    return;
#endif
    const uint32_t cmask = ((m.memMask + 1) >> 4) - 1u, wc = w >> 4;
    const uint32_t c = ((w - (uint32_t)d) & m.memMask) >> 4;
    const uint32_t age = (wc - c) & cmask;  // how many chunks behind the writer this lane reads (0: inside the write window)
    // (no cache maintenance: a voice's ring is written and read by one wavefront only, and a CU's vector cache is coherent for its
    // own wavefronts' stores - an acquire fence here would also wait for every store and load in flight, a memory round trip per
    // 16 samples: measured 2 x the launch time)
    // what the coming 16 samples overwrite in memory is a ring cycle old in the rows from now on
    if (((w - lo) & m.memMask) < len || ((lo - w) & m.memMask) < (uint32_t)kTChunk) len = 0;
    // a lane that reads less than three chunks behind the writer keeps its own history instead of asking (and what it may have
    // asked for while it read further back is dropped: it would wait in the loaders' registers for ever and hold up the wavefront)
    hist = age < 3u;
    if (hist) pend = kTNone;
    // what was asked for at the last boundary goes into its rows - unless those still hold a chunk the coming 16 samples read
    // (the read position moved by less than a chunk: the launch's first boundary came mid-chunk, a delay time grew): then it
    // waits in the loaders' registers for another period
    const uint32_t c1 = (c + 1u) & cmask;
    const uint32_t shares = ((c ^ pend) & 1u) ? c1 : c;  // the chunk of the coming reads that lives in the rows pend would take
    const bool inUse = shares != pend && (held(m, shares * kTChunk) || held(m, shares * kTChunk + kTChunk - 1));
    const uint32_t commit = (pend != kTNone && !inUse) ? pend : kTNone;
    if (__builtin_amdgcn_ballot_w64(commit != kTNone) != 0)
    {
      storeRound(m, ringIdx, commit, stage);
      if (commit != kTNone)
      {
        installed(m, commit * kTChunk, kTChunk);
        pend = kTNone;
      }
    }
    if (!primed)
    {
      // the launch's first sample: the two chunks the coming 16 samples read, those of them that are complete in memory
      primed = true;
      f32x4r tmp[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
      const uint32_t q0 = (age >= 1u) ? c : kTNone, q1 = (age >= 2u) ? c1 : kTNone;
      if (__builtin_amdgcn_ballot_w64(q0 != kTNone) != 0)
      {
        loadRound(m, ringIdx, q0, (wc - 2u) & cmask, tmp);
        __builtin_amdgcn_s_waitcnt(kWaitVm0);   // here, once a launch - not left pending into the sample loop (see sampleT)
        storeRound(m, ringIdx, q0, tmp);
        if (q0 != kTNone) installed(m, q0 * kTChunk, kTChunk);
      }
      if (__builtin_amdgcn_ballot_w64(q1 != kTNone) != 0)
      {
        loadRound(m, ringIdx, q1, (wc - 2u) & cmask, tmp);
        __builtin_amdgcn_s_waitcnt(kWaitVm0);
        storeRound(m, ringIdx, q1, tmp);
        if (q1 != kTNone) installed(m, q1 * kTChunk, kTChunk);
      }
    }
    // the chunk after those: asked for now, used from the next boundary on - if the writer is done with it (two or more behind)
    // and nothing of this lane's is still waiting in the loaders' registers
    const uint32_t ask = (pend == kTNone && age >= 3u) ? ((c + 2u) & cmask) : kTNone;
    // (a lane whose request still waits keeps its staged pieces: the round is skipped for the whole wavefront then - a delay time
    // that jumped, rare - and asked again at the next boundary)
    const bool allFree = __builtin_amdgcn_ballot_w64(pend != kTNone) == 0;
    if (allFree && __builtin_amdgcn_ballot_w64(ask != kTNone) != 0)
    {
      loadRound(m, ringIdx, ask, (wc - 2u) & cmask, stage);
      pend = ask;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  MLD float sampleT(const VoiceMem& m, int ringIdx, float x, int32_t d)
  {
    const uint32_t lane = threadIdx.x & 63u;
    if (!uniformW)
    {
      // voices of one wavefront with different write indices (a host set them so): every sample straight to and from memory
      float* own = chunkOf(m, ringIdx, lane, w) + (w & (kTChunk - 1));
      __hip_atomic_store(own, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const uint32_t r = (w - (uint32_t)d) & m.memMask;
      float y = __hip_atomic_load(chunkOf(m, ringIdx, lane, r) + (r & (kTChunk - 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("" : "+v"(y));   // the wait for it here, not where the two paths join (see below)
      w = (w + 1) & m.memMask;
      return y;
    }
    float* col = strip(m, ringIdx);
    col[(w & (kTWrite - 1)) * kTRowPad] = x;
    if (!primed || (w & (kTChunk - 1)) == 0) boundaryT(m, ringIdx, d);
    const uint32_t r = (w - (uint32_t)d) & m.memMask;
    const bool inW = (r >> 3) == (w >> 3), inR = held(m, r);
    const uint32_t row = inW ? (r & (kTWrite - 1)) : (uint32_t)kTWrite + (r & (2 * kTChunk - 1));
#ifdef MLGPU_RING_X_NOLDSREAD
    float y = x + (float)row;
#else
    float y = col[row * kTRowPad];
#endif
#ifndef MLGPU_RING_X_NOMISS
    if (__builtin_amdgcn_ballot_w64(!(inW || inR)) != 0)
    {
      // a sample no window holds (a delay of 8 to 47 samples: its chunk is too close behind the writer to be fetched a period
      // ahead; a delay time that jumped): from memory, complete there since the flush that ended its eight samples
      unpark(m, ringIdx);
#ifdef MLGPU_RING_X_MISSPOISON
      if (!(inW || inR)) y = 12345.f; else
#endif
      if (!(inW || inR)) y = __hip_atomic_load(chunkOf(m, ringIdx, lane, r) + (r & (kTChunk - 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // The wait for that load belongs INSIDE this branch. Left to the join below, the compiler guards every later use of y with
      // s_waitcnt vmcnt(0) on the path that never loaded anything too - a memory round trip per sample behind the newest store,
      // 1 024 per launch: it was what made this layout slower than layout 1 with less traffic (1.82 -> 1.34 ms without this branch,
      // profiles/r05_ring_layouts.txt). A use of y here puts the wait here.
      asm volatile("" : "+v"(y));
    }
#endif
    if ((w & (kTWrite - 1)) == kTWrite - 1)
    {
      flushT(m, ringIdx, w & ~(uint32_t)(kTWrite - 1));
      keepHistory(m, ringIdx, w & ~(uint32_t)(kTWrite - 1));
    }
    w = (w + 1) & m.memMask;
    return y;
  }

  // ---- layout 4: sector trips (see the constants above) ----
  // The held sector lives in LDS between trips (2 KiB per wavefront and ring, [half][lane][4]: 16-byte accesses without a bank
  // conflict): eight more registers per ring across the trip boundary were what made the 8-ring graph spill (254 + 406 spilled).
  float sN[8], sY[8], sX[8];  // the sector loaded in the prologue, this trip's reads, this trip's writes
  uint32_t sTag;              // which sector the LDS slot holds (kTNone: none)
  uint32_t sNTag;             // which sector sN holds or will hold - its load may still be in flight (kTNone: none)
  int32_t sD0;                // the delay time of the trip's first sample
  // ... and the node's last 16 written samples, [position & 15][lane] (4 KiB per wavefront and node): a lane whose delay time is
  // under 16 samples reads there - its sectors are not in memory yet -, so that short delay times cost an LDS access and not a
  // memory round trip per sample (the first form of this layout sent the whole wavefront through memory for the trip: 0.06 of the
  // HBM peak on a bank of strings whose top octave reads 1 .. 16 samples back).
  static constexpr int kSectorLdsFloats = 512, kHistLdsFloats = 1024;  // per wavefront: a ring's held sector, a node's history
  MLD float* held(const VoiceMem& m, int slot) const { return m.lds + (size_t)slot * kSectorLdsFloats + (threadIdx.x & 63u) * 4u; }
  MLD float* histRows(const VoiceMem& m) const { return m.lds + m.ldsHist + (threadIdx.x & 63u); }
  MLD void holdSector(const VoiceMem& m, int slot, const float (&v)[8]) const
  {
    float* h = held(m, slot);
    *(f32x4r*)h = f32x4r{v[0], v[1], v[2], v[3]};
    *(f32x4r*)(h + 256) = f32x4r{v[4], v[5], v[6], v[7]};
  }
  MLD void heldSector(const VoiceMem& m, int slot, float (&v)[8]) const
  {
    const float* h = held(m, slot);
    const f32x4r a = *(const f32x4r*)h, b = *(const f32x4r*)(h + 256);
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
    v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
  }
  bool sAligned, sFast;              // launch: w is the same multiple of 8 in every lane; trip: served from sH / sN
  MLD void beginS(const VoiceMem& m, int ringIdx, bool histWriter)
  {
    const uint32_t w0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)w);
    sAligned = __builtin_amdgcn_ballot_w64(w != w0) == 0 && (w0 & 7u) == 0u;
    sTag = sNTag = kTNone;
    sFast = false;
    sD0 = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) sN[j] = sY[j] = sX[j] = 0.f;
    if (sAligned && histWriter)
    {
      // the 16 samples before the launch's first come back into the history rows
      const uint32_t smask = m.memMask >> 3, ws = w >> 3;
      float a[8], b[8];
      loadSector(m, ringIdx, (ws - 2u) & smask, a);
      loadSector(m, ringIdx, (ws - 1u) & smask, b);
      float* h = histRows(m);
#pragma unroll
      for (int j = 0; j < 8; ++j)
      {
        h[((w - 16u + (uint32_t)j) & 15u) * 64u] = a[j];
        h[((w - 8u + (uint32_t)j) & 15u) * 64u] = b[j];
      }
    }
  }
  MLD void loadSector(const VoiceMem& m, int ringIdx, uint32_t sector, float (&out)[8]) const
  {
    const f32x4r* src = (const f32x4r*)chunkMem(m, ringIdx, sector * 8u);
    const f32x4r a = src[0], b = src[1];
    out[0] = a[0]; out[1] = a[1]; out[2] = a[2]; out[3] = a[3];
    out[4] = b[0]; out[5] = b[1]; out[6] = b[2]; out[7] = b[3];
  }
  // the trip's prologue (the generated kernel calls it for every ring before the trip's first sample): dPred = the delay time in effect
  // (slot: the LDS slot of the held sector, the ring's own unless a PitchbendableDelay's second core reads the first one's ring)
  MLD void tripBegin(const VoiceMem& m, int ringIdx, int32_t dPred, int slot = -1)
  {
    if (slot < 0) slot = ringIdx;
    if (!sAligned) return;
    // (a lane that reads under 16 samples back is served from the history rows: what it loads here is never looked at)
    const uint32_t smask = m.memMask >> 3;
    const uint32_t s0 = ((w - (uint32_t)dPred) & m.memMask) >> 3, s1 = (s0 + 1u) & smask;
    // the held sector is the one needed (the usual case: the read position moved on by 8) - else it comes from memory too. One branch
    // for the wavefront, unconditional loads inside it: a lane that holds the right sector already gets the same eight floats again.
    const bool far = dPred >= kSectorMinDelay;
    if (__builtin_amdgcn_ballot_w64(far && sTag != s0) != 0)
    {
      float h[8];
      loadSector(m, ringIdx, s0, h);
      holdSector(m, slot, h);
      sTag = far ? s0 : kTNone;  // (a tag vouches for a sector that was complete in memory when it was loaded: see tripStart)
    }
    // ... and the upper sector: usually asked for a whole trip ago already (tripStart's early request)
    if (__builtin_amdgcn_ballot_w64(far && sNTag != s1) != 0)
    {
      loadSector(m, ringIdx, s1, sN);
      sNTag = far ? s1 : kTNone;
    }
  }
  // sY = positions r0 .. r0 + 7 of the sixteen floats sH, sN (r0 & 7 = o): a barrel shifter, 11 + 9 + 8 selects
  MLD void barrel(uint32_t o, const float (&sH)[8])
  {
    const bool b4 = (o & 4u) != 0u, b2 = (o & 2u) != 0u, b1 = (o & 1u) != 0u;
    // Scalars made opaque one by one, not arrays: given `c ? a[j + 4] : a[j]` on array slots the optimizer makes ONE access with a
    // selected index, and the arrays then live in scratch memory behind a per-lane offset (what the first build of this did).
#define MLD_BARREL_IN(i, src) float e##i = src; asm("" : "+v"(e##i))
    MLD_BARREL_IN(0, sH[0]); MLD_BARREL_IN(1, sH[1]); MLD_BARREL_IN(2, sH[2]); MLD_BARREL_IN(3, sH[3]);
    MLD_BARREL_IN(4, sH[4]); MLD_BARREL_IN(5, sH[5]); MLD_BARREL_IN(6, sH[6]); MLD_BARREL_IN(7, sH[7]);
    MLD_BARREL_IN(8, sN[0]); MLD_BARREL_IN(9, sN[1]); MLD_BARREL_IN(10, sN[2]); MLD_BARREL_IN(11, sN[3]);
    MLD_BARREL_IN(12, sN[4]); MLD_BARREL_IN(13, sN[5]); MLD_BARREL_IN(14, sN[6]);
#undef MLD_BARREL_IN
    const float f0 = b4 ? e4 : e0, f1 = b4 ? e5 : e1, f2 = b4 ? e6 : e2, f3 = b4 ? e7 : e3, f4 = b4 ? e8 : e4, f5 = b4 ? e9 : e5,
                f6 = b4 ? e10 : e6, f7 = b4 ? e11 : e7, f8 = b4 ? e12 : e8, f9 = b4 ? e13 : e9, f10 = b4 ? e14 : e10;
    const float g0 = b2 ? f2 : f0, g1 = b2 ? f3 : f1, g2 = b2 ? f4 : f2, g3 = b2 ? f5 : f3, g4 = b2 ? f6 : f4, g5 = b2 ? f7 : f5,
                g6 = b2 ? f8 : f6, g7 = b2 ? f9 : f7, g8 = b2 ? f10 : f8;
    sY[0] = b1 ? g1 : g0;
    sY[1] = b1 ? g2 : g1;
    sY[2] = b1 ? g3 : g2;
    sY[3] = b1 ? g4 : g3;
    sY[4] = b1 ? g5 : g4;
    sY[5] = b1 ? g6 : g5;
    sY[6] = b1 ? g7 : g6;
    sY[7] = b1 ? g8 : g7;
  }
  // ---- a trip's steps (sAligned launches; w is the trip's base, the position of sample K is w + K) ----
  // sample 0, the delay time known: `fast` (the caller's wave-uniform decision: every delay time that reads this ring's memory in this
  // trip is at least 16 samples) - the eight reads r0 .. r0 + 7 into sY, from the held sector and the prologue's if the prologue took
  // the right position, else from memory now
  MLD void tripStart(const VoiceMem& m, int ringIdx, int32_t d, bool fast, int slot = -1)
  {
    if (slot < 0) slot = ringIdx;
    sD0 = d;
    sFast = fast;
    if (!fast)
    {
      sTag = sNTag = kTNone;
      return;
    }
    const uint32_t smask = m.memMask >> 3;
    const uint32_t r0 = (w - (uint32_t)d) & m.memMask, s0 = r0 >> 3, s1 = (s0 + 1u) & smask;
    float h[8];
    if (__builtin_amdgcn_ballot_w64(d >= kSectorMinDelay && (sTag != s0 || sNTag != s1)) != 0)
    {
      // the delay time is not what the prologue took it for (or no prologue ran): both sectors now, waited for here
      loadSector(m, ringIdx, s0, h);
      loadSector(m, ringIdx, s1, sN);
      asm volatile("" : "+v"(h[0]), "+v"(sN[0]));
    }
    else
      heldSector(m, slot, h);
    barrel(r0 & 7u, h);
    holdSector(m, slot, sN);  // the upper sector is the next trip's lower one
    // A tag says "this sector's eight samples, as they are in the ring": only a lane that reads 16 or more back may set one. A lane
    // under 16 has been loading sectors the writer had not finished (it does not look at them) - left tagged, a delay time that then
    // moves to 16 .. 23 finds "its" sectors held and reads the ring's previous lap (tools/ring_layout_soak.py, seed 64 case 102: a
    // FractionalDelay going 15.4 -> 17.5; the 1 300 cases of the first soak did not have it).
    const bool far = d >= kSectorMinDelay;
    sTag = far ? s1 : kTNone;
    // sN is free now: the sector after it - the NEXT trip's upper one if the delay time stays - is asked for at once and has this
    // whole trip's arithmetic to arrive in (a single wavefront per SIMD cannot hide a load behind another wavefront). It is complete
    // in memory for a lane that reads at least 32 samples behind the writer (one that reads under 16 back does not look at it).
    if (__builtin_amdgcn_ballot_w64(d >= kSectorMinDelay && d < 2 * kSectorMinDelay) == 0)
    {
      const uint32_t next = (s1 + 1u) & smask;
      loadSector(m, ringIdx, next, sN);
      sNTag = far ? next : kTNone;  // (far lanes read 32 or more back in this branch)
    }
    else
      sNTag = kTNone;
  }
  // sample K's write: a fast trip keeps the sector's eight samples in registers and stores them TOGETHER after the eighth (two
  // 16-byte halves four samples apart were two partial writes of a 32-byte sector: 1.5 x the write traffic, profiles/r06_ring_layouts.txt)
  MLD void writeS(const VoiceMem& m, int ringIdx, float x, int K)
  {
    if (sFast)
    {
      sX[K] = x;
      histRows(m)[((w + (uint32_t)K) & 15u) * 64u] = x;
      if (K == 7)
      {
        f32x4r* dst = (f32x4r*)chunkMem(m, ringIdx, w);
        dst[0] = f32x4r{sX[0], sX[1], sX[2], sX[3]};
        dst[1] = f32x4r{sX[4], sX[5], sX[6], sX[7]};
      }
    }
    else
    {
      const uint32_t pos = w + (uint32_t)K;
      __hip_atomic_store(chunkMem(m, ringIdx, pos) + (pos & 7u), x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  // sample K's read (after its write, as the reference: a delay time of 0 reads the sample just written). VARY: the delay time may
  // differ from sample to sample (IntegerDelay / FractionalDelay with a delay-time signal); a PitchbendableDelay's moves at samples 0
  // and 16 of 32 only. wr: the core whose registers hold this trip's written samples (itself; a PitchbendableDelay's second core
  // reads the first one's ring).
  template <bool VARY>
  MLD float readS(const VoiceMem& m, int ringIdx, int32_t d, int K, const RingCore& wr)
  {
    float y;
    if (sFast)
    {
      y = sY[K];
      if (__builtin_amdgcn_ballot_w64(d < kSectorMinDelay) != 0)
      {
        // lanes that read under 16 samples back: the history rows (one LDS read for the wavefront, its row by lane; d = 0 is the
        // sample just written)
        const float hy = histRows(m)[((w + (uint32_t)K - (uint32_t)d) & 15u) * 64u];
        y = (d < kSectorMinDelay) ? hy : y;
      }
      if (VARY && K != 0 && __builtin_amdgcn_ballot_w64(d != sD0 && d >= kSectorMinDelay) != 0)
      {
        // a delay time that moves inside the trip and stays 16 or more: those lanes read their sample straight from memory (it is
        // there: everything older than this trip is)
        if (d != sD0 && d >= kSectorMinDelay)
        {
          const uint32_t r = (w + (uint32_t)K - (uint32_t)d) & m.memMask;
          y = __hip_atomic_load(chunkMem(m, ringIdx, r) + (r & 7u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("" : "+v"(y));
      }
    }
    else
    {
      const uint32_t r = (w + (uint32_t)K - (uint32_t)d) & m.memMask;
      y = __hip_atomic_load(chunkMem(m, ringIdx, r) + (r & 7u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("" : "+v"(y));  // (the wait for it here, not where the paths join: see sampleT)
    }
    return y;
  }
  MLD void advanceS(const VoiceMem& m, int K)
  {
    if (K == 7) w = (w + 8u) & m.memMask;
  }
  // one ring, one reader (IntegerDelay, FractionalDelay)
  template <bool VARY>
  MLD float sampleS(const VoiceMem& m, int ringIdx, float x, int32_t d, int K)
  {
    if (!sAligned)
    {
      // write indices that differ inside the wavefront or are not a multiple of 8 (a host set them so): sample by sample through memory
      __hip_atomic_store(chunkMem(m, ringIdx, w) + (w & 7u), x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const uint32_t r = (w - (uint32_t)d) & m.memMask;
      float y = __hip_atomic_load(chunkMem(m, ringIdx, r) + (r & 7u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("" : "+v"(y));
      w = (w + 1) & m.memMask;
      return y;
    }
    if (K == 0) tripStart(m, ringIdx, d, true);
    writeS(m, ringIdx, x, K);
    const float y = readS<VARY>(m, ringIdx, d, K, *this);
    advanceS(m, K);
    return y;
  }

  // ---- layout 0 (rows): the sample's reads ahead of its arithmetic ----
  // A ring read depends on nothing the current sample computes (d >= 1: it is d samples old), but written where the delay node stands
  // in the graph it sits behind that sample's stores to the other rings - which the compiler must assume to alias - so a graph with
  // many rings walks one memory round trip after the other: the reference's reverb example (24 rings) waited 0.66 of its wave cycles
  // at 3.4 TB/s, and a plain load moved up in the source is moved back down to its use by the scheduler. The generated kernel
  // therefore calls readEarly as soon as the node's delay time is known - for the reverb at the top of the sample, all 24 loads back
  // to back - and the load is LDS-DMA (global_load_lds_dword: no destination register, the wavefront's 64 values land in a 256-byte
  // slot of LDS, m.lds + slot * 64), which the compiler can neither sink below the stores nor see the completion of: early<PENDING>
  // waits for it by count (memory operations return in order; PENDING = the loads the kernel issued AFTER this one before it issues
  // anything else, a lower bound of what may still be in flight when this one has landed) and reads the slot. The value is the
  // sample itself where the read lands on the write (a delay time of 0, or of the ring's whole length: the reference writes, then
  // reads).
  MLD void readEarly(const VoiceMem& m, int ringIdx, int32_t d, int slot) const
  {
    const float* src = m.ringPtr(ringIdx, (w - (uint32_t)d) & m.memMask);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(m.lds + slot * 64), 4, 0, 0);
  }
  template <int PENDING>
  static MLD void earlyWait()
  {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PENDING < 63 ? PENDING : 63) : "memory");
  }
  MLD float early(const VoiceMem& m, int32_t d, int slot, float x) const
  {
    const float y = m.lds[slot * 64 + (int)(threadIdx.x & 63u)];
    return ((uint32_t)d & m.memMask) == 0u ? x : y;
  }
  template <int PENDING>
  MLD float finishRows(const VoiceMem& m, int ringIdx, float x, int32_t d, int slot)
  {
    m.ringSet(ringIdx, w, x);
    earlyWait<PENDING>();
    const float y = early(m, d, slot, x);
    w = (w + 1) & m.memMask;
    return y;
  }

  // K: the sample's place in its trip of 8 (layout 4 only; the generated kernel passes a constant)
  template <bool VARY = false>
  MLD float sample(const VoiceMem& m, int ringIdx, float x, int32_t d, int K = 0)
  {
    if (kRingSectors) return sampleS<VARY>(m, ringIdx, x, d, K);
    if (kRingTransposed) return sampleT(m, ringIdx, x, d);
    if (kRingWindows) return sampleWindowed(m, ringIdx, x, d);
    m.ringSet(ringIdx, w, x);
    const uint32_t r = (w - (uint32_t)d) & m.memMask;
    const float y = m.ring(ringIdx, r);
    w = (w + 1) & m.memMask;
    return y;
  }
};

template <>
struct Proc<MLGPU_PROC_INTEGER_DELAY>  // :801-914   C{}  S{writeIndex:u32, delayInSamples:i32}
{
  static constexpr int NC = 0, NS = 2;
  static constexpr int kRings = 1;
  RingCore ringc;
  int32_t delay;
  VoiceMem mem;
  MLD void load(const VoiceMem& m, const KernelTables&)
  {
    mem = m;
    ringc.w = m.s(0) & m.memMask;  // (a write index a host set out of range cannot leave the ring)
    delay = (int32_t)m.s(1);
    ringc.begin(m, 0);
  }
  MLD void store(const VoiceMem& m) const
  {
    ringc.end(m, 0);
    m.set(0, ringc.w);
    m.set(1, (uint32_t)delay);
  }
  MLD float next(float x) { return ringc.sample(mem, 0, x, delay); }
  MLD float next(float x, float d)  // operator()(x, delay): mIntDelayInSamples = static_cast<int>(delay[n]), :877-897
  {
    delay = sse_cvtt(d);
    return ringc.sample(mem, 0, x, delay);
  }
  // layout 0: the read as early as the delay time is known (pre), the write and the value where the node stands (post)
  MLD void pre() { ringc.readEarly(mem, 0, delay, 0); }
  MLD void pre(float d)
  {
    delay = sse_cvtt(d);
    ringc.readEarly(mem, 0, delay, 0);
  }
  template <int PENDING>
  MLD float post(float x)
  {
    return ringc.finishRows<PENDING>(mem, 0, x, delay, 0);
  }
  // layout 4: the same with the sample's place K in its trip of 8; trip_begin before each trip
  MLD void trip_begin() { ringc.tripBegin(mem, 0, delay); }
  MLD float next_k(int K, float x) { return ringc.sample(mem, 0, x, delay, K); }
  MLD float next_k(int K, float x, float d)
  {
    delay = sse_cvtt(d);
    return ringc.sample<true>(mem, 0, x, delay, K);
  }
  MLD void end_vector() {}
};

template <>
struct Proc<MLGPU_PROC_ALLPASS1>  // :918-964  C{coeff}  S{x1, y1}
{
  static constexpr int NC = 1, NS = 2;
  float coeff, x1, y1;
  MLD void load(const VoiceMem& m, const KernelTables&)
  {
    coeff = m.c(0);
    x1 = u2f(m.s(0));
    y1 = u2f(m.s(1));
  }
  MLD void store(const VoiceMem& m) const
  {
    m.set(0, f2u(x1));
    m.set(1, f2u(y1));
  }
  MLD float next(float x)
  {
    const float y = x1 + (x - y1) * coeff;
    x1 = x;
    y1 = y;
    return y;
  }
  MLD void end_vector() {}
};

// FractionalDelay's arithmetic (:971-1044) on one ring: an IntegerDelay followed by an Allpass1 whose coefficient
// comes from the fractional part of the delay.
struct FracCore
{
  RingCore ringc;
  float x1, y1, apCoeff;
  int32_t delayInt;
  MLD void loadFrom(const VoiceMem& m, int s0)
  {
    ringc.w = m.s(s0) & m.memMask;
    x1 = u2f(m.s(s0 + 1));
    y1 = u2f(m.s(s0 + 2));
    delayInt = (int32_t)m.s(s0 + 3);
    apCoeff = u2f(m.s(s0 + 4));
  }
  MLD void storeTo(const VoiceMem& m, int s0) const
  {
    m.set(s0, ringc.w);
    m.set(s0 + 1, f2u(x1));
    m.set(s0 + 2, f2u(y1));
    m.set(s0 + 3, (uint32_t)delayInt);
    m.set(s0 + 4, f2u(apCoeff));
  }
  static MLD void delayFor(float d, int32_t& delayIntOut, float& apCoeffOut)  // setDelayInSamples, :991-1007; Allpass1::makeCoeffs :938-943
  {
    const float fDelayInt = __builtin_floorf(d);
    int32_t di = sse_cvtt(fDelayInt);
    float frac = d - fDelayInt;
    if ((frac < 0.618f) && (di > 0))
    {
      frac += 1.f;
      di -= 1;
    }
    delayIntOut = di;
    const float xm1 = (frac - 1.f);
    apCoeffOut = -0.53f * xm1 + 0.24f * xm1 * xm1;
  }
  MLD void setDelay(float d) { delayFor(d, delayInt, apCoeff); }
  MLD float ap(float d)  // the Allpass1 behind the integer delay, :945-953
  {
    const float y = x1 + (d - y1) * apCoeff;
    x1 = d;
    y1 = y;
    return y;
  }
  template <bool VARY = false>
  MLD float sample(const VoiceMem& m, int ringIdx, float x, int K = 0)
  {
    return ap(ringc.sample<VARY>(m, ringIdx, x, delayInt, K));
  }
};

template <>
struct Proc<MLGPU_PROC_FRACTIONAL_DELAY>  // :971-1044  C{}  S{writeIndex, x1, y1, delayInt:i32, allpassCoeff}
{
  static constexpr int NC = 0, NS = 5;
  static constexpr int kRings = 1;
  FracCore f;
  VoiceMem mem;
  MLD void load(const VoiceMem& m, const KernelTables&)
  {
    mem = m;
    f.loadFrom(m, 0);
    f.ringc.begin(m, 0);
  }
  MLD void store(const VoiceMem& m) const
  {
    f.ringc.end(m, 0);
    f.storeTo(m, 0);
  }
  MLD float next(float x) { return f.sample(mem, 0, x); }
  MLD float next(float x, float d)  // varying delay time, :1016-1025
  {
    f.setDelay(d);
    return f.sample(mem, 0, x);
  }
  MLD float next(float x, float d, float ticks)  // delay changes only where the int mask vChangeTicks is non-zero, :1029-1043
  {
    if (f2u(ticks) != 0u) f.setDelay(d);
    return f.sample(mem, 0, x);
  }
  MLD void pre() { f.ringc.readEarly(mem, 0, f.delayInt, 0); }
  MLD void pre(float d)
  {
    f.setDelay(d);
    f.ringc.readEarly(mem, 0, f.delayInt, 0);
  }
  MLD void pre(float d, float ticks)
  {
    if (f2u(ticks) != 0u) f.setDelay(d);
    f.ringc.readEarly(mem, 0, f.delayInt, 0);
  }
  template <int PENDING>
  MLD float post(float x)
  {
    return f.ap(f.ringc.finishRows<PENDING>(mem, 0, x, f.delayInt, 0));
  }
  MLD void trip_begin() { f.ringc.tripBegin(mem, 0, f.delayInt); }
  MLD float next_k(int K, float x) { return f.sample(mem, 0, x, K); }
  MLD float next_k(int K, float x, float d)
  {
    f.setDelay(d);
    return f.sample<true>(mem, 0, x, K);
  }
  MLD float next_k(int K, float x, float d, float ticks)
  {
    if (f2u(ticks) != 0u) f.setDelay(d);
    return f.sample<true>(mem, 0, x, K);
  }
  MLD void end_vector() {}
};

template <>
struct Proc<MLGPU_PROC_PITCHBENDABLE_DELAY>  // :1050-1106  C{}  S{delay1: 5 words, delay2: 5 words}; two rings
{
  static constexpr int NC = 0, NS = 10;
  static constexpr int kRings = 2;
  static constexpr bool kNeedsIndex = true;
  FracCore f1, f2;
  VoiceMem mem;
  // Ring layout 4. The reference feeds both FractionalDelays the SAME input (:1102-1104), so their rings hold the same samples
  // whenever their write indices agree - which they do unless a host set them apart: ring 0 alone is then written and read by both
  // cores (oneRing, decided per wavefront at the launch's start), and in a trip in which both cores have the same delay time - all
  // the time while the delay-time signal rests - ONE read serves both (shared): 8 bytes per sample of ring traffic instead of 16.
  // The default rows (layout 0) do the same (rowsOne): per sample one 4-byte row write and - while the two delay times agree - one row
  // read instead of two and two. (The second ring's memory then stays as it was: a host that sets the two write indices apart does so
  // before the first launch after a clear.)
  bool oneRing, shared, rowsOne;
  MLD void load(const VoiceMem& m, const KernelTables&)
  {
    mem = m;
    f1.loadFrom(m, 0);
    f2.loadFrom(m, 5);
    f1.ringc.begin(m, 0);
    f2.ringc.begin(m, 1, false);
    shared = false;
    oneRing = kRingSectors && f1.ringc.sAligned && f2.ringc.sAligned && __builtin_amdgcn_ballot_w64(f1.ringc.w != f2.ringc.w) == 0;
    // write indices a host set apart: the two rings are two rings, served sample by sample through memory (no trips: the second core
    // then needs no registers for a sector of its own writes)
    if (kRingSectors && !oneRing) f1.ringc.sAligned = f2.ringc.sAligned = false;
    rowsOne = !kRingWindows && __builtin_amdgcn_ballot_w64(f1.ringc.w != f2.ringc.w) == 0;
  }
  MLD void store(const VoiceMem& m) const
  {
    f1.ringc.end(m, 0);
    f2.ringc.end(m, 1);
    f1.storeTo(m, 0);
    f2.storeTo(m, 5);
  }
  // n = sample index inside the DSPVector. kFadePeriod = 32; delay 1 may change when n % 32 == 16, delay 2 when
  // n % 32 == 0; kvFade is the triangle 0..1..0 over the period (:1054-1062); result = lerp(d1, d2, fade) (:1102-1104)
  MLD void trip_begin()
  {
    f1.ringc.tripBegin(mem, 0, f1.delayInt);
    if (!oneRing) f2.ringc.tripBegin(mem, 1, f2.delayInt);
    else if (__builtin_amdgcn_ballot_w64(f1.delayInt != f2.delayInt) != 0) f2.ringc.tripBegin(mem, 0, f2.delayInt, 1);
    else f2.ringc.sTag = f2.ringc.sNTag = kTNone;  // (one read for both, as far as the prologue can tell)
  }
  MLD void update_delays(int n, float d)
  {
    const int r = n & 31;
    if ((n & 15) == 0)  // one branch and selects over values: two `if (...) fN.setDelay(d)` get merged by the optimizer
    {                   // into stores through a selected pointer, which keeps both cores in scratch memory
      int32_t ndInt;   // (not a copy of a core: its ring windows are register arrays)
      float ndCoeff;
      FracCore::delayFor(d, ndInt, ndCoeff);
      const bool first = (r == 16);
      f1.delayInt = first ? ndInt : f1.delayInt;
      f1.apCoeff = first ? ndCoeff : f1.apCoeff;
      f2.delayInt = first ? f2.delayInt : ndInt;
      f2.apCoeff = first ? f2.apCoeff : ndCoeff;
    }
  }
  // layout 0: the two reads as soon as the delay time is known (the second ring is the first while the write indices agree: rowsOne),
  // the write(s) and the crossfade where the node stands
  MLD void pre_i(int n, float d)
  {
    update_delays(n, d);
    f2.ringc.w = rowsOne ? f1.ringc.w : f2.ringc.w;
    f1.ringc.readEarly(mem, 0, f1.delayInt, 0);
    f2.ringc.readEarly(mem, rowsOne ? 0 : 1, f2.delayInt, 1);
  }
  template <int PENDING>
  MLD float post_i(int n, float x)
  {
    const int r = n & 31;
    float a, b;
    if (rowsOne)
    {
      const uint32_t w = f1.ringc.w;
      mem.ringSet(0, w, x);
      RingCore::earlyWait<PENDING>();
      a = f1.ringc.early(mem, f1.delayInt, 0, x);
      b = f2.ringc.early(mem, f2.delayInt, 1, x);
      f1.ringc.w = f2.ringc.w = (w + 1) & mem.memMask;
    }
    else
    {
      a = f1.ringc.finishRows<PENDING>(mem, 0, x, f1.delayInt, 0);
      b = f2.ringc.finishRows<PENDING>(mem, 1, x, f2.delayInt, 1);
    }
    const float y1 = f1.ap(a), y2 = f2.ap(b);
    const float fade = 2.f * ((r > 16) ? 1.0f - (float)r / 32.f : (float)r / 32.f);
    return y1 + (fade * (y2 - y1));
  }
  MLD float next_i(int n, float x, float d, int K = 0)
  {
    const int r = n & 31;
    update_delays(n, d);
    float y1, y2;
    if (kRingSectors && oneRing)
    {
      if (K == 0)
      {
        const bool fast = true;
        shared = __builtin_amdgcn_ballot_w64(f1.delayInt != f2.delayInt) == 0;
        f1.ringc.tripStart(mem, 0, f1.delayInt, fast);
        if (!shared) f2.ringc.tripStart(mem, 0, f2.delayInt, fast, 1);
        else
        {
          f2.ringc.sFast = true;
          f2.ringc.sTag = f2.ringc.sNTag = kTNone;
        }
      }
      f1.ringc.writeS(mem, 0, x, K);
      const float a = f1.ringc.readS<false>(mem, 0, f1.delayInt, K, f1.ringc);
      float b = a;
      if (!shared) b = f2.ringc.readS<false>(mem, 0, f2.delayInt, K, f1.ringc);
      f1.ringc.advanceS(mem, K);
      f2.ringc.advanceS(mem, K);
      y1 = f1.ap(a);
      y2 = f2.ap(b);
    }
    else if (!kRingWindows && rowsOne)
    {
      const uint32_t w = f1.ringc.w;
      mem.ringSet(0, w, x);
      // (two loads, no branch: while the delay times agree the second one finds the first one's line in the vector cache. A wave-uniform
      // branch around it - and, tried next, the quad's reads of every ring issued together at the quad's top behind such branches - kept
      // the loads of a many-ring graph from overlapping: the reverb example 4.6 -> 7.2 / 6.3 ms, profiles/r06_ring_layouts.txt)
      const float a = mem.ring(0, (w - (uint32_t)f1.delayInt) & mem.memMask);
      const float b = mem.ring(0, (w - (uint32_t)f2.delayInt) & mem.memMask);
      f1.ringc.w = f2.ringc.w = (w + 1) & mem.memMask;
      y1 = f1.ap(a);
      y2 = f2.ap(b);
    }
    else
    {
      y1 = f1.sample(mem, 0, x, K);
      y2 = f2.sample(mem, 1, x, K);
    }
    const float fade = 2.f * ((r > 16) ? 1.0f - (float)r / 32.f : (float)r / 32.f);
    return y1 + (fade * (y2 - y1));
  }
  MLD void end_vector() {}
};

// ---- HalfBandFilter, MLDSPFilters.h:1245-1310: the edges of a rate region (Upsample2xFunction / Downsample2xFunction) ----
// Polyphase pair of two-section allpass chains; order 4, rejection 70 dB, transition band 0.1 (:1306-1308).
//   upsample (:1248-1270):   y[2i] = apa1(apa0(x[i]));  y[2i+1] = apb1(apb0(x[i]))
//   downsample (:1272-1294): a0 = apa1(apa0(x[2i])); b0 = apb1(apb0(x[2i+1])); y[i] = (a0 + b1) * 0.5; b1 = b0
struct Ap1Core  // Allpass1::processSample, :945-953
{
  float x1, y1;
  MLD float step(float x, float coeff)
  {
    const float y = x1 + (x - y1) * coeff;
    x1 = x;
    y1 = y;
    return y;
  }
};

struct HalfBandCore
{
  Ap1Core a0, a1, b0, b1ap;
  float b1;
  MLD void load(const VoiceMem& m)
  {
    a0.x1 = u2f(m.s(0)); a0.y1 = u2f(m.s(1)); a1.x1 = u2f(m.s(2)); a1.y1 = u2f(m.s(3));
    b0.x1 = u2f(m.s(4)); b0.y1 = u2f(m.s(5)); b1ap.x1 = u2f(m.s(6)); b1ap.y1 = u2f(m.s(7));
    b1 = u2f(m.s(8));
  }
  MLD void store(const VoiceMem& m) const
  {
    m.set(0, f2u(a0.x1)); m.set(1, f2u(a0.y1)); m.set(2, f2u(a1.x1)); m.set(3, f2u(a1.y1));
    m.set(4, f2u(b0.x1)); m.set(5, f2u(b0.y1)); m.set(6, f2u(b1ap.x1)); m.set(7, f2u(b1ap.y1));
    m.set(8, f2u(b1));
  }
  MLD float pathA(float x) { return a1.step(a0.step(x, 0.07986642623635751f), 0.5453536510711322f); }
  MLD float pathB(float x) { return b1ap.step(b0.step(x, 0.28382934487410993f), 0.8344118914807379f); }
  MLD float down(float xe, float xo)
  {
    const float va = pathA(xe);
    const float vb = pathB(xo);
    const float y = (va + b1) * 0.5f;
    b1 = vb;
    return y;
  }
};

template <>
struct Proc<MLGPU_PROC_HALF_BAND>
{
  static constexpr int NC = 0, NS = 9;
  HalfBandCore hb;
  MLD void load(const VoiceMem& m, const KernelTables&) { hb.load(m); }
  MLD void store(const VoiceMem& m) const { hb.store(m); }
  MLD float up_a(float x) { return hb.pathA(x); }
  MLD float up_b(float x) { return hb.pathB(x); }
  MLD float down(float xe, float xo) { return hb.down(xe, xo); }
  MLD void end_vector() {}
};

// The output side of Downsample2xFunction (MLDSPFunctional.h:172-213): mUppers[0] plus the DSPVector that is handed out one
// process call late (mOutputBuffer and the returned first half together are 64 samples of delay on the upsampled stream).
// The 64 samples stay in the SoA state array (slot 9 + n, coalesced), read at sample n before the pair (n - 1, n) is
// replaced.
template <>
struct Proc<MLGPU_PROC_HALF_BAND_BUFFERED>
{
  static constexpr int NC = 0, NS = 9 + MLGPU_FLOATS_PER_DSPVECTOR;
  HalfBandCore hb;
  VoiceMem mem;
  MLD void load(const VoiceMem& m, const KernelTables&)
  {
    mem = m;
    hb.load(m);
  }
  MLD void store(const VoiceMem& m) const { hb.store(m); }
  float cur[4];
  MLD void begin_quad(int q)  // the quad's four delayed samples, fetched before push() rewrites any of them
  {
#pragma unroll
    for (int k = 0; k < 4; ++k) cur[k] = u2f(mem.s(9 + 4 * q + k));
  }
  MLD float delayed(int n) const { return cur[n & 3]; }
  MLD float delayedAt(int m) const { return u2f(mem.s(9 + m)); }  // a nested region: m is not the kernel's own sample index
  MLD void push(int n, float x)  // n odd: the inner sample made from outer samples n - 1 and n
  {
    const float ya = hb.pathA(x);
    const float yb = hb.pathB(x);
    mem.set(9 + n - 1, f2u(ya));
    mem.set(9 + n, f2u(yb));
  }
  MLD void end_vector() {}
};

// ---- TempoLock, MLDSPFilters.h:1478-1579 ---------------------------------------------------------------------------
// operator()(DSPVector x, float dydx, float isr): follows an input clock phasor at the ratio dydx. Per DSPVector it looks
// only at x[0] and x[1], then writes a phasor ramp; so it is a vector-rate processor whose first input must be a STREAMED
// graph input (the kernel reads the first two samples of the vector from it), the other two are floats per vector.
template <>
struct Proc<MLGPU_PROC_TEMPO_LOCK>  // C{}  S{omega, x1v}
{
  static constexpr int NC = 0, NS = 2;
  float omega, x1v, dydt;
  bool stopped;
  MLD void load(const VoiceMem& m, const KernelTables&)
  {
    omega = u2f(m.s(0));
    x1v = u2f(m.s(1));
    dydt = 0.f;
    stopped = false;
  }
  MLD void store(const VoiceMem& m) const
  {
    m.set(0, f2u(omega));
    m.set(1, f2u(x1v));
  }
  MLD void begin_vector(float x0, float x1, float dydx, float isr)
  {
    stopped = (x0 == -1.0f);  // input phasor inactive: reset and output 0 (:1503-1507)
    if (stopped)
    {
      omega = -1.0f;
      return;
    }
    float dxdt;
    if (omega > -1.f)
    {
      float dx = x0 - x1v;  // already running: average input slope over the last vector
      if (dx < 0.f) dx += 1.f;
      dxdt = dx / 64.f;
      dydt = dxdt * dydx;
      x1v = x0;
    }
    else
    {
      dxdt = x1 - x0;  // startup: jump to the input's phase
      dydt = dxdt * dydx;
      x1v = x0 - dxdt * 64.f;
      omega = __builtin_fmodf(x0 * dydx, 1.0f);
    }
    bool lock = false;  // lock when the ratio or its reciprocal is close to an integer (:1534-1539)
    if (abs_ps(dydx - __builtin_roundf(dydx)) < 0.001f) lock = true;
    const float rdydx = 1.0f / dydx;
    if (abs_ps(rdydx - __builtin_roundf(rdydx)) < 0.001f) lock = true;
    if (lock)
    {
      float error;
      if (dydx >= 1.f)
      {
        const float ref = x0 * dydx;
        error = omega - (ref - __builtin_floorf(ref));
      }
      else
      {
        const float ref = omega / dydx;
        error = (ref - __builtin_floorf(ref)) - x0;
      }
      const float errorDiff = __builtin_roundf(error) - error;
      float correction = errorDiff * isr * 4.0f;
      const float lo = -dydt * 0.5f, hi = dydt * 1.0f;
      correction = (correction < lo) ? lo : (correction > hi ? hi : correction);  // ml::clamp, MLDSPScalarMath.h:68-72
      dydt += correction;
    }
  }
  MLD float next_n(int)
  {
    if (stopped) return 0.f;
    const float y = omega;
    omega += dydt;
    if (omega > 1.0f) omega -= 1.0f;
    return y;
  }
  // x computed inside the graph instead of streamed in: the same arithmetic, taken sample by sample. Everything a vector needs
  // is in x[0] except, in a start-up vector, the slope x[1] - x[0] (:1523) - and that is first used for sample 1: sample 0 of a
  // start-up vector is the phase jumped to, fmod(x[0] * dydx, 1) (:1526).
  float x0h;
  bool starting;
  MLD float next_x(int n, float x, float dydx, float isr)
  {
    if (n == 0)
    {
      x0h = x;
      starting = !(x == -1.0f) && !(omega > -1.f);
      if (starting)
      {
        stopped = false;
        return __builtin_fmodf(x * dydx, 1.0f);
      }
      begin_vector(x, 0.f, dydx, isr);
    }
    else if (n == 1 && starting)
    {
      begin_vector(x0h, x, dydx, isr);
      omega += dydt;  // sample 0's step
      if (omega > 1.0f) omega -= 1.0f;
    }
    return next_n(n);
  }
  MLD void end_vector() {}
};

// ---- compile-time chains -------------------------------------------------------------------

// Processors with a cheaper evaluation for well-behaved, launch-constant input declare
// next_fast(x) and input_is_odd(x); the kernel tests input_is_odd once per launch.
// A SawGen and a PulseGen on the same per-voice frequency whose phase counters are EQUAL (reset together, as the two waveforms of
// one analog-style oscillator always are): they advance by the same step every sample, so they stay equal, and a trip needs the
// phases, their extremes, the two corrections of the step at phase 0 and the zone tests only once - the saw subtracts the very
// value the pulse adds (both are `lo ? cLo : hi ? cHi : 0` of the same operands). The caller asks once per launch whether every
// lane of the wavefront has the two counters equal and neither oscillator is dense (`locked`, graph.hip); otherwise, or when the
// trip is suspect, the two processors make their own trips as before. Same operands through the same operations: the same bits.
// Returns false, with nothing done, when the trip is suspect: the caller then runs the two trips one by one.
template <int N>
MLD bool trip_locked(Proc<MLGPU_PROC_SAW_GEN>& saw, Proc<MLGPU_PROC_PULSE_GEN>& pulse, float cps, float w, float (&outSaw)[N], float (&outPulse)[N])
{
  {
    const BlepFreq<true> f = BlepFreq<true>::make(cps);
    const float dtS = f.dt * kPhaseScale, omdtS = f.omdt * kPhaseScale, wS = w * kPhaseScale;
    uint32_t omega32 = saw.omega32;
    float h[N], d[N];
    uint32_t lo = 0u, hi = 0u, dlo = 0u, dhi = 0u;
#pragma unroll
    for (int i = 0; i < N; ++i)
    {
      h[i] = phasor_next_hi(omega32, cps);
      d[i] = __builtin_amdgcn_fractf(__builtin_fmaf(h[i], kPhaseUnit, -w) + 1.0f);
      lo = i ? trip_umin(lo, f2u(h[i])) : f2u(h[i]);
      hi = i ? trip_umax(hi, f2u(h[i])) : f2u(h[i]);
      dlo = i ? trip_umin(dlo, f2u(d[i])) : f2u(d[i]);
      dhi = i ? trip_umax(dhi, f2u(d[i])) : f2u(d[i]);
    }
    const bool suspect = u2f(lo) < kTripTiny * kPhaseScale || u2f(hi) > kTripNearOne * kPhaseScale || u2f(dlo) < kTripTinyShifted || u2f(dhi) > kTripNearOneShifted;
    if (__builtin_amdgcn_ballot_w64(suspect) == 0)
    {
      const float cLo = f.correction(u2f(lo) * kPhaseUnit, true, false), cHi = f.correction(u2f(hi) * kPhaseUnit, false, false);
      const float cDownLo = f.correction(u2f(dlo), true, false), cDownHi = f.correction(u2f(dhi), false, false);
#pragma unroll
      for (int i = 0; i < N; ++i)
      {
        const float cUp = (h[i] < dtS) ? cLo : ((h[i] > omdtS) ? cHi : 0.f);
        const float cDown = f.lo(d[i]) ? cDownLo : (f.hi(d[i]) ? cDownHi : 0.f);
        outSaw[i] = __builtin_fmaf(h[i], 2.f * kPhaseUnit, -1.f) - cUp;
        outPulse[i] = (((h[i] >= wS) ? -1.f : 1.f) + cUp) - cDown;
      }
      saw.omega32 = omega32;
      pulse.omega32 = omega32;
      return true;
    }
  }
  return false;
}

// The same pair on a STREAMED frequency (a pitch signal: exp2Approx -> freq, the instrument bank's voice), one sample at a time.
// With equal counters (the caller's wave-uniform test at the start of the launch; they advance by the same step for ever) the two
// processors' per-sample forms repeat each other: the phase counter and its conversion, the zone tests of the step at phase 0,
// the wave-uniform skip question, and - the saw's polyBLEP being the very value the pulse adds for its rising step - the division
// and the polynomial. Here each is made once: one phase, the four zone tests, ONE skip question for both, and in the usual case
// (no lane inside two zones at once) one correction. Same operands through the same operations as Proc<SAW_GEN>::next and
// Proc<PULSE_GEN>::next_sw: saw - (nearUp ? c : 0) with the idle correction exactly 0.f, x - 0 == x; `full` (some lane needs the
// IEEE division) now covers the pulse's shifted phase for the saw too, which changes nothing for operands div_nr is exact on.
template <bool REGULAR_W>
MLD void step_locked_stream(Proc<MLGPU_PROC_SAW_GEN>& saw, Proc<MLGPU_PROC_PULSE_GEN>& pulse, float cps, float w, float& outSaw, float& outPulse)
{
  const float p = phasor_next(saw.omega32, cps);
  pulse.omega32 = saw.omega32;
  const float sawv = __builtin_fmaf(p, 2.f, -1.f);  // phasor_to_saw, PARITY note there
  const float pulsev = (p >= w) ? -1.f : 1.f;
  const float d = p - w + 1.0f;
  const float down = REGULAR_W ? __builtin_amdgcn_fractf(d) : d - (float)sse_cvtt(d);  // fractionalPart
  const bool downOdd = !REGULAR_W && !(abs_ps(down) <= 2.0f);
  const BlepFreq<false> f = BlepFreq<false>::make(cps, downOdd);
  const bool loUp = f.lo(p), nearUp = loUp || f.hi(p);
  const bool loDown = f.lo(down), nearDown = loDown || f.hi(down);
  outSaw = sawv;
  outPulse = pulsev;
  if (__builtin_amdgcn_ballot_w64(nearUp || nearDown) == 0) return;
  const bool full = f.anyLaneOdd();
  if (MLGPU_PULSE_SINGLE_BLEP && __builtin_amdgcn_ballot_w64(nearUp && nearDown) == 0)
  {
    const float c = f.correction(nearDown ? down : p, nearDown ? loDown : loUp, full);
    outPulse = nearDown ? (pulsev - c) : (nearUp ? (pulsev + c) : pulsev);
    outSaw = nearUp ? (sawv - c) : sawv;
    return;
  }
  const float cUp = f.correction(p, loUp, full), cDown = f.correction(down, loDown, full);
  outPulse = (pulsev + (nearUp ? cUp : 0.f)) - (nearDown ? cDown : 0.f);
  outSaw = sawv - (nearUp ? cUp : 0.f);
}

template <class P, class = void>
struct HasFastPath
{
  static constexpr bool value = false;
};
template <class P>
struct HasFastPath<P, decltype((void)&P::input_is_odd)>
{
  static constexpr bool value = true;
};

template <int... KINDS>
struct Chain;

template <>
struct Chain<>
{
  static constexpr int NC = 0, NS = 0;
  static constexpr bool kHasImpulse = false;
  MLD void load(VoiceMem, const KernelTables&) {}
  MLD void store(VoiceMem) const {}
  MLD float next(float x) { return x; }
  MLD void end_vector() {}
  static constexpr bool kHeadHasFastPath = false;
  static MLD bool head_input_is_odd(float) { return false; }
  template <bool FAST>
  MLD float next_head(float x)
  {
    return x;
  }
};

template <int K0, int... KS>
struct Chain<K0, KS...>
{
  Proc<K0> head;
  Chain<KS...> tail;
  static constexpr int NC = Proc<K0>::NC + Chain<KS...>::NC;
  static constexpr int NS = Proc<K0>::NS + Chain<KS...>::NS;
  static constexpr bool kHasImpulse = (K0 == MLGPU_PROC_IMPULSE_GEN) || Chain<KS...>::kHasImpulse;
  MLD void load(VoiceMem m, const KernelTables& t)
  {
    head.load(m, t);
    m.coeffs += (size_t)Proc<K0>::NC * m.V;
    m.state += (size_t)Proc<K0>::NS * m.V;
    tail.load(m, t);
  }
  MLD void store(VoiceMem m) const
  {
    head.store(m);
    m.coeffs += (size_t)Proc<K0>::NC * m.V;
    m.state += (size_t)Proc<K0>::NS * m.V;
    tail.store(m);
  }
  MLD float next(float x) { return tail.next(head.next(x)); }
  // FAST: the caller has checked head_input_is_odd(x) is false on every lane of the wavefront
  static constexpr bool kHeadHasFastPath = HasFastPath<Proc<K0>>::value;
  static MLD bool head_input_is_odd(float x)
  {
    if constexpr (kHeadHasFastPath)
      return Proc<K0>::input_is_odd(x);
    else
      return false;
  }
  template <bool FAST>
  MLD float next_head(float x)
  {
    if constexpr (FAST && kHeadHasFastPath)
      return tail.next(head.next_fast(x));
    else
      return tail.next(head.next(x));
  }
  MLD void end_vector()
  {
    head.end_vector();
    tail.end_vector();
  }
};

}  // namespace mldev
