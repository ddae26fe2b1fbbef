// mldsp_math.hpp — per-lane float math of the mldsp.h hot path for gfx950.
//
// Hand-written device restatement of the arithmetic in the reference's
// source/DSP/MLDSPMathSSE.h and the DEFINE_OP* families of source/DSP/MLDSPOps.h.
// One wavefront lane evaluates one element; there are no cross-lane operations here.
//
// Numerical contract: every function returns the same bits as the
// reference's SSE2 path, except the two hardware-approximate ops (sqrtApprox,
// divideApprox: v_rsq_f32 / v_rcp_f32 here, rsqrtps / rcpps there; 2^-11 relative).
// That requires (a) NO mul+add contraction — this translation unit is compiled with
// -ffp-contract=off and every fused operation is spelled __builtin_fmaf explicitly where
// it is provably bit-identical; (b) SSE semantics for min/max/convert, restated below.
#pragma once
#ifndef __HIPCC_RTC__  // hiprtc pre-includes the HIP runtime header (graph.hip compiles this at run time)
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

namespace mldev
{
#define MLD __device__ __forceinline__

// Floating-point mode of a kernel. The reference runs its DSP code either with the default MXCSR (denormals honoured: the
// kernels' default too, MODE.fp_denorm = 3 from the kernel descriptor) or inside an ml::UsingFlushDenormalsToZero scope
// (MLDSPUtils.h:51-96: MXCSR FZ | DAZ - denormal sources read as zero, denormal results written as zero, signs kept).
// gfx950's MODE[5:4] = 0 is the same rule for every f32 VALU instruction used here (add, sub, mul, fma, the packed forms,
// compares, the division expansion, sqrt: tools/ftz_probe.hip compares each with the SSE instruction under FZ | DAZ);
// moves, selects and bit operations pass denormals through unchanged on both machines. One scalar instruction at kernel
// entry, wave-uniform: the same code object serves both modes. hwreg(HW_REG_MODE = 1, offset 4, width 2).
MLD void apply_fp_mode(uint32_t flags)
{
  if (flags & 1u /* MLGPU_KFLAG_FLUSH_DENORMALS */) __builtin_amdgcn_s_setreg(1 | (4 << 6) | (1 << 11), 0);
}

// A SIMD picks, among the wavefronts that are ready, the one with the highest priority and then the OLDEST: left alone, the first
// wavefront of a SIMD runs ahead of the others for the whole launch. A voice bank is launched as exactly as many wavefronts as the
// chip holds (4 per SIMD at 262 144 voices), nothing is waiting to take a finished wavefront's place, and so the launch ends with a
// long stretch of three, two and at last one wavefront per SIMD - and one wavefront alone issues at 40 % of a SIMD's rate (DESIGN
// 3.11). Measured with per-wavefront clocks (tools/wave_clock.py): config 5's wavefronts ended at 52 %, 60 %, 80 % and 100 % of the
// launch. The cure is to take turns: every `turn` (a trip of a few samples) a wavefront takes the next of the four priority levels,
// offset by its hardware slot, so the wavefronts of a SIMD hold four different levels that rotate - each gets the same share, they
// arrive together, and the SIMD keeps its full issue rate to the end. (s_setprio takes an immediate: hence the switch.)
MLD uint32_t wave_slot() { return (uint32_t)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (3 << 11)) & 3u; }  // HW_REG_HW_ID.wave_id
// the same by the clock all wavefronts share (100 MHz): the four levels rotate every 2^shift ticks whatever the wavefront's own progress
MLD void take_turns_by_clock(uint32_t slot, int shift)
{
  const uint32_t now = (uint32_t)(__builtin_amdgcn_s_memrealtime() >> shift);
  switch ((now + slot) & 3u)
  {
    case 0: __builtin_amdgcn_s_setprio(0); break;
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    default: __builtin_amdgcn_s_setprio(3); break;
  }
}
MLD void take_turns(uint32_t turn)
{
  switch (turn & 3u)
  {
    case 0: __builtin_amdgcn_s_setprio(0); break;
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    default: __builtin_amdgcn_s_setprio(3); break;
  }
}

MLD float u2f(uint32_t u) { return __uint_as_float(u); }
MLD uint32_t f2u(float f) { return __float_as_uint(f); }

// _mm_min_ps / _mm_max_ps (MLDSPMathSSE.h:80-81): (a<b)?a:b / (a>b)?a:b, i.e. the SECOND
// operand when either is NaN. Not fminf/fmaxf.
// They are arithmetic-class instructions: under MXCSR.DAZ (flush mode) a denormal source is replaced by a signed zero
// before the comparison, and that zero is what comes back. v_max_f32 x, x (canonicalize) does exactly that in flush
// mode and is the identity for every non-NaN value in the default mode (tools/ftz_probe.hip, tests/test_gpu_denormals.py).
MLD float sse_canon(float x)
{
  float r;
  asm("v_max_f32 %0, %1, %1" : "=v"(r) : "v"(x));
  return r;
}
// Canonicalizing the RESULT is enough (one instruction, not two): in flush mode the compare itself already reads a
// denormal source as a signed zero, so the same operand is selected as under DAZ, and it is then returned flushed.
MLD float sse_min(float a, float b) { return sse_canon((a < b) ? a : b); }
MLD float sse_max(float a, float b) { return sse_canon((a > b) ? a : b); }
// The same against a bound the caller knows: a constant (or any value) that is neither NaN nor zero. v_min_f32 / v_max_f32 differ
// from minps / maxps only when the SECOND operand is NaN, when the operands are zeros of different sign, and for a signaling NaN in
// the first operand (quieted where the SSE instruction hands back the bound) - so canonicalize the first operand (a signaling NaN
// becomes a quiet one, which both machines replace by the bound; in flush mode a denormal becomes the signed zero that DAZ reads)
// and take the hardware instruction: two instructions for three, and no compare writing a lane mask.
MLD float hw_min(float a, float b)
{
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
MLD float hw_max(float a, float b)
{
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
MLD float sse_min_bound(float a, float bound) { return hw_min(sse_canon(a), bound); }
MLD float sse_max_bound(float a, float bound) { return hw_max(sse_canon(a), bound); }
// ... and when the first operand is the result of an arithmetic instruction (or of one of these), never a signaling NaN, and already
// flushed in flush mode: the hardware instruction alone.
MLD float sse_min_arith(float a, float bound) { return hw_min(a, bound); }
MLD float sse_max_arith(float a, float bound) { return hw_max(a, bound); }

// lane l of the result = bit l of `mask` ? a : b, for a wave-uniform 64-bit lane mask (a ballot, or scalar logic on ballots):
// one v_cndmask_b32 that reads the SGPR pair, where `(mask >> lane) & 1` would be 64-bit vector shifts.
MLD float lane_select(uint64_t mask, float a, float b)
{
  // (the "s" constraint needs a value the compiler KNOWS to be uniform; a mask that reaches here through a struct member and a
  // few branches may have lost that mark. readfirstlane restores it and folds away where it was never lost.)
  // (the builtin returns int: without the casts the low half would be sign-extended over the high one)
  const uint64_t m = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(mask >> 32)) << 32) |
                     (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)mask);
  float r;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
  return r;
}

// The sum of G adjacent lanes (G = 2, 4, 8, 16: a group never leaves its DPP row of 16) in the order Synth::processVector adds its
// voices (source/app/MLSynth.h:43-57): ((((0 + x0) + x1) + ...) + x[G-1]). The result is in the LAST lane of each group; the
// other lanes hold partial sums of no meaning. A lane shift is a modifier of the add, not an instruction of its own.
template <int SH>
MLD float row_shr(float x)
{
  return u2f((uint32_t)__builtin_amdgcn_update_dpp(0, (int)f2u(x), 0x110 + SH, 0xF, 0xF, true));
}
template <int G, int I>
struct GroupSumStep
{
  static MLD float run(float acc, float x) { return GroupSumStep<G, I - 1>::run(acc + row_shr<I>(x), x); }
};
template <int G>
struct GroupSumStep<G, 0>
{
  static MLD float run(float acc, float x) { return acc + x; }
};
template <int G>
MLD float group_sum_in_order(float x)
{
  static_assert(G == 2 || G == 4 || G == 8 || G == 16, "a group is a power of two inside one row of 16 lanes");
  return GroupSumStep<G, G - 2>::run(0.f + row_shr<G - 1>(x), x);
}

// The same sums for groups of 16 (an instrument's voices) through LDS: a wavefront parks the f32x4 of its 64 voices for four
// consecutive quads in a strip of its own ([quad][group][voice][4] with a few pad words so that the reads below spread over the
// banks), then lane (g, s) adds up group g's sixteen voices at sample s of the 16 in voice order and stores that one float. Per
// voice-sample: 1/4 LDS write, 1 LDS read, 1 add, 1/16 store - the lane-shift chain above is 16 dependent adds per sample.
// LDS operations of one wavefront complete in issue order; the barrier only keeps the compiler from moving them across each other.
constexpr int kGroup16Quad = 4 * 80 + 4;       // floats per parked quad: 4 groups x (16 voices x 4 samples + 16 pad) + 4 pad
constexpr int kGroup16Strip = 4 * kGroup16Quad;
typedef float group16_f32x4 __attribute__((ext_vector_type(4)));
MLD void group16_park(float* strip, int qq, group16_f32x4 y)
{
  const unsigned l = threadIdx.x & 63u;
  *(group16_f32x4*)(strip + qq * kGroup16Quad + (l >> 4) * 80u + (l & 15u) * 4u) = y;
}
MLD void group16_sum_store(const float* strip, group16_f32x4* outQuad0, size_t strideQ)
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const unsigned l = threadIdx.x & 63u, s = l & 15u, qq = s >> 2, k = s & 3u;
  const float* p = strip + qq * kGroup16Quad + (l >> 4) * 80u + k;
  float acc = 0.f + p[0];  // ((0 + x0) + x1) + ...: Synth::processVector starts from a cleared output (MLSynth.h:43-57)
#pragma unroll
  for (int v = 1; v < 16; ++v) acc = acc + p[v * 4];
  __builtin_nontemporal_store(acc, (float*)(outQuad0 + (size_t)qq * strideQ) + k);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// clamp(x, lo, hi) = min(max(x, lo), hi) (MLDSPOps.h:747) in TWO instructions instead of six, for the common case the graph
// generator can prove: lo and hi are constants of the kernel, neither NaN nor zero, lo <= hi, and x is the result of an
// arithmetic instruction (so never a signaling NaN). v_max_f32 / v_min_f32 differ from maxps / minps only (a) when the SECOND
// operand is NaN, (b) when the operands compare equal with different bits, i.e. +0 against -0, and (c) for a signaling NaN in
// the first operand (quieted instead of replaced) - all excluded here; a quiet NaN x gives lo on both machines, a denormal x
// is flushed in flush mode exactly as sse_max's canonicalization would.
// Round 4: the pair is ONE instruction, v_med3_f32 - for lo <= hi the median of (x, lo, hi) is min(max(x, lo), hi), a NaN x gives
// min3 = lo like the pair, and a denormal x is flushed in flush mode like any other operand of the float unit.
#ifndef MLGPU_CLAMP_MED3
#define MLGPU_CLAMP_MED3 1
#endif
MLD float clamp_const_bounds(float x, float lo, float hi)
{
#if MLGPU_CLAMP_MED3
  float r;
  asm("v_med3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(lo), "v"(hi));
  return r;
#else
  float m, r;
  asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(x), "v"(lo));
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(m), "v"(hi));
  return r;
#endif
}

// _mm_cvttps_epi32 / _mm_cvtps_epi32 (MLDSPMathSSE.h:124-125): NaN and out-of-range give
// 0x80000000 ("integer indefinite"); v_cvt_i32_f32 would saturate / give 0 instead.
MLD int32_t sse_cvtt(float x)
{
  const bool ok = (x < 2147483648.0f) && (x >= -2147483648.0f);
  const int32_t r = (int32_t)(ok ? x : 0.0f);
  return ok ? r : INT32_MIN;
}
MLD int32_t sse_cvt(float x)
{
  // in range <=> |x| < 2^31: ONE compare (the source modifier is free). x == -2^31 itself lands on the other side, where the
  // answer is the same bits: cvtps2dq(-2^31) = 0x80000000 = the integer indefinite. NaN compares false: indefinite, as on x86.
  const bool ok = __builtin_fabsf(x) < 2147483648.0f;
  const int32_t r = (int32_t)__builtin_rintf(ok ? x : 0.0f);  // v_rndne_f32 + v_cvt_i32_f32
  return ok ? r : INT32_MIN;
}
// truncate when the caller has already bounded x inside int32 range
MLD int32_t cvtt_inrange(float x) { return (int32_t)x; }

// vecUnsignedIntToFloat, MLDSPMathSSE.h:130-135 (drops the LSB on purpose)
MLD float uint_to_float(uint32_t v)
{
  const float hi = (float)(int32_t)(v >> 1);
  return hi + hi;
}

MLD float abs_ps(float x) { return u2f(f2u(x) & 0x7FFFFFFFu); }  // vecAbs :86
MLD float sign_ps(float x)                                        // vecSign :88-90
{
  const uint32_t s = (f2u(x) & 0x80000000u) | 0x3F800000u;
  return u2f(!(x == -0.0f) ? s : 0u);
}
MLD float signbit_ps(float x) { return u2f((f2u(x) & 0x80000000u) | 0x3F800000u); }  // :92

// ---------------------------------------------------------------------------------------
// precise transcendentals — cephes via Pommier's sse_mathfun, MLDSPMathSSE.h:292-636.
// Same operation order, separate mul and add.

MLD float vec_log(float x)  // :308-373
{
  const bool invalid = (x <= 0.0f);
  x = sse_max(x, u2f(0x00800000u));
  int32_t emm0 = (int32_t)(f2u(x) >> 23);
  x = u2f((f2u(x) & ~0x7f800000u) | 0x3f000000u);
  emm0 -= 0x7f;
  float e = (float)emm0;
  e = e + 1.0f;
  const bool mask = (x < 0.707106781186547524f);
  float tmp = mask ? x : 0.0f;
  x = x - 1.0f;
  e = e - (mask ? 1.0f : 0.0f);
  x = x + tmp;
  const float z = x * x;
  float y = 7.0376836292E-2f;
  y = y * x;
  y = y + -1.1514610310E-1f;
  y = y * x;
  y = y + 1.1676998740E-1f;
  y = y * x;
  y = y + -1.2420140846E-1f;
  y = y * x;
  y = y + 1.4249322787E-1f;
  y = y * x;
  y = y + -1.6668057665E-1f;
  y = y * x;
  y = y + 2.0000714765E-1f;
  y = y * x;
  y = y + -2.4999993993E-1f;
  y = y * x;
  y = y + 3.3333331174E-1f;
  y = y * x;
  y = y * z;
  tmp = e * -2.12194440e-4f;
  y = y + tmp;
  tmp = z * 0.5f;
  y = y - tmp;
  tmp = e * 0.693359375f;
  x = x + y;
  x = x + tmp;
  return invalid ? u2f(0xFFFFFFFFu) : x;  // x | invalid_mask
}

MLD float vec_exp(float x)  // :389-440
{
  x = sse_min_bound(x, 88.3762626647949f);
  x = sse_max_arith(x, -88.3762626647949f);
  float fx = x * 1.44269504088896341f;
  fx = fx + 0.5f;
  int32_t emm0 = cvtt_inrange(fx);  // |fx| <= 128
  float tmp = (float)emm0;
  const float mask = (tmp > fx) ? 1.0f : 0.0f;
  fx = tmp - mask;
  tmp = fx * 0.693359375f;
  float z = fx * -2.12194440e-4f;
  x = x - tmp;
  x = x - z;
  z = x * x;
  float y = 1.9875691500E-4f;
  y = y * x;
  y = y + 1.3981999507E-3f;
  y = y * x;
  y = y + 8.3334519073E-3f;
  y = y * x;
  y = y + 4.1665795894E-2f;
  y = y * x;
  y = y + 1.6666665459E-1f;
  y = y * x;
  y = y + 5.0000001201E-1f;
  y = y * z;
  y = y + x;
  y = y + 1.0f;
  emm0 = cvtt_inrange(fx);
  const uint32_t p = ((uint32_t)emm0 + 0x7fu) << 23;
  return y * u2f(p);
}

// both cephes polynomials evaluated, then masked: :520-557 / :603-633
MLD float sincos_poly(float x, bool poly_mask, uint32_t sign_bit)
{
  const float z = x * x;
  float y = 2.443315711809948E-005f;
  y = y * z;
  y = y + -1.388731625493765E-003f;
  y = y * z;
  y = y + 4.166664568298827E-002f;
  y = y * z;
  y = y * z;
  const float tmp = z * 0.5f;
  y = y - tmp;
  y = y + 1.0f;
  float y2 = -1.9515295891E-4f;
  y2 = y2 * z;
  y2 = y2 + 8.3321608736E-3f;
  y2 = y2 * z;
  y2 = y2 + -1.6666654611E-1f;
  y2 = y2 * z;
  y2 = y2 * x;
  y2 = y2 + x;
  y2 = poly_mask ? y2 : 0.0f;
  y = poly_mask ? 0.0f : y;
  y = y + y2;
  return u2f(f2u(y) ^ sign_bit);
}

MLD float vec_sin(float x)  // :479-559
{
  uint32_t sign_bit = f2u(x) & 0x80000000u;
  x = abs_ps(x);
  float y = x * 1.27323954473516f;
  int32_t emm2 = sse_cvtt(y);
  emm2 = (int32_t)(((uint32_t)emm2 + 1u) & ~1u);
  y = (float)emm2;
  const uint32_t emm0 = ((uint32_t)emm2 & 4u) << 29;
  const bool poly_mask = (((uint32_t)emm2 & 2u) == 0u);
  sign_bit ^= emm0;
  const float xmm1 = y * -0.78515625f;
  const float xmm2 = y * -2.4187564849853515625e-4f;
  const float xmm3 = y * -3.77489497744594108e-8f;
  x = x + xmm1;
  x = x + xmm2;
  x = x + xmm3;
  return sincos_poly(x, poly_mask, sign_bit);
}

MLD float vec_cos(float x)  // :562-636
{
  x = abs_ps(x);
  float y = x * 1.27323954473516f;
  int32_t emm2 = sse_cvtt(y);
  emm2 = (int32_t)(((uint32_t)emm2 + 1u) & ~1u);
  y = (float)emm2;
  emm2 = (int32_t)((uint32_t)emm2 - 2u);
  const uint32_t emm0 = (~(uint32_t)emm2 & 4u) << 29;
  const bool poly_mask = (((uint32_t)emm2 & 2u) == 0u);
  const float xmm1 = y * -0.78515625f;
  const float xmm2 = y * -2.4187564849853515625e-4f;
  const float xmm3 = y * -3.77489497744594108e-8f;
  x = x + xmm1;
  x = x + xmm2;
  x = x + xmm3;
  return sincos_poly(x, poly_mask, emm0);
}

// ---------------------------------------------------------------------------------------
// approximate transcendentals — Horner polynomials, MLDSPMathSSE.h:752-864.
// Valid on [-pi, pi] with no range reduction, exactly like the reference.

MLD float vec_sin_approx(float x)  // :752-772
{
  const float x2 = x * x;
  float p = x2 * 2.147840177713078446686267852783203125e-6f;
  p = -1.92649182281456887722015380859375e-4f + p;
  p = x2 * p;
  p = 8.30897875130176544189453125e-3f + p;
  p = x2 * p;
  p = -0.166624367237091064453125f + p;
  p = x2 * p;
  p = 0.99997937679290771484375f + p;
  return x * p;
}
MLD float vec_cos_approx(float x)  // :774-792
{
  const float x2 = x * x;
  float p = x2 * 1.8791708498611114919185638427734375e-5f;
  p = -1.33926304988563060760498046875e-3f + p;
  p = x2 * p;
  p = 4.1496001183986663818359375e-2f + p;
  p = x2 * p;
  p = -0.4997930824756622314453125f + p;
  p = x2 * p;
  return 0.999959766864776611328125f + p;
}
MLD float vec_exp_approx(float x)  // :793-829
{
  const float val2 = x * 12102203.1615614f + 1065353216.f;
  // val2 is an arithmetic result; the zero bound of the second step can only hand back a zero of the other sign than maxps would,
  // and the conversion that follows reads both as 0
  const float val3 = sse_min_arith(val2, 2139095040.f);
  const float val4 = sse_max_arith(val3, 0.0f);
  const uint32_t val4i = (uint32_t)cvtt_inrange(val4);  // 0 <= val4 <= 2139095040 < 2^31
  const float xu = u2f(val4i & 0x7F800000u);
  const float b = u2f((val4i & 0x7FFFFFu) | 0x3F800000u);
  float p = b * 1.3671023382430374383648148e-2f;
  p = -2.88093587581985443087955e-3f + p;
  p = b * p;
  p = 0.168143436463395944830000f + p;
  p = b * p;
  p = 0.310670891004095530771135f + p;
  p = b * p;
  p = 0.510397365625862338668154f + p;
  return xu * p;
}
MLD float vec_log_approx(float val)  // :831-864
{
  const uint32_t vi = f2u(val);
  const int32_t expi = (int32_t)(vi >> 23);
  const float addcst = (val > 0.0f) ? -89.970756366f : 1.17549435e-38f /* FLT_MIN */;
  const float x = u2f((vi & 0x7FFFFFu) | 0x3F800000u);
  float p = x * 3.110401639e-2f;
  p = -0.288739945f + p;
  p = x * p;
  p = 1.130626167f + p;
  p = x * p;
  p = -2.461222105f + p;
  p = x * p;
  p = 3.529304993f + p;
  p = x * p;
  const float addCstResult = addcst + 0.69314718055995f * (float)expi;
  return p + addCstResult;
}

constexpr float kLogTwo = 0.69314718055994529f;   // MLDSPOps.h:601
constexpr float kLogTwoR = 1.4426950408889634f;   // MLDSPOps.h:602

// hardware-approximate forms (reference: rsqrtps / rcpps, 12-bit). gfx950 v_rsq_f32 /
// v_rcp_f32 are ~1 ulp, i.e. well inside the reference's own 1.5*2^-12 error band.
MLD float sqrt_approx(float x) { return x * __builtin_amdgcn_rsqf(x); }          // :84-85
MLD float div_approx(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }  // :79

// ---------------------------------------------------------------------------------------
// libm_sinf — the host libm's sinf, for coefficient code the reference evaluates PER SAMPLE
// (Lopass::makeCoeffsVec calls sinf twice per sample, MLDSPFilters.h:104-113).
//
// The reference's arithmetic here is a third-party dependency that is not under /root/reference:
// glibc 2.35 libm (sysdeps/ieee754/flt-32/s_sinf.c, the ARM optimized-routines algorithm by
// Szabolcs Nagy / Wilco Dijkstra): double-precision range reduction by pi/2 (fast path below 120,
// 192-bit 4/pi table above) and two degree-7/8 minimax polynomials in double. Restated here with
// the same operation order, every multiply and add separate (this file is built -ffp-contract=off).
// Constants: __sincosf_table / __inv_pio4 of that release. Pinned by the test suite's identical CPU
// restatement, which tests/test_oracle_golden.py checks against the host libm over ALL 2^32 inputs:
// identical everywhere except 12 arguments with 53 < |x| < 120 where the x86-64 FMA ifunc variant of
// glibc differs in the last bit (the SVF coefficient code only ever passes |x| <= pi).
MLD float libm_sinf_poly(double x, double x2, bool neg, int n)
{
  // polynomial of quadrant table 0; table 1 (n & 2) is the same with the cosine coefficients negated
  const double c0 = neg ? -0x1p0 : 0x1p0;
  const double c1 = neg ? 0x1.ffffffd0c621cp-2 : -0x1.ffffffd0c621cp-2;
  const double c2 = neg ? -0x1.55553e1068f19p-5 : 0x1.55553e1068f19p-5;
  const double c3 = neg ? 0x1.6c087e89a359dp-10 : -0x1.6c087e89a359dp-10;
  const double c4 = neg ? -0x1.99343027bf8c3p-16 : 0x1.99343027bf8c3p-16;
  const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
  if ((n & 1) == 0)
  {
    const double x3 = x * x2;
    const double t1 = s2 + x2 * s3;
    const double x7 = x3 * x2;
    const double s = x + x3 * s1;
    return (float)(s + x7 * t1);
  }
  const double x4 = x2 * x2;
  const double t2 = c3 + x2 * c4;
  const double t1 = c0 + x2 * c1;
  const double x6 = x4 * x2;
  const double c = t1 + x4 * c2;
  return (float)(c + x6 * t2);
}

MLD float libm_sinf(float y)
{
  const uint32_t top = (f2u(y) >> 20) & 0x7ffu;  // abstop12
  double x = (double)y;
  if (top < 0x3f4u)  // |y| < pi/4
  {
    if (top < 0x398u) return y;  // |y| < 2^-12
    return libm_sinf_poly(x, x * x, false, 0);
  }
  if (top < 0x42fu)  // |y| < 120
  {
    const double r = x * 0x1.45F306DC9C883p+23;  // 2/pi * 2^24
    const int n = ((int32_t)r + 0x800000) >> 24;
    x = x - (double)n * 0x1.921FB54442D18p0;
    const double sgn = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;  // sign[4] = {1,-1,-1,1}
    return libm_sinf_poly(x * sgn, x * x, (n & 2) != 0, n);
  }
  if (top < 0x7f8u)  // finite: 4/pi as a 192-bit integer, three 32x32 partial products
  {
    const uint32_t kInvPio4[24] = {0xa2u,       0xa2f9u,     0xa2f983u,   0xa2f9836eu, 0xf9836e4eu, 0x836e4e44u,
                                   0x6e4e4415u, 0x4e441529u, 0x441529fcu, 0x1529fc27u, 0x29fc2757u, 0xfc2757d1u,
                                   0x2757d1f5u, 0x57d1f534u, 0xd1f534ddu, 0xf534ddc0u, 0x34ddc0dbu, 0xddc0db62u,
                                   0xc0db6295u, 0xdb629599u, 0x6295993cu, 0x95993c43u, 0x993c4390u, 0x3c439041u};
    uint32_t xi = f2u(y);
    const int sign = (int)(xi >> 31);
    const uint32_t* arr = &kInvPio4[(xi >> 26) & 15];
    const int shift = (int)((xi >> 23) & 7);
    xi = (xi & 0xffffffu) | 0x800000u;
    xi <<= shift;
    uint64_t res0 = (uint64_t)(uint32_t)(xi * arr[0]);
    const uint64_t res1 = (uint64_t)xi * arr[4];
    const uint64_t res2 = (uint64_t)xi * arr[8];
    res0 = (res2 >> 32) | (res0 << 32);
    res0 += res1;
    const uint64_t nn = (res0 + (1ULL << 61)) >> 62;
    res0 -= nn << 62;
    x = (double)(int64_t)res0 * 0x1.921FB54442D18p-62;
    const int n = (int)nn;
    const int q = (n + sign) & 3;
    const double sgn = (q == 1 || q == 2) ? -1.0 : 1.0;
    return libm_sinf_poly(x * sgn, x * x, ((n + sign) & 2) != 0, n);
  }
  return (y - y) / (y - y);  // inf / NaN -> NaN
}

// ---- the same function on the SVF coefficient code's own ground, cheaper -------------------------------------------------------
// Lopass::makeCoeffsVec passes pi * omega and 2 pi * omega with omega <= 0.5 (clamped), i.e. arguments in [0, pi_f]. There the
// algorithm above is: y < 0.75 -> the sine polynomial on y itself; else n = round(y * 2 / pi) in {0, 1, 2}, x = y - n * pi/2 (the
// product is exact for these n), and the sine polynomial of x, the cosine polynomial of x, or the sine polynomial of -x. Written
// out as glibc does, that is 9-11 separate double operations per polynomial. The results below come from FEWER operations - Horner
// forms with fused multiply-adds, 5 instead of 9 (sine) and 5 instead of 11 (cosine), the quadrant from two float comparisons -
// whose doubles differ from glibc's in the last bits, but whose ROUNDED FLOATS do not, for any argument of the domain: checked
// exhaustively, all 113 840 092 floats of [2^-12, pi_f], against the host libm (the test suite's CPU checker holds the same
// sequence in C: tests/test_oracle_golden.py::test_fast_sinf_forms_exhaustively; below 0.75, where glibc skips the reduction,
// the quadrant is 0 and x = y - 0 is y itself). -(sine polynomial of x) is the sine
// polynomial of -x exactly (every product and sum changes sign with its operands).
// kSinfT1 / kSinfT2: the smallest floats whose quadrant ((int32)(y * 2/pi * 2^24) + 2^23) >> 24 is 1 / 2 (found by the same scan).
constexpr float kSinfMin = 0x1p-12f, kSinfT1 = 0x1.921fb6p-1f, kSinfT2 = 0x1.2d97c8p+1f, kSinfMax = 0x1.921fb6p+1f;
MLD double sinf_sin_poly_fma(double x, double x2)
{
  double p = __builtin_fma(x2, -0x1.994eb3774cf24p-13, 0x1.1107605230bc4p-7);
  p = __builtin_fma(x2, p, -0x1.555545995a603p-3);
  return __builtin_fma(x * x2, p, x);
}
MLD double sinf_cos_poly_fma(double x2)
{
  double p = __builtin_fma(x2, 0x1.99343027bf8c3p-16, -0x1.6c087e89a359dp-10);
  p = __builtin_fma(x2, p, 0x1.55553e1068f19p-5);
  p = __builtin_fma(x2, p, -0x1.ffffffd0c621cp-2);
  return __builtin_fma(x2, p, 0x1p0);
}
// kSinfMin <= y < kSinfT1: quadrant 0, the sine polynomial on the argument itself (below 0.75 glibc skips the reduction, from
// there to kSinfT1 it subtracts 0 * pi/2)
MLD float libm_sinf_q0(float y)
{
  const double x = (double)y;
  return (float)sinf_sin_poly_fma(x, x * x);
}
// Both sines of the SVF coefficient code for a wavefront in which some lane is past quadrant 0: y1 = pi * omega in [kSinfMin,
// pi_f / 2] (quadrant 0 or 1), y2 = 2 * y1 in [2 kSinfMin, pi_f] (quadrant 0, 1 or 2). Branch-free: both polynomials, selects.
MLD void libm_sinf_pair(float y1, float y2, float& s1, float& s2)
{
  constexpr double kPio2 = 0x1.921FB54442D18p0;
  const bool q2b = (y2 >= kSinfT2), q1a = (y1 >= kSinfT1), q1b = (y2 >= kSinfT1) && !q2b;
  const double xa = (double)y1 - (q1a ? kPio2 : 0.0);
  const double xb = (double)y2 - (q2b ? 2.0 * kPio2 : (q1b ? kPio2 : 0.0));
  const double xa2 = xa * xa, xb2 = xb * xb;
  const float sa = (float)sinf_sin_poly_fma(xa, xa2), sb = (float)sinf_sin_poly_fma(xb, xb2);
  const float ca = (float)sinf_cos_poly_fma(xa2), cb = (float)sinf_cos_poly_fma(xb2);
  s1 = q1a ? ca : sa;
  s2 = q2b ? -sb : (q1b ? cb : sb);
}

}  // namespace mldev
