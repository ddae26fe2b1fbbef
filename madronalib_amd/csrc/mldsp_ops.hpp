// mldsp_ops.hpp — the stateless DSPVector ops of MLDSPOps.h as one per-lane function template.
//
// apply<OP>(a, b, c) evaluates one element of the reference's DEFINE_OP1/OP2/OP3/... families
// (source/DSP/MLDSPOps.h:570-917) on 32-bit patterns (float or int32 as the op demands). Shared by
// the streaming op kernels (ops.hip) and by run-time generated graph kernels (graph.hip).
// Compile with -ffp-contract=off (see mldsp_math.hpp).
#pragma once
#include "mldsp_procs.hpp"  // the oscillators' waveshape functions double as stateless ops (pulls in mldsp_math.hpp)
#ifdef __HIPCC_RTC__
#include "mlgpu.h"  // provided as an in-memory header by graph.hip
#else
#include "../../include/mlgpu.h"
#endif

namespace mldev
{
template <int OP>
__device__ __forceinline__ uint32_t apply(uint32_t ua, uint32_t ub, uint32_t uc)
{
  const float a = u2f(ua), b = u2f(ub), c = u2f(uc);
  if constexpr (OP == MLGPU_OP_SQRT) return f2u(__builtin_sqrtf(a));  // correctly rounded (hipcc default)
  else if constexpr (OP == MLGPU_OP_SQRT_APPROX) return f2u(sqrt_approx(a));
  else if constexpr (OP == MLGPU_OP_ABS) return f2u(abs_ps(a));
  else if constexpr (OP == MLGPU_OP_SIGN) return f2u(sign_ps(a));
  else if constexpr (OP == MLGPU_OP_SIGN_BIT) return f2u(signbit_ps(a));
  else if constexpr (OP == MLGPU_OP_SIN) return f2u(vec_sin(a));
  else if constexpr (OP == MLGPU_OP_COS) return f2u(vec_cos(a));
  else if constexpr (OP == MLGPU_OP_LOG) return f2u(vec_log(a));
  else if constexpr (OP == MLGPU_OP_EXP) return f2u(vec_exp(a));
  else if constexpr (OP == MLGPU_OP_LOG2) return f2u(vec_log(a) * kLogTwoR);
  else if constexpr (OP == MLGPU_OP_EXP2) return f2u(vec_exp(kLogTwo * a));
  else if constexpr (OP == MLGPU_OP_SIN_APPROX) return f2u(vec_sin_approx(a));
  else if constexpr (OP == MLGPU_OP_COS_APPROX) return f2u(vec_cos_approx(a));
  else if constexpr (OP == MLGPU_OP_EXP_APPROX) return f2u(vec_exp_approx(a));
  else if constexpr (OP == MLGPU_OP_LOG_APPROX) return f2u(vec_log_approx(a));
  else if constexpr (OP == MLGPU_OP_LOG2_APPROX) return f2u(vec_log_approx(a) * kLogTwoR);
  else if constexpr (OP == MLGPU_OP_EXP2_APPROX) return f2u(vec_exp_approx(kLogTwo * a));
  else if constexpr (OP == MLGPU_OP_FRACTIONAL_PART) return f2u(a - (float)sse_cvtt(a));
  else if constexpr (OP == MLGPU_OP_ROUND_FLOAT_TO_INT) return (uint32_t)sse_cvt(a);
  else if constexpr (OP == MLGPU_OP_TRUNCATE_FLOAT_TO_INT) return (uint32_t)sse_cvtt(a);
  else if constexpr (OP == MLGPU_OP_INT_TO_FLOAT) return f2u((float)(int32_t)ua);
  else if constexpr (OP == MLGPU_OP_UNSIGNED_INT_TO_FLOAT) return f2u(uint_to_float(ua));
  else if constexpr (OP == MLGPU_OP_EXP_APPROX_OF_SIN_APPROX) return f2u(vec_exp_approx(vec_sin_approx(a)));
  else if constexpr (OP == MLGPU_OP_PHASOR_TO_SINE) return f2u(phasor_to_sine(a));
  else if constexpr (OP == MLGPU_OP_PHASOR_TO_SAW) return f2u(phasor_to_saw<false, true, true>(a, b));
  else if constexpr (OP == MLGPU_OP_PHASOR_TO_PULSE) return f2u(phasor_to_pulse<false, true, true>(a, b, c));
  else if constexpr (OP == MLGPU_OP_ADD) return f2u(a + b);
  else if constexpr (OP == MLGPU_OP_SUBTRACT) return f2u(a - b);
  else if constexpr (OP == MLGPU_OP_MULTIPLY) return f2u(a * b);
  else if constexpr (OP == MLGPU_OP_DIVIDE) return f2u(a / b);  // IEEE-correct v_div sequence
  else if constexpr (OP == MLGPU_OP_DIVIDE_APPROX) return f2u(div_approx(a, b));
  else if constexpr (OP == MLGPU_OP_POW) return f2u(vec_exp(vec_log(a) * b));
  else if constexpr (OP == MLGPU_OP_POW_APPROX) return f2u(vec_exp_approx(vec_log_approx(a) * b));
  else if constexpr (OP == MLGPU_OP_MIN) return f2u(sse_min(a, b));
  else if constexpr (OP == MLGPU_OP_MAX) return f2u(sse_max(a, b));
  else if constexpr (OP == MLGPU_OP_ADD_INT32) return ua + ub;
  else if constexpr (OP == MLGPU_OP_SUBTRACT_INT32) return ua - ub;
  else if constexpr (OP == MLGPU_OP_EQUAL) return (a == b) ? 0xFFFFFFFFu : 0u;
  else if constexpr (OP == MLGPU_OP_NOT_EQUAL) return (a != b) ? 0xFFFFFFFFu : 0u;
  else if constexpr (OP == MLGPU_OP_GREATER_THAN) return (a > b) ? 0xFFFFFFFFu : 0u;
  else if constexpr (OP == MLGPU_OP_GREATER_THAN_OR_EQUAL) return (a >= b) ? 0xFFFFFFFFu : 0u;
  else if constexpr (OP == MLGPU_OP_LESS_THAN) return (a < b) ? 0xFFFFFFFFu : 0u;
  else if constexpr (OP == MLGPU_OP_LESS_THAN_OR_EQUAL) return (a <= b) ? 0xFFFFFFFFu : 0u;
  else if constexpr (OP == MLGPU_OP_LERP) return f2u(a + (c * (b - a)));
  else if constexpr (OP == MLGPU_OP_INVERSE_LERP) return f2u((c - a) / (b - a));
  else if constexpr (OP == MLGPU_OP_CLAMP) return f2u(sse_min(sse_max(a, b), c));
  else if constexpr (OP == MLGPU_OP_WITHIN) return ((a >= b) && (a < c)) ? 0xFFFFFFFFu : 0u;
  else /* SELECT, SELECT_INT */ return (uc & ua) | (~uc & ub);
}

template <int OP>
constexpr int arity()
{
  return OP >= 64 ? 3 : (OP >= 32 ? 2 : 1);
}

// ops evaluated on floats inside a fused graph: float in, float out (masks travel as bit patterns)
template <int OP>
MLD float apply_f(float a, float b = 0.f, float c = 0.f)
{
  return u2f(apply<OP>(f2u(a), f2u(b), f2u(c)));
}

// index-dependent single-vector generators, MLDSPOps.h:962-990: element n of columnIndex(),
// rangeOpen / rangeClosed / interpolateDSPVectorLinear(start, end) = columnIndex() * DSPVector(interval)
// + DSPVector(offset): multiply, then add. /64 is an exact scaling; /63 is a true IEEE division.
template <int VOP>
MLD float vop(int n, float a = 0.f, float b = 0.f)
{
  const float idx = (float)n;
  if constexpr (VOP == MLGPU_VOP_COLUMN_INDEX) return idx;
  else if constexpr (VOP == MLGPU_VOP_RANGE_OPEN) return idx * ((b - a) / 64.f) + a;
  else if constexpr (VOP == MLGPU_VOP_RANGE_CLOSED) return idx * ((b - a) / 63.f) + a;
  else
  {
    const float interval = (b - a) / 64.f;
    return idx * interval + (a + interval);
  }
}

// ---- routing, MLDSPRouting.h:83-234 (one sample; x[] holds the N candidate inputs) ----------------------
// The reference converts `inputU * nInputs` to size_t: for a negative or NaN selector that is undefined
// behaviour there (an out-of-bounds read); here such samples select input 0 / no output. `s - truncf(s)`
// is fractionalPart without the SSE saturation quirk (truncf is exact for every float).
#define MLGPU_ROUTE_MAX 8
MLD int route_index(float u, int n)  // size_t(u * n), clamped into [0, n)
{
  const float r = u * (float)n;
  const int i = (r >= 0.f && r < 2147483648.f) ? (int)r : 0;
  return (i < n) ? i : 0;
}
MLD float route_pick(const float* x, int idx)
{
  float y = x[0];
#pragma unroll
  for (int k = 1; k < MLGPU_ROUTE_MAX; ++k) y = (idx == k) ? x[k] : y;
  return y;
}
MLD float route_multiplex(float s, const float* x, int n)  // :83-105
{
  const float u = s - __builtin_truncf(s);
  return route_pick(x, route_index(u, n));
}
MLD float route_multiplex_linear(float s, const float* x, int n)  // :111-137, scalar lerp a + m*(b - a)
{
  const float u = s - __builtin_truncf(s);
  const float real = u * (float)n;
  const float ip = __builtin_truncf(real);
  const float frac = real - ip;
  int i1 = (ip >= 0.f && ip < 2147483648.f) ? (int)ip : 0;
  if (i1 >= n) i1 = 0;
  const int i2 = (i1 + 1) % n;
  const float a = route_pick(x, i1), b = route_pick(x, i2);
  return a + frac * (b - a);
}
template <class... F>
MLD float route_multiplex_v(float s, F... xs)
{
  const float x[MLGPU_ROUTE_MAX] = {xs...};
  return route_multiplex(s, x, (int)sizeof...(F));
}
template <class... F>
MLD float route_multiplex_linear_v(float s, F... xs)
{
  const float x[MLGPU_ROUTE_MAX] = {xs...};
  return route_multiplex_linear(s, x, (int)sizeof...(F));
}
MLD float route_demultiplex(float s, float x, int j, int n)  // :142-174
{
  const float u = s - __builtin_truncf(s);
  return (route_index(u, n) == j) ? x : 0.f;
}
MLD float route_demultiplex_linear(float s, float x, int j, int n)  // :180-234
{
  const float u = s - __builtin_truncf(s);
  const float real = u * (float)n;
  const float ip = __builtin_truncf(real);
  int i1 = (ip >= 0.f && ip < 2147483648.f) ? (int)ip : 0;
  if (i1 >= n) i1 = 0;
  const float m = real - ip;
  const int i2 = (i1 + 1) % n;
  if (j == i1) return x * (1.f - m);
  if (j == i2) return x * m;
  return 0.f;
}

}  // namespace mldev
