/*
 * oracle/ml_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, scalar, CPU restatement of the madronalib mldsp.h hot path:
 * the SSE elementwise math, the generators and the IIR filters, each function citing the
 * reference file:line it follows (paths relative to /root/reference).
 *
 * Parity pinning: PINNED. tests/test_oracle_vs_ref.py checks every function here
 * bit-for-bit against the compiled reference itself (oracle/_ref/libmlref.so, built from
 * the reference headers by oracle/Makefile) when that library is present, and
 * tests/test_oracle_golden.py checks it against the committed golden vectors in
 * tests/golden/ (generated from the compiled reference by tests/golden/make_golden.py)
 * plus the reference's own test assertions (Tests/dspOpsTest.cpp:103-104,154,164;
 * Tests/dspGensTest.cpp:31).  Exceptions, by nature: sqrtApprox/divideApprox and the
 * Peak/RMS outputs use x86 rcpps/rsqrtps (12-bit hardware tables) in the reference; here
 * they are computed exactly and compared with a 2^-11 relative tolerance.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Build: gcc -std=c11 -O2 -ffp-contract=off (oracle/Makefile). x86-64 scalar float math
 * is SSE scalar math, i.e. lane-for-lane the same IEEE single operations as the
 * reference's 4-wide SSE2 code, provided the compiler never fuses mul+add (hence
 * -ffp-contract=off and no -march=native).
 */
#define _POSIX_C_SOURCE 200809L
#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../include/mlgpu.h"

#define VEC 64 /* kFloatsPerDSPVector, source/DSP/MLDSPMath.h:8-9 */

/* UsingFlushDenormalsToZero (source/DSP/MLDSPUtils.h:51-96): the reference sets the DAZ and FZ bits (0x8040) of MXCSR for
 * the scope of a process function. This restatement is plain scalar SSE arithmetic (gcc, x86-64), so the same two bits on
 * the calling thread give it the same mode; worker threads are created per call and inherit the caller's MXCSR. Returns
 * the previous setting (1 = flushing) so a test can restore it. */
#if defined(__SSE__)
#include <xmmintrin.h>
int mlorc_set_flush_denormals(int on)
{
  const unsigned csr = _mm_getcsr();
  _mm_setcsr(on ? (csr | 0x8040u) : (csr & ~0x8040u));
  return (csr & 0x8040u) == 0x8040u;
}
#else
int mlorc_set_flush_denormals(int on) { (void)on; return 0; }
#endif

/* ------------------------------------------------------------------------- */
/* bit casts and SSE-semantics primitives (source/DSP/MLDSPMathSSE.h:73-135)  */

static inline uint32_t f2u(float f)
{
  uint32_t u;
  memcpy(&u, &f, 4);
  return u;
}
static inline float u2f(uint32_t u)
{
  float f;
  memcpy(&f, &u, 4);
  return f;
}

/* _mm_min_ps / _mm_max_ps: (a<b)?a:b and (a>b)?a:b — returns b if either is NaN. They are arithmetic-class SSE
 * instructions: with MXCSR.DAZ set (ml::UsingFlushDenormalsToZero) a denormal source is replaced by a zero of its sign
 * BEFORE the comparison, so the operand that comes back is the zero, not the denormal (checked against the compiled
 * reference, tests/test_denormals_cpu.py). A C conditional would hand the denormal through, hence daz(). */
#if defined(__SSE__)
static inline float daz(float x)
{
  if (_mm_getcsr() & 0x0040u)
  {
    uint32_t u;
    memcpy(&u, &x, 4);
    if ((u & 0x7F800000u) == 0) u &= 0x80000000u;
    memcpy(&x, &u, 4);
  }
  return x;
}
#else
static inline float daz(float x) { return x; }
#endif
static inline float sse_min(float a, float b) { a = daz(a); b = daz(b); return (a < b) ? a : b; }
static inline float sse_max(float a, float b) { a = daz(a); b = daz(b); return (a > b) ? a : b; }

/* _mm_cvtps_epi32: round to nearest even; NaN / out of range -> 0x80000000. */
static inline int32_t sse_cvt(float x)
{
  if (!(x < 2147483648.0f) || !(x >= -2147483648.0f)) return INT32_MIN;
  return (int32_t)nearbyintf(x); /* default rounding mode = RNE, as MXCSR default */
}
/* _mm_cvttps_epi32: truncate; NaN / out of range -> 0x80000000. */
static inline int32_t sse_cvtt(float x)
{
  if (!(x < 2147483648.0f) || !(x >= -2147483648.0f)) return INT32_MIN;
  return (int32_t)x;
}
/* vecUnsignedIntToFloat, MLDSPMathSSE.h:130-135: drops the LSB on purpose. */
static inline float uint_to_float(uint32_t v)
{
  float hi = (float)(int32_t)(v >> 1);
  return hi + hi;
}
static inline float fabs_bits(float x) { return u2f(f2u(x) & 0x7FFFFFFFu); } /* vecAbs :86 */
/* vecSign :88-90 */
static inline float sign_ps(float x)
{
  uint32_t s = (f2u(x) & 0x80000000u) | 0x3F800000u;
  int neq = !(x == -0.0f); /* cmpneq: true when unordered */
  return u2f(neq ? s : 0u);
}
/* vecSignBit :92 */
static inline float signbit_ps(float x) { return u2f((f2u(x) & 0x80000000u) | 0x3F800000u); }

/* ------------------------------------------------------------------------- */
/* precise transcendentals: cephes / Pommier, MLDSPMathSSE.h:292-636          */

static float vec_log(float x) /* :308-373 */
{
  const int invalid = (x <= 0.0f);
  x = sse_max(x, u2f(0x00800000u)); /* cut off denormalized stuff */
  int32_t emm0 = (int32_t)(f2u(x) >> 23);
  x = u2f((f2u(x) & ~0x7f800000u) | f2u(0.5f));
  emm0 -= 0x7f;
  float e = (float)emm0;
  e = e + 1.0f;
  const int mask = (x < 0.707106781186547524f);
  float tmp = mask ? x : 0.0f;
  x = x - 1.0f;
  e = e - (mask ? 1.0f : 0.0f);
  x = x + tmp;
  float z = x * x;
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
  if (invalid) return u2f(0xFFFFFFFFu); /* x | all-ones: negative arg will be NaN */
  return x;
}

static float vec_exp(float x) /* :389-440 */
{
  x = sse_min(x, 88.3762626647949f);
  x = sse_max(x, -88.3762626647949f);
  float fx = x * 1.44269504088896341f;
  fx = fx + 0.5f;
  int32_t emm0 = sse_cvtt(fx);
  float tmp = (float)emm0;
  float mask = (tmp > fx) ? 1.0f : 0.0f;
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
  emm0 = sse_cvtt(fx);
  emm0 = (int32_t)((uint32_t)emm0 + 0x7fu);
  uint32_t p = (uint32_t)emm0 << 23;
  return y * u2f(p);
}

/* shared tail of vecSin/vecCos: both polynomials evaluated, then masked (:520-557) */
static inline float sincos_poly(float x, int poly_mask, uint32_t sign_bit)
{
  float z = x * x;
  float y = 2.443315711809948E-005f;
  y = y * z;
  y = y + -1.388731625493765E-003f;
  y = y * z;
  y = y + 4.166664568298827E-002f;
  y = y * z;
  y = y * z;
  float tmp = z * 0.5f;
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
  y2 = poly_mask ? y2 : 0.0f;  /* _mm_and_ps(mask, y2) */
  y = poly_mask ? 0.0f : y;    /* _mm_andnot_ps(mask, y) */
  y = y + y2;
  return u2f(f2u(y) ^ sign_bit);
}

static float vec_sin(float x) /* :479-559 */
{
  uint32_t sign_bit = f2u(x) & 0x80000000u;
  x = fabs_bits(x);
  float y = x * 1.27323954473516f;
  int32_t emm2 = sse_cvtt(y);
  emm2 = (int32_t)(((uint32_t)emm2 + 1u) & ~1u);
  y = (float)emm2;
  uint32_t emm0 = ((uint32_t)emm2 & 4u) << 29;
  int poly_mask = (((uint32_t)emm2 & 2u) == 0u);
  sign_bit ^= emm0;
  float xmm1 = y * -0.78515625f;
  float xmm2 = y * -2.4187564849853515625e-4f;
  float xmm3 = y * -3.77489497744594108e-8f;
  x = x + xmm1;
  x = x + xmm2;
  x = x + xmm3;
  return sincos_poly(x, poly_mask, sign_bit);
}

static float vec_cos(float x) /* :562-636 */
{
  x = fabs_bits(x);
  float y = x * 1.27323954473516f;
  int32_t emm2 = sse_cvtt(y);
  emm2 = (int32_t)(((uint32_t)emm2 + 1u) & ~1u);
  y = (float)emm2;
  emm2 = (int32_t)((uint32_t)emm2 - 2u);
  uint32_t emm0 = (~(uint32_t)emm2 & 4u) << 29;
  int poly_mask = (((uint32_t)emm2 & 2u) == 0u);
  float xmm1 = y * -0.78515625f;
  float xmm2 = y * -2.4187564849853515625e-4f;
  float xmm3 = y * -3.77489497744594108e-8f;
  x = x + xmm1;
  x = x + xmm2;
  x = x + xmm3;
  return sincos_poly(x, poly_mask, emm0);
}

/* ------------------------------------------------------------------------- */
/* approximate transcendentals, MLDSPMathSSE.h:752-864                        */

static float vec_sin_approx(float x) /* :752-772 */
{
  float x2 = x * x;
  return x * (0.99997937679290771484375f +
              x2 * (-0.166624367237091064453125f +
                    x2 * (8.30897875130176544189453125e-3f +
                          x2 * (-1.92649182281456887722015380859375e-4f +
                                x2 * 2.147840177713078446686267852783203125e-6f))));
}
static float vec_cos_approx(float x) /* :774-792 */
{
  float x2 = x * x;
  return 0.999959766864776611328125f +
         x2 * (-0.4997930824756622314453125f +
               x2 * (4.1496001183986663818359375e-2f +
                     x2 * (-1.33926304988563060760498046875e-3f +
                           x2 * 1.8791708498611114919185638427734375e-5f)));
}
static float vec_exp_approx(float x) /* :793-829 */
{
  float val2 = x * 12102203.1615614f + 1065353216.f;
  float val3 = sse_min(val2, 2139095040.f);
  float val4 = sse_max(val3, 0.0f);
  uint32_t val4i = (uint32_t)sse_cvtt(val4);
  float xu = u2f(val4i & 0x7F800000u);
  float b = u2f((val4i & 0x7FFFFFu) | 0x3F800000u);
  return xu * (0.510397365625862338668154f +
               b * (0.310670891004095530771135f +
                    b * (0.168143436463395944830000f +
                         b * (-2.88093587581985443087955e-3f +
                              b * 1.3671023382430374383648148e-2f))));
}
static float vec_log_approx(float val) /* :831-864 */
{
  uint32_t vi = f2u(val);
  int32_t expi = (int32_t)(vi >> 23);
  float addcst = (val > 0.0f) ? -89.970756366f : FLT_MIN;
  float x = u2f((vi & 0x7FFFFFu) | 0x3F800000u);
  float poly = x * (3.529304993f +
                    x * (-2.461222105f +
                         x * (1.130626167f + x * (-0.288739945f + x * 3.110401639e-2f))));
  float addCstResult = addcst + 0.69314718055995f * (float)expi;
  return poly + addCstResult;
}

#define K_LOG_TWO 0.69314718055994529f   /* MLDSPOps.h:601 */
#define K_LOG_TWO_R 1.4426950408889634f  /* MLDSPOps.h:602 */

/* ------------------------------------------------------------------------- */
/* elementwise op dispatch, MLDSPOps.h:570-917                                */

/* the waveshape functions (defined with the generators below) are also elementwise ops */
static inline float phasor_to_sine(float p);
static inline float phasor_to_saw(float p, float freq);
static inline float phasor_to_pulse(float p, float freq, float width);

static uint32_t op_scalar(int op, uint32_t ua, uint32_t ub, uint32_t uc, int* ok)
{
  const float a = u2f(ua), b = u2f(ub), c = u2f(uc);
  *ok = 1;
  switch (op)
  {
    case MLGPU_OP_SQRT: return f2u(sqrtf(a));                      /* :584 */
    case MLGPU_OP_SQRT_APPROX: return f2u(a * (1.0f / sqrtf(a)));  /* :585 x*rsqrtps(x): 2^-11 */
    case MLGPU_OP_ABS: return f2u(fabs_bits(a));                   /* :586 */
    case MLGPU_OP_SIGN: return f2u(sign_ps(a));                    /* :589 */
    case MLGPU_OP_SIGN_BIT: return f2u(signbit_ps(a));             /* :592 */
    case MLGPU_OP_SIN: return f2u(vec_sin(a));                     /* :595 */
    case MLGPU_OP_COS: return f2u(vec_cos(a));
    case MLGPU_OP_LOG: return f2u(vec_log(a));
    case MLGPU_OP_EXP: return f2u(vec_exp(a));
    case MLGPU_OP_LOG2: return f2u(vec_log(a) * K_LOG_TWO_R);      /* :603 */
    case MLGPU_OP_EXP2: return f2u(vec_exp(K_LOG_TWO * a));        /* :604 */
    case MLGPU_OP_SIN_APPROX: return f2u(vec_sin_approx(a));       /* :607 */
    case MLGPU_OP_COS_APPROX: return f2u(vec_cos_approx(a));
    case MLGPU_OP_EXP_APPROX: return f2u(vec_exp_approx(a));
    case MLGPU_OP_LOG_APPROX: return f2u(vec_log_approx(a));
    case MLGPU_OP_LOG2_APPROX: return f2u(vec_log_approx(a) * K_LOG_TWO_R); /* :613 */
    case MLGPU_OP_EXP2_APPROX: return f2u(vec_exp_approx(K_LOG_TWO * a));   /* :614 */
    case MLGPU_OP_FRACTIONAL_PART: return f2u(a - (float)sse_cvtt(a));      /* :825 */
    case MLGPU_OP_ROUND_FLOAT_TO_INT: return (uint32_t)sse_cvt(a);          /* :796 */
    case MLGPU_OP_TRUNCATE_FLOAT_TO_INT: return (uint32_t)sse_cvtt(a);      /* :797 */
    case MLGPU_OP_INT_TO_FLOAT: return f2u((float)(int32_t)ua);             /* :819 */
    case MLGPU_OP_UNSIGNED_INT_TO_FLOAT: return f2u(uint_to_float(ua));     /* :820 */
    case MLGPU_OP_EXP_APPROX_OF_SIN_APPROX: return f2u(vec_exp_approx(vec_sin_approx(a)));
    case MLGPU_OP_ADD: return f2u(a + b);                          /* :640-643 */
    case MLGPU_OP_SUBTRACT: return f2u(a - b);
    case MLGPU_OP_MULTIPLY: return f2u(a * b);
    case MLGPU_OP_DIVIDE: return f2u(a / b);
    case MLGPU_OP_DIVIDE_APPROX: return f2u(a * (1.0f / b));       /* :645 a*rcpps(b): 2^-11 */
    case MLGPU_OP_POW: return f2u(vec_exp(vec_log(a) * b));        /* :646 */
    case MLGPU_OP_POW_APPROX: return f2u(vec_exp_approx(vec_log_approx(a) * b)); /* :647 */
    case MLGPU_OP_MIN: return f2u(sse_min(a, b));                  /* :648 */
    case MLGPU_OP_MAX: return f2u(sse_max(a, b));                  /* :649 */
    case MLGPU_OP_ADD_INT32: return ua + ub;                       /* :714 */
    case MLGPU_OP_SUBTRACT_INT32: return ua - ub;                  /* :713 */
    case MLGPU_OP_EQUAL: return (a == b) ? 0xFFFFFFFFu : 0u;       /* :851-856 */
    case MLGPU_OP_NOT_EQUAL: return (a != b) ? 0xFFFFFFFFu : 0u;
    case MLGPU_OP_GREATER_THAN: return (a > b) ? 0xFFFFFFFFu : 0u;
    case MLGPU_OP_GREATER_THAN_OR_EQUAL: return (a >= b) ? 0xFFFFFFFFu : 0u;
    case MLGPU_OP_LESS_THAN: return (a < b) ? 0xFFFFFFFFu : 0u;
    case MLGPU_OP_LESS_THAN_OR_EQUAL: return (a <= b) ? 0xFFFFFFFFu : 0u;
    case MLGPU_OP_LERP: return f2u(a + (c * (b - a)));             /* :744 */
    case MLGPU_OP_INVERSE_LERP: return f2u((c - a) / (b - a));     /* :745 */
    case MLGPU_OP_CLAMP: return f2u(sse_min(sse_max(a, b), c));    /* :747 */
    case MLGPU_OP_WITHIN: return ((a >= b) && (a < c)) ? 0xFFFFFFFFu : 0u; /* :748 */
    case MLGPU_OP_SELECT:                                          /* :886 */
    case MLGPU_OP_SELECT_INT: return (uc & ua) | (~uc & ub);       /* :917 */
    case MLGPU_OP_PHASOR_TO_SINE: return f2u(phasor_to_sine(a));              /* MLDSPGens.h:316-338 */
    case MLGPU_OP_PHASOR_TO_SAW: return f2u(phasor_to_saw(a, b));             /* :362-369 */
    case MLGPU_OP_PHASOR_TO_PULSE: return f2u(phasor_to_pulse(a, b, c));      /* :342-358 */
    default: *ok = 0; return 0;
  }
}

int mlorc_op_apply(int op, const void* va, const void* vb, const void* vc, void* vout, size_t n)
{
  const uint32_t* a = (const uint32_t*)va;
  const uint32_t* b = (const uint32_t*)vb;
  const uint32_t* c = (const uint32_t*)vc;
  uint32_t* out = (uint32_t*)vout;
  int ok = 1;
  for (size_t i = 0; i < n; ++i)
  {
    out[i] = op_scalar(op, a ? a[i] : 0, b ? b[i] : 0, c ? c[i] : 0, &ok);
    if (!ok) return MLGPU_ERR_INVALID;
  }
  return MLGPU_OK;
}

/* add1..max1: second operand is one row, repeated (MLDSPOps.h:655-687) */
int mlorc_op_apply_rows1(int op, const float* a, const float* b64, float* out, size_t n_rows)
{
  int ok = 1;
  for (size_t r = 0; r < n_rows; ++r)
    for (int i = 0; i < VEC; ++i)
    {
      out[r * VEC + i] = u2f(op_scalar(op, f2u(a[r * VEC + i]), f2u(b64[i]), 0, &ok));
      if (!ok) return MLGPU_ERR_INVALID;
    }
  return MLGPU_OK;
}

/* horizontal ops, MLDSPOps.h:995-1035 with vecSumH/MaxH/MinH (MLDSPMathSSE.h:246-265):
 * per 4-group (x0 op x2) op (x1 op x3), then sequentially over the 16 groups. */
int mlorc_row_reduce(int rowop, const float* rows, float* out, size_t n_rows)
{
  for (size_t r = 0; r < n_rows; ++r)
  {
    const float* x = rows + r * VEC;
    float acc;
    switch (rowop)
    {
      case MLGPU_ROWOP_SUM:
      case MLGPU_ROWOP_MEAN:
        acc = 0.f;
        for (int g = 0; g < 16; ++g)
        {
          const float* q = x + 4 * g;
          float t0 = q[0] + q[2], t1 = q[1] + q[3];
          acc += (t0 + t1);
        }
        out[r] = (rowop == MLGPU_ROWOP_MEAN) ? acc * (1.0f / VEC) : acc; /* :1007-1011 */
        break;
      case MLGPU_ROWOP_MAX:
        acc = FLT_MIN; /* sic: smallest positive normal, MLDSPOps.h:1016 */
        for (int g = 0; g < 16; ++g)
        {
          const float* q = x + 4 * g;
          float t0 = sse_max(q[0], q[2]), t1 = sse_max(q[1], q[3]);
          float h = sse_max(t0, t1);
          acc = (acc > h) ? acc : h; /* ml::max scalar, MLDSPScalarMath.h: (a > b) ? a : b */
        }
        out[r] = acc;
        break;
      case MLGPU_ROWOP_MIN:
        acc = FLT_MAX;
        for (int g = 0; g < 16; ++g)
        {
          const float* q = x + 4 * g;
          float t0 = sse_min(q[0], q[2]), t1 = sse_min(q[1], q[3]);
          float h = sse_min(t0, t1);
          acc = (acc < h) ? acc : h;
        }
        out[r] = acc;
        break;
      default: return MLGPU_ERR_INVALID;
    }
  }
  return MLGPU_OK;
}

/* ------------------------------------------------------------------------- */
/* generators, source/DSP/MLDSPGens.h                                         */

/* constants of phasorToSine (:316-338). const_math::sqrt(2.0f) is itself an
 * approximation: the reference's sqrt2 is 0x3fb50505, not the correctly rounded
 * 0x3fb504f3; all five constants below are the reference's constexpr results. */
#define K_SQRT2 0x3fb50505u
#define K_SINE_DOMAIN 0x40b50505u /* sqrt2 * 4 */
#define K_SINE_SCALE 0x3f87c3b6u  /* 1 / (sqrt2 - sqrt2^3/6) */
#define K_SINE_FLIP 0x40350505u   /* sqrt2 * 2 */
#define K_ONE_SIXTH 0x3e2aaaabu

#define K_STEPS_PER_CYCLE 4294967296.0f            /* :184 2^32 */
#define K_CYCLES_PER_STEP 2.3283064365386963e-10f  /* :185 2^-32 */

/* PhasorGen::operator(), :187-203 */
static void phasor64(uint32_t* omega32, const float* cps, float* out)
{
  for (int n = 0; n < VEC; ++n)
  {
    float steps = cps[n] * K_STEPS_PER_CYCLE;
    int32_t istep = sse_cvt(steps); /* roundFloatToInt */
    *omega32 += (uint32_t)istep;
    out[n] = uint_to_float(*omega32) * K_CYCLES_PER_STEP;
  }
}

/* polyBLEP, :285-311 */
static inline float poly_blep(float t, float dt)
{
  float c = 0.f;
  if (t < dt)
  {
    t = t / dt;
    c = t + t - t * t - 1.0f;
  }
  else if (t > 1.0f - dt)
  {
    t = (t - 1.0f) / dt;
    c = t * t + t + t + 1.0f;
  }
  return c;
}

/* phasorToSine, :316-338 */
static inline float phasor_to_sine(float p)
{
  const float sqrt2 = u2f(K_SQRT2);
  float omega = p * u2f(K_SINE_DOMAIN) + (-sqrt2);
  float tri = (omega > sqrt2) ? (u2f(K_SINE_FLIP) - omega) : omega;
  /* scaleV * triangleV * (oneV - triangleV * triangleV * oneSixthV): left-assoc */
  return (u2f(K_SINE_SCALE) * tri) * (1.0f - (tri * tri) * u2f(K_ONE_SIXTH));
}
/* phasorToSaw, :362-369 */
static inline float phasor_to_saw(float p, float freq)
{
  float saw = p * 2.f - 1.f;
  return saw - poly_blep(p, freq);
}
/* phasorToPulse, :342-358 */
static inline float phasor_to_pulse(float p, float freq, float width)
{
  float pulse = (p >= width) ? -1.f : 1.f; /* select(-1, 1, omega >= width) */
  pulse = pulse + poly_blep(p, freq);
  float d = p - width + 1.0f;
  float down = d - (float)sse_cvtt(d); /* fractionalPart */
  pulse = pulse - poly_blep(down, freq);
  return pulse;
}

/* NoiseGen::operator(), :135-145 */
static void noise64(uint32_t* seed, float* out)
{
  for (int i = 0; i < VEC; ++i)
  {
    *seed = *seed * 0x0019660Du + 0x3C6EF35Fu;
    uint32_t temp = ((*seed >> 9) & 0x007FFFFFu) | 0x3F800000u;
    out[i] = u2f(temp) * 2.f - 3.f;
  }
}

/* TickGen::operator(), :29-46 */
static void tick64(float* omega, const float* cps, float* out)
{
  for (int n = 0; n < VEC; ++n)
  {
    out[n] = 0.f;
    *omega += cps[n];
    if (*omega > 1.0f)
    {
      *omega -= 1.0f;
      out[n] = 1.0f;
    }
  }
}

/* ImpulseGen table, :65-78 with makeWindow (MLDSPUtils.h:22-26), dspwindows::blackman
 * (:34-35), projections::linear (MLDSPProjections.h:147-167), normalize (MLDSPOps.h:1041). */
#define IMPULSE_TABLE_SIZE 17
static float g_impulse_table[VEC];
static int g_impulse_table_ready = 0;
static const float kTwoPiF = 6.2831853071795864769252867f; /* MLDSPScalarMath.h:23 */
static const float kPiF = 3.1415926535897932384626433f;    /* :24 */

static void build_impulse_table(void)
{
  float window[VEC], sinc[VEC], prod[VEC];
  memset(window, 0, sizeof(window));
  for (int i = 0; i < IMPULSE_TABLE_SIZE; ++i)
  {
    /* linear({0, size-1}, {0, 1}): m = (b2-b1)/(a2-a1); m*(x-a1)+b1 */
    float m = (1.f - 0.f) / ((IMPULSE_TABLE_SIZE - 1.f) - 0.f);
    float x = m * ((float)i - 0.f) + 0.f;
    window[i] = 0.42f - 0.5f * cosf(kTwoPiF * x) + 0.08f * cosf(2.f * kTwoPiF * x);
  }
  const float omega = 0.25f;
  for (int n = 0; n < VEC; ++n)
  {
    int i = n - (IMPULSE_TABLE_SIZE - 1) / 2;
    float pi_x = kTwoPiF * omega * i;
    sinc[n] = (i == 0) ? 1.f : sinf(pi_x) / pi_x;
  }
  for (int n = 0; n < VEC; ++n) prod[n] = sinc[n] * window[n];
  float s;
  mlorc_row_reduce(MLGPU_ROWOP_SUM, prod, &s, 1);
  for (int n = 0; n < VEC; ++n) g_impulse_table[n] = prod[n] / s;
  g_impulse_table_ready = 1;
}
void mlorc_impulse_table(float* out17)
{
  if (!g_impulse_table_ready) build_impulse_table();
  memcpy(out17, g_impulse_table, IMPULSE_TABLE_SIZE * sizeof(float));
}

/* ImpulseGen::operator(), :81-103 */
static void impulse64(float* omega, int32_t* counter, const float* cps, float* out)
{
  if (!g_impulse_table_ready) build_impulse_table();
  for (int n = 0; n < VEC; ++n)
  {
    out[n] = 0.f;
    *omega += cps[n];
    if (*omega > 1.0f)
    {
      *omega -= 1.0f;
      *counter = 0;
    }
    if (*counter < IMPULSE_TABLE_SIZE)
    {
      out[n] = g_impulse_table[*counter];
      (*counter)++;
    }
  }
}

/* OneShotGen::operator(), :238-258 */
static void oneshot64(uint32_t* omega32, uint32_t* gate, uint32_t* prev, const float* cps,
                      float* out)
{
  for (int n = 0; n < VEC; ++n)
  {
    float steps = cps[n] * K_STEPS_PER_CYCLE;
    int32_t istep = sse_cvt(steps);
    *omega32 += (uint32_t)istep * *gate;
    if (*omega32 < *prev)
    {
      *gate = 0;
      *omega32 = 0;
    }
    *prev = *omega32;
    out[n] = uint_to_float(*omega32) * K_CYCLES_PER_STEP;
  }
}

/* ------------------------------------------------------------------------- */
/* filters, source/DSP/MLDSPFilters.h                                         */

/* Lopass :118-133, Hipass :180-196, Bandpass :224-239 share the core */
static void svf64(int kind, const float* C, float* ic1, float* ic2, const float* in, float* out)
{
  const float g0 = C[0], g1 = C[1], g2 = C[2];
  for (int n = 0; n < VEC; ++n)
  {
    float v0 = in[n];
    float t0 = v0 - *ic2;
    float t1 = g0 * t0 + g1 * *ic1;
    float t2 = g2 * t0 + g0 * *ic1;
    float v1 = t1 + *ic1;
    float v2 = t2 + *ic2;
    *ic1 += 2.0f * t1;
    *ic2 += 2.0f * t2;
    if (kind == MLGPU_PROC_LOPASS)
      out[n] = v2;
    else if (kind == MLGPU_PROC_BANDPASS)
      out[n] = v1;
    else
      out[n] = v0 - C[3] * v1 - v2; /* Hipass, k = C[3] */
  }
}

/* LoShelf :288-302, HiShelf :369-383, Bell :427-441 share the core */
static void shelf64(int kind, const float* C, float* ic1, float* ic2, const float* in, float* out)
{
  const float a1 = C[0], a2 = C[1], a3 = C[2];
  for (int n = 0; n < VEC; ++n)
  {
    float v0 = in[n];
    float v3 = v0 - *ic2;
    float v1 = a1 * *ic1 + a2 * v3;
    float v2 = *ic2 + a2 * *ic1 + a3 * v3;
    *ic1 = 2 * v1 - *ic1;
    *ic2 = 2 * v2 - *ic2;
    if (kind == MLGPU_PROC_LO_SHELF)
      out[n] = v0 + C[3] * v1 + C[4] * v2; /* m1, m2 */
    else if (kind == MLGPU_PROC_HI_SHELF)
      out[n] = C[3] * v0 + C[4] * v1 + C[5] * v2; /* m0, m1, m2 */
    else
      out[n] = v0 + C[3] * v1; /* Bell, m1 */
  }
}

/* OnePole :466-475 */
static void onepole64(const float* C, float* y1, const float* in, float* out)
{
  for (int n = 0; n < VEC; ++n)
  {
    *y1 = C[0] * in[n] + C[1] * *y1;
    out[n] = *y1;
  }
}
/* DCBlocker :500-512 */
static void dcblock64(const float* C, float* x1, float* y1, const float* in, float* out)
{
  for (int n = 0; n < VEC; ++n)
  {
    const float x0 = in[n];
    const float y0 = x0 - *x1 + C[0] * *y1;
    *y1 = y0;
    *x1 = x0;
    out[n] = y0;
  }
}
/* Differentiator :522-534 */
static void diff64(float* x1, const float* in, float* out)
{
  out[0] = in[0] - *x1;
  for (int n = 1; n < VEC; ++n) out[n] = in[n] - in[n - 1];
  *x1 = in[VEC - 1];
}
/* Integrator :547-557 */
static void integ64(const float* C, float* y1, const float* in, float* out)
{
  for (int n = 0; n < VEC; ++n)
  {
    *y1 -= *y1 * C[0];
    *y1 += in[n];
    out[n] = *y1;
  }
}
/* Peak :582-614 (sqrtApprox computed exactly here: 2^-11 tolerance vs rsqrtps) */
static void peak64(const float* C, float* y1, int32_t* counter, const float* in, float* out)
{
  const int32_t hold = (int32_t)f2u(C[2]);
  float vy[VEC];
  for (int n = 0; n < VEC; ++n)
  {
    float xsq = in[n] * in[n];
    if (xsq > *y1)
    {
      *y1 = xsq;
      *counter = hold;
    }
    else if (*counter <= 0)
    {
      *y1 = C[0] * xsq + C[1] * *y1;
    }
    vy[n] = *y1;
  }
  if (*counter > 0) *counter -= VEC;
  for (int n = 0; n < VEC; ++n) out[n] = (vy[n] > 1e-20f) ? vy[n] * (1.0f / sqrtf(vy[n])) : 0.f;
}
/* RMS :637-652 */
static void rms64(const float* C, float* y1, const float* in, float* out)
{
  for (int n = 0; n < VEC; ++n)
  {
    float xsq = in[n] * in[n];
    *y1 = C[0] * xsq + C[1] * *y1;
    out[n] = (*y1 > 1e-20f) ? *y1 * (1.0f / sqrtf(*y1)) : 0.f;
  }
}

/* ADSR::processSample :704-786 ; state words: y,y1,x1,threshold,target,k,amp,segment */
enum { ADSR_A = 0, ADSR_D = 1, ADSR_S = 2, ADSR_R = 3, ADSR_OFF = 4 };
static float adsr_sample(const float* C, uint32_t* S, float x)
{
  float y = u2f(S[0]), y1 = u2f(S[1]), x1 = u2f(S[2]), threshold = u2f(S[3]);
  float target = u2f(S[4]), k = u2f(S[5]), amp = u2f(S[6]);
  int32_t segment = (int32_t)S[7];
  const float bias = 0.1f;

  if ((segment == ADSR_OFF) && (x == 0.f)) return 0.f;

  int crossed = ((y1 > threshold) != (y > threshold));
  int recalc = 0;
  if (crossed && (segment < ADSR_OFF))
  {
    segment++;
    recalc = 1;
  }
  int trigOn = (x1 == 0.f) && (x > 0.f);
  int trigOff = (x1 > 0.f) && (x == 0.f);
  if (trigOn)
  {
    segment = ADSR_A;
    amp = x;
    recalc = 1;
  }
  else if (trigOff)
  {
    segment = ADSR_R;
    recalc = 1;
  }
  if (recalc)
  {
    float startEnv = 0.f, endEnv = 0.f;
    switch (segment)
    {
      case ADSR_A: startEnv = 0.f; endEnv = 1.f; k = C[0]; break;
      case ADSR_D: startEnv = 1.f; endEnv = C[2]; k = C[1]; break;
      case ADSR_S: startEnv = C[2]; endEnv = C[2]; k = 0.f; y1 = C[2]; y = C[2]; break;
      case ADSR_R: startEnv = C[2]; endEnv = 0.f; k = C[3]; break;
      case ADSR_OFF: startEnv = 0.f; endEnv = 0.f; k = 0.f; y1 = 0.f; y = 0.f; break;
    }
    float segmentBias = (endEnv - startEnv) * bias;
    threshold = endEnv;
    target = endEnv + segmentBias;
  }
  x1 = x;
  y1 = y;
  y = y + k * (target - y);
  S[0] = f2u(y); S[1] = f2u(y1); S[2] = f2u(x1); S[3] = f2u(threshold);
  S[4] = f2u(target); S[5] = f2u(k); S[6] = f2u(amp); S[7] = (uint32_t)segment;
  return y * amp;
}

/* ------------------------------------------------------------------------- */
/* chains of processors                                                      */
float mlorc_libm_sinf(float y);

int mlorc_proc_num_coeffs(int kind)
{
  switch (kind)
  {
    case MLGPU_PROC_PHASOR_GEN: case MLGPU_PROC_SINE_GEN: case MLGPU_PROC_SAW_GEN:
    case MLGPU_PROC_NOISE_GEN: case MLGPU_PROC_TICK_GEN: case MLGPU_PROC_IMPULSE_GEN:
    case MLGPU_PROC_ONE_SHOT_GEN: case MLGPU_PROC_DIFFERENTIATOR: case MLGPU_PROC_TEST_SINE_GEN: return 0;
    case MLGPU_PROC_PULSE_GEN: case MLGPU_PROC_DC_BLOCKER: case MLGPU_PROC_INTEGRATOR:
    case MLGPU_PROC_GAIN: return 1;
    case MLGPU_PROC_ONE_POLE: case MLGPU_PROC_RMS: case MLGPU_PROC_LINEAR_GLIDE:
    case MLGPU_PROC_SAMPLE_ACCURATE_LINEAR_GLIDE: return 2;
    case MLGPU_PROC_INTERPOLATOR1: case MLGPU_PROC_INTEGER_DELAY: case MLGPU_PROC_FRACTIONAL_DELAY:
    case MLGPU_PROC_PITCHBENDABLE_DELAY: case MLGPU_PROC_TEMPO_LOCK: return 0;
    case MLGPU_PROC_ALLPASS1: return 1;
    case MLGPU_PROC_LOPASS: case MLGPU_PROC_BANDPASS: case MLGPU_PROC_PEAK: return 3;
    case MLGPU_PROC_HIPASS: case MLGPU_PROC_BELL: case MLGPU_PROC_ADSR: return 4;
    case MLGPU_PROC_LO_SHELF: return 5;
    case MLGPU_PROC_HI_SHELF: return 6;
    default: return -1;
  }
}
int mlorc_proc_num_state(int kind)
{
  switch (kind)
  {
    case MLGPU_PROC_GAIN: return 0;
    case MLGPU_PROC_PHASOR_GEN: case MLGPU_PROC_SINE_GEN: case MLGPU_PROC_SAW_GEN:
    case MLGPU_PROC_PULSE_GEN: case MLGPU_PROC_NOISE_GEN: case MLGPU_PROC_TICK_GEN:
    case MLGPU_PROC_ONE_POLE: case MLGPU_PROC_DIFFERENTIATOR: case MLGPU_PROC_INTEGRATOR:
    case MLGPU_PROC_RMS: case MLGPU_PROC_INTERPOLATOR1: case MLGPU_PROC_TEST_SINE_GEN: return 1;
    case MLGPU_PROC_SAMPLE_ACCURATE_LINEAR_GLIDE: return 4;
    case MLGPU_PROC_INTEGER_DELAY: case MLGPU_PROC_ALLPASS1: case MLGPU_PROC_TEMPO_LOCK: return 2;
    case MLGPU_PROC_FRACTIONAL_DELAY: return 5;
    case MLGPU_PROC_PITCHBENDABLE_DELAY: return 10;
    case MLGPU_PROC_LINEAR_GLIDE: return 3 + VEC;
    case MLGPU_PROC_IMPULSE_GEN: case MLGPU_PROC_LOPASS: case MLGPU_PROC_HIPASS:
    case MLGPU_PROC_BANDPASS: case MLGPU_PROC_LO_SHELF: case MLGPU_PROC_HI_SHELF:
    case MLGPU_PROC_BELL: case MLGPU_PROC_DC_BLOCKER: case MLGPU_PROC_PEAK: return 2;
    case MLGPU_PROC_ONE_SHOT_GEN: return 3;
    case MLGPU_PROC_ADSR: return 8;
    default: return -1;
  }
}

/* state after T::clear() (and after default construction, which differs for SineGen and ADSR) */
static void proc_state_init(int kind, uint32_t* S, int cleared)
{
  int ns = mlorc_proc_num_state(kind);
  for (int i = 0; i < ns; ++i) S[i] = 0;
  if (kind == MLGPU_PROC_SINE_GEN && cleared) S[0] = 0xC0000000u; /* kZeroPhase, MLDSPGens.h:375 */
  if (kind == MLGPU_PROC_ADSR) S[7] = ADSR_OFF;                   /* MLDSPFilters.h:700,702 */
  if (kind == MLGPU_PROC_TEMPO_LOCK) S[0] = 0xBF800000u;          /* _omega{-1.f}, MLDSPFilters.h:1481,1487 */
  if (kind == MLGPU_PROC_LINEAR_GLIDE) S[2] = 0xFFFFFFFFu;        /* mVectorsRemaining{-1}, MLDSPGens.h:441,513 */
  if (kind == MLGPU_PROC_SAMPLE_ACCURATE_LINEAR_GLIDE) S[3] = 0xFFFFFFFFu; /* mSamplesRemaining{-1}, :524,588 */
}

int mlorc_chain_clear(const int32_t* procs, int n_procs, size_t V, uint32_t* state)
{
  int s = 0;
  uint32_t sbuf[80];
  for (int p = 0; p < n_procs; ++p)
  {
    int ns = mlorc_proc_num_state(procs[p]);
    if (ns < 0) return MLGPU_ERR_INVALID;
    proc_state_init(procs[p], sbuf, 1);
    for (int i = 0; i < ns; ++i)
      for (size_t v = 0; v < V; ++v) state[(size_t)(s + i) * V + v] = sbuf[i];
    s += ns;
  }
  return MLGPU_OK;
}
int mlorc_chain_default_state(const int32_t* procs, int n_procs, size_t V, uint32_t* state)
{
  int s = 0;
  uint32_t sbuf[80];
  for (int p = 0; p < n_procs; ++p)
  {
    int ns = mlorc_proc_num_state(procs[p]);
    if (ns < 0) return MLGPU_ERR_INVALID;
    proc_state_init(procs[p], sbuf, 0);
    for (int i = 0; i < ns; ++i)
      for (size_t v = 0; v < V; ++v) state[(size_t)(s + i) * V + v] = sbuf[i];
    s += ns;
  }
  return MLGPU_OK;
}

static float sample_glide_next(const float* C, uint32_t* S, float f);

static void proc_process64(int kind, const float* C, uint32_t* S, const float* in, float* out)
{
  float f0, f1;
  switch (kind)
  {
    case MLGPU_PROC_PHASOR_GEN: phasor64(&S[0], in, out); break;
    case MLGPU_PROC_SINE_GEN: /* :380 */
      phasor64(&S[0], in, out);
      for (int n = 0; n < VEC; ++n) out[n] = phasor_to_sine(out[n]);
      break;
    case MLGPU_PROC_SAW_GEN: /* :401 */
    {
      float ph[VEC];
      phasor64(&S[0], in, ph);
      for (int n = 0; n < VEC; ++n) out[n] = phasor_to_saw(ph[n], in[n]);
      break;
    }
    case MLGPU_PROC_PULSE_GEN: /* :390-393 */
    {
      float ph[VEC];
      phasor64(&S[0], in, ph);
      for (int n = 0; n < VEC; ++n) out[n] = phasor_to_pulse(ph[n], in[n], C[0]);
      break;
    }
    case MLGPU_PROC_NOISE_GEN: noise64(&S[0], out); break;
    case MLGPU_PROC_TICK_GEN:
      f0 = u2f(S[0]);
      tick64(&f0, in, out);
      S[0] = f2u(f0);
      break;
    case MLGPU_PROC_IMPULSE_GEN:
      f0 = u2f(S[0]);
      impulse64(&f0, (int32_t*)&S[1], in, out);
      S[0] = f2u(f0);
      break;
    case MLGPU_PROC_ONE_SHOT_GEN: oneshot64(&S[0], &S[1], &S[2], in, out); break;
    case MLGPU_PROC_TEST_SINE_GEN: /* TestSineGen::operator(), MLDSPGens.h:158-170; sinf = the libm restated below */
      f0 = u2f(S[0]);
      for (int n = 0; n < VEC; ++n)
      {
        const float step = 6.2831853071795864769252867f * in[n]; /* ml::kTwoPi, MLDSPScalarMath.h:23 */
        f0 += step;
        if (f0 > 6.2831853071795864769252867f) f0 -= 6.2831853071795864769252867f;
        out[n] = mlorc_libm_sinf(f0);
      }
      S[0] = f2u(f0);
      break;
    case MLGPU_PROC_LOPASS: case MLGPU_PROC_HIPASS: case MLGPU_PROC_BANDPASS:
      f0 = u2f(S[0]); f1 = u2f(S[1]);
      svf64(kind, C, &f0, &f1, in, out);
      S[0] = f2u(f0); S[1] = f2u(f1);
      break;
    case MLGPU_PROC_LO_SHELF: case MLGPU_PROC_HI_SHELF: case MLGPU_PROC_BELL:
      f0 = u2f(S[0]); f1 = u2f(S[1]);
      shelf64(kind, C, &f0, &f1, in, out);
      S[0] = f2u(f0); S[1] = f2u(f1);
      break;
    case MLGPU_PROC_ONE_POLE:
      f0 = u2f(S[0]);
      onepole64(C, &f0, in, out);
      S[0] = f2u(f0);
      break;
    case MLGPU_PROC_DC_BLOCKER:
      f0 = u2f(S[0]); f1 = u2f(S[1]);
      dcblock64(C, &f0, &f1, in, out);
      S[0] = f2u(f0); S[1] = f2u(f1);
      break;
    case MLGPU_PROC_DIFFERENTIATOR:
      f0 = u2f(S[0]);
      diff64(&f0, in, out);
      S[0] = f2u(f0);
      break;
    case MLGPU_PROC_INTEGRATOR:
      f0 = u2f(S[0]);
      integ64(C, &f0, in, out);
      S[0] = f2u(f0);
      break;
    case MLGPU_PROC_PEAK:
      f0 = u2f(S[0]);
      peak64(C, &f0, (int32_t*)&S[1], in, out);
      S[0] = f2u(f0);
      break;
    case MLGPU_PROC_RMS:
      f0 = u2f(S[0]);
      rms64(C, &f0, in, out);
      S[0] = f2u(f0);
      break;
    case MLGPU_PROC_ADSR:
      for (int n = 0; n < VEC; ++n) out[n] = adsr_sample(C, S, in[n]);
      break;
    case MLGPU_PROC_GAIN: /* x * DSPVector(gain), MLDSPOps.h:157,345-348 */
      for (int n = 0; n < VEC; ++n) out[n] = in[n] * C[0];
      break;
    case MLGPU_PROC_SAMPLE_ACCURATE_LINEAR_GLIDE:
      for (int n = 0; n < VEC; ++n) out[n] = sample_glide_next(C, S, in[n]);
      break;
    case MLGPU_PROC_ALLPASS1: /* MLDSPFilters.h:945-953: y = x1 + (x - y1)*coeffs */
      f0 = u2f(S[0]); f1 = u2f(S[1]);
      for (int n = 0; n < VEC; ++n)
      {
        const float y = f0 + (in[n] - f1) * C[0];
        f0 = in[n];
        f1 = y;
        out[n] = y;
      }
      S[0] = f2u(f0); S[1] = f2u(f1);
      break;
    default: break;
  }
}

/* SampleAccurateLinearGlide::nextSample, MLDSPGens.h:541-580.
 * C{samplesPerGlide:i32, dyPerSample}  S{curr, step, target, samplesRemaining:i32} */
static float sample_glide_next(const float* C, uint32_t* S, float f)
{
  const int32_t perGlide = (int32_t)f2u(C[0]);
  const float dyPerSample = C[1];
  float curr = u2f(S[0]), step = u2f(S[1]), target = u2f(S[2]);
  int32_t remaining = (int32_t)S[3];
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
  S[0] = f2u(curr); S[1] = f2u(step); S[2] = f2u(target); S[3] = (uint32_t)remaining;
  return curr;
}

typedef struct
{
  const int32_t* procs;
  int n_procs;
  size_t V, T, v0, v1;
  const float* coeffs;
  uint32_t* state;
  const float* in_signal;
  const float* in_const;
  float* out;
} chain_job;

static void* chain_worker(void* arg)
{
  chain_job* j = (chain_job*)arg;
  int cOff[64], sOff[64], c = 0, s = 0;
  for (int p = 0; p < j->n_procs; ++p)
  {
    cOff[p] = c;
    sOff[p] = s;
    c += mlorc_proc_num_coeffs(j->procs[p]);
    s += mlorc_proc_num_state(j->procs[p]);
  }
  const size_t S = j->T * VEC;
  float C[64][8];
  uint32_t St[64][8];
  for (size_t v = j->v0; v < j->v1; ++v)
  {
    for (int p = 0; p < j->n_procs; ++p)
    {
      int nc = mlorc_proc_num_coeffs(j->procs[p]), ns = mlorc_proc_num_state(j->procs[p]);
      for (int i = 0; i < nc; ++i) C[p][i] = j->coeffs[(size_t)(cOff[p] + i) * j->V + v];
      for (int i = 0; i < ns; ++i) St[p][i] = j->state[(size_t)(sOff[p] + i) * j->V + v];
    }
    for (size_t t = 0; t < j->T; ++t)
    {
      float x[VEC], y[VEC];
      if (j->in_signal)
        memcpy(x, j->in_signal + v * S + t * VEC, sizeof(x));
      else
        for (int n = 0; n < VEC; ++n) x[n] = j->in_const ? j->in_const[v] : 0.f;
      for (int p = 0; p < j->n_procs; ++p)
      {
        proc_process64(j->procs[p], C[p], St[p], x, y);
        memcpy(x, y, sizeof(x));
      }
      if (j->out) memcpy(j->out + v * S + t * VEC, x, sizeof(x));
    }
    for (int p = 0; p < j->n_procs; ++p)
    {
      int ns = mlorc_proc_num_state(j->procs[p]);
      for (int i = 0; i < ns; ++i) j->state[(size_t)(sOff[p] + i) * j->V + v] = St[p][i];
    }
  }
  return NULL;
}

/* same contract as mlref_chain_process (oracle/ref_wrapper.cpp):
 * coeffs [totalNC][V], state [totalNS][V] in/out, in_signal [V][64T] or NULL,
 * in_const [V] or NULL, out [V][64T] or NULL. */
int mlorc_chain_process(const int32_t* procs, int n_procs, size_t V, size_t T, const float* coeffs,
                        uint32_t* state, const float* in_signal, const float* in_const, float* out,
                        int n_threads)
{
  if (n_procs < 1 || n_procs > 64) return MLGPU_ERR_INVALID;
  for (int p = 0; p < n_procs; ++p)
    if (mlorc_proc_num_coeffs(procs[p]) < 0) return MLGPU_ERR_INVALID;
  if (!g_impulse_table_ready) build_impulse_table();
  if (n_threads < 1) n_threads = 1;
  if (n_threads > 256) n_threads = 256;
  chain_job jobs[256];
  pthread_t th[256];
  size_t per = (V + (size_t)n_threads - 1) / (size_t)n_threads;
  int used = 0;
  for (int i = 0; i < n_threads; ++i)
  {
    size_t a = per * (size_t)i, b = per * (size_t)(i + 1);
    if (a > V) a = V;
    if (b > V) b = V;
    if (a >= b) continue;
    chain_job j = {procs, n_procs, V, T, a, b, coeffs, state, in_signal, in_const, out};
    jobs[used] = j;
    if (n_threads == 1)
      chain_worker(&jobs[used]);
    else
      pthread_create(&th[used], NULL, chain_worker, &jobs[used]);
    used++;
  }
  if (n_threads > 1)
    for (int i = 0; i < used; ++i) pthread_join(th[i], NULL);
  return MLGPU_OK;
}

/* PulseGen with an audio-rate width input, PulseGen::operator()(freq, width) MLDSPGens.h:390-393.
 * omega32 [V] in/out, freq / width / out [V][64T]. */
int mlorc_pulse2_process(size_t V, size_t T, uint32_t* omega32, const float* freq, const float* width, float* out)
{
  const size_t S = T * VEC;
  for (size_t v = 0; v < V; ++v)
    for (size_t t = 0; t < T; ++t)
    {
      float ph[VEC];
      const float* f = freq + v * S + t * VEC;
      const float* w = width + v * S + t * VEC;
      phasor64(&omega32[v], f, ph);
      for (int n = 0; n < VEC; ++n) out[v * S + t * VEC + n] = phasor_to_pulse(ph[n], f[n], w[n]);
    }
  return MLGPU_OK;
}

/* wall-clock seconds of one mlorc_chain_process call (CPU baseline, kind "port") */
double mlorc_chain_time(const int32_t* procs, int n_procs, size_t V, size_t T, const float* coeffs,
                        uint32_t* state, const float* in_signal, const float* in_const, float* out,
                        int n_threads)
{
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  mlorc_chain_process(procs, n_procs, V, T, coeffs, state, in_signal, in_const, out, n_threads);
  clock_gettime(CLOCK_MONOTONIC, &t1);
  return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* ------------------------------------------------------------------------- */
/* coefficient makers (host libm), source/DSP/MLDSPFilters.h                  */

static void svf_coeffs(float omega, float k, float* o) /* :85-95 == :168-177 == :212-221 */
{
  float piOmega = kPiF * omega;
  float s1 = sinf(piOmega);
  float s2 = sinf(2.0f * piOmega);
  float nrm = 1.0f / (2.f + k * s2);
  o[0] = s2 * nrm;
  o[1] = (-2.f * s1 * s1 - k * s2) * nrm;
  o[2] = (2.0f * s1 * s1) * nrm;
}
void mlorc_lopass_make_coeffs(float omega, float k, float* o) { svf_coeffs(omega, k, o); }
void mlorc_bandpass_make_coeffs(float omega, float k, float* o) { svf_coeffs(omega, k, o); }
void mlorc_hipass_make_coeffs(float omega, float k, float* o)
{
  svf_coeffs(omega, k, o);
  o[3] = k;
}
void mlorc_loshelf_make_coeffs(float omega, float k, float A, float* r) /* :270-281 */
{
  float piOmega = kPiF * omega;
  float g = tanf(piOmega) / sqrtf(A);
  r[0] = 1.f / (1.f + g * (g + k));
  r[1] = g * r[0];
  r[2] = g * r[1];
  r[3] = k * (A - 1.f);
  r[4] = (A * A - 1.f);
}
void mlorc_hishelf_make_coeffs(float omega, float k, float A, float* r) /* :350-362 */
{
  float piOmega = kPiF * omega;
  float g = tanf(piOmega) * sqrtf(A);
  r[0] = 1.f / (1.f + g * (g + k));
  r[1] = g * r[0];
  r[2] = g * r[1];
  r[3] = A * A;
  r[4] = k * (1.f - A) * A;
  r[5] = (1.f - A * A);
}
void mlorc_bell_make_coeffs(float omega, float k, float A, float* r) /* :415-425 */
{
  float kc = k / A;
  float piOmega = kPiF * omega;
  float g = tanf(piOmega);
  float a1 = 1.f / (1.f + g * (g + kc));
  float a2 = g * a1;
  float a3 = g * a2;
  float m1 = kc * (A * A - 1.f);
  r[0] = a1; r[1] = a2; r[2] = a3; r[3] = m1;
}
void mlorc_onepole_make_coeffs(float omega, float* o) /* :458-462 */
{
  float x = expf(-omega * kTwoPiF);
  o[0] = 1.f - x;
  o[1] = x;
}
float mlorc_dcblocker_make_coeffs(float omega) { return cosf(omega); } /* :498 */
void mlorc_adsr_calc_coeffs(float a, float d, float s, float r, float sr, float* o) /* :679-686 */
{
  const float minSegmentTime = 0.0002f;
  const float invSr = 1.0f / sr;
  o[0] = kTwoPiF * invSr / ((a > minSegmentTime) ? a : minSegmentTime);
  o[1] = kTwoPiF * invSr / ((d > minSegmentTime) ? d : minSegmentTime);
  o[2] = s;
  o[3] = kTwoPiF * invSr / ((r > minSegmentTime) ? r : minSegmentTime);
}
float mlorc_db_to_gain(float dB) { return powf(10.f, dB / 40.f); } /* :30 */

/* rangeClosed / rangeOpen, MLDSPOps.h:967-980: columnIndex()*interval + start */
void mlorc_range_closed(float start, float end, float* out64)
{
  float interval = (end - start) / (VEC - 1.f);
  for (int i = 0; i < VEC; ++i) out64[i] = (float)i * interval + start;
}
void mlorc_range_open(float start, float end, float* out64)
{
  float interval = (end - start) / (VEC);
  for (int i = 0; i < VEC; ++i) out64[i] = (float)i * interval + start;
}

/* ------------------------------------------------------------------------- */
/* host libm sinf, restated                                                   */
/*
 * Lopass::makeCoeffsVec (MLDSPFilters.h:97-115) calls sinf PER SAMPLE, so for that form the reference's
 * arithmetic includes a third-party dependency that is not under /root/reference: glibc 2.35 libm
 * (Ubuntu GLIBC 2.35-0ubuntu3.11 in this image), sysdeps/ieee754/flt-32/s_sinf.c — the ARM
 * optimized-routines sinf (Szabolcs Nagy, Wilco Dijkstra): range reduction by pi/2 in double
 * (fast path |x| < 120; 192 bits of 4/pi above), then a degree-7 sine or degree-8 cosine minimax polynomial
 * in double, rounded once to float. Constants: __sincosf_table and __inv_pio4 of that release.
 * Pinned: mlorc_sinf_check compares it with the host libm's sinf over any range of bit patterns;
 * tests/test_oracle_golden.py runs it over ALL 2^32 inputs (identical except 12 arguments with
 * 53 < |x| < 120 where glibc's x86-64 FMA ifunc variant rounds differently; none for |x| <= pi,
 * the only range the SVF coefficient code uses).
 */
static float sinf_poly_restated(double x, double x2, int neg, int n)
{
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

float mlorc_libm_sinf(float y)
{
  static const uint32_t inv_pio4[24] = {0xa2, 0xa2f9, 0xa2f983, 0xa2f9836e, 0xf9836e4e, 0x836e4e44, 0x6e4e4415, 0x4e441529,
                                        0x441529fc, 0x1529fc27, 0x29fc2757, 0xfc2757d1, 0x2757d1f5, 0x57d1f534, 0xd1f534dd, 0xf534ddc0,
                                        0x34ddc0db, 0xddc0db62, 0xc0db6295, 0xdb629599, 0x6295993c, 0x95993c43, 0x993c4390, 0x3c439041};
  const uint32_t top = (f2u(y) >> 20) & 0x7ffu;
  double x = (double)y;
  if (top < 0x3f4u) /* |y| < pi/4 */
  {
    if (top < 0x398u) return y; /* |y| < 2^-12 */
    return sinf_poly_restated(x, x * x, 0, 0);
  }
  if (top < 0x42fu) /* |y| < 120 */
  {
    const double r = x * 0x1.45F306DC9C883p+23;
    const int n = ((int32_t)r + 0x800000) >> 24;
    x = x - (double)n * 0x1.921FB54442D18p0;
    const double sgn = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
    return sinf_poly_restated(x * sgn, x * x, (n & 2) != 0, n);
  }
  if (top < 0x7f8u)
  {
    uint32_t xi = f2u(y);
    const int sign = (int)(xi >> 31);
    const uint32_t* arr = &inv_pio4[(xi >> 26) & 15];
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
    return sinf_poly_restated(x * sgn, x * x, ((n + sign) & 2) != 0, n);
  }
  return (y - y) / (y - y);
}

/* count bit patterns u in [lo, hi] (both signs are covered by the caller's range) where the restatement and the
 * host libm disagree (any NaN == any NaN); up to `max_list` offending patterns are written to `list`. */
typedef struct { uint32_t lo, hi; uint64_t bad; uint32_t* list; int max_list, n_list; } sinf_job;
static void* sinf_worker(void* arg)
{
  sinf_job* j = (sinf_job*)arg;
  for (uint64_t u = j->lo; u <= j->hi; ++u)
  {
    const float x = u2f((uint32_t)u);
    const float a = sinf(x), b = mlorc_libm_sinf(x);
    if (f2u(a) != f2u(b) && !(a != a && b != b))
    {
      if (j->n_list < j->max_list) j->list[j->n_list++] = (uint32_t)u;
      j->bad++;
    }
  }
  return NULL;
}
uint64_t mlorc_sinf_check(uint32_t lo, uint32_t hi, int n_threads, uint32_t* list, int max_list)
{
  if (n_threads < 1) n_threads = 1;
  if (n_threads > 64) n_threads = 64;
  sinf_job jobs[64];
  pthread_t th[64];
  uint32_t lists[64][32];
  const uint64_t span = (uint64_t)hi - lo + 1, per = (span + n_threads - 1) / n_threads;
  int used = 0;
  for (int i = 0; i < n_threads; ++i)
  {
    const uint64_t a = lo + per * i, b = a + per - 1;
    if (a > hi) break;
    sinf_job j = {(uint32_t)a, (uint32_t)(b > hi ? hi : b), 0, lists[i], 32, 0};
    jobs[used] = j;
    pthread_create(&th[used], NULL, sinf_worker, &jobs[used]);
    used++;
  }
  uint64_t bad = 0;
  int nl = 0;
  for (int i = 0; i < used; ++i)
  {
    pthread_join(th[i], NULL);
    bad += jobs[i].bad;
    for (int k = 0; k < jobs[i].n_list && nl < max_list; ++k) list[nl++] = jobs[i].list[k];
  }
  return bad;
}

/* The device's CHEAPER forms of the same function on the SVF coefficient code's domain [2^-12, pi_f] (mldsp_math.hpp:
 * libm_sinf_q0 for [2^-12, kSinfT1) / libm_sinf_pair for [2^-12, pi_f]), restated here operation for operation - Horner polynomials with fused multiply-adds, the
 * quadrant from two float comparisons - so that their claim can be checked where the host libm lives: mlorc_sinf_fast_check
 * compares both with sinf() over a range of bit patterns (tests/test_oracle_golden.py runs the whole domain, 113 840 092 floats).
 * fma() is the C library's correctly rounded one (hardware FMA through glibc's ifunc where the CPU has it). */
static double fast_sin_poly(double x, double x2)
{
  double p = fma(x2, -0x1.994eb3774cf24p-13, 0x1.1107605230bc4p-7);
  p = fma(x2, p, -0x1.555545995a603p-3);
  return fma(x * x2, p, x);
}
static double fast_cos_poly(double x2)
{
  double p = fma(x2, 0x1.99343027bf8c3p-16, -0x1.6c087e89a359dp-10);
  p = fma(x2, p, 0x1.55553e1068f19p-5);
  p = fma(x2, p, -0x1.ffffffd0c621cp-2);
  return fma(x2, p, 0x1p0);
}
float mlorc_sinf_direct(float y) /* 2^-12 <= y < 0.75 */
{
  const double x = (double)y;
  return (float)fast_sin_poly(x, x * x);
}
float mlorc_sinf_0_pi(float y) /* 2^-12 <= y <= pi_f */
{
  const int q1 = (y >= 0x1.921fb6p-1f), q2 = (y >= 0x1.2d97c8p+1f);
  const double k = q2 ? 2.0 * 0x1.921FB54442D18p0 : (q1 ? 0x1.921FB54442D18p0 : 0.0);
  const double x = (double)y - k, x2 = x * x;
  const float sn = (float)fast_sin_poly(x, x2), cs = (float)fast_cos_poly(x2);
  return q2 ? -sn : (q1 ? cs : sn);
}
/* which = 0: mlorc_sinf_direct, 1: mlorc_sinf_0_pi; mismatches against the host libm over bit patterns [lo, hi] */
typedef struct { uint32_t lo, hi; int which; uint64_t bad; uint32_t first; } sinf_fast_job;
static void* sinf_fast_worker(void* arg)
{
  sinf_fast_job* j = (sinf_fast_job*)arg;
  for (uint64_t u = j->lo; u <= j->hi; ++u)
  {
    const float x = u2f((uint32_t)u);
    const float a = sinf(x), b = j->which ? mlorc_sinf_0_pi(x) : mlorc_sinf_direct(x);
    if (f2u(a) != f2u(b))
    {
      if (!j->bad) j->first = (uint32_t)u;
      j->bad++;
    }
  }
  return NULL;
}
uint64_t mlorc_sinf_fast_check(int which, uint32_t lo, uint32_t hi, int n_threads, uint32_t* first_bad)
{
  if (n_threads < 1) n_threads = 1;
  if (n_threads > 64) n_threads = 64;
  sinf_fast_job jobs[64];
  pthread_t th[64];
  const uint64_t span = (uint64_t)hi - lo + 1, per = (span + n_threads - 1) / n_threads;
  int used = 0;
  for (int i = 0; i < n_threads; ++i)
  {
    const uint64_t a = lo + per * i, b = a + per - 1;
    if (a > hi) break;
    sinf_fast_job j = {(uint32_t)a, (uint32_t)(b > hi ? hi : b), which, 0, 0};
    jobs[used] = j;
    pthread_create(&th[used], NULL, sinf_fast_worker, &jobs[used]);
    used++;
  }
  uint64_t bad = 0;
  for (int i = 0; i < used; ++i)
  {
    pthread_join(th[i], NULL);
    if (jobs[i].bad && !bad && first_bad) *first_bad = jobs[i].first;
    bad += jobs[i].bad;
  }
  return bad;
}
/* the quadrant glibc computes for 0.75 <= y < 120: ((int32)(y * 2/pi * 2^24) + 2^23) >> 24 */
int mlorc_sinf_quadrant(float y)
{
  const double r = (double)y * 0x1.45F306DC9C883p+23;
  return ((int32_t)r + 0x800000) >> 24;
}

/* ------------------------------------------------------------------------- */
/* the other operator() forms and the control-rate processors (graph nodes)   */
/*
 * One processor, V voices, T vectors, n_inputs audio-rate input signals each [V][64T] (a control-rate value is
 * passed repeated 64 times; vector-rate processors read sample 0 of each vector). coeffs [NC][V], state [NS][V]
 * in/out. Forms:
 *   PULSE_GEN 2: (freq, width)                    MLDSPGens.h:390-393
 *   LOPASS 3: (x, omega, k)                       MLDSPFilters.h:136-152 with makeCoeffsVec :97-115
 *   LO_SHELF 6: (x, a1,a2,a3,m1,m2)               :304-318
 *   HI_SHELF 7: (x, a1,a2,a3,m0,m1,m2)            :385-399
 *   INTERPOLATOR1 1, LINEAR_GLIDE 1 (one float per vector)   MLDSPGens.h:412-423, 433-515
 *   anything with 1 input: same as mlorc_chain_process of that one processor
 */
int mlorc_proc_process_multi(int kind, size_t V, size_t T, const float* coeffs, uint32_t* state,
                             const float* const* inputs, int n_inputs, float* out)
{
  const int nc = mlorc_proc_num_coeffs(kind), ns = mlorc_proc_num_state(kind);
  if (nc < 0 || n_inputs < 0 || n_inputs > 8) return MLGPU_ERR_INVALID;
  if (!g_impulse_table_ready) build_impulse_table();
  const size_t S = T * VEC;
  float C[8];
  uint32_t St[80];
  for (size_t v = 0; v < V; ++v)
  {
    for (int i = 0; i < nc; ++i) C[i] = coeffs[(size_t)i * V + v];
    for (int i = 0; i < ns; ++i) St[i] = state[(size_t)i * V + v];
    for (size_t t = 0; t < T; ++t)
    {
      const float* in[8];
      for (int i = 0; i < n_inputs; ++i) in[i] = inputs[i] + v * S + t * VEC;
      float* y = out + v * S + t * VEC;
      if (kind == MLGPU_PROC_PULSE_GEN && n_inputs == 2)
      {
        float ph[VEC];
        phasor64(&St[0], in[0], ph);
        for (int n = 0; n < VEC; ++n) y[n] = phasor_to_pulse(ph[n], in[0][n], in[1][n]);
      }
      else if (kind == MLGPU_PROC_LOPASS && n_inputs == 3)
      {
        float ic1 = u2f(St[0]), ic2 = u2f(St[1]);
        for (int n = 0; n < VEC; ++n)
        {
          const float omega = sse_min(in[1][n], 0.5f);
          const float k = sse_max(in[2][n], 0.01f);
          const float piOmega = 3.1415926535897932384626433832795f * omega;
          const float s1 = sinf(piOmega); /* the HOST libm, exactly what the reference calls */
          const float s2 = sinf(2.0f * piOmega);
          const float nrm = 1.0f / (2.f + k * s2);
          const float g0 = s2 * nrm;
          const float g1 = (-2.f * s1 * s1 - k * s2) * nrm;
          const float g2 = (2.0f * s1 * s1) * nrm;
          const float v0 = in[0][n];
          const float t0 = v0 - ic2;
          const float t1 = g0 * t0 + g1 * ic1;
          const float t2 = g2 * t0 + g0 * ic1;
          const float v2 = t2 + ic2;
          ic1 += 2.0f * t1;
          ic2 += 2.0f * t2;
          y[n] = v2;
        }
        St[0] = f2u(ic1); St[1] = f2u(ic2);
      }
      else if ((kind == MLGPU_PROC_LO_SHELF && n_inputs == 6) || (kind == MLGPU_PROC_HI_SHELF && n_inputs == 7))
      {
        float ic1 = u2f(St[0]), ic2 = u2f(St[1]);
        for (int n = 0; n < VEC; ++n)
        {
          const float v0 = in[0][n];
          const float a1 = in[1][n], a2 = in[2][n], a3 = in[3][n];
          const float v3 = v0 - ic2;
          const float v1 = a1 * ic1 + a2 * v3;
          const float v2 = ic2 + a2 * ic1 + a3 * v3;
          ic1 = 2 * v1 - ic1;
          ic2 = 2 * v2 - ic2;
          if (kind == MLGPU_PROC_LO_SHELF)
            y[n] = v0 + in[4][n] * v1 + in[5][n] * v2;
          else
            y[n] = in[4][n] * v0 + in[5][n] * v1 + in[6][n] * v2;
        }
        St[0] = f2u(ic1); St[1] = f2u(ic2);
      }
      else if (kind == MLGPU_PROC_INTERPOLATOR1 && n_inputs == 1)
      {
        const float f = in[0][0], cur = u2f(St[0]);
        const float dydt = f - cur;
        for (int n = 0; n < VEC; ++n) y[n] = cur + ((float)(n + 1) / (float)VEC) * dydt; /* kUnityRampVec :409-410 */
        St[0] = f2u(f);
      }
      else if (kind == MLGPU_PROC_LINEAR_GLIDE && n_inputs == 1)
      {
        /* C{vectorsPerGlide:i32, dyPerVector}  S{target, step, vectorsRemaining:i32, currVec[64]} */
        const float f = in[0][0];
        const int32_t perGlide = (int32_t)f2u(C[0]);
        float target = u2f(St[0]), step = u2f(St[1]);
        int32_t remaining = (int32_t)St[2];
        uint32_t* cur = &St[3];
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
          for (int n = 0; n < VEC; ++n) cur[n] = f2u(target);
          step = 0.f;
          remaining--;
        }
        else if (remaining == perGlide)
        {
          const float currentValue = u2f(cur[VEC - 1]);
          const float dydv = (target - currentValue) * C[1];
          step = dydv;
          for (int n = 0; n < VEC; ++n) cur[n] = f2u(currentValue + ((float)(n + 1) / (float)VEC) * step);
          remaining--;
        }
        else
        {
          for (int n = 0; n < VEC; ++n) cur[n] = f2u(u2f(cur[n]) + step);
          remaining--;
        }
        for (int n = 0; n < VEC; ++n) y[n] = u2f(cur[n]);
        St[0] = f2u(target); St[1] = f2u(step); St[2] = (uint32_t)remaining;
      }
      else if (kind == MLGPU_PROC_TEMPO_LOCK && n_inputs == 3)
      { /* MLDSPFilters.h:1492-1578; in[0] the phasor to follow, in[1][0] = dydx, in[2][0] = isr */
        float omega = u2f(St[0]), x1v = u2f(St[1]);
        const float x0 = in[0][0], dydx = in[1][0], isr = in[2][0];
        float dxdt = 0.f, dydt = 0.f;
        if (x0 == -1.0f)
        {
          omega = -1.0f;
          for (int n = 0; n < VEC; ++n) y[n] = 0.f;
        }
        else
        {
          if (omega > -1.f)
          {
            float dx = x0 - x1v;
            if (dx < 0.f) dx += 1.f;
            dxdt = dx / VEC;
            dydt = dxdt * dydx;
            x1v = x0;
          }
          else
          {
            dxdt = in[0][1] - x0;
            dydt = dxdt * dydx;
            x1v = x0 - dxdt * VEC;
            omega = fmodf(x0 * dydx, 1.0f);
          }
          int lock = 0;
          const float lockDist = 0.001f;
          if (fabsf(dydx - roundf(dydx)) < lockDist) lock = 1;
          const float rdydx = 1.0f / dydx;
          if (fabsf(rdydx - roundf(rdydx)) < lockDist) lock = 1;
          if (lock)
          {
            float ref, refWrap, error;
            if (dydx >= 1.f)
            {
              ref = x0 * dydx;
              refWrap = ref - floorf(ref);
              error = omega - refWrap;
            }
            else
            {
              ref = omega / dydx;
              refWrap = ref - floorf(ref);
              error = refWrap - x0;
            }
            const float errorDiff = roundf(error) - error;
            float correction = errorDiff * isr * 4.0f;
            const float lo = -dydt * 0.5f, hi = dydt * 1.0f;
            correction = (correction < lo) ? lo : (correction > hi ? hi : correction);
            dydt += correction;
          }
          for (int n = 0; n < VEC; ++n)
          {
            y[n] = omega;
            omega += dydt;
            if (omega > 1.0f) omega -= 1.0f;
          }
        }
        St[0] = f2u(omega); St[1] = f2u(x1v);
      }
      else if (n_inputs <= 1 && kind != MLGPU_PROC_INTERPOLATOR1 && kind != MLGPU_PROC_LINEAR_GLIDE)
      {
        float zero[VEC] = {0};
        proc_process64(kind, C, St, n_inputs ? in[0] : zero, y);
      }
      else
        return MLGPU_ERR_UNSUPPORTED;
    }
    for (int i = 0; i < ns; ++i) state[(size_t)i * V + v] = St[i];
  }
  return MLGPU_OK;
}

/* index-dependent generators over a whole signal: a, b are [V][64T] (sample 0 of each vector is the float
 * argument) or NULL; MLDSPOps.h:962-990 */
int mlorc_vop(int vop, size_t V, size_t T, const float* a, const float* b, float* out)
{
  for (size_t r = 0; r < V * T; ++r)
  {
    const float start = a ? a[r * VEC] : 0.f, end = b ? b[r * VEC] : 0.f;
    float* y = out + r * VEC;
    float interval;
    switch (vop)
    {
      case MLGPU_VOP_COLUMN_INDEX:
        for (int i = 0; i < VEC; ++i) y[i] = (float)i;
        break;
      case MLGPU_VOP_RANGE_OPEN: mlorc_range_open(start, end, y); break;
      case MLGPU_VOP_RANGE_CLOSED: mlorc_range_closed(start, end, y); break;
      case MLGPU_VOP_INTERPOLATE_LINEAR: /* :986-990 */
        interval = (end - start) / (VEC);
        for (int i = 0; i < VEC; ++i) y[i] = (float)i * interval + (start + interval);
        break;
      default: return MLGPU_ERR_INVALID;
    }
  }
  return MLGPU_OK;
}

void mlorc_linear_glide_make_coeffs(float t, float* o) /* MLDSPGens.h:444-449 */
{
  int32_t n = (int32_t)(t / VEC);
  if (n < 1) n = 1;
  memcpy(&o[0], &n, 4);
  o[1] = 1.0f / (n + 0.f);
}
void mlorc_sample_accurate_linear_glide_make_coeffs(float t, float* o) /* :527-532 */
{
  int32_t n = (int32_t)t;
  if (n < 1) n = 1;
  memcpy(&o[0], &n, 4);
  o[1] = 1.0f / n;
}

/* ------------------------------------------------------------------------- */
/* row plumbing (MLDSPOps.h:1041-1374) and routing (MLDSPRouting.h:59-234)    */
/*
 * Same contracts as mlgpu_rows_map / rows_add / rows_normalize / rows_index / multiplex / demultiplex in
 * include/mlgpu.h. The reference's functions are templates over compile-time row counts; tests compose these
 * rule-based calls exactly as the host would (tests/rows_functions.py) and compare with the compiled
 * reference's own repeatRows / stretchRows / ... instantiations (oracle/ref_wrapper.cpp: mlref_rows_case).
 */
static long rows_source(int rule, long p0, long p1, long N, long count, long j)
{
  long k;
  switch (rule)
  {
    case MLGPU_ROWS_REPEAT: return j % N;                                          /* :1062-1066 */
    case MLGPU_ROWS_STRETCH:                                                       /* :1080 */
      return (count < 2) ? 0 : (long)roundf(((float)j * ((float)N - 1.f)) / ((float)count - 1.f));
    case MLGPU_ROWS_SHIFT: k = j - p0; return (k >= 0 && k < N) ? k : -1;          /* :1107-1118, :1094-1098 */
    case MLGPU_ROWS_ROTATE: k = (j - p0) % N; return k < 0 ? k + N : k;            /* :1132-1136 */
    default: k = p0 + j * p1; return (k >= 0 && k < N) ? k : -1;                   /* separate/even/odd/concat/shuffle */
  }
}

int mlorc_rows_map(int rule, long p0, long p1, int sample_rotate, const float* src, size_t src_rows, float* dst,
                   size_t dst_rows, size_t dst_offset, size_t dst_step, size_t count, size_t groups)
{
  if (rule < MLGPU_ROWS_REPEAT || rule > MLGPU_ROWS_STRIDED || src_rows == 0) return MLGPU_ERR_INVALID;
  for (size_t g = 0; g < groups; ++g)
    for (size_t j = 0; j < count; ++j)
    {
      const long s = rows_source(rule, p0, p1, (long)src_rows, (long)count, (long)j);
      float* y = dst + (g * dst_rows + dst_offset + j * dst_step) * VEC;
      if (s < 0)
      {
        for (int n = 0; n < VEC; ++n) y[n] = 0.f;
        continue;
      }
      const float* x = src + (g * src_rows + (size_t)s) * VEC;
      /* rotateLeft: vecShuffleLeft(v1, v2) = {v1[1], v1[2], v1[3], v2[0]} :1219-1245; rotateRight the mirror image */
      for (int n = 0; n < VEC; ++n) y[n] = x[(n + sample_rotate + VEC) % VEC];
    }
  return MLGPU_OK;
}

int mlorc_rows_add(const float* rows, size_t rows_per_group, float* out, size_t groups) /* :1349-1359 */
{
  for (size_t g = 0; g < groups; ++g)
    for (int n = 0; n < VEC; ++n)
    {
      float acc = 0.f;
      for (size_t j = 0; j < rows_per_group; ++j) acc = acc + rows[(g * rows_per_group + j) * VEC + n];
      out[g * VEC + n] = acc;
    }
  return MLGPU_OK;
}

int mlorc_rows_normalize(const float* rows, float* out, size_t n_rows) /* :1041-1050, sum() :995-1005 */
{
  for (size_t r = 0; r < n_rows; ++r)
  {
    const float* x = rows + r * VEC;
    float sum = 0.f;
    for (int g = 0; g < 16; ++g)
    {
      const float t0 = x[4 * g] + x[4 * g + 2], t1 = x[4 * g + 1] + x[4 * g + 3];
      sum += (t0 + t1);
    }
    for (int n = 0; n < VEC; ++n) out[r * VEC + n] = x[n] / sum;
  }
  return MLGPU_OK;
}

int mlorc_rows_index(float* out, size_t rows_per_group, size_t groups) /* :1365-1374 */
{
  for (size_t g = 0; g < groups; ++g)
    for (size_t j = 0; j < rows_per_group; ++j)
      for (int n = 0; n < VEC; ++n) out[(g * rows_per_group + j) * VEC + n] = (float)j;
  return MLGPU_OK;
}

/* the reference converts a float to size_t; negative / NaN selectors are undefined there (documented: index 0) */
static int route_index(float u, int n)
{
  const float r = u * (float)n;
  const int i = (r >= 0.f && r < 2147483648.f) ? (int)r : 0;
  return (i < n) ? i : 0;
}

int mlorc_multiplex(const float* sel, size_t sel_elems, const float* const* ins, int n, float* out, size_t n_elems, int linear)
{
  if (n < 1 || n > 8 || sel_elems == 0) return MLGPU_ERR_INVALID;
  for (size_t i = 0; i < n_elems; ++i)
  {
    const float s = sel[i % sel_elems];
    const float u = s - truncf(s);
    if (!linear)
      out[i] = ins[route_index(u, n)][i]; /* MLDSPRouting.h:94-101 */
    else
    { /* :122-133 */
      const float real = u * (float)n;
      const float ip = truncf(real);
      const float frac = real - ip;
      int i1 = (ip >= 0.f && ip < 2147483648.f) ? (int)ip : 0;
      if (i1 >= n) i1 = 0;
      const int i2 = (i1 + 1) % n;
      const float a = ins[i1][i], b = ins[i2][i];
      out[i] = a + frac * (b - a);
    }
  }
  return MLGPU_OK;
}

int mlorc_demultiplex(const float* sel, size_t sel_elems, const float* in, float* const* outs, int n, size_t n_elems, int linear)
{
  if (n < 1 || n > 8 || sel_elems == 0) return MLGPU_ERR_INVALID;
  for (size_t i = 0; i < n_elems; ++i)
  {
    const float s = sel[i % sel_elems];
    const float u = s - truncf(s);
    if (!linear)
    { /* :151-173 */
      const int idx = route_index(u, n);
      for (int j = 0; j < n; ++j) outs[j][i] = (idx == j) ? in[i] : 0.f;
    }
    else
    { /* :193-233 */
      const float real = u * (float)n;
      const float ip = truncf(real);
      int i1 = (ip >= 0.f && ip < 2147483648.f) ? (int)ip : 0;
      if (i1 >= n) i1 = 0;
      const float m = real - ip;
      const int i2 = (i1 + 1) % n;
      for (int j = 0; j < n; ++j) outs[j][i] = (j == i1) ? in[i] * (1.f - m) : ((j == i2) ? in[i] * m : 0.f);
    }
  }
  return MLGPU_OK;
}

/* mixdown, same contract and the SAME summation order as mlgpu_mixdown (include/mlgpu.h): pairwise tree inside each
 * group of 64 consecutive voices (a[i] += a[i + d], d = 1, 2, ... 32; voices beyond V count as +0); then the group sums
 * left to right 64 consecutive ones at a time, and so the results, until one is left (up to 4096 voices: the groups left to
 * right). The reference has no mixdown function — a Synth accumulates voices with `outputs +=` in voice order
 * (source/app/MLSynth.h:43-57); tests also bound the difference from that sequential order. sig: [V][64T]. */
int mlorc_mixdown(const float* sig, size_t V, size_t T, const float* gains, float* out)
{
  const size_t S = T * VEC, groups = (V + 63) / 64;
  float* rows = (float*)malloc(sizeof(float) * (groups ? groups : 1));
  if (!rows) return MLGPU_ERR_OOM;
  for (size_t s = 0; s < S; ++s)
  {
    for (size_t g = 0; g < groups; ++g)
    {
      float a[64];
      for (int i = 0; i < 64; ++i)
      {
        const size_t v = g * 64 + (size_t)i;
        a[i] = (v < V) ? (gains ? sig[v * S + s] * gains[v] : sig[v * S + s]) : 0.f;
      }
      for (int d = 1; d < 64; d <<= 1)
        for (int i = 0; i + d < 64; i += 2 * d) a[i] = a[i] + a[i + d];
      rows[g] = a[0];
    }
    size_t n = groups;
    do
    {
      const size_t nOut = (n + 63) / 64;
      for (size_t r = 0; r < nOut; ++r)
      {
        const size_t first = r * 64, m = (n - first < 64) ? n - first : 64;
        float acc = rows[first];
        for (size_t g = 1; g < m; ++g) acc = acc + rows[first + g];
        rows[r] = acc;  /* r <= first: never overwrites a row still to be read */
      }
      n = nOut;
    } while (n > 1);
    out[s] = rows[0];
  }
  free(rows);
  return MLGPU_OK;
}

/* The same tree for a bank SPLIT INTO SHARDS (include/mlgpu.h: mlgpu_bank_process_mixdown_shard / mlgpu_mixdown_finish): a shard
 * of Vs voices (a multiple of 64^L, L >= 1 the largest such) hands over its Vs / 64^L level-L sums, the host adds up all shards'
 * rows 64 at a time left to right until one is left. mlorc_mixdown_shard_rows: one shard's rows, [rows][64T], from sig [Vs][64T];
 * mlorc_mixdown_rows: the finish over n_rows rows. shards x (V / shards) voices must give mlorc_mixdown's result for V voices
 * bit for bit (tests/test_oracle_golden.py::test_mixdown_shards_equal_one_bank). */
size_t mlorc_mixdown_shard_rows(size_t Vs)
{
  if (Vs == 0 || Vs % 64) return 0;
  size_t rows = Vs / 64;
  while (rows % 64 == 0) rows /= 64;
  return rows;
}
int mlorc_mixdown_shard(const float* sig, size_t Vs, size_t T, const float* gains, float* rows_out)
{
  const size_t S = T * VEC, nOutRows = mlorc_mixdown_shard_rows(Vs);
  if (nOutRows == 0) return MLGPU_ERR_INVALID;
  const size_t span = Vs / nOutRows; /* voices under one handed-over row: 64^L */
  for (size_t r = 0; r < nOutRows; ++r)
  {
    const int st = mlorc_mixdown(sig + r * span * S, span, T, gains ? gains + r * span : 0, rows_out + r * S);
    if (st != MLGPU_OK) return st;
  }
  return MLGPU_OK;
}
int mlorc_mixdown_rows(const float* rows, size_t n_rows, size_t T, float* out)
{
  const size_t S = T * VEC;
  if (n_rows == 0) return MLGPU_ERR_INVALID;
  float* col = (float*)malloc(sizeof(float) * n_rows);
  if (!col) return MLGPU_ERR_OOM;
  for (size_t s = 0; s < S; ++s)
  {
    for (size_t r = 0; r < n_rows; ++r) col[r] = rows[r * S + s];
    size_t n = n_rows;
    while (n > 1)
    {
      const size_t nOut = (n + 63) / 64;
      for (size_t r = 0; r < nOut; ++r)
      {
        const size_t first = r * 64, m = (n - first < 64) ? n - first : 64;
        float acc = col[first];
        for (size_t g = 1; g < m; ++g) acc = acc + col[first + g];
        col[r] = acc;
      }
      n = nOut;
    }
    out[s] = col[0];
  }
  free(col);
  return MLGPU_OK;
}

/* ------------------------------------------------------------------------- */
/* delay lines, MLDSPFilters.h:799-1106                                       */
/*
 * One delay processor, V voices, T vectors. state [NS][V] in/out; mem [V][rings][len] in/out (len a power of two =
 * the size IntegerDelay::setMaxDelayInSamples allocated); inputs each [V][64T]. Evaluated per sample like
 * processSample (:899-914); tests/test_oracle_vs_ref.py checks it against the reference objects' own operator(),
 * including the block form (:834-875) for constant delays.
 *   INTEGER_DELAY        S{writeIndex, delay:i32}                          forms (x), (x, delay)
 *   FRACTIONAL_DELAY     S{writeIndex, x1, y1, delayInt:i32, allpassCoeff} forms (x), (x, delay), (x, delay, ticks)
 *   PITCHBENDABLE_DELAY  S{delay1[5], delay2[5]}, two rings                form (x, delay)
 */
static float ring_sample(float* ring, uint32_t mask, uint32_t* w, float x, int32_t d)
{
  ring[*w] = x;
  const uint32_t r = (*w - (uint32_t)d) & mask;
  const float y = ring[r];
  *w = (*w + 1) & mask;
  return y;
}
float mlorc_allpass1_make_coeffs(float d) /* :938-943 */
{
  float xm1 = (d - 1.f);
  return -0.53f * xm1 + 0.24f * xm1 * xm1;
}
static void frac_set_delay(uint32_t* S, float d) /* :991-1007 */
{
  float fDelayInt = floorf(d);
  int32_t delayInt = sse_cvtt(fDelayInt);
  float delayFrac = d - fDelayInt;
  if ((delayFrac < 0.618f) && (delayInt > 0))
  {
    delayFrac += 1.f;
    delayInt -= 1;
  }
  S[3] = (uint32_t)delayInt;
  S[4] = f2u(mlorc_allpass1_make_coeffs(delayFrac));
}
void mlorc_fractional_delay_make_state(float d, float* o)
{
  uint32_t S[5];
  frac_set_delay(S, d);
  o[0] = u2f(S[3]);
  o[1] = u2f(S[4]);
}
static float frac_sample(float* ring, uint32_t mask, uint32_t* S, float x)
{
  const float d = ring_sample(ring, mask, &S[0], x, (int32_t)S[3]);
  const float x1 = u2f(S[1]), y1 = u2f(S[2]);
  const float y = x1 + (d - y1) * u2f(S[4]);
  S[1] = f2u(d);
  S[2] = f2u(y);
  return y;
}

int mlorc_delay_process(int kind, size_t V, size_t T, uint32_t* state, float* mem, size_t len, const float* const* inputs, int n_inputs,
                        float* out)
{
  const int ns = mlorc_proc_num_state(kind);
  const int rings = (kind == MLGPU_PROC_PITCHBENDABLE_DELAY) ? 2 : 1;
  if (ns < 0 || len == 0 || (len & (len - 1))) return MLGPU_ERR_INVALID;
  const uint32_t mask = (uint32_t)(len - 1);
  const size_t S = T * VEC;
  uint32_t St[16];
  for (size_t v = 0; v < V; ++v)
  {
    for (int i = 0; i < ns; ++i) St[i] = state[(size_t)i * V + v];
    float* ring = mem + v * (size_t)rings * len;
    for (size_t s = 0; s < S; ++s)
    {
      const float x = inputs[0][v * S + s];
      float y;
      if (kind == MLGPU_PROC_INTEGER_DELAY)
      {
        if (n_inputs == 2) St[1] = (uint32_t)sse_cvtt(inputs[1][v * S + s]); /* static_cast<int>(delay[n]) :887 */
        y = ring_sample(ring, mask, &St[0], x, (int32_t)St[1]);
      }
      else if (kind == MLGPU_PROC_FRACTIONAL_DELAY)
      {
        if (n_inputs == 2 || (n_inputs == 3 && f2u(inputs[2][v * S + s]) != 0u)) frac_set_delay(St, inputs[1][v * S + s]);
        y = frac_sample(ring, mask, St, x);
      }
      else if (kind == MLGPU_PROC_PITCHBENDABLE_DELAY && n_inputs == 2)
      {
        const int r = (int)(s % VEC) % 32;                     /* fadeRamp, :1057 */
        if (r == 16) frac_set_delay(St, inputs[1][v * S + s]);       /* kvDelay1Changes, :1058 */
        if (r == 0) frac_set_delay(St + 5, inputs[1][v * S + s]);    /* kvDelay2Changes, :1059 */
        const float y1 = frac_sample(ring, mask, St, x);
        const float y2 = frac_sample(ring + len, mask, St + 5, x);
        const float fade = 2.f * ((r > 16) ? 1.0f - r / (32 + 0.f) : r / (32 + 0.f)); /* fadeFn, :1060-1065 */
        y = y1 + (fade * (y2 - y1));                          /* lerp, MLDSPOps.h:744 */
      }
      else
        return MLGPU_ERR_UNSUPPORTED;
      out[v * S + s] = y;
    }
    for (int i = 0; i < ns; ++i) state[(size_t)i * V + v] = St[i];
  }
  return MLGPU_OK;
}

/* ------------------------------------------------------------------------- */
/* HalfBandFilter / Downsampler / Upsampler, MLDSPFilters.h:1245-1473         */
/*
 * Same contract as mlgpu_resampler_process: state [octaves*9][V] in/out (per octave: apa0 x1 y1, apa1 x1 y1, apb0 x1 y1,
 * apb1 x1 y1, b1), in [V][64*T_in], out [V][64*T_out]. Written as the stream cascade the block schedule of
 * Downsampler::write (:1349-1387) / Upsampler::write (:1428-1452) amounts to; tests compare with those classes.
 */
static float ap1_step(float* st, float x, float coeff) /* Allpass1::processSample :945-953 */
{
  const float y = st[0] + (x - st[1]) * coeff;
  st[0] = x;
  st[1] = y;
  return y;
}
static float hb_a(float* f, float x) { return ap1_step(f + 2, ap1_step(f + 0, x, 0.07986642623635751f), 0.5453536510711322f); } /* apa0, apa1 :1307 */
static float hb_b(float* f, float x) { return ap1_step(f + 6, ap1_step(f + 4, x, 0.28382934487410993f), 0.8344118914807379f); }  /* apb0, apb1 :1308 */
static float down_rec(float* f, int h, const float* x)
{
  if (h == 0) return x[0];
  const float e = down_rec(f, h - 1, x);
  const float o = down_rec(f, h - 1, x + (1 << (h - 1)));
  float* st = f + (h - 1) * 9;
  const float a0 = hb_a(st, e), b0 = hb_b(st, o);
  const float y = (a0 + st[8]) * 0.5f; /* :1281-1283 */
  st[8] = b0;
  return y;
}
static void up_rec(float* f, int h, int total, float x, float* y)
{
  if (h == 0)
  {
    y[0] = x;
    return;
  }
  float* st = f + (total - h) * 9;
  const float ya = hb_a(st, x), yb = hb_b(st, x); /* :1254-1255 */
  up_rec(f, h - 1, total, ya, y);
  up_rec(f, h - 1, total, yb, y + (1 << (h - 1)));
}
/*
 * Upsample2xFunction<2> / Downsample2xFunction<2> (MLDSPFunctional.h:114-213) around one stateful function, then a gain,
 * written in the reference's own block form (whole DSPVectors, mPhase, mInputBuffer, mOutputBuffer):
 *   fn(v) = Lopass(coeffs)((clamp(v.row(0) * 3, -1, 1) + SawGen(freq)) * v.row(1));   out = F(fn, {x, m}) * 0.5
 * freq[V], x / m / out [V][64*T]; all objects start default-constructed.
 */
typedef struct
{
  uint32_t omega32; /* SawGen's PhasorGen, MLDSPGens.h:179 (mOmega32 = 0) */
  float ic1, ic2;   /* Lopass, MLDSPFilters.h:79 */
  float freq;
  const float* co;
} rate_fn_state;
static void rate_fn(rate_fn_state* st, const float* v0, const float* v1, float* y)
{
  float cps[VEC], ph[VEC], am[VEC];
  for (int n = 0; n < VEC; ++n) cps[n] = st->freq;
  phasor64(&st->omega32, cps, ph);
  for (int n = 0; n < VEC; ++n)
  {
    const float drive = v0[n] * 3.0f;
    const float sat = sse_min(sse_max(drive, -1.0f), 1.0f); /* clamp, MLDSPOps.h:747 */
    const float mix = sat + phasor_to_saw(ph[n], cps[n]);
    am[n] = mix * v1[n];
  }
  svf64(MLGPU_PROC_LOPASS, st->co, &st->ic1, &st->ic2, am, y);
}
static void hb_up_half(float* f, const float* x, int second, float* y) /* upsampleFirstHalf / SecondHalf, MLDSPFilters.h:1248-1270 */
{
  int i2 = 0;
  for (int i = second ? VEC / 2 : 0; i < (second ? VEC : VEC / 2); ++i)
  {
    y[i2++] = hb_a(f, x[i]);
    y[i2++] = hb_b(f, x[i]);
  }
}
static void hb_down2(float* f, const float* x1, const float* x2, float* y) /* downsample(vx1, vx2), :1272-1294 */
{
  for (int half = 0; half < 2; ++half)
  {
    const float* x = half ? x2 : x1;
    for (int i = 0; i < VEC / 2; ++i)
    {
      const float a0 = hb_a(f, x[2 * i]), b0 = hb_b(f, x[2 * i + 1]);
      y[half * (VEC / 2) + i] = (a0 + f[8]) * 0.5f;
      f[8] = b0;
    }
  }
}
int mlorc_rate_function_run(int up, size_t V, size_t T, const float* freq, const float* lopassCoeffs, const float* x, const float* m, float* out)
{
  for (size_t v = 0; v < V; ++v)
  {
    rate_fn_state st = {0u, 0.f, 0.f, freq[v], lopassCoeffs};
    float fin[2][9] = {{0}}, fout[9] = {0};       /* mUppers / mDowners, one HalfBandFilter per row */
    float inBuf[2][VEC] = {{0}}, outBuf[VEC] = {0}; /* Downsample2xFunction::mInputBuffer, mOutputBuffer */
    int phase = 0;                                  /* mPhase{false} */
    for (size_t t = 0; t < T; ++t)
    {
      const float* vx = x + (v * T + t) * VEC;
      const float* vm = m + (v * T + t) * VEC;
      float y[VEC];
      if (up) /* :126-143 */
      {
        float a0[VEC], a1[VEC], b0[VEC], b1[VEC], o1[VEC], o2[VEC];
        hb_up_half(fin[0], vx, 0, a0);
        hb_up_half(fin[0], vx, 1, b0);
        hb_up_half(fin[1], vm, 0, a1);
        hb_up_half(fin[1], vm, 1, b1);
        rate_fn(&st, a0, a1, o1);
        rate_fn(&st, b0, b1, o2);
        hb_down2(fout, o1, o2, y);
      }
      else if (phase) /* :179-198 */
      {
        float d0[VEC], d1[VEC], o[VEC];
        hb_down2(fin[0], inBuf[0], vx, d0);
        hb_down2(fin[1], inBuf[1], vm, d1);
        rate_fn(&st, d0, d1, o);
        hb_up_half(fout, o, 0, y);
        hb_up_half(fout, o, 1, outBuf);
      }
      else /* :200-206 */
      {
        memcpy(inBuf[0], vx, sizeof(float) * VEC);
        memcpy(inBuf[1], vm, sizeof(float) * VEC);
        memcpy(y, outBuf, sizeof(float) * VEC);
      }
      if (!up) phase = !phase;
      for (int n = 0; n < VEC; ++n) out[(v * T + t) * VEC + n] = y[n] * 0.5f;
    }
  }
  return MLGPU_OK;
}

int mlorc_resample(int octaves, int up, size_t V, size_t T_in, float* state, const float* in, float* out)
{
  if (octaves < 0 || octaves > 6) return MLGPU_ERR_INVALID;
  const size_t R = (size_t)1 << octaves, Sin = T_in * VEC, Sout = up ? Sin * R : Sin / R;
  float f[6 * 9];
  for (size_t v = 0; v < V; ++v)
  {
    for (int i = 0; i < octaves * 9; ++i) f[i] = state[(size_t)i * V + v];
    if (up)
      for (size_t s = 0; s < Sin; ++s) up_rec(f, octaves, octaves, in[v * Sin + s], out + v * Sout + s * R);
    else
      for (size_t s = 0; s < Sout; ++s) out[v * Sout + s] = down_rec(f, octaves, in + v * Sin + s * R);
    for (int i = 0; i < octaves * 9; ++i) state[(size_t)i * V + v] = f[i];
  }
  return MLGPU_OK;
}

/* ==== AudioContext::ProcessTime (source/app/MLAudioContext.h:27-57, MLAudioContext.cpp:16-104): the quarter-note phasor behind
 * ctx->getBeatPhase(). A script drives it as a plug-in wrapper does: kind 0 = updateTime(ppq, bpm, playing, sr) -> setTimeAndRate,
 * kind 1 = `vectors` x processVector, kind 2 = clear(). out: the concatenated phasor; since[i]: samplesSinceStart after step i. ==== */
typedef struct
{
  int kind, vectors, playing, pad;
  double ppq, bpm, sr;
} mlorc_transport_step;

int mlorc_transport_run(const mlorc_transport_step* steps, int n_steps, float* out, uint64_t* since)
{
  /* ProcessTime's members and their initial values, MLAudioContext.h:44-56 */
  float omega = 0.f;
  int playing1 = 0, active1 = 0;
  double dpdt = 0., ppqPos1 = -1., ppqPhase1 = 0., bpm = 0., sampleRate = 0.;
  uint64_t samplesSincePreviousTime = 0, samplesSinceStart = 0;
  size_t pos = 0;
  (void)active1;
  (void)bpm;
  for (int i = 0; i < n_steps; ++i)
  {
    const mlorc_transport_step* st = &steps[i];
    if (st->kind == 0) /* setTimeAndRate, :16-80 */
    {
      const double ppqPos = st->ppq, bpmIn = st->bpm;
      const int isPlaying = st->playing != 0;
      if (!(isnan(ppqPos) || isinf(ppqPos) || isnan(bpmIn) || isinf(bpmIn))) /* :20-26 */
      {
        sampleRate = st->sr;
        bpm = bpmIn;
        const int active = (ppqPos1 != ppqPos) && isPlaying;
        const int justStarted = isPlaying && !playing1;
        double ppqPhase = 0.;
        if (active)
        {
          ppqPhase = (ppqPos > 0.f) ? ppqPos - floor(ppqPos) : ppqPos;
          omega = (float)ppqPhase;
          if (justStarted)
          {
            samplesSinceStart = 0;
            omega = 0.f;
            const double dsdt = 1. / sampleRate;
            const double minutesPerSample = dsdt / 60.;
            dpdt = bpm * minutesPerSample;
          }
          else
          {
            double dPhase = ppqPhase - ppqPhase1;
            if (dPhase < 0.) dPhase += 1.;
            const double x = dPhase / (double)samplesSincePreviousTime;
            dpdt = (x < 0.) ? 0. : (x > 1. ? 1. : x); /* ml::clamp, MLDSPScalarMath.h:69-72 */
          }
        }
        else
        {
          omega = -1.f;
          dpdt = 0.;
        }
        ppqPos1 = ppqPos;
        ppqPhase1 = ppqPhase;
        active1 = active;
        playing1 = isPlaying;
        samplesSincePreviousTime = 0;
      }
    }
    else if (st->kind == 2) /* clear, :82-87 */
    {
      dpdt = 0.;
      active1 = 0;
      playing1 = 0;
    }
    else
      for (int v = 0; v < st->vectors; ++v) /* processVector, :91-104 */
      {
        for (int n = 0; n < VEC; ++n)
        {
          out[pos++] = omega;
          omega = (float)((double)omega + dpdt); /* float += double */
          if (omega > 1.f) omega -= 1.f;
        }
        samplesSincePreviousTime += VEC;
        samplesSinceStart += VEC;
      }
    since[i] = samplesSinceStart;
  }
  return 0;
}

/* ==== EventsToSignals::SmoothedController (MLEventsToSignals.h:170-180, .cpp:264-281), what AudioContext::getInputController(n)
 * returns: per DSPVector of an awake instrument output = glide(inputValue), a LinearGlide of int(sr * 0.02) samples; zeros while
 * the instrument has never seen an event (:386). values[t] = the controller's inputValue during vector t (the last controller event
 * so far), awake_from = the first vector processed after the instrument's first event. out: n_vectors x 64. ==== */
int mlorc_smoothed_controller_run(double sr, const float* values, size_t n_vectors, size_t awake_from, float* out)
{
  float C[2];
  uint32_t S[3 + VEC];
  const int procs[1] = {MLGPU_PROC_LINEAR_GLIDE};
  mlorc_linear_glide_make_coeffs((float)(int)(sr * 0.02f), C); /* int glideTimeInSamples = sr * kControllerGlideTimeSeconds, :275 */
  mlorc_chain_default_state(procs, 1, 1, S);
  for (size_t t = 0; t < n_vectors; ++t)
  {
    if (t < awake_from)
    {
      for (int n = 0; n < VEC; ++n) out[t * VEC + n] = 0.f;
      continue;
    }
    float in[VEC];
    for (int n = 0; n < VEC; ++n) in[n] = values[t];
    const float* ins[1] = {in};
    const int st = mlorc_proc_process_multi(MLGPU_PROC_LINEAR_GLIDE, 1, 1, C, S, ins, 1, out + t * VEC);
    if (st) return st;
  }
  return 0;
}
