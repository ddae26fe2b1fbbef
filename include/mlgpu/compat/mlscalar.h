// mlscalar.h — the HOST-SIDE scalar helpers of madronalib's DSP layer, for programs that build against the MI355X shim without a
// madronalib checkout: the constants and scalar templates of source/DSP/MLDSPScalarMath.h (:21-211), its compile-time math
// (const_math, :215-371) and the interval / projection helpers of source/DSP/MLDSPProjections.h (:13-300). None of this runs on
// the device - it is what a process function computes with plain floats before it hands them to DSPVector code (the decay knob
// of the reference's reverb.cpp, the controller-to-frequency map of controllers-to-audio.cpp) - but those floats become constants
// of the captured kernel, so the values matter: every function here performs the reference's arithmetic in the reference's order
// (tests/test_host_cpp.py::test_scalar_helpers_match_the_reference compares a sweep of each against the reference's own header,
// bit for bit). mldsp.h includes it when madronalib's own headers are not on the include path. The scalar templates, the random
// source and the projections are written for this repository against that contract; the compile-time math block (const_math) follows
// a third party's algorithms step for step and carries that author's notice - see there.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <initializer_list>
#include <limits>
#include <vector>

namespace ml
{
constexpr float kTwoPi = 6.2831853071795864769252867f;
constexpr float kPi = 3.1415926535897932384626433f;
constexpr float kOneOverTwoPi = 1.0f / kTwoPi;
constexpr float kE = 2.718281828459045f;
constexpr float kTwelfthRootOfTwo = 1.05946309436f;
constexpr float kMinGain = 0.00001f;

// ---- integers -------------------------------------------------------------------------------------------------------------
// the exponent of the smallest power of two that is >= x (0 for x <= 1)
inline size_t bitsToContain(int x)
{
  size_t e = 0;
  while ((1 << e) < x) ++e;
  return e;
}
// x rounded up to a multiple of 2^chunkSizeExponent
inline int chunkSizeToContain(int chunkSizeExponent, int x)
{
  const int chunk = 1 << chunkSizeExponent;
  return (x + chunk - 1) & ~(chunk - 1);
}
inline int modulo(int a, int b) { return a >= 0 ? a % b : (b - std::abs(a % b)) % b; }
// (the quotient is formed in float; the rest of the expression runs in double - the reference calls the C library's floor(double))
inline float modulo(float a, float b) { return (float)((double)a - (double)b * ::floor((double)(a / b))); }
// floor(log2(x)) for x >= 1, 0 below
inline int ilog2(int x)
{
  int b = 0;
  for (int s = 16; s >= 1; s >>= 1)
    if (x >= (1 << s))
    {
      x >>= s;
      b |= s;
    }
  return b;
}

// ---- scalar templates (the DSPVector forms of the same names are the shim's) ---------------------------------------------------
template <class c>
constexpr c(min)(const c& a, const c& b)
{
  return (a < b) ? a : b;
}
template <class c>
constexpr c(max)(const c& a, const c& b)
{
  return (a > b) ? a : b;  // (the second operand when either is NaN, like min)
}
template <class c>
constexpr c(clamp)(const c& x, const c& lo, const c& hi)
{
  return (x < lo) ? lo : ((x > hi) ? hi : x);
}
template <class c>
constexpr c lerp(const c& a, const c& b, const c& m)
{
  return a + m * (b - a);
}
template <class c>
constexpr bool within(const c& x, const c& lo, const c& hi)
{
  return (x >= lo) && (x < hi);
}
template <class c>
constexpr bool withinClosedInterval(const c& x, const c& lo, const c& hi)
{
  return (x >= lo) && (x <= hi);
}
template <class c>
constexpr int(sign)(const c& x)
{
  return (x == 0) ? 0 : ((x > 0) ? 1 : -1);
}

inline int isNaN(float x) { return std::isnan(x); }
inline int isNaN(double x) { return std::isnan(x); }
inline int isInfinite(float x) { return std::isinf(x); }
inline int isInfinite(double x) { return std::isinf(x); }

inline float smoothstep(float a, float b, float x)
{
  x = clamp((x - a) / (b - a), 0.f, 1.f);
  return x * x * (3 - 2 * x);
}
inline float boolToFloat(uint32_t b) { return b ? 1.0f : 0.0f; }
// 1 for a clear sign bit, 0 for a set one
inline float fSignBit(float f)
{
  uint32_t u;
  std::memcpy(&u, &f, sizeof u);
  return (u >> 31) ? 0.0f : 1.0f;
}
inline float lerpBipolar(const float a, const float b, const float c, const float m)
{
  const float absm = std::fabs(m);
  const float pos = m > 0., neg = m < 0.;
  const float q = pos * c + neg * a;
  return b + (q - b) * absm;
}
// 4-point, 3rd-order Hermite interpolation through t[1] .. t[2]
inline float herp(const float* t, float phase)
{
  const float c = (t[2] - t[0]) * 0.5f;
  const float v = t[1] - t[2];
  const float w = c + v;
  const float a = w + v + (t[3] - t[1]) * 0.5f;
  const float b = w + a;
  return (((a * phase) - b) * phase + c) * phase + t[1];
}
inline float ampTodB(float a) { return 20.f * log10f(a); }
inline float dBToAmp(float dB) { return powf(10.f, dB / 20.f); }

// the 32-bit linear congruential generator NoiseGen also uses (MLDSPGens.h:115): floats on [-1, 1) from bits 9 .. 31
class RandomScalarSource
{
 public:
  RandomScalarSource() : seed_(0) {}
  inline void step() { seed_ = seed_ * 0x0019660Du + 0x3C6EF35Fu; }
  float getFloat()
  {
    step();
    const uint32_t bits = ((seed_ >> 9) & 0x007FFFFFu) | 0x3F800000u;  // [1, 2)
    float f;
    std::memcpy(&f, &bits, sizeof f);
    f *= 2.f;
    f -= 3.f;
    return f;
  }
  uint32_t getUInt32()
  {
    step();
    return seed_;
  }
  uint32_t seed_;
};

// ---- compile-time math: crude but constexpr, and what the reference's own constants are made with (SineGen's sqrt(2) is
// const_math::sqrt's 1.41421568, not 1.41421356). Same recurrences, same stopping rule (an absolute tolerance of 0.001). ---------
//
// ATTRIBUTION. The functions of this namespace have to return the reference's values bit for bit (they are compile-time constants of
// DSP code), which pins the algorithm of each - series, recurrence, stopping rule, operation order - to that of the reference's
// block (source/DSP/MLDSPScalarMath.h:215-371), which is itself third-party code carrying this notice:
//     C++11 constexpr versions of cmath functions needed for the FFT.
//     Copyright Paul Keir 2012-2016
//     Distributed under the Boost Software License, Version 1.0.
//     (See accompanying file license.txt or copy at http://boost.org/LICENSE_1_0.txt)
// Several functions are restated iteratively here (sqrt, pow, mantissa, exponent) or under other helper names, but this block is a
// derivative of that work as far as its algorithms go and is distributed under the same Boost Software License 1.0.
namespace const_math
{
constexpr double tol = 0.001;
constexpr double abs(const double x) { return x < 0.0 ? -x : x; }
constexpr double square(const double x) { return x * x; }
constexpr double cube(const double x) { return x * x * x; }
// Newton's iteration from 1, until a step is shorter than tol
constexpr double sqrt(const double x)
{
  double g = 1.0;
  while (!(abs(g - x / g) < tol)) g = (g + x / g) / 2.0;
  return g;
}
constexpr double pow(double base, int exponent)
{
  if (exponent < 0) return 1.0 / pow(base, -exponent);
  if (exponent == 0) return 1.;
  double r = base;  // base * (base * (... * base)): the product is built from the innermost factor outwards
  for (int i = 1; i < exponent; ++i) r = base * r;
  return r;
}
// sin 3x = 3 sin x - 4 sin^3 x, from the argument divided by three until it is below tol
constexpr double sin_tripled(const double x)
{
  if (x < tol) return x;
  const double s = sin_tripled(x / 3.0);
  return 3 * s - 4 * cube(s);
}
constexpr double sin(const double x) { return sin_tripled(x < 0 ? -x + kPi : x); }
constexpr double sinh_tripled(const double x)
{
  if (x < tol) return x;
  const double s = sinh_tripled(x / 3.0);
  return 3 * s + 4 * cube(s);
}
constexpr double sinh(const double x) { return x < 0 ? -sinh_tripled(-x) : sinh_tripled(x); }
constexpr double cos(const double x) { return sin(kPi * 0.5 - x); }
constexpr double cosh(const double x) { return sqrt(1.0 + square(sinh(x))); }
// the arctangent series x - x^3/3 + x^5/5 - ..., two terms at a time, until a pair is below tol
constexpr double atan_pairs(const double res, const double num1, const double den1, const double delta)
{
  return res < tol ? res : res + atan_pairs((num1 * delta) / (den1 + 2.) - num1 / den1, num1 * delta * delta, den1 + 4., delta);
}
constexpr double atan_poly(const double x) { return x + atan_pairs(pow(x, 5) / 5. - pow(x, 3) / 3., pow(x, 7), 7., x * x); }
constexpr double atan_reduced(const double x)
{
  return x <= (2. - sqrt(3.)) ? atan_poly(x) : (kTwoPi / 3.) + atan_poly((sqrt(3.) * x - 1) / (sqrt(3.) + x));
}
constexpr double atan_folded(const double x) { return (x < 1) ? atan_reduced(x) : kTwoPi - atan_reduced(1 / x); }
constexpr double atan(const double x) { return (x >= 0) ? atan_folded(x) : -atan_folded(-x); }
constexpr double atan2(const double y, const double x)
{
  if (x > 0) return atan(y / x);
  if (y >= 0 && x < 0) return atan(y / x) + kPi;
  if (y < 0 && x < 0) return atan(y / x) - kPi;
  if (y > 0 && x == 0) return kTwoPi;
  if (y < 0 && x == 0) return -kTwoPi;
  return 0;
}
constexpr double nearest(double x) { return (x - 0.5) > (int)x ? (int)(x + 0.5) : (int)x; }
constexpr double fraction(double x) { return (x - 0.5) > (int)x ? -(((double)(int)(x + 0.5)) - x) : x - ((double)(int)(x)); }
constexpr double exp_series(const double r)
{
  return 1.0 + r + pow(r, 2) / 2.0 + pow(r, 3) / 6.0 + pow(r, 4) / 24.0 + pow(r, 5) / 120.0 + pow(r, 6) / 720.0 + pow(r, 7) / 5040.0;
}
constexpr double exp(const double x) { return pow(kE, (int)nearest(x)) * exp_series(fraction(x)); }
// decimal mantissa in [1, 10) and its exponent
constexpr double mantissa(double x)
{
  while (x >= 10.0 || x < 1.0) x = (x >= 10.0) ? x * 0.1 : x * 10.0;
  return x;
}
constexpr int exponent(double x)
{
  int e = 0;
  while (x >= 10.0 || x < 1.0)
  {
    if (x >= 10.0)
    {
      x = x * 0.1;
      ++e;
    }
    else
    {
      x = x * 10.0;
      --e;
    }
  }
  return e;
}
constexpr double log_series(const double y)
{
  return 2.0 * (y + pow(y, 3) / 3.0 + pow(y, 5) / 5.0 + pow(y, 7) / 7.0 + pow(y, 9) / 9.0 + pow(y, 11) / 11.0);
}
constexpr double log(const double x)
{
  if (x == 0) return -std::numeric_limits<double>::infinity();
  if (x < 0) return std::numeric_limits<double>::quiet_NaN();
  const double root = sqrt(mantissa(x));
  return 2.0 * log_series((root - 1.0) / (root + 1.0)) + 2.3025851 * exponent(x);
}
}  // namespace const_math

// ---- intervals and projections (float -> float maps a host builds its parameter curves from) -------------------------------------
struct Interval
{
  float x1;
  float x2;
  bool operator==(const Interval& b) const { return (x1 == b.x1) && (x2 == b.x2); }
  bool operator!=(const Interval& b) const { return !(*this == b); }
  const Interval operator*(const float b) const { return Interval{x1 * b, x2 * b}; }
  const Interval operator*=(const float b)
  {
    x1 *= b;
    x2 *= b;
    return *this;
  }
};
inline float midpoint(Interval m) { return (m.x1 + m.x2) * 0.5f; }
inline bool within(float f, const Interval m) { return (f >= m.x1) && (f < m.x2); }

using Projection = std::function<float(float)>;
inline Projection compose(Projection a, Projection b)
{
  return [=](float x) { return a(b(x)); };
}

namespace projections
{
// shapes on [0, 1]
static const Projection zero{[](float) { return 0.f; }};
static const Projection unity{[](float x) { return x; }};
static const Projection squared{[](float x) { return x * x; }};
static const Projection flip{[](float x) { return 1 - x; }};
static const Projection clip{[](float x) { return ml::clamp(x, 0.f, 1.f); }};
static const Projection smoothstep{[](float x) { return 3 * x * x - 2 * x * x * x; }};
static const Projection flatcenter{[](float x) {
  const float c = (x - 0.5f);
  return 4 * c * c * c + 0.5f;
}};
static const Projection bell{[](float x) {
  const float px = x * 2 - 1;
  return powf(2.f, -(10.f * px * px));
}};
static const Projection easeOut{[](float x) {
  const float m = x - 1;
  return 1 - m * m;
}};
static const Projection easeIn{[](float x) { return x * x; }};
static const Projection easeInOut{[](float x) { return (x < 0.5f) ? easeIn(x * 2.f) * 0.5f : easeOut(x * 2.f - 1.f) * 0.5f + 0.5f; }};
static const Projection easeOutCubic{[](float x) {
  const float n = 1 - x;
  return 1 - n * n * n;
}};
static const Projection easeInCubic{[](float x) { return x * x * x; }};
static const Projection easeInOutCubic{[](float x) { return (x < 0.5f) ? easeInCubic(x * 2.f) * 0.5f : easeOutCubic(x * 2.f - 1.f) * 0.5f + 0.5f; }};
static const Projection easeOutQuartic{[](float x) {
  const float m = x - 1;
  return 1 - m * m * m * m;
}};
static const Projection easeInQuartic{[](float x) { return x * x * x * x; }};
static const Projection easeInOutQuartic{[](float x) { return (x < 0.5f) ? easeInQuartic(x * 2.f) * 0.5f : easeOutQuartic(x * 2.f - 1.f) * 0.5f + 0.5f; }};
static const Projection overshoot{[](float x) { return 3 * x - 2 * x * x; }};
static const Projection bisquared{[](float x) { return std::fabs(x) * x; }};
static const Projection invBisquared{[](float x) { return sqrtf(std::fabs(x)) * ml::sign(x); }};

inline Projection constant(const float k)
{
  return [=](float) { return k; };
}
// [0, 1] -> a logarithmic curve over [a, b], scaled back to [0, 1]; positive a < b
inline Projection log(Interval m)
{
  const float a = m.x1, b = m.x2;
  if (b - a == 0.f) return [=](float) { return a; };
  if (a == 0.f) return [=](float) { return 0.f; };
  return [=](float x) { return a * (powf((b / a), x) - 1) / (b - a); };
}
// its inverse
inline Projection exp(Interval m)
{
  const float a = m.x1, b = m.x2;
  if (b - a == 0.f) return [=](float) { return a; };
  if (a == 0.f) return [=](float) { return 0.f; };
  return [=](float x) { return logf((x * (b - a) + a) / a) / logf(b / a); };
}
inline Projection linear(const Interval a, const Interval b)
{
  const float a1 = a.x1, a2 = a.x2, b1 = b.x1, b2 = b.x2;
  if (a1 - a2 == 0.f) return [=](float) { return b1; };
  return [=](float x) {
    const float m = (b2 - b1) / (a2 - a1);
    return m * (x - a1) + b1;
  };
}
inline Projection add(float f)
{
  return [=](float x) { return x + f; };
}
// interval a -> [0, 1] -> the shape c -> interval b
inline Projection intervalMap(const Interval a, const Interval b, Projection c)
{
  return [=](float x) {
    const float scaleA = 1 / (a.x2 - a.x1);
    const float offsetA = (-a.x1) / (a.x2 - a.x1);
    const float scaleB = (b.x2 - b.x1);
    const float offsetB = b.x1;
    return c(x * scaleA + offsetA) * scaleB + offsetB;
  };
}
inline Projection unityToLogParam(Interval paramInterval) { return intervalMap({0, 1}, paramInterval, projections::log(paramInterval)); }
inline Projection logParamToUnity(Interval paramInterval) { return intervalMap(paramInterval, {0, 1}, projections::exp(paramInterval)); }

// n values spread evenly over [0, 1], joined by straight lines - or, with one shape per segment, by that shape
inline Projection piecewise(std::initializer_list<float> valueList, std::initializer_list<Projection> shapeList)
{
  const std::vector<float> table(valueList);
  const std::vector<Projection> shapes(shapeList);
  if (table.empty()) return [](float) { return 0.f; };
  if (table.size() == 1) return [=](float) { return table[0]; };
  return [=](float x) {
    const int last = (int)table.size() - 1;
    if (!(x < 1.0f)) return table[(size_t)last];
    const float xf = static_cast<float>(last) * clamp(x, 0.f, 1.f);
    const int xi = static_cast<int>(xf);
    const float xr = xf - xi;
    return lerp(table[(size_t)xi], table[(size_t)xi + 1], shapes.empty() ? xr : shapes[(size_t)xi](xr));
  };
}
inline Projection piecewiseLinear(std::initializer_list<float> values) { return piecewise(values, {}); }
}  // namespace projections
}  // namespace ml
