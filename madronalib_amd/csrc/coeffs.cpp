// coeffs.cpp — host-side coefficient makers of the reference filters (glibc libm).
//
// The reference computes filter coefficients with scalar libm calls on the host
// (sinf/tanf/expf/cosf/sqrtf/powf: source/DSP/MLDSPFilters.h:30,85-95,270-281,350-362,415-425,
// 458-462,498,679-686) and so does this engine: coefficients are made here, uploaded once,
// and device code never has to reproduce libm. Formulas and
// operation order follow the cited lines; compiled with -ffp-contract=off like the rest.
#include <math.h>
#include <string.h>

#include "mlgpu_internal.hpp"

namespace
{
constexpr float kTwoPi = 6.2831853071795864769252867f;  // MLDSPScalarMath.h:23
constexpr float kPi = 3.1415926535897932384626433f;     // MLDSPScalarMath.h:24

void svfCoeffs(float omega, float k, float* o)  // MLDSPFilters.h:85-95 (== :168-177, :212-221)
{
  const float piOmega = kPi * omega;
  const float s1 = sinf(piOmega);
  const float s2 = sinf(2.0f * piOmega);
  const float nrm = 1.0f / (2.f + k * s2);
  o[0] = s2 * nrm;
  o[1] = (-2.f * s1 * s1 - k * s2) * nrm;
  o[2] = (2.0f * s1 * s1) * nrm;
}
}  // namespace

extern "C"
{
  void mlgpu_lopass_make_coeffs(float omega, float k, float out3[3]) { svfCoeffs(omega, k, out3); }
  void mlgpu_bandpass_make_coeffs(float omega, float k, float out3[3]) { svfCoeffs(omega, k, out3); }
  void mlgpu_hipass_make_coeffs(float omega, float k, float out4[4])
  {
    svfCoeffs(omega, k, out4);
    out4[3] = k;
  }
  void mlgpu_loshelf_make_coeffs(float omega, float k, float A, float r[5])  // :270-281
  {
    const float piOmega = kPi * omega;
    const float g = tanf(piOmega) / sqrtf(A);
    r[0] = 1.f / (1.f + g * (g + k));
    r[1] = g * r[0];
    r[2] = g * r[1];
    r[3] = k * (A - 1.f);
    r[4] = (A * A - 1.f);
  }
  void mlgpu_hishelf_make_coeffs(float omega, float k, float A, float r[6])  // :350-362
  {
    const float piOmega = kPi * omega;
    const float g = tanf(piOmega) * sqrtf(A);
    r[0] = 1.f / (1.f + g * (g + k));
    r[1] = g * r[0];
    r[2] = g * r[1];
    r[3] = A * A;
    r[4] = k * (1.f - A) * A;
    r[5] = (1.f - A * A);
  }
  void mlgpu_bell_make_coeffs(float omega, float k, float A, float r[4])  // :415-425
  {
    const float kc = k / A;
    const float piOmega = kPi * omega;
    const float g = tanf(piOmega);
    const float a1 = 1.f / (1.f + g * (g + kc));
    const float a2 = g * a1;
    const float a3 = g * a2;
    const float m1 = kc * (A * A - 1.f);
    r[0] = a1;
    r[1] = a2;
    r[2] = a3;
    r[3] = m1;
  }
  void mlgpu_onepole_make_coeffs(float omega, float out2[2])  // :458-462
  {
    const float x = expf(-omega * kTwoPi);
    out2[0] = 1.f - x;
    out2[1] = x;
  }
  float mlgpu_dcblocker_make_coeffs(float omega) { return cosf(omega); }  // :498
  void mlgpu_adsr_calc_coeffs(float a, float d, float s, float r, float sr, float o[4])  // :679-686
  {
    const float minSegmentTime = 0.0002f;
    const float invSr = 1.0f / sr;
    o[0] = kTwoPi * invSr / ((a > minSegmentTime) ? a : minSegmentTime);
    o[1] = kTwoPi * invSr / ((d > minSegmentTime) ? d : minSegmentTime);
    o[2] = s;
    o[3] = kTwoPi * invSr / ((r > minSegmentTime) ? r : minSegmentTime);
  }
  float mlgpu_db_to_gain(float dB) { return powf(10.f, dB / 40.f); }  // :30

  float mlgpu_allpass1_make_coeffs(float d)  // :938-943
  {
    const float xm1 = (d - 1.f);
    return -0.53f * xm1 + 0.24f * xm1 * xm1;
  }
  void mlgpu_fractional_delay_make_state(float d, float* o)  // :991-1007
  {
    const float fDelayInt = floorf(d);
    int32_t delayInt = static_cast<int32_t>(fDelayInt);
    float delayFrac = d - fDelayInt;
    if ((delayFrac < 0.618f) && (delayInt > 0))
    {
      delayFrac += 1.f;
      delayInt -= 1;
    }
    memcpy(&o[0], &delayInt, 4);
    o[1] = mlgpu_allpass1_make_coeffs(delayFrac);
  }

  // LinearGlide::setGlideTimeInSamples, MLDSPGens.h:444-449: glide time quantized to whole DSPVectors
  void mlgpu_linear_glide_make_coeffs(float t, float* o)
  {
    int32_t vectorsPerGlide = static_cast<int32_t>(t / 64);
    if (vectorsPerGlide < 1) vectorsPerGlide = 1;
    memcpy(&o[0], &vectorsPerGlide, 4);
    o[1] = 1.0f / (vectorsPerGlide + 0.f);
  }
  // SampleAccurateLinearGlide::setGlideTimeInSamples, MLDSPGens.h:527-532
  void mlgpu_sample_accurate_linear_glide_make_coeffs(float t, float* o)
  {
    int32_t samplesPerGlide = static_cast<int32_t>(t);
    if (samplesPerGlide < 1) samplesPerGlide = 1;
    memcpy(&o[0], &samplesPerGlide, 4);
    o[1] = 1.0f / samplesPerGlide;
  }
}

// ImpulseGen's 17-tap Blackman-windowed sinc, normalised to unit sum: MLDSPGens.h:65-78 with
// makeWindow (MLDSPUtils.h:22-26), dspwindows::blackman (:34-35), projections::linear
// (MLDSPProjections.h:147-167) and normalize/sum (MLDSPOps.h:995-1005,1041-1050).
void mlgpu_build_impulse_table(float* out17)
{
  constexpr int N = 17;
  float prod[64];
  for (int n = 0; n < 64; ++n)
  {
    float window = 0.f;
    if (n < N)
    {
      const float m = (1.f - 0.f) / ((N - 1.f) - 0.f);
      const float x = m * ((float)n - 0.f) + 0.f;
      window = 0.42f - 0.5f * cosf(kTwoPi * x) + 0.08f * cosf(2.f * kTwoPi * x);
    }
    const int i = n - (N - 1) / 2;
    const float pi_x = kTwoPi * 0.25f * i;
    const float sinc = (i == 0) ? 1.f : sinf(pi_x) / pi_x;
    prod[n] = sinc * window;
  }
  float sum = 0.f;  // association order of sum(): (x0+x2)+(x1+x3) per group of 4, then in sequence
  for (int g = 0; g < 16; ++g)
  {
    const float* q = prod + 4 * g;
    const float t0 = q[0] + q[2], t1 = q[1] + q[3];
    sum += (t0 + t1);
  }
  for (int n = 0; n < N; ++n) out17[n] = prod[n] / sum;
}

extern "C"
{
  // makeWindow(pDest, size, dspwindows::shape), source/DSP/MLDSPUtils.h:22-47: element i = shape(m * (i - 0) + 0) with
  // m = (1 - 0) / ((size - 1) - 0) - projections::linear({0, size - 1}, {0, 1}), MLDSPProjections.h:147-167, whose degenerate
  // interval (size 1) maps everything to 0. Host libm (cosf), like the coefficient makers.
  int mlgpu_make_window(float* dest, size_t size, int shape)
  {
    if (!dest || shape < 0 || shape > MLGPU_WINDOW_FLAT_TOP) return MLGPU_ERR_INVALID;
    const float a2 = size - 1.f;
    const bool degenerate = (0.f - a2 == 0.f);
    const float m = degenerate ? 0.f : (1.f - 0.f) / (a2 - 0.f);
    for (size_t i = 0; i < size; ++i)
    {
      const float x = degenerate ? 0.f : m * ((float)(int)i - 0.f) + 0.f;
      float w = 0.f;
      switch (shape)
      {
        case MLGPU_WINDOW_RECTANGLE: w = (x > 0.75f) ? 0.f : ((x < 0.25f) ? 0.f : 1.f); break;
        case MLGPU_WINDOW_TRIANGLE: w = (x > 0.5f) ? (2.f - 2.f * x) : (2.f * x); break;
        case MLGPU_WINDOW_RAISED_COSINE: w = 0.5f - 0.5f * cosf(kTwoPi * x); break;
        case MLGPU_WINDOW_HAMMING: w = 0.54f - 0.46f * cosf(kTwoPi * x); break;
        case MLGPU_WINDOW_BLACKMAN: w = 0.42f - 0.5f * cosf(kTwoPi * x) + 0.08f * cosf(2.f * kTwoPi * x); break;
        default:
        {
          const float a0 = 0.21557895f, a1 = 0.41663158f, a2c = 0.277263158f, a3 = 0.083578947f, a4 = 0.006947368f;
          w = a0 - a1 * cosf(kTwoPi * x) + a2c * cosf(2.f * kTwoPi * x) - a3 * cosf(3.f * kTwoPi * x) + a4 * cosf(4.f * kTwoPi * x);
        }
      }
      dest[i] = w;
    }
    return MLGPU_OK;
  }
}
