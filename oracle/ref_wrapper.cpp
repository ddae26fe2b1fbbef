// oracle/ref_wrapper.cpp — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// A thin extern "C" harness around the UNMODIFIED reference headers, compiled from the
// sources where they lie under /root/reference (see oracle/Makefile; flags pinned to
// -std=c++17 -O2, no -march/-mfma). The resulting oracle/_ref/libmlref.so
// is the ground truth that (1) pins the plain-C restatement in oracle/ml_oracle.c,
// (2) generates tests/golden/*, and (3) is the "reference" CPU baseline timed by bench.py.
// Nothing in the product (madronalib_amd/, include/) links or loads this file.
//
// No reference source is copied here: this file only #includes the reference's public
// headers and calls their objects. State of the reference objects is private in places,
// so the harness is compiled with g++'s -fno-access-control (a test-only switch).

// standard headers first, so the access hack below never touches them
#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <functional>
#include <iostream>
#include <iterator>
#include <memory>
#include <thread>
#include <type_traits>
#include <vector>
#include <emmintrin.h>
#include <float.h>

#include "mldsp.h"

#include "../include/mlgpu.h"

using namespace ml;

namespace
{
inline uint32_t f2u(float f)
{
  uint32_t u;
  std::memcpy(&u, &f, 4);
  return u;
}
inline float u2f(uint32_t u)
{
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}

// ---------------------------------------------------------------------------
// one reference processor of a chain, behind a tiny virtual interface

struct RefProc
{
  virtual ~RefProc() {}
  virtual int nc() const = 0;
  virtual int ns() const = 0;
  virtual void setCoeffs(const float* c) = 0;     // c[nc]
  virtual void setState(const uint32_t* s) = 0;   // s[ns]
  virtual void getState(uint32_t* s) const = 0;
  virtual void clear() = 0;                        // T::clear() semantics
  virtual DSPVector process(const DSPVector& in) = 0;
};

struct RPhasor : RefProc
{
  PhasorGen g;
  int nc() const override { return 0; }
  int ns() const override { return 1; }
  void setCoeffs(const float*) override {}
  void setState(const uint32_t* s) override { g.mOmega32 = s[0]; }
  void getState(uint32_t* s) const override { s[0] = g.mOmega32; }
  void clear() override { g.clear(); }
  DSPVector process(const DSPVector& in) override { return g(in); }
};
struct RSine : RefProc
{
  SineGen g;
  int nc() const override { return 0; }
  int ns() const override { return 1; }
  void setCoeffs(const float*) override {}
  void setState(const uint32_t* s) override { g._phasor.mOmega32 = s[0]; }
  void getState(uint32_t* s) const override { s[0] = g._phasor.mOmega32; }
  void clear() override { g.clear(); }
  DSPVector process(const DSPVector& in) override { return g(in); }
};
struct RSaw : RefProc
{
  SawGen g;
  int nc() const override { return 0; }
  int ns() const override { return 1; }
  void setCoeffs(const float*) override {}
  void setState(const uint32_t* s) override { g._phasor.mOmega32 = s[0]; }
  void getState(uint32_t* s) const override { s[0] = g._phasor.mOmega32; }
  void clear() override { g.clear(); }
  DSPVector process(const DSPVector& in) override { return g(in); }
};
struct RPulse : RefProc
{
  PulseGen g;
  float width{0.5f};
  int nc() const override { return 1; }
  int ns() const override { return 1; }
  void setCoeffs(const float* c) override { width = c[0]; }
  void setState(const uint32_t* s) override { g._phasor.mOmega32 = s[0]; }
  void getState(uint32_t* s) const override { s[0] = g._phasor.mOmega32; }
  void clear() override { g.clear(); }
  DSPVector process(const DSPVector& in) override { return g(in, DSPVector(width)); }
};
struct RNoise : RefProc
{
  NoiseGen g;
  int nc() const override { return 0; }
  int ns() const override { return 1; }
  void setCoeffs(const float*) override {}
  void setState(const uint32_t* s) override { g.mSeed = s[0]; }
  void getState(uint32_t* s) const override { s[0] = g.mSeed; }
  void clear() override { g.reset(); }
  DSPVector process(const DSPVector&) override { return g(); }
};
struct RTick : RefProc
{
  TickGen g;
  int nc() const override { return 0; }
  int ns() const override { return 1; }
  void setCoeffs(const float*) override {}
  void setState(const uint32_t* s) override { g.mOmega = u2f(s[0]); }
  void getState(uint32_t* s) const override { s[0] = f2u(g.mOmega); }
  void clear() override { g.mOmega = 0; }
  DSPVector process(const DSPVector& in) override { return g(in); }
};
struct RImpulse : RefProc
{
  ImpulseGen g;
  int nc() const override { return 0; }
  int ns() const override { return 2; }
  void setCoeffs(const float*) override {}
  void setState(const uint32_t* s) override
  {
    g._omega = u2f(s[0]);
    g._outputCounter = (int)s[1];
  }
  void getState(uint32_t* s) const override
  {
    s[0] = f2u(g._omega);
    s[1] = (uint32_t)g._outputCounter;
  }
  void clear() override
  {
    g._omega = 0;
    g._outputCounter = 0;
  }
  DSPVector process(const DSPVector& in) override { return g(in); }
};
struct RTestSine : RefProc
{
  TestSineGen g;
  int nc() const override { return 0; }
  int ns() const override { return 1; }
  void setCoeffs(const float*) override {}
  void setState(const uint32_t* s) override { memcpy(&g.mOmega, &s[0], 4); }
  void getState(uint32_t* s) const override { memcpy(&s[0], &g.mOmega, 4); }
  void clear() override { g.clear(); }
  DSPVector process(const DSPVector& in) override { return g(in); }
};
struct ROneShot : RefProc
{
  OneShotGen g;
  int nc() const override { return 0; }
  int ns() const override { return 3; }
  void setCoeffs(const float*) override {}
  void setState(const uint32_t* s) override
  {
    g.mOmega32 = s[0];
    g.mGate = s[1];
    g.mOmegaPrev = s[2];
  }
  void getState(uint32_t* s) const override
  {
    s[0] = g.mOmega32;
    s[1] = g.mGate;
    s[2] = g.mOmegaPrev;
  }
  void clear() override
  {
    g.mOmega32 = 0;
    g.mGate = 0;
    g.mOmegaPrev = 0;
  }
  DSPVector process(const DSPVector& in) override { return g(in); }
};

template <class F, int NC>
struct RSvf : RefProc
{
  F f;
  int nc() const override { return NC; }
  int ns() const override { return 2; }
  void setState(const uint32_t* s) override
  {
    f.ic1eq = u2f(s[0]);
    f.ic2eq = u2f(s[1]);
  }
  void getState(uint32_t* s) const override
  {
    s[0] = f2u(f.ic1eq);
    s[1] = f2u(f.ic2eq);
  }
  void clear() override
  {
    f.ic1eq = 0;
    f.ic2eq = 0;
  }
  DSPVector process(const DSPVector& in) override { return f(in); }
};
struct RLopass : RSvf<Lopass, 3>
{
  void setCoeffs(const float* c) override { f.coeffs = {c[0], c[1], c[2]}; }
};
struct RHipass : RSvf<Hipass, 4>
{
  void setCoeffs(const float* c) override { f.coeffs = {c[0], c[1], c[2], c[3]}; }
};
struct RBandpass : RSvf<Bandpass, 3>
{
  void setCoeffs(const float* c) override { f.coeffs = {c[0], c[1], c[2]}; }
};
struct RLoShelf : RSvf<LoShelf, 5>
{
  void setCoeffs(const float* c) override { f.coeffs = {c[0], c[1], c[2], c[3], c[4]}; }
};
struct RHiShelf : RSvf<HiShelf, 6>
{
  void setCoeffs(const float* c) override { f.coeffs = {c[0], c[1], c[2], c[3], c[4], c[5]}; }
};
struct RBell : RSvf<Bell, 4>
{
  void setCoeffs(const float* c) override { f.coeffs = {c[0], c[1], c[2], c[3]}; }
};

struct ROnePole : RefProc
{
  OnePole f;
  int nc() const override { return 2; }
  int ns() const override { return 1; }
  void setCoeffs(const float* c) override { f.coeffs = {c[0], c[1]}; }
  void setState(const uint32_t* s) override { f.y1 = u2f(s[0]); }
  void getState(uint32_t* s) const override { s[0] = f2u(f.y1); }
  void clear() override { f.clear(); }
  DSPVector process(const DSPVector& in) override { return f(in); }
};
struct RDCBlocker : RefProc
{
  DCBlocker f;
  int nc() const override { return 1; }
  int ns() const override { return 2; }
  void setCoeffs(const float* c) override { f.coeffs = c[0]; }
  void setState(const uint32_t* s) override
  {
    f.x1 = u2f(s[0]);
    f.y1 = u2f(s[1]);
  }
  void getState(uint32_t* s) const override
  {
    s[0] = f2u(f.x1);
    s[1] = f2u(f.y1);
  }
  void clear() override
  {
    f.x1 = 0;
    f.y1 = 0;
  }
  DSPVector process(const DSPVector& in) override { return f(in); }
};
struct RDifferentiator : RefProc
{
  Differentiator f;
  int nc() const override { return 0; }
  int ns() const override { return 1; }
  void setCoeffs(const float*) override {}
  void setState(const uint32_t* s) override { f._x1 = u2f(s[0]); }
  void getState(uint32_t* s) const override { s[0] = f2u(f._x1); }
  void clear() override { f._x1 = 0; }
  DSPVector process(const DSPVector& in) override { return f(in); }
};
struct RIntegrator : RefProc
{
  Integrator f;
  int nc() const override { return 1; }
  int ns() const override { return 1; }
  void setCoeffs(const float* c) override { f.mLeak = c[0]; }
  void setState(const uint32_t* s) override { f.y1 = u2f(s[0]); }
  void getState(uint32_t* s) const override { s[0] = f2u(f.y1); }
  void clear() override { f.y1 = 0; }
  DSPVector process(const DSPVector& in) override { return f(in); }
};
struct RPeak : RefProc
{
  Peak f;
  int nc() const override { return 3; }
  int ns() const override { return 2; }
  void setCoeffs(const float* c) override
  {
    f.coeffs = {c[0], c[1]};
    f.peakHoldSamples = (int)f2u(c[2]);
  }
  void setState(const uint32_t* s) override
  {
    f.y1 = u2f(s[0]);
    f.peakHoldCounter = (int)s[1];
  }
  void getState(uint32_t* s) const override
  {
    s[0] = f2u(f.y1);
    s[1] = (uint32_t)f.peakHoldCounter;
  }
  void clear() override
  {
    f.y1 = 0;
    f.peakHoldCounter = 0;
  }
  DSPVector process(const DSPVector& in) override { return f(in); }
};
struct RRms : RefProc
{
  RMS f;
  int nc() const override { return 2; }
  int ns() const override { return 1; }
  void setCoeffs(const float* c) override { f.coeffs = {c[0], c[1]}; }
  void setState(const uint32_t* s) override { f.y1 = u2f(s[0]); }
  void getState(uint32_t* s) const override { s[0] = f2u(f.y1); }
  void clear() override { f.y1 = 0; }
  DSPVector process(const DSPVector& in) override { return f(in); }
};
struct RAdsr : RefProc
{
  ADSR f;
  int nc() const override { return 4; }
  int ns() const override { return 8; }
  void setCoeffs(const float* c) override { f.coeffs = {c[0], c[1], c[2], c[3]}; }
  void setState(const uint32_t* s) override
  {
    f.y = u2f(s[0]);
    f.y1 = u2f(s[1]);
    f.x1 = u2f(s[2]);
    f.threshold = u2f(s[3]);
    f.target = u2f(s[4]);
    f.k = u2f(s[5]);
    f.amp = u2f(s[6]);
    f.segment = (int)s[7];
  }
  void getState(uint32_t* s) const override
  {
    s[0] = f2u(f.y);
    s[1] = f2u(f.y1);
    s[2] = f2u(f.x1);
    s[3] = f2u(f.threshold);
    s[4] = f2u(f.target);
    s[5] = f2u(f.k);
    s[6] = f2u(f.amp);
    s[7] = (uint32_t)f.segment;
  }
  void clear() override { f.clear(); }
  DSPVector process(const DSPVector& in) override { return f(in); }
};
struct RGain : RefProc
{
  float gain{1.f};
  int nc() const override { return 1; }
  int ns() const override { return 0; }
  void setCoeffs(const float* c) override { gain = c[0]; }
  void setState(const uint32_t*) override {}
  void getState(uint32_t*) const override {}
  void clear() override {}
  DSPVector process(const DSPVector& in) override { return in * gain; }  // x * DSPVector(gain)
};

struct RSampleGlide : RefProc  // SampleAccurateLinearGlide, nextSample() once per sample
{
  SampleAccurateLinearGlide g;
  int nc() const override { return 2; }
  int ns() const override { return 4; }
  void setCoeffs(const float* c) override
  {
    g.mSamplesPerGlide = (int32_t)f2u(c[0]);
    g.mDyPerSample = c[1];
  }
  void setState(const uint32_t* s) override
  {
    g.mCurrValue = u2f(s[0]);
    g.mStepValue = u2f(s[1]);
    g.mTargetValue = u2f(s[2]);
    g.mSamplesRemaining = (int32_t)s[3];
  }
  void getState(uint32_t* s) const override
  {
    s[0] = f2u(g.mCurrValue);
    s[1] = f2u(g.mStepValue);
    s[2] = f2u(g.mTargetValue);
    s[3] = (uint32_t)g.mSamplesRemaining;
  }
  void clear() override { g.clear(); }
  DSPVector process(const DSPVector& in) override
  {
    DSPVector y;
    for (int n = 0; n < kFloatsPerDSPVector; ++n) y[n] = g.nextSample(in[n]);
    return y;
  }
};
// the two vector-rate ramps: process() takes the float argument from sample 0 of the input vector
struct RInterp1 : RefProc
{
  Interpolator1 g;
  int nc() const override { return 0; }
  int ns() const override { return 1; }
  void setCoeffs(const float*) override {}
  void setState(const uint32_t* s) override { g.currentValue = u2f(s[0]); }
  void getState(uint32_t* s) const override { s[0] = f2u(g.currentValue); }
  void clear() override { g.currentValue = 0; }
  DSPVector process(const DSPVector& in) override { return g(in[0]); }
};
struct RLinearGlide : RefProc
{
  LinearGlide g;
  int nc() const override { return 2; }
  int ns() const override { return 3 + kFloatsPerDSPVector; }
  void setCoeffs(const float* c) override
  {
    g.mVectorsPerGlide = (int32_t)f2u(c[0]);
    g.mDyPerVector = c[1];
  }
  void setState(const uint32_t* s) override
  {
    g.mTargetValue = u2f(s[0]);
    g.mStepVec = DSPVector(u2f(s[1]));
    g.mVectorsRemaining = (int32_t)s[2];
    for (int n = 0; n < kFloatsPerDSPVector; ++n) g.mCurrVec[n] = u2f(s[3 + n]);
  }
  void getState(uint32_t* s) const override
  {
    s[0] = f2u(g.mTargetValue);
    s[1] = f2u(g.mStepVec[0]);
    s[2] = (uint32_t)g.mVectorsRemaining;
    for (int n = 0; n < kFloatsPerDSPVector; ++n) s[3 + n] = f2u(g.mCurrVec[n]);
  }
  void clear() override { g.clear(); }
  DSPVector process(const DSPVector& in) override { return g(in[0]); }
};

struct RAllpass1 : RefProc
{
  Allpass1 f{0.f};
  int nc() const override { return 1; }
  int ns() const override { return 2; }
  void setCoeffs(const float* c) override { f.coeffs = c[0]; }
  void setState(const uint32_t* s) override
  {
    f.x1 = u2f(s[0]);
    f.y1 = u2f(s[1]);
  }
  void getState(uint32_t* s) const override
  {
    s[0] = f2u(f.x1);
    s[1] = f2u(f.y1);
  }
  void clear() override { f.clear(); }
  DSPVector process(const DSPVector& in) override { return f(in); }
};

struct RTempoLock : RefProc
{
  TempoLock g;
  int nc() const override { return 0; }
  int ns() const override { return 2; }
  void setCoeffs(const float*) override {}
  void setState(const uint32_t* s) override
  {
    g._omega = u2f(s[0]);
    g._x1v = u2f(s[1]);
  }
  void getState(uint32_t* s) const override
  {
    s[0] = f2u(g._omega);
    s[1] = f2u(g._x1v);
  }
  void clear() override { g.clear(); }
  DSPVector process(const DSPVector& in) override { return in; }
};

RefProc* makeProc(int kind)
{
  switch (kind)
  {
    case MLGPU_PROC_TEMPO_LOCK: return new RTempoLock;
    case MLGPU_PROC_ALLPASS1: return new RAllpass1;
    case MLGPU_PROC_SAMPLE_ACCURATE_LINEAR_GLIDE: return new RSampleGlide;
    case MLGPU_PROC_INTERPOLATOR1: return new RInterp1;
    case MLGPU_PROC_LINEAR_GLIDE: return new RLinearGlide;
    case MLGPU_PROC_PHASOR_GEN: return new RPhasor;
    case MLGPU_PROC_SINE_GEN: return new RSine;
    case MLGPU_PROC_SAW_GEN: return new RSaw;
    case MLGPU_PROC_PULSE_GEN: return new RPulse;
    case MLGPU_PROC_NOISE_GEN: return new RNoise;
    case MLGPU_PROC_TICK_GEN: return new RTick;
    case MLGPU_PROC_IMPULSE_GEN: return new RImpulse;
    case MLGPU_PROC_ONE_SHOT_GEN: return new ROneShot;
    case MLGPU_PROC_TEST_SINE_GEN: return new RTestSine;
    case MLGPU_PROC_LOPASS: return new RLopass;
    case MLGPU_PROC_HIPASS: return new RHipass;
    case MLGPU_PROC_BANDPASS: return new RBandpass;
    case MLGPU_PROC_LO_SHELF: return new RLoShelf;
    case MLGPU_PROC_HI_SHELF: return new RHiShelf;
    case MLGPU_PROC_BELL: return new RBell;
    case MLGPU_PROC_ONE_POLE: return new ROnePole;
    case MLGPU_PROC_DC_BLOCKER: return new RDCBlocker;
    case MLGPU_PROC_DIFFERENTIATOR: return new RDifferentiator;
    case MLGPU_PROC_INTEGRATOR: return new RIntegrator;
    case MLGPU_PROC_PEAK: return new RPeak;
    case MLGPU_PROC_RMS: return new RRms;
    case MLGPU_PROC_ADSR: return new RAdsr;
    case MLGPU_PROC_GAIN: return new RGain;
    default: return nullptr;
  }
}

// run voices [v0, v1) of a chain
void runVoices(const int32_t* procs, int nProcs, size_t V, size_t T, size_t v0, size_t v1,
               const float* coeffs, uint32_t* state, const float* inSignal, const float* inConst,
               float* out)
{
  std::vector<std::unique_ptr<RefProc>> chain;
  std::vector<int> cOff(nProcs), sOff(nProcs);
  int c = 0, s = 0;
  for (int p = 0; p < nProcs; ++p)
  {
    chain.emplace_back(makeProc(procs[p]));
    cOff[p] = c;
    sOff[p] = s;
    c += chain[p]->nc();
    s += chain[p]->ns();
  }
  const size_t S = T * kFloatsPerDSPVector;
  float cbuf[16];
  uint32_t sbuf[80];
  for (size_t v = v0; v < v1; ++v)
  {
    for (int p = 0; p < nProcs; ++p)
    {
      for (int i = 0; i < chain[p]->nc(); ++i) cbuf[i] = coeffs[(size_t)(cOff[p] + i) * V + v];
      for (int i = 0; i < chain[p]->ns(); ++i) sbuf[i] = state[(size_t)(sOff[p] + i) * V + v];
      chain[p]->setCoeffs(cbuf);
      chain[p]->setState(sbuf);
    }
    for (size_t t = 0; t < T; ++t)
    {
      DSPVector x;
      if (inSignal)
        load(x, inSignal + v * S + t * kFloatsPerDSPVector);
      else if (inConst)
        x = DSPVector(inConst[v]);
      for (int p = 0; p < nProcs; ++p) x = chain[p]->process(x);
      if (out) store(x, out + v * S + t * kFloatsPerDSPVector);
    }
    for (int p = 0; p < nProcs; ++p)
    {
      chain[p]->getState(sbuf);
      for (int i = 0; i < chain[p]->ns(); ++i) state[(size_t)(sOff[p] + i) * V + v] = sbuf[i];
    }
  }
}

template <class Fn>
void parallelFor(size_t V, int nThreads, Fn fn)
{
  if (nThreads <= 1)
  {
    fn(0, V);
    return;
  }
  std::vector<std::thread> th;
  size_t per = (V + nThreads - 1) / nThreads;
  for (int i = 0; i < nThreads; ++i)
  {
    size_t a = std::min(V, per * i), b = std::min(V, per * (i + 1));
    if (a < b) th.emplace_back([=]() { fn(a, b); });
  }
  for (auto& t : th) t.join();
}

// The same for the CPU baselines, timed the way BASELINE.md 3 asks: the threads exist and wait BEFORE the clock starts, are
// released together, and the clock stops when the last one is done (thread creation is not part of the measured work).
template <class Fn>
double timedParallelFor(size_t V, int nThreads, Fn fn)
{
  nThreads = std::max(1, nThreads);
  std::atomic<int> ready{0};
  std::atomic<bool> go{false};
  std::vector<std::thread> th;
  const size_t per = (V + nThreads - 1) / nThreads;
  int started = 0;
  for (int i = 0; i < nThreads; ++i)
  {
    const size_t a = std::min(V, per * i), b = std::min(V, per * (i + 1));
    if (a >= b) continue;
    ++started;
    th.emplace_back(
        [&, a, b]()
        {
          ready.fetch_add(1);
          while (!go.load(std::memory_order_acquire)) {}
          fn(a, b);
        });
  }
  while (ready.load() < started) std::this_thread::yield();
  const auto t0 = std::chrono::steady_clock::now();
  go.store(true, std::memory_order_release);
  for (auto& t : th) t.join();
  const auto t1 = std::chrono::steady_clock::now();
  return std::chrono::duration<double>(t1 - t0).count();
}

}  // namespace

// ml::UsingFlushDenormalsToZero (MLDSPUtils.h:51-96) is scoped to a process function; a test brackets the calls it wants
// in that mode with this pair instead (same MXCSR bits, same thread; threads started by a call inherit them).
extern "C" int mlref_set_flush_denormals(int on)
{
  static thread_local ml::UsingFlushDenormalsToZero* scope = nullptr;
  const bool was = scope != nullptr;
  if (on && !scope) scope = new ml::UsingFlushDenormalsToZero();   // the reference's own constructor sets DAZ | FZ
  if (!on && scope)
  {
    delete scope;                                                   // and its destructor restores the previous MXCSR
    scope = nullptr;
  }
  return was ? 1 : 0;
}

extern "C"
{
  // ---- elementwise ops through the reference's own DSPVector functions ----
  int mlref_op_apply(int op, const void* va, const void* vb, const void* vc, void* vout,
                     size_t nElems)
  {
    if (nElems % kFloatsPerDSPVector) return MLGPU_ERR_INVALID;
    const float* a = (const float*)va;
    const float* b = (const float*)vb;
    const float* c = (const float*)vc;
    float* out = (float*)vout;
    size_t nVec = nElems / kFloatsPerDSPVector;
    for (size_t i = 0; i < nVec; ++i)
    {
      DSPVector x1, x2, x3, y;
      DSPVectorInt i1, i2, i3, yi;
      const size_t o = i * kFloatsPerDSPVector;
      if (a)
      {
        load(x1, a + o);
        std::memcpy(i1.getBuffer(), a + o, 256);
      }
      if (b)
      {
        load(x2, b + o);
        std::memcpy(i2.getBuffer(), b + o, 256);
      }
      if (c)
      {
        load(x3, c + o);
        std::memcpy(i3.getBuffer(), c + o, 256);
      }
      bool intOut = false;
      switch (op)
      {
        case MLGPU_OP_SQRT: y = sqrt(x1); break;
        case MLGPU_OP_SQRT_APPROX: y = sqrtApprox(x1); break;
        case MLGPU_OP_ABS: y = abs(x1); break;
        case MLGPU_OP_SIGN: y = sign(x1); break;
        case MLGPU_OP_SIGN_BIT: y = signBit(x1); break;
        case MLGPU_OP_SIN: y = sin(x1); break;
        case MLGPU_OP_COS: y = cos(x1); break;
        case MLGPU_OP_LOG: y = log(x1); break;
        case MLGPU_OP_EXP: y = exp(x1); break;
        case MLGPU_OP_LOG2: y = log2(x1); break;
        case MLGPU_OP_EXP2: y = exp2(x1); break;
        case MLGPU_OP_SIN_APPROX: y = sinApprox(x1); break;
        case MLGPU_OP_COS_APPROX: y = cosApprox(x1); break;
        case MLGPU_OP_EXP_APPROX: y = expApprox(x1); break;
        case MLGPU_OP_LOG_APPROX: y = logApprox(x1); break;
        case MLGPU_OP_LOG2_APPROX: y = log2Approx(x1); break;
        case MLGPU_OP_EXP2_APPROX: y = exp2Approx(x1); break;
        case MLGPU_OP_FRACTIONAL_PART: y = fractionalPart(x1); break;
        case MLGPU_OP_ROUND_FLOAT_TO_INT:
          yi = roundFloatToInt(x1);
          intOut = true;
          break;
        case MLGPU_OP_TRUNCATE_FLOAT_TO_INT:
          yi = truncateFloatToInt(x1);
          intOut = true;
          break;
        case MLGPU_OP_INT_TO_FLOAT: y = intToFloat(i1); break;
        case MLGPU_OP_UNSIGNED_INT_TO_FLOAT: y = unsignedIntToFloat(i1); break;
        case MLGPU_OP_EXP_APPROX_OF_SIN_APPROX: y = expApprox(sinApprox(x1)); break;
        case MLGPU_OP_ADD: y = add(x1, x2); break;
        case MLGPU_OP_SUBTRACT: y = subtract(x1, x2); break;
        case MLGPU_OP_MULTIPLY: y = multiply(x1, x2); break;
        case MLGPU_OP_DIVIDE: y = divide(x1, x2); break;
        case MLGPU_OP_DIVIDE_APPROX: y = divideApprox(x1, x2); break;
        case MLGPU_OP_POW: y = pow(x1, x2); break;
        case MLGPU_OP_POW_APPROX: y = powApprox(x1, x2); break;
        case MLGPU_OP_MIN: y = min(x1, x2); break;
        case MLGPU_OP_MAX: y = max(x1, x2); break;
        case MLGPU_OP_ADD_INT32:
          yi = addInt32(i1, i2);
          intOut = true;
          break;
        case MLGPU_OP_SUBTRACT_INT32:
          yi = subtractInt32(i1, i2);
          intOut = true;
          break;
        case MLGPU_OP_EQUAL:
          yi = equal(x1, x2);
          intOut = true;
          break;
        case MLGPU_OP_NOT_EQUAL:
          yi = notEqual(x1, x2);
          intOut = true;
          break;
        case MLGPU_OP_GREATER_THAN:
          yi = greaterThan(x1, x2);
          intOut = true;
          break;
        case MLGPU_OP_GREATER_THAN_OR_EQUAL:
          yi = greaterThanOrEqual(x1, x2);
          intOut = true;
          break;
        case MLGPU_OP_LESS_THAN:
          yi = lessThan(x1, x2);
          intOut = true;
          break;
        case MLGPU_OP_LESS_THAN_OR_EQUAL:
          yi = lessThanOrEqual(x1, x2);
          intOut = true;
          break;
        case MLGPU_OP_LERP: y = lerp(x1, x2, x3); break;
        case MLGPU_OP_INVERSE_LERP: y = inverseLerp(x1, x2, x3); break;
        case MLGPU_OP_CLAMP: y = clamp(x1, x2, x3); break;
        case MLGPU_OP_WITHIN: y = within(x1, x2, x3); break;
        case MLGPU_OP_SELECT: y = select(x1, x2, i3); break;
        case MLGPU_OP_SELECT_INT:
          yi = select(i1, i2, i3);
          intOut = true;
          break;
        case MLGPU_OP_PHASOR_TO_SINE: y = phasorToSine(x1); break;
        case MLGPU_OP_PHASOR_TO_SAW: y = phasorToSaw(x1, x2); break;
        case MLGPU_OP_PHASOR_TO_PULSE: y = phasorToPulse(x1, x2, x3); break;
        default: return MLGPU_ERR_INVALID;
      }
      if (intOut)
        std::memcpy(out + o, yi.getConstBuffer(), 256);
      else
        store(y, out + o);
    }
    return MLGPU_OK;
  }

  int mlref_op_apply_rows1(int op, const float* a, const float* b64, float* out, size_t nRows)
  {
    DSPVector x2;
    load(x2, b64);
    for (size_t r = 0; r < nRows; ++r)
    {
      DSPVector x1, y;
      load(x1, a + r * 64);
      switch (op)
      {
        case MLGPU_OP_ADD: y = add1(x1, x2); break;
        case MLGPU_OP_SUBTRACT: y = subtract1(x1, x2); break;
        case MLGPU_OP_MULTIPLY: y = multiply1(x1, x2); break;
        case MLGPU_OP_DIVIDE: y = divide1(x1, x2); break;
        case MLGPU_OP_DIVIDE_APPROX: y = divideApprox1(x1, x2); break;
        case MLGPU_OP_POW: y = pow1(x1, x2); break;
        case MLGPU_OP_POW_APPROX: y = powApprox1(x1, x2); break;
        case MLGPU_OP_MIN: y = min1(x1, x2); break;
        case MLGPU_OP_MAX: y = max1(x1, x2); break;
        default: return MLGPU_ERR_INVALID;
      }
      store(y, out + r * 64);
    }
    return MLGPU_OK;
  }

  int mlref_row_reduce(int rowop, const float* rows, float* out, size_t nRows)
  {
    for (size_t r = 0; r < nRows; ++r)
    {
      DSPVector x;
      load(x, rows + r * 64);
      switch (rowop)
      {
        case MLGPU_ROWOP_SUM: out[r] = sum(x); break;
        case MLGPU_ROWOP_MEAN: out[r] = mean(x); break;
        case MLGPU_ROWOP_MAX: out[r] = max(x); break;
        case MLGPU_ROWOP_MIN: out[r] = min(x); break;
        default: return MLGPU_ERR_INVALID;
      }
    }
    return MLGPU_OK;
  }

  // ---- chains of reference processors ----
  int mlref_proc_num_coeffs(int kind)
  {
    std::unique_ptr<RefProc> p(makeProc(kind));
    return p ? p->nc() : -1;
  }
  int mlref_proc_num_state(int kind)
  {
    std::unique_ptr<RefProc> p(makeProc(kind));
    return p ? p->ns() : -1;
  }

  // state <- T::clear() for each processor; state is [totalNS][V]
  int mlref_chain_clear(const int32_t* procs, int nProcs, size_t V, uint32_t* state)
  {
    int s = 0;
    uint32_t sbuf[80];
    for (int p = 0; p < nProcs; ++p)
    {
      std::unique_ptr<RefProc> rp(makeProc(procs[p]));
      if (!rp) return MLGPU_ERR_INVALID;
      rp->clear();
      rp->getState(sbuf);
      // ADSR::clear() only sets segment=off; the other words keep their values. For a
      // fresh object they are zero, which is what we report.
      for (int i = 0; i < rp->ns(); ++i)
        for (size_t v = 0; v < V; ++v) state[(size_t)(s + i) * V + v] = sbuf[i];
      s += rp->ns();
    }
    return MLGPU_OK;
  }

  // default-constructed state of the reference objects (== what a new bank holds)
  int mlref_chain_default_state(const int32_t* procs, int nProcs, size_t V, uint32_t* state)
  {
    int s = 0;
    uint32_t sbuf[80];
    for (int p = 0; p < nProcs; ++p)
    {
      std::unique_ptr<RefProc> rp(makeProc(procs[p]));
      if (!rp) return MLGPU_ERR_INVALID;
      rp->getState(sbuf);
      for (int i = 0; i < rp->ns(); ++i)
        for (size_t v = 0; v < V; ++v) state[(size_t)(s + i) * V + v] = sbuf[i];
      s += rp->ns();
    }
    return MLGPU_OK;
  }

  // coeffs [totalNC][V], state [totalNS][V] (in/out), inSignal [V][64T] or NULL,
  // inConst [V] or NULL, out [V][64T] (may be NULL for timing runs)
  int mlref_chain_process(const int32_t* procs, int nProcs, size_t V, size_t T,
                          const float* coeffs, uint32_t* state, const float* inSignal,
                          const float* inConst, float* out, int nThreads)
  {
    for (int p = 0; p < nProcs; ++p)
    {
      std::unique_ptr<RefProc> rp(makeProc(procs[p]));
      if (!rp) return MLGPU_ERR_INVALID;
    }
    parallelFor(V, nThreads, [=](size_t a, size_t b)
                { runVoices(procs, nProcs, V, T, a, b, coeffs, state, inSignal, inConst, out); });
    return MLGPU_OK;
  }

  // PulseGen with an audio-rate width input: PulseGen::operator()(freq, width)
  int mlref_pulse2_process(size_t V, size_t T, uint32_t* omega32, const float* freq, const float* width, float* out)
  {
    const size_t S = T * kFloatsPerDSPVector;
    for (size_t v = 0; v < V; ++v)
    {
      PulseGen g;
      g._phasor.mOmega32 = omega32[v];
      for (size_t t = 0; t < T; ++t)
      {
        DSPVector f, w;
        load(f, freq + v * S + t * kFloatsPerDSPVector);
        load(w, width + v * S + t * kFloatsPerDSPVector);
        store(g(f, w), out + v * S + t * kFloatsPerDSPVector);
      }
      omega32[v] = g._phasor.mOmega32;
    }
    return MLGPU_OK;
  }

  // ---- CPU baselines: the bench chains written exactly as user code would ----
  // config 3: y = bp(saw(freq)) * gain  (SURVEY §3.2). Each thread owns a contiguous
  // range of voices as an array of {SawGen, Bandpass}; loops vectors x voices.
  // Returns seconds of wall time for T vectors over V voices. `sink` receives a checksum
  // so the work cannot be optimised away; out may be NULL.
  double mlref_bench_saw_bandpass_gain(size_t V, size_t T, const float* freq, const float* g0,
                                       const float* g1, const float* g2, float gain,
                                       int nThreads, float* out, double* sink)
  {
    struct Voice
    {
      SawGen saw;
      Bandpass bp;
    };
    std::vector<Voice> voices(V);
    for (size_t v = 0; v < V; ++v)
    {
      voices[v].saw.clear();
      voices[v].bp.coeffs = {g0[v], g1[v], g2[v]};
    }
    std::vector<double> partial(std::max(1, nThreads), 0.0);
    const size_t S = T * kFloatsPerDSPVector;
    std::atomic<int> tid{0};
    const double seconds = timedParallelFor(V, nThreads,
                [&](size_t a, size_t b)
                {
                  int me = tid.fetch_add(1);
                  double acc = 0;
                  for (size_t t = 0; t < T; ++t)
                  {
                    for (size_t v = a; v < b; ++v)
                    {
                      DSPVector y = voices[v].bp(voices[v].saw(DSPVector(freq[v]))) * gain;
                      if (out) store(y, out + v * S + t * kFloatsPerDSPVector);
                      acc += y[63];
                    }
                  }
                  partial[me] = acc;
                });
    double s = 0;
    for (double p : partial) s += p;
    if (sink) *sink = s;
    return seconds;
  }

  // config 4: 8 cascaded Lopass sections over a noise input (NoiseGen seeded per channel).
  double mlref_bench_lopass_cascade8(size_t V, size_t T, const float* coeffs /*[8][3]*/,
                                     int nThreads, double* sink)
  {
    struct Chan
    {
      NoiseGen noise;
      Lopass lp[8];
    };
    std::vector<Chan> ch(V);
    for (size_t v = 0; v < V; ++v)
    {
      ch[v].noise.setSeed((uint32_t)v);
      for (int i = 0; i < 8; ++i) ch[v].lp[i].coeffs = {coeffs[i * 3], coeffs[i * 3 + 1], coeffs[i * 3 + 2]};
    }
    std::vector<double> partial(std::max(1, nThreads), 0.0);
    std::atomic<int> tid{0};
    const double seconds = timedParallelFor(V, nThreads,
                [&](size_t a, size_t b)
                {
                  int me = tid.fetch_add(1);
                  double acc = 0;
                  for (size_t t = 0; t < T; ++t)
                  {
                    for (size_t v = a; v < b; ++v)
                    {
                      DSPVector y = ch[v].noise();
                      for (int i = 0; i < 8; ++i) y = ch[v].lp[i](y);
                      acc += y[63];
                    }
                  }
                  partial[me] = acc;
                });
    double s = 0;
    for (double p : partial) s += p;
    if (sink) *sink = s;
    return seconds;
  }

  // config 5: the 16-node synth voice of madronalib_amd/patches.py synth16(), written the way user code writes it with the
  // reference's objects (one struct of processors per voice, one DSPVector at a time). params [5][V] = pitch, baseFreq,
  // width, lfoFreq, noiseLevel; coefficient arrays [n][V]; gate / out [V][64 T]. All objects start cleared (clear() where
  // the class has one), NoiseGen seeded per voice. nThreads > 0: voices split over threads; returns seconds.
  namespace
  {
  struct Synth16Voice
  {
    SawGen saw;
    PulseGen pulse;
    SineGen lfo;
    NoiseGen noise;
    Lopass lp;
    Hipass hp;
    OnePole smooth;
    DCBlocker dc;
    ADSR env;
    float pitch, baseFreq, width, lfoFreq, noiseLevel;
    DSPVector process(const DSPVector gate)
    {
      const DSPVector ratio = exp2Approx(DSPVector(pitch));
      const DSPVector freq = ratio * DSPVector(baseFreq);
      const DSPVector vSaw = saw(freq);
      const DSPVector vPulse = pulse(freq, DSPVector(width));
      const DSPVector vLfo = lfo(DSPVector(lfoFreq));
      const DSPVector vNoise = noise();
      const DSPVector osc = vSaw + vPulse * vLfo;
      const DSPVector pre = osc + vNoise * DSPVector(noiseLevel);
      const DSPVector filtered = dc(smooth(hp(lp(pre))));
      return clamp(filtered * env(gate), DSPVector(-1.f), DSPVector(1.f));
    }
  };
  }  // namespace
  double mlref_synth16_run(size_t V, size_t T, const float* params, const float* lpC, const float* hpC, const float* smoothC, const float* dcC,
                           const float* envC, const uint32_t* seeds, const float* gate, float* out, int nThreads)
  {
    std::vector<Synth16Voice> vs(V);
    for (size_t v = 0; v < V; ++v)
    {
      Synth16Voice& s = vs[v];
      s.saw.clear();
      s.pulse.clear();
      s.lfo.clear();
      s.lp.clear();
      s.smooth.clear();
      s.env.clear();
      s.noise.setSeed(seeds[v]);
      s.pitch = params[0 * V + v];
      s.baseFreq = params[1 * V + v];
      s.width = params[2 * V + v];
      s.lfoFreq = params[3 * V + v];
      s.noiseLevel = params[4 * V + v];
      s.lp.coeffs = {lpC[0 * V + v], lpC[1 * V + v], lpC[2 * V + v]};
      s.hp.coeffs = {hpC[0 * V + v], hpC[1 * V + v], hpC[2 * V + v], hpC[3 * V + v]};
      s.smooth.coeffs = {smoothC[0 * V + v], smoothC[1 * V + v]};
      s.dc.coeffs = dcC[v];
      s.env.coeffs = {envC[0 * V + v], envC[1 * V + v], envC[2 * V + v], envC[3 * V + v]};
    }
    const double seconds = timedParallelFor(V, std::max(1, nThreads),
                [&](size_t a, size_t b)
                {
                  for (size_t t = 0; t < T; ++t)
                    for (size_t v = a; v < b; ++v)
                    {
                      DSPVector g;
                      load(g, gate + (v * T + t) * kFloatsPerDSPVector);
                      const DSPVector y = vs[v].process(g);
                      store(y, out + (v * T + t) * kFloatsPerDSPVector);
                    }
                });
    return seconds;
  }

  // the same voice as SURVEY §8d lists it (patches.synth16(full=True)): a filter envelope, the cutoff per sample through
  // exp2Approx, and the Lopass in its per-sample-coefficient form (MLDSPFilters.h:136-152). params [9][V] = pitch, baseFreq,
  // width, lfoFreq, noiseLevel, cutoffOct, envAmount, cutoffBase, resonance.
  namespace
  {
  struct Synth16FullVoice
  {
    SawGen saw;
    PulseGen pulse;
    SineGen lfo;
    NoiseGen noise;
    Lopass lp;
    Hipass hp;
    OnePole smooth;
    DCBlocker dc;
    ADSR env, fenv;
    float pitch, baseFreq, width, lfoFreq, noiseLevel, cutoffOct, envAmount, cutoffBase, resonance;
    DSPVector process(const DSPVector gate)
    {
      const DSPVector freq = exp2Approx(DSPVector(pitch)) * DSPVector(baseFreq);
      const DSPVector vSaw = saw(freq);
      const DSPVector vPulse = pulse(freq, DSPVector(width));
      const DSPVector vLfo = lfo(DSPVector(lfoFreq));
      const DSPVector vNoise = noise();
      const DSPVector pre = (vSaw + vPulse * vLfo) + vNoise * DSPVector(noiseLevel);
      const DSPVector omega = exp2Approx(DSPVector(cutoffOct) + fenv(gate) * DSPVector(envAmount)) * DSPVector(cutoffBase);
      const DSPVector filtered = dc(smooth(hp(lp(pre, omega, DSPVector(resonance)))));
      return clamp(filtered * env(gate), DSPVector(-1.f), DSPVector(1.f));
    }
  };
  }  // namespace
  double mlref_synth16full_run(size_t V, size_t T, const float* params, const float* hpC, const float* smoothC, const float* dcC, const float* envC,
                               const float* fenvC, const uint32_t* seeds, const float* gate, float* out, int nThreads)
  {
    std::vector<Synth16FullVoice> vs(V);
    for (size_t v = 0; v < V; ++v)
    {
      Synth16FullVoice& s = vs[v];
      s.saw.clear();
      s.pulse.clear();
      s.lfo.clear();
      s.lp.clear();
      s.smooth.clear();
      s.env.clear();
      s.fenv.clear();
      s.noise.setSeed(seeds[v]);
      float* p[9] = {&s.pitch, &s.baseFreq, &s.width, &s.lfoFreq, &s.noiseLevel, &s.cutoffOct, &s.envAmount, &s.cutoffBase, &s.resonance};
      for (int i = 0; i < 9; ++i) *p[i] = params[(size_t)i * V + v];
      s.hp.coeffs = {hpC[0 * V + v], hpC[1 * V + v], hpC[2 * V + v], hpC[3 * V + v]};
      s.smooth.coeffs = {smoothC[0 * V + v], smoothC[1 * V + v]};
      s.dc.coeffs = dcC[v];
      s.env.coeffs = {envC[0 * V + v], envC[1 * V + v], envC[2 * V + v], envC[3 * V + v]};
      s.fenv.coeffs = {fenvC[0 * V + v], fenvC[1 * V + v], fenvC[2 * V + v], fenvC[3 * V + v]};
    }
    const double seconds = timedParallelFor(V, std::max(1, nThreads),
                [&](size_t a, size_t b)
                {
                  for (size_t t = 0; t < T; ++t)
                    for (size_t v = a; v < b; ++v)
                    {
                      DSPVector g;
                      load(g, gate + (v * T + t) * kFloatsPerDSPVector);
                      const DSPVector y = vs[v].process(g);
                      store(y, out + (v * T + t) * kFloatsPerDSPVector);
                    }
                });
    return seconds;
  }

  // config 2: elementwise op over n elements, nThreads; returns seconds.
  double mlref_bench_op(int op, const float* in, float* out, size_t nElems, int nThreads, int reps)
  {
    size_t nVec = nElems / kFloatsPerDSPVector;
    const double seconds = timedParallelFor(nVec, nThreads,
                [&](size_t a, size_t b)
                {
                  for (int r = 0; r < reps; ++r)
                    mlref_op_apply(op, in + a * 64, nullptr, nullptr, out + a * 64, (b - a) * 64);
                });
    return seconds;
  }

  // ---- coefficient makers straight from the reference ----
  void mlref_lopass_make_coeffs(float omega, float k, float* o)
  {
    auto c = Lopass::makeCoeffs(omega, k);
    o[0] = c[0];
    o[1] = c[1];
    o[2] = c[2];
  }
  void mlref_hipass_make_coeffs(float omega, float k, float* o)
  {
    auto c = Hipass::makeCoeffs(omega, k);
    o[0] = c.g0;
    o[1] = c.g1;
    o[2] = c.g2;
    o[3] = c.k;
  }
  void mlref_bandpass_make_coeffs(float omega, float k, float* o)
  {
    auto c = Bandpass::makeCoeffs(omega, k);
    o[0] = c.g0;
    o[1] = c.g1;
    o[2] = c.g2;
  }
  void mlref_loshelf_make_coeffs(float omega, float k, float A, float* o)
  {
    auto c = LoShelf::makeCoeffs({omega, k, A});
    for (int i = 0; i < 5; ++i) o[i] = c[i];
  }
  void mlref_hishelf_make_coeffs(float omega, float k, float A, float* o)
  {
    auto c = HiShelf::makeCoeffs({omega, k, A});
    for (int i = 0; i < 6; ++i) o[i] = c[i];
  }
  void mlref_bell_make_coeffs(float omega, float k, float A, float* o)
  {
    auto c = Bell::makeCoeffs(omega, k, A);
    o[0] = c.a1;
    o[1] = c.a2;
    o[2] = c.a3;
    o[3] = c.m1;
  }
  void mlref_onepole_make_coeffs(float omega, float* o)
  {
    auto c = OnePole::makeCoeffs(omega);
    o[0] = c.a0;
    o[1] = c.b1;
  }
  float mlref_dcblocker_make_coeffs(float omega) { return DCBlocker::makeCoeffs(omega); }
  void mlref_adsr_calc_coeffs(float a, float d, float s, float r, float sr, float* o)
  {
    auto c = ADSR::calcCoeffs(a, d, s, r, sr);
    o[0] = c.ka;
    o[1] = c.kd;
    o[2] = c.s;
    o[3] = c.kr;
  }
  float mlref_db_to_gain(float dB) { return dBToGain(dB); }

  // the ImpulseGen windowed-sinc table as the reference's constructor builds it
  void mlref_impulse_table(float* out17)
  {
    ImpulseGen g;
    for (int i = 0; i < 17; ++i) out17[i] = g._table[i];
  }

  // ---- the other operator() forms + vector-rate ramps: same contract as mlorc_proc_process_multi ----
  int mlref_proc_process_multi(int kind, size_t V, size_t T, const float* coeffs, uint32_t* state,
                               const float* const* inputs, int nInputs, float* out)
  {
    const size_t S = T * kFloatsPerDSPVector;
    float cbuf[16];
    uint32_t sbuf[80];
    for (size_t v = 0; v < V; ++v)
    {
      std::unique_ptr<RefProc> rp(makeProc(kind));
      if (!rp) return MLGPU_ERR_INVALID;
      for (int i = 0; i < rp->nc(); ++i) cbuf[i] = coeffs[(size_t)i * V + v];
      for (int i = 0; i < rp->ns(); ++i) sbuf[i] = state[(size_t)i * V + v];
      rp->setCoeffs(cbuf);
      rp->setState(sbuf);
      for (size_t t = 0; t < T; ++t)
      {
        DSPVector in[8], y;
        for (int i = 0; i < nInputs; ++i) load(in[i], inputs[i] + v * S + t * kFloatsPerDSPVector);
        if (kind == MLGPU_PROC_PULSE_GEN && nInputs == 2)
          y = static_cast<RPulse*>(rp.get())->g(in[0], in[1]);
        else if (kind == MLGPU_PROC_LOPASS && nInputs == 3)
          y = static_cast<RLopass*>(rp.get())->f(in[0], in[1], in[2]);
        else if (kind == MLGPU_PROC_LO_SHELF && nInputs == 6)
        {
          DSPVectorArray<5> vc;
          for (int r = 0; r < 5; ++r) vc.row(r) = in[1 + r];
          y = static_cast<RLoShelf*>(rp.get())->f(in[0], vc);
        }
        else if (kind == MLGPU_PROC_HI_SHELF && nInputs == 7)
        {
          DSPVectorArray<6> vc;
          for (int r = 0; r < 6; ++r) vc.row(r) = in[1 + r];
          y = static_cast<RHiShelf*>(rp.get())->f(in[0], vc);
        }
        else if (kind == MLGPU_PROC_TEMPO_LOCK && nInputs == 3)
          y = static_cast<RTempoLock*>(rp.get())->g(in[0], in[1][0], in[2][0]);
        else if (nInputs <= 1)
          y = rp->process(nInputs ? in[0] : DSPVector(0.f));
        else
          return MLGPU_ERR_UNSUPPORTED;
        store(y, out + v * S + t * kFloatsPerDSPVector);
      }
      rp->getState(sbuf);
      for (int i = 0; i < rp->ns(); ++i) state[(size_t)i * V + v] = sbuf[i];
    }
    return MLGPU_OK;
  }

  int mlref_vop(int vop, size_t V, size_t T, const float* a, const float* b, float* out)
  {
    for (size_t r = 0; r < V * T; ++r)
    {
      const float start = a ? a[r * kFloatsPerDSPVector] : 0.f, end = b ? b[r * kFloatsPerDSPVector] : 0.f;
      DSPVector y;
      switch (vop)
      {
        case MLGPU_VOP_COLUMN_INDEX: y = columnIndex(); break;
        case MLGPU_VOP_RANGE_OPEN: y = rangeOpen(start, end); break;
        case MLGPU_VOP_RANGE_CLOSED: y = rangeClosed(start, end); break;
        case MLGPU_VOP_INTERPOLATE_LINEAR: y = interpolateDSPVectorLinear(start, end); break;
        default: return MLGPU_ERR_INVALID;
      }
      store(y, out + r * kFloatsPerDSPVector);
    }
    return MLGPU_OK;
  }

  void mlref_linear_glide_make_coeffs(float t, float* o)
  {
    LinearGlide g;
    g.setGlideTimeInSamples(t);
    const int32_t n = g.mVectorsPerGlide;
    std::memcpy(&o[0], &n, 4);
    o[1] = g.mDyPerVector;
  }
  void mlref_sample_accurate_linear_glide_make_coeffs(float t, float* o)
  {
    SampleAccurateLinearGlide g;
    g.setGlideTimeInSamples(t);
    const int32_t n = g.mSamplesPerGlide;
    std::memcpy(&o[0], &n, 4);
    o[1] = g.mDyPerSample;
  }

  // ---- delay lines: the reference objects themselves, same contract as mlorc_delay_process ----
  namespace
  {
  void loadRing(IntegerDelay& d, size_t len, uint32_t w, int32_t delay, const float* ring)
  {
    d.setMaxDelayInSamples((float)(len - kFloatsPerDSPVector));  // allocates exactly `len` samples (:823-831)
    std::memcpy(d.mBuffer.data(), ring, sizeof(float) * len);
    d.mWriteIndex = w;
    d.mIntDelayInSamples = delay;
  }
  void storeRing(const IntegerDelay& d, size_t len, uint32_t* w, uint32_t* delay, float* ring)
  {
    std::memcpy(ring, d.mBuffer.data(), sizeof(float) * len);
    *w = (uint32_t)d.mWriteIndex;
    *delay = (uint32_t)d.mIntDelayInSamples;
  }
  void loadFrac(FractionalDelay& f, size_t len, const uint32_t* S, const float* ring)
  {
    loadRing(f.mIntegerDelay, len, S[0], (int32_t)S[3], ring);
    f.mAllpassSection.x1 = u2f(S[1]);
    f.mAllpassSection.y1 = u2f(S[2]);
    f.mAllpassSection.coeffs = u2f(S[4]);
  }
  void storeFrac(const FractionalDelay& f, size_t len, uint32_t* S, float* ring)
  {
    storeRing(f.mIntegerDelay, len, &S[0], &S[3], ring);
    S[1] = f2u(f.mAllpassSection.x1);
    S[2] = f2u(f.mAllpassSection.y1);
    S[4] = f2u(f.mAllpassSection.coeffs);
  }
  }  // namespace

  int mlref_delay_process(int kind, size_t V, size_t T, uint32_t* state, float* mem, size_t len, const float* const* inputs, int nInputs,
                          float* out)
  {
    const int ns = (kind == MLGPU_PROC_INTEGER_DELAY) ? 2 : (kind == MLGPU_PROC_FRACTIONAL_DELAY ? 5 : 10);
    const int rings = (kind == MLGPU_PROC_PITCHBENDABLE_DELAY) ? 2 : 1;
    const size_t S = T * kFloatsPerDSPVector;
    uint32_t St[16];
    for (size_t v = 0; v < V; ++v)
    {
      for (int i = 0; i < ns; ++i) St[i] = state[(size_t)i * V + v];
      float* ring = mem + v * (size_t)rings * len;
      IntegerDelay idl;
      FractionalDelay fdl;
      PitchbendableDelay pdl;
      if (kind == MLGPU_PROC_INTEGER_DELAY) loadRing(idl, len, St[0], (int32_t)St[1], ring);
      else if (kind == MLGPU_PROC_FRACTIONAL_DELAY) loadFrac(fdl, len, St, ring);
      else
      {
        loadFrac(pdl.mDelay1, len, St, ring);
        loadFrac(pdl.mDelay2, len, St + 5, ring + len);
      }
      for (size_t t = 0; t < T; ++t)
      {
        DSPVector in[3], y;
        for (int i = 0; i < nInputs; ++i) load(in[i], inputs[i] + v * S + t * kFloatsPerDSPVector);
        if (kind == MLGPU_PROC_INTEGER_DELAY) y = (nInputs == 1) ? idl(in[0]) : idl(in[0], in[1]);
        else if (kind == MLGPU_PROC_FRACTIONAL_DELAY)
        {
          if (nInputs == 1) y = fdl(in[0]);
          else if (nInputs == 2) y = fdl(in[0], in[1]);
          else
          {
            DSPVectorInt ticks;
            for (int n = 0; n < kFloatsPerDSPVector; ++n) ticks[n] = (int32_t)f2u(in[2][n]);
            y = fdl(in[0], in[1], ticks);
          }
        }
        else y = pdl(in[0], in[1]);
        store(y, out + v * S + t * kFloatsPerDSPVector);
      }
      if (kind == MLGPU_PROC_INTEGER_DELAY) storeRing(idl, len, &St[0], &St[1], ring);
      else if (kind == MLGPU_PROC_FRACTIONAL_DELAY) storeFrac(fdl, len, St, ring);
      else
      {
        storeFrac(pdl.mDelay1, len, St, ring);
        storeFrac(pdl.mDelay2, len, St + 5, ring + len);
      }
      for (int i = 0; i < ns; ++i) state[(size_t)i * V + v] = St[i];
    }
    return MLGPU_OK;
  }
  float mlref_allpass1_make_coeffs(float d) { return Allpass1::makeCoeffs(d); }
  void mlref_fractional_delay_make_state(float d, float* o)
  {
    FractionalDelay f;
    f.setDelayInSamples(d);
    const int32_t di = f.mIntegerDelay.mIntDelayInSamples;
    std::memcpy(&o[0], &di, 4);
    o[1] = f.mAllpassSection.coeffs;
  }

  // ---- composites built on one-vector feedback: the reference classes run as user code would ----
  // which: 0 Allpass<IntegerDelay>, 1 Allpass<FractionalDelay> (constant delay d), 2 Allpass<PitchbendableDelay> (delay signal)
  int mlref_allpass_run(int which, size_t T, float gain, float maxDelay, float d, const float* delaySig, const float* in, float* out)
  {
    Allpass<IntegerDelay> a0;
    Allpass<FractionalDelay> a1;
    Allpass<PitchbendableDelay> a2;
    a0.mGain = a1.mGain = a2.mGain = gain;
    a0.setMaxDelayInSamples(maxDelay);
    a1.setMaxDelayInSamples(maxDelay);
    a2.setMaxDelayInSamples(maxDelay);
    a0.setDelayInSamples(d);
    a1.setDelayInSamples(d);
    for (size_t t = 0; t < T; ++t)
    {
      DSPVector x, dl, y;
      load(x, in + t * kFloatsPerDSPVector);
      if (which == 2) load(dl, delaySig + t * kFloatsPerDSPVector);
      y = (which == 0) ? a0(x) : (which == 1 ? a1(x) : a2(x, dl));
      store(y, out + t * kFloatsPerDSPVector);
    }
    return MLGPU_OK;
  }
  // FDN<4>: delay times, OnePole cutoffs, feedback gains; the class has no way to allocate its IntegerDelays
  // (MLDSPFilters.h:1163-1181 never calls setMaxDelayInSamples), so the harness does it through the private member
  int mlref_fdn4_run(size_t T, const float* times, const float* omegas, const float* gains, float maxDelay, const float* in, float* outL,
                     float* outR)
  {
    FDN<4> fdn;
    for (auto& d : fdn.mDelays) d.setMaxDelayInSamples(maxDelay);
    fdn.setDelaysInSamples({times[0], times[1], times[2], times[3]});
    fdn.setFilterCutoffs({omegas[0], omegas[1], omegas[2], omegas[3]});
    fdn.mFeedbackGains = {gains[0], gains[1], gains[2], gains[3]};
    for (size_t t = 0; t < T; ++t)
    {
      DSPVector x;
      load(x, in + t * kFloatsPerDSPVector);
      DSPVectorArray<2> y = fdn(x);
      store(y.constRow(0), outL + t * kFloatsPerDSPVector);
      store(y.constRow(1), outR + t * kFloatsPerDSPVector);
    }
    return MLGPU_OK;
  }
  // FeedbackDelayFunction around fn = Lopass(coeffs): y = fn(x + vy1*gain); vy1 = PitchbendableDelay(y, delay - 64)
  int mlref_feedback_delay_run(size_t T, float feedbackGain, float maxDelay, const float* lopassCoeffs, const float* delaySig, const float* in,
                               float* out)
  {
    FeedbackDelayFunction f;
    f.feedbackGain = feedbackGain;
    f.mDelays[0].setMaxDelayInSamples(maxDelay);
    Lopass lp;
    lp.coeffs = {lopassCoeffs[0], lopassCoeffs[1], lopassCoeffs[2]};
    for (size_t t = 0; t < T; ++t)
    {
      DSPVector x, dl;
      load(x, in + t * kFloatsPerDSPVector);
      load(dl, delaySig + t * kFloatsPerDSPVector);
      DSPVector y = f(x, [&](const DSPVector v) { return lp(v); }, dl);
      store(y, out + t * kFloatsPerDSPVector);
    }
    return MLGPU_OK;
  }

  // Upsample2xFunction<2> / Downsample2xFunction<2> (MLDSPFunctional.h:114-213) around one stateful function, then a gain:
  //   fn(v) = Lopass(coeffs)((clamp(v.row(0) * 3, -1, 1) + SawGen(freq)) * v.row(1));   out = F(fn, {x, m}) * 0.5
  // The objects fn uses live across its calls, as in user code (one fn called twice per vector / once per two vectors).
  int mlref_rate_function_run(int up, size_t V, size_t T, const float* freq, const float* lopassCoeffs, const float* x, const float* m, float* out)
  {
    for (size_t v = 0; v < V; ++v)
    {
      Upsample2xFunction<2> upFn;
      Downsample2xFunction<2> downFn;
      SawGen saw;
      Lopass lp;
      lp.coeffs = {lopassCoeffs[0], lopassCoeffs[1], lopassCoeffs[2]};
      const float f = freq[v];
      auto fn = [&](const DSPVectorArray<2> vv) {
        const DSPVector sat = clamp(vv.constRow(0) * DSPVector(3.0f), DSPVector(-1.0f), DSPVector(1.0f));
        const DSPVector mix = sat + saw(DSPVector(f));
        return lp(mix * vv.constRow(1));
      };
      for (size_t t = 0; t < T; ++t)
      {
        DSPVector vx, vm;
        load(vx, x + (v * T + t) * kFloatsPerDSPVector);
        load(vm, m + (v * T + t) * kFloatsPerDSPVector);
        const DSPVectorArray<2> in = concatRows(vx, vm);
        const DSPVector y = up ? upFn(fn, in) : downFn(fn, in);
        store(y * DSPVector(0.5f), out + (v * T + t) * kFloatsPerDSPVector);
      }
    }
    return MLGPU_OK;
  }

  // The same two function objects around a function that keeps a DSPVector between ITS calls: fn = Allpass<IntegerDelay>
  // (gain, delay; its vy1 is one-vector feedback at fn's own rate) followed by a OnePole. x / out [V][64 T].
  int mlref_rate_allpass_run(int up, size_t V, size_t T, float gain, float maxDelay, float delay, const float* onePoleCoeffs, const float* x, float* out)
  {
    for (size_t v = 0; v < V; ++v)
    {
      Upsample2xFunction<1> upFn;
      Downsample2xFunction<1> downFn;
      Allpass<IntegerDelay> ap;
      ap.mGain = gain;
      ap.setMaxDelayInSamples(maxDelay);
      ap.setDelayInSamples(delay);
      OnePole lp;
      lp.coeffs = {onePoleCoeffs[0], onePoleCoeffs[1]};
      auto fn = [&](const DSPVector vx) { return lp(ap(vx)); };
      for (size_t t = 0; t < T; ++t)
      {
        DSPVector vx;
        load(vx, x + (v * T + t) * kFloatsPerDSPVector);
        const DSPVector y = up ? upFn(fn, vx) : downFn(fn, vx);
        store(y, out + (v * T + t) * kFloatsPerDSPVector);
      }
    }
    return MLGPU_OK;
  }

  // Nested rate functions: out = Outer(mid, x) with mid(v) = OnePole(Inner(fn, v)) + v * 0.5 and fn(w) = Lopass(clamp(w * 3, -1, 1));
  // Outer / Inner each an Upsample2xFunction<1> (up = 1) or a Downsample2xFunction<1> (up = 0). x / out [V][64 T].
  int mlref_rate_nested_run(int outerUp, int innerUp, size_t V, size_t T, const float* lopassCoeffs, const float* onePoleCoeffs, const float* x, float* out)
  {
    for (size_t v = 0; v < V; ++v)
    {
      Upsample2xFunction<1> upO, upI;
      Downsample2xFunction<1> downO, downI;
      Lopass lp;
      lp.coeffs = {lopassCoeffs[0], lopassCoeffs[1], lopassCoeffs[2]};
      OnePole op;
      op.coeffs = {onePoleCoeffs[0], onePoleCoeffs[1]};
      auto fn = [&](const DSPVector w) { return lp(clamp(w * DSPVector(3.0f), DSPVector(-1.0f), DSPVector(1.0f))); };
      auto mid = [&](const DSPVector vv)
      {
        const DSPVector inner = innerUp ? upI(fn, vv) : downI(fn, vv);
        return op(inner) + vv * DSPVector(0.5f);
      };
      for (size_t t = 0; t < T; ++t)
      {
        DSPVector vx;
        load(vx, x + (v * T + t) * kFloatsPerDSPVector);
        const DSPVector y = outerUp ? upO(mid, vx) : downO(mid, vx);
        store(y, out + (v * T + t) * kFloatsPerDSPVector);
      }
    }
    return MLGPU_OK;
  }

  // ---- Downsampler / Upsampler: the reference classes driven vector by vector, same contract as mlorc_resample ----
  namespace
  {
  void hbLoad(HalfBandFilter& h, const float* st, size_t V)
  {
    h.apa0.x1 = st[0 * V]; h.apa0.y1 = st[1 * V]; h.apa1.x1 = st[2 * V]; h.apa1.y1 = st[3 * V];
    h.apb0.x1 = st[4 * V]; h.apb0.y1 = st[5 * V]; h.apb1.x1 = st[6 * V]; h.apb1.y1 = st[7 * V];
    h.b1 = st[8 * V];
  }
  void hbStore(const HalfBandFilter& h, float* st, size_t V)
  {
    st[0 * V] = h.apa0.x1; st[1 * V] = h.apa0.y1; st[2 * V] = h.apa1.x1; st[3 * V] = h.apa1.y1;
    st[4 * V] = h.apb0.x1; st[5 * V] = h.apb0.y1; st[6 * V] = h.apb1.x1; st[7 * V] = h.apb1.y1;
    st[8 * V] = h.b1;
  }
  }  // namespace
  int mlref_resample(int octaves, int up, size_t V, size_t Tin, float* state, const float* in, float* out)
  {
    const size_t R = (size_t)1 << octaves, Sin = Tin * kFloatsPerDSPVector, Sout = up ? Sin * R : Sin / R;
    for (size_t v = 0; v < V; ++v)
    {
      if (up)
      {
        Upsampler u(octaves);
        for (int h = 0; h < octaves; ++h) hbLoad(u._filters[h], state + (size_t)h * 9 * V + v, V);
        size_t o = 0;
        for (size_t t = 0; t < Tin; ++t)
        {
          DSPVector x;
          load(x, in + v * Sin + t * kFloatsPerDSPVector);
          if (octaves == 0) { store(x, out + v * Sout + o); o += kFloatsPerDSPVector; continue; }  // Upsampler(0) has no buffers
          u.write(x);
          for (size_t r = 0; r < R; ++r, o += kFloatsPerDSPVector) store(u.read(), out + v * Sout + o);
        }
        for (int h = 0; h < octaves; ++h) hbStore(u._filters[h], state + (size_t)h * 9 * V + v, V);
      }
      else
      {
        Downsampler d(octaves);
        for (int h = 0; h < octaves; ++h) hbLoad(d._filters[h], state + (size_t)h * 9 * V + v, V);
        size_t o = 0;
        for (size_t t = 0; t < Tin; ++t)
        {
          DSPVector x;
          load(x, in + v * Sin + t * kFloatsPerDSPVector);
          if (octaves == 0) { store(x, out + v * Sout + o); o += kFloatsPerDSPVector; continue; }
          if (d.write(x))
          {
            store(d.read(), out + v * Sout + o);
            o += kFloatsPerDSPVector;
          }
        }
        for (int h = 0; h < octaves; ++h) hbStore(d._filters[h], state + (size_t)h * 9 * V + v, V);
      }
    }
    return MLGPU_OK;
  }

  // ---- row plumbing and routing: the reference's own templates at fixed sizes ----
  // in[k] are DSPVectorArrays (row-major, as many rows as the case needs); out receives the result rows
  // (for demultiplex cases: the outputs one after the other). Returns the number of output rows, < 0 if unknown.
  int mlref_rows_case(const char* name, const float* const* in, float* out)
  {
    const std::string n(name);
    auto ld = [&](auto& x, const float* p) { std::memcpy(x.getBuffer(), p, sizeof(float) * kFloatsPerDSPVector * (sizeof(x) / sizeof(DSPVector))); };
    auto st = [&](const auto& y) {
      const int rows = (int)(sizeof(y) / sizeof(DSPVector));
      std::memcpy(out, y.getConstBuffer(), sizeof(float) * kFloatsPerDSPVector * rows);
      return rows;
    };
    DSPVectorArray<1> a1; DSPVectorArray<2> a2; DSPVectorArray<3> a3; DSPVectorArray<4> a4; DSPVectorArray<5> a5; DSPVectorArray<6> a6;
    if (n == "repeatRows<3>(2)") { ld(a2, in[0]); return st(repeatRows<3>(a2)); }
    if (n == "stretchRows<7>(3)") { ld(a3, in[0]); return st(stretchRows<7>(a3)); }
    if (n == "stretchRows<4>(6)") { ld(a6, in[0]); return st(stretchRows<4>(a6)); }
    if (n == "zeroPadRows<5>(3)") { ld(a3, in[0]); return st(zeroPadRows<5>(a3)); }
    if (n == "zeroPadRows<2>(3)") { ld(a3, in[0]); return st(zeroPadRows<2>(a3)); }
    if (n == "shiftRows<5>(+2)") { ld(a5, in[0]); return st(shiftRows(a5, 2)); }
    if (n == "shiftRows<5>(-1)") { ld(a5, in[0]); return st(shiftRows(a5, -1)); }
    if (n == "rotateRows<5>(+2)") { ld(a5, in[0]); return st(rotateRows(a5, 2)); }
    if (n == "rotateRows<5>(-7)") { ld(a5, in[0]); return st(rotateRows(a5, -7)); }
    if (n == "concatRows(2,3)") { ld(a2, in[0]); ld(a3, in[1]); return st(concatRows(a2, a3)); }
    if (n == "concatRows(1,2,3)") { ld(a1, in[0]); ld(a2, in[1]); ld(a3, in[2]); return st(concatRows(a1, a2, a3)); }
    if (n == "concatRows(1,2,3,1)") { ld(a1, in[0]); ld(a2, in[1]); ld(a3, in[2]); DSPVectorArray<1> b1; ld(b1, in[3]); return st(concatRows(a1, a2, a3, b1)); }
    if (n == "rotateLeft<3>") { ld(a3, in[0]); return st(rotateLeft(a3)); }
    if (n == "rotateRight<3>") { ld(a3, in[0]); return st(rotateRight(a3)); }
    if (n == "shuffleRows(2,4)") { ld(a2, in[0]); ld(a4, in[1]); return st(shuffleRows(a2, a4)); }
    if (n == "shuffleRows(4,1)") { ld(a4, in[0]); ld(a1, in[1]); return st(shuffleRows(a4, a1)); }
    if (n == "evenRows<5>") { ld(a5, in[0]); return st(evenRows(a5)); }
    if (n == "oddRows<5>") { ld(a5, in[0]); return st(oddRows(a5)); }
    if (n == "separateRows<1,4>(6)") { ld(a6, in[0]); return st(separateRows<1, 4>(a6)); }
    if (n == "addRows<5>") { ld(a5, in[0]); return st(addRows(a5)); }
    if (n == "rowIndex<4>") { return st(rowIndex<4>()); }
    if (n == "columnIndex<3>") { return st(columnIndex<3>()); }
    if (n == "normalize<3>") { ld(a3, in[0]); return st(normalize(a3)); }
    // routing: in[0] = selector (1 row); signals have 2 rows
    DSPVector sel;
    DSPVectorArray<2> x0, x1, x2;
    if (n == "multiplex(3)x2") { ld(sel, in[0]); ld(x0, in[1]); ld(x1, in[2]); ld(x2, in[3]); return st(multiplex(sel, x0, x1, x2)); }
    if (n == "multiplexLinear(3)x2") { ld(sel, in[0]); ld(x0, in[1]); ld(x1, in[2]); ld(x2, in[3]); return st(multiplexLinear(sel, x0, x1, x2)); }
    if (n == "demultiplex(3)x2" || n == "demultiplexLinear(3)x2")
    {
      ld(sel, in[0]); ld(x0, in[1]);
      DSPVectorArray<2> o0, o1, o2;
      if (n == "demultiplex(3)x2") demultiplex(sel, x0, &o0, &o1, &o2); else demultiplexLinear(sel, x0, &o0, &o1, &o2);
      return st(concatRows(o0, o1, o2));
    }
    if (n == "mix(3)x2")  // gains: 3 rows; three 2-row inputs
    {
      ld(a3, in[0]); ld(x0, in[1]); ld(x1, in[2]); ld(x2, in[3]);
      return st(mix(a3, x0, x1, x2));
    }
    return -1;
  }

  // rangeClosed / rangeOpen helpers for the anchor tests
  void mlref_range_closed(float a, float b, float* out64) { store(rangeClosed(a, b), out64); }
  void mlref_range_open(float a, float b, float* out64) { store(rangeOpen(a, b), out64); }

  // ---- makeWindow / dspwindows (MLDSPUtils.h:22-47) ----
  void mlref_make_window(float* dest, size_t size, int shape)
  {
    const Projection shapes[6] = {dspwindows::rectangle, dspwindows::triangle, dspwindows::raisedCosine, dspwindows::hamming, dspwindows::blackman, dspwindows::flatTop};
    makeWindow(dest, size, shapes[shape]);
  }
  // ---- DSPBuffer (MLDSPBuffer.h) for pinning the host ring restatement ----
  void* mlref_dspbuffer_create(int size)
  {
    auto* b = new DSPBuffer();
    b->resize(size);
    return b;
  }
  void mlref_dspbuffer_destroy(void* p) { delete (DSPBuffer*)p; }
  size_t mlref_dspbuffer_read_available(void* p) { return ((DSPBuffer*)p)->getReadAvailable(); }
  size_t mlref_dspbuffer_write_available(void* p) { return ((DSPBuffer*)p)->getWriteAvailable(); }
  void mlref_dspbuffer_write(void* p, const float* src, size_t n) { ((DSPBuffer*)p)->write(src, n); }
  size_t mlref_dspbuffer_read(void* p, float* dst, size_t n) { return ((DSPBuffer*)p)->read(dst, n); }
  void mlref_dspbuffer_discard(void* p, size_t n) { ((DSPBuffer*)p)->discard(n); }
  void mlref_dspbuffer_clear(void* p) { ((DSPBuffer*)p)->clear(); }
  void mlref_dspbuffer_write_overlap_add(void* p, const float* src, size_t n, size_t overlap)
  {
    ((DSPBuffer*)p)->writeWithOverlapAdd(src, n, overlap);
  }
  void mlref_dspbuffer_read_overlap(void* p, float* dst, size_t n, size_t overlap)
  {
    ((DSPBuffer*)p)->readWithOverlap(dst, n, overlap);
  }
  void mlref_dspbuffer_peek_most_recent(void* p, float* dst, size_t n)
  {
    ((DSPBuffer*)p)->peekMostRecent(dst, n);
  }
}
