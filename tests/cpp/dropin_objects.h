// USER CODE #6: the generators, filters and delay lines the other drop-in sources do not touch, used the way user code uses
// them - constructed, configured through their public members and setters, called by operator() - in one process function.
// Compiled unchanged against the reference (oracle/dropin_ref.cpp) and against include/mlgpu/compat (tests/cpp/dropin_gpu.cpp);
// tests/test_gpu_dropin.py compares every output bit for bit (Peak and RMS, whose outputs go through the hardware's reciprocal
// square root, at the stated tolerance). What this pins is the shim's classes: their names, setters, coefficient makers and the
// operator() forms.
constexpr int kObjectsOutputs = 8;

struct ObjectsState
{
  TickGen tick;
  ImpulseGen impulse;
  PhasorGen phasor;
  OneShotGen shot;
  TestSineGen testSine;
  TempoLock lock, lockComputed;
  Bandpass bandpass;
  HiShelf hiShelf;
  Integrator integrator;
  Differentiator differentiator;
  Peak peak;
  RMS rms;
  LinearGlide glide;
  Interpolator1 interpolator;
  IntegerDelay integerDelay{64};
  IntegerDelay modulatedDelay;
  FractionalDelay fractionalDelay{300.f};
  PitchbendableDelay bendDelay;
};

inline void objectsSetup(ObjectsState& s)
{
  s.phasor.clear();
  s.testSine.clear();
  s.shot.trigger();
  s.bandpass.coeffs = Bandpass::makeCoeffs(0.03f, 0.6f);
  s.hiShelf.coeffs = HiShelf::makeCoeffs({0.08f, 0.9f, 1.8f});
  s.integrator.mLeak = 0.002f;
  s.peak.coeffs = Peak::makeCoeffs(0.002f);
  s.peak.peakHoldSamples = 300;
  s.rms.coeffs = RMS::makeCoeffs(0.004f);
  s.glide.setGlideTimeInSamples(400.f);
  s.integerDelay.setDelayInSamples(37);
  s.modulatedDelay.setMaxDelayInSamples(256.f);
  s.fractionalDelay.setDelayInSamples(101.25f);
  s.bendDelay.setMaxDelayInSamples(512.f);
}

// inputs: [0] an audio signal, [1] a slow phasor on [0, 1) (a host transport: TempoLock follows a streamed input).   outputs: kObjectsOutputs signals
inline void objectsProcess(AudioContext* ctx, void* stateData)
{
  auto s = static_cast<ObjectsState*>(stateData);
  const DSPVector x = ctx->inputs[0], slow = ctx->inputs[1];
  const DSPVector freq = DSPVector(220.f / 48000.f) + slow * (110.f / 48000.f);
  const DSPVector ph = s->phasor(freq);
  // 0: pulse-like generators
  ctx->outputs[0] = s->tick(freq * 0.25f) + s->impulse(freq) * 0.5f + s->shot(DSPVector(3.f / 48000.f)) * 0.25f;
  // 1: phase generators
  // one TempoLock follows the host's phasor (a streamed input), one a phasor computed right here
  ctx->outputs[1] = ph + s->testSine(freq * 0.5f) * 0.5f + s->lock(slow, 2.f, 1.f / 48000.f) * 0.25f + s->lockComputed(ph, 0.5f, 1.f / 48000.f) * 0.125f;
  // 2: the two second-order sections no other drop-in uses
  ctx->outputs[2] = s->bandpass(x) + s->hiShelf(x) * 0.5f;
  // 3: one-state recurrences
  ctx->outputs[3] = s->integrator(x * 0.01f) + s->differentiator(x) * 0.5f;
  // 4, 5: envelope followers (hardware-approximate outputs)
  ctx->outputs[4] = s->peak(x);
  ctx->outputs[5] = s->rms(x);
  // 6: control-rate ramps
  ctx->outputs[6] = s->glide(0.75f) + s->interpolator(0.5f) * 0.5f;
  // 7: delay lines: a fixed integer delay, a modulated integer delay, a fixed fractional delay, a pitch-bendable one
  const DSPVector delayTime = DSPVector(40.f) + slow * 150.f;
  ctx->outputs[7] = s->integerDelay(x) + s->modulatedDelay(x, delayTime) * 0.5f + s->fractionalDelay(x) * 0.25f + s->bendDelay(x, delayTime * 2.f) * 0.125f;
}
