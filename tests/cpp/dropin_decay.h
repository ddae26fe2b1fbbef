// USER CODE #5 written against madronalib's public API only: a process function that opens with
// `UsingFlushDenormalsToZero f;` exactly as the reference's own examples/audio-and-midi/fdtd.cpp:161 does, around
// recurrences that ring out long after their input has stopped (a resonant Lopass, a slow OnePole, an
// Allpass<IntegerDelay> with its kept DSPVector, a leaky feedback path through a DSPVector member). Compiled unchanged
// against the reference (oracle/dropin_ref.cpp: the scope sets MXCSR FZ | DAZ) and against include/mlgpu/compat
// (dropin_gpu.cpp: the captured program runs in the engine's flush mode). The second entry point is the same body
// without the scope: the default mode, where the tails cross the denormal range instead of being cut.
struct DecayState
{
  Lopass ring;
  OnePole slow;
  DCBlocker dc;
  Allpass<IntegerDelay> smear;
  DSPVector loop;  // kept from one call to the next: one DSPVector of feedback
};

inline void decaySetup(DecayState& s)
{
  s.ring.coeffs = Lopass::makeCoeffs(0.07f, 0.15f);   // lightly damped: rings for thousands of samples
  s.slow.coeffs = OnePole::makeCoeffs(0.004f);
  s.dc.coeffs = DCBlocker::makeCoeffs(0.2f);
  s.smear.mGain = 0.6f;
  s.smear.setMaxDelayInSamples(300.f);
  s.smear.setDelayInSamples(171.f);
}

inline void decayBody(AudioContext* ctx, DecayState* s)
{
  const DSPVector in = ctx->inputs[0];
  const DSPVector a = s->ring(in + s->loop * DSPVector(0.3f));
  const DSPVector b = s->smear(s->slow(a) * DSPVector(0.5f) + a * DSPVector(1e-3f));
  s->loop = s->dc(b);
  ctx->outputs[0] = a * DSPVector(1e-25f);   // a second, much lower copy: reaches the denormal range thousands of samples earlier
  ctx->outputs[1] = b + s->loop * DSPVector(0.25f);
}

inline void decayProcessFlush(AudioContext* ctx, void* stateData)
{
  UsingFlushDenormalsToZero f;
  decayBody(ctx, static_cast<DecayState*>(stateData));
}

inline void decayProcess(AudioContext* ctx, void* stateData) { decayBody(ctx, static_cast<DecayState*>(stateData)); }
