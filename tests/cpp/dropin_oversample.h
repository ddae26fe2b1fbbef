// USER CODE #4 written against madronalib's public API only: an oversampled waveshaper (Upsample2xFunction around a
// stateful function) and a half-rate branch (Downsample2xFunction with two input rows), as MLDSPFunctional.h:104-213
// documents them. Compiled unchanged against the reference (oracle/dropin_ref.cpp) and against include/mlgpu/compat
// (dropin_gpu.cpp).
// host tables: a fade-in window made by a function of the index, and a comb of gains read from memory
inline float fadeWindow(int n) { return (n < 16) ? (float)n * (1.f / 16.f) : 1.f - (float)(n - 16) * (0.25f / 48.f); }
struct OversampleState
{
  float combGains[64];
  DSPVector comb;    // filled by load() in the setup function, outside any capture
  Upsample2xFunction<1> upper;
  Upsample2xFunction<1> quadOuter, quadInner;  // one inside the other: a 4x oversampled hard clipper
  Downsample2xFunction<2> downer;
  Lopass preFilter;  // lives inside the 2x function: sees 128 samples per DSPVector of input
  Allpass<IntegerDelay> smear;  // so does this one: its ring and its kept DSPVector (vy1) run at 2x
  SineGen carrier;   // lives inside the half-rate function: advances 32 samples per DSPVector of input
  OnePole smooth;
  DCBlocker dc;
  PhasorGen phasor;  // drives the free functions phasorToSine / phasorToSaw / phasorToPulse
  TempoLock lock;    // follows the second input as if it were a clock phasor, at twice its rate
};

inline void oversampleSetup(OversampleState& s)
{
  s.preFilter.coeffs = Lopass::makeCoeffs(0.15f, 0.9f);
  s.smear.mGain = 0.5f;
  s.smear.setMaxDelayInSamples(200.f);
  s.smear.setDelayInSamples(113.f);
  s.smooth.coeffs = OnePole::makeCoeffs(0.1f);
  s.dc.coeffs = DCBlocker::makeCoeffs(0.002f);
  for (int n = 0; n < 64; ++n) s.combGains[n] = (n % 5 == 0) ? 1.25f : 0.75f + 0.001f * (float)n;
  load(s.comb, s.combGains);
}

// inputs: [0] audio, [1] modulation.  outputs: [0] shaped, [1] mix
inline void oversampleProcess(AudioContext* ctx, void* stateData)
{
  OversampleState* s = static_cast<OversampleState*>(stateData);
  const DSPVector in = ctx->inputs[0];
  const DSPVector mod = ctx->inputs[1];

  DSPVector shaped = s->upper(
      [&](const DSPVector x)
      {
        // a table inside the 2x function: applied to each of its two DSPVectors per outer vector
        DSPVector driven = s->smear(s->preFilter(x * DSPVector(4.0f) * DSPVector(s->combGains)));
        return clamp(driven - driven * driven * driven * DSPVector(0.333f), DSPVector(-1.f), DSPVector(1.f));
      },
      in);

  // map (MLDSPFunctional.h:76-87): a function of (row, row index) applied to each row
  const DSPVectorArray<2> scaled = map([](const DSPVector v, int row) { return v * DSPVector(row ? 0.5f : 2.f); }, concatRows(in, mod));

  DSPVector lofi = s->downer(
      [&](const DSPVectorArray<2> v) { return s->smooth(v.constRow(0) * s->carrier(DSPVector(0.01f)) + v.constRow(1) * DSPVector(0.1f)); },
      scaled);

  DSPVector clipped4x = s->quadOuter(
      [&](const DSPVector x)
      { return s->quadInner([&](const DSPVector y) { return clamp(y * DSPVector(2.5f), DSPVector(-1.f), DSPVector(1.f)); }, x); },
      in);

  // the waveshape functions of MLDSPGens.h:313-369 called directly on a phasor, with a moving frequency and pulse width
  const DSPVector f = DSPVector(0.004f) + abs(mod) * DSPVector(0.01f);
  const DSPVector p = s->phasor(f);
  const DSPVector shapes = phasorToSine(p) + phasorToSaw(p, f) + phasorToPulse(p, f, DSPVector(0.3f) + mod * DSPVector(0.2f));

  const DSPVector window(fadeWindow);
  ctx->outputs[0] = s->dc(shaped) * s->comb + shapes * DSPVector(0.1f) * window;
  ctx->outputs[1] = lofi * DSPVector(0.5f) + shaped * DSPVector(0.5f) + clipped4x * DSPVector(0.25f) + s->lock(ctx->inputs[1], 2.0f, 1.0f / 48000.f) * DSPVector(0.1f);
}
