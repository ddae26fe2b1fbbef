// USER CODE written against madronalib's public API only (namespace ml: DSPVector, generators, filters,
// AudioContext, the SignalProcessFn signature of source/app/MLSignalProcessBuffer.h:18). It includes nothing and is
// compiled twice, unchanged:
//   oracle/dropin_ref.cpp   against the reference's own headers (CPU, one voice per state object)
//   tests/cpp/dropin_gpu.cpp against include/mlgpu/compat (MI355X, captured once, V voices per launch)
// tests/test_gpu_dropin.py compares the two bit for bit.
struct PatchState
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
  LoShelf shelf;
  Bell bell;
};

inline void patchSetup(PatchState& s)
{
  s.lfo.clear();
  s.saw.clear();
  s.pulse.clear();
  s.hp.coeffs = Hipass::makeCoeffs(0.002f, 1.f);
  s.smooth.coeffs = OnePole::makeCoeffs(0.25f);
  s.dc.coeffs = DCBlocker::makeCoeffs(0.045f);
  s.env.coeffs = ADSR::calcCoeffs(0.005f, 0.05f, 0.6f, 0.1f, 48000.f);
  s.shelf.coeffs = LoShelf::makeCoeffs({0.01f, 0.8f, 1.6f});
  s.bell.coeffs = Bell::makeCoeffs(0.1f, 0.5f, 1.5f);
}

// inputs: [0] gate, [1] pitch in octaves above 110 Hz.   outputs: [0] voice, [1] aux
inline void patchProcess(AudioContext* ctx, void* stateData)
{
  auto s = static_cast<PatchState*>(stateData);
  const DSPVector gate = ctx->inputs[0];
  const DSPVector pitch = ctx->inputs[1];
  const DSPVector freq = exp2Approx(pitch) * (110.f / 48000.f);
  const DSPVector lfo = s->lfo(2.f / 48000.f);
  DSPVector osc = s->saw(freq) + s->pulse(freq, 0.5f + lfo * 0.3f) * 0.5f;
  osc += s->noise() * 0.05f;
  // Lopass with per-sample cutoff and resonance (coefficients made per sample, MLDSPFilters.h:136)
  const DSPVector cutoff = clamp(freq * 8.f + lfo * 0.01f, DSPVector(0.001f), DSPVector(0.45f));
  DSPVector y = s->lp(osc, cutoff, DSPVector(0.6f));
  y = s->dc(s->smooth(s->hp(y)));
  y = s->bell(s->shelf(y));
  const DSPVector env = s->env(gate);
  const DSPVector vca = y * env;
  ctx->outputs[0] = clamp(vca, DSPVector(-1.f), DSPVector(1.f));
  ctx->outputs[1] = select(vca, DSPVector(0.f), greaterThan(env, DSPVector(0.01f))) * 0.5f + lerp(osc, y, 0.25f) * 0.1f;
}
