// USER CODE #8: the routing functions of MLDSPRouting.h and the function wrappers of MLDSPFunctional.h that the other drop-in
// sources do not use, by name: mix, multiplex, multiplexLinear, demultiplex, demultiplexLinear, Bank<T, ROWS> (operator() with row
// arguments, operator[], clear()), map over rows (three forms). Compiled unchanged against the reference (oracle/dropin_ref.cpp)
// and against include/mlgpu/compat (tests/cpp/dropin_gpu.cpp).
// Not here: FeedbackDelayFunction / FeedbackDelayFunctionWithTap. The reference's own objects cannot run - their delay lines are
// private, default-constructed (an empty buffer, MLDSPFilters.h:803-809) and have no way to be given a length, so the first call
// writes through a null pointer (MLDSPFunctional.h:262-276). The device forms are checked against the written-out patch instead
// (tests/test_gpu_delays.py).
constexpr int kRoutingOutputs = 8;

struct RoutingState
{
  Bank<SineGen, 3> oscillators;
  Bank<Lopass, 2> filters;
  OnePole smoothers[2];
};

inline void routingSetup(RoutingState& s)
{
  s.oscillators.clear();
  s.filters[0].coeffs = Lopass::makeCoeffs(0.05f, 0.8f);
  s.filters[1].coeffs = Lopass::makeCoeffs(0.12f, 0.6f);
  s.smoothers[0].coeffs = OnePole::makeCoeffs(0.2f);
  s.smoothers[1].coeffs = OnePole::makeCoeffs(0.05f);
}

// inputs: [0] an audio signal, [1] a second one, [2] a selector that wanders through [0, 1) and beyond.   outputs: kRoutingOutputs
inline void routingProcess(AudioContext* ctx, void* stateData)
{
  auto s = static_cast<RoutingState*>(stateData);
  const DSPVector a = ctx->inputs[0], b = ctx->inputs[1], sel = ctx->inputs[2];
  const DSPVector c = a * b;

  // 0: mix(gains, inputs...): each input times its row of the gains
  const DSPVectorArray<3> gains = concatRows(DSPVector(0.5f), sel, DSPVector(-0.25f));
  ctx->outputs[0] = mix(gains, a, b, c);
  // 1, 2: multiplex / multiplexLinear over three candidates
  ctx->outputs[1] = multiplex(sel, a, b, c);
  ctx->outputs[2] = multiplexLinear(sel, a, b, c);
  // 3: demultiplex into three, recombined with different weights
  DSPVector d0, d1, d2;
  demultiplex(sel, a, &d0, &d1, &d2);
  ctx->outputs[3] = d0 + d1 * 2.f + d2 * 4.f;
  // 4: demultiplexLinear into two
  DSPVector l0, l1;
  demultiplexLinear(sel, b, &l0, &l1);
  ctx->outputs[4] = l0 - l1;
  // 5: two banks: three sine oscillators with a frequency per row, summed by rows; two filters on two rows
  const DSPVectorArray<3> freqs = concatRows(DSPVector(110.f / 48000.f), DSPVector(220.f / 48000.f) + sel * (5.f / 48000.f), DSPVector(331.f / 48000.f));
  const DSPVectorArray<3> tones = s->oscillators(freqs);
  const DSPVectorArray<2> filtered = s->filters(concatRows(a, b));
  ctx->outputs[5] = tones.constRow(0) + tones.constRow(1) + tones.constRow(2) + filtered.constRow(0) - filtered.constRow(1);
  // 6: map() with a function of a row, and of a row and its index (a stateful object per row)
  const DSPVectorArray<2> ab = concatRows(a, b);
  const DSPVectorArray<2> squared = map([](const DSPVector v) { return v * v; }, ab);
  const DSPVectorArray<2> smoothed = map([&](const DSPVector v, int row) { return s->smoothers[row](v) * DSPVector(row ? -0.5f : 1.5f); }, ab);
  ctx->outputs[6] = squared.constRow(0) - squared.constRow(1) + smoothed.constRow(0) + smoothed.constRow(1);
  // 7: routing functions on arrays of rows: two rows switched and mixed at once
  const DSPVectorArray<2> ba = concatRows(b, a);
  const DSPVectorArray<2> switched = multiplex(sel, ab, ba);
  const DSPVectorArray<2> blended = mix(concatRows(sel, DSPVector(1.f) - sel), ab, ba);
  ctx->outputs[7] = switched.constRow(0) - switched.constRow(1) * 0.5f + blended.constRow(0) + blended.constRow(1) * 0.25f;
}
