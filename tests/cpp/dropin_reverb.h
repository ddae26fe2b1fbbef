// USER CODE #2 written against madronalib's public API only: a small plate-style reverb in the manner of the
// reference's examples/audio-and-midi/reverb.cpp (smoothed float parameters, a chain of Allpass<PitchbendableDelay>
// diffusers, a stereo tank whose feedback is kept in two DSPVector members of the state struct between calls).
// Compiled unchanged against the reference (oracle/dropin_ref.cpp) and against include/mlgpu/compat (dropin_gpu.cpp).
struct PlateState
{
  LinearGlide smoothFeedback;
  LinearGlide smoothSize;
  FractionalDelay predelay;
  Allpass<IntegerDelay> fixedDiffuser;
  Allpass<PitchbendableDelay> d1, d2, d3, tankL1, tankR1, tankL2, tankR2;
  PitchbendableDelay lineL, lineR;
  OnePole dampL, dampR;
  DSPVector feedbackL, feedbackR;  // written at the end of one call, read at the start of the next
  float size{0.8f}, feedback{0.55f}, damping{0.2f};  // the knobs: plain floats the host changes whenever it likes
};

inline void plateTurnKnobs(PlateState& p)
{
  p.size = 0.55f;
  p.feedback = 0.7f;
  p.damping = 0.1f;
}

inline void plateSetup(PlateState& p)
{
  p.smoothFeedback.setGlideTimeInSamples(0.02f * 48000);
  p.smoothSize.setGlideTimeInSamples(0.02f * 48000);
  p.predelay.setMaxDelayInSamples(600.f);
  p.predelay.setDelayInSamples(331.37f);
  p.fixedDiffuser.mGain = 0.6f;
  p.fixedDiffuser.setMaxDelayInSamples(300.f);
  p.fixedDiffuser.setDelayInSamples(211.f);
  p.d1.mGain = 0.75f;
  p.d2.mGain = 0.7f;
  p.d3.mGain = 0.625f;
  p.tankL1.mGain = p.tankR1.mGain = 0.7f;
  p.tankL2.mGain = p.tankR2.mGain = 0.5f;
  p.d1.setMaxDelayInSamples(600.f);
  p.d2.setMaxDelayInSamples(600.f);
  p.d3.setMaxDelayInSamples(1200.f);
  p.tankL1.setMaxDelayInSamples(3000.f);
  p.tankR1.setMaxDelayInSamples(3000.f);
  p.tankL2.setMaxDelayInSamples(9000.f);
  p.tankR2.setMaxDelayInSamples(9000.f);
  p.lineL.setMaxDelayInSamples(4000.f);
  p.lineR.setMaxDelayInSamples(4000.f);
  p.dampL.coeffs = OnePole::makeCoeffs(0.2f);
  p.dampR.coeffs = OnePole::makeCoeffs(0.17f);
}

// inputs: [0] left, [1] right.  outputs: [0] left, [1] right
inline void plateProcess(AudioContext* ctx, void* stateData)
{
  PlateState* p = static_cast<PlateState*>(stateData);
  const float sr = 48000.f;

  // control-rate parameters arrive as floats and are smoothed to signals
  DSPVector vSize = p->smoothSize(p->size);
  DSPVector vFeedback = p->smoothFeedback(p->feedback);
  p->dampL.coeffs = OnePole::makeCoeffs(p->damping);
  DSPVector vMin(kFloatsPerDSPVector);
  DSPVector sizeInSamples = sr * vSize;
  DSPVector t1 = max(0.0047 * sizeInSamples, vMin);
  DSPVector t2 = max(0.0036 * sizeInSamples, vMin);
  DSPVector t3 = max(0.0127 * sizeInSamples, vMin);
  DSPVector t4 = max(0.031 * sizeInSamples, vMin);
  DSPVector t5 = max(0.027 * sizeInSamples, vMin);
  DSPVector t6 = max(0.093 * sizeInSamples, vMin);
  DSPVector t7 = max(0.081 * sizeInSamples, vMin);

  DSPVector mono = (ctx->inputs[0] + ctx->inputs[1]) * 0.5f;
  DSPVector diffused = p->d3(p->d2(p->d1(p->fixedDiffuser(p->predelay(mono)), t1), t2), t3);

  DSPVector lineTimeL = max(0.0413 * sizeInSamples - vMin, DSPVector(0.f));
  DSPVector lineTimeR = max(0.0471 * sizeInSamples - vMin, DSPVector(0.f));
  DSPVector tapL = p->tankL2(p->tankL1(diffused + p->lineL(p->feedbackL, lineTimeL), t4), t6);
  DSPVector tapR = p->tankR2(p->tankR1(diffused + p->lineR(p->feedbackR, lineTimeR), t5), t7);

  // cross-coupled, damped feedback for the next call
  p->feedbackR = p->dampL(tapL) * vFeedback;
  p->feedbackL = p->dampR(tapR) * vFeedback;

  ctx->outputs[0] = tapL + ctx->inputs[0] * 0.25f;
  ctx->outputs[1] = tapR + ctx->inputs[1] * 0.25f;
}
