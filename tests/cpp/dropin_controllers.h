// USER CODE #9: the process function of the reference's examples/audio-and-midi/controllers-to-audio.cpp - eight sine oscillators
// whose frequencies follow eight MIDI controllers, a ninth controller for the volume - with its one host-side step written on
// whole vectors: the example reads ctrlSig[0] into a float and maps it through a std::function projection
// (projections::unityToLogParam({110, 440}): 110 * 4^x) once per DSPVector; here the same mapping is applied to the smoothed
// controller signal itself, 110 * 2^(2x) with exp2Approx. Everything else is the example's own text. Compiled unchanged against the
// reference (oracle/dropin_ref.cpp) and against include/mlgpu/compat (tests/cpp/dropin_gpu.cpp: one such program per "voice",
// each with its own controllers).
constexpr float kCtlAudioOutputGain = 0.5f;

struct CtlAudioState
{
  std::vector<int> sineControllers{19, 23, 27, 31, 49, 53, 57, 61};
  const int volumeControl{62};
  std::vector<SineGen> sineGens;
};

inline void ctlAudioProcess(AudioContext* ctx, void* untypedState)
{
  CtlAudioState* state = reinterpret_cast<CtlAudioState*>(untypedState);

  float sr = ctx->getSampleRate();
  DSPVector accum;

  // accumulate sine oscillators
  auto nSines = state->sineControllers.size();
  for (int i = 0; i < (int)nSines; ++i)
  {
    int ctrlNum = state->sineControllers[i];
    DSPVector ctrlSig = ctx->getInputController(ctrlNum);
    DSPVector freqInHz = exp2Approx(ctrlSig * 2.f) * 110.f;
    DSPVector sineSig = state->sineGens[i](freqInHz / sr);
    accum += sineSig;
  }

  // scale total volume and write context output
  DSPVector volumeSig = ctx->getInputController(state->volumeControl);
  accum *= volumeSig * kCtlAudioOutputGain / nSines;
  ctx->outputs[0] = accum;
  ctx->outputs[1] = accum;
}
