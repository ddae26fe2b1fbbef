// oracle/example_ref_params.cpp — TEST INFRASTRUCTURE. The reference's examples/audio-and-midi/params.cpp, included
// unchanged (see example_ref_reverb.cpp, which also holds the AudioTask stubs), driven with the same parameter steps as
// tests/cpp/example_gpu_params.cpp.
#include <cstddef>

#define main mlref_example_params_main
#include "examples/audio-and-midi/params.cpp"
#undef main

extern "C" int example_params_ref_run(size_t nSegments, size_t Tseg, const float* steps, float* out0, float* out1, float* realValues)
{
  ExampleProcessor proc;
  AudioContext ctx(kInputChannels, kOutputChannels, kSampleRate);
  ParameterDescriptionList pdl;
  readParameterDescriptions(pdl);
  proc.buildParams(pdl);
  proc.setDefaultParams();
  proc.setParamFromNormalizedValue(runtimePath("freq2"), 0.6);
  for (size_t s = 0; s < nSegments; ++s)
  {
    if (s > 0)
    {
      proc.setParamFromNormalizedValue(runtimePath("freq1"), steps[3 * s]);
      proc.setParamFromNormalizedValue(runtimePath("freq2"), steps[3 * s + 1]);
      proc.setParamFromNormalizedValue(runtimePath("gain"), steps[3 * s + 2]);
    }
    realValues[3 * s] = proc.getRealFloatParam("freq1");
    realValues[3 * s + 1] = proc.getRealFloatParam("freq2");
    realValues[3 * s + 2] = proc.getRealFloatParam("gain");
    for (size_t t = 0; t < Tseg; ++t)
    {
      processParamsExample(&ctx, &proc);
      store(ctx.outputs[0], out0 + (s * Tseg + t) * kFloatsPerDSPVector);
      store(ctx.outputs[1], out1 + (s * Tseg + t) * kFloatsPerDSPVector);
    }
  }
  return 0;
}
