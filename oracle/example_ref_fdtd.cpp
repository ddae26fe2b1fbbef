// oracle/example_ref_fdtd.cpp — TEST INFRASTRUCTURE. The reference's examples/audio-and-midi/fdtd.cpp, included unchanged
// (see example_ref_reverb.cpp, which also holds the AudioTask stubs).
#include <cstddef>

#define main mlref_example_fdtd_main
#include "examples/audio-and-midi/fdtd.cpp"
#undef main

extern "C" int example_fdtd_ref_run(size_t T, float* out0, float* out1)
{
  FDTDState state;
  AudioContext ctx(kInputChannels, kOutputChannels, kSampleRate);
  for (size_t t = 0; t < T; ++t)
  {
    processFDTD(&ctx, &state);
    store(ctx.outputs[0], out0 + t * kFloatsPerDSPVector);
    store(ctx.outputs[1], out1 + t * kFloatsPerDSPVector);
  }
  return 0;
}
