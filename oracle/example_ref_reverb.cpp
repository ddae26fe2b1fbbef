// oracle/example_ref_reverb.cpp — TEST INFRASTRUCTURE. The reference's own example program
// examples/audio-and-midi/reverb.cpp (the Aaltoverb algorithm), included UNCHANGED from /root/reference and compiled against
// the reference's headers; its process function is run the way AudioTask runs it (one call per 64 frames, one state per
// instance). main() is renamed out of the way; AudioTask itself needs RtAudio and is stubbed (never run).
#include <cstddef>
#include <cstring>
#include <memory>

#define main mlref_example_reverb_main
#include "examples/audio-and-midi/reverb.cpp"
#undef main

namespace ml
{
struct AudioTask::Impl
{
};
AudioTask::AudioTask(AudioContext*, SignalProcessFn, void*) {}
AudioTask::~AudioTask() {}
int AudioTask::startAudio() { return -1; }
void AudioTask::stopAudio() {}
int AudioTask::runConsoleApp() { return -1; }
}  // namespace ml

extern "C" int example_reverb_ref_run(size_t V, size_t T, const float* in0, const float* in1, float* out0, float* out1)
{
  const size_t S = T * kFloatsPerDSPVector;
  for (size_t v = 0; v < V; ++v)
  {
    AaltoverbState r;
    initializeReverb(r);
    AudioContext ctx(2, 2, kSampleRate);  // the example sums inputs[0] + inputs[1]
    for (size_t t = 0; t < T; ++t)
    {
      load(ctx.inputs[0], in0 + v * S + t * kFloatsPerDSPVector);
      load(ctx.inputs[1], in1 + v * S + t * kFloatsPerDSPVector);
      processVector(&ctx, &r);
      store(ctx.outputs[0], out0 + v * S + t * kFloatsPerDSPVector);
      store(ctx.outputs[1], out1 + v * S + t * kFloatsPerDSPVector);
    }
  }
  return 0;
}
