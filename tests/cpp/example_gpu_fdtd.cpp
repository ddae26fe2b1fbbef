// The reference's examples/audio-and-midi/fdtd.cpp, included unchanged and compiled against the MI355X shim. Its process function is
// not a DSPVector graph: per sample it reads floats out of signals (`freq[i]`, `inputVec[i]`), steps a 16 x 16 finite-difference
// mesh in plain host code and writes `outLVec[i]` - so it runs the way the reference runs it, called once per DSPVector, in the
// shim's immediate mode: the generators and the vector arithmetic around the mesh are launches on the device, the mesh is the
// program's own loop (under the flush-denormals scope the function opens, on the host and on the device alike).
#include <cstddef>
#include <cstdio>

#define main mlgpu_example_fdtd_main
#include "examples/audio-and-midi/fdtd.cpp"
#undef main

extern "C" int example_fdtd_gpu_run(size_t T, float* out0, float* out1, char* err, size_t errLen)
{
  try
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
  catch (const std::exception& e)
  {
    if (err && errLen) snprintf(err, errLen, "%s", e.what());
    return -1;
  }
}
