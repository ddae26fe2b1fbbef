// The reference's examples/audio-and-midi/sine.cpp, included unchanged and compiled against the MI355X shim.
#include <cstddef>
#include <cstdio>

#define main mlgpu_example_sine_main
#include "examples/audio-and-midi/sine.cpp"
#undef main

extern "C" int example_sine_gpu_run(size_t V, size_t T, float* out0, float* out1, char* err, size_t errLen)
{
  try
  {
    gpu::Engine eng(0);
    SineExampleState state;
    AudioContext ctx(kInputChannels, kOutputChannels, kSampleRate);
    gpu::VoiceProgram prog(eng, V, &ctx, sineProcess, &state);
    gpu::DeviceSignal o0(eng, V, T, MLGPU_LAYOUT_VOICE_MAJOR), o1(eng, V, T, MLGPU_LAYOUT_VOICE_MAJOR);
    prog.process({}, {&o0, &o1});
    eng.check(mlgpu_download(eng.handle(), out0, o0.data(), o0.bytes()));
    eng.check(mlgpu_download(eng.handle(), out1, o1.data(), o1.bytes()));
    return 0;
  }
  catch (const gpu::Error& e)
  {
    if (err && errLen) snprintf(err, errLen, "%s", e.what());
    return e.status ? e.status : -1;
  }
  catch (const std::exception& e)
  {
    if (err && errLen) snprintf(err, errLen, "%s", e.what());
    return -1;
  }
}
