// The reference's examples/audio-and-midi/params.cpp, included unchanged and compiled against the MI355X shim. The example
// keeps its three parameters (freq1, freq2 on a log range, gain) in a SignalProcessor's ParameterTree and reads them in
// the process function every vector; here the tree is madronalib's own (source/app/MLParameters.h on the include path after
// the shim), the process function is captured once, and a parameter change from the host is VoiceProgram::update().
#include <cstddef>
#include <cstdio>

#define main mlgpu_example_params_main
#include "examples/audio-and-midi/params.cpp"
#undef main

// steps[s] = {normalized freq1, normalized freq2, normalized gain} set before segment s of Tseg DSPVectors
extern "C" int example_params_gpu_run(size_t V, size_t nSegments, size_t Tseg, const float* steps, float* out0, float* out1, float* realValues, char* err, size_t errLen)
{
  try
  {
    gpu::Engine eng(0);
    ExampleProcessor proc;
    AudioContext ctx(kInputChannels, kOutputChannels, kSampleRate);
    ParameterDescriptionList pdl;
    readParameterDescriptions(pdl);
    proc.buildParams(pdl);
    proc.setDefaultParams();
    proc.setParamFromNormalizedValue(runtimePath("freq2"), 0.6);   // as the example's main() does
    gpu::VoiceProgramOptions opt;
    opt.liveConstants = true;   // parameters turn into constants of the kernel that update() can change
    gpu::VoiceProgram prog(eng, V, &ctx, processParamsExample, &proc, opt);
    const size_t T = nSegments * Tseg;
    gpu::DeviceSignal o0(eng, V, T, MLGPU_LAYOUT_QUAD), o1(eng, V, T, MLGPU_LAYOUT_QUAD);
    for (size_t s = 0; s < nSegments; ++s)
    {
      if (s > 0)   // segment 0 runs with the example's own settings
      {
        proc.setParamFromNormalizedValue(runtimePath("freq1"), steps[3 * s]);
        proc.setParamFromNormalizedValue(runtimePath("freq2"), steps[3 * s + 1]);
        proc.setParamFromNormalizedValue(runtimePath("gain"), steps[3 * s + 2]);
        prog.update();
      }
      realValues[3 * s] = proc.getRealFloatParam("freq1");
      realValues[3 * s + 1] = proc.getRealFloatParam("freq2");
      realValues[3 * s + 2] = proc.getRealFloatParam("gain");
      const size_t off = s * Tseg * 64 * V;
      float* outs[2] = {o0.data() + off, o1.data() + off};
      eng.check(mlgpu_graph_process(prog.graph(), Tseg, nullptr, MLGPU_LAYOUT_QUAD, outs, MLGPU_LAYOUT_QUAD));
    }
    gpu::DeviceSignal v0(eng, V, T, MLGPU_LAYOUT_VOICE_MAJOR), v1(eng, V, T, MLGPU_LAYOUT_VOICE_MAJOR);
    eng.check(mlgpu_layout_convert(eng.handle(), o0.data(), MLGPU_LAYOUT_QUAD, v0.data(), MLGPU_LAYOUT_VOICE_MAJOR, V, T));
    eng.check(mlgpu_layout_convert(eng.handle(), o1.data(), MLGPU_LAYOUT_QUAD, v1.data(), MLGPU_LAYOUT_VOICE_MAJOR, V, T));
    eng.check(mlgpu_download(eng.handle(), out0, v0.data(), v0.bytes()));
    eng.check(mlgpu_download(eng.handle(), out1, v1.data(), v1.bytes()));
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
