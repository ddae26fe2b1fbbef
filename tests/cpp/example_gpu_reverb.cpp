// The reference's own example program examples/audio-and-midi/reverb.cpp (the Aaltoverb algorithm), included UNCHANGED from
// the reference checkout and compiled against the MI355X shim (include/mlgpu/compat): its process function is captured once
// and run for V independent reverbs per launch. Built only where the reference checkout exists; the library travels.
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <memory>

#define main mlgpu_example_reverb_main
#include "examples/audio-and-midi/reverb.cpp"
#undef main

extern "C" int example_reverb_gpu_run(size_t V, size_t T, int launches, const float* in0, const float* in1, float* out0, float* out1, char* err, size_t errLen)
{
  try
  {
    gpu::Engine eng(0);
    AaltoverbState r;
    initializeReverb(r);
    AudioContext ctx(2, 2, kSampleRate);
    gpu::VoiceProgram prog(eng, V, &ctx, processVector, &r);
    const size_t Tl = T / (size_t)launches;
    gpu::DeviceSignal vm0(eng, V, T, MLGPU_LAYOUT_VOICE_MAJOR), vm1(eng, V, T, MLGPU_LAYOUT_VOICE_MAJOR);
    gpu::DeviceSignal q0(eng, V, T), q1(eng, V, T), o0(eng, V, T), o1(eng, V, T);
    eng.check(mlgpu_upload(eng.handle(), vm0.data(), in0, vm0.bytes()));
    eng.check(mlgpu_upload(eng.handle(), vm1.data(), in1, vm1.bytes()));
    eng.check(mlgpu_layout_convert(eng.handle(), vm0.data(), MLGPU_LAYOUT_VOICE_MAJOR, q0.data(), MLGPU_LAYOUT_QUAD, V, T));
    eng.check(mlgpu_layout_convert(eng.handle(), vm1.data(), MLGPU_LAYOUT_VOICE_MAJOR, q1.data(), MLGPU_LAYOUT_QUAD, V, T));
    for (int l = 0; l < launches; ++l)
    {
      const size_t off = (size_t)l * Tl * 64 * V;
      const float* ins[2] = {q0.data() + off, q1.data() + off};
      float* outs[2] = {o0.data() + off, o1.data() + off};
      eng.check(mlgpu_graph_process(prog.graph(), Tl, ins, MLGPU_LAYOUT_QUAD, outs, MLGPU_LAYOUT_QUAD));
    }
    eng.check(mlgpu_layout_convert(eng.handle(), o0.data(), MLGPU_LAYOUT_QUAD, vm0.data(), MLGPU_LAYOUT_VOICE_MAJOR, V, T));
    eng.check(mlgpu_layout_convert(eng.handle(), o1.data(), MLGPU_LAYOUT_QUAD, vm1.data(), MLGPU_LAYOUT_VOICE_MAJOR, V, T));
    eng.check(mlgpu_download(eng.handle(), out0, vm0.data(), vm0.bytes()));
    eng.check(mlgpu_download(eng.handle(), out1, vm1.data(), vm1.bytes()));
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

// The captured program on an engine the caller owns, kept open: bench.py --workload reverb launches its graph itself
// (mlgpu_graph_process on example_reverb_gpu_graph) and times it with the engine's events like every other workload.
struct ReverbProgram
{
  gpu::Engine eng;
  AaltoverbState r;
  AudioContext ctx;
  std::unique_ptr<gpu::VoiceProgram> prog;
  ReverbProgram(mlgpu_engine* e) : eng(e, gpu::Engine::Borrowed{}), ctx(2, 2, kSampleRate) {}
};
extern "C" void* example_reverb_gpu_open(void* engine, size_t V, int options /* as example_reverb_gpu_bench; bits 4-5: mlgpu_graph_set_delay_layout */, char* err, size_t errLen)
{
  try
  {
    std::unique_ptr<ReverbProgram> p(new ReverbProgram((mlgpu_engine*)engine));
    initializeReverb(p->r);
    gpu::VoiceProgramOptions opt;
    opt.delayWindows = (options & 1) != 0;
    opt.autotune = (options & 2) != 0;
    opt.liveConstants = (options & 4) != 0;
    p->prog.reset(new gpu::VoiceProgram(p->eng, V, &p->ctx, processVector, &p->r, opt));
    return p.release();
  }
  catch (const std::exception& e)
  {
    if (err && errLen) snprintf(err, errLen, "%s", e.what());
    return nullptr;
  }
}
extern "C" void* example_reverb_gpu_graph(void* program) { return program ? ((ReverbProgram*)program)->prog->graph() : nullptr; }
extern "C" void example_reverb_gpu_close(void* program) { delete (ReverbProgram*)program; }

// throughput of the same captured program: V reverbs x T vectors per launch, `launches` launches timed with the engine's events
extern "C" int example_reverb_gpu_bench(size_t V, size_t T, int launches, int options /* bit 0: windowed rings, bit 1: online tuning, bit 2: live constants */, float* msPerLaunch,
                                        char* err, size_t errLen)
{
  try
  {
    gpu::Engine eng(0);
    AaltoverbState r;
    initializeReverb(r);
    AudioContext ctx(2, 2, kSampleRate);
    gpu::VoiceProgramOptions opt;
    opt.delayWindows = (options & 1) != 0;
    opt.autotune = (options & 2) != 0;
    opt.liveConstants = (options & 4) != 0;
    gpu::VoiceProgram prog(eng, V, &ctx, processVector, &r, opt);
    gpu::DeviceSignal q0(eng, V, T), q1(eng, V, T), o0(eng, V, T), o1(eng, V, T);
    eng.check(mlgpu_fill32(eng.handle(), (uint32_t*)q0.data(), 0x3c23d70au /* 0.01f */, V * T * 64));
    eng.check(mlgpu_fill32(eng.handle(), (uint32_t*)q1.data(), 0xbc23d70au, V * T * 64));
    const float* ins[2] = {q0.data(), q1.data()};
    float* outs[2] = {o0.data(), o1.data()};
    for (int l = 0; l < 8; ++l) eng.check(mlgpu_graph_process(prog.graph(), T, ins, MLGPU_LAYOUT_QUAD, outs, MLGPU_LAYOUT_QUAD));  // warm-up and tuning
    eng.check(mlgpu_engine_sync(eng.handle()));
    eng.check(mlgpu_timer_start(eng.handle()));
    for (int l = 0; l < launches; ++l) eng.check(mlgpu_graph_process(prog.graph(), T, ins, MLGPU_LAYOUT_QUAD, outs, MLGPU_LAYOUT_QUAD));
    float ms = 0.f;
    eng.check(mlgpu_timer_stop_ms(eng.handle(), &ms));
    *msPerLaunch = ms / (float)launches;
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
