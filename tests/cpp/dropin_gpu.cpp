// The same user code (dropin_patch.h) compiled against the MI355X shim: captured once, run for V voices.
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "madronalib.h"  // include/mlgpu/compat/madronalib.h
using namespace ml;

#include "dropin_patch.h"

// gate / pitch / outputs: VOICE_MAJOR [V][64T] host arrays. Returns 0, or an mlgpu status.
extern "C" int dropin_gpu_run(size_t V, size_t T, const float* gate, const float* pitch, float* out0, float* out1, char* err, size_t errLen)
{
  try
  {
    gpu::Engine eng(0);
    PatchState state;
    patchSetup(state);
    AudioContext ctx(2, 2, 48000);
    gpu::VoiceProgram prog(eng, V, &ctx, patchProcess, &state);
    // per-voice noise seeds are all 0 in the reference run too (one fresh NoiseGen per voice)
    gpu::DeviceSignal dGate(eng, V, T, MLGPU_LAYOUT_VOICE_MAJOR), dPitch(eng, V, T, MLGPU_LAYOUT_VOICE_MAJOR);
    gpu::DeviceSignal dOut0(eng, V, T, MLGPU_LAYOUT_VOICE_MAJOR), dOut1(eng, V, T, MLGPU_LAYOUT_VOICE_MAJOR);
    eng.check(mlgpu_upload(eng.handle(), dGate.data(), gate, dGate.bytes()));
    eng.check(mlgpu_upload(eng.handle(), dPitch.data(), pitch, dPitch.bytes()));
    prog.process({&dGate, &dPitch}, {&dOut0, &dOut1});
    eng.check(mlgpu_download(eng.handle(), out0, dOut0.data(), dOut0.bytes()));
    eng.check(mlgpu_download(eng.handle(), out1, dOut1.data(), dOut1.bytes()));
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

// ... and with VoiceProgramOptions::mixOutputs: the two outputs as the sum of all V voices (one channel each, [64T])
extern "C" int dropin_gpu_run_mixed(size_t V, size_t T, const float* gate, const float* pitch, float* mix0, float* mix1, char* err, size_t errLen)
{
  try
  {
    gpu::Engine eng(0);
    PatchState state;
    patchSetup(state);
    AudioContext ctx(2, 2, 48000);
    gpu::VoiceProgramOptions opt;
    opt.mixOutputs = true;
    gpu::VoiceProgram prog(eng, V, &ctx, patchProcess, &state, opt);
    prog.reserveMixdown(T);
    gpu::DeviceSignal dGate(eng, V, T, MLGPU_LAYOUT_VOICE_MAJOR), dPitch(eng, V, T, MLGPU_LAYOUT_VOICE_MAJOR);
    gpu::DeviceSignal dMix0(eng, 1, T, MLGPU_LAYOUT_VOICE_MAJOR), dMix1(eng, 1, T, MLGPU_LAYOUT_VOICE_MAJOR);
    eng.check(mlgpu_upload(eng.handle(), dGate.data(), gate, dGate.bytes()));
    eng.check(mlgpu_upload(eng.handle(), dPitch.data(), pitch, dPitch.bytes()));
    prog.process({&dGate, &dPitch}, {&dMix0, &dMix1});
    eng.check(mlgpu_download(eng.handle(), mix0, dMix0.data(), dMix0.bytes()));
    eng.check(mlgpu_download(eng.handle(), mix1, dMix1.data(), dMix1.bytes()));
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

#include "dropin_reverb.h"
// the plate reverb for V independent instances; launches: how many process calls the T vectors are split into
// knobsAt: the vector (a launch boundary) before which the host turns the knobs and calls VoiceProgram::update(); >= T: never
extern "C" int plate_gpu_run(size_t V, size_t T, int launches, size_t knobsAt, const float* inL, const float* inR, float* outL, float* outR, char* err,
                             size_t errLen)
{
  try
  {
    gpu::Engine eng(0);
    PlateState state;
    plateSetup(state);
    AudioContext ctx(2, 2, 48000);
    gpu::VoiceProgramOptions opt;
    opt.liveConstants = knobsAt < T;  // the knob values become DSPVector(f) constants of the captured code
    gpu::VoiceProgram prog(eng, V, &ctx, plateProcess, &state, opt);
    const size_t Tl = T / (size_t)launches;
    // QUAD signals: the T vectors of one launch are contiguous, so consecutive launches are consecutive slices
    gpu::DeviceSignal vmL(eng, V, T, MLGPU_LAYOUT_VOICE_MAJOR), vmR(eng, V, T, MLGPU_LAYOUT_VOICE_MAJOR);
    gpu::DeviceSignal qL(eng, V, T), qR(eng, V, T), oL(eng, V, T), oR(eng, V, T);
    eng.check(mlgpu_upload(eng.handle(), vmL.data(), inL, vmL.bytes()));
    eng.check(mlgpu_upload(eng.handle(), vmR.data(), inR, vmR.bytes()));
    eng.check(mlgpu_layout_convert(eng.handle(), vmL.data(), MLGPU_LAYOUT_VOICE_MAJOR, qL.data(), MLGPU_LAYOUT_QUAD, V, T));
    eng.check(mlgpu_layout_convert(eng.handle(), vmR.data(), MLGPU_LAYOUT_VOICE_MAJOR, qR.data(), MLGPU_LAYOUT_QUAD, V, T));
    for (int l = 0; l < launches; ++l)
    {
      const size_t off = (size_t)l * Tl * 64 * V;
      if ((size_t)l * Tl == knobsAt)
      {
        plateTurnKnobs(state);
        prog.update();  // the process function runs once more on the host; no recompilation, no state touched
      }
      const float* ins[2] = {qL.data() + off, qR.data() + off};
      float* outs[2] = {oL.data() + off, oR.data() + off};
      eng.check(mlgpu_graph_process(prog.graph(), Tl, ins, MLGPU_LAYOUT_QUAD, outs, MLGPU_LAYOUT_QUAD));
    }
    eng.check(mlgpu_layout_convert(eng.handle(), oL.data(), MLGPU_LAYOUT_QUAD, vmL.data(), MLGPU_LAYOUT_VOICE_MAJOR, V, T));
    eng.check(mlgpu_layout_convert(eng.handle(), oR.data(), MLGPU_LAYOUT_QUAD, vmR.data(), MLGPU_LAYOUT_VOICE_MAJOR, V, T));
    eng.check(mlgpu_download(eng.handle(), outL, vmL.data(), vmL.bytes()));
    eng.check(mlgpu_download(eng.handle(), outR, vmR.data(), vmR.bytes()));
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

#include "dropin_oversample.h"
// the oversampled shaper for V independent instances, split into `launches` process calls of T / launches vectors
extern "C" int oversample_gpu_run(size_t V, size_t T, int launches, const float* in0, const float* in1, float* out0, float* out1, char* err, size_t errLen)
{
  try
  {
    gpu::Engine eng(0);
    OversampleState state;
    oversampleSetup(state);
    AudioContext ctx(2, 2, 48000);
    gpu::VoiceProgram prog(eng, V, &ctx, oversampleProcess, &state);
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

#include "dropin_decay.h"
// the ringing-out patch for V independent instances in one launch; flush != 0: the process function that opens with
// UsingFlushDenormalsToZero. *usedFlushMode reports what the captured program found out by itself.
extern "C" int decay_gpu_run(size_t V, size_t T, int flush, const float* in0, float* out0, float* out1, int* usedFlushMode, char* err, size_t errLen)
{
  try
  {
    gpu::Engine eng(0);
    DecayState state;
    decaySetup(state);
    AudioContext ctx(1, 2, 48000);
    gpu::VoiceProgram prog(eng, V, &ctx, flush ? decayProcessFlush : decayProcess, &state);
    if (usedFlushMode) *usedFlushMode = prog.flushesDenormals() ? 1 : 0;
    gpu::DeviceSignal vm0(eng, V, T, MLGPU_LAYOUT_VOICE_MAJOR), vm1(eng, V, T, MLGPU_LAYOUT_VOICE_MAJOR);
    gpu::DeviceSignal q0(eng, V, T), o0(eng, V, T), o1(eng, V, T);
    eng.check(mlgpu_upload(eng.handle(), vm0.data(), in0, vm0.bytes()));
    eng.check(mlgpu_layout_convert(eng.handle(), vm0.data(), MLGPU_LAYOUT_VOICE_MAJOR, q0.data(), MLGPU_LAYOUT_QUAD, V, T));
    prog.process({&q0}, {&o0, &o1});
    if (mlgpu_engine_get_flush_denormals(eng.handle())) throw std::logic_error("the engine's mode leaked out of VoiceProgram::process");
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

// ---- host data: scalar map forms, windows through getBuffer(), DSPBuffer overlap-add, v[n] on host vectors ----
#include "dropin_eager.h"
// the same suite through the shim's immediate mode: every call a launch on the device
extern "C" long immediate_gpu_run(float* out, size_t cap, char* names, size_t namesLen, char* err, size_t errLen)
{
  try
  {
    ImmediateLog log;
    immediateSuite(log);
    for (size_t i = 0; i < log.data.size() && i < cap; ++i) out[i] = log.data[i];
    std::string nm;
    for (size_t i = 0; i < log.names.size(); ++i) nm += log.names[i] + "@" + std::to_string(log.starts[i]) + ";";
    if (names && namesLen) snprintf(names, namesLen, "%s", nm.c_str());
    return (long)log.data.size();
  }
  catch (const std::exception& e)
  {
    if (err && errLen) snprintf(err, errLen, "%s", e.what());
    return -1;
  }
}

// ---- the shim's feedback composites the reference cannot run (its FDN and FeedbackDelayFunction have no way to size their delay
// lines): the same process function captured into a kernel and run imperatively must give the same bits -------------------------
namespace loopsTest
{
constexpr int kOutputs = 4;
struct State
{
  FDN<4> fdn;
  FeedbackDelayFunction fbd;
  FeedbackDelayFunctionWithTap fbt;
  OnePole inLoop, inTapLoop;
  Allpass<PitchbendableDelay> ap;
};
inline void setup(State* s)
{
  s->fdn.setDelaysInSamples({{67.f, 73.f, 91.f, 103.f}});
  s->fdn.setFilterCutoffs({{0.1f, 0.2f, 0.3f, 0.4f}});
  s->fdn.mFeedbackGains = {{0.5f, 0.55f, 0.6f, 0.65f}};
  s->fbd.feedbackGain = 0.7f;
  s->fbd.setMaxDelayInSamples(400.f);
  s->fbt.feedbackGain = 0.6f;
  s->fbt.setMaxDelayInSamples(400.f);
  s->inLoop.coeffs = OnePole::makeCoeffs(0.1f);
  s->inTapLoop.coeffs = OnePole::makeCoeffs(0.2f);
  s->ap.setMaxDelayInSamples(500.f);
  s->ap.mGain = 0.5f;
}
inline void process(AudioContext* ctx, void* untyped)
{
  State* s = static_cast<State*>(untyped);
  const DSPVector x = ctx->inputs[0];
  const DSPVectorArray<2> wet = s->fdn(x);
  ctx->outputs[0] = wet.constRow(0);
  ctx->outputs[1] = wet.constRow(1);
  ctx->outputs[2] = s->fbd(x, [&](const DSPVector in) { return s->inLoop(in) * 0.9f; }, DSPVector(150.f) + x * 20.f);
  ctx->outputs[3] = s->fbt(x, [&](const DSPVector in, DSPVector& tap) {
    tap = s->inTapLoop(in);
    return tap * 0.8f;
  }, DSPVector(131.f)) + s->ap(x, DSPVector(200.f) + x * 30.f);
}
}  // namespace loopsTest
extern "C" int loops_captured_and_immediate_run(size_t T, const float* in0, float* captured /* [4][64 T] */, float* immediate, char* err, size_t errLen)
{
  try
  {
    using namespace loopsTest;
    const size_t S = T * 64;
    {
      gpu::Engine eng(0);
      State state;
      setup(&state);
      AudioContext ctx(1, kOutputs, 48000);
      gpu::VoiceProgram prog(eng, 1, &ctx, process, &state);
      gpu::DeviceSignal q0(eng, 1, T);
      eng.check(mlgpu_upload(eng.handle(), q0.data(), in0, q0.bytes()));   // one voice: QUAD is the samples in order
      std::vector<gpu::DeviceSignal> o;
      std::vector<gpu::DeviceSignal*> po;
      o.reserve(kOutputs);
      for (int i = 0; i < kOutputs; ++i) o.emplace_back(eng, 1, T);
      for (auto& x : o) po.push_back(&x);
      prog.process({&q0}, po);
      for (int i = 0; i < kOutputs; ++i) eng.check(mlgpu_download(eng.handle(), captured + (size_t)i * S, o[(size_t)i].data(), S * 4));
    }
    {
      State state;
      setup(&state);
      AudioContext ctx(1, kOutputs, 48000);
      for (size_t t = 0; t < T; ++t)
      {
        load(ctx.inputs[0], in0 + t * 64);
        process(&ctx, &state);
        for (int o = 0; o < kOutputs; ++o) store(ctx.outputs[o], immediate + (size_t)o * S + t * 64);
      }
    }
    return 0;
  }
  catch (const std::exception& e)
  {
    if (err && errLen) snprintf(err, errLen, "%s", e.what());
    return -1;
  }
}

#include "dropin_hostdata.h"
extern "C" int hostdata_gpu_run(size_t V, size_t T, const float* in0, float* outs /* [kHostDataOutputs][V][64 T] */, char* err, size_t errLen)
{
  try
  {
    gpu::Engine eng(0);
    HostDataState state;
    hostDataSetup(&state);
    AudioContext ctx(1, kHostDataOutputs, 48000);
    gpu::VoiceProgram prog(eng, V, &ctx, hostDataProcess, &state);
    gpu::DeviceSignal vm(eng, V, T, MLGPU_LAYOUT_VOICE_MAJOR), q0(eng, V, T);
    eng.check(mlgpu_upload(eng.handle(), vm.data(), in0, vm.bytes()));
    eng.check(mlgpu_layout_convert(eng.handle(), vm.data(), MLGPU_LAYOUT_VOICE_MAJOR, q0.data(), MLGPU_LAYOUT_QUAD, V, T));
    std::vector<gpu::DeviceSignal> o;
    std::vector<gpu::DeviceSignal*> po;
    o.reserve(kHostDataOutputs);
    for (int i = 0; i < kHostDataOutputs; ++i) o.emplace_back(eng, V, T);
    for (auto& x : o) po.push_back(&x);
    prog.process({&q0}, po);
    const size_t n = V * T * 64;
    for (int i = 0; i < kHostDataOutputs; ++i)
    {
      eng.check(mlgpu_layout_convert(eng.handle(), o[(size_t)i].data(), MLGPU_LAYOUT_QUAD, vm.data(), MLGPU_LAYOUT_VOICE_MAJOR, V, T));
      eng.check(mlgpu_download(eng.handle(), outs + (size_t)i * n, vm.data(), vm.bytes()));
    }
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

#include "dropin_ops.h"
extern "C" int ops_gpu_run(size_t V, size_t T, const float* in0, const float* in1, float* outs /* [kOpsOutputs][V][64 T] */, char* err, size_t errLen)
{
  try
  {
    gpu::Engine eng(0);
    AudioContext ctx(2, kOpsOutputs, 48000);
    gpu::VoiceProgram prog(eng, V, &ctx, opsProcess, nullptr);
    gpu::DeviceSignal vm(eng, V, T, MLGPU_LAYOUT_VOICE_MAJOR), q0(eng, V, T), q1(eng, V, T);
    eng.check(mlgpu_upload(eng.handle(), vm.data(), in0, vm.bytes()));
    eng.check(mlgpu_layout_convert(eng.handle(), vm.data(), MLGPU_LAYOUT_VOICE_MAJOR, q0.data(), MLGPU_LAYOUT_QUAD, V, T));
    eng.check(mlgpu_upload(eng.handle(), vm.data(), in1, vm.bytes()));
    eng.check(mlgpu_layout_convert(eng.handle(), vm.data(), MLGPU_LAYOUT_VOICE_MAJOR, q1.data(), MLGPU_LAYOUT_QUAD, V, T));
    std::vector<gpu::DeviceSignal> o;
    std::vector<gpu::DeviceSignal*> po;
    o.reserve(kOpsOutputs);
    for (int i = 0; i < kOpsOutputs; ++i) o.emplace_back(eng, V, T);
    for (auto& x : o) po.push_back(&x);
    prog.process({&q0, &q1}, po);
    const size_t n = V * T * 64;
    for (int i = 0; i < kOpsOutputs; ++i)
    {
      eng.check(mlgpu_layout_convert(eng.handle(), o[(size_t)i].data(), MLGPU_LAYOUT_QUAD, vm.data(), MLGPU_LAYOUT_VOICE_MAJOR, V, T));
      eng.check(mlgpu_download(eng.handle(), outs + (size_t)i * n, vm.data(), vm.bytes()));
    }
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

#include "dropin_routing.h"
extern "C" int routing_gpu_run(size_t V, size_t T, int launches, const float* in0, const float* in1, const float* in2, float* outs, char* err, size_t errLen)
{
  try
  {
    gpu::Engine eng(0);
    RoutingState state;
    routingSetup(state);
    AudioContext ctx(3, kRoutingOutputs, 48000);
    gpu::VoiceProgram prog(eng, V, &ctx, routingProcess, &state);
    const size_t Tl = T / (size_t)launches, S = T * 64, Sl = Tl * 64;
    gpu::DeviceSignal vm(eng, V, Tl, MLGPU_LAYOUT_VOICE_MAJOR);
    std::vector<gpu::DeviceSignal> q, o;
    for (int i = 0; i < 3; ++i) q.emplace_back(eng, V, Tl);
    for (int i = 0; i < kRoutingOutputs; ++i) o.emplace_back(eng, V, Tl);
    std::vector<gpu::DeviceSignal*> po;
    for (auto& y : o) po.push_back(&y);
    const float* src[3] = {in0, in1, in2};
    std::vector<float> h(V * Sl);
    for (int l = 0; l < launches; ++l)
    {
      for (int which = 0; which < 3; ++which)
      {
        for (size_t v = 0; v < V; ++v) memcpy(h.data() + v * Sl, src[which] + v * S + (size_t)l * Sl, sizeof(float) * Sl);
        eng.check(mlgpu_upload(eng.handle(), vm.data(), h.data(), vm.bytes()));
        eng.check(mlgpu_layout_convert(eng.handle(), vm.data(), MLGPU_LAYOUT_VOICE_MAJOR, q[(size_t)which].data(), MLGPU_LAYOUT_QUAD, V, Tl));
      }
      prog.process({&q[0], &q[1], &q[2]}, po);
      for (int i = 0; i < kRoutingOutputs; ++i)
      {
        eng.check(mlgpu_layout_convert(eng.handle(), o[(size_t)i].data(), MLGPU_LAYOUT_QUAD, vm.data(), MLGPU_LAYOUT_VOICE_MAJOR, V, Tl));
        eng.check(mlgpu_download(eng.handle(), h.data(), vm.data(), vm.bytes()));
        for (size_t v = 0; v < V; ++v) memcpy(outs + ((size_t)i * V + v) * S + (size_t)l * Sl, h.data() + v * Sl, sizeof(float) * Sl);
      }
    }
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

#include "dropin_objects.h"
extern "C" int objects_gpu_run_retrigger(size_t V, size_t T, int launches, const float* in0, const float* in1, float* outs, int retriggerAtLaunch, int everyOther,
                                         char* err, size_t errLen);
extern "C" int objects_gpu_run(size_t V, size_t T, int launches, const float* in0, const float* in1, float* outs /* [kObjectsOutputs][V][64 T] */, char* err,
                               size_t errLen)
{
  return objects_gpu_run_retrigger(V, T, launches, in0, in1, outs, -1, 0, err, errLen);
}
// retriggerAtLaunch >= 0: before that launch the one-shot is triggered again (gpu::VoiceProgram::trigger), on every voice or on the even ones
extern "C" int objects_gpu_run_retrigger(size_t V, size_t T, int launches, const float* in0, const float* in1, float* outs, int retriggerAtLaunch, int everyOther,
                                         char* err, size_t errLen)
{
  try
  {
    gpu::Engine eng(0);
    ObjectsState state;
    objectsSetup(state);
    AudioContext ctx(2, kObjectsOutputs, 48000);
    gpu::VoiceProgram prog(eng, V, &ctx, objectsProcess, &state);
    const size_t Tl = T / (size_t)launches, S = T * 64, Sl = Tl * 64;
    gpu::DeviceSignal vm(eng, V, Tl, MLGPU_LAYOUT_VOICE_MAJOR), q0(eng, V, Tl), q1(eng, V, Tl);
    std::vector<gpu::DeviceSignal> o;
    std::vector<gpu::DeviceSignal*> po;
    o.reserve(kObjectsOutputs);
    for (int i = 0; i < kObjectsOutputs; ++i) o.emplace_back(eng, V, Tl);
    for (auto& y : o) po.push_back(&y);
    std::vector<float> h(V * Sl);
    for (int l = 0; l < launches; ++l)  // state carried from launch to launch
    {
      if (l == retriggerAtLaunch)
      {
        if (!everyOther) prog.trigger(state.shot);
        else
        {
          std::vector<uint8_t> which(V);
          for (size_t v = 0; v < V; ++v) which[v] = (v & 1) == 0;
          prog.trigger(state.shot, which);
        }
      }
      for (int which = 0; which < 2; ++which)
      {
        const float* src = which ? in1 : in0;
        for (size_t v = 0; v < V; ++v) memcpy(h.data() + v * Sl, src + v * S + (size_t)l * Sl, sizeof(float) * Sl);
        eng.check(mlgpu_upload(eng.handle(), vm.data(), h.data(), vm.bytes()));
        eng.check(mlgpu_layout_convert(eng.handle(), vm.data(), MLGPU_LAYOUT_VOICE_MAJOR, (which ? q1 : q0).data(), MLGPU_LAYOUT_QUAD, V, Tl));
      }
      prog.process({&q0, &q1}, po);
      for (int i = 0; i < kObjectsOutputs; ++i)
      {
        eng.check(mlgpu_layout_convert(eng.handle(), o[(size_t)i].data(), MLGPU_LAYOUT_QUAD, vm.data(), MLGPU_LAYOUT_VOICE_MAJOR, V, Tl));
        eng.check(mlgpu_download(eng.handle(), h.data(), vm.data(), vm.bytes()));
        for (size_t v = 0; v < V; ++v) memcpy(outs + ((size_t)i * V + v) * S + (size_t)l * Sl, h.data() + v * Sl, sizeof(float) * Sl);
      }
    }
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

#include <array>
#include "dropin_synth.h"
struct SynthGpuEvent
{
  uint8_t type, channel;
  uint16_t sourceIdx;
  int32_t time;
  float value1, value2;
};
// nInstruments instruments; events[i] belongs to instrument eventInstrument[i]; out: [nInstruments][nBlocks * blockFrames] per channel
template <class SYNTH>
static int synth_gpu_run_t(size_t nInstruments, const SynthGpuEvent* events, const int* eventInstrument, int nEvents, float glideSeconds, float drift,
                           int blockFrames, int nBlocks, int vectorsPerLaunch, float* outL, float* outR, size_t scopeInstrument, float* scope,
                           size_t* scopeCounts, int scopeFramesPerRead, char* err, size_t errLen, bool eventRowsInKernel, int* rowsInKernel)
{
  try
  {
    gpu::Engine eng(0);
    SYNTH synth;
    gpu::VoiceProgramOptions opt;
    opt.eventRowsInKernel = eventRowsInKernel;
    opt.eventsOnOwnStream = getenv("MLGPU_TEST_EVENTS_OWN_STREAM") != nullptr;  // the same run with EventsToSignals on a second engine
    gpu::SynthProgram prog(eng, synth, nInstruments, 2, 48000, opt);
    if (rowsInKernel) *rowsInKernel = prog.program().eventRowsInKernel() ? 1 : 0;
    if (const char* want = getenv("MLGPU_TEST_EVENTS_OWN_STREAM"))  // "1": must have taken effect, "0": must have been declined
      if ((want[0] == '1') != prog.eventsOnOwnStream()) throw std::logic_error("eventsOnOwnStream: not what this synth should get");
    prog.setPublishedInstrument(scopeInstrument);
    size_t scopePos = 0;
    eng.check(mlgpu_events_set_pitch_glide_seconds(prog.events(), glideSeconds));
    eng.check(mlgpu_events_set_drift_amount(prog.events(), drift));
    const size_t S = (size_t)nBlocks * blockFrames;
    HostTransport host;
    for (int b = 0; b < nBlocks; ++b)
    {
      const int start = b * blockFrames;
      host.beforeBlock(b);
      prog.updateTime(host.ppq, host.bpm, host.playing, 48000.);  // one host application behind every instrument
      if (b == nBlocks / 2)
      {
        synth.setEnvelope(0.02f, 0.2f, 0.3f, 0.4f);
        prog.update();  // coefficients only: no live constants needed
      }
      for (int i = 0; i < nEvents; ++i)
        if (events[i].time >= start && events[i].time < start + blockFrames)
        {
          Event ev;
          ev.type = events[i].type;
          ev.channel = events[i].channel;
          ev.sourceIdx = events[i].sourceIdx;
          ev.time = events[i].time - start;
          ev.value1 = events[i].value1;
          ev.value2 = events[i].value2;
          prog.addInputEvent((size_t)eventInstrument[i], ev);
        }
      const int vecs = blockFrames / 64;
      for (int done = 0; done < vecs;)
      {
        const int n = (vecs - done < vectorsPerLaunch) ? vecs - done : vectorsPerLaunch;
        gpu::DeviceSignal mixL(eng, nInstruments, n, MLGPU_LAYOUT_VOICE_MAJOR), mixR(eng, nInstruments, n, MLGPU_LAYOUT_VOICE_MAJOR);
        prog.process((size_t)n, done * 64, {&mixL, &mixR});
        std::vector<float> hL(mixL.size()), hR(mixR.size());
        eng.check(mlgpu_download(eng.handle(), hL.data(), mixL.data(), mixL.bytes()));
        eng.check(mlgpu_download(eng.handle(), hR.data(), mixR.data(), mixR.bytes()));
        for (size_t i = 0; i < nInstruments; ++i)
        {
          memcpy(outL + i * S + start + (size_t)done * 64, hL.data() + i * (size_t)n * 64, sizeof(float) * (size_t)n * 64);
          memcpy(outR + i * S + start + (size_t)done * 64, hR.data() + i * (size_t)n * 64, sizeof(float) * (size_t)n * 64);
        }
        done += n;
      }
      prog.clearInputEvents();
      host.afterBlock(blockFrames);
      if (scope)
      {
        scopeCounts[b] = synth.getPublishedSignals()["scope"]->read(scope + scopePos, scopeFramesPerRead);
        scopePos += scopeCounts[b];
      }
    }
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

extern "C" int synth_gpu_run(size_t nInstruments, const SynthGpuEvent* events, const int* eventInstrument, int nEvents, float glideSeconds, float drift,
                             int blockFrames, int nBlocks, int vectorsPerLaunch, float* outL, float* outR, size_t scopeInstrument, float* scope,
                             size_t* scopeCounts, int scopeFramesPerRead, char* err, size_t errLen)
{
  return synth_gpu_run_t<SmallSynth>(nInstruments, events, eventInstrument, nEvents, glideSeconds, drift, blockFrames, nBlocks, vectorsPerLaunch, outL, outR,
                                     scopeInstrument, scope, scopeCounts, scopeFramesPerRead, err, errLen, false, nullptr);
}
// the synth that reads the instrument's controllers through the AudioContext
extern "C" int controller_synth_gpu_run(size_t nInstruments, const SynthGpuEvent* events, const int* eventInstrument, int nEvents, float glideSeconds, float drift,
                                        int blockFrames, int nBlocks, int vectorsPerLaunch, float* outL, float* outR, int eventRowsInKernel, int* rowsInKernel,
                                        char* err, size_t errLen)
{
  return synth_gpu_run_t<ControllerSynth>(nInstruments, events, eventInstrument, nEvents, glideSeconds, drift, blockFrames, nBlocks, vectorsPerLaunch, outL, outR,
                                          0, nullptr, nullptr, 0, err, errLen, eventRowsInKernel != 0, rowsInKernel);
}
// ---- controllers-to-audio: N instances of the process function, each with its own controllers (one context per voice) ----
#include "dropin_controllers.h"
extern "C" int ctl_audio_gpu_run(size_t N, const SynthGpuEvent* events, const int* eventInstrument, int nEvents, int blockFrames, int nBlocks,
                                 int vectorsPerLaunch, float* out /* [N][nBlocks * blockFrames] */, char* err, size_t errLen)
{
  mlgpu_events* ev = nullptr;
  try
  {
    gpu::Engine eng(0);
    CtlAudioState state;
    state.sineGens.resize(state.sineControllers.size());
    AudioContext ctx(0, 2, 48000);
    gpu::VoiceProgramOptions opt;
    opt.voicesPerContext = 1;
    gpu::VoiceProgram prog(eng, N, &ctx, ctlAudioProcess, &state, opt);
    // the controllers the captured code reads, in the order it first asked for them
    std::vector<int> numbers = prog.contextInputs();
    eng.check(mlgpu_events_create(eng.handle(), N, 1, &ev));
    eng.check(mlgpu_events_set_sample_rate(ev, 48000.));
    eng.check(mlgpu_events_set_wanted_rows(ev, 0));  // no voice rows: only the controllers
    eng.check(mlgpu_events_watch_controllers(ev, numbers.data(), (int)numbers.size(), (size_t)vectorsPerLaunch));
    std::vector<const float*> contextSignals;
    for (size_t c = 0; c < numbers.size(); ++c) contextSignals.push_back(mlgpu_events_controller_signal(ev, (int)c));
    float* noRows[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    const size_t S = (size_t)nBlocks * blockFrames;
    for (int b = 0; b < nBlocks; ++b)
    {
      const int start = b * blockFrames;
      for (int i = 0; i < nEvents; ++i)
        if (events[i].time >= start && events[i].time < start + blockFrames)
        {
          mlgpu_event m{events[i].type, events[i].channel, events[i].sourceIdx, events[i].time - start, events[i].value1, events[i].value2};
          eng.check(mlgpu_events_add_event(ev, (size_t)eventInstrument[i], &m));
        }
      const int vecs = blockFrames / 64;
      for (int done = 0; done < vecs;)
      {
        const int n = (vecs - done < vectorsPerLaunch) ? vecs - done : vectorsPerLaunch;
        gpu::DeviceSignal o0(eng, N, (size_t)n, MLGPU_LAYOUT_VOICE_MAJOR), o1(eng, N, (size_t)n, MLGPU_LAYOUT_VOICE_MAJOR);
        eng.check(mlgpu_events_process(ev, (size_t)n, done * 64, noRows, MLGPU_LAYOUT_QUAD));
        prog.process({}, {&o0, &o1}, nullptr, contextSignals.data());
        std::vector<float> h(o0.size());
        eng.check(mlgpu_download(eng.handle(), h.data(), o0.data(), o0.bytes()));
        for (size_t i = 0; i < N; ++i) memcpy(out + i * S + start + (size_t)done * 64, h.data() + i * (size_t)n * 64, sizeof(float) * (size_t)n * 64);
        done += n;
      }
      eng.check(mlgpu_events_clear_events(ev));
    }
    mlgpu_events_destroy(ev);
    return 0;
  }
  catch (const gpu::Error& e)
  {
    if (ev) mlgpu_events_destroy(ev);
    if (err && errLen) snprintf(err, errLen, "%s", e.what());
    return e.status ? e.status : -1;
  }
  catch (const std::exception& e)
  {
    if (ev) mlgpu_events_destroy(ev);
    if (err && errLen) snprintf(err, errLen, "%s", e.what());
    return -1;
  }
}

// ---- a bank of plug-ins: nInstruments PluginSynths behind ONE mlgpu_process_buffer (the host's block sizes), their events and
// time reports per block, the instruments' outputs summed in instrument order into the host's stereo pair ----
namespace
{
struct PluginBank
{
  gpu::Engine* eng;
  gpu::SynthProgram* prog;
  gpu::DeviceSignal *mixL, *mixR;
  size_t nInstruments;
  std::string error;
};
int pluginBankVectors(void* user, size_t nVectors, const float* const*, float* const* d_out)
{
  PluginBank* pb = static_cast<PluginBank*>(user);
  try
  {
    // event times are relative to the host block, the first DSPVector computed in this call starts at 0 (MLSignalProcessBuffer.cpp:55-66)
    pb->prog->process(nVectors, 0, {pb->mixL, pb->mixR});
    pb->eng->check(mlgpu_mixdown_groups(pb->eng->handle(), pb->mixL->data(), pb->mixL->layout(), 1, pb->nInstruments, nVectors, d_out[0], MLGPU_LAYOUT_QUAD));
    pb->eng->check(mlgpu_mixdown_groups(pb->eng->handle(), pb->mixR->data(), pb->mixR->layout(), 1, pb->nInstruments, nVectors, d_out[1], MLGPU_LAYOUT_QUAD));
    return MLGPU_OK;
  }
  catch (const std::exception& e)
  {
    pb->error = e.what();
    return MLGPU_ERR_INVALID;
  }
}
}  // namespace
extern "C" int plugin_gpu_run(size_t nInstruments, const SynthGpuEvent* events, const int* eventInstrument, int nEvents, float glideSeconds, float drift,
                              const int* blocks, int nBlocks, int maxFrames, int eventRowsInKernel, float* outL, float* outR, char* err, size_t errLen)
{
  mlgpu_process_buffer* spb = nullptr;
  try
  {
    gpu::Engine eng(0);
    PluginSynth synth;
    gpu::VoiceProgramOptions opt;
    opt.eventRowsInKernel = eventRowsInKernel != 0;
    gpu::SynthProgram prog(eng, synth, nInstruments, 2, 48000, opt);
    eng.check(mlgpu_events_set_pitch_glide_seconds(prog.events(), glideSeconds));
    eng.check(mlgpu_events_set_drift_amount(prog.events(), drift));
    const size_t maxVectors = (size_t)maxFrames / 64 + 2;
    gpu::DeviceSignal mixL(eng, nInstruments, maxVectors), mixR(eng, nInstruments, maxVectors);
    PluginBank bank{&eng, &prog, &mixL, &mixR, nInstruments, {}};
    eng.check(mlgpu_process_buffer_create(eng.handle(), 0, 2, (size_t)maxFrames, &spb));
    HostTransport host;
    int pos = 0, st = MLGPU_OK;
    for (int b = 0; b < nBlocks && st == MLGPU_OK; ++b)
    {
      host.beforeBlock(b);
      prog.updateTime(host.ppq, host.bpm, host.playing, 48000.);
      for (int i = 0; i < nEvents; ++i)
        if (events[i].time >= pos && events[i].time < pos + blocks[b])
        {
          Event ev;
          ev.type = events[i].type;
          ev.channel = events[i].channel;
          ev.sourceIdx = events[i].sourceIdx;
          ev.time = events[i].time - pos;
          ev.value1 = events[i].value1;
          ev.value2 = events[i].value2;
          prog.addInputEvent((size_t)eventInstrument[i], ev);
        }
      float* outs[2] = {outL + pos, outR + pos};
      st = mlgpu_process_buffer_process(spb, nullptr, outs, blocks[b], pluginBankVectors, &bank);
      prog.clearInputEvents();  // SignalProcessBuffer::process ends with context->clearInputEvents() (:89)
      host.afterBlock(blocks[b]);
      pos += blocks[b];
    }
    mlgpu_process_buffer_destroy(spb);
    spb = nullptr;
    if (st != MLGPU_OK)
    {
      if (err && errLen) snprintf(err, errLen, "%s / %s", bank.error.c_str(), mlgpu_last_error(eng.handle()));
      return st;
    }
    return 0;
  }
  catch (const gpu::Error& e)
  {
    if (spb) mlgpu_process_buffer_destroy(spb);
    if (err && errLen) snprintf(err, errLen, "%s", e.what());
    return e.status ? e.status : -1;
  }
  catch (const std::exception& e)
  {
    if (spb) mlgpu_process_buffer_destroy(spb);
    if (err && errLen) snprintf(err, errLen, "%s", e.what());
    return -1;
  }
}

// the synth with the tempo-synced tremolo (ctx->getBeatPhase() into a TempoLock per voice)
extern "C" int tempo_synth_gpu_run(size_t nInstruments, const SynthGpuEvent* events, const int* eventInstrument, int nEvents, float glideSeconds, float drift,
                                   int blockFrames, int nBlocks, int vectorsPerLaunch, float* outL, float* outR, int eventRowsInKernel, int* rowsInKernel,
                                   char* err, size_t errLen)
{
  return synth_gpu_run_t<TempoSynth>(nInstruments, events, eventInstrument, nEvents, glideSeconds, drift, blockFrames, nBlocks, vectorsPerLaunch, outL, outR, 0,
                                     nullptr, nullptr, 0, err, errLen, eventRowsInKernel != 0, rowsInKernel);
}
// the pitch-and-gate-only synth; eventRowsInKernel: the voice kernel computes the two rows itself; *rowsInKernel: whether it did
extern "C" int lean_synth_gpu_run(size_t nInstruments, const SynthGpuEvent* events, const int* eventInstrument, int nEvents, float glideSeconds, float drift,
                                  int blockFrames, int nBlocks, int vectorsPerLaunch, float* outL, float* outR, int eventRowsInKernel, int* rowsInKernel,
                                  char* err, size_t errLen)
{
  return synth_gpu_run_t<LeanSynth>(nInstruments, events, eventInstrument, nEvents, glideSeconds, drift, blockFrames, nBlocks, vectorsPerLaunch, outL, outR, 0,
                                    nullptr, nullptr, 0, err, errLen, eventRowsInKernel != 0, rowsInKernel);
}

