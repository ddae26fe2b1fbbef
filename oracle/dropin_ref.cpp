// oracle/dropin_ref.cpp — TEST INFRASTRUCTURE. Compiles tests/cpp/dropin_patch.h (user code) against the
// UNMODIFIED reference (its headers and its own AudioContext) and runs it the reference's way: one state object per
// voice, the process function called once per 64-frame vector. Output: oracle/_ref/libdropin_ref.so.
#include <cstddef>
#include <cstring>

#include "madronalib.h"
using namespace ml;

#include "../tests/cpp/dropin_patch.h"

extern "C" int dropin_ref_run(size_t V, size_t T, const float* gate, const float* pitch, float* out0, float* out1)
{
  const size_t S = T * kFloatsPerDSPVector;
  for (size_t v = 0; v < V; ++v)
  {
    PatchState state;
    patchSetup(state);
    AudioContext ctx(2, 2, 48000);
    for (size_t t = 0; t < T; ++t)
    {
      load(ctx.inputs[0], gate + v * S + t * kFloatsPerDSPVector);
      load(ctx.inputs[1], pitch + v * S + t * kFloatsPerDSPVector);
      patchProcess(&ctx, &state);
      store(ctx.outputs[0], out0 + v * S + t * kFloatsPerDSPVector);
      store(ctx.outputs[1], out1 + v * S + t * kFloatsPerDSPVector);
    }
  }
  return 0;
}

#include "../tests/cpp/dropin_reverb.h"
// knobsAt: the vector before which the host turns the knobs (plateTurnKnobs); >= T: never
extern "C" int plate_ref_run(size_t V, size_t T, size_t knobsAt, const float* inL, const float* inR, float* outL, float* outR)
{
  const size_t S = T * kFloatsPerDSPVector;
  for (size_t v = 0; v < V; ++v)
  {
    PlateState state;
    plateSetup(state);
    AudioContext ctx(2, 2, 48000);
    for (size_t t = 0; t < T; ++t)
    {
      if (t == knobsAt) plateTurnKnobs(state);
      load(ctx.inputs[0], inL + v * S + t * kFloatsPerDSPVector);
      load(ctx.inputs[1], inR + v * S + t * kFloatsPerDSPVector);
      plateProcess(&ctx, &state);
      store(ctx.outputs[0], outL + v * S + t * kFloatsPerDSPVector);
      store(ctx.outputs[1], outR + v * S + t * kFloatsPerDSPVector);
    }
  }
  return 0;
}

#include "../tests/cpp/dropin_oversample.h"
extern "C" int oversample_ref_run(size_t V, size_t T, const float* in0, const float* in1, float* out0, float* out1)
{
  const size_t S = T * kFloatsPerDSPVector;
  for (size_t v = 0; v < V; ++v)
  {
    OversampleState state;
    oversampleSetup(state);
    AudioContext ctx(2, 2, 48000);
    for (size_t t = 0; t < T; ++t)
    {
      load(ctx.inputs[0], in0 + v * S + t * kFloatsPerDSPVector);
      load(ctx.inputs[1], in1 + v * S + t * kFloatsPerDSPVector);
      oversampleProcess(&ctx, &state);
      store(ctx.outputs[0], out0 + v * S + t * kFloatsPerDSPVector);
      store(ctx.outputs[1], out1 + v * S + t * kFloatsPerDSPVector);
    }
  }
  return 0;
}

#include "../tests/cpp/dropin_decay.h"
extern "C" int decay_ref_run(size_t V, size_t T, int flush, const float* in0, float* out0, float* out1)
{
  const size_t S = T * kFloatsPerDSPVector;
  for (size_t v = 0; v < V; ++v)
  {
    DecayState state;
    decaySetup(state);
    AudioContext ctx(1, 2, 48000);
    for (size_t t = 0; t < T; ++t)
    {
      load(ctx.inputs[0], in0 + v * S + t * kFloatsPerDSPVector);
      if (flush)
        decayProcessFlush(&ctx, &state);
      else
        decayProcess(&ctx, &state);
      store(ctx.outputs[0], out0 + v * S + t * kFloatsPerDSPVector);
      store(ctx.outputs[1], out1 + v * S + t * kFloatsPerDSPVector);
    }
  }
  return 0;
}

// ---- host data (tests/cpp/dropin_hostdata.h) ----
#include "../tests/cpp/dropin_hostdata.h"
extern "C" int hostdata_ref_run(size_t V, size_t T, const float* in0, float* outs /* [kHostDataOutputs][V][64 T] */)
{
  const size_t S = T * kFloatsPerDSPVector;
  for (size_t v = 0; v < V; ++v)
  {
    HostDataState state;
    hostDataSetup(&state);
    AudioContext ctx(1, kHostDataOutputs, 48000);
    for (size_t t = 0; t < T; ++t)
    {
      load(ctx.inputs[0], in0 + v * S + t * kFloatsPerDSPVector);
      hostDataProcess(&ctx, &state);
      for (int o = 0; o < kHostDataOutputs; ++o) store(ctx.outputs[o], outs + ((size_t)o * V + v) * S + t * kFloatsPerDSPVector);
    }
  }
  return 0;
}

// ---- imperative code: objects made, called and read on the spot (tests/cpp/dropin_eager.h) ----
#include "../tests/cpp/dropin_eager.h"
// returns the number of floats recorded (and copies min(that, cap) of them); names: "name@start;" per block
extern "C" long immediate_ref_run(float* out, size_t cap, char* names, size_t namesLen)
{
  ImmediateLog log;
  immediateSuite(log);
  for (size_t i = 0; i < log.data.size() && i < cap; ++i) out[i] = log.data[i];
  std::string nm;
  for (size_t i = 0; i < log.names.size(); ++i) nm += log.names[i] + "@" + std::to_string(log.starts[i]) + ";";
  if (names && namesLen) snprintf(names, namesLen, "%s", nm.c_str());
  return (long)log.data.size();
}

// ---- every free function of MLDSPOps.h by name (tests/cpp/dropin_ops.h) ----
#include "../tests/cpp/dropin_ops.h"
extern "C" int ops_ref_run(size_t V, size_t T, const float* in0, const float* in1, float* outs /* [kOpsOutputs][V][64 T] */)
{
  const size_t S = T * kFloatsPerDSPVector;
  for (size_t v = 0; v < V; ++v)
  {
    AudioContext ctx(2, kOpsOutputs, 48000);
    for (size_t t = 0; t < T; ++t)
    {
      load(ctx.inputs[0], in0 + v * S + t * kFloatsPerDSPVector);
      load(ctx.inputs[1], in1 + v * S + t * kFloatsPerDSPVector);
      opsProcess(&ctx, nullptr);
      for (int o = 0; o < kOpsOutputs; ++o) store(ctx.outputs[o], outs + ((size_t)o * V + v) * S + t * kFloatsPerDSPVector);
    }
  }
  return 0;
}

// ---- routing functions and function wrappers by name (tests/cpp/dropin_routing.h) ----
#include "../tests/cpp/dropin_routing.h"
extern "C" int routing_ref_run(size_t V, size_t T, const float* in0, const float* in1, const float* in2, float* outs /* [kRoutingOutputs][V][64 T] */)
{
  const size_t S = T * kFloatsPerDSPVector;
  const float* ins[3] = {in0, in1, in2};
  for (size_t v = 0; v < V; ++v)
  {
    RoutingState state;
    routingSetup(state);
    AudioContext ctx(3, kRoutingOutputs, 48000);
    for (size_t t = 0; t < T; ++t)
    {
      for (int i = 0; i < 3; ++i) load(ctx.inputs[i], ins[i] + v * S + t * kFloatsPerDSPVector);
      routingProcess(&ctx, &state);
      for (int o = 0; o < kRoutingOutputs; ++o) store(ctx.outputs[o], outs + ((size_t)o * V + v) * S + t * kFloatsPerDSPVector);
    }
  }
  return 0;
}

// ---- the stateful objects the other drop-ins do not touch, by name (tests/cpp/dropin_objects.h) ----
#include "../tests/cpp/dropin_objects.h"
// retriggerAt >= 0: OneShotGen::trigger() is called again before that DSPVector - on every voice, or (everyOther) on the even ones
extern "C" int objects_ref_run_retrigger(size_t V, size_t T, const float* in0, const float* in1, float* outs, int retriggerAt, int everyOther);
extern "C" int objects_ref_run(size_t V, size_t T, const float* in0, const float* in1, float* outs /* [kObjectsOutputs][V][64 T] */)
{
  return objects_ref_run_retrigger(V, T, in0, in1, outs, -1, 0);
}
extern "C" int objects_ref_run_retrigger(size_t V, size_t T, const float* in0, const float* in1, float* outs, int retriggerAt, int everyOther)
{
  const size_t S = T * kFloatsPerDSPVector;
  for (size_t v = 0; v < V; ++v)
  {
    ObjectsState state;
    objectsSetup(state);
    AudioContext ctx(2, kObjectsOutputs, 48000);
    for (size_t t = 0; t < T; ++t)
    {
      if ((int)t == retriggerAt && (!everyOther || (v & 1) == 0)) state.shot.trigger();
      load(ctx.inputs[0], in0 + v * S + t * kFloatsPerDSPVector);
      load(ctx.inputs[1], in1 + v * S + t * kFloatsPerDSPVector);
      objectsProcess(&ctx, &state);
      for (int o = 0; o < kObjectsOutputs; ++o) store(ctx.outputs[o], outs + ((size_t)o * V + v) * S + t * kFloatsPerDSPVector);
    }
  }
  return 0;
}

// ---- the app layer: AudioContext with its events, Synth, SignalProcessBuffer (in the immediate build the shim's own, stepped by these loops) ----
// ---- a Synth subclass run by the reference's own Synth::processVector, AudioContext and EventsToSignals ----
#include "MLSynth.h"
#include "../tests/cpp/dropin_synth.h"
struct SynthRefEvent
{
  uint8_t type, channel;
  uint16_t sourceIdx;
  int32_t time;
  float value1, value2;
};
// one instrument; events with absolute onset times; host blocks of blockFrames; out: [2][nBlocks * blockFrames]
// scope / scopeCounts (may be NULL): after every block the UI side reads what the "scope" published signal holds
// (PublishedSignal::read of up to scopeFramesPerRead frames of 2 channels); scopeCounts[b] = floats read after block b.
template <class SYNTH>
static int synth_ref_run_t(const SynthRefEvent* events, int nEvents, float glideSeconds, float drift, int blockFrames, int nBlocks, float* outL, float* outR,
                           float* scope, size_t* scopeCounts, int scopeFramesPerRead)
{
  size_t scopePos = 0;
  SYNTH synth;
  AudioContext ctx(0, 2, 48000);
  ctx.setInputPolyphony(kSynthVoices);
  ctx.setInputGlideTimeInSeconds(glideSeconds);
  ctx.setInputDriftAmount(drift);
  HostTransport host;
  for (int b = 0; b < nBlocks; ++b)
  {
    const int start = b * blockFrames;
    host.beforeBlock(b);
    ctx.updateTime(host.ppq, host.bpm, host.playing, 48000.);
    if (b == nBlocks / 2) synth.setEnvelope(0.02f, 0.2f, 0.3f, 0.4f);  // the host turns the envelope knobs half way through
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
        ctx.addInputEvent(ev);
      }
    for (int off = 0; off < blockFrames; off += kFloatsPerDSPVector)
    {
      ctx.processVector(off);
      synth.processVector(ctx.inputs, ctx.outputs, &ctx);
      store(ctx.outputs[0], outL + start + off);
      store(ctx.outputs[1], outR + start + off);
    }
    ctx.clearInputEvents();
    host.afterBlock(blockFrames);
    if (scope)
    {
      scopeCounts[b] = synth.getPublishedSignals()["scope"]->read(scope + scopePos, scopeFramesPerRead);
      scopePos += scopeCounts[b];
    }
  }
  return 0;
}

extern "C" int synth_ref_run(const SynthRefEvent* events, int nEvents, float glideSeconds, float drift, int blockFrames, int nBlocks, float* outL, float* outR,
                             float* scope, size_t* scopeCounts, int scopeFramesPerRead)
{
  return synth_ref_run_t<SmallSynth>(events, nEvents, glideSeconds, drift, blockFrames, nBlocks, outL, outR, scope, scopeCounts, scopeFramesPerRead);
}
extern "C" int controller_synth_ref_run(const SynthRefEvent* events, int nEvents, float glideSeconds, float drift, int blockFrames, int nBlocks, float* outL,
                                        float* outR)
{
  return synth_ref_run_t<ControllerSynth>(events, nEvents, glideSeconds, drift, blockFrames, nBlocks, outL, outR, nullptr, nullptr, 0);
}
extern "C" int tempo_synth_ref_run(const SynthRefEvent* events, int nEvents, float glideSeconds, float drift, int blockFrames, int nBlocks, float* outL, float* outR)
{
  return synth_ref_run_t<TempoSynth>(events, nEvents, glideSeconds, drift, blockFrames, nBlocks, outL, outR, nullptr, nullptr, 0);
}
extern "C" int lean_synth_ref_run(const SynthRefEvent* events, int nEvents, float glideSeconds, float drift, int blockFrames, int nBlocks, float* outL, float* outR)
{
  return synth_ref_run_t<LeanSynth>(events, nEvents, glideSeconds, drift, blockFrames, nBlocks, outL, outR, nullptr, nullptr, 0);
}

// ---- the reference's SignalProcessBuffer driven with a sequence of host block sizes -----------------------------
// process function: out0 = Lopass(in0) (stateful), out1 = in0 * 0.5. blocks[i] frames per call; in / out0 / out1
// hold the concatenated blocks. Pins the behaviour of mlgpu_process_buffer (latency, zero-fill, ring sizes).
#include "MLSignalProcessBuffer.h"
namespace
{
struct SpbState
{
  Lopass lp;
};
void spbProcess(AudioContext* ctx, void* st)
{
  auto s = static_cast<SpbState*>(st);
  ctx->outputs[0] = s->lp(ctx->inputs[0]);
  ctx->outputs[1] = ctx->inputs[0] * 0.5f;
}
}  // namespace

extern "C" int spb_ref_run(int maxFrames, const int* blocks, int nBlocks, const float* in, float* out0, float* out1)
{
  SpbState state;
  state.lp.coeffs = Lopass::makeCoeffs(0.05f, 0.9f);
  AudioContext ctx(1, 2, 48000);
  SignalProcessBuffer spb(1, 2, maxFrames);
  size_t pos = 0;
  for (int b = 0; b < nBlocks; ++b)
  {
    const float* ins[1] = {in + pos};
    float* outs[2] = {out0 + pos, out1 + pos};
    spb.process(ins, outs, blocks[b], &ctx, spbProcess, &state);
    pos += blocks[b];
  }
  return 0;
}

// ---- a plug-in: PluginSynth behind the reference's SignalProcessBuffer, AudioContext and EventsToSignals, host blocks of
// arbitrary sizes; events[i].time is absolute, handed over relative to the block it falls into ----
extern "C" int plugin_ref_run(const SynthRefEvent* events, int nEvents, float glideSeconds, float drift, const int* blocks, int nBlocks, int maxFrames,
                              float* outL, float* outR)
{
  PluginSynth synth;
  AudioContext ctx(0, 2, 48000);
  ctx.setInputPolyphony(kSynthVoices);
  ctx.setInputGlideTimeInSeconds(glideSeconds);
  ctx.setInputDriftAmount(drift);
  SignalProcessBuffer spb(0, 2, maxFrames);
  HostTransport host;
  int pos = 0;
  for (int b = 0; b < nBlocks; ++b)
  {
    host.beforeBlock(b);
    ctx.updateTime(host.ppq, host.bpm, host.playing, 48000.);
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
        ctx.addInputEvent(ev);
      }
    float* outs[2] = {outL + pos, outR + pos};
    spb.process(nullptr, outs, blocks[b], &ctx, [](AudioContext* c, void* s) { static_cast<PluginSynth*>(s)->processVector(c->inputs, c->outputs, c); }, &synth);
    host.afterBlock(blocks[b]);
    pos += blocks[b];
  }
  return 0;
}

// ---- the reference's EventsToSignals driven like AudioContext / SignalProcessBuffer drive it -------------------------
// events: absolute onset times in frames; the harness cuts time into host blocks of blockFrames (a multiple of 64), adds the
// events of a block with block-relative times, calls processVector(offset) per 64 frames and clearEvents() per block.
// out: [8 rows][polyphony voices][nBlocks * blockFrames] in the order of VoiceOutputSignals.
struct RefEvent
{
  uint8_t type, channel;
  uint16_t sourceIdx;
  int32_t time;
  float value1, value2;
};
static int e2sRefRun(int polyphony, int mpe, int unison, double sr, float glideSeconds, float drift, float bendRange, float mpeBendRange, int modCC,
                     const RefEvent* events, int nEvents, int blockFrames, int nBlocks, float* out, const int* ctlNumbers, int nCtl, float* ctlOut,
                     int ctlFromVector = 0)
{
  EventsToSignals e2s;
  e2s.setSampleRate(sr);
  if (ctlFromVector < 0)  // a host that settles on its voice count in two steps, with an event in between (setPolyphony = clear(), :322-327)
  {
    ctlFromVector = -ctlFromVector;
    e2s.setPolyphony(polyphony == 16 ? 8 : polyphony + 1);
    e2s.setPitchBendInSemitones(3.f);
    Event ev;
    ev.type = kNoteOn;
    ev.channel = 1;
    ev.sourceIdx = 60;
    ev.time = 0;
    ev.value1 = 60.f;
    ev.value2 = 0.8f;
    e2s.addEvent(ev);
  }
  e2s.setPolyphony(polyphony);
  e2s.setProtocol(mpe ? Symbol("MPE") : Symbol("MIDI"));
  e2s.setUnison(unison != 0);
  e2s.setPitchGlideInSeconds(glideSeconds);
  e2s.setDriftAmount(drift);
  e2s.setPitchBendInSemitones(bendRange);
  e2s.setMPEPitchBendInSemitones(mpeBendRange);
  e2s.setModCC(modCC);
  const size_t S = (size_t)nBlocks * blockFrames;
  for (int b = 0; b < nBlocks; ++b)
  {
    const int start = b * blockFrames;
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
        e2s.addEvent(ev);
      }
    for (int off = 0; off < blockFrames; off += kFloatsPerDSPVector)
    {
      e2s.processVector(off);
      for (int v = 0; v < polyphony; ++v)
        for (int r = 0; r < kNumVoiceOutputRows; ++r)
          store(e2s.getVoice(v).outputs.constRow(r), out + ((size_t)r * polyphony + v) * S + start + off);
      // what AudioContext::getInputController(n) hands a process function (MLAudioContext.cpp:129)
      // (ctlFromVector > 0: a process function that starts to look at its controllers late - the signals before stay zero here)
      if ((start + off) / (int)kFloatsPerDSPVector >= ctlFromVector)
        for (int c = 0; c < nCtl; ++c) store(e2s.getController((size_t)ctlNumbers[c]).output, ctlOut + (size_t)c * S + start + off);
    }
    e2s.clearEvents();
  }
  return 0;
}
extern "C" int e2s_ref_run(int polyphony, int mpe, int unison, double sr, float glideSeconds, float drift, float bendRange, float mpeBendRange, int modCC,
                           const RefEvent* events, int nEvents, int blockFrames, int nBlocks, float* out)
{
  return e2sRefRun(polyphony, mpe, unison, sr, glideSeconds, drift, bendRange, mpeBendRange, modCC, events, nEvents, blockFrames, nBlocks, out, nullptr, 0, nullptr);
}
// the same performance, and the smoothed signals of controllers ctlNumbers[] next to the voice rows: ctlOut [nCtl][frames]
extern "C" int e2s_ref_run_controllers(int polyphony, int mpe, int unison, double sr, float glideSeconds, float drift, float bendRange, float mpeBendRange,
                                       int modCC, const RefEvent* events, int nEvents, int blockFrames, int nBlocks, float* out, const int* ctlNumbers,
                                       int nCtl, float* ctlOut)
{
  return e2sRefRun(polyphony, mpe, unison, sr, glideSeconds, drift, bendRange, mpeBendRange, modCC, events, nEvents, blockFrames, nBlocks, out, ctlNumbers, nCtl,
                   ctlOut);
}
extern "C" int e2s_ref_run_controllers_from(int polyphony, int mpe, int unison, double sr, float glideSeconds, float drift, float bendRange, float mpeBendRange,
                                            int modCC, const RefEvent* events, int nEvents, int blockFrames, int nBlocks, float* out, const int* ctlNumbers,
                                            int nCtl, float* ctlOut, int ctlFromVector)
{
  return e2sRefRun(polyphony, mpe, unison, sr, glideSeconds, drift, bendRange, mpeBendRange, modCC, events, nEvents, blockFrames, nBlocks, out, ctlNumbers, nCtl,
                   ctlOut, ctlFromVector);
}

// ---- controllers-to-audio (tests/cpp/dropin_controllers.h): one AudioContext with its controller events ----
#include "../tests/cpp/dropin_controllers.h"
extern "C" int ctl_audio_ref_run(const RefEvent* events, int nEvents, int blockFrames, int nBlocks, float* out)
{
  CtlAudioState state;
  state.sineGens.resize(state.sineControllers.size());
  AudioContext ctx(0, 2, 48000);
  for (int b = 0; b < nBlocks; ++b)
  {
    const int start = b * blockFrames;
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
        ctx.addInputEvent(ev);
      }
    for (int off = 0; off < blockFrames; off += kFloatsPerDSPVector)
    {
      ctx.processVector(off);
      ctlAudioProcess(&ctx, &state);
      store(ctx.outputs[0], out + start + off);
    }
    ctx.clearInputEvents();
  }
  return 0;
}

// ---- AudioContext's transport (ProcessTime, MLAudioContext.cpp:16-104) driven as a plug-in wrapper drives it: updateTime()
// with what the host reports before a block, processVector() per 64 frames, getBeatPhase() read by the process function ----
struct TransportStep
{
  int kind;  // 0: updateTime(ppq, bpm, playing, sr)   1: `vectors` x processVector   2: clear()
  int vectors, playing, pad;
  double ppq, bpm, sr;
};
extern "C" int transport_ref_run(const TransportStep* steps, int nSteps, float* out, uint64_t* samplesSinceStart)
{
  AudioContext ctx(0, 1, 48000);
  size_t pos = 0;
  for (int i = 0; i < nSteps; ++i)
  {
    const TransportStep& st = steps[i];
    if (st.kind == 0) ctx.updateTime(st.ppq, st.bpm, st.playing != 0, st.sr);
    else if (st.kind == 2) ctx.clear();
    else
      for (int v = 0; v < st.vectors; ++v)
      {
        ctx.processVector(v * kFloatsPerDSPVector);
        store(ctx.getBeatPhase(), out + pos);
        pos += kFloatsPerDSPVector;
      }
    samplesSinceStart[i] = ctx.getTimeInfo().samplesSinceStart;
  }
  return 0;
}

// ---- SignalProcessor::PublishedSignal driven as processors drive it: storePublishedSignal per voice in rotation ----
// ops[i]: 0 = write the next DSPVector of every voice (2 channels), 1 = read(args[i] frames), 2 = readLatest(args[i]),
// 3 = peekLatest(args[i]). Results of the read ops are concatenated in `out`; counts[i] = floats the op returned.
extern "C" int published_ref_run(int maxFrames, int maxVoices, int octavesDown, int nVoices, size_t T, const float* ch0, const float* ch1,
                                 const int* ops, const int* args, int nOps, float* out, size_t* counts)
{
  SignalProcessor::PublishedSignal ps(maxFrames, maxVoices, 2, octavesDown);
  size_t t = 0, pos = 0;
  for (int i = 0; i < nOps; ++i)
  {
    counts[i] = 0;
    if (ops[i] == 0)
    {
      if (t >= T) return 1;
      for (int v = 0; v < nVoices; ++v)
      {
        DSPVectorArray<2> x;
        load(x.row(0), ch0 + ((size_t)v * T + t) * kFloatsPerDSPVector);
        load(x.row(1), ch1 + ((size_t)v * T + t) * kFloatsPerDSPVector);
        ps.writeQuick(x, kFloatsPerDSPVector, v);
      }
      ++t;
    }
    else if (ops[i] == 1)
      counts[i] = ps.read(out + pos, args[i]);
    else if (ops[i] == 2)
      counts[i] = ps.readLatest(out + pos, args[i]);
    else
    {
      ps.peekLatest(out + pos, args[i]);
      counts[i] = (size_t)args[i] * 2;
    }
    pos += (size_t)args[i] * 2 * (ops[i] != 0);
  }
  return 0;
}
