// The reference's examples/audio-and-midi/controllers-to-audio.cpp, included unchanged and compiled against the MI355X shim.
// Its process function reads sample 0 of eight smoothed controller signals into host floats and maps them through a
// std::function projection before they become oscillator frequencies (`ctrlToFreq(ctrlSig[0])`): host code on signal values,
// once per DSPVector. gpu::VoiceProgramOptions::hostContextSamples runs it that way - per DSPVector: the controllers' signals
// are made on the device (mlgpu_events, ctl_kernel), their 64 samples fetched (readContextSamples), the process function run
// again on the host with them (update(): its floats go into the kernel's constant table) and the captured kernel launched.
#include <cstddef>
#include <cstdio>

#define main mlgpu_example_controllers_main
#include "examples/audio-and-midi/controllers-to-audio.cpp"
#undef main

struct CtlExampleEvent
{
  int type, channel, sourceIdx, time;  // time in frames from the start of the run
  float value1, value2;
};

extern "C" int example_controllers_gpu_run(const CtlExampleEvent* events, int nEvents, int nVectors, float* out0, float* out1, char* err, size_t errLen)
{
  mlgpu_events* ev = nullptr;
  try
  {
    gpu::Engine eng(0);
    ExampleState state;
    state.sineGens.resize(state.sineControllers.size());
    AudioContext ctx(kInputChannels, kOutputChannels, kSampleRate);
    gpu::VoiceProgramOptions opt;
    opt.voicesPerContext = 1;
    opt.liveConstants = true;
    opt.hostContextSamples = true;
    gpu::VoiceProgram prog(eng, 1, &ctx, processAudio, &state, opt);
    std::vector<int> numbers = prog.contextInputs();  // the controllers the code reads, in the order it first asked for them
    eng.check(mlgpu_events_create(eng.handle(), 1, 1, &ev));
    eng.check(mlgpu_events_set_sample_rate(ev, (double)kSampleRate));
    eng.check(mlgpu_events_set_wanted_rows(ev, 0));  // no voice rows: only the controllers
    eng.check(mlgpu_events_watch_controllers(ev, numbers.data(), (int)numbers.size(), 1));
    std::vector<const float*> contextSignals;
    for (size_t c = 0; c < numbers.size(); ++c) contextSignals.push_back(mlgpu_events_controller_signal(ev, (int)c));
    float* noRows[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    gpu::DeviceSignal o0(eng, 1, 1, MLGPU_LAYOUT_VOICE_MAJOR), o1(eng, 1, 1, MLGPU_LAYOUT_VOICE_MAJOR);
    for (int v = 0; v < nVectors; ++v)
    {
      const int start = v * 64;
      for (int i = 0; i < nEvents; ++i)
        if (events[i].time >= start && events[i].time < start + 64)
        {
          // what the example's MIDI handler does: ctx.addInputEvent(MIDIMessageToEvent(m))
          Event e;
          e.type = events[i].type;
          e.channel = events[i].channel;
          e.sourceIdx = events[i].sourceIdx;
          e.time = events[i].time - start;
          e.value1 = events[i].value1;
          e.value2 = events[i].value2;
          ctx.addInputEvent(e);
        }
      for (const Event& e : ctx.pendingEvents_)
      {
        mlgpu_event m{e.type, e.channel, e.sourceIdx, e.time, e.value1, e.value2};
        eng.check(mlgpu_events_add_event(ev, 0, &m));
      }
      eng.check(mlgpu_events_process(ev, 1, 0, noRows, MLGPU_LAYOUT_QUAD));  // AudioContext::processVector: the controllers' next vector
      prog.readContextSamples(contextSignals.data());
      prog.update();  // processAudio on the host, with this vector's ctrlSig[0]
      prog.process({}, {&o0, &o1}, nullptr, contextSignals.data());
      eng.check(mlgpu_download(eng.handle(), out0 + start, o0.data(), o0.bytes()));
      eng.check(mlgpu_download(eng.handle(), out1 + start, o1.data(), o1.bytes()));
      eng.check(mlgpu_events_clear_events(ev));
      ctx.clearInputEvents();
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
