// oracle/example_ref_controllers.cpp — TEST INFRASTRUCTURE. The reference's own example program
// examples/audio-and-midi/controllers-to-audio.cpp, included UNCHANGED from /root/reference and compiled against the
// reference's headers: eight sine oscillators whose frequencies the process function computes ON THE HOST, once per DSPVector,
// from sample 0 of eight smoothed MIDI controller signals (`ctrlToFreq(ctrlSig[0])`). Run the way AudioTask runs it: one
// AudioContext, processVector() then the process function per 64 frames. main() is renamed out of the way; the MIDI input
// and timer classes it names need RtMidi / an OS timer and are stubbed (never run); AudioTask's stubs are in
// example_ref_reverb.cpp.
#include <cstddef>
#include <cstring>
#include <memory>

#define main mlref_example_controllers_main
#include "examples/audio-and-midi/controllers-to-audio.cpp"
#undef main

namespace ml
{
struct MIDIInput::Impl
{
};
MIDIInput::MIDIInput() {}
MIDIInput::~MIDIInput() {}
bool MIDIInput::start(MIDIMessageHandler) { return false; }
void MIDIInput::stop() {}
Event MIDIMessageToEvent(const MIDIMessage&) { return Event(); }  // MLMIDI.cpp sits on RtMidi; the test feeds Events directly
const int Timers::kMillisecondsResolution = 10;
void Timers::start(bool) {}
void Timers::stop() {}
}  // namespace ml

struct CtlExampleEvent
{
  int type, channel, sourceIdx, time;  // time in frames from the start of the run
  float value1, value2;
};

extern "C" int example_controllers_ref_run(const CtlExampleEvent* events, int nEvents, int nVectors, float* out0, float* out1)
{
  ExampleState state;
  state.sineGens.resize(state.sineControllers.size());
  AudioContext ctx(kInputChannels, kOutputChannels, kSampleRate);
  for (int v = 0; v < nVectors; ++v)
  {
    const int start = v * (int)kFloatsPerDSPVector;
    for (int i = 0; i < nEvents; ++i)
      if (events[i].time >= start && events[i].time < start + (int)kFloatsPerDSPVector)
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
    ctx.processVector(0);
    processAudio(&ctx, &state);
    store(ctx.outputs[0], out0 + start);
    store(ctx.outputs[1], out1 + start);
    ctx.clearInputEvents();
  }
  return 0;
}
