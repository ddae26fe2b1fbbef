// madronalib.h (MI355X drop-in): the reference's umbrella header (include/madronalib.h) pulls in mldsp.h plus the
// app layer; of the app layer only the process-function boundary (AudioContext inputs/outputs, SignalProcessFn)
// exists here — see mldsp.h in this directory.
#pragma once
#include "mldsp.h"

// ---- the device layer of the reference's console examples: names only ----------------------------------------------------
// source/app/MLMIDI.h, MLTimer.h, MLSharedResource.h sit on RtMidi and OS timers, which this engine does not have (SURVEY: OUT OF
// SCOPE). They exist here, like AudioTask, so that an example's main() compiles; a program feeds events through
// AudioContext::addInputEvent / gpu::SynthProgram::addInputEvent from whatever MIDI source it has.
#include <functional>
#include <memory>
#include <vector>
namespace ml
{
enum MIDIMessageType { kMIDINoteOff = 0, kMIDINoteOn = 1, kMIDIPolyPressure = 2, kMIDIControlChange = 3, kMIDIProgramChange = 4, kMIDIChannelPressure = 5, kMIDIPitchBend = 6 };
using MIDIMessage = std::vector<unsigned char>;
using MIDIMessageHandler = std::function<void(const MIDIMessage&)>;
class MIDIInput
{
 public:
  bool start(MIDIMessageHandler) { return false; }  // no MIDI device behind this engine
  void stop() {}
  std::string getAPIDisplayName() { return "none"; }
  std::string getPortName() { return "none"; }
};
// A channel-voice MIDI message as an Event (MLMIDI.cpp:134-194 is the reference's): status nibble -> event type, channel 1..16,
// data bytes as 0..1 values, the 14-bit pitch bend centred on 0.
inline Event MIDIMessageToEvent(const MIDIMessage& m)
{
  Event e{};
  if (m.empty()) return e;
  const int status = (m[0] >> 4) & 7, d1 = m.size() > 1 ? (m[1] & 0x7F) : 0, d2 = m.size() > 2 ? (m[2] & 0x7F) : 0;
  e.channel = (m[0] & 0x0F) + 1;
  const auto unit = [](int v) { return (float)v / 127.f; };
  switch (status)
  {
    case kMIDINoteOff: e.type = kNoteOff; e.sourceIdx = d1; e.value1 = unit(d2); break;
    case kMIDINoteOn: e.type = kNoteOn; e.sourceIdx = d1; e.value1 = unit(d2); break;
    case kMIDIPolyPressure: e.type = kNotePressure; e.sourceIdx = d1; e.value1 = unit(d2); break;
    case kMIDIControlChange: e.type = kController; e.sourceIdx = d1; e.value1 = unit(d2); break;
    case kMIDIProgramChange: e.type = kProgramChange; e.sourceIdx = d1; break;
    case kMIDIChannelPressure: e.type = kChannelPressure; e.value1 = unit(d1); break;
    case kMIDIPitchBend: e.type = kPitchBend; e.value1 = ((float)((d2 << 7) | d1) - 8192.f) / 8192.f; break;
    default: break;
  }
  return e;
}
class Timers
{
 public:
  void start(bool = false) {}
  void stop() {}
};
template <class T>
class SharedResourcePointer
{
  std::shared_ptr<T> p_;
  static std::shared_ptr<T> shared()
  {
    static std::weak_ptr<T> w;
    std::shared_ptr<T> s = w.lock();
    if (!s) w = s = std::make_shared<T>();
    return s;
  }

 public:
  SharedResourcePointer() : p_(shared()) {}
  T* operator->() const { return p_.get(); }
  T& operator*() const { return *p_; }
  T* get() const { return p_.get(); }
};
}  // namespace ml
