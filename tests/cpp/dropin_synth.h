// USER CODE #3 written against madronalib's public API only: a polyphonic synth as a Synth subclass (source/app/MLSynth.h) —
// per-voice DSP in processVoice(), driven by the 8 voice control rows of EventsToSignals (pitch, gate, ..., mod).
// Compiled unchanged against the reference (oracle/dropin_ref.cpp: its own Synth, AudioContext, EventsToSignals) and against
// include/mlgpu/compat (dropin_gpu.cpp: captured once, every voice of every instrument in one launch).
constexpr int kSynthVoices = 6;

class SmallSynth : public Synth
{
  struct VoiceDSP
  {
    SawGen saw;
    PulseGen pulse;
    Lopass lp;
    ADSR env;
    OnePole smooth;
  };
  std::array<VoiceDSP, kSynthVoices> dsp_;

 public:
  SmallSynth() : Synth(kSynthVoices)
  {
    for (auto& d : dsp_)
    {
      d.env.coeffs = ADSR::calcCoeffs(0.004f, 0.08f, 0.5f, 0.15f, 48000.f);
      d.smooth.coeffs = OnePole::makeCoeffs(0.05f);
      d.saw.clear();
    }
    // a scope for the UI: every 4th frame of each voice's signal and envelope (SignalProcessor::publishSignal)
    publishSignal("scope", kScopeFrames, kSynthVoices, 2, 2);
  }
  static constexpr int kScopeFrames = 256;

  // a host-side parameter change while notes are sounding (what a plug-in's parameter callback does)
  void setEnvelope(float a, float d, float s, float r)
  {
    for (auto& v : dsp_) v.env.coeffs = ADSR::calcCoeffs(a, d, s, r, 48000.f);
  }

  void processVoice(int v, const EventsToSignals::Voice& voice, const DSPVectorDynamic& inputs, DSPVectorDynamic& outputs,
                    AudioContext* ctx) override
  {
    VoiceDSP& d = dsp_[v];
    const DSPVector pitch = voice.outputs.constRow(kPitch);   // octaves re middle C (the example's event pitches)
    const DSPVector gate = voice.outputs.constRow(kGate);
    const DSPVector mod = voice.outputs.constRow(kMod);
    const DSPVector vox = voice.outputs.constRow(kVoice);
    const DSPVector freq = exp2Approx(pitch) * (261.6256f / 48000.f);
    const DSPVector osc = d.saw(freq) + d.pulse(freq * 0.5f, 0.3f + mod * 0.4f) * 0.6f;
    // brightness follows key pressure (z) and the mod wheel; per-voice stereo position from the voice index row
    const DSPVector cutoff = clamp(freq * (2.f + 6.f * d.smooth(voice.outputs.constRow(kZ) + mod)), DSPVector(0.001f), DSPVector(0.45f));
    const DSPVector env = d.env(gate);
    const DSPVector y = d.lp(osc, cutoff, DSPVector(0.5f)) * env;
    storePublishedSignal("scope", concatRows(y, env), kFloatsPerDSPVector, v);
    const DSPVector pan = vox * (1.f / kSynthVoices);
    outputs[0] += y * (1.f - pan);
    outputs[1] += y * pan;
  }
};

// USER CODE #3b: a Synth whose voices read nothing but pitch and gate (most do). On the GPU side such a voice kernel can compute
// the two rows itself from the events' records (gpu::VoiceProgramOptions::eventRowsInKernel): same source, same bits either way.
class LeanSynth : public Synth
{
  struct VoiceDSP
  {
    SawGen saw;
    Lopass lp;
    ADSR env;
  };
  std::array<VoiceDSP, kSynthVoices> dsp_;

 public:
  LeanSynth() : Synth(kSynthVoices)
  {
    for (auto& d : dsp_)
    {
      d.env.coeffs = ADSR::calcCoeffs(0.004f, 0.08f, 0.5f, 0.15f, 48000.f);
      d.lp.coeffs = Lopass::makeCoeffs(0.12f, 0.8f);
      d.saw.clear();
    }
  }
  void setEnvelope(float a, float d, float s, float r)
  {
    for (auto& v : dsp_) v.env.coeffs = ADSR::calcCoeffs(a, d, s, r, 48000.f);
  }
  void processVoice(int v, const EventsToSignals::Voice& voice, const DSPVectorDynamic& inputs, DSPVectorDynamic& outputs,
                    AudioContext* ctx) override
  {
    VoiceDSP& d = dsp_[v];
    const DSPVector freq = exp2Approx(voice.outputs.constRow(kPitch)) * (261.6256f / 48000.f);
    const DSPVector y = d.lp(d.saw(freq)) * d.env(voice.outputs.constRow(kGate));
    outputs[0] += y;
    outputs[1] += y * 0.5f;
  }
};


// USER CODE #3c: a Synth whose voices also read the instrument's smoothed MIDI controllers through the AudioContext
// (ctx->getInputController(n), MLAudioContext.h:91): brightness (74) opens the filter, the mod wheel (1) bends the pitch, channel
// pressure (controller slot 128) swells the level. On the GPU side each controller is one signal per instrument made from the
// controller events (mlgpu_events_watch_controllers) and read by the instrument's voices.
class ControllerSynth : public Synth
{
  struct VoiceDSP
  {
    SawGen saw;
    Lopass lp;
    ADSR env;
  };
  std::array<VoiceDSP, kSynthVoices> dsp_;

 public:
  ControllerSynth() : Synth(kSynthVoices)
  {
    for (auto& d : dsp_)
    {
      d.env.coeffs = ADSR::calcCoeffs(0.004f, 0.08f, 0.5f, 0.15f, 48000.f);
      d.saw.clear();
    }
  }
  void setEnvelope(float a, float d, float s, float r)
  {
    for (auto& v : dsp_) v.env.coeffs = ADSR::calcCoeffs(a, d, s, r, 48000.f);
  }
  void processVoice(int v, const EventsToSignals::Voice& voice, const DSPVectorDynamic& inputs, DSPVectorDynamic& outputs,
                    AudioContext* ctx) override
  {
    VoiceDSP& d = dsp_[v];
    const DSPVector brightness = ctx->getInputController(74);
    const DSPVector wheel = ctx->getInputController(1);
    const DSPVector pressure = ctx->getInputController(128);
    const DSPVector freq = exp2Approx(voice.outputs.constRow(kPitch) + wheel * 0.05f) * (261.6256f / 48000.f);
    const DSPVector omega = brightness * 0.2f + 0.02f;
    const DSPVector y = d.lp(d.saw(freq), omega, DSPVector(0.7f)) * d.env(voice.outputs.constRow(kGate)) * (pressure * 0.5f + 0.5f);
    outputs[0] += y;
    outputs[1] += y * wheel;
  }
};

// What the host application reports before each block (AudioContext::updateTime) in the Synth runs of both sides: stopped at
// first, started at block 1, a tempo change, a stop, a restart somewhere else.
struct HostTransport
{
  double ppq{0.75}, bpm{126.};
  bool playing{false};
  void beforeBlock(int b)
  {
    if (b == 1) playing = true;
    if (b == 4) bpm = 97.5;
    if (b == 6) playing = false;
    if (b == 7)
    {
      playing = true;
      ppq = 3.5;
    }
  }
  void afterBlock(int blockFrames)
  {
    if (playing) ppq += blockFrames * bpm / 60. / 48000.;
  }
};

// USER CODE #3d: a Synth with a tempo-synced tremolo: every voice locks an LFO phasor at twice the host's quarter-note phase
// (ctx->getBeatPhase() into a TempoLock, MLAudioContext.h:82, MLDSPFilters.h:1478). On the GPU side the beat phase is one device
// signal per instrument (mlgpu_transport) read by that instrument's voices.
class TempoSynth : public Synth
{
  struct VoiceDSP
  {
    SawGen saw;
    ADSR env;
    TempoLock lock;
  };
  std::array<VoiceDSP, kSynthVoices> dsp_;

 public:
  TempoSynth() : Synth(kSynthVoices)
  {
    for (auto& d : dsp_)
    {
      d.env.coeffs = ADSR::calcCoeffs(0.004f, 0.08f, 0.5f, 0.15f, 48000.f);
      d.saw.clear();
    }
  }
  void setEnvelope(float a, float d, float s, float r)
  {
    for (auto& v : dsp_) v.env.coeffs = ADSR::calcCoeffs(a, d, s, r, 48000.f);
  }
  void processVoice(int v, const EventsToSignals::Voice& voice, const DSPVectorDynamic& inputs, DSPVectorDynamic& outputs,
                    AudioContext* ctx) override
  {
    VoiceDSP& d = dsp_[v];
    const DSPVector beat = ctx->getBeatPhase();
    const DSPVector lfo = d.lock(beat, 2.f, 1.f / 48000.f);
    const DSPVector freq = exp2Approx(voice.outputs.constRow(kPitch)) * (261.6256f / 48000.f);
    const DSPVector y = d.saw(freq) * d.env(voice.outputs.constRow(kGate)) * (DSPVector(1.f) - lfo * 0.5f);
    outputs[0] += y;
    outputs[1] += y * beat;
  }
};

// USER CODE #3e: what a plug-in wrapper runs - a Synth behind SignalProcessBuffer::process with whatever block sizes the host
// asks for (MLSignalProcessBuffer.cpp:36-90): note and controller events with block-relative times, the host's time report
// before every block, ctx->processVector(offset) per 64 frames, clearInputEvents() per block. The voice reads all three kinds of
// context: its voice rows, a controller, the beat phase.
class PluginSynth : public Synth
{
  struct VoiceDSP
  {
    SawGen saw;
    Lopass lp;
    ADSR env;
    TempoLock lock;
  };
  std::array<VoiceDSP, kSynthVoices> dsp_;

 public:
  PluginSynth() : Synth(kSynthVoices)
  {
    for (auto& d : dsp_)
    {
      d.env.coeffs = ADSR::calcCoeffs(0.004f, 0.08f, 0.5f, 0.15f, 48000.f);
      d.lp.coeffs = Lopass::makeCoeffs(0.1f, 0.9f);
      d.saw.clear();
    }
  }
  void processVoice(int v, const EventsToSignals::Voice& voice, const DSPVectorDynamic& inputs, DSPVectorDynamic& outputs,
                    AudioContext* ctx) override
  {
    VoiceDSP& d = dsp_[v];
    const DSPVector lfo = d.lock(ctx->getBeatPhase(), 4.f, 1.f / 48000.f);
    const DSPVector level = ctx->getInputController(74) * 0.75f + 0.25f;
    const DSPVector freq = exp2Approx(voice.outputs.constRow(kPitch)) * (261.6256f / 48000.f);
    const DSPVector y = d.lp(d.saw(freq)) * d.env(voice.outputs.constRow(kGate)) * level;
    outputs[0] += y * (DSPVector(1.f) - lfo * 0.5f);
    outputs[1] += y * lfo;
  }
};
