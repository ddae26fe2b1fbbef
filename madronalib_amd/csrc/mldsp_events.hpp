// mldsp_events.hpp — the device side of EventsToSignals (source/app/MLEventsToSignals.{h,cpp}) shared by the events kernel
// (events.hip: all 8 rows into HBM) and the run-time fused graph kernels (graph.hip: pitch and gate as source nodes of a voice
// graph, never written to memory): the record format the host router produces, the per-voice state layout, LinearGlide with
// its 64 slots, note_frame - one frame of a vector that holds note records - and CtlVoice: the control records of
// e2s_ctl_kernel expanded to the pitch and gate rows inside a voice kernel, one quad of frames at a time.
#pragma once
#include "mlgpu_device_args.hpp"
#include "mldsp_math.hpp"

namespace mlev
{
using namespace mldev;

// ---- records ------------------------------------------------------------------------------------------------------
enum RecType : uint32_t
{
  REC_AWAKE = 0,      // the instrument received its first event: processVector stops being a no-op (:383-386)
  REC_NOTE_ON = 1,    // writeNoteEvent kNoteOn (:129-152):   v1 pitch, v2 velocity, flags bit0 doGlide bit1 doReset
  REC_NOTE_RETRIG = 2,
  REC_NOTE_OFF = 3,
  REC_SET_BEND = 4,   // currentPitchBend = v1 (:700-731)
  REC_SET_MOD = 5,
  REC_SET_X = 6,
  REC_SET_Y = 7,
  REC_SET_Z = 8,
  REC_SET_CHANNEL_PRESSURE = 9  // controllers[128].inputValue (MIDI mode, :620-626)
};
struct Rec
{
  uint32_t vec;    // DSPVector index inside this launch
  uint32_t typeTimeFlags;  // type | time << 8 | flags << 16
  float v1, v2;
};

// ---- device state layout (uint32 words per voice, SoA [word][lanes]) --------------------------------------------------
enum : int
{
  S_AWAKE = 0, S_VELOCITY, S_PITCH, S_BEND, S_MOD, S_X, S_Y, S_Z, S_CHANPRESS, S_AGE, S_AGE_STEP, S_INHIBIT_GLIDE,
  S_PG_CURR, S_PG_STEP, S_PG_TARGET, S_PG_REMAINING, S_PG_PER_GLIDE, S_PG_DY,
  S_DRIFT_SEED, S_DRIFT_COUNTER, S_DRIFT_VALUE, S_DRIFT_NEXT,
  S_RECALC,  // Voice::recalcNeeded (:45-54): set by setSampleRate / setPitchGlideInSeconds, consumed by the next beginProcess
  S_GLIDES  // 7 glides follow: bend, mod, x, y, z, drift, channel pressure
};
constexpr int kNumGlides = 7;
constexpr int kGlideWords = 5 + 64;  // target, step, remaining, isUniform, uniformValue, currVec[64]
constexpr int kStateWords = S_GLIDES + kNumGlides * kGlideWords;


struct E2SArgs
{
  uint32_t* state;            // [kStateWords][lanes]
  const Rec* recs;            // all records of this launch, grouped by lane, time-ordered inside a lane
  uint2* recRange;            // [lanes]: a lane's records of this launch are recs[x .. y); a kernel that read them writes {0, 0} back
  SignalView out[8];          // pitch, gate, vox, z, x, y, mod, elapsed time: V = instruments * polyphony voices
  size_t lanes, T;
  int group, polyphony, slotBase;  // lane = instrument * group + (voice slot - slotBase)
  uint32_t rowMask;                // rows that are computed (mlgpu_events_set_wanted_rows); bit r = row r of `out`
  uint32_t flags;                  // MLGPU_KFLAG_*
  int blockPath;                   // 0: every vector on its own (MLGPU_E2S_NO_BLOCKS in the environment, for A / B measurements)
  E2SSettings s;
};

// LinearGlide (MLDSPGens.h:433-515) with one shortcut that does not change results: between glides mCurrVec is a
// broadcast of one value, kept in a register instead of 64 words of HBM. `st` is this glide's first word for this lane
// (stride = lanes); it is passed in instead of stored to keep the register count of seven glides down.
struct Glide
{
  float target, step, uniformValue, startValue;
  int32_t remaining;
  int modeFlags;  // bits 0-1: mode (0 hold, 1 end, 2 start, 3 continue); bit 2: mCurrVec is uniform
  MLD bool isUniform() const { return (modeFlags & 4) != 0; }
  MLD int mode() const { return modeFlags & 3; }
  MLD void load(const uint32_t* st, size_t stride)
  {
    target = u2f(st[0]);
    step = u2f(st[stride]);
    remaining = (int32_t)st[2 * stride];
    modeFlags = st[3 * stride] ? 4 : 0;
    uniformValue = u2f(st[4 * stride]);
    startValue = 0.f;
  }
  MLD void store(uint32_t* st, size_t stride) const
  {
    st[0] = f2u(target);
    st[stride] = f2u(step);
    st[2 * stride] = (uint32_t)remaining;
    st[3 * stride] = isUniform() ? 1u : 0u;
    st[4 * stride] = f2u(uniformValue);
  }
  // beginVector for a caller that knows what mCurrVec[63] holds (it has the slots in registers): no memory access
  MLD void beginVectorKnown(float f, int32_t perGlide, float dyPerVector, float slot63)
  {
    if (f != target)
    {
      target = f;
      remaining = perGlide;
    }
    int m;
    if (remaining < 0) m = 0;
    else if (remaining == 0)
    {
      m = 1;
      step = 0.f;
      remaining--;
    }
    else if (remaining == perGlide)
    {
      m = 2;
      startValue = isUniform() ? uniformValue : slot63;
      step = (target - startValue) * dyPerVector;
      remaining--;
    }
    else
    {
      m = 3;
      remaining--;
    }
    modeFlags = (modeFlags & 4) | m;
  }
  MLD void beginVector(const uint32_t* st, size_t stride, float f, int32_t perGlide, float dyPerVector)
  {
    if (f != target)
    {
      target = f;
      remaining = perGlide;
    }
    int m;
    if (remaining < 0) m = 0;
    else if (remaining == 0)
    {
      m = 1;
      step = 0.f;
      remaining--;
    }
    else if (remaining == perGlide)
    {
      m = 2;
      startValue = isUniform() ? uniformValue : u2f(st[(size_t)(5 + 63) * stride]);
      step = (target - startValue) * dyPerVector;
      remaining--;
    }
    else
    {
      m = 3;
      remaining--;
    }
    modeFlags = (modeFlags & 4) | m;
  }
  // mCurrVec[n] is read and rewritten at sample n only, so a quad's four slots can be fetched together (and a quad ahead):
  // a load per sample in the middle of the load -> add -> store chain made the whole kernel wait out a memory round trip
  // per sample (62 us per DSPVector per wavefront).
  MLD bool readsCurrVec() const { return !isUniform() && (mode() == 0 || mode() == 3); }
  MLD void preload(const uint32_t* st, size_t stride, int q, float cur[4]) const
  {
    if (readsCurrVec())
    {
#pragma unroll
      for (int k = 0; k < 4; ++k) cur[k] = u2f(st[(size_t)(5 + 4 * q + k) * stride]);
    }
  }
  MLD float next(uint32_t* st, size_t stride, int n) const  // one sample at a time (the record-walking path)
  {
    return nextWith(st, stride, n, readsCurrVec() ? u2f(st[(size_t)(5 + n) * stride]) : 0.f);
  }
  MLD float nextWith(uint32_t* st, size_t stride, int n, float cur) const  // cur: what preload fetched for slot n
  {
    const int m = mode();
    if (m == 0) return isUniform() ? uniformValue : cur;
    if (m == 1) return target;
    float c;
    if (m == 2) c = startValue + ((float)(n + 1) * 0.015625f) * step;
    else c = (isUniform() ? uniformValue : cur) + step;
    st[(size_t)(5 + n) * stride] = f2u(c);
    return c;
  }
  MLD void endVector()
  {
    const int m = mode();
    if (m == 1)
    {
      modeFlags = 4;
      uniformValue = target;
    }
    else if (m >= 2)
      modeFlags = 0;
    else
      modeFlags &= 4;
  }
};


// One frame of a vector that holds note records: writeNoteEvent (:115-216) and its neighbours walked frame by frame - the note
// records that end on frame n are applied, the gate, the sample-accurate pitch glide and the event age take their step. Shared by
// e2s_kernel (all rows) and EventsVoice (pitch and gate inside a voice graph). The caller's state comes in by reference;
// setPitchGlideTime(samples) and pitchGlideNext(pitch) are its two glide operations; vTime is written when wantTime.
// The frames of a vector look at the same pending record again and again (a note that starts at frame 40 is read by frames 0..40):
// one record kept in registers, fetched again only when another index is asked for. Without it every frame is a memory round trip
// behind the stores of the frame before (loads and stores of a wavefront complete in issue order).
struct RecCache
{
  uint32_t idx{0xFFFFFFFFu};
  Rec r;
  MLD const Rec& at(const Rec* recs, uint32_t i)
  {
    if (i != idx)
    {
      r = recs[i];
      idx = i;
    }
    return r;
  }
};

template <class SetGlideTime, class GlideNext>
MLD void note_frame(const Rec* recs, RecCache& cache, uint32_t& nc, uint32_t vend, int n, bool& preApplied, float& velocity, float& pitch, uint32_t& age, uint32_t& ageStep,
                    bool& inhibit, int32_t pitchGlideSamples, bool wantTime, double srD, SetGlideTime setPitchGlideTime, GlideNext pitchGlideNext,
                    float& vPitch, float& vGate, float& vTime)
{
  bool retrigFrame = false;
  while (nc < vend)
  {
    const Rec rc = cache.at(recs, nc);
    const uint32_t type = rc.typeTimeFlags & 0xFF;
    if (type != REC_NOTE_ON && type != REC_NOTE_RETRIG && type != REC_NOTE_OFF)
    {
      ++nc;
      continue;
    }
    int dest = (int)((rc.typeTimeFlags >> 8) & 0xFF);
    const uint32_t flags = rc.typeTimeFlags >> 16;
    if (!preApplied)
    {
      if (type != REC_NOTE_OFF)
      {
        if (flags & 2) age = 0;  // doReset
        ageStep = 1;
      }
      if (type == REC_NOTE_ON)
      {
        inhibit = !(flags & 1);
        setPitchGlideTime((flags & 1) ? pitchGlideSamples : 0);
      }
      preApplied = true;
    }
    if (type == REC_NOTE_RETRIG)
    {
      if (dest == 0) dest = 1;                 // make room for the retrigger frame, :163-167
      if (n == dest - 1) retrigFrame = true;   // gate 0 for one frame, :171-175
    }
    if (dest == n)
    {
      if (type == REC_NOTE_OFF) velocity = 0.f;
      else
      {
        pitch = rc.v1;
        velocity = rc.v2;
      }
      ++nc;
      preApplied = false;
      continue;
    }
    break;
  }
  vGate = retrigFrame ? 0.f : velocity;
  vPitch = pitchGlideNext(pitch);
  age += ageStep;
  if (wantTime) vTime = (float)((double)age / srD);
  // A retrigger that lands on the frame where the previous note event of this voice ended (a note-on and a steal of
  // the same voice on one frame) makes the reference REWRITE frame dest - 1, which that previous event had already
  // written (:163-175): the glide is stepped and the event age counted once more, with the previous event's new
  // pitch. Look ahead for exactly that pattern and redo this frame the same way.
  while (nc < vend)
  {
    uint32_t pi = nc;
    while (pi < vend && ((cache.at(recs, pi).typeTimeFlags & 0xFF) < REC_NOTE_ON || (cache.at(recs, pi).typeTimeFlags & 0xFF) > REC_NOTE_OFF)) ++pi;
    if (pi >= vend) break;
    const Rec P = cache.at(recs, pi);
    const uint32_t ptype = P.typeTimeFlags & 0xFF;
    int pdest = (int)((P.typeTimeFlags >> 8) & 0xFF);
    if (ptype == REC_NOTE_RETRIG && pdest == 0) pdest = 1;
    if (pdest != n + 1) break;
    uint32_t ri = pi + 1;
    while (ri < vend && ((recs[ri].typeTimeFlags & 0xFF) < REC_NOTE_ON || (recs[ri].typeTimeFlags & 0xFF) > REC_NOTE_OFF)) ++ri;
    if (ri >= vend) break;
    const Rec R = recs[ri];
    int rdest = (int)((R.typeTimeFlags >> 8) & 0xFF);
    // (a retrigger on frame 0 makes room on frame 1 like every other, :163-167 - two steals of one voice on a vector's first frame
    // are this pattern too; without this line the second one's reset of the event age came a frame late: tools/events_soak.py)
    if ((R.typeTimeFlags & 0xFF) == REC_NOTE_RETRIG && rdest == 0) rdest = 1;
    if ((R.typeTimeFlags & 0xFF) != REC_NOTE_RETRIG || rdest != n + 1) break;
    if (!preApplied)  // P's own bookkeeping, if this frame is the first one it sees
    {
      const uint32_t pflags = P.typeTimeFlags >> 16;
      if (ptype != REC_NOTE_OFF)
      {
        if (pflags & 2) age = 0;
        ageStep = 1;
      }
      if (ptype == REC_NOTE_ON)
      {
        inhibit = !(pflags & 1);
        setPitchGlideTime((pflags & 1) ? pitchGlideSamples : 0);
      }
    }
    if (ptype == REC_NOTE_OFF) velocity = 0.f;  // P's new values
    else
    {
      pitch = P.v1;
      velocity = P.v2;
    }
    nc = ri;                                    // R is the current note record now, its bookkeeping done here
    if ((R.typeTimeFlags >> 16) & 2) age = 0;
    ageStep = 1;
    preApplied = true;
    vGate = 0.f;                                // the retrigger frame
    vPitch = pitchGlideNext(pitch);
    age += ageStep;
    if (wantTime) vTime = (float)((double)age / srD);
  }
}

// flags of a note record (typeTimeFlags >> 16): bit 0 doGlide, bit 1 doReset, bit 2 REC_FLAG_REWIND
constexpr uint32_t REC_FLAG_REWIND = 4u;

// A vector that holds a note record flagged REC_FLAG_REWIND: a record on frame 0 for a voice whose earlier note records of the same
// vector ended later than that. Events reach the reference sorted by time, so only an event it makes up itself can be such a record: the
// note-off of a sustain-pedal release, built with Event's default time 0 (MLEventsToSignals.cpp:833-836). writeNoteEvent then sets
// nextFrameToProcess BACK to 0 (writeOutputFrames ends with `nextFrameToProcess = endFrame`, :141), and every frame the earlier records
// had written is written AGAIN by what follows - the pitch glide and the event age stepping on from where the first pass left them.
// note_rewind is that first pass, record by record as the reference walks them (values discarded, state kept):
//   note-on / note-off on frame d: the frames [next, d) step, the record's values apply, next = d           (:146-166, :185-193)
//   retrigger on frame d (>= 1): the frames [next, d - 1) step, then frame d - 1 - always, also when next == d - steps, next = d  (:169-183)
// up to the LAST flagged record of the vector; nc is left there and the caller walks the vector from frame 0 as usual.
// Found by tools/events_soak.py (8 of 2 100 random configurations: a note, then the pedal's release and the note's end in one DSPVector).
template <class SetGlideTime, class GlideNext>
MLD void note_rewind(const Rec* recs, uint32_t& nc, uint32_t vend, float& velocity, float& pitch, uint32_t& age, uint32_t& ageStep, bool& inhibit,
                     int32_t pitchGlideSamples, SetGlideTime setPitchGlideTime, GlideNext pitchGlideNext)
{
  uint32_t last = vend;
  for (uint32_t j = nc; j < vend; ++j)
  {
    const uint32_t w = recs[j].typeTimeFlags;
    if ((w & 0xFF) >= REC_NOTE_ON && (w & 0xFF) <= REC_NOTE_OFF && ((w >> 16) & REC_FLAG_REWIND)) last = j;
  }
  if (last == vend) return;
  int next = 0;
  for (; nc < last; ++nc)
  {
    const Rec rc = recs[nc];
    const uint32_t type = rc.typeTimeFlags & 0xFF;
    if (type != REC_NOTE_ON && type != REC_NOTE_RETRIG && type != REC_NOTE_OFF) continue;
    int dest = (int)((rc.typeTimeFlags >> 8) & 0xFF);
    const uint32_t flags = rc.typeTimeFlags >> 16;
    if (type != REC_NOTE_OFF)
    {
      if (flags & 2) age = 0;  // doReset
      ageStep = 1;
    }
    if (type == REC_NOTE_ON)
    {
      inhibit = !(flags & 1);
      setPitchGlideTime((flags & 1) ? pitchGlideSamples : 0);
    }
    int steps;
    if (type == REC_NOTE_RETRIG)
    {
      if (dest == 0) dest = 1;
      steps = (dest - 1 > next ? dest - 1 - next : 0) + 1;
    }
    else
      steps = dest > next ? dest - next : 0;
    for (int i = 0; i < steps; ++i)
    {
      (void)pitchGlideNext(pitch);
      age += ageStep;
    }
    if (type == REC_NOTE_OFF) velocity = 0.f;
    else
    {
      pitch = rc.v1;
      velocity = rc.v2;
    }
    next = dest;
  }
}

// ---- EventsToSignals inside a fused voice graph ------------------------------------------------------------------------------
// The pitch and gate rows of one voice as source nodes of a graph kernel, never written to memory as signals. Round 5 splits the
// work by RATE (the round-4 form walked records, three glides and the drift random walk inside the voice kernel: 0.35 scalar and
// branch instructions per vector instruction, 134 spilled registers):
//   * e2s_ctl_kernel (events.hip) runs everything that happens once per DSPVector or once per note event - the record walk, the
//     drift random walk, the bend glide, the sample-accurate pitch glide, Voice::beginProcess / writeNoteEvent / endProcess
//     (:75-262) - and emits ONE CONTROL RECORD of kCtlRecWords words per voice and DSPVector: the pitch before drift, the held gate,
//     the drift glide's input of this vector, flags. A vector in which the pitch or the gate moves inside the vector (a note event,
//     a portamento in progress, a moving bend: a few per cent of all vectors) is flagged CF_ROWS and its 64 frames of pitch-before-
//     drift and gate are written to two side signals, which only the flagged lanes read back.
//   * the voice kernel (CtlVoice below) expands a record to audio rate. The one per-sample state machine left to it is the drift
//     LinearGlide (8 s per glide, a new target every 8-16 s: moving most of the time in most wavefronts, its mCurrVec slots
//     read and rewritten every vector - 8 B per voice-sample, what reading the two rows used to cost): pitch[n] = P[n] +
//     (drift[n] * driftAmount) * kDriftScale, the reference's own operation order (:244, :247).
// Same operations on the same values as e2s_kernel: a launch of either form leaves the state words the other expects.
// MIDI protocol only: one lane per playing voice, lane == voice index.
enum : int
{
  C_PITCH = 0,   // float: the pitch glide at rest + the held bend term (vPitch after :244), the same for all 64 frames
  C_GATE,        // float: the held velocity
  C_DRIFT_IN,    // float: currentDriftValue, the drift glide's input for this vector (:241)
  C_FLAGS,       // CF_*
  kCtlRecWords
};
constexpr uint32_t CF_ON = 1u;    // the voice's instrument is awake: processVector is not a no-op (:383-386)
constexpr uint32_t CF_ROWS = 2u;  // the pitch before drift of this vector is in the side signal, frame by frame
constexpr uint32_t CF_GATE_ROW = 4u;  // ... and so is the gate (a vector with a note event)

// Memory of this object is addressed the CDNA way: a 128-bit buffer descriptor in scalar registers (built from kernel arguments
// only, so provably wave-uniform), ONE 32-bit per-lane byte offset in a vector register, and a scalar offset per access
// (buffer_load_dword v, v_off, s[rsrc], s_off offen). With plain pointers the compiler keeps a 64-bit address per lane and access
// stream in vector registers - 59 such loads, 46 spilled registers in the instrument bank's voice kernel - and those are what a
// generated voice kernel is short of (128 per lane at four wavefronts per SIMD).
typedef __amdgpu_buffer_rsrc_t BufRsrc;
MLD BufRsrc lane_buffer(const void* uniformBase, size_t bytes)
{
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(uniformBase), (short)0, (int)(bytes > 0xFFFFFFFFull ? 0xFFFFFFFFull : bytes), 0x00020000);
}

struct CtlVoice
{
  typedef float f32x4e __attribute__((ext_vector_type(4)));
  // Per lane this object keeps the lane's byte offset, the drift glide's few words, the vector's pitch and gate and ONE set of four
  // words that holds, in turn, the coming quad's mCurrVec slots and - during a vector's last quad - the next vector's control record.
  EventsDev d;    // (uniform: scalar registers / kernel arguments)
  size_t T;
  BufRsrc glide;  // the drift glide's words [kGlideWords][lanes]
  uint32_t rowBytes;   // 4 * lanes: one word of every lane
  uint32_t laneBytes;  // 4 * lane
  Glide gd;
  float P, gate;
  bool on, rows, gateRow, moving, fetch, ramp;
  bool anyRows, anyRamp, anyOff, anyFetch;  // wave-uniform
  float nd[4];

  MLD uint32_t glideWord(int w) const { return __builtin_amdgcn_raw_buffer_load_b32(glide, (int)laneBytes, (int)((uint32_t)w * rowBytes), 0); }
  MLD void setGlideWord(int w, uint32_t x) const { __builtin_amdgcn_raw_buffer_store_b32(x, glide, (int)laneBytes, (int)((uint32_t)w * rowBytes), 0); }
  MLD void fetchRecord(size_t t)  // into nd[]: C_PITCH, C_GATE, C_DRIFT_IN, C_FLAGS
  {
    const BufRsrc rec = lane_buffer(d.ctl + t * (size_t)kCtlRecWords * d.lanes, (size_t)kCtlRecWords * rowBytes);
#pragma unroll
    for (int k = 0; k < 4; ++k) nd[k] = u2f(__builtin_amdgcn_raw_buffer_load_b32(rec, (int)laneBytes, (int)((uint32_t)k * rowBytes), 0));
  }
  MLD void load(const EventsDev& a, size_t voice, size_t nVectors)
  {
    d = a;
    T = nVectors;
    rowBytes = (uint32_t)d.lanes * 4u;
    laneBytes = (uint32_t)voice * 4u;
    glide = lane_buffer(d.state + (size_t)(S_GLIDES + 5 * kGlideWords) * d.lanes, (size_t)kGlideWords * rowBytes);  // glide 5: drift
    gd.target = u2f(glideWord(0));
    gd.step = u2f(glideWord(1));
    gd.remaining = (int32_t)glideWord(2);
    gd.modeFlags = glideWord(3) ? 4 : 0;
    gd.uniformValue = u2f(glideWord(4));
    gd.startValue = 0.f;
    fetchRecord(0);
  }
  MLD void store() const
  {
    setGlideWord(0, f2u(gd.target));
    setGlideWord(1, f2u(gd.step));
    setGlideWord(2, (uint32_t)gd.remaining);
    setGlideWord(3, gd.isUniform() ? 1u : 0u);
    setGlideWord(4, f2u(gd.uniformValue));
  }

  MLD void begin_vector(size_t t)
  {
    (void)t;
    const uint32_t f = f2u(nd[C_FLAGS]);
    const float din = nd[C_DRIFT_IN];
    on = (f & CF_ON) != 0;
    rows = (f & CF_ROWS) != 0;
    gateRow = (f & CF_GATE_ROW) != 0;
    P = on ? nd[C_PITCH] : 0.f;
    gate = on ? nd[C_GATE] : 0.f;
    // a glide that starts from a non-uniform mCurrVec reads its slot 63 - this lane's own store of the vector before (once per
    // 8-16 s and voice: asked wave-uniformly)
    float slot63 = 0.f;
    const bool starts = on && ((din != gd.target) || gd.remaining == d.s.driftGlideVectors) && !gd.isUniform();
    if (__builtin_amdgcn_ballot_w64(starts) != 0)
    {
      if (starts) slot63 = u2f(glideWord(5 + 63));
    }
    if (on) gd.beginVectorKnown(din, d.s.driftGlideVectors, d.s.driftGlideDy, slot63);
    const int m = gd.mode();  // (a lane that is not on keeps mode 0 from the end of its last vector)
    moving = on && m >= 2;
    ramp = on && m == 2;
    fetch = on && gd.readsCurrVec();
    const float hv = (m == 1) ? gd.target : gd.uniformValue;
    anyRows = __builtin_amdgcn_ballot_w64(rows) != 0;
    anyRamp = __builtin_amdgcn_ballot_w64(ramp) != 0;
    anyOff = __builtin_amdgcn_ballot_w64(!on) != 0;
    anyFetch = __builtin_amdgcn_ballot_w64(fetch) != 0;
    nd[0] = nd[1] = nd[2] = nd[3] = hv;
    if (anyFetch)
    {
      if (fetch)
      {
#pragma unroll
        for (int k = 0; k < 4; ++k) nd[k] = u2f(glideWord(5 + k));
      }
    }
  }

  MLD void quad(size_t t, int q, f32x4e& oPitch, f32x4e& oGate)
  {
    const float prev[4] = {nd[0], nd[1], nd[2], nd[3]};
    if (q < 15)
    {
      if (anyFetch)
      {
        if (fetch)
        {
#pragma unroll
          for (int k = 0; k < 4; ++k) nd[k] = u2f(glideWord(5 + 4 * q + 4 + k));
        }
      }
    }
    else if (t + 1 < T)
      fetchRecord(t + 1);
    f32x4e Pq = {P, P, P, P};
    oGate = f32x4e{gate, gate, gate, gate};
    if (anyRows)
    {
      if (rows)
      {
        const size_t at = (t * 16 + (size_t)q) * d.lanes + (size_t)(laneBytes >> 2);
        Pq = __builtin_nontemporal_load((const f32x4e*)d.rowP + at);
        if (gateRow) oGate = __builtin_nontemporal_load((const f32x4e*)d.rowG + at);
      }
    }
    // (the rare questions are asked once per quad, not once per sample: a wave-uniform branch is a scalar compare, a branch and a
    // possible fetch stall, and scalar issue is not hidden on this chip)
    float c[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) c[k] = prev[k] + gd.step;  // a continuing glide: mCurrVec[n] += step (LinearGlide, MLDSPGens.h:497-505)
    if (anyRamp)  // a glide's first vector ramps from its start value (:481-495); once per 8-16 s and voice
    {
#pragma unroll
      for (int k = 0; k < 4; ++k)
      {
        const float r = gd.startValue + ((float)(4 * q + k + 1) * 0.015625f) * gd.step;
        c[k] = ramp ? r : c[k];
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
    {
      const float driftSig = moving ? c[k] : prev[k];
      oPitch[k] = Pq[k] + (driftSig * d.s.driftAmount) * 0.02f;  // kDriftScale, :247
    }
    if (anyOff)  // voices of instruments that have not seen an event yet: processVector is a no-op, the rows are zero (:383-386)
    {
#pragma unroll
      for (int k = 0; k < 4; ++k) oPitch[k] = on ? oPitch[k] : 0.f;
    }
    if (moving)
    {
#pragma unroll
      for (int k = 0; k < 4; ++k) setGlideWord(5 + 4 * q + k, f2u(c[k]));
    }
  }

  MLD void end_vector()
  {
    if (on) gd.endVector();
  }
};

}  // namespace mlev
