// mldsp_events.hpp — the device side of EventsToSignals (source/app/MLEventsToSignals.{h,cpp}) shared by the events kernel
// (events.hip: all 8 rows into HBM) and the run-time fused graph kernels (graph.hip: pitch and gate as source nodes of a voice
// graph, never written to memory): the record format the host router produces, the per-voice state layout, LinearGlide with
// its 64 slots, and EventsVoice - Voice::beginProcess / writeNoteEvent / endProcess (:75-262) for the pitch and gate rows,
// one quad of frames at a time.
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
  const uint32_t* recStart;   // [lanes + 1]
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
template <class SetGlideTime, class GlideNext>
MLD void note_frame(const Rec* recs, uint32_t& nc, uint32_t vend, int n, bool& preApplied, float& velocity, float& pitch, uint32_t& age, uint32_t& ageStep,
                    bool& inhibit, int32_t pitchGlideSamples, bool wantTime, double srD, SetGlideTime setPitchGlideTime, GlideNext pitchGlideNext,
                    float& vPitch, float& vGate, float& vTime)
{
  bool retrigFrame = false;
  while (nc < vend)
  {
    const Rec rc = recs[nc];
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
    while (pi < vend && ((recs[pi].typeTimeFlags & 0xFF) < REC_NOTE_ON || (recs[pi].typeTimeFlags & 0xFF) > REC_NOTE_OFF)) ++pi;
    if (pi >= vend) break;
    const Rec P = recs[pi];
    const uint32_t ptype = P.typeTimeFlags & 0xFF;
    int pdest = (int)((P.typeTimeFlags >> 8) & 0xFF);
    if (ptype == REC_NOTE_RETRIG && pdest == 0) pdest = 1;
    if (pdest != n + 1) break;
    uint32_t ri = pi + 1;
    while (ri < vend && ((recs[ri].typeTimeFlags & 0xFF) < REC_NOTE_ON || (recs[ri].typeTimeFlags & 0xFF) > REC_NOTE_OFF)) ++ri;
    if (ri >= vend) break;
    const Rec R = recs[ri];
    const int rdest = (int)((R.typeTimeFlags >> 8) & 0xFF);
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

// ---- EventsToSignals inside a fused voice graph ------------------------------------------------------------------------------
// The pitch and gate rows of one voice, produced a quad of frames at a time for the graph kernel that consumes them, from the
// same records and the same per-voice state words as e2s_kernel (a launch of either leaves the state the other expects; rows
// that are not computed keep their glides where they are, as with mlgpu_events_set_wanted_rows). MIDI protocol only: one lane
// per playing voice, lane == voice index. A vector with a note event is walked with note_frame(), as e2s_kernel walks it.
template <bool B>
struct BoolTag
{
  static constexpr bool value = B;
};

struct EventsVoice
{
  typedef float f32x4e __attribute__((ext_vector_type(4)));
  uint32_t* S;  // this lane's first state word
  size_t ln;
  const Rec* recs;
  E2SSettings s;
  bool live, awake, inhibit, needsRecalc;
  float velocity, pitch, bend;
  uint32_t age, ageStep;
  float pgCurr, pgStep, pgTarget, pgDy;
  int32_t pgRemaining, pgPerGlide;
  uint32_t cursor, recEnd;
  // per vector
  uint32_t vend, nc;
  bool on, quiet, preApplied;
  float gateHeld;
  Glide gb, gd;
  float nb[4], nd[4];  // bend / drift mCurrVec slots, one quad ahead
  // how a quiet vector's pitch row is produced (wave-uniform, decided in begin_vector):
  //   0  frame by frame through the three glides, as e2s_kernel does
  //   1  one value: the pitch glide at rest, bend and drift held
  //   2  pitch glide at rest, bend held, the drift glide moving in some lane: base + drift, straight-line
  int pitchForm;
  float pitchBase, driftHeld, frameNo;
  uint64_t mDriftMoves, mDriftRamps;
  bool driftMoving, driftFetches;

  MLD uint32_t& sw(int i) const { return S[(size_t)i * ln]; }
  MLD uint32_t* gs(int i) const { return S + (size_t)(S_GLIDES + i * kGlideWords) * ln; }
  MLD void setPitchGlideTime(int32_t t)  // SampleAccurateLinearGlide::setGlideTimeInSamples, MLDSPGens.h:527-532
  {
    pgPerGlide = t < 1 ? 1 : t;
    pgDy = 1.0f / (float)pgPerGlide;
  }
  MLD float pitchGlideNext(float f)  // nextSample, :541-580
  {
    if (f != pgTarget)
    {
      pgTarget = f;
      pgRemaining = pgPerGlide;
    }
    if (pgRemaining < 0) {}
    else if (pgRemaining == 0)
    {
      pgCurr = pgTarget;
      pgStep = 0.f;
      pgRemaining--;
    }
    else if (pgRemaining == pgPerGlide)
    {
      pgStep = (pgTarget - pgCurr) * pgDy;
      pgRemaining--;
    }
    else
    {
      pgCurr += pgStep;
      pgRemaining--;
    }
    return pgCurr;
  }

  MLD void load(const EventsDev& a, size_t lane)
  {
    live = lane < a.lanes;
    const size_t L = live ? lane : 0;
    ln = a.lanes;
    S = a.state + L;
    recs = (const Rec*)a.recs;
    s = a.s;
    awake = sw(S_AWAKE) != 0;
    velocity = u2f(sw(S_VELOCITY));
    pitch = u2f(sw(S_PITCH));
    bend = u2f(sw(S_BEND));
    age = sw(S_AGE);
    ageStep = sw(S_AGE_STEP);
    inhibit = sw(S_INHIBIT_GLIDE) != 0;
    needsRecalc = sw(S_RECALC) != 0;
    pgCurr = u2f(sw(S_PG_CURR));
    pgStep = u2f(sw(S_PG_STEP));
    pgTarget = u2f(sw(S_PG_TARGET));
    pgDy = u2f(sw(S_PG_DY));
    pgRemaining = (int32_t)sw(S_PG_REMAINING);
    pgPerGlide = (int32_t)sw(S_PG_PER_GLIDE);
    cursor = live ? a.recStart[L] : 0;
    recEnd = live ? a.recStart[L + 1] : 0;
  }
  MLD void store() const
  {
    if (!live) return;
    sw(S_AWAKE) = awake ? 1u : 0u;
    sw(S_VELOCITY) = f2u(velocity);
    sw(S_PITCH) = f2u(pitch);
    sw(S_BEND) = f2u(bend);
    sw(S_AGE) = age;
    sw(S_AGE_STEP) = ageStep;
    sw(S_INHIBIT_GLIDE) = inhibit ? 1u : 0u;
    sw(S_RECALC) = needsRecalc ? 1u : 0u;
    sw(S_PG_CURR) = f2u(pgCurr);
    sw(S_PG_STEP) = f2u(pgStep);
    sw(S_PG_TARGET) = f2u(pgTarget);
    sw(S_PG_DY) = f2u(pgDy);
    sw(S_PG_REMAINING) = (uint32_t)pgRemaining;
    sw(S_PG_PER_GLIDE) = (uint32_t)pgPerGlide;
  }

  // Records of this lane in DSPVector t: [cursor, vend). The generated kernel asks every lane of the wavefront, and runs
  // the vector through begin_vector / quad<NO_RECS = true> when no lane has one (the usual case by far: a voice sees a
  // handful of events per second): that instance has no record walk and no note-frame loop in it, and the registers the
  // note path needs are not held through the vectors that do not use it.
  MLD bool scan(size_t t)
  {
    vend = cursor;
    while (vend < recEnd && recs[vend].vec == (uint32_t)t) ++vend;
    return vend != cursor;
  }
  template <bool NO_RECS = false>
  MLD void begin_vector(size_t t)
  {
    (void)t;  // [cursor, vend) comes from scan(t), which the kernel calls first
    if constexpr (!NO_RECS)
    {
      if (!awake)
        for (uint32_t r = cursor; r < vend; ++r)
          if ((recs[r].typeTimeFlags & 0xFF) == REC_AWAKE) awake = true;
    }
    float finalVelocity = velocity;
    bool noteHere = false;
    float driftValue = 0.f;
    on = awake && live;
    if (on)
    {
      // ---- Voice::beginProcess, :75-113 (the drift walk's four words stay in memory: they are touched once per vector) ----
      if (needsRecalc)
      {
        if (!inhibit) setPitchGlideTime(s.pitchGlideSamples);
        needsRecalc = false;
      }
      int32_t driftCounter = (int32_t)sw(S_DRIFT_COUNTER) + MLGPU_FLOATS_PER_DSPVECTOR;
      driftValue = u2f(sw(S_DRIFT_VALUE));
      if (driftCounter >= (int32_t)sw(S_DRIFT_NEXT))
      {
        uint32_t driftSeed = sw(S_DRIFT_SEED);
        driftSeed = driftSeed * 0x0019660Du + 0x3C6EF35Fu;  // RandomScalarSource::getFloat, MLDSPScalarMath.h:189-202
        const float d = u2f(((driftSeed >> 9) & 0x007FFFFFu) | 0x3F800000u) * 2.f - 3.f;
        driftSeed = driftSeed * 0x0019660Du + 0x3C6EF35Fu;
        const float d2 = u2f(((driftSeed >> 9) & 0x007FFFFFu) | 0x3F800000u) * 2.f - 3.f;
        const float nextTimeMul = 1.0f + abs_ps(d2);
        driftValue = d;
        driftCounter = 0;
        sw(S_DRIFT_SEED) = driftSeed;
        sw(S_DRIFT_VALUE) = f2u(driftValue);
        sw(S_DRIFT_NEXT) = (uint32_t)(int32_t)(s.sr * (double)nextTimeMul * (double)8.0f);
      }
      sw(S_DRIFT_COUNTER) = (uint32_t)driftCounter;
      // ---- values that only matter at the end of the vector (endProcess, :218-247); the rows this object does not compute
      //      keep their values in memory ----
      for (uint32_t r = cursor; !NO_RECS && r < vend; ++r)
      {
        const Rec rc = recs[r];
        switch (rc.typeTimeFlags & 0xFF)
        {
          case REC_SET_BEND: bend = rc.v1; break;
          case REC_SET_MOD: sw(S_MOD) = f2u(rc.v1); break;
          case REC_SET_X: sw(S_X) = f2u(rc.v1); break;
          case REC_SET_Y: sw(S_Y) = f2u(rc.v1); break;
          case REC_SET_Z: sw(S_Z) = f2u(rc.v1); break;
          case REC_SET_CHANNEL_PRESSURE: sw(S_CHANPRESS) = f2u(rc.v1); break;
          case REC_NOTE_ON: case REC_NOTE_RETRIG: finalVelocity = rc.v2; noteHere = true; break;
          case REC_NOTE_OFF: finalVelocity = 0.f; noteHere = true; break;
          default: break;
        }
      }
      if (finalVelocity == 0.f) sw(S_Z) = 0u;  // :238-241
    }
    gb.load(gs(0), ln);
    gd.load(gs(5), ln);
    if (on)
    {
      gb.beginVector(gs(0), ln, bend, s.glideVectors, s.glideDy);
      gd.beginVector(gs(5), ln, driftValue, s.driftGlideVectors, s.driftGlideDy);
    }
    nc = cursor;
    preApplied = false;
    quiet = NO_RECS || __builtin_amdgcn_ballot_w64(noteHere) == 0;
    pitchForm = 0;
    mDriftMoves = mDriftRamps = 0;
    if (quiet)
    {
      gateHeld = on ? velocity : 0.f;
      if (on) age += (uint32_t)MLGPU_FLOATS_PER_DSPVECTOR * ageStep;
      // The pitch row = pitch glide + bend glide + drift glide, three small per-lane state machines: walked as such they are
      // ~36 vector and ~44 scalar / branch instructions per frame, although in a vector without note events nearly all of it
      // is decided once: the pitch glide is usually at rest (no portamento in progress: nextSample returns mCurr and changes
      // nothing), the bend is usually held (one value for the vector), a LinearGlide's mode is fixed for the vector. Classify
      // once per vector, wave-uniformly, and run the frames of the common classes as straight-line code: the same operations
      // on the same values.
      const bool pgBusy = on && (pitch != pgTarget || pgRemaining >= 0);
      const int bm = gb.mode(), dm = gd.mode();
      const bool heldB = (bm == 1) || (bm == 0 && gb.isUniform()), heldD = (dm == 1) || (dm == 0 && gd.isUniform());
      const float hvB = (bm == 1) ? gb.target : gb.uniformValue, hvD = (dm == 1) ? gd.target : gd.uniformValue;
      if (__builtin_amdgcn_ballot_w64(pgBusy || (on && !heldB)) == 0)
      {
        pitchBase = pgCurr + (hvB * s.pitchBendRange) * (1.f / 12);  // the frames' vPitch after :244, the same for all 64
        if (__builtin_amdgcn_ballot_w64(on && !heldD) == 0)
        {
          pitchForm = 1;
          pitchBase = on ? pitchBase + (hvD * s.driftAmount) * 0.02f : 0.f;
        }
        else
        {
          // per lane and vector: a moving drift glide ramps (its first vector) or adds a step to its mCurrVec slot and writes
          // it back; the others hold a value
          pitchForm = 2;
          mDriftMoves = __builtin_amdgcn_ballot_w64(on && dm >= 2);
          mDriftRamps = __builtin_amdgcn_ballot_w64(on && dm == 2);
          driftMoving = on && dm >= 2;
          driftFetches = on && gd.readsCurrVec();
          driftHeld = hvD;  // what a lane that neither moves nor fetches holds (mode 3 with a broadcast mCurrVec adds to it)
          frameNo = 0.f;
          nd[0] = nd[1] = nd[2] = nd[3] = driftHeld;
          if (driftFetches)
          {
            const uint32_t* slots = gs(5) + (size_t)5 * ln;
#pragma unroll
            for (int k = 0; k < 4; ++k) nd[k] = u2f(slots[(size_t)k * ln]);
          }
        }
      }
      else
      {
        nb[0] = nb[1] = nb[2] = nb[3] = 0.f;
        nd[0] = nd[1] = nd[2] = nd[3] = 0.f;
        if (on)
        {
          gb.preload(gs(0), ln, 0, nb);
          gd.preload(gs(5), ln, 0, nd);
        }
      }
    }
  }

  template <bool NO_RECS = false>
  MLD void quad(int q, f32x4e& oPitch, f32x4e& oGate)
  {
    const float pitchBendScale = s.pitchBendRange;  // MIDI protocol, :417-423
    if (pitchForm == 1)
    {
      oPitch = f32x4e{pitchBase, pitchBase, pitchBase, pitchBase};
      oGate = f32x4e{gateHeld, gateHeld, gateHeld, gateHeld};
      return;
    }
    if (pitchForm == 2)
    {
      uint32_t* slots = gs(5) + (size_t)5 * ln;
      const float prev[4] = {nd[0], nd[1], nd[2], nd[3]};
      if (driftFetches && q < 15)
      {
#pragma unroll
        for (int k = 0; k < 4; ++k) nd[k] = u2f(slots[(size_t)(4 * q + 4 + k) * ln]);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k)
      {
        frameNo += 1.0f;  // (float)(n + 1)
        const float c = lane_select(mDriftRamps, gd.startValue + (frameNo * 0.015625f) * gd.step, prev[k] + gd.step);
        if (driftMoving) slots[(size_t)(4 * q + k) * ln] = f2u(c);
        const float driftSig = lane_select(mDriftMoves, c, prev[k]);
        oPitch[k] = on ? pitchBase + (driftSig * s.driftAmount) * 0.02f : 0.f;
        oGate[k] = gateHeld;
      }
      return;
    }
    if (NO_RECS || quiet)
    {
      const float cb[4] = {nb[0], nb[1], nb[2], nb[3]}, cd[4] = {nd[0], nd[1], nd[2], nd[3]};
      if (on && q < 15)
      {
        gb.preload(gs(0), ln, q + 1, nb);
        gd.preload(gs(5), ln, q + 1, nd);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k)
      {
        const int n = q * 4 + k;
        float vPitch = 0.f;
        if (on)
        {
          vPitch = pitchGlideNext(pitch);
          const float bendSig = gb.nextWith(gs(0), ln, n, cb[k]), driftSig = gd.nextWith(gs(5), ln, n, cd[k]);
          vPitch = vPitch + (bendSig * pitchBendScale) * (1.f / 12);  // :244
          vPitch = vPitch + (driftSig * s.driftAmount) * 0.02f;       // kDriftScale, :247
        }
        oPitch[k] = vPitch;
        oGate[k] = gateHeld;
      }
      return;
    }
    oPitch = f32x4e{0.f, 0.f, 0.f, 0.f};
    oGate = oPitch;
    if constexpr (NO_RECS) return;  // not reached: a vector without records is quiet
#pragma unroll 1
    for (int k = 0; k < 4; ++k)
    {
      const int n = q * 4 + k;
      float vPitch = 0.f, vGate = 0.f;
      if (on)
      {
      float vTime = 0.f;
      note_frame(recs, nc, vend, n, preApplied, velocity, pitch, age, ageStep, inhibit, s.pitchGlideSamples, false, 1.0,
                 [&](int32_t t) { setPitchGlideTime(t); }, [&](float f) { return pitchGlideNext(f); }, vPitch, vGate, vTime);
      const float bendSig = gb.next(gs(0), ln, n), driftSig = gd.next(gs(5), ln, n);
      vPitch = vPitch + (bendSig * pitchBendScale) * (1.f / 12);         // :244
      vPitch = vPitch + (driftSig * s.driftAmount) * 0.02f;           // kDriftScale, :247
      }
      // k is a loop variable here (the body is large): insert with selects instead of a dynamic register index
      oPitch = f32x4e{k == 0 ? vPitch : oPitch[0], k == 1 ? vPitch : oPitch[1], k == 2 ? vPitch : oPitch[2], k == 3 ? vPitch : oPitch[3]};
      oGate = f32x4e{k == 0 ? vGate : oGate[0], k == 1 ? vGate : oGate[1], k == 2 ? vGate : oGate[2], k == 3 ? vGate : oGate[3]};
    }
  }

  MLD void end_vector()
  {
    if (on)
    {
      gb.endVector();
      gd.endVector();
      gb.store(gs(0), ln);
      gd.store(gs(5), ln);
    }
    cursor = vend;
  }
};

}  // namespace mlev
