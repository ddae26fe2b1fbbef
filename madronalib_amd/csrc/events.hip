// events.hip — EventsToSignals (source/app/MLEventsToSignals.{h,cpp}) for N independent instruments.
//
// The reference turns a time-sorted list of performance events (note on/off, controllers, pitch bend, pressure,
// sustain pedal) into 8 control signals per voice — pitch, gate, vox, z, x, y, mod, elapsed time
// (VoiceOutputSignals, MLEventsToSignals.h:15-26) — one DSPVector at a time on the audio thread. Split here as
// SURVEY §8f-1 prescribes:
//   HOST   the event routing: key states, voice allocation and stealing, unison, sustain pedal, MIDI / MPE channel
//          rules (processEvent & co, MLEventsToSignals.cpp:445-870; findFreeVoice / findNearestVoice :892-935).
//          Integer bookkeeping on a handful of events per block; it produces, per voice, a short list of timed
//          records ("note on at frame 17 with pitch p, velocity v", "pitch bend is now b", ...).
//   DEVICE everything per sample: Voice::beginProcess / writeNoteEvent / endProcess (:75-262) — the sample-accurate
//          pitch glide, event age -> seconds, five vector-rate glides, the drift random walk, channel pressure and
//          the MPE main-voice sums — one wavefront lane per voice, all state in HBM between launches only.
// Uploading the signals themselves would cost 8 x 256 B per voice per DSPVector; the records are a few bytes per event.
//
// One instrument occupies G = nextpow2(polyphony + 1) consecutive lanes: lane 0 is the MPE main voice (voices[0] in the
// reference), lanes 1..polyphony the playing voices, so the main voice's signals reach the others with one wave shuffle.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <new>
#include <vector>

#include "mlgpu_internal.hpp"
#include "mldsp_math.hpp"
#include "mldsp_events.hpp"

using namespace mldev;
using namespace mlev;

namespace
{
inline Rec makeRec(uint32_t vec, uint32_t type, int time, uint32_t flags, float v1, float v2)
{
  const uint32_t t = (uint32_t)std::min(std::max(time, 0), 64);  // destTime = clamp(e.time, 0, 64) (:121)
  return Rec{vec, type | (t << 8) | (flags << 16), v1, v2};
}

#ifndef MLGPU_E2S_WAVES
#define MLGPU_E2S_WAVES 4
#endif
// ROWS01: only the pitch and gate rows are wanted (what a Synth's voices usually read): an instance of the kernel without the controller
// rows, the voice-index row and the elapsed-time row. Those are half of the general loop's code and registers; without them the
// instance keeps its state in registers instead of scratch, and its loop has fewer memory round trips per DSPVector.
template <bool ROWS01>
__global__ __launch_bounds__(256, MLGPU_E2S_WAVES) void e2s_kernel(const E2SArgs aIn)
{
  E2SArgs a = aIn;
  if constexpr (ROWS01) a.rowMask &= 3u;
  apply_fp_mode(a.flags);
  // XCD-aware workgroup -> lane mapping (as the voice-bank kernels): every XCD writes one contiguous eighth of each row
  size_t blk = blockIdx.x;
  const size_t nbFull = (size_t)gridDim.x & ~(size_t)7;
  if (blk < nbFull) blk = (blk & 7) * (nbFull >> 3) + (blk >> 3);
  const size_t lane = blk * 256 + threadIdx.x;
  const bool live = lane < a.lanes;
  const size_t L = live ? lane : 0;
  const int slot = (int)(L % (size_t)a.group) + a.slotBase;  // 0 = MPE main voice (MPE mode only), 1..polyphony = playing voices
  const bool isVoice = live && slot >= 1 && slot <= a.polyphony;
  const bool active = live && slot <= a.polyphony;       // lanes beyond polyphony are padding
  const size_t outVoice = (L / (size_t)a.group) * (size_t)a.polyphony + (size_t)(slot > 0 ? slot - 1 : 0);
  uint32_t* S = a.state + L;  // reassigned before the final stores
  const size_t ln = a.lanes;
#define SW(i) S[(size_t)(i) * ln]

  bool awake = SW(S_AWAKE) != 0;
  float velocity = u2f(SW(S_VELOCITY)), pitch = u2f(SW(S_PITCH)), bend = u2f(SW(S_BEND)), mod = u2f(SW(S_MOD));
  float cx = u2f(SW(S_X)), cy = u2f(SW(S_Y)), cz = u2f(SW(S_Z)), chanPress = u2f(SW(S_CHANPRESS));
  uint32_t age = SW(S_AGE), ageStep = SW(S_AGE_STEP);
  bool inhibit = SW(S_INHIBIT_GLIDE) != 0, needsRecalc = SW(S_RECALC) != 0;
  float pgCurr = u2f(SW(S_PG_CURR)), pgStep = u2f(SW(S_PG_STEP)), pgTarget = u2f(SW(S_PG_TARGET)), pgDy = u2f(SW(S_PG_DY));
  int32_t pgRemaining = (int32_t)SW(S_PG_REMAINING), pgPerGlide = (int32_t)SW(S_PG_PER_GLIDE);
  uint32_t driftSeed = SW(S_DRIFT_SEED);
  int32_t driftCounter = (int32_t)SW(S_DRIFT_COUNTER), driftNext = (int32_t)SW(S_DRIFT_NEXT);
  float driftValue = u2f(SW(S_DRIFT_VALUE));
  // the seven glides stay in HBM between uses: each row loop loads the one or two it needs (5 words), so their
  // registers are not live everywhere
#define GS(i) (S + (size_t)(S_GLIDES + (i) * kGlideWords) * ln)

  auto setPitchGlideTime = [&](int32_t t) {  // SampleAccurateLinearGlide::setGlideTimeInSamples, MLDSPGens.h:527-532
    pgPerGlide = t < 1 ? 1 : t;
    pgDy = 1.0f / (float)pgPerGlide;
  };
  auto pitchGlideNext = [&](float f) {  // nextSample, :541-580
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
  };

  const uint2 recRange = live ? a.recRange[L] : make_uint2(0u, 0u);  // this lane's records of this launch: [x, y)
  uint32_t cursor = recRange.x;
  const uint32_t recEnd = recRange.y;
  const float pitchBendScale = (a.s.mpe && slot != 0) ? a.s.mpePitchBendRange : a.s.pitchBendRange;  // :417-423
  const double srD = (double)(float)a.s.sr;  // samplesToSeconds(uint32_t, float sr), :13-19
  const unsigned mainLane = (unsigned)((threadIdx.x & 63) / (unsigned)a.group) * (unsigned)a.group;
  const bool wantTime = !ROWS01 && (a.rowMask & (1u << 7)) != 0;  // the elapsed-time row costs an f64 division per sample

  // ---- blocks of vectors in which nothing happens ------------------------------------------------------------------------------
  // The duration of a launch is the duration of its slowest wavefront, and memory operations of a wavefront complete in issue
  // order: every load behind a store waits for that store's acknowledgement (several microseconds while the chip writes rows at
  // its ceiling). The general vector loop below has a dozen such round trips per DSPVector - glide state, slots a quad at a
  // time, records - and a wavefront that carried a note event used to run it for all its vectors, ~40 us each, long after the
  // others were done (profiles/archive/r03_e2s_latency.txt). So: kBlock vectors at a time, where no lane of the wavefront has a record, the
  // bend and the wanted controllers are at rest, take this path - ONE round trip: the glide words and all 64 slots of the drift
  // glide (8 s per glide: always moving) are fetched together, the block is computed in time order in registers (the pitch glide may
  // be moving: it is stepped sample by sample as ever; what each vector does to the drift glide - hold / end / start / continue -
  // depends on the drift counter alone), rows are written as they come, the slots go back once. Same operations on the same
  // values as the general loop; the state variables are this kernel's own, so the two forms alternate freely.
#ifndef MLGPU_E2S_BLOCK
#define MLGPU_E2S_BLOCK 4
#endif
  constexpr int kBlock = MLGPU_E2S_BLOCK;
  typedef float f32x4b __attribute__((ext_vector_type(4)));
  for (size_t t = 0; t < a.T; ++t)
  {
    if (a.blockPath && !a.s.mpe && t + kBlock <= a.T)
    {
      const bool onB = awake && active;
      bool ok = (cursor >= recEnd) || (a.recs[cursor].vec >= (uint32_t)(t + kBlock));
      float heldBend = 0.f, heldMod = 0.f, heldX = 0.f, heldY = 0.f, heldZ = 0.f;
      const float czEff = (velocity == 0.f) ? 0.f : cz;  // :238-241 with finalVelocity == velocity
      if (ok && onB)
      {
        auto rests = [&](int gi, float value, float& held) {  // glide gi holds `value` with a broadcast mCurrVec: its vectors change nothing
          Glide g;
          g.load(GS(gi), ln);
          held = g.uniformValue;
          return g.remaining < 0 && g.isUniform() && g.target == value;
        };
        ok = !needsRecalc && rests(0, bend, heldBend);
        if (!ROWS01 && (a.rowMask & (1u << 6))) ok = ok && rests(1, mod, heldMod);
        if (!ROWS01 && (a.rowMask & (1u << 4))) ok = ok && rests(2, cx, heldX);
        if (!ROWS01 && (a.rowMask & (1u << 5))) ok = ok && rests(3, cy, heldY);
        if (!ROWS01 && (a.rowMask & (1u << 3)))
        {
          float heldP = 0.f;
          ok = ok && rests(4, czEff, heldZ) && rests(6, chanPress, heldP);
          heldZ = heldZ + heldP;  // MIDI mode: z adds the smoothed channel pressure (:437-445)
        }
      }
      if (__builtin_amdgcn_ballot_w64(!ok) == 0)
      {
        uint32_t* gs = GS(5);
        Glide gd;
        gd.load(gs, ln);
        const bool slotsLive = onB && !gd.isUniform();  // mCurrVec is in memory as the block starts
        if (onB) cz = czEff;
        auto rowOfOne = [&](int row, float value) {  // one value per lane for the whole block
          const SignalView& sv = a.out[row];
          if ((ROWS01 && row > 1) || !((a.rowMask >> row) & 1u) || !sv.base || !isVoice) return;
          f32x4b* p = (f32x4b*)sv.base + t * sv.strideT + outVoice * sv.strideV;
          const f32x4b v = {value, value, value, value};
          for (int b = 0; b < kBlock; ++b, p += sv.strideT)
          {
            f32x4b* pq = p;
#pragma unroll 4
            for (int q = 0; q < 16; ++q, pq += sv.strideQ) __builtin_nontemporal_store(v, pq);
          }
        };
        rowOfOne(1, onB ? velocity : 0.f);
        rowOfOne(2, (float)(slot - 1));
        rowOfOne(3, onB ? heldZ : 0.f);
        rowOfOne(4, onB ? heldX : 0.f);
        rowOfOne(5, onB ? heldY : 0.f);
        rowOfOne(6, onB ? heldMod : 0.f);
        if (wantTime && a.out[7].base && isVoice)
        {
          const SignalView& sv = a.out[7];
          f32x4b* p = (f32x4b*)sv.base + t * sv.strideT + outVoice * sv.strideV;
          uint32_t ageNow = age;
          for (int b = 0; b < kBlock; ++b, p += sv.strideT)
          {
            f32x4b* pq = p;
            for (int q = 0; q < 16; ++q, pq += sv.strideQ)
            {
              f32x4b v;
#pragma unroll
              for (int k = 0; k < 4; ++k) v[k] = onB ? (float)((double)(ageNow + (uint32_t)(q * 4 + k + 1) * ageStep) / srD) : 0.f;
              __builtin_nontemporal_store(v, pq);
            }
            ageNow += (uint32_t)MLGPU_FLOATS_PER_DSPVECTOR * ageStep;
          }
        }
        if (onB) age += (uint32_t)(kBlock * MLGPU_FLOATS_PER_DSPVECTOR) * ageStep;
        // ---- the pitch row ----
        // What each of the block's vectors does to the drift glide, worked out first (it needs slot 63 only: what a starting
        // glide reads). Then the 64 slots in two halves of 32: a half's slots are fetched together, stepped through the kBlock
        // vectors in registers and written back. The pitch glide is a per-sample recurrence in time order: each half walks it
        // through the whole block from the block's start state, using the samples of its own half (when it rests, as mostly,
        // that is a constant).
        const SignalView& sp = a.out[0];
        const bool storePitch = sp.base && isVoice;
        const float bendTerm = (heldBend * pitchBendScale) * (1.f / 12);  // :244
        float eff63 = slotsLive ? u2f(gs[(size_t)(5 + 63) * ln]) : 0.f;
        int cm[kBlock];
        bool cu[kBlock];
        float cstep[kBlock], cstart[kBlock], ctarget[kBlock], cuval[kBlock];
        bool wrote = false;
#pragma unroll
        for (int b = 0; b < kBlock; ++b)
        {
          cm[b] = 0;
          cu[b] = true;
          cstep[b] = cstart[b] = ctarget[b] = cuval[b] = 0.f;
          if (onB)
          {
            // the drift part of Voice::beginProcess (:115-126), then the drift glide's own start of a vector
            driftCounter += MLGPU_FLOATS_PER_DSPVECTOR;
            if (driftCounter >= driftNext)
            {
              driftSeed = driftSeed * 0x0019660Du + 0x3C6EF35Fu;
              const float d = u2f(((driftSeed >> 9) & 0x007FFFFFu) | 0x3F800000u) * 2.f - 3.f;
              driftSeed = driftSeed * 0x0019660Du + 0x3C6EF35Fu;
              const float d2 = u2f(((driftSeed >> 9) & 0x007FFFFFu) | 0x3F800000u) * 2.f - 3.f;
              const float nextTimeMul = 1.0f + abs_ps(d2);
              driftValue = d;
              driftCounter = 0;
              driftNext = (int32_t)(a.s.sr * (double)nextTimeMul * (double)8.0f);
            }
            gd.beginVectorKnown(driftValue, a.s.driftGlideVectors, a.s.driftGlideDy, eff63);
            cm[b] = gd.mode();
            cu[b] = gd.isUniform();
            cstep[b] = gd.step;
            cstart[b] = gd.startValue;
            ctarget[b] = gd.target;
            cuval[b] = gd.uniformValue;
            // slot 63 after this vector (n = 63: (float)(n + 1) * 0.015625f == 1)
            if (cm[b] == 2) eff63 = cstart[b] + 1.0f * cstep[b];
            else if (cm[b] == 3) eff63 = (cu[b] ? cuval[b] : eff63) + cstep[b];
            wrote = wrote || (cm[b] >= 2);
            gd.endVector();
          }
        }
        const bool pgMoves = __builtin_amdgcn_ballot_w64(onB && !(pgRemaining < 0 && pgTarget == pitch)) != 0;  // any lane's pitch glide
        const float pg0Curr = pgCurr, pg0Step = pgStep, pg0Target = pgTarget;  // the pitch glide as the block starts
        const int32_t pg0Remaining = pgRemaining;
        constexpr int kHalf = 32;
#pragma unroll 1
        for (int h = 0; h < 64; h += kHalf)
        {
          float c[kHalf];
#pragma unroll
          for (int i = 0; i < kHalf; ++i) c[i] = slotsLive ? u2f(gs[(size_t)(5 + h + i) * ln]) : 0.f;
          pgCurr = pg0Curr;
          pgStep = pg0Step;
          pgTarget = pg0Target;
          pgRemaining = pg0Remaining;
          f32x4b* pb = (f32x4b*)sp.base + t * sp.strideT + outVoice * sp.strideV + (size_t)(h / 4) * sp.strideQ;
#pragma unroll
          for (int b = 0; b < kBlock; ++b, pb += sp.strideT)
          {
            if (onB && pgMoves)
              for (int n = 0; n < h; ++n) (void)pitchGlideNext(pitch);  // the samples of this vector before this half
            f32x4b* pq = pb;
#pragma unroll
            for (int q = 0; q < kHalf / 4; ++q, pq += sp.strideQ)
            {
              f32x4b o;
#pragma unroll
              for (int k = 0; k < 4; ++k)
              {
                const int i = q * 4 + k, n = h + i;
                float vPitch = 0.f;
                if (onB)
                {
                  const int m = cm[b];
                  float v;
                  if (m == 0) v = cu[b] ? cuval[b] : c[i];
                  else if (m == 1) v = ctarget[b];
                  else
                  {
                    if (m == 2) v = cstart[b] + ((float)(n + 1) * 0.015625f) * cstep[b];
                    else v = (cu[b] ? cuval[b] : c[i]) + cstep[b];
                    c[i] = v;
                  }
                  vPitch = pitchGlideNext(pitch);
                  vPitch = vPitch + bendTerm;
                  vPitch = vPitch + (v * a.s.driftAmount) * 0.02f;           // kDriftScale, :247
                }
                o[k] = vPitch;
              }
              if (storePitch) __builtin_nontemporal_store(o, pq);
            }
            if (onB && pgMoves)
              for (int n = h + kHalf; n < 64; ++n) (void)pitchGlideNext(pitch);  // ... and after it
          }
          if (wrote)
#pragma unroll
            for (int i = 0; i < kHalf; ++i) gs[(size_t)(5 + h + i) * ln] = f2u(c[i]);
        }
        if (onB) gd.store(gs, ln);
        t += kBlock - 1;
        continue;
      }
    }
    // records of this vector: [cursor, vend)
    uint32_t vend = cursor;
    while (vend < recEnd && a.recs[vend].vec == (uint32_t)t) ++vend;
    if (!awake)
      for (uint32_t r = cursor; r < vend; ++r)
        if ((a.recs[r].typeTimeFlags & 0xFF) == REC_AWAKE) awake = true;

    float finalVelocity = velocity;
    bool noteHere = false;  // a note record of this lane falls into this vector
    if (awake && active)
    {
      // ---- Voice::beginProcess, :75-113 ----
      if (needsRecalc)
      {
        if (!inhibit) setPitchGlideTime(a.s.pitchGlideSamples);
        needsRecalc = false;
      }
      driftCounter += MLGPU_FLOATS_PER_DSPVECTOR;
      if (driftCounter >= driftNext)
      {
        driftSeed = driftSeed * 0x0019660Du + 0x3C6EF35Fu;  // RandomScalarSource::getFloat, MLDSPScalarMath.h:189-202
        const float d = u2f(((driftSeed >> 9) & 0x007FFFFFu) | 0x3F800000u) * 2.f - 3.f;
        driftSeed = driftSeed * 0x0019660Du + 0x3C6EF35Fu;
        const float d2 = u2f(((driftSeed >> 9) & 0x007FFFFFu) | 0x3F800000u) * 2.f - 3.f;
        const float nextTimeMul = 1.0f + abs_ps(d2);
        driftValue = d;
        driftCounter = 0;
        driftNext = (int32_t)(a.s.sr * (double)nextTimeMul * (double)8.0f);
      }
      // ---- values that only matter at the end of the vector: apply them now (endProcess, :218-247) ----
      for (uint32_t r = cursor; r < vend; ++r)
      {
        const Rec rc = a.recs[r];
        switch (rc.typeTimeFlags & 0xFF)
        {
          case REC_SET_BEND: bend = rc.v1; break;
          case REC_SET_MOD: mod = rc.v1; break;
          case REC_SET_X: cx = rc.v1; break;
          case REC_SET_Y: cy = rc.v1; break;
          case REC_SET_Z: cz = rc.v1; break;
          case REC_SET_CHANNEL_PRESSURE: chanPress = rc.v1; break;
          case REC_NOTE_ON: case REC_NOTE_RETRIG: finalVelocity = rc.v2; noteHere = true; break;
          case REC_NOTE_OFF: finalVelocity = 0.f; noteHere = true; break;
          default: break;
        }
      }
      if (finalVelocity == 0.f) cz = 0.f;  // :238-241
    }
    const bool on = awake && active;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    auto put = [&](int row, int q, const f32x4 v) {
      const SignalView& sv = a.out[row];
      if (sv.base && isVoice) __builtin_nontemporal_store(v, (f32x4*)sv.base + t * sv.strideT + (size_t)q * sv.strideQ + outVoice * sv.strideV);
    };
    // MPE: the main voice's signal is added to every playing voice of its instrument (:447-458)
    auto withMain = [&](float v) {
      if (!a.s.mpe) return v;
      const float m = __shfl(v, mainLane, 64);
      return (isVoice && awake) ? v + m : v;
    };

    // ---- rows that are one glide each: mod, x, y; z adds the smoothed channel pressure in MIDI mode (:437-445) ----
    // One short loop per row keeps the live state of the other rows out of the registers.
    // A glide that is not moving gives one value for the whole vector (hold with a broadcast mCurrVec, or the vector that
    // ends a glide): when that is so for every lane of the wavefront - controllers rarely move - the row is 16 stores.
    auto heldValue = [&](const Glide& gl, bool& held) {
      const int m = gl.mode();
      held = (m == 1) || (m == 0 && gl.isUniform());
      return (m == 1) ? gl.target : gl.uniformValue;
    };
    auto glideRow = [&](int glideIdx, int row, float value) {
      Glide gl;
      gl.load(GS(glideIdx), ln);
      if (on) gl.beginVector(GS(glideIdx), ln, value, a.s.glideVectors, a.s.glideDy);
      bool held;
      const float hv = heldValue(gl, held);
      if (__builtin_amdgcn_ballot_w64(on && !held) == 0)
      {
        const float c = withMain(on ? hv : 0.f);
        const f32x4 v = {c, c, c, c};
#pragma unroll 1
        for (int q = 0; q < 16; ++q) put(row, q, v);
      }
      else
      {
        float nx[4] = {0.f, 0.f, 0.f, 0.f};
        if (on) gl.preload(GS(glideIdx), ln, 0, nx);
#pragma unroll 1
        for (int q = 0; q < 16; ++q)
        {
          const float cur[4] = {nx[0], nx[1], nx[2], nx[3]};
          if (on && q < 15) gl.preload(GS(glideIdx), ln, q + 1, nx);
          f32x4 v;
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] = withMain(on ? gl.nextWith(GS(glideIdx), ln, q * 4 + k, cur[k]) : 0.f);
          put(row, q, v);
        }
      }
      if (on)
      {
        gl.endVector();
        gl.store(GS(glideIdx), ln);
      }
    };
    if (!ROWS01 && (a.rowMask & (1u << 6))) glideRow(1, 6, mod);
    if (!ROWS01 && (a.rowMask & (1u << 4))) glideRow(2, 4, cx);
    if (!ROWS01 && (a.rowMask & (1u << 5))) glideRow(3, 5, cy);
    if (!ROWS01 && (a.rowMask & (1u << 3)))
    {
      Glide gz, gp;
      gz.load(GS(4), ln);
      gp.load(GS(6), ln);
      if (on)
      {
        gz.beginVector(GS(4), ln, cz, a.s.glideVectors, a.s.glideDy);
        gp.beginVector(GS(6), ln, chanPress, a.s.ctlGlideVectors, a.s.ctlGlideDy);  // SmoothedController::process, :268-280
      }
      bool heldZ, heldP;
      const float hz = heldValue(gz, heldZ), hp = heldValue(gp, heldP);
      if (__builtin_amdgcn_ballot_w64(on && !(heldZ && heldP)) == 0)
      {
        float z = 0.f;
        if (on)
        {
          z = hz;
          if (!a.s.mpe) z = z + hp;
        }
        const float c = withMain(z);
        const f32x4 v = {c, c, c, c};
#pragma unroll 1
        for (int q = 0; q < 16; ++q) put(3, q, v);
      }
      else
#pragma unroll 1
      for (int q = 0; q < 16; ++q)
      {
        float cz4[4] = {0.f, 0.f, 0.f, 0.f}, cp4[4] = {0.f, 0.f, 0.f, 0.f};
        if (on)
        {
          gz.preload(GS(4), ln, q, cz4);
          gp.preload(GS(6), ln, q, cp4);
        }
        f32x4 v;
#pragma unroll
        for (int k = 0; k < 4; ++k)
        {
          float z = 0.f;
          if (on)
          {
            z = gz.nextWith(GS(4), ln, q * 4 + k, cz4[k]);
            const float press = gp.nextWith(GS(6), ln, q * 4 + k, cp4[k]);
            if (!a.s.mpe) z = z + press;
          }
          v[k] = withMain(z);
        }
        put(3, q, v);
      }
      if (on)
      {
        gz.endVector();
        gp.endVector();
        gz.store(GS(4), ln);
        gp.store(GS(6), ln);
      }
    }
    if (!ROWS01 && (a.rowMask & (1u << 2)))
    {
      const float vox = (float)(slot - 1);  // row kVoice: DSPVector((float)i - 1), :302
      const f32x4 v = {vox, vox, vox, vox};
#pragma unroll 1
      for (int q = 0; q < 16; ++q) put(2, q, v);
    }

    // ---- gate, pitch, elapsed time: writeNoteEvent (:115-216) and endProcess (:218-262) walked frame by frame ----
    Glide gb, gd;
    gb.load(GS(0), ln);
    gd.load(GS(5), ln);
    if (on)
    {
      gb.beginVector(GS(0), ln, bend, a.s.glideVectors, a.s.glideDy);
      gd.beginVector(GS(5), ln, driftValue, a.s.driftGlideVectors, a.s.driftGlideDy);
    }
    uint32_t nc = cursor;      // next note record
    bool preApplied = false;   // a note event's bookkeeping applies from the frame the previous one ended at
    // Most vectors of most wavefronts hold no note event at all: then every frame is the same few steps (gate = velocity,
    // one step of the pitch glide, the event age, bend and drift), written out without the record walk.
    const bool quietWave = __builtin_amdgcn_ballot_w64(noteHere) == 0;
    if (quietWave)
    {
      // one short loop per row, as above: the gate is the held velocity, the event age advances by ageStep per frame
      {
        const float g = on ? velocity : 0.f;
        const f32x4 v = {g, g, g, g};
#pragma unroll 1
        for (int q = 0; q < 16; ++q) put(1, q, v);
      }
      if (wantTime)
      {
#pragma unroll 1
        for (int q = 0; q < 16; ++q)
        {
          f32x4 v;
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] = on ? (float)((double)(age + (uint32_t)(q * 4 + k + 1) * ageStep) / srD) : 0.f;
          put(7, q, v);
        }
      }
      if (on) age += (uint32_t)MLGPU_FLOATS_PER_DSPVECTOR * ageStep;
      float nb[4] = {0.f, 0.f, 0.f, 0.f}, nd[4] = {0.f, 0.f, 0.f, 0.f};  // bend / drift mCurrVec, one quad ahead
      if (on)
      {
        gb.preload(GS(0), ln, 0, nb);
        gd.preload(GS(5), ln, 0, nd);
      }
#pragma unroll 1
      for (int q = 0; q < 16; ++q)
      {
        const float cb[4] = {nb[0], nb[1], nb[2], nb[3]}, cd[4] = {nd[0], nd[1], nd[2], nd[3]};
        if (on && q < 15)
        {
          gb.preload(GS(0), ln, q + 1, nb);
          gd.preload(GS(5), ln, q + 1, nd);
        }
        f32x4 oPitch;
#pragma unroll
        for (int k = 0; k < 4; ++k)
        {
          const int n = q * 4 + k;
          float vPitch = 0.f;
          if (on)
          {
            vPitch = pitchGlideNext(pitch);
            const float bendSig = gb.nextWith(GS(0), ln, n, cb[k]), driftSig = gd.nextWith(GS(5), ln, n, cd[k]);
            vPitch = vPitch + (bendSig * pitchBendScale) * (1.f / 12);         // :244
            vPitch = vPitch + (driftSig * a.s.driftAmount) * 0.02f;           // kDriftScale, :247
          }
          oPitch[k] = withMain(vPitch);
        }
        put(0, q, oPitch);
      }
    }
    else
    {
    RecCache recCache;
    if (on) note_rewind(a.recs, nc, vend, velocity, pitch, age, ageStep, inhibit, a.s.pitchGlideSamples, setPitchGlideTime, pitchGlideNext);
#pragma unroll 1
    for (int q = 0; q < 16; ++q)
    {
      f32x4 oPitch = {0.f, 0.f, 0.f, 0.f}, oGate = oPitch, oTime = oPitch;
#pragma unroll 1
      for (int k = 0; k < 4; ++k)
      {
        const int n = q * 4 + k;
        float vPitch = 0.f, vGate = 0.f, vTime = 0.f;
        if (on)
        {
          note_frame(a.recs, recCache, nc, vend, n, preApplied, velocity, pitch, age, ageStep, inhibit, a.s.pitchGlideSamples, wantTime, srD, setPitchGlideTime,
                     pitchGlideNext, vPitch, vGate, vTime);
          const float bendSig = gb.next(GS(0), ln, n), driftSig = gd.next(GS(5), ln, n);
          vPitch = vPitch + (bendSig * pitchBendScale) * (1.f / 12);         // :244
          vPitch = vPitch + (driftSig * a.s.driftAmount) * 0.02f;           // kDriftScale, :247
        }
        vPitch = withMain(vPitch);
        // k is a loop variable here (the body is large): insert with selects instead of a dynamic register index
        oPitch = {k == 0 ? vPitch : oPitch[0], k == 1 ? vPitch : oPitch[1], k == 2 ? vPitch : oPitch[2], k == 3 ? vPitch : oPitch[3]};
        oGate = {k == 0 ? vGate : oGate[0], k == 1 ? vGate : oGate[1], k == 2 ? vGate : oGate[2], k == 3 ? vGate : oGate[3]};
        oTime = {k == 0 ? vTime : oTime[0], k == 1 ? vTime : oTime[1], k == 2 ? vTime : oTime[2], k == 3 ? vTime : oTime[3]};
      }
      put(0, q, oPitch);
      put(1, q, oGate);
      put(7, q, oTime);
    }
    }
    if (on)
    {
      gb.endVector();
      gd.endVector();
      gb.store(GS(0), ln);
      gd.store(GS(5), ln);
    }
    cursor = vend;
  }

  if (!live) return;
  // Recompute the state addresses from a value the optimiser cannot relate to the loads at the top: otherwise ~60 64-bit
  // per-lane pointers stay live across the whole kernel (it needed more than 256 VGPRs).
  {
    size_t Lend = L;
    asm volatile("" : "+v"(Lend));
    S = a.state + Lend;
    if (recRange.x != recRange.y) a.recRange[Lend] = make_uint2(0u, 0u);  // consumed: the next launch finds "no records" unless told otherwise
  }
  SW(S_AWAKE) = awake ? 1u : 0u;
  SW(S_VELOCITY) = f2u(velocity); SW(S_PITCH) = f2u(pitch); SW(S_BEND) = f2u(bend); SW(S_MOD) = f2u(mod);
  SW(S_X) = f2u(cx); SW(S_Y) = f2u(cy); SW(S_Z) = f2u(cz); SW(S_CHANPRESS) = f2u(chanPress);
  SW(S_AGE) = age; SW(S_AGE_STEP) = ageStep; SW(S_INHIBIT_GLIDE) = inhibit ? 1u : 0u; SW(S_RECALC) = needsRecalc ? 1u : 0u;
  SW(S_PG_CURR) = f2u(pgCurr); SW(S_PG_STEP) = f2u(pgStep); SW(S_PG_TARGET) = f2u(pgTarget); SW(S_PG_REMAINING) = (uint32_t)pgRemaining;
  SW(S_PG_PER_GLIDE) = (uint32_t)pgPerGlide; SW(S_PG_DY) = f2u(pgDy);
  SW(S_DRIFT_SEED) = driftSeed; SW(S_DRIFT_COUNTER) = (uint32_t)driftCounter; SW(S_DRIFT_VALUE) = f2u(driftValue); SW(S_DRIFT_NEXT) = (uint32_t)driftNext;
#undef GS
#undef SW
}

// ---- the control-rate half of EventsToSignals for voice graphs (mldsp_events.hpp: CtlVoice is the audio-rate half) ----------------
// One lane per voice (MIDI protocol). Per DSPVector: the records of the vector, Voice::beginProcess (:75-126) with the drift random
// walk, the bend glide's start of a vector, and a decision: does anything move INSIDE this vector apart from the drift glide? If
// not - no note event, the pitch glide at rest, the bend held: all but a few per cent of the vectors - the vector is four words
// (pitch before drift, gate, the drift glide's input, flags) and costs no per-sample work at all. If so, the lane walks the 64
// frames with note_frame() as e2s_kernel does and writes its pitch-before-drift and gate to the side signals (16 quads each);
// a portamento in progress with nothing else going on is the pitch glide's 64 steps alone (pitch side signal only). The drift glide is not touched here: its slots belong to the
// voice kernel, and nothing in this kernel depends on them (the drift term is added last, :247).
// the lanes that have records in the coming launch: {lane, first, one past the last}
__global__ void set_rec_ranges_kernel(const uint4* list, size_t n, uint2* ranges)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) ranges[list[i].x] = make_uint2(list[i].y, list[i].z);
}

// ... and the same lanes back to "no records": what the consuming kernel does itself, for a block whose consuming kernel was never
// launched (a failure between set_rec_ranges_kernel and that launch: abandonRanges)
__global__ void clear_rec_ranges_kernel(const uint4* list, size_t n, uint2* ranges)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) ranges[list[i].x] = make_uint2(0u, 0u);
}

struct E2SCtlArgs
{
  uint32_t* state;
  const Rec* recs;
  uint2* recRange;
  uint32_t* ctl;   // [T][kCtlRecWords][lanes]
  float4* rowP;    // QUAD [16 T][lanes][4], written for CF_ROWS vectors only
  float4* rowG;
  size_t lanes, T;
  uint32_t flags;
  E2SSettings s;
};

__global__ __launch_bounds__(256) void e2s_ctl_kernel(const E2SCtlArgs a)
{
  apply_fp_mode(a.flags);
  size_t blk = blockIdx.x;
  const size_t nbFull = (size_t)gridDim.x & ~(size_t)7;
  if (blk < nbFull) blk = (blk & 7) * (nbFull >> 3) + (blk >> 3);
  const size_t lane = blk * 256 + threadIdx.x;
  if (lane >= a.lanes) return;
  uint32_t* S = a.state + lane;
  const size_t ln = a.lanes;
#define SW(i) S[(size_t)(i) * ln]
  bool awake = SW(S_AWAKE) != 0;
  float velocity = u2f(SW(S_VELOCITY)), pitch = u2f(SW(S_PITCH)), bend = u2f(SW(S_BEND)), cz = u2f(SW(S_Z));
  uint32_t age = SW(S_AGE), ageStep = SW(S_AGE_STEP);
  bool inhibit = SW(S_INHIBIT_GLIDE) != 0, needsRecalc = SW(S_RECALC) != 0;
  float pgCurr = u2f(SW(S_PG_CURR)), pgStep = u2f(SW(S_PG_STEP)), pgTarget = u2f(SW(S_PG_TARGET)), pgDy = u2f(SW(S_PG_DY));
  int32_t pgRemaining = (int32_t)SW(S_PG_REMAINING), pgPerGlide = (int32_t)SW(S_PG_PER_GLIDE);
  uint32_t driftSeed = SW(S_DRIFT_SEED);
  int32_t driftCounter = (int32_t)SW(S_DRIFT_COUNTER), driftNext = (int32_t)SW(S_DRIFT_NEXT);
  float driftValue = u2f(SW(S_DRIFT_VALUE));
  // controller values that only pass through (their rows are not computed by this form): kept current as mlgpu_events_set_wanted_rows does
  uint32_t* GB = S + (size_t)(S_GLIDES + 0 * kGlideWords) * ln;
  Glide gb;
  gb.load(GB, ln);

  auto setPitchGlideTime = [&](int32_t t) {  // SampleAccurateLinearGlide::setGlideTimeInSamples, MLDSPGens.h:527-532
    pgPerGlide = t < 1 ? 1 : t;
    pgDy = 1.0f / (float)pgPerGlide;
  };
  auto pitchGlideNext = [&](float f) {  // nextSample, :541-580
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
  };

  const uint2 recRange = a.recRange[lane];  // this lane's records of this launch: [x, y)
  uint32_t cursor = recRange.x;
  const uint32_t recEnd = recRange.y;
  // the vector of this lane's next record, in a register: a lane with a record later in the launch does not ask memory every vector
  uint32_t nextVec = cursor < recEnd ? a.recs[cursor].vec : 0xFFFFFFFFu;
  const float pitchBendScale = a.s.pitchBendRange;  // MIDI protocol, :417-423
  uint32_t* rec = a.ctl + lane;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  for (size_t t = 0; t < a.T; ++t, rec += (size_t)kCtlRecWords * ln)
  {
    uint32_t vend = cursor;
    if (nextVec == (uint32_t)t)
      while (vend < recEnd && a.recs[vend].vec == (uint32_t)t) ++vend;
    if (!awake)
      for (uint32_t r = cursor; r < vend; ++r)
        if ((a.recs[r].typeTimeFlags & 0xFF) == REC_AWAKE) awake = true;
    const bool on = awake;
    bool noteHere = false;
    if (on)
    {
      // ---- Voice::beginProcess, :75-126 ----
      if (needsRecalc)
      {
        if (!inhibit) setPitchGlideTime(a.s.pitchGlideSamples);
        needsRecalc = false;
      }
      driftCounter += MLGPU_FLOATS_PER_DSPVECTOR;
      if (driftCounter >= driftNext)
      {
        driftSeed = driftSeed * 0x0019660Du + 0x3C6EF35Fu;  // RandomScalarSource::getFloat, MLDSPScalarMath.h:189-202
        const float d = u2f(((driftSeed >> 9) & 0x007FFFFFu) | 0x3F800000u) * 2.f - 3.f;
        driftSeed = driftSeed * 0x0019660Du + 0x3C6EF35Fu;
        const float d2 = u2f(((driftSeed >> 9) & 0x007FFFFFu) | 0x3F800000u) * 2.f - 3.f;
        const float nextTimeMul = 1.0f + abs_ps(d2);
        driftValue = d;
        driftCounter = 0;
        driftNext = (int32_t)(a.s.sr * (double)nextTimeMul * (double)8.0f);
      }
      // ---- values that only matter at the end of the vector (endProcess, :218-247); the rows this form does not compute keep their
      //      values in memory ----
      float finalVelocity = velocity;
      for (uint32_t r = cursor; r < vend; ++r)
      {
        const Rec rc = a.recs[r];
        switch (rc.typeTimeFlags & 0xFF)
        {
          case REC_SET_BEND: bend = rc.v1; break;
          case REC_SET_MOD: SW(S_MOD) = f2u(rc.v1); break;
          case REC_SET_X: SW(S_X) = f2u(rc.v1); break;
          case REC_SET_Y: SW(S_Y) = f2u(rc.v1); break;
          case REC_SET_Z: cz = rc.v1; break;
          case REC_SET_CHANNEL_PRESSURE: SW(S_CHANPRESS) = f2u(rc.v1); break;
          case REC_NOTE_ON: case REC_NOTE_RETRIG: finalVelocity = rc.v2; noteHere = true; break;
          case REC_NOTE_OFF: finalVelocity = 0.f; noteHere = true; break;
          default: break;
        }
      }
      if (finalVelocity == 0.f) cz = 0.f;  // :238-241
      gb.beginVector(GB, ln, bend, a.s.glideVectors, a.s.glideDy);
    }
    const int bm = gb.mode();
    const bool heldB = (bm == 1) || (bm == 0 && gb.isUniform());
    const bool pgBusy = on && (pitch != pgTarget || pgRemaining >= 0);
    const bool walk = on && (noteHere || !heldB);
    const bool glideOnly = on && !walk && pgBusy;  // a portamento in progress, nothing else: the pitch glide's 64 steps, the gate held
    const float hvB = (bm == 1) ? gb.target : gb.uniformValue;
    float P = 0.f, gate = 0.f;
    if (!walk)
    {
      if (on)
      {
        gate = velocity;
        age += (uint32_t)MLGPU_FLOATS_PER_DSPVECTOR * ageStep;
        const float bendTerm = (hvB * pitchBendScale) * (1.f / 12);  // :244
        if (!glideOnly) P = pgCurr + bendTerm;  // the same for all 64 frames
        else
        {
          f32x4* oP = (f32x4*)a.rowP + (t * 16) * ln + lane;
          // in the middle of a glide - 64 or more steps to go, the target unchanged - every frame of the vector is nextSample's last
          // branch (:571-576): mCurr += mStep. 64 dependent adds instead of 64 trips through the state machine.
          if (pitch == pgTarget && pgRemaining >= MLGPU_FLOATS_PER_DSPVECTOR && pgRemaining < pgPerGlide)
          {
#pragma unroll 4
            for (int q = 0; q < 16; ++q, oP += ln)
            {
              f32x4 vP;
#pragma unroll
              for (int k = 0; k < 4; ++k)
              {
                pgCurr += pgStep;
                vP[k] = pgCurr + bendTerm;
              }
              __builtin_nontemporal_store(vP, oP);
            }
            pgRemaining -= MLGPU_FLOATS_PER_DSPVECTOR;
          }
          else
#pragma unroll 1
          for (int q = 0; q < 16; ++q, oP += ln)
          {
            f32x4 vP;
#pragma unroll
            for (int k = 0; k < 4; ++k) vP[k] = pitchGlideNext(pitch) + bendTerm;
            __builtin_nontemporal_store(vP, oP);
          }
        }
      }
    }
    else
    {
      // ---- gate and pitch frame by frame: writeNoteEvent (:115-216) and endProcess (:218-244) ----
      // note_frame() is the whole state machine of a frame (~150 instructions); most frames of a vector with a note event need
      // none of it: until the frame before the next note record's own frame (where a retrigger opens its one-frame gap, and
      // where the rewrite pattern at the end of note_frame looks ahead) a frame is the held gate, one step of the pitch glide
      // and one of the event age - what note_frame does on such a frame, given that the pending record's bookkeeping (which
      // applies from the first frame that sees the record) has been done. plan() does that bookkeeping and says how far.
      uint32_t nc = cursor;
      bool preApplied = false;
      RecCache recCache;
      auto plan = [&](int from) -> int {
        while (nc < vend)
        {
          const Rec rc = recCache.at(a.recs, nc);
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
              setPitchGlideTime((flags & 1) ? a.s.pitchGlideSamples : 0);
            }
            preApplied = true;
          }
          if (type == REC_NOTE_RETRIG && dest == 0) dest = 1;
          return dest - 1 > from ? dest - 1 : from;
        }
        return 64;
      };
      note_rewind(a.recs, nc, vend, velocity, pitch, age, ageStep, inhibit, a.s.pitchGlideSamples, setPitchGlideTime, pitchGlideNext);
      int quietUntil = heldB ? plan(0) : 0;  // frames [n, quietUntil) need no state machine (a moving bend: every frame does)
      f32x4* oP = (f32x4*)a.rowP + (t * 16) * ln + lane;
      f32x4* oG = (f32x4*)a.rowG + (t * 16) * ln + lane;
      const float bendTerm = (hvB * pitchBendScale) * (1.f / 12);
#pragma unroll 1
      for (int q = 0; q < 16; ++q, oP += ln, oG += ln)
      {
        f32x4 vP = {0.f, 0.f, 0.f, 0.f}, vG = vP;
#pragma unroll 1
        for (int k = 0; k < 4; ++k)
        {
          const int n = q * 4 + k;
          float vPitch = 0.f, vGate = 0.f, vTime = 0.f;
          if (n < quietUntil)
          {
            vGate = velocity;
            vPitch = pitchGlideNext(pitch) + bendTerm;
            age += ageStep;
          }
          else
          {
            note_frame(a.recs, recCache, nc, vend, n, preApplied, velocity, pitch, age, ageStep, inhibit, a.s.pitchGlideSamples, false, 1.0, setPitchGlideTime,
                       pitchGlideNext, vPitch, vGate, vTime);
            const float bendSig = gb.next(GB, ln, n);
            vPitch = vPitch + (bendSig * pitchBendScale) * (1.f / 12);  // :244
            if (heldB) quietUntil = plan(n + 1);
          }
          vP = f32x4{k == 0 ? vPitch : vP[0], k == 1 ? vPitch : vP[1], k == 2 ? vPitch : vP[2], k == 3 ? vPitch : vP[3]};
          vG = f32x4{k == 0 ? vGate : vG[0], k == 1 ? vGate : vG[1], k == 2 ? vGate : vG[2], k == 3 ? vGate : vG[3]};
        }
        __builtin_nontemporal_store(vP, oP);
        __builtin_nontemporal_store(vG, oG);
      }
    }
    rec[0] = f2u(P);
    rec[ln] = f2u(gate);
    rec[2 * ln] = f2u(driftValue);
    rec[3 * ln] = (on ? CF_ON : 0u) | ((walk || glideOnly) ? CF_ROWS : 0u) | (walk ? CF_GATE_ROW : 0u);
    if (on) gb.endVector();
    if (vend != cursor)
    {
      cursor = vend;
      nextVec = cursor < recEnd ? a.recs[cursor].vec : 0xFFFFFFFFu;
    }
  }
  {
    size_t Lend = lane;
    asm volatile("" : "+v"(Lend));
    S = a.state + Lend;
    if (recRange.x != recRange.y) a.recRange[Lend] = make_uint2(0u, 0u);  // consumed
  }
  gb.store(S + (size_t)(S_GLIDES + 0 * kGlideWords) * ln, ln);
  SW(S_AWAKE) = awake ? 1u : 0u;
  SW(S_VELOCITY) = f2u(velocity); SW(S_PITCH) = f2u(pitch); SW(S_BEND) = f2u(bend); SW(S_Z) = f2u(cz);
  SW(S_AGE) = age; SW(S_AGE_STEP) = ageStep; SW(S_INHIBIT_GLIDE) = inhibit ? 1u : 0u; SW(S_RECALC) = needsRecalc ? 1u : 0u;
  SW(S_PG_CURR) = f2u(pgCurr); SW(S_PG_STEP) = f2u(pgStep); SW(S_PG_TARGET) = f2u(pgTarget); SW(S_PG_REMAINING) = (uint32_t)pgRemaining;
  SW(S_PG_PER_GLIDE) = (uint32_t)pgPerGlide; SW(S_PG_DY) = f2u(pgDy);
  SW(S_DRIFT_SEED) = driftSeed; SW(S_DRIFT_COUNTER) = (uint32_t)driftCounter; SW(S_DRIFT_VALUE) = f2u(driftValue); SW(S_DRIFT_NEXT) = (uint32_t)driftNext;
#undef SW
}

// ---- host: the event routing of EventsToSignals ------------------------------------------------------------------------
constexpr int kMaxVoices = 16;        // EventsToSignals::kMaxVoices, MLEventsToSignals.h:48
constexpr int kMaxPhysicalKeys = 128;
// ---- smoothed controller signals (SmoothedController, MLEventsToSignals.h:170-180, .cpp:264-281; read by a process function
// through AudioContext::getInputController, MLAudioContext.cpp:129) ----------------------------------------------------------
// One signal per instrument per WATCHED controller number (mlgpu_events_watch_controllers): one lane per (slot, instrument).
// Per DSPVector of an awake instrument: output = glide(inputValue), inputValue = the value of the last controller event of
// that vector or before (:744, :431-436). The records are (vector, value) pairs; lanes without records just keep gliding.
struct CtlRec
{
  uint32_t vecKind;  // vector index inside this launch << 1 | kind (0: inputValue = value, 1: the instrument woke up)
  float value;
};
enum : int { C_AWAKE = 0, C_INPUT, C_GLIDE, kCtlWords = C_GLIDE + kGlideWords };
struct CtlArgs
{
  uint32_t* state;           // [kCtlWords][lanes]
  const CtlRec* recs;
  const uint32_t* recStart;  // [lanes + 1]
  float* out;                // [slot][16 maxVectors][nInstruments][4]: slot s is a QUAD signal of nInstruments voices
  size_t nInstruments, lanes, T, slotStride;
  int32_t glideVectors;
  float glideDy;
};
__global__ __launch_bounds__(256) void ctl_kernel(const CtlArgs a)
{
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const size_t lane = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (lane >= a.lanes) return;
  const size_t slot = lane / a.nInstruments, inst = lane - slot * a.nInstruments, ln = a.lanes;
  uint32_t* st = a.state + lane;
  uint32_t* gs = st + (size_t)C_GLIDE * ln;
  bool awake = st[(size_t)C_AWAKE * ln] != 0;
  float input = u2f(st[(size_t)C_INPUT * ln]);
  uint32_t r = a.recStart[lane];
  const uint32_t rend = a.recStart[lane + 1];
  f32x4* out = (f32x4*)(a.out + slot * a.slotStride) + inst;
  Glide gl;
  gl.load(gs, ln);
  for (size_t t = 0; t < a.T; ++t)
  {
    for (; r < rend && (a.recs[r].vecKind >> 1) == (uint32_t)t; ++r)
    {
      if (a.recs[r].vecKind & 1u) awake = true;
      else input = a.recs[r].value;
    }
    f32x4* o = out + t * 16 * a.nInstruments;
    if (!awake)  // processVector returns before anything is computed (:386): the outputs keep their initial zeros
    {
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      for (int q = 0; q < 16; ++q) o[(size_t)q * a.nInstruments] = z;
      continue;
    }
    gl.beginVector(gs, ln, input, a.glideVectors, a.glideDy);
    const int m = gl.mode();
    if ((m == 1) || (m == 0 && gl.isUniform()))  // not moving: one value for the whole vector
    {
      const float c = (m == 1) ? gl.target : gl.uniformValue;
      const f32x4 v = {c, c, c, c};
      for (int q = 0; q < 16; ++q) o[(size_t)q * a.nInstruments] = v;
    }
    else
    {
      // mCurrVec[n] of the previous vector feeds sample n only: all 64 loads go out together (there are few controller lanes -
      // one wavefront per SIMD - so nothing else would hide 16 round trips in a row)
      float cur[64];
      if (gl.readsCurrVec())
      {
#pragma unroll
        for (int n = 0; n < 64; ++n) cur[n] = u2f(gs[(size_t)(5 + n) * ln]);
      }
#pragma unroll
      for (int q = 0; q < 16; ++q)
      {
        f32x4 v;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = gl.nextWith(gs, ln, q * 4 + k, cur[q * 4 + k]);
        o[(size_t)q * a.nInstruments] = v;
      }
    }
    gl.endVector();
  }
  gl.store(gs, ln);
  st[(size_t)C_AWAKE * ln] = awake ? 1u : 0u;
  st[(size_t)C_INPUT * ln] = f2u(input);
}


constexpr int kNumControllers = 129;
constexpr int kChannelPressureControllerIdx = 128;

struct KeyState
{
  int state{0};  // 0 off, 1 on, 2 sustained
  float pitch{0.f};
  uint32_t noteOnIndex{0};
};
struct HostVoice
{
  size_t creatorKeyIdx{0};
  float currentVelocity{0.f};
  int nextFrame{0};  // Voice::nextFrameToProcess inside the vector being routed (0 at its start: beginProcess, :115)
};
struct Instrument
{
  std::vector<mlgpu_event> events;  // time-sorted (addEvent, :367-372)
  KeyState keys[kMaxPhysicalKeys];
  HostVoice voices[kMaxVoices + 1];
  int lastFreeVoiceFound{-1};
  int newestVoice{-1};
  bool sustainPedal{false};
  uint32_t currentNoteOnIndex{0};
  bool awake{false}, awakeSent{false};
  std::vector<float> ctlInput;  // controllers[n].inputValue (:744), from the first controller event on
};
}  // namespace

struct mlgpu_events
{
  mlgpu_engine* e{nullptr};
  size_t nInstruments{0};
  int polyphony{0}, group{1}, slotBase{1};  // MIDI: one lane per playing voice; MPE: pow2 groups with the main voice at lane 0
  size_t maxLanes{0};
  bool mpe{false}, unison{false};
  int voiceModCC{16};
  double sr{0};
  float pitchBendRange{7.f}, mpePitchBendRange{24.f}, pitchGlideSeconds{0.f}, driftAmount{0.f};
  std::vector<Instrument> inst;
  std::vector<std::vector<Rec>> laneRecs;  // per lane, this launch
  std::vector<uint32_t> dirtyLanes;        // lanes with records (most have none)
  uint32_t* d_state{nullptr};
  // [lanes]: where a lane's records of the coming launch are, {0, 0} for none. Only the lanes that have records are written per
  // launch (a few hundred of 262 144: the list goes up in kilobytes where a start offset per lane was a megabyte over PCIe
  // every block), and a kernel that consumed a lane's records puts the {0, 0} back.
  uint2* d_recRange{nullptr};
  // Two sets of upload buffers (pinned host + device): the records of launch k + 1 are routed and copied while the kernel of
  // launch k still runs; a set is reused only after the launch that read it has finished (its event).
  struct Staging
  {
    Rec* h_recs{nullptr};
    Rec* d_recs{nullptr};
    uint4* h_dirty{nullptr};   // {lane, first record, one past the last, 0} for every lane that has records in this launch
    uint4* d_dirty{nullptr};
    size_t recCapacity{0}, dirtyCapacity{0};
    size_t nDirtySet{0};       // lanes whose range set_rec_ranges_kernel has set for the block in flight and no kernel has consumed yet
    hipEvent_t done{nullptr};
    bool pending{false};
  } stage[2];
  int stageIdx{0};
  // watched controllers (mlgpu_events_watch_controllers): lane = slot * nInstruments + instrument
  std::vector<int> watched;
  int slotOf[kNumControllers];
  size_t ctlMaxVectors{0}, ctlCapacityVectors{0};  // longest launch allowed / what d_ctlOut was allocated for
  float* d_ctlOut{nullptr};
  uint32_t* d_ctlState{nullptr};
  std::vector<std::vector<CtlRec>> ctlLaneRecs;
  std::vector<uint32_t> ctlDirty;
  struct CtlStaging
  {
    CtlRec* h_recs{nullptr};
    CtlRec* d_recs{nullptr};
    uint32_t* h_recStart{nullptr};
    uint32_t* d_recStart{nullptr};
    size_t recCapacity{0};
  } ctlStage[2];
  size_t ctlLanes() const { return watched.size() * nInstruments; }
  void pushCtl(size_t instrument, int slot, uint32_t vec, uint32_t kind, float value)
  {
    const size_t l = (size_t)slot * nInstruments + instrument;
    if (ctlLaneRecs[l].empty()) ctlDirty.push_back((uint32_t)l);
    ctlLaneRecs[l].push_back(CtlRec{(vec << 1) | kind, value});
  }
  uint32_t rowMask{0xFFu};                 // mlgpu_events_set_wanted_rows
  // e2s_ctl_kernel's outputs (mlgpu_events_prepare_for_graph): control records [T][kCtlRecWords][lanes] and the two side signals
  uint32_t* d_ctlRecs{nullptr};
  float* d_rowP{nullptr};
  float* d_rowG{nullptr};
  size_t ctlRecVectors{0};
  bool ctlRecReserved{false};  // mlgpu_events_reserve_for_graph was called: process calls never allocate, longer blocks are refused
  size_t lanes() const { return nInstruments * (size_t)group; }
};

void mlgpu_graph_forget_events(mlgpu_events* ev);  // graph.hip

namespace
{
int efail(mlgpu_events* ev, int st, const std::string& what)
{
  if (ev && ev->e) ev->e->lastError = what;
  return st;
}
bool soonerThan(const mlgpu_event& a, const mlgpu_event& b)  // :356-364
{
  if (a.time != b.time) return a.time < b.time;
  return a.type < b.type;
}
int keyIndex(const mlgpu_events* ev, const mlgpu_event& e) { return ev->mpe ? e.channel : e.source_idx; }  // getKeyIndex, :21-42

struct Router  // one instrument, one vector
{
  mlgpu_events* ev;
  Instrument& in;
  size_t instIdx;
  uint32_t vec;
  void push(int voice, const Rec& r)
  {
    if (voice < 0 || voice > ev->polyphony) return;  // voices the device does not simulate (beyond the polyphony)
    if (voice < ev->slotBase) return;  // MIDI mode: the MPE main voice is not simulated (its signals are not used, :437-445)
    std::vector<Rec>& lr = ev->laneRecs[instIdx * (size_t)ev->group + (size_t)(voice - ev->slotBase)];
    if (lr.empty()) ev->dirtyLanes.push_back((uint32_t)(instIdx * (size_t)ev->group + (size_t)(voice - ev->slotBase)));
    lr.push_back(r);
  }
  // Voice::writeNoteEvent's host-visible effects (:115-216): creatorKeyIdx_ and currentVelocity
  void note(int v, const mlgpu_event& e, uint32_t type, int keyIdx, bool doGlide, bool doReset)
  {
    HostVoice& hv = in.voices[v];
    // writeNoteEvent ends with nextFrameToProcess = its own frame (:141, :204) - also when that lies BEFORE the frame the voice's
    // previous note event of this vector ended on. Events come sorted by time, so only an event the reference makes up itself can do
    // that: the note-off of a sustain-pedal release, built with Event's default time 0 (:833-836). The frames written so far are then
    // written again by what follows; the record says so (REC_FLAG_REWIND) and the kernels replay it (mlev::note_rewind).
    int dest = std::min(std::max((int)e.time, 0), MLGPU_FLOATS_PER_DSPVECTOR);
    if (type == MLGPU_EVENT_NOTE_RETRIG && dest == 0) dest = 1;
    const uint32_t rewind = (type == MLGPU_EVENT_NOTE_ON || type == MLGPU_EVENT_NOTE_RETRIG || type == MLGPU_EVENT_NOTE_OFF) && dest == 0 && hv.nextFrame > 0 ? (uint32_t)REC_FLAG_REWIND : 0u;
    if (type == MLGPU_EVENT_NOTE_ON || type == MLGPU_EVENT_NOTE_RETRIG)
    {
      hv.creatorKeyIdx = (size_t)keyIdx;
      hv.currentVelocity = e.value2;
      hv.nextFrame = dest;
      push(v, makeRec(vec, type == MLGPU_EVENT_NOTE_ON ? REC_NOTE_ON : REC_NOTE_RETRIG, e.time, (doGlide ? 1u : 0u) | (doReset ? 2u : 0u) | rewind, e.value1, e.value2));
    }
    else if (type == MLGPU_EVENT_NOTE_OFF)
    {
      hv.creatorKeyIdx = 0;
      hv.currentVelocity = 0.f;
      hv.nextFrame = dest;
      push(v, makeRec(vec, REC_NOTE_OFF, e.time, rewind, 0.f, 0.f));
    }
    // kNoteSustain and everything else: no change (default:, :211-213)
  }
  size_t countHeldNotes() const  // :471-482
  {
    size_t n = 0;
    for (int i = 0; i < kMaxPhysicalKeys; ++i) n += (in.keys[i].state == 1);
    return n;
  }
  int findFreeVoice()  // :892-912
  {
    const int highest = ev->polyphony + 1;
    int t = in.lastFreeVoiceFound;
    for (int i = 1; i < ev->polyphony + 1; ++i)
    {
      t++;
      if (t >= highest) t = 1;
      if (in.voices[t].creatorKeyIdx == 0)
      {
        in.lastFreeVoiceFound = t;
        return t;
      }
    }
    return -1;
  }
  int findNearestVoice(int note)  // :922-937
  {
    int r = 0;
    size_t minDist = 128;
    for (int v = 1; v < ev->polyphony + 1; ++v)
    {
      const size_t dist = (size_t)std::abs(note - (int)in.voices[v].creatorKeyIdx);
      if (dist < minDist)
      {
        minDist = dist;
        r = v;
      }
    }
    return r;
  }
  void noteOn(const mlgpu_event& e)  // :519-560
  {
    const int k = keyIndex(ev, e) & (kMaxPhysicalKeys - 1);
    in.keys[k].state = 1;
    in.keys[k].noteOnIndex = in.currentNoteOnIndex++;
    in.keys[k].pitch = e.value1;
    if (ev->unison)
    {
      const bool firstNote = (countHeldNotes() == 1);
      for (int v = 1; v < ev->polyphony + 1; ++v) note(v, e, MLGPU_EVENT_NOTE_ON, k, !firstNote, firstNote);
    }
    else
    {
      int v = findFreeVoice();
      if (v >= 1) note(v, e, MLGPU_EVENT_NOTE_ON, k, true, true);
      else
      {
        v = findNearestVoice(e.source_idx);  // findVoiceToSteal, :914-918
        note(v, e, MLGPU_EVENT_NOTE_RETRIG, k, true, true);
      }
      in.newestVoice = v;
    }
  }
  void noteOff(const mlgpu_event& e)  // :562-632
  {
    const int k = keyIndex(ev, e) & (kMaxPhysicalKeys - 1);
    in.keys[k].state = in.sustainPedal ? 2 : 0;
    if (ev->unison)
    {
      if (countHeldNotes() == 0)
      {
        for (int v = 1; v < ev->polyphony + 1; ++v) note(v, e, MLGPU_EVENT_NOTE_OFF, 0, true, true);
      }
      else if ((size_t)k == in.voices[1].creatorKeyIdx)
      {
        mlgpu_event f = e;  // change note without retriggering the envelope, keeping the current velocity
        f.value2 = in.voices[1].currentVelocity;
        uint32_t maxIdx = 0, mostRecent = 0;
        for (int i = 0; i < kMaxPhysicalKeys; ++i)
          if (in.keys[i].state == 1 && in.keys[i].noteOnIndex > maxIdx)
          {
            maxIdx = in.keys[i].noteOnIndex;
            mostRecent = (uint32_t)i;
          }
        f.value1 = in.keys[mostRecent].pitch;
        for (int v = 1; v < ev->polyphony + 1; ++v) note(v, f, MLGPU_EVENT_NOTE_ON, (int)mostRecent, true, true);
      }
    }
    else if (!in.sustainPedal)
    {
      for (int v = 1; v < ev->polyphony + 1; ++v)
        if (in.voices[v].creatorKeyIdx == (size_t)k) note(v, e, MLGPU_EVENT_NOTE_OFF, k, true, true);
    }
  }
  void setAll(uint32_t rec, float val)
  {
    for (int v = 1; v < ev->polyphony + 1; ++v) push(v, makeRec(vec, rec, 0, 0, val, 0.f));
  }
  void setMatching(uint32_t rec, int channel, float val)
  {
    for (int v = 1; v < ev->polyphony + 1; ++v)
      if (in.voices[v].creatorKeyIdx == (size_t)channel) push(v, makeRec(vec, rec, 0, 0, val, 0.f));
  }
  void setControllerInput(size_t ctrl, float val)  // controllers[ctrl].inputValue = val (:650, :744)
  {
    if (in.ctlInput.empty()) in.ctlInput.assign(kNumControllers, 0.f);  // kept from the first controller event on, watched or not
    in.ctlInput[ctrl] = val;
    if (ev->slotOf[ctrl] >= 0) ev->pushCtl(instIdx, ev->slotOf[ctrl], vec, 0u, val);
  }
  void controller(const mlgpu_event& e)  // :735-822
  {
    const float val = e.value1;
    const size_t ctrl = std::min((size_t)e.source_idx, (size_t)kNumControllers - 1);
    setControllerInput(ctrl, val);
    if (ctrl == kChannelPressureControllerIdx)  // controllers[128].inputValue is what MIDI channel pressure writes too
      for (int v = 0; v < ev->polyphony + 1; ++v) push(v, makeRec(vec, REC_SET_CHANNEL_PRESSURE, 0, 0, val, 0.f));
    if (ctrl == 120) return;  // "all sound off" clears the event buffer it is iterating in the reference (:749-755): not reproduced
    if (ctrl == 123)
    {
      if (val == 0)  // all notes off, :757-769
        for (int v = 0; v < kMaxVoices + 1; ++v) note(v, e, MLGPU_EVENT_NOTE_OFF, 0, false, true);
      return;
    }
    for (int v = 1; v < ev->polyphony + 1; ++v)
    {
      if (ev->mpe && in.voices[v].creatorKeyIdx != (size_t)e.channel) continue;
      if ((int)ctrl == ev->voiceModCC) push(v, makeRec(vec, REC_SET_MOD, 0, 0, val, 0.f));
      if (ctrl == 73) push(v, makeRec(vec, REC_SET_X, 0, 0, val, 0.f));
      else if (ctrl == 74) push(v, makeRec(vec, REC_SET_Y, 0, 0, val, 0.f));
    }
  }
  void process(const mlgpu_event& e)  // processEvent, :485-515
  {
    switch (e.type)
    {
      case MLGPU_EVENT_NOTE_ON: noteOn(e); break;
      case MLGPU_EVENT_NOTE_OFF: noteOff(e); break;
      case MLGPU_EVENT_CONTROLLER: controller(e); break;
      case MLGPU_EVENT_PITCH_BEND:  // :700-731
        if (!ev->mpe) setAll(REC_SET_BEND, e.value1);
        else if (e.channel == 1) push(0, makeRec(vec, REC_SET_BEND, 0, 0, e.value1, 0.f));
        else if (e.channel != 0) setMatching(REC_SET_BEND, e.channel, e.value1);
        break;
      case MLGPU_EVENT_NOTE_PRESSURE:  // :676-698: per-key pressure in MIDI mode, ignored in MPE mode
        if (!ev->mpe) setMatching(REC_SET_Z, e.source_idx, e.value1);
        break;
      case MLGPU_EVENT_CHANNEL_PRESSURE:  // :637-674
        if (!ev->mpe)
        {
          setControllerInput(kChannelPressureControllerIdx, e.value1);
          for (int v = 0; v < ev->polyphony + 1; ++v) push(v, makeRec(vec, REC_SET_CHANNEL_PRESSURE, 0, 0, e.value1, 0.f));
        }
        else if (e.channel == 1) push(0, makeRec(vec, REC_SET_Z, 0, 0, e.value1, 0.f));
        else if (e.channel != 0) setMatching(REC_SET_Z, e.channel, e.value1);
        break;
      case MLGPU_EVENT_SUSTAIN_PEDAL:  // :824-842
        in.sustainPedal = (e.value1 > 0.5f);
        if (!in.sustainPedal)
          for (int i = 1; i < ev->polyphony + 1; ++i)
            if (in.keys[in.voices[i].creatorKeyIdx & (kMaxPhysicalKeys - 1)].state == 2)
            {
              mlgpu_event off{};
              off.type = MLGPU_EVENT_NOTE_OFF;
              note(i, off, MLGPU_EVENT_NOTE_OFF, 0, true, true);
            }
        break;
      default: break;
    }
  }
};

// the state of freshly constructed / reset voices (EventsToSignals ctor :290-305, Voice::reset :58-84)
void initialState(const mlgpu_events* ev, std::vector<uint32_t>& st)
{
  const size_t lanes = ev->lanes();
  st.assign((size_t)kStateWords * lanes, 0u);
  auto W = [&](int word, size_t lane) -> uint32_t& { return st[(size_t)word * lanes + lane]; };
  const uint32_t minusOne = 0xFFFFFFFFu;
  for (size_t lane = 0; lane < lanes; ++lane)
  {
    const int slot = (int)(lane % (size_t)ev->group) + ev->slotBase;
    W(S_PG_REMAINING, lane) = minusOne;       // SampleAccurateLinearGlide defaults, MLDSPGens.h:519-524
    W(S_PG_PER_GLIDE, lane) = 32;
    const float dy = 1.f / 32;
    memcpy(&W(S_PG_DY, lane), &dy, 4);
    W(S_DRIFT_SEED, lane) = (uint32_t)(slot * 232);  // driftSource.seed_ = voiceIndex * 232, :60
    W(S_RECALC, lane) = 1u;
    for (int gl = 0; gl < kNumGlides; ++gl)
    {
      const int base = S_GLIDES + gl * kGlideWords;
      // reset() calls setValue(0) on bend / mod / x / y / z: remaining = 0; the drift and controller glides are
      // default-constructed: remaining = -1 (MLDSPGens.h:441)
      W(base + 2, lane) = (gl <= 4) ? 0u : minusOne;
      W(base + 3, lane) = 1u;  // mCurrVec is all zeros: uniform
    }
  }
}
}  // namespace

static void freeControllers(mlgpu_events* ev)
{
  if (ev->d_ctlOut) hipFree(ev->d_ctlOut);
  if (ev->d_ctlState) hipFree(ev->d_ctlState);
  ev->d_ctlOut = nullptr;
  ev->d_ctlState = nullptr;
  for (mlgpu_events::CtlStaging& st : ev->ctlStage)
  {
    if (st.h_recs) hipHostFree(st.h_recs);
    if (st.d_recs) hipFree(st.d_recs);
    if (st.h_recStart) hipHostFree(st.h_recStart);
    if (st.d_recStart) hipFree(st.d_recStart);
    st = mlgpu_events::CtlStaging();
  }
  ev->watched.clear();
  ev->ctlLaneRecs.clear();
  ev->ctlDirty.clear();
  ev->ctlMaxVectors = 0;
  for (int& x : ev->slotOf) x = -1;
}
extern "C"
{
  // What a recorded sequence may still read is DEVICE memory only (event routing is refused while recording, so no replay ever
  // touches the pinned staging or the hipEvents): the host side goes at once, the device buffers when no sequence can replay.
  static void freeEventsHostSide(mlgpu_events* ev)
  {
    for (mlgpu_events::Staging& st : ev->stage)
    {
      if (st.h_recs) hipHostFree(st.h_recs);
      if (st.h_dirty) hipHostFree(st.h_dirty);
      if (st.done) hipEventDestroy(st.done);
      st.h_recs = nullptr;
      st.h_dirty = nullptr;
      st.done = nullptr;
    }
    for (mlgpu_events::CtlStaging& st : ev->ctlStage)
    {
      if (st.h_recs) hipHostFree(st.h_recs);
      if (st.h_recStart) hipHostFree(st.h_recStart);
      st.h_recs = nullptr;
      st.h_recStart = nullptr;
    }
    std::vector<Instrument>().swap(ev->inst);
    std::vector<std::vector<Rec>>().swap(ev->laneRecs);
    std::vector<std::vector<CtlRec>>().swap(ev->ctlLaneRecs);
  }
  static void freeEventsDeviceSide(mlgpu_events* ev)
  {
    if (ev->d_state) hipFree(ev->d_state);
    if (ev->d_ctlRecs) hipFree(ev->d_ctlRecs);
    if (ev->d_rowP) hipFree(ev->d_rowP);
    if (ev->d_rowG) hipFree(ev->d_rowG);
    if (ev->d_recRange) hipFree(ev->d_recRange);
    freeControllers(ev);
    for (mlgpu_events::Staging& st : ev->stage)
    {
      if (st.d_recs) hipFree(st.d_recs);
      if (st.d_dirty) hipFree(st.d_dirty);
    }
    delete ev;
  }
  int mlgpu_events_destroy(mlgpu_events* ev)
  {
    if (!ev) return MLGPU_ERR_INVALID;
    // waiting for the stream would invalidate a capture in progress
    if (ev->e->recording) return efail(ev, MLGPU_ERR_INVALID, "events_destroy waits for the device: not while recording a sequence");
    mlgpu_graph_forget_events(ev);  // graphs bound to this object (mlgpu_graph_bind_events) go back to "no events object"
    hipSetDevice(ev->e->device);
    hipStreamSynchronize(ev->e->stream);
    freeEventsHostSide(ev);
    // A recorded sequence of this engine may replay launches that read this object's device memory: the handle is gone for the
    // caller now, that memory goes when the last sequence does (or with the engine). A host that keeps one long-lived sequence and
    // churns events objects so holds on to their device buffers only - state and signals -, not to pinned memory and events.
    if (ev->e->liveSequences > 0)
    {
      ev->e->deferredFrees.push_back([ev]() { freeEventsDeviceSide(ev); });
      return MLGPU_OK;
    }
    freeEventsDeviceSide(ev);
    return MLGPU_OK;
  }

  int mlgpu_events_clear(mlgpu_events* ev)  // EventsToSignals::clear, :330-340
  {
    if (!ev) return MLGPU_ERR_INVALID;
    for (Instrument& in : ev->inst)
    {
      in.events.clear();
      for (HostVoice& v : in.voices) v = HostVoice();
      in.lastFreeVoiceFound = 0;
    }
    // Voice::reset keeps the glides' and the drift's running state except what setValue(0) touches; a freshly built
    // bank and a cleared one differ only there. This implementation resets the device state completely.
    std::vector<uint32_t> st;
    initialState(ev, st);
    if (hipSetDevice(ev->e->device) != hipSuccess) return efail(ev, MLGPU_ERR_HIP, "hipSetDevice");
    return mlgpu_upload(ev->e, ev->d_state, st.data(), st.size() * sizeof(uint32_t));
  }

  int mlgpu_events_create(mlgpu_engine* e, size_t nInstruments, int polyphony, mlgpu_events** out)
  {
    if (!e || !out) return MLGPU_ERR_INVALID;
    *out = nullptr;
    if (nInstruments == 0 || polyphony < 1 || polyphony > kMaxVoices)
    {
      e->lastError = "events_create: 1+ instruments, polyphony 1..16 (EventsToSignals::kMaxVoices)";
      return MLGPU_ERR_INVALID;
    }
    mlgpu_events* ev = new (std::nothrow) mlgpu_events();
    if (!ev) return MLGPU_ERR_OOM;
    ev->e = e;
    for (int& x : ev->slotOf) x = -1;
    ev->nInstruments = nInstruments;
    ev->polyphony = polyphony;
    int pow2 = 1;
    while (pow2 < polyphony + 1) pow2 <<= 1;
    ev->maxLanes = nInstruments * (size_t)pow2;
    ev->group = polyphony;  // MIDI (the default protocol)
    ev->slotBase = 1;
    ev->inst.resize(nInstruments);
    ev->laneRecs.resize(ev->maxLanes);
    hipError_t err = hipSetDevice(e->device);
    if (err == hipSuccess) err = hipMalloc((void**)&ev->d_state, sizeof(uint32_t) * (size_t)kStateWords * ev->maxLanes);
    if (err == hipSuccess) err = hipMalloc((void**)&ev->d_recRange, sizeof(uint2) * ev->maxLanes);
    if (err == hipSuccess) err = hipMemsetAsync(ev->d_recRange, 0, sizeof(uint2) * ev->maxLanes, e->stream);
    for (mlgpu_events::Staging& st : ev->stage)
      if (err == hipSuccess) err = hipEventCreateWithFlags(&st.done, hipEventDisableTiming);
    if (err != hipSuccess)
    {
      e->lastError = std::string("events_create: ") + hipGetErrorString(err);
      mlgpu_events_destroy(ev);
      return err == hipErrorOutOfMemory ? MLGPU_ERR_OOM : MLGPU_ERR_HIP;
    }
    for (Instrument& in : ev->inst) in.lastFreeVoiceFound = -1;
    const int st = mlgpu_events_clear(ev);
    for (Instrument& in : ev->inst) in.lastFreeVoiceFound = 0;  // setPolyphony calls clear() (:316-321)
    if (st != MLGPU_OK)
    {
      mlgpu_events_destroy(ev);
      return st;
    }
    *out = ev;
    return MLGPU_OK;
  }

  static int markRecalc(mlgpu_events* ev)
  {
    if (hipSetDevice(ev->e->device) != hipSuccess) return efail(ev, MLGPU_ERR_HIP, "hipSetDevice");
    return mlgpu_fill32(ev->e, ev->d_state + (size_t)S_RECALC * ev->lanes(), 1u, ev->lanes());
  }
  int mlgpu_events_set_sample_rate(mlgpu_events* ev, double sr)
  {
    if (!ev) return MLGPU_ERR_INVALID;
    ev->sr = sr;
    return markRecalc(ev);
  }
  int mlgpu_events_set_protocol(mlgpu_events* ev, int mpe)  // setProtocol clears (:92-96)
  {
    if (!ev) return MLGPU_ERR_INVALID;
    ev->mpe = mpe != 0;
    if (ev->mpe)
    {
      ev->group = 1;
      while (ev->group < ev->polyphony + 1) ev->group <<= 1;
      ev->slotBase = 0;
    }
    else
    {
      ev->group = ev->polyphony;
      ev->slotBase = 1;
    }
    return mlgpu_events_clear(ev);
  }
  int mlgpu_events_set_unison(mlgpu_events* ev, int on) { return ev ? (ev->unison = on != 0, MLGPU_OK) : MLGPU_ERR_INVALID; }
  int mlgpu_events_set_mod_cc(mlgpu_events* ev, int cc) { return ev ? (ev->voiceModCC = cc, MLGPU_OK) : MLGPU_ERR_INVALID; }
  int mlgpu_events_set_pitch_bend_semitones(mlgpu_events* ev, float f) { return ev ? (ev->pitchBendRange = f, MLGPU_OK) : MLGPU_ERR_INVALID; }
  int mlgpu_events_set_mpe_pitch_bend_semitones(mlgpu_events* ev, float f) { return ev ? (ev->mpePitchBendRange = f, MLGPU_OK) : MLGPU_ERR_INVALID; }
  int mlgpu_events_set_drift_amount(mlgpu_events* ev, float f) { return ev ? (ev->driftAmount = f, MLGPU_OK) : MLGPU_ERR_INVALID; }
  int mlgpu_events_set_pitch_glide_seconds(mlgpu_events* ev, float f)
  {
    if (!ev) return MLGPU_ERR_INVALID;
    ev->pitchGlideSeconds = f;
    return markRecalc(ev);
  }
  int mlgpu_events_set_wanted_rows(mlgpu_events* ev, unsigned mask)
  {
    if (!ev) return MLGPU_ERR_INVALID;
    ev->rowMask = mask & 0xFFu;
    return MLGPU_OK;
  }
  size_t mlgpu_events_num_voices(mlgpu_events* ev) { return ev ? ev->nInstruments * (size_t)ev->polyphony : 0; }
  int mlgpu_events_newest_voice(mlgpu_events* ev, size_t instrument) { return (ev && instrument < ev->nInstruments) ? ev->inst[instrument].newestVoice - 1 : -2; }

  int mlgpu_events_add_event(mlgpu_events* ev, size_t instrument, const mlgpu_event* e)  // addEvent, :367-372
  {
    if (!ev || !e) return MLGPU_ERR_INVALID;
    if (instrument >= ev->nInstruments) return efail(ev, MLGPU_ERR_RANGE, "events_add_event: instrument out of range");
    Instrument& in = ev->inst[instrument];
    in.awake = true;
    in.events.insert(std::lower_bound(in.events.begin(), in.events.end(), *e, soonerThan), *e);
    return MLGPU_OK;
  }
  int mlgpu_events_add_events(mlgpu_events* ev, const uint32_t* instruments, const mlgpu_event* events, size_t n)  // a block's events in one call
  {
    if (!ev || (n && (!instruments || !events))) return MLGPU_ERR_INVALID;
    for (size_t i = 0; i < n; ++i)
      if (instruments[i] >= ev->nInstruments) return efail(ev, MLGPU_ERR_RANGE, "events_add_events: instrument out of range");
    for (size_t i = 0; i < n; ++i)
    {
      Instrument& in = ev->inst[instruments[i]];
      in.awake = true;
      in.events.insert(std::lower_bound(in.events.begin(), in.events.end(), events[i], soonerThan), events[i]);
    }
    return MLGPU_OK;
  }
  int mlgpu_events_clear_events(mlgpu_events* ev)  // clearEvents, once per host block (MLSignalProcessBuffer.cpp:89)
  {
    if (!ev) return MLGPU_ERR_INVALID;
    for (Instrument& in : ev->inst) in.events.clear();
    return MLGPU_OK;
  }

  // The controller lanes of one launch: their records uploaded into the staging set of this launch (free once the launch
  // before last has finished, which prepare() has just waited for), then ctl_kernel on the engine's stream - ahead of the
  // kernel that reads the signals.
  static int processControllers(mlgpu_events* ev, size_t nVectors, int stageIdx)
  {
    mlgpu_engine* e = ev->e;
    mlgpu_events::CtlStaging& sg = ev->ctlStage[stageIdx];
    const size_t lanes = ev->ctlLanes();
    size_t nRecs = 0;
    for (uint32_t l : ev->ctlDirty) nRecs += ev->ctlLaneRecs[l].size();
    if (nRecs + 1 > sg.recCapacity)
    {
      if (sg.h_recs) hipHostFree(sg.h_recs);
      if (sg.d_recs) hipFree(sg.d_recs);
      sg.h_recs = sg.d_recs = nullptr;
      sg.recCapacity = std::max<size_t>(1024, 2 * (nRecs + 1));
      if (hipMalloc((void**)&sg.d_recs, sizeof(CtlRec) * sg.recCapacity) != hipSuccess || hipHostMalloc((void**)&sg.h_recs, sizeof(CtlRec) * sg.recCapacity) != hipSuccess)
      {
        sg.recCapacity = 0;
        return efail(ev, MLGPU_ERR_OOM, "events_process: controller record buffer");
      }
    }
    std::sort(ev->ctlDirty.begin(), ev->ctlDirty.end());
    size_t next = 0, n = 0;
    for (uint32_t l : ev->ctlDirty)
    {
      for (; next <= l; ++next) sg.h_recStart[next] = (uint32_t)n;
      const std::vector<CtlRec>& lr = ev->ctlLaneRecs[l];
      memcpy(sg.h_recs + n, lr.data(), sizeof(CtlRec) * lr.size());
      n += lr.size();
    }
    for (; next <= lanes; ++next) sg.h_recStart[next] = (uint32_t)n;
    hipError_t err = hipMemcpyAsync(sg.d_recStart, sg.h_recStart, sizeof(uint32_t) * (lanes + 1), hipMemcpyHostToDevice, e->stream);
    if (err == hipSuccess && nRecs) err = hipMemcpyAsync(sg.d_recs, sg.h_recs, sizeof(CtlRec) * nRecs, hipMemcpyHostToDevice, e->stream);
    if (err != hipSuccess) return efail(ev, MLGPU_ERR_HIP, std::string("events_process controller upload: ") + hipGetErrorString(err));
    CtlArgs a;
    a.state = ev->d_ctlState;
    a.recs = sg.d_recs;
    a.recStart = sg.d_recStart;
    a.out = ev->d_ctlOut;
    a.nInstruments = ev->nInstruments;
    a.lanes = lanes;
    a.T = nVectors;
    a.slotStride = 64 * ev->ctlCapacityVectors * ev->nInstruments;  // (the capacity, not the current limit: a slot's signal stays where it is while the buffer is not replaced)
    float c[2];
    mlgpu_linear_glide_make_coeffs((float)(int)(ev->sr * 0.02f), c);  // int glideTimeInSamples = sr * kControllerGlideTimeSeconds (:275)
    memcpy(&a.glideVectors, &c[0], 4);
    a.glideDy = c[1];
    hipLaunchKernelGGL(ctl_kernel, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, e->stream, a);
    err = hipGetLastError();
    if (err != hipSuccess) return efail(ev, MLGPU_ERR_HIP, std::string("events_process controller launch: ") + hipGetErrorString(err));
    return MLGPU_OK;
  }

  // Everything of processVector (:376-466) that happens on the host for nVectors DSPVectors starting at frame startOffset of the
  // event times: the block's events routed into per-voice records, the records uploaded (asynchronously, into the staging set
  // that is free), the settings the device needs. The caller launches the kernel that consumes them - e2s_kernel, or a voice
  // graph whose pitch and gate rows are source nodes (graph.hip) - and then calls launched().
  // The per-lane record ranges are device state: set from the block's lane list, cleared by the kernel that consumes them. A block that
  // fails after they were set and before that kernel ran would leave them pointing into a staging buffer the NEXT block does not
  // use: put them back to "no records" (stream-ordered, the list is still in sg.d_dirty).
  static void abandonRanges(mlgpu_events* ev, mlgpu_events::Staging& sg)
  {
    if (!sg.nDirtySet) return;
    hipLaunchKernelGGL(clear_rec_ranges_kernel, dim3((unsigned)((sg.nDirtySet + 255) / 256)), dim3(256), 0, ev->e->stream, (const uint4*)sg.d_dirty, sg.nDirtySet, ev->d_recRange);
    (void)hipGetLastError();
    sg.nDirtySet = 0;
    // (the list must outlive the clearing pass: the next use of this staging buffer waits for `done`)
    if (hipEventRecord(sg.done, ev->e->stream) == hipSuccess) sg.pending = true;
  }

  static int prepare(mlgpu_events* ev, size_t nVectors, int startOffset, EventsDev& dev, mlgpu_events::Staging*& sgOut)
  {
    mlgpu_engine* e = ev->e;
    // ---- route this launch's events into per-voice records ----
    for (uint32_t l : ev->dirtyLanes) ev->laneRecs[l].clear();
    ev->dirtyLanes.clear();
    for (uint32_t l : ev->ctlDirty) ev->ctlLaneRecs[l].clear();
    ev->ctlDirty.clear();
    if (!ev->watched.empty() && nVectors > ev->ctlMaxVectors)
      return efail(ev, MLGPU_ERR_RANGE, "events_process: more DSPVectors than events_watch_controllers reserved the controller signals for");
    for (size_t i = 0; i < ev->nInstruments; ++i)
    {
      Instrument& in = ev->inst[i];
      if (!in.awake) continue;
      if (in.events.empty() && in.awakeSent) continue;  // nothing to route: the voices just keep gliding on the device
      for (size_t t = 0; t < nVectors; ++t)
      {
        Router r{ev, in, i, (uint32_t)t};
        for (int v = 0; v < kMaxVoices + 1; ++v) in.voices[v].nextFrame = 0;
        if (!in.awakeSent)
        {
          for (int v = 0; v < ev->polyphony + 1; ++v) r.push(v, makeRec((uint32_t)t, REC_AWAKE, 0, 0, 0.f, 0.f));
          for (size_t sl = 0; sl < ev->watched.size(); ++sl) ev->pushCtl(i, (int)sl, (uint32_t)t, 1u, 0.f);
          in.awakeSent = true;
        }
        const int start = startOffset + (int)t * MLGPU_FLOATS_PER_DSPVECTOR, end = start + MLGPU_FLOATS_PER_DSPVECTOR;
        for (const mlgpu_event& evt : in.events)
          if (evt.time >= start && evt.time < end)
          {
            mlgpu_event local = evt;
            local.time -= start;
            r.process(local);
          }
      }
    }
    const size_t lanes = ev->lanes();
    if (hipSetDevice(e->device) != hipSuccess) return efail(ev, MLGPU_ERR_HIP, "hipSetDevice");
    mlgpu_events::Staging& sg = ev->stage[ev->stageIdx];
    sgOut = &sg;
    ev->stageIdx ^= 1;
    if (sg.pending && hipEventSynchronize(sg.done) != hipSuccess) return efail(ev, MLGPU_ERR_HIP, "events_process: waiting for the launch before last");
    sg.pending = false;
    size_t nRecs = 0;
    for (uint32_t l : ev->dirtyLanes) nRecs += ev->laneRecs[l].size();
    if (nRecs + 1 > sg.recCapacity)
    {
      if (sg.h_recs) hipHostFree(sg.h_recs);
      if (sg.d_recs) hipFree(sg.d_recs);
      sg.h_recs = sg.d_recs = nullptr;
      sg.recCapacity = std::max<size_t>(4096, 2 * (nRecs + 1));
      if (hipMalloc((void**)&sg.d_recs, sizeof(Rec) * sg.recCapacity) != hipSuccess || hipHostMalloc((void**)&sg.h_recs, sizeof(Rec) * sg.recCapacity) != hipSuccess)
      {
        sg.recCapacity = 0;
        return efail(ev, MLGPU_ERR_OOM, "events_process: record buffer");
      }
    }
    const size_t nDirty = ev->dirtyLanes.size();
    if (nDirty > sg.dirtyCapacity)
    {
      if (sg.h_dirty) hipHostFree(sg.h_dirty);
      if (sg.d_dirty) hipFree(sg.d_dirty);
      sg.h_dirty = sg.d_dirty = nullptr;
      sg.dirtyCapacity = std::max<size_t>(1024, 2 * nDirty);
      if (hipMalloc((void**)&sg.d_dirty, sizeof(uint4) * sg.dirtyCapacity) != hipSuccess || hipHostMalloc((void**)&sg.h_dirty, sizeof(uint4) * sg.dirtyCapacity) != hipSuccess)
      {
        sg.dirtyCapacity = 0;
        return efail(ev, MLGPU_ERR_OOM, "events_process: lane list");
      }
    }
    std::sort(ev->dirtyLanes.begin(), ev->dirtyLanes.end());
    {
      size_t n = 0, i = 0;
      for (uint32_t l : ev->dirtyLanes)
      {
        const std::vector<Rec>& lr = ev->laneRecs[l];
        memcpy(sg.h_recs + n, lr.data(), sizeof(Rec) * lr.size());
        sg.h_dirty[i++] = make_uint4(l, (uint32_t)n, (uint32_t)(n + lr.size()), 0u);
        n += lr.size();
      }
    }
    hipError_t cerr = hipSuccess;
    if (nDirty) cerr = hipMemcpyAsync(sg.d_dirty, sg.h_dirty, sizeof(uint4) * nDirty, hipMemcpyHostToDevice, e->stream);
    if (cerr == hipSuccess && nRecs) cerr = hipMemcpyAsync(sg.d_recs, sg.h_recs, sizeof(Rec) * nRecs, hipMemcpyHostToDevice, e->stream);
    if (cerr == hipSuccess && nDirty)
    {
      hipLaunchKernelGGL(set_rec_ranges_kernel, dim3((unsigned)((nDirty + 255) / 256)), dim3(256), 0, e->stream, (const uint4*)sg.d_dirty, nDirty, ev->d_recRange);
      cerr = hipGetLastError();
      if (cerr == hipSuccess) sg.nDirtySet = nDirty;
    }
    if (cerr != hipSuccess) return efail(ev, MLGPU_ERR_HIP, std::string("events_process upload: ") + hipGetErrorString(cerr));

    const int cst = ev->watched.empty() ? MLGPU_OK : processControllers(ev, nVectors, ev->stageIdx ^ 1);
    if (cst != MLGPU_OK)
    {
      abandonRanges(ev, sg);
      return cst;
    }

    memset(&dev.s, 0, sizeof(dev.s));
    dev.s.sr = ev->sr;
    dev.s.pitchBendRange = ev->pitchBendRange;
    dev.s.mpePitchBendRange = ev->mpePitchBendRange;
    dev.s.driftAmount = ev->driftAmount;
    dev.s.pitchGlideSamples = (int32_t)(ev->sr * ev->pitchGlideSeconds);  // :90
    float c[2];
    mlgpu_linear_glide_make_coeffs((float)(ev->sr * 0.02f), c);          // kGlideTimeSeconds / kControllerGlideTimeSeconds
    memcpy(&dev.s.glideVectors, &c[0], 4);
    dev.s.glideDy = c[1];
    mlgpu_linear_glide_make_coeffs((float)(ev->sr * 8.0f), c);           // kDriftTimeSeconds
    memcpy(&dev.s.driftGlideVectors, &c[0], 4);
    dev.s.driftGlideDy = c[1];
    mlgpu_linear_glide_make_coeffs((float)(int)(ev->sr * 0.02f), c);    // SmoothedController: int glideTimeInSamples = sr * 0.02f
    memcpy(&dev.s.ctlGlideVectors, &c[0], 4);
    dev.s.ctlGlideDy = c[1];
    dev.s.mpe = ev->mpe ? 1 : 0;
    dev.ctl = nullptr;
    dev.rowP = dev.rowG = nullptr;
    dev.state = ev->d_state;
    dev.recs = sg.d_recs;
    dev.recRange = ev->d_recRange;
    dev.lanes = lanes;
    return MLGPU_OK;
  }
  static int launched(mlgpu_events* ev, mlgpu_events::Staging& sg)
  {
    sg.nDirtySet = 0;  // (the consuming kernel clears the ranges it read)
    if (hipEventRecord(sg.done, ev->e->stream) != hipSuccess) return efail(ev, MLGPU_ERR_HIP, "events_process: event");
    sg.pending = true;  // no wait here: the host goes on routing the next block while this one runs
    return MLGPU_OK;
  }

  // processVector (:376-466) for n_vectors consecutive DSPVectors starting at frame start_offset of the event times.
  int mlgpu_events_process(mlgpu_events* ev, size_t nVectors, int startOffset, float* const* d_outputs, int layout)
  {
    if (!ev || !d_outputs) return MLGPU_ERR_INVALID;
    mlgpu_engine* e = ev->e;
    if (nVectors == 0) return MLGPU_OK;
    if (e->recording) return efail(ev, MLGPU_ERR_INVALID, "events_process routes events on the host: not while recording a sequence");
    if (ev->sr == 0) return efail(ev, MLGPU_ERR_INVALID, "events_process: no sample rate (the reference does nothing, :385)");
    if (layout < 0 || layout > MLGPU_LAYOUT_VOICE_MAJOR) return efail(ev, MLGPU_ERR_INVALID, "events_process: bad layout");
    // everything that can be refused is refused BEFORE the router consumes the block's note events: after this point the
    // host-side voice allocator (keys, creatorKeyIdx, sustain pedal, lastFreeVoiceFound) has advanced, and an error return
    // (allocation, upload, launch) leaves host and device voice state out of step - reset the object with
    // mlgpu_events_set_protocol / a fresh mlgpu_events if that ever happens.
    for (int r = 0; r < 8; ++r)
    {
      if (d_outputs[r] && ((uintptr_t)d_outputs[r] & 15)) return efail(ev, MLGPU_ERR_INVALID, "events_process: misaligned output");
      if (d_outputs[r] && !((ev->rowMask >> r) & 1u)) return efail(ev, MLGPU_ERR_INVALID, "events_process: an output was passed for a row outside events_set_wanted_rows");
    }
    EventsDev dev;
    mlgpu_events::Staging* sg = nullptr;
    const int pst = prepare(ev, nVectors, startOffset, dev, sg);
    if (pst != MLGPU_OK) return pst;
    E2SArgs a;
    memset(&a, 0, sizeof(a));
    a.state = dev.state;
    a.recs = (const Rec*)dev.recs;
    a.recRange = dev.recRange;
    const size_t lanes = dev.lanes;
    const size_t V = ev->nInstruments * (size_t)ev->polyphony;
    for (int r = 0; r < 8; ++r) a.out[r] = makeView(d_outputs[r], layout, V, nVectors);
    a.lanes = lanes;
    a.T = nVectors;
    a.rowMask = ev->rowMask;
    a.flags = e->kflags;
    static const bool noBlocks = getenv("MLGPU_E2S_NO_BLOCKS") != nullptr;
    a.blockPath = noBlocks ? 0 : 1;
    a.group = ev->group;
    a.slotBase = ev->slotBase;
    a.polyphony = ev->polyphony;
    a.s = dev.s;
    if ((a.rowMask & ~3u) == 0)
      hipLaunchKernelGGL(e2s_kernel<true>, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, e->stream, a);
    else
      hipLaunchKernelGGL(e2s_kernel<false>, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, e->stream, a);
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess)
    {
      abandonRanges(ev, *sg);
      return efail(ev, MLGPU_ERR_HIP, std::string("events_process launch: ") + hipGetErrorString(err));
    }
    return launched(ev, *sg);
  }

  // control records [T][kCtlRecWords][lanes] + the two side signals [64 T][lanes] of e2s_ctl_kernel for blocks of up to nVectors
  static int reserveCtlRecs(mlgpu_events* ev, size_t nVectors)
  {
    mlgpu_engine* e = ev->e;
    const size_t lanes = ev->lanes();
    if (hipSetDevice(e->device) != hipSuccess) return efail(ev, MLGPU_ERR_HIP, "hipSetDevice");
    hipStreamSynchronize(e->stream);
    hipFree(ev->d_ctlRecs);
    hipFree(ev->d_rowP);
    hipFree(ev->d_rowG);
    ev->d_ctlRecs = nullptr;
    ev->d_rowP = ev->d_rowG = nullptr;
    ev->ctlRecVectors = 0;
    const size_t rowBytes = sizeof(float) * 64 * nVectors * lanes;
    if (hipMalloc((void**)&ev->d_ctlRecs, sizeof(uint32_t) * kCtlRecWords * nVectors * lanes) != hipSuccess || hipMalloc((void**)&ev->d_rowP, rowBytes) != hipSuccess ||
        hipMalloc((void**)&ev->d_rowG, rowBytes) != hipSuccess)
    {
      hipFree(ev->d_ctlRecs);
      hipFree(ev->d_rowP);
      hipFree(ev->d_rowG);
      ev->d_ctlRecs = nullptr;
      ev->d_rowP = ev->d_rowG = nullptr;
      return efail(ev, MLGPU_ERR_OOM, "events as graph source nodes: control records and side signals");
    }
    ev->ctlRecVectors = nVectors;
    return MLGPU_OK;
  }
  int mlgpu_events_reserve_for_graph(mlgpu_events* ev, size_t maxVectors)
  {
    if (!ev || maxVectors == 0) return MLGPU_ERR_INVALID;
    if (ev->e->recording) return efail(ev, MLGPU_ERR_INVALID, "events_reserve_for_graph allocates: not while recording a sequence");
    if (ev->mpe) return efail(ev, MLGPU_ERR_UNSUPPORTED, "events as graph source nodes: MIDI protocol only (one lane per voice)");
    if (maxVectors != ev->ctlRecVectors)
    {
      const int st = reserveCtlRecs(ev, maxVectors);
      if (st != MLGPU_OK) return st;
    }
    ev->ctlRecReserved = true;
    return MLGPU_OK;
  }
  size_t mlgpu_events_graph_reserve_bytes(mlgpu_events* ev, size_t maxVectors)
  {
    if (!ev) return 0;
    return (sizeof(uint32_t) * kCtlRecWords + 2 * sizeof(float) * 64) * maxVectors * ev->lanes();
  }

  // for graph.hip: a graph whose event rows are bound to this object (mlgpu_graph_bind_events)
  int mlgpu_events_prepare_for_graph(mlgpu_events* ev, size_t nVectors, int startOffset, EventsDev* dev, void** staging)
  {
    if (!ev || !dev || !staging) return MLGPU_ERR_INVALID;
    if (nVectors == 0) return MLGPU_OK;
    if (ev->e->recording) return efail(ev, MLGPU_ERR_INVALID, "events route on the host: not while recording a sequence");
    if (ev->sr == 0) return efail(ev, MLGPU_ERR_INVALID, "events: no sample rate (the reference does nothing, :385)");
    if (ev->mpe) return efail(ev, MLGPU_ERR_UNSUPPORTED, "events as graph source nodes: MIDI protocol only (one lane per voice)");
    mlgpu_engine* e = ev->e;
    const size_t lanes = ev->lanes();
    // the control records and the two side signals of e2s_ctl_kernel: sized by mlgpu_events_reserve_for_graph at setup. An object that
    // was never reserved grows them here on the first / a longer block (a setup-time convenience: it waits for the stream and allocates);
    // once a host has reserved, a longer block is refused before the router consumes its events and nothing is ever allocated here.
    if (nVectors > ev->ctlRecVectors)
    {
      if (ev->ctlRecReserved)
        return efail(ev, MLGPU_ERR_RANGE, "graph_process_events: more DSPVectors than mlgpu_events_reserve_for_graph reserved the control records for");
      const int rst = reserveCtlRecs(ev, nVectors);
      if (rst != MLGPU_OK) return rst;
    }
    mlgpu_events::Staging* sg = nullptr;
    const int st = prepare(ev, nVectors, startOffset, *dev, sg);
    *staging = sg;
    if (st != MLGPU_OK) return st;
    E2SCtlArgs a;
    memset(&a, 0, sizeof(a));
    a.state = dev->state;
    a.recs = (const Rec*)dev->recs;
    a.recRange = dev->recRange;
    a.ctl = ev->d_ctlRecs;
    a.rowP = (float4*)ev->d_rowP;
    a.rowG = (float4*)ev->d_rowG;
    a.lanes = lanes;
    a.T = nVectors;
    a.flags = e->kflags;
    a.s = dev->s;
    hipLaunchKernelGGL(e2s_ctl_kernel, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, e->stream, a);
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess)
    {
      abandonRanges(ev, *sg);
      return efail(ev, MLGPU_ERR_HIP, std::string("events control kernel launch: ") + hipGetErrorString(err));
    }
    dev->ctl = ev->d_ctlRecs;
    dev->rowP = (const float4*)ev->d_rowP;
    dev->rowG = (const float4*)ev->d_rowG;
    return MLGPU_OK;
  }
  // the graph's voice kernel (which consumes the ranges) could not be launched after prepare_for_graph had succeeded
  int mlgpu_events_abandoned_by_graph(mlgpu_events* ev, void* staging)
  {
    if (!ev || !staging) return MLGPU_ERR_INVALID;
    abandonRanges(ev, *(mlgpu_events::Staging*)staging);
    return MLGPU_OK;
  }
  int mlgpu_events_launched_by_graph(mlgpu_events* ev, void* staging)
  {
    if (!ev || !staging) return MLGPU_ERR_INVALID;
    return launched(ev, *(mlgpu_events::Staging*)staging);
  }
  int mlgpu_events_watch_controllers(mlgpu_events* ev, const int* numbers, int n, size_t maxVectors)
  {
    if (!ev || n < 0 || (n > 0 && !numbers)) return MLGPU_ERR_INVALID;
    mlgpu_engine* e = ev->e;
    if (e->recording) return efail(ev, MLGPU_ERR_INVALID, "events_watch_controllers allocates: not while recording a sequence");
    if (n > MLGPU_EVENTS_MAX_WATCHED_CONTROLLERS) return efail(ev, MLGPU_ERR_RANGE, "events_watch_controllers: at most MLGPU_EVENTS_MAX_WATCHED_CONTROLLERS");
    if (n > 0 && maxVectors == 0) return efail(ev, MLGPU_ERR_INVALID, "events_watch_controllers: max_vectors = the longest launch, at least 1");
    for (int i = 0; i < n; ++i)
    {
      if (numbers[i] < 0 || numbers[i] >= kNumControllers) return efail(ev, MLGPU_ERR_RANGE, "events_watch_controllers: controller numbers 0..128");
      for (int j = 0; j < i; ++j)
        if (numbers[j] == numbers[i]) return efail(ev, MLGPU_ERR_INVALID, "events_watch_controllers: a controller number twice");
    }
    if (hipSetDevice(e->device) != hipSuccess) return efail(ev, MLGPU_ERR_HIP, "hipSetDevice");
    hipStreamSynchronize(e->stream);
    if (n > 0 && ev->watched == std::vector<int>(numbers, numbers + n))  // the same controllers: only the reserved length changes
    {
      if (maxVectors <= ev->ctlCapacityVectors)  // the signals' pointers (mlgpu_events_controller_signal) are only ever replaced to grow
      {
        ev->ctlMaxVectors = maxVectors;
        return MLGPU_OK;
      }
      if (e->liveSequences > 0)
        return efail(ev, MLGPU_ERR_INVALID, "events_watch_controllers would move the controller signals that recorded sequences of this engine may read: destroy them first");
      float* fresh = nullptr;
      const size_t bytes = sizeof(float) * 64 * maxVectors * ev->ctlLanes();
      if (hipMalloc((void**)&fresh, bytes) != hipSuccess) return efail(ev, MLGPU_ERR_OOM, "events_watch_controllers: controller signals");
      hipMemsetAsync(fresh, 0, bytes, e->stream);
      hipFree(ev->d_ctlOut);
      ev->d_ctlOut = fresh;
      ev->ctlMaxVectors = ev->ctlCapacityVectors = maxVectors;
      return MLGPU_OK;
    }
    if (e->liveSequences > 0 && ev->d_ctlOut)
      return efail(ev, MLGPU_ERR_INVALID, "events_watch_controllers would free controller signals that recorded sequences of this engine may read: destroy them first");
    freeControllers(ev);
    if (n == 0) return MLGPU_OK;
    ev->watched.assign(numbers, numbers + n);
    for (int i = 0; i < n; ++i) ev->slotOf[numbers[i]] = i;
    ev->ctlMaxVectors = ev->ctlCapacityVectors = maxVectors;
    const size_t lanes = ev->ctlLanes();
    ev->ctlLaneRecs.resize(lanes);
    hipError_t err = hipMalloc((void**)&ev->d_ctlOut, sizeof(float) * 64 * maxVectors * lanes);
    if (err == hipSuccess) err = hipMalloc((void**)&ev->d_ctlState, sizeof(uint32_t) * (size_t)kCtlWords * lanes);
    for (mlgpu_events::CtlStaging& st : ev->ctlStage)
    {
      if (err == hipSuccess) err = hipMalloc((void**)&st.d_recStart, sizeof(uint32_t) * (lanes + 1));
      if (err == hipSuccess) err = hipHostMalloc((void**)&st.h_recStart, sizeof(uint32_t) * (lanes + 1));
    }
    if (err == hipSuccess) err = hipMemsetAsync(ev->d_ctlOut, 0, sizeof(float) * 64 * maxVectors * lanes, e->stream);
    if (err != hipSuccess)
    {
      freeControllers(ev);
      return efail(ev, err == hipErrorOutOfMemory ? MLGPU_ERR_OOM : MLGPU_ERR_HIP, std::string("events_watch_controllers: ") + hipGetErrorString(err));
    }
    // A smoother that starts being watched now starts settled on its controller's current value (the reference's has been
    // running all along: the same thing 20 ms after the controller last moved); an instrument that has not seen an event yet
    // is asleep and gives zeros (:386).
    std::vector<uint32_t> st((size_t)kCtlWords * lanes, 0u);
    auto W = [&](int word, size_t lane) -> uint32_t& { return st[(size_t)word * lanes + lane]; };
    for (int sl = 0; sl < n; ++sl)
      for (size_t i = 0; i < ev->nInstruments; ++i)
      {
        const size_t lane = (size_t)sl * ev->nInstruments + i;
        const float v = ev->inst[i].ctlInput.empty() ? 0.f : ev->inst[i].ctlInput[(size_t)numbers[sl]];
        uint32_t bits;
        memcpy(&bits, &v, 4);
        W(C_AWAKE, lane) = ev->inst[i].awakeSent ? 1u : 0u;
        W(C_INPUT, lane) = bits;
        W(C_GLIDE + 0, lane) = bits;         // target
        W(C_GLIDE + 2, lane) = 0xFFFFFFFFu;  // remaining = -1: holding (MLDSPGens.h:441)
        W(C_GLIDE + 3, lane) = 1u;           // mCurrVec is one value
        W(C_GLIDE + 4, lane) = bits;
      }
    return mlgpu_upload(e, ev->d_ctlState, st.data(), st.size() * sizeof(uint32_t));
  }
  const float* mlgpu_events_controller_signal(mlgpu_events* ev, int slot)
  {
    if (!ev || slot < 0 || (size_t)slot >= ev->watched.size()) return nullptr;
    return ev->d_ctlOut + (size_t)slot * 64 * ev->ctlCapacityVectors * ev->nInstruments;
  }
  int mlgpu_events_is_midi(mlgpu_events* ev) { return (ev && !ev->mpe) ? 1 : 0; }
  mlgpu_engine* mlgpu_events_engine(mlgpu_events* ev) { return ev ? ev->e : nullptr; }
}
