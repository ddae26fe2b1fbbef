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
