// USER CODE #5: every free function of MLDSPOps.h that takes whole DSPVectors, called the way user code calls them - by name,
// with the reference's argument order - in one process function. Compiled unchanged against the reference (oracle/dropin_ref.cpp)
// and against include/mlgpu/compat (tests/cpp/dropin_gpu.cpp); tests/test_gpu_dropin.py feeds both the same general floats
// (infinities, NaNs, denormals, huge and tiny values included) and compares every output bit for bit. What this pins is the
// shim's wiring: which device operation a name stands for, and which argument goes where.
// Operations that look at the bits of a NaN (sign, signBit, the conversions, the bitwise select) are applied to the INPUTS only:
// a NaN made on the way carries the sign of the machine that made it (include/mlgpu.h, numerical contract).
constexpr int kOpsOutputs = 8;  // (a graph has at most eight outputs)

inline void opsProcess(AudioContext* ctx, void*)
{
  const DSPVector a = ctx->inputs[0], b = ctx->inputs[1];
  const DSPVector lo = min(a, b), hi = max(a, b);
  // 0: arithmetic by name and by operator; min / max / clamp argument order (they differ on NaN: minps returns its second operand)
  ctx->outputs[0] = (subtract(multiply(add(a, b), a), divide(b, a + DSPVector(3.f))) - (a * b) / (b - DSPVector(0.5f))) +
                    (min(a, b) + max(b, a) * 0.5f + clamp(a, DSPVector(-0.75f), b) * 0.25f);
  // 1: precise transcendentals
  ctx->outputs[1] = sin(a) + cos(b) * 0.5f + exp(clamp(a, DSPVector(-20.f), DSPVector(20.f))) * 0.001f + log(abs(b) + DSPVector(1e-3f)) * 0.01f;
  // 2: base-2 forms and pow
  ctx->outputs[2] = exp2(clamp(b, DSPVector(-30.f), DSPVector(30.f))) * 1e-4f + log2(abs(a) + DSPVector(1e-6f)) + pow(abs(a) + DSPVector(0.1f), clamp(b, DSPVector(-3.f), DSPVector(3.f))) * 0.01f;
  // 3: approximations (polynomial ones: the same bits)
  ctx->outputs[3] = sinApprox(clamp(a, DSPVector(-3.14f), DSPVector(3.14f))) + cosApprox(clamp(b, DSPVector(-3.14f), DSPVector(3.14f))) * 0.5f +
                    expApprox(clamp(a, DSPVector(-10.f), DSPVector(10.f))) * 0.01f + logApprox(abs(b) + DSPVector(0.01f)) * 0.1f +
                    exp2Approx(clamp(b, DSPVector(-10.f), DSPVector(10.f))) * 0.01f + log2Approx(abs(a) + DSPVector(0.01f)) * 0.1f +
                    powApprox(abs(a) + DSPVector(0.5f), clamp(b, DSPVector(-2.f), DSPVector(2.f))) * 0.01f;
  // 4: sqrt, abs, sign, signBit, fractionalPart on the inputs; within(x, lo, hi) - a mask in a float vector (all bits set reads as
  //    NaN: minps hands back its second operand then) - and lerp / inverseLerp
  ctx->outputs[4] = (sqrt(abs(a)) + sign(b) * 0.25f + signBit(a) * 0.125f + fractionalPart(b)) +
                    (min(within(a, DSPVector(-0.5f), DSPVector(0.5f)), DSPVector(1.f)) * 1024.f + lerp(a, b, DSPVector(0.25f)) + lerp(b, a, 0.75f) +
                    inverseLerp(DSPVector(-2.f), DSPVector(6.f), a));
  // 5: comparisons and the bitwise selects: select(a, b, mask) takes a where the mask is set; a mask used twice
  const DSPVectorInt ra = roundFloatToInt(a), tb = truncateFloatToInt(b);
  const DSPVectorInt m = greaterThan(b, a);
  ctx->outputs[5] = (select(a, b, greaterThan(a, b)) + select(DSPVector(1.f), DSPVector(2.f), lessThanOrEqual(a, b)) +
                    select(DSPVector(4.f), DSPVector(8.f), equal(lo, a)) + select(DSPVector(16.f), DSPVector(32.f), notEqual(hi, a)) +
                    select(DSPVector(64.f), DSPVector(128.f), greaterThanOrEqual(b, DSPVector(0.f))) + select(DSPVector(256.f), DSPVector(512.f), lessThan(b, a))) +
                    (intToFloat(select(ra, tb, m)) + select(b, a, m) * 0.5f) * 1024.f;
  // 6: conversions and integer arithmetic on the inputs; index vectors and row plumbing
  const DSPVectorArray<2> ab = concatRows(a, b);
  const DSPVectorArray<4> four = repeatRows<2>(ab);
  ctx->outputs[6] = (intToFloat(addInt32(ra, tb)) * 0.5f + intToFloat(subtractInt32(tb, ra)) * 0.25f + unsignedIntToFloat(ra) * 1e-9f) +
                    (columnIndex() * 0.01f + rangeOpen(-1.f, 1.f) + rangeClosed(0.f, 2.f) * 0.5f + interpolateDSPVectorLinear(3.f, 5.f) * 0.1f +
                     addRows(four) * 0.125f + rotateRows(ab, 1).row(0) * 0.5f + shiftRows(ab, 1).row(1) * 0.25f + evenRows(four).row(1) + oddRows(four).row(0));
  // 7: the hardware-approximate pair, kept apart (relative tolerance, include/mlgpu.h)
  ctx->outputs[7] = sqrtApprox(abs(a) + DSPVector(0.01f)) + divideApprox(b, abs(a) + DSPVector(1.f)) * 0.5f;
}
