// USER CODE #10: the places where the reference's DSPVector code touches single floats ON THE HOST - map() with scalar functions
// that do not look at device data (MLDSPFunctional.h:24-35, 50-60), window tables written through getBuffer()
// (MLDSPUtils.h:22-47), the overlap-add use of DSPBuffer (MLDSPBuffer.h:289-320, Tests/dspBufferTest.cpp "overlap"), and v[n] on
// vectors the host made. Compiled unchanged against the reference (oracle/dropin_ref.cpp) and against include/mlgpu/compat
// (tests/cpp/dropin_gpu.cpp); every output bit for bit.
constexpr int kHostDataOutputs = 4;

struct HostDataState
{
  DSPVector window;      // made once in the setup function, outside any process call
  DSPVector overlapSum;  // what eight half-overlapping windows add up to, read back from a DSPBuffer
  Lopass lp;
  int counter{0};
};

inline void hostDataSetup(HostDataState* s)
{
  makeWindow(s->window.getBuffer(), kFloatsPerDSPVector, dspwindows::hamming);
  DSPVector tri;
  makeWindow(tri.getBuffer(), kFloatsPerDSPVector, dspwindows::triangle);
  DSPBuffer buf;
  buf.resize(256);
  for (int i = 0; i < 8; ++i) buf.writeWithOverlapAdd(tri.getBuffer(), kFloatsPerDSPVector, kFloatsPerDSPVector / 2);
  DSPVector startup;
  buf.read(startup);
  buf.read(s->overlapSum);
  s->lp.coeffs = Lopass::makeCoeffs(0.1f, 1.5f);
}

inline void hostDataProcess(AudioContext* ctx, void* untypedState)
{
  HostDataState* s = reinterpret_cast<HostDataState*>(untypedState);
  const DSPVector x = ctx->inputs[0];
  // 0: the input under a host-made window, filtered
  ctx->outputs[0] = s->lp(x * s->window);
  // 1: map(float()): a stateful host function, evaluated in element order (a decaying table made on the spot)
  float level = 1.f;
  const DSPVector decay = map([&]() { level *= 0.95f; return level; }, DSPVector());
  ctx->outputs[1] = x * decay;
  // 2: map(float(int)) over columnIndexInt(): a host table indexed by sample position
  const DSPVector steps = map([](int i) { return (float)(i / 8) * 0.125f - 0.4375f; }, columnIndexInt());
  ctx->outputs[2] = x + steps;
  // 3: single floats of host vectors, by index
  ctx->outputs[3] = x * s->window[10] + DSPVector(s->overlapSum[20]) + s->overlapSum * steps[63];
}
