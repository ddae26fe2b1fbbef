// USER CODE #11: imperative DSPVector code - no process function, no AudioContext: objects are made, called and read on the spot,
// the way the reference's unit tests (Tests/dspOpsTest.cpp, dspGensTest.cpp, dspFiltersTest.cpp) and offline tools use the library.
// Compiled unchanged against the reference (oracle/dropin_ref.cpp: every call is CPU arithmetic) and against include/mlgpu/compat
// (tests/cpp/dropin_gpu.cpp: every call is a launch on the device, the shim's immediate mode); every recorded float bit for bit.
// Left out on purpose: sqrtApprox / divideApprox / Peak / RMS (hardware-approximate on both sides, include/mlgpu.h's contract).
#include <string>
#include <vector>

struct ImmediateLog
{
  std::vector<float> data;
  std::vector<std::string> names;
  std::vector<size_t> starts;
  void mark(const char* name)
  {
    names.push_back(name);
    starts.push_back(data.size());
  }
  template <size_t ROWS>
  void put(const DSPVectorArray<ROWS>& v)
  {
    for (size_t i = 0; i < kFloatsPerDSPVector * ROWS; ++i) data.push_back(v[i]);
  }
  void put(float f) { data.push_back(f); }
};

inline void immediateSuite(ImmediateLog& rec)
{
  // ---- elementwise ops on host-made vectors -------------------------------------------------------------------------
  const DSPVector a(rangeClosed(-kPi, kPi));
  const DSPVector pos = abs(a) + DSPVector(0.25f);
  rec.mark("index generators");
  rec.put(columnIndex());
  rec.put(rangeOpen(-1.f, 3.f));
  rec.put(a);
  rec.put(interpolateDSPVectorLinear(0.25f, -0.75f));
  rec.mark("precise and approximate transcendentals");
  rec.put(sin(a));
  rec.put(cos(a));
  rec.put(log(pos));
  rec.put(exp(a));
  rec.put(log2(pos));
  rec.put(exp2(a));
  rec.put(sinApprox(a));
  rec.put(cosApprox(a));
  rec.put(expApprox(a));
  rec.put(logApprox(pos));
  rec.put(log2Approx(pos));
  rec.put(exp2Approx(a));
  rec.put(pow(pos, DSPVector(1.5f)));
  rec.put(powApprox(pos, a * 0.5f));
  rec.mark("arithmetic, compare, select, convert");
  DSPVector b;  // written sample by sample on the host, like dspOpsTest's "native" vectors
  for (int i = 0; i < kFloatsPerDSPVector; ++i) b[i] = (float)((i * 37) % 64) * 0.03125f - 1.f;
  rec.put(a + b);
  rec.put(a - b * 2.f);
  rec.put(a / (b + 1.5f));
  rec.put(sqrt(pos));
  rec.put(min(a, b));
  rec.put(max(a, b));
  rec.put(clamp(a, DSPVector(-1.f), b + 1.f));
  rec.put(lerp(a, b, 0.3f));
  rec.put(lerp(a, b, pos * 0.25f));
  rec.put(inverseLerp(DSPVector(-4.f), DSPVector(4.f), a));
  rec.put(sign(b));
  rec.put(fractionalPart(a * 3.f));
  rec.put(select(a, b, greaterThan(a, b)));
  rec.put(select(a, b, lessThanOrEqual(b, DSPVector(0.f))));
  rec.put(select(DSPVector(1.f), DSPVector(2.f), notEqual(b, rotateLeft(b))));
  rec.put(intToFloat(roundFloatToInt(a * 10.f)));
  rec.put(intToFloat(truncateFloatToInt(a * 10.f)));
  rec.put(intToFloat(addInt32(roundFloatToInt(a * 10.f), truncateFloatToInt(b * 100.f))));
  DSPVector acc;
  acc += a;
  acc *= b;
  acc -= 0.125f;
  rec.put(acc);
  rec.mark("horizontal operators and what is built on them");
  rec.put(sum(a + 0.01f));
  rec.put(sum(b));
  rec.put(mean(pos));
  rec.put(max(b));
  rec.put(min(b));
  rec.put(max(abs(sin(a) - sinApprox(a))));
  rec.put(normalize(pos));
  rec.put(rotateLeft(b));
  rec.put(rotateRight(b));
  rec.put((float)(a == a) + 2.f * (float)(a == b));

  // ---- rows -----------------------------------------------------------------------------------------------------------
  rec.mark("row operations, map, mix, multiplex");
  const DSPVectorArray<2> two{repeatRows<2>(columnIndex())};
  const auto g = map([&](DSPVector x, int j) { return x * (j + 1) + b; }, two);
  rec.put(g);
  rec.put(stretchRows<5>(g));
  rec.put(rotateRows(zeroPadRows<3>(g), -1) * 3.f);
  rec.put(shiftRows(zeroPadRows<3>(g), 1));
  rec.put(addRows(g));
  rec.put(map([](float x) { return x * x - 3.f; }, g));
  rec.put(map([](DSPVector x) { return sinApprox(x * 0.05f); }, g));
  const DSPVectorArray<3> gains = concatRows(DSPVector{0.300f}, DSPVector{0.030f}, DSPVector{0.003f});
  rec.put(mix(gains, g, g * 2.f, g + 1.f));
  DSPVectorArray<2> m0{7}, m1{11}, m2{13}, m3{17};
  rec.put(multiplex(rangeOpen(0, 1), m0, m1, g, m3));
  rec.put(multiplexLinear(rangeClosed(0, 3.f / 4.f), m0, m1, g, m3));
  demultiplexLinear(rangeClosed(0, 3.f / 4.f), g, &m0, &m1, &m2, &m3);
  rec.put(add(m0, m1, m2, m3));
  demultiplex(rangeOpen(0, 1), g, &m0, &m1, &m2, &m3);
  rec.put(m2);

  // ---- generators: state carried from call to call ----------------------------------------------------------------------
  rec.mark("generators");
  DSPVector freq;  // a different frequency per sample, written on the host
  for (int i = 0; i < kFloatsPerDSPVector; ++i) freq[i] = 0.01f + 0.0005f * i;
  PhasorGen ph;
  ph.clear();
  SineGen sine;
  sine.clear();
  SawGen saw;
  saw.clear();
  PulseGen pulse;
  pulse.clear();
  NoiseGen noise;
  noise.setSeed(1234);
  TickGen tick;
  ImpulseGen imp;
  OneShotGen shot;
  for (int v = 0; v < 5; ++v)
  {
    rec.put(ph(freq));
    rec.put(sine(1.f / kFloatsPerDSPVector));
    rec.put(saw(freq * 3.f));
    rec.put(pulse(freq * 5.f, DSPVector(0.3f) + freq));
    rec.put(noise());
    rec.put(tick(freq));
    rec.put(imp(freq * 0.5f));
    if (v == 2) shot.trigger();
    rec.put(shot(3.f / kFloatsPerDSPVector / 4.f));
  }
  noise.step();
  rec.put(noise());
  rec.put(noise.getSample());
  rec.put(noise());
  Bank<PulseGen, 3> pulses;
  const auto bankFreqs = rowIndex<3>() * 0.01f + 0.05f;
  for (int v = 0; v < 3; ++v) rec.put(pulses(bankFreqs, rowIndex<3>() * 0.1f + 0.3f));
  Bank<NoiseGen, 2> noises;
  noises[1].setSeed(99);
  rec.put(noises());
  rec.put(noises());

  // ---- filters: coefficients changed between calls, clear(), a copy continuing from the original's state ---------------
  rec.mark("filters");
  Lopass lp;
  lp.coeffs = Lopass::makeCoeffs(0.05f, 0.7f);
  Hipass hp;
  hp.coeffs = Hipass::makeCoeffs(0.1f, 1.2f);
  Bandpass bp;
  bp.coeffs = Bandpass::makeCoeffs(0.2f, 0.4f);
  LoShelf ls;
  ls.coeffs = LoShelf::makeCoeffs({0.1f, 1.f, 2.f});
  HiShelf hs;
  hs.coeffs = HiShelf::makeCoeffs({0.2f, 1.f, 0.5f});
  Bell bell;
  bell.coeffs = Bell::makeCoeffs(0.15f, 0.5f, 3.f);
  OnePole op;
  op.coeffs = OnePole::makeCoeffs(0.05f);
  DCBlocker dc;
  dc.coeffs = DCBlocker::makeCoeffs(0.045f);
  Integrator integ;
  Differentiator diff;
  NoiseGen src;
  src.setSeed(7);
  for (int v = 0; v < 6; ++v)
  {
    const DSPVector x = src();
    if (v == 3) lp.coeffs = Lopass::makeCoeffs(0.2f, 1.9f);
    if (v == 4) lp.clear();
    rec.put(lp(x));
    rec.put(hp(x));
    rec.put(bp(x));
    rec.put(ls(x));
    rec.put(hs(x));
    rec.put(bell(x));
    rec.put(op(x));
    rec.put(dc(x + 0.5f));
    rec.put(integ(x * 0.01f));
    rec.put(diff(x));
  }
  Lopass lpCopy = lp;  // value semantics: from here on two filters with the same memory
  const DSPVector y0 = src();
  rec.put(lp(y0));
  rec.put(lpCopy(y0));
  rec.put(lpCopy(src()));
  rec.mark("per-sample coefficient forms");
  Lopass lpm;
  for (int v = 0; v < 3; ++v)
  {
    const DSPVector x = src();
    const DSPVector omega = freq * 4.f + 0.01f * v, k = DSPVector(0.5f) + freq * 10.f;
    rec.put(lpm(x, omega, k));
  }
  rec.mark("envelope and glides");
  ADSR env;
  env.coeffs = ADSR::calcCoeffs(0.002f, 0.004f, 0.6f, 0.01f, 48000.f);
  LinearGlide glide;
  glide.setGlideTimeInSamples(192.f);
  for (int v = 0; v < 8; ++v)
  {
    DSPVector gate;
    for (int i = 0; i < kFloatsPerDSPVector; ++i) gate[i] = ((v * 64 + i) > 40 && (v * 64 + i) < 300) ? 0.8f : 0.f;
    rec.put(env(gate));
    rec.put(glide(v < 3 ? 1.f : -0.5f));
  }

  // ---- delay lines and the loops built on them ----------------------------------------------------------------------------
  rec.mark("delays, allpass");  // (the reference's FDN and FeedbackDelayFunction cannot size their lines: not callable there)
  IntegerDelay idl;
  idl.setMaxDelayInSamples(300.f);
  idl.setDelayInSamples(100);
  FractionalDelay fdl;
  fdl.setMaxDelayInSamples(300.f);
  fdl.setDelayInSamples(70.35f);
  PitchbendableDelay pbd;
  pbd.setMaxDelayInSamples(400.f);
  Allpass<IntegerDelay> ap;
  ap.setMaxDelayInSamples(400.f);
  ap.setDelayInSamples(131.f);
  ap.mGain = 0.6f;
  NoiseGen burst;
  burst.setSeed(3);
  for (int v = 0; v < 8; ++v)
  {
    const DSPVector x = v < 2 ? burst() : DSPVector(0.f);
    rec.put(idl(x));
    rec.put(fdl(x));
    rec.put(pbd(x, DSPVector(120.f) + freq * 400.f));
    rec.put(ap(x));
  }

  // ---- vector-scheduled resamplers (Tests/dspFiltersTest.cpp) ------------------------------------------------------------
  rec.mark("Upsampler / Downsampler");
  constexpr int kOctaves = 2;
  Upsampler upper(kOctaves);
  Downsampler downer(kOctaves);
  SineGen tone;
  tone.clear();
  for (int v = 0; v < 5; ++v)
  {
    upper.write(tone(0.02f));
    bool ready = false;
    for (int i = 0; i < (1 << kOctaves); ++i)
    {
      const DSPVector up = upper.read();
      rec.put(up);
      ready = downer.write(up);
    }
    rec.put(ready ? 1.f : 0.f);
    rec.put(downer.read());
  }
}
