// tests/cpp/host_mirror_test.cpp — the reference's own test style (Tests/dspGensTest.cpp,
// Tests/dspOpsTest.cpp) written against the C++ host mirror include/mlgpu/mldsp_gpu.hpp.
// Runs on the GPU box (pytest -m gpu drives it); exits non-zero on the first failed REQUIRE.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "mlgpu/mldsp_gpu.hpp"

using namespace ml::gpu;

static int failures = 0;
#define REQUIRE(cond)                                                   \
  do                                                                    \
  {                                                                     \
    if (!(cond))                                                        \
    {                                                                   \
      printf("REQUIRE failed at line %d: %s\n", __LINE__, #cond);       \
      ++failures;                                                       \
    }                                                                   \
  } while (0)

static float hexf(const char* s) { return strtof(s, nullptr); }

int main()
{
  Engine engine(0);

  // "madronalib/core/dsp_gens" (Tests/dspGensTest.cpp:13-31): SineGen after clear(), one cycle at
  // 1/64 ends within -120 dB of zero.
  {
    VoiceBank<SineGen> s1(engine, 1);
    s1.clear();
    s1.input(0, 1.f / kFloatsPerDSPVector);
    DeviceSignal v1(engine, 1, 1, MLGPU_LAYOUT_ROWS);
    s1(v1);
    auto h = v1.toRows();
    const float epsilon = powf(10.f, -120.f / 20.f);
    REQUIRE(fabsf(h[kFloatsPerDSPVector - 1]) < epsilon);
  }

  // BASELINE config 1: SineGen -> Lopass, 1 voice, 1 DSPVector; SURVEY Appendix B anchor.
  {
    VoiceBank<SineGen, Lopass> bank(engine, 1);
    bank.clear();
    bank.coeffs<1>(0, Lopass::makeCoeffs(0.1f, 1.0f));
    bank.input(0, 220.f / 48000.f);
    DeviceSignal y(engine, 1, 1, MLGPU_LAYOUT_ROWS);
    bank(y);
    auto h = y.toRows();
    REQUIRE(h[0] == hexf("-0x1.09fe14p-9"));
    REQUIRE(h[1] == hexf("-0x1.5d1a5cp-7"));
    REQUIRE(h[2] == hexf("-0x1.d1fd4ap-6"));
    REQUIRE(h[3] == hexf("-0x1.bb038p-5"));
    REQUIRE(h[63] == hexf("-0x1.f1279ep-1"));
  }

  // BASELINE config 3 shape: y = bp(saw(freq)) * gain for a bank of voices; voice 0 is the
  // Appendix B anchor, the rest checks Bank semantics (every row is its own processor).
  {
    const size_t V = 1000;  // ragged: not a multiple of 64
    VoiceBank<SawGen, Bandpass, Gain> bank(engine, V);
    REQUIRE(bank.fused());
    bank.clear();
    for (size_t v = 0; v < V; ++v)
    {
      bank.coeffs<1>(v, Bandpass::makeCoeffs(0.05f, 0.5f));
      bank.coeffs<2>(v, std::array<float, 1>{0.25f});
      bank.input(v, (v % 2 == 0) ? 440.f / 48000.f : 0.f);
    }
    DeviceSignal z(engine, V, 2, MLGPU_LAYOUT_QUAD);
    bank(z);
    auto h = z.toRows();  // [t][v][64]
    REQUIRE(h[0] == hexf("-0x1.205afcp-5"));
    REQUIRE(h[1] == hexf("-0x1.8c0f74p-4"));
    REQUIRE(h[2] == hexf("-0x1.1d388ep-3"));
    REQUIRE(h[3] == hexf("-0x1.4b4bd8p-3"));
    REQUIRE(h[63] == hexf("0x1.b99136p-7"));
    // even voices are identical, odd voices (freq 0) stay silent after the first sample's BLEP
    bool evenSame = true;
    for (size_t v = 2; v < V; v += 2)
      for (int t = 0; t < 2; ++t) evenSame = evenSame && !memcmp(&h[(t * V + v) * 64], &h[(t * V) * 64], 256);
    REQUIRE(evenSame);
    // state carries across calls: two 1-vector calls == one 2-vector call
    VoiceBank<SawGen, Bandpass, Gain> bank2(engine, V);
    bank2.clear();
    bank2.coeffsAll<1>(Bandpass::makeCoeffs(0.05f, 0.5f));
    bank2.coeffsAll<2>(std::array<float, 1>{0.25f});
    for (size_t v = 0; v < V; ++v) bank2.input(v, (v % 2 == 0) ? 440.f / 48000.f : 0.f);
    DeviceSignal a(engine, V, 1, MLGPU_LAYOUT_ROWS), b(engine, V, 1, MLGPU_LAYOUT_ROWS);
    bank2(a);
    bank2(b);
    auto ha = a.toRows(), hb = b.toRows();
    REQUIRE(!memcmp(ha.data(), h.data(), ha.size() * 4));
    REQUIRE(!memcmp(hb.data(), h.data() + ha.size(), hb.size() * 4));
  }

  // a filter bank fed by a streamed signal: OnePole impulse response, Appendix B anchor
  {
    VoiceBank<OnePole> lp(engine, 64);
    auto c = OnePole::makeCoeffs(0.15f);
    uint32_t bits[2];
    memcpy(bits, c.data(), 8);
    REQUIRE(bits[0] == 0x3f1c3f2cu);
    REQUIRE(bits[1] == 0x3ec781a9u);
    lp.coeffsAll<0>(c);
    DeviceSignal x(engine, 64, 1, MLGPU_LAYOUT_ROWS), y(engine, 64, 1, MLGPU_LAYOUT_ROWS);
    std::vector<float> imp(64 * 64, 0.f);
    for (int v = 0; v < 64; ++v) imp[v * 64] = 1.f;
    x.fromRows(imp);
    lp(x, y);
    auto h = y.toRows();
    uint32_t o1, o63;
    memcpy(&o1, &h[1], 4);
    memcpy(&o63, &h[63], 4);
    REQUIRE(o1 == 0x3e73887cu);
    REQUIRE(o63 == 0x14458c96u);
  }

  // error channel: the reference silently returns; the mirror throws
  {
    bool threw = false;
    try
    {
      VoiceBank<Lopass> bad(engine, 0);
    }
    catch (const Error& e)
    {
      threw = (e.status == MLGPU_ERR_INVALID);
    }
    REQUIRE(threw);
  }

  if (failures == 0) printf("All tests passed\n");
  return failures ? 1 : 0;
}
