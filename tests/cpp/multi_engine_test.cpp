// tests/cpp/multi_engine_test.cpp — the in-process multi-device host (ml::gpu::DeviceGroup) through the C-ABI:
// BASELINE config 3's chain (SawGen -> Bandpass -> gain, per-voice parameters a function of the GLOBAL voice index) run
// (a) unsharded on one engine and (b) sharded over a DeviceGroup, one host thread + engine + stream per device, all
// devices launching at once. The union of the shards must equal the unsharded run bit for bit, outputs and final state
// (voices share nothing: MLDSPFunctional.h:321-349). With one visible GPU the group has one member and the same property
// is checked with two ENGINES on that GPU driven from two threads (thread-compatibility of the per-engine ABI); with two or
// more, devices 0 and 1 (or argv[1] devices). Runs on the GPU box (pytest -m gpu drives it).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <set>
#include <string>
#include <thread>

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

using Bank3 = VoiceBank<SawGen, Bandpass, Gain>;

struct Shard
{
  std::vector<float> rows;                 // [T][n][64]
  std::vector<uint32_t> phase, ic1, ic2;   // final state
};

// voices [lo, hi) of `total`, two launches of T vectors each (state carried in HBM between them)
static Shard runRange(Engine& e, size_t lo, size_t hi, size_t total, size_t T)
{
  const size_t n = hi - lo;
  Bank3 bank(e, n);
  bank.clear();
  for (size_t i = 0; i < n; ++i)
  {
    const double v = (double)(lo + i);
    const float freq = (float)(55.0 * std::pow(2.0, 5.0 * v / (double)total) / 48000.0);
    const float omega = std::fmin(0.45f, 4.f * freq);
    bank.coeffs<1>(i, Bandpass::makeCoeffs(omega, 0.5f));
    bank.coeffs<2>(i, std::array<float, 1>{0.25f});
    bank.input(i, freq);
  }
  DeviceSignal a(e, n, T, MLGPU_LAYOUT_QUAD), b(e, n, T, MLGPU_LAYOUT_QUAD);
  bank(a);
  bank(b);
  Shard s;
  auto ha = a.toRows(), hb = b.toRows();
  s.rows = ha;
  s.rows.insert(s.rows.end(), hb.begin(), hb.end());
  s.phase = bank.state(0, 0);
  s.ic1 = bank.state(1, 0);
  s.ic2 = bank.state(1, 1);
  return s;
}

static bool sameVoice(const Shard& whole, size_t V, size_t v, const Shard& part, size_t n, size_t i, size_t T2)
{
  for (size_t t = 0; t < T2; ++t)
    if (memcmp(&whole.rows[(t * V + v) * 64], &part.rows[(t * n + i) * 64], 256)) return false;
  return whole.phase[v] == part.phase[i] && whole.ic1[v] == part.ic1[i] && whole.ic2[v] == part.ic2[i];
}

int main(int argc, char** argv)
{
  const int have = mlgpu_device_count();
  if (have < 1)
  {
    printf("no GPU\n");
    return 2;
  }
  const int want = argc > 1 ? atoi(argv[1]) : (have >= 2 ? 2 : 1);
  const size_t V = 3001, T = 3;   // ragged: not a multiple of the wavefront or of the group size

  // asking for more devices than exist fails loudly
  {
    bool threw = false;
    try
    {
      DeviceGroup tooMany(have + 1);
    }
    catch (const Error& e)
    {
      threw = (e.status == MLGPU_ERR_NO_DEVICE);
    }
    REQUIRE(threw);
  }

  Engine e0(0);
  const Shard whole = runRange(e0, 0, V, V, T);

  // (b) the group: every device computes its range on its own thread, concurrently
  {
    DeviceGroup group(want);
    REQUIRE(group.size() == want);
    std::set<std::string> buses;
    for (int g = 0; g < group.size(); ++g)
    {
      char bus[64] = {0};
      REQUIRE(mlgpu_device_pci_bus_id(mlgpu_engine_device(group.engine(g).handle()), bus, sizeof(bus)) == MLGPU_OK);
      buses.insert(bus);
    }
    REQUIRE((int)buses.size() == want);   // distinct GPUs
    std::vector<Shard> shards((size_t)want);
    std::vector<std::pair<size_t, size_t>> spans((size_t)want);
    group.forEach(V, [&](int g, Engine& e, size_t lo, size_t hi) {
      spans[(size_t)g] = {lo, hi};
      shards[(size_t)g] = runRange(e, lo, hi, V, T);
    });
    group.sync();
    size_t covered = 0;
    bool same = true;
    for (int g = 0; g < want; ++g)
    {
      const size_t lo = spans[(size_t)g].first, hi = spans[(size_t)g].second;
      REQUIRE(lo == covered);
      covered = hi;
      for (size_t v = lo; v < hi; ++v) same = same && sameVoice(whole, V, v, shards[(size_t)g], hi - lo, v - lo, 2 * T);
    }
    REQUIRE(covered == V);
    REQUIRE(same);
    // an exception inside one device's job comes back to the caller
    bool threw = false;
    try
    {
      group.forEach(V, [&](int g, Engine& e, size_t, size_t) {
        if (g == want - 1) Bank3 bad(e, 0);
      });
    }
    catch (const Error& e)
    {
      threw = (e.status == MLGPU_ERR_INVALID);
    }
    REQUIRE(threw);
    printf("DeviceGroup of %d device(s): union of shards == unsharded (%zu voices x %zu vectors, outputs and state)\n", want, V, 2 * T);
  }

  // two engines driven from two plain threads (same device when there is only one): the ABI keeps no global state
  {
    Engine ea(0), eb(have >= 2 ? 1 : 0);
    const auto s0 = DeviceGroup::partition(V, 2, 0), s1 = DeviceGroup::partition(V, 2, 1);
    Shard a, b;
    std::thread ta([&] { a = runRange(ea, s0.first, s0.second, V, T); });
    std::thread tb([&] { b = runRange(eb, s1.first, s1.second, V, T); });
    ta.join();
    tb.join();
    bool same = true;
    for (size_t v = s0.first; v < s0.second; ++v) same = same && sameVoice(whole, V, v, a, s0.second - s0.first, v - s0.first, 2 * T);
    for (size_t v = s1.first; v < s1.second; ++v) same = same && sameVoice(whole, V, v, b, s1.second - s1.first, v - s1.first, 2 * T);
    REQUIRE(same);
  }

  // The real-time block across a group (GroupMixdown): all voices of a sharded bank summed to one channel, each member's voices summed
  // inside its voice kernel up to the hand-over level of the mixdown tree, the tree finished on the host - the bits ONE engine gives
  // for all the voices. Groups of 2 and 8 members (on a one-GPU box: engines of that GPU, an explicit device list with repeats), two
  // blocks with carried state; voices = 8 x 4096 (every member of the 8-group hands over one level-2 row, of the 2-group four).
  {
    const size_t Vm = 32768, Tm = 2;
    auto setup = [&](Bank3& bank, size_t lo, size_t n) {
      bank.clear();
      for (size_t i = 0; i < n; ++i)
      {
        const double v = (double)(lo + i);
        const float freq = (float)(55.0 * std::pow(2.0, 5.0 * v / (double)Vm) / 48000.0);
        bank.coeffs<1>(i, Bandpass::makeCoeffs(std::fmin(0.45f, 4.f * freq), 0.5f));
        bank.coeffs<2>(i, std::array<float, 1>{0.25f});
        bank.input(i, freq);
      }
    };
    std::vector<float> one(2 * 64 * Tm);
    {
      Bank3 bank(e0, Vm);
      setup(bank, 0, Vm);
      e0.check(mlgpu_mixdown_reserve(e0.handle(), Vm, Tm));
      DeviceSignal mix(e0, 1, Tm, MLGPU_LAYOUT_QUAD);
      for (int block = 0; block < 2; ++block)
      {
        bank.mixdown(Tm, mix.data());
        e0.check(mlgpu_download(e0.handle(), one.data() + (size_t)block * 64 * Tm, mix.data(), sizeof(float) * 64 * Tm));
      }
    }
    for (int members : {2, 8})
    {
      std::vector<int> devices;
      for (int m = 0; m < members; ++m) devices.push_back(m % have);
      DeviceGroup group(devices);
      REQUIRE(group.size() == members);
      GroupMixdown gm(group, Vm, Tm);
      REQUIRE(gm.voicesPerMember() == Vm / (size_t)members);
      REQUIRE(gm.rowsPerMember() == (members == 8 ? 1u : 4u));
      std::vector<std::unique_ptr<Bank3>> banks((size_t)members);
      group.forEach(Vm, [&](int g, Engine& e, size_t lo, size_t hi) {
        banks[(size_t)g].reset(new Bank3(e, hi - lo));
        setup(*banks[(size_t)g], lo, hi - lo);
      });
      std::vector<float> got(2 * 64 * Tm);
      for (int block = 0; block < 2; ++block)
        gm.process(Tm, [&](int g, Engine&, size_t, size_t, float* dRows) { banks[(size_t)g]->mixdownShard(Tm, dRows); }, got.data() + (size_t)block * 64 * Tm);
      const bool same = memcmp(got.data(), one.data(), sizeof(float) * got.size()) == 0;
      REQUIRE(same);
      float peak = 0.f;
      for (float x : got) peak = std::fmax(peak, std::fabs(x));
      REQUIRE(peak > 1e-3f);
      group.forEach(Vm, [&](int g, Engine&, size_t, size_t) { banks[(size_t)g].reset(); });   // (a bank goes before its engine)
      printf("GroupMixdown, %d members x %zu voices: the channel of all %zu voices == one engine's, bit for bit (%zu samples)\n", members, Vm / (size_t)members, Vm, got.size());
    }
    // shares that are not whole first-stage groups are refused at setup
    bool threw = false;
    try
    {
      DeviceGroup two(std::vector<int>{0, 0});
      GroupMixdown bad(two, 2 * 100, 1);
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
