// chains.hip — voice-bank kernels: one wavefront lane per voice, DSPVectors walked serially.
//
// chain_kernel<Chain<K0,K1,...>> evaluates `pN(...p1(p0(x)))` for T DSPVectors of every voice in
// ONE launch: the lane loads its voice's coefficients and state (SoA, coalesced 4 B/lane),
// keeps them in VGPRs for the whole launch, produces 4 samples at a time and moves them with
// one 16-byte access per lane (1 KiB per wavefront per instruction in the QUAD layout), then
// writes the state back. Intermediate signals between processors never touch memory.
// HBM traffic per voice-sample = 4 B out (+4 B in when a signal is streamed)
// + (4*(NC+NS) read + 4*NS written)/(64*T) — DESIGN.md §Kernels.
//
// No LDS except the 17-tap ImpulseGen table (a genuinely shared coefficient table, staged once
// per workgroup); no MFMA: the path is elementwise/recurrent, not a contraction.
//
// Compile with -ffp-contract=off (see mldsp_math.hpp).
#include <string.h>

#include "mlgpu_internal.hpp"
#include "mldsp_kernels.hpp"

using namespace mldev;
static_assert(kHostMixStripFloats == kMixStrip && kHostGroup16StripFloats == kGroup16Strip, "graph.hip's LDS budget counts these strips");

namespace
{
template <bool HAS_SIGNAL, int... KS>
hipError_t launchChain(const ChainArgs& a, hipStream_t stream, int /*cuCount*/)
{
  const unsigned blocks = (unsigned)((a.V + kChainBlock - 1) / kChainBlock);
  hipLaunchKernelGGL((chain_kernel<Chain<KS...>, HAS_SIGNAL>), dim3(blocks), dim3(kChainBlock), 0, stream, a);
  return hipGetLastError();
}

template <bool HAS_SIGNAL, int... KS>
hipError_t launchChainMix(const ChainArgs& a, hipStream_t stream, int /*cuCount*/)
{
  const unsigned blocks = (unsigned)((a.V + kChainBlock - 1) / kChainBlock);
  hipLaunchKernelGGL((chain_mix_kernel<Chain<KS...>, HAS_SIGNAL>), dim3(blocks), dim3(kChainBlock), 0, stream, a);
  return hipGetLastError();
}

// How many wavefront lanes share one channel of a head-less SVF cascade (cascade_lanes_kernel). One lane per channel is the
// cheapest per sample, but a bank needs ~49 000 channels before its wavefronts (two per SIMD at ~190 VGPRs) fill the chip;
// smaller banks are spread over 2 or 4 lanes per channel (4 or 6 wavefronts per SIMD). Thresholds from
// profiles/archive/r03_cascade_lanes.txt (8 x Lopass, 32 DSPVectors per launch, us with 4 / 2 / 1 lanes): 4 096 channels 84 / 117 /
// 173, 16 384: 99 / 119 / 174, 32 768: 140 / 134 / 177, 49 152: 203 / 205 / 194, 65 536: 265 / 242 / 224.
template <int N>
int cascadeLanesFor(size_t V, uint32_t flags)
{
  const unsigned forced = (flags & MLGPU_KFLAG_CASCADE_MASK) >> MLGPU_KFLAG_CASCADE_SHIFT;
  int lpc = V < 24576 ? 4 : (V < 40960 ? 2 : 1);
  if (forced == 1 || forced == 2 || forced == 4) lpc = (int)forced;
  if (forced == 7) return 0;
  while (lpc > 1 && (N % (2 * lpc)) != 0) lpc >>= 1;  // every lane runs an even number of stages
  return lpc;
}

template <int KIND, int N, int LPC, int MINW, bool HAS_SIGNAL>
hipError_t launchLanes(const ChainArgs& a, hipStream_t stream)
{
  constexpr unsigned channelsPerBlock = kChainBlock / LPC;
  const unsigned blocks = (unsigned)((a.V + channelsPerBlock - 1) / channelsPerBlock);
  hipLaunchKernelGGL((cascade_lanes_kernel<KIND, N, LPC, 8, MINW, HAS_SIGNAL>), dim3(blocks), dim3(kChainBlock), 0, stream, a);
  return hipGetLastError();
}

template <bool HAS_SIGNAL, int KIND, int N, int... HEADS>
hipError_t launchCascade(const ChainArgs& a, hipStream_t stream, int /*cuCount*/)
{
  if constexpr (sizeof...(HEADS) == 0)
  {
    switch (cascadeLanesFor<N>(a.V, a.flags))
    {
      case 1: return launchLanes<KIND, N, 1, 2, HAS_SIGNAL>(a, stream);
      case 2: if constexpr (N % 4 == 0) return launchLanes<KIND, N, 2, 4, HAS_SIGNAL>(a, stream); break;
      case 4: if constexpr (N % 8 == 0) return launchLanes<KIND, N, 4, 6, HAS_SIGNAL>(a, stream); break;
      default: break;
    }
  }
  const unsigned blocks = (unsigned)((a.V + kChainBlock - 1) / kChainBlock);
  hipLaunchKernelGGL((cascade_kernel<Chain<HEADS...>, KIND, N, HAS_SIGNAL>), dim3(blocks), dim3(kChainBlock), 0, stream, a);
  return hipGetLastError();
}

// what a profiler prints for the kernel a cascade bank of V channels runs
template <int KIND, int N, int... HEADS>
const char* cascadeKernelName(size_t V, uint32_t flags)
{
  static std::string names[5];
  int lpc = 0;
  if constexpr (sizeof...(HEADS) == 0) lpc = cascadeLanesFor<N>(V, flags);
  std::string& n = names[lpc];
  if (n.empty())
  {
    if (lpc > 0)
      n = "cascade_lanes_kernel<" + std::to_string(KIND) + ", " + std::to_string(N) + ", " + std::to_string(lpc) + ", 8, " +
          std::to_string(lpc == 1 ? 2 : (lpc == 2 ? 4 : 6));
    else
    {
      n = "cascade_kernel<mldev::Chain<";
      const int hs[] = {HEADS..., 0};
      for (size_t i = 0; i < sizeof...(HEADS); ++i) n += (i ? ", " : "") + std::to_string(hs[i]);
      n += ">, " + std::to_string(KIND) + ", " + std::to_string(N);
    }
  }
  return n.c_str();
}

template <int KIND, int N, int... HEADS>
ChainEntry makeCascadeEntry(const char* name)
{
  ChainEntry e;
  e.kinds = {HEADS...};
  for (int s = 0; s < N; ++s) e.kinds.push_back(KIND);
  e.launchSignal = &launchCascade<true, KIND, N, HEADS...>;
  e.launchConst = &launchCascade<false, KIND, N, HEADS...>;
  static const std::string profName = [] {
    std::string n = "cascade_kernel<mldev::Chain<";
    const int hs[] = {HEADS..., 0};
    for (size_t i = 0; i < sizeof...(HEADS); ++i) n += (i ? ", " : "") + std::to_string(hs[i]);
    return n + ">, " + std::to_string(KIND) + ", " + std::to_string(N);
  }();
  e.kernelName = profName.c_str();
  e.kernelNameFor = &cascadeKernelName<KIND, N, HEADS...>;
  e.alias = name;
  e.nc = Chain<HEADS...>::NC + SvfCascade<KIND, N>::NC;
  e.ns = Chain<HEADS...>::NS + SvfCascade<KIND, N>::NS;
  return e;
}

template <int... KS>
ChainEntry makeEntry(const char* name)
{
  ChainEntry e;
  e.kinds = {KS...};
  e.launchSignal = &launchChain<true, KS...>;
  e.launchConst = &launchChain<false, KS...>;
  // what a profiler prints for this kernel: "chain_kernel<mldev::Chain<2, 18, 48>, false>(ChainArgs)";
  // `name` is the human-readable alias used in logs.
  static const std::string profName = [] {
    std::string n = "chain_kernel<mldev::Chain<";
    const int ks[] = {KS...};
    for (size_t i = 0; i < sizeof...(KS); ++i) n += (i ? ", " : "") + std::to_string(ks[i]);
    return n + ">";
  }();
  e.kernelName = profName.c_str();
  e.alias = name;
  e.nc = Chain<KS...>::NC;
  e.ns = Chain<KS...>::NS;
  return e;
}

// ... and with the form that sums the voices in the kernel (mlgpu_bank_process_mixdown): the fused voice chains
template <int... KS>
ChainEntry makeMixEntry(const char* name)
{
  ChainEntry e = makeEntry<KS...>(name);
  e.launchMixSignal = &launchChainMix<true, KS...>;
  e.launchMixConst = &launchChainMix<false, KS...>;
  return e;
}

#define P(x) MLGPU_PROC_##x

const std::vector<ChainEntry>& registry()
{
  static const std::vector<ChainEntry> r = {
      // every reference processor on its own (also the building blocks of unfused chains)
      makeEntry<P(PHASOR_GEN)>("chain_kernel<PhasorGen>"),
      makeEntry<P(SINE_GEN)>("chain_kernel<SineGen>"),
      makeEntry<P(SAW_GEN)>("chain_kernel<SawGen>"),
      makeEntry<P(PULSE_GEN)>("chain_kernel<PulseGen>"),
      makeEntry<P(NOISE_GEN)>("chain_kernel<NoiseGen>"),
      makeEntry<P(TICK_GEN)>("chain_kernel<TickGen>"),
      makeEntry<P(IMPULSE_GEN)>("chain_kernel<ImpulseGen>"),
      makeEntry<P(ONE_SHOT_GEN)>("chain_kernel<OneShotGen>"),
      makeEntry<P(TEST_SINE_GEN)>("chain_kernel<TestSineGen>"),
      makeEntry<P(LOPASS)>("chain_kernel<Lopass>"),
      makeEntry<P(HIPASS)>("chain_kernel<Hipass>"),
      makeEntry<P(BANDPASS)>("chain_kernel<Bandpass>"),
      makeEntry<P(LO_SHELF)>("chain_kernel<LoShelf>"),
      makeEntry<P(HI_SHELF)>("chain_kernel<HiShelf>"),
      makeEntry<P(BELL)>("chain_kernel<Bell>"),
      makeEntry<P(ONE_POLE)>("chain_kernel<OnePole>"),
      makeEntry<P(DC_BLOCKER)>("chain_kernel<DCBlocker>"),
      makeEntry<P(DIFFERENTIATOR)>("chain_kernel<Differentiator>"),
      makeEntry<P(INTEGRATOR)>("chain_kernel<Integrator>"),
      makeEntry<P(PEAK)>("chain_kernel<Peak>"),
      makeEntry<P(RMS)>("chain_kernel<RMS>"),
      makeEntry<P(ADSR)>("chain_kernel<ADSR>"),
      makeEntry<P(GAIN)>("chain_kernel<Gain>"),
      makeEntry<P(SAMPLE_ACCURATE_LINEAR_GLIDE)>("chain_kernel<SampleAccurateLinearGlide>"),
      makeEntry<P(ALLPASS1)>("chain_kernel<Allpass1>"),
      // fused chains of the BASELINE.json configs
      makeMixEntry<P(SINE_GEN), P(LOPASS)>("chain_kernel<SineGen,Lopass>"),                       // config 1
      makeMixEntry<P(SAW_GEN), P(BANDPASS), P(GAIN)>("chain_kernel<SawGen,Bandpass,Gain>"),       // config 3
      makeMixEntry<P(SAW_GEN), P(BANDPASS)>("chain_kernel<SawGen,Bandpass>"),
      makeCascadeEntry<P(LOPASS), 8>("cascade_kernel<Lopass x8>"),                             // config 4
      makeCascadeEntry<P(LOPASS), 8, P(NOISE_GEN)>("cascade_kernel<NoiseGen,Lopass x8>"),
      // shorter / other SVF cascades (filter banks, steeper slopes)
      makeCascadeEntry<P(LOPASS), 2>("cascade_kernel<Lopass x2>"),
      makeCascadeEntry<P(LOPASS), 4>("cascade_kernel<Lopass x4>"),
      makeCascadeEntry<P(HIPASS), 2>("cascade_kernel<Hipass x2>"),
      makeCascadeEntry<P(HIPASS), 4>("cascade_kernel<Hipass x4>"),
      makeCascadeEntry<P(BANDPASS), 2>("cascade_kernel<Bandpass x2>"),
      makeCascadeEntry<P(BANDPASS), 4>("cascade_kernel<Bandpass x4>"),
      makeCascadeEntry<P(LOPASS), 4, P(SAW_GEN)>("cascade_kernel<SawGen,Lopass x4>"),
      // other common voices
      makeMixEntry<P(PULSE_GEN), P(HIPASS), P(ONE_POLE)>("chain_kernel<PulseGen,Hipass,OnePole>"),
      makeMixEntry<P(SAW_GEN), P(LOPASS), P(GAIN)>("chain_kernel<SawGen,Lopass,Gain>"),
      makeMixEntry<P(SINE_GEN), P(GAIN)>("chain_kernel<SineGen,Gain>"),
  };
  return r;
}
}  // namespace

const ChainEntry* mlgpu_find_chain(const int32_t* kinds, int n)
{
  for (const ChainEntry& e : registry())
  {
    if ((int)e.kinds.size() != n) continue;
    bool same = true;
    for (int i = 0; i < n; ++i) same = same && (e.kinds[i] == kinds[i]);
    if (same) return &e;
  }
  return nullptr;
}

// vector-rate processors (one float per DSPVector in): graph nodes only, no chain kernel
static bool graphOnlyInfo(int kind, int* nc, int* ns)
{
  switch (kind)
  {
    case MLGPU_PROC_INTERPOLATOR1: *nc = Proc<MLGPU_PROC_INTERPOLATOR1>::NC; *ns = Proc<MLGPU_PROC_INTERPOLATOR1>::NS; return true;
    case MLGPU_PROC_LINEAR_GLIDE: *nc = Proc<MLGPU_PROC_LINEAR_GLIDE>::NC; *ns = Proc<MLGPU_PROC_LINEAR_GLIDE>::NS; return true;
    case MLGPU_PROC_TEMPO_LOCK: *nc = Proc<MLGPU_PROC_TEMPO_LOCK>::NC; *ns = Proc<MLGPU_PROC_TEMPO_LOCK>::NS; return true;
    // delay lines own HBM rings that only a graph allocates
    case MLGPU_PROC_INTEGER_DELAY: *nc = Proc<MLGPU_PROC_INTEGER_DELAY>::NC; *ns = Proc<MLGPU_PROC_INTEGER_DELAY>::NS; return true;
    case MLGPU_PROC_FRACTIONAL_DELAY: *nc = Proc<MLGPU_PROC_FRACTIONAL_DELAY>::NC; *ns = Proc<MLGPU_PROC_FRACTIONAL_DELAY>::NS; return true;
    case MLGPU_PROC_PITCHBENDABLE_DELAY: *nc = Proc<MLGPU_PROC_PITCHBENDABLE_DELAY>::NC; *ns = Proc<MLGPU_PROC_PITCHBENDABLE_DELAY>::NS; return true;
    // the resampling filters at the edges of a rate region
    case MLGPU_PROC_HALF_BAND: *nc = Proc<MLGPU_PROC_HALF_BAND>::NC; *ns = Proc<MLGPU_PROC_HALF_BAND>::NS; return true;
    case MLGPU_PROC_HALF_BAND_BUFFERED: *nc = Proc<MLGPU_PROC_HALF_BAND_BUFFERED>::NC; *ns = Proc<MLGPU_PROC_HALF_BAND_BUFFERED>::NS; return true;
    default: return false;
  }
}
bool mlgpu_proc_is_vector_rate(int kind) { return kind == MLGPU_PROC_INTERPOLATOR1 || kind == MLGPU_PROC_LINEAR_GLIDE || kind == MLGPU_PROC_TEMPO_LOCK; }
bool mlgpu_proc_is_graph_only(int kind)
{
  int nc, ns;
  return graphOnlyInfo(kind, &nc, &ns);
}
int mlgpu_proc_rings(int kind)  // delay rings per voice
{
  switch (kind)
  {
    case MLGPU_PROC_INTEGER_DELAY: case MLGPU_PROC_FRACTIONAL_DELAY: return 1;
    case MLGPU_PROC_PITCHBENDABLE_DELAY: return 2;
    default: return 0;
  }
}
// which state words T::clear() resets (bit i = word i); delay lines keep their write index and delay time
uint64_t mlgpu_proc_clear_mask(int kind)
{
  switch (kind)
  {
    case MLGPU_PROC_ADSR: return 1ull << 7;             // only the segment, MLDSPFilters.h:698
    case MLGPU_PROC_TEMPO_LOCK: return 0x1;              // clear() { _omega = -1 }, :1487
    case MLGPU_PROC_INTEGER_DELAY: return 0;             // only the buffer, :833
    case MLGPU_PROC_FRACTIONAL_DELAY: return 0x6;        // + Allpass1 x1, y1, :985-989
    case MLGPU_PROC_PITCHBENDABLE_DELAY: return 0x6 | (0x6 << 5);
    default: return ~0ull;
  }
}
int mlgpu_proc_nc(int kind)
{
  const int32_t k = kind;
  const ChainEntry* e = mlgpu_find_chain(&k, 1);
  int nc = -1, ns = -1;
  if (!e) graphOnlyInfo(kind, &nc, &ns);
  return e ? e->nc : nc;
}
int mlgpu_proc_ns(int kind)
{
  const int32_t k = kind;
  const ChainEntry* e = mlgpu_find_chain(&k, 1);
  int nc = -1, ns = -1;
  if (!e) graphOnlyInfo(kind, &nc, &ns);
  return e ? e->ns : ns;
}

// `coeffs` of a default-constructed reference object where that is not all zeros
void mlgpu_proc_default_coeffs(int kind, float* c /*[nc]*/)
{
  const int nc = mlgpu_proc_nc(kind);
  for (int i = 0; i < nc; ++i) c[i] = 0.f;
  if (kind == MLGPU_PROC_LINEAR_GLIDE || kind == MLGPU_PROC_SAMPLE_ACCURATE_LINEAR_GLIDE)
  {
    const int32_t n = 32;  // mVectorsPerGlide{32} / mSamplesPerGlide{32}, MLDSPGens.h:440,522
    memcpy(&c[0], &n, 4);
    c[1] = 1.f / 32;       // mDyPerVector / mDyPerSample
  }
  if (kind == MLGPU_PROC_PEAK && nc >= 3)
  {
    const int32_t hold = 44100;  // peakHoldSamples{44100}, MLDSPFilters.h:574 (an int travelling in a coefficient slot)
    memcpy(&c[2], &hold, 4);
  }
}

// state words of a default-constructed (cleared == false) or clear()ed reference object
void mlgpu_proc_clear_state(int kind, uint32_t* words, bool cleared)
{
  const int ns = mlgpu_proc_ns(kind);
  for (int i = 0; i < ns; ++i) words[i] = 0;
  if (kind == MLGPU_PROC_SINE_GEN && cleared) words[0] = 0xC0000000u;  // kZeroPhase, MLDSPGens.h:375,379
  if (kind == MLGPU_PROC_ADSR) words[7] = 4;                            // segment{off}, MLDSPFilters.h:700-702
  if (kind == MLGPU_PROC_TEMPO_LOCK) words[0] = 0xBF800000u;            // _omega{-1.f} (and clear()), MLDSPFilters.h:1481,1487
  if (kind == MLGPU_PROC_TEMPO_LOCK && cleared) words[1] = 0;
  if (kind == MLGPU_PROC_LINEAR_GLIDE) words[2] = 0xFFFFFFFFu;          // mVectorsRemaining{-1}, MLDSPGens.h:441,513
  if (kind == MLGPU_PROC_SAMPLE_ACCURATE_LINEAR_GLIDE) words[3] = 0xFFFFFFFFu;  // mSamplesRemaining{-1}, :524,588
}
