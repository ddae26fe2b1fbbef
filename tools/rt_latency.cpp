// tools/rt_latency.cpp — what a real-time host sees: wall time of one mlgpu_process_buffer_process call (the
// SignalProcessBuffer of source/app/MLSignalProcessBuffer.cpp:36-90) per host block, p50 / p99 / max over many blocks,
// for 64-, 128- and 512-frame blocks at 1, 1 024 and 262 144 voices, in the synchronous mode (a call returns its own
// block: H2D + launches + D2H + wait) and in the pipelined mode (two staging sets; the call returns what the previous
// calls computed). The process function is BASELINE config 3's voice bank (SawGen -> Bandpass -> gain) summed to one
// output channel (mlgpu_mixdown), i.e. a polyphonic instrument's block. Host C++ over the C-ABI only.
//   g++ -std=c++17 -O2 -Iinclude tools/rt_latency.cpp -o tools/bin/rt_latency -Lmadronalib_amd/csrc -lmlgpu -Wl,-rpath,'$ORIGIN/../../madronalib_amd/csrc' -Wl,-rpath,/opt/rocm/lib
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "mlgpu/mldsp_gpu.hpp"

using namespace ml::gpu;
using Clock = std::chrono::steady_clock;

struct Instrument
{
  Engine* eng;
  mlgpu_bank* raw;
  float* d_voices;
  size_t V;
};

static int onVectors(void* user, size_t nVectors, const float* const*, float* const* dOut)
{
  Instrument* in = static_cast<Instrument*>(user);
  int st = mlgpu_bank_process(in->raw, nVectors, nullptr, MLGPU_LAYOUT_QUAD, in->d_voices, MLGPU_LAYOUT_QUAD);
  if (st != MLGPU_OK) return st;
  return mlgpu_mixdown(in->eng->handle(), in->d_voices, MLGPU_LAYOUT_QUAD, in->V, nVectors, nullptr, dOut[0]);
}

int main(int argc, char** argv)
{
  const int blocks = argc > 1 ? atoi(argv[1]) : 2000;
  Engine eng(0);
  printf("{\"tool\": \"rt_latency\", \"blocks_per_case\": %d, \"process\": \"SawGen->Bandpass->gain voice bank + mixdown to one channel\", \"cases\": [\n", blocks);
  bool first = true;
  for (size_t V : {(size_t)1, (size_t)1024, (size_t)262144})
    for (int frames : {64, 128, 512})
      for (int pipelined = 0; pipelined < 2; ++pipelined)
      {
        const size_t maxVectors = (size_t)frames / 64 + 1;
        DeviceSignal voices(eng, V, maxVectors, MLGPU_LAYOUT_QUAD);
        eng.check(mlgpu_mixdown_reserve(eng.handle(), V, maxVectors));
        mlgpu_bank* raw = nullptr;
        const int32_t kinds[3] = {MLGPU_PROC_SAW_GEN, MLGPU_PROC_BANDPASS, MLGPU_PROC_GAIN};
        eng.check(mlgpu_bank_create(eng.handle(), kinds, 3, V, &raw));
        eng.check(mlgpu_bank_clear(raw));
        {
          auto c = Bandpass::makeCoeffs(0.05f, 0.5f);
          for (int i = 0; i < 3; ++i) eng.check(mlgpu_bank_set_coeff_uniform(raw, 1, i, c[(size_t)i]));
          eng.check(mlgpu_bank_set_coeff_uniform(raw, 2, 0, 1.0f / (float)std::sqrt((double)V)));
          std::vector<float> f(V);
          for (size_t v = 0; v < V; ++v) f[v] = (float)(55.0 * std::pow(2.0, 5.0 * (double)v / (double)V) / 48000.0);
          eng.check(mlgpu_bank_set_input_const(raw, f.data()));
        }
        Instrument inst{&eng, raw, voices.data(), V};
        mlgpu_process_buffer* pb = nullptr;
        eng.check(mlgpu_process_buffer_create(eng.handle(), 0, 1, (size_t)frames, &pb));
        eng.check(mlgpu_process_buffer_set_pipelined(pb, pipelined));
        std::vector<float> out((size_t)frames);
        float* outs[1] = {out.data()};
        std::vector<double> us;
        us.reserve((size_t)blocks);
        double peak = 0;
        for (int b = 0; b < blocks + 50; ++b)
        {
          const auto t0 = Clock::now();
          eng.check(mlgpu_process_buffer_process(pb, nullptr, outs, frames, onVectors, &inst));
          const auto t1 = Clock::now();
          if (b >= 50) us.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
          for (float x : out) peak = std::max(peak, (double)std::fabs(x));
          if (pipelined)
          {
            // a host's block period: the next callback arrives one block later (at 48 kHz), which is when the work overlaps
            const auto until = t0 + std::chrono::duration<double, std::micro>(frames / 48000.0 * 1e6);
            while (Clock::now() < until) {}
          }
        }
        std::sort(us.begin(), us.end());
        const double p50 = us[us.size() / 2], p99 = us[(size_t)(us.size() * 0.99)], mx = us.back();
        printf("%s  {\"voices\": %zu, \"frames\": %d, \"mode\": \"%s\", \"latency_frames\": %zu, \"block_period_us\": %.1f, \"call_us_p50\": %.1f, \"call_us_p99\": %.1f, "
               "\"call_us_max\": %.1f, \"output_peak\": %.3g}",
               first ? "" : ",\n", V, frames, pipelined ? "pipelined" : "synchronous", mlgpu_process_buffer_latency_frames(pb), frames / 48000.0 * 1e6, p50, p99, mx, peak);
        first = false;
        fflush(stdout);
        mlgpu_process_buffer_destroy(pb);
        mlgpu_bank_destroy(raw);
      }
  printf("\n]}\n");
  return 0;
}
