// tools/rt_target.cpp — BASELINE.json's north_star target, measured: ">= 10^6 concurrent SawGen -> SVF -> gain voice chains at
// 48 kHz real-time on one MI355X at >= 60 % HBM roofline". A real-time host calls once per block (the reference's loop is
// SignalProcessBuffer::process, source/app/MLSignalProcessBuffer.cpp:57-78: one process call per 64-frame DSPVector of a host
// block of up to kMaxBlockSize = 4096 frames, source/app/MLAudioTask.h:25), so per (voices, block size):
//   1. the voice kernel alone, free-running, HIP-event time per block -> voice-samples/s and the fraction of the 8 TB/s HBM peak
//      its ALGORITHMIC bytes make (4 B per voice-sample written + 44 B per voice and launch of coefficients, state and frequency);
//   2. paced at 48 kHz through mlgpu_process_buffer_process (voice bank + mixdown to one channel + D2H of that channel), the host
//      waiting out each block period: wall time per call p50 / p99 / max against the period, and the number of calls that took
//      longer than the period (deadline misses), synchronous and pipelined.
// Host C++ over the C-ABI only.   usage: rt_target [blocks=1500] [voices] [frames] > profiles/r04_rt_target.json
//   g++ -std=c++17 -O2 -Iinclude tools/rt_target.cpp -o tools/bin/rt_target -Lmadronalib_amd/csrc -lmlgpu -Wl,-rpath,'$ORIGIN/../../madronalib_amd/csrc' -Wl,-rpath,/opt/rocm/lib
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
  // round 5: the voices summed inside the voice kernel, their signals never written (the same bits as the two calls below);
  // RT_TWO_CALLS=1 in the environment runs round 4's form
  static const bool twoCalls = getenv("RT_TWO_CALLS") != nullptr;
  if (!twoCalls) return mlgpu_bank_process_mixdown(in->raw, nVectors, nullptr, MLGPU_LAYOUT_QUAD, nullptr, dOut[0]);
  int st = mlgpu_bank_process(in->raw, nVectors, nullptr, MLGPU_LAYOUT_QUAD, in->d_voices, MLGPU_LAYOUT_QUAD);
  if (st != MLGPU_OK) return st;
  return mlgpu_mixdown(in->eng->handle(), in->d_voices, MLGPU_LAYOUT_QUAD, in->V, nVectors, nullptr, dOut[0]);
}

int main(int argc, char** argv)
{
  const int blocks = argc > 1 ? atoi(argv[1]) : 1500;
  const size_t onlyV = argc > 2 ? (size_t)atoll(argv[2]) : 0;   // one voice count (for a rocprofv3 run of that case alone)
  const int onlyFrames = argc > 3 ? atoi(argv[3]) : 0;
  Engine eng(0);
  printf("{\"tool\": \"rt_target\", \"blocks_per_case\": %d, \"sample_rate\": 48000, \"process\": \"SawGen->Bandpass(k=0.5)->gain voice bank (BASELINE configs[2] per-voice frequencies), "
         "then mixdown to one channel and D2H in the paced legs\", \"hbm_peak_GBps\": 8000, \"cases\": [\n", blocks);
  bool first = true;
  for (size_t V : {(size_t)262144, (size_t)1048576, (size_t)2097152, (size_t)4194304, (size_t)8388608})
    for (int frames : {64, 512})
    {
      if ((onlyV && V != onlyV) || (onlyFrames && frames != onlyFrames)) continue;
      const size_t T = (size_t)frames / 64;
      if (V * (size_t)frames * 4 > ((size_t)40 << 30)) continue;
      DeviceSignal voices(eng, V, T + 1, MLGPU_LAYOUT_QUAD);
      eng.check(mlgpu_mixdown_reserve(eng.handle(), V, T + 1));
      mlgpu_bank* raw = nullptr;
      const int32_t kinds[3] = {MLGPU_PROC_SAW_GEN, MLGPU_PROC_BANDPASS, MLGPU_PROC_GAIN};
      eng.check(mlgpu_bank_create(eng.handle(), kinds, 3, V, &raw));
      eng.check(mlgpu_bank_clear(raw));
      {
        std::vector<float> f(V), c0(V), c1(V), c2(V);
        for (size_t v = 0; v < V; ++v)
        {
          f[v] = (float)(55.0 * std::pow(2.0, 5.0 * (double)v / (double)V) / 48000.0);
          auto c = Bandpass::makeCoeffs(std::min(0.45f, 4.0f * f[v]), 0.5f);
          c0[v] = c[0];
          c1[v] = c[1];
          c2[v] = c[2];
        }
        eng.check(mlgpu_bank_set_coeff(raw, 1, 0, c0.data()));
        eng.check(mlgpu_bank_set_coeff(raw, 1, 1, c1.data()));
        eng.check(mlgpu_bank_set_coeff(raw, 1, 2, c2.data()));
        eng.check(mlgpu_bank_set_coeff_uniform(raw, 2, 0, 0.25f));
        eng.check(mlgpu_bank_set_input_const(raw, f.data()));
      }
      // 1. the voice kernel alone, free-running
      const int reps = std::max(20, std::min(400, (int)(2.0e11 / ((double)V * frames))));
      for (int i = 0; i < 10; ++i) eng.check(mlgpu_bank_process(raw, T, nullptr, MLGPU_LAYOUT_QUAD, voices.data(), MLGPU_LAYOUT_QUAD));
      eng.sync();
      eng.check(mlgpu_timer_start(eng.handle()));
      for (int i = 0; i < reps; ++i) eng.check(mlgpu_bank_process(raw, T, nullptr, MLGPU_LAYOUT_QUAD, voices.data(), MLGPU_LAYOUT_QUAD));
      float ms = 0;
      eng.check(mlgpu_timer_stop_ms(eng.handle(), &ms));
      const double kernelUs = 1000.0 * ms / reps;
      const double algBytes = (double)V * (4.0 * frames + 44.0);
      const double hbmFrac = algBytes / (kernelUs * 1e-6) / 8e12;
      const double periodUs = frames / 48000.0 * 1e6;
      for (int pipelined = 0; pipelined < 2; ++pipelined)
      {
        Instrument inst{&eng, raw, voices.data(), V};
        mlgpu_process_buffer* pb = nullptr;
        eng.check(mlgpu_process_buffer_create(eng.handle(), 0, 1, (size_t)frames, &pb));
        eng.check(mlgpu_process_buffer_set_pipelined(pb, pipelined));
        std::vector<float> out((size_t)frames);
        float* outs[1] = {out.data()};
        std::vector<double> us;
        us.reserve((size_t)blocks);
        double peak = 0;
        int misses = 0;
        auto next = Clock::now();
        for (int b = 0; b < blocks + 50; ++b)
        {
          // the host's callback arrives once per block period; a call that overruns makes the next one late (it starts at once)
          while (Clock::now() < next) {}
          const auto t0 = Clock::now();
          eng.check(mlgpu_process_buffer_process(pb, nullptr, outs, frames, onVectors, &inst));
          const auto t1 = Clock::now();
          const double d = std::chrono::duration<double, std::micro>(t1 - t0).count();
          if (b >= 50)
          {
            us.push_back(d);
            if (d > periodUs) ++misses;
          }
          for (float x : out) peak = std::max(peak, (double)std::fabs(x));
          next = std::max(next + std::chrono::duration_cast<Clock::duration>(std::chrono::duration<double, std::micro>(periodUs)), t1);
        }
        std::sort(us.begin(), us.end());
        const double p50 = us[us.size() / 2], p99 = us[(size_t)(us.size() * 0.99)], mx = us.back();
        printf("%s  {\"voices\": %zu, \"frames\": %d, \"mode\": \"%s\", \"block_period_us\": %.1f, \"voice_kernel_us_free_running\": %.2f, \"voice_samples_per_s_free_running\": %.4g, "
               "\"realtime_48k_voices_free_running\": %.4g, \"algorithmic_bytes_per_block\": %.0f, \"hbm_frac_voice_kernel\": %.3f, \"call_us_p50\": %.1f, \"call_us_p99\": %.1f, "
               "\"call_us_max\": %.1f, \"p99_over_period\": %.3f, \"deadline_misses\": %d, \"latency_frames\": %zu, \"output_peak\": %.3g}",
               first ? "" : ",\n", V, frames, pipelined ? "pipelined" : "synchronous", periodUs, kernelUs, (double)V * frames / (kernelUs * 1e-6),
               (double)V * frames / (kernelUs * 1e-6) / 48000.0, algBytes, hbmFrac, p50, p99, mx, p99 / periodUs, misses, mlgpu_process_buffer_latency_frames(pb), peak);
        first = false;
        fflush(stdout);
        mlgpu_process_buffer_destroy(pb);
      }
      mlgpu_bank_destroy(raw);
    }
  printf("\n]}\n");
  return 0;
}
