// tools/spread_probe.hip - developer tool (GPU): WHY does the headline launch take 323 .. 546 us for identical work?
// The product's chain_kernel_body<Chain<SawGen, Bandpass, Gain>> (from the product headers, same flags) inside a wrapper that stamps,
// per wavefront, the constant 100 MHz clock (s_memrealtime) and the shader-clock counter (s_memtime) at entry and exit, with HW_ID and
// XCC_ID. Per launch the tool writes: the HIP-event duration, the device-side span, the gap to the previous launch and - per XCD - when
// its first / last wavefront started and ended, its median wavefront life and the shader clock it held (d memtime / d realtime).
// A host thread samples the board's power / sclk / mclk / temperature from sysfs beside it.
//   build: see tools/spread_probe.sh        run: spread_probe --tag base [--V n] [--T n] [--nbuf n] [--launches n] [--per_step n]
//          [--gap_us n] [--sync_each] [--out dir] [--dump 3]
#include <hip/hip_runtime.h>
#include <ctype.h>
#include <dirent.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "../madronalib_amd/csrc/mldsp_kernels.hpp"

using namespace mldev;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

using CH3 = Chain<MLGPU_PROC_SAW_GEN, MLGPU_PROC_BANDPASS, MLGPU_PROC_GAIN>;
using CHG = Chain<MLGPU_PROC_GAIN>;  // what bench.py's "store_ceiling" runs: the same store stream next to one multiply
constexpr int kStampWords = 8;  // per wavefront: rt0, rt1, sc0, sc1, hw_id, xcc_id, -, -

template <class CH>
__global__ __launch_bounds__(kChainBlock) void probe_kernel(const ChainArgs a, unsigned long long* stamps)
{
  const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long sc0 = __builtin_amdgcn_s_memtime();
  chain_kernel_body<CH, false>(a);
  const unsigned long long sc1 = __builtin_amdgcn_s_memtime();
  const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
  if (stamps && (threadIdx.x & 63) == 0)
  {
    unsigned long long* s = stamps + ((size_t)blockIdx.x * (kChainBlock / 64) + (threadIdx.x >> 6)) * kStampWords;
    s[0] = rt0;
    s[1] = rt1;
    s[2] = sc0;
    s[3] = sc1;
    s[4] = (unsigned)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));   // HW_REG_HW_ID
    s[5] = (unsigned)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));  // HW_REG_XCC_ID
  }
}
// the product kernel as it ships (no stamps): what the events time when --plain
template <class CH>
__global__ __launch_bounds__(kChainBlock) void plain_kernel(const ChainArgs a) { chain_kernel_body<CH, false>(a); }

static long long now_ns()
{
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (long long)ts.tv_sec * 1000000000ll + ts.tv_nsec;
}

// ---- sysfs sampler ----
struct Probe { std::string name, path; };
static std::vector<Probe> find_probes(const char* busId)
{
  std::vector<Probe> out;
  // the device this process computes on, by its PCI address (the box may show other tenants' boards under /sys/class/drm)
  {
    std::string base = std::string("/sys/bus/pci/devices/") + busId + "/hwmon";
    DIR* d = opendir(base.c_str());
    if (d)
    {
      while (dirent* e = readdir(d))
      {
        if (strncmp(e->d_name, "hwmon", 5) != 0) continue;
        for (const char* f : {"power1_average", "power1_input", "freq1_input", "freq2_input", "temp1_input", "temp2_input", "temp3_input"})
        {
          std::string p = base + "/" + e->d_name + "/" + f;
          if (access(p.c_str(), R_OK) == 0) out.push_back({std::string("mine.") + f, p});
        }
      }
      closedir(d);
    }
    std::string busy = std::string("/sys/bus/pci/devices/") + busId + "/gpu_busy_percent";
    if (access(busy.c_str(), R_OK) == 0) out.push_back({"mine.busy", busy});
  }
  for (int card = 0; card < 16; ++card)
  {
    char base[256];
    snprintf(base, sizeof base, "/sys/class/drm/card%d/device/hwmon", card);
    DIR* d = opendir(base);
    if (!d) continue;
    while (dirent* e = readdir(d))
    {
      if (strncmp(e->d_name, "hwmon", 5) != 0) continue;
      for (const char* f : {"power1_average", "power1_input", "freq1_input", "freq2_input", "temp1_input", "temp2_input", "temp3_input", "in0_input"})
      {
        std::string p = std::string(base) + "/" + e->d_name + "/" + f;
        if (access(p.c_str(), R_OK) == 0) out.push_back({std::string("card") + std::to_string(card) + "." + f, p});
      }
    }
    closedir(d);
    std::string busy = std::string("/sys/class/drm/card") + std::to_string(card) + "/device/gpu_busy_percent";
    if (access(busy.c_str(), R_OK) == 0) out.push_back({std::string("card") + std::to_string(card) + ".busy", busy});
  }
  return out;
}
static double read_num(const std::string& p)
{
  FILE* f = fopen(p.c_str(), "r");
  if (!f) return NAN;
  double x = NAN;
  if (fscanf(f, "%lf", &x) != 1) x = NAN;
  fclose(f);
  return x;
}

int main(int argc, char** argv)
{
  size_t V = 262144, T = 30;
  int nbuf = 2, launches = 600, perStep = 25, gapUs = 0, dump = 3, warm = 50;
  bool syncEach = false, plain = false, gainOnly = false, l2only = false;
  std::string tag = "base", outDir = "gpurun_out/spread";
  for (int i = 1; i < argc; ++i)
  {
    auto is = [&](const char* s) { return strcmp(argv[i], s) == 0; };
    if (is("--V")) V = (size_t)atoll(argv[++i]);
    else if (is("--T")) T = (size_t)atoll(argv[++i]);
    else if (is("--nbuf")) nbuf = atoi(argv[++i]);
    else if (is("--launches")) launches = atoi(argv[++i]);
    else if (is("--per_step")) perStep = atoi(argv[++i]);
    else if (is("--gap_us")) gapUs = atoi(argv[++i]);
    else if (is("--dump")) dump = atoi(argv[++i]);
    else if (is("--warm")) warm = atoi(argv[++i]);
    else if (is("--sync_each")) syncEach = true;
    else if (is("--plain")) plain = true;
    else if (is("--gain")) gainOnly = true;      // Chain<Gain>: the store stream with (almost) no arithmetic
    else if (is("--l2only")) l2only = true;      // every quad of a voice to the same 16 bytes: the stores are issued, HBM sees 4 MB per launch
    else if (is("--tag")) tag = argv[++i];
    else if (is("--out")) outDir = argv[++i];
    else { printf("unknown option %s\n", argv[i]); return 2; }
  }
  const size_t n = V * T * 64, waves = (V + 63) / 64, blocks = (V + kChainBlock - 1) / kChainBlock;
  std::vector<float> co(4 * V), fr(V);
  for (size_t v = 0; v < V; ++v)
  {
    fr[v] = (float)(55.0 * pow(2.0, 5.0 * v / (double)V) / 48000.0);
    const float omega = fminf(0.45f, 4.f * fr[v]), k = 0.5f;
    const float piOmega = 3.14159265f * omega, s1 = sinf(piOmega), s2 = sinf(2.f * piOmega), nrm = 1.f / (2.f + k * s2);
    co[v] = s2 * nrm; co[V + v] = (-2.f * s1 * s1 - k * s2) * nrm; co[2 * V + v] = (2.f * s1 * s1) * nrm; co[3 * V + v] = 0.25f;
  }
  float *dco, *dfr; uint32_t* dst;
  CK(hipMalloc(&dco, 16 * V)); CK(hipMalloc(&dfr, 4 * V)); CK(hipMalloc(&dst, 12 * V));
  CK(hipMemcpy(dco, co.data(), 16 * V, hipMemcpyHostToDevice)); CK(hipMemcpy(dfr, fr.data(), 4 * V, hipMemcpyHostToDevice));
  CK(hipMemset(dst, 0, 12 * V));
  std::vector<float4*> outs(nbuf);
  for (auto& o : outs) { CK(hipMalloc(&o, 4 * n)); CK(hipMemset(o, 0, 4 * n)); }
  unsigned long long* dStamps = nullptr;
  const size_t stampBytes = waves * kStampWords * 8;
  if (!plain) { CK(hipMalloc(&dStamps, stampBytes * launches)); CK(hipMemset(dStamps, 0, stampBytes * launches)); }
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  auto args = [&](int k) {
    ChainArgs a{};
    a.coeffs = dco; a.state = dst; a.inConst = dfr; a.in = SignalView{nullptr, 0, 0, 0};
    a.out = l2only ? SignalView{outs[k % nbuf], 0, 0, 1} : SignalView{outs[k % nbuf], 16 * V, V, 1};
    a.V = V; a.T = T; a.impulseTable = nullptr; a.flags = 0; a.mix = nullptr; a.mixGains = nullptr;
    return a;
  };
  auto launch = [&](int k, unsigned long long* s) {
    if (gainOnly)
    {
      if (plain) hipLaunchKernelGGL(plain_kernel<CHG>, dim3(blocks), dim3(kChainBlock), 0, st, args(k));
      else hipLaunchKernelGGL(probe_kernel<CHG>, dim3(blocks), dim3(kChainBlock), 0, st, args(k), s);
    }
    else if (plain) hipLaunchKernelGGL(plain_kernel<CH3>, dim3(blocks), dim3(kChainBlock), 0, st, args(k));
    else hipLaunchKernelGGL(probe_kernel<CH3>, dim3(blocks), dim3(kChainBlock), 0, st, args(k), s);
  };
  for (int k = 0; k < warm; ++k) launch(k, nullptr);
  CK(hipStreamSynchronize(st));

  // sampler
  char busId[64] = "";
  int dev = 0;
  CK(hipGetDevice(&dev));
  CK(hipDeviceGetPCIBusId(busId, sizeof busId, dev));
  for (char* c = busId; *c; ++c) *c = (char)tolower(*c);
  std::vector<Probe> probes = find_probes(busId);
  std::atomic<bool> stop{false};
  struct Sample { long long ns; std::vector<double> v; };
  std::vector<Sample> samples;
  std::thread sampler([&] {
    while (!stop.load())
    {
      Sample s{now_ns(), {}};
      for (auto& p : probes) s.v.push_back(read_num(p.path));
      samples.push_back(std::move(s));
      usleep(4000);
    }
  });

  std::vector<hipEvent_t> ev(launches + 1);
  for (auto& e : ev) CK(hipEventCreate(&e));
  std::vector<long long> submitNs(launches), stepStartNs, stepEndNs;
  const long long t0 = now_ns();
  CK(hipEventRecord(ev[0], st));
  for (int k = 0; k < launches; ++k)
  {
    if (k % perStep == 0) stepStartNs.push_back(now_ns());
    submitNs[k] = now_ns();
    launch(k, plain ? nullptr : dStamps + (size_t)k * waves * kStampWords);
    CK(hipEventRecord(ev[k + 1], st));
    if (syncEach || gapUs > 0)
    {
      CK(hipStreamSynchronize(st));
      if (gapUs > 0) { const long long until = now_ns() + 1000ll * gapUs; while (now_ns() < until) {} }
    }
    if ((k + 1) % perStep == 0) { CK(hipStreamSynchronize(st)); stepEndNs.push_back(now_ns()); }
  }
  CK(hipStreamSynchronize(st));
  const long long t1 = now_ns();
  stop.store(true);
  sampler.join();

  std::vector<float> durUs(launches);
  for (int k = 0; k < launches; ++k) { float ms; CK(hipEventElapsedTime(&ms, ev[k], ev[k + 1])); durUs[k] = ms * 1000.f; }
  std::vector<unsigned long long> stamps;
  if (!plain) { stamps.resize((size_t)launches * waves * kStampWords); CK(hipMemcpy(stamps.data(), dStamps, stampBytes * launches, hipMemcpyDeviceToHost)); }

  std::string cmd = "mkdir -p " + outDir;
  if (system(cmd.c_str()) != 0) return 3;
  // ---- per-launch CSV ----
  std::string path = outDir + "/" + tag + "_launches.csv";
  FILE* f = fopen(path.c_str(), "w");
  fprintf(f, "launch,buf,submit_us,event_us,dev_start_us,dev_span_us,gap_before_us,start_skew_us");
  for (int x = 0; x < 8; ++x) fprintf(f, ",x%d_last_end_us,x%d_median_life_us,x%d_mhz", x, x, x);
  fprintf(f, "\n");
  unsigned long long base = 0, prevEnd = 0;
  std::vector<double> spanAll;
  for (int k = 0; k < launches; ++k)
  {
    fprintf(f, "%d,%d,%.1f,%.1f", k, k % nbuf, (submitNs[k] - t0) / 1000.0, durUs[k]);
    if (plain) { fprintf(f, "\n"); continue; }
    const unsigned long long* s = stamps.data() + (size_t)k * waves * kStampWords;
    unsigned long long first = ~0ull, lastStart = 0, lastEnd = 0;
    std::vector<double> life[8], mhz[8];
    unsigned long long xEnd[8] = {0};
    for (size_t w = 0; w < waves; ++w)
    {
      const unsigned long long* q = s + w * kStampWords;
      if (q[1] == 0) continue;
      first = std::min(first, q[0]); lastStart = std::max(lastStart, q[0]); lastEnd = std::max(lastEnd, q[1]);
      const int x = (int)(q[5] & 7);
      xEnd[x] = std::max(xEnd[x], q[1]);
      life[x].push_back((q[1] - q[0]) / 100.0);
      if (q[1] > q[0]) mhz[x].push_back((double)(q[3] - q[2]) / ((q[1] - q[0]) / 100.0));
    }
    if (k == 0) base = first;
    const double span = (lastEnd - first) / 100.0;
    spanAll.push_back(span);
    fprintf(f, ",%.2f,%.2f,%.2f,%.2f", (first - base) / 100.0, span, k ? ((double)first - (double)prevEnd) / 100.0 : 0.0, (lastStart - first) / 100.0);
    prevEnd = lastEnd;
    for (int x = 0; x < 8; ++x)
    {
      auto med = [](std::vector<double>& a) { if (a.empty()) return 0.0; std::nth_element(a.begin(), a.begin() + a.size() / 2, a.end()); return a[a.size() / 2]; };
      fprintf(f, ",%.2f,%.2f,%.1f", xEnd[x] ? (xEnd[x] - first) / 100.0 : 0.0, med(life[x]), med(mhz[x]));
    }
    fprintf(f, "\n");
  }
  fclose(f);
  // ---- sysfs samples ----
  path = outDir + "/" + tag + "_sysfs.csv";
  f = fopen(path.c_str(), "w");
  fprintf(f, "t_us");
  for (auto& p : probes) fprintf(f, ",%s", p.name.c_str());
  fprintf(f, "\n");
  for (auto& s : samples) { fprintf(f, "%.1f", (s.ns - t0) / 1000.0); for (double v : s.v) fprintf(f, ",%.0f", v); fprintf(f, "\n"); }
  fclose(f);
  // ---- raw wave tables of the fastest / median / slowest launches ----
  if (!plain && dump > 0)
  {
    std::vector<int> order(launches);
    for (int k = 0; k < launches; ++k) order[k] = k;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return spanAll[a] < spanAll[b]; });
    std::vector<int> pick{order[0], order[launches / 2], order[launches - 1]};
    for (int i = 0; i < (int)pick.size() && i < dump; ++i)
    {
      const char* nm[] = {"fastest", "median", "slowest"};
      path = outDir + "/" + tag + "_waves_" + nm[i] + ".bin";
      f = fopen(path.c_str(), "wb");
      fwrite(stamps.data() + (size_t)pick[i] * waves * kStampWords, 8, waves * kStampWords, f);
      fclose(f);
      printf("  %s launch = %d (span %.1f us)\n", nm[i], pick[i], spanAll[pick[i]]);
    }
  }
  // ---- summary line ----
  std::vector<float> d = durUs;
  std::sort(d.begin(), d.end());
  double mean = 0;
  for (float x : d) mean += x;
  mean /= launches;
  double var = 0;
  for (float x : d) var += (x - mean) * (x - mean);
  const double bytes = (double)V * (T * 256.0 + (gainOnly ? 8.0 : 44.0)) * (l2only ? 0.0 : 1.0) + (l2only ? 16.0 * V : 0.0);
  if (gainOnly || l2only) tag += gainOnly ? (l2only ? " (gain, l2only)" : " (gain)") : " (l2only)";
  printf("[pci %s] ", busId);
  printf("%s: V %zu T %zu nbuf %d launches %d gap_us %d sync_each %d plain %d | event us min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f mean %.1f sd %.1f | "
         "mean frac of 8 TB/s %.3f, p10 %.3f | wall %.1f ms for all, %zu sysfs samples of %zu probes\n",
         tag.c_str(), V, T, nbuf, launches, gapUs, (int)syncEach, (int)plain, d[0], d[launches / 10], d[launches / 2], d[launches * 9 / 10], d[launches - 1], mean,
         sqrt(var / launches), bytes / (mean * 1e-6) / 8e12, bytes / (d[launches / 10] * 1e-6) / 8e12, (t1 - t0) / 1e6, samples.size(), probes.size());
  return 0;
}
