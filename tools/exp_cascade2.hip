// tools/exp_cascade2.hip — round-3 lab for config 4 (8 x Lopass, streamed in and out): the product's one-lane-per-channel
// cascade_kernel against cascade_lanes_kernel (2 or 4 lanes per channel, DPP hand-over between stage groups), bit-compared
// on outputs and final state, two launches each (the launch boundary is part of the check). Developer tool.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize \
//         -mllvm -amdgpu-sched-strategy=max-ilp tools/exp_cascade2.hip -o /tmp/exp_cascade2
//   /tmp/exp_cascade2 [channels] [vectors] [rounds]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <functional>
#include <string>
#include <vector>

#include "../madronalib_amd/csrc/mldsp_kernels.hpp"

using namespace mldev;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <class F>
float timeit(F f, int reps)
{
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps;
}

template <int KIND, int N, int LPC, int R, int MINW>
void launchLanes(const ChainArgs& a)
{
  const unsigned cpb = kChainBlock / LPC;
  hipLaunchKernelGGL((cascade_lanes_kernel<KIND, N, LPC, R, MINW, true>), dim3((unsigned)((a.V + cpb - 1) / cpb)), dim3(kChainBlock), 0, 0, a);
}

int main(int argc, char** argv)
{
  const size_t V = argc > 1 ? (size_t)atol(argv[1]) : 131072, T = argc > 2 ? (size_t)atol(argv[2]) : 32, n = V * T * 64;
  const int rounds = argc > 3 ? atoi(argv[3]) : 7;
  // mode 0: the real streams. 1: input and output rows collapsed onto one row each (strideT = strideQ = 0): the same
  // instructions, but every access hits the same 16 B x V, i.e. the caches - what the kernel costs without HBM.
  // 2: only the output collapsed (reads stream), 3: only the input collapsed (writes stream).
  const int mode = argc > 4 ? atoi(argv[4]) : 0;
  constexpr int K = MLGPU_PROC_LOPASS;
  std::vector<float> co(24 * V), x(n);
  for (int s = 0; s < 8; ++s)
  {
    const float omega = 0.02f * (s + 1), k = 0.7f;
    const float piOmega = 3.14159265f * omega, s1 = sinf(piOmega), s2 = sinf(2.f * piOmega), nrm = 1.f / (2.f + k * s2);
    for (size_t v = 0; v < V; ++v)
    {
      const float w = 1.f + 0.1f * (float)(v % 97) / 97.f;  // channels differ
      co[(3 * s) * V + v] = s2 * nrm * w; co[(3 * s + 1) * V + v] = (-2.f * s1 * s1 - k * s2) * nrm; co[(3 * s + 2) * V + v] = (2.f * s1 * s1) * nrm;
    }
  }
  uint32_t seed = 12345;
  for (size_t i = 0; i < n; ++i) { seed = seed * 0x0019660Du + 0x3C6EF35Fu; uint32_t t = ((seed >> 9) & 0x7FFFFF) | 0x3F800000; float f; memcpy(&f, &t, 4); x[i] = f * 2.f - 3.f; }
  float* dco; uint32_t* dst; float4 *din, *out0, *out1;
  CK(hipMalloc(&dco, 96 * V)); CK(hipMalloc(&dst, 64 * V)); CK(hipMalloc(&din, 4 * n)); CK(hipMalloc(&out0, 4 * n)); CK(hipMalloc(&out1, 4 * n));
  CK(hipMemcpy(dco, co.data(), 96 * V, hipMemcpyHostToDevice)); CK(hipMemcpy(din, x.data(), 4 * n, hipMemcpyHostToDevice));
  std::vector<uint32_t> ref(n), got(n), st0(16 * V), st1(16 * V);
  struct Var { std::string name; std::function<void(const ChainArgs&)> launch; std::vector<float> ms; size_t bad; };
  std::vector<Var> vars;
  auto add = [&](const char* name, std::function<void(const ChainArgs&)> f) { vars.push_back({name, f, {}, 0}); };
  add("product: 1 lane/channel", [&](const ChainArgs& a) {
    hipLaunchKernelGGL((cascade_kernel<Chain<>, K, 8, true>), dim3((unsigned)((a.V + 255) / 256)), dim3(256), 0, 0, a); });
  add("lanes LPC=2 R=4 w4", [&](const ChainArgs& a) { launchLanes<K, 8, 2, 4, 4>(a); });
  add("lanes LPC=2 R=8 w4", [&](const ChainArgs& a) { launchLanes<K, 8, 2, 8, 4>(a); });
  add("lanes LPC=4 R=4 w8", [&](const ChainArgs& a) { launchLanes<K, 8, 4, 4, 8>(a); });
  add("lanes LPC=4 R=8 w6", [&](const ChainArgs& a) { launchLanes<K, 8, 4, 8, 6>(a); });
  add("lanes LPC=1 R=4 w2", [&](const ChainArgs& a) { launchLanes<K, 8, 1, 4, 2>(a); });
  add("lanes LPC=1 R=8 w2", [&](const ChainArgs& a) { launchLanes<K, 8, 1, 8, 2>(a); });
  add("lanes LPC=1 R=16 w2", [&](const ChainArgs& a) { launchLanes<K, 8, 1, 16, 2>(a); });
  add("lanes LPC=2 R=16 w4", [&](const ChainArgs& a) { launchLanes<K, 8, 2, 16, 4>(a); });
  auto mkArgs = [&](float4* out) {
    ChainArgs a{};
    a.coeffs = dco; a.state = dst; a.inConst = nullptr;
    a.in = SignalView{din, 16 * V, V, 1};   // QUAD: t*16V + q*V + v
    a.out = SignalView{out, 16 * V, V, 1};
    if (mode == 1 || mode == 3) a.in.strideT = a.in.strideQ = 0;
    if (mode == 1 || mode == 2) a.out.strideT = a.out.strideQ = 0;
    a.V = V; a.T = T; a.impulseTable = nullptr; a.flags = 0;
    return a;
  };
  for (size_t i = 0; i < vars.size(); ++i)
  {
    const ChainArgs a = mkArgs(i == 0 ? out0 : out1);
    CK(hipMemset(dst, 0, 64 * V));
    CK(hipMemset(a.out.base, 0xFF, 4 * n));
    vars[i].launch(a);
    vars[i].launch(a);  // second launch continues from carried state: checks the launch boundary
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(got.data(), a.out.base, 4 * n, hipMemcpyDeviceToHost));
    CK(hipMemcpy(st1.data(), dst, 64 * V, hipMemcpyDeviceToHost));
    if (i == 0) { ref = got; st0 = st1; }
    else
    {
      for (size_t j = 0; j < n; ++j) vars[i].bad += (got[j] != ref[j]);
      for (size_t j = 0; j < 16 * V; ++j) vars[i].bad += (st1[j] != st0[j]);
    }
  }
  for (int r = 0; r < rounds; ++r)
    for (auto& v : vars)
    {
      int k = 0;
      const ChainArgs a0 = mkArgs(out0), a1 = mkArgs(out1);
      v.ms.push_back(timeit([&] { v.launch((k++ & 1) ? a1 : a0); }, 20));
    }
  printf("# mode %d%s\n", mode, mode ? " (collapsed rows: mismatches are expected, GB/s are nominal)" : "");
  printf("# %zu channels x %zu DSPVectors, 8 x Lopass, QUAD in and out; algorithmic bytes per launch %.4f GB\n", V, T, (8.0 * n + 4.0 * V * (24 + 16 + 16)) / 1e9);
  for (auto& v : vars)
  {
    std::sort(v.ms.begin(), v.ms.end());
    const float mn = v.ms.front(), md = v.ms[v.ms.size() / 2];
    const double bytes = 8.0 * n + 4.0 * V * (24 + 16 + 16);
    printf("%-26s min %.4f ms (%.0f GB/s)  median %.4f ms (%.0f GB/s = %.3f of 8 TB/s, %.3e ch-smp/s)  mismatches %zu\n", v.name.c_str(), mn,
           bytes / mn / 1e6, md, bytes / md / 1e6, bytes / md / 1e6 / 8000.0, n / md * 1e3, v.bad);
  }
  return 0;
}
