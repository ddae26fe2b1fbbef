// mlgpu_device_args.hpp — plain-old-data kernel argument blocks shared by the ahead-of-time
// kernels (chains.hip), the run-time generated graph kernels (graph.hip via hiprtc) and the host.
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#include <stddef.h>
#endif
#include <stdint.h>

// signal addressing in float4 units: element (vector t, quad q, voice v) lives at
//   base + t*strideT + q*strideQ + v*strideV        (layouts: mlgpu_layout in include/mlgpu.h)
struct SignalView
{
  float4* base;
  size_t strideT, strideQ, strideV;
};

// bits of the `flags` word every arithmetic kernel receives
#define MLGPU_KFLAG_FLUSH_DENORMALS 1u  // run with f32 denormal sources and results flushed (mlgpu_engine_set_flush_denormals)

struct ChainArgs
{
  const float* coeffs;   // [NC][V]
  uint32_t* state;       // [NS][V]
  const float* inConst;  // [V] or nullptr
  SignalView in;         // base == nullptr when no streamed input
  SignalView out;
  size_t V, T;
  const float* impulseTable;
  uint32_t flags;  // MLGPU_KFLAG_*
};

// a fused graph kernel: up to 16 streamed inputs, up to 4 outputs, per-voice constants [P][V]
#define MLGPU_GRAPH_MAX_INPUTS 16
#define MLGPU_GRAPH_MAX_OUTPUTS 8
#define MLGPU_GRAPH_MAX_CONTROLS 8
struct GraphArgs
{
  const float* coeffs;  // [NC][V]
  uint32_t* state;      // [NS][V]
  const float* params;  // [NP][V] per-voice constants (broadcast as DSPVector(f))
  SignalView in[MLGPU_GRAPH_MAX_INPUTS];
  SignalView out[MLGPU_GRAPH_MAX_OUTPUTS];
  const float* ctl[MLGPU_GRAPH_MAX_CONTROLS];  // control-rate inputs, [T][V]: one float per DSPVector per voice
  float* mem;  // delay-line rings of all delay nodes, [sample][V] each
  size_t V, T;
  const float* impulseTable;
  const float* consts;  // live constants (mlgpu_graph_set_live_constants): one float per const node, the same for all voices
  size_t t0;  // DSPVectors processed since the last clear (a Downsample2xFunction region pairs vectors 2k, 2k + 1)
  uint32_t flags;  // MLGPU_KFLAG_*
};
