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
// host-side only (read by the cascade launchers, chains.hip): which form of an SVF cascade to run (mlgpu_engine_set_cascade_lanes).
// 0 = by bank size, 1 / 2 / 4 = that many wavefront lanes per channel, 7 = the round-2 kernel (cascade_kernel)
#define MLGPU_KFLAG_CASCADE_SHIFT 8
#define MLGPU_KFLAG_CASCADE_MASK (7u << MLGPU_KFLAG_CASCADE_SHIFT)

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
  float* mix;      // chain_mix_kernel: the rows of 64-voice group sums, [(group * T + t) * 64 + sample] (mlgpu_mixdown's first stage); else unused
  const float* mixGains;  // ... and the per-voice gains the voices are scaled by before (mlgpu_mixdown's d_gains), or nullptr
};

// EventsToSignals settings the device needs (events.hip, mldsp_events.hpp)
struct E2SSettings
{
  double sr;
  float pitchBendRange, mpePitchBendRange, driftAmount;
  int32_t pitchGlideSamples;           // sr * pitchGlideTimeInSeconds (:90)
  int32_t glideVectors;  float glideDy;        // bend / mod / x / y / z / controllers: sr * 0.02 s (:97-101, 275)
  int32_t driftGlideVectors;  float driftGlideDy;  // sr * 8 s (:103)
  int32_t ctlGlideVectors;  float ctlGlideDy;      // SmoothedController: int(sr * 0.02 s) samples (:274-275) — truncated first
  int32_t mpe;                         // protocol
};

// the EventsToSignals object whose pitch / gate rows are source nodes of a graph (mlgpu_graph_bind_events); state == nullptr: none
struct EventsDev
{
  uint32_t* state;           // [kStateWords][lanes]
  const void* recs;          // mlev::Rec: this launch's records, grouped by lane, time-ordered inside a lane
  uint2* recRange;           // [lanes]: a lane's records of this launch are recs[x .. y)
  size_t lanes;
  E2SSettings s;
  // the launch's control records [T][kCtlRecWords][lanes] and the two side signals (QUAD layout) of e2s_ctl_kernel: what a voice
  // kernel expands to audio rate (mlev::CtlVoice); null for mlgpu_events_process, which writes whole rows
  const uint32_t* ctl;
  const float4* rowP;
  const float4* rowG;
};

// a fused graph kernel: up to MLGPU_GRAPH_MAX_INPUTS (32) streamed inputs, MLGPU_GRAPH_MAX_OUTPUTS (8) outputs, per-voice constants [P][V]
#define MLGPU_GRAPH_MAX_INPUTS 32
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
  EventsDev events;
  unsigned long long* waveClock;  // developer aid (MLGPU_GRAPH_WAVE_CLOCK): [wavefront][4] = start, end (100 MHz), HW_ID, XCC_ID; else nullptr
};

// row `row` of the state memory at the voice whose byte offset in a row is lane4: the row's address is wave-uniform (scalar
// arithmetic), the lane's place a 32-bit offset on it - the memory instruction's own addressing, no vector add (graph.hip: stateRef)
__device__ __forceinline__ uint32_t* state_row(const GraphArgs& a, size_t row, uint32_t lane4)
{
  return (uint32_t*)((char*)(a.state + row * a.V) + lane4);
}
