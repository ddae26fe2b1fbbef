// mlgpu_internal.hpp — shared between the translation units of libmlgpu.so (not installed).
#pragma once
#include <atomic>
#include <functional>
#include <vector>
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/mlgpu.h"
#include "mlgpu_device_args.hpp"

struct mlgpu_engine
{
  int device{0};
  hipStream_t stream{nullptr};
  bool ownsStream{false};
  int cuCount{256};
  std::string lastError;
  float* d_impulseTable{nullptr};  // 17 floats (ImpulseGen windowed sinc), built on the host
  hipEvent_t ev0{nullptr}, ev1{nullptr};
  std::vector<hipEvent_t> lapEvents;  // mlgpu_timer_laps_*: created once, reused
  size_t lapCount{0}, lapMax{0};
  bool jitEnabled{true};  // fuse unknown chains / graphs with hiprtc (mlgpu_engine_set_jit)
  bool strictSvf{false};  // banks and graphs made from now on get kernels compiled with MLGPU_SVF_STRICT 1 (mlgpu_engine_set_strict_svf)
  float* d_mixScratch{nullptr};  // mixdown partial sums, grown on demand
  size_t mixScratchFloats{0};
  unsigned long long* d_validate{nullptr};  // {count, first index} of mlgpu_validate, allocated with the engine
  uint32_t kflags{0};  // MLGPU_KFLAG_* handed to every arithmetic kernel (mlgpu_engine_set_flush_denormals)
  bool recording{false};  // between mlgpu_engine_begin_recording and _end_recording: launches are captured, not run
  int liveSequences{0};   // recorded sequences not yet destroyed: they hold device pointers, so buffers handed out must not move
  // objects destroyed while recorded sequences might still replay launches that read them: freed when the last sequence goes
  // (mlgpu_sequence_destroy) or with the engine
  std::vector<std::function<void()>> deferredFrees;
  void runDeferredFrees()
  {
    std::vector<std::function<void()>> todo;
    todo.swap(deferredFrees);
    for (auto& f : todo) f();
  }
};

struct mlgpu_fence  // mlgpu_engine_signal / mlgpu_engine_wait: a point in one engine's stream that another engine's stream can wait for
{
  int device{0};
  hipEvent_t ev{nullptr};
  std::atomic<bool> signalled{false};  // set by the signalling engine's host thread, read by the waiting engine's
};

struct mlgpu_sequence  // a recorded launch sequence: a hipGraph instantiated once, replayed with one launch
{
  mlgpu_engine* e{nullptr};
  hipGraph_t graph{nullptr};
  hipGraphExec_t exec{nullptr};
  size_t nodes{0};
};

inline SignalView makeView(const float* p, int layout, size_t V, size_t T)
{
  SignalView s;
  s.base = (float4*)p;
  switch (layout)
  {
    case MLGPU_LAYOUT_QUAD: s.strideT = 16 * V; s.strideQ = V; s.strideV = 1; break;
    case MLGPU_LAYOUT_ROWS: s.strideT = 16 * V; s.strideQ = 1; s.strideV = 16; break;
    case MLGPU_LAYOUT_BROADCAST: s.strideT = 16; s.strideQ = 1; s.strideV = 0; break;
    default: /* VOICE_MAJOR */ s.strideT = 16; s.strideQ = 1; s.strideV = 16 * T; break;
  }
  return s;
}

typedef hipError_t (*ChainLauncher)(const ChainArgs& a, hipStream_t stream, int cuCount);

struct ChainEntry
{
  std::vector<int> kinds;
  ChainLauncher launchSignal;  // streamed input
  ChainLauncher launchConst;   // per-voice constant (or no) input
  ChainLauncher launchMixSignal{nullptr}, launchMixConst{nullptr};  // chain_mix_kernel (the voices' sum instead of their signals), where instantiated
  const char* kernelName;  // prefix of the name a profiler shows for the device kernel
  const char* (*kernelNameFor)(size_t V, uint32_t flags){nullptr};  // where the kernel depends on the bank's size (SVF cascades)
  const char* alias;       // e.g. "chain_kernel<SawGen,Bandpass,Gain>"
  int nc, ns;
};

// chains.hip
const ChainEntry* mlgpu_find_chain(const int32_t* kinds, int n);
int mlgpu_proc_nc(int kind);  // -1 if unknown
int mlgpu_proc_ns(int kind);
constexpr int MLGPU_MAX_PROC_STATE = 80;  // LinearGlide: 3 + 64
constexpr int MLGPU_MAX_PROC_COEFFS = 8;
void mlgpu_proc_clear_state(int kind, uint32_t* words /*[ns <= MLGPU_MAX_PROC_STATE]*/, bool cleared);
void mlgpu_proc_default_coeffs(int kind, float* c /*[nc <= MLGPU_MAX_PROC_COEFFS]*/);
bool mlgpu_proc_is_graph_only(int kind);   // vector-rate ramps and delay lines: no chain kernel
int mlgpu_proc_rings(int kind);            // delay rings per voice (0 for processors without delay memory)
uint64_t mlgpu_proc_clear_mask(int kind);  // state words T::clear() resets
bool mlgpu_proc_is_vector_rate(int kind);  // one float per DSPVector in (Interpolator1, LinearGlide): graphs only

// ops.hip
hipError_t mlgpu_launch_op(int op, const void* a, const void* b, const void* c, void* out, size_t n,
                           hipStream_t stream, int cuCount, bool* known, uint32_t flags);
hipError_t mlgpu_launch_op_rows1(int op, const void* a, const void* b64, void* out, size_t nRows,
                                 hipStream_t stream, int cuCount, bool* known, uint32_t flags);
hipError_t mlgpu_launch_row_reduce(int rowop, const float* rows, float* out, size_t nRows,
                                   hipStream_t stream, bool* known, uint32_t flags);
hipError_t mlgpu_launch_layout_convert(const float* src, int srcLayout, float* dst, int dstLayout, size_t V,
                                       size_t T, hipStream_t stream);
hipError_t mlgpu_launch_fill32(uint32_t* dst, uint32_t value, size_t n, hipStream_t stream);
hipError_t mlgpu_launch_validate(const float* x, size_t n, unsigned long long* d_result, hipStream_t stream, int cuCount);
hipError_t mlgpu_launch_rows_map(int rule, long p0, long p1, int sampleRotate, const float* src, size_t srcRows, float* dst,
                                 size_t dstRows, size_t dstOffset, size_t dstStep, size_t count, size_t groups, hipStream_t stream);
hipError_t mlgpu_launch_rows_add(const float* rows, size_t rowsPerGroup, float* out, size_t groups, hipStream_t stream, uint32_t flags);
hipError_t mlgpu_launch_rows_normalize(const float* rows, float* out, size_t nRows, hipStream_t stream, uint32_t flags);
hipError_t mlgpu_launch_rows_index(float* out, size_t rowsPerGroup, size_t groups, hipStream_t stream);
hipError_t mlgpu_launch_mixdown(const float* sig, int layout, size_t V, size_t T, const float* gains, float* partial, float* out,
                                hipStream_t stream, uint32_t flags);
int mlgpu_mixdown_reserve_floats(mlgpu_engine* e, size_t floats);  // capi.hip: the mixdown scratch grown to at least that
hipError_t mlgpu_launch_mixdown_rows(size_t groups, size_t T, float* partial, float* out, hipStream_t stream, uint32_t flags);
hipError_t mlgpu_launch_mixdown_rows_partial(size_t groups, size_t T, float* partial, float* out, int reductions, hipStream_t stream, uint32_t flags);
hipError_t mlgpu_launch_mixdown_stage1(const float* sig, int layout, size_t V, size_t T, const float* gains, float* partial, hipStream_t stream, uint32_t flags);
hipError_t mlgpu_launch_mixdown_groups(const float* sig, int layout, size_t groups, size_t P, size_t T, float* out, int outLayout, hipStream_t stream, uint32_t flags);
hipError_t mlgpu_launch_route(bool demux, bool linear, const float* sel, size_t selElems, const float* const* ins, float* const* outs, int n,
                              size_t nElems, hipStream_t stream, uint32_t flags);

// graph.hip — run-time fused kernels (hiprtc)
// LDS strips of the generated kernels, in floats per WAVEFRONT, for the host's LDS budget (graph.hip does not include the device
// headers; chains.hip asserts they equal mldev::kMixStrip / kGroup16Strip)
constexpr int kHostMixStripFloats = 64 * 20 + 3 * 16 + 16;
constexpr int kHostGroup16StripFloats = 4 * (4 * 80 + 4);
bool mlgpu_jit_chain(mlgpu_engine* e, const int32_t* kinds, int n, void** fnSignal, void** fnConst, std::string& log);  // honours e->strictSvf
hipError_t mlgpu_jit_chain_launch(void* fn, const ChainArgs& a, hipStream_t stream);
bool mlgpu_jit_chain_mix(mlgpu_engine* e, const int32_t* kinds, int n, void** fnSignal, void** fnConst, std::string& log);

// coeffs.cpp
void mlgpu_build_impulse_table(float* out17);

// events.hip, for graph.hip: the host half of an EventsToSignals block (routing, record upload) for a graph kernel that computes
// the pitch and gate rows itself (mlgpu_graph_bind_events)
struct mlgpu_events;
extern "C" int mlgpu_events_abandoned_by_graph(mlgpu_events* ev, void* staging);
extern "C" int mlgpu_events_prepare_for_graph(mlgpu_events* ev, size_t nVectors, int startOffset, EventsDev* dev, void** staging);
extern "C" int mlgpu_events_launched_by_graph(mlgpu_events* ev, void* staging);
extern "C" int mlgpu_events_is_midi(mlgpu_events* ev);
extern "C" mlgpu_engine* mlgpu_events_engine(mlgpu_events* ev);

