// transport.hip — AudioContext::ProcessTime (source/app/MLAudioContext.h:27-57, MLAudioContext.cpp:16-104) for N independent
// contexts: the quarter-note phasor a process function reads through ctx->getBeatPhase() and locks tempo-synced oscillators to
// (TempoLock, MLDSPFilters.h:1478).
//
//   HOST   setTimeAndRate (:16-80): what the host application reports before a processing block - position in quarter notes,
//          tempo, playing or not - turned into the phasor's restart value and slope. Double-precision bookkeeping on one
//          report per context per block; it leaves, per context, "omega is now x" and "the slope is now d".
//   DEVICE processVector (:91-104): the phasor itself, one lane per context - omega_ is a float, the slope a double, the sum is
//          taken in double and rounded back every sample, wrapped above 1. Sequential per context, 64 T steps per launch.
// One signal per context comes out (QUAD layout over N "voices"); a voice graph reads it through an input shared by the voices
// of an instrument (mlgpu_graph_set_input_group).
#include <math.h>
#include <string.h>

#include <new>
#include <vector>

#include "mlgpu_internal.hpp"

namespace
{
struct TimeState  // ProcessTime's members, MLAudioContext.h:44-56
{
  double bpm{0}, sampleRate{0};
  uint64_t samplesSinceStart{0};
  bool playing1{false}, active1{false};
  double dpdt{0};
  size_t samplesSincePreviousTime{0};
  double ppqPos1{-1.}, ppqPhase1{0};
  bool dirty{false}, setOmega{false};
  float omega{0};  // valid when setOmega: between reports the phasor's value lives on the device
};
struct Update
{
  uint32_t index, setOmega;
  float omega, pad;
  double dpdt;
};

__global__ void transport_update_kernel(const Update* u, size_t n, float* omega, double* dpdt)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Update x = u[i];
  if (x.setOmega) omega[x.index] = x.omega;
  dpdt[x.index] = x.dpdt;
}

__global__ __launch_bounds__(256) void transport_kernel(float* omegaState, const double* dpdtState, float* out, size_t N, size_t T)
{
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  float omega = omegaState[i];
  const double dpdt = dpdtState[i];
  f32x4* o = (f32x4*)out + i;
  for (size_t q = 0; q < 16 * T; ++q)
  {
    f32x4 v;
#pragma unroll
    for (int k = 0; k < 4; ++k)  // :93-101
    {
      v[k] = omega;
      omega = (float)((double)omega + dpdt);  // float += double: the sum in double, rounded once
      if (omega > 1.f) omega = omega - 1.f;
    }
    o[q * N] = v;
  }
  omegaState[i] = omega;
}
}  // namespace

struct mlgpu_transport
{
  mlgpu_engine* e{nullptr};
  size_t n{0}, maxVectors{0}, capacityVectors{0}, vectors{0};  // capacityVectors: what d_out was allocated for (>= maxVectors)
  std::vector<TimeState> st;
  std::vector<uint32_t> dirty;
  float* d_omega{nullptr};
  double* d_dpdt{nullptr};
  float* d_out{nullptr};
  struct Staging
  {
    Update* h{nullptr};
    Update* d{nullptr};
    hipEvent_t done{nullptr};
    bool pending{false};
  } stage[2];
  int stageIdx{0};
};

namespace
{
int tfail(mlgpu_transport* t, int st, const char* what)
{
  if (t && t->e) t->e->lastError = what;
  return st;
}
void markDirty(mlgpu_transport* t, size_t i)
{
  if (!t->st[i].dirty) t->dirty.push_back((uint32_t)i);
  t->st[i].dirty = true;
}
void setTimeAndRate(mlgpu_transport* t, size_t i, double ppqPos, double bpmIn, bool isPlaying, double sampleRateIn)  // :16-80
{
  TimeState& s = t->st[i];
  if (isnan(ppqPos) || isinf(ppqPos) || isnan(bpmIn) || isinf(bpmIn)) return;  // :20-26
  s.sampleRate = sampleRateIn;
  s.bpm = bpmIn;
  const bool active = (s.ppqPos1 != ppqPos) && isPlaying;
  const bool justStarted = isPlaying && !s.playing1;
  double ppqPhase = 0.;
  if (active)
  {
    ppqPhase = (ppqPos > 0.f) ? ppqPos - floor(ppqPos) : ppqPos;
    s.omega = (float)ppqPhase;
    if (justStarted)
    {
      s.samplesSinceStart = 0;
      s.omega = 0.f;
      const double dsdt = 1. / s.sampleRate;
      const double minutesPerSample = dsdt / 60.;
      s.dpdt = s.bpm * minutesPerSample;
    }
    else
    {
      double dPhase = ppqPhase - s.ppqPhase1;
      if (dPhase < 0.) dPhase += 1.;
      const double x = dPhase / (double)s.samplesSincePreviousTime;
      s.dpdt = (x < 0.) ? 0. : (x > 1. ? 1. : x);  // ml::clamp, MLDSPScalarMath.h:69-72
    }
  }
  else
  {
    s.omega = -1.f;
    s.dpdt = 0.;
  }
  s.setOmega = true;
  s.ppqPos1 = ppqPos;
  s.ppqPhase1 = ppqPhase;
  s.active1 = active;
  s.playing1 = isPlaying;
  s.samplesSincePreviousTime = 0;
  markDirty(t, i);
}
void clearOne(mlgpu_transport* t, size_t i)  // :82-87
{
  TimeState& s = t->st[i];
  s.dpdt = 0.;
  s.active1 = false;
  s.playing1 = false;
  markDirty(t, i);
}
}  // namespace

extern "C"
{
  int mlgpu_transport_destroy(mlgpu_transport* t)
  {
    if (!t) return MLGPU_ERR_INVALID;
    hipSetDevice(t->e->device);
    hipStreamSynchronize(t->e->stream);
    if (t->d_omega) hipFree(t->d_omega);
    if (t->d_dpdt) hipFree(t->d_dpdt);
    if (t->d_out) hipFree(t->d_out);
    for (mlgpu_transport::Staging& s : t->stage)
    {
      if (s.h) hipHostFree(s.h);
      if (s.d) hipFree(s.d);
      if (s.done) hipEventDestroy(s.done);
    }
    delete t;
    return MLGPU_OK;
  }

  int mlgpu_transport_create(mlgpu_engine* e, size_t n, size_t maxVectors, mlgpu_transport** out)
  {
    if (!e || !out) return MLGPU_ERR_INVALID;
    *out = nullptr;
    if (n == 0 || maxVectors == 0 || n > 0xFFFFFFFFull)
    {
      e->lastError = "transport_create: 1+ contexts, max_vectors = the longest launch (1+)";
      return MLGPU_ERR_INVALID;
    }
    mlgpu_transport* t = new (std::nothrow) mlgpu_transport();
    if (!t) return MLGPU_ERR_OOM;
    t->e = e;
    t->n = n;
    t->maxVectors = t->capacityVectors = maxVectors;
    t->st.resize(n);
    hipError_t err = hipSetDevice(e->device);
    if (err == hipSuccess) err = hipMalloc((void**)&t->d_omega, sizeof(float) * n);
    if (err == hipSuccess) err = hipMalloc((void**)&t->d_dpdt, sizeof(double) * n);
    if (err == hipSuccess) err = hipMalloc((void**)&t->d_out, sizeof(float) * 64 * maxVectors * n);
    for (mlgpu_transport::Staging& s : t->stage)
    {
      if (err == hipSuccess) err = hipMalloc((void**)&s.d, sizeof(Update) * n);
      if (err == hipSuccess) err = hipHostMalloc((void**)&s.h, sizeof(Update) * n);
      if (err == hipSuccess) err = hipEventCreateWithFlags(&s.done, hipEventDisableTiming);
    }
    if (err == hipSuccess) err = hipMemsetAsync(t->d_omega, 0, sizeof(float) * n, e->stream);  // omega_{0}, dpdt_{0}
    if (err == hipSuccess) err = hipMemsetAsync(t->d_dpdt, 0, sizeof(double) * n, e->stream);
    if (err == hipSuccess) err = hipMemsetAsync(t->d_out, 0, sizeof(float) * 64 * maxVectors * n, e->stream);
    if (err != hipSuccess)
    {
      e->lastError = std::string("transport_create: ") + hipGetErrorString(err);
      mlgpu_transport_destroy(t);
      return err == hipErrorOutOfMemory ? MLGPU_ERR_OOM : MLGPU_ERR_HIP;
    }
    *out = t;
    return MLGPU_OK;
  }

  // Longer (or shorter) launches from now on; the phasors go on. The signal buffer - whose pointer callers hold
  // (mlgpu_transport_beat_phase) - is only ever replaced to GROW, and never while a recorded sequence could still replay a
  // launch that reads the old one.
  int mlgpu_transport_reserve(mlgpu_transport* t, size_t maxVectors)
  {
    if (!t || maxVectors == 0) return MLGPU_ERR_INVALID;
    if (maxVectors <= t->capacityVectors)
    {
      t->maxVectors = maxVectors;  // a signal of T vectors is laid out by T, not by the reservation: nothing moves
      return MLGPU_OK;
    }
    if (t->e->recording) return tfail(t, MLGPU_ERR_INVALID, "transport_reserve allocates: not while recording a sequence");
    if (t->e->liveSequences > 0)
      return tfail(t, MLGPU_ERR_INVALID, "transport_reserve would move the beat-phase signal that recorded sequences of this engine may read: destroy them first");
    if (hipSetDevice(t->e->device) != hipSuccess) return tfail(t, MLGPU_ERR_HIP, "hipSetDevice");
    hipStreamSynchronize(t->e->stream);
    float* fresh = nullptr;
    if (hipMalloc((void**)&fresh, sizeof(float) * 64 * maxVectors * t->n) != hipSuccess) return tfail(t, MLGPU_ERR_OOM, "transport_reserve");
    hipMemsetAsync(fresh, 0, sizeof(float) * 64 * maxVectors * t->n, t->e->stream);
    hipFree(t->d_out);
    t->d_out = fresh;  // mlgpu_transport_beat_phase returns the new pointer from now on
    t->maxVectors = t->capacityVectors = maxVectors;
    return MLGPU_OK;
  }

  int mlgpu_transport_set_time_and_rate(mlgpu_transport* t, size_t index, double ppqPos, double bpm, int isPlaying, double sampleRate)
  {
    if (!t) return MLGPU_ERR_INVALID;
    if (index == MLGPU_TRANSPORT_ALL)
    {
      for (size_t i = 0; i < t->n; ++i) setTimeAndRate(t, i, ppqPos, bpm, isPlaying != 0, sampleRate);
      return MLGPU_OK;
    }
    if (index >= t->n) return tfail(t, MLGPU_ERR_RANGE, "transport_set_time_and_rate: no such context");
    setTimeAndRate(t, index, ppqPos, bpm, isPlaying != 0, sampleRate);
    return MLGPU_OK;
  }

  int mlgpu_transport_clear(mlgpu_transport* t, size_t index)
  {
    if (!t) return MLGPU_ERR_INVALID;
    if (index == MLGPU_TRANSPORT_ALL)
    {
      for (size_t i = 0; i < t->n; ++i) clearOne(t, i);
      return MLGPU_OK;
    }
    if (index >= t->n) return tfail(t, MLGPU_ERR_RANGE, "transport_clear: no such context");
    clearOne(t, index);
    return MLGPU_OK;
  }

  int mlgpu_transport_process(mlgpu_transport* t, size_t nVectors)
  {
    if (!t) return MLGPU_ERR_INVALID;
    mlgpu_engine* e = t->e;
    if (nVectors == 0) return MLGPU_OK;
    if (nVectors > t->maxVectors) return tfail(t, MLGPU_ERR_RANGE, "transport_process: more DSPVectors than transport_create reserved");
    // the host half (reports waiting to be uploaded, the sample counters the next report is measured against) cannot be replayed
    if (e->recording) return tfail(t, MLGPU_ERR_INVALID, "transport_process keeps host-side time: not while recording a sequence");
    if (hipSetDevice(e->device) != hipSuccess) return tfail(t, MLGPU_ERR_HIP, "hipSetDevice");
    if (!t->dirty.empty())
    {
      mlgpu_transport::Staging& sg = t->stage[t->stageIdx];
      t->stageIdx ^= 1;
      if (sg.pending && hipEventSynchronize(sg.done) != hipSuccess) return tfail(t, MLGPU_ERR_HIP, "transport_process: waiting for the launch before last");
      sg.pending = false;
      size_t n = 0;
      for (uint32_t i : t->dirty)
      {
        TimeState& s = t->st[i];
        sg.h[n++] = Update{i, s.setOmega ? 1u : 0u, s.omega, 0.f, s.dpdt};
        s.dirty = s.setOmega = false;
      }
      t->dirty.clear();
      if (hipMemcpyAsync(sg.d, sg.h, sizeof(Update) * n, hipMemcpyHostToDevice, e->stream) != hipSuccess) return tfail(t, MLGPU_ERR_HIP, "transport_process: upload");
      hipLaunchKernelGGL(transport_update_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, e->stream, sg.d, n, t->d_omega, t->d_dpdt);
      if (hipEventRecord(sg.done, e->stream) != hipSuccess) return tfail(t, MLGPU_ERR_HIP, "transport_process: event");
      sg.pending = true;
    }
    hipLaunchKernelGGL(transport_kernel, dim3((unsigned)((t->n + 255) / 256)), dim3(256), 0, e->stream, t->d_omega, t->d_dpdt, t->d_out, t->n, nVectors);
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess)
    {
      e->lastError = std::string("transport_process launch: ") + hipGetErrorString(err);
      return MLGPU_ERR_HIP;
    }
    for (TimeState& s : t->st)  // :102-103
    {
      s.samplesSincePreviousTime += MLGPU_FLOATS_PER_DSPVECTOR * nVectors;
      s.samplesSinceStart += MLGPU_FLOATS_PER_DSPVECTOR * nVectors;
    }
    t->vectors = nVectors;
    return MLGPU_OK;
  }

  const float* mlgpu_transport_beat_phase(mlgpu_transport* t) { return t ? t->d_out : nullptr; }
  uint64_t mlgpu_transport_samples_since_start(mlgpu_transport* t, size_t index) { return (t && index < t->n) ? t->st[index].samplesSinceStart : 0; }
  double mlgpu_transport_bpm(mlgpu_transport* t, size_t index) { return (t && index < t->n) ? t->st[index].bpm : 0.; }
}
