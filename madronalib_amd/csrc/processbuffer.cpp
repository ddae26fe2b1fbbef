// processbuffer.cpp — the engine's counterpart of the reference's SignalProcessBuffer
// (source/app/MLSignalProcessBuffer.{h,cpp}): serves a host main loop that asks for arbitrary block sizes, buffers
// inputs and outputs in DSPBuffer rings, and computes in DSPVector-sized chunks.
//
// Same observable behaviour as SignalProcessBuffer::process (:36-90): external inputs are written to the input
// rings; 64-frame vectors are processed until the first output ring holds nFrames; nFrames are read back out; a
// call with nFrames > maxFrames, no outputs or a null output list returns without doing anything (the reference
// silently returns, :42-44; here that is also reported as a status).
// MI355X-first difference: the reference calls its process function once per 64-frame vector; a GPU launch per
// 64 frames would be all overhead, so the K vectors a block needs are gathered first (reading the input rings
// ahead is unobservable: processing never writes to them) and the callback is invoked ONCE with K vectors in
// HBM — one H2D copy, one or a few kernel launches, one D2H copy per block, through pinned staging buffers on the
// engine's stream.
#include <new>
#include <vector>

#include <string.h>

#include "mlgpu_internal.hpp"

extern "C"
{
  mlgpu_dspbuffer* mlgpu_dspbuffer_create(void);
  void mlgpu_dspbuffer_destroy(mlgpu_dspbuffer* b);
  size_t mlgpu_dspbuffer_resize(mlgpu_dspbuffer* b, int sizeInSamples);
  size_t mlgpu_dspbuffer_read_available(mlgpu_dspbuffer* b);
  size_t mlgpu_dspbuffer_size(mlgpu_dspbuffer* b);
  void mlgpu_dspbuffer_write(mlgpu_dspbuffer* b, const float* src, size_t samples);
  size_t mlgpu_dspbuffer_read(mlgpu_dspbuffer* b, float* dst, size_t samples);
  int mlgpu_dspbuffer_read_vector(mlgpu_dspbuffer* b, float* dst64);
}

struct mlgpu_process_buffer
{
  mlgpu_engine* e{nullptr};
  std::vector<mlgpu_dspbuffer*> in, out;
  size_t maxFrames{0}, maxVectors{0};
  // two sets of staging buffers: pinned host and device, each [(nIn + nOut)][maxVectors * 64]. The synchronous mode uses
  // set 0 only; the pipelined mode alternates, so block k + 1 is gathered and uploaded while block k is still on the device
  struct Stage
  {
    float* h{nullptr};
    float* d{nullptr};
    hipEvent_t done{nullptr};
    size_t K{0};          // vectors in flight in this set (0: nothing pending)
  } stage[2];
  int cur{0};
  bool pipelined{false};
  size_t latency{0};      // pipelined: frames of silence the output rings were primed with
  size_t syncRingSize{0}; // size of an output ring of the synchronous mode (the K rule saturates there)
  std::vector<const float*> d_in;
  std::vector<float*> d_out;
};

namespace
{
// what was submitted in stage `sg` has come back: move it into the output rings, one DSPVector at a time as the reference
int retire(mlgpu_process_buffer* p, mlgpu_process_buffer::Stage& sg)
{
  if (!sg.K) return MLGPU_OK;
  if (hipEventSynchronize(sg.done) != hipSuccess)
  {
    p->e->lastError = "process_buffer_process: waiting for a block";
    return MLGPU_ERR_HIP;
  }
  const size_t nIn = p->in.size(), nOut = p->out.size(), chan = sg.K * MLGPU_FLOATS_PER_DSPVECTOR;
  for (size_t k = 0; k < sg.K; ++k)
    for (size_t c = 0; c < nOut; ++c)
      mlgpu_dspbuffer_write(p->out[c], sg.h + (nIn + c) * chan + k * MLGPU_FLOATS_PER_DSPVECTOR, MLGPU_FLOATS_PER_DSPVECTOR);
  sg.K = 0;
  return MLGPU_OK;
}
}  // namespace

extern "C"
{
  int mlgpu_process_buffer_destroy(mlgpu_process_buffer* p)
  {
    if (!p) return MLGPU_ERR_INVALID;
    if (p->e && p->e->recording)
    {
      p->e->lastError = "process_buffer_destroy waits for the device: not while recording a sequence";
      return MLGPU_ERR_INVALID;
    }
    if (p->e)
    {
      hipSetDevice(p->e->device);
      hipStreamSynchronize(p->e->stream);
    }
    for (auto* b : p->in) mlgpu_dspbuffer_destroy(b);
    for (auto* b : p->out) mlgpu_dspbuffer_destroy(b);
    for (auto& sg : p->stage)
    {
      if (sg.h) hipHostFree(sg.h);
      if (sg.d) hipFree(sg.d);
      if (sg.done) hipEventDestroy(sg.done);
    }
    delete p;
    return MLGPU_OK;
  }

  int mlgpu_process_buffer_create(mlgpu_engine* e, size_t nInputs, size_t nOutputs, size_t maxFrames, mlgpu_process_buffer** out)
  {
    if (!e || !out) return MLGPU_ERR_INVALID;
    *out = nullptr;
    // rings are sized with int arithmetic, as the reference's DSPBuffer (MLDSPBuffer.h:73-100): 2^30 frames is the limit
    if (maxFrames == 0 || maxFrames > ((size_t)1 << 30) || nInputs > 64 || nOutputs > 64)
    {
      e->lastError = "process_buffer_create: bad sizes";
      return MLGPU_ERR_INVALID;
    }
    mlgpu_process_buffer* p = new (std::nothrow) mlgpu_process_buffer();
    if (!p) return MLGPU_ERR_OOM;
    p->e = e;
    p->maxFrames = maxFrames;
    // a block of maxFrames needs at most ceil(maxFrames / 64) new vectors (plus one when the rings are out of phase)
    p->maxVectors = (maxFrames + MLGPU_FLOATS_PER_DSPVECTOR - 1) / MLGPU_FLOATS_PER_DSPVECTOR + 1;
    for (size_t i = 0; i < nInputs + nOutputs; ++i)
    {
      mlgpu_dspbuffer* b = mlgpu_dspbuffer_create();
      // SignalProcessBuffer sizes every ring to maxFrames (:24,30). Same size here, so the same samples are dropped
      // when a host overdrives them (a ring that is full overwrites its oldest data, MLDSPBuffer.h:162-167)
      if (!b || mlgpu_dspbuffer_resize(b, (int)maxFrames) == 0)
      {
        if (b) mlgpu_dspbuffer_destroy(b);
        mlgpu_process_buffer_destroy(p);
        return MLGPU_ERR_OOM;
      }
      (i < nInputs ? p->in : p->out).push_back(b);
    }
    const size_t floats = (nInputs + nOutputs) * p->maxVectors * MLGPU_FLOATS_PER_DSPVECTOR;
    p->syncRingSize = p->out.empty() ? 0 : mlgpu_dspbuffer_size(p->out[0]);
    hipError_t err = hipSetDevice(e->device);
    for (auto& sg : p->stage)
    {
      if (err == hipSuccess) err = hipHostMalloc((void**)&sg.h, sizeof(float) * (floats + 4), hipHostMallocDefault);
      if (err == hipSuccess) err = hipMalloc((void**)&sg.d, sizeof(float) * (floats + 4));
      if (err == hipSuccess) err = hipEventCreateWithFlags(&sg.done, hipEventDisableTiming);
    }
    if (err != hipSuccess)
    {
      e->lastError = std::string("process_buffer_create: ") + hipGetErrorString(err);
      mlgpu_process_buffer_destroy(p);
      return err == hipErrorOutOfMemory ? MLGPU_ERR_OOM : MLGPU_ERR_HIP;
    }
    p->d_in.resize(nInputs);
    p->d_out.resize(nOutputs);
    *out = p;
    return MLGPU_OK;
  }

  // Pipelined mode: every process call returns at once with the audio the PREVIOUS calls computed, while its own block is
  // uploaded, processed and downloaded behind the host's back - the copy of block k + 1 overlaps the kernels of block k
  // and the host thread never waits for a kernel it has just launched. The price is a fixed delay of
  // mlgpu_process_buffer_latency_frames() frames (one maximum block rounded up to DSPVectors), which a plug-in host
  // compensates like any other reported latency; the output stream is exactly the synchronous mode's, delayed by it.
  int mlgpu_process_buffer_set_pipelined(mlgpu_process_buffer* p, int on)
  {
    if (!p) return MLGPU_ERR_INVALID;
    if ((on != 0) == p->pipelined) return MLGPU_OK;
    for (auto& sg : p->stage)
      if (int st = retire(p, sg)) return st;
    p->pipelined = on != 0;
    p->latency = p->pipelined ? p->maxVectors * MLGPU_FLOATS_PER_DSPVECTOR : 0;
    // rings start again: big enough for the delay plus one block in flight, primed with the delay's worth of silence
    std::vector<float> zeros(p->latency, 0.f);
    for (auto* b : p->out)
    {
      if (mlgpu_dspbuffer_resize(b, (int)(p->pipelined ? 2 * p->latency + p->maxFrames : p->maxFrames)) == 0) return MLGPU_ERR_OOM;
      if (p->latency) mlgpu_dspbuffer_write(b, zeros.data(), p->latency);
    }
    return MLGPU_OK;
  }
  size_t mlgpu_process_buffer_latency_frames(mlgpu_process_buffer* p) { return p ? p->latency : 0; }

  int mlgpu_process_buffer_process(mlgpu_process_buffer* p, const float* const* inputs, float* const* outputs, int nFrames,
                                   mlgpu_process_vectors_fn fn, void* user)
  {
    if (!p || !fn) return MLGPU_ERR_INVALID;
    mlgpu_engine* e = p->e;
    if (e->recording)
    {
      e->lastError = "process_buffer_process copies through the host: not while recording a sequence";
      return MLGPU_ERR_INVALID;
    }
    const size_t nIn = p->in.size(), nOut = p->out.size();
    // the reference returns silently in these cases (MLSignalProcessBuffer.cpp:42-44)
    if (nOut < 1 || !outputs || nFrames < 0 || (size_t)nFrames > p->maxFrames)
    {
      e->lastError = "process_buffer_process: no outputs / null output list / nFrames > maxFrames (nothing done, as the reference)";
      return MLGPU_ERR_INVALID;
    }
    for (size_t c = 0; c < nIn; ++c)
      if (inputs && inputs[c]) mlgpu_dspbuffer_write(p->in[c], inputs[c], (size_t)nFrames);

    // how many times the reference's `while (outputBuffers_[0].getReadAvailable() < externalFrames)` loop would run:
    // every pass adds one vector to the output ring, which saturates at its size. Pipelined mode: the rings hold `latency`
    // more frames than the synchronous mode's would, some of them still in flight - the count the rule sees is the same.
    size_t inFlight = 0;
    for (auto& sg : p->stage) inFlight += sg.K * MLGPU_FLOATS_PER_DSPVECTOR;
    size_t have = mlgpu_dspbuffer_read_available(p->out[0]) + inFlight - p->latency;
    size_t K = 0;
    while (have < (size_t)nFrames && K < p->maxVectors)
    {
      have += MLGPU_FLOATS_PER_DSPVECTOR;
      if (have > p->syncRingSize) have = p->syncRingSize;
      ++K;
    }
    mlgpu_process_buffer::Stage& sg = p->stage[p->cur];
    if (K > 0)
    {
      if (int st = retire(p, sg)) return st;  // (pipelined: this set was submitted two calls ago and has long landed)
      const size_t chan = K * MLGPU_FLOATS_PER_DSPVECTOR;  // floats per channel in this block
      for (size_t c = 0; c < nIn; ++c)
        for (size_t k = 0; k < K; ++k)  // one DSPVector at a time: zeros when the ring runs dry (DSPBuffer::read(), :257)
          mlgpu_dspbuffer_read_vector(p->in[c], sg.h + c * chan + k * MLGPU_FLOATS_PER_DSPVECTOR);
      hipError_t err = hipSetDevice(e->device);
      if (err == hipSuccess && nIn) err = hipMemcpyAsync(sg.d, sg.h, sizeof(float) * nIn * chan, hipMemcpyHostToDevice, e->stream);
      if (err != hipSuccess)
      {
        e->lastError = std::string("process_buffer_process (H2D): ") + hipGetErrorString(err);
        return MLGPU_ERR_HIP;
      }
      for (size_t c = 0; c < nIn; ++c) p->d_in[c] = sg.d + c * chan;
      for (size_t c = 0; c < nOut; ++c) p->d_out[c] = sg.d + (nIn + c) * chan;
      const int st = fn(user, K, p->d_in.data(), p->d_out.data());
      if (st != MLGPU_OK) return st;
      err = hipMemcpyAsync(sg.h + nIn * chan, sg.d + nIn * chan, sizeof(float) * nOut * chan, hipMemcpyDeviceToHost, e->stream);
      if (err == hipSuccess) err = hipEventRecord(sg.done, e->stream);
      if (err != hipSuccess)
      {
        e->lastError = std::string("process_buffer_process (D2H): ") + hipGetErrorString(err);
        return MLGPU_ERR_HIP;
      }
      sg.K = K;
    }
    if (p->pipelined)
    {
      // the block submitted by the PREVIOUS call has had a whole host block of time to come back; this call's block
      // stays in flight while the host goes on. The rings were primed with `latency` frames, so they cannot run dry.
      if (K > 0) p->cur ^= 1;
      if (int st = retire(p, p->stage[p->cur])) return st;
    }
    else if (int st = retire(p, sg))
      return st;
    for (size_t c = 0; c < nOut; ++c)
      if (outputs[c]) mlgpu_dspbuffer_read(p->out[c], outputs[c], (size_t)nFrames);
    return MLGPU_OK;
  }
}
