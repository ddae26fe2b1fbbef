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
  float* h_stage{nullptr};  // pinned: [(nIn + nOut)][maxVectors * 64]
  float* d_stage{nullptr};  // device: same shape
  std::vector<const float*> d_in;
  std::vector<float*> d_out;
};

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
    if (p->h_stage) hipHostFree(p->h_stage);
    if (p->d_stage) hipFree(p->d_stage);
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
    hipError_t err = hipSetDevice(e->device);
    if (err == hipSuccess) err = hipHostMalloc((void**)&p->h_stage, sizeof(float) * (floats + 4), hipHostMallocDefault);
    if (err == hipSuccess) err = hipMalloc((void**)&p->d_stage, sizeof(float) * (floats + 4));
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
    // every pass adds one vector to the output ring, which saturates at its size
    size_t have = mlgpu_dspbuffer_read_available(p->out[0]);
    const size_t ringSize = mlgpu_dspbuffer_size(p->out[0]);
    size_t K = 0;
    while (have < (size_t)nFrames && K < p->maxVectors)
    {
      have += MLGPU_FLOATS_PER_DSPVECTOR;
      if (have > ringSize) have = ringSize;
      ++K;
    }
    if (K > 0)
    {
      const size_t chan = K * MLGPU_FLOATS_PER_DSPVECTOR;  // floats per channel in this block
      for (size_t c = 0; c < nIn; ++c)
        for (size_t k = 0; k < K; ++k)  // one DSPVector at a time: zeros when the ring runs dry (DSPBuffer::read(), :257)
          mlgpu_dspbuffer_read_vector(p->in[c], p->h_stage + c * chan + k * MLGPU_FLOATS_PER_DSPVECTOR);
      hipError_t err = hipSetDevice(e->device);
      if (err == hipSuccess && nIn) err = hipMemcpyAsync(p->d_stage, p->h_stage, sizeof(float) * nIn * chan, hipMemcpyHostToDevice, e->stream);
      if (err != hipSuccess)
      {
        e->lastError = std::string("process_buffer_process (H2D): ") + hipGetErrorString(err);
        return MLGPU_ERR_HIP;
      }
      for (size_t c = 0; c < nIn; ++c) p->d_in[c] = p->d_stage + c * chan;
      for (size_t c = 0; c < nOut; ++c) p->d_out[c] = p->d_stage + (nIn + c) * chan;
      const int st = fn(user, K, p->d_in.data(), p->d_out.data());
      if (st != MLGPU_OK) return st;
      err = hipMemcpyAsync(p->h_stage + nIn * chan, p->d_stage + nIn * chan, sizeof(float) * nOut * chan, hipMemcpyDeviceToHost, e->stream);
      if (err == hipSuccess) err = hipStreamSynchronize(e->stream);
      if (err != hipSuccess)
      {
        e->lastError = std::string("process_buffer_process (D2H): ") + hipGetErrorString(err);
        return MLGPU_ERR_HIP;
      }
      for (size_t k = 0; k < K; ++k)  // one DSPVector at a time, as the reference writes them
        for (size_t c = 0; c < nOut; ++c)
          mlgpu_dspbuffer_write(p->out[c], p->h_stage + (nIn + c) * chan + k * MLGPU_FLOATS_PER_DSPVECTOR, MLGPU_FLOATS_PER_DSPVECTOR);
    }
    for (size_t c = 0; c < nOut; ++c)
      if (outputs[c]) mlgpu_dspbuffer_read(p->out[c], outputs[c], (size_t)nFrames);
    return MLGPU_OK;
  }
}
