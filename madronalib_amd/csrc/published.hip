// published.hip — SignalProcessor::PublishedSignal (source/app/MLSignalProcessor.h:26-105, MLSignalProcessor.cpp:9-38):
// a decimated, frame-major copy of a few channels of a few voices for code outside the DSP calculation (displays).
//
// The reference calls storePublishedSignal(name, DSPVectorArray<CHANNELS>, 64, voice) for every voice in rotation inside
// processVector; writeQuick (:59-83) keeps every (1 << octavesDown)-th frame (no filtering) and appends the kept frames of
// that voice, channels interleaved, to a DSPBuffer: per DSPVector the ring receives [voice][kept frame][channel]. Here the
// signals are device signals of many voices: one gather kernel writes exactly that order for a range of voices and all
// DSPVectors of a launch into a staging buffer, one D2H copy brings it to pinned memory, and the host appends it to the
// same ring type (mlgpu_dspbuffer) with the reference's write granularity (one write per voice per vector, so a ring that
// overflows drops the same samples). read / readLatest / peekLatest are the reference's.
#include <new>
#include <vector>

#include "mlgpu_internal.hpp"

namespace
{
constexpr int kMaxPublishedChannels = 16;

struct PublishArgs
{
  SignalView ch[kMaxPublishedChannels];
  float* out;       // [T][nVoices][framesPerVector][channels]
  size_t firstVoice, nVoices, T;
  int channels, step, first;  // keep sample `first + i * step` of each vector, i < framesPerVector
  int framesPerVector;
};

// one lane per (vector, voice, kept frame); channels are few
__global__ __launch_bounds__(256) void publish_gather_kernel(const PublishArgs a)
{
  const size_t total = a.T * a.nVoices * (size_t)a.framesPerVector;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride)
  {
    const size_t f = i % (size_t)a.framesPerVector;
    const size_t v = (i / (size_t)a.framesPerVector) % a.nVoices;
    const size_t t = i / ((size_t)a.framesPerVector * a.nVoices);
    const int n = a.first + (int)f * a.step;
    for (int j = 0; j < a.channels; ++j)
    {
      const float* quad = (const float*)(a.ch[j].base + t * a.ch[j].strideT + (size_t)(n >> 2) * a.ch[j].strideQ + (a.firstVoice + v) * a.ch[j].strideV);
      a.out[i * (size_t)a.channels + (size_t)j] = quad[n & 3];
    }
  }
}
}  // namespace

struct mlgpu_published_signal
{
  mlgpu_engine* e{nullptr};
  mlgpu_dspbuffer* ring{nullptr};
  size_t maxFrames{0}, maxVoices{0};
  int channels{0}, octavesDown{0};
  int downsampleCtr{0};  // PublishedSignal::downsampleCtr_, carried from call to call
  float* d_stage{nullptr};
  float* h_stage{nullptr};
  size_t stageFloats{0};
};

extern "C"
{
  int mlgpu_published_signal_destroy(mlgpu_published_signal* p)
  {
    if (!p) return MLGPU_ERR_INVALID;
    if (p->e)
    {
      hipSetDevice(p->e->device);
      hipStreamSynchronize(p->e->stream);
    }
    if (p->ring) mlgpu_dspbuffer_destroy(p->ring);
    if (p->d_stage) hipFree(p->d_stage);
    if (p->h_stage) hipHostFree(p->h_stage);
    delete p;
    return MLGPU_OK;
  }

  int mlgpu_published_signal_create(mlgpu_engine* e, int maxFrames, int maxVoices, int channels, int octavesDown, mlgpu_published_signal** out)
  {
    if (!e || !out) return MLGPU_ERR_INVALID;
    *out = nullptr;
    if (maxFrames < 1 || maxVoices < 1 || channels < 1 || channels > kMaxPublishedChannels || octavesDown < 0 || octavesDown > 6)
    {
      e->lastError = "published_signal_create: frames >= 1, voices >= 1, 1..16 channels, 0..6 octaves down";
      return MLGPU_ERR_INVALID;
    }
    mlgpu_published_signal* p = new (std::nothrow) mlgpu_published_signal();
    if (!p) return MLGPU_ERR_OOM;
    p->e = e;
    p->maxFrames = (size_t)maxFrames;
    p->maxVoices = (size_t)maxVoices;
    p->channels = channels;
    p->octavesDown = octavesDown;
    p->ring = mlgpu_dspbuffer_create();
    // buffer_.resize(maxFrames * channels * maxVoices), MLSignalProcessor.cpp:16
    if (!p->ring || mlgpu_dspbuffer_resize(p->ring, maxFrames * channels * maxVoices) == 0)
    {
      mlgpu_published_signal_destroy(p);
      return MLGPU_ERR_OOM;
    }
    *out = p;
    return MLGPU_OK;
  }

  int mlgpu_published_signal_write(mlgpu_published_signal* p, size_t nVectors, const float* const* d_channels, int layout, size_t nVoicesTotal,
                                   size_t firstVoice, size_t nVoices)
  {
    if (!p || !d_channels) return MLGPU_ERR_INVALID;
    mlgpu_engine* e = p->e;
    if (e->recording)
    {
      e->lastError = "published_signal_write waits for the device: not while recording a sequence";
      return MLGPU_ERR_INVALID;
    }
    if (nVectors == 0 || nVoices == 0) return MLGPU_OK;
    if (firstVoice + nVoices > nVoicesTotal || layout < MLGPU_LAYOUT_QUAD || layout > MLGPU_LAYOUT_BROADCAST)
    {
      e->lastError = "published_signal_write: voice range outside the signals / unknown layout";
      return MLGPU_ERR_RANGE;
    }
    // writeQuick (:59-83): the counter is bumped per frame and a frame is kept when it reaches 1 << octavesDown. 64 is a
    // multiple of every step, so each voice's call keeps 64 / step frames and leaves the counter where it found it.
    const int step = 1 << p->octavesDown;
    const int first = step - 1 - p->downsampleCtr;
    const int framesPerVector = MLGPU_FLOATS_PER_DSPVECTOR / step;
    const size_t floats = nVectors * nVoices * (size_t)framesPerVector * (size_t)p->channels;
    if (hipSetDevice(e->device) != hipSuccess) return MLGPU_ERR_HIP;
    if (floats > p->stageFloats)
    {
      hipStreamSynchronize(e->stream);
      if (p->d_stage) hipFree(p->d_stage);
      if (p->h_stage) hipHostFree(p->h_stage);
      p->d_stage = p->h_stage = nullptr;
      p->stageFloats = 0;
      if (hipMalloc((void**)&p->d_stage, floats * sizeof(float)) != hipSuccess || hipHostMalloc((void**)&p->h_stage, floats * sizeof(float)) != hipSuccess)
      {
        e->lastError = "published_signal_write: out of memory for the staging buffers";
        return MLGPU_ERR_OOM;
      }
      p->stageFloats = floats;
    }
    PublishArgs a;
    for (int j = 0; j < p->channels; ++j)
    {
      if (!d_channels[j] || ((uintptr_t)d_channels[j] & 15))
      {
        e->lastError = "published_signal_write: null / misaligned channel signal";
        return MLGPU_ERR_INVALID;
      }
      a.ch[j] = makeView(d_channels[j], layout, nVoicesTotal, nVectors);
    }
    a.out = p->d_stage;
    a.firstVoice = firstVoice;
    a.nVoices = nVoices;
    a.T = nVectors;
    a.channels = p->channels;
    a.step = step;
    a.first = first;
    a.framesPerVector = framesPerVector;
    size_t blocks = (nVectors * nVoices * (size_t)framesPerVector + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(publish_gather_kernel, dim3((unsigned)blocks), dim3(256), 0, e->stream, a);
    hipError_t err = hipGetLastError();
    if (err == hipSuccess) err = hipMemcpyAsync(p->h_stage, p->d_stage, floats * sizeof(float), hipMemcpyDeviceToHost, e->stream);
    if (err == hipSuccess) err = hipStreamSynchronize(e->stream);
    if (err != hipSuccess)
    {
      e->lastError = std::string("published_signal_write: ") + hipGetErrorString(err);
      return MLGPU_ERR_HIP;
    }
    const size_t perVoice = (size_t)framesPerVector * (size_t)p->channels;
    for (size_t i = 0; i < nVectors * nVoices; ++i) mlgpu_dspbuffer_write(p->ring, p->h_stage + i * perVoice, perVoice);
    return MLGPU_OK;
  }

  size_t mlgpu_published_signal_num_channels(mlgpu_published_signal* p) { return p ? (size_t)p->channels : 0; }
  size_t mlgpu_published_signal_read_available(mlgpu_published_signal* p) { return p ? mlgpu_dspbuffer_read_available(p->ring) : 0; }
  size_t mlgpu_published_signal_available_frames(mlgpu_published_signal* p) { return p ? mlgpu_dspbuffer_read_available(p->ring) / (size_t)p->channels : 0; }
  size_t mlgpu_published_signal_read(mlgpu_published_signal* p, float* dest, size_t framesRequested)  // MLSignalProcessor.cpp:35-38
  {
    if (!p || !dest) return 0;
    return mlgpu_dspbuffer_read(p->ring, dest, framesRequested * (size_t)p->channels);
  }
  size_t mlgpu_published_signal_read_latest(mlgpu_published_signal* p, float* dest, size_t framesRequested)  // :19-28
  {
    if (!p || !dest) return 0;
    const size_t avail = mlgpu_dspbuffer_read_available(p->ring), want = framesRequested * (size_t)p->channels;
    if (avail > want) mlgpu_dspbuffer_discard(p->ring, avail - want);
    return mlgpu_dspbuffer_read(p->ring, dest, want);
  }
  void mlgpu_published_signal_peek_latest(mlgpu_published_signal* p, float* dest, size_t framesRequested)  // :30-33
  {
    if (p && dest) mlgpu_dspbuffer_peek_most_recent(p->ring, dest, framesRequested * (size_t)p->channels);
  }
}
