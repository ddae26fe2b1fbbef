// dspbuffer.cpp — host-side single-producer / single-consumer float ring with the semantics of the reference's
// DSPBuffer (source/DSP/MLDSPBuffer.h:20-384): power-of-two size >= max(n, 64) (:104-133); read and write indices
// kept modulo 2*size so "full" (write - read == size) and "empty" (== 0) are distinguishable (:124-130, after
// PortAudio's ring buffer); a write into a full buffer clobbers the oldest data and moves the read index (:162-167);
// acquire / release atomics so one reader thread and one writer thread need no lock. This is host code by nature
// (it buffers audio between the outside world's block sizes and 64-frame DSPVectors); the engine's
// mlgpu_process_buffer (processbuffer.cpp) is built on it.
#include <atomic>
#include <new>
#include <vector>

#include <string.h>

#include "mlgpu_internal.hpp"

struct mlgpu_dspbuffer
{
  std::vector<float> data;
  size_t size{0}, dataMask{0}, distanceMask{0};
  std::atomic<size_t> writeIndex{0}, readIndex{0};

  struct Regions
  {
    float* p1;
    size_t n1;
    float* p2;
    size_t n2;
  };
  Regions regions(size_t index, size_t elems)
  {
    const size_t start = index & dataMask;
    if (start + elems > size) return Regions{data.data() + start, size - start, data.data(), elems - (size - start)};
    return Regions{data.data() + start, elems, nullptr, 0};
  }
  size_t advance(size_t i, size_t n) const { return (i + n) & distanceMask; }
  size_t rewind(size_t i, size_t n) const { return (i - n) & distanceMask; }
  size_t readAvailable() const
  {
    // both acquire: the READER calls this to learn how much the writer has published (it must then see the floats the
    // writer copied before its release store of writeIndex), the WRITER to learn how much the reader has consumed. The
    // reference loads the write index relaxed here (MLDSPBuffer.h:136-141), which ThreadSanitizer rightly reports as a
    // race on the ring's floats (tests/cpp/sanitize_host_test.cpp); on x86 both compile to the same plain load.
    const size_t a = readIndex.load(std::memory_order_acquire);
    const size_t b = writeIndex.load(std::memory_order_acquire);
    return (b - a) & distanceMask;
  }
};

static int bitsToContain(int n)  // MLDSPScalarMath.h: smallest exp with (1 << exp) >= n
{
  int exp = 0;
  while ((1 << exp) < n) exp++;
  return exp;
}

extern "C"
{
  mlgpu_dspbuffer* mlgpu_dspbuffer_create(void) { return new (std::nothrow) mlgpu_dspbuffer(); }
  void mlgpu_dspbuffer_destroy(mlgpu_dspbuffer* b) { delete b; }

  size_t mlgpu_dspbuffer_resize(mlgpu_dspbuffer* b, int sizeInSamples)  // :104-133
  {
    if (!b) return 0;
    b->readIndex = 0;
    b->writeIndex = 0;
    if (sizeInSamples < 0 || sizeInSamples > (1 << 30)) return 0;  // 1 << 31 does not fit the int the size is computed in
    const int bits = bitsToContain(sizeInSamples);
    size_t sz = (size_t)1 << bits;
    if (sz < MLGPU_FLOATS_PER_DSPVECTOR) sz = MLGPU_FLOATS_PER_DSPVECTOR;
    try
    {
      b->data.resize(sz);
    }
    catch (const std::bad_alloc&)
    {
      b->size = b->dataMask = b->distanceMask = 0;
      return 0;
    }
    b->size = sz;
    b->dataMask = sz - 1;
    b->distanceMask = sz * 2 - 1;
    return sz;
  }

  void mlgpu_dspbuffer_clear(mlgpu_dspbuffer* b)  // :96-100
  {
    b->readIndex.store(b->writeIndex.load(std::memory_order_acquire), std::memory_order_release);
  }
  size_t mlgpu_dspbuffer_size(mlgpu_dspbuffer* b) { return b ? b->size : 0; }
  size_t mlgpu_dspbuffer_read_available(mlgpu_dspbuffer* b) { return b->readAvailable(); }
  size_t mlgpu_dspbuffer_write_available(mlgpu_dspbuffer* b) { return b->size - b->readAvailable(); }

  void mlgpu_dspbuffer_write(mlgpu_dspbuffer* b, const float* src, size_t samples)  // :147-168
  {
    if (samples > b->size)  // the reference overruns its storage here; keep the newest `size` samples instead
    {
      src += samples - b->size;
      samples = b->size;
    }
    const bool full = (b->size - b->readAvailable() < samples);
    const size_t w = b->writeIndex.load(std::memory_order_acquire);
    const auto r = b->regions(w, samples);
    memcpy(r.p1, src, r.n1 * sizeof(float));
    if (r.p2) memcpy(r.p2, src + r.n1, r.n2 * sizeof(float));
    b->writeIndex.store(b->advance(w, samples), std::memory_order_release);
    if (full) b->readIndex.store(b->rewind(b->writeIndex, b->size), std::memory_order_release);  // oldest data was clobbered
  }

  size_t mlgpu_dspbuffer_read(mlgpu_dspbuffer* b, float* dst, size_t samples)  // :207-224
  {
    const size_t avail = b->readAvailable();
    if (samples > avail) samples = avail;
    const size_t ri = b->readIndex.load(std::memory_order_acquire);
    const auto r = b->regions(ri, samples);
    memcpy(dst, r.p1, r.n1 * sizeof(float));
    if (r.p2) memcpy(dst + r.n1, r.p2, r.n2 * sizeof(float));
    b->readIndex.store(b->advance(ri, samples), std::memory_order_release);
    return samples;
  }

  // DSPVector read() (:253-277): one whole vector, or zeros (and no index change) when fewer than 64 samples wait
  int mlgpu_dspbuffer_read_vector(mlgpu_dspbuffer* b, float* dst64)
  {
    if (b->readAvailable() < MLGPU_FLOATS_PER_DSPVECTOR)
    {
      memset(dst64, 0, sizeof(float) * MLGPU_FLOATS_PER_DSPVECTOR);
      return 0;
    }
    mlgpu_dspbuffer_read(b, dst64, MLGPU_FLOATS_PER_DSPVECTOR);
    return 1;
  }

  void mlgpu_dspbuffer_discard(mlgpu_dspbuffer* b, size_t samples)  // :280-286
  {
    const size_t avail = b->readAvailable();
    if (samples > avail) samples = avail;
    const size_t ri = b->readIndex.load(std::memory_order_acquire);
    b->readIndex.store(b->advance(ri, samples), std::memory_order_release);
  }

  void mlgpu_dspbuffer_write_with_overlap_add(mlgpu_dspbuffer* b, const float* src, size_t samples, size_t overlap)  // :289-320
  {
    const size_t available = b->size - b->readAvailable();
    const size_t required = samples * 2 - overlap;
    if (available < required) return;  // don't write partial windows
    size_t w = b->writeIndex.load(std::memory_order_acquire);
    auto r = b->regions(w, samples);
    for (size_t i = 0; i < r.n1; ++i) r.p1[i] += src[i];
    if (r.p2)
      for (size_t i = 0; i < r.n2; ++i) r.p2[i] += src[r.n1 + i];
    w = b->advance(w, samples);
    r = b->regions(w, samples - overlap);  // clear samples for the next overlapped add
    memset(r.p1, 0, r.n1 * sizeof(float));
    if (r.p2) memset(r.p2, 0, r.n2 * sizeof(float));
    b->writeIndex.store(b->rewind(w, overlap), std::memory_order_release);
  }

  void mlgpu_dspbuffer_read_with_overlap(mlgpu_dspbuffer* b, float* dst, size_t samples, size_t overlap)  // :323-338
  {
    const size_t available = b->readAvailable() + overlap;
    if (samples > available) samples = available;
    const size_t ri = b->readIndex.load(std::memory_order_acquire);
    const auto r = b->regions(ri, samples);
    memcpy(dst, r.p1, r.n1 * sizeof(float));
    if (r.p2) memcpy(dst + r.n1, r.p2, r.n2 * sizeof(float));
    b->readIndex.store(b->advance(ri, samples - overlap), std::memory_order_release);
  }

  void mlgpu_dspbuffer_peek_most_recent(mlgpu_dspbuffer* b, float* dst, size_t samples)  // :342-383
  {
    const size_t avail = b->readAvailable();
    if (avail < samples) return;
    const size_t ri = b->readIndex.load(std::memory_order_acquire);
    const auto r = b->regions(ri, avail);
    if (!r.p2)
    {
      memcpy(dst, r.p1 + r.n1 - samples, samples * sizeof(float));
    }
    else if (r.n2 >= samples)
    {
      memcpy(dst, r.p2 + r.n2 - samples, samples * sizeof(float));
    }
    else
    {
      const size_t n1 = samples - r.n2;
      memcpy(dst, r.p1 + r.n1 - n1, n1 * sizeof(float));
      memcpy(dst + n1, r.p2, r.n2 * sizeof(float));
    }
  }
}
