// resample.hip — HalfBandFilter / Downsampler / Upsampler (MLDSPFilters.h:1245-1473) for V voices.
//
// The reference runs these as cascades of half-band polyphase allpass filters, one HalfBandFilter per octave, moving
// whole DSPVectors between little buffers (Downsampler::write :1349-1387, Upsampler::write :1428-1452). Every stage is a
// causal per-sample recurrence over its own input stream, and the block schedule feeds each stage its samples in stream
// order, so the same cascade is evaluated here sample by sample: one wavefront lane per voice, the 9 floats of every
// stage (4 x Allpass1 {x1, y1} + b1) in registers for the whole launch, 16-byte coalesced accesses in the QUAD layout.
//   downsample (:1274-1296):  a0 = apa1(apa0(x[2i]));  b0 = apb1(apb0(x[2i+1]));  y[i] = (a0 + b1) * 0.5;  b1 = b0
//   upsample   (:1249-1272):  y[2i] = apa1(apa0(x[i]));  y[2i+1] = apb1(apb0(x[i]))
// HBM-bound by construction: 4 B in + 4 B out per sample at the higher rate, ~14 VALU ops per sample per stage.
#include "mlgpu_internal.hpp"
#include "mldsp_math.hpp"

using namespace mldev;

namespace
{
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kMaxOctaves = 6;

struct Ap1  // Allpass1::processSample, MLDSPFilters.h:945-953
{
  float x1, y1;
  MLD float step(float x, float coeff)
  {
    const float y = x1 + (x - y1) * coeff;
    x1 = x;
    y1 = y;
    return y;
  }
};

struct HalfBand  // :1245-1310; order 4, rejection 70 dB, transition band 0.1 (:1306-1308)
{
  Ap1 a0, a1, b0, b1ap;
  float b1;
  MLD void load(const float* st, size_t V)
  {
    a0.x1 = st[0 * V]; a0.y1 = st[1 * V]; a1.x1 = st[2 * V]; a1.y1 = st[3 * V];
    b0.x1 = st[4 * V]; b0.y1 = st[5 * V]; b1ap.x1 = st[6 * V]; b1ap.y1 = st[7 * V];
    b1 = st[8 * V];
  }
  MLD void store(float* st, size_t V) const
  {
    st[0 * V] = a0.x1; st[1 * V] = a0.y1; st[2 * V] = a1.x1; st[3 * V] = a1.y1;
    st[4 * V] = b0.x1; st[5 * V] = b0.y1; st[6 * V] = b1ap.x1; st[7 * V] = b1ap.y1;
    st[8 * V] = b1;
  }
  MLD float pathA(float x) { return a1.step(a0.step(x, 0.07986642623635751f), 0.5453536510711322f); }
  MLD float pathB(float x) { return b1ap.step(b0.step(x, 0.28382934487410993f), 0.8344118914807379f); }
  MLD float down(float xe, float xo)
  {
    const float va = pathA(xe);
    const float vb = pathB(xo);
    const float y = (va + b1) * 0.5f;
    b1 = vb;
    return y;
  }
};

// one output sample of an H-stage downsampler from 2^H consecutive input samples x[0 .. 2^H)
template <int H>
struct DownCascade
{
  static MLD float run(HalfBand* f, const float* x)
  {
    const float e = DownCascade<H - 1>::run(f, x);
    const float o = DownCascade<H - 1>::run(f, x + (1 << (H - 1)));
    return f[H - 1].down(e, o);
  }
};
template <>
struct DownCascade<0>
{
  static MLD float run(HalfBand*, const float* x) { return x[0]; }
};

// 2^H output samples of an H-stage upsampler from one input sample; stage 0 sees the original signal (:1436-1450)
template <int H, int TOTAL>
struct UpCascade
{
  static MLD void run(HalfBand* f, float x, float* y)
  {
    constexpr int stage = TOTAL - H;
    const float ya = f[stage].pathA(x);
    const float yb = f[stage].pathB(x);
    UpCascade<H - 1, TOTAL>::run(f, ya, y);
    UpCascade<H - 1, TOTAL>::run(f, yb, y + (1 << (H - 1)));
  }
};
template <int TOTAL>
struct UpCascade<0, TOTAL>
{
  static MLD void run(HalfBand*, float x, float* y) { y[0] = x; }
};

struct ResampleArgs
{
  SignalView in, out;
  float* state;  // [octaves * 9][V]
  size_t V, quadsOut, quadsIn;
  uint32_t flags;  // MLGPU_KFLAG_*
};

MLD f32x4* quadPtr(const SignalView& s, size_t v, size_t qi) { return (f32x4*)s.base + (qi >> 4) * s.strideT + (qi & 15) * s.strideQ + v * s.strideV; }

// XCD-aware workgroup -> voice mapping, as the voice-bank kernels (mldsp_kernels.hpp): XCD x works on the x-th contiguous
// eighth of the voices, so every XCD's L2 streams one contiguous segment of each signal row
MLD size_t xcdVoice()
{
  size_t blk = blockIdx.x;
  const size_t nbFull = (size_t)gridDim.x & ~(size_t)7;
  if (blk < nbFull) blk = (blk & 7) * (nbFull >> 3) + (blk >> 3);
  return blk * 256 + threadIdx.x;
}

template <int H>
__global__ __launch_bounds__(256) void downsample_kernel(const ResampleArgs a)
{
  apply_fp_mode(a.flags);
  const size_t v = xcdVoice();
  if (v >= a.V) return;
  HalfBand f[H > 0 ? H : 1];
#pragma unroll
  for (int h = 0; h < H; ++h) f[h].load(a.state + (size_t)h * 9 * a.V + v, a.V);
  constexpr int R = 1 << H;  // input quads per output quad
  // the R input quads of output quad qo + 1 are in flight while quad qo is filtered
  f32x4 nx[R];
  if (a.quadsOut)
  {
#pragma unroll
    for (int r = 0; r < R; ++r) nx[r] = __builtin_nontemporal_load(quadPtr(a.in, v, (size_t)r));
  }
  for (size_t qo = 0; qo < a.quadsOut; ++qo)
  {
    float x[4 * R];
#pragma unroll
    for (int r = 0; r < R; ++r)
    {
      x[4 * r] = nx[r].x; x[4 * r + 1] = nx[r].y; x[4 * r + 2] = nx[r].z; x[4 * r + 3] = nx[r].w;
    }
    const size_t qn = (qo + 1 < a.quadsOut) ? qo + 1 : qo;  // after the last one: fetch it again (never used)
#pragma unroll
    for (int r = 0; r < R; ++r) nx[r] = __builtin_nontemporal_load(quadPtr(a.in, v, qn * R + r));
    f32x4 y;
#pragma unroll
    for (int k = 0; k < 4; ++k) y[k] = DownCascade<H>::run(f, x + k * R);
    __builtin_nontemporal_store(y, quadPtr(a.out, v, qo));
  }
#pragma unroll
  for (int h = 0; h < H; ++h) f[h].store(a.state + (size_t)h * 9 * a.V + v, a.V);
}

template <int H>
__global__ __launch_bounds__(256) void upsample_kernel(const ResampleArgs a)
{
  apply_fp_mode(a.flags);
  const size_t v = xcdVoice();
  if (v >= a.V) return;
  HalfBand f[H > 0 ? H : 1];
#pragma unroll
  for (int h = 0; h < H; ++h) f[h].load(a.state + (size_t)h * 9 * a.V + v, a.V);
  constexpr int R = 1 << H;  // output quads per input quad
  for (size_t qi = 0; qi < a.quadsIn; ++qi)
  {
    const f32x4 q = __builtin_nontemporal_load(quadPtr(a.in, v, qi));
    float y[4 * R];
#pragma unroll
    for (int k = 0; k < 4; ++k) UpCascade<H, H>::run(f, q[k], y + k * R);
#pragma unroll
    for (int r = 0; r < R; ++r)
    {
      f32x4 o = {y[4 * r], y[4 * r + 1], y[4 * r + 2], y[4 * r + 3]};
      __builtin_nontemporal_store(o, quadPtr(a.out, v, qi * R + r));
    }
  }
#pragma unroll
  for (int h = 0; h < H; ++h) f[h].store(a.state + (size_t)h * 9 * a.V + v, a.V);
}

template <int H>
hipError_t launchResample(bool up, const ResampleArgs& a, hipStream_t stream)
{
  const unsigned blocks = (unsigned)((a.V + 255) / 256);
  if (up) hipLaunchKernelGGL(upsample_kernel<H>, dim3(blocks), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL(downsample_kernel<H>, dim3(blocks), dim3(256), 0, stream, a);
  return hipGetLastError();
}
}  // namespace

struct mlgpu_resampler
{
  mlgpu_engine* e{nullptr};
  size_t V{0};
  int octaves{0};
  bool up{false};
  float* d_state{nullptr};
};

extern "C"
{
  int mlgpu_resampler_destroy(mlgpu_resampler* r)
  {
    if (!r) return MLGPU_ERR_INVALID;
    hipSetDevice(r->e->device);
    hipStreamSynchronize(r->e->stream);
    if (r->d_state) hipFree(r->d_state);
    delete r;
    return MLGPU_OK;
  }

  int mlgpu_resampler_clear(mlgpu_resampler* r)  // Downsampler::clear :1391-1399 / Upsampler::clear :1461-1469
  {
    if (!r) return MLGPU_ERR_INVALID;
    if (hipSetDevice(r->e->device) != hipSuccess) return MLGPU_ERR_HIP;
    const hipError_t err = hipMemsetAsync(r->d_state, 0, sizeof(float) * r->V * (size_t)(r->octaves * 9 + 1), r->e->stream);
    if (err != hipSuccess)
    {
      r->e->lastError = std::string("resampler_clear: ") + hipGetErrorString(err);
      return MLGPU_ERR_HIP;
    }
    return MLGPU_OK;
  }

  int mlgpu_resampler_create(mlgpu_engine* e, size_t nVoices, int octaves, int up, mlgpu_resampler** out)
  {
    if (!e || !out) return MLGPU_ERR_INVALID;
    *out = nullptr;
    if (nVoices == 0 || octaves < 0 || octaves > kMaxOctaves)
    {
      e->lastError = "resampler_create: 1+ voices, 0..6 octaves";
      return MLGPU_ERR_INVALID;
    }
    mlgpu_resampler* r = new (std::nothrow) mlgpu_resampler();
    if (!r) return MLGPU_ERR_OOM;
    r->e = e;
    r->V = nVoices;
    r->octaves = octaves;
    r->up = up != 0;
    hipError_t err = hipSetDevice(e->device);
    if (err == hipSuccess) err = hipMalloc((void**)&r->d_state, sizeof(float) * nVoices * (size_t)(octaves * 9 + 1));
    if (err != hipSuccess)
    {
      e->lastError = std::string("resampler_create: ") + hipGetErrorString(err);
      delete r;
      return err == hipErrorOutOfMemory ? MLGPU_ERR_OOM : MLGPU_ERR_HIP;
    }
    const int st = mlgpu_resampler_clear(r);
    if (st != MLGPU_OK)
    {
      mlgpu_resampler_destroy(r);
      return st;
    }
    *out = r;
    return MLGPU_OK;
  }

  int mlgpu_resampler_get_state(mlgpu_resampler* r, float* h) { return r ? mlgpu_download(r->e, h, r->d_state, sizeof(float) * r->V * (size_t)r->octaves * 9) : MLGPU_ERR_INVALID; }
  int mlgpu_resampler_set_state(mlgpu_resampler* r, const float* h) { return r ? mlgpu_upload(r->e, r->d_state, h, sizeof(float) * r->V * (size_t)r->octaves * 9) : MLGPU_ERR_INVALID; }

  int mlgpu_resampler_process(mlgpu_resampler* r, size_t nVectorsIn, const float* d_in, int inLayout, float* d_out, int outLayout)
  {
    if (!r) return MLGPU_ERR_INVALID;
    mlgpu_engine* e = r->e;
    if (nVectorsIn == 0) return MLGPU_OK;
    if (!d_in || !d_out || ((uintptr_t)d_in & 15) || ((uintptr_t)d_out & 15) || inLayout < 0 || inLayout > MLGPU_LAYOUT_VOICE_MAJOR || outLayout < 0 ||
        outLayout > MLGPU_LAYOUT_VOICE_MAJOR)
    {
      e->lastError = "resampler_process: null / misaligned signal or bad layout";
      return MLGPU_ERR_INVALID;
    }
    const size_t ratio = (size_t)1 << r->octaves;
    if (!r->up && (nVectorsIn % ratio))
    {
      e->lastError = "resampler_process: a downsampler takes a multiple of 2^octaves DSPVectors (Downsampler::write yields one vector per 2^octaves writes)";
      return MLGPU_ERR_INVALID;
    }
    const size_t nOut = r->up ? nVectorsIn * ratio : nVectorsIn / ratio;
    ResampleArgs a;
    a.in = makeView(d_in, inLayout, r->V, nVectorsIn);
    a.out = makeView(d_out, outLayout, r->V, nOut);
    a.state = r->d_state;
    a.V = r->V;
    a.flags = e->kflags;
    a.quadsIn = nVectorsIn * 16;
    a.quadsOut = nOut * 16;
    if (hipSetDevice(e->device) != hipSuccess) return MLGPU_ERR_HIP;
    hipError_t err = hipSuccess;
    switch (r->octaves)
    {
      case 0: err = launchResample<0>(r->up, a, e->stream); break;
      case 1: err = launchResample<1>(r->up, a, e->stream); break;
      case 2: err = launchResample<2>(r->up, a, e->stream); break;
      case 3: err = launchResample<3>(r->up, a, e->stream); break;
      case 4: err = launchResample<4>(r->up, a, e->stream); break;
      case 5: err = launchResample<5>(r->up, a, e->stream); break;
      default: err = launchResample<6>(r->up, a, e->stream); break;
    }
    if (err != hipSuccess)
    {
      e->lastError = std::string("resampler_process launch: ") + hipGetErrorString(err);
      return MLGPU_ERR_HIP;
    }
    return MLGPU_OK;
  }
}
