// mldsp_gpu.hpp — C++17 host mirror of the mldsp.h operator interface over the mlgpu C-ABI.
//
// The reference's interface for this path is header-only C++ (include/mldsp.h): value-type
// functors with `static makeCoeffs(...)`, a public `coeffs` member, `clear()` and
// `operator()(DSPVector)` (source/DSP/MLDSPFilters.h:199-240, MLDSPGens.h:395-402), composed per
// voice and replicated with `Bank<T,ROWS>` (MLDSPFunctional.h:321-360). This header keeps those
// names and argument meanings, but a bank here is runtime-sized and lives on an MI355X:
//
//   reference (one voice, CPU)                    this header (V voices, GPU)
//   ------------------------------------------    -------------------------------------------------
//   SawGen saw; Bandpass bp;                      VoiceBank<SawGen, Bandpass, Gain> bank(engine, V);
//   saw.clear();                                  bank.clear();
//   bp.coeffs = Bandpass::makeCoeffs(om, k);      bank.coeffs<1>(v, Bandpass::makeCoeffs(om, k));
//   y = bp(saw(DSPVector(f))) * gain;             bank.input(v, f); bank.coeffs<2>(v, {gain});
//                                                 bank.commit(); bank(T, out);   // T DSPVectors, all voices
//
// Everything numeric happens in libmlgpu.so (HIP kernels); this file only forwards. Errors of the
// C-ABI become ml::gpu::Error exceptions (the reference has no error channel).
#pragma once

#include <algorithm>
#include <array>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <tuple>
#include <utility>
#include <vector>

#include "../mlgpu.h"

namespace ml
{
namespace gpu
{
constexpr size_t kFloatsPerDSPVector = MLGPU_FLOATS_PER_DSPVECTOR;

struct Error : std::runtime_error
{
  int status;
  Error(int st, const std::string& what) : std::runtime_error(what), status(st) {}
};

class Engine
{
  mlgpu_engine* e_{nullptr};
  bool owned_{true};

 public:
  // an engine someone else made and will destroy (a host that holds the C handle already: bench.py's reverb workload)
  struct Borrowed
  {
  };
  Engine(mlgpu_engine* existing, Borrowed) : e_(existing), owned_(false)
  {
    if (!existing) throw Error(MLGPU_ERR_INVALID, "Engine: null handle");
  }
  explicit Engine(int device = 0)
  {
    const int st = mlgpu_engine_create(device, &e_);
    if (st != MLGPU_OK) throw Error(st, std::string("mlgpu_engine_create: ") + mlgpu_status_string(st));
  }
  // a second engine of a device for short launches that should slip in beside another engine's long ones (mlgpu_engine_create_urgency:
  // +1 the device's greatest stream priority, -1 its least); order the two with Fence
  Engine(int device, int urgency)
  {
    const int st = mlgpu_engine_create_urgency(device, urgency, &e_);
    if (st != MLGPU_OK) throw Error(st, std::string("mlgpu_engine_create_urgency: ") + mlgpu_status_string(st));
  }
  int device() const { return mlgpu_engine_device(e_); }
  Engine(const Engine&) = delete;
  Engine& operator=(const Engine&) = delete;
  ~Engine()
  {
    if (e_ && owned_) mlgpu_engine_destroy(e_);
  }
  mlgpu_engine* handle() const { return e_; }
  void check(int st) const
  {
    if (st != MLGPU_OK) throw Error(st, std::string(mlgpu_status_string(st)) + ": " + mlgpu_last_error(e_));
  }
  void sync() const { check(mlgpu_engine_sync(e_)); }
};

// A point in one engine's stream that another engine of the same device can wait for (mlgpu_fence).
class Fence
{
  mlgpu_fence* f_{nullptr};

 public:
  explicit Fence(const Engine& e) { e.check(mlgpu_fence_create(e.handle(), &f_)); }
  Fence(const Fence&) = delete;
  Fence& operator=(const Fence&) = delete;
  ~Fence()
  {
    if (f_) mlgpu_fence_destroy(f_);
  }
  void signalFrom(const Engine& e) { e.check(mlgpu_engine_signal(e.handle(), f_)); }  // everything e has enqueued so far
  void awaitedBy(const Engine& e) { e.check(mlgpu_engine_wait(e.handle(), f_)); }     // e's later launches start after that (no-op before any signal)
};

// The GPUs of one node as one voice bank host: voices share nothing (every Bank row / Synth voice owns its
// state, MLDSPFunctional.h:321-349, source/app/MLSynth.h:49-57), so device g simply owns the contiguous voice range
// partition(total, size(), g) with its own engine, stream and host thread. There is no collective on the path; time is
// never split across devices (phases and filter memories are recurrences). The C-ABI is per-engine and thread-compatible
// (one caller thread per engine, the reference's single audio thread), which is exactly what the worker threads are.
class DeviceGroup
{
  struct Worker
  {
    std::unique_ptr<Engine> engine;
    std::thread thread;
    std::function<void()> job;
    std::exception_ptr error;
    bool busy{false}, quit{false};
    std::mutex m;
    std::condition_variable cv;
  };
  std::vector<std::unique_ptr<Worker>> workers_;

  static void loop(Worker* w)
  {
    std::unique_lock<std::mutex> lock(w->m);
    for (;;)
    {
      w->cv.wait(lock, [w] { return w->busy || w->quit; });
      if (w->quit) return;
      lock.unlock();
      try
      {
        w->job();
      }
      catch (...)
      {
        w->error = std::current_exception();
      }
      lock.lock();
      w->busy = false;
      w->cv.notify_all();
    }
  }

 public:
  // engines on devices 0 .. nDevices-1; fewer visible devices than asked for is an error, never a smaller group
  explicit DeviceGroup(int nDevices)
  {
    const int have = mlgpu_device_count();
    if (nDevices < 1 || nDevices > have)
      throw Error(MLGPU_ERR_NO_DEVICE, "DeviceGroup: " + std::to_string(nDevices) + " device(s) asked for, " + std::to_string(have) + " visible");
    for (int d = 0; d < nDevices; ++d)
    {
      workers_.emplace_back(new Worker());
      workers_.back()->engine.reset(new Engine(d));
    }
    for (auto& w : workers_) w->thread = std::thread(loop, w.get());
  }
  // one engine (and host thread) per entry of an explicit device list; a device may appear more than once - several engines of one
  // GPU (what tests/cpp/multi_engine_test.cpp uses to run an 8-member group on a one-GPU box; two engines per GPU are two streams)
  explicit DeviceGroup(const std::vector<int>& devices)
  {
    const int have = mlgpu_device_count();
    if (devices.empty()) throw Error(MLGPU_ERR_INVALID, "DeviceGroup: empty device list");
    for (int d : devices)
      if (d < 0 || d >= have) throw Error(MLGPU_ERR_NO_DEVICE, "DeviceGroup: device " + std::to_string(d) + " asked for, " + std::to_string(have) + " visible");
    for (int d : devices)
    {
      workers_.emplace_back(new Worker());
      workers_.back()->engine.reset(new Engine(d));
    }
    for (auto& w : workers_) w->thread = std::thread(loop, w.get());
  }
  DeviceGroup(const DeviceGroup&) = delete;
  DeviceGroup& operator=(const DeviceGroup&) = delete;
  ~DeviceGroup()
  {
    for (auto& w : workers_)
    {
      {
        std::lock_guard<std::mutex> lock(w->m);
        w->quit = true;
      }
      w->cv.notify_all();
      if (w->thread.joinable()) w->thread.join();
    }
  }
  int size() const { return (int)workers_.size(); }
  Engine& engine(int g) { return *workers_.at((size_t)g)->engine; }

  // contiguous voice range [lo, hi) of device `rank`; sizes differ by at most one voice
  static std::pair<size_t, size_t> partition(size_t totalVoices, int world, int rank)
  {
    const size_t per = totalVoices / (size_t)world, rem = totalVoices % (size_t)world;
    const size_t lo = (size_t)rank * per + std::min<size_t>((size_t)rank, rem);
    return {lo, lo + per + ((size_t)rank < rem ? 1 : 0)};
  }

  // fn(rank, engine, lo, hi) on every device's own host thread, all devices at once; returns when all are back
  // (their launches may still be running: call sync()). The first exception thrown by any fn is rethrown here.
  template <class F>
  void forEach(size_t totalVoices, F fn)
  {
    const int n = size();
    for (int g = 0; g < n; ++g)
    {
      Worker* w = workers_[(size_t)g].get();
      const auto span = partition(totalVoices, n, g);
      std::lock_guard<std::mutex> lock(w->m);
      w->error = nullptr;
      w->job = [fn, g, w, span]() mutable { fn(g, *w->engine, span.first, span.second); };
      w->busy = true;
      w->cv.notify_all();
    }
    std::exception_ptr first;
    for (auto& w : workers_)
    {
      std::unique_lock<std::mutex> lock(w->m);
      w->cv.wait(lock, [&w] { return !w->busy; });
      if (w->error && !first) first = w->error;
    }
    if (first) std::rethrow_exception(first);
  }
  void sync()
  {
    for (auto& w : workers_) w->engine->sync();
  }
};

// The real-time block ACROSS the devices of a group: all voices of a sharded bank summed to one channel - the reference's
// `outputs += voice` over every voice of a Synth (source/app/MLSynth.h:43-57), SURVEY 8e's "per-GPU reduction then host add".
// Every member turns its contiguous voice range into the rows of the mixdown tree at the level its voice count is whole at
// (mlgpu_bank_process_mixdown_shard, or a graph output in shard form: the sum is made inside the voice kernel, the voices' signals are
// never written), copies them (rows x 256 bytes per DSPVector: one row per 262 144 voices) to the host, and the host finishes the
// SAME tree over all members' rows (mlgpu_mixdown_finish): N members x V / N voices give the bits ONE engine gives for V voices.
// No collective, no peer access: the exchange is N x rows x 256 bytes per DSPVector through pinned-size host memory.
// The voices must divide evenly over the members and every member's share must be a multiple of 64.
class GroupMixdown
{
  DeviceGroup& group_;
  size_t total_, per_, rows_, maxVectors_;
  std::vector<float*> dRows_;
  std::vector<float> hRows_, scratch_;
  int flush_{0};

 public:
  GroupMixdown(DeviceGroup& g, size_t totalVoices, size_t maxVectors) : group_(g), total_(totalVoices), maxVectors_(maxVectors)
  {
    const size_t n = (size_t)g.size();
    if (totalVoices == 0 || totalVoices % n) throw Error(MLGPU_ERR_INVALID, "GroupMixdown: the voices do not divide evenly over the group's members");
    per_ = totalVoices / n;
    rows_ = mlgpu_mixdown_shard_rows(per_);
    if (rows_ == 0) throw Error(MLGPU_ERR_INVALID, "GroupMixdown: a member's share of the voices must be a multiple of 64");
    dRows_.assign(n, nullptr);
    hRows_.assign(n * rows_ * 64 * maxVectors, 0.f);
    const size_t allRows = n * rows_;
    scratch_.assign(((allRows + 63) / 64 + (allRows + 4095) / 4096) * 64 * maxVectors, 0.f);
    for (size_t m = 0; m < n; ++m)
    {
      Engine& e = g.engine((int)m);
      e.check(mlgpu_mixdown_reserve(e.handle(), per_, maxVectors));
      void* p = nullptr;
      e.check(mlgpu_alloc(e.handle(), sizeof(float) * rows_ * 64 * maxVectors, &p));
      dRows_[m] = static_cast<float*>(p);
    }
  }
  GroupMixdown(const GroupMixdown&) = delete;
  GroupMixdown& operator=(const GroupMixdown&) = delete;
  ~GroupMixdown()
  {
    for (size_t m = 0; m < dRows_.size(); ++m)
      if (dRows_[m]) mlgpu_free(group_.engine((int)m).handle(), dRows_[m]);
  }
  size_t voicesPerMember() const { return per_; }
  size_t rowsPerMember() const { return rows_; }
  void setFlushDenormals(bool on) { flush_ = on ? 1 : 0; }  // what the engines run with (mlgpu_engine_set_flush_denormals)
  // One block of nVectors DSPVectors: shard(rank, engine, lo, hi, d_rows) enqueues the member's voices -> rows launch (e.g.
  // mlgpu_bank_process_mixdown_shard(bank[rank], nVectors, nullptr, 0, nullptr, d_rows)); every member runs it on its own host
  // thread, all at once; `out` gets the 64 * nVectors samples of the sum of ALL voices.
  template <class F>
  void process(size_t nVectors, F shard, float* out)
  {
    if (nVectors == 0) return;
    if (nVectors > maxVectors_) throw Error(MLGPU_ERR_RANGE, "GroupMixdown::process: more DSPVectors than reserved");
    const size_t rowFloats = 64 * nVectors;
    group_.forEach(total_, [&](int rank, Engine& e, size_t lo, size_t hi) {
      shard(rank, e, lo, hi, dRows_[(size_t)rank]);
      e.check(mlgpu_download(e.handle(), hRows_.data() + (size_t)rank * rows_ * rowFloats, dRows_[(size_t)rank], sizeof(float) * rows_ * rowFloats));
    });
    const int st = mlgpu_mixdown_finish(hRows_.data(), (size_t)group_.size() * rows_, nVectors, flush_, out, scratch_.data());
    if (st != MLGPU_OK) throw Error(st, "mlgpu_mixdown_finish");
  }
};

// DSPBuffer (source/DSP/MLDSPBuffer.h:20-384): the reference's single-producer single-consumer float ring, on the host, with
// its method names. One reader thread and one writer thread may use it concurrently, as in the reference.
class DSPBuffer
{
  mlgpu_dspbuffer* b_;

 public:
  DSPBuffer() : b_(mlgpu_dspbuffer_create()) {}
  explicit DSPBuffer(int sizeInSamples) : DSPBuffer() { resize(sizeInSamples); }
  ~DSPBuffer() { mlgpu_dspbuffer_destroy(b_); }
  DSPBuffer(const DSPBuffer&) = delete;
  DSPBuffer& operator=(const DSPBuffer&) = delete;
  size_t resize(int sizeInSamples) { return mlgpu_dspbuffer_resize(b_, sizeInSamples); }
  size_t size() const { return mlgpu_dspbuffer_size(b_); }
  void clear() { mlgpu_dspbuffer_clear(b_); }
  size_t getReadAvailable() const { return mlgpu_dspbuffer_read_available(b_); }
  size_t getWriteAvailable() const { return mlgpu_dspbuffer_write_available(b_); }
  void write(const float* src, size_t samples) { mlgpu_dspbuffer_write(b_, src, samples); }
  size_t read(float* dst, size_t samples) { return mlgpu_dspbuffer_read(b_, dst, samples); }
  bool readVector(float* dst64) { return mlgpu_dspbuffer_read_vector(b_, dst64) != 0; }
  void discard(size_t samples) { mlgpu_dspbuffer_discard(b_, samples); }
  void writeWithOverlapAdd(const float* src, size_t samples, int overlap) { mlgpu_dspbuffer_write_with_overlap_add(b_, src, samples, (size_t)overlap); }
  void readWithOverlap(float* dst, size_t samples, int overlap) { mlgpu_dspbuffer_read_with_overlap(b_, dst, samples, (size_t)overlap); }
  void peekMostRecent(float* dst, size_t samples) const { mlgpu_dspbuffer_peek_most_recent(b_, dst, samples); }
  mlgpu_dspbuffer* handle() const { return b_; }
};

// makeWindow(dest, size, dspwindows::<shape>) (source/DSP/MLDSPUtils.h:22-47) by shape number (mlgpu_window)
inline void makeWindow(float* dest, size_t size, int shape)
{
  if (mlgpu_make_window(dest, size, shape) != MLGPU_OK) throw std::invalid_argument("ml::gpu::makeWindow: unknown window shape");
}

// A V-voice, T-vector float signal in HBM.
class DeviceSignal
{
  const Engine* eng_{nullptr};
  float* d_{nullptr};
  size_t voices_{0}, vectors_{0};
  int layout_{MLGPU_LAYOUT_QUAD};

 public:
  DeviceSignal() = default;
  DeviceSignal(const Engine& e, size_t voices, size_t vectors, int layout = MLGPU_LAYOUT_QUAD)
      : eng_(&e), voices_(voices), vectors_(vectors), layout_(layout)
  {
    void* p = nullptr;
    e.check(mlgpu_alloc(e.handle(), bytes(), &p));
    d_ = static_cast<float*>(p);
  }
  DeviceSignal(DeviceSignal&& o) noexcept { *this = std::move(o); }
  DeviceSignal& operator=(DeviceSignal&& o) noexcept
  {
    std::swap(eng_, o.eng_);
    std::swap(d_, o.d_);
    std::swap(voices_, o.voices_);
    std::swap(vectors_, o.vectors_);
    std::swap(layout_, o.layout_);
    return *this;
  }
  DeviceSignal(const DeviceSignal&) = delete;
  DeviceSignal& operator=(const DeviceSignal&) = delete;
  ~DeviceSignal()
  {
    if (d_) mlgpu_free(eng_->handle(), d_);
  }
  float* data() const { return d_; }
  size_t voices() const { return voices_; }
  size_t vectors() const { return vectors_; }
  int layout() const { return layout_; }
  size_t size() const { return voices_ * vectors_ * kFloatsPerDSPVector; }
  size_t bytes() const { return size() * sizeof(float); }

  // host copies in the reference's DSPVectorArray<V> order: [vector][voice][64]
  std::vector<float> toRows() const
  {
    std::vector<float> h(size());
    if (layout_ == MLGPU_LAYOUT_ROWS)
    {
      eng_->check(mlgpu_download(eng_->handle(), h.data(), d_, bytes()));
    }
    else
    {
      DeviceSignal tmp(*eng_, voices_, vectors_, MLGPU_LAYOUT_ROWS);
      eng_->check(mlgpu_layout_convert(eng_->handle(), d_, layout_, tmp.data(), MLGPU_LAYOUT_ROWS, voices_, vectors_));
      eng_->check(mlgpu_download(eng_->handle(), h.data(), tmp.data(), bytes()));
    }
    return h;
  }
  void fromRows(const std::vector<float>& h)
  {
    if (h.size() != size()) throw Error(MLGPU_ERR_INVALID, "DeviceSignal::fromRows: size mismatch");
    if (layout_ == MLGPU_LAYOUT_ROWS)
    {
      eng_->check(mlgpu_upload(eng_->handle(), d_, h.data(), bytes()));
    }
    else
    {
      DeviceSignal tmp(*eng_, voices_, vectors_, MLGPU_LAYOUT_ROWS);
      eng_->check(mlgpu_upload(eng_->handle(), tmp.data(), h.data(), bytes()));
      eng_->check(mlgpu_layout_convert(eng_->handle(), tmp.data(), MLGPU_LAYOUT_ROWS, d_, layout_, voices_, vectors_));
    }
  }
};

// ---- processor tags: same names, same makeCoeffs signatures as the reference -------------------

struct PhasorGen { static constexpr int kind = MLGPU_PROC_PHASOR_GEN; static constexpr int nCoeffs = 0; };
struct SineGen { static constexpr int kind = MLGPU_PROC_SINE_GEN; static constexpr int nCoeffs = 0; };
struct SawGen { static constexpr int kind = MLGPU_PROC_SAW_GEN; static constexpr int nCoeffs = 0; };
struct PulseGen { static constexpr int kind = MLGPU_PROC_PULSE_GEN; static constexpr int nCoeffs = 1; /* width */ };
struct NoiseGen { static constexpr int kind = MLGPU_PROC_NOISE_GEN; static constexpr int nCoeffs = 0; };
struct TickGen { static constexpr int kind = MLGPU_PROC_TICK_GEN; static constexpr int nCoeffs = 0; };
struct ImpulseGen { static constexpr int kind = MLGPU_PROC_IMPULSE_GEN; static constexpr int nCoeffs = 0; };
struct OneShotGen { static constexpr int kind = MLGPU_PROC_ONE_SHOT_GEN; static constexpr int nCoeffs = 0; };

struct Lopass  // MLDSPFilters.h:51-153
{
  static constexpr int kind = MLGPU_PROC_LOPASS;
  static constexpr int nCoeffs = 3;
  using Coeffs = std::array<float, 3>;
  static Coeffs makeCoeffs(float omega, float k)
  {
    Coeffs c;
    mlgpu_lopass_make_coeffs(omega, k, c.data());
    return c;
  }
};
struct Hipass  // :155-197
{
  static constexpr int kind = MLGPU_PROC_HIPASS;
  static constexpr int nCoeffs = 4;
  using Coeffs = std::array<float, 4>;
  static Coeffs makeCoeffs(float omega, float k)
  {
    Coeffs c;
    mlgpu_hipass_make_coeffs(omega, k, c.data());
    return c;
  }
};
struct Bandpass  // :199-240
{
  static constexpr int kind = MLGPU_PROC_BANDPASS;
  static constexpr int nCoeffs = 3;
  using Coeffs = std::array<float, 3>;
  static Coeffs makeCoeffs(float omega, float k)
  {
    Coeffs c;
    mlgpu_bandpass_make_coeffs(omega, k, c.data());
    return c;
  }
};
struct LoShelf  // :242-319
{
  static constexpr int kind = MLGPU_PROC_LO_SHELF;
  static constexpr int nCoeffs = 5;
  using Coeffs = std::array<float, 5>;
  using params = std::array<float, 3>;  // omega, k, A
  static Coeffs makeCoeffs(params p)
  {
    Coeffs c;
    mlgpu_loshelf_make_coeffs(p[0], p[1], p[2], c.data());
    return c;
  }
};
struct HiShelf  // :321-400
{
  static constexpr int kind = MLGPU_PROC_HI_SHELF;
  static constexpr int nCoeffs = 6;
  using Coeffs = std::array<float, 6>;
  using params = std::array<float, 3>;
  static Coeffs makeCoeffs(params p)
  {
    Coeffs c;
    mlgpu_hishelf_make_coeffs(p[0], p[1], p[2], c.data());
    return c;
  }
};
struct Bell  // :402-442
{
  static constexpr int kind = MLGPU_PROC_BELL;
  static constexpr int nCoeffs = 4;
  using Coeffs = std::array<float, 4>;
  static Coeffs makeCoeffs(float omega, float k, float A)
  {
    Coeffs c;
    mlgpu_bell_make_coeffs(omega, k, A, c.data());
    return c;
  }
};
struct OnePole  // :446-481
{
  static constexpr int kind = MLGPU_PROC_ONE_POLE;
  static constexpr int nCoeffs = 2;
  using Coeffs = std::array<float, 2>;
  static Coeffs makeCoeffs(float omega)
  {
    Coeffs c;
    mlgpu_onepole_make_coeffs(omega, c.data());
    return c;
  }
  static Coeffs passthru() { return {1.f, 0.f}; }
};
struct DCBlocker  // :489-513
{
  static constexpr int kind = MLGPU_PROC_DC_BLOCKER;
  static constexpr int nCoeffs = 1;
  using Coeffs = std::array<float, 1>;
  static Coeffs makeCoeffs(float omega) { return {mlgpu_dcblocker_make_coeffs(omega)}; }
};
struct Differentiator { static constexpr int kind = MLGPU_PROC_DIFFERENTIATOR; static constexpr int nCoeffs = 0; };
struct Integrator { static constexpr int kind = MLGPU_PROC_INTEGRATOR; static constexpr int nCoeffs = 1; /* mLeak */ };
struct RMS  // :619-653
{
  static constexpr int kind = MLGPU_PROC_RMS;
  static constexpr int nCoeffs = 2;
  using Coeffs = std::array<float, 2>;
  static Coeffs makeCoeffs(float omega) { return OnePole::makeCoeffs(omega); }
};
struct ADSR  // :657-797
{
  static constexpr int kind = MLGPU_PROC_ADSR;
  static constexpr int nCoeffs = 4;
  using Coeffs = std::array<float, 4>;
  static Coeffs calcCoeffs(float a, float d, float s, float r, float sr)
  {
    Coeffs c;
    mlgpu_adsr_calc_coeffs(a, d, s, r, sr, c.data());
    return c;
  }
};
struct Gain { static constexpr int kind = MLGPU_PROC_GAIN; static constexpr int nCoeffs = 1; };

inline float dBToGain(float dB) { return mlgpu_db_to_gain(dB); }  // MLDSPFilters.h:30

// ---- VoiceBank: runtime-sized Bank<T, ROWS> where T is the chain Procs... -----------------------

template <class... Procs>
class VoiceBank
{
  const Engine& eng_;
  mlgpu_bank* b_{nullptr};
  size_t voices_;
  static constexpr int kNumProcs = sizeof...(Procs);
  static constexpr std::array<int, sizeof...(Procs)> kNC{Procs::nCoeffs...};
  std::array<std::vector<float>, sizeof...(Procs)> hostCoeffs_;  // [proc][coeff * V + v]
  std::vector<float> hostInput_;
  std::array<bool, sizeof...(Procs)> dirty_{};
  bool inputDirty_{false};

 public:
  VoiceBank(const Engine& e, size_t nVoices) : eng_(e), voices_(nVoices)
  {
    const int32_t kinds[] = {Procs::kind...};
    eng_.check(mlgpu_bank_create(e.handle(), kinds, kNumProcs, nVoices, &b_));
    for (int p = 0; p < kNumProcs; ++p) hostCoeffs_[p].assign((size_t)kNC[p] * nVoices, 0.f);
    hostInput_.assign(nVoices, 0.f);
  }
  VoiceBank(const VoiceBank&) = delete;
  VoiceBank& operator=(const VoiceBank&) = delete;
  ~VoiceBank()
  {
    if (b_) mlgpu_bank_destroy(b_);
  }

  size_t voices() const { return voices_; }
  bool fused() const { return mlgpu_bank_is_fused(b_) != 0; }

  // Bank::clear(), MLDSPFunctional.h:351-357
  void clear() { eng_.check(mlgpu_bank_clear(b_)); }

  // `bank[v].proc<I>.coeffs = c` of the reference; staged on the host until commit()
  template <int I, size_t N>
  void coeffs(size_t voice, const std::array<float, N>& c)
  {
    static_assert(I >= 0 && I < kNumProcs, "processor index out of range");
    static_assert((int)N == kNC[I], "wrong number of coefficients for this processor");
    for (size_t i = 0; i < N; ++i) hostCoeffs_[I][i * voices_ + voice] = c[i];
    dirty_[I] = true;
  }
  // same coefficients for every voice
  template <int I, size_t N>
  void coeffsAll(const std::array<float, N>& c)
  {
    static_assert((int)N == kNC[I], "wrong number of coefficients for this processor");
    for (size_t i = 0; i < N; ++i) eng_.check(mlgpu_bank_set_coeff_uniform(b_, I, (int)i, c[i]));
    for (size_t i = 0; i < N; ++i) std::fill_n(hostCoeffs_[I].begin() + i * voices_, voices_, c[i]);
  }
  // the scalar the head processor sees, i.e. `saw(DSPVector(f))`: per-voice constant input
  void input(size_t voice, float f)
  {
    hostInput_[voice] = f;
    inputDirty_ = true;
  }
  // upload staged coefficients / inputs
  void commit()
  {
    for (int p = 0; p < kNumProcs; ++p)
    {
      if (!dirty_[p]) continue;
      for (int i = 0; i < kNC[p]; ++i) eng_.check(mlgpu_bank_set_coeff(b_, p, i, hostCoeffs_[p].data() + (size_t)i * voices_));
      dirty_[p] = false;
    }
    if (inputDirty_) eng_.check(mlgpu_bank_set_input_const(b_, hostInput_.data()));
    inputDirty_ = false;
  }

  // Bank::operator(), MLDSPFunctional.h:328-337, for out.vectors() DSPVectors of every voice.
  // Generators / chains fed by the per-voice constant input:
  void operator()(DeviceSignal& out)
  {
    commit();
    eng_.check(mlgpu_bank_process(b_, out.vectors(), nullptr, MLGPU_LAYOUT_QUAD, out.data(), out.layout()));
  }
  // Chains fed by a streamed signal (filters):
  void operator()(const DeviceSignal& in, DeviceSignal& out)
  {
    if (in.voices() != voices_ || out.voices() != voices_ || in.vectors() != out.vectors())
      throw Error(MLGPU_ERR_INVALID, "VoiceBank: signal shape mismatch");
    commit();
    eng_.check(mlgpu_bank_process(b_, out.vectors(), in.data(), in.layout(), out.data(), out.layout()));
  }

  // ... and when only the SUM of the voices is wanted (a Synth's `outputs += voice`, MLSynth.h:43-57): operator() followed by
  // mlgpu_mixdown of its output, in one launch and without the voices' signals in memory - the same bits (mlgpu_bank_process_mixdown:
  // one fused kernel, mlgpu_mixdown_reserve at setup). `mix` is a single-voice signal of `vectors` DSPVectors.
  void mixdown(size_t vectors, float* mix, const float* gains = nullptr)   // gains: per-voice, on the device, or none
  {
    commit();
    eng_.check(mlgpu_bank_process_mixdown(b_, vectors, nullptr, MLGPU_LAYOUT_QUAD, gains, mix));
  }
  void mixdown(const DeviceSignal& in, float* mix, const float* gains = nullptr)
  {
    if (in.voices() != voices_) throw Error(MLGPU_ERR_INVALID, "VoiceBank: signal shape mismatch");
    commit();
    eng_.check(mlgpu_bank_process_mixdown(b_, in.vectors(), in.data(), in.layout(), gains, mix));
  }
  // ... for a bank that is one member's share of a larger one (GroupMixdown): mlgpu_mixdown_shard_rows(voices) rows of 64 * vectors
  // floats, the mixdown tree up to the level this bank's voice count is whole at
  void mixdownShard(size_t vectors, float* rows, const float* gains = nullptr)
  {
    commit();
    eng_.check(mlgpu_bank_process_mixdown_shard(b_, vectors, nullptr, MLGPU_LAYOUT_QUAD, gains, rows));
  }

  // raw state (checkpoint / resume)
  std::vector<uint32_t> state(int proc, int idx) const
  {
    std::vector<uint32_t> s(voices_);
    eng_.check(mlgpu_bank_get_state(b_, proc, idx, s.data()));
    return s;
  }
  void setState(int proc, int idx, const std::vector<uint32_t>& s) { eng_.check(mlgpu_bank_set_state(b_, proc, idx, s.data())); }
  void setStateAll(int proc, int idx, uint32_t v) { eng_.check(mlgpu_bank_set_state_uniform(b_, proc, idx, v)); }
};

}  // namespace gpu
}  // namespace ml
