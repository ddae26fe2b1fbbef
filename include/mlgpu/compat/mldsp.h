// mldsp.h (MI355X drop-in) — source-compatible shim of madronalib's `#include "mldsp.h"` functional surface.
//
// Put include/mlgpu/compat on the include path INSTEAD of the reference's include/ + source/DSP and existing
// proc chains compile unchanged (reference include/mldsp.h:7-16; DSPVector value semantics MLDSPOps.h:94-361;
// free functions MLDSPOps.h:570-1383; generators MLDSPGens.h; filters MLDSPFilters.h; Bank MLDSPFunctional.h:321).
// The difference is WHEN the arithmetic happens: here a `DSPVector` is a handle to a node of a per-voice signal
// graph. Running the user's process function ONCE records ("captures") the graph; ml::gpu::VoiceProgram then
// compiles it into one fused gfx950 kernel (mlgpu_graph, include/mlgpu.h) and evaluates it for V voices x T
// DSPVectors per launch, one wavefront lane per voice.
//
//   reference (1 voice, CPU, called once per 64 frames)          here (V voices, GPU)
//   ----------------------------------------------------         -------------------------------------------------
//   void proc(AudioContext* ctx, void* st) {                      same source, unchanged
//     auto s = static_cast<State*>(st);
//     ctx->outputs[0] = s->lp(s->saw(220.f/48000.f)) * 0.1f; }
//   AudioTask task(&ctx, proc, &state); task.run...               ml::gpu::VoiceProgram prog(engine, V, &ctx, proc, &state);
//                                                                 prog.process(T, {}, {&out});
//
// What can be captured: everything that is data flow on whole DSPVectors — operators, the DEFINE_OP* free
// functions, compare/select, conversions, row plumbing, index generators, mix/multiplex, every generator and filter
// object with its makeCoeffs / coeffs / clear() / operator() forms, Bank<T,ROWS>, the rate functions. What cannot: anything
// that needs a float of a signal the kernel computes ON THE HOST (`v[n]`, store(), sum()/mean()/max()/min() to float, ==,
// rotateLeft, map() with a scalar function) — inside a capture those throw and say so.
// Per-voice variation comes in through ml::gpu::VoiceParam (a per-voice constant), streamed inputs
// (ctx->inputs[c]) and per-voice coefficients (VoiceProgram::setCoeff).
//
// IMMEDIATE MODE. The same calls made OUTSIDE a capture — a unit test, an offline tool, a setup function — run at once, on the
// device, for one voice: a DSPVector is then host data, as in the reference, every call is one launch (gpu::Eager below), and all of
// the above exists, host access included. The reference's own unit tests (Tests/dspOpsTest.cpp, dspGensTest.cpp, dspFiltersTest.cpp,
// dspBufferTest.cpp) compile unchanged against this header — include/mlgpu/compat/dsp forwards madronalib's header names — and pass
// (tests/test_gpu_immediate.py); a process function written for capture can also be called directly, vector by vector. It is the
// same arithmetic as the fused kernels, a round trip per call, and no CPU fallback: without a device the calls throw.
#pragma once

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <initializer_list>
#include <map>
#include <memory>
#include <mutex>
#include <ostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../mlgpu.h"
#include "../mldsp_gpu.hpp"

// Host-side scalar helpers (ml::Projection, ml::Interval, projections::*, the scalar clamp / lerp / dBToAmp ...) are not
// part of the DSPVector engine: they are plain host code that a process function calls with floats (SURVEY §2 lists
// MLDSPProjections.h / MLDSPScalarMath.h as out of scope). A program that uses them already has madronalib's own headers;
// when they are on the include path AFTER this directory they are used as they are, so the numbers a process function
// computes on the host are madronalib's own (the reference's reverb.cpp example builds its decay knob that way).
// The same goes for the parameter layer (source/app/MLParameters.h: Path, Symbol, Value, ParameterDescription,
// ParameterTree - names, ranges, normalized <-> real mappings): host-side control plane. With madronalib's source/app on the
// include path it is used as it is and ml::SignalProcessor below offers the reference's parameter calls over it (buildParams,
// setDefaultParams, setParamFromNormalizedValue, getParameterTree().getRealFloatValue...: what the reference's params.cpp
// example uses); the floats a process function reads from it become constants of the captured kernel, refreshed by
// gpu::VoiceProgram::update() when the host changes a parameter.
#if defined(__has_include)
#if __has_include("MLParameters.h")
#include "MLParameters.h"  // brings MLDSPProjections.h with it
#define MLGPU_COMPAT_HAS_MADRONALIB_SCALAR_HEADERS 1
#define MLGPU_COMPAT_HAS_MADRONALIB_APP_HEADERS 1
#elif __has_include("MLDSPProjections.h")
#include "MLDSPProjections.h"
#define MLGPU_COMPAT_HAS_MADRONALIB_SCALAR_HEADERS 1
#endif
#endif
// ... and without a madronalib checkout the shim brings its own (round 4): the same names with the same arithmetic, so that
// sine.cpp, reverb.cpp, fdtd.cpp and controllers-to-audio.cpp of the reference's examples build with this directory alone
#ifndef MLGPU_COMPAT_HAS_MADRONALIB_SCALAR_HEADERS
#include "mlscalar.h"
#define MLGPU_COMPAT_OWN_SCALAR_HEADER 1
#endif
#include <iostream>  // (madronalib's DSP headers bring it; user code says std::cout without asking)

namespace ml
{
constexpr size_t kFloatsPerDSPVector = MLGPU_FLOATS_PER_DSPVECTOR;
#if !defined(MLGPU_COMPAT_HAS_MADRONALIB_SCALAR_HEADERS) && !defined(MLGPU_COMPAT_OWN_SCALAR_HEADER)
constexpr float kPi = 3.1415926535897932384626433832795f;
constexpr float kTwoPi = kPi * 2.f;
#endif

namespace gpu
{
// ---- capture context ---------------------------------------------------------------------------------------
struct Capture
{
  const Engine* eng{nullptr};
  mlgpu_graph* g{nullptr};
  std::map<uint32_t, int> constNodes;
  struct Deferred
  {
    int node, what, idx;  // what: 0 coeff (float), 1 clear(), 2 state word
    uint32_t bits;
  };
  std::vector<Deferred> deferred;
  struct Tap  // storePublishedSignal(name, DSPVectorArray<CH>, 64, voice) seen in the captured code
  {
    std::string name;
    std::vector<int> nodes;  // one per channel
  };
  std::vector<Tap> taps;
  // Two capture passes ("epochs"): a DSPVector that still holds a node of the PREVIOUS pass when it is used is a value
  // the user code kept from one process call to the next (Allpass::vy1, a state struct's own feedback members, ...).
  // It becomes a feedback node; its source is the node with the same creation ordinal in the current pass.
  bool flushDenormals{false};  // the captured code constructed an ml::UsingFlushDenormalsToZero
  uint32_t epoch{1};
  int nextOrd{0};
  std::vector<int> nodeOfOrd;           // this pass: ordinal -> node
  std::map<int, int> feedbackOfOrd;     // previous-pass ordinal -> feedback node of this pass
  int computed(int node)                // register a node created by an op / processor / generator / routing call
  {
    nodeOfOrd.push_back(node);
    return nextOrd++;
  }
  int feedbackFor(int ord)
  {
    auto it = feedbackOfOrd.find(ord);
    if (it != feedbackOfOrd.end()) return it->second;
    return feedbackOfOrd[ord] = ret(mlgpu_graph_add_feedback(g, nullptr));
  }

  // Context signals - one row per instrument, not per voice: AudioContext::getInputController(n), getBeatPhase(). Each becomes
  // a streamed graph input shared by the `contextGroup` adjacent voices of an instrument (mlgpu_graph_set_input_group), added
  // where the captured code first asks for it.
  static constexpr int kBeatPhase = 1000;
  // Samples of a context signal on the host (`ctrlSig[0]` in the reference's controllers-to-audio.cpp): what the signal
  // holds in the DSPVector about to be processed, handed over by VoiceProgram::readContextSamples before update() runs the
  // process function again. Only a program of ONE context can have them (a host float is one number for the whole kernel).
  const std::map<int, std::array<float, 64>>* hostContext{nullptr};
  bool hostContextAllowed{false};
  float hostContextSample(int code, int n) const
  {
    if (!hostContextAllowed)
      throw std::logic_error("mldsp GPU shim: a sample of a context signal was read on the host (v[n]); construct the program with "
                             "VoiceProgramOptions::hostContextSamples (one context per program) and call readContextSamples() + update() per DSPVector");
    if (!hostContext) return 0.f;  // before anything was processed: controllers rest at 0, the transport at phase 0
    auto it = hostContext->find(code);
    return it == hostContext->end() ? 0.f : it->second[(size_t)(n & 63)];
  }
  int inputCount{0};               // graph inputs added so far in this pass
  size_t contextGroup{1};
  std::vector<int> contextInputs;  // this pass, in input order: a controller number, or kBeatPhase
  std::map<int, int> contextNode;
  int contextInput(int code);

  static Capture*& current()
  {
    static thread_local Capture* c = nullptr;
    return c;
  }
  static Capture& get()
  {
    if (!current()) throw std::logic_error("mldsp GPU shim: DSPVector arithmetic outside ml::gpu::VoiceProgram capture");
    return *current();
  }
  int ret(int r) const
  {
    if (r < 0) throw Error(-r, std::string("mlgpu_graph: ") + mlgpu_last_error(eng->handle()));
    return r;
  }
  // constant DSPVectors of this pass, by contents and by the rate region they are used in (a node belongs to one region)
  std::map<std::pair<int, std::array<uint32_t, 64>>, int> tableNodes;
  int regionCounter{0}, curRegion{0};
  int constantVector(const std::array<float, 64>& t)
  {
    std::pair<int, std::array<uint32_t, 64>> key;
    key.first = curRegion;
    std::memcpy(key.second.data(), t.data(), sizeof(key.second));
    auto it = tableNodes.find(key);
    if (it != tableNodes.end()) return it->second;
    return tableNodes[key] = ret(mlgpu_graph_add_const_vector(g, t.data(), nullptr));
  }
  bool dedupeConstants{true};
  int constant(float f)
  {
    if (!dedupeConstants) return ret(mlgpu_graph_add_const(g, f));
    uint32_t u;
    std::memcpy(&u, &f, 4);
    auto it = constNodes.find(u);
    if (it != constNodes.end()) return it->second;
    return constNodes[u] = ret(mlgpu_graph_add_const(g, f));
  }
  void deferCoeff(int node, int idx, float v)
  {
    Deferred d{node, 0, idx, 0};
    std::memcpy(&d.bits, &v, 4);
    deferred.push_back(d);
  }
  void deferClear(int node) { deferred.push_back(Deferred{node, 1, 0, 0}); }
  void deferState(int node, int idx, uint32_t bits) { deferred.push_back(Deferred{node, 2, idx, bits}); }
};
inline int Capture::contextInput(int code)
{
  auto it = contextNode.find(code);
  if (it != contextNode.end()) return it->second;
  const std::string name = code == kBeatPhase ? std::string("beatPhase") : "controller" + std::to_string(code);
  const int node = ret(mlgpu_graph_add_input(g, name.c_str()));
  int st = mlgpu_graph_set_input_group(g, inputCount, (int)contextGroup);
  if (st == MLGPU_OK) st = mlgpu_graph_set_input_layout(g, inputCount, MLGPU_LAYOUT_QUAD);  // whatever layout the audio inputs come in
  if (st != MLGPU_OK) throw Error(st, std::string("mlgpu_graph: ") + mlgpu_last_error(eng->handle()));
  ++inputCount;
  contextInputs.push_back(code);
  return contextNode[code] = node;
}


// ---- immediate mode -----------------------------------------------------------------------------------------
// DSPVector code that runs OUTSIDE a VoiceProgram capture - the reference's unit tests, an offline tool, a setup function
// that builds a table with DSPVector arithmetic - is evaluated at once, on the device, for ONE voice: every operator, free
// function, generator and filter call is one launch of the same kernels the fused programs are made of (mlgpu_op_apply, a
// one-voice mlgpu_graph around the processor object, mlgpu_row_reduce, ...) and hands back host data, exactly as the
// reference's value-type DSPVector does. It is slow (a round trip per call) and it is not the product path - that is
// VoiceProgram - but it is the same arithmetic, so code written against madronalib compiles and gives madronalib's bits
// before anybody has restructured it. There is no CPU arithmetic behind it: without a device these calls throw.
struct Eager
{
  const Engine* eng{nullptr};
  float* d{nullptr};  // device scratch, `cap` floats
  size_t cap{0};
  int device{0};
  std::recursive_mutex m;
  static Eager& get()
  {
    static Eager* e = new Eager();  // never destroyed: processor objects with static storage may outlive any static of ours
    return *e;
  }
  const Engine& engine()
  {
    if (!eng) eng = new Engine(device);
    return *eng;
  }
  float* scratch(size_t nFloats)
  {
    const Engine& e = engine();
    if (nFloats > cap)
    {
      if (d) mlgpu_free(e.handle(), d);
      d = nullptr;
      cap = 0;
      void* p = nullptr;
      const size_t want = nFloats < 1024 ? 1024 : nFloats;
      e.check(mlgpu_alloc(e.handle(), want * sizeof(float), &p));
      d = static_cast<float*>(p);
      cap = want;
    }
    return d;
  }
};
// the engine immediate-mode calls run on (default: one made on device 0 at the first call). Call before any such call.
inline void setImmediateEngine(const Engine& e) { Eager::get().eng = &e; }
inline const Engine& immediateEngine() { return Eager::get().engine(); }
inline bool capturing() { return Capture::current() != nullptr; }

}  // namespace gpu

// UsingFlushDenormalsToZero (MLDSPUtils.h:51-96): in the reference, MXCSR FZ | DAZ for the lifetime of the object -
// in practice the first statement of a process function (examples/audio-and-midi/fdtd.cpp:161). Captured code that
// constructs one makes its whole VoiceProgram run in the engine's flush mode (mlgpu_engine_set_flush_denormals around
// every launch of that program); the host thread's own MXCSR is set too, as in the reference, so host-side float
// arithmetic of the process function sees the same mode on both builds.
// Outside a capture (immediate mode) the scope puts the immediate-mode engine into flush mode for its lifetime: the launches made
// inside it run with denormals flushed, like the reference's instructions under the MXCSR bits.
struct UsingFlushDenormalsToZero
{
  int immediateWas_{-1};  // >= 0: the immediate engine's mode to restore
  void enter()
  {
    if (gpu::Capture::current())
    {
      gpu::Capture::current()->flushDenormals = true;
      return;
    }
    gpu::Eager& E = gpu::Eager::get();
    std::lock_guard<std::recursive_mutex> lock(E.m);
    if (!E.eng && mlgpu_device_count() <= 0) return;  // no device: nothing runs here anyway (the first DSPVector call will say so)
    const int was = mlgpu_engine_get_flush_denormals(E.engine().handle());
    E.engine().check(mlgpu_engine_set_flush_denormals(E.engine().handle(), 1));
    immediateWas_ = was > 0 ? 1 : 0;
  }
  void leave()
  {
    if (immediateWas_ >= 0) mlgpu_engine_set_flush_denormals(gpu::Eager::get().engine().handle(), immediateWas_);
  }
#if defined(__SSE__)
  unsigned MXCRState;
  UsingFlushDenormalsToZero() : MXCRState(__builtin_ia32_stmxcsr())
  {
    __builtin_ia32_ldmxcsr(MXCRState | 0x8040u);
    enter();
  }
  ~UsingFlushDenormalsToZero()
  {
    leave();
    __builtin_ia32_ldmxcsr(MXCRState);
  }
#else
  UsingFlushDenormalsToZero() { enter(); }
  ~UsingFlushDenormalsToZero() { leave(); }
#endif
  UsingFlushDenormalsToZero(const UsingFlushDenormalsToZero&) = delete;
  UsingFlushDenormalsToZero& operator=(const UsingFlushDenormalsToZero&) = delete;
};

namespace gpu
{
// one row of a DSPVectorArray: a graph node, or a float literal / a table of 64 floats not yet materialised
struct Sig
{
  int node{-1};
  float lit{0.f};
  std::shared_ptr<const std::array<float, 64>> table;  // DSPVector(const float*), DSPVector(fn), load(): host data, may be built
                                                       // outside a capture (a static window, a member filled in a setup function)
  uint32_t epoch{0};  // capture pass that made `node`
  int ord{-1};        // creation ordinal among the computed nodes of that pass (-1: input / param / const)
  bool hostMutable{false};  // `table` is a buffer the user writes through getBuffer(): copies take their own floats (value semantics)
  Sig(const Sig& o) : node(o.node), lit(o.lit), table(o.table), epoch(o.epoch), ord(o.ord), hostMutable(o.hostMutable), hostCtx(o.hostCtx)
  {
    if (hostMutable && table) table = std::make_shared<const std::array<float, 64>>(*table);
  }
  Sig& operator=(const Sig& o)
  {
    if (this != &o)
    {
      node = o.node; lit = o.lit; table = o.table; epoch = o.epoch; ord = o.ord; hostMutable = o.hostMutable; hostCtx = o.hostCtx;
      if (hostMutable && table) table = std::make_shared<const std::array<float, 64>>(*table);
    }
    return *this;
  }
  int hostCtx{-1};    // >= 0: the node is the context signal of that code (a controller number / kBeatPhase): its samples can be read on the host
  // one sample of this signal as a host float - possible only where the host has the data (MLDSPOps.h:167-168 `operator[]`)
  float hostSample(int n) const
  {
    if (node < 0 && table) return (*table)[(size_t)(n & 63)];
    if (node < 0) return lit;
    if (hostCtx >= 0) return Capture::get().hostContextSample(hostCtx, n);
    throw std::logic_error("mldsp GPU shim: v[n] / getBuffer() on a signal the kernel computes - its samples exist only on the device; "
                           "keep the step in DSPVector form, or read the signal back after a launch");
  }
  Sig() {}
  Sig(int n, float l) : node(n), lit(l), epoch(n >= 0 ? Capture::get().epoch : 0) {}
  explicit Sig(std::shared_ptr<const std::array<float, 64>> t) : table(std::move(t)) {}  // immediate mode: a result the host now holds
  void hostCopy(float* dst64) const
  {
    for (int i = 0; i < 64; ++i) dst64[i] = hostSample(i);
  }
  explicit Sig(const float* p64) : table(std::make_shared<const std::array<float, 64>>(toArray(p64))) {}
  static std::array<float, 64> toArray(const float* p)
  {
    std::array<float, 64> a;
    std::memcpy(a.data(), p, sizeof(a));
    return a;
  }
  int id() const
  {
    if (node < 0 && table) return Capture::get().constantVector(*table);
    if (node < 0) return Capture::get().constant(lit);
    Capture& c = Capture::get();
    if (epoch == c.epoch) return node;
    if (ord < 0) throw std::logic_error("mldsp GPU shim: a DSPVector kept from an earlier process call holds an input / parameter, not a computed signal");
    return c.feedbackFor(ord);
  }
};
inline Sig computedSig(int node)
{
  Sig s(node, 0.f);
  s.ord = Capture::get().computed(node);
  return s;
}

// immediate mode: out = op(in...) over one DSPVector, on the device (mlgpu_op_apply)
inline Sig immediateOp(int op, std::initializer_list<Sig> in)
{
  Eager& E = Eager::get();
  std::lock_guard<std::recursive_mutex> lock(E.m);
  const Engine& e = E.engine();
  float* d = E.scratch(4 * 64);
  float host[3 * 64];
  int n = 0;
  for (const Sig& s : in) s.hostCopy(host + 64 * n++);
  e.check(mlgpu_upload(e.handle(), d, host, (size_t)n * 256));
  e.check(mlgpu_op_apply(e.handle(), op, d, n > 1 ? d + 64 : nullptr, n > 2 ? d + 128 : nullptr, d + 192, 64));
  auto out = std::make_shared<std::array<float, 64>>();
  e.check(mlgpu_download(e.handle(), out->data(), d + 192, 256));
  return Sig(std::shared_ptr<const std::array<float, 64>>(std::move(out)));
}

inline Sig opNode(int op, std::initializer_list<Sig> in)
{
  if (!Capture::current()) return immediateOp(op, in);
  Capture& c = Capture::get();
  int ids[3];
  int n = 0;
  for (const Sig& s : in) ids[n++] = s.id();
  return computedSig(c.ret(mlgpu_graph_add_op(c.g, op, ids, n, nullptr)));
}
inline Sig vopNode(int vop, std::initializer_list<Sig> in)
{
  if (!Capture::current())
  {
    // immediate mode: the reference's own expressions (MLDSPOps.h:965-990) - the interval is host float arithmetic there too
    auto idx = std::make_shared<std::array<float, 64>>();
    for (int i = 0; i < 64; ++i) (*idx)[(size_t)i] = (float)i;
    const Sig column{std::shared_ptr<const std::array<float, 64>>(std::move(idx))};
    if (vop == MLGPU_VOP_COLUMN_INDEX) return column;
    const float start = in.begin()->hostSample(0), end = (in.begin() + 1)->hostSample(0);
    const float interval = (end - start) / (vop == MLGPU_VOP_RANGE_CLOSED ? (64 - 1.f) : 64.f);
    const float offset = vop == MLGPU_VOP_INTERPOLATE_LINEAR ? start + interval : start;
    return immediateOp(MLGPU_OP_ADD, {immediateOp(MLGPU_OP_MULTIPLY, {column, Sig(-1, interval)}), Sig(-1, offset)});
  }
  Capture& c = Capture::get();
  int ids[2];
  int n = 0;
  for (const Sig& s : in) ids[n++] = s.id();
  return computedSig(c.ret(mlgpu_graph_add_vop(c.g, vop, ids, n, nullptr)));
}
}  // namespace gpu

// ---- DSPVectorArray<ROWS>, DSPVector, DSPVectorArrayInt<ROWS> (MLDSPOps.h:94-498) -------------------------------

template <size_t ROWS>
class DSPVectorArray;
template <size_t ROWS>
DSPVectorArray<ROWS> add(const DSPVectorArray<ROWS>&, const DSPVectorArray<ROWS>&);
template <size_t ROWS>
DSPVectorArray<ROWS> subtract(const DSPVectorArray<ROWS>&, const DSPVectorArray<ROWS>&);
template <size_t ROWS>
DSPVectorArray<ROWS> multiply(const DSPVectorArray<ROWS>&, const DSPVectorArray<ROWS>&);
template <size_t ROWS>
DSPVectorArray<ROWS> divide(const DSPVectorArray<ROWS>&, const DSPVectorArray<ROWS>&);

template <size_t ROWS>
class DSPVectorArray
{
 public:
  std::array<gpu::Sig, ROWS> sig_;  // implementation detail of the shim

  DSPVectorArray() {}  // zero-filled, MLDSPOps.h:153
  DSPVectorArray(float k)  // broadcast conversion ctor, MLDSPOps.h:157
  {
    for (auto& s : sig_) s = gpu::Sig(-1, k);
  }
  explicit DSPVectorArray(gpu::Sig s)
  {
    static_assert(ROWS == 1, "a single signal is a DSPVector");
    sig_[0] = s;
  }
  // host data: the same 64 * ROWS floats for every voice and every process call (MLDSPOps.h:140-161). Evaluated on the host,
  // carried into the kernel as constant tables (mlgpu_graph_add_const_vector).
  explicit DSPVectorArray(const float* pData)
  {
    for (size_t j = 0; j < ROWS; ++j) sig_[j] = gpu::Sig(pData + j * 64);
  }
  explicit DSPVectorArray(float* pData) : DSPVectorArray(static_cast<const float*>(pData)) {}
  DSPVectorArray(std::array<float, 64 * ROWS> a) : DSPVectorArray(a.data()) {}
  DSPVectorArray(float (*fn)(int))
  {
    std::array<float, 64 * ROWS> a;
    for (size_t i = 0; i < 64 * ROWS; ++i) a[i] = fn((int)i);
    for (size_t j = 0; j < ROWS; ++j) sig_[j] = gpu::Sig(a.data() + j * 64);
  }
  DSPVectorArray& operator=(float k)
  {
    for (auto& s : sig_) s = gpu::Sig(-1, k);
    return *this;
  }

  // The 64 * ROWS floats of a vector the HOST made (MLDSPOps.h:130-136 getBuffer / getConstBuffer): a default-constructed or
  // literal vector, host tables. The first call gives the vector its own host buffer; what the host writes there is what the
  // kernel gets when the vector is next used (as constant tables). A computed signal has no host buffer.
  float* getBuffer()
  {
    bool mine = true;  // already one contiguous host buffer of this vector's own?
    for (size_t j = 0; j < ROWS; ++j) mine = mine && sig_[j].hostMutable && sig_[j].table && sig_[j].table.get() == sig_[0].table.get() + j;
    if (!mine)
    {
      auto buf = std::make_shared<std::vector<float>>(64 * ROWS);
      for (size_t i = 0; i < 64 * ROWS; ++i) (*buf)[i] = sig_[i / 64].hostSample((int)(i & 63));
      for (size_t j = 0; j < ROWS; ++j)
      {
        gpu::Sig& t = sig_[j];  // field by field: assigning a mutable Sig would give it a copy of its floats
        t.node = -1;
        t.lit = 0.f;
        t.epoch = 0;
        t.ord = t.hostCtx = -1;
        t.hostMutable = true;
        t.table = std::shared_ptr<const std::array<float, 64>>(buf, reinterpret_cast<const std::array<float, 64>*>(buf->data() + 64 * j));
      }
    }
    return const_cast<float*>(sig_[0].table->data());
  }
  const float* getConstBuffer() const { return const_cast<DSPVectorArray*>(this)->getBuffer(); }

  // one sample on the host (MLDSPOps.h:167-168): literals, host tables, immediate-mode results and - in a one-context program -
  // context signals (gpu::Sig::hostSample). The writable form gives the vector its own host buffer, like getBuffer().
  float operator[](size_t i) const { return sig_[i / 64].hostSample((int)(i & 63)); }
  float& operator[](size_t i)
  {
    const gpu::Sig& s = sig_[i / 64];
    if (s.node >= 0)  // a signal of the kernel (a context signal in a one-context program): readable, not a host buffer
    {
      // The reference hands out a float& here, so this must too - into a small ring of read-only cells. A write through it would
      // vanish; it is caught at the next access instead of passing silently: every cell remembers what it was given.
      static thread_local float readOnly[8], given[8];
      static thread_local unsigned next;
      for (unsigned k = 0; k < 8 && k < next; ++k)
        if (std::memcmp(&readOnly[k], &given[k], sizeof(float)) != 0)
        {
          std::memcpy(&readOnly[k], &given[k], sizeof(float));
          throw std::logic_error("mldsp GPU shim: `v[n] = x` on a signal the kernel computes was discarded - such a signal can be read on the host, "
                                 "not written; build the value with DSPVector operations instead");
        }
      const unsigned k = next++ & 7;
      readOnly[k] = given[k] = s.hostSample((int)(i & 63));
      return readOnly[k];
    }
    return getBuffer()[i];
  }
  // equality by value (MLDSPOps.h:190-200): host data on both sides
  bool operator==(const DSPVectorArray& x) const
  {
    for (size_t n = 0; n < 64 * ROWS; ++n)
      if ((*this)[n] != x[n]) return false;
    return true;
  }
  bool operator!=(const DSPVectorArray& x) const { return !(*this == x); }

  DSPVectorArray<1>& row(int j) { return *reinterpret_cast<DSPVectorArray<1>*>(&sig_[j]); }
  const DSPVectorArray<1>& constRow(int j) const { return *reinterpret_cast<const DSPVectorArray<1>*>(&sig_[j]); }
  DSPVectorArray<1> getRowVectorUnchecked(size_t j) const { return constRow((int)j); }
  void setRowVectorUnchecked(size_t j, const DSPVectorArray<1> x) { row((int)j) = x; }
  template <int J>
  DSPVectorArray<1> getRowVector() const
  {
    static_assert((J >= 0) && (J < (int)ROWS), "getRowVector index out of bounds");
    return constRow(J);
  }
  template <int J>
  void setRowVector(const DSPVectorArray<1> x)
  {
    static_assert((J >= 0) && (J < (int)ROWS), "setRowVector index out of bounds");
    row(J) = x;
  }

  DSPVectorArray& operator+=(const DSPVectorArray& x) { return *this = add(*this, x); }
  DSPVectorArray& operator-=(const DSPVectorArray& x) { return *this = subtract(*this, x); }
  DSPVectorArray& operator*=(const DSPVectorArray& x) { return *this = multiply(*this, x); }
  DSPVectorArray& operator/=(const DSPVectorArray& x) { return *this = divide(*this, x); }
  // in-class friends so either operand converts implicitly from float (MLDSPOps.h:333-352)
  friend DSPVectorArray operator+(const DSPVectorArray& a, const DSPVectorArray& b) { return add(a, b); }
  friend DSPVectorArray operator-(const DSPVectorArray& a, const DSPVectorArray& b) { return subtract(a, b); }
  friend DSPVectorArray operator*(const DSPVectorArray& a, const DSPVectorArray& b) { return multiply(a, b); }
  friend DSPVectorArray operator/(const DSPVectorArray& a, const DSPVectorArray& b) { return divide(a, b); }
};
typedef DSPVectorArray<1> DSPVector;

template <size_t ROWS>
class DSPVectorArrayInt  // int32 masks / integers travel as bit patterns (MLDSPOps.h:370-498)
{
 public:
  std::array<gpu::Sig, ROWS> sig_;
  // integers the host knows (columnIndexInt()): what map(float(int), x) hands to its function; empty for computed signals
  std::array<std::shared_ptr<const std::array<int32_t, 64>>, ROWS> host_;
  DSPVectorArrayInt() {}
  int32_t hostInt(int n) const
  {
    const auto& t = host_[(size_t)n / 64];
    if (!t) throw std::logic_error("mldsp GPU shim: map(float(int), x) needs integers the host knows (columnIndexInt()); x is a computed signal");
    return (*t)[(size_t)(n & 63)];
  }
};
typedef DSPVectorArrayInt<1> DSPVectorInt;

class DSPVectorDynamic final  // MLDSPOps.h:503-517
{
 public:
  DSPVectorDynamic() = default;
  explicit DSPVectorDynamic(size_t rows) { data_.resize(rows); }
  void resize(size_t rows) { data_.resize(rows); }
  size_t size() const { return data_.size(); }
  DSPVector& operator[](int j) { return data_[j]; }
  const DSPVector& operator[](int j) const { return data_[j]; }

 private:
  std::vector<DSPVector> data_;
};

// ---- the DEFINE_OP* families (MLDSPOps.h:570-936) -----------------------------------------------------------------

#define MLGPU_SHIM_OP1(name, OP)                                       \
  template <size_t ROWS>                                               \
  inline DSPVectorArray<ROWS> name(const DSPVectorArray<ROWS>& a)      \
  {                                                                    \
    DSPVectorArray<ROWS> y;                                            \
    for (size_t j = 0; j < ROWS; ++j) y.sig_[j] = gpu::opNode(OP, {a.sig_[j]}); \
    return y;                                                          \
  }
MLGPU_SHIM_OP1(sqrt, MLGPU_OP_SQRT)
MLGPU_SHIM_OP1(sqrtApprox, MLGPU_OP_SQRT_APPROX)
MLGPU_SHIM_OP1(abs, MLGPU_OP_ABS)
MLGPU_SHIM_OP1(sign, MLGPU_OP_SIGN)
MLGPU_SHIM_OP1(signBit, MLGPU_OP_SIGN_BIT)
MLGPU_SHIM_OP1(sin, MLGPU_OP_SIN)
MLGPU_SHIM_OP1(cos, MLGPU_OP_COS)
MLGPU_SHIM_OP1(log, MLGPU_OP_LOG)
MLGPU_SHIM_OP1(exp, MLGPU_OP_EXP)
MLGPU_SHIM_OP1(log2, MLGPU_OP_LOG2)
MLGPU_SHIM_OP1(exp2, MLGPU_OP_EXP2)
MLGPU_SHIM_OP1(sinApprox, MLGPU_OP_SIN_APPROX)
MLGPU_SHIM_OP1(cosApprox, MLGPU_OP_COS_APPROX)
MLGPU_SHIM_OP1(expApprox, MLGPU_OP_EXP_APPROX)
MLGPU_SHIM_OP1(logApprox, MLGPU_OP_LOG_APPROX)
MLGPU_SHIM_OP1(log2Approx, MLGPU_OP_LOG2_APPROX)
MLGPU_SHIM_OP1(exp2Approx, MLGPU_OP_EXP2_APPROX)
MLGPU_SHIM_OP1(fractionalPart, MLGPU_OP_FRACTIONAL_PART)

#define MLGPU_SHIM_OP2(name, OP)                                                                      \
  template <size_t ROWS>                                                                              \
  inline DSPVectorArray<ROWS> name(const DSPVectorArray<ROWS>& a, const DSPVectorArray<ROWS>& b)      \
  {                                                                                                   \
    DSPVectorArray<ROWS> y;                                                                           \
    for (size_t j = 0; j < ROWS; ++j) y.sig_[j] = gpu::opNode(OP, {a.sig_[j], b.sig_[j]});            \
    return y;                                                                                         \
  }
MLGPU_SHIM_OP2(add, MLGPU_OP_ADD)
MLGPU_SHIM_OP2(subtract, MLGPU_OP_SUBTRACT)
MLGPU_SHIM_OP2(multiply, MLGPU_OP_MULTIPLY)
MLGPU_SHIM_OP2(divide, MLGPU_OP_DIVIDE)
MLGPU_SHIM_OP2(divideApprox, MLGPU_OP_DIVIDE_APPROX)
MLGPU_SHIM_OP2(pow, MLGPU_OP_POW)
MLGPU_SHIM_OP2(powApprox, MLGPU_OP_POW_APPROX)
MLGPU_SHIM_OP2(min, MLGPU_OP_MIN)
MLGPU_SHIM_OP2(max, MLGPU_OP_MAX)

// ROWS x 1-row broadcast forms add1..max1 (MLDSPOps.h:655-687)
#define MLGPU_SHIM_OP2_1(name, OP)                                                                \
  template <size_t ROWS>                                                                          \
  inline DSPVectorArray<ROWS> name(const DSPVectorArray<ROWS>& a, const DSPVectorArray<1>& b)     \
  {                                                                                               \
    DSPVectorArray<ROWS> y;                                                                       \
    for (size_t j = 0; j < ROWS; ++j) y.sig_[j] = gpu::opNode(OP, {a.sig_[j], b.sig_[0]});        \
    return y;                                                                                     \
  }
MLGPU_SHIM_OP2_1(add1, MLGPU_OP_ADD)
MLGPU_SHIM_OP2_1(subtract1, MLGPU_OP_SUBTRACT)
MLGPU_SHIM_OP2_1(multiply1, MLGPU_OP_MULTIPLY)
MLGPU_SHIM_OP2_1(divide1, MLGPU_OP_DIVIDE)
MLGPU_SHIM_OP2_1(divideApprox1, MLGPU_OP_DIVIDE_APPROX)
MLGPU_SHIM_OP2_1(pow1, MLGPU_OP_POW)
MLGPU_SHIM_OP2_1(powApprox1, MLGPU_OP_POW_APPROX)
MLGPU_SHIM_OP2_1(min1, MLGPU_OP_MIN)
MLGPU_SHIM_OP2_1(max1, MLGPU_OP_MAX)

#define MLGPU_SHIM_OP3(name, OP)                                                                              \
  template <size_t ROWS>                                                                                      \
  inline DSPVectorArray<ROWS> name(const DSPVectorArray<ROWS>& a, const DSPVectorArray<ROWS>& b,              \
                                   const DSPVectorArray<ROWS>& c)                                             \
  {                                                                                                           \
    DSPVectorArray<ROWS> y;                                                                                   \
    for (size_t j = 0; j < ROWS; ++j) y.sig_[j] = gpu::opNode(OP, {a.sig_[j], b.sig_[j], c.sig_[j]});         \
    return y;                                                                                                 \
  }
MLGPU_SHIM_OP3(lerp, MLGPU_OP_LERP)                // lerp(a, b, mix)
MLGPU_SHIM_OP3(inverseLerp, MLGPU_OP_INVERSE_LERP) // inverseLerp(a, b, x)
MLGPU_SHIM_OP3(clamp, MLGPU_OP_CLAMP)              // clamp(x, lo, hi)

template <size_t ROWS>  // lerp with a float mix, MLDSPOps.h:753-774
inline DSPVectorArray<ROWS> lerp(const DSPVectorArray<ROWS>& a, const DSPVectorArray<ROWS>& b, float m)
{
  return lerp(a, b, DSPVectorArray<ROWS>(m));
}

template <size_t ROWS>  // within(x, lo, hi): lo <= x < hi as a mask in a FLOAT vector (DEFINE_OP3, MLDSPOps.h:748: all bits set or none)
inline DSPVectorArray<ROWS> within(const DSPVectorArray<ROWS>& x, const DSPVectorArray<ROWS>& lo, const DSPVectorArray<ROWS>& hi)
{
  DSPVectorArray<ROWS> y;
  for (size_t j = 0; j < ROWS; ++j) y.sig_[j] = gpu::opNode(MLGPU_OP_WITHIN, {x.sig_[j], lo.sig_[j], hi.sig_[j]});
  return y;
}

#define MLGPU_SHIM_CMP(name, OP)                                                                        \
  template <size_t ROWS>                                                                                \
  inline DSPVectorArrayInt<ROWS> name(const DSPVectorArray<ROWS>& a, const DSPVectorArray<ROWS>& b)     \
  {                                                                                                     \
    DSPVectorArrayInt<ROWS> y;                                                                          \
    for (size_t j = 0; j < ROWS; ++j) y.sig_[j] = gpu::opNode(OP, {a.sig_[j], b.sig_[j]});              \
    return y;                                                                                           \
  }
MLGPU_SHIM_CMP(equal, MLGPU_OP_EQUAL)
MLGPU_SHIM_CMP(notEqual, MLGPU_OP_NOT_EQUAL)
MLGPU_SHIM_CMP(greaterThan, MLGPU_OP_GREATER_THAN)
MLGPU_SHIM_CMP(greaterThanOrEqual, MLGPU_OP_GREATER_THAN_OR_EQUAL)
MLGPU_SHIM_CMP(lessThan, MLGPU_OP_LESS_THAN)
MLGPU_SHIM_CMP(lessThanOrEqual, MLGPU_OP_LESS_THAN_OR_EQUAL)

template <size_t ROWS>  // select(a, b, mask): bitwise, MLDSPOps.h:886
inline DSPVectorArray<ROWS> select(const DSPVectorArray<ROWS>& a, const DSPVectorArray<ROWS>& b, const DSPVectorArrayInt<ROWS>& m)
{
  DSPVectorArray<ROWS> y;
  for (size_t j = 0; j < ROWS; ++j) y.sig_[j] = gpu::opNode(MLGPU_OP_SELECT, {a.sig_[j], b.sig_[j], m.sig_[j]});
  return y;
}
template <size_t ROWS>
inline DSPVectorArrayInt<ROWS> select(const DSPVectorArrayInt<ROWS>& a, const DSPVectorArrayInt<ROWS>& b, const DSPVectorArrayInt<ROWS>& m)
{
  DSPVectorArrayInt<ROWS> y;
  for (size_t j = 0; j < ROWS; ++j) y.sig_[j] = gpu::opNode(MLGPU_OP_SELECT_INT, {a.sig_[j], b.sig_[j], m.sig_[j]});
  return y;
}

#define MLGPU_SHIM_CONV(name, OP, FROM, TO)                       \
  template <size_t ROWS>                                          \
  inline TO<ROWS> name(const FROM<ROWS>& a)                       \
  {                                                               \
    TO<ROWS> y;                                                   \
    for (size_t j = 0; j < ROWS; ++j) y.sig_[j] = gpu::opNode(OP, {a.sig_[j]}); \
    return y;                                                     \
  }
MLGPU_SHIM_CONV(roundFloatToInt, MLGPU_OP_ROUND_FLOAT_TO_INT, DSPVectorArray, DSPVectorArrayInt)
MLGPU_SHIM_CONV(truncateFloatToInt, MLGPU_OP_TRUNCATE_FLOAT_TO_INT, DSPVectorArray, DSPVectorArrayInt)
MLGPU_SHIM_CONV(intToFloat, MLGPU_OP_INT_TO_FLOAT, DSPVectorArrayInt, DSPVectorArray)
MLGPU_SHIM_CONV(unsignedIntToFloat, MLGPU_OP_UNSIGNED_INT_TO_FLOAT, DSPVectorArrayInt, DSPVectorArray)

template <size_t ROWS>
inline DSPVectorArrayInt<ROWS> addInt32(const DSPVectorArrayInt<ROWS>& a, const DSPVectorArrayInt<ROWS>& b)
{
  DSPVectorArrayInt<ROWS> y;
  for (size_t j = 0; j < ROWS; ++j) y.sig_[j] = gpu::opNode(MLGPU_OP_ADD_INT32, {a.sig_[j], b.sig_[j]});
  return y;
}
template <size_t ROWS>
inline DSPVectorArrayInt<ROWS> subtractInt32(const DSPVectorArrayInt<ROWS>& a, const DSPVectorArrayInt<ROWS>& b)
{
  DSPVectorArrayInt<ROWS> y;
  for (size_t j = 0; j < ROWS; ++j) y.sig_[j] = gpu::opNode(MLGPU_OP_SUBTRACT_INT32, {a.sig_[j], b.sig_[j]});
  return y;
}

// n-ary add, right fold a + (b + (c + ...)), MLDSPOps.h:925-936
template <size_t ROWS>
inline DSPVectorArray<ROWS> add(const DSPVectorArray<ROWS>& a)
{
  return a;
}
template <size_t ROWS, typename... Args>
inline DSPVectorArray<ROWS> add(const DSPVectorArray<ROWS>& first, const DSPVectorArray<ROWS>& second, const DSPVectorArray<ROWS>& third,
                                Args... args)
{
  return add(first, add(second, third, args...));
}

// ---- index generators (MLDSPOps.h:962-990, 1365-1383) ---------------------------------------------------------------
// load (MLDSPOps.h: load(DSPVectorArray&, const float*)): host floats into a vector — a constant table in the captured kernel.
// store() has no counterpart: a captured DSPVector has no host values; route it to ctx->outputs or a published signal.
template <size_t ROWS>
inline void load(DSPVectorArray<ROWS>& vecDest, const float* pSrc)
{
  vecDest = DSPVectorArray<ROWS>(pSrc);
}
template <size_t ROWS>
inline void loadAligned(DSPVectorArray<ROWS>& vecDest, const float* pSrc)
{
  vecDest = DSPVectorArray<ROWS>(pSrc);
}
// store (MLDSPOps.h:530-534, 551-562): the vector's floats to host memory - for vectors the host holds (literals, host tables,
// immediate-mode results); a signal of a captured kernel has no host values (Sig::hostSample says so)
template <size_t ROWS>
inline void store(const DSPVectorArray<ROWS>& vecSrc, float* pDest)
{
  for (size_t n = 0; n < 64 * ROWS; ++n) pDest[n] = vecSrc[n];
}
template <size_t ROWS>
inline void storeAligned(const DSPVectorArray<ROWS>& vecSrc, float* pDest)
{
  store(vecSrc, pDest);
}

// ---- horizontal operators returning float (MLDSPOps.h:995-1035), normalize (:1041-1050), rotateLeft / rotateRight (:1219-1276) ----
// A float result is host data, so these exist in immediate mode only (inside a capture there is nothing to return): the
// reduction runs on the device with the reference's association order and seeds (mlgpu_row_reduce), one row at a time.
namespace gpu
{
inline float immediateReduce(int rowop, const Sig& row, const char* what)
{
  if (Capture::current())
    throw std::logic_error(std::string("mldsp GPU shim: ") + what + "(DSPVector) returns a float to the host; inside a captured process function "
                           "there is no host value - keep the step in DSPVector form (e.g. divide by a DSPVector) or compute it outside the capture");
  Eager& E = Eager::get();
  std::lock_guard<std::recursive_mutex> lock(E.m);
  const Engine& e = E.engine();
  float* d = E.scratch(64 + 4);
  float host[64];
  row.hostCopy(host);
  e.check(mlgpu_upload(e.handle(), d, host, 256));
  e.check(mlgpu_row_reduce(e.handle(), rowop, d, d + 64, 1));
  float r = 0.f;
  e.check(mlgpu_download(e.handle(), &r, d + 64, 4));
  return r;
}
template <size_t ROWS>
inline DSPVectorArray<ROWS> immediateRowsMap(const DSPVectorArray<ROWS>& x, int sampleRotate, const char* what)
{
  if (Capture::current())
    throw std::logic_error(std::string("mldsp GPU shim: ") + what + " moves samples across a DSPVector; a captured kernel walks a vector sample by sample - "
                           "use it outside the capture, or mlgpu_rows_map on a signal in memory");
  Eager& E = Eager::get();
  std::lock_guard<std::recursive_mutex> lock(E.m);
  const Engine& e = E.engine();
  float* d = E.scratch(2 * 64 * ROWS);
  std::vector<float> host(64 * ROWS);
  for (size_t n = 0; n < 64 * ROWS; ++n) host[n] = x[n];
  e.check(mlgpu_upload(e.handle(), d, host.data(), host.size() * 4));
  e.check(mlgpu_rows_map(e.handle(), MLGPU_ROWS_REPEAT, 0, 0, sampleRotate, d, ROWS, d + 64 * ROWS, ROWS, 0, 1, ROWS, 1));
  e.check(mlgpu_download(e.handle(), host.data(), d + 64 * ROWS, host.size() * 4));
  return DSPVectorArray<ROWS>(static_cast<const float*>(host.data()));
}
}  // namespace gpu
inline float sum(const DSPVector& x) { return gpu::immediateReduce(MLGPU_ROWOP_SUM, x.sig_[0], "sum"); }
inline float mean(const DSPVector& x) { return gpu::immediateReduce(MLGPU_ROWOP_MEAN, x.sig_[0], "mean"); }
inline float max(const DSPVector& x) { return gpu::immediateReduce(MLGPU_ROWOP_MAX, x.sig_[0], "max"); }
inline float min(const DSPVector& x) { return gpu::immediateReduce(MLGPU_ROWOP_MIN, x.sig_[0], "min"); }
template <size_t ROWS>
inline DSPVectorArray<ROWS> normalize(const DSPVectorArray<ROWS>& x)
{
  DSPVectorArray<ROWS> vy;
  for (size_t j = 0; j < ROWS; ++j)
  {
    const DSPVector row = x.constRow((int)j);
    vy.row((int)j) = row / DSPVector(sum(row));
  }
  return vy;
}
template <size_t ROWS>
inline DSPVectorArray<ROWS> rotateLeft(const DSPVectorArray<ROWS>& x)
{
  return gpu::immediateRowsMap(x, +1, "rotateLeft");
}
template <size_t ROWS>
inline DSPVectorArray<ROWS> rotateRight(const DSPVectorArray<ROWS>& x)
{
  return gpu::immediateRowsMap(x, -1, "rotateRight");
}
template <size_t ROWS>
inline std::ostream& operator<<(std::ostream& out, const DSPVectorArray<ROWS>& v)  // MLDSPOps.h:1390-1406
{
  for (size_t r = 0; r < ROWS; ++r)
  {
    if (ROWS > 1) out << "\n    v" << r << ": ";
    out << "[";
    for (size_t i = 0; i < 64; ++i) out << v[r * 64 + i] << " ";
    out << "] ";
  }
  return out;
}
// validate(DSPVector) (MLDSPOps.h:1430-1445): false, with a report on std::cout, when a sample is NaN or beyond +-1e8. It reads the
// vector's samples on the host, so it works wherever v[n] does - literals, host tables, the results of immediate mode - and throws
// like v[n] on a signal that exists only inside a captured kernel (there mlgpu_validate scans the output buffer after a launch).
inline bool validate(const DSPVector& x)
{
  for (size_t n = 0; n < 64; ++n)
  {
    const float maxUsefulValue = 1e8;
    const float v = x[n];
    if (std::isnan(v) || (std::fabs(v) > maxUsefulValue))
    {
      std::cout << "error: " << v << " at index " << n << "\n";
      std::cout << x << "\n";
      return false;
    }
  }
  return true;
}
inline DSPVector columnIndex() { return DSPVector(gpu::vopNode(MLGPU_VOP_COLUMN_INDEX, {})); }
inline DSPVector rangeOpen(float start, float end) { return DSPVector(gpu::vopNode(MLGPU_VOP_RANGE_OPEN, {gpu::Sig(-1, start), gpu::Sig(-1, end)})); }
inline DSPVector rangeClosed(float start, float end)
{
  return DSPVector(gpu::vopNode(MLGPU_VOP_RANGE_CLOSED, {gpu::Sig(-1, start), gpu::Sig(-1, end)}));
}
inline DSPVector interpolateDSPVectorLinear(float start, float end)
{
  return DSPVector(gpu::vopNode(MLGPU_VOP_INTERPOLATE_LINEAR, {gpu::Sig(-1, start), gpu::Sig(-1, end)}));
}

// ---- row plumbing (MLDSPOps.h:1057-1383): with rows as separate signals this is wiring, no arithmetic ---------------
template <size_t ROWS, size_t N>
inline DSPVectorArray<ROWS * N> repeatRows(const DSPVectorArray<N>& x)
{
  DSPVectorArray<ROWS * N> y;
  for (size_t j = 0, k = 0; j < ROWS * N; ++j)
  {
    y.sig_[j] = x.sig_[k];
    if (++k >= N) k = 0;
  }
  return y;
}
template <size_t ROWS, size_t N>
inline DSPVectorArray<ROWS> stretchRows(const DSPVectorArray<N>& x)
{
  DSPVectorArray<ROWS> y;
  for (size_t j = 0; j < ROWS; ++j) y.sig_[j] = x.sig_[(size_t)roundf((j * (N - 1.f)) / (ROWS - 1.f))];
  return y;
}
template <size_t ROWS, size_t N>
inline DSPVectorArray<ROWS> zeroPadRows(const DSPVectorArray<N>& x)
{
  DSPVectorArray<ROWS> y;
  for (size_t j = 0; j < (ROWS < N ? ROWS : N); ++j) y.sig_[j] = x.sig_[j];
  return y;
}
template <size_t ROWS>
inline DSPVectorArray<ROWS> shiftRows(const DSPVectorArray<ROWS>& x, int rowsToShift)
{
  DSPVectorArray<ROWS> y;
  int k = -rowsToShift;
  for (size_t j = 0; j < ROWS; ++j, ++k)
    if (k >= 0 && k < (int)ROWS) y.sig_[j] = x.sig_[k];
  return y;
}
template <size_t ROWS>
inline DSPVectorArray<ROWS> rotateRows(const DSPVectorArray<ROWS>& x, int rowsToRotate)
{
  DSPVectorArray<ROWS> y;
  int k = (-rowsToRotate) % (int)ROWS;
  if (k < 0) k += (int)ROWS;
  for (size_t j = 0; j < ROWS; ++j)
  {
    y.sig_[j] = x.sig_[k];
    if (++k >= (int)ROWS) k = 0;
  }
  return y;
}
template <size_t ROWSA, size_t ROWSB>
inline DSPVectorArray<ROWSA + ROWSB> concatRows(const DSPVectorArray<ROWSA>& a, const DSPVectorArray<ROWSB>& b)
{
  DSPVectorArray<ROWSA + ROWSB> y;
  for (size_t j = 0; j < ROWSA; ++j) y.sig_[j] = a.sig_[j];
  for (size_t j = 0; j < ROWSB; ++j) y.sig_[ROWSA + j] = b.sig_[j];
  return y;
}
template <size_t ROWSA, size_t ROWSB, size_t ROWSC>
inline DSPVectorArray<ROWSA + ROWSB + ROWSC> concatRows(const DSPVectorArray<ROWSA>& a, const DSPVectorArray<ROWSB>& b, const DSPVectorArray<ROWSC>& c)
{
  return concatRows(concatRows(a, b), c);
}
template <size_t ROWSA, size_t ROWSB, size_t ROWSC, size_t ROWSD>
inline DSPVectorArray<ROWSA + ROWSB + ROWSC + ROWSD> concatRows(const DSPVectorArray<ROWSA>& a, const DSPVectorArray<ROWSB>& b,
                                                                const DSPVectorArray<ROWSC>& c, const DSPVectorArray<ROWSD>& d)
{
  return concatRows(concatRows(a, b, c), d);
}
template <size_t ROWSA, size_t ROWSB>
inline DSPVectorArray<ROWSA + ROWSB> shuffleRows(const DSPVectorArray<ROWSA> a, const DSPVectorArray<ROWSB> b)
{
  DSPVectorArray<ROWSA + ROWSB> y;
  size_t ja = 0, jb = 0, jy = 0;
  while ((ja < ROWSA) || (jb < ROWSB))
  {
    if (ja < ROWSA) y.sig_[jy++] = a.sig_[ja++];
    if (jb < ROWSB) y.sig_[jy++] = b.sig_[jb++];
  }
  return y;
}
template <size_t ROWS>
inline DSPVectorArray<(ROWS + 1) / 2> evenRows(const DSPVectorArray<ROWS>& x)
{
  DSPVectorArray<(ROWS + 1) / 2> y;
  for (size_t j = 0; j < (ROWS + 1) / 2; ++j) y.sig_[j] = x.sig_[j * 2];
  return y;
}
template <size_t ROWS>
inline DSPVectorArray<ROWS / 2> oddRows(const DSPVectorArray<ROWS>& x)
{
  DSPVectorArray<ROWS / 2> y;
  for (size_t j = 0; j < ROWS / 2; ++j) y.sig_[j] = x.sig_[j * 2 + 1];
  return y;
}
template <size_t A, size_t B, size_t ROWS>
inline DSPVectorArray<B - A> separateRows(const DSPVectorArray<ROWS>& x)
{
  static_assert(B <= ROWS, "separateRows: range out of bounds");
  DSPVectorArray<B - A> y;
  for (size_t j = A; j < B; ++j) y.sig_[j - A] = x.sig_[j];
  return y;
}
template <size_t ROWS>
inline DSPVector addRows(const DSPVectorArray<ROWS>& x)  // vy = 0; vy = vy + row j
{
  DSPVector y{0.f};
  for (size_t j = 0; j < ROWS; ++j) y = add(y, x.constRow((int)j));
  return y;
}
template <size_t ROWS>
inline DSPVectorArray<ROWS> rowIndex()
{
  DSPVectorArray<ROWS> y;
  for (size_t j = 0; j < ROWS; ++j) y.row((int)j) = DSPVector((float)j);
  return y;
}
template <size_t ROWS>
inline DSPVectorArray<ROWS> columnIndex()
{
  return repeatRows<ROWS>(columnIndex());
}

// ---- routing (MLDSPRouting.h:59-137) ---------------------------------------------------------------------------
template <size_t ROWS, size_t INPUTS>
inline DSPVectorArray<ROWS> mix_n(size_t inputIndex, DSPVectorArray<INPUTS> gains, DSPVectorArray<ROWS> first)
{
  return first * repeatRows<ROWS>(gains.getRowVectorUnchecked(inputIndex));
}
template <size_t ROWS, size_t INPUTS, typename... Args>
inline DSPVectorArray<ROWS> mix_n(size_t inputIndex, DSPVectorArray<INPUTS> gains, DSPVectorArray<ROWS> first, Args... args)
{
  return first * repeatRows<ROWS>(gains.getRowVectorUnchecked(inputIndex)) + mix_n(inputIndex + 1, gains, args...);
}
template <size_t ROWS, size_t INPUTS, typename... Args>
inline DSPVectorArray<ROWS> mix(DSPVectorArray<INPUTS> gains, DSPVectorArray<ROWS> first, Args... args)
{
  return mix_n(0, gains, first, args...);
}
namespace gpu
{
template <size_t ROWS, typename... Args>
inline DSPVectorArray<ROWS> routeMux(int route, DSPVector selector, DSPVectorArray<ROWS> first, Args... args)
{
  const DSPVectorArray<ROWS> inputs[]{first, args...};
  constexpr int n = sizeof...(Args) + 1;
  static_assert(n <= MLGPU_ROUTE_MAX_SIGNALS, "multiplex: at most 8 inputs");
  if (!Capture::current())  // immediate mode: mlgpu_multiplex over the rows, one selector DSPVector for all of them
  {
    Eager& E = Eager::get();
    std::lock_guard<std::recursive_mutex> lock(E.m);
    const Engine& e = E.engine();
    const size_t rowFloats = 64 * ROWS;
    float* d = E.scratch(64 + (n + 1) * rowFloats);
    std::vector<float> host(64 + n * rowFloats);
    selector.sig_[0].hostCopy(host.data());
    for (int k = 0; k < n; ++k)
      for (size_t i = 0; i < rowFloats; ++i) host[64 + k * rowFloats + i] = inputs[k][i];
    e.check(mlgpu_upload(e.handle(), d, host.data(), host.size() * 4));
    const float* ins[MLGPU_ROUTE_MAX_SIGNALS];
    for (int k = 0; k < n; ++k) ins[k] = d + 64 + k * rowFloats;
    float* out = d + 64 + n * rowFloats;
    e.check(mlgpu_multiplex(e.handle(), d, 64, ins, n, out, rowFloats, route == MLGPU_ROUTE_MULTIPLEX_LINEAR ? 1 : 0));
    std::vector<float> res(rowFloats);
    e.check(mlgpu_download(e.handle(), res.data(), out, rowFloats * 4));
    return DSPVectorArray<ROWS>(static_cast<const float*>(res.data()));
  }
  Capture& c = Capture::get();
  DSPVectorArray<ROWS> y;
  for (size_t j = 0; j < ROWS; ++j)
  {
    int ids[1 + MLGPU_ROUTE_MAX_SIGNALS];
    ids[0] = selector.sig_[0].id();
    for (int k = 0; k < n; ++k) ids[1 + k] = inputs[k].sig_[j].id();
    y.sig_[j] = computedSig(c.ret(mlgpu_graph_add_route(c.g, route, ids, 1 + n, 0, 0, nullptr)));
  }
  return y;
}
template <size_t ROWS, typename... Args>
inline void routeDemux(int route, DSPVector selector, DSPVectorArray<ROWS> input, DSPVectorArray<ROWS>* firstOutput, Args... args)
{
  DSPVectorArray<ROWS>* outputs[]{firstOutput, args...};
  constexpr int n = sizeof...(Args) + 1;
  static_assert(n <= MLGPU_ROUTE_MAX_SIGNALS, "demultiplex: at most 8 outputs");
  if (!Capture::current())
  {
    Eager& E = Eager::get();
    std::lock_guard<std::recursive_mutex> lock(E.m);
    const Engine& e = E.engine();
    const size_t rowFloats = 64 * ROWS;
    float* d = E.scratch(64 + (n + 1) * rowFloats);
    std::vector<float> host(64 + rowFloats);
    selector.sig_[0].hostCopy(host.data());
    for (size_t i = 0; i < rowFloats; ++i) host[64 + i] = input[i];
    e.check(mlgpu_upload(e.handle(), d, host.data(), host.size() * 4));
    float* outs[MLGPU_ROUTE_MAX_SIGNALS];
    for (int k = 0; k < n; ++k) outs[k] = d + 64 + (k + 1) * rowFloats;
    e.check(mlgpu_demultiplex(e.handle(), d, 64, d + 64, outs, n, rowFloats, route == MLGPU_ROUTE_DEMULTIPLEX_LINEAR ? 1 : 0));
    std::vector<float> res(n * rowFloats);
    e.check(mlgpu_download(e.handle(), res.data(), outs[0], res.size() * 4));
    for (int k = 0; k < n; ++k) *outputs[k] = DSPVectorArray<ROWS>(static_cast<const float*>(res.data() + k * rowFloats));
    return;
  }
  Capture& c = Capture::get();
  for (int k = 0; k < n; ++k)
    for (size_t j = 0; j < ROWS; ++j)
    {
      const int ids[2] = {selector.sig_[0].id(), input.sig_[j].id()};
      outputs[k]->sig_[j] = computedSig(c.ret(mlgpu_graph_add_route(c.g, route, ids, 2, k, n, nullptr)));
    }
}
}  // namespace gpu
// MLDSPRouting.h:141-236: the input goes to the output the selector names (the others get 0) / is split between two neighbours
template <size_t ROWS, typename... Args>
inline void demultiplex(DSPVector selector, DSPVectorArray<ROWS> input, DSPVectorArray<ROWS>* firstOutput, Args... args)
{
  gpu::routeDemux(MLGPU_ROUTE_DEMULTIPLEX, selector, input, firstOutput, args...);
}
template <size_t ROWS, typename... Args>
inline void demultiplexLinear(DSPVector selector, DSPVectorArray<ROWS> input, DSPVectorArray<ROWS>* firstOutput, Args... args)
{
  gpu::routeDemux(MLGPU_ROUTE_DEMULTIPLEX_LINEAR, selector, input, firstOutput, args...);
}
template <size_t ROWS, typename... Args>
inline DSPVectorArray<ROWS> multiplex(DSPVector selector, DSPVectorArray<ROWS> first, Args... args)
{
  return gpu::routeMux(MLGPU_ROUTE_MULTIPLEX, selector, first, args...);
}
template <size_t ROWS, typename... Args>
inline DSPVectorArray<ROWS> multiplexLinear(DSPVector selector, DSPVectorArray<ROWS> first, Args... args)
{
  return gpu::routeMux(MLGPU_ROUTE_MULTIPLEX_LINEAR, selector, first, args...);
}

// ---- stateful objects ----------------------------------------------------------------------------------------------
namespace gpu
{
// A reference functor becomes ONE processor node the first time its operator() runs inside a capture. Its `coeffs`
// (a plain public member, as in the reference) are recorded as the node's initial, uniform coefficients; per-voice
// values are set afterwards with VoiceProgram::setCoeff(object, ...).
template <int KIND>
struct ProcNode
{
  int node_{-1};
  uint32_t nodeEpoch_{0};
  bool cleared_{false};
  uint32_t initState0_{0};
  bool hasInitState0_{false};
  float maxDelay_{-1.f};                            // delay lines: setMaxDelayInSamples
  std::vector<std::pair<int, uint32_t>> initState_;  // further state words set before the first call

  // immediate mode: the object's own one-voice graph (inputs -> this processor -> output), made at its first call outside a
  // capture. The processor's state lives in that graph between calls, as it lives in the reference object; a copy of the
  // object takes the state words along (a delay line's ring is not copied).
  struct Immediate
  {
    mlgpu_graph* g{nullptr};
    int node{-1}, nIn{0};
    float* d{nullptr};  // device: nIn inputs, then the output, 64 floats each
    ~Immediate()
    {
      std::lock_guard<std::recursive_mutex> lock(Eager::get().m);  // the immediate engine has one caller at a time
      if (g) mlgpu_graph_destroy(g);
      if (d) mlgpu_free(Eager::get().engine().handle(), d);
    }
  };
  std::shared_ptr<Immediate> imm_;
  // inputs that are one float per DSPVector (the reference's `float` arguments): control inputs of the one-voice graph
  static constexpr unsigned kControlInputs =
      (KIND == MLGPU_PROC_LINEAR_GLIDE || KIND == MLGPU_PROC_INTERPOLATOR1) ? 1u : (KIND == MLGPU_PROC_TEMPO_LOCK ? 6u : 0u);
  ProcNode() = default;
  ProcNode(const ProcNode& o) { copyFrom(o); }
  ProcNode& operator=(const ProcNode& o)
  {
    if (this != &o) copyFrom(o);
    return *this;
  }
  void copyFrom(const ProcNode& o)
  {
    node_ = o.node_; nodeEpoch_ = o.nodeEpoch_; cleared_ = o.cleared_; initState0_ = o.initState0_; hasInitState0_ = o.hasInitState0_;
    maxDelay_ = o.maxDelay_; initState_ = o.initState_;
    imm_.reset();
    if (o.imm_ && o.imm_->g)  // value semantics: the copy starts from the state the original has now
    {
      std::lock_guard<std::recursive_mutex> lock(Eager::get().m);
      const int ns = mlgpu_graph_num_state(o.imm_->g, o.imm_->node);
      for (int i = 0; i < ns; ++i)
      {
        uint32_t w = 0;
        Eager::get().engine().check(mlgpu_graph_get_state(o.imm_->g, o.imm_->node, i, &w));
        presetState(i, w);
      }
      cleared_ = hasInitState0_ = false;
    }
  }
  Sig emitImmediate(std::initializer_list<Sig> ins, const float* coeffs, int nc)
  {
    Eager& E = Eager::get();
    std::lock_guard<std::recursive_mutex> lock(E.m);
    const Engine& e = E.engine();
    const int n = (int)ins.size();
    if (!imm_ || imm_->nIn != n)
    {
      // another call form than last time (IntegerDelay(x) after IntegerDelay(x, delay)): the one-voice graph is built anew for
      // this arity, and starts from the state words the old one holds now - the reference object keeps its members across its
      // overloads. (A delay's ring memory belongs to the graph and starts empty again.)
      std::vector<std::pair<int, uint32_t>> carried;
      if (imm_ && imm_->g)
      {
        const int ns = mlgpu_graph_num_state(imm_->g, imm_->node);
        for (int i = 0; i < ns; ++i)
        {
          uint32_t w = 0;
          e.check(mlgpu_graph_get_state(imm_->g, imm_->node, i, &w));
          carried.push_back({i, w});
        }
      }
      auto im = std::make_shared<Immediate>();
      e.check(mlgpu_graph_create(e.handle(), 1, &im->g));
      auto ret = [&](int r) {
        if (r < 0) throw Error(-r, std::string("mlgpu_graph (immediate mode): ") + mlgpu_graph_last_error(im->g));
        return r;
      };
      int ids[8];
      // LinearGlide / Interpolator1 take one float per DSPVector (the reference's operator()(float)): a control input
      for (int k = 0; k < n; ++k)
        ids[k] = ret(((kControlInputs >> k) & 1u) ? mlgpu_graph_add_control(im->g, nullptr) : mlgpu_graph_add_input(im->g, nullptr));
      im->node = ret(mlgpu_graph_add_proc(im->g, KIND, ids, n, nullptr));
      if (maxDelay_ >= 0.f) e.check(mlgpu_graph_set_max_delay(im->g, im->node, maxDelay_));
      e.check(mlgpu_graph_add_output(im->g, im->node));
      e.check(mlgpu_graph_compile(im->g));
      void* p = nullptr;
      e.check(mlgpu_alloc(e.handle(), (size_t)(n + 1) * 256, &p));
      im->d = static_cast<float*>(p);
      im->nIn = n;
      for (auto& kv : carried) e.check(mlgpu_graph_set_state_uniform(im->g, im->node, kv.first, kv.second));
      imm_ = im;
    }
    Immediate& im = *imm_;
    for (int i = 0; i < nc; ++i) e.check(mlgpu_graph_set_coeff_uniform(im.g, im.node, i, coeffs[i]));
    // what was asked of the object since its last call, in the order a capture applies it: clear(), then state words
    if (cleared_) e.check(mlgpu_graph_clear_proc(im.g, im.node));
    if (hasInitState0_) e.check(mlgpu_graph_set_state_uniform(im.g, im.node, 0, initState0_));
    for (auto& kv : initState_) e.check(mlgpu_graph_set_state_uniform(im.g, im.node, kv.first, kv.second));
    cleared_ = hasInitState0_ = false;
    initState_.clear();
    float host[8 * 64];
    int k = 0, nAudio = 0, nCtl = 0;
    const float *inPtr[8], *ctlPtr[8];
    for (const Sig& sgn : ins)
    {
      sgn.hostCopy(host + 64 * k);
      if ((kControlInputs >> k) & 1u)
        ctlPtr[nCtl++] = im.d + 64 * k;  // a control's one float is the first of the 64 uploaded
      else
        inPtr[nAudio++] = im.d + 64 * k;
      ++k;
    }
    if (n) e.check(mlgpu_upload(e.handle(), im.d, host, (size_t)n * 256));
    float* outPtr[1] = {im.d + 64 * n};
    if (nCtl)
      e.check(mlgpu_graph_process_ctl(im.g, 1, nAudio ? inPtr : nullptr, MLGPU_LAYOUT_QUAD, ctlPtr, outPtr, MLGPU_LAYOUT_QUAD));
    else
      e.check(mlgpu_graph_process(im.g, 1, nAudio ? inPtr : nullptr, MLGPU_LAYOUT_QUAD, outPtr, MLGPU_LAYOUT_QUAD));
    auto out = std::make_shared<std::array<float, 64>>();
    e.check(mlgpu_download(e.handle(), out->data(), outPtr[0], 256));
    return Sig(std::shared_ptr<const std::array<float, 64>>(std::move(out)));
  }

  Sig emit(std::initializer_list<Sig> ins, const float* coeffs, int nc)
  {
    if (!Capture::current()) return emitImmediate(ins, coeffs, nc);
    Capture& c = Capture::get();
    if (node_ >= 0 && nodeEpoch_ == c.epoch)
      throw std::logic_error("mldsp GPU shim: a stateful object was called twice in one process function (one call = one state update)");
    int ids[8];
    int n = 0;
    for (const Sig& s : ins) ids[n++] = s.id();
    node_ = c.ret(mlgpu_graph_add_proc(c.g, KIND, ids, n, nullptr));
    nodeEpoch_ = c.epoch;
    if (maxDelay_ >= 0.f) c.eng->check(mlgpu_graph_set_max_delay(c.g, node_, maxDelay_));
    for (int i = 0; i < nc; ++i) c.deferCoeff(node_, i, coeffs[i]);
    if (cleared_) c.deferClear(node_);
    if (hasInitState0_) c.deferState(node_, 0, initState0_);
    for (auto& kv : initState_) c.deferState(node_, kv.first, kv.second);
    return computedSig(node_);
  }
  void presetState(int idx, uint32_t bits)
  {
    for (auto& kv : initState_)
      if (kv.first == idx)
      {
        kv.second = bits;
        return;
      }
    initState_.push_back({idx, bits});
  }
  int node() const { return node_; }
};
inline float bitsOf(int32_t i)
{
  float f;
  std::memcpy(&f, &i, 4);
  return f;
}
}  // namespace gpu

// generators, MLDSPGens.h
class TickGen : public gpu::ProcNode<MLGPU_PROC_TICK_GEN>
{
 public:
  void clear() { cleared_ = true; }
  DSPVector operator()(const DSPVector cyclesPerSample) { return DSPVector(emit({cyclesPerSample.sig_[0]}, nullptr, 0)); }
};
class ImpulseGen : public gpu::ProcNode<MLGPU_PROC_IMPULSE_GEN>
{
 public:
  void clear() { cleared_ = true; }
  DSPVector operator()(const DSPVector cyclesPerSample) { return DSPVector(emit({cyclesPerSample.sig_[0]}, nullptr, 0)); }
};
class NoiseGen : public gpu::ProcNode<MLGPU_PROC_NOISE_GEN>
{
 public:
  void reset() { seed(0); }
  void setSeed(uint32_t s) { seed(s); }  // MLDSPGens.h:115
  void seed(uint32_t s)  // per-voice seeds: VoiceProgram::setState(noise, 0, seeds)
  {
    initState0_ = s;
    hasInitState0_ = true;
  }
  DSPVector operator()() { return DSPVector(emit({}, nullptr, 0)); }
  // the scalar members (MLDSPGens.h:115-129): one step of the generator's own state word on the host - the object's seed as it
  // stands now (what a call outside a capture left in its one-voice graph, or what was set since)
  void step() { seed(currentSeed() * 0x0019660Du + 0x3C6EF35Fu); }
  uint32_t getIntSample()
  {
    step();
    return initState0_;
  }
  float getSample()
  {
    step();
    const uint32_t bits = ((initState0_ >> 9) & 0x007FFFFFu) | 0x3F800000u;
    float f;
    std::memcpy(&f, &bits, 4);
    return f * 2.f - 3.f;
  }

 private:
  uint32_t currentSeed()
  {
    if (hasInitState0_) return initState0_;
    if (cleared_ || !imm_ || !imm_->g) return 0;
    uint32_t w = 0;
    std::lock_guard<std::recursive_mutex> lock(gpu::Eager::get().m);
    gpu::Eager::get().engine().check(mlgpu_graph_get_state(imm_->g, imm_->node, 0, &w));
    return w;
  }
};
inline DSPVectorInt columnIndexInt()  // MLDSPOps.h: 0 .. 63 as integers (exact)
{
  DSPVectorInt y = truncateFloatToInt(columnIndex());
  auto t = std::make_shared<std::array<int32_t, 64>>();
  for (int i = 0; i < 64; ++i) (*t)[(size_t)i] = i;
  y.host_[0] = t;
  return y;
}

// phasorToSine / phasorToSaw / phasorToPulse (MLDSPGens.h:313-369, public free functions over a phasor): one stateless op
// node each. The device evaluates them with the oscillators' own per-lane code - the branch-free polyBLEP with its
// hoisted Newton-Raphson division (mldsp_procs.hpp) - instead of the twenty-odd elementwise nodes (two IEEE divisions per
// sample among them) the expression would expand to.
inline DSPVector phasorToSine(DSPVector phasor) { return DSPVector(gpu::opNode(MLGPU_OP_PHASOR_TO_SINE, {phasor.sig_[0]})); }
inline DSPVector phasorToSaw(DSPVector phasor, DSPVector freq) { return DSPVector(gpu::opNode(MLGPU_OP_PHASOR_TO_SAW, {phasor.sig_[0], freq.sig_[0]})); }
inline DSPVector phasorToPulse(DSPVector phasor, DSPVector freq, DSPVector pulseWidth)
{
  return DSPVector(gpu::opNode(MLGPU_OP_PHASOR_TO_PULSE, {phasor.sig_[0], freq.sig_[0], pulseWidth.sig_[0]}));
}

class PhasorGen : public gpu::ProcNode<MLGPU_PROC_PHASOR_GEN>
{
 public:
  void clear(uint32_t omega = 0)
  {
    initState0_ = omega;
    hasInitState0_ = true;
  }
  DSPVector operator()(const DSPVector cyclesPerSample) { return DSPVector(emit({cyclesPerSample.sig_[0]}, nullptr, 0)); }
};
class TestSineGen : public gpu::ProcNode<MLGPU_PROC_TEST_SINE_GEN>  // MLDSPGens.h:151-171
{
 public:
  void clear() { cleared_ = true; }
  DSPVector operator()(const DSPVector freq) { return DSPVector(emit({freq.sig_[0]}, nullptr, 0)); }
};
class OneShotGen : public gpu::ProcNode<MLGPU_PROC_ONE_SHOT_GEN>
{
 public:
  // MLDSPGens.h:229 - before the first call: every voice starts triggered. Afterwards: VoiceProgram::trigger(shot[, which voices])
  void trigger()
  {
    presetState(0, 0);
    presetState(1, 1);
    presetState(2, 0);
  }
  DSPVector operator()(const DSPVector cyclesPerSample) { return DSPVector(emit({cyclesPerSample.sig_[0]}, nullptr, 0)); }
};
class SineGen : public gpu::ProcNode<MLGPU_PROC_SINE_GEN>
{
 public:
  void clear() { cleared_ = true; }
  DSPVector operator()(const DSPVector freq) { return DSPVector(emit({freq.sig_[0]}, nullptr, 0)); }
};
class SawGen : public gpu::ProcNode<MLGPU_PROC_SAW_GEN>
{
 public:
  void clear() { cleared_ = true; }
  DSPVector operator()(const DSPVector freq) { return DSPVector(emit({freq.sig_[0]}, nullptr, 0)); }
};
class PulseGen : public gpu::ProcNode<MLGPU_PROC_PULSE_GEN>
{
 public:
  void clear() { cleared_ = true; }
  DSPVector operator()(const DSPVector freq, const DSPVector width)
  {
    const float w = 0.5f;
    return DSPVector(emit({freq.sig_[0], width.sig_[0]}, &w, 1));
  }
};
class SampleAccurateLinearGlide : public gpu::ProcNode<MLGPU_PROC_SAMPLE_ACCURATE_LINEAR_GLIDE>
{
  float c_[2]{gpu::bitsOf(32), 1.f / 32};

 public:
  void setGlideTimeInSamples(float t) { mlgpu_sample_accurate_linear_glide_make_coeffs(t, c_); }
  void clear() { cleared_ = true; }
  // the reference's nextSample(float) once per sample == one audio-rate input here
  DSPVector operator()(const DSPVector target) { return DSPVector(emit({target.sig_[0]}, c_, 2)); }
};
class LinearGlide : public gpu::ProcNode<MLGPU_PROC_LINEAR_GLIDE>
{
  float c_[2]{gpu::bitsOf(32), 1.f / 32};

 public:
  void setGlideTimeInSamples(float t) { mlgpu_linear_glide_make_coeffs(t, c_); }
  void clear() { cleared_ = true; }
  // `f` is one float per DSPVector: a constant here, or a gpu::VoiceParam / control converted to DSPVector
  DSPVector operator()(const DSPVector f) { return DSPVector(emit({f.sig_[0]}, c_, 2)); }
};
struct Interpolator1 : public gpu::ProcNode<MLGPU_PROC_INTERPOLATOR1>
{
  DSPVector operator()(const DSPVector f) { return DSPVector(emit({f.sig_[0]}, nullptr, 0)); }
};

// filters, MLDSPFilters.h
inline float dBToGain(float dB) { return mlgpu_db_to_gain(dB); }

template <size_t COEFFS_SIZE>  // MLDSPFilters.h:34-44
inline DSPVectorArray<COEFFS_SIZE> interpolateCoeffsLinear(const std::array<float, COEFFS_SIZE> c0, const std::array<float, COEFFS_SIZE> c1)
{
  DSPVectorArray<COEFFS_SIZE> vy;
  for (size_t i = 0; i < COEFFS_SIZE; ++i) vy.row((int)i) = interpolateDSPVectorLinear(c0[i], c1[i]);
  return vy;
}

struct Lopass : public gpu::ProcNode<MLGPU_PROC_LOPASS>
{
  enum coeffNames { g0, g1, g2, nCoeffs };
  typedef std::array<float, nCoeffs> Coeffs;
  Coeffs coeffs{};
  void clear() { cleared_ = true; }
  static Coeffs makeCoeffs(float omega, float k)
  {
    Coeffs c;
    mlgpu_lopass_make_coeffs(omega, k, c.data());
    return c;
  }
  DSPVector operator()(const DSPVector vx) { return DSPVector(emit({vx.sig_[0]}, coeffs.data(), 3)); }
  DSPVector operator()(const DSPVector vx, const DSPVector omega, const DSPVector k)
  {
    return DSPVector(emit({vx.sig_[0], omega.sig_[0], k.sig_[0]}, coeffs.data(), 3));
  }
};
class Hipass : public gpu::ProcNode<MLGPU_PROC_HIPASS>
{
 public:
  typedef std::array<float, 4> Coeffs;
  Coeffs coeffs{};
  void clear() { cleared_ = true; }
  static Coeffs makeCoeffs(float omega, float k)
  {
    Coeffs c;
    mlgpu_hipass_make_coeffs(omega, k, c.data());
    return c;
  }
  DSPVector operator()(const DSPVector vx) { return DSPVector(emit({vx.sig_[0]}, coeffs.data(), 4)); }
};
class Bandpass : public gpu::ProcNode<MLGPU_PROC_BANDPASS>
{
 public:
  typedef std::array<float, 3> Coeffs;
  Coeffs coeffs{};
  void clear() { cleared_ = true; }
  static Coeffs makeCoeffs(float omega, float k)
  {
    Coeffs c;
    mlgpu_bandpass_make_coeffs(omega, k, c.data());
    return c;
  }
  DSPVector operator()(const DSPVector vx) { return DSPVector(emit({vx.sig_[0]}, coeffs.data(), 3)); }
};
class LoShelf : public gpu::ProcNode<MLGPU_PROC_LO_SHELF>
{
 public:
  typedef std::array<float, 5> Coeffs;
  typedef DSPVectorArray<5> _vcoeffs;
  typedef std::array<float, 3> params;  // omega, k, A
  Coeffs coeffs{};
  void clear() { cleared_ = true; }
  static Coeffs makeCoeffs(params p)
  {
    Coeffs c;
    mlgpu_loshelf_make_coeffs(p[0], p[1], p[2], c.data());
    return c;
  }
  static _vcoeffs vcoeffs(const params p0, const params p1) { return interpolateCoeffsLinear(makeCoeffs(p0), makeCoeffs(p1)); }
  DSPVector operator()(const DSPVector vx) { return DSPVector(emit({vx.sig_[0]}, coeffs.data(), 5)); }
  DSPVector operator()(const DSPVector vx, const _vcoeffs vc)
  {
    return DSPVector(emit({vx.sig_[0], vc.sig_[0], vc.sig_[1], vc.sig_[2], vc.sig_[3], vc.sig_[4]}, coeffs.data(), 5));
  }
};
class HiShelf : public gpu::ProcNode<MLGPU_PROC_HI_SHELF>
{
 public:
  typedef std::array<float, 6> Coeffs;
  typedef DSPVectorArray<6> _vcoeffs;
  typedef std::array<float, 3> params;
  Coeffs coeffs{};
  void clear() { cleared_ = true; }
  static Coeffs makeCoeffs(params p)
  {
    Coeffs c;
    mlgpu_hishelf_make_coeffs(p[0], p[1], p[2], c.data());
    return c;
  }
  static _vcoeffs vcoeffs(const params p0, const params p1) { return interpolateCoeffsLinear(makeCoeffs(p0), makeCoeffs(p1)); }
  DSPVector operator()(const DSPVector vx) { return DSPVector(emit({vx.sig_[0]}, coeffs.data(), 6)); }
  DSPVector operator()(const DSPVector vx, const _vcoeffs vc)
  {
    return DSPVector(emit({vx.sig_[0], vc.sig_[0], vc.sig_[1], vc.sig_[2], vc.sig_[3], vc.sig_[4], vc.sig_[5]}, coeffs.data(), 6));
  }
};
class Bell : public gpu::ProcNode<MLGPU_PROC_BELL>
{
 public:
  typedef std::array<float, 4> Coeffs;
  Coeffs coeffs{};
  void clear() { cleared_ = true; }
  static Coeffs makeCoeffs(float omega, float k, float A)
  {
    Coeffs c;
    mlgpu_bell_make_coeffs(omega, k, A, c.data());
    return c;
  }
  DSPVector operator()(const DSPVector vx) { return DSPVector(emit({vx.sig_[0]}, coeffs.data(), 4)); }
};
struct OnePole : public gpu::ProcNode<MLGPU_PROC_ONE_POLE>
{
  typedef std::array<float, 2> Coeffs;
  Coeffs coeffs{};
  void clear() { cleared_ = true; }
  static Coeffs makeCoeffs(float omega)
  {
    Coeffs c;
    mlgpu_onepole_make_coeffs(omega, c.data());
    return c;
  }
  static Coeffs passthru() { return {1.f, 0.f}; }
  DSPVector operator()(const DSPVector vx) { return DSPVector(emit({vx.sig_[0]}, coeffs.data(), 2)); }
};
class DCBlocker : public gpu::ProcNode<MLGPU_PROC_DC_BLOCKER>
{
 public:
  typedef float Coeffs;
  Coeffs coeffs{0.045f};
  void clear() { cleared_ = true; }
  static Coeffs makeCoeffs(float omega) { return mlgpu_dcblocker_make_coeffs(omega); }
  DSPVector operator()(const DSPVector vx) { return DSPVector(emit({vx.sig_[0]}, &coeffs, 1)); }
};
class Differentiator : public gpu::ProcNode<MLGPU_PROC_DIFFERENTIATOR>
{
 public:
  void clear() { cleared_ = true; }
  DSPVector operator()(const DSPVector vx) { return DSPVector(emit({vx.sig_[0]}, nullptr, 0)); }
};
class Integrator : public gpu::ProcNode<MLGPU_PROC_INTEGRATOR>
{
 public:
  float mLeak{0.f};  // set leak to a value such as 0.001 for stability
  void clear() { cleared_ = true; }
  DSPVector operator()(const DSPVector vx) { return DSPVector(emit({vx.sig_[0]}, &mLeak, 1)); }
};
// Peak with exponential decay, MLDSPFilters.h:562-615. peakHoldSamples travels as the third coefficient word (an int).
class Peak : public gpu::ProcNode<MLGPU_PROC_PEAK>
{
 public:
  struct Coeffs
  {
    float a0, b1;
  };
  Coeffs coeffs{0.f, 0.f};
  int peakHoldSamples{44100};
  static Coeffs makeCoeffs(float omega)
  {
    float c[2];
    mlgpu_onepole_make_coeffs(omega, c);  // the same formula: x = expf(-omega * kTwoPi); {1 - x, x} (:575-579 == :458-464)
    return {c[0], c[1]};
  }
  static Coeffs passthru() { return {1.f, 0.f}; }
  DSPVector operator()(const DSPVector vx)
  {
    float c[3] = {coeffs.a0, coeffs.b1, 0.f};
    std::memcpy(&c[2], &peakHoldSamples, 4);
    return DSPVector(emit({vx.sig_[0]}, c, 3));
  }
};
// TempoLock, MLDSPFilters.h:1478-1579: x must be one of the streamed inputs (only x[0], x[1] of each vector are looked at)
class TempoLock : public gpu::ProcNode<MLGPU_PROC_TEMPO_LOCK>
{
 public:
  void clear() { cleared_ = true; }
  DSPVector operator()(DSPVector x, float dydx, float isr)
  {
    return DSPVector(emit({x.sig_[0], gpu::Sig(-1, dydx), gpu::Sig(-1, isr)}, nullptr, 0));
  }
};
class RMS : public gpu::ProcNode<MLGPU_PROC_RMS>
{
 public:
  typedef std::array<float, 2> Coeffs;
  Coeffs coeffs{};
  void clear() { cleared_ = true; }
  static Coeffs makeCoeffs(float omega) { return OnePole::makeCoeffs(omega); }
  DSPVector operator()(const DSPVector vx) { return DSPVector(emit({vx.sig_[0]}, coeffs.data(), 2)); }
};
struct ADSR : public gpu::ProcNode<MLGPU_PROC_ADSR>
{
  struct Coeffs
  {
    float ka, kd, s, kr;
  };
  Coeffs coeffs{0, 0, 0, 0};
  void clear() { cleared_ = true; }
  static Coeffs calcCoeffs(float a, float d, float s, float r, float sr)
  {
    float c[4];
    mlgpu_adsr_calc_coeffs(a, d, s, r, sr, c);
    return {c[0], c[1], c[2], c[3]};
  }
  DSPVector operator()(const DSPVector vx)
  {
    const float c[4] = {coeffs.ka, coeffs.kd, coeffs.s, coeffs.kr};
    return DSPVector(emit({vx.sig_[0]}, c, 4));
  }
};

// ---- delay lines, MLDSPFilters.h:799-1239 ------------------------------------------------------------------------------
namespace gpu
{
inline uint32_t bitsOfFloat(float f)
{
  uint32_t u;
  std::memcpy(&u, &f, 4);
  return u;
}
constexpr float kDefaultMaxDelay = 65536.f - 64.f;  // for the two reference classes that cannot size their own delays
}  // namespace gpu

class IntegerDelay : public gpu::ProcNode<MLGPU_PROC_INTEGER_DELAY>
{
 public:
  IntegerDelay() = default;
  IntegerDelay(int d)
  {
    setMaxDelayInSamples(static_cast<float>(d));
    setDelayInSamples(d);
  }
  void setDelayInSamples(int d) { presetState(1, (uint32_t)d); }
  void setMaxDelayInSamples(float d) { maxDelay_ = d; }
  void clear() { cleared_ = true; }
  DSPVector operator()(const DSPVector vx) { return DSPVector(emit({vx.sig_[0]}, nullptr, 0)); }
  DSPVector operator()(const DSPVector x, const DSPVector delay) { return DSPVector(emit({x.sig_[0], delay.sig_[0]}, nullptr, 0)); }
};

class Allpass1 : public gpu::ProcNode<MLGPU_PROC_ALLPASS1>
{
 public:
  typedef float Coeffs;
  Coeffs coeffs{0.f};
  Allpass1(float a) : coeffs(a) {}
  void clear() { cleared_ = true; }
  static float makeCoeffs(float d) { return mlgpu_allpass1_make_coeffs(d); }
  DSPVector operator()(const DSPVector vx) { return DSPVector(emit({vx.sig_[0]}, &coeffs, 1)); }
};

class FractionalDelay : public gpu::ProcNode<MLGPU_PROC_FRACTIONAL_DELAY>
{
 public:
  FractionalDelay() = default;
  FractionalDelay(float d)
  {
    setMaxDelayInSamples(d);
    setDelayInSamples(d);
  }
  void clear() { cleared_ = true; }
  void setDelayInSamples(float d)
  {
    float st[2];
    mlgpu_fractional_delay_make_state(d, st);
    presetState(3, gpu::bitsOfFloat(st[0]));
    presetState(4, gpu::bitsOfFloat(st[1]));
  }
  void setMaxDelayInSamples(float d) { maxDelay_ = floorf(d); }
  DSPVector operator()(const DSPVector vx) { return DSPVector(emit({vx.sig_[0]}, nullptr, 0)); }
  DSPVector operator()(const DSPVector vx, const DSPVector vDelayInSamples) { return DSPVector(emit({vx.sig_[0], vDelayInSamples.sig_[0]}, nullptr, 0)); }
  DSPVector operator()(const DSPVector vx, const DSPVector vDelayInSamples, const DSPVectorInt vChangeTicks)
  {
    return DSPVector(emit({vx.sig_[0], vDelayInSamples.sig_[0], vChangeTicks.sig_[0]}, nullptr, 0));
  }
};

class PitchbendableDelay : public gpu::ProcNode<MLGPU_PROC_PITCHBENDABLE_DELAY>
{
 public:
  PitchbendableDelay() = default;
  void setMaxDelayInSamples(float d) { maxDelay_ = floorf(d); }
  void clear() { cleared_ = true; }
  DSPVector operator()(const DSPVector vInput, const DSPVector vDelayInSamples)
  {
    if (maxDelay_ < 0.f) maxDelay_ = gpu::kDefaultMaxDelay;
    return DSPVector(emit({vInput.sig_[0], vDelayInSamples.sig_[0]}, nullptr, 0));
  }
};

// ---- composites around a delay line with one DSPVector of loop latency -------------------------------------------------
// Allpass<>, FDN<> and FeedbackDelayFunction(WithTap) (MLDSPFilters.h:1110-1239, MLDSPFunctional.h:262-316) all close a loop
// through "what the delay line returned for the previous DSPVector". In the engine that is a feedback node of the graph
// (64 state words per voice, read and rewritten in place); gpu::FeedbackLoop is that node as an object: value() is last
// vector's signal, close(next) names the signal to keep for the next one. The classes below are wiring around it - the
// arithmetic is the published structure of each filter (Schroeder allpass, Householder feedback matrix), evaluated in the
// reference's operation order because the outputs are compared bit for bit.
namespace gpu
{
class FeedbackLoop
{
  int node_{-1};
  DSPVector* kept_{nullptr};  // immediate mode: the owner's DSPVector that carries the loop from one call to the next

 public:
  FeedbackLoop() : node_(Capture::get().ret(mlgpu_graph_add_feedback(Capture::get().g, nullptr))) {}
  // `kept` is a member of the object that owns the loop: in immediate mode it IS the loop's memory (the value one call leaves for
  // the next, zero at first), inside a capture it is not touched
  explicit FeedbackLoop(DSPVector& kept) : kept_(&kept)
  {
    if (Capture::current()) node_ = Capture::get().ret(mlgpu_graph_add_feedback(Capture::get().g, nullptr));
  }
  DSPVector value() const
  {
    if (!Capture::current())
    {
      if (!kept_) throw std::logic_error("mldsp GPU shim: a feedback loop outside a VoiceProgram capture needs its owner's storage");
      return *kept_;
    }
    return DSPVector(Sig(node_, 0.f));
  }
  void close(const DSPVector& next)
  {
    if (!Capture::current())
    {
      *kept_ = next;
      return;
    }
    Capture& c = Capture::get();
    const int st = mlgpu_graph_set_feedback(c.g, node_, next.sig_[0].id());
    if (st != MLGPU_OK) throw Error(st, std::string("mlgpu_graph_set_feedback: ") + mlgpu_last_error(c.eng->handle()));
  }
};
}  // namespace gpu

// Allpass<DELAY_TYPE>: Schroeder allpass, gain g = mGain, around DELAY_TYPE.   w = x + g d ;  y = d - g w ;  d' = line(w)
// (with -g as the multiplier, as the reference has it). The line is shorter than the nominal delay by the loop's one
// DSPVector of latency.
template <typename DELAY_TYPE>
class Allpass
{
  DELAY_TYPE line_;
  DSPVector kept_;  // immediate mode: what the line gave at the last call (the reference's vy1)
  template <class THROUGH_LINE>
  DSPVector run(const DSPVector& x, THROUGH_LINE throughLine)
  {
    gpu::FeedbackLoop loop(kept_);
    const DSPVector d = loop.value(), minusG(-mGain);
    const DSPVector w = x - d * minusG;
    const DSPVector y = w * minusG + d;
    loop.close(throughLine(w));
    return y;
  }

 public:
  float mGain{0.f};
  void setDelayInSamples(float d) { line_.setDelayInSamples(d - kFloatsPerDSPVector); }
  void setMaxDelayInSamples(float d) { line_.setMaxDelayInSamples(d - kFloatsPerDSPVector); }
  void clear()  // the loop's kept DSPVector starts at zero like the line
  {
    line_.clear();
    kept_ = DSPVector();
  }
  DSPVector operator()(const DSPVector x)
  {
    return run(x, [this](const DSPVector& w) { return line_(w); });
  }
  DSPVector operator()(const DSPVector x, const DSPVector delayInSamples)
  {
    return run(x, [&](const DSPVector& w) { return line_(w, delayInSamples - DSPVector((float)kFloatsPerDSPVector)); });
  }
};

// FDN<SIZE>: SIZE delay lines, each followed by a OnePole and a gain, mixed back through the Householder reflection
// v - (2 / SIZE) sum(v); even lines feed the right output, odd lines the left (whole pairs only). (The reference class
// never allocates its IntegerDelays; here each gets a ring that holds the delay time it is given.)
template <int SIZE>
class FDN
{
  std::array<IntegerDelay, SIZE> lines_;
  std::array<OnePole, SIZE> damping_;
  std::array<DSPVector, SIZE> kept_;  // immediate mode: the loops' memories

 public:
  std::array<float, SIZE> mFeedbackGains{{0}};
  void setDelaysInSamples(std::array<float, SIZE> times)
  {
    for (int n = 0; n < SIZE; ++n)
    {
      const int len = std::max(1, (int)(times[n] - kFloatsPerDSPVector));  // the loop itself is one DSPVector long
      lines_[n].setMaxDelayInSamples((float)len);
      lines_[n].setDelayInSamples(len);
    }
  }
  void setFilterCutoffs(std::array<float, SIZE> omegas)
  {
    for (int n = 0; n < SIZE; ++n) damping_[n].coeffs = OnePole::makeCoeffs(omegas[n]);
  }
  DSPVectorArray<2> operator()(const DSPVector x)
  {
    std::vector<gpu::FeedbackLoop> loops;  // what goes into line n on the next DSPVector
    loops.reserve(SIZE);
    for (int n = 0; n < SIZE; ++n) loops.emplace_back(kept_[n]);
    std::array<DSPVector, SIZE> taps;
    for (int n = 0; n < SIZE; ++n) taps[n] = lines_[n](loops[n].value());
    DSPVector left, right, total;  // each accumulates from a zero vector, in line order
    for (int n = 0; n < (SIZE & ~1); ++n) (n & 1 ? left : right) += taps[n];
    for (int n = 0; n < SIZE; ++n) total += taps[n];
    total *= DSPVector(2.0f / SIZE);
    for (int n = 0; n < SIZE; ++n) loops[n].close(damping_[n](taps[n] - total) * DSPVector(mFeedbackGains[n]) + x);
    return concatRows(left, right);
  }
};

// FeedbackDelayFunction(WithTap): y = fn(x + feedbackGain * d) ;  d' = PitchbendableDelay(y, time - one DSPVector)
class FeedbackDelayFunction
{
  using ProcessFn = std::function<DSPVector(const DSPVector)>;
  PitchbendableDelay line_;
  DSPVector kept_;

 public:
  float feedbackGain{1.f};
  void setMaxDelayInSamples(float d) { line_.setMaxDelayInSamples(d); }  // extension: the reference offers no way to size it
  DSPVector operator()(const DSPVector x, ProcessFn fn, const DSPVector delayTime)
  {
    gpu::FeedbackLoop loop(kept_);
    const DSPVector y = fn(x + loop.value() * DSPVector(feedbackGain));
    loop.close(line_(y, delayTime - DSPVector((float)kFloatsPerDSPVector)));
    return y;
  }
};
class FeedbackDelayFunctionWithTap  // fn returns what is fed back and hands the listener's signal out through its second argument
{
  using ProcessFn = std::function<DSPVector(const DSPVector, DSPVector&)>;
  PitchbendableDelay line_;
  DSPVector kept_;

 public:
  float feedbackGain{1.f};
  void setMaxDelayInSamples(float d) { line_.setMaxDelayInSamples(d); }
  DSPVector operator()(const DSPVector x, ProcessFn fn, const DSPVector delayTime)
  {
    gpu::FeedbackLoop loop(kept_);
    DSPVector tap;
    const DSPVector fedBack = fn(x + loop.value() * DSPVector(feedbackGain), tap);
    loop.close(line_(fedBack, delayTime - DSPVector((float)kFloatsPerDSPVector)));
    return tap;
  }
};

// map, MLDSPFunctional.h:18-100. The row-wise forms apply f to each row (a captured sub-graph per row). The element-wise
// forms take a scalar HOST function evaluated per sample: that cannot run on the device, so they refuse at capture time.
template <size_t ROWS>
inline DSPVectorArray<ROWS> map(std::function<DSPVector(const DSPVector)> f, const DSPVectorArray<ROWS> x)
{
  DSPVectorArray<ROWS> y;
  for (size_t j = 0; j < ROWS; ++j) y.row((int)j) = f(x.constRow((int)j));
  return y;
}
template <size_t ROWS>
inline DSPVectorArray<ROWS> map(std::function<DSPVector(const DSPVector, int)> f, const DSPVectorArray<ROWS> x)
{
  DSPVectorArray<ROWS> y;
  for (size_t j = 0; j < ROWS; ++j) y.row((int)j) = f(x.constRow((int)j), (int)j);
  return y;
}
template <size_t ROWS>
inline DSPVectorArray<ROWS> map(std::function<DSPVector(const DSPVector, const DSPVector)> f, const DSPVectorArray<ROWS> x)
{
  DSPVectorArray<ROWS> y;
  for (size_t j = 0; j < ROWS; ++j) y.row((int)j) = f(x.constRow((int)j), DSPVector((float)j));  // f(row, j): j converts to DSPVector(float), :95
  return y;
}
// map(float()) and map(float(int)) (MLDSPFunctional.h:24-35, 50-60) do not depend on device data: the host evaluates the
// function - in the reference's element order, so a stateful f() (a counter, a random source) gives the same numbers - and the
// 64 * ROWS results go into the kernel as constant tables, like DSPVector(const float*). What "the same numbers" cannot
// cover: the reference calls f once per process call, a captured program once per capture (VoiceProgram::update() repeats it).
template <size_t ROWS>
inline DSPVectorArray<ROWS> map(std::function<float()> f, const DSPVectorArray<ROWS>)
{
  std::array<float, 64 * ROWS> a;
  for (size_t n = 0; n < 64 * ROWS; ++n) a[n] = f();
  return DSPVectorArray<ROWS>(a.data());
}
// the integer argument must be host data too: columnIndexInt() / a literal DSPVectorInt (the shim's DSPVectorArrayInt of a
// computed signal has no host values)
template <size_t ROWS>
inline DSPVectorArray<ROWS> map(std::function<float(int)> f, const DSPVectorArrayInt<ROWS> x)
{
  std::array<float, 64 * ROWS> a;
  for (size_t n = 0; n < 64 * ROWS; ++n) a[n] = f(x.hostInt((int)n));
  return DSPVectorArray<ROWS>(a.data());
}
// map(float(float)) (MLDSPFunctional.h:37-48): the caller's own scalar function over the samples. On host data (immediate mode,
// tables) that is what it says - f is the user's code, not the library's arithmetic; over a signal a captured kernel computes
// there are no samples to hand to a host function.
template <size_t ROWS>
inline DSPVectorArray<ROWS> map(std::function<float(float)> f, const DSPVectorArray<ROWS> x)
{
  for (size_t j = 0; j < ROWS; ++j)
    if (x.sig_[j].node >= 0)
      throw std::logic_error("mldsp GPU shim: map() with a scalar host function over a signal the kernel computes - its samples exist only on the "
                             "device; write the step with DSPVector ops");
  std::array<float, 64 * ROWS> a;
  for (size_t n = 0; n < 64 * ROWS; ++n) a[n] = f(x[n]);
  return DSPVectorArray<ROWS>(a.data());
}

// Upsampler / Downsampler (MLDSPFilters.h:1316-1473): the vector-scheduled classes - write a DSPVector, read 2^octaves of them
// (resp. write 2^octaves, read one). They hand whole DSPVectors back and forth on the caller's schedule, so they exist in
// immediate mode only: a one-voice mlgpu_resampler per object (the same HalfBandFilter cascade; a Downsampler runs it when the
// 2^octaves-th vector has arrived - the filters see the same samples in the same order as in the reference's per-write schedule).
// Inside a captured process function use Upsample2xFunction / Downsample2xFunction, or mlgpu_resampler on device signals.
namespace gpu
{
class ImmediateResampler
{
  mlgpu_resampler* r_{nullptr};
  float* d_{nullptr};
  int octaves_, up_;

 public:
  ImmediateResampler(int octaves, bool up) : octaves_(octaves), up_(up ? 1 : 0)
  {
    if (octaves < 0 || octaves > 6) throw std::invalid_argument("mldsp GPU shim: Upsampler / Downsampler octaves 0..6");
  }
  ImmediateResampler(const ImmediateResampler& o) : octaves_(o.octaves_), up_(o.up_) { takeStateOf(o); }
  ImmediateResampler& operator=(const ImmediateResampler& o)
  {
    if (this != &o)
    {
      release();
      octaves_ = o.octaves_;
      up_ = o.up_;
      takeStateOf(o);
    }
    return *this;
  }
  ~ImmediateResampler() { release(); }
  void release()
  {
    if (!r_ && !d_) return;
    std::lock_guard<std::recursive_mutex> lock(Eager::get().m);
    if (r_) mlgpu_resampler_destroy(r_);
    if (d_) mlgpu_free(Eager::get().engine().handle(), d_);
    r_ = nullptr;
    d_ = nullptr;
  }
  void takeStateOf(const ImmediateResampler& o)  // value semantics: the copy continues from the original's filter memories
  {
    if (!o.r_ || !octaves_) return;
    std::lock_guard<std::recursive_mutex> lock(Eager::get().m);
    std::vector<float> st((size_t)octaves_ * 9);
    Eager::get().engine().check(mlgpu_resampler_get_state(o.r_, st.data()));
    make();
    Eager::get().engine().check(mlgpu_resampler_set_state(r_, st.data()));
  }
  void make()
  {
    const Engine& e = Eager::get().engine();
    e.check(mlgpu_resampler_create(e.handle(), 1, octaves_, up_, &r_));
    void* p = nullptr;
    e.check(mlgpu_alloc(e.handle(), (size_t)((1 << octaves_) + 1) * 256, &p));
    d_ = static_cast<float*>(p);
  }
  void clear()
  {
    if (r_) Eager::get().engine().check(mlgpu_resampler_clear(r_));
  }
  // nIn DSPVectors in, nIn << octaves (up) or nIn >> octaves (down) out
  void run(const float* in, size_t nIn, float* out)
  {
    if (Capture::current())
      throw std::logic_error("mldsp GPU shim: Upsampler / Downsampler write() / read() run on the caller's schedule, outside a capture; inside a "
                             "process function use Upsample2xFunction / Downsample2xFunction");
    std::lock_guard<std::recursive_mutex> lock(Eager::get().m);
    const Engine& e = Eager::get().engine();
    if (!r_) make();
    const size_t nOut = up_ ? nIn << octaves_ : nIn >> octaves_;
    float* dIn = up_ ? d_ + 64 * ((size_t)1 << octaves_) : d_;
    float* dOut = up_ ? d_ : d_ + 64 * ((size_t)1 << octaves_);
    e.check(mlgpu_upload(e.handle(), dIn, in, nIn * 256));
    e.check(mlgpu_resampler_process(r_, nIn, dIn, MLGPU_LAYOUT_QUAD, dOut, MLGPU_LAYOUT_QUAD));
    e.check(mlgpu_download(e.handle(), out, dOut, nOut * 256));
  }
};
}  // namespace gpu


// HalfBandFilter (MLDSPFilters.h:1245-1310) as an object of its own, immediate mode: one octave of that cascade per direction.
// upsampleSecondHalf(x) hands out the second half of what upsampleFirstHalf(x) computed (the reference's users call them in
// that order on the same vector); the upsampling and the downsampling memories are separate, as in every use the reference makes.
class HalfBandFilter
{
  gpu::ImmediateResampler up_{1, true}, down_{1, false};
  std::array<float, 128> upOut_{};

 public:
  DSPVector upsampleFirstHalf(const DSPVector vx)
  {
    float in[64];
    store(vx, in);
    up_.run(in, 1, upOut_.data());
    return DSPVector(static_cast<const float*>(upOut_.data()));
  }
  DSPVector upsampleSecondHalf(const DSPVector) { return DSPVector(static_cast<const float*>(upOut_.data() + 64)); }
  DSPVector downsample(const DSPVector vx1, const DSPVector vx2)
  {
    float in[128], out[64];
    store(vx1, in);
    store(vx2, in + 64);
    down_.run(in, 2, out);
    return DSPVector(static_cast<const float*>(out));
  }
  void clear()
  {
    up_.clear();
    down_.clear();
    upOut_.fill(0.f);
  }
};

// Upsample2xFunction / Downsample2xFunction, MLDSPFunctional.h:114-213: fn runs at twice / half the rate between two
// HalfBandFilters. Captured as a rate region of the graph (mlgpu_graph_begin_region): fn is called ONCE here, on the
// resampled inputs, and the kernel evaluates its nodes twice per sample (resp. every second sample) on the same
// processor objects - what the reference does by calling the one stateful fn twice per DSPVector (resp. once per two).
namespace gpu
{
template <size_t IN_ROWS, class FN>
inline DSPVectorArray<1> rateRegion(int kind, FN& fn, const DSPVectorArray<IN_ROWS>& vx)
{
  Capture& c = Capture::get();
  int ins[IN_ROWS ? IN_ROWS : 1], inner[IN_ROWS ? IN_ROWS : 1];
  for (size_t j = 0; j < IN_ROWS; ++j) ins[j] = vx.sig_[j].id();
  const int st = mlgpu_graph_begin_region(c.g, kind, ins, (int)IN_ROWS, inner);
  if (st != MLGPU_OK) throw Error(st, std::string("mlgpu_graph_begin_region: ") + mlgpu_last_error(c.eng->handle()));
  DSPVectorArray<IN_ROWS> resampled;
  for (size_t j = 0; j < IN_ROWS; ++j) resampled.sig_[j] = computedSig(inner[j]);
  const int outerRegion = c.curRegion;
  c.curRegion = ++c.regionCounter;
  const DSPVectorArray<1> y = fn(resampled);
  const int result = y.sig_[0].id();
  c.curRegion = outerRegion;
  return DSPVectorArray<1>(computedSig(c.ret(mlgpu_graph_end_region(c.g, result, nullptr))));
}
}  // namespace gpu

template <int IN_ROWS>
class Upsample2xFunction
{
  using inputType = const DSPVectorArray<IN_ROWS>;
  using outputType = DSPVectorArray<1>;
  using ProcessFn = std::function<outputType(inputType)>;
  // immediate mode: the reference's own schedule (MLDSPFunctional.h:128-151) - both halves of every input row, fn twice, one
  // DSPVector down - over HalfBandFilters on the device
  std::array<HalfBandFilter, (IN_ROWS > 0 ? IN_ROWS : 1)> uppers_;
  HalfBandFilter downer_;

 public:
  outputType operator()(ProcessFn fn, inputType vx)
  {
    if (gpu::Capture::current()) return gpu::rateRegion<IN_ROWS>(MLGPU_REGION_UPSAMPLE_2X, fn, vx);
    DSPVectorArray<IN_ROWS> in1, in2;
    for (int j = 0; j < IN_ROWS; ++j)
    {
      in1.row(j) = uppers_[(size_t)j].upsampleFirstHalf(vx.constRow(j));
      in2.row(j) = uppers_[(size_t)j].upsampleSecondHalf(vx.constRow(j));
    }
    const outputType out1 = fn(in1);
    const outputType out2 = fn(in2);
    return downer_.downsample(out1, out2);
  }
};

template <int IN_ROWS>
class Downsample2xFunction
{
  using inputType = const DSPVectorArray<IN_ROWS>;
  using outputType = DSPVectorArray<1>;
  using ProcessFn = std::function<outputType(inputType)>;

 public:
  outputType operator()(ProcessFn fn, const DSPVectorArray<IN_ROWS> vx = DSPVectorArray<0>())
  {
    if (gpu::Capture::current()) return gpu::rateRegion<IN_ROWS>(MLGPU_REGION_DOWNSAMPLE_2X, fn, vx);
    // immediate mode: the reference's two-phase schedule (MLDSPFunctional.h:176-211)
    outputType vy;
    if (phase_)
    {
      DSPVectorArray<IN_ROWS> down;
      for (int j = 0; j < IN_ROWS; ++j) down.row(j) = downers_[(size_t)j].downsample(inputBuffer_.constRow(j), vx.constRow(j));
      const outputType y = fn(down);
      vy = upper_.upsampleFirstHalf(y);
      outputBuffer_ = upper_.upsampleSecondHalf(y);
    }
    else
    {
      inputBuffer_ = vx;
      vy = outputBuffer_;
    }
    phase_ = !phase_;
    return vy;
  }

 private:
  std::array<HalfBandFilter, (IN_ROWS > 0 ? IN_ROWS : 1)> downers_;
  HalfBandFilter upper_;
  DSPVectorArray<IN_ROWS> inputBuffer_;
  outputType outputBuffer_;
  bool phase_{false};
};

struct Upsampler
{
  gpu::ImmediateResampler r_;
  std::vector<float> out_;
  int octaves_, readIdx_{0};
  explicit Upsampler(int octavesUp) : r_(octavesUp, true), out_((size_t)64 << octavesUp, 0.f), octaves_(octavesUp) {}
  void write(DSPVector x)
  {
    float in[64];
    store(x, in);
    r_.run(in, 1, out_.data());
    readIdx_ = 0;
  }
  DSPVector read()  // after a write, 1 << octaves reads are available
  {
    const DSPVector y(static_cast<const float*>(out_.data() + 64 * (size_t)readIdx_));
    ++readIdx_;
    return y;
  }
  void clear()
  {
    r_.clear();
    std::fill(out_.begin(), out_.end(), 0.f);
    readIdx_ = 0;
  }
};

class Downsampler
{
  gpu::ImmediateResampler r_;
  std::vector<float> in_;
  std::array<float, 64> out_{};
  int octaves_;
  uint32_t counter_{0};

 public:
  explicit Downsampler(int octavesDown) : r_(octavesDown, false), in_((size_t)64 << octavesDown, 0.f), octaves_(octavesDown) {}
  // true when this write completed an output vector (every 2^octaves writes)
  bool write(DSPVector v)
  {
    store(v, in_.data() + 64 * (size_t)counter_);
    counter_ = (counter_ + 1) & ((1u << octaves_) - 1u);
    if (counter_ != 0) return false;
    r_.run(in_.data(), (size_t)1 << octaves_, out_.data());
    return true;
  }
  DSPVector read() { return DSPVector(static_cast<const float*>(out_.data())); }
  void clear()
  {
    r_.clear();
    out_.fill(0.f);
    counter_ = 0;
  }
};

// ml::Sample (MLDSPSample.h): a host container of audio - channels, a sample rate, the frames interleaved in one float vector -
// and its free functions. Host data plumbing, the same interface.
struct Sample
{
  size_t channels{0};
  size_t sampleRate{0};
  std::vector<float> sampleData;
  float operator[](size_t i) const { return sampleData[i]; }
  float& operator[](size_t i) { return sampleData[i]; }
};
inline size_t getSize(const Sample& s) { return s.sampleData.size(); }
inline size_t getFrames(const Sample& s) { return s.channels ? s.sampleData.size() / s.channels : 0; }
inline const float* getConstFramePtr(const Sample& s, size_t frameIdx = 0) { return s.sampleData.data() + frameIdx * s.channels; }
inline float* getFramePtr(Sample& s, size_t frameIdx = 0) { return s.sampleData.data() + frameIdx * s.channels; }
inline float getRate(const Sample& s) { return (float)s.sampleRate; }
inline float getDuration(const Sample& s) { return s.sampleRate ? getFrames(s) / (float)s.sampleRate : 0.f; }
inline bool usable(const Sample* pSample) { return pSample && !pSample->sampleData.empty(); }
inline float* resize(Sample& s, size_t newFrames, size_t newChans = 1)  // nullptr when the allocation fails
{
  try
  {
    s.sampleData.resize(newFrames * newChans);
  }
  catch (const std::exception&)
  {
    return nullptr;
  }
  s.channels = newChans;
  return s.sampleData.data();
}
inline float findMaximumValue(const Sample& x) { return *std::max_element(x.sampleData.begin(), x.sampleData.end()); }
inline void normalize(Sample& x)
{
  if (x.sampleData.empty()) return;
  const float ratio = 1.0f / findMaximumValue(x);
  for (float& f : x.sampleData) f *= ratio;
}
inline void clear(Sample& x) { x.sampleData.clear(); }

// Bank<T, ROWS>, MLDSPFunctional.h:321-360
template <typename T, size_t ROWS>
class Bank
{
  std::array<T, ROWS> _processors;

 public:
  template <typename... Args>
  DSPVectorArray<ROWS> operator()(Args... args)
  {
    DSPVectorArray<ROWS> output;
    for (size_t i = 0; i < ROWS; ++i) output.row((int)i) = _processors[i](args.constRow((int)i)...);
    return output;
  }
  // each processor gets its arguments by subscripting the inputs (std::array / std::vector of DSPVectors), :341-350
  template <typename... Args>
  DSPVectorArray<ROWS> processArrays(Args... args)
  {
    DSPVectorArray<ROWS> output;
    for (size_t i = 0; i < ROWS; ++i) output.row((int)i) = _processors[i](args[i]...);
    return output;
  }
  void clear()
  {
    for (auto& p : _processors) p.clear();
  }
  T& operator[](size_t n) { return _processors[n]; }
};

// ---- the process-function boundary (source/app/MLAudioContext.h:60-110, MLSignalProcessBuffer.h:18) -----------------

// rows per voice output signal, source/app/MLEventsToSignals.h:15-26
enum VoiceOutputSignals
{
  kPitch = 0,
  kGate,
  kVoice,
  kZ,
  kX,
  kY,
  kMod,
  kElapsedTime,
  kNumVoiceOutputRows
};

enum EventType  // source/app/MLEvent.h:13-26
{
  kNull = 0,
  kNoteOn,
  kNoteRetrig,
  kNoteSustain,
  kNoteOff,
  kSustainPedal,
  kController,
  kPitchBend,
  kNotePressure,
  kChannelPressure,
  kProgramChange,
  kNumEventTypes
};
struct Event  // MLEvent.h:31-53; same layout as mlgpu_event
{
  uint8_t type{kNull};
  uint8_t channel{0};
  uint16_t sourceIdx{0};
  int time{0};
  float value1{0};
  float value2{0};
  explicit operator bool() const { return type != kNull; }
};

#ifndef MLGPU_COMPAT_HAS_MADRONALIB_APP_HEADERS
// the little of ml::Symbol that EventsToSignals::setProtocol needs when madronalib's own app headers are not on the include path
class Symbol
{
  std::string text_;

 public:
  Symbol() = default;
  Symbol(const char* s) : text_(s ? s : "") {}
  bool operator==(const Symbol& o) const { return text_ == o.text_; }
  const std::string& text() const { return text_; }
};
#endif

// EventsToSignals (source/app/MLEventsToSignals.h:44-181). Inside a capture only its Voice type matters (the rows a captured
// processVoice reads are graph inputs fed by mlgpu_events for every instrument at once - gpu::SynthProgram). As an OBJECT - made,
// configured, fed events and stepped a DSPVector at a time, the way the reference's AudioContext drives it - it works in immediate
// mode: a one-instrument mlgpu_events on the immediate engine, its eight rows fetched after every processVector(). The smoothed
// controllers (getController(n).output) come from helper objects, one per number asked for, each of which has been given the
// object's whole history of calls when it was made - so a number that is first asked for late still has the signal the
// reference's always-running smoother has.
class EventsToSignals
{
 public:
  static constexpr size_t kMaxVoices{16};
  static constexpr size_t kNumControllers{129};
  static constexpr int kChannelPressureControllerIdx{128};
  struct Voice  // what a process function reads of EventsToSignals::Voice (MLEventsToSignals.h:102-168): its output rows
  {
    DSPVectorArray<kNumVoiceOutputRows> outputs;
  };
  struct SmoothedController  // :170-180: what getController(n) hands out
  {
    DSPVector output;
  };

  EventsToSignals()
  {
    if (const char* s = std::getenv("MLGPU_E2S_HISTORY_OPS")) historyLimit_ = (size_t)std::strtoull(s, nullptr, 10);
  }
  EventsToSignals(const EventsToSignals&) = delete;
  EventsToSignals& operator=(const EventsToSignals&) = delete;
  ~EventsToSignals()
  {
    main_.release();
    for (auto& h : helpers_) h.release();
  }
  void setSampleRate(double r) { apply(Op{kSampleRate, r}); }
  // setPolyphony (:322-327) = clear() + the new voice count. An mlgpu_events has its polyphony from creation: before the first
  // processVector the object is simply made for n voices; afterwards the same n is clear(), and another n is refused (the
  // reference's voices keep their pitch glides and drift phases through that change, a new object would not).
  size_t setPolyphony(size_t n)
  {
    n = n < 1 ? 1 : (n > kMaxVoices ? kMaxVoices : n);
    if (main_.processed && n != polyphony_)
      throw std::logic_error("mldsp GPU shim: EventsToSignals::setPolyphony to another voice count after processVector() is not supported "
                             "in immediate mode - set the polyphony before processing");
    polyphony_ = n;
    apply(Op{kPolyphony, (double)n});
    return polyphony_;
  }
  size_t getPolyphony() { return polyphony_; }
  void clear() { apply(Op{kClear, 0.}); }
  void addEvent(const Event& e)
  {
    Op o{kAddEvent, 0.};
    o.e = e;
    apply(o);
  }
  void clearEvents() { apply(Op{kClearEvents, 0.}); }
  void setPitchBendInSemitones(float f) { apply(Op{kBend, (double)f}); }
  void setMPEPitchBendInSemitones(float f) { apply(Op{kMpeBend, (double)f}); }
  void setPitchGlideInSeconds(float f) { apply(Op{kGlide, (double)f}); }
  void setDriftAmount(float f) { apply(Op{kDrift, (double)f}); }
  void setUnison(bool b) { apply(Op{kUnison, b ? 1. : 0.}); }
  void setProtocol(Symbol p) { apply(Op{kProtocol, (p == Symbol("MPE")) ? 1. : 0.}); }
  void setModCC(int c) { apply(Op{kModCC, (double)c}); }
  // the events of [startOffset, startOffset + 64) of the current host block -> this DSPVector of every voice's rows
  void processVector(int startOffset)
  {
    apply(Op{kProcess, (double)startOffset});
    fetchRows();
    if (controllersRunning_)
      for (size_t n = 0; n < kNumControllers; ++n)
        if (asked_[n]) fetchController(n);
  }
  const Voice& getVoice(int n) const { return voices_[(size_t)n]; }
  int getNewestVoice()
  {
    main_.make(polyphony_, -1);
    return mlgpu_events_newest_voice(main_.ev, 0);
  }
  // The reference keeps all 129 controller glides running from the first event on. Here the smoothed controllers are made by
  // helper objects (mlgpu_events with no voice rows, 32 watched controller numbers each) that start when a controller is first
  // asked for: they are then given everything this object has been told and has processed so far, so the signal is the
  // reference's from that vector on, whenever it is first read. An object that never asks keeps that history only up to
  // kHistoryOps operations (about a minute and a half of DSPVectors; MLGPU_E2S_HISTORY_OPS in the environment sets another bound); then
  // the helpers are started anyway and the history dropped.
  // clear() before the first processVector is the reference's; later it is mlgpu_events_clear: a complete reset of the voices, where
  // the reference's keep the running phase of their pitch glides and drift.
  const SmoothedController& getController(size_t n)
  {
    if (n >= kNumControllers) n = kNumControllers - 1;
    if (!controllersRunning_) startControllers();
    if (!asked_[n])
    {
      asked_[n] = true;
      fetchController(n);
    }
    return controllers_[n];
  }
  static constexpr size_t kHistoryOps{1u << 16};

 private:
  enum OpKind { kSampleRate, kPolyphony, kClear, kAddEvent, kClearEvents, kBend, kMpeBend, kGlide, kDrift, kUnison, kProtocol, kModCC, kProcess };
  static constexpr int kWatchedPerHelper{MLGPU_EVENTS_MAX_WATCHED_CONTROLLERS};
  static constexpr size_t kHelpers{(kNumControllers + kWatchedPerHelper - 1) / kWatchedPerHelper};
  struct Op
  {
    int kind;
    double x;
    Event e{};
  };
  // one mlgpu_events of one instrument: the object itself (all eight rows, no controllers) or a controller helper (no rows, the
  // controller numbers [first, first + 32))
  struct Impl
  {
    mlgpu_events* ev{nullptr};
    float* d{nullptr};  // device: the rows of one DSPVector
    size_t polyphony{0};
    bool processed{false};
    void release()
    {
      if (!ev && !d) return;
      std::lock_guard<std::recursive_mutex> lock(gpu::Eager::get().m);
      if (ev) mlgpu_events_destroy(ev);
      if (d) mlgpu_free(gpu::Eager::get().engine().handle(), d);
      ev = nullptr;
      d = nullptr;
    }
    // returns whether a new object was made
    bool make(size_t P, int firstController)
    {
      if (ev && (polyphony == P || firstController >= 0)) return false;  // controllers do not depend on the voice count
      release();
      const gpu::Engine& e = gpu::Eager::get().engine();
      e.check(mlgpu_events_create(e.handle(), 1, (int)P, &ev));
      polyphony = P;
      if (firstController >= 0)
      {
        int numbers[kWatchedPerHelper];
        int n = 0;
        for (; n < kWatchedPerHelper && firstController + n < (int)kNumControllers; ++n) numbers[n] = firstController + n;
        e.check(mlgpu_events_set_wanted_rows(ev, 0));
        e.check(mlgpu_events_watch_controllers(ev, numbers, n, 1));
      }
      else
      {
        void* p = nullptr;
        e.check(mlgpu_alloc(e.handle(), (size_t)kNumVoiceOutputRows * P * 64 * sizeof(float), &p));
        d = static_cast<float*>(p);
      }
      return true;
    }
    // firstController < 0: the object itself. `settings`: the last value of every setter, for an object made afresh (a change of
    // polyphony during set-up); `awake`: an event has been added before (awake_, :374 - it survives clear())
    void run(const Op& o, int firstController, const std::array<std::pair<bool, double>, kProcess>& settings, bool awake)
    {
      if (gpu::Capture::current())
        throw std::logic_error("mldsp GPU shim: an EventsToSignals object is stepped outside a captured process function (immediate mode); "
                               "inside one, gpu::SynthProgram feeds the voice rows");
      std::lock_guard<std::recursive_mutex> lock(gpu::Eager::get().m);
      const gpu::Engine& e = gpu::Eager::get().engine();
      const bool helper = firstController >= 0;
      const size_t wantP = o.kind == kPolyphony ? (size_t)o.x : (polyphony ? polyphony : 1);
      if (make(wantP, firstController))
      {
        for (int k = 0; k < (int)kProcess; ++k)
          if (settings[(size_t)k].first && (k == kSampleRate || (k >= kBend && k <= kModCC))) set(e, k, settings[(size_t)k].second);
        if (awake)
        {
          const mlgpu_event none{0, 0, 0, 0, 0.f, 0.f};  // addEvent wakes the object (:374), clearEvents takes the event away again
          e.check(mlgpu_events_add_event(ev, 0, &none));
          e.check(mlgpu_events_clear_events(ev));
        }
      }
      switch (o.kind)
      {
        case kPolyphony:  // clear(), :324
        case kClear:      // :331-341: the event buffer and the voices, not the controller smoothers
          e.check(helper ? mlgpu_events_clear_events(ev) : mlgpu_events_clear(ev));
          break;
        case kAddEvent:
        {
          const mlgpu_event m{o.e.type, o.e.channel, o.e.sourceIdx, o.e.time, o.e.value1, o.e.value2};
          e.check(mlgpu_events_add_event(ev, 0, &m));
          break;
        }
        case kClearEvents: e.check(mlgpu_events_clear_events(ev)); break;
        case kProcess:
        {
          float* rows[kNumVoiceOutputRows];
          for (int r = 0; r < kNumVoiceOutputRows; ++r) rows[r] = helper ? nullptr : d + (size_t)r * polyphony * 64;
          e.check(mlgpu_events_process(ev, 1, (int)o.x, rows, MLGPU_LAYOUT_VOICE_MAJOR));
          processed = true;
          break;
        }
        default: set(e, o.kind, o.x); break;
      }
    }
    void set(const gpu::Engine& e, int kind, double x)
    {
      switch (kind)
      {
        case kSampleRate: e.check(mlgpu_events_set_sample_rate(ev, x)); break;
        case kBend: e.check(mlgpu_events_set_pitch_bend_semitones(ev, (float)x)); break;
        case kMpeBend: e.check(mlgpu_events_set_mpe_pitch_bend_semitones(ev, (float)x)); break;
        case kGlide: e.check(mlgpu_events_set_pitch_glide_seconds(ev, (float)x)); break;
        case kDrift: e.check(mlgpu_events_set_drift_amount(ev, (float)x)); break;
        case kUnison: e.check(mlgpu_events_set_unison(ev, x != 0.)); break;
        case kProtocol: e.check(mlgpu_events_set_protocol(ev, x != 0.)); break;
        case kModCC: e.check(mlgpu_events_set_mod_cc(ev, (int)x)); break;
        default: break;
      }
    }
  };
  void apply(const Op& o)
  {
    main_.run(o, -1, settings_, awake_);
    if (controllersRunning_)
      for (size_t k = 0; k < kHelpers; ++k) helpers_[k].run(o, (int)k * kWatchedPerHelper, settings_, awake_);
    else
    {
      history_.push_back(o);
      if (history_.size() >= historyLimit_) startControllers();
    }
    if (o.kind < (int)kProcess) settings_[(size_t)o.kind] = {true, o.x};
    if (o.kind == kAddEvent) awake_ = true;
  }
  void startControllers()
  {
    static const std::array<std::pair<bool, double>, kProcess> none{};
    for (size_t k = 0; k < kHelpers; ++k)
      for (const Op& o : history_) helpers_[k].run(o, (int)k * kWatchedPerHelper, none, false);
    history_.clear();
    history_.shrink_to_fit();
    controllersRunning_ = true;
  }
  void fetchRows()
  {
    std::lock_guard<std::recursive_mutex> lock(gpu::Eager::get().m);
    const gpu::Engine& e = gpu::Eager::get().engine();
    const size_t P = main_.polyphony;
    std::vector<float> h((size_t)kNumVoiceOutputRows * P * 64);
    e.check(mlgpu_download(e.handle(), h.data(), main_.d, h.size() * sizeof(float)));
    for (size_t v = 0; v < P; ++v)
      for (int r = 0; r < kNumVoiceOutputRows; ++r)
        voices_[v].outputs.row(r) = DSPVector(static_cast<const float*>(h.data() + ((size_t)r * P + v) * 64));
  }
  void fetchController(size_t n)
  {
    Impl& h = helpers_[n / kWatchedPerHelper];
    if (!h.ev || !h.processed) return;  // nothing processed yet: the controller rests at zero
    std::lock_guard<std::recursive_mutex> lock(gpu::Eager::get().m);
    const gpu::Engine& e = gpu::Eager::get().engine();
    float host[64];  // one instrument: QUAD is the samples in order
    e.check(mlgpu_download(e.handle(), host, mlgpu_events_controller_signal(h.ev, (int)(n % kWatchedPerHelper)), sizeof(host)));
    controllers_[n].output = DSPVector(static_cast<const float*>(host));
  }

  size_t polyphony_{1}, historyLimit_{kHistoryOps};
  bool awake_{false}, controllersRunning_{false};
  std::array<std::pair<bool, double>, kProcess> settings_{};
  std::vector<Op> history_;
  Impl main_;
  std::array<Impl, kHelpers> helpers_;
  std::array<bool, kNumControllers> asked_{};
  std::array<Voice, kMaxVoices> voices_;
  std::array<SmoothedController, kNumControllers> controllers_;
};

class AudioContext
{
 public:
  // what getTimeInfo() hands out of ProcessTime (MLAudioContext.h:27-57)
  struct ProcessTime
  {
    double bpm{0.}, sampleRate{0.};
    uint64_t samplesSinceStart{0};
    DSPVector quarterNotesPhase_;
  };
  AudioContext(size_t nInputs, size_t nOutputs) : inputs(nInputs), outputs(nOutputs) {}
  AudioContext(size_t nInputs, size_t nOutputs, int rate) : inputs(nInputs), outputs(nOutputs), sampleRate_(rate) {}

  // Two lives. (1) Handed to gpu::VoiceProgram / gpu::SynthProgram, the context is what the captured process function reads its
  // voice rows, controllers and beat phase from: graph inputs (below, "in a capture"). (2) IMMEDIATE MODE - a host loop that steps
  // the context itself, as the reference's SignalProcessBuffer or a test does: processVector(offset) outside any capture makes this
  // an object like the reference's (MLAudioContext.cpp:104-140): its own EventsToSignals (one instrument, on the immediate
  // engine) and transport, getInputVoice(v) / getInputController(n) / getBeatPhase() returning host data. The device objects are
  // made at the first processVector(): what was set before is applied then - sample rate and the setInput...() values first, then
  // the events added so far.
  void setSampleRate(int r)
  {
    sampleRate_ = r;
    if (imm_) imm_->e2s.setSampleRate(r);
  }
  double getSampleRate() { return sampleRate_; }
  void setInputPolyphony(int voices)
  {
    polyphony_ = voices;
    if (imm_) imm_->e2s.setPolyphony((size_t)voices);
  }
  size_t getInputPolyphony() { return (size_t)polyphony_; }
  void setInputPitchBend(float p) { setInput(kBend, p); }
  void setInputMPEPitchBend(float p) { setInput(kMpeBend, p); }
  void setInputGlideTimeInSeconds(float s) { setInput(kGlide, s); }
  void setInputDriftAmount(float d) { setInput(kDrift, d); }
  void setInputUnison(bool u) { setInput(kUnison, u ? 1.f : 0.f); }
  void setInputProtocol(Symbol p) { setInput(kProtocol, (p == Symbol("MPE")) ? 1.f : 0.f); }
  void setInputModCC(int c) { setInput(kModCC, (float)c); }
  int getNewestInputVoice() { return immediate().e2s.getNewestVoice(); }

  // AudioContext::processVector (MLAudioContext.cpp:122-126): the transport's DSPVector, then the events of
  // [startOffset, startOffset + 64) of the host block into the voices' rows and the controllers. Immediate mode only.
  void processVector(int startOffset)
  {
    if (gpu::Capture::current())
      throw std::logic_error("mldsp GPU shim: AudioContext::processVector inside a captured process function (the program that runs the capture steps the context)");
    Immediate& m = immediate();
    stepTransport(m);
    m.e2s.processVector(startOffset);
  }
  void updateTime(const double ppqPos, const double bpm, bool isPlaying, double sampleRate)  // :135-139 -> ProcessTime::setTimeAndRate
  {
    if (timeReports_.size() >= 4096) timeReports_.erase(timeReports_.begin());  // (a context nobody steps: do not grow)
    timeReports_.push_back(TimeReport{ppqPos, bpm, sampleRate, isPlaying});
    if (imm_) flushTimeReports(*imm_);
  }
  void clear()  // :116-120
  {
    pendingEvents_.clear();
    timeReports_.clear();
    if (!imm_) return;
    std::lock_guard<std::recursive_mutex> lock(gpu::Eager::get().m);
    gpu::Eager::get().engine().check(mlgpu_transport_clear(imm_->transport, 0));
    imm_->e2s.clear();
  }
  const ProcessTime& getTimeInfo()
  {
    Immediate& m = immediate();
    flushTimeReports(m);
    m.time.samplesSinceStart = mlgpu_transport_samples_since_start(m.transport, 0);
    m.time.bpm = mlgpu_transport_bpm(m.transport, 0);
    m.time.sampleRate = sampleRate_;
    return m.time;
  }

  // In a capture there is ONE generic voice: whatever index is asked for, the rows are those of "this lane's voice"
  // (graph inputs fed by mlgpu_events). Per-voice constants belong in the kVoice row, not in the index.
  const EventsToSignals::Voice& getInputVoice(int n)
  {
    if (!gpu::Capture::current()) return immediate().e2s.getVoice(n);
    usesVoice_ = true;
    return voice_;
  }
  // Context signals (MLAudioContext.h:82,91): in a capture each is a graph input with one row per instrument, fed by
  // mlgpu_events_controller_signal / mlgpu_transport_beat_phase (gpu::SynthProgram does that; gpu::VoiceProgram::process takes the
  // pointers in the order of VoiceProgram::contextInputs()).
  DSPVector getInputController(size_t n) const
  {
    if (n > 128) n = 128;
    if (!gpu::Capture::current()) return immediate().e2s.getController(n).output;
    gpu::Sig s(gpu::Capture::get().contextInput((int)n), 0.f);
    s.hostCtx = (int)n;
    return DSPVector(s);
  }
  DSPVector getBeatPhase()
  {
    if (!gpu::Capture::current()) return immediate().time.quarterNotesPhase_;
    gpu::Sig s(gpu::Capture::get().contextInput(gpu::Capture::kBeatPhase), 0.f);
    s.hostCtx = gpu::Capture::kBeatPhase;
    return DSPVector(s);
  }
  // AudioContext::addInputEvent / clearInputEvents (MLAudioContext.h:60-62): events wait here until whoever runs the program
  // hands them to mlgpu_events (gpu::SynthProgram::addInputEvent does so directly for a bank of instruments); an immediate
  // context gives them to its own EventsToSignals
  void addInputEvent(const Event& e)
  {
    pendingEvents_.push_back(e);
    if (imm_) imm_->e2s.addEvent(e);
  }
  void clearInputEvents()
  {
    pendingEvents_.clear();
    if (imm_) imm_->e2s.clearEvents();
  }
  std::vector<Event> pendingEvents_;
  DSPVectorDynamic inputs;
  DSPVectorDynamic outputs;

  // used by the capture (gpu::VoiceProgram / gpu::SynthProgram)
  EventsToSignals::Voice voice_;
  bool usesVoice_{false};

 private:
  enum InputSetting { kBend, kMpeBend, kGlide, kDrift, kUnison, kProtocol, kModCC, kNumInputSettings };
  struct TimeReport
  {
    double ppq, bpm, sampleRate;
    bool playing;
  };
  struct Immediate
  {
    EventsToSignals e2s;
    mlgpu_transport* transport{nullptr};
    ProcessTime time;
    ~Immediate()
    {
      if (!transport) return;
      std::lock_guard<std::recursive_mutex> lock(gpu::Eager::get().m);
      mlgpu_transport_destroy(transport);
    }
  };
  void setInput(InputSetting k, float x)
  {
    inputSettings_[(size_t)k] = {true, x};
    if (imm_) applyInput(*imm_, k, x);
  }
  static void applyInput(Immediate& m, InputSetting k, float x)
  {
    switch (k)
    {
      case kBend: m.e2s.setPitchBendInSemitones(x); break;
      case kMpeBend: m.e2s.setMPEPitchBendInSemitones(x); break;
      case kGlide: m.e2s.setPitchGlideInSeconds(x); break;
      case kDrift: m.e2s.setDriftAmount(x); break;
      case kUnison: m.e2s.setUnison(x != 0.f); break;
      case kProtocol: m.e2s.setProtocol(x != 0.f ? Symbol("MPE") : Symbol("MIDI")); break;
      case kModCC: m.e2s.setModCC((int)x); break;
      default: break;
    }
  }
  Immediate& immediate() const
  {
    if (imm_) return *imm_;
    if (gpu::Capture::current()) throw std::logic_error("mldsp GPU shim: this AudioContext call belongs to immediate mode (outside a capture)");
    std::lock_guard<std::recursive_mutex> lock(gpu::Eager::get().m);
    const gpu::Engine& e = gpu::Eager::get().engine();
    auto m = std::make_shared<Immediate>();
    e.check(mlgpu_transport_create(e.handle(), 1, 1, &m->transport));
    if (sampleRate_ > 0) m->e2s.setSampleRate(sampleRate_);
    if (polyphony_ > 0) m->e2s.setPolyphony((size_t)polyphony_);
    for (size_t k = 0; k < (size_t)kNumInputSettings; ++k)
      if (inputSettings_[k].first) applyInput(*m, (InputSetting)k, inputSettings_[k].second);
    for (const Event& ev : pendingEvents_) m->e2s.addEvent(ev);
    imm_ = m;
    return *imm_;
  }
  void flushTimeReports(Immediate& m)
  {
    std::lock_guard<std::recursive_mutex> lock(gpu::Eager::get().m);
    const gpu::Engine& e = gpu::Eager::get().engine();
    for (const TimeReport& r : timeReports_) e.check(mlgpu_transport_set_time_and_rate(m.transport, 0, r.ppq, r.bpm, r.playing ? 1 : 0, r.sampleRate));
    timeReports_.clear();
  }
  void stepTransport(Immediate& m)
  {
    flushTimeReports(m);
    std::lock_guard<std::recursive_mutex> lock(gpu::Eager::get().m);
    const gpu::Engine& e = gpu::Eager::get().engine();
    e.check(mlgpu_transport_process(m.transport, 1));
    float host[64];  // one context: its DSPVector in sample order
    e.check(mlgpu_download(e.handle(), host, mlgpu_transport_beat_phase(m.transport), sizeof(host)));
    m.time.quarterNotesPhase_ = DSPVector(static_cast<const float*>(host));
  }

  double sampleRate_{0};
  int polyphony_{0};
  std::array<std::pair<bool, float>, kNumInputSettings> inputSettings_{};
  std::vector<TimeReport> timeReports_;
  mutable std::shared_ptr<Immediate> imm_;  // (shared: a copy of the context keeps stepping the same objects)
};
using SignalProcessFn = void (*)(AudioContext*, void*);

// AudioTask (source/app/MLAudioTask.h): the RtAudio main loop of the reference's console examples. There is no sound device
// behind this engine: the class exists so that such a program compiles; run its process function with
// ml::gpu::VoiceProgram(engine, voices, &ctx, processFn, &state) instead.
class AudioTask
{
 public:
  AudioTask(AudioContext*, SignalProcessFn, void*) {}
  int startAudio() { return 0; }
  void stopAudio() {}
  int runConsoleApp()
  {
    throw std::logic_error("mldsp GPU shim: AudioTask has no audio device; hand the process function to ml::gpu::VoiceProgram");
  }
};

#ifndef MLGPU_COMPAT_HAS_MADRONALIB_APP_HEADERS
// Path (source/app/MLPath.h): here only a name, as published signals use it.
class Path
{
  std::string text_;

 public:
  Path() = default;
  Path(const char* t) : text_(t) {}
  Path(const std::string& t) : text_(t) {}
  const std::string& text() const { return text_; }
  bool operator<(const Path& b) const { return text_ < b.text_; }
  bool operator==(const Path& b) const { return text_ == b.text_; }
};
namespace gpu
{
inline std::string pathText(const Path& p) { return p.text(); }
}  // namespace gpu
#else
namespace gpu
{
inline std::string pathText(const Path& p) { return std::string(pathToText(p).getText()); }  // madronalib's own Path
}  // namespace gpu
#endif

// SignalProcessor / Synth (source/app/MLSignalProcessor.h:121, MLSynth.h:26-94): the parts a DSP subclass overrides, and
// published signals. Parameters and the plug-in adapters of the reference are host-side plumbing and not part of the shim.
class SignalProcessor
{
 public:
  virtual ~SignalProcessor() = default;
  virtual void processVector(const DSPVectorDynamic& inputs, DSPVectorDynamic& outputs, void* stateData = nullptr) {}
  virtual void setSampleRate(double sr) { sampleRate_ = sr; }
  double getSampleRate() const { return sampleRate_; }

  // SignalProcessor::PublishedSignal (MLSignalProcessor.h:26-105). publishSignal() records the shape; the ring itself is an
  // mlgpu_published_signal that gpu::SynthProgram creates for every name the captured code stores to, and fills after each
  // launch with the voices of one instrument in rotation. The reading side is the reference's.
  struct PublishedSignal
  {
    int maxFrames_, maxVoices_, channels_, octavesDown_;
    mlgpu_published_signal* handle_{nullptr};  // owned by the gpu::SynthProgram that runs this processor - or, in immediate mode, by this object
    bool ownsHandle_{false};
    float* staging_{nullptr};  // immediate mode: the rows of one DSPVectorArray on the device
    PublishedSignal(int frames, int maxVoices, int channels, int octavesDown)
        : maxFrames_(frames), maxVoices_(maxVoices), channels_(channels), octavesDown_(octavesDown)
    {
    }
    PublishedSignal(const PublishedSignal&) = delete;
    PublishedSignal& operator=(const PublishedSignal&) = delete;
    ~PublishedSignal()
    {
      if (!ownsHandle_) return;
      std::lock_guard<std::recursive_mutex> lock(gpu::Eager::get().m);
      if (handle_) mlgpu_published_signal_destroy(handle_);
      if (staging_) mlgpu_free(gpu::Eager::get().engine().handle(), staging_);
    }
    // PublishedSignal::writeQuick (MLSignalProcessor.h:52-80) called directly - immediate mode only, whole DSPVectors
    template <size_t CHANNELS>
    void writeQuick(const DSPVectorArray<CHANNELS>& inputVector, size_t frames, size_t voice)
    {
      (void)voice;
      if (gpu::Capture::current()) throw std::logic_error("mldsp GPU shim: PublishedSignal::writeQuick inside a capture (use storePublishedSignal)");
      if (frames != kFloatsPerDSPVector) throw std::logic_error("mldsp GPU shim: PublishedSignal::writeQuick stores whole DSPVectors (frames == 64)");
      if ((int)CHANNELS != channels_) throw std::logic_error("mldsp GPU shim: PublishedSignal::writeQuick: channel count differs from the signal's");
      const float* rows[CHANNELS];
      for (size_t c = 0; c < CHANNELS; ++c) rows[c] = inputVector.constRow((int)c).getConstBuffer();
      writeImmediate(rows, CHANNELS);
    }
    // immediate mode (MLSignalProcessor.h:52-80, writeQuick): one voice's DSPVectorArray, host data, into the ring - every
    // (1 << octavesDown)-th frame, frame-major; voices in the order the processor stores them
    void writeImmediate(const float* const* rows, size_t channels)
    {
      std::lock_guard<std::recursive_mutex> lock(gpu::Eager::get().m);
      const gpu::Engine& e = gpu::Eager::get().engine();
      if (handle_ && !ownsHandle_)
        throw std::logic_error("mldsp GPU shim: this published signal is fed by a gpu::SynthProgram; storePublishedSignal outside its capture has nowhere to go");
      if (!handle_)
      {
        e.check(mlgpu_published_signal_create(e.handle(), maxFrames_, maxVoices_, channels_, octavesDown_, &handle_));
        ownsHandle_ = true;
        void* p = nullptr;
        e.check(mlgpu_alloc(e.handle(), (size_t)channels_ * 64 * sizeof(float), &p));
        staging_ = static_cast<float*>(p);
      }
      std::vector<const float*> d(channels);
      for (size_t c = 0; c < channels; ++c)
      {
        e.check(mlgpu_upload(e.handle(), staging_ + c * 64, rows[c], 64 * sizeof(float)));
        d[c] = staging_ + c * 64;
      }
      e.check(mlgpu_published_signal_write(handle_, 1, d.data(), MLGPU_LAYOUT_VOICE_MAJOR, 1, 0, 1));
    }
    size_t getNumChannels() const { return (size_t)channels_; }
    int getAvailableFrames() const { return handle_ ? (int)mlgpu_published_signal_available_frames(handle_) : 0; }
    int getReadAvailable() const { return handle_ ? (int)mlgpu_published_signal_read_available(handle_) : 0; }
    size_t read(float* dest, size_t framesRequested) { return handle_ ? mlgpu_published_signal_read(handle_, dest, framesRequested) : 0; }
    size_t readLatest(float* dest, size_t framesRequested) { return handle_ ? mlgpu_published_signal_read_latest(handle_, dest, framesRequested) : 0; }
    void peekLatest(float* dest, size_t framesRequested)
    {
      if (handle_) mlgpu_published_signal_peek_latest(handle_, dest, framesRequested);
    }
  };
  // stands in for Tree<std::unique_ptr<PublishedSignal>> (MLSignalProcessor.h:135-139, :175): operator[], find and iteration,
  // keyed by the path's text (so it works with this header's own Path and with madronalib's)
  class PublishedSignalTree
  {
    std::map<std::string, std::unique_ptr<PublishedSignal>> m_;

   public:
    std::unique_ptr<PublishedSignal>& operator[](const Path& p) { return m_[gpu::pathText(p)]; }
    auto find(const Path& p) { return m_.find(gpu::pathText(p)); }
    auto begin() { return m_.begin(); }
    auto end() { return m_.end(); }
    auto begin() const { return m_.begin(); }
    auto end() const { return m_.end(); }
    size_t size() const { return m_.size(); }
  };
  PublishedSignalTree& getPublishedSignals() { return publishedSignals_; }
  const PublishedSignalTree& getPublishedSignals() const { return publishedSignals_; }

#ifdef MLGPU_COMPAT_HAS_MADRONALIB_APP_HEADERS
  // the parameter calls of the reference's SignalProcessor (MLSignalProcessor.h:127-170), over madronalib's own ParameterTree
  ParameterTree& getParameterTree() { return params_; }
  const ParameterTree& getParameterTree() const { return params_; }
  size_t getParameterCount() const { return params_.descriptions.size(); }
  void setParamFromNormalizedValue(Path pname, float val) { params_.setFromNormalizedValue(pname, val); }
  void setParamFromRealValue(Path pname, float val) { params_.setFromRealValue(pname, val); }
  void buildParams(const ParameterDescriptionList& paramList) { buildParameterTree(paramList, params_); }
  void setDefaultParams() { setDefaults(params_); }
  float getRealFloatParam(Path pname) { return params_.getRealFloatValueAtPath(pname); }
  float getNormalizedFloatParam(Path pname) { return params_.getNormalizedFloatValueAtPath(pname); }

 protected:
  ParameterTree params_;

 public:
#endif

 protected:
  double sampleRate_{48000.0};
  PublishedSignalTree publishedSignals_;

  // MLSignalProcessor.h:184-187
  void publishSignal(Path signalName, int maxFrames, int maxVoices, int channels, int octavesDown)
  {
    publishedSignals_[signalName] = std::make_unique<PublishedSignal>(maxFrames, maxVoices, channels, octavesDown);
  }
  // MLSignalProcessor.h:192-200. In a capture the DSPVectorArray's rows become further outputs of the fused kernel; `voice` is
  // the lane's voice (the reference does not use it either). Only whole DSPVectors (frames == kFloatsPerDSPVector).
  template <size_t CHANNELS>
  void storePublishedSignal(Path signalName, const DSPVectorArray<CHANNELS>& inputVec, int frames, int voice)
  {
    (void)voice;
    auto it = publishedSignals_.find(signalName);
    if (it == publishedSignals_.end() || !it->second) return;  // not published: ignored, as in the reference
    if (frames != (int)kFloatsPerDSPVector) throw std::logic_error("mldsp GPU shim: storePublishedSignal stores whole DSPVectors (frames == 64)");
    if ((int)CHANNELS != it->second->channels_) throw std::logic_error("mldsp GPU shim: storePublishedSignal: channel count differs from publishSignal");
    if (!gpu::Capture::current())  // immediate mode: this voice's DSPVectorArray is host data
    {
      const float* rows[CHANNELS];
      for (size_t c = 0; c < CHANNELS; ++c) rows[c] = inputVec.constRow((int)c).getConstBuffer();
      it->second->writeImmediate(rows, CHANNELS);
      return;
    }
    gpu::Capture& cap = gpu::Capture::get();
    for (auto& t : cap.taps)
      if (t.name == gpu::pathText(signalName)) throw std::logic_error("mldsp GPU shim: one storePublishedSignal per name in the captured voice code");
    gpu::Capture::Tap tap;
    tap.name = gpu::pathText(signalName);
    for (size_t c = 0; c < CHANNELS; ++c) tap.nodes.push_back(inputVec.sig_[c].id());
    cap.taps.push_back(tap);
  }
};

class Synth : public SignalProcessor
{
 public:
  static constexpr int kDefaultNumVoices = 8;
  Synth(int numVoices = kDefaultNumVoices) : numVoices_(numVoices) {}
  virtual ~Synth() = default;
  // per-voice DSP: mix into outputs with += (MLSynth.h:62-72). Captured ONCE for a generic voice (voiceIndex 0 selects the
  // per-voice objects a subclass keeps in arrays) and run for every voice of every instrument; the voice sum of
  // Synth::processVector (MLSynth.h:43-57) is mlgpu_mixdown_groups, in the same order.
  virtual void processVoice(int voiceIndex, const EventsToSignals::Voice& voice, const DSPVectorDynamic& inputs, DSPVectorDynamic& outputs,
                            AudioContext* audioContext) = 0;
  virtual bool isVoiceActive(int, const EventsToSignals::Voice&) { return true; }
  int getNumVoices() const { return numVoices_; }
  // Synth::processVector (MLSynth.h:38-61) - immediate mode, a context that is stepped by its host loop: the outputs start from
  // zero and every active voice is mixed in, in voice order, with the rows of its own voice. (A captured program never comes
  // here: gpu::SynthProgram captures processVoice for one generic voice and sums the voices on the device in the same order.)
  void processVector(const DSPVectorDynamic& inputs, DSPVectorDynamic& outputs, void* stateData) override
  {
    AudioContext* ctx = static_cast<AudioContext*>(stateData);
    if (!ctx) return;
    if (gpu::Capture::current())
      throw std::logic_error("mldsp GPU shim: Synth::processVector inside a capture - hand the Synth to ml::gpu::SynthProgram instead");
    for (size_t i = 0; i < outputs.size(); ++i) outputs[i] = DSPVector(0.f);
    int active = 0;
    for (int v = 0; v < numVoices_; ++v)
    {
      const EventsToSignals::Voice& voice = ctx->getInputVoice(v);
      if (!isVoiceActive(v, voice)) continue;
      ++active;
      processVoice(v, voice, inputs, outputs, ctx);
    }
    activeVoiceCount_ = active;
  }
  int getActiveVoiceCount() const { return activeVoiceCount_; }
  virtual bool hasActiveVoices() const { return activeVoiceCount_ > 0; }

 protected:
  int numVoices_;
  int activeVoiceCount_{0};
};

// ---- host-side helpers of MLDSPUtils.h / MLDSPBuffer.h ------------------------------------------------------------------
#ifndef MLGPU_COMPAT_HAS_MADRONALIB_SCALAR_HEADERS
using Projection = std::function<float(float)>;  // MLDSPProjections.h:33
#endif
// mapIndices / makeWindow (MLDSPUtils.h:15-26): pDest[i] = shape(i mapped linearly from [0, size - 1] to [0, 1]) - the
// mapping spelled out (projections::linear, MLDSPProjections.h:147-167) so that it does not need madronalib's header
inline void mapIndices(float* pDest, size_t size, Projection p)
{
  for (size_t i = 0; i < size; ++i) pDest[i] = p((float)(int)i);
}
inline void makeWindow(float* pDest, size_t size, Projection windowShape)
{
  const float a2 = size - 1.f;
  if (0.f - a2 == 0.f)
  {
    mapIndices(pDest, size, [=](float) { return windowShape(0.f); });
    return;
  }
  const float m = (1.f - 0.f) / (a2 - 0.f);
  mapIndices(pDest, size, [=](float x) { return windowShape(m * (x - 0.f) + 0.f); });
}
namespace dspwindows  // MLDSPUtils.h:28-47
{
const Projection rectangle([](float x) { return (x > 0.75f) ? 0.f : ((x < 0.25f) ? 0.f : 1.f); });
const Projection triangle([](float x) { return (x > 0.5f) ? (2.f - 2.f * x) : (2.f * x); });
const Projection raisedCosine([](float x) { return 0.5f - 0.5f * cosf(kTwoPi * x); });
const Projection hamming([](float x) { return 0.54f - 0.46f * cosf(kTwoPi * x); });
const Projection blackman([](float x) { return 0.42f - 0.5f * cosf(kTwoPi * x) + 0.08f * cosf(2.f * kTwoPi * x); });
const Projection flatTop([](float x) {
  const float a0 = 0.21557895f, a1 = 0.41663158f, a2 = 0.277263158f, a3 = 0.083578947f, a4 = 0.006947368f;
  return a0 - a1 * cosf(kTwoPi * x) + a2 * cosf(2.f * kTwoPi * x) - a3 * cosf(3.f * kTwoPi * x) + a4 * cosf(4.f * kTwoPi * x);
});
}  // namespace dspwindows

// DSPBuffer (MLDSPBuffer.h): the host ring of the C-ABI under the reference's name, with its DSPVector forms - for vectors the
// host holds (made with getBuffer(), tables, literals); what it reads comes back as host vectors
class DSPBuffer : public gpu::DSPBuffer
{
 public:
  using gpu::DSPBuffer::DSPBuffer;
  using gpu::DSPBuffer::read;
  using gpu::DSPBuffer::write;
  template <size_t ROWS>
  void write(const DSPVectorArray<ROWS>& v)  // :171-204
  {
    gpu::DSPBuffer::write(v.getConstBuffer(), 64 * ROWS);
  }
  template <size_t ROWS>
  void read(DSPVectorArray<ROWS>& dest)  // :227-277: zeros when fewer samples wait
  {
    float* p = dest.getBuffer();
    if (getReadAvailable() >= 64 * ROWS) gpu::DSPBuffer::read(p, 64 * ROWS);
    else std::fill(p, p + 64 * ROWS, 0.f);
  }
};

// SignalProcessBuffer (source/app/MLSignalProcessBuffer.h:20-35, .cpp:16-96) in immediate mode: the adaptor between a host's blocks
// of any length and a process function that works a DSPVector at a time. Host blocks go into one ring per input; while the output
// rings hold less than the host asked for, one more DSPVector is made - the inputs' next 64 samples (zeros while a ring is still
// short), ctx->processVector(offset) with the offset counted from the start of this host block, the process function, its outputs
// into the output rings; then the host's frames are read out and the block's events dropped. A block longer than maxFrames, or
// no output pointers, is ignored, as in the reference. (The device counterpart for banks of voices is mlgpu_process_buffer.)
class SignalProcessBuffer final
{
 public:
  SignalProcessBuffer(size_t inputs, size_t outputs, size_t maxFrames) : in_(inputs), out_(outputs), maxFrames_(maxFrames)
  {
    for (DSPBuffer& b : in_) b.resize((int)maxFrames);
    for (DSPBuffer& b : out_) b.resize((int)maxFrames);
  }
  void process(const float** externalInputs, float** externalOutputs, int externalFrames, AudioContext* context, SignalProcessFn processFn, void* state)
  {
    if (out_.empty() || !externalOutputs || externalFrames > (int)maxFrames_) return;
    for (size_t c = 0; c < in_.size(); ++c)
      if (externalInputs[c]) in_[c].write(externalInputs[c], (size_t)externalFrames);
    for (int offset = 0; (int)out_[0].getReadAvailable() < externalFrames; offset += (int)kFloatsPerDSPVector)
    {
      for (size_t c = 0; c < in_.size(); ++c) in_[c].read(context->inputs[(int)c]);
      context->processVector(offset);
      processFn(context, state);
      for (size_t c = 0; c < out_.size(); ++c) out_[c].write(context->outputs[(int)c]);
    }
    for (size_t c = 0; c < out_.size(); ++c)
      if (externalOutputs[c]) out_[c].read(externalOutputs[c], (size_t)externalFrames);
    context->clearInputEvents();
  }

 private:
  std::vector<DSPBuffer> in_, out_;
  size_t maxFrames_;
};

namespace gpu
{
// A per-voice constant (`DSPVector(f)` with a different f for every voice): converts to DSPVector.
class VoiceParam
{
  std::string name_;
  int node_{-1};
  uint32_t epoch_{0};

 public:
  explicit VoiceParam(const char* name) : name_(name) {}
  operator DSPVector()
  {
    Capture& c = Capture::get();
    if (node_ < 0 || epoch_ != c.epoch) node_ = c.ret(mlgpu_graph_add_param(c.g, name_.c_str()));
    epoch_ = c.epoch;
    return DSPVector(Sig(node_, 0.f));
  }
  int node() const { return node_; }
  const std::string& name() const { return name_; }
};

// how the captured graph is built (all default off): the ring layout for per-voice delay times (mlgpu_graph_set_delay_layout)
// and online tuning of the kernel form (mlgpu_graph_set_autotune)
struct VoiceProgramOptions
{
  bool delayWindows{false};
  bool autotune{false};
  bool liveConstants{false};  // constants are read from a device table: VoiceProgram::update() can change them (mlgpu_graph_set_live_constants)
  // SynthProgram: if processVoice reads no voice row but pitch and gate, the voice kernel computes those two rows itself from the
  // events' records (mlgpu_graph_add_event_row) instead of reading them from memory - a quarter faster end to end. MIDI protocol
  // only (mlgpu_events_set_protocol(..., 1) makes process() fail), hence opt-in.
  bool eventRowsInKernel{false};
  // Context signals (AudioContext::getInputController / getBeatPhase) have one row per this many adjacent voices: the
  // instrument's polyphony. 0: ctx->getInputPolyphony() if that was set, else one row for the whole bank.
  size_t voicesPerContext{0};
  // The process function reads samples of context signals into host floats (`ctrlToFreq(ctrlSig[0])`, the reference's
  // controllers-to-audio.cpp): allowed for a program of ONE context, run a DSPVector at a time - before each launch
  // readContextSamples() fetches the signals' next 64 samples and update() runs the function again with them, so its host
  // floats are what the reference computes for that vector. Needs liveConstants.
  bool hostContextSamples{false};
  // SynthProgram: EventsToSignals on an engine (HIP stream) of its own, so that its kernel for block k + 1 runs beside the voice
  // kernel of block k (two sets of row signals, fences either way round). Same results; -10 % per block where the events kernel is a
  // visible part of it. Used when the rows are read from memory and processVoice reads no controller signal (those are made by the
  // events call into one buffer); otherwise the one-stream order is kept.
  bool eventsOnOwnStream{false};
  // VoiceProgram: every audio output (ctx->outputs[c]) is the SUM of all voices - one channel, `outputs += voice` over the whole
  // bank in mlgpu_mixdown's order, made inside the voice kernel (mlgpu_graph_set_output_mixdown): pass single-voice DeviceSignals
  // for them to process(), and call reserveMixdown(max vectors per launch) once at setup. Published signals stay per voice.
  bool mixOutputs{false};
};

// Captures a reference-style process function once and runs it for `voices` voices on the GPU.
class VoiceProgram
{
  const Engine& eng_;
  mlgpu_graph* g_{nullptr};
  size_t voices_;
  size_t nIn_{0}, nOut_{0};
  bool usesVoice_{false};
  unsigned voiceRowMask_{0};  // which of the 8 voice control rows the captured code reads: only those are graph inputs
  std::vector<Capture::Tap> taps_;  // published signals: graph outputs after the nOut_ audio outputs, in this order
  std::vector<int> contextInputs_;  // context signals the code reads: graph inputs after the audio inputs and the voice rows
  AudioContext* ctx_{nullptr};
  std::function<void(AudioContext*)> body_;
  VoiceProgramOptions opt_;
  bool flush_{false};  // the process function runs inside an ml::UsingFlushDenormalsToZero scope
  std::map<int, std::array<float, 64>> hostContext_;  // VoiceProgramOptions::hostContextSamples: the context signals' next DSPVector

 public:
  VoiceProgram(const Engine& e, size_t voices, AudioContext* ctx, SignalProcessFn fn, void* state, VoiceProgramOptions opt = VoiceProgramOptions())
      : VoiceProgram(e, voices, ctx, [fn, state](AudioContext* c) { fn(c, state); }, opt)
  {
  }
  // general form: `body` is run twice in capture mode; it reads ctx->inputs / ctx->getInputVoice() and writes ctx->outputs
  VoiceProgram(const Engine& e, size_t voices, AudioContext* ctx, std::function<void(AudioContext*)> body, VoiceProgramOptions opt = VoiceProgramOptions())
      : eng_(e), voices_(voices), ctx_(ctx), body_(std::move(body)), opt_(opt)
  {
    ctx->usesVoice_ = false;
    nIn_ = ctx->inputs.size();
    nOut_ = ctx->outputs.size();
    Capture cap;
    cap.eng = &e;
    cap.dedupeConstants = !opt.liveConstants;  // live: one node per use, so that a later capture with other numbers lines up
    cap.contextGroup = contextGroup();
    if (voices % cap.contextGroup) throw Error(MLGPU_ERR_INVALID, "VoiceProgram: the voices are not a whole number of instruments (voicesPerContext)");
    if (opt.hostContextSamples && (voices != cap.contextGroup || !opt.liveConstants))
      throw Error(MLGPU_ERR_INVALID, "VoiceProgram: hostContextSamples needs a program of one context (voices == voicesPerContext) and liveConstants");
    cap.hostContextAllowed = opt.hostContextSamples;
    CaptureScope scope(&cap);
    // pass 1 records what the process function leaves behind in the user's state (DSPVectors kept for the next call);
    // pass 2 builds the graph that is compiled, reading those as one-vector feedback
    for (int pass = 0; pass < 2; ++pass)
    {
      if (g_) mlgpu_graph_destroy(g_);
      g_ = nullptr;
      eng_.check(mlgpu_graph_create(e.handle(), voices, &g_));
      if (opt.delayWindows) eng_.check(mlgpu_graph_set_delay_layout(g_, 3));  // transposed pieces where they apply, else sectors
      if (opt.autotune) eng_.check(mlgpu_graph_set_autotune(g_, 1));
      if (opt.liveConstants) eng_.check(mlgpu_graph_set_live_constants(g_, 1));
      int rowNode[kNumVoiceOutputRows];
      capturePass(cap, g_, pass == 0, rowNode);
      if (pass == 0 && ctx->usesVoice_)
        for (int r = 0; r < kNumVoiceOutputRows; ++r)
        {
          bool used = mlgpu_graph_node_use_count(g_, rowNode[r]) > 0;
          for (size_t c = 0; c < nOut_; ++c) used = used || (ctx->outputs[(int)c].sig_[0].node == rowNode[r]);
          if (used) voiceRowMask_ |= 1u << r;
        }
    }
    usesVoice_ = voiceRowMask_ != 0;
    flush_ = cap.flushDenormals;
    finishGraph(cap, g_);
    taps_ = cap.taps;
    contextInputs_ = cap.contextInputs;
    if (opt.mixOutputs)
      for (size_t c = 0; c < nOut_; ++c) eng_.check(mlgpu_graph_set_output_mixdown(g_, (int)c, 1));
    eng_.check(mlgpu_graph_compile(g_));
    for (const Capture::Deferred& d : cap.deferred)
    {
      float f;
      std::memcpy(&f, &d.bits, 4);
      if (d.what == 0) eng_.check(mlgpu_graph_set_coeff_uniform(g_, d.node, d.idx, f));
      else if (d.what == 1) eng_.check(mlgpu_graph_clear_proc(g_, d.node));
      else eng_.check(mlgpu_graph_set_state_uniform(g_, d.node, d.idx, d.bits));
    }
  }

  // VoiceProgramOptions::mixOutputs: the scratch the mixdown's later stages need, for launches of up to maxVectors DSPVectors
  // (setup; process calls never allocate)
  void reserveMixdown(size_t maxVectors) { eng_.check(mlgpu_graph_reserve_mixdown(g_, maxVectors)); }

  // Host-side numbers changed (a parameter the process function turns into `DSPVector(value)`, a float argument of a
  // LinearGlide, `filter.coeffs = makeCoeffs(...)`): run the process function once more in capture mode and take the new
  // constants and coefficients into the running kernel. State (phases, filter memories, delay lines) is untouched; nothing is
  // recompiled. Needs VoiceProgramOptions::liveConstants when constants change (coefficients alone do not); throws
  // gpu::Error(MLGPU_ERR_UNSUPPORTED) when the function took a different path and built a different graph.
  void update()
  {
    Capture cap;
    cap.eng = &eng_;
    cap.dedupeConstants = !opt_.liveConstants;
    cap.contextGroup = contextGroup();
    cap.hostContextAllowed = opt_.hostContextSamples;
    cap.hostContext = hostContext_.empty() ? nullptr : &hostContext_;
    CaptureScope scope(&cap);
    mlgpu_graph* tmp = nullptr;
    eng_.check(mlgpu_graph_create(eng_.handle(), voices_, &tmp));
    struct Guard
    {
      mlgpu_graph* g;
      ~Guard() { mlgpu_graph_destroy(g); }
    } guard{tmp};
    int rowNode[kNumVoiceOutputRows];
    capturePass(cap, tmp, false, rowNode);  // the user's kept DSPVectors hold nodes of the previous capture: feedback, as in pass 2
    finishGraph(cap, tmp);
    eng_.check(mlgpu_graph_update_constants_from(g_, tmp));
    for (const Capture::Deferred& d : cap.deferred)
      if (d.what == 0)
      {
        float f;
        std::memcpy(&f, &d.bits, 4);
        eng_.check(mlgpu_graph_set_coeff_uniform(g_, d.node, d.idx, f));
      }
  }

 private:
  struct CaptureScope
  {
    Capture* prev;
    explicit CaptureScope(Capture* c) : prev(Capture::current()) { Capture::current() = c; }
    ~CaptureScope() { Capture::current() = prev; }
  };
  static uint32_t& epochCounter()
  {
    static uint32_t n = 0;
    return n;
  }
  // one run of the process function against graph g
  void capturePass(Capture& cap, mlgpu_graph* g, bool firstPass, int* rowNode)
  {
    cap.g = g;
    cap.epoch = ++epochCounter();
    cap.nextOrd = 0;
    cap.nodeOfOrd.clear();
    cap.feedbackOfOrd.clear();
    cap.constNodes.clear();
    cap.tableNodes.clear();
    cap.regionCounter = cap.curRegion = 0;
    cap.deferred.clear();
    cap.taps.clear();
    cap.contextInputs.clear();
    cap.contextNode.clear();
    cap.inputCount = 0;
    for (size_t c = 0; c < nIn_; ++c, ++cap.inputCount)
      ctx_->inputs[(int)c] = DSPVector(Sig(cap.ret(mlgpu_graph_add_input(g, ("in" + std::to_string(c)).c_str())), 0.f));
    // the 8 rows of this lane's voice (EventsToSignals) are further streamed inputs, after the audio inputs; the first pass
    // finds out whether the code reads them at all
    for (int r = 0; r < kNumVoiceOutputRows; ++r)
    {
      rowNode[r] = -1;
      if (firstPass || ((voiceRowMask_ >> r) & 1u))
      {
        if (!firstPass && eventRowsInKernel()) rowNode[r] = cap.ret(mlgpu_graph_add_event_row(g, r, ("voice" + std::to_string(r)).c_str()));
        else
        {
          rowNode[r] = cap.ret(mlgpu_graph_add_input(g, ("voice" + std::to_string(r)).c_str()));
          ++cap.inputCount;
        }
        ctx_->voice_.outputs.row(r) = DSPVector(Sig(rowNode[r], 0.f));
      }
      else
        ctx_->voice_.outputs.row(r) = DSPVector(0.f);  // never read (pass 1 saw no use of it)
    }
    for (size_t c = 0; c < nOut_; ++c) ctx_->outputs[(int)c] = DSPVector(0.f);
    body_(ctx_);
  }
  void finishGraph(Capture& cap, mlgpu_graph* g)
  {
    for (auto& kv : cap.feedbackOfOrd)
    {
      if (kv.first >= (int)cap.nodeOfOrd.size()) throw std::logic_error("mldsp GPU shim: the process function took different paths in its two capture passes");
      eng_.check(mlgpu_graph_set_feedback(g, kv.second, cap.nodeOfOrd[kv.first]));
    }
    for (size_t c = 0; c < nOut_; ++c) eng_.check(mlgpu_graph_add_output(g, ctx_->outputs[(int)c].sig_[0].id()));
    for (const Capture::Tap& t : cap.taps)
      for (int node : t.nodes) eng_.check(mlgpu_graph_add_output(g, node));
  }

 public:
  VoiceProgram(const VoiceProgram&) = delete;
  VoiceProgram& operator=(const VoiceProgram&) = delete;
  ~VoiceProgram()
  {
    if (g_) mlgpu_graph_destroy(g_);
  }

  size_t voices() const { return voices_; }
  bool flushesDenormals() const { return flush_; }  // the captured code holds an ml::UsingFlushDenormalsToZero
  unsigned voiceRowMask() const { return voiceRowMask_; }  // bit r: the captured code reads voice row r (VoiceOutputSignals)
  // the voice rows are computed inside the kernel (VoiceProgramOptions::eventRowsInKernel and nothing but pitch / gate is read)
  bool eventRowsInKernel() const { return opt_.eventRowsInKernel && voiceRowMask_ != 0 && (voiceRowMask_ & ~3u) == 0; }
  mlgpu_graph* graph() const { return g_; }
  // the context signals the captured code reads, in the order process() takes them: a controller number (0..128), or
  // Capture::kBeatPhase; each is a signal of voices() / contextGroup() rows
  const std::vector<int>& contextInputs() const { return contextInputs_; }
  size_t contextGroup() const
  {
    if (opt_.voicesPerContext) return opt_.voicesPerContext;
    return ctx_->getInputPolyphony() ? ctx_->getInputPolyphony() : voices_;
  }
  const std::vector<Capture::Tap>& taps() const { return taps_; }
  size_t tapChannels() const
  {
    size_t n = 0;
    for (auto& t : taps_) n += t.nodes.size();
    return n;
  }
  const char* source() const { return mlgpu_graph_source(g_); }  // the generated HIP kernel

  // VoiceProgramOptions::hostContextSamples: fetch DSPVector `vector` of every context signal (the pointers process() takes,
  // device memory, one row: QUAD == plain order) to the host, for the process function to index. One small copy and a wait
  // per call: the price of running host code on signal values, as the reference does per DSPVector.
  void readContextSamples(const float* const* contextSignals, size_t vector = 0)
  {
    if (!opt_.hostContextSamples) throw Error(MLGPU_ERR_INVALID, "VoiceProgram::readContextSamples: not a hostContextSamples program");
    for (size_t c = 0; c < contextInputs_.size(); ++c)
      eng_.check(mlgpu_download(eng_.handle(), hostContext_[contextInputs_[c]].data(), contextSignals[c] + vector * 64, 64 * sizeof(float)));
  }
  // the same from host memory (a caller that computes its context signals itself)
  void setContextSamples(int code, const float* samples64) { std::memcpy(hostContext_[code].data(), samples64, 64 * sizeof(float)); }

  // per-voice values: [voices] floats
  void setParam(const VoiceParam& p, const std::vector<float>& perVoice) { eng_.check(mlgpu_graph_set_param(g_, p.node(), perVoice.data())); }
  template <class P>
  void setCoeff(const P& object, int coeffIdx, const std::vector<float>& perVoice)
  {
    eng_.check(mlgpu_graph_set_coeff(g_, object.node(), coeffIdx, perVoice.data()));
  }
  template <class P>
  void setState(const P& object, int stateIdx, const std::vector<uint32_t>& perVoice)
  {
    eng_.check(mlgpu_graph_set_state(g_, object.node(), stateIdx, perVoice.data()));
  }

  // OneShotGen::trigger() between two process() calls: for every voice, or for the voices whose flag is set
  void trigger(const ::ml::OneShotGen& shot)
  {
    eng_.check(mlgpu_graph_set_state_uniform(g_, shot.node(), 0, 0));
    eng_.check(mlgpu_graph_set_state_uniform(g_, shot.node(), 1, 1));
    eng_.check(mlgpu_graph_set_state_uniform(g_, shot.node(), 2, 0));
  }
  void trigger(const ::ml::OneShotGen& shot, const std::vector<uint8_t>& whichVoices)
  {
    std::vector<uint32_t> w(whichVoices.size());
    for (int idx = 0; idx < 3; ++idx)
    {
      eng_.check(mlgpu_graph_get_state(g_, shot.node(), idx, w.data()));
      for (size_t v = 0; v < w.size(); ++v)
        if (whichVoices[v]) w[v] = idx == 1 ? 1u : 0u;
      eng_.check(mlgpu_graph_set_state(g_, shot.node(), idx, w.data()));
    }
  }

  // one call = T DSPVectors of every voice (the reference calls the process function T times)
  // voiceRows: the 8 signals of mlgpu_events_process for the same voices (needed when the captured code called
  // getInputVoice(); nullptr otherwise)
  // contextSignals: one device signal (QUAD, voices() / contextGroup() rows) per entry of contextInputs()
  void process(const std::vector<const DeviceSignal*>& ins, const std::vector<DeviceSignal*>& outs, const float* const* voiceRows = nullptr,
               const float* const* contextSignals = nullptr)
  {
    if (!contextInputs_.empty() && !contextSignals)
      throw Error(MLGPU_ERR_INVALID, "VoiceProgram::process: this program reads context signals (controllers / beat phase); pass them");
    if (ins.size() != nIn_ || outs.size() != nOut_ + tapChannels() || outs.empty())
      throw Error(MLGPU_ERR_INVALID, "VoiceProgram::process: wrong number of signals (outputs: the audio outputs, then one per published channel)");
    if (usesVoice_ && !voiceRows) throw Error(MLGPU_ERR_INVALID, "VoiceProgram::process: this program reads the voice control rows; pass them");
    std::vector<const float*> pi;
    std::vector<float*> po;
    for (auto* s : ins) pi.push_back(s->data());
    // rows the code does not read still need a valid pointer: any output buffer of this launch will do
    for (int r = 0; r < kNumVoiceOutputRows; ++r)
      if ((voiceRowMask_ >> r) & 1u)
      {
        if (!voiceRows[r]) throw Error(MLGPU_ERR_INVALID, "VoiceProgram::process: the program reads voice row " + std::to_string(r) + "; pass it");
        pi.push_back(voiceRows[r]);
      }
    for (size_t c = 0; c < contextInputs_.size(); ++c)
    {
      if (!contextSignals[c]) throw Error(MLGPU_ERR_INVALID, "VoiceProgram::process: null context signal");
      pi.push_back(contextSignals[c]);
    }
    for (auto* s : outs) po.push_back(s->data());
    const int inLayout = ins.empty() ? MLGPU_LAYOUT_QUAD : ins[0]->layout();
    // the mode travels with the launch (a kernel argument), so it is restored as soon as the launch is enqueued
    const int prevMode = mlgpu_engine_get_flush_denormals(eng_.handle());
    if (flush_ && !prevMode) eng_.check(mlgpu_engine_set_flush_denormals(eng_.handle(), 1));
    const int st = mlgpu_graph_process(g_, outs[0]->vectors(), pi.data(), inLayout, po.data(), outs[0]->layout());
    if (flush_ && !prevMode) mlgpu_engine_set_flush_denormals(eng_.handle(), 0);
    eng_.check(st);
  }
};

// A whole polyphonic instrument bank on the GPU: events in, mixed audio out.
//   mlgpu_events (EventsToSignals for nInstruments x polyphony voices)  ->  the Synth subclass's processVoice, captured once
//   and fused into one kernel  ->  mlgpu_mixdown_groups (the per-instrument voice sum of Synth::processVector).
class SynthProgram
{
  const Engine& eng_;
  size_t nInstruments_;
  int polyphony_;
  size_t nOut_;
  AudioContext ctx_;
  mlgpu_events* ev_{nullptr};
  VoiceProgram prog_;
  size_t capacityT_{0};
  std::vector<DeviceSignal> rows_, voiceOut_, tapOut_;
  std::vector<SignalProcessor::PublishedSignal*> published_;  // one per tap of prog_, in tap order
  size_t publishedInstrument_{0};
  std::vector<int> controllers_;  // controller numbers processVoice reads through ctx->getInputController(n)
  mlgpu_transport* transport_{nullptr};  // there when processVoice reads ctx->getBeatPhase(): one ProcessTime per instrument
  // VoiceProgramOptions::eventsOnOwnStream
  bool wantOwnStream_{false};
  std::unique_ptr<Engine> evEngine_;
  std::vector<DeviceSignal> rowsB_;                    // the second set of row signals
  std::unique_ptr<Fence> written_[2], read_[2];        // row set s: written by the events kernel / read by the voice kernel
  size_t blockCount_{0};
  const Engine& eventsEngine() const { return evEngine_ ? *evEngine_ : eng_; }
  static VoiceProgramOptions perInstrument(VoiceProgramOptions o, int polyphony)
  {
    o.voicesPerContext = (size_t)polyphony;
    return o;
  }

 public:
  SynthProgram(const Engine& e, Synth& synth, size_t nInstruments, size_t nOutputs, int sampleRate, VoiceProgramOptions opt = VoiceProgramOptions())
      : eng_(e),
        nInstruments_(nInstruments),
        polyphony_(synth.getNumVoices()),
        nOut_(nOutputs),
        ctx_(0, nOutputs, sampleRate),
        prog_(e, nInstruments * (size_t)synth.getNumVoices(), &ctx_,
              [&synth](AudioContext* c) { synth.processVoice(0, c->getInputVoice(0), c->inputs, c->outputs, c); }, perInstrument(opt, synth.getNumVoices()))
  {
    for (int code : prog_.contextInputs())
      if (code != Capture::kBeatPhase) controllers_.push_back(code);
      else eng_.check(mlgpu_transport_create(e.handle(), nInstruments, 1, &transport_));  // reserved at the first process call
    wantOwnStream_ = opt.eventsOnOwnStream && !prog_.eventRowsInKernel() && controllers_.empty() && prog_.voiceRowMask() != 0;
    if (wantOwnStream_)
    {
      evEngine_.reset(new Engine(e.device(), +1));
      for (int i = 0; i < 2; ++i)
      {
        written_[i].reset(new Fence(*evEngine_));
        read_[i].reset(new Fence(e));
      }
    }
    eng_.check(mlgpu_events_create(eventsEngine().handle(), nInstruments, polyphony_, &ev_));
    eng_.check(mlgpu_events_set_sample_rate(ev_, (double)sampleRate));
    eng_.check(mlgpu_events_set_wanted_rows(ev_, prog_.voiceRowMask()));  // rows processVoice never reads are not made
    if (prog_.eventRowsInKernel()) eng_.check(mlgpu_graph_bind_events(prog_.graph(), ev_));
    for (const Capture::Tap& t : prog_.taps())
    {
      SignalProcessor::PublishedSignal* ps = synth.getPublishedSignals()[Path(t.name.c_str())].get();
      if (ps->handle_) throw Error(MLGPU_ERR_INVALID, "SynthProgram: this Synth's published signals already belong to another SynthProgram");
      eng_.check(mlgpu_published_signal_create(e.handle(), ps->maxFrames_, ps->maxVoices_, ps->channels_, ps->octavesDown_, &ps->handle_));
      published_.push_back(ps);
    }
  }
  SynthProgram(const SynthProgram&) = delete;
  SynthProgram& operator=(const SynthProgram&) = delete;
  ~SynthProgram()
  {
    if (ev_) mlgpu_events_destroy(ev_);
    if (transport_) mlgpu_transport_destroy(transport_);
    for (auto* ps : published_)
    {
      mlgpu_published_signal_destroy(ps->handle_);
      ps->handle_ = nullptr;
    }
  }
  // storePublishedSignal() in processVoice publishes the voices of ONE instrument (a display shows one): this one
  void setPublishedInstrument(size_t instrument)
  {
    if (instrument >= nInstruments_) throw Error(MLGPU_ERR_INVALID, "SynthProgram::setPublishedInstrument: no such instrument");
    publishedInstrument_ = instrument;
  }
  bool eventsOnOwnStream() const { return wantOwnStream_; }  // whether the option took effect (see VoiceProgramOptions)
  mlgpu_events* events() const { return ev_; }  // protocol, glide, drift, bend range: the mlgpu_events_set_* calls
  VoiceProgram& program() { return prog_; }
  void update() { prog_.update(); }  // the Synth's host-side numbers changed (coefficients, parameters): VoiceProgram::update()
  size_t voices() const { return nInstruments_ * (size_t)polyphony_; }

  void addInputEvent(size_t instrument, const Event& e)  // AudioContext::addInputEvent
  {
    mlgpu_event m{e.type, e.channel, e.sourceIdx, e.time, e.value1, e.value2};
    eng_.check(mlgpu_events_add_event(ev_, instrument, &m));
  }
  void clearInputEvents() { eng_.check(mlgpu_events_clear_events(ev_)); }
  // AudioContext::updateTime for one instrument's context, or for all of them (one host application behind the whole bank);
  // without effect when processVoice does not read ctx->getBeatPhase()
  void updateTime(size_t instrument, double ppqPos, double bpm, bool isPlaying, double sampleRate)
  {
    if (transport_) eng_.check(mlgpu_transport_set_time_and_rate(transport_, instrument, ppqPos, bpm, isPlaying ? 1 : 0, sampleRate));
  }
  void updateTime(double ppqPos, double bpm, bool isPlaying, double sampleRate) { updateTime(MLGPU_TRANSPORT_ALL, ppqPos, bpm, isPlaying, sampleRate); }
  mlgpu_transport* transport() const { return transport_; }

  // nVectors DSPVectors starting at frame startOffset of the current host block. mixed[c]: a signal of nInstruments
  // "voices" x nVectors vectors per output channel (the instruments' outputs).
  void process(size_t nVectors, int startOffset, const std::vector<DeviceSignal*>& mixed)
  {
    if (mixed.size() != nOut_) throw Error(MLGPU_ERR_INVALID, "SynthProgram::process: one mixed signal per output channel");
    if (nVectors > capacityT_)
    {
      rows_.clear();
      voiceOut_.clear();
      tapOut_.clear();
      for (size_t c = 0; c < prog_.tapChannels(); ++c) tapOut_.emplace_back(eng_, voices(), nVectors);
      rowsB_.clear();
      for (int r = 0; r < kNumVoiceOutputRows; ++r)
        rows_.emplace_back(eng_, (!prog_.eventRowsInKernel() && ((prog_.voiceRowMask() >> r) & 1u)) ? voices() : 1, nVectors);
      if (wantOwnStream_)
      {
        eng_.sync();  // nothing of the old signals is in flight on either stream
        evEngine_->sync();
        for (int r = 0; r < kNumVoiceOutputRows; ++r) rowsB_.emplace_back(eng_, ((prog_.voiceRowMask() >> r) & 1u) ? voices() : 1, nVectors);
      }
      for (size_t c = 0; c < nOut_; ++c) voiceOut_.emplace_back(eng_, voices(), nVectors);
      // the controllers' smoothers go on when only the reserved length changes
      if (!controllers_.empty()) eng_.check(mlgpu_events_watch_controllers(ev_, controllers_.data(), (int)controllers_.size(), nVectors));
      if (transport_) eng_.check(mlgpu_transport_reserve(transport_, nVectors));
      capacityT_ = nVectors;
    }
    const bool inKernel = prog_.eventRowsInKernel();
    const size_t set = wantOwnStream_ ? (blockCount_ & 1) : 0;
    std::vector<DeviceSignal>& rowSet = set ? rowsB_ : rows_;
    float* rowPtrs[kNumVoiceOutputRows];
    std::vector<const float*> pi;
    for (int r = 0; r < kNumVoiceOutputRows; ++r)
    {
      rowPtrs[r] = (!inKernel && ((prog_.voiceRowMask() >> r) & 1u)) ? rowSet[r].data() : nullptr;
      if (rowPtrs[r]) pi.push_back(rowPtrs[r]);
    }
    {
      size_t slot = 0;  // in the order processVoice first asked for them
      for (int code : prog_.contextInputs())
        pi.push_back(code == Capture::kBeatPhase ? mlgpu_transport_beat_phase(transport_) : mlgpu_events_controller_signal(ev_, (int)slot++));  // controllers: made by the events call below
    }
    if (transport_) eng_.check(mlgpu_transport_process(transport_, nVectors));
    std::vector<float*> po;
    for (auto& s : voiceOut_) po.push_back(s.data());
    for (auto& s : tapOut_) po.push_back(s.data());
    if (inKernel)  // the voice kernel walks the block's event records itself: pitch and gate never exist in memory
      eng_.check(mlgpu_graph_process_events(prog_.graph(), nVectors, startOffset, pi.data(), MLGPU_LAYOUT_QUAD, nullptr, po.data(), MLGPU_LAYOUT_QUAD));
    else if (wantOwnStream_)
    {
      // the events kernel of this block may start as soon as the voice kernel of the block before last has read this row set -
      // i.e. while the voice kernel of the previous block is still running on the other stream
      read_[set]->awaitedBy(*evEngine_);
      evEngine_->check(mlgpu_events_process(ev_, nVectors, startOffset, rowPtrs, MLGPU_LAYOUT_QUAD));
      written_[set]->signalFrom(*evEngine_);
      written_[set]->awaitedBy(eng_);
      eng_.check(mlgpu_graph_process(prog_.graph(), nVectors, pi.data(), MLGPU_LAYOUT_QUAD, po.data(), MLGPU_LAYOUT_QUAD));
      read_[set]->signalFrom(eng_);
      ++blockCount_;
    }
    else
    {
      eng_.check(mlgpu_events_process(ev_, nVectors, startOffset, rowPtrs, MLGPU_LAYOUT_QUAD));
      eng_.check(mlgpu_graph_process(prog_.graph(), nVectors, pi.data(), MLGPU_LAYOUT_QUAD, po.data(), MLGPU_LAYOUT_QUAD));
    }
    // SignalProcessor::storePublishedSignal for each voice of the published instrument in rotation, vector by vector
    size_t tapCh = 0;
    for (size_t i = 0; i < published_.size(); ++i)
    {
      std::vector<const float*> ch;
      for (size_t c = 0; c < prog_.taps()[i].nodes.size(); ++c) ch.push_back(tapOut_[tapCh++].data());
      eng_.check(mlgpu_published_signal_write(published_[i]->handle_, nVectors, ch.data(), MLGPU_LAYOUT_QUAD, voices(),
                                              publishedInstrument_ * (size_t)polyphony_, (size_t)polyphony_));
    }
    for (size_t c = 0; c < nOut_; ++c)
      eng_.check(mlgpu_mixdown_groups(eng_.handle(), voiceOut_[c].data(), MLGPU_LAYOUT_QUAD, nInstruments_, (size_t)polyphony_, nVectors,
                                      mixed[c]->data(), mixed[c]->layout()));
  }
};
}  // namespace gpu
}  // namespace ml
