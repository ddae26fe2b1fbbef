// graph.hip — run-time defined DSP graphs ("procs") compiled to ONE fused gfx950 kernel.
//
// The reference's dynamic-graph layer is a stub (source/procs/MLProcMultiply.cpp: named
// inputs/outputs/params, a process() that combines mldsp.h objects, registration by name),
// so this is the engine's own executor for BASELINE configs[4]: a DAG whose nodes are the same
// processors (MLGPU_PROC_*) and stateless ops (MLGPU_OP_*) the banks use, with named nodes.
//
// Execution model: the graph is translated to HIP source that instantiates the hand-written device
// building blocks (mldsp_procs.hpp / mldsp_ops.hpp) in topological order inside the voice-bank loop
// (one lane per voice, state in registers, one 16-byte access per lane per quad) and compiled for
// gfx950 with hiprtc. Every edge of the graph is a register; only graph inputs and outputs touch
// HBM. Identical graphs share one compiled module per process. The same generator produces fused
// kernels for processor chains that have no ahead-of-time instantiation (mlgpu_jit_chain).
#include <hip/hiprtc.h>  // (types and enumerators only: the library is looked up at run time, see Hiprtc below)
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <fcntl.h>
#include <sys/file.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>

#include <algorithm>
#include <atomic>
#include <thread>
#include <functional>
#include <map>
#include <mutex>
#include <new>
#include <set>
#include <fstream>
#include <sstream>

#include "mlgpu_internal.hpp"

extern const char mlgpu_device_source_hash_str[];
struct mlgpu_graph;
static std::mutex g_boundMutex;
static std::set<mlgpu_graph*> g_boundGraphs;  // graphs with an events object bound (mlgpu_graph_bind_events)
extern const int mlgpu_embedded_count;
extern const char* const mlgpu_embedded_names[];
extern const char* const mlgpu_embedded_sources[];

namespace
{
struct CompiledModule
{
  hipModule_t module{nullptr};
  std::map<std::string, hipFunction_t> fns;
};

std::mutex g_cacheMutex;
std::map<std::string, CompiledModule> g_cache;  // key: device id + source

// The options every run-time kernel is compiled with (part of the disk cache's key): the ahead-of-time build's own
// (csrc/Makefile) apart from its scheduling strategy. MLGPU_JIT_EXTRA_OPTS adds space-separated options for A/B
// measurements, e.g. "-mllvm -amdgpu-sched-strategy=max-ilp" (profiles/archive/r03_jit_maxilp.txt: what it does to config 5).
const std::vector<std::string>& jitOptions()
{
  static const std::vector<std::string> opts = [] {
    std::vector<std::string> o = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize"};
    if (const char* extra = getenv("MLGPU_JIT_EXTRA_OPTS"))
    {
      std::istringstream in(extra);
      std::string tok;
      while (in >> tok) o.push_back(tok);
    }
    return o;
  }();
  return opts;
}

// hiprtc, looked up at RUN TIME (round 6): libmlgpu.so does not link it, so an installation without the compiler still loads and
// runs every ahead-of-time kernel, and every generated one whose code it is given (the disk cache, mlgpu_jit_cache_import). A
// graph or chain that needs a compile there fails with MLGPU_ERR_UNSUPPORTED and says why. MLGPU_HIPRTC=off: behave as if the
// library were absent (what tests/test_abi.py uses); MLGPU_HIPRTC=<path>: that library.
struct Hiprtc
{
  void* lib{nullptr};
  decltype(&::hiprtcCreateProgram) createProgram{nullptr};
  decltype(&::hiprtcCompileProgram) compileProgram{nullptr};
  decltype(&::hiprtcGetProgramLogSize) getProgramLogSize{nullptr};
  decltype(&::hiprtcGetProgramLog) getProgramLog{nullptr};
  decltype(&::hiprtcGetCodeSize) getCodeSize{nullptr};
  decltype(&::hiprtcGetCode) getCode{nullptr};
  decltype(&::hiprtcDestroyProgram) destroyProgram{nullptr};
  decltype(&::hiprtcGetErrorString) getErrorString{nullptr};
  decltype(&::hiprtcVersion) version{nullptr};
  std::string why;
  Hiprtc()
  {
    const char* knob = getenv("MLGPU_HIPRTC");
    for (const char* name : {knob && strcmp(knob, "off") ? knob : "libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"})
    {
      lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (lib) break;
    }
    if (!lib)
    {
      why = std::string("libhiprtc.so cannot be loaded (") + (dlerror() ? dlerror() : "not found") + ")";
      return;
    }
#define MLGPU_RTC_SYM(member, symbol) member = (decltype(member))dlsym(lib, #symbol)
    MLGPU_RTC_SYM(createProgram, hiprtcCreateProgram);
    MLGPU_RTC_SYM(compileProgram, hiprtcCompileProgram);
    MLGPU_RTC_SYM(getProgramLogSize, hiprtcGetProgramLogSize);
    MLGPU_RTC_SYM(getProgramLog, hiprtcGetProgramLog);
    MLGPU_RTC_SYM(getCodeSize, hiprtcGetCodeSize);
    MLGPU_RTC_SYM(getCode, hiprtcGetCode);
    MLGPU_RTC_SYM(destroyProgram, hiprtcDestroyProgram);
    MLGPU_RTC_SYM(getErrorString, hiprtcGetErrorString);
    MLGPU_RTC_SYM(version, hiprtcVersion);
#undef MLGPU_RTC_SYM
    if (!createProgram || !compileProgram || !getProgramLogSize || !getProgramLog || !getCodeSize || !getCode || !destroyProgram || !getErrorString)
    {
      why = "libhiprtc.so lacks an entry point this library uses";
      dlclose(lib);
      lib = nullptr;
    }
  }
};
// nullptr (and `why`) where the compiler is not there - or a test says so
const Hiprtc* hiprtc(std::string* why = nullptr)
{
  static const Hiprtc rtc;
  const char* knob = getenv("MLGPU_HIPRTC");
  if (knob && !strcmp(knob, "off"))
  {
    if (why) *why = "run-time compilation is switched off (MLGPU_HIPRTC=off)";
    return nullptr;
  }
  if (!rtc.lib)
  {
    if (why) *why = rtc.why;
    return nullptr;
  }
  return &rtc;
}

// compile `source` for gfx950 and load it on the current device; returns nullptr and fills `log` on failure
bool compileToCode(const std::string& source, std::vector<char>& code, std::string& log)
{
  hiprtcProgram prog;
  code.clear();
  std::string why;
  const Hiprtc* rtc = hiprtc(&why);
  if (!rtc)
  {
    log = "this kernel is not in the memory or disk cache and " + why + ": compile it where hiprtc is installed and bring its code along (mlgpu_jit_cache_export / _import)";
    return false;
  }
  if (rtc->createProgram(&prog, source.c_str(), "mlgpu_jit.hip", mlgpu_embedded_count, (const char**)mlgpu_embedded_sources,
                          (const char**)mlgpu_embedded_names) != HIPRTC_SUCCESS)
  {
    log = "hiprtcCreateProgram failed";
    return false;
  }
  std::vector<const char*> opts;
  for (const std::string& o : jitOptions()) opts.push_back(o.c_str());
  const hiprtcResult r = rtc->compileProgram(prog, (int)opts.size(), opts.data());
  size_t logSize = 0;
  rtc->getProgramLogSize(prog, &logSize);
  if (logSize > 1)
  {
    log.resize(logSize);
    rtc->getProgramLog(prog, &log[0]);
  }
  size_t codeSize = 0;
  if (r == HIPRTC_SUCCESS) rtc->getCodeSize(prog, &codeSize);
  if (codeSize)
  {
    code.resize(codeSize);
    rtc->getCode(prog, code.data());
  }
  rtc->destroyProgram(&prog);
  if (r != HIPRTC_SUCCESS && log.empty()) log = rtc->getErrorString(r);
  return r == HIPRTC_SUCCESS && codeSize > 0;
}

// hiprtc results by source. Two levels: in memory (identical graphs and the size probe of graph_compile compile once per
// process) and on disk (a process that starts again - a plug-in host reloading, the next benchmark run - finds the code
// object of every graph it has built before and skips hiprtc, which takes 0.3-2 s per kernel). The disk key is a hash of
// everything that decides the code object: the generated source, the compile options, every embedded device header and
// the hiprtc version. Files are written to a temporary name and renamed, so concurrent processes (one rank per GPU) can
// share the directory. MLGPU_CACHE_DIR names it (default $XDG_CACHE_HOME/mlgpu or ~/.cache/mlgpu); MLGPU_CACHE_DIR=off
// disables the disk level.
std::mutex g_codeMutex;
std::map<std::string, std::vector<char>> g_codeCache;
struct JitStats
{
  uint64_t compiles{0}, diskHits{0}, memoryHits{0}, diskWrites{0};
  double compileSeconds{0}, diskLoadSeconds{0};
} g_jitStats;


uint64_t fnv1a(uint64_t h, const void* data, size_t n)
{
  const unsigned char* p = (const unsigned char*)data;
  for (size_t i = 0; i < n; ++i) h = (h ^ p[i]) * 0x100000001b3ull;
  return h;
}

// The cache directory, or "" when the disk level is off or the directory cannot be trusted: code objects are loaded into
// the GPU as they are, so the directory must belong to this user and be writable by nobody else (a directory another user
// can write to would let them choose the code this process runs).
std::string cacheDir()
{
  const char* d = getenv("MLGPU_CACHE_DIR");
  if (d && !strcmp(d, "off")) return "";
  std::string dir;
  if (d && *d)
    dir = d;
  else if (const char* x = getenv("XDG_CACHE_HOME"))
    dir = std::string(x) + "/mlgpu";
  else if (const char* h = getenv("HOME"))
    dir = std::string(h) + "/.cache/mlgpu";
  else
    return "";
  // mkdir -p; what we create is ours alone
  for (size_t i = 1; i <= dir.size(); ++i)
    if (i == dir.size() || dir[i] == '/') mkdir(dir.substr(0, i).c_str(), 0700);
  struct stat st;
  if (stat(dir.c_str(), &st) != 0 || !S_ISDIR(st.st_mode)) return "";
  if (st.st_uid != geteuid() || (st.st_mode & (S_IWGRP | S_IWOTH)))
  {
    fprintf(stderr, "mlgpu: kernel cache directory %s is not owned by this user or is writable by others: disk cache off\n", dir.c_str());
    return "";
  }
  return dir;
}

// Everything that decides a code object besides the generated source: compile options, the fingerprint of the device
// headers of this build (embed.py: the headers hiprtc is given are part of it), the HIP runtime / hiprtc versions with
// their patch level, and the library's ABI version.
const std::string& cacheContext()
{
  static const std::string ctx = [] {
    std::string c = "mlgpu-kernel-cache 2\n";
    for (const std::string& o : jitOptions()) c += o + " ";
    c += "\ndevice-sources " + std::string(mlgpu_device_source_hash_str);
    int major = 0, minor = 0, runtime = 0, driver = 0;
    if (const Hiprtc* rtc = hiprtc())
      if (rtc->version) rtc->version(&major, &minor);  // (0.0 without the compiler: such a process only ever READS code it is given)
    if (hipRuntimeGetVersion(&runtime) != hipSuccess) runtime = -1;
    if (hipDriverGetVersion(&driver) != hipSuccess) driver = -1;
    c += "\nhiprtc " + std::to_string(major) + "." + std::to_string(minor) + " runtime " + std::to_string(runtime) + " driver " + std::to_string(driver);
#ifdef HIP_VERSION_GITHASH
    c += std::string(" built-with ") + HIP_VERSION_GITHASH;
#endif
    c += "\nabi " + std::to_string(MLGPU_ABI_VERSION) + "\n";
    return c;
  }();
  return ctx;
}

std::string cacheFile(const std::string& source)
{
  // (looked up again whenever MLGPU_CACHE_DIR changes: a host - or a test - may switch the disk level off after the first kernel)
  static std::mutex m;
  static std::string dirFor, dirValue;
  static bool dirKnown = false;
  std::string dir;
  {
    std::lock_guard<std::mutex> lock(m);
    const char* envNow = getenv("MLGPU_CACHE_DIR");
    const std::string key = envNow ? envNow : "";
    if (!dirKnown || key != dirFor)
    {
      dirValue = cacheDir();
      dirFor = key;
      dirKnown = true;
    }
    dir = dirValue;
  }
  if (dir.empty()) return "";
  const std::string& ctx = cacheContext();
  uint64_t h = 0xcbf29ce484222325ull;
  h = fnv1a(h, ctx.data(), ctx.size());
  h = fnv1a(h, source.data(), source.size());
  char name[64];
  snprintf(name, sizeof(name), "/%016llx-%zu.co", (unsigned long long)h, source.size());
  return dir + name;
}

// A cache file = header line "MLGPUCO2 <context bytes> <source bytes> <code bytes>\n", the context, the generated source,
// the code object. The file name is only a 64-bit hash: a file is used when its context and its source are byte for byte
// the ones asked for, so neither a hash collision nor another compiler / library build can hand back a different kernel.
bool readCacheFile(const std::string& path, const std::string& source, std::vector<char>& code)
{
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  bool ok = false;
  char magic[16] = {0};
  unsigned long long nCtx = 0, nSrc = 0, nCode = 0;
  const std::string& ctx = cacheContext();
  if (fscanf(f, "%15s %llu %llu %llu", magic, &nCtx, &nSrc, &nCode) == 4 && fgetc(f) == '\n' && !strcmp(magic, "MLGPUCO2") &&
      nCtx == ctx.size() && nSrc == source.size() && nCode > 64 && nCode < (1ull << 30))
  {
    std::string gotCtx(nCtx, '\0'), gotSrc(nSrc, '\0');
    code.resize((size_t)nCode);
    ok = fread(&gotCtx[0], 1, nCtx, f) == nCtx && fread(&gotSrc[0], 1, nSrc, f) == nSrc && fread(code.data(), 1, (size_t)nCode, f) == (size_t)nCode &&
         fgetc(f) == EOF && gotCtx == ctx && gotSrc == source && !memcmp(code.data(), "\177ELF", 4);
  }
  fclose(f);
  if (!ok) code.clear();
  return ok;
}

bool writeCacheFile(const std::string& path, const std::string& source, const std::vector<char>& code)
{
  // a temporary file of our own in the same directory (mkstemp: unique whatever shares the directory - other processes,
  // containers with the same pids, hosts on a network file system), then an atomic rename
  std::string tmp = path + ".XXXXXX";
  const int fd = mkstemp(&tmp[0]);
  if (fd < 0) return false;
  FILE* f = fdopen(fd, "wb");
  if (!f)
  {
    close(fd);
    remove(tmp.c_str());
    return false;
  }
  const std::string& ctx = cacheContext();
  bool ok = fprintf(f, "MLGPUCO2 %zu %zu %zu\n", ctx.size(), source.size(), code.size()) > 0;
  ok = ok && fwrite(ctx.data(), 1, ctx.size(), f) == ctx.size() && fwrite(source.data(), 1, source.size(), f) == source.size() &&
       fwrite(code.data(), 1, code.size(), f) == code.size();
  ok = (fclose(f) == 0) && ok;
  if (ok && rename(tmp.c_str(), path.c_str()) == 0) return true;
  remove(tmp.c_str());
  return false;
}

bool compileToCode(const std::string& source, std::vector<char>& code, std::string& log);

bool getCode(const std::string& source, std::vector<char>& code, std::string& log)
{
  {
    std::lock_guard<std::mutex> lock(g_codeMutex);
    auto it = g_codeCache.find(source);
    if (it != g_codeCache.end())
    {
      code = it->second;
      ++g_jitStats.memoryHits;
      return true;
    }
  }
  // One build at a time: the host threads of a DeviceGroup ask for the same kernels at the same moment, and the second one
  // should find the first one's result instead of running hiprtc again beside it.
  static std::mutex buildMutex;
  std::lock_guard<std::mutex> building(buildMutex);
  {
    std::lock_guard<std::mutex> lock(g_codeMutex);
    auto it = g_codeCache.find(source);
    if (it != g_codeCache.end())
    {
      code = it->second;
      ++g_jitStats.memoryHits;
      return true;
    }
  }
  const std::string path = cacheFile(source);
  bool fromDisk = false;
  // One build per MACHINE too: the ranks of a multi-GPU job start together and all ask for the same kernel. Whoever gets
  // the advisory lock on <entry>.lock first compiles and writes the entry; the others block in flock(), then find it. The
  // kernel drops the lock when its holder dies, so a killed rank cannot strand the rest.
  struct FileLock
  {
    int fd{-1};
    explicit FileLock(const std::string& p)
    {
      if (p.empty()) return;
      fd = open((p + ".lock").c_str(), O_CREAT | O_RDWR | O_CLOEXEC, 0600);
      if (fd >= 0 && flock(fd, LOCK_EX) != 0)
      {
        close(fd);
        fd = -1;
      }
    }
    ~FileLock()
    {
      if (fd >= 0)
      {
        flock(fd, LOCK_UN);
        close(fd);
      }
    }
  } entryLock(path);
  if (!path.empty())
  {
    const auto t0 = std::chrono::steady_clock::now();
    fromDisk = readCacheFile(path, source, code);  // anything else under that name (a truncated write, another build's file) is ignored and rebuilt
    if (fromDisk)
    {
      std::lock_guard<std::mutex> lock(g_codeMutex);
      ++g_jitStats.diskHits;
      g_jitStats.diskLoadSeconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
  }
  if (!fromDisk)
  {
    const auto t0 = std::chrono::steady_clock::now();
    if (!compileToCode(source, code, log)) return false;
    {
      std::lock_guard<std::mutex> lock(g_codeMutex);
      ++g_jitStats.compiles;
      g_jitStats.compileSeconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    if (!path.empty() && writeCacheFile(path, source, code))
    {
      std::lock_guard<std::mutex> lock(g_codeMutex);
      ++g_jitStats.diskWrites;
    }
  }
  std::lock_guard<std::mutex> lock(g_codeMutex);
  g_codeCache[source] = code;
  return true;
}

CompiledModule* compileAndLoad(int device, const std::string& source, std::string& log)
{
  const std::string key = std::to_string(device) + "\n" + source;
  {
    std::lock_guard<std::mutex> lock(g_cacheMutex);
    auto it = g_cache.find(key);
    if (it != g_cache.end()) return &it->second;
  }
  // hiprtc (seconds, on a compile job's worker thread) runs OUTSIDE the module cache's lock: a thread that only looks a loaded module
  // up - an autotune trial inside graph_process, a bank's chain, another graph - never waits for someone else's compile
  std::vector<char> code;
  if (!getCode(source, code, log)) return nullptr;

  std::lock_guard<std::mutex> lock(g_cacheMutex);
  auto it = g_cache.find(key);  // (another thread may have loaded the same source meanwhile)
  if (it != g_cache.end()) return &it->second;
  CompiledModule cm;
  const hipError_t e = hipModuleLoadData(&cm.module, code.data());
  if (e != hipSuccess)
  {
    log = std::string("hipModuleLoadData: ") + hipGetErrorString(e);
    return nullptr;
  }
  return &(g_cache[key] = cm);
}

// compile only (no device needed): used by mlgpu_jit_selftest
bool compileOnly(const std::string& source, std::string& log)
{
  std::vector<char> code;
  return compileToCode(source, code, log);
}

hipFunction_t getFunction(CompiledModule* cm, const char* name, std::string& log)
{
  auto it = cm->fns.find(name);
  if (it != cm->fns.end()) return it->second;
  hipFunction_t f = nullptr;
  const hipError_t e = hipModuleGetFunction(&f, cm->module, name);
  if (e != hipSuccess)
  {
    log = std::string("hipModuleGetFunction(") + name + "): " + hipGetErrorString(e);
    return nullptr;
  }
  cm->fns[name] = f;
  return f;
}

template <class ARGS>
hipError_t launchJit(hipFunction_t fn, const ARGS& args, size_t V, hipStream_t stream)
{
  ARGS copy = args;
  size_t size = sizeof(ARGS);
  void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &copy, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
  const unsigned blocks = (unsigned)((V + 255) / 256);
  return hipModuleLaunchKernel(fn, blocks, 1, 1, 256, 1, 1, 0, stream, nullptr, config);
}

enum NodeType
{
  NODE_INPUT = 0,
  NODE_PARAM = 1,
  NODE_CONST = 2,
  NODE_PROC = 3,
  NODE_OP = 4,
  NODE_CONTROL = 5,  // streamed, one float per DSPVector per voice
  NODE_VOP = 6,      // index-dependent vector generator (columnIndex, rangeOpen, ...)
  NODE_ROUTE = 7,    // multiplex / demultiplex (MLDSPRouting.h); in[0] is the selector
  NODE_FEEDBACK = 8,  // value of another node one DSPVector ago (64 state words per voice)
  NODE_EVENT_ROW = 9  // a row of the bound EventsToSignals object, computed in this kernel (slot: 0 pitch, 1 gate)
};

// how often a node's value changes: per voice (params, consts and ops on them), per DSPVector (controls and
// ops on them), per sample. Decides where the generated code evaluates it.
enum Rate
{
  RATE_VOICE = 0,
  RATE_VECTOR = 1,
  RATE_AUDIO = 2
};

struct Node
{
  int type;
  int kind;  // proc kind or op
  std::vector<int> in;
  std::string name;
  float value{0.f};
  int slot{0};            // input index / param index / control index; demultiplex: output index
  int nOut{0};            // demultiplex: number of outputs
  size_t ringLen{0};      // delay nodes: floats per ring (power of two), 0 = not set
  int ringSlot{0};        // delay nodes: index of this node's first ring among all rings of the graph (LDS windows)
  int earlySlot{-1};      // ring layout 0 with early reads: this node's first 256-byte landing slot in the wavefront's LDS
  int earlyPending{0};    //   and the loads the kernel issues right after this node's (RingCore::earlyWait), set by the generator
  size_t memOff{0};       // delay nodes: first ring at d_mem + memOff * V
  int fbSource{-1};       // feedback nodes: the node whose value is stored for the next vector
  int rate{RATE_AUDIO};
  int cOff{0}, sOff{0}, nc{0}, ns{0};
  int region{-1};         // the rate region whose function this node belongs to (-1: the outer graph)
  std::vector<uint32_t> table;  // MLGPU_VOP_TABLE: the 64 floats of a constant DSPVector (bit patterns)
  int role{0};            // ROLE_REGION_IN: HalfBandFilter carrying an outer node into region `region`;
                          // ROLE_REGION_OUT: HalfBandFilter bringing region `slot`'s result back (an outer node)
};

enum { ROLE_NONE = 0, ROLE_REGION_IN = 1, ROLE_REGION_OUT = 2 };

// Upsample2xFunction / Downsample2xFunction (MLDSPFunctional.h:114-213) with fn written out as nodes
struct Region
{
  int kind{0};            // mlgpu_region
  int parent{-1};         // the region this one is nested in (-1: the outer graph)
  std::vector<int> ins;   // ROLE_REGION_IN nodes
  int result{-1};         // fn's return value (a node of the region)
  int out{-1};            // ROLE_REGION_OUT node
};

int opArity(int op) { return op >= 64 ? 3 : (op >= 32 ? 2 : 1); }
bool opKnown(int op)
{
  return (op >= 0 && op <= MLGPU_OP_PHASOR_TO_SINE) || (op >= MLGPU_OP_ADD && op <= MLGPU_OP_PHASOR_TO_SAW) ||
         (op >= MLGPU_OP_LERP && op <= MLGPU_OP_PHASOR_TO_PULSE);
}

std::string floatLiteral(float f)
{
  uint32_t u;
  memcpy(&u, &f, 4);
  char buf[48];
  snprintf(buf, sizeof(buf), "u2f(0x%08xu)", u);  // exact bits, no decimal round trip
  return buf;
}
}  // namespace

struct mlgpu_graph
{
  mlgpu_engine* e{nullptr};
  size_t V{0};
  std::vector<Node> nodes;
  std::vector<int> outputs;
  int inputGroup[MLGPU_GRAPH_MAX_INPUTS] = {};  // > 1: the input has one row per that many adjacent voices (mlgpu_graph_set_input_group)
  bool outputMixShard[MLGPU_GRAPH_MAX_OUTPUTS] = {false, false, false, false, false, false, false, false};  // ... handed over as the rows of a SHARD (graph_set_output_mixdown(.., 2): mlgpu_mixdown_shard_rows(V) rows for mlgpu_mixdown_finish)
  bool outputMix[MLGPU_GRAPH_MAX_OUTPUTS] = {false, false, false, false, false, false, false, false};  // the output is the mixdown of all voices (graph_set_output_mixdown)
  int outputGroup[MLGPU_GRAPH_MAX_OUTPUTS] = {0, 0, 0, 0, 0, 0, 0, 0};  // > 0: the output is the in-order sum of groups of that many adjacent voices
  int nInputs{0}, nParams{0}, nControls{0}, NC{0}, NS{0};
  bool compiled{false};
  bool hasImpulse{false};
  bool strictSvf{false};  // the engine's mode when the graph was made (mlgpu_engine_set_strict_svf)
  std::string source, log;
  hipFunction_t fn{nullptr};
  float* d_coeffs{nullptr};
  uint32_t* d_state{nullptr};
  float* d_params{nullptr};
  float* d_consts{nullptr};      // live constants: [nConsts] floats
  int nConsts{0};
  bool liveConsts{false};        // const nodes read d_consts instead of being literals of the generated code
  float* d_mem{nullptr};
  std::vector<char> emitted;     // mlgpu_graph_emit: the gfx950 code object
  size_t memFloatsPerVoice{0};
  bool windowedRings{false};     // rings as [block][chunk][lane][8] behind LDS windows (mlgpu_graph_set_delay_layout)
  bool fbAhead{true};            // kept DSPVectors (feedback nodes) are fetched two quads ahead; MLGPU_GRAPH_FB_AHEAD=0 for A / B
  bool transposedIfPossible{false};  // graph_set_delay_layout(3)
  bool rowAddr32{false};         // layout 0: ring rows behind 32-bit offsets from a wave-uniform base where a ring allows it (VoiceMem::ringPtr); set at compile
  bool earlyRows{false};         // layout 0: the ring reads of the outer graph's delay nodes issued ahead by LDS-DMA (RingCore::readEarly); set at compile
  int earlySlots{0};             // their landing slots per wavefront
  bool sectorRings{false};       // layout 4: layout 1's memory, no LDS, trips of 8 samples with every ring's loads in the trip's prologue (implies windowedRings)
  bool transposedRings{false};   // layout 2: [block][chunk][lane][16], every global access a 64-byte piece made by four lanes, on a wave-uniform clock (implies windowedRings)
  int totalRings{0};
  size_t memVoices() const { return windowedRings ? ((V + 255) & ~(size_t)255) : V; }  // voices the ring memory is laid out for
  std::vector<Region> regions;
  int openRegion{-1};            // between graph_begin_region and graph_end_region
  size_t vectorCount{0};         // DSPVectors processed since the last clear (GraphArgs::t0)
  int voicesPerLane{0};          // 0 = choose at compile (graphVoicesPerLane); 1 or 2 = forced
  int compiledVoicesPerLane{1};
  int unrollQ{1};                // quads per trip of the sample loop
  int oscTripQ{2};               // quads per trip of the oscillators' sparse polyBLEP (0: per sample; mldsp_procs.hpp: trip_u)
  std::string waveClockPath;     // MLGPU_GRAPH_WAVE_CLOCK=<file>: every wavefront of a launch stamps its start and end; the last launch's table is written there
  unsigned long long* d_waveClock{nullptr};
  bool lockOscillators{true};    // a SawGen and a PulseGen on one frequency node share their trip while their phase counters are equal; MLGPU_GRAPH_LOCK_OSC=0 for A / B
  int takeTurns{2};              // the wavefronts of a SIMD rotate through the priority levels (mldsp_math.hpp): 0 off, 1 by their own progress, 2 by the shared clock; MLGPU_GRAPH_TURNS
  int turnClockShift{13};        // a turn of the clock form lasts 2^shift ticks of 10 ns (82 us: the best of 2^7 .. 2^18, profiles/r04_take_turns.txt); MLGPU_GRAPH_TURN_CLOCK
  int prefetchQ{1};              // streamed inputs are loaded one quad ahead of their use (0: where they are used)
  std::string lastError;         // mlgpu_graph_last_error
  mlgpu_events* events{nullptr}; // mlgpu_graph_bind_events: the object the NODE_EVENT_ROW nodes read
  bool hasEventRows{false};
  // mlgpu_graph_compile_async: code generation and hiprtc on a thread of the library's, off the caller's (audio) thread. While the
  // job is in flight the graph belongs to that thread: every other call on the graph answers MLGPU_ERR_BUSY (or refuses as it would
  // for a graph that is not compiled yet) without touching it.
  struct CompileJob
  {
    std::thread th;
    std::atomic<bool> done{false};
    int status{MLGPU_OK};
    std::string error;
  };
  CompileJob* job{nullptr};
  bool aotDone{false};           // an engine-less graph whose ahead-of-time compile has been collected (mlgpu_graph_compile_poll keeps answering MLGPU_OK)
  int eventOffset{-1};           // frame offset of the block being processed (mlgpu_graph_process_events), -1: none pending
  int minWaves{0};               // wavefronts per SIMD the kernel's register budget must allow (0: the compiler's choice), generateBudgeted
  // Online tuning (mlgpu_graph_set_autotune): every variant (voices per lane x quads per trip) computes the same bits from
  // the same state arrays, so the first process calls simply take turns, are timed, and the fastest one stays.
  struct Variant
  {
    int vl{1}, unroll{1};
    hipFunction_t fn{nullptr};
    bool failed{false};
    int runs{0};
    float bestMs{1e30f};
  };
  bool autotune{false}, tuned{false};
  std::vector<Variant> variants;
  hipEvent_t tuneEv0{nullptr}, tuneEv1{nullptr};
  int activeVl{1};               // of the kernel in `fn`
  int inLayoutOverride[MLGPU_GRAPH_MAX_INPUTS];  // -1: none
  mlgpu_graph()
  {
    for (int& x : inLayoutOverride) x = -1;
  }
};

namespace
{
static thread_local bool t_compileWorker = false;  // this thread is a graph's compile job: failures stay with the graph
int gfail(mlgpu_graph* g, int status, const std::string& what)
{
  // a graph whose compile is in flight belongs to that thread: a call that fails on it meanwhile writes nothing and says busy
  if (g && g->job && !t_compileWorker) return MLGPU_ERR_BUSY;
  if (g && g->e && !t_compileWorker) g->e->lastError = what;
  if (g) g->lastError = what;  // a graph created without an engine (offline code generation) has nowhere else to keep it
  return status;
}

// the C++ expression of node i for lane-group l (its inputs are the locals n<j>_<l>)
// Inside a rate region the values of the region's nodes carry the phase suffix `ph` ("a" / "b" for the two samples an
// Upsample2x region makes per outer sample) and `idx` is the sample index inside fn's own DSPVector.
// number of Upsample2x regions on the way from the outer graph down to region r (each adds one phase letter to a value's name)
int upDepth(const mlgpu_graph* g, int r)
{
  int d = 0;
  for (; r >= 0; r = g->regions[(size_t)r].parent) d += (g->regions[(size_t)r].kind == MLGPU_REGION_UPSAMPLE_2X);
  return d;
}

// clamp(x, lo, hi) whose bounds are literal constants of the kernel (not NaN, not zero, lo <= hi) and whose x is produced by
// an arithmetic instruction - a node that can never hand a signaling NaN on: then two hardware instructions give what the
// six of the general form do (clamp_const_bounds, mldsp_math.hpp). Inputs, parameters, feedback vectors, delay lines,
// selects and the bit-twiddling approximations carry raw bit patterns and keep the general form.
bool clampHasConstBounds(const mlgpu_graph* g, const Node& n)
{
  if (n.in.size() != 3 || g->liveConsts) return false;
  const Node &x = g->nodes[(size_t)n.in[0]], &lo = g->nodes[(size_t)n.in[1]], &hi = g->nodes[(size_t)n.in[2]];
  if (lo.type != NODE_CONST || hi.type != NODE_CONST) return false;
  if (!(lo.value <= hi.value) || lo.value == 0.f || hi.value == 0.f) return false;  // (a NaN bound fails the comparison)
  if (x.type == NODE_OP)
    switch (x.kind)
    {
      case MLGPU_OP_ADD: case MLGPU_OP_SUBTRACT: case MLGPU_OP_MULTIPLY: case MLGPU_OP_DIVIDE: case MLGPU_OP_LERP: case MLGPU_OP_INVERSE_LERP: return true;
      default: return false;
    }
  if (x.type == NODE_PROC)
    switch (x.kind)
    {
      case MLGPU_PROC_SINE_GEN: case MLGPU_PROC_SAW_GEN: case MLGPU_PROC_PULSE_GEN: case MLGPU_PROC_NOISE_GEN:
      case MLGPU_PROC_LOPASS: case MLGPU_PROC_HIPASS: case MLGPU_PROC_BANDPASS: case MLGPU_PROC_LO_SHELF: case MLGPU_PROC_HI_SHELF: case MLGPU_PROC_BELL:
      case MLGPU_PROC_ONE_POLE: case MLGPU_PROC_DC_BLOCKER: case MLGPU_PROC_INTEGRATOR: case MLGPU_PROC_DIFFERENTIATOR: case MLGPU_PROC_GAIN:
        return true;  // every output sample is the result of an add / sub / mul / fma
      default: return false;
    }
  return false;
}

// A SawGen / PulseGen of the outer graph whose frequency (and width) are per voice, not per sample: its samples are made a trip of
// oscTripQ quads at a time (Proc<>::trip_u: the polyBLEP corrections once per zone per trip) into registers the sample loop reads.
static bool isOscTrip(const mlgpu_graph* g, const Node& n)
{
  if (g->oscTripQ <= 0 || n.type != NODE_PROC || n.region >= 0 || n.rate != RATE_AUDIO) return false;
  if (n.kind != MLGPU_PROC_SAW_GEN && n.kind != MLGPU_PROC_PULSE_GEN) return false;
  if (n.in.empty() || g->nodes[n.in[0]].rate != RATE_VOICE) return false;
  return n.kind == MLGPU_PROC_SAW_GEN || n.in.size() == 1 || g->nodes[n.in[1]].rate == RATE_VOICE;
}
// The PulseGen trip node on the same frequency node as SawGen trip node i (-1: none): the pair is run by trip_locked when, at the
// start of a launch, every lane of the wavefront has the two phase counters equal (mldsp_procs.hpp).
static int lockedPartner(const mlgpu_graph* g, size_t i)
{
  const Node& n = g->nodes[i];
  if (!g->lockOscillators || !isOscTrip(g, n) || n.kind != MLGPU_PROC_SAW_GEN) return -1;
  for (size_t j = 0; j < g->nodes.size(); ++j)
  {
    const Node& m = g->nodes[j];
    if (isOscTrip(g, m) && m.kind == MLGPU_PROC_PULSE_GEN && m.in[0] == n.in[0])
    {
      // the first saw on that frequency takes the first pulse on it
      for (size_t k = 0; k < i; ++k)
        if (isOscTrip(g, g->nodes[k]) && g->nodes[k].kind == MLGPU_PROC_SAW_GEN && g->nodes[k].in[0] == n.in[0]) return -1;
      return (int)j;
    }
  }
  return -1;
}
static bool hasOscTrips(const mlgpu_graph* g)
{
  for (const Node& n : g->nodes)
    if (isOscTrip(g, n)) return true;
  return false;
}

// The same pairing for a STREAMED frequency (the instrument bank's voice: pitch signal -> exp2Approx -> freq): the PulseGen of the
// outer graph on the same audio-rate frequency node as SawGen i, its width per voice. The pair runs as step_locked_stream
// (mldsp_procs.hpp) while the two phase counters are equal in every lane of the wavefront (slocked<i>, asked once per launch).
static int streamLockPulseOf(const mlgpu_graph* g, size_t i)
{
  const Node& n = g->nodes[i];
  auto streamedOsc = [&](const Node& m, int kind) {
    return m.type == NODE_PROC && m.kind == kind && m.region < 0 && m.rate == RATE_AUDIO && !m.in.empty() && g->nodes[m.in[0]].rate != RATE_VOICE;
  };
  if (!g->lockOscillators || !streamedOsc(n, MLGPU_PROC_SAW_GEN)) return -1;
  for (size_t j = 0; j < g->nodes.size(); ++j)
  {
    const Node& m = g->nodes[j];
    if (streamedOsc(m, MLGPU_PROC_PULSE_GEN) && m.in[0] == n.in[0] && (m.in.size() == 1 || g->nodes[m.in[1]].rate == RATE_VOICE))
    {
      for (size_t k = 0; k < i; ++k)  // the first saw on that frequency takes the first pulse on it
        if (streamedOsc(g->nodes[k], MLGPU_PROC_SAW_GEN) && g->nodes[k].in[0] == n.in[0]) return -1;
      return (int)j;
    }
  }
  return -1;
}
static int streamLockSawOf(const mlgpu_graph* g, size_t i)  // the saw of the pair node i belongs to, -1: none
{
  if (g->nodes[i].type != NODE_PROC) return -1;
  if (g->nodes[i].kind == MLGPU_PROC_SAW_GEN) return streamLockPulseOf(g, i) >= 0 ? (int)i : -1;
  if (g->nodes[i].kind != MLGPU_PROC_PULSE_GEN) return -1;
  for (size_t k = 0; k < g->nodes.size(); ++k)
    if (g->nodes[k].type == NODE_PROC && g->nodes[k].kind == MLGPU_PROC_SAW_GEN && streamLockPulseOf(g, k) == (int)i) return (int)k;
  return -1;
}

// ring layout 0, a delay node of the outer graph: its ring read is issued ahead of its place in the graph (RingCore::readEarly)
static bool earlyRingReads(const mlgpu_graph* g, const Node& n) { return g->earlyRows && n.earlySlot >= 0; }

std::string nodeExpr(const mlgpu_graph* g, size_t i, int l, const std::string& ph = "", const std::string& idx = "q * 4 + k")
{
  const Node& n = g->nodes[i];
  std::ostringstream s;
  const std::string L = "_" + std::to_string(l);
  // an input that lives in an enclosing region (or outside) carries only the phase letters of ITS regions
  auto arg = [&](size_t j) { return "n" + std::to_string(n.in[j]) + ph.substr(0, (size_t)upDepth(g, g->nodes[n.in[j]].region)) + L; };
  switch (n.type)
  {
    case NODE_INPUT: s << "xin" << n.slot << L << "[k]"; break;
    case NODE_CONTROL: s << "ctl" << n.slot << L << "[t * a.V]"; break;
    case NODE_EVENT_ROW: s << (n.slot == 0 ? "evP" : "evG") << L << "[k]"; break;
    case NODE_PARAM: s << "a.params[(size_t)" << n.slot << " * a.V + v" << L << "]"; break;
    case NODE_CONST:
      if (g->liveConsts) s << "a.consts[" << n.slot << "]";  // wave-uniform: a scalar load, kept in an SGPR
      else s << floatLiteral(n.value);
      break;
    case NODE_PROC:
      if (ph.empty() && streamLockSawOf(g, i) >= 0)  // made with its partner just before the first of the two (emitNodes)
        s << "sl" << streamLockSawOf(g, i) << (n.kind == MLGPU_PROC_SAW_GEN ? "s" : "p") << L;
      else if (n.kind == MLGPU_PROC_TEMPO_LOCK && g->nodes[n.in[0]].type != NODE_INPUT)  // the phasor to follow is computed in this graph
        s << "p" << i << L << ".next_x(" << idx << ", " << arg(0) << ", " << arg(1) << ", " << arg(2) << ")";
      else if (mlgpu_proc_is_vector_rate(n.kind))
        s << "p" << i << L << ".next_n(" << idx << ")";
      else if (earlyRingReads(g, n))  // ring layout 0: the read was issued as soon as the delay time was known (emitNodes: pre), here the write and the value
        s << "p" << i << L << (n.kind == MLGPU_PROC_PITCHBENDABLE_DELAY ? ".post_i<" : ".post<") << n.earlyPending << ">("
          << (n.kind == MLGPU_PROC_PITCHBENDABLE_DELAY ? idx + ", " : std::string()) << arg(0) << ")";
      else if (n.kind == MLGPU_PROC_PITCHBENDABLE_DELAY)
        s << "p" << i << L << ".next_i(" << idx << ", " << arg(0) << ", " << arg(1) << ((g->sectorRings && n.region < 0) ? ", qq * 4 + k" : "") << ")";
      else if (g->sectorRings && n.region < 0 && mlgpu_proc_rings(n.kind))
      {
        // ring layout 4: the sample's place in its trip of 8 (a constant once qq and k are unrolled)
        s << "p" << i << L << ".next_k(qq * 4 + k, " << arg(0);
        for (size_t j = 1; j < n.in.size(); ++j) s << ", " << arg(j);
        s << ")";
      }
      else if (isOscTrip(g, n))
        s << "osc" << i << L << "[qq * 4 + k]";  // made for the whole trip before the sample loop
      else if ((n.kind == MLGPU_PROC_SAW_GEN || n.kind == MLGPU_PROC_PULSE_GEN) && g->nodes[n.in[0]].rate == RATE_VOICE)
      {
        // launch-constant frequency: the polyBLEP range test was done once per wavefront (odd<i>)
        const bool widthSignal = n.in.size() == 2 && g->nodes[n.in[1]].rate != RATE_VOICE;
        s << "p" << i << L << (widthSignal ? ".next_uw(" : ".next_u(") << arg(0);
        if (n.in.size() == 2) s << ", " << arg(1);
        s << ", odd" << i << ")";
      }
      else if (n.kind == MLGPU_PROC_PULSE_GEN && (n.in.size() == 1 || g->nodes[n.in[1]].rate == RATE_VOICE))
      {
        // streamed frequency, launch-constant width: the width's range test was done once per wavefront (oddw<i>)
        s << "p" << i << L << ".next_sw(" << arg(0);
        if (n.in.size() == 2) s << ", " << arg(1);
        s << ", oddw" << i << ")";
      }
      else if (n.kind == MLGPU_PROC_PULSE_GEN && n.in.size() == 2)
        s << "p" << i << L << ".next2(" << arg(0) << ", " << arg(1) << ")";
      else
      {
        s << "p" << i << L << ".next(" << (n.in.empty() ? std::string("0.f") : arg(0));
        for (size_t j = 1; j < n.in.size(); ++j) s << ", " << arg(j);
        s << ")";
      }
      break;
    case NODE_OP:
      if (n.kind == MLGPU_OP_CLAMP && clampHasConstBounds(g, n))
      {
        s << "clamp_const_bounds(" << arg(0) << ", " << arg(1) << ", " << arg(2) << ")";  // two instructions (mldsp_math.hpp)
        break;
      }
      s << "apply_f<" << n.kind << ">(" << arg(0);
      for (size_t j = 1; j < n.in.size(); ++j) s << ", " << arg(j);
      s << ")";
      break;
    case NODE_FEEDBACK:
      if (n.region < 0)
        s << "fbv" << i << L << "[k]";  // fetched for the whole quad before the sample loop
      else
        s << "u2f(a.state[(size_t)(" << n.sOff << " + " << idx << ") * a.V + v" << L << "])";
      break;
    case NODE_ROUTE:
      if (n.kind == MLGPU_ROUTE_MULTIPLEX || n.kind == MLGPU_ROUTE_MULTIPLEX_LINEAR)
      {
        s << (n.kind == MLGPU_ROUTE_MULTIPLEX ? "route_multiplex_v(" : "route_multiplex_linear_v(") << arg(0);
        for (size_t j = 1; j < n.in.size(); ++j) s << ", " << arg(j);
        s << ")";
      }
      else
        s << (n.kind == MLGPU_ROUTE_DEMULTIPLEX ? "route_demultiplex(" : "route_demultiplex_linear(") << arg(0) << ", " << arg(1) << ", "
          << n.slot << ", " << n.nOut << ")";
      break;
    case NODE_VOP:
      if (n.kind == MLGPU_VOP_TABLE)
      {
        s << "u2f(cv" << i << "[" << idx << "])";  // same index for every lane: a scalar load from constant memory
        break;
      }
      s << "vop<" << n.kind << ">(" << idx;
      for (size_t j = 0; j < n.in.size(); ++j) s << ", " << arg(j);
      s << ")";
      break;
  }
  return s.str();
}

// Voices per lane and quads per trip of the sample loop. A fused voice is ONE dependent chain of VALU instructions per lane;
// two voices per lane (voice v and v + 256 of the same workgroup: loads and stores stay coalesced) interleave two chains,
// two quads per trip give the scheduler a longer window. Both also double the code and cost registers, and on this chip the
// plain form - one voice, one quad - is the fastest for every graph measured so far (config 5: 0.82 ms per launch against
// 0.95 with two voices per lane and 0.93 with two quads; a 34 KiB loop body falls off the instruction cache and runs at half
// speed). So the plain form is the default; mlgpu_graph_set_voices_per_lane forces two voices, and
// mlgpu_graph_set_autotune lets the first launches try all four forms and keep the fastest.
int graphVoicesPerLane(const mlgpu_graph* g)
{
  for (size_t o = 0; o < g->outputs.size(); ++o)
    if (g->outputMix[o]) return 1;
  if (g->voicesPerLane > 0)
  {
    for (const Node& n : g->nodes)
      if (n.type == NODE_FEEDBACK || (n.type == NODE_PROC && (mlgpu_proc_rings(n.kind) || mlgpu_proc_is_vector_rate(n.kind)))) return 1;
    return g->voicesPerLane;
  }
  return 1;
}

std::string generateGraphSource(mlgpu_graph* g, int forceVl = 0)
{
  const int VL = forceVl > 0 ? forceVl : graphVoicesPerLane(g);
  g->compiledVoicesPerLane = VL;
  std::ostringstream s;
  auto sfx = [](int l) { return "_" + std::to_string(l); };
  s << "// generated by libmlgpu graph.hip (" << VL << " voice" << (VL > 1 ? "s" : "") << " per lane)\n"
    << (g->transposedRings ? "#define MLGPU_RING_WINDOWS 2\n" : (g->sectorRings ? "#define MLGPU_RING_WINDOWS 3\n" : (g->windowedRings ? "#define MLGPU_RING_WINDOWS 1\n" : ""))) << (g->strictSvf ? "#define MLGPU_SVF_STRICT 1\n" : "") << "#include \"mldsp_kernels.hpp\"\n#include \"mldsp_ops.hpp\"\n" << (g->hasEventRows ? "#include \"mldsp_events.hpp\"\n" : "") << "using namespace mldev;\n";
  for (size_t i = 0; i < g->nodes.size(); ++i)
    if (g->nodes[i].type == NODE_VOP && g->nodes[i].kind == MLGPU_VOP_TABLE)
    {
      s << "__constant__ unsigned cv" << i << "[64] = {";
      for (int j = 0; j < 64; ++j) s << (j ? ", " : "") << "0x" << std::hex << g->nodes[i].table[j] << std::dec << "u";
      s << "};\n";
    }
  // windowed rings: the latency of a sector refill is hidden by other waves only, so keep at least two per SIMD
  s << "extern \"C\" __global__ __launch_bounds__(256" << ((g->transposedRings && g->totalRings == 1) ? ", 4" : (g->sectorRings && g->totalRings) ? (getenv("MLGPU_SECTOR_WAVES") ? std::string(", ") + getenv("MLGPU_SECTOR_WAVES") : std::string(", 1")) : (g->windowedRings && g->totalRings) ? ", 2" : (g->minWaves ? ", " + std::to_string(g->minWaves) : std::string())) << ") void mlgpu_graph_kernel(const GraphArgs a)\n{\n  apply_fp_mode(a.flags);\n";
  if (g->hasImpulse)
  {
    s << "  __shared__ float ldsTable[32];\n  if (threadIdx.x < 17) ldsTable[threadIdx.x] = a.impulseTable[threadIdx.x];\n  __syncthreads();\n";
    s << "  const KernelTables tables{ldsTable};\n";
  }
  else
  {
    s << "  const KernelTables tables{nullptr};\n";
  }
  if (g->transposedRings && g->totalRings) s << "  __shared__ float ldsRings[" << (size_t)g->totalRings * 4 << " * kTStrip];  // [ring][wavefront][40 rows][64]: write window + two read chunks\n";
  // ring layout 4, per wavefront: every delay node's held sectors (512 floats per ring) and history rows (1024 floats per node)
  std::vector<size_t> sectorLdsOff(g->nodes.size(), 0);
  size_t sectorLdsPerWave = 0;
  if (g->sectorRings)
    for (size_t i = 0; i < g->nodes.size(); ++i)
      if (g->nodes[i].type == NODE_PROC && g->nodes[i].ringLen)
      {
        sectorLdsOff[i] = sectorLdsPerWave;
        sectorLdsPerWave += (size_t)mlgpu_proc_rings(g->nodes[i].kind) * 512 + 1024;
      }
  if (g->sectorRings && g->totalRings) s << "  __shared__ __attribute__((aligned(16))) float ldsRings[" << 4 * sectorLdsPerWave << "];  // [wavefront][node: held sectors, history rows]\n";
  else if (!g->transposedRings && g->windowedRings && g->totalRings) s << "  __shared__ float ldsRings[" << (size_t)g->totalRings * 8 * 256 << "];  // write windows, [ring][8][256 lanes]\n";
  if (g->earlyRows)
    s << "  __shared__ float ldsEarly[" << 4 * g->earlySlots * 64 << "];  // [wavefront][ring read][64 lanes]: where the early ring reads land\n"
      << "  float* const ldsEarlyWave = ldsEarly + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * " << g->earlySlots * 64 << ";\n";
  // a group sum of 16 voices (one instrument's voices): four quads of the wavefront's 64 voices are parked in LDS and every lane
  // then adds up ONE (instrument, sample) pair in voice order - 2.3 instructions per voice-sample where the lane-shift chain
  // (group_sum_in_order) takes 16 (MLGPU_GRAPH_GROUP_SUM=dpp: that form)
  static const bool dppSum = getenv("MLGPU_GRAPH_GROUP_SUM") && !strcmp(getenv("MLGPU_GRAPH_GROUP_SUM"), "dpp");
  auto ldsSum = [&](size_t o) { return VL == 1 && g->outputGroup[o] == 16 && !dppSum; };
  for (size_t o = 0; o < g->outputs.size(); ++o)
    if (ldsSum(o))
      s << "  __shared__ float ldsSum" << o << "[4 * kGroup16Strip];\n  float* const strip" << o << " = ldsSum" << o << " + (threadIdx.x >> 6) * kGroup16Strip;\n";
  // Ring layout 2 moves a voice's pieces with its NEIGHBOURS' lanes: a bank whose last wavefront is not full keeps that wavefront's
  // spare lanes running. They run the bank's last voice again - same inputs, same state, same stores - on ring memory and LDS
  // columns of their own (vr: the lane's place; the rings are laid out for whole 256-voice blocks and cleared together, so a spare
  // lane's ring always holds what the last voice's holds).
  bool anyMix = false;
  for (size_t o = 0; o < g->outputs.size(); ++o)
    if (g->outputMix[o])
    {
      anyMix = true;
      s << "  __shared__ __attribute__((aligned(16))) float ldsMix" << o << "[4 * kMixStrip];\n  float* const mstrip" << o << " = ldsMix" << o << " + (threadIdx.x >> 6) * kMixStrip;\n";
    }
  // (... and so does an output that is the mixdown of all voices: the tree over a wavefront's 64 lanes, a spare lane adds +0)
  const bool partialWaves = ((g->transposedRings && g->totalRings) || anyMix) && (g->V % 64);
  s << "  size_t blk = blockIdx.x;\n  const size_t nbFull = (size_t)gridDim.x & ~(size_t)7;\n"
       "  if (blk < nbFull) blk = (blk & 7) * (nbFull >> 3) + (blk >> 3);\n";
  if (partialWaves)
    s << "  const size_t vr_0 = blk * 256 + threadIdx.x;\n  if ((vr_0 & ~(size_t)63) >= a.V) return;\n  const size_t v_0 = vr_0 < a.V ? vr_0 : a.V - 1;\n";
  else
    s << "  const size_t v_0 = blk * " << 256 * VL << " + threadIdx.x;\n  if (v_0 >= a.V) return;\n";
  const std::string ringLane = partialWaves ? "vr" : "v";
  // a row of the state memory at this lane: the row's address is wave-uniform (scalar arithmetic), the lane's place a 32-bit offset on
  // it - one memory instruction where `a.state[row * a.V + v]` with a 64-bit v is a 64-bit vector add in front of it
  const bool stateAddr32 = g->V < ((size_t)1 << 30) && !(getenv("MLGPU_GRAPH_ROW_ADDR32") && !strcmp(getenv("MLGPU_GRAPH_ROW_ADDR32"), "0"));  // (developer knob, A / B)
  if (stateAddr32) s << "  const uint32_t v4_0 = (uint32_t)v_0 * 4u;\n";
  auto stateRef = [&](const std::string& row, int l) {
    return stateAddr32 ? "*state_row(a, " + row + ", v4" + sfx(l) + ")" : "a.state[(size_t)(" + row + ") * a.V + v" + sfx(l) + "]";
  };
  if (!g->waveClockPath.empty()) s << "  const unsigned long long waveClock0 = __builtin_amdgcn_s_memrealtime();\n";
  // a lane whose second voice does not exist recomputes its first one: same inputs, same state, same stores
  for (int l = 1; l < VL; ++l)
  {
    s << "  const size_t v" << sfx(l) << " = (v_0 + " << 256 * l << " < a.V) ? v_0 + " << 256 * l << " : v_0;\n";
    if (stateAddr32) s << "  const uint32_t v4" << sfx(l) << " = (uint32_t)v" << sfx(l) << " * 4u;\n";
  }
  auto emit = [&](size_t i, const char* indent) {
    for (int l = 0; l < VL; ++l)
    {
      s << indent << "const float n" << i << sfx(l) << " = " << nodeExpr(g, i, l) << ";";
      if (l == 0 && !g->nodes[i].name.empty()) s << "  // " << g->nodes[i].name;
      s << "\n";
    }
  };
  // once per voice: processor state, signal bases, voice-rate nodes
  for (size_t i = 0; i < g->nodes.size(); ++i)
  {
    const Node& n = g->nodes[i];
    for (int l = 0; l < VL; ++l)
    {
      const std::string L = sfx(l);
      if (n.type == NODE_PROC)
      {
        s << "  Proc<" << n.kind << "> p" << i << L << ";\n  const VoiceMem m" << i << L << "{a.coeffs + (size_t)" << n.cOff << " * a.V + v" << L
          << ", a.state + (size_t)" << n.sOff << " * a.V + v" << L << ", a.V";
        // (a ring of at most 4 GiB over the bank: 32-bit row offsets from the wave-uniform start of the ring)
        const bool a32 = n.ringLen && g->rowAddr32 && (size_t)n.ringLen * g->V * sizeof(float) <= ((size_t)1 << 32) && n.ringLen < ((size_t)1 << 24);
        if (n.ringLen && !g->windowedRings) s << ", a.mem + (size_t)" << n.memOff << " * a.V" << (a32 ? std::string() : " + v" + std::string(L)) << ", " << (n.ringLen - 1) << "u";
        if (n.ringLen && !g->windowedRings && (earlyRingReads(g, n) || a32)) s << ", " << (earlyRingReads(g, n) ? "ldsEarlyWave + " + std::to_string(n.earlySlot * 64) : std::string("nullptr"));
        if (a32) s << ", 0u, (uint32_t)v" << L << " * 4u, (uint32_t)a.V * 4u, true";
        if (n.ringLen && g->transposedRings)
          s << ", a.mem + (size_t)" << n.memOff << " * ((a.V + 255) & ~(size_t)255) + (" << ringLane << L << " >> 8) * (size_t)" << n.ringLen * (size_t)mlgpu_proc_rings(n.kind) * 256
            << " + (" << ringLane << L << " & 255) * 16, " << (n.ringLen - 1)
            << "u, ldsRings + (" << (size_t)n.ringSlot * 4 << " + (threadIdx.x >> 6)) * kTStrip + (threadIdx.x & 63)";
        else if (n.ringLen && g->windowedRings)
          s << ", a.mem + (size_t)" << n.memOff << " * ((a.V + 255) & ~(size_t)255) + (v" << L << " >> 8) * (size_t)" << n.ringLen * (size_t)mlgpu_proc_rings(n.kind) * 256
            << " + (v" << L << " & 255) * 8, " << (n.ringLen - 1)
            << "u, ldsRings + " << (g->sectorRings ? "(threadIdx.x >> 6) * " + std::to_string(sectorLdsPerWave) + " + " + std::to_string(sectorLdsOff[i]) + ", " + std::to_string((size_t)mlgpu_proc_rings(n.kind) * 512) + "u"
                                                   : std::to_string((size_t)n.ringSlot * 8 * 256) + " + threadIdx.x");
        s << "};\n  p" << i << L << ".load(m" << i << L << ", tables);\n";
      }
      else if (n.type == NODE_INPUT)
      {
        const std::string row = g->inputGroup[n.slot] > 1 ? "(v" + std::string(L) + " / " + std::to_string(g->inputGroup[n.slot]) + ")" : "v" + std::string(L);
        s << "  const f32x4* in" << n.slot << L << " = (const f32x4*)a.in[" << n.slot << "].base + " << row << " * a.in[" << n.slot << "].strideV;\n";
      }
      else if (n.type == NODE_CONTROL)
      {
        s << "  const float* ctl" << n.slot << L << " = a.ctl[" << n.slot << "] + v" << L << ";\n";
      }
    }
    if (n.rate == RATE_VOICE && n.type != NODE_PROC) emit(i, "  ");
  }
  for (size_t i = 0; i < g->nodes.size(); ++i)
  {
    const Node& n = g->nodes[i];
    if (n.type == NODE_PROC && (n.kind == MLGPU_PROC_SAW_GEN || n.kind == MLGPU_PROC_PULSE_GEN) && g->nodes[n.in[0]].rate == RATE_VOICE)
    {
      s << "  const bool odd" << i << " = __builtin_amdgcn_ballot_w64(blep_freq_is_odd(n" << n.in[0] << "_0)";
      for (int l = 1; l < VL; ++l) s << " || blep_freq_is_odd(n" << n.in[0] << sfx(l) << ")";
      // a PulseGen whose width is per voice too (its own coefficient, or a voice-rate node): the width's range joins the test
      if (n.kind == MLGPU_PROC_PULSE_GEN && (n.in.size() == 1 || g->nodes[n.in[1]].rate == RATE_VOICE))
        for (int l = 0; l < VL; ++l)
        {
          if (n.in.size() == 1) s << " || pulse_width_is_odd(p" << i << sfx(l) << ".width)";
          else s << " || pulse_width_is_odd(n" << n.in[1] << sfx(l) << ")";
        }
      s << ") != 0;\n";
      if (isOscTrip(g, n))
      {
        s << "  const bool dense" << i << " = odd" << i << " || __builtin_amdgcn_ballot_w64(";
        for (int l = 0; l < VL; ++l) s << (l ? " || " : "") << "trip_freq_is_dense(n" << n.in[0] << sfx(l) << ", " << g->oscTripQ * 4 << ")";
        s << ") != 0;\n";
      }
    }
    else if (n.type == NODE_PROC && n.kind == MLGPU_PROC_PULSE_GEN && (n.in.size() == 1 || g->nodes[n.in[1]].rate == RATE_VOICE))
    {
      s << "  const bool oddw" << i << " = __builtin_amdgcn_ballot_w64(";
      for (int l = 0; l < VL; ++l)
      {
        if (l) s << " || ";
        if (n.in.size() == 1) s << "pulse_width_is_odd(p" << i << sfx(l) << ".width)";
        else s << "pulse_width_is_odd(n" << n.in[1] << sfx(l) << ")";
      }
      s << ") != 0;\n";
    }
  }
  for (size_t i = 0; i < g->nodes.size(); ++i)
  {
    const int j = lockedPartner(g, i);
    if (j < 0) continue;
    s << "  const bool locked" << i << " = !dense" << i << " && !dense" << j << " && __builtin_amdgcn_ballot_w64(";
    for (int l = 0; l < VL; ++l) s << (l ? " || " : "") << "p" << i << sfx(l) << ".omega32 != p" << j << sfx(l) << ".omega32";
    s << ") == 0;\n";
  }
  for (size_t i = 0; i < g->nodes.size(); ++i)
  {
    const int j = streamLockPulseOf(g, i);
    if (j < 0) continue;
    s << "  const bool slocked" << i << " = __builtin_amdgcn_ballot_w64(";
    for (int l = 0; l < VL; ++l) s << (l ? " || " : "") << "p" << i << sfx(l) << ".omega32 != p" << j << sfx(l) << ".omega32";
    s << ") == 0;\n";
  }
  for (size_t o = 0; o < g->outputs.size(); ++o)
    for (int l = 0; l < VL; ++l)
    {
      if (g->outputMix[o])  // the rows of 64-voice group sums (mlgpu_mixdown's first stage): [(group * T + t) * 64 + sample]
        s << "  float* const out" << o << sfx(l) << " = (float*)a.out[" << o << "].base + ((" << (partialWaves ? "vr" : "v") << sfx(l) << " >> 6) * a.T) * 64;\n";
      else if (g->outputGroup[o])
        s << "  f32x4* out" << o << sfx(l) << " = (f32x4*)a.out[" << o << "].base + (v" << sfx(l) << " / " << g->outputGroup[o] << ") * a.out[" << o << "].strideV;\n";
      else
        s << "  f32x4* out" << o << sfx(l) << " = (f32x4*)a.out[" << o << "].base + v" << sfx(l) << " * a.out[" << o << "].strideV;\n";
    }
  // a Downsample2x region's filter pairs its parent's samples (m - 1, m): the previous sample of each of its sources
  for (const Region& R : g->regions)
    if (R.kind == MLGPU_REGION_DOWNSAMPLE_2X)
      for (int in : R.ins)
        for (int l = 0; l < VL; ++l) s << "  float prev" << in << sfx(l) << " = 0.f;\n";
  // EventsToSignals rows made here from the control records of e2s_ctl_kernel: one CtlVoice per voice (lane == voice index: MIDI protocol)
  if (g->hasEventRows)
    for (int l = 0; l < VL; ++l) s << "  mlev::CtlVoice ev" << sfx(l) << ";\n  ev" << sfx(l) << ".load(a.events, v" << sfx(l) << ", a.T);\n";
  // Streamed inputs one quad (or one trip) ahead: a wavefront that loads a quad and waits for it right away stands still for a
  // whole HBM round trip per quad, and with four wavefronts per SIMD there are long stretches with only one or two of them able to
  // issue (one wavefront alone issues at 40 % of the SIMD's rate). The very last quad of a launch loads itself again.
  const int PF = (g->nInputs && g->prefetchQ) ? 1 : 0;
  if (PF) s << "  if (a.T == 0) return;\n";
  for (int i = 0; PF && i < g->nInputs; ++i)
    for (int l = 0; l < VL; ++l)
      s << "  const f32x4* pf" << i << sfx(l) << " = in" << i << sfx(l) << ";\n  f32x4 nx" << i << sfx(l) << " = __builtin_nontemporal_load(pf" << i << sfx(l) << ");\n";
  if (g->fbAhead)
    for (size_t i = 0; i < g->nodes.size(); ++i)
      if (g->nodes[i].type == NODE_FEEDBACK && g->nodes[i].region < 0)
        for (int l = 0; l < VL; ++l)
        {
          const std::string nm = std::to_string(i) + sfx(l);
          s << "  float fbn" << nm << "[4], fbm" << nm << "[4];\n#pragma unroll\n  for (int kk = 0; kk < 4; ++kk)\n  {\n    fbn" << nm << "[kk] = u2f(" << stateRef(std::to_string(g->nodes[i].sOff) + " + kk", l)
            << ");\n    fbm" << nm << "[kk] = u2f(" << stateRef(std::to_string(g->nodes[i].sOff) + " + 4 + kk", l) << ");\n  }\n";
        }
  if (g->takeTurns) s << "  const uint32_t turn0 = wave_slot();\n";
  s << "  for (size_t t = 0; t < a.T; ++t)\n  {\n";
  // (Rounds 3-4 walked the event records inside this kernel - 134 spilled registers, 0.35 scalar / branch instructions per vector
  // one; round 5: the record walk is e2s_ctl_kernel's, this kernel expands its control records - mldsp_events.hpp.)
  if (g->hasEventRows)
    for (int l = 0; l < VL; ++l) s << "    ev" << sfx(l) << ".begin_vector(t);\n";
  // once per DSPVector: vector-rate nodes, then the vector-rate processors' begin_vector
  for (size_t i = 0; i < g->nodes.size(); ++i)
  {
    const Node& n = g->nodes[i];
    if (n.rate == RATE_VECTOR) emit(i, "    ");
    if (n.type == NODE_PROC && n.kind == MLGPU_PROC_TEMPO_LOCK && g->nodes[n.in[0]].type != NODE_INPUT)
    {
    }
    else if (n.type == NODE_PROC && n.kind == MLGPU_PROC_TEMPO_LOCK)
    {
      const int slot = g->nodes[n.in[0]].slot;  // the streamed input: first two samples of this vector
      for (int l = 0; l < VL; ++l)
        s << "    { const f32x4 x01 = in" << slot << sfx(l) << "[t * a.in[" << slot << "].strideT]; p" << i << sfx(l) << ".begin_vector(x01[0], x01[1], n"
          << n.in[1] << sfx(l) << ", n" << n.in[2] << sfx(l) << "); }\n";
    }
    else if (n.type == NODE_PROC && mlgpu_proc_is_vector_rate(n.kind))
      for (int l = 0; l < VL; ++l) s << "    p" << i << sfx(l) << ".begin_vector(n" << n.in[0] << sfx(l) << ");\n";
  }
  const bool oscTrips = hasOscTrips(g);
  const bool ringTrips = g->sectorRings && g->totalRings;  // ring layout 4: trips of two quads, every ring's loads in the trip's prologue
  if (oscTrips || ringTrips)
  {
    // the quads in trips of oscTripQ: the oscillators' samples of a trip first, then its quads (fully unrolled: qq is a constant)
    const int tq = ringTrips ? 2 : g->oscTripQ, unroll = (g->windowedRings && g->totalRings) ? 1 : std::max(1, g->unrollQ / tq);
    s << "#pragma unroll " << unroll << "\n    for (int q2 = 0; q2 < 16; q2 += " << tq << ")\n    {\n";
    if (g->takeTurns == 2) s << "    take_turns_by_clock(turn0, " << g->turnClockShift << ");\n";
    else if (g->takeTurns) s << "    take_turns(turn0 + (uint32_t)t * " << 16 / tq << "u + (uint32_t)(q2 / " << tq << "));\n";
    std::vector<char> paired(g->nodes.size(), 0);
    for (size_t i = 0; i < g->nodes.size(); ++i)
      if (lockedPartner(g, i) >= 0) paired[i] = paired[(size_t)lockedPartner(g, i)] = 1;
    for (size_t i = 0; i < g->nodes.size(); ++i)
    {
      const Node& n = g->nodes[i];
      if (!isOscTrip(g, n)) continue;
      for (int l = 0; l < VL; ++l) s << "    float osc" << i << sfx(l) << "[" << tq * 4 << "];\n";
      if (paired[i]) continue;  // made with its partner below
      for (int l = 0; l < VL; ++l)
      {
        s << "    p" << i << sfx(l) << ".trip_u<" << tq * 4 << ">(n" << n.in[0] << sfx(l);
        if (n.in.size() == 2) s << ", n" << n.in[1] << sfx(l);
        s << ", odd" << i << ", dense" << i << ", osc" << i << sfx(l) << ");\n";
      }
    }
    for (size_t i = 0; i < g->nodes.size(); ++i)
    {
      const int j = lockedPartner(g, i);
      if (j < 0) continue;
      const Node &n = g->nodes[i], &m = g->nodes[(size_t)j];
      auto width = [&](int l) { return m.in.size() == 2 ? "n" + std::to_string(m.in[1]) + sfx(l) : "p" + std::to_string(j) + sfx(l) + ".width"; };
      for (int l = 0; l < VL; ++l) s << "    const uint32_t keep" << i << sfx(l) << " = p" << i << sfx(l) << ".omega32, keep" << j << sfx(l) << " = p" << j << sfx(l) << ".omega32;\n";
      s << "    bool made" << i << " = locked" << i << ";\n";
      for (int l = 0; l < VL; ++l)
        s << "    if (made" << i << ") made" << i << " = trip_locked<" << tq * 4 << ">(p" << i << sfx(l) << ", p" << j << sfx(l) << ", n" << n.in[0] << sfx(l) << ", " << width(l)
          << ", osc" << i << sfx(l) << ", osc" << j << sfx(l) << ");\n";
      // (two voices per lane: a suspect trip of the second voice sends both back - the first one's counters are restored below)
      s << "    if (!made" << i << ")\n    {\n";
      for (int l = 0; l < VL; ++l)
      {
        s << "      p" << i << sfx(l) << ".omega32 = keep" << i << sfx(l) << ";\n      p" << j << sfx(l) << ".omega32 = keep" << j << sfx(l) << ";\n";
        s << "      p" << i << sfx(l) << ".trip_u<" << tq * 4 << ">(n" << n.in[0] << sfx(l) << ", odd" << i << ", dense" << i << ", osc" << i << sfx(l) << ");\n";
        s << "      p" << j << sfx(l) << ".trip_u<" << tq * 4 << ">(n" << m.in[0] << sfx(l);
        if (m.in.size() == 2) s << ", n" << m.in[1] << sfx(l);
        s << ", odd" << j << ", dense" << j << ", osc" << j << sfx(l) << ");\n";
      }
      s << "    }\n";
    }
    if (ringTrips)
      for (size_t i = 0; i < g->nodes.size(); ++i)
        if (g->nodes[i].type == NODE_PROC && g->nodes[i].region < 0 && mlgpu_proc_rings(g->nodes[i].kind))
          for (int l = 0; l < VL; ++l) s << "    p" << i << sfx(l) << ".trip_begin();\n";
    s << "#pragma unroll\n    for (int qq = 0; qq < " << tq << "; ++qq)\n    {\n      const int q = q2 + qq;\n";
  }
  else
  {
    s << "#pragma unroll " << ((g->windowedRings && g->totalRings) ? 1 : g->unrollQ) << "\n    for (int q = 0; q < 16; ++q)\n    {\n";
    if (g->takeTurns == 2) s << "      if ((q & 1) == 0) take_turns_by_clock(turn0, " << g->turnClockShift << ");\n";
    else if (g->takeTurns) s << "      if ((q & 1) == 0) take_turns(turn0 + (uint32_t)t * 8u + (uint32_t)(q >> 1));\n";
  }
  // the next quad's address: one step on; from a vector's last quad to the next vector's first; the launch's last quad stays
  if (PF) s << "      const bool lastQ = (q == 15), lastT = (t + 1 == a.T);\n";
  for (int i = 0; i < g->nInputs; ++i)
    for (int l = 0; l < VL; ++l)
    {
      if (PF == 0)
        s << "      const f32x4 xin" << i << sfx(l) << " = __builtin_nontemporal_load(in" << i << sfx(l) << " + t * a.in[" << i << "].strideT + q * a.in["
          << i << "].strideQ);\n";
      else
        s << "      const f32x4 xin" << i << sfx(l) << " = nx" << i << sfx(l) << ";\n      pf" << i << sfx(l) << " += lastQ ? (lastT ? (size_t)0 : a.in[" << i
          << "].strideT - 15 * a.in[" << i << "].strideQ) : a.in[" << i << "].strideQ;\n      nx" << i << sfx(l) << " = __builtin_nontemporal_load(pf" << i << sfx(l) << ");\n";
    }
  for (size_t o = 0; o < g->outputs.size(); ++o)
    for (int l = 0; l < VL; ++l) s << "      f32x4 y" << o << sfx(l) << ";\n";
  if (g->hasEventRows)
    for (int l = 0; l < VL; ++l)
      s << "      mlev::CtlVoice::f32x4e evP" << sfx(l) << ", evG" << sfx(l) << ";\n      ev" << sfx(l) << ".quad(t, q, evP" << sfx(l) << ", evG" << sfx(l) << ");\n";
  // A kept DSPVector's slot n is read and rewritten at sample n only: the quad's four slots are fetched together - and TWO QUADS
  // AHEAD (round 5; they were written 14 quads ago). Fetched at the top of the quad that uses them, every quad of a feedback graph
  // stood still for a memory round trip, behind the stores of the quad before (memory operations of a wavefront complete in issue
  // order): 256 round trips per launch of 16 DSPVectors were the whole launch time of the plucked-string bank, whatever the ring
  // layout. MLGPU_GRAPH_FB_AHEAD=0: the round-4 form (A / B).
  for (size_t i = 0; i < g->nodes.size(); ++i)
    if (g->nodes[i].type == NODE_FEEDBACK && g->nodes[i].region < 0)
      for (int l = 0; l < VL; ++l)
      {
        const std::string nm = std::to_string(i) + sfx(l);
        if (!g->fbAhead)
          s << "      float fbv" << nm << "[4];\n#pragma unroll\n      for (int kk = 0; kk < 4; ++kk) fbv" << nm << "[kk] = u2f("
            << stateRef(std::to_string(g->nodes[i].sOff) + " + q * 4 + kk", l) << ");\n";
        else
          s << "      float fbv" << nm << "[4];\n#pragma unroll\n      for (int kk = 0; kk < 4; ++kk)\n      {\n        fbv" << nm << "[kk] = fbn" << nm << "[kk];\n        fbn" << nm
            << "[kk] = fbm" << nm << "[kk];\n        fbm" << nm << "[kk] = u2f(" << stateRef(std::to_string(g->nodes[i].sOff) + " + ((q + 2) & 15) * 4 + kk", l) << ");\n      }\n";
      }
  for (size_t i = 0; i < g->nodes.size(); ++i)
    if (g->nodes[i].type == NODE_PROC && (g->nodes[i].kind == MLGPU_PROC_LINEAR_GLIDE || g->nodes[i].kind == MLGPU_PROC_HALF_BAND_BUFFERED))
      for (int l = 0; l < VL; ++l) s << "      p" << i << sfx(l) << ".begin_quad(q);\n";
  s << "#pragma unroll\n      for (int k = 0; k < 4; ++k)\n      {\n";
  {
    // Rate regions are emitted in place, recursively. A context = where we are in the tree of regions: the phase letters
    // that name its values, the sample index inside the current function's own DSPVector, and that function's vector count.
    struct Ctx
    {
      std::string sfx, idx, vec, indent;
    };
    auto name = [&](int j, const std::string& ph, int l) { return "n" + std::to_string(j) + ph + sfx(l); };
    auto isInside = [&](int r, int ancestor) {  // r == ancestor or nested somewhere inside it
      for (; r >= 0; r = g->regions[(size_t)r].parent)
        if (r == ancestor) return true;
      return false;
    };
    auto childUnder = [&](int r, int ancestor) {  // the region directly under `ancestor` that contains r
      while (g->regions[(size_t)r].parent != ancestor) r = g->regions[(size_t)r].parent;
      return r;
    };
    std::function<void(int, const Ctx&)> emitNodes;   // the nodes of region r (-1: the outer graph) in context c
    std::function<void(int, const Ctx&)> emitRegion;  // region r, entered from context c of its parent
    // ring layout 0: where a delay node's read goes - right after the last audio-rate node its delay time needs (-1: at the sample's top)
    auto emitPre = [&](size_t dn, const Ctx& c) {
      const Node& m = g->nodes[dn];
      for (int l = 0; l < VL; ++l)
      {
        s << c.indent << "p" << dn << sfx(l) << (m.kind == MLGPU_PROC_PITCHBENDABLE_DELAY ? ".pre_i(" + c.idx : ".pre(");
        for (size_t a = 1; a < m.in.size(); ++a) s << ((a > 1 || m.kind == MLGPU_PROC_PITCHBENDABLE_DELAY) ? ", " : "") << name(m.in[a], "", l);
        s << ");\n";
      }
    };
    // The nodes a delay time is computed from go to the top of the sample with the reads behind them, where nothing of this sample
    // has been stored yet: plain nodes only (operators, inputs, one-vector feedback values, processors without rings - each keeps its
    // own state, so their order among independent nodes is free), and only those whose inputs are such nodes themselves.
    std::vector<char> movable(g->nodes.size(), 0), hoisted(g->nodes.size(), 0);
    std::vector<size_t> batch;  // the delay nodes whose reads are issued at the top, in issue order
    if (g->earlyRows)
    {
      for (size_t j = 0; j < g->nodes.size(); ++j)
      {
        const Node& m = g->nodes[j];
        if (m.rate != RATE_AUDIO)
        {
          movable[j] = 1;  // (a value per voice or per DSPVector: there before the sample loop)
          continue;
        }
        bool ok = m.region < 0 && m.role == ROLE_NONE && (m.type == NODE_OP || m.type == NODE_INPUT || m.type == NODE_FEEDBACK || m.type == NODE_VOP || (m.type == NODE_PROC && !mlgpu_proc_rings(m.kind) && streamLockSawOf(g, j) < 0));
        if (m.type != NODE_FEEDBACK)
          for (int in : m.in) ok = ok && movable[(size_t)in];
        movable[j] = ok;
      }
      std::function<void(int)> want = [&](int j) {
        if (hoisted[(size_t)j] || g->nodes[(size_t)j].rate != RATE_AUDIO) return;
        hoisted[(size_t)j] = 1;
        if (g->nodes[(size_t)j].type != NODE_FEEDBACK)
          for (int in : g->nodes[(size_t)j].in) want(in);
      };
      for (size_t dn = 0; dn < g->nodes.size(); ++dn)
      {
        const Node& m = g->nodes[dn];
        if (!earlyRingReads(g, m)) continue;
        bool all = true;
        for (size_t a = 1; a < m.in.size(); ++a) all = all && movable[(size_t)m.in[a]];
        if (!all) continue;
        for (size_t a = 1; a < m.in.size(); ++a) want(m.in[a]);
        batch.push_back(dn);
      }
      // what is still in flight behind a node's loads when they have landed: at least the loads of the batch issued after them
      int after = 0;
      for (size_t b = batch.size(); b-- > 0;)
      {
        g->nodes[batch[b]].earlyPending = after;
        after += g->nodes[batch[b]].kind == MLGPU_PROC_PITCHBENDABLE_DELAY ? 2 : 1;
      }
    }
    auto inBatch = [&](size_t dn) { return std::find(batch.begin(), batch.end(), dn) != batch.end(); };
    auto emitPlain = [&](size_t j, const Ctx& c) {
      const Node& m = g->nodes[j];
      for (int l = 0; l < VL; ++l)
        s << c.indent << "const float " << name((int)j, c.sfx, l) << " = " << nodeExpr(g, j, l, c.sfx, c.idx) << ";"
          << (l == 0 && !m.name.empty() ? "  // " + m.name : std::string()) << "\n";
    };
    emitNodes = [&](int r, const Ctx& c) {
      std::vector<char> entered(g->regions.size(), 0);
      if (r < 0 && g->earlyRows)
      {
        for (size_t j = 0; j < g->nodes.size(); ++j)
          if (hoisted[j]) emitPlain(j, c);
        for (size_t dn : batch) emitPre(dn, c);
      }
      for (size_t j = 0; j < g->nodes.size(); ++j)
      {
        const Node& m = g->nodes[j];
        if (m.rate != RATE_AUDIO) continue;
        if (r < 0 && hoisted[j]) continue;  // at the top of the sample
        if (m.region != r)
        {
          // the first node of a region nested directly here: the whole region goes in at this point
          if (m.region >= 0 && (r < 0 || isInside(m.region, r)) && m.region != r)
          {
            const int child = childUnder(m.region, r);
            if (!entered[(size_t)child])
            {
              entered[(size_t)child] = 1;
              emitRegion(child, c);
            }
          }
          continue;
        }
        if (m.role == ROLE_REGION_IN) continue;  // made by emitRegion
        if (m.role == ROLE_REGION_OUT)
        {
          const Region& R = g->regions[(size_t)m.slot];
          if (R.kind == MLGPU_REGION_DOWNSAMPLE_2X) continue;  // read before the region's block, see emitRegion
          for (int l = 0; l < VL; ++l)
            s << c.indent << "const float " << name((int)j, c.sfx, l) << " = p" << j << sfx(l) << ".down(" << name(m.in[0], c.sfx + "a", l) << ", "
              << name(m.in[0], c.sfx + "b", l) << ");" << (l == 0 && !m.name.empty() ? "  // " + m.name : std::string()) << "\n";
          continue;
        }
        if (r < 0 && streamLockSawOf(g, j) >= 0)
        {
          // a SawGen / PulseGen pair on one streamed frequency: both values are made where the first of the two stands
          const int si = streamLockSawOf(g, j), pj = streamLockPulseOf(g, (size_t)si);
          if ((int)j == std::min(si, pj))
          {
            const Node &sn = g->nodes[(size_t)si], &pn = g->nodes[(size_t)pj];
            for (int l = 0; l < VL; ++l)
            {
              const std::string freq = "n" + std::to_string(sn.in[0]) + sfx(l);
              const std::string width = pn.in.size() == 2 ? "n" + std::to_string(pn.in[1]) + sfx(l) : "p" + std::to_string(pj) + sfx(l) + ".width";
              s << c.indent << "float sl" << si << "s" << sfx(l) << ", sl" << si << "p" << sfx(l) << ";\n";
              // (the usual case - counters equal, widths regular - behind ONE wave-uniform test per sample)
              s << c.indent << "if (slocked" << si << " && !oddw" << pj << ") step_locked_stream<true>(p" << si << sfx(l) << ", p" << pj << sfx(l) << ", " << freq << ", " << width << ", sl" << si
                << "s" << sfx(l) << ", sl" << si << "p" << sfx(l) << ");\n";
              s << c.indent << "else if (slocked" << si << ") step_locked_stream<false>(p" << si << sfx(l) << ", p" << pj << sfx(l) << ", " << freq << ", " << width << ", sl" << si << "s" << sfx(l)
                << ", sl" << si << "p" << sfx(l) << ");\n";
              s << c.indent << "else\n" << c.indent << "{\n";
              s << c.indent << "  sl" << si << "s" << sfx(l) << " = p" << si << sfx(l) << ".next(" << freq << ");\n";
              s << c.indent << "  sl" << si << "p" << sfx(l) << " = p" << pj << sfx(l) << ".next_sw(" << freq << (pn.in.size() == 2 ? ", " + width : std::string()) << ", oddw" << pj << ");\n";
              s << c.indent << "}\n";
            }
          }
        }
        if (r < 0 && earlyRingReads(g, m) && !inBatch(j)) emitPre(j, c);  // (a delay time made of this sample's own signal: read and value together)
        emitPlain(j, c);
      }
      if (r < 0) return;
      // fn's own one-vector feedback (slot = the sample index inside fn's DSPVector), then the end of fn's DSPVector
      for (size_t j = 0; j < g->nodes.size(); ++j)
        if (g->nodes[j].region == r && g->nodes[j].type == NODE_FEEDBACK && g->nodes[j].fbSource >= 0)
          for (int l = 0; l < VL; ++l)
            s << c.indent << stateRef(std::to_string(g->nodes[j].sOff) + " + " + c.idx, l) << " = f2u(" << name(g->nodes[j].fbSource, c.sfx, l)
              << ");\n";
      bool any = false;
      for (size_t j = 0; j < g->nodes.size(); ++j)
      {
        const Node& m = g->nodes[j];
        if (m.type != NODE_PROC || m.region != r || m.role != ROLE_NONE) continue;
        if (!any) s << c.indent << "if (" << c.idx << " == 63)\n" << c.indent << "{\n";
        any = true;
        for (int l = 0; l < VL; ++l) s << c.indent << "  p" << j << sfx(l) << ".end_vector();\n";
      }
      if (any) s << c.indent << "}\n";
    };
    emitRegion = [&](int r, const Ctx& c) {
      const Region& R = g->regions[(size_t)r];
      if (R.kind == MLGPU_REGION_UPSAMPLE_2X)
      {
        // fn on the two samples the HalfBandFilters make of this sample (upsampleFirstHalf / SecondHalf in stream order)
        for (int phase = 0; phase < 2; ++phase)
        {
          const std::string ph = phase ? "b" : "a";
          Ctx cc;
          cc.sfx = c.sfx + ph;
          cc.idx = "((2 * (" + c.idx + ") + " + std::to_string(phase) + ") & 63)";
          cc.vec = "(2 * (" + c.vec + ") + ((2 * (" + c.idx + ") + " + std::to_string(phase) + ") >> 6))";
          cc.indent = c.indent;
          for (int in : R.ins)
            for (int l = 0; l < VL; ++l)
              s << c.indent << "const float " << name(in, cc.sfx, l) << " = p" << in << sfx(l) << ".up_" << ph << "("
                << name(g->nodes[(size_t)in].in[0], c.sfx.substr(0, (size_t)upDepth(g, g->nodes[(size_t)g->nodes[(size_t)in].in[0]].region)), l) << ");\n";
          emitNodes(r, cc);
        }
      }
      else
      {
        // the region's output is what its upsampler made one DSPVector (of the parent's) ago; fn runs on the parent's odd samples
        const bool top = (R.parent < 0);
        for (int l = 0; l < VL; ++l)
          s << c.indent << "const float " << name(R.out, c.sfx, l) << " = p" << R.out << sfx(l) << (top ? ".delayed(" : ".delayedAt(") << c.idx << ");\n";
        s << c.indent << "if ((" << c.idx << ") & 1)\n" << c.indent << "{\n";
        Ctx cc;
        cc.sfx = c.sfx;
        cc.idx = "((((" + c.idx + ") - 1) >> 1) + 32 * (int)((" + c.vec + ") & 1))";
        cc.vec = "((" + c.vec + ") >> 1)";
        cc.indent = c.indent + "  ";
        for (int in : R.ins)
          for (int l = 0; l < VL; ++l)
            s << cc.indent << "const float " << name(in, cc.sfx, l) << " = p" << in << sfx(l) << ".down(prev" << in << sfx(l) << ", "
              << name(g->nodes[(size_t)in].in[0], c.sfx.substr(0, (size_t)upDepth(g, g->nodes[(size_t)g->nodes[(size_t)in].in[0]].region)), l) << ");\n";
        emitNodes(r, cc);
        for (int l = 0; l < VL; ++l) s << cc.indent << "p" << R.out << sfx(l) << ".push(" << c.idx << ", " << name(R.result, cc.sfx, l) << ");\n";
        s << c.indent << "}\n";
        for (int in : R.ins)
          for (int l = 0; l < VL; ++l)
            s << c.indent << "prev" << in << sfx(l) << " = "
              << name(g->nodes[(size_t)in].in[0], c.sfx.substr(0, (size_t)upDepth(g, g->nodes[(size_t)g->nodes[(size_t)in].in[0]].region)), l) << ";\n";
      }
    };
    Ctx outer;
    outer.idx = "(q * 4 + k)";
    outer.vec = "(a.t0 + t)";
    outer.indent = "        ";
    emitNodes(-1, outer);
  }
  for (size_t o = 0; o < g->outputs.size(); ++o)
    for (int l = 0; l < VL; ++l)
    {
      if (g->outputGroup[o] && !ldsSum(o)) s << "        y" << o << sfx(l) << "[k] = group_sum_in_order<" << g->outputGroup[o] << ">(n" << g->outputs[o] << sfx(l) << ");\n";
      else s << "        y" << o << sfx(l) << "[k] = n" << g->outputs[o] << sfx(l) << ";\n";
    }
  // feedback: keep this sample's value for the same sample of the next DSPVector (its old value was read above)
  for (size_t i = 0; i < g->nodes.size(); ++i)
    if (g->nodes[i].type == NODE_FEEDBACK && g->nodes[i].fbSource >= 0 && g->nodes[i].region < 0)
      for (int l = 0; l < VL; ++l)
        s << "        " << stateRef(std::to_string(g->nodes[i].sOff) + " + q * 4 + k", l) << " = f2u(n" << g->nodes[i].fbSource << sfx(l) << ");\n";
  s << "      }\n";
  for (size_t o = 0; o < g->outputs.size(); ++o)
    if (ldsSum(o))
      s << "      group16_park(strip" << o << ", q & 3, y" << o << "_0);\n      if ((q & 3) == 3) group16_sum_store(strip" << o << ", out" << o << "_0 + t * a.out[" << o
        << "].strideT + (q - 3) * a.out[" << o << "].strideQ, a.out[" << o << "].strideQ);\n";
  for (size_t o = 0; o < g->outputs.size(); ++o)
    if (g->outputMix[o])
      s << "      mix64_park(mstrip" << o << ", q & 3, " << (partialWaves ? "(vr_0 < a.V) ? y" + std::to_string(o) + "_0 : f32x4{0.f, 0.f, 0.f, 0.f}" : "y" + std::to_string(o) + "_0")
        << ");\n      if ((q & 3) == 3) mix64_sum_store(mstrip" << o << ", out" << o << "_0 + t * 64 + (q - 3) * 4);\n";
  for (size_t o = 0; o < g->outputs.size(); ++o)
    for (int l = 0; l < VL && !ldsSum(o) && !g->outputMix[o]; ++l)
      s << "      " << (g->outputGroup[o] ? "if ((threadIdx.x & " + std::to_string(g->outputGroup[o] - 1) + ") == " + std::to_string(g->outputGroup[o] - 1) + ") " : std::string())
        << "__builtin_nontemporal_store(y" << o << sfx(l) << ", out" << o << sfx(l) << " + t * a.out[" << o << "].strideT + q * a.out[" << o << "].strideQ);\n";
  s << "    }\n";
  if (oscTrips || ringTrips) s << "    }\n";
  for (size_t i = 0; i < g->nodes.size(); ++i)
    if (g->nodes[i].type == NODE_PROC && (g->nodes[i].region < 0 || g->nodes[i].role != ROLE_NONE))
      for (int l = 0; l < VL; ++l) s << "    p" << i << sfx(l) << ".end_vector();\n";
  if (g->hasEventRows)
    for (int l = 0; l < VL; ++l) s << "    ev" << sfx(l) << ".end_vector();\n";
  s << "  }\n";
  if (g->hasEventRows)
    for (int l = 0; l < VL; ++l) s << "  ev" << sfx(l) << ".store();\n";
  for (size_t i = 0; i < g->nodes.size(); ++i)
    if (g->nodes[i].type == NODE_PROC)
      for (int l = 0; l < VL; ++l) s << "  p" << i << sfx(l) << ".store(m" << i << sfx(l) << ");\n";
  if (!g->waveClockPath.empty())
    s << "  if ((threadIdx.x & 63) == 0 && a.waveClock)\n  {\n    unsigned long long* w = a.waveClock + (blk * 4 + threadIdx.x / 64) * 4;\n"
         "    w[0] = waveClock0;\n    w[1] = __builtin_amdgcn_s_memrealtime();\n    w[2] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));\n"
         "    w[3] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));\n  }\n";
  s << "}\n";
  return s.str();
}

// Registers and scratch bytes per lane of the (only) kernel of a code object, read from its metadata note (msgpack: the key
// string, then an unsigned integer).
static bool codeObjectNumber(const std::vector<char>& code, const char* key, long& value)
{
  const size_t klen = strlen(key);
  for (size_t i = 0; i + klen + 1 < code.size(); ++i)
  {
    if (memcmp(code.data() + i, key, klen) != 0) continue;
    const unsigned char* p = (const unsigned char*)code.data() + i + klen;
    const size_t left = code.size() - (i + klen);
    if (p[0] <= 0x7f) { value = p[0]; return true; }
    if (p[0] == 0xcc && left >= 2) { value = p[1]; return true; }
    if (p[0] == 0xcd && left >= 3) { value = (p[1] << 8) | p[2]; return true; }
    if (p[0] == 0xce && left >= 5) { value = ((long)p[1] << 24) | (p[2] << 16) | (p[3] << 8) | p[4]; return true; }
  }
  return false;
}

// Source + code object of the graph kernel for `vl` voices per lane, with the register budget chosen: a voice bank is launched
// as whole blocks of four wavefronts, one per SIMD, and a big bank is a few blocks per CU - so a kernel that needs more than
// 128 VGPRs (three, two or one wavefront per SIMD) runs its blocks in rounds where one that fits 128 runs them all at once. If
// the bank is big enough for that to matter (65 536 voices: a block per CU) and the kernel is above 128, it is generated again
// with a bound of four wavefronts per SIMD - then three, then two - and the first build that spills moderately (up to 640 bytes
// of scratch per lane) is kept (the patch of SURVEY 8d: 170 VGPRs ->
// 128 + 156 bytes of scratch per lane, 1.82 -> 1.46 ms; the voice with its EventsToSignals rows inside: 259 -> 128 + 528 bytes,
// 3.13 -> 1.56 ms, where bounds of two and three wavefronts give 1.96 and 1.85). MLGPU_GRAPH_MIN_WAVES=0 / N overrides
// (developer knob).
static bool generateBudgeted(mlgpu_graph* g, int vl, std::string& source, std::vector<char>& code, std::string& log)
{
  const char* knob = getenv("MLGPU_GRAPH_MIN_WAVES");
  g->minWaves = knob ? atoi(knob) : 0;
  source = generateGraphSource(g, vl);
  // developer aid for elimination experiments (what would the launch cost WITHOUT this branch / that test?): the kernel source comes
  // from a file instead - an edited copy of mlgpu_graph_source()'s text. Its results are whatever the file computes.
  if (const char* file = getenv("MLGPU_GRAPH_SOURCE_FILE"))
  {
    std::ifstream in(file);
    std::stringstream text;
    text << in.rdbuf();
    if (!text.str().empty())
    {
      source = text.str();
      return getCode(source, code, log);
    }
  }
  if (!getCode(source, code, log)) return false;
  long vgprs = 0;
  if (knob || (g->windowedRings && g->totalRings) || g->V < 65536 || !codeObjectNumber(code, ".vgpr_count", vgprs) || vgprs <= 128) return true;
  // the tightest bound whose build spills moderately: four wavefronts per SIMD, else three, else two
  for (int waves = 4; waves >= 2; --waves)
  {
    if (((vgprs + 7) & ~7L) * waves <= 512) break;  // the unbounded kernel already allows that many (registers come in blocks of 8 of a SIMD's 512)
    g->minWaves = waves;
    const std::string bounded = generateGraphSource(g, vl);
    g->minWaves = 0;
    std::vector<char> boundedCode;
    std::string boundedLog;
    long scratch = 0;
    if (getCode(bounded, boundedCode, boundedLog) && codeObjectNumber(boundedCode, ".private_segment_fixed_size", scratch) && scratch <= 640)
    {
      source = bounded;
      code.swap(boundedCode);
      break;
    }
  }
  return true;
}

int addNode(mlgpu_graph* g, Node&& n)
{
  if (g->job) return -MLGPU_ERR_BUSY;
    if (g->compiled) return -gfail(g, MLGPU_ERR_INVALID, "graph already compiled");
  for (int id : n.in)
    if (id < 0 || id >= (int)g->nodes.size()) return -gfail(g, MLGPU_ERR_RANGE, "graph node input refers to an unknown node");
  switch (n.type)
  {
    case NODE_PARAM: case NODE_CONST: n.rate = RATE_VOICE; break;
    case NODE_CONTROL: n.rate = RATE_VECTOR; break;
    case NODE_OP: case NODE_ROUTE:
      n.rate = RATE_VOICE;
      for (int id : n.in) n.rate = std::max(n.rate, g->nodes[id].rate);
      break;
    default: n.rate = RATE_AUDIO; break;
  }
  // rate regions: a region's nodes are visible only inside it, and inside it only the region's own nodes and per-voice
  // floats of the outer graph are (fn sees its upsampled / downsampled arguments, nothing else moves at its rate)
  if (n.role == ROLE_NONE)
  {
    for (int id : n.in)
    {
      const Node& src = g->nodes[(size_t)id];
      if (src.region >= 0 && src.region != g->openRegion) return -gfail(g, MLGPU_ERR_INVALID, "graph: a node of a closed rate region is used outside it");
      if (g->openRegion >= 0 && src.region < 0 && src.rate != RATE_VOICE)
        return -gfail(g, MLGPU_ERR_INVALID, "graph: inside a rate region only the region's inputs and per-voice floats (params, consts) can be used");
    }
    if (g->openRegion >= 0)
    {
      const bool vectorProc = (n.type == NODE_PROC && mlgpu_proc_is_vector_rate(n.kind));
      if (n.type == NODE_INPUT || n.type == NODE_CONTROL || n.type == NODE_EVENT_ROW || vectorProc || n.rate == RATE_VECTOR)
        return -gfail(g, MLGPU_ERR_UNSUPPORTED, "graph: streamed inputs, controls and vector-rate processors cannot live inside a rate region");
      if (n.rate == RATE_AUDIO) n.region = g->openRegion;
    }
  }
  g->nodes.push_back(std::move(n));
  return (int)g->nodes.size() - 1;
}

// a processor node with its coefficient and state slots (the checks of mlgpu_graph_add_proc are the caller's)
int addProcNode(mlgpu_graph* g, int kind, const int* inputs, int nIn, const char* name, int role = ROLE_NONE, int region = -1, int slot = 0)
{
  Node n;
  n.type = NODE_PROC;
  n.kind = kind;
  if (nIn) n.in.assign(inputs, inputs + nIn);
  n.name = name ? name : "";
  n.nc = mlgpu_proc_nc(kind);
  n.ns = mlgpu_proc_ns(kind);
  n.cOff = g->NC;
  n.sOff = g->NS;
  n.role = role;
  n.region = region;
  n.slot = slot;
  const int nc = n.nc, ns = n.ns;
  const int id = addNode(g, std::move(n));
  if (id >= 0)
  {
    g->NC += nc;
    g->NS += ns;
    if (kind == MLGPU_PROC_IMPULSE_GEN) g->hasImpulse = true;
  }
  return id;
}

int checkNode(mlgpu_graph* g, int node, int type)
{
  if (!g) return MLGPU_ERR_INVALID;
  if (g->job) return MLGPU_ERR_BUSY;
  if (node < 0 || node >= (int)g->nodes.size()) return gfail(g, MLGPU_ERR_RANGE, "node index out of range");
  if (g->nodes[node].type != type) return gfail(g, MLGPU_ERR_INVALID, "node has the wrong type for this call");
  return MLGPU_OK;
}
// nodes that own state words: processors and feedback nodes (their stored DSPVector, 64 words)
int checkStateNode(mlgpu_graph* g, int node)
{
  if (!g) return MLGPU_ERR_INVALID;
  if (g->job) return MLGPU_ERR_BUSY;
  if (node < 0 || node >= (int)g->nodes.size()) return gfail(g, MLGPU_ERR_RANGE, "node index out of range");
  if (g->nodes[node].type != NODE_PROC && g->nodes[node].type != NODE_FEEDBACK) return gfail(g, MLGPU_ERR_INVALID, "node has no state");
  return MLGPU_OK;
}
}  // namespace

// ---- fused kernels for processor chains without an ahead-of-time instantiation -------------------
// Generates `chain_kernel_body<Chain<kinds...>, HAS_SIGNAL>` wrappers; used by mlgpu_bank_create.
static std::string chainSource(const int32_t* kinds, int n, bool strictSvf = false)
{
  std::ostringstream s;
  s << "// generated by libmlgpu graph.hip (chain)\n" << (strictSvf ? "#define MLGPU_SVF_STRICT 1\n" : "") << "#include \"mldsp_kernels.hpp\"\nusing namespace mldev;\n";
  // a plain cascade of 2, 4 or 8 equal SVF sections keeps its stage-skewed form (one lane per channel; chains.hip picks wider
  // forms by bank size for the ahead-of-time kernels): strict mode must not cost config 4 its kernel
  bool cascade = (n == 2 || n == 4 || n == 8) && (kinds[0] == MLGPU_PROC_LOPASS || kinds[0] == MLGPU_PROC_HIPASS || kinds[0] == MLGPU_PROC_BANDPASS);
  for (int i = 1; i < n; ++i) cascade = cascade && kinds[i] == kinds[0];
  if (cascade)
  {
    for (int sig = 1; sig >= 0; --sig)
      s << "extern \"C\" __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void mlgpu_chain_" << (sig ? "signal" : "const")
        << "(const ChainArgs a) { cascade_lanes_body<" << kinds[0] << ", " << n << ", 1, 8, " << (sig ? "true" : "false") << ">(a); }\n";
    return s.str();
  }
  s << "using CH = Chain<";
  for (int i = 0; i < n; ++i) s << (i ? ", " : "") << kinds[i];
  s << ">;\n"
       "extern \"C\" __global__ __launch_bounds__(256) void mlgpu_chain_signal(const ChainArgs a) { chain_kernel_body<CH, true>(a); }\n"
       "extern \"C\" __global__ __launch_bounds__(256) void mlgpu_chain_const(const ChainArgs a) { chain_kernel_body<CH, false>(a); }\n";
  return s.str();
}

bool mlgpu_jit_chain(mlgpu_engine* e, const int32_t* kinds, int n, void** fnSignal, void** fnConst, std::string& log)
{
  const std::string src = chainSource(kinds, n, e->strictSvf);
  CompiledModule* cm = compileAndLoad(e->device, src, log);
  if (!cm) return false;
  *fnSignal = (void*)getFunction(cm, "mlgpu_chain_signal", log);
  *fnConst = (void*)getFunction(cm, "mlgpu_chain_const", log);
  return *fnSignal && *fnConst;
}

hipError_t mlgpu_jit_chain_launch(void* fn, const ChainArgs& a, hipStream_t stream) { return launchJit((hipFunction_t)fn, a, a.V, stream); }

// ... and the form of a chain that sums its voices inside the kernel (chain_kernel_body<CH, HAS_SIGNAL, true>, mlgpu_bank_prepare_mixdown):
// generated on request for the chains chains.hip has no ahead-of-time instantiation of
bool mlgpu_jit_chain_mix(mlgpu_engine* e, const int32_t* kinds, int n, void** fnSignal, void** fnConst, std::string& log)
{
  std::ostringstream s;
  s << "// generated by libmlgpu graph.hip (chain, voices summed in the kernel)\n" << (e->strictSvf ? "#define MLGPU_SVF_STRICT 1\n" : "")
    << "#include \"mldsp_kernels.hpp\"\nusing namespace mldev;\nusing CH = Chain<";
  for (int i = 0; i < n; ++i) s << (i ? ", " : "") << kinds[i];
  s << ">;\n"
       "extern \"C\" __global__ __launch_bounds__(256) void mlgpu_chain_mix_signal(const ChainArgs a) { chain_kernel_body<CH, true, true>(a); }\n"
       "extern \"C\" __global__ __launch_bounds__(256) void mlgpu_chain_mix_const(const ChainArgs a) { chain_kernel_body<CH, false, true>(a); }\n";
  CompiledModule* cm = compileAndLoad(e->device, s.str(), log);
  if (!cm) return false;
  *fnSignal = (void*)getFunction(cm, "mlgpu_chain_mix_signal", log);
  *fnConst = (void*)getFunction(cm, "mlgpu_chain_mix_const", log);
  return *fnSignal && *fnConst;
}

extern "C"
{
  // Every generated kernel this process holds (compiled here or read from the disk cache), as one relocatable blob: what an
  // installation WITHOUT hiprtc is given so that its graphs and chains find their code. Header: magic, the fingerprint of the device
  // headers the kernels were generated from (a bundle of another build is refused: its kernels would be looked up by other sources
  // anyway), count; then per kernel the generated source (the key) and the code object.
  static const char kBundleMagic[8] = {'M', 'L', 'G', 'P', 'U', 'K', 'B', '1'};
  int mlgpu_jit_cache_export(void* buffer, size_t capacity, size_t* needed)
  {
    std::lock_guard<std::mutex> lock(g_codeMutex);
    const std::string fp = mlgpu_device_source_hash_str;
    size_t total = sizeof(kBundleMagic) + 8 + fp.size() + 8;
    for (const auto& kv : g_codeCache) total += 16 + kv.first.size() + kv.second.size();
    if (needed) *needed = total;
    if (!buffer) return MLGPU_OK;
    if (capacity < total) return MLGPU_ERR_RANGE;
    char* p = (char*)buffer;
    auto put = [&p](const void* src, size_t n) {
      memcpy(p, src, n);
      p += n;
    };
    auto put64 = [&put](uint64_t v) { put(&v, 8); };
    put(kBundleMagic, sizeof(kBundleMagic));
    put64(fp.size());
    put(fp.data(), fp.size());
    put64(g_codeCache.size());
    for (const auto& kv : g_codeCache)
    {
      put64(kv.first.size());
      put64(kv.second.size());
      put(kv.first.data(), kv.first.size());
      put(kv.second.data(), kv.second.size());
    }
    return MLGPU_OK;
  }
  int mlgpu_jit_cache_import(const void* buffer, size_t size, size_t* kernels)
  {
    if (kernels) *kernels = 0;
    if (!buffer) return MLGPU_ERR_INVALID;
    const char *p = (const char*)buffer, *end = p + size;
    auto get64 = [&p, end](uint64_t& v) {
      if ((size_t)(end - p) < 8) return false;
      memcpy(&v, p, 8);
      p += 8;
      return true;
    };
    if (size < sizeof(kBundleMagic) || memcmp(p, kBundleMagic, sizeof(kBundleMagic)) != 0) return MLGPU_ERR_INVALID;
    p += sizeof(kBundleMagic);
    uint64_t n = 0;
    if (!get64(n) || (size_t)(end - p) < n) return MLGPU_ERR_INVALID;
    if (std::string(p, (size_t)n) != mlgpu_device_source_hash_str) return MLGPU_ERR_UNSUPPORTED;  // kernels of another build of the device code
    p += n;
    uint64_t count = 0;
    if (!get64(count)) return MLGPU_ERR_INVALID;
    std::vector<std::pair<std::string, std::vector<char>>> items;
    for (uint64_t i = 0; i < count; ++i)
    {
      uint64_t ns = 0, nc = 0;
      if (!get64(ns) || !get64(nc) || (size_t)(end - p) < ns || (size_t)(end - p) - ns < nc) return MLGPU_ERR_INVALID;
      if (nc < 4 || memcmp(p + ns, "\x7f" "ELF", 4) != 0) return MLGPU_ERR_INVALID;  // (what goes to the module loader is at least an ELF file)
      items.emplace_back(std::string(p, (size_t)ns), std::vector<char>(p + ns, p + ns + nc));
      p += ns + nc;
    }
    std::lock_guard<std::mutex> lock(g_codeMutex);
    for (auto& it : items) g_codeCache[it.first] = std::move(it.second);
    if (kernels) *kernels = items.size();
    return MLGPU_OK;
  }
  // (tests) forget the kernels held in memory: the next request goes to the disk cache or the compiler again
  int mlgpu_jit_cache_clear_memory(void)
  {
    std::lock_guard<std::mutex> lock(g_codeMutex);
    g_codeCache.clear();
    return MLGPU_OK;
  }
  // 1 where run-time compilation is there (libhiprtc.so can be loaded and MLGPU_HIPRTC is not "off"), else 0
  int mlgpu_jit_compiler_available(void) { return hiprtc() ? 1 : 0; }

  int mlgpu_jit_stats(uint64_t* compiles, uint64_t* diskHits, uint64_t* memoryHits, double* compileSeconds, double* diskLoadSeconds)
  {
    std::lock_guard<std::mutex> lock(g_codeMutex);
    if (compiles) *compiles = g_jitStats.compiles;
    if (diskHits) *diskHits = g_jitStats.diskHits;
    if (memoryHits) *memoryHits = g_jitStats.memoryHits;
    if (compileSeconds) *compileSeconds = g_jitStats.compileSeconds;
    if (diskLoadSeconds) *diskLoadSeconds = g_jitStats.diskLoadSeconds;
    return MLGPU_OK;
  }

  int mlgpu_jit_selftest(char* logOut, size_t logLen)
  {
    std::string log, all;
    bool ok = true;
    // (1) a chain that has no ahead-of-time instantiation, including the LDS-table ImpulseGen
    const int32_t chain[] = {MLGPU_PROC_IMPULSE_GEN, MLGPU_PROC_LO_SHELF, MLGPU_PROC_ADSR, MLGPU_PROC_PEAK, MLGPU_PROC_GAIN};
    ok = compileOnly(chainSource(chain, 5), log) && ok;
    all += log;
    // (1b) the strict-SVF forms: a generic chain and a plain cascade (which keeps its stage-skewed kernel)
    const int32_t svfChain[] = {MLGPU_PROC_SAW_GEN, MLGPU_PROC_BANDPASS, MLGPU_PROC_HI_SHELF, MLGPU_PROC_GAIN};
    ok = compileOnly(chainSource(svfChain, 4, true), log) && ok;
    all += log;
    const int32_t casc[] = {MLGPU_PROC_HIPASS, MLGPU_PROC_HIPASS, MLGPU_PROC_HIPASS, MLGPU_PROC_HIPASS};
    ok = compileOnly(chainSource(casc, 4, true), log) && ok;
    all += log;
    // (2) a graph touching every node type
    mlgpu_graph g;
    g.V = 64;
    const int gate = mlgpu_graph_add_input(&g, "gate");
    const int pitch = mlgpu_graph_add_param(&g, "pitch");
    const int two = mlgpu_graph_add_const(&g, 2.0f);
    const int f = mlgpu_graph_add_op(&g, MLGPU_OP_EXP2_APPROX, &pitch, 1, "freq");
    const int saw = mlgpu_graph_add_proc(&g, MLGPU_PROC_SAW_GEN, &f, 1, "saw");
    const int pin[2] = {f, gate};
    const int pulse = mlgpu_graph_add_proc(&g, MLGPU_PROC_PULSE_GEN, pin, 2, "pulse");
    const int noise = mlgpu_graph_add_proc(&g, MLGPU_PROC_NOISE_GEN, nullptr, 0, "noise");
    const int env = mlgpu_graph_add_proc(&g, MLGPU_PROC_ADSR, &gate, 1, "env");
    const int mixIn[2] = {saw, pulse};
    const int mix = mlgpu_graph_add_op(&g, MLGPU_OP_ADD, mixIn, 2, "mix");
    const int lerpIn[3] = {mix, noise, two};
    const int l = mlgpu_graph_add_op(&g, MLGPU_OP_LERP, lerpIn, 3, "lerp");
    const int lp = mlgpu_graph_add_proc(&g, MLGPU_PROC_LOPASS, &l, 1, "lp");
    const int vcaIn[2] = {lp, env};
    const int vca = mlgpu_graph_add_op(&g, MLGPU_OP_MULTIPLY, vcaIn, 2, "vca");
    // control-rate inputs, ramps, index generators, the other operator() forms
    const int cut = mlgpu_graph_add_control(&g, "cutoff");
    const int glide = mlgpu_graph_add_proc(&g, MLGPU_PROC_LINEAR_GLIDE, &cut, 1, "glide");
    const int interp = mlgpu_graph_add_proc(&g, MLGPU_PROC_INTERPOLATOR1, &cut, 1, "interp");
    const int sg = mlgpu_graph_add_proc(&g, MLGPU_PROC_SAMPLE_ACCURATE_LINEAR_GLIDE, &gate, 1, "sglide");
    const int lp3In[3] = {vca, glide, interp};
    const int lp3 = mlgpu_graph_add_proc(&g, MLGPU_PROC_LOPASS, lp3In, 3, "lpmod");
    const int rampIn[2] = {pitch, cut};
    const int ramp = mlgpu_graph_add_vop(&g, MLGPU_VOP_INTERPOLATE_LINEAR, rampIn, 2, "ramp");
    const int ci = mlgpu_graph_add_vop(&g, MLGPU_VOP_COLUMN_INDEX, nullptr, 0, "idx");
    const int rc = mlgpu_graph_add_vop(&g, MLGPU_VOP_RANGE_CLOSED, rampIn, 2, "rc");
    const int ro = mlgpu_graph_add_vop(&g, MLGPU_VOP_RANGE_OPEN, rampIn, 2, "ro");
    const int lsIn[6] = {lp3, ramp, ci, rc, ro, sg};
    const int ls = mlgpu_graph_add_proc(&g, MLGPU_PROC_LO_SHELF, lsIn, 6, "loshelf");
    const int hsIn[7] = {ls, ramp, ci, rc, ro, sg, two};
    const int hs = mlgpu_graph_add_proc(&g, MLGPU_PROC_HI_SHELF, hsIn, 7, "hishelf");
    // delay lines and one-vector feedback (an Allpass<PitchbendableDelay> written out)
    const int fb = mlgpu_graph_add_feedback(&g, "vy1");
    const int idIn[2] = {hs, glide};
    const int idl = mlgpu_graph_add_proc(&g, MLGPU_PROC_INTEGER_DELAY, idIn, 2, "idelay");
    const int fdIn[3] = {idl, glide, gate};
    const int fdl = mlgpu_graph_add_proc(&g, MLGPU_PROC_FRACTIONAL_DELAY, fdIn, 3, "fdelay");
    const int pbIn[2] = {fdl, glide};
    const int pbd = mlgpu_graph_add_proc(&g, MLGPU_PROC_PITCHBENDABLE_DELAY, pbIn, 2, "pbdelay");
    const int ap1 = mlgpu_graph_add_proc(&g, MLGPU_PROC_ALLPASS1, &pbd, 1, "ap1");
    const int fbmIn[2] = {ap1, fb};
    const int fbm = mlgpu_graph_add_op(&g, MLGPU_OP_ADD, fbmIn, 2, "fbsum");
    ok = (fb > 0) && (fbm > 0) && (mlgpu_graph_set_feedback(&g, fb, fbm) == MLGPU_OK) && (mlgpu_graph_set_max_delay(&g, idl, 1000.f) == MLGPU_OK) &&
         (mlgpu_graph_set_max_delay(&g, fdl, 100.f) == MLGPU_OK) && (mlgpu_graph_set_max_delay(&g, pbd, 3000.f) == MLGPU_OK) &&
         (mlgpu_graph_add_output(&g, fbm) == MLGPU_OK) && ok;
    for (Node& nn : g.nodes)  // what graph_compile does before generating code
      if (nn.type == NODE_PROC && mlgpu_proc_rings(nn.kind))
      {
        nn.memOff = g.memFloatsPerVoice;
        nn.ringSlot = g.totalRings;
        g.totalRings += mlgpu_proc_rings(nn.kind);
        g.memFloatsPerVoice += nn.ringLen * (size_t)mlgpu_proc_rings(nn.kind);
      }
    const int muxIn[4] = {gate, hs, ls, lp3};
    const int mux = mlgpu_graph_add_route(&g, MLGPU_ROUTE_MULTIPLEX, muxIn, 4, 0, 0, "mux");
    const int muxl = mlgpu_graph_add_route(&g, MLGPU_ROUTE_MULTIPLEX_LINEAR, muxIn, 4, 0, 0, "muxl");
    const int dmIn[2] = {gate, mux};
    const int dm = mlgpu_graph_add_route(&g, MLGPU_ROUTE_DEMULTIPLEX, dmIn, 2, 1, 3, "dm1");
    const int dmlIn[2] = {gate, muxl};
    const int dml = mlgpu_graph_add_route(&g, MLGPU_ROUTE_DEMULTIPLEX_LINEAR, dmlIn, 2, 2, 3, "dml2");
    ok = (vca > 0) && (hs > 0) && (dm > 0) && (dml > 0) && (mlgpu_graph_add_output(&g, hs) == MLGPU_OK) && (mlgpu_graph_add_output(&g, dm) == MLGPU_OK) &&
         (mlgpu_graph_add_output(&g, dml) == MLGPU_OK) && ok;
    log.clear();
    ok = compileOnly(generateGraphSource(&g), log) && ok;
    all += log;
    // (3) the same graph with per-voice rings behind LDS windows
    g.windowedRings = true;
    g.voicesPerLane = 1;
    g.totalRings = 20;  // the largest LDS footprint graph_compile accepts (160 KiB)
    log.clear();
    ok = compileOnly(generateGraphSource(&g), log) && ok;
    all += log;
    if (logOut && logLen)
    {
      snprintf(logOut, logLen, "%s", all.c_str());
    }
    return ok ? MLGPU_OK : MLGPU_ERR_UNSUPPORTED;
  }

  int mlgpu_graph_create(mlgpu_engine* e, size_t nVoices, mlgpu_graph** out)
  {
    if (!out) return MLGPU_ERR_INVALID;  // e == NULL: a graph for mlgpu_graph_emit only (no device)
    *out = nullptr;
    if (nVoices == 0)
    {
      if (e) e->lastError = "graph_create: zero voices";
      return MLGPU_ERR_INVALID;
    }
    mlgpu_graph* g = new (std::nothrow) mlgpu_graph();
    if (!g) return MLGPU_ERR_OOM;
    g->e = e;
    g->strictSvf = e && e->strictSvf;
    g->V = nVoices;
    *out = g;
    return MLGPU_OK;
  }

  int mlgpu_graph_destroy(mlgpu_graph* g)
  {
    if (!g) return MLGPU_ERR_INVALID;
    if (g->job)  // a compile in flight owns the graph: wait for it (seconds at most), then let it go
    {
      g->job->th.join();
      delete g->job;
      g->job = nullptr;
    }
    {
      std::lock_guard<std::mutex> lock(g_boundMutex);
      g_boundGraphs.erase(g);
    }
    if (g->e)
    {
      hipSetDevice(g->e->device);
      hipStreamSynchronize(g->e->stream);
    }
    if (g->d_waveClock)
    {
      const size_t words = (g->V + 255) / 256 * 4 * 4;
      std::vector<unsigned long long> host(words);
      if (hipMemcpy(host.data(), g->d_waveClock, words * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess)
        if (FILE* f = fopen(g->waveClockPath.c_str(), "wb"))
        {
          fwrite(host.data(), sizeof(unsigned long long), words, f);
          fclose(f);
        }
      hipFree(g->d_waveClock);
    }
    if (g->d_coeffs) hipFree(g->d_coeffs);
    if (g->d_state) hipFree(g->d_state);
    if (g->d_params) hipFree(g->d_params);
    if (g->d_consts) hipFree(g->d_consts);
    if (g->d_mem) hipFree(g->d_mem);
    if (g->tuneEv0) hipEventDestroy(g->tuneEv0);
    if (g->tuneEv1) hipEventDestroy(g->tuneEv1);
    delete g;
    return MLGPU_OK;
  }

  int mlgpu_graph_add_input(mlgpu_graph* g, const char* name)
  {
    if (!g) return -MLGPU_ERR_INVALID;
    if (g->nInputs >= MLGPU_GRAPH_MAX_INPUTS) return -gfail(g, MLGPU_ERR_UNSUPPORTED, "too many graph inputs");
    Node n;
    n.type = NODE_INPUT;
    n.kind = 0;
    n.name = name ? name : "";
    n.slot = g->nInputs;
    const int id = addNode(g, std::move(n));
    if (id >= 0) g->nInputs++;
    return id;
  }
  int mlgpu_graph_add_param(mlgpu_graph* g, const char* name)
  {
    if (!g) return -MLGPU_ERR_INVALID;
    Node n;
    n.type = NODE_PARAM;
    n.kind = 0;
    n.name = name ? name : "";
    n.slot = g->nParams;
    const int id = addNode(g, std::move(n));
    if (id >= 0) g->nParams++;
    return id;
  }
  int mlgpu_graph_add_control(mlgpu_graph* g, const char* name)
  {
    if (!g) return -MLGPU_ERR_INVALID;
    if (g->nControls >= MLGPU_GRAPH_MAX_CONTROLS) return -gfail(g, MLGPU_ERR_UNSUPPORTED, "too many graph control inputs");
    Node n;
    n.type = NODE_CONTROL;
    n.kind = 0;
    n.name = name ? name : "";
    n.slot = g->nControls;
    const int id = addNode(g, std::move(n));
    if (id >= 0) g->nControls++;
    return id;
  }
  // a row of an EventsToSignals object computed inside this graph's kernel instead of read from memory (0: pitch, 1: gate)
  int mlgpu_graph_add_event_row(mlgpu_graph* g, int row, const char* name)
  {
    if (!g) return -MLGPU_ERR_INVALID;
    if (row != 0 && row != 1) return -gfail(g, MLGPU_ERR_UNSUPPORTED, "graph_add_event_row: rows 0 (pitch) and 1 (gate) can be source nodes");
    for (const Node& m : g->nodes)
      if (m.type == NODE_EVENT_ROW && m.slot == row) return -gfail(g, MLGPU_ERR_INVALID, "graph_add_event_row: this row is a node already");
    Node n;
    n.type = NODE_EVENT_ROW;
    n.kind = 0;
    n.name = name ? name : "";
    n.slot = row;
    const int id = addNode(g, std::move(n));
    if (id >= 0) g->hasEventRows = true;
    return id;
  }
  // (file scope, see below) every graph currently bound to an events object, so that destroying the object can unbind them
  int mlgpu_graph_bind_events(mlgpu_graph* g, mlgpu_events* ev)
  {
    if (!g || !ev) return MLGPU_ERR_INVALID;
    if (g->job) return MLGPU_ERR_BUSY;
    if (!g->hasEventRows) return gfail(g, MLGPU_ERR_INVALID, "graph_bind_events: the graph has no event rows (graph_add_event_row)");
    if (mlgpu_events_engine(ev) != g->e) return gfail(g, MLGPU_ERR_INVALID, "graph_bind_events: the events object belongs to another engine");
    if (!mlgpu_events_is_midi(ev)) return gfail(g, MLGPU_ERR_UNSUPPORTED, "graph_bind_events: MIDI protocol only (one lane per voice)");
    if (mlgpu_events_num_voices(ev) != g->V) return gfail(g, MLGPU_ERR_INVALID, "graph_bind_events: instruments x polyphony must equal the graph's voices");
    {
      std::lock_guard<std::mutex> lock(g_boundMutex);  // (mlgpu_graph_forget_events reads and clears g->events under the same lock)
      g_boundGraphs.insert(g);
      g->events = ev;
    }
    return MLGPU_OK;
  }
  int mlgpu_graph_add_vop(mlgpu_graph* g, int vop, const int* inputs, int nIn, const char* name)
  {
    if (!g) return -MLGPU_ERR_INVALID;
    if (vop < MLGPU_VOP_COLUMN_INDEX || vop > MLGPU_VOP_INTERPOLATE_LINEAR) return -gfail(g, MLGPU_ERR_INVALID, "graph_add_vop: unknown generator");
    const int want = (vop == MLGPU_VOP_COLUMN_INDEX) ? 0 : 2;
    if (nIn != want || (nIn > 0 && !inputs)) return -gfail(g, MLGPU_ERR_INVALID, "graph_add_vop: wrong number of inputs");
    for (int j = 0; j < nIn; ++j)
      if (inputs[j] < 0 || inputs[j] >= (int)g->nodes.size() || g->nodes[inputs[j]].rate > RATE_VECTOR)
        return -gfail(g, MLGPU_ERR_INVALID, "graph_add_vop: start / end are floats (a control, param or const node)");
    Node n;
    n.type = NODE_VOP;
    n.kind = vop;
    if (nIn) n.in.assign(inputs, inputs + nIn);
    n.name = name ? name : "";
    return addNode(g, std::move(n));
  }
  int mlgpu_graph_add_const_vector(mlgpu_graph* g, const float* values, const char* name)
  {
    if (!g) return -MLGPU_ERR_INVALID;
    if (!values) return -gfail(g, MLGPU_ERR_INVALID, "graph_add_const_vector: 64 floats");
    Node n;
    n.type = NODE_VOP;
    n.kind = MLGPU_VOP_TABLE;
    n.table.resize(MLGPU_FLOATS_PER_DSPVECTOR);
    memcpy(n.table.data(), values, sizeof(float) * MLGPU_FLOATS_PER_DSPVECTOR);
    n.name = name ? name : "";
    return addNode(g, std::move(n));
  }
  int mlgpu_graph_add_feedback(mlgpu_graph* g, const char* name)
  {
    if (!g) return -MLGPU_ERR_INVALID;
    Node n;
    n.type = NODE_FEEDBACK;
    n.kind = 0;
    n.name = name ? name : "";
    n.ns = MLGPU_FLOATS_PER_DSPVECTOR;
    n.sOff = g->NS;
    const int id = addNode(g, std::move(n));
    if (id >= 0) g->NS += MLGPU_FLOATS_PER_DSPVECTOR;
    return id;
  }
  int mlgpu_graph_set_feedback(mlgpu_graph* g, int fbNode, int valueNode)
  {
    int st = checkNode(g, fbNode, NODE_FEEDBACK);
    if (st) return st;
    if (g->job) return MLGPU_ERR_BUSY;
    if (g->compiled) return gfail(g, MLGPU_ERR_INVALID, "graph already compiled");
    if (valueNode < 0 || valueNode >= (int)g->nodes.size()) return gfail(g, MLGPU_ERR_RANGE, "graph_set_feedback: unknown value node");
    g->nodes[fbNode].fbSource = valueNode;
    return MLGPU_OK;
  }
  int mlgpu_graph_set_max_delay(mlgpu_graph* g, int node, float maxDelayInSamples)
  {
    int st = checkNode(g, node, NODE_PROC);
    if (st) return st;
    if (g->job) return MLGPU_ERR_BUSY;
    if (g->compiled) return gfail(g, MLGPU_ERR_INVALID, "graph already compiled");
    if (mlgpu_proc_rings(g->nodes[node].kind) == 0) return gfail(g, MLGPU_ERR_INVALID, "graph_set_max_delay: not a delay node");
    if (!(maxDelayInSamples >= 0.f) || maxDelayInSamples > 16777216.f) return gfail(g, MLGPU_ERR_RANGE, "graph_set_max_delay: 0 .. 2^24 samples");
    // IntegerDelay::setMaxDelayInSamples, MLDSPFilters.h:823-831
    const int dMax = (int)floorf(maxDelayInSamples);
    int bits = 0;
    while ((1 << bits) < dMax + MLGPU_FLOATS_PER_DSPVECTOR) bits++;
    g->nodes[node].ringLen = (size_t)1 << bits;
    return MLGPU_OK;
  }
  int mlgpu_graph_add_route(mlgpu_graph* g, int route, const int* inputs, int nIn, int index, int nOutputs, const char* name)
  {
    if (!g) return -MLGPU_ERR_INVALID;
    if (route < MLGPU_ROUTE_MULTIPLEX || route > MLGPU_ROUTE_DEMULTIPLEX_LINEAR || !inputs) return -gfail(g, MLGPU_ERR_INVALID, "graph_add_route: unknown routing node");
    const bool mux = (route == MLGPU_ROUTE_MULTIPLEX || route == MLGPU_ROUTE_MULTIPLEX_LINEAR);
    if (mux && (nIn < 2 || nIn > 1 + MLGPU_ROUTE_MAX_SIGNALS)) return -gfail(g, MLGPU_ERR_INVALID, "graph_add_route: multiplex takes a selector and 1..8 signals");
    if (!mux && (nIn != 2 || nOutputs < 1 || nOutputs > MLGPU_ROUTE_MAX_SIGNALS || index < 0 || index >= nOutputs))
      return -gfail(g, MLGPU_ERR_INVALID, "graph_add_route: demultiplex takes (selector, signal), 1..8 outputs, 0 <= index < n_outputs");
    Node n;
    n.type = NODE_ROUTE;
    n.kind = route;
    n.in.assign(inputs, inputs + nIn);
    n.name = name ? name : "";
    n.slot = index;
    n.nOut = nOutputs;
    return addNode(g, std::move(n));
  }
  int mlgpu_graph_add_const(mlgpu_graph* g, float value)
  {
    if (!g) return -MLGPU_ERR_INVALID;
    Node n;
    n.type = NODE_CONST;
    n.kind = 0;
    n.value = value;
    n.slot = g->nConsts;
    const int id = addNode(g, std::move(n));
    if (id >= 0) g->nConsts++;
    return id;
  }
  int mlgpu_graph_add_proc(mlgpu_graph* g, int kind, const int* inputs, int nIn, const char* name)
  {
    if (!g) return -MLGPU_ERR_INVALID;
    const int nc = mlgpu_proc_nc(kind), ns = mlgpu_proc_ns(kind);
    if (nc < 0) return -gfail(g, MLGPU_ERR_INVALID, "graph_add_proc: unknown processor kind");
    if (kind == MLGPU_PROC_HALF_BAND || kind == MLGPU_PROC_HALF_BAND_BUFFERED)
      return -gfail(g, MLGPU_ERR_INVALID, "graph_add_proc: HalfBandFilter nodes are made by graph_begin_region / graph_end_region");
    // forms of operator(): 1 input, plus PulseGen(freq, width) MLDSPGens.h:390, Lopass(x, omega, k) MLDSPFilters.h:136,
    // LoShelf(x, 5 coefficient signals) :304, HiShelf(x, 6 coefficient signals) :385; NoiseGen has none
    bool okArity = (nIn == 1);
    if (kind == MLGPU_PROC_NOISE_GEN) okArity = (nIn == 0 || nIn == 1);
    if (kind == MLGPU_PROC_PULSE_GEN) okArity = (nIn == 1 || nIn == 2);
    if (kind == MLGPU_PROC_LOPASS) okArity = (nIn == 1 || nIn == 3);
    if (kind == MLGPU_PROC_LO_SHELF) okArity = (nIn == 1 || nIn == 6);
    if (kind == MLGPU_PROC_HI_SHELF) okArity = (nIn == 1 || nIn == 7);
    if (kind == MLGPU_PROC_INTEGER_DELAY) okArity = (nIn == 1 || nIn == 2);     // (x), (x, delay) MLDSPFilters.h:834,877
    if (kind == MLGPU_PROC_FRACTIONAL_DELAY) okArity = (nIn >= 1 && nIn <= 3);  // (x), (x, delay), (x, delay, ticks) :1013-1043
    if (kind == MLGPU_PROC_PITCHBENDABLE_DELAY) okArity = (nIn == 2);           // (x, delay) :1098
    if (kind == MLGPU_PROC_TEMPO_LOCK) okArity = (nIn == 3);                    // (x, dydx, isr) :1492
    if (!okArity || (nIn > 0 && !inputs)) return -gfail(g, MLGPU_ERR_INVALID, "graph_add_proc: wrong number of inputs");
    if (kind == MLGPU_PROC_TEMPO_LOCK)
    {
      for (int j = 0; j < 3; ++j)
        if (inputs[j] < 0 || inputs[j] >= (int)g->nodes.size()) return -gfail(g, MLGPU_ERR_RANGE, "graph node input refers to an unknown node");
      if (g->nodes[inputs[1]].rate > RATE_VECTOR || g->nodes[inputs[2]].rate > RATE_VECTOR)
        return -gfail(g, MLGPU_ERR_INVALID, "graph_add_proc: TempoLock(x, dydx, isr): dydx and isr are floats per vector");
    }
    else if (mlgpu_proc_is_vector_rate(kind) && (inputs[0] < 0 || inputs[0] >= (int)g->nodes.size() || g->nodes[inputs[0]].rate > RATE_VECTOR))
      return -gfail(g, MLGPU_ERR_INVALID, "graph_add_proc: Interpolator1 / LinearGlide take one float per DSPVector (a control, param or const node)");
    (void)ns;
    return addProcNode(g, kind, inputs, nIn, name);
  }

  int mlgpu_graph_begin_region(mlgpu_graph* g, int region, const int* inputs, int nIn, int* regionInputs)
  {
    if (!g) return MLGPU_ERR_INVALID;
    if (g->job) return MLGPU_ERR_BUSY;
    if (g->compiled) return gfail(g, MLGPU_ERR_INVALID, "graph already compiled");
    if (region != MLGPU_REGION_UPSAMPLE_2X && region != MLGPU_REGION_DOWNSAMPLE_2X) return gfail(g, MLGPU_ERR_INVALID, "graph_begin_region: unknown region kind");
    if (nIn < 0 || nIn > 8 || (nIn > 0 && (!inputs || !regionInputs))) return gfail(g, MLGPU_ERR_INVALID, "graph_begin_region: 0..8 inputs");
    int depth = 0;
    for (int r = g->openRegion; r >= 0; r = g->regions[(size_t)r].parent) depth++;
    if (depth >= 3) return gfail(g, MLGPU_ERR_UNSUPPORTED, "graph_begin_region: rate regions nest three deep at most");
    for (int j = 0; j < nIn; ++j)
    {
      if (inputs[j] < 0 || inputs[j] >= (int)g->nodes.size()) return gfail(g, MLGPU_ERR_RANGE, "graph_begin_region: unknown input node");
      const Node& src = g->nodes[(size_t)inputs[j]];
      // the inputs of a nested region are signals of the enclosing one (or per-voice floats)
      if (src.region != g->openRegion && !(src.region < 0 && src.rate == RATE_VOICE))
        return gfail(g, MLGPU_ERR_INVALID, "graph_begin_region: an input must be a node of the enclosing region (or of the outer graph for an outermost region)");
    }
    const int r = (int)g->regions.size();
    g->regions.emplace_back();
    g->regions.back().kind = region;
    g->regions.back().parent = g->openRegion;
    for (int j = 0; j < nIn; ++j)
    {
      // one HalfBandFilter per input row: mUppers[j] (MLDSPFunctional.h:125-130) / mDowners[j] (:181-184)
      const int id = addProcNode(g, MLGPU_PROC_HALF_BAND, &inputs[j], 1, nullptr, ROLE_REGION_IN, r);
      if (id < 0) return -id;  // (cannot happen after the checks above; the region then simply stays without a result)
      g->nodes[(size_t)id].rate = RATE_AUDIO;
      g->regions[(size_t)r].ins.push_back(id);
      regionInputs[j] = id;
    }
    g->openRegion = r;
    return MLGPU_OK;
  }

  int mlgpu_graph_end_region(mlgpu_graph* g, int result, const char* name)
  {
    if (!g) return -MLGPU_ERR_INVALID;
    if (g->job) return -MLGPU_ERR_BUSY;
    if (g->compiled) return -gfail(g, MLGPU_ERR_INVALID, "graph already compiled");
    const int r = g->openRegion;
    if (r < 0) return -gfail(g, MLGPU_ERR_INVALID, "graph_end_region: no region is open");
    if (result < 0 || result >= (int)g->nodes.size() || g->nodes[(size_t)result].region != r)
      return -gfail(g, MLGPU_ERR_INVALID, "graph_end_region: the result must be an audio-rate node of the region");
    const int parent = g->regions[(size_t)r].parent;
    g->openRegion = parent;
    g->regions[(size_t)r].result = result;
    // mDowners[0] (MLDSPFunctional.h:137-141) resp. mUppers[0] + mOutputBuffer (:191-197); the node belongs to the enclosing region
    const int kind = (g->regions[(size_t)r].kind == MLGPU_REGION_UPSAMPLE_2X) ? MLGPU_PROC_HALF_BAND : MLGPU_PROC_HALF_BAND_BUFFERED;
    const int id = addProcNode(g, kind, &result, 1, name, ROLE_REGION_OUT, parent, r);
    if (id < 0)
    {
      g->openRegion = r;
      return id;
    }
    g->regions[(size_t)r].out = id;
    return id;
  }
  int mlgpu_graph_add_op(mlgpu_graph* g, int op, const int* inputs, int nIn, const char* name)
  {
    if (!g) return -MLGPU_ERR_INVALID;
    if (!opKnown(op)) return -gfail(g, MLGPU_ERR_INVALID, "graph_add_op: unknown op");
    if (nIn != opArity(op) || !inputs) return -gfail(g, MLGPU_ERR_INVALID, "graph_add_op: wrong number of inputs");
    Node n;
    n.type = NODE_OP;
    n.kind = op;
    n.in.assign(inputs, inputs + nIn);
    n.name = name ? name : "";
    return addNode(g, std::move(n));
  }
  int mlgpu_graph_add_output(mlgpu_graph* g, int node)
  {
    if (!g) return MLGPU_ERR_INVALID;
    if (g->job) return MLGPU_ERR_BUSY;
    if (g->compiled) return gfail(g, MLGPU_ERR_INVALID, "graph already compiled");
    if (node < 0 || node >= (int)g->nodes.size()) return gfail(g, MLGPU_ERR_RANGE, "graph_add_output: unknown node");
    if (g->outputs.size() >= MLGPU_GRAPH_MAX_OUTPUTS) return gfail(g, MLGPU_ERR_UNSUPPORTED, "too many graph outputs");
    g->outputs.push_back(node);
    return MLGPU_OK;
  }
  // Synth::processVector's voice sum (source/app/MLSynth.h:43-57) as an output mode: output `index` becomes a signal of
  // voices / group channels, channel c = ((0 + voice[c * group]) + voice[c * group + 1]) + ... in that order
  int mlgpu_graph_set_output_group_sum(mlgpu_graph* g, int index, int group)
  {
    if (!g) return MLGPU_ERR_INVALID;
    if (g->job) return MLGPU_ERR_BUSY;
    if (g->compiled) return gfail(g, MLGPU_ERR_INVALID, "graph already compiled");
    if (index < 0 || index >= (int)g->outputs.size()) return gfail(g, MLGPU_ERR_RANGE, "graph_set_output_group_sum: no such output");
    if (group != 0 && group != 2 && group != 4 && group != 8 && group != 16)
      return gfail(g, MLGPU_ERR_UNSUPPORTED, "graph_set_output_group_sum: groups of 2, 4, 8 or 16 voices (other sizes: mlgpu_mixdown_groups)");
    if (group && g->V % (size_t)group) return gfail(g, MLGPU_ERR_INVALID, "graph_set_output_group_sum: the voices are not a whole number of groups");
    if (group && g->outputMix[index]) return gfail(g, MLGPU_ERR_INVALID, "graph_set_output_group_sum: the output is a mixdown already");
    g->outputGroup[index] = group;
    return MLGPU_OK;
  }
  // Output `index` becomes ONE channel: the mixdown of all voices (mlgpu_mixdown's order and bits, its first stage inside the voice
  // kernel - the voices' signal of that output is never written). graph_process then wants 64 * n_vectors floats for it, whatever
  // the output layout, and scratch reserved with mlgpu_mixdown_reserve(engine, voices x mixed outputs, vectors).
  int mlgpu_graph_set_output_mixdown(mlgpu_graph* g, int index, int on)
  {
    if (!g) return MLGPU_ERR_INVALID;
    if (g->job) return MLGPU_ERR_BUSY;
    if (g->compiled) return gfail(g, MLGPU_ERR_INVALID, "graph already compiled");
    if (index < 0 || index >= (int)g->outputs.size()) return gfail(g, MLGPU_ERR_RANGE, "graph_set_output_mixdown: no such output");
    if (on && g->outputGroup[index]) return gfail(g, MLGPU_ERR_INVALID, "graph_set_output_mixdown: the output is a group sum already");
    if (on == 2 && mlgpu_mixdown_shard_level(g->V) == 0)
      return gfail(g, MLGPU_ERR_INVALID, "graph_set_output_mixdown: the shard form needs a voice count that is a multiple of 64 (whole first-stage groups of the tree)");
    g->outputMix[index] = on != 0;
    g->outputMixShard[index] = on == 2;
    return MLGPU_OK;
  }
  // setup: the engine's mixdown scratch for this graph's mixed-down outputs, launches of up to maxVectors DSPVectors
  int mlgpu_graph_reserve_mixdown(mlgpu_graph* g, size_t maxVectors)
  {
    if (!g) return MLGPU_ERR_INVALID;
    if (!g->e) return gfail(g, MLGPU_ERR_INVALID, "graph_reserve_mixdown: a graph without an engine");
    size_t nMix = 0;
    for (size_t o = 0; o < g->outputs.size(); ++o) nMix += g->outputMix[o] ? 1 : 0;
    const size_t groups = (g->V + 63) / 64;
    const int st = mlgpu_mixdown_reserve_floats(g->e, nMix * (groups + (groups + 63) / 64) * maxVectors * 64);
    return st == MLGPU_OK ? st : gfail(g, st, "graph_reserve_mixdown: see the engine's last error");
  }
  int mlgpu_graph_node(mlgpu_graph* g, const char* name)
  {
    if (!g || !name) return -MLGPU_ERR_INVALID;
    if (g->job) return -MLGPU_ERR_BUSY;  // (the worker reads the names; and a name may be set below by this thread only)
    for (size_t i = 0; i < g->nodes.size(); ++i)
      if (g->nodes[i].name == name) return (int)i;
    return -MLGPU_ERR_RANGE;
  }
  int mlgpu_graph_num_nodes(mlgpu_graph* g) { return g ? (int)g->nodes.size() : -1; }
  int mlgpu_graph_set_node_name(mlgpu_graph* g, int node, const char* name)
  {
    if (g && g->job) return MLGPU_ERR_BUSY;  // (code generation prints the names into the source's comments)
    if (!g || !name || node < 0 || node >= (int)g->nodes.size()) return MLGPU_ERR_RANGE;
    g->nodes[(size_t)node].name = name;
    return MLGPU_OK;
  }
  int mlgpu_graph_node_kind(mlgpu_graph* g, int node)
  {
    if (!g || node < 0 || node >= (int)g->nodes.size()) return -MLGPU_ERR_RANGE;
    const Node& n = g->nodes[(size_t)node];
    return (n.type == NODE_PROC || n.type == NODE_OP || n.type == NODE_VOP) ? n.kind : -MLGPU_ERR_INVALID;
  }
  int mlgpu_graph_node_use_count(mlgpu_graph* g, int node)
  {
    if (!g || node < 0 || node >= (int)g->nodes.size()) return -MLGPU_ERR_RANGE;
    int n = 0;
    for (const Node& m : g->nodes)
    {
      for (int id : m.in) n += (id == node);
      n += (m.type == NODE_FEEDBACK && m.fbSource == node);
    }
    for (int o : g->outputs) n += (o == node);
    return n;
  }

  // ring placement + code generation: everything graph_compile does that needs no device
  static int layoutAndGenerate(mlgpu_graph* g)
  {
    if (!g->source.empty()) return MLGPU_OK;
    if (g->outputs.empty()) return gfail(g, MLGPU_ERR_INVALID, "graph_compile: no outputs");
    if (g->openRegion >= 0) return gfail(g, MLGPU_ERR_INVALID, "graph_compile: a rate region is still open (graph_end_region)");
    for (int o : g->outputs)
      if (g->nodes[(size_t)o].region >= 0) return gfail(g, MLGPU_ERR_INVALID, "graph_compile: an output is a node inside a rate region");
    for (const Node& n : g->nodes)
      if (n.type == NODE_FEEDBACK && n.fbSource >= 0 && g->nodes[(size_t)n.fbSource].region != n.region)
        return gfail(g, MLGPU_ERR_INVALID, "graph_compile: a feedback node and its source must be in the same rate region (or both outside)");
    size_t memFloats = 0;
    g->totalRings = 0;
    for (Node& n : g->nodes)
    {
      if (n.type == NODE_FEEDBACK && n.fbSource < 0) return gfail(g, MLGPU_ERR_INVALID, "graph_compile: feedback node '" + n.name + "' has no source (graph_set_feedback)");
      if (n.type != NODE_PROC || mlgpu_proc_rings(n.kind) == 0) continue;
      if (n.ringLen == 0) return gfail(g, MLGPU_ERR_INVALID, "graph_compile: delay node '" + n.name + "' has no memory (graph_set_max_delay)");
      n.memOff = memFloats;
      n.ringSlot = g->totalRings;
      g->totalRings += mlgpu_proc_rings(n.kind);
      memFloats += n.ringLen * (size_t)mlgpu_proc_rings(n.kind);
    }
    g->memFloatsPerVoice = memFloats;
    // (three rings: layout 2 fits but leaves a CU one workgroup, and layout 1 is 9 % faster - profiles/r05_ring_layouts.txt)
    // a bank whose last wavefront is not full: its spare lanes run the last voice again (generateGraphSource) - not where voices are
    // summed in groups inside the kernel or read event records, which go by lane
    bool groupedOrEvents = g->hasEventRows;
    for (size_t o = 0; o < g->outputs.size(); ++o) groupedOrEvents = groupedOrEvents || g->outputGroup[o] != 0;
    const bool partialOk = g->V % 64 == 0 || !groupedOrEvents;
    if (g->transposedRings && g->totalRings && !partialOk)
      return gfail(g, MLGPU_ERR_UNSUPPORTED, "graph_compile: delay layout 2 with voice sums or event rows inside the kernel needs whole wavefronts (voices a multiple of 64)");
    for (size_t o = 0; o < g->outputs.size(); ++o)
      if (g->outputMix[o] && !partialOk)
        return gfail(g, MLGPU_ERR_UNSUPPORTED, "graph_compile: an output that is the mixdown of all voices, next to group sums or event rows, needs whole wavefronts (voices a multiple of 64)");
    // LDS of a workgroup: the ring strips, the impulse table and a strip per output that is summed inside the kernel (a whole-bank
    // mixdown: 4 wavefronts x kMixStrip floats = 21 KiB; a 16-voice group sum: 4 x kGroup16Strip = 20.3 KiB). A layout that does not fit
    // next to them falls back (layout 3) or is refused here with the sizes, not by hiprtc / the module loader.
    size_t ldsOther = g->hasImpulse ? 128 : 0;
    {
      const bool dppSum = getenv("MLGPU_GRAPH_GROUP_SUM") && !strcmp(getenv("MLGPU_GRAPH_GROUP_SUM"), "dpp");
      for (size_t o = 0; o < g->outputs.size(); ++o)
      {
        if (g->outputMix[o]) ldsOther += sizeof(float) * 4 * (size_t)kHostMixStripFloats;
        else if (g->outputGroup[o] == 16 && !dppSum) ldsOther += sizeof(float) * 4 * (size_t)kHostGroup16StripFloats;
      }
    }
    constexpr size_t kLdsBytes = 160 * 1024;
    const size_t ldsLayout2 = (size_t)g->totalRings * 4 * 40 * 64 * sizeof(float), ldsLayout1 = (size_t)g->totalRings * 8 * 256 * sizeof(float);
    size_t ldsLayout4 = 0;  // per workgroup: 2 KiB per ring and wavefront (the held sector) + 4 KiB per delay node and wavefront (its last 16 samples)
    for (const Node& n : g->nodes)
      if (n.type == NODE_PROC && mlgpu_proc_rings(n.kind)) ldsLayout4 += 4 * sizeof(float) * ((size_t)mlgpu_proc_rings(n.kind) * 512 + 1024);
    auto kib = [](size_t b) { return std::to_string((b + 1023) / 1024) + " KiB"; };
    // layout 4 (sector trips) serves delay nodes of the outer graph; one inside a rate region keeps layout 1's per-sample form
    bool ringInRegion = false;
    for (const Node& n : g->nodes) ringInRegion = ringInRegion || (n.type == NODE_PROC && n.region >= 0 && mlgpu_proc_rings(n.kind) != 0);
    if (g->sectorRings && ringInRegion)
      return gfail(g, MLGPU_ERR_UNSUPPORTED, "graph_compile: delay layout 4 (sector trips) does not serve a delay line inside a rate region (layout 1 or 3 for this graph)");
    if (g->transposedIfPossible)
    {
      // "the best form": one or two rings - the transposed windows (0.72-0.74 of the HBM peak on the strings bank); more - the sector
      // trips (no LDS, every ring's loads in the trip's prologue: profiles/r06_ring_layouts.txt); where neither applies, layout 1
      // (measured, profiles/r06_ring_layouts.txt: one PitchbendableDelay 0.74 of the HBM peak in layout 4 - it keeps one ring and makes
      // one read for both cores - against 0.58 in layout 2; one / two FractionalDelays 0.63 / 0.60 in layout 2 against 0.50 / 0.40;
      // three / four 0.42 / 0.35 in layout 4 against 0.33 / 0.32 in layout 2 and 0.34 / 0.20 in layout 1)
      bool anyPitchbendable = false;
      for (const Node& n : g->nodes) anyPitchbendable = anyPitchbendable || (n.type == NODE_PROC && n.kind == MLGPU_PROC_PITCHBENDABLE_DELAY);
      const bool sectorFits = !ringInRegion && g->totalRings > 0 && ldsLayout4 + ldsOther <= kLdsBytes;
      const bool preferSectors = sectorFits && (anyPitchbendable || g->totalRings > 2);
      g->transposedRings = !preferSectors && partialOk && g->totalRings <= 4 && ldsLayout2 + ldsOther <= kLdsBytes;
      g->sectorRings = !g->transposedRings && sectorFits;
      // more rings than any windowed form has LDS for (the reference's reverb example: 24): the default rows
      if (!g->transposedRings && !g->sectorRings && ldsLayout1 + ldsOther > kLdsBytes) g->windowedRings = false;
    }
    if (g->transposedRings && ldsLayout2 + ldsOther > kLdsBytes)
      return gfail(g, MLGPU_ERR_UNSUPPORTED, "graph_compile: delay layout 2 needs 40 KiB of LDS per ring (" + kib(ldsLayout2) + " for " + std::to_string(g->totalRings) +
                                                 " rings) next to " + kib(ldsOther) + " of output strips and tables; a workgroup has 160 KiB (layout 1 or 3 for this graph)");
    if (g->sectorRings && ldsLayout4 + ldsOther > kLdsBytes)
      return gfail(g, MLGPU_ERR_UNSUPPORTED, "graph_compile: delay layout 4 needs 8 KiB of LDS per ring and 16 KiB per delay node (" + kib(ldsLayout4) + " for this graph) next to " +
                                                 kib(ldsOther) + " of output strips and tables; a workgroup has 160 KiB (layout 1 or 3 for this graph)");
    if (!g->transposedRings && !g->sectorRings && g->windowedRings && ldsLayout1 + ldsOther > kLdsBytes)
      return gfail(g, MLGPU_ERR_UNSUPPORTED, "graph_compile: delay layouts 1 and 4 need 8 KiB of LDS per ring (" + kib(ldsLayout1) + " for " + std::to_string(g->totalRings) +
                                                 " rings) next to " + kib(ldsOther) + " of output strips and tables; a workgroup has 160 KiB");
    if (ldsOther > kLdsBytes)
      return gfail(g, MLGPU_ERR_UNSUPPORTED, "graph_compile: " + kib(ldsOther) + " of LDS for the outputs summed inside the kernel (21 KiB per mixed-down output, 20.3 KiB per 16-voice group sum); a workgroup has 160 KiB");
    // ring layout 0: rows behind 32-bit offsets where every delay node's memory stays below 4 GiB (VoiceMem::ringPtr)
    g->rowAddr32 = false;
    if (!g->windowedRings && g->totalRings && g->V < ((size_t)1 << 22) && !(getenv("MLGPU_GRAPH_ROW_ADDR32") && !strcmp(getenv("MLGPU_GRAPH_ROW_ADDR32"), "0")))
      g->rowAddr32 = true;  // (node by node in the generator: a ring of the bank at most 4 GiB)
    // ring layout 0: the outer graph's ring reads by LDS-DMA ahead of the sample's arithmetic, a 256-byte landing slot per read and wavefront
    g->earlyRows = false;
    g->earlySlots = 0;
    for (Node& n : g->nodes) n.earlySlot = -1;
    {
      const char* er = getenv("MLGPU_GRAPH_EARLY_READS");  // developer knob (A / B): 0 = the plain loads
      if (!g->windowedRings && g->totalRings && !(er && !strcmp(er, "0")))
      {
        int slots = 0;
        for (Node& n : g->nodes)
          if (n.type == NODE_PROC && n.region < 0 && n.role == ROLE_NONE && mlgpu_proc_rings(n.kind))
          {
            n.earlySlot = slots;
            slots += n.kind == MLGPU_PROC_PITCHBENDABLE_DELAY ? 2 : 1;
          }
        // (one ring - a plucked string - has nothing to issue together: 0.127 of the peak with the early read against 0.142 without)
        if (slots >= 3 && (size_t)slots * 4 * 64 * sizeof(float) + ldsOther <= kLdsBytes)
        {
          g->earlyRows = true;
          g->earlySlots = slots;
        }
        else
          for (Node& n : g->nodes) n.earlySlot = -1;
      }
    }
    const char* forced = getenv("MLGPU_GRAPH_UNROLL");  // developer knob: quads per trip of the sample loop
    // delay graphs wait on their ring reads: two quads per trip keep more of them in flight (allpass4: 5.4 vs 4.5 x 10^10)
    g->unrollQ = forced ? std::max(1, atoi(forced)) : ((g->totalRings && !g->windowedRings) ? 2 : 1);
    if (const char* wc = getenv("MLGPU_GRAPH_WAVE_CLOCK")) g->waveClockPath = wc;
    if (const char* tt = getenv("MLGPU_GRAPH_TURNS")) g->takeTurns = std::min(2, std::max(0, atoi(tt)));
    if (const char* tc = getenv("MLGPU_GRAPH_TURN_CLOCK")) g->turnClockShift = std::min(24, std::max(0, atoi(tc)));
    if (const char* lk = getenv("MLGPU_GRAPH_LOCK_OSC")) g->lockOscillators = atoi(lk) != 0;
    if (const char* pf = getenv("MLGPU_GRAPH_PREFETCH")) g->prefetchQ = atoi(pf) != 0;  // developer knob (A / B)
    if (const char* fa = getenv("MLGPU_GRAPH_FB_AHEAD")) g->fbAhead = atoi(fa) != 0;
    if (const char* trip = getenv("MLGPU_GRAPH_OSC_TRIP"))  // developer knob: 0 = polyBLEP per sample (A / B), else 1, 2 or 4 quads per trip
    {
      const int t = atoi(trip);
      g->oscTripQ = (t == 1 || t == 2 || t == 4) ? t : 0;
    }
    if (g->sectorRings && g->totalRings && g->oscTripQ > 0) g->oscTripQ = 2;  // (one trip structure: the rings' trips are two quads)
    if (!generateBudgeted(g, 0, g->source, g->emitted, g->log)) return gfail(g, MLGPU_ERR_UNSUPPORTED, "graph_compile (hiprtc): " + g->log);
    return MLGPU_OK;
  }

  // (while a compile is in flight the strings are the worker's to write: "" until mlgpu_graph_compile_poll has collected the job)
  const char* mlgpu_graph_last_error(mlgpu_graph* g) { return (g && !g->job) ? g->lastError.c_str() : ""; }

  int mlgpu_graph_emit(mlgpu_graph* g, const void** code, size_t* codeSize)
  {
    if (!g) return MLGPU_ERR_INVALID;
    if (g->job) return MLGPU_ERR_BUSY;
    const int st = layoutAndGenerate(g);
    if (st != MLGPU_OK) return st;
    if (g->emitted.empty()) return gfail(g, MLGPU_ERR_UNSUPPORTED, "graph_emit (hiprtc): " + g->log);
    if (code) *code = g->emitted.data();
    if (codeSize) *codeSize = g->emitted.size();
    return MLGPU_OK;
  }

  // the part of a compile that takes seconds and needs no stream: code generation, hiprtc (or the caches), module load
  static int compileBuild(mlgpu_graph* g)
  {
    mlgpu_engine* e = g->e;
    const int st = layoutAndGenerate(g);
    if (st != MLGPU_OK) return st;
    if (!e) return MLGPU_OK;  // ahead of time: the code is in the memory and disk caches now (mlgpu_graph_compile_async on a graph without an engine)
    if (hipSetDevice(e->device) != hipSuccess) return gfail(g, MLGPU_ERR_HIP, "hipSetDevice");
    CompiledModule* cm = compileAndLoad(e->device, g->source, g->log);
    if (!cm) return gfail(g, MLGPU_ERR_UNSUPPORTED, "graph_compile (hiprtc): " + g->log);
    g->fn = getFunction(cm, "mlgpu_graph_kernel", g->log);
    if (!g->fn) return gfail(g, MLGPU_ERR_HIP, g->log);
    return MLGPU_OK;
  }
  static int compileFinish(mlgpu_graph* g);

  int mlgpu_graph_compile(mlgpu_graph* g)
  {
    if (!g) return MLGPU_ERR_INVALID;
    if (g->job) return MLGPU_ERR_BUSY;  // (not gfail: the graph is the job's until mlgpu_graph_compile_poll has collected it)
    if (g->compiled) return MLGPU_OK;
    if (!g->e) return gfail(g, MLGPU_ERR_INVALID, "graph_compile: the graph was created without an engine (graph_emit only)");
    const int st = compileBuild(g);
    if (st != MLGPU_OK) return st;
    return compileFinish(g);
  }

  int mlgpu_graph_compile_async(mlgpu_graph* g)
  {
    if (!g) return MLGPU_ERR_INVALID;
    if (g->job) return MLGPU_ERR_BUSY;
    if (g->compiled || g->aotDone) return MLGPU_OK;
    mlgpu_graph::CompileJob* job = new (std::nothrow) mlgpu_graph::CompileJob();
    if (!job) return MLGPU_ERR_OOM;
    g->job = job;
    try
    {
      job->th = std::thread([g, job]() {
        t_compileWorker = true;
        job->status = compileBuild(g);
        job->error = g->lastError;
        job->done.store(true, std::memory_order_release);
      });
    }
    catch (...)
    {
      g->job = nullptr;
      delete job;
      return gfail(g, MLGPU_ERR_OOM, "graph_compile_async: could not start a thread");
    }
    return MLGPU_OK;
  }

  int mlgpu_graph_compile_poll(mlgpu_graph* g)
  {
    if (!g) return MLGPU_ERR_INVALID;
    if (!g->job) return (g->compiled || g->aotDone) ? MLGPU_OK : gfail(g, MLGPU_ERR_INVALID, "graph_compile_poll: no compile in flight (mlgpu_graph_compile_async)");
    if (!g->job->done.load(std::memory_order_acquire)) return MLGPU_ERR_BUSY;
    mlgpu_graph::CompileJob* job = g->job;
    job->th.join();
    g->job = nullptr;
    const int st = job->status;
    const std::string err = job->error;
    delete job;
    if (st != MLGPU_OK) return gfail(g, st, err);
    if (!g->e)
    {
      g->aotDone = true;  // ahead of time: nothing to allocate, the graph stays a description (and every later poll says OK)
      return MLGPU_OK;
    }
    return compileFinish(g);     // allocations and the initial fills, on the caller's thread and stream: microseconds
  }

  static int compileFinish(mlgpu_graph* g)
  {
    mlgpu_engine* e = g->e;
    if (hipSetDevice(e->device) != hipSuccess) return gfail(g, MLGPU_ERR_HIP, "hipSetDevice");
    g->activeVl = g->compiledVoicesPerLane;
    if (g->autotune)
    {
      // candidates: 1 or 2 voices per lane (where the graph allows two and the caller did not force one), 1 or 2 quads per trip
      const bool twoOk = g->voicesPerLane == 0 && [&] {
        for (size_t o = 0; o < g->outputs.size(); ++o)
          if (g->outputMix[o]) return false;
        for (const Node& n : g->nodes)
          if (n.type == NODE_FEEDBACK || n.type == NODE_EVENT_ROW || (n.type == NODE_PROC && (mlgpu_proc_rings(n.kind) || mlgpu_proc_is_vector_rate(n.kind)))) return false;
        return true;
      }();
      const bool unrollFree = !(g->windowedRings && g->totalRings);
      for (int vl = 1; vl <= (twoOk ? 2 : 1); ++vl)
        for (int u = 1; u <= (unrollFree ? 2 : 1); ++u)
        {
          mlgpu_graph::Variant v;
          v.vl = g->voicesPerLane > 0 ? g->voicesPerLane : vl;
          v.unroll = u;
          if (v.vl == g->activeVl && u == g->unrollQ) v.fn = g->fn;  // the default variant is already built
          g->variants.push_back(v);
        }
      g->tuned = g->variants.size() < 2;
    }
    const size_t V = g->V;
    hipError_t err = hipMalloc((void**)&g->d_coeffs, sizeof(float) * V * (size_t)(g->NC + 1));
    if (err == hipSuccess) err = hipMalloc((void**)&g->d_state, sizeof(uint32_t) * V * (size_t)(g->NS + 1));
    if (err == hipSuccess) err = hipMalloc((void**)&g->d_params, sizeof(float) * V * (size_t)(g->nParams + 1));
    const size_t memV = g->memVoices();
    if (err == hipSuccess && g->memFloatsPerVoice) err = hipMalloc((void**)&g->d_mem, sizeof(float) * memV * g->memFloatsPerVoice);
    if (err == hipSuccess && g->memFloatsPerVoice) err = hipMemsetAsync(g->d_mem, 0, sizeof(float) * memV * g->memFloatsPerVoice, e->stream);
    if (err == hipSuccess) err = hipMemsetAsync(g->d_state, 0, sizeof(uint32_t) * V * (size_t)(g->NS + 1), e->stream);
    if (err == hipSuccess) err = hipMemsetAsync(g->d_coeffs, 0, sizeof(float) * V * (size_t)(g->NC + 1), e->stream);
    if (err == hipSuccess) err = hipMemsetAsync(g->d_params, 0, sizeof(float) * V * (size_t)(g->nParams + 1), e->stream);
    if (err == hipSuccess && g->liveConsts)
    {
      err = hipMalloc((void**)&g->d_consts, sizeof(float) * (size_t)(g->nConsts + 1));
      for (const Node& n : g->nodes)
        if (n.type == NODE_CONST && err == hipSuccess)
        {
          uint32_t u;
          memcpy(&u, &n.value, 4);
          err = mlgpu_launch_fill32((uint32_t*)g->d_consts + n.slot, u, 1, e->stream);
        }
    }
    for (const Node& n : g->nodes)
    {
      if (n.type != NODE_PROC) continue;
      float dc[MLGPU_MAX_PROC_COEFFS];
      mlgpu_proc_default_coeffs(n.kind, dc);
      for (int i = 0; i < n.nc && err == hipSuccess; ++i)
      {
        uint32_t u;
        memcpy(&u, &dc[i], 4);
        if (u) err = mlgpu_launch_fill32((uint32_t*)g->d_coeffs + (size_t)(n.cOff + i) * V, u, V, e->stream);
      }
      uint32_t words[MLGPU_MAX_PROC_STATE];
      mlgpu_proc_clear_state(n.kind, words, false);
      for (int i = 0; i < n.ns && err == hipSuccess; ++i)
        err = mlgpu_launch_fill32(g->d_state + (size_t)(n.sOff + i) * V, words[i], V, e->stream);
    }
    if (err != hipSuccess) return gfail(g, err == hipErrorOutOfMemory ? MLGPU_ERR_OOM : MLGPU_ERR_HIP, std::string("graph_compile: ") + hipGetErrorString(err));
    g->compiled = true;
    return MLGPU_OK;
  }

  const char* mlgpu_graph_source(mlgpu_graph* g) { return (g && !g->job) ? g->source.c_str() : ""; }

  // T::clear() of one node: the state words clear() resets (mlgpu_proc_clear_mask), a delay node's rings, a
  // feedback node's stored vector
  static int clearNode(mlgpu_graph* g, const Node& n)
  {
    hipError_t err = hipSetDevice(g->e->device);  // (a host thread may drive engines on several devices in turn)
    if (n.type == NODE_FEEDBACK)
    {
      for (int i = 0; i < n.ns && err == hipSuccess; ++i) err = mlgpu_launch_fill32(g->d_state + (size_t)(n.sOff + i) * g->V, 0u, g->V, g->e->stream);
    }
    else if (n.type == NODE_PROC)
    {
      uint32_t words[MLGPU_MAX_PROC_STATE];
      mlgpu_proc_clear_state(n.kind, words, true);
      const uint64_t mask = mlgpu_proc_clear_mask(n.kind);
      for (int i = 0; i < n.ns && err == hipSuccess; ++i)
        if ((mask >> (i < 64 ? i : 63)) & 1) err = mlgpu_launch_fill32(g->d_state + (size_t)(n.sOff + i) * g->V, words[i], g->V, g->e->stream);
      if (err == hipSuccess && n.ringLen)
        err = mlgpu_launch_fill32((uint32_t*)g->d_mem + n.memOff * g->memVoices(), 0u, n.ringLen * (size_t)mlgpu_proc_rings(n.kind) * g->memVoices(), g->e->stream);
    }
    if (err != hipSuccess) return gfail(g, MLGPU_ERR_HIP, hipGetErrorString(err));
    return MLGPU_OK;
  }

  int mlgpu_graph_clear(mlgpu_graph* g)
  {
    if (!g) return MLGPU_ERR_INVALID;
    if (g->job) return MLGPU_ERR_BUSY;
    if (!g->compiled) return MLGPU_ERR_INVALID;
    for (const Node& n : g->nodes)
    {
      const int st = clearNode(g, n);
      if (st) return st;
    }
    g->vectorCount = 0;
    return MLGPU_OK;
  }

  int mlgpu_graph_clear_proc(mlgpu_graph* g, int node)
  {
    if (!g) return MLGPU_ERR_INVALID;
    if (g->job) return MLGPU_ERR_BUSY;
    if (node < 0 || node >= (int)g->nodes.size()) return gfail(g, MLGPU_ERR_RANGE, "node index out of range");
    if (g->nodes[node].type != NODE_PROC && g->nodes[node].type != NODE_FEEDBACK) return gfail(g, MLGPU_ERR_INVALID, "graph_clear_proc: not a processor / feedback node");
    if (!g->compiled) return gfail(g, MLGPU_ERR_INVALID, "graph_clear_proc: compile first");
    return clearNode(g, g->nodes[node]);
  }

  int mlgpu_graph_set_param(mlgpu_graph* g, int node, const float* h)
  {
    int st = checkNode(g, node, NODE_PARAM);
    if (st) return st;
    if (!g->compiled || !h) return gfail(g, MLGPU_ERR_INVALID, "graph_set_param: compile first / null");
    return mlgpu_upload(g->e, g->d_params + (size_t)g->nodes[node].slot * g->V, h, sizeof(float) * g->V);
  }
  int mlgpu_graph_set_param_uniform(mlgpu_graph* g, int node, float value)
  {
    int st = checkNode(g, node, NODE_PARAM);
    if (st) return st;
    if (!g->compiled) return gfail(g, MLGPU_ERR_INVALID, "graph_set_param: compile first");
    uint32_t u;
    memcpy(&u, &value, 4);
    return mlgpu_fill32(g->e, g->d_params + (size_t)g->nodes[node].slot * g->V, u, g->V);
  }
  int mlgpu_graph_num_coeffs(mlgpu_graph* g, int node) { return checkNode(g, node, NODE_PROC) ? -1 : g->nodes[node].nc; }
  int mlgpu_graph_num_state(mlgpu_graph* g, int node) { return checkStateNode(g, node) ? -1 : g->nodes[node].ns; }
  int mlgpu_graph_set_coeff(mlgpu_graph* g, int node, int idx, const float* h)
  {
    int st = checkNode(g, node, NODE_PROC);
    if (st) return st;
    if (!g->compiled || !h) return gfail(g, MLGPU_ERR_INVALID, "graph_set_coeff: compile first / null");
    if (idx < 0 || idx >= g->nodes[node].nc) return gfail(g, MLGPU_ERR_RANGE, "coefficient index out of range");
    return mlgpu_upload(g->e, g->d_coeffs + (size_t)(g->nodes[node].cOff + idx) * g->V, h, sizeof(float) * g->V);
  }
  int mlgpu_graph_set_coeff_uniform(mlgpu_graph* g, int node, int idx, float value)
  {
    int st = checkNode(g, node, NODE_PROC);
    if (st) return st;
    if (!g->compiled) return gfail(g, MLGPU_ERR_INVALID, "graph_set_coeff: compile first");
    if (idx < 0 || idx >= g->nodes[node].nc) return gfail(g, MLGPU_ERR_RANGE, "coefficient index out of range");
    uint32_t u;
    memcpy(&u, &value, 4);
    return mlgpu_fill32(g->e, g->d_coeffs + (size_t)(g->nodes[node].cOff + idx) * g->V, u, g->V);
  }
  int mlgpu_graph_get_state(mlgpu_graph* g, int node, int idx, uint32_t* h)
  {
    int st = checkStateNode(g, node);
    if (st) return st;
    if (!g->compiled || !h) return gfail(g, MLGPU_ERR_INVALID, "graph_get_state: compile first / null");
    if (idx < 0 || idx >= g->nodes[node].ns) return gfail(g, MLGPU_ERR_RANGE, "state index out of range");
    return mlgpu_download(g->e, h, g->d_state + (size_t)(g->nodes[node].sOff + idx) * g->V, sizeof(uint32_t) * g->V);
  }
  int mlgpu_graph_set_state(mlgpu_graph* g, int node, int idx, const uint32_t* h)
  {
    int st = checkStateNode(g, node);
    if (st) return st;
    if (!g->compiled || !h) return gfail(g, MLGPU_ERR_INVALID, "graph_set_state: compile first / null");
    if (idx < 0 || idx >= g->nodes[node].ns) return gfail(g, MLGPU_ERR_RANGE, "state index out of range");
    return mlgpu_upload(g->e, g->d_state + (size_t)(g->nodes[node].sOff + idx) * g->V, h, sizeof(uint32_t) * g->V);
  }

  int mlgpu_graph_set_delay_layout(mlgpu_graph* g, int windowed)
  {
    if (!g) return MLGPU_ERR_INVALID;
    if (g->job) return MLGPU_ERR_BUSY;
    if (g->compiled) return gfail(g, MLGPU_ERR_INVALID, "graph already compiled");
    if (windowed < 0 || windowed > 4)
      return gfail(g, MLGPU_ERR_INVALID, "graph_set_delay_layout: 0 (rows), 1 (32-byte sectors), 2 (transposed 64-byte pieces), 4 (sector trips) or 3 (the best of 2 / 4 / 1 for the graph)");
    g->windowedRings = windowed != 0;
    g->transposedRings = windowed == 2;
    g->sectorRings = windowed == 4;
    g->transposedIfPossible = windowed == 3;   // decided at compile, when the number of rings is known
    return MLGPU_OK;
  }

  int mlgpu_graph_set_autotune(mlgpu_graph* g, int on)
  {
    if (!g) return MLGPU_ERR_INVALID;
    if (g->job) return MLGPU_ERR_BUSY;
    if (g->compiled) return gfail(g, MLGPU_ERR_INVALID, "graph already compiled");
    g->autotune = on != 0;
    return MLGPU_OK;
  }
  int mlgpu_graph_set_live_constants(mlgpu_graph* g, int on)
  {
    if (!g) return MLGPU_ERR_INVALID;
    if (g->job) return MLGPU_ERR_BUSY;
    if (g->compiled) return gfail(g, MLGPU_ERR_INVALID, "graph_set_live_constants: before mlgpu_graph_compile");
    g->liveConsts = on != 0;
    return MLGPU_OK;
  }
  int mlgpu_graph_set_const(mlgpu_graph* g, int node, float value)
  {
    if (!g) return MLGPU_ERR_INVALID;
    if (g->job) return MLGPU_ERR_BUSY;  // (the worker turns the values into literals of the kernel)
    if (node < 0 || node >= (int)g->nodes.size() || g->nodes[node].type != NODE_CONST) return gfail(g, MLGPU_ERR_INVALID, "graph_set_const: not a const node");
    if (g->compiled && !g->liveConsts)
      return gfail(g, MLGPU_ERR_INVALID, "graph_set_const: constants of this graph are literals of its kernel (mlgpu_graph_set_live_constants before compile)");
    g->nodes[node].value = value;
    if (!g->compiled) return MLGPU_OK;
    uint32_t u;
    memcpy(&u, &value, 4);
    return mlgpu_fill32(g->e, g->d_consts + g->nodes[node].slot, u, 1);
  }
  // Same nodes, same wiring? (what a second capture of the same user code produces when only host-side numbers changed)
  static const char* structureDifference(const mlgpu_graph* a, const mlgpu_graph* b)
  {
    if (a->V != b->V) return "number of voices";
    if (a->nodes.size() != b->nodes.size()) return "number of nodes";
    for (size_t i = 0; i < a->nodes.size(); ++i)
    {
      const Node &x = a->nodes[i], &y = b->nodes[i];
      if (x.type != y.type || x.kind != y.kind || x.in != y.in || x.slot != y.slot || x.nOut != y.nOut || x.fbSource != y.fbSource || x.region != y.region ||
          x.role != y.role || x.rate != y.rate)
        return "a node or its inputs";
      if (x.ringLen != y.ringLen) return "a delay line's maximum length";
      if (x.table != y.table) return "a constant DSPVector";
      if (x.type == NODE_CONST && !a->liveConsts && memcmp(&x.value, &y.value, 4) != 0) return "a constant (this graph was not compiled with live constants)";
    }
    if (a->outputs != b->outputs) return "outputs";
    if (a->regions.size() != b->regions.size()) return "rate regions";
    return nullptr;
  }
  int mlgpu_graph_update_constants_from(mlgpu_graph* g, mlgpu_graph* other)
  {
    if (!g || !other) return MLGPU_ERR_INVALID;
    if (g->job || other->job) return MLGPU_ERR_BUSY;
    if (!g->compiled) return gfail(g, MLGPU_ERR_INVALID, "graph_update_constants_from: compile the graph first");
    if (const char* why = structureDifference(g, other)) return gfail(g, MLGPU_ERR_UNSUPPORTED, std::string("graph_update_constants_from: the graphs differ in ") + why);
    for (size_t i = 0; i < g->nodes.size(); ++i)
    {
      Node& n = g->nodes[i];
      if (n.type != NODE_CONST || memcmp(&n.value, &other->nodes[i].value, 4) == 0) continue;
      const int st = mlgpu_graph_set_const(g, (int)i, other->nodes[i].value);
      if (st != MLGPU_OK) return st;
    }
    return MLGPU_OK;
  }
  size_t mlgpu_graph_device_bytes(mlgpu_graph* g)
  {
    if (!g || g->job || !g->compiled) return 0;
    return sizeof(float) * g->V * (size_t)(g->NC + 1) + sizeof(uint32_t) * g->V * (size_t)(g->NS + 1) + sizeof(float) * g->V * (size_t)(g->nParams + 1) +
           sizeof(float) * g->memVoices() * g->memFloatsPerVoice;
  }
  int mlgpu_graph_tuning(mlgpu_graph* g, int* voicesPerLane, int* quadsPerTrip)
  {
    if (!g) return -MLGPU_ERR_INVALID;
    if (g->job) return -MLGPU_ERR_BUSY;
    if (!g->compiled) return -MLGPU_ERR_INVALID;
    if (voicesPerLane) *voicesPerLane = g->activeVl;
    if (quadsPerTrip) *quadsPerTrip = g->unrollQ;
    return (g->autotune && !g->tuned) ? 0 : 1;
  }

  // how many workgroups of this graph's kernel a CU holds at once (registers, LDS and wavefront slots together): what decides whether a
  // bank runs in one round (voices <= 256 x that x the CU count) or the last workgroups run alone after the others
  int mlgpu_graph_workgroups_per_cu(mlgpu_graph* g)
  {
    if (!g) return -MLGPU_ERR_INVALID;
    if (g->job) return -MLGPU_ERR_BUSY;
    if (!g->compiled || !g->fn) return -MLGPU_ERR_INVALID;
    int n = 0;
    if (hipSetDevice(g->e->device) != hipSuccess) return -MLGPU_ERR_HIP;
    if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&n, g->fn, 256, 0) != hipSuccess) return -MLGPU_ERR_HIP;
    return n;
  }

  // the ring layout in effect (after compile: what layout 3 came out as)
  int mlgpu_graph_delay_layout(mlgpu_graph* g)
  {
    if (!g) return -MLGPU_ERR_INVALID;
    if (g->job) return -MLGPU_ERR_BUSY;  // (layout 3 is being decided)
    if (g->transposedIfPossible && !g->compiled) return 3;
    return g->transposedRings ? 2 : (g->sectorRings ? 4 : (g->windowedRings ? 1 : 0));
  }

  int mlgpu_graph_set_voices_per_lane(mlgpu_graph* g, int n)
  {
    if (!g) return MLGPU_ERR_INVALID;
    if (g->job) return MLGPU_ERR_BUSY;
    if (g->compiled) return gfail(g, MLGPU_ERR_INVALID, "graph already compiled");
    if (n < 0 || n > 2) return gfail(g, MLGPU_ERR_INVALID, "graph_set_voices_per_lane: 0 (automatic), 1 or 2");
    g->voicesPerLane = n;
    return MLGPU_OK;
  }

  int mlgpu_graph_set_input_layout(mlgpu_graph* g, int inputIndex, int layout)
  {
    if (!g) return MLGPU_ERR_INVALID;
    if (g->job) return MLGPU_ERR_BUSY;
    if (inputIndex < 0 || inputIndex >= g->nInputs) return gfail(g, MLGPU_ERR_RANGE, "graph_set_input_layout: input index out of range");
    if (layout < -1 || layout > MLGPU_LAYOUT_BROADCAST) return gfail(g, MLGPU_ERR_INVALID, "graph_set_input_layout: bad layout");
    g->inLayoutOverride[inputIndex] = layout;
    return MLGPU_OK;
  }

  int mlgpu_graph_set_input_group(mlgpu_graph* g, int inputIndex, int group)
  {
    if (!g) return MLGPU_ERR_INVALID;
    if (g->job) return MLGPU_ERR_BUSY;
    if (g->compiled) return gfail(g, MLGPU_ERR_INVALID, "graph already compiled");
    if (inputIndex < 0 || inputIndex >= g->nInputs) return gfail(g, MLGPU_ERR_RANGE, "graph_set_input_group: input index out of range");
    if (group < 1 || (size_t)group > g->V || g->V % (size_t)group) return gfail(g, MLGPU_ERR_INVALID, "graph_set_input_group: the voices are not a whole number of groups");
    g->inputGroup[inputIndex] = group;
    return MLGPU_OK;
  }

  int mlgpu_graph_set_state_uniform(mlgpu_graph* g, int node, int idx, uint32_t value)
  {
    int st = checkStateNode(g, node);
    if (st) return st;
    if (!g->compiled) return gfail(g, MLGPU_ERR_INVALID, "graph_set_state: compile first");
    if (idx < 0 || idx >= g->nodes[node].ns) return gfail(g, MLGPU_ERR_RANGE, "state index out of range");
    return mlgpu_fill32(g->e, g->d_state + (size_t)(g->nodes[node].sOff + idx) * g->V, value, g->V);
  }

  int mlgpu_graph_process(mlgpu_graph* g, size_t T, const float* const* d_inputs, int inLayout, float* const* d_outputs, int outLayout)
  {
    return mlgpu_graph_process_ctl(g, T, d_inputs, inLayout, nullptr, d_outputs, outLayout);
  }
  // a graph with event rows: T DSPVectors starting at frame startOffset of the bound object's event times (as mlgpu_events_process)
  int mlgpu_graph_process_events(mlgpu_graph* g, size_t T, int startOffset, const float* const* d_inputs, int inLayout, const float* const* d_controls,
                                 float* const* d_outputs, int outLayout)
  {
    if (!g) return MLGPU_ERR_INVALID;
    if (g->job) return MLGPU_ERR_BUSY;
    if (!g->hasEventRows) return gfail(g, MLGPU_ERR_INVALID, "graph_process_events: the graph has no event rows");
    if (startOffset < 0) return gfail(g, MLGPU_ERR_INVALID, "graph_process_events: negative frame offset");
    g->eventOffset = startOffset;
    const int st = mlgpu_graph_process_ctl(g, T, d_inputs, inLayout, d_controls, d_outputs, outLayout);
    g->eventOffset = -1;
    return st;
  }

  int mlgpu_graph_process_ctl(mlgpu_graph* g, size_t T, const float* const* d_inputs, int inLayout, const float* const* d_controls,
                              float* const* d_outputs, int outLayout)
  {
    if (!g) return MLGPU_ERR_INVALID;
    if (g->job) return MLGPU_ERR_BUSY;
    if (!g->compiled) return gfail(g, MLGPU_ERR_INVALID, "graph_process: compile first");
    if (T == 0) return MLGPU_OK;
    if (inLayout < 0 || inLayout > MLGPU_LAYOUT_BROADCAST || outLayout < 0 || outLayout > MLGPU_LAYOUT_VOICE_MAJOR)
      return gfail(g, MLGPU_ERR_INVALID, "graph_process: bad layout");
    if ((g->nInputs && !d_inputs) || (g->nControls && !d_controls) || !d_outputs) return gfail(g, MLGPU_ERR_INVALID, "graph_process: null signal list");
    GraphArgs a;
    memset(&a, 0, sizeof(a));
    a.coeffs = g->d_coeffs;
    a.state = g->d_state;
    a.params = g->d_params;
    a.consts = g->d_consts;
    a.mem = g->d_mem;
    a.V = g->V;
    a.T = T;
    a.t0 = g->vectorCount;
    a.flags = g->e->kflags;
    a.impulseTable = g->e->d_impulseTable;
    if (!g->waveClockPath.empty())
    {
      const size_t waves = (g->V + 255) / 256 * 4;
      if (!g->d_waveClock && hipMalloc((void**)&g->d_waveClock, waves * 4 * sizeof(unsigned long long)) != hipSuccess) g->d_waveClock = nullptr;
      a.waveClock = g->d_waveClock;
    }
    for (int i = 0; i < g->nInputs; ++i)
    {
      if (!d_inputs[i] || ((uintptr_t)d_inputs[i] & 15)) return gfail(g, MLGPU_ERR_INVALID, "graph_process: null / misaligned input");
      const int lay = (g->inLayoutOverride[i] >= 0) ? g->inLayoutOverride[i] : inLayout;
      a.in[i] = makeView(d_inputs[i], lay, g->inputGroup[i] > 1 ? g->V / (size_t)g->inputGroup[i] : g->V, T);
    }
    for (int i = 0; i < g->nControls; ++i)
    {
      if (!d_controls[i]) return gfail(g, MLGPU_ERR_INVALID, "graph_process: null control signal");
      a.ctl[i] = d_controls[i];
    }
    for (size_t o = 0; o < g->outputs.size(); ++o)
    {
      if (!d_outputs[o] || ((uintptr_t)d_outputs[o] & 15)) return gfail(g, MLGPU_ERR_INVALID, "graph_process: null / misaligned output");
      a.out[o] = makeView(d_outputs[o], outLayout, g->outputGroup[o] ? g->V / (size_t)g->outputGroup[o] : g->V, T);
    }
    // outputs that are mixdowns: the kernel writes the rows of 64-voice group sums into the engine's mixdown scratch (one region per
    // such output), the later stages follow the launch
    const size_t mixGroups = (g->V + 63) / 64, mixRegion = (mixGroups + (mixGroups + 63) / 64) * T * 64;
    size_t nMix = 0;
    for (size_t o = 0; o < g->outputs.size(); ++o)
      if (g->outputMix[o])
      {
        a.out[o] = makeView(g->e->d_mixScratch + nMix * mixRegion, MLGPU_LAYOUT_QUAD, g->V, T);
        ++nMix;
      }
    if (nMix * mixRegion > g->e->mixScratchFloats)
      return gfail(g, MLGPU_ERR_INVALID, "graph_process: call mlgpu_graph_reserve_mixdown(graph, max vectors) at setup (mlgpu_mixdown_reserve's scratch; process calls do not allocate)");
    if (g->e->recording)
    {
      if (g->autotune && !g->tuned) return gfail(g, MLGPU_ERR_INVALID, "graph_process: a graph that is still tuning cannot be recorded into a sequence");
      for (const Region& R : g->regions)
        if (R.kind == MLGPU_REGION_DOWNSAMPLE_2X)
          return gfail(g, MLGPU_ERR_INVALID, "graph_process: a graph with a DOWNSAMPLE_2X region counts DSPVectors and cannot be recorded into a sequence");
    }
    if (hipSetDevice(g->e->device) != hipSuccess) return gfail(g, MLGPU_ERR_HIP, "hipSetDevice");
    // online tuning: launches big enough to time take turns through the variants (3 runs each, the first one discarded)
    mlgpu_graph::Variant* trial = nullptr;
    if (g->autotune && !g->tuned && g->V * T * MLGPU_FLOATS_PER_DSPVECTOR >= ((size_t)1 << 22))
    {
      for (mlgpu_graph::Variant& v : g->variants)
        if (!v.failed && v.runs < 3 && (!trial || v.runs < trial->runs)) trial = &v;
      if (trial && !trial->fn)
      {
        const int keepUnroll = g->unrollQ, keepVl = g->compiledVoicesPerLane;
        g->unrollQ = trial->unroll;
        std::string src, log;
        std::vector<char> code;
        const bool built = generateBudgeted(g, trial->vl, src, code, log);
        g->unrollQ = keepUnroll;
        g->compiledVoicesPerLane = keepVl;
        CompiledModule* cm = built ? compileAndLoad(g->e->device, src, log) : nullptr;
        trial->fn = cm ? getFunction(cm, "mlgpu_graph_kernel", log) : nullptr;
        if (!trial->fn)
        {
          trial->failed = true;
          trial = nullptr;
        }
      }
      if (!trial)
      {
        // every variant has its runs: keep the fastest; the default form stays unless another one is at least 3 % faster
        // (two timed launches per form are a coarse measurement)
        const mlgpu_graph::Variant* best = nullptr;
        for (const mlgpu_graph::Variant& v : g->variants)
        {
          if (v.failed || !v.fn || v.runs < 2) continue;
          const float handicap = (v.fn == g->fn) ? 0.97f : 1.0f;
          if (!best || v.bestMs * handicap < best->bestMs * ((best->fn == g->fn) ? 0.97f : 1.0f)) best = &v;
        }
        if (best)
        {
          g->fn = best->fn;
          g->activeVl = best->vl;
          g->unrollQ = best->unroll;
        }
        g->tuned = true;
      }
    }
    const hipFunction_t fn = trial ? trial->fn : g->fn;
    const int vl = trial ? trial->vl : g->activeVl;
    if (trial && !g->tuneEv0 && (hipEventCreate(&g->tuneEv0) != hipSuccess || hipEventCreate(&g->tuneEv1) != hipSuccess))
      return gfail(g, MLGPU_ERR_HIP, "graph_process: hipEventCreate");
    // event rows: the host half of the EventsToSignals block (routing the block's events into records, their upload) goes first,
    // on the same stream; the kernel then walks the records itself
    void* eventStaging = nullptr;
    if (g->hasEventRows)
    {
      if (!g->events) return gfail(g, MLGPU_ERR_INVALID, "graph_process: the graph has event rows but no events object (graph_bind_events)");
      if (g->eventOffset < 0) return gfail(g, MLGPU_ERR_INVALID, "graph_process: a graph with event rows is run with mlgpu_graph_process_events");
      const int est = mlgpu_events_prepare_for_graph(g->events, T, g->eventOffset, &a.events, &eventStaging);
      g->eventOffset = -1;
      if (est != MLGPU_OK) return gfail(g, est, "graph_process: the events object refused the block (see its last error)");
    }
    if (trial) hipEventRecord(g->tuneEv0, g->e->stream);
    const hipError_t err = launchJit(fn, a, (g->V + (size_t)vl - 1) / (size_t)vl, g->e->stream);
    if (err != hipSuccess)
    {
      if (eventStaging) mlgpu_events_abandoned_by_graph(g->events, eventStaging);  // (the lanes' record ranges back to "none": no kernel will consume them)
      return gfail(g, MLGPU_ERR_HIP, std::string("graph_process launch: ") + hipGetErrorString(err));
    }
    if (eventStaging)
    {
      const int est = mlgpu_events_launched_by_graph(g->events, eventStaging);
      if (est != MLGPU_OK) return gfail(g, est, "graph_process: events bookkeeping after the launch");
    }
    for (size_t o = 0, r = 0; o < g->outputs.size(); ++o)
      if (g->outputMix[o])
      {
        const hipError_t merr = g->outputMixShard[o]
                                    ? mlgpu_launch_mixdown_rows_partial(mixGroups, T, g->e->d_mixScratch + r * mixRegion, d_outputs[o], mlgpu_mixdown_shard_level(g->V) - 1, g->e->stream, g->e->kflags)
                                    : mlgpu_launch_mixdown_rows(mixGroups, T, g->e->d_mixScratch + r * mixRegion, d_outputs[o], g->e->stream, g->e->kflags);
        if (merr != hipSuccess) return gfail(g, MLGPU_ERR_HIP, "graph_process: the mixdown's later stages");
        ++r;
      }
    if (trial)
    {
      hipEventRecord(g->tuneEv1, g->e->stream);
      float ms = 0.f;
      if (hipEventSynchronize(g->tuneEv1) == hipSuccess && hipEventElapsedTime(&ms, g->tuneEv0, g->tuneEv1) == hipSuccess)
      {
        if (trial->runs >= 1) trial->bestMs = std::min(trial->bestMs, ms / (float)T);
        trial->runs++;
      }
      else
        trial->failed = true;
    }
    g->vectorCount += T;
    return MLGPU_OK;
  }
}

// events.hip, mlgpu_events_destroy: no graph keeps a pointer to an events object that is gone
void mlgpu_graph_forget_events(mlgpu_events* ev)
{
  std::lock_guard<std::mutex> lock(g_boundMutex);
  for (auto it = g_boundGraphs.begin(); it != g_boundGraphs.end();)
  {
    if ((*it)->events == ev)
    {
      (*it)->events = nullptr;
      it = g_boundGraphs.erase(it);
    }
    else
      ++it;
  }
}
