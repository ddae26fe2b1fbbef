// capi.hip — the extern "C" boundary declared in include/mlgpu.h.
//
// Engine = one HIP device + one stream. Bank = V voices of one processor chain with SoA
// coefficient/state arrays in HBM. Nothing here computes signal values on the host: every
// compute entry either launches a gfx950 kernel or fails (no CPU fallback by design).
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <new>

#include "mlgpu_internal.hpp"

extern const char mlgpu_device_source_hash_str[];  // embedded_sources.cpp (embed.py)

namespace
{
int fail(mlgpu_engine* e, int status, const char* what, hipError_t herr = hipSuccess)
{
  if (e)
  {
    char buf[512];
    if (herr != hipSuccess)
      snprintf(buf, sizeof(buf), "%s: %s (%s)", what, hipGetErrorString(herr), hipGetErrorName(herr));
    else
      snprintf(buf, sizeof(buf), "%s", what);
    e->lastError = buf;
  }
  return status;
}

#define HIP_TRY(e, call)                                            \
  do                                                                \
  {                                                                 \
    hipError_t _err = (call);                                       \
    if (_err != hipSuccess) return fail((e), MLGPU_ERR_HIP, #call, _err); \
  } while (0)

int createEngine(int device, hipStream_t stream, bool ownStream, mlgpu_engine** out, int urgency = 0)
{
  if (!out) return MLGPU_ERR_INVALID;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return MLGPU_ERR_NO_DEVICE;
  if (device < 0 || device >= count) return MLGPU_ERR_NO_DEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return MLGPU_ERR_NO_DEVICE;
  // the code object in this library is gfx950 only
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return MLGPU_ERR_NO_DEVICE;
  if (hipSetDevice(device) != hipSuccess) return MLGPU_ERR_HIP;
  mlgpu_engine* e = new (std::nothrow) mlgpu_engine();
  if (!e) return MLGPU_ERR_OOM;
  e->device = device;
  e->cuCount = prop.multiProcessorCount;
  if (ownStream)
  {
    // urgency: the dispatcher hands out a more urgent stream's workgroups first, so a short memory-bound kernel on it is not
    // starved by a long kernel that fills every CU from a normal stream (HIP: the numerically LOWER priority is the greater one)
    int least = 0, greatest = 0;
    if (urgency != 0 && hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = greatest = 0;
    const int prio = urgency > 0 ? greatest : (urgency < 0 ? least : 0);
    if ((urgency == 0 ? hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking)
                      : hipStreamCreateWithPriority(&e->stream, hipStreamNonBlocking, prio)) != hipSuccess)
    {
      delete e;
      return MLGPU_ERR_HIP;
    }
    e->ownsStream = true;
  }
  else
  {
    e->stream = stream;
  }
  float table[17];
  mlgpu_build_impulse_table(table);
  if (hipMalloc((void**)&e->d_validate, 2 * sizeof(unsigned long long)) != hipSuccess ||
      hipMalloc((void**)&e->d_impulseTable, sizeof(table)) != hipSuccess ||
      hipMemcpy(e->d_impulseTable, table, sizeof(table), hipMemcpyHostToDevice) != hipSuccess)
  {
    if (e->ownsStream) hipStreamDestroy(e->stream);
    delete e;
    return MLGPU_ERR_HIP;
  }
  *out = e;
  return MLGPU_OK;
}
}  // namespace

struct mlgpu_bank
{
  mlgpu_engine* e{nullptr};
  std::vector<int32_t> kinds;
  std::vector<int> cOff, sOff, nc, ns;
  int NC{0}, NS{0};
  size_t V{0};
  float* d_coeffs{nullptr};
  uint32_t* d_state{nullptr};
  float* d_inConst{nullptr};
  const ChainEntry* fused{nullptr};   // ahead-of-time fused kernel, if the chain is in the catalogue
  void* jitSignal{nullptr};           // else: fused kernels generated with hiprtc at bank creation
  void* jitConst{nullptr};
  std::string jitName;
  std::vector<const ChainEntry*> singles;
  float* d_scratch[2]{nullptr, nullptr};
  size_t scratchVectors{0};
};

extern "C"
{
  int mlgpu_abi_version(void) { return MLGPU_ABI_VERSION; }
  const char* mlgpu_device_source_hash(void) { return mlgpu_device_source_hash_str; }

  int mlgpu_behaviour_revision(void) { return 3; }
  const char* mlgpu_status_string(int s)
  {
    switch (s)
    {
      case MLGPU_OK: return "ok";
      case MLGPU_ERR_INVALID: return "invalid argument";
      case MLGPU_ERR_NO_DEVICE: return "no gfx950 (MI355X) HIP device";
      case MLGPU_ERR_HIP: return "HIP runtime error";
      case MLGPU_ERR_OOM: return "out of memory";
      case MLGPU_ERR_UNSUPPORTED: return "unsupported";
      case MLGPU_ERR_RANGE: return "index out of range";
      case MLGPU_ERR_BUSY: return "busy: a job started on this object has not finished";
      default: return "unknown status";
    }
  }

  int mlgpu_device_count(void)
  {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
  }

  int mlgpu_device_info(int device, char* name, size_t nameLen, int* cuCount, uint64_t* memBytes)
  {
    hipDeviceProp_t prop;
    if (device < 0 || device >= mlgpu_device_count()) return MLGPU_ERR_NO_DEVICE;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return MLGPU_ERR_HIP;
    if (name && nameLen)
    {
      snprintf(name, nameLen, "%s (%s)", prop.name, prop.gcnArchName);
    }
    if (cuCount) *cuCount = prop.multiProcessorCount;
    if (memBytes) *memBytes = (uint64_t)prop.totalGlobalMem;
    return MLGPU_OK;
  }

  int mlgpu_device_pci_bus_id(int device, char* buf, size_t bufLen)
  {
    if (!buf || bufLen < 13) return MLGPU_ERR_INVALID;
    if (device < 0 || device >= mlgpu_device_count()) return MLGPU_ERR_NO_DEVICE;
    return hipDeviceGetPCIBusId(buf, (int)bufLen, device) == hipSuccess ? MLGPU_OK : MLGPU_ERR_HIP;
  }

  int mlgpu_device_synchronize(int device)
  {
    if (device < 0 || device >= mlgpu_device_count()) return MLGPU_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return MLGPU_ERR_HIP;
    return hipDeviceSynchronize() == hipSuccess ? MLGPU_OK : MLGPU_ERR_HIP;
  }

  int mlgpu_engine_set_flush_denormals(mlgpu_engine* e, int on)
  {
    if (!e) return MLGPU_ERR_INVALID;
    if (e->recording) return fail(e, MLGPU_ERR_INVALID, "mlgpu_engine_set_flush_denormals: not while recording a launch sequence");
    e->kflags = on ? (e->kflags | MLGPU_KFLAG_FLUSH_DENORMALS) : (e->kflags & ~MLGPU_KFLAG_FLUSH_DENORMALS);
    return MLGPU_OK;
  }
  int mlgpu_engine_get_flush_denormals(mlgpu_engine* e) { return (e && (e->kflags & MLGPU_KFLAG_FLUSH_DENORMALS)) ? 1 : 0; }

  int mlgpu_engine_set_strict_svf(mlgpu_engine* e, int on)
  {
    if (!e) return MLGPU_ERR_INVALID;
    e->strictSvf = on != 0;
    return MLGPU_OK;
  }
  int mlgpu_engine_get_strict_svf(mlgpu_engine* e) { return (e && e->strictSvf) ? 1 : 0; }

  int mlgpu_engine_set_cascade_lanes(mlgpu_engine* e, int lanes)
  {
    if (!e) return MLGPU_ERR_INVALID;
    if (e->recording) return fail(e, MLGPU_ERR_INVALID, "mlgpu_engine_set_cascade_lanes: not while recording a launch sequence");
    if (lanes != 0 && lanes != 1 && lanes != 2 && lanes != 4 && lanes != -1) return fail(e, MLGPU_ERR_INVALID, "mlgpu_engine_set_cascade_lanes: 0 (by size), 1, 2, 4 or -1 (the one-lane round-2 kernel)");
    const uint32_t code = lanes < 0 ? 7u : (uint32_t)lanes;
    e->kflags = (e->kflags & ~MLGPU_KFLAG_CASCADE_MASK) | (code << MLGPU_KFLAG_CASCADE_SHIFT);
    return MLGPU_OK;
  }
  int mlgpu_engine_get_cascade_lanes(mlgpu_engine* e)
  {
    if (!e) return 0;
    const uint32_t code = (e->kflags & MLGPU_KFLAG_CASCADE_MASK) >> MLGPU_KFLAG_CASCADE_SHIFT;
    return code == 7 ? -1 : (int)code;
  }

  int mlgpu_engine_create(int device, mlgpu_engine** out) { return createEngine(device, nullptr, true, out); }
  int mlgpu_engine_create_urgency(int device, int urgency, mlgpu_engine** out)
  {
    if (urgency < -1 || urgency > 1) return MLGPU_ERR_INVALID;
    return createEngine(device, nullptr, true, out, urgency);
  }
  int mlgpu_engine_create_on_stream(int device, void* hipStream, mlgpu_engine** out)
  {
    return createEngine(device, (hipStream_t)hipStream, false, out);
  }

  int mlgpu_engine_destroy(mlgpu_engine* e)
  {
    if (!e) return MLGPU_ERR_INVALID;
    hipSetDevice(e->device);
    hipStreamSynchronize(e->stream);
    e->runDeferredFrees();
    if (e->ev0) hipEventDestroy(e->ev0);
    if (e->ev1) hipEventDestroy(e->ev1);
    if (e->d_impulseTable) hipFree(e->d_impulseTable);
    if (e->d_validate) hipFree(e->d_validate);
    if (e->d_mixScratch) hipFree(e->d_mixScratch);
    if (e->ownsStream) hipStreamDestroy(e->stream);
    delete e;
    return MLGPU_OK;
  }

  int mlgpu_engine_sync(mlgpu_engine* e)
  {
    if (!e) return MLGPU_ERR_INVALID;
    if (e->recording) return fail(e, MLGPU_ERR_INVALID, "engine_sync: not while recording a sequence");
    HIP_TRY(e, hipSetDevice(e->device));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    return MLGPU_OK;
  }
  void* mlgpu_engine_stream(mlgpu_engine* e) { return e ? (void*)e->stream : nullptr; }

  // ---- recorded launch sequences (hipGraph) ----
  int mlgpu_engine_begin_recording(mlgpu_engine* e)
  {
    if (!e) return MLGPU_ERR_INVALID;
    if (e->recording) return fail(e, MLGPU_ERR_INVALID, "begin_recording: already recording");
    HIP_TRY(e, hipSetDevice(e->device));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    HIP_TRY(e, hipStreamBeginCapture(e->stream, hipStreamCaptureModeRelaxed));
    e->recording = true;
    return MLGPU_OK;
  }
  int mlgpu_engine_end_recording(mlgpu_engine* e, mlgpu_sequence** out)
  {
    if (!e || !out) return MLGPU_ERR_INVALID;
    *out = nullptr;
    if (!e->recording) return fail(e, MLGPU_ERR_INVALID, "end_recording: not recording");
    hipSetDevice(e->device);
    e->recording = false;
    hipGraph_t graph = nullptr;
    HIP_TRY(e, hipStreamEndCapture(e->stream, &graph));
    if (!graph) return fail(e, MLGPU_ERR_HIP, "end_recording: the capture was invalidated (a call that waits for the device ran while recording)");
    mlgpu_sequence* s = new (std::nothrow) mlgpu_sequence();
    if (s) ++e->liveSequences;
    if (!s)
    {
      hipGraphDestroy(graph);
      return MLGPU_ERR_OOM;
    }
    s->e = e;
    s->graph = graph;
    const hipError_t err = hipGraphInstantiate(&s->exec, graph, nullptr, nullptr, 0);
    if (err != hipSuccess)
    {
      hipGraphDestroy(graph);
      delete s;
      // the sequence never came to be: without this the engine kept counting it, and the objects destroyed from then on (their
      // frees are deferred while a sequence lives) waited for the engine's own destruction
      if (--e->liveSequences == 0) e->runDeferredFrees();
      return fail(e, MLGPU_ERR_HIP, "end_recording: hipGraphInstantiate", err);
    }
    size_t n = 0;
    hipGraphGetNodes(graph, nullptr, &n);
    s->nodes = n;
    *out = s;
    return MLGPU_OK;
  }
  int mlgpu_sequence_launch(mlgpu_sequence* s)
  {
    if (!s) return MLGPU_ERR_INVALID;
    if (s->e->recording) return fail(s->e, MLGPU_ERR_INVALID, "sequence_launch: the engine is recording");
    HIP_TRY(s->e, hipSetDevice(s->e->device));
    HIP_TRY(s->e, hipGraphLaunch(s->exec, s->e->stream));
    return MLGPU_OK;
  }
  size_t mlgpu_sequence_num_nodes(mlgpu_sequence* s) { return s ? s->nodes : 0; }
  int mlgpu_sequence_destroy(mlgpu_sequence* s)
  {
    if (!s) return MLGPU_ERR_INVALID;
    // waiting for the stream would invalidate a capture in progress (end_recording then fails with a generic message)
    if (s->e->recording) return fail(s->e, MLGPU_ERR_INVALID, "sequence_destroy waits for the device: not while recording a sequence");
    hipSetDevice(s->e->device);
    hipStreamSynchronize(s->e->stream);
    if (s->exec) hipGraphExecDestroy(s->exec);
    if (s->graph) hipGraphDestroy(s->graph);
    if (s->e->liveSequences > 0) --s->e->liveSequences;
    if (s->e->liveSequences == 0) s->e->runDeferredFrees();
    delete s;
    return MLGPU_OK;
  }
  int mlgpu_engine_set_jit(mlgpu_engine* e, int enabled)
  {
    if (!e) return MLGPU_ERR_INVALID;
    e->jitEnabled = enabled != 0;
    return MLGPU_OK;
  }
  int mlgpu_engine_device(mlgpu_engine* e) { return e ? e->device : -1; }
  const char* mlgpu_last_error(mlgpu_engine* e) { return e ? e->lastError.c_str() : "null engine"; }

  int mlgpu_alloc(mlgpu_engine* e, size_t bytes, void** d_out)
  {
    if (!e || !d_out) return MLGPU_ERR_INVALID;
    *d_out = nullptr;
    if (bytes == 0) return MLGPU_OK;
    HIP_TRY(e, hipSetDevice(e->device));
    hipError_t err = hipMalloc(d_out, bytes);
    if (err == hipErrorOutOfMemory) return fail(e, MLGPU_ERR_OOM, "hipMalloc", err);
    if (err != hipSuccess) return fail(e, MLGPU_ERR_HIP, "hipMalloc", err);
    return MLGPU_OK;
  }
  int mlgpu_free(mlgpu_engine* e, void* d_ptr)
  {
    if (!e) return MLGPU_ERR_INVALID;
    if (!d_ptr) return MLGPU_OK;
    if (e->recording) return fail(e, MLGPU_ERR_INVALID, "free waits for the device: not while recording a sequence");
    HIP_TRY(e, hipSetDevice(e->device));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    HIP_TRY(e, hipFree(d_ptr));
    return MLGPU_OK;
  }
  int mlgpu_upload(mlgpu_engine* e, void* d_dst, const void* h_src, size_t bytes)
  {
    if (!e || (bytes && (!d_dst || !h_src))) return MLGPU_ERR_INVALID;
    if (e->recording) return fail(e, MLGPU_ERR_INVALID, "upload waits for the device: not while recording a sequence");
    if (!bytes) return MLGPU_OK;
    HIP_TRY(e, hipSetDevice(e->device));
    HIP_TRY(e, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));  // h_src may be pageable and reused by the caller
    return MLGPU_OK;
  }
  int mlgpu_download(mlgpu_engine* e, void* h_dst, const void* d_src, size_t bytes)
  {
    if (!e || (bytes && (!h_dst || !d_src))) return MLGPU_ERR_INVALID;
    if (e->recording) return fail(e, MLGPU_ERR_INVALID, "download waits for the device: not while recording a sequence");
    if (!bytes) return MLGPU_OK;
    HIP_TRY(e, hipSetDevice(e->device));
    HIP_TRY(e, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    return MLGPU_OK;
  }
  int mlgpu_fill32(mlgpu_engine* e, void* d_dst, uint32_t value, size_t n)
  {
    if (!e || (n && !d_dst)) return MLGPU_ERR_INVALID;
    if (!n) return MLGPU_OK;
    HIP_TRY(e, hipSetDevice(e->device));
    HIP_TRY(e, mlgpu_launch_fill32((uint32_t*)d_dst, value, n, e->stream));
    return MLGPU_OK;
  }

  // ---- fences: ordering between two engines (= two HIP streams) of one device ----
  int mlgpu_fence_create(mlgpu_engine* e, mlgpu_fence** out)
  {
    if (!e || !out) return MLGPU_ERR_INVALID;
    *out = nullptr;
    HIP_TRY(e, hipSetDevice(e->device));
    mlgpu_fence* f = new (std::nothrow) mlgpu_fence();
    if (!f) return MLGPU_ERR_OOM;
    f->device = e->device;
    if (hipEventCreateWithFlags(&f->ev, hipEventDisableTiming) != hipSuccess)
    {
      delete f;
      return fail(e, MLGPU_ERR_HIP, "fence_create: hipEventCreate");
    }
    *out = f;
    return MLGPU_OK;
  }
  int mlgpu_fence_destroy(mlgpu_fence* f)
  {
    if (!f) return MLGPU_ERR_INVALID;
    hipSetDevice(f->device);
    hipEventDestroy(f->ev);
    delete f;
    return MLGPU_OK;
  }
  int mlgpu_engine_signal(mlgpu_engine* e, mlgpu_fence* f)
  {
    if (!e || !f) return MLGPU_ERR_INVALID;
    if (f->device != e->device) return fail(e, MLGPU_ERR_INVALID, "engine_signal: the fence belongs to another device");
    if (e->recording) return fail(e, MLGPU_ERR_INVALID, "engine_signal: not while recording a sequence (a fence ties two streams together)");
    HIP_TRY(e, hipSetDevice(e->device));
    HIP_TRY(e, hipEventRecord(f->ev, e->stream));
    f->signalled = true;
    return MLGPU_OK;
  }
  int mlgpu_engine_wait(mlgpu_engine* e, mlgpu_fence* f)
  {
    if (!e || !f) return MLGPU_ERR_INVALID;
    if (f->device != e->device) return fail(e, MLGPU_ERR_INVALID, "engine_wait: the fence belongs to another device");
    if (e->recording) return fail(e, MLGPU_ERR_INVALID, "engine_wait: not while recording a sequence");
    if (!f->signalled) return MLGPU_OK;  // nothing to wait for yet (the first trip round a ring of buffers)
    HIP_TRY(e, hipSetDevice(e->device));
    HIP_TRY(e, hipStreamWaitEvent(e->stream, f->ev, 0));
    return MLGPU_OK;
  }

  int mlgpu_timer_start(mlgpu_engine* e)
  {
    if (!e) return MLGPU_ERR_INVALID;
    HIP_TRY(e, hipSetDevice(e->device));
    if (!e->ev0)
    {
      HIP_TRY(e, hipEventCreate(&e->ev0));
      HIP_TRY(e, hipEventCreate(&e->ev1));
    }
    HIP_TRY(e, hipEventRecord(e->ev0, e->stream));
    return MLGPU_OK;
  }
  int mlgpu_timer_stop_ms(mlgpu_engine* e, float* msOut)
  {
    if (!e || !msOut || !e->ev0) return MLGPU_ERR_INVALID;
    if (e->recording) return fail(e, MLGPU_ERR_INVALID, "timer_stop waits for the device: not while recording a sequence");
    HIP_TRY(e, hipSetDevice(e->device));
    HIP_TRY(e, hipEventRecord(e->ev1, e->stream));
    HIP_TRY(e, hipEventSynchronize(e->ev1));
    HIP_TRY(e, hipEventElapsedTime(msOut, e->ev0, e->ev1));
    return MLGPU_OK;
  }

  int mlgpu_validate(mlgpu_engine* e, const float* d_signal, size_t n, uint64_t* count, uint64_t* firstIndex)
  {
    if (!e || !count) return MLGPU_ERR_INVALID;
    *count = 0;
    if (firstIndex) *firstIndex = ~(uint64_t)0;
    if (n == 0) return MLGPU_OK;
    if (!d_signal || ((uintptr_t)d_signal & 15)) return fail(e, MLGPU_ERR_INVALID, "validate: null / misaligned signal");
    if (e->recording) return fail(e, MLGPU_ERR_INVALID, "validate waits for the device: not while recording a sequence");
    HIP_TRY(e, hipSetDevice(e->device));
    HIP_TRY(e, mlgpu_launch_validate(d_signal, n, e->d_validate, e->stream, e->cuCount));
    unsigned long long r[2];
    HIP_TRY(e, hipMemcpyAsync(r, e->d_validate, sizeof(r), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    *count = r[0];
    if (firstIndex) *firstIndex = r[1];
    return MLGPU_OK;
  }

  // ---- stateless ops ----------------------------------------------------------------------

  int mlgpu_op_apply(mlgpu_engine* e, int op, const void* a, const void* b, const void* c, void* out, size_t n)
  {
    if (!e) return MLGPU_ERR_INVALID;
    const int arity = op >= 64 ? 3 : (op >= 32 ? 2 : 1);
    if (n == 0) return MLGPU_OK;
    if (!a || !out || (arity >= 2 && !b) || (arity >= 3 && !c)) return fail(e, MLGPU_ERR_INVALID, "op_apply: null operand");
    if ((arity < 2 && b) || (arity < 3 && c)) return fail(e, MLGPU_ERR_INVALID, "op_apply: unused operand must be NULL");
    if (((uintptr_t)a | (uintptr_t)out | (uintptr_t)b | (uintptr_t)c) & 15)
      return fail(e, MLGPU_ERR_INVALID, "op_apply: operands must be 16-byte aligned");
    HIP_TRY(e, hipSetDevice(e->device));
    bool known = false;
    hipError_t err = mlgpu_launch_op(op, a, b, c, out, n, e->stream, e->cuCount, &known, e->kflags);
    if (!known) return fail(e, MLGPU_ERR_INVALID, "op_apply: unknown op");
    if (err != hipSuccess) return fail(e, MLGPU_ERR_HIP, "op_apply launch", err);
    return MLGPU_OK;
  }

  int mlgpu_op_apply_rows1(mlgpu_engine* e, int op, const void* a, const void* b64, void* out, size_t nRows)
  {
    if (!e) return MLGPU_ERR_INVALID;
    if (nRows == 0) return MLGPU_OK;
    if (!a || !b64 || !out) return fail(e, MLGPU_ERR_INVALID, "op_apply_rows1: null operand");
    HIP_TRY(e, hipSetDevice(e->device));
    bool known = false;
    hipError_t err = mlgpu_launch_op_rows1(op, a, b64, out, nRows, e->stream, e->cuCount, &known, e->kflags);
    if (!known) return fail(e, MLGPU_ERR_INVALID, "op_apply_rows1: op must be ADD..MAX");
    if (err != hipSuccess) return fail(e, MLGPU_ERR_HIP, "op_apply_rows1 launch", err);
    return MLGPU_OK;
  }

  int mlgpu_row_reduce(mlgpu_engine* e, int rowop, const float* rows, float* out, size_t nRows)
  {
    if (!e) return MLGPU_ERR_INVALID;
    if (nRows == 0) return MLGPU_OK;
    if (!rows || !out) return fail(e, MLGPU_ERR_INVALID, "row_reduce: null operand");
    HIP_TRY(e, hipSetDevice(e->device));
    bool known = false;
    hipError_t err = mlgpu_launch_row_reduce(rowop, rows, out, nRows, e->stream, &known, e->kflags);
    if (!known) return fail(e, MLGPU_ERR_INVALID, "row_reduce: unknown rowop");
    if (err != hipSuccess) return fail(e, MLGPU_ERR_HIP, "row_reduce launch", err);
    return MLGPU_OK;
  }

  int mlgpu_layout_convert(mlgpu_engine* e, const float* src, int srcLayout, float* dst, int dstLayout, size_t V,
                           size_t T)
  {
    if (!e) return MLGPU_ERR_INVALID;
    if (V == 0 || T == 0) return MLGPU_OK;
    if (!src || !dst || src == dst) return fail(e, MLGPU_ERR_INVALID, "layout_convert: bad pointers (must not alias)");
    if (srcLayout < 0 || srcLayout > 2 || dstLayout < 0 || dstLayout > 2)
      return fail(e, MLGPU_ERR_INVALID, "layout_convert: bad layout");
    HIP_TRY(e, hipSetDevice(e->device));
    HIP_TRY(e, mlgpu_launch_layout_convert(src, srcLayout, dst, dstLayout, V, T, e->stream));
    return MLGPU_OK;
  }

  // ---- row plumbing and routing ---------------------------------------------------------------

  int mlgpu_rows_map(mlgpu_engine* e, int rule, long p0, long p1, int sampleRotate, const float* src, size_t srcRows, float* dst,
                     size_t dstRows, size_t dstOffset, size_t dstStep, size_t count, size_t groups)
  {
    if (!e) return MLGPU_ERR_INVALID;
    if (count == 0 || groups == 0) return MLGPU_OK;
    if (!src || !dst || src == dst) return fail(e, MLGPU_ERR_INVALID, "rows_map: bad pointers (must not alias)");
    if (rule < MLGPU_ROWS_REPEAT || rule > MLGPU_ROWS_STRIDED) return fail(e, MLGPU_ERR_INVALID, "rows_map: unknown rule");
    if (srcRows == 0 || sampleRotate < -1 || sampleRotate > 1) return fail(e, MLGPU_ERR_INVALID, "rows_map: bad src_rows / sample_rotate");
    if (dstStep == 0 || dstOffset + (count - 1) * dstStep >= dstRows) return fail(e, MLGPU_ERR_RANGE, "rows_map: destination rows out of range");
    HIP_TRY(e, hipSetDevice(e->device));
    HIP_TRY(e, mlgpu_launch_rows_map(rule, p0, p1, sampleRotate, src, srcRows, dst, dstRows, dstOffset, dstStep, count, groups, e->stream));
    return MLGPU_OK;
  }

  int mlgpu_rows_add(mlgpu_engine* e, const float* rows, size_t rowsPerGroup, float* out, size_t groups)
  {
    if (!e) return MLGPU_ERR_INVALID;
    if (groups == 0) return MLGPU_OK;
    if (!rows || !out) return fail(e, MLGPU_ERR_INVALID, "rows_add: null operand");
    HIP_TRY(e, hipSetDevice(e->device));
    HIP_TRY(e, mlgpu_launch_rows_add(rows, rowsPerGroup, out, groups, e->stream, e->kflags));
    return MLGPU_OK;
  }

  int mlgpu_rows_normalize(mlgpu_engine* e, const float* rows, float* out, size_t nRows)
  {
    if (!e) return MLGPU_ERR_INVALID;
    if (nRows == 0) return MLGPU_OK;
    if (!rows || !out) return fail(e, MLGPU_ERR_INVALID, "rows_normalize: null operand");
    HIP_TRY(e, hipSetDevice(e->device));
    HIP_TRY(e, mlgpu_launch_rows_normalize(rows, out, nRows, e->stream, e->kflags));
    return MLGPU_OK;
  }

  int mlgpu_rows_index(mlgpu_engine* e, float* out, size_t rowsPerGroup, size_t groups)
  {
    if (!e) return MLGPU_ERR_INVALID;
    if (groups == 0 || rowsPerGroup == 0) return MLGPU_OK;
    if (!out) return fail(e, MLGPU_ERR_INVALID, "rows_index: null operand");
    HIP_TRY(e, hipSetDevice(e->device));
    HIP_TRY(e, mlgpu_launch_rows_index(out, rowsPerGroup, groups, e->stream));
    return MLGPU_OK;
  }

  int mlgpu_multiplex(mlgpu_engine* e, const float* sel, size_t selElems, const float* const* ins, int n, float* out, size_t nElems, int linear)
  {
    if (!e) return MLGPU_ERR_INVALID;
    if (nElems == 0) return MLGPU_OK;
    if (!sel || !ins || !out || selElems == 0) return fail(e, MLGPU_ERR_INVALID, "multiplex: null operand");
    if (n < 1 || n > MLGPU_ROUTE_MAX_SIGNALS) return fail(e, MLGPU_ERR_UNSUPPORTED, "multiplex: 1..8 inputs");
    for (int k = 0; k < n; ++k)
      if (!ins[k]) return fail(e, MLGPU_ERR_INVALID, "multiplex: null input");
    HIP_TRY(e, hipSetDevice(e->device));
    float* outs[1] = {out};
    HIP_TRY(e, mlgpu_launch_route(false, linear != 0, sel, selElems, ins, outs, n, nElems, e->stream, e->kflags));
    return MLGPU_OK;
  }

  int mlgpu_demultiplex(mlgpu_engine* e, const float* sel, size_t selElems, const float* in, float* const* outs, int n, size_t nElems, int linear)
  {
    if (!e) return MLGPU_ERR_INVALID;
    if (nElems == 0) return MLGPU_OK;
    if (!sel || !in || !outs || selElems == 0) return fail(e, MLGPU_ERR_INVALID, "demultiplex: null operand");
    if (n < 1 || n > MLGPU_ROUTE_MAX_SIGNALS) return fail(e, MLGPU_ERR_UNSUPPORTED, "demultiplex: 1..8 outputs");
    for (int k = 0; k < n; ++k)
      if (!outs[k]) return fail(e, MLGPU_ERR_INVALID, "demultiplex: null output");
    HIP_TRY(e, hipSetDevice(e->device));
    const float* ins[1] = {in};
    HIP_TRY(e, mlgpu_launch_route(true, linear != 0, sel, selElems, ins, outs, n, nElems, e->stream, e->kflags));
    return MLGPU_OK;
  }

  int mlgpu_mixdown_reserve(mlgpu_engine* e, size_t maxVoices, size_t maxVectors)
  {
    if (!e) return MLGPU_ERR_INVALID;
    if (e->recording) return fail(e, MLGPU_ERR_INVALID, "mixdown_reserve allocates: not while recording a sequence");
    const size_t groups = (maxVoices + 63) / 64;
    const size_t need = (groups + (groups + 63) / 64) * maxVectors * 64;  // the group sums, and their sums 64 at a time (later passes fit the first part again)
    if (need <= e->mixScratchFloats) return MLGPU_OK;
    HIP_TRY(e, hipSetDevice(e->device));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    // a recorded mlgpu_mixdown has the scratch pointer baked in: while a sequence of this engine lives the old buffer stays where
    // it is (freed with the last sequence, like an events object destroyed under a sequence) and the new one serves from now on
    if (e->d_mixScratch)
    {
      float* old = e->d_mixScratch;
      if (e->liveSequences > 0) e->deferredFrees.push_back([old]() { hipFree(old); });
      else hipFree(old);
    }
    e->d_mixScratch = nullptr;
    e->mixScratchFloats = 0;
    const hipError_t err = hipMalloc((void**)&e->d_mixScratch, sizeof(float) * need);
    if (err != hipSuccess) return fail(e, err == hipErrorOutOfMemory ? MLGPU_ERR_OOM : MLGPU_ERR_HIP, "mixdown_reserve: scratch allocation", err);
    e->mixScratchFloats = need;
    return MLGPU_OK;
  }

  int mlgpu_mixdown(mlgpu_engine* e, const float* sig, int layout, size_t V, size_t T, const float* gains, float* out)
  {
    if (!e) return MLGPU_ERR_INVALID;
    if (V == 0 || T == 0) return MLGPU_OK;
    if (!sig || !out || ((uintptr_t)sig & 15) || ((uintptr_t)out & 15)) return fail(e, MLGPU_ERR_INVALID, "mixdown: null / misaligned signal");
    if (layout < 0 || layout > MLGPU_LAYOUT_VOICE_MAJOR) return fail(e, MLGPU_ERR_INVALID, "mixdown: bad layout");
    // a process call never allocates (mlgpu.h: "no allocation inside process"): the partial sums live in scratch the host
    // reserved at setup
    if (((V + 63) / 64 + ((V + 63) / 64 + 63) / 64) * T * 64 > e->mixScratchFloats)
      return fail(e, MLGPU_ERR_INVALID, "mixdown: call mlgpu_mixdown_reserve(engine, max voices, max vectors) at setup (process calls do not allocate)");
    HIP_TRY(e, hipSetDevice(e->device));
    HIP_TRY(e, mlgpu_launch_mixdown(sig, layout, V, T, gains, e->d_mixScratch, out, e->stream, e->kflags));
    return MLGPU_OK;
  }

  int mlgpu_mixdown_groups(mlgpu_engine* e, const float* sig, int layout, size_t groups, size_t groupSize, size_t T, float* out, int outLayout)
  {
    if (!e) return MLGPU_ERR_INVALID;
    if (groups == 0 || groupSize == 0 || T == 0) return MLGPU_OK;
    if (!sig || !out || ((uintptr_t)sig & 15) || ((uintptr_t)out & 15)) return fail(e, MLGPU_ERR_INVALID, "mixdown_groups: null / misaligned signal");
    if (layout < 0 || layout > MLGPU_LAYOUT_VOICE_MAJOR || outLayout < 0 || outLayout > MLGPU_LAYOUT_VOICE_MAJOR)
      return fail(e, MLGPU_ERR_INVALID, "mixdown_groups: bad layout");
    HIP_TRY(e, hipSetDevice(e->device));
    HIP_TRY(e, mlgpu_launch_mixdown_groups(sig, layout, groups, groupSize, T, out, outLayout, e->stream, e->kflags));
    return MLGPU_OK;
  }

  // ---- banks --------------------------------------------------------------------------------

  int mlgpu_bank_destroy(mlgpu_bank* b)
  {
    if (!b) return MLGPU_ERR_INVALID;
    mlgpu_engine* e = b->e;
    if (e->recording) return fail(e, MLGPU_ERR_INVALID, "bank_destroy waits for the device: not while recording a sequence");
    hipSetDevice(e->device);
    hipStreamSynchronize(e->stream);
    if (b->d_coeffs) hipFree(b->d_coeffs);
    if (b->d_state) hipFree(b->d_state);
    if (b->d_inConst) hipFree(b->d_inConst);
    if (b->d_scratch[0]) hipFree(b->d_scratch[0]);
    if (b->d_scratch[1]) hipFree(b->d_scratch[1]);
    delete b;
    return MLGPU_OK;
  }

  int mlgpu_bank_create(mlgpu_engine* e, const int32_t* procs, int nProcs, size_t nVoices, mlgpu_bank** out)
  {
    if (!e || !out) return MLGPU_ERR_INVALID;
    *out = nullptr;
    if (!procs || nProcs < 1 || nProcs > 64 || nVoices == 0) return fail(e, MLGPU_ERR_INVALID, "bank_create: bad arguments");
    mlgpu_bank* b = new (std::nothrow) mlgpu_bank();
    if (!b) return MLGPU_ERR_OOM;
    b->e = e;
    b->V = nVoices;
    for (int p = 0; p < nProcs; ++p)
    {
      const int nc = mlgpu_proc_nc(procs[p]), ns = mlgpu_proc_ns(procs[p]);
      if (nc < 0)
      {
        delete b;
        return fail(e, MLGPU_ERR_INVALID, "bank_create: unknown processor kind");
      }
      if (mlgpu_proc_is_graph_only(procs[p]))
      {
        delete b;
        return fail(e, MLGPU_ERR_UNSUPPORTED,
                    "bank_create: Interpolator1 / LinearGlide (one float per DSPVector in) and the delay lines (per-voice rings) are graph nodes");
      }
      b->kinds.push_back(procs[p]);
      b->cOff.push_back(b->NC);
      b->sOff.push_back(b->NS);
      b->nc.push_back(nc);
      b->ns.push_back(ns);
      b->NC += nc;
      b->NS += ns;
      const int32_t k = procs[p];
      b->singles.push_back(mlgpu_find_chain(&k, 1));
    }
    // strict mode: the ahead-of-time kernels are the fused-2t build, so every chain is generated (hiprtc, MLGPU_SVF_STRICT 1)
    b->fused = e->strictSvf ? nullptr : mlgpu_find_chain(procs, nProcs);
    hipError_t err = hipSetDevice(e->device);
    if (!b->fused && (e->jitEnabled || e->strictSvf) && err == hipSuccess)
    {
      std::string log;
      if (!mlgpu_jit_chain(e, procs, nProcs, &b->jitSignal, &b->jitConst, log))
      {
        if (e->strictSvf)
        {
          delete b;
          return fail(e, MLGPU_ERR_UNSUPPORTED, ("bank_create: strict SVF mode needs hiprtc: " + log).c_str());
        }
        b->jitSignal = b->jitConst = nullptr;  // fall back to processor-by-processor execution
        e->lastError = "hiprtc chain fusion unavailable: " + log;
      }
      else
      {
        b->jitName = "mlgpu_chain_signal/mlgpu_chain_const (hiprtc, Chain<";
        for (int p = 0; p < nProcs; ++p) b->jitName += (p ? ", " : "") + std::to_string(procs[p]);
        b->jitName += ">)";
      }
    }
    const size_t V = nVoices;
    // +1 slot so zero-coefficient / zero-state chains still have a valid base pointer
    if (err == hipSuccess) err = hipMalloc((void**)&b->d_coeffs, sizeof(float) * V * (size_t)(b->NC + 1));
    if (err == hipSuccess) err = hipMalloc((void**)&b->d_state, sizeof(uint32_t) * V * (size_t)(b->NS + 1));
    if (err == hipSuccess) err = hipMalloc((void**)&b->d_inConst, sizeof(float) * V);
    if (err == hipSuccess && !b->fused && !b->jitSignal)
    {
      // unfused chains ping-pong through two QUAD-layout scratch signals, processed in
      // slices of scratchVectors DSPVectors so each scratch stays <= 256 MiB
      const size_t bytesPerVector = V * 64 * sizeof(float);
      b->scratchVectors = std::max<size_t>(1, ((size_t)256 << 20) / bytesPerVector);
      b->scratchVectors = std::min<size_t>(b->scratchVectors, 64);
      err = hipMalloc((void**)&b->d_scratch[0], bytesPerVector * b->scratchVectors);
      if (err == hipSuccess) err = hipMalloc((void**)&b->d_scratch[1], bytesPerVector * b->scratchVectors);
    }
    if (err == hipSuccess) err = hipMemsetAsync(b->d_coeffs, 0, sizeof(float) * V * (size_t)(b->NC + 1), e->stream);
    if (err == hipSuccess) err = hipMemsetAsync(b->d_inConst, 0, sizeof(float) * V, e->stream);
    if (err != hipSuccess)
    {
      const int st = (err == hipErrorOutOfMemory) ? MLGPU_ERR_OOM : MLGPU_ERR_HIP;
      fail(e, st, "bank_create: device allocation", err);
      mlgpu_bank_destroy(b);
      return st;
    }
    // default-constructed state (and non-zero default coefficients) of the reference objects
    for (int p = 0; p < nProcs; ++p)
    {
      float dc[MLGPU_MAX_PROC_COEFFS];
      mlgpu_proc_default_coeffs(b->kinds[p], dc);
      for (int i = 0; i < b->nc[p]; ++i)
      {
        uint32_t u;
        memcpy(&u, &dc[i], 4);
        if (u) mlgpu_launch_fill32((uint32_t*)b->d_coeffs + (size_t)(b->cOff[p] + i) * V, u, V, e->stream);
      }
      uint32_t words[MLGPU_MAX_PROC_STATE];
      mlgpu_proc_clear_state(b->kinds[p], words, false);
      for (int i = 0; i < b->ns[p]; ++i)
      {
        err = mlgpu_launch_fill32(b->d_state + (size_t)(b->sOff[p] + i) * V, words[i], V, e->stream);
        if (err != hipSuccess)
        {
          fail(e, MLGPU_ERR_HIP, "bank_create: state init", err);
          mlgpu_bank_destroy(b);
          return MLGPU_ERR_HIP;
        }
      }
    }
    *out = b;
    return MLGPU_OK;
  }

  size_t mlgpu_bank_num_voices(mlgpu_bank* b) { return b ? b->V : 0; }
  int mlgpu_bank_num_procs(mlgpu_bank* b) { return b ? (int)b->kinds.size() : -1; }
  int mlgpu_bank_num_coeffs(mlgpu_bank* b, int p)
  {
    if (!b || p < 0 || p >= (int)b->kinds.size()) return -1;
    return b->nc[p];
  }
  int mlgpu_bank_num_state(mlgpu_bank* b, int p)
  {
    if (!b || p < 0 || p >= (int)b->kinds.size()) return -1;
    return b->ns[p];
  }
  int mlgpu_bank_is_fused(mlgpu_bank* b) { return (b && (b->fused || b->jitSignal)) ? 1 : 0; }
  const char* mlgpu_bank_kernel_name(mlgpu_bank* b)
  {
    if (!b) return "";
    if (b->fused && b->fused->kernelNameFor) return b->fused->kernelNameFor(b->V, b->e->kflags);
    if (b->fused) return b->fused->kernelName;
    if (b->jitSignal) return b->jitName.c_str();
    return "chain_kernel<per-processor>";
  }

  int mlgpu_bank_clear(mlgpu_bank* b)
  {
    if (!b) return MLGPU_ERR_INVALID;
    mlgpu_engine* e = b->e;
    HIP_TRY(e, hipSetDevice(e->device));
    for (size_t p = 0; p < b->kinds.size(); ++p)
    {
      uint32_t words[MLGPU_MAX_PROC_STATE];
      mlgpu_proc_clear_state(b->kinds[p], words, true);
      for (int i = 0; i < b->ns[p]; ++i)
      {
        // e.g. ADSR::clear() only sets segment = off (MLDSPFilters.h:702); its other members keep their values
        if (!((mlgpu_proc_clear_mask(b->kinds[p]) >> i) & 1)) continue;
        HIP_TRY(e, mlgpu_launch_fill32(b->d_state + (size_t)(b->sOff[p] + i) * b->V, words[i], b->V, e->stream));
      }
    }
    return MLGPU_OK;
  }

  static int checkSlot(mlgpu_bank* b, int p, int idx, bool coeff)
  {
    if (!b) return MLGPU_ERR_INVALID;
    if (p < 0 || p >= (int)b->kinds.size()) return fail(b->e, MLGPU_ERR_RANGE, "processor index out of range");
    const int n = coeff ? b->nc[p] : b->ns[p];
    if (idx < 0 || idx >= n) return fail(b->e, MLGPU_ERR_RANGE, coeff ? "coefficient index out of range" : "state index out of range");
    return MLGPU_OK;
  }

  int mlgpu_bank_set_coeff(mlgpu_bank* b, int p, int idx, const float* h)
  {
    int st = checkSlot(b, p, idx, true);
    if (st) return st;
    if (!h) return fail(b->e, MLGPU_ERR_INVALID, "set_coeff: null");
    return mlgpu_upload(b->e, b->d_coeffs + (size_t)(b->cOff[p] + idx) * b->V, h, sizeof(float) * b->V);
  }
  int mlgpu_bank_get_coeff(mlgpu_bank* b, int p, int idx, float* h)
  {
    int st = checkSlot(b, p, idx, true);
    if (st) return st;
    if (!h) return fail(b->e, MLGPU_ERR_INVALID, "get_coeff: null");
    return mlgpu_download(b->e, h, b->d_coeffs + (size_t)(b->cOff[p] + idx) * b->V, sizeof(float) * b->V);
  }
  int mlgpu_bank_set_coeff_uniform(mlgpu_bank* b, int p, int idx, float value)
  {
    int st = checkSlot(b, p, idx, true);
    if (st) return st;
    uint32_t u;
    memcpy(&u, &value, 4);
    return mlgpu_fill32(b->e, b->d_coeffs + (size_t)(b->cOff[p] + idx) * b->V, u, b->V);
  }
  int mlgpu_bank_get_state(mlgpu_bank* b, int p, int idx, uint32_t* h)
  {
    int st = checkSlot(b, p, idx, false);
    if (st) return st;
    if (!h) return fail(b->e, MLGPU_ERR_INVALID, "get_state: null");
    return mlgpu_download(b->e, h, b->d_state + (size_t)(b->sOff[p] + idx) * b->V, sizeof(uint32_t) * b->V);
  }
  int mlgpu_bank_set_state(mlgpu_bank* b, int p, int idx, const uint32_t* h)
  {
    int st = checkSlot(b, p, idx, false);
    if (st) return st;
    if (!h) return fail(b->e, MLGPU_ERR_INVALID, "set_state: null");
    return mlgpu_upload(b->e, b->d_state + (size_t)(b->sOff[p] + idx) * b->V, h, sizeof(uint32_t) * b->V);
  }
  int mlgpu_bank_set_state_uniform(mlgpu_bank* b, int p, int idx, uint32_t value)
  {
    int st = checkSlot(b, p, idx, false);
    if (st) return st;
    return mlgpu_fill32(b->e, b->d_state + (size_t)(b->sOff[p] + idx) * b->V, value, b->V);
  }
  int mlgpu_bank_set_input_const(mlgpu_bank* b, const float* h)
  {
    if (!b) return MLGPU_ERR_INVALID;
    if (!h) return fail(b->e, MLGPU_ERR_INVALID, "set_input_const: null");
    return mlgpu_upload(b->e, b->d_inConst, h, sizeof(float) * b->V);
  }

  // bank_process + mixdown(gains = NULL) without the voices' signals in between: the voice kernel adds up each wavefront's 64 voices
  // itself (chain_mix_kernel: the first stage's tree, the same bits), the later stages follow
  int mlgpu_bank_process_mixdown(mlgpu_bank* b, size_t T, const float* d_in, int inLayout, const float* d_gains, float* d_out)
  {
    if (!b) return MLGPU_ERR_INVALID;
    mlgpu_engine* e = b->e;
    if (T == 0) return MLGPU_OK;
    if (!d_out) return fail(e, MLGPU_ERR_INVALID, "bank_process_mixdown: null output");
    if (d_in && (inLayout < 0 || inLayout > MLGPU_LAYOUT_BROADCAST)) return fail(e, MLGPU_ERR_INVALID, "bank_process_mixdown: bad layout");
    if ((((uintptr_t)d_out) | ((uintptr_t)d_in)) & 15) return fail(e, MLGPU_ERR_INVALID, "bank_process_mixdown: signals must be 16-byte aligned");
    const size_t V = b->V;
    if (!b->fused || !b->fused->launchMixSignal)
      return fail(e, MLGPU_ERR_UNSUPPORTED, "bank_process_mixdown: for the fused voice chains - use bank_process and mixdown");
    const size_t groups = (V + 63) / 64;
    if ((groups + (groups + 63) / 64) * T * 64 > e->mixScratchFloats)
      return fail(e, MLGPU_ERR_INVALID, "bank_process_mixdown: call mlgpu_mixdown_reserve(engine, max voices, max vectors) at setup (process calls do not allocate)");
    HIP_TRY(e, hipSetDevice(e->device));
    ChainArgs a;
    a.V = V;
    a.T = T;
    a.flags = e->kflags;
    a.impulseTable = e->d_impulseTable;
    a.inConst = b->d_inConst;
    a.coeffs = b->d_coeffs;
    a.state = b->d_state;
    a.in = makeView(d_in, inLayout, V, T);
    a.out = makeView(nullptr, MLGPU_LAYOUT_QUAD, V, T);
    a.mix = e->d_mixScratch;
    a.mixGains = d_gains;
    hipError_t err = d_in ? b->fused->launchMixSignal(a, e->stream, e->cuCount) : b->fused->launchMixConst(a, e->stream, e->cuCount);
    if (err != hipSuccess) return fail(e, MLGPU_ERR_HIP, "bank_process_mixdown launch", err);
    HIP_TRY(e, mlgpu_launch_mixdown_rows(groups, T, e->d_mixScratch, d_out, e->stream, e->kflags));
    return MLGPU_OK;
  }

  int mlgpu_bank_process(mlgpu_bank* b, size_t T, const float* d_in, int inLayout, float* d_out, int outLayout)
  {
    if (!b) return MLGPU_ERR_INVALID;
    mlgpu_engine* e = b->e;
    if (T == 0) return MLGPU_OK;
    if (!d_out) return fail(e, MLGPU_ERR_INVALID, "bank_process: null output");
    if (outLayout < 0 || outLayout > MLGPU_LAYOUT_VOICE_MAJOR || (d_in && (inLayout < 0 || inLayout > MLGPU_LAYOUT_BROADCAST)))
      return fail(e, MLGPU_ERR_INVALID, "bank_process: bad layout");
    if ((((uintptr_t)d_out) | ((uintptr_t)d_in)) & 15) return fail(e, MLGPU_ERR_INVALID, "bank_process: signals must be 16-byte aligned");
    HIP_TRY(e, hipSetDevice(e->device));
    const size_t V = b->V;

    ChainArgs a;
    a.mix = nullptr;
    a.mixGains = nullptr;
    a.V = V;
    a.flags = e->kflags;
    a.impulseTable = e->d_impulseTable;
    a.inConst = b->d_inConst;

    if (b->fused)
    {
      a.coeffs = b->d_coeffs;
      a.state = b->d_state;
      a.T = T;
      a.in = makeView(d_in, inLayout, V, T);
      a.out = makeView(d_out, outLayout, V, T);
      hipError_t err = d_in ? b->fused->launchSignal(a, e->stream, e->cuCount) : b->fused->launchConst(a, e->stream, e->cuCount);
      if (err != hipSuccess) return fail(e, MLGPU_ERR_HIP, "bank_process launch", err);
      return MLGPU_OK;
    }

    if (b->jitSignal)
    {
      a.coeffs = b->d_coeffs;
      a.state = b->d_state;
      a.T = T;
      a.in = makeView(d_in, inLayout, V, T);
      a.out = makeView(d_out, outLayout, V, T);
      hipError_t err = mlgpu_jit_chain_launch(d_in ? b->jitSignal : b->jitConst, a, e->stream);
      if (err != hipSuccess) return fail(e, MLGPU_ERR_HIP, "bank_process launch (jit)", err);
      return MLGPU_OK;
    }

    // unfused: processor by processor over slices of scratchVectors DSPVectors
    const SignalView fullIn = makeView(d_in, inLayout, V, T);
    const SignalView fullOut = makeView(d_out, outLayout, V, T);
    const int nP = (int)b->kinds.size();
    for (size_t t0 = 0; t0 < T; t0 += b->scratchVectors)
    {
      const size_t Tc = std::min(b->scratchVectors, T - t0);
      for (int p = 0; p < nP; ++p)
      {
        a.coeffs = b->d_coeffs + (size_t)b->cOff[p] * V;
        a.state = b->d_state + (size_t)b->sOff[p] * V;
        a.T = Tc;
        bool hasSignal;
        if (p == 0)
        {
          hasSignal = (d_in != nullptr);
          a.in = fullIn;
          if (hasSignal) a.in.base = fullIn.base + t0 * fullIn.strideT;
        }
        else
        {
          hasSignal = true;
          a.in = makeView(b->d_scratch[(p - 1) & 1], MLGPU_LAYOUT_QUAD, V, Tc);
        }
        if (p == nP - 1)
        {
          a.out = fullOut;
          a.out.base = fullOut.base + t0 * fullOut.strideT;
        }
        else
        {
          a.out = makeView(b->d_scratch[p & 1], MLGPU_LAYOUT_QUAD, V, Tc);
        }
        const ChainEntry* ce = b->singles[p];
        hipError_t err = hasSignal ? ce->launchSignal(a, e->stream, e->cuCount) : ce->launchConst(a, e->stream, e->cuCount);
        if (err != hipSuccess) return fail(e, MLGPU_ERR_HIP, "bank_process launch (unfused)", err);
      }
    }
    return MLGPU_OK;
  }
}
