/*
 * mlgpu.h — C-ABI of the MI355X-native DSPVector engine.
 *
 * This is the drop-in boundary for the mldsp.h hot path of madronalabs/madronalib
 * (reference include/mldsp.h:7-16). The reference has no FFI of its own: it is a
 * header-only C++ template library whose "operator interface" is
 *   B1  the functor protocol  `DSPVector T::operator()(const DSPVector in)` +
 *       `static Coeffs T::makeCoeffs(...)` + `void T::clear()`
 *       (e.g. source/DSP/MLDSPFilters.h:199-240, source/DSP/MLDSPGens.h:395-402),
 *   B2  the voice-bank protocol `Bank<T,ROWS>` (source/DSP/MLDSPFunctional.h:321-360),
 *   B3  the C-style process callback `SignalProcessFn = void(*)(AudioContext*, void*)`
 *       (source/app/MLSignalProcessBuffer.h:18).
 * Every entry point below cites the reference interface it replaces.
 *
 * Conventions
 *  - plain pointers and sizes only; no C++/torch types cross this boundary.
 *  - every function returns an mlgpu_status (0 = OK). The reference has no error
 *    channel (it silently returns: source/app/MLSignalProcessBuffer.cpp:42-44).
 *  - "d_" pointers are device (HBM) pointers on the engine's GPU, "h_" are host.
 *  - one caller thread per engine (mirrors the single audio thread of the reference);
 *    no locks, no allocation inside mlgpu_bank_process / mlgpu_op_apply.
 *  - all work is enqueued on the engine's HIP stream; mlgpu_engine_sync waits for it.
 *  - there is NO CPU fallback: without a gfx950 device every compute entry fails
 *    with MLGPU_ERR_NO_DEVICE.
 *
 * Element type is float32 (int32/uint32 for the integer ops) exactly as in the
 * reference (source/DSP/MLDSPOps.h:115-123). One DSPVector = 64 floats.
 *
 * Numerical contract (what "the same result as the reference" means for every compute entry below):
 *  - every op, generator and filter returns the bits of the reference's SSE2 build, outputs and final state, for every
 *    input including infinities, denormals and out-of-range conversions, in both floating-point modes
 *    (mlgpu_engine_set_flush_denormals) - with two stated exceptions:
 *  - a float result that is NaN is "some NaN": payloads and signs of NaNs are not reproduced (x86 and gfx950 propagate
 *    them differently), so parity tests compare any NaN == any NaN; integer and mask results are exact. (x86 makes the NaN
 *    of an invalid operation negative, gfx950 positive: an operation that looks at the BITS of a NaN produced earlier in the
 *    same computation - sign(), signBit(), an integer reinterpretation - can therefore see the other sign);
 *  - the hardware-approximate operations - MLGPU_OP_SQRT_APPROX, MLGPU_OP_DIVIDE_APPROX and the OUTPUTS of MLGPU_PROC_PEAK
 *    and MLGPU_PROC_RMS (their state is exact) - use v_rsq_f32 / v_rcp_f32 where the reference uses x86 rsqrtps / rcpps
 *    (12-bit tables no other hardware reproduces): relative error <= 1.5 * 2^-11 against the reference;
 *  - a state-variable filter whose internal state has passed FLT_MAX / 2 (1.7e38: a filter that has blown up) may reach
 *    infinity one sample later than the reference does (a fused 2 t + s where the reference rounds 2 t first) - unless the
 *    engine is in strict mode (mlgpu_engine_set_strict_svf), whose kernels spend the second instruction and are exact there too.
 *
 * ABI history: MLGPU_ABI_VERSION 2 (round 3). Against version 1: mlgpu_mixdown needs mlgpu_mixdown_reserve first (it used to
 * grow its scratch on demand), GraphArgs grew (32 inputs, 8 outputs), and the entries added since - strict SVF, cascade
 * lanes, device-source fingerprint, windows, transports, registry, published signals ... - are only in a version-2 library.
 * A host checks mlgpu_abi_version() == MLGPU_ABI_VERSION once at load.
 */
#ifndef MLGPU_H
#define MLGPU_H

#ifndef __HIPCC_RTC__ /* hiprtc (the engine's own run-time kernel generator) predefines size_t */
#include <stddef.h>
#endif
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MLGPU_FLOATS_PER_DSPVECTOR 64 /* kFloatsPerDSPVector, source/DSP/MLDSPMath.h:8-9 */
#define MLGPU_ABI_VERSION 2

/* ------------------------------------------------------------------------- */
/* status codes                                                              */

typedef enum mlgpu_status
{
  MLGPU_OK = 0,
  MLGPU_ERR_INVALID = 1,     /* null pointer, bad enum, bad size */
  MLGPU_ERR_NO_DEVICE = 2,   /* no HIP device / not gfx950 */
  MLGPU_ERR_HIP = 3,         /* a HIP runtime call failed; see mlgpu_last_error */
  MLGPU_ERR_OOM = 4,         /* device or host allocation failed */
  MLGPU_ERR_UNSUPPORTED = 5, /* valid request this build has no kernel for */
  MLGPU_ERR_RANGE = 6,       /* index out of range (proc, coeff, state, voice) */
  MLGPU_ERR_BUSY = 7         /* the object is being worked on by a job the caller started (mlgpu_graph_compile_async): ask again */
} mlgpu_status;

/* ------------------------------------------------------------------------- */
/* signal layouts in HBM                                                     */
/*
 * A "signal" is S = 64*T samples for each of V voices.
 *  QUAD        [S/4][V][4]  device-native: the 4-sample SIMD group of the reference
 *                           (SIMDVectorFloat, MLDSPMathSSE.h:51) of voice v is one
 *                           16-byte word; consecutive lanes (voices) are adjacent, so a
 *                           wavefront moves 1 KiB per memory instruction.
 *  ROWS        [T][V][64]   the reference's DSPVectorArray<V> per vector step
 *                           (row-major [ROWS][64], MLDSPOps.h:283-310).
 *  VOICE_MAJOR [V][S]       each voice's whole stream contiguous (host / oracle order).
 * For T == 1, ROWS and VOICE_MAJOR coincide.
 */
typedef enum mlgpu_layout
{
  MLGPU_LAYOUT_QUAD = 0,
  MLGPU_LAYOUT_ROWS = 1,
  MLGPU_LAYOUT_VOICE_MAJOR = 2,
  /* INPUT signals only: ONE voice's stream [S], read by every voice (e.g. one host audio channel feeding all voices) */
  MLGPU_LAYOUT_BROADCAST = 3
} mlgpu_layout;

/* ------------------------------------------------------------------------- */
/* stateless elementwise ops  (reference: source/DSP/MLDSPOps.h)             */
/*
 * Operands are flat arrays of n 4-byte elements (a DSPVectorArray<ROWS> is
 * n = 64*ROWS elements); elementwise ops are layout-agnostic. Unused operand
 * pointers must be NULL. Integer-typed operands/results are int32 bit patterns
 * (the reference stores them in float-typed storage too: MLDSPOps.h:370-498).
 */
typedef enum mlgpu_op
{
  /* unary float -> float   MLDSPOps.h:584-614, 825 */
  MLGPU_OP_SQRT = 0,
  MLGPU_OP_SQRT_APPROX = 1, /* x*rsqrt(x): hardware-approximate, 2^-11 rel */
  MLGPU_OP_ABS = 2,
  MLGPU_OP_SIGN = 3,
  MLGPU_OP_SIGN_BIT = 4,
  MLGPU_OP_SIN = 5,
  MLGPU_OP_COS = 6,
  MLGPU_OP_LOG = 7,
  MLGPU_OP_EXP = 8,
  MLGPU_OP_LOG2 = 9,
  MLGPU_OP_EXP2 = 10,
  MLGPU_OP_SIN_APPROX = 11,
  MLGPU_OP_COS_APPROX = 12,
  MLGPU_OP_EXP_APPROX = 13,
  MLGPU_OP_LOG_APPROX = 14,
  MLGPU_OP_LOG2_APPROX = 15,
  MLGPU_OP_EXP2_APPROX = 16,
  MLGPU_OP_FRACTIONAL_PART = 17,
  /* float -> int32         MLDSPOps.h:796-797 */
  MLGPU_OP_ROUND_FLOAT_TO_INT = 18,
  MLGPU_OP_TRUNCATE_FLOAT_TO_INT = 19,
  /* int32 -> float         MLDSPOps.h:819-820 */
  MLGPU_OP_INT_TO_FLOAT = 20,
  MLGPU_OP_UNSIGNED_INT_TO_FLOAT = 21,
  /* fused pair used by the config-2 bench: expApprox(sinApprox(x)) */
  MLGPU_OP_EXP_APPROX_OF_SIN_APPROX = 22,
  /* the waveshape functions of MLDSPGens.h:313-369 (public free functions over a phasor on [0, 1)) */
  MLGPU_OP_PHASOR_TO_SINE = 23,   /* phasorToSine(phasor) */

  /* binary float,float -> float   MLDSPOps.h:640-649 */
  MLGPU_OP_ADD = 32,
  MLGPU_OP_SUBTRACT = 33,
  MLGPU_OP_MULTIPLY = 34,
  MLGPU_OP_DIVIDE = 35,
  MLGPU_OP_DIVIDE_APPROX = 36, /* a*rcp(b): hardware-approximate, 2^-11 rel */
  MLGPU_OP_POW = 37,
  MLGPU_OP_POW_APPROX = 38,
  MLGPU_OP_MIN = 39, /* SSE semantics (a<b)?a:b */
  MLGPU_OP_MAX = 40, /* SSE semantics (a>b)?a:b */
  /* binary int32,int32 -> int32   MLDSPOps.h:713-714 */
  MLGPU_OP_ADD_INT32 = 41,
  MLGPU_OP_SUBTRACT_INT32 = 42,
  /* binary float,float -> int32 mask   MLDSPOps.h:851-856 */
  MLGPU_OP_EQUAL = 43,
  MLGPU_OP_NOT_EQUAL = 44,
  MLGPU_OP_GREATER_THAN = 45,
  MLGPU_OP_GREATER_THAN_OR_EQUAL = 46,
  MLGPU_OP_LESS_THAN = 47,
  MLGPU_OP_LESS_THAN_OR_EQUAL = 48,
  MLGPU_OP_PHASOR_TO_SAW = 49,    /* phasorToSaw(phasor, freq): band-limited by polyBLEP */

  /* ternary   MLDSPOps.h:744-748, 886, 917 */
  MLGPU_OP_LERP = 64,         /* lerp(a, b, mix) = a + mix*(b - a) */
  MLGPU_OP_INVERSE_LERP = 65, /* (x - a)/(b - a) with operands (a, b, x) */
  MLGPU_OP_CLAMP = 66,        /* clamp(x, lo, hi) = min(max(x, lo), hi) */
  MLGPU_OP_WITHIN = 67,       /* mask: lo <= x < hi */
  MLGPU_OP_SELECT = 68,       /* select(a, b, maskInt): bitwise (m&a)|(~m&b) */
  MLGPU_OP_SELECT_INT = 69,
  MLGPU_OP_PHASOR_TO_PULSE = 70   /* phasorToPulse(phasor, freq, pulseWidth) */
} mlgpu_op;

/* ------------------------------------------------------------------------- */
/* stateful processors (reference: MLDSPGens.h, MLDSPFilters.h)              */
/*
 * A processor has NC per-voice coefficients (float) and NS per-voice state words
 * (float or uint32 bit patterns), in the order listed. Its audio-rate input is the
 * previous processor's output (or the bank input for processor 0): for generators the
 * input is cyclesPerSample (f/sr), exactly the argument of the reference's operator().
 */
typedef enum mlgpu_proc
{
  /* generators, MLDSPGens.h */
  MLGPU_PROC_PHASOR_GEN = 0,  /* :177-217  C{}            S{omega32:u32} */
  MLGPU_PROC_SINE_GEN = 1,    /* :373-381  C{}            S{omega32:u32}; clear() -> 0xC0000000 */
  MLGPU_PROC_SAW_GEN = 2,     /* :395-402  C{}            S{omega32:u32} */
  MLGPU_PROC_PULSE_GEN = 3,   /* :383-393  C{width}       S{omega32:u32} (width per voice, constant) */
  MLGPU_PROC_NOISE_GEN = 4,   /* :109-148  C{}            S{seed:u32}; no input */
  MLGPU_PROC_TICK_GEN = 5,    /* :24-47    C{}            S{omega} */
  MLGPU_PROC_IMPULSE_GEN = 6, /* :53-104   C{}            S{omega, outputCounter:i32}; 17-tap table in LDS */
  MLGPU_PROC_ONE_SHOT_GEN = 7,/* :221-282  C{}            S{omega32:u32, gate:u32, omegaPrev:u32} */
  MLGPU_PROC_TEST_SINE_GEN = 8, /* :151-171 C{}           S{omega}: omega += 2 pi f, wrapped at 2 pi; y = sinf(omega), the host
                               *                          libm's sinf (glibc 2.35) restated on the device */
  /* SVF family, MLDSPFilters.h */
  MLGPU_PROC_LOPASS = 16,     /* :51-153   C{g0,g1,g2}    S{ic1eq,ic2eq} */
  MLGPU_PROC_HIPASS = 17,     /* :155-197  C{g0,g1,g2,k}  S{ic1eq,ic2eq} */
  MLGPU_PROC_BANDPASS = 18,   /* :199-240  C{g0,g1,g2}    S{ic1eq,ic2eq} */
  MLGPU_PROC_LO_SHELF = 19,   /* :242-319  C{a1,a2,a3,m1,m2}    S{ic1eq,ic2eq} */
  MLGPU_PROC_HI_SHELF = 20,   /* :321-400  C{a1,a2,a3,m0,m1,m2} S{ic1eq,ic2eq} */
  MLGPU_PROC_BELL = 21,       /* :402-442  C{a1,a2,a3,m1} S{ic1eq,ic2eq} */
  /* one-state recurrences, MLDSPFilters.h */
  MLGPU_PROC_ONE_POLE = 32,   /* :446-481  C{a0,b1}       S{y1} */
  MLGPU_PROC_DC_BLOCKER = 33, /* :489-513  C{c}           S{x1,y1} */
  MLGPU_PROC_DIFFERENTIATOR = 34, /* :517-535 C{}         S{x1} */
  MLGPU_PROC_INTEGRATOR = 35, /* :539-558  C{leak}        S{y1} */
  MLGPU_PROC_PEAK = 36,       /* :562-615  C{a0,b1,peakHoldSamples:i32} S{y1,peakHoldCounter:i32}; sqrtApprox: 2^-11 rel */
  MLGPU_PROC_RMS = 37,        /* :619-653  C{a0,b1}       S{y1}; sqrtApprox: 2^-11 rel */
  MLGPU_PROC_ADSR = 38,       /* :657-797  C{ka,kd,s,kr}  S{y,y1,x1,threshold,target,k,amp,segment:i32} */
  /* stateless per-voice scaling: `x * DSPVector(gain)`, MLDSPOps.h:157,345-348 */
  MLGPU_PROC_GAIN = 48,       /*           C{gain}        S{} */
  /* control-rate -> audio-rate ramps, MLDSPGens.h:404-590. INTERPOLATOR1 and LINEAR_GLIDE take one float per
   * DSPVector (`operator()(float)`): graph nodes only, fed by a control input / param / const. */
  MLGPU_PROC_INTERPOLATOR1 = 64, /* :412-423 C{}          S{currentValue} */
  MLGPU_PROC_LINEAR_GLIDE = 65,  /* :433-515 C{vectorsPerGlide:i32, dyPerVector} S{target, step, vectorsRemaining:i32,
                                  *          currVec[64]}; defaults 32, 1/32, remaining -1 (:437-441);
                                  *          setGlideTimeInSamples: mlgpu_linear_glide_make_coeffs;
                                  *          setValue(f): target = f, vectorsRemaining = 0 */
  MLGPU_PROC_SAMPLE_ACCURATE_LINEAR_GLIDE = 66, /* :517-590 C{samplesPerGlide:i32, dyPerSample}
                                  *          S{curr, step, target, samplesRemaining:i32}; nextSample per sample */
  /* delay lines, MLDSPFilters.h:799-1106: per-voice rings in HBM, sized with mlgpu_graph_set_max_delay (graph nodes
   * only). Valid delays are 0 <= d <= ring length - 64, as in the reference (:817-819). */
  MLGPU_PROC_INTEGER_DELAY = 80,      /* :801-914   C{}      S{writeIndex:u32, delayInSamples:i32}; forms (x), (x, delay) */
  MLGPU_PROC_ALLPASS1 = 81,           /* :918-964   C{coeff} S{x1, y1}; coeff from mlgpu_allpass1_make_coeffs */
  MLGPU_PROC_FRACTIONAL_DELAY = 82,   /* :971-1044  C{}      S{writeIndex, x1, y1, delayInt:i32, allpassCoeff}; forms (x),
                                       *            (x, delay), (x, delay, changeTicks:int mask); setDelayInSamples:
                                       *            mlgpu_fractional_delay_make_state -> state words 3, 4 */
  MLGPU_PROC_PITCHBENDABLE_DELAY = 83, /* :1050-1106 C{}      S{delay1[5], delay2[5]} two rings; form (x, delay) */
  /* TempoLock, MLDSPFilters.h:1478-1579: operator()(DSPVector x, float dydx, float isr). Graph node with 3 inputs: x the phasor to
   * follow - a streamed INPUT node (only x[0], x[1] of each vector are read, once per vector) or any signal computed in the graph
   * (taken sample by sample; same results) -, dydx and isr one float per vector (control / param / const). */
  MLGPU_PROC_TEMPO_LOCK = 96,         /* C{} S{omega (phase, -1 = stopped), x1v}; clear(): omega = -1 */
  /* HalfBandFilter, MLDSPFilters.h:1245-1310: the resampling filter at the edges of a rate region
   * (mlgpu_graph_begin_region / end_region create these nodes; they cannot be added directly).
   * S{apa0.x1, apa0.y1, apa1.x1, apa1.y1, apb0.x1, apb0.y1, apb1.x1, apb1.y1, b1} as mlgpu_resampler. */
  MLGPU_PROC_HALF_BAND = 112,
  MLGPU_PROC_HALF_BAND_BUFFERED = 113 /* + Downsample2xFunction::mOutputBuffer: S{the 9 above, 64 samples} */
} mlgpu_proc;

/* rate regions of a graph: Upsample2xFunction / Downsample2xFunction, MLDSPFunctional.h:114-213 */
typedef enum mlgpu_region
{
  MLGPU_REGION_UPSAMPLE_2X = 0,   /* the nodes inside run at twice the rate (two samples per outer sample) */
  MLGPU_REGION_DOWNSAMPLE_2X = 1  /* the nodes inside run at half the rate; one DSPVector of delay (:165-170) */
} mlgpu_region;

/* ------------------------------------------------------------------------- */
/* engine                                                                    */

typedef struct mlgpu_engine mlgpu_engine;

/* Create an engine bound to HIP device `device` with its own non-blocking stream. */
int mlgpu_engine_create(int device, mlgpu_engine** out);
/* Same, with a stream priority: urgency +1 = the device's greatest priority, 0 = normal (== mlgpu_engine_create), -1 = its least.
 * For a second engine on the same device whose short, memory-bound launches should slip in beside a long launch of the first (see
 * mlgpu_fence below): the dispatcher serves the more urgent stream's workgroups first. */
int mlgpu_engine_create_urgency(int device, int urgency, mlgpu_engine** out);
/* Same, but enqueue on a caller-owned hipStream_t (e.g. torch's current stream). */
int mlgpu_engine_create_on_stream(int device, void* hip_stream, mlgpu_engine** out);
int mlgpu_engine_destroy(mlgpu_engine* e);
/* Block until all enqueued work is done. */
int mlgpu_engine_sync(mlgpu_engine* e);
/* Floating-point mode of every kernel this engine launches from now on (stream-ordered with the launches).
 * Reference: ml::UsingFlushDenormalsToZero (source/DSP/MLDSPUtils.h:51-96) sets MXCSR FZ | DAZ for the scope of a
 * process function: denormal operands read as zero, denormal results are written as zero (signs kept). on != 0 gives the
 * kernels the same rule (gfx950 MODE.fp_denorm, set at kernel entry; one code object serves both modes); 0 (the default,
 * like the reference's default MXCSR) honours denormals. Values that are only moved, selected or bit-manipulated pass
 * through unchanged in both modes, on both machines. Results are bit-identical to the reference run under the same mode
 * (tests/test_gpu_denormals.py). Not allowed while a launch sequence is being recorded. */
int mlgpu_engine_set_flush_denormals(mlgpu_engine* e, int on);
int mlgpu_engine_get_flush_denormals(mlgpu_engine* e);
/* How a bank that is a plain cascade of SVF sections (Lopass / Hipass / Bandpass x 2, 4 or 8: the reference's
 * `for (auto& f : filters) x = f(x)`, source/DSP/MLDSPFilters.h:118-133 per section) is laid over the wavefront:
 * `lanes` = 1, 2 or 4 wavefront lanes per channel (each lane runs 1/lanes of the sections and hands its output to the
 * next lane with a DPP move; more lanes = more wavefronts for a small bank), 0 (the default) = chosen from the bank's
 * size, -1 = the one-lane kernel of round 2. Every form computes each section with the same operations in the same
 * order: results are bit-identical (tests/test_gpu_parity.py::test_cascade_forms_agree). A tuning knob, not a contract.
 * Not allowed while a launch sequence is being recorded. */
int mlgpu_engine_set_cascade_lanes(mlgpu_engine* e, int lanes);
/* Strict SVF arithmetic for the banks and graphs created from now on. The state-variable filters (Lopass, Hipass,
 * Bandpass, LoShelf, HiShelf, Bell; source/DSP/MLDSPFilters.h:118-133, 189-193, 234-237, 288-302, 369-383, 427-441) update
 * their memories with `ic += 2.0f * t` / `ic = 2 * v - ic`. By default that is one fused instruction - the same float
 * unless 2 t alone overflows while the sum does not (|t| > 1.7e38, a filter already blowing up; numerical contract above).
 * on != 0 spends the second instruction, and such kernels equal the reference bit for bit in that corner too
 * (tests/test_gpu_parity.py::test_strict_svf_hostile_input). Kernels of a strict engine are generated with hiprtc when the
 * bank or graph is made (cached on disk); existing banks and graphs keep the arithmetic they were made with. Cost:
 * profiles/archive/r03_strict_svf.txt. */
int mlgpu_engine_set_strict_svf(mlgpu_engine* e, int on);
int mlgpu_engine_get_strict_svf(mlgpu_engine* e);
int mlgpu_engine_get_cascade_lanes(mlgpu_engine* e);
/* Two engines on one device are two HIP streams: their work overlaps on the GPU (e.g. the HBM-bound events kernel of block
 * k + 1 under the VALU-bound voice kernel of block k: bench.py --workload synth --two-streams). A fence orders them
 * where they share a buffer: mlgpu_engine_signal(a, f) marks a point in a's stream, mlgpu_engine_wait(b, f) makes everything b
 * enqueues afterwards wait for that point (the host never blocks). Waiting for a fence that was never signalled is a no-op -
 * convenient for the first trip round a ring of buffers. Same device only; not while recording a sequence. */
typedef struct mlgpu_fence mlgpu_fence;
int mlgpu_fence_create(mlgpu_engine* e, mlgpu_fence** out);
int mlgpu_fence_destroy(mlgpu_fence* f);
int mlgpu_engine_signal(mlgpu_engine* e, mlgpu_fence* f);
int mlgpu_engine_wait(mlgpu_engine* e, mlgpu_fence* f);
/* The hipStream_t work is enqueued on (for HIP-event timing by the caller). */
void* mlgpu_engine_stream(mlgpu_engine* e);
int mlgpu_engine_device(mlgpu_engine* e);
/* Human-readable text for the last failure on this engine (never NULL). */
const char* mlgpu_last_error(mlgpu_engine* e);
const char* mlgpu_status_string(int status);
int mlgpu_abi_version(void);
/* A number that goes up whenever a DOCUMENTED behaviour of an existing entry point changes under an unchanged ABI (same symbols,
 * same arguments, other last bits or other side effects), so that a host can tell which contract it runs against:
 *   1  rounds 1-3
 *   2  round 4: mlgpu_mixdown above 4 096 voices adds the group sums 64 at a time (other last bits than the serial chain of
 *      rounds 1-3; INTEGRATION.md "Behavioural differences")
 *   3  round 5: MLGPU_ERR_BUSY exists (a graph with a compile in flight); mlgpu_graph_process_events runs the events'
 *      control-rate half as a kernel of its own before the voice kernel (same outputs and state, one launch more per block) */
int mlgpu_behaviour_revision(void);
/* SHA-256 (hex) of every device source file of this build and its compiler flags. Measurements that cannot be taken
 * inside a run (rocprofv3 PMC counters, profiles/pmc_workloads.json) are recorded with it and refused by bench.py when
 * the loaded library's differs, so a kernel change cannot inherit old counters. */
const char* mlgpu_device_source_hash(void);
/* Number of visible HIP devices (0 when there is none); never fails. */
int mlgpu_device_count(void);
/* Device facts used by the bench (name, CU count, memory bytes). */
int mlgpu_device_info(int device, char* name, size_t name_len, int* cu_count, uint64_t* mem_bytes);
/* PCI bus id of a device ("0000:05:00.0"): what tells two ranks of a multi-GPU job that they really run on different
 * GPUs (a device INDEX is relative to each process's HIP_VISIBLE_DEVICES). */
int mlgpu_device_pci_bus_id(int device, char* buf, size_t buf_len);
/* Block until everything enqueued on `device` by this process, on any stream, is done (hipDeviceSynchronize): the
 * device-wide fence a benchmark puts on both sides of its timed region. */
int mlgpu_device_synchronize(int device);

/* device memory owned by the caller, allocated on the engine's device */
int mlgpu_alloc(mlgpu_engine* e, size_t bytes, void** d_out);
int mlgpu_free(mlgpu_engine* e, void* d_ptr);
int mlgpu_upload(mlgpu_engine* e, void* d_dst, const void* h_src, size_t bytes);
int mlgpu_download(mlgpu_engine* e, void* h_dst, const void* d_src, size_t bytes);
int mlgpu_fill32(mlgpu_engine* e, void* d_dst, uint32_t value, size_t n_elems);

/* validate(), the reference's debugging check (source/DSP/MLDSPOps.h:1430-1445): a sample is bad when it is NaN or larger
 * than 1e8 in magnitude. One read-only pass over n floats of a device signal (any layout: the layout only decides what
 * the index means); *count = how many bad samples, *first_index = flat index of the first one (~0 when none; may be NULL).
 * Waits for the result. */
int mlgpu_validate(mlgpu_engine* e, const float* d_signal, size_t n_elems, uint64_t* count, uint64_t* first_index);

/* HIP-event timing on the engine's stream (events are created lazily, reused). */
/* Recorded launch sequences (hipGraph). A real-time host calls the same few process functions on the same device buffers
 * every block, and with small banks those launches are launch-bound (microseconds of kernel behind ~6 us of launch each).
 * Between begin_recording and end_recording the engine's launches - bank / graph / op / rows / routing / resampler / mixdown /
 * layout_convert calls - are captured instead of run; mlgpu_sequence_launch replays them all with ONE graph launch, with the
 * arguments (buffers, vector counts) they were recorded with. Not recordable (MLGPU_ERR_INVALID while recording): calls that
 * wait for the device or work on the host per call (upload / download / sync, events_process, published_signal_write,
 * process_buffer), and graphs that count DSPVectors or are still tuning (a DOWNSAMPLE_2X region, autotune not settled). */
typedef struct mlgpu_sequence mlgpu_sequence;
int mlgpu_engine_begin_recording(mlgpu_engine* e);
int mlgpu_engine_end_recording(mlgpu_engine* e, mlgpu_sequence** out);
int mlgpu_sequence_launch(mlgpu_sequence* s);
size_t mlgpu_sequence_num_nodes(mlgpu_sequence* s);
int mlgpu_sequence_destroy(mlgpu_sequence* s);

int mlgpu_timer_start(mlgpu_engine* e);
int mlgpu_timer_stop_ms(mlgpu_engine* e, float* ms_out); /* records + waits */
/* Lap timer (measurement aid, no reference counterpart; the reference times with std::chrono around its loops, Tests/testUtils.h:136-189):
 * laps_begin records an event on the engine's stream, every lap another, laps_end waits for the last and writes the
 * durations between consecutive events (ms) - the DISTRIBUTION of a kernel's launch times, where timer_start / stop give the mean. */
int mlgpu_timer_laps_begin(mlgpu_engine* e, size_t max_laps);
int mlgpu_timer_lap(mlgpu_engine* e);
int mlgpu_timer_laps_end(mlgpu_engine* e, float* ms_out, size_t capacity, size_t* n_out);

/* ------------------------------------------------------------------------- */
/* stateless ops                                                             */

/* out[i] = op(a[i] [, b[i] [, c[i]]]) for i < n_elems.
 * Replaces the DEFINE_OP1/OP2/OP3/... families of MLDSPOps.h:570-917. */
int mlgpu_op_apply(mlgpu_engine* e, int op, const void* d_a, const void* d_b, const void* d_c,
                   void* d_out, size_t n_elems);

/* ROWS x 1-row broadcast forms add1..max1 (MLDSPOps.h:655-687): b has 64 elements
 * and repeats for every row of a. `op` is one of MLGPU_OP_ADD..MLGPU_OP_MAX. */
int mlgpu_op_apply_rows1(mlgpu_engine* e, int op, const void* d_a, const void* d_b64, void* d_out,
                         size_t n_rows);

/* Horizontal per-row reductions sum/mean/max/min (MLDSPOps.h:995-1035), one float
 * per row of 64, association order and FLT_MIN/FLT_MAX seeds as the reference. */
typedef enum mlgpu_rowop
{
  MLGPU_ROWOP_SUM = 0,
  MLGPU_ROWOP_MEAN = 1,
  MLGPU_ROWOP_MAX = 2,
  MLGPU_ROWOP_MIN = 3
} mlgpu_rowop;
int mlgpu_row_reduce(mlgpu_engine* e, int rowop, const float* d_rows, float* d_out, size_t n_rows);

/* Re-lay a V-voice, T-vector signal between layouts (pure data movement; replaces the
 * row plumbing a host would do with row()/setRowVector, MLDSPOps.h:283-310). */
int mlgpu_layout_convert(mlgpu_engine* e, const float* d_src, int src_layout, float* d_dst,
                         int dst_layout, size_t n_voices, size_t n_vectors);

/* ------------------------------------------------------------------------- */
/* row plumbing and routing                                                  */
/*
 * Row plumbing (MLDSPOps.h:1057-1343) is data movement between DSPVectorArrays: destination row j of each
 * group (= one DSPVectorArray; `n_groups` of them per call, e.g. one per voice) is a copy of source row
 * rule(j) of the same group, or zeros. For j in [0, count):
 *     dst[g][dst_offset + j*dst_step] = rotate(src[g][rule(j)], sample_rotate)
 * with src_rows / dst_rows rows per group. sample_rotate: 0, +1 = rotateLeft (y[n] = x[(n+1)%64], :1219-1245),
 * -1 = rotateRight (:1249-1276). The reference's functions are one or a few calls:
 *   repeatRows<R>(x)      REPEAT,  count = R*N                       stretchRows<R>(x)  STRETCH, count = R
 *   zeroPadRows<R>(x)     SHIFT p0 = 0, count = R                    shiftRows(x, k)    SHIFT p0 = k
 *   rotateRows(x, k)      ROTATE p0 = k                              separateRows<A,B>  STRIDED p0 = A, p1 = 1, count = B-A
 *   evenRows / oddRows    STRIDED p0 = 0 / 1, p1 = 2                 concatRows(a, b..) one STRIDED call per operand, dst_offset
 *   shuffleRows(a, b)     STRIDED calls with dst_step = 2 for the interleaved part, 1 for the appended excess
 */
typedef enum mlgpu_rows_rule
{
  MLGPU_ROWS_REPEAT = 0,  /* rule(j) = j % src_rows                                  :1057-1068 */
  MLGPU_ROWS_STRETCH = 1, /* rule(j) = roundf(j*(src_rows-1.f)/(count-1.f))          :1073-1083 */
  MLGPU_ROWS_SHIFT = 2,   /* rule(j) = j - p0; rows from outside [0, src_rows) are 0 :1088-1121 */
  MLGPU_ROWS_ROTATE = 3,  /* rule(j) = (j - p0) mod src_rows                         :1126-1139 */
  MLGPU_ROWS_STRIDED = 4  /* rule(j) = p0 + j*p1; outside [0, src_rows) -> 0         :1145-1343 */
} mlgpu_rows_rule;
int mlgpu_rows_map(mlgpu_engine* e, int rule, long p0, long p1, int sample_rotate, const float* d_src, size_t src_rows,
                   float* d_dst, size_t dst_rows, size_t dst_offset, size_t dst_step, size_t count, size_t n_groups);
/* addRows (MLDSPOps.h:1349-1359): out[g] = ((0 + row 0) + row 1) + ... over each group's rows. */
int mlgpu_rows_add(mlgpu_engine* e, const float* d_rows, size_t rows_per_group, float* d_out, size_t n_groups);
/* normalize (MLDSPOps.h:1041-1050): every row divided by its sum(). */
int mlgpu_rows_normalize(mlgpu_engine* e, const float* d_rows, float* d_out, size_t n_rows);
/* rowIndex<ROWS>() (MLDSPOps.h:1365-1374): row j of every group filled with (float)j. */
int mlgpu_rows_index(mlgpu_engine* e, float* d_out, size_t rows_per_group, size_t n_groups);

/* Routing (MLDSPRouting.h:83-234): per-sample selection among n signals by a selector in [0, 1).
 * Signals are flat arrays of n_elems floats; selector element i % sel_elems is used for element i
 * (sel_elems = 64: one selector DSPVector for every row, as the reference; sel_elems = n_elems: a selector
 * per sample). `linear`: multiplexLinear / demultiplexLinear. A negative or NaN selector is undefined
 * behaviour in the reference (out-of-bounds read); here it selects index 0. n <= MLGPU_ROUTE_MAX_SIGNALS. */
#define MLGPU_ROUTE_MAX_SIGNALS 8
int mlgpu_multiplex(mlgpu_engine* e, const float* d_selector, size_t sel_elems, const float* const* d_inputs, int n_inputs,
                    float* d_out, size_t n_elems, int linear);
int mlgpu_demultiplex(mlgpu_engine* e, const float* d_selector, size_t sel_elems, const float* d_input,
                      float* const* d_outputs, int n_outputs, size_t n_elems, int linear);

/* ------------------------------------------------------------------------- */
/* voice banks                                                               */
/*
 * mlgpu_bank is the runtime-sized counterpart of `Bank<T,ROWS>`
 * (MLDSPFunctional.h:321-360) where T is a chain of reference processors applied in
 * order, e.g. {SAW_GEN, BANDPASS, GAIN} is `bp(saw(freq)) * gain`
 * (SURVEY §3.2). One wavefront lane runs one voice; the DSPVector is walked serially
 * with the voice's state in registers; state lives in HBM (SoA) between calls.
 */
typedef struct mlgpu_bank mlgpu_bank;

int mlgpu_bank_create(mlgpu_engine* e, const int32_t* procs, int n_procs, size_t n_voices,
                      mlgpu_bank** out);
int mlgpu_bank_destroy(mlgpu_bank* b);
size_t mlgpu_bank_num_voices(mlgpu_bank* b);
int mlgpu_bank_num_procs(mlgpu_bank* b);
/* NC / NS of processor `proc_idx` (see mlgpu_proc). Negative = error. */
int mlgpu_bank_num_coeffs(mlgpu_bank* b, int proc_idx);
int mlgpu_bank_num_state(mlgpu_bank* b, int proc_idx);

/* T::clear() on every processor of every voice (Bank::clear, MLDSPFunctional.h:351-357):
 * filters -> zero state, SineGen -> phase 0xC0000000 (MLDSPGens.h:375,379), etc.
 * A freshly created bank is in the default-constructed state of the reference objects
 * (all zero; ADSR segment = off) which differs from clear() only for SineGen. */
int mlgpu_bank_clear(mlgpu_bank* b);

/* `coeffs` member of the reference objects. Per-voice array of n_voices floats for one
 * coefficient, or one value broadcast to all voices. */
int mlgpu_bank_set_coeff(mlgpu_bank* b, int proc_idx, int coeff_idx, const float* h_per_voice);
int mlgpu_bank_set_coeff_uniform(mlgpu_bank* b, int proc_idx, int coeff_idx, float value);
/* Read one coefficient of every voice back (checkpointing; and what a freshly created bank holds are the `coeffs` of a
 * default-constructed reference object, e.g. Peak::peakHoldSamples{44100} as an int in slot 2, MLDSPFilters.h:574). */
int mlgpu_bank_get_coeff(mlgpu_bank* b, int proc_idx, int coeff_idx, float* h_per_voice);

/* Raw state access (checkpoint / resume; the reference's state is plain POD members). */
int mlgpu_bank_get_state(mlgpu_bank* b, int proc_idx, int state_idx, uint32_t* h_per_voice);
int mlgpu_bank_set_state(mlgpu_bank* b, int proc_idx, int state_idx, const uint32_t* h_per_voice);
int mlgpu_bank_set_state_uniform(mlgpu_bank* b, int proc_idx, int state_idx, uint32_t value);

/* Bank input when no signal is streamed: voice v's processor 0 sees DSPVector(h[v])
 * (the implicit float->DSPVector broadcast, MLDSPOps.h:157), e.g. a per-voice freq. */
int mlgpu_bank_set_input_const(mlgpu_bank* b, const float* h_per_voice);

/* Process n_vectors DSPVectors for every voice.
 *   d_in   input signal in `in_layout`, or NULL to use the per-voice input constant.
 *   d_out  output signal in `out_layout` (64*n_vectors*n_voices floats).
 * Replaces Bank::operator() (MLDSPFunctional.h:328-337) called n_vectors times.
 * Chains that match a fused kernel run as ONE launch; other chains run processor by
 * processor through the bank's HBM scratch signal (still on the GPU). */
int mlgpu_bank_process(mlgpu_bank* b, size_t n_vectors, const float* d_in, int in_layout,
                       float* d_out, int out_layout);
/* 1 if the chain maps to a single fused kernel, 0 if it runs processor by processor. */
int mlgpu_bank_is_fused(mlgpu_bank* b);
/* Name of the device kernel that dominates mlgpu_bank_process (for profile lookup). */
const char* mlgpu_bank_kernel_name(mlgpu_bank* b);

/* ------------------------------------------------------------------------- */
/* graphs ("procs")                                                           */
/*
 * A run-time defined DAG of processors (MLGPU_PROC_*) and stateless ops (MLGPU_OP_*) per voice,
 * with named nodes, after the naming convention of the reference's dynamic-graph stub
 * (source/procs/MLProcMultiply.cpp:12-18,29-32,46: named params / inputs / outputs, a process()
 * made of mldsp.h calls, registration by name). The reference has no executor; this
 * one compiles the whole graph into ONE fused gfx950 kernel with hiprtc: every edge is a
 * register, only graph inputs and outputs touch HBM.
 *
 * Build: add nodes in topological order (a node's inputs must already exist; there are no
 * feedback edges), mark outputs, compile, then set params/coeffs and process.
 *   input  streamed per-voice signal (one DeviceSignal per input, in the order added)
 *   param  per-voice constant, seen by its consumers as DSPVector(f) (MLDSPOps.h:157)
 *   const  one float for all voices
 *   proc   stateful processor: 1 signal input (generators: cyclesPerSample; NoiseGen: none), or one of the
 *          reference's other operator() forms: PulseGen(freq, width) MLDSPGens.h:390; Lopass(x, omega, k)
 *          MLDSPFilters.h:136 (coefficients made per sample, libm sinf restated on the device);
 *          LoShelf(x, a1,a2,a3,m1,m2) :304 and HiShelf(x, a1,a2,a3,m0,m1,m2) :385 (coefficient signals);
 *          Interpolator1 / LinearGlide: one control / param / const input
 *   vop    index-dependent generator (mlgpu_vop)
 *   op     stateless elementwise op with 1..3 inputs (masks travel as bit patterns)
 * add_* return the node id (>= 0) or -(mlgpu_status) on error.
 */
typedef struct mlgpu_graph mlgpu_graph;

/* index-dependent single-vector generators (MLDSPOps.h:962-990); start / end are floats in the reference, so
 * their inputs must be control / param / const nodes (or ops on those) */
typedef enum mlgpu_vop
{
  MLGPU_VOP_COLUMN_INDEX = 0,       /* columnIndex()                         :965   no inputs */
  MLGPU_VOP_RANGE_OPEN = 1,         /* rangeOpen(start, end)                 :970-974 */
  MLGPU_VOP_RANGE_CLOSED = 2,       /* rangeClosed(start, end)               :978-982 */
  MLGPU_VOP_INTERPOLATE_LINEAR = 3, /* interpolateDSPVectorLinear(start, end) :986-990; one row of
                                     * interpolateCoeffsLinear (MLDSPFilters.h:34-44) */
  MLGPU_VOP_TABLE = 4               /* a constant DSPVector; made by mlgpu_graph_add_const_vector only */
} mlgpu_vop;

int mlgpu_graph_create(mlgpu_engine* e, size_t n_voices, mlgpu_graph** out);
int mlgpu_graph_destroy(mlgpu_graph* g);
int mlgpu_graph_add_input(mlgpu_graph* g, const char* name);
int mlgpu_graph_add_param(mlgpu_graph* g, const char* name);
/* control: streamed, ONE float per DSPVector per voice ([T][V] floats) — what the reference passes as a `float`
 * argument per process() call (LinearGlide::operator()(float), lerp(a, b, float m) MLDSPOps.h:753, ...).
 * Audio-rate consumers see it as DSPVector(f). */
int mlgpu_graph_add_control(mlgpu_graph* g, const char* name);
int mlgpu_graph_add_vop(mlgpu_graph* g, int vop, const int* input_nodes, int n_inputs, const char* name);
int mlgpu_graph_add_const(mlgpu_graph* g, float value);
/* Live constants. By default a const node is a literal of the generated kernel. After mlgpu_graph_set_live_constants(g, 1)
 * (before compile) const nodes are read from a small device table instead — a scalar load per constant and launch — and
 * mlgpu_graph_set_const changes one between launches (stream-ordered, no recompilation): what a host-side parameter that the
 * reference's process function turns into `DSPVector(value)` or a `float` argument needs in order to stay adjustable.
 * mlgpu_graph_update_constants_from(g, other): `other` is a second graph built by the same code with other numbers (it need
 * not be compiled); when both have the same nodes and wiring, g's constants take other's values, otherwise
 * MLGPU_ERR_UNSUPPORTED and nothing changes. Same results as a graph compiled with those literals. */
int mlgpu_graph_set_live_constants(mlgpu_graph* g, int on);
int mlgpu_graph_set_const(mlgpu_graph* g, int const_node, float value);
int mlgpu_graph_update_constants_from(mlgpu_graph* g, mlgpu_graph* other);
/* A constant DSPVector: the same 64 floats for every voice and every vector — DSPVector(const float*), DSPVector(float (*)(int)),
 * DSPVector(std::array<float, 64>) (MLDSPOps.h:140-161) evaluated on the host when the graph is built (windows, index maps,
 * tables). Bit patterns are kept as given. Inside a rate region the vector is the region function's own DSPVector. */
int mlgpu_graph_add_const_vector(mlgpu_graph* g, const float* values /* [64] */, const char* name);
/* Delay-line memory of a delay node (before compile): IntegerDelay::setMaxDelayInSamples (MLDSPFilters.h:823-831),
 * i.e. rings of 2^bitsToContain(floor(d) + 64) floats per voice (PitchbendableDelay: two of them). */
int mlgpu_graph_set_max_delay(mlgpu_graph* g, int proc_node, float max_delay_in_samples);
/* Layout of the delay rings (before compile). 0 (default): [sample][voice] — one coalesced row per time step, best when
 * neighbouring voices use the same delay times. 1: [256-voice block][sample / 8][voice][8] — a voice moves its samples in
 * 32-byte sectors through LDS windows, so no bandwidth is wasted when delay times differ from voice to voice (DESIGN.md
 * §3.6); costs 8 KiB of LDS per ring per workgroup, at most 20 rings per graph. 2 (round 5): [256-voice block][sample / 16][voice][16]
 * moved as whole 64-byte pieces by neighbouring lanes on a wave-uniform clock, read windows in LDS, requests a period ahead, a
 * voice's own recent samples kept there for short delay times: the algorithmic traffic and not a byte more (0.98 x measured
 * against 1.13 x for layout 1) at 0.72-0.74 of the HBM peak against 0.56-0.59 (profiles/r05_ring_layouts.txt) - the one to use
 * for per-voice delay times where it applies: at most 4 rings per graph (40 KiB of LDS per ring and workgroup), and a voice count
 * that is a multiple of 64 if the graph sums voices in groups or reads event rows inside its kernel (other graphs: any count - the
 * spare lanes of the bank's last wavefront run its last voice again). 4 (round 6, "sector trips", for graphs with MANY rings): layout
 * 1's memory, the sample loop in trips of 8 samples on the write clock - a ring's eight written samples leave as ONE 32-byte sector,
 * its eight reads come from two neighbouring sectors (the lower one held in LDS since the trip before, the upper one asked for a
 * whole trip ahead) through a barrel shifter, every ring's loads of a trip issued together; a voice's last 16 samples stay in LDS
 * for delay times under 16. A PitchbendableDelay keeps ONE ring (the reference feeds both of its FractionalDelays the same input,
 * source/DSP/MLDSPFilters.h:1096-1105) and makes one read for both while their delay times agree. 8 KiB of LDS per ring + 16 KiB per
 * delay node and workgroup; one wavefront per SIMD (its windows are registers). 4 x Allpass<PitchbendableDelay> with per-voice
 * delay times: 0.65-0.67 of the HBM peak at 0.85 x the algorithmic traffic against 0.26 at 1.9 x in layout 1
 * (profiles/r06_ring_layouts.txt). Not for a delay line inside a rate region. 3: "voices have their own delay times, take the best
 * form": layout 2 for one or two rings, layout 4 for more (where its LDS fits and no delay line sits in a rate region), else layout
 * 1, else - more rings than LDS - the default rows; decided by graph_compile. With one delay time for all voices layout 2 is as
 * fast as the default rows (0.85 / 0.85 of the HBM peak on the strings bank; layout 1: 0.72). Same results in every layout for
 * delay times within the node's maximum (graph_set_max_delay). mlgpu_graph_delay_layout: the layout in effect (0 / 1 / 2 / 4; 3
 * before compile), negative: a status. */
int mlgpu_graph_set_delay_layout(mlgpu_graph* g, int windowed);
int mlgpu_graph_delay_layout(mlgpu_graph* g);
/* One-vector feedback: a DSPVector the reference keeps from one process call to the next (Allpass::vy1
 * MLDSPFilters.h:1115, FDN::mDelayInputVectors :1168, FeedbackDelayFunction::vy1 MLDSPFunctional.h:276, or a user's
 * own state member). add_feedback returns a node whose value at sample n is what set_feedback's `value_node` had at
 * sample n of the PREVIOUS DSPVector (zeros at first); set_feedback may name any node, also one added later. */
int mlgpu_graph_add_feedback(mlgpu_graph* g, const char* name);
/* Rate regions: `Upsample2xFunction<IN_ROWS>()(fn, vx)` / `Downsample2xFunction<IN_ROWS>()(fn, vx)` with fn written out as
 * graph nodes. begin_region resamples the n_inputs outer nodes with one HalfBandFilter each and returns, in
 * region_inputs[], the nodes that carry them inside the region; every audio-rate node added until end_region belongs to fn
 * and runs at the region's rate, on the same processor objects for both halves as in the reference (fn is one stateful
 * function called twice per DSPVector, :132-133, resp. once per two, :186). end_region resamples `result` back and returns
 * the node of the outer graph (negative status on failure). Inside a region: processors, ops, routing and index generators
 * on the region's inputs and on per-voice floats (params / consts) of the outer graph, and feedback nodes whose source is in
 * the same region (fn's own state kept from one of ITS DSPVectors to the next), and further regions (a 4x oversampled
 * function is an UPSAMPLE_2X region inside another; up to three deep, their inputs being nodes of the enclosing region); no
 * streamed inputs, controls or vector-rate processors. A DOWNSAMPLE_2X region adds the reference's one DSPVector of
 * delay and pairs DSPVectors (2k, 2k + 1) counted from the last mlgpu_graph_clear. */
int mlgpu_graph_begin_region(mlgpu_graph* g, int region /* mlgpu_region */, const int* inputs, int n_inputs, int* region_inputs);
int mlgpu_graph_end_region(mlgpu_graph* g, int result, const char* name);
int mlgpu_graph_set_feedback(mlgpu_graph* g, int feedback_node, int value_node);
/* routing nodes (MLDSPRouting.h): input_nodes[0] is the selector.
 *   MLGPU_ROUTE_MULTIPLEX / _LINEAR      input_nodes[1..n] the candidates (n <= 8); `index` ignored
 *   MLGPU_ROUTE_DEMULTIPLEX / _LINEAR    input_nodes[1] the signal; this node is output `index` of `n_outputs` */
typedef enum mlgpu_route
{
  MLGPU_ROUTE_MULTIPLEX = 0,
  MLGPU_ROUTE_MULTIPLEX_LINEAR = 1,
  MLGPU_ROUTE_DEMULTIPLEX = 2,
  MLGPU_ROUTE_DEMULTIPLEX_LINEAR = 3
} mlgpu_route;
int mlgpu_graph_add_route(mlgpu_graph* g, int route, const int* input_nodes, int n_inputs, int index, int n_outputs,
                          const char* name);
int mlgpu_graph_add_proc(mlgpu_graph* g, int proc_kind, const int* input_nodes, int n_inputs, const char* name);
int mlgpu_graph_add_op(mlgpu_graph* g, int op, const int* input_nodes, int n_inputs, const char* name);
/*
 * Processors and ops BY NAME. The reference's dynamic-proc stub registers a class under a string ("multiply",
 * source/procs/MLProcMultiply.cpp:44-47) and names its params, inputs and outputs with strings (:12-18, :29-32). Every node
 * kind a graph can hold has such an entry here - the reference's function / class name in lower_snake_case ("multiply",
 * "saw_gen", "lopass", "exp2_approx", "pitchbendable_delay", ...), its input names, its coefficient names in slot order, its
 * output name - so a host can describe a whole patch with strings (a preset, a UI) without touching an enum.
 */
enum { MLGPU_REGISTRY_OP = 0, MLGPU_REGISTRY_PROC = 1, MLGPU_REGISTRY_VOP = 2 };
typedef struct mlgpu_registry_entry
{
  const char* name;        /* static storage */
  int node_type;           /* MLGPU_REGISTRY_* */
  int kind;                /* MLGPU_OP_* / MLGPU_PROC_* / MLGPU_VOP_* */
  int n_inputs;            /* all inputs, the optional ones of the modulated forms (Lopass: omega, k) last */
  int n_required_inputs;
  int n_params;            /* coefficient slots (the `coeffs` member of the reference object) */
  const char* output_name; /* "out", or "mask" for the compare ops */
} mlgpu_registry_entry;
int mlgpu_registry_count(void);
int mlgpu_registry_get(int index, mlgpu_registry_entry* out);
int mlgpu_registry_lookup(const char* name, mlgpu_registry_entry* out);       /* MLGPU_ERR_RANGE when unknown; out may be NULL */
int mlgpu_registry_input_name(const char* name, int index, char* buf, size_t buf_len);
int mlgpu_registry_param_name(const char* name, int index, char* buf, size_t buf_len);
int mlgpu_registry_param_index(const char* name, const char* param);          /* coefficient slot, or < 0 */
/* add the node `node_name` of registered kind `proc_name`; its inputs are the nodes called input_node_names[i] (in the
 * entry's input order). Returns the node id or -status. */
int mlgpu_graph_add_named(mlgpu_graph* g, const char* proc_name, const char* node_name, const char* const* input_node_names, int n_inputs);
/* per-voice tables by name: a processor's coefficient (node name + coefficient name), a param node (its name).
 * h_per_voice == NULL: `uniform` for every voice. */
int mlgpu_graph_set_named_coeff(mlgpu_graph* g, const char* node_name, const char* coeff_name, const float* h_per_voice, float uniform);
int mlgpu_graph_set_param_by_name(mlgpu_graph* g, const char* param_name, const float* h_per_voice, float uniform);

/* Limits per graph: 32 streamed inputs, 8 controls, 8 outputs (MLGPU_GRAPH_MAX_INPUTS / _CONTROLS / _OUTPUTS in mlgpu_device_args.hpp; MLGPU_ERR_UNSUPPORTED beyond). */
int mlgpu_graph_add_output(mlgpu_graph* g, int node);
int mlgpu_graph_node(mlgpu_graph* g, const char* name); /* id of the node called `name`, or < 0 */
/* (re)name a node - e.g. a constant, which mlgpu_graph_add_const creates nameless - so that it can be wired by name */
int mlgpu_graph_set_node_name(mlgpu_graph* g, int node, const char* name);
/* MLGPU_PROC_* / MLGPU_OP_* / MLGPU_VOP_* of a processor / op / generator node (< 0 for inputs, params, constants ...) */
int mlgpu_graph_node_kind(mlgpu_graph* g, int node);
/* how many times `node` is referenced: as an input of other nodes, as a feedback source, as a graph output */
int mlgpu_graph_node_use_count(mlgpu_graph* g, int node);
int mlgpu_graph_num_nodes(mlgpu_graph* g);
/* Voices evaluated by one wavefront lane (before compile): 0 = automatic (default), 1, or 2. Two voices per lane interleave
 * two independent dependency chains, which helps arithmetic-bound graphs; results are identical. */
int mlgpu_graph_set_voices_per_lane(mlgpu_graph* g, int n);
/* Online tuning (before compile). The generated kernel exists in up to four forms - one or two voices per lane, one or two
 * quads per trip of the sample loop - that compute the same bits from the same state, and which one is fastest depends on
 * the graph (registers, code size against the instruction cache) and on the regime (burst or sustained). With tuning on, the
 * first process calls of at least 4 Mi voice-samples take turns through the forms (three calls each, timed with events: these
 * calls wait for the device; a form is compiled when its turn comes), then the fastest stays. Results never differ.
 * mlgpu_graph_tuning: 1 when settled (or tuning is off), 0 while still measuring; reports the form in use. */
/* Device memory a compiled graph owns: coefficients, state, per-voice constants and delay rings (a reverb with 24 rings of
 * up to 16 384 samples is 0.55 MB per voice: 36 GB for 65 536 voices - sized for the 288 GB of an MI355X). */
size_t mlgpu_graph_device_bytes(mlgpu_graph* g);
int mlgpu_graph_set_autotune(mlgpu_graph* g, int on);
int mlgpu_graph_tuning(mlgpu_graph* g, int* voices_per_lane, int* quads_per_trip);
/* Workgroups (256 voices each) of this graph's kernel that one CU holds at once - registers, LDS and wavefront slots together
 * (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor); negative: a status. A bank of more than 256 x this x the CU count voices runs
 * its last workgroups after the others. */
int mlgpu_graph_workgroups_per_cu(mlgpu_graph* g);
/* Generate + compile (hiprtc, gfx950) + load the fused kernel; allocate state/coeffs/params.
 * A COLD compile - a graph whose generated source is in neither the process's memory cache nor the disk cache
 * ($MLGPU_CACHE_DIR, default ~/.cache/mlgpu) - runs hiprtc for 2 to 5 seconds (measured: the 16-node synth voice 1.9 s, the
 * survey's 22-node patch 4.9 s); warm it is milliseconds. NEVER call this from an audio callback for a patch the user has just
 * edited: use the two calls below, or compile ahead of time. */
int mlgpu_graph_compile(mlgpu_graph* g);
/* The same work on a thread of the library's: returns at once. Until mlgpu_graph_compile_poll has answered something other than
 * MLGPU_ERR_BUSY the graph belongs to that job - every other call on it (build calls, mlgpu_graph_compile, mlgpu_graph_process,
 * mlgpu_graph_emit, setters) returns MLGPU_ERR_BUSY and changes nothing; mlgpu_graph_destroy waits for the job. A host keeps
 * processing its OLD graph block after block, polls once per block, and swaps when the new one is ready (tests/test_gpu_graph.py::
 * test_patch_swap_with_async_compile: no block over its period while a cold compile runs). The reference's equivalent is building
 * a new processor list off the audio thread (naming convention: source/procs/MLProcMultiply.cpp:12-47).
 * On a graph created WITHOUT an engine (mlgpu_graph_create(NULL, ...)) the job is an ahead-of-time compile: it fills the memory
 * and disk caches for this description and the graph stays a description (poll then answers MLGPU_OK once the code exists). */
int mlgpu_graph_compile_async(mlgpu_graph* g);
/* MLGPU_ERR_BUSY: still compiling. MLGPU_OK: compiled (state, coefficients and parameters allocated on the calling thread, on the
 * engine's stream: microseconds) - the graph is ready for setters and mlgpu_graph_process. Anything else: the compile failed with
 * that status (mlgpu_graph_last_error says why) and the graph can be destroyed. */
int mlgpu_graph_compile_poll(mlgpu_graph* g);
/* The generated HIP source (valid after compile; for inspection). */
const char* mlgpu_graph_source(mlgpu_graph* g);
/* Offline code generation: place the rings, generate the kernel source (mlgpu_graph_source) and compile it to a gfx950 code
 * object with hiprtc, without touching a device. `e` of mlgpu_graph_create may be NULL for a graph used this way
 * (mlgpu_graph_compile then fails with MLGPU_ERR_INVALID). *code stays valid until the graph is destroyed. */
int mlgpu_graph_emit(mlgpu_graph* g, const void** code, size_t* code_size);
const char* mlgpu_graph_last_error(mlgpu_graph* g);  /* the text of this graph's last failure (also for graphs without an engine: the hiprtc log) */
int mlgpu_graph_clear(mlgpu_graph* g); /* T::clear() on every processor node */
int mlgpu_graph_clear_proc(mlgpu_graph* g, int proc_node); /* T::clear() on one processor node */
int mlgpu_graph_set_state_uniform(mlgpu_graph* g, int proc_node, int state_idx, uint32_t value);
int mlgpu_graph_set_param(mlgpu_graph* g, int param_node, const float* h_per_voice);
int mlgpu_graph_set_param_uniform(mlgpu_graph* g, int param_node, float value);
int mlgpu_graph_num_coeffs(mlgpu_graph* g, int proc_node);
int mlgpu_graph_num_state(mlgpu_graph* g, int proc_node);
int mlgpu_graph_set_coeff(mlgpu_graph* g, int proc_node, int coeff_idx, const float* h_per_voice);
int mlgpu_graph_set_coeff_uniform(mlgpu_graph* g, int proc_node, int coeff_idx, float value);
int mlgpu_graph_get_state(mlgpu_graph* g, int proc_node, int state_idx, uint32_t* h_per_voice);
int mlgpu_graph_set_state(mlgpu_graph* g, int proc_node, int state_idx, const uint32_t* h_per_voice);
/* n_vectors DSPVectors for every voice. d_inputs[i] / d_outputs[o]: device signals in the order the
 * inputs / outputs were added, all in `in_layout` / `out_layout`. */
int mlgpu_graph_process(mlgpu_graph* g, size_t n_vectors, const float* const* d_inputs, int in_layout,
                        float* const* d_outputs, int out_layout);
/* Give streamed input `input_index` (order of mlgpu_graph_add_input) its own layout, overriding the in_layout of the
 * process calls — e.g. MLGPU_LAYOUT_BROADCAST for a host audio channel next to per-voice inputs. -1 removes it. */
int mlgpu_graph_set_input_layout(mlgpu_graph* g, int input_index, int layout);
/* Same, for graphs with control inputs: d_controls[i] is the i-th control's [n_vectors][n_voices] floats. */
int mlgpu_graph_process_ctl(mlgpu_graph* g, size_t n_vectors, const float* const* d_inputs, int in_layout,
                            const float* const* d_controls, float* const* d_outputs, int out_layout);

/* Chains without an ahead-of-time kernel are fused with hiprtc when their bank is created
 * (default on). With jit off they run processor by processor through HBM scratch signals. */
int mlgpu_engine_set_jit(mlgpu_engine* e, int enabled);
/* Device-free check that the run-time code generator's output compiles for gfx950 (a chain and a
 * graph); used by the CPU build check. Writes the compiler log (if any) to `log`. */
int mlgpu_jit_selftest(char* log, size_t log_len);
/* Run-time fused kernels are cached twice: per process, and on disk (MLGPU_CACHE_DIR, default $XDG_CACHE_HOME/mlgpu or
 * ~/.cache/mlgpu; "off" disables it), keyed by a hash of the generated source, the compile options, the embedded device
 * headers and the hiprtc version - so only the first process to build a given graph pays hiprtc (0.3-2 s per kernel).
 * Counters since the library was loaded (any pointer may be NULL): kernels compiled by hiprtc and the seconds that took,
 * code objects read from the disk cache and the seconds that took, requests served from memory. */
int mlgpu_jit_stats(uint64_t* compiles, uint64_t* disk_hits, uint64_t* memory_hits, double* compile_seconds, double* disk_load_seconds);
/* Installations WITHOUT the compiler (round 6). libmlgpu.so looks libhiprtc.so up at run time and does not link it: where it is
 * missing, every ahead-of-time kernel works as before and a generated one (a graph, a chain without an ahead-of-time form, a strict-SVF
 * bank, a bank's summing form) works when its code is there - from the disk cache or from a BUNDLE: mlgpu_jit_cache_export writes every
 * generated kernel this process holds (sources as keys + gfx950 code objects) into `buffer` (`needed`: its size; buffer NULL: only
 * that), mlgpu_jit_cache_import takes such a bundle (MLGPU_ERR_UNSUPPORTED: made by another build of the device code;
 * MLGPU_ERR_INVALID: not a bundle). Make it where hiprtc is installed - run the patches once, or compile them ahead of time on
 * graphs without an engine (mlgpu_graph_compile_async) -, ship it with the plug-in, import it at load. A kernel that is nowhere to be
 * found fails with MLGPU_ERR_UNSUPPORTED and says so. mlgpu_jit_compiler_available: 1 / 0. (MLGPU_HIPRTC=off in the environment:
 * behave as if the compiler were absent.) */
int mlgpu_jit_cache_export(void* buffer, size_t capacity, size_t* needed);
int mlgpu_jit_cache_import(const void* buffer, size_t size, size_t* kernels);
int mlgpu_jit_cache_clear_memory(void);
int mlgpu_jit_compiler_available(void);

/* ------------------------------------------------------------------------- */
/* performance events -> per-voice control signals                            */
/*
 * mlgpu_events is EventsToSignals (source/app/MLEventsToSignals.{h,cpp}) for n_instruments independent instruments of
 * `polyphony` voices each: events in, 8 control signals per voice out — pitch, gate, vox, z, x, y, mod, elapsed time
 * (VoiceOutputSignals, MLEventsToSignals.h:15-26) — for V = n_instruments * polyphony voices (instrument-major), ready to
 * be the inputs of a bank or graph. Voice allocation, key states, unison, sustain and the MIDI / MPE channel rules run on
 * the host; glides, drift and the sample-accurate note timing run on the device.
 * Usage per host block, as AudioContext / SignalProcessBuffer do (MLSignalProcessBuffer.cpp:47-89): add_event()s with
 * times in frames from the start of the block, process(n_vectors, start_offset) once or several times, clear_events().
 * Not reproduced: controller 120 ("all sound off"), which clears the event buffer the reference is iterating (:749-755).
 */
typedef enum mlgpu_event_type /* ml::EventType, source/app/MLEvent.h:13-26 */
{
  MLGPU_EVENT_NULL = 0,
  MLGPU_EVENT_NOTE_ON = 1,
  MLGPU_EVENT_NOTE_RETRIG = 2,
  MLGPU_EVENT_NOTE_SUSTAIN = 3,
  MLGPU_EVENT_NOTE_OFF = 4,
  MLGPU_EVENT_SUSTAIN_PEDAL = 5,
  MLGPU_EVENT_CONTROLLER = 6,
  MLGPU_EVENT_PITCH_BEND = 7,
  MLGPU_EVENT_NOTE_PRESSURE = 8,
  MLGPU_EVENT_CHANNEL_PRESSURE = 9,
  MLGPU_EVENT_PROGRAM_CHANGE = 10
} mlgpu_event_type;
typedef struct mlgpu_event /* ml::Event, MLEvent.h:31-53 */
{
  uint8_t type;        /* mlgpu_event_type */
  uint8_t channel;
  uint16_t source_idx; /* key or controller number */
  int32_t time;        /* onset in frames from the start of the current host block */
  float value1;        /* note: pitch; controller / bend / pressure: value */
  float value2;        /* note: velocity */
} mlgpu_event;
typedef struct mlgpu_events mlgpu_events;
int mlgpu_events_create(mlgpu_engine* e, size_t n_instruments, int polyphony /* setPolyphony, 1..16 */, mlgpu_events** out);
int mlgpu_events_destroy(mlgpu_events* ev);   /* (while recorded sequences of the engine live, the memory is released with the last of them) */
int mlgpu_events_clear(mlgpu_events* ev);                                   /* clear(), :330-340 */
int mlgpu_events_set_sample_rate(mlgpu_events* ev, double sr);
int mlgpu_events_set_protocol(mlgpu_events* ev, int mpe);                   /* setProtocol("MIDI" / "MPE"); clears */
int mlgpu_events_set_unison(mlgpu_events* ev, int on);
int mlgpu_events_set_mod_cc(mlgpu_events* ev, int cc);
int mlgpu_events_set_pitch_bend_semitones(mlgpu_events* ev, float f);
int mlgpu_events_set_mpe_pitch_bend_semitones(mlgpu_events* ev, float f);
int mlgpu_events_set_pitch_glide_seconds(mlgpu_events* ev, float f);
int mlgpu_events_set_drift_amount(mlgpu_events* ev, float f);
/* Rows a consumer never reads need not be made: bit r of `mask` = row r of mlgpu_events_process' outputs (pitch, gate, vox,
 * z, x, y, mod, elapsed time); default all. Rows outside the mask are not computed at all (their glides do not advance:
 * choose the mask once, before the first process call) and their output pointers must be NULL. */
int mlgpu_events_set_wanted_rows(mlgpu_events* ev, unsigned mask);
size_t mlgpu_events_num_voices(mlgpu_events* ev);
int mlgpu_events_newest_voice(mlgpu_events* ev, size_t instrument);        /* getNewestVoice() */
int mlgpu_events_add_event(mlgpu_events* ev, size_t instrument, const mlgpu_event* e);   /* addEvent, :367-372 */
/* the same for a whole block at once: events[i] goes to instrument instruments[i] (nothing is added if one is out of range) */
int mlgpu_events_add_events(mlgpu_events* ev, const uint32_t* instruments, const mlgpu_event* events, size_t n);
int mlgpu_events_clear_events(mlgpu_events* ev);                            /* clearEvents */
/* processVector (:376-466) for n_vectors consecutive DSPVectors; start_offset = frame of the first one in the host block.
 * d_outputs[8]: device signals of V voices x n_vectors vectors in `layout` (any may be NULL = not wanted), in the order
 * pitch, gate, vox, z, x, y, mod, elapsed time.
 * Arguments are checked before any event is consumed; a later failure (allocation, upload, launch) has already advanced the
 * host-side voice allocator and the object must be reset (mlgpu_events_set_protocol) before it is used again.
 * BEHAVIOURAL DIFFERENCE: controller 120 ("all sound off") is ignored. The reference calls clear() there
 * (MLEventsToSignals.cpp:749-755), which empties the event vector the enclosing loop is iterating over (undefined
 * behaviour, :404-423) - there is no defined result to match. Send controller 123 (all notes off, reproduced) or reset
 * the object with mlgpu_events_set_protocol to silence an instrument. */
int mlgpu_events_process(mlgpu_events* ev, size_t n_vectors, int start_offset, float* const* d_outputs, int layout);

/* Smoothed controller signals: what a process function reads through AudioContext::getInputController(n)
 * (source/app/MLAudioContext.cpp:129 -> EventsToSignals::SmoothedController, MLEventsToSignals.h:170-180, .cpp:264-281, 431-436).
 * One signal per INSTRUMENT per watched controller number (0..128; 128 is what MIDI channel pressure writes too, :650),
 * made on the device from the controller events of each DSPVector: output = LinearGlide(20 ms)(value of the last controller
 * event so far); zeros while the instrument has not seen any event (:386). watch_controllers reserves the signals for launches
 * of up to max_vectors DSPVectors (it allocates: call it at set-up time; n = 0 releases them); a smoother that starts being
 * watched later than the first event starts settled on its controller's current value; calling it again with the same numbers
 * only changes max_vectors (the smoothers go on, the signal pointers change). From then on every
 * mlgpu_events_process / mlgpu_graph_process_events call also advances the controller signals by the same DSPVectors.
 * mlgpu_events_controller_signal(ev, slot): the device signal of numbers[slot] for the DSPVectors of the LAST such call - QUAD
 * layout over nInstruments "voices" ([16 n_vectors][nInstruments][4] floats), valid until the next one; the pointer itself
 * does not change between calls. Feed it to a voice graph as an input with mlgpu_graph_set_input_group(g, input, polyphony). */
#define MLGPU_EVENTS_MAX_WATCHED_CONTROLLERS 32
int mlgpu_events_watch_controllers(mlgpu_events* ev, const int* numbers, int n, size_t max_vectors);
const float* mlgpu_events_controller_signal(mlgpu_events* ev, int slot);

/* ---- AudioContext::ProcessTime (source/app/MLAudioContext.h:27-57, MLAudioContext.cpp:16-104) for n independent contexts:
 * the quarter-note phasor a process function reads through ctx->getBeatPhase(). set_time_and_rate = AudioContext::updateTime ->
 * ProcessTime::setTimeAndRate (host arithmetic in double, as in the reference; the report refers to the start of the next
 * process call; NaN / infinite positions and tempi are ignored, :20-26); process = ProcessTime::processVector for n_vectors
 * DSPVectors on the device (omega_ float, slope double, wrapped above 1; -1 while the host is not playing);
 * beat_phase = the signal of the last process call, QUAD layout over the n contexts ([16 n_vectors][n][4] floats; the pointer
 * does not change): feed it to a voice graph as an input with mlgpu_graph_set_input_group(g, input, voices per context).
 * index MLGPU_TRANSPORT_ALL: every context (one host application, many instruments). */
typedef struct mlgpu_transport mlgpu_transport;
#define MLGPU_TRANSPORT_ALL ((size_t)-1)
int mlgpu_transport_create(mlgpu_engine* e, size_t n, size_t max_vectors, mlgpu_transport** out);
int mlgpu_transport_destroy(mlgpu_transport* t);
int mlgpu_transport_reserve(mlgpu_transport* t, size_t max_vectors);  /* another launch length; the phasors go on, the signal pointer changes */
int mlgpu_transport_clear(mlgpu_transport* t, size_t index);  /* ProcessTime::clear */
int mlgpu_transport_set_time_and_rate(mlgpu_transport* t, size_t index, double ppq_pos, double bpm, int is_playing, double sample_rate);
int mlgpu_transport_process(mlgpu_transport* t, size_t n_vectors);
const float* mlgpu_transport_beat_phase(mlgpu_transport* t);
uint64_t mlgpu_transport_samples_since_start(mlgpu_transport* t, size_t index);  /* ProcessTime::samplesSinceStart */
double mlgpu_transport_bpm(mlgpu_transport* t, size_t index);

/* A streamed input shared by groups of adjacent voices: input `input_index` (the order of mlgpu_graph_add_input) is a signal of
 * voices / group rows, voice v reads row v / group - one controller or transport signal per instrument of `group` voices
 * (group = voices: one row for the whole bank). voices must be a multiple of group. Before compile. */
int mlgpu_graph_set_input_group(mlgpu_graph* g, int input_index, int group);

/* Synth::processVector's per-instrument voice sum (source/app/MLSynth.h:43-57) inside the voice kernel: output `output_index`
 * becomes a signal of voices / group channels, channel c = ((0 + voice[c group]) + voice[c group + 1]) + ... in that order (the
 * bits of mlgpu_mixdown_groups), written once; groups of 2, 4, 8 or 16 adjacent voices. Before compile. */
int mlgpu_graph_set_output_group_sum(mlgpu_graph* g, int output_index, int group);
/* Output `output_index` becomes ONE channel, the mixdown of ALL voices - mlgpu_mixdown's order and bits (no gains), its first stage (the
 * tree over each wavefront's 64 voices) inside the voice kernel, so that the voices' signal of that output is never written (the graph
 * counterpart of mlgpu_bank_process_mixdown). mlgpu_graph_process then takes 64 * n_vectors floats for it whatever the output layout;
 * mlgpu_graph_reserve_mixdown(g, max vectors) at setup sizes the engine's mixdown scratch for it (process calls never allocate).
 * Any voice count - unless the graph also sums groups or reads event rows in its kernel: then a multiple of 64.
 * on = 2: the SHARD form (a graph that is one engine's part of a larger bank): the output takes mlgpu_mixdown_shard_rows(voices) rows of
 * 64 * n_vectors floats for mlgpu_mixdown_finish, as mlgpu_bank_process_mixdown_shard; voices a multiple of 64. */
int mlgpu_graph_set_output_mixdown(mlgpu_graph* g, int output_index, int on);
int mlgpu_graph_reserve_mixdown(mlgpu_graph* g, size_t max_vectors);

/*
 * The pitch and gate rows as SOURCE NODES of a voice graph: a Synth's voices read voice.outputs.row(kPitch / kGate) straight
 * from EventsToSignals (source/app/MLSynth.h:43-57, MLEventsToSignals.h:15-26); here the graph kernel makes those two rows
 * itself from the same per-voice state as mlgpu_events_process - the rows never exist in memory. Since round 5 (behaviour revision 3)
 * mlgpu_graph_process_events is two launches: a light control kernel runs everything that happens once per DSPVector or once per
 * note event (record walk, bend and pitch glides, the drift random walk: source/app/MLEventsToSignals.cpp:75-262) and writes a
 * 16-byte control record per voice and DSPVector - plus 64 frames of pitch and gate for the few vectors that hold a note event or
 * a portamento -, and the voice kernel expands the records and walks the drift glide. MIDI protocol; instruments x polyphony must equal the graph's voices; voice v of the graph is voice v of the object
 * (instrument v / polyphony, voice v % polyphony). Per block: mlgpu_events_add_event(s), then mlgpu_graph_process_events with
 * the block's frame offset (what mlgpu_events_process takes), then mlgpu_events_clear_events as usual. A block may be processed
 * by either call: both leave the state the other expects (rows a call does not compute keep their glides where they are, as
 * with mlgpu_events_set_wanted_rows).
 */
int mlgpu_graph_add_event_row(mlgpu_graph* g, int row /* 0 pitch, 1 gate */, const char* name); /* before compile; node id */
int mlgpu_graph_bind_events(mlgpu_graph* g, mlgpu_events* ev);
/* Device memory of the two-launch form, per voice and DSPVector of the longest block: a 16-byte control record + 2 x 256 bytes of side
 * signal (only the flagged vectors' frames are ever written or read, but the buffers are addressed [vector][voice]) = 528 bytes:
 * 2.2 GB for 262 144 voices x 16 DSPVectors (mlgpu_events_graph_reserve_bytes says it for an object). Reserve it at SETUP with
 * mlgpu_events_reserve_for_graph(ev, longest block in DSPVectors): from then on mlgpu_graph_process_events never allocates and
 * answers MLGPU_ERR_RANGE to a longer block (before the router consumes its events). Without a reserve the first block - and any
 * longer one later - waits for the stream and allocates inside the process call (fine for tools and tests, not for an audio thread). */
int mlgpu_events_reserve_for_graph(mlgpu_events* ev, size_t max_vectors);
size_t mlgpu_events_graph_reserve_bytes(mlgpu_events* ev, size_t max_vectors);
int mlgpu_graph_process_events(mlgpu_graph* g, size_t n_vectors, int start_offset, const float* const* d_inputs, int in_layout,
                               const float* const* d_controls, float* const* d_outputs, int out_layout);

/* ------------------------------------------------------------------------- */
/* published signals                                                         */
/*
 * SignalProcessor::PublishedSignal (source/app/MLSignalProcessor.h:26-105, MLSignalProcessor.cpp:9-38): a decimated,
 * frame-major copy of `channels` signals of a few voices for code outside the DSP calculation (displays). write() is
 * `storePublishedSignal(name, DSPVectorArray<channels>, 64, voice)` for voices first_voice .. first_voice + n_voices - 1 in
 * rotation, for n_vectors DSPVectors: every (1 << octaves_down)-th frame is kept (no filtering) and the ring receives, per
 * DSPVector, [voice][kept frame][channel]. The ring is the reference's DSPBuffer of max_frames * channels * max_voices
 * floats; read / read_latest / peek_latest are its PublishedSignal::read / readLatest / peekLatest (frames of `channels`
 * floats; the return value counts floats, like the reference's). write() waits for the device (one gather kernel and one
 * D2H copy through pinned memory per call).
 */
typedef struct mlgpu_published_signal mlgpu_published_signal;
int mlgpu_published_signal_create(mlgpu_engine* e, int max_frames, int max_voices, int channels, int octaves_down, mlgpu_published_signal** out);
int mlgpu_published_signal_destroy(mlgpu_published_signal* p);
int mlgpu_published_signal_write(mlgpu_published_signal* p, size_t n_vectors, const float* const* d_channels, int layout, size_t n_voices_total,
                                 size_t first_voice, size_t n_voices);
size_t mlgpu_published_signal_num_channels(mlgpu_published_signal* p);      /* getNumChannels */
size_t mlgpu_published_signal_read_available(mlgpu_published_signal* p);    /* getReadAvailable (floats) */
size_t mlgpu_published_signal_available_frames(mlgpu_published_signal* p);  /* getAvailableFrames */
size_t mlgpu_published_signal_read(mlgpu_published_signal* p, float* dest, size_t frames_requested);
size_t mlgpu_published_signal_read_latest(mlgpu_published_signal* p, float* dest, size_t frames_requested);
void mlgpu_published_signal_peek_latest(mlgpu_published_signal* p, float* dest, size_t frames_requested);

/* ------------------------------------------------------------------------- */
/* sample-rate conversion by powers of two                                   */
/*
 * Downsampler / Upsampler (MLDSPFilters.h:1316-1473): a cascade of HalfBandFilters (:1245-1310), one per octave, for every
 * voice. `up` = 0: every 2^octaves input DSPVectors give one output vector (Downsampler::write returns true, read());
 * n_vectors_in must be a multiple of 2^octaves. `up` = 1: every input vector gives 2^octaves output vectors
 * (Upsampler::write, then 2^octaves read()s). State: 9 floats per octave per voice, SoA [octave*9 + i][V], i in
 * {apa0.x1, apa0.y1, apa1.x1, apa1.y1, apb0.x1, apb0.y1, apb1.x1, apb1.y1, b1}. octaves 0..6 (0 copies).
 */
typedef struct mlgpu_resampler mlgpu_resampler;
int mlgpu_resampler_create(mlgpu_engine* e, size_t n_voices, int octaves, int up, mlgpu_resampler** out);
int mlgpu_resampler_destroy(mlgpu_resampler* r);
int mlgpu_resampler_clear(mlgpu_resampler* r);
int mlgpu_resampler_get_state(mlgpu_resampler* r, float* h_state);
int mlgpu_resampler_set_state(mlgpu_resampler* r, const float* h_state);
int mlgpu_resampler_process(mlgpu_resampler* r, size_t n_vectors_in, const float* d_in, int in_layout, float* d_out, int out_layout);

/* ------------------------------------------------------------------------- */
/* host ring and block adaptor                                               */
/*
 * mlgpu_dspbuffer — the reference's DSPBuffer (source/DSP/MLDSPBuffer.h:20-384): a single-producer /
 * single-consumer float ring on the HOST. resize() allocates the power of two >= max(n, 64) and returns it (0 on
 * failure); a write into a full ring overwrites the oldest samples; read_vector returns 1 and 64 samples, or 0 and
 * 64 zeros when fewer than 64 samples wait (DSPVector read(), :253-277). One reader thread and one writer thread
 * may use a ring concurrently without a lock, exactly as in the reference.
 */
typedef struct mlgpu_dspbuffer mlgpu_dspbuffer;
mlgpu_dspbuffer* mlgpu_dspbuffer_create(void);
void mlgpu_dspbuffer_destroy(mlgpu_dspbuffer* b);
size_t mlgpu_dspbuffer_resize(mlgpu_dspbuffer* b, int size_in_samples);                     /* :104-133 */
size_t mlgpu_dspbuffer_size(mlgpu_dspbuffer* b);
void mlgpu_dspbuffer_clear(mlgpu_dspbuffer* b);                                              /* :96-100 */
size_t mlgpu_dspbuffer_read_available(mlgpu_dspbuffer* b);                                   /* :136-141 */
size_t mlgpu_dspbuffer_write_available(mlgpu_dspbuffer* b);                                  /* :144 */
void mlgpu_dspbuffer_write(mlgpu_dspbuffer* b, const float* src, size_t samples);            /* :147-168 */
size_t mlgpu_dspbuffer_read(mlgpu_dspbuffer* b, float* dst, size_t samples);                 /* :207-224 */
int mlgpu_dspbuffer_read_vector(mlgpu_dspbuffer* b, float* dst64);                           /* :253-277 */
void mlgpu_dspbuffer_discard(mlgpu_dspbuffer* b, size_t samples);                            /* :280-286 */
void mlgpu_dspbuffer_write_with_overlap_add(mlgpu_dspbuffer* b, const float* src, size_t samples, size_t overlap); /* :289-320 */
void mlgpu_dspbuffer_read_with_overlap(mlgpu_dspbuffer* b, float* dst, size_t samples, size_t overlap);           /* :323-338 */
void mlgpu_dspbuffer_peek_most_recent(mlgpu_dspbuffer* b, float* dst, size_t samples);       /* :342-383 */

/*
 * mlgpu_process_buffer — the reference's SignalProcessBuffer (source/app/MLSignalProcessBuffer.h:22-41, .cpp:36-90):
 * the host asks for blocks of arbitrary size nFrames <= max_frames; inputs and outputs are buffered in rings and the
 * processing happens in whole DSPVectors. Where the reference calls `SignalProcessFn(AudioContext*, void*)` once per
 * 64-frame vector (MLSignalProcessBuffer.h:18), this calls `fn` ONCE per block with all K vectors the block needs,
 * already in HBM: d_inputs[c] / d_outputs[c] are single-voice signals of K*64 floats on the engine's device (use
 * MLGPU_LAYOUT_BROADCAST to feed one to every voice of a bank or graph, mlgpu_mixdown to sum voices into one).
 * `fn` enqueues work on the engine's stream and returns an mlgpu_status; it must not block.
 * inputs[c] may be NULL (channel not connected); outputs[c] may be NULL (not wanted).
 */
typedef struct mlgpu_process_buffer mlgpu_process_buffer;
typedef int (*mlgpu_process_vectors_fn)(void* user, size_t n_vectors, const float* const* d_inputs, float* const* d_outputs);
int mlgpu_process_buffer_create(mlgpu_engine* e, size_t n_inputs, size_t n_outputs, size_t max_frames, mlgpu_process_buffer** out);
int mlgpu_process_buffer_destroy(mlgpu_process_buffer* p);
/* Two staging sets, double buffered. on = 0 (default): a process call returns the audio of its own block - H2D copy, the
 * callback's launches, D2H copy and a wait, i.e. the reference's behaviour exactly. on != 0: the call returns immediately
 * with what earlier calls computed while its own block travels and runs behind the host's back (the copy of block k + 1
 * overlaps the kernels of block k; the host thread never waits for a kernel it has just launched). The output stream is
 * the synchronous one delayed by mlgpu_process_buffer_latency_frames() frames (the largest block, rounded up to whole
 * DSPVectors, plus one), silence first - a fixed latency a host compensates like any plug-in's. (Its rings are larger than
 * the reference's, so a host that overdrives those - leftovers of an out-of-phase block plus a full-size one - loses no
 * samples here, where the reference and the synchronous mode overwrite their oldest ones.) Switch before the first process
 * call (switching later restarts the output rings). */
int mlgpu_process_buffer_set_pipelined(mlgpu_process_buffer* p, int on);
size_t mlgpu_process_buffer_latency_frames(mlgpu_process_buffer* p);
int mlgpu_process_buffer_process(mlgpu_process_buffer* p, const float* const* inputs, float* const* outputs, int n_frames,
                                 mlgpu_process_vectors_fn fn, void* user);

/* Sum the voices of a signal into ONE single-voice signal of 64*n_vectors floats (what a Synth does with
 * `outputs += voice` in its voice loop, source/app/MLSynth.h:43-57), optionally scaled by per-voice gains
 * (d_gains may be NULL). Summation order (deterministic): pairwise tree inside each group of 64 consecutive voices; then
 * the group sums left to right, 64 consecutive ones at a time, and so their results, until one is left - for up to 4096 voices
 * that is "the groups left to right". (Round 4. Before, ALL groups were added in one left-to-right chain: the same bits up to
 * 4096 voices, other last bits above - and 480 us of serial additions per 64-frame block at 2^20 voices.)
 * The partial sums of the first stage need scratch memory: reserve it once at setup with mlgpu_mixdown_reserve for the
 * largest (voices, vectors) a process call will pass - mlgpu_mixdown itself never allocates and returns
 * MLGPU_ERR_INVALID when the reservation is too small. */
int mlgpu_mixdown_reserve(mlgpu_engine* e, size_t max_voices, size_t max_vectors);
int mlgpu_mixdown(mlgpu_engine* e, const float* d_signal, int layout, size_t n_voices, size_t n_vectors, const float* d_gains,
                  float* d_out);
/* mlgpu_bank_process followed by mlgpu_mixdown of its output (d_gains: per-voice gains or NULL, as there), in one call and without the voices' signals ever reaching
 * memory: the voice kernel adds up the 64 voices of each wavefront itself - the first stage's tree, so d_out has the SAME BITS as
 * the two calls give - and the later stages follow. For a 2^20-voice bank paced at 48 kHz that halves the device time of a block
 * (DESIGN.md 3.7). For the fused voice chains (SawGen -> Bandpass -> Gain and the others mlgpu_bank_kernel_name shows as one
 * kernel), any voice count; MLGPU_ERR_UNSUPPORTED for other banks until mlgpu_bank_prepare_mixdown has been called for them. Needs the same
 * mlgpu_mixdown_reserve as mlgpu_mixdown. State and coefficients as after mlgpu_bank_process. */
int mlgpu_bank_process_mixdown(mlgpu_bank* bank, size_t n_vectors, const float* d_in, int in_layout, const float* d_gains, float* d_out);
/* The same block across SEVERAL ENGINES - the GPUs of a node, each with a contiguous range of the voices (the reference's
 * `outputs += voice` loop runs over all voices of a Synth, source/app/MLSynth.h:43-57). The order above is a tree: 64 voices pairwise,
 * then 64 consecutive sums at a time left to right. A shard whose voice count is a multiple of 64^L (L >= 1, the largest such:
 * mlgpu_mixdown_shard_level) holds whole sub-trees up to level L, so it hands over its mlgpu_mixdown_shard_rows() = voices / 64^L
 * level-L sums - d_rows [rows][64 * n_vectors] floats - and the host finishes the tree over all shards' rows in voice order with
 * mlgpu_mixdown_finish (plain float additions, no device; 8 x 262 144 voices: 8 rows; 8 x 2^20: 32 rows). N shards of V / N voices
 * then give the SAME BITS one engine gives for V voices (tests/cpp/multi_engine_test.cpp: 2 and 8 engines; tests/test_gpu_parity.py).
 * Voices a multiple of 64 per shard; otherwise MLGPU_ERR_INVALID (level 0). mlgpu_mixdown_shard: the same for a signal in memory.
 * mlgpu_mixdown_finish: h_rows [n_rows][64 * n_vectors]; flush_denormals = what the engines run with; h_scratch: only for more than 64
 * rows, (ceil(n / 64) + ceil(n / 4096)) * 64 * n_vectors floats. No allocation in any of them; the scratch is mlgpu_mixdown_reserve's. */
int mlgpu_mixdown_shard_level(size_t voices_per_shard);
size_t mlgpu_mixdown_shard_rows(size_t voices_per_shard);
int mlgpu_bank_process_mixdown_shard(mlgpu_bank* bank, size_t n_vectors, const float* d_in, int in_layout, const float* d_gains, float* d_rows);
int mlgpu_mixdown_shard(mlgpu_engine* e, const float* d_signal, int layout, size_t n_voices, size_t n_vectors, const float* d_gains, float* d_rows);
int mlgpu_mixdown_finish(const float* h_rows, size_t n_rows, size_t n_vectors, int flush_denormals, float* h_out, float* h_scratch);
/* Setup, for a bank whose chain is not one of those: generate the summing form of its kernel (hiprtc; cached on disk like every generated
 * kernel), after which mlgpu_bank_process_mixdown serves it too - any chain of bank processors, fused or not. MLGPU_OK at once where the
 * form is there already. */
int mlgpu_bank_prepare_mixdown(mlgpu_bank* bank);

/* Sum every `group_size` consecutive voices into one: out voice g = ((0 + v[g*P]) + v[g*P+1]) + ... in voice order — the
 * `outputs[c] += ...` accumulation of Synth::processVector (source/app/MLSynth.h:43-57), bit for bit. The result is a signal
 * of n_groups voices (e.g. one per instrument of an mlgpu_events bank). */
int mlgpu_mixdown_groups(mlgpu_engine* e, const float* d_signal, int layout, size_t n_groups, size_t group_size, size_t n_vectors,
                         float* d_out, int out_layout);

/* ------------------------------------------------------------------------- */
/* coefficient makers — host-side, glibc libm, formulas of the reference     */
/* (kept on the host so device code never has to match libm: SURVEY App. A 11) */

void mlgpu_lopass_make_coeffs(float omega, float k, float out3[3]);    /* MLDSPFilters.h:85-95 */
void mlgpu_hipass_make_coeffs(float omega, float k, float out4[4]);    /* :168-178 */
void mlgpu_bandpass_make_coeffs(float omega, float k, float out3[3]);  /* :212-222 */
void mlgpu_loshelf_make_coeffs(float omega, float k, float A, float out5[5]); /* :270-281 */
void mlgpu_hishelf_make_coeffs(float omega, float k, float A, float out6[6]); /* :350-362 */
void mlgpu_bell_make_coeffs(float omega, float k, float A, float out4[4]);    /* :415-425 */
void mlgpu_onepole_make_coeffs(float omega, float out2[2]);            /* :458-462 */
float mlgpu_dcblocker_make_coeffs(float omega);                        /* :498 */
void mlgpu_adsr_calc_coeffs(float a, float d, float s, float r, float sr, float out4[4]); /* :679-686 */
float mlgpu_db_to_gain(float dB);                                      /* :30 */
float mlgpu_allpass1_make_coeffs(float d);                              /* Allpass1::makeCoeffs, :938-943 */
/* FractionalDelay::setDelayInSamples (:991-1007) -> {delayInt as int32 bits, allpass coefficient} = state words 3, 4 */
void mlgpu_fractional_delay_make_state(float delay_in_samples, float out2[2]);
/* LinearGlide::setGlideTimeInSamples (MLDSPGens.h:444-449) -> C{vectorsPerGlide:i32 bits, dyPerVector} */
void mlgpu_linear_glide_make_coeffs(float glide_time_in_samples, float out2[2]);
/* SampleAccurateLinearGlide::setGlideTimeInSamples (:527-532) -> C{samplesPerGlide:i32 bits, dyPerSample} */
void mlgpu_sample_accurate_linear_glide_make_coeffs(float glide_time_in_samples, float out2[2]);

/* ------------------------------------------------------------------------- */
/* window tables — host-side: makeWindow(pDest, size, dspwindows::<shape>),   */
/* source/DSP/MLDSPUtils.h:22-47, for the overlap-add use of mlgpu_dspbuffer  */
/* (Tests/dspBufferTest.cpp "overlap"). Same floats as the reference's.       */
enum mlgpu_window
{
  MLGPU_WINDOW_RECTANGLE = 0,
  MLGPU_WINDOW_TRIANGLE = 1,
  MLGPU_WINDOW_RAISED_COSINE = 2,
  MLGPU_WINDOW_HAMMING = 3,
  MLGPU_WINDOW_BLACKMAN = 4,
  MLGPU_WINDOW_FLAT_TOP = 5
};
int mlgpu_make_window(float* dest, size_t size, int shape);

#ifdef __cplusplus
}
#endif
#endif /* MLGPU_H */
