// tools/instbench.hip — issue cost of individual gfx950 VALU instructions (inline asm, 4 independent chains per wave, 4
// waves per SIMD, sustained >= 40 ms after a warm-up). The question it answers: which instructions of a voice kernel cost
// a full-rate slot and which a half-rate one — on gfx950 plain FP32 add/mul/fma are HALF rate (16 lanes per clock per
// SIMD) and only their packed forms reach the chip's FP32 peak, while integer, logic, compare, select and conversion
// instructions are full rate.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

#define BODY4(INS) INS(a0) INS(a1) INS(a2) INS(a3)
#define REP16(X) X X X X X X X X X X X X X X X X

#define DEFK(NAME, INS, TYPE, INIT)                                                     \
  __global__ __launch_bounds__(256) void NAME(float* out, int iters, float fa, float fb) \
  {                                                                                     \
    TYPE a0 = INIT(0), a1 = INIT(1), a2 = INIT(2), a3 = INIT(3);                        \
    float b = fb, c = fa;                                                               \
    (void)b; (void)c;                                                                   \
    for (int it = 0; it < iters; ++it) { REP16(BODY4(INS)) }                            \
    float t = SUM(a0) + SUM(a1) + SUM(a2) + SUM(a3);                                    \
    if (t == 1234.5f) out[0] = t;                                                       \
  }

#define FINIT(i) (fa + threadIdx.x + i)
#define PINIT(i) f2{fa + threadIdx.x + i, fb + i}
#define SUM(x) sum1(x)
__device__ inline float sum1(float x) { return x; }
__device__ inline float sum1(f2 x) { return x.x + x.y; }

#define I_FMA(r) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r) : "v"(c), "v"(b));
#define I_MUL(r) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r) : "v"(c));
#define I_ADD(r) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r) : "v"(b));
#define I_ADDU(r) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r) : "v"(b));
#define I_LSHR(r) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(r));
#define I_AND(r) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r) : "v"(b));
#define I_CVT(r) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(r));
#define I_CVTI(r) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(r));
#define I_MOV(r) asm volatile("v_mov_b32 %0, %1" : "+v"(r) : "v"(b));
#define I_CMP(r) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(r), "v"(b) : "vcc");
#define I_CND(r) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r) : "v"(b) : "vcc");
#define I_CMPCND(r) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(r) : "v"(b), "v"(c) : "vcc");
#define I_MAX(r) asm volatile("v_max_f32 %0, %0, %1" : "+v"(r) : "v"(b));
#define I_RCP(r) asm volatile("v_rcp_f32 %0, %0" : "+v"(r));
#define I_FRACT(r) asm volatile("v_fract_f32 %0, %0" : "+v"(r));
#define I_RNDNE(r) asm volatile("v_rndne_f32 %0, %0" : "+v"(r));
#define I_PKFMA(r) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(r) : "v"(r));
#define I_PKMUL(r) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(r));
#define I_PKADD(r) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(r));
#define I_PKMOV(r) asm volatile("v_pk_mov_b32 %0, %0, %0" : "+v"(r));
#define I_MUL64(r) asm volatile("v_mul_f64 %0, %0, %0" : "+v"(r));

#define I_MUL_LIT(r) asm volatile("v_mul_f32 %0, 0x30000000, %0" : "+v"(r));
#define I_MUL_SGPR(r) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(r) : "s"(c));
#define I_FMA_SGPR(r) asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(r) : "s"(c), "v"(b));
#define I_FMA_INL(r) asm volatile("v_fma_f32 %0, %0, 2.0, -1.0" : "+v"(r));
#define I_ADD_INL(r) asm volatile("v_add_f32 %0, -1.0, %0" : "+v"(r));
#define I_ADD3(r) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(r) : "v"(b));
#define I_CVTU(r) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(r));
#define I_CMP64(r) asm volatile("v_cmp_lt_f32 s[20:21], %0, %1" : : "v"(r), "v"(b) : "s20", "s21");
#define I_CND64(r) asm volatile("v_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(r) : "v"(b));
#define I_LDEXP(r) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(r) : "v"(b));
#define I_BFI(r) asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(r) : "v"(b), "v"(c));
#define I_SUB(r) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(r) : "v"(b));
#define I_FMAC(r) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(r) : "v"(b), "v"(c));
#define I_MAD_U32(r) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(r) : "v"(b), "v"(c));
#define I_SAND(r) asm volatile("s_and_b64 s[20:21], s[20:21], s[28:29]\n s_xor_b64 s[22:23], s[22:23], s[28:29]\n s_or_b64 s[24:25], s[24:25], s[28:29]\n s_andn2_b64 s[26:27], s[26:27], s[28:29]" : : : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "scc");
#define I_MIX11(r) asm volatile("v_fma_f32 %0, %0, %1, %2\n s_and_b64 s[20:21], s[20:21], s[28:29]" : "+v"(r) : "v"(c), "v"(b) : "s20", "s21", "scc");
#define I_MIX13(r) asm volatile("v_fma_f32 %0, %0, %1, %2\n s_and_b64 s[20:21], s[20:21], s[28:29]\n s_xor_b64 s[22:23], s[22:23], s[28:29]\n s_or_b64 s[24:25], s[24:25], s[28:29]" : "+v"(r) : "v"(c), "v"(b) : "s20", "s21", "s22", "s23", "s24", "s25", "scc");
#define I_BRANCH(r) asm volatile("v_fma_f32 %0, %0, %1, %2\n s_cmp_eq_u64 exec, 0\n s_cbranch_scc1 8" : "+v"(r) : "v"(c), "v"(b) : "scc");
// double precision (the modulated Lopass' libm sinf runs in double): 64-bit register pairs
#define DINIT(i) ((double)(fa + threadIdx.x + i))
__device__ inline float sum1(double x) { return (float)x; }
#define I_DFMA(r) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(r) : "v"(dc), "v"(db));
#define I_DMUL(r) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(r) : "v"(dc));
#define I_DADD(r) asm volatile("v_add_f64 %0, %0, %1" : "+v"(r) : "v"(db));
#define I_DCVT(r) asm volatile("v_cvt_f32_f64 %1, %0\n v_cvt_f64_f32 %0, %1" : "+v"(r), "+v"(tmp));
#define DEFKD(NAME, INS)                                                                 \
  __global__ __launch_bounds__(256) void NAME(float* out, int iters, float fa, float fb) \
  {                                                                                     \
    double a0 = DINIT(0), a1 = DINIT(1), a2 = DINIT(2), a3 = DINIT(3);                  \
    double db = fb, dc = fa;                                                            \
    float tmp = fa;                                                                     \
    (void)db; (void)dc; (void)tmp;                                                      \
    for (int it = 0; it < iters; ++it) { REP16(BODY4(INS)) }                            \
    float t = SUM(a0) + SUM(a1) + SUM(a2) + SUM(a3) + tmp;                              \
    if (t == 1234.5f) out[0] = t;                                                       \
  }
DEFKD(k_dfma, I_DFMA)
DEFKD(k_dmul, I_DMUL)
DEFKD(k_dadd, I_DADD)
DEFKD(k_dcvt, I_DCVT)
DEFK(k_sand, I_SAND, float, FINIT)
DEFK(k_mix11, I_MIX11, float, FINIT)
DEFK(k_mix13, I_MIX13, float, FINIT)
DEFK(k_branch, I_BRANCH, float, FINIT)
DEFK(k_mul_lit, I_MUL_LIT, float, FINIT)
DEFK(k_mul_sgpr, I_MUL_SGPR, float, FINIT)
DEFK(k_fma_sgpr, I_FMA_SGPR, float, FINIT)
DEFK(k_fma_inl, I_FMA_INL, float, FINIT)
DEFK(k_add_inl, I_ADD_INL, float, FINIT)
DEFK(k_add3, I_ADD3, float, FINIT)
DEFK(k_cvtu, I_CVTU, float, FINIT)
DEFK(k_cmp64, I_CMP64, float, FINIT)
DEFK(k_cnd64, I_CND64, float, FINIT)
DEFK(k_ldexp, I_LDEXP, float, FINIT)
DEFK(k_bfi, I_BFI, float, FINIT)
DEFK(k_sub, I_SUB, float, FINIT)
DEFK(k_fmac, I_FMAC, float, FINIT)
DEFK(k_fma, I_FMA, float, FINIT)
DEFK(k_mul, I_MUL, float, FINIT)
DEFK(k_add, I_ADD, float, FINIT)
DEFK(k_addu, I_ADDU, float, FINIT)
DEFK(k_lshr, I_LSHR, float, FINIT)
DEFK(k_and, I_AND, float, FINIT)
DEFK(k_cvt, I_CVT, float, FINIT)
DEFK(k_cvti, I_CVTI, float, FINIT)
DEFK(k_mov, I_MOV, float, FINIT)
DEFK(k_cmp, I_CMP, float, FINIT)
DEFK(k_cnd, I_CND, float, FINIT)
DEFK(k_cmpcnd, I_CMPCND, float, FINIT)
DEFK(k_max, I_MAX, float, FINIT)
DEFK(k_rcp, I_RCP, float, FINIT)
DEFK(k_fract, I_FRACT, float, FINIT)
DEFK(k_rndne, I_RNDNE, float, FINIT)
DEFK(k_pkfma, I_PKFMA, f2, PINIT)
DEFK(k_pkmul, I_PKMUL, f2, PINIT)
DEFK(k_pkadd, I_PKADD, f2, PINIT)
DEFK(k_pkmov, I_PKMOV, f2, PINIT)

typedef void (*K)(float*, int, float, float);
static void run(const char* name, K k, float* out, int instPerBody)
{
  const int iters = 2000, wavesPerSimd = 4, blocks = 256 * wavesPerSimd;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0000001f, 0.5f); CK(hipDeviceSynchronize());
  int reps = 1; float ms = 0;
  for (;;)
  {
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0000001f, 0.5f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms >= 40.f) break;
    reps *= 2;
  }
  const double per = ms / reps;
  const double inst = (double)iters * 16 * 4 * instPerBody;          // per wave
  const double ns = per * 1e6 / inst / wavesPerSimd;                  // per instruction per SIMD
  printf("%-22s %.3f ns/instr/SIMD = %.2f cycles at 2.4 GHz  (%.1f T lane-instr/s)\n", name, ns, ns * 2.4, 1024.0 * 64 / ns / 1e3);
}
int main()
{
  float* out; CK(hipMalloc(&out, 64));
  for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k_fma, dim3(1024), dim3(256), 0, 0, out, 4000, 1.0000001f, 0.5f);
  CK(hipDeviceSynchronize());
  run("v_fma_f32", k_fma, out, 1); run("v_mul_f32", k_mul, out, 1); run("v_add_f32", k_add, out, 1);
  run("v_pk_fma_f32", k_pkfma, out, 1); run("v_pk_mul_f32", k_pkmul, out, 1); run("v_pk_add_f32", k_pkadd, out, 1); run("v_pk_mov_b32", k_pkmov, out, 1);
  run("v_add_u32", k_addu, out, 1); run("v_lshrrev_b32", k_lshr, out, 1); run("v_and_b32", k_and, out, 1);
  run("v_cvt_f32_i32", k_cvt, out, 1); run("v_cvt_i32_f32", k_cvti, out, 1); run("v_mov_b32", k_mov, out, 1);
  run("v_cmp_lt_f32", k_cmp, out, 1); run("v_cndmask_b32", k_cnd, out, 1); run("v_cmp + v_cndmask", k_cmpcnd, out, 2);
  run("v_max_f32", k_max, out, 1); run("v_fract_f32", k_fract, out, 1); run("v_rndne_f32", k_rndne, out, 1); run("v_rcp_f32", k_rcp, out, 1);
  run("v_mul_f32 literal", k_mul_lit, out, 1); run("v_mul_f32 sgpr src", k_mul_sgpr, out, 1); run("v_fma_f32 sgpr src", k_fma_sgpr, out, 1);
  run("v_fma_f32 x,2.0,-1.0", k_fma_inl, out, 1); run("v_add_f32 -1.0,x", k_add_inl, out, 1); run("v_sub_f32", k_sub, out, 1); run("v_fmac_f32", k_fmac, out, 1);
  run("v_add3_u32", k_add3, out, 1); run("v_cvt_f32_u32", k_cvtu, out, 1); run("v_cmp_lt_f32 -> sgpr", k_cmp64, out, 1);
  run("v_cndmask_b32 sgpr mask", k_cnd64, out, 1); run("v_ldexp_f32", k_ldexp, out, 1); run("v_bfi_b32", k_bfi, out, 1);
  // the scalar pipe: 64-bit mask logic alone, interleaved with VALU (different waves can issue the two in the same cycle), and a
  // never-taken scalar compare-and-branch after every VALU instruction (exec is never 0)
  run("s_and/xor/or/andn2_b64", k_sand, out, 4); run("v_fma + 1 s_and_b64 (per pair)", k_mix11, out, 1); run("v_fma + 3 SALU (per group)", k_mix13, out, 1);
  run("v_fma + s_cmp + s_cbranch", k_branch, out, 1);
  run("v_fma_f64", k_dfma, out, 1); run("v_mul_f64", k_dmul, out, 1); run("v_add_f64", k_dadd, out, 1); run("v_cvt_f32_f64 + v_cvt_f64_f32", k_dcvt, out, 2);
  return 0;
}
