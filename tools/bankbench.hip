// tools/bankbench.hip — does the VGPR bank of a packed-FP32 instruction's operands change its issue cost on gfx950?
// Inline asm with explicit registers (the compiler's allocator is what we are second-guessing), no data dependencies
// between consecutive instructions unless the case says so, no memory traffic. Developer tool (round 3, config 4).
//   hipcc --offload-arch=gfx950 -O2 tools/bankbench.hip -o tools/bin/bankbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

#define CLOB "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", \
             "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39"
#define R4(X) X X X X
#define R16(X) R4(X) R4(X) R4(X) R4(X)

// BODY = 4 instructions (one "group"); a loop trip is 16 groups = 64 instructions
#define DEFK(NAME, BODY)                                                                 \
  __global__ __launch_bounds__(256) void NAME(float* out, int iters, float fa)           \
  {                                                                                      \
    asm volatile("v_mov_b32 v0, %0\n v_mov_b32 v1, %0\n v_mov_b32 v2, %0\n v_mov_b32 v3, %0\n"     \
                 "v_mov_b32 v4, %0\n v_mov_b32 v5, %0\n v_mov_b32 v6, %0\n v_mov_b32 v7, %0\n"     \
                 "v_mov_b32 v8, %0\n v_mov_b32 v9, %0\n v_mov_b32 v10, %0\n v_mov_b32 v11, %0\n"   \
                 "v_mov_b32 v12, %0\n v_mov_b32 v13, %0\n v_mov_b32 v14, %0\n v_mov_b32 v15, %0\n" \
                 "v_mov_b32 v16, %0\n v_mov_b32 v17, %0\n v_mov_b32 v18, %0\n v_mov_b32 v19, %0\n" \
                 "v_mov_b32 v20, %0\n v_mov_b32 v21, %0\n v_mov_b32 v22, %0\n v_mov_b32 v23, %0\n" \
                 "v_mov_b32 v24, %0\n v_mov_b32 v25, %0\n v_mov_b32 v26, %0\n v_mov_b32 v27, %0\n" \
                 "v_mov_b32 v28, %0\n v_mov_b32 v29, %0\n v_mov_b32 v30, %0\n v_mov_b32 v31, %0\n" \
                 "v_mov_b32 v32, %0\n v_mov_b32 v33, %0\n v_mov_b32 v34, %0\n v_mov_b32 v35, %0\n" \
                 "v_mov_b32 v36, %0\n v_mov_b32 v37, %0\n v_mov_b32 v38, %0\n v_mov_b32 v39, %0\n" \
                 : : "v"(fa) : CLOB);                                                    \
    for (int it = 0; it < iters; ++it) asm volatile(R16(BODY) : : : CLOB);               \
    float t;                                                                             \
    asm volatile("v_add_f32 %0, v20, v21\n v_add_f32 %0, %0, v22\n v_add_f32 %0, %0, v23\n v_add_f32 %0, %0, v24\n v_add_f32 %0, %0, v26\n v_add_f32 %0, %0, v28\n v_add_f32 %0, %0, v30" : "=v"(t) : : CLOB); \
    if (t == 1234.5f) out[0] = t;                                                        \
  }

// --- packed multiply, two VGPR-pair sources; destinations rotate over 4 pairs, never read
DEFK(k_pkmul_same, "v_pk_mul_f32 v[20:21], v[4:5], v[8:9]\n v_pk_mul_f32 v[24:25], v[0:1], v[12:13]\n v_pk_mul_f32 v[28:29], v[4:5], v[16:17]\n v_pk_mul_f32 v[32:33], v[8:9], v[12:13]\n")
DEFK(k_pkmul_diff, "v_pk_mul_f32 v[20:21], v[4:5], v[10:11]\n v_pk_mul_f32 v[24:25], v[0:1], v[14:15]\n v_pk_mul_f32 v[28:29], v[4:5], v[18:19]\n v_pk_mul_f32 v[32:33], v[8:9], v[14:15]\n")
DEFK(k_pkmul_diff_dst, "v_pk_mul_f32 v[22:23], v[4:5], v[10:11]\n v_pk_mul_f32 v[26:27], v[0:1], v[14:15]\n v_pk_mul_f32 v[30:31], v[4:5], v[18:19]\n v_pk_mul_f32 v[34:35], v[8:9], v[14:15]\n")
// --- packed fma, three VGPR-pair sources
DEFK(k_pkfma_same3, "v_pk_fma_f32 v[20:21], v[4:5], v[8:9], v[12:13]\n v_pk_fma_f32 v[24:25], v[0:1], v[12:13], v[16:17]\n v_pk_fma_f32 v[28:29], v[4:5], v[16:17], v[0:1]\n v_pk_fma_f32 v[32:33], v[8:9], v[12:13], v[4:5]\n")
DEFK(k_pkfma_2banks, "v_pk_fma_f32 v[20:21], v[4:5], v[10:11], v[12:13]\n v_pk_fma_f32 v[24:25], v[0:1], v[14:15], v[16:17]\n v_pk_fma_f32 v[28:29], v[4:5], v[18:19], v[0:1]\n v_pk_fma_f32 v[32:33], v[8:9], v[14:15], v[4:5]\n")
DEFK(k_pkfma_inline2, "v_pk_fma_f32 v[20:21], v[4:5], 2.0, v[10:11] op_sel_hi:[1,0,1]\n v_pk_fma_f32 v[24:25], v[0:1], 2.0, v[14:15] op_sel_hi:[1,0,1]\n v_pk_fma_f32 v[28:29], v[4:5], 2.0, v[18:19] op_sel_hi:[1,0,1]\n v_pk_fma_f32 v[32:33], v[8:9], 2.0, v[14:15] op_sel_hi:[1,0,1]\n")
DEFK(k_pkfma_inline2_same, "v_pk_fma_f32 v[20:21], v[4:5], 2.0, v[8:9] op_sel_hi:[1,0,1]\n v_pk_fma_f32 v[24:25], v[0:1], 2.0, v[12:13] op_sel_hi:[1,0,1]\n v_pk_fma_f32 v[28:29], v[4:5], 2.0, v[16:17] op_sel_hi:[1,0,1]\n v_pk_fma_f32 v[32:33], v[8:9], 2.0, v[12:13] op_sel_hi:[1,0,1]\n")
// --- packed add with the sub's modifiers
DEFK(k_pkadd_diff, "v_pk_add_f32 v[20:21], v[4:5], v[10:11]\n v_pk_add_f32 v[24:25], v[0:1], v[14:15]\n v_pk_add_f32 v[28:29], v[4:5], v[18:19]\n v_pk_add_f32 v[32:33], v[8:9], v[14:15]\n")
DEFK(k_pkadd_opsel, "v_pk_add_f32 v[20:21], v[4:5], v[10:11] op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 v[24:25], v[0:1], v[14:15] op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n"
                    "v_pk_add_f32 v[28:29], v[4:5], v[18:19] op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 v[32:33], v[8:9], v[14:15] op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n")
// --- plain ops
DEFK(k_mul_same, "v_mul_f32 v20, v4, v8\n v_mul_f32 v21, v0, v12\n v_mul_f32 v22, v4, v16\n v_mul_f32 v23, v8, v12\n")
DEFK(k_mul_diff, "v_mul_f32 v20, v4, v9\n v_mul_f32 v21, v0, v13\n v_mul_f32 v22, v4, v17\n v_mul_f32 v23, v8, v13\n")
DEFK(k_fma_same3, "v_fma_f32 v20, v4, v8, v12\n v_fma_f32 v21, v0, v12, v16\n v_fma_f32 v22, v4, v16, v0\n v_fma_f32 v23, v8, v12, v4\n")
DEFK(k_fma_diff3, "v_fma_f32 v20, v4, v9, v14\n v_fma_f32 v21, v0, v13, v18\n v_fma_f32 v22, v4, v17, v2\n v_fma_f32 v23, v8, v13, v6\n")
// two of the three sources in one bank (v4 / v8 / v12 / v16 / v0: bank 0; v9 / v13 / v17: bank 1; v14 / v18 / v2 / v6: bank 2)
DEFK(k_fma_ab_same, "v_fma_f32 v20, v4, v8, v14\n v_fma_f32 v21, v0, v12, v18\n v_fma_f32 v22, v4, v16, v2\n v_fma_f32 v23, v8, v12, v6\n")
DEFK(k_fma_ac_same, "v_fma_f32 v20, v4, v9, v8\n v_fma_f32 v21, v0, v13, v12\n v_fma_f32 v22, v4, v17, v16\n v_fma_f32 v23, v8, v13, v12\n")
DEFK(k_fma_bc_same, "v_fma_f32 v20, v9, v4, v8\n v_fma_f32 v21, v13, v0, v12\n v_fma_f32 v22, v17, v4, v16\n v_fma_f32 v23, v13, v8, v12\n")
// v_fmac (dst is the addend) with the two factors in one bank / in two
DEFK(k_fmac_same, "v_fmac_f32 v20, v4, v8\n v_fmac_f32 v21, v0, v12\n v_fmac_f32 v22, v4, v16\n v_fmac_f32 v23, v8, v12\n")
DEFK(k_fmac_all_same, "v_fmac_f32 v20, v4, v8\n v_fmac_f32 v24, v0, v12\n v_fmac_f32 v28, v4, v16\n v_fmac_f32 v32, v8, v12\n")
DEFK(k_fmac_diff, "v_fmac_f32 v20, v5, v10\n v_fmac_f32 v21, v2, v15\n v_fmac_f32 v22, v7, v16\n v_fmac_f32 v23, v9, v14\n")
// fma with an inline constant (2.0) and the other two in one bank / in two
DEFK(k_fma_const_same, "v_fma_f32 v20, v4, 2.0, v8\n v_fma_f32 v21, v0, 2.0, v12\n v_fma_f32 v22, v4, 2.0, v16\n v_fma_f32 v23, v8, 2.0, v12\n")
// --- dependent chains of packed ops: 4 / 2 / 1 independent chains
DEFK(k_pk_dep4, "v_pk_mul_f32 v[20:21], v[20:21], v[10:11]\n v_pk_mul_f32 v[24:25], v[24:25], v[14:15]\n v_pk_mul_f32 v[28:29], v[28:29], v[18:19]\n v_pk_mul_f32 v[32:33], v[32:33], v[14:15]\n")
DEFK(k_pk_dep2, "v_pk_mul_f32 v[20:21], v[20:21], v[10:11]\n v_pk_mul_f32 v[24:25], v[24:25], v[14:15]\n v_pk_mul_f32 v[20:21], v[20:21], v[18:19]\n v_pk_mul_f32 v[24:25], v[24:25], v[14:15]\n")
DEFK(k_pk_dep1, "v_pk_mul_f32 v[20:21], v[20:21], v[10:11]\n v_pk_mul_f32 v[20:21], v[20:21], v[14:15]\n v_pk_mul_f32 v[20:21], v[20:21], v[18:19]\n v_pk_mul_f32 v[20:21], v[20:21], v[14:15]\n")
// --- DPP moves
DEFK(k_dpp_shr, "v_mov_b32_dpp v20, v4 row_shr:4 row_mask:0xf bank_mask:0xe\n v_mov_b32_dpp v21, v0 row_shr:4 row_mask:0xf bank_mask:0xe\n v_mov_b32_dpp v22, v8 row_shr:4 row_mask:0xf bank_mask:0xe\n v_mov_b32_dpp v23, v12 row_shr:4 row_mask:0xf bank_mask:0xe\n")
// --- a mix like the cascade tick: sub, 4 mul, 2 add, add, 2 fma on 4 independent operand sets, registers as the compiler might place them
DEFK(k_mov, "v_mov_b32 v20, v4\n v_mov_b32 v21, v0\n v_mov_b32 v22, v8\n v_mov_b32 v23, v12\n")

typedef void (*K)(float*, int, float);
static void run(const char* name, K k, float* out, int wavesPerSimd)
{
  const int iters = 1000, blocks = 256 * wavesPerSimd;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0000001f); CK(hipDeviceSynchronize());
  int reps = 1; float ms = 0;
  for (;;)
  {
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0000001f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms >= 40.f) break;
    reps *= 2;
  }
  const double per = ms / reps;
  const double inst = (double)iters * 64;                             // per wave
  const double ns = per * 1e6 / inst / wavesPerSimd;                  // per instruction per SIMD
  printf("%-28s waves/SIMD=%d  %.3f ns/instr/SIMD\n", name, wavesPerSimd, ns);
}
int main()
{
  float* out; CK(hipMalloc(&out, 64));
  for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_fma_diff3, dim3(1024), dim3(256), 0, 0, out, 4000, 1.0000001f);
  CK(hipDeviceSynchronize());
  for (int w : {4, 2})
  {
    run("pk_mul srcs same banks", k_pkmul_same, out, w); run("pk_mul srcs diff banks", k_pkmul_diff, out, w);
    run("pk_mul diff, dst other bank", k_pkmul_diff_dst, out, w);
    run("pk_fma 3 srcs same banks", k_pkfma_same3, out, w); run("pk_fma 2 banks", k_pkfma_2banks, out, w);
    run("pk_fma x,2.0,y diff", k_pkfma_inline2, out, w); run("pk_fma x,2.0,y same", k_pkfma_inline2_same, out, w);
    run("pk_add diff", k_pkadd_diff, out, w); run("pk_add op_sel+neg", k_pkadd_opsel, out, w);
    run("mul same bank", k_mul_same, out, w); run("mul diff bank", k_mul_diff, out, w);
    run("fma 3 same bank", k_fma_same3, out, w); run("fma 3 diff banks", k_fma_diff3, out, w);
    run("fma a,b same bank", k_fma_ab_same, out, w); run("fma a,c same bank", k_fma_ac_same, out, w); run("fma b,c same bank", k_fma_bc_same, out, w);
    run("fmac factors same bank", k_fmac_same, out, w); run("fmac all same bank", k_fmac_all_same, out, w); run("fmac diff banks", k_fmac_diff, out, w);
    run("fma x,2.0,y same bank", k_fma_const_same, out, w);
    run("pk_mul 4 chains", k_pk_dep4, out, w); run("pk_mul 2 chains", k_pk_dep2, out, w); run("pk_mul 1 chain", k_pk_dep1, out, w);
    run("v_mov_b32_dpp row_shr:4", k_dpp_shr, out, w); run("v_mov_b32", k_mov, out, w);
  }
  return 0;
}
