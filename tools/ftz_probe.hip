// tools/ftz_probe.hip — what MODE.fp_denorm = 0 (set at run time with s_setreg) does to each gfx950 instruction the
// engine uses, next to what x86 MXCSR FZ|DAZ (ml::UsingFlushDenormalsToZero, MLDSPUtils.h:51-96) does to the SSE
// instruction the reference uses for the same operation. Host and device results for the same operand pairs, both modes.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <xmmintrin.h>
#include <emmintrin.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

enum { OP_ADD, OP_SUB, OP_MUL, OP_DIV, OP_LT, OP_EQ, OP_MIN, OP_MAX, OP_CVT, OP_MUL1, OP_FMA2, OP_SQRT, OP_PKMUL, OP_PKADD, OP_CANON, OP_NEG_CMP, OP_N };
static const char* names[] = {"add", "sub", "mul", "div(IEEE expansion)", "cmp lt", "cmp eq", "min (a<b?a:b)", "max (a>b?a:b)", "cvt f32->i32",
                              "mul by 1.0", "fma(2,a,b) vs b+2a", "sqrt", "pk_mul", "pk_add", "canonicalize (v_max x,x)", "select(a<b, a, b) bits"};
typedef float f2 __attribute__((ext_vector_type(2)));

__global__ void probe(const float* a, const float* b, uint32_t* out, int n, int flush)
{
  if (flush) __builtin_amdgcn_s_setreg(1 | (4 << 6) | (1 << 11), 0);   // MODE[5:4] = 0: flush f32 denormal sources and results
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = a[i], y = b[i];
  float r[OP_N];
  r[OP_ADD] = x + y;
  r[OP_SUB] = x - y;
  r[OP_MUL] = x * y;
  r[OP_DIV] = x / y;
  r[OP_LT] = (x < y) ? 1.f : 0.f;
  r[OP_EQ] = (x == y) ? 1.f : 0.f;
  r[OP_MIN] = __builtin_amdgcn_fmed3f(x, x, x);  // placeholder, replaced below
  {
    float mn, mx;
    asm volatile("v_min_f32 %0, %1, %2" : "=v"(mn) : "v"(x), "v"(y));
    asm volatile("v_max_f32 %0, %1, %2" : "=v"(mx) : "v"(x), "v"(y));
    r[OP_MIN] = mn;
    r[OP_MAX] = mx;
  }
  r[OP_CVT] = (float)__builtin_amdgcn_readfirstlane(0) + (float)(int)__builtin_rintf(x * 1e38f);
  {
    float one;
    asm volatile("v_mov_b32 %0, 1.0" : "=v"(one));
    r[OP_MUL1] = x * one;
  }
  r[OP_FMA2] = __builtin_fmaf(2.0f, x, y);
  r[OP_SQRT] = __builtin_sqrtf(x);
  {
    f2 p = {x, y}, q = {y, x};
    f2 m = p * q, s = p + q;
    r[OP_PKMUL] = m.x;
    r[OP_PKADD] = s.x;
  }
  {
    float c;
    asm volatile("v_max_f32 %0, %1, %1" : "=v"(c) : "v"(x));
    r[OP_CANON] = c;
  }
  r[OP_NEG_CMP] = (x < y) ? x : y;
  for (int k = 0; k < OP_N; ++k) memcpy(&out[(size_t)k * n + i], &r[k], 4);
}

static float hostop(int op, float x, float y)
{
  __m128 a = _mm_set_ss(x), b = _mm_set_ss(y);
  switch (op)
  {
    case OP_ADD: case OP_PKADD: return _mm_cvtss_f32(_mm_add_ss(a, b));
    case OP_SUB: return _mm_cvtss_f32(_mm_sub_ss(a, b));
    case OP_MUL: case OP_PKMUL: return _mm_cvtss_f32(_mm_mul_ss(a, b));
    case OP_DIV: return _mm_cvtss_f32(_mm_div_ss(a, b));
    case OP_LT: return _mm_comilt_ss(a, b) ? 1.f : 0.f;
    case OP_EQ: return _mm_comieq_ss(a, b) ? 1.f : 0.f;
    case OP_MIN: return _mm_cvtss_f32(_mm_min_ss(a, b));
    case OP_MAX: return _mm_cvtss_f32(_mm_max_ss(a, b));
    case OP_CVT: return (float)_mm_cvtss_si32(_mm_mul_ss(a, _mm_set_ss(1e38f)));
    case OP_MUL1: return _mm_cvtss_f32(_mm_mul_ss(a, _mm_set_ss(1.0f)));
    case OP_FMA2: return _mm_cvtss_f32(_mm_add_ss(b, _mm_mul_ss(_mm_set_ss(2.0f), a)));
    case OP_SQRT: return _mm_cvtss_f32(_mm_sqrt_ss(a));
    case OP_CANON: return _mm_cvtss_f32(_mm_max_ss(a, a));
    case OP_NEG_CMP: { __m128 m = _mm_cmplt_ss(a, b); return _mm_cvtss_f32(_mm_or_ps(_mm_and_ps(m, a), _mm_andnot_ps(m, b))); }
  }
  return 0;
}

int main()
{
  std::vector<float> vals;
  auto bits = [](uint32_t u) { float f; memcpy(&f, &u, 4); return f; };
  const uint32_t pats[] = {0x00000000, 0x00000001, 0x00000100, 0x00400000, 0x007fffff, 0x00800000, 0x00800001, 0x00c00000, 0x01000000, 0x01800000,
                           0x0c000000, 0x1f800000, 0x20000000, 0x3f800000, 0x3f000000, 0x40000000, 0x7f7fffff, 0x5f000000, 0x00200000, 0x00000002};
  for (uint32_t p : pats) { vals.push_back(bits(p)); vals.push_back(bits(p | 0x80000000u)); }
  std::vector<float> A, B;
  for (float x : vals) for (float y : vals) { A.push_back(x); B.push_back(y); }
  const int n = (int)A.size();
  float *da, *db; uint32_t* dout;
  CK(hipMalloc(&da, 4 * n)); CK(hipMalloc(&db, 4 * n)); CK(hipMalloc(&dout, 4 * n * OP_N));
  CK(hipMemcpy(da, A.data(), 4 * n, hipMemcpyHostToDevice)); CK(hipMemcpy(db, B.data(), 4 * n, hipMemcpyHostToDevice));
  std::vector<uint32_t> got((size_t)n * OP_N);
  for (int flush = 0; flush < 2; ++flush)
  {
    hipLaunchKernelGGL(probe, dim3((n + 255) / 256), dim3(256), 0, 0, da, db, dout, n, flush);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(got.data(), dout, 4 * (size_t)n * OP_N, hipMemcpyDeviceToHost));
    const unsigned saved = _mm_getcsr();
    if (flush) _mm_setcsr(saved | 0x8040);
    printf("== %s: gfx950 %s  vs  x86 MXCSR %s  (%d operand pairs)\n", flush ? "FLUSH" : "IEEE", flush ? "MODE.fp_denorm32 = 0" : "default mode",
           flush ? "FZ|DAZ" : "default", n);
    for (int op = 0; op < OP_N; ++op)
    {
      int bad = 0, shown = 0;
      for (int i = 0; i < n; ++i)
      {
        const float h = hostop(op, A[i], B[i]);
        uint32_t hb; memcpy(&hb, &h, 4);
        const uint32_t gb = got[(size_t)op * n + i];
        const bool bothNaN = (hb & 0x7fffffff) > 0x7f800000 && (gb & 0x7fffffff) > 0x7f800000;
        if (hb != gb && !bothNaN)
        {
          ++bad;
          if (shown++ < 3)
          {
            uint32_t ab, bb; memcpy(&ab, &A[i], 4); memcpy(&bb, &B[i], 4);
            printf("      %-26s a=%08x b=%08x  x86=%08x gpu=%08x\n", names[op], ab, bb, hb, gb);
          }
        }
      }
      printf("   %-28s mismatches %d\n", names[op], bad);
    }
    _mm_setcsr(saved);
  }
  return 0;
}
