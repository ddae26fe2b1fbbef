// ops.hip — stateless DSPVector ops of MLDSPOps.h as gfx950 streaming kernels.
//
// Elementwise ops are pure HBM streaming (8-16 B/element, 2-60 VALU ops): every lane moves
// 16 bytes per memory instruction (global_load/store_dwordx4), a wavefront 1 KiB, and the
// grid is sized to a few waves per SIMD with a grid-stride loop so the launch is one wave of
// workgroups over all 256 CUs. No LDS, no MFMA: there is no reuse and no contraction.
//
// Compile with -ffp-contract=off (see mldsp_math.hpp).
#include "mlgpu_internal.hpp"
#include "mldsp_ops.hpp"

using namespace mldev;

namespace
{
// native clang vector (HIP's uint4 is a struct and cannot feed the nontemporal builtins)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// UNROLL independent 16-byte loads per lane in flight before any math: enough bytes in
// flight per CU to cover HBM latency at ~4 waves/SIMD.
constexpr int kOpBlock = 256;
constexpr int kOpUnroll = 4;

template <int OP>
__global__ __launch_bounds__(kOpBlock) void op_kernel(const u32x4* a, const u32x4* b,
                                                      const u32x4* c, u32x4* out,
                                                      size_t n4, size_t n)
{
  constexpr int AR = arity<OP>();
  const size_t stride = (size_t)gridDim.x * kOpBlock;
  size_t i = (size_t)blockIdx.x * kOpBlock + threadIdx.x;
  for (; i + (kOpUnroll - 1) * stride < n4; i += kOpUnroll * stride)
  {
    u32x4 va[kOpUnroll], vb[kOpUnroll], vc[kOpUnroll];
#pragma unroll
    for (int u = 0; u < kOpUnroll; ++u)
    {
      va[u] = __builtin_nontemporal_load(&a[i + u * stride]);
      if (AR >= 2) vb[u] = __builtin_nontemporal_load(&b[i + u * stride]);
      if (AR >= 3) vc[u] = __builtin_nontemporal_load(&c[i + u * stride]);
    }
#pragma unroll
    for (int u = 0; u < kOpUnroll; ++u)
    {
      u32x4 r;
      r.x = apply<OP>(va[u].x, AR >= 2 ? vb[u].x : 0u, AR >= 3 ? vc[u].x : 0u);
      r.y = apply<OP>(va[u].y, AR >= 2 ? vb[u].y : 0u, AR >= 3 ? vc[u].y : 0u);
      r.z = apply<OP>(va[u].z, AR >= 2 ? vb[u].z : 0u, AR >= 3 ? vc[u].z : 0u);
      r.w = apply<OP>(va[u].w, AR >= 2 ? vb[u].w : 0u, AR >= 3 ? vc[u].w : 0u);
      __builtin_nontemporal_store(r, &out[i + u * stride]);
    }
  }
  for (; i < n4; i += stride)
  {
    const u32x4 va = a[i];
    u32x4 vb = va, vc = va;
    if (AR >= 2) vb = b[i];
    if (AR >= 3) vc = c[i];
    u32x4 r;
    r.x = apply<OP>(va.x, vb.x, vc.x);
    r.y = apply<OP>(va.y, vb.y, vc.y);
    r.z = apply<OP>(va.z, vb.z, vc.z);
    r.w = apply<OP>(va.w, vb.w, vc.w);
    out[i] = r;
  }
  // scalar tail (n not a multiple of 4): handled by the first few lanes of block 0
  if (blockIdx.x == 0)
  {
    const size_t t = n4 * 4 + threadIdx.x;
    if (t < n)
    {
      const uint32_t* sa = (const uint32_t*)a;
      const uint32_t* sb = (const uint32_t*)b;
      const uint32_t* sc = (const uint32_t*)c;
      ((uint32_t*)out)[t] = apply<OP>(sa[t], AR >= 2 ? sb[t] : 0u, AR >= 3 ? sc[t] : 0u);
    }
  }
}

template <int OP>
hipError_t launchOp(const void* a, const void* b, const void* c, void* out, size_t n, hipStream_t stream,
                    int cuCount)
{
  const size_t n4 = n / 4;
  size_t blocks = (n4 + (size_t)kOpBlock * kOpUnroll - 1) / ((size_t)kOpBlock * kOpUnroll);
  const size_t cap = (size_t)cuCount * 8;  // 8 x 256-thread workgroups per CU = 32 waves/CU
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(op_kernel<OP>, dim3((unsigned)blocks), dim3(kOpBlock), 0, stream, (const u32x4*)a,
                     (const u32x4*)b, (const u32x4*)c, (u32x4*)out, n4, n);
  return hipGetLastError();
}

// add1..max1: b is one row of 64, repeated for every row of a (MLDSPOps.h:655-687)
template <int OP>
__global__ __launch_bounds__(kOpBlock) void op_rows1_kernel(const u32x4* a,
                                                            const u32x4* b64,
                                                            u32x4* out, size_t n4)
{
  const size_t stride = (size_t)gridDim.x * kOpBlock;
  for (size_t i = (size_t)blockIdx.x * kOpBlock + threadIdx.x; i < n4; i += stride)
  {
    const u32x4 va = a[i];
    const u32x4 vb = b64[i & 15];  // 16 float4 per row
    u32x4 r;
    r.x = apply<OP>(va.x, vb.x, 0u);
    r.y = apply<OP>(va.y, vb.y, 0u);
    r.z = apply<OP>(va.z, vb.z, 0u);
    r.w = apply<OP>(va.w, vb.w, 0u);
    out[i] = r;
  }
}
template <int OP>
hipError_t launchRows1(const void* a, const void* b, void* out, size_t nRows, hipStream_t stream, int cuCount)
{
  const size_t n4 = nRows * 16;
  size_t blocks = (n4 + kOpBlock - 1) / kOpBlock;
  const size_t cap = (size_t)cuCount * 8;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(op_rows1_kernel<OP>, dim3((unsigned)blocks), dim3(kOpBlock), 0, stream, (const u32x4*)a,
                     (const u32x4*)b, (u32x4*)out, n4);
  return hipGetLastError();
}

// horizontal per-row ops, MLDSPOps.h:995-1035 + vecSumH/MaxH/MinH (MLDSPMathSSE.h:246-265):
// one lane per row, 16 x float4 loads; association order exactly as the reference:
// per 4-group (x0 op x2) op (x1 op x3), then left-to-right over the 16 groups.
template <int ROWOP>
__global__ __launch_bounds__(256) void row_reduce_kernel(const float4* rows, float* out,
                                                         size_t nRows)
{
  const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nRows) return;
  const float4* p = rows + r * 16;
  float acc = (ROWOP == MLGPU_ROWOP_MAX) ? 1.17549435e-38f /* FLT_MIN, sic: MLDSPOps.h:1016 */
              : (ROWOP == MLGPU_ROWOP_MIN) ? 3.402823466e+38f
                                           : 0.f;
#pragma unroll
  for (int g = 0; g < 16; ++g)
  {
    const float4 q = p[g];
    if (ROWOP == MLGPU_ROWOP_SUM || ROWOP == MLGPU_ROWOP_MEAN)
    {
      const float t0 = q.x + q.z, t1 = q.y + q.w;
      acc += (t0 + t1);
    }
    else if (ROWOP == MLGPU_ROWOP_MAX)
    {
      const float h = sse_max(sse_max(q.x, q.z), sse_max(q.y, q.w));
      acc = (acc > h) ? acc : h;
    }
    else
    {
      const float h = sse_min(sse_min(q.x, q.z), sse_min(q.y, q.w));
      acc = (acc < h) ? acc : h;
    }
  }
  out[r] = (ROWOP == MLGPU_ROWOP_MEAN) ? acc * (1.0f / 64.f) : acc;
}

// layout conversion: one lane per (vector, quad, voice) float4; reads follow the source
// order, writes scatter in 16-byte units (always whole 16-byte words in every layout).
__global__ __launch_bounds__(256) void layout_convert_kernel(SignalView src, SignalView dst, size_t V, size_t T,
                                                             int dstLayout)
{
  const size_t total = V * T * 16;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride)
  {
    // enumerate in DESTINATION order so stores are coalesced
    size_t t, q, v;
    if (dstLayout == MLGPU_LAYOUT_QUAD)
    {
      v = i % V;
      q = (i / V) % 16;
      t = i / (16 * V);
    }
    else if (dstLayout == MLGPU_LAYOUT_ROWS)
    {
      q = i % 16;
      v = (i / 16) % V;
      t = i / (16 * V);
    }
    else
    {
      q = i % 16;
      t = (i / 16) % T;
      v = i / (16 * T);
    }
    dst.base[t * dst.strideT + q * dst.strideQ + v * dst.strideV] =
        src.base[t * src.strideT + q * src.strideQ + v * src.strideV];
  }
}

__global__ __launch_bounds__(256) void fill32_kernel(uint32_t* dst, uint32_t value, size_t n)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = value;
}
}  // namespace

#define OP_CASE(OP) \
  case OP: return launchOp<OP>(a, b, c, out, n, stream, cuCount);

hipError_t mlgpu_launch_op(int op, const void* a, const void* b, const void* c, void* out, size_t n,
                           hipStream_t stream, int cuCount, bool* known)
{
  *known = true;
  switch (op)
  {
    OP_CASE(MLGPU_OP_SQRT)
    OP_CASE(MLGPU_OP_SQRT_APPROX)
    OP_CASE(MLGPU_OP_ABS)
    OP_CASE(MLGPU_OP_SIGN)
    OP_CASE(MLGPU_OP_SIGN_BIT)
    OP_CASE(MLGPU_OP_SIN)
    OP_CASE(MLGPU_OP_COS)
    OP_CASE(MLGPU_OP_LOG)
    OP_CASE(MLGPU_OP_EXP)
    OP_CASE(MLGPU_OP_LOG2)
    OP_CASE(MLGPU_OP_EXP2)
    OP_CASE(MLGPU_OP_SIN_APPROX)
    OP_CASE(MLGPU_OP_COS_APPROX)
    OP_CASE(MLGPU_OP_EXP_APPROX)
    OP_CASE(MLGPU_OP_LOG_APPROX)
    OP_CASE(MLGPU_OP_LOG2_APPROX)
    OP_CASE(MLGPU_OP_EXP2_APPROX)
    OP_CASE(MLGPU_OP_FRACTIONAL_PART)
    OP_CASE(MLGPU_OP_ROUND_FLOAT_TO_INT)
    OP_CASE(MLGPU_OP_TRUNCATE_FLOAT_TO_INT)
    OP_CASE(MLGPU_OP_INT_TO_FLOAT)
    OP_CASE(MLGPU_OP_UNSIGNED_INT_TO_FLOAT)
    OP_CASE(MLGPU_OP_EXP_APPROX_OF_SIN_APPROX)
    OP_CASE(MLGPU_OP_ADD)
    OP_CASE(MLGPU_OP_SUBTRACT)
    OP_CASE(MLGPU_OP_MULTIPLY)
    OP_CASE(MLGPU_OP_DIVIDE)
    OP_CASE(MLGPU_OP_DIVIDE_APPROX)
    OP_CASE(MLGPU_OP_POW)
    OP_CASE(MLGPU_OP_POW_APPROX)
    OP_CASE(MLGPU_OP_MIN)
    OP_CASE(MLGPU_OP_MAX)
    OP_CASE(MLGPU_OP_ADD_INT32)
    OP_CASE(MLGPU_OP_SUBTRACT_INT32)
    OP_CASE(MLGPU_OP_EQUAL)
    OP_CASE(MLGPU_OP_NOT_EQUAL)
    OP_CASE(MLGPU_OP_GREATER_THAN)
    OP_CASE(MLGPU_OP_GREATER_THAN_OR_EQUAL)
    OP_CASE(MLGPU_OP_LESS_THAN)
    OP_CASE(MLGPU_OP_LESS_THAN_OR_EQUAL)
    OP_CASE(MLGPU_OP_LERP)
    OP_CASE(MLGPU_OP_INVERSE_LERP)
    OP_CASE(MLGPU_OP_CLAMP)
    OP_CASE(MLGPU_OP_WITHIN)
    OP_CASE(MLGPU_OP_SELECT)
    OP_CASE(MLGPU_OP_SELECT_INT)
    default: *known = false; return hipSuccess;
  }
}

#define ROWS1_CASE(OP) \
  case OP: return launchRows1<OP>(a, b64, out, nRows, stream, cuCount);

hipError_t mlgpu_launch_op_rows1(int op, const void* a, const void* b64, void* out, size_t nRows,
                                 hipStream_t stream, int cuCount, bool* known)
{
  *known = true;
  switch (op)
  {
    ROWS1_CASE(MLGPU_OP_ADD)
    ROWS1_CASE(MLGPU_OP_SUBTRACT)
    ROWS1_CASE(MLGPU_OP_MULTIPLY)
    ROWS1_CASE(MLGPU_OP_DIVIDE)
    ROWS1_CASE(MLGPU_OP_DIVIDE_APPROX)
    ROWS1_CASE(MLGPU_OP_POW)
    ROWS1_CASE(MLGPU_OP_POW_APPROX)
    ROWS1_CASE(MLGPU_OP_MIN)
    ROWS1_CASE(MLGPU_OP_MAX)
    default: *known = false; return hipSuccess;
  }
}

hipError_t mlgpu_launch_row_reduce(int rowop, const float* rows, float* out, size_t nRows, hipStream_t stream,
                                   bool* known)
{
  *known = true;
  const unsigned blocks = (unsigned)((nRows + 255) / 256);
  switch (rowop)
  {
    case MLGPU_ROWOP_SUM:
      hipLaunchKernelGGL(row_reduce_kernel<MLGPU_ROWOP_SUM>, dim3(blocks), dim3(256), 0, stream,
                         (const float4*)rows, out, nRows);
      break;
    case MLGPU_ROWOP_MEAN:
      hipLaunchKernelGGL(row_reduce_kernel<MLGPU_ROWOP_MEAN>, dim3(blocks), dim3(256), 0, stream,
                         (const float4*)rows, out, nRows);
      break;
    case MLGPU_ROWOP_MAX:
      hipLaunchKernelGGL(row_reduce_kernel<MLGPU_ROWOP_MAX>, dim3(blocks), dim3(256), 0, stream,
                         (const float4*)rows, out, nRows);
      break;
    case MLGPU_ROWOP_MIN:
      hipLaunchKernelGGL(row_reduce_kernel<MLGPU_ROWOP_MIN>, dim3(blocks), dim3(256), 0, stream,
                         (const float4*)rows, out, nRows);
      break;
    default: *known = false; return hipSuccess;
  }
  return hipGetLastError();
}

hipError_t mlgpu_launch_layout_convert(const float* src, int srcLayout, float* dst, int dstLayout, size_t V,
                                       size_t T, hipStream_t stream)
{
  const size_t total = V * T * 16;
  size_t blocks = (total + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(layout_convert_kernel, dim3((unsigned)blocks), dim3(256), 0, stream,
                     makeView(src, srcLayout, V, T), makeView(dst, dstLayout, V, T), V, T, dstLayout);
  return hipGetLastError();
}

hipError_t mlgpu_launch_fill32(uint32_t* dst, uint32_t value, size_t n, hipStream_t stream)
{
  size_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(fill32_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, dst, value, n);
  return hipGetLastError();
}
