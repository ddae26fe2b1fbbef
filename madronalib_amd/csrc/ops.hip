// ops.hip — stateless DSPVector ops of MLDSPOps.h as gfx950 streaming kernels.
//
// Elementwise ops are pure HBM streaming (8-16 B/element, 2-60 VALU ops): every lane moves
// 16 bytes per memory instruction (global_load/store_dwordx4), a wavefront 1 KiB, and the
// grid is sized to a few waves per SIMD with a grid-stride loop so the launch is one wave of
// workgroups over all 256 CUs. No LDS, no MFMA: there is no reuse and no contraction.
//
// Compile with -ffp-contract=off (see mldsp_math.hpp).
#include <string.h>

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
                                                      size_t n4, size_t n, uint32_t flags)
{
  apply_fp_mode(flags);
  constexpr int AR = arity<OP>();
  // XCD-aware element mapping: workgroup b runs on XCD b % 8; give XCD x the x-th contiguous eighth of the array (each
  // XCD's L2 then streams one contiguous segment), grid-stride inside it. Any bijection is correct; this is for speed:
  // expApprox(sinApprox(x)) over 1 GiB 4.99 -> 5.67 TB/s, over 16 MiB (Infinity-Cache resident) 6.15 -> 6.35.
  size_t stride = (size_t)gridDim.x * kOpBlock;
  size_t i = (size_t)blockIdx.x * kOpBlock + threadIdx.x;
  size_t end = n4;
  if ((gridDim.x & 7) == 0 && n4 >= (size_t)gridDim.x * kOpBlock * kOpUnroll)
  {
    const size_t per = (n4 / 8) & ~(size_t)(kOpBlock - 1), x = blockIdx.x & 7;
    stride = (size_t)(gridDim.x >> 3) * kOpBlock;
    i = x * per + (size_t)(blockIdx.x >> 3) * kOpBlock + threadIdx.x;
    end = (x == 7) ? n4 : (x + 1) * per;
  }
  n4 = end;
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
                    int cuCount, uint32_t flags)
{
  const size_t n4 = n / 4;
  size_t blocks = (n4 + (size_t)kOpBlock * kOpUnroll - 1) / ((size_t)kOpBlock * kOpUnroll);
  const size_t cap = (size_t)cuCount * 8;  // 8 x 256-thread workgroups per CU = 32 waves/CU
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(op_kernel<OP>, dim3((unsigned)blocks), dim3(kOpBlock), 0, stream, (const u32x4*)a,
                     (const u32x4*)b, (const u32x4*)c, (u32x4*)out, n4, n, flags);
  return hipGetLastError();
}

// add1..max1: b is one row of 64, repeated for every row of a (MLDSPOps.h:655-687)
template <int OP>
__global__ __launch_bounds__(kOpBlock) void op_rows1_kernel(const u32x4* a,
                                                            const u32x4* b64,
                                                            u32x4* out, size_t n4, uint32_t flags)
{
  apply_fp_mode(flags);
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
hipError_t launchRows1(const void* a, const void* b, void* out, size_t nRows, hipStream_t stream, int cuCount, uint32_t flags)
{
  const size_t n4 = nRows * 16;
  size_t blocks = (n4 + kOpBlock - 1) / kOpBlock;
  const size_t cap = (size_t)cuCount * 8;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(op_rows1_kernel<OP>, dim3((unsigned)blocks), dim3(kOpBlock), 0, stream, (const u32x4*)a,
                     (const u32x4*)b, (u32x4*)out, n4, flags);
  return hipGetLastError();
}

// horizontal per-row ops, MLDSPOps.h:995-1035 + vecSumH/MaxH/MinH (MLDSPMathSSE.h:246-265):
// one lane per row, 16 x float4 loads; association order exactly as the reference:
// per 4-group (x0 op x2) op (x1 op x3), then left-to-right over the 16 groups.
template <int ROWOP>
__global__ __launch_bounds__(256) void row_reduce_kernel(const float4* rows, float* out,
                                                         size_t nRows, uint32_t flags)
{
  apply_fp_mode(flags);
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
      acc = sse_max(acc, h);  // vecMax(acc, h)
    }
    else
    {
      const float h = sse_min(sse_min(q.x, q.z), sse_min(q.y, q.w));
      acc = sse_min(acc, h);  // vecMin(acc, h)
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

// QUAD <-> a layout whose 16 quads of one (voice, vector) are contiguous (VOICE_MAJOR, ROWS): a transpose. One wavefront per
// tile of 64 voices x 16 quads, through its own 16 KiB of LDS: both the global reads and the global writes run along the
// contiguous direction of their side (256-byte voice rows on one, 1 KiB quad rows on the other). The quad index is XOR-ed
// with the voice index in the LDS address, which keeps both phases free of bank conflicts without padding.
// Tile order: with VOICE_MAJOR on the row side a voice's DSPVectors are adjacent (256 bytes each), so neighbouring wavefronts take
// neighbouring vectors of the same 64 voices (vectorsFirst) and each voice's stretch of DRAM is visited once, 1 KiB or more at a
// time, instead of once per vector 256 bytes at a time; with ROWS the 64 voices of one vector are one contiguous 16 KiB already.
template <bool TO_QUAD>
__global__ __launch_bounds__(256) void layout_transpose_kernel(SignalView src, SignalView dst, size_t V, size_t T, bool vectorsFirst)
{
  __shared__ float4 tiles[4][64 * 16];
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float4* tile = tiles[wave];
  const size_t voiceBlocks = (V + 63) / 64, items = T * voiceBlocks;
  for (size_t item = (size_t)blockIdx.x * 4 + wave; item < items; item += (size_t)gridDim.x * 4)
  {
    const size_t t = vectorsFirst ? item % T : item / voiceBlocks, v0 = (vectorsFirst ? item / T : item % voiceBlocks) * 64;
    const SignalView& rowSide = TO_QUAD ? src : dst;   // voice rows of 16 quads
    const SignalView& quadSide = TO_QUAD ? dst : src;  // quad rows of 64 voices
    if (TO_QUAD)
    {
#pragma unroll
      for (int i = 0; i < 16; ++i)
      {
        const unsigned vl = 4 * i + (lane >> 4), q = lane & 15;
        if (v0 + vl < V) tile[vl * 16 + (q ^ (vl & 15))] = rowSide.base[t * rowSide.strideT + q * rowSide.strideQ + (v0 + vl) * rowSide.strideV];
      }
#pragma unroll
      for (int q = 0; q < 16; ++q)
        if (v0 + lane < V) quadSide.base[t * quadSide.strideT + q * quadSide.strideQ + (v0 + lane) * quadSide.strideV] = tile[lane * 16 + (q ^ (lane & 15))];
    }
    else
    {
#pragma unroll
      for (int q = 0; q < 16; ++q)
        if (v0 + lane < V) tile[lane * 16 + (q ^ (lane & 15))] = quadSide.base[t * quadSide.strideT + q * quadSide.strideQ + (v0 + lane) * quadSide.strideV];
#pragma unroll
      for (int i = 0; i < 16; ++i)
      {
        const unsigned vl = 4 * i + (lane >> 4), q = lane & 15;
        if (v0 + vl < V) rowSide.base[t * rowSide.strideT + q * rowSide.strideQ + (v0 + vl) * rowSide.strideV] = tile[vl * 16 + (q ^ (vl & 15))];
      }
    }
    // the tile is private to this wavefront and its LDS operations complete in order: no barrier between tiles
  }
}

__global__ __launch_bounds__(256) void fill32_kernel(uint32_t* dst, uint32_t value, size_t n)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = value;
}

// ---------------------------------------------------------------------------------------------
// row plumbing (MLDSPOps.h:1057-1343): pure data movement between DSPVectorArrays. One lane moves one
// float4 of a destination row; the source row comes from a small closed-form rule (mlgpu_rows_rule),
// so no index table has to be uploaded. `groups` independent arrays (e.g. one DSPVectorArray per
// voice) are processed by one launch.
struct RowsMapArgs
{
  const float4* src;
  float4* dst;
  int rule;
  long p0, p1;
  int sampleRotate;  // 0, +1 (rotateLeft: y[n] = x[n+1]), -1 (rotateRight: y[n] = x[n-1]), wrapping inside the row
  size_t srcRows, dstRows, dstOffset, dstStep, count, groups;
};

__device__ __forceinline__ long rowsSource(const RowsMapArgs& a, long j)
{
  const long N = (long)a.srcRows;
  switch (a.rule)
  {
    case MLGPU_ROWS_REPEAT: return j % N;
    case MLGPU_ROWS_STRETCH:  // int k = roundf((j * (N - 1.f)) / (ROWS - 1.f)), MLDSPOps.h:1080
      return (a.count < 2) ? 0 : (long)__builtin_roundf(((float)j * ((float)N - 1.f)) / ((float)a.count - 1.f));
    case MLGPU_ROWS_SHIFT:
    {
      const long k = j - a.p0;
      return (k >= 0 && k < N) ? k : -1;
    }
    case MLGPU_ROWS_ROTATE:
    {
      long k = (j - a.p0) % N;
      return k < 0 ? k + N : k;
    }
    default:  // MLGPU_ROWS_STRIDED
    {
      const long k = a.p0 + j * a.p1;
      return (k >= 0 && k < N) ? k : -1;
    }
  }
}

__global__ __launch_bounds__(256) void rows_map_kernel(const RowsMapArgs a)
{
  const size_t total = a.groups * a.count * 16;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride)
  {
    const size_t q = i & 15, j = (i >> 4) % a.count, g = (i >> 4) / a.count;
    const long s = rowsSource(a, (long)j);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (s >= 0)
    {
      const float4* row = a.src + (g * a.srcRows + (size_t)s) * 16;
      if (a.sampleRotate == 0)
        v = row[q];
      else
      {
        const float* f = (const float*)row;
        const int n0 = (int)q * 4 + a.sampleRotate;
        v.x = f[(n0 + 64) & 63];
        v.y = f[(n0 + 65) & 63];
        v.z = f[(n0 + 66) & 63];
        v.w = f[(n0 + 67) & 63];
      }
    }
    a.dst[(g * a.dstRows + a.dstOffset + j * a.dstStep) * 16 + q] = v;
  }
}

// addRows, MLDSPOps.h:1349-1359: vy = 0; vy = vy + row_j for j = 0..ROWS-1 (left to right, starting from +0)
__global__ __launch_bounds__(256) void rows_add_kernel(const float4* rows, float4* out, size_t rowsPerGroup, size_t groups, uint32_t flags)
{
  apply_fp_mode(flags);
  const size_t total = groups * 16;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride)
  {
    const size_t q = i & 15, g = i >> 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (size_t j = 0; j < rowsPerGroup; ++j)
    {
      const float4 x = rows[(g * rowsPerGroup + j) * 16 + q];
      acc.x = acc.x + x.x;
      acc.y = acc.y + x.y;
      acc.z = acc.z + x.z;
      acc.w = acc.w + x.w;
    }
    out[i] = acc;
  }
}

// normalize, MLDSPOps.h:1041-1050: row / sum(row), sum in the reference's association order; one lane per row
__global__ __launch_bounds__(256) void rows_normalize_kernel(const float4* rows, float4* out, size_t nRows, uint32_t flags)
{
  apply_fp_mode(flags);
  const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nRows) return;
  const float4* p = rows + r * 16;
  float acc = 0.f;
#pragma unroll
  for (int g = 0; g < 16; ++g)
  {
    const float4 q = p[g];
    const float t0 = q.x + q.z, t1 = q.y + q.w;
    acc += (t0 + t1);
  }
#pragma unroll
  for (int g = 0; g < 16; ++g)
  {
    const float4 q = p[g];
    out[r * 16 + g] = make_float4(q.x / acc, q.y / acc, q.z / acc, q.w / acc);
  }
}

// rowIndex<ROWS>(), MLDSPOps.h:1365-1374: row j of every group filled with (float)j
__global__ __launch_bounds__(256) void rows_index_kernel(float4* out, size_t rowsPerGroup, size_t groups)
{
  const size_t total = groups * rowsPerGroup * 16;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride)
  {
    const float j = (float)((i >> 4) % rowsPerGroup);
    out[i] = make_float4(j, j, j, j);
  }
}

// ---------------------------------------------------------------------------------------------
// routing, MLDSPRouting.h:83-234: per-sample selection among N signals by a selector in [0, 1).
// Scalar per sample in the reference ("TODO SIMD"); here one lane per element, N pointers in the args.
struct RouteArgs
{
  const float* sel;
  size_t selElems;  // selector index = i % selElems (64: one selector row for every row, as the reference)
  const float* in[MLGPU_ROUTE_MAX];
  float* out[MLGPU_ROUTE_MAX];
  int n;
  size_t nElems;
  uint32_t flags;
};

template <bool LINEAR>
__global__ __launch_bounds__(256) void multiplex_kernel(const RouteArgs a)
{
  apply_fp_mode(a.flags);
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.nElems; i += stride)
  {
    const float s = a.sel[i % a.selElems];
    float x[MLGPU_ROUTE_MAX];
#pragma unroll
    for (int k = 0; k < MLGPU_ROUTE_MAX; ++k) x[k] = (k < a.n) ? a.in[k][i] : 0.f;
    a.out[0][i] = LINEAR ? route_multiplex_linear(s, x, a.n) : route_multiplex(s, x, a.n);
  }
}

template <bool LINEAR>
__global__ __launch_bounds__(256) void demultiplex_kernel(const RouteArgs a)
{
  apply_fp_mode(a.flags);
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.nElems; i += stride)
  {
    const float s = a.sel[i % a.selElems];
    const float x = a.in[0][i];
    for (int k = 0; k < a.n; ++k) a.out[k][i] = LINEAR ? route_demultiplex_linear(s, x, k, a.n) : route_demultiplex(s, x, k, a.n);
  }
}
}  // namespace

#define OP_CASE(OP) \
  case OP: return launchOp<OP>(a, b, c, out, n, stream, cuCount, flags);

hipError_t mlgpu_launch_op(int op, const void* a, const void* b, const void* c, void* out, size_t n,
                           hipStream_t stream, int cuCount, bool* known, uint32_t flags)
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
    OP_CASE(MLGPU_OP_PHASOR_TO_SINE)
    OP_CASE(MLGPU_OP_PHASOR_TO_SAW)
    OP_CASE(MLGPU_OP_PHASOR_TO_PULSE)
    default: *known = false; return hipSuccess;
  }
}

#define ROWS1_CASE(OP) \
  case OP: return launchRows1<OP>(a, b64, out, nRows, stream, cuCount, flags);

hipError_t mlgpu_launch_op_rows1(int op, const void* a, const void* b64, void* out, size_t nRows,
                                 hipStream_t stream, int cuCount, bool* known, uint32_t flags)
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
                                   bool* known, uint32_t flags)
{
  *known = true;
  const unsigned blocks = (unsigned)((nRows + 255) / 256);
  switch (rowop)
  {
    case MLGPU_ROWOP_SUM:
      hipLaunchKernelGGL(row_reduce_kernel<MLGPU_ROWOP_SUM>, dim3(blocks), dim3(256), 0, stream,
                         (const float4*)rows, out, nRows, flags);
      break;
    case MLGPU_ROWOP_MEAN:
      hipLaunchKernelGGL(row_reduce_kernel<MLGPU_ROWOP_MEAN>, dim3(blocks), dim3(256), 0, stream,
                         (const float4*)rows, out, nRows, flags);
      break;
    case MLGPU_ROWOP_MAX:
      hipLaunchKernelGGL(row_reduce_kernel<MLGPU_ROWOP_MAX>, dim3(blocks), dim3(256), 0, stream,
                         (const float4*)rows, out, nRows, flags);
      break;
    case MLGPU_ROWOP_MIN:
      hipLaunchKernelGGL(row_reduce_kernel<MLGPU_ROWOP_MIN>, dim3(blocks), dim3(256), 0, stream,
                         (const float4*)rows, out, nRows, flags);
      break;
    default: *known = false; return hipSuccess;
  }
  return hipGetLastError();
}

hipError_t mlgpu_launch_layout_convert(const float* src, int srcLayout, float* dst, int dstLayout, size_t V,
                                       size_t T, hipStream_t stream)
{
  const size_t total = V * T * 16;
  const bool toQuad = (dstLayout == MLGPU_LAYOUT_QUAD && srcLayout != MLGPU_LAYOUT_QUAD && srcLayout != MLGPU_LAYOUT_BROADCAST);
  const bool fromQuad = (srcLayout == MLGPU_LAYOUT_QUAD && dstLayout != MLGPU_LAYOUT_QUAD);
  if (toQuad || fromQuad)
  {
    const size_t items = T * ((V + 63) / 64);
    size_t blocks = (items + 3) / 4;
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (blocks < 1) blocks = 1;
    const bool vectorsFirst = (toQuad ? srcLayout : dstLayout) == MLGPU_LAYOUT_VOICE_MAJOR;
    if (toQuad)
      hipLaunchKernelGGL(layout_transpose_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, stream, makeView(src, srcLayout, V, T), makeView(dst, dstLayout, V, T), V, T, vectorsFirst);
    else
      hipLaunchKernelGGL(layout_transpose_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, stream, makeView(src, srcLayout, V, T), makeView(dst, dstLayout, V, T), V, T, vectorsFirst);
    return hipGetLastError();
  }
  size_t blocks = (total + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(layout_convert_kernel, dim3((unsigned)blocks), dim3(256), 0, stream,
                     makeView(src, srcLayout, V, T), makeView(dst, dstLayout, V, T), V, T, dstLayout);
  return hipGetLastError();
}

// validate(): ml::validate(DSPVector) (MLDSPOps.h:1430-1445) flags a sample that is NaN or larger than 1e8 in magnitude
// ("maxUsefulValue"). Here: one pass over a whole signal, result[0] = how many such samples, result[1] = index of the
// first one (~0 when none). Read-only streaming at 16 B per lane; a wave reduces with DPP-free shuffles, one atomic pair
// per wave that found something.
__global__ __launch_bounds__(256) void validate_kernel(const float4* x, size_t n4, size_t n, unsigned long long* result)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  unsigned long long count = 0, first = ~0ull;
  auto bad = [](float v) { return (v != v) || (__builtin_fabsf(v) > 1e8f); };
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride)
  {
    const float4 q = x[i];
    const bool b0 = bad(q.x), b1 = bad(q.y), b2 = bad(q.z), b3 = bad(q.w);
    if (b0 | b1 | b2 | b3)
    {
      count += (unsigned)b0 + (unsigned)b1 + (unsigned)b2 + (unsigned)b3;
      const unsigned long long at = 4ull * i + (b0 ? 0 : b1 ? 1 : b2 ? 2 : 3);
      first = at < first ? at : first;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3))   // scalar tail
  {
    const size_t t = 4 * n4 + threadIdx.x;
    if (bad(((const float*)x)[t]))
    {
      ++count;
      first = t < first ? (unsigned long long)t : first;
    }
  }
  if (__builtin_amdgcn_ballot_w64(count != 0) == 0) return;   // the common case: nothing to report from this wave
  for (int off = 32; off > 0; off >>= 1)
  {
    count += __shfl_down(count, off, 64);
    const unsigned long long o = __shfl_down(first, off, 64);
    first = o < first ? o : first;
  }
  if ((threadIdx.x & 63) == 0)
  {
    atomicAdd(&result[0], count);
    atomicMin(&result[1], first);
  }
}

hipError_t mlgpu_launch_validate(const float* x, size_t n, unsigned long long* d_result, hipStream_t stream, int cuCount)
{
  const unsigned long long init[2] = {0ull, ~0ull};
  hipError_t err = hipMemcpyAsync(d_result, init, sizeof(init), hipMemcpyHostToDevice, stream);
  if (err != hipSuccess) return err;
  const size_t n4 = n / 4;
  size_t blocks = (n4 + 255) / 256;
  if (blocks > (size_t)cuCount * 8) blocks = (size_t)cuCount * 8;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(validate_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const float4*)x, n4, n, d_result);
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

// ---------------------------------------------------------------------------------------------
// mixdown: sum the voices of a signal into one single-voice signal (a Synth's `outputs += voice`, MLSynth.h:43-57).
// Stage 1: one wavefront per group of 64 consecutive voices, lane = voice, fixed pairwise tree over the lanes
// (a[i] += a[i + d] for d = 1, 2, ... 32); then the group sums are added left to right, 64 consecutive ones at a time,
// and so are the results, until one is left (up to 4096 voices that is simply "the groups left to right"). Missing voices
// of the last group count as +0. The order is part of the contract (include/mlgpu.h) so results are reproducible and checkable.
namespace
{
// One wavefront per (group of 64 voices, DSPVector): lane v loads its voice's 16 quads (coalesced: 1 KiB per quad and wavefront),
// scales them, and parks them as row v of a 64 x 64 tile in LDS (row stride 68 floats: 16-byte aligned for the b128 writes, and the
// column reads below fall on 64 different banks); lane l then adds up column l - sample l of the 64 voices - in the contract's order,
// the balanced tree over consecutive voices, and the wavefront writes its 64 sums as one 256-byte row. (Round 3's form reduced every
// quad across the lanes with six shuffle-and-add rounds per component, one quad per wavefront at T = 1: 0.35 of HBM and a quarter of a
// million tiny wavefronts for 2^20 voices.) a[i] += a[i + d] for d = 1, 2, 4 ... is exactly tree(0, 64) below.
constexpr int kMixRow = 68;
// mixdown_stage1_kernel parks 4 wavefronts x 64 voices x 68 floats = 69 632 bytes in LDS: more than the 64 KiB a workgroup gets on
// every AMD target but gfx950 (160 KiB per CU). This library is written for gfx950 alone; say so instead of an opaque backend error.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libmlgpu is written for gfx950 (MI355X): mixdown_stage1_kernel needs 68 KiB of LDS per workgroup, other targets allow 64 KiB"
#endif
static_assert(4 * 64 * kMixRow * sizeof(float) <= 160 * 1024, "mixdown_stage1_kernel's LDS tile must fit gfx950's 160 KiB per CU");
template <int LO, int N>
struct MixTree
{
  static __device__ __forceinline__ float sum(const float* col) { return MixTree<LO, N / 2>::sum(col) + MixTree<LO + N / 2, N / 2>::sum(col); }
};
template <int LO>
struct MixTree<LO, 1>
{
  static __device__ __forceinline__ float sum(const float* col) { return col[LO * kMixRow]; }
};
__global__ __launch_bounds__(256) void mixdown_stage1_kernel(SignalView sig, size_t V, size_t T, const float* gains, float* partial, uint32_t flags)
{
  apply_fp_mode(flags);
  __shared__ float tile[4][64 * kMixRow];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t groups = (V + 63) / 64, item = (size_t)blockIdx.x * 4 + wave;  // item = group * T + t
  if (item >= groups * T) return;
  const size_t group = item / T, t = item - group * T;
  const size_t v = group * 64 + lane;
  const bool live = v < V;
  const float g = (live && gains) ? gains[v] : 1.f;
  float* row = tile[wave] + lane * kMixRow;
  typedef float f32x4m __attribute__((ext_vector_type(4)));
  const f32x4m* src = (const f32x4m*)sig.base + t * sig.strideT + v * sig.strideV;
  f32x4m x[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) x[q] = live ? __builtin_nontemporal_load(src + q * sig.strideQ) : f32x4m{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < 16; ++q)
  {
    f32x4m y = x[q];
    if (gains)
    {
      y[0] *= g;
      y[1] *= g;
      y[2] *= g;
      y[3] *= g;
    }
    *(f32x4m*)(row + 4 * q) = y;
  }
  // (the tile is this wavefront's own: its lanes run in lockstep, the LDS operations of a wavefront complete in order)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const float total = MixTree<0, 64>::sum(tile[wave] + lane);
  partial[(group * T + t) * 64 + lane] = total;
}

// Rows of partial sums, 64 consecutive ones at a time, left to right: rows [64 r, 64 r + 64) of `in` -> row r of `out`. Applied
// until one row is left (64 groups = 4096 voices per row after the first pass, 262 144 after the second, ...): the single serial
// chain over ALL groups that this replaces took 16 384 dependent additions per sample for 2^20 voices - 480 us of a 64-frame
// block whose voice kernel takes 62 (profiles/r04_rt_1M.json).
__global__ __launch_bounds__(64) void mixdown_rows64_kernel(const float4* in, size_t rows, size_t nQuads, float4* out, uint32_t flags)
{
  apply_fp_mode(flags);
  const size_t qi = (size_t)blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
  if (qi >= nQuads) return;
  const size_t first = r * 64, n = (rows - first < 64) ? rows - first : 64;
  const float4* p = in + first * nQuads + qi;
  float4 acc = p[0];
  size_t g = 1;
  for (; g + 16 <= n; g += 16)
  {
    float4 x[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) x[u] = p[(g + u) * nQuads];
#pragma unroll
    for (int u = 0; u < 16; ++u)
    {
      acc.x = acc.x + x[u].x;
      acc.y = acc.y + x[u].y;
      acc.z = acc.z + x[u].z;
      acc.w = acc.w + x[u].w;
    }
  }
  for (; g < n; ++g)
  {
    const float4 x = p[g * nQuads];
    acc.x = acc.x + x.x;
    acc.y = acc.y + x.y;
    acc.z = acc.z + x.z;
    acc.w = acc.w + x.w;
  }
  out[r * nQuads + qi] = acc;
}
// The last two of those passes in one launch, for up to 4096 rows: thread (q, r) of a workgroup adds up rows [64 r, 64 r + 64) of its
// quad as above, the sums meet in LDS, and the threads of r = 0 add them left to right. Same order, one launch less per mixdown
// (2^20 voices: 16 384 -> 256 by the kernel above, 256 -> 4 -> 1 here).
__global__ __launch_bounds__(1024) void mixdown_rows_last2_kernel(const float4* in, size_t rows, size_t nQuads, float4* out, uint32_t flags)
{
  apply_fp_mode(flags);
  __shared__ float4 mid[64][16];
  const size_t qi = (size_t)blockIdx.x * 16 + threadIdx.x, r = threadIdx.y;
  const size_t first = r * 64, n = (rows - first < 64) ? rows - first : 64;
  if (qi < nQuads)
  {
    const float4* p = in + first * nQuads + qi;
    float4 acc = p[0];
    size_t g = 1;
    for (; g + 16 <= n; g += 16)
    {
      float4 x[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) x[u] = p[(g + u) * nQuads];
#pragma unroll
      for (int u = 0; u < 16; ++u)
      {
        acc.x = acc.x + x[u].x;
        acc.y = acc.y + x[u].y;
        acc.z = acc.z + x[u].z;
        acc.w = acc.w + x[u].w;
      }
    }
    for (; g < n; ++g)
    {
      const float4 x = p[g * nQuads];
      acc.x = acc.x + x.x;
      acc.y = acc.y + x.y;
      acc.z = acc.z + x.z;
      acc.w = acc.w + x.w;
    }
    mid[r][threadIdx.x] = acc;
  }
  __syncthreads();
  if (r != 0 || qi >= nQuads) return;
  float4 acc = mid[0][threadIdx.x];
  for (unsigned m = 1; m < blockDim.y; ++m)
  {
    const float4 x = mid[m][threadIdx.x];
    acc.x = acc.x + x.x;
    acc.y = acc.y + x.y;
    acc.z = acc.z + x.z;
    acc.w = acc.w + x.w;
  }
  out[qi] = acc;
}
}  // namespace

// grouped mixdown: out[g] = ((0 + v[g*P]) + v[g*P + 1]) + ... — the voices of one instrument summed in voice order, exactly
// the `outputs[c] += ...` accumulation of Synth::processVector (source/app/MLSynth.h:43-57). One lane per (group, quad).
namespace
{
// A wavefront owns 64 consecutive groups of one quad (t, q): it loads their 64 * P float4 with P fully coalesced instructions
// (lane l takes elements l, l + 64, ...), parks them in its own LDS strip, and lane l then adds up the P voices of group l in
// voice order. One spare float4 per group keeps the strided reads of the second phase off a single LDS bank. (The first
// version let every lane read its own group straight from memory: 16-byte accesses at a stride of P * 16 bytes, 1.1 TB/s.)
__global__ __launch_bounds__(256) void mixdown_groups_kernel(SignalView sig, SignalView out, size_t groups, size_t P, size_t T, uint32_t flags)
{
  apply_fp_mode(flags);
  extern __shared__ float4 mixLds[];
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float4* strip = mixLds + (size_t)wave * 64 * (P + 1);
  const size_t groupBlocks = (groups + 63) / 64;
  const size_t items = T * 16 * groupBlocks;  // (quad, block of 64 groups), the group block fastest
  const size_t V = groups * P;
  const size_t wavesPerBlock = blockDim.x >> 6;
  for (size_t item = (size_t)blockIdx.x * wavesPerBlock + wave; item < items; item += (size_t)gridDim.x * wavesPerBlock)
  {
    const size_t gb = item % groupBlocks, qi = item / groupBlocks;
    const size_t t = qi >> 4, q = qi & 15;
    const size_t firstVoice = gb * 64 * P;
    const float4* src = sig.base + t * sig.strideT + q * sig.strideQ;
    size_t i = 0;
    for (; i + 4 <= P; i += 4)  // four loads in flight per lane, then their LDS stores
    {
      float4 x[4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
      {
        const size_t e = (i + u) * 64 + lane;
        x[u] = (firstVoice + e < V) ? src[(firstVoice + e) * sig.strideV] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
      {
        const size_t e = (i + u) * 64 + lane;
        strip[e + e / P] = x[u];
      }
    }
    for (; i < P; ++i)
    {
      const size_t e = i * 64 + lane;
      if (firstVoice + e < V) strip[e + e / P] = src[(firstVoice + e) * sig.strideV];
    }
    // the strip is private to this wavefront and LDS operations of one wavefront complete in order
    const size_t g = gb * 64 + lane;
    if (g < groups)
    {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      const float4* mine = strip + (size_t)lane * (P + 1);
      for (size_t p = 0; p < P; ++p)
      {
        const float4 x = mine[p];
        acc.x = acc.x + x.x;
        acc.y = acc.y + x.y;
        acc.z = acc.z + x.z;
        acc.w = acc.w + x.w;
      }
      out.base[t * out.strideT + q * out.strideQ + g * out.strideV] = acc;
    }
  }
}
// groups too large for an LDS strip: one lane per (group, quad), straight from memory
__global__ __launch_bounds__(256) void mixdown_groups_direct_kernel(SignalView sig, SignalView out, size_t groups, size_t P, size_t T, uint32_t flags)
{
  apply_fp_mode(flags);
  const size_t total = groups * T * 16;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride)
  {
    const size_t g = i % groups, qi = i / groups;
    const size_t t = qi >> 4, q = qi & 15;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (size_t p = 0; p < P; ++p)
    {
      const float4 x = sig.base[t * sig.strideT + q * sig.strideQ + (g * P + p) * sig.strideV];
      acc.x = acc.x + x.x;
      acc.y = acc.y + x.y;
      acc.z = acc.z + x.z;
      acc.w = acc.w + x.w;
    }
    out.base[t * out.strideT + q * out.strideQ + g * out.strideV] = acc;
  }
}
}  // namespace

hipError_t mlgpu_launch_mixdown_groups(const float* sig, int layout, size_t groups, size_t P, size_t T, float* out, int outLayout, hipStream_t stream, uint32_t flags)
{
  const SignalView in = makeView(sig, layout, groups * P, T), ov = makeView(out, outLayout, groups, T);
  const size_t stripBytes = 64 * (P + 1) * sizeof(float4);  // per wavefront
  if (stripBytes > 64 * 1024)
  {
    size_t blocks = (groups * T * 16 + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(mixdown_groups_direct_kernel, dim3((unsigned)(blocks ? blocks : 1)), dim3(256), 0, stream, in, ov, groups, P, T, flags);
    return hipGetLastError();
  }
  size_t waves = 4;  // per workgroup, as many as fit 64 KiB of LDS
  while (waves > 1 && waves * stripBytes > 64 * 1024) waves >>= 1;
  const size_t items = T * 16 * ((groups + 63) / 64);
  size_t blocks = (items + waves - 1) / waves;
  if (blocks > 256 * 8) blocks = 256 * 8;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(mixdown_groups_kernel, dim3((unsigned)blocks), dim3((unsigned)(64 * waves)), waves * stripBytes, stream, in, ov, groups, P, T, flags);
  return hipGetLastError();
}

hipError_t mlgpu_launch_mixdown(const float* sig, int layout, size_t V, size_t T, const float* gains, float* partial, float* out,
                                hipStream_t stream, uint32_t flags)
{
  const size_t groups = (V + 63) / 64;
  hipLaunchKernelGGL(mixdown_stage1_kernel, dim3((unsigned)((groups * T + 3) / 4)), dim3(256), 0, stream, makeView(sig, layout, V, T), V, T, gains, partial, flags);
  return mlgpu_launch_mixdown_rows(groups, T, partial, out, stream, flags);
}

hipError_t mlgpu_launch_mixdown_stage1(const float* sig, int layout, size_t V, size_t T, const float* gains, float* partial, hipStream_t stream, uint32_t flags)
{
  const size_t groups = (V + 63) / 64;
  hipLaunchKernelGGL(mixdown_stage1_kernel, dim3((unsigned)((groups * T + 3) / 4)), dim3(256), 0, stream, makeView(sig, layout, V, T), V, T, gains, partial, flags);
  return hipGetLastError();
}

// the later stages alone: `partial` holds the rows of 64-voice group sums (the first stage's output, or a voice kernel's that made
// them itself - chain_mix_kernel)
hipError_t mlgpu_launch_mixdown_rows(size_t groups, size_t T, float* partial, float* out, hipStream_t stream, uint32_t flags)
{
  const size_t nQuads = T * 16;
  // the rows of group sums (in `partial`), 64 at a time, until one is left; the passes alternate between the two parts of the
  // scratch (mlgpu_mixdown_reserve: the second holds the first pass's rows / 64), the last one writes `out`
  float4* a = (float4*)partial;
  float4* b = a + groups * nQuads;
  size_t rows = groups;
  do
  {
    if (rows <= 4096)
    {
      hipLaunchKernelGGL(mixdown_rows_last2_kernel, dim3((unsigned)((nQuads + 15) / 16)), dim3(16, (unsigned)((rows + 63) / 64)), 0, stream, (const float4*)a, rows,
                         nQuads, (float4*)out, flags);
      break;
    }
    const size_t rowsOut = (rows + 63) / 64;
    hipLaunchKernelGGL(mixdown_rows64_kernel, dim3((unsigned)((nQuads + 63) / 64), (unsigned)rowsOut), dim3(64), 0, stream, (const float4*)a, rows, nQuads,
                       rowsOut == 1 ? (float4*)out : b, flags);
    float4* t = a;
    a = b;
    b = t;
    rows = rowsOut;
  } while (rows > 1);
  return hipGetLastError();
}

// The later stages up to a level: `reductions` passes of 64 rows -> 1 (each exact: the caller guarantees groups is a multiple of
// 64^reductions), the last one into `out` - what one SHARD of a voice bank hands to the host, which finishes the same tree over the
// shards' rows (mlgpu_mixdown_finish). reductions == 0: the group sums themselves.
hipError_t mlgpu_launch_mixdown_rows_partial(size_t groups, size_t T, float* partial, float* out, int reductions, hipStream_t stream, uint32_t flags)
{
  const size_t nQuads = T * 16;
  float4* a = (float4*)partial;
  float4* b = a + groups * nQuads;
  size_t rows = groups;
  if (reductions == 0) return hipMemcpyAsync(out, partial, sizeof(float4) * groups * nQuads, hipMemcpyDeviceToDevice, stream);
  for (int r = 0; r < reductions; ++r)
  {
    const size_t rowsOut = (rows + 63) / 64;
    hipLaunchKernelGGL(mixdown_rows64_kernel, dim3((unsigned)((nQuads + 63) / 64), (unsigned)rowsOut), dim3(64), 0, stream, (const float4*)a, rows, nQuads,
                       r + 1 == reductions ? (float4*)out : b, flags);
    float4* t = a;
    a = b;
    b = t;
    rows = rowsOut;
  }
  return hipGetLastError();
}

static unsigned gridFor(size_t items)
{
  size_t blocks = (items + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}

hipError_t mlgpu_launch_rows_map(int rule, long p0, long p1, int sampleRotate, const float* src, size_t srcRows, float* dst,
                                 size_t dstRows, size_t dstOffset, size_t dstStep, size_t count, size_t groups, hipStream_t stream)
{
  RowsMapArgs a;
  a.src = (const float4*)src;
  a.dst = (float4*)dst;
  a.rule = rule;
  a.p0 = p0;
  a.p1 = p1;
  a.sampleRotate = sampleRotate;
  a.srcRows = srcRows;
  a.dstRows = dstRows;
  a.dstOffset = dstOffset;
  a.dstStep = dstStep;
  a.count = count;
  a.groups = groups;
  hipLaunchKernelGGL(rows_map_kernel, dim3(gridFor(groups * count * 16)), dim3(256), 0, stream, a);
  return hipGetLastError();
}

hipError_t mlgpu_launch_rows_add(const float* rows, size_t rowsPerGroup, float* out, size_t groups, hipStream_t stream, uint32_t flags)
{
  hipLaunchKernelGGL(rows_add_kernel, dim3(gridFor(groups * 16)), dim3(256), 0, stream, (const float4*)rows, (float4*)out, rowsPerGroup, groups, flags);
  return hipGetLastError();
}

hipError_t mlgpu_launch_rows_normalize(const float* rows, float* out, size_t nRows, hipStream_t stream, uint32_t flags)
{
  hipLaunchKernelGGL(rows_normalize_kernel, dim3((unsigned)((nRows + 255) / 256)), dim3(256), 0, stream, (const float4*)rows, (float4*)out, nRows, flags);
  return hipGetLastError();
}

hipError_t mlgpu_launch_rows_index(float* out, size_t rowsPerGroup, size_t groups, hipStream_t stream)
{
  hipLaunchKernelGGL(rows_index_kernel, dim3(gridFor(groups * rowsPerGroup * 16)), dim3(256), 0, stream, (float4*)out, rowsPerGroup, groups);
  return hipGetLastError();
}

hipError_t mlgpu_launch_route(bool demux, bool linear, const float* sel, size_t selElems, const float* const* ins, float* const* outs, int n,
                              size_t nElems, hipStream_t stream, uint32_t flags)
{
  RouteArgs a;
  memset(&a, 0, sizeof(a));
  a.flags = flags;
  a.sel = sel;
  a.selElems = selElems;
  a.n = n;
  a.nElems = nElems;
  if (demux)
  {
    a.in[0] = ins[0];
    for (int k = 0; k < n; ++k) a.out[k] = outs[k];
  }
  else
  {
    for (int k = 0; k < n; ++k) a.in[k] = ins[k];
    a.out[0] = outs[0];
  }
  const dim3 grid(gridFor(nElems));
  if (demux && linear) hipLaunchKernelGGL(demultiplex_kernel<true>, grid, dim3(256), 0, stream, a);
  else if (demux) hipLaunchKernelGGL(demultiplex_kernel<false>, grid, dim3(256), 0, stream, a);
  else if (linear) hipLaunchKernelGGL(multiplex_kernel<true>, grid, dim3(256), 0, stream, a);
  else hipLaunchKernelGGL(multiplex_kernel<false>, grid, dim3(256), 0, stream, a);
  return hipGetLastError();
}
