// tools/fetchcal.hip — calibrates rocprofv3's FETCH_SIZE on the access patterns the delay rings use. The guide
// (MI355X_MICROARCH.md, HBM) says FETCH_SIZE reports half the bytes of a wide coalesced 16 B / lane stream on gfx950 and
// that other widths are uncalibrated; the windowed rings read one 32-byte sector per lane at a per-lane position. Every
// kernel here reads a KNOWN number of bytes exactly once (1 GiB, far beyond the caches):
//   stream     : grid-stride float4, 16 B / lane, the calibrated case
//   sector32   : layout [chunk][lane][8 floats]; lane l of a 256-lane block reads ITS sector of chunk perm_l(k), k = 0..: 2 x 16 B
//   sector64   : layout [chunk][lane][16 floats]: 4 x 16 B per visit
//   sector32eq : as sector32 but every lane visits the chunks in the same order (equal delay times: the block's sectors of
//                a chunk are one contiguous 8 KiB piece)
// Run:  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- tools/bin/fetchcal   (then tools/pmc_summary.py pmc out)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ __launch_bounds__(256) void stream(const f32x4* a, float* sink, size_t n)
{
  const size_t stride = (size_t)gridDim.x * 256;
  f32x4 acc = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) acc += __builtin_nontemporal_load(&a[i]);
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) *sink = acc.x;
}

// one 256-lane block owns `chunks` chunks of SECTOR floats per lane; lane l visits chunk (k * stepOf(l) + l) mod chunks
template <int SECTOR, bool EQUAL>
__global__ __launch_bounds__(256) void sectors(const f32x4* a, float* sink, size_t chunks)
{
  const size_t blockBase = (size_t)blockIdx.x * chunks * 256 * (SECTOR / 4);   // in float4
  const unsigned l = threadIdx.x;
  const size_t step = EQUAL ? 1 : (2 * (size_t)l + 1);     // odd: a permutation of the chunks when `chunks` is a power of two
  f32x4 acc = {0, 0, 0, 0};
  for (size_t k = 0; k < chunks; ++k)
  {
    const size_t c = (k * step + (EQUAL ? 0 : l * 7)) & (chunks - 1);
    const f32x4* p = a + blockBase + (c * 256 + l) * (SECTOR / 4);
#pragma unroll
    for (int j = 0; j < SECTOR / 4; ++j) acc += p[j];
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) *sink = acc.x;
}

int main()
{
  const size_t bytes = (size_t)1 << 30, n = bytes / 16;
  f32x4* a; float* sink;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(a, 0, bytes));
  const int blocks = 1024;
  for (int rep = 0; rep < 4; ++rep)
  {
    hipLaunchKernelGGL(stream, dim3(2048), dim3(256), 0, 0, a, sink, n);
    hipLaunchKernelGGL((sectors<8, false>), dim3(blocks), dim3(256), 0, 0, a, sink, bytes / blocks / 256 / 32);
    hipLaunchKernelGGL((sectors<16, false>), dim3(blocks), dim3(256), 0, 0, a, sink, bytes / blocks / 256 / 64);
    hipLaunchKernelGGL((sectors<8, true>), dim3(blocks), dim3(256), 0, 0, a, sink, bytes / blocks / 256 / 32);
    CK(hipDeviceSynchronize());
  }
  printf("each kernel read %zu bytes exactly once per launch (4 launches each)\n", bytes);
  return 0;
}
