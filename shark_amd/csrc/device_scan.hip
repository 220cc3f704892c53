// device_scan.hip -- kernels behind device_scan.hpp (exclusive prefix sum, gfx950).
#include "device_scan.hpp"

namespace shk {

// inclusive scan of one value per thread across a 256-thread workgroup;
// returns the inclusive prefix, *total receives the workgroup total
__device__ __forceinline__ uint64_t block_inclusive_scan_u64(uint64_t v, uint64_t *total, uint64_t *lds /* >= 4 */)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint64_t t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  if (lane == 63) lds[wave] = v;
  __syncthreads();
  uint64_t add = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < SCAN_THREADS / 64; ++w) {
    const uint64_t s = lds[w];
    if (w < wave) add += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return v + add;
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_tile_sums_kernel(const uint32_t *__restrict__ in, uint64_t n, uint64_t *__restrict__ tile_sums)
{
  __shared__ uint64_t lds[4];
  const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
  uint64_t s = 0;
  if (base + SCAN_ITEMS <= n) {
    const uint4 a = *reinterpret_cast<const uint4 *>(in + base);
    const uint4 b = *reinterpret_cast<const uint4 *>(in + base + 4);
    s = (uint64_t)a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
  } else {
    for (int i = 0; i < SCAN_ITEMS; ++i)
      if (base + i < n) s += in[base + i];
  }
  uint64_t tot;
  block_inclusive_scan_u64(s, &tot, lds);
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = tot;
}

// one workgroup: exclusive scan of the tile sums in place; total -> *total_out
__global__ __launch_bounds__(SCAN_THREADS) void scan_tile_offsets_kernel(uint64_t *__restrict__ tile_sums, uint64_t ntiles, uint64_t *__restrict__ total_out)
{
  __shared__ uint64_t lds[4];
  uint64_t carry = 0;
  for (uint64_t b = 0; b < ntiles; b += SCAN_THREADS) {
    const uint64_t i = b + threadIdx.x;
    const uint64_t v = i < ntiles ? tile_sums[i] : 0;
    uint64_t tot;
    const uint64_t inc = block_inclusive_scan_u64(v, &tot, lds);
    if (i < ntiles) tile_sums[i] = carry + inc - v;
    carry += tot;
  }
  if (threadIdx.x == 0) *total_out = carry;
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_apply_kernel(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, uint64_t n,
                                                                 const uint64_t *__restrict__ tile_offs)
{
  __shared__ uint64_t lds[4];
  const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
  uint32_t v[SCAN_ITEMS];
  uint64_t s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    v[i] = base + i < n ? in[base + i] : 0u;
    s += v[i];
  }
  uint64_t tot;
  const uint64_t inc = block_inclusive_scan_u64(s, &tot, lds);
  uint64_t run = tile_offs[blockIdx.x] + inc - s;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    if (base + i < n) out[base + i] = (uint32_t)run;
    run += v[i];
  }
}

// exclusive scan in -> out (may alias); returns the device pointer holding the 64-bit total
const uint64_t *exclusive_scan_u32(const uint32_t *in, uint32_t *out, uint64_t n, uint64_t *temp, hipStream_t stream)
{
  const uint64_t ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
  uint64_t *total = temp + ntiles;
  if (n == 0) {
    (void)hipMemsetAsync(total, 0, sizeof(uint64_t), stream);
    return total;
  }
  hipLaunchKernelGGL(scan_tile_sums_kernel, dim3((unsigned)ntiles), dim3(SCAN_THREADS), 0, stream, in, n, temp);
  hipLaunchKernelGGL(scan_tile_offsets_kernel, dim3(1), dim3(SCAN_THREADS), 0, stream, temp, ntiles, total);
  hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)ntiles), dim3(SCAN_THREADS), 0, stream, in, out, n, temp);
  return total;
}

}  // namespace shk
