// device_scan.hip -- kernels behind device_scan.hpp (exclusive prefix sum, gfx950).
#include "device_scan.hpp"

namespace shk {

static_assert(SCAN_ITEMS == 16 && SCAN_THREADS == 256, "the kernels below load four uint4 per thread and combine four waves");

// the 16 items of a thread: vector accesses when the tile is whole and the pointer aligned, else item by item (0 behind n)
template <bool VEC>
__device__ __forceinline__ void load_items(const uint32_t *__restrict__ in, uint64_t base, uint64_t n, uint32_t (&v)[SCAN_ITEMS])
{
  if (VEC && base + SCAN_ITEMS <= n) {
    const uint4 *p = reinterpret_cast<const uint4 *>(in + base);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint4 a = p[q];
      v[4 * q] = a.x; v[4 * q + 1] = a.y; v[4 * q + 2] = a.z; v[4 * q + 3] = a.w;
    }
  } else {
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) v[i] = base + i < n ? in[base + i] : 0u;
  }
}

__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v)
{
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// inclusive scan of one 32-bit value per lane across the wave (modulo 2^32)
__device__ __forceinline__ uint32_t wave_inclusive_scan_u32(uint32_t v)
{
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  return v;
}

template <bool VEC>
__global__ __launch_bounds__(SCAN_THREADS) void scan_tile_sums_kernel(const uint32_t *__restrict__ in, uint64_t n, uint64_t *__restrict__ tile_sums)
{
  __shared__ uint64_t lds[SCAN_THREADS / 64];
  const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
  uint32_t v[SCAN_ITEMS];
  load_items<VEC>(in, base, n, v);
  uint64_t s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) s += v[i];
  s = wave_sum_u64(s);
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = lds[0] + lds[1] + lds[2] + lds[3];
}

// one workgroup of 1024 threads: exclusive scan of the tile sums in place (each thread a run of consecutive tiles); total -> *total_out
constexpr int SCAN_OFFS_THREADS = 1024;
__global__ __launch_bounds__(SCAN_OFFS_THREADS) void scan_tile_offsets_kernel(uint64_t *__restrict__ tile_sums, uint64_t ntiles, uint64_t *__restrict__ total_out)
{
  __shared__ uint64_t lds[SCAN_OFFS_THREADS / 64];
  const uint64_t per = (ntiles + SCAN_OFFS_THREADS - 1) / SCAN_OFFS_THREADS;
  const uint64_t lo = (uint64_t)threadIdx.x * per, hi = lo + per < ntiles ? lo + per : ntiles;
  uint64_t s = 0;
  for (uint64_t i = lo; i < hi; ++i) s += tile_sums[i];
  // inclusive scan of the threads' sums: within the wave, then over the 16 wave totals
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint64_t inc = s;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint64_t t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) lds[wave] = inc;
  __syncthreads();
  uint64_t add = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < SCAN_OFFS_THREADS / 64; ++w) {
    const uint64_t t = lds[w];
    if (w < wave) add += t;
    tot += t;
  }
  uint64_t run = add + inc - s;
  for (uint64_t i = lo; i < hi; ++i) {
    const uint64_t t = tile_sums[i];
    tile_sums[i] = run;
    run += t;
  }
  if (threadIdx.x == 0) *total_out = tot;
}

template <bool VEC>
__global__ __launch_bounds__(SCAN_THREADS) void scan_apply_kernel(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, uint64_t n,
                                                                 const uint64_t *__restrict__ tile_offs)
{
  __shared__ uint32_t lds[SCAN_THREADS / 64];
  const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
  uint32_t v[SCAN_ITEMS];
  load_items<VEC>(in, base, n, v);
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) s += v[i];
  // (the outputs are 32-bit: everything below is modulo 2^32; the 64-bit total comes from the tile sums)
  const uint32_t inc = wave_inclusive_scan_u32(s);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 63) lds[wave] = inc;
  __syncthreads();
  uint32_t add = 0;
#pragma unroll
  for (int w = 0; w < SCAN_THREADS / 64; ++w) add += w < wave ? lds[w] : 0u;
  uint32_t run = (uint32_t)tile_offs[blockIdx.x] + add + inc - s;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    const uint32_t t = v[i];
    v[i] = run;
    run += t;
  }
  if (VEC && base + SCAN_ITEMS <= n) {
    uint4 *p = reinterpret_cast<uint4 *>(out + base);
#pragma unroll
    for (int q = 0; q < 4; ++q) p[q] = make_uint4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
  } else {
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i)
      if (base + i < n) out[base + i] = v[i];
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
  const bool vec = ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15u) == 0;
  if (vec) hipLaunchKernelGGL(scan_tile_sums_kernel<true>, dim3((unsigned)ntiles), dim3(SCAN_THREADS), 0, stream, in, n, temp);
  else hipLaunchKernelGGL(scan_tile_sums_kernel<false>, dim3((unsigned)ntiles), dim3(SCAN_THREADS), 0, stream, in, n, temp);
  hipLaunchKernelGGL(scan_tile_offsets_kernel, dim3(1), dim3(SCAN_OFFS_THREADS), 0, stream, temp, ntiles, total);
  if (vec) hipLaunchKernelGGL(scan_apply_kernel<true>, dim3((unsigned)ntiles), dim3(SCAN_THREADS), 0, stream, in, out, n, temp);
  else hipLaunchKernelGGL(scan_apply_kernel<false>, dim3((unsigned)ntiles), dim3(SCAN_THREADS), 0, stream, in, out, n, temp);
  return total;
}

}  // namespace shk
