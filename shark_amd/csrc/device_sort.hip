// device_sort.hip -- kernels behind device_sort.hpp.
//
// One pass per 8-bit digit: (1) per-workgroup digit histograms, stored digit-major; (2) exclusive
// scan of the whole histogram array = first output position of every (digit, workgroup);
// (3) each workgroup re-reads its tile and scatters every key to base[digit][workgroup] + its rank
// among the earlier keys of the tile with the same digit.  The rank is built from wave ballots
// (eight ballots give the set of lanes with the same digit) plus a per-wave count table in LDS, so
// the scatter is stable and needs no atomics on global memory.
#include "device_sort.hpp"

#include "device_scan.hpp"

namespace shk {

__global__ __launch_bounds__(RS_THREADS) void rs_histogram_kernel(const uint64_t *__restrict__ keys, uint64_t n, unsigned shift,
                                                                  uint32_t *__restrict__ hist, uint32_t n_blocks)
{
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const uint64_t base = (uint64_t)blockIdx.x * RS_TILE;
#pragma unroll 4
  for (int r = 0; r < RS_ITEMS; ++r) {
    const uint64_t i = base + (uint64_t)r * RS_THREADS + threadIdx.x;
    if (i < n) atomicAdd(&h[(keys[i] >> shift) & 255u], 1u);
  }
  __syncthreads();
  hist[(uint64_t)threadIdx.x * n_blocks + blockIdx.x] = h[threadIdx.x];
}

__global__ __launch_bounds__(RS_THREADS) void rs_scatter_kernel(const uint64_t *__restrict__ keys, uint64_t *__restrict__ out, uint64_t n, unsigned shift,
                                                                const uint32_t *__restrict__ offs, uint32_t n_blocks)
{
  __shared__ uint32_t run[256];            // next output position per digit for this workgroup
  __shared__ uint32_t wcnt[RS_THREADS / 64][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  run[threadIdx.x] = offs[(uint64_t)threadIdx.x * n_blocks + blockIdx.x];
  const uint64_t base = (uint64_t)blockIdx.x * RS_TILE;
  for (int r = 0; r < RS_ITEMS; ++r) {
    for (int i = threadIdx.x; i < (RS_THREADS / 64) * 256; i += RS_THREADS) (&wcnt[0][0])[i] = 0;
    __syncthreads();
    const uint64_t i = base + (uint64_t)r * RS_THREADS + threadIdx.x;
    const bool valid = i < n;
    const uint64_t key = valid ? keys[i] : 0ull;
    const uint32_t digit = (uint32_t)(key >> shift) & 255u;
    // lanes of this wave holding the same digit
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = (digit >> b) & 1u;
      const unsigned long long m = __ballot(bit);
      peers &= bit ? m : ~m;
    }
    const uint32_t rank_in_wave = (uint32_t)__builtin_popcountll(peers & ((1ull << lane) - 1ull));
    const uint32_t cnt_in_wave = (uint32_t)__builtin_popcountll(peers);
    if (valid && rank_in_wave == 0) wcnt[wave][digit] = cnt_in_wave;
    __syncthreads();
    if (valid) {
      uint32_t o = run[digit] + rank_in_wave;
      for (int w = 0; w < wave; ++w) o += wcnt[w][digit];
      out[o] = key;
    }
    __syncthreads();
    if (valid && rank_in_wave == 0) atomicAdd(&run[digit], cnt_in_wave);
    __syncthreads();
  }
}

uint64_t *radix_sort_u64(uint64_t *a, uint64_t *b, uint64_t n, unsigned end_bit, uint32_t *hist, uint64_t *scan_tmp, hipStream_t stream)
{
  if (n == 0) return a;
  const uint32_t n_blocks = (uint32_t)((n + RS_TILE - 1) / RS_TILE);
  uint64_t *src = a, *dst = b;
  for (unsigned shift = 0; shift < end_bit; shift += 8) {
    hipLaunchKernelGGL(rs_histogram_kernel, dim3(n_blocks), dim3(RS_THREADS), 0, stream, (const uint64_t *)src, n, shift, hist, n_blocks);
    exclusive_scan_u32(hist, hist, 256ull * n_blocks, scan_tmp, stream);
    hipLaunchKernelGGL(rs_scatter_kernel, dim3(n_blocks), dim3(RS_THREADS), 0, stream, (const uint64_t *)src, dst, n, shift, (const uint32_t *)hist, n_blocks);
    if (hipGetLastError() != hipSuccess) return nullptr;
    uint64_t *t = src; src = dst; dst = t;
  }
  return src;
}

}  // namespace shk
