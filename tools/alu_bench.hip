// VALU throughput microbenchmark on gfx950: cycles per wave-instruction per SIMD for the
// integer ops XXH64 is made of.  Many waves, independent chains.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../shark_amd/csrc/kmer_device.hpp"
#define ITER 4096
template <int OP> __global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed) {
  uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9E3779B1u, c = a + 77, d = b + 99;
  uint64_t x = ((uint64_t)a << 32) | b, y = ((uint64_t)c << 32) | d;
  for (int i = 0; i < ITER; ++i) {
    if (OP == 0) { a += b; b += c; c += d; d += a; }
    if (OP == 1) { a = a * 0x85EBCA87u; b = b * 0x27D4EB4Fu; c = c * 0x9E3779F9u; d = d * 0xC2B2AE63u; }   // v_mul_lo_u32
    if (OP == 2) { a = __umulhi(a, 0x85EBCA87u) + 1; b = __umulhi(b, 0x27D4EB4Fu) + 1; c = __umulhi(c, 0x9E3779F9u) + 1; d = __umulhi(d, 0xC2B2AE63u) + 1; }
    if (OP == 3) { x = (uint64_t)(uint32_t)x * 0x85EBCA87u + y; y = (uint64_t)(uint32_t)y * 0x27D4EB4Fu + x; }  // v_mad_u64_u32 x2
    if (OP == 4) { a = __umul24(a, 0x5BCA87) ; b = __umul24(b, 0x54EB4F); c = __umul24(c, 0x3779F9); d = __umul24(d, 0xB2AE63); a ^= c; b ^= d; }
    if (OP == 5) { x = x * 0xC2B2AE3D27D4EB4Full; y = y * 0x9E3779B185EBCA87ull; }   // 2 full 64-bit multiplies
    if (OP == 6) { x = shk::xxh64_u64(x); }
    if (OP == 7) { x = (x << 13) | (x >> 51); y = (y << 7) ^ x; }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a ^ b ^ c ^ d ^ (uint32_t)x ^ (uint32_t)(x >> 32) ^ (uint32_t)y;
}
template <int OP> void run(const char* name, double ops_per_iter) {
  uint32_t* d; hipMalloc(&d, 8192 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<OP><<<8192, 256>>>(d, 1);
  hipEventRecord(e0); k<OP><<<8192, 256>>>(d, 2); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // wave-instructions per SIMD: 8192 blocks * 4 waves / 1024 SIMDs = 32 waves per SIMD
  double wave_instr_per_simd = 32.0 * ITER * ops_per_iter;
  printf("%-28s %8.3f ms  -> %.2f ns per wave-op per SIMD (%.1f cycles @2.4GHz)\n", name, ms, ms * 1e6 / wave_instr_per_simd, ms * 1e6 / wave_instr_per_simd * 2.4);
  hipFree(d);
}
int main() {
  run<0>("v_add_u32 x4", 4); run<1>("v_mul_lo_u32 x4", 4); run<2>("v_mul_hi_u32(+add) x4", 4); run<3>("v_mad_u64_u32 x2", 2);
  run<4>("v_mul_u32_u24 x4 (+2 xor)", 4); run<5>("u64*u64 x2", 2); run<6>("xxh64_u64 x1", 1); run<7>("64-bit shift/rot group", 1);
  return 0;
}
