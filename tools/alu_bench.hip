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
    if (OP == 1) { a = a * (b | 1u); b = b * (c | 1u); c = c * (d | 1u); d = d * (a | 1u); }   // v_mul_lo_u32 (+v_or) x4
    if (OP == 2) { a = __umulhi(a, 0x85EBCA87u) + 1; b = __umulhi(b, 0x27D4EB4Fu) + 1; c = __umulhi(c, 0x9E3779F9u) + 1; d = __umulhi(d, 0xC2B2AE63u) + 1; }
    if (OP == 3) { x = (uint64_t)(uint32_t)x * 0x85EBCA87u + y; y = (uint64_t)(uint32_t)y * 0x27D4EB4Fu + x; }  // v_mad_u64_u32 x2
    if (OP == 4) { a = __umul24(a, 0x5BCA87) ; b = __umul24(b, 0x54EB4F); c = __umul24(c, 0x3779F9); d = __umul24(d, 0xB2AE63); a ^= c; b ^= d; }
    if (OP == 5) { x = x * 0xC2B2AE3D27D4EB4Full + y; y = y * 0x9E3779B185EBCA87ull + x; }   // 2 full 64-bit multiplies (+2 add64)
    if (OP == 6) { x = shk::xxh64_u64(x); }
    if (OP == 7) { x = (x << 13) | (x >> 51); y = (y << 7) ^ x; }
    if (OP == 8) { a = __builtin_amdgcn_alignbit(a, b, c); b = __builtin_amdgcn_alignbit(b, c, d); c = __builtin_amdgcn_alignbit(c, d, a); d = __builtin_amdgcn_alignbit(d, a, b); }
    if (OP == 9) { x = x << (a & 31); y = y >> (b & 31); x ^= 1; y ^= 0x8000000000000000ull; }   // 2 variable 64-bit shifts + 2 (4x32) xor
    if (OP == 10) { a = __builtin_amdgcn_perm(a, b, 0x07020500u); b = __builtin_amdgcn_perm(b, c, 0x07020500u); c = __builtin_amdgcn_perm(c, d, 0x07020500u); d = __builtin_amdgcn_perm(d, a, 0x07020500u); }
    if (OP == 11) { x = x < y ? x + 1 : y + 3; y ^= 0x5555; }    // 64-bit compare + selects
    if (OP == 12) { a = __builtin_bitreverse32(a) + 1; b = __builtin_bitreverse32(b) + 1; c = __builtin_bitreverse32(c) + 1; d = __builtin_bitreverse32(d) + 1; }
    if (OP == 13) { a += 1; b += 3; a = (a & 0x55555555u) | (b & ~0x55555555u); b = (b & 0x33333333u) | (c & ~0x33333333u); c = (c & 0x0f0f0f0fu) | (d & ~0x0f0f0f0fu); d = (d & 0x00ff00ffu) | (a & ~0x00ff00ffu); }   // v_bfi x4
    if (OP == 14) { x = (x << 31) | (x >> 33); y = (y << 27) | (y >> 37); x += 1; y += 1; }   // constant 64-bit rotates
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
  run<8>("v_alignbit_b32 x4", 4); run<9>("var 64-bit shift x2 (+4 xor)", 2); run<10>("v_perm_b32 x4", 4); run<11>("u64 cmp+select group", 1);
  run<12>("v_bfrev(+add) x4", 4); run<13>("v_bfi x4", 4); run<14>("const rotl64 x2 (+2 add64)", 2);
  return 0;
}
