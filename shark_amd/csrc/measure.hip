// measure.hip -- measurement entry points of libsharkhip that are on no product path.
//
// shk_measure_random_lookups: what this part sustains for INDEPENDENT random 16-byte lookups in a table of a given size --
// the access pattern of the position table (one 16-byte bucket per k-mer at a hashed address).  On an index far
// beyond the caches every such lookup is one memory-side request (a 128-byte line, of which 16 bytes are used), and their
// rate -- not the bytes -- is what bounds the classify kernel there; bench.py measures this ceiling in the run whose
// fraction of it it reports.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "kmer_device.hpp"
#include "shark_internal.hpp"

namespace shk {

__device__ __forceinline__ uint64_t mix64(uint64_t x)
{
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
  return x;
}

typedef uint32_t m_u32x4 __attribute__((ext_vector_type(4)));

// five lookups in flight per lane (what the classify kernel has per round of 2 x 150 bp); NT = streaming loads.
// BOTH: every lookup also reads 16 bytes of the OTHER 64-byte half of its 128-byte line (address ^ 64) -- the calibration of what a
// random lookup moves: if the memory side fetches whole 128-byte lines the second read is served by the first's line and rate and
// request count stay what they are for one read per line; if it fetches 64-byte halves on demand both double their cost
template <bool NT, bool BOTH>
__global__ __launch_bounds__(256) void random_lookup_kernel(const uint8_t *__restrict__ tab, uint64_t mask16, uint32_t iters, uint32_t *__restrict__ out)
{
  constexpr int U = 5;
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  uint64_t x = mix64(tid + 1);
  for (uint32_t it = 0; it < iters; ++it) {
    uint64_t a[U];
    m_u32x4 v[U];
#pragma unroll
    for (int j = 0; j < U; ++j) { x = mix64(x + j + 1); a[j] = (x & mask16) << 4; }
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const m_u32x4 *p = reinterpret_cast<const m_u32x4 *>(tab + a[j]);
      v[j] = NT ? __builtin_nontemporal_load(p) : *p;
    }
#pragma unroll
    for (int j = 0; j < U; ++j) acc += v[j].x ^ v[j].w;
    if (BOTH) {
      m_u32x4 u[U];
#pragma unroll
      for (int j = 0; j < U; ++j) {
        const m_u32x4 *p = reinterpret_cast<const m_u32x4 *>(tab + (a[j] ^ 64ull));
        u[j] = NT ? __builtin_nontemporal_load(p) : *p;
      }
#pragma unroll
      for (int j = 0; j < U; ++j) acc += u[j].y ^ u[j].z;
    }
  }
  if (acc == 0x12345678u) out[tid & 1023] = acc;   // keeps the loads alive
}

// shk_measure_valu_mix: the ISSUE ceiling of the exact-table classify kernel's own instruction mix.  One iteration is the
// arithmetic that kernel does for a read on its shortest way through -- stage eight bases (classify4 / pack4 / gather4, the two code
// streams), cut a slot's two windows out of six dwords, canonical form, XXH64, the exact table's address arithmetic and compare,
// the validity window and the coverage step -- on REGISTER operands: no LDS, no memory, nothing to wait for but the VALU itself;
// every lane carries its own dependent chain, as in the kernel, and W waves per SIMD interleave.  bench.py divides the kernel's
// measured VALU rate by this kernel's (instructions from the same counter pass, time from here): how far the classify kernel is
// from what the SIMDs can issue of THIS mix -- 15 of XXH64's 38 instructions are quarter-rate multiplies -- rather than from the
// 2-cycle peak no integer code reaches.
__global__ __launch_bounds__(256) void valu_mix_kernel(const uint32_t iters, const uint32_t k, uint32_t *__restrict__ out)
{
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t d0 = tid * 2654435761u + 0x41434754u, d1 = d0 ^ 0x47544341u, d2 = d1 + 0x54474341u;
  const uint64_t kmer_mask = (1ull << (2u * k)) - 1ull, kmask0 = (1ull << k) - 1ull;
  const uint32_t tagmask = (1u << 18) - 1u, gmask2 = ((1u << 13) - 1u) << 1;
  uint32_t acc = 0;
  // the clock the SIMDs hold while this mix runs (the chip lowers it under load): shader cycles (s_memtime) over the constant 100 MHz
  // counter (s_memrealtime) around the loop, summed over the waves -- out[1024..1027] as two 64-bit sums nothing else reads
  const uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (uint32_t it = 0; it < iters; ++it) {
    // stage (classify_uni.hpp, "stage")
    const uint32_t sh = d0 & 3u;
    const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, sh), hi = __builtin_amdgcn_alignbyte(d2, d1, sh);
    uint32_t c_lo, c_hi, i_lo, i_hi;
    classify4(lo, c_lo, i_lo);
    classify4(hi, c_hi, i_hi);
    const uint32_t msb16 = (pack4(c_lo) << 8) | pack4(c_hi);
    const uint32_t inv8 = gather4(i_lo) | (gather4(i_hi) << 4);
    uint32_t lsb = __builtin_bitreverse32(msb16);
    lsb = ((lsb >> 1) & 0x55555555u) | ((lsb & 0x55555555u) << 1);
    // a slot's windows out of six dwords (`windows`)
    const uint32_t f0 = lsb ^ d0, f1 = lsb + d1, f2 = msb16 ^ d2, e0 = msb16 + d0, e1 = lsb ^ d2, e2 = msb16 ^ d1;
    const uint32_t sf = (lane & 15u) << 1, sr = ((lane * 7u + it) & 15u) << 1;
    const uint64_t x = ((uint64_t)__builtin_amdgcn_alignbit(f2, f1, sf) << 32) | __builtin_amdgcn_alignbit(f1, f0, sf);
    const uint64_t y = ((uint64_t)__builtin_amdgcn_alignbit(e2, e1, sr) << 32) | __builtin_amdgcn_alignbit(e1, e0, sr);
    const uint64_t fwd = y & kmer_mask, rc = ~x & kmer_mask;
    const uint64_t h = xxh64_u64(fwd < rc ? fwd : rc);
    // the exact table's probe (`lx_hit_at`: addresses and compare; the two LDS reads themselves are not VALU work)
    const uint32_t di = ((uint32_t)h >> 14) & gmask2;
    const uint32_t tg = __builtin_amdgcn_alignbit((uint32_t)(h >> 32), (uint32_t)h, 15) & tagmask;
    const uint32_t base = (uint32_t)h + (tg >> 13) * 40503u;
    const uint32_t ti = ((base + di) << 2) & ((32768u - 1u) << 2);
    const uint32_t ee = ti ^ d1;
    const bool hit = (ee >> 13) == ((tg << 1) | 1u);
    // validity window and coverage step of a matched slot (`slot_valid`, `sparse_first`)
    const uint64_t v0 = ((uint64_t)d2 << 32) | inv8, v1 = ((uint64_t)d0 << 32) | lsb;
    const uint32_t vs = lane;
    const uint64_t win = (v0 >> vs) | ((v1 << 1) << (63u - vs));
    const bool valid = (win & kmask0) == kmask0;
    const uint64_t nx = (v1 >> lane) >> 1;
    const uint32_t step = nx ? 2u * ((uint32_t)__builtin_ctzll(nx) + 1u) : k;
    const uint32_t cv = (hit & valid) ? (step < k ? step : k) : 0u;
    acc += cv;
    d0 ^= (uint32_t)h + inv8;
    d1 += hit ? lsb : msb16;
    d2 ^= (uint32_t)(h >> 32) + ti;
  }
  if (acc == 0x12345678u && d0 == d1) out[tid & 1023u] = acc + d2;   // (keeps the chain alive)
  const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (lane == 0) {
    unsigned long long *clk = reinterpret_cast<unsigned long long *>(out + 1024);
    atomicAdd(&clk[0], (unsigned long long)(t1 - t0));
    atomicAdd(&clk[1], (unsigned long long)(r1 - r0));
  }
}

}  // namespace shk

using namespace shk;

extern "C" int shk_measure_valu_mix(shk_ctx *cctx, int waves_per_simd, uint32_t iters, double *ms_out, uint64_t *wave_iterations)
{
  return shk_measure_valu_mix_clock(cctx, waves_per_simd, iters, ms_out, wave_iterations, nullptr);
}

extern "C" int shk_measure_valu_mix_clock(shk_ctx *cctx, int waves_per_simd, uint32_t iters, double *ms_out, uint64_t *wave_iterations, double *shader_ghz)
{
  Ctx *ctx = cctx;
  if (!ctx || !ms_out || !wave_iterations || waves_per_simd < 1 || waves_per_simd > 8 || iters == 0) return SHK_ERR_ARG;
  SHK_HIP(ctx, hipSetDevice(ctx->prm.device));
  hipDeviceProp_t prop;
  SHK_HIP(ctx, hipGetDeviceProperties(&prop, ctx->prm.device));
  uint32_t *out = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  auto done = [&](int r) {
    if (out) (void)hipFree(out);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    return r;
  };
#define MS_HIP(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) return done(set_hip_error(ctx, e__, #call)); } while (0)
  MS_HIP(hipMalloc((void **)&out, 4096 + 64));
  MS_HIP(hipEventCreate(&e0));
  MS_HIP(hipEventCreate(&e1));
  // one 256-thread workgroup puts a wave on each of a CU's four SIMDs: W workgroups per CU are W waves per SIMD, all resident at once
  const unsigned grid = (unsigned)prop.multiProcessorCount * (unsigned)waves_per_simd;
  hipLaunchKernelGGL(valu_mix_kernel, dim3(grid), dim3(256), 0, ctx->stream, 64u, ctx->prm.k, out);
  MS_HIP(hipGetLastError());
  MS_HIP(hipMemsetAsync(out + 1024, 0, 64, ctx->stream));
  MS_HIP(hipEventRecord(e0, ctx->stream));
  hipLaunchKernelGGL(valu_mix_kernel, dim3(grid), dim3(256), 0, ctx->stream, iters, ctx->prm.k, out);
  MS_HIP(hipGetLastError());
  MS_HIP(hipEventRecord(e1, ctx->stream));
  MS_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  MS_HIP(hipEventElapsedTime(&ms, e0, e1));
  if (shader_ghz) {
    unsigned long long clk[2] = {0, 0};
    MS_HIP(hipMemcpy(clk, out + 1024, sizeof(clk), hipMemcpyDeviceToHost));
    *shader_ghz = clk[1] ? 0.1 * (double)clk[0] / (double)clk[1] : 0.0;   // (s_memrealtime: 100 MHz)
  }
#undef MS_HIP
  *ms_out = ms;
  *wave_iterations = (uint64_t)grid * 4ull * iters;
  return done(SHK_OK);
}

extern "C" int shk_measure_random_lookups(shk_ctx *cctx, uint64_t table_bytes, uint64_t n_lookups, int nontemporal, double *g_lookups_per_s)
{
  Ctx *ctx = cctx;
  if (!ctx || !g_lookups_per_s || table_bytes < (1ull << 20) || (table_bytes & (table_bytes - 1)) || n_lookups == 0) return SHK_ERR_ARG;
  SHK_HIP(ctx, hipSetDevice(ctx->prm.device));
  uint8_t *tab = nullptr;
  uint32_t *out = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int rc = SHK_OK;
  auto done = [&](int r) {
    if (tab) (void)hipFree(tab);
    if (out) (void)hipFree(out);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    return r;
  };
#define MS_HIP(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) return done(set_hip_error(ctx, e__, #call)); } while (0)
  MS_HIP(hipMalloc((void **)&tab, table_bytes));
  MS_HIP(hipMalloc((void **)&out, 4096));
  MS_HIP(hipMemsetAsync(tab, 1, table_bytes, ctx->stream));
  MS_HIP(hipEventCreate(&e0));
  MS_HIP(hipEventCreate(&e1));
  const unsigned grid = 256u * 8u;                       // 8 waves per SIMD
  const uint64_t per_iter = (uint64_t)grid * 256 * 5;
  const uint32_t iters = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(n_lookups / per_iter, 1u << 20));
  const uint64_t mask16 = table_bytes / 16 - 1;
  const bool nt = (nontemporal & 1) != 0, both = (nontemporal & 2) != 0;   // (bit 1: both halves of every line, see random_lookup_kernel)
  auto launch = [&](uint32_t it) {
    if (both) {
      if (nt) hipLaunchKernelGGL((random_lookup_kernel<true, true>), dim3(grid), dim3(256), 0, ctx->stream, tab, mask16, it, out);
      else hipLaunchKernelGGL((random_lookup_kernel<false, true>), dim3(grid), dim3(256), 0, ctx->stream, tab, mask16, it, out);
    } else {
      if (nt) hipLaunchKernelGGL((random_lookup_kernel<true, false>), dim3(grid), dim3(256), 0, ctx->stream, tab, mask16, it, out);
      else hipLaunchKernelGGL((random_lookup_kernel<false, false>), dim3(grid), dim3(256), 0, ctx->stream, tab, mask16, it, out);
    }
  };
  launch(8);                                             // warm-up (page tables, clocks)
  MS_HIP(hipGetLastError());
  MS_HIP(hipEventRecord(e0, ctx->stream));
  launch(iters);
  MS_HIP(hipGetLastError());
  MS_HIP(hipEventRecord(e1, ctx->stream));
  MS_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  MS_HIP(hipEventElapsedTime(&ms, e0, e1));
#undef MS_HIP
  *g_lookups_per_s = ms > 0.f ? (double)per_iter * iters / (ms * 1e-3) / 1e9 : 0.0;
  return done(rc);
}
