// measure.hip -- measurement entry points of libsharkhip that are on no product path.
//
// shk_measure_random_lookups: what this part sustains for INDEPENDENT random 16-byte lookups in a table of a given size --
// the access pattern of the position table (one 16-byte bucket per k-mer at a hashed address).  On an index far
// beyond the caches every such lookup is one memory-side request (a 128-byte line, of which 16 bytes are used), and their
// rate -- not the bytes -- is what bounds the classify kernel there; bench.py measures this ceiling in the run whose
// fraction of it it reports.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "shark_internal.hpp"

namespace shk {

__device__ __forceinline__ uint64_t mix64(uint64_t x)
{
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
  return x;
}

typedef uint32_t m_u32x4 __attribute__((ext_vector_type(4)));

// five lookups in flight per lane (what the classify kernel has per round of 2 x 150 bp); NT = streaming loads
template <bool NT>
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
  }
  if (acc == 0x12345678u) out[tid & 1023] = acc;   // keeps the loads alive
}

}  // namespace shk

using namespace shk;

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
  auto launch = [&](uint32_t it) {
    if (nontemporal) hipLaunchKernelGGL(random_lookup_kernel<true>, dim3(grid), dim3(256), 0, ctx->stream, tab, mask16, it, out);
    else hipLaunchKernelGGL(random_lookup_kernel<false>, dim3(grid), dim3(256), 0, ctx->stream, tab, mask16, it, out);
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
