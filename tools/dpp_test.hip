// checks the DPP wave reductions against shuffle reductions on random data
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../shark_amd/csrc/kmer_device.hpp"
__global__ void k(const uint32_t* in, uint32_t* out_min, uint32_t* out_sum) {
  uint32_t v = in[blockIdx.x * 64 + threadIdx.x];
  uint32_t a = shk::wave_min_u32(v), b = shk::wave_sum_u32(v & 0xFFFF);
  uint32_t m = v, s = v & 0xFFFF;
  for (int o = 32; o > 0; o >>= 1) { uint32_t t = __shfl_xor(m, o, 64); m = t < m ? t : m; s += __shfl_xor(s, o, 64); }
  if (threadIdx.x == 0) { out_min[blockIdx.x] = (a == m); out_sum[blockIdx.x] = (b == s); }
}
int main() {
  const int nb = 4096; uint32_t *h = (uint32_t*)malloc(nb*64*4);
  for (int i = 0; i < nb*64; ++i) h[i] = (uint32_t)rand() * 2654435761u;
  uint32_t *d, *m, *s; hipMalloc(&d, nb*64*4); hipMalloc(&m, nb*4); hipMalloc(&s, nb*4);
  hipMemcpy(d, h, nb*64*4, hipMemcpyHostToDevice);
  k<<<nb, 64>>>(d, m, s);
  uint32_t *hm = (uint32_t*)malloc(nb*4), *hs = (uint32_t*)malloc(nb*4);
  hipMemcpy(hm, m, nb*4, hipMemcpyDeviceToHost); hipMemcpy(hs, s, nb*4, hipMemcpyDeviceToHost);
  int okm = 0, oks = 0; for (int i = 0; i < nb; ++i) { okm += hm[i]; oks += hs[i]; }
  printf("min ok %d/%d sum ok %d/%d\n", okm, nb, oks, nb);
  return 0;
}
