// classify_uni_kernel<5, ...>: every probe mode / quality / LDS variant of this unroll (classify_uni.hpp)
#include "classify_uni.hpp"

namespace shk {
void launch_uni_u5(const ClassifyParams &p, int mode, bool hasq, bool big, bool lx, int rmode, unsigned grid, hipStream_t s)
{
  launch_uni_u<5>(p, mode, hasq, big, lx, rmode, grid, s);
}
}  // namespace shk

#if SHK_STAMPS
// (diagnostic build only) the per-phase clock sums of the three-pairs kernel since the last reset, summed over the waves: out[16]
extern "C" int shk_debug_read_stamps(unsigned long long *out, int reset)
{
  static unsigned long long rows[4096 * 16];
  if (hipMemcpyFromSymbol(rows, HIP_SYMBOL(shk::shk_stamp_acc), sizeof(rows)) != hipSuccess) return -1;
  for (int i = 0; i < 16; ++i) out[i] = 0;
  for (int w = 0; w < 4096; ++w)
    for (int i = 0; i < 16; ++i) out[i] += rows[16 * w + i];
  if (reset) {
    for (auto &x : rows) x = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(shk::shk_stamp_acc), rows, sizeof(rows)) != hipSuccess) return -1;
  }
  return 0;
}

// ... and the 100 MHz counter at the start and the end of every wave's loop in the last launch: out[2 * 4096]
extern "C" int shk_debug_read_wave_times(unsigned long long *out)
{
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(shk::shk_stamp_waves), sizeof(unsigned long long) * 2 * 4096) == hipSuccess ? 0 : -1;
}
#endif
