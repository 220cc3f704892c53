// classify_uni_kernel<5, ...>: every probe mode / quality / LDS variant of this unroll (classify_uni.hpp)
#include "classify_uni.hpp"

namespace shk {
void launch_uni_u5(const ClassifyParams &p, int mode, bool hasq, bool big, bool lx, int rmode, unsigned grid, hipStream_t s)
{
  launch_uni_u<5>(p, mode, hasq, big, lx, rmode, grid, s);
}
}  // namespace shk

#if SHK_STAMPS
// (diagnostic build only) the per-phase clock sums of the three-pairs kernel since the last reset: out[16]
extern "C" int shk_debug_read_stamps(unsigned long long *out, int reset)
{
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(shk::shk_stamp_acc), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(shk::shk_stamp_acc), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}

// ... and the 100 MHz counter at the start and the end of every wave's loop in the last launch: out[2 * 4096]
extern "C" int shk_debug_read_wave_times(unsigned long long *out)
{
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(shk::shk_stamp_waves), sizeof(unsigned long long) * 2 * 4096) == hipSuccess ? 0 : -1;
}
#endif
