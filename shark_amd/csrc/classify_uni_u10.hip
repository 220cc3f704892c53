// classify_uni_kernel<10, ...>: every probe mode / quality / LDS variant of this unroll (classify_uni.hpp) -- pairs of up to 640 slots,
// two staging groups per lane (2 x 300 bp)
#include "classify_uni.hpp"

namespace shk {
void launch_uni_u10(const ClassifyParams &p, int mode, bool hasq, bool big, bool lx, int rmode, unsigned grid, hipStream_t s)
{
  launch_uni_u<10>(p, mode, hasq, big, lx, rmode, grid, s);
}
}  // namespace shk
