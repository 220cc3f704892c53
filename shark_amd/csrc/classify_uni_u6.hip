// classify_uni_kernel<6, ...>: every probe mode / quality / LDS variant of this unroll (classify_uni.hpp)
#include "classify_uni.hpp"

namespace shk {
void launch_uni_u6(const ClassifyParams &p, int mode, bool hasq, bool big, bool lx, bool uni, unsigned grid, hipStream_t s)
{
  launch_uni_u<6>(p, mode, hasq, big, lx, uni, grid, s);
}
}  // namespace shk
