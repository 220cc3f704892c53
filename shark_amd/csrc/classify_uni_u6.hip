// classify_uni_kernel<6, ...>: every probe mode / quality / LDS variant of this unroll (classify_uni.hpp)
#include "classify_uni.hpp"

namespace shk {
void launch_uni_u6(const ClassifyParams &p, int mode, bool hasq, bool big, bool lx, int rmode, unsigned grid, hipStream_t s)
{
  launch_uni_u<6>(p, mode, hasq, big, lx, rmode, grid, s);
}
}  // namespace shk
