// device_scan.hpp -- exclusive prefix sum of uint32 arrays on gfx950.
//
// Three launches (tile sums -> scan of the tile sums in one workgroup -> per-tile scan with the
// tile's offset); 16 contiguous items per thread as four 16-byte accesses, wave64 shuffles inside
// a workgroup.  Streaming, bandwidth-bound: 2 reads + 1 write of the array (10 M items: 30 + 10 +
// 45 us; the first version -- 8 items per thread, dword accesses, 64-bit shuffles -- took 290).  Used by the rank
// directory (bloomfilter.h:121 init_support(_brank)), the gene-list CSR
// (bloomfilter.h:142-167) and the result offsets of classify.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace shk {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 16;                     // per thread
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;  // 4096

// temp: (ntiles + 1) uint64_t; the grand total lands in temp[ntiles].
inline uint64_t scan_temp_words(uint64_t n) { return (n + SCAN_TILE - 1) / SCAN_TILE + 1; }

// exclusive scan in -> out (may alias); returns the device pointer holding the 64-bit total
const uint64_t *exclusive_scan_u32(const uint32_t *in, uint32_t *out, uint64_t n, uint64_t *temp, hipStream_t stream);

}  // namespace shk
