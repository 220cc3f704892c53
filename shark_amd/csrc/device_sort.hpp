// device_sort.hpp -- LSD radix sort of uint64 keys on gfx950 (8-bit digits, stable).
//
// Used by the index build to order the (rank << 16 | gene) pairs of all reference k-mers, which is
// what turns "append gene g to the list of every bit it sets, genes in file order"
// (main.cpp:154-189, bloomfilter.h:61-75) into one data-parallel step.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace shk {

constexpr int RS_THREADS = 256;
constexpr int RS_ITEMS = 16;
constexpr int RS_TILE = RS_THREADS * RS_ITEMS;   // keys per workgroup

// u32 words of histogram/offset storage needed for n keys
inline uint64_t radix_sort_hist_words(uint64_t n) { return 256ull * ((n + RS_TILE - 1) / RS_TILE); }

// Sorts the low `end_bit` bits of n keys (n < 2^32).  `a` holds the input; `b` is a same-sized
// buffer; `hist` has radix_sort_hist_words(n) u32; `scan_tmp` as for exclusive_scan_u32 over that
// many words.  Returns the buffer that holds the sorted keys (a or b), or nullptr on launch error.
uint64_t *radix_sort_u64(uint64_t *a, uint64_t *b, uint64_t n, unsigned end_bit, uint32_t *hist, uint64_t *scan_tmp, hipStream_t stream);

}  // namespace shk
