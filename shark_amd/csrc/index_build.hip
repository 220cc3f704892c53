// index_build.hip -- on-device construction of shark's k-mer index (gfx950).
//
// Replaces, with identical resulting content,
//   pass 1   KmerBuilder.hpp:40-72 + BloomfilterFiller.hpp:38-46 + BF::add_at
//            bloomfilter.h:57-59                    -> ref_kmer_kernel<MODE_SET>
//   switch_mode(1)  bloomfilter.h:112-125 (rank)    -> bf_word_popcount + scan
//   pass 2   main.cpp:154-189 + BF::add_to_kmer bloomfilter.h:61-75
//                                                   -> ref_kmer_kernel<MODE_KEYS> + radix sort (device_sort.hip)
//   switch_mode(2)  bloomfilter.h:126-184           -> unique + CSR + list-entry kernels
// plus the summary level, which only restates the filter.
//
// The reference rolls k-mers serially and restarts after an invalid
// character; here every base position is an independent thread: the k-mer
// STARTING at position s of a record exists iff s + k <= len and characters
// s..s+k-1 are all valid, which is the same set (kmer_utils.hpp:57-71).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "device_scan.hpp"
#include "device_sort.hpp"
#include "kmer_device.hpp"
#include "shark_internal.hpp"

namespace shk {

enum { MODE_SET = 0, MODE_KEYS = 1 };
constexpr int RK_THREADS = 256;

// to_int (kmer_utils.hpp:29-41) for one byte: 0..3, or 4 = invalid
__device__ __forceinline__ uint32_t base_code(uint32_t c)
{
  const uint32_t t = c & 0xDFu;
  const uint32_t code = ((t >> 1) ^ (t >> 2)) & 3u;
  const uint32_t expect = (0x54474341u >> (8 * code)) & 0xFFu;
  return expect == t ? code : 4u;
}

// one thread per base position of the concatenated reference records
template <int MODE>
__global__ __launch_bounds__(RK_THREADS) void ref_kmer_kernel(const uint8_t *__restrict__ bytes, uint64_t total,
                                                              const uint64_t *__restrict__ rec_off, uint32_t n_rec, uint32_t k,
                                                              uint64_t *__restrict__ bf64, uint64_t bf_bits, uint64_t bf_mask, int pow2,
                                                              uint8_t *__restrict__ rec_has, unsigned long long *__restrict__ n_valid,
                                                              const uint32_t *__restrict__ rank_w, const uint32_t *__restrict__ rec_nidx,
                                                              uint64_t *__restrict__ keys, uint64_t sentinel, int wrap)
{
  __shared__ uint8_t codes[RK_THREADS + 32];
  const uint64_t b0 = (uint64_t)blockIdx.x * RK_THREADS;
  const uint64_t i = b0 + threadIdx.x;
  if (i < total) codes[threadIdx.x] = (uint8_t)base_code(bytes[i]);
  if (threadIdx.x < k - 1) {
    const uint64_t j = b0 + RK_THREADS + threadIdx.x;
    codes[RK_THREADS + threadIdx.x] = j < total ? (uint8_t)base_code(bytes[j]) : (uint8_t)4;
  }
  __syncthreads();
  if (i >= total) return;

  // record containing position i: last r with rec_off[r] <= i
  uint32_t lo = 0, hi = n_rec;  // invariant: rec_off[lo] <= i < rec_off[hi]
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (rec_off[mid] <= i) lo = mid; else hi = mid;
  }
  const uint32_t r = lo;
  const uint64_t s = i - rec_off[r];
  const uint64_t len = rec_off[r + 1] - rec_off[r];
  bool valid = s + k <= len;
  uint64_t fw = 0;
  if (valid) {
    uint32_t bad = 0;
    for (uint32_t j = 0; j < k; ++j) {
      const uint32_t c = codes[threadIdx.x + j];
      bad |= c >> 2;
      fw = (fw << 2) | (c & 3u);
    }
    valid = bad == 0;
  }
  if (MODE == MODE_SET) {
    // the number of valid reference k-mers, one atomic per wave (1.8 x 10^8 same-address atomics per build otherwise)
    const unsigned long long vm = __ballot(valid);
    if ((threadIdx.x & 63u) == 0u && vm) atomicAdd(n_valid, (unsigned long long)__builtin_popcountll(vm));
  }
  if (!valid) {
    if (MODE == MODE_KEYS) keys[i] = sentinel;
    return;
  }
  const uint64_t canon = canonical_from_top(fw << (64 - 2 * k), k);
  const uint64_t h = xxh64_u64(canon);
  const uint64_t pos = pow2 ? (h & bf_mask) : (h % bf_bits);
  if (MODE == MODE_SET) {
    // BF::add_at, bloomfilter.h:57-59 (idempotent => the filter is order independent)
    atomicOr(reinterpret_cast<uint32_t *>(bf64) + (pos >> 5), 1u << (pos & 31));
    rec_has[r] = 1;                                 // record owns >= 1 valid k-mer (main.cpp:165)
  } else {
    // bloomfilter.h:70: kmer_rank = _brank(bf_idx); the list entry is (rank, gene)
    const uint32_t rk = bf_rank(rank_w, bf64[pos >> 6], pos);
    const uint32_t g = rec_nidx[r];
    // normal: (rank, gene).  More than 65 536 genes (wrap): (rank, gene & 0xFFFF, came-from-a-gene-above-65535) -- see build_index
    keys[i] = wrap ? (((uint64_t)rk << 17) | ((uint64_t)(g & 0xFFFFu) << 1) | (g > 0xFFFFu ? 1u : 0u)) : (((uint64_t)rk << 16) | (uint64_t)g);
  }
}

// popcount of each 64-bit filter word (two words per lane, 16-byte loads)
__global__ __launch_bounds__(256) void bf_word_popcount_kernel(const uint4 *__restrict__ bf, uint64_t n_vec, uint32_t *__restrict__ counts)
{
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_vec) return;
  const uint4 v = bf[g];
  uint2 c;
  c.x = __builtin_popcount(v.x) + __builtin_popcount(v.y);
  c.y = __builtin_popcount(v.z) + __builtin_popcount(v.w);
  reinterpret_cast<uint2 *>(counts)[g] = c;
}

// summary level: bit j = OR of filter bits [j<<shift, (j+1)<<shift); only non-empty words write
__global__ __launch_bounds__(256) void bf_summary_kernel(const uint64_t *__restrict__ bf64, uint64_t n_words, uint32_t shift, uint32_t *__restrict__ sum32)
{
  const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n_words) return;
  if (bf64[w] != 0ull) {
    const uint64_t sb = (w << 6) >> shift;
    atomicOr(&sum32[sb >> 5], 1u << (sb & 31));
  }
}

// sorted keys -> head flags (first occurrence of each distinct (rank, gene))
// (wrap: an entry of a gene above 65535 is never dropped -- bloomfilter.h:72 compares the uint16_t last() with the int index)
__global__ __launch_bounds__(256) void unique_flags_kernel(const uint64_t *__restrict__ keys, uint64_t n, uint64_t sentinel, int wrap, uint32_t *__restrict__ flags)
{
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t key = keys[i];
  flags[i] = (key < sentinel && (i == 0 || keys[i - 1] != key || (wrap && (key & 1ull)))) ? 1u : 0u;
}

// write the CSR: ids[o] for every kept key, offsets[r] at the first key of rank r
__global__ __launch_bounds__(256) void csr_write_kernel(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ idx, uint64_t n, uint64_t sentinel,
                                                        int wrap, uint32_t *__restrict__ csr_off, uint16_t *__restrict__ csr_ids)
{
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t key = keys[i];
  if (key >= sentinel) return;
  const bool head = i == 0 || keys[i - 1] != key || (wrap && (key & 1ull));
  if (!head) return;
  const uint32_t o = idx[i];
  const uint32_t sh = wrap ? 17u : 16u;
  csr_ids[o] = (uint16_t)((key >> (wrap ? 1 : 0)) & 0xFFFFu);
  const uint64_t r = key >> sh;
  if (i == 0 || (keys[i - 1] >> sh) != r) csr_off[r] = o;
}

// list entries: {start, len (clipped), first gene} per set bit
__global__ __launch_bounds__(256) void list_entry_kernel(const uint32_t *__restrict__ csr_off, const uint16_t *__restrict__ ids, uint64_t n_set, uint32_t tot,
                                                         ListEntry *__restrict__ ent)
{
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r > n_set) return;
  ListEntry e;
  if (r == n_set) {
    e.start = tot; e.len = 0; e.gene0 = 0;
  } else {
    const uint32_t s0 = csr_off[r], s1 = csr_off[r + 1];
    e.start = s0;
    e.len = (uint16_t)((s1 - s0) > 0xFFFFu ? 0xFFFFu : (s1 - s0));
    e.gene0 = ids[s0];
  }
  ent[r] = e;
}

// position table: one thread per 64-bit filter word inserts its set bits.
// slot claim by 64-bit CAS; an entry sits in the first bucket with a free slot
// on its probe path, so an empty slot ends every unsuccessful search.
__global__ __launch_bounds__(256) void table_build_kernel(const uint64_t *__restrict__ bf64, uint64_t n_words, const uint32_t *__restrict__ rank_w,
                                                          const ListEntry *__restrict__ ent, unsigned long long *__restrict__ tab, uint32_t tab_lg,
                                                          uint32_t *__restrict__ fail)
{
  const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n_words) return;
  uint64_t word = bf64[w];
  if (word == 0ull) return;
  uint32_t r = rank_w[w];
  const uint64_t bmask = (1ull << tab_lg) - 1ull;
  while (word) {
    const uint32_t b = (uint32_t)__builtin_ctzll(word);
    word &= word - 1ull;
    const uint64_t pos = (w << 6) | b;
    const ListEntry le = ent[r];
    const bool multi = le.len != 1;
    // slot: tag(24) | valid (bit 39) | displacement (6) in the high word -- the word the kernel compares --
    // and multi (bit 31) | overflow (bit 30, slot 0 of a bucket, see kmer_device.hpp) | rank or gene in the low word
    const uint64_t base = ((pos >> tab_lg) << 40) | (1ull << 39) | ((uint64_t)multi << 31) | (uint64_t)(multi ? r : (uint32_t)le.gene0);
    const uint64_t home = pos & bmask;
    uint64_t bkt = home;
    bool done = false;
    for (uint32_t d = 0; d < 64 && !done; ++d) {
      const unsigned long long e = base | ((unsigned long long)d << 32);
      for (int sidx = 0; sidx < 2 && !done; ++sidx)
        done = atomicCAS(&tab[2 * bkt + sidx], 0ull, e) == 0ull;
      // placed behind its home bucket: mark that bucket (both its slots are taken and final, so the OR races with nothing)
      if (done && d) atomicOr(&tab[2 * home], (unsigned long long)TAB_OVERFLOW);
      bkt = (bkt + 1) & bmask;
    }
    if (!done) atomicAdd(fail, 1u);
    ++r;
  }
}

// ---- the reference as the table kernels' second way to a k-mer's list (anchored extension, classify_uni.hpp) ----
// 2-bit codes of the concatenated records, 16 bases per dword, first base in the low bits (one thread per dword)
__global__ __launch_bounds__(256) void ref_pack2_kernel(const uint8_t *__restrict__ bytes, uint64_t total, uint32_t *__restrict__ ref2, uint64_t n_dwords)
{
  const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n_dwords) return;
  uint32_t v = 0;
  const uint64_t b = w << 4;
  for (uint32_t j = 0; j < 16; ++j) {
    const uint64_t i = b + j;
    const uint32_t c = i < total ? base_code(bytes[i]) : 0u;
    v |= (c & 3u) << (2 * j);
  }
  ref2[w] = v;
}

// one thread per base position x: the k-mer starting there is looked up in the position table exactly as the classify kernels
// look it up (home bucket, then its probe path); refpay[x] = the slot's low word, and the slot learns x as an occurrence of its
// key (the smallest x | strand << 31 over the occurrences, so the index is the same whatever order the threads run in)
__global__ __launch_bounds__(RK_THREADS) void ref_anchor_kernel(const uint8_t *__restrict__ bytes, uint64_t total, const uint64_t *__restrict__ rec_off,
                                                                uint32_t n_rec, uint32_t k, uint64_t bf_bits, uint64_t bf_mask, int pow2,
                                                                const uint64_t *__restrict__ tab, uint32_t tab_lg, uint32_t *__restrict__ refpay,
                                                                uint32_t *__restrict__ anchor, uint32_t *__restrict__ lost)
{
  __shared__ uint8_t codes[RK_THREADS + 32];
  const uint64_t b0 = (uint64_t)blockIdx.x * RK_THREADS;
  const uint64_t i = b0 + threadIdx.x;
  if (i < total) codes[threadIdx.x] = (uint8_t)base_code(bytes[i]);
  if (threadIdx.x < k - 1) {
    const uint64_t j = b0 + RK_THREADS + threadIdx.x;
    codes[RK_THREADS + threadIdx.x] = j < total ? (uint8_t)base_code(bytes[j]) : (uint8_t)4;
  }
  __syncthreads();
  if (i >= total) return;
  uint32_t lo = 0, hi = n_rec;  // invariant: rec_off[lo] <= i < rec_off[hi]
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (rec_off[mid] <= i) lo = mid; else hi = mid;
  }
  const uint64_t s = i - rec_off[lo];
  const uint64_t len = rec_off[lo + 1] - rec_off[lo];
  bool valid = s + k <= len;
  uint64_t fw = 0;
  if (valid) {
    uint32_t bad = 0;
    for (uint32_t j = 0; j < k; ++j) {
      const uint32_t c = codes[threadIdx.x + j];
      bad |= c >> 2;
      fw = (fw << 2) | (c & 3u);
    }
    valid = bad == 0;
  }
  if (!valid) { refpay[i] = REFPAY_NONE; return; }
  const uint64_t rc = revcomp_left_aligned(fw << (64 - 2 * k), k);
  const bool stored_rc = !(fw < rc);                  // canonical = fw < rc ? fw : rc (KmerBuilder.hpp:49)
  const uint64_t h = xxh64_u64(stored_rc ? rc : fw);
  const uint64_t pos = pow2 ? (h & bf_mask) : (h % bf_bits);
  const uint64_t bmask = (1ull << tab_lg) - 1ull;
  const uint32_t want = ((uint32_t)(pos >> tab_lg) << 8) | 0x80u;
  uint64_t bkt = pos & bmask;
  for (uint32_t d = 0; d < 64; ++d) {
    for (uint32_t sidx = 0; sidx < 2; ++sidx) {
      const uint64_t e = tab[2 * bkt + sidx];
      if ((uint32_t)(e >> 32) == (want | d)) {
        refpay[i] = (uint32_t)e & ~TAB_OVERFLOW;
        atomicMin(&anchor[2 * bkt + sidx], (uint32_t)i | (stored_rc ? 0x80000000u : 0u));
        return;
      }
    }
    bkt = (bkt + 1) & bmask;
  }
  refpay[i] = REFPAY_NONE;   // (cannot happen: every set bit is a key of the table)
  atomicAdd(lost, 1u);
}

// the position table with a slot's occurrence in the reference in place of its list (DeviceIndex::atab)
__global__ void atab_fill_kernel(const uint64_t *__restrict__ tab, const uint32_t *__restrict__ anchor, uint64_t *__restrict__ atab, uint64_t n_slots)
{
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_slots; i += (uint64_t)gridDim.x * blockDim.x)
    atab[i] = (tab[i] & 0xFFFFFFFF00000000ull) | anchor[i];
}

// Multi-gene lists in REFERENCE order.  A probe that matches a multi-gene list gets its rank r and then reads ent[r] and ids[...] at
// addresses that have nothing to do with each other from one k-mer of a read to the next.  For the slots the anchored extension
// settles, the same lists are kept a second time, laid out along the reference: entry n_set + 1 + x of `ent` belongs to the k-mer at
// reference position x (its `start` indexes the copy of its list behind ids[tot_idx)), and refpay[x] carries that entry's index as
// its payload -- the vote's loads of consecutive slots then fall on consecutive addresses.
// pass 1: list length per position (0 unless the k-mer at x has a multi-gene list)
__global__ __launch_bounds__(256) void ref_multi_len_kernel(const uint32_t *__restrict__ refpay, uint64_t total, const ListEntry *__restrict__ ent,
                                                            uint32_t *__restrict__ lens)
{
  const uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (x > total) return;
  uint32_t len = 0;
  if (x < total) {
    const uint32_t lp = refpay[x];
    if (lp != REFPAY_NONE && (lp >> 31)) {
      const uint32_t r = lp & TAB_PAYLOAD;
      const ListEntry e = ent[r];
      len = e.len != 0xFFFFu ? (uint32_t)e.len : ent[r + 1].start - e.start;
    }
  }
  lens[x] = len;
}

// pass 2: the per-position entries and the copies of the lists; refpay[x] of a multi-gene k-mer now names its per-position entry
__global__ __launch_bounds__(256) void ref_multi_write_kernel(uint32_t *__restrict__ refpay, uint64_t total, ListEntry *__restrict__ ent, uint64_t n_set,
                                                              uint16_t *__restrict__ ids, uint32_t tot_idx, const uint32_t *__restrict__ offs)
{
  const uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (x > total) return;
  ListEntry out;
  out.start = tot_idx + offs[x];
  out.len = 0;
  out.gene0 = 0;
  if (x < total) {
    const uint32_t lp = refpay[x];
    if (lp != REFPAY_NONE && (lp >> 31)) {
      const uint32_t r = lp & TAB_PAYLOAD;
      const ListEntry e = ent[r];
      const uint32_t len = e.len != 0xFFFFu ? (uint32_t)e.len : ent[r + 1].start - e.start;
      for (uint32_t i = 0; i < len; ++i) ids[out.start + i] = ids[e.start + i];
      out.len = (uint16_t)(len > 0xFFFFu ? 0xFFFFu : len);
      out.gene0 = e.gene0;
      refpay[x] = 0x80000000u | (uint32_t)(n_set + 1 + x);
    }
  }
  ent[n_set + 1 + x] = out;
}

// What anchor_verdict_kernel needs to know about the surroundings of an anchor (DeviceIndex::refext, refmul).
// refext[x], x a position where a valid k-mer starts: g | left << 16 | right << 24 -- g = the gene of x's RECORD, left / right = how
// many positions directly in front of / behind x start a valid k-mer too (clipped at REFEXT_CLIP).  Such a run never leaves its
// record (k >= 2: the last k - 1 positions of a record start no k-mer), so the list under every k-mer of the run contains g; g is
// read off the nearest position of the run whose list is {g} alone (a single-gene list under a k-mer of g's record is {g}).
// REFEXT_NONE: no valid k-mer starts at x, or no single-gene list within reach.
// One thread per position, neighbours from refpay (2 x 254 coalesced, cache-resident loads at most).
__global__ __launch_bounds__(256) void ref_extent_kernel(const uint32_t *__restrict__ refpay, uint64_t total, uint32_t *__restrict__ refext)
{
  const uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= total) return;
  const uint32_t v = refpay[x];
  if (v == REFPAY_NONE) { refext[x] = REFEXT_NONE; return; }
  constexpr uint32_t NF = 0xFFFFFFFFu;
  auto single = [](const uint32_t u) { return (u >> 31) == 0u && u <= 0xFFFFu; };
  uint32_t gene = single(v) ? v : NF, left = 0, right = 0;
  while (left < REFEXT_CLIP && x > left) {
    const uint32_t u = refpay[x - 1 - left];
    if (u == REFPAY_NONE) break;
    if (gene == NF && single(u)) gene = u;
    ++left;
  }
  while (right < REFEXT_CLIP && x + 1 + right < total) {
    const uint32_t u = refpay[x + 1 + right];
    if (u == REFPAY_NONE) break;
    if (gene == NF && single(u)) gene = u;
    ++right;
  }
  refext[x] = gene == NF ? REFEXT_NONE : (gene | (left << 16) | (right << 24));
}

// refmul: one bit per position, 32 positions per dword, LSB first: 1 = the list under the k-mer that starts there is NOT a single-gene
// list (several genes -- a stretch that genes share, or a collision in the filter --, or no valid k-mer at all); positions behind the
// reference read as 1.  One thread per dword.
__global__ __launch_bounds__(256) void ref_multi_bits_kernel(const uint32_t *__restrict__ refpay, uint64_t total, uint32_t *__restrict__ refmul, uint64_t n_dwords)
{
  const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n_dwords) return;
  uint32_t v = 0;
  for (uint32_t j = 0; j < 32; ++j) {
    const uint64_t x = (w << 5) + j;
    const uint32_t u = x < total ? refpay[x] : REFPAY_NONE;
    v |= ((u >> 31) != 0u || u > 0xFFFFu ? 1u : 0u) << j;
  }
  refmul[w] = v;
}

// ---- the k-mer keyed, minimiser-bucketed table (DeviceIndex::ktab) ----
// Every canonical k-mer x (4^k / 2 of them) is asked what the reference asks of it: is bit XXH64(x) mod size set
// (bloomfilter.h:87-89)?  The k-mers for which it is -- the reference's own and the filter's false positives alike -- become
// the keys; a key's payload is the low word of the position table's slot for that bit, i.e. exactly what a probe of the k-mer
// through `tab` returns.  Nothing in the classify kernels can then tell the two tables apart but the addresses they touch.
// A key sits in the first bucket with a free slot on its path home, home + one line, home + two lines, ... (64-bit CAS);
// a key placed behind its home bucket marks that bucket, as in `tab`.
__global__ __launch_bounds__(256) void ktab_build_kernel(const uint32_t k, const uint32_t w, const uint64_t *__restrict__ bf64, const uint64_t bf_mask,
                                                         const uint64_t *__restrict__ tab, const uint32_t tab_lg, unsigned long long *__restrict__ ktab,
                                                         const uint32_t ktab_lg, unsigned long long *__restrict__ n_keys, uint32_t *__restrict__ fail,
                                                         uint32_t *__restrict__ lost)
{
  const uint64_t n = 1ull << (2u * k);
  const uint64_t tmask = (1ull << tab_lg) - 1ull;
  const uint32_t kmask = (uint32_t)((1ull << ktab_lg) - 1ull);
  unsigned long long mine = 0;
  for (uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; x < n; x += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t rc = revcomp_left_aligned(x << (64u - 2u * k), k);
    if (x > rc) continue;                                    // (its reverse complement is the canonical form: enumerated there)
    const uint64_t pos = xxh64_u64(x) & bf_mask;
    if (!((bf64[pos >> 6] >> (pos & 63u)) & 1ull)) continue;
    // what a probe of that position returns (the position table holds every set bit)
    const uint32_t pwant = ((uint32_t)(pos >> tab_lg) << 8) | 0x80u;
    uint64_t pb = pos & tmask;
    uint32_t lo = 0;
    bool have = false;
    for (uint32_t d = 0; d < 64 && !have; ++d) {
      for (uint32_t sidx = 0; sidx < 2 && !have; ++sidx) {
        const uint64_t e = tab[2 * pb + sidx];
        if ((uint32_t)(e >> 32) == (pwant | d)) { lo = (uint32_t)e & ~TAB_OVERFLOW; have = true; }
      }
      pb = (pb + 1) & tmask;
    }
    if (!have) { atomicAdd(lost, 1u); continue; }
    uint32_t home, want;
    ktab_home(x, rc, k, w, ktab_lg - 3u, home, want);
    const unsigned long long e = ((unsigned long long)want << 32) | lo;
    uint32_t bkt = home;
    bool done = false;
    for (uint32_t d = 0; d < 64 && !done; ++d) {
      for (int sidx = 0; sidx < 2 && !done; ++sidx)
        done = atomicCAS(&ktab[2ull * bkt + sidx], 0ull, e) == 0ull;
      if (done && d) { atomicOr(&ktab[2ull * home], (unsigned long long)TAB_OVERFLOW); atomicAdd(fail + 2, 1u); atomicMax(fail + 3, d); }
      bkt = (bkt + 8u) & kmask;                              // the same bucket of the next line
    }
    if (!done) atomicAdd(fail, 1u);
    ++mine;
  }
  if (mine) atomicAdd(n_keys, mine);
}

// LDS-resident exact table of a tiny index (lds_table.hpp): the keys are read back from the position table just built.
// false = no displacement fits some group (the caller keeps the LDS-summary chain).
// *one_gene: the payload all keys share (every probe that matches answers with that single-gene list), 0xFFFFFFFF when they differ
static bool build_lds_table(const std::vector<uint64_t> &tab, uint32_t tab_lg, std::vector<uint32_t> &img, uint32_t *mul, uint32_t *one_gene)
{
  std::vector<LtabKey> keys;
  const uint64_t n_buckets = 1ull << tab_lg;
  for (uint64_t b = 0; b < n_buckets; ++b)
    for (int sidx = 0; sidx < 2; ++sidx) {
      const uint64_t e = tab[2 * b + sidx];
      const uint32_t hi = (uint32_t)(e >> 32), lo = (uint32_t)e;
      if (!(hi & 0x80u)) continue;
      const uint64_t home = (b - (hi & 0x3Fu)) & (n_buckets - 1);
      const uint32_t gene = lo & TAB_PAYLOAD;
      keys.push_back(LtabKey{((uint64_t)(hi >> 8) << tab_lg) | home, ((lo >> 31) || gene >= LTAB_ESC) ? LTAB_ESC : gene});
    }
  *one_gene = 0xFFFFFFFFu;
  if (!keys.empty() && keys[0].payload != LTAB_ESC) {
    *one_gene = keys[0].payload;
    for (const LtabKey &key : keys)
      if (key.payload != keys[0].payload) { *one_gene = 0xFFFFFFFFu; break; }
  }
  return ltab_build(keys, img, mul);
}

static unsigned grid_for(uint64_t n, unsigned threads) { return (unsigned)((n + threads - 1) / threads); }

int build_index(Ctx *ctx)
{
  DeviceIndex &ix = ctx->idx;
  hipStream_t st = ctx->stream;
  const uint32_t k = ctx->prm.k;
  const uint64_t total = ctx->ref_bytes.size();
  const uint32_t n_rec = (uint32_t)ctx->n_records;

  // ---- rank directory needs to exist even for an empty reference ----------
  uint8_t *d_bytes = nullptr;
  uint64_t *d_rec_off = nullptr;
  uint8_t *d_rec_has = nullptr;
  unsigned long long *d_n_valid = nullptr;
  uint32_t *d_rec_nidx = nullptr;
  uint64_t *d_keys = nullptr, *d_keys_alt = nullptr;
  uint32_t *d_flags = nullptr;
  void *d_sort_tmp = nullptr;
  uint64_t *d_scan_tmp = nullptr;
  uint32_t *d_csr_off = nullptr;
  int rc = SHK_OK;

  auto cleanup = [&]() {
    (void)hipFree(d_bytes); (void)hipFree(d_rec_off); (void)hipFree(d_rec_has); (void)hipFree(d_n_valid); (void)hipFree(d_rec_nidx);
    (void)hipFree(d_keys); (void)hipFree(d_keys_alt); (void)hipFree(d_flags); (void)hipFree(d_sort_tmp); (void)hipFree(d_scan_tmp); (void)hipFree(d_csr_off);
  };
#define BI_HIP(call)                                                   \
  do {                                                                 \
    hipError_t e__ = (call);                                           \
    if (e__ != hipSuccess) { rc = set_hip_error(ctx, e__, #call); cleanup(); return rc; } \
  } while (0)

  BI_HIP(hipMalloc((void **)&d_n_valid, sizeof(unsigned long long)));
  BI_HIP(hipMemsetAsync(d_n_valid, 0, sizeof(unsigned long long), st));
  std::vector<uint8_t> h_has(n_rec ? n_rec : 1, 0);

  if (total > 0) {
    BI_HIP(hipMalloc((void **)&d_bytes, total));
    BI_HIP(hipMemcpyAsync(d_bytes, ctx->ref_bytes.data(), total, hipMemcpyHostToDevice, st));
    BI_HIP(hipMalloc((void **)&d_rec_off, (n_rec + 1) * sizeof(uint64_t)));
    BI_HIP(hipMemcpyAsync(d_rec_off, ctx->ref_off.data(), (n_rec + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    BI_HIP(hipMalloc((void **)&d_rec_has, n_rec));
    BI_HIP(hipMemsetAsync(d_rec_has, 0, n_rec, st));

    // ---- pass 1: set bits ---------------------------------------------------
    hipLaunchKernelGGL(ref_kmer_kernel<MODE_SET>, dim3(grid_for(total, RK_THREADS)), dim3(RK_THREADS), 0, st,
                       d_bytes, total, d_rec_off, n_rec, k, ix.bf64, ix.bf_bits, ix.bf_bits - 1, ix.pow2 ? 1 : 0,
                       d_rec_has, d_n_valid, (const uint32_t *)nullptr, (const uint32_t *)nullptr, (uint64_t *)nullptr, 0ull, 0);
    BI_HIP(hipGetLastError());
    BI_HIP(hipMemcpyAsync(h_has.data(), d_rec_has, n_rec, hipMemcpyDeviceToHost, st));
  }

  // ---- switch_mode(1): rank directory -------------------------------------
  const uint64_t n_words = ix.bf_words64;   // multiple of 8
  BI_HIP(hipMalloc((void **)&ix.rank_w, (n_words + 2) * sizeof(uint32_t)));
  BI_HIP(hipMalloc((void **)&d_scan_tmp, scan_temp_words(std::max<uint64_t>(std::max<uint64_t>(n_words + 1, total), 1)) * sizeof(uint64_t)));
  BI_HIP(hipMemsetAsync(ix.rank_w + n_words, 0, 2 * sizeof(uint32_t), st));
  hipLaunchKernelGGL(bf_word_popcount_kernel, dim3(grid_for(n_words / 2, 256)), dim3(256), 0, st,
                     reinterpret_cast<const uint4 *>(ix.bf64), n_words / 2, ix.rank_w);
  BI_HIP(hipGetLastError());
  const uint64_t *d_total = exclusive_scan_u32(ix.rank_w, ix.rank_w, n_words + 1, d_scan_tmp, st);
  uint64_t n_set = 0;
  unsigned long long n_valid = 0;
  BI_HIP(hipMemcpyAsync(&n_set, d_total, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
  BI_HIP(hipMemcpyAsync(&n_valid, d_n_valid, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
  BI_HIP(hipStreamSynchronize(st));
  ix.n_set = n_set;
  ctx->n_ref_kmers = n_valid;
  if (n_set >= (1ull << 31)) { cleanup(); ctx->last_error = "index has >= 2^31 set bits (int kmer_rank, bloomfilter.h:70)"; return SHK_ERR_INDEX_TOO_LARGE; }

  // ---- gene numbering: main.cpp:158-186 -------------------------------------
  // nidx advances for every record EXCEPT one that is at least k long but has
  // no valid k-mer (`continue` at :165 skips `++nidx` at :185).
  std::vector<uint32_t> h_nidx(n_rec ? n_rec : 1, 0);
  uint64_t nidx = 0;
  for (uint32_t r = 0; r < n_rec; ++r) {
    h_nidx[r] = (uint32_t)nidx;
    const uint64_t len = ctx->ref_off[r + 1] - ctx->ref_off[r];
    if (len >= k && !h_has[r]) continue;
    ++nidx;
  }
  ctx->nidx = nidx;
  // Gene ids are stored as uint16_t (small_vector.hpp:46).  With more than 65 536 genes the reference keeps going: the
  // index wraps (gene 65536+x is stored as x and reported under x's name), and because bloomfilter.h:72 compares the
  // uint16_t last() with the int gene index -- never equal above 65535 -- such a gene is appended once per k-mer
  // OCCURRENCE, duplicates included.  A set bit's list is then [ascending unique ids of the genes below 65536] followed by
  // one entry per occurrence from the genes above, in gene order.  ReadAnalyzer only ever accumulates over a list, so the
  // order inside it does not matter, the multiplicities do: the lists are stored sorted by id with their duplicates
  // (`wrap` mode), and the classify kernels count multiplicities (classify.hip, WRAP).
  bool wrap = false;
  for (uint32_t r = 0; r < n_rec; ++r)
    if (h_has[r] && h_nidx[r] > 0xFFFFu) wrap = true;
  ix.wrap = wrap;

  // ---- pass 2 + switch_mode(2): (rank, gene) keys -> sort -> unique -> CSR ----
  BI_HIP(hipMalloc((void **)&d_csr_off, (n_set + 2) * sizeof(uint32_t)));
  uint64_t tot_idx = 0;
  if (n_valid > 0) {
    const uint64_t sentinel = n_set << (wrap ? 17 : 16);
    BI_HIP(hipMalloc((void **)&d_rec_nidx, n_rec * sizeof(uint32_t)));
    BI_HIP(hipMemcpyAsync(d_rec_nidx, h_nidx.data(), n_rec * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    BI_HIP(hipMalloc((void **)&d_keys, total * sizeof(uint64_t)));
    BI_HIP(hipMalloc((void **)&d_keys_alt, total * sizeof(uint64_t)));
    hipLaunchKernelGGL(ref_kmer_kernel<MODE_KEYS>, dim3(grid_for(total, RK_THREADS)), dim3(RK_THREADS), 0, st,
                       d_bytes, total, d_rec_off, n_rec, k, ix.bf64, ix.bf_bits, ix.bf_bits - 1, ix.pow2 ? 1 : 0,
                       (uint8_t *)nullptr, (unsigned long long *)nullptr, (const uint32_t *)ix.rank_w, (const uint32_t *)d_rec_nidx, d_keys, sentinel, wrap ? 1 : 0);
    BI_HIP(hipGetLastError());

    // bits needed to order keys up to and including the sentinel
    unsigned end_bit = wrap ? 18 : 17;
    while (end_bit < 64 && (sentinel >> end_bit) != 0) ++end_bit;
    if (total >= (1ull << 32)) { cleanup(); ctx->last_error = "reference has >= 2^32 bases"; return SHK_ERR_INDEX_TOO_LARGE; }
    uint32_t *d_hist = nullptr;
    uint64_t *d_sort_scan = nullptr;
    BI_HIP(hipMalloc((void **)&d_hist, radix_sort_hist_words(total) * sizeof(uint32_t)));
    d_sort_tmp = d_hist;   // freed by cleanup()
    BI_HIP(hipMalloc((void **)&d_sort_scan, scan_temp_words(radix_sort_hist_words(total)) * sizeof(uint64_t)));
    const uint64_t *sorted = radix_sort_u64(d_keys, d_keys_alt, total, end_bit, d_hist, d_sort_scan, st);
    BI_HIP(hipStreamSynchronize(st));
    (void)hipFree(d_sort_scan);
    if (!sorted) { cleanup(); ctx->last_error = "radix sort launch failed"; return SHK_ERR_HIP; }

    BI_HIP(hipMalloc((void **)&d_flags, total * sizeof(uint32_t)));
    hipLaunchKernelGGL(unique_flags_kernel, dim3(grid_for(total, 256)), dim3(256), 0, st, sorted, total, sentinel, wrap ? 1 : 0, d_flags);
    BI_HIP(hipGetLastError());
    const uint64_t *d_tot = exclusive_scan_u32(d_flags, d_flags, total, d_scan_tmp, st);
    BI_HIP(hipMemcpyAsync(&tot_idx, d_tot, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    BI_HIP(hipStreamSynchronize(st));
    if (tot_idx >= (1ull << 31)) { cleanup(); ctx->last_error = "index has >= 2^31 list entries (int tot_idx, bloomfilter.h:130)"; return SHK_ERR_INDEX_TOO_LARGE; }
    BI_HIP(hipMalloc((void **)&ix.ids, (tot_idx + 8) * sizeof(uint16_t)));
    hipLaunchKernelGGL(csr_write_kernel, dim3(grid_for(total, 256)), dim3(256), 0, st, sorted, (const uint32_t *)d_flags, total, sentinel, wrap ? 1 : 0, d_csr_off, ix.ids);
    BI_HIP(hipGetLastError());
  } else {
    BI_HIP(hipMalloc((void **)&ix.ids, 8 * sizeof(uint16_t)));
  }
  const uint32_t tot32 = (uint32_t)tot_idx;
  BI_HIP(hipMemcpyAsync(d_csr_off + n_set, &tot32, sizeof(uint32_t), hipMemcpyHostToDevice, st));
  BI_HIP(hipMalloc((void **)&ix.ent, (n_set + 1) * sizeof(ListEntry)));
  hipLaunchKernelGGL(list_entry_kernel, dim3(grid_for(n_set + 1, 256)), dim3(256), 0, st, (const uint32_t *)d_csr_off, (const uint16_t *)ix.ids, n_set, tot32, ix.ent);
  BI_HIP(hipGetLastError());
  ix.tot_idx = tot_idx;

  BI_HIP(hipStreamSynchronize(st));

  // ---- position table: exact sparse encoding of the set bits -----
  ix.tab_lg = 0;
  uint64_t table_bytes = 0;
  const char *force = getenv("SHK_PROBE");   // "bitvector" disables the table (tests exercise both paths)
  if (n_set > 0 && n_set <= TAB_PAYLOAD && !(force && force[0] == 'b')) {   // (any filter size: buckets and tags are cut from the position, not from the hash)
    uint32_t lgB = 0;
    while ((1ull << lgB) < ix.bf_bits) ++lgB;
    uint32_t lg = 9;                                        // >= 512 buckets
    // Smallest power of two with a load factor <= 0.30 -- and roomier while the table stays small, because a search that
    // leaves its home bucket is a dependent memory round trip for the whole wave: down to a load of 0.05 up to 2 MiB (half
    // of an XCD's L2), and of 0.15 up to 8 MiB, where random lookups still run at 121 G/s (82 G/s at 16 MiB, 55 G/s beyond;
    // tools/gather_bench).  Per 10 M pairs, 50 % on-target (profiles/README.md):
    //   20 000 k-mers: load 0.15 -> 0.076: 9.8 -> 9.3 ms;   71 000: 2 MiB, 0.27 -> 4 MiB, 0.14: 13.6 -> 11.8 ms;
    //   143 000: 4 MiB, 0.27 -> 8 MiB, 0.14: 18.1 -> 12.6 ms;   238 000: 8 MiB, 0.23 stays (16 MiB, 0.11: 16.4 -> 19.7 ms)
    // Denser than 0.30 is worse even where it would bring the table back into L2 (load 0.46: 100 genes 16.3 -> 22.0 ms).
    // SHK_TAB_DENSE=1 (tests): load up to 0.8, so that long probe paths are exercised.
#ifndef SHK_TAB_LOAD10
#define SHK_TAB_LOAD10 3
#endif
    if (getenv("SHK_TAB_DENSE")) {
      while ((2ull << lg) * 8ull < 10ull * n_set) ++lg;
    } else {
      while ((2ull << lg) * SHK_TAB_LOAD10 < 10ull * n_set) ++lg;
      // (never beyond half of the filter's positions: bucket index and tag are cut from the position)
      while (lg < 17 && lg + 1 < lgB && (2ull << lg) < 20ull * n_set) ++lg;
      while (lg < 19 && lg + 1 < lgB && (2ull << lg) * 3ull < 20ull * n_set) ++lg;
    }
    if (lgB > 24 && lg < lgB - 24) lg = lgB - 24;           // tag must fit 24 bits
    if (lg < lgB && lg <= 31) {                             // (bucket indices are 32-bit in the kernel)
      const uint64_t slots = 2ull << lg;
      uint32_t *d_fail = nullptr;
      // (+1 bucket that stays empty: where the kernel sends probes that need no answer)
      BI_HIP(hipMalloc((void **)&ix.tab, (slots + 2) * sizeof(uint64_t)));
      BI_HIP(hipMemsetAsync(ix.tab, 0, (slots + 2) * sizeof(uint64_t), st));
      BI_HIP(hipMalloc((void **)&d_fail, sizeof(uint32_t)));
      BI_HIP(hipMemsetAsync(d_fail, 0, sizeof(uint32_t), st));
      hipLaunchKernelGGL(table_build_kernel, dim3(grid_for(n_words, 256)), dim3(256), 0, st, (const uint64_t *)ix.bf64, n_words,
                         (const uint32_t *)ix.rank_w, (const ListEntry *)ix.ent, reinterpret_cast<unsigned long long *>(ix.tab), lg, d_fail);
      BI_HIP(hipGetLastError());
      uint32_t h_fail = 0;
      BI_HIP(hipMemcpyAsync(&h_fail, d_fail, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
      BI_HIP(hipStreamSynchronize(st));
      (void)hipFree(d_fail);
      if (h_fail == 0) {
        ix.tab_lg = lg;
        // small indices: a 2^18-bit summary that the table kernel keeps in LDS answers
        // almost every miss without touching the memory system (same proof as sum32)
        ix.lsum_shift = 0;
        if (lgB > LDS_SUM_LOG2 && !getenv("SHK_NO_LDS_SUMMARY")) {
          const uint32_t sh = lgB - LDS_SUM_LOG2;
          const double pass = 1.0 - std::exp(-(double)n_set * (double)(1ull << sh) / (double)ix.bf_bits);
          if (sh >= 6 && sh < 32 && pass <= 0.30) {
            BI_HIP(hipMalloc((void **)&ix.lsum32, (LDS_SUM_BITS / 32 + 2) * sizeof(uint32_t)));
            BI_HIP(hipMemsetAsync(ix.lsum32, 0, (LDS_SUM_BITS / 32 + 2) * sizeof(uint32_t), st));
            hipLaunchKernelGGL(bf_summary_kernel, dim3(grid_for(n_words, 256)), dim3(256), 0, st, (const uint64_t *)ix.bf64, n_words, sh, ix.lsum32);
            BI_HIP(hipGetLastError());
            BI_HIP(hipStreamSynchronize(st));
            ix.lsum_shift = sh;
          }
        }
        // too dense for 2^18 bits: a 2^20-bit summary for the uniform-length kernel (128 KiB of LDS, classify_uni.hpp LSL = 20)
        ix.lbig_shift = 0;
        // (only where the table has outgrown an XCD's L2: 100 genes 24.4 -> 20.2 ms, 150 genes 26.3 -> 24.5 ms per 10 M pairs;
        //  with a 4 MiB table probing it directly is as fast, 60 genes 19.7 ms)
        if (!ix.lsum_shift && lgB > 20 && slots * sizeof(uint64_t) > (4ull << 20) && !getenv("SHK_NO_LDS_SUMMARY") && !getenv("SHK_NO_BIG_LDS_SUMMARY")) {
          const uint32_t sh = lgB - 20;
          const double pass = 1.0 - std::exp(-(double)n_set * (double)(1ull << sh) / (double)ix.bf_bits);
          if (sh >= 6 && sh < 32 && pass <= 0.30) {
            BI_HIP(hipMalloc((void **)&ix.lbig32, ((1u << 20) / 32 + 2) * sizeof(uint32_t)));
            BI_HIP(hipMemsetAsync(ix.lbig32, 0, ((1u << 20) / 32 + 2) * sizeof(uint32_t), st));
            hipLaunchKernelGGL(bf_summary_kernel, dim3(grid_for(n_words, 256)), dim3(256), 0, st, (const uint64_t *)ix.bf64, n_words, sh, ix.lbig32);
            BI_HIP(hipGetLastError());
            BI_HIP(hipStreamSynchronize(st));
            ix.lbig_shift = sh;
            ix.lbig_pass = pass;
          }
        }
        // tiny indices (a gene or a few): the whole table fits the LDS of a CU as a perfect hash -- uniform batches then
        // touch no memory but their own bases (classify_uni.hpp LSL = 21); the chains above stay for trimmed reads
        if (ix.ltab) { (void)hipFree(ix.ltab); ix.ltab = nullptr; }
        ix.ltab_gene = 0xFFFFFFFFu;
        ix.ltab_sparse = false;
        if (ix.pow2 && ix.lsum_shift && lgB >= 24 && lgB <= LTAB_MAX_POS_LG && n_set <= LTAB_MAX_KEYS && !getenv("SHK_NO_LDS_TABLE")) {
          std::vector<uint64_t> h_tab(slots);
          std::vector<uint32_t> img;
          BI_HIP(hipMemcpyAsync(h_tab.data(), ix.tab, slots * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
          BI_HIP(hipStreamSynchronize(st));
          uint32_t one_gene = 0xFFFFFFFFu;
          if (build_lds_table(h_tab, lg, img, &ix.ltab_mul, &one_gene)) {
            if (!getenv("SHK_NO_SPARSE")) ix.ltab_gene = one_gene;   // (SHK_NO_SPARSE=1: the usual probe order on one-gene indices too; the tests run both)
            ix.ltab_sparse = !getenv("SHK_NO_SPARSE");
            BI_HIP(hipMalloc((void **)&ix.ltab, LTAB_BYTES));
            BI_HIP(hipMemcpyAsync(ix.ltab, img.data(), LTAB_BYTES, hipMemcpyHostToDevice, st));
            BI_HIP(hipStreamSynchronize(st));
          }
        }
        table_bytes = slots * sizeof(uint64_t);
        // the reference itself, for the anchored extension of the table kernels (SHK_NO_ANCHOR=1: not built; the tests run both).
        // Optional: when its memory cannot be had the index is complete without it.
        ix.ref_total = 0;
        if (!wrap && total > 0 && total < (1ull << 31) && !getenv("SHK_NO_ANCHOR")) {
          const uint64_t n_dw = (total + 15) / 16;
          uint32_t *d_lost = nullptr, *d_lens = nullptr, *d_anchor = nullptr;
          uint64_t *d_stmp = nullptr;
          ListEntry *ent_all = nullptr;
          uint16_t *ids_all = nullptr;
          auto drop_anchor = [&]() {
            (void)hipFree(ix.ref2); (void)hipFree(ix.refpay); (void)hipFree(ix.atab); (void)hipFree(ix.refext); (void)hipFree(ix.refmul);
            ix.ref2 = ix.refpay = ix.refext = ix.refmul = nullptr;
            ix.atab = nullptr;
            ix.ref_total = 0;
          };
          // Everything in here is optional, so nothing in here may fail the build: an allocation that cannot be had, a launch or a
          // copy that fails -- the temporaries are freed, the extension is dropped, the sticky error is cleared and the index is
          // complete without it.
#define AX_HIP(call) do { if ((call) != hipSuccess) return false; } while (0)
          bool have = hipMalloc((void **)&ix.ref2, (n_dw + 4) * sizeof(uint32_t)) == hipSuccess &&
                      hipMalloc((void **)&ix.refpay, (total + 8) * sizeof(uint32_t)) == hipSuccess &&   // (+8: read 16 bytes at a time by tools)
                      hipMalloc((void **)&d_anchor, (slots + 2) * sizeof(uint32_t)) == hipSuccess &&
                      hipMalloc((void **)&ix.atab, (slots + 2) * sizeof(uint64_t)) == hipSuccess &&
                      hipMalloc((void **)&d_lost, sizeof(uint32_t)) == hipSuccess;
          have = have && [&]() -> bool {
            AX_HIP(hipMemsetAsync(ix.ref2 + n_dw, 0, 4 * sizeof(uint32_t), st));
            AX_HIP(hipMemsetAsync(ix.refpay + total, 0xFF, 8 * sizeof(uint32_t), st));
            AX_HIP(hipMemsetAsync(d_anchor, 0xFF, (slots + 2) * sizeof(uint32_t), st));
            AX_HIP(hipMemsetAsync(d_lost, 0, sizeof(uint32_t), st));
            hipLaunchKernelGGL(ref_pack2_kernel, dim3(grid_for(n_dw, 256)), dim3(256), 0, st, d_bytes, total, ix.ref2, n_dw);
            AX_HIP(hipGetLastError());
            hipLaunchKernelGGL(ref_anchor_kernel, dim3(grid_for(total, RK_THREADS)), dim3(RK_THREADS), 0, st, d_bytes, total, d_rec_off, n_rec, k, ix.bf_bits,
                               ix.bf_bits - 1, ix.pow2 ? 1 : 0, (const uint64_t *)ix.tab, lg, ix.refpay, d_anchor, d_lost);
            AX_HIP(hipGetLastError());
            hipLaunchKernelGGL(atab_fill_kernel, dim3(grid_for(slots + 2, 256)), dim3(256), 0, st, (const uint64_t *)ix.tab, (const uint32_t *)d_anchor, ix.atab,
                               (uint64_t)(slots + 2));
            AX_HIP(hipGetLastError());
            uint32_t h_lost = 0;
            AX_HIP(hipMemcpyAsync(&h_lost, d_lost, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            AX_HIP(hipStreamSynchronize(st));
            return h_lost == 0;   // (a key the table does not hold: leave the extension out rather than trust it)
          }();
          (void)hipFree(d_lost);
          (void)hipFree(d_anchor);
          // multi-gene lists along the reference (see ref_multi_len_kernel); left out -- the lists then stay where the ranks point --
          // when the copies would not fit the 32-bit list offsets or the 30-bit payload, or their memory cannot be had.  A failure
          // behind the point where refpay has been rewritten drops the whole extension (ix.ent / ix.ids stay valid either way: their
          // rank-ordered parts are copied before they are replaced).
          have = have && [&]() -> bool {
            if (!(n_set + 1 + total + 1 <= TAB_PAYLOAD && hipMalloc((void **)&d_lens, (total + 1) * sizeof(uint32_t)) == hipSuccess &&
                  hipMalloc((void **)&d_stmp, scan_temp_words(total + 1) * sizeof(uint64_t)) == hipSuccess))
              return true;
            hipLaunchKernelGGL(ref_multi_len_kernel, dim3(grid_for(total + 1, 256)), dim3(256), 0, st, (const uint32_t *)ix.refpay, total, (const ListEntry *)ix.ent, d_lens);
            AX_HIP(hipGetLastError());
            const uint64_t *d_R = exclusive_scan_u32(d_lens, d_lens, total + 1, d_stmp, st);
            uint64_t R = 0;
            AX_HIP(hipMemcpyAsync(&R, d_R, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
            AX_HIP(hipStreamSynchronize(st));
            if (!(tot_idx + R + 8 < (1ull << 32) && R <= 16 * total && hipMalloc((void **)&ent_all, (n_set + 1 + total + 1) * sizeof(ListEntry)) == hipSuccess &&
                  hipMalloc((void **)&ids_all, (tot_idx + R + 8) * sizeof(uint16_t)) == hipSuccess))
              return true;
            AX_HIP(hipMemcpyAsync(ent_all, ix.ent, (n_set + 1) * sizeof(ListEntry), hipMemcpyDeviceToDevice, st));
            AX_HIP(hipMemcpyAsync(ids_all, ix.ids, (tot_idx + 8) * sizeof(uint16_t), hipMemcpyDeviceToDevice, st));
            AX_HIP(hipStreamSynchronize(st));
            (void)hipFree(ix.ent); (void)hipFree(ix.ids);
            ix.ent = ent_all; ix.ids = ids_all;
            ent_all = nullptr; ids_all = nullptr;
            hipLaunchKernelGGL(ref_multi_write_kernel, dim3(grid_for(total + 1, 256)), dim3(256), 0, st, ix.refpay, total, ix.ent, n_set, ix.ids, (uint32_t)tot_idx,
                               (const uint32_t *)d_lens);
            AX_HIP(hipGetLastError());
            AX_HIP(hipStreamSynchronize(st));
            return true;
          }();
          // the surroundings of every anchor (ref_extent_kernel, ref_multi_bits_kernel), for anchor_verdict_kernel; optional by
          // itself (SHK_NO_REFEXT=1: not built; the tests run both).  k = 1: runs of valid positions do not end with their record
          if (ix.refext) { (void)hipFree(ix.refext); ix.refext = nullptr; }
          if (ix.refmul) { (void)hipFree(ix.refmul); ix.refmul = nullptr; }
          if (have && k >= 2 && !getenv("SHK_NO_REFEXT")) {
            const uint64_t n_mw = (total + 31) / 32 + 2;
            const bool ok = [&]() -> bool {
              AX_HIP(hipMalloc((void **)&ix.refext, (total + 8) * sizeof(uint32_t)));
              AX_HIP(hipMalloc((void **)&ix.refmul, n_mw * sizeof(uint32_t)));
              AX_HIP(hipMemsetAsync(ix.refext + total, 0xFF, 8 * sizeof(uint32_t), st));
              hipLaunchKernelGGL(ref_extent_kernel, dim3(grid_for(total, 256)), dim3(256), 0, st, (const uint32_t *)ix.refpay, total, ix.refext);
              AX_HIP(hipGetLastError());
              hipLaunchKernelGGL(ref_multi_bits_kernel, dim3(grid_for(n_mw, 256)), dim3(256), 0, st, (const uint32_t *)ix.refpay, total, ix.refmul, n_mw);
              AX_HIP(hipGetLastError());
              AX_HIP(hipStreamSynchronize(st));
              return true;
            }();
            if (!ok) { (void)hipFree(ix.refext); (void)hipFree(ix.refmul); ix.refext = ix.refmul = nullptr; }
          }
#undef AX_HIP
          (void)hipGetLastError();   // (nothing that failed above is an error of the build)
          (void)hipFree(ent_all); (void)hipFree(ids_all); (void)hipFree(d_lens); (void)hipFree(d_stmp);
          if (!have) drop_anchor();
          if (have) ix.ref_total = (uint32_t)total;
        }
        // ---- the k-mer keyed, minimiser-bucketed table (k = 15 ... 17, power-of-two filters, no wrap mode): for tables far beyond
        // the caches, where every probe of `tab` is a memory-side request of its own and their RATE is the wall;
        // SHK_KTAB=1 builds it for any table (tests), SHK_NO_KTAB=1 never.  Optional like the anchored extension: when its memory
        // cannot be had, or a key finds no place, the index is complete without it.
        ix.ktab_lg = 0;
        const bool ktab_forced = getenv("SHK_KTAB") != nullptr;
        if (ix.pow2 && !wrap && k >= 15 && k <= 17 && !getenv("SHK_NO_KTAB") && (ktab_forced || table_bytes > (256ull << 20))) {
          // (SHK_KTAB_W / SHK_KTAB_LOAD: the minimiser's length and the table's load in per cent, for A/B timing)
          // Measured on the 60 000-gene index (2^36-bit filter, 1.9 x 10^8 keys), per 10 M pairs at 0 / 50 / 100 % on-target, against
          // 28.8 / 19.2 / 17.9 ms through the position table: w = 14, load 0.18: 19.9 / 19.6 / 19.1 ms (11 % of the keys behind their
          // home bucket -- a minimiser that many k-mers share overfills its line); w = 15, load 0.18: 18.2 / 18.3 / 18.8 (4 %);
          // w = 15, load 0.09 (2^27 lines, 16 GiB): 17.8 / 17.8 / 18.8 (2 %)
          uint32_t w = 15;
          double load = 0.09;
          if (const char *e = getenv("SHK_KTAB_W")) { const int v = atoi(e); if (v >= 8 && v <= 15) w = (uint32_t)v; }
          if (const char *e = getenv("SHK_KTAB_LOAD")) { const int v = atoi(e); if (v >= 2 && v <= 50) load = v / 100.0; }
          if (w > k) w = k;
          if (k - w > 3) w = k - 3;
          // keys: the set bits' own k-mers (about n_set) + the filter's false positives among the other canonical k-mers
          const double canon = 0.5 * std::pow(4.0, (double)k);
          const double est = (double)n_set + canon * ((double)n_set / (double)ix.bf_bits);
          uint32_t line_lg = 6;
          while (line_lg < 28 && (double)(16ull << line_lg) * load < est) ++line_lg;
          // What the device has left decides as well: the table may take what is free less an eighth of the device's memory
          // (at least 8 GiB) -- the batches' buffers, the arrays built behind this one, other workers' indices on the same device
          // (`--devices 0,0,0,0` builds one per worker).  Too little for the size wanted: half of it, twice the load (0.18 measured 3 %
          // behind 0.09), while the keys still fill at most half of the slots; else the index goes without -- it is complete without.
          size_t mem_free = 0, mem_total = 0;
          bool mem_known = hipMemGetInfo(&mem_free, &mem_total) == hipSuccess;
          if (const char *e = getenv("SHK_TEST_MEM_FREE")) { mem_free = (size_t)strtoull(e, nullptr, 10); mem_total = 0; mem_known = true; }   // (tests: as if that much were free)
          const uint64_t mem_keep = mem_known ? std::max<uint64_t>(8ull << 30, (uint64_t)mem_total / 8) : 0;
          auto fits_device = [&](const uint32_t lg2) { return !mem_known || ((16ull << lg2) + 16) * sizeof(uint64_t) + mem_keep <= (uint64_t)mem_free; };
          while (!fits_device(line_lg) && line_lg > 6 && (double)(16ull << (line_lg - 1)) * 0.5 >= est) --line_lg;
          const uint64_t kslots = 16ull << line_lg;
          if ((double)kslots * 0.5 >= est && kslots * sizeof(uint64_t) <= (64ull << 30) && fits_device(line_lg)) {
            unsigned long long *d_nk = nullptr;
            uint32_t *d_kf = nullptr;
            bool have = hipMalloc((void **)&ix.ktab, (kslots + 16) * sizeof(uint64_t)) == hipSuccess &&   // (+8 buckets that stay empty: probes that need no answer)
                        hipMalloc((void **)&d_nk, sizeof(unsigned long long)) == hipSuccess && hipMalloc((void **)&d_kf, 4 * sizeof(uint32_t)) == hipSuccess;
            unsigned long long h_nk = 0;
            uint32_t h_kf[4] = {0, 0, 0, 0};   // keys without a place, set bits the position table does not hold, keys behind their home bucket, longest path
            have = have && hipMemsetAsync(ix.ktab, 0, (kslots + 16) * sizeof(uint64_t), st) == hipSuccess && hipMemsetAsync(d_nk, 0, sizeof(unsigned long long), st) == hipSuccess &&
                   hipMemsetAsync(d_kf, 0, 4 * sizeof(uint32_t), st) == hipSuccess;
            if (have) {
              hipLaunchKernelGGL(ktab_build_kernel, dim3(256 * 16), dim3(256), 0, st, k, w, (const uint64_t *)ix.bf64, ix.bf_bits - 1, (const uint64_t *)ix.tab, lg,
                                 reinterpret_cast<unsigned long long *>(ix.ktab), line_lg + 3u, d_nk, d_kf, d_kf + 1);
              have = hipGetLastError() == hipSuccess && hipMemcpyAsync(&h_nk, d_nk, sizeof(h_nk), hipMemcpyDeviceToHost, st) == hipSuccess &&
                     hipMemcpyAsync(h_kf, d_kf, sizeof(h_kf), hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
            }
            (void)hipGetLastError();
            (void)hipFree(d_nk); (void)hipFree(d_kf);
            if (getenv("SHK_KTAB_STATS"))
              fprintf(stderr, "[shk/ktab] k=%u w=%u lines=2^%u keys=%llu (estimated %.0f) load=%.3f displaced=%u (%.2f %%) longest path=%u no place=%u lost=%u\n", k, w, line_lg,
                      h_nk, est, (double)h_nk / (double)kslots, h_kf[2], h_nk ? 100.0 * h_kf[2] / (double)h_nk : 0.0, h_kf[3], h_kf[0], h_kf[1]);
            if (have && h_kf[0] == 0 && h_kf[1] == 0) {
              ix.ktab_lg = line_lg + 3u;
              ix.ktab_w = w;
              ix.ktab_keys = h_nk;
            } else {
              (void)hipFree(ix.ktab);
              ix.ktab = nullptr;
            }
          }
        }
      } else {
        (void)hipFree(ix.tab);   // displacement overflow: keep the bit-vector path
        ix.tab = nullptr;
      }
    }
  }
  // ---- summary level (result preserving) -------------------------
  // A clear summary bit proves 2^shift filter bits clear.  A summary probe that misses L2 is itself
  // a memory-side request, and the kernel is bound by the RATE of such requests on large indices
  // (~54 G/s measured, whether the Infinity Cache or HBM serves them), so:
  //  * in front of the position table only an L2-sized summary (<= 2^24 bits) that passes <= 30 %
  //    of random probes is used, and only when neither the LDS summary nor a cache-resident table
  //    makes it pointless (the Infinity-Cache-sized summary made the 60 000-gene index 23 % slower);
  //  * in front of plain filter words (no table): L2-sized at <= 5 %, else Infinity-Cache-sized
  //    (<= 2^30 bits) at <= 50 %.
  ix.sum_shift = 0;
  ix.tab_with_summary = false;
  if (ix.pow2 && ix.bf_bits >= (1ull << 12) && n_set > 0 && !getenv("SHK_NO_SUMMARY")) {
    auto pass_rate = [&](uint32_t sh) { return 1.0 - std::exp(-(double)n_set * (double)(1ull << sh) / (double)ix.bf_bits); };
    uint32_t lg = 0;
    while ((1ull << lg) < ix.bf_bits) ++lg;
    const uint32_t sh_l2 = std::max<uint32_t>(6, lg > 24 ? lg - 24 : 6);
    const uint32_t sh_ic = std::max<uint32_t>(6, lg > 30 ? lg - 30 : 6);
    if (ix.tab_lg) {
      if (!ix.lsum_shift && table_bytes > (4ull << 20) && sh_l2 < lg && pass_rate(sh_l2) <= 0.30) {
        ix.sum_shift = sh_l2;
        ix.tab_with_summary = true;
      }
    } else {
      if (sh_l2 < lg && pass_rate(sh_l2) <= 0.05) ix.sum_shift = sh_l2;
      else if (sh_ic < lg && pass_rate(sh_ic) <= 0.5) ix.sum_shift = sh_ic;
    }
  }
  if (ix.sum_shift) {
    ix.sum_bits = ix.bf_bits >> ix.sum_shift;
    const uint64_t sw = (ix.sum_bits + 31) / 32 + 2;
    BI_HIP(hipMalloc((void **)&ix.sum32, sw * sizeof(uint32_t)));
    BI_HIP(hipMemsetAsync(ix.sum32, 0, sw * sizeof(uint32_t), st));
    hipLaunchKernelGGL(bf_summary_kernel, dim3(grid_for(n_words, 256)), dim3(256), 0, st, (const uint64_t *)ix.bf64, n_words, ix.sum_shift, ix.sum32);
    BI_HIP(hipGetLastError());
  }
  BI_HIP(hipStreamSynchronize(st));

  cleanup();
#undef BI_HIP
  return SHK_OK;
}

}  // namespace shk
