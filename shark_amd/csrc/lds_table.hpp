// lds_table.hpp -- the LDS-resident exact table of a tiny index: layout, host-side construction and the lookup rule
// (plain C++: index_build.hip builds the image, classify_uni.hpp reads it in LDS with the same rule, the host-only tool
// shark-ltab-check lets the CPU tests verify construction and exactness without a GPU).
//
// A perfect hash by displacement over the set bits of the filter (positions < 2^33).  Group g = bits [15, 28) of the
// position, D[g] = the group's displacement, hi = the bits above the group ([28, 33)),
//   slot = (pos + hi * mul + D[g]) mod 2^15,
//   T[slot] = tag(18) = pos >> 15 | valid(1) | payload(13): the gene of a single-gene list, or LTAB_ESC (ask the position table).
// The tag contains the group and hi, so a matching entry was placed with the same offset: its low 15 bits are the probe's --
// a match is exact, and a lookup is two reads without any search.  (The hi term separates keys of one group that agree in
// their low 15 bits and differ above bit 28 -- one such pair is expected among 20 000 keys of a 2^33-bit filter, and no
// displacement could.  Two keys of a group still share a slot before the displacement when their low bits differ by exactly
// (difference of hi) * mul: about one index in three has such a pair for a given odd `mul`, so the builder tries up to
// LTAB_MUL_TRIES multipliers and hands the one that works to the kernel.)
#pragma once
#include <stdint.h>

#include <algorithm>
#include <vector>

namespace shk {

constexpr uint32_t LTAB_SLOT_LG = 15, LTAB_GROUP_LG = 13;
constexpr uint32_t LTAB_T_WORDS = 1u << LTAB_SLOT_LG;                                  // 128 KiB
constexpr uint32_t LTAB_BYTES = LTAB_T_WORDS * 4u + (1u << LTAB_GROUP_LG) * 2u;        // + 16 KiB of displacements
constexpr uint32_t LTAB_ESC = 0x1FFFu;
constexpr uint32_t LTAB_MUL_TRIES = 16;
inline constexpr uint32_t ltab_mul(uint32_t attempt) { return 1021u + 5462u * attempt; }   // odd
constexpr uint32_t LTAB_MAX_KEYS = 26000;                                              // load <= 0.8
constexpr uint32_t LTAB_MAX_POS_LG = LTAB_SLOT_LG + 18;                                // the tag has 18 bits

struct LtabKey {
  uint64_t pos;       // filter position (< 2^LTAB_MAX_POS_LG)
  uint32_t payload;   // gene of a single-gene list (< LTAB_ESC), or LTAB_ESC
};

// The image (T then D, LTAB_BYTES) for a set of distinct keys and a multiplier.  The displacements are found greedily, largest
// group first (a few thousand groups of a handful of keys: microseconds).  false = two keys of a group share a base slot, or no
// displacement fits some group.
inline bool ltab_build_with(const std::vector<LtabKey> &keys, const uint32_t mul, std::vector<uint32_t> &img)
{
  struct K { uint32_t base, tag, payload; };   // base = slot before the displacement
  constexpr uint32_t NG = 1u << LTAB_GROUP_LG, NS = 1u << LTAB_SLOT_LG;
  std::vector<std::vector<K>> groups(NG);
  for (const LtabKey &k : keys) {
    const uint32_t tag = (uint32_t)(k.pos >> LTAB_SLOT_LG);
    groups[tag & (NG - 1)].push_back(K{((uint32_t)k.pos + (tag >> LTAB_GROUP_LG) * mul) & (NS - 1), tag, k.payload});
  }
  std::vector<uint32_t> order(NG);
  for (uint32_t g = 0; g < NG; ++g) order[g] = g;
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return groups[a].size() > groups[b].size(); });
  img.assign(LTAB_BYTES / 4, 0u);
  uint16_t *D = reinterpret_cast<uint16_t *>(img.data() + LTAB_T_WORDS);
  for (const uint32_t g : order) {
    auto &ks = groups[g];
    if (ks.empty()) break;
    // two keys of a group on one base slot cannot be told apart by any displacement: another multiplier has to be tried
    std::sort(ks.begin(), ks.end(), [](const K &a, const K &b) { return a.base < b.base; });
    for (size_t i = 1; i < ks.size(); ++i)
      if (ks[i].base == ks[i - 1].base) return false;
    bool placed = false;
    for (uint32_t d = 0; d < NS && !placed; ++d) {
      bool free_all = true;
      for (const K &kk : ks)
        if (img[(kk.base + d) & (NS - 1)]) { free_all = false; break; }
      if (!free_all) continue;
      for (const K &kk : ks) img[(kk.base + d) & (NS - 1)] = (kk.tag << 14) | (1u << 13) | kk.payload;
      D[g] = (uint16_t)d;
      placed = true;
    }
    if (!placed) return false;
  }
  return true;
}

// ... trying the multipliers in turn.  *mul = the one the image was built with.
inline bool ltab_build(const std::vector<LtabKey> &keys, std::vector<uint32_t> &img, uint32_t *mul)
{
  for (uint32_t a = 0; a < LTAB_MUL_TRIES; ++a)
    if (ltab_build_with(keys, ltab_mul(a), img)) { *mul = ltab_mul(a); return true; }
  return false;
}

// The lookup rule, as classify_uni_kernel applies it to the raw 64-bit hash `h` of a filter with 2^lgB bits (bf_mask =
// 2^lgB - 1): true = the position is a key, *payload = its entry's payload.
inline bool ltab_lookup(const uint32_t *img, uint32_t mul, uint64_t h, uint64_t bf_mask, uint32_t *payload)
{
  const uint32_t tagmask = (uint32_t)(bf_mask >> LTAB_SLOT_LG);
  const uint32_t gmask = tagmask & ((1u << LTAB_GROUP_LG) - 1u);
  const uint16_t *D = reinterpret_cast<const uint16_t *>(img + LTAB_T_WORDS);
  const uint32_t lo = (uint32_t)h;
  const uint32_t tag = (uint32_t)(h >> LTAB_SLOT_LG) & tagmask;
  const uint32_t d = D[(lo >> LTAB_SLOT_LG) & gmask];
  const uint32_t e = img[(lo + (tag >> LTAB_GROUP_LG) * mul + d) & (LTAB_T_WORDS - 1u)];
  *payload = e & LTAB_ESC;
  return (e >> 13) == ((tag << 1) | 1u);
}

}  // namespace shk
