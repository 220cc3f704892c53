// shark-ltab-check -- host-only check of the LDS-resident exact table (lds_table.hpp): builds the image for N random distinct
// filter positions of a 2^LGB-bit filter and verifies, with the lookup rule the kernel uses, that every key is found with its
// payload and that M random hashes are answered exactly as a std::set of the keys answers them.
// usage: shark-ltab-check N LGB SEED [M]   -> one JSON object
#include <cstdio>
#include <cstdlib>
#include <random>
#include <set>

#include "lds_table.hpp"

int main(int argc, char **argv)
{
  if (argc < 4) { fprintf(stderr, "usage: %s N LGB SEED [M]\n", argv[0]); return 2; }
  const uint32_t n = (uint32_t)atoi(argv[1]), lgB = (uint32_t)atoi(argv[2]);
  const uint64_t seed = strtoull(argv[3], nullptr, 10), m = argc > 4 ? strtoull(argv[4], nullptr, 10) : 1000000ull;
  const uint64_t mask = (1ull << lgB) - 1ull;
  std::mt19937_64 rng(seed);
  std::set<uint64_t> pos;
  while (pos.size() < n) pos.insert(rng() & mask);
  std::vector<shk::LtabKey> keys;
  for (uint64_t p : pos) keys.push_back(shk::LtabKey{p, (uint32_t)(rng() % 3 == 0 ? shk::LTAB_ESC : rng() % shk::LTAB_ESC)});
  std::vector<uint32_t> img;
  uint32_t mul = 0;
  const bool built = shk::ltab_build(keys, img, &mul);
  unsigned long long missing = 0, wrong_payload = 0, false_pos = 0, false_neg = 0, used = 0;
  if (built) {
    for (uint32_t i = 0; i < shk::LTAB_T_WORDS; ++i) used += img[i] != 0u;
    for (const shk::LtabKey &k : keys) {
      uint32_t pl = 0;
      // the kernel sees the raw hash: bits above the filter size must not matter
      const uint64_t h = k.pos | (rng() & ~mask);
      if (!shk::ltab_lookup(img.data(), mul, h, mask, &pl)) ++missing;
      else if (pl != k.payload) ++wrong_payload;
    }
    for (uint64_t i = 0; i < m; ++i) {
      uint64_t h = rng();
      if (i % 4 == 1 && !keys.empty()) h = (keys[h % keys.size()].pos ^ (1ull << (h % lgB))) | (h & ~mask);   // a near miss: one bit off a key
      uint32_t pl = 0;
      const bool got = shk::ltab_lookup(img.data(), mul, h, mask, &pl), want = pos.count(h & mask) != 0;
      false_pos += got && !want;
      false_neg += !got && want;
    }
  }
  printf("{\"built\": %s, \"mul\": %u, \"keys\": %u, \"slots_used\": %llu, \"missing\": %llu, \"wrong_payload\": %llu, \"false_pos\": %llu, \"false_neg\": %llu, \"probes\": %llu}\n",
         built ? "true" : "false", mul, n, used, missing, wrong_payload, false_pos, false_neg, (unsigned long long)m);
  return 0;
}
