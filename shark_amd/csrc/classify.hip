// classify.hip -- per-read k-mer classification on gfx950 (wave64).
//
// One WAVEFRONT per read (pair).  Replaces, with bit-identical results,
//   FastqSplitter.hpp:63,:83,:104-109   mate join + quality mask (semantics)
//   ReadAnalyzer::operator()            ReadAnalyzer.hpp:39-110
//   BF::get_index                       bloomfilter.h:78-102
//
// How the reference's serial loop is re-expressed:
//  * k-mer slots.  The rolling walk with restart-on-invalid (ReadAnalyzer.hpp
//    :51-77, kmer_utils.hpp:57-71) visits exactly the k-mers whose k
//    characters are all valid.  The mates are laid out at packed positions
//    (mate 1 at 0, mate 2 at P2 = L1 rounded up to 8); slot pp is the k-mer
//    starting at packed position pp and exists for pp < nk1 or 0 <= pp-P2 < nk2
//    (nk_m = max(0, L_m-k+1)).  The joiner 'N' (FastqSplitter.hpp:63) is
//    invalid, so no k-mer spans the mates and the two mates are two slot ranges.
//  * per-gene coverage.  For a gene g with hit end positions p0<p1<..., the
//    reference accumulates cov = k + sum min(k, p_j - p_{j-1}), nk = #hits
//    (ReadAnalyzer.hpp:56-62,:79-86; a fresh map entry is ((0,0),0) and its
//    first increment is min(k, pos-0) = k).  Hits in different mates are more
//    than k apart in joined coordinates, so the min clamps to k there -- and at
//    least k apart in packed positions.  cov is the size of the union of the
//    intervals [p_j, p_j + k), which is how the kernel counts it.
//  * gene order.  Every set bit's list is ascending and duplicate free
//    (bloomfilter.h:68-74), so a k-way merge over the lists of all hit slots
//    visits genes in ascending id order -- the std::map iteration order used by
//    the arg-max with ties (ReadAnalyzer.hpp:90-102).
//  * threshold.  `max >= c*len` is evaluated in double exactly as written
//    (ReadAnalyzer.hpp:104); gfx950 has IEEE fp64 multiply/compare.
//
// Data path per read: 8 bases per lane are loaded as aligned dwords (prefetched one read ahead),
// classified with SWAR, and written to LDS as two streams of 2-bit codes (forward and mirrored)
// plus a 1-bit validity stream.  Every lane then cuts both orientations of its k-mers out of LDS
// with v_alignbit_b32, hashes the canonical one (XXH64, 15 integer multiplies) and issues all its
// probes before the first wait.  `__ballot` of the pass / hit flags ends the read immediately when
// nothing hit -- the common case for off-target reads.  DESIGN.md 3 has the full description and
// what bounds the kernel.
#include <cstdio>
#include <cstring>

#include "classify_common.hpp"

namespace shk {

// WRAP (general kernel only): the index has more than 65 536 genes, so a list may hold an id several times
// (index_build.hip).  The reference's accumulation then sees that id several times for one k-mer (ReadAnalyzer.hpp:56-62,
// :79-86): behind the FIRST valid k-mer of the read, a repeated id adds min(k, pos - last) = min(k, 0) = 0 to the coverage
// and 1 to the k-mer count; at the first k-mer itself (handled separately at :56-62, `nk = 1`, last = pos - 1) every
// repetition adds min(k, 1) = 1 to the coverage and leaves the count at 1.  So with m_j = multiplicity of gene g in the
// list of hit j:  nk = sum m_j - (m_first - 1),  cov = (union of the k-mer intervals) + (m_first - 1).
template <int U, int MODE, bool HASQ, bool FAST, bool EMIT, bool WRAP = false>
__device__ __forceinline__ void process_read(const ClassifyParams &P, const uint64_t read, const int lane, const WaveStore st,
                                             const uint32_t slot_cap, const uint32_t tie_cov, const uint32_t tie_nk,
                                             const uint32_t *lsum, const ReadMeta meta, const bool pre, const Raw8 pre_w, const Raw8 pre_q)
{
  constexpr bool POW2 = pm_pow2(MODE);
  constexpr bool SUM = MODE == PM_BV_SUM || MODE == PM_TAB_SUM;
  constexpr bool LSUM = pm_lds(MODE);
  constexpr bool TAB = pm_tab(MODE);
  // LAZY: with the summary in LDS a probe of a non-existent slot costs no memory traffic, so the
  // slot's existence and validity are only evaluated for the (few) probes that match in the table
  constexpr bool LAZY = FAST && LSUM;
  const uint32_t k = P.k;
  const uint32_t L1 = meta.L1, L2 = meta.L2;
  const uint32_t nk1 = L1 >= k ? L1 - k + 1 : 0;
  const uint32_t nk2 = L2 >= k ? L2 - k + 1 : 0;
  const uint32_t P2 = (L1 + 7u) & ~7u;            // packed position of mate 2's first base
  const uint32_t ns = nk2 ? P2 + nk2 : nk1;       // slots are the packed positions [0, ns)

  if (FAST || (!EMIT && P.work == nullptr)) {   // (general kernel over ALL reads, wrap mode: same queue for what does not fit its scratch)
    if (ns > slot_cap) {  // does not fit the LDS specialisation: general kernel
      if (lane == 0) {
        const ClassifyOut *O = out_ptrs(P);
        const uint32_t q = atomicAdd(&O->counters[CTR_LONG], 1u);
        O->long_queue[q] = (uint32_t)read;
        atomicMax(&O->counters[CTR_MAX_SLOTS], ns);
        O->count[read] = 0;
      }
      return;
    }
  }

  // ---- stage the read: 8 bases per lane -> the two code streams + validity ----
  const uint32_t g2 = P2 >> 3;
  const uint32_t n_groups = g2 + ((L2 + 7) >> 3);
  const uint32_t rv_last = (st.rcap >> 3) - 1u;   // 16-bit chunk of `rv` that mirrors chunk 0 of `fw`
  uint32_t my_valid = 0;
  // classify + pack the 8 bases of group gi (already in registers) and write them out
  auto stage_group = [&](const uint32_t gi, const uint64_t w, const uint64_t q) {
    const bool m2 = gi >= g2;
    const uint32_t b = (m2 ? gi - g2 : gi) << 3;
    const uint32_t L = m2 ? L2 : L1;
    uint32_t msb16 = 0, valid8 = 0;
    if (b < L) {
      const uint32_t rem = L - b;
      uint32_t c_lo, c_hi, i_lo, i_hi;
      classify4((uint32_t)w, c_lo, i_lo);
      classify4((uint32_t)(w >> 32), c_hi, i_hi);
      msb16 = (pack4(c_lo) << 8) | pack4(c_hi);           // first base in bits 15:14
      uint32_t inv8 = gather4(i_lo) | (gather4(i_hi) << 4);
      if (HASQ) inv8 |= gather4(qmask4((uint32_t)q, P.mq)) | (gather4(qmask4((uint32_t)(q >> 32), P.mq)) << 4);
      if (rem < 8) inv8 |= 0xFFu << rem;
      valid8 = ~inv8 & 0xFFu;
    }
    // the same 8 codes with the first base in the LOW bits: reverse the bits, swap each pair back
    uint32_t lsb = __builtin_bitreverse32(msb16);         // lands in the high half
    lsb = ((lsb >> 1) & 0x55555555u) | ((lsb & 0x55555555u) << 1);
    reinterpret_cast<uint16_t *>(st.fw)[gi] = (uint16_t)(lsb >> 16);
    reinterpret_cast<uint16_t *>(st.rv)[rv_last - gi] = (uint16_t)msb16;
    reinterpret_cast<uint8_t *>(st.vbits)[gi] = (uint8_t)valid8;
    my_valid += __builtin_popcount(valid8);
  };
  uint32_t gi0 = (uint32_t)lane;
  if (pre) {
    // this lane's first group was fetched one read ahead: no load (and no wait) on this path
    if (gi0 < n_groups) stage_group(gi0, load8_finish(pre_w), HASQ ? load8_finish(pre_q) : 0ull);
    gi0 += 64;
  }
  for (uint32_t gi = gi0; gi < n_groups; gi += 64) {
    Raw8 w, q;
    fetch_group<HASQ>(P, meta, gi, w, q);
    stage_group(gi, load8_finish(w), HASQ ? load8_finish(q) : 0ull);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  // ---- probe every k-mer slot ----------------------------------------------
  const uint64_t kmask = (1ull << k) - 1ull;          // k validity bits
  const uint64_t kmer_mask = (1ull << (2u * k)) - 1ull;
  // slot pp exists and all its k characters are valid (equivalent to the reference's roll/restart
  // walk, kmer_utils.hpp:57-71 + ReadAnalyzer.hpp:51-77)
  auto slot_ok = [&](const uint32_t pp) -> bool {
    const bool act = (pp < nk1) | ((pp - P2) < nk2);
    const uint32_t V = pp >> 6, vs = pp & 63u;
    const uint64_t v0 = st.vbits[V], v1 = st.vbits[V + 1];
    const uint64_t win = (v0 >> vs) | ((v1 << 1) << (63u - vs));
    return act & ((win & kmask) == kmask);
  };
  bool any_hit = false;
  uint32_t my_first = GENE_INF;   // WRAP: smallest valid slot of this lane
  unsigned long long wk_kmers = 0, wk_hits = 0, wk_ids = 0;
  // slot records of the (single) round of the fast kernel live in registers
  uint32_t cur[U], rs[U], re[U];
#pragma unroll
  for (int j = 0; j < U; ++j) { cur[j] = GENE_INF; rs[j] = 0; re[j] = 0; }

  for (uint32_t base = 0; base < ns; base += 64 * U) {
    uint64_t pos[U];
    uint64_t word[U];
    bool ok[U];
#pragma unroll
    for (int j = 0; j < U; ++j) {
      // fast kernel: one round, pp = lane + 64 j for every read, so the addresses and shift
      // amounts below are loop invariant and the 64 j become instruction offsets
      const uint32_t t = (FAST ? 0u : base) + (uint32_t)lane + 64u * j;
      const uint32_t pp = FAST ? t : (t < ns ? t : 0u);
      // (fast kernel: everything is derived from slot U-1 / slot 0 of the lane plus constants, so
      // only two addresses and two shift amounts stay live across reads)
      const uint32_t qU = st.rcap - k - ((uint32_t)lane + 64u * (U - 1));
      const uint32_t q0 = FAST ? qU + 64u * (U - 1 - j) : st.rcap - k - pp;
      const uint32_t *f = FAST ? st.fw + ((uint32_t)lane >> 4) + 4 * j : st.fw + (pp >> 4);
      const uint32_t *r = FAST ? st.rv + (qU >> 4) + 4 * (U - 1 - j) : st.rv + (q0 >> 4);
      const uint32_t d0 = f[0], d1 = f[1], d2 = f[2];
      const uint32_t e0 = r[0], e1 = r[1], e2 = r[2];
      const uint32_t sf = ((FAST ? (uint32_t)lane : pp) & 15u) << 1, sr = ((FAST ? qU : q0) & 15u) << 1;
      const uint64_t x = ((uint64_t)__builtin_amdgcn_alignbit(d2, d1, sf) << 32) | __builtin_amdgcn_alignbit(d1, d0, sf);
      const uint64_t y = ((uint64_t)__builtin_amdgcn_alignbit(e2, e1, sr) << 32) | __builtin_amdgcn_alignbit(e1, e0, sr);
      const uint64_t fwd = y & kmer_mask, rc = ~x & kmer_mask;
      const uint64_t canon = fwd < rc ? fwd : rc;         // KmerBuilder.hpp:49, ReadAnalyzer.hpp:55
      // (LDS-summary mode with a power-of-two size keeps the raw hash: every use below masks the bits it needs)
      const uint64_t hsh = xxh64_u64(canon);
      pos[j] = POW2 ? (LAZY ? hsh : (hsh & P.bf_mask)) : bf_pos_np(hsh, P);
      ok[j] = LAZY ? true : ((FAST || t < ns) && slot_ok(pp));
      if (WRAP && ok[j] && t < my_first) my_first = t;
    }
    if (!FAST && P.work_counters) {
#pragma unroll
      for (int j = 0; j < U; ++j) wk_kmers += ok[j];
    }
    // all probes of this lane are issued before the first use (bloomfilter.h:87-89)
    if (SUM) {
      // summary level first (cache resident): a clear bit proves the filter bit clear
      uint32_t sw[U];
#pragma unroll
      for (int j = 0; j < U; ++j) sw[j] = ok[j] ? P.sum32[(pos[j] >> P.sum_shift) >> 5] : 0u;
#pragma unroll
      for (int j = 0; j < U; ++j) ok[j] = (sw[j] >> ((uint32_t)(pos[j] >> P.sum_shift) & 31u)) & 1u;
    }
    uint32_t okm[U];   // LDS-summary mode: all ones where the summary bit is set, else 0
    if (LSUM) {
      // LDS-resident summary (2^LDS_SUM_LOG2 bits, bit i = OR of filter bits [i << lsum_shift, +2^lsum_shift)):
      // same proof, no memory-system traffic for a miss.  lsum_shift < 32 (index_build.hip).
      uint32_t si[U], sw[U];
#pragma unroll
      for (int j = 0; j < U; ++j) {
        si[j] = __builtin_amdgcn_alignbit((uint32_t)(pos[j] >> 32), (uint32_t)pos[j], P.lsum_shift);   // low 18 bits = summary index
        sw[j] = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(lsum) + ((si[j] >> 3) & (LDS_SUM_BITS / 8 - 4)));
      }
      uint32_t any = 0;
#pragma unroll
      for (int j = 0; j < U; ++j) {
        okm[j] = (uint32_t)__builtin_amdgcn_sbfe((int)sw[j], si[j], 1u);   // v_bfe_i32: bit (si & 31), sign extended
        ok[j] = okm[j] != 0u;
        any |= okm[j];
      }
      if (!__ballot(any != 0u)) break;   // every probe of this read is proven clear
    } else if (FAST && SUM) {
      bool lane_ok = false;
#pragma unroll
      for (int j = 0; j < U; ++j) lane_ok |= ok[j];
      if (!__ballot(lane_ok)) break;   // every probe of this read is proven clear
    }
    bool hit[U];
    bool lane_any = false;
    if (TAB) {
      // ---- position table: one 16-byte bucket answers membership AND the list ----
      // slot = tag(24) | valid | displacement(6) in the high word, multi(1) | overflow(1) | payload(30) in the low
      // word, so membership is ONE compare per slot and nothing else is decoded unless it matched
      const uint4 *tab16 = reinterpret_cast<const uint4 *>(P.tab);
      const uint32_t bmask = (uint32_t)((1ull << P.tab_lg) - 1ull) & (uint32_t)P.bf_mask;   // tab_lg <= 31; bucket = pos & bmask
      const uint32_t tagmask = (uint32_t)(P.bf_mask >> P.tab_lg);
      const uint32_t spare = 1u << P.tab_lg;                                    // one bucket past the table, always empty
      auto want_of = [&](const int j) -> uint32_t {                              // tag, valid, displacement 0
        const uint32_t tag = __builtin_amdgcn_alignbit((uint32_t)(pos[j] >> 32), (uint32_t)pos[j], P.tab_lg) & tagmask;
        return (tag << 8) | 0x80u;
      };
      uint4 bk[U];
#pragma unroll
      for (int j = 0; j < U; ++j) {
        // probes that are already proven clear read the spare bucket behind the table: it is empty,
        // so they neither match nor continue, and nothing below has to look at ok[j] again
        const uint32_t b = (uint32_t)pos[j] & bmask;
        const uint32_t bi = LSUM ? ((b & okm[j]) | (spare & ~okm[j])) : (ok[j] ? b : spare);
        if (!LSUM && P.tab_nt) {
          // a table far beyond the caches: streaming loads (measured +4 % on the 8 GiB table of the
          // 60 000-gene index, but 1.5x slower on an L2-resident table)
          const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(tab16) + bi);
          bk[j] = make_uint4(v.x, v.y, v.z, v.w);
        } else {
          bk[j] = tab16[bi];
        }
      }
      bool more[U];
      bool lane_more = false;
#pragma unroll
      for (int j = 0; j < U; ++j) {
        const uint32_t want = want_of(j);
        const bool match = (bk[j].y == want) | (bk[j].w == want);
        lane_any |= match;
        more[j] = !match & ((bk[j].x & TAB_OVERFLOW) != 0u);   // some key of this home bucket lives further down the path
        lane_more |= more[j];
      }
      // rare: the key may sit behind its (full) home bucket -> walk the probe path
      if (__ballot(lane_more))
        walk_probe_paths<U, true>(tab16, bk, more, lane_any, want_of, [&](const int j, const uint32_t d) { return ((uint32_t)pos[j] + d) & bmask; });
      bool round_any = __ballot(lane_any) != 0ull;
      if (FAST && !round_any) break;
      // something matched: decode.  In LDS-summary mode the slots now have to exist and be valid
      // k-mers as well (their probes were issued unconditionally).
      uint32_t payload[U];
      bool multi[U];
      lane_any = false;
#pragma unroll
      for (int j = 0; j < U; ++j) {
        const uint32_t want = want_of(j);
        const bool m0 = bk[j].y == want, m1 = bk[j].w == want;
        hit[j] = m0 | m1;
        if (LAZY) hit[j] = hit[j] && slot_ok((uint32_t)lane + 64u * j);
        lane_any |= hit[j];
        const uint32_t lo = m0 ? bk[j].x : bk[j].z;
        payload[j] = lo & TAB_PAYLOAD;
        multi[j] = (lo >> 31) != 0u;
      }
      if (!FAST && P.work_counters) {
#pragma unroll
        for (int j = 0; j < U; ++j) wk_hits += hit[j];
      }
      if (LAZY) {
        round_any = __ballot(lane_any) != 0ull;
        if (!round_any) break;
      }
      if (SHK_ABL(P, 1u)) break;
      any_hit |= round_any;
      // multi-gene lists (rare): entry r gives start/len/first gene
      bool lane_multi = false;
#pragma unroll
      for (int j = 0; j < U; ++j) lane_multi |= hit[j] & multi[j];
      if (__ballot(lane_multi)) {
        ListEntry le[U];
#pragma unroll
        for (int j = 0; j < U; ++j) le[j] = P.ent[(hit[j] & multi[j]) ? payload[j] : 0u];
#pragma unroll
        for (int j = 0; j < U; ++j) {
          if (hit[j] & multi[j]) {
            rs[j] = le[j].start;
            re[j] = le[j].len != 0xFFFFu ? le[j].start + le[j].len : P.ent[payload[j] + 1].start;
            cur[j] = le[j].gene0;
          } else {
            rs[j] = 0; re[j] = 0; cur[j] = hit[j] ? (payload[j] & 0xFFFFu) : GENE_INF;
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < U; ++j) { rs[j] = 0; re[j] = 0; cur[j] = hit[j] ? (payload[j] & 0xFFFFu) : GENE_INF; }
      }
      if (!FAST && P.work_counters) {
#pragma unroll
        for (int j = 0; j < U; ++j) wk_ids += hit[j] ? (multi[j] ? re[j] - rs[j] : 1u) : 0u;
      }
    } else {
      if (SUM) {
#pragma unroll
        for (int j = 0; j < U; ++j) word[j] = ok[j] ? P.bf64[pos[j] >> 6] : 0ull;
      } else {
#pragma unroll
        for (int j = 0; j < U; ++j) word[j] = ok[j] ? __builtin_nontemporal_load(P.bf64 + (pos[j] >> 6)) : 0ull;
      }
#pragma unroll
      for (int j = 0; j < U; ++j) {
        hit[j] = (word[j] >> (pos[j] & 63u)) & 1ull;
        lane_any |= hit[j];
      }
      const bool round_any = __ballot(lane_any) != 0ull;
      if (!FAST && P.work_counters) {
#pragma unroll
        for (int j = 0; j < U; ++j) wk_hits += hit[j];
      }
      if (FAST && !round_any) break;  // single round: nothing hit, nothing to record
      if (SHK_ABL(P, 1u)) break;       // ablation: stop after the probes
      any_hit |= round_any;
      // ---- hits: rank -> list entry (bloomfilter.h:90-94).  Unconditional loads
      // from safe addresses (entry 0 for non-hits) so the U chains overlap. ----
      if (round_any) {
        uint32_t rw[U];
#pragma unroll
        for (int j = 0; j < U; ++j) rw[j] = P.rank_w[hit[j] ? (pos[j] >> 6) : 0ull];
        ListEntry le[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
          const uint64_t below = word[j] & ((1ull << (pos[j] & 63u)) - 1ull);
          const uint32_t r = hit[j] ? rw[j] + (uint32_t)__builtin_popcountll(below) : 0u;
          rw[j] = r;
          le[j] = P.ent[r];
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
          if (hit[j]) {
            rs[j] = le[j].start;
            re[j] = le[j].len != 0xFFFFu ? le[j].start + le[j].len : P.ent[rw[j] + 1].start;
            cur[j] = le[j].gene0;
            if (!FAST && P.work_counters) wk_ids += re[j] - rs[j];
          } else {
            rs[j] = 0; re[j] = 0; cur[j] = GENE_INF;
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < U; ++j) { rs[j] = 0; re[j] = 0; cur[j] = GENE_INF; }
      }
    }
    if (!FAST) {
#pragma unroll
      for (int j = 0; j < U; ++j) {
        const uint32_t t = base + lane + 64 * j;
        if (t < ns) { st.rec_start[t] = rs[j]; st.rec_end[t] = re[j]; st.cur[t] = cur[j]; }
      }
    }
  }
  if (!FAST && P.work_counters) {
    const uint32_t a = wave_sum_u32((uint32_t)wk_kmers), b = wave_sum_u32((uint32_t)wk_hits), c = wave_sum_u32((uint32_t)wk_ids);
    if (lane == 0) {
      atomicAdd(&P.work_counters[0], (unsigned long long)a);
      atomicAdd(&P.work_counters[1], (unsigned long long)b);
      atomicAdd(&P.work_counters[2], (unsigned long long)c);
      atomicAdd(&P.work_counters[3], (unsigned long long)(L1 + L2) * (HASQ ? 2ull : 1ull));
    }
  }

  uint32_t best_cov = 0, best_nk = 0, n_best = 0;
  uint32_t best_id[SHK_INLINE_IDS] = {0, 0, 0, 0};
  uint32_t n_emit = 0;
  uint32_t len = 0;
  const uint32_t first_valid = WRAP ? wave_min_u32(my_first) : 0u;   // the read's first valid k-mer (ReadAnalyzer.hpp:51-62)
  if (any_hit && !SHK_ABL(P, 2u)) {   // ablation 2: skip the vote
    // len = number of valid characters of the joined string (ReadAnalyzer.hpp:46-49);
    // only needed for the threshold, i.e. when something hit
    len = wave_sum_u32(my_valid);
    if (!FAST) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    const uint32_t out_base = EMIT ? P.gene_off[read] : 0u;
    // ---- k-way merge over the hit lists, ascending gene id -------------------
    for (;;) {
      uint32_t mymin = GENE_INF;
      if (FAST) {
#pragma unroll
        for (int j = 0; j < U; ++j) mymin = cur[j] < mymin ? cur[j] : mymin;
      } else {
        for (uint32_t t = lane; t < ns; t += 64) {
          const uint32_t g = st.cur[t];
          mymin = g < mymin ? g : mymin;
        }
      }
      const uint32_t g = wave_min_u32(mymin);
      if (g == GENE_INF) break;
      // Gene g's hits as bit masks over the packed positions (one ballot per 64 slots).  Its
      // coverage k + sum min(k, p_j - p_(j-1))  (ReadAnalyzer.hpp:56-62,:79-86) is the size of the union
      // of the intervals [p_j, p_j + k): position x is covered iff a hit lies in (x-k, x].  Lane l of
      // chunk c looks at the 64 positions ending at x = 64c + l -- the chunk's mask shifted up, the
      // previous chunk's mask shifted down -- and tests the top k bits; the covered positions are
      // counted by popcounts of ballots on the scalar unit.  Mate 2 starts at P2 >= L1 and mate 1's
      // last slot is L1-k, so an interval never reaches into the other mate: same clamp as the
      // reference's joined coordinates.
      uint32_t nk = 0, cov = 0;
      const uint64_t kthr = 1ull << (64u - k);
      auto cover = [&](const uint64_t Hc, const uint64_t Hp) -> uint32_t {
        const uint64_t t = (Hc << (63u - (uint32_t)lane)) | ((Hp >> 1) >> (uint32_t)lane);
        return (uint32_t)__builtin_popcountll(__ballot(t >= kthr));
      };
      if (FAST) {
        uint64_t H[U];
        bool more_ids = false;
#pragma unroll
        for (int j = 0; j < U; ++j) {
          const bool h = cur[j] == g;
          H[j] = __ballot(h);
          rs[j] += h ? 1u : 0u;                // advance this slot's cursor past g
          more_ids |= h & (rs[j] < re[j]);
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
          nk += (uint32_t)__builtin_popcountll(H[j]);
          cov += cover(H[j], j ? H[j - 1] : 0ull);
        }
        cov += cover(0ull, H[U - 1]);
        if (__ballot(more_ids)) {              // multi-gene lists only
#pragma unroll
          for (int j = 0; j < U; ++j)
            if ((H[j] >> lane) & 1ull) cur[j] = rs[j] < re[j] ? (uint32_t)P.ids[rs[j]] : GENE_INF;
        } else {
#pragma unroll
          for (int j = 0; j < U; ++j) cur[j] = ((H[j] >> lane) & 1ull) ? GENE_INF : cur[j];
        }
      } else {
        uint64_t Hprev = 0;
        for (uint32_t tb = 0; tb < ns; tb += 64) {
          const uint32_t t = tb + lane;
          const bool in = t < ns;
          const uint32_t c_cur = in ? st.cur[t] : GENE_INF;
          const bool h = c_cur == g;
          const uint64_t Hc = __ballot(h);
          nk += (uint32_t)__builtin_popcountll(Hc);
          cov += cover(Hc, Hprev);
          uint32_t mult = 0;
          if (h) {
            uint32_t c_rs = st.rec_start[t] + 1u;
            mult = 1;
            if (WRAP) {
              const uint32_t c_end = st.rec_end[t];
              while (c_rs < c_end && (uint32_t)P.ids[c_rs] == g) { ++c_rs; ++mult; }   // the id again: same gene, same k-mer
            }
            st.rec_start[t] = c_rs;
            st.cur[t] = c_rs < st.rec_end[t] ? (uint32_t)P.ids[c_rs] : GENE_INF;
          }
          if (WRAP) {
            nk += wave_sum_u32(h ? mult - 1u : 0u);
            const uint32_t fl = first_valid - tb;            // wave-uniform
            if (fl < 64u && ((Hc >> fl) & 1ull)) {
              const uint32_t extra = (uint32_t)__builtin_amdgcn_readlane((int)mult, (int)fl) - 1u;
              cov += extra;
              nk -= extra;
            }
          }
          Hprev = Hc;
        }
        cov += cover(0ull, Hprev);
      }
      if (EMIT) {
        if (cov == tie_cov && nk == tie_nk) {
          if (lane == 0) P.gene_ids[out_base + n_emit] = (uint16_t)g;
          ++n_emit;
        }
      } else {
        // arg-max with ties in ascending gene order (ReadAnalyzer.hpp:90-102).
        // Written as selects on two predicates: hipcc (ROCm 7.2) drops the
        // best_id[0] update on the `cov == best && nk > best_nk` edge when this
        // is an if/else-if chain over wave-uniform (SGPR) values.
        const bool gt = (cov > best_cov) | ((cov == best_cov) & (nk > best_nk));
        const bool eq = (cov == best_cov) & (nk == best_nk);
        best_id[0] = gt ? g : best_id[0];
#pragma unroll
        for (int i = 1; i < SHK_INLINE_IDS; ++i) best_id[i] = (eq & (n_best == (uint32_t)i)) ? g : best_id[i];
        n_best = gt ? 1u : (eq ? n_best + 1u : n_best);
        best_cov = gt ? cov : best_cov;
        best_nk = gt ? nk : best_nk;
      }
    }
  }
  if (EMIT) return;

  // ---- threshold + --single (ReadAnalyzer.hpp:104) ---------------------------
  uint32_t n_out = 0;
  if (n_best > 0 && (double)best_cov >= P.c * (double)len && (!P.single || n_best == 1)) n_out = n_best;
  if (lane == 0) {
    // No same-address atomics here: the association total comes from the scan
    // of `count`, the per-gene counts from gather_inline_kernel (wave-aggregated).
    const ClassifyOut *O = out_ptrs(P);
    O->count[read] = n_out;
    if (n_out > 0) {
      uint16_t *o = O->inl + read * SHK_INLINE_IDS;
#pragma unroll
      for (int i = 0; i < SHK_INLINE_IDS; ++i)
        if ((uint32_t)i < n_out) o[i] = (uint16_t)best_id[i];
      if (n_out > SHK_INLINE_IDS) {
        const uint32_t q = atomicAdd(&O->counters[CTR_TIE], 1u);
        O->tie_queue[3 * q + 0] = (uint32_t)read;
        O->tie_queue[3 * q + 1] = best_cov;
        O->tie_queue[3 * q + 2] = best_nk;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// fast kernel: everything per wave lives in LDS; slot capacity 64*U
// ---------------------------------------------------------------------------
template <int MODE, int U>
struct FastGeom {
  static constexpr int WAVES = pm_lds(MODE) ? 8 : CF_WAVES;   // 512-thread workgroups share one LDS summary
  static constexpr int THREADS = WAVES * 64;
  // LDS-summary mode: four 512-thread workgroups per CU (4 x 34 KiB of LDS) = 8 waves per SIMD, which
  // needs <= 64 VGPRs; the specialisations for more than 320 slots do not fit that and run 6 waves
  static constexpr int MIN_WAVES_PER_SIMD = pm_lds(MODE) ? (U <= 5 ? 8 : 6) : ((MODE == PM_TAB || MODE == PM_TAB_SUM || MODE == PM_TAB_MOD) && U <= 5 ? 8 : 1);
  static constexpr uint32_t SUM_WORDS64 = pm_lds(MODE) ? LDS_SUM_BITS / 64 : 0;
};

template <int U, int MODE, bool HASQ>
__global__ __launch_bounds__((FastGeom<MODE, U>::THREADS), (FastGeom<MODE, U>::MIN_WAVES_PER_SIMD)) void classify_fast_kernel(const ClassifyParams P)
{
  using G = FastGeom<MODE, U>;
  constexpr uint32_t S = 64 * U;
  constexpr uint32_t WORDS = stage_words_for(S);   // slot records stay in registers
  __shared__ uint64_t lds[G::SUM_WORDS64 + G::WAVES * WORDS];
  const int lane = threadIdx.x & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (pm_tab(MODE) && P.uni_flag && P.uni_flag[0] == 1u) return;   // a uniform batch: classify_uni_kernel has it
  const uint32_t *lsum = nullptr;
  if (pm_lds(MODE)) {
    // stage the summary: 32 KiB, 16 bytes per thread per pass, once per (persistent) workgroup
    const uint4 *src = reinterpret_cast<const uint4 *>(P.lsum32);
    uint4 *dst = reinterpret_cast<uint4 *>(lds);
    for (uint32_t i = threadIdx.x; i < LDS_SUM_BITS / 128; i += G::THREADS) dst[i] = src[i];
    __syncthreads();
    lsum = reinterpret_cast<const uint32_t *>(lds);
  }
  uint64_t *base = lds + G::SUM_WORDS64 + wave * WORDS;
  WaveStore st;
  st.fw = reinterpret_cast<uint32_t *>(base);
  st.rv = st.fw + code_dwords_for(S);
  st.rcap = stage_cap_bases(S);
  st.vbits = base + code_dwords_for(S);
  st.rec_start = nullptr;
  st.rec_end = nullptr;
  st.cur = nullptr;
  // software pipeline over this wave's reads: the offsets are fetched two reads ahead and the
  // bases one read ahead (all ordinary loads), so their HBM latency hides under the hashing of the
  // current read
  // (ablation: the in-place base loads cost 4.4 of 14 ms on the all-miss workload)
  // (read indices fit 32 bits: shk_classify* refuses batches of 2^32-1 reads or more)
  const uint32_t stride = gridDim.x * G::WAVES, n32 = (uint32_t)P.n;
  uint32_t read = blockIdx.x * G::WAVES + wave;
  if (read >= n32) return;
  ReadMeta m_cur = fetch_meta(P, read);
  Raw8 w_cur, q_cur;
  fetch_group<HASQ>(P, m_cur, (uint32_t)lane, w_cur, q_cur);
  retire_loads(w_cur);   // same reason as at the loop end
  retire_loads(q_cur);
  // nxt / nn saturate at n32 (no wrap-around near 2^32)
  uint32_t nxt = n32 - read > stride ? read + stride : n32;
  ReadMeta m_nxt = fetch_meta(P, nxt < n32 ? nxt : read);
  for (;;) {
    Raw8 w_nxt = Raw8{0u, 0u, 0u, 0u}, q_nxt = Raw8{0u, 0u, 0u, 0u};
    const bool have_nxt = nxt < n32;
    if (have_nxt) fetch_group<HASQ>(P, m_nxt, (uint32_t)lane, w_nxt, q_nxt);
    const uint32_t nn = (have_nxt && n32 - nxt > stride) ? nxt + stride : n32;
    ReadMetaRaw r_nn = fetch_meta_issue(P, nn < n32 ? nn : read);          // clamped index
    process_read<U, MODE, HASQ, true, false>(P, read, lane, st, S, 0u, 0u, lsum, m_cur, true, w_cur, q_cur);
    if (!have_nxt) break;
    // The prefetched bases landed long ago.  Retire them HERE and hand the compiler plain register
    // values: otherwise it carries "a load may be pending" around the loop and, because the number of
    // younger loads is branch dependent, protects the first use with s_waitcnt vmcnt(0) -- right
    // behind the next prefetch, which would serialise the pipeline again.
    retire_loads(w_nxt);
    retire_loads(q_nxt);
    retire_meta(r_nn);
    read = nxt; m_cur = m_nxt; w_cur = w_nxt; q_cur = q_nxt;
    nxt = nn; m_nxt = meta_finish(r_nn);
  }
}

// Does every read of the batch have the same length per mate, and does such a read fit `slot_cap` slots and 64 staging
// groups?  flag = {1 | 0, L1, L2}.  One pass over the offsets (16 B per pair), a few tens of microseconds for 10 M pairs.
__global__ __launch_bounds__(1024) void uniform_check_kernel(const ClassifyParams P, uint32_t slot_cap, uint32_t *__restrict__ flag)
{
  // flag[3] (scratch) counts violations; the last workgroup to finish writes the verdict
  __shared__ uint32_t bad_s;
  if (threadIdx.x == 0) bad_s = 0;
  __syncthreads();
  const uint64_t n = P.n;
  const uint64_t L1 = n ? P.off1[1] - P.off1[0] : 0, L2 = (n && P.seq2) ? P.off2[1] - P.off2[0] : 0;
  __shared__ uint32_t max_s[2];
  if (threadIdx.x == 0) { max_s[0] = 0; max_s[1] = 0; }
  __syncthreads();
  uint32_t bad = 0;
  uint64_t mx1 = 0, mx2 = 0;   // the longest mates: the ragged instantiation stages the whole batch in their layout
  if (blockIdx.x == 0 && threadIdx.x == 0 && n) bad |= (P.off1[0] != 0) | (P.seq2 && P.off2[0] != 0);
  // (a read's end is the next one's start.  Offsets on a 16-byte boundary -- whole allocations are --: two reads per lane from one
  //  16-byte load, the third offset from the next lane; else one read per lane from an 8-byte load.  Lane 63's last offset, and the
  //  last lanes' at the end of the batch, come from memory.)
  const int lane_u = threadIdx.x & 63;
  const bool wide = ((reinterpret_cast<uintptr_t>(P.off1) | (P.seq2 ? reinterpret_cast<uintptr_t>(P.off2) : 0)) & 15u) == 0;
  if (wide) {
    // (four strides per trip, every load of the trip issued before the first is used: 160 MB of offsets per 10 M pairs want more
    //  in flight than one 16-byte load per lane)
    constexpr int TRIP = 4;
    const uint64_t stride = 2ull * gridDim.x * blockDim.x;
    auto mate = [&](const uint64_t *__restrict__ off, const uint64_t i_first, const uint64_t L, uint64_t &mx) {
      // per stride u: entries off[i], off[i + 1], off[i + 2] -> the reads i and i + 1 (i even)
      ulonglong2 a[TRIP];
      uint64_t tail[TRIP];
#pragma unroll
      for (int u = 0; u < TRIP; ++u) {
        const uint64_t i = i_first + (uint64_t)u * stride;
        a[u] = make_ulonglong2(0, 0);
        tail[u] = 0;
        if (i + 1 <= n) a[u] = *reinterpret_cast<const ulonglong2 *>(off + i);
        if ((lane_u == 63 || i + 3 > n) && i + 2 <= n) tail[u] = off[i + 2];
      }
#pragma unroll
      for (int u = 0; u < TRIP; ++u) {
        const uint64_t i = i_first + (uint64_t)u * stride;
        uint64_t c = __shfl_down(a[u].x, 1);
        if (lane_u == 63 || i + 3 > n) c = tail[u];
        const uint64_t la = i < n ? a[u].y - a[u].x : L, lb = i + 1 < n ? c - a[u].y : L;
        bad |= (la != L) | (lb != L);
        mx = la > mx ? la : mx;
        mx = lb > mx ? lb : mx;
      }
    };
    for (uint64_t i0 = 2ull * blockIdx.x * blockDim.x; i0 < n; i0 += TRIP * stride) {
      const uint64_t i = i0 + 2ull * threadIdx.x;
      mate(P.off1, i, L1, mx1);
      if (P.seq2) mate(P.off2, i, L2, mx2);
    }
  } else {
    for (uint64_t i0 = (uint64_t)blockIdx.x * blockDim.x; i0 < n; i0 += (uint64_t)gridDim.x * blockDim.x) {
      const uint64_t i = i0 + threadIdx.x;
      const bool valid = i < n;
      const uint64_t a1 = P.off1[valid ? i : n];
      uint64_t b1 = __shfl_down(a1, 1);
      if (lane_u == 63) b1 = P.off1[valid ? i + 1 : n];
      const uint64_t l1 = valid ? b1 - a1 : L1;
      bad |= l1 != L1;
      mx1 = l1 > mx1 ? l1 : mx1;
      if (P.seq2) {
        const uint64_t a2 = P.off2[valid ? i : n];
        uint64_t b2 = __shfl_down(a2, 1);
        if (lane_u == 63) b2 = P.off2[valid ? i + 1 : n];
        const uint64_t l2 = valid ? b2 - a2 : L2;
        bad |= l2 != L2;
        mx2 = l2 > mx2 ? l2 : mx2;
      }
    }
  }
  // (per wave, then per workgroup)
  const bool wave_bad = __ballot(bad != 0u) != 0ull;
  if (wave_bad) {
    for (int d = 32; d; d >>= 1) {
      const uint64_t o1 = __shfl_xor(mx1, d), o2 = __shfl_xor(mx2, d);
      mx1 = o1 > mx1 ? o1 : mx1;
      mx2 = o2 > mx2 ? o2 : mx2;
    }
    if ((threadIdx.x & 63) == 0) {
      atomicOr(&bad_s, 1u);
      // (clipped: a mate of 2^31 bases or more fits no specialisation and the kernel clamps the layout anyway)
      atomicMax(&max_s[0], (uint32_t)(mx1 < 0x7FFFFFFFull ? mx1 : 0x7FFFFFFFull));
      atomicMax(&max_s[1], (uint32_t)(mx2 < 0x7FFFFFFFull ? mx2 : 0x7FFFFFFFull));
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (bad_s) {
      atomicOr(&flag[3], 1u);
      atomicMax(&flag[5], max_s[0]);
      atomicMax(&flag[6], max_s[1]);
    }
    __threadfence();
    const uint32_t done = atomicAdd(&flag[4], 1u) + 1u;
    if (done == gridDim.x) {
      const uint32_t any_bad = atomicOr(&flag[3], 0u);
      const uint32_t k = P.k;
      const uint64_t nk1 = L1 >= k ? L1 - k + 1 : 0, nk2 = L2 >= k ? L2 - k + 1 : 0;
      const uint64_t ns = nk2 ? ((L1 + 7) & ~7ull) + nk2 : nk1;
      const uint64_t groups = ((L1 + 7) >> 3) + ((L2 + 7) >> 3);
      const bool ok = n && !any_bad && ns <= slot_cap && groups <= (slot_cap > 512u ? 128u : 64u) && L1 < (1ull << 31) && L2 < (1ull << 31);
      // uniform: the one length per mate; else the longest mates (threads that saw only reads of the first read's lengths did not
      // report: those lengths count as well)
      const uint32_t f1 = (uint32_t)(L1 < 0x7FFFFFFFull ? L1 : 0x7FFFFFFFull), f2 = (uint32_t)(L2 < 0x7FFFFFFFull ? L2 : 0x7FFFFFFFull);
      const uint32_t m1 = atomicMax(&flag[5], f1), m2 = atomicMax(&flag[6], f2);
      flag[1] = any_bad ? (m1 > f1 ? m1 : f1) : f1;
      flag[2] = any_bad ? (m2 > f2 ? m2 : f2) : f2;
      __threadfence();
      // 3: mixed lengths, and the launcher has the three-pairs kernel by offsets in the stream (P.tro) for a layout of these longest
      // mates -- exactly tri_applies() of classify_uni.hpp
      uint32_t verdict = ok ? 1u : 0u;
      // (P.tro == 2, SHK_FORCE_TRO=1: a batch of one length takes that kernel too -- same results; what the offsets cost on their own)
      if (n && P.tro && ((!ok && any_bad) || (ok && P.tro == 2u))) {
        const uint32_t a1 = flag[1], a2 = P.seq2 ? flag[2] : 0u;
        const uint32_t d1 = (a1 + 15u) >> 4, d2 = (a2 + 15u) >> 4;
        const uint32_t q2 = a2 >= k ? a2 - k + 1u : 0u, q1 = a1 >= k ? a1 - k + 1u : 0u;
        const uint32_t nsl = q2 ? (d1 << 4) + q2 : q1;
        if (a1 != 0u && d1 + d2 <= 21u && nsl <= slot_cap) verdict = 3u;
      }
      flag[0] = verdict;
    }
  }
}

// ---- a batch of mixed lengths, class by class (classify_uni_kernel's CLS instantiation) ---------------------------------------
// Three passes over the offsets (16 B per pair each) behind uniform_check_kernel, all of which return at once unless the batch is
// ragged, its longest pair fits the specialisation and (L1 + 1) (L2 + 1) classes fit the histogram:
//   class_hist_kernel     pairs per class c = l1 (L2 + 1) + l2
//   class_plan_kernel     one workgroup: the classes' first entries (the scatter's cursors), the list of non-empty classes {l1, l2,
//                         first entry, entries}, the class each of the CLS_SHARES shares of the entries starts in; the verdict --
//                         flag[0] = 2 -- only if a non-empty class holds cls_min_fill pairs on average (a batch of a million
//                         classes stays ragged)
//   class_scatter_kernel  a pair's entry {o1, o2 | read, safe} goes to its class's next place
// Counting is done in LDS: a workgroup takes a contiguous stretch of the batch and keeps the counters of a WINDOW of classes -- the
// CLS_WIN classes nearest the longest mates' lengths, 151 x 151 of them for 2 x 150 bp, all there are -- and touches a class's
// global counter once (one atomic per pair, or per wave and class, took 3-4 ms per pass for 10 M pairs: a trimmed sample has one
// class with most of its pairs, and same-address atomics queue up).  The scatter reserves a workgroup's places per class the same
// way -- count in LDS, one atomic per class for the range, then places handed out in LDS.  A pair outside the window (reads far
// shorter than the longest: rare) goes to its global counter directly.
constexpr uint32_t CLS_WIN = 36864;        // counters a workgroup keeps: 144 KiB
constexpr uint32_t CLS_WIN_L2 = 192;       // mate-2 lengths of the window (paired batches)

__device__ __forceinline__ bool class_path_ok(const ClassifyParams &P, const uint32_t slot_cap, const uint32_t *flag, uint32_t &L1, uint32_t &L2)
{
  if (flag[0] != 0u) return false;
  L1 = flag[1];
  L2 = P.seq2 ? flag[2] : 0u;
  const uint32_t k = P.k;
  const uint64_t nk1 = L1 >= k ? L1 - k + 1 : 0, nk2 = L2 >= k ? L2 - k + 1 : 0;
  const uint64_t ns = nk2 ? (((uint64_t)L1 + 7) & ~7ull) + nk2 : nk1;
  const uint64_t groups = (((uint64_t)L1 + 7) >> 3) + (((uint64_t)L2 + 7) >> 3);
  const bool fits = ns <= slot_cap && groups <= (slot_cap > 512u ? 128u : 64u);
  return fits && ((uint64_t)L1 + 1) * ((uint64_t)L2 + 1) <= (uint64_t)P.cls_cap;
}

// the pair's two lengths (and offsets)
__device__ __forceinline__ void pair_lengths(const ClassifyParams &P, const uint64_t i, uint64_t &o1, uint64_t &o2, uint32_t &l1, uint32_t &l2)
{
  o1 = P.off1[i];
  l1 = (uint32_t)(P.off1[i + 1] - o1);
  o2 = 0;
  l2 = 0;
  if (P.seq2) { o2 = P.off2[i]; l2 = (uint32_t)(P.off2[i + 1] - o2); }
}

// the window of classes a workgroup counts in LDS: lengths (b1 .. L1) x (b2 .. L2)
struct ClassWindow {
  uint32_t W, w1, w2, b1, b2;
  __device__ ClassWindow(const uint32_t L1, const uint32_t L2)
  {
    W = L2 + 1u;
    w2 = W < CLS_WIN_L2 ? W : CLS_WIN_L2;
    const uint32_t room = CLS_WIN / w2;
    w1 = L1 + 1u < room ? L1 + 1u : room;
    b1 = L1 + 1u - w1;
    b2 = L2 + 1u - w2;
  }
  __device__ uint32_t size() const { return w1 * w2; }
  __device__ bool holds(const uint32_t l1, const uint32_t l2) const { return l1 >= b1 && l2 >= b2; }
  __device__ uint32_t local(const uint32_t l1, const uint32_t l2) const { return (l1 - b1) * w2 + (l2 - b2); }
  __device__ uint32_t global_of(const uint32_t j) const { const uint32_t r = j / w2; return (r + b1) * W + (j - r * w2 + b2); }
};

__global__ __launch_bounds__(1024) void class_hist_kernel(const ClassifyParams P, uint32_t slot_cap, const uint32_t *__restrict__ flag, uint32_t *__restrict__ hist)
{
  uint32_t L1, L2;
  if (!class_path_ok(P, slot_cap, flag, L1, L2)) return;
  __shared__ uint32_t lh[CLS_WIN];
  const ClassWindow win(L1, L2);
  const uint32_t nw = win.size();
  for (uint32_t j = threadIdx.x; j < nw; j += blockDim.x) lh[j] = 0u;
  __syncthreads();
  const uint64_t n = P.n, per = (n + gridDim.x - 1) / gridDim.x;
  const uint64_t lo = (uint64_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
  for (uint64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    uint64_t o1, o2;
    uint32_t l1, l2;
    pair_lengths(P, i, o1, o2, l1, l2);
    if (win.holds(l1, l2)) atomicAdd(&lh[win.local(l1, l2)], 1u);
    else atomicAdd(&hist[l1 * win.W + l2], 1u);
  }
  __syncthreads();
  for (uint32_t j = threadIdx.x; j < nw; j += blockDim.x) {
    const uint32_t v = lh[j];
    if (v) atomicAdd(&hist[win.global_of(j)], v);
  }
}

__global__ __launch_bounds__(1024) void class_plan_kernel(const ClassifyParams P, uint32_t slot_cap, uint32_t *__restrict__ flag, const uint32_t *__restrict__ hist,
                                                          uint32_t *__restrict__ cursor, uint4 *__restrict__ list, uint32_t *__restrict__ share_first)
{
  uint32_t L1, L2;
  if (!class_path_ok(P, slot_cap, flag, L1, L2)) return;
  const uint32_t classes = (L1 + 1u) * (L2 + 1u), W = L2 + 1u;
  // (the counts are walked twice by single threads: from LDS when they fit)
  __shared__ uint32_t lh[CLS_WIN];
  const bool staged = classes <= CLS_WIN;
  if (staged) {
    for (uint32_t c = threadIdx.x; c < classes; c += blockDim.x) lh[c] = hist[c];
    __syncthreads();
  }
  auto pairs_of = [&](const uint32_t c) -> uint32_t { return staged ? lh[c] : hist[c]; };
  const uint32_t per = (classes + blockDim.x - 1u) / blockDim.x;
  const uint32_t c_lo = threadIdx.x * per < classes ? threadIdx.x * per : classes, c_hi = c_lo + per < classes ? c_lo + per : classes;
  uint32_t sum_n = 0, sum_c = 0;
  for (uint32_t c = c_lo; c < c_hi; ++c) {
    const uint32_t h = pairs_of(c);
    sum_n += h;
    sum_c += h ? 1u : 0u;
  }
  // exclusive scan of (sum_n, sum_c) over the workgroup: within the wave by shuffles, the 16 wave totals by every thread
  __shared__ uint32_t wave_n[16], wave_c[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t inc_n = sum_n, inc_c = sum_c;
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t a = (uint32_t)__shfl_up((int)inc_n, d), b = (uint32_t)__shfl_up((int)inc_c, d);
    if (lane >= d) { inc_n += a; inc_c += b; }
  }
  if (lane == 63) { wave_n[wave] = inc_n; wave_c[wave] = inc_c; }
  __syncthreads();
  uint32_t run_n = inc_n - sum_n, run_c = inc_c - sum_c, tot_n = 0, tot_c = 0;
  for (int w = 0; w < 16; ++w) {
    if (w < wave) { run_n += wave_n[w]; run_c += wave_c[w]; }
    tot_n += wave_n[w];
    tot_c += wave_c[w];
  }
  // (tot_n == n: every pair of a batch whose longest pair fits has a class)
  if ((uint64_t)tot_n != P.n || (uint64_t)tot_c * P.cls_min_fill > P.n) return;
  // the cursors, the list of non-empty classes, and the class every share of the list of entries starts in
  const uint32_t share_len = (uint32_t)((P.n + CLS_SHARES - 1u) / CLS_SHARES);
  for (uint32_t c = c_lo; c < c_hi; ++c) {
    const uint32_t h = pairs_of(c);
    cursor[c] = run_n;
    if (h) {
      const uint32_t l1 = c / W;
      list[run_c++] = make_uint4(l1, c - l1 * W, run_n, h);
    }
    run_n += h;
  }
  __threadfence();
  __syncthreads();
  // (by search, every thread a few shares: a trimmed sample's untouched pairs are one class that two thirds of the shares start in)
  // (a thread takes a run of consecutive shares: one search, then steps forward along the sorted list)
  const uint32_t per_thread = (CLS_SHARES + blockDim.x - 1u) / blockDim.x;
  uint32_t lo = 0;
  for (uint32_t i = 0, sh = threadIdx.x * per_thread; i < per_thread && sh < CLS_SHARES && (uint64_t)sh * share_len < P.n; ++i, ++sh) {
    const uint32_t e = sh * share_len;
    if (i == 0) {
      uint32_t hi = tot_c;                          // the last class that starts at or before entry e
      while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (list[mid].z <= e) lo = mid; else hi = mid;
      }
    } else {
      while (lo + 1u < tot_c && list[lo + 1u].z <= e) ++lo;
    }
    share_first[sh] = lo;
  }
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) flag[0] = 2u;
}

__global__ __launch_bounds__(1024) void class_scatter_kernel(const ClassifyParams P, const uint32_t *__restrict__ flag, uint32_t *__restrict__ cursor,
                                                             uint4 *__restrict__ entries)
{
  if (flag[0] != 2u) return;
  __shared__ uint32_t lh[CLS_WIN];
  const ClassWindow win(flag[1], P.seq2 ? flag[2] : 0u);
  const uint32_t nw = win.size();
  for (uint32_t j = threadIdx.x; j < nw; j += blockDim.x) lh[j] = 0u;
  __syncthreads();
  const uint64_t n = P.n, per = (n + gridDim.x - 1) / gridDim.x;
  const uint64_t lo = (uint64_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
  const uint64_t end1 = P.off1[n], end2 = P.seq2 ? P.off2[n] : ~0ull;
  // the stretch's pairs per class; its places per class, reserved with one atomic each; the places handed out
  for (uint64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    uint64_t o1, o2;
    uint32_t l1, l2;
    pair_lengths(P, i, o1, o2, l1, l2);
    if (win.holds(l1, l2)) atomicAdd(&lh[win.local(l1, l2)], 1u);
  }
  __syncthreads();
#pragma unroll 4
  for (uint32_t j = threadIdx.x; j < nw; j += blockDim.x) {
    const uint32_t v = lh[j];
    if (v) lh[j] = atomicAdd(&cursor[win.global_of(j)], v);
  }
  __syncthreads();
  for (uint64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    uint64_t o1, o2;
    uint32_t l1, l2;
    pair_lengths(P, i, o1, o2, l1, l2);
    const uint32_t e = win.holds(l1, l2) ? atomicAdd(&lh[win.local(l1, l2)], 1u) : atomicAdd(&cursor[l1 * win.W + l2], 1u);
    // (.y of the second half: every 8-base group of the class's layout, and the 11 bytes behind its first, lie inside the buffers --
    //  three unconditional aligned dwords per group then, see classify_uni_kernel's fetch_groups)
    entries[2ull * e] = make_uint4((uint32_t)o1, (uint32_t)(o1 >> 32), (uint32_t)o2, (uint32_t)(o2 >> 32));
    entries[2ull * e + 1ull] = make_uint4((uint32_t)i, (o1 + l1 + 16u <= end1 && o2 + l2 + 16u <= end2) ? 1u : 0u, 0u, 0u);
  }
}

// ---------------------------------------------------------------------------
// general kernel: same algorithm, per-wave storage in a global scratch slice,
// k-mer slots processed in rounds of 64*U; used for reads that exceed the
// fast kernel's capacity (MAIN) and to write out tie lists longer than
// SHK_INLINE_IDS (EMIT).  Work items come from a queue.
// ---------------------------------------------------------------------------
template <bool POW2, bool HASQ, bool EMIT, bool WRAP>
__global__ __launch_bounds__(CF_THREADS) void classify_general_kernel(const ClassifyParams P)
{
  constexpr int U = 4;
  const int lane = threadIdx.x & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint64_t gw = (uint64_t)blockIdx.x * CF_WAVES + wave;
  uint64_t *base = P.scratch + gw * P.scratch_stride_words;
  const uint32_t S = P.scratch_slots;   // multiple of 64
  WaveStore st;
  st.fw = reinterpret_cast<uint32_t *>(base);
  st.rv = st.fw + code_dwords_for(S);
  st.rcap = stage_cap_bases(S);
  st.vbits = base + code_dwords_for(S);
  st.rec_start = reinterpret_cast<uint32_t *>(base + stage_words_for(S));
  st.rec_end = st.rec_start + S;
  st.cur = st.rec_end + S;
  const uint64_t stride = (uint64_t)gridDim.x * CF_WAVES;
  // the number of queued items may only be known on the device (tie queue: no host round trip)
  const uint64_t n_work = P.work_count ? (uint64_t)__builtin_amdgcn_readfirstlane(*P.work_count) : P.n_work;
  if (EMIT && P.flags && P.flags[CTR_OVERFLOW]) return;   // gene_ids too small: the host grows it and relaunches
  for (uint64_t w = gw; w < n_work; w += stride) {
    uint64_t read = w;
    uint32_t tc = 0, tn = 0;
    if (EMIT) {
      read = P.work[3 * w];
      tc = P.work[3 * w + 1];
      tn = P.work[3 * w + 2];
    } else if (P.work) {
      read = P.work[w];
    }
    process_read<U, POW2 ? PM_BV : PM_BV_MOD, HASQ, false, EMIT, WRAP>(P, read, lane, st, S, tc, tn, nullptr, fetch_meta(P, read), false, Raw8{0u, 0u, 0u, 0u}, Raw8{0u, 0u, 0u, 0u});
  }
}

// copy the inline ids of reads with 1..SHK_INLINE_IDS genes into the CSR result (reads with more
// genes are written by the general kernel in EMIT mode)
constexpr int GI_THREADS = 256;
__global__ __launch_bounds__(GI_THREADS) void gather_inline_kernel(const uint32_t *__restrict__ count, const uint16_t *__restrict__ inl,
                                                                   const uint32_t *__restrict__ gene_off, uint16_t *__restrict__ gene_ids, uint64_t n,
                                                                   const uint32_t *__restrict__ flags)
{
  if (flags[CTR_OVERFLOW]) return;   // gene_ids too small for this batch: the host grows it and relaunches
  for (uint64_t i = (uint64_t)blockIdx.x * GI_THREADS + threadIdx.x; i < n; i += (uint64_t)gridDim.x * GI_THREADS) {
    const uint32_t c = count[i];
    if (c == 0 || c > SHK_INLINE_IDS) continue;
    const uint32_t o = gene_off[i];
    const uint2 v = *reinterpret_cast<const uint2 *>(inl + i * SHK_INLINE_IDS);
    gene_ids[o] = (uint16_t)v.x;
    if (c > 1) gene_ids[o + 1] = (uint16_t)(v.x >> 16);
    if (c > 2) gene_ids[o + 2] = (uint16_t)v.y;
    if (c > 3) gene_ids[o + 3] = (uint16_t)(v.y >> 16);
  }
}

// the scan's 64-bit total -> counters (read by the host through pinned memory) + capacity check
__global__ void finalize_total_kernel(const uint64_t *__restrict__ total, uint32_t *__restrict__ counters, uint64_t gene_ids_cap)
{
  const uint64_t t = *total;
  counters[CTR_ASSOC_LO] = (uint32_t)t;
  counters[CTR_ASSOC_HI] = (uint32_t)(t >> 32);
  counters[CTR_OVERFLOW] = t > gene_ids_cap ? 1u : 0u;
}

// Per-gene number of assigned reads (the quantity all-reduced across GPUs): histogram of the batch's
// final gene_ids.  Gene counters are hot (one gene can own most reads) and same-address global
// atomics serialise at ~12 ns each, so the counts are combined twice before they reach HBM: equal
// genes inside a wave by ballot, then per workgroup in an LDS hash table that is flushed once.
// Runs once per batch, after every kernel that writes gene_ids; skipped when the batch has to be
// finished by the host's slow path (overflow, or queued long reads the launch did not expect), which
// then runs it itself -- so a batch is never counted twice.
constexpr uint32_t GI_TABLE = 2048;   // LDS histogram slots per workgroup
// DIRECT (an index of thousands of genes): a workgroup's few thousand associations hardly ever repeat a gene, the table would only
// fill up and pass everything on -- equal genes inside a wave are combined, then the counter in HBM is touched at once.
template <bool DIRECT>
__global__ __launch_bounds__(GI_THREADS) void gene_hist_kernel(const uint16_t *__restrict__ gene_ids, const uint32_t *__restrict__ counters,
                                                               uint32_t skip_if_long, unsigned long long *__restrict__ gene_counts)
{
  if (counters[CTR_OVERFLOW] || (skip_if_long && counters[CTR_LONG])) return;
  const uint64_t n = ((uint64_t)counters[CTR_ASSOC_HI] << 32) | counters[CTR_ASSOC_LO];
  if (DIRECT) {
    const int lane_d = threadIdx.x & 63;
    const uint64_t n_round_d = (n + GI_THREADS - 1) / GI_THREADS * GI_THREADS;
    for (uint64_t i = (uint64_t)blockIdx.x * GI_THREADS + threadIdx.x; i < n_round_d; i += (uint64_t)gridDim.x * GI_THREADS) {
      const bool pending = i < n;
      const uint32_t g = pending ? gene_ids[i] : 0xFFFFFFFFu;
      // (a lane adds for the lanes above it that hold the same gene -- neighbours in a read's list never do, reads of one gene do)
      // (both shuffles by every lane, outside any condition: a lane that sits out a shuffle hands its neighbour 0 -- with the second
      //  one behind `lane_d == 0 ||`, lane 1 took lane 0 for gene 0 and a read of gene 0 there went uncounted)
      const uint32_t up = (uint32_t)__shfl_down((int)g, 1), down = (uint32_t)__shfl_up((int)g, 1);
      const bool first = lane_d == 0 || down != g;
      // length of the run of equal genes that starts here (within the wave)
      const unsigned long long differs = __ballot(up != g) | (1ull << 63);
      if (pending && first) atomicAdd(&gene_counts[g], (unsigned long long)(__builtin_ctzll(differs >> lane_d) + 1));
    }
    return;
  }
  __shared__ uint32_t h_key[GI_TABLE];
  __shared__ uint32_t h_cnt[GI_TABLE];
  for (uint32_t i = threadIdx.x; i < GI_TABLE; i += GI_THREADS) { h_key[i] = 0xFFFFFFFFu; h_cnt[i] = 0; }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  // a wave's count of the gene it met last stays in a register until another gene comes (a one-gene index: one LDS update per wave
  // and launch instead of one per 64 associations; with the 1 024 workgroups of the first version, each flushing the same counter,
  // 0.13 ms per 5 M associations)
  uint32_t run_gene = 0xFFFFFFFFu, run_cnt = 0;          // (wave-uniform)
  auto flush = [&](const uint32_t lg, const uint32_t add) {
    if (lane != 0 || !add) return;
    uint32_t slot = (lg * 2654435761u) >> 21;   // 11 bits
    bool done = false;
    for (uint32_t probe = 0; probe < 16 && !done; ++probe) {
      const uint32_t old = atomicCAS(&h_key[slot], 0xFFFFFFFFu, lg);
      if (old == 0xFFFFFFFFu || old == lg) {
        atomicAdd(&h_cnt[slot], add);
        done = true;
      }
      slot = (slot + 1) & (GI_TABLE - 1);
    }
    if (!done) atomicAdd(&gene_counts[lg], (unsigned long long)add);   // table crowded: straight to HBM
  };
  const uint64_t n_round = (n + GI_THREADS - 1) / GI_THREADS * GI_THREADS;   // keep whole waves in the loop (ballots)
  for (uint64_t i = (uint64_t)blockIdx.x * GI_THREADS + threadIdx.x; i < n_round; i += (uint64_t)gridDim.x * GI_THREADS) {
    const bool pending = i < n;
    const uint32_t g = pending ? gene_ids[i] : 0u;
    unsigned long long todo = __ballot(pending);
    while (todo) {
      const int leader = __builtin_ctzll(todo);
      const uint32_t lg = (uint32_t)__builtin_amdgcn_readlane((int)g, leader);
      const unsigned long long same = __ballot(pending && g == lg);
      const uint32_t add = (uint32_t)__builtin_popcountll(same);
      if (lg == run_gene) {
        run_cnt += add;
      } else {
        flush(run_gene, run_cnt);
        run_gene = lg;
        run_cnt = add;
      }
      todo &= ~same;
    }
  }
  flush(run_gene, run_cnt);
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < GI_TABLE; i += GI_THREADS)
    if (h_cnt[i]) atomicAdd(&gene_counts[h_key[i]], (unsigned long long)h_cnt[i]);
}

// Results leave the device through KERNEL STORES into pinned host memory, not through copy-engine commands: a
// device-to-host copy enqueued behind a batch's kernels sits in the copy engine's in-order queue until those kernels
// have finished, and the next batch's host-to-device copy queues up behind it -- measured on MI355X / ROCm 7.2: H2D and
// kernels took turns (46 GB/s of a 57 GB/s link) until the last such copy was gone from the stream.
// counters (8 words) -> host; gene_off[0..n] -> host; gene_ids[0..total) -> host (unless the batch overflowed)
__global__ __launch_bounds__(256) void publish_results_kernel(const uint32_t *__restrict__ counters, uint32_t *__restrict__ h_counters,
                                                             const uint32_t *__restrict__ gene_off, uint32_t *__restrict__ h_gene_off, uint64_t n_off,
                                                             const uint16_t *__restrict__ gene_ids, uint16_t *__restrict__ h_gene_ids, uint64_t h_ids_cap,
                                                             const uint32_t *__restrict__ uni_flag)
{
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (uint64_t)gridDim.x * blockDim.x;
  if (h_gene_off) {
    // 16 bytes per lane where aligned (both buffers come from hipMalloc / hipHostMalloc: 256-byte aligned)
    const uint64_t n4 = n_off / 4;
    const uint4 *src = reinterpret_cast<const uint4 *>(gene_off);
    uint4 *dst = reinterpret_cast<uint4 *>(h_gene_off);
    for (uint64_t i = tid; i < n4; i += nth) dst[i] = src[i];
    for (uint64_t i = n4 * 4 + tid; i < n_off; i += nth) h_gene_off[i] = gene_off[i];
    if (!counters[CTR_OVERFLOW]) {
      const uint64_t total = ((uint64_t)counters[CTR_ASSOC_HI] << 32) | counters[CTR_ASSOC_LO];
      const uint64_t m = total < h_ids_cap ? total : h_ids_cap;
      const uint64_t m8 = m / 8;
      const uint4 *s8 = reinterpret_cast<const uint4 *>(gene_ids);
      uint4 *d8 = reinterpret_cast<uint4 *>(h_gene_ids);
      for (uint64_t i = tid; i < m8; i += nth) d8[i] = s8[i];
      for (uint64_t i = m8 * 8 + tid; i < m; i += nth) h_gene_ids[i] = gene_ids[i];
    }
  }
  if (tid < CTR_WORDS) h_counters[tid] = tid == CTR_VERDICT ? (uni_flag ? 1u + uni_flag[0] : 0u) : counters[tid];
}

// off[i] = i * stride: batches whose reads all have one length need no offsets over PCIe
__global__ void fill_offsets_kernel(uint64_t *__restrict__ off, uint64_t n_plus_1, uint64_t stride)
{
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_plus_1; i += (uint64_t)gridDim.x * blockDim.x) off[i] = i * stride;
}

// ---------------------------------------------------------------------------
// host-side dispatch
// ---------------------------------------------------------------------------
uint32_t fast_kernel_max_slots() { return 64 * 10; }

uint32_t fast_kernel_unroll(uint32_t max_slots)
{
  const uint32_t u = max_slots == 0 ? 5 : (max_slots + 63) / 64;
  if (u <= 2) return 2;
  if (u <= 6) return u;
  return u <= 8 ? 8 : 10;
}

// staging groups (8 bases each) per pair that classify_uni_kernel's specialisation for `max_slots` slots can stage: one per lane
// up to 512 slots, two beyond (classify_uni.hpp, G)
uint32_t uni_kernel_max_groups(uint32_t max_slots) { return fast_kernel_unroll(max_slots) > 8 ? 128u : 64u; }

static int probe_mode(const DeviceIndex &ix)
{
  if (ix.tab_lg && ix.lsum_shift) return ix.pow2 ? PM_LDS_TAB : PM_LDS_TAB_MOD;
  if (ix.tab_lg) return ix.pow2 ? (ix.tab_with_summary ? PM_TAB_SUM : PM_TAB) : PM_TAB_MOD;
  if (!ix.pow2) return PM_BV_MOD;
  return ix.sum_shift ? PM_BV_SUM : PM_BV;
}

const char *probe_mode_name(const Ctx *ctx)
{
  static const char *names[] = {"bitvector-mod", "bitvector", "summary+bitvector", "table", "summary+table", "lds-summary+table",
                                "table-mod", "lds-summary+table-mod"};
  if (ctx->idx.ltab) return "lds-table";   // (uniform batches of up to 5 rounds; other batches take lds-summary+table)
  if (ctx->idx.ktab_lg && probe_mode(ctx->idx) == PM_TAB) return "minimiser-table";   // (classify_uni_kernel; batches of very long reads take `table`)
  return names[probe_mode(ctx->idx)];
}

template <int U>
static void launch_fast_u(const ClassifyParams &p, int mode, bool hasq, unsigned grid, hipStream_t s)
{
#define LF(M_, HQ_) hipLaunchKernelGGL((classify_fast_kernel<U, M_, HQ_>), dim3(grid), dim3(FastGeom<M_, U>::THREADS), 0, s, p)
  switch (mode) {
  case PM_BV_MOD: if (hasq) LF(PM_BV_MOD, true); else LF(PM_BV_MOD, false); break;
  case PM_BV: if (hasq) LF(PM_BV, true); else LF(PM_BV, false); break;
  case PM_BV_SUM: if (hasq) LF(PM_BV_SUM, true); else LF(PM_BV_SUM, false); break;
  case PM_TAB: if (hasq) LF(PM_TAB, true); else LF(PM_TAB, false); break;
  case PM_LDS_TAB: if (hasq) LF(PM_LDS_TAB, true); else LF(PM_LDS_TAB, false); break;
  case PM_TAB_MOD: if (hasq) LF(PM_TAB_MOD, true); else LF(PM_TAB_MOD, false); break;
  case PM_LDS_TAB_MOD: if (hasq) LF(PM_LDS_TAB_MOD, true); else LF(PM_LDS_TAB_MOD, false); break;
  default: if (hasq) LF(PM_TAB_SUM, true); else LF(PM_TAB_SUM, false); break;
  }
#undef LF
}

int launch_classify_fast(Ctx *ctx, const ClassifyParams &p, uint32_t max_slots, hipStream_t stream)
{
  if (p.n == 0) return SHK_OK;
  const bool hasq = p.hasq != 0;
  const int mode = probe_mode(ctx->idx);
  // persistent grid: enough workgroups to fill 256 CUs several times over
  // persistent workgroups; the LDS-summary mode runs 2 x 1024-thread workgroups per CU
  const uint64_t wpb = pm_lds(mode) ? 8 : CF_WAVES;
  const uint32_t u = fast_kernel_unroll(max_slots);
  const uint64_t cap = pm_lds(mode) ? (u <= 5 ? 1024 : 768) : 4096;   // (u = 10 likewise 768: six waves per SIMD at most)   // LDS mode: exactly the resident workgroups
  const uint64_t want = (p.n + wpb - 1) / wpb;
  const unsigned grid = (unsigned)(want < cap ? want : cap);
  if (u == 2) launch_fast_u<2>(p, mode, hasq, grid, stream);
  else if (u == 3) launch_fast_u<3>(p, mode, hasq, grid, stream);
  else if (u == 4) launch_fast_u<4>(p, mode, hasq, grid, stream);
  else if (u == 5) launch_fast_u<5>(p, mode, hasq, grid, stream);
  else if (u == 6) launch_fast_u<6>(p, mode, hasq, grid, stream);
  else if (u == 8) launch_fast_u<8>(p, mode, hasq, grid, stream);
  else launch_fast_u<10>(p, mode, hasq, grid, stream);
  SHK_HIP(ctx, hipGetLastError());
  snprintf(ctx->last_kernel, sizeof(ctx->last_kernel), "classify_fast_kernel<%u, %d, %s>", u, mode, hasq ? "true" : "false");
  return SHK_OK;
}

// (SHK_FORCE_GENERIC=1: every batch through classify_fast_kernel / process_read -- the tests run both code paths)
bool uni_kernel_available(const Ctx *ctx)
{
  return pm_tab(probe_mode(ctx->idx)) && !ctx->idx.wrap && !ctx->env_force_generic;
}

// (classify_uni.hpp's tri_applies, restated for the launcher's report: the lengths for which three pairs share a staging pass)
static bool tri_applies_host(uint32_t L1, uint32_t L2, uint32_t k, uint32_t S)
{
  const uint32_t c1 = (L1 + 15u) >> 4, c2 = (L2 + 15u) >> 4;
  const uint32_t nk2 = L2 >= k ? L2 - k + 1u : 0u, nk1 = L1 >= k ? L1 - k + 1u : 0u;
  return L1 != 0u && c1 + c2 <= 21u && (nk2 ? (c1 << 4) + nk2 : nk1) <= S;
}

// classify_uni_kernel lives in classify_uni.hpp, instantiated per unroll in classify_uni_u<U>.hip
void launch_uni_u2(const ClassifyParams &p, int mode, bool hasq, bool big, bool lx, int rmode, unsigned grid, hipStream_t s);
void launch_uni_u3(const ClassifyParams &p, int mode, bool hasq, bool big, bool lx, int rmode, unsigned grid, hipStream_t s);
void launch_uni_u4(const ClassifyParams &p, int mode, bool hasq, bool big, bool lx, int rmode, unsigned grid, hipStream_t s);
void launch_uni_u5(const ClassifyParams &p, int mode, bool hasq, bool big, bool lx, int rmode, unsigned grid, hipStream_t s);
void launch_uni_u6(const ClassifyParams &p, int mode, bool hasq, bool big, bool lx, int rmode, unsigned grid, hipStream_t s);
void launch_uni_u8(const ClassifyParams &p, int mode, bool hasq, bool big, bool lx, int rmode, unsigned grid, hipStream_t s);
void launch_uni_u10(const ClassifyParams &p, int mode, bool hasq, bool big, bool lx, int rmode, unsigned grid, hipStream_t s);

// the table kernel (every index with a position table), for uniform batches (`uni`) or ragged ones; with p.uni_flag set each
// of the two launches decides on the device whether it is the one that runs
// rmode: 0 = the ragged instantiation, 1 = the uniform one, 2 = class by class (a batch of mixed lengths sorted by them; exists where the
// exact table is held in LDS: class_kernel_available)
int launch_classify_uni(Ctx *ctx, const ClassifyParams &p_in, uint32_t max_slots, int rmode, hipStream_t stream)
{
  if (p_in.n == 0) return SHK_OK;
  const bool uni = rmode != 0;      // (1, 2, 3: the uniform loop -- of the batch, of a class, of the longest mates' layout)
  ClassifyParams p = p_in;
  const bool hasq = p.hasq != 0;
  int mode = probe_mode(ctx->idx);
  const uint32_t u = fast_kernel_unroll(max_slots);
  // indices too dense for the 32 KiB LDS summary may still have the 128 KiB one (index_build.hip): uniform batches then
  // run in LDS-summary mode with it, whatever chain ragged batches use on this index
  // ... unless the batch just finished says that most pairs are the panel's: behind the LDS summary every hit of such a pair is a
  // table lookup through L2 (150 genes, per 10 M pairs at 0 / 50 / 100 % on-target: 5.6 / 16.5 / 26.2 ms), while the position-table
  // kernel with the anchored extension reads the reference's payloads instead (7.9 / 11.0 / 14.8 ms).  Both give the same
  // results; the denser the summary, the earlier the other kernel wins (measured crossover: 64 / 42 / 17 % of the pairs assigned
  // at 60 / 100 / 150 genes, pass rates 0.12 / 0.20 / 0.28).  SHK_BIG_LDS_ALWAYS=1: no switching (A/B timing, tests)
  // tables far beyond the caches (k = 15 ... 17): the k-mer keyed, minimiser-bucketed table -- while the batch just finished left
  // a fifth of its pairs unassigned or more.  Pairs from a gene hardly probe either table (the anchored extension settles them from
  // the reference itself), and what they do probe are isolated slots that share no line with a neighbour: the same results at 18.8
  // against 17.9 ms per 10 M pairs on the 60 000-gene index at 100 % on-target -- and at 17.8 against 28.8 ms at 0 %, 17.8 / 19.2 at 50 %.
  // (SHK_KTAB=1, the tests' switch, keeps it on whatever the stream looks like)
  if (mode == PM_TAB && ctx->idx.ktab_lg) {
    const bool mostly_assigned = ctx->last.last_n_reads != 0 && !ctx->env_ktab_always &&
                                 (double)ctx->last.last_n_assoc > 0.8 * (double)ctx->last.last_n_reads;
    if (!mostly_assigned) mode = PM_KTAB;
  }
  // The anchored extension settles a pair from a gene from the reference itself -- behind a SAMPLE of eight probes, a dependent memory
  // round trip that a pair from elsewhere pays in front of its first rounds for nothing.  While the batch just finished left seven
  // eighths of its reads unassigned or more, the kernel is launched without it (ref_total = 0: `anchored` returns at once, every
  // read takes the usual order; results are the same either way): 60 000 genes at 0 / 2 / 10 % on-target 17.7 / 17.7 / 18.3 -> 14.5 /
  // 15.0 / 16.6 ms per 10 M pairs, 10 000 genes 15.2 / 15.4 / 15.2 -> 13.6 / 14.0 / 14.8; the two meet at a quarter of the pairs
  // on-target.  On the position table the gain is smaller and the two meet earlier (1 000 genes behind the L2 summary 9.4 / 9.5 / 9.7
  // -> 8.1 / 8.6 / 10.2; the configs[4] shape, k = 31, 24.2 -> 22.4 at 0 %, 24.3 -> 24.7 at 10 %): one twentieth there.
  // (SHK_ANCHOR_ALWAYS=1, the tests' switch, keeps it on whatever the stream looks like)
  if (!pm_lds(mode) && p.ref_total && ctx->last.last_n_reads != 0 && !ctx->env_anchor_always &&
      (double)ctx->last.last_n_assoc < (mode == PM_KTAB ? 0.125 : 0.05) * (double)ctx->last.last_n_reads)
    p.ref_total = 0;
  // anchor_verdict_kernel (below) in front of this launch?  -- uniform batches on an index that carries the reference arrays, while the
  // anchored extension is on (above) and the batch just finished had 15 reads in 100 assigned or more: the kernel costs EVERY pair of
  // the batch a pass (1.5 ms per 10 M pairs, 2.6 on the 60 000-gene index), which the pairs it settles repay several times over and
  // the others not at all -- 1 000 genes at 5 / 10 / 25 % on-target 9.6 / 9.8 / 10.3 ms without it, 10.9 / 10.6 / 9.7 with
  const bool pre_pays = ctx->last.last_n_reads == 0 || ctx->env_anchor_always || (double)ctx->last.last_n_assoc >= 0.15 * (double)ctx->last.last_n_reads;
  // (trimmed batches too, rmode 0, in the lane layout of the batch's longest mates: the ragged instantiation behind it passes over
  //  the reads that have their result by a flag per read)
  const bool pre = (rmode == 1 || rmode == 0) && !pm_lds(mode) && !ctx->env_no_pre_verdict && pre_pays && anchor_verdict_applies(p);
  // (the switch described above: with that kernel in front the pairs that made the table kernel the better one hardly reach either
  //  kernel -- the LDS summary's stays until four reads in five are assigned: 100 genes at 50 / 100 % on-target 5.7 / 5.25 ms behind
  //  the summary, 6.75 / 4.73 through the table kernel)
  bool many_assigned = false;
  if (ctx->last.last_n_reads && ctx->idx.ref_total && ctx->idx.lbig_shift && !ctx->env_big_lds_always) {
    double f = 1.0 - 3.0 * ctx->idx.lbig_pass;
    f = f < 0.1 ? 0.1 : (f > 0.9 ? 0.9 : f);
    if (pre && f < 0.8) f = 0.8;
    many_assigned = (double)ctx->last.last_n_assoc > f * (double)ctx->last.last_n_reads;
  }
  const bool big = uni && !pm_lds(mode) && ctx->idx.lbig_shift != 0 && (u <= 5 || u == 10) && !many_assigned;
  if (big) {
    mode = ctx->idx.pow2 ? PM_LDS_TAB : PM_LDS_TAB_MOD;
    p.lsum32 = ctx->idx.lbig32;
    p.lsum_shift = ctx->idx.lbig_shift;
  }
  // tiny indices: the exact table in LDS (uniform batches; power-of-two filters).  Trimmed reads stay on the LDS-summary chain
  // (measured 9.8 ms per 10 M pairs with the table in LDS, 4 waves per SIMD, against 9.7 ms at 6 waves per SIMD) -- unless the
  // index has one gene: the sparse first round (classify_uni.hpp) needs the exact table
  const bool lx = (uni || ctx->idx.ltab_gene != 0xFFFFFFFFu) && mode == PM_LDS_TAB && ctx->idx.ltab != nullptr && (u <= 5 || u == 10);
  if (lx) {
    p.lsum32 = ctx->idx.ltab;
    p.lsum_shift = ctx->idx.ltab_mul;   // (no summary in this mode: the field carries the table's slot multiplier)
    p.lx_gene = ctx->idx.ltab_gene;
    p.lx_multi = (uni && ctx->idx.ltab_sparse && ctx->idx.ltab_gene == 0xFFFFFFFFu) ? 1u : 0u;
    // three pairs per staging pass (classify_uni.hpp, TRI): uniform batches without qualities, U = 3 ... 5 (SHK_NO_TRI=1: not)
    p.tri = ((rmode == 1 || rmode == 3) && !hasq && u >= 3 && u <= 5 && !ctx->env_no_tri) ? 1u : 0u;
    // ... with a round of disjoint k-mers for the three pairs together in front (classify_uni.hpp, TF: an instantiation of its own)
    // while the batch just finished had a quarter of its reads assigned or more: a pair from the gene then ends behind a THIRD of a
    // hash round, a pair from elsewhere pays that third on top of its own two rounds (217 VALU instructions against 191).  Per 10 M
    // pairs at 0 / 50 / 100 % on-target: 3.23 / 3.15 / 3.11 ms without, (3.7) / 2.69 / 1.69 with -- the two meet at about a quarter
    p.tile_first = (p.tri && p.lx_gene != 0xFFFFFFFFu && ctx->env_tile_first >= 0 &&
                    (ctx->env_tile_first > 0 || (ctx->last.last_n_reads != 0 && (double)ctx->last.last_n_assoc >= 0.25 * (double)ctx->last.last_n_reads))) ? 1u : 0u;
  }
  // The pairs a base-for-base comparison with the reference settles, settled in front of the table kernel (anchor_verdict.hip), also in
  // front of the 128 KiB LDS summary's kernel.  The kernel behind it passes over the reads that have their result
  // (SHK_NO_PRE_VERDICT=1: never)
  if (pre) {
    if (int rc = launch_anchor_verdict(p, ctx->idx.pow2, rmode == 0, stream)) return rc;
    p.pre_verdict = 1u;
  }
  const bool wg16 = big || lx;   // one 1024-thread workgroup per CU
  const int min_waves = wg16 ? 4 : ((u > 8 || (u > 5 && !pm_lds(mode))) ? 4 : (u > 5 ? 6 : (pm_lds(mode) ? SHK_UNI_WAVES : (mode == PM_KTAB ? SHK_KT_WAVES : SHK_TAB_WAVES))));   // (= UniGeom::MIN_WAVES)
  const uint64_t wpb = lx ? SHK_LX_WAVES : (wg16 ? 16 : 8);
  const uint64_t cap = wg16 ? 256ull : 256ull * (uint64_t)(min_waves / 2);   // exactly the resident workgroups
  const uint64_t want = (p.n + wpb - 1) / wpb;
  const unsigned grid = (unsigned)(want < cap ? want : cap);
  if ((rmode == 2 || rmode == 3) && !lx) return SHK_OK;      // (no such instantiation on this index: the verdict is never "by classes" / "offsets" there)
  if (u == 2) launch_uni_u2(p, mode, hasq, big, lx, rmode, grid, stream);
  else if (u == 3) launch_uni_u3(p, mode, hasq, big, lx, rmode, grid, stream);
  else if (u == 4) launch_uni_u4(p, mode, hasq, big, lx, rmode, grid, stream);
  else if (u == 5) launch_uni_u5(p, mode, hasq, big, lx, rmode, grid, stream);
  else if (u == 6) launch_uni_u6(p, mode, hasq, false, false, rmode, grid, stream);
  else if (u == 8) launch_uni_u8(p, mode, hasq, false, false, rmode, grid, stream);
  else launch_uni_u10(p, mode, hasq, big, lx, rmode, grid, stream);
  SHK_HIP(ctx, hipGetLastError());
  if (rmode == 2) return SHK_OK;             // (shk_last_kernel names the uniform / ragged launch)
  if (rmode == 3) {
    if (!p.uni_flag)      // (the host knows the batch is of mixed lengths: this launch is the one that works)
      snprintf(ctx->last_kernel, sizeof(ctx->last_kernel), "classify_uni_kernel<%u, %d, false, 21, offsets> +three-pairs%s%s", u, mode,
               p.tile_first ? " +tiles-first" : "", p.lx_multi ? " +sparse-first-rounds" : (p.lx_gene != 0xFFFFFFFFu ? " +sparse-first-round" : ""));
    return SHK_OK;
  }
  // (with p.uni_flag both instantiations are launched and one returns at once: the name says UNI = "device")
  if (uni || !p.uni_flag)
    snprintf(ctx->last_kernel, sizeof(ctx->last_kernel), "classify_uni_kernel<%u, %d, %s, %d, %s>%s%s%s", u, mode, hasq ? "true" : "false",
             (lx && (u <= 5 || u == 10)) ? 21 : ((big && (u <= 5 || u == 10)) ? 20 : 18), p.uni_flag ? "device" : (uni ? "true" : "false"),
             (!pm_lds(mode) && p.ref_total) ? " +anchored-extension" : "", pre ? " +pre-verdict" : "",
             (lx && p.lx_gene != 0xFFFFFFFFu) ? " +sparse-first-round" : ((lx && p.lx_multi) ? " +sparse-first-rounds" : ""));
  if (lx && p.tri && (uni || !p.uni_flag) && (p.uni_flag || tri_applies_host(p.uni_L1, p.uni_L2, p.k, 64u * u))) {
    const size_t l = strlen(ctx->last_kernel);
    snprintf(ctx->last_kernel + l, sizeof(ctx->last_kernel) - l, "%s%s", p.uni_flag ? " +three-pairs-if-they-fit" : " +three-pairs", p.tile_first ? " +tiles-first" : "");
  }
  return SHK_OK;
}

// the TRO instantiations (mixed lengths through the three-pairs kernel): exact table in LDS, no qualities, U = 3 ... 5, and lengths
// for which three pairs share a staging pass (SHK_NO_TRO=1 / SHK_NO_TRI=1: never)
bool offsets_kernel_available(const Ctx *ctx, uint32_t max_slots, bool hasq)
{
  const uint32_t u = fast_kernel_unroll(max_slots);
  return uni_kernel_available(ctx) && probe_mode(ctx->idx) == PM_LDS_TAB && ctx->idx.ltab != nullptr && !hasq && u >= 3 && u <= 5 && !ctx->env_no_tri && !ctx->env_no_tro;
}
bool offsets_kernel_fits(const Ctx *ctx, uint32_t max_slots, uint32_t L1, uint32_t L2)
{
  return tri_applies_host(L1, L2, ctx->prm.k, 64u * fast_kernel_unroll(max_slots));
}

// the CLS instantiation exists where uniform batches take the exact table in LDS
bool class_kernel_available(const Ctx *ctx, uint32_t max_slots)
{
  const uint32_t u = fast_kernel_unroll(max_slots);
  return uni_kernel_available(ctx) && probe_mode(ctx->idx) == PM_LDS_TAB && ctx->idx.ltab != nullptr && (u <= 5 || u == 10);
}

int launch_class_prepass(const ClassifyParams &p, uint32_t slot_cap, uint32_t *flag, hipStream_t stream)
{
  // one workgroup per CU at most (144 KiB of LDS each), 4 096 pairs or more per workgroup
  const uint64_t want = (p.n + 4095) / 4096;
  const unsigned grid = (unsigned)(want < 1 ? 1 : (want < 256 ? want : 256));
  hipLaunchKernelGGL(class_hist_kernel, dim3(grid), dim3(1024), 0, stream, p, slot_cap, flag, p.cls_hist);
  if (hipGetLastError() != hipSuccess) return SHK_ERR_HIP;
  hipLaunchKernelGGL(class_plan_kernel, dim3(1), dim3(1024), 0, stream, p, slot_cap, flag, p.cls_hist, p.cls_hist + p.cls_cap, p.cls_list, p.cls_share_first);
  if (hipGetLastError() != hipSuccess) return SHK_ERR_HIP;
  hipLaunchKernelGGL(class_scatter_kernel, dim3(grid), dim3(1024), 0, stream, p, flag, p.cls_hist + p.cls_cap, p.cls_entries);
  return hipGetLastError() == hipSuccess ? SHK_OK : SHK_ERR_HIP;
}

int launch_uniform_check(const ClassifyParams &p, uint32_t slot_cap, uint32_t *flag, hipStream_t stream)
{
  // (flag[] is zero: cleared with the batch's counters, shark_hip.hip enqueue_classify)
  // (every workgroup ends with an atomic on the same word -- and three more when the batch is ragged --, 40-50 ns each one after the
  //  other: 1 024 workgroups spent half of the kernel's 0.08 ms there.  So 256 workgroups -- of 1 024 threads for large batches:
  //  160 MB of offsets per 10 M pairs want more loads in flight than 65 536 threads have)
  const unsigned threads = p.n >= (1ull << 20) ? 1024u : 256u;
  const uint64_t want = (p.n + threads - 1) / threads;
  hipLaunchKernelGGL(uniform_check_kernel, dim3((unsigned)(want < 1 ? 1 : (want < 256 ? want : 256))), dim3(threads), 0, stream, p, slot_cap, flag);
  return hipGetLastError() == hipSuccess ? SHK_OK : SHK_ERR_HIP;
}

int launch_classify_general(Ctx *ctx, const ClassifyParams &p, bool emit, unsigned n_waves, hipStream_t stream)
{
  if (p.n_work == 0 && !p.work_count) return SHK_OK;
  const bool pow2 = ctx->idx.pow2, hasq = p.hasq != 0, wrap = ctx->idx.wrap;
  const unsigned grid = (n_waves + CF_WAVES - 1) / CF_WAVES;
#define LG3(P2_, HQ_, EM_) do { if (wrap) hipLaunchKernelGGL((classify_general_kernel<P2_, HQ_, EM_, true>), dim3(grid), dim3(CF_THREADS), 0, stream, p); \
                                else hipLaunchKernelGGL((classify_general_kernel<P2_, HQ_, EM_, false>), dim3(grid), dim3(CF_THREADS), 0, stream, p); } while (0)
#define LG(P2_, HQ_, EM_) LG3(P2_, HQ_, EM_)
  if (emit) {
    if (pow2) { if (hasq) LG(true, true, true); else LG(true, false, true); }
    else { if (hasq) LG(false, true, true); else LG(false, false, true); }
  } else {
    if (pow2) { if (hasq) LG(true, true, false); else LG(true, false, false); }
    else { if (hasq) LG(false, true, false); else LG(false, false, false); }
  }
#undef LG3
#undef LG
  SHK_HIP(ctx, hipGetLastError());
  return SHK_OK;
}

int launch_gather_inline(const uint32_t *count, const uint16_t *inl, const uint32_t *gene_off, uint16_t *gene_ids, uint64_t n,
                         const uint32_t *counters, hipStream_t stream)
{
  if (n == 0) return SHK_OK;
  const uint64_t want = (n + GI_THREADS - 1) / GI_THREADS;
  hipLaunchKernelGGL(gather_inline_kernel, dim3((unsigned)(want < 4096 ? want : 4096)), dim3(GI_THREADS), 0, stream, count, inl, gene_off, gene_ids, n, counters);
  return hipGetLastError() == hipSuccess ? SHK_OK : SHK_ERR_HIP;
}

int launch_finalize_total(const uint64_t *total, uint32_t *counters, uint64_t gene_ids_cap, hipStream_t stream)
{
  hipLaunchKernelGGL(finalize_total_kernel, dim3(1), dim3(1), 0, stream, total, counters, gene_ids_cap);
  return hipGetLastError() == hipSuccess ? SHK_OK : SHK_ERR_HIP;
}

int launch_gene_hist(const uint16_t *gene_ids, const uint32_t *counters, bool skip_if_long, unsigned long long *gene_counts, uint64_t n_reads, uint64_t n_genes,
                     hipStream_t stream)
{
  // the number of associations is only known on the device; it is of the order of the number of reads
  const uint64_t want = (n_reads + GI_THREADS - 1) / GI_THREADS;
  if (n_genes > GI_TABLE) {
    const unsigned grid = (unsigned)(want < 1 ? 1 : (want < 2048 ? want : 2048));
    hipLaunchKernelGGL(gene_hist_kernel<true>, dim3(grid), dim3(GI_THREADS), 0, stream, gene_ids, counters, skip_if_long ? 1u : 0u, gene_counts);
    return hipGetLastError() == hipSuccess ? SHK_OK : SHK_ERR_HIP;
  }
  const unsigned grid = (unsigned)(want < 1 ? 1 : (want < 512 ? want : 512));
  hipLaunchKernelGGL(gene_hist_kernel<false>, dim3(grid), dim3(GI_THREADS), 0, stream, gene_ids, counters, skip_if_long ? 1u : 0u, gene_counts);
  return hipGetLastError() == hipSuccess ? SHK_OK : SHK_ERR_HIP;
}

int launch_publish_results(const uint32_t *counters, uint32_t *h_counters, const uint32_t *gene_off, uint32_t *h_gene_off, uint64_t n_off,
                           const uint16_t *gene_ids, uint16_t *h_gene_ids, uint64_t h_ids_cap, const uint32_t *uni_flag, hipStream_t stream)
{
  const uint64_t want = h_gene_off ? (n_off / 4 + 255) / 256 : 1;
  hipLaunchKernelGGL(publish_results_kernel, dim3((unsigned)(want < 1 ? 1 : (want < 512 ? want : 512))), dim3(256), 0, stream, counters, h_counters,
                     gene_off, h_gene_off, n_off, gene_ids, h_gene_ids, h_ids_cap, uni_flag);
  return hipGetLastError() == hipSuccess ? SHK_OK : SHK_ERR_HIP;
}

// shk_classify_device_submit takes the caller's word for "every read of mate m has length L_m": the kernels then never read an
// offset.  One thread looks at three offsets per mate (first, middle, last: r L_m each); what it finds goes to the host with the
// batch's counters and shk_classify_wait refuses the batch (SHK_ERR_ARG) instead of handing out results of reads cut at the wrong places.
__global__ void vouch_check_kernel(const uint64_t *__restrict__ off1, const uint64_t *__restrict__ off2, uint64_t n, uint32_t L1, uint32_t L2,
                                   uint32_t *__restrict__ counters)
{
  const uint64_t h = n / 2;
  bool bad = off1[0] != 0ull || off1[h] != h * L1 || off1[n] != n * L1;
  if (off2) bad = bad || off2[0] != 0ull || off2[h] != h * L2 || off2[n] != n * L2;
  if (bad) counters[CTR_VOUCH_BAD] = 1u;
}

int launch_vouch_check(const ClassifyParams &p, uint32_t L1, uint32_t L2, uint32_t *counters, hipStream_t stream)
{
  hipLaunchKernelGGL(vouch_check_kernel, dim3(1), dim3(1), 0, stream, p.off1, p.seq2 ? p.off2 : nullptr, p.n, L1, L2, counters);
  return hipGetLastError() == hipSuccess ? SHK_OK : SHK_ERR_HIP;
}

int launch_fill_offsets(uint64_t *off, uint64_t n_plus_1, uint64_t stride, hipStream_t stream)
{
  const uint64_t want = (n_plus_1 + 255) / 256;
  hipLaunchKernelGGL(fill_offsets_kernel, dim3((unsigned)(want < 2048 ? want : 2048)), dim3(256), 0, stream, off, n_plus_1, stride);
  return hipGetLastError() == hipSuccess ? SHK_OK : SHK_ERR_HIP;
}

}  // namespace shk
