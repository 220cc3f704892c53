// anchor_verdict.hip -- anchor_verdict_kernel: the pairs a reference base-for-base comparison settles, settled in FRONT of
// classify_uni_kernel's table instantiations, three pairs per wavefront pass (round 6).
//
// What it replaces.  ReadAnalyzer::operator() (ReadAnalyzer.hpp:39-110) looks every k-mer of a read up (bloomfilter.h:78-102) and
// votes.  For a pair drawn from ONE gene's own sequence nearly all of that is foregone: its k-mers are the reference's k-mers at
// neighbouring positions, and the index knows, per reference position, how far around it every k-mer answers with one and the same
// single-gene list {g} (DeviceIndex::refext).  classify_uni_kernel's anchored extension uses the reference that way per slot (sample
// -> anchor -> per-slot payloads -> match-bit windows -> vote: ~570 VALU + 350 scalar instructions per pair, three dependent memory
// round trips per pair and wave).  This kernel keeps only what the verdict needs:
//   * a lane holds 16 bases of one mate (a chunk); 10 lanes a 150-bp mate, 20 a pair, THREE pairs per wave pass;
//   * the k-mer that starts at a chunk's first base is one more look at the neighbouring lane's bases -- so every chunk lane samples
//     one k-mer (hash, bucket of `atab`: "is it in the index, and where in the reference") in the SAME instructions;
//   * the first sampled k-mer of a mate that is in the index anchors the mate: every chunk lane of the mate compares its 16 bases
//     with the 16 reference bases they stand against (one xor) and counts the bases that disagree or are invalid characters;
//   * `refext` at the anchor says whether every slot of the mate falls on positions that answer {g}, same g for both mates.
// THE VERDICT (the early decision's argument, classify_uni.hpp `vote`, made on counts): a base that disagrees lies in at most k
// slots, so with e such bases at least n = nk1 + nk2 - e k slots hold k-mers EQUAL to the reference's -- equal filter positions,
// hence hits of g and of g alone -- covering at least n + k - 1 bases (the union of [p, p + k) holds every p and k - 1 more bases);
// every other slot -- the only ones another gene's list can sit under -- lies within k - 1 bases of a disagreeing base: together
// they cover at most e (2 k - 1) bases.  n + k - 1 >= ceil(c len) and n + k - 1 > e (2 k - 1) therefore make g the pair's one
// association (ReadAnalyzer.hpp:90-108) whatever the other probes would say: count = 1, gene g, exactly what the vote writes.
// (2 x 150 bp, k = 17, c = 0.6: up to five disagreeing bases.)  A pair that does not pass is simply left alone -- count[] stays 0 --
// and classify_uni_kernel, launched behind this kernel with `pre_verdict` set, skips the pairs that have their result.
// Nothing is assumed: a chance anchor, a repeat, an indel, another gene's mate all show as disagreeing bases or uncovered extents.
#include "classify_common.hpp"

namespace shk {

constexpr int AV_WAVES = 4;            // waves per workgroup
constexpr uint32_t AV_PASSES = 8;      // consecutive passes (groups of up to three pairs) a wave takes

struct AvRaw { uint32_t d0, d1, d2, d3, d4, sh; };

// uniform batches (one length per mate), pairs of at most 64 chunks of 16 bases.  POW2: the filter size is a power of two
template <bool POW2, bool HASQ>
__global__ __launch_bounds__(AV_WAVES * 64) void anchor_verdict_kernel(const ClassifyParams P)
{
  uint32_t L1 = P.uni_L1, L2 = P.uni_L2;
  if (P.uni_flag) {
    if (P.uni_flag[0] != 1u) return;   // (the device's verdict: not a uniform batch)
    L1 = P.uni_flag[1];
    L2 = P.uni_flag[2];
  }
  L1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)L1);
  L2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)L2);
  const uint32_t k = P.k;
  const uint32_t c1 = (L1 + 15u) >> 4, c2 = (L2 + 15u) >> 4, lp = c1 + c2;   // chunks per mate, lanes per pair
  if (lp == 0u || lp > 64u) return;
  const uint32_t ppw = 64u / lp < 3u ? 64u / lp : 3u;                        // pairs per pass
  const uint32_t nk1 = L1 >= k ? L1 - k + 1u : 0u, nk2 = L2 >= k ? L2 - k + 1u : 0u, nks = nk1 + nk2;
  const uint32_t thr = cov_threshold(P.c, L1 + L2);   // (a pair with invalid characters has a lower threshold: passing this one is sufficient)
  if (nks == 0u || thr == 0u) return;
  const uint32_t n_reads = (uint32_t)P.n;
  const uint32_t n_pass = (n_reads + ppw - 1u) / ppw;
  const int lane = threadIdx.x & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * AV_WAVES + (threadIdx.x >> 6)));
  uint32_t g = wave * AV_PASSES;
  if (g >= n_pass) return;
  const uint32_t g_end = g + AV_PASSES < n_pass ? g + AV_PASSES : n_pass;

  // ---- lane geometry: pair of the pass, chunk of the pair, mate ----
  const uint32_t ln = (uint32_t)lane;
  const bool act = ln < ppw * lp;
  const uint32_t pr = act ? (ln >= lp ? 1u : 0u) + (ln >= 2u * lp ? 1u : 0u) : 0u;
  const uint32_t cl = act ? ln - pr * lp : 0u;
  const bool in2 = cl >= c1;
  const uint32_t cm = in2 ? cl - c1 : cl;                 // chunk of the mate
  const uint32_t bofs = cm << 4;                          // its first base (mate coordinates)
  const uint32_t Lm = in2 ? L2 : L1, nkm = in2 ? nk2 : nk1;
  const uint32_t nb = act ? (Lm - bofs < 16u ? Lm - bofs : 16u) : 0u;       // bases of the mate in the chunk (>= 1 for an active lane)
  const uint32_t tail = nb < 16u ? (0xFFFFu << nb) & 0xFFFFu : 0u;
  const bool has_kmer = act && bofs + k <= Lm;            // a k-mer starts at the chunk's first base
  // (a probe is a memory-side request on every index whose table has outgrown an XCD's L2, and their rate is what this kernel waits
  //  for: every third chunk is sampled -- slots 0, 48, 96 of a 150-bp mate; all three miss with 1 % errors once in 150 mates)
  const bool sampled = has_kmer && cm % 3u == 0u;
  const uint32_t fm = pr * lp + (in2 ? c1 : 0u);          // first lane of my mate
  const uint32_t cmn = in2 ? c2 : c1;                     // its lanes
  const uint32_t f1 = pr * lp + (nk1 ? 0u : c1);          // first lane of the pair's first mate that has slots
  const uint64_t pair_mask = (lp < 64u ? (1ull << lp) - 1ull : ~0ull) << (pr * lp);
  const uint8_t *sb = (in2 ? P.seq2 : P.seq1) + bofs;
  const uint8_t *qb = HASQ ? (in2 ? P.qual2 : P.qual1) + bofs : nullptr;
  const uint64_t kmer_mask = (1ull << (2u * k)) - 1ull;   // (k <= 31)
  const uint32_t Lmin = L2 ? (L1 < L2 ? L1 : L2) : L1;
  const uint32_t guard = Lmin ? (19u + Lmin - 1u) / Lmin : 0u;   // an unguarded fetch reads up to 19 bytes from the chunk's first: the last reads take the guarded form
  const uint32_t pmask = (uint32_t)((1ull << P.tab_lg) - 1ull) & (uint32_t)P.bf_mask;
  const uint32_t pspare = 1u << P.tab_lg;
  const uint32_t tagmask = (uint32_t)(P.bf_mask >> P.tab_lg);
  const uint4 *atab16 = reinterpret_cast<const uint4 *>(P.atab);
  const uint32_t ref_total = P.ref_total;

  auto issue = [&](const uint8_t *base, const uint32_t gg) -> AvRaw {
    AvRaw r{0u, 0u, 0u, 0u, 0u, 0u};
    const uint32_t rd = ppw * gg + pr;
    if (act && rd < n_reads) {
      const uint8_t *sp = base + (uint64_t)rd * Lm;
      const uint32_t sh = (uint32_t)reinterpret_cast<uintptr_t>(sp) & 3u;
      const uint32_t *q = reinterpret_cast<const uint32_t *>(sp - sh);
      r.sh = sh;
      if (n_reads - rd > guard) {
        r.d0 = q[0]; r.d1 = q[1]; r.d2 = q[2]; r.d3 = q[3]; r.d4 = q[4];
      } else {
        const uint32_t last = sh + nb - 1u;              // index of the last wanted byte relative to q
        r.d0 = q[0];
        r.d1 = last >= 4u ? q[1] : 0u;
        r.d2 = last >= 8u ? q[2] : 0u;
        r.d3 = last >= 12u ? q[3] : 0u;
        r.d4 = last >= 16u ? q[4] : 0u;
      }
    }
    return r;
  };
  AvRaw cur = issue(sb, g), qcur{0u, 0u, 0u, 0u, 0u, 0u};
  if (HASQ) qcur = issue(qb, g);
  for (; g < g_end; ++g) {
    AvRaw nxt{0u, 0u, 0u, 0u, 0u, 0u}, qnxt{0u, 0u, 0u, 0u, 0u, 0u};
    if (g + 1u < g_end) {
      nxt = issue(sb, g + 1u);
      if (HASQ) qnxt = issue(qb, g + 1u);
    }
    const uint32_t rd = ppw * g + pr;
    const bool live = act && rd < n_reads;
    // ---- the chunk's 16 bases as 2-bit codes, first base LOW (classify_uni.hpp's `fw` stream), and which of them are invalid ----
    uint32_t code, inv16;
    {
      const uint32_t b0 = __builtin_amdgcn_alignbyte(cur.d1, cur.d0, cur.sh), b1 = __builtin_amdgcn_alignbyte(cur.d2, cur.d1, cur.sh);
      const uint32_t b2 = __builtin_amdgcn_alignbyte(cur.d3, cur.d2, cur.sh), b3 = __builtin_amdgcn_alignbyte(cur.d4, cur.d3, cur.sh);
      uint32_t e0, e1, e2, e3, i0, i1, i2, i3;
      classify4(b0, e0, i0);
      classify4(b1, e1, i1);
      classify4(b2, e2, i2);
      classify4(b3, e3, i3);
      const uint32_t msb32 = (pack4(e0) << 24) | (pack4(e1) << 16) | (pack4(e2) << 8) | pack4(e3);      // first base in bits 31:30
      inv16 = gather4(i0) | (gather4(i1) << 4) | (gather4(i2) << 8) | (gather4(i3) << 12) | tail;
      if (HASQ) {
        const uint32_t q0 = __builtin_amdgcn_alignbyte(qcur.d1, qcur.d0, qcur.sh), q1 = __builtin_amdgcn_alignbyte(qcur.d2, qcur.d1, qcur.sh);
        const uint32_t q2 = __builtin_amdgcn_alignbyte(qcur.d3, qcur.d2, qcur.sh), q3 = __builtin_amdgcn_alignbyte(qcur.d4, qcur.d3, qcur.sh);
        inv16 |= gather4(qmask4(q0, P.mq)) | (gather4(qmask4(q1, P.mq)) << 4) | (gather4(qmask4(q2, P.mq)) << 8) | (gather4(qmask4(q3, P.mq)) << 12);   // FastqSplitter.hpp:104-109
      }
      code = __builtin_bitreverse32(msb32);
      code = ((code >> 1) & 0x55555555u) | ((code & 0x55555555u) << 1);
      if (!live) inv16 = 0xFFFFu;
    }
    // ---- the k-mer that starts at the chunk's first base: this chunk's bases and the next chunk's first k - 16 ----
    const uint32_t nx_code = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((ln + 1u) & 63u) << 2), (int)code);
    const uint32_t nx_inv = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((ln + 1u) & 63u) << 2), (int)inv16);
    const uint64_t x = (((uint64_t)nx_code << 32) | code) & kmer_mask;        // first base low: ~x is the reverse complement (kmer_utils.hpp:47-55)
    const uint32_t own_need = k >= 16u ? 0xFFFFu : (1u << k) - 1u, nx_need = k > 16u ? (1u << (k - 16u)) - 1u : 0u;
    const bool kvalid = has_kmer && ((inv16 & own_need) | (nx_inv & nx_need)) == 0u;
    uint64_t fwd;                                                              // the k-mer as kmer_utils.hpp:67-69 packs it: first base most significant
    {
      uint32_t lo = __builtin_bitreverse32((uint32_t)(x >> 32)), hi = __builtin_bitreverse32((uint32_t)x);
      lo = ((lo >> 1) & 0x55555555u) | ((lo & 0x55555555u) << 1);
      hi = ((hi >> 1) & 0x55555555u) | ((hi & 0x55555555u) << 1);
      fwd = (((uint64_t)hi << 32) | lo) >> (64u - 2u * k);
    }
    const uint64_t rc = ~x & kmer_mask;
    const bool isrc = !(fwd < rc);                                             // canonical = min (KmerBuilder.hpp:49, ReadAnalyzer.hpp:55)
    const uint64_t hsh = xxh64_u64(isrc ? rc : fwd);
    const uint64_t pos = POW2 ? (hsh & P.bf_mask) : bf_pos_np(hsh, P);         // bloomfilter.h:87-88
    const bool probe = kvalid && sampled;
    const uint32_t bi = probe ? ((uint32_t)pos & pmask) : pspare;              // (the spare bucket stays empty: a probe that needs no answer)
    uint4 bk;
    if (P.tab_nt) {
      const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(atab16) + bi);
      bk = make_uint4(v.x, v.y, v.z, v.w);
    } else {
      bk = atab16[bi];
    }
    const uint32_t want = ((__builtin_amdgcn_alignbit((uint32_t)(pos >> 32), (uint32_t)pos, P.tab_lg) & tagmask) << 8) | 0x80u;
    const bool m0 = bk.y == want, m1 = bk.w == want;                           // (home bucket only: a displaced key gives no anchor)
    uint32_t a = m0 ? bk.x : bk.z;                                             // where in the reference: x | strand << 31
    const bool hit = probe && (m0 | m1) && a != 0xFFFFFFFFu;
    a ^= isrc ? 0x80000000u : 0u;                                              // bit 31 now: the read shows the other strand
    // ---- the mate's anchor: its first sampled k-mer that is in the index ----
    const uint64_t HB = __ballot(hit);
    const uint64_t mine = (HB >> fm) & ((1ull << cmn) - 1ull);
    const bool have = act && nkm != 0u && mine != 0ull;
    const uint32_t src = fm + (have ? (uint32_t)__builtin_ctzll(mine) : 0u);
    const uint32_t a_src = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)a);
    const uint32_t x0 = a_src & 0x7FFFFFFFu, s0 = (src - fm) << 4;             // reference position and slot (mate coordinates) of the anchor
    const bool opp = (a_src >> 31) != 0u;
    // ---- what the reference says there: the anchor's surroundings, and the 16 bases this chunk stands against ----
    // (the base at mate position b stands against reference base x0 - s0 + b, or, on the other strand, against the complement of base
    //  x0 + s0 + k - 1 - b: classify_uni.hpp, anchored extension (3))
    const uint32_t ev = have ? P.refext[x0] : REFEXT_NONE;
    const uint32_t lo = opp ? x0 + s0 + k - 16u - bofs : x0 + bofs - s0;       // (mod 2^32: a chunk that would leave the reference fails the bound)
    const bool inr = have && lo < ref_total;
    const uint32_t ls = inr ? lo : 0u;
    const uint32_t g0 = P.ref2[ls >> 4], g1 = P.ref2[(ls >> 4) + 1u];
    uint32_t G = __builtin_amdgcn_alignbit(g1, g0, (ls & 15u) << 1);
    if (opp) {   // the other strand: base order reversed (2-bit groups), complemented
      G = __builtin_bitreverse32(G);
      G = ~(((G >> 1) & 0x55555555u) | ((G & 0x55555555u) << 1));
    }
    const uint32_t df = code ^ G;
    uint32_t e = ~(df | (df >> 1)) & 0x55555555u;                              // bit 2 i: base i agrees
    e = (e | (e >> 1)) & 0x33333333u;
    e = (e | (e >> 2)) & 0x0F0F0F0Fu;
    e = (e | (e >> 4)) & 0x00FF00FFu;
    e = (e | (e >> 8)) & 0xFFFFu;
    const uint32_t ok16 = inr ? (e & ~inv16) : 0u;                             // (inv16 holds the bases behind the mate's end as well)
    const uint32_t mis = (live && nkm != 0u) ? nb - (uint32_t)__builtin_popcount(ok16) : 0u;
    // every slot of the mate on a position that answers with the anchor's single-gene list?  (slots [0, nkm) <-> positions
    // x0 - s0 ... x0 - s0 + nkm - 1, or mirrored)
    const uint32_t before = s0, after = nkm - 1u - s0, left = (ev >> 16) & 0xFFu, right = ev >> 24;
    bool okm = nkm == 0u || (have && ev != REFEXT_NONE && (opp ? (right >= before) & (left >= after) : (left >= before) & (right >= after)));
    const uint32_t gene = ev & 0xFFFFu;
    const uint32_t gene1 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(f1 << 2), (int)gene);
    okm = okm && (nkm == 0u || gene == gene1);
    const uint64_t BAD = __ballot(live && !okm);
    const uint32_t tot = wave_sum_u32(mis << (10u * pr));                      // (three fields of ten bits: at most 64 x 16 per pair)
    const uint32_t e_mis = (tot >> (10u * pr)) & 1023u, killed = e_mis * k;
    const uint32_t cov_lb = nks - killed + k - 1u;
    const bool pass = (BAD & pair_mask) == 0ull && killed < nks && cov_lb >= thr && cov_lb > e_mis * (2u * k - 1u);
    if (live && cl == 0u && pass) {
      const ClassifyOut *O = P.out;
      O->count[rd] = 1u;
      uint2 pk;
      pk.x = gene1;
      pk.y = 0u;
      *reinterpret_cast<uint2 *>(O->inl + (uint64_t)rd * SHK_INLINE_IDS) = pk;
    }
    cur = nxt;
    if (HASQ) qcur = qnxt;
  }
}

// does the kernel apply to uniform batches of these lengths?  (lengths only the device knows: it decides itself)
bool anchor_verdict_applies(const ClassifyParams &p)
{
  if (!p.ref_total || !p.refext || !p.atab || !p.ref2 || p.n == 0 || p.n >= (1ull << 32)) return false;
  if (p.uni_flag) return true;
  const uint32_t lp = ((p.uni_L1 + 15u) >> 4) + ((p.seq2 ? p.uni_L2 : 0u) + 15u) / 16u;
  return lp != 0u && lp <= 64u;
}

int launch_anchor_verdict(const ClassifyParams &p, bool pow2, hipStream_t s)
{
  // (lengths only the device knows: a pass may hold one pair only -- the grid is sized for that, waves without work return at once)
  const uint32_t lp = p.uni_flag ? 0u : ((p.uni_L1 + 15u) >> 4) + ((p.seq2 ? p.uni_L2 : 0u) + 15u) / 16u;
  const uint64_t ppw = p.uni_flag ? 1u : (64u / lp < 3u ? 64u / lp : 3u);
  const uint64_t n_pass = p.uni_flag ? p.n : (p.n + ppw - 1) / ppw;
  const uint64_t waves = (n_pass + AV_PASSES - 1) / AV_PASSES;
  const unsigned grid = (unsigned)((waves + AV_WAVES - 1) / AV_WAVES);
  const bool hasq = p.hasq != 0;
  if (pow2) {
    if (hasq) hipLaunchKernelGGL((anchor_verdict_kernel<true, true>), dim3(grid), dim3(AV_WAVES * 64), 0, s, p);
    else hipLaunchKernelGGL((anchor_verdict_kernel<true, false>), dim3(grid), dim3(AV_WAVES * 64), 0, s, p);
  } else {
    if (hasq) hipLaunchKernelGGL((anchor_verdict_kernel<false, true>), dim3(grid), dim3(AV_WAVES * 64), 0, s, p);
    else hipLaunchKernelGGL((anchor_verdict_kernel<false, false>), dim3(grid), dim3(AV_WAVES * 64), 0, s, p);
  }
  return hipGetLastError() == hipSuccess ? SHK_OK : SHK_ERR_HIP;
}

}  // namespace shk
