// anchor_verdict.hip -- anchor_verdict_kernel: the pairs a reference base-for-base comparison settles, settled in FRONT of
// classify_uni_kernel's table instantiations, six pairs per wavefront pass (round 6).
//
// What it replaces.  ReadAnalyzer::operator() (ReadAnalyzer.hpp:39-110) looks every k-mer of a read up (bloomfilter.h:78-102) and
// votes.  For a pair drawn from ONE gene's own sequence nearly all of that is foregone: its k-mers are the reference's k-mers at
// neighbouring positions, and the index knows, per reference position, the gene of the record it lies in and whether the list under
// its k-mer is that gene alone (DeviceIndex::refext, refmul).  classify_uni_kernel's anchored extension uses the reference that way
// per slot (sample -> anchor -> per-slot payloads -> match-bit windows -> vote: ~570 VALU + 350 scalar instructions per pair, three
// dependent memory round trips per pair and wave).  This kernel keeps only what the verdict needs:
//   * a lane holds 32 bases of one mate (a chunk) as 2-bit codes in a register pair; 5 lanes a 150-bp mate, 10 a pair, SIX pairs per
//     wave pass -- everything below is done once per pass, for six pairs at a time;
//   * the k-mer that starts at a chunk's first base lies inside the chunk (k <= 31) -- so a chunk lane samples one k-mer (hash,
//     bucket of `atab`: "is it in the index, and where in the reference") without looking anywhere else;
//   * the first sampled k-mer of a mate that is in the index anchors the mate: every chunk lane of the mate compares its 32 bases
//     with the 32 reference bases they stand against (one 64-bit xor): one bit per base, "agrees and is a valid character";
//   * `refext` at the anchor says whether every slot of the mate falls on positions of g's record that start a valid k-mer (same
//     g for both mates), `refmul` which of those positions carry a list of several genes.
// THE VERDICT (the early decision's argument, classify_uni.hpp `vote`, made on bit masks).  A slot whose k bases all agree with the
// reference holds the reference's k-mer at that position: equal k-mers, equal filter positions (bloomfilter.h:87-88), hence exactly
// that position's list.  Where that list is {g} alone the slot is a hit of g and of no other gene -- a MATCHED slot.  Every other
// existing slot -- a disagreeing or invalid base in its window, a list of several genes under it -- is OPEN: nothing is assumed
// about it.  g's coverage is then at least the bases the matched slots cover (the union of their [p, p + k), counted per mate as
// the vote counts it), any other gene's at most the bases the open slots cover.  matched >= ceil(c len) and matched > open make
// g the pair's one association (ReadAnalyzer.hpp:90-108) whatever a probe of the open slots would say: count = 1, gene g, exactly
// what the vote writes.  A pair that does not pass is simply left alone -- count[] stays 0 -- and the kernel launched behind this one
// with `pre_verdict` set passes over the pairs that have their result.  A chance anchor, a repeat, an indel, another gene's mate
// all show as disagreeing bases or as positions outside the anchor's run: such pairs do not pass.
#include "classify_common.hpp"

namespace shk {

constexpr int AV_WAVES = 4;            // waves per workgroup
constexpr uint32_t AV_PASSES = 6;      // consecutive passes (groups of up to six pairs) a wave takes at least ...
constexpr uint32_t AV_PASSES_MAX = 24; // ... and at most: large batches (ClassifyParams::av_passes, launch_anchor_verdict)
constexpr uint32_t AV_PPW = 6;         // pairs per pass at most

struct AvRaw { uint32_t d[9]; uint32_t sh; };

// 2-bit groups of a dword in reverse order
__device__ __forceinline__ uint32_t av_rev2(const uint32_t v)
{
  const uint32_t r = __builtin_bitreverse32(v);
  return ((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1);
}
// 16 x 2 bits -> 16 bits: bit i = both bits of group i set ... of `a`, where a has bit 2 i set iff group i qualifies
__device__ __forceinline__ uint32_t av_squeeze(uint32_t e)
{
  e = (e | (e >> 1)) & 0x33333333u;
  e = (e | (e >> 2)) & 0x0F0F0F0Fu;
  e = (e | (e >> 4)) & 0x00FF00FFu;
  return (e | (e >> 8)) & 0xFFFFu;
}

// Batches of one length per mate (RAGGED = false), or of mixed lengths (trimmed reads: RAGGED = true) -- those in the lane layout of
// the batch's LONGEST mates, a pair bringing its own two lengths (its chunks beyond them stay idle).  Pairs of at most 64 chunks of
// 32 bases.  POW2: the filter size is a power of two
template <bool POW2, bool HASQ, bool RAGGED>
__global__ __launch_bounds__(AV_WAVES * 64) void anchor_verdict_kernel(const ClassifyParams P)
{
  uint32_t L1 = P.uni_L1, L2 = P.uni_L2;
  if (P.uni_flag) {
    if (P.uni_flag[0] != (RAGGED ? 0u : 1u)) return;   // (the device's verdict: not this instantiation's kind of batch)
    L1 = P.uni_flag[1];
    L2 = P.uni_flag[2];
  }
  L1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)L1);
  L2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)L2);
  if (!P.seq2) L2 = 0u;
  const uint32_t k = P.k;
  const uint32_t c1 = (L1 + 31u) >> 5, c2 = (L2 + 31u) >> 5, lp = c1 + c2;   // chunks per mate, lanes per pair
  if (lp == 0u || lp > 64u) return;
  const uint32_t ppw = 64u / lp < AV_PPW ? 64u / lp : AV_PPW;                // pairs per pass
  const uint32_t n_reads = (uint32_t)P.n;
  const uint32_t n_pass = (n_reads + ppw - 1u) / ppw;
  const int lane = threadIdx.x & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * AV_WAVES + (threadIdx.x >> 6)));
  const uint32_t passes = P.av_passes;
  const uint64_t g64 = (uint64_t)wave * passes;
  if (g64 >= n_pass) return;
  uint32_t g = (uint32_t)g64;
  const uint32_t g_end = n_pass - g > passes ? g + passes : n_pass;

  // ---- lane geometry: pair of the pass, chunk of the pair, mate ----
  const uint32_t ln = (uint32_t)lane;
  const bool act = ln < ppw * lp;
  uint32_t pr = 0u;
  for (uint32_t j = 1u; j < AV_PPW; ++j) pr += (act && ln >= j * lp) ? 1u : 0u;
  const uint32_t cl = act ? ln - pr * lp : 0u;
  const bool in2 = cl >= c1;
  const uint32_t cm = in2 ? cl - c1 : cl;                 // chunk of the mate
  const uint32_t bofs = cm << 5;                          // its first base (mate coordinates)
  const uint32_t fm = pr * lp + (in2 ? c1 : 0u);          // first lane of my mate
  const uint32_t fo = pr * lp + (in2 ? 0u : c1);          // first lane of the pair's other mate
  const uint32_t cmn = in2 ? c2 : c1;                     // my mate's lanes
  const uint64_t pair_mask = (lp < 64u ? (1ull << lp) - 1ull : ~0ull) << (pr * lp);
  // the pair's field in the packed sums (pairs 0-2 in one word, 3-5 in another): ten bits while a pair has at most 21 lanes (672
  // bases), fifteen for the longer pairs, of which a pass holds two at most
  const uint32_t fbits = ppw > 2u ? 10u : 15u, fsh = fbits * (pr % 3u), fmask = (1u << fbits) - 1u;
  const bool hi3 = pr >= 3u;
  const uint8_t *sb = (in2 ? P.seq2 : P.seq1) + bofs;
  const uint8_t *qb = HASQ ? (in2 ? P.qual2 : P.qual1) + bofs : nullptr;
  const uint64_t *offm = in2 ? P.off2 : P.off1;
  const uint64_t kmer_mask = (1ull << (2u * k)) - 1ull;   // (k <= 31)
  const uint32_t kbits = (1u << k) - 1u;
  const uint32_t pmask = (uint32_t)((1ull << P.tab_lg) - 1ull) & (uint32_t)P.bf_mask;
  const uint32_t pspare = 1u << P.tab_lg;
  const uint32_t tagmask = (uint32_t)(P.bf_mask >> P.tab_lg);
  const uint4 *atab16 = reinterpret_cast<const uint4 *>(P.atab);
  const uint32_t ref_total = P.ref_total;
  // where the lane's mate buffer ends: an unguarded fetch reads up to 35 bytes from the chunk's first, so only chunks that far from the
  // end take it (RAGGED: from the offsets; else the mates are n reads of one length)
  const uint64_t end_m = RAGGED ? (act ? offm[n_reads] : 0ull) : (uint64_t)n_reads * (in2 ? L2 : L1);

  // ---- what depends on the pair's two lengths: once for a batch of one length per mate, per pass for trimmed reads ----
  uint32_t nb = 0u, tail = 0xFFFFFFFFu, nkm = 0u, nks = 0u, thr = 0u, cmid = 0u, f1 = 0u;
  bool has_kmer = false, sampled = false;
  auto set_lengths = [&](const uint32_t lm, const uint32_t lo) {      // my mate's length, the other mate's
    nb = (act && lm > bofs) ? (lm - bofs < 32u ? lm - bofs : 32u) : 0u;                 // bases of the mate in the chunk
    tail = nb < 32u ? 0xFFFFFFFFu << nb : 0u;
    nkm = lm >= k ? lm - k + 1u : 0u;
    const uint32_t nko = lo >= k ? lo - k + 1u : 0u;
    nks = nkm + nko;
    thr = cov_threshold(P.c, lm + lo);   // (a pair with invalid characters has a lower threshold: passing this one is sufficient)
    has_kmer = nb != 0u && bofs + k <= lm;            // a k-mer starts at the chunk's first base (and ends inside the chunk: k <= 31)
    // (a probe is a memory-side request on every index whose table has outgrown an XCD's L2: every other chunk is sampled -- slots 0,
    //  64, 128 of a 150-bp mate; all three hold an error, at 1 % per base, once in 150 mates)
    const uint32_t nkc = lm >= k ? ((lm - k) >> 5) + 1u : 0u;   // chunks of the mate at whose first base a k-mer starts
    sampled = has_kmer && (nkc <= 3u || (cm & 1u) == 0u);
    cmid = (nkm >> 1) >> 5;                           // the chunk in which the mate's middle slot starts
    f1 = pr * lp + ((in2 ? nko : nkm) ? 0u : c1);     // first lane of the pair's first mate that has slots
  };
  if (!RAGGED) {
    set_lengths(in2 ? L2 : L1, in2 ? L1 : L2);
    if (__builtin_amdgcn_readfirstlane((int)(nks == 0u || thr == 0u))) return;
  }

  // the 32 bytes of the lane's chunk of the read that starts at byte `om` of its mate's buffer (nbx of them belong to the read)
  auto issue = [&](const uint8_t *base, const uint64_t om, const uint32_t nbx) -> AvRaw {
    AvRaw r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.d[i] = 0u;
    r.sh = 0u;
    if (nbx != 0u) {
      const uint8_t *sp = base + om;
      const uint32_t sh = (uint32_t)reinterpret_cast<uintptr_t>(sp) & 3u;
      const uint32_t *q = reinterpret_cast<const uint32_t *>(sp - sh);
      r.sh = sh;
      if (om + bofs + 36ull <= end_m) {
#pragma unroll
        for (int i = 0; i < 9; ++i) r.d[i] = q[i];
      } else {
        const uint32_t last = sh + nbx - 1u;             // index of the last wanted byte relative to q: only dwords that hold a byte of the mate are touched
#pragma unroll
        for (int i = 0; i < 9; ++i) r.d[i] = last >= 4u * (uint32_t)i ? q[i] : 0u;
      }
    }
    return r;
  };
  // 32 bytes -> their 2-bit codes (first base LOW: classify_uni.hpp's `fw` stream) and one bit per byte: not one of ACGTacgt
  auto codes_of = [&](const AvRaw &r, uint64_t &code, uint32_t &inv) {
    uint32_t msb[2] = {0u, 0u};
    inv = 0u;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint32_t b = __builtin_amdgcn_alignbyte(r.d[i + 1], r.d[i], r.sh);
      uint32_t c4, i4;
      classify4(b, c4, i4);
      msb[i >> 2] |= pack4(c4) << (24 - 8 * (i & 3));      // first base in bits 31:30
      inv |= gather4(i4) << (4 * i);
    }
    code = ((uint64_t)av_rev2(msb[1]) << 32) | av_rev2(msb[0]);
  };
  auto qmask_of = [&](const AvRaw &r) -> uint32_t {
    uint32_t m = 0u;
#pragma unroll
    for (int i = 0; i < 8; ++i) m |= gather4(qmask4(__builtin_amdgcn_alignbyte(r.d[i + 1], r.d[i], r.sh), P.mq)) << (4 * i);   // FastqSplitter.hpp:104-109
    return m;
  };
  // (trimmed reads) a pass's offsets: the lane's mate's start and end of read ppw gg + pr; its length, and the other mate's by a
  // look at that mate's first lane
  struct Off { uint64_t a, b; };
  auto off_issue = [&](const uint32_t gg) -> Off {
    const uint32_t rd = ppw * gg + pr;
    Off o{0ull, 0ull};
    if (act && gg < g_end && rd < n_reads) { o.a = offm[rd]; o.b = offm[rd + 1u]; }
    return o;
  };
  auto lengths_of = [&](const Off &o, uint32_t &lm, uint32_t &lo) {
    const uint64_t d = o.b - o.a;
    lm = d < 0x7FFFFFFFull ? (uint32_t)d : 0x7FFFFFFFu;
    lo = c2 ? (uint32_t)__builtin_amdgcn_ds_bpermute((int)(fo << 2), (int)lm) : 0u;
    // (a read longer than the layout's mates -- the lengths came from the caller as a bound that does not hold --: nothing of it is looked at)
    if (lm > (in2 ? L2 : L1) || lo > (in2 ? L1 : L2)) { lm = 0u; lo = 0u; }
  };
  AvRaw cur, qcur;
  Off off_n{0ull, 0ull};
  uint32_t lm_c = 0u, lo_c = 0u;
  if (RAGGED) {
    const Off o0 = off_issue(g);
    lengths_of(o0, lm_c, lo_c);
    set_lengths(lm_c, lo_c);
    cur = issue(sb, o0.a, nb);
    if (HASQ) qcur = issue(qb, o0.a, nb);
    off_n = off_issue(g + 1u);
  } else {
    const uint32_t rd0 = ppw * g + pr;
    const uint32_t nb0 = (act && rd0 < n_reads) ? nb : 0u;
    cur = issue(sb, (uint64_t)rd0 * (in2 ? L2 : L1), nb0);
    if (HASQ) qcur = issue(qb, (uint64_t)rd0 * (in2 ? L2 : L1), nb0);
  }
  for (; g < g_end; ++g) {
    AvRaw nxt, qnxt;
    uint32_t lm_n = 0u, lo_n = 0u;
    if (RAGGED) {
      // the next pass's lengths (its offsets were asked for a pass ago), its bases, and the offsets of the pass behind it
      lengths_of(off_n, lm_n, lo_n);
      const uint32_t nbn = (act && g + 1u < g_end && lm_n > bofs) ? (lm_n - bofs < 32u ? lm_n - bofs : 32u) : 0u;
      nxt = issue(sb, off_n.a, nbn);
      if (HASQ) qnxt = issue(qb, off_n.a, nbn);
      off_n = off_issue(g + 2u);
    } else {
      const uint32_t rdn = ppw * (g + 1u) + pr;
      const uint32_t nbn = (act && g + 1u < g_end && rdn < n_reads) ? nb : 0u;
      nxt = issue(sb, (uint64_t)rdn * (in2 ? L2 : L1), nbn);
      if (HASQ) qnxt = issue(qb, (uint64_t)rdn * (in2 ? L2 : L1), nbn);
    }
    const uint32_t rd = ppw * g + pr;
    const bool live = act && rd < n_reads && (!RAGGED || nb != 0u);
    // ---- the chunk's 32 bases as 2-bit codes, and which of them are invalid (N, a masked quality, behind the mate's end) ----
    uint64_t code;
    uint32_t inv;
    codes_of(cur, code, inv);
    inv |= tail;
    if (HASQ) inv |= qmask_of(qcur);
    if (!live) inv = 0xFFFFFFFFu;
    // ---- the k-mer that starts at the chunk's first base ----
    const uint64_t x = code & kmer_mask;                                       // first base low: ~x is the reverse complement (kmer_utils.hpp:47-55)
    const bool kvalid = has_kmer && (inv & kbits) == 0u;
    const uint64_t fwd = (((uint64_t)av_rev2((uint32_t)x) << 32) | av_rev2((uint32_t)(x >> 32))) >> (64u - 2u * k);   // first base most significant (kmer_utils.hpp:67-69)
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
    // ---- the mate's anchor: of its sampled k-mers that are in the index the one nearest the mate's middle (the first at or behind the
    // middle chunk, else the last in front of it) -- `refext` counts at most 254 positions either way, which from the middle reaches both
    // ends of mates of up to 500 bases, and from the mate's first k-mer only those of up to 270 ----
    const uint64_t HB = __ballot(hit);
    const uint64_t mine = (HB >> fm) & ((1ull << cmn) - 1ull);
    const bool have = act && nkm != 0u && mine != 0ull;
    const uint64_t behind = mine >> cmid;
    const uint32_t src = fm + (have ? (behind ? cmid + (uint32_t)__builtin_ctzll(behind) : 63u - (uint32_t)__builtin_clzll(mine)) : 0u);
    const uint32_t a_src = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)a);
    const uint32_t x0 = a_src & 0x7FFFFFFFu, s0 = (src - fm) << 5;             // reference position and slot (mate coordinates) of the anchor
    const bool opp = (a_src >> 31) != 0u;
    // ---- what the reference says there: the anchor's run, the 32 bases this chunk stands against, the lists under its 32 slots ----
    // (the base at mate position b stands against reference base x0 - s0 + b, or, on the other strand, against the complement of base
    //  x0 + s0 + k - 1 - b; slot b on position x0 - s0 + b, or x0 + s0 - b: classify_uni.hpp, anchored extension (3))
    const uint32_t ev = have ? P.refext[x0] : REFEXT_NONE;
    const uint32_t lo = opp ? x0 + s0 + k - 32u - bofs : x0 + bofs - s0;       // (mod 2^32: a chunk that would leave the reference fails the bound)
    const bool inr = have && lo < ref_total;
    const uint32_t ls = inr ? lo : 0u;
    const uint32_t r0 = P.ref2[ls >> 4], r1 = P.ref2[(ls >> 4) + 1u], r2 = P.ref2[(ls >> 4) + 2u];
    const uint32_t w = opp ? x0 + s0 - bofs - 31u : x0 + bofs - s0;
    const bool inw = have && w < ref_total;
    const uint32_t wd = inw ? w : 0u;
    const uint32_t f0 = P.refmul[wd >> 5], f1w = P.refmul[(wd >> 5) + 1u];
    uint32_t ok32;                                                             // bases that agree with the reference and are valid characters
    {
      uint32_t Glo = __builtin_amdgcn_alignbit(r1, r0, (ls & 15u) << 1), Ghi = __builtin_amdgcn_alignbit(r2, r1, (ls & 15u) << 1);
      if (opp) {   // the other strand: base order reversed (2-bit groups), complemented
        const uint32_t t = ~av_rev2(Ghi);
        Ghi = ~av_rev2(Glo);
        Glo = t;
      }
      const uint32_t dl = (uint32_t)code ^ Glo, dh = (uint32_t)(code >> 32) ^ Ghi;
      const uint32_t e32 = av_squeeze(~(dl | (dl >> 1)) & 0x55555555u) | (av_squeeze(~(dh | (dh >> 1)) & 0x55555555u) << 16);
      ok32 = inr ? (e32 & ~inv) : 0u;
    }
    // ---- which of the chunk's 32 slots (the k-mers that START at its bases) equal the reference's k-mer: k agreeing bases in a row,
    // i.e. the agreement bits of this chunk and of the next one (k <= 31), eroded by k ----
    uint32_t eq32;
    {
      const uint32_t n1 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((ln + 1u) & 63u) << 2), (int)ok32);
      uint64_t E = (uint64_t)ok32 | ((uint64_t)(cm + 1u < cmn ? n1 : 0u) << 32);
      uint32_t len = 1u;
      while (2u * len <= k) { E &= E >> len; len <<= 1; }
      if (len < k) E &= E >> (k - len);
      eq32 = (uint32_t)E;
    }
    // the lists under those 32 positions: 1 = not a single-gene list
    uint32_t fo32 = 0xFFFFFFFFu;
    {
      uint32_t F = __builtin_amdgcn_alignbit(f1w, f0, wd & 31u);               // bit j: position w + j
      if (opp) F = __builtin_bitreverse32(F);
      if (inw) fo32 = F;
    }
    const uint32_t nex = nkm > bofs ? nkm - bofs : 0u;                        // slots of the mate that start in this chunk
    const uint32_t ex32 = (live && nex != 0u) ? (nex >= 32u ? 0xFFFFFFFFu : (1u << nex) - 1u) : 0u;
    const uint32_t m32 = eq32 & ex32 & ~fo32;                                  // matched: g's, and g's alone
    const uint32_t o32 = ex32 & ~m32;                                          // open
    // ---- bases of this chunk that the matched / the open slots cover: slots of this chunk and of the one in front, dilated by k ----
    uint32_t covm, covo;
    {
      const uint32_t pm = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((ln + 63u) & 63u) << 2), (int)m32);
      const uint32_t po = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((ln + 63u) & 63u) << 2), (int)o32);
      uint64_t DM = (uint64_t)(cm >= 1u ? pm : 0u) | ((uint64_t)m32 << 32);
      uint64_t DO = (uint64_t)(cm >= 1u ? po : 0u) | ((uint64_t)o32 << 32);
      uint32_t len = 1u;
      while (2u * len <= k) { DM |= DM << len; DO |= DO << len; len <<= 1; }
      if (len < k) { DM |= DM << (k - len); DO |= DO << (k - len); }
      covm = (uint32_t)__builtin_popcount((uint32_t)(DM >> 32) & ~tail);
      covo = (uint32_t)__builtin_popcount((uint32_t)(DO >> 32) & ~tail);
    }
    // every slot of the mate on a position of the anchor's run (valid k-mer starts of one record)?  (slots [0, nkm) <-> positions
    // x0 - s0 ... x0 - s0 + nkm - 1, or mirrored)
    const uint32_t before = s0, after = nkm - 1u - s0, left = (ev >> 16) & 0xFFu, right = ev >> 24;
    bool okm = nkm == 0u || (have && ev != REFEXT_NONE && (opp ? (right >= before) & (left >= after) : (left >= before) & (right >= after)));
    const uint32_t gene = ev & 0xFFFFu;
    const uint32_t gene1 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(f1 << 2), (int)gene);
    okm = okm && (nkm == 0u || gene == gene1);
    const uint64_t BAD = __ballot(live && !okm);
    // per pair: the bases its matched / open slots cover, summed over its lanes in packed fields (three pairs a word)
    const uint32_t sm_lo = wave_sum_u32(hi3 ? 0u : covm << fsh), so_lo = wave_sum_u32(hi3 ? 0u : covo << fsh);
    uint32_t sm_hi = 0u, so_hi = 0u;
    if (ppw > 3u) {   // (wave-uniform)
      sm_hi = wave_sum_u32(hi3 ? covm << fsh : 0u);
      so_hi = wave_sum_u32(hi3 ? covo << fsh : 0u);
    }
    const uint32_t cov_m = ((hi3 ? sm_hi : sm_lo) >> fsh) & fmask, cov_o = ((hi3 ? so_hi : so_lo) >> fsh) & fmask;
    const bool pass = (BAD & pair_mask) == 0ull && thr != 0u && cov_m >= thr && cov_m > cov_o;
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
    if (RAGGED) set_lengths(lm_n, lo_n);
  }
}

// does the kernel apply to batches of these lengths (`ragged`: the longest mates of a batch of mixed lengths)?  Lengths only the device
// knows: it decides itself
bool anchor_verdict_applies(const ClassifyParams &p)
{
  if (!p.ref_total || !p.refext || !p.refmul || !p.atab || !p.ref2 || p.n == 0 || p.n >= (1ull << 32)) return false;
  if (p.uni_flag) return true;
  const uint32_t lp = ((p.uni_L1 + 31u) >> 5) + (((p.seq2 ? p.uni_L2 : 0u) + 31u) >> 5);
  return lp != 0u && lp <= 64u;
}

int launch_anchor_verdict(const ClassifyParams &p, bool pow2, bool ragged, hipStream_t s)
{
  // (lengths only the device knows: a pass may hold one pair only -- the grid is sized for that, waves without work return at once)
  const uint32_t lp = p.uni_flag ? 0u : ((p.uni_L1 + 31u) >> 5) + (((p.seq2 ? p.uni_L2 : 0u) + 31u) >> 5);
  const uint64_t ppw = p.uni_flag ? 1u : (64u / lp < AV_PPW ? 64u / lp : AV_PPW);
  const uint64_t n_pass = (p.n + ppw - 1) / ppw;
  // passes per wave: six while that still gives every CU several workgroups (the command's batches of 2^16 ... 2^18 pairs), up to 24 for
  // the large ones (10 M pairs at 100 % on-target, 1 000 / 60 000 genes: 5.56 / 8.14 ms with six, 5.43 / 8.08 with 16 or 24)
  ClassifyParams q = p;
  q.av_passes = (uint32_t)std::min<uint64_t>(AV_PASSES_MAX, std::max<uint64_t>(AV_PASSES, (p.n / AV_PPW) / 16384));
  const uint64_t waves = (n_pass + q.av_passes - 1) / q.av_passes;
  const unsigned grid = (unsigned)((waves + AV_WAVES - 1) / AV_WAVES);
  const bool hasq = p.hasq != 0;
#define AVL(P2_, HQ_, RG_) hipLaunchKernelGGL((anchor_verdict_kernel<P2_, HQ_, RG_>), dim3(grid), dim3(AV_WAVES * 64), 0, s, q)
  if (ragged) {
    if (pow2) { if (hasq) AVL(true, true, true); else AVL(true, false, true); }
    else { if (hasq) AVL(false, true, true); else AVL(false, false, true); }
  } else {
    if (pow2) { if (hasq) AVL(true, true, false); else AVL(true, false, false); }
    else { if (hasq) AVL(false, true, false); else AVL(false, false, false); }
  }
#undef AVL
  return hipGetLastError() == hipSuccess ? SHK_OK : SHK_ERR_HIP;
}

}  // namespace shk
