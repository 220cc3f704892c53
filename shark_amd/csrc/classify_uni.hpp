// classify_uni.hpp -- classify_uni_kernel and its launcher template.  Included by classify_uni_u<U>.hip (one translation unit
// per unroll U: the instantiations of one U take about half a minute to compile, and there are six).
#pragma once
#include "classify_common.hpp"

namespace shk {

// ---------------------------------------------------------------------------
// classify_uni_kernel: batches whose reads all have ONE length per mate (what a sequencer delivers) on an index
// in LDS-summary + position-table mode.  Same algorithm and data layout as classify_fast_kernel<.., PM_LDS_TAB>, but
// everything that depends only on the read lengths is computed once per wave instead of once per read: which mate and
// which 8 bases a lane stages, tail masks, LDS addresses; a read's bytes are at read * L, so no offsets are loaded at
// all; the 8 bases are fetched as three unconditional aligned dwords (only the last reads of the batch, where the
// third dword could leave the buffer, take the guarded loads).  The code for a pair without any hit -- stage, U
// canonical k-mers, U hashes, U summary probes, table probes of the few that pass -- is one straight line; everything
// a hit needs (decode, lazy validity, vote, emit) sits behind the wave-uniform "something matched" branch and re-reads
// its parameters there, so it holds no registers while off-target reads stream through.  count[] is zeroed by the host
// before the launch; only reads with associations write it.
// ---------------------------------------------------------------------------
#ifndef SHK_STREAM_READS
#define SHK_STREAM_READS 0
#endif
__device__ __forceinline__ Raw8 load8_issue_all(const uint8_t *p, uint32_t nbytes)
{
  const uint32_t sh = (uint32_t)reinterpret_cast<uintptr_t>(p) & 3u;
  const uint32_t *q = reinterpret_cast<const uint32_t *>(p - sh);
  Raw8 r;
#if SHK_STREAM_READS
  r.d0 = __builtin_nontemporal_load(q);
  r.d1 = __builtin_nontemporal_load(q + 1);
  r.d2 = __builtin_nontemporal_load(q + 2);
#else
  r.d0 = q[0];
  r.d1 = q[1];
  r.d2 = q[2];
#endif
  r.shn = sh | (nbytes << 4);
  return r;
}

// kernel arguments re-read where they are needed: scalar loads from the kernarg segment (the ClassifyParams is the
// kernel's only argument, at offset 0); the empty asm keeps LICM from turning them into loop-long SGPRs.  (Taking the
// address of the by-value parameter instead would make the compiler copy all of it to scratch memory.)
typedef const ClassifyParams __attribute__((address_space(4))) * KernargParams;
__device__ __forceinline__ KernargParams kernarg_params()
{
  KernargParams p = (KernargParams)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return p;
}

// (the ends of the mates' buffers are read once per wave through the constant address space: scalar loads.  A read's offsets are
// NOT: scalar loads are counted by lgkmcnt like LDS accesses and return out of order, so every wait for LDS data would also wait for
// the offsets of the read after next on their way from memory -- measured on the exact-table kernel: 7.4 -> 8.7 ms per 10 M trimmed pairs)
typedef const uint64_t __attribute__((address_space(4))) *ConstU64;

// ---- the bound cut -----------------------------------------------------------------------------------------------
// ReadAnalyzer.hpp:104 keeps a read iff max >= c*len, where max is the largest per-gene coverage: the size of the union of
// the intervals [p, p+k) of that gene's hits (header comment).  A gene's coverage is therefore at most the number of
// bases covered by ALL k-mers that are in the filter.  classify_uni_kernel probes the slots of its first E rounds
// (packed positions < 64 E) first; when none of them is in the filter, every hit the read can still have lies in the
// remaining slots, which cover bases_behind(64 E) bases.  If that is below the threshold, no gene can reach it, the read
// has no association whatever the remaining probes would say -- and they are not made.  (2 x 150 bp, k = 17, c = 0.6:
// after two rounds the rest covers 22 + 150 = 172 < 180 bases, so an off-target pair costs 128 probes instead of 320.)
// The result is the reference's for every read; tests/test_gpu_parity.py walks chimeric reads across the boundary.

// (cov_threshold: classify_common.hpp)
// bases covered by the existing slots at packed positions >= s (mate 1: slots [0, nk1) cover [0, l1); mate 2 likewise at P2)
__device__ __forceinline__ uint32_t bases_behind(const uint32_t s, const uint32_t nk1, const uint32_t nk2, const uint32_t P2,
                                                 const uint32_t l1, const uint32_t l2)
{
  uint32_t u = s < nk1 ? l1 - s : 0u;
  const uint32_t s2 = s > P2 ? s - P2 : 0u;
  u += s2 < nk2 ? l2 - s2 : 0u;
  return u;
}

// The number of first rounds is a compile-time constant of the code that runs (exact register liveness; a run-time split
// keeps every round's state alive across both phases and spills).  Per U, the candidates that the usual shapes need at the
// reference's default c = 0.6 (paired mates of equal length: about half the rounds; U = 5 is 2 x 150 bp: two rounds, three
// for c down to 0.37; U = 3 also serves single-end 150 bp reads: one round).  A read whose bound holds for neither is probed
// in one go.
template <int U> struct CutPlan { static constexpr int E0 = U / 2, E1 = U / 2; };
template <> struct CutPlan<3> { static constexpr int E0 = 1, E1 = 2; };
template <> struct CutPlan<5> { static constexpr int E0 = 2, E1 = 3; };
// (-DSHK_NO_CUT=1: a build without the cut, for A/B timing)
#ifndef SHK_NO_CUT
#define SHK_NO_CUT 0
#endif
#ifndef SHK_NO_TOL
#define SHK_NO_TOL 0
#endif
// the bound cut behind the L2-resident summary as well (-DSHK_CUT_SUM=0: not)
#ifndef SHK_CUT_SUM
#define SHK_CUT_SUM 1
#endif
// (-DSHK_NO_ANCHOR=1: a build without the anchored extension, for A/B timing; at run time SHK_NO_ANCHOR=1 when the index is built)
#ifndef SHK_NO_ANCHOR
#define SHK_NO_ANCHOR 0
#endif
// (-DSHK_ANCH_CUT=n: timing-only ablation, results are wrong: the anchored path ends behind its step n)
#ifndef SHK_ANCH_CUT
#define SHK_ANCH_CUT 0
#endif
// how the anchored extension compares a mate with the reference: 1 = base by base (one xor per 16 bases into a bit stream, then the
// validity window test per slot), 0 = k-mer by k-mer (round 3: two 2k-bit windows per slot and round).  Same results; measured on one
// box per 10 M pairs at 50 / 100 % on-target (profiles/README.md, round 4): configs[2] index 20.4 / 17.8 ms against 21.8 / 18.2;
// 1 000 genes 11.25 / 14.15 against 11.5 / 14.2.  (The form alone changed little -- its first version, whose refpay loads waited for
// the match bits, was 9 % SLOWER: a second dependent memory round trip; what brought both forms under round 3's 21.7 / 18.9 ms are
// the sampled bucket carrying its reference position itself (atab) and the vote's scalar bounds.)
#ifndef SHK_ANCH_BASEWISE
#define SHK_ANCH_BASEWISE 1
#endif
#if SHK_ANCH_CUT != 0 && !defined(SHK_TIMING_ONLY)
#error "-DSHK_ANCH_CUT=n builds a library whose results are WRONG (timing-only ablation): say so with -DSHK_TIMING_ONLY as well"
#endif
#if defined(SHK_ABLATION) && !defined(SHK_TIMING_ONLY)
#error "-DSHK_ABLATION builds a library that can return wrong results (SHK_ABLATE bits): say so with -DSHK_TIMING_ONLY as well"
#endif
// (-DSHK_NO_SPARSE=1: a build without the sparse first round of one-gene indices, for A/B timing; at run time SHK_NO_SPARSE=1 when the index is built)
// -DSHK_STAMPS=1 (a diagnostic build of one translation unit, tools/stamps.sh): the three-pairs instantiation reads the shader
// clock (s_memtime) at its phase boundaries and every wave adds its per-phase sums to shk_stamp_acc -- the cycle budget of
// profiles/r05_headline_stamps.json.  A stamp waits for the scalar unit's outstanding loads (lgkmcnt covers the LDS as well), so a
// phase that ends behind a probe is charged the probe's LDS latency; the product build contains none of this.
#ifndef SHK_STAMPS
#define SHK_STAMPS 0
#endif
#if SHK_STAMPS
// per wave (4096 = 256 workgroups of 16 waves): its sums, added up over the launches since the last reset -- a row of its own per wave:
// 4 096 waves adding to sixteen shared words as they finish kept the memory side busy for 0.6 ms, which the waves still running paid
__device__ unsigned long long shk_stamp_acc[4096 * 16];
__device__ unsigned long long shk_stamp_waves[2 * 4096];   // per wave of the last launch: the 100 MHz counter at its loop's start and end
#define SHK_STAMP(i) do { if constexpr (TRI) { const uint64_t t__ = __builtin_readcyclecounter(); st_acc[i] += t__ - st_last; st_last = t__; } } while (0)
#else
#define SHK_STAMP(i) do {} while (0)
#endif
#ifndef SHK_DYN_ALL
#define SHK_DYN_ALL 1   // (0: turn-taking only where the exact table sits in LDS)
#endif
#ifndef SHK_NO_DYN
#define SHK_NO_DYN 0   // (-DSHK_NO_DYN=1: every wave walks its own fixed sequence of reads, for A/B timing)
#endif
#ifndef SHK_NO_TILE_FIRST
#define SHK_NO_TILE_FIRST 0   // (-DSHK_NO_TILE_FIRST=1: a build without the tiles' round in front of three staged pairs)
#endif
#ifndef SHK_NO_SPARSE
#define SHK_NO_SPARSE 0
#endif
// (-DSHK_NO_PARTIAL=1: the round between the cut's two stops is always probed whole, for A/B timing)
#ifndef SHK_NO_PARTIAL
#define SHK_NO_PARTIAL 0
#endif
// the partial round leaves room for matches that cover SHK_PART_MARGIN_K * k + 1 bases of one gene
#ifndef SHK_PART_MARGIN_K
#define SHK_PART_MARGIN_K 2
#endif
// slots per mate that the anchored extension samples through the table (a power of two, at most 16)
#ifndef SHK_ANCH_SAMPLE
#define SHK_ANCH_SAMPLE 4
#endif
// MODE: PM_LDS_TAB(_MOD) as described above; PM_TAB(_MOD) / PM_TAB_SUM for indices too dense for the LDS summary -- there a
// probe costs memory traffic, so a slot's existence and validity are settled BEFORE its probe (as in process_read), and
// only real k-mers (that pass the L2-resident summary, PM_TAB_SUM) read their bucket.
// LSL = 21: no summary -- the workgroup keeps the index's LDS-resident exact table (shark_internal.hpp LTAB_*; 144 KiB, one
// 1024-thread workgroup per CU): a probe is two LDS reads, a pair of a tiny index touches no memory but its own bases.
// LSL: log2 of the LDS summary's bits.  18 = 32 KiB, several 512-thread workgroups per CU (the sparse indices of a few
// genes); 20 = 128 KiB shared by ONE 1024-thread workgroup per CU -- four times the reach (pass rate <= 30 % up to ~3x10^5
// set bits, i.e. panels of a hundred genes) at 4 waves per SIMD, for indices that would otherwise probe an L2-resident
// summary through the vector L1 (one cache line per clock per CU) for every k-mer.
template <int U, int MODE, int LSL>
struct UniGeom {
  static constexpr bool LX = pm_lds(MODE) && LSL == 21;   // the exact table of a tiny index in LDS instead of a summary
  static constexpr int WAVES = (pm_lds(MODE) && LSL >= 20) ? (LSL == 21 ? SHK_LX_WAVES : 16) : 8;
  static constexpr int THREADS = WAVES * 64;
  // LDS summary: 3 x 512 threads per CU at <= 80 VGPRs (SHK_UNI_WAVES); table modes likewise (SHK_TAB_WAVES, classify_common.hpp)
  // (U = 10, and the table modes beyond U = 5 -- they carry the anchored extension --: 4 waves per SIMD, 128 VGPRs; at 80 the U = 8 table
  //  kernels spilled 50-130 VGPRs)
  static constexpr int MIN_WAVES = (WAVES == 16 || LX) ? WAVES / 4 : ((U > 8 || (U > 5 && !pm_lds(MODE))) ? 4 : (U > 5 ? 6 : (pm_lds(MODE) ? SHK_UNI_WAVES : (MODE == PM_KTAB ? SHK_KT_WAVES : SHK_TAB_WAVES))));
  static constexpr uint32_t SUM_BITS = pm_lds(MODE) ? (LX ? LTAB_BYTES * 8u : (1u << LSL)) : 0u;   // what the workgroup keeps in LDS
  static constexpr uint32_t SUM_WORDS64 = SUM_BITS / 64;
};

// UNI = false: the same kernel for batches of mixed read lengths (trimmed reads).  The geometry is then per read --
// offsets prefetched two reads ahead, bases one read ahead, as classify_fast_kernel does it -- but the structure is this
// kernel's: straight-line miss path, the hit path behind one branch with its own parameter loads, count[] pre-zeroed.  A
// read with more than 64 U slots or more than 64 staging groups goes to the general kernel's queue.
// CLS (with UNI; exact-table instantiations): a batch of mixed lengths taken CLASS BY CLASS.  The pre-pass (class_hist_kernel /
// class_plan_kernel / class_scatter_kernel, classify.hip) has sorted the batch's pairs by their two lengths into one list of entries
// {o1, o2 | read, "unguarded loads are safe"}; the list is cut into CLS_SHARES equal shares, a wave takes shares, and walks the
// classes its share runs through.  Within a class everything that depends on the lengths -- slots per mate, tail masks, the cut's
// and the sparse round's plan, and whatever the compiler derives from them -- is a wave constant exactly as for a uniform batch;
// a pair brings its two offsets.  (The ragged instantiation re-derives all that per read: 317 VALU + 168 scalar instructions per
// trimmed pair against 199 + 78; 10 M pairs trimmed to 100-150 bases: 7.1 -> 4.6 ms.)
// LXM (exact-table instantiations, uniform batches): an index of SEVERAL genes -- the sparse first rounds with the early decision's
// argument (sparse_first).  A compile-time form: as a run-time one it cost the one-gene index 3 % (4.18 -> 4.30 ms per 10 M pairs).
// TRI (exact-table instantiations, uniform batches without qualities): THREE pairs per staging pass.  Staging is what a pair
// costs before its first probe -- 65 VALU wave-instructions, a third of an off-target pair's -- and with 8 bases per lane only 38 of a
// wave's 64 lanes have a 2 x 150 bp pair's bases to stage.  Here a lane stages 16 bases: ten lanes a mate, twenty a pair, and the
// wave stages three consecutive pairs at once into three staging areas (60 lanes busy; the twice longer per-lane work is paid once
// for three pairs), then classifies them one after the other exactly as before.  Mate 2 is packed at 16 x ceil(L1 / 16) instead of
// 8 x ceil(L1 / 8); results do not depend on where mate 2 is packed (see FIXLAY).  tri_applies() says for which lengths.
__host__ __device__ inline bool tri_applies(const uint32_t L1, const uint32_t L2, const uint32_t k, const uint32_t S)
{
  const uint32_t c1 = (L1 + 15u) >> 4, c2 = (L2 + 15u) >> 4;
  const uint32_t nk2 = L2 >= k ? L2 - k + 1u : 0u, nk1 = L1 >= k ? L1 - k + 1u : 0u;
  const uint32_t ns = nk2 ? (c1 << 4) + nk2 : nk1;
  return L1 != 0u && c1 + c2 <= 21u && ns <= S;
}
// TRO (round 6; a form of TRI): a batch of MIXED read lengths (trimmed reads) through the three-pairs kernel, in the lane layout of the
// batch's LONGEST mates, a read found by its offsets.  What lies behind a read's own end in that layout is marked invalid -- as an N
// would be -- and that is all there is to it: ReadAnalyzer counts valid characters for `len` (ReadAnalyzer.hpp:46-49), skips every k-mer
// with an invalid character (:64-71) and clamps the step between two hits at k (:56-62,:79-86), so a read followed by invalid
// characters -- the mate joiner is one (FastqSplitter.hpp:63) -- has the associations of the read itself.  Slots behind a read's end
// "exist" in the layout and are not valid k-mers: such a pair is treated as one with invalid characters (a validity window per slot).
// The plan's structure (rounds, tiles) is the layout's for every pair; its bounds -- what the slots behind a stop cover, the threshold --
// are computed from the pair's own two lengths (see per_pair).  No sorting by length, no pass over the offsets but
// uniform_check_kernel's.
template <int U, int MODE, bool HASQ, int LSL, bool UNI, bool CLS = false, bool LXM = false, bool TRI = false, bool TFK = false, bool TRO = false>
__global__ __launch_bounds__((UniGeom<U, MODE, LSL>::THREADS), (UniGeom<U, MODE, LSL>::MIN_WAVES)) void classify_uni_kernel(const ClassifyParams P)
{
  static_assert(!TRO || TRI, "TRO is a form of the three-pairs instantiation");
  static_assert(!CLS || UNI, "CLS is a form of the uniform instantiation");
  static_assert(!LXM || (UNI && !CLS && pm_lds(MODE) && LSL == 21), "LXM is a form of the uniform exact-table instantiation");
  static_assert(!TRI || (UNI && !CLS && !HASQ && pm_lds(MODE) && LSL == 21 && U <= 8), "TRI is a form of the uniform exact-table instantiation without qualities");
  constexpr bool POW2 = pm_pow2(MODE);
  constexpr bool LSUM = pm_lds(MODE);
  constexpr bool SUM = MODE == PM_TAB_SUM;
  // PM_KTAB: a table mode whose probes go to the k-mer keyed, minimiser-bucketed table (DeviceIndex::ktab) instead of the position
  // table: no XXH64 (the key is the canonical k-mer itself; the table holds the filter's false positives too), and consecutive slots
  // share their 128-byte line.  Everything behind the probe -- what a slot's low word means, the cut, the votes -- is the same.
  constexpr bool KT = MODE == PM_KTAB;
  // the bound cut.  (Behind the L2-resident summary it used to be left out: an off-target pair is cheap there and the second
  // dependent step cost on-target pairs more than the cut saved.  Since the anchored extension takes the on-target pairs off this
  // path it pays: 250 / 1 000 genes at 0 % on-target 10.1 / 12.2 -> 8.2 / 9.8 ms per 10 M pairs, at 50 % 11.5 / 12.6 -> 11.2 / 12.0,
  // at 100 % 14.3 / 14.8 -> 14.8 / 15.3)
  constexpr bool CUT = (!SUM || SHK_CUT_SUM) && !SHK_NO_CUT;
  constexpr bool TOL = CUT && !pm_lds(MODE) && !SHK_NO_TOL;   // table modes: matches are counted, and there is a second cut point
  constexpr bool ACCEPT = !SHK_NO_ACCEPT;                      // the early decision (vote<J> with J < U)
  constexpr bool ANCH = !pm_lds(MODE) && !SHK_NO_ANCHOR;        // table modes: the anchored extension (when the index carries ref2 / refpay / anchor)
  using UG = UniGeom<U, MODE, LSL>;
  constexpr bool LX = UG::LX;
  constexpr int WAVES = UG::WAVES;
  constexpr uint32_t S = 64 * U;
  // per wave: fw + rv + validity, and -- table modes, for the base-by-base form of the anchored extension -- one more bit stream: agreement with the reference
  constexpr uint32_t WORDS = stage_words_for(S) + ((ANCH && SHK_ANCH_BASEWISE) ? vbit_words_for(S) : 0u);
  constexpr uint32_t AREAS = TRI ? 3u : 1u;      // staging areas per wave
  // DYN (exact table in LDS, uniform batches): the workgroup's waves take their reads (TRI: triples) from a counter in LDS instead of
  // every wave walking its own fixed sequence.  The SIMD issues oldest wave first: with fixed sequences of equal length the four
  // waves of a SIMD finished at 2.8 / 3.5 / 4.4 / 5.3 ms of a 5.3 ms launch (s_memrealtime at every wave's start and end, the stamped
  // build of tools/stamps.sh) -- the SIMD spent the last third of the launch with two waves, then one, to hide its latencies
  // (likewise behind an LDS or L2 summary -- 100 / 1 000 genes at 0 / 50 / 100 % on-target 5.0 / 10.7 / 14.2 -> 4.7 / 10.4 / 13.9 and 9.7 / 11.7 /
  //  14.7 -> 9.4 / 11.5 / 14.5 ms per 10 M pairs; not on the position / minimiser tables of large indices, whose waves wait on memory at even
  //  rates: 17.6-18.1 / 17.9-18.4 -> 18.0-18.2 / 18.7-18.8 ms at 0 / 50 %, the turn's registers spilled there)
  constexpr bool DYN = (LX || (SHK_DYN_ALL && (pm_lds(MODE) || MODE == PM_TAB_SUM))) && UNI && !CLS && !SHK_NO_DYN;
  // (Workgroups taking CHUNKS of the batch from one counter of the launch, on top of this, was built and measured: 3.15 -> 3.21 ms.  The
  //  spread of the workgroups' ends that suggested it -- 4.2 ... 4.85 ms in the stamped build -- was that build's own doing: 4 096 waves
  //  adding their sums to sixteen words of one cache line as they finish.)
  // CLS: the same for the SHARES of a batch taken class by class (sixteen per wave)
  constexpr bool DYNC = CLS && !SHK_NO_DYN;
  __shared__ uint64_t lds[UG::SUM_WORDS64 + WAVES * WORDS * AREAS + ((DYN || DYNC) ? 2 : 0)];
  uint32_t *const dyn_ctr = reinterpret_cast<uint32_t *>(lds + UG::SUM_WORDS64 + WAVES * WORDS * AREAS);
  const int lane = threadIdx.x & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint32_t L1 = P.uni_L1, L2 = P.uni_L2;
  if (P.uni_flag) {
    // every launch is made when only the device knows what the batch is like -- 0: ragged, 1: uniform, 2: by classes, 3: mixed lengths
    // through the three-pairs kernel (TRO) -- and exactly one of them works
    const uint32_t verdict = P.uni_flag[0];
    if (verdict != (CLS ? 2u : (TRO ? 3u : (UNI ? 1u : 0u)))) return;
    L1 = P.uni_flag[1];
    L2 = P.uni_flag[2];
  } else if (CLS) {
    return;
  }
  L1 = __builtin_amdgcn_readfirstlane(L1);
  L2 = __builtin_amdgcn_readfirstlane(L2);
  // the three-pairs-per-pass instantiation and the ordinary one are launched side by side when only the device knows the lengths
  // (P.tri): exactly one of them works
  if (TRI && !tri_applies(L1, L2, P.k, 64u * U)) return;
  if (!TRI && UNI && !CLS && P.tri && tri_applies(L1, L2, P.k, 64u * U)) return;
  // G: staging groups (8 bases) per lane -- one up to 512 bases per pair, two beyond (U = 10: 2 x 300 bp)
  constexpr int G = U > 8 ? 2 : 1;
  // FIXLAY: the ragged instantiation that keeps the exact table in LDS (one-gene indices: instruction-bound, 128 VGPRs to its name).
  // The others keep the per-read layout: the table modes wait on memory, not on instructions, and holding a lane's layout constants
  // next to a read's offsets, lengths and plan across the loop took them from 80 to 106-119 VGPRs -- spilled, or at four waves per
  // SIMD, trimmed reads ran a quarter slower on them (100 / 1 000 / 60 000 genes: 12.9 / 13.9 / 21.1 -> 16.7 / 18.2 / 25.3 ms);
  // the LDS-summary instantiations (80 VGPRs, three workgroups per CU) spilled as well
  constexpr bool FIXLAY = !UNI && UG::LX;
  if (FIXLAY) {
    // Trimmed reads: (L1, L2) are the LONGEST mates of the batch, and the whole batch is staged in THEIR packed layout -- mate 2 of
    // every pair at P2 = L1 rounded up to 8, whatever the pair's own mate 1 --, so which bases a lane stages, their LDS addresses and
    // the mate a slot belongs to are computed once per wave, as for uniform batches; a read brings only its two lengths (the number
    // of slots per mate, the tail masks of its last groups) and its plan (below).  Results do not depend on where mate 2 is packed:
    // an interval [p, p + k) never reaches the other mate for any P2 >= the pair's mate 1.  Should the longest pair not fit this
    // specialisation, the layout is that of the longest mates that do; longer reads go to the general kernel's queue, as ever.
    const uint32_t k0 = P.k;
    const uint32_t nkA = L1 >= k0 ? L1 - k0 + 1u : 0u, nkB = L2 >= k0 ? L2 - k0 + 1u : 0u;
    const uint32_t ns_max = nkB ? ((L1 + 7u) & ~7u) + nkB : nkA;
    const uint32_t gr_max = ((L1 + 7u) >> 3) + ((L2 + 7u) >> 3);
    if (ns_max > 64u * U || gr_max > 64u * G) {
      uint32_t cap = P.seq2 ? (64u * U + k0 - 8u) / 2u : 64u * U + k0 - 1u;
      const uint32_t capg = P.seq2 ? 256u * G : 512u * G;
      cap = cap < capg ? cap : capg;
      L1 = L1 < cap ? L1 : cap;
      L2 = L2 < cap ? L2 : cap;
    }
  }
  if (LSUM) {
    // stage the summary: 16 bytes per thread per pass, once per (persistent) workgroup
    const uint4 *src = reinterpret_cast<const uint4 *>(P.lsum32);
    uint4 *dst = reinterpret_cast<uint4 *>(lds);
    for (uint32_t i = threadIdx.x; i < UG::SUM_BITS / 128; i += WAVES * 64) dst[i] = src[i];
    if (DYN && threadIdx.x == 0) *dyn_ctr = (UNI && !CLS && !LX && P.pre_verdict) ? (uint32_t)WAVES : (TRO ? 3u : 2u) * WAVES;   // (a wave's first two turns are its own: wave, WAVES + wave -- TRO: three --; of blocks -- see BM below --, the first)
    if (DYNC && threadIdx.x == 0) *dyn_ctr = (uint32_t)WAVES;   // (its first share)
    __syncthreads();
  } else if (DYN) {
    if (threadIdx.x == 0) *dyn_ctr = (UNI && !CLS && !LX && P.pre_verdict) ? (uint32_t)WAVES : 2u * WAVES;
    __syncthreads();
  }
  const uint32_t *lsum = reinterpret_cast<const uint32_t *>(lds);
  uint64_t *wbase = lds + UG::SUM_WORDS64 + wave * WORDS * AREAS;
  // (TRI: re-pointed to the area of the pair at hand)
  uint32_t *fw = reinterpret_cast<uint32_t *>(wbase);
  uint32_t *rv = fw + code_dwords_for(S);
  uint64_t *vbits = wbase + code_dwords_for(S);
  [[maybe_unused]] uint64_t *const mbits = vbits + vbit_words_for(S);   // (base-by-base form of the anchored extension only) bit p: the read's base at packed position p equals the reference's under the mate's anchor
  constexpr uint32_t rcap = stage_cap_bases(S);

  // CLS: the wave's shares of the list of entries, and within a share the classes it runs through -- one "segment" (entries
  // [ent_first, ent_end) of class cls_j) per pass of this loop; else one pass over the batch
  const uint32_t n_waves = gridDim.x * WAVES;
  const uint32_t share_len = CLS ? (uint32_t)((P.n + CLS_SHARES - 1u) / CLS_SHARES) : 0u;
  uint32_t share = blockIdx.x * WAVES + wave, share_end = 0, cls_j = 0, ent_first = 0, ent_end = 0;
  if (CLS) {
    if (share >= CLS_SHARES || (uint64_t)share * share_len >= P.n) return;
    ent_end = share * share_len;        // (where the first segment starts)
    share_end = (uint64_t)ent_end + share_len < P.n ? ent_end + share_len : (uint32_t)P.n;
    cls_j = (uint32_t)__builtin_amdgcn_readfirstlane((int)P.cls_share_first[share]);
  }
  for (;;) {
  if (CLS) {
    const uint4 cd = P.cls_list[cls_j];       // {l1, l2, first entry, entries}
    L1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)cd.x);
    L2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)cd.y);
    const uint32_t cls_end = (uint32_t)__builtin_amdgcn_readfirstlane((int)(cd.z + cd.w));
    ent_first = ent_end;
    ent_end = cls_end < share_end ? cls_end : share_end;
  }

  // ---- geometry: of every read of the batch (UNI), of the unit's class (CLS) or of the current read ----------
  const uint32_t k = P.k;
  uint32_t nk1, nk2, P2, g2, n_groups;
  uint32_t tail_inv[G], Lm[G], bofs[G];
  bool act[G], m2[G];
  auto set_geometry = [&](const uint32_t l1, const uint32_t l2) {
    nk1 = l1 >= k ? l1 - k + 1 : 0;
    nk2 = l2 >= k ? l2 - k + 1 : 0;
    P2 = (l1 + 7u) & ~7u;
    g2 = P2 >> 3;
    n_groups = g2 + ((l2 + 7u) >> 3);                 // <= 64 G (UNI: checked before this kernel is chosen; else: per read below)
    // lane -> the 8 bases it stages (per group)
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const uint32_t gi = (uint32_t)lane + 64u * g;
      act[g] = gi < n_groups;
      m2[g] = gi >= g2;
      bofs[g] = (m2[g] ? gi - g2 : gi) << 3;
      Lm[g] = m2[g] ? l2 : l1;
      const uint32_t rem = act[g] ? Lm[g] - bofs[g] : 8u;
      tail_inv[g] = rem < 8u ? (0xFFu << rem) & 0xFFu : 0u;   // positions of the group behind the mate's end
    }
  };
  set_geometry(L1, L2);
  // TRI: 16 bases per lane.  Lane -> (pair of the triple, chunk of the pair); a pair has c1 + c2 chunks, mate 2's first one at packed
  // position P2 = 16 c1.  The 8-base arrays above stay unused but for act[0], which lane_valid_bases() reads: a pair's validity
  // bytes are 2 (c1 + c2).
  uint32_t tri_lp = 0, tri_pr = 0, tri_cl = 0, tri_bofs = 0, tri_tail = 0, tri_Lm = 0;
  bool tri_act = false;
  const uint8_t *tri_base = nullptr;
  if (TRI) {
    const uint32_t c1 = (L1 + 15u) >> 4, c2 = (L2 + 15u) >> 4;
    tri_lp = c1 + c2;
    P2 = c1 << 4;
    g2 = P2 >> 3;
    n_groups = 2u * tri_lp;
    act[0] = (uint32_t)lane < n_groups;
    tri_pr = ((uint32_t)lane >= tri_lp ? 1u : 0u) + ((uint32_t)lane >= 2u * tri_lp ? 1u : 0u);
    tri_cl = (uint32_t)lane - tri_pr * tri_lp;
    tri_act = (uint32_t)lane < 3u * tri_lp;
    const bool in2 = tri_cl >= c1;
    tri_bofs = (in2 ? tri_cl - c1 : tri_cl) << 4;
    tri_Lm = in2 ? L2 : L1;
    const uint32_t rem = tri_act ? tri_Lm - tri_bofs : 16u;       // (>= 1: a chunk holds a base of its mate)
    tri_tail = rem < 16u ? (0xFFFFu << rem) & 0xFFFFu : 0u;
    tri_base = (in2 ? P.seq2 : P.seq1) + tri_bofs;
  }
  // (ragged batches) what a read of the batch brings of its own: slots per mate, and which of a lane's bases lie behind its mate's end
  auto set_read = [&](const uint32_t l1, const uint32_t l2) {
    nk1 = l1 >= k ? l1 - k + 1 : 0;
    nk2 = l2 >= k ? l2 - k + 1 : 0;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      Lm[g] = m2[g] ? l2 : l1;
      const uint32_t rem = Lm[g] > bofs[g] ? Lm[g] - bofs[g] : 0u;
      tail_inv[g] = rem < 8u ? (0xFFu << rem) & 0xFFu : 0u;
    }
  };
  // the bound cut (above): rounds [0, cutE) are probed first; cutUb = bases the slots of the other rounds cover.  cutE = U: no cut
  // ubJA = the same for the stop of the early decision (JA_ROUNDS rounds; vote<J> below)
  constexpr int JA_ROUNDS = (ACCEPT && CutPlan<U>::E0 + 1 < U) ? CutPlan<U>::E0 + 1 : U;
  uint32_t cutE = U, cutUb = 0, ubJA = 0, thr_full = 0;
  auto plan_cut = [&](const uint32_t l1, const uint32_t l2) {
    cutE = U;
    cutUb = 0;
    if (JA_ROUNDS < U) ubJA = bases_behind(64u * (uint32_t)JA_ROUNDS, nk1, nk2, P2, l1, l2);
    thr_full = cov_threshold(P.c, l1 + l2);   // len <= l1 + l2: the joiner is not a valid character
    if (!CUT) return;
    // bases_behind falls with e: the smaller candidate is tried last and wins when it qualifies
    if (CutPlan<U>::E1 != CutPlan<U>::E0) {
      const uint32_t ub = bases_behind(64u * (uint32_t)CutPlan<U>::E1, nk1, nk2, P2, l1, l2);
      if (ub < thr_full) { cutE = (uint32_t)CutPlan<U>::E1; cutUb = ub; }
    }
    const uint32_t ub = bases_behind(64u * (uint32_t)CutPlan<U>::E0, nk1, nk2, P2, l1, l2);
    if (ub < thr_full) { cutE = (uint32_t)CutPlan<U>::E0; cutUb = ub; }
  };
  if (UNI) plan_cut(L1, L2);
  // ---- the sparse first round (one-gene index in LDS).  With ONE gene in the index nothing competes: a read is that
  // gene's iff the bases covered by its k-mers that are in the filter reach c * len, and a LOWER bound on that coverage which
  // passes settles it.  So the 128 probes in front of the cut's stop are made in another order: round A = the even slots of
  // the prefix [0, 128 - T) (57 k-mers that tile 129 bases of mate 1, and an error costs a base or two, not k) plus T tiles --
  // disjoint k-mers k apart, counted back from the last slot of the pair, k bases each --; round B = the rest of the prefix.
  // A read from the gene is through behind round A (2 x 150 bp: 248 of the 180 bases needed, 97 % of the on-target pairs at
  // 1 % errors, instead of behind three rounds and a vote); a read without any match behind B is cut as before, the prefix
  // being what the cut needs (bases_behind(128 - T) < c * len fixes T).  Everything else -- a few per cent -- is brought into the
  // usual order and tried again over the whole prefix and the tiles; what is still open probes the T left-over slots and goes on
  // as ever.  spT = 0: not used (several genes, another geometry).
  // (Ragged batches on a one-gene index run this kernel with the exact table too -- see launch_classify_uni -- and plan per read.)
  constexpr bool SPARSE = LX && CUT && ACCEPT && !SHK_NO_SPARSE && U >= 3 && JA_ROUNDS >= 2 && JA_ROUNDS <= 3;
  uint32_t spT = 0, spLast = 0, spUb = 0;
  // (where a settled read's result goes: held across the loop.  Re-read per read like the vote does, the two dependent scalar
  //  loads were 5 % of an on-target launch: 4.41 -> 4.21 ms per 10 M pairs at 50 % on-target.  Fetching the bases two reads ahead
  //  instead of one was measured as well and changed nothing: 4.21 -> 4.29 ms)
  uint32_t *sp_count = nullptr;
  uint16_t *sp_inl = nullptr;
  // sp_one: ONE gene in the index (the argument above).  Several genes (the LXM instantiation): the same two rounds in the same order
  // settle a read too, by the early decision's argument instead -- see sparse_first
  constexpr bool sp_one = !LXM;
  const bool sp_on = SPARSE && (LXM ? P.lx_multi != 0u : P.lx_gene != 0xFFFFFFFFu);
  // floor(x / k) and floor(x / (k - 1)) for x < 2048 as a multiplication (k <= 32)
  const uint32_t sp_rk = sp_on ? (65536u + k - 1u) / k : 0u, sp_rk1 = (sp_on && k > 1u) ? (65536u + k - 2u) / (k - 1u) : 0u;
  // the smallest slot s with bases_behind(s) <= B
  auto slot_for_ub = [&](const uint32_t B, const uint32_t l1, const uint32_t l2) -> uint32_t {
    const uint32_t c1 = nk1 ? l1 : 0u, c2 = nk2 ? l2 : 0u;   // (a mate shorter than k has no slot and covers nothing)
    if (c1 + c2 <= B) return 0u;
    if (nk1 && c1 + c2 - B < nk1) return c1 + c2 - B;
    if (c2 <= B) return nk1;
    return P2 + (c2 - B < nk2 ? c2 - B : nk2);
  };
  auto plan_sparse = [&](const uint32_t l1, const uint32_t l2) {
    spT = 0;
    if (!sp_on || cutE != 2u || !(nk1 | nk2) || !thr_full) return;
    const uint32_t nkl = nk2 ? nk2 : nk1;
    spLast = (nk2 ? P2 : 0u) + nkl - 1u;                       // the last slot of the pair; tile t is the slot spLast - t k
    // the largest T <= 16 whose tiles stay inside the last mate ((T - 1) k < nkl), do not touch what the prefix covers
    // (spLast - (T - 1) k >= (128 - T) + k - 1, i.e. T (k - 1) <= spLast - 127), and whose prefix still carries the cut
    // (bases_behind(128 - T) < c * len, i.e. 128 - T >= the first slot with at most c * len - 1 bases behind it)
    uint32_t T = 16u;
    const uint32_t t_in = (((nkl - 1u) * sp_rk) >> 16) + 1u;
    T = T < t_in ? T : t_in;
    if (k > 1u) {
      if (spLast < 127u) return;
      const uint32_t t_ap = ((spLast - 127u) * sp_rk1) >> 16;
      T = T < t_ap ? T : t_ap;
    }
    const uint32_t X = slot_for_ub(thr_full - 1u, l1, l2);
    if (X > 127u) return;
    T = T < 128u - X ? T : 128u - X;
    if (T == 0u) return;
    // (the conditions themselves, not their closed forms, decide)
    const uint32_t back = (T - 1u) * k;
    const uint32_t ub = bases_behind(128u - T, nk1, nk2, P2, l1, l2);
    if (back < nkl && spLast >= back && spLast - back >= (128u - T) + k - 1u && ub < thr_full) { spT = T; spUb = ub; }
  };
  if (sp_on) {
    // (through readfirstlane: values of their own from here on.  As plain loads the compiler is free to re-do them at every use when
    //  scalar registers run short -- kernarg -> *P.out -> store, a dependent memory round trip and a vmcnt(0) per settled read; some
    //  builds of the three-pairs kernel did)
    const uint64_t pc = reinterpret_cast<uint64_t>(P.out->count), pi = reinterpret_cast<uint64_t>(P.out->inl);
    sp_count = reinterpret_cast<uint32_t *>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(pc >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pc));
    sp_inl = reinterpret_cast<uint16_t *>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(pi >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pi));
    if (UNI) plan_sparse(L1, L2);
  }
  // (TRO) the plan of the layout's own lengths, kept: a pair of those lengths takes it as a pair of a uniform batch would
  struct { uint32_t cutE, spT, cutUb, ubJA, thr_full, spLast, spUb; } plan0{cutE, spT, cutUb, ubJA, thr_full, spLast, spUb};
  // (ragged, fixed layout) where the mates' buffers end: off[n]
  uint64_t end1 = 0, end2 = 0;
  if (FIXLAY) {
    end1 = ((ConstU64)(uintptr_t)P.off1)[P.n];
    end2 = P.seq2 ? ((ConstU64)(uintptr_t)P.off2)[P.n] : ~0ull;
  }
  const uint8_t *sbase[G], *qbase[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    sbase[g] = (m2[g] ? P.seq2 : P.seq1) + bofs[g];
    qbase[g] = HASQ ? (m2[g] ? P.qual2 : P.qual1) + bofs[g] : nullptr;
  }
  // the unguarded loads read up to 11 bytes behind a group's first byte: fine while that stays inside the mate's buffer
  // (TRI: the loop below walks TRIPLES of consecutive reads)
  const uint32_t n_reads = (uint32_t)P.n;
  const uint32_t n32 = CLS ? ent_end : (TRI ? (n_reads + 2u) / 3u : n_reads), stride = CLS ? 1u : gridDim.x * WAVES;
  const uint32_t Lmin = L2 ? (L1 < L2 ? L1 : L2) : L1;
  const uint32_t guard_reads = Lmin >= 12u ? 1u : (Lmin ? (12u + Lmin - 1u) / Lmin : n32);   // trailing reads with guarded loads

  // UNI: read r of a mate is at r * L.  The reads a wave fetches are `stride` apart: in the summary / table modes the lane's place in
  // the next one is a running pointer (one 64-bit add per fetch instead of two quarter-rate 64-bit multiply-adds: 1 000 genes at
  // 0 / 50 / 100 % on-target 9.8 / 12.1 / 15.4 -> 9.4 / 11.7 / 15.2 ms per 10 M pairs); the exact-table kernel keeps the product
  // (with the pointer it measured 4.25 -> 4.40 ms)
  constexpr bool RUNPTR = !LX && !DYN;   // (a wave that takes turns has no fixed step)
  const uint8_t *snext[G], *qnext[G];
  uint32_t r_at = blockIdx.x * WAVES + wave;   // the read the running pointers stand at
  {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const uint64_t o = (uint64_t)r_at * Lm[g];
      snext[g] = sbase[g] + o;
      qnext[g] = HASQ ? qbase[g] + o : nullptr;
    }
  }
  // (moves the running pointers on to read r -- `stride` reads further, or, where reads that have their result are passed over, as
  //  many as it takes -- and fetches it)
  auto issue = [&](const uint32_t r, Raw8 (&w)[G], Raw8 (&q)[G]) {
    const uint32_t r_step = r - r_at;
    if (RUNPTR) r_at = r;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      w[g] = Raw8{0u, 0u, 0u, 0u};
      q[g] = Raw8{0u, 0u, 0u, 0u};
      if (RUNPTR) {
        const uint64_t d = (uint64_t)r_step * Lm[g];
        snext[g] += d;
        if (HASQ) qnext[g] += d;
      }
      if (act[g]) {
        const uint64_t o = RUNPTR ? 0ull : (uint64_t)r * Lm[g];
        const uint8_t *sp = RUNPTR ? snext[g] : sbase[g] + o;
        const uint8_t *qp = HASQ ? (RUNPTR ? qnext[g] : qbase[g] + o) : nullptr;
        if (n32 - r > guard_reads) {
          w[g] = load8_issue_all(sp, 8u);
          if (HASQ) q[g] = load8_issue_all(qp, 8u);
        } else {
          const uint32_t rem = Lm[g] - bofs[g];   // (the last reads of the batch: what is left of the mate decides which dwords exist)
          w[g] = load8_issue(sp, rem);
          if (HASQ) q[g] = load8_issue(qp, rem);
        }
      }
    }
  };
  // (ragged batches) the 8 bases a lane stages of the read at offsets m.o1 / m.o2, by the batch's layout; guarded: only dwords that
  // hold bytes of the mate are touched
  auto fetch_groups = [&](const ReadMeta &m, Raw8 (&w)[G], Raw8 (&q)[G], const bool entry_safe = false) {
    const bool safe = CLS ? entry_safe : (m.o1 + L1 + 16u <= end1 && m.o2 + L2 + 16u <= end2);
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (!FIXLAY && !CLS) {   // the read's own layout (table modes)
        fetch_group<HASQ>(P, m, (uint32_t)lane + 64u * g, w[g], q[g]);
        continue;
      }
      w[g] = Raw8{0u, 0u, 0u, 0u};
      q[g] = Raw8{0u, 0u, 0u, 0u};
      const uint32_t Lr = m2[g] ? m.L2 : m.L1;
      const uint64_t o = m2[g] ? m.o2 : m.o1;
      // (wave-uniform) every group of the layout, and the 11 bytes behind its first, lie inside the mates' buffers: three unconditional
      // aligned dwords per group, as for uniform batches (bytes behind the read's own end are its neighbour's, masked by tail_inv);
      // only the last reads of the batch take the guarded loads
      if (safe) {
        if (act[g]) {
          w[g] = load8_issue_all(sbase[g] + o, 8u);
          if (HASQ) q[g] = load8_issue_all(qbase[g] + o, 8u);
        }
      } else if (act[g] && bofs[g] < Lr) {
        w[g] = load8_issue(sbase[g] + o, Lr - bofs[g]);
        if (HASQ) q[g] = load8_issue(qbase[g] + o, Lr - bofs[g]);
      }
    }
  };
  auto retire = [&](Raw8 (&w)[G], Raw8 (&q)[G]) {
#pragma unroll
    for (int g = 0; g < G; ++g) { retire_loads(w[g]); retire_loads(q[g]); }
  };
  // valid characters among the bases this lane staged (their sum over the wave is the read's len, ReadAnalyzer.hpp:46-49)
  auto lane_valid_bases = [&]() -> uint32_t {
    uint32_t c = 0;
#pragma unroll
    for (int g = 0; g < G; ++g) c += act[g] ? (uint32_t)__builtin_popcount((uint32_t)reinterpret_cast<const uint8_t *>(vbits)[(uint32_t)lane + 64u * g]) : 0u;
    return c;
  };

  // (TRI) the 16 bases lane (pair tri_pr, chunk tri_cl) stages of triple t: five aligned dwords.  An unguarded fetch reads up to 19
  // bytes from the chunk's first -- into the reads behind it --, so the last reads of the batch take the guarded form (only dwords
  // that hold a byte of the mate)
  struct Raw16 { uint32_t d0, d1, d2, d3, d4, sh; };
  const uint32_t tri_guard = Lmin ? (19u + Lmin - 1u) / Lmin : 0u;
  // (TRO) the lane's mate's offsets of the reads of triple t, asked for a third of a pass before its bases; where that mate's buffer ends
  struct TriOff { uint64_t a, b; };
  const uint64_t *tri_off = nullptr;
  uint64_t tri_end = 0ull;
  if (TRO) {
    tri_off = (tri_cl >= ((L1 + 15u) >> 4)) ? P.off2 : P.off1;
    tri_end = tri_act ? tri_off[n_reads] : 0ull;
  }
#ifndef SHK_TRO_ABL
#define SHK_TRO_ABL 0      // (timing-only ablations of the offsets path on batches of ONE length, SHK_FORCE_TRO=1: 1 offsets by arithmetic, 2 no per-pair bounds, 4 the layout's tail)
#endif
#if SHK_TRO_ABL != 0 && !defined(SHK_TIMING_ONLY)
#error "-DSHK_TRO_ABL builds a library whose results are WRONG on trimmed batches (timing-only ablation): say so with -DSHK_TIMING_ONLY as well"
#endif
  auto tri_off_issue = [&](const uint32_t t) -> TriOff {
    TriOff o{0ull, 0ull};
    const uint32_t rd = 3u * t + tri_pr;
    if (SHK_TRO_ABL & 1) {
      if (tri_act && rd < n_reads) { o.a = (uint64_t)rd * tri_Lm; o.b = o.a + tri_Lm; }
      return o;
    }
    if (tri_act && rd < n_reads) { o.a = tri_off[rd]; o.b = tri_off[rd + 1u]; }
    return o;
  };
  // (TRO) the 16 bases of the lane's chunk of the read at offsets o; lr: the read's own length of that mate (what lies behind it is the
  // next read's: fetched like the read's own bases where the buffer allows, masked when staged)
  auto tri_issue_at = [&](const TriOff &o, uint32_t &lr) -> Raw16 {
    Raw16 r{0u, 0u, 0u, 0u, 0u, 0u};
    const uint64_t d = o.b - o.a;
    lr = d < (uint64_t)tri_Lm ? (uint32_t)d : tri_Lm;
    if (tri_act && tri_bofs < lr) {
      const uint8_t *sp = tri_base + o.a;
      const uint32_t sh = (uint32_t)reinterpret_cast<uintptr_t>(sp) & 3u;
      const uint32_t *q = reinterpret_cast<const uint32_t *>(sp - sh);
      r.sh = sh;
      if (o.a + tri_bofs + 20ull <= tri_end) {
        r.d0 = q[0]; r.d1 = q[1]; r.d2 = q[2]; r.d3 = q[3]; r.d4 = q[4];
      } else {
        const uint32_t nb = lr - tri_bofs < 16u ? lr - tri_bofs : 16u;
        const uint32_t last = sh + nb - 1u;
        r.d0 = q[0];
        r.d1 = last >= 4u ? q[1] : 0u;
        r.d2 = last >= 8u ? q[2] : 0u;
        r.d3 = last >= 12u ? q[3] : 0u;
        r.d4 = last >= 16u ? q[4] : 0u;
      }
    }
    return r;
  };
  uint32_t tlr_cur = 0u;      // (TRO) the current triple's read's own length of the lane's mate
  auto tri_issue = [&](const uint32_t t) -> Raw16 {
    Raw16 r{0u, 0u, 0u, 0u, 0u, 0u};
    const uint32_t rd = 3u * t + tri_pr;
    if (tri_act && rd < n_reads) {
      const uint8_t *sp = tri_base + (uint64_t)rd * tri_Lm;
      const uint32_t sh = (uint32_t)reinterpret_cast<uintptr_t>(sp) & 3u;
      const uint32_t *q = reinterpret_cast<const uint32_t *>(sp - sh);
      r.sh = sh;
      if (n_reads - rd > tri_guard) {
        r.d0 = q[0]; r.d1 = q[1]; r.d2 = q[2]; r.d3 = q[3]; r.d4 = q[4];
      } else {
        const uint32_t nb = tri_Lm - tri_bofs < 16u ? tri_Lm - tri_bofs : 16u;
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
  auto tri_retire = [&](Raw16 &r) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(r.d0), "+v"(r.d1), "+v"(r.d2), "+v"(r.d3), "+v"(r.d4)); };
  Raw16 t_cur{0u, 0u, 0u, 0u, 0u, 0u};
  TriOff to_cur{0ull, 0ull};      // (TRO) the offsets of the NEXT triple, asked for a pass before its bases

  // DYN: the position of the workgroup's turn t (n32: behind the batch's end; the turns behind such a turn are too)
  auto dyn_pos = [&](const uint32_t t) -> uint32_t {
    const uint64_t v = (uint64_t)(t / (uint32_t)WAVES) * stride + (blockIdx.x * WAVES + t % (uint32_t)WAVES);
    return v < n32 ? (uint32_t)v : n32;
  };
  uint32_t it = CLS ? ent_first : blockIdx.x * WAVES + wave;   // position in the batch (CLS: in the list of entries; TRI: the triple)
  if (!CLS && it >= n32) return;
  uint32_t read = it;                                          // the read's index in the batch: where its result goes
  Raw8 w_cur[G], q_cur[G];
  ReadMeta m_cur{}, m_nxt{};
  // CLS: 64 entries of the segment at a time, one per lane in six registers; a read's own offsets by v_readlane where its bases are
  // fetched -- one load per 64 reads, and no scalars of two reads held across the loop body
  uint4 ent_a = make_uint4(0u, 0u, 0u, 0u);
  uint2 ent_b = make_uint2(0u, 0u);
  uint32_t read_nxt = 0;
  // (fetches the bases of entry e, leaves its read's index in read_nxt)
  auto entry_fetch = [&](const uint32_t e, Raw8 (&w)[G], Raw8 (&q)[G]) {
    const int l = (int)((e - ent_first) & 63u);
    if (l == 0) {
      const uint32_t mine = e + (uint32_t)lane;
      if (mine < ent_end) {
        ent_a = P.cls_entries[2ull * mine];
        ent_b = *reinterpret_cast<const uint2 *>(P.cls_entries + 2ull * mine + 1ull);
      }
    }
    ReadMeta m;
    m.o1 = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)ent_a.y, l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)ent_a.x, l);
    m.o2 = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)ent_a.w, l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)ent_a.z, l);
    m.L1 = L1;
    m.L2 = L2;
    read_nxt = (uint32_t)__builtin_amdgcn_readlane((int)ent_b.x, l);
    fetch_groups(m, w, q, __builtin_amdgcn_readlane((int)ent_b.y, l) != 0);
  };
  // (ragged batches) a read's plan -- which rounds first, the bounds behind them, the sparse order -- depends on its two lengths alone.
  // Computing it per read is about 150 scalar instructions, and the CU's one scalar unit serves all its waves (measured: 7.9 ms per
  // 10 M trimmed pairs against 4.2 ms untrimmed, nearly all of it scalar); so the launch carries a table indexed by (l1, l2), cleared
  // by the host: the first wave to meet a pair of lengths computes the plan and leaves it there, everybody else loads 16 bytes --
  // one read ahead, with the bases.  A stale or missing entry only means computing the (same) plan again.
  constexpr uint32_t PLAN_VALID = 0x504C414Eu;
  // (with the fixed layout only: elsewhere the eight registers of two plans in flight made the quality-mask instantiations spill)
  const bool has_plans = FIXLAY && P.plan_tab != nullptr && (uint64_t)(L1 + 1u) * (L2 + 1u) <= (uint64_t)P.plan_cap;
  auto plan_index = [&](const ReadMeta &m) -> uint32_t { return (m.L1 <= L1 && m.L2 <= L2) ? m.L1 * (L2 + 1u) + m.L2 : 0u; };
  auto plan_issue = [&](const ReadMeta &m) -> uint4 { return has_plans ? P.plan_tab[plan_index(m)] : make_uint4(0u, 0u, 0u, 0u); };
  uint4 pl_cur = make_uint4(0u, 0u, 0u, 0u);
  // PRE (uniform batches beyond the exact-table instantiations): anchor_verdict_kernel may have run in front of this launch
  // (P.pre_verdict): a read whose count[] is set has its result already.  The batch is then walked in BLOCKS of 64 consecutive reads --
  // a wave's sequence of positions (fixed steps, or turns) is a sequence of blocks --: one coalesced load brings a block's 64 flags,
  // fetched a block ahead, a ballot says which of its reads are left, and only those are ever fetched, staged or waited for.
  // (Passing over a settled read inside the usual loop cost its prefetch's round trip: 1.2 ms per 10 M pairs at 100 % on-target.)
  constexpr bool PRE = UNI && !CLS && !LX;
  // PRE_R: the ragged instantiations of the table modes behind that kernel (trimmed batches).  Their loop fetches a read's offsets two
  // reads ahead and its bases one ahead at fixed steps: a settled read is passed over by its flag, fetched with the next read's bases
  // (the read's own prefetch is paid -- these kernels wait on memory at 12-17 ms per 10 M pairs; the block walk is the uniform loop's)
  constexpr bool PRE_R = !UNI && !pm_lds(MODE);
  const uint32_t *pre_count = nullptr;
  if ((PRE || PRE_R) && P.pre_verdict) pre_count = P.out->count;
  const bool BM = PRE && pre_count != nullptr;
  uint32_t done_cur = 0u;
  auto pre_fetch = [&](const uint32_t r) -> uint32_t {
    const uint32_t *cp = pre_count + r;
    asm volatile("" : "+v"(cp));   // (a vector load: a scalar one would share lgkmcnt with the LDS accesses of the read at hand)
    return *cp;
  };
  const uint32_t n_blk = (n_reads + 63u) >> 6;
  uint32_t bm_pos = 0u, bm_pn = 0u, bm_fn = 1u;   // the block at hand, the next one of the wave's sequence and (per lane) its flags
  uint64_t bm_bits = 0ull;                          // reads of the block at hand that are left behind the current one
  // the position behind `cur` in the wave's sequence of blocks (n_blk: none)
  auto bm_next_pos = [&](const uint32_t cur) -> uint32_t {
    if (DYN) {
      uint32_t t = 0u;
      if (lane == 0) t = atomicAdd(dyn_ctr, 1u);
      t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
      const uint64_t v = (uint64_t)(t / (uint32_t)WAVES) * stride + (blockIdx.x * WAVES + t % (uint32_t)WAVES);
      return v < n_blk ? (uint32_t)v : n_blk;
    }
    return n_blk - cur > stride ? cur + stride : n_blk;
  };
  // (per lane) 0: read 64 pos + lane is left to do
  auto bm_flags = [&](const uint32_t pos) -> uint32_t {
    const uint32_t r = (pos << 6) + (uint32_t)lane;
    return (pos < n_blk && r < n_reads) ? pre_count[r] : 1u;
  };
  if (BM) {
    bm_pos = it;                                     // (the wave's first position, as a block)
    if (bm_pos >= n_blk) return;
    uint64_t m = __ballot(bm_flags(bm_pos) == 0u);
    bm_pn = bm_next_pos(bm_pos);
    bm_fn = bm_flags(bm_pn);
    while (m == 0ull) {
      if (bm_pn >= n_blk) return;
      bm_pos = bm_pn;
      m = __ballot(bm_fn == 0u);
      bm_pn = bm_next_pos(bm_pos);
      bm_fn = bm_flags(bm_pn);
    }
    it = (bm_pos << 6) + (uint32_t)__builtin_ctzll(m);
    bm_bits = m & (m - 1ull);
    read = it;
  }
  if (CLS) {
    entry_fetch(it, w_cur, q_cur);
    read = read_nxt;
  } else if (TRI) {
    if (TRO) {
      const TriOff o0 = tri_off_issue(it);
      t_cur = tri_issue_at(o0, tlr_cur);
      const uint32_t p1 = DYN ? dyn_pos((uint32_t)WAVES + wave) : (n32 - it > stride ? it + stride : n32);
      to_cur = tri_off_issue(p1 < n32 ? p1 : n32);      // (behind the batch's end: no loads)
    } else {
      t_cur = tri_issue(it);
    }
    tri_retire(t_cur);
  } else if (UNI) {
    issue(read, w_cur, q_cur);
  } else {
    m_cur = fetch_meta(P, read);
    fetch_groups(m_cur, w_cur, q_cur);
    if (PRE_R && pre_count) done_cur = pre_fetch(read);
    pl_cur = plan_issue(m_cur);
    const uint32_t n1 = n32 - read > stride ? read + stride : n32;
    m_nxt = fetch_meta(P, n1 < n32 ? n1 : read);
  }
  retire(w_cur, q_cur);
  const uint64_t kmer_mask = (1ull << (2u * k)) - 1ull;
#if SHK_STAMPS
  // phases: 0 loop head (next triple's loads issued) | 1 staging | 2 a pair's set-up | 3 round A's probe | 4 A's validation, coverage,
  // verdict | 5 round B's probe | 6 everything else of a pair | 7 loop tail | 8 (count) triples | 9 (count) pairs
  uint64_t st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  uint64_t st_last = __builtin_readcyclecounter();
  // 13 / 14: the wave's whole loop in shader cycles and in ticks of the constant 100 MHz counter -- their quotient is the clock the
  // chip held during the launch
  const uint64_t st_t0 = st_last, st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  // DYN: turn q of the workgroup is position blockIdx.x * WAVES + q % WAVES + (q / WAVES) * stride of the batch -- the positions the
  // workgroup's waves walk together in the fixed order, whoever takes them
  uint32_t dyn_q = (uint32_t)WAVES + wave;
  // (TRO draws its turns a pass earlier: a triple's OFFSETS are asked for a pass before its bases, so the turn after next has to be known)
  uint32_t dyn_q2 = 2u * (uint32_t)WAVES + wave;
  // TF (three pairs per pass, one-gene index; P.tile_first: the host's reading of the stream): THE TILES' ROUND.  A pair is the gene's
  // as soon as k-mers of it that are in the filter are seen to cover c * len bases (sparse_first: "a lower bound that passes settles
  // a read") -- and DISJOINT k-mers cover k bases each, so up to 8 per mate, 16 per pair, one lane each, say so for THREE pairs in ONE
  // hash round: 2 x 150 bp, k = 17, c = 0.6 needs 11 of its 16 tiles (187 >= 180 bases), which a pair from the gene with 1 % errors has
  // 97 times in 100.  Such a pair ends here, behind a third of a round instead of a whole one and without any of its own set-up; a
  // pair with fewer matching tiles (more errors; from elsewhere) goes through the usual rounds as if nothing had happened -- having
  // paid a third of a round for it, which is why the host asks for this only behind a batch with many reads assigned (a quarter: where
  // the two kernels were measured to meet).  Pairs with invalid characters take part: a tile counts when its k characters are valid.
  // (An instantiation of its own, TFK: as a run-time switch of the three-pairs kernel its per-lane constants and masks cost that
  //  kernel a fifth of its speed with the round switched OFF -- 3.18 -> 3.82 ms per 10 M pairs from elsewhere; 110 -> 173 spilled scalars.)
  constexpr bool TF = TFK && TRI && SPARSE && !LXM && !SHK_NO_TILE_FIRST;
  bool tf_on = false, tf_want = false;
  uint32_t tf_slot = 0u, tf_area = 0u;
  if (TF && spT != 0u && cutE == 2u && thr_full != 0u) {
    const uint32_t m1 = nk1 ? ((nk1 - 1u) / k + 1u < 8u ? (nk1 - 1u) / k + 1u : 8u) : 0u;   // disjoint k-mers of a mate: slots 0, k, 2k, ...
    const uint32_t m2 = nk2 ? ((nk2 - 1u) / k + 1u < 8u ? (nk2 - 1u) / k + 1u : 8u) : 0u;
    if ((m1 + m2) * k >= thr_full) {
      tf_on = true;
      const uint32_t j = (uint32_t)lane & 15u;
      if (lane < 48) {
        tf_area = (uint32_t)lane >> 4;
        tf_want = j < 8u ? j < m1 : j - 8u < m2;
        tf_slot = tf_want ? (j < 8u ? j * k : P2 + (j - 8u) * k) : 0u;
      }
    }
  }
  for (;;) {
    uint32_t nxt;
    if (BM) {
      // the next read that is left: of the block at hand, else of the next block of the wave's sequence that has one
      nxt = n32;
      if (bm_bits) {
        nxt = (bm_pos << 6) + (uint32_t)__builtin_ctzll(bm_bits);
        bm_bits &= bm_bits - 1ull;
      } else {
        while (bm_pn < n_blk) {
          const uint64_t m = __ballot(bm_fn == 0u);
          bm_pos = bm_pn;
          bm_pn = bm_next_pos(bm_pos);
          bm_fn = bm_flags(bm_pn);
          if (m) {
            nxt = (bm_pos << 6) + (uint32_t)__builtin_ctzll(m);
            bm_bits = m & (m - 1ull);
            break;
          }
        }
      }
    } else if (DYN) nxt = dyn_pos(dyn_q);
    else nxt = n32 - it > stride ? it + stride : n32;   // saturates at n32
    const bool have_nxt = nxt < n32;
    Raw8 w_nxt[G], q_nxt[G];
#pragma unroll
    for (int g = 0; g < G; ++g) { w_nxt[g] = Raw8{0u, 0u, 0u, 0u}; q_nxt[g] = Raw8{0u, 0u, 0u, 0u}; }
    ReadMetaRaw r_nn{};
    uint32_t nn = n32;
    uint32_t done_nxt = 0u;
    uint4 pl_nxt = make_uint4(0u, 0u, 0u, 0u);
    Raw16 t_nxt{0u, 0u, 0u, 0u, 0u, 0u};
    TriOff to_nxt{0ull, 0ull};
    uint32_t tlr_nxt = 0u;
    if (CLS) {
      if (have_nxt) entry_fetch(nxt, w_nxt, q_nxt);
    } else if (TRI) {
      if (TRO) {
        // the next triple's bases (its offsets were asked for a pass ago) and the offsets of the triple behind it
        if (have_nxt) t_nxt = tri_issue_at(to_cur, tlr_nxt);
        const uint32_t nn2 = DYN ? dyn_pos(dyn_q2) : ((have_nxt && n32 - nxt > stride) ? nxt + stride : n32);
        to_nxt = tri_off_issue(nn2 < n32 ? nn2 : n32);
      } else if (have_nxt) t_nxt = tri_issue(nxt);
    } else if (UNI) {
      if (have_nxt) issue(nxt, w_nxt, q_nxt);
    } else {
      if (have_nxt) { fetch_groups(m_nxt, w_nxt, q_nxt); pl_nxt = plan_issue(m_nxt); }
      if (PRE_R && pre_count && have_nxt) done_nxt = pre_fetch(nxt);
      nn = (have_nxt && n32 - nxt > stride) ? nxt + stride : n32;
      r_nn = fetch_meta_issue(P, nn < n32 ? nn : read);          // clamped index
      if (FIXLAY) set_read(m_cur.L1, m_cur.L2); else set_geometry(m_cur.L1, m_cur.L2);
    }
    // (DYN) the turn after the next: asked for here, behind the next one's loads, wanted at the end of this pass
    uint32_t dyn_take = 0u;
    if (DYN && !BM) {
      if (lane == 0) dyn_take = atomicAdd(dyn_ctr, 1u);
    }
    SHK_STAMP(0);
    bool skip = false;
    if (!UNI) {
      const uint32_t ns = nk2 ? ((m_cur.L1 + 7u) & ~7u) + nk2 : nk1;
      // longer than the batch's layout (FIXLAY) / this specialisation holds: the general kernel's queue (as process_read does)
      if (FIXLAY ? (m_cur.L1 > L1 || m_cur.L2 > L2) : (ns > S || n_groups > 64u * G)) {
        if (lane == 0) {
          const ClassifyOut *O = out_ptrs(P);
          const uint32_t qi = atomicAdd(&O->counters[CTR_LONG], 1u);
          O->long_queue[qi] = read;
          atomicMax(&O->counters[CTR_MAX_SLOTS], ns);
        }
        skip = true;
      }
    }
    // (behind the check above: a read this specialisation does not hold is queued for the general kernel whether or not it has its
    //  result already -- the host may have COUNTED the queue's entries (LONG_KNOWN), and that kernel writes the same result again)
    if (PRE_R && __builtin_amdgcn_readfirstlane((int)done_cur) != 0) skip = true;
    if (!skip) {

    // (TRO) which pairs of the triple have a mate shorter than the layout's: bit p.  A pair of the layout's own lengths -- most pairs of
    // a trimmed sample -- is a pair of a uniform batch: the wave's plan, nothing fetched, nothing recomputed
    uint32_t tro_short = 0u;
    if constexpr (TRO) {
      const uint64_t sm = (SHK_TRO_ABL & 2) ? 0ull : __ballot(tri_act && tlr_cur != tri_Lm);
      const uint64_t pm = (1ull << tri_lp) - 1ull;
#pragma unroll
      for (uint32_t p3 = 0; p3 < 3u; ++p3) tro_short |= ((sm >> (p3 * tri_lp)) & pm) ? (1u << p3) : 0u;
    }
    // (the two lengths of pair p3: a mate's first chunk lane knows its read's)
    auto tro_len = [&](const uint32_t p3, uint32_t &a, uint32_t &b) {
      a = (uint32_t)__builtin_amdgcn_readlane((int)tlr_cur, (int)(p3 * tri_lp));
      b = L2 ? (uint32_t)__builtin_amdgcn_readlane((int)tlr_cur, (int)(p3 * tri_lp + ((L1 + 15u) >> 4))) : 0u;
    };
    // (TRO) the thresholds of the triple's pairs, lane p's = pair p's: ONE pass through the double arithmetic of cov_threshold per triple
    // that holds a short pair (per pair and use -- the tiles' round, the bounds -- it was a seventh of a short pair's cycles)
    uint32_t tro_thr = 0u;
    if constexpr (TRO) {
      tro_thr = plan0.thr_full;
      if (tro_short) {
        uint32_t a0, b0, a1, b1, a2, b2;
        tro_len(0u, a0, b0); tro_len(1u, a1, b1); tro_len(2u, a2, b2);
        const uint32_t lenv = lane == 0 ? a0 + b0 : (lane == 1 ? a1 + b1 : a2 + b2);
        tro_thr = cov_threshold(P.c, lenv);
      }
    }
    // ---- stage: 8 bases per lane -> the two code streams + validity (see process_read) ----
    uint32_t inv_real = 0u;   // invalid characters among the lane's bases that belong to the read
    if (TRI) {
      // sixteen bases of pair tri_pr of the triple into that pair's area: the same three streams, a dword / a dword / 16 bits per chunk
      if (tri_act && 3u * it + tri_pr < n_reads) {
        const uint32_t sh = t_cur.sh;
        const uint32_t b0 = __builtin_amdgcn_alignbyte(t_cur.d1, t_cur.d0, sh), b1 = __builtin_amdgcn_alignbyte(t_cur.d2, t_cur.d1, sh);
        const uint32_t b2 = __builtin_amdgcn_alignbyte(t_cur.d3, t_cur.d2, sh), b3 = __builtin_amdgcn_alignbyte(t_cur.d4, t_cur.d3, sh);
        uint32_t c0, c1, c2, c3, i0, i1, i2, i3;
        classify4(b0, c0, i0);
        classify4(b1, c1, i1);
        classify4(b2, c2, i2);
        classify4(b3, c3, i3);
        const uint32_t msb32 = (pack4(c0) << 24) | (pack4(c1) << 16) | (pack4(c2) << 8) | pack4(c3);      // first base in bits 31:30
        uint32_t tail16 = tri_tail;
        if (TRO && !(SHK_TRO_ABL & 4)) {   // what lies behind the read's own end is invalid (the layout is the longest mates')
          const uint32_t rem = tlr_cur > tri_bofs ? tlr_cur - tri_bofs : 0u;
          tail16 = rem < 16u ? (0xFFFFu << rem) & 0xFFFFu : 0u;
        }
        const uint32_t inv16 = gather4(i0) | (gather4(i1) << 4) | (gather4(i2) << 8) | (gather4(i3) << 12) | tail16;
        uint32_t lsb32 = __builtin_bitreverse32(msb32);
        lsb32 = ((lsb32 >> 1) & 0x55555555u) | ((lsb32 & 0x55555555u) << 1);
        uint64_t *area = wbase + tri_pr * WORDS;
        uint32_t *fwa = reinterpret_cast<uint32_t *>(area);
        fwa[tri_cl] = lsb32;
        (fwa + code_dwords_for(S))[(rcap >> 4) - 1u - tri_cl] = msb32;
        reinterpret_cast<uint16_t *>(area + code_dwords_for(S))[tri_cl] = (uint16_t)(~inv16 & 0xFFFFu);
        inv_real = inv16 & ~tail16;   // (TRO: a short mate is not an invalid character -- its pair's slots end where it ends: nk1e / nk2e below)
      }
    }
#pragma unroll
    for (int g = 0; g < (TRI ? 0 : G); ++g) {
      if (act[g]) {
        const uint32_t gi = (uint32_t)lane + 64u * g;
        const uint32_t sh = w_cur[g].shn & 3u;
        const uint32_t lo = __builtin_amdgcn_alignbyte(w_cur[g].d1, w_cur[g].d0, sh);
        const uint32_t hi = __builtin_amdgcn_alignbyte(w_cur[g].d2, w_cur[g].d1, sh);
        uint32_t c_lo, c_hi, i_lo, i_hi;
        classify4(lo, c_lo, i_lo);
        classify4(hi, c_hi, i_hi);
        const uint32_t msb16 = (pack4(c_lo) << 8) | pack4(c_hi);           // first base in bits 15:14
        // (bytes behind the mate's end are whatever follows in the buffer: their codes land at packed positions that no
        // existing slot's window covers, and tail_inv marks them invalid)
        uint32_t inv8 = gather4(i_lo) | (gather4(i_hi) << 4) | tail_inv[g];
        if (HASQ) {
          const uint32_t qs = q_cur[g].shn & 3u;
          const uint32_t qlo = __builtin_amdgcn_alignbyte(q_cur[g].d1, q_cur[g].d0, qs);
          const uint32_t qhi = __builtin_amdgcn_alignbyte(q_cur[g].d2, q_cur[g].d1, qs);
          inv8 |= gather4(qmask4(qlo, P.mq)) | (gather4(qmask4(qhi, P.mq)) << 4);
        }
        uint32_t lsb = __builtin_bitreverse32(msb16);                      // lands in the high half
        lsb = ((lsb >> 1) & 0x55555555u) | ((lsb & 0x55555555u) << 1);
        reinterpret_cast<uint16_t *>(fw)[gi] = (uint16_t)(lsb >> 16);
        reinterpret_cast<uint16_t *>(rv)[(rcap >> 3) - 1u - gi] = (uint16_t)msb16;
        reinterpret_cast<uint8_t *>(vbits)[gi] = (uint8_t)(~inv8 & 0xFFu);
        inv_real |= inv8 & ~tail_inv[g];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    SHK_STAMP(1);
#if SHK_STAMPS
    if constexpr (TRI) st_acc[8] += 1;
#endif

    // (TRI: the three pairs of the triple one after the other, each in its own staging area)
    // (a generic lambda called once per pair, not a loop: with the pair's area a compile-time offset from the wave's, the LDS
    //  addresses of a pair's windows stay what they are for one area -- the lane's offset from a loop-invariant base, the area as the
    //  instruction's immediate -- instead of an addition per address and pair)
    // the slot ss of a staged pair (its two code streams at fwp / rvp) as a probe of the exact LDS table (LX): is its k-mer in the
    // filter?  payload = the gene of the matched entry's single-gene list, or the escape value.  (Not validated: the caller looks
    // at the slot when something matched.)
    auto lx_probe = [&](const uint32_t *fwp, const uint32_t *rvp, const uint32_t ss, uint32_t &payload) -> bool {
      const uint32_t q = rcap - k - ss;
      const uint32_t *f = fwp + (ss >> 4);
      const uint32_t *r = rvp + (q >> 4);
      const uint32_t d0 = f[0], d1 = f[1], d2 = f[2];
      const uint32_t e0 = r[0], e1 = r[1], e2 = r[2];
      const uint32_t af = (ss & 15u) << 1, ar = (q & 15u) << 1;
      const uint64_t x = ((uint64_t)__builtin_amdgcn_alignbit(d2, d1, af) << 32) | __builtin_amdgcn_alignbit(d1, d0, af);
      const uint64_t y = ((uint64_t)__builtin_amdgcn_alignbit(e2, e1, ar) << 32) | __builtin_amdgcn_alignbit(e1, e0, ar);
      const uint64_t fwd = y & kmer_mask, rc = ~x & kmer_mask;
      const uint64_t h = xxh64_u64(fwd < rc ? fwd : rc);
      const uint32_t *T = lsum;
      const char *D = reinterpret_cast<const char *>(lsum + LTAB_T_WORDS);
      const uint32_t tagmask15 = (uint32_t)(P.bf_mask >> LTAB_SLOT_LG);
      const uint32_t gmask2 = (tagmask15 & ((1u << LTAB_GROUP_LG) - 1u)) << 1;
      const uint32_t dd = *reinterpret_cast<const uint16_t *>(D + (((uint32_t)h >> (LTAB_SLOT_LG - 1)) & gmask2));
      const uint32_t tg = __builtin_amdgcn_alignbit((uint32_t)(h >> 32), (uint32_t)h, LTAB_SLOT_LG) & tagmask15;
      const uint32_t base = (uint32_t)h + (tg >> LTAB_GROUP_LG) * P.lsum_shift;
      const uint32_t ee = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(T) + (((base + dd) << 2) & ((LTAB_T_WORDS - 1u) << 2)));
      payload = ee & LTAB_ESC;
      return (ee >> 13) == ((tg << 1) | 1u);
    };
    // (TRI) does pair p of the triple hold an invalid character (N)?  -- its threshold is then lower than the plan's, and its slots need
    // their validity windows: per_pair's business
    bool inv3[3] = {false, false, false};
    if (TRI) {
#pragma unroll
      for (uint32_t p3 = 0; p3 < 3u; ++p3) inv3[p3] = __ballot((inv_real != 0u) & (tri_pr == p3)) != 0ull;
    }
    // TF: the tiles' round (see above) -- bit p of tf_done: pair p of the triple is settled
    uint32_t tf_done = 0u;
    if constexpr (TF) {
      if (tf_on) {
        uint32_t pay;
        const uint32_t *fwp = reinterpret_cast<const uint32_t *>(wbase + tf_area * WORDS);
        bool hit = tf_want & lx_probe(fwp, fwp + code_dwords_for(S), tf_slot, pay);
        if (inv3[0] | inv3[1] | inv3[2] | (TRO && tro_short != 0u)) {   // (TRO: a tile behind a short mate's end is no k-mer of the pair; the staging marked those bases invalid)
          // a pair with invalid characters: a tile counts when its k characters are valid (slot_valid's window); the pair's threshold
          // is at most the plan's (fewer valid bases), so the plan's is the safe one to pass
          const uint64_t *vb = reinterpret_cast<const uint64_t *>(fwp) + code_dwords_for(S);   // (the area's validity words behind its two code streams)
          const uint32_t V = tf_slot >> 6, vs = tf_slot & 63u;
          const uint64_t v0 = vb[V], v1 = vb[V + 1];
          const uint64_t win = (v0 >> vs) | ((v1 << 1) << (63u - vs)), km = (1ull << k) - 1ull;
          hit = hit & ((win & km) == km);
        }
        const uint64_t Hb = __ballot(hit);
#pragma unroll
        for (uint32_t p3 = 0; p3 < 3u; ++p3) {
          const uint32_t cnt = (uint32_t)__builtin_popcount((uint32_t)(Hb >> (16u * p3)) & 0xFFFFu);
          const uint32_t rd = 3u * it + p3;
          // (said to be uniform in so many words: taken for a per-lane value, tf_done lived in a vector register and every pair of the
          //  triple behind an exec-mask branch)
          // (TRO: the pair's own threshold -- of its two lengths; a pair with N has a lower one still: passing this one is sufficient)
          const uint32_t thr_p = (TRO && !(SHK_TRO_ABL & 2)) ? (uint32_t)__builtin_amdgcn_readlane((int)tro_thr, (int)p3) : thr_full;      // (TRO: thr_full is the last pair's)
          if (__builtin_amdgcn_readfirstlane((int)((!TRO || thr_p != 0u) && cnt * k >= thr_p && rd < n_reads))) {
            if (lane == 0 && !SHK_ABL(P, 64u)) {
              sp_count[rd] = 1u;
              uint2 pk;
              pk.x = P.lx_gene & 0xFFFFu;
              pk.y = 0u;
              *reinterpret_cast<uint2 *>(sp_inl + (uint64_t)rd * SHK_INLINE_IDS) = pk;
            }
            tf_done |= 1u << p3;
          }
        }
      }
    }
    auto per_pair = [&](auto tp_const) -> void {
    constexpr uint32_t tp = decltype(tp_const)::value;
    if (TF && (((uint32_t)__builtin_amdgcn_readfirstlane((int)tf_done) >> tp) & 1u)) return;
    if (TRI) {
      read = 3u * it + tp;
      if (read >= n_reads) return;
#if SHK_STAMPS
      st_acc[9] += 1;
#endif
      uint64_t *area = wbase + tp * WORDS;
      fw = reinterpret_cast<uint32_t *>(area);
      rv = fw + code_dwords_for(S);
      vbits = area + code_dwords_for(S);
    }
    SHK_STAMP(10);
    // ---- the bound cut: which rounds are probed first, and may the read end behind them? --------
    uint32_t nk1e = nk1, nk2e = nk2;   // (TRO: the slots of THIS pair's two mates; else the batch's)
    if constexpr (TRO && !(SHK_TRO_ABL & 2)) {
      // The plan's STRUCTURE -- which rounds first, where the tiles sit, which slots a lane holds -- stays the layout's for every pair
      // (it is what the compiler hoists out of this loop; a plan per pair, tried, cost every pair of the batch a quarter more).  Its
      // BOUNDS are the pair's own: what the slots behind a stop cover is a function of the pair's two lengths, and a short pair's
      // threshold is lower -- with the layout's 172 bases behind the first two rounds no pair of 2 x 125 bp (c len = 150) would ever be cut
      cutUb = plan0.cutUb; ubJA = plan0.ubJA; spUb = plan0.spUb; thr_full = plan0.thr_full;
      nk1e = nk1; nk2e = nk2;
      if ((tro_short >> tp) & 1u) {
        uint32_t l1p, l2p;
        tro_len(tp, l1p, l2p);
        const uint32_t n1p = l1p >= k ? l1p - k + 1u : 0u, n2p = l2p >= k ? l2p - k + 1u : 0u;
        nk1e = n1p; nk2e = n2p;       // (the pair's own slots: what slot_valid calls existing)
        if (cutE < (uint32_t)U) cutUb = bases_behind(64u * cutE, n1p, n2p, P2, l1p, l2p);
        if (JA_ROUNDS < U) ubJA = bases_behind(64u * (uint32_t)JA_ROUNDS, n1p, n2p, P2, l1p, l2p);
        if (spT) spUb = bases_behind(128u - spT, n1p, n2p, P2, l1p, l2p);
        thr_full = (uint32_t)__builtin_amdgcn_readlane((int)tro_thr, (int)tp);
      }
    }
    if (!UNI) {
      const uint32_t pw = (uint32_t)__builtin_amdgcn_readfirstlane((int)pl_cur.w);
      if (pw == PLAN_VALID) {
        const uint32_t px = (uint32_t)__builtin_amdgcn_readfirstlane((int)pl_cur.x), py = (uint32_t)__builtin_amdgcn_readfirstlane((int)pl_cur.y),
                       pz = (uint32_t)__builtin_amdgcn_readfirstlane((int)pl_cur.z);
        cutE = px & 0xFFu; spT = (px >> 8) & 0xFFu; cutUb = px >> 16;
        ubJA = py & 0xFFFFu; thr_full = py >> 16;
        spLast = pz & 0xFFFFu; spUb = pz >> 16;
      } else {
        plan_cut(m_cur.L1, m_cur.L2);
        if (SPARSE) plan_sparse(m_cur.L1, m_cur.L2);
        // (every field fits: lengths below 2^15 are all the table is offered, see the host side)
        if (has_plans && lane == 0)
          P.plan_tab[plan_index(m_cur)] = make_uint4(cutE | (spT << 8) | (cutUb << 16), ubJA | (thr_full << 16), spLast | (spUb << 16), PLAN_VALID);
      }
    }
    uint32_t thr_r = thr_full;   // the smallest coverage that passes c * len for this read
    // a read without any invalid character (N, masked quality) -- most reads -- needs no validity window per slot: every existing
    // slot is a valid k-mer (table modes test validity before a probe; uniform branch around eight instructions per slot and round)
    const bool any_inv = TRI ? inv3[tp < 3u ? tp : 0u] : __ballot(inv_real != 0u) != 0ull;
    if (cutE < (uint32_t)U || JA_ROUNDS < U) {
      // the plan assumed len = L1 + L2; a read with invalid characters (N, masked qualities) has a lower threshold
      if (any_inv) {
        const uint32_t len = wave_sum_u32(lane_valid_bases());
        thr_r = cov_threshold(P.c, len);
      }
    }

    SHK_STAMP(11);
    // ---- everything behind the staging, for a compile-time E: rounds [0, E) first; E == U: all at once, no cut; E < 0: the anchored
    // extension (table modes), which returns false when the read has to take one of the other sequences after all.
    // (State and steps are declared INSIDE the lambda: shared between its instantiations from outside, hipcc 7.2 stops with
    //  "illegal VGPR to SGPR copy" on the ragged U = 6 / 8 table kernels; unused steps cost an instantiation nothing.) ----
    auto classify_staged = [&](auto e_const) -> bool {
    constexpr int E = decltype(e_const)::value;
    uint64_t pos[U];
    uint32_t okm[U];   // all ones where the slot's probe has to be made, else 0
    uint4 bk[U];
    // each probe ends up with one word of its bucket: the low word of the slot that matched (mt[j] says whether one did)
    bool mt[U];
    uint32_t slo[U];
    bool lane_any = false;
    // REP: a bucket is reduced to that word right away (LDS-summary modes; table modes since the anchored extension fills mt / slo
    // from the reference as well).  WALK_ROUNDS: the instantiations with registers to spare (LDS-summary modes, 80+ VGPRs) walk all
    // their probes per round; the table modes walk probe by probe (the per-round form spilled there: 5-18 % slower on large indices)
#ifndef SHK_ROUNDS_ALL
#define SHK_ROUNDS_ALL 0
#endif
    constexpr bool WALK_ROUNDS = LSUM || SHK_ROUNDS_ALL;
    constexpr bool ROUNDS = LSUM || ANCH || SHK_ROUNDS_ALL;
    const uint4 *tab16 = reinterpret_cast<const uint4 *>(KT ? P.ktab : P.tab);
    // (the position table: as below, for the anchored extension's sample, which probes `atab` in every table mode)
    const uint32_t pmask = (uint32_t)((1ull << P.tab_lg) - 1ull) & (uint32_t)P.bf_mask;
    const uint32_t pspare = 1u << P.tab_lg;
    const uint32_t bmask = KT ? (uint32_t)((1ull << P.ktab_lg) - 1ull) : pmask;
    const uint32_t tagmask = (uint32_t)(P.bf_mask >> P.tab_lg);
    const uint32_t spare = KT ? (1u << P.ktab_lg) : pspare;
    const bool tab_stream = KT ? (P.ktab_nt != 0u) : (P.tab_nt != 0u);
    // the word a slot's high word is compared with (0 = empty slot): tag | valid | displacement 0
    auto want_for = [&](const uint64_t ps) -> uint32_t {
      const uint32_t tag = __builtin_amdgcn_alignbit((uint32_t)(ps >> 32), (uint32_t)ps, P.tab_lg) & tagmask;
      return (tag << 8) | 0x80u;
    };
    // (KT: pos[j] = the compared word << 32 | the home bucket; a key that left its home bucket is in the same bucket of a later line)
    auto want_of = [&](const int j) -> uint32_t { return KT ? (uint32_t)(pos[j] >> 32) : want_for(pos[j]); };
    auto bucket_of = [&](const int j, const uint32_t d) -> uint32_t { return ((uint32_t)pos[j] + (KT ? 8u * d : d)) & bmask; };
    // both orientations of the k-mer of slot (lane, j): x = its bases first-base-low (the fw stream's window), y = first-base-high
    // (the rv stream's window = the k-mer as kmer_utils.hpp:67-69 packs it); ~x is the reverse complement (kmer_utils.hpp:47-55)
    auto windows = [&](const int j, uint64_t &x, uint64_t &y) {
      const uint32_t qU = rcap - k - ((uint32_t)lane + 64u * (U - 1));
      const uint32_t sf = ((uint32_t)lane & 15u) << 1, sr = (qU & 15u) << 1;
      const uint32_t *f = fw + ((uint32_t)lane >> 4) + 4 * j;
      const uint32_t *r = rv + (qU >> 4) + 4 * (U - 1 - j);
      const uint32_t d0 = f[0], d1 = f[1], d2 = f[2];
      const uint32_t e0 = r[0], e1 = r[1], e2 = r[2];
      x = ((uint64_t)__builtin_amdgcn_alignbit(d2, d1, sf) << 32) | __builtin_amdgcn_alignbit(d1, d0, sf);
      y = ((uint64_t)__builtin_amdgcn_alignbit(e2, e1, sr) << 32) | __builtin_amdgcn_alignbit(e1, e0, sr);
    };
    // slot pp exists and all its k characters are valid (process_read, slot_ok)
    const uint64_t kmask0 = (1ull << k) - 1ull;
    auto slot_valid = [&](const uint32_t pp) -> bool {
      const bool exists = (TRO && !(SHK_TRO_ABL & 2)) ? ((pp < nk1e) | ((pp - P2) < nk2e)) : ((pp < nk1) | ((pp - P2) < nk2));
      if (!any_inv) return exists;
      const uint32_t V = pp >> 6, vs = pp & 63u;
      const uint64_t v0 = vbits[V], v1 = vbits[V + 1];
      const uint64_t win = (v0 >> vs) | ((v1 << 1) << (63u - vs));
      return exists & ((win & kmask0) == kmask0);
    };
    // ---- canonical k-mers, hashes, summary probes, table probes of the rounds [JLO, JHI) ----
    // (returns false when nothing of these rounds can have matched: no probe passed its summary / no slot is a valid k-mer)
    // known (table modes, anchored extension): bit j set = slot (lane, j) is settled already -- mt[j] / slo[j] stay, no probe is made
    auto probe_rounds = [&](auto lo_const, auto hi_const, const uint32_t known) -> bool {
      constexpr int JLO = decltype(lo_const)::value, JHI = decltype(hi_const)::value;
      constexpr bool ALL = JLO == 0 && JHI == U;   // the only phase: its caller ends the read when nothing can have matched
#pragma unroll
      for (int j = JLO; j < JHI; ++j) {
        uint64_t x, y;
        windows(j, x, y);
        const uint64_t fwd = y & kmer_mask, rc = ~x & kmer_mask;
        if (KT) {
          uint32_t hb, hw;
          ktab_home(fwd, rc, k, P.ktab_w, P.ktab_lg - 3u, hb, hw);
          pos[j] = ((uint64_t)hw << 32) | hb;
          continue;
        }
        const uint64_t canon = fwd < rc ? fwd : rc;         // KmerBuilder.hpp:49, ReadAnalyzer.hpp:55
        const uint64_t hsh = xxh64_u64(canon);
        // (LDS-summary mode with a power-of-two size keeps the raw hash: every use below masks the bits it needs)
        pos[j] = POW2 ? (LSUM ? hsh : (hsh & P.bf_mask)) : bf_pos_np(hsh, P);
      }
      bool something = false;
      if (!LSUM) {
        // slot pp exists and all its k characters are valid; then the L2-resident summary
        bool ok[U];
#pragma unroll
        for (int j = JLO; j < JHI; ++j) {
          ok[j] = slot_valid((uint32_t)lane + 64u * j);
          if (ANCH) ok[j] = ok[j] & (((known >> j) & 1u) == 0u);
        }
        if (SUM) {
          uint32_t sw[U];
#pragma unroll
          for (int j = JLO; j < JHI; ++j) sw[j] = ok[j] ? P.sum32[(pos[j] >> P.sum_shift) >> 5] : 0u;
#pragma unroll
          for (int j = JLO; j < JHI; ++j) ok[j] = (sw[j] >> ((uint32_t)(pos[j] >> P.sum_shift) & 31u)) & 1u;
        }
        bool any = false;
#pragma unroll
        for (int j = JLO; j < JHI; ++j) { okm[j] = ok[j] ? 0xFFFFFFFFu : 0u; any |= ok[j]; }
        something = __ballot(any) != 0ull;
      } else if (LX) {
        // the exact table in LDS: displacement of the position's group, then the slot (shark_internal.hpp)
        const uint32_t *T = lsum;
        const char *D = reinterpret_cast<const char *>(lsum + LTAB_T_WORDS);
        const uint32_t tagmask15 = (uint32_t)(P.bf_mask >> LTAB_SLOT_LG);
        const uint32_t gmask2 = (tagmask15 & ((1u << LTAB_GROUP_LG) - 1u)) << 1;   // (pos[] is the raw hash: only the filter's bits count)
        uint32_t dd[U], ee[U], tg[U];
#pragma unroll
        for (int j = JLO; j < JHI; ++j) {
          dd[j] = *reinterpret_cast<const uint16_t *>(D + (((uint32_t)pos[j] >> (LTAB_SLOT_LG - 1)) & gmask2));
          tg[j] = __builtin_amdgcn_alignbit((uint32_t)(pos[j] >> 32), (uint32_t)pos[j], LTAB_SLOT_LG) & tagmask15;
        }
#pragma unroll
        for (int j = JLO; j < JHI; ++j) {
          const uint32_t base = (uint32_t)pos[j] + (tg[j] >> LTAB_GROUP_LG) * P.lsum_shift;   // (lds_table.hpp: the bits above the group spread the slots; lsum_shift = the multiplier)
          ee[j] = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(T) + (((base + dd[j]) << 2) & ((LTAB_T_WORDS - 1u) << 2)));
        }
        bool esc = false, any = false;
#pragma unroll
        for (int j = JLO; j < JHI; ++j) {
          mt[j] = (ee[j] >> 13) == ((tg[j] << 1) | 1u);
          slo[j] = ee[j] & LTAB_ESC;
          okm[j] = mt[j] ? 0xFFFFFFFFu : 0u;
          any |= mt[j];
          esc |= mt[j] & (slo[j] == LTAB_ESC);
        }
        // a multi-gene list (or a gene id beyond 13 bits) is not in the entry: the position table answers for these rounds
        something = __ballot(esc) != 0ull;
        if (!something) lane_any |= any;
      } else {
        uint32_t si[U], sw[U];
#pragma unroll
        for (int j = JLO; j < JHI; ++j) {
          si[j] = __builtin_amdgcn_alignbit((uint32_t)(pos[j] >> 32), (uint32_t)pos[j], P.lsum_shift);   // low LSL bits = summary index
          sw[j] = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(lsum) + ((si[j] >> 3) & (UG::SUM_BITS / 8 - 4)));
        }
        uint32_t any = 0;
#pragma unroll
        for (int j = JLO; j < JHI; ++j) {
          okm[j] = (uint32_t)__builtin_amdgcn_sbfe((int)sw[j], si[j], 1u);   // v_bfe_i32: bit (si & 31), sign extended
          any |= okm[j];
        }
        something = __ballot(any != 0u) != 0ull;
      }
      if (something) {
        // ---- position table: the probes that passed read their home bucket, the others the spare empty bucket ----
#pragma unroll
        for (int j = JLO; j < JHI; ++j) {
          const uint32_t bb = (uint32_t)pos[j] & bmask;
          const uint32_t bi = (bb & okm[j]) | (spare & ~okm[j]);
          if (!LSUM && tab_stream) {   // a table far beyond the caches: streaming loads (49.8 -> 54.6 G lookups/s, tools/gather_bench)
            const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(tab16) + bi);
            bk[j] = make_uint4(v.x, v.y, v.z, v.w);
          } else {
            bk[j] = load_bucket<LSUM && UNI>(tab16, bi);   // (the ragged instantiation runs out of registers with the short addresses)
          }
        }
        bool lane_more = false;
        bool more[U];
#pragma unroll
        for (int j = JLO; j < JHI; ++j) {
          const uint32_t want = want_of(j);
          const bool m0 = bk[j].y == want, m1 = bk[j].w == want;
          if (ROUNDS) {
            const bool kn = ANCH && ((known >> j) & 1u) != 0u;   // (a settled slot read the spare bucket: nothing matched)
            mt[j] = kn ? mt[j] : (m0 | m1);
            slo[j] = kn ? slo[j] : (m0 ? bk[j].x : bk[j].z);
          }
          lane_any |= m0 | m1;
          more[j] = !(m0 | m1) & ((bk[j].x & TAB_OVERFLOW) != 0u);   // some key of this home bucket lives further down the path
          lane_more |= more[j];
        }
        if (__ballot(lane_more) && !SHK_ABL(P, 8u)) {   // (ablation 8: no walks)
          // rare: the key may sit behind its (full) home bucket
          if (WALK_ROUNDS) {
            // round d looks at bucket home+d of every probe that is still searching, all loads in flight together (the
            // others read the spare bucket: one line for the wave) -- a memory round trip per displacement, not per probe
            uint32_t d = 0;
            do {
              ++d;
#pragma unroll
              for (int j = JLO; j < JHI; ++j) bk[j] = load_bucket<LSUM && UNI>(tab16, more[j] ? bucket_of(j, d) : spare);
              lane_more = false;
#pragma unroll
              for (int j = JLO; j < JHI; ++j) {
                const uint32_t want = KT ? want_of(j) : (want_of(j) | d);   // (a slot of the position table carries its displacement)
                const bool n0 = bk[j].y == want, n1 = bk[j].w == want;
                const bool found = more[j] & (n0 | n1);
                const bool ends = (bk[j].y == 0u) | (bk[j].w == 0u) | (d >= 63u);   // a free slot ends every search
                slo[j] = found ? (n0 ? bk[j].x : bk[j].z) : slo[j];
                mt[j] |= found;
                lane_any |= found;
                more[j] = more[j] & !found & !ends;
                lane_more |= more[j];
              }
            } while (__ballot(lane_more));
          } else {
            bool walked[U];
#pragma unroll
            for (int j = JLO; j < JHI; ++j) walked[j] = more[j];
            walk_probe_paths<U, !KT>(tab16, bk, more, lane_any, want_of, bucket_of, (uint32_t)JLO, (uint32_t)JHI);   // (a key found is moved into bk[j] in home form)
            if (ROUNDS) {
#pragma unroll
              for (int j = JLO; j < JHI; ++j) {
                const bool f = walked[j] & (bk[j].y == want_of(j));
                mt[j] |= f;
                slo[j] = f ? bk[j].x : slo[j];
              }
            }
          }
        }
      } else if (!LX && !ALL) {
        // nothing of these rounds passed the summary: no matches (the hit path may still run for the other rounds)
#pragma unroll
        for (int j = JLO; j < JHI; ++j) {
          if (ROUNDS) {
            const bool kn = ANCH && ((known >> j) & 1u) != 0u;   // (settled slots stay)
            mt[j] = kn && mt[j];
            slo[j] = kn ? slo[j] : 0u;
          } else bk[j] = make_uint4(0u, 0u, 0u, 0u);   // (an empty slot's compared word is 0: matches nothing)
        }
      }
      return LX || something;
    };
    // bases covered by the slots of a hit mask (one ballot per round): the union of [p, p + k), counted as the hit path
    // counts a gene's coverage
    const uint64_t kthr_c = 1ull << (64u - k);
    auto cover = [&](const uint64_t Hc, const uint64_t Hp) -> uint32_t {
      const uint64_t t = (Hc << (63u - (uint32_t)lane)) | ((Hp >> 1) >> (uint32_t)lane);
      return (uint32_t)__builtin_popcountll(__ballot(t >= kthr_c));
    };
    // ... by ALL k-mers of the rounds [0, J) that are in the filter
    auto found_cover = [&](auto j_const) -> uint32_t {
      constexpr int J = decltype(j_const)::value;
      uint64_t Hp = 0ull;
      uint32_t cv = 0;
#pragma unroll
      for (int j = 0; j < J; ++j) {
        bool m;
        if (ROUNDS) m = mt[j];
        else { const uint32_t want = want_of(j); m = (bk[j].y == want) | (bk[j].w == want); }
        const uint64_t Hc = __ballot(m);
        cv += cover(Hc, Hp);
        Hp = Hc;
      }
      return cv + cover(0ull, Hp);
    };
    // ... by the k-mers of the rounds [0, J) that can belong to ONE gene, maximised over the genes: a match with a single-gene
    // list counts for that gene only, a match with a multi-gene list for every gene.  (The matches of an off-target read of a
    // large reference are isolated k-mers of different genes: k bases each, whatever their number.)  At most 8 genes are
    // looked at; beyond that the answer is "everything" (no cut).
    auto gene_cover = [&](auto j_const) -> uint32_t {
      constexpr int J = decltype(j_const)::value;
      uint64_t H[J], W[J];
      uint32_t gid[J];
#pragma unroll
      for (int j = 0; j < J; ++j) {
        uint32_t lo;
        bool m;
        if (ROUNDS) { lo = slo[j]; m = mt[j]; }
        else {
          const uint32_t want = want_of(j);
          const bool m0 = bk[j].y == want, m1 = bk[j].w == want;
          lo = m0 ? bk[j].x : bk[j].z;
          m = m0 | m1;
        }
        const bool multi = (lo >> 31) != 0u;
        gid[j] = lo & 0xFFFFu;
        H[j] = __ballot(m & !multi);
        W[j] = __ballot(m & multi);
      }
      uint32_t best = 0;
      {
        uint64_t Hp = 0ull;
#pragma unroll
        for (int j = 0; j < J; ++j) { best += cover(W[j], Hp); Hp = W[j]; }
        best += cover(0ull, Hp);
      }
      for (int it = 0; it < 8; ++it) {
        uint32_t g = 0;
        bool have = false;
#pragma unroll
        for (int j = 0; j < J; ++j)
          if (!have && H[j] != 0ull) {
            g = (uint32_t)__builtin_amdgcn_readlane((int)gid[j], (int)__builtin_ctzll(H[j]));
            have = true;
          }
        if (!have) return best;
        uint64_t Hp = 0ull;
        uint32_t cv = 0;
#pragma unroll
        for (int j = 0; j < J; ++j) {
          const uint64_t G = __ballot((((H[j] >> (uint32_t)lane) & 1ull) != 0ull) & (gid[j] == g));
          H[j] &= ~G;
          const uint64_t M = G | W[j];
          cv += cover(M, Hp);
          Hp = M;
        }
        cv += cover(0ull, Hp);
        best = cv > best ? cv : best;
      }
#pragma unroll
      for (int j = 0; j < J; ++j)
        if (H[j] != 0ull) return 0xFFFFFFFFu;
      return best;
    };
    // ---- the vote over the matches of the rounds [0, J) (ReadAnalyzer.hpp:56-62, :79-108) ------------------------------------
    // FINAL: every slot is settled, the result is final and written.  Else (the early decision): the slots not settled yet cover
    // `ub_rest` bases, so every gene's final coverage is at most (its coverage now) + ub_rest, and the best gene's is at least
    // what it is now.  If that gene alone is best, passes c * len already, and leads every other gene -- those without a match so
    // far included -- by more than ub_rest, the remaining probes cannot change the outcome: it is the read's only association
    // (whatever its final coverage and k-mer count, which the reference does not output).  Returns true when the read is settled.
    auto vote = [&](auto j_const, auto final_const, const uint32_t ub_rest) -> bool {
      constexpr int J = decltype(j_const)::value;
      constexpr bool FINAL = decltype(final_const)::value;
      if (!__ballot(lane_any) || SHK_ABL(P, 16u)) return FINAL;   // nothing matched (ablation 16: no hit path)
      {
        // ================= something matched in the table: the hit path =================
        KernargParams H = kernarg_params();
        const uint32_t hk = H->k;
        const uint64_t kmask = (1ull << hk) - 1ull;
        uint32_t cur[J], rs[J], re[J];
        bool hit[J], multi[J];
        uint32_t payload[J];
        bool any2 = false;
#pragma unroll
        for (int j = 0; j < J; ++j) {
          // the probe was issued without looking at the slot: it has to exist and be a valid k-mer (process_read, slot_ok)
          const uint32_t pp = (uint32_t)lane + 64u * j;
          const bool exists = (pp < nk1) | ((pp - P2) < nk2);
          const uint32_t V = pp >> 6, vs = pp & 63u;
          const uint64_t v0 = vbits[V], v1 = vbits[V + 1];
          const uint64_t win = (v0 >> vs) | ((v1 << 1) << (63u - vs));
          if (!ROUNDS) {
            const uint32_t want = want_of(j);
            const bool m0 = bk[j].y == want, m1 = bk[j].w == want;
            mt[j] = m0 | m1;
            slo[j] = m0 ? bk[j].x : bk[j].z;
          }
          hit[j] = LSUM ? (mt[j] & exists & ((win & kmask) == kmask)) : mt[j];   // (table modes settled that before the probe)
          any2 |= hit[j];
          payload[j] = slo[j] & TAB_PAYLOAD;
          multi[j] = (slo[j] >> 31) != 0u;
        }
        if (!__ballot(any2)) return FINAL;
        // ---- one gene only, or one gene far ahead?  The usual read: (almost) every k-mer found belongs to a single-gene list of one
        // and the same gene g.  With nothing else found, ReadAnalyzer's map has the one entry g: its coverage is the union of the
        // hits' intervals, its k-mer count their number -- no merge, no lists (exact, FINAL or not).  With a few other k-mers found
        // (a large reference: every fiftieth k-mer of a read is also some other gene's), bounds settle the EARLY decision without
        // reading a list: g's coverage is at least what its single-gene hits cover, any other gene's at most what all other hits
        // cover.  If the first passes c * len and exceeds the second by more than the slots not settled yet can still cover, g is
        // the read's only association whatever the lists say.
        {
          uint64_t Sg[J];
          uint32_t p0 = 0;
          bool have0 = false;
#pragma unroll
          for (int j = 0; j < J; ++j) {
            Sg[j] = __ballot(hit[j] & !multi[j]);
            if (!have0 && Sg[j] != 0ull) {
              p0 = (uint32_t)__builtin_amdgcn_readlane((int)payload[j], (int)__builtin_ctzll(Sg[j]));
              have0 = true;
            }
          }
          bool lane_other = false;
#pragma unroll
          for (int j = 0; j < J; ++j) lane_other |= hit[j] & (multi[j] | (payload[j] != p0));
          const bool alone = __ballot(lane_other) == 0ull;   // (then Sg[] are g's hits already)
          if (have0 && (alone || !FINAL)) {
            uint32_t cov = 0, oth = 0;
            // Bounds on the scalar unit first -- they settle the usual read from a gene, whose coverage passes with room to spare:
            // g's hits cover AT LEAST one base each and k - 1 more behind the last (the union of [p, p + k) contains every p);
            // the other hits cover AT MOST their number + (k - 1) per run of neighbouring slots.  thr_r is the smallest coverage that
            // passes c * len (or more, where it was planned for a read without invalid characters: still a sufficient test).
            bool out1 = false, settled_by_bounds = false;
            {
              uint32_t n_mine = 0, n_oth = 0, runs_oth = 0;
              uint64_t carry = 0ull;
#pragma unroll
              for (int j = 0; j < J; ++j) {
                if (!alone) {
                  const bool mine = hit[j] & !multi[j] & (payload[j] == p0);
                  Sg[j] = __ballot(mine);
                  const uint64_t Oc = __ballot(hit[j] & !mine);
                  n_oth += (uint32_t)__builtin_popcountll(Oc);
                  runs_oth += (uint32_t)__builtin_popcountll(Oc & ~((Oc << 1) | carry));
                  carry = Oc >> 63;
                }
                n_mine += (uint32_t)__builtin_popcountll(Sg[j]);
              }
              const uint32_t cov_lb = n_mine + hk - 1u, oth_ub = n_oth + (hk - 1u) * runs_oth;
              if (thr_r != 0u && cov_lb >= thr_r && (FINAL || cov_lb > oth_ub + ub_rest)) { out1 = true; settled_by_bounds = true; }
            }
            if (!settled_by_bounds) {
              // (rare: a read near the threshold, or with many k-mers of other genes) the exact counts
              if (!alone) {
                uint64_t Op = 0ull;
#pragma unroll
                for (int j = 0; j < J; ++j) {
                  const uint64_t Oc = __ballot(hit[j] & !(!multi[j] & (payload[j] == p0)));
                  oth += cover(Oc, Op);
                  Op = Oc;
                }
                oth += cover(0ull, Op);
              }
#pragma unroll
              for (int j = 0; j < J; ++j) cov += cover(Sg[j], j ? Sg[j - 1] : 0ull);
              cov += cover(0ull, Sg[J - 1]);
              const uint32_t len = wave_sum_u32(lane_valid_bases());
              const bool pass = (double)cov >= H->c * (double)len;
              // FINAL (alone): the threshold decides (one gene: --single changes nothing).  Else: g has to lead by more than ub_rest
              out1 = pass && (FINAL || cov > oth + ub_rest);
            }
            if (out1 && lane == 0 && !SHK_ABL(P, 64u)) {
              const ClassifyOut *O = H->out;
              O->count[read] = 1u;
              uint2 pk;
              pk.x = p0 & 0xFFFFu;
              pk.y = 0u;
              *reinterpret_cast<uint2 *>(O->inl + (uint64_t)read * SHK_INLINE_IDS) = pk;
            }
            if (alone || out1) return FINAL || out1;
          }
        }
        {
          bool lane_multi = false;
#pragma unroll
          for (int j = 0; j < J; ++j) lane_multi |= hit[j] & multi[j];
          if (!FINAL) {
            // The early decision settles a read only when ONE gene leads.  A read whose k-mers mostly carry multi-gene lists -- a
            // fragment of a region that genes share -- ends as a tie or is decided by the few k-mers that are one gene's alone: the
            // merge below would run to the end, fail, and run again behind the remaining probes (measured on the configs[2]
            // reference, a tenth of whose on-target pairs are such: 17.8 ms per 10 M pairs at 100 % on-target against 14.0 on the
            // same reference without shared halves).  So: most matches multi-gene -- no early attempt.
            uint32_t n_multi = 0, n_hits = 0;
#pragma unroll
            for (int j = 0; j < J; ++j) {
              n_multi += (uint32_t)__builtin_popcountll(__ballot(hit[j] & multi[j]));
              n_hits += (uint32_t)__builtin_popcountll(__ballot(hit[j]));
            }
            if (2u * n_multi > n_hits) return false;
          }
          if (__ballot(lane_multi)) {   // multi-gene lists (rare): entry r gives start/len/first gene
            ListEntry le[J];
#pragma unroll
            for (int j = 0; j < J; ++j) le[j] = H->ent[(hit[j] & multi[j]) ? payload[j] : 0u];
#pragma unroll
            for (int j = 0; j < J; ++j) {
              if (hit[j] & multi[j]) {
                rs[j] = le[j].start;
                re[j] = le[j].len != 0xFFFFu ? le[j].start + le[j].len : H->ent[payload[j] + 1].start;
                cur[j] = le[j].gene0;
              } else {
                rs[j] = 0; re[j] = 0; cur[j] = hit[j] ? (payload[j] & 0xFFFFu) : GENE_INF;
              }
            }
          } else {
#pragma unroll
            for (int j = 0; j < J; ++j) { rs[j] = 0; re[j] = 0; cur[j] = hit[j] ? (payload[j] & 0xFFFFu) : GENE_INF; }
          }
          // len = number of valid characters of the joined string (ReadAnalyzer.hpp:46-49)
          const uint32_t len = wave_sum_u32(lane_valid_bases());
          uint32_t best_cov = 0, best_nk = 0, n_best = 0, second_cov = 0;
          uint32_t best_id[SHK_INLINE_IDS] = {0, 0, 0, 0};
          // ---- k-way merge over the hit lists, ascending gene id (see process_read for the derivation) ----
          for (;;) {
            uint32_t mymin = GENE_INF;
#pragma unroll
            for (int j = 0; j < J; ++j) mymin = cur[j] < mymin ? cur[j] : mymin;
            const uint32_t g = wave_min_u32(mymin);
            if (g == GENE_INF || SHK_ABL(P, 32u)) break;   // (ablation 32: no merge)
            uint32_t nk = 0, cov = 0;
            uint64_t Hm[J];
            bool more_ids = false;
#pragma unroll
            for (int j = 0; j < J; ++j) {
              const bool h = cur[j] == g;
              Hm[j] = __ballot(h);
              rs[j] += h ? 1u : 0u;
              more_ids |= h & (rs[j] < re[j]);
            }
#pragma unroll
            for (int j = 0; j < J; ++j) {
              nk += (uint32_t)__builtin_popcountll(Hm[j]);
              cov += cover(Hm[j], j ? Hm[j - 1] : 0ull);
            }
            cov += cover(0ull, Hm[J - 1]);
            if (__ballot(more_ids)) {
#pragma unroll
              for (int j = 0; j < J; ++j)
                if ((Hm[j] >> lane) & 1ull) cur[j] = rs[j] < re[j] ? (uint32_t)H->ids[rs[j]] : GENE_INF;
            } else {
#pragma unroll
              for (int j = 0; j < J; ++j) cur[j] = ((Hm[j] >> lane) & 1ull) ? GENE_INF : cur[j];
            }
            // arg-max with ties in ascending gene order, as selects (see the compiler note in process_read)
            { const uint32_t lower = cov < best_cov ? cov : best_cov; second_cov = lower > second_cov ? lower : second_cov; }   // largest coverage that is not the best one's
            const bool gt = (cov > best_cov) | ((cov == best_cov) & (nk > best_nk));
            const bool eq = (cov == best_cov) & (nk == best_nk);
            best_id[0] = gt ? g : best_id[0];
#pragma unroll
            for (int i = 1; i < SHK_INLINE_IDS; ++i) best_id[i] = (eq & (n_best == (uint32_t)i)) ? g : best_id[i];
            n_best = gt ? 1u : (eq ? n_best + 1u : n_best);
            best_cov = gt ? cov : best_cov;
            best_nk = gt ? nk : best_nk;
          }
          // ---- threshold + --single (ReadAnalyzer.hpp:104) ----
          uint32_t n_out = 0;
          if (FINAL) {
            if (n_best > 0 && (double)best_cov >= H->c * (double)len && (!H->single || n_best == 1)) n_out = n_best;
          } else {
            if (n_best == 1 && best_cov > second_cov + ub_rest && (double)best_cov >= H->c * (double)len) n_out = 1;
          }
          if (n_out > 0 && lane == 0 && !SHK_ABL(P, 64u)) {   // (ablation 64: no result store)
            const ClassifyOut *O = H->out;
            O->count[read] = n_out;
            uint2 pk;
            pk.x = (best_id[0] & 0xFFFFu) | (best_id[1] << 16);
            pk.y = (best_id[2] & 0xFFFFu) | (best_id[3] << 16);
            *reinterpret_cast<uint2 *>(O->inl + (uint64_t)read * SHK_INLINE_IDS) = pk;
            if (n_out > SHK_INLINE_IDS) {
              const uint32_t qi = atomicAdd(&O->counters[CTR_TIE], 1u);
              O->tie_queue[3 * qi + 0] = read;
              O->tie_queue[3 * qi + 1] = best_cov;
              O->tie_queue[3 * qi + 2] = best_nk;
            }
          }
          return FINAL || n_out > 0;
        }
      }
    };
    using I0 = std::integral_constant<int, 0>;
    using IU = std::integral_constant<int, U>;
    // ---- the anchored extension (table modes) ---------------------------------------------------------------------
    // A k-mer that is in the table knows where it occurs in the reference (`anchor`, one occurrence per table slot).  A read
    // drawn from a gene matches the reference base for base around such a place, so most of its k-mers are the reference's k-mers
    // at the neighbouring positions -- and what a probe of those returns is stored per reference position (`refpay`), contiguously:
    // 64 slots of a round read 256 bytes of it and 24 bytes of the packed reference instead of 64 buckets at 64 hashed addresses.
    //  (1) sample: SHK_ANCH_SAMPLE (four) slots spread over each mate, one per lane, probed through the table as ever (one hash per lane;
    //      measured on the configs[2] index per 10 M pairs at 0 / 50 / 100 % on-target: 8 slots 36.4 / 28.7 / 27.9 ms, 4: 34.5 / 27.5 / 27.8, 2: 33.5 / 26.5 / 28.5);
    //  (2) the first match of each mate gives that mate's anchor -- reference position + relative orientation --, which the sampled
    //      bucket carries itself (atab);
    //  (3) every slot of an anchored mate is compared with the reference k-mer at its implied position, as a 2k-bit compare of the
    //      read's window with the reference's.  Equal k-mers have equal filter positions, hence equal table slots: the slot is
    //      settled with EXACTLY what a probe would have returned.  Unequal: the slot stays open -- nothing is assumed;
    //  (4) the early decision over everything settled, the open slots counting as "could all be this gene's": decided reads
    //      never probe their open slots (typically the k-mers around a sequencing error);
    //  (5) otherwise the open slots are probed as ever and the final vote runs.
    // A read whose sample matches nothing, or whose anchors do not hold (fewer than 4 slots confirmed: a chance match), takes the
    // usual path below.  Results are the reference's for every read (tests: test_anchored_extension_*, the fuzzer, the scale tests).
    auto anchored = [&]() -> bool {
      KernargParams H = kernarg_params();
      const uint32_t ref_total = H->ref_total;
      if (!ref_total) return false;
      // (1)  (SHK_ANCH_SAMPLE slots per mate, a power of two)
      constexpr uint32_t NS = SHK_ANCH_SAMPLE;
      const uint32_t st1 = nk1 > NS ? nk1 / NS : 1u, st2 = nk2 > NS ? nk2 / NS : 1u;
      const bool sm2 = ((uint32_t)lane & NS) != 0u;
      const uint32_t in_mate = ((uint32_t)lane & (NS - 1u)) * (sm2 ? st2 : st1);
      const bool s_exists = ((uint32_t)lane < 2u * NS) & (in_mate < (sm2 ? nk2 : nk1));
      const uint32_t ss = s_exists ? (sm2 ? P2 + in_mate : in_mate) : 0u;
      uint64_t s_fwd, s_rc;
      {
        const uint32_t q = rcap - k - ss;
        const uint32_t *f = fw + (ss >> 4);
        const uint32_t *r = rv + (q >> 4);
        const uint32_t d0 = f[0], d1 = f[1], d2 = f[2];
        const uint32_t e0 = r[0], e1 = r[1], e2 = r[2];
        const uint32_t af = (ss & 15u) << 1, ar = (q & 15u) << 1;
        const uint64_t x = ((uint64_t)__builtin_amdgcn_alignbit(d2, d1, af) << 32) | __builtin_amdgcn_alignbit(d1, d0, af);
        const uint64_t y = ((uint64_t)__builtin_amdgcn_alignbit(e2, e1, ar) << 32) | __builtin_amdgcn_alignbit(e1, e0, ar);
        s_fwd = y & kmer_mask;
        s_rc = ~x & kmer_mask;
      }
      bool s_ok = s_exists && slot_valid(ss);
      if (!__ballot(s_ok)) return false;   // (no sampled slot is a valid k-mer -- masked qualities, N: nothing to hash)
      const bool s_isrc = !(s_fwd < s_rc);
      const uint64_t s_hash = xxh64_u64(s_isrc ? s_rc : s_fwd);
      const uint64_t s_pos = POW2 ? (s_hash & P.bf_mask) : bf_pos_np(s_hash, P);
      if (SUM) {
        const uint32_t sw = s_ok ? P.sum32[(s_pos >> P.sum_shift) >> 5] : 0u;
        s_ok = (sw >> ((uint32_t)(s_pos >> P.sum_shift) & 31u)) & 1u;
      }
      if (!__ballot(s_ok)) return false;
      // (the sample probes `atab`: the position table's buckets with a slot's occurrence in the reference in place of its list --
      //  "is this k-mer in the index" and "where in the reference" in one memory round trip instead of two dependent ones)
      const uint32_t sb = s_ok ? ((uint32_t)s_pos & pmask) : pspare;
      const uint4 *atab16 = reinterpret_cast<const uint4 *>(H->atab);
      uint4 sbk;
      if (P.tab_nt) {
        const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(atab16) + sb);
        sbk = make_uint4(v.x, v.y, v.z, v.w);
      } else {
        sbk = atab16[sb];
      }
      const uint32_t s_want = want_for(s_pos);
      const bool sm0 = sbk.y == s_want, sm1 = sbk.w == s_want;       // (home bucket only: a displaced key gives no anchor)
      const uint64_t SH = __ballot(s_ok & (sm0 | sm1));
      if (!SH) return false;
#if SHK_ANCH_CUT == 1
      return true;
#endif
      // (2)
      const uint32_t s_anc = sm0 ? sbk.x : sbk.z;
      uint32_t ax[2] = {0u, 0u}, as0[2] = {0u, 0u};
      bool aopp[2] = {false, false}, ahave[2] = {false, false};
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const uint32_t mask = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(SH >> (NS * m)) & ((1u << NS) - 1u));
        if (mask) {
          const int ln = __builtin_amdgcn_readfirstlane(__builtin_ctz(mask) + (int)NS * m);
          const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)s_anc, ln);
          ahave[m] = a != 0xFFFFFFFFu;
          ax[m] = a & 0x7FFFFFFFu;
          aopp[m] = ((a >> 31) != 0u) != (__builtin_amdgcn_readlane((int)(s_isrc ? 1u : 0u), ln) != 0);
          as0[m] = (uint32_t)__builtin_amdgcn_readlane((int)ss, ln);
        }
      }
      if (!(ahave[0] | ahave[1])) return false;
#if SHK_ANCH_CUT == 2
      return ax[0] + ax[1] != 12345u;
#endif
#if SHK_ANCH_BASEWISE
      // (3) base by base.  An anchor maps a mate onto the reference linearly: the base at packed position b stands against
      // reference base x0 - s0 + b (same strand), or against the complement of reference base x0 + s0 + k - 1 - b (other strand).
      // Lane (m, c) = (lane >> 5, lane & 31) compares the 16 bases of chunk c of mate m in ONE xor of two dwords -- the read's
      // 2-bit codes (fw stream) against the reference's (ref2, same layout; reversed and complemented for the other strand) -- and
      // leaves one bit per base in `mbits`.  A slot's k-mer equals the reference k-mer at its implied position iff its k bits are
      // all set: the same window test the validity stream gets (round 3 compared two 2k-bit windows per slot and round: six LDS
      // reads, three reference dwords and four funnel shifts per slot instead of one 64-bit window).  Equal k-mers have equal filter
      // positions, hence equal table slots: such a slot is settled with EXACTLY what a probe would have returned (refpay).
      // (32 lanes per mate, 16 bases each: mates of more than 512 bases take the usual path)
      if (nk1 + k > 513u || nk2 + k > 513u) return false;
      const uint32_t *refpay = H->refpay;
      // what a probe of the reference k-mer at each slot's implied position returns: requested FIRST, for every slot whose position
      // lies in the reference (64 slots of a round read 256 contiguous bytes), so that these loads and the reference bases below
      // are one memory round trip, not two
      uint32_t inb_mask = 0u, okv_mask = 0u;
#pragma unroll
      for (int j = 0; j < U; ++j) {
        const uint32_t pp = (uint32_t)lane + 64u * j;
        const bool in2 = (pp - P2) < nk2;
        const bool okv = slot_valid(pp);
        const bool have = in2 ? ahave[1] : ahave[0];
        const bool opp = in2 ? aopp[1] : aopp[0];
        const uint32_t x0 = in2 ? ax[1] : ax[0];
        const uint32_t dd = pp - (in2 ? as0[1] : as0[0]);          // (modulo 2^32: out-of-range positions fail the bound below)
        const uint32_t xr = opp ? x0 - dd : x0 + dd;               // where the slot's k-mer starts in the reference
        const bool inb = okv & have & (xr < ref_total);
        slo[j] = refpay[inb ? xr : 0u];
        inb_mask |= inb ? (1u << j) : 0u;
        okv_mask |= okv ? (1u << j) : 0u;
      }
      {
        const uint32_t *ref2 = H->ref2;
        const uint32_t m = (uint32_t)lane >> 5, c16 = ((uint32_t)lane & 31u) << 4;
        const uint32_t len_m = m ? (nk2 ? nk2 + k - 1u : 0u) : (nk1 ? nk1 + k - 1u : 0u);
        const uint32_t b0 = (m ? P2 : 0u) + c16;                       // packed position of the chunk's first base
        const uint32_t n_in = c16 < len_m ? (len_m - c16 < 16u ? len_m - c16 : 16u) : 0u;
        const bool hv = m ? ahave[1] : ahave[0], op = m ? aopp[1] : aopp[0];
        const uint32_t x0 = m ? ax[1] : ax[0], s0 = m ? as0[1] : as0[0];
        // the 16 reference bases the chunk stands against, first one at `lo` (mod 2^32: a chunk that would leave the reference
        // fails the bound and matches nothing -- its slots stay open)
        const uint32_t lo = op ? x0 + s0 + k - 16u - b0 : x0 + b0 - s0;
        const bool inr = hv & (n_in != 0u) & (lo < ref_total);
        const uint32_t ls = inr ? lo : 0u, bs = n_in ? b0 : 0u;
        const uint32_t g0 = ref2[ls >> 4], g1 = ref2[(ls >> 4) + 1u];
        const uint32_t r0 = fw[bs >> 4], r1 = fw[(bs >> 4) + 1u];
        uint32_t G = __builtin_amdgcn_alignbit(g1, g0, (ls & 15u) << 1);
        if (op) {   // the other strand: base order reversed (2-bit groups), complemented
          G = __builtin_bitreverse32(G);
          G = ~(((G >> 1) & 0x55555555u) | ((G & 0x55555555u) << 1));
        }
        const uint32_t R = __builtin_amdgcn_alignbit(r1, r0, (bs & 15u) << 1);
        const uint32_t df = R ^ G;
        uint32_t e = ~(df | (df >> 1)) & 0x55555555u;                 // bit 2 i: base i agrees
        e = (e | (e >> 1)) & 0x33333333u;
        e = (e | (e >> 2)) & 0x0F0F0F0Fu;
        e = (e | (e >> 4)) & 0x00FF00FFu;
        e = (e | (e >> 8)) & 0xFFFFu;
        const uint32_t M16 = inr ? (e & ((1u << n_in) - 1u)) : 0u;
        // bytes 2 c and 2 c + 1 of the mate's part of the stream (mate 2 starts at byte P2 / 8; mate 1's last chunk may reach past it:
        // those bytes are mate 2's).  Bytes behind the mate are cleared as far as the stream goes: no stale bit of an earlier read
        constexpr uint32_t MBYTES = vbit_words_for(S) * 8u;
        uint8_t *mb = reinterpret_cast<uint8_t *>(mbits);
        const uint32_t by = b0 >> 3, end = m ? MBYTES : (P2 >> 3);
        if (by < end) mb[by] = (uint8_t)M16;
        if (by + 1u < end) mb[by + 1u] = (uint8_t)(M16 >> 8);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      uint32_t known = 0u, n_match = 0u, n_open = 0u, n_runs = 0u;
      uint64_t carry = 0ull;
#pragma unroll
      for (int j = 0; j < U; ++j) {
        const bool okv = ((okv_mask >> j) & 1u) != 0u;
        const uint64_t m0 = mbits[j], m1 = mbits[j + 1];           // (slot pp = lane + 64 j: word j, bit `lane`)
        const uint64_t mwin = (m0 >> (uint32_t)lane) | ((m1 << 1) << (63u - (uint32_t)lane));
        const bool mm = (((inb_mask >> j) & 1u) != 0u) & ((mwin & kmask0) == kmask0) & (slo[j] != REFPAY_NONE);
        mt[j] = mm;
        known |= (mm | !okv) ? (1u << j) : 0u;                      // (a slot that does not exist or is no valid k-mer needs no probe either)
        lane_any |= mm;
        n_match += (uint32_t)__builtin_popcountll(__ballot(mm));
        // what the open slots can still cover, as an upper bound from their runs (scalar): r consecutive slots cover r + k - 1
        // bases; runs less than k - 1 slots apart overlap, which the bound ignores (it is exact for the usual read: a run per error)
        const uint64_t Uc = __ballot(okv & !mm);
        n_open += (uint32_t)__builtin_popcountll(Uc);
        n_runs += (uint32_t)__builtin_popcountll(Uc & ~((Uc << 1) | carry));
        carry = Uc >> 63;
      }
      const uint32_t ub = n_open + (k - 1u) * n_runs;
#else
      // (3)
      uint32_t known = 0u, n_match = 0u, ub = 0u;
      uint64_t Uprev = 0ull;
      const uint32_t *refpay = H->refpay;
      const uint32_t *ref2 = H->ref2;
#pragma unroll
      for (int j = 0; j < U; ++j) {
        const uint32_t pp = (uint32_t)lane + 64u * j;
        const bool in2 = (pp - P2) < nk2;
        const bool okv = slot_valid(pp);
        const bool have = in2 ? ahave[1] : ahave[0];
        const bool opp = in2 ? aopp[1] : aopp[0];
        const uint32_t x0 = in2 ? ax[1] : ax[0];
        const uint32_t dd = pp - (in2 ? as0[1] : as0[0]);          // (modulo 2^32: out-of-range positions fail the bound below)
        const uint32_t xr = opp ? x0 - dd : x0 + dd;
        const bool inb = okv & have & (xr < ref_total);
        const uint32_t xs = inb ? xr : 0u;
        const uint32_t rp = refpay[xs];
        const uint32_t *rw = ref2 + (xs >> 4);
        const uint32_t g0 = rw[0], g1 = rw[1], g2 = rw[2];
        uint64_t x, y;
        windows(j, x, y);
        const uint32_t sg = (xs & 15u) << 1;
        const uint64_t W = ((uint64_t)__builtin_amdgcn_alignbit(g2, g1, sg) << 32) | __builtin_amdgcn_alignbit(g1, g0, sg);
        const bool eq = opp ? ((y & kmer_mask) == (~W & kmer_mask)) : ((x & kmer_mask) == (W & kmer_mask));
        const bool mm = inb & (rp != REFPAY_NONE) & eq;
        mt[j] = mm;
        slo[j] = rp;
        known |= (mm | !okv) ? (1u << j) : 0u;                      // (a slot that does not exist or is no valid k-mer needs no probe either)
        lane_any |= mm;
        n_match += (uint32_t)__builtin_popcountll(__ballot(mm));
        const uint64_t Uc = __ballot(okv & !mm);                    // open slots
        ub += cover(Uc, Uprev);
        Uprev = Uc;
      }
      ub += cover(0ull, Uprev);
#endif
      if (n_match < 4u) { lane_any = false; return false; }
#if SHK_ANCH_CUT == 3
      return n_match != 12345u;
#endif
      // (4)
      if (vote(IU{}, std::false_type{}, ub)) {
#ifdef SHK_ANCH_STATS
        if (lane == 0) atomicAdd(&H->out->counters[CTR_UNUSED3], 1u);
#endif
        return true;
      }
#if SHK_ANCH_CUT == 4
      return true;
#endif
      // (5)
#ifdef SHK_ANCH_STATS
      if (lane == 0) atomicAdd(&H->out->counters[CTR_UNUSED5], 1u);
#endif
#pragma unroll
      for (int j = 0; j < U; ++j)
        if (!((known >> j) & 1u)) { mt[j] = false; slo[j] = 0u; }
      probe_rounds(I0{}, IU{}, known);
      vote(IU{}, std::true_type{}, 0u);
      return true;
    };
    if constexpr (E < 0) {
      return anchored();
    } else {
    using IE = std::integral_constant<int, E>;
    // JA: the stop at which the early decision is tried (one round behind the usual first stop), and where the table modes
    // try the cut a second time
    constexpr int JA = JA_ROUNDS;
    using IA = std::integral_constant<int, JA>;
    // the bound cut at a stop behind the rounds [0, J): what is in the filter so far covers `cv` bases, the slots not probed
    // yet cover `ub`: together fewer than c * len -- no gene can reach the threshold (ReadAnalyzer.hpp:104), the read has no
    // association.  LDS modes stop only without any match (their matches are not validated yet, and a gene's k-mers are rare
    // among an off-target read's); the table modes -- large references, where a random k-mer IS in the filter every few dozen
    // slots -- count what the matches cover: the union of everything found first, then the largest coverage for ONE gene.
    auto ruled_out = [&](auto j_const, const uint32_t ub, const bool per_gene) -> bool {
      if (ub >= thr_r) return false;
      if (!__ballot(lane_any)) return true;
      if (!TOL) return false;
      if (found_cover(j_const) < thr_r - ub) return true;
      return per_gene && gene_cover(j_const) < thr_r - ub;
    };
    // the slot ss of the read as a probe of the exact LDS table (LX): is its k-mer in the filter?  (`want` false: no probe)
    auto lx_hit_at = [&](const uint32_t ss_in, const bool want, uint32_t &payload) -> bool {
      return want & lx_probe(fw, rv, want ? ss_in : 0u, payload);
    };
    // the first two rounds of a one-gene index in the sparse order (see spT above).  true: the read is settled.  false: mt / slo of
    // the rounds 0 and 1 are what probe_rounds would have left (matches validated), the read goes on behind the cut's first stop.
    // Returns 1: the read is settled.  0: not settled, mt / slo of the rounds 0 and 1 are what probe_rounds would have left (one-gene
    // indices).  2: not settled and nothing left behind -- the caller probes the first rounds in the usual order (indices of several
    // genes: a read whose matches are not all one gene's single-gene lists, or do not reach the threshold yet; rare).
    // SEVERAL GENES.  Every entry of the exact table carries its list's one gene (or the escape value: a list of several genes, a
    // gene beyond 13 bits).  Say every match of the rounds A and B -- the whole prefix [0, 128 - T) and the T tiles -- is a
    // single-gene list of ONE gene g, and what those matches cover (as the vote counts it) is cov.  Any other gene's k-mers can
    // then only sit in slots that were not probed, all at packed positions >= 128 - T, which cover at most spUb bases: its final
    // coverage is <= spUb, while g's is >= cov.  With cov >= c * len (thr_r) and cov > spUb, g is the read's only association
    // (ReadAnalyzer.hpp:90-108) whatever the other 140 probes would say: 128 probes instead of 192 and no vote -- the early
    // decision's argument (above), made one round earlier because the tiles bring mate 2's coverage forward.
    auto sparse_first = [&]() -> int {
      const uint32_t T = spT, nA = 64u - T, ln = (uint32_t)lane;
      // (the read's one association)
      auto write_the_gene = [&](const uint32_t g) {
        if (lane == 0 && !SHK_ABL(P, 64u)) {
          sp_count[read] = 1u;
          uint2 pk;
          pk.x = g & 0xFFFFu;
          pk.y = 0u;
          *reinterpret_cast<uint2 *>(sp_inl + (uint64_t)read * SHK_INLINE_IDS) = pk;
        }
      };
      const bool tile = ln >= nA;
      const uint32_t sA = tile ? spLast - (ln - nA) * k : 2u * ln;
      uint32_t pA = 0u, pB = 0u;
      SHK_STAMP(2);
      bool hA = lx_hit_at(sA, true, pA);
      uint64_t HA = __ballot(hA);
      SHK_STAMP(3);
      if (HA) {
        hA = hA && slot_valid(sA);   // (the slot has to exist and be a valid k-mer)
        HA = __ballot(hA);
      }
      if (HA && sp_one) {
        // bases covered by what matched: an even slot adds min(k, distance to the next even match), a tile k
        const uint64_t H1 = HA & ((1ull << nA) - 1ull);
        const uint64_t nx = (H1 >> ln) >> 1;
        const uint32_t step = nx ? 2u * ((uint32_t)__builtin_ctzll(nx) + 1u) : k;
        const uint32_t cov = wave_sum_u32((hA && !tile) ? (step < k ? step : k) : 0u) + k * (uint32_t)__builtin_popcountll(HA >> nA);
        if (cov >= thr_r) {
          write_the_gene(P.lx_gene);
          SHK_STAMP(4);
          return 1;
        }
      }
      SHK_STAMP(4);
      const uint32_t sB = ln < nA - 1u ? 2u * ln + 1u : ln + nA;
      bool hB = lx_hit_at(sB, true, pB);
      uint64_t HB = __ballot(hB);
      SHK_STAMP(5);
      if (HB) {
        hB = hB && slot_valid(sB);
        HB = __ballot(hB);
      }
      if (!(HA | HB) && spUb < thr_r) return 1;   // the bound cut: nothing of the prefix is in the filter
      bool one_g = sp_one;
      if (!sp_one) {
        // several genes.  A list of several genes among the matches (the escape value): the usual order, whose rounds ask the
        // position table for such lists.  Else: one gene's single-gene lists only?
        if (!(HA | HB)) return 2;
        if (__ballot((hA & (pA == LTAB_ESC)) | (hB & (pB == LTAB_ESC)))) return 2;
        const uint32_t g = HA ? (uint32_t)__builtin_amdgcn_readlane((int)pA, (int)__builtin_ctzll(HA))
                              : (uint32_t)__builtin_amdgcn_readlane((int)pB, (int)__builtin_ctzll(HB));
        one_g = __ballot((hA & (pA != g)) | (hB & (pB != g))) == 0ull;
      }
      // (a few per cent of the reads) the matches in the usual order: slot s was probed by lane s / 2 of round A or B (even / odd
      // s < 2 nA - 1) or by lane s - nA of B; the T slots in front of the stop that the tiles displaced are not probed yet
      auto settle_round = [&](const int j, const uint32_t v) {
        const uint32_t sl = ln + 64u * (uint32_t)j;
        const bool eo = sl < 2u * nA - 1u, inB = sl < 128u - T;
        const uint32_t src = eo ? sl >> 1 : (inB ? sl - nA : sl - (128u - T));
        const uint32_t bit = eo ? sl & 1u : (inB ? 1u : 2u);
        const uint32_t pv = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)v);
        mt[j] = ((pv >> bit) & 1u) != 0u;
        // (several genes: the matched entry's gene travels with its bit -- A's in bits 3..15, B's in 16..28, F's in a word of its own)
        slo[j] = sp_one ? P.lx_gene : ((bit == 0u ? pv >> 3 : pv >> 16) & LTAB_ESC);
        okm[j] = mt[j] ? 0xFFFFFFFFu : 0u;
      };
      const uint32_t vAB = (hA ? 1u : 0u) | (hB ? 2u : 0u) | (sp_one ? 0u : ((pA << 3) | (pB << 16)));
      settle_round(0, vAB);
      settle_round(1, vAB);
      if (HA | HB) {
        // the whole prefix is known now: what it covers (as the vote counts it) + the tiles -- a read with more errors than the
        // even slots forgive is settled here, behind the two rounds an off-target read costs
        const uint64_t H0 = __ballot(mt[0]), H1 = __ballot(mt[1]);
        const uint32_t cov = cover(H0, 0ull) + cover(H1, H0) + cover(0ull, H1) + k * (uint32_t)__builtin_popcountll(HA >> nA);
        if (one_g && cov >= thr_r && (sp_one || cov > spUb)) {
          write_the_gene(sp_one ? P.lx_gene : (uint32_t)__builtin_amdgcn_readfirstlane((int)(HA ? __builtin_amdgcn_readlane((int)pA, (int)__builtin_ctzll(HA))
                                                                                                   : __builtin_amdgcn_readlane((int)pB, (int)__builtin_ctzll(HB)))));
          return 1;
        }
      }
      // ... then the displaced slots, and on behind the cut's first stop as ever
      uint32_t pF = 0u;
      const bool hF = lx_hit_at(128u - T + ln, ln < T, pF) && slot_valid(ln < T ? 128u - T + ln : 0u);
      if (!sp_one && __ballot(hF & (pF == LTAB_ESC))) return 2;
      settle_round(1, vAB | (hF ? 4u : 0u));
      if (!sp_one) {
        // (the displaced slots' genes: slot sl >= 128 - T of round 1 was probed by lane sl - (128 - T))
        const uint32_t sl = ln + 64u;
        const uint32_t pvF = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((sl >= 128u - T ? sl - (128u - T) : 0u) << 2), (int)pF);
        if (sl >= 128u - T) slo[1] = pvF & LTAB_ESC;
      }
      lane_any = mt[0] | mt[1];
      return 0;
    };
    SHK_STAMP(12);
    if constexpr (JA >= U) {
      // nothing to decide early (two rounds): the cut's first stop at most
      if constexpr (E < U) {
        probe_rounds(I0{}, IE{}, 0u);
        if (ruled_out(IE{}, cutUb, false)) return true;
        probe_rounds(std::integral_constant<int, (E < U ? E : 0)>{}, IU{}, 0u);
      } else {
        if (!probe_rounds(I0{}, IU{}, 0u)) return true;
      }
    } else {
      if constexpr (E < JA) {
        bool first_done = false;
        if constexpr (SPARSE && E == 2) {
          if (spT) {
            const int sf = sparse_first();
            if (sf == 1) return true;
            first_done = sf == 0;
          }
        }
        if (!first_done) probe_rounds(I0{}, IE{}, 0u);
        if (ruled_out(IE{}, cutUb, false)) return true;
        constexpr int EJ = E < JA ? E : 0;   // (= E: this branch)
        using IEJ = std::integral_constant<int, EJ>;
        bool whole = true;
        if constexpr (TOL && ANCH && !SHK_NO_PARTIAL) {
          // Table modes (large references): the second stop rules a read out when no gene's matches cover c * len less what is
          // still unprobed -- and the chance matches of an off-target read are isolated k-mers of different genes, k bases each.
          // So the round between the stops is probed only as far as that argument needs: up to the first slot X behind which
          // the rest covers at most c * len - (SHK_PART_MARGIN_K k + 1) bases (2 x 150 bp, k = 17, c = 0.6: 11 of the round's 46
          // slots).  Ruled out there: the other 35 memory requests are never made; else the rest of the round follows.
          const uint32_t margin = (uint32_t)SHK_PART_MARGIN_K * k + 1u;
          if (thr_r > margin) {
            const uint32_t B = thr_r - margin;
            const uint32_t l1 = nk1 ? nk1 + k - 1u : 0u, l2 = nk2 ? nk2 + k - 1u : 0u;
            const uint32_t X = slot_for_ub(B, l1, l2);   // the smallest slot with at most B bases behind it
            const uint32_t lo = 64u * (uint32_t)EJ, hi = 64u * (uint32_t)JA;
            // slots of the round at or behind X that exist (the ones the partial round leaves out)
            const uint32_t e1 = nk1 > X ? (nk1 < hi ? nk1 : hi) - (X > lo ? X : lo) : 0u;
            const uint32_t b2 = X > P2 ? X : P2, t2 = P2 + nk2 < hi ? P2 + nk2 : hi;
            const uint32_t left_out = (X < hi ? e1 : 0u) + ((X < hi && t2 > b2 && b2 >= lo) ? t2 - b2 : 0u);
            if (X > lo && X < hi && left_out >= 8u) {
              const uint32_t skip = ((uint32_t)lane + lo >= X) ? (1u << EJ) : 0u;
              mt[EJ] = false;
              slo[EJ] = 0u;
              probe_rounds(IEJ{}, IA{}, skip);
              if (ruled_out(IA{}, bases_behind(X, nk1, nk2, P2, l1, l2), true)) return true;
              probe_rounds(IEJ{}, IA{}, skip ^ (1u << EJ));
              whole = false;
            }
          }
        }
        if (whole) probe_rounds(IEJ{}, IA{}, 0u);
        if (ruled_out(IA{}, ubJA, true)) return true;
      } else {
        // (E == JA: the cut's first stop.  E == U: no stop was planned -- c is small, or the index sits behind the L2 summary,
        //  where a stop of its own cost more than it saved -- but this one exists anyway, so the cut is tried at it)
        bool first_done = false;
        if constexpr (SPARSE && E == 2 && JA == 2) {
          if (spT) {
            const int sf = sparse_first();
            if (sf == 1) return true;
            first_done = sf == 0;
          }
        }
        if (!first_done) probe_rounds(I0{}, IA{}, 0u);
        if (ruled_out(IA{}, ubJA, true)) return true;
      }
      if (vote(IA{}, std::false_type{}, ubJA)) return true;
      probe_rounds(std::integral_constant<int, (JA < U ? JA : 0)>{}, IU{}, 0u);
    }
    vote(IU{}, std::true_type{}, 0u);
    return true;
    }
    };   // classify_staged
    bool settled = false;
    if constexpr (ANCH) settled = classify_staged(std::integral_constant<int, -1>{});
    if (!settled) {
      // the first-round counts this specialisation is compiled for (CutPlan<U>): cutE is one of them, or U
      if (CUT && cutE == (uint32_t)CutPlan<U>::E0) classify_staged(std::integral_constant<int, (CUT ? CutPlan<U>::E0 : U)>{});
      else if (CUT && CutPlan<U>::E1 != CutPlan<U>::E0 && cutE == (uint32_t)CutPlan<U>::E1) classify_staged(std::integral_constant<int, (CUT ? CutPlan<U>::E1 : U)>{});
      else classify_staged(std::integral_constant<int, U>{});
    }
    };   // per_pair
    per_pair(std::integral_constant<uint32_t, 0u>{});
    SHK_STAMP(6);
    if constexpr (TRI) {
      per_pair(std::integral_constant<uint32_t, 1u>{});
      SHK_STAMP(6);
      per_pair(std::integral_constant<uint32_t, 2u>{});
      SHK_STAMP(6);
    }
    }   // !skip
#if SHK_STAMPS
    if constexpr (TRI) {
      if (!have_nxt) {
        SHK_STAMP(7);
        st_acc[13] = st_last - st_t0;
        const uint64_t st_r1 = __builtin_amdgcn_s_memrealtime();
        st_acc[14] = st_r1 - st_r0;
        if (lane == 0 && blockIdx.x * WAVES + wave < 4096u) {
          shk_stamp_waves[2u * (blockIdx.x * WAVES + wave)] = st_r0;
          shk_stamp_waves[2u * (blockIdx.x * WAVES + wave) + 1u] = st_r1;
        }
        if (lane == 0 && blockIdx.x * WAVES + wave < 4096u)
          for (int i = 0; i < 16; ++i) shk_stamp_acc[16u * (blockIdx.x * WAVES + wave) + i] += (unsigned long long)st_acc[i];
      }
    }
#endif
    if (!have_nxt) break;
    if (TRI) { tri_retire(t_nxt); t_cur = t_nxt; if (TRO) { tlr_cur = tlr_nxt; to_cur = to_nxt; } }
    retire(w_nxt, q_nxt);
    if (!UNI) {
      retire_meta(r_nn);
      m_cur = m_nxt;
      m_nxt = meta_finish(r_nn);
      pl_cur = pl_nxt;
    }
    it = nxt;
    if (PRE_R) done_cur = done_nxt;
    if (DYN && !BM) {
      if (TRO) { dyn_q = dyn_q2; dyn_q2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)dyn_take); }
      else dyn_q = (uint32_t)__builtin_amdgcn_readfirstlane((int)dyn_take);
    }
    read = CLS ? read_nxt : it;
#pragma unroll
    for (int g = 0; g < G; ++g) { w_cur[g] = w_nxt[g]; q_cur[g] = q_nxt[g]; }
    SHK_STAMP(7);
  }
  if (!CLS) break;
  // the next class of the share, or the wave's next share
  if (ent_end < share_end) {
    ++cls_j;
  } else {
    if (DYNC) {
      uint32_t t = 0u;
      if (lane == 0) t = atomicAdd(dyn_ctr, 1u);
      t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
      const uint64_t sh = (uint64_t)(t / (uint32_t)WAVES) * n_waves + (blockIdx.x * WAVES + t % (uint32_t)WAVES);
      if (sh >= CLS_SHARES) break;
      share = (uint32_t)sh;
    } else {
      share += n_waves;
    }
    if (share >= CLS_SHARES || (uint64_t)share * share_len >= P.n) break;
    ent_end = share * share_len;
    share_end = (uint64_t)ent_end + share_len < P.n ? ent_end + share_len : (uint32_t)P.n;
    cls_j = (uint32_t)__builtin_amdgcn_readfirstlane((int)P.cls_share_first[share]);
  }
  }   // segments
}

// rmode: 0 = the ragged instantiation, 1 = the uniform one, 2 = by classes (CLS: exact-table instantiations only), 3 = mixed lengths by offsets (TRO)
template <int U>
static void launch_uni_u(const ClassifyParams &p, int mode, bool hasq, bool big, bool lx, int rmode, unsigned grid, hipStream_t s)
{
  if (rmode == 2) {
    if constexpr (U <= 5 || U == 10) {
      if (lx && mode == PM_LDS_TAB) {
        if (hasq) hipLaunchKernelGGL((classify_uni_kernel<U, PM_LDS_TAB, true, 21, true, true>), dim3(grid), dim3(UniGeom<U, PM_LDS_TAB, 21>::THREADS), 0, s, p);
        else hipLaunchKernelGGL((classify_uni_kernel<U, PM_LDS_TAB, false, 21, true, true>), dim3(grid), dim3(UniGeom<U, PM_LDS_TAB, 21>::THREADS), 0, s, p);
      }
    }
    return;
  }
  // rmode 3: a batch of mixed lengths through the three-pairs kernel by offsets (TRO): exact table in LDS, no qualities, U = 3 ... 5
  if (rmode == 3) {
    if constexpr (U >= 3 && U <= 5) {
      if (lx && mode == PM_LDS_TAB && !hasq) {
        if (p.lx_multi) hipLaunchKernelGGL((classify_uni_kernel<U, PM_LDS_TAB, false, 21, true, false, true, true, false, true>), dim3(grid), dim3(UniGeom<U, PM_LDS_TAB, 21>::THREADS), 0, s, p);
        else if (p.tile_first) hipLaunchKernelGGL((classify_uni_kernel<U, PM_LDS_TAB, false, 21, true, false, false, true, true, true>), dim3(grid), dim3(UniGeom<U, PM_LDS_TAB, 21>::THREADS), 0, s, p);
        else hipLaunchKernelGGL((classify_uni_kernel<U, PM_LDS_TAB, false, 21, true, false, false, true, false, true>), dim3(grid), dim3(UniGeom<U, PM_LDS_TAB, 21>::THREADS), 0, s, p);
      }
    }
    return;
  }
  const bool uni = rmode == 1;
  // the exact table in LDS, uniform batches without qualities: three pairs per staging pass where the lengths allow (TRI) -- decided
  // here when the host knows the lengths, else both instantiations are launched and the device's verdict picks (p.tri)
  if constexpr (U >= 3 && U <= 5) {
    if (uni && lx && mode == PM_LDS_TAB && !hasq && p.tri) {
      const bool host_knows = p.uni_flag == nullptr;
      const bool applies = host_knows && tri_applies(p.uni_L1, p.uni_L2, p.k, 64u * U);
      if (!host_knows || applies) {
        if (p.lx_multi) hipLaunchKernelGGL((classify_uni_kernel<U, PM_LDS_TAB, false, 21, true, false, true, true>), dim3(grid), dim3(UniGeom<U, PM_LDS_TAB, 21>::THREADS), 0, s, p);
        else if (p.tile_first) hipLaunchKernelGGL((classify_uni_kernel<U, PM_LDS_TAB, false, 21, true, false, false, true, true>), dim3(grid), dim3(UniGeom<U, PM_LDS_TAB, 21>::THREADS), 0, s, p);
        else hipLaunchKernelGGL((classify_uni_kernel<U, PM_LDS_TAB, false, 21, true, false, false, true>), dim3(grid), dim3(UniGeom<U, PM_LDS_TAB, 21>::THREADS), 0, s, p);
        if (applies) return;
      }
    }
  }
  if (uni && lx && mode == PM_LDS_TAB && p.lx_multi) {
    if constexpr (U <= 5 || U == 10) {
      if (hasq) hipLaunchKernelGGL((classify_uni_kernel<U, PM_LDS_TAB, true, 21, true, false, true>), dim3(grid), dim3(UniGeom<U, PM_LDS_TAB, 21>::THREADS), 0, s, p);
      else hipLaunchKernelGGL((classify_uni_kernel<U, PM_LDS_TAB, false, 21, true, false, true>), dim3(grid), dim3(UniGeom<U, PM_LDS_TAB, 21>::THREADS), 0, s, p);
    }
    return;
  }
#define LU4(M_, L_, HQ_, UN_) hipLaunchKernelGGL((classify_uni_kernel<U, M_, HQ_, L_, UN_>), dim3(grid), dim3(UniGeom<U, M_, L_>::THREADS), 0, s, p)
#define LU(M_, L_) do { if (uni) { if (hasq) LU4(M_, L_, true, true); else LU4(M_, L_, false, true); } \
                        else if (L_ != 20) { if (hasq) LU4(M_, L_, true, false); else LU4(M_, L_, false, false); } } while (0)
  switch (mode) {
  case PM_LDS_TAB:
    if (lx) { if constexpr (U <= 5 || U == 10) LU(PM_LDS_TAB, 21); }   // (launch_classify_uni asks for it only where it is compiled)
    else if (big) LU(PM_LDS_TAB, 20);
    else LU(PM_LDS_TAB, 18);
    break;
  case PM_LDS_TAB_MOD: if (big) LU(PM_LDS_TAB_MOD, 20); else LU(PM_LDS_TAB_MOD, 18); break;
  case PM_TAB: LU(PM_TAB, 18); break;
  case PM_KTAB: LU(PM_KTAB, 18); break;
  case PM_TAB_MOD: LU(PM_TAB_MOD, 18); break;
  default: LU(PM_TAB_SUM, 18); break;
  }
#undef LU
#undef LU4
}

}  // namespace shk
