// classify_uni.hpp -- classify_uni_kernel and its launcher template.  Included by classify_uni_u<U>.hip (one translation unit
// per unroll U: the instantiations of one U take about half a minute to compile, and there are six).
//
// The kernel is ONE function -- its state lives in the registers of one wave, and every step is a lambda over that state, instantiated
// per compile-time plan --, so its body is split by PHASE into fragments that are included where they stand, not into functions:
//   this file                    configuration macros, geometry constants (UniGeom), the kernel's head (parameters, LDS layout), the
//                                loop over reads / triples / classes, the order of a pair's steps (per_pair, classify_staged), the launchers
//   classify_uni_plan.inc        a read's geometry in slots and staging groups; the plans of the bound cut and of the sparse first round
//   classify_uni_loads.inc       how a wave gets its reads (prefetch, three pairs per pass, by offsets, by class entries, turns, blocks)
//   classify_uni_staging.inc     bases -> 2-bit code streams + validity in LDS
//   classify_uni_tiles.inc       the exact LDS table's probe; the tiles' round over three staged pairs
//   classify_uni_rounds.inc      the rounds: k-mer windows, canonical form, XXH64, summary and table probes
//   classify_uni_vote.inc        coverage, the vote, the early decision, the result write
//   classify_uni_anchored.inc    the anchored extension (table modes)
//   classify_uni_sparse.inc      the bound cut at a stop; the sparse first rounds of indices held in LDS
// (The split is textual: the seven objects built from it are byte-identical to those of the single file.)
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

#include "classify_uni_plan.inc"
#include "classify_uni_loads.inc"
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
#include "classify_uni_staging.inc"
#include "classify_uni_tiles.inc"
    // (TRI: the three pairs of the triple one after the other, each in its own staging area)
    // (a generic lambda called once per pair, not a loop: with the pair's area a compile-time offset from the wave's, the LDS
    //  addresses of a pair's windows stay what they are for one area -- the lane's offset from a loop-invariant base, the area as the
    //  instruction's immediate -- instead of an addition per address and pair)
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
#include "classify_uni_rounds.inc"
#include "classify_uni_vote.inc"
#include "classify_uni_anchored.inc"
    if constexpr (E < 0) {
      return anchored();
    } else {
    using IE = std::integral_constant<int, E>;
    // JA: the stop at which the early decision is tried (one round behind the usual first stop), and where the table modes
    // try the cut a second time
    constexpr int JA = JA_ROUNDS;
    using IA = std::integral_constant<int, JA>;
#include "classify_uni_sparse.inc"
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
