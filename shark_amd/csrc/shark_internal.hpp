// shark_internal.hpp -- context layout and kernel launch entry points shared
// by the translation units of libsharkhip.  Not part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/shark_hip.h"
#include "lds_table.hpp"   // LTAB_*: the LDS-resident exact table of a tiny index

namespace shk {

// ---- device-resident index (immutable after shk_ref_finalize) -------------
// HBM layout:
//   bf64     : the Bloom filter, sdsl::bit_vector layout (LSB-first 64-bit
//              words), zero-padded to a multiple of 512 bits.
//   rank_w   : uint32 ones-before-word directory, one entry per 64-bit word
//              (+1 sentinel = n_set): rank(pos) = rank_w[pos>>6] +
//              popcount(word & below(pos)) -- the probed word is all it needs.
//   ent      : one 8-byte entry per set bit r (r = rank): {start, len, first
//              gene} of its gene list; ids[start .. start+len) is the list.
//              (+1 sentinel {tot_idx,0,0}).  Explicit form of
//              _bv/_select_bv/_index_kmer (bloomfilter.h:142-167).
//   ids      : uint16 gene indices, ascending and unique inside a list.
//   sum32    : optional summary level: bit j = OR of filter bits
//              [j<<sum_shift, (j+1)<<sum_shift).  A clear summary bit proves
//              the filter bit clear, so the result of every probe is
//              unchanged; sized to stay in L2 (sparse filters) or in the
//              Infinity Cache.  Only for power-of-two filter sizes.
//   tab      : optional POSITION TABLE (power-of-two filter sizes): an exact
//              sparse encoding of the filter's set bits -- bit `pos` is set iff
//              `pos` is a key of the table -- with the list entry inline, so a
//              probe answers "is the bit set" AND "which list" with one
//              16-byte load instead of filter word + rank word + entry.
//              Buckets of two 8-byte slots, linear probing over buckets, home
//              bucket = pos & (n_buckets-1) (pos is already a hash).  Slot:
//                high word: [31:8] tag = pos >> tab_lg   [7] valid   [5:0] displacement (buckets from home)
//                low word : [31] multi   [30:0] gene (single-gene list) or rank r (multi: ent[r])
//              The filter itself stays in HBM as the ground truth (it defines
//              rank, is exported, and is what the tests compare bit for bit).
constexpr uint32_t LDS_SUM_LOG2 = 18;                 // 2^18 bits = 32 KiB of LDS per workgroup
constexpr uint32_t LDS_SUM_BITS = 1u << LDS_SUM_LOG2;

struct ListEntry {
  uint32_t start;
  uint16_t len;     // clipped at 0xFFFF: then the true end is the next entry's start
  uint16_t gene0;   // first gene of the list
};

struct DeviceIndex {
  uint64_t *bf64 = nullptr;
  uint64_t bf_bits = 0;
  uint64_t bf_words64 = 0;   // padded to whole 512-bit blocks
  bool pow2 = false;
  uint32_t *rank_w = nullptr;
  ListEntry *ent = nullptr;
  uint16_t *ids = nullptr;
  uint32_t *sum32 = nullptr;
  uint32_t sum_shift = 0;    // 0 = no summary level
  uint64_t sum_bits = 0;
  uint32_t *lsum32 = nullptr; // 2^18-bit summary staged into LDS by the table kernel (small indices only)
  uint32_t lsum_shift = 0;    // 0 = not used
  uint32_t *lbig32 = nullptr; // 2^20-bit summary (128 KiB of LDS, one 1024-thread workgroup per CU): classify_uni_kernel on
  uint32_t lbig_shift = 0;    //   indices too dense for the 2^18-bit one; 0 = not built
  double lbig_pass = 0.0;     //   the share of random k-mers that pass it
  uint32_t *ltab = nullptr;   // LDS-resident EXACT table of a tiny index (LTAB_BYTES image: T[2^15] then D[2^13]; lds_table.hpp)
  uint32_t ltab_mul = 0;      //   the multiplier its slots were computed with
  uint32_t ltab_gene = 0xFFFFFFFFu;   //   the ONE gene every key of that table answers with (a one-gene index), else 0xFFFFFFFF
  bool ltab_sparse = false;           //   the sparse first rounds are allowed on it (not SHK_NO_SPARSE=1 at build time)
  uint64_t *tab = nullptr;   // 2 slots per bucket
  uint32_t tab_lg = 0;       // log2(number of buckets); 0 = no table
  bool tab_with_summary = false;
  uint64_t n_set = 0;
  uint64_t tot_idx = 0;
  bool wrap = false;         // more than 65 536 genes: lists sorted by 16-bit id WITH duplicates (index_build.hip), WRAP kernels
  // ---- the reference itself, for the anchored extension of the table kernels (classify_uni.hpp) ----
  //   ref2   : all records back to back as 2-bit codes, base p in bits [2 (p & 15), +2) of dword p >> 4 (LSB first, invalid
  //            characters as 0), padded by four dwords
  //   refpay : per base position x the low word of the position-table slot of the k-mer STARTING at x (multi | payload; what a
  //            probe of that k-mer returns), REFPAY_NONE where no valid k-mer starts (record end, invalid character)
  //   atab   : the position table a second time with, in place of a slot's low word (its list), one occurrence of its key in the
  //            reference: x | strand << 31, strand = 1 when the k-mer at x is stored reverse-complemented (its canonical form is the
  //            reverse complement); 0xFFFFFFFF = unset.  Same buckets, same high words: the anchored extension's sample probes THIS
  //            table, and the bucket that says "the k-mer is in the index" says where in the reference in the same memory round trip
  //            (round 3 kept the occurrences in an array of their own, indexed by the slot that had matched: a dependent load)
  //   refext : per base position x where a valid k-mer starts: g | left << 16 | right << 24 -- the gene of x's record and how many
  //            positions directly in front of / behind x start a valid k-mer too (clipped at REFEXT_CLIP; such a run stays inside its
  //            record); REFEXT_NONE elsewhere.  refmul: one bit per position, 1 = the list under its k-mer is not a single-gene
  //            list.  What anchor_verdict_kernel reads instead of a payload per slot (anchor_verdict.hip)
  uint32_t *ref2 = nullptr, *refpay = nullptr, *refext = nullptr, *refmul = nullptr;
  uint64_t *atab = nullptr;
  uint32_t ref_total = 0;    // bases in ref2 / entries in refpay; 0 = not built
  // ---- the index a third time, keyed by the K-MER and bucketed by its MINIMISER (k = 15 ... 17, tables beyond the caches) ----
  //   ktab : 2^ktab_lg buckets of two 8-byte slots; eight consecutive buckets are one 128-byte LINE.  Keys: every canonical k-mer
  //          whose filter bit is set -- the reference's k-mers AND the filter's false positives, enumerated over all 4^k / 2
  //          canonical k-mers when the index is built --, so "is the key there" is exactly "is the bit set" (bloomfilter.h:87-89).
  //          line = hash of the k-mer's minimiser (the smallest hash among its k - w + 1 canonical w-mers), bucket within the
  //          line = the key's low three bits: consecutive k-mers of a read share their minimiser, hence their line -- one memory-side
  //          request serves several probes.  slot: low word as in `tab` (what a probe of the position returns), high word
  //          (key >> 3) << 1 | 1; a key that does not fit its bucket goes to the same bucket of the next line (so the bucket a slot
  //          sits in always names the key's low bits), marked as in `tab`.
  uint64_t *ktab = nullptr;
  uint32_t ktab_lg = 0;      // log2(buckets); 0 = not built
  uint32_t ktab_w = 0;       // w (the minimiser's length)
  uint64_t ktab_keys = 0;    // keys it holds (reference k-mers + false positives of the filter)
};
constexpr uint32_t REFEXT_NONE = 0xFFFFFFFFu, REFEXT_CLIP = 254u;   // (an extent never reads 255: no entry looks like REFEXT_NONE)
constexpr uint32_t REFPAY_NONE = 0xFFFFFFFFu;   // (multi with payload 2^30-1: not a rank, n_set <= 2^30-1 entries have ranks below that)

// per-wave staging area of a read for a slot capacity S (layout: classify.hip)
__host__ __device__ constexpr uint32_t stage_cap_bases(uint32_t S) { return S + 96; }                  // S = slot capacity, multiple of 64
__host__ __device__ constexpr uint32_t code_dwords_for(uint32_t S) { return stage_cap_bases(S) / 16 + 2; }   // one stream (even)
__host__ __device__ constexpr uint32_t vbit_words_for(uint32_t S) { return stage_cap_bases(S) / 64 + 2; }
__host__ __device__ constexpr uint32_t stage_words_for(uint32_t S) { return code_dwords_for(S) + vbit_words_for(S); }   // 64-bit words: fw + rv + validity

// result / queue pointers of a classify launch.  They live in device memory behind ONE pointer so
// that the hot loop does not pin ~14 SGPRs for values it needs once per read (the fast kernel is at
// the 106-SGPR limit and spills wave-uniform values into VGPRs otherwise).
struct ClassifyOut {
  uint32_t *count;           // n : number of genes kept (exact)
  uint16_t *inl;             // n * SHK_INLINE_IDS : first genes
  uint32_t *counters;        // see CTR_*
  uint32_t *long_queue;      // read indices that did not fit the fast kernel
  uint32_t *tie_queue;       // 3 words per entry: read index, best cov, best nk
};

// ---- classify kernel parameters (passed by value) --------------------------
struct ClassifyParams {
  // index
  const uint64_t *bf64;
  const uint32_t *rank_w;
  const ListEntry *ent;
  const uint16_t *ids;
  const uint32_t *sum32;
  uint32_t sum_shift;
  const uint64_t *tab;
  uint32_t tab_lg;
  uint32_t tab_nt;           // 1 = table far larger than the caches: probe it with non-temporal loads
  // non power-of-two filter sizes: bits = mod_m << mod_shift; mod_fast = the 32-bit direct remainder applies
  uint32_t mod_fast, mod_shift, mod_m;
  uint64_t mod_c;
  const uint32_t *lsum32;    // LDS_SUM_BITS-bit summary (global copy), staged into LDS per workgroup
  uint32_t lsum_shift;
  uint32_t lx_gene;          // exact table in LDS (LSL = 21): the gene of a one-gene index (DeviceIndex::ltab_gene), else 0xFFFFFFFF
  uint32_t tro;              // 1 = the three-pairs kernel by offsets (classify_uni.hpp, TRO) is in the stream: uniform_check_kernel's verdict 3 for a batch of mixed lengths its layout holds
  uint32_t tri;              // 1 = the three-pairs-per-pass instantiation (classify_uni.hpp, TRI) is launched beside the ordinary uniform one: the one whose lengths qualify works
  uint32_t tile_first;       // (with tri, one-gene index) 1 = a round of disjoint k-mers for the three staged pairs together in front of the pairs' own rounds (classify_uni.hpp, TF)
  uint32_t lx_multi;         // exact table in LDS of an index of SEVERAL genes: the sparse first rounds with the early decision's argument (classify_uni.hpp)
  uint64_t bf_bits;
  uint64_t bf_mask;
  // options
  uint32_t k;
  uint32_t hasq;     // 0 = no masking (the reference's `char min_quality` is 0, FastqSplitter.hpp:52)
  int32_t mq;        // (signed char)(min_quality + 33), FastqSplitter.hpp:70 -- wraps for -q > 94 exactly as the reference's char does
  int32_t single;
  double c;
  // batch
  uint64_t n;
  const uint8_t *seq1;
  const uint64_t *off1;
  const uint8_t *seq2;
  const uint64_t *off2;
  const uint8_t *qual1;
  const uint8_t *qual2;
  // anchored extension (table modes): DeviceIndex::ref2 / refpay / atab; ref_total = 0: not available
  const uint32_t *ref2;
  const uint32_t *refpay;
  const uint64_t *atab;
  uint32_t ref_total;
  const uint32_t *refext;    // DeviceIndex::refext / refmul; nullptr: no anchor_verdict_kernel
  const uint32_t *refmul;
  uint32_t pre_verdict;      // 1 = anchor_verdict_kernel ran in front of this launch: a read whose count[] is set has its result
  uint32_t av_passes;        // anchor_verdict_kernel: consecutive passes a wave takes (set by its launcher from the batch's size)
  // the k-mer keyed, minimiser-bucketed table (DeviceIndex::ktab; classify_uni_kernel's PM_KTAB instantiations)
  const uint64_t *ktab;
  uint32_t ktab_lg, ktab_w;
  uint32_t ktab_nt;          // 1 = probe it with non-temporal loads
  // batches whose reads all have one length per mate (the usual sequencer output) take classify_uni_kernel's UNI instantiation,
  // the others its ragged one, which stages every read in the layout of the batch's LONGEST mates:
  // uni_flag == nullptr: the host knows (uni_L1, uni_L2) = the one length per mate, or the longest; else {1 = uniform and fits,
  // L1, L2 (the same)} written by uniform_check_kernel
  uint32_t uni_L1, uni_L2;
  const uint32_t *uni_flag;
  // ragged batches (the lengths above are then the LONGEST mates): per-launch table of read plans indexed by (l1, l2), 16 bytes each,
  // cleared by the host and filled by the kernel itself (classify_uni.hpp); nullptr / 0 = every read computes its plan
  uint4 *plan_tab;
  uint32_t plan_cap;               // entries
  // batches taken class by class (uni_flag[0] == 2; classify_uni_kernel's CLS instantiation): the pairs sorted by their two lengths,
  // 2 x uint4 per entry {o1, o2 | read index, "unguarded loads are safe", -, -}; cls_list: the non-empty classes in that order
  // {l1, l2, first entry, entries}; cls_share_first[s]: the class that entry s * ceil(n / CLS_SHARES) belongs to;
  // cls_hist: 2 x cls_cap words (pairs per class | the scatter's cursors)
  uint4 *cls_entries;
  uint4 *cls_list;
  uint32_t *cls_share_first;
  uint32_t *cls_hist;
  uint32_t cls_cap;                // classes the histogram holds: a batch with (L1 + 1) (L2 + 1) beyond that stays ragged
  uint32_t cls_min_fill;           // pairs per non-empty class the batch needs on average
  // per-read results and queues (device copy of ClassifyOut)
  const ClassifyOut *out;
  unsigned long long *gene_counts;  // 65536 (general kernel, EMIT mode)
  // work list for the general kernel (nullptr => all reads 0..n)
  const uint32_t *work;
  uint64_t n_work;
  const uint32_t *work_count;      // non-null: the number of work items is read from the device (tie queue) instead of n_work
  const uint32_t *flags;           // the launch's counters (CTR_OVERFLOW is honoured by the kernels that write gene_ids)
  // general-kernel scratch (per wave)
  uint64_t *scratch;
  uint64_t scratch_stride_words;   // per wave
  uint32_t scratch_slots;          // slot capacity per wave
  // emit mode (general kernel): write every gene whose (cov,nk) equals the
  // recorded best to gene_ids[gene_off[read] + i]
  const uint32_t *gene_off;
  uint16_t *gene_ids;
  // work counters (count_work build only)
  unsigned long long *work_counters;
  // timing-only ablation switches (env SHK_ABLATE; results are wrong when set; never set in tests/bench)
  uint32_t ablate;
};

enum {
  CTR_LONG = 0,      // entries in long_queue
  CTR_TIE = 1,       // entries in tie_queue
  CTR_OVERFLOW = 2,  // 1 = the batch has more associations than gene_ids holds (finalize_total_kernel)
  CTR_UNUSED3 = 3,
  CTR_MAX_SLOTS = 4, // max k-mer slots over queued long reads
  CTR_UNUSED5 = 5,
  CTR_VOUCH_BAD = 5, // 1 = the caller of shk_classify_device_submit vouched for read lengths that the batch's offsets do not have (vouch_check_kernel)
  CTR_ASSOC_LO = 6,  // number of associations of the batch (the scan's total), 64 bits
  CTR_ASSOC_HI = 7,
  CTR_VERDICT = 8,   // (host copy only) 1 + uni_flag[0] of a batch the device looked at: 1 ragged, 2 uniform, 3 by classes; 0: the host knew
  CTR_WORDS = 9
};

constexpr uint32_t UNI_FLAG_WORDS = 16;
constexpr uint32_t COUNTER_BLOCK_WORDS = 16;   // CTR_WORDS rounded up: Slot::d_uni_flag = d_counters + this
static_assert(CTR_WORDS <= COUNTER_BLOCK_WORDS, "the counters and the uniformity words share one allocation");
constexpr uint32_t CLS_SHARES = 65536;    // equal shares of a sorted batch: sixteen per wave of the exact-table kernels' grid, taken in turns (classify_uni.hpp, DYN)
constexpr uint32_t CLS_MIN_FILL = 16;     // by classes only if a non-empty class holds that many pairs on average

// ---- one batch in flight ---------------------------------------------------------------------
constexpr int PIPE_DEPTH = 3;      // batches shk_classify_submit keeps in flight per context

struct Slot {
  // device copies of a host batch (host-buffer entry points only)
  uint8_t *d_seq1 = nullptr, *d_seq2 = nullptr, *d_qual1 = nullptr, *d_qual2 = nullptr;
  uint64_t *d_off1 = nullptr, *d_off2 = nullptr;
  size_t cap_seq1 = 0, cap_seq2 = 0, cap_qual1 = 0, cap_qual2 = 0, cap_off1 = 0, cap_off2 = 0;
  // device results
  uint32_t *d_count = nullptr;    size_t cap_count = 0;
  uint16_t *d_inl = nullptr;      size_t cap_inl = 0;
  uint32_t *d_gene_off = nullptr; size_t cap_gene_off = 0;
  uint16_t *d_gene_ids = nullptr; size_t cap_gene_ids = 0;
  uint32_t *d_long_queue = nullptr; size_t cap_long_queue = 0;
  uint32_t *d_tie_queue = nullptr;  size_t cap_tie_queue = 0;
  uint32_t *d_counters = nullptr;
  uint32_t *d_uni_flag = nullptr;  // UNI_FLAG_WORDS words: verdict of uniform_check_kernel / class_plan_kernel (0 ragged, 1 uniform, 2 by classes), lengths, scratch
  uint4 *d_plan = nullptr; size_t cap_plan = 0;   // read plans of a ragged batch (ClassifyParams::plan_tab)
  // a batch taken class by class (ClassifyParams::cls_*)
  uint4 *d_cls_entries = nullptr; size_t cap_cls_entries = 0;
  uint4 *d_cls_list = nullptr;    size_t cap_cls_list = 0;
  uint32_t *d_cls_share = nullptr;
  uint32_t *d_cls_hist = nullptr; size_t cap_cls_hist = 0;
  uint64_t *d_scan_temp = nullptr; size_t cap_scan_temp = 0;
  ClassifyOut *d_out = nullptr;
  ClassifyOut out_shadow{};        // what *d_out holds (rewritten only when a buffer moved)
  // pinned host: counters, result offsets and ids
  uint32_t *h_counters = nullptr;  // CTR_WORDS
  uint32_t *h_gene_off = nullptr;  size_t cap_h_gene_off = 0;
  uint16_t *h_gene_ids = nullptr;  size_t cap_h_gene_ids = 0;
  hipEvent_t ev_h2d = nullptr, ev_done = nullptr, ev_d2h = nullptr;
  // the batch
  uint64_t ticket = 0;             // 0 = free
  bool waited = true;              // its results were handed out (the slot may be reused by a later submit)
  uint64_t n = 0;
  ClassifyParams p{};              // launch parameters (kept for the slow paths)
  uint32_t fast_cap = 0, gen_slots = 0;
  bool host_batch = false;
  bool long_speculative = false;   // (device-resident submit) the caller's length bound was taken on trust: checked in wait
};

struct Ctx;

// index_build.hip
int build_index(Ctx *ctx);

// classify.hip
int launch_classify_fast(Ctx *ctx, const ClassifyParams &p, uint32_t max_slots, hipStream_t stream);
const char *probe_mode_name(const Ctx *ctx);
int launch_classify_general(Ctx *ctx, const ClassifyParams &p, bool emit, unsigned n_waves, hipStream_t stream);
int launch_gather_inline(const uint32_t *count, const uint16_t *inl, const uint32_t *gene_off, uint16_t *gene_ids, uint64_t n,
                         const uint32_t *counters, hipStream_t stream);
int launch_finalize_total(const uint64_t *total, uint32_t *counters, uint64_t gene_ids_cap, hipStream_t stream);
int launch_gene_hist(const uint16_t *gene_ids, const uint32_t *counters, bool skip_if_long, unsigned long long *gene_counts, uint64_t n_reads, uint64_t n_genes,
                     hipStream_t stream);
int launch_fill_offsets(uint64_t *off, uint64_t n_plus_1, uint64_t stride, hipStream_t stream);
int launch_vouch_check(const ClassifyParams &p, uint32_t L1, uint32_t L2, uint32_t *counters, hipStream_t stream);
int launch_publish_results(const uint32_t *counters, uint32_t *h_counters, const uint32_t *gene_off, uint32_t *h_gene_off, uint64_t n_off,
                           const uint16_t *gene_ids, uint16_t *h_gene_ids, uint64_t h_ids_cap, const uint32_t *uni_flag, hipStream_t stream);
int launch_classify_uni(Ctx *ctx, const ClassifyParams &p, uint32_t max_slots, int rmode, hipStream_t stream);   // 0 ragged, 1 uniform, 2 by classes (CLS)
int launch_class_prepass(const ClassifyParams &p, uint32_t slot_cap, uint32_t *flag, hipStream_t stream);   // behind launch_uniform_check: histogram, plan, scatter
bool class_kernel_available(const Ctx *ctx, uint32_t max_slots);
bool offsets_kernel_available(const Ctx *ctx, uint32_t max_slots, bool hasq);
bool offsets_kernel_fits(const Ctx *ctx, uint32_t max_slots, uint32_t L1, uint32_t L2);
int launch_uniform_check(const ClassifyParams &p, uint32_t slot_cap, uint32_t *flag, hipStream_t stream);
bool uni_kernel_available(const Ctx *ctx);
// anchor_verdict.hip
bool anchor_verdict_applies(const ClassifyParams &p);
int launch_anchor_verdict(const ClassifyParams &p, bool pow2, bool ragged, hipStream_t stream);
uint32_t fast_kernel_max_slots();
uint32_t fast_kernel_unroll(uint32_t max_slots);  // U of the specialisation chosen for max_slots (0 = unknown)
uint32_t uni_kernel_max_groups(uint32_t max_slots);   // staging groups per pair classify_uni_kernel can take at that specialisation

struct Ctx {
  shk_params prm{};
  int mode = 0;  // 0 = accepting references, 2 = frozen (bloomfilter.h:104-110)
  hipStream_t stream = nullptr;
  std::string last_error;

  // buffered reference records (host)
  std::vector<char> ref_bytes;
  std::vector<uint64_t> ref_off{0};

  DeviceIndex idx;
  uint64_t n_records = 0, nidx = 0, n_ref_kmers = 0;

  // classify workspace: one Slot per batch in flight (PIPE_DEPTH for shk_classify_submit/_wait plus one for
  // shk_classify_device); the general kernel's scratch is shared (kernels of one context run on one stream)
  Slot slots[PIPE_DEPTH + 1];
  uint64_t next_ticket = 1;
  hipStream_t h2d_stream = nullptr, d2h_stream = nullptr;
  uint64_t *d_scratch = nullptr;   size_t cap_scratch = 0;
  unsigned long long *d_gene_counts = nullptr;
  unsigned long long *d_gene_totals = nullptr;   // result of the all-reduce (the per-GPU counters stay local)
  unsigned long long *d_work_counters = nullptr;
  int8_t q8 = 0;                                  // min_quality as the reference's `char` (argument_parser.hpp:59,:144)
  // RCCL communicators, created once: one-process-per-GPU (shk_dist_init) and one-process-many-GPUs (allreduce over contexts)
  void *dist_comm = nullptr;
  int dist_rank = 0, dist_world = 1;
  void *group_comm = nullptr;
  std::vector<int> group_devs;

  // test / A-B switches of the environment, read ONCE when the context is created (never per launch):
  //   SHK_FORCE_GENERIC=1  every batch through classify_fast_kernel (the tests run both code paths)
  //   SHK_BIG_LDS_ALWAYS=1 panels of 60-150 genes stay on the 128 KiB LDS summary whatever the previous batch said
  int env_tile_first = 0;           // SHK_TILE_FIRST=1: the tiles' round for every batch it can serve (tests); =0: never; unset: by the last batch's assigned fraction
  bool env_no_tro = false;          // SHK_NO_TRO=1: trimmed batches never through the three-pairs kernel by offsets (the class-by-class path instead; tests, A/B timing)
  bool env_force_tro = false;      // SHK_FORCE_TRO=1: batches of one length through the offsets kernel as well (diagnostic: what the offsets cost)
  bool env_no_tri = false;          // SHK_NO_TRI=1: no three-pairs-per-pass instantiation (A/B timing, tests)
  bool env_no_pre_verdict = false;  // SHK_NO_PRE_VERDICT=1: no anchor_verdict_kernel in front of the table kernels (A/B timing, tests)
  bool env_anchor_always = false;   // SHK_ANCHOR_ALWAYS=1: the anchored extension for every batch of an index that has the reference arrays (tests, A/B timing)
  bool env_ktab_always = false;     // SHK_KTAB=1: the minimiser table for every batch of an index that has it (tests)
  bool env_ktab_nt = false, env_ktab_plain = false;   // SHK_KTAB_NT=1 / 0: the minimiser table probed with / without non-temporal loads whatever its size (A/B timing, tests)
  bool env_force_generic = false, env_big_lds_always = false, env_cls_always = false;   // SHK_CLS_MIN_FILL given: no adapting to the stream
  uint32_t last_verdict = 0;        // CTR_VERDICT of the last batch finished
  uint32_t env_cls_min_fill = CLS_MIN_FILL;   // SHK_CLS_MIN_FILL: pairs per non-empty class a batch needs to go class by class (0: never; tests: 1)
  // which classify kernel the last batch's main launch was (shk_last_kernel): the choice can depend on the batch before it
  char last_kernel[160] = "";

  // timing
  bool timing = false;
  std::vector<hipEvent_t> ev_start, ev_stop, ev_pre;   // per launch: in front of the classify launches, behind them, in front of the passes over the offsets
  size_t ev_used = 0;
  shk_timing last{};
};

// error helpers
int set_hip_error(Ctx *ctx, hipError_t e, const char *what);
#define SHK_HIP(ctx, call)                                            \
  do {                                                                \
    hipError_t e__ = (call);                                          \
    if (e__ != hipSuccess) return shk::set_hip_error(ctx, e__, #call); \
  } while (0)

template <typename T>
int ensure_capacity(Ctx *ctx, T **ptr, size_t *cap, size_t need)
{
  if (need <= *cap && *ptr) return SHK_OK;
  if (*ptr) {
    hipError_t e = hipFree(*ptr);
    *ptr = nullptr;
    *cap = 0;
    if (e != hipSuccess) return set_hip_error(ctx, e, "hipFree");
  }
  size_t want = need + need / 4 + 64;
  hipError_t e = hipMalloc((void **)ptr, want * sizeof(T));
  if (e != hipSuccess) return set_hip_error(ctx, e, "hipMalloc");
  *cap = want;
  return SHK_OK;
}

}  // namespace shk

// the opaque context of the C ABI
struct shk_ctx : public shk::Ctx {};
