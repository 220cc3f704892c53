/*
 * shark_hip.h -- C ABI of libsharkhip: the MI355X (gfx950) implementation of
 * shark's k-mer classification hot path.
 *
 * shark (AlgoLab/shark) has no plugin/FFI interface; the seam this library
 * replaces is the group of functor calls inside the two worker loops of
 * main.cpp plus the BF methods they use.  Each entry point below cites the
 * reference interface it stands in for (file:line relative to the reference
 * tree).  Plain pointers and sizes only; no C++/torch types; every function
 * returns 0 (SHK_OK) or a negative error code and never throws.
 *
 * Mode machine (bloomfilter.h:104-110): shk_ref_add* -> shk_ref_finalize ->
 * shk_classify*; going backwards returns SHK_ERR_STATE.
 *
 * There is NO CPU fallback: without a HIP device every call that computes
 * fails with SHK_ERR_HIP / SHK_ERR_NO_DEVICE.
 *
 * Threading: a context serialises nothing by itself -- use one context per
 * host thread (and per GPU); different contexts may be used concurrently.  The
 * reference's ReadAnalyzer is const over a frozen index (ReadAnalyzer.hpp:39,
 * main.cpp:193); here every context owns a replica of that index.
 */
#ifndef SHARK_HIP_H
#define SHARK_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SHK_OK                    0
#define SHK_ERR_ARG              -1  /* bad argument (k outside [1,31], c outside [0,1], NULL, ...) */
#define SHK_ERR_STATE            -2  /* call not allowed in the current mode */
#define SHK_ERR_HIP              -3  /* HIP runtime error; text via shk_last_error */
#define SHK_ERR_NOMEM            -4
#define SHK_ERR_TOO_MANY_GENES   -5  /* >= 2^31 FASTA records (more than 65 536 genes are handled as the reference handles them:
                                        ids wrap to uint16_t, small_vector.hpp:46, and are not de-duplicated, bloomfilter.h:72) */
#define SHK_ERR_INDEX_TOO_LARGE  -6  /* >= 2^31 set bits or list entries (int in bloomfilter.h:70,:130) */
#define SHK_ERR_NO_DEVICE        -7

#define SHK_INLINE_IDS 4             /* gene ids stored inline per read in the device result */

typedef struct shk_ctx shk_ctx;

/* argument_parser.hpp:49-63 (namespace opt) */
typedef struct shk_params {
  uint32_t k;            /* -k, default 17, range [1,31]      (:56, :112-121) */
  double   c;            /* -c, default 0.6, range [0,1]      (:57, :122-129) */
  uint64_t bf_bits;      /* filter size in BITS; -b N => N<<33 (:58, :130-134) */
  int32_t  min_quality;  /* -q, 0 = no masking; any value >= 0: stored as the reference's `char`
                            (:59, :135-145), so -q > 94 wraps exactly as it does there          */
  int32_t  single;       /* -s                                 (:60, :146-148) */
  int32_t  device;       /* HIP device ordinal */
} shk_params;

/* BF::BF(size) bloomfilter.h:48-53 + ReadAnalyzer ctor ReadAnalyzer.hpp:36-37 */
int  shk_create(const shk_params *params, shk_ctx **out);
void shk_destroy(shk_ctx *ctx);
const char *shk_strerror(int code);
const char *shk_last_error(const shk_ctx *ctx);

/* ---- index build -------------------------------------------------------- */
/* One FASTA record, in FILE ORDER, including records shorter than k and
 * records without any valid k-mer (the library reproduces main.cpp:162-186's
 * gene numbering, quirk included).  Stands in for
 *   FastaSplitter::operator()   FastaSplitter.hpp:42-54  (record order = legend order)
 *   KmerBuilder::operator()     KmerBuilder.hpp:40-72
 *   BloomfilterFiller::operator() BloomfilterFiller.hpp:38-46 / BF::add_at bloomfilter.h:57-59
 *   pass 2 + BF::add_to_kmer    main.cpp:154-189, bloomfilter.h:61-75
 * The sequence bytes are only buffered here; all k-mer work happens on the
 * device in shk_ref_finalize. */
int shk_ref_add(shk_ctx *ctx, const char *seq, uint64_t len);

/* BF::switch_mode(1) + switch_mode(2)  bloomfilter.h:111-188, main.cpp:148,:193.
 * Runs on the device: canonical k-mers -> XXH64 -> bit set; rank directory;
 * (set bit -> ascending unique gene id list) CSR. */
int shk_ref_finalize(shk_ctx *ctx);

typedef struct shk_index_info {
  uint64_t n_records;    /* FASTA records added (= legend_ID.size(), FastaSplitter.hpp:48) */
  uint64_t nidx;         /* final gene counter (main.cpp:191) */
  uint64_t bf_bits;
  uint64_t n_set_bits;   /* num_kmer, bloomfilter.h:122 */
  uint64_t tot_idx;      /* _index_kmer.size(), bloomfilter.h:130-133 */
  uint64_t n_ref_kmers;  /* valid reference k-mer occurrences hashed */
} shk_index_info;
int shk_index_info_get(const shk_ctx *ctx, shk_index_info *info);
/* How the classify kernels look a k-mer's filter position up on this index:
 * "bitvector-mod", "bitvector", "summary+bitvector", "table", "summary+table",
 * "lds-summary+table", "lds-table" (tiny indices: the exact table is held
 * in LDS for batches of one read length; other batches of such an index take
 * lds-summary+table), and for filter sizes that are not a power of two
 * "table-mod", "lds-summary+table-mod"
 * (every mode returns exactly the filter's bit).  Environment
 * SHK_PROBE=bitvector at finalize time disables the table, SHK_TAB_DENSE=1
 * builds it at up to 0.8 load (long probe paths), SHK_NO_LDS_TABLE=1 leaves
 * the LDS-resident table out; all three are for the tests. */
const char *shk_probe_mode(const shk_ctx *ctx);
/* The classify kernel instantiation the LAST batch's main launch ran, as rocprofv3 names it (e.g.
 * "classify_uni_kernel<5, 5, false, 21, true>"; the last argument reads "device" when uniform_check_kernel decided on the
 * device which of the two launched instantiations did the work).  On a panel-sized index the choice depends on the batch
 * before (shk_probe_mode names the index's chains, this names what ran), so a timing or a profile can say what it measured.
 * New: the reference has no counterpart.  The environment's test switches (SHK_FORCE_GENERIC, SHK_BIG_LDS_ALWAYS) are read
 * once, by shk_create. */
const char *shk_last_kernel(const shk_ctx *ctx);

/* Parity introspection: copy the device-resident index to host buffers.
 * words: (bf_bits+63)/64 uint64_t in sdsl::bit_vector layout, bit i =
 * (words[i>>6] >> (i&63)) & 1 (bloomfilter.h:51,:58,:89).
 * offsets[n_set_bits+1] / ids[tot_idx]: list r (the r-th set bit, r = rank)
 * is ids[offsets[r]..offsets[r+1]) -- the explicit form of _bv/_select_bv/
 * _index_kmer (bloomfilter.h:142-167). */
int shk_index_copy_bf(const shk_ctx *ctx, uint64_t *words, uint64_t n_words);
int shk_index_copy_lists(const shk_ctx *ctx, uint32_t *offsets, uint16_t *ids);

/* ---- classification ------------------------------------------------------ */
/* A batch of reads as structure-of-arrays.  Mate 1 of read i is
 * seq1[off1[i] .. off1[i+1]); seq2/off2 NULL => single-end.  qual1/qual2 use
 * the same offsets and may be NULL when min_quality == 0.  The mate join
 * ("N", FastqSplitter.hpp:63) and the quality mask (FastqSplitter.hpp:70,
 * :104-109) are applied ON THE DEVICE; the caller passes raw FASTQ fields. */
typedef struct shk_batch {
  uint64_t        n;
  const char     *seq1;
  const uint64_t *off1;   /* n+1 */
  const char     *seq2;
  const uint64_t *off2;   /* n+1 */
  const char     *qual1;
  const char     *qual2;
} shk_batch;

/* Associations per read: read i belongs to genes
 * gene_ids[gene_off[i] .. gene_off[i+1]) (ascending), the indices the
 * reference passes to legend_ID[] at ReadAnalyzer.hpp:106.  Buffers are owned
 * by the context and stay valid until the next shk_classify* call on it. */
typedef struct shk_result {
  uint64_t        n;
  const uint32_t *gene_off;  /* n+1 */
  const uint16_t *gene_ids;  /* gene_off[n] */
  uint64_t        n_assoc;
} shk_result;

/* ReadAnalyzer::operator()(const vector<elem_t>&, vector<assoc_t>&) const
 * ReadAnalyzer.hpp:39-110 over host buffers (H2D, kernels, D2H): submit + wait of one batch. */
int shk_classify(shk_ctx *ctx, const shk_batch *batch, shk_result *result);

/* The same as a pipeline, for callers that stream batches -- the reference overlaps split / analyze /
 * output across its worker threads (main.cpp:66-77, :219-223); here up to SHK_PIPE_DEPTH batches are in
 * flight per context: while the caller waits for batch i, the H2D copies of batches i+1.. overlap the
 * kernels of batch i on separate HIP streams, and no step in between waits for the host.
 *   submit: reads `batch` (host buffers; pinned memory from shk_alloc_pinned makes the copies truly
 *           asynchronous), enqueues everything and returns a ticket.  The host buffers must stay
 *           untouched until the ticket has been waited for.  SHK_ERR_STATE when SHK_PIPE_DEPTH
 *           tickets are outstanding.
 *   wait  : blocks until that batch is classified and returns its associations in pinned host
 *           buffers owned by the context; they stay valid until SHK_PIPE_DEPTH further submits.
 *           (A ticket of shk_classify_device_submit: device pointers instead.)
 * Tickets must be waited for in the order they were submitted. */
#define SHK_PIPE_DEPTH 3
int shk_classify_submit(shk_ctx *ctx, const shk_batch *batch, uint64_t *ticket);
int shk_classify_wait(shk_ctx *ctx, uint64_t ticket, shk_result *result);

/* Same, for inputs ALREADY RESIDENT IN HBM: every pointer in `batch` is a
 * device pointer; the result pointers returned in `result` are DEVICE
 * pointers owned by the context.  max_read_len is an upper bound on the
 * longest mate (0 = unknown); it only selects the kernel specialisation --
 * reads that do not fit are routed to the general kernel, never dropped.
 * Work is enqueued on the context's stream and the call returns after the
 * stream has drained (one host synchronisation per call when max_read_len is given). */
int shk_classify_device(shk_ctx *ctx, const shk_batch *batch, uint32_t max_read_len, shk_result *result);

/* ... and as a pipeline: the device-resident entry point WITHOUT its host synchronisation (the reference's analyzer threads never
 * wait for the output stage either, main.cpp:66-77).  Everything is enqueued on the context's stream and the call returns a
 * ticket; shk_classify_wait(ticket) then returns DEVICE pointers (as shk_classify_device does), valid until SHK_PIPE_DEPTH
 * further submits.  Tickets of this call and of shk_classify_submit share the context's SHK_PIPE_DEPTH slots and are waited for
 * in submission order.
 *   max_read_len   an upper bound on the longest mate, REQUIRED here (> 0): without one the device would have to be asked
 *                  between two kernels; SHK_ERR_ARG when 0.  A bound that does not hold is noticed and repaired in wait.
 *   uniform_len1/2 when the caller KNOWS that every mate 1 (mate 2) has exactly this length and off1[i] = i * uniform_len1
 *                  (off2 likewise) -- a sequencer's output, generated or copied to the device by the caller itself -- it says so
 *                  here, as the host entry points find out by scanning the offsets: the kernel then never reads an offset and
 *                  the pass that would verify them on the device (61 us per 10 M pairs) is not made.  0, 0: unknown, the
 *                  device looks (as shk_classify_device always does).  The caller vouches for what it states; one thread on
 *                  the device compares three offsets per mate (first, middle, last) with r * length, and shk_classify_wait
 *                  returns SHK_ERR_ARG for a batch whose caller vouched wrongly (no results are handed out). */
int shk_classify_device_submit(shk_ctx *ctx, const shk_batch *batch, uint32_t max_read_len, uint32_t uniform_len1, uint32_t uniform_len2,
                               uint64_t *ticket);

/* Per-gene number of assigned reads accumulated over all classify calls (all
 * waited tickets) since the last reset (counts[g] for g in [0, 65536)); the quantity all-reduced
 * across GPUs.  n must be <= 65536. */
int shk_gene_counts(shk_ctx *ctx, uint64_t *counts, uint32_t n);
int shk_gene_counts_reset(shk_ctx *ctx);

/* The path's one exchange step when the read stream is sharded over several GPUs of one node (index
 * replicated, reads split): all-reduce (sum) of the per-gene counters over RCCL (xGMI).  New: the reference is
 * single process and has no counterpart (its per-read lines are merged by the output mutex, ReadOutput.hpp:38).
 * The totals land in a separate device buffer of every context; the per-GPU counters are left as they are, so
 * the call can be repeated.  RCCL is loaded on first use (librccl.so.1) and the communicators are created once.
 *
 * (a) one process, one context per GPU (`shark --gpus N`): `n_ctx` contexts; with one context and without
 *     SHK_FORCE_RCCL=1 in the environment no collective is needed and none is issued. */
int shk_gene_counts_allreduce(shk_ctx **ctxs, int n_ctx, uint64_t *totals, uint32_t n);
/* (b) one process per GPU (torch.distributed.run, mpirun): rank 0 calls shk_dist_unique_id, the launcher's own
 *     channel carries the SHK_DIST_ID_BYTES bytes to every rank, every rank calls shk_dist_init with its context;
 *     shk_dist_gene_counts_allreduce is then collective over the ranks (stream-ordered behind the classify calls
 *     made so far) and returns counts[0..n) of the totals.  Without shk_dist_init it returns the local counters. */
#define SHK_DIST_ID_BYTES 128
int shk_dist_unique_id(uint8_t *id);
int shk_dist_init(shk_ctx *ctx, const uint8_t *id, int rank, int world);
int shk_dist_gene_counts_allreduce(shk_ctx *ctx, uint64_t *totals, uint32_t n);
/* What the communicator itself says about the job this context joined with shk_dist_init: ncclCommUserRank /
 * ncclCommCount (a context that never joined one reports rank 0 of 1).  bench.py prints it as `ranks_seen`, so a
 * run that was asked for N GPUs and reached the collective with fewer cannot go unnoticed. */
int shk_dist_info(const shk_ctx *ctx, int *rank, int *world);

/* ---- measurement --------------------------------------------------------- */
/* HIP-event timing of the dominant kernel (classify) on the context's own
 * stream.  enable=1 starts recording one event pair per launch. */
int shk_timing_enable(shk_ctx *ctx, int enable);
/* n_launches classify-kernel launches since enable, their summed duration,
 * and the algorithmic work counters of the LAST classify call.  prepass_ms:
 * what ran in front of those launches for batches whose read lengths only the
 * device sees (shk_classify_device) or that are known to be ragged -- the
 * uniformity check over the offsets and, for batches of mixed lengths on small
 * indices, the sort by length; 0 for batches the host knows to be uniform. */
typedef struct shk_timing {
  uint64_t n_launches;
  double   total_ms;
  uint64_t last_n_reads;     /* pairs (or single reads) in the last call */
  uint64_t last_n_long;      /* reads routed to the general kernel */
  uint64_t last_n_tie;       /* reads with more than SHK_INLINE_IDS genes */
  uint64_t last_n_assoc;
  double   prepass_ms;
} shk_timing;
int shk_timing_get(shk_ctx *ctx, shk_timing *t);

/* Exact per-call work counters for the roofline's algorithmic-byte formula
 * (SURVEY.md 8d): counts k-mers probed, probes that hit, and gene-list
 * entries read during the NEXT classify call when enabled (a separate,
 * slower kernel build is used; never enabled in timed runs). */
typedef struct shk_work_counters {
  uint64_t n_kmers;      /* valid k-mers probed */
  uint64_t n_hits;       /* probes whose bit was set */
  uint64_t n_list_ids;   /* sum of list lengths over hits */
  uint64_t n_bases;      /* input base bytes consumed */
} shk_work_counters;
int shk_count_work(shk_ctx *ctx, const shk_batch *dev_batch, shk_work_counters *out);

/* The part's ceiling for INDEPENDENT random 16-byte lookups in a table of `table_bytes` (a power of two >= 1 MiB) on the
 * context's device: the access pattern of the position table, one bucket per k-mer at a hashed address.  On an index far
 * beyond the caches each lookup is one memory-side request (a 128-byte line of which 16 bytes are used) and the RATE of
 * those bounds the classify kernel; bench.py measures the ceiling with this call in the run whose fraction of it it
 * reports.  Allocates the table, performs about `n_lookups` lookups (five in flight per lane, 8 waves per SIMD;
 * `nontemporal` bit 0: streaming loads; bit 1: every lookup also reads 16 bytes of the other 64-byte half of its 128-byte
 * line -- the pair of rates says whether a random lookup moves a whole line or half of one, bench.py's calibration of
 * FETCH_SIZE for this access pattern; the figure returned is LINES per second either way), frees it again.  On no product
 * path; new (the reference has no counterpart). */
int shk_measure_random_lookups(shk_ctx *ctx, uint64_t table_bytes, uint64_t n_lookups, int nontemporal, double *g_lookups_per_s);

/* The ISSUE ceiling of the exact-table classify kernel's own instruction mix: a kernel that does, on register operands only
 * (no LDS, no memory), the arithmetic that kernel does for a read on its shortest way through -- stage eight bases, a slot's two
 * windows, canonical form, XXH64 (kmer_utils.hpp:81-83), the exact table's address arithmetic and compare, validity window and
 * coverage step -- `iters` times per lane with `waves_per_simd` (1 ... 8) waves resident per SIMD; *ms = its duration,
 * *wave_iterations = waves x iters.  bench.py takes the instructions per iteration from the same rocprofv3 counter pass that
 * counts the classify kernel's, and prints the classify kernel's VALU rate as a fraction of this kernel's (`mix_ceiling`) next
 * to the fraction of the 2-cycle peak.  On no product path; new (the reference has no counterpart). */
int shk_measure_valu_mix(shk_ctx *ctx, int waves_per_simd, uint32_t iters, double *ms, uint64_t *wave_iterations);

/* The same, and *shader_ghz = the clock the SIMDs held while that kernel ran: shader cycles (s_memtime) over the constant 100 MHz
 * counter (s_memrealtime) around each wave's loop, summed over the waves.  The chip lowers its clock under load, so a rate in
 * instructions per second prices a kernel against 2.4 GHz it may not have had; bench.py reports cycles per instruction beside it
 * (`roofline.clock`).  On no product path; new (the reference has no counterpart). */
int shk_measure_valu_mix_clock(shk_ctx *ctx, int waves_per_simd, uint32_t iters, double *ms, uint64_t *wave_iterations, double *shader_ghz);

/* pinned host memory helpers for callers that stream batches */
void *shk_alloc_pinned(size_t bytes);
void  shk_free_pinned(void *p);

/* library / build identification */
const char *shk_version(void);

#ifdef __cplusplus
}
#endif
#endif
