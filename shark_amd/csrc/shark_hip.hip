// shark_hip.hip -- the C ABI of libsharkhip (include/shark_hip.h): context,
// device memory, and the orchestration around the HIP kernels.  No CPU
// implementation of any hot-path step lives here: without a device every
// computing entry point fails.
#include <hip/hip_runtime.h>
#include <chrono>
#include <mutex>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <dlfcn.h>
#include <unistd.h>
#include <cstring>
#include <new>

#include "device_scan.hpp"
#include "shark_internal.hpp"

namespace shk {

int set_hip_error(Ctx *ctx, hipError_t e, const char *what)
{
  if (ctx) {
    ctx->last_error = std::string(what) + ": " + hipGetErrorString(e);
  }
  return e == hipErrorOutOfMemory ? SHK_ERR_NOMEM : SHK_ERR_HIP;
}

static void free_index(DeviceIndex &ix)
{
  hipFree(ix.bf64); hipFree(ix.rank_w); hipFree(ix.ent); hipFree(ix.ids); hipFree(ix.sum32); hipFree(ix.tab); hipFree(ix.lsum32); hipFree(ix.lbig32); hipFree(ix.ltab); hipFree(ix.ref2); hipFree(ix.refpay); hipFree(ix.refext); hipFree(ix.refmul); hipFree(ix.atab); hipFree(ix.ktab);
  ix = DeviceIndex{};
}

// ---- tracing hook: roctx ranges around the library's phases, so that a rocprofv3 --marker-trace /
// --sys-trace timeline shows index build, submit and wait next to the kernels.  Off unless SHK_ROCTX=1; libroctx64 is
// loaded on first use, nothing is linked.
namespace {
struct Roctx {
  int (*push)(const char *) = nullptr;
  int (*pop)() = nullptr;
  bool tried = false;
};
Roctx &roctx()
{
  static Roctx r;
  if (!r.tried) {
    r.tried = true;
    const char *e = getenv("SHK_ROCTX");
    if (e && e[0] == '1') {
      void *lib = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_LOCAL);
      if (!lib) lib = dlopen("libroctx64.so", RTLD_NOW | RTLD_LOCAL);
      if (lib) {
        r.push = (int (*)(const char *))dlsym(lib, "roctxRangePushA");
        r.pop = (int (*)())dlsym(lib, "roctxRangePop");
      }
    }
  }
  return r;
}
}  // namespace
struct TraceRange {
  explicit TraceRange(const char *name) { if (roctx().push) { (void)roctx().push(name); on = true; } }
  ~TraceRange() { if (on && roctx().pop) (void)roctx().pop(); }
  bool on = false;
};

// slot positions of a read (pair) whose mates are `len` long: classify.hip packs mate 2 at the next
// multiple of 8 after mate 1 and uses packed positions as k-mer slots
static uint32_t slots_for_len(uint32_t len, uint32_t k, bool paired)
{
  const uint32_t per = len >= k ? len - k + 1 : 0;
  return (paired && per) ? ((len + 7u) & ~7u) + per : per;
}

// exact slot count of one read (pair): the `ns` of classify.hip's process_read
static uint32_t slots_of_read(uint64_t L1, uint64_t L2, uint32_t k)
{
  const uint64_t nk1 = L1 >= k ? L1 - k + 1 : 0, nk2 = L2 >= k ? L2 - k + 1 : 0;
  const uint64_t ns = nk2 ? ((L1 + 7u) & ~7ull) + nk2 : nk1;
  return ns > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)ns;
}

template <typename T>
static int ensure_pinned(Ctx *ctx, T **ptr, size_t *cap, size_t need)
{
  if (need <= *cap && *ptr) return SHK_OK;
  if (*ptr) { (void)hipHostFree(*ptr); *ptr = nullptr; *cap = 0; }
  const size_t want = need + need / 4 + 64;
  hipError_t e = hipHostMalloc((void **)ptr, want * sizeof(T), hipHostMallocDefault);
  if (e != hipSuccess) return set_hip_error(ctx, e, "hipHostMalloc");
  *cap = want;
  return SHK_OK;
}

static int slot_init(Ctx *ctx, Slot &s)
{
  // (the batch's counters and, behind them, the words of the uniformity check: one allocation, cleared by one memset per batch)
  SHK_HIP(ctx, hipMalloc((void **)&s.d_counters, (COUNTER_BLOCK_WORDS + UNI_FLAG_WORDS) * sizeof(uint32_t)));
  s.d_uni_flag = s.d_counters + COUNTER_BLOCK_WORDS;
  SHK_HIP(ctx, hipMalloc((void **)&s.d_out, sizeof(ClassifyOut)));
  SHK_HIP(ctx, hipHostMalloc((void **)&s.h_counters, CTR_WORDS * sizeof(uint32_t), hipHostMallocDefault));
  SHK_HIP(ctx, hipEventCreateWithFlags(&s.ev_h2d, hipEventDisableTiming));
  SHK_HIP(ctx, hipEventCreateWithFlags(&s.ev_done, hipEventDisableTiming));
  SHK_HIP(ctx, hipEventCreateWithFlags(&s.ev_d2h, hipEventDisableTiming));
  return SHK_OK;
}

static void slot_free(Slot &s)
{
  hipFree(s.d_seq1); hipFree(s.d_seq2); hipFree(s.d_qual1); hipFree(s.d_qual2); hipFree(s.d_off1); hipFree(s.d_off2);
  hipFree(s.d_count); hipFree(s.d_inl); hipFree(s.d_gene_off); hipFree(s.d_gene_ids);
  hipFree(s.d_long_queue); hipFree(s.d_tie_queue); hipFree(s.d_counters); hipFree(s.d_scan_temp); hipFree(s.d_out); hipFree(s.d_plan); hipFree(s.d_cls_entries); hipFree(s.d_cls_list); hipFree(s.d_cls_share); hipFree(s.d_cls_hist);
  if (s.h_counters) (void)hipHostFree(s.h_counters);
  if (s.h_gene_off) (void)hipHostFree(s.h_gene_off);
  if (s.h_gene_ids) (void)hipHostFree(s.h_gene_ids);
  if (s.ev_h2d) (void)hipEventDestroy(s.ev_h2d);
  if (s.ev_done) (void)hipEventDestroy(s.ev_done);
  if (s.ev_d2h) (void)hipEventDestroy(s.ev_d2h);
  s = Slot{};
}

// result buffers of a batch of n reads; *d_out is rewritten only when one of them moved (a blocking
// copy: never on the steady-state path of the pipeline)
static int slot_reserve(Ctx *ctx, Slot &s, uint64_t n)
{
  int rc;
  if ((rc = ensure_capacity(ctx, &s.d_count, &s.cap_count, n + 1))) return rc;
  if ((rc = ensure_capacity(ctx, &s.d_inl, &s.cap_inl, n * SHK_INLINE_IDS + 8))) return rc;
  if ((rc = ensure_capacity(ctx, &s.d_gene_off, &s.cap_gene_off, n + 1))) return rc;
  if ((rc = ensure_capacity(ctx, &s.d_long_queue, &s.cap_long_queue, n + 1))) return rc;
  if ((rc = ensure_capacity(ctx, &s.d_tie_queue, &s.cap_tie_queue, 3 * n + 3))) return rc;
  if ((rc = ensure_capacity(ctx, &s.d_scan_temp, &s.cap_scan_temp, scan_temp_words(n + 1)))) return rc;
  // associations: two per read to start with; a batch that needs more is finished by the overflow path
  if ((rc = ensure_capacity(ctx, &s.d_gene_ids, &s.cap_gene_ids, 2 * n + 4096))) return rc;
  ClassifyOut ho;
  ho.count = s.d_count; ho.inl = s.d_inl; ho.counters = s.d_counters; ho.long_queue = s.d_long_queue; ho.tie_queue = s.d_tie_queue;
  if (memcmp(&ho, &s.out_shadow, sizeof(ho)) != 0) {
    SHK_HIP(ctx, hipMemcpy(s.d_out, &ho, sizeof(ho), hipMemcpyHostToDevice));
    s.out_shadow = ho;
  }
  return SHK_OK;
}

static void fill_params(Ctx *ctx, Slot &s, const shk_batch *b)
{
  ClassifyParams p{};
  const DeviceIndex &ix = ctx->idx;
  p.bf64 = ix.bf64; p.rank_w = ix.rank_w; p.ent = ix.ent; p.ids = ix.ids;
  p.sum32 = ix.sum_shift ? ix.sum32 : nullptr; p.sum_shift = ix.sum_shift;
  p.tab = ix.tab_lg ? ix.tab : nullptr; p.tab_lg = ix.tab_lg;
  p.tab_nt = ix.tab_lg && (16ull << ix.tab_lg) > (256ull << 20);   // beyond L2 + Infinity Cache
  p.lsum32 = ix.lsum_shift ? ix.lsum32 : nullptr; p.lsum_shift = ix.lsum_shift;
  p.lx_gene = 0xFFFFFFFFu;   // (launch_classify_uni sets it when it chooses the exact LDS table)
  p.ref2 = ix.ref2; p.refpay = ix.refpay; p.atab = ix.atab; p.ref_total = ix.ref_total; p.refext = ix.refext; p.refmul = ix.refmul;
  p.ktab = ix.ktab_lg ? ix.ktab : nullptr; p.ktab_lg = ix.ktab_lg; p.ktab_w = ix.ktab_w;
  // (far beyond the caches: streaming loads -- 17.8 / 18.1 -> 16.7 / 17.3 ms per 10 M pairs at 0 / 50 % on-target on the 60 000-gene index)
  p.ktab_nt = (ix.ktab_lg && (16ull << ix.ktab_lg) > (256ull << 20) && !ctx->env_ktab_plain) || ctx->env_ktab_nt ? 1u : 0u;
  p.bf_bits = ix.bf_bits;
  p.bf_mask = ix.pow2 ? ix.bf_bits - 1 : ~0ull;   // (non power-of-two: positions are reduced explicitly, the masks become no-ops)
  if (!ix.pow2) {
    const uint32_t sh = (uint32_t)__builtin_ctzll(ix.bf_bits);
    const uint64_t m = ix.bf_bits >> sh;
    p.mod_shift = sh;
    p.mod_fast = sh >= 32 && m < (1ull << 32);
    p.mod_m = p.mod_fast ? (uint32_t)m : 0;
    p.mod_c = p.mod_fast ? 0xFFFFFFFFFFFFFFFFull / m + 1 : 0;
  }
  p.k = ctx->prm.k; p.c = ctx->prm.c; p.single = ctx->prm.single;
  // FastqSplitter.hpp:52,:70 with the reference's `char min_quality` (argument_parser.hpp:59,:144): no masking when the
  // char is 0; otherwise the threshold is (char)(min_quality + 33), which wraps for -q > 94 just as it does there
  p.hasq = ctx->q8 != 0;
  p.mq = (int32_t)(int8_t)(uint8_t)((int)ctx->q8 + 33);
  p.n = b->n;
  p.seq1 = (const uint8_t *)b->seq1; p.off1 = b->off1;
  p.seq2 = (const uint8_t *)b->seq2; p.off2 = b->off2;
  p.qual1 = (const uint8_t *)b->qual1; p.qual2 = (const uint8_t *)b->qual2;
  p.out = s.d_out;
  p.flags = s.d_counters;
#ifdef SHK_ABLATION
  p.ablate = getenv("SHK_ABLATE") ? (uint32_t)atoi(getenv("SHK_ABLATE")) : 0u;
#endif
  s.p = p;
}

// scratch of the general kernel for `n_items` work items of at most `slots` k-mer slots
static int size_scratch(Ctx *ctx, ClassifyParams &p, uint32_t slots, uint64_t n_items, unsigned *n_waves)
{
  const uint32_t S = ((slots + 63) / 64) * 64;
  const uint64_t stride = (uint64_t)stage_words_for(S) + (3ull * S) / 2 + 2;   // staging area + 3 u32 slot records
  uint64_t waves = std::min<uint64_t>(std::max<uint64_t>(n_items, 1), 4096);
  const uint64_t budget_words = (1ull << 31) / 8;  // at most 2 GiB of scratch
  if (waves * stride > budget_words) waves = std::max<uint64_t>(1, budget_words / stride);
  waves = ((waves + 3) / 4) * 4;
  int r = ensure_capacity(ctx, &ctx->d_scratch, &ctx->cap_scratch, waves * stride);
  if (r) return r;
  p.scratch = ctx->d_scratch;
  p.scratch_stride_words = stride;
  p.scratch_slots = S;
  *n_waves = (unsigned)waves;
  return SHK_OK;
}

// reads queued for the general kernel because they exceed the fast kernel's slot capacity
static int run_long_reads(Ctx *ctx, Slot &s, uint32_t n_long)
{
  if (!n_long) return SHK_OK;
  ClassifyParams p = s.p;
  unsigned n_waves = 0;
  int rc;
  if ((rc = size_scratch(ctx, p, s.gen_slots, n_long, &n_waves))) return rc;
  p.work = s.d_long_queue;
  p.n_work = n_long;
  p.work_count = nullptr;
  return launch_classify_general(ctx, p, false, n_waves, ctx->stream);
}

// Everything behind the classify kernels, WITHOUT a host round trip: per-read counts -> offsets (scan), inline ids ->
// CSR (gather), reads with more than SHK_INLINE_IDS genes (EMIT pass of the general kernel over the tie queue, whose
// length stays on the device), per-gene histogram, counters -> pinned host memory, event.
static int enqueue_tail(Ctx *ctx, Slot &s, bool skip_hist_if_long, bool count_genes)
{
  hipStream_t st = ctx->stream;
  const uint64_t n = s.n;
  int rc;
  const uint64_t *d_total = exclusive_scan_u32(s.d_count, s.d_gene_off, n + 1, s.d_scan_temp, st);
  SHK_HIP(ctx, hipGetLastError());
  if ((rc = launch_finalize_total(d_total, s.d_counters, s.cap_gene_ids, st))) return rc;
  if ((rc = launch_gather_inline(s.d_count, s.d_inl, s.d_gene_off, s.d_gene_ids, n, s.d_counters, st))) return rc;
  {
    ClassifyParams p = s.p;
    unsigned n_waves = 0;
    if ((rc = size_scratch(ctx, p, s.gen_slots, std::min<uint64_t>(n, 1024), &n_waves))) return rc;
    p.work = s.d_tie_queue;
    p.n_work = 0;
    p.work_count = s.d_counters + CTR_TIE;
    p.gene_off = s.d_gene_off;
    p.gene_ids = s.d_gene_ids;
    if ((rc = launch_classify_general(ctx, p, true, n_waves, st))) return rc;
  }
  if (count_genes && (rc = launch_gene_hist(s.d_gene_ids, s.d_counters, skip_hist_if_long, ctx->d_gene_counts, n, ctx->n_records, st))) return rc;
  // counters (and, for host batches, the associations) are stored into pinned host memory by a kernel: no copy-engine
  // command ever sits in this stream waiting for kernels (see publish_results_kernel)
  if (s.host_batch) {
    if ((rc = ensure_pinned(ctx, &s.h_gene_off, &s.cap_h_gene_off, n + 1))) return rc;
    if ((rc = ensure_pinned(ctx, &s.h_gene_ids, &s.cap_h_gene_ids, s.cap_gene_ids))) return rc;
  }
  if ((rc = launch_publish_results(s.d_counters, s.h_counters, s.d_gene_off, s.host_batch ? s.h_gene_off : nullptr, n + 1, s.d_gene_ids,
                                   s.h_gene_ids, s.cap_h_gene_ids, s.p.uni_flag, st)))
    return rc;
  SHK_HIP(ctx, hipEventRecord(s.ev_done, st));
  return SHK_OK;
}

enum LongMode {
  LONG_NONE_EXPECTED = 0,   // the caller's length bound says every read fits the fast kernel; checked after the fact
  LONG_KNOWN = 1,           // the host has counted the reads that do not fit (host batches)
  LONG_UNKNOWN = 2          // no bound: one host round trip behind the fast kernel
};

// what the device said about the batch just finished (0: the host knew); shk_last_kernel() carries it
static void note_verdict(Ctx *ctx, uint32_t v)
{
  ctx->last_verdict = v;
  if (char *e = strstr(ctx->last_kernel, " verdict=")) *e = 0;
  if (v) {
    const size_t l = strlen(ctx->last_kernel);
    snprintf(ctx->last_kernel + l, sizeof(ctx->last_kernel) - l, " verdict=%s", v == 1u ? "ragged" : (v == 2u ? "uniform" : (v == 3u ? "classes" : "offsets")));
  }
}

// all kernels of one batch whose inputs are (or will be, in stream order) resident in HBM; batch pointers are device
// pointers.  Returns with the work enqueued on ctx->stream; finish_classify() completes the rare slow paths.
enum UniMode {
  UNI_ASK_DEVICE = 0,  // lengths unknown on the host: the device decides (uniform / ragged / by classes), every candidate kernel is launched, all but one return at once
  UNI_YES = 1,         // the host has seen the offsets: one length per mate (uni_L1, uni_L2), and such a read fits
  UNI_NO = 2
};

static int enqueue_classify(Ctx *ctx, Slot &s, const shk_batch *b, uint32_t max_slots, int long_mode, uint32_t n_long_host,
                            uint32_t long_slots_host, bool count_genes, int uni_mode = UNI_ASK_DEVICE, uint32_t uni_L1 = 0, uint32_t uni_L2 = 0,
                            bool groups_fit = true)
{
  hipStream_t st = ctx->stream;
  const uint64_t n = b->n;
  int rc;
  if ((rc = slot_reserve(ctx, s, n))) return rc;
  s.n = n;
  fill_params(ctx, s, b);
  SHK_HIP(ctx, hipMemsetAsync(s.d_counters, 0, (COUNTER_BLOCK_WORDS + UNI_FLAG_WORDS) * sizeof(uint32_t), st));
  if (max_slots > fast_kernel_max_slots()) max_slots = fast_kernel_max_slots();
  s.fast_cap = 64 * fast_kernel_unroll(max_slots);
  s.gen_slots = s.fast_cap;
  // an index with a position table: classify_uni_kernel, uniform or not -- unless the batch is known to hold reads of more
  // than 64 staging groups (> 512 bases per pair), which only classify_fast_kernel stages without the general kernel's help
  const bool table_kernel = uni_kernel_available(ctx) && n != 0 && groups_fit;
  // a batch of mixed lengths on an index whose uniform batches take the exact table in LDS: sorted by the pairs' two lengths on
  // the device and classified class by class (classify_uni_kernel's CLS instantiation) -- when the classes are few enough for
  // the histogram (the caller's bound on the read length says) and, the device decides, full enough; also when the host knows
  // the batch is ragged: it has not counted
  // ((l1, l2) tables -- the classes' counters, the ragged instantiation's plans -- are sized by the caller's bound on the read length,
  //  2^20 entries at most and when there is no bound or a loose one; the device checks the batch's longest mates against them)
  const uint64_t hint_classes = uni_L1 ? ((uint64_t)uni_L1 + 1) * ((uint64_t)(b->seq2 ? uni_L2 : 0) + 1) : 0;
  const uint64_t classes = table_kernel ? std::min<uint64_t>(hint_classes ? hint_classes : (1ull << 20), 1ull << 20) : 0;
  // (a batch only the device can judge, in a stream whose last batch was uniform -- a sequencer's output --: the four launches of
  //  this path are left out, 2 % of such a batch; should it be ragged after all, the ragged instantiation takes it, and the next one
  //  comes here again)
  const bool stream_is_uniform = uni_mode == UNI_ASK_DEVICE && ctx->last_verdict == 2u && !ctx->env_cls_always;
  bool by_classes = table_kernel && uni_mode != UNI_YES && class_kernel_available(ctx, max_slots) && n < (1ull << 31) && ctx->env_cls_min_fill != 0 &&
                    !stream_is_uniform;
  // read plans of a ragged batch, indexed by (l1, l2) up to the longest mates (uni_L1 / uni_L2: known, or the caller's bound for
  // a resident batch; no bound, or a loose one: 2^20 entries, of which the batch's own longest mates decide how many are used).
  // Cleared per launch, filled by the kernel.
  // ... or, where it exists and the longest mates fit its layout, through the three-pairs kernel by offsets (TRO): no sorting.  A host
  // batch known to be ragged (UNI_NO: uni_L1 / uni_L2 are its longest mates) takes it instead of the ragged instantiation; for a batch
  // only the device can judge it is launched beside the others and uniform_check_kernel's verdict 3 picks it
  const bool tro_avail = table_kernel && uni_mode != UNI_YES && offsets_kernel_available(ctx, max_slots, ctx->q8 != 0) && n < (1ull << 31);
  const bool tro_host = tro_avail && uni_mode == UNI_NO && offsets_kernel_fits(ctx, max_slots, uni_L1, b->seq2 ? uni_L2 : 0u);
  const bool tro_ask = tro_avail && uni_mode == UNI_ASK_DEVICE;
  if (tro_host) by_classes = false;
  const bool with_plans = table_kernel && uni_mode != UNI_YES;
  if (by_classes) {
    // (32 bytes per pair of extra HBM: a device too full for them classifies the batch with the ragged instantiation instead)
    if (ensure_capacity(ctx, &s.d_cls_entries, &s.cap_cls_entries, (size_t)(2 * n)) != SHK_OK ||
        ensure_capacity(ctx, &s.d_cls_list, &s.cap_cls_list, (size_t)classes) != SHK_OK ||
        ensure_capacity(ctx, &s.d_cls_hist, &s.cap_cls_hist, (size_t)(2 * classes)) != SHK_OK ||
        (!s.d_cls_share && hipMalloc((void **)&s.d_cls_share, CLS_SHARES * sizeof(uint32_t)) != hipSuccess)) {
      (void)hipGetLastError();
      ctx->last_error.clear();
      by_classes = false;
    }
  }
  if (by_classes) {
    s.p.cls_entries = s.d_cls_entries;
    s.p.cls_list = s.d_cls_list;
    s.p.cls_share_first = s.d_cls_share;
    s.p.cls_hist = s.d_cls_hist;
    s.p.cls_cap = (uint32_t)classes;
    s.p.cls_min_fill = ctx->env_cls_min_fill;
  }
  if (with_plans) {
    if ((rc = ensure_capacity(ctx, &s.d_plan, &s.cap_plan, (size_t)classes))) return rc;
    s.p.plan_tab = s.d_plan;
    s.p.plan_cap = (uint32_t)classes;
  }
  if (!table_kernel) {
    uni_mode = UNI_NO;
    SHK_HIP(ctx, hipMemsetAsync(s.d_count + n, 0, sizeof(uint32_t), st));
  } else {
    // classify_uni_kernel writes count[] only for reads with associations
    SHK_HIP(ctx, hipMemsetAsync(s.d_count, 0, (n + 1) * sizeof(uint32_t), st));
  }
  // timed: the classify launches (shk_timing::total_ms) and, for a batch the device has to look at first, the passes over its
  // offsets in front of them (prepass_ms)
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (ctx->timing) {
    if (ctx->ev_used == ctx->ev_start.size()) {
      hipEvent_t a, c, d;
      SHK_HIP(ctx, hipEventCreate(&a));
      SHK_HIP(ctx, hipEventCreate(&c));
      SHK_HIP(ctx, hipEventCreate(&d));
      ctx->ev_start.push_back(a);
      ctx->ev_stop.push_back(c);
      ctx->ev_pre.push_back(d);
    }
    e0 = ctx->ev_start[ctx->ev_used];
    e1 = ctx->ev_stop[ctx->ev_used];
    SHK_HIP(ctx, hipEventRecord(ctx->ev_pre[ctx->ev_used], st));
    ctx->ev_used++;
  }
  if (table_kernel) {
    if (by_classes) {
      uni_mode = UNI_ASK_DEVICE;
      SHK_HIP(ctx, hipMemsetAsync(s.d_cls_hist, 0, classes * sizeof(uint32_t), st));
    }
    if (uni_mode == UNI_ASK_DEVICE) {
      s.p.tro = tro_ask ? (ctx->env_force_tro ? 2u : 1u) : 0u;
      if ((rc = launch_uniform_check(s.p, s.fast_cap, s.d_uni_flag, st))) return rc;
      if (by_classes && (rc = launch_class_prepass(s.p, s.fast_cap, s.d_uni_flag, st))) return rc;
      s.p.uni_flag = s.d_uni_flag;
    } else {
      // the one length per mate (UNI_YES), or the longest mates: the ragged instantiation stages the batch in their layout
      s.p.uni_L1 = uni_L1;
      s.p.uni_L2 = uni_L2;
      // (a resident batch whose caller vouched for the lengths: nobody has looked at its offsets)
      if (uni_mode == UNI_YES && !s.host_batch && n != 0 && (rc = launch_vouch_check(s.p, uni_L1, uni_L2, s.d_counters, st))) return rc;
    }
    if (with_plans) SHK_HIP(ctx, hipMemsetAsync(s.d_plan, 0, classes * sizeof(uint4), st));
  }
  if (ctx->timing) SHK_HIP(ctx, hipEventRecord(e0, st));

  // the timed launch is the one that does the work: the uniform kernel when the host knows it applies (or has to ask
  // the device: then the generic kernel is launched behind it and returns at once for a uniform batch)
  if (ctx->idx.wrap) {
    // more than 65 536 genes: lists carry multiplicities, which only the general kernel counts (classify.hip, WRAP); it
    // runs over all reads with the fast kernel's slot capacity and queues what does not fit, as the fast kernel would
    ClassifyParams p = s.p;
    unsigned n_waves = 0;
    if ((rc = size_scratch(ctx, p, s.fast_cap, n, &n_waves))) return rc;
    p.work = nullptr;
    p.n_work = n;
    p.work_count = nullptr;
    if ((rc = launch_classify_general(ctx, p, false, n_waves, st))) return rc;
    snprintf(ctx->last_kernel, sizeof(ctx->last_kernel), "classify_general_kernel<wrap>");
  }
  if (!ctx->idx.wrap) {
    if (!table_kernel) {
      if ((rc = launch_classify_fast(ctx, s.p, max_slots, st))) return rc;           // bit-vector probe chains
    } else {
      // what the host knows decides the launch; when only the device knows, both are made and one returns at once
      if (uni_mode != UNI_NO && (rc = launch_classify_uni(ctx, s.p, max_slots, 1, st))) return rc;
      if ((tro_host || tro_ask) && (rc = launch_classify_uni(ctx, s.p, max_slots, 3, st))) return rc;
      if (by_classes && (rc = launch_classify_uni(ctx, s.p, max_slots, 2, st))) return rc;
      if (uni_mode != UNI_YES && !tro_host && (rc = launch_classify_uni(ctx, s.p, max_slots, 0, st))) return rc;
    }
  }
  if (ctx->timing) SHK_HIP(ctx, hipEventRecord(e1, st));

  uint32_t n_long = 0;
  if (long_mode == LONG_UNKNOWN) {
    if ((rc = launch_publish_results(s.d_counters, s.h_counters, nullptr, nullptr, 0, nullptr, nullptr, 0, s.p.uni_flag, st))) return rc;
    SHK_HIP(ctx, hipStreamSynchronize(st));
    n_long = s.h_counters[CTR_LONG];
    s.gen_slots = std::max(s.fast_cap, s.h_counters[CTR_MAX_SLOTS]);
  } else if (long_mode == LONG_KNOWN) {
    n_long = n_long_host;
    s.gen_slots = std::max(s.fast_cap, long_slots_host);
  }
  if ((rc = run_long_reads(ctx, s, n_long))) return rc;
  return enqueue_tail(ctx, s, long_mode == LONG_NONE_EXPECTED, count_genes);
}

// after ev_done: complete what the enqueued work could not (both paths are rare and synchronous)
//  * a length bound that did not hold: reads sit in the long queue -> general kernel, tail again
//  * more associations than gene_ids holds -> grow it, tail again
// The histogram kernel skipped itself in exactly these cases, so no batch is counted twice.
static int finish_classify(Ctx *ctx, Slot &s, bool long_was_speculative, bool count_genes, bool *redone)
{
  hipStream_t st = ctx->stream;
  int rc;
  *redone = false;
  if (long_was_speculative && s.h_counters[CTR_LONG]) {
    s.gen_slots = std::max(s.fast_cap, s.h_counters[CTR_MAX_SLOTS]);
    if ((rc = run_long_reads(ctx, s, s.h_counters[CTR_LONG]))) return rc;
    if ((rc = enqueue_tail(ctx, s, false, count_genes))) return rc;
    SHK_HIP(ctx, hipStreamSynchronize(st));
    *redone = true;
  }
  if (s.h_counters[CTR_OVERFLOW]) {
    const uint64_t n_assoc = ((uint64_t)s.h_counters[CTR_ASSOC_HI] << 32) | s.h_counters[CTR_ASSOC_LO];
    if (n_assoc >= 0xFFFFFFFFull) { ctx->last_error = "more than 2^32-1 associations in one batch"; return SHK_ERR_ARG; }
    SHK_HIP(ctx, hipStreamSynchronize(st));     // gene_ids is about to be replaced
    if ((rc = ensure_capacity(ctx, &s.d_gene_ids, &s.cap_gene_ids, n_assoc + 8))) return rc;
    if ((rc = enqueue_tail(ctx, s, false, count_genes))) return rc;
    SHK_HIP(ctx, hipStreamSynchronize(st));
    *redone = true;
  }
  return SHK_OK;
}

// a batch resident in HBM, start to finish (shk_classify_device, shk_count_work)
static int classify_resident(Ctx *ctx, const shk_batch *b, uint32_t max_read_len, shk_result *res, shk_work_counters *wc = nullptr)
{
  hipStream_t st = ctx->stream;
  Slot &s = ctx->slots[PIPE_DEPTH];
  const uint64_t n = b->n;
  if (n >= 0xFFFFFFFFull) { ctx->last_error = "batch too large (n must be < 2^32-1)"; return SHK_ERR_ARG; }
  const bool paired = b->seq2 != nullptr;
  const uint32_t max_slots = max_read_len ? slots_for_len(max_read_len, ctx->prm.k, paired) : 0;
  const int long_mode = (max_read_len && max_slots <= fast_kernel_max_slots()) ? LONG_NONE_EXPECTED : LONG_UNKNOWN;
  int rc;
  const uint64_t hint_groups = (((uint64_t)max_read_len + 7) >> 3) * (paired ? 2 : 1);   // (no bound given: the table kernel queues what it cannot stage)
  // (UNI_ASK_DEVICE: the lengths passed here only size the plan table of a batch that turns out ragged: the caller's bound per mate)
  if ((rc = enqueue_classify(ctx, s, b, max_slots, long_mode, 0, 0, wc == nullptr, UNI_ASK_DEVICE, max_read_len, paired ? max_read_len : 0,
                             hint_groups <= uni_kernel_max_groups(max_slots))))
    return rc;
  SHK_HIP(ctx, hipStreamSynchronize(st));
  bool redone = false;
  if ((rc = finish_classify(ctx, s, long_mode == LONG_NONE_EXPECTED, wc == nullptr, &redone))) return rc;
  const uint64_t n_assoc = ((uint64_t)s.h_counters[CTR_ASSOC_HI] << 32) | s.h_counters[CTR_ASSOC_LO];

  if (wc) {
    // measurement only: re-run every read through the general kernel with the
    // exact work counters switched on (results are rewritten with equal values)
    ClassifyParams p = s.p;
    unsigned n_waves = 0;
    if ((rc = size_scratch(ctx, p, s.gen_slots, n, &n_waves))) return rc;
    SHK_HIP(ctx, hipMemsetAsync(s.d_counters, 0, CTR_WORDS * sizeof(uint32_t), st));
    SHK_HIP(ctx, hipMemsetAsync(ctx->d_work_counters, 0, 4 * sizeof(unsigned long long), st));
    p.work = nullptr;
    p.n_work = n;
    p.work_count = nullptr;
    p.work_counters = ctx->d_work_counters;
    if ((rc = launch_classify_general(ctx, p, false, n_waves, st))) return rc;
    unsigned long long h[4] = {0, 0, 0, 0};
    SHK_HIP(ctx, hipMemcpyAsync(h, ctx->d_work_counters, sizeof(h), hipMemcpyDeviceToHost, st));
    SHK_HIP(ctx, hipStreamSynchronize(st));
    wc->n_kmers = h[0]; wc->n_hits = h[1]; wc->n_list_ids = h[2]; wc->n_bases = h[3];
  }

  ctx->last.last_n_reads = n;
  ctx->last.last_n_long = s.h_counters[CTR_LONG];
  ctx->last.last_n_tie = s.h_counters[CTR_TIE];
  note_verdict(ctx, s.h_counters[CTR_VERDICT]);
#ifdef SHK_ANCH_STATS   // (experiments: reads the anchored extension settled early / settled after probing its open slots)
  ctx->last.last_n_long = s.h_counters[CTR_UNUSED3];
  ctx->last.last_n_tie = s.h_counters[CTR_UNUSED5];
#endif
  ctx->last.last_n_assoc = n_assoc;
  res->n = n;
  res->gene_off = s.d_gene_off;
  res->gene_ids = s.d_gene_ids;
  res->n_assoc = n_assoc;
  return SHK_OK;
}

}  // namespace shk

// ---- RCCL, loaded on first use so that single-GPU users never pay for (or conflict over) it ----
namespace shk {
typedef void *rccl_comm_t;
struct rccl_uid { char internal[128]; };   // ncclUniqueId
struct RcclApi {
  void *lib = nullptr;
  int (*CommInitAll)(rccl_comm_t *, int, const int *) = nullptr;
  int (*CommInitRank)(rccl_comm_t *, int, rccl_uid, int) = nullptr;
  int (*GetUniqueId)(rccl_uid *) = nullptr;
  int (*CommDestroy)(rccl_comm_t) = nullptr;
  int (*CommCount)(rccl_comm_t, int *) = nullptr;
  int (*CommUserRank)(rccl_comm_t, int *) = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, rccl_comm_t, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  bool ok = false;
};
static RcclApi &rccl()
{
  static RcclApi api;
  if (!api.lib) {
    api.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!api.lib) api.lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (api.lib) {
      api.CommInitAll = (int (*)(rccl_comm_t *, int, const int *))dlsym(api.lib, "ncclCommInitAll");
      api.CommInitRank = (int (*)(rccl_comm_t *, int, rccl_uid, int))dlsym(api.lib, "ncclCommInitRank");
      api.GetUniqueId = (int (*)(rccl_uid *))dlsym(api.lib, "ncclGetUniqueId");
      api.CommDestroy = (int (*)(rccl_comm_t))dlsym(api.lib, "ncclCommDestroy");
      api.CommCount = (int (*)(rccl_comm_t, int *))dlsym(api.lib, "ncclCommCount");
      api.CommUserRank = (int (*)(rccl_comm_t, int *))dlsym(api.lib, "ncclCommUserRank");
      api.AllReduce = (int (*)(const void *, void *, size_t, int, int, rccl_comm_t, hipStream_t))dlsym(api.lib, "ncclAllReduce");
      api.GroupStart = (int (*)())dlsym(api.lib, "ncclGroupStart");
      api.GroupEnd = (int (*)())dlsym(api.lib, "ncclGroupEnd");
      api.ok = api.CommInitAll && api.CommInitRank && api.GetUniqueId && api.CommDestroy && api.CommCount && api.CommUserRank && api.AllReduce &&
               api.GroupStart && api.GroupEnd;
    }
  }
  return api;
}

// RCCL may print a version banner on stdout when it initialises; stdout is the ssv stream of the CLI
// (ReadOutput.hpp:43), so anything RCCL prints during init is sent to stderr instead
template <typename F>
static int with_stdout_on_stderr(F f)
{
  fflush(stdout);
  const int saved_stdout = dup(1);
  if (saved_stdout >= 0) (void)dup2(2, 1);
  const int rc = f();
  if (saved_stdout >= 0) {
    fflush(stdout);
    (void)dup2(saved_stdout, 1);
    close(saved_stdout);
  }
  return rc;
}

static void dist_release(Ctx *ctx)
{
  if (ctx->dist_comm) { (void)rccl().CommDestroy(ctx->dist_comm); ctx->dist_comm = nullptr; }
  if (ctx->group_comm) { (void)rccl().CommDestroy(ctx->group_comm); ctx->group_comm = nullptr; }
}
}  // namespace shk

using namespace shk;

extern "C" {

const char *shk_version(void) { return "sharkhip 0.1 (gfx950)"; }

const char *shk_strerror(int code)
{
  switch (code) {
  case SHK_OK: return "ok";
  case SHK_ERR_ARG: return "invalid argument";
  case SHK_ERR_STATE: return "call not allowed in the current mode";
  case SHK_ERR_HIP: return "HIP runtime error";
  case SHK_ERR_NOMEM: return "out of memory";
  case SHK_ERR_TOO_MANY_GENES: return "more than 65536 reference sequences";
  case SHK_ERR_INDEX_TOO_LARGE: return "index exceeds the reference's 2^31 limits";
  case SHK_ERR_NO_DEVICE: return "no HIP device";
  default: return "unknown error";
  }
}

const char *shk_last_error(const shk_ctx *ctx) { return ctx ? ctx->last_error.c_str() : ""; }

int shk_create(const shk_params *prm, shk_ctx **out)
{
  if (!prm || !out) return SHK_ERR_ARG;
  *out = nullptr;
  if (prm->k == 0 || prm->k > 31) return SHK_ERR_ARG;                  // argument_parser.hpp:115
  if (!(prm->c >= 0.0 && prm->c <= 1.0)) return SHK_ERR_ARG;           // :124
  if (prm->min_quality < 0) return SHK_ERR_ARG;                        // :138
  if (prm->bf_bits == 0) return SHK_ERR_ARG;
  int n_dev = 0;
  {
    // the first HIP call of a process brings the runtime up (tens of milliseconds).  N workers created on N threads (shark --gpus N) all
    // arrive here at once: behind one mutex the first of them initialises the runtime and the others sleep -- racing into the runtime's
    // own initialisation they took three times as long together (measured: "contexts created" 0.22 s for two workers against 0.10 s)
    static std::mutex init_m;
    std::lock_guard<std::mutex> l(init_m);
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return SHK_ERR_NO_DEVICE;
  }
  if (prm->device < 0 || prm->device >= n_dev) return SHK_ERR_ARG;
  shk_ctx *ctx = new (std::nothrow) shk_ctx();
  if (!ctx) return SHK_ERR_NOMEM;
  ctx->prm = *prm;
  ctx->prm.single = prm->single ? 1 : 0;
  ctx->q8 = (int8_t)(uint8_t)(prm->min_quality & 0xFF);                // static_cast<char>(mq), argument_parser.hpp:144
  {
    const char *e = getenv("SHK_FORCE_GENERIC");
    ctx->env_force_generic = e && e[0] == '1';
    ctx->env_big_lds_always = getenv("SHK_BIG_LDS_ALWAYS") != nullptr;
    { const char *nt = getenv("SHK_KTAB_NT"); ctx->env_ktab_nt = nt && nt[0] == '1'; ctx->env_ktab_plain = nt && nt[0] == '0'; }
    ctx->env_ktab_always = getenv("SHK_KTAB") != nullptr;
    ctx->env_anchor_always = getenv("SHK_ANCHOR_ALWAYS") != nullptr;
    ctx->env_no_pre_verdict = getenv("SHK_NO_PRE_VERDICT") != nullptr;
    ctx->env_no_tri = getenv("SHK_NO_TRI") != nullptr;
    ctx->env_no_tro = getenv("SHK_NO_TRO") != nullptr;
    ctx->env_force_tro = getenv("SHK_FORCE_TRO") != nullptr;
    ctx->env_tile_first = getenv("SHK_TILE_FIRST") ? (getenv("SHK_TILE_FIRST")[0] == '0' ? -1 : 1) : 0;
    if (const char *f = getenv("SHK_CLS_MIN_FILL")) { ctx->env_cls_min_fill = (uint32_t)strtoul(f, nullptr, 10); ctx->env_cls_always = true; }
  }
  auto fail = [&](int rc) { shk_destroy(ctx); return rc; };
  // (SHK_TRACE_CREATE=1: where the time of this call goes, on stderr -- tools/workers_start.py)
  const bool trace = getenv("SHK_TRACE_CREATE") != nullptr;
  const auto t_start = std::chrono::steady_clock::now();
  auto stamp = [&](const char *what) {
    if (trace) fprintf(stderr, "[shk/create dev %d ctx %p] %-28s %8.3f ms\n", prm->device, (void *)ctx, what,
                       std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count());
  };
  stamp("device count");
#define CR_HIP(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) return fail(e__ == hipErrorOutOfMemory ? SHK_ERR_NOMEM : SHK_ERR_HIP); } while (0)
  CR_HIP(hipSetDevice(prm->device));
  stamp("hipSetDevice");
  CR_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
  CR_HIP(hipStreamCreateWithFlags(&ctx->h2d_stream, hipStreamNonBlocking));
  CR_HIP(hipStreamCreateWithFlags(&ctx->d2h_stream, hipStreamNonBlocking));
  stamp("three streams");
  DeviceIndex &ix = ctx->idx;
  ix.bf_bits = prm->bf_bits;
  ix.pow2 = (prm->bf_bits & (prm->bf_bits - 1)) == 0;
  ix.bf_words64 = ((prm->bf_bits + 511) / 512) * 8;
  // BF::BF(size): size zero bits (bloomfilter.h:48-53)
  CR_HIP(hipMalloc((void **)&ix.bf64, ix.bf_words64 * sizeof(uint64_t)));
  stamp("filter allocated");
  CR_HIP(hipMemsetAsync(ix.bf64, 0, ix.bf_words64 * sizeof(uint64_t), ctx->stream));
  stamp("filter clear enqueued");
  for (Slot &sl : ctx->slots)
    if (slot_init(ctx, sl) != SHK_OK) return fail(SHK_ERR_HIP);
  stamp("slots");
  CR_HIP(hipMalloc((void **)&ctx->d_gene_counts, 65536 * sizeof(unsigned long long)));
  CR_HIP(hipMemsetAsync(ctx->d_gene_counts, 0, 65536 * sizeof(unsigned long long), ctx->stream));
  CR_HIP(hipMalloc((void **)&ctx->d_gene_totals, 65536 * sizeof(unsigned long long)));
  CR_HIP(hipMalloc((void **)&ctx->d_work_counters, 4 * sizeof(unsigned long long)));
  stamp("counters");
  CR_HIP(hipStreamSynchronize(ctx->stream));
  stamp("synchronised");
#undef CR_HIP
  *out = ctx;
  return SHK_OK;
}

void shk_destroy(shk_ctx *ctx)
{
  if (!ctx) return;
  (void)hipSetDevice(ctx->prm.device);
  // tickets that were submitted and never waited for (a caller bailing out) may still have copies in flight on the copy
  // streams -- into slot buffers that are about to be freed, out of caller memory that may be released right after this call
  if (ctx->h2d_stream) (void)hipStreamSynchronize(ctx->h2d_stream);
  if (ctx->d2h_stream) (void)hipStreamSynchronize(ctx->d2h_stream);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  free_index(ctx->idx);
  for (Slot &sl : ctx->slots) slot_free(sl);
  hipFree(ctx->d_scratch); hipFree(ctx->d_gene_counts); hipFree(ctx->d_gene_totals); hipFree(ctx->d_work_counters);
  dist_release(ctx);
  for (auto e : ctx->ev_start) (void)hipEventDestroy(e);
  for (auto e : ctx->ev_stop) (void)hipEventDestroy(e);
  for (auto e : ctx->ev_pre) (void)hipEventDestroy(e);
  if (ctx->h2d_stream) (void)hipStreamDestroy(ctx->h2d_stream);
  if (ctx->d2h_stream) (void)hipStreamDestroy(ctx->d2h_stream);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

int shk_ref_add(shk_ctx *ctx, const char *seq, uint64_t len)
{
  if (!ctx || (!seq && len)) return SHK_ERR_ARG;
  if (ctx->mode != 0) return SHK_ERR_STATE;
  if (ctx->n_records >= 0x7FFFFFFFull) return SHK_ERR_TOO_MANY_GENES;
  try {
    ctx->ref_bytes.insert(ctx->ref_bytes.end(), seq, seq + len);
    ctx->ref_off.push_back(ctx->ref_bytes.size());
  } catch (...) {
    return SHK_ERR_NOMEM;
  }
  ctx->n_records++;
  return SHK_OK;
}

int shk_ref_finalize(shk_ctx *ctx)
{
  if (!ctx) return SHK_ERR_ARG;
  if (ctx->mode != 0) return SHK_ERR_STATE;
  SHK_HIP(ctx, hipSetDevice(ctx->prm.device));
  TraceRange tr("shk_ref_finalize: index build");
  const int rc = build_index(ctx);
  if (rc != SHK_OK) return rc;
  ctx->mode = 2;
  std::vector<char>().swap(ctx->ref_bytes);
  return SHK_OK;
}

int shk_index_info_get(const shk_ctx *ctx, shk_index_info *info)
{
  if (!ctx || !info) return SHK_ERR_ARG;
  info->n_records = ctx->n_records;
  info->nidx = ctx->nidx;
  info->bf_bits = ctx->idx.bf_bits;
  info->n_set_bits = ctx->idx.n_set;
  info->tot_idx = ctx->idx.tot_idx;
  info->n_ref_kmers = ctx->n_ref_kmers;
  return SHK_OK;
}

const char *shk_probe_mode(const shk_ctx *ctx)
{
  if (!ctx || ctx->mode != 2) return "";
  return probe_mode_name(ctx);
}

const char *shk_last_kernel(const shk_ctx *ctx) { return ctx ? ctx->last_kernel : ""; }

int shk_index_copy_bf(const shk_ctx *cctx, uint64_t *words, uint64_t n_words)
{
  shk_ctx *ctx = const_cast<shk_ctx *>(cctx);
  if (!ctx || !words) return SHK_ERR_ARG;
  if (ctx->mode != 2) return SHK_ERR_STATE;
  const uint64_t have = (ctx->idx.bf_bits + 63) / 64;
  if (n_words > have) return SHK_ERR_ARG;
  SHK_HIP(ctx, hipSetDevice(ctx->prm.device));
  SHK_HIP(ctx, hipMemcpy(words, ctx->idx.bf64, n_words * sizeof(uint64_t), hipMemcpyDeviceToHost));
  return SHK_OK;
}

int shk_index_copy_lists(const shk_ctx *cctx, uint32_t *offsets, uint16_t *ids)
{
  shk_ctx *ctx = const_cast<shk_ctx *>(cctx);
  if (!ctx || !offsets) return SHK_ERR_ARG;
  if (ctx->mode != 2) return SHK_ERR_STATE;
  SHK_HIP(ctx, hipSetDevice(ctx->prm.device));
  std::vector<ListEntry> ent;
  try { ent.resize(ctx->idx.n_set + 1); } catch (...) { return SHK_ERR_NOMEM; }
  SHK_HIP(ctx, hipMemcpy(ent.data(), ctx->idx.ent, ent.size() * sizeof(ListEntry), hipMemcpyDeviceToHost));
  for (size_t r = 0; r < ent.size(); ++r) offsets[r] = ent[r].start;
  if (ctx->idx.tot_idx && ids)
    SHK_HIP(ctx, hipMemcpy(ids, ctx->idx.ids, ctx->idx.tot_idx * sizeof(uint16_t), hipMemcpyDeviceToHost));
  return SHK_OK;
}

static int check_batch(const shk_ctx *ctx, const shk_batch *b)
{
  if (!b) return SHK_ERR_ARG;
  if (b->n == 0) return SHK_OK;
  if (!b->seq1 || !b->off1) return SHK_ERR_ARG;
  if ((b->seq2 == nullptr) != (b->off2 == nullptr)) return SHK_ERR_ARG;
  if (ctx->q8 != 0 && (!b->qual1 || (b->seq2 && !b->qual2))) return SHK_ERR_ARG;
  return SHK_OK;
}

int shk_classify_device(shk_ctx *ctx, const shk_batch *batch, uint32_t max_read_len, shk_result *result)
{
  if (!ctx || !result) return SHK_ERR_ARG;
  if (ctx->mode != 2) return SHK_ERR_STATE;
  int rc = check_batch(ctx, batch);
  if (rc) return rc;
  SHK_HIP(ctx, hipSetDevice(ctx->prm.device));
  TraceRange tr("shk_classify_device");
  return classify_resident(ctx, batch, max_read_len, result);
}

// ---- host batches: a pipeline of PIPE_DEPTH batches per context ---------------------------------
// submit(i): host scan of the offsets, H2D on the copy stream, every kernel on the compute stream behind an event,
//            no host round trip.  wait(i): the batch's event, then D2H of exactly its results into the slot's pinned
//            buffers.  While the host waits for batch i, the H2D of batch i+1 and i+2 overlaps the kernels of batch i
//            (the reference overlaps split / analyze / output across its worker threads, main.cpp:66-77).
int shk_classify_submit(shk_ctx *ctx, const shk_batch *b, uint64_t *ticket)
{
  if (!ctx || !ticket) return SHK_ERR_ARG;
  if (ctx->mode != 2) return SHK_ERR_STATE;
  int rc = check_batch(ctx, b);
  if (rc) return rc;
  SHK_HIP(ctx, hipSetDevice(ctx->prm.device));
  TraceRange tr("shk_classify_submit");
  Slot &s = ctx->slots[(ctx->next_ticket - 1) % PIPE_DEPTH];
  if (s.ticket != 0 && !s.waited) {
    ctx->last_error = "pipeline full: shk_classify_wait the oldest ticket before submitting another batch";
    return SHK_ERR_STATE;
  }
  const uint64_t n = b->n;
  if (n >= 0xFFFFFFFFull) { ctx->last_error = "batch too large (n must be < 2^32-1)"; return SHK_ERR_ARG; }
  const bool paired = b->seq2 != nullptr;
  const uint32_t k = ctx->prm.k;
  // one pass over the offsets: validation, the longest mate (selects the kernel specialisation), and whether every
  // read has the same length (then the offsets are generated on the device instead of crossing PCIe)
  uint64_t max1 = 0, max2 = 0, bad = 0;
  const uint64_t f1 = n ? b->off1[1] - b->off1[0] : 0, f2 = (n && paired) ? b->off2[1] - b->off2[0] : 0;
  uint64_t nonuni = n ? (b->off1[0] | (paired ? b->off2[0] : 0)) : 0;
  for (uint64_t i = 0; i < n; ++i) {
    const uint64_t a = b->off1[i], e = b->off1[i + 1];
    bad |= (uint64_t)(e < a);
    const uint64_t l = e - a;
    max1 = l > max1 ? l : max1;
    nonuni |= l ^ f1;
  }
  if (paired)
    for (uint64_t i = 0; i < n; ++i) {
      const uint64_t a = b->off2[i], e = b->off2[i + 1];
      bad |= (uint64_t)(e < a);
      const uint64_t l = e - a;
      max2 = l > max2 ? l : max2;
      nonuni |= l ^ f2;
    }
  if (bad) return SHK_ERR_ARG;
  const bool uniform = n && !nonuni;
  const uint64_t bytes1 = n ? b->off1[n] : 0;
  const uint64_t bytes2 = (n && paired) ? b->off2[n] : 0;
  // slot count of the longest read decides the specialisation; only when even the largest one is too small do reads
  // go to the general kernel, and then the host counts them here (so the device never has to be asked)
  uint32_t max_slots = slots_of_read(max1, max2, k);   // an upper bound for every read: the slot count is monotone in both lengths
  // Reads with more slots than the largest specialisation go to the general kernel; the host counts them here with the
  // kernels' own criterion (classify.hip), so the device never has to be asked.  Only when the bound says there are any.
  uint32_t n_long = 0, long_slots = 0;
  const uint64_t groups_bound = ((max1 + 7) >> 3) + ((max2 + 7) >> 3);
  const bool groups_fit = groups_bound <= uni_kernel_max_groups(max_slots);   // else the batch runs on classify_fast_kernel, which stages any number of groups
  if (max_slots > fast_kernel_max_slots()) {
    const uint32_t cap = 64 * fast_kernel_unroll(std::min(max_slots, fast_kernel_max_slots()));
    for (uint64_t i = 0; i < n; ++i) {
      const uint64_t l1 = b->off1[i + 1] - b->off1[i], l2 = paired ? b->off2[i + 1] - b->off2[i] : 0;
      const uint32_t ns = slots_of_read(l1, l2, k);
      if (ns > cap) {
        long_slots = std::max(long_slots, ns);
        ++n_long;
      }
    }
    if (max_slots > fast_kernel_max_slots()) max_slots = fast_kernel_max_slots();
  }

  hipStream_t up = ctx->h2d_stream, st = ctx->stream;
  shk_batch d{};
  d.n = n;
  if ((rc = ensure_capacity(ctx, &s.d_seq1, &s.cap_seq1, bytes1 + 16))) return rc;
  if ((rc = ensure_capacity(ctx, &s.d_off1, &s.cap_off1, n + 1))) return rc;
  if (paired) {
    if ((rc = ensure_capacity(ctx, &s.d_seq2, &s.cap_seq2, bytes2 + 16))) return rc;
    if ((rc = ensure_capacity(ctx, &s.d_off2, &s.cap_off2, n + 1))) return rc;
  }
  const bool hasq = ctx->q8 != 0;
  if (hasq) {
    if ((rc = ensure_capacity(ctx, &s.d_qual1, &s.cap_qual1, bytes1 + 16))) return rc;
    if (paired && (rc = ensure_capacity(ctx, &s.d_qual2, &s.cap_qual2, bytes2 + 16))) return rc;
  }
  if (n) {
    SHK_HIP(ctx, hipMemcpyAsync(s.d_seq1, b->seq1, bytes1, hipMemcpyHostToDevice, up));
    if (paired) SHK_HIP(ctx, hipMemcpyAsync(s.d_seq2, b->seq2, bytes2, hipMemcpyHostToDevice, up));
    if (hasq) {
      SHK_HIP(ctx, hipMemcpyAsync(s.d_qual1, b->qual1, bytes1, hipMemcpyHostToDevice, up));
      if (paired) SHK_HIP(ctx, hipMemcpyAsync(s.d_qual2, b->qual2, bytes2, hipMemcpyHostToDevice, up));
    }
    if (uniform) {
      if ((rc = launch_fill_offsets(s.d_off1, n + 1, f1, st))) return rc;
      if (paired && (rc = launch_fill_offsets(s.d_off2, n + 1, f2, st))) return rc;
    } else {
      SHK_HIP(ctx, hipMemcpyAsync(s.d_off1, b->off1, (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, up));
      if (paired) SHK_HIP(ctx, hipMemcpyAsync(s.d_off2, b->off2, (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, up));
    }
  }
  SHK_HIP(ctx, hipEventRecord(s.ev_h2d, up));
  SHK_HIP(ctx, hipStreamWaitEvent(st, s.ev_h2d, 0));
  d.seq1 = (const char *)s.d_seq1; d.off1 = s.d_off1;
  if (paired) { d.seq2 = (const char *)s.d_seq2; d.off2 = s.d_off2; }
  if (hasq) { d.qual1 = (const char *)s.d_qual1; if (paired) d.qual2 = (const char *)s.d_qual2; }
  s.host_batch = true;
  s.long_speculative = false;
  const bool uni_fits = uniform && n_long == 0 && groups_fit;
  if ((rc = enqueue_classify(ctx, s, &d, max_slots, LONG_KNOWN, n_long, long_slots, true, uni_fits ? UNI_YES : UNI_NO, (uint32_t)max1, (uint32_t)max2,
                             groups_fit)))
    return rc;
  s.ticket = ctx->next_ticket++;
  s.waited = false;
  *ticket = s.ticket;
  return SHK_OK;
}

// the device-resident entry point as a pipeline (no host synchronisation): see the header
int shk_classify_device_submit(shk_ctx *ctx, const shk_batch *b, uint32_t max_read_len, uint32_t uniform_len1, uint32_t uniform_len2, uint64_t *ticket)
{
  if (!ctx || !ticket || max_read_len == 0) return SHK_ERR_ARG;
  if (ctx->mode != 2) return SHK_ERR_STATE;
  int rc = check_batch(ctx, b);
  if (rc) return rc;
  SHK_HIP(ctx, hipSetDevice(ctx->prm.device));
  TraceRange tr("shk_classify_device_submit");
  Slot &s = ctx->slots[(ctx->next_ticket - 1) % PIPE_DEPTH];
  if (s.ticket != 0 && !s.waited) {
    ctx->last_error = "pipeline full: shk_classify_wait the oldest ticket before submitting another batch";
    return SHK_ERR_STATE;
  }
  const uint64_t n = b->n;
  if (n >= 0xFFFFFFFFull) { ctx->last_error = "batch too large (n must be < 2^32-1)"; return SHK_ERR_ARG; }
  const bool paired = b->seq2 != nullptr;
  if (uniform_len1 > max_read_len || uniform_len2 > max_read_len || (!paired && uniform_len2)) {
    ctx->last_error = "uniform_len1 / uniform_len2 exceed max_read_len, or uniform_len2 given for a single-end batch";
    return SHK_ERR_ARG;
  }
  uint32_t max_slots = slots_for_len(max_read_len, ctx->prm.k, paired);
  // a bound beyond the largest specialisation: the device would have to be asked how many reads do not fit
  if (max_slots > fast_kernel_max_slots()) { ctx->last_error = "max_read_len beyond the kernels' specialisations: use shk_classify_device"; return SHK_ERR_ARG; }
  const uint64_t hint_groups = (((uint64_t)max_read_len + 7) >> 3) * (paired ? 2 : 1);
  const bool groups_fit = hint_groups <= uni_kernel_max_groups(max_slots);
  const bool vouched = n != 0 && uniform_len1 != 0 && (paired ? uniform_len2 != 0 : true) && groups_fit;
  s.host_batch = false;
  s.long_speculative = true;
  if ((rc = enqueue_classify(ctx, s, b, max_slots, LONG_NONE_EXPECTED, 0, 0, true, vouched ? UNI_YES : UNI_ASK_DEVICE,
                             vouched ? uniform_len1 : max_read_len, vouched ? uniform_len2 : (paired ? max_read_len : 0), groups_fit)))
    return rc;
  s.ticket = ctx->next_ticket++;
  s.waited = false;
  *ticket = s.ticket;
  return SHK_OK;
}

int shk_classify_wait(shk_ctx *ctx, uint64_t ticket, shk_result *result)
{
  if (!ctx || !result || ticket == 0) return SHK_ERR_ARG;
  SHK_HIP(ctx, hipSetDevice(ctx->prm.device));
  TraceRange tr("shk_classify_wait");
  Slot &s = ctx->slots[(ticket - 1) % PIPE_DEPTH];
  if (s.ticket != ticket || s.waited) { ctx->last_error = "unknown or already waited ticket"; return SHK_ERR_STATE; }
  SHK_HIP(ctx, hipEventSynchronize(s.ev_done));
  bool redone = false;
  int rc = finish_classify(ctx, s, s.long_speculative, true, &redone);
  if (rc) { s.waited = true; return rc; }
  if (!s.host_batch && s.h_counters[CTR_VOUCH_BAD]) {
    ctx->last_error = "uniform_len1 / uniform_len2 do not describe the batch: its offsets are not r * length (shk_classify_device_submit)";
    s.waited = true;
    return SHK_ERR_ARG;
  }
  // (finish_classify re-ran the tail when it had to, and the tail publishes the results again)
  const uint64_t n = s.n;
  const uint64_t n_assoc = ((uint64_t)s.h_counters[CTR_ASSOC_HI] << 32) | s.h_counters[CTR_ASSOC_LO];
  if (s.h_counters[CTR_OVERFLOW] || (s.host_batch && n_assoc > s.cap_h_gene_ids)) { ctx->last_error = "result publication failed"; s.waited = true; return SHK_ERR_HIP; }
  s.waited = true;
  ctx->last.last_n_reads = n;
  ctx->last.last_n_long = s.h_counters[CTR_LONG];
  ctx->last.last_n_tie = s.h_counters[CTR_TIE];
  note_verdict(ctx, s.h_counters[CTR_VERDICT]);
  ctx->last.last_n_assoc = n_assoc;
  result->n = n;
  result->gene_off = s.host_batch ? s.h_gene_off : s.d_gene_off;     // (a device-resident ticket: device pointers)
  result->gene_ids = s.host_batch ? s.h_gene_ids : s.d_gene_ids;
  result->n_assoc = n_assoc;
  return SHK_OK;
}

int shk_classify(shk_ctx *ctx, const shk_batch *b, shk_result *result)
{
  if (!ctx || !result) return SHK_ERR_ARG;
  uint64_t t = 0;
  int rc = shk_classify_submit(ctx, b, &t);
  if (rc) return rc;
  return shk_classify_wait(ctx, t, result);
}

int shk_gene_counts(shk_ctx *ctx, uint64_t *counts, uint32_t n)
{
  if (!ctx || !counts || n > 65536) return SHK_ERR_ARG;
  SHK_HIP(ctx, hipSetDevice(ctx->prm.device));
  SHK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  SHK_HIP(ctx, hipMemcpy(counts, ctx->d_gene_counts, (size_t)n * sizeof(uint64_t), hipMemcpyDeviceToHost));
  return SHK_OK;
}

int shk_gene_counts_reset(shk_ctx *ctx)
{
  if (!ctx) return SHK_ERR_ARG;
  SHK_HIP(ctx, hipSetDevice(ctx->prm.device));
  SHK_HIP(ctx, hipMemsetAsync(ctx->d_gene_counts, 0, 65536 * sizeof(unsigned long long), ctx->stream));
  SHK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SHK_OK;
}

// ---- the path's one exchange step: all-reduce of the per-gene counters over RCCL ------------------
// Both forms reduce INTO the contexts' separate totals buffers: the per-GPU counters stay local, so classifying more
// reads and reducing again gives the totals again (not totals times the number of GPUs).

// contexts that share a device: their counters are added on that device (RCCL takes a device once per communicator)
__global__ void gene_counts_add_kernel(unsigned long long *__restrict__ dst, const unsigned long long *__restrict__ src, uint32_t n)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] += src[i];
}

// one process, several contexts (`shark --gpus N [--devices ...]`): normally one context per GPU; contexts that share a GPU
// (--devices 0,0: the N-context code paths rehearsed on one device) are summed on it first, and the collective runs over
// the distinct devices' first contexts
int shk_gene_counts_allreduce(shk_ctx **ctxs, int n_ctx, uint64_t *totals, uint32_t n)
{
  if (!ctxs || n_ctx < 1 || n > 65536) return SHK_ERR_ARG;
  for (int i = 0; i < n_ctx; ++i)
    if (!ctxs[i]) return SHK_ERR_ARG;
  shk_ctx *c0 = ctxs[0];
  for (int i = 0; i < n_ctx; ++i) {
    SHK_HIP(ctxs[i], hipSetDevice(ctxs[i]->prm.device));
    SHK_HIP(ctxs[i], hipStreamSynchronize(ctxs[i]->stream));
  }
  // leaders: the first context of every distinct device, in the order given
  std::vector<int> leader_of((size_t)n_ctx, 0), leaders;
  for (int i = 0; i < n_ctx; ++i) {
    int l = -1;
    for (int j : leaders)
      if (ctxs[j]->prm.device == ctxs[i]->prm.device) { l = j; break; }
    if (l < 0) { l = i; leaders.push_back(i); }
    leader_of[(size_t)i] = l;
  }
  const int n_lead = (int)leaders.size();
  const bool shared = n_lead != n_ctx;
  const size_t bytes = 65536 * sizeof(unsigned long long);
  if (shared) {
    // per device: totals of the leader = sum of the counters of the contexts on it (the streams are idle: synchronised above)
    for (int l : leaders) {
      shk_ctx *lc = ctxs[l];
      SHK_HIP(lc, hipSetDevice(lc->prm.device));
      SHK_HIP(lc, hipMemcpyAsync(lc->d_gene_totals, lc->d_gene_counts, bytes, hipMemcpyDeviceToDevice, lc->stream));
      for (int i = 0; i < n_ctx; ++i)
        if (i != l && leader_of[(size_t)i] == l)
          gene_counts_add_kernel<<<65536 / 256, 256, 0, lc->stream>>>(lc->d_gene_totals, ctxs[i]->d_gene_counts, 65536);
      SHK_HIP(lc, hipGetLastError());
      SHK_HIP(lc, hipStreamSynchronize(lc->stream));
    }
  }
  const char *force = getenv("SHK_FORCE_RCCL");
  const unsigned long long *src = shared ? c0->d_gene_totals : c0->d_gene_counts;
  if (n_lead > 1 || (force && force[0] == '1')) {
    RcclApi &api = rccl();
    if (!api.ok) { c0->last_error = "librccl.so.1 could not be loaded"; return SHK_ERR_HIP; }
    std::vector<int> devs((size_t)n_lead);
    for (int i = 0; i < n_lead; ++i) devs[(size_t)i] = ctxs[leaders[(size_t)i]]->prm.device;
    // the communicators of a group of contexts are created once and live in the (leading) contexts
    bool have = true;
    for (int l : leaders) have = have && ctxs[l]->group_comm && ctxs[l]->group_devs == devs;
    if (!have) {
      for (int i = 0; i < n_ctx; ++i)
        if (ctxs[i]->group_comm) { (void)api.CommDestroy(ctxs[i]->group_comm); ctxs[i]->group_comm = nullptr; }
      std::vector<rccl_comm_t> comms((size_t)n_lead, nullptr);
      const int init_rc = with_stdout_on_stderr([&] { return api.CommInitAll(comms.data(), n_lead, devs.data()); });
      if (init_rc != 0) { c0->last_error = "ncclCommInitAll failed"; return SHK_ERR_HIP; }
      for (int i = 0; i < n_lead; ++i) { ctxs[leaders[(size_t)i]]->group_comm = comms[(size_t)i]; ctxs[leaders[(size_t)i]]->group_devs = devs; }
      // RCCL sets its channels up on a communicator's FIRST collective (of the order of the whole reduction of 512 KiB, or more): one
      // throw-away all-reduce (of a scratch-free kind: the totals buffers onto themselves, before they hold anything when no device is
      // shared; with shared devices their sums are made again below), so that the reduction a caller times is a steady-state one
      int wrc = api.GroupStart();
      for (int i = 0; i < n_lead && wrc == 0; ++i) {
        shk_ctx *lc = ctxs[leaders[(size_t)i]];
        (void)hipSetDevice(devs[(size_t)i]);
        wrc = api.AllReduce(lc->d_gene_totals, lc->d_gene_totals, 65536, 5, 0, lc->group_comm, lc->stream);
      }
      if (wrc == 0) wrc = api.GroupEnd();
      for (int i = 0; i < n_lead; ++i) {
        (void)hipSetDevice(devs[(size_t)i]);
        (void)hipStreamSynchronize(ctxs[leaders[(size_t)i]]->stream);
      }
      if (wrc != 0) { c0->last_error = "ncclAllReduce (warm-up) failed"; return SHK_ERR_HIP; }
      if (shared) {   // the warm-up has overwritten the per-device sums: make them again
        for (int l : leaders) {
          shk_ctx *lc = ctxs[l];
          SHK_HIP(lc, hipSetDevice(lc->prm.device));
          SHK_HIP(lc, hipMemcpyAsync(lc->d_gene_totals, lc->d_gene_counts, bytes, hipMemcpyDeviceToDevice, lc->stream));
          for (int i = 0; i < n_ctx; ++i)
            if (i != l && leader_of[(size_t)i] == l)
              gene_counts_add_kernel<<<65536 / 256, 256, 0, lc->stream>>>(lc->d_gene_totals, ctxs[i]->d_gene_counts, 65536);
          SHK_HIP(lc, hipGetLastError());
          SHK_HIP(lc, hipStreamSynchronize(lc->stream));
        }
      }
    }
    int rc = api.GroupStart();
    for (int i = 0; i < n_lead && rc == 0; ++i) {
      shk_ctx *lc = ctxs[leaders[(size_t)i]];
      (void)hipSetDevice(devs[(size_t)i]);
      // ncclUint64 = 5, ncclSum = 0; 65 536 x 8 B per GPU (in place over the per-device sums when contexts share a device)
      rc = api.AllReduce(shared ? lc->d_gene_totals : lc->d_gene_counts, lc->d_gene_totals, 65536, 5, 0, lc->group_comm, lc->stream);
    }
    if (rc == 0) rc = api.GroupEnd();
    for (int i = 0; i < n_lead; ++i) {
      (void)hipSetDevice(devs[(size_t)i]);
      (void)hipStreamSynchronize(ctxs[leaders[(size_t)i]]->stream);
    }
    if (rc != 0) { c0->last_error = "ncclAllReduce failed"; return SHK_ERR_HIP; }
    src = c0->d_gene_totals;
  }
  if (shared) {   // every context's totals buffer holds the totals, as with one context per device
    for (int i = 0; i < n_ctx; ++i) {
      const int l = leader_of[(size_t)i];
      if (l == i) continue;
      SHK_HIP(ctxs[i], hipSetDevice(ctxs[i]->prm.device));
      SHK_HIP(ctxs[i], hipMemcpy(ctxs[i]->d_gene_totals, ctxs[l]->d_gene_totals, bytes, hipMemcpyDeviceToDevice));
    }
  }
  if (totals) {
    SHK_HIP(c0, hipSetDevice(c0->prm.device));
    SHK_HIP(c0, hipMemcpy(totals, src, (size_t)n * sizeof(uint64_t), hipMemcpyDeviceToHost));
  }
  return SHK_OK;
}

// one process per GPU (torch.distributed.run / mpirun style launchers): rank 0 makes an id, the launcher's own
// channel carries its SHK_DIST_ID_BYTES bytes to the other ranks, every rank joins with its context
int shk_dist_unique_id(uint8_t *id)
{
  if (!id) return SHK_ERR_ARG;
  RcclApi &api = rccl();
  if (!api.ok) return SHK_ERR_HIP;
  rccl_uid u;
  if (api.GetUniqueId(&u) != 0) return SHK_ERR_HIP;
  static_assert(sizeof(u) == SHK_DIST_ID_BYTES, "ncclUniqueId is 128 bytes");
  memcpy(id, &u, sizeof(u));
  return SHK_OK;
}

int shk_dist_init(shk_ctx *ctx, const uint8_t *id, int rank, int world)
{
  if (!ctx || !id || world < 1 || rank < 0 || rank >= world) return SHK_ERR_ARG;
  RcclApi &api = rccl();
  if (!api.ok) { ctx->last_error = "librccl.so.1 could not be loaded"; return SHK_ERR_HIP; }
  SHK_HIP(ctx, hipSetDevice(ctx->prm.device));
  if (ctx->dist_comm) { (void)api.CommDestroy(ctx->dist_comm); ctx->dist_comm = nullptr; }
  rccl_uid u;
  memcpy(&u, id, sizeof(u));
  rccl_comm_t comm = nullptr;
  const int rc = with_stdout_on_stderr([&] { return api.CommInitRank(&comm, world, u, rank); });
  if (rc != 0) { ctx->last_error = "ncclCommInitRank failed"; return SHK_ERR_HIP; }
  ctx->dist_comm = comm;
  ctx->dist_rank = rank;
  ctx->dist_world = world;
  // the communicator's first collective sets RCCL's channels up: made here (every rank is in this call), on the totals buffer, so
  // that shk_dist_gene_counts_allreduce is a steady-state collective from its first use on
  if (api.AllReduce(ctx->d_gene_totals, ctx->d_gene_totals, 65536, 5, 0, comm, ctx->stream) != 0) {
    ctx->last_error = "ncclAllReduce (warm-up) failed";
    return SHK_ERR_HIP;
  }
  SHK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SHK_OK;
}

int shk_dist_info(const shk_ctx *ctx, int *rank, int *world)
{
  if (!ctx || !rank || !world) return SHK_ERR_ARG;
  *rank = 0;
  *world = 1;
  if (ctx->dist_comm) {
    // asked of the communicator, not remembered from shk_dist_init's arguments
    if (rccl().CommUserRank(ctx->dist_comm, rank) != 0 || rccl().CommCount(ctx->dist_comm, world) != 0) return SHK_ERR_HIP;
  }
  return SHK_OK;
}

int shk_dist_gene_counts_allreduce(shk_ctx *ctx, uint64_t *totals, uint32_t n)
{
  if (!ctx || n > 65536) return SHK_ERR_ARG;
  SHK_HIP(ctx, hipSetDevice(ctx->prm.device));
  const unsigned long long *src = ctx->d_gene_counts;
  if (ctx->dist_comm) {
    // stream order: behind every classify call enqueued so far
    if (rccl().AllReduce(ctx->d_gene_counts, ctx->d_gene_totals, 65536, 5, 0, ctx->dist_comm, ctx->stream) != 0) {
      ctx->last_error = "ncclAllReduce failed";
      return SHK_ERR_HIP;
    }
    src = ctx->d_gene_totals;
  }
  SHK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (totals) SHK_HIP(ctx, hipMemcpy(totals, src, (size_t)n * sizeof(uint64_t), hipMemcpyDeviceToHost));
  return SHK_OK;
}

int shk_timing_enable(shk_ctx *ctx, int enable)
{
  if (!ctx) return SHK_ERR_ARG;
  ctx->timing = enable != 0;
  ctx->ev_used = 0;
  return SHK_OK;
}

int shk_timing_get(shk_ctx *ctx, shk_timing *t)
{
  if (!ctx || !t) return SHK_ERR_ARG;
  SHK_HIP(ctx, hipSetDevice(ctx->prm.device));
  SHK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  double total = 0.0, pre = 0.0;
  for (size_t i = 0; i < ctx->ev_used; ++i) {
    float ms = 0.f;
    SHK_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev_start[i], ctx->ev_stop[i]));
    total += ms;
    SHK_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev_pre[i], ctx->ev_start[i]));
    pre += ms;
  }
  *t = ctx->last;
  t->n_launches = ctx->ev_used;
  t->total_ms = total;
  t->prepass_ms = pre;
  return SHK_OK;
}

int shk_count_work(shk_ctx *ctx, const shk_batch *b, shk_work_counters *out)
{
  if (!ctx || !out) return SHK_ERR_ARG;
  if (ctx->mode != 2) return SHK_ERR_STATE;
  int rc = check_batch(ctx, b);
  if (rc) return rc;
  SHK_HIP(ctx, hipSetDevice(ctx->prm.device));
  shk_result tmp{};
  return classify_resident(ctx, b, 0, &tmp, out);
}

void *shk_alloc_pinned(size_t bytes)
{
  void *p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
  return p;
}

void shk_free_pinned(void *p)
{
  if (p) (void)hipHostFree(p);
}

}  // extern "C"
