// shark_hip.hip -- the C ABI of libsharkhip (include/shark_hip.h): context,
// device memory, and the orchestration around the HIP kernels.  No CPU
// implementation of any hot-path step lives here: without a device every
// computing entry point fails.
#include <hip/hip_runtime.h>

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
  hipFree(ix.bf64); hipFree(ix.rank_w); hipFree(ix.ent); hipFree(ix.ids); hipFree(ix.sum32); hipFree(ix.tab); hipFree(ix.lsum32);
  ix = DeviceIndex{};
}

// slot positions of a read (pair) whose mates are `len` long: classify.hip packs mate 2 at the next
// multiple of 8 after mate 1 and uses packed positions as k-mer slots
static uint32_t slots_for_len(uint32_t len, uint32_t k, bool paired)
{
  const uint32_t per = len >= k ? len - k + 1 : 0;
  return (paired && per) ? ((len + 7u) & ~7u) + per : per;
}

// everything after the inputs are resident in HBM; batch pointers are device pointers
static int classify_core(Ctx *ctx, const shk_batch *b, uint32_t max_read_len, shk_result *res, shk_work_counters *wc = nullptr)
{
  hipStream_t st = ctx->stream;
  const uint64_t n = b->n;
  if (n >= 0xFFFFFFFFull) { ctx->last_error = "batch too large (n must be < 2^32-1)"; return SHK_ERR_ARG; }
  int rc;
  if ((rc = ensure_capacity(ctx, &ctx->d_count, &ctx->cap_count, n + 1))) return rc;
  if ((rc = ensure_capacity(ctx, &ctx->d_inl, &ctx->cap_inl, n * SHK_INLINE_IDS + 8))) return rc;
  if ((rc = ensure_capacity(ctx, &ctx->d_gene_off, &ctx->cap_gene_off, n + 1))) return rc;
  if ((rc = ensure_capacity(ctx, &ctx->d_long_queue, &ctx->cap_long_queue, n + 1))) return rc;
  if ((rc = ensure_capacity(ctx, &ctx->d_tie_queue, &ctx->cap_tie_queue, 3 * n + 3))) return rc;
  if ((rc = ensure_capacity(ctx, &ctx->d_scan_temp, &ctx->cap_scan_temp, scan_temp_words(n + 1)))) return rc;

  SHK_HIP(ctx, hipMemsetAsync(ctx->d_counters, 0, CTR_WORDS * sizeof(uint32_t), st));
  SHK_HIP(ctx, hipMemsetAsync(ctx->d_count + n, 0, sizeof(uint32_t), st));

  ClassifyParams p{};
  const DeviceIndex &ix = ctx->idx;
  p.bf64 = ix.bf64; p.rank_w = ix.rank_w; p.ent = ix.ent; p.ids = ix.ids;
  p.sum32 = ix.sum_shift ? ix.sum32 : nullptr; p.sum_shift = ix.sum_shift;
  p.tab = ix.tab_lg ? ix.tab : nullptr; p.tab_lg = ix.tab_lg;
  p.tab_nt = ix.tab_lg && (16ull << ix.tab_lg) > (256ull << 20);   // beyond L2 + Infinity Cache
  p.lsum32 = ix.lsum_shift ? ix.lsum32 : nullptr; p.lsum_shift = ix.lsum_shift;
  p.bf_bits = ix.bf_bits;
  p.bf_mask = ix.pow2 ? ix.bf_bits - 1 : ~0ull;   // (non power-of-two: positions are reduced explicitly, the masks become no-ops)
  if (!ix.pow2) {
    const uint32_t s = (uint32_t)__builtin_ctzll(ix.bf_bits);
    const uint64_t m = ix.bf_bits >> s;
    p.mod_shift = s;
    p.mod_fast = s >= 32 && m < (1ull << 32);
    p.mod_m = p.mod_fast ? (uint32_t)m : 0;
    p.mod_c = p.mod_fast ? 0xFFFFFFFFFFFFFFFFull / m + 1 : 0;
  }
  p.k = ctx->prm.k; p.c = ctx->prm.c; p.single = ctx->prm.single;
  p.mq = ctx->prm.min_quality ? ctx->prm.min_quality + 33 : 0;  // FastqSplitter.hpp:70
  p.n = n;
  p.seq1 = (const uint8_t *)b->seq1; p.off1 = b->off1;
  p.seq2 = (const uint8_t *)b->seq2; p.off2 = b->off2;
  p.qual1 = (const uint8_t *)b->qual1; p.qual2 = (const uint8_t *)b->qual2;
  ClassifyOut ho;
  ho.count = ctx->d_count; ho.inl = ctx->d_inl;
  ho.counters = ctx->d_counters; ho.long_queue = ctx->d_long_queue; ho.tie_queue = ctx->d_tie_queue;
  SHK_HIP(ctx, hipMemcpyAsync(ctx->d_out, &ho, sizeof(ho), hipMemcpyHostToDevice, st));   // (pageable source: staged before return)
  p.out = ctx->d_out;
  p.gene_counts = wc ? nullptr : ctx->d_gene_counts;
  p.work_counters = nullptr;
  p.ablate = getenv("SHK_ABLATE") ? (uint32_t)atoi(getenv("SHK_ABLATE")) : 0u;

  const bool paired = b->seq2 != nullptr;
  uint32_t max_slots = max_read_len ? slots_for_len(max_read_len, p.k, paired) : 0;
  if (max_slots > fast_kernel_max_slots()) max_slots = fast_kernel_max_slots();
  const uint32_t fast_cap = 64 * fast_kernel_unroll(max_slots);

  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (ctx->timing) {
    if (ctx->ev_used == ctx->ev_start.size()) {
      hipEvent_t a, c;
      SHK_HIP(ctx, hipEventCreate(&a));
      SHK_HIP(ctx, hipEventCreate(&c));
      ctx->ev_start.push_back(a);
      ctx->ev_stop.push_back(c);
    }
    e0 = ctx->ev_start[ctx->ev_used];
    e1 = ctx->ev_stop[ctx->ev_used];
    ctx->ev_used++;
    SHK_HIP(ctx, hipEventRecord(e0, st));
  }
  if ((rc = launch_classify_fast(ctx, p, max_slots, st))) return rc;
  if (ctx->timing) SHK_HIP(ctx, hipEventRecord(e1, st));

  SHK_HIP(ctx, hipMemcpyAsync(ctx->h_counters, ctx->d_counters, CTR_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  SHK_HIP(ctx, hipStreamSynchronize(st));
  const uint32_t n_long = ctx->h_counters[CTR_LONG];
  uint32_t gen_slots = fast_cap;

  auto size_scratch = [&](uint64_t n_items, unsigned *n_waves) -> int {
    const uint32_t S = ((gen_slots + 63) / 64) * 64;
    const uint64_t stride = (uint64_t)stage_words_for(S) + (3ull * S) / 2 + 2;   // staging area + 3 u32 slot records
    uint64_t waves = std::min<uint64_t>(n_items, 4096);
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
  };

  if (n_long) {
    gen_slots = std::max(gen_slots, ctx->h_counters[CTR_MAX_SLOTS]);
    unsigned n_waves = 0;
    if ((rc = size_scratch(n_long, &n_waves))) return rc;
    p.work = ctx->d_long_queue;
    p.n_work = n_long;
    if ((rc = launch_classify_general(ctx, p, false, n_waves, st))) return rc;
    SHK_HIP(ctx, hipMemcpyAsync(ctx->h_counters, ctx->d_counters, CTR_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    SHK_HIP(ctx, hipStreamSynchronize(st));
  }
  const uint32_t n_tie = ctx->h_counters[CTR_TIE];

  // associations -> CSR (gene_off, gene_ids); the grand total of the scan is the number of associations
  const uint64_t *d_total = exclusive_scan_u32(ctx->d_count, ctx->d_gene_off, n + 1, ctx->d_scan_temp, st);
  SHK_HIP(ctx, hipGetLastError());
  uint64_t *h_total = reinterpret_cast<uint64_t *>(ctx->h_counters + CTR_WORDS);
  SHK_HIP(ctx, hipMemcpyAsync(h_total, d_total, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
  SHK_HIP(ctx, hipStreamSynchronize(st));
  const uint64_t n_assoc = *h_total;
  if (n_assoc >= 0xFFFFFFFFull) { ctx->last_error = "more than 2^32-1 associations in one batch"; return SHK_ERR_ARG; }
  if ((rc = ensure_capacity(ctx, &ctx->d_gene_ids, &ctx->cap_gene_ids, n_assoc + 8))) return rc;
  if ((rc = launch_gather_inline(ctx->d_count, ctx->d_inl, ctx->d_gene_off, ctx->d_gene_ids, n, wc ? nullptr : ctx->d_gene_counts, st))) return rc;
  if (n_tie) {
    unsigned n_waves = 0;
    if ((rc = size_scratch(n_tie, &n_waves))) return rc;
    p.work = ctx->d_tie_queue;
    p.n_work = n_tie;
    p.gene_off = ctx->d_gene_off;
    p.gene_ids = ctx->d_gene_ids;
    if ((rc = launch_classify_general(ctx, p, true, n_waves, st))) return rc;
  }
  SHK_HIP(ctx, hipStreamSynchronize(st));

  if (wc) {
    // measurement only: re-run every read through the general kernel with the
    // exact work counters switched on (results are rewritten with equal values)
    unsigned n_waves = 0;
    if ((rc = size_scratch(n, &n_waves))) return rc;
    SHK_HIP(ctx, hipMemsetAsync(ctx->d_counters, 0, CTR_WORDS * sizeof(uint32_t), st));
    SHK_HIP(ctx, hipMemsetAsync(ctx->d_work_counters, 0, 4 * sizeof(unsigned long long), st));
    p.work = nullptr;
    p.n_work = n;
    p.work_counters = ctx->d_work_counters;
    if ((rc = launch_classify_general(ctx, p, false, n_waves, st))) return rc;
    unsigned long long h[4] = {0, 0, 0, 0};
    SHK_HIP(ctx, hipMemcpyAsync(h, ctx->d_work_counters, sizeof(h), hipMemcpyDeviceToHost, st));
    SHK_HIP(ctx, hipStreamSynchronize(st));
    wc->n_kmers = h[0]; wc->n_hits = h[1]; wc->n_list_ids = h[2]; wc->n_bases = h[3];
  }

  ctx->last.last_n_reads = n;
  ctx->last.last_n_long = n_long;
  ctx->last.last_n_tie = n_tie;
  ctx->last.last_n_assoc = n_assoc;
  res->n = n;
  res->gene_off = ctx->d_gene_off;
  res->gene_ids = ctx->d_gene_ids;
  res->n_assoc = n_assoc;
  return SHK_OK;
}

}  // namespace shk

using namespace shk;

struct shk_ctx : public shk::Ctx {};

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
  if (prm->min_quality < 0 || prm->min_quality > 94) return SHK_ERR_ARG; // :138; Q+33 must fit a char
  if (prm->bf_bits == 0) return SHK_ERR_ARG;
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return SHK_ERR_NO_DEVICE;
  if (prm->device < 0 || prm->device >= n_dev) return SHK_ERR_ARG;
  shk_ctx *ctx = new (std::nothrow) shk_ctx();
  if (!ctx) return SHK_ERR_NOMEM;
  ctx->prm = *prm;
  ctx->prm.single = prm->single ? 1 : 0;
  auto fail = [&](int rc) { shk_destroy(ctx); return rc; };
#define CR_HIP(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) return fail(e__ == hipErrorOutOfMemory ? SHK_ERR_NOMEM : SHK_ERR_HIP); } while (0)
  CR_HIP(hipSetDevice(prm->device));
  CR_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
  DeviceIndex &ix = ctx->idx;
  ix.bf_bits = prm->bf_bits;
  ix.pow2 = (prm->bf_bits & (prm->bf_bits - 1)) == 0;
  ix.bf_words64 = ((prm->bf_bits + 511) / 512) * 8;
  // BF::BF(size): size zero bits (bloomfilter.h:48-53)
  CR_HIP(hipMalloc((void **)&ix.bf64, ix.bf_words64 * sizeof(uint64_t)));
  CR_HIP(hipMemsetAsync(ix.bf64, 0, ix.bf_words64 * sizeof(uint64_t), ctx->stream));
  CR_HIP(hipMalloc((void **)&ctx->d_counters, CTR_WORDS * sizeof(uint32_t)));
  CR_HIP(hipMalloc((void **)&ctx->d_out, sizeof(ClassifyOut)));
  CR_HIP(hipMalloc((void **)&ctx->d_gene_counts, 65536 * sizeof(unsigned long long)));
  CR_HIP(hipMemsetAsync(ctx->d_gene_counts, 0, 65536 * sizeof(unsigned long long), ctx->stream));
  CR_HIP(hipMalloc((void **)&ctx->d_work_counters, 4 * sizeof(unsigned long long)));
  CR_HIP(hipHostMalloc((void **)&ctx->h_counters, (CTR_WORDS + 2) * sizeof(uint32_t), hipHostMallocDefault));
  CR_HIP(hipStreamSynchronize(ctx->stream));
#undef CR_HIP
  *out = ctx;
  return SHK_OK;
}

void shk_destroy(shk_ctx *ctx)
{
  if (!ctx) return;
  (void)hipSetDevice(ctx->prm.device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  free_index(ctx->idx);
  hipFree(ctx->d_seq1); hipFree(ctx->d_seq2); hipFree(ctx->d_qual1); hipFree(ctx->d_qual2);
  hipFree(ctx->d_off1); hipFree(ctx->d_off2);
  hipFree(ctx->d_count); hipFree(ctx->d_inl); hipFree(ctx->d_gene_off); hipFree(ctx->d_gene_ids);
  hipFree(ctx->d_long_queue); hipFree(ctx->d_tie_queue); hipFree(ctx->d_counters); hipFree(ctx->d_out);
  hipFree(ctx->d_scan_temp); hipFree(ctx->d_scratch); hipFree(ctx->d_gene_counts); hipFree(ctx->d_work_counters);
  if (ctx->h_counters) (void)hipHostFree(ctx->h_counters);
  for (auto e : ctx->ev_start) (void)hipEventDestroy(e);
  for (auto e : ctx->ev_stop) (void)hipEventDestroy(e);
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
  if (ctx->prm.min_quality != 0 && (!b->qual1 || (b->seq2 && !b->qual2))) return SHK_ERR_ARG;
  return SHK_OK;
}

int shk_classify_device(shk_ctx *ctx, const shk_batch *batch, uint32_t max_read_len, shk_result *result)
{
  if (!ctx || !result) return SHK_ERR_ARG;
  if (ctx->mode != 2) return SHK_ERR_STATE;
  int rc = check_batch(ctx, batch);
  if (rc) return rc;
  SHK_HIP(ctx, hipSetDevice(ctx->prm.device));
  return classify_core(ctx, batch, max_read_len, result);
}

int shk_classify(shk_ctx *ctx, const shk_batch *b, shk_result *result)
{
  if (!ctx || !result) return SHK_ERR_ARG;
  if (ctx->mode != 2) return SHK_ERR_STATE;
  int rc = check_batch(ctx, b);
  if (rc) return rc;
  SHK_HIP(ctx, hipSetDevice(ctx->prm.device));
  hipStream_t st = ctx->stream;
  const uint64_t n = b->n;
  const uint64_t bytes1 = n ? b->off1[n] : 0;
  const uint64_t bytes2 = (n && b->seq2) ? b->off2[n] : 0;
  // longest mate: selects the kernel specialisation only
  uint32_t max_len = 1;
  for (uint64_t i = 0; i < n; ++i) {
    if (b->off1[i + 1] < b->off1[i]) return SHK_ERR_ARG;
    max_len = std::max<uint64_t>(max_len, std::min<uint64_t>(b->off1[i + 1] - b->off1[i], 0xFFFFFFFFull));
    if (b->seq2) {
      if (b->off2[i + 1] < b->off2[i]) return SHK_ERR_ARG;
      max_len = std::max<uint64_t>(max_len, std::min<uint64_t>(b->off2[i + 1] - b->off2[i], 0xFFFFFFFFull));
    }
  }
  shk_batch d{};
  d.n = n;
  if ((rc = ensure_capacity(ctx, &ctx->d_seq1, &ctx->cap_seq1, bytes1 + 16))) return rc;
  if ((rc = ensure_capacity(ctx, &ctx->d_off1, &ctx->cap_off1, n + 1))) return rc;
  if (n) {
    SHK_HIP(ctx, hipMemcpyAsync(ctx->d_seq1, b->seq1, bytes1, hipMemcpyHostToDevice, st));
    SHK_HIP(ctx, hipMemcpyAsync(ctx->d_off1, b->off1, (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, st));
  }
  d.seq1 = (const char *)ctx->d_seq1;
  d.off1 = ctx->d_off1;
  if (b->seq2 && n) {
    if ((rc = ensure_capacity(ctx, &ctx->d_seq2, &ctx->cap_seq2, bytes2 + 16))) return rc;
    if ((rc = ensure_capacity(ctx, &ctx->d_off2, &ctx->cap_off2, n + 1))) return rc;
    SHK_HIP(ctx, hipMemcpyAsync(ctx->d_seq2, b->seq2, bytes2, hipMemcpyHostToDevice, st));
    SHK_HIP(ctx, hipMemcpyAsync(ctx->d_off2, b->off2, (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    d.seq2 = (const char *)ctx->d_seq2;
    d.off2 = ctx->d_off2;
  }
  if (ctx->prm.min_quality != 0 && n) {
    if ((rc = ensure_capacity(ctx, &ctx->d_qual1, &ctx->cap_qual1, bytes1 + 16))) return rc;
    SHK_HIP(ctx, hipMemcpyAsync(ctx->d_qual1, b->qual1, bytes1, hipMemcpyHostToDevice, st));
    d.qual1 = (const char *)ctx->d_qual1;
    if (b->seq2) {
      if ((rc = ensure_capacity(ctx, &ctx->d_qual2, &ctx->cap_qual2, bytes2 + 16))) return rc;
      SHK_HIP(ctx, hipMemcpyAsync(ctx->d_qual2, b->qual2, bytes2, hipMemcpyHostToDevice, st));
      d.qual2 = (const char *)ctx->d_qual2;
    }
  }
  shk_result dr{};
  if ((rc = classify_core(ctx, &d, max_len, &dr))) return rc;
  try {
    ctx->h_gene_off.resize(n + 1);
    ctx->h_gene_ids.resize(dr.n_assoc + 1);
  } catch (...) {
    return SHK_ERR_NOMEM;
  }
  SHK_HIP(ctx, hipMemcpyAsync(ctx->h_gene_off.data(), dr.gene_off, (n + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  if (dr.n_assoc)
    SHK_HIP(ctx, hipMemcpyAsync(ctx->h_gene_ids.data(), dr.gene_ids, dr.n_assoc * sizeof(uint16_t), hipMemcpyDeviceToHost, st));
  SHK_HIP(ctx, hipStreamSynchronize(st));
  result->n = n;
  result->gene_off = ctx->h_gene_off.data();
  result->gene_ids = ctx->h_gene_ids.data();
  result->n_assoc = dr.n_assoc;
  return SHK_OK;
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

// ---- RCCL, loaded on first use so that single-GPU users never pay for (or conflict over) it ----
namespace {
typedef void *rccl_comm_t;
struct RcclApi {
  void *lib = nullptr;
  int (*CommInitAll)(rccl_comm_t *, int, const int *) = nullptr;
  int (*CommDestroy)(rccl_comm_t) = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, rccl_comm_t, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  bool ok = false;
};
RcclApi &rccl()
{
  static RcclApi api;
  if (!api.lib) {
    api.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!api.lib) api.lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (api.lib) {
      api.CommInitAll = (int (*)(rccl_comm_t *, int, const int *))dlsym(api.lib, "ncclCommInitAll");
      api.CommDestroy = (int (*)(rccl_comm_t))dlsym(api.lib, "ncclCommDestroy");
      api.AllReduce = (int (*)(const void *, void *, size_t, int, int, rccl_comm_t, hipStream_t))dlsym(api.lib, "ncclAllReduce");
      api.GroupStart = (int (*)())dlsym(api.lib, "ncclGroupStart");
      api.GroupEnd = (int (*)())dlsym(api.lib, "ncclGroupEnd");
      api.ok = api.CommInitAll && api.CommDestroy && api.AllReduce && api.GroupStart && api.GroupEnd;
    }
  }
  return api;
}
}  // namespace

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
  const char *force = getenv("SHK_FORCE_RCCL");
  if (n_ctx > 1 || (force && force[0] == '1')) {
    RcclApi &api = rccl();
    if (!api.ok) { c0->last_error = "librccl.so.1 could not be loaded"; return SHK_ERR_HIP; }
    std::vector<int> devs((size_t)n_ctx);
    for (int i = 0; i < n_ctx; ++i) devs[(size_t)i] = ctxs[i]->prm.device;
    std::vector<rccl_comm_t> comms((size_t)n_ctx, nullptr);
    // RCCL may print a version banner on stdout when it initialises; stdout is the ssv stream of
    // the CLI (ReadOutput.hpp:43), so anything RCCL prints during init is sent to stderr instead
    fflush(stdout);
    const int saved_stdout = dup(1);
    if (saved_stdout >= 0) (void)dup2(2, 1);
    const int init_rc = api.CommInitAll(comms.data(), n_ctx, devs.data());
    if (saved_stdout >= 0) {
      fflush(stdout);
      (void)dup2(saved_stdout, 1);
      close(saved_stdout);
    }
    if (init_rc != 0) { c0->last_error = "ncclCommInitAll failed"; return SHK_ERR_HIP; }
    int rc = api.GroupStart();
    for (int i = 0; i < n_ctx && rc == 0; ++i) {
      (void)hipSetDevice(devs[(size_t)i]);
      // ncclUint64 = 5, ncclSum = 0; in place on every GPU's 65 536 x 8 B counter block
      rc = api.AllReduce(ctxs[i]->d_gene_counts, ctxs[i]->d_gene_counts, 65536, 5, 0, comms[(size_t)i], ctxs[i]->stream);
    }
    if (rc == 0) rc = api.GroupEnd();
    for (int i = 0; i < n_ctx; ++i) {
      (void)hipSetDevice(devs[(size_t)i]);
      (void)hipStreamSynchronize(ctxs[i]->stream);
      (void)api.CommDestroy(comms[(size_t)i]);
    }
    if (rc != 0) { c0->last_error = "ncclAllReduce failed"; return SHK_ERR_HIP; }
  }
  if (totals) {
    SHK_HIP(c0, hipSetDevice(c0->prm.device));
    SHK_HIP(c0, hipMemcpy(totals, c0->d_gene_counts, (size_t)n * sizeof(uint64_t), hipMemcpyDeviceToHost));
  }
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
  double total = 0.0;
  for (size_t i = 0; i < ctx->ev_used; ++i) {
    float ms = 0.f;
    SHK_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev_start[i], ctx->ev_stop[i]));
    total += ms;
  }
  *t = ctx->last;
  t->n_launches = ctx->ev_used;
  t->total_ms = total;
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
  return classify_core(ctx, b, 0, &tmp, out);
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
