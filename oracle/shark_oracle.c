/*
 * shark_oracle.c -- CPU ORACLE (test infrastructure, see shark_oracle.h).
 *
 * Literal plain-C restatement of the reference hot path.  Every function
 * names the reference file:line it follows.  Written from the reference's
 * behaviour; no reference source text is reproduced.
 */
#define _GNU_SOURCE
#include "shark_oracle.h"

#include <pthread.h>
#include <stdio.h>
#include <time.h>
#include <stdlib.h>
#include <string.h>

/* ======================================================================== */
/* kmer_utils.hpp                                                           */
/* ======================================================================== */

/* kmer_utils.hpp:29-41 -- 128-entry table: A/a=1 C/c=2 G/g=3 T/t=4 else 0.
 * The reference indexes the table with a (signed) char; bytes >= 128 are
 * undefined behaviour there.  The oracle (and the product) define them as 0
 * (= invalid), which is the only interpretation that never reads outside the
 * table. */
uint8_t so_to_int(char c)
{
  switch (c) {
  case 'A': case 'a': return 1;
  case 'C': case 'c': return 2;
  case 'G': case 'g': return 3;
  case 'T': case 't': return 4;
  default: return 0;
  }
}

/* kmer_utils.hpp:43-45 */
uint8_t so_reverse_char(uint8_t c) { return (uint8_t)((~c) & 3); }

/* kmer_utils.hpp:47-55 */
uint64_t so_revcompl(uint64_t kmer, uint8_t k)
{
  uint64_t rckmer = 0;
  kmer = ~kmer;
  for (unsigned i = 0; i < k; ++i) {
    rckmer = (rckmer << 2) | (kmer & 3);
    kmer >>= 2;
  }
  return rckmer;
}

/* kmer_utils.hpp:57-71.  NOTE the scan bound `_p < p + k` is re-evaluated
 * with the UPDATED p (:58-60), so the scan keeps sliding until k consecutive
 * valid characters have been seen or the string ends. */
int64_t so_build_kmer(const char *seq, int n, int *p, uint8_t k)
{
  for (int _p = *p; _p < n && _p < *p + k; ++_p) {
    if (so_to_int(seq[_p]) == 0) *p = _p + 1;
  }
  if (*p + k > n) {
    *p = n;
    return -1;
  }
  uint64_t kmer = 0;
  for (int end = *p + k; *p < end; ++(*p)) {
    kmer = (kmer << 2) | (uint64_t)(so_to_int(seq[*p]) - 1);
  }
  return (int64_t)kmer;
}

/* kmer_utils.hpp:73-75 */
uint64_t so_lsappend(uint64_t kmer, uint64_t c, uint64_t k)
{
  return ((kmer << 2) | c) & ((1UL << 2 * k) - 1);
}

/* kmer_utils.hpp:77-79 */
uint64_t so_rsprepend(uint64_t kmer, uint64_t c, uint64_t k)
{
  return (kmer >> 2) | (c << (2 * k - 2));
}

/* ---- xxhash.hpp (RedSpah xxhash_cpp 0.6.5 = XXH64 by Y. Collet) --------- */
/* primes: xxhash.hpp:349 */
#define P64_1 11400714785074694791ULL
#define P64_2 14029467366897019727ULL
#define P64_3 1609587929392839161ULL
#define P64_4 9650029242287828579ULL
#define P64_5 2870177450012600261ULL

static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }

static inline uint64_t read_le64(const uint8_t *p)
{
  uint64_t v = 0;
  for (int i = 7; i >= 0; --i) v = (v << 8) | p[i];
  return v;
}
static inline uint32_t read_le32(const uint8_t *p)
{
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

/* xxhash.hpp:366-373 */
static inline uint64_t xxh_round(uint64_t seed, uint64_t input)
{
  seed += input * P64_2;
  seed = rotl64(seed, 31);
  seed *= P64_1;
  return seed;
}
/* xxhash.hpp:375-381 */
static inline uint64_t xxh_merge_round(uint64_t acc, uint64_t val)
{
  val = xxh_round(0, val);
  acc ^= val;
  acc = acc * P64_1 + P64_4;
  return acc;
}

/* xxhash.hpp:458-492 (endian_align) + :424-456 (ending) */
uint64_t so_xxh64(const void *data, size_t len, uint64_t seed)
{
  const uint8_t *p = (const uint8_t *)data;
  const uint8_t *end = p + len;
  uint64_t h;
  if (len >= 32) {
    const uint8_t *limit = end - 32;
    uint64_t v1 = seed + P64_1 + P64_2, v2 = seed + P64_2, v3 = seed, v4 = seed - P64_1;
    do {
      v1 = xxh_round(v1, read_le64(p)); p += 8;
      v2 = xxh_round(v2, read_le64(p)); p += 8;
      v3 = xxh_round(v3, read_le64(p)); p += 8;
      v4 = xxh_round(v4, read_le64(p)); p += 8;
    } while (p <= limit);
    h = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
    h = xxh_merge_round(h, v1);
    h = xxh_merge_round(h, v2);
    h = xxh_merge_round(h, v3);
    h = xxh_merge_round(h, v4);
  } else {
    h = seed + P64_5;
  }
  h += (uint64_t)len;
  while (p + 8 <= end) {
    uint64_t k1 = xxh_round(0, read_le64(p));
    h ^= k1;
    h = rotl64(h, 27) * P64_1 + P64_4;
    p += 8;
  }
  if (p + 4 <= end) {
    h ^= (uint64_t)read_le32(p) * P64_1;
    h = rotl64(h, 23) * P64_2 + P64_3;
    p += 4;
  }
  while (p < end) {
    h ^= (*p) * P64_5;
    h = rotl64(h, 11) * P64_1;
    p++;
  }
  h ^= h >> 33;
  h *= P64_2;
  h ^= h >> 29;
  h *= P64_3;
  h ^= h >> 32;
  return h;
}

/* kmer_utils.hpp:81-83: XXH64 of the k-mer's 8 in-memory (little-endian)
 * bytes, seed 0. */
uint64_t so_get_hash(uint64_t kmer)
{
  uint8_t b[8];
  for (int i = 0; i < 8; ++i) b[i] = (uint8_t)(kmer >> (8 * i));
  return so_xxh64(b, 8, 0);
}

/* ======================================================================== */
/* small_vector.hpp:25-89 -- semantics only: an appendable uint16_t list.   */
/* (The reference's 8-byte inline/heap union is a memory optimisation.)     */
/* ======================================================================== */
/* (Like the reference's, a list of up to four entries lives in the eight bytes of the pointer: the 60 000-gene indices of the
 *  scale tests hold 170 M lists, nearly all of one entry -- a heap block each made their build spend two thirds of its time in
 *  malloc, free and the cache misses of walking them.)  cap == 0: inline. */
#define SV_INLINE 4u
typedef struct {
  union { uint16_t *d; uint16_t in[SV_INLINE]; } u;
  uint32_t n, cap;
} smallvec;

static inline const uint16_t *sv_data(const smallvec *v) { return v->cap ? v->u.d : v->u.in; }
static inline uint16_t sv_back(const smallvec *v) { return sv_data(v)[v->n - 1]; }
static inline void sv_free(smallvec *v) { if (v->cap) free(v->u.d); }

static void sv_push(smallvec *v, uint16_t x) /* small_vector.hpp:44-56 */
{
  if (v->cap == 0) {
    if (v->n < SV_INLINE) { v->u.in[v->n++] = x; return; }
    uint16_t *d = (uint16_t *)malloc(2u * SV_INLINE * sizeof(uint16_t));
    memcpy(d, v->u.in, SV_INLINE * sizeof(uint16_t));
    v->u.d = d;
    v->cap = 2u * SV_INLINE;
  } else if (v->n == v->cap) {
    v->cap *= 2;
    v->u.d = (uint16_t *)realloc(v->u.d, v->cap * sizeof(uint16_t));
  }
  v->u.d[v->n++] = x;
}

/* ======================================================================== */
/* bloomfilter.h : class BF                                                 */
/* ======================================================================== */
struct so_bf {
  uint64_t size;      /* _size */
  int mode;           /* _mode */
  uint64_t *bf;       /* _bf   (sdsl::bit_vector: LSB-first 64-bit words) */
  uint64_t nwords;
  uint64_t *brank;    /* _brank: ones before each 512-bit block */
  uint64_t num_kmer;
  smallvec *set_index; /* _set_index */
  uint64_t *bv;       /* _bv */
  uint64_t tot_idx;
  uint32_t *select_bv; /* _select_bv: select_bv[i] = position of the i-th one, 1-based */
  uint16_t *index_kmer; /* _index_kmer */
};

/* bloomfilter.h:48-53 */
so_bf *so_bf_new(uint64_t size_bits)
{
  so_bf *b = (so_bf *)calloc(1, sizeof(so_bf));
  b->size = size_bits;
  b->mode = 0;
  b->nwords = (size_bits + 63) / 64;
  /* round the allocation up to a whole 512-bit block for the rank scan */
  uint64_t alloc_words = ((b->nwords + 7) / 8) * 8 + 8;
  b->bf = (uint64_t *)calloc(alloc_words, sizeof(uint64_t));
  if (!b->bf) { free(b); return NULL; }
  return b;
}

void so_bf_free(so_bf *b)
{
  if (!b) return;
  if (b->set_index) {
    for (uint64_t i = 0; i < b->num_kmer; ++i) sv_free(&b->set_index[i]);
    free(b->set_index);
  }
  free(b->bf); free(b->brank); free(b->bv); free(b->select_bv); free(b->index_kmer);
  free(b);
}

/* bloomfilter.h:57-59 */
void so_bf_add_at(so_bf *b, uint64_t p)
{
  uint64_t i = p % b->size;
  b->bf[i >> 6] |= (uint64_t)1 << (i & 63);
}

/* sdsl rank_support_v semantics: rank(i) = number of ones in [0, i) */
uint64_t so_bf_rank(const so_bf *b, uint64_t i)
{
  uint64_t blk = i >> 9;
  uint64_t r = b->brank[blk];
  uint64_t w0 = blk << 3, w = i >> 6;
  for (uint64_t j = w0; j < w; ++j) r += (uint64_t)__builtin_popcountll(b->bf[j]);
  if (i & 63) r += (uint64_t)__builtin_popcountll(b->bf[w] & (((uint64_t)1 << (i & 63)) - 1));
  return r;
}

static int cmp_u64(const void *a, const void *b)
{
  uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
  return x < y ? -1 : (x > y ? 1 : 0);
}

/* bloomfilter.h:61-75 */
void so_bf_add_to_kmer(so_bf *b, uint64_t *kmers, size_t n, int input_idx)
{
  if (b->mode != 1) return;
  for (size_t i = 0; i < n; ++i) kmers[i] = so_get_hash(kmers[i]) % b->size;
  qsort(kmers, n, sizeof(uint64_t), cmp_u64);
  for (size_t i = 0; i < n; ++i) {
    int kmer_rank = (int)so_bf_rank(b, kmers[i]);
    smallvec *sv = &b->set_index[kmer_rank];
    /* :72 compares uint16_t last() with the int input_idx */
    if (sv->n == 0 || (int)sv_back(sv) != input_idx) sv_push(sv, (uint16_t)input_idx);
  }
}

/* bloomfilter.h:78-102 (NDEBUG build: the mode check at :82-85 is compiled out) */
void so_bf_get_index(const so_bf *b, uint64_t kmer, int *start_pos, int *end_pos)
{
  *start_pos = 0;
  *end_pos = -1;
  uint64_t hash = so_get_hash(kmer);
  uint64_t bf_idx = hash % b->size;
  if ((b->bf[bf_idx >> 6] >> (bf_idx & 63)) & 1) {
    uint64_t rank_searched = so_bf_rank(b, bf_idx + 1);
    if (rank_searched > 1) *start_pos = (int)b->select_bv[rank_searched - 1] + 1;
    *end_pos = (int)b->select_bv[rank_searched];
  }
}

/* bloomfilter.h:111-188 */
int so_bf_switch_mode(so_bf *b, int new_mode)
{
  if (b->mode == 0 && new_mode == 1) {
    b->mode = new_mode;
    /* :121 init_support(_brank,&_bf) */
    uint64_t nblk = (b->nwords + 7) / 8 + 1;
    b->brank = (uint64_t *)malloc((nblk + 1) * sizeof(uint64_t));
    uint64_t acc = 0;
    for (uint64_t blk = 0; blk < nblk; ++blk) {
      b->brank[blk] = acc;
      for (int j = 0; j < 8; ++j) acc += (uint64_t)__builtin_popcountll(b->bf[blk * 8 + j]);
    }
    b->brank[nblk] = acc;
    /* :122 */
    b->num_kmer = so_bf_rank(b, b->size);
    /* :123-124 */
    if (b->num_kmer != 0) b->set_index = (smallvec *)calloc(b->num_kmer, sizeof(smallvec));
    return 1;
  } else if (b->mode == 1 && new_mode == 2) {
    b->mode = new_mode;
    /* :130-133 (int in the reference: valid while < 2^31) */
    uint64_t tot_idx = 0;
    for (uint64_t i = 0; i < b->num_kmer; ++i) tot_idx += b->set_index[i].n;
    b->tot_idx = tot_idx;
    /* :142-147 the bit vector with a one at the last entry of every list; :148 its select support (position of the i-th one,
     * i = 1..popcount); :156-167 the lists one behind the other -- in ONE walk over the lists (three walks of 170 M sixteen-byte
     * entries were a tenth of the scale tests' index build) */
    b->bv = (uint64_t *)calloc(tot_idx / 64 + 2, sizeof(uint64_t));
    b->select_bv = (uint32_t *)malloc((b->num_kmer + 2) * sizeof(uint32_t));
    b->index_kmer = (uint16_t *)malloc((tot_idx + 1) * sizeof(uint16_t));
    int64_t pos = -1;
    uint64_t ones = 0, ins = 0;
    for (uint64_t i = 0; i < b->num_kmer; ++i) {
      const smallvec *sv = &b->set_index[i];
      pos += sv->n;
      if (pos >= 0) {                                        /* (a list is never empty: every set bit was hit by its own k-mer) */
        if (!((b->bv[pos >> 6] >> (pos & 63)) & 1)) b->select_bv[++ones] = (uint32_t)pos;
        b->bv[pos >> 6] |= (uint64_t)1 << (pos & 63);
      }
      memcpy(b->index_kmer + ins, sv_data(sv), sv->n * sizeof(uint16_t));
      ins += sv->n;
    }
    /* :183 */
    for (uint64_t i = 0; i < b->num_kmer; ++i) sv_free(&b->set_index[i]);
    free(b->set_index);
    b->set_index = NULL;
    return 1;
  }
  return 0;
}

uint64_t so_bf_size(const so_bf *b) { return b->size; }
const uint64_t *so_bf_words(const so_bf *b) { return b->bf; }
uint64_t so_bf_num_kmer(const so_bf *b) { return b->num_kmer; }
uint64_t so_bf_tot_idx(const so_bf *b) { return b->tot_idx; }
const uint16_t *so_bf_index_kmer(const so_bf *b) { return b->index_kmer; }

/* ======================================================================== */
/* KmerBuilder.hpp:40-72                                                    */
/* ======================================================================== */
size_t so_kmer_builder(const char *seq, size_t n, uint32_t k, uint64_t *out)
{
  size_t cnt = 0;
  if (n >= k) {                                             /* :44 */
    int _pos = 0;
    uint64_t kmer = (uint64_t)so_build_kmer(seq, (int)n, &_pos, (uint8_t)k); /* :46 */
    if (kmer == (uint64_t)-1) return 0;                     /* :47 */
    uint64_t rckmer = so_revcompl(kmer, (uint8_t)k);        /* :48 */
    uint64_t key = kmer < rckmer ? kmer : rckmer;           /* :49 */
    out[cnt++] = so_get_hash(key);                          /* :50 */
    for (int pos = _pos; pos < (int)n; ++pos) {             /* :52 */
      uint8_t new_char = so_to_int(seq[pos]);
      if (new_char == 0) {                                  /* :54-59 */
        ++pos;
        kmer = (uint64_t)so_build_kmer(seq, (int)n, &pos, (uint8_t)k);
        if (kmer == (uint64_t)-1) break;
        rckmer = so_revcompl(kmer, (uint8_t)k);
        --pos;
      } else {                                              /* :60-64 */
        --new_char;
        kmer = so_lsappend(kmer, new_char, k);
        rckmer = so_rsprepend(rckmer, so_reverse_char(new_char), k);
      }
      key = kmer < rckmer ? kmer : rckmer;                  /* :65-66 */
      out[cnt++] = so_get_hash(key);
    }
  }
  return cnt;
}

/* ======================================================================== */
/* FastqSplitter.hpp:47-93 + mask_seq :104-113                              */
/* ======================================================================== */
size_t so_join_mask(const char *s1, size_t l1, const char *q1,
                    const char *s2, size_t l2, const char *q2,
                    int paired, char min_quality, char *out)
{
  size_t n = 0;
  memcpy(out, s1, l1); n = l1;
  if (paired) {                                             /* :63 / :83 */
    out[n++] = 'N';
    memcpy(out + n, s2, l2); n += l2;
  }
  if (min_quality != 0) {
    const char mq = (char)(min_quality + 33);               /* :70 */
    /* quality string: qual1 [+ "\33" + qual2] (:84; \33 is octal 27) */
    for (size_t i = 0; i < n; ++i) {
      char q;
      if (i < l1) q = q1[i];
      else if (i == l1) q = '\33';
      else q = q2[i - l1 - 1];
      if (q < mq) out[i] = (char)(out[i] - 64);             /* :106 */
    }
  }
  return n;
}

/* ======================================================================== */
/* main.cpp + ReadAnalyzer.hpp                                              */
/* ======================================================================== */
struct so_shark {
  uint32_t k;
  double c;
  int min_quality;
  int single;
  so_bf *bf;
  int nidx;
};

so_shark *so_shark_new(uint32_t k, double c, uint64_t bf_bits, int min_quality, int single)
{
  so_shark *s = (so_shark *)calloc(1, sizeof(so_shark));
  s->k = k; s->c = c; s->min_quality = min_quality; s->single = single;
  s->bf = so_bf_new(bf_bits);                               /* main.cpp:108 */
  if (!s->bf) { free(s); return NULL; }
  return s;
}

void so_shark_free(so_shark *s)
{
  if (!s) return;
  so_bf_free(s->bf);
  free(s);
}

const so_bf *so_shark_bf(const so_shark *s) { return s->bf; }

int so_shark_build(so_shark *s, const char *const *seqs, const uint64_t *lens, size_t n_records)
{
  const uint32_t k = s->k;
  /* ---- pass 1: main.cpp:128-144 (FastaSplitter -> KmerBuilder -> BloomfilterFiller) */
  for (size_t r = 0; r < n_records; ++r) {
    size_t n = (size_t)lens[r];
    if (n < k) continue;                                    /* KmerBuilder.hpp:44 */
    uint64_t *hashes = (uint64_t *)malloc((n - k + 1) * sizeof(uint64_t));
    size_t cnt = so_kmer_builder(seqs[r], n, k, hashes);
    for (size_t i = 0; i < cnt; ++i) so_bf_add_at(s->bf, hashes[i]); /* BloomfilterFiller.hpp:41-43 */
    free(hashes);
  }
  so_bf_switch_mode(s->bf, 1);                              /* main.cpp:148 */

  /* ---- pass 2: main.cpp:154-189 */
  int nidx = 0;
  for (size_t r = 0; r < n_records; ++r) {
    const char *seq = seqs[r];
    int seq_len = (int)lens[r];
    if ((unsigned)seq_len >= k) {                           /* :162 */
      uint64_t *kmers = (uint64_t *)malloc(((size_t)seq_len - k + 1) * sizeof(uint64_t));
      size_t cnt = 0;
      int _p = 0;
      uint64_t kmer = (uint64_t)so_build_kmer(seq, seq_len, &_p, (uint8_t)k); /* :164 */
      if (kmer == (uint64_t)-1) { free(kmers); continue; }  /* :165 -- skips ++nidx (quirk A) */
      uint64_t rckmer = so_revcompl(kmer, (uint8_t)k);
      kmers[cnt++] = kmer < rckmer ? kmer : rckmer;         /* :167 */
      for (int p = _p; p < seq_len; ++p) {                  /* :168 */
        uint8_t new_char = so_to_int(seq[p]);
        if (new_char == 0) {                                /* :170-175 */
          ++p;
          kmer = (uint64_t)so_build_kmer(seq, seq_len, &p, (uint8_t)k);
          if (kmer == (uint64_t)-1) break;
          rckmer = so_revcompl(kmer, (uint8_t)k);
          --p;
        } else {                                            /* :176-180 */
          --new_char;
          kmer = so_lsappend(kmer, new_char, k);
          rckmer = so_rsprepend(rckmer, so_reverse_char(new_char), k);
        }
        kmers[cnt++] = kmer < rckmer ? kmer : rckmer;       /* :181 */
      }
      so_bf_add_to_kmer(s->bf, kmers, cnt, nidx);           /* :183 */
      free(kmers);
    }
    ++nidx;                                                 /* :185 */
  }
  so_bf_switch_mode(s->bf, 2);                              /* :193 */
  s->nidx = nidx;
  return nidx;
}

/* ---- the same build on several threads (test infrastructure's own convenience: the 60 000-gene indices of the scale
 * tests take the serial build 80 s each).  Nothing about the RESULT changes, and the argument is the reference's own:
 *   pass 1  BloomfilterFiller.hpp:38-46 sets bits under a mutex in whatever order the worker threads deliver their
 *           batches; setting a bit is idempotent, so the filter does not depend on the order.  Here: records are dealt to
 *           threads, bits are set with an atomic OR.
 *   pass 2  main.cpp:154-189 walks the records in file order on one thread.  What a record does to the index is: number
 *           itself (nidx, with the skipped increment of :165), hash and sort its k-mers (bloomfilter.h:65-68), and append
 *           its number to the list of every set bit it hits unless that list already ends with it (:69-74).  The lists
 *           are independent of each other, and a list sees the records in file order.  Here: (A) threads compute every
 *           record's sorted positions and their ranks -- pure functions of the frozen filter --; (B) the record numbers
 *           are the running count of :162-185; (C) every thread owns a range of ranks and walks ALL records in file order,
 *           applying the literal append rule of :72 to the lists in its range only.
 * tests/test_oracle.py checks so_shark_build_mt == so_shark_build (filter words, lists) incl. the numbering quirk and
 * more than 65 536 genes. */
typedef struct {
  so_shark *s;
  const char *const *seqs;
  const uint64_t *lens;
  size_t n_records;
  int tid, nthreads;
  /* per record */
  uint32_t **ranks;      /* sorted by position: rank of every k-mer occurrence */
  uint32_t *cnt;         /* k-mers of the record (0: none valid, or shorter than k) */
  uint8_t *advances;     /* the record takes a gene number (everything but `continue` at main.cpp:165) */
  int *nidx_of;
} mt_job;

static size_t record_kmers(const char *seq, int seq_len, uint32_t k, uint64_t *kmers, int *skipped)
{
  /* main.cpp:163-182, as in so_shark_build */
  size_t cnt = 0;
  int _p = 0;
  *skipped = 0;
  uint64_t kmer = (uint64_t)so_build_kmer(seq, seq_len, &_p, (uint8_t)k);
  if (kmer == (uint64_t)-1) { *skipped = 1; return 0; }
  uint64_t rckmer = so_revcompl(kmer, (uint8_t)k);
  kmers[cnt++] = kmer < rckmer ? kmer : rckmer;
  for (int p = _p; p < seq_len; ++p) {
    uint8_t new_char = so_to_int(seq[p]);
    if (new_char == 0) {
      ++p;
      kmer = (uint64_t)so_build_kmer(seq, seq_len, &p, (uint8_t)k);
      if (kmer == (uint64_t)-1) break;
      rckmer = so_revcompl(kmer, (uint8_t)k);
      --p;
    } else {
      --new_char;
      kmer = so_lsappend(kmer, new_char, k);
      rckmer = so_rsprepend(rckmer, so_reverse_char(new_char), k);
    }
    kmers[cnt++] = kmer < rckmer ? kmer : rckmer;
  }
  return cnt;
}

static void *mt_pass1(void *arg)
{
  mt_job *j = (mt_job *)arg;
  so_bf *b = j->s->bf;
  const uint32_t k = j->s->k;
  for (size_t r = (size_t)j->tid; r < j->n_records; r += (size_t)j->nthreads) {
    size_t n = (size_t)j->lens[r];
    if (n < k) continue;
    uint64_t *hashes = (uint64_t *)malloc((n - k + 1) * sizeof(uint64_t));
    size_t cnt = so_kmer_builder(j->seqs[r], n, k, hashes);
    for (size_t i = 0; i < cnt; ++i) {
      uint64_t p = hashes[i] % b->size;                       /* so_bf_add_at, atomically */
      __atomic_fetch_or(&b->bf[p >> 6], (uint64_t)1 << (p & 63), __ATOMIC_RELAXED);
    }
    free(hashes);
  }
  return NULL;
}

static void *mt_pass2a(void *arg)
{
  mt_job *j = (mt_job *)arg;
  so_bf *b = j->s->bf;
  const uint32_t k = j->s->k;
  for (size_t r = (size_t)j->tid; r < j->n_records; r += (size_t)j->nthreads) {
    int seq_len = (int)j->lens[r];
    j->cnt[r] = 0;
    j->ranks[r] = NULL;
    j->advances[r] = 1;
    if ((unsigned)seq_len < k) continue;
    uint64_t *kmers = (uint64_t *)malloc(((size_t)seq_len - k + 1) * sizeof(uint64_t));
    int skipped = 0;
    size_t cnt = record_kmers(j->seqs[r], seq_len, k, kmers, &skipped);
    if (skipped) { j->advances[r] = 0; free(kmers); continue; }
    for (size_t i = 0; i < cnt; ++i) kmers[i] = so_get_hash(kmers[i]) % b->size;   /* bloomfilter.h:65-67 */
    qsort(kmers, cnt, sizeof(uint64_t), cmp_u64);                                   /* :68 */
    uint32_t *rk = (uint32_t *)malloc((cnt ? cnt : 1) * sizeof(uint32_t));
    for (size_t i = 0; i < cnt; ++i) rk[i] = (uint32_t)so_bf_rank(b, kmers[i]);     /* :70 */
    free(kmers);
    j->ranks[r] = rk;
    j->cnt[r] = (uint32_t)cnt;
  }
  return NULL;
}

static void *mt_pass2c(void *arg)
{
  mt_job *j = (mt_job *)arg;
  so_bf *b = j->s->bf;
  const uint64_t per = (b->num_kmer + (uint64_t)j->nthreads - 1) / (uint64_t)j->nthreads;
  const uint64_t lo = per * (uint64_t)j->tid, hi = lo + per;
  for (size_t r = 0; r < j->n_records; ++r) {
    const uint32_t *rk = j->ranks[r];
    const uint32_t n = j->cnt[r];
    const int input_idx = j->nidx_of[r];
    for (uint32_t i = 0; i < n; ++i) {
      const uint32_t kr = rk[i];
      if (kr < lo || kr >= hi) continue;
      smallvec *sv = &b->set_index[kr];
      if (sv->n == 0 || (int)sv_back(sv) != input_idx) sv_push(sv, (uint16_t)input_idx);   /* :72, literally */
    }
  }
  return NULL;
}

static double mt_now(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void mt_run(void *(*fn)(void *), mt_job *jobs, int nthreads)
{
  pthread_t *th = (pthread_t *)malloc((size_t)nthreads * sizeof(pthread_t));
  for (int t = 0; t < nthreads; ++t) pthread_create(&th[t], NULL, fn, &jobs[t]);
  for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
  free(th);
}

/* so_bf_switch_mode(b, 1) with the popcounts of the rank support taken by several threads (an 8 GB filter is four seconds of
 * one thread's reading): every thread counts a range of 512-bit blocks into running sums of its own, the ranges' totals are
 * added up in order, and every thread lifts its range by what lies in front of it.  Same arrays, same values. */
typedef struct { so_bf *b; uint64_t lo, hi, sum, base; } rank_job;

static void *rank_count(void *arg)
{
  rank_job *j = (rank_job *)arg;
  uint64_t acc = 0;
  for (uint64_t blk = j->lo; blk < j->hi; ++blk) {
    j->b->brank[blk] = acc;
    for (int w = 0; w < 8; ++w) acc += (uint64_t)__builtin_popcountll(j->b->bf[blk * 8 + w]);
  }
  j->sum = acc;
  return NULL;
}

static void *rank_lift(void *arg)
{
  rank_job *j = (rank_job *)arg;
  for (uint64_t blk = j->lo; blk < j->hi; ++blk) j->b->brank[blk] += j->base;
  return NULL;
}

static int bf_rank_support_mt(so_bf *b, int nthreads)
{
  if (b->mode != 0) return 0;
  b->mode = 1;
  const uint64_t nblk = (b->nwords + 7) / 8 + 1;            /* as so_bf_switch_mode */
  b->brank = (uint64_t *)malloc((nblk + 1) * sizeof(uint64_t));
  rank_job *jobs = (rank_job *)calloc((size_t)nthreads, sizeof(rank_job));
  pthread_t *th = (pthread_t *)malloc((size_t)nthreads * sizeof(pthread_t));
  const uint64_t per = (nblk + (uint64_t)nthreads - 1) / (uint64_t)nthreads;
  for (int t = 0; t < nthreads; ++t) {
    jobs[t].b = b;
    jobs[t].lo = per * (uint64_t)t < nblk ? per * (uint64_t)t : nblk;
    jobs[t].hi = jobs[t].lo + per < nblk ? jobs[t].lo + per : nblk;
  }
  for (int t = 0; t < nthreads; ++t) pthread_create(&th[t], NULL, rank_count, &jobs[t]);
  for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
  uint64_t acc = 0;
  for (int t = 0; t < nthreads; ++t) { jobs[t].base = acc; acc += jobs[t].sum; }
  for (int t = 0; t < nthreads; ++t) pthread_create(&th[t], NULL, rank_lift, &jobs[t]);
  for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
  b->brank[nblk] = acc;
  free(th); free(jobs);
  b->num_kmer = so_bf_rank(b, b->size);                      /* :122 */
  if (b->num_kmer != 0) b->set_index = (smallvec *)calloc(b->num_kmer, sizeof(smallvec));   /* :123-124 */
  return 1;
}

int so_shark_build_mt(so_shark *s, const char *const *seqs, const uint64_t *lens, size_t n_records, int nthreads)
{
  if (nthreads <= 1) return so_shark_build(s, seqs, lens, n_records);
  mt_job *jobs = (mt_job *)calloc((size_t)nthreads, sizeof(mt_job));
  uint32_t **ranks = (uint32_t **)calloc(n_records ? n_records : 1, sizeof(uint32_t *));
  uint32_t *cnt = (uint32_t *)calloc(n_records ? n_records : 1, sizeof(uint32_t));
  uint8_t *advances = (uint8_t *)calloc(n_records ? n_records : 1, 1);
  int *nidx_of = (int *)calloc(n_records ? n_records : 1, sizeof(int));
  for (int t = 0; t < nthreads; ++t) {
    jobs[t].s = s; jobs[t].seqs = seqs; jobs[t].lens = lens; jobs[t].n_records = n_records; jobs[t].tid = t; jobs[t].nthreads = nthreads;
    jobs[t].ranks = ranks; jobs[t].cnt = cnt; jobs[t].advances = advances; jobs[t].nidx_of = nidx_of;
  }
  const int timing = getenv("SO_TIMING") != NULL;            /* (phase times of the test infrastructure's own build, to stderr) */
  double t0 = mt_now();
  mt_run(mt_pass1, jobs, nthreads);
  if (timing) { fprintf(stderr, "[oracle] pass 1 %.2f s\n", mt_now() - t0); t0 = mt_now(); }
  bf_rank_support_mt(s->bf, nthreads);                      /* main.cpp:148 */
  if (timing) { fprintf(stderr, "[oracle] rank support %.2f s\n", mt_now() - t0); t0 = mt_now(); }
  mt_run(mt_pass2a, jobs, nthreads);
  if (timing) { fprintf(stderr, "[oracle] pass 2a %.2f s\n", mt_now() - t0); t0 = mt_now(); }
  int nidx = 0;                                             /* main.cpp:156, :165, :185 */
  for (size_t r = 0; r < n_records; ++r) {
    nidx_of[r] = nidx;
    if (advances[r]) ++nidx;
  }
  if (s->bf->num_kmer) mt_run(mt_pass2c, jobs, nthreads);
  if (timing) { fprintf(stderr, "[oracle] pass 2c %.2f s\n", mt_now() - t0); t0 = mt_now(); }
  for (size_t r = 0; r < n_records; ++r) free(ranks[r]);
  free(ranks); free(cnt); free(advances); free(nidx_of); free(jobs);
  so_bf_switch_mode(s->bf, 2);                              /* :193 */
  if (timing) fprintf(stderr, "[oracle] lists flattened %.2f s\n", mt_now() - t0);
  s->nidx = nidx;
  return nidx;
}

/* ---- ordered map<int, ((cov, nk), last)>  (ReadAnalyzer.hpp:41-42) ------ */
typedef struct {
  int *key;
  unsigned *cov, *nk, *last;
  int n, cap;
} covmap;

static int covmap_find_or_insert(covmap *m, int key)
{
  int lo = 0, hi = m->n;
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (m->key[mid] < key) lo = mid + 1; else hi = mid;
  }
  if (lo < m->n && m->key[lo] == key) return lo;
  if (m->n == m->cap) {
    m->cap = m->cap ? m->cap * 2 : 16;
    m->key = (int *)realloc(m->key, m->cap * sizeof(int));
    m->cov = (unsigned *)realloc(m->cov, m->cap * sizeof(unsigned));
    m->nk = (unsigned *)realloc(m->nk, m->cap * sizeof(unsigned));
    m->last = (unsigned *)realloc(m->last, m->cap * sizeof(unsigned));
  }
  memmove(m->key + lo + 1, m->key + lo, (m->n - lo) * sizeof(int));
  memmove(m->cov + lo + 1, m->cov + lo, (m->n - lo) * sizeof(unsigned));
  memmove(m->nk + lo + 1, m->nk + lo, (m->n - lo) * sizeof(unsigned));
  memmove(m->last + lo + 1, m->last + lo, (m->n - lo) * sizeof(unsigned));
  m->key[lo] = key;
  m->cov[lo] = 0; m->nk[lo] = 0; m->last[lo] = 0; /* value-initialised ((0,0),0) */
  m->n++;
  return lo;
}

static void covmap_free(covmap *m) { free(m->key); free(m->cov); free(m->nk); free(m->last); }

static inline unsigned umin(unsigned a, unsigned b) { return a < b ? a : b; }

/* ReadAnalyzer.hpp:44-108, body of the per-read loop, with a caller-owned map */
static int analyze_read_with_map(const so_shark *s, covmap *m, const char *read_seq, size_t n,
                                 int *genes_out, int cap,
                                 unsigned *max_out, unsigned *maxk_out, unsigned *len_out)
{
  const unsigned k = s->k;
  const so_bf *bf = s->bf;
  const uint16_t *index_kmer = bf->index_kmer;
  m->n = 0;                                                 /* :45 */
  unsigned len = 0;
  for (unsigned pos = 0; pos < n; ++pos)                    /* :47-49 */
    len += so_to_int(read_seq[pos]) > 0 ? 1 : 0;
  if (len_out) *len_out = len;
  if (max_out) *max_out = 0;
  if (maxk_out) *maxk_out = 0;
  if (len >= k) {                                           /* :50 */
    int pos = 0;
    uint64_t kmer = (uint64_t)so_build_kmer(read_seq, (int)n, &pos, (uint8_t)k); /* :52 */
    if (kmer == (uint64_t)-1) return 0;                     /* :53 `continue` */
    uint64_t rckmer = so_revcompl(kmer, (uint8_t)k);        /* :54 */
    int first, second;
    so_bf_get_index(bf, kmer < rckmer ? kmer : rckmer, &first, &second); /* :55 */
    while (first <= second) {                               /* :56-62 */
      int e = covmap_find_or_insert(m, (int)index_kmer[first]);
      m->cov[e] += umin(k, (unsigned)pos - m->last[e]);
      m->nk[e] = 1;
      m->last[e] = (unsigned)(pos - 1);
      ++first;
    }
    for (; pos < (int)n; ++pos) {                           /* :64 */
      uint8_t new_char = so_to_int(read_seq[pos]);
      if (new_char == 0) {                                  /* :66-71 */
        ++pos;
        kmer = (uint64_t)so_build_kmer(read_seq, (int)n, &pos, (uint8_t)k);
        if (kmer == (uint64_t)-1) break;
        rckmer = so_revcompl(kmer, (uint8_t)k);
        --pos;
      } else {                                              /* :72-76 */
        --new_char;
        kmer = so_lsappend(kmer, new_char, k);
        rckmer = so_rsprepend(rckmer, so_reverse_char(new_char), k);
      }
      so_bf_get_index(bf, kmer < rckmer ? kmer : rckmer, &first, &second); /* :77 */
      while (first <= second) {                             /* :79-86 */
        int e = covmap_find_or_insert(m, (int)index_kmer[first]);
        m->cov[e] += umin(k, (unsigned)pos - m->last[e]);
        m->nk[e] += 1;
        m->last[e] = (unsigned)pos;
        ++first;
      }
    }
  }

  /* :90-102 arg-max with ties, ascending gene id */
  unsigned max = 0, maxk = 0;
  int ng = 0;
  for (int i = 0; i < m->n; ++i) {
    if (m->cov[i] == max && m->nk[i] == maxk) {
      if (ng < cap) genes_out[ng] = m->key[i];
      ++ng;
    } else if (m->cov[i] > max || (m->cov[i] == max && m->nk[i] > maxk)) {
      ng = 0;
      max = m->cov[i];
      maxk = m->nk[i];
      if (ng < cap) genes_out[ng] = m->key[i];
      ++ng;
    }
  }
  if (max_out) *max_out = max;
  if (maxk_out) *maxk_out = maxk;
  /* :104 -- unsigned max promoted to double, compared with c*len in double */
  if ((double)max >= s->c * (double)len && (!s->single || ng == 1)) return ng;
  return 0;
}

int so_analyze_read(const so_shark *s, const char *read, size_t n, int *genes_out, int cap,
                    unsigned *max_out, unsigned *maxk_out, unsigned *len_out)
{
  covmap m = {0};
  int r = analyze_read_with_map(s, &m, read, n, genes_out, cap, max_out, maxk_out, len_out);
  covmap_free(&m);
  return r;
}

/* ---- batch driver: main.cpp:66-77, :215-223 ----------------------------- */
typedef struct {
  const so_shark *s;
  uint64_t n;
  const char *seq1; const uint64_t *off1;
  const char *seq2; const uint64_t *off2;
  const char *qual1; const char *qual2;
  /* FastqSplitter state */
  pthread_mutex_t split_mtx;
  uint64_t next;
  /* ReadOutput state */
  pthread_mutex_t out_mtx;
  uint32_t *count;      /* per read */
  uint16_t **ids;       /* per read, malloc'd when count>0 */
} batch_job;

#define SO_CHUNK 50000 /* main.cpp:215 */

static void *batch_worker(void *arg)
{
  batch_job *j = (batch_job *)arg;
  const so_shark *s = j->s;
  covmap m = {0};
  /* chunk-local storage (FastqSplitter::output_t / ReadAnalyzer::output_t) */
  char **strs = (char **)calloc(SO_CHUNK, sizeof(char *));
  size_t *lens = (size_t *)calloc(SO_CHUNK, sizeof(size_t));
  int gcap = 64;
  int *genes = (int *)malloc(gcap * sizeof(int));
  for (;;) {
    uint64_t b, e;
    /* FastqSplitter::operator() under its mutex: builds the joined / masked
     * strings for up to 50 000 reads (FastqSplitter.hpp:47-93) */
    pthread_mutex_lock(&j->split_mtx);
    b = j->next;
    e = b + SO_CHUNK; if (e > j->n) e = j->n;
    j->next = e;
    for (uint64_t i = b; i < e; ++i) {
      size_t l1 = (size_t)(j->off1[i + 1] - j->off1[i]);
      size_t l2 = j->seq2 ? (size_t)(j->off2[i + 1] - j->off2[i]) : 0;
      char *str = (char *)malloc(l1 + l2 + 2);
      lens[i - b] = so_join_mask(j->seq1 + j->off1[i], l1, j->qual1 ? j->qual1 + j->off1[i] : NULL,
                                 j->seq2 ? j->seq2 + j->off2[i] : NULL, l2,
                                 j->qual2 ? j->qual2 + j->off2[i] : NULL,
                                 j->seq2 != NULL, (char)s->min_quality, str);
      strs[i - b] = str;
    }
    pthread_mutex_unlock(&j->split_mtx);
    if (b >= e) break;
    /* ReadAnalyzer::operator() -- lock free */
    uint32_t *cnt = (uint32_t *)calloc(e - b, sizeof(uint32_t));
    uint16_t **ids = (uint16_t **)calloc(e - b, sizeof(uint16_t *));
    for (uint64_t i = b; i < e; ++i) {
      int ng = analyze_read_with_map(s, &m, strs[i - b], lens[i - b], genes, gcap, NULL, NULL, NULL);
      if (ng > gcap) {
        gcap = ng; genes = (int *)realloc(genes, gcap * sizeof(int));
        ng = analyze_read_with_map(s, &m, strs[i - b], lens[i - b], genes, gcap, NULL, NULL, NULL);
      }
      cnt[i - b] = (uint32_t)ng;
      if (ng > 0) {
        ids[i - b] = (uint16_t *)malloc(ng * sizeof(uint16_t));
        for (int g = 0; g < ng; ++g) ids[i - b][g] = (uint16_t)genes[g];
      }
      free(strs[i - b]);
    }
    /* ReadOutput::operator() under its mutex */
    pthread_mutex_lock(&j->out_mtx);
    for (uint64_t i = b; i < e; ++i) { j->count[i] = cnt[i - b]; j->ids[i] = ids[i - b]; }
    pthread_mutex_unlock(&j->out_mtx);
    free(cnt); free(ids);
  }
  covmap_free(&m);
  free(strs); free(lens); free(genes);
  return NULL;
}

int so_classify_batch(const so_shark *s, uint64_t n,
                      const char *seq1, const uint64_t *off1,
                      const char *seq2, const uint64_t *off2,
                      const char *qual1, const char *qual2,
                      int nthreads, uint32_t *gene_off, uint16_t **gene_ids)
{
  if (nthreads < 1) nthreads = 1;
  if (s->min_quality != 0 && (!qual1 || (seq2 && !qual2))) return -1;
  batch_job j;
  memset(&j, 0, sizeof(j));
  j.s = s; j.n = n; j.seq1 = seq1; j.off1 = off1; j.seq2 = seq2; j.off2 = off2;
  j.qual1 = qual1; j.qual2 = qual2;
  pthread_mutex_init(&j.split_mtx, NULL);
  pthread_mutex_init(&j.out_mtx, NULL);
  j.count = (uint32_t *)calloc(n ? n : 1, sizeof(uint32_t));
  j.ids = (uint16_t **)calloc(n ? n : 1, sizeof(uint16_t *));
  pthread_t *th = (pthread_t *)malloc(nthreads * sizeof(pthread_t));
  for (int t = 0; t < nthreads; ++t) pthread_create(&th[t], NULL, batch_worker, &j);
  for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
  free(th);
  uint64_t tot = 0;
  for (uint64_t i = 0; i < n; ++i) { gene_off[i] = (uint32_t)tot; tot += j.count[i]; }
  gene_off[n] = (uint32_t)tot;
  uint16_t *out = (uint16_t *)malloc((tot ? tot : 1) * sizeof(uint16_t));
  for (uint64_t i = 0; i < n; ++i) {
    if (j.count[i]) {
      memcpy(out + gene_off[i], j.ids[i], j.count[i] * sizeof(uint16_t));
      free(j.ids[i]);
    }
  }
  *gene_ids = out;
  free(j.count); free(j.ids);
  pthread_mutex_destroy(&j.split_mtx);
  pthread_mutex_destroy(&j.out_mtx);
  return 0;
}

void so_free(void *p) { free(p); }
