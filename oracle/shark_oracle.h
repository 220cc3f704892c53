/*
 * shark_oracle.h -- CPU ORACLE for the shark k-mer classification hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference
 * algorithm (AlgoLab/shark, files cited per function as file:line relative to
 * the reference tree).  It is the checker for the HIP product path in
 * shark_amd/csrc; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may call it.  Nothing under shark_amd/ links, imports or
 * executes it.
 *
 * Parity pinning: the full reference cannot be compiled in this image (its
 * bloomfilter.h needs sdsl-lite v2.1.1, an un-vendored, empty git submodule),
 * so this oracle is pinned (a) end to end on the reference's only golden
 * vectors, example/*.truth.* (tests/golden/example), (b) on XXH64 known
 * answers produced by python-xxhash (tests/golden/xxh64_kat.json), and (c)
 * primitive by primitive against the sdsl-free reference headers compiled in
 * place into oracle/_ref (kmer_utils.hpp, xxhash.hpp, FastqSplitter.hpp,
 * ReadOutput.hpp, kseq.h).
 *
 * The restatement is deliberately LITERAL (rolling k-mers, restart on invalid
 * characters, ordered per-gene map, rank/select index) so that it shares no
 * algebra with the closed forms the HIP kernels use.
 */
#ifndef SHARK_ORACLE_H
#define SHARK_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- kmer_utils.hpp ---------------------------------------------------- */
uint8_t  so_to_int(char c);                                   /* :29-41 */
uint8_t  so_reverse_char(uint8_t c);                          /* :43-45 */
uint64_t so_revcompl(uint64_t kmer, uint8_t k);               /* :47-55 */
int64_t  so_build_kmer(const char *seq, int n, int *p, uint8_t k); /* :57-71 */
uint64_t so_lsappend(uint64_t kmer, uint64_t c, uint64_t k);  /* :73-75 */
uint64_t so_rsprepend(uint64_t kmer, uint64_t c, uint64_t k); /* :77-79 */
uint64_t so_get_hash(uint64_t kmer);                          /* :81-83 */
/* xxhash.hpp:495-500 restricted to what _get_hash needs, but kept general:
 * XXH64 of an arbitrary byte string (used to pin the 8-byte path on KATs). */
uint64_t so_xxh64(const void *data, size_t len, uint64_t seed);

/* ---- bloomfilter.h : class BF ------------------------------------------ */
typedef struct so_bf so_bf;
so_bf   *so_bf_new(uint64_t size_bits);                       /* :48-53 */
void     so_bf_free(so_bf *bf);
void     so_bf_add_at(so_bf *bf, uint64_t p);                 /* :57-59 */
void     so_bf_add_to_kmer(so_bf *bf, uint64_t *kmers, size_t n, int input_idx); /* :61-75 */
/* :78-102; returns inclusive [start_pos,end_pos] into index_kmer (end=-1, start=0 on miss) */
void     so_bf_get_index(const so_bf *bf, uint64_t kmer, int *start_pos, int *end_pos);
int      so_bf_switch_mode(so_bf *bf, int new_mode);          /* :111-188 */
/* introspection for tests */
uint64_t        so_bf_size(const so_bf *bf);
const uint64_t *so_bf_words(const so_bf *bf);     /* sdsl int_vector<1> layout: bit i = (w[i>>6]>>(i&63))&1 */
uint64_t        so_bf_num_kmer(const so_bf *bf);  /* number of set bits (after switch_mode(1)) */
uint64_t        so_bf_tot_idx(const so_bf *bf);   /* length of index_kmer (after switch_mode(2)) */
const uint16_t *so_bf_index_kmer(const so_bf *bf);
uint64_t        so_bf_rank(const so_bf *bf, uint64_t i);      /* ones in [0,i) */

/* ---- KmerBuilder.hpp:40-72 --------------------------------------------- */
/* hashes of every canonical k-mer of one sequence, appended to out (caller
 * guarantees room for max(0,n-k+1)); returns the number written. */
size_t   so_kmer_builder(const char *seq, size_t n, uint32_t k, uint64_t *out);

/* ---- FastqSplitter.hpp:47-93,104-113 (join + mask semantics) ----------- */
/* builds the classified string into out (room for l1+1+l2); returns its length */
size_t   so_join_mask(const char *s1, size_t l1, const char *q1,
                      const char *s2, size_t l2, const char *q2,
                      int paired, char min_quality, char *out);

/* ---- main.cpp orchestration + ReadAnalyzer.hpp ------------------------- */
typedef struct so_shark so_shark;
so_shark *so_shark_new(uint32_t k, double c, uint64_t bf_bits, int min_quality, int single);
void      so_shark_free(so_shark *s);
/* main.cpp:128-193: pass 1, switch_mode(1), pass 2, switch_mode(2) over the
 * FASTA records given in file order.  Returns final nidx (main.cpp:191). */
int       so_shark_build(so_shark *s, const char *const *seqs, const uint64_t *lens, size_t n_records);
/* the same index built on `nthreads` threads (the scale tests' 60 000-gene indices); see the comment at its definition for why
 * the result cannot differ */
int       so_shark_build_mt(so_shark *s, const char *const *seqs, const uint64_t *lens, size_t n_records, int nthreads);
const so_bf *so_shark_bf(const so_shark *s);

/* ReadAnalyzer.hpp:39-110 for ONE already joined/masked read string.  Writes
 * the kept gene indices (ascending) to genes_out (up to cap) and returns how
 * many were kept (may exceed cap; 0 = no association).  Optional debug outs:
 * max coverage, max k-mer count, len (valid bases), and the number of
 * distinct genes seen. */
int       so_analyze_read(const so_shark *s, const char *read, size_t n,
                          int *genes_out, int cap,
                          unsigned *max_out, unsigned *maxk_out, unsigned *len_out);

/* Batch form over the same SoA layout the C-ABI uses (include/shark_hip.h):
 * mate1 of read i = seq1[off1[i]..off1[i+1]); seq2/off2 NULL => single-end;
 * qual1/qual2 may be NULL when min_quality==0.  Mirrors main.cpp:66-77 /
 * :215-223: nthreads workers pull 50 000-read chunks under a mutex
 * (FastqSplitter.hpp:48), analyse lock-free, and publish under a second
 * mutex (ReadOutput.hpp:38); results are stored per read so the outcome is
 * order independent.  gene_off must have room for n+1 entries; *gene_ids is
 * malloc'd by the oracle (free with so_free).  Returns 0 on success. */
int       so_classify_batch(const so_shark *s, uint64_t n,
                            const char *seq1, const uint64_t *off1,
                            const char *seq2, const uint64_t *off2,
                            const char *qual1, const char *qual2,
                            int nthreads,
                            uint32_t *gene_off, uint16_t **gene_ids);
void      so_free(void *p);

#ifdef __cplusplus
}
#endif
#endif
