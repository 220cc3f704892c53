/*
 * ref_driver.cpp -- thin C exports over the sdsl-free REFERENCE headers,
 * compiled in place from /root/reference into oracle/_ref/libsharkref.so by
 * oracle/Makefile (test infrastructure; the reference sources are included
 * by path, never copied).
 *
 * What the reference can contribute without sdsl-lite:
 *   kmer_utils.hpp + xxhash.hpp   to_int, reverse_char, revcompl, build_kmer,
 *                                 lsappend, rsprepend, _get_hash
 *   FastqSplitter.hpp + kseq.h    record parsing, mate join, quality masking
 *   FastaSplitter.hpp + kseq.h    FASTA record parsing / legend order
 *   small_vector.hpp              per-set-bit gene list container
 * bloomfilter.h, KmerBuilder.hpp, BloomfilterFiller.hpp, ReadAnalyzer.hpp and
 * main.cpp include <sdsl/...> and are therefore NOT built here.
 */
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
#include <zlib.h>

#include "kseq.h"
KSEQ_INIT(gzFile, gzread)

#include "common.hpp"
#include "kmer_utils.hpp"
#include "small_vector.hpp"
#include "FastaSplitter.hpp"
#include "FastqSplitter.hpp"

extern "C" {

uint8_t ref_to_int(unsigned char c) { return c < 128 ? to_int[c] : 0; }
uint8_t ref_reverse_char(uint8_t c) { return reverse_char(c); }
uint64_t ref_revcompl(uint64_t kmer, uint8_t k) { return revcompl(kmer, k); }
int64_t ref_build_kmer(const char *seq, int n, int *p, uint8_t k)
{
  std::string s(seq, (size_t)n);
  return build_kmer(s, *p, k);
}
uint64_t ref_lsappend(uint64_t kmer, uint64_t c, uint64_t k) { return lsappend(kmer, c, k); }
uint64_t ref_rsprepend(uint64_t kmer, uint64_t c, uint64_t k) { return rsprepend(kmer, c, k); }
uint64_t ref_get_hash(uint64_t kmer) { return _get_hash(kmer); }
uint64_t ref_xxh64(const void *p, size_t len, uint64_t seed) { return xxh::xxhash<64>(p, len, seed); }

/* small_vector.hpp: push all, read back */
size_t ref_small_vector(const uint16_t *in, size_t n, uint16_t *out, uint16_t *last)
{
  small_vector_t v;
  for (size_t i = 0; i < n; ++i) v.push_back(in[i]);
  size_t m = v.size();
  if (m) { std::memcpy(out, v.begin(), m * sizeof(uint16_t)); *last = v.last(); }
  return m;
}

/* FastaSplitter over a file: returns the records (name, seq) in legend order */
struct ref_fasta {
  std::vector<std::string> names, seqs;
};
void *ref_fasta_read(const char *path)
{
  gzFile f = gzopen(path, "r");
  if (!f) return nullptr;
  kseq_t *ks = kseq_init(f);
  ref_fasta *r = new ref_fasta();
  FastaSplitter fs(ks, 100, &r->names);
  while (auto *batch = fs()) {
    for (auto &p : *batch) r->seqs.push_back(p.second);
    delete batch;
  }
  kseq_destroy(ks);
  gzclose(f);
  return r;
}
size_t ref_fasta_count(void *h) { return ((ref_fasta *)h)->names.size(); }
const char *ref_fasta_name(void *h, size_t i) { return ((ref_fasta *)h)->names[i].c_str(); }
const char *ref_fasta_seq(void *h, size_t i) { return ((ref_fasta *)h)->seqs[i].c_str(); }
size_t ref_fasta_seq_len(void *h, size_t i) { return ((ref_fasta *)h)->seqs[i].size(); }
void ref_fasta_free(void *h) { delete (ref_fasta *)h; }

/* FastqSplitter over one or two files: the exact strings ReadAnalyzer sees */
struct ref_fastq {
  FastqSplitter::output_t reads;
};
void *ref_fastq_read(const char *p1, const char *p2, int min_quality)
{
  gzFile f1 = gzopen(p1, "r");
  if (!f1) return nullptr;
  gzFile f2 = p2 ? gzopen(p2, "r") : nullptr;
  kseq_t *k1 = kseq_init(f1);
  kseq_t *k2 = f2 ? kseq_init(f2) : nullptr;
  ref_fastq *r = new ref_fastq();
  {
    FastqSplitter fs(k1, k2, 50000, (char)min_quality, true);
    for (;;) {
      FastqSplitter::output_t chunk;
      fs(chunk);
      if (chunk.empty()) break;
      for (auto &e : chunk) r->reads.push_back(e);
    }
  }
  kseq_destroy(k1);
  gzclose(f1);
  if (k2) { kseq_destroy(k2); gzclose(f2); }
  return r;
}
size_t ref_fastq_count(void *h) { return ((ref_fastq *)h)->reads.size(); }
const char *ref_fastq_joined(void *h, size_t i, size_t *len)
{
  const std::string &s = ((ref_fastq *)h)->reads[i].first;
  *len = s.size();
  return s.data();
}
const char *ref_fastq_id(void *h, size_t i, int mate)
{
  auto &e = ((ref_fastq *)h)->reads[i];
  return mate == 0 ? e.second.first.id.c_str() : e.second.second.id.c_str();
}
const char *ref_fastq_seq(void *h, size_t i, int mate)
{
  auto &e = ((ref_fastq *)h)->reads[i];
  return mate == 0 ? e.second.first.seq.c_str() : e.second.second.seq.c_str();
}
const char *ref_fastq_qual(void *h, size_t i, int mate)
{
  auto &e = ((ref_fastq *)h)->reads[i];
  return mate == 0 ? e.second.first.qual.c_str() : e.second.second.qual.c_str();
}
void ref_fastq_free(void *h) { delete (ref_fastq *)h; }

} /* extern "C" */
