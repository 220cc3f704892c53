// shark-fastq-parts -- prints the record-aligned byte ranges the `shark` CLI's parallel readers use (fastq_partition.hpp).
// Host-only (no GPU): lets the CPU tests check the partition against an independent parse, and shows a user how a
// sample would be split.   usage: shark-fastq-parts BATCH THREADS file_1.fq [file_2.fq]   |   shark-fastq-parts --records file
// Output: one JSON object: n_records (pairs in the strict part), batches = [[begin1, end1, begin2, end2, records, regular], ...]
#include <cstdio>
#include <cstdlib>
#include <string>

#include "fastq_lean_reader.hpp"
#include "fastq_partition.hpp"
#include "fastx_reader.hpp"

int main(int argc, char **argv)
{
  if (argc == 3 && std::string(argv[1]) == "--records") {
    // what the serial kseq-rule reader delivers (plain, gzip or BGZF input): record count, bases, FNV-1a of every field
    shk::FastxReader r(argv[2]);
    if (!r.ok()) { printf("{\"ok\": false}\n"); return 0; }
    shk::FastxRecord rec;
    unsigned long long n = 0, bases = 0, h = 1469598103934665603ull;
    while (r.read(rec) >= 0) {
      ++n;
      bases += rec.seq.size();
      for (const std::string *f : {&rec.name, &rec.seq, &rec.qual}) {
        for (char c : *f) h = (h ^ (unsigned char)c) * 1099511628211ull;
        h = (h ^ 0xFFu) * 1099511628211ull;
      }
    }
    printf("{\"ok\": true, \"records\": %llu, \"bases\": %llu, \"fnv\": \"%llx\", \"bgzf\": %s}\n", n, bases, h, r.parallel_inflate() ? "true" : "false");
    return 0;
  }
  if (argc >= 4 && std::string(argv[1]) == "--lean") {
    // what the CLI's reader threads + output stage deliver for a file (fastq_lean_reader.hpp): every batch through
    // lean_parse_range, every record read back through RecordFetcher (sparse and dense must agree); FNV-1a over name,
    // sequence, quality of the records of the leading regular batches -- comparable with --records on a strict file
    const uint64_t batch = strtoull(argv[2], nullptr, 10);
    const bool with_qual = argc > 4 && std::string(argv[4]) == "qual";
    shk::BatchTable t;
    uint64_t rl = 0;
    bool fixed = shk::fixed_record_file(argv[3], t, rl);
    if (fixed) {
      shk::fixed_record_batches(t, rl, batch, t.n_records);
    } else {
      if (t.fd >= 0) { ::close(t.fd); t.fd = -1; }
      std::vector<uint64_t> c;
      shk::count_file(argv[3], 4, t, c);
      if (!t.ok) { printf("{\"ok\": false}\n"); return 0; }
      shk::locate_batches(t, c, batch, t.n_records, 4);
      if (!t.ok) { printf("{\"ok\": false}\n"); return 0; }
    }
    shk::RecordLayout lay;
    {
      std::vector<char> head((size_t)std::min<uint64_t>(1u << 16, t.file_size));
      if (!head.empty() && shk::pread_all(t.fd, head.data(), 0, head.size())) shk::layout_of(head.data(), head.size(), lay);
    }
    const uint64_t nb = (t.n_records + batch - 1) / batch;
    unsigned long long n = 0, bases = 0, h = 1469598103934665603ull, n_fixed = 0;
    long long first_irregular = -1;
    auto mix = [&](const char *p, size_t len) {
      for (size_t i = 0; i < len; ++i) h = (h ^ (unsigned char)p[i]) * 1099511628211ull;
      h = (h ^ 0xFFu) * 1099511628211ull;
    };
    shk::LeanScratch sc;
    std::vector<char, shk::NoInitAlloc<char>> seq, qual;
    std::vector<uint64_t> off;
    shk::RecordFetcher sparse, dense;
    for (uint64_t i = 0; i < nb; ++i) {
      const size_t want = (size_t)std::min<uint64_t>(batch, t.n_records - i * batch);
      shk::BatchFilePart part;
      const size_t ok = shk::lean_parse_range(t.fd, t.off[i], t.off[i + 1], want, lay, with_qual, sc, seq, off, qual, part);
      if (ok < want) { first_irregular = (long long)i; break; }
      n_fixed += part.fixed_width != 0;
      dense.load_dense(part, 0, want);
      for (size_t r = 0; r < want; ++r) {
        shk::RecordFetcher::View a, b;
        if (!sparse.get(part, r, a) || !dense.get(part, r, b)) { printf("{\"ok\": false, \"why\": \"fetch\"}\n"); return 0; }
        const size_t sl = (size_t)(off[r + 1] - off[r]);
        if (a.seq_len != sl || b.seq_len != sl || a.id_len != b.id_len || memcmp(a.id, b.id, a.id_len) || memcmp(a.seq, seq.data() + off[r], sl) ||
            memcmp(b.seq, seq.data() + off[r], sl) || memcmp(a.qual, b.qual, sl) || (with_qual && memcmp(a.qual, qual.data() + off[r], sl))) {
          printf("{\"ok\": false, \"why\": \"fields of record %zu of batch %llu disagree\"}\n", r, (unsigned long long)i);
          return 0;
        }
        mix(a.id, a.id_len);
        mix(seq.data() + off[r], sl);
        mix(a.qual, sl);
        ++n;
        bases += sl;
      }
    }
    printf("{\"ok\": true, \"records\": %llu, \"bases\": %llu, \"fnv\": \"%llx\", \"first_irregular_batch\": %lld, \"batches\": %llu, "
           "\"fixed_width_file\": %s, \"fixed_width_batches\": %llu, \"avx2\": %s}\n",
           n, bases, h, first_irregular, (unsigned long long)nb, fixed ? "true" : "false", n_fixed, shk::cpu_has_avx2() ? "true" : "false");
    return 0;
  }
  if (argc < 4) {
    fprintf(stderr, "usage: %s BATCH THREADS file_1.fq [file_2.fq]\n", argv[0]);
    return 2;
  }
  const uint64_t batch = strtoull(argv[1], nullptr, 10);
  const unsigned threads = (unsigned)atoi(argv[2]);
  const bool paired = argc > 4;
  if (batch == 0 || threads == 0) return 2;
  shk::BatchTable t1, t2;
  std::vector<uint64_t> c1, c2;
  shk::count_file(argv[3], threads, t1, c1);
  if (paired) shk::count_file(argv[4], threads, t2, c2);
  if (!t1.ok || (paired && !t2.ok)) {
    printf("{\"ok\": false}\n");
    return 0;
  }
  const uint64_t n = paired ? std::min(t1.n_records, t2.n_records) : t1.n_records;
  shk::locate_batches(t1, c1, batch, n, threads);
  if (paired) shk::locate_batches(t2, c2, batch, n, threads);
  if (!t1.ok || (paired && !t2.ok)) {
    printf("{\"ok\": false}\n");
    return 0;
  }
  const uint64_t nb = (n + batch - 1) / batch;
  printf("{\"ok\": true, \"n_records\": %llu, \"records_1\": %llu, \"records_2\": %llu, \"batches\": [", (unsigned long long)n,
         (unsigned long long)t1.n_records, (unsigned long long)(paired ? t2.n_records : 0));
  shk::ParsedBatch p1, p2;
  for (uint64_t i = 0; i < nb; ++i) {
    const size_t want = (size_t)std::min<uint64_t>(batch, n - i * batch);
    const size_t ok1 = shk::parse_strict_batch(t1.fd, t1.off[i], t1.off[i + 1], want, p1);
    const size_t ok2 = paired ? shk::parse_strict_batch(t2.fd, t2.off[i], t2.off[i + 1], want, p2) : want;
    printf("%s[%llu, %llu, %llu, %llu, %zu, %s]", i ? ", " : "", (unsigned long long)t1.off[i], (unsigned long long)t1.off[i + 1],
           (unsigned long long)(paired ? t2.off[i] : 0), (unsigned long long)(paired ? t2.off[i + 1] : 0), want,
           (ok1 == want && ok2 == want) ? "true" : "false");
  }
  printf("]}\n");
  return 0;
}
