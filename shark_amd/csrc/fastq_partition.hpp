// fastq_partition.hpp -- record-aligned byte ranges of a FASTQ file (pair), so that MANY readers can
// parse one sample at once and feed several GPUs.
//
// The reference has one splitter: every worker thread pulls its next 50 000 reads through one mutex
// (FastqSplitter.hpp:48, main.cpp:219-223).  Here a pre-pass counts the newlines of a plain file in parallel
// (memory-speed), which gives the byte offset of every B-th record of a STRICT four-line file; batch i is then
// the byte range [off[i], off[i+1]) of each mate file, and any number of reader threads can parse batches
// independently -- pread, newline index, record validation, copy into a pinned structure-of-arrays batch.
// Nothing is assumed about the file that is not checked: a reader validates every record of its batch with the
// block reader's rules (fastq_block_reader.hpp); the first batch with an irregular record, and everything behind
// it, is re-read by the serial kseq-rule reader, so the records delivered are always the reference parser's.
#pragma once
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cctype>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "fastq_block_reader.hpp"

namespace shk {

constexpr uint64_t PART_SUB = 1u << 20;   // newline counts are kept per MiB of the file

struct BatchTable {
  bool ok = false;              // plain regular file that could be read
  uint64_t file_size = 0;
  uint64_t n_records = 0;       // whole four-line groups (newlines / 4)
  uint64_t batch = 0;           // records per batch
  std::vector<uint64_t> off;    // off[i] = byte offset of record i * batch, for i = 0 .. ceil(n_records / batch)
  int fd = -1;
  ~BatchTable()
  {
    if (fd >= 0) ::close(fd);
  }
  BatchTable() = default;
  BatchTable(const BatchTable &) = delete;
  BatchTable &operator=(const BatchTable &) = delete;
};

inline bool pread_all(int fd, char *dst, uint64_t off, uint64_t len)
{
  while (len) {
    const ssize_t got = pread(fd, dst, (size_t)std::min<uint64_t>(len, 1u << 30), (off_t)off);
    if (got <= 0) return false;
    dst += got; off += (uint64_t)got; len -= (uint64_t)got;
  }
  return true;
}

// number of '\n' in [p, p+n): a plain byte loop, which the compiler turns into wide compares
inline uint64_t count_newlines(const char *p, size_t n)
{
  uint64_t c = 0;
  for (size_t i = 0; i < n; ++i) c += p[i] == '\n';
  return c;
}

// Phase 1: newline counts per MiB (parallel; the whole file is read once, at memory speed from the page cache).
inline void count_file(const std::string &path, unsigned threads, BatchTable &t, std::vector<uint64_t> &cnt)
{
  t.ok = false;
  t.fd = ::open(path.c_str(), O_RDONLY);
  if (t.fd < 0) return;
  struct stat st;
  if (fstat(t.fd, &st) != 0 || !S_ISREG(st.st_mode)) return;
  t.file_size = (uint64_t)st.st_size;
  unsigned char magic[2] = {0, 0};
  if (t.file_size >= 2 && pread(t.fd, magic, 2, 0) == 2 && magic[0] == 0x1f && magic[1] == 0x8b) return;   // gzip: serial reader
  const uint64_t n_sub = (t.file_size + PART_SUB - 1) / PART_SUB;
  cnt.assign(n_sub + 1, 0);
  std::vector<char> bad(threads ? threads : 1, 0);
  parallel_for(threads, (size_t)n_sub, [&](size_t b, size_t e, unsigned tid) {
    std::vector<char> buf(PART_SUB);
    for (size_t s = b; s < e; ++s) {
      const uint64_t o = (uint64_t)s * PART_SUB, len = std::min<uint64_t>(PART_SUB, t.file_size - o);
      if (!pread_all(t.fd, buf.data(), o, len)) { bad[tid] = 1; return; }
      cnt[s + 1] = count_newlines(buf.data(), (size_t)len);
    }
  });
  for (char c : bad) if (c) return;
  for (uint64_t s = 0; s < n_sub; ++s) cnt[s + 1] += cnt[s];   // cnt[s] = newlines before sub-block s
  t.n_records = cnt[n_sub] / 4;
  t.ok = true;
}

// Shortcut for files whose records all have ONE byte length (fixed-width names and reads, e.g. simulators and SRA dumps):
// record i then starts at i * rec_len and no counting pass is needed.  The guess is cheap to refute -- the length of the
// first record, file size a multiple of it, and a sample of record starts that must read "\n@" -- and costs nothing if
// wrong elsewhere: every reader still validates every record of its batch, and a batch that is not strict four-line
// FASTQ hands over to the serial reader as always.  Returns false when the file does not look fixed-width.
inline bool fixed_record_file(const std::string &path, BatchTable &t, uint64_t &rec_len)
{
  t.ok = false;
  t.fd = ::open(path.c_str(), O_RDONLY);
  if (t.fd < 0) return false;
  struct stat st;
  if (fstat(t.fd, &st) != 0 || !S_ISREG(st.st_mode)) return false;
  t.file_size = (uint64_t)st.st_size;
  std::vector<char> head((size_t)std::min<uint64_t>(t.file_size, 1u << 16));
  if (head.empty() || !pread_all(t.fd, head.data(), 0, head.size())) return false;
  if ((unsigned char)head[0] == 0x1f) return false;   // gzip
  const char *p = head.data(), *pe = p + head.size();
  for (int l = 0; l < 4; ++l) {
    p = (const char *)memchr(p, '\n', (size_t)(pe - p));
    if (!p) return false;
    ++p;
  }
  rec_len = (uint64_t)(p - head.data());
  if (rec_len < 8 || t.file_size % rec_len) return false;
  t.n_records = t.file_size / rec_len;
  // 4096 record starts spread over the file
  char two[2];
  const uint64_t step = std::max<uint64_t>(1, t.n_records / 4096);
  for (uint64_t r = step; r < t.n_records; r += step) {
    if (pread(t.fd, two, 2, (off_t)(r * rec_len - 1)) != 2 || two[0] != '\n' || two[1] != '@') return false;
  }
  t.ok = true;
  return true;
}

inline void fixed_record_batches(BatchTable &t, uint64_t rec_len, uint64_t batch, uint64_t limit)
{
  t.batch = batch;
  const uint64_t n_b = (limit + batch - 1) / batch;
  t.off.assign(n_b + 1, 0);
  for (uint64_t i = 0; i <= n_b; ++i) t.off[i] = std::min<uint64_t>(i * batch, limit) * rec_len;
}

// Phase 2: off[i] = byte offset of record min(i * batch, limit) for i = 0 .. ceil(limit / batch); `limit` <= n_records is
// the number of records the pair stream has (it ends with the shorter mate file, FastqSplitter.hpp:60).
inline void locate_batches(BatchTable &t, const std::vector<uint64_t> &cnt, uint64_t batch, uint64_t limit, unsigned threads)
{
  if (!t.ok) return;
  t.batch = batch;
  const uint64_t n_b = (limit + batch - 1) / batch;
  t.off.assign(n_b + 1, 0);
  std::vector<char> bad(threads ? threads : 1, 0);
  parallel_for(threads, (size_t)n_b, [&](size_t b, size_t e, unsigned tid) {
    std::vector<char> buf(PART_SUB);
    for (size_t i = b; i < e; ++i) {
      // batch i+1 starts behind newline number `target` (1-based)
      const uint64_t target = 4 * std::min<uint64_t>((uint64_t)(i + 1) * batch, limit);
      const uint64_t s = (uint64_t)(std::lower_bound(cnt.begin(), cnt.end(), target) - cnt.begin()) - 1;   // cnt[s] < target <= cnt[s+1]
      const uint64_t o = s * PART_SUB, len = std::min<uint64_t>(PART_SUB, t.file_size - o);
      if (!pread_all(t.fd, buf.data(), o, len)) { bad[tid] = 1; return; }
      uint64_t need = target - cnt[s];
      const char *p = buf.data(), *pe = p + len;
      while (need) {
        p = (const char *)memchr(p, '\n', (size_t)(pe - p));
        if (!p) { bad[tid] = 1; return; }
        ++p;
        --need;
      }
      t.off[i + 1] = o + (uint64_t)(p - buf.data());
    }
  });
  for (char c : bad) if (c) t.ok = false;
}

// One batch of a strict four-line file, parsed by ONE thread: the bytes [b, e) of the file are read into `buf`,
// the newlines are indexed and every record is checked with the block reader's rules.  Returns the number of
// leading regular records (== want when the whole batch is regular).  The range has to hold EXACTLY `want` records: when
// the want-th record does not end at `e` (offsets derived arithmetically from a fixed-width guess that does not hold
// inside this range -- say two half-length records -- leave surplus bytes), nothing of the batch is accepted and the
// serial reader takes over from its first byte.
struct ParsedBatch {
  std::vector<char, NoInitAlloc<char>> buf;
  std::vector<uint64_t, NoInitAlloc<uint64_t>> nl;     // 4 per record, offsets into buf
  std::vector<uint32_t, NoInitAlloc<uint32_t>> id_len, seq_len;
  size_t n = 0;
  uint64_t begin(size_t i) const { return i == 0 ? 0 : nl[i - 1] + 1; }
};

inline size_t parse_strict_batch(int fd, uint64_t b, uint64_t e, size_t want, ParsedBatch &pb)
{
  pb.n = 0;
  const uint64_t len = e - b;
  if (pb.buf.size() < len) pb.buf.resize((size_t)(len + len / 8 + 4096));
  if (!pread_all(fd, pb.buf.data(), b, len)) return 0;
  const char *base = pb.buf.data();
  pb.nl.resize(want * 4);
  size_t got = 0;
  for (const char *p = base, *pe = base + len; got < want * 4 && p < pe;) {
    const char *q = (const char *)memchr(p, '\n', (size_t)(pe - p));
    if (!q) break;
    pb.nl[got++] = (uint64_t)(q - base);
    p = q + 1;
  }
  size_t n = got / 4;
  if (n > want) n = want;
  pb.id_len.resize(n);
  pb.seq_len.resize(n);
  size_t r = 0;
  for (; r < n; ++r) {
    const uint64_t h0 = pb.begin(4 * r), h1 = pb.nl[4 * r];
    const uint64_t s0 = h1 + 1, s1 = pb.nl[4 * r + 1];
    const uint64_t p0 = s1 + 1, p1 = pb.nl[4 * r + 2];
    const uint64_t q0 = p1 + 1, q1 = pb.nl[4 * r + 3];
    bool good = base[h0] == '@' && p1 > p0 && base[p0] == '+' && (s1 - s0) == (q1 - q0) && h1 > h0 && s1 > s0;
    if (good && (base[s0] == '@' || base[s0] == '>' || base[s0] == '+')) good = false;
    if (good && (base[h1 - 1] == '\r' || base[s1 - 1] == '\r' || base[q1 - 1] == '\r')) good = false;
    if (good && (memchr(base + s0, 0, (size_t)(s1 - s0)) || memchr(base + q0, 0, (size_t)(q1 - q0)))) good = false;
    if (!good) break;
    uint64_t p = h0 + 1;
    while (p < h1 && !isspace((unsigned char)base[p]) && base[p] != 0) ++p;
    if (p < h1 && base[p] == 0) break;
    pb.id_len[r] = (uint32_t)(p - (h0 + 1));
    pb.seq_len[r] = (uint32_t)(s1 - s0);
  }
  if (r == want && want && pb.nl[4 * want - 1] + 1 != len) r = 0;   // surplus bytes behind the last record: not this batch's range
  pb.n = r;
  return r;
}

// copy the first `want` records of a parsed batch into structure-of-arrays strings (names up to the first whitespace,
// sequences, qualities sharing the sequence offsets)
template <typename Ids, typename Seqs>
inline void fill_soa(const ParsedBatch &pb, size_t want, Ids &id, Seqs &seq, Seqs &qual)
{
  id.off.resize(want + 1);
  seq.off.resize(want + 1);
  qual.off.resize(want + 1);
  uint64_t ai = 0, as = 0;
  for (size_t k = 0; k < want; ++k) {
    id.off[k] = ai; seq.off[k] = as; qual.off[k] = as;
    ai += pb.id_len[k]; as += pb.seq_len[k];
  }
  id.off[want] = ai; seq.off[want] = as; qual.off[want] = as;
  id.bytes.resize(ai);
  seq.bytes.resize(as);
  qual.bytes.resize(as);
  const char *base = pb.buf.data();
  for (size_t k = 0; k < want; ++k) {
    memcpy(id.bytes.data() + id.off[k], base + pb.begin(4 * k) + 1, pb.id_len[k]);
    memcpy(seq.bytes.data() + seq.off[k], base + pb.nl[4 * k] + 1, pb.seq_len[k]);
    memcpy(qual.bytes.data() + qual.off[k], base + pb.nl[4 * k + 2] + 1, pb.seq_len[k]);
  }
}

}  // namespace shk
