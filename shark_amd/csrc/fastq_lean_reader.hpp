// fastq_lean_reader.hpp -- the CLI's reader threads: one pass over a batch's bytes that checks every record and copies
// only what the GPU needs.
//
// The reference's splitter copies id, sequence and quality of every read into strings under one mutex
// (FastqSplitter.hpp:47-93) although ReadOutput prints those fields only for the reads that were associated with a
// gene (ReadOutput.hpp:37-50) -- a few per cent of a sample.  Here a reader thread streams the byte range of its
// batch through a small buffer that stays in its core's L2 (pread in 512 KiB pieces), checks each record against
// the block reader's rules (fastq_block_reader.hpp: '@' / '+', exactly four lines, equal sequence and quality
// lengths, no CR, no NUL, a sequence that does not start with '@' '>' '+') and copies the sequence (and, with -q, the
// quality) straight into the batch the GPU will read.  Names and qualities stay in the file; the output stage
// fetches them for the associated reads only (RecordFetcher).  A record that fails any check makes the batch
// irregular: it and everything behind it is re-read by the serial kseq-rule reader, exactly as before.
//
// Files whose records all have one layout (the same four line lengths as the first record -- simulators, SRA dumps,
// most fixed-length sequencer output) are checked 32 bytes at a time (AVX2 where the CPU has it): the newlines of a
// record must be exactly the four the layout expects and no byte may be NUL.  Other records take the line-by-line
// path (memchr).  Both accept exactly the same records.
#pragma once
#include <unistd.h>

#include <algorithm>
#include <cctype>
#include <cstdint>
#include <cstring>
#include <vector>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include "fastq_partition.hpp"

namespace shk {

// the four line lengths of a record, newline included (l1 == l3: sequence and quality)
struct RecordLayout {
  uint32_t l0 = 0, l1 = 0, l2 = 0, l3 = 0;
  uint32_t width() const { return l0 + l1 + l2 + l3; }
  bool usable() const { return l0 >= 2 && l1 >= 2 && l2 >= 2 && l1 == l3; }
};

// layout of the record that starts at p (avail bytes readable); false when it is not four complete lines
inline bool layout_of(const char *p, size_t avail, RecordLayout &lay)
{
  const char *q = p, *e = p + avail;
  uint32_t l[4];
  for (int i = 0; i < 4; ++i) {
    const char *n = (const char *)memchr(q, '\n', (size_t)(e - q));
    if (!n) return false;
    l[i] = (uint32_t)(n - q) + 1;
    q = n + 1;
  }
  lay.l0 = l[0]; lay.l1 = l[1]; lay.l2 = l[2]; lay.l3 = l[3];
  return true;
}

#if defined(__x86_64__)
// newlines of [p, p + w) as counted against the expected four, and "some byte is NUL"
__attribute__((target("avx2"))) inline bool scan_record_avx2(const char *p, uint32_t w, uint32_t &n_newlines, bool &has_nul)
{
  const __m256i nl = _mm256_set1_epi8('\n'), zero = _mm256_setzero_si256();
  uint32_t cnt = 0, z = 0, i = 0;
  for (; i + 32 <= w; i += 32) {
    const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(p + i));
    cnt += (uint32_t)__builtin_popcount((uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v, nl)));
    z |= (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v, zero));
  }
  if (i < w) {
    if (w >= 32) {   // the last 32 bytes again, without the ones already looked at
      const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(p + w - 32));
      const uint32_t keep = 0xFFFFFFFFu << (32 - (w - i));
      cnt += (uint32_t)__builtin_popcount((uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v, nl)) & keep);
      z |= (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v, zero)) & keep;
    } else {
      for (; i < w; ++i) { cnt += p[i] == '\n'; z |= p[i] == 0; }
    }
  }
  n_newlines = cnt;
  has_nul = z != 0;
  return true;
}
#endif

inline bool cpu_has_avx2()
{
#if defined(__x86_64__)
  static const bool have = __builtin_cpu_supports("avx2");
  return have;
#else
  return false;
#endif
}

// Is the record at p (at least lay.width() bytes readable) a strict record of exactly this layout?
inline bool record_has_layout(const char *p, const RecordLayout &lay)
{
  const uint32_t w = lay.width();
  const uint32_t e0 = lay.l0 - 1, e1 = e0 + lay.l1, e2 = e1 + lay.l2, e3 = w - 1;
  if (p[e0] != '\n' || p[e1] != '\n' || p[e2] != '\n' || p[e3] != '\n') return false;
  if (p[0] != '@' || p[e1 + 1] != '+') return false;
  const char s0 = p[e0 + 1];
  if (s0 == '@' || s0 == '>' || s0 == '+') return false;
  if (p[e0 - 1] == '\r' || p[e1 - 1] == '\r' || p[e3 - 1] == '\r') return false;
  uint32_t n_nl = 0;
  bool nul = false;
#if defined(__x86_64__)
  if (cpu_has_avx2()) {
    scan_record_avx2(p, w, n_nl, nul);
    return n_nl == 4 && !nul;
  }
#endif
  for (uint32_t i = 0; i < w; ++i) { n_nl += p[i] == '\n'; nul |= p[i] == 0; }
  return n_nl == 4 && !nul;
}

// where a record's fields are (offsets relative to its first byte); filled by parse_record
struct RecordFields {
  uint32_t id_len = 0;               // name: bytes [1, 1 + id_len)
  uint32_t seq_off = 0, seq_len = 0;
  uint32_t qual_off = 0;             // quality: seq_len bytes
  uint32_t rec_len = 0;              // whole record, last newline included
};

// One strict record at p by the block reader's rules (fastq_block_reader.hpp, parse_strict_batch).
// 1 = strict (f filled), 0 = irregular, -1 = the buffer ends inside the record (more bytes are needed)
inline int parse_record(const char *p, size_t avail, RecordFields &f)
{
  const char *e = p + avail;
  const char *n0 = (const char *)memchr(p, '\n', avail);
  if (!n0) return -1;
  const char *n1 = (const char *)memchr(n0 + 1, '\n', (size_t)(e - n0 - 1));
  if (!n1) return -1;
  const char *n2 = (const char *)memchr(n1 + 1, '\n', (size_t)(e - n1 - 1));
  if (!n2) return -1;
  const char *n3 = (const char *)memchr(n2 + 1, '\n', (size_t)(e - n2 - 1));
  if (!n3) return -1;
  const char *s0 = n0 + 1, *p0 = n1 + 1, *q0 = n2 + 1;
  bool good = p[0] == '@' && n2 > p0 && *p0 == '+' && (n1 - s0) == (n3 - q0) && n0 > p && n1 > s0;
  if (good && (*s0 == '@' || *s0 == '>' || *s0 == '+')) good = false;
  if (good && (n0[-1] == '\r' || n1[-1] == '\r' || n3[-1] == '\r')) good = false;
  if (good && (memchr(s0, 0, (size_t)(n1 - s0)) || memchr(q0, 0, (size_t)(n3 - q0)))) good = false;
  if (!good) return 0;
  const char *c = p + 1;
  while (c < n0 && !isspace((unsigned char)*c) && *c != 0) ++c;
  if (c < n0 && *c == 0) return 0;
  f.id_len = (uint32_t)(c - (p + 1));
  f.seq_off = (uint32_t)(s0 - p);
  f.seq_len = (uint32_t)(n1 - s0);
  f.qual_off = (uint32_t)(q0 - p);
  f.rec_len = (uint32_t)(n3 - p) + 1;
  return 1;
}

// One mate file's share of a batch, as the output stage needs it to fetch a record again
struct BatchFilePart {
  int fd = -1;
  const char *mem = nullptr;              // != nullptr: the batch's text is in memory (compressed samples: inflated text), not in a file
  uint64_t off0 = 0, off1 = 0;            // byte range of the batch in the file (in memory: [0, length))
  uint32_t fixed_width = 0;               // != 0: every record of this batch has this many bytes (record r starts at off0 + r * width)
  std::vector<uint64_t> rec_off;          // else: start of record r relative to off0, n + 1 entries
  uint64_t start_of(size_t r) const { return fixed_width ? (uint64_t)r * fixed_width : rec_off[r]; }
};

// Stream the byte range [off0, off1) of `fd`, which must hold exactly `want` strict records: sequences to seq (offsets in
// seq_off, want + 1 entries), qualities to qual when asked for.  Returns the number of leading strict records; `want` only when
// the range also ends with the last of them.  `hint`: the layout most records are expected to have (0-width: none).
struct LeanScratch {
  std::vector<char, NoInitAlloc<char>> buf;
};

template <typename Bytes, typename Offs>
inline size_t lean_parse_range(int fd, uint64_t off0, uint64_t off1, size_t want, const RecordLayout &hint, bool with_qual, LeanScratch &sc,
                               Bytes &seq, Offs &seq_off, Bytes &qual, BatchFilePart &part)
{
  constexpr size_t CHUNK = 512u << 10;
  const uint64_t len = off1 - off0;
  part.fd = fd; part.off0 = off0; part.off1 = off1; part.fixed_width = 0; part.rec_off.clear();
  if (sc.buf.size() < 2 * CHUNK) sc.buf.resize(2 * CHUNK);
  seq_off.resize(want + 1);
  // (an upper bound: a strict record spends more than half of its bytes outside the sequence)
  seq.resize((size_t)(len / 2 + 64));
  if (with_qual) qual.resize(seq.size());
  const uint32_t hw = hint.usable() ? hint.width() : 0;
  bool all_hint = hw != 0;          // every record so far had the hinted layout: record r starts at r * hw
  size_t have = 0;                  // bytes in the buffer
  uint64_t next = off0;             // file offset of the first byte not read yet
  uint64_t buf_file = off0;         // file offset of the buffer's first byte (always a record start)
  size_t r = 0;
  uint64_t so = 0;
  bool bad = false;
  while (r < want && !bad) {
    if (have == sc.buf.size()) sc.buf.resize(sc.buf.size() * 2);     // one record longer than the buffer
    char *const b = sc.buf.data();
    if (next < off1) {
      const size_t get = (size_t)std::min<uint64_t>(sc.buf.size() - have, off1 - next);
      if (!pread_all(fd, b + have, next, get)) { bad = true; break; }
      have += get;
      next += get;
    }
    const bool eof = next >= off1;
    size_t at = 0;
    while (r < want) {
      const size_t avail = have - at;
      RecordFields f;
      if (hw && avail >= hw && record_has_layout(b + at, hint)) {
        f.seq_off = hint.l0; f.seq_len = hint.l1 - 1; f.qual_off = hint.l0 + hint.l1 + hint.l2; f.rec_len = hw;
      } else {
        const int rc = parse_record(b + at, avail, f);
        if (rc < 0) {               // the buffer ends inside the record
          bad = eof;                // ... and so does the range: not a whole number of records
          break;
        }
        if (rc == 0) { bad = true; break; }
        if (all_hint) {
          // the first record of another shape: from here on record starts are kept explicitly
          part.rec_off.resize(r);
          for (size_t i = 0; i < r; ++i) part.rec_off[i] = (uint64_t)i * hw;
          all_hint = false;
        }
      }
      if (!all_hint) part.rec_off.push_back(buf_file + at - off0);
      if (so + f.seq_len > seq.size()) { seq.resize((size_t)((so + f.seq_len) * 2 + 64)); if (with_qual) qual.resize(seq.size()); }
      memcpy(seq.data() + so, b + at + f.seq_off, f.seq_len);
      if (with_qual) memcpy(qual.data() + so, b + at + f.qual_off, f.seq_len);
      seq_off[r] = so;
      so += f.seq_len;
      at += f.rec_len;
      ++r;
    }
    memmove(b, b + at, have - at);   // the partial record (if any) goes to the front
    have -= at;
    buf_file += at;
  }
  seq_off[r] = so;
  // surplus bytes behind the last record (or bytes of the range never read): not this batch's range (parse_strict_batch's rule)
  if (r == want && (have != 0 || next < off1)) r = 0;
  if (r == want) {
    seq.resize((size_t)so);
    if (with_qual) qual.resize((size_t)so);
    if (all_hint) part.fixed_width = hw; else part.rec_off.push_back(off1 - off0);
  }
  return r;
}

// The same over text that is already in memory (a compressed sample's inflated text, cut at record boundaries by the caller):
// [text, text + len) must hold exactly `want` strict records.  The text has to stay alive while the batch is in flight: the
// output stage reads names and qualities from it (part.mem).
template <typename Bytes, typename Offs>
inline size_t lean_parse_mem(const char *text, size_t len, size_t want, const RecordLayout &hint, bool with_qual, Bytes &seq, Offs &seq_off, Bytes &qual,
                             BatchFilePart &part)
{
  part.fd = -1; part.mem = text; part.off0 = 0; part.off1 = len; part.fixed_width = 0; part.rec_off.clear();
  seq_off.resize(want + 1);
  seq.resize(len / 2 + 64);
  if (with_qual) qual.resize(seq.size());
  const uint32_t hw = hint.usable() ? hint.width() : 0;
  bool all_hint = hw != 0;
  size_t at = 0, r = 0;
  uint64_t so = 0;
  while (r < want) {
    const size_t avail = len - at;
    RecordFields f;
    if (hw && avail >= hw && record_has_layout(text + at, hint)) {
      f.seq_off = hint.l0; f.seq_len = hint.l1 - 1; f.qual_off = hint.l0 + hint.l1 + hint.l2; f.rec_len = hw;
    } else {
      if (parse_record(text + at, avail, f) != 1) break;     // irregular, or the text ends inside the record
      if (all_hint) {
        part.rec_off.resize(r);
        for (size_t i = 0; i < r; ++i) part.rec_off[i] = (uint64_t)i * hw;
        all_hint = false;
      }
    }
    if (!all_hint) part.rec_off.push_back(at);
    if (so + f.seq_len > seq.size()) { seq.resize((size_t)((so + f.seq_len) * 2 + 64)); if (with_qual) qual.resize(seq.size()); }
    memcpy(seq.data() + so, text + at + f.seq_off, f.seq_len);
    if (with_qual) memcpy(qual.data() + so, text + at + f.qual_off, f.seq_len);
    seq_off[r] = so;
    so += f.seq_len;
    at += f.rec_len;
    ++r;
  }
  seq_off[r] = so;
  if (r == want && at != len) r = 0;          // surplus bytes behind the last record: not this batch's text
  if (r == want) {
    seq.resize((size_t)so);
    if (with_qual) qual.resize((size_t)so);
    if (all_hint) part.fixed_width = hw; else part.rec_off.push_back(len);
  }
  return r;
}

// The output stage's way back to a record's name, sequence and quality (ReadOutput.hpp:43-47 prints them for associated reads).
// sparse: one pread per record; dense (chosen by the caller when many records of a range are needed): the range is read in WINDOWS
// of about 256 KiB that follow the records asked for -- callers ask in ascending order --, so the text goes from the page cache
// through a buffer that stays in the core's L2 into the output text, instead of through a 16 MB copy of the whole range that is
// written to memory and read back (the command with half the sample written out again moves 4 KB of memory traffic per pair;
// that copy was a seventh of it).  Pointers stay valid until the next call.
class RecordFetcher {
 public:
  struct View { const char *id; uint32_t id_len; const char *seq; uint32_t seq_len; const char *qual; };
  // (SHARK_FETCH_WINDOW_KB: the window in KiB, 0 = the whole range at once -- for A/B timing)
  static uint64_t window_bytes()
  {
    static const uint64_t w = [] {
      const char *e = getenv("SHARK_FETCH_WINDOW_KB");
      return e ? (uint64_t)strtoull(e, nullptr, 10) << 10 : (uint64_t)256 << 10;
    }();
    return w;
  }
  // records [r0, r1) of `part` are about to be asked for, many of them, in ascending order
  bool load_dense(const BatchFilePart &part, size_t r0, size_t r1)
  {
    dense_ = false;
    if (part.mem) return true;                           // (text in memory: every record is there already)
    dense_ = true;
    r_end_ = r1;
    const uint64_t bytes = part.start_of(r1) - part.start_of(r0);
    const uint64_t W = window_bytes();
    per_window_ = (r1 > r0 && W) ? (size_t)std::max<uint64_t>(1, W * (uint64_t)(r1 - r0) / std::max<uint64_t>(bytes, 1)) : std::max<size_t>(1, r1 - r0);
    w0_ = w1_ = r0;                                      // (no window yet)
    return true;
  }
  void unload() { dense_ = false; }
  bool get(const BatchFilePart &part, size_t r, View &v)
  {
    const uint64_t a = part.start_of(r), e = part.start_of(r + 1);
    const char *p;
    if (part.mem) {
      p = part.mem + a;
    } else if (dense_ && r < r_end_) {
      if (r < w0_ || r >= w1_) {
        w0_ = r;
        w1_ = std::min(r_end_, r + per_window_);
        base_ = a;
        const uint64_t end = part.start_of(w1_);
        buf_.resize((size_t)(end - base_));
        if (!pread_all(part.fd, buf_.data(), part.off0 + base_, end - base_)) { w1_ = w0_; return false; }
      }
      p = buf_.data() + (a - base_);
    } else {
      one_.resize((size_t)(e - a));
      if (!pread_all(part.fd, one_.data(), part.off0 + a, e - a)) return false;
      p = one_.data();
    }
    RecordFields f;
    if (parse_record(p, (size_t)(e - a), f) != 1) return false;   // (cannot happen: the reader accepted this record)
    v.id = p + 1; v.id_len = f.id_len; v.seq = p + f.seq_off; v.seq_len = f.seq_len; v.qual = p + f.qual_off;
    return true;
  }

 private:
  std::vector<char, NoInitAlloc<char>> buf_, one_;
  uint64_t base_ = 0;
  size_t r_end_ = 0, per_window_ = 1, w0_ = 0, w1_ = 0;
  bool dense_ = false;
};

}  // namespace shk
