// fastx_reader.hpp -- FASTA/FASTQ record reader for the shark CLI.
//
// Own implementation over zlib (plain or gzip input, like gzopen in
// main.cpp:88,129,202) that honours the record rules the reference inherits
// from kseq.h:177-218:
//   * a record starts at the next '>' or '@';
//   * the name is the header up to the first whitespace (:188);
//   * sequence lines are concatenated until a line starts with '>', '@' or '+'
//     (:194-198), a trailing '\r' is dropped (:135);
//   * after '+', quality lines are concatenated until they are at least as
//     long as the sequence (:213); a length mismatch ends the stream (-2).
#pragma once
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <cctype>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "gzip_parallel.hpp"

namespace shk {

// ---------------------------------------------------------------------------
// Where the reader's bytes come from.  The reference reads everything through gzopen/gzread on the parsing thread
// (main.cpp:88,:129,:202); here decompression runs AHEAD of the parser on its own thread(s):
//   * BGZF (bgzip's blocked gzip, the usual form of compressed FASTQ): every block is an independent deflate stream
//     of known compressed size (BSIZE in the 'BC' extra field), so the blocks of a chunk are inflated in parallel;
//   * an ordinary gzip file of text (one member or many, any writer): cut into chunks that are inflated in parallel, two passes
//     (gzip_parallel.hpp); SHARK_GZ_SERIAL=1 in the environment turns that off;
//   * anything else (small files, a plain file, gzip of something that is not text): gzread on a read-ahead thread, double
//     buffered, so that inflating overlaps parsing.
// seek() (uncompressed offsets; used when the block reader hands a plain file over) restarts the read-ahead.
// ---------------------------------------------------------------------------
// A sample may be a pipe (`-1 <(zcat a.fq.gz)`: the reference's gzopen reads those too): whatever is read from a pipe is gone, and a
// second open waits for a writer that never comes -- so only regular files are looked into ahead of the reader, or opened twice.
inline bool regular_file(const std::string &path)
{
  struct stat st;
  return stat(path.c_str(), &st) == 0 && S_ISREG(st.st_mode);
}

class InflateAhead {
 public:
  explicit InflateAhead(const std::string &path, unsigned bgzf_threads = 4) : path_(path), bgzf_threads_(bgzf_threads ? bgzf_threads : 1)
  {
    regular_ = regular_file(path);
    if (!regular_) {       // straight to zlib's gzread, which takes gzip and plain text alike, opened once
      ok_ = start(0);
      return;
    }
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return;
    unsigned char h[18];
    const size_t got = fread(h, 1, sizeof(h), f);
    fclose(f);
    // gzip member with FEXTRA whose first subfield is BC/2: BGZF
    bgzf_ = got == 18 && h[0] == 0x1f && h[1] == 0x8b && h[2] == 8 && (h[3] & 4) && h[12] == 'B' && h[13] == 'C' && h[14] == 2 && h[15] == 0;
    ok_ = start(0);
  }
  ~InflateAhead() { stop(); }
  InflateAhead(const InflateAhead &) = delete;
  InflateAhead &operator=(const InflateAhead &) = delete;
  bool ok() const { return ok_; }
  bool is_bgzf() const { return bgzf_; }

  bool seek(uint64_t off)
  {
    stop();
    ok_ = start(off);
    return ok_;
  }

  // the next chunk of uncompressed bytes (valid until the next call); false at the end of the stream
  bool next(const char *&data, size_t &len)
  {
    if (pg_) return pg_->next(data, len);
    std::unique_lock<std::mutex> l(m_);
    if (have_) {   // give the chunk handed out last time back to the producer
      have_ = false;
      full_[cons_] = false;
      cons_ ^= 1;
      cv_.notify_all();
    }
    cv_.wait(l, [&] { return full_[cons_] || done_; });
    if (!full_[cons_]) return false;
    data = buf_[cons_].data();
    len = fill_[cons_];
    have_ = true;
    return true;
  }

 private:
  static constexpr size_t CHUNK = 8u << 20;

  bool start(uint64_t off)
  {
    done_ = false; quit_ = false; have_ = false; cons_ = 0;
    fallback_gz_ = false;   // (a restart reads the file from `off` again: whatever made the last pass fall back is met again)
    full_[0] = full_[1] = false;
    pg_.reset();
    if (regular_ && !bgzf_ && off == 0 && !getenv("SHARK_GZ_SERIAL")) {
      // ordinary gzip: the parallel two-pass inflate when the file qualifies (a gzip member of text, more than a few chunks)
      pg_.reset(new ParallelGunzip(path_, bgzf_threads_));
      if (pg_->usable()) return true;
      pg_.reset();
    }
    if (bgzf_ && off == 0) {
      raw_ = fopen(path_.c_str(), "rb");
      if (!raw_) return false;
    } else {
      bgzf_ = bgzf_ && off == 0;
      gz_ = gzopen(path_.c_str(), "r");
      if (!gz_) return false;
      gzbuffer(gz_, 1 << 18);
      if (off) {
        // (zlib seeks a plain file with lseek only once it has LOOKED at it -- right after gzopen it would skip by reading
        //  everything in front of `off`, a quarter of a second per 5 GB; one byte read settles that)
        char c;
        (void)gzread(gz_, &c, 1);
        if (gzseek(gz_, (z_off_t)off, SEEK_SET) < 0) return false;
      }
    }
    for (auto &b : buf_) if (b.size() < CHUNK + (1u << 16)) b.resize(CHUNK + (1u << 16));
    th_ = std::thread([this] { produce(); });
    return true;
  }

  void stop()
  {
    {
      std::lock_guard<std::mutex> l(m_);
      quit_ = true;
      cv_.notify_all();
    }
    if (th_.joinable()) th_.join();
    pg_.reset();
    if (gz_) { gzclose(gz_); gz_ = nullptr; }
    if (raw_) { fclose(raw_); raw_ = nullptr; }
  }

  void produce()
  {
    int p = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> l(m_);
        cv_.wait(l, [&] { return !full_[p] || quit_; });
        if (quit_) return;
      }
      const size_t n = raw_ ? pull_bgzf(buf_[p]) : pull_gz(buf_[p]);
      std::lock_guard<std::mutex> l(m_);
      if (n == 0) {
        done_ = true;
        cv_.notify_all();
        return;
      }
      fill_[p] = n;
      full_[p] = true;
      cv_.notify_all();
      p ^= 1;
    }
  }

  size_t pull_gz(std::vector<char> &out)
  {
    size_t n = 0;
    while (n < CHUNK) {
      const int got = gzread(gz_, out.data() + n, (unsigned)(CHUNK - n));
      if (got <= 0) break;
      n += (size_t)got;
    }
    return n;
  }

  // up to CHUNK bytes of output from whole BGZF blocks, inflated by bgzf_threads_ threads.  A block that is not BGZF
  // after all (a plain gzip member appended to the file, ...) ends the parallel path: the rest is inflated by zlib's
  // gzread from that compressed offset on (gzopen on a descriptor positioned there).
  size_t pull_bgzf(std::vector<char> &out)
  {
    if (fallback_gz_) return pull_gz(out);
    struct Blk { size_t coff, clen, uoff, ulen; };
    std::vector<Blk> blks;
    size_t cbytes = 0, ubytes = 0;
    cbuf_.clear();
    while (ubytes < CHUNK) {
      unsigned char h[18];
      const long at = ftell(raw_);
      const size_t got = fread(h, 1, 18, raw_);
      if (got == 0) break;
      const bool is_blk = got == 18 && h[0] == 0x1f && h[1] == 0x8b && h[2] == 8 && (h[3] & 4) && h[10] == 6 && h[11] == 0 && h[12] == 'B' &&
                          h[13] == 'C' && h[14] == 2 && h[15] == 0;
      if (!is_blk) {
        // not a BGZF block: hand the remainder to zlib
        fseek(raw_, at, SEEK_SET);
        if (blks.empty()) {
          const int fd = dup(fileno(raw_));
          lseek(fd, at, SEEK_SET);
          gz_ = gzdopen(fd, "r");
          if (!gz_) { ::close(fd); return 0; }
          fallback_gz_ = true;
          return pull_gz(out);
        }
        break;
      }
      const size_t bsize = (size_t)(h[16] | (h[17] << 8)) + 1;     // whole block, header and trailer included
      if (bsize < 26) return 0;
      const size_t body = bsize - 18;
      const size_t o = cbuf_.size();
      cbuf_.resize(o + body);
      if (fread(cbuf_.data() + o, 1, body, raw_) != body) return 0;
      const unsigned char *tr = reinterpret_cast<const unsigned char *>(cbuf_.data() + o + body - 4);
      const size_t isize = (size_t)tr[0] | ((size_t)tr[1] << 8) | ((size_t)tr[2] << 16) | ((size_t)tr[3] << 24);
      if (ubytes + isize > out.size()) { fseek(raw_, at, SEEK_SET); cbuf_.resize(o); break; }
      blks.push_back({o, body - 8, ubytes, isize});   // (the 8 trailer bytes CRC32 + ISIZE follow the deflate data)
      ubytes += isize;
      cbytes += bsize;
    }
    if (blks.empty()) return 0;
    const unsigned T = (unsigned)std::min<size_t>(bgzf_threads_, blks.size());
    std::vector<char> bad(T, 0);
    auto work = [&](unsigned t) {
      z_stream zs;
      for (size_t i = t; i < blks.size(); i += T) {
        memset(&zs, 0, sizeof(zs));
        if (inflateInit2(&zs, -15) != Z_OK) { bad[t] = 1; return; }
        zs.next_in = reinterpret_cast<Bytef *>(cbuf_.data() + blks[i].coff);
        zs.avail_in = (uInt)blks[i].clen;
        zs.next_out = reinterpret_cast<Bytef *>(out.data() + blks[i].uoff);
        zs.avail_out = (uInt)blks[i].ulen;
        const int rc = blks[i].ulen ? inflate(&zs, Z_FINISH) : Z_STREAM_END;
        bool good = rc == Z_STREAM_END && zs.avail_out == 0;
        inflateEnd(&zs);
        if (good) {   // the member's CRC32, as gzread checks it
          const unsigned char *tr = reinterpret_cast<const unsigned char *>(cbuf_.data() + blks[i].coff + blks[i].clen);
          const uint32_t want = (uint32_t)tr[0] | ((uint32_t)tr[1] << 8) | ((uint32_t)tr[2] << 16) | ((uint32_t)tr[3] << 24);
          good = (uint32_t)crc32(crc32(0L, Z_NULL, 0), reinterpret_cast<const Bytef *>(out.data() + blks[i].uoff), (uInt)blks[i].ulen) == want;
        }
        if (!good) { bad[t] = 1; return; }
      }
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < T; ++t) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
    for (char c : bad) if (c) return 0;   // corrupt block: the stream ends here (gzread would report an error as well)
    (void)cbytes;
    return ubytes;
  }

  std::string path_;
  unsigned bgzf_threads_;
  std::unique_ptr<ParallelGunzip> pg_;   // ordinary gzip, inflated in parallel (then none of the members below is in use)
  bool bgzf_ = false, ok_ = false, fallback_gz_ = false, regular_ = true;
  gzFile gz_ = nullptr;
  FILE *raw_ = nullptr;
  std::vector<char> buf_[2], cbuf_;
  size_t fill_[2] = {0, 0};
  bool full_[2] = {false, false};
  int cons_ = 0;
  bool have_ = false, done_ = false, quit_ = false;
  std::mutex m_;
  std::condition_variable cv_;
  std::thread th_;
};

struct FastxRecord {
  std::string name, seq, qual;
};

class FastxReader {
 public:
  explicit FastxReader(const std::string &path, unsigned inflate_threads = 4) : src_(path, inflate_threads) {}
  FastxReader(const FastxReader &) = delete;
  FastxReader &operator=(const FastxReader &) = delete;
  bool ok() const { return src_.ok(); }
  bool parallel_inflate() const { return src_.is_bgzf(); }

  // restart parsing at a byte offset of the (uncompressed) input -- used when the block-parallel
  // reader hands an irregular tail over; the offset must be a record boundary
  bool seek(uint64_t off)
  {
    if (!src_.seek(off)) return false;
    cur_ = nullptr;
    pos_ = end_ = 0;
    eof_ = false;
    last_ = 0;
    return true;
  }

  // returns the sequence length, or <0 at end of stream / malformed record
  int read(FastxRecord &r)
  {
    int c;
    if (last_ == 0) {
      while ((c = getc()) >= 0 && c != '>' && c != '@') {}
      if (c < 0) return -1;
      last_ = c;
    }
    r.name.clear();
    r.seq.clear();
    r.qual.clear();
    bool got = false;
    while ((c = getc()) >= 0) {
      got = true;
      if (isspace(c)) break;
      r.name.push_back((char)c);
    }
    if (!got) return -1;
    if (c >= 0 && c != '\n') skip_line();
    while ((c = getc()) >= 0 && c != '>' && c != '+' && c != '@') {
      if (c == '\n') continue;
      r.seq.push_back((char)c);
      append_line(r.seq);
    }
    if (c == '>' || c == '@') last_ = c;
    if (c != '+') return (int)r.seq.size();
    while ((c = getc()) >= 0 && c != '\n') {}
    if (c < 0) return -2;
    while (append_line(r.qual) && r.qual.size() < r.seq.size()) {}
    last_ = 0;
    if (r.qual.size() != r.seq.size()) return -2;
    return (int)r.seq.size();
  }

 private:
  int getc()
  {
    if (pos_ >= end_) {
      if (eof_) return -1;
      size_t n = 0;
      if (!src_.next(cur_, n) || n == 0) {
        eof_ = true;
        return -1;
      }
      pos_ = 0;
      end_ = n;
    }
    return (unsigned char)cur_[pos_++];
  }
  void skip_line()
  {
    int c;
    while ((c = getc()) >= 0 && c != '\n') {}
  }
  // appends the rest of the current line; false when nothing was left to read
  bool append_line(std::string &s)
  {
    bool got = false;
    for (;;) {
      if (pos_ >= end_) {
        const int c = getc();
        if (c < 0) break;
        --pos_;
      }
      got = true;
      const char *b = cur_ + pos_;
      const char *e = cur_ + end_;
      const char *nl = (const char *)memchr(b, '\n', (size_t)(e - b));
      if (nl) {
        s.append(b, nl);
        pos_ += (size_t)(nl - b) + 1;
        break;
      }
      s.append(b, e);
      pos_ = end_;
    }
    if (!got) return false;
    if (s.size() > 1 && s.back() == '\r') s.pop_back();
    return true;
  }

  InflateAhead src_;
  const char *cur_ = nullptr;
  size_t pos_ = 0, end_ = 0;
  bool eof_ = false;
  int last_ = 0;
};

}  // namespace shk
