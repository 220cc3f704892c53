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
#include <zlib.h>

#include <cctype>
#include <cstring>
#include <string>
#include <vector>

namespace shk {

struct FastxRecord {
  std::string name, seq, qual;
};

class FastxReader {
 public:
  explicit FastxReader(const std::string &path) : buf_(1 << 20)
  {
    f_ = gzopen(path.c_str(), "r");
    if (f_) gzbuffer(f_, 1 << 18);
  }
  ~FastxReader()
  {
    if (f_) gzclose(f_);
  }
  FastxReader(const FastxReader &) = delete;
  FastxReader &operator=(const FastxReader &) = delete;
  bool ok() const { return f_ != nullptr; }

  // restart parsing at a byte offset of the (uncompressed) input -- used when the block-parallel
  // reader hands an irregular tail over; the offset must be a record boundary
  bool seek(uint64_t off)
  {
    if (!f_ || gzseek(f_, (z_off_t)off, SEEK_SET) < 0) return false;
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
      const int n = gzread(f_, buf_.data(), (unsigned)buf_.size());
      if (n <= 0) {
        eof_ = true;
        return -1;
      }
      pos_ = 0;
      end_ = (size_t)n;
    }
    return (unsigned char)buf_[pos_++];
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
      const char *b = buf_.data() + pos_;
      const char *e = buf_.data() + end_;
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

  gzFile f_ = nullptr;
  std::vector<char> buf_;
  size_t pos_ = 0, end_ = 0;
  bool eof_ = false;
  int last_ = 0;
};

}  // namespace shk
