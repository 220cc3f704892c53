// fastq_block_reader.hpp -- block-parallel FASTQ ingest for the shark CLI
// (the reference's FastqSplitter parses under one mutex,
// FastqSplitter.hpp:48, kseq.h:177-218, which caps the whole tool at a few
// million reads/s).
//
// Fast path: a PLAIN (not gzip'd) file of STRICT four-line records is mmap'd,
// newline positions are indexed by several threads at once, and the records
// of a batch are copied into the structure-of-arrays batch in parallel.  The
// record rules are kseq's: name = header up to the first whitespace
// (kseq.h:188); sequence and quality are one line each of equal length.
// Anything else -- gzip input, CR/LF, multi-line records, a quality line of a
// different length, stray text between records -- is handed to the serial
// FastxReader from the first irregular record on, so the bytes delivered are
// always those the reference's parser would deliver.
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <new>
#include <cctype>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <string>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <utility>
#include <vector>

namespace shk {

// Persistent worker pool behind parallel_for: the reader calls it several times per batch, and
// spawning 16-128 std::threads per call cost more than some of the stages themselves.  One job at a
// time (callers on different threads queue up on run_m_); jobs must not call parallel_for again.
class WorkerPool {
 public:
  static WorkerPool &instance()
  {
    static WorkerPool p;
    return p;
  }
  // run task(i) for i in [0, n_tasks) on up to `width` threads (the caller is one of them)
  template <typename Task>
  void run(unsigned width, unsigned n_tasks, Task &&task)
  {
    if (n_tasks == 0) return;
    if (width <= 1 || n_tasks == 1) {
      for (unsigned i = 0; i < n_tasks; ++i) task(i);
      return;
    }
    std::lock_guard<std::mutex> serial(run_m_);
    grow(width - 1);
    std::function<void(unsigned)> fn = std::ref(task);
    {
      std::lock_guard<std::mutex> lk(m_);
      job_ = &fn;
      n_tasks_ = n_tasks;
      next_.store(0, std::memory_order_relaxed);
      helpers_ = std::min<unsigned>(width - 1, (unsigned)th_.size());
      active_ = helpers_;
      ++gen_;
    }
    cv_.notify_all();
    work(fn, n_tasks);
    std::unique_lock<std::mutex> lk(m_);
    done_cv_.wait(lk, [&] { return active_ == 0; });
    job_ = nullptr;
  }
  ~WorkerPool()
  {
    {
      std::lock_guard<std::mutex> lk(m_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto &t : th_) t.join();
  }

 private:
  WorkerPool() = default;
  void grow(unsigned want)
  {
    while (th_.size() < want) {
      const unsigned id = (unsigned)th_.size();
      th_.emplace_back([this, id] { loop(id); });
    }
  }
  void work(std::function<void(unsigned)> &fn, unsigned n_tasks)
  {
    for (unsigned i; (i = next_.fetch_add(1, std::memory_order_relaxed)) < n_tasks;) fn(i);
  }
  void loop(unsigned id)
  {
    uint64_t seen = 0;
    for (;;) {
      std::function<void(unsigned)> *fn;
      unsigned n;
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return stop_ || (gen_ != seen && id < helpers_); });
        if (stop_) return;
        seen = gen_;
        fn = job_;
        n = n_tasks_;
      }
      work(*fn, n);
      std::lock_guard<std::mutex> lk(m_);
      if (--active_ == 0) done_cv_.notify_one();
    }
  }
  std::vector<std::thread> th_;
  std::mutex m_, run_m_;
  std::condition_variable cv_, done_cv_;
  std::function<void(unsigned)> *job_ = nullptr;
  std::atomic<unsigned> next_{0};
  unsigned n_tasks_ = 0, helpers_ = 0, active_ = 0;
  uint64_t gen_ = 0;
  bool stop_ = false;
};

// f(begin, end, part) over `n_threads` contiguous parts of [0, n)
template <typename F>
inline void parallel_for(unsigned n_threads, size_t n, F f)
{
  if (n == 0) return;
  n_threads = (unsigned)std::max<size_t>(1, std::min<size_t>(n_threads, n));
  if (n_threads == 1) { f(0, n, 0u); return; }
  const size_t per = (n + n_threads - 1) / n_threads;
  WorkerPool::instance().run(n_threads, n_threads, [&](unsigned t) {
    const size_t b = std::min(n, per * t), e = std::min(n, per * (t + 1));
    if (b < e) f(b, e, t);
  });
}

// Large host buffers on transparent huge pages.  The readers' buffers are written once per batch by many threads at
// once; with 4 KiB pages every first touch is a page fault that takes the process's address-space lock, which is what made
// 32-128 reader threads slower than 16 (profiles/README.md).  Blocks of 2 MiB and more come from mmap + MADV_HUGEPAGE
// (512 times fewer faults where the kernel grants huge pages; ordinary pages otherwise), smaller ones from malloc.
inline void *big_alloc(size_t bytes)
{
  constexpr size_t HUGE = 2u << 20;
  if (bytes < HUGE) return malloc(bytes ? bytes : 1);
  const size_t len = (bytes + HUGE - 1) & ~(HUGE - 1);
  void *p = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (p == MAP_FAILED) return nullptr;
#ifdef MADV_HUGEPAGE
  (void)madvise(p, len, MADV_HUGEPAGE);
#endif
  return p;
}
inline void big_free(void *p, size_t bytes)
{
  constexpr size_t HUGE = 2u << 20;
  if (!p) return;
  if (bytes < HUGE) free(p);
  else munmap(p, (bytes + HUGE - 1) & ~(HUGE - 1));
}

// allocator that leaves chars uninitialised on resize (every byte is overwritten) and puts large blocks on huge pages
template <typename T>
struct NoInitAlloc {
  using value_type = T;
  NoInitAlloc() = default;
  template <typename U> NoInitAlloc(const NoInitAlloc<U> &) {}
  template <typename U> struct rebind { using other = NoInitAlloc<U>; };
  T *allocate(size_t n)
  {
    void *p = big_alloc(n * sizeof(T));
    if (!p) throw std::bad_alloc();
    return static_cast<T *>(p);
  }
  void deallocate(T *p, size_t n) { big_free(p, n * sizeof(T)); }
  template <typename U> void construct(U *p) noexcept { ::new (static_cast<void *>(p)) U; }
  template <typename U, typename... A> void construct(U *p, A &&...a) { ::new (static_cast<void *>(p)) U(std::forward<A>(a)...); }
  template <typename U> bool operator==(const NoInitAlloc<U> &) const { return true; }
  template <typename U> bool operator!=(const NoInitAlloc<U> &) const { return false; }
};

// a view of the next `n` strict records of one file
struct RecordBlock {
  const char *base = nullptr;          // mmap base
  std::vector<uint64_t, NoInitAlloc<uint64_t>> nl;            // offsets of the 4*n newlines (record r: lines 4r..4r+3)
  std::vector<uint32_t, NoInitAlloc<uint32_t>> id_len, seq_len;  // per record: name length (up to the first whitespace), sequence length
  uint64_t first = 0;                  // offset of the first record's '@'
  size_t n = 0;
  // line i spans [begin(i), nl[i])
  uint64_t begin(size_t i) const { return i == 0 ? first : nl[i - 1] + 1; }
};

class FastqMmap {   // (historic name: the window is now filled with parallel pread, see populate note below)
 public:
  explicit FastqMmap(const std::string &path, unsigned threads) : threads_(threads)
  {
    // (a pipe is not even opened here: the open would wait for its writer and, closed again, break it)
    struct stat st0;
    if (stat(path.c_str(), &st0) != 0 || !S_ISREG(st0.st_mode)) return;
    fd_ = ::open(path.c_str(), O_RDONLY);
    if (fd_ < 0) return;
    struct stat st;
    if (fstat(fd_, &st) != 0 || !S_ISREG(st.st_mode)) return;
    size_ = (uint64_t)st.st_size;
    if (size_ == 0) { ok_ = true; return; }
    unsigned char magic[2] = {0, 0};
    if (pread(fd_, magic, 2, 0) < 0) return;
    ok_ = !(size_ >= 2 && magic[0] == 0x1f && magic[1] == 0x8b);   // gzip -> not for the fast path
    (void)posix_fadvise(fd_, 0, 0, POSIX_FADV_SEQUENTIAL);
  }
  ~FastqMmap()
  {
    if (fd_ >= 0) ::close(fd_);
  }
  FastqMmap(const FastqMmap &) = delete;
  FastqMmap &operator=(const FastqMmap &) = delete;

  double t_read = 0, t_scan = 0, t_merge = 0, t_valid = 0;   // seconds per stage (verbose report)
  bool usable() const { return ok_; }
  bool at_end() const { return cur_ >= size_; }
  uint64_t cursor() const { return cur_; }
  uint64_t size() const { return size_; }

  // Index up to `want` strict records starting at the cursor.  Returns the
  // number of records that are certainly regular (may be < want at the end of
  // the file or at the first irregular record; `irregular` tells which).
  // The block points into this reader's window buffer: consume it before the
  // next call.
  size_t next_block(size_t want, RecordBlock &blk, bool &irregular)
  {
    irregular = false;
    blk.nl.clear();
    blk.n = 0;
    blk.first = cur_;
    blk.base = nullptr;
    if (cur_ >= size_) return 0;
    const size_t need = want * 4;
    uint64_t scan = cur_;
    double bytes_per_line = est_line_;
    win_start_ = cur_;
    while (blk.nl.size() < need && scan < size_) {
      const size_t missing = need - blk.nl.size();
      uint64_t win = (uint64_t)(missing * bytes_per_line * 1.05) + (1u << 16);
      const uint64_t end = std::min<uint64_t>(size_, scan + win);
      // window [win_start_, end) must be resident: read [scan, end) with all threads.
      // (An mmap of the file costs one minor fault per 4 KiB page -- ~0.6 us each, serialised on
      // the address-space lock -- which capped indexing at ~5 GB/s; pread into a reused buffer does not.)
      if (buf_.size() < end - win_start_) buf_.resize((size_t)((end - win_start_) * 1.25) + (1u << 20));
      char *const wbase = buf_.data() - win_start_;   // wbase[file offset] is the byte at that offset
      auto c0 = std::chrono::steady_clock::now();
      parallel_for(threads_, (size_t)(end - scan), [&](size_t b, size_t e, unsigned) {
        uint64_t off = scan + b;
        const uint64_t stop = scan + e;
        while (off < stop) {
          const ssize_t got = pread(fd_, wbase + off, (size_t)std::min<uint64_t>(stop - off, 1u << 30), (off_t)off);
          if (got <= 0) { memset(wbase + off, 0, (size_t)(stop - off)); break; }   // I/O error: NULs make the records irregular
          off += (uint64_t)got;
        }
      });
      base_ = wbase;
      auto c1 = std::chrono::steady_clock::now();
      t_read += std::chrono::duration<double>(c1 - c0).count();
      if (scan == cur_ && base_[cur_] != '@') { irregular = true; blk.base = base_; return 0; }   // kseq would skip to the next '@' / '>'
      const unsigned T = threads_;
      std::vector<std::vector<uint64_t>> &parts = parts_;   // (kept across calls: no fresh pages per batch)
      parts.resize(T);
      for (auto &v : parts) v.clear();
      parallel_for(T, (size_t)(end - scan), [&](size_t b, size_t e, unsigned t) {
        std::vector<uint64_t> &v = parts[t];
        v.reserve((size_t)((e - b) / std::max(1.0, bytes_per_line)) + 16);
        const char *p = base_ + scan + b, *pe = base_ + scan + e;
        while (p < pe) {
          const char *q = (const char *)memchr(p, '\n', (size_t)(pe - p));
          if (!q) break;
          v.push_back((uint64_t)(q - base_));
          p = q + 1;
        }
      });
      auto c2 = std::chrono::steady_clock::now();
      t_scan += std::chrono::duration<double>(c2 - c1).count();
      {
        // concatenate the per-thread offset lists (in order) with all threads
        std::vector<size_t> at(T + 1, blk.nl.size());
        for (unsigned t = 0; t < T; ++t) at[t + 1] = std::min(need, at[t] + parts[t].size());
        blk.nl.resize(at[T]);
        parallel_for(T, T, [&](size_t b, size_t e, unsigned) {
          for (size_t t = b; t < e; ++t)
            if (at[t + 1] > at[t]) memcpy(blk.nl.data() + at[t], parts[t].data(), (at[t + 1] - at[t]) * sizeof(uint64_t));
        });
      }
      t_merge += std::chrono::duration<double>(std::chrono::steady_clock::now() - c2).count();
      scan = end;
      if (!blk.nl.empty()) bytes_per_line = std::max(8.0, (double)(blk.nl.back() - cur_) / (double)blk.nl.size());
    }
    est_line_ = bytes_per_line;
    blk.base = base_;
    // a last record without a trailing newline is left to the serial reader
    size_t n = blk.nl.size() / 4;
    // validate in parallel: '@', '+', equal lengths, no '\r'
    auto c3 = std::chrono::steady_clock::now();
    std::vector<size_t> bad(threads_, (size_t)-1);
    blk.id_len.resize(n);
    blk.seq_len.resize(n);
    parallel_for(threads_, n, [&](size_t b, size_t e, unsigned t) {
      for (size_t r = b; r < e; ++r) {
        const uint64_t h0 = blk.begin(4 * r), h1 = blk.nl[4 * r];
        const uint64_t s0 = h1 + 1, s1 = blk.nl[4 * r + 1];
        const uint64_t p0 = s1 + 1, p1 = blk.nl[4 * r + 2];
        const uint64_t q0 = p1 + 1, q1 = blk.nl[4 * r + 3];
        bool good = base_[h0] == '@' && p1 > p0 && base_[p0] == '+' && (s1 - s0) == (q1 - q0) && h1 > h0 && s1 > s0;
        // characters kseq treats specially at the start of a sequence line, CR/LF, embedded NULs
        if (good && (base_[s0] == '@' || base_[s0] == '>' || base_[s0] == '+')) good = false;
        if (good && (base_[h1 - 1] == '\r' || base_[s1 - 1] == '\r' || base_[q1 - 1] == '\r')) good = false;
        if (good && (memchr(base_ + s0, 0, (size_t)(s1 - s0)) || memchr(base_ + q0, 0, (size_t)(q1 - q0)))) good = false;
        if (good) {
          // name = header up to the first whitespace (kseq.h:188); a NUL in it is irregular too
          uint64_t p = h0 + 1;
          while (p < h1 && !isspace((unsigned char)base_[p]) && base_[p] != 0) ++p;
          if (p < h1 && base_[p] == 0) good = false;
          blk.id_len[r] = (uint32_t)(p - (h0 + 1));
          blk.seq_len[r] = (uint32_t)(s1 - s0);
        }
        if (!good) { bad[t] = r; break; }
      }
    });
    t_valid += std::chrono::duration<double>(std::chrono::steady_clock::now() - c3).count();
    size_t first_bad = n;
    for (size_t v : bad) if (v != (size_t)-1) first_bad = std::min(first_bad, v);
    if (first_bad < n) { irregular = true; n = first_bad; }
    else if (n < want && (n * 4 < blk.nl.size() || (n ? blk.nl[4 * n - 1] + 1 : cur_) < size_)) irregular = true;   // trailing partial record
    blk.nl.resize(n * 4);
    blk.n = n;
    return n;
  }

  // consume the first n records of the last block
  void advance(const RecordBlock &blk, size_t n)
  {
    if (n) cur_ = blk.nl[4 * n - 1] + 1;
  }

 private:
  int fd_ = -1;
  const char *base_ = nullptr;   // window buffer biased by the window's file offset
  std::vector<char, NoInitAlloc<char>> buf_;
  std::vector<std::vector<uint64_t>> parts_;
  uint64_t size_ = 0, cur_ = 0, win_start_ = 0;
  bool ok_ = false;
  unsigned threads_;
  double est_line_ = 80.0;
};

}  // namespace shk
