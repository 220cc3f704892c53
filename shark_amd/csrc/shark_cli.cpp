// shark_cli.cpp -- the `shark` command line on top of libsharkhip.
//
// Drop-in for the reference binary: identical flags, defaults and validation
// (argument_parser.hpp:29-174), ssv on stdout (ReadOutput.hpp:43), surviving
// reads as FASTQ in -o/-p (ReadOutput.hpp:44-47), `[shark/...] Time elapsed`
// lines on stderr (main.cpp:47-54).  Output order is the reference's `-t 1`
// order (input order, genes ascending).
//
// Host structure mirrors main.cpp's three functor stages per worker loop
// (main.cpp:66-77): a splitter thread fills SoA batches (FastqSplitter role,
// but WITHOUT joining or masking -- the device does that), one analyzer thread
// per GPU calls shk_classify (ReadAnalyzer role), and the main thread writes
// batches in input order (ReadOutput role).  Extra flags: --gpus N, --batch N.
#include <getopt.h>
#include <sched.h>
#include <sys/stat.h>

#include <algorithm>
#include <chrono>
#include <fstream>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <iostream>
#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <type_traits>
#include <utility>
#include <vector>

#include "../../include/shark_hip.h"
#include "fastq_block_reader.hpp"
#include "fastq_lean_reader.hpp"
#include "fastq_partition.hpp"
#include "fastx_reader.hpp"
#include "gzip_parallel.hpp"

namespace {

// ---- argument_parser.hpp ---------------------------------------------------
const char *USAGE_MESSAGE =
    "Usage: shark -r <references> -1 <sample1> [OPTIONAL ARGUMENTS]\n"
    "\n"
    "Arguments:\n"
    "      -r, --reference                   reference sequences in FASTA format (can be gzipped)\n"
    "      -1, --sample1                     sample in FASTQ (can be gzipped)\n"
    "\n"
    "Optional arguments:\n"
    "      -h, --help                        display this help and exit\n"
    "      -2, --sample2                     second sample in FASTQ (optional, can be gzipped)\n"
    "      -o, --out1                        first output sample in FASTQ (default: sharked_sample.1)\n"
    "      -p, --out2                        second output sample in FASTQ (default: sharked_sample.2)\n"
    "      -k, --kmer-size                   size of the kmers to index (default:17, max:31)\n"
    "      -c, --confidence                  confidence for associating a read to a gene (default:0.6)\n"
    "      -b, --bf-size                     bloom filter size in GB (default:1)\n"
    "      -q, --min-base-quality            minimum base quality (assume FASTQ Illumina 1.8+ Phred scale, default:0, i.e., no filtering)\n"
    "      -s, --single                      report an association only if a single gene is found\n"
    "      -t, --threads                     number of threads (default:1)\n"
    "      -v, --verbose                     verbose mode\n"
    "\n"
    "MI355X build only:\n"
    "          --gpus N                      number of GPUs to shard the reads over (default:1)\n"
    "          --devices LIST                the devices of the N workers, comma separated (default:0,1,...,N-1; a device may be\n"
    "                                        named more than once: several workers then share it)\n"
    "          --batch N                     reads per device batch (default:65536, up to 262144 for large plain samples)\n"
    "          --gene-counts FILE            write <gene> <assigned reads> per gene (summed over the GPUs with RCCL)\n"
    "      -t N also sets the number of host threads that parse FASTQ / format output (default: up to 16)\n";

struct Options {
  std::string fasta_path, sample1_path, sample2_path, out1_path, out2_path;
  bool paired_flag = false;
  unsigned k = 17;
  double c = 0.6;
  uint64_t bf_size = (uint64_t)1 << 33;
  int min_quality = 0;     // as typed; the library narrows it to the reference's `char` (argument_parser.hpp:144)
  bool single = false, verbose = false;
  int nThreads = 1;
  int gpus = 1;
  bool gpus_given = false;
  std::vector<int> devices;     // --devices: worker g runs on devices[g] (empty: worker g on device g)
  uint64_t batch = 1u << 16;
  bool batch_given = false;     // (--batch; otherwise the sample's size decides: auto_batch below)
  std::string gene_counts_path;
};

// The command line is described by one table: option names, whether a value follows, and a handler that
// stores the value and applies that option's own check at once -- options are checked in the order they
// appear, as the reference does (argument_parser.hpp:84-174), so the first bad option decides the message.
// Flags, messages and exit codes are the contract (tests/test_cabi_cpu.py::test_cli_argument_contract).
[[noreturn]] void reject(const char *before, const char *message)
{
  std::cerr << before << message << std::endl << "aborting..." << std::endl;
  exit(EXIT_FAILURE);
}

// values are extracted the way operator>> does it (leading blanks skipped, trailing text ignored, 0 on failure)
template <typename T>
T value_of(const char *text)
{
  T v{};
  std::istringstream in(text ? text : "");
  in >> v;
  return v;
}

struct OptionRow {
  int key;                 // short option character, or >= 1000 for long-only options
  const char *name;
  bool takes_value;
  void (*apply)(Options &, const char *);
};

const OptionRow OPTION_TABLE[] = {
    {'r', "reference", true, [](Options &o, const char *v) { o.fasta_path = value_of<std::string>(v); }},
    {'t', "threads", true,
     [](Options &o, const char *v) {
       o.nThreads = value_of<int>(v);
       // (the reference prints the literal word here, argument_parser.hpp:95)
       if (o.nThreads <= 0) reject("USAGE_MESSAGE", "shark: at least 1 thread is required.");
     }},
    {'1', "sample1", true, [](Options &o, const char *v) { o.sample1_path = value_of<std::string>(v); }},
    {'2', "sample2", true, [](Options &o, const char *v) { o.sample2_path = value_of<std::string>(v); o.paired_flag = true; }},
    {'o', "out1", true, [](Options &o, const char *v) { o.out1_path = value_of<std::string>(v); }},
    {'p', "out2", true, [](Options &o, const char *v) { o.out2_path = value_of<std::string>(v); }},
    {'k', "kmer-size", true,
     [](Options &o, const char *v) {
       o.k = value_of<unsigned>(v);
       if (o.k < 1 || o.k > 31) reject(USAGE_MESSAGE, "shark: k must be in the range [1, 31].");
     }},
    {'c', "confidence", true,
     [](Options &o, const char *v) {
       o.c = value_of<double>(v);
       if (o.c < 0 || o.c > 1) reject("", "shark: c must be in the range [0, 1].");
     }},
    {'b', "bf-size", true, [](Options &o, const char *v) { o.bf_size = value_of<uint64_t>(v) << 33; /* GB -> bits */ }},
    {'q', "min-base-quality", true,
     [](Options &o, const char *v) {
       o.min_quality = value_of<int>(v);
       if (o.min_quality < 0) reject(USAGE_MESSAGE, "shark: q must be a positive value.");
     }},
    {'s', "single", false, [](Options &o, const char *) { o.single = true; }},
    {'v', "verbose", false, [](Options &o, const char *) { o.verbose = true; }},
    {'h', "help", false, [](Options &, const char *) { std::cerr << USAGE_MESSAGE; exit(EXIT_SUCCESS); }},
    {1000, "gpus", true, [](Options &o, const char *v) { o.gpus = std::max(1, value_of<int>(v)); o.gpus_given = true; }},
    {1003, "devices", true,
     [](Options &o, const char *v) {
       // a list of non-negative device numbers; anything else is refused (a typo must not silently become device 0)
       o.devices.clear();
       const std::string text = v ? v : "";
       size_t at = 0;
       while (at <= text.size()) {
         const size_t comma = std::min(text.find(',', at), text.size());
         const std::string item = text.substr(at, comma - at);
         if (item.empty() || item.size() > 4 || item.find_first_not_of("0123456789") != std::string::npos)
           reject("", "shark: --devices takes a comma separated list of device numbers.");
         o.devices.push_back(atoi(item.c_str()));
         at = comma + 1;
       }
     }},
    {1001, "batch", true, [](Options &o, const char *v) { o.batch = std::max<uint64_t>(1, value_of<uint64_t>(v)); o.batch_given = true; }},
    {1002, "gene-counts", true, [](Options &o, const char *v) { o.gene_counts_path = value_of<std::string>(v); }},
};

// Reads per device batch when --batch does not say.  A batch costs the device path 0.5-2 ms of launches, copies and bookkeeping
// whatever its size, and the ring of batch buffers (two dozen of them) is paid for in page faults at the start and at exit:
// 64 M pairs, 2 % on-target, sample phase 0.63 s at 65 536 pairs per batch, 0.47 s at 262 144, 0.44 s at 524 288; end to end 64 M
// pairs want 262 144 (half the sample written out again: 3.33 -> 2.75 s), 16 M pairs 65 536 (0.42 s against 0.52 s), compressed
// samples, whose size in records nobody knows up front, 65 536 (8 M pairs: 0.74 s against 0.88 s).  So: by the sample's size in
// records, estimated from the first record of a plain file.
inline uint64_t auto_batch(const std::string &sample1, uint64_t fallback)
{
  // (a regular file only: reading the head of a pipe would take it away from the reader)
  struct stat st;
  if (stat(sample1.c_str(), &st) != 0 || !S_ISREG(st.st_mode)) return fallback;
  FILE *f = fopen(sample1.c_str(), "rb");
  if (!f) return fallback;
  std::vector<char> head(1u << 16);
  const size_t got = fread(head.data(), 1, head.size(), f);
  uint64_t size = 0;
  if (fseeko(f, 0, SEEK_END) == 0) size = (uint64_t)ftello(f);
  fclose(f);
  if (got < 2 || ((unsigned char)head[0] == 0x1f && (unsigned char)head[1] == 0x8b)) return fallback;
  size_t rec = 0;
  int lines = 0;
  for (size_t i = 0; i < got && lines < 4; ++i)
    if (head[i] == '\n' && ++lines == 4) rec = i + 1;
  if (!rec) return fallback;
  const uint64_t records = size / rec;
  return records >= 48000000ull ? (1u << 18) : (records >= 24000000ull ? (1u << 17) : fallback);
}

Options parse_arguments(int argc, char **argv)
{
  // getopt_long's two descriptions are generated from the table
  std::string shorts;
  std::vector<struct option> longs;
  for (const OptionRow &row : OPTION_TABLE) {
    if (row.key < 256) {
      shorts.push_back((char)row.key);
      if (row.takes_value) shorts.push_back(':');
    }
    longs.push_back({row.name, row.takes_value ? required_argument : no_argument, nullptr, row.key});
  }
  longs.push_back({nullptr, 0, nullptr, 0});

  Options opt;
  int key;
  while ((key = getopt_long(argc, argv, shorts.c_str(), longs.data(), nullptr)) != -1) {
    const OptionRow *hit = nullptr;
    for (const OptionRow &row : OPTION_TABLE)
      if (row.key == key) hit = &row;
    if (!hit) {
      std::cerr << "shark : unknown argument" << std::endl << "\n" << USAGE_MESSAGE;
      exit(EXIT_FAILURE);
    }
    hit->apply(opt, optarg);
  }
  if (opt.fasta_path.empty() || opt.sample1_path.empty()) {
    std::cerr << "shark : missing required files" << std::endl << "\n" << USAGE_MESSAGE;
    exit(EXIT_FAILURE);
  }
  if (opt.out1_path.empty()) opt.out1_path = "sharked_sample.1";
  if (opt.out2_path.empty() && !opt.sample2_path.empty()) opt.out2_path = "sharked_sample.2";
  // --devices alone says how many workers there are; with --gpus N it has to name N devices
  if (!opt.devices.empty()) {
    if (!opt.gpus_given) opt.gpus = (int)opt.devices.size();
    if ((size_t)opt.gpus != opt.devices.size()) reject("", "shark: --devices must name as many devices as --gpus says.");
  } else {
    for (int g = 0; g < opt.gpus; ++g) opt.devices.push_back(g);
  }
  return opt;
}

// progress lines on stderr in the reference's format (main.cpp:49-54; whole seconds since start)
class Progress {
 public:
  void operator()(const std::string &stage) const
  {
    const auto secs = std::chrono::duration_cast<std::chrono::seconds>(std::chrono::steady_clock::now() - t0_).count();
    std::cerr << "[shark/" << stage << "] Time elapsed " << secs << std::endl;
  }

 private:
  std::chrono::steady_clock::time_point t0_ = std::chrono::steady_clock::now();
};
const Progress pelapsed;

// -v: a millisecond timeline of the run's phases on stderr (stderr is not part of the contract)
class Timeline {
 public:
  void on() { on_ = true; }
  void operator()(const char *what)
  {
    if (!on_) return;
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0_).count();
    const double epoch = std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count();
    std::lock_guard<std::mutex> l(m_);
    std::cerr << "[shark/ms] " << what << " " << ms << " (epoch " << std::fixed << epoch << std::defaultfloat << ")" << std::endl;
    // (SHARK_TRACE_RSS=1: the resident anonymous memory beside every stage -- what holds how much when)
    static const bool rss = getenv("SHARK_TRACE_RSS") != nullptr;
    if (rss) {
      std::ifstream st("/proc/self/status");
      std::string line;
      while (std::getline(st, line))
        if (line.compare(0, 8, "RssAnon:") == 0) std::cerr << "[shark/rss] " << what << " " << line.substr(8) << std::endl;
    }
  }

 private:
  std::chrono::steady_clock::time_point t0_ = std::chrono::steady_clock::now();
  bool on_ = false;
  std::mutex m_;
};
Timeline timeline;

// ---- one batch of reads, structure of arrays ---------------------------------
// allocator that leaves chars uninitialised on resize (the fillers overwrite every byte); large blocks on huge pages
template <typename T>
using default_init_allocator = shk::NoInitAlloc<T>;

// Batch buffers are page-locked once the HIP runtime is up (shk_alloc_pinned: 0.07 s per GB, tools/pin_cost.py; a recycled batch
// keeps its buffers); those allocated earlier -- the readers start while the runtime initialises -- are ordinary memory.
std::atomic<bool> g_pin_batches{false};

template <typename T>
struct pinned_allocator {
  using value_type = T;
  pinned_allocator() = default;
  template <typename U> pinned_allocator(const pinned_allocator<U> &) {}
  template <typename U> struct rebind { using other = pinned_allocator<U>; };
  T *allocate(size_t n)
  {
    // header word in front of the block: 1 = pinned, 0 = malloc
    const size_t bytes = n * sizeof(T) + 64;
    char *raw = g_pin_batches.load(std::memory_order_relaxed) ? static_cast<char *>(shk_alloc_pinned(bytes)) : nullptr;
    bool pinned = raw != nullptr;
    if (!raw) raw = static_cast<char *>(shk::big_alloc(bytes));
    if (!raw) throw std::bad_alloc();
    *reinterpret_cast<uint64_t *>(raw) = pinned ? 1 : 0;
    return reinterpret_cast<T *>(raw + 64);
  }
  void deallocate(T *p, size_t n)
  {
    char *raw = reinterpret_cast<char *>(p) - 64;
    if (*reinterpret_cast<uint64_t *>(raw)) shk_free_pinned(raw); else shk::big_free(raw, n * sizeof(T) + 64);
  }
  template <typename U> void construct(U *p) noexcept { ::new (static_cast<void *>(p)) U; }
  template <typename U, typename... A> void construct(U *p, A &&...a) { ::new (static_cast<void *>(p)) U(std::forward<A>(a)...); }
  template <typename U> bool operator==(const pinned_allocator<U> &) const { return true; }
  template <typename U> bool operator!=(const pinned_allocator<U> &) const { return false; }
};

template <typename CharAlloc, typename OffAlloc>
struct StringsT {
  std::vector<char, CharAlloc> bytes;
  std::vector<uint64_t, OffAlloc> off{0};
  void push(const char *p, size_t n) { bytes.insert(bytes.end(), p, p + n); off.push_back(bytes.size()); }
  size_t size() const { return off.size() - 1; }
  const char *at(size_t i) const { return bytes.data() + off[i]; }
  size_t len(size_t i) const { return (size_t)(off[i + 1] - off[i]); }
  void reset() { bytes.clear(); off.assign(1, 0); }
  void truncate(size_t keep) { off.resize(keep + 1); bytes.resize(off[keep]); }
};
using Strings = StringsT<default_init_allocator<char>, std::allocator<uint64_t>>;           // read names (host only)
using DevStrings = StringsT<pinned_allocator<char>, pinned_allocator<uint64_t>>;             // what the GPU reads

struct ReadBatch {
  uint64_t index = 0;       // position in the input stream (ordering contract)
  uint64_t first_read = 0;  // global index of its first read
  Strings id1, id2;
  DevStrings seq1, qual1, seq2, qual2;
  // records whose quality string does not have the sequence's length as the reference sees them (C strings: a NUL cuts the
  // sequence short, FastqSplitter.hpp:55,:63; a FASTA-style record has no qualities): the device reads qualities at the
  // sequence offsets, the output (ReadOutput.hpp:44-47) prints the quality string as it was
  std::map<size_t, std::string> qual_as_read1, qual_as_read2;
  // lean batches (the parallel readers', fastq_lean_reader.hpp): only what the GPU reads was copied; names and qualities of the
  // associated reads are fetched from the files again by the output stage
  bool lean = false;
  shk::BatchFilePart part1, part2;
  // compressed samples: the inflated text of this batch's records, per mate (part1.mem / part2.mem point into them); the buffers are
  // handed round between the cutters and the batches, never freed
  std::vector<char, default_init_allocator<char>> text1, text2;
  std::shared_ptr<const struct FormattedBatch> text;   // what the output stage will write for this batch (filled by a formatter thread)
  // result
  std::vector<uint32_t> gene_off;
  std::vector<uint16_t> gene_ids;
  int rc = 0;
  void reset()
  {
    id1.reset(); id2.reset(); seq1.reset(); qual1.reset(); seq2.reset(); qual2.reset();
    qual_as_read1.clear(); qual_as_read2.clear();
    lean = false;
    text.reset();
    gene_off.clear(); gene_ids.clear();
    rc = 0;
  }
};

// Batches are recycled.  Once the HIP runtime is up the pool becomes a RING of `limit` batches whose bases live in page-locked
// memory: filled once (0.07 s per GB to lock, tools/pin_cost.py), handed round for the whole sample -- their copies to the device
// are DMA at the link's rate, where a copy from ordinary memory makes the runtime lock and unlock the pages every time (which is
// what bounded the CLI's GPU phase: 4.8 GB of bases took 0.3 s on the submitting thread).  A small ring also means a small
// process: leaving it costs the kernel a few milliseconds instead of the 0.15 s it took to free gigabytes of parsed batches.
class BatchPool {
 public:
  // nullptr after shutdown()
  std::unique_ptr<ReadBatch> acquire()
  {
    std::unique_lock<std::mutex> l(m_);
    cv_.wait(l, [&] { return stop_ || !free_.empty() || limit_ == 0 || made_ < limit_; });
    if (stop_) return nullptr;
    if (!free_.empty()) {
      std::unique_ptr<ReadBatch> b = std::move(free_.back());
      free_.pop_back();
      return b;
    }
    ++made_;
    l.unlock();
    return std::unique_ptr<ReadBatch>(new ReadBatch());
  }
  void release(std::unique_ptr<ReadBatch> b)
  {
    b->reset();
    std::lock_guard<std::mutex> l(m_);
    if (limit_ || free_.size() < 64) free_.push_back(std::move(b));
    else --made_;
    cv_.notify_one();
  }
  // from now on at most `limit` batches exist; `reserve_bytes` per mate are page-locked for each right away (by the caller's thread:
  // concurrent page-locking from many threads is several times slower than one thread doing it all)
  void make_ring(size_t limit, size_t reserve_bytes, size_t reserve_reads, bool paired, bool with_qual)
  {
    std::vector<std::unique_ptr<ReadBatch>> fresh;
    for (size_t i = 0; i < limit; ++i) {
      std::unique_ptr<ReadBatch> b(new ReadBatch());
      b->seq1.bytes.reserve(reserve_bytes);
      b->seq1.off.reserve(reserve_reads + 1);
      if (with_qual) b->qual1.bytes.reserve(reserve_bytes);
      if (paired) {
        b->seq2.bytes.reserve(reserve_bytes);
        b->seq2.off.reserve(reserve_reads + 1);
        if (with_qual) b->qual2.bytes.reserve(reserve_bytes);
      }
      fresh.push_back(std::move(b));
    }
    std::lock_guard<std::mutex> l(m_);
    for (auto &b : fresh) free_.push_back(std::move(b));
    made_ += limit;
    limit_ = made_;
    cv_.notify_all();
  }
  void shutdown()
  {
    std::lock_guard<std::mutex> l(m_);
    stop_ = true;
    cv_.notify_all();
  }

 private:
  std::mutex m_;
  std::condition_variable cv_;
  std::vector<std::unique_ptr<ReadBatch>> free_;
  size_t made_ = 0, limit_ = 0;   // limit_ == 0: unbounded
  bool stop_ = false;
};

// FastqSplitter role (FastqSplitter.hpp:47-93): batches of reads in input order.
// Plain four-line FASTQ goes through the block-parallel reader; everything else
// (gzip, multi-line records, CR/LF, ...) through the serial kseq-rule reader.
class BatchSplitter {
 public:
  // (compressed samples: the host threads are shared between the mate files' inflaters)
  BatchSplitter(const Options &o, unsigned threads, BatchPool &pool)
      : r1_(o.sample1_path, std::max(2u, threads / (o.paired_flag ? 2u : 1u))), pool_(pool), paired_(o.paired_flag), maxnum_(o.batch), threads_(threads)
  {
    if (paired_) r2_.reset(new shk::FastxReader(o.sample2_path, std::max(2u, threads / 2u)));
    m1_.reset(new shk::FastqMmap(o.sample1_path, threads));
    if (paired_) m2_.reset(new shk::FastqMmap(o.sample2_path, threads));
    fast_ = m1_->usable() && (!paired_ || m2_->usable()) && !getenv("SHARK_SERIAL_READER");
  }
  bool ok() const { return r1_.ok() && (!paired_ || r2_->ok()); }
  bool fast_path() const { return fast_; }
  // continue with the serial kseq-rule reader at a record boundary (behind the batches the parallel feed delivered)
  void resume_serial(uint64_t off1, uint64_t off2, uint64_t next_index, uint64_t n_reads)
  {
    fast_ = false;
    r1_.seek(off1);
    if (paired_) r2_->seek(off2);
    next_index_ = next_index;
    n_reads_ = n_reads;
  }
  // continue with the serial kseq-rule reader behind the first `n` records of each mate file (compressed samples whose parallel
  // parse met an irregular record: strict records are the kseq reader's records, so the first n are simply read over)
  void skip_records(uint64_t n, uint64_t next_index)
  {
    fast_ = false;
    auto skip = [n](shk::FastxReader &r) {
      shk::FastxRecord a;
      for (uint64_t i = 0; i < n; ++i)
        if (r.read(a) < 0) break;
    };
    if (paired_) {
      std::thread t2([&] { skip(*r2_); });
      skip(r1_);
      t2.join();
    } else {
      skip(r1_);
    }
    next_index_ = next_index;
    n_reads_ = n;
  }
  std::string stage_report() const
  {
    std::ostringstream o;
    o << "read " << m1_->t_read + (m2_ ? m2_->t_read : 0) << " scan " << m1_->t_scan + (m2_ ? m2_->t_scan : 0) << " merge "
      << m1_->t_merge + (m2_ ? m2_->t_merge : 0) << " validate " << m1_->t_valid + (m2_ ? m2_->t_valid : 0);
    return o.str();
  }
  double t_index = 0, t_fill = 0, t_serial = 0;   // seconds spent (verbose report)
  std::unique_ptr<ReadBatch> operator()()
  {
    std::unique_ptr<ReadBatch> b = pool_.acquire();
    if (!b) return nullptr;
    b->index = next_index_++;
    b->first_read = n_reads_;
    if (fast_) {
      shk::RecordBlock &k1 = blk1_, &k2 = blk2_;   // reused: their buffers keep their capacity
      bool irr1 = false, irr2 = false;
      auto ta = std::chrono::steady_clock::now();
      size_t n = m1_->next_block(maxnum_, k1, irr1);
      if (paired_) n = std::min(n, m2_->next_block(maxnum_, k2, irr2));
      auto tb = std::chrono::steady_clock::now();
      t_index += std::chrono::duration<double>(tb - ta).count();
      if (n) {
        fill(k1, n, b->id1, b->seq1, b->qual1);
        m1_->advance(k1, n);
        if (paired_) {
          fill(k2, n, b->id2, b->seq2, b->qual2);
          m2_->advance(k2, n);
        }
      }
      t_fill += std::chrono::duration<double>(std::chrono::steady_clock::now() - tb).count();
      if (n < maxnum_) {
        // end of a file or an irregular record: the serial reader takes over from here
        fast_ = false;
        if (m1_->at_end() || (paired_ && m2_->at_end())) {
          done_ = true;   // a mate file is exhausted: the reference's read loop ends here too (FastqSplitter.hpp:53,60)
        } else {
          r1_.seek(m1_->cursor());
          if (paired_) r2_->seek(m2_->cursor());
        }
      }
    }
    if (!fast_ && !done_) {
      auto ts = std::chrono::steady_clock::now();
      // the two mate files are parsed (and, for .gz, inflated) by two threads at once; the pair
      // stream ends with the shorter file, as in the reference's read loop (FastqSplitter.hpp:60)
      const size_t have = b->seq1.size(), want = (size_t)maxnum_ - have;
      auto fill_serial = [want](shk::FastxReader &r, Strings &id, DevStrings &seq, DevStrings &qual, std::map<size_t, std::string> &qual_as_read) {
        shk::FastxRecord a;
        size_t got = 0;
        while (got < want && r.read(a) >= 0) {
          // the reference builds std::string from C strings (FastqSplitter.hpp:55,63): stop at NUL
          id.push(a.name.c_str(), strlen(a.name.c_str()));
          const size_t sl = strnlen(a.seq.data(), a.seq.size());
          seq.push(a.seq.data(), sl);
          // The device reads qualities at the sequence offsets.  The reference masks position i only for i < qual.length()
          // (FastqSplitter.hpp:104-109 with the C-string lengths of :55,:63): a record without a quality line, or one cut
          // short by a NUL, is not masked behind the end of its quality string -- those positions get the top quality.
          const size_t qfull = strnlen(a.qual.data(), a.qual.size());
          if (qfull != sl) qual_as_read[seq.size() - 1] = std::string(a.qual.data(), qfull);
          const size_t ql = std::min(sl, qfull);
          a.qual.resize(ql);
          a.qual.resize(sl, '\x7f');
          qual.push(a.qual.data(), sl);
          ++got;
        }
        return got;
      };
      size_t got1 = 0, got2 = 0;
      if (paired_) {
        std::thread t2([&] { got2 = fill_serial(*r2_, b->id2, b->seq2, b->qual2, b->qual_as_read2); });
        got1 = fill_serial(r1_, b->id1, b->seq1, b->qual1, b->qual_as_read1);
        t2.join();
        const size_t keep = have + std::min(got1, got2);
        if (got1 != got2) {   // one file ended: drop the unpaired surplus, nothing more will be read
          b->id1.truncate(keep); b->id2.truncate(keep);
          b->seq1.truncate(keep); b->qual1.truncate(keep); b->seq2.truncate(keep); b->qual2.truncate(keep);
          done_ = true;
        }
      } else {
        got1 = fill_serial(r1_, b->id1, b->seq1, b->qual1, b->qual_as_read1);
      }
      if (got1 < want) done_ = true;
      t_serial += std::chrono::duration<double>(std::chrono::steady_clock::now() - ts).count();
    }
    n_reads_ += b->seq1.size();
    if (b->seq1.size() == 0) { pool_.release(std::move(b)); return nullptr; }
    return b;
  }

 private:
  // copy n strict records into the structure-of-arrays strings, in parallel
  void fill(const shk::RecordBlock &k, size_t n, Strings &id, DevStrings &seq, DevStrings &qual)
  {
    const auto &idl = k.id_len, &sql = k.seq_len;   // measured while the block was validated
    id.off.resize(n + 1);
    seq.off.resize(n + 1);
    qual.off.resize(n + 1);
    uint64_t ai = 0, as = 0;
    for (size_t r = 0; r < n; ++r) {
      id.off[r] = ai; seq.off[r] = as; qual.off[r] = as;
      ai += idl[r]; as += sql[r];
    }
    id.off[n] = ai; seq.off[n] = as; qual.off[n] = as;
    id.bytes.resize(ai);
    seq.bytes.resize(as);
    qual.bytes.resize(as);
    shk::parallel_for(threads_, n, [&](size_t b, size_t e, unsigned) {
      for (size_t r = b; r < e; ++r) {
        memcpy(id.bytes.data() + id.off[r], k.base + k.begin(4 * r) + 1, idl[r]);
        memcpy(seq.bytes.data() + seq.off[r], k.base + k.nl[4 * r] + 1, sql[r]);
        memcpy(qual.bytes.data() + qual.off[r], k.base + k.nl[4 * r + 2] + 1, sql[r]);
      }
    });
  }

  shk::FastxReader r1_;
  BatchPool &pool_;
  std::unique_ptr<shk::FastxReader> r2_;
  std::unique_ptr<shk::FastqMmap> m1_, m2_;
  shk::RecordBlock blk1_, blk2_;
  bool paired_, fast_ = false, done_ = false;
  uint64_t maxnum_;
  unsigned threads_;
  uint64_t next_index_ = 0, n_reads_ = 0;
};

template <typename T>
class BoundedQueue {
 public:
  explicit BoundedQueue(size_t cap) : cap_(cap) {}
  void push(T v)
  {
    std::unique_lock<std::mutex> l(m_);
    cv_space_.wait(l, [&] { return q_.size() < cap_; });
    q_.push_back(std::move(v));
    cv_item_.notify_one();
  }
  bool pop(T &v)
  {
    std::unique_lock<std::mutex> l(m_);
    cv_item_.wait(l, [&] { return !q_.empty() || closed_; });
    if (q_.empty()) return false;
    v = std::move(q_.front());
    q_.pop_front();
    cv_space_.notify_one();
    return true;
  }
  // 1 = got an item, 0 = nothing there right now, -1 = closed and drained
  int try_pop(T &v)
  {
    std::lock_guard<std::mutex> l(m_);
    if (q_.empty()) return closed_ ? -1 : 0;
    v = std::move(q_.front());
    q_.pop_front();
    cv_space_.notify_one();
    return 1;
  }
  void close()
  {
    std::lock_guard<std::mutex> l(m_);
    closed_ = true;
    cv_item_.notify_all();
  }

 private:
  std::mutex m_;
  std::condition_variable cv_item_, cv_space_;
  std::deque<T> q_;
  size_t cap_;
  bool closed_ = false;
};


// ---- compressed samples ------------------------------------------------------------------------------------------------------
// The reference reads a .gz sample through gzread on the parsing thread.  Here the text comes out of the parallel inflaters
// (fastx_reader.hpp: BGZF blocks, or ordinary gzip in two passes, gzip_parallel.hpp) faster than one kseq-rule parser takes it
// (0.7 GB/s per file), so it is parsed like a plain file -- by several threads, sequences only, names and qualities read back
// for the associated reads --, from memory instead of from the file: one CUTTER per mate file takes the inflated text in order,
// counts newlines (32 bytes at a time) and cuts it into pieces of exactly --batch records (four lines each); the pieces of the two
// mates are joined by index and parsed by the parser threads (lean_parse_mem), which is also where a record that is not strict
// four-line FASTQ is noticed: that batch and everything behind it is then read by the serial reader, as for plain files.
struct GzPiece {
  std::vector<char, default_init_allocator<char>> text;
  size_t records = 0;      // whole groups of four lines in `text`
  bool last = false;       // the stream ended behind this piece
};

#if defined(__x86_64__)
__attribute__((target("avx2"))) inline size_t newlines_until_avx2(const char *p, size_t n, uint64_t need, uint64_t &found)
{
  // scans [p, p + n) until `need` newlines have been seen; returns the number of bytes scanned (ends right behind the need-th
  // newline when it is reached), found = newlines in the scanned part
  const __m256i nl = _mm256_set1_epi8('\n');
  size_t i = 0;
  uint64_t c = 0;
  for (; i + 32 <= n; i += 32) {
    const uint32_t m = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256(reinterpret_cast<const __m256i *>(p + i)), nl));
    const unsigned k = (unsigned)__builtin_popcount(m);
    if (c + k >= need) {
      uint32_t mm = m;
      for (uint64_t skip = need - c - 1; skip; --skip) mm &= mm - 1;      // drop the newlines in front of the wanted one
      found = need;
      return i + (size_t)__builtin_ctz(mm) + 1;
    }
    c += k;
  }
  for (; i < n; ++i)
    if (p[i] == '\n' && ++c == need) { found = c; return i + 1; }
  found = c;
  return n;
}
#endif
inline size_t newlines_until(const char *p, size_t n, uint64_t need, uint64_t &found)
{
#if defined(__x86_64__)
  if (shk::cpu_has_avx2()) return newlines_until_avx2(p, n, need, found);
#endif
  uint64_t c = 0;
  const char *q = p, *e = p + n;
  while (q < e) {
    const char *x = (const char *)memchr(q, '\n', (size_t)(e - q));
    if (!x) break;
    q = x + 1;
    if (++c == need) { found = c; return (size_t)(q - p); }
  }
  found = c;
  return n;
}

class GzCutter {
 public:
  GzCutter(const std::string &path, unsigned inflate_threads, uint64_t batch) : src_(path, inflate_threads), batch_(batch), q_(3) {}
  ~GzCutter() { stop(); }
  bool ok() const { return src_.ok(); }
  void start() { th_ = std::thread([this] { run(); }); }
  // the next piece in stream order; nullptr behind the last one
  std::unique_ptr<GzPiece> next()
  {
    std::unique_ptr<GzPiece> p;
    if (!q_.pop(p)) return nullptr;
    return p;
  }
  void recycle(std::unique_ptr<GzPiece> p)
  {
    std::lock_guard<std::mutex> l(m_);
    if (free_.size() < 8) free_.push_back(std::move(p));
  }
  // ends the cutter early (a failure elsewhere): whatever it still wants to hand over is dropped
  void stop()
  {
    quit_ = true;
    std::thread drain([this] { std::unique_ptr<GzPiece> p; while (q_.pop(p)) {} });
    if (th_.joinable()) th_.join();
    q_.close();
    drain.join();
  }
  double seconds = 0;   // spent scanning and copying (verbose report)

 private:
  std::unique_ptr<GzPiece> fresh()
  {
    {
      std::lock_guard<std::mutex> l(m_);
      if (!free_.empty()) {
        std::unique_ptr<GzPiece> p = std::move(free_.back());
        free_.pop_back();
        p->text.clear();
        p->records = 0;
        p->last = false;
        return p;
      }
    }
    return std::unique_ptr<GzPiece>(new GzPiece());
  }
  void run()
  {
    std::unique_ptr<GzPiece> cur = fresh();
    uint64_t lines = 0;                 // newlines in cur
    const uint64_t per = 4 * batch_;
    const char *d;
    size_t n;
    char last_byte = '\n';
    while (!quit_ && src_.next(d, n)) {
      auto t0 = std::chrono::steady_clock::now();
      size_t at = 0;
      while (at < n && !quit_) {
        uint64_t found = 0;
        const size_t used = newlines_until(d + at, n - at, per - lines, found);
        const size_t o = cur->text.size();
        cur->text.resize(o + used);
        memcpy(cur->text.data() + o, d + at, used);
        at += used;
        lines += found;
        if (lines == per) {
          cur->records = batch_;
          seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
          q_.push(std::move(cur));
          t0 = std::chrono::steady_clock::now();
          cur = fresh();
          lines = 0;
        }
      }
      if (n) last_byte = d[n - 1];
      seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    if (!quit_) {
      // a last record without its newline is a record all the same (kseq.h reads the quality up to the end of the file)
      if (!cur->text.empty() && last_byte != '\n') { cur->text.push_back('\n'); ++lines; }
      cur->records = (size_t)(lines / 4);
      cur->last = true;
      q_.push(std::move(cur));
    }
    q_.close();
  }
  shk::InflateAhead src_;
  uint64_t batch_;
  BoundedQueue<std::unique_ptr<GzPiece>> q_;
  std::mutex m_;
  std::vector<std::unique_ptr<GzPiece>> free_;
  std::atomic<bool> quit_{false};
  std::thread th_;
};

// ReadAnalyzer role (ReadAnalyzer.hpp:39-110): reads -> associations, on one GPU.  Up to SHK_PIPE_DEPTH batches are in
// flight: the copies of the next batches overlap the kernels of the current one (shk_classify_submit / _wait).
class ReadAnalyzer {
 public:
  ReadAnalyzer(shk_ctx *ctx, bool need_qual) : ctx_(ctx), need_qual_(need_qual) {}
  // false: the batch failed at once (b.rc is set) and is not in flight
  bool submit(std::unique_ptr<ReadBatch> b)
  {
    shk_batch in{};
    in.n = b->seq1.size();
    in.seq1 = b->seq1.bytes.data();
    in.off1 = b->seq1.off.data();
    if (b->seq2.size() == in.n && (b->lean || b->id2.size() == in.n) && in.n) {
      in.seq2 = b->seq2.bytes.data();
      in.off2 = b->seq2.off.data();
    }
    if (need_qual_) {
      // quality strings share the sequence offsets (kseq.h:216 guarantees equal lengths; the serial reader pads the rest)
      in.qual1 = b->qual1.bytes.data();
      if (in.seq2) in.qual2 = b->qual2.bytes.data();
    }
    uint64_t ticket = 0;
    b->rc = shk_classify_submit(ctx_, &in, &ticket);
    const bool ok = b->rc == SHK_OK;
    if (ok) flying_.emplace_back(ticket, std::move(b)); else failed_ = std::move(b);
    return ok;
  }
  size_t in_flight() const { return flying_.size(); }
  std::unique_ptr<ReadBatch> take_failed() { return std::move(failed_); }
  // the oldest batch in flight, classified
  std::unique_ptr<ReadBatch> wait()
  {
    std::unique_ptr<ReadBatch> b = std::move(flying_.front().second);
    const uint64_t ticket = flying_.front().first;
    flying_.pop_front();
    shk_result out{};
    b->rc = shk_classify_wait(ctx_, ticket, &out);
    if (b->rc == SHK_OK) {
      b->gene_off.assign(out.gene_off, out.gene_off + out.n + 1);
      b->gene_ids.assign(out.gene_ids, out.gene_ids + out.n_assoc);
    }
    return b;
  }

 private:
  shk_ctx *ctx_;
  bool need_qual_;
  std::deque<std::pair<uint64_t, std::unique_ptr<ReadBatch>>> flying_;
  std::unique_ptr<ReadBatch> failed_;
};

// ReadOutput role (ReadOutput.hpp:37-50).  The reference starts a new output
// call -- and clears previd -- every 50 000 input reads (main.cpp:215,
// ReadOutput.hpp:39), so 50 000-read chunks are independent and are formatted
// in parallel; the text is then written in input order.
// An output FASTQ file written at explicit offsets: the thread that emits the batches in order only assigns each piece its
// place in the file; the bytes are written by a few helper threads (pwrite), so an on-target-heavy sample -- gigabytes of output
// FASTQ -- is not throttled by one thread's write calls.  Not seekable (a pipe, a terminal): written in place, in order.
class OffsetWriter {
 public:
  OffsetWriter() = default;
  // (every error return of main() that runs while a writer is in scope comes through here: the helper threads are joined, never
  // destroyed while joinable -- that would be std::terminate instead of the exit code the caller was promised)
  ~OffsetWriter() { (void)close(); }
  OffsetWriter(const OffsetWriter &) = delete;
  OffsetWriter &operator=(const OffsetWriter &) = delete;
  bool open(const std::string &path, unsigned helpers)
  {
    fd_ = ::open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
    if (fd_ < 0) return false;
    seekable_ = lseek(fd_, 0, SEEK_CUR) != (off_t)-1;
    if (seekable_)
      for (unsigned i = 0; i < std::max(1u, helpers); ++i) th_.emplace_back([this] { work(); });
    return true;
  }
  bool is_open() const { return fd_ >= 0; }
  // `keep` keeps the bytes alive until they are written
  void append(const char *p, size_t n, const std::shared_ptr<const void> &keep)
  {
    if (!n) return;
    if (!seekable_) { write_all(p, n, (uint64_t)-1); return; }
    {
      std::unique_lock<std::mutex> l(m_);
      space_.wait(l, [&] { return q_.size() < 256; });   // (the text of at most that many pieces waits to be written)
      q_.push_back(Job{p, n, off_, keep});
    }
    off_ += n;
    cv_.notify_one();
  }
  bool close()
  {
    {
      std::lock_guard<std::mutex> l(m_);
      closing_ = true;
    }
    cv_.notify_all();
    for (auto &t : th_) t.join();
    th_.clear();
    const bool ok = !failed_.load() && (fd_ < 0 || ::close(fd_) == 0);
    fd_ = -1;
    return ok;
  }

 private:
  struct Job { const char *p; size_t n; uint64_t off; std::shared_ptr<const void> keep; };
  void write_all(const char *p, size_t n, uint64_t off)
  {
    while (n) {
      const ssize_t w = off == (uint64_t)-1 ? ::write(fd_, p, n) : ::pwrite(fd_, p, n, (off_t)off);
      if (w <= 0) { failed_ = true; return; }
      p += w; n -= (size_t)w;
      if (off != (uint64_t)-1) off += (uint64_t)w;
    }
  }
  void work()
  {
    for (;;) {
      Job j;
      {
        std::unique_lock<std::mutex> l(m_);
        cv_.wait(l, [&] { return !q_.empty() || closing_; });
        if (q_.empty()) return;
        j = std::move(q_.front());
        q_.pop_front();
        space_.notify_one();
      }
      const auto t0 = std::chrono::steady_clock::now();
      write_all(j.p, j.n, j.off);
      j.keep.reset();                                    // (the text's last owner frees it here, on this thread)
      busy_ns_ += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
      bytes_ += j.n;
    }
  }
 public:
  // (-v) bytes written by the helper threads and the seconds they spent writing and freeing
  uint64_t bytes_written() const { return bytes_.load(); }
  double busy_seconds() const { return 1e-9 * (double)busy_ns_.load(); }
 private:
  std::atomic<uint64_t> bytes_{0}, busy_ns_{0};
  int fd_ = -1;
  bool seekable_ = false, closing_ = false;
  uint64_t off_ = 0;
  std::atomic<bool> failed_{false};
  std::mutex m_;
  std::condition_variable cv_, space_;
  std::deque<Job> q_;
  std::vector<std::thread> th_;
};

// The text is produced per batch by any thread, in any order (format); the batches are then written in input order
// (emit), which is also where the one thing that crosses a batch boundary is settled: whether the first associated read
// of a batch that starts in the middle of a 50 000-read chunk repeats the previous batch's last read name.
struct FormattedSegment {
  std::string ssv, fq1, fq2;        // fq: the FASTQ records behind the segment's first one
  std::string head1, head2;         // the first associated read's FASTQ records, printed unless its name equals the carried one
  std::string head_id, last_id;
  bool has_assoc = false, carries = false;
  void reset()
  {
    ssv.clear(); fq1.clear(); fq2.clear(); head1.clear(); head2.clear(); head_id.clear(); last_id.clear();
    has_assoc = carries = false;
  }
};
struct FormattedBatch {
  std::vector<FormattedSegment> segs;   // (the first n of them are this batch's; the others keep their strings' memory for the next use)
  size_t n = 0;
};
// The text of a batch lives from its formatter to the writer threads' last write of it; then the object -- its strings keep their
// capacity -- goes back here for a later batch.  (Without the pool every batch's text was fresh memory: 20 GB of page faults on the
// formatter threads and as many pages unmapped by the writer threads, each unmap stopping every thread that faults, for a sample
// half of which is written out again.)  Never more objects than were alive at once.
class TextPool {
 public:
  std::shared_ptr<FormattedBatch> get()
  {
    static const bool off = getenv("SHARK_NO_TEXT_POOL") != nullptr;      // (A/B timing)
    if (off) return std::make_shared<FormattedBatch>();
    FormattedBatch *p = nullptr;
    {
      std::lock_guard<std::mutex> l(st_->m);
      if (!st_->free.empty()) { p = st_->free.back(); st_->free.pop_back(); }
    }
    if (!p) p = new FormattedBatch();
    // (the deleter owns the pool's state: a text may outlive this object on an error return)
    std::shared_ptr<State> st = st_;
    return std::shared_ptr<FormattedBatch>(p, [st](FormattedBatch *q) {
      std::lock_guard<std::mutex> l(st->m);
      st->free.push_back(q);
      st->cv.notify_one();
    });
  }
  // No text will be asked for any more (every batch is formatted): from now on `threads` helpers give the pages of every text
  // that comes back -- and of those that are back already -- to the system at once, side by side (MADV_DONTNEED takes the address
  // space's lock shared; unmapping takes it exclusively).  What the process still holds when it ends, its end has to give back on
  // ONE thread: 0.08 s per GB, 0.7-0.9 s for the 10 GB of a 64 M-pair sample half of which is written out again.
  void retire(unsigned threads)
  {
    for (unsigned t = 0; t < std::max(1u, threads); ++t)
      reapers_.emplace_back([st = st_] {
        for (;;) {
          FormattedBatch *q;
          {
            std::unique_lock<std::mutex> l(st->m);
            st->cv.wait(l, [&] { return !st->free.empty() || st->done; });
            if (st->free.empty()) return;
            q = st->free.back();
            st->free.pop_back();
          }
          for (FormattedSegment &sg : q->segs)
            for (std::string *x : {&sg.ssv, &sg.fq1, &sg.fq2}) drop_pages(*x);
          // (the object itself and its small strings are left to the process's end)
        }
      });
  }
  // every text is back (the writers are closed): the helpers finish what is left
  void finish()
  {
    {
      std::lock_guard<std::mutex> l(st_->m);
      st_->done = true;
    }
    st_->cv.notify_all();
    for (auto &t : reapers_) t.join();
    reapers_.clear();
  }
  // (an error return in between: the helpers are joined, never destroyed while joinable; the state is left to the process's end)
  ~TextPool() { finish(); new std::shared_ptr<State>(st_); }

 private:
  static void drop_pages(std::string &x)
  {
    if (x.capacity() < (1u << 20)) return;
    const uintptr_t a = ((uintptr_t)x.data() + 4095u) & ~(uintptr_t)4095u, e = ((uintptr_t)x.data() + x.capacity()) & ~(uintptr_t)4095u;
    if (e > a) (void)madvise((void *)a, (size_t)(e - a), MADV_DONTNEED);
  }
  struct State {
    std::mutex m;
    std::condition_variable cv;
    std::vector<FormattedBatch *> free;
    bool done = false;
  };
  std::shared_ptr<State> st_ = std::make_shared<State>();
  std::vector<std::thread> reapers_;
};

class ReadOutput {
 public:
  ReadOutput(OffsetWriter *out1, OffsetWriter *out2, const std::vector<std::string> &legend) : out1_(out1), out2_(out2), legend_(legend) {}
  bool failed() const { return failed_.load(); }

  // thread-safe; nothing is written
  void format(const ReadBatch &b, FormattedBatch &out) const
  {
    const size_t n = b.seq1.size();
    out.n = 0;
    shk::RecordFetcher f1, f2;
    // segments: [first, last) read ranges that do not cross a 50 000 boundary
    for (size_t first = 0; first < n;) {
      const uint64_t g = b.first_read + first;
      const size_t last = (size_t)std::min<uint64_t>(n, first + (50000 - g % 50000));
      if (out.n == out.segs.size()) out.segs.emplace_back();
      FormattedSegment &sg = out.segs[out.n++];
      sg.reset();
      sg.carries = g % 50000 != 0;                  // continues the previous batch's chunk: previd is known only when the batches are written
      std::string previd;
      bool reserved = false;
      if (b.lean) {
        // many associated reads in this segment: its byte range in one read; few: one read per record
        const size_t n_assoc = b.gene_off[last] - b.gene_off[first];
        if (n_assoc * 10 > last - first) {
          f1.load_dense(b.part1, first, last);
          if (out2_) f2.load_dense(b.part2, first, last);
        } else {
          f1.unload();
          f2.unload();
        }
      }
      for (size_t i = first; i < last; ++i) {
        if (b.gene_off[i] == b.gene_off[i + 1]) continue;
        shk::RecordFetcher::View v1{nullptr, 0, nullptr, 0, nullptr}, v2{nullptr, 0, nullptr, 0, nullptr};
        const char *id;
        size_t id_len;
        if (b.lean) {
          if (!f1.get(b.part1, i, v1) || (out2_ && !f2.get(b.part2, i, v2))) { failed_ = true; continue; }
          id = v1.id;
          id_len = v1.id_len;
        } else {
          id = b.id1.at(i);
          id_len = b.id1.len(i);
        }
        for (uint32_t j = b.gene_off[i]; j < b.gene_off[i + 1]; ++j) {
          const std::string &gene = legend_[b.gene_ids[j]];
          sg.ssv.append(id, id_len);
          sg.ssv.push_back(' ');
          sg.ssv.append(gene);
          sg.ssv.push_back('\n');
          const bool head = !sg.has_assoc;            // the segment's first association
          const bool same = !(head && sg.carries) && previd.size() == id_len && memcmp(previd.data(), id, id_len) == 0;
          if (!same) {
            std::string &o1 = (head && sg.carries) ? sg.head1 : sg.fq1, &o2 = (head && sg.carries) ? sg.head2 : sg.fq2;
            if (b.lean) {
              if (out1_) record_view(o1, v1);
              if (out2_) record_view(o2, v2);
            } else {
              if (out1_) record(o1, b.id1, b.seq1, b.qual1, b.qual_as_read1, i);
              if (out2_) record(o2, b.id2, b.seq2, b.qual2, b.qual_as_read2, i);
            }
            // (the segment's text in one allocation: by its first record and the reads it has left -- a string that doubles its way
            //  to 8 MB copies itself twice over and page-faults every step)
            if (!reserved && &o1 == &sg.fq1) {
              reserved = true;
              size_t left = 0;
              for (size_t r = i; r < last; ++r) left += b.gene_off[r] != b.gene_off[r + 1];
              if (out1_) sg.fq1.reserve(sg.fq1.size() * left + (sg.fq1.size() * left >> 4) + 64);
              if (out2_) sg.fq2.reserve(sg.fq2.size() * left + (sg.fq2.size() * left >> 4) + 64);
              sg.ssv.reserve(sg.ssv.size() * left + (sg.ssv.size() * left >> 3) + 64);
            }
          }
          if (head) sg.head_id.assign(id, id_len);
          previd.assign(id, id_len);
          sg.has_assoc = true;
        }
      }
      sg.last_id = previd;
      first = last;
    }
  }

  // in input order, one thread
  void emit(const std::shared_ptr<const FormattedBatch> &fp)
  {
    const FormattedBatch &f = *fp;
    for (size_t si = 0; si < f.n; ++si) {
      const FormattedSegment &sg = f.segs[si];
      fwrite(sg.ssv.data(), 1, sg.ssv.size(), stdout);
      // (ReadOutput.hpp:44-48: a read's FASTQ records are printed unless its name equals the one printed just before it)
      const bool head_repeats = sg.carries && sg.has_assoc && sg.head_id == carry_;
      if (out1_) {
        if (!head_repeats) out1_->append(sg.head1.data(), sg.head1.size(), fp);
        out1_->append(sg.fq1.data(), sg.fq1.size(), fp);
      }
      if (out2_) {
        if (!head_repeats) out2_->append(sg.head2.data(), sg.head2.size(), fp);
        out2_->append(sg.fq2.data(), sg.fq2.size(), fp);
      }
      carry_ = sg.has_assoc ? sg.last_id : (sg.carries ? carry_ : std::string());
    }
  }

 private:
  static void record_view(std::string &f, const shk::RecordFetcher::View &v)
  {
    f.push_back('@');
    f.append(v.id, v.id_len);
    f.push_back('\n');
    f.append(v.seq, v.seq_len);
    f.append("\n+\n");
    f.append(v.qual, v.seq_len);
    f.push_back('\n');
  }
  static void record(std::string &f, const Strings &id, const DevStrings &seq, const DevStrings &qual, const std::map<size_t, std::string> &qual_as_read, size_t i)
  {
    f.push_back('@');
    if (i < id.size()) f.append(id.at(i), id.len(i));
    f.push_back('\n');
    if (i < seq.size()) f.append(seq.at(i), seq.len(i));
    f.append("\n+\n");
    const auto odd = qual_as_read.empty() ? qual_as_read.end() : qual_as_read.find(i);
    if (odd != qual_as_read.end()) f.append(odd->second);
    else if (i < qual.size()) f.append(qual.at(i), qual.len(i));
    f.push_back('\n');
  }
  OffsetWriter *out1_, *out2_;
  const std::vector<std::string> &legend_;
  mutable std::atomic<bool> failed_{false};   // a record could not be read back from its file (I/O error)
  std::string carry_;   // previd at the end of the previous batch (only used when a batch starts mid-chunk)
};

}  // namespace

// CPUs this process can actually run threads on: the machine's count, cut down to its affinity mask and to its cgroup's CPU quota
// (cgroup v2 cpu.max, v1 cpu.cfs_quota_us / cpu.cfs_period_us)
static unsigned usable_cpus()
{
  unsigned n = std::max(1u, std::thread::hardware_concurrency());
  cpu_set_t set;
  CPU_ZERO(&set);
  if (sched_getaffinity(0, sizeof(set), &set) == 0) {
    const int c = CPU_COUNT(&set);
    if (c > 0) n = std::min(n, (unsigned)c);
  }
  long quota = -1, period = -1;
  if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char q[64];
    if (fscanf(f, "%63s %ld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atol(q);
    fclose(f);
  } else {
    if (FILE *fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(fq, "%ld", &quota) != 1) quota = -1; fclose(fq); }
    if (FILE *fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(fp, "%ld", &period) != 1) period = -1; fclose(fp); }
  }
  if (quota > 0 && period > 0) n = std::min(n, (unsigned)std::max(1L, (quota + period - 1) / period));
  return std::max(1u, n);
}

int main(int argc, char *argv[])
{
  Options opt_parsed = parse_arguments(argc, argv);
  if (!opt_parsed.batch_given) opt_parsed.batch = auto_batch(opt_parsed.sample1_path, opt_parsed.batch);
  const Options opt = opt_parsed;
  if (opt.verbose) timeline.on();
  timeline("arguments parsed");

  if (opt.verbose) {
    std::cerr << "shark (MI355X): reference " << opt.fasta_path << ", sample " << opt.sample1_path;
    if (opt.paired_flag) std::cerr << " + " << opt.sample2_path;
    std::cerr << "; k=" << opt.k << " c=" << opt.c << " q=" << opt.min_quality << (opt.single ? " single" : "") << " bf=" << (opt.bf_size >> 33)
              << "GB gpus=" << opt.gpus << " devices=";
    for (size_t g = 0; g < opt.devices.size(); ++g) std::cerr << (g ? "," : "") << opt.devices[g];
    std::cerr << "\n" << std::endl;
  }

  // the reference opens its inputs unchecked (main.cpp:88-106) and then reads nothing from a file that is not there; here a
  // sample that cannot be opened is reported before any work is done
  for (const std::string *path : {&opt.sample1_path, &opt.sample2_path}) {
    if (path->empty()) continue;
    // (a pipe is not opened for the check: it can be opened once.  access() says whether it may be read.)
    struct stat st;
    if (stat(path->c_str(), &st) == 0 && !S_ISREG(st.st_mode) && !S_ISDIR(st.st_mode)) {
      if (access(path->c_str(), R_OK) == 0) continue;
      std::cerr << "shark: cannot open the sample " << *path << std::endl;
      return EXIT_FAILURE;
    }
    FILE *f = fopen(path->c_str(), "rb");
    if (!f) {
      std::cerr << "shark: cannot open the sample " << *path << std::endl;
      return EXIT_FAILURE;
    }
    fclose(f);
  }

  // ---- contexts: one per GPU, index replicated by deterministic rebuild.  Creating the first context initialises the HIP runtime
  // (a few hundred milliseconds); that happens on a thread of its own while this one reads the reference and the readers
  // (which need no GPU) already parse the sample -------
  const int n_gpus = opt.gpus;
  std::vector<shk_ctx *> ctxs((size_t)n_gpus, nullptr);
  int ctx_rc = SHK_OK, ctx_bad = -1;
  std::mutex ctx_m;
  std::condition_variable ctx_cv;
  bool ctx_done = false, ctx_created = false;
  BatchPool &pool = *new BatchPool;   // (never destroyed: the process leaves through _exit)
  // what the ring of page-locked batches has to hold; known once the sample is partitioned (this thread), used by the context
  // thread, which builds the ring as soon as the HIP runtime is up
  struct RingPlan { bool known = false; size_t limit = 0, bytes = 0, reads = 0; bool paired = false, with_qual = false; } ring_plan;
  // The ring lives in ordinary memory by default: its buffers are handed round, so the runtime's registration of them is paid once
  // per buffer (measured: the same rate as page-locked buffers at what the readers deliver), it exists before the HIP runtime does --
  // the readers fill it while the runtime initialises -- and it costs nothing to set up.  SHARK_PINNED=1: page-locked.
  const bool pin_ring = getenv("SHARK_PINNED") && getenv("SHARK_PINNED")[0] == '1';
  std::thread ctx_thread([&] {
    // N workers start like N workers (main.cpp:219-223 starts the reference's N threads at once): every context is created on a
    // thread of its own -- the runtime comes up once, whoever gets there first; streams, the filter's allocation and its clearing,
    // the slots' buffers then proceed side by side (serially, two contexts took 2.3 x one context's time, eight would have taken
    // longer than a 16 M-pair sample on one GPU)
    std::vector<int> rcs((size_t)n_gpus, SHK_OK);
    auto create = [&](const int g) {
      shk_params p{};
      p.k = opt.k; p.c = opt.c; p.bf_bits = opt.bf_size; p.min_quality = opt.min_quality; p.single = opt.single;
      p.device = opt.devices[(size_t)g];   // worker g's device (--devices)
      rcs[(size_t)g] = shk_create(&p, &ctxs[(size_t)g]);
    };
    {
      std::vector<std::thread> th;
      for (int g = 1; g < n_gpus; ++g) th.emplace_back(create, g);
      create(0);
      for (auto &t : th) t.join();
    }
    for (int g = n_gpus - 1; g >= 0; --g)
      if (rcs[(size_t)g] != SHK_OK) { ctx_rc = rcs[(size_t)g]; ctx_bad = g; }
    timeline("contexts created");
    std::unique_lock<std::mutex> l(ctx_m);
    ctx_created = true;
    ctx_cv.notify_all();
    if (pin_ring) {
      // SHARK_PINNED=1: the ring in page-locked memory, which only exists once the HIP runtime is up
      ctx_cv.wait(l, [&] { return ring_plan.known; });
      const RingPlan plan = ring_plan;
      l.unlock();
      if (ctx_rc == SHK_OK) g_pin_batches = true;
      pool.make_ring(plan.limit, plan.bytes, plan.reads, plan.paired, plan.with_qual);
      timeline("batch ring ready");
      l.lock();
      ctx_done = true;
      ctx_cv.notify_all();
    }
  });
  struct CtxJoin {   // (every early return below has to wait for that thread)
    std::thread &t;
    ~CtxJoin() { if (t.joinable()) t.join(); }
  } ctx_join{ctx_thread};
  std::vector<std::string> legend_ID;   // gene names in file order (FastaSplitter.hpp:48); filled below, read by the output stage
  // ---- 3. sample ---------------------------------------------------------------
  // Three roles, as in main.cpp:66-77 (split / analyze / output), decoupled by queues:
  //   readers   plain four-line FASTQ: a parallel newline count gives the byte range of every batch of each mate file
  //             (fastq_partition.hpp), and `readers` threads parse whole batches independently into pinned
  //             structure-of-arrays batches -- the feed scales with host cores instead of one splitter mutex
  //             (FastqSplitter.hpp:48).  gzip'd or irregular input: the serial kseq-rule reader, from the first batch
  //             that is not strict on.
  //   analyzers one thread per GPU, batch i -> GPU i mod N, SHK_PIPE_DEPTH batches in flight per GPU
  //   output    this thread, batches in input order (ReadOutput.hpp:37-50)
  {
    // readers / parsers / formatters: sixteen per worker (what one GPU's feed was measured to use), as far as this process has cores
    // to run them on -- its affinity mask and its cgroup's CPU quota count, not the machine's (a one-GPU share of a host is 16 cores
    // whatever hardware_concurrency() says: more threads than that only take turns)
    const unsigned hw = usable_cpus();
    unsigned io_threads = opt.nThreads > 1 ? (unsigned)opt.nThreads : std::max(1u, std::min(16u * (unsigned)n_gpus, hw));
    // (the reference opens its outputs unchecked and writes nothing to a file it could not open, main.cpp:99-106: same here)
    TextPool text_pool;               // (in front of the writers: they hand the last texts back while they close)
    OffsetWriter w1, w2;
    // one writer thread per output file: tmpfs takes 8.7 GB/s from ONE thread writing a file and 3.6-4.6 GB/s from 2-12 threads
    // writing disjoint parts of it (tools/tmpfs_write_bench.cpp); the command with half the sample written out again, 32 M pairs,
    // the same files: 2.74 / 2.80 s with one helper per file, 2.93-3.38 s with three, 3.65 s with two
    unsigned write_helpers = 1;
    if (const char *e = getenv("SHARK_WRITE_HELPERS")) write_helpers = (unsigned)std::max(1, atoi(e));
    w1.open(opt.out1_path, write_helpers);
    if (opt.paired_flag && opt.out2_path != "") w2.open(opt.out2_path, write_helpers);
    OffsetWriter *out1 = w1.is_open() ? &w1 : nullptr, *out2 = w2.is_open() ? &w2 : nullptr;
    ReadOutput ro(out1, out2, legend_ID);
    setvbuf(stdout, nullptr, _IOFBF, 1 << 22);

    // per-GPU input queues; batch i goes to GPU i mod N
    std::vector<std::unique_ptr<BoundedQueue<std::unique_ptr<ReadBatch>>>> todo;
    for (int g = 0; g < n_gpus; ++g) todo.emplace_back(new BoundedQueue<std::unique_ptr<ReadBatch>>(1u << 20));   // (the readers' window bounds what is in flight)
    std::mutex done_m;
    std::condition_variable done_cv;
    std::map<uint64_t, std::unique_ptr<ReadBatch>> done;
    uint64_t n_batches = 0;          // batches handed to the GPUs so far (guarded by done_m)
    bool split_finished = false;
    auto dispatch = [&](std::unique_ptr<ReadBatch> b) {
      {
        std::lock_guard<std::mutex> l(done_m);
        n_batches = std::max(n_batches, b->index + 1);
      }
      todo[(size_t)(b->index % (uint64_t)n_gpus)]->push(std::move(b));
    };

    // ---- the parallel feed -------------------------------------------------------
    shk::BatchTable tab1, tab2;
    // (the parallel feeds look into the sample files, read them twice and seek in them: regular files only -- a pipe goes to the
    //  serial reader, which opens it once and keeps names and qualities in memory)
    const bool samples_are_files = shk::regular_file(opt.sample1_path) && (!opt.paired_flag || shk::regular_file(opt.sample2_path));
    bool parallel_feed = samples_are_files && !getenv("SHARK_SERIAL_READER") && !getenv("SHARK_SINGLE_SPLITTER");
    // compressed samples (gzip magic in both mate files): inflated in parallel, cut and parsed from memory (GzCutter above);
    // SHARK_GZ_SERIAL_PARSE=1: the serial kseq-rule reader behind the inflaters, as in round 3 (the tests run both)
    auto is_gzip = [](const std::string &path) {
      unsigned char h[2] = {0, 0};
      FILE *f = fopen(path.c_str(), "rb");
      const bool gz = f && fread(h, 1, 2, f) == 2 && h[0] == 0x1f && h[1] == 0x8b;
      if (f) fclose(f);
      return gz;
    };
    const bool gz_feed = parallel_feed && !getenv("SHARK_GZ_SERIAL_PARSE") && is_gzip(opt.sample1_path) && (!opt.paired_flag || is_gzip(opt.sample2_path));
    if (gz_feed) parallel_feed = false;
    const unsigned gz_inflaters = std::max(2u, io_threads / (opt.paired_flag ? 2u : 1u));
    const unsigned n_gz_parsers = gz_feed ? std::max(2u, io_threads / 3u) : 0u;
    std::unique_ptr<GzCutter> cut1, cut2;
    if (gz_feed) {
      cut1.reset(new GzCutter(opt.sample1_path, gz_inflaters, opt.batch));
      if (opt.paired_flag) cut2.reset(new GzCutter(opt.sample2_path, gz_inflaters, opt.batch));
      cut1->start();
      if (cut2) cut2->start();
    }
    uint64_t n_par_records = 0;       // records both mate files certainly have: the pair stream of the strict part
    bool fixed_width = false;
    if (parallel_feed) {
      // fixed-width records: batch offsets are arithmetic; otherwise a parallel newline count of both files
      uint64_t rl1 = 0, rl2 = 0;
      fixed_width = shk::fixed_record_file(opt.sample1_path, tab1, rl1) && (!opt.paired_flag || shk::fixed_record_file(opt.sample2_path, tab2, rl2));
      if (fixed_width) {
        n_par_records = opt.paired_flag ? std::min(tab1.n_records, tab2.n_records) : tab1.n_records;
        shk::fixed_record_batches(tab1, rl1, opt.batch, n_par_records);
        if (opt.paired_flag) shk::fixed_record_batches(tab2, rl2, opt.batch, n_par_records);
      } else {
        if (tab1.fd >= 0) { ::close(tab1.fd); tab1.fd = -1; }
        if (tab2.fd >= 0) { ::close(tab2.fd); tab2.fd = -1; }
        std::vector<uint64_t> cnt1, cnt2;
        shk::count_file(opt.sample1_path, io_threads, tab1, cnt1);
        parallel_feed = tab1.ok;
        if (parallel_feed && opt.paired_flag) {
          shk::count_file(opt.sample2_path, io_threads, tab2, cnt2);
          parallel_feed = tab2.ok;
        }
        if (parallel_feed) {
          n_par_records = opt.paired_flag ? std::min(tab1.n_records, tab2.n_records) : tab1.n_records;
          shk::locate_batches(tab1, cnt1, opt.batch, n_par_records, io_threads);
          if (opt.paired_flag) shk::locate_batches(tab2, cnt2, opt.batch, n_par_records, io_threads);
          parallel_feed = tab1.ok && (!opt.paired_flag || tab2.ok);
          if (!parallel_feed) n_par_records = 0;
        }
      }
    }
    timeline("sample partitioned");
    const uint64_t n_par_batches = (n_par_records + opt.batch - 1) / opt.batch;
    std::atomic<uint64_t> next_batch{0};
    std::atomic<uint64_t> irregular_at{UINT64_MAX};     // first batch a reader found not to be strict four-line FASTQ
    const unsigned n_readers = parallel_feed ? (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(io_threads, n_par_batches)) : 0;
    std::vector<double> t_reader(n_readers, 0.0);
    // a reader must not run ahead of the drain without bound: at most `window` batches beyond the one being written
    // the ring: a batch per reader, what the GPUs hold in flight, and a few being turned into text or waiting for their turn to be written
    // (never more batches than the sample has, plus what the serial reader and the GPU pipelines need: --batch may be large)
    const uint64_t window = gz_feed ? (uint64_t)n_gz_parsers + (uint64_t)n_gpus * (SHK_PIPE_DEPTH + 2) + 6
                                    : std::min<uint64_t>((uint64_t)n_readers + (uint64_t)n_gpus * (SHK_PIPE_DEPTH + 2) + 6,
                                                         n_par_batches + (uint64_t)n_gpus * (SHK_PIPE_DEPTH + 2) + 2);
    {
      uint64_t widest = 0;
      for (uint64_t i = 0; i < n_par_batches; ++i) {
        widest = std::max(widest, tab1.off[i + 1] - tab1.off[i]);
        if (opt.paired_flag) widest = std::max(widest, tab2.off[i + 1] - tab2.off[i]);
      }
      std::lock_guard<std::mutex> l(ctx_m);
      ring_plan.known = true;
      ring_plan.limit = (size_t)window;
      ring_plan.bytes = n_par_batches ? (size_t)(widest / 2 + 64) : (gz_feed ? (size_t)opt.batch * 160 : 0);
      ring_plan.reads = (n_par_batches || gz_feed) ? (size_t)opt.batch : 0;
      ring_plan.paired = opt.paired_flag;
      ring_plan.with_qual = static_cast<char>(opt.min_quality) != 0;
      ctx_cv.notify_all();
    }
    if (!pin_ring) {
      pool.make_ring(ring_plan.limit, ring_plan.bytes, ring_plan.reads, ring_plan.paired, ring_plan.with_qual);
      timeline("batch ring ready");
      std::lock_guard<std::mutex> l(ctx_m);
      ctx_done = true;
      ctx_cv.notify_all();
    }
    uint64_t drained = 0;                                // guarded by done_m
    // Batch j may only leave a reader once every batch before it is KNOWN to be strict four-line FASTQ: an irregular batch
    // i < j that keeps the four-line alignment (an empty read, a sequence/quality length mismatch, a lone CR, a NUL) lets
    // batch j validate, yet everything from i on belongs to the serial reader -- which numbers its batches from i again
    // and may cut the records differently.  `validated` = number of leading batches known to be strict (guarded by done_m).
    uint64_t validated = 0;
    const bool need_qual = static_cast<char>(opt.min_quality) != 0;   // the device reads qualities only with -q (the reference's char, argument_parser.hpp:144)
    shk::RecordLayout lay1, lay2;                                     // the layout of each file's first record: the readers' fast check
    if (parallel_feed) {
      std::vector<char> head(1u << 16);
      for (int m = 0; m < (opt.paired_flag ? 2 : 1); ++m) {
        shk::BatchTable &t = m ? tab2 : tab1;
        const size_t got = (size_t)std::min<uint64_t>(head.size(), t.file_size);
        if (got && shk::pread_all(t.fd, head.data(), 0, got)) shk::layout_of(head.data(), got, m ? lay2 : lay1);
      }
    }
    std::vector<std::thread> readers;
    for (unsigned r = 0; r < n_readers; ++r) {
      readers.emplace_back([&, r] {
        shk::LeanScratch sc;
        {
          std::unique_lock<std::mutex> l(ctx_m);   // the ring of batches exists (at once; page-locked: once the HIP runtime is up)
          ctx_cv.wait(l, [&] { return ctx_done; });
        }
        for (;;) {
          const uint64_t i = next_batch.fetch_add(1);
          if (i >= n_par_batches || i >= irregular_at.load()) break;
          {
            std::unique_lock<std::mutex> l(done_m);
            done_cv.wait(l, [&] { return i < drained + window || i >= irregular_at.load(); });
          }
          if (i >= irregular_at.load()) break;
          auto t0 = std::chrono::steady_clock::now();
          const size_t want = (size_t)std::min<uint64_t>(opt.batch, n_par_records - i * opt.batch);
          std::unique_ptr<ReadBatch> b = pool.acquire();
          if (!b) break;
          b->index = i;
          b->first_read = i * opt.batch;
          b->lean = true;
          size_t ok1 = shk::lean_parse_range(tab1.fd, tab1.off[i], tab1.off[i + 1], want, lay1, need_qual, sc, b->seq1.bytes, b->seq1.off, b->qual1.bytes, b->part1);
          size_t ok2 = want;
          if (opt.paired_flag && ok1 == want)
            ok2 = shk::lean_parse_range(tab2.fd, tab2.off[i], tab2.off[i + 1], want, lay2, need_qual, sc, b->seq2.bytes, b->seq2.off, b->qual2.bytes, b->part2);
          if (ok1 < want || ok2 < want) {
            // not strict here: this batch and everything behind it belongs to the serial reader
            pool.release(std::move(b));
            uint64_t cur = irregular_at.load();
            while (i < cur && !irregular_at.compare_exchange_weak(cur, i)) {}
            done_cv.notify_all();
            break;
          }
          t_reader[r] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
          {
            // (batch indices are handed out in increasing order, so the reader of the smallest outstanding one never waits here)
            std::unique_lock<std::mutex> l(done_m);
            done_cv.wait(l, [&] { return validated == i || i >= irregular_at.load(); });
            if (i >= irregular_at.load()) { l.unlock(); pool.release(std::move(b)); break; }
            validated = i + 1;
          }
          done_cv.notify_all();
          dispatch(std::move(b));
        }
      });
    }

    // ---- compressed samples: pieces of the two mates joined by index, parsed from memory by n_gz_parsers threads ----
    struct GzJob { uint64_t index; std::unique_ptr<GzPiece> p1, p2; size_t want; };
    BoundedQueue<std::unique_ptr<GzJob>> gz_jobs(2);
    std::atomic<uint64_t> gz_batches{0};          // batches the joiner handed out
    std::thread gz_joiner;
    if (gz_feed) {
      gz_joiner = std::thread([&] {
        for (uint64_t i = 0;; ++i) {
          std::unique_ptr<GzPiece> a = cut1->next(), b2;
          if (cut2) b2 = cut2->next();
          if (!a || (cut2 && !b2)) break;
          // the pair stream ends with the shorter mate file (FastqSplitter.hpp:60)
          const size_t want = cut2 ? std::min(a->records, b2->records) : a->records;
          const bool ends = a->last || (b2 && b2->last) || want < opt.batch;
          if (want) {
            std::unique_ptr<GzJob> j(new GzJob{i, std::move(a), std::move(b2), want});
            gz_batches = i + 1;
            gz_jobs.push(std::move(j));
          }
          if (ends || i >= irregular_at.load()) break;
        }
        gz_jobs.close();
        // (whatever the cutters still hold is not part of the pair stream -- or belongs to the serial reader)
        cut1->stop();
        if (cut2) cut2->stop();
      });
      for (unsigned r = 0; r < n_gz_parsers; ++r) {
        readers.emplace_back([&] {
          {
            std::unique_lock<std::mutex> l(ctx_m);
            ctx_cv.wait(l, [&] { return ctx_done; });
          }
          shk::RecordLayout gl1, gl2;
          std::unique_ptr<GzJob> j;
          while (gz_jobs.pop(j)) {
            const uint64_t i = j->index;
            if (i >= irregular_at.load()) continue;
            {
              std::unique_lock<std::mutex> l(done_m);
              done_cv.wait(l, [&] { return i < drained + window || i >= irregular_at.load(); });
            }
            if (i >= irregular_at.load()) continue;
            std::unique_ptr<ReadBatch> b = pool.acquire();
            if (!b) break;
            b->index = i;
            b->first_read = i * opt.batch;
            b->lean = true;
            // the text moves into the batch (its old buffer goes back to the cutter); a piece with more records than the pair
            // stream takes is cut behind the want-th record
            auto take = [&](GzPiece &p, std::vector<char, default_init_allocator<char>> &text, shk::RecordLayout &lay) -> size_t {
              text.swap(p.text);
              size_t len = text.size();
              if (p.records > j->want) {
                uint64_t found = 0;
                len = newlines_until(text.data(), text.size(), 4 * (uint64_t)j->want, found);
              }
              if (!lay.usable()) shk::layout_of(text.data(), len, lay);
              return len;
            };
            const size_t len1 = take(*j->p1, b->text1, gl1);
            size_t ok1 = shk::lean_parse_mem(b->text1.data(), len1, j->want, gl1, need_qual, b->seq1.bytes, b->seq1.off, b->qual1.bytes, b->part1);
            size_t ok2 = j->want;
            if (j->p2 && ok1 == j->want) {
              const size_t len2 = take(*j->p2, b->text2, gl2);
              ok2 = shk::lean_parse_mem(b->text2.data(), len2, j->want, gl2, need_qual, b->seq2.bytes, b->seq2.off, b->qual2.bytes, b->part2);
            }
            cut1->recycle(std::move(j->p1));
            if (j->p2) cut2->recycle(std::move(j->p2));
            if (ok1 < j->want || ok2 < j->want) {
              // not strict four-line FASTQ here: this batch and everything behind it belongs to the serial reader
              pool.release(std::move(b));
              uint64_t cur = irregular_at.load();
              while (i < cur && !irregular_at.compare_exchange_weak(cur, i)) {}
              done_cv.notify_all();
              continue;
            }
            {
              std::unique_lock<std::mutex> l(done_m);
              done_cv.wait(l, [&] { return validated == i || i >= irregular_at.load(); });
              if (i >= irregular_at.load()) { l.unlock(); pool.release(std::move(b)); continue; }
              validated = i + 1;
            }
            done_cv.notify_all();
            dispatch(std::move(b));
          }
        });
      }
    }

    // ---- 1+2. reference: legend in file order (FastaSplitter.hpp:48) + index -- while the readers above already parse the sample ----
    auto stop_feed = [&] {
      // a failure before the analyzers exist: end the readers (they stop at an irregular batch 0) and take their batches back
      uint64_t cur = irregular_at.load();
      while (0 < cur && !irregular_at.compare_exchange_weak(cur, 0)) {}
      for (auto &q : todo) q->close();
      pool.shutdown();
      done_cv.notify_all();
      std::thread drain_q([&] {
        for (auto &q : todo) {
          std::unique_ptr<ReadBatch> b;
          while (q->pop(b)) {}
        }
      });
      if (gz_feed) {
        // (the parsers drop every job once batch 0 counts as irregular; the joiner notices the same and stops the cutters)
        std::thread drop_jobs([&] { std::unique_ptr<GzJob> j; while (gz_jobs.pop(j)) {} });
        if (gz_joiner.joinable()) gz_joiner.join();
        drop_jobs.join();
      }
      for (auto &t : readers) t.join();
      drain_q.join();
    };
    {
      shk::FastxReader fa(opt.fasta_path);
      if (!fa.ok()) {
        stop_feed();
        std::cerr << "shark: cannot open " << opt.fasta_path << std::endl;
        return EXIT_FAILURE;
      }
      {
        std::unique_lock<std::mutex> l(ctx_m);   // (the contexts; that thread goes on to build the batch ring)
        ctx_cv.wait(l, [&] { return ctx_created; });
      }
      if (ctx_rc != SHK_OK) {
        stop_feed();
        std::cerr << "shark: cannot create a context on GPU " << opt.devices[(size_t)ctx_bad] << ": " << shk_strerror(ctx_rc) << std::endl;
        return EXIT_FAILURE;
      }
      shk::FastxRecord rec;
      while (fa.read(rec) >= 0) {
        legend_ID.push_back(rec.name.c_str());
        const size_t len = strnlen(rec.seq.data(), rec.seq.size());  // C-string semantics (main.cpp:164)
        for (auto *ctx : ctxs) {
          const int rc = shk_ref_add(ctx, rec.seq.data(), len);
          if (rc != SHK_OK) {
            stop_feed();
            std::cerr << "shark: " << shk_strerror(rc) << std::endl;
            return EXIT_FAILURE;
          }
        }
      }
    }
    pelapsed("Transcript file processed");
    timeline("reference read");
    {
      std::vector<std::thread> th;
      std::vector<int> rcs((size_t)n_gpus, 0);
      for (int g = 0; g < n_gpus; ++g) th.emplace_back([&, g] { rcs[(size_t)g] = shk_ref_finalize(ctxs[(size_t)g]); });
      for (auto &t : th) t.join();
      for (int g = 0; g < n_gpus; ++g)
        if (rcs[(size_t)g] != SHK_OK) {
          stop_feed();
          std::cerr << "shark: index build failed on GPU " << g << ": " << shk_strerror(rcs[(size_t)g]) << " " << shk_last_error(ctxs[(size_t)g]) << std::endl;
          return EXIT_FAILURE;
        }
    }
    timeline("index built");
    pelapsed("First switch performed");
    {
      shk_index_info info{};
      shk_index_info_get(ctxs[0], &info);
      pelapsed("BF created from transcripts (" + std::to_string(info.nidx) + " genes)");
    }
    pelapsed("Second switch performed");

    // ---- the serial feed: everything the parallel readers did not (or could not) deliver ---------
    std::unique_ptr<BatchSplitter> fs;
    bool serial_needed = false, serial_failed = false;
    std::thread splitter([&] {
      if (gz_feed && gz_joiner.joinable()) gz_joiner.join();
      for (auto &t : readers) t.join();
      timeline("parallel readers done");
      if (gz_feed) {
        // compressed sample: the parsers delivered batches [0, stop); an irregular record sends the rest through the serial reader,
        // which reads over the records already delivered (they are strict: the kseq reader's records are the same)
        const uint64_t stop = std::min<uint64_t>(irregular_at.load(), gz_batches.load());
        serial_needed = irregular_at.load() != UINT64_MAX;
        bool serial_ok = true;
        if (serial_needed) {
          fs.reset(new BatchSplitter(opt, io_threads, pool));
          serial_ok = fs->ok();
          if (serial_ok) {
            fs->skip_records(stop * opt.batch, stop);
            for (;;) {
              auto b = (*fs)();
              if (!b) break;
              dispatch(std::move(b));
            }
          }
        }
        serial_failed = !serial_ok;
        for (auto &q : todo) q->close();
        timeline("serial reader done");
        std::lock_guard<std::mutex> l(done_m);
        split_finished = true;
        done_cv.notify_all();
        return;
      }
      const uint64_t stop = std::min<uint64_t>(irregular_at.load(), n_par_batches);   // batches [0, stop) came from the readers
      // nothing left when the readers delivered every batch and a mate file ends exactly there (the pair stream ends with the
      // shorter file, FastqSplitter.hpp:60)
      serial_needed = !parallel_feed || stop < n_par_batches || !(tab1.off[stop] >= tab1.file_size || (opt.paired_flag && tab2.off[stop] >= tab2.file_size));
      bool serial_ok = true;
      if (serial_needed) {
        fs.reset(new BatchSplitter(opt, io_threads, pool));
        serial_ok = fs->ok();
      }
      serial_failed = !serial_ok;
      if (serial_needed && serial_ok && parallel_feed) {
        const uint64_t o1 = tab1.off[stop];
        const uint64_t o2 = opt.paired_flag ? tab2.off[stop] : 0;
        fs->resume_serial(o1, o2, stop, std::min<uint64_t>(stop * opt.batch, n_par_records));
      }
      if (serial_needed && serial_ok) {
        for (;;) {
          auto b = (*fs)();
          if (!b) break;
          dispatch(std::move(b));
        }
      }
      for (auto &q : todo) q->close();
      timeline("serial reader done");
      std::lock_guard<std::mutex> l(done_m);
      split_finished = true;
      done_cv.notify_all();
    });

    std::vector<double> t_gpu((size_t)n_gpus, 0.0);
    // classified batches are turned into text by `formatters` threads, in any order (ReadOutput::format: names and qualities of
    // the associated reads are fetched from the sample files there); the drain below writes the text in input order
    BoundedQueue<std::unique_ptr<ReadBatch>> to_format(1u << 20);
    auto hand_over = [&](std::unique_ptr<ReadBatch> b) {
      std::lock_guard<std::mutex> l(done_m);
      done[b->index] = std::move(b);
      done_cv.notify_all();
    };
    std::vector<std::thread> formatters;
    // half as many formatters as readers: with half the sample written out again the two writer threads are what the command waits for,
    // and they get their cores only if the others leave some (-t 12 on a 16-core share, 16 M / 64 M pairs at 0.50 on-target: 0.84-0.85 /
    // 2.25 s with twelve formatters, 0.75 / 2.09 s with six, 0.81 / 2.07 s with four, 1.12 s with three -- then THEY are the wait)
    unsigned n_formatters = std::max(2u, (io_threads + 1) / 2);
    if (const char *e = getenv("SHARK_FORMATTERS")) n_formatters = (unsigned)std::max(1, atoi(e));      // (A/B timing)
    for (unsigned f = 0; f < n_formatters; ++f) {
      formatters.emplace_back([&] {
        std::unique_ptr<ReadBatch> b;
        while (to_format.pop(b)) {
          if (b->rc == SHK_OK) {
            std::shared_ptr<FormattedBatch> text = text_pool.get();
            ro.format(*b, *text);
            b->text = text;
          }
          hand_over(std::move(b));
        }
      });
    }
    std::vector<std::thread> analyzers;
    for (int g = 0; g < n_gpus; ++g) {
      analyzers.emplace_back([&, g] {
        ReadAnalyzer ra(ctxs[(size_t)g], need_qual);
        bool open = true;
        while (open || ra.in_flight()) {
          // keep the pipeline full; block for input only when nothing is in flight
          while (open && ra.in_flight() < SHK_PIPE_DEPTH) {
            std::unique_ptr<ReadBatch> b;
            int got;
            if (ra.in_flight() == 0) got = todo[(size_t)g]->pop(b) ? 1 : -1;
            else got = todo[(size_t)g]->try_pop(b);
            if (got < 0) { open = false; break; }
            if (got == 0) break;
            auto t0 = std::chrono::steady_clock::now();
            if (!ra.submit(std::move(b))) to_format.push(ra.take_failed());
            t_gpu[(size_t)g] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
          }
          if (ra.in_flight()) {
            auto t0 = std::chrono::steady_clock::now();
            std::unique_ptr<ReadBatch> b = ra.wait();
            t_gpu[(size_t)g] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            to_format.push(std::move(b));
          }
        }
      });
    }
    // ordered drain
    uint64_t next = 0;
    int failed = 0;
    double t_out = 0;
    for (;;) {
      std::unique_ptr<ReadBatch> b;
      {
        std::unique_lock<std::mutex> l(done_m);
        done_cv.wait(l, [&] { return done.count(next) || (split_finished && next >= n_batches); });
        if (!done.count(next)) break;
        b = std::move(done[next]);
        done.erase(next);
      }
      if (b->rc != SHK_OK) {
        failed = b->rc;
      } else {
        auto t0 = std::chrono::steady_clock::now();
        ro.emit(b->text);
        t_out += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      }
      pool.release(std::move(b));
      ++next;
      {
        std::lock_guard<std::mutex> l(done_m);
        drained = next;
      }
      done_cv.notify_all();
    }
    timeline("drain done");
    splitter.join();
    for (auto &t : analyzers) t.join();
    to_format.close();
    for (auto &t : formatters) t.join();
    text_pool.retire(std::min(8u, io_threads));
    fflush(stdout);
    timeline("pipeline threads joined");
    if (serial_failed) {
      std::cerr << "shark: cannot open the sample" << std::endl;
      return EXIT_FAILURE;
    }
    if (ro.failed()) {
      std::cerr << "shark: cannot read the sample again for the output" << std::endl;
      return EXIT_FAILURE;
    }
    if (shk::parallel_gunzip_out_of_memory().load()) {
      std::cerr << "shark: out of memory while inflating the sample" << std::endl;
      return EXIT_FAILURE;
    }
    if (opt.verbose) {
      double tr = 0;
      for (double x : t_reader) tr += x;
      std::cerr << "[shark/io] threads " << io_threads << ", parallel readers " << n_readers << (fixed_width ? " fixed-width records (" : " (") << std::min<uint64_t>(irregular_at.load(), n_par_batches)
                << " batches, " << tr << " thread-seconds), serial reader: " << (fs ? "index " + std::to_string(fs->t_index) + " s (" + fs->stage_report() + "), fill " + std::to_string(fs->t_fill) + " s, serial " + std::to_string(fs->t_serial) + " s" : std::string("not needed"))
                << "; classify(gpu0) " << t_gpu[0] << " s, output " << t_out << " s" << std::endl;
      // per GPU: seconds its analyzer thread spent inside shk_classify_submit / _wait (the host-fed multi-GPU leg of bench.py reads this)
      std::cerr << "[shark/gpu-busy]";
      for (int g = 0; g < n_gpus; ++g) std::cerr << " " << t_gpu[(size_t)g];
      std::cerr << std::endl;
    }
    bool written = true;
    if (out1) written = w1.close() && written;
    if (out2) written = w2.close() && written;
    text_pool.finish();
    if (opt.verbose)
      std::cerr << "[shark/writers] " << w1.bytes_written() << " + " << w2.bytes_written() << " bytes, busy " << w1.busy_seconds() << " + " << w2.busy_seconds() << " s" << std::endl;
    if (!written) {
      std::cerr << "shark: cannot write the output FASTQ" << std::endl;
      return EXIT_FAILURE;
    }
    if (failed) {
      std::cerr << "shark: classification failed: " << shk_strerror(failed) << std::endl;
      return EXIT_FAILURE;
    }
    timeline("output files closed");
  }
  timeline("outputs closed");
  pelapsed("Sample completed");

  // per-gene assigned-read counts: the one exchange step of the sharded run (RCCL all-reduce over xGMI)
  if (opt.gene_counts_path != "" || (opt.verbose && n_gpus > 1)) {
    std::vector<uint64_t> totals(legend_ID.size() ? legend_ID.size() : 1, 0);
    const uint32_t ng = (uint32_t)std::min<size_t>(legend_ID.size(), 65536);
    {
      std::vector<int> distinct(opt.devices);
      std::sort(distinct.begin(), distinct.end());
      distinct.erase(std::unique(distinct.begin(), distinct.end()), distinct.end());
      if (distinct.size() != opt.devices.size())
        std::cerr << "shark: " << n_gpus << " workers on " << distinct.size() << " device(s): the per-gene counts of workers that share a device are "
                  << "added on it" << (distinct.size() > 1 ? ", RCCL reduces over the distinct devices" : " (no collective)") << std::endl;
    }
    const int rc = shk_gene_counts_allreduce(ctxs.data(), n_gpus, totals.data(), ng);
    if (rc != SHK_OK) {
      std::cerr << "shark: gene count reduction failed: " << shk_strerror(rc) << " " << shk_last_error(ctxs[0]) << std::endl;
      return EXIT_FAILURE;
    }
    if (opt.gene_counts_path != "") {
      FILE *gc = fopen(opt.gene_counts_path.c_str(), "w");
      if (gc) {
        for (uint32_t g = 0; g < ng; ++g)
          if (totals[g]) fprintf(gc, "%s %llu\n", legend_ID[g].c_str(), (unsigned long long)totals[g]);
        fclose(gc);
      }
    }
    if (opt.verbose) {
      uint64_t sum = 0;
      for (uint32_t g = 0; g < ng; ++g) sum += totals[g];
      std::cerr << "[shark/counts] " << sum << " associations over " << n_gpus << " GPU(s)" << std::endl;
    }
  }

  if (opt.verbose) {
    // (what the process still holds is what its end has to give back: resident and peak resident memory)
    std::ifstream st("/proc/self/status");
    std::string line;
    while (std::getline(st, line))
      if (line.compare(0, 6, "VmRSS:") == 0 || line.compare(0, 6, "VmHWM:") == 0 || line.compare(0, 8, "RssAnon:") == 0 || line.compare(0, 9, "RssShmem:") == 0)
        std::cerr << "[shark/mem] " << line << std::endl;
  }
  if (ctx_thread.joinable()) ctx_thread.join();
  for (auto *ctx : ctxs) shk_destroy(ctx);
  timeline("contexts destroyed");
  pelapsed("Association done");
  // everything is written and closed; leaving through _exit skips the teardown of the HIP runtime and of the worker threads'
  // statics, which costs more than a tenth of a second and changes nothing
  fflush(stdout);
  fflush(stderr);
  // (a profiler writes its results from exit handlers: under one -- ROCP_TOOL_LIBRARIES, or an LD_PRELOAD that names a rocprofiler
  //  library -- or when asked to (SHARK_CLEAN_EXIT=1), leave the ordinary way.  Any other preloaded library, an allocator say, does
  //  not change how the process leaves.  The contexts are destroyed above: exit handlers find no live context.)
  bool clean_exit = false;
  if (const char *v = getenv("ROCP_TOOL_LIBRARIES")) clean_exit = v[0] != 0;
  if (const char *v = getenv("SHARK_CLEAN_EXIT")) clean_exit = clean_exit || (v[0] != 0 && v[0] != '0');
  if (const char *v = getenv("LD_PRELOAD")) clean_exit = clean_exit || strstr(v, "rocprof") != nullptr || strstr(v, "roctracer") != nullptr;
  if (clean_exit) exit(0);
  _exit(0);
}
