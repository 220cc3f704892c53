// feed_bench -- host-only rate of the CLI's parallel FASTQ feed (count pre-pass + R reader threads that parse whole batches
// into structure-of-arrays buffers, recycled as the CLI recycles its batches).  No GPU.  tools/, not product.
// usage: feed_bench BATCH READERS file_1.fq [file_2.fq]
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#include "../shark_amd/csrc/fastq_partition.hpp"
struct S { std::vector<char, shk::NoInitAlloc<char>> bytes; std::vector<uint64_t> off{0}; };
int main(int argc, char **argv)
{
  if (argc < 4) return 2;
  const uint64_t batch = strtoull(argv[1], nullptr, 10);
  const unsigned R = (unsigned)atoi(argv[2]);
  const bool paired = argc > 4;
  auto t0 = std::chrono::steady_clock::now();
  shk::BatchTable t1, t2;
  std::vector<uint64_t> c1, c2;
  shk::count_file(argv[3], R, t1, c1);
  if (paired) shk::count_file(argv[4], R, t2, c2);
  const uint64_t n = paired ? std::min(t1.n_records, t2.n_records) : t1.n_records;
  shk::locate_batches(t1, c1, batch, n, R);
  if (paired) shk::locate_batches(t2, c2, batch, n, R);
  auto t1c = std::chrono::steady_clock::now();
  const uint64_t nb = (n + batch - 1) / batch;
  std::atomic<uint64_t> next{0}, bases{0};
  std::vector<std::thread> th;
  for (unsigned r = 0; r < R; ++r)
    th.emplace_back([&] {
      shk::ParsedBatch p1, p2;
      S id1, s1, q1, id2, s2, q2;   // one recycled batch per reader
      for (;;) {
        const uint64_t i = next.fetch_add(1);
        if (i >= nb) break;
        const size_t want = (size_t)std::min<uint64_t>(batch, n - i * batch);
        shk::parse_strict_batch(t1.fd, t1.off[i], t1.off[i + 1], want, p1);
        if (paired) shk::parse_strict_batch(t2.fd, t2.off[i], t2.off[i + 1], want, p2);
        shk::fill_soa(p1, want, id1, s1, q1);
        if (paired) shk::fill_soa(p2, want, id2, s2, q2);
        bases += s1.bytes.size() + s2.bytes.size();
      }
    });
  for (auto &t : th) t.join();
  auto t2c = std::chrono::steady_clock::now();
  const double a = std::chrono::duration<double>(t1c - t0).count(), b = std::chrono::duration<double>(t2c - t1c).count();
  const double gb = (double)(t1.file_size + t2.file_size) / 1e9;
  printf("{\"pairs\": %llu, \"batch\": %llu, \"readers\": %u, \"count_s\": %.3f, \"parse_s\": %.3f, \"GBps_parse\": %.2f, \"M_reads_per_s\": %.1f}\n",
         (unsigned long long)n, (unsigned long long)batch, R, a, b, gb / b, (paired ? 2.0 : 1.0) * n / (a + b) / 1e6);
  return 0;
}
