// gunzip_tool.cpp -- shark-gunzip: the CLI's parallel gzip reader on its own (tests, timing).
//   shark-gunzip FILE.gz [threads]   -> the uncompressed bytes on stdout ("-n": only count them)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>

#include "gzip_parallel.hpp"

int main(int argc, char **argv)
{
  bool count_only = false;
  int a = 1;
  if (a < argc && !strcmp(argv[a], "-n")) { count_only = true; ++a; }
  if (a >= argc) { fprintf(stderr, "usage: shark-gunzip [-n] FILE.gz [threads]\n"); return 2; }
  const unsigned threads = a + 1 < argc ? (unsigned)atoi(argv[a + 1]) : 8u;
  const auto t0 = std::chrono::steady_clock::now();
  shk::ParallelGunzip z(argv[a], threads);
  if (!z.usable()) { fprintf(stderr, "shark-gunzip: not usable on this file (too small, not gzip, or not text): the CLI falls back to gzread\n"); return 3; }
  const char *d;
  size_t n, total = 0;
  while (z.next(d, n)) {
    total += n;
    if (!count_only && fwrite(d, 1, n, stdout) != n) return 1;
  }
  const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  double cs = 0, c1 = 0, c2 = 0;
  z.cpu_seconds(cs, c1, c2);
  // peak resident set of THIS program (VmHWM: ru_maxrss starts from what the parent had resident when it forked)
  long hwm_kb = 0;
  if (FILE *st = fopen("/proc/self/status", "r")) {
    char line[256];
    while (fgets(line, sizeof line, st))
      if (!strncmp(line, "VmHWM:", 6)) hwm_kb = atol(line + 6);
    fclose(st);
  }
  fprintf(stderr, "shark-gunzip: %zu bytes in %.3f s (%.1f MB/s), %u threads; worker CPU s: search %.3f, pass 1 %.3f, pass 2 %.3f; maxrss_kb %ld\n", total, s,
          total / s / 1e6, threads, cs, c1, c2, hwm_kb);
  if (z.failed()) { fprintf(stderr, "shark-gunzip: out of memory, the stream was not delivered whole\n"); return 1; }
  return 0;
}
