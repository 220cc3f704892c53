// tmpfs_write_bench.cpp -- what /dev/shm takes from T threads writing one file: pwrite at disjoint offsets (mode 0) or memcpy into a shared mapping (mode 1)
//   g++ -O2 -o /tmp/wt tools/tmpfs_write_bench.cpp -pthread && /tmp/wt <mode> <threads> <GiB>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <thread>
#include <vector>
int main(int argc, char **argv)
{
  const int mode = atoi(argv[1]), T = atoi(argv[2]);
  const size_t total = (size_t)atoi(argv[3]) << 30, piece = 8u << 20;
  std::vector<char> src(piece, 'x');
  int fd = open("/dev/shm/wt.bin", O_RDWR | O_CREAT | O_TRUNC, 0666);
  char *map = nullptr;
  if (mode == 1) { if (ftruncate(fd, total)) return 1; map = (char *)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0); }
  auto t0 = std::chrono::steady_clock::now();
  std::vector<std::thread> th;
  const size_t n = total / piece;
  for (int t = 0; t < T; ++t)
    th.emplace_back([&, t] {
      for (size_t i = t; i < n; i += T) {
        if (mode == 0) { if (pwrite(fd, src.data(), piece, i * piece) != (ssize_t)piece) abort(); }
        else memcpy(map + i * piece, src.data(), piece);
      }
    });
  for (auto &x : th) x.join();
  double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  printf("mode %s threads %d: %.2f GB/s\n", mode ? "mmap" : "pwrite", T, total / s / 1e9);
  if (map) munmap(map, total);
  close(fd);
  unlink("/dev/shm/wt.bin");
}
