// test driver for kaiju_amd/csrc/host/pargz.h: inflates a .gz file with N threads to stdout
//   g++ -O2 -std=c++17 -o pargz_test pargz_test.cpp -lz -lpthread ;  pargz_test file.gz [threads] > out
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "../../kaiju_amd/csrc/host/pargz.h"

int main(int argc, char **argv) {
  if (argc < 2) return 2;
  const unsigned threads = argc > 2 ? (unsigned)atoi(argv[2]) : 8;
  int fd = open(argv[1], O_RDONLY);
  struct stat st;
  if (fd < 0 || fstat(fd, &st)) return 2;
  void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
  if (m == MAP_FAILED) return 2;
  pargz::Reader r;
  const auto t0 = std::chrono::steady_clock::now();
  if (!r.open((const uint8_t *)m, (size_t)st.st_size, threads, [](const std::string &msg) { fprintf(stderr, "FATAL %s\n", msg.c_str()); _exit(3); })) {
    fprintf(stderr, "not a gzip file\n");
    return 4;
  }
  std::vector<char> buf(1 << 24);
  size_t total = 0;
  const bool quiet = getenv("PARGZ_NO_OUTPUT") != nullptr;
  for (;;) {
    const size_t n = r.read(buf.data(), buf.size());
    if (!n) break;
    total += n;
    if (!quiet && fwrite(buf.data(), 1, n, stdout) != n) return 5;
  }
  const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  fprintf(stderr, "%zu bytes in %.3f s = %.1f MB/s; pieces entered %llu, absorbed %llu\n", total, s, total / s / 1e6,
          (unsigned long long)r.pieces_entered, (unsigned long long)r.pieces_absorbed);
  fprintf(stderr, "producer: find %.3f s, inflate %.3f s, resolve + crc %.3f s, waiting for the reader %.3f s\n", r.t_find, r.t_inflate, r.t_resolve, r.t_wait);
  return 0;
}
