// mkfmi.cpp — index builder: protein FASTA -> Kaiju .fmi (format-compatible with the
// reference's kaiju-mkbwt + kaiju-mkfmi, so that the reference binary and this library read the
// very same file).
//
// The reference builds the index off-line with a bucketed multikey quicksort
// (bwt/mkbwt.c:922-1099), then byte-codes the BWT and samples the rank checkpoints
// (bwt/mkfmi.c, bwt/compactfmi.c:446-457, bwt/fmicommon.h:95-171).  This is an independent
// implementation of the same definition:
//   * text = the sequences in file order, letter codes 1..20 ("*ACDEFGHIKLMNPQRSTVWY"), every
//     sequence followed by a terminator that sorts below all letters; terminators of
//     different sequences compare by file order (mkbwt.c:readOrder/encodeOrder, revsort off);
//   * BWT rows 0..nseq-1 are the terminator suffixes in file order (write_term, mkbwt.c:862-876),
//     then all letter suffixes in lexicographic order; the BWT letter in front of a sequence
//     start is the terminator;
//   * SA samples every 2^e rows (k >= nseq) hold (rank of the sequence among all sequences in
//     sorted order, offset), big-endian in nbytes (suffixArray.c:45-53,195-226);
//   * sequence names/lengths are stored in sorted-sequence order (SortSeqs, mkbwt.c:700-733);
//   * FMI: index1 every 2^16, index2 every 2^8, BWT bytes re-coded as (letter, in-block count)
//     with the code table of find_startLcode (compactfmi.c:108-150).
// Suffixes are sorted bucket-wise (first two letters) with std::sort on a word-at-a-time
// comparator, buckets in parallel.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>
#include <atomic>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <time.h>
#include <vector>

#include "mkfmi.h"

namespace {

thread_local std::string g_err;

// KAIJU_GPU_LOAD_TIMES=1: wall time of the builder's phases on stderr
struct BuildClock {
  bool on;
  double tl;
  static double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
  BuildClock() : on(getenv("KAIJU_GPU_LOAD_TIMES") != nullptr), tl(now()) {}
  void mark(const char *what) {
    if (!on) return;
    const double t = now();
    fprintf(stderr, "[kaiju mkfmi] %-34s %8.2f s\n", what, t - tl);
    tl = t;
  }
};

struct Seq { std::string id; uint64_t start; uint64_t len; };

int bits_needed(long k) { int i = 0; while (k >> i) ++i; return i; }   // suffixArray.c:57

template <class F> void parallel_chunks(unsigned nt, uint64_t n, F &&fn) {
  if (nt <= 1 || n < 2) { fn(0, n, 0u); return; }
  std::vector<std::thread> th;
  const uint64_t step = (n + nt - 1) / nt;
  for (unsigned t = 0; t < nt; t++) {
    const uint64_t b = std::min<uint64_t>(n, t * step), e = std::min<uint64_t>(n, b + step);
    if (b >= e) break;
    th.emplace_back([=, &fn]() { fn(b, e, t); });
  }
  for (auto &x : th) x.join();
}

// big arrays without the value-initialisation of std::vector (one thread zeroing tens of GB)
template <class T> struct RawArr {
  T *p = nullptr; uint64_t n = 0;
  RawArr() = default;
  explicit RawArr(uint64_t m) { alloc(m); }
  void alloc(uint64_t m) { free(p); n = m; p = static_cast<T *>(malloc((size_t)(m ? m : 1) * sizeof(T))); if (!p) throw std::bad_alloc(); }
  ~RawArr() { free(p); }
  RawArr(const RawArr &) = delete;
  RawArr &operator=(const RawArr &) = delete;
  T &operator[](uint64_t i) { return p[i]; }
  const T &operator[](uint64_t i) const { return p[i]; }
  T *data() { return p; }
  const T *data() const { return p; }
  uint64_t size() const { return n; }
};

// FASTA reader with the reference's conventions (readFasta.c:35-170): id = header up to the first
// blank; a record starts at '>' in column 0 (but not directly after a header line); letters are
// translated case-insensitively, alphabetic characters outside the alphabet become the last
// letter, everything else is skipped.  The file is mapped and cut into pieces at record starts that are certain
// ('>' in column 0 behind a line that is no header line); the pieces are parsed side by side.
struct FastaPiece { std::vector<Seq> seqs; std::vector<uint8_t> T; };
void parse_piece(const char *b, const char *e, const signed char *trans, FastaPiece &out, bool &bad) {
  const char *q = b;
  while (q < e && *q != '>') q++;
  while (q < e) {
    // header line (q at '>')
    const char *h = ++q;
    while (q < e && *q != '\n') q++;
    if (q >= e) { bad = true; return; }                 // EOF while reading an ID line
    size_t l = (size_t)(q - h);
    if (l > 10000) l = 10000;
    size_t w = 0;
    while (w < l && h[w] != ' ' && h[w] != '\t') w++;
    Seq s; s.id.assign(h, w); s.start = out.T.size(); s.len = 0;
    q++;                                               // behind the newline of the header
    int lastc = 0;
    while (q < e) {
      const int c = (unsigned char)*q;
      if (c == '>' && lastc == '\n') break;
      if (c < 128 && trans[c] >= 0) { out.T.push_back((uint8_t)trans[c]); s.len++; }
      lastc = c;
      q++;
    }
    out.T.push_back(0);                                // terminator
    out.seqs.push_back(std::move(s));
  }
}
bool read_fasta(const char *path, const signed char *trans, unsigned nt, std::vector<Seq> &seqs, RawArr<uint8_t> &T, uint64_t &tlen) {
  const int fd = open(path, O_RDONLY);
  if (fd < 0) { g_err = std::string("cannot open ") + path; return false; }
  struct stat st;
  if (fstat(fd, &st) != 0 || st.st_size <= 0) { close(fd); g_err = "no sequences read"; return false; }
  const uint64_t fsz = (uint64_t)st.st_size;
  const char *buf = static_cast<const char *>(mmap(nullptr, fsz, PROT_READ, MAP_PRIVATE, fd, 0));
  close(fd);
  if (buf == MAP_FAILED) { g_err = std::string("cannot map ") + path; return false; }
  (void)madvise(const_cast<char *>(buf), fsz, MADV_SEQUENTIAL);
  // cut points: the first '>' of the file, then '>' in column 0 behind a line that does not start with '>'
  std::vector<uint64_t> cut;
  const unsigned np = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(nt * 4ull, fsz >> 16));
  cut.push_back(0);
  for (unsigned k = 1; k < np; k++) {
    uint64_t p = fsz / np * k;
    if (p <= cut.back()) continue;
    for (;; p++) {
      if (p + 1 >= fsz) { p = fsz; break; }
      if (buf[p] == '\n' && buf[p + 1] == '>') {
        uint64_t ls = p;                               // start of the line that ends at p
        while (ls > 0 && buf[ls - 1] != '\n') ls--;
        if (buf[ls] != '>' && ls > 0) { p = p + 1; break; }
      }
    }
    if (p < fsz && p > cut.back()) cut.push_back(p);
  }
  cut.push_back(fsz);
  const size_t npieces = cut.size() - 1;
  std::vector<FastaPiece> pc(npieces);
  std::vector<uint8_t> badv(npieces, 0);
  {
    std::atomic<size_t> next(0);
    auto worker = [&]() {
      for (;;) {
        const size_t k = next.fetch_add(1);
        if (k >= npieces) break;
        pc[k].T.reserve((size_t)(cut[k + 1] - cut[k]));
        bool bad = false;
        parse_piece(buf + cut[k], buf + cut[k + 1], trans, pc[k], bad);
        badv[k] = bad;
      }
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < std::min<size_t>(nt, npieces); t++) th.emplace_back(worker);
    worker();
    for (auto &x : th) x.join();
  }
  munmap(const_cast<char *>(buf), fsz);
  for (size_t k = 0; k < npieces; k++) if (badv[k]) { g_err = "EOF while reading an ID line"; return false; }
  std::vector<uint64_t> toff(npieces + 1, 0), soff(npieces + 1, 0);
  for (size_t k = 0; k < npieces; k++) { toff[k + 1] = toff[k] + pc[k].T.size(); soff[k + 1] = soff[k] + pc[k].seqs.size(); }
  tlen = toff[npieces];
  if (soff[npieces] == 0) { g_err = "no sequences read"; return false; }
  T.alloc(tlen + 16);
  memset(T.data() + tlen, 0, 16);                      // padding for the word-wise comparator
  seqs.resize((size_t)soff[npieces]);
  {
    std::atomic<size_t> next(0);
    auto worker = [&]() {
      for (;;) {
        const size_t k = next.fetch_add(1);
        if (k >= npieces) break;
        if (!pc[k].T.empty()) memcpy(T.data() + toff[k], pc[k].T.data(), pc[k].T.size());
        for (size_t i = 0; i < pc[k].seqs.size(); i++) { Seq &d = seqs[(size_t)soff[k] + i]; d = std::move(pc[k].seqs[i]); d.start += toff[k]; }
        std::vector<uint8_t>().swap(pc[k].T);
        std::vector<Seq>().swap(pc[k].seqs);
      }
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < std::min<size_t>(nt, npieces); t++) th.emplace_back(worker);
    worker();
    for (auto &x : th) x.join();
  }
  return true;
}

// big arrays to a file: pieces written side by side (pwrite), the stream position moved behind them
bool big_write(FILE *fp, const void *src, uint64_t n, unsigned nt) {
  const uint64_t piece = 64ull << 20;
  if (n <= 2 * piece || nt <= 1) return n == 0 || fwrite(src, 1, (size_t)n, fp) == n;
  if (fflush(fp) != 0) return false;
  const off_t base = ftello(fp);
  if (base < 0) return fwrite(src, 1, (size_t)n, fp) == n;
  const int fd = fileno(fp);
  const uint8_t *s = static_cast<const uint8_t *>(src);
  const uint64_t np = (n + piece - 1) / piece;
  std::atomic<uint64_t> next(0);
  std::atomic<bool> ok(true);
  auto worker = [&]() {
    for (;;) {
      const uint64_t k = next.fetch_add(1);
      if (k >= np) break;
      uint64_t b = k * piece;
      const uint64_t e = std::min<uint64_t>(n, b + piece);
      while (b < e) { const ssize_t r = pwrite(fd, s + b, (size_t)(e - b), base + (off_t)b); if (r <= 0) { ok = false; return; } b += (uint64_t)r; }
    }
  };
  std::vector<std::thread> th;
  for (unsigned t = 1; t < std::min<uint64_t>(std::min(nt, 32u), np); t++) th.emplace_back(worker);
  worker();
  for (auto &x : th) x.join();
  return ok.load() && fseeko(fp, base + (off_t)n, SEEK_SET) == 0;
}

// suffix comparison from `depth` on: letters until a difference; two terminators at the same
// offset belong to different sequences and compare by file order == by position
struct SufLess {
  const uint8_t *T;
  int depth;
  bool operator()(uint32_t a, uint32_t b) const { return less(a, b); }
  bool operator()(uint64_t a, uint64_t b) const { return less(a, b); }
  template <class I> bool less(I a, I b) const {
    const uint8_t *pa = T + a + depth, *pb = T + b + depth;
    for (;;) {
      uint64_t wa, wb;
      memcpy(&wa, pa, 8); memcpy(&wb, pb, 8);
      const uint64_t x = wa ^ wb;
      const uint64_t zero = (wa - 0x0101010101010101ull) & ~wa & 0x8080808080808080ull;   // lowest set bit is exact
      if (x | zero) {
        const int dx = x ? __builtin_ctzll(x) >> 3 : 8;        // first differing byte
        const int dz = zero ? __builtin_ctzll(zero) >> 3 : 8;  // first terminator of a
        if (dz < dx) return a < b;          // both reach their terminator together: file order
        return pa[dx] < pb[dx];             // plain difference (a terminator is the smallest symbol)
      }
      pa += 8; pb += 8;
    }
  }
};

// copies > 1: the index of the database in which every sequence of the FASTA file occurs `copies` times in a row (copy t of
// sequence i is sequence i * copies + t of that database) - WITHOUT sorting it again: equal suffixes of different sequences
// compare by file order, so row r of the index of the file becomes the rows r * copies .. r * copies + copies - 1.  The output
// is byte for byte what this builder (and the reference's) writes for the FASTA with the repeats spelled out
// (tests/test_mkfmi_pin.py); a 2^32-row index for the tests of the wide path takes half a minute this way instead of the
// seven minutes of sorting 4.3 G suffixes.  copy_taxids (optional): copy t of a sequence named X_<id> is named
// X_<copy_taxids[(i + t) % n]>, so that the copies do not all carry one taxon.
template <class I>
int build(const char *faa, const char *out, int threads, int chpt_exp, uint64_t copies = 1, const uint64_t *copy_taxids = nullptr,
          uint32_t n_copy_taxids = 0) {
  // alphabet of kaiju-makedb:373 with the terminator in front (mkbwt.c:read_alphabet)
  const char alphabet[] = "*ACDEFGHIKLMNPQRSTVWY";
  const int alen = 21;
  signed char trans[128];
  trans[0] = 0;
  for (int i = 1; i < 128; i++) trans[i] = isalpha(i) ? (signed char)(alen - 1) : (signed char)-1;
  for (int i = 0; i < alen; i++) { trans[toupper(alphabet[i])] = (signed char)i; trans[tolower(alphabet[i])] = (signed char)i; }
  // the reference's table maps the terminator character itself ('*') to code 0, i.e. a '*' inside
  // the FASTA would plant a terminator in the middle of a sequence; it is skipped here
  trans[(int)'*'] = -1;
  std::vector<Seq> seqs;
  RawArr<uint8_t> T;
  BuildClock bc;
  const unsigned nt = (unsigned)std::max(1, threads);
  uint64_t tlen = 0;                              // == bwtlen of the file's own index
  if (!read_fasta(faa, trans, nt, seqs, T, tlen)) return KAIJU_GPU_ERR_IO;
  const uint64_t nseq = seqs.size();
  bc.mark("read FASTA");
  if (sizeof(I) == 4 && tlen >= 0xfffffff0ull) { g_err = "internal: index type too small"; return KAIJU_GPU_ERR_ARG; }

  // ---- bucket the letter suffixes by their first three symbols (what follows a terminator reads as terminators: the
  //      suffixes of a bucket whose key ends in a terminator are equal and stay in file order) ---------------------------
  const int NB = 21 * 21 * 21;
  auto bucket_of = [&](uint64_t p) -> uint32_t {
    const uint32_t c0 = T[p], c1 = T[p + 1], c2 = c1 ? T[p + 2] : 0u;
    return (c0 * 21u + c1) * 21u + c2;
  };
  std::vector<uint64_t> bstart(NB + 1, 0);
  std::vector<std::vector<uint64_t>> cnt(nt, std::vector<uint64_t>(NB, 0));
  parallel_chunks(nt, tlen, [&](uint64_t b, uint64_t e, unsigned t) {
    auto &c = cnt[t];
    for (uint64_t p = b; p < e; p++) if (T[p]) c[bucket_of(p)]++;
  });
  for (int k = 0; k < NB; k++) {
    uint64_t at = bstart[k];
    for (unsigned t = 0; t < nt; t++) { const uint64_t c = cnt[t][k]; cnt[t][k] = at; at += c; }   // where thread t's share of bucket k starts
    bstart[k + 1] = at;
  }
  bc.mark("bucket counts");
  const uint64_t nsuf = bstart[NB];
  if (nsuf + nseq != tlen) { g_err = "internal: suffix count"; return KAIJU_GPU_ERR_ARG; }
  RawArr<I> SA(nsuf);
  // positions enter their bucket in increasing order (needed for the terminator tie rule): the pieces of the text are the
  // ones of the counting pass, piece t fills its own share of every bucket
  parallel_chunks(nt, tlen, [&](uint64_t b, uint64_t e, unsigned t) {
    auto &fill = cnt[t];
    for (uint64_t p = b; p < e; p++) if (T[p]) SA[fill[bucket_of(p)]++] = (I)p;
  });
  cnt.clear(); cnt.shrink_to_fit();
  bc.mark("bucket fill");
  {
    std::vector<int> order;
    for (int k = 0; k < NB; k++) if (bstart[k + 1] - bstart[k] > 1 && (k % 21) != 0) order.push_back(k);
    std::sort(order.begin(), order.end(), [&](int a, int b) { return bstart[a + 1] - bstart[a] > bstart[b + 1] - bstart[b]; });
    std::atomic<size_t> next(0);
    auto worker = [&]() {
      for (;;) {
        const size_t q = next.fetch_add(1);
        if (q >= order.size()) break;
        const int k = order[q];
        std::sort(SA.data() + bstart[k], SA.data() + bstart[k + 1], SufLess{T.data(), 3});
      }
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; t++) th.emplace_back(worker);
    worker();
    for (auto &x : th) x.join();
  }
  bc.mark("suffix sort");

  // ---- BWT of the file's own index: rows 0 .. nseq-1 are the terminator suffixes in file order, then the sorted suffixes;
  //      a terminator in front of a suffix says that it is a whole sequence --------------------------------------------
  RawArr<uint8_t> B(tlen);
  parallel_chunks(nt, tlen, [&](uint64_t b, uint64_t e, unsigned) {
    for (uint64_t r = b; r < e; r++) {
      if (r < nseq) B[r] = seqs[r].len ? T[seqs[r].start + seqs[r].len - 1] : 0;
      else { const uint64_t p = SA[r - nseq]; B[r] = p ? T[p - 1] : 0; }
    }
  });
  bc.mark("BWT of the file");
  // ---- sequence ranks (order of the whole-sequence suffixes), names in sorted order ---------
  std::vector<uint64_t> starts(nseq);
  for (uint64_t i = 0; i < nseq; i++) starts[i] = seqs[i].start;
  auto seq_index = [&](uint64_t p) { return (uint64_t)(std::upper_bound(starts.begin(), starts.end(), p) - starts.begin() - 1); };
  std::vector<uint32_t> rank_of(nseq, 0), read_of_rank(nseq, 0);
  {
    uint32_t r0 = 0;
    for (uint64_t i = 0; i < nseq; i++) if (seqs[i].len == 0) { rank_of[i] = r0; read_of_rank[r0] = (uint32_t)i; r0++; }   // empty sequences sort first
    std::vector<uint64_t> zc(nt + 1, 0);
    parallel_chunks(nt, nsuf, [&](uint64_t b, uint64_t e, unsigned t) {
      uint64_t z = 0;
      for (uint64_t k = b; k < e; k++) z += B[nseq + k] == 0;
      zc[t + 1] = z;
    });
    for (unsigned t = 0; t < nt; t++) zc[t + 1] += zc[t];
    if (r0 + zc[nt] != nseq) { g_err = "internal: sequence ranks"; return KAIJU_GPU_ERR_ARG; }
    parallel_chunks(nt, nsuf, [&](uint64_t b, uint64_t e, unsigned t) {
      uint64_t r = r0 + zc[t];
      for (uint64_t k = b; k < e; k++) if (B[nseq + k] == 0) {
        const uint64_t i = seq_index(SA[k]);
        rank_of[i] = (uint32_t)r; read_of_rank[r] = (uint32_t)i; r++;
      }
    });
  }
  bc.mark("sequence ranks");

  // ---- BWT -----------------------------------------------------------------------------------
  const uint64_t N = copies < 1 ? 1 : copies;
  const uint64_t otlen = tlen * N, onseq = nseq * N;          // what the output describes
  if (onseq >= 0x7fffffffull) { g_err = "too many sequences for the .fmi header (int32)"; return KAIJU_GPU_ERR_ARG; }
  RawArr<uint8_t> bwt_own;
  uint8_t *bwt = B.data();
  if (N > 1) {
    bwt_own.alloc(otlen);
    bwt = bwt_own.data();
    parallel_chunks(nt, tlen, [&](uint64_t b, uint64_t e, unsigned) {
      for (uint64_t r = b; r < e; r++) memset(bwt + r * N, B[r], (size_t)N);
    });
  }
  bc.mark("BWT");

  // ---- suffix array samples (init_suffixArray suffixArray.c:140-180) -------------------------
  long maxlen = 0;
  for (auto &s : seqs) if ((long)s.len > maxlen) maxlen = (long)s.len;
  const int sbits = bits_needed((long)onseq), pbits = bits_needed(maxlen);
  const int nbytes = (7 + sbits + pbits) / 8;
  const int64_t ncheck = (int64_t)(otlen >> chpt_exp) - (int64_t)(onseq >> chpt_exp);   // the header value
  const int64_t mask = (1 << pbits) - 1, check = (1 << chpt_exp) - 1;
  RawArr<uint8_t> sa((uint64_t)std::max<int64_t>(ncheck, 0) * nbytes);
  {
    const uint64_t step = 1ull << chpt_exp;
    const uint64_t first = ((onseq + step - 1) >> chpt_exp) << chpt_exp;   // first sampled row >= nseq
    const uint64_t nsamp = first < otlen ? ((otlen - 1 - first) >> chpt_exp) + 1 : 0;
    if ((int64_t)nsamp < ncheck) memset(sa.data() + nsamp * nbytes, 0, (size_t)(ncheck - (int64_t)nsamp) * nbytes);
    parallel_chunks(nt, nsamp, [&](uint64_t b, uint64_t e, unsigned) {
      for (uint64_t q = b; q < e; q++) {
        if ((int64_t)q >= ncheck) break;            // mkfmi only carries ncheck entries over
        const uint64_t k = first + (q << chpt_exp);
        const uint64_t r = k / N, t = k % N;         // row of the file's own index, copy
        const uint64_t p = SA[r - nseq];
        const uint64_t i = seq_index(p);
        long val = (long)((uint64_t)rank_of[i] * N + t);
        val = (val << pbits) + (long)(p - seqs[i].start);
        uint8_t *c = sa.data() + q * nbytes;
        for (int n = nbytes; n-- > 0;) { c[n] = (uint8_t)val; val >>= 8; }
      }
    });
  }
  bc.mark("suffix array samples");

  // ---- FMI (fmicommon.h:77-171, compactfmi.c:108-150,399-457) ---------------------------------
  const int64_t bwtlen = (int64_t)otlen;
  int N1 = (int)(((bwtlen - 1) >> 16) + 2);
  if (((int64_t)N1 << 16) == bwtlen) N1 -= 1;
  int N2 = (int)(((bwtlen - 1) >> 8) + 2);
  if (((int64_t)N1 << 8) == bwtlen) N2 -= 1;
  std::vector<int64_t> index1((size_t)N1 * alen, 0);
  RawArr<uint16_t> index2((uint64_t)N2 * alen);
  int64_t total[32] = {0};
  {
    // letter counts of every piece of 2^16 rows, their running sums (index1 before C[] is added), then the 2^8 checkpoints
    // of every piece relative to its start
    const uint64_t nR1 = ((uint64_t)bwtlen + 65535) >> 16;
    parallel_chunks(nt, nR1, [&](uint64_t b, uint64_t e, unsigned) {
      for (uint64_t R = b; R < e; R++) {
        int64_t h[32] = {0};
        const uint64_t lo = R << 16, hi = std::min<uint64_t>((uint64_t)bwtlen, lo + 65536);
        for (uint64_t ii = lo; ii < hi; ii++) h[bwt[ii]]++;
        for (int a = 0; a < alen; a++) index1[(size_t)R * alen + a] = h[a];
      }
    });
    for (uint64_t R = 0; R < nR1; R++)
      for (int a = 0; a < alen; a++) { const int64_t h = index1[(size_t)R * alen + a]; index1[(size_t)R * alen + a] = total[a]; total[a] += h; }
    // (a last row of index2 behind the end of the BWT, and index1 rows behind the last piece, as the serial loop leaves them)
    parallel_chunks(nt, nR1, [&](uint64_t b, uint64_t e, unsigned) {
      for (uint64_t R = b; R < e; R++) {
        int64_t run[32] = {0};
        const uint64_t lo = R << 16, hi = std::min<uint64_t>((uint64_t)bwtlen, lo + 65536);
        for (uint64_t ii = lo; ii < hi; ++ii) {
          if (!(ii & 255)) {
            const uint64_t R2 = ii >> 8;
            if (ii > 0) for (int a = 0; a < alen; a++) index2[R2 * alen + a] = (uint16_t)run[a];
            else for (int a = 0; a < alen; a++) index2[a] = 0;
          }
          run[bwt[ii]]++;
        }
      }
    });
    const uint64_t R1 = ((uint64_t)bwtlen - 1) >> 16;        // the piece the serial loop was in when it ended
    const int64_t R2 = N2 - 1;
    for (uint64_t q = (((uint64_t)bwtlen - 1) >> 8) + 1; q < (uint64_t)R2; q++) for (int a = 0; a < alen; a++) index2[q * alen + a] = 0;
    for (int a = 0; a < alen; a++) index2[(uint64_t)R2 * alen + a] = (uint16_t)(total[a] - index1[(size_t)R1 * alen + a]);
    int64_t *last = &index1[(size_t)(N1 - 1) * alen];
    last[0] = 0;
    for (int a = 1; a < alen; a++) last[a] = last[a - 1] + total[a - 1];
    for (int64_t r = 0; r < N1 - 1; ++r) for (int a = 1; a < alen; a++) index1[(size_t)r * alen + a] += last[a];
  }
  bc.mark("FMI checkpoints");
  // find_startLcode, compactfmi.c:108-150
  int startLcode[32] = {0};
  {
    int maxN[32] = {0};
    int64_t tot = 0;
    for (int a = 0; a < alen; a++) tot += total[a];
    int sum = 0, mx = 0;
    for (int a = 0; a < alen; a++) {
      maxN[a] = (int)(256 * ((double)total[a] / tot));
      if (maxN[a] < 2) maxN[a] = 2;
      if (maxN[a] > maxN[mx]) mx = a;
      sum += maxN[a];
    }
    if (sum < 256) maxN[mx] += 256 - sum;
    while (sum > 256) {
      int mn = 0;
      for (int a = 1; a < alen; a++) {
        if (maxN[mn] <= 2) mn = a;
        if (maxN[a] > 2 && maxN[a] < maxN[mn]) mn = a;
      }
      maxN[mn] -= 1;
      sum -= 1;
    }
    startLcode[0] = 0;
    for (int a = 0; a < alen; a++) startLcode[a + 1] = startLcode[a] + maxN[a];
    startLcode[alen] = 256;
  }
  // FMIrecode, compactfmi.c:399-440: first half of a 256-block stores the number of equal letters
  // before the position, second half the number after it, saturating at the letter's top code
  {
    auto encode = [&](uint8_t c, int n) {
      const int mx = startLcode[c + 1] - startLcode[c] - 1;
      if (n > mx) n = mx;
      return (uint8_t)(startLcode[c] + n);
    };
    const uint64_t nblk = ((uint64_t)bwtlen + 255) >> 8;
    parallel_chunks(nt, nblk, [&](uint64_t b, uint64_t e, unsigned) {
      int current[32], delta[256];
      for (uint64_t blk = b; blk < e; blk++) {
        uint8_t *s = bwt + (blk << 8);
        const int n = (int)std::min<uint64_t>(256, (uint64_t)bwtlen - (blk << 8));
        for (int a = 0; a < alen; a++) current[a] = 0;
        for (int i = 0; i < n; i++) { delta[i] = current[s[i]]; current[s[i]] += 1; }
        int j = 0;
        for (; j < 128 && j < n; ++j) s[j] = encode(s[j], delta[j]);
        for (; j < n; ++j) s[j] = encode(s[j], (current[s[j]] - delta[j]) - 1);
      }
    });
  }

  bc.mark("recode");
  // ---- write the file (bwt.c:38-44, suffixArray.c:255-275, fmicommon.h:176-186, compactfmi.c:175-178)
  FILE *fp = fopen(out, "wb");
  if (!fp) { g_err = std::string("cannot write ") + out; return KAIJU_GPU_ERR_IO; }
  setvbuf(fp, nullptr, _IOFBF, 1 << 22);
  auto W = [&](const void *p, size_t n) { return n == 0 || fwrite(p, 1, n, fp) == n; };
  bool ok = true;
  const int32_t nseq32 = (int32_t)onseq, alen32 = alen;
  ok &= W(&bwtlen, 8); ok &= W(&nseq32, 4); ok &= W(&alen32, 4); ok &= W(alphabet, alen);
  ok &= W(&bwtlen, 8); ok &= W(&ncheck, 8);
  const int32_t e32 = chpt_exp, nb32 = nbytes, sb32 = sbits, pb32 = pbits;
  ok &= W(&e32, 4); ok &= W(&nb32, 4); ok &= W(&sb32, 4); ok &= W(&pb32, 4);
  ok &= W(&mask, 8); ok &= W(&check, 8); ok &= W(&nseq32, 4);
  for (uint64_t r = 0; r < nseq; r++) {
    const uint64_t fi = read_of_rank[r];
    const std::string &id0 = seqs[fi].id;
    for (uint64_t t = 0; t < N; t++) {
      std::string idt;
      const std::string *id = &id0;
      if (N > 1 && n_copy_taxids) {
        const size_t us = id0.rfind('_');
        idt = (us == std::string::npos ? id0 : id0.substr(0, us)) + "_" + std::to_string(copy_taxids[(fi + t) % n_copy_taxids]);
        id = &idt;
      }
      const uint8_t l = (uint8_t)std::min<size_t>(255, id->size());
      ok &= W(&l, 1); ok &= W(id->data(), l);
    }
  }
  {
    std::vector<int32_t> sto(onseq);
    std::vector<int64_t> sl(onseq);
    for (uint64_t r = 0; r < nseq; r++)
      for (uint64_t t = 0; t < N; t++) { sto[r * N + t] = (int32_t)(read_of_rank[r] * N + t); sl[r * N + t] = (int64_t)seqs[read_of_rank[r]].len; }
    ok &= W(sto.data(), onseq * 4); ok &= W(sl.data(), onseq * 8);
  }
  ok &= big_write(fp, sa.data(), sa.size(), nt);
  const int32_t n1 = N1, n2 = N2;
  ok &= W(&alen32, 4); ok &= W(&bwtlen, 8); ok &= W(&n1, 4); ok &= W(&n2, 4);
  ok &= big_write(fp, bwt, (uint64_t)bwtlen, nt);
  ok &= W(index1.data(), index1.size() * 8);
  ok &= big_write(fp, index2.data(), index2.size() * 2, nt);
  ok &= W(startLcode, (size_t)(alen + 1) * 4);
  ok &= fclose(fp) == 0;
  if (!ok) { g_err = "write error"; return KAIJU_GPU_ERR_IO; }
  bc.mark("write");
  return KAIJU_GPU_OK;
}

}  // namespace

extern "C" int kaiju_build_fmi(const char *faa_path, const char *out_fmi_path, int threads, int chpt_exp) {
  if (!faa_path || !out_fmi_path || chpt_exp < 0 || chpt_exp > 20) return KAIJU_GPU_ERR_ARG;
  if (threads <= 0) threads = (int)std::max(1u, std::thread::hardware_concurrency());
  // 32-bit suffix positions suffice below 4 G symbols; the 64-bit instantiation covers the rest
  FILE *fp = fopen(faa_path, "rb");
  if (!fp) { g_err = std::string("cannot open ") + faa_path; return KAIJU_GPU_ERR_IO; }
  fseeko(fp, 0, SEEK_END);
  const off_t sz = ftello(fp);
  fclose(fp);
  // (KAIJU_MKFMI_FORCE64=1: the 64-bit instantiation on a small file - tests/test_mkfmi_pin.py)
  if ((uint64_t)sz < 0xf0000000ull && !getenv("KAIJU_MKFMI_FORCE64")) return build<uint32_t>(faa_path, out_fmi_path, threads, chpt_exp);
  return build<uint64_t>(faa_path, out_fmi_path, threads, chpt_exp);
}

extern "C" int kaiju_build_fmi_replicated(const char *faa_path, const char *out_fmi_path, int threads, int chpt_exp, uint64_t copies,
                                          const uint64_t *copy_taxids, uint32_t n_copy_taxids) {
  if (!faa_path || !out_fmi_path || chpt_exp < 0 || chpt_exp > 20 || copies < 1 || (n_copy_taxids && !copy_taxids)) return KAIJU_GPU_ERR_ARG;
  if (threads <= 0) threads = (int)std::max(1u, std::thread::hardware_concurrency());
  // (the FILE is sorted, so 32-bit suffix positions do as long as it has fewer than 4 G symbols)
  FILE *fp = fopen(faa_path, "rb");
  if (!fp) { g_err = std::string("cannot open ") + faa_path; return KAIJU_GPU_ERR_IO; }
  fseeko(fp, 0, SEEK_END);
  const off_t sz = ftello(fp);
  fclose(fp);
  if ((uint64_t)sz < 0xf0000000ull && !getenv("KAIJU_MKFMI_FORCE64")) return build<uint32_t>(faa_path, out_fmi_path, threads, chpt_exp, copies, copy_taxids, n_copy_taxids);
  return build<uint64_t>(faa_path, out_fmi_path, threads, chpt_exp, copies, copy_taxids, n_copy_taxids);
}

extern "C" const char *kaiju_build_fmi_error(void) { return g_err.c_str(); }
