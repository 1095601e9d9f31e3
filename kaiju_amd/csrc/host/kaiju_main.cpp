// kaiju_main.cpp — drop-in `kaiju` command line on top of the C-ABI (include/kaiju_gpu.h).
//
// Same options, input handling and output format as the reference driver
// (/root/reference/src/kaiju.cpp:52-452): FASTA/FASTQ (optionally gzip) single or paired input,
// names cut at the first of " /\t\r", sequences strip()'d of non-letters, one output line per
// read  "C\tname\ttaxon" / "U\tname\t0"  in input order (the reference's -z 1 order).  What
// differs is the mechanics: instead of one ReadItem per mutex hand-off
// (kaiju.cpp:288-394, ProducerConsumerQueue.tpp:38-84) the input is cut into blocks of whole
// records by one reader thread per file (gzip or plain), the blocks are parsed in parallel into the
// batch layout of the C-ABI, classified on the GPU batch-wise by two contexts that ping-pong (copies
// of one batch overlap with the kernels of the other), and formatted in parallel; a writer puts
// the text out in input order.  Without -v only 16-byte records (LCA computed on the device) come
// back from the GPU.
//
// -v prints the reference's columns 4-7 (match length / score, matching taxon ids, accessions,
// matched peptides).  -p: the reads are protein sequences (kaiju.cpp:94, ConsumerThread.cpp:640-696).
// Started under the name kaijux the program reports database sequences instead of taxa (kaijux.cpp), under
// kaijup it does that for protein reads (kaijup.cpp, ConsumerThreadp.cpp), under kaiju-multi it takes comma
// separated file lists (kaiju-multi.cpp).
#include <cerrno>
#include <getopt.h>
#include <malloc.h>
#include <sys/mman.h>
#include <signal.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#include "pargz.h"
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include <algorithm>
#include <atomic>
#include <map>

#include "../../../include/kaiju_gpu.h"

namespace {

void usage(const char *prog) {
  fprintf(stderr, "Kaiju (MI355X path) — classification of sequencing reads with a protein FM-index\n\n");
  fprintf(stderr, "Usage:\n   %s -t nodes.dmp -f kaiju_db.fmi -i reads.fastq [-j reads2.fastq]\n\n", prog);
  fprintf(stderr, "Mandatory arguments:\n");
  fprintf(stderr, "   -t FILENAME   Name of nodes.dmp file\n");
  fprintf(stderr, "   -f FILENAME   Name of database (.fmi) file\n");
  fprintf(stderr, "   -i FILENAME   Name of input file containing reads in FASTA or FASTQ format\n\n");
  fprintf(stderr, "Optional arguments:\n");
  fprintf(stderr, "   -j FILENAME   Name of second input file for paired-end reads\n");
  fprintf(stderr, "   -o FILENAME   Name of output file. If not specified, output will be printed to STDOUT\n");
  fprintf(stderr, "   -z INT        Accepted for compatibility (the search runs on the GPU)\n");
  fprintf(stderr, "   -a STRING     Run mode, either \"mem\"  or \"greedy\" (default: greedy)\n");
  fprintf(stderr, "   -e INT        Number of mismatches allowed in Greedy mode (default: 3)\n");
  fprintf(stderr, "   -m INT        Minimum match length (default: 11)\n");
  fprintf(stderr, "   -s INT        Minimum match score in Greedy mode (default: 65)\n");
  fprintf(stderr, "   -E FLOAT      Minimum E-value in Greedy mode (default: 0.01)\n");
  fprintf(stderr, "   -x            Enable SEG low complexity filter (enabled by default)\n");
  fprintf(stderr, "   -X            Disable SEG low complexity filter\n");
  fprintf(stderr, "   -p            Input sequences are protein sequences\n");
  fprintf(stderr, "   -v            Enable verbose output\n");
  exit(EXIT_FAILURE);
}

void die(const std::string &msg) {
  // may be called from a reader / parser / GPU worker thread while others are inside HIP calls: no atexit handlers, no
  // static destructors (exit() there can hang or crash) - flush what is ours and leave
  fprintf(stderr, "Error: %s\n\n", msg.c_str());
  fflush(nullptr);
  _exit(EXIT_FAILURE);
}

// the output could not be written (fwrite / fflush / fclose failed).  A closed pipe (`kaiju ... | head`) ends the process
// quietly with a failure status - the reference dies of SIGPIPE there -, anything else (disk full, I/O error) with a message.
void write_failed() {
  if (errno == EPIPE) { fflush(stderr); _exit(EXIT_FAILURE); }
  die(std::string("could not write the output: ") + strerror(errno));
}

std::string now() {
  time_t t = time(nullptr);
  char buf[16] = {0};
  strftime(buf, sizeof buf, "%H:%M:%S", localtime(&t));
  return buf;
}

// ---------------------------------------------------------------------------------------------
// stage 1: a reader thread per input file cuts the (decompressed) text into blocks of whole records
// ---------------------------------------------------------------------------------------------
struct RawBlock {
  const char *text = nullptr;  // whole records (a last line may lack its '\n')
  size_t size = 0;
  std::vector<char> own;       // backing store unless the file is memory-mapped
  uint32_t n_records = 0;
  bool fastq = false;
  // .gz input inflated by several threads: the block's text as stretches of the inflated pieces, which `keep` holds alive;
  // gather() - run by the PARSER that takes the block, not by the one reader thread - makes it one stretch
  std::vector<std::pair<const char *, size_t>> segs;
  std::vector<std::shared_ptr<pargz::Text>> keep;
  void gather() {
    if (segs.empty()) return;
    if (segs.size() == 1) { text = segs[0].first; return; }      // (the pieces stay alive with the block)
    own.clear(); own.reserve(size);
    for (const auto &sg : segs) own.insert(own.end(), sg.first, sg.first + sg.second);
    text = own.data();
    segs.clear(); keep.clear();
  }
};

// Hands out blocks of `want` whole records.  Plain files are memory-mapped (blocks point into the
// mapping), gzip files are inflated through zlib into a growing buffer.  Record boundaries as the
// reference finds them (kaiju.cpp:288-331): empty lines before a header are skipped; FASTQ = header +
// 3 lines; FASTA = header + every line up to the next one starting with '>'.
// newline positions of a stretch of text, found 16 bytes at a time: the sequential walk over a memory-mapped file asks
// for four line ends per FASTQ record, and one memchr call per line costs more than the bytes it looks at
struct NewlineScan {
  std::vector<size_t> nl;                  // positions of the '\n' bytes in [from, scanned_to), ascending
  size_t idx = 0, scanned_to = 0;
  void refill(const char *data, size_t from, size_t size) {
    nl.clear(); idx = 0;
    const size_t to = std::min(size, from + (256u << 10));
    size_t p = from;
#if defined(__SSE2__)
    const __m128i nlv = _mm_set1_epi8('\n');
    for (; p + 16 <= to; p += 16) {
      unsigned m = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i *>(data + p)), nlv));
      while (m) { nl.push_back(p + (size_t)__builtin_ctz(m)); m &= m - 1; }
    }
#endif
    for (; p < to; p++) if (data[p] == '\n') nl.push_back(p);
    scanned_to = to;
  }
  // one past the first '\n' at or behind p; `size` when the text ends without one (p < size)
  size_t line_end(const char *data, size_t p, size_t size) {
    for (;;) {
      while (idx < nl.size() && nl[idx] < p) idx++;
      if (idx < nl.size()) return nl[idx] + 1;
      if (scanned_to >= size && p <= scanned_to) return size;
      refill(data, std::max(p, scanned_to), size);
    }
  }
};

// input files mapped for the sample that is being processed (kaiju-multi: unmapped between samples)
static std::mutex g_map_mutex;
static std::vector<std::pair<void *, size_t>> g_mappings;
static void release_mappings() {
  std::lock_guard<std::mutex> lk(g_map_mutex);
  for (auto &m : g_mappings) munmap(m.first, m.second);
  g_mappings.clear();
}

struct BlockReader {
  std::string path;
  bool ok = false, mapped = false;
  // mapped
  const char *map = nullptr;
  size_t map_size = 0;
  // streamed
  gzFile fp = nullptr;
  std::shared_ptr<pargz::Reader> pz;       // .gz files of some size: inflated by several threads (pargz.h)
  std::vector<char> buf;
  // common view of the text not yet handed out: [data + pos, data + size)
  const char *data = nullptr;
  size_t size = 0, pos = 0;
  bool eof = false, first = true, fastq = false;
  static constexpr size_t NEED_MORE = ~(size_t)0, NONE = ~(size_t)0 - 1;

  explicit BlockReader(const std::string &p) : path(p) {
    unsigned char magic[2] = {0, 0};
    FILE *f = fopen(p.c_str(), "rb");
    if (!f) return;
    const size_t got = fread(magic, 1, 2, f);
    const bool gz = got == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
    struct stat st;
    const bool regular = fstat(fileno(f), &st) == 0 && S_ISREG(st.st_mode);
    if (!gz && regular && st.st_size > 0) {
      void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fileno(f), 0);
      if (m != MAP_FAILED) {
        madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
        map = static_cast<const char *>(m); map_size = (size_t)st.st_size;
        { std::lock_guard<std::mutex> lk(g_map_mutex); g_mappings.emplace_back(m, map_size); }
        mapped = true; data = map; size = map_size; eof = true; ok = true;
      }
    }
    // a .gz file of some size is mapped as it is and inflated by several threads (pargz.h: one zlib stream feeds 1.7 M reads/s,
    // the GPU takes fifty times that); KAIJU_GPU_GZ_THREADS=1, small files and pipes keep zlib's gzread
    long gz_min = 2 << 20;
    if (const char *e = getenv("KAIJU_GPU_GZ_MIN")) gz_min = atol(e);       // (tests: the several-thread path on small files)
    if (gz && regular && st.st_size >= gz_min && st.st_size >= 64) {
      unsigned zt = std::min(32u, std::max(1u, std::thread::hardware_concurrency() / 2));
      if (const char *e = getenv("KAIJU_GPU_GZ_THREADS")) zt = (unsigned)std::max(1, atoi(e));
      if (zt >= 2) {
        void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fileno(f), 0);
        if (m != MAP_FAILED) {
          madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
          { std::lock_guard<std::mutex> lk(g_map_mutex); g_mappings.emplace_back(m, (size_t)st.st_size); }
          pz.reset(new pargz::Reader());
          const std::string name = p;
          if (pz->open(static_cast<const uint8_t *>(m), (size_t)st.st_size, zt, [name](const std::string &msg) { die(msg + " (" + name + ")"); })) ok = true;
          else pz.reset();
        }
      }
    }
    fclose(f);
    if (!mapped && !pz) {
      fp = gzopen(p.c_str(), "rb");
      if (fp) { gzbuffer(fp, 1 << 20); ok = true; }
    }
  }
  ~BlockReader() {
    if (pz && getenv("KAIJU_GPU_STAGE_TIMES"))
      fprintf(stderr, "[gz %s] producer: block-start search %.2f s, inflate %.2f s, markers + newlines + crc %.2f s, waiting for the reader %.2f s; "
              "%llu pieces entered at a found block start, %llu inflated by the thread in front of them\n", path.c_str(), pz->t_find, pz->t_inflate,
              pz->t_resolve, pz->t_wait, (unsigned long long)pz->pieces_entered, (unsigned long long)pz->pieces_absorbed);
    for (auto &x : scanners) if (x.joinable()) x.join();
    if (fp) gzclose(fp);
    // (the mapping outlives the reader: blocks in the pipeline still point into it; run_sample() unmaps it when every
    // block of the sample has been consumed - release_mappings())
  }
  bool more() {                // streamed files: append more text; false at end of file
    if (eof) return false;
    if (pos > 0) { buf.erase(buf.begin(), buf.begin() + (long)pos); pos = 0; }
    const size_t old = buf.size(), chunk = 1 << 24;
    buf.resize(old + chunk);
    const long n = pz ? (long)pz->read(buf.data() + old, chunk) : (long)gzread(fp, buf.data() + old, (unsigned)chunk);
    if (!pz && n < 0) {                       // (a damaged .gz file is an error, not the end of the input: zstr throws there)
      int errnum = 0;
      const char *msg = gzerror(fp, &errnum);
      die("Error while reading " + path + ": " + (msg ? msg : "zlib error"));
    }
    buf.resize(old + (size_t)(n > 0 ? n : 0));
    if (n <= 0) eof = true;
    data = buf.data(); size = buf.size();
    return n > 0;
  }
  // one past the '\n' of the line starting at p; at end of file a last line without '\n' ends at size
  size_t line_end(size_t p) const {
    if (p >= size) return NEED_MORE;
    const void *nl = memchr(data + p, '\n', size - p);
    if (nl) return (size_t)((const char *)nl - data) + 1;
    return eof ? size : NEED_MORE;
  }
  // end of the record that starts at the first non-empty line at or after p
  size_t record_end(size_t p) { return record_end_with(p, [this](size_t q) { return line_end(q); }); }
  NewlineScan scan;                          // the streaming walk over a mapped file (one thread)
  size_t record_end_scan(size_t p) {
    return record_end_with(p, [this](size_t q) { return q >= size ? NEED_MORE : scan.line_end(data, q, size); });
  }
  template <class LineEnd>
  size_t record_end_with(size_t p, LineEnd &&line_end) {
    size_t q = p, e;
    for (;;) {
      if (q >= size) return eof ? NONE : NEED_MORE;
      e = line_end(q);
      if (e == NEED_MORE) return NEED_MORE;
      if (e - q == 1 && data[q] == '\n') { q = e; continue; }      // getline gave an empty line
      break;
    }
    if (first) {
      if (data[q] == '@') fastq = true;
      else if (data[q] != '>') die("Auto-detection of file type for file " + path + " failed.");
      first = false;
    }
    size_t r = e;
    if (fastq) {
      for (int lines = 0; lines < 3; lines++) {
        if (r >= size) { if (eof) break; return NEED_MORE; }      // fewer lines at the end of the file are fine
        const size_t e2 = line_end(r);
        if (e2 == NEED_MORE) return NEED_MORE;
        r = e2;
      }
      return r;
    }
    for (;;) {
      if (r >= size) return eof ? r : NEED_MORE;
      if (data[r] == '>') return r;
      const size_t e2 = line_end(r);
      if (e2 == NEED_MORE) return NEED_MORE;
      r = e2;
    }
  }
  // ---- memory-mapped files: the record boundaries of the whole file, found by several threads -----------------
  // Thread t guesses a record start at or behind byte t*size/T (FASTA: a line starting with '>'; FASTQ: a line starting
  // with '@' whose second next line starts with '+') and walks the records from there with the very state machine of
  // record_end().  The guess of thread t+1 is confirmed if thread t's walk arrives exactly there; otherwise (a quality
  // line that looks like a header, a file that is not made of four-line records ...) everything is redone
  // sequentially, so the boundaries are always those of the sequential walk.
  std::vector<size_t> rec_start;        // start offset of every record (leading empty lines included), then the end
  bool prescanned = false;
  size_t next_rec = 0;
  // header position of the record at/after p (empty lines skipped), NONE if only empty lines are left
  size_t header_at(size_t p) const {
    for (;;) {
      if (p >= size) return NONE;
      const size_t e = line_end(p);
      if (e - p == 1 && data[p] == '\n') { p = e; continue; }
      return p;
    }
  }
  size_t guess_start(size_t from) const {
    size_t q = from;
    if (q > 0) { const void *nl = memchr(data + q - 1, '\n', size - (q - 1)); if (!nl) return size; q = (size_t)((const char *)nl - data) + 1; }
    while (q < size) {
      const size_t e1 = line_end(q);
      if (!fastq) { if (data[q] == '>') return q; }
      else if (data[q] == '@' && e1 < size) {
        const size_t e2 = line_end(e1);
        if (e2 < size && data[e2] == '+') return q;
      }
      q = e1;
    }
    return size;
  }
  // The file is cut into pieces (64 MB, at least one per thread) that a pool of threads walks in file order; next() merges
  // the pieces one after the other as it needs records, so the first blocks are on their way while the rest of a large
  // file is still being scanned.
  struct Piece { size_t guess = 0, limit = 0, stop = 0; std::vector<size_t> starts; bool done = false; };
  std::vector<Piece> pieces;
  std::vector<std::thread> scanners;
  std::mutex pm;
  std::condition_variable pcv;
  std::atomic<size_t> next_piece{0};
  size_t merged = 0, prev_end = 0;
  bool all_merged = false;
  void prescan(unsigned threads) {
    // default on hosts with 16 hardware threads or more (one reader thread walks ~13 M records/s, the ceiling of the whole
    // pipeline there); on the 8-core build box the streaming walk, which overlaps with parsing, is as fast.
    // KAIJU_GPU_PRESCAN=0 / 1 switches it off / on everywhere
    const char *pe = getenv("KAIJU_GPU_PRESCAN");
    const bool want = pe ? atoi(pe) != 0 : std::thread::hardware_concurrency() >= 16;
    if (!mapped || prescanned || !want) return;
    prescanned = true;
    rec_start.clear();
    if (record_end(pos) == NONE) { rec_start.assign(1, size); all_merged = true; return; }   // (also settles fastq / fasta)
    size_t min_bytes = 32u << 20, piece_bytes = 64u << 20;
    if (const char *e = getenv("KAIJU_GPU_PRESCAN_MIN")) min_bytes = (size_t)atol(e);     // (tests: several threads on small files)
    if (const char *e = getenv("KAIJU_GPU_PRESCAN_PIECE")) piece_bytes = (size_t)std::max(1L, atol(e));
    const unsigned T = size >= min_bytes ? std::max(1u, threads) : 1u;
    const size_t P = T == 1 ? 1 : std::max<size_t>(T, (size - pos + piece_bytes - 1) / piece_bytes);
    pieces.resize(P);
    pieces[0].guess = pos;
    for (size_t t = 1; t < P; t++) pieces[t].guess = guess_start(pos + (size_t)((unsigned __int128)(size - pos) * t / P));
    for (size_t t = 1; t < P; t++) if (pieces[t].guess < pieces[t - 1].guess) pieces[t].guess = pieces[t - 1].guess;
    for (size_t t = 0; t < P; t++) pieces[t].limit = t + 1 < P ? pieces[t + 1].guess : size;
    prev_end = pos;
    auto work = [this] {
      for (;;) {
        const size_t t = next_piece.fetch_add(1);
        if (t >= pieces.size()) return;
        Piece &pc = pieces[t];
        size_t p = pc.guess;
        while (p < pc.limit) {
          const size_t h = header_at(p);
          if (h == NONE || h >= pc.limit) break;
          pc.starts.push_back(p);
          p = record_end(p);                   // (const for mapped files once the type is known)
        }
        pc.stop = p;
        { std::lock_guard<std::mutex> lk(pm); pc.done = true; }
        pcv.notify_all();
      }
    };
    const unsigned nthreads = (unsigned)std::min<size_t>(T, P);
    for (unsigned t = 0; t < nthreads; t++) scanners.emplace_back(work);
  }
  void join_scanners() { for (auto &x : scanners) x.join(); scanners.clear(); }
  // appends the record starts of the next piece (waiting for its walk); the guess a piece started from is confirmed when
  // the walk of the piece before it arrives exactly there with nothing but empty lines in between - otherwise (a quality
  // line that looks like a header, a file that is not made of four-line records ...) the rest of the file is walked
  // sequentially from the last confirmed record, so the boundaries are always those of the sequential walk
  void merge_next_piece() {
    if (all_merged) return;
    const size_t t = merged;
    { std::unique_lock<std::mutex> lk(pm); pcv.wait(lk, [&] { return pieces[t].done; }); }
    Piece &pc = pieces[t];
    bool good = true;
    if (t + 1 < pieces.size()) {
      const size_t h = header_at(pc.stop), want_h = header_at(pieces[t + 1].guess);
      if (pc.stop > pieces[t + 1].guess || h != want_h) good = false;
    }
    if (good) {
      // a record starts where the one before it ended (empty lines in front of a header belong to its record)
      for (size_t k = 0; k < pc.starts.size(); k++) rec_start.push_back(k == 0 ? prev_end : pc.starts[k]);
      if (!pc.starts.empty()) prev_end = pc.stop;
      std::vector<size_t>().swap(pc.starts);
      merged++;
      if (merged == pieces.size()) { rec_start.push_back(size); all_merged = true; join_scanners(); }
    } else {
      join_scanners();                          // (they only fill their own pieces; let them finish)
      size_t p = prev_end;
      for (;;) { const size_t e = record_end(p); if (e == NONE) break; rec_start.push_back(p); p = e; }
      rec_start.push_back(size);
      all_merged = true;
    }
  }
  // records in the file: exact once everything is merged, else extrapolated from the part that is
  uint64_t estimated_records() {
    if (!prescanned) return 0;
    while (!all_merged && rec_start.size() < 2) merge_next_piece();
    if (all_merged) return rec_start.size() - 1;
    const size_t bytes = prev_end - pos;
    return bytes ? (uint64_t)((double)(rec_start.size()) * (double)(size - pos) / (double)bytes) : 0;
  }

  // ---- .gz files inflated by several threads (pargz.h): the text arrives in pieces, each with the positions of its newlines -
  // the record walk below costs a few loads per record (one thread; the memchr walk of the streamed path does 7 M records/s)
  // and every byte is copied once, from its piece into the block
  std::deque<std::shared_ptr<pargz::Text>> pq;   // pieces not handed out in full yet (blocks in the pipeline hold them too)
  uint64_t pq_base = 0;                  // position in the inflated text of pq[0]'s first byte
  uint64_t p_pos = 0;                    // ... of the next record
  bool p_eof = false;                    // no piece is left to fetch
  size_t nl_k = 0, nl_i = 0;             // the newline cursor: piece pq[nl_k], entry nl_i of its list
  bool p_fetch() {
    if (p_eof) return false;
    // (a piece the last block is through with goes back to the inflater, which uses its buffers again)
    std::shared_ptr<pargz::Reader> r = pz;
    std::shared_ptr<pargz::Text> t(new pargz::Text(), [r](pargz::Text *x) { r->recycle(std::move(*x)); delete x; });
    if (!pz->next_piece(*t)) { p_eof = true; return false; }
    pq.push_back(std::move(t));
    return true;
  }
  uint64_t p_end() const { uint64_t e = pq_base; for (const auto &t : pq) e += t->n; return e; }
  // is there a byte at position a (fetches pieces as needed)?
  bool p_have(uint64_t a) { while (a >= p_end()) if (!p_fetch()) return false; return true; }
  char p_ch(uint64_t a) const {
    uint64_t o = a - pq_base;
    for (const auto &t : pq) { if (o < t->n) return (char)t->buf.data()[o]; o -= t->n; }
    return 0;
  }
  // one past the first newline at or behind a (a < end of the text); at the end of the file a last line without one ends there
  uint64_t p_line_end(uint64_t a) {
    for (;;) {
      uint64_t start = pq_base;
      for (size_t k = 0; k < nl_k && k < pq.size(); k++) start += pq[k]->n;
      while (nl_k < pq.size()) {
        const pargz::Text &t = *pq[nl_k];
        while (nl_i < t.nl.size() && start + t.nl[nl_i] < a) nl_i++;
        if (nl_i < t.nl.size()) return start + t.nl[nl_i] + 1;
        start += t.n; nl_k++; nl_i = 0;
      }
      if (!p_fetch()) return p_end();
    }
  }
  // the end of the record that starts with the first non-empty line at or behind a; NONE: nothing but empty lines is left
  uint64_t p_record_end(uint64_t a) {
    uint64_t q = a, e;
    for (;;) {
      if (!p_have(q)) return NONE;
      e = p_line_end(q);
      if (e - q == 1 && p_ch(q) == '\n') { q = e; continue; }      // getline gave an empty line
      break;
    }
    if (first) {
      if (p_ch(q) == '@') fastq = true;
      else if (p_ch(q) != '>') die("Auto-detection of file type for file " + path + " failed.");
      first = false;
    }
    uint64_t r = e;
    if (fastq) {
      for (int lines = 0; lines < 3; lines++) {
        if (!p_have(r)) break;                                      // fewer lines at the end of the file are fine
        r = p_line_end(r);
      }
      return r;
    }
    for (;;) {
      if (!p_have(r)) return r;
      if (p_ch(r) == '>') return r;
      r = p_line_end(r);
    }
  }
  bool next_pieces(RawBlock &out, uint32_t want) {
    const uint64_t from = p_pos;
    uint64_t p = p_pos;
    while (out.n_records < want) {
      const uint64_t e = p_record_end(p);
      if (e == NONE) { p = p_end(); break; }
      p = e; out.n_records++;
    }
    p_pos = p;
    if (out.n_records == 0) { pq.clear(); return false; }
    // the block's text: stretches of the pieces (RawBlock::gather, run by the parser that takes the block, copies them once)
    out.segs.clear(); out.keep.clear();
    uint64_t start = pq_base;
    for (const auto &t : pq) {
      const uint64_t lo = std::max(from, start), hi = std::min(p, start + t->n);
      if (lo < hi) { out.segs.emplace_back(reinterpret_cast<const char *>(t->buf.data()) + (lo - start), (size_t)(hi - lo)); out.keep.push_back(t); }
      start += t->n;
    }
    out.text = nullptr; out.size = (size_t)(p - from); out.fastq = fastq;
    // pieces that lie in front of the next record are done with
    while (!pq.empty() && pq_base + pq.front()->n <= p_pos) {
      pq_base += pq.front()->n;
      pq.pop_front();
      if (nl_k > 0) nl_k--; else nl_i = 0;
    }
    return true;
  }

  // false when the file is exhausted and nothing was produced
  bool next(RawBlock &out, uint32_t want) {
    out.n_records = 0; out.own.clear();
    if (pz) return next_pieces(out, want);
    if (prescanned) {
      // (the end of a block is the start of the record behind it: one more start than records must be known)
      while (!all_merged && rec_start.size() <= next_rec + (size_t)want) merge_next_piece();
      const size_t n = rec_start.size() - 1;
      if (next_rec >= n) return false;
      const size_t last = std::min(n, next_rec + want);
      out.text = data + rec_start[next_rec]; out.size = rec_start[last] - rec_start[next_rec];
      out.n_records = (uint32_t)(last - next_rec); out.fastq = fastq;
      next_rec = last;
      return true;
    }
    size_t p = pos;
    while (out.n_records < want) {
      const size_t e = mapped ? record_end_scan(p) : record_end(p);
      if (e == NEED_MORE) {
        const size_t keep = p - pos;
        more();                               // (at end of file this only sets eof: the record is rescanned)
        p = pos + keep;
        continue;
      }
      if (e == NONE) { p = size; break; }     // nothing but empty lines left
      p = e; out.n_records++;
    }
    if (out.n_records == 0) { pos = p; return false; }
    if (mapped) { out.text = data + pos; out.size = p - pos; }
    else { out.own.assign(data + pos, data + p); out.text = out.own.data(); out.size = out.own.size(); }
    out.fastq = fastq;
    pos = p;
    return true;
  }
};

// ---------------------------------------------------------------------------------------------
// stage 2: parse blocks into the batch layout (parallel)
// ---------------------------------------------------------------------------------------------
// Allocator of the batch buffers (tens of megabytes each, gigabytes in flight): big blocks are 2 MB aligned with transparent
// huge pages asked for - first touch and release cost per page, and on the virtualised hosts this runs on a 4 KB page
// costs about as much as a 2 MB one - and elements are default-initialised (resize() before a memcpy writes nothing).
template <class T> struct HugeAlloc {
  typedef T value_type;
  HugeAlloc() = default;
  template <class U> HugeAlloc(const HugeAlloc<U> &) {}
  T *allocate(size_t n) {
    const size_t bytes = n * sizeof(T);
    void *p = nullptr;
    if (bytes >= (4u << 20) && !getenv("KAIJU_GPU_NO_HUGEPAGES")) {
      if (posix_memalign(&p, 2u << 20, bytes) != 0) throw std::bad_alloc();
      (void)madvise(p, bytes, MADV_HUGEPAGE);
    } else if (!(p = malloc(bytes ? bytes : 1))) throw std::bad_alloc();
    return static_cast<T *>(p);
  }
  void deallocate(T *p, size_t) { free(p); }
  template <class U> void construct(U *p) { ::new (static_cast<void *>(p)) U; }
  template <class U, class... A> void construct(U *p, A &&...a) { ::new (static_cast<void *>(p)) U(static_cast<A &&>(a)...); }
  template <class U> bool operator==(const HugeAlloc<U> &) const { return true; }
  template <class U> bool operator!=(const HugeAlloc<U> &) const { return false; }
};
template <class T> using HugeVec = std::vector<T, HugeAlloc<T>>;
typedef HugeVec<char> CharVec;

// bytes of column-7 text per call of the verbose entry point (KAIJU_GPU_VERBOSE_BUDGET: tests make it small)
static uint64_t verbose_text_budget() {
  static const uint64_t v = [] { const char *e = getenv("KAIJU_GPU_VERBOSE_BUDGET"); const long long x = e ? atoll(e) : 0; return x > 0 ? (uint64_t)x : 1ull << 30; }();
  return v;
}
struct Batch {
  CharVec seqs;
  HugeVec<uint64_t> off{0};
  CharVec names;                           // concatenated
  HugeVec<uint32_t> name_off{0};
  size_t n() const { return name_off.size() - 1; }
  // results
  HugeVec<kaiju_gpu_hit> hits;             // -v only
  HugeVec<kaiju_gpu_verbose> vrec;         // -v only: columns 6/7
  // -v only: the matched peptides (column 7), one string for the batch (kaiju_gpu_classify_batch_verbose_packed); read r's
  // starts at vpos[r].  On the DEVICE a read still has a row for the longest read of its call - up to 64 KB - so a batch is
  // classified in pieces of at most verbose_text_budget() bytes of rows: one 10-kb read in a batch of a million does not ask
  // for 64 GB
  CharVec vtext;
  HugeVec<uint64_t> vpos;
  const char *vtext_of(size_t r) const { return vtext.data() + vpos[r]; }
  HugeVec<kaiju_gpu_compact> compact;
  std::string text;
  void reset() {                           // empty, capacities kept
    seqs.clear(); off.assign(1, 0); names.clear(); name_off.assign(1, 0);
    hits.clear(); vrec.clear(); vtext.clear(); vpos.clear(); compact.clear(); text.clear();
  }
};

// batches are recycled: their buffers (hundreds of megabytes per batch) are faulted in once, not per batch
struct BatchPool {
  std::mutex m;
  std::condition_variable cv;
  std::vector<std::unique_ptr<Batch>> free_list;
  // Batches alive at once (KAIJU_GPU_MAX_BATCHES): the stages in front of the slowest one would otherwise run ahead by the
  // whole depth of the pipeline - some forty batches of 100 MB, every page of which is touched once and given back at
  // exit.  A thread asks for its batch BEFORE it takes the next block of input, so the batches alive are always the
  // ones with the lowest sequence numbers and the ordered queues behind never wait for one that cannot get a batch.
  size_t max_alive = 12, alive = 0;
  BatchPool() { if (const char *e = getenv("KAIJU_GPU_MAX_BATCHES")) max_alive = (size_t)std::max(3L, atol(e)); }
  std::unique_ptr<Batch> get() {
    std::unique_lock<std::mutex> lk(m);
    cv.wait(lk, [&] { return !free_list.empty() || alive < max_alive; });
    if (!free_list.empty()) { std::unique_ptr<Batch> b = std::move(free_list.back()); free_list.pop_back(); return b; }
    alive++;
    lk.unlock();
    return std::unique_ptr<Batch>(new Batch());
  }
  void put(std::unique_ptr<Batch> b) {
    b->reset();
    { std::lock_guard<std::mutex> lk(m); free_list.push_back(std::move(b)); }
    cv.notify_one();
  }
};

// CPU time per pipeline stage (KAIJU_GPU_STAGE_TIMES=1 prints it): where the host side spends its time
std::atomic<uint64_t> g_ns_read{0}, g_ns_parse{0}, g_ns_gpu{0}, g_ns_format{0}, g_ns_write{0};
struct StageTimer {
  std::atomic<uint64_t> &acc;
  timespec t0;
  explicit StageTimer(std::atomic<uint64_t> &a) : acc(a) { clock_gettime(CLOCK_THREAD_CPUTIME_ID, &t0); }
  ~StageTimer() {
    timespec t1; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &t1);
    acc += (uint64_t)((t1.tv_sec - t0.tv_sec) * 1000000000ll + (t1.tv_nsec - t0.tv_nsec));
  }
};

// KAIJU_GPU_STAGE_TIMES=1: wall-clock marks since the start of main(), on stderr (where the end-to-end time goes)
double g_wall0 = 0;
inline double wall_now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
bool g_marks = false;
inline void wall_mark(const char *what, long long k = -1) {
  if (!g_marks) return;
  if (k >= 0) fprintf(stderr, "[wall %8.1f ms] %s %lld\n", (wall_now() - g_wall0) * 1e3, what, k);
  else fprintf(stderr, "[wall %8.1f ms] %s\n", (wall_now() - g_wall0) * 1e3, what);
}

// kaijup keeps the whole header line as the read name (kaijup.cpp:249-262 has no suffix cutting)
bool g_keep_names = false;

inline void append_stripped(CharVec &dst, const char *s, size_t n) {   // strip(), util.cpp:25-32
  const size_t old = dst.size();
  dst.resize(old + n);
  char *d = dst.data() + old;
  // almost every line consists of letters only: test that eight bytes at a time, copy wholesale
  bool all = true;
  size_t i = 0;
  for (; i + 8 <= n; i += 8) {
    uint64_t w; memcpy(&w, s + i, 8);
    const uint64_t H = 0x8080808080808080ull;
    // per byte (high bit of the input clear): (c | 0x20 | 0x80) - 'a' = 0x80 + ((c | 32) - 'a'), no borrow between bytes;
    // a letter gives 0x80..0x99: high bit set, and the low seven bits + 0x66 stay below 0x80
    const uint64_t l = (w | 0x2020202020202020ull | H) - 0x6161616161616161ull;
    if ((w & H) || (l & H) != H || (((l & ~H) + 0x6666666666666666ull) & H)) { all = false; break; }
  }
  if (all) for (; i < n; i++) { const unsigned char c = (unsigned char)(s[i] | 32); if (c < 'a' || c > 'z') { all = false; break; } }
  if (all) { memcpy(d, s, n); return; }
  size_t k = 0;
  for (i = 0; i < n; i++) { const char c = s[i]; if ((c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z')) d[k++] = c; }
  dst.resize(old + k);
}

// walks the records of a block exactly like the reference's loop (kaiju.cpp:288-331 / :333-386)
struct BlockCursor {
  const RawBlock &b;
  size_t p = 0;
  explicit BlockCursor(const RawBlock &blk) : b(blk) {}
  bool line(const char *&s, size_t &n) {   // next line without its '\n'
    if (p >= b.size) return false;
    const char *base = b.text;
    const void *nl = memchr(base + p, '\n', b.size - p);
    const size_t e = nl ? (size_t)((const char *)nl - base) : b.size;
    s = base + p; n = e - p; p = e + 1;
    return true;
  }
  // name (cut at the first of " /\t\r") and stripped sequence of the next record
  bool next(const char *&name, size_t &name_len, CharVec &seqs) {
    const char *s; size_t n;
    do { if (!line(s, n)) return false; } while (n == 0);
    s++; n--;                               // erase(0,1)
    size_t cut = 0;
    if (g_keep_names) cut = n;
    else while (cut < n && s[cut] != ' ' && s[cut] != '/' && s[cut] != '\t' && s[cut] != '\r') cut++;
    name = s; name_len = cut;
    if (b.fastq) {
      if (line(s, n)) append_stripped(seqs, s, n);
      line(s, n); line(s, n);
    } else {
      while (p < b.size && b.text[p] != '>') { line(s, n); append_stripped(seqs, s, n); }
    }
    return true;
  }
};

void parse_blocks(const RawBlock &b1, const RawBlock *b2, const std::string &fn1, const std::string &fn2, Batch &out) {
  out.seqs.reserve(b1.size / 2 + (b2 ? b2->size / 2 : 0));
  BlockCursor c1(b1);
  std::unique_ptr<BlockCursor> c2;
  if (b2) c2.reset(new BlockCursor(*b2));
  const char *nm; size_t nl;
  while (c1.next(nm, nl, out.seqs)) {
    out.off.push_back(out.seqs.size());
    if (b2) {
      const char *nm2; size_t nl2;
      if (!c2->next(nm2, nl2, out.seqs)) die("File " + fn1 + " contains more reads then file " + fn2);
      if (nl != nl2 || memcmp(nm, nm2, nl) != 0)
        die("Read names are not identical between the two input files. Probably reads are not in the same order in both files.");
    }
    out.off.push_back(out.seqs.size());
    out.names.insert(out.names.end(), nm, nm + nl);
    out.name_off.push_back((uint32_t)out.names.size());
  }
}

// ---------------------------------------------------------------------------------------------
// ordered hand-over between the stages: items carry their sequence number
// ---------------------------------------------------------------------------------------------
template <class T>
struct OrderedQueue {
  std::mutex m;
  std::condition_variable cv;
  std::map<uint64_t, T> items;
  uint64_t end = ~0ull;                     // number of items of the whole run, once known
  size_t cap;
  uint64_t taken = 0;                       // items handed out so far (unordered consumers)
  explicit OrderedQueue(size_t c) : cap(c) {}
  void put(uint64_t seq, T v) {
    std::unique_lock<std::mutex> lk(m);
    // the item everyone is waiting for must never be blocked by the capacity
    cv.wait(lk, [&] { return items.size() < cap || items.empty() || seq < items.begin()->first; });
    items.emplace(seq, std::move(v));
    cv.notify_all();
  }
  void finish(uint64_t n_items) { std::lock_guard<std::mutex> lk(m); end = n_items; cv.notify_all(); }
  // in order: blocks until item `seq` is there; false when seq >= end
  bool take(uint64_t seq, T &v) {
    std::unique_lock<std::mutex> lk(m);
    cv.wait(lk, [&] { return items.count(seq) || seq >= end; });
    if (!items.count(seq)) return false;
    v = std::move(items[seq]); items.erase(seq);
    cv.notify_all();
    return true;
  }
  // any order (parallel consumers): the smallest item present
  bool take_any(uint64_t &seq, T &v) {
    std::unique_lock<std::mutex> lk(m);
    cv.wait(lk, [&] { return !items.empty() || taken >= end; });
    if (items.empty()) return false;
    auto it = items.begin();
    seq = it->first; v = std::move(it->second); items.erase(it); taken++;
    cv.notify_all();
    return true;
  }
};

inline void append_u64(std::string &s, unsigned long long v) {
  char tmp[24]; int k = 0;
  do { tmp[k++] = (char)('0' + v % 10); v /= 10; } while (v);
  while (k) s.push_back(tmp[--k]);
}

}  // namespace

// kaijup's choice between "U<TAB>name<TAB>0" and "U<TAB>name" (ConsumerThreadp.cpp:22-71): does the protein read yield
// any fragment, i.e. a run of at least -m amino-acid letters that (Greedy) scores at least -s on the BLOSUM62 diagonal?
bool protein_has_fragment(const char *s, uint64_t len, const kaiju_gpu_params &p) {
  static const int8_t diag[26] = {4, 0, 9, 6, 5, 6, 6, 8, 4, 0, 5, 4, 5, 6, 0, 7, 5, 5, 4, 5, 0, 4, 11, 0, 7, 0};   // A..Z, 0: no amino acid
  uint64_t run = 0; uint32_t score = 0;
  for (uint64_t i = 0; i <= len; i++) {
    int d = 0;
    if (i < len) { const int c = s[i] & ~32; if (c >= 'A' && c <= 'Z' && ((s[i] >= 'a' && s[i] <= 'z') || (s[i] >= 'A' && s[i] <= 'Z'))) d = diag[c - 'A']; }
    if (d) { run++; score += (uint32_t)d; continue; }
    if (run >= p.min_fragment_length && (p.mode == 0 || score >= p.min_score)) return true;
    run = 0; score = 0;
  }
  return false;
}

int main(int argc, char **argv) {
  g_wall0 = wall_now();
  g_marks = getenv("KAIJU_GPU_STAGE_TIMES") != nullptr;
  signal(SIGPIPE, SIG_IGN);                   // a closed output pipe is a write error, not the end of the process
  // batches come and go by the hundred megabytes: keep that memory in the heap instead of mapping and unmapping (and
  // page-faulting) it for every batch
  mallopt(M_MMAP_THRESHOLD, 1 << 30);
  mallopt(M_TRIM_THRESHOLD, 1 << 30);
  std::atomic<uint64_t> inexact_batches{0}, inexact_reads{0};
  kaiju_gpu_params params;
  kaiju_gpu_default_params(&params, 1);
  std::string nodes_fn, fmi_fn, in1_fn, in2_fn, out_fn;
  bool verbose = false, paired = false, protein = false;
  int c;
  while ((c = getopt(argc, argv, "a:hdpxXvn:m:e:E:l:t:f:i:j:s:z:o:")) != -1) {
    switch (c) {
      case 'a':
        if (std::string(optarg) == "mem") { params.mode = 0; params.use_evalue = 0; }
        else if (std::string(optarg) == "greedy") params.mode = 1;
        else { fprintf(stderr, "-a must be a valid mode.\n"); usage(argv[0]); }
        break;
      case 'h': usage(argv[0]); break;
      case 'd': break;
      case 'v': verbose = true; break;
      case 'p': protein = true; break;
      case 'x': params.seg = 1; break;
      case 'X': params.seg = 0; break;
      case 'o': out_fn = optarg; break;
      case 'f': fmi_fn = optarg; break;
      case 't': nodes_fn = optarg; break;
      case 'i': in1_fn = optarg; break;
      case 'j': in2_fn = optarg; paired = true; break;
      case 'l': { int v = atoi(optarg); if (v < 7) die("Seed length must be >= 7."); params.seed_length = (uint32_t)v; break; }
      case 's': { int v = atoi(optarg); if (v <= 0) die("Min Score (-s) must be greater than 0."); params.min_score = (uint32_t)v; break; }
      case 'm': { int v = atoi(optarg); if (v <= 0) die("Min fragment length (-m) must be greater than 0."); params.min_fragment_length = (uint32_t)v; break; }
      case 'e': { int v = atoi(optarg); if (v < 0) die("Number of mismatches must be >= 0."); params.mismatches = (uint32_t)v; break; }
      case 'E': { double v = atof(optarg); if (v <= 0.0) die("E-value threshold must be greater than 0."); params.min_evalue = v; break; }
      case 'z': { if (atoi(optarg) <= 0) die("Number of threads (-z) must be greater than 0."); break; }
      case 'n': break;
      default: usage(argv[0]);
    }
  }
  if (getenv("KAIJU_GPU_PARSE_ONLY")) { if (nodes_fn.empty()) nodes_fn = "-"; if (fmi_fn.empty()) fmi_fn = "-"; }
  // which program this is (kaiju, kaiju-multi, kaijux) is decided by the name it was started under
  std::string prog = argv[0];
  if (prog.find('/') != std::string::npos) prog = prog.substr(prog.rfind('/') + 1);
  if ((prog.find("kaijux") != std::string::npos || prog.find("kaijup") != std::string::npos) && nodes_fn.empty()) nodes_fn = "-";
  if (nodes_fn.empty()) { fprintf(stderr, "Error: Please specify the location of the nodes.dmp file, using the -t option.\n\n"); usage(argv[0]); }
  if (fmi_fn.empty()) { fprintf(stderr, "Error: Please specify the location of the FMI file, using the -f option.\n\n"); usage(argv[0]); }
  if (in1_fn.empty()) { fprintf(stderr, "Error: Please specify the location of the input file, using the -i option.\n\n"); usage(argv[0]); }
  // kaijup (kaijup.cpp, ConsumerThreadp.cpp): kaijux for protein reads
  const bool pmode = prog.find("kaijup") != std::string::npos;
  if (pmode) { protein = true; g_keep_names = true; if (nodes_fn.empty()) nodes_fn = "-"; }
  if (paired && protein) { fprintf(stderr, "Error: Protein input only supports one input file.\n\n"); usage(argv[0]); }
  params.input_is_protein = protein ? 1 : 0;
  if (params.use_evalue && params.mode == 0) die("E-value calculation is only possible in Greedy run mode.");

  // kaijux (kaijux.cpp, ConsumerThreadx.cpp): the same search, but a read is reported with the database sequences it
  // matches (no taxonomy, no nodes.dmp): "C<TAB>name<TAB>score<TAB>seqname,...<TAB>[peptides with -v]" / "U<TAB>name"
  const bool xmode = prog.find("kaijux") != std::string::npos || prog.find("kaijup") != std::string::npos;
  if (xmode && nodes_fn.empty()) nodes_fn = "-";
  // developer/test switch: run the ingest stages only and print "name<TAB>mate1<TAB>mate2" per read
  const bool parse_only = getenv("KAIJU_GPU_PARSE_ONLY") != nullptr;
  if (verbose) fprintf(stderr, "%s Reading database\n", now().c_str());
  // Which GPUs: KAIJU_GPU_DEVICE=<n> (one, default 0) or KAIJU_GPU_DEVICES=<n,n,...|all>: the index is replicated on each of
  // them (parsed and packed once), input block b goes to context b mod (2 x GPUs) - SURVEY 8e's "block b to GPU b mod N" -
  // and the formatter stage puts the lines back in input order.
  std::vector<int> devices;
  if (const char *e = getenv("KAIJU_GPU_DEVICES")) {
    if (!strcmp(e, "all")) { const int nd = kaiju_gpu_device_count(); for (int d = 0; d < nd; d++) devices.push_back(d); }
    else {
      // a comma separated list of device numbers: every entry a number of an existing device, none twice (a replica of the
      // index per entry: "0,,1" or "a,b" must not quietly become GPU 0 more than once); KAIJU_GPU_DEVICES_ALLOW_REPEAT=1
      // lets tests put several replicas on one GPU
      const int nd = parse_only ? 1 << 20 : kaiju_gpu_device_count();
      for (const char *q = e; ; ) {
        char *end = nullptr;
        const long v = strtol(q, &end, 10);
        if (end == q || (*end && *end != ',')) die(std::string("KAIJU_GPU_DEVICES: not a list of device numbers: ") + e);
        if (v < 0 || v >= nd) die(std::string("KAIJU_GPU_DEVICES: no such device: ") + std::to_string(v));
        if (!getenv("KAIJU_GPU_DEVICES_ALLOW_REPEAT"))
          for (int d : devices) if (d == (int)v) die(std::string("KAIJU_GPU_DEVICES: device listed twice: ") + std::to_string(v));
        devices.push_back((int)v);
        if (!*end) break;
        q = end + 1;
      }
    }
  }
  if (devices.empty()) devices.push_back(getenv("KAIJU_GPU_DEVICE") ? atoi(getenv("KAIJU_GPU_DEVICE")) : 0);
  const int n_dev = (int)devices.size();
  // nodes.dmp is parsed while the index loads
  kaiju_taxonomy *tax = nullptr;
  std::vector<kaiju_gpu_index *> indexes((size_t)n_dev, nullptr);
  kaiju_gpu_index *index = nullptr;           // the first replica: sequence names, info
  kaiju_gpu_index_info info;
  memset(&info, 0, sizeof info);
  std::vector<kaiju_gpu_taxonomy *> dtaxes((size_t)n_dev, nullptr);
  const int n_ctx = 2 * n_dev;              // per GPU two: copies of one batch overlap with kernels of the other
  std::vector<kaiju_gpu_ctx *> ctx((size_t)n_ctx, nullptr);
  // The index loads (and nodes.dmp is parsed, and the contexts are created) on a thread of its own while the input is
  // already being read and parsed: only the GPU stage of the pipeline waits for it.
  struct Ready { std::mutex m; std::condition_variable cv; bool done = false; } gpu_ready;
  std::thread loader;
  if (!parse_only)
    loader = std::thread([&] {
    int tax_rc = 0, rc = 0;
    std::thread tax_loader([&] { if (!xmode) tax_rc = kaiju_taxonomy_load(nodes_fn.c_str(), &tax); });
    // a device image next to the index (<file>.kjimg, written by `python -m kaiju_amd.mkimage` or KAIJU_GPU_WRITE_IMAGE=1)
    // that is not older than it loads without parsing and packing (kaiju_gpu_index_write_image)
    std::string load_fn = fmi_fn;
    {
      struct stat sa, sb;
      const std::string img = fmi_fn + ".kjimg";
      // (usable: not older than the index AND made from a .fmi of this very size - time stamps alone survive cp -p / rsync -t)
      uint64_t src_bytes = 0;
      if (stat(fmi_fn.c_str(), &sa) == 0 && stat(img.c_str(), &sb) == 0 && sb.st_mtime >= sa.st_mtime && !getenv("KAIJU_GPU_NO_IMAGE") &&
          kaiju_gpu_index_image_source_bytes(img.c_str(), &src_bytes) == 0 && src_bytes == (uint64_t)sa.st_size) load_fn = img;
      else if (getenv("KAIJU_GPU_WRITE_IMAGE") && kaiju_gpu_index_write_image(fmi_fn.c_str(), img.c_str()) == 0) load_fn = img;
    }
    rc = kaiju_gpu_index_load_devices(load_fn.c_str(), devices.data(), n_dev, xmode ? KAIJU_GPU_IDS_SEQUENCE : KAIJU_GPU_IDS_TAXON, indexes.data());
    index = indexes[0];
    wall_mark("index on the device");
    tax_loader.join();
    wall_mark("nodes.dmp parsed");
    if (tax_rc != 0) die("Could not open file " + nodes_fn);
    if (rc != 0) die(std::string("Could not load ") + fmi_fn + ": " + kaiju_gpu_strerror(rc) + " (" + kaiju_gpu_last_error() + ")");
    kaiju_gpu_index_get_info(index, &info);
    if (info.warnings && verbose) fprintf(stderr, " Warning: the index triggers a latent bug of the reference (flags %u)\n", info.warnings);
    if (!verbose && !xmode)
      for (int d = 0; d < n_dev; d++) {
        rc = kaiju_gpu_taxonomy_upload(tax, devices[(size_t)d], &dtaxes[(size_t)d]);
        if (rc != 0) die(std::string("kaiju_gpu_taxonomy_upload: ") + kaiju_gpu_strerror(rc) + " (" + kaiju_gpu_last_error() + ")");
      }
    for (int k = 0; k < n_ctx; k++) {          // contexts 2d and 2d + 1 live on GPU d
      rc = kaiju_gpu_create(&ctx[(size_t)k], indexes[(size_t)(k / 2)], &params);
      if (rc != 0) die(std::string("kaiju_gpu_create: ") + kaiju_gpu_strerror(rc) + " (" + kaiju_gpu_last_error() + ")");
    }
    wall_mark("taxonomy uploaded, contexts created");
    { std::lock_guard<std::mutex> lk(gpu_ready.m); gpu_ready.done = true; }
    gpu_ready.cv.notify_all();
  });
  auto wait_for_gpu = [&] { std::unique_lock<std::mutex> lk(gpu_ready.m); gpu_ready.cv.wait(lk, [&] { return gpu_ready.done; }); };

  // kaiju-multi (kaiju-multi.cpp:221-334): comma separated lists of input / output files, one index load
  const bool multi = prog.find("multi") != std::string::npos;
  auto split_list = [](const std::string &v) {
    std::vector<std::string> out;
    size_t begin = 0, pos;
    while ((pos = v.find(',', begin)) != std::string::npos) { if (pos > begin) out.push_back(v.substr(begin, pos - begin)); begin = pos + 1; }
    if (begin < v.size()) out.push_back(v.substr(begin));
    return out;
  };
  std::vector<std::string> list1{in1_fn}, list2, list_out;
  if (paired) list2.push_back(in2_fn);
  if (!out_fn.empty()) list_out.push_back(out_fn);
  if (multi) {
    list1 = split_list(in1_fn); list2 = split_list(in2_fn); list_out = split_list(out_fn);
    if ((!out_fn.empty() && ((paired && (list1.size() != list2.size() || list1.size() != list_out.size())) ||
                             (!paired && list1.size() != list_out.size()))) ||
        (out_fn.empty() && paired && list1.size() != list2.size()))
      die("Length of input/output file lists differs");
    for (const auto &f : list1) { FILE *t = fopen(f.c_str(), "r"); if (!t) die("Could not open file " + f); fclose(t); }
    for (const auto &f : list2) { FILE *t = fopen(f.c_str(), "r"); if (!t) die("Could not open file " + f); fclose(t); }
  }

  auto run_sample = [&](const std::string &in1_fn, const std::string &in2_fn, const std::string &out_fn) {
    FILE *out = stdout;
    if (!out_fn.empty()) {
      out = fopen(out_fn.c_str(), "w");
      if (!out) die("Could not open file " + out_fn + " for writing");
    }
    // (once per stream and before the first write: kaiju-multi comes back here with stdout for every sample)
    static bool stdout_buffered = false;
    if (out != stdout || !stdout_buffered) setvbuf(out, nullptr, _IOFBF, 1 << 22);
    if (out == stdout) stdout_buffered = true;

    uint32_t batch_reads = 500000;
    if (const char *e = getenv("KAIJU_GPU_BATCH")) batch_reads = (uint32_t)std::max(1L, atol(e));
    unsigned n_workers = std::max(2u, std::min(16u, std::thread::hardware_concurrency() / 2));
    if (const char *e = getenv("KAIJU_GPU_HOST_THREADS")) n_workers = (unsigned)std::max(1, atoi(e));
    if (verbose) fprintf(stderr, "%s Start classification on GPU %d%s\n", now().c_str(), devices[0], n_dev > 1 ? " and further ones (KAIJU_GPU_DEVICES)" : "");

    struct RawPair { std::unique_ptr<RawBlock> a, b; };
    // (one pool for all samples of kaiju-multi, never torn down: giving gigabytes of batch buffers back page by page
    // when the last sample is done would only delay the end of the process)
    static BatchPool &pool = *new BatchPool();
    OrderedQueue<RawPair> q_raw(8);
    OrderedQueue<std::unique_ptr<Batch>> q_parsed(6), q_done(6), q_text(8);

    // stage 1: readers (file 2 is read by its own thread, blocks are paired up here)
    std::thread reader([&] {
      BlockReader r1(in1_fn);
      if (!r1.ok) die("Could not open file " + in1_fn);
      { StageTimer tm(g_ns_read); r1.prescan(n_workers); }
      wall_mark("input mapped and prescanned");
      // reads per batch: small enough that a sample fills the pipeline (sixteen batches), large enough that the persistent
      // lanes of the search kernels get more than a read or two each
      if (!getenv("KAIJU_GPU_BATCH") && r1.prescanned) {
        const uint64_t n_rec = r1.estimated_records();
        batch_reads = (uint32_t)std::min<uint64_t>(1000000, std::max<uint64_t>(250000, (n_rec / 16 + 49999) / 50000 * 50000));
      }
      std::unique_ptr<BlockReader> r2;
      OrderedQueue<std::unique_ptr<RawBlock>> q2(4);
      std::thread reader2;
      if (paired) {
        r2.reset(new BlockReader(in2_fn));
        if (!r2->ok) die("Could not open file " + in2_fn);
        r2->prescan(n_workers);
        reader2 = std::thread([&] {
          uint64_t k = 0;
          for (;;) {
            std::unique_ptr<RawBlock> b(new RawBlock());
            if (!r2->next(*b, batch_reads)) break;
            q2.put(k++, std::move(b));
          }
          q2.finish(k);
        });
      }
      uint64_t seq = 0;
      for (;;) {
        RawPair pr;
        pr.a.reset(new RawBlock());
        { StageTimer tm(g_ns_read); if (!r1.next(*pr.a, batch_reads)) break; }
        if (paired) {
          if (!q2.take(seq, pr.b)) die("File " + in1_fn + " contains more reads then file " + in2_fn);
          if (pr.b->n_records < pr.a->n_records) die("File " + in1_fn + " contains more reads then file " + in2_fn);
        }
        q_raw.put(seq++, std::move(pr));
      }
      if (paired) {
        std::unique_ptr<RawBlock> extra;
        if (q2.take(seq, extra)) fprintf(stderr, "Warning: File %s has more reads then file %s\n", in2_fn.c_str(), in1_fn.c_str());
        // drain so that reader 2 can finish
        for (uint64_t k = seq + 1; q2.take(k, extra); k++) {}
        reader2.join();
      }
      q_raw.finish(seq);
      q_parsed.finish(seq); q_done.finish(seq); q_text.finish(seq);
    });

    // stage 2: parsers
    std::vector<std::thread> parsers;
    for (unsigned w = 0; w < n_workers; w++)
      parsers.emplace_back([&] {
        uint64_t seq; RawPair pr;
        for (;;) {
          std::unique_ptr<Batch> b = pool.get();               // (first the batch, then the block: see BatchPool)
          if (!q_raw.take_any(seq, pr)) { pool.put(std::move(b)); break; }
          { StageTimer tm(g_ns_parse); pr.a->gather(); if (pr.b) pr.b->gather(); parse_blocks(*pr.a, pr.b.get(), in1_fn, in2_fn, *b); }
          if (paired && pr.b->n_records > pr.a->n_records)
            fprintf(stderr, "Warning: File %s has more reads then file %s\n", in2_fn.c_str(), in1_fn.c_str());
          q_parsed.put(seq, std::move(b));
        }
      });

    // stage 3: the GPU, one thread per context, batches alternate between them
    std::vector<std::thread> gpu_threads;
    for (int k = 0; k < n_ctx; k++)
      gpu_threads.emplace_back([&, k] {
        std::unique_ptr<Batch> b;
        for (uint64_t seq = (uint64_t)k; q_parsed.take(seq, b); seq += n_ctx) {
          const uint32_t n = (uint32_t)b->n();
          int r = 0;
          if (parse_only) { q_done.put(seq, std::move(b)); continue; }
          wait_for_gpu();
          StageTimer tm(g_ns_gpu);
          wall_mark("gpu call begins, batch", (long long)seq);
          uint64_t piece_error_flags = 0;
          if (verbose) {
            b->hits.resize(n);
            b->vrec.resize(n);
            b->vpos.resize(n);
            std::vector<uint64_t> poff;
            piece_error_flags = 0;
            for (uint32_t lo = 0; lo < n && r == 0;) {
              // the longest piece from read lo on whose text fits the budget (at least one read)
              uint64_t maxpair = 0;
              uint32_t hi = lo;
              while (hi < n) {
                const uint64_t mp = std::max<uint64_t>(maxpair, b->off[2 * (size_t)hi + 2] - b->off[2 * (size_t)hi]);
                const uint32_t st = kaiju_gpu_verbose_text_stride((uint32_t)mp, protein ? 1 : 0);
                if (hi > lo && (uint64_t)(hi - lo + 1) * st > verbose_text_budget()) break;
                maxpair = mp; hi++;
              }
              const uint32_t pn = hi - lo;
              const uint64_t base = b->off[2 * (size_t)lo];
              const uint64_t *po = b->off.data() + 2 * (size_t)lo;
              if (base != 0) {                   // (the entry point wants off[0] == 0)
                poff.resize(2 * (size_t)pn + 1);
                for (size_t q = 0; q < poff.size(); q++) poff[q] = po[q] - base;
                po = poff.data();
              }
              const char *ptext = nullptr;
              uint64_t pbytes = 0;
              r = kaiju_gpu_classify_batch_verbose_packed(ctx[k], b->seqs.data() + base, po, pn, paired ? 1 : 0, b->hits.data() + lo,
                                                          b->vrec.data() + lo, b->vpos.data() + lo, &ptext, &pbytes);
              if (r == 0) {
                // (the string belongs to the context until its next call: the piece's text moves behind the batch's)
                const uint64_t at = b->vtext.size();
                b->vtext.insert(b->vtext.end(), ptext, ptext + pbytes);
                if (at) for (uint32_t q = lo; q < hi; q++) b->vpos[q] += at;
              }
              // (the device counters are zeroed per launch: the flags of every piece count, not only the last one's)
              { kaiju_gpu_stats ps; if (r == 0 && kaiju_gpu_get_stats(ctx[k], &ps) == 0) piece_error_flags |= ps.error_flags; }
              lo = hi;
            }
          } else if (xmode) {
            b->hits.resize(n);
            r = kaiju_gpu_classify_batch(ctx[k], b->seqs.data(), b->off.data(), n, paired ? 1 : 0, b->hits.data());
          } else {
            b->compact.resize(n);
            r = kaiju_gpu_classify_batch_compact(ctx[k], dtaxes[(size_t)(k / 2)], b->seqs.data(), b->off.data(), n, paired ? 1 : 0, b->compact.data());
          }
          if (r != 0) die(std::string("classification failed: ") + kaiju_gpu_strerror(r) + " (" + kaiju_gpu_last_error() + ")");
          // reads for which a capacity bound of the kernels was exceeded (KAIJU_HIT_INEXACT, kaiju_gpu_stats.error_flags)
          {
            kaiju_gpu_stats st;
            if ((kaiju_gpu_get_stats(ctx[k], &st) == 0 && st.error_flags) || piece_error_flags) inexact_batches++;
            uint64_t ni = 0;
            if (!b->hits.empty()) { for (uint32_t q = 0; q < n; q++) ni += (b->hits[q].flags & KAIJU_HIT_INEXACT) ? 1 : 0; }
            else for (uint32_t q = 0; q < n; q++) ni += (b->compact[q].info & KAIJU_HIT_INEXACT) ? 1 : 0;
            if (ni) inexact_reads += ni;
          }

          wall_mark("gpu call done, batch", (long long)seq);
          q_done.put(seq, std::move(b));
        }
      });

    // stage 4: E-value gate, (LCA,) C/U decision and text, in parallel
    std::vector<std::thread> formatters;
    for (unsigned w = 0; w < n_workers; w++)
      formatters.emplace_back([&] {
        uint64_t seq; std::unique_ptr<Batch> b;
        std::vector<kaiju_result> res;
        while (q_done.take_any(seq, b)) {
          StageTimer tm(g_ns_format);
          const uint32_t n = (uint32_t)b->n();
          if (parse_only) {
            std::string &text = b->text;
            for (uint32_t r = 0; r < n; r++) {
              text.append(b->names.data() + b->name_off[r], b->name_off[r + 1] - b->name_off[r]); text += '\t';
              text.append(b->seqs.data() + b->off[2 * r], b->off[2 * r + 1] - b->off[2 * r]); text += '\t';
              text.append(b->seqs.data() + b->off[2 * r + 1], b->off[2 * r + 2] - b->off[2 * r + 1]);
              // (kaijup: a fourth column, the decision behind its two kinds of U lines)
              if (pmode) { text += '\t'; text += protein_has_fragment(b->seqs.data() + b->off[2 * r], b->off[2 * r + 1] - b->off[2 * r], params) ? '1' : '0'; }
              text += '\n';
            }
            q_text.put(seq, std::move(b));
            continue;
          }
          res.resize(n);
          if (xmode) {
            // E-value gate and C/U decision as for kaiju; the ids are sequence numbers: names in sequence order
            // (the reference iterates a std::set<char *> of the names, i.e. in the order they were allocated)
            b->compact.resize(n);
            for (uint32_t r = 0; r < n; r++) { b->compact[r].lca = b->hits[r].n_ids ? 1 : 0; b->compact[r].best = b->hits[r].best; b->compact[r].info = b->hits[r].n_ids; }
            kaiju_finalize_compact(&params, info.db_length, b->compact.data(), b->off.data(), n, paired ? 1 : 0, res.data());
            std::string &text = b->text;
            text.clear();
            for (uint32_t r = 0; r < n; r++) {
              const char *nm = b->names.data() + b->name_off[r];
              const size_t nl = b->name_off[r + 1] - b->name_off[r];
              if (!res[r].classified) {
                // reads below the length gate get the three-column line of kaiju (ConsumerThreadx.cpp:202-207), others "U<TAB>name"
                const uint64_t l1 = b->off[2 * (size_t)r + 1] - b->off[2 * (size_t)r], l2 = b->off[2 * (size_t)r + 2] - b->off[2 * (size_t)r + 1];
                const uint64_t m3 = 3ull * params.min_fragment_length;
                bool gated = paired ? (l1 < m3 && l2 < m3) : (l1 < m3);
                // kaijup: the same line for reads shorter than -m and for reads without any fragment (ConsumerThreadp.cpp:17-21,67-71)
                if (pmode) gated = l1 < params.min_fragment_length || !protein_has_fragment(b->seqs.data() + b->off[2 * (size_t)r], l1, params);
                text += "U\t"; text.append(nm, nl); text += gated ? "\t0\n" : "\n";
                continue;
              }
              text += "C\t"; text.append(nm, nl); text += '\t'; append_u64(text, res[r].best); text += '\t';
              uint64_t ids[KAIJU_GPU_MAX_IDS];
              const uint32_t k = b->hits[r].n_ids;
              for (uint32_t q = 0; q < k; q++) ids[q] = b->hits[r].taxid[q];
              std::sort(ids, ids + k);
              for (uint32_t q = 0; q < k; q++) { const char *sn = kaiju_gpu_index_seq_name(index, (uint32_t)ids[q]); if (sn) text += sn; text += ','; }
              text += '\t';
              if (verbose) {
                text.append(b->vtext_of(r), b->vrec[r].text_len);
                if (b->vrec[r].truncated) { fprintf(stderr, "Warning: matched peptides of read %.*s truncated\n", (int)nl, nm); inexact_reads++; }
              }
              text += '\n';
            }
            q_text.put(seq, std::move(b));
            continue;
          }
          if (verbose) kaiju_finalize_hits(tax, &params, info.db_length, b->hits.data(), b->off.data(), n, paired ? 1 : 0, res.data());
          else kaiju_finalize_compact(&params, info.db_length, b->compact.data(), b->off.data(), n, paired ? 1 : 0, res.data());
          std::string &text = b->text;
          text.clear();
          text.reserve((size_t)n * 24 + b->names.size());
          for (uint32_t r = 0; r < n; r++) {
            const char *nm = b->names.data() + b->name_off[r];
            const size_t nl = b->name_off[r + 1] - b->name_off[r];
            if (res[r].classified) {
              text += "C\t"; text.append(nm, nl); text += '\t';
              append_u64(text, res[r].taxon);
              if (verbose) {
                text += '\t'; append_u64(text, res[r].best); text += '\t';
                uint64_t ids[KAIJU_GPU_MAX_IDS];
                const uint32_t k = b->hits[r].n_ids;
                for (uint32_t q = 0; q < k; q++) ids[q] = b->hits[r].taxid[q];
                std::sort(ids, ids + k);                       // std::set iteration order, :527-536
                for (uint32_t q = 0; q < k; q++) { append_u64(text, ids[q]); text += ','; }
                // column 6: the set of accessions (name up to its last '_'), column 7: the matched peptides
                text += '\t';
                const kaiju_gpu_verbose &v = b->vrec[r];
                std::string acc[KAIJU_GPU_MAX_ACC];
                uint32_t na = 0;
                for (uint32_t q = 0; q < v.n_acc; q++) {
                  const char *nm2 = kaiju_gpu_index_seq_name(index, v.acc_iseq[q]);
                  const char *us = nm2 ? strrchr(nm2, '_') : nullptr;
                  if (us) acc[na++].assign(nm2, (size_t)(us - nm2));
                }
                std::sort(acc, acc + na);
                for (uint32_t q = 0; q < na; q++) if (q == 0 || acc[q] != acc[q - 1]) { text += acc[q]; text += ','; }
                text += '\t';
                text.append(b->vtext_of(r), v.text_len);
                if (v.truncated) { fprintf(stderr, "Warning: matched peptides of read %.*s truncated\n", (int)nl, nm); inexact_reads++; }
              }
              text += '\n';
            } else { text += "U\t"; text.append(nm, nl); text += "\t0\n"; }
          }
          q_text.put(seq, std::move(b));
        }
      });

    // stage 5: write in input order
    {
      std::unique_ptr<Batch> b;
      for (uint64_t seq = 0; q_text.take(seq, b); seq++) {
        StageTimer tm(g_ns_write);
        if (fwrite(b->text.data(), 1, b->text.size(), out) != b->text.size()) write_failed();
        pool.put(std::move(b));
      }
    }
    reader.join();
    for (auto &t : parsers) t.join();
    for (auto &t : gpu_threads) t.join();
    for (auto &t : formatters) t.join();
    wall_mark("last batch written");
    if (fflush(out) != 0) write_failed();
    if (out != stdout && fclose(out) != 0) write_failed();
    release_mappings();
    wall_mark("output closed, input unmapped");
  };
  for (size_t i = 0; i < list1.size(); i++) {
    if (verbose && multi)
      fprintf(stderr, "%s Processing input file %s%s%s\n", now().c_str(), list1[i].c_str(), paired ? " and " : "", paired ? list2[i].c_str() : "");
    run_sample(list1[i], paired ? list2[i] : std::string(), i < list_out.size() ? list_out[i] : std::string());
  }
  if (verbose) fprintf(stderr, "%s Finished.\n", now().c_str());
  if (getenv("KAIJU_GPU_STAGE_TIMES"))
    fprintf(stderr, "[CPU time per stage, summed over its threads] read %.3f s, parse %.3f s, gpu calls %.3f s, format %.3f s, "
                    "write %.3f s\n", g_ns_read.load() * 1e-9, g_ns_parse.load() * 1e-9, g_ns_gpu.load() * 1e-9,
            g_ns_format.load() * 1e-9, g_ns_write.load() * 1e-9);
  if (loader.joinable()) loader.join();       // (an input without reads: the verdict on index and nodes.dmp is still due;
                                              //  joined BEFORE any return: leaving main with a joinable thread is std::terminate)
#ifdef KAIJU_CLI_TEST_HOOKS
  if (getenv("KAIJU_GPU_TEST_INEXACT")) inexact_reads++;     // (test build only: the exit path below without a read that overflows anything)
#endif
  if (inexact_batches.load() || inexact_reads.load()) {
    // the reference has no such bounds: say so instead of printing lines that may differ from its output silently
    fprintf(stderr, "%s: a capacity bound of the GPU kernels was exceeded (%llu reads flagged; %llu batches in which the exact pass for "
                    "fragments with many low-complexity regions ran out of room): their lines may differ from the reference's.\n",
            getenv("KAIJU_GPU_ALLOW_INEXACT") ? "Warning" : "Error", (unsigned long long)inexact_reads.load(),
            (unsigned long long)inexact_batches.load());
    if (!getenv("KAIJU_GPU_ALLOW_INEXACT")) { fflush(nullptr); _exit(3); }
  }
  wall_mark("before teardown");
  // Everything is written and closed.  Returning the device memory allocation by allocation and unloading the HIP runtime
  // costs a sizeable fraction of a second that no caller is waiting for, so the process ends here; KAIJU_GPU_CLEAN_EXIT=1 (and
  // any run under a tool library: profilers collect at exit) takes the orderly way out.
  if (!getenv("KAIJU_GPU_CLEAN_EXIT") && !getenv("HSA_TOOLS_LIB") && !getenv("ROCP_TOOL_LIBRARIES") && !getenv("LD_PRELOAD")) {
    fflush(nullptr);
    _exit(EXIT_SUCCESS);
  }
  for (int k = 0; k < n_ctx; k++) if (ctx[(size_t)k]) kaiju_gpu_destroy(ctx[(size_t)k]);
  for (kaiju_gpu_taxonomy *t : dtaxes) if (t) kaiju_gpu_taxonomy_free(t);
  for (kaiju_gpu_index *ix : indexes) if (ix) kaiju_gpu_index_free(ix);
  if (tax) kaiju_taxonomy_free(tax);
  wall_mark("teardown done");
  return EXIT_SUCCESS;
}
