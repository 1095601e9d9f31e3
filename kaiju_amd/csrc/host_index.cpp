// host_index.cpp — see host_index.h
#include "host_index.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>
#include <atomic>
#include <cctype>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <thread>
#include <time.h>

#include "../../include/kaiju_gpu.h"

namespace kj {

void *big_alloc(size_t bytes) {
  void *p = nullptr;
  if (bytes >= (4u << 20)) {
    if (posix_memalign(&p, 2u << 20, bytes) != 0) throw std::bad_alloc();
    (void)madvise(p, bytes, MADV_HUGEPAGE);
  } else {
    p = malloc(bytes ? bytes : 1);
    if (!p) throw std::bad_alloc();
  }
  return p;
}
void big_free(void *p, size_t) { free(p); }

namespace {
struct Reader {
  FILE *fp;
  bool ok = true;
  template <class T> void get(T &v) { if (ok && fread(&v, sizeof(T), 1, fp) != 1) ok = false; }
  void bytes(void *dst, size_t n) { if (ok && n && fread(dst, 1, n, fp) != n) ok = false; }
  void big(void *dst, size_t n);      // the same for the big arrays: pieces read by several threads
  void skip(int64_t n) { if (ok && fseeko(fp, (off_t)n, SEEK_CUR) != 0) ok = false; }
};
unsigned hw_threads() {
  unsigned n = std::thread::hardware_concurrency();
  return n ? std::min(n, 64u) : 4u;
}
// KAIJU_GPU_LOAD_TIMES=1: wall time of the packing phases, on stderr (as capi.hip does for the load as a whole)
struct PackClock {
  bool on;
  double tl;
  static double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
  PackClock() : on(getenv("KAIJU_GPU_LOAD_TIMES") != nullptr), tl(now()) {}
  void mark(const char *what) {
    if (!on) return;
    const double t = now();
    fprintf(stderr, "[kaiju_gpu pack]   %-32s %8.1f ms\n", what, (t - tl) * 1e3);
    tl = t;
  }
};
template <class F> void parallel_for(uint64_t n, F &&fn) {
  const unsigned nt = (unsigned)std::min<uint64_t>(hw_threads(), n ? n : 1);
  if (nt <= 1) { for (uint64_t i = 0; i < n; i++) fn(i); return; }
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nt; t++)
    th.emplace_back([&, t]() { for (uint64_t i = t; i < n; i += nt) fn(i); });
  for (auto &x : th) x.join();
}
void Reader::big(void *dst, size_t n) {
  const size_t piece = 16u << 20;
  if (!ok || n <= 2 * piece) { bytes(dst, n); return; }
  const off_t base = ftello(fp);
  if (base < 0) { bytes(dst, n); return; }     // not seekable
  const int fd = fileno(fp);
  uint8_t *d = static_cast<uint8_t *>(dst);
  std::atomic<bool> good{true};
  parallel_for((n + piece - 1) / piece, [&](uint64_t c) {
    size_t b = (size_t)c * piece;
    const size_t e = std::min(n, b + piece);
    while (b < e) { const ssize_t r = pread(fd, d + b, e - b, base + (off_t)b); if (r <= 0) { good = false; return; } b += (size_t)r; }
  });
  if (!good.load() || fseeko(fp, base + (off_t)n, SEEK_SET) != 0) ok = false;
}
}  // namespace

bool parse_taxid(const char *name, uint64_t &id) {
  // "AX1235.1_4567", "WP_12345.1_987" or "987": digits after the last '_' (or the whole name)
  const char *pch = strrchr(name, '_');
  const unsigned long v = strtoul(pch ? pch + 1 : name, nullptr, 10);
  id = (uint64_t)v;
  return v != ULONG_MAX;
}

void dense_taxa(const std::vector<uint64_t> &seq_taxid, const std::vector<uint8_t> &seq_valid, std::vector<uint32_t> &seq_dense,
                std::vector<uint64_t> &tax_of_dense) {
  const size_t n = seq_taxid.size();
  seq_dense.assign(n, 0xffffffffu);
  tax_of_dense.clear();
  // open addressing over the distinct ids (a database has far fewer taxa than sequences)
  size_t cap = 1024;
  while (cap < 2 * n) cap <<= 1;
  std::vector<uint32_t> slot(cap, 0xffffffffu);
  for (size_t i = 0; i < n; i++) {
    if (!seq_valid[i]) continue;
    const uint64_t id = seq_taxid[i];
    size_t h = (size_t)((id * 0x9E3779B97F4A7C15ull) >> 20) & (cap - 1);
    while (slot[h] != 0xffffffffu && tax_of_dense[slot[h]] != id) h = (h + 1) & (cap - 1);
    if (slot[h] == 0xffffffffu) { slot[h] = (uint32_t)tax_of_dense.size(); tax_of_dense.push_back(id); }
    seq_dense[i] = slot[h];
  }
}

int FmiFile::load(const char *path, std::string &msg, bool lazy) {
  FILE *fp = fopen(path, "rb");
  if (!fp) { msg = std::string("cannot open ") + path; return KAIJU_GPU_ERR_IO; }
  (void)setvbuf(fp, nullptr, _IOFBF, 8u << 20);      // (the names: two small reads per sequence, 200 M sequences in a refseq-class file)
  Reader rd{fp};
  // (sizes in the headers are checked against the size of the file before anything is allocated from them)
  int64_t fsize = 0;
  { struct stat st; if (fstat(fileno(fp), &st) == 0) fsize = (int64_t)st.st_size; }
  if (fsize <= 0) fsize = INT64_MAX;           // not a regular file: no bound
  // BWT header, bwt/bwt.c:51-61
  rd.get(len); rd.get(nseq); rd.get(alen);
  if (!rd.ok || alen <= 1 || alen > 60 || nseq <= 0 || len <= 0 || len > fsize || (int64_t)nseq * 13 > fsize) {
    fclose(fp); msg = "not a Kaiju .fmi file (bad BWT header)"; return KAIJU_GPU_ERR_FORMAT;
  }
  alphabet.resize((size_t)alen);
  rd.bytes(&alphabet[0], (size_t)alen);
  // suffix array header, bwt/suffixArray.c:282-312
  rd.get(salen); rd.get(ncheck); rd.get(chpt_exp); rd.get(nbytes); rd.get(sbits); rd.get(pbits);
  rd.get(mask); rd.get(check); rd.get(sa_nseq);
  if (!rd.ok || sa_nseq != nseq || nbytes <= 0 || nbytes > 8 || chpt_exp < 0 || chpt_exp > 30 || ncheck < 0 ||
      ncheck > fsize / nbytes) {
    fclose(fp); msg = "not a Kaiju .fmi file (bad suffix array header)"; return KAIJU_GPU_ERR_FORMAT;
  }
  ids.resize((size_t)nseq);
  for (int32_t i = 0; i < nseq && rd.ok; i++) {
    const int l = getc_unlocked(fp);                   // (one thread reads this file: no lock per byte)
    if (l == EOF) { rd.ok = false; break; }
    ids[(size_t)i].resize((size_t)l);
    if (l && fread_unlocked(&ids[(size_t)i][0], 1, (size_t)l, fp) != (size_t)l) rd.ok = false;
  }
  rd.skip((int64_t)nseq * 4);   // seqTermOrder: not used by the search
  rd.skip((int64_t)nseq * 8);   // seqlengths: not used by the search
  if (lazy) {
    const off_t at = ftello(fp);
    if (at < 0) rd.ok = false;
    sa_off = (uint64_t)at;
    if ((int64_t)ncheck * nbytes > fsize - (int64_t)at) rd.ok = false; else rd.skip((int64_t)ncheck * nbytes);
  } else {
    sa.resize((size_t)ncheck * (size_t)nbytes);
    rd.big(sa.data(), sa.size());
  }
  // FMI, bwt/fmicommon.h:190-217 + compactfmi.c:165-171
  rd.get(f_alen); rd.get(bwtlen); rd.get(N1); rd.get(N2);
  if (!rd.ok || f_alen != alen || bwtlen != len || N1 <= 0 || N2 <= 0 || (int64_t)N1 * alen * 8 > fsize || (int64_t)N2 * alen * 2 > fsize) {
    fclose(fp); msg = "not a Kaiju .fmi file (bad FMI header)"; return KAIJU_GPU_ERR_FORMAT;
  }
  if (lazy) {
    const off_t at = ftello(fp);
    if (at < 0) rd.ok = false;
    bwt_off = (uint64_t)at;
    if (bwtlen > fsize - (int64_t)at) rd.ok = false; else rd.skip(bwtlen);
  } else {
    bwt.resize((size_t)bwtlen);
    rd.big(bwt.data(), bwt.size());
  }
  rd.skip((int64_t)(N1 - 1) * alen * 8);
  index1_last.resize((size_t)alen);
  rd.bytes(index1_last.data(), (size_t)alen * 8);
  rd.skip((int64_t)N2 * alen * 2);   // index2: rebuilt in the packed layout
  startLcode.resize((size_t)alen + 1);
  rd.bytes(startLcode.data(), ((size_t)alen + 1) * 4);
  const bool ok = rd.ok;
  fclose(fp);
  if (!ok) { msg = "truncated .fmi file"; return KAIJU_GPU_ERR_IO; }
  id_ptrs.resize(ids.size());
  for (size_t i = 0; i < ids.size(); i++) id_ptrs[i] = ids[i].c_str();
  return 0;
}

HostIndexView FmiFile::view() const {
  HostIndexView v;
  v.bwtlen = bwtlen; v.nseq = nseq; v.alen = alen; v.alphabet = alphabet.data();
  v.bwt = bwt.data(); v.startLcode = startLcode.data();
  v.sa = sa.data(); v.ncheck = ncheck; v.chpt_exp = chpt_exp; v.nbytes = nbytes; v.pbits = pbits;
  v.ids = id_ptrs.data();
  return v;
}

int PackedIndex::build_head(const HostIndexView &v, uint8_t *lcode, std::string &msg) {
  if (!v.startLcode || !v.alphabet || !v.ids || v.bwtlen <= 0 || v.nseq <= 0) {
    msg = "incomplete index view"; return KAIJU_GPU_ERR_ARG;
  }
  if (v.alen < 2 || v.alen > 21) {
    msg = "alphabets with more than 20 letters + terminator are not supported"; return KAIJU_GPU_ERR_UNSUPPORTED;
  }
  alen = (uint32_t)v.alen; bwtlen = (uint64_t)v.bwtlen; nseq = (uint32_t)v.nseq; chpt_exp = (uint32_t)v.chpt_exp;
  alphabet.assign(v.alphabet, (size_t)v.alen);
  warnings = 0;
  // translation_table(alphabet, NULL, dummy = len-1, case-insensitive), sequence.c:68-97,132-141
  trans[0] = 0;
  for (int i = 1; i < 128; i++) trans[i] = isalpha(i) ? (uint8_t)(alen - 1) : (uint8_t)255;
  for (uint32_t i = 0; i < alen; i++) {
    trans[toupper((unsigned char)alphabet[i]) & 127] = (uint8_t)i;
    trans[tolower((unsigned char)alphabet[i]) & 127] = (uint8_t)i;
  }
  // byte code -> letter (fmi_fill_codes, compactfmi.c:75-89)
  memset(lcode, 31, 256);
  for (uint32_t a = 0; a < alen; a++) {
    const int s = v.startLcode[a], e = v.startLcode[a + 1];
    if (s < 0 || e > 256 || s > e) { msg = "bad startLcode table"; return KAIJU_GPU_ERR_FORMAT; }
    for (int k = s; k < e; k++) lcode[k] = (uint8_t)a;
  }
  // KAIJU_GPU_FORCE_WIDE=<shift>: treat the index as one with 64-bit positions (tests of that path on small
  // indexes; the value is the log2 of the rows per count base, 16..31)
  wide = bwtlen >= 0xffffffffull;
  mb_shift = 31;
  if (const char *e = getenv("KAIJU_GPU_FORCE_WIDE")) { wide = true; const int v = atoi(e); if (v >= (int)kSbShift && v <= 31) mb_shift = (uint32_t)v; }
  if (v.nbytes < 1 || v.nbytes > 8 || v.pbits < 0 || v.pbits > 62) { msg = "bad suffix array coding"; return KAIJU_GPU_ERR_FORMAT; }
  return 0;
}

// sample geometry, warnings, names and taxon ids of the sequences
void PackedIndex::build_names(const HostIndexView &v, std::vector<std::string> *owned) {
  sa_skip = (((uint64_t)nseq - 1) >> chpt_exp) + 1;
  n_sa = (uint64_t)v.ncheck;
  {
    const uint64_t need = bwtlen > 0 ? (((bwtlen - 1) >> chpt_exp) - sa_skip + 1) : 0;
    if (((bwtlen - 1) >> chpt_exp) >= sa_skip && need > n_sa) warnings |= KAIJU_IDX_WARN_SA_SHORT;
  }
  if (bwtlen > 65536 && (bwtlen % 65536 >= 65408 || bwtlen % 65536 == 0)) warnings |= KAIJU_IDX_WARN_RANK_BUG;
  // taxon ids
  seq_taxid.assign(nseq, 0);
  seq_valid.assign(nseq, 0);
  names.clear();
  const bool take = owned && owned->size() == nseq;       // (a streamed load hands its strings over instead of copying 200 M of them)
  if (take) names.swap(*owned); else names.resize(nseq);
  parallel_for(((uint64_t)nseq + 16383) / 16384, [&](uint64_t chunk) {
    const uint32_t b = (uint32_t)(chunk * 16384), e = (uint32_t)std::min<uint64_t>(nseq, (uint64_t)b + 16384);
    for (uint32_t i = b; i < e; i++) {
      if (!take) names[i] = v.ids[i] ? v.ids[i] : "";
      const char *nm = names[i].c_str();
      uint64_t id = 0;
      // bit 0: usable taxon id; bit 1: the name has an accession part in front of the last '_' (verbose column 6)
      seq_valid[i] = parse_taxid(nm, id) ? (uint8_t)(strrchr(nm, '_') ? 3 : 1) : 0;
      seq_taxid[i] = seq_valid[i] ? id : ~0ull;               // (~0: no id - what the locate walks test, one table instead of two)
    }
  });
}

// the small parts of a .fmi whose big arrays stay in the file (fmi_stream.h)
int PackedIndex::build_streamed(FmiFile &f, const char *path, std::string &msg) {
  const HostIndexView v = f.view();
  stream = FmiStreamSource{};
  int rc = build_head(v, stream.lcode, msg);
  if (rc) return rc;
  build_names(v, &f.ids);                                  // (f.ids / f.id_ptrs are gone afterwards)
  f.id_ptrs.clear();
  stream.path = path; stream.sa_off = f.sa_off; stream.bwt_off = f.bwt_off; stream.nbytes = f.nbytes; stream.pbits = f.pbits;
  blocks64.clear(); mb_base.clear(); sa_taxid.clear(); sa_iseq.clear(); sa_pos.clear(); term_pos.clear(); kmer32.clear(); kmer64.clear();
  kline.clear(); kmer_k = 0;
  lazy = ImageLazy{};
  lazy.blocks64.n = (bwtlen >> 6) + 1; lazy.sa_iseq.n = n_sa; lazy.term_pos.n = nseq;
  if (!wide) { lazy.sa_taxid.n = n_sa + 2; if (f.pbits <= 32) lazy.sa_pos.n = n_sa; }
  return 0;
}

int PackedIndex::build(const HostIndexView &v, std::string &msg) {
  if (!v.bwt || !v.sa) { msg = "incomplete index view"; return KAIJU_GPU_ERR_ARG; }
  uint8_t lcode[256];
  {
    const int rc = build_head(v, lcode, msg);
    if (rc) return rc;
  }
  stream = FmiStreamSource{};
  const uint64_t nsb = (bwtlen >> kSbShift) + 1;            // the host packs in pieces ("superblocks") of 2^kSbShift symbols
  PackClock pc;
  std::vector<uint64_t> sb((size_t)nsb * 20, 0);            // C[c] + occurrences of c before every piece (not uploaded)
  pc.mark("allocate rank blocks");
  // pass 1: letter histogram of every superblock
  std::vector<uint64_t> hist((size_t)nsb * 21, 0);
  const uint8_t *bwt = v.bwt;
  parallel_for(nsb, [&](uint64_t s) {
    const uint64_t b = s << kSbShift, e = std::min<uint64_t>(bwtlen, b + (1ull << kSbShift));
    // four histograms: consecutive equal letters do not wait for each other's increment
    uint32_t h[4][256];
    memset(h, 0, sizeof h);
    uint64_t k = b;
    for (; k + 4 <= e; k += 4) { h[0][bwt[k]]++; h[1][bwt[k + 1]]++; h[2][bwt[k + 2]]++; h[3][bwt[k + 3]]++; }
    for (; k < e; k++) h[0][bwt[k]]++;
    uint64_t hl[32] = {0};
    for (int c = 0; c < 256; c++) hl[lcode[c]] += (uint64_t)h[0][c] + h[1][c] + h[2][c] + h[3][c];
    for (int a = 0; a < 21; a++) hist[(size_t)s * 21 + a] = hl[a];
    hist[(size_t)s * 21] = hl[0];
    if (hl[31]) hist[(size_t)s * 21] |= 1ull << 63;          // a byte outside the code table
  });
  uint64_t total[21] = {0};
  bool stray = false;
  for (uint64_t s = 0; s < nsb; s++) {
    if (hist[(size_t)s * 21] >> 63) { stray = true; hist[(size_t)s * 21] &= ~(1ull << 63); }
    for (int a = 0; a < 21; a++) total[a] += hist[(size_t)s * 21 + a];
  }
  {
    uint64_t sum = 0;
    for (int a = 0; a < 21; a++) sum += total[a];
    if (stray || sum != bwtlen) { msg = "BWT contains byte codes outside the code table"; return KAIJU_GPU_ERR_FORMAT; }
    if (total[0] != nseq) { msg = "number of terminators in the BWT differs from nseq"; return KAIJU_GPU_ERR_FORMAT; }
  }
  pc.mark("letter histograms");
  // C[] (index1[N1-1], fmicommon.h:160-165; InitialSI bwt.c:146-152 uses bwtlen after the last letter)
  C[0] = 0;
  for (uint32_t a = 1; a < alen; a++) C[a] = C[a - 1] + total[a - 1];
  for (uint32_t a = alen; a < 22; a++) C[a] = bwtlen;
  // terminators in front of every superblock: where its rows of term_pos go
  std::vector<uint64_t> term_before((size_t)nsb + 1, 0);
  {
    uint64_t run[21] = {0};
    for (uint64_t s = 0; s < nsb; s++) {
      term_before[(size_t)s + 1] = term_before[(size_t)s] + hist[(size_t)s * 21];
      for (int a = 1; a < 21; a++) {
        sb[(size_t)s * 20 + (a - 1)] = C[a] + run[a];
        run[a] += hist[(size_t)s * 21 + a];
      }
    }
  }
  blocks64.clear(); mb_base.clear();
  const uint64_t nb64 = (bwtlen >> 6) + 1;
  blocks64.resize((size_t)nb64);
  if (wide) {
    // counts relative to the start of every 2^mb_shift rows (a multiple of the superblock size)
    const uint64_t nmb = (bwtlen >> mb_shift) + 1;
    mb_base.assign((size_t)nmb * 20, 0);
    for (uint64_t m = 0; m < nmb; m++)
      for (int a = 0; a < 20; a++) mb_base[(size_t)m * 20 + a] = sb[(size_t)(m << (mb_shift - kSbShift)) * 20 + a];
  }
  term_pos.clear();
  term_pos.resize((size_t)nseq);
  pc.mark("superblock sums, allocations");
  // pass 2, one sweep per piece: the rank blocks (64 symbols, 32-bit counts: absolute, or relative to mb_base) and the rows of
  // the terminators (rank_term).  (every element of the big arrays is written here, by the thread that owns its piece)
  parallel_for(nsb, [&](uint64_t s) {
    uint32_t cnt[32] = {0};
    uint64_t abs64[32];
    for (int a = 1; a < 21; a++) {
      abs64[a] = sb[(size_t)s * 20 + (a - 1)];
      if (wide) abs64[a] -= mb_base[(size_t)((s << kSbShift) >> mb_shift) * 20 + (a - 1)];
    }
    uint64_t *tp = term_pos.data() + term_before[(size_t)s];
    const uint64_t b0 = s << (kSbShift - 6), b1 = std::min<uint64_t>(nb64, b0 + (1ull << (kSbShift - 6)));
    for (uint64_t bi = b0; bi < b1; bi++) {
      RankBlock64 &r64 = blocks64[(size_t)bi];
      for (int a = 1; a < 21; a++) r64.cnt[a - 1] = (uint32_t)(abs64[a] + cnt[a]);
      r64.pad[0] = r64.pad[1] = 0;
      const uint64_t h0 = bi << 6;
      uint64_t pl[5] = {0, 0, 0, 0, 0};
      const uint32_t nsym = h0 >= bwtlen ? 0u : (uint32_t)std::min<uint64_t>(64, bwtlen - h0);
      for (uint32_t t = 0; t < nsym; t++) {
        const uint32_t c = lcode[bwt[h0 + t]];
        cnt[c]++;
        if (c == 0) *tp++ = h0 + t;
        pl[0] |= (uint64_t)(c & 1u) << t; pl[1] |= (uint64_t)((c >> 1) & 1u) << t; pl[2] |= (uint64_t)((c >> 2) & 1u) << t;
        pl[3] |= (uint64_t)((c >> 3) & 1u) << t; pl[4] |= (uint64_t)((c >> 4) & 1u) << t;
      }
      if (nsym < 64) {                                       // padding (code 31) never matches a letter
        const uint64_t pad = nsym ? ~0ull << nsym : ~0ull;
        for (int q = 0; q < 5; q++) pl[q] |= pad;
      }
      for (int q = 0; q < 5; q++) r64.plane[q] = pl[q];
    }
  });
  pc.mark("rank blocks + terminator rows");
  // sampled suffix array: only the sequence number is needed (suffixArray.h:37-51)
  build_names(v);
  pc.mark("names, taxon ids");
  sa_iseq.clear(); sa_iseq.resize((size_t)n_sa);
  {
    const uint8_t *sa = v.sa;
    const int nb = v.nbytes, pb = v.pbits;
    uint32_t *dst = sa_iseq.data();
    // the offsets too (text verification, narrow indexes): they fit 32 bits when pbits <= 32
    sa_pos.clear();
    if (!wide && pb <= 32) sa_pos.resize((size_t)n_sa);
    uint32_t *dpos = sa_pos.empty() ? nullptr : sa_pos.data();
    const uint64_t pmask = pb >= 64 ? ~0ull : ((1ull << pb) - 1ull);
    parallel_for((n_sa + 65535) / 65536, [&](uint64_t chunk) {
      const uint64_t b = chunk * 65536, e = std::min<uint64_t>(n_sa, b + 65536);
      for (uint64_t i = b; i < e; i++) {
        const uint8_t *c = sa + i * (uint64_t)nb;
        uint64_t val = 0;
        for (int q = 0; q < nb; q++) val = (val << 8) + c[q];
        dst[i] = (uint32_t)(val >> pb);
        if (dpos) dpos[i] = (uint32_t)(val & pmask);
      }
    });
  }
  pc.mark("suffix array samples");
  // the taxon id of every sampled row (second-generation lanes: the locate ends with ONE load).  Not for the wide layout:
  // 8 bytes per sample is a byte per row at e = 3 - its lanes read the sequence number and then seq_taxid
  sa_taxid.clear();
  if (!wide) {
    sa_taxid.resize((size_t)n_sa + 2);
    sa_taxid[(size_t)n_sa] = sa_taxid[(size_t)n_sa + 1] = ~0ull;
    parallel_for((n_sa + 65535) / 65536, [&](uint64_t chunk) {
      const uint64_t b = chunk * 65536, e = std::min<uint64_t>(n_sa, b + 65536);
      for (uint64_t q = b; q < e; q++) {
        const uint32_t is = sa_iseq[(size_t)q];
        sa_taxid[(size_t)q] = (is < nseq && seq_valid[is]) ? seq_taxid[is] : ~0ull;
      }
    });
  }
  pc.mark("taxon ids of the samples");
  {
    // the host builds at most 5 letters; the device grows the table further (capi.hip)
    uint32_t k = 5;
    if (const char *e = getenv("KAIJU_GPU_KMER")) { k = (uint32_t)atoi(e); if (k > 6) k = 6; }   // (6: tests/tools, 0.9 GB of host memory)
    build_kmer_table(k);
  }
  pc.mark("k-mer table (host part)");
  return 0;
}

void PackedIndex::build_kmer_table(uint32_t k) {
  kmer32.clear(); kmer64.clear(); kmer_k = 0; kline.clear();
  if (k < 2 || k > 6 || alen != 21) return;
  uint64_t n = 1;
  for (uint32_t q = 0; q < k; q++) n *= 20;
  const bool small = !wide;
  if (small) kmer32.assign((size_t)n, uint2{0, 0}); else kmer64.assign((size_t)n, ulonglong2{0, 0});
  const DevIndex d = host_view();
  // word index = (((c0-1)*20 + (c1-1))*20 + ...), c0 = the letter matched first (InitialSI)
  struct Rec {
    static void go(const DevIndex &d, PackedIndex &pk, bool small, uint32_t k, uint32_t depth, uint64_t idx,
                   uint64_t lo, uint64_t hi) {
      if (depth == k) {
        if (small) pk.kmer32[(size_t)idx] = uint2{(uint32_t)lo, (uint32_t)(hi - lo)};
        else pk.kmer64[(size_t)idx] = ulonglong2{lo, hi - lo};
        return;
      }
      for (uint32_t c = 1; c <= 20; c++) {
        uint64_t nlo = 0, nhi = 0;
        if (lo < hi) { nlo = rank_c(d, c, lo); nhi = rank_c(d, c, hi); if (nlo >= nhi) nlo = nhi = 0; }
        go(d, pk, small, k, depth + 1, idx * 20 + (c - 1), nlo, nhi);
      }
    }
  };
  parallel_for(400, [&](uint64_t t) {
    const uint32_t c0 = (uint32_t)(t / 20) + 1, c1 = (uint32_t)(t % 20) + 1;
    uint64_t lo = C[c0], hi = C[c0 + 1];                       // InitialSI
    uint64_t nlo = 0, nhi = 0;
    if (lo < hi) { nlo = rank_c(d, c1, lo); nhi = rank_c(d, c1, hi); if (nlo >= nhi) nlo = nhi = 0; }
    Rec::go(d, *this, small, k, 2, (uint64_t)(c0 - 1) * 20 + (c1 - 1), nlo, nhi);
  });
  kmer_k = k;
}

// the k-mer lines of the host's table (tests/emu; the device runs the same kline_build_one over the table it has grown)
void PackedIndex::build_klines() {
  kline.clear();
  if (!kmer_k || kmer32.empty() || blocks64.empty()) return;
  const uint64_t nlines = kmer32.size() / 20;
  kline.assign((size_t)nlines * kKLineBytes, 0);
  const DevIndex d = host_view();
  parallel_for((nlines + 4095) / 4096, [&](uint64_t chunk) {
    const uint64_t b = chunk * 4096, e = std::min<uint64_t>(nlines, b + 4096);
    for (uint64_t code = b; code < e; code++) kline_build_one(d, kmer_k, code, kline.data() + (size_t)code * kKLineBytes);
  });
}

void PackedIndex::to_sequence_ids() {
  for (uint32_t i = 0; i < nseq; i++) { seq_taxid[i] = i; seq_valid[i] = 3; }
  if (!sa_taxid.empty()) for (uint64_t q = 0; q < n_sa; q++) sa_taxid[(size_t)q] = sa_iseq[(size_t)q] < nseq ? (uint64_t)sa_iseq[(size_t)q] : ~0ull;
}

// ---- device image file: header, then every array as (u64 element count, raw elements) ----
namespace {
const char kImageMagic[8] = {'K', 'J', 'G', 'P', 'U', 'I', 'M', '6'};   // (6: seq_taxid holds ~0 for names without an id)
struct ImgHeader {
  char magic[8];
  uint64_t sizes[8];          // [0] size in bytes of the .fmi the image was made from, [1..3] sizeof RankBlock64, uint2, ulonglong2
                              // (layout guard), the rest spare
  uint64_t C[22];
  uint64_t bwtlen, n_sa, sa_skip;
  uint32_t nseq, chpt_exp, alen, warnings, kmer_k, mb_shift_wide;   // mb_shift | wide << 8
  uint8_t trans[128];
  char alphabet[64];
};
template <class T, class A> bool put_vec(FILE *fp, const std::vector<T, A> &v) {
  const uint64_t n = v.size();
  if (fwrite(&n, 8, 1, fp) != 1) return false;
  const uint64_t bytes = n * sizeof(T), piece = 64ull << 20;
  if (bytes <= 2 * piece) return n == 0 || fwrite(v.data(), sizeof(T), n, fp) == n;
  // a big array: pieces written side by side (one thread copying tens of GB into the page cache takes a minute)
  if (fflush(fp) != 0) return false;
  const off_t base = ftello(fp);
  if (base < 0) return fwrite(v.data(), sizeof(T), n, fp) == n;
  const int fd = fileno(fp);
  const uint8_t *s = reinterpret_cast<const uint8_t *>(v.data());
  std::atomic<bool> ok{true};
  parallel_for((bytes + piece - 1) / piece, [&](uint64_t c) {
    uint64_t b = c * piece;
    const uint64_t e = std::min<uint64_t>(bytes, b + piece);
    while (b < e) { const ssize_t r = pwrite(fd, s + b, (size_t)(e - b), base + (off_t)b); if (r <= 0) { ok = false; return; } b += (uint64_t)r; }
  });
  return ok.load() && fseeko(fp, base + (off_t)bytes, SEEK_SET) == 0;
}
// positional reads, big arrays in pieces by several threads (an image is about a GB: one thread copying it out of the
// page cache takes a quarter of a second)
struct ImgReader {
  int fd;
  uint64_t pos, size;
  bool raw(void *dst, uint64_t n) {
    if (n > size - pos) return false;
    uint8_t *d = static_cast<uint8_t *>(dst);
    const uint64_t base = pos;
    pos += n;
    const uint64_t piece = 16ull << 20;
    if (n <= 2 * piece) {
      uint64_t done = 0;
      while (done < n) { const ssize_t r = pread(fd, d + done, n - done, (off_t)(base + done)); if (r <= 0) return false; done += (uint64_t)r; }
      return true;
    }
    std::atomic<bool> ok{true};
    parallel_for((n + piece - 1) / piece, [&](uint64_t c) {
      uint64_t b = c * piece;
      const uint64_t e = std::min<uint64_t>(n, b + piece);
      while (b < e) { const ssize_t r = pread(fd, d + b, e - b, (off_t)(base + b)); if (r <= 0) { ok = false; return; } b += (uint64_t)r; }
    });
    return ok.load();
  }
  template <class T, class A> bool vec(std::vector<T, A> &v) {
    uint64_t n = 0;
    if (!raw(&n, 8) || n > (size - pos) / sizeof(T)) return false;
    v.clear();
    v.resize((size_t)n);
    return n == 0 || raw(v.data(), n * sizeof(T));
  }
  // the same array left where it is: count and offset noted, the elements skipped
  template <class T, class A> bool vec_lazy(std::vector<T, A> &v, LazyArr &l, bool lazy) {
    if (!lazy) { l = LazyArr{}; return vec(v); }
    uint64_t n = 0;
    if (!raw(&n, 8) || n > (size - pos) / sizeof(T)) return false;
    v.clear();
    l.n = n; l.off = pos;
    pos += n * sizeof(T);
    return true;
  }
};
}  // namespace

int PackedIndex::write_image(const char *path, std::string &msg) const {
  FILE *fp = fopen(path, "wb");
  if (!fp) { msg = std::string("cannot write ") + path; return KAIJU_GPU_ERR_IO; }
  ImgHeader h;
  memset(&h, 0, sizeof h);
  memcpy(h.magic, kImageMagic, 8);
  h.sizes[0] = src_fmi_bytes; h.sizes[1] = sizeof(RankBlock64); h.sizes[2] = sizeof(uint2); h.sizes[3] = sizeof(ulonglong2);
  memcpy(h.C, C, sizeof C);
  h.bwtlen = bwtlen; h.n_sa = n_sa; h.sa_skip = sa_skip; h.nseq = nseq; h.chpt_exp = chpt_exp; h.alen = alen;
  h.warnings = warnings; h.kmer_k = kmer_k; h.mb_shift_wide = mb_shift | (wide ? 256u : 0u);
  memcpy(h.trans, trans, 128);
  snprintf(h.alphabet, sizeof h.alphabet, "%s", alphabet.c_str());
  bool ok = fwrite(&h, sizeof h, 1, fp) == 1;
  ok = ok && put_vec(fp, blocks64) && put_vec(fp, sa_taxid) &&
       put_vec(fp, sa_iseq) && put_vec(fp, sa_pos) && put_vec(fp, seq_taxid) && put_vec(fp, seq_valid) && put_vec(fp, term_pos) &&
       put_vec(fp, kmer32) && put_vec(fp, kmer64) && put_vec(fp, mb_base);
  // names: lengths then the characters
  std::vector<uint32_t> nl(names.size());
  std::vector<char> nc;
  for (size_t i = 0; i < names.size(); i++) { nl[i] = (uint32_t)names[i].size(); nc.insert(nc.end(), names[i].begin(), names[i].end()); }
  ok = ok && put_vec(fp, nl) && put_vec(fp, nc);
  ok = (fclose(fp) == 0) && ok;
  if (!ok) { msg = std::string("short write to ") + path; return KAIJU_GPU_ERR_IO; }
  return 0;
}

int PackedIndex::read_image(const char *path, std::string &msg, bool lazy_big) {
  const int fd = open(path, O_RDONLY);
  if (fd < 0) { msg = std::string("cannot open ") + path; return KAIJU_GPU_ERR_IO; }
  ImgReader rd{fd, 0, 0};
  { struct stat st; if (fstat(fd, &st) == 0 && st.st_size > 0) rd.size = (uint64_t)st.st_size; }
  ImgHeader h;
  bool ok = rd.raw(&h, sizeof h) && memcmp(h.magic, kImageMagic, 8) == 0 &&
            h.sizes[1] == sizeof(RankBlock64) && h.sizes[2] == sizeof(uint2) && h.sizes[3] == sizeof(ulonglong2);
  if (!ok) { close(fd); msg = "not a kaiju GPU index image (or written by another version)"; return KAIJU_GPU_ERR_FORMAT; }
  memcpy(C, h.C, sizeof C);
  src_fmi_bytes = h.sizes[0];
  bwtlen = h.bwtlen; n_sa = h.n_sa; sa_skip = h.sa_skip; nseq = h.nseq; chpt_exp = h.chpt_exp; alen = h.alen;
  warnings = h.warnings; kmer_k = h.kmer_k; mb_shift = h.mb_shift_wide & 255u; wide = (h.mb_shift_wide & 256u) != 0;
  memcpy(trans, h.trans, 128);
  h.alphabet[sizeof h.alphabet - 1] = 0;
  alphabet = h.alphabet;
  std::vector<uint32_t> nl;
  std::vector<char> nc;
  lazy = ImageLazy{};
  if (lazy_big) lazy.path = path;
  ok = rd.vec_lazy(blocks64, lazy.blocks64, lazy_big) && rd.vec(sa_taxid) &&
       rd.vec_lazy(sa_iseq, lazy.sa_iseq, lazy_big) && rd.vec_lazy(sa_pos, lazy.sa_pos, lazy_big) && rd.vec(seq_taxid) && rd.vec(seq_valid) &&
       rd.vec_lazy(term_pos, lazy.term_pos, lazy_big) &&
       rd.vec_lazy(kmer32, lazy.kmer32, lazy_big) && rd.vec_lazy(kmer64, lazy.kmer64, lazy_big) && rd.vec(mb_base) && rd.vec(nl) && rd.vec(nc);
  close(fd);
  uint64_t total = 0;
  for (uint32_t l : nl) total += l;
  // consistency of what the kernels will index
  ok = ok && total == nc.size() && nl.size() == nseq && seq_taxid.size() == nseq && seq_valid.size() == nseq &&
       count(blocks64, lazy.blocks64) == (bwtlen >> 6) + 1 && count(sa_iseq, lazy.sa_iseq) >= n_sa && (wide ? sa_taxid.empty() : sa_taxid.size() >= n_sa) &&
       (!wide || mb_base.size() == (size_t)((bwtlen >> mb_shift) + 1) * 20) &&
       (count(sa_pos, lazy.sa_pos) == 0 || count(sa_pos, lazy.sa_pos) == count(sa_iseq, lazy.sa_iseq)) && count(term_pos, lazy.term_pos) == nseq &&
       ((count(kmer32, lazy.kmer32) == 0) != (count(kmer64, lazy.kmer64) == 0) || kmer_k == 0);
  if (!ok) { msg = "truncated or inconsistent index image"; return KAIJU_GPU_ERR_FORMAT; }
  names.clear();
  names.resize(nl.size());
  std::vector<uint64_t> no(nl.size() + 1, 0);
  for (size_t i = 0; i < nl.size(); i++) no[i + 1] = no[i] + nl[i];
  parallel_for((nl.size() + 16383) / 16384, [&](uint64_t chunk) {
    const size_t b = (size_t)chunk * 16384, e = std::min(nl.size(), b + 16384);
    for (size_t i = b; i < e; i++) names[i].assign(nc.data() + no[i], nl[i]);
  });
  return 0;
}

// The text and the full suffix array of a narrow index, on the host (test emulation; capi.hip runs the same steps as
// kernels: k_suffix_walk, k_text_build): every row's (sequence, offset) by get_suffix, the sequences' lengths from the rows
// of their terminator suffixes, sequence s (in the numbering of the samples) at text[off[s]] = 0, text[off[s] + 1 ..] = residues.
void PackedIndex::build_text_wide(uint32_t shift) {
  sa_full.clear(); text.clear(); row_seq.clear(); sa_tpos5.clear();
  tv_shift = shift;
  if (!wide || blocks64.empty() || term_pos.empty() || shift > 8 || (warnings & KAIJU_IDX_WARN_SA_SHORT)) return;   // (as capi.hip decides)
  const DevIndex d = host_view();
  std::vector<uint32_t> t_seq(nseq), len(nseq, 0);
  std::atomic<bool> ok{true};
  parallel_for(((uint64_t)nseq + 255) / 256, [&](uint64_t chunk) {
    const uint64_t b = chunk * 256, e = std::min<uint64_t>(nseq, b + 256);
    for (uint64_t t = b; t < e; t++) {
      uint32_t q = 0;
      const uint64_t n = seq_walk_len(d, t, q);
      if (q >= nseq || n >= 0xffffffffull) { ok = false; q = 0; }
      t_seq[(size_t)t] = q;
      len[q] = (uint32_t)n;                    // (every sequence is the end of exactly one walk)
    }
  });
  if (!ok.load()) return;
  std::vector<uint64_t> off((size_t)nseq + 1, 0);
  off[0] = kTextPad;
  for (uint32_t q = 0; q < nseq; q++) off[(size_t)q + 1] = off[q] + len[q] + 1;
  if (off[nseq] + 2 * kTextPad >= kTposNone) return;
  text.assign((size_t)(off[nseq] + 2 * kTextPad), 0);
  sa_tpos5.assign((size_t)(((bwtlen >> shift) + 1) * 5 + 16), 0xff);
  // the row -> dense taxon index table of a wide index (capi.hip builds it where HBM has room; KAIJU_EMU_NO_ROW_TAX: without)
  std::vector<uint32_t> seq_dense;
  const bool rt = !getenv("KAIJU_EMU_NO_ROW_TAX");
  if (rt) {
    dense_taxa(seq_taxid, seq_valid, seq_dense, tax_of_dense);
    row_seq.resize((size_t)bwtlen + 16);
    for (size_t r = 0; r < row_seq.size(); r++) row_seq[r] = 0xffffffffu;
  }
  parallel_for(((uint64_t)nseq + 255) / 256, [&](uint64_t chunk) {
    const uint64_t b = chunk * 256, e = std::min<uint64_t>(nseq, b + 256);
    for (uint64_t t = b; t < e; t++) {
      const uint32_t q = t_seq[(size_t)t];
      seq_walk_fill(d, t, off[(size_t)q + 1], len[q], text.data(), sa_tpos5.data(), shift, rt ? row_seq.data() : nullptr, rt ? seq_dense[q] : 0xffffffffu);
    }
  });
}

void PackedIndex::build_text() {
  sa_full.clear(); text.clear(); row_seq.clear(); sa_tpos5.clear();
  if (wide || sa_pos.empty() || blocks64.empty() || bwtlen + nseq + 2 * kTextPad >= 0xffffffffull) return;
  const DevIndex d = host_view();
  BigVec<uint32_t> rs((size_t)bwtlen + 4), rp((size_t)bwtlen);      // (rs: four entries of padding for the locate's 16-byte loads)
  for (int q = 0; q < 4; q++) rs[(size_t)bwtlen + q] = 0xffffffffu;
  std::atomic<bool> ok{true};
  // (an index with the reference's short sample array, KAIJU_IDX_WARN_SA_SHORT: the handful of rows whose walk runs into the
  //  missing sample are resolved through the next one; a locate of such a row stays undefined, i.e. skipped - see below)
  std::mutex bm;
  std::vector<uint64_t> beyond_rows;
  parallel_for((bwtlen + 65535) / 65536, [&](uint64_t chunk) {
    const uint64_t b = chunk * 65536, e = std::min<uint64_t>(bwtlen, b + 65536);
    for (uint64_t r = b; r < e; r++) {
      bool by = false;
      if (!suffix_of_row(d, sa_pos.data(), r, rs[(size_t)r], rp[(size_t)r], &by)) ok = false;
      if (by) { std::lock_guard<std::mutex> lk(bm); beyond_rows.push_back(r); }
    }
  });
  if (!ok.load() || beyond_rows.size() > kBeyondRowsMax) return;
  std::vector<uint64_t> off((size_t)nseq + 1, 0);
  {
    std::vector<uint32_t> len(nseq, 0);
    for (uint32_t r = 0; r < nseq; r++) { if (rs[r] >= nseq) return; len[rs[r]] = rp[r]; }
    off[0] = kTextPad;
    for (uint32_t q = 0; q < nseq; q++) off[(size_t)q + 1] = off[q] + len[q] + 1;
  }
  if (off[nseq] + kTextPad >= 0xffffffffull) return;
  text.assign((size_t)(off[nseq] + 2 * kTextPad), 0);
  sa_full.resize((size_t)bwtlen);
  parallel_for((bwtlen + 65535) / 65536, [&](uint64_t chunk) {
    const uint64_t b = chunk * 65536, e = std::min<uint64_t>(bwtlen, b + 65536);
    for (uint64_t r = b; r < e; r++) {
      const uint64_t g = off[rs[(size_t)r]] + 1 + rp[(size_t)r];
      sa_full[(size_t)r] = (uint32_t)g;
      text[(size_t)(g - 1)] = (uint8_t)symbol_at(d, r);
    }
  });
  row_seq.swap(rs);
  for (uint64_t r : beyond_rows) row_seq[(size_t)r] = 0xffffffffu;        // located rows: no sequence (the reference reads out of bounds)
  beyond_lo = beyond_n = beyond_row = 0;
  if (!beyond_rows.empty()) {
    // they lie next to each other in the text (one walk towards the missing sample): DevIndex::beyond_lo
    uint32_t lo = 0xffffffffu, hi = 0;
    for (uint64_t r : beyond_rows) { lo = std::min(lo, sa_full[(size_t)r]); hi = std::max(hi, sa_full[(size_t)r]); }
    if (hi - lo + 1 != beyond_rows.size()) { sa_full.clear(); text.clear(); row_seq.clear(); return; }
    beyond_lo = lo; beyond_n = (uint32_t)beyond_rows.size(); beyond_row = (uint32_t)beyond_rows[0];
  }
  // row -> sequence becomes row -> dense taxon index (DevIndex::row_tax)
  std::vector<uint32_t> seq_dense;
  dense_taxa(seq_taxid, seq_valid, seq_dense, tax_of_dense);
  parallel_for((bwtlen + 65535) / 65536, [&](uint64_t chunk) {
    const uint64_t b = chunk * 65536, e = std::min<uint64_t>(bwtlen, b + 65536);
    for (uint64_t r = b; r < e; r++) { const uint32_t q = row_seq[(size_t)r]; row_seq[(size_t)r] = q < nseq ? seq_dense[q] : 0xffffffffu; }
  });
}

int PackedIndex::image_source_bytes(const char *path, uint64_t &bytes, std::string &msg) {
  FILE *fp = fopen(path, "rb");
  if (!fp) { msg = std::string("cannot open ") + path; return KAIJU_GPU_ERR_IO; }
  ImgHeader h;
  const bool ok = fread(&h, sizeof h, 1, fp) == 1 && memcmp(h.magic, kImageMagic, 8) == 0;
  fclose(fp);
  if (!ok) { msg = "not a kaiju GPU index image (or written by another version)"; return KAIJU_GPU_ERR_FORMAT; }
  bytes = h.sizes[0];
  return 0;
}

uint64_t PackedIndex::bytes() const {
  return count(sa_iseq, lazy.sa_iseq) * 4 + seq_taxid.size() * 8 + seq_valid.size() + count(term_pos, lazy.term_pos) * 8 +
         count(kmer32, lazy.kmer32) * 8 + count(kmer64, lazy.kmer64) * 16 +
         mb_base.size() * 8 + count(blocks64, lazy.blocks64) * sizeof(RankBlock64) + count(sa_taxid, lazy.sa_taxid) * 8;
}

DevIndex PackedIndex::host_view() const {
  DevIndex d;
  d.blocks64 = blocks64.empty() ? nullptr : blocks64.data(); d.sa_taxid = sa_taxid.empty() ? nullptr : sa_taxid.data();
  d.sa_iseq = sa_iseq.data();
  d.seq_taxid = seq_taxid.data(); d.seq_valid = seq_valid.data(); d.term_pos = term_pos.data();
  for (int a = 0; a < 22; a++) d.C[a] = C[a];
  d.bwtlen = bwtlen; d.n_sa = n_sa; d.sa_skip = sa_skip; d.nseq = nseq; d.chpt_exp = chpt_exp;
  d.mb_base = mb_base.empty() ? nullptr : mb_base.data(); d.mb_shift = mb_shift;
  d.kmer32 = kmer32.empty() ? nullptr : kmer32.data();
  d.kmer64 = kmer64.empty() ? nullptr : kmer64.data();
  d.kmer_k = kmer_k;
  d.kline = kline.empty() ? nullptr : kline.data(); d.kline_k = kline.empty() ? 0u : kmer_k;
  d.sa_full = sa_full.empty() ? nullptr : sa_full.data();
  d.text = text.empty() ? nullptr : text.data();
  d.row_tax = row_seq.empty() ? nullptr : row_seq.data();
  d.tax_of_dense = tax_of_dense.empty() ? nullptr : tax_of_dense.data(); d.n_dense = (uint32_t)tax_of_dense.size();
  d.beyond_lo = beyond_lo; d.beyond_n = getenv("KAIJU_EMU_NO_BEYOND_RULE") ? 0u : beyond_n; d.beyond_row = beyond_row;   // (the knob: tests show what the rule is for)
  d.sa_tpos5 = sa_tpos5.empty() ? nullptr : sa_tpos5.data(); d.tv_shift = tv_shift;
  return d;
}

}  // namespace kj
