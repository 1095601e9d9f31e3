// pargz.h - gzip input inflated by SEVERAL threads (host code of the drop-in command line; no GPU in here).
//
// Why: the command line classifies plain FASTQ at ~95 M reads/s (profiles/r06_cli/steady.txt) and .gz input - what sequencing
// reads normally arrive as - at 1.7 M reads/s: one zlib inflate() at 570 MB/s of text feeds the whole GPU.  The reference reads
// .gz through zstr / zlib on its one reader thread as well (src/kaiju.cpp:288-394, src/include/zstr); at its 0.2 M reads/s that
// never shows.
//
// How: a deflate stream has no index, but it can be entered at any BLOCK boundary if one accepts not knowing the 32 KB of
// text in front of it (the approach of pugz / rapidgzip, restated):
//   1. the compressed file (memory-mapped) is cut at nominal byte positions; for every piece a thread looks for the first bit
//      position at which a dynamic-Huffman block starts - the block header is tried at every bit offset: the code-length code
//      and both Huffman codes must be complete prefix codes (what zlib's inflate_table demands), the block must decode to its
//      end-of-block symbol, every literal must be 7-bit text, every distance must stay inside what can exist, and the header
//      of the block behind it must be valid too;
//   2. every piece is inflated from its start to the start of the next piece into 16-bit symbols: a byte, or a MARKER
//      "byte i of the unknown 32 KB window" - copies of unknown bytes stay markers;
//   3. the pieces are visited in file order: the first one has no unknown window; the last 32 KB of a finished piece are the
//      window of the next, whose markers are replaced (again by all threads);
//   4. a piece whose inflation does not END exactly where the next piece STARTS proves that start wrong (a bit pattern that
//      only looked like a block): it simply goes on through that piece - every boundary that is used has been reached by the
//      decoder in front of it, so the text is the text zlib would have produced; the CRC-32 and the length in every
//      member's trailer are checked on top (crc32_combine over the pieces), a mismatch is a fatal error.
// Files with several members (bgzip, concatenated files) need nothing special: the decoder walks over trailers and headers.
// Files the search finds no entry into (stored or fixed-Huffman blocks only, pieces smaller than a block) are inflated by the
// first piece's thread alone.  KAIJU_GPU_GZ_THREADS=1 keeps zlib's gzread (the caller's old path).
#pragma once
#include <zlib.h>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace pargz {

constexpr uint32_t kWindow = 32768;
constexpr uint64_t kNoStart = ~0ull;

// ---- bits ------------------------------------------------------------------------------------------------------------
struct Bits {
  const uint8_t *p = nullptr;
  size_t n = 0, pos = 0;         // pos: next byte to pull into bb
  uint64_t bb = 0;
  unsigned bc = 0;
  bool over = false;             // more bits were asked for than the file holds
  void init(const uint8_t *data, size_t size, uint64_t bitpos) {
    p = data; n = size; pos = (size_t)(bitpos >> 3); bb = 0; bc = 0; over = false;
    refill();
    const unsigned skip = (unsigned)(bitpos & 7u);
    if (bc >= skip) { bb >>= skip; bc -= skip; } else over = true;
  }
  inline void refill() {
    if (pos + 8 <= n) {
      uint64_t v;
      memcpy(&v, p + pos, 8);
      bb |= v << bc;
      const unsigned adv = (63u - bc) >> 3;
      pos += adv; bc += adv * 8u;
    } else {
      while (bc <= 56 && pos < n) { bb |= (uint64_t)p[pos++] << bc; bc += 8; }
    }
  }
  inline uint32_t peek(unsigned k) const { return (uint32_t)(bb & ((1ull << k) - 1ull)); }
  inline void drop(unsigned k) { if (k > bc) { over = true; bb = 0; bc = 0; } else { bb >>= k; bc -= k; } }
  inline uint32_t take(unsigned k) { if (bc < k) refill(); const uint32_t v = peek(k); drop(k); return v; }
  uint64_t bitpos() const { return (uint64_t)pos * 8u - bc; }
  void align_byte() { drop(bc & 7u); }
};

// ---- Huffman tables (canonical codes, LSB-first in the stream): primary table + subtables -----------------------------
// entry: low 8 bits = bits to drop (bit 7 set: subtable pointer, low 7 bits = its index width), high bits = symbol / subtable offset
struct Huff {
  std::vector<uint32_t> tab;
  unsigned pbits = 0;
  bool empty = true;             // no symbol at all (a distance code of a block without matches)
};
inline uint32_t rev_bits(uint32_t v, unsigned n) {
  uint32_t r = 0;
  for (unsigned i = 0; i < n; i++) { r = (r << 1) | (v & 1u); v >>= 1; }
  return r;
}
// 0 = fine; -1 = over-subscribed or incomplete (zlib's inflate_table rules: an incomplete code is accepted only when it has
// a single symbol of length 1; no symbol at all is accepted for the distance code, `allow_empty`)
inline int build_huff(const uint8_t *lens, int nsym, unsigned pbits, bool allow_empty, Huff &h) {
  unsigned count[16] = {0};
  for (int s = 0; s < nsym; s++) count[lens[s]]++;
  unsigned maxl = 15;
  while (maxl > 0 && count[maxl] == 0) maxl--;
  h.pbits = pbits;
  if (maxl == 0) {
    if (!allow_empty) return -1;
    h.empty = true;
    h.tab.assign((size_t)1 << pbits, 0u);                     // length 0 = invalid on use
    return 0;
  }
  int left = 1;
  for (unsigned l = 1; l <= 15; l++) { left <<= 1; left -= (int)count[l]; if (left < 0) return -1; }
  if (left > 0 && maxl != 1) return -1;
  h.empty = false;
  unsigned next[16];
  { unsigned code = 0; count[0] = 0; for (unsigned l = 1; l <= 15; l++) { code = (code + count[l - 1]) << 1; next[l] = code; } }
  const unsigned pb = std::min(pbits, maxl);
  h.pbits = pb;
  // subtables: one per distinct primary prefix of the codes longer than pb; sized by the longest code under that prefix
  h.tab.assign((size_t)1 << pb, 0u);
  if (maxl > pb) {
    std::vector<uint8_t> sub_len((size_t)1 << pb, 0);
    unsigned nx[16];
    memcpy(nx, next, sizeof nx);
    for (int s = 0; s < nsym; s++) {
      const unsigned l = lens[s];
      if (l <= pb) { if (l) nx[l]++; continue; }
      const uint32_t code = nx[l]++, r = rev_bits(code, l);
      uint8_t &m = sub_len[r & ((1u << pb) - 1u)];
      if (l - pb > m) m = (uint8_t)(l - pb);
    }
    for (size_t i = 0; i < sub_len.size(); i++) if (sub_len[i]) {
      const uint32_t off = (uint32_t)h.tab.size();
      h.tab.resize(h.tab.size() + ((size_t)1 << sub_len[i]), 0u);
      h.tab[i] = off << 8 | 0x80u | sub_len[i];
    }
  }
  for (int s = 0; s < nsym; s++) {
    const unsigned l = lens[s];
    if (!l) continue;
    const uint32_t code = next[l]++, r = rev_bits(code, l);
    if (l <= pb) {
      for (uint32_t i = r; i < (1u << pb); i += 1u << l) h.tab[i] = (uint32_t)s << 8 | l;
    } else {
      const uint32_t pe = h.tab[r & ((1u << pb) - 1u)], off = pe >> 8;
      const unsigned sb = pe & 0x7fu;
      for (uint32_t i = r >> pb; i < (1u << sb); i += 1u << (l - pb)) h.tab[off + i] = (uint32_t)s << 8 | (l - pb);
    }
  }
  return 0;
}
// the symbol at the head of the bit buffer (at least 15 bits in it, or the end of the file); -1: no such code
inline int huff_get(const Huff &h, Bits &b) {
  uint32_t e = h.tab[b.peek(h.pbits)];
  if (e & 0x80u) {
    b.drop(h.pbits);
    e = h.tab[(e >> 8) + b.peek(e & 0x7fu)];
  }
  const unsigned l = e & 0xffu;
  if (l == 0) return -1;
  b.drop(l);
  return (int)(e >> 8);
}

static const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
static const uint8_t kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

struct BlockCodes { Huff lit, dist; };
inline void fixed_codes(BlockCodes &c) {
  uint8_t l[288], d[30];
  for (int i = 0; i < 144; i++) l[i] = 8;
  for (int i = 144; i < 256; i++) l[i] = 9;
  for (int i = 256; i < 280; i++) l[i] = 7;
  for (int i = 280; i < 288; i++) l[i] = 8;
  for (int i = 0; i < 30; i++) d[i] = 5;
  build_huff(l, 288, 10, false, c.lit);
  // (30 codes of 5 bits: incomplete by the letter, accepted as the standard's fixed code)
  c.dist.pbits = 5; c.dist.empty = false; c.dist.tab.assign(32, 0u);
  for (uint32_t s = 0; s < 30; s++) c.dist.tab[rev_bits(s, 5)] = s << 8 | 5u;
}
// the header of a dynamic block behind its three type bits: 0 fine, -1 not a valid header (zlib: "invalid code lengths set",
// "invalid bit length repeat", "invalid literal/lengths set", "invalid distances set", "missing end-of-block")
inline int read_dynamic_header(Bits &b, BlockCodes &c) {
  b.refill();
  const unsigned hlit = b.take(5) + 257u, hdist = b.take(5) + 1u, hclen = b.take(4) + 4u;
  if (hlit > 286u || hdist > 30u) return -1;
  uint8_t cl[19] = {0};
  for (unsigned i = 0; i < hclen; i++) cl[kClOrder[i]] = (uint8_t)b.take(3);
  if (b.over) return -1;
  Huff clh;
  if (build_huff(cl, 19, 7, false, clh)) return -1;
  uint8_t lens[286 + 30];
  unsigned n = 0;
  const unsigned total = hlit + hdist;
  while (n < total) {
    b.refill();
    const int s = huff_get(clh, b);
    if (s < 0 || b.over) return -1;
    if (s < 16) { lens[n++] = (uint8_t)s; continue; }
    unsigned rep, val = 0;
    if (s == 16) { if (n == 0) return -1; val = lens[n - 1]; rep = 3u + b.take(2); }
    else if (s == 17) rep = 3u + b.take(3);
    else rep = 11u + b.take(7);
    if (n + rep > total) return -1;
    while (rep--) lens[n++] = (uint8_t)val;
  }
  if (b.over || lens[256] == 0) return -1;
  if (build_huff(lens, (int)hlit, 11, false, c.lit)) return -1;
  if (build_huff(lens + hlit, (int)hdist, 8, true, c.dist)) return -1;
  return 0;
}

// a buffer that grows without zero-filling what it grows by (std::vector::resize writes every new element: a pass over
// tens of megabytes per piece that nobody reads)
template <class T>
struct RawBuf {
  T *p = nullptr;
  size_t cap = 0;
  RawBuf() = default;
  RawBuf(const RawBuf &) = delete;
  RawBuf &operator=(const RawBuf &) = delete;
  RawBuf(RawBuf &&o) noexcept : p(o.p), cap(o.cap) { o.p = nullptr; o.cap = 0; }
  RawBuf &operator=(RawBuf &&o) noexcept { if (this != &o) { free(p); p = o.p; cap = o.cap; o.p = nullptr; o.cap = 0; } return *this; }
  ~RawBuf() { free(p); }
  size_t size() const { return cap; }
  void resize(size_t n) {
    if (n <= cap) return;
    T *q = static_cast<T *>(realloc(p, n * sizeof(T)));
    if (!q) { fprintf(stderr, "pargz: out of memory\n"); abort(); }
    p = q; cap = n;
  }
  T *data() { return p; }
  const T *data() const { return p; }
  T &operator[](size_t i) { return p[i]; }
};

// ---- the inflater: output type T = uint8_t (window known) or uint16_t (bytes, or 256 + i = byte i of the unknown window) --
struct MemberEnd { uint64_t out_pos; uint32_t crc, isize; };   // a member ended in front of output position out_pos
template <class T>
struct Inflater {
  Bits b;
  RawBuf<T> out;                 // (T = uint16_t: the kWindow markers in front of the text are part of it)
  size_t o = 0;                  // next output position in out
  size_t member_from = 0;        // output position at which the current member began (distances cannot reach in front of it:
                                 // a decoder that entered the stream in mid-member takes 0 = the front of its window)
  bool text_only = false;        // the search's trial runs: a literal beyond 7 bits ends them
  std::vector<MemberEnd> ends;
  bool file_done = false;        // the last member's trailer has been read
  BlockCodes codes;
  const char *err = nullptr;

  void grow(size_t need) { if (out.size() < o + need) out.resize(std::max(out.size() * 2, o + need + (1u << 20))); }

  // one block; 0 fine, 1 = it was a member's last block (the trailer has been consumed; file_done if nothing follows), -1 error
  int block(size_t max_out) {
    b.refill();
    if (b.bc < 3) { err = "truncated (block header)"; return -1; }
    const uint32_t bfinal = b.take(1), btype = b.take(2);
    if (btype == 3) { err = "invalid block type"; return -1; }
    if (btype == 0) {
      b.align_byte();
      b.refill();
      const uint32_t len = b.take(16);
      b.refill();
      const uint32_t nlen = b.take(16);
      if (b.over || (len ^ 0xffffu) != nlen) { err = "invalid stored block lengths"; return -1; }
      // (the bit buffer holds whole bytes now: hand them back and copy from the file)
      size_t src = b.pos - b.bc / 8;
      if (src + len > b.n) { err = "truncated (stored block)"; return -1; }
      if (o + len > max_out) { err = "output limit"; return -1; }
      grow(len);
      for (uint32_t i = 0; i < len; i++) {
        const uint8_t v = b.p[src + i];
        if (text_only && v >= 128) { err = "not text"; return -1; }
        out[o++] = (T)v;
      }
      b.init(b.p, b.n, (uint64_t)(src + len) * 8u);
    } else {
      if (btype == 1) fixed_codes(codes);
      else if (read_dynamic_header(b, codes)) { err = "invalid dynamic block header"; return -1; }
      for (;;) {
        if (out.size() < o + 260) grow(1u << 16);
        b.refill();
        int s = huff_get(codes.lit, b);
        if (s < 256) {
          if (s < 0) { err = "invalid literal/length code"; return -1; }
          if (text_only && s >= 128) { err = "not text"; return -1; }
          out[o++] = (T)s;
          // (two more literals from the same refill: 3 x 15 bits fit what refill() guarantees)
          s = huff_get(codes.lit, b);
          if (s < 256) {
            if (s < 0) { err = "invalid literal/length code"; return -1; }
            if (text_only && s >= 128) { err = "not text"; return -1; }
            out[o++] = (T)s;
            continue;
          }
        }
        if (s == 256) break;
        s -= 257;
        if (s >= 29) { err = "invalid literal/length code"; return -1; }
        b.refill();
        const uint32_t len = kLenBase[s] + b.take(kLenExtra[s]);
        if (codes.dist.empty) { err = "distance code in a block without distances"; return -1; }
        const int ds = huff_get(codes.dist, b);
        if (ds < 0 || ds >= 30) { err = "invalid distance code"; return -1; }
        b.refill();
        const uint32_t dist = kDistBase[ds] + b.take(kDistExtra[ds]);
        if (b.over) { err = "truncated (match)"; return -1; }
        if ((size_t)dist > o - member_from) { err = "invalid distance too far back"; return -1; }
        if (o + len > max_out) { err = "output limit"; return -1; }
        const T *src = out.data() + (o - dist);
        T *dst = out.data() + o;
        if (dist >= len) memcpy(dst, src, (size_t)len * sizeof(T));
        else for (uint32_t i = 0; i < len; i++) dst[i] = src[i];
        o += len;
      }
      if (b.over) { err = "truncated (block)"; return -1; }
    }
    if (!bfinal) return 0;
    // the member's trailer, and the header of the next member if one follows
    b.align_byte();
    size_t at = b.pos - b.bc / 8;
    if (at + 8 > b.n) { err = "truncated (trailer)"; return -1; }
    MemberEnd me;
    me.out_pos = o;
    memcpy(&me.crc, b.p + at, 4); memcpy(&me.isize, b.p + at + 4, 4);
    ends.push_back(me);
    at += 8;
    member_from = o;
    // (zero padding behind the last member is tolerated, as gzip -d does)
    size_t q = at;
    while (q < b.n && b.p[q] == 0) q++;
    if (q >= b.n) { file_done = true; b.init(b.p, b.n, (uint64_t)b.n * 8u); return 1; }
    const long hl = gzip_header_len(b.p + at, b.n - at);
    if (hl < 0) { err = "garbage behind a gzip member"; return -1; }
    b.init(b.p, b.n, (uint64_t)(at + (size_t)hl) * 8u);
    return 1;
  }
  // length of the gzip member header at p, -1 if there is none
  static long gzip_header_len(const uint8_t *p, size_t n) {
    if (n < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || (p[3] & 0xe0)) return -1;
    const uint8_t flg = p[3];
    size_t q = 10;
    if (flg & 4) { if (q + 2 > n) return -1; q += 2u + (size_t)(p[q] | p[q + 1] << 8); }
    if (flg & 8) { while (q < n && p[q]) q++; q++; }
    if (flg & 16) { while (q < n && p[q]) q++; q++; }
    if (flg & 2) q += 2;
    return q < n ? (long)q : -1;
  }
};

// ---- the search for a block start ----------------------------------------------------------------------------------------
// the first bit position in [from, to) at which a dynamic block starts (see the top of the file for what that means), kNoStart if none
inline uint64_t find_block_start(const uint8_t *data, size_t size, uint64_t from, uint64_t to, Inflater<uint16_t> &trial) {
  trial.text_only = true;
  for (uint64_t s = from; s < to; s++) {
    // the cheap part first: BFINAL = 0, BTYPE = 2, HLIT <= 29, HDIST <= 29, and a complete code-length code
    {
      const size_t byte = (size_t)(s >> 3);
      if (byte + 16 > size) return kNoStart;
      uint64_t v;
      memcpy(&v, data + byte, 8);
      v >>= (s & 7u);
      if ((v & 7u) != 4u) continue;                          // bits: 0 (not final), then 10 binary LSB first = 2
      const unsigned hlit = (unsigned)(v >> 3) & 31u, hdist = (unsigned)(v >> 8) & 31u, hclen = ((unsigned)(v >> 13) & 15u) + 4u;
      if (hlit > 29u || hdist > 29u) continue;
      // code-length code lengths: 3 bits each from bit 17
      Bits q;
      q.init(data, size, s + 17);
      unsigned cnt[8] = {0};
      for (unsigned i = 0; i < hclen; i++) { if (q.bc < 3) q.refill(); cnt[q.peek(3)]++; q.drop(3); }
      int left = 1;
      bool bad = false;
      for (unsigned l = 1; l <= 7; l++) { left <<= 1; left -= (int)cnt[l]; if (left < 0) { bad = true; break; } }
      unsigned nz = 0;
      for (unsigned l = 1; l <= 7; l++) nz += cnt[l];
      if (bad || nz == 0 || (left > 0 && !(nz == 1 && cnt[1] == 1))) continue;
    }
    // the block itself and the one behind it (its header; all of it if it is short)
    trial.b.init(data, size, s);
    trial.o = kWindow; trial.member_from = 0; trial.ends.clear(); trial.file_done = false; trial.err = nullptr;
    if (trial.out.size() < kWindow + (1u << 20)) trial.out.resize(kWindow + (1u << 20));
    const int r1 = trial.block(kWindow + (64u << 20));
    if (r1 != 0) continue;
    if (trial.o - kWindow < 1024) continue;                  // (a real block holds thousands of symbols; a lucky pattern ends at once)
    const uint64_t after = trial.b.bitpos();
    const int r2 = trial.block(kWindow + (128u << 20));
    if (r2 < 0 && !(trial.err && !strcmp(trial.err, "output limit"))) continue;
    (void)after;
    return s;
  }
  return kNoStart;
}

// ---- the reader ----------------------------------------------------------------------------------------------------------
// read(dst, n): the next bytes of the inflated text (0 at the end); errors are fatal through the callback given to open()
// one stretch of the inflated text, in file order; nl: the positions of its '\n' bytes (found by the thread that resolved it:
// the caller's record walk then costs a few loads per record instead of a memchr per line)
struct Text {
  RawBuf<uint8_t> buf;
  size_t n = 0;
  std::vector<uint32_t> nl;
};
inline void find_newlines(const uint8_t *p, size_t n, std::vector<uint32_t> &nl) {
  nl.clear();
  nl.reserve(n / 64 + 16);
  size_t i = 0;
#if defined(__SSE2__)
  const __m128i nlv = _mm_set1_epi8('\n');
  for (; i + 16 <= n; i += 16) {
    unsigned m = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i *>(p + i)), nlv));
    while (m) { nl.push_back((uint32_t)(i + (size_t)__builtin_ctz(m))); m &= m - 1; }
  }
#endif
  for (; i < n; i++) if (p[i] == '\n') nl.push_back((uint32_t)i);
}

class Reader {
 public:
  ~Reader() { stop(); }
  // the next stretch of text (blocks until one is there); false at the end of the file
  bool next_piece(Text &out) {
    std::unique_lock<std::mutex> lk(m_);
    cv_.wait(lk, [this] { return !ready_.empty() || done_; });
    if (ready_.empty()) return false;
    out = std::move(ready_.front()); ready_.pop_front();
    lk.unlock();
    cv_.notify_all();
    return true;
  }
  // data / size: the whole .gz file (memory-mapped by the caller); threads >= 2
  bool open(const uint8_t *data, size_t size, unsigned threads, std::function<void(const std::string &)> fatal) {
    d_ = data; n_ = size; T_ = std::max(2u, threads); fatal_ = std::move(fatal);
    const long hl = Inflater<uint8_t>::gzip_header_len(data, size);
    if (hl < 0) return false;
    first_bit_ = (uint64_t)hl * 8u;
    piece_ = 4u << 20;
    if (const char *e = getenv("KAIJU_GPU_GZ_PIECE")) piece_ = (size_t)std::max(1L, atol(e));
    producer_ = std::thread([this] { produce(); });
    return true;
  }
  size_t read(char *dst, size_t want) {
    size_t got = 0;
    while (got < want) {
      if (cur_at_ == cur_.n) {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [this] { return !ready_.empty() || done_; });
        if (ready_.empty()) break;
        if (cur_.buf.size() && free_texts_.size() < 4 * (size_t)T_) free_texts_.push_back(std::move(cur_));
        cur_ = std::move(ready_.front()); ready_.pop_front(); cur_at_ = 0;
        lk.unlock();
        cv_.notify_all();
        continue;
      }
      const size_t k = std::min(want - got, cur_.n - cur_at_);
      memcpy(dst + got, cur_.buf.data() + cur_at_, k);
      got += k; cur_at_ += k;
    }
    return got;
  }
  double t_find = 0, t_inflate = 0, t_resolve = 0, t_wait = 0;   // wall time of the producer in its phases (seconds)
  // a piece the caller is through with: its buffers serve a later piece (a fresh 18 MB allocation per piece is an mmap, its
  // first touch 4 500 page faults and its release a TLB shootdown on every thread: more than the inflation itself on 64 threads)
  void recycle(Text &&t) {
    std::lock_guard<std::mutex> lk(m_);
    if (free_texts_.size() < 4 * (size_t)T_) free_texts_.push_back(std::move(t));
  }
  uint64_t pieces_entered = 0, pieces_absorbed = 0;         // statistics: pieces a thread entered at a found start / pieces
                                                            // that were inflated by the thread of the piece in front of them
 private:
  struct Piece {
    uint64_t start = kNoStart;       // bit position of the block it starts with
    Inflater<uint16_t> inf;          // (piece 0 of the file: no unknown window - its markers are never referenced)
    uint64_t end_bit = 0;            // where its inflation stopped
    size_t stopped_at = 0;           // index of the piece whose start it reached (or pieces.size(): the end of the round / file)
    bool failed = false;
    Text text;                       // resolved: bytes and newlines
    Inflater<uint16_t> trial;        // the search's scratch
    void reset() {
      start = kNoStart; end_bit = 0; stopped_at = 0; failed = false; crc_parts.clear();
      inf.ends.clear(); inf.file_done = false; inf.err = nullptr; inf.text_only = false;
      text.n = 0; text.nl.clear();
    }
    std::vector<std::pair<size_t, uint32_t>> crc_parts;   // per stretch between member ends: (length, crc32)
  };
  const uint8_t *d_ = nullptr;
  size_t n_ = 0, piece_ = 0;
  unsigned T_ = 2;
  uint64_t first_bit_ = 0;
  std::function<void(const std::string &)> fatal_;
  std::thread producer_;
  std::vector<Piece> ps_;
  std::mutex m_;
  std::condition_variable cv_;
  std::deque<Text> ready_;
  std::vector<Text> free_texts_;
  bool done_ = false, quit_ = false;
  Text cur_;
  size_t cur_at_ = 0;

  void stop() {
    { std::lock_guard<std::mutex> lk(m_); quit_ = true; }
    cv_.notify_all();
    if (producer_.joinable()) producer_.join();
  }
  template <class F> void parallel(size_t n, F &&f) {
    std::atomic<size_t> next{0};
    std::vector<std::thread> th;
    const unsigned k = (unsigned)std::min<size_t>(T_, n);
    for (unsigned t = 1; t < k; t++) th.emplace_back([&] { for (size_t i; (i = next.fetch_add(1)) < n;) f(i); });
    for (size_t i; (i = next.fetch_add(1)) < n;) f(i);
    for (auto &x : th) x.join();
  }
  void emit(Text &&t) {
    std::unique_lock<std::mutex> lk(m_);
    cv_.wait(lk, [this] { return ready_.size() < 2 * (size_t)T_ || quit_; });
    if (quit_) return;
    ready_.push_back(std::move(t));
    lk.unlock();
    cv_.notify_all();
  }
  void produce() {
    // rounds of T pieces.  `at`: the bit position the round starts at (a block start the decoder has REACHED: the first one
    // behind the header, later where the round before stopped); `win`: the 32 KB of text in front of it
    uint64_t at = first_bit_;
    std::vector<uint8_t> win(kWindow, 0);
    size_t win_valid = 0;                                     // bytes of the current member in front of `at` (capped at kWindow)
    uint32_t crc_run = crc32(0L, Z_NULL, 0);                  // of the current member so far
    uint64_t len_run = 0;
    bool file_done = false;
    while (!file_done) {
      { std::lock_guard<std::mutex> lk(m_); if (quit_) return; }
      // the pieces of this round: piece 0 starts at `at`, piece k at the first block start behind byte (at / 8 + k * piece_)
      const size_t base = (size_t)(at >> 3);
      size_t np = std::min<size_t>(T_, (n_ - base + piece_ - 1) / piece_);
      if (np == 0) np = 1;
      if (ps_.size() < np + 1) ps_.resize((size_t)T_ + 1);    // (+ 1: the first piece of the NEXT round, only its start is looked for)
      std::vector<Piece> &ps = ps_;
      const size_t n_ps = np + 1;
      for (size_t k = 0; k < n_ps; k++) ps[k].reset();
      ps[0].start = at;
      auto tnow = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
      double t0 = tnow();
      parallel(np, [&](size_t k1) {
        const size_t k = k1 + 1;
        const size_t from = base + k * piece_;
        if (from >= n_) return;
        const uint64_t lim = (uint64_t)std::min(n_, from + piece_) * 8u;
        ps[k].start = find_block_start(d_, n_, (uint64_t)from * 8u, lim, ps[k].trial);
      });
      t_find += tnow() - t0; t0 = tnow();
      // inflate: piece k from its start until it stands on the start of a later piece (or the end of the file)
      parallel(np, [&](size_t k) {
        Piece &pc = ps[k];
        if (pc.start == kNoStart) { pc.failed = true; return; }
        auto &inf = pc.inf;
        inf.b.init(d_, n_, pc.start);
        inf.out.resize(kWindow + (size_t)piece_ * 8u);
        for (uint32_t i = 0; i < kWindow; i++) inf.out[i] = (uint16_t)(256u + i);
        inf.o = kWindow;
        inf.member_from = k == 0 ? kWindow - win_valid : 0;    // (only piece 0 knows how much of its member lies in front of it)
        size_t nxt = k + 1;
        for (;;) {
          const int r = inf.block(~(size_t)0);
          if (r < 0) { pc.failed = true; break; }
          if (inf.file_done) { nxt = n_ps; break; }
          const uint64_t p = inf.b.bitpos();
          while (nxt < n_ps && (ps[nxt].start == kNoStart || ps[nxt].start < p)) nxt++;
          if (nxt >= n_ps) {
            // beyond the last start of the round: stop at this block boundary if it lies in the next round's territory
            if ((size_t)(p >> 3) >= base + np * piece_) break;
            continue;
          }
          if (ps[nxt].start == p) break;
        }
        pc.end_bit = inf.b.bitpos();
        pc.stopped_at = nxt;
      });
      t_inflate += tnow() - t0; t0 = tnow();
      // the pieces that count, in file order: piece 0, then the piece the one before it stopped at
      std::vector<size_t> live;
      for (size_t k = 0; k < np;) {
        if (ps[k].failed) {
          // (the decoder in front of it stood on its start, or it is piece 0: the stream is damaged)
          fatal_(std::string("gzip: ") + (ps[k].inf.err ? ps[k].inf.err : "no block start"));
          { std::lock_guard<std::mutex> lk(m_); done_ = true; }
          cv_.notify_all();
          return;
        }
        live.push_back(k);
        k = ps[k].stopped_at;
      }
      pieces_entered += live.size() - 1;
      pieces_absorbed += np - live.size();
      // windows, one piece after the other (only the last 32 KB of each are resolved here), then everything in parallel
      std::vector<std::vector<uint8_t>> wins(live.size());
      for (size_t x = 0; x < live.size(); x++) {
        wins[x] = win;
        Piece &pc = ps[live[x]];
        const size_t n_out = pc.inf.o - kWindow;
        const size_t keep = std::min<size_t>(n_out, kWindow);
        std::vector<uint8_t> nw(kWindow);
        if (keep < kWindow) memcpy(nw.data(), win.data() + keep, kWindow - keep);
        const uint16_t *src = pc.inf.out.data() + pc.inf.o - keep;
        for (size_t i = 0; i < keep; i++) { const uint16_t v = src[i]; nw[kWindow - keep + i] = v < 256 ? (uint8_t)v : win[v - 256u]; }
        win.swap(nw);
      }
      parallel(live.size(), [&](size_t x) {
        Piece &pc = ps[live[x]];
        const size_t n_out = pc.inf.o - kWindow;
        {
          std::lock_guard<std::mutex> lk(m_);
          if (!free_texts_.empty()) { pc.text = std::move(free_texts_.back()); free_texts_.pop_back(); pc.text.nl.clear(); }
        }
        pc.text.buf.resize(n_out + 1);
        pc.text.n = n_out;
        const uint16_t *src = pc.inf.out.data() + kWindow;
        const uint8_t *w = wins[x].data();
        uint8_t *dst = pc.text.buf.data();
        for (size_t i = 0; i < n_out; i++) { const uint16_t v = src[i]; dst[i] = v < 256 ? (uint8_t)v : w[v - 256u]; }
        find_newlines(dst, n_out, pc.text.nl);
        // crc32 of the stretches between member ends
        size_t from = 0;
        for (size_t e = 0; e <= pc.inf.ends.size(); e++) {
          const size_t to = e < pc.inf.ends.size() ? (size_t)(pc.inf.ends[e].out_pos - kWindow) : n_out;
          uint32_t c = (uint32_t)crc32(0L, Z_NULL, 0);
          size_t q = from;
          while (q < to) { const size_t k2 = std::min<size_t>(to - q, 1u << 30); c = (uint32_t)crc32(c, dst + q, (uInt)k2); q += k2; }
          pc.crc_parts.emplace_back(to - from, c);
          from = to;
        }
      });
      t_resolve += tnow() - t0; t0 = tnow();
      // the members' trailers, and the text on its way
      for (size_t x = 0; x < live.size(); x++) {
        Piece &pc = ps[live[x]];
        for (size_t e = 0; e < pc.crc_parts.size(); e++) {
          crc_run = (uint32_t)crc32_combine(crc_run, pc.crc_parts[e].second, (z_off_t)pc.crc_parts[e].first);
          len_run += pc.crc_parts[e].first;
          if (e < pc.inf.ends.size()) {
            if (crc_run != pc.inf.ends[e].crc || (uint32_t)len_run != pc.inf.ends[e].isize) {
              fatal_("gzip: CRC-32 / length of a member do not match its trailer (damaged file)");
              { std::lock_guard<std::mutex> lk(m_); done_ = true; }
              cv_.notify_all();
              return;
            }
            crc_run = (uint32_t)crc32(0L, Z_NULL, 0); len_run = 0;
          }
        }
        if (pc.inf.file_done) file_done = true;
      }
      // where the next round starts: the end of the last live piece; how much of its member lies in front of that
      {
        Piece &last = ps[live.back()];
        at = last.end_bit;
        // bytes of the current member in front of `at`: everything since the last member end, or what was there before plus this round
        size_t since = 0;
        bool ended = false;
        for (size_t x = live.size(); x-- > 0 && !ended;) {
          Piece &pc = ps[live[x]];
          const size_t n_out = pc.text.n;
          if (!pc.inf.ends.empty()) { since += n_out - (size_t)(pc.inf.ends.back().out_pos - kWindow); ended = true; }
          else since += n_out;
        }
        win_valid = std::min<size_t>(kWindow, ended ? since : win_valid + since);
      }
      t0 = tnow();
      for (size_t x = 0; x < live.size(); x++) emit(std::move(ps[live[x]].text));
      t_wait += tnow() - t0;
    }
    { std::lock_guard<std::mutex> lk(m_); done_ = true; }
    cv_.notify_all();
  }
};

}  // namespace pargz
