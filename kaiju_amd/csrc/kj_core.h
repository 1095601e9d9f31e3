// kj_core.h — per-lane logic of the Kaiju classification kernels for gfx950.
//
// Everything here is written as plain inline functions over raw pointers so that
//   (a) hipcc compiles it into the __global__ kernels of kj_kernels.hip, and
//   (b) tests/emu/ compiles the very same source with g++ to check the kernel
//       logic against the oracle on machines without a GPU (test infrastructure
//       only — the product library has no CPU path).
//
// Design: the FM-index is re-packed into 128-byte rank blocks
// (one L2 line = one rank query); every lane runs a small state machine that
// advances ONE dependent memory step (a backward-extension step = two rank
// queries, or one LF step of the locate walk) per loop iteration and pulls the
// next read from a global work counter when it is done, so all 64 lanes of a
// wavefront always have a memory request in flight regardless of how long their
// individual read takes.
//
// Reference behaviour reproduced (file:line in /root/reference/src) is cited at
// each function.
#pragma once
#include <type_traits>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define KJ_HD __host__ __device__ __forceinline__
#define KJ_HD_NOINLINE __host__ __device__ __noinline__
#else
#define KJ_HD inline
#define KJ_HD_NOINLINE inline
struct uint2 { uint32_t x, y; };
struct ulonglong2 { uint64_t x, y; };
#endif

namespace kj {

struct u128 { uint64_t x, y; };
typedef u128 u128_unaligned __attribute__((aligned(1)));
typedef uint32_t u32_unaligned __attribute__((aligned(1)));
typedef uint64_t u64_unaligned __attribute__((aligned(1)));

// ----------------------------------------------------------------------------
// constants shared with the host
// ----------------------------------------------------------------------------
constexpr int kMaxIds = 21;
constexpr int kSbShift = 16;          // host packing works in pieces of 65536 symbols; the wide layout's count bases are multiples
constexpr uint32_t kHitIdCap = 1u, kHitSiCap = 2u;
constexpr uint32_t kHitInternalOverflow = 0x80000000u;   // scratch too small even in the retry pass
constexpr uint32_t kHitRetry = 0x40000000u;              // internal: queued for the retry pass
constexpr uint32_t kHitLocPending = 0x20000000u;         // internal: the ids are still to be located (mem_locate_read); n_ids = matches noted in taxid[]
constexpr int kWin = 64, kWinStride = 68;                // per-lane peptide window (bytes / LDS stride)
constexpr int kLocWideShift = 40;                        // kHitLocPending on a wide index: a match is row | length << 40 ...
constexpr uint32_t kLocWideMaxLen = 1u << 24;            // ... for lengths below this (others are located by the search lane)

// THE rank structure (all lanes, all index sizes since round 3): 64 symbols per 128-byte line, five 64-bit planes + 32-bit
// counts (C[c] folded in) of letters 1..20 before the block - absolute for indexes below 2^32 rows, relative to a base every
// 2^mb_shift rows (DevIndex::mb_base) for larger ones.  One rank query = one line: 16+16+8 bytes of planes, 4 bytes of count.
// (Round 1-2 also kept a 128-symbol layout with 16-bit counts and superblocks for the first-generation lanes: 0.94 B/row that
// a refseq-class index has no room for.)
struct alignas(128) RankBlock64 {
  uint64_t plane[5];
  uint32_t cnt[20];
  uint32_t pad[2];
};
static_assert(sizeof(RankBlock64) == 128, "RankBlock64 must be one 128-byte line");

struct DevIndex {
  const RankBlock64 *blocks64; // [(bwtlen >> 6) + 1]; counts absolute (bwtlen < 2^32) or relative to mb_base
  const uint64_t *mb_base;   // "wide" layout (any bwtlen): [nmb][20] counts (C[c] folded in) at the start of every
  uint32_t mb_shift;         //   2^mb_shift rows; nullptr / 0 when the counts in blocks64 are absolute
  const uint64_t *sa_taxid;  // taxon id of every sampled SA row (~0 = unusable name), for the MEM kernel
  const uint32_t *sa_iseq;   // sequence number of every sampled SA row (rows >= nseq)
  const uint64_t *seq_taxid; // taxon id per sequence (rule of ConsumerThread.cpp:809-833); ~0 where the name gives none (seq_valid 0):
                             // the locate walks read ONE table per row (round 5; strtoul's ULONG_MAX is the reference's "no id")
  const uint8_t *seq_valid;  // 0 where strtoul gave ULONG_MAX
  const uint64_t *term_pos;  // sorted BWT positions holding the terminator (letter 0)
  uint64_t C[22];            // C[c] = first SA row of suffixes starting with letter c; C[21] = bwtlen
  uint64_t bwtlen;
  uint64_t n_sa;             // number of entries in sa_iseq
  uint64_t sa_skip;          // ((nseq-1) >> e) + 1, see get_suffix bwt.c:115-116
  uint32_t nseq;
  uint32_t chpt_exp;
  // k-mer table: suffix interval of every k-letter word, i.e. the result of InitialSI + (k-1)
  // UpdateSI, so that a search from an end position starts k letters in with ONE lookup.  Exact:
  // an empty entry means the match is shorter than k < min(m, seed_length), which is never
  // recorded and never satisfies the "match starts at position <= 1" break of the reference.
  const uint2 *kmer32;       // {lo, len} when bwtlen < 2^32
  const ulonglong2 *kmer64;  // {lo, len} otherwise
  uint32_t kmer_k;           // 0 = no table
  // k-mer LINES (narrow indexes, second-generation lanes): the same table transposed and packed so that ONE 128-byte line
  // serves TWO consecutive end positions of a search.  Line number = the (k-1)-letter word M = w[j-k+1 .. j-1]; it holds
  // the suffix interval of M.a for all twenty letters a (a = w[j]: the k-mer that ends at j, 6 bytes each, KLine below) and
  // one bit per letter b saying whether b.M occurs at all (b = w[j-k]: the k-mer that ends at j-1).  Three of four k-mers
  // of a read are not in the index: their lookups cost no line of their own any more.
  const uint8_t *kline;      // [20^(kline_k-1)] lines of 128 bytes; nullptr = none
  uint32_t kline_k;          // letters of the words the LINES describe.  The device builds them from a table of that depth and then
                             // keeps only a five-letter table (25 MB instead of 10 GB at k = 7) for the lanes that read the table -
                             // the first generation: verbose output, retry pass - so kmer_k <= kline_k there
  // TEXT VERIFICATION (narrow indexes that leave room): the database itself - text[] in index-alphabet codes,
  // every sequence behind a 0 byte - and the position in it of the suffix of EVERY row (the full suffix array, 4 bytes a row).
  // Once a backward search has narrowed to one row and still has letters to go, its match grows exactly as far as the read
  // agrees with the text in front of that row's suffix (UpdateSI on a one-row interval succeeds iff the BWT letter of the row
  // = the letter in front of the suffix equals the next letter of the read): one suffix-array load and one 64-byte text load
  // replace - in the benchmark workload - twenty dependent rank steps per such match, 36 of 76 steps per read.
  const uint32_t *sa_full;   // [bwtlen] position in text[] of the suffix of row r; nullptr = no text verification
  const uint8_t *text;       // 64 zero bytes, then per sequence (in the order of the sampled sequence numbers) 0 + its residues
  const uint32_t *row_tax;   // [bwtlen] the TAXON of the sequence the suffix of row r lies in (get_suffix's walk to a sampled row,
                             // done for every row along with sa_full), as a dense index into tax_of_dense; 0xffffffff = the row
                             // contributes no id (a name without a usable taxon id, ids_from_SI :809-833; a row behind the missing
                             // sample of a KAIJU_IDX_WARN_SA_SHORT index).  The rows of a match are neighbours here: ids_from_SI's
                             // scan over them is contiguous loads only (round 4 kept the SEQUENCE per row: two more, dependent,
                             // random loads a row - seq_valid, seq_taxid - which was half the step on a database of protein
                             // families).  nullptr = the locate walks
  const uint64_t *tax_of_dense;  // [n_dense] taxon id of a dense index (kaijux ids: the sequence number itself)
  uint32_t n_dense;
  // KAIJU_IDX_WARN_SA_SHORT indexes with text arrays: the text positions [beyond_lo, beyond_lo + beyond_n) are the suffixes
  // whose get_suffix walk runs into the missing sample (the reference reads out of bounds there; the walking locate skips such a
  // row).  A match that the text grew is recorded through the row where the text took over, not through its own end row - so
  // the lane applies the skip to the END of the grown match itself (K_SAPOS / K_TEXT), and the records do not depend on whether
  // the text arrays exist.  beyond_row: a row of that range (its row_tax says "no id").  beyond_n = 0: nothing to do
  uint32_t beyond_lo, beyond_n, beyond_row;
  // the same for indexes with 64-bit positions (mb_base set) that leave room: text[] as above and the position in it of the
  // suffix of every 2^tv_shift-th row, 40 bits each (tv_shift = 0 where HBM allows, e.g. 6 B per row at 4 G rows; 1 at
  // refseq_ref's 28 G rows: 3.5 B per row).  A one-row search steps on until its row is one of those (tv_shift = 1: one more
  // step on average), then compares with the text as the narrow lane does.
  const uint8_t *sa_tpos5;   // [(bwtlen >> tv_shift) + 1] entries of 5 bytes (little endian), 16 bytes of padding behind; nullptr = none
  uint32_t tv_shift;
};
constexpr uint64_t kTposNone = (1ull << 40) - 1ull;
constexpr uint32_t kBeyondRowsMax = 4096;    // rows whose walk passes the missing sample of a KAIJU_IDX_WARN_SA_SHORT index (a few)
constexpr uint32_t kTextPad = 64;            // zero bytes in front of the first sequence (a text window never starts below 0)
#ifndef KJ_TEXT_CMP
#define KJ_TEXT_CMP 48
#endif
constexpr int kTextCmp = KJ_TEXT_CMP;        // letters one K_TEXT round compares (a multiple of 4, at most kWin = the 64 bytes loaded)
constexpr int kTextMinLeft = 3;              // letters left in front of the match for the text comparison to be worth its two loads
constexpr int kTextTrigLen = 9;              // ... and the match at least this long: intervals shrink to one row at six to eight letters
                                             // (190 M to 4 G rows) and four of five such matches end right there - the ones that
                                             // have grown two letters past that point go on for twenty more on average

// ---- the span rule ---------------------------------------------------------------------------------------------------------
// maxMatches / greedyExact search every end position j of a fragment anew (bwt.c:265, :356).  Let an earlier search of the
// fragment, from end position j' > j, have ended at i' (w[i'..j'] occurs, w[i'-1..j'] does not) and let the k-mer
// w[j-k+1..j] of the search at hand lie inside that match (j-k+1 >= i') with ONE row in the index.  That row is the k-mer's
// copy inside the (then also unique) occurrence of w[i'..j'], so the UpdateSI chain of this search walks that occurrence: it
// succeeds down to i' and fails at i'-1, where the text holds a letter that is not w[i'-1].  The search ends at i' - known
// without a step - and is never recorded: greedyExact wants l >= L with L >= j'-i'+1 > j-i'+1 (bwt.c:364-371), maxMatches
// wants i < cur->qi with cur->qi <= i' once the search from j' was recorded, and l >= L otherwise failed there already
// (bwt.c:274-276).  The lanes keep i' in `i` (set to the fragment's length when a fragment starts) and go straight to the
// end-of-match bookkeeping, which evaluates those very conditions.  Measured: DESIGN.md 3.3 / 3.4.
#ifdef KJ_NO_SPAN_RULE
constexpr bool kSpanRule = false;
#else
constexpr bool kSpanRule = true;
#endif
constexpr bool kSpanRuleStep = kSpanRule;          // Greedy: also for intervals that shrink to one row behind the k-mer lookup
// ---- the span rule for intervals of ANY size (round 4) ------------------------------------------------------------------------
// The rule above asks for a k-mer with ONE row.  What it uses is that every occurrence of the k-mer lies inside an occurrence
// of the earlier match M = w[i'..j'] - and that holds whenever the k-mer K (inside M) has AS MANY rows as M: every occurrence
// of M holds one of K at a fixed offset, so |occ(K)| >= |occ(M)|, and with equality that map is onto.  Then every occurrence
// of K is preceded by w[j-k], ..., w[i'] and none by w[i'-1] (UpdateSI of that letter on M's interval found nothing): the
// search from j ends at i', as a one-row one would.  Databases are redundant - strains, paralogues, the benchmark's mutated
// copies, a refseq-class index in which every protein occurs seven times: a true match there NEVER has one row, and without
// this the searches from the forty end positions inside a 50-letter match each walk the whole of it again.  `last_sz` = the
// size of the interval in which the last search ended (MEM: of `i`; Greedy also: of the last recorded match, `last_qi`).
#ifdef KJ_NO_SPAN_EQ
constexpr bool kSpanEq = false;
#else
constexpr bool kSpanEq = true;
#endif
// ---- probes (MEM lane, narrow) ---------------------------------------------------------------------------------------------
// greedyExact records a match only if it is at least L long (L = min_fragment_length, then the longest so far, bwt.c:364).
// A match of L letters that ends at j' in [j-L+k, j] contains the k-mer that ends at e = j-L+k (letters j-L+1 .. e): if that
// k-mer is not in the index - or is one row inside the last match (the span rule above: every search that reaches it ends
// where that match ended) - none of the L-k+1 end positions e..j can be recorded, and the searches from them have no other
// effect (`if (i<=1) break`, bwt.c:376, only ends a loop that ends anyway: an unrecorded match is shorter than L <= j+1).
// One lookup passes them all; a k-mer that is there sends the lane to the usual search from j.
#ifdef KJ_NO_PROBE
constexpr bool kMemProbe = false;
#else
constexpr bool kMemProbe = true;
#endif
// Greedy lane (narrow): maxMatches records a match only if it starts in front of the last recorded one (i < cur->qi,
// bwt.c:276).  Once a match [q, J] is recorded, a recordable match that ends at j' >= q+k-2 contains the letters q-1 .. q+k-2:
// ONE lookup of that k-mer right behind the recording; absent (it holds the letter the match failed on) = the end positions
// q+k-2 .. J-1 are passed at once (they cannot break the loop either: a search from a smaller end position reaches at least as
// far as the one from J, so without that k-mer their matches start at q itself, and q > 1 or the loop had ended at J).
#ifdef KJ_NO_PROBE
constexpr bool kGreedyProbe = false;
#else
constexpr bool kGreedyProbe = true;
#endif

// ---- k-mer lines ----------------------------------------------------------------------------------------------------------
// bytes 0..119: twenty entries {lo: 32 bit, len16: 16 bit} for a = 1..20; bytes 120..123: bit b-1 set = the word b.M has a
// non-empty interval.  len16: 0 = empty; 1..kKLineMaxLen = the interval's length; kKLineSingle | c = ONE row whose BWT
// letter is c (0 = terminator): the UpdateSI behind the lookup fails unless the next letter of the read is c - decided
// without fetching the rank line; kKLineEscape = longer than 16 bits say (the lane starts that search with InitialSI).
constexpr uint32_t kKLineBytes = 128, kKLinePresent = 120;
#ifdef KJ_G_SMALL                                  // tests: intervals of more than two rows take the escape path
constexpr uint32_t kKLineMaxLen = 2u;
#else
constexpr uint32_t kKLineMaxLen = 0xffbfu;
#endif
constexpr uint32_t kKLineSingle = 0xffc0u, kKLineEscape = 0xffffu;
// line number of the word M whose letters are handed over from w[j-1] down to w[j-k+1]
KJ_HD uint32_t kline_code(uint32_t code, uint32_t c) { return code * 20u + (c - 1u); }
// what the lane keeps per lookup: line number << 6 | offset of the 16 bytes it loads for entry a, in units of 2 bytes
// (entry a sits at byte 6 (a-1); the load of the last one starts two bytes early so that it stays inside the line)
KJ_HD uint32_t kline_ref(uint32_t code, uint32_t a) { const uint32_t o = 3u * (a - 1u); return code << 6 | (o > 56u ? 56u : o); }
KJ_HD uint64_t kline_entry(const u128 &v, uint32_t ref) {          // the 48 bits of the entry from the 16 bytes loaded
  return ((ref & 63u) == 56u ? (v.x >> 16 | v.y << 48) : v.x) & 0xffffffffffffull;
}
// builds line `code` of a table of k-letter words from the word table kmer32 (index: letter matched first = most
// significant digit) and the BWT letters in blocks64
KJ_HD void kline_build_one(const DevIndex &ix, uint32_t k, uint64_t code, uint8_t *line) {
  uint64_t pw = 1;
  for (uint32_t q = 1; q < k; q++) pw *= 20u;              // 20^(k-1): digit of the letter matched first
  uint32_t out[32];
  for (int x = 0; x < 32; x++) out[x] = 0;
  uint16_t *o16 = reinterpret_cast<uint16_t *>(out);
  for (uint32_t a = 1; a <= 20u; a++) {
    const uint2 e = ix.kmer32[code + (uint64_t)(a - 1u) * pw];
    uint32_t l16 = 0;
    if (e.y != 0) {
      if (e.y == 1u) {
        const RankBlock64 &rb = ix.blocks64[e.x >> 6];
        const uint32_t s = e.x & 63u;
        const uint32_t c = (uint32_t)((rb.plane[0] >> s) & 1ull) | (uint32_t)((rb.plane[1] >> s) & 1ull) << 1 |
                           (uint32_t)((rb.plane[2] >> s) & 1ull) << 2 | (uint32_t)((rb.plane[3] >> s) & 1ull) << 3 |
                           (uint32_t)((rb.plane[4] >> s) & 1ull) << 4;
        l16 = kKLineSingle | c;
      } else l16 = e.y <= kKLineMaxLen ? e.y : kKLineEscape;
    }
    o16[3 * (a - 1u)] = (uint16_t)e.x; o16[3 * (a - 1u) + 1] = (uint16_t)(e.x >> 16); o16[3 * (a - 1u) + 2] = (uint16_t)l16;
  }
  uint32_t pres = 0;
  for (uint32_t b = 1; b <= 20u; b++) if (ix.kmer32[code * 20u + (b - 1u)].y != 0) pres |= 1u << (b - 1u);
  out[kKLinePresent / 4] = pres;
  uint32_t *dst = reinterpret_cast<uint32_t *>(line);
  for (int x = 0; x < 32; x++) dst[x] = out[x];
}

struct Params {
  int32_t mode;              // 0 MEM, 1 GREEDY
  uint32_t m;                // min_fragment_length
  uint32_t mismatches, min_score, seed_length;
  int32_t seg;
  uint32_t max_matches_SI, max_match_ids;
  uint32_t debug = 0;        // developer timing experiments only (KAIJU_GPU_DEBUG): parts of stage 1 skipped, results wrong
  uint32_t flags = 0;        // kParamXOrder | kParamProtein
};
// Params::flags
constexpr uint32_t kParamDeferLocate = 8u;   // MEM, second-generation narrow lane: reads with one or two longest matches leave them in the hit
                                             // record and k_mem_locate walks to their ids afterwards (mem_locate_read)
constexpr uint32_t kParamXOrder = 1u;    // kaijux: MEM matches of a fragment are visited in maxMatches' list order (see mem_lane)
constexpr uint32_t kParamProtein = 2u;   // reads are protein sequences (kaiju -p, kaijup): stage 1 = k_fragments_protein
constexpr uint32_t kParamLazySeg = 4u;   // MEM: fragments are searched unsplit first; the lanes report the fragments that hold
                                         // a longest match in Hit::reserved (kWin*) for the check pass
// Hit::reserved between the search and the check pass of the lazy SEG flow
constexpr uint32_t kWinMulti = 0x80000000u;   // more than one fragment holds a longest match
constexpr uint32_t kWinForce = 0xffffffffu;   // the read has to take the SEG pass whatever its fragments look like

// fragment descriptor (16 bytes)
struct Frag {
  uint32_t start;            // offset of the first residue relative to the read's peptide base
  uint32_t len;
  uint32_t key;              // MEM: length, GREEDY: BLOSUM62 diagonal score
  uint32_t flags;            // bit0: SEG-checked, bit1: removed (transient)
};

// SEG runs as its own pass over the fragments whose 12-windows reach the trigger entropy
// (stage 1 only detects them); one record per such fragment:
struct SegWork { uint32_t read, frag; };
constexpr int kSegRecRegions = 15;
struct SegRec {              // 64 bytes
  uint16_t n;                // number of low-complexity regions (ascending, merged)
  uint16_t overflow;
  uint16_t lr[kSegRecRegions][2];
};
static_assert(sizeof(SegRec) == 64, "SegRec is 64 bytes");
struct SegQueue {
  SegWork *items;
  SegRec *recs;
  uint32_t *count;
  uint32_t cap;
};
constexpr uint32_t kFragChecked = 1u, kFragRemoved = 2u, kFragSlotShift = 8;
constexpr uint32_t kNfragSegPending = 0x80000000u;   // nfrag[r] bit: read has fragments awaiting SEG

struct SIEntry {             // one maximal match (16 bytes)
  uint64_t lo;
  uint32_t len;              // interval length (int truncation as alloc_SI, bwt.c:180)
  uint32_t frag;
};

struct Hit {                 // == kaiju_gpu_hit
  uint32_t best, n_ids, flags, reserved;
  uint64_t taxid[kMaxIds];
};

// SEG tables built on the host (host_tables.cpp)
struct SegTables {
  const double *lnfact;      // ln(n!) rounded to 6 decimals, blast_seg.c:52-1308
  uint32_t lnfact_n;
  int64_t ent_g[13];         // fixed-point entropy contribution of a letter seen c times in a 12-window
  int64_t ent_locut, ent_hicut;   // thresholds in the same fixed-point scale
  int32_t ent_g32[17];       // (13 used) the same at a 32-bit scale: enough to decide H <= locut (stage 1's trigger test)
  int32_t ent_locut32;
};

// Constant tables (ConsumerThread.cpp:6-187); the kernels keep a copy in LDS.
struct ConstTables {
  int8_t b62[20][20];        // BLOSUM62 in aa2int order ARNDCQEGHILKMFPSTWYV
  uint8_t subst[20][19];     // blosum_subst as aa2int codes, ConsumerThread.cpp:10-30
  uint8_t codon_aa[64];      // codon (A,C,G,T=0..3; n0*16+n1*4+n2) -> aa2int code, 255 = stop
  uint8_t nuc[256];          // nuc2int (255 = not ACGTU)
  uint8_t aa_to_idx[20];     // aa2int code -> index-alphabet code (astruct->trans)
  uint8_t idx_to_aa[32];     // index-alphabet code -> aa2int code
  int8_t diag_idx[32];       // BLOSUM62 diagonal by index-alphabet code (blosum62diag, :61-80)
  int8_t b62_idx[20][20];    // b62[a][idx_to_aa[c]] at [a][c-1]: row of an aa2int code, columns by index-alphabet code
  uint8_t subst_rank[20][20];// position of the letter with index code c in subst[a] at [a][c-1] (255: a itself)
  uint8_t codon_idx[64];     // codon -> index-alphabet code of the amino acid, 0 = stop
};

struct ReadMeta {            // written by stage 1 (16 bytes): saves the search lanes the layout arithmetic
  uint64_t pep;              // offset of the read's peptide area in Batch::pep
  uint32_t frag;             // first fragment slot in Batch::frags
  uint32_t nfrag;            // number of fragments (| kNfragSegPending)
};

struct Batch {
  const uint8_t *seqs;       // ASCII nucleotides
  const uint64_t *off;       // [2n+1]
  uint32_t n_reads;
  int32_t paired;
  uint8_t *pep;              // six-frame translations, index-alphabet codes, 0 = stop
  Frag *frags;               // canonical fragment lists
  ReadMeta *meta;            // [n] where read r's peptides / fragment list live
  Hit *hits;                 // [n]
};

// layout helpers (closed form, no prefix sums needed) -------------------------
// peptide area of read r: <= 2*(len1+len2) + 12 bytes of strings after 8 bytes of front padding (build_fragments), or six
// strings of whole 16-byte units per mate, <= 2*(len1+len2) + 192 bytes (build_fragments_fast)
constexpr uint64_t kPepPerRead = 208;
// (the fast stage 1 writes six strings of whole 16-byte units per mate: 96 * (len / 48 + 1) bytes, at most 2 * len + 192 per
//  read with two mates; pep_base() rounds the start down by up to 15 bytes - ADVICE r02: tie the constant to that layout)
static_assert(kPepPerRead >= 2 * 96 + 15 + 1, "kPepPerRead must cover the unit padding of build_fragments_fast plus the rounding of pep_base");
KJ_HD uint64_t pep_base(const uint64_t *off, uint32_t r) { return ((2 * off[2 * (uint64_t)r] + 15) & ~15ull) + kPepPerRead * r + 16; }
// fragment slots of read r: a read has at most (2*(len1+len2)+12)/(m+1) disjoint fragments;
// twice that is reserved so that SEG pieces can sit next to their parents
KJ_HD uint64_t frag_base(const uint64_t *off, uint32_t r, uint32_t m) {
  return 2 * ((2 * off[2 * (uint64_t)r]) / (m + 1) + 7ull * r);
}
KJ_HD uint32_t frag_cap(const uint64_t *off, uint32_t r, uint32_t m) {
  return (uint32_t)(frag_base(off, r + 1, m) - frag_base(off, r, m));
}

KJ_HD uint32_t popc64(uint64_t x) { return (uint32_t)__builtin_popcountll(x); }

// wave-level helpers (the host emulation runs one lane at a time)
#if defined(__HIP_DEVICE_COMPILE__)
KJ_HD uint64_t kj_ballot(bool p) { return __ballot(p); }
KJ_HD uint32_t kj_lane() { return threadIdx.x & 63u; }
KJ_HD uint32_t kj_bcast(uint32_t v, uint32_t src_lane) { return (uint32_t)__shfl((int)v, (int)src_lane, 64); }
// number of lanes below this one whose bit is set in mask (v_mbcnt: no lane id, no lane mask to keep in registers)
KJ_HD uint32_t kj_rank_below(uint64_t mask) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
// value of lane src_lane, src_lane the same in all lanes (v_readlane)
KJ_HD uint32_t kj_bcast_uniform(uint32_t v, uint32_t src_lane) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)__builtin_amdgcn_readfirstlane((int)src_lane));
}
// bytes s .. s+3 (s = 0..3) of the eight bytes lo, hi (v_alignbyte_b32)
KJ_HD uint32_t kj_alignbyte(uint32_t hi, uint32_t lo, uint32_t s) { return __builtin_amdgcn_alignbyte(hi, lo, s); }
#else
KJ_HD uint32_t kj_alignbyte(uint32_t hi, uint32_t lo, uint32_t s) { return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8u * (s & 3u))); }
KJ_HD uint64_t kj_ballot(bool p) { return p ? 1ull : 0ull; }
KJ_HD uint32_t kj_lane() { return 0; }
KJ_HD uint32_t kj_bcast(uint32_t v, uint32_t) { return v; }
KJ_HD uint32_t kj_rank_below(uint64_t) { return 0; }
KJ_HD uint32_t kj_bcast_uniform(uint32_t v, uint32_t) { return v; }
#endif

// ----------------------------------------------------------------------------
// rank / LF on the packed index
// ----------------------------------------------------------------------------
// C[c] + #{ i < k : bwt[i] == c }  for c in 1..20  (== FMindex, compactfmi.c:267-307).
// P = uint32_t for indexes below 2^32 symbols (half the address/count arithmetic), else uint64_t.
template <class P>
KJ_HD P rank_p(const DevIndex &ix, uint32_t c, P k) {
  const RankBlock64 &b = ix.blocks64[(uint64_t)k >> 6];
  uint64_t m = (1ull << ((uint32_t)k & 63u)) - 1ull;
#pragma unroll
  for (int p = 0; p < 5; p++) m &= ((c >> p) & 1u) ? b.plane[p] : ~b.plane[p];
  const uint64_t base = ix.mb_base ? ix.mb_base[(size_t)((uint64_t)k >> ix.mb_shift) * 20 + (c - 1)] : 0ull;
  return (P)(base + b.cnt[c - 1] + popc64(m));
}
KJ_HD uint64_t rank_c(const DevIndex &ix, uint32_t c, uint64_t k) { return rank_p<uint64_t>(ix, c, k); }

KJ_HD uint32_t symbol_at(const DevIndex &ix, uint64_t k) {
  const RankBlock64 &b = ix.blocks64[k >> 6];
  const uint32_t s = (uint32_t)k & 63u;
  uint32_t c = 0;
#pragma unroll
  for (int p = 0; p < 5; p++) c |= (uint32_t)((b.plane[p] >> s) & 1ull) << p;
  return c;
}

// ---- quad-cooperative fetch of rank blocks (wide lanes; DESIGN.md 3.3, profiles/r03_randreach_*.txt) --------------------------
// On a footprint beyond a few GiB a lane that pulls its rank block with four loads of its own (16 + 16 + 8 + 4 bytes) gets a
// third of the line rate of one that issues ONE load per line (19 against 38-46 G lines/s): what costs is the number of
// far-reaching requests in flight, not the bytes.  Here the four lanes of a quad fetch the block of each of them in turn -
// lane 0..2 the three 16-byte pieces that hold the five bit planes, lane 3 the piece with the count of the letter asked for -
// straight into LDS (global_load_lds: destination = wave-uniform base + lane * 16, i.e. 64 contiguous bytes per query), one
// wave instruction = sixteen lines with four lanes each.  Two blocks (the two ends of an interval) per lane and iteration:
// eight instructions, as before, but a quarter of the distinct requests; the lane then reads its 2 x 52 bytes back from LDS.
constexpr int kCoopRound = 1040;                   // LDS bytes per round (64 lanes x 16 + 16 of skew against bank conflicts)
constexpr int kCoopBytesPerWave = 8 * kCoopRound;  // two blocks x four rounds
struct RankLines { u128 a01, a23; uint64_t a4; uint32_t ca; u128 b01, b23; uint64_t b4; uint32_t cb; };
#if defined(__HIP_DEVICE_COMPILE__)
template <int T> KJ_HD uint32_t kj_quad_bcast(uint32_t v) {         // value of lane T of the quad, in all four lanes
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, T * 0x55, 0xf, 0xf, false);
}
template <int T> KJ_HD void coop_round(const RankBlock64 *blk0, uint64_t blkA, uint64_t blkB, uint32_t cnt_off, uint8_t *lds_wave) {
  typedef __attribute__((address_space(3))) void lds_void;
  typedef __attribute__((address_space(1))) const void glb_void;
  const uint64_t bA = (uint64_t)kj_quad_bcast<T>((uint32_t)blkA) | (uint64_t)kj_quad_bcast<T>((uint32_t)(blkA >> 32)) << 32;
  const uint64_t bB = (uint64_t)kj_quad_bcast<T>((uint32_t)blkB) | (uint64_t)kj_quad_bcast<T>((uint32_t)(blkB >> 32)) << 32;
  const uint32_t co = kj_quad_bcast<T>(cnt_off);
  const uint32_t l4 = threadIdx.x & 3u;
  const uint32_t off = l4 < 3u ? l4 * 16u : co;
  const uint8_t *ga = reinterpret_cast<const uint8_t *>(blk0 + bA) + off, *gb = reinterpret_cast<const uint8_t *>(blk0 + bB) + off;
  __builtin_amdgcn_global_load_lds((glb_void *)ga, (lds_void *)(lds_wave + T * kCoopRound), 16, 0, 0);
  __builtin_amdgcn_global_load_lds((glb_void *)gb, (lds_void *)(lds_wave + (4 + T) * kCoopRound), 16, 0, 0);
}
// blkA / blkB: block numbers of the lane's two queries, cc: its letter (1..20); lds_wave: the wavefront's kCoopBytesPerWave bytes
KJ_HD void coop_fetch2(const RankBlock64 *blk0, uint64_t blkA, uint64_t blkB, uint32_t cc, uint8_t *lds_wave, RankLines &r) {
  const uint32_t cnt_at = 40u + 4u * (cc - 1u);                     // byte offset of the count in the block
  const uint32_t cnt_off = cnt_at > 112u ? 112u : cnt_at;          // the 16 bytes fetched for it stay inside the line
  coop_round<0>(blk0, blkA, blkB, cnt_off, lds_wave);
  coop_round<1>(blk0, blkA, blkB, cnt_off, lds_wave);
  coop_round<2>(blk0, blkA, blkB, cnt_off, lds_wave);
  coop_round<3>(blk0, blkA, blkB, cnt_off, lds_wave);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const uint32_t lane = threadIdx.x & 63u;
  const uint8_t *ma = lds_wave + (lane & 3u) * kCoopRound + (lane >> 2) * 64u, *mb = ma + 4 * kCoopRound;
  r.a01 = *reinterpret_cast<const u128 *>(ma); r.a23 = *reinterpret_cast<const u128 *>(ma + 16);
  r.a4 = *reinterpret_cast<const uint64_t *>(ma + 32);
  r.ca = *reinterpret_cast<const uint32_t *>(ma + 48 + (cnt_at - cnt_off));
  r.b01 = *reinterpret_cast<const u128 *>(mb); r.b23 = *reinterpret_cast<const u128 *>(mb + 16);
  r.b4 = *reinterpret_cast<const uint64_t *>(mb + 32);
  r.cb = *reinterpret_cast<const uint32_t *>(mb + 48 + (cnt_at - cnt_off));
}
#else
KJ_HD void coop_fetch2(const RankBlock64 *blk0, uint64_t blkA, uint64_t blkB, uint32_t cc, uint8_t *, RankLines &r) {
  const RankBlock64 *pa = blk0 + blkA, *pb = blk0 + blkB;            // (host emulation: one lane, plain reads)
  r.a01.x = pa->plane[0]; r.a01.y = pa->plane[1]; r.a23.x = pa->plane[2]; r.a23.y = pa->plane[3]; r.a4 = pa->plane[4]; r.ca = pa->cnt[cc - 1];
  r.b01.x = pb->plane[0]; r.b01.y = pb->plane[1]; r.b23.x = pb->plane[2]; r.b23.y = pb->plane[3]; r.b4 = pb->plane[4]; r.cb = pb->cnt[cc - 1];
}
#endif

// number of terminators in bwt[0, k)  (FMindexCurrent for letter 0; C[0] == 0)
KJ_HD uint64_t rank_term(const DevIndex &ix, uint64_t k) {
  uint64_t lo = 0, hi = ix.nseq;
  while (lo < hi) {
    const uint64_t mid = (lo + hi) >> 1;
    if (ix.term_pos[mid] < k) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// get_suffix (bwt.c:105-121) for row r: sequence number (in the order of the samples: rank among the sorted sequences) and
// offset of the suffix of row r.  smp_pos: the offset part of every sampled row (the sequence part is sa_iseq).  Rows below
// nseq (the suffixes that consist of a terminator only - no search ever asks for them, the text builder does) take their first
// LF step unconditionally.  Returns false where the reference would read beyond its sample array.
// `beyond` (optional): an index whose header counts one sample too few (KAIJU_IDX_WARN_SA_SHORT: the reference reads out of
// bounds at that row, its answer there is undefined) - the walk passes the missing sample as if the row were not sampled,
// goes on to the next one and says so; without `beyond` such a row makes the function fail as before.
KJ_HD bool suffix_of_row(const DevIndex &ix, const uint32_t *smp_pos, uint64_t r, uint32_t &iseq, uint32_t &pos, bool *beyond = nullptr) {
  const uint64_t check = (1ull << ix.chpt_exp) - 1ull;
  uint64_t k = r;
  uint32_t steps = 0;
  bool first = r < ix.nseq;
  for (;;) {
    if (!first && (k & check) == 0) {
      const uint64_t q = (k >> ix.chpt_exp) - ix.sa_skip;
      if (q < ix.n_sa) {
        iseq = ix.sa_iseq[q]; pos = smp_pos[q] + steps;
        return true;
      }
      if (!beyond) return false;
      *beyond = true;
    }
    first = false;
    const uint32_t c = symbol_at(ix, k);
    steps++;
    if (c == 0) { iseq = (uint32_t)rank_term(ix, k); pos = steps - 1u; return true; }
    k = rank_c(ix, c, k);
  }
}

// Text arrays of an index with 64-bit positions, built by walking every SEQUENCE once from its terminator suffix (rows
// 0 .. nseq-1) to its first letter: one LF step per index row and no sample offsets (suffix_of_row costs 2^chpt_exp - 1
// steps per row on average and wants the offset of every sampled row, which the wide layout does not keep).
//   seq_walk_len:  length of the sequence that ends at terminator row t, and its number (the rank of the terminator that the
//                  walk ends at, as in suffix_of_row)
//   seq_walk_fill: g_end = text position of that terminator (off[iseq + 1] in the layout of DevIndex::text); every row on
//                  the way writes its BWT letter in front of its suffix and - if it is a 2^tv_shift-th row - its position
KJ_HD uint64_t seq_walk_len(const DevIndex &ix, uint64_t t, uint32_t &iseq) {
  uint64_t k = t, n = 0;
  for (;;) {
    const uint32_t c = symbol_at(ix, k);
    if (c == 0) { iseq = (uint32_t)rank_term(ix, k); return n; }
    k = rank_c(ix, c, k);
    if (++n > ix.bwtlen) { iseq = 0xffffffffu; return n; }      // (a damaged index: no walk is longer than the text)
  }
}
KJ_HD void put_tpos5(uint8_t *a, uint64_t idx, uint64_t g) {
  uint8_t *e = a + idx * 5u;
  e[0] = (uint8_t)g; e[1] = (uint8_t)(g >> 8); e[2] = (uint8_t)(g >> 16); e[3] = (uint8_t)(g >> 24); e[4] = (uint8_t)(g >> 32);
}
// row_tax (optional): every row on the way also gets `dense`, the dense taxon index of the walk's sequence (DevIndex::row_tax
// of an index with 64-bit positions; text / tpos5 may then be null)
KJ_HD void seq_walk_fill(const DevIndex &ix, uint64_t t, uint64_t g_end, uint64_t n, uint8_t *text, uint8_t *tpos5, uint32_t tv_shift,
                         uint32_t *row_tax = nullptr, uint32_t dense = 0xffffffffu) {
  const uint64_t tvm = (1ull << tv_shift) - 1ull;
  uint64_t k = t, g = g_end;
  for (uint64_t step = 0; step <= n; step++) {
    if (tpos5 && (k & tvm) == 0) put_tpos5(tpos5, k >> tv_shift, g);
    if (row_tax) row_tax[k] = dense;
    const uint32_t c = symbol_at(ix, k);
    if (text) text[g - 1] = (uint8_t)c;
    if (c == 0) return;
    k = rank_c(ix, c, k);
    g--;
  }
}

// k-mer table lookup: letters w[0] (matched first, i.e. the rightmost residue) .. w[k-1]
KJ_HD uint32_t kmer_index(uint32_t idx, uint32_t c) { return idx * 20u + (c - 1u); }
KJ_HD void kmer_lookup(const DevIndex &ix, uint32_t idx, uint64_t &lo, uint64_t &hi) {
  if (ix.kmer32) { const uint2 e = ix.kmer32[idx]; lo = e.x; hi = (uint64_t)e.x + e.y; }
  else { const ulonglong2 e = ix.kmer64[idx]; lo = e.x; hi = e.x + e.y; }
}

// nucleotide triplet -> aa2int code (255 = stop).  Any non-ACGTU base gives a stop
// (codon_to_int + codon2aa default, ConsumerThread.cpp:111,869-875).
KJ_HD uint32_t codon_fwd(const ConstTables &t, const uint8_t *s) {
  const uint32_t a = t.nuc[s[0]], b = t.nuc[s[1]], c = t.nuc[s[2]];
  if ((a | b | c) > 3u) return 255u;
  return t.codon_aa[a * 16 + b * 4 + c];
}
KJ_HD uint32_t codon_rev(const ConstTables &t, const uint8_t *s) {
  const uint32_t a = t.nuc[s[2]], b = t.nuc[s[1]], c = t.nuc[s[0]];
  if ((a | b | c) > 3u) return 255u;
  return t.codon_aa[(3 - a) * 16 + (3 - b) * 4 + (3 - c)];
}

// ----------------------------------------------------------------------------
// SEG — bit-exact device version of SeqBufferSeg with Kaiju's parameters
// (blast_seg.c:1596-2332: window 12, locut 2.2, hicut 2.5, maxtrim 50, overlaps)
// ----------------------------------------------------------------------------
// The entropy of a 12-window only enters through the comparisons H <= locut and
// H > hicut.  H is a function of the partition of 12 formed by the letter counts
// (77 cases); the host evaluates the reference's floating-point expression for
// every partition and verifies that the integer score  sum_letters ent_g[count]
// classifies all of them identically (host_tables.cpp), so the device only adds
// integers.  The trim step (s_Trim/s_GetProb) is done in double precision in
// exactly the reference's operation order (adds/subtracts of table entries, no
// contraction).  Letters: the peptide buffer holds index codes 1..20; SEG only
// looks at letter identity, so code-1 (0..19) is used directly.
//
// Work decomposition on the device: a fragment is handled by one wavefront.  The scan
// for trigger windows is cheap and runs redundantly in all lanes (uniform control
// flow); the expensive part, s_Trim's search over every sub-window of a raw segment,
// is spread over the lanes (one sub-window per lane and round, letter counts packed
// into two 64-bit registers) and finished with a wave-wide lexicographic
// (probability, visiting order) minimum — which is what the reference's sequential
// "strictly smaller wins" loop computes.
constexpr int kSegWindow = 12, kSegDown = 5, kSegUp = 7, kSegMaxTrim = 50;
constexpr int kSegMaxRegions = 32;
constexpr int kSegPacked = 63;         // windows up to this length use the packed-count path
constexpr int kSegLnf = 64;            // ln(n!) entries mirrored in LDS

#define KJ_SL(s, i) ((uint32_t)(s)[(i)] - 1u)

struct SegCtx {                        // tables as the SEG code sees them (LDS copies on the device)
  const int64_t *ent_g;                // [13]
  int64_t ent_locut, ent_hicut;
  const double *lnf;                   // [kSegLnf] head of the ln(n!) table
  const double *lnfact;                // whole table (global memory)
  uint32_t lnfact_n;
};
KJ_HD SegCtx seg_ctx(const SegTables &st, const int64_t *ent_g, const double *lnf) {
  SegCtx c; c.ent_g = ent_g; c.ent_locut = st.ent_locut; c.ent_hicut = st.ent_hicut;
  c.lnf = lnf; c.lnfact = st.lnfact; c.lnfact_n = st.lnfact_n;
  return c;
}

// how the sub-windows of s_Trim are spread: one lane (host emulation) or a wavefront
// pref(): 2 * (kSegPacked + 1) 64-bit words shared by the lanes (the prefix counts of s_Trim's raw segment) or nullptr: every
// sub-window then counts its own letters; sync(): the lanes' writes to it are visible to each other
struct CoopSerial {
  uint64_t *prefix = nullptr;
  KJ_HD int lane() const { return 0; }
  KJ_HD int width() const { return 1; }
  KJ_HD void reduce_min(double &, int &) const {}
  KJ_HD uint64_t *pref() const { return prefix; }
  KJ_HD void sync() const {}
};

struct SegWin {              // 12-window: nibble-packed counts of the 20 letters + entropy score
  uint64_t c0, c1;
  int64_t score;
};
KJ_HD void segwin_add(SegWin &w, const SegCtx &cx, uint32_t a, int d) {
  const uint32_t old = a < 16 ? (uint32_t)(w.c0 >> (4 * a)) & 15u : (uint32_t)(w.c1 >> (4 * (a - 16))) & 15u;
  w.score += cx.ent_g[(int)old + d] - cx.ent_g[old];
  if (a < 16) w.c0 += (uint64_t)(int64_t)d << (4 * a);
  else w.c1 += (uint64_t)(int64_t)d << (4 * (a - 16));
}
template <class S>
KJ_HD void segwin_open(SegWin &w, const SegCtx &cx, const S &s, int start) {
  w.c0 = w.c1 = 0; w.score = 0;
  for (int i = 0; i < kSegWindow; i++) segwin_add(w, cx, KJ_SL(s, start + i), +1);
}
template <class S>
KJ_HD void segwin_shift(SegWin &w, const SegCtx &cx, const S &s, int start) {   // start -> start+1
  segwin_add(w, cx, KJ_SL(s, start), -1);
  segwin_add(w, cx, KJ_SL(s, start + kSegWindow), +1);
}

KJ_HD double kj_lnfact(const SegCtx &cx, int n) {
  if (n < kSegLnf) return cx.lnf[n];
  return cx.lnfact[(uint32_t)n < cx.lnfact_n ? n : cx.lnfact_n - 1];
}

KJ_HD double seg_finish_prob(double ans1, double ans2, int total) {
  const double totseq = ((double)total) * 2.9957322735539909;   // kLn20, blast_seg.c:2193
#if defined(__HIP_DEVICE_COMPILE__)
  return __dsub_rn(__dadd_rn(ans1, ans2), totseq);
#else
  volatile double t = ans1 + ans2;
  return t - totseq;
#endif
}

// number of 6-bit fields of x (10 fields) that are zero
KJ_HD int zero_fields6(uint64_t x) {
  const uint64_t LOW5 = 0x7DF7DF7DF7DF7DFull;     // low five bits of each field (bits 0..59)
  const uint64_t HI = 0x820820820820820ull;       // top bit of each field
  const uint64_t nzmask = (((x & LOW5) + LOW5) | x) & HI;
  return 10 - (int)popc64(nzmask);
}

// ln P0 of the sub-window s[i .. i+l) (s_GetProb blast_seg.c:1944-1967 with s_LnAss :1890-1933 and
// s_LnPerm :1864-1879 on the descending state vector), l <= kSegPacked
KJ_HD double seg_prob_of_counts(const SegCtx &cx, uint64_t c0, uint64_t c1, int l);
KJ_HD double seg_window_prob_packed(const SegCtx &cx, const uint8_t *s, int l, int i) {
  uint64_t c0 = 0, c1 = 0;             // 6-bit counts, letters 0..9 and 10..19
  for (int k = 0; k < l; k++) {
    const uint32_t a = KJ_SL(s, i + k);
    if (a < 10) c0 += 1ull << (6 * a); else c1 += 1ull << (6 * (a - 10));
  }
  return seg_prob_of_counts(cx, c0, c1, l);
}
// ... from the packed counts of the window's letters (l of them)
KJ_HD double seg_prob_of_counts(const SegCtx &cx, uint64_t c0, uint64_t c1, int l) {
  const uint64_t REP = 0x041041041041041ull;      // 1 in every field
  double ans1 = cx.lnf[20];
  double ans2 = kj_lnfact(cx, l);
  int nz = 0, rem = l;
  // the largest count: counts below 32 leave the top bit of their field free, so
  // "some field >= t" is a carry test; bisect (values above it would only be skipped one by one)
  int vtop = l;
  if (l < 32) {
    const uint64_t HI = 0x820820820820820ull;
    int lo = 1, hi = l;                            // invariant: some count >= lo, none > hi
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      const uint64_t T = REP * (uint64_t)(32 - mid);
      if (((c0 + T) | (c1 + T)) & HI) lo = mid; else hi = mid - 1;
    }
    vtop = lo;
  }
  uint64_t V = REP * (uint64_t)vtop;
  for (int v = vtop; v >= 1 && rem > 0; v--, V -= REP) {    // descending count value, V = v in every field
    const int n = zero_fields6(c0 ^ V) + zero_fields6(c1 ^ V);   // letters occurring exactly v times
    if (n) {
      ans1 -= cx.lnf[n];
      nz += n;
      rem -= n * v;
      const double lv = cx.lnf[v];
      for (int q = 0; q < n; q++) ans2 -= lv;
    }
  }
  if (nz > 0 && nz < 20) ans1 -= cx.lnf[20 - nz];
  return seg_finish_prob(ans1, ans2, l);
}
// the same for windows of any length
KJ_HD double seg_window_prob_generic(const SegCtx &cx, const uint8_t *s, int l, int i) {
  uint32_t comp[20];
  for (int a = 0; a < 20; a++) comp[a] = 0;
  for (int k = 0; k < l; k++) comp[KJ_SL(s, i + k)]++;
  double ans1 = cx.lnf[20];
  double ans2 = kj_lnfact(cx, l);
  int nz = 0;
  uint32_t bound = 0xFFFFFFFFu;                   // next distinct value strictly below bound
  for (;;) {
    uint32_t v = 0; int n = 0;
    for (int a = 0; a < 20; a++) if (comp[a] < bound && comp[a] > v) v = comp[a];
    if (v == 0) break;
    for (int a = 0; a < 20; a++) if (comp[a] == v) n++;
    ans1 -= cx.lnf[n]; nz += n;
    const double lv = kj_lnfact(cx, (int)v);
    for (int q = 0; q < n; q++) ans2 -= lv;
    bound = v;
  }
  if (nz > 0 && nz < 20) ans1 -= cx.lnf[20 - nz];
  return seg_finish_prob(ans1, ans2, l);
}

// s_Trim (blast_seg.c:1971-2015) on s[0..len): trimmed [lend, rend] relative to s.  The reference
// visits window lengths len, len-1, ..., minlen+1 and for each all start positions left to right,
// keeping a window only if its probability is strictly smaller than the best so far; with the
// visiting order t = d(d+1)/2 + i (d = len - l) that is the lexicographic minimum of (prob, t).
template <class Coop>
KJ_HD void seg_trim(const SegCtx &cx, const Coop &coop, const uint8_t *s, int len, int &lend_out, int &rend_out) {
  int minlen = 1;
  if (len - kSegMaxTrim > minlen) minlen = len - kSegMaxTrim;
  const int D = len - minlen;                     // number of window lengths
  const int W = D * (D + 1) / 2;                  // number of windows
  double best = 1.;
  int best_t = 0x7fffffff;
  int d = 0, base = 0;                            // base = d(d+1)/2 <= t
  // The counts of a sub-window are the difference of two PREFIX counts of the raw segment (fields of six bits, no borrows: a
  // prefix count never falls): pref[2j], pref[2j + 1] = the packed counts of s[0 .. j), written once per call - lane j counts
  // its j letters - where every one of the len (len - 1) / 2 sub-windows used to count its own (a third of the instructions
  // of this function, which is where the SEG pass spends its time: profiles/r06_seg_teams)
  uint64_t *pref = len <= kSegPacked ? coop.pref() : nullptr;
  if (pref) {
    coop.sync();                                  // (the previous call's prefix counts are no longer read)
    for (int j = coop.lane(); j <= len; j += coop.width()) {
      uint64_t c0 = 0, c1 = 0;
      for (int k = 0; k < j; k++) {
        const uint32_t a = KJ_SL(s, k);
        if (a < 10) c0 += 1ull << (6 * a); else c1 += 1ull << (6 * (a - 10));
      }
      pref[2 * j] = c0; pref[2 * j + 1] = c1;
    }
    coop.sync();
  }
  for (int t = coop.lane(); t < W; t += coop.width()) {
    while (base + d + 1 <= t) { base += d + 1; d++; }
    const int i = t - base, l = len - d;
    const double prob = pref ? seg_prob_of_counts(cx, pref[2 * (i + l)] - pref[2 * i], pref[2 * (i + l) + 1] - pref[2 * i + 1], l)
                        : l <= kSegPacked ? seg_window_prob_packed(cx, s, l, i) : seg_window_prob_generic(cx, s, l, i);
    if (prob < best) { best = prob; best_t = t; }
  }
  coop.reduce_min(best, best_t);
  int lend = 0, rend = len - 1;
  if (best_t != 0x7fffffff) {
    int dd = 0, bb = 0;
    while (bb + dd + 1 <= best_t) { bb += dd + 1; dd++; }
    lend = best_t - bb;
    rend = (len - dd) + lend - 1;
  }
  lend_out = lend; rend_out = rend;
}

// Entropy class of every 12-window of s[0..len), spread over the lanes of the team: bit 0: H <= locut
// (the window triggers), bit 1: H <= hicut (the window extends a segment).  cls[t] is the window
// starting at residue t.
template <class Coop>
KJ_HD void seg_classes(const SegCtx &cx, const Coop &coop, const uint8_t *s, int len, uint8_t *cls) {
  const int nwin = len - kSegWindow + 1;
  for (int t = coop.lane(); t < nwin; t += coop.width()) {
    SegWin w{0, 0, 0};
    segwin_open(w, cx, s, t);
    cls[t] = (uint8_t)((w.score <= cx.ent_locut ? 1 : 0) | (w.score <= cx.ent_hicut ? 2 : 0));
  }
}

// One level of s_SegSeq (blast_seg.c:2027-2113) on s[0..len).  At the top level
// (`TOP`) a trigger window lying left of its trimmed segment starts a second scan of
// the left remainder, of which only the LAST segment survives (:2093-2097: the head of
// the nested list is linked in, its tail is dropped); the nested scan therefore never
// needs to recurse itself.  Segments are appended in creation order.
// `cls` (optional): the classes of the windows of s as computed by seg_classes on the top-level
// string, cls[0] being the window that starts at s[0].
template <bool TOP, class Coop>
KJ_HD int seg_scan(const SegCtx &cx, const Coop &coop, const uint8_t *s, int len, int offset,
                   int32_t *beg, int32_t *end, int n, int cap, bool &overflow, const uint8_t *cls) {
  if (len < kSegWindow) return n;
  const int first = kSegDown, last = len - kSegUp;
  int lowlim = first;
  SegWin w{0, 0, 0};
  int wi = -1000;                        // position the window w is centred at
  for (int i = first; i <= last; i++) {
    int loi = i, hii = i;
    if (cls) {
      if (!(cls[i - first] & 1)) continue;                  // H > locut: no trigger
      while (loi - 1 >= lowlim && (cls[loi - 1 - first] & 2)) loi--;      // s_FindLow (:1810-1822)
      while (hii + 1 <= last && (cls[hii + 1 - first] & 2)) hii++;        // s_FindHigh (:1833-1845)
    } else {
      if (wi + 1 == i) segwin_shift(w, cx, s, wi - first);
      else if (wi != i) segwin_open(w, cx, s, i - first);
      wi = i;
      if (w.score > cx.ent_locut) continue;                 // H > locut: no trigger
      // s_FindLow (:1810-1822): down from i to lowlim while H <= hicut
      {
        SegWin b;
        while (loi - 1 >= lowlim) {
          segwin_open(b, cx, s, loi - 1 - first);
          if (b.score > cx.ent_hicut) break;
          loi--;
        }
      }
      // s_FindHigh (:1833-1845): up from i to last while H <= hicut
      {
        SegWin f = w;
        while (hii + 1 <= last) {
          segwin_shift(f, cx, s, hii - first);
          if (f.score > cx.ent_hicut) break;
          hii++;
        }
      }
    }
    const int rawleft = loi - kSegDown, rawright = hii + kSegUp - 1;
    int tl, tr;
    seg_trim(cx, coop, s + rawleft, rawright - rawleft + 1, tl, tr);
    const int leftend = rawleft + tl, rightend = rawleft + tr;
    if constexpr (TOP) {
      if (i + kSegUp - 1 < leftend) {
        // nested scan (a different instantiation: no recursion on the device); it keeps only
        // its most recent segment, which is all that survives in the reference
        int32_t tb[1], te[1];
        bool ov = false;
        const int k = seg_scan<false>(cx, coop, s + rawleft, leftend - rawleft, offset + rawleft, tb, te, 0, 1, ov,
                                      cls ? cls + rawleft : nullptr);
        if (k > 0) {
          if (n < cap) { beg[n] = tb[0]; end[n] = te[0]; n++; } else overflow = true;
        }
      }
    }
    if constexpr (TOP) {
      if (n < cap) { beg[n] = leftend + offset; end[n] = rightend + offset; n++; } else overflow = true;
    } else {
      beg[0] = leftend + offset; end[0] = rightend + offset; n = 1;   // only the last one is kept
    }
    i = hii < rightend + kSegDown ? hii : rightend + kSegDown;   // loop adds 1
    lowlim = i + 1;
  }
  return n;
}

// SeqBufferSeg (blast_seg.c:2278-2332).  Writes the merged regions in ascending order.
// `work` = 2 * kSegMaxRegions ints of scratch (LDS shared by the team on the device: every lane of
// the team computes the same values, so one copy serves all and no registers are spent on it).
// `cls` (optional): len bytes shared by the team for the window classes; `team_sync` makes them
// visible to all its lanes.
template <class Coop, class Sync>
KJ_HD int seg_regions(const SegCtx &cx, const Coop &coop, const uint8_t *s, int len,
                      int32_t *left, int32_t *right, bool &overflow, int32_t *work, uint8_t *cls, Sync &&team_sync,
                      int cap = kSegMaxRegions) {
  // (cap: capacity of the two scan lists in `work` and of left/right; the exact pass of long fragments passes more)
  int32_t *b = work, *e = work + cap;
  if (cls && len >= kSegWindow) {
    seg_classes(cx, coop, s, len, cls);
    team_sync();
  }
  const int n = seg_scan<true>(cx, coop, s, len, 0, b, e, 0, cap, overflow, cls);
  if (n == 0) return 0;
  // the reference's list is in reverse creation order; s_MergeSegs (:2122-2152, hilenmin 0)
  // walks it from the head and merges a node with its successor while they overlap
  int m = 0;
  int32_t cb = b[n - 1], ce = e[n - 1];
  for (int k = n - 2; k >= 0; k--) {
    if (cb - e[k] - 1 < 0) {
      if (ce < e[k]) ce = e[k];
      if (cb > b[k]) cb = b[k];
    } else {
      left[m] = cb; right[m] = ce; m++;
      cb = b[k]; ce = e[k];
    }
  }
  left[m] = cb; right[m] = ce; m++;
  // s_SegsToBlastSeqLoc (:2162-2171) reverses the list once more
  for (int a = 0, z = m - 1; a < z; a++, z--) {
    int32_t t = left[a]; left[a] = left[z]; left[z] = t;
    t = right[a]; right[a] = right[z]; right[z] = t;
  }
  return m;
}

// ----------------------------------------------------------------------------
// work distribution
// ----------------------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
KJ_HD uint32_t fetch_work(uint32_t *counter) { return atomicAdd(counter, 1u); }
KJ_HD uint32_t append_slot(uint32_t *counter) { return atomicAdd(counter, 1u); }
#else
KJ_HD uint32_t fetch_work(uint32_t *counter) { return (*counter)++; }
KJ_HD uint32_t append_slot(uint32_t *counter) { return (*counter)++; }
#endif

// ----------------------------------------------------------------------------
// stage 1: six-frame translation + canonical fragment list of one read
// (getAllFragmentsBits ConsumerThread.cpp:190-270; for MEM also the SEG split of
//  getNextFragment :272-342 applied eagerly, which SURVEY.md §8a shows equivalent)
// ----------------------------------------------------------------------------
// Where stage 1 keeps the peptides of a read while it works on them: byte p lives at
// base[(p >> 2) * row + (p & 3)].  row = 4 is plain linear memory; on the device the kernel stages
// the six frame strings in LDS, interleaved dword-wise over the 64 lanes of the wavefront
// (row = 256: conflict-free for lanes at the same position), and copies them out with 16-byte
// stores at the end, so that HBM only ever sees full 32-byte sectors instead of a trickle of byte
// stores that each cost a read-modify-write.
struct PepBuf {
  uint8_t *base;
  uint32_t row;
  KJ_HD uint8_t get(uint32_t p) const { return base[(size_t)(p >> 2) * row + (p & 3u)]; }
  KJ_HD void put(uint32_t p, uint8_t v) const { base[(size_t)(p >> 2) * row + (p & 3u)] = v; }
  KJ_HD uint32_t get32(uint32_t p) const { return *reinterpret_cast<const uint32_t *>(base + (size_t)(p >> 2) * row); }
};
struct PepView {             // a fragment inside a PepBuf, indexable like a byte string
  PepBuf b;
  uint32_t off;
  KJ_HD uint8_t operator[](int i) const { return b.get(off + (uint32_t)i); }
};

template <class S>
KJ_HD uint32_t diag_score(const ConstTables &t, const S &pep, uint32_t start, uint32_t len) {
  uint32_t s = 0;
  for (uint32_t i = 0; i < len; i++) { const uint32_t a = t.idx_to_aa[pep[start + i]]; s += (uint32_t)t.b62[a][a]; }
  return s;
}

// std::multimap<unsigned, Fragment*, std::greater>::emplace: behind every key >= f.key
KJ_HD void frag_insert(Frag *list, uint32_t &n, uint32_t cap, const Frag &f) {
  if (n >= cap) return;               // cannot happen: cap is a proven bound
  uint32_t pos = n;
  while (pos > 0 && list[pos - 1].key < f.key) { list[pos] = list[pos - 1]; pos--; }
  list[pos] = f;
  n++;
}

// nucleotides of a read for a walk with ascending positions: 16 bytes at a time, the next 16 already
// in flight (one lane reads one read: the loads of a wavefront touch 64 lines, better few and early)
struct NucReader {
  const uint8_t *s;
  uint32_t len, wq;
  u128 cur, nxt;
  KJ_HD u128 chunk(uint32_t q) const {                    // bytes [16q, 16q+16) of the read, zeros behind its end
    u128 v{0, 0};
    if (16 * q + 16 <= len) v = *reinterpret_cast<const u128_unaligned *>(s + 16 * q);
    else if (16 * q < len) {
      if (len >= 16) {
        // the last 16 bytes of the read, shifted down to where the chunk starts
        v = *reinterpret_cast<const u128_unaligned *>(s + len - 16);
        const uint32_t k = 16 * q + 16 - len;               // 1..15 bytes too far
        if (k >= 8) { v.x = v.y >> (8 * (k - 8)); v.y = 0; }
        else { v.x = (v.x >> (8 * k)) | (v.y << (64 - 8 * k)); v.y >>= 8 * k; }
      } else for (uint32_t x = 16 * q; x < len; x++) {
        if ((x & 15u) < 8) v.x |= (uint64_t)s[x] << (8 * (x & 7u)); else v.y |= (uint64_t)s[x] << (8 * (x & 7u));
      }
    }
    return v;
  }
  KJ_HD void open(const uint8_t *seq, uint32_t n) { s = seq; len = n; wq = 0; cur = chunk(0); nxt = chunk(1); }
  KJ_HD uint32_t at(uint32_t pos) {
    const uint32_t q = pos >> 4;
    if (q != wq) { wq = q; cur = nxt; nxt = chunk(q + 1); }
    const uint64_t w = (pos & 8u) ? cur.y : cur.x;
    return (uint32_t)(w >> (8u * (pos & 7u))) & 255u;
  }
};

// SEG trigger test of stage 1, incremental: the 12-window ending at the newest residue of a run.
// Counts of the 20 letters in nibbles, the last 12 letters in a 60-bit shift register, the entropy
// score at the 32-bit scale (host_tables.cpp checks that it classifies every window like libm).
struct TrigCtx { const int32_t *g32; int32_t locut32; };
struct TrigWin { uint64_t c0, c1, hist; int32_t score; };
KJ_HD void trig_reset(TrigWin &w) { w.c0 = w.c1 = w.hist = 0; w.score = 0; }
KJ_HD void trig_count(TrigWin &w, const TrigCtx &tc, uint32_t x, int d) {
  const uint32_t sh = 4u * (x & 15u);
  const uint32_t old = (uint32_t)((x < 16u ? w.c0 : w.c1) >> sh) & 15u;
  w.score += tc.g32[(int)old + d] - tc.g32[old];
  const uint64_t inc = (uint64_t)(int64_t)d << sh;
  if (x < 16u) w.c0 += inc; else w.c1 += inc;
}
// residue with index-alphabet code a joins a run that now has run_len residues: does the 12-window
// ending here reach the trigger entropy (H <= locut)?
KJ_HD bool trig_push(TrigWin &w, const TrigCtx &tc, uint32_t a, uint32_t run_len) {
  const uint32_t x = a - 1u;
  trig_count(w, tc, x, +1);
  if (run_len > (uint32_t)kSegWindow) trig_count(w, tc, (uint32_t)(w.hist >> 55) & 31u, -1);
  w.hist = (w.hist << 5 | x) & ((1ull << 60) - 1ull);
  return run_len >= (uint32_t)kSegWindow && w.score <= tc.locut32;
}

// a run of residues between two stops becomes a fragment if it is long enough (and, Greedy, scores
// enough).  Fragments are appended as they are found, flags = seq << 1 | trig, where seq is the
// moment at which the reference's getAllFragmentsBits would have emitted the run; build_fragments
// sorts by (key descending, seq ascending) = the order of the reference's multimap.
KJ_HD void emit_run(const Params &p, Frag *list, uint32_t &n, uint32_t cap, uint32_t start, uint32_t len,
                    uint32_t sum, uint32_t seq, bool trig) {
  if (len < p.m) return;
  Frag f; f.start = start; f.len = len; f.flags = seq << 1 | (trig ? 1u : 0u);
  if (p.mode == 1) {
    f.key = sum;                       // BLOSUM62 diagonal over the run
    if (f.key < p.min_score) return;
  } else f.key = len;
  if (n >= cap) return;               // cannot happen: cap is a proven bound
  if (!(p.debug & 1u)) list[n] = f;
  n++;
}

// Translate one mate into its six frame strings at pep[base..] and append its fragments.  ONE pass
// over the codon positions serves both strands: position cnt yields residue cnt/3 of forward frame
// cnt%3 (ConsumerThread.cpp:196-233) and, from the complemented codon read backwards, residue
// (top-cnt)/3 of reverse frame cnt%3 (:235-268, which walks cnt downwards: the reverse strings are
// filled from their end here).  codon_to_int / revcomp_codon_to_int (:869-875): a base that is not
// ACGTU makes the codon a stop.  The loop is unrolled over the three frames, all per-frame state is
// in registers, the SEG trigger test rides along.
KJ_HD void translate_mate(const ConstTables &t, const Params &p, const TrigCtx &tc, const uint8_t *s, uint32_t len,
                          uint32_t seq_base, const PepBuf &pep, uint32_t base, Frag *list, uint32_t &n, uint32_t cap) {
  const uint32_t fcap = len / 3 + 1;         // room of one frame string incl. closing stop
  const uint32_t top = len - 3, rbase = base + 3 * fcap;
  // emission moments: forward stop at cnt -> cnt, closing runs -> len + f; then the reverse walk,
  // stop visited at cnt -> (top - cnt), closing runs -> len + f
  const uint32_t seqF = seq_base, seqR = seq_base + len + 3;
  uint64_t dg0 = 0, dg1 = 0;                 // BLOSUM62 diagonal by index-alphabet code, 4 bits each
  if (p.mode == 1) {
    for (int x = 0; x < 16; x++) dg0 |= (uint64_t)((uint32_t)t.diag_idx[x] & 15u) << (4 * x);
    for (int x = 16; x < 32; x++) dg1 |= (uint64_t)((uint32_t)t.diag_idx[x] & 15u) << (4 * (x - 16));
  }
  uint32_t F_start[3], F_len[3], F_sum[3], R_len[3], R_sum[3], R_pend[3], R0[3];
  bool F_trig[3], R_trig[3];
  TrigWin Fw[3], Rw[3];
#pragma unroll
  for (int g = 0; g < 3; g++) {
    F_start[g] = base + (uint32_t)g * fcap; F_len[g] = F_sum[g] = 0; F_trig[g] = false; trig_reset(Fw[g]);
    R_len[g] = R_sum[g] = 0; R_trig[g] = false; trig_reset(Rw[g]);
    R_pend[g] = seqR + len + (uint32_t)g;    // the run at the end of the string is closed after the walk
    R0[g] = (top - (uint32_t)g) / 3;         // string index of the reverse residue of position g (len >= 3m: top >= 2)
  }
  NucReader nr;
  nr.open(s, len);
  uint32_t a = t.nuc[nr.at(0)], bb = t.nuc[nr.at(1)];
  uint32_t q = 0;
  // Three positions (one per frame) per iteration, written in phases so that the table lookups of
  // a phase are in flight together: nucleotide codes, the six codons, the entropy table entries of
  // the six windows; only then the updates, and the (rare) stops last.
  for (uint32_t cnt = 0; cnt <= top; cnt += 3, q++) {
    bool ok[3];
    uint32_t cN[3];
#pragma unroll
    for (int g = 0; g < 3; g++) {
      ok[g] = cnt + (uint32_t)g <= top;
      cN[g] = (p.debug & 32u) ? ((cnt * 7u + (uint32_t)g * 3u) >> 2) & 3u
                              : t.nuc[nr.at(cnt + (uint32_t)g + 2)];        // (positions past the end read as 0 bytes: not ACGTU)
    }
    const uint32_t n0[3] = {a, bb, cN[0]}, n1[3] = {bb, cN[0], cN[1]}, n2[3] = {cN[0], cN[1], cN[2]};
    uint32_t aa[6];                                         // [g]: forward, [3+g]: reverse
#pragma unroll
    for (int g = 0; g < 3; g++) {
      const bool bad = (n0[g] | n1[g] | n2[g]) > 3u;
      const uint32_t vf = t.codon_idx[(n0[g] * 16 + n1[g] * 4 + n2[g]) & 63u];
      const uint32_t vr = t.codon_idx[(63u - (n2[g] * 16 + n1[g] * 4 + n0[g])) & 63u];
      aa[g] = bad ? 0u : vf;
      aa[3 + g] = bad ? 0u : vr;
    }
    a = cN[1]; bb = cN[2];
    // entropy of the 12-windows: letter x joins, letter y (12 back) leaves once the run is longer than 12
    int32_t dsc[6];
    uint32_t xs[6], ys[6];
    bool rem[6];
    if (p.seg) {
#pragma unroll
      for (int h = 0; h < 6; h++) {
        const TrigWin &w = h < 3 ? Fw[h] : Rw[h - 3];
        const uint32_t rl = (h < 3 ? F_len[h] : R_len[h - 3]) + 1u;
        const uint32_t x = (aa[h] - 1u) & 31u, y = (uint32_t)(w.hist >> 55) & 31u;
        const uint32_t cx = (uint32_t)((x < 16u ? w.c0 : w.c1) >> (4u * (x & 15u))) & 15u;
        const uint32_t cy = (uint32_t)((y < 16u ? w.c0 : w.c1) >> (4u * (y & 15u))) & 15u;
        rem[h] = rl > (uint32_t)kSegWindow;
        const int32_t dA = tc.g32[cx + 1u] - tc.g32[cx];
        const int32_t dR = tc.g32[(cy - 1u) & 15u] - tc.g32[cy];
        const bool same = rem[h] && x == y;                 // the same letter joins and leaves: nothing changes
        dsc[h] = same ? 0 : dA + (rem[h] ? dR : 0);
        xs[h] = x; ys[h] = y;
        if (same) rem[h] = false;
        if (same) xs[h] = 32u;                               // (32: no count changes)
      }
    }
#pragma unroll
    for (int h = 0; h < 6; h++) {
      const int g = h % 3;
      const bool fwd = h < 3;
      uint32_t &rlen = fwd ? F_len[g] : R_len[g];
      uint32_t &rsum = fwd ? F_sum[g] : R_sum[g];
      bool &trig = fwd ? F_trig[g] : R_trig[g];
      TrigWin &w = fwd ? Fw[g] : Rw[g];
      const uint32_t fpos = base + (uint32_t)g * fcap + q, rpos = rbase + (uint32_t)g * fcap + (R0[g] - q);
      const uint32_t pos = fwd ? fpos : rpos;
      if (ok[g]) {
        if (!(p.debug & 2u)) pep.put(pos, (uint8_t)aa[h]);
        if (aa[h] != 0) {
          rlen++;
          if (p.mode == 1) rsum += (uint32_t)(((aa[h] & 16u) ? dg1 : dg0) >> (4u * (aa[h] & 15u))) & 15u;
          if (p.seg) {
            w.score += dsc[h];
            if (xs[h] < 32u) { const uint64_t inc = 1ull << (4u * (xs[h] & 15u)); if (xs[h] < 16u) w.c0 += inc; else w.c1 += inc; }
            if (rem[h]) { const uint64_t dec = 1ull << (4u * (ys[h] & 15u)); if (ys[h] < 16u) w.c0 -= dec; else w.c1 -= dec; }
            w.hist = (w.hist << 5 | ((aa[h] - 1u) & 31u)) & ((1ull << 60) - 1ull);
            trig = trig || (rlen >= (uint32_t)kSegWindow && w.score <= tc.locut32);
          }
        }
      }
    }
    // stops close runs
#pragma unroll
    for (int g = 0; g < 3; g++) {
      if (ok[g] && aa[g] == 0) {
        const uint32_t fpos = base + (uint32_t)g * fcap + q;
        emit_run(p, list, n, cap, F_start[g], F_len[g], F_sum[g], seqF + cnt + (uint32_t)g, F_trig[g]);
        F_start[g] = fpos + 1; F_len[g] = F_sum[g] = 0; F_trig[g] = false; trig_reset(Fw[g]);
      }
      if (ok[g] && aa[3 + g] == 0) {
        // the run behind this stop (string indices rpos+1 ..) is complete; the reference emits it
        // when its walk reaches the stop in front of it, or after the walk
        const uint32_t rpos = rbase + (uint32_t)g * fcap + (R0[g] - q);
        emit_run(p, list, n, cap, rpos + 1, R_len[g], R_sum[g], R_pend[g], R_trig[g]);
        R_pend[g] = seqR + (top - (cnt + (uint32_t)g));
        R_len[g] = R_sum[g] = 0; R_trig[g] = false; trig_reset(Rw[g]);
      }
    }
  }
#pragma unroll
  for (int g = 0; g < 3; g++) {
    const uint32_t nres = R0[g] + 1;         // codons of frame g
    emit_run(p, list, n, cap, F_start[g], F_len[g], F_sum[g], seqF + len + (uint32_t)g, F_trig[g]);
    pep.put(base + (uint32_t)g * fcap + nres, 0);
    emit_run(p, list, n, cap, rbase + (uint32_t)g * fcap, R_len[g], R_sum[g], R_pend[g], R_trig[g]);
    pep.put(rbase + (uint32_t)g * fcap + nres, 0);
  }
}

// does any 12-window of the fragment reach the trigger entropy (H <= locut)?  Exactly then
// SeqBufferSeg reports at least one region (s_SegSeq, blast_seg.c:2061).
template <class S>
KJ_HD bool seg_triggers(const SegCtx &cx, const S &s, int len) {
  if (len < kSegWindow) return false;
  SegWin w{0, 0, 0};
  segwin_open(w, cx, s, 0);
  if (w.score <= cx.ent_locut) return true;
  for (int start = 0; start + kSegWindow < len; start++) {
    segwin_shift(w, cx, s, start);
    if (w.score <= cx.ent_locut) return true;
  }
  return false;
}

// the SEG pass proper: regions of one flagged fragment -> record (written by lane 0 of the team)
// `stage` (optional, `stage_cap` bytes, shared by the lanes of the team) receives a copy of the
// fragment so that the scan does not go to device memory for every residue.
template <class Coop, class Sync>
KJ_HD void seg_compute(const SegCtx &cx, const Coop &coop, const Batch &b, const Params &p, const SegQueue &sq,
                       uint32_t slot, uint8_t *stage, uint32_t stage_cap, int32_t *work, uint8_t *cls, Sync &&team_sync) {
  const SegWork wk = sq.items[slot];
  const ReadMeta rm = b.meta[wk.read];
  const Frag f = b.frags[rm.frag + wk.frag];
  const uint8_t *pep = b.pep + rm.pep;
  const uint8_t *src = pep + f.start;
  if (stage && f.len <= stage_cap) {
    team_sync();                                            // the previous fragment is no longer read
    for (uint32_t x = (uint32_t)coop.lane(); x < f.len; x += (uint32_t)coop.width()) stage[x] = src[x];
    team_sync();
    src = stage;
  }
  // scratch of the team: [0, 2R) scan lists, [2R, 4R) merged regions (R = kSegMaxRegions)
  int32_t *left = work + 2 * kSegMaxRegions, *right = work + 3 * kSegMaxRegions;
  bool ov = false;
  // (cls holds stage_cap bytes; the previous fragment's classes are no longer read: the syncs above)
  const int n = seg_regions(cx, coop, src, (int)f.len, left, right, ov, work, (stage && f.len <= stage_cap) ? cls : nullptr, team_sync);
  if (coop.lane() != 0) return;
  SegRec rec;
  rec.overflow = (ov || n > kSegRecRegions || f.len > 65535u) ? 1 : 0;
  rec.n = (uint16_t)(n > kSegRecRegions ? kSegRecRegions : n);
  for (int k = 0; k < kSegRecRegions; k++) {
    rec.lr[k][0] = k < n ? (uint16_t)left[k] : 0;
    rec.lr[k][1] = k < n ? (uint16_t)right[k] : 0;
  }
  sq.recs[slot] = rec;
}

// getNextFragment's SEG split (ConsumerThread.cpp:285-339) of ONE fragment given its regions:
// hands the unmasked pieces (length > m strictly, :298,312; Greedy: score >= min_score) to sink.
template <class Sink>
KJ_HD void seg_split(const ConstTables &t, const Params &p, const SegRec &rec, const uint8_t *pep,
                     const Frag &f, Sink &&sink) {
  const int nreg = rec.n;
  uint64_t start = 0;
  for (int r = 0; r <= nreg; r++) {
    // size_t arithmetic as in the reference (a region starting left of `start` wraps)
    const uint64_t length = (r < nreg ? (uint64_t)rec.lr[r][0] : (uint64_t)f.len) - start;
    if (length > p.m) {
      const uint64_t avail = start <= f.len ? f.len - start : 0;
      const uint32_t take = (uint32_t)(length < avail ? length : avail);   // std::string::substr clamps
      Frag q; q.start = f.start + (uint32_t)start; q.len = take; q.flags = kFragChecked;
      if (p.mode == 1) {
        q.key = diag_score(t, pep, q.start, take);
        if (q.key >= p.min_score) sink(q);
      } else { q.key = (uint32_t)length; sink(q); }
    }
    if (r < nreg) start = (uint64_t)rec.lr[r][1] + 1;
  }
}
// the same from a list of any length (exact pass): (left, right) pairs
struct BigRegs { const int32_t *lr; int cnt; };
template <class Sink>
KJ_HD void seg_split_regs(const ConstTables &t, const Params &p, const BigRegs &regs, const uint8_t *pep,
                          const Frag &f, Sink &&sink) {
  const int nreg = regs.cnt;
  uint64_t start = 0;
  for (int r = 0; r <= nreg; r++) {
    const uint64_t length = (r < nreg ? (uint64_t)(uint32_t)regs.lr[2 * r] : (uint64_t)f.len) - start;
    if (length > p.m) {
      const uint64_t avail = start <= f.len ? f.len - start : 0;
      const uint32_t take = (uint32_t)(length < avail ? length : avail);
      Frag q; q.start = f.start + (uint32_t)start; q.len = take; q.flags = kFragChecked;
      if (p.mode == 1) {
        q.key = diag_score(t, pep, q.start, take);
        if (q.key >= p.min_score) sink(q);
      } else { q.key = (uint32_t)length; sink(q); }
    }
    if (r < nreg) start = (uint64_t)(uint32_t)regs.lr[2 * r + 1] + 1;
  }
}

// ----------------------------------------------------------------------------
// The exact pass.  A SegRec holds 15 regions and 16-bit positions; a fragment that needs more (long,
// low-complexity-rich proteins or contigs) is marked `overflow` by the SEG pass.  Reads with such a fragment
// are done again after the main and retry passes by a few small kernels (capi.hip: k_redo_*): stage 1 into a
// queue of its own, SEG with lists of any length into a pool of (left, right) pairs, the split from those,
// the search by the first-generation lanes.  Nothing of this is on the hot path: the kernels of the main
// pass are untouched, the exact pass finds an empty list for all but unusual inputs.
// ----------------------------------------------------------------------------
struct BigSeg {
  uint2 *index;              // [slot of the exact pass's queue] -> (first pair in lr, number of regions)
  int32_t *lr;               // pool of (left, right) pairs
  uint32_t *count;           // pairs handed out
  uint32_t cap;              // pairs in the pool
};
constexpr uint32_t kBigSegLost = 0xffffffffu;     // index[slot].y: the pool was too small (reported, KAIJU_HIT_INEXACT)
#if defined(__HIP_DEVICE_COMPILE__)
KJ_HD uint32_t append_many(uint32_t *counter, uint32_t n) { return atomicAdd(counter, n); }
#else
KJ_HD uint32_t append_many(uint32_t *counter, uint32_t n) { const uint32_t v = *counter; *counter += n; return v; }
#endif
// SEG of one queued fragment with lists of any length; `work` = 4 * cap ints and `cls` = f.len bytes of scratch of the
// team (device memory), cap > 2 * f.len
template <class Coop, class Sync>
KJ_HD void seg_compute_big(const SegCtx &cx, const Coop &coop, const Batch &b, const SegQueue &sq, const BigSeg &big,
                           uint32_t slot, int32_t *work, int cap, uint8_t *cls, uint32_t *err_flags, Sync &&team_sync) {
  const SegWork wk = sq.items[slot];
  const ReadMeta rm = b.meta[wk.read];
  const Frag f = b.frags[rm.frag + wk.frag];
  const uint8_t *src = b.pep + rm.pep + f.start;
  int32_t *left = work + 2 * cap, *right = work + 3 * cap;
  bool ov = false;
  team_sync();                                              // the previous fragment's lists are no longer read
  const int n = seg_regions(cx, coop, src, (int)f.len, left, right, ov, work, cls, team_sync, cap);
  if (coop.lane() != 0) return;
  uint2 ent; ent.x = 0; ent.y = kBigSegLost;
  if (!ov) {
    const uint32_t off = append_many(big.count, (uint32_t)n);
    if ((uint64_t)off + (uint32_t)n <= big.cap) {
      for (int k = 0; k < n; k++) { big.lr[2 * ((size_t)off + k)] = left[k]; big.lr[2 * ((size_t)off + k) + 1] = right[k]; }
      ent.x = off; ent.y = (uint32_t)n;
    }
  }
  if (ent.y == kBigSegLost && err_flags) *err_flags |= 4u;
  big.index[slot] = ent;
}

struct FragAppend {
  Frag *dst; uint32_t *n; uint32_t cap;
  KJ_HD void operator()(const Frag &q) const { if (*n < cap) dst[(*n)++] = q; }
};

// stage 1 for read r: translation, fragment list in queue order, SEG trigger detection.
// `stage` is the lane's staging area (LDS on the device, stage_words dwords) or nullptr (peptides
// are written straight to their place: long reads, host emulation of that path).
KJ_HD void build_fragments(const ConstTables &t, const Params &p, const TrigCtx &tc, const Batch &b,
                           const SegQueue &sq, uint32_t r, uint32_t *err_flags, uint8_t *stage, uint32_t stage_row,
                           uint32_t stage_words) {
  const uint64_t o0 = b.off[2 * (uint64_t)r], o1 = b.off[2 * (uint64_t)r + 1], o2 = b.off[2 * (uint64_t)r + 2];
  const uint32_t len1 = (uint32_t)(o1 - o0), len2 = (uint32_t)(o2 - o1);
  const uint32_t m3 = p.m * 3;
  Frag *list = b.frags + frag_base(b.off, r, p.m);
  const uint32_t cap = frag_cap(b.off, r, p.m);
  const uint64_t pbase = pep_base(b.off, r);
  PepBuf pep;
  if (stage) { pep.base = stage; pep.row = stage_row; } else { pep.base = b.pep + pbase; pep.row = 4; }
  uint32_t n = 0, pending = 0;
  // length gate, ConsumerThread.cpp:647-654
  const bool skip = b.paired ? (len1 < m3 && len2 < m3) : (len1 < m3);
  if (!skip) {
    const uint32_t seq2 = 2 * len1 + 6, seq_end = seq2 + 2 * len2 + 6;
    if (len1 >= m3) translate_mate(t, p, tc, b.seqs + o0, len1, 0, pep, 0, list, n, cap);
    if (b.paired && len2 >= m3) translate_mate(t, p, tc, b.seqs + o1, len2, seq2, pep, 6 * (len1 / 3 + 1), list, n, cap);
    if (stage && !(p.debug & 8u)) {
      // copy the strings out with 16-byte stores (the area of a read is 16-byte aligned)
      const uint32_t used = 6 * (len1 / 3 + 1) + (b.paired ? 6 * (len2 / 3 + 1) : 0);
      u128 *dst = reinterpret_cast<u128 *>(b.pep + pbase);
      for (uint32_t q = 0; q * 16 < used; q++) {
        u128 v;
        v.x = (uint64_t)pep.get32(16 * q) | (uint64_t)pep.get32(16 * q + 4) << 32;
        v.y = (uint64_t)pep.get32(16 * q + 8) | (uint64_t)pep.get32(16 * q + 12) << 32;
        dst[q] = v;
      }
    }
    if (p.debug & 16u) n = 0;
    // queue order: std::multimap<unsigned, Fragment*, std::greater>::emplace puts a fragment behind
    // every key >= its own: by descending key, equal keys in the order of emission
    uint32_t sb = 1;
    while ((1u << sb) <= seq_end) sb++;
    const uint32_t kbound = 11u * ((len1 > len2 ? len1 : len2) / 3u + 1u);     // no key exceeds this
    // bit 0 of the flags so far: some 12-window reaches the trigger entropy, i.e. SeqBufferSeg would
    // report at least one region (s_SegSeq, blast_seg.c:2061): those fragments go to the SEG pass
    auto final_flags = [&](uint32_t trig, uint32_t k) -> uint32_t {
      if (!p.seg) return 0u;
      if (!trig) return kFragChecked;                        // SEG would report nothing for this fragment
      const uint32_t slot = append_slot(sq.count);
      if (slot >= sq.cap) { if (err_flags) *err_flags |= 2u; return kFragChecked; }
      SegWork wk; wk.read = r; wk.frag = k;
      sq.items[slot] = wk;
      pending = kNfragSegPending;
      return (slot + 1) << kFragSlotShift;
    };
    if (stage && n <= stage_words / 2 && sb <= 20 && kbound < (1u << (31 - sb))) {
      // rank by counting in the (now free) staging row:
      // dword k = key << (sb+1) | (2^sb - 1 - seq) << 1 | trig, dword half+k = start << 16 | len
      uint32_t *row = reinterpret_cast<uint32_t *>(stage);
      const uint32_t rw = stage_row / 4, half = stage_words / 2, smax = (1u << sb) - 1u;
      for (uint32_t k0 = 0; k0 < n; k0 += 4) {               // four entries in flight at a time
        Frag f[4];
#pragma unroll
        for (int x = 0; x < 4; x++) f[x] = list[k0 + (uint32_t)x < n ? k0 + (uint32_t)x : k0];
#pragma unroll
        for (int x = 0; x < 4; x++) {
          const uint32_t k = k0 + (uint32_t)x;
          if (k < n) {
            row[(size_t)k * rw] = f[x].key << (sb + 1) | (smax - (f[x].flags >> 1)) << 1 | (f[x].flags & 1u);
            row[(size_t)(half + k) * rw] = f[x].start << 16 | f[x].len;
          }
        }
      }
      for (uint32_t k = 0; k < n; k++) {
        const uint32_t ck = row[(size_t)k * rw];
        uint32_t rank = 0;
        for (uint32_t q = 0; q < n; q++) rank += row[(size_t)q * rw] > ck ? 1u : 0u;
        const uint32_t sl = row[(size_t)(half + k) * rw];
        Frag f; f.start = sl >> 16; f.len = sl & 0xffffu; f.key = ck >> (sb + 1); f.flags = final_flags(ck & 1u, rank);
        list[rank] = f;
      }
    } else {
      for (uint32_t k = 1; k < n; k++) {                    // insertion sort in place (long reads)
        const Frag f = list[k];
        uint32_t pos = k;
        while (pos > 0 && (list[pos - 1].key < f.key || (list[pos - 1].key == f.key && list[pos - 1].flags > f.flags))) {
          list[pos] = list[pos - 1]; pos--;
        }
        list[pos] = f;
      }
      for (uint32_t k = 0; k < n; k++) list[k].flags = final_flags(list[k].flags & 1u, k);
    }
  }
  ReadMeta rm; rm.pep = pbase; rm.frag = (uint32_t)frag_base(b.off, r, p.m); rm.nfrag = n | pending;
  b.meta[r] = rm;
}

// ----------------------------------------------------------------------------
// stage 1, fast path (mates of up to kS1MaxLen nucleotides; build_fragments above stays the general path).
//
// Same results as build_fragments - six frame strings per mate (getAllFragmentsBits, ConsumerThread.cpp:190-270), the
// fragment list in the order of the reference's multimap, optionally the SEG trigger flags - produced with a small
// fraction of its instructions (profiles/r02_recon: the old kernel issued 430 VALU + 200 SALU wave instructions per read
// at 1.6 waves per SIMD):
//   * no per-residue control flow at all: a nucleotide becomes a 3-bit code (one LDS lookup), three codes index two
//     512-byte tables (codon -> residue of the forward / reverse strand, 0 for stops and for codons with a base that is
//     not ACGTU, codon_to_int :869-875), the residues are packed four to a register;
//   * a frame string is a row of 16-residue UNITS and every unit is stored once, as one aligned 16-byte store straight
//     from registers (no LDS staging, so nothing limits the occupancy but registers): forward strings start at a unit
//     boundary, reverse strings - produced from their end, :235-268 walk the strand downwards - END at one;
//   * stops are not acted upon inside the loop; the runs between them are taken from a bit mask per frame afterwards
//     (bit tricks), with the emission moments of the reference in closed form, and ranked in a small LDS list;
//   * TRIG (Greedy; MEM looks at SEG lazily, see mem_lane2): a second pass over the six strings of a mate, one at a time
//     from a small per-lane LDS buffer, keeps the entropy of the 12-window ending at every residue (trig_scan).
// ----------------------------------------------------------------------------
struct Stage1Tables {        // built on the host (host_tables.cpp: build_stage1_tables); the kernel keeps a copy in LDS
  uint8_t nuc3[256];         // nucleotide -> 0..3 (nuc2int, ConsumerThread.cpp:6-9), 4 = not ACGTU
  uint8_t tf[512];           // three codes, the first nucleotide in bits 6..8 -> index-alphabet code of the codon, 0 = stop / invalid
  uint8_t tr[512];           // ... of the reverse-complemented codon (revcomp_codon_to_int)
  int32_t dtab[32];          // [c], c = 0..11: change of the (2^26-scaled) 12-window entropy when a letter present c times
                             // joins (leaves with c left); [13..25]: the same for the stop "letter": kS1Big
  uint8_t diag[32];          // BLOSUM62 diagonal by index-alphabet code (0 for the stop)
  int32_t locut32;           // SegTables::ent_locut32
  int32_t pad[3];
};
static_assert(sizeof(Stage1Tables) % 16 == 0, "Stage1Tables is copied in 16-byte pieces");
constexpr int kS1Units = 4;                          // 16-residue units per frame string: the benchmark's instantiation (150-bp reads)
constexpr uint32_t kS1MaxLen = 48 * kS1Units - 1;    // len / 3 + 1 <= 16 * kS1Units
constexpr int kS1UnitsLong = 6;                      // ... and the one for mates of 192 .. 287 nucleotides (250-bp MiSeq reads): the
constexpr uint32_t kS1MaxLenLong = 48 * kS1UnitsLong - 1;   // per-frame masks ("residue is no stop") are 128 bits wide there
// the masks of the fast stage 1: one bit per codon of a frame - 64 bits for kS1Units, two words for kS1UnitsLong
struct Mask128 { uint64_t lo, hi; };
KJ_HD uint64_t mk_zero(uint64_t) { return 0; }
KJ_HD Mask128 mk_zero(Mask128) { return Mask128{0, 0}; }
KJ_HD void mk_or16(uint64_t &m, uint32_t bits, uint32_t b) { m |= (uint64_t)bits << (16u * b); }
KJ_HD void mk_or16(Mask128 &m, uint32_t bits, uint32_t b) { if (b < 4u) m.lo |= (uint64_t)bits << (16u * b); else m.hi |= (uint64_t)bits << (16u * (b - 4u)); }
KJ_HD void mk_setbit(uint64_t &m, uint32_t k, bool v) { m |= (uint64_t)(v ? 1u : 0u) << k; }
KJ_HD void mk_setbit(Mask128 &m, uint32_t k, bool v) { if (k < 64u) m.lo |= (uint64_t)(v ? 1u : 0u) << k; else m.hi |= (uint64_t)(v ? 1u : 0u) << (k - 64u); }
KJ_HD bool mk_any(uint64_t m) { return m != 0; }
KJ_HD bool mk_any(const Mask128 &m) { return (m.lo | m.hi) != 0; }
KJ_HD uint64_t mk_shr(uint64_t m, uint32_t n) { return m >> n; }            // (n < 64 at every call: erosion steps, a + 11 <= 63)
KJ_HD Mask128 mk_shr(const Mask128 &m, uint32_t n) {
  if (n == 0u) return m;
  if (n >= 128u) return Mask128{0, 0};
  if (n >= 64u) return Mask128{m.hi >> (n - 64u), 0};
  return Mask128{(m.lo >> n) | (m.hi << (64u - n)), m.hi >> n};
}
KJ_HD uint64_t mk_shl(uint64_t m, uint32_t n) { return m << n; }
KJ_HD Mask128 mk_shl(const Mask128 &m, uint32_t n) {
  if (n == 0u) return m;
  if (n >= 128u) return Mask128{0, 0};
  if (n >= 64u) return Mask128{0, m.lo << (n - 64u)};
  return Mask128{m.lo << n, (m.hi << n) | (m.lo >> (64u - n))};
}
KJ_HD uint64_t mk_and(uint64_t a, uint64_t b) { return a & b; }
KJ_HD Mask128 mk_and(const Mask128 &a, const Mask128 &b) { return Mask128{a.lo & b.lo, a.hi & b.hi}; }
KJ_HD uint32_t mk_ctz(uint64_t m) { return (uint32_t)__builtin_ctzll(m); }
KJ_HD uint32_t mk_ctz(const Mask128 &m) { return m.lo ? (uint32_t)__builtin_ctzll(m.lo) : 64u + (uint32_t)__builtin_ctzll(m.hi); }
// the lowest run of ones of E cleared; its length in `run`
KJ_HD uint64_t mk_clear_lowest_run(uint64_t E, uint32_t &run) {
  const uint64_t low = E & (~E + 1ull), E2 = E & (E + low);
  run = popc64(E ^ E2);
  return E2;
}
KJ_HD Mask128 mk_clear_lowest_run(const Mask128 &E, uint32_t &run) {
  // low = E & -E; sum = E + low (128-bit); E2 = E & sum
  Mask128 low;
  if (E.lo) { low.lo = E.lo & (~E.lo + 1ull); low.hi = 0; } else { low.lo = 0; low.hi = E.hi & (~E.hi + 1ull); }
  Mask128 sum;
  sum.lo = E.lo + low.lo;
  sum.hi = E.hi + low.hi + (sum.lo < E.lo ? 1ull : 0ull);
  const Mask128 E2{E.lo & sum.lo, E.hi & sum.hi};
  run = popc64(E.lo ^ E2.lo) + popc64(E.hi ^ E2.hi);
  return E2;
}
// are any of the `count` bits from bit `from` on set?
KJ_HD bool mk_range_any(uint64_t m, uint32_t from, uint32_t count) {
  return (mk_shr(m, from) & (count >= 64u ? ~0ull : ((1ull << count) - 1ull))) != 0;
}
KJ_HD bool mk_range_any(const Mask128 &m, uint32_t from, uint32_t count) {
  const Mask128 t = mk_shr(m, from);
  if (count >= 128u) return mk_any(t);
  if (count >= 64u) return t.lo != 0 || (t.hi & (count == 64u ? 0ull : ((1ull << (count - 64u)) - 1ull))) != 0;
  return (t.lo & ((1ull << count) - 1ull)) != 0;
}
constexpr int kS1ListCap = 24;                       // fragments ranked in LDS (two words each); a longer list is sorted in place
constexpr int kS1CntRow = 24;                        // bytes of the letter-count row of a trigger scan (21 used)
constexpr int kS1CntStride = kS1CntRow + 4;          // per lane: 7 dwords apart (odd: lanes spread over the banks)
constexpr uint32_t kS1StopCnt = 13 * 4;              // the stop's count is kept 13 higher: its dtab entries are kS1Big
constexpr uint32_t kErrReadTooLong = 16u;            // error flag: a mate longer than kaiju_gpu_set_max_read_length() allowed

struct S1Lane {              // per-lane LDS of the fast stage 1
  uint32_t *codes;           // [2 * kS1ListCap] words of the fragment list, `code_stride` dwords apart
  uint32_t code_stride;
  uint8_t *cnt;              // TRIG: letter-count row of the trigger scan, kS1CntRow bytes
  uint8_t *tsbuf;            // TRIG: staging buffer of the trigger scan, kTsBuf bytes
};

// One mate: strings to dst (16-byte aligned, six strings of u = len / 48 + 1 units) and the masks of the residues that are
// no stops (ns), bit = processing index (the codon number along the read for both strands: string index for a forward
// frame, distance from the string end for a reverse one).
template <class MaskT>
KJ_HD void s1_mate(const Stage1Tables &t, const uint8_t *s, uint32_t len, uint8_t *dst, MaskT *ns) {
  const uint32_t u = len / 48u + 1u;
  for (int f = 0; f < 6; f++) ns[f] = mk_zero(MaskT{});
  NucReader nr;
  nr.s = s; nr.len = len;
  u128 c0 = nr.chunk(0), c1 = nr.chunk(1), c2 = nr.chunk(2), c3 = nr.chunk(3);
  uint32_t idx = (uint32_t)t.nuc3[c0.x & 255u] << 3 | t.nuc3[(c0.x >> 8) & 255u];
  for (uint32_t b = 0; b < u; b++) {                        // (all u blocks: every unit of the strings gets written)
    const uint32_t w[13] = {(uint32_t)c0.x, (uint32_t)(c0.x >> 32), (uint32_t)c0.y, (uint32_t)(c0.y >> 32),
                            (uint32_t)c1.x, (uint32_t)(c1.x >> 32), (uint32_t)c1.y, (uint32_t)(c1.y >> 32),
                            (uint32_t)c2.x, (uint32_t)(c2.x >> 32), (uint32_t)c2.y, (uint32_t)(c2.y >> 32), (uint32_t)c3.x};
    // the next block's nucleotides are requested before this block's arithmetic
    const u128 n1 = nr.chunk(3 * b + 4), n2 = nr.chunk(3 * b + 5), n3 = nr.chunk(3 * b + 6);
    uint32_t fu[3][4], ru[3][4], mf[3] = {0, 0, 0}, mr[3] = {0, 0, 0};
#pragma unroll
    for (int g = 0; g < 3; g++)
#pragma unroll
      for (int d = 0; d < 4; d++) fu[g][d] = ru[g][d] = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) {
#pragma unroll
      for (int g = 0; g < 3; g++) {
        const int q = 3 * j + g + 2;                         // byte of the block that completes this codon
        const uint32_t byte = (w[q >> 2] >> (8 * (q & 3))) & 255u;
        idx = ((idx << 3) | t.nuc3[byte]) & 511u;
        const uint32_t af = t.tf[idx], ar = t.tr[idx];
        fu[g][j >> 2] |= af << (8 * (j & 3));
        ru[g][3 - (j >> 2)] |= ar << (8 * (3 - (j & 3)));
        mf[g] |= (af != 0u ? 1u : 0u) << j;
        mr[g] |= (ar != 0u ? 1u : 0u) << j;
      }
    }
    // whole units, one aligned 16-byte store each
#pragma unroll
    for (int g = 0; g < 3; g++) {
      u128 v;
      v.x = fu[g][0] | (uint64_t)fu[g][1] << 32; v.y = fu[g][2] | (uint64_t)fu[g][3] << 32;
      *reinterpret_cast<u128 *>(dst + (size_t)g * (16 * u) + 16 * b) = v;
      v.x = ru[g][0] | (uint64_t)ru[g][1] << 32; v.y = ru[g][2] | (uint64_t)ru[g][3] << 32;
      *reinterpret_cast<u128 *>(dst + (size_t)(4 + g) * (16 * u) - 16 * (b + 1)) = v;
      mk_or16(ns[g], mf[g], b);
      mk_or16(ns[3 + g], mr[g], b);
    }
    c0 = c3; c1 = n1; c2 = n2; c3 = n3;
  }
}

// The SEG trigger scan: the entropy (2^26-scaled, SegTables::ent_g32) of the window of the twelve residues ending at every
// byte of x0[0, n), kept incrementally with the letter counts of the window in the lane's LDS row - two byte updates and two
// table lookups per residue.  The window starts out as twelve stops; the letter that leaves is x0[k - 12], for the first
// `lead` steps a stop whatever lies in front of x0.  A stop in the window adds a constant no entropy reaches
// (Stage1Tables::dtab), so stops need no special handling and no state is ever reset.  Returns, MASK, bit k = the window
// ending at x0[k] is at or below the trigger entropy H <= 2.2 (n <= 64); otherwise whether any window is.
constexpr int kTsLead = 16;                           // zero bytes in front of a staged string
constexpr int kTsBuf = kTsLead + 80 + 4;              // per lane: the zeros, up to five units, 25 dwords in all (odd: banks)
constexpr int kTsBufLong = kTsLead + 96 + 4;          // ... six units (kS1UnitsLong), 29 dwords
template <bool MASK, class MaskT = uint64_t>
KJ_HD MaskT trig_scan(const Stage1Tables &t, const uint8_t *x0, uint32_t n, uint32_t lead, const uint8_t *zero, uint8_t *row) {
  uint32_t *c32 = reinterpret_cast<uint32_t *>(row);
#pragma unroll
  for (int q = 0; q < kS1CntRow / 4; q++) c32[q] = q == 0 ? (kS1StopCnt + 12u * 4u) : 0u;
  int32_t sc = 12 * t.dtab[13];
  const uint8_t *db = reinterpret_cast<const uint8_t *>(t.dtab);
  MaskT mask = mk_zero(MaskT{});
  for (uint32_t k = 0; k < n; k++) {
    const uint32_t x = x0[k];
    const uint32_t y = *(k < lead ? zero : x0 + k - 12);
    const uint32_t cx = row[x];
    row[x] = (uint8_t)(cx + 4u);
    sc += *reinterpret_cast<const int32_t *>(db + cx);
    const uint32_t cy = row[y] - 4u;
    row[y] = (uint8_t)cy;
    sc -= *reinterpret_cast<const int32_t *>(db + cy);
    mk_setbit(mask, MASK ? k : 0u, sc <= t.locut32);
  }
  return mask;
}
// The same scan over the nu <= UNITS 16-byte units of a whole frame string at src (16-byte aligned), the residues taken from
// REGISTERS: the unit that is scanned and the one before it (the letter that leaves the window ending at byte j of a unit is
// byte j + 4 of the previous unit or byte j - 12 of this one; in front of the string: stops).  Both counts of a step are read
// before either is written - a letter that joins AND leaves changes nothing -, so a step waits for ONE trip to LDS; the scan
// through the staging buffer above waits for six (its byte pointers may alias each other as far as the compiler can tell, so
// the loads of x, count[x], y, count[y] and the two stores between them stay in program order).
template <int UNITS, class MaskT>
KJ_HD MaskT trig_scan_units(const Stage1Tables &t, const uint8_t *src, uint32_t nu, uint8_t *row) {
  uint32_t *c32 = reinterpret_cast<uint32_t *>(row);
#pragma unroll
  for (int q = 0; q < kS1CntRow / 4; q++) c32[q] = q == 0 ? (kS1StopCnt + 12u * 4u) : 0u;
  int32_t sc = 12 * t.dtab[13];
  const uint8_t *db = reinterpret_cast<const uint8_t *>(t.dtab);
  MaskT mask = mk_zero(MaskT{});
  uint32_t pw[4] = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int k = 0; k < UNITS; k++) {
    if ((uint32_t)k < nu) {
      const u128 v = *reinterpret_cast<const u128 *>(src + 16 * k);
      const uint32_t w[4] = {(uint32_t)v.x, (uint32_t)(v.x >> 32), (uint32_t)v.y, (uint32_t)(v.y >> 32)};
#pragma unroll
      for (int j = 0; j < 16; j++) {
        const uint32_t x = (w[j >> 2] >> (8 * (j & 3))) & 255u;
        const uint32_t y = j < 12 ? (pw[(j + 4) >> 2] >> (8 * ((j + 4) & 3))) & 255u : (w[(j - 12) >> 2] >> (8 * ((j - 12) & 3))) & 255u;
        const uint32_t rx = row[x], ry = row[y];
        const bool same = x == y;
        row[x] = (uint8_t)(same ? rx : rx + 4u);
        row[y] = (uint8_t)(same ? rx : ry - 4u);
        const int32_t dx = *reinterpret_cast<const int32_t *>(db + rx), dy = *reinterpret_cast<const int32_t *>(db + ry - 4u);
        sc += same ? 0 : dx - dy;
        mk_setbit(mask, (uint32_t)(16 * k + j), sc <= t.locut32);
      }
#pragma unroll
      for (int q = 0; q < 4; q++) pw[q] = w[q];
    }
  }
  return mask;
}
// nu 16-byte units from device memory (16-byte aligned) behind the zeros of the lane's staging buffer
KJ_HD void trig_stage(const uint8_t *src, uint32_t nu, uint8_t *buf) {
  uint32_t *b32 = reinterpret_cast<uint32_t *>(buf);
#pragma unroll
  for (int q = 0; q < kTsLead / 4; q++) b32[q] = 0;
  for (uint32_t k = 0; k < nu; k++) {
    const u128 v = *reinterpret_cast<const u128 *>(src + 16 * k);
    b32[kTsLead / 4 + 4 * k] = (uint32_t)v.x; b32[kTsLead / 4 + 4 * k + 1] = (uint32_t)(v.x >> 32);
    b32[kTsLead / 4 + 4 * k + 2] = (uint32_t)v.y; b32[kTsLead / 4 + 4 * k + 3] = (uint32_t)(v.y >> 32);
  }
}
// does SEG report a region for the fragment pep[start, start + len)?  (exactly when some 12-window reaches the trigger
// entropy, s_SegSeq blast_seg.c:2061; same answer as seg_triggers); fragments longer than 64 residues: seg_triggers
KJ_HD bool trig_fragment(const Stage1Tables &t, const uint8_t *pep, uint32_t start, uint32_t len, uint8_t *buf, uint8_t *row) {
  if (len < (uint32_t)kSegWindow) return false;
  // (a fragment of more than 64 residues - reads beyond 191 nt - in pieces of 64 that share eleven residues: every 12-window
  //  lies inside one of them, and a piece with its alignment slack fits the five units of the staging buffer)
  for (uint32_t at = 0;; at += 64u - ((uint32_t)kSegWindow - 1u)) {
    const uint32_t n = len - at < 64u ? len - at : 64u;
    const uint32_t a = (start + at) & 15u;
    trig_stage(pep + (start + at - a), (a + n + 15u) >> 4, buf);
    if (trig_scan<false>(t, buf + kTsLead + a, n, 12, buf, row) != 0) return true;
    if (at + n >= len) return false;
  }
}
// The same without the staging buffer, for a fragment of any length: the 16-byte units that hold the fragment are loaded from
// device memory one after the other, the bytes outside the fragment made stops (a window with a stop in it never triggers,
// and the scan starts out as twelve stops: exactly the windows inside the fragment count), and the sixteen steps of a unit
// take their residues from registers - the unit itself and the one before it, as trig_scan_units does: ONE trip to LDS per
// residue (the two counts) where the staged scan makes six one after the other.  This is what the lazy SEG check of MEM runs per
// read with a hit (k_mem_post1: 1.8 of its 3.4 ms per 10 M reads were this test, profiles/r06_l29).
KJ_HD bool trig_fragment_units(const Stage1Tables &t, const uint8_t *pep, uint32_t start, uint32_t len, uint8_t *row) {
  if (len < (uint32_t)kSegWindow) return false;
  const uint32_t a = start & 15u, end = a + len, nu = (end + 15u) >> 4;
  const uint8_t *src = pep + (start - a);
  uint32_t *c32 = reinterpret_cast<uint32_t *>(row);
#pragma unroll
  for (int q = 0; q < kS1CntRow / 4; q++) c32[q] = q == 0 ? (kS1StopCnt + 12u * 4u) : 0u;
  int32_t sc = 12 * t.dtab[13];
  const uint8_t *db = reinterpret_cast<const uint8_t *>(t.dtab);
  uint32_t pw[4] = {0u, 0u, 0u, 0u};
  bool any = false;
  u128 nxt = *reinterpret_cast<const u128 *>(src);
  for (uint32_t k = 0; k < nu && !any; k++) {
    const u128 v = nxt;
    if (k + 1u < nu) nxt = *reinterpret_cast<const u128 *>(src + 16 * (k + 1u));   // (on its way while this unit is scanned)
    uint32_t w[4] = {(uint32_t)v.x, (uint32_t)(v.x >> 32), (uint32_t)v.y, (uint32_t)(v.y >> 32)};
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const uint32_t P = 16u * k + 4u * (uint32_t)q;           // position of the dword's first byte
      const uint32_t lo = a <= P ? 0xffffffffu : (a - P >= 4u ? 0u : 0xffffffffu << (8u * (a - P)));
      const uint32_t hi = end >= P + 4u ? 0xffffffffu : (end <= P ? 0u : 0xffffffffu >> (8u * (P + 4u - end)));
      w[q] &= lo & hi;
    }
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const uint32_t x = (w[j >> 2] >> (8 * (j & 3))) & 255u;
      const uint32_t y = j < 12 ? (pw[(j + 4) >> 2] >> (8 * ((j + 4) & 3))) & 255u : (w[(j - 12) >> 2] >> (8 * ((j - 12) & 3))) & 255u;
      const uint32_t rx = row[x], ry = row[y];
      const bool same = x == y;
      row[x] = (uint8_t)(same ? rx : rx + 4u);
      row[y] = (uint8_t)(same ? rx : ry - 4u);
      const int32_t dx = *reinterpret_cast<const int32_t *>(db + rx), dy = *reinterpret_cast<const int32_t *>(db + ry - 4u);
      sc += same ? 0 : dx - dy;
      any = any || sc <= t.locut32;
    }
#pragma unroll
    for (int q = 0; q < 4; q++) pw[q] = w[q];
  }
  return any;
}
// Lazy SEG, the check behind the unsplit search (k_trigcheck, k_mem_post1): v = the record's `reserved` word
// (!= 0: the fragment(s) that hold the read's longest match(es)).  Does the read have to take the SEG pass?  (buf: the lane's
// staging buffer for the staged form of the trigger test - host emulation of it -, nullptr: trig_fragment_units)
KJ_HD bool lazy_seg_needed(const Stage1Tables &t, const Params &p, const Batch &b, uint32_t r, const Hit *h, uint32_t v, uint8_t *buf, uint8_t *row) {
  if (v == kWinForce) return true;
  const ReadMeta rm = b.meta[r];
  const Frag *F = b.frags + rm.frag;
  const uint8_t *pep = b.pep + rm.pep;
  if (!(v & kWinMulti)) {
    const Frag f = F[(v & ~kWinMulti) - 1u];
    return buf ? trig_fragment(t, pep, f.start, f.len, buf, row) : trig_fragment_units(t, pep, f.start, f.len, row);
  }
  // several fragments hold a longest match: every fragment long enough to be one of them is looked at
  const uint32_t best = h->best, nf = rm.nfrag & ~kNfragSegPending;
  for (uint32_t k = 0; k < nf; k++) {
    const Frag f = F[k];
    if (f.len >= best && (buf ? trig_fragment(t, pep, f.start, f.len, buf, row) : trig_fragment_units(t, pep, f.start, f.len, row))) return true;
  }
  return false;
}
KJ_HD uint64_t bitrev64(uint64_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __brevll(v);
#else
  uint64_t r = 0;
  for (int i = 0; i < 64; i++) r |= ((v >> i) & 1ull) << (63 - i);
  return r;
#endif
}

// the low `nbits` bits of m in reverse order (nbits = 16 u: a whole frame string)
KJ_HD uint64_t mk_rev_low(uint64_t m, uint32_t nbits) { return bitrev64(m) >> (64u - nbits); }
KJ_HD Mask128 mk_rev_low(const Mask128 &m, uint32_t nbits) { return mk_shr(Mask128{bitrev64(m.hi), bitrev64(m.lo)}, 128u - nbits); }

// where fragment [a, a + l) (processing indices) of string f of a mate lies in the mate's area
KJ_HD uint32_t s1_start(int f, uint32_t u, uint32_t a, uint32_t l) {
  return f < 3 ? (uint32_t)f * (16 * u) + a : (uint32_t)(f + 1) * (16 * u) - a - l;
}

// Fragment list of read r from the masks of its mates.  `emit` receives (start in the read's area, length, emission
// moment, trigger flag) of every run of at least m residues, in no particular order.
template <class MaskT, class Emit>
KJ_HD void s1_runs(const MaskT *ns, const MaskT *tg, uint32_t len, uint32_t m, uint32_t area, uint32_t seq_base,
                   Emit &&emit) {
  if (m > 64u || m == 0u) return;
  const uint32_t top = len - 3, u = len / 48u + 1u;
  // emission moments (ConsumerThread.cpp:196-268): a forward run is emitted when the scan meets the stop behind it
  // (position = moment), the runs that reach the end of their string after the scan (len + frame); then the reverse
  // strand, scanned downwards: stop at position p -> len + 3 + (top - p), leftovers 2 * len + 3 + frame
  const uint32_t seqF = seq_base, seqR = seq_base + len + 3;
  for (int f = 0; f < 6; f++) {
    const int g = f < 3 ? f : f - 3;
    // E: positions where a run of m residues starts
    MaskT E = ns[f];
    uint32_t w = 1;
    while (2 * w <= m) { E = mk_and(E, mk_shr(E, w)); w *= 2; }
    if (w < m) E = mk_and(E, mk_shr(E, m - w));
    while (mk_any(E)) {
      const uint32_t a = mk_ctz(E);
      uint32_t run = 0;
      E = mk_clear_lowest_run(E, run);                       // the lowest run of E cleared
      const uint32_t l = run + m - 1;
      uint32_t seq;
      if (f < 3) {
        const uint32_t p = 3 * (a + l) + (uint32_t)g;        // position of the stop behind the run
        seq = p <= top ? seqF + p : seqF + len + (uint32_t)g;
      } else {
        seq = a == 0 ? seqR + len + (uint32_t)g : seqR + (top - (3 * (a - 1) + (uint32_t)g));
      }
      const bool trig = l >= 12u && mk_range_any(tg[f], a + 11u, l - 11u);
      emit(area + s1_start(f, u, a, l), l, seq, trig);
    }
  }
}

// stage 1 for read r, fast path.  TRIG: fragments whose 12-windows reach the SEG trigger entropy are queued for the SEG
// pass (as build_fragments does); otherwise the fragment flags are 0 and SEG is looked at lazily (kParamLazySeg) or not
// at all (-X).
template <bool TRIG, int UNITS = kS1Units>
KJ_HD void build_fragments_fast(const Stage1Tables &t, const Params &p, const Batch &b, const SegQueue &sq, uint32_t r,
                                uint32_t *err_flags, const S1Lane &ln) {
  typedef typename std::conditional<(UNITS <= 4), uint64_t, Mask128>::type MaskT;
  constexpr uint32_t kMaxLen = 48u * (uint32_t)UNITS - 1u;
  static_assert(UNITS == kS1Units || UNITS == kS1UnitsLong, "frame strings of four or six units");
  const uint64_t o0 = b.off[2 * (uint64_t)r], o1 = b.off[2 * (uint64_t)r + 1], o2 = b.off[2 * (uint64_t)r + 2];
  uint32_t len1 = (uint32_t)(o1 - o0), len2 = (uint32_t)(o2 - o1);
  const uint32_t m3 = p.m * 3;
  const uint32_t fbase = (uint32_t)frag_base(b.off, r, p.m);
  Frag *list = b.frags + fbase;
  const uint32_t cap = frag_cap(b.off, r, p.m);
  const uint64_t pbase = pep_base(b.off, r);
  uint32_t n = 0, pending = 0;
  if (len1 > kMaxLen || len2 > kMaxLen) {                    // the caller promised shorter reads (kaiju_gpu_set_max_read_length)
    if (err_flags) *err_flags |= kErrReadTooLong;
    len1 = len2 = 0;
  }
  // length gate, ConsumerThread.cpp:647-654
  const bool skip = b.paired ? (len1 < m3 && len2 < m3) : (len1 < m3);
  if (!skip) {
    uint8_t *area = b.pep + pbase;
    const uint32_t mate_bytes = len1 >= m3 ? 96u * (len1 / 48u + 1u) : 0u;      // where the second mate's strings start
    // the list as it is found: two words per fragment in LDS - hi sorts by key descending, then emission moment ascending -
    // and, should a read have more than kS1ListCap fragments, the rest in their list slots in device memory
    auto emit = [&](uint32_t start, uint32_t l, uint32_t seq, bool trig) {
      uint32_t key = l;
      if (p.mode == 1) {
        // (the fragment's letters, four per load: a byte load per residue of every fragment made the Greedy instantiation read
        //  its own strings back with 250 requests per read - round 6)
        key = 0;
        uint32_t x = 0;
        for (; x + 4 <= l; x += 4) {
          const uint32_t w = *reinterpret_cast<const u32_unaligned *>(area + start + x);
          key += (uint32_t)t.diag[w & 255u] + t.diag[(w >> 8) & 255u] + t.diag[(w >> 16) & 255u] + t.diag[w >> 24];
        }
        for (; x < l; x++) key += t.diag[area[start + x]];
        if (key < p.min_score) return;
      }
      if (n >= cap) return;                                   // cannot happen: cap is a proven bound
      if (n < (uint32_t)kS1ListCap) {
        ln.codes[(size_t)(2 * n) * ln.code_stride] = key << 11 | (2047u - seq);
        ln.codes[(size_t)(2 * n + 1) * ln.code_stride] = start << 8 | l << 1 | (trig ? 1u : 0u);
      } else {
        Frag f; f.start = start; f.len = l; f.key = key; f.flags = seq << 1 | (trig ? 1u : 0u);
        list[n] = f;
      }
      n++;
    };
    MaskT ns[6], tg[6];
    for (int f = 0; f < 6; f++) tg[f] = mk_zero(MaskT{});
    // TRIG: the trigger windows of a mate's six strings (stops and all), in memory order; for a reverse string that is
    // the other way round: the window ending at byte o of 16u is the one that starts at processing index 16u - 1 - o
    auto scan_mate = [&](uint8_t *dst, uint32_t len) {
      const uint32_t u = len / 48u + 1u;
      for (int f = 0; f < 6; f++) {
        const MaskT mm = trig_scan_units<UNITS, MaskT>(t, dst + (size_t)f * 16 * u, u, ln.cnt);
        const MaskT tv = f < 3 ? mm : mk_shl(mk_rev_low(mm, 16 * u), 11);
#pragma unroll
        for (int q = 0; q < 6; q++) if (q == f) tg[q] = tv;  // (the loop over f stays a loop - 64 to 96 unrolled steps each -: no dynamic index into registers)
      }
    };
    if (len1 >= m3) {
      s1_mate(t, b.seqs + o0, len1, area, ns);
#ifndef KJ_S1_NOSCAN                                         // (timing experiments only: wrong results)
      if (TRIG && p.seg) scan_mate(area, len1);
#endif
      s1_runs(ns, tg, len1, p.m, 0, 0, emit);
    }
    if (b.paired && len2 >= m3) {
      s1_mate(t, b.seqs + o1, len2, area + mate_bytes, ns);
      if (TRIG && p.seg) scan_mate(area + mate_bytes, len2);
      s1_runs(ns, tg, len2, p.m, mate_bytes, 2 * len1 + 6, emit);
    }
    // (emission moments stay below 2 * (len1 + len2) + 12 <= 780 - 1160 with six units -, keys below 11 * 96, starts below
    //  1152, lengths <= 96: the list words hold 11 bits of moment, 7 bits of length)
    auto final_flags = [&](uint32_t trig, uint32_t k) -> uint32_t {
      if (!TRIG || !p.seg) return 0u;
      if (!trig) return kFragChecked;                        // SEG would report nothing for this fragment
      const uint32_t slot = append_slot(sq.count);
      if (slot >= sq.cap) { if (err_flags) *err_flags |= 2u; return kFragChecked; }
      SegWork wk; wk.read = r; wk.frag = k;
      sq.items[slot] = wk;
      pending = kNfragSegPending;
      return (slot + 1) << kFragSlotShift;
    };
    // queue order: std::multimap<unsigned, Fragment*, std::greater>::emplace puts a fragment behind every key >= its
    // own: by descending key, equal keys in the order of emission
    if (n <= (uint32_t)kS1ListCap) {
      for (uint32_t k = 0; k < n; k++) {                     // rank by counting
        const uint32_t hi = ln.codes[(size_t)(2 * k) * ln.code_stride], lo = ln.codes[(size_t)(2 * k + 1) * ln.code_stride];
        uint32_t rank = 0;
        for (uint32_t q = 0; q < n; q++) rank += ln.codes[(size_t)(2 * q) * ln.code_stride] > hi ? 1u : 0u;
        Frag f; f.start = lo >> 8; f.len = (lo >> 1) & 127u; f.key = hi >> 11; f.flags = final_flags(lo & 1u, rank);
        list[rank] = f;
      }
    } else {
      for (uint32_t k = 0; k < (uint32_t)kS1ListCap; k++) {
        const uint32_t hi = ln.codes[(size_t)(2 * k) * ln.code_stride], lo = ln.codes[(size_t)(2 * k + 1) * ln.code_stride];
        Frag f; f.start = lo >> 8; f.len = (lo >> 1) & 127u; f.key = hi >> 11; f.flags = (2047u - (hi & 2047u)) << 1 | (lo & 1u);
        list[k] = f;
      }
      for (uint32_t k = 1; k < n; k++) {                    // insertion sort in place
        const Frag f = list[k];
        uint32_t pos = k;
        while (pos > 0 && (list[pos - 1].key < f.key || (list[pos - 1].key == f.key && list[pos - 1].flags > f.flags))) {
          list[pos] = list[pos - 1]; pos--;
        }
        list[pos] = f;
      }
      for (uint32_t k = 0; k < n; k++) list[k].flags = final_flags(list[k].flags & 1u, k);
    }
  }
  ReadMeta rm; rm.pep = pbase; rm.frag = fbase; rm.nfrag = n | pending;
  b.meta[r] = rm;
}

// ----------------------------------------------------------------------------
// stage 1 for protein input (kaiju -p: ConsumerThread.cpp:640-646,659-696; kaijup: ConsumerThreadp.cpp:17-63):
// the read is upper-cased and split at every character that is not one of the 20 amino acids; runs of at
// least m residues (Greedy: scoring at least min_score) become the fragments, in left-to-right order behind
// equal keys.  No translation, one strand: the "peptide area" of the read holds the read itself as
// index-alphabet codes (0 at separators).  The mate of a pair does not exist here (kaiju.cpp:201).
// ----------------------------------------------------------------------------
// ASCII -> index-alphabet code of the amino acid (either case), 0 for every other byte
KJ_HD void protein_code_entry(const ConstTables &t, uint32_t a, uint8_t *tbl) {
  const char letters[21] = "ARNDCQEGHILKMFPSTWYV";        // aa2int order, ConsumerThread.cpp:40-60
  const uint8_t ch = (uint8_t)letters[a];
  tbl[ch] = tbl[ch | 32u] = t.aa_to_idx[a];
}
KJ_HD void build_fragments_protein(const ConstTables &t, const uint8_t *aa_code, const Params &p, const TrigCtx &tc,
                                   const Batch &b, const SegQueue &sq, uint32_t r, uint32_t *err_flags) {
  const uint64_t o0 = b.off[2 * (uint64_t)r], o1 = b.off[2 * (uint64_t)r + 1];
  const uint32_t len = (uint32_t)(o1 - o0);
  Frag *list = b.frags + frag_base(b.off, r, p.m);
  const uint32_t cap = frag_cap(b.off, r, p.m);
  const uint64_t pbase = pep_base(b.off, r);
  uint8_t *pep = b.pep + pbase;
  uint32_t n = 0, pending = 0;
  if (len >= p.m) {                                         // length gate, ConsumerThread.cpp:640-646
    NucReader rd;
    rd.open(b.seqs + o0, len);
    uint32_t run_start = 0, run_len = 0, sum = 0, seq = 0;
    bool trig = false;
    TrigWin w;
    trig_reset(w);
    uint64_t wacc = 0;                                      // eight codes per store (the area is 16-byte aligned)
    for (uint32_t x = 0; x < len; x++) {
      const uint32_t a = aa_code[rd.at(x)];
      wacc |= (uint64_t)a << (8u * (x & 7u));
      if ((x & 7u) == 7u) { *reinterpret_cast<uint64_t *>(pep + (x & ~7u)) = wacc; wacc = 0; }
      if (a) {
        if (run_len == 0) run_start = x;
        run_len++;
        sum += (uint32_t)t.diag_idx[a];
        if (trig_push(w, tc, a, run_len)) trig = true;
      } else {
        emit_run(p, list, n, cap, run_start, run_len, sum, seq++, trig);
        run_len = 0; sum = 0; trig = false;
        trig_reset(w);
      }
    }
    *reinterpret_cast<uint64_t *>(pep + (len & ~7u)) = wacc;           // the rest and the closing 0
    emit_run(p, list, n, cap, run_start, run_len, sum, seq, trig);     // the remaining sequence, :683-694
    // queue order (std::multimap<unsigned, Fragment*, std::greater>): descending key, equal keys as emitted
    for (uint32_t k = 1; k < n; k++) {
      const Frag f = list[k];
      uint32_t pos = k;
      while (pos > 0 && (list[pos - 1].key < f.key || (list[pos - 1].key == f.key && list[pos - 1].flags > f.flags))) {
        list[pos] = list[pos - 1]; pos--;
      }
      list[pos] = f;
    }
    for (uint32_t k = 0; k < n; k++) {
      // (flags so far: emission number << 1 | some 12-window reaches the SEG trigger entropy; see build_fragments)
      uint32_t fl = 0;
      if (p.seg) {
        fl = kFragChecked;
        if (list[k].flags & 1u) {
          const uint32_t slot = append_slot(sq.count);
          if (slot >= sq.cap) { if (err_flags) *err_flags |= 2u; }
          else {
            SegWork wk; wk.read = r; wk.frag = k;
            sq.items[slot] = wk;
            pending = kNfragSegPending;
            fl = (slot + 1) << kFragSlotShift;
          }
        }
      }
      list[k].flags = fl;
    }
  }
  ReadMeta rm; rm.pep = pbase; rm.frag = (uint32_t)frag_base(b.off, r, p.m); rm.nfrag = n | pending;
  b.meta[r] = rm;
}

// MEM only: apply the SEG results eagerly (equivalent to the lazy split): split
// parents are dropped and their pieces re-inserted behind all equal keys, in parent order,
// left to right
KJ_HD void seg_apply_mem(const ConstTables &t, const Params &p, const Batch &b, const SegQueue &sq, uint32_t r,
                         uint32_t *err_flags) {
  const ReadMeta rm = b.meta[r];
  const uint32_t raw = rm.nfrag;
  if (!(raw & kNfragSegPending)) return;
  const uint32_t n_orig = raw & ~kNfragSegPending;
  Frag *list = b.frags + rm.frag;
  const uint32_t cap = frag_cap(b.off, r, p.m);
  const uint8_t *pep = b.pep + rm.pep;
  uint32_t np = 0;
  for (uint32_t k = 0; k < n_orig; k++) {
    const Frag f = list[k];
    const uint32_t slot1 = f.flags >> kFragSlotShift;
    if (!slot1) continue;
    const SegRec rec = sq.recs[slot1 - 1];
    if (rec.overflow && err_flags) *err_flags |= 1u;
    uint32_t cnt = n_orig + np;
    seg_split(t, p, rec, pep, f, FragAppend{list, &cnt, cap});
    np = cnt - n_orig;
    list[k].flags |= kFragRemoved;
  }
  uint32_t w = 0;
  for (uint32_t k = 0; k < n_orig; k++) if (!(list[k].flags & kFragRemoved)) list[w++] = list[k];
  // at least one parent was dropped, so w <= n_orig - 1 and the insertion below never
  // overwrites a piece that has not been read yet
  for (uint32_t q = 0; q < np; q++) { const Frag pc = list[n_orig + q]; frag_insert(list, w, cap, pc); }
  b.meta[r].nfrag = w;
}

// the same in the exact pass: the regions come from the pool (a lost list leaves the fragment unsplit; reported)
KJ_HD void seg_apply_mem_big(const ConstTables &t, const Params &p, const Batch &b, const BigSeg &big, uint32_t r) {
  const ReadMeta rm = b.meta[r];
  const uint32_t raw = rm.nfrag;
  if (!(raw & kNfragSegPending)) return;
  const uint32_t n_orig = raw & ~kNfragSegPending;
  Frag *list = b.frags + rm.frag;
  const uint32_t cap = frag_cap(b.off, r, p.m);
  const uint8_t *pep = b.pep + rm.pep;
  uint32_t np = 0;
  for (uint32_t k = 0; k < n_orig; k++) {
    const Frag f = list[k];
    const uint32_t slot1 = f.flags >> kFragSlotShift;
    if (!slot1) continue;
    const uint2 ent = big.index[slot1 - 1];
    if (ent.y == kBigSegLost) continue;
    uint32_t cnt = n_orig + np;
    seg_split_regs(t, p, BigRegs{big.lr + 2 * (size_t)ent.x, (int)ent.y}, pep, f, FragAppend{list, &cnt, cap});
    np = cnt - n_orig;
    list[k].flags |= kFragRemoved;
  }
  uint32_t w = 0;
  for (uint32_t k = 0; k < n_orig; k++) if (!(list[k].flags & kFragRemoved)) list[w++] = list[k];
  // at least one parent was dropped, so w <= n_orig - 1 and the insertion below never
  // overwrites a piece that has not been read yet
  for (uint32_t q = 0; q < np; q++) { const Frag pc = list[n_orig + q]; frag_insert(list, w, cap, pc); }
  b.meta[r].nfrag = w;
}

// ----------------------------------------------------------------------------
// per-lane peptide window (LDS on the device): 64 residues of the current fragment
// ----------------------------------------------------------------------------
struct LaneWin {
  uint8_t *w;                // kWin bytes, 4-byte aligned
  int32_t q;                 // fragment position of w[0]
};
// always copies kWin bytes starting at fragment position q (bytes behind the fragment end are
// never looked at; the peptide buffer is padded so that the read stays inside it)
KJ_HD void win_fill(LaneWin &lw, const uint8_t *fs, int flen, int top) {
  (void)flen;
  int q = top - (kWin - 1);
  if (q < 0) q = 0;
  lw.q = q;
  const u32_unaligned *src = reinterpret_cast<const u32_unaligned *>(fs + q);
  uint32_t *dst = reinterpret_cast<uint32_t *>(lw.w);
#pragma unroll
  for (int t = 0; t < kWin / 4; t++) dst[t] = src[t];
}
KJ_HD uint32_t win_get(LaneWin &lw, const uint8_t *fs, int flen, int pos) {
  if (pos < lw.q || pos >= lw.q + kWin) win_fill(lw, fs, flen, pos);
  return lw.w[pos - lw.q];
}

// -DKJ_PROF (developer build, KAIJU_GPU_LIB=libkaiju_gpu_prof.so tests/tools/prof_run.py): where a wavefront of greedy_lane2 / mem_lane2 spends its cycles.  KJ_P(section)
// charges the cycles since the previous mark to the PREVIOUS section and notes with how many active lanes the new one is
// entered; sums per wavefront in LDS, added to the counter block (byte 1024 on) when the wavefront ends.
enum ProfSec : int { PS_HEAD, PS_AFTER_SEARCH, PS_VAR_NEXT, PS_VAR_MATCH, PS_EVAL_NEXT, PS_EVAL_MATCH, PS_POP, PS_POP_SEG, PS_FINISH,
                     PS_HANDOUT, PS_LOAD, PS_LOAD10, PS_STEP, PS_KMER, PS_LF1, PS_SA, PS_VM_RANK, PS_VM_PUSH, PS_META, PS_FRAG,
                     PS_FILL, PS_MLOAD, PS_END_MATCH, PS_START_J, PS_LOC_ROW, PS_TAIL, PS_N };
enum ProfSecM : int { PM_HEAD, PM_LOAD, PM_LOADFILL, PM_STEP, PM_KMER, PM_LF1, PM_SA, PM_META, PM_FRAG, PM_FILL, PM_TAIL, PM_END_MATCH,
                      PM_START_J, PM_NEXT_FRAG, PM_LOC_INIT, PM_LOC_NEXT_SI, PM_LOC_ROW, PM_FINISH, PM_N };
#if defined(KJ_PROF) && defined(__HIP_DEVICE_COMPILE__)
#define KJ_P(sec) kj_prof_mark(gs.prof, (sec))
#define KJ_PM(sec) kj_prof_mark(ls.prof, (sec))
__device__ __forceinline__ void kj_prof_mark(unsigned long long *pw, int sec) {
  const unsigned long long now = __builtin_readcyclecounter();
  const unsigned long long ex = __builtin_amdgcn_ballot_w64(true);
  if ((threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(ex)) {
    const unsigned long long prev = pw[1];
    pw[2 + 3 * prev] += now - pw[0];
    pw[2 + 3 * sec + 1] += 1ull;
    pw[2 + 3 * sec + 2] += (unsigned long long)__builtin_popcountll(ex);
    pw[0] = now; pw[1] = (unsigned long long)sec;
  }
}
#else
#define KJ_P(sec)
#define KJ_PM(sec)
#endif

struct LaneScratch {
  SIEntry *si;               // this lane's match buffer
  uint32_t si_cap;
  uint8_t *win;              // this lane's peptide window (kWin bytes)
  unsigned long long *prof = nullptr;   // -DKJ_PROF: the wavefront's LDS row (2 + 3 * PM_N)
  uint8_t *coop = nullptr;   // wide lanes: the wavefront's kCoopBytesPerWave bytes of LDS (coop_fetch2; 16-byte aligned, wave-uniform)
  uint32_t *vbm = nullptr;   // VERBOSE instantiations of mem_lane2: [n][kVbAcc] words, where every recorded match lies in its read
                             // (fragment << 16 | start, in the order found) - mem_verbose_read turns them into columns 6 / 7
};

struct WorkList {
  uint32_t *counter;         // next work item
  const uint32_t *reads;     // nullptr: item i is read i; else read ids (retry pass)
  const uint32_t *n_items_ptr; // number of items lives in device memory for the retry pass
  uint32_t n_items;          // used when n_items_ptr == nullptr
  uint32_t *retry_list;      // reads whose match buffer overflowed are appended here (or nullptr)
  uint32_t *retry_count;
};
// the counting instantiations of the second-generation lanes add their totals (kOpc*) behind the batch counters:
// `counter` of the main pass is the first word of a 1024-byte block, the totals start at byte 512
constexpr int kOpcOffsetBytes = 512;
KJ_HD unsigned long long *opc_of(const WorkList &wl) {
  return reinterpret_cast<unsigned long long *>(reinterpret_cast<uint8_t *>(wl.counter) + kOpcOffsetBytes);
}

// What the counting instantiations (mem_lane2<.., COUNT>, greedy_lane2<COUNT>) add up over a batch: the memory steps
// of THIS algorithm, from which bench.py computes the algorithmic bytes of a launch.  Never part of
// a timed launch.
enum OpCount : int {
  kOpcKmer, kOpcStep, kOpcStepLines, kOpcLf, kOpcLfLines, kOpcSa, kOpcMeta, kOpcFrag, kOpcFill, kOpcTerm, kOpcSiSpill,
  kOpcHit, kOpcVmulti, kOpcPopItem, kOpcMload, kOpcPush, kOpcMatchWr, kOpcIters, kOpcLaneIters, kOpcRecBytes,
  kOpcPruned,                // Greedy: one-row variant chains that were not queued (kChainPrune)
  kOpcFillLines,             // 128-byte lines the 64-byte window / text loads touch (unaligned: one or two each; MEM lanes)
  kOpcN
};
#if defined(__HIP_DEVICE_COMPILE__)
KJ_HD void opc_flush(unsigned long long *dst, const uint32_t *oc) {
  for (int x = 0; x < kOpcN; x++) {
    uint32_t v = oc[x];
    for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o, 64);
    if ((threadIdx.x & 63u) == 0 && v) atomicAdd(dst + x, (unsigned long long)v);
  }
}
#else
KJ_HD void opc_flush(unsigned long long *dst, const uint32_t *oc) { for (int x = 0; x < kOpcN; x++) dst[x] += oc[x]; }
#endif

// ----------------------------------------------------------------------------
// verbose output (-v columns 6 and 7, ConsumerThread.cpp:527-536, :614-623, :820-824): produced by the
// first-generation lanes only.  Column 6: the sequences whose names contribute an accession, first
// kVbAcc distinct ones in the order ids_from_SI visits the rows (the host turns them into the sorted
// set of name prefixes).  Column 7: the matched peptides, written as index-alphabet codes each followed
// by 255 (MEM: the head match of every fragment that holds a longest match, :580-590; Greedy: every
// best SI with the substitutions of its variant applied, :780-790).
// ----------------------------------------------------------------------------
constexpr int kVbAcc = 20;
constexpr int kVbMem = 16;    // matches of a read that the VERBOSE instantiations of mem_lane2 describe (= the lanes' match buffer, si_cap)
struct VerboseOut {           // all pointers null: verbose output off
  uint32_t *n_acc;            // [n]
  uint32_t *acc;              // [n][kVbAcc] sequence numbers
  uint32_t *text_len;         // [n] bytes the peptides need (more than text_cap: truncated)
  uint8_t *text;              // [n][text_cap]
  uint32_t text_cap;
};
KJ_HD void vb_reset(const VerboseOut &vb, uint32_t r) { if (vb.n_acc) { vb.n_acc[r] = 0; vb.text_len[r] = 0; } }
KJ_HD void vb_acc(const VerboseOut &vb, const DevIndex &ix, uint32_t r, uint32_t iseq) {
  if (!vb.n_acc || iseq >= ix.nseq || (ix.seq_valid[iseq] & 3u) != 3u) return;
  const uint32_t n = vb.n_acc[r];
  if (n >= (uint32_t)kVbAcc) return;                       // match_dbnames.size() < max_match_acc, :822
  uint32_t *a = vb.acc + (size_t)r * kVbAcc;
  for (uint32_t q = 0; q < n; q++) if (a[q] == iseq) return;
  a[n] = iseq; vb.n_acc[r] = n + 1;
}
// peptide pep[0..len) with up to nsub substitutions (positions relative to pep[0])
KJ_HD void vb_text(const VerboseOut &vb, uint32_t r, const uint8_t *pep, uint32_t len, uint32_t nsub,
                   const uint16_t *sub_pos, const uint8_t *sub_aa, int sub_shift) {
  if (!vb.text) return;
  uint32_t w = vb.text_len[r];
  uint8_t *dst = vb.text + (size_t)r * vb.text_cap;
  for (uint32_t x = 0; x < len; x++, w++) {
    uint8_t c = pep[x];
    for (uint32_t q = 0; q < nsub; q++) if ((int)sub_pos[q] - sub_shift == (int)x) c = sub_aa[q];
    if (w < vb.text_cap) dst[w] = c;
  }
  if (w < vb.text_cap) dst[w] = 255;
  vb.text_len[r] = w + 1;
}
// the query position of a match rides in the top bits of SIEntry::lo (rows need 40 bits)
constexpr int kSiQiShift = 40;
constexpr uint64_t kSiLoMask = (1ull << kSiQiShift) - 1ull;

// ----------------------------------------------------------------------------
// the locate half shared by MEM and Greedy (ids_from_SI ConsumerThread.cpp:799-845,
// get_suffix bwt.c:105-121): one LF step per memory step
// ----------------------------------------------------------------------------
KJ_HD bool sa_lookup(const DevIndex &ix, uint64_t k, uint32_t &iseq) {
  const uint64_t idx = (k >> ix.chpt_exp) - ix.sa_skip;
  if (idx >= ix.n_sa) return false;      // the reference reads out of bounds here
  iseq = ix.sa_iseq[idx];
  return true;
}
KJ_HD void add_id(const DevIndex &ix, Hit *hit, uint32_t &nids, uint32_t iseq) {
  if (iseq >= ix.nseq || !ix.seq_valid[iseq]) return;
  const uint64_t id = ix.seq_taxid[iseq];
  for (uint32_t q = 0; q < nids; q++) if (hit->taxid[q] == id) return;
  if (nids < (uint32_t)kMaxIds) hit->taxid[nids++] = id;
}

// ----------------------------------------------------------------------------
// MEM lane: classify_length (ConsumerThread.cpp:543-628) + greedyExact (bwt.c:347-380)
// ----------------------------------------------------------------------------
// States below MS_STEP are bookkeeping (no index access) and are resolved in the inner loop;
// the three states from MS_STEP on each cost exactly one dependent index access.
enum MemState : int {
  MS_FETCH, MS_NEXT_FRAG, MS_END_MATCH, MS_LOC_INIT, MS_LOC_NEXT_SI, MS_LOC_ROW, MS_ADD_ID, MS_LOC_DONE,
  MS_STEP, MS_LF, MS_KMER, MS_EXIT
};

template <class P>
KJ_HD void mem_lane(const DevIndex &ix, const Params &p, const Batch &b, const WorkList &wl,
                    const LaneScratch &ls, const VerboseOut &vb = VerboseOut{nullptr, nullptr, nullptr, nullptr, 0}) {
  int st = MS_FETCH;
  uint32_t r = 0, nf = 0, f = 0, fcur = 0;
  const Frag *F = nullptr;
  const uint8_t *pep = nullptr, *fs = nullptr;
  int flen = 0, j = 0, i = 0;
  P lo = 0, hi = 0;
  uint32_t L = p.m, nsi = 0;
  bool found = false, ovf = false;
  // locate state
  uint32_t gs = 0, ge = 0, cur = 0, nids = 0, flags = 0, iseq = 0;
  P row = 0, rowend = 0, k = 0;
  Hit *hit = nullptr;
  LaneWin lw{ls.win, 0};
  const P check = (P)((1ull << ix.chpt_exp) - 1);
  const uint32_t n_items = wl.n_items_ptr ? *wl.n_items_ptr : wl.n_items;
  const uint32_t kk = (ix.kmer_k >= 2 && ix.kmer_k <= p.m) ? ix.kmer_k : 0;   // matches shorter than m never count
  const uint32_t xo = (p.flags & kParamXOrder) ? 1u : 0u;
  uint32_t kidx = 0;
  uint64_t klo64 = 0, khi64 = 0;

  // for (j = len-1; j >= L-1; --j), L = max(m, longest) and growing (bwt.c:356): set up the search
  // from end position j, or leave the fragment
  auto start_j = [&]() {
    if (j < (int)L - 1) { st = MS_NEXT_FRAG; return; }
    if (kk && j >= (int)kk - 1) {            // start kk letters in with one table lookup
      kidx = 0;
      for (uint32_t q = 0; q < kk; q++) kidx = kmer_index(kidx, win_get(lw, fs, flen, j - (int)q));
      // issued here, consumed (untouched until then) in the MS_KMER step: overlaps with the other lanes' loads
      if (ix.kmer32) { const uint2 e = ix.kmer32[kidx]; klo64 = e.x; khi64 = e.y; }
      else { const ulonglong2 e = ix.kmer64[kidx]; klo64 = e.x; khi64 = e.y; }
      st = MS_KMER;
      return;
    }
    const uint32_t c = win_get(lw, fs, flen, j);
    lo = (P)ix.C[c]; hi = (P)ix.C[c + 1];    // InitialSI, bwt.c:146-152
    i = j;
    st = i > 0 ? MS_STEP : MS_END_MATCH;
  };
  // next SA row of the located matches -> LF walk or sampled row (get_suffix, bwt.c:105-121)
  auto lf_check = [&]() {
    for (;;) {
      if ((k & check) != 0) { st = MS_LF; return; }
      const uint64_t idx = ((uint64_t)k >> ix.chpt_exp) - ix.sa_skip;
      if (idx < ix.n_sa) { iseq = ix.sa_iseq[idx]; st = MS_ADD_ID; return; }
      // (the reference reads out of bounds here): skip the row
      row++;
      if (row >= rowend) { st = MS_LOC_NEXT_SI; return; }
      if (nids > p.max_match_ids) { flags |= kHitIdCap; st = MS_LOC_DONE; return; }
      k = row;
    }
  };

#if defined(KJ_STATS) && defined(__HIP_DEVICE_COMPILE__)
  uint32_t stat_iters = 0, stat_step = 0, stat_kmer = 0, stat_lf = 0, stat_passes = 0;
#define KJ_STAT(x) x
#else
#define KJ_STAT(x)
#endif
  for (;;) {
    // ---- bookkeeping that needs no index access ----
    // The blocks are ordered along the usual flow (match ended -> next end position / next
    // fragment -> locate -> hit record -> next read -> first fragment) so that a lane normally gets
    // from one index access to the next in a single pass; only the rare back edges (another SA row,
    // another match) take a further pass of the loop.
    while (st < MS_STEP) {
      KJ_STAT({ const unsigned long long bal = __ballot(1); if ((threadIdx.x & 63) == (unsigned)(__ffsll((long long)bal) - 1)) stat_passes++; })
      if (st == MS_ADD_ID) {
        vb_acc(vb, ix, r, iseq);                            // (before the id: :822-824, then :834)
        add_id(ix, hit, nids, iseq);
        row++;
        st = MS_LOC_ROW;
      }
      if (st == MS_END_MATCH) {
        const uint32_t l = (uint32_t)(j - i + 1);
        if (l >= L) {
          if (l > L) { nsi = 0; ovf = false; L = l; }      // shorter matches are dropped (bwt.c:366-370, :577-582)
          if (nsi < ls.si_cap) {
            SIEntry e; e.lo = (uint64_t)lo | (uint64_t)(uint32_t)i << kSiQiShift; e.len = (uint32_t)(int32_t)(hi - lo); e.frag = fcur;
            ls.si[nsi] = e;
          } else ovf = true;
          nsi++;
          found = true;
        }
        if (i <= 1) st = MS_NEXT_FRAG;                     // bwt.c:376
        else { j--; start_j(); }
      }
      for (int again = 0; again < 2; again++) {            // second round: lanes that just fetched a read
        if (st == MS_NEXT_FRAG) {
          // getNextFragment(longest): stop when the best remaining key < longest (:550, :279)
          bool more = f < nf;
          Frag d; d.start = d.len = d.key = d.flags = 0;
          if (more) { d = F[f]; if (found && d.key < L) more = false; }
          if (!more) st = MS_LOC_INIT;
          else {
            fcur = f; f++;
            fs = pep + d.start; flen = (int)d.len;
            j = flen - 1;
            win_fill(lw, fs, flen, j);
            start_j();
          }
        }
        if (again) break;
        if (st == MS_LOC_INIT) {
          hit = b.hits + r;
          nids = 0; flags = 0;
          hit->best = found ? L : 0u;
          hit->reserved = 0;
          if (!found) st = MS_LOC_DONE;
          else if (ovf) {
            if (wl.retry_list) { wl.retry_list[append_slot(wl.retry_count)] = r; flags = kHitRetry; }
            else flags = kHitInternalOverflow;
            st = MS_LOC_DONE;
          } else { gs = ge = 0; cur = xo; st = MS_LOC_NEXT_SI; }
        }
        if (st == MS_LOC_NEXT_SI) {
          // matches of one fragment were found for descending j but are visited for ascending j
          // (greedyExact prepends, ids_from_SI_recursive walks from the head, :835-845).
          // kParamXOrder (kaijux: maxMatches(.., 1), ConsumerThreadx.cpp:135): the head is the match found first, the
          // others follow newest first (insert_SI_sorted, bwt.c:241-245): gs, ge-1, .., gs+1 (cur == ge stands for gs)
          bool any = true;
          if (cur == gs + xo) {
            gs = ge;
            if (gs >= nsi) { st = MS_LOC_DONE; any = false; }
            else {
              const uint32_t fr = ls.si[gs].frag;
              ge = gs + 1;
              while (ge < nsi && ls.si[ge].frag == fr) ge++;
              cur = ge + xo;
              // verbose: the head of the fragment's list gives the peptide (greedyExact: the match found last;
              // maxMatches: the one found first)
              if (vb.text) vb_text(vb, r, pep + F[fr].start + (uint32_t)(ls.si[xo ? gs : ge - 1].lo >> kSiQiShift), L, 0, nullptr, nullptr, 0);
            }
          }
          if (any) {
            cur--;
            const uint32_t e = (xo && cur == ge) ? gs : cur;
            row = (P)(ls.si[e].lo & kSiLoMask); rowend = row + (P)(int32_t)ls.si[e].len;
            st = MS_LOC_ROW;
          }
        }
        if (st == MS_LOC_ROW) {
          if (row >= rowend) st = MS_LOC_NEXT_SI;
          else if (nids > p.max_match_ids) { flags |= kHitIdCap; st = MS_LOC_DONE; }   // :805-807
          else { k = row; lf_check(); }
        }
        if (st == MS_LOC_DONE) {
          if (vb.text && (flags & kHitIdCap)) {
            // ids_from_SI's limit ended the traversal - but classify_length pushed the peptide of EVERY fragment that holds a
            // longest match while it searched (:580-590): the fragments behind the one at hand (round 6: a third of the reads
            // of a database of protein families end this way, and their column 7 was short, profiles/r06_l46)
            for (uint32_t g = ge; g < nsi;) {
              const uint32_t fr = ls.si[g].frag;
              uint32_t e = g + 1;
              while (e < nsi && ls.si[e].frag == fr) e++;
              vb_text(vb, r, pep + F[fr].start + (uint32_t)(ls.si[xo ? g : e - 1].lo >> kSiQiShift), L, 0, nullptr, nullptr, 0);
              g = e;
            }
          }
          hit->n_ids = nids; hit->flags = flags;
          st = MS_FETCH;
        }
        if (st == MS_FETCH) {
          const uint32_t item = fetch_work(wl.counter);
          if (item >= n_items) st = MS_EXIT;
          else {
            r = wl.reads ? wl.reads[item] : item;
            const ReadMeta rm = b.meta[r];
            nf = rm.nfrag & ~kNfragSegPending;
            F = b.frags + rm.frag;
            pep = b.pep + rm.pep;
            f = 0; L = p.m; nsi = 0; found = false; ovf = false;
            vb_reset(vb, r);
            st = MS_NEXT_FRAG;
          }
        }
      }
    }
    if (st == MS_EXIT) break;
    KJ_STAT(stat_iters++; if (st == MS_STEP) stat_step++; else if (st == MS_KMER) stat_kmer++; else stat_lf++;)
    // ---- one dependent index access ----
    if (st == MS_STEP) {
      // UpdateSI(str[i-1]) (bwt.c:160-173)
      const uint32_t c = win_get(lw, fs, flen, i - 1);
      const P nlo = rank_p<P>(ix, c, lo), nhi = rank_p<P>(ix, c, hi);
      if (nlo >= nhi) st = MS_END_MATCH;
      else { lo = nlo; hi = nhi; i--; if (i == 0) st = MS_END_MATCH; }
    } else if (st == MS_KMER) {
      // InitialSI + (kk-1) UpdateSI in one lookup
      lo = (P)klo64; hi = (P)(klo64 + khi64);
      if (lo >= hi) { i = j; st = MS_END_MATCH; }          // match shorter than kk: never recorded, i > 1
      else { i = j - (int)kk + 1; st = i > 0 ? MS_STEP : MS_END_MATCH; }
    } else {
      // one LF step of get_suffix (bwt.c:109-112): FMindexCurrent
      const uint32_t c = symbol_at(ix, k);
      if (c == 0) { iseq = (uint32_t)rank_term(ix, k); st = MS_ADD_ID; }
      else { k = rank_p<P>(ix, c, k); lf_check(); }
    }
  }
#if defined(KJ_STATS) && defined(__HIP_DEVICE_COMPILE__)
  {
    // experiment only: per-wave loop statistics into the spare counter slots
    uint32_t mx = stat_iters, mxp = stat_passes;
    for (int o = 32; o > 0; o >>= 1) { uint32_t a = __shfl_xor(mx, o, 64); if (a > mx) mx = a; mxp += __shfl_xor(mxp, o, 64); }
    unsigned long long *acc = reinterpret_cast<unsigned long long *>((reinterpret_cast<uintptr_t>(wl.counter) & ~(uintptr_t)63) + 32);
    atomicAdd(acc + 0, (unsigned long long)stat_iters);
    atomicAdd(acc + 1, (unsigned long long)stat_step);
    atomicAdd(acc + 2, (unsigned long long)stat_kmer);
    atomicAdd(acc + 3, (unsigned long long)stat_lf);
    if ((threadIdx.x & 63) == 0) { atomicAdd(acc + 4, (unsigned long long)mx); atomicAdd(acc + 5, (unsigned long long)mxp); }
  }
#endif
}

// ----------------------------------------------------------------------------
// MEM lane, second generation (indexes below 2^32 symbols, RankBlock64 layout).
//
// Same algorithm and results as mem_lane above; what changes is the shape of the loop.  Profiling
// of the first version showed that a wavefront iteration contained ~5 *serialised* memory round
// trips (the compiler waits for every load where it is first used, and uses were spread over the
// bookkeeping and the three access kinds) and that the texture-address unit, which takes ~64
// cycles per divergent wave-level load instruction, was the next limit.  Here every iteration has
//   1. ONE branch-free load phase: two rank blocks (4 loads each: 16+16+8 B of planes, 4 B count),
//      one generic 16-byte load (k-mer entry, read meta, fragment descriptor or SA sample), and -
//      only when some lane of the wave needs it - the 64-byte peptide window of a new fragment;
//      lanes that need less load a hot dummy line;
//   2. one wait (implicit, at the first use);
//   3. the per-kind compute and the bookkeeping, which touch no device memory on the hot paths.
// Every kind of step therefore costs one iteration: STEP (UpdateSI), KMER (table start), LF1/LF2
// (letter, then rank, of one LF step), SA (sampled row -> taxon id), META / FRAG / FILL (read meta,
// first fragment descriptor, peptide window + next descriptor).  Work is handed out to wavefronts
// in chunks of consecutive reads (one atomic per chunk, guided chunk size), lanes take reads
// from the wave's chunk with a ballot/prefix count.
// ----------------------------------------------------------------------------
enum MemKind : int { K_STEP, K_KMER, K_META, K_FRAG, K_FILL, K_IDLE, K_EXIT, K_WAIT,
                     K_SAPOS, K_TEXT,     // text verification: suffix-array entry of the row, then the text in front of it
                     K_PROBE,             // narrow: a k-mer lookup that decides L-k+1 end positions at once (kMemProbe)
                     K_BK = 16 };         // K_BK + b: bookkeeping block b of MemBk is due (no memory access)
enum MemBk : int { BK_NONE, BK_END_MATCH, BK_START_J, BK_NEXT_FRAG, BK_LOC_INIT, BK_FINISH };


#if defined(__HIP_DEVICE_COMPILE__)
KJ_HD uint32_t kj_nwaves() { return (gridDim.x * blockDim.x) >> 6; }
KJ_HD uint32_t kj_fetch_chunk(uint32_t *counter, uint32_t n) { return atomicAdd(counter, n); }
#else
KJ_HD uint32_t kj_nwaves() { return 1; }
KJ_HD uint32_t kj_fetch_chunk(uint32_t *counter, uint32_t n) { const uint32_t v = *counter; *counter += n; return v; }
#endif

#if defined(KJ_HIST) && !defined(__HIP_DEVICE_COMPILE__)
extern unsigned long long kj_hist[16][64];
#define KJ_HISTO(h, v) kj_hist[h][(v) < 63 ? (v) : 63]++
#ifndef KJ_HIST_QSHIFT
#define KJ_HIST_QSHIFT 0
#endif
// (MEM lane: how many letters a match still grows after its interval has shrunk to one row - what a text comparison could replace)
static int kj_single_at = 0;
#define KJ_HIST_SINGLE(is1, len) { if ((is1) && !kj_single_at) kj_single_at = (len); }
#define KJ_HIST_SINGLE_END(l) { if (kj_single_at) { KJ_HISTO(3, (int)(l) - kj_single_at); KJ_HISTO(2, kj_single_at); } kj_single_at = 0; }
#else
#define KJ_HISTO(h, v)
#define KJ_HIST_QSHIFT 0
#define KJ_HIST_SINGLE(is1, len)
#define KJ_HIST_SINGLE_END(l)
#endif
template <bool WIDE, bool XORDER = false, bool COUNT = false, bool VERBOSE = false>
KJ_HD void mem_lane2(const DevIndex &ix, const Params &p, const Batch &b, const WorkList &wl,
                     const LaneScratch &ls) {
  uint32_t oc[kOpcN];
  if constexpr (COUNT) for (int x = 0; x < kOpcN; x++) oc[x] = 0;
  // WIDE: 64-bit positions, block counts relative to mb_base, 16-byte k-mer entries (indexes >= 2^32 rows)
  typedef typename std::conditional<WIDE, uint64_t, uint32_t>::type P;
  int kind = K_IDLE;
  // read
  uint32_t r = 0, nf = 0, f = 0, fcur = 0, fbase = 0;
  uint64_t pepoff = 0;
  Frag dnext; dnext.start = dnext.len = dnext.key = dnext.flags = 0;
  // fragment / search
  uint64_t fsoff = 0;
  int flen = 0, j = 0, i = 0;
  P lo = 0, hi = 0;
  uint32_t c = 1, L = p.m, nsi = 0, kidx = 0;
  P last_sz = 1;                              // rows of the interval the last search ended in (kSpanEq; 1: the one-row rule)
  bool found = false, ovf = false, multi = false;
  int fill_top = 0;
  bool fill_newfrag = false, fill_step = false;
  // first two maximal matches live in registers, further ones in the lane's scratch
  P s0lo = 0, s1lo = 0; uint32_t s0len = 0, s1len = 0, s0frag = 0, s1frag = 0;
  // (the ids: every read leaves its longest matches in the hit record and k_mem_locate* walk them - since round 5 also reads
  //  with three and more, whose walks used to run here with one lane of the wavefront at work and, on a database of protein
  //  families or with an unlucky sample, for thousands of iterations behind the end of everything else)
  uint32_t nids = 0, flags = 0;
  P k = 0;                                    // (wide: text position of the row at hand, K_SAPOS / K_TEXT)
  Hit *hit = nullptr;
  LaneWin lw{ls.win, 0};
  const P check = (P)((1ull << ix.chpt_exp) - 1);
  const uint32_t n_items = wl.n_items_ptr ? *wl.n_items_ptr : wl.n_items;
  // the narrow lane looks k-mers up in the k-mer LINES (DevIndex::kline), the wide one in the table of 16-byte entries
  const uint32_t ktab = WIDE ? ix.kmer_k : ix.kline_k;      // (narrow: the k-mer lines; wide: the table)
  const uint32_t kk = (ktab >= 2 && ktab <= p.m && (WIDE ? (const void *)ix.kmer64 : (const void *)ix.kline)) ? ktab : 0;
  const uint32_t nwaves = kj_nwaves();
  uint32_t wnext = 0, wend = 0;                 // the wave's chunk of work items (wave-uniform)
  const RankBlock64 *const blk0 = ix.blocks64;
  bool skipj = false;                           // narrow: the k-mer that ends at the NEXT end position (j - 1) is not in the index
  // wide: a probe is a whole search from e = j - (L - pw) (k-mer lookup, then steps): if it dies with fewer than pw letters the
  // word j-L+1 .. e is not in the index and the end positions e .. j are passed (kMemProbe).  pw = the shortest word that a
  // fifth of the index's rows could hold at most (rows / 20^pw <= 0.2: eight letters at 4.3 G rows, nine at 66 G); pj = the end
  // position the probe stands in for (-1: the search at hand is no probe; -2: nor may the next one be)
  int pj = -1;
  uint32_t pw = kk;
  // narrow: a k-mer lookup as a probe pays while fewer than nine k-mers of ten exist (a probe that finds its k-mer costs an
  // iteration on top of the search it could not spare): rows <= 2 * 20^k, i.e. any narrow index at k = 7 below 2.56 G rows
  bool nprobe = kMemProbe && !WIDE;
  if constexpr (!WIDE) {
    uint64_t words = 2;
    for (uint32_t q = 0; q < kk; q++) words *= 20u;
    nprobe = nprobe && ix.bwtlen <= words;
  }
  if constexpr (WIDE) {
    uint64_t words = 1;
    for (uint32_t q = 0; q < kk; q++) words *= 20u;
    while (pw < 14u && ix.bwtlen > words / 5u) { words *= 20u; pw++; }
#ifdef KJ_PROBE_W_ADD
    pw += KJ_PROBE_W_ADD;                       // (tests on small indexes: probes of several steps)
#endif
  }

  auto si_lo = [&](uint32_t e) -> P { return e == 0 ? s0lo : e == 1 ? s1lo : (P)ls.si[e].lo; };
  auto si_len = [&](uint32_t e) -> uint32_t { return e == 0 ? s0len : e == 1 ? s1len : ls.si[e].len; };
  auto si_frag = [&](uint32_t e) -> uint32_t { return e == 0 ? s0frag : e == 1 ? s1frag : ls.si[e].frag; };
  auto in_win = [&](int pos) -> bool { return pos >= lw.q && pos < lw.q + kWin; };

  // the fragment switches (K_META / K_FRAG / K_FILL: a fifth of the kernel's cycles with two lanes of 64 active when they run
  // in every iteration, profiles/r02_gprof) only run in every second iteration; a lane that needs one waits for it (K_WAIT).
  // Measured (profiles/r03_variants): every 2nd -4.2 %, every 4th -4.0 % of the kernel
  constexpr uint32_t kMemRareGate = 1u;                 // (behind the probes: 0 -> +2.5 %, 3 -> +3.5 %, profiles/r03_l17)
  uint32_t gate_it = 0;
  for (;;) {
    KJ_PM(PM_HEAD);
    const bool rare_ok = (gate_it++ & kMemRareGate) == 0u;
    const int kind_saved = kind;
    const bool parked = !rare_ok && (kind == K_META || kind == K_FRAG || kind == K_FILL || kind == K_TEXT);
    if (parked) kind = K_WAIT;
    // ---- (0) hand out reads to the lanes that finished one (wave-uniform control flow) ----
    {
      const bool need = kind == K_IDLE;
      const uint64_t mask = kj_ballot(need);
      if (mask) {
        const uint32_t n = popc64(mask);
        const uint32_t rank = kj_rank_below(mask);
        const uint32_t avail = wend - wnext;
        uint32_t newbase = 0, ch = 0;
        if (n > avail) {
          // guided chunk size: large while much work is left, small near the end of the batch
          const uint32_t left = n_items > wend ? n_items - wend : 0;
          ch = left / (nwaves * 4u);
          if (ch > 128u) ch = 128u;
          if (ch < 8u) ch = 8u;
          if (ch < n - avail) ch = n - avail;
          const uint32_t leader = (uint32_t)__builtin_ctzll(mask);
          uint32_t got = 0;
          if (need && rank == 0) got = kj_fetch_chunk(wl.counter, ch);   // the lowest lane that needs work
          newbase = kj_bcast_uniform(got, leader);
        }
        if (need) {
          const uint32_t item = rank < avail ? wnext + rank : newbase + (rank - avail);
          if (item >= n_items) kind = K_EXIT;
          else { r = wl.reads ? wl.reads[item] : item; kind = K_META; }
        }
        if (n > avail) { wnext = newbase + (n - avail); wend = newbase + ch; }
        else wnext += n;
      }
      if (kj_ballot(kind != K_EXIT) == 0) break;
    }

    // ---- (1) load phase: no branches between the loads and their first use ----
    KJ_PM(PM_LOAD);
    KJ_HISTO(5, kind);
    if (kind == K_STEP) KJ_HISTO(4, (uint32_t)(j - i + 1));   // match length before this step
    const bool is_step = kind == K_STEP;
    const bool is_kmer = kind == K_KMER || (!WIDE && kind == K_PROBE);
    const P posA = is_step ? lo : 0;
    const P posB = is_step ? hi : posA;
    if constexpr (COUNT) {
      // one or two rank block lines per UpdateSI
      oc[kOpcLaneIters] += (kind != K_EXIT) ? 1u : 0u;
      if (kj_lane() == 0) oc[kOpcIters]++;
      if (is_kmer) oc[kOpcKmer]++;
      else if (kind == K_STEP) { oc[kOpcStep]++; oc[kOpcStepLines] += ((posA >> 6) != (posB >> 6)) ? 2u : 1u; }
      else if (kind == K_SAPOS) oc[kOpcSa]++;                                  // (a suffix-array line)
      else if (kind == K_TEXT) oc[kOpcFill]++;                                 // (64 bytes of text: priced like a window)
      else if (kind == K_META) oc[kOpcMeta]++;
      else if (kind == K_FRAG) oc[kOpcFrag]++;
      else if (kind == K_FILL) { oc[kOpcFill]++; if (fill_newfrag && f < nf) oc[kOpcFrag]++; }
    }
    const uint32_t cc = is_step ? c : 1u;
    const bool kline_step = !WIDE && is_kmer;
    u128 a01, a23, b01, b23;
    uint64_t a4, b4;
    uint32_t ca, cb;
    if constexpr (WIDE) {
      // the two rank blocks through LDS, fetched by the quad together (coop_fetch2)
      RankLines rl;
      coop_fetch2(blk0, (uint64_t)posA >> 6, (uint64_t)posB >> 6, cc, ls.coop, rl);
      a01 = rl.a01; a23 = rl.a23; a4 = rl.a4; ca = rl.ca; b01 = rl.b01; b23 = rl.b23; b4 = rl.b4; cb = rl.cb;
    } else {
      const RankBlock64 *pa = blk0 + (posA >> 6), *pb = blk0 + (posB >> 6);
      a01 = *reinterpret_cast<const u128 *>(&pa->plane[0]);
      a23 = *reinterpret_cast<const u128 *>(&pa->plane[2]);
      a4 = pa->plane[4];
      ca = pa->cnt[cc - 1];
      b01 = *reinterpret_cast<const u128 *>(&pb->plane[0]);
      b23 = *reinterpret_cast<const u128 *>(&pb->plane[2]);
      b4 = pb->plane[4];
      // (K_KMER: the second block is not needed - this load fetches the presence bits of the k-mer line instead)
      const uint32_t *cbp = &pb->cnt[cc - 1];
      if (kline_step) cbp = reinterpret_cast<const uint32_t *>(ix.kline + (size_t)(kidx >> 6) * kKLineBytes + kKLinePresent);
      cb = *cbp;
    }
    uint64_t mba = 0, mbb = 0;                             // WIDE: counts at the start of the 2^mb_shift rows
    if (WIDE) {
      mba = ix.mb_base[(size_t)((uint64_t)posA >> ix.mb_shift) * 20 + (cc - 1)];
      mbb = ix.mb_base[(size_t)((uint64_t)posB >> ix.mb_shift) * 20 + (cc - 1)];
    }
    const uint8_t *gaddr = reinterpret_cast<const uint8_t *>(blk0);
    if (is_kmer) gaddr = WIDE ? reinterpret_cast<const uint8_t *>(ix.kmer64 + kidx) : ix.kline + (size_t)kidx * 2u;
    else if (kind == K_SAPOS) gaddr = WIDE ? ix.sa_tpos5 + (size_t)((uint64_t)lo >> ix.tv_shift) * 5u : reinterpret_cast<const uint8_t *>(ix.sa_full + lo);
    else if (kind == K_META) gaddr = reinterpret_cast<const uint8_t *>(b.meta + r);
    else if (kind == K_FRAG) gaddr = reinterpret_cast<const uint8_t *>(b.frags + fbase);
    else if (kind == K_FILL && fill_newfrag && f < nf) gaddr = reinterpret_cast<const uint8_t *>(b.frags + fbase + f);
    const uint32_t ghalf = (uint32_t)(reinterpret_cast<uintptr_t>(gaddr) >> 3) & 1u;
    // 16 bytes: aligned around an 8- or 16-byte item, or (k-mer line) starting at the 2-byte aligned entry
    // (... or, wide, at the 5-byte entry of a text position)
    const bool g_asis = kline_step || (WIDE && kind == K_SAPOS);
    const u128 gv = *reinterpret_cast<const u128_unaligned *>(g_asis ? reinterpret_cast<uintptr_t>(gaddr)
                                                                     : reinterpret_cast<uintptr_t>(gaddr) & ~(uintptr_t)15);
    u128 w0{0, 0}, w1{0, 0}, w2{0, 0}, w3{0, 0};
    int fq = 0;
    if (kj_ballot(kind == K_FILL || kind == K_TEXT)) {     // wave-uniform
      KJ_PM(PM_LOADFILL);
      fq = fill_top - (kWin - 1);
      if (fq < 0) fq = 0;
      // (K_TEXT: the 64 bytes of the database that lie where fragment positions 0..63 would if the match went on)
      // (K_TEXT: the kTextCmp bytes of the database in front of the suffix, i.e. where fragment positions i-kTextCmp .. i-1 would lie)
      const uint8_t *src = kind == K_FILL ? b.pep + fsoff + fq : kind != K_TEXT ? reinterpret_cast<const uint8_t *>(blk0)
                           : WIDE ? ix.text + (size_t)((uint64_t)k - (uint64_t)kTextCmp) : ix.text + (kidx - (uint32_t)kTextCmp);
      const u128_unaligned *s16 = reinterpret_cast<const u128_unaligned *>(src);
      w0 = s16[0]; w1 = s16[1]; w2 = s16[2]; w3 = s16[3];
      if constexpr (COUNT)
        if (kind == K_FILL || kind == K_TEXT) oc[kOpcFillLines] += ((uint32_t)(reinterpret_cast<uintptr_t>(src) & 127u) + 64u > 128u) ? 2u : 1u;
    }

    // ---- (2) compute ----
    int bk = BK_NONE;
    bool noprobe = false;                                 // the search from j comes next whatever its k-mers say (a probe found its k-mer)
    if (is_step) {
      KJ_PM(PM_STEP);
      const uint64_t ia = (cc & 1u) ? 0ull : ~0ull, ib = (cc & 2u) ? 0ull : ~0ull, ic = (cc & 4u) ? 0ull : ~0ull,
                     id = (cc & 8u) ? 0ull : ~0ull, ie = (cc & 16u) ? 0ull : ~0ull;
      const uint64_t ma = (a01.x ^ ia) & (a01.y ^ ib) & (a23.x ^ ic) & (a23.y ^ id) & (a4 ^ ie);
      const P ra = (P)(mba + ca + popc64(ma & ((1ull << (posA & 63u)) - 1ull)));
      {
        // UpdateSI(str[i-1]) (bwt.c:160-173)
        const uint64_t mb = (b01.x ^ ia) & (b01.y ^ ib) & (b23.x ^ ic) & (b23.y ^ id) & (b4 ^ ie);
        const P rb = (P)(mbb + cb + popc64(mb & ((1ull << (posB & 63u)) - 1ull)));
        if (ra >= rb) bk = BK_END_MATCH;
        else {
          lo = ra; hi = rb; i--;
          KJ_HIST_SINGLE(hi - lo == 1, j - i + 1);
          if (i == 0) bk = BK_END_MATCH;
          else if (WIDE && pj >= 0 && j - i + 1 >= (int)pw) bk = BK_END_MATCH;   // a probe whose word is in the index: no need to go on
          else if ((WIDE ? ix.sa_tpos5 != nullptr && ((uint64_t)lo & ((1ull << ix.tv_shift) - 1ull)) == 0 : ix.text != nullptr) &&
                   hi - lo == 1 && j - i + 1 >= kTextTrigLen && i >= kTextMinLeft && lw.q == 0 && i <= kWin) {
            // one row left and letters to go: the rest of this match is read off the database text (K_SAPOS, K_TEXT).
            // (wide: only rows whose text position is kept - the search steps on until it stands on one)
            kind = K_SAPOS;
          }
          else if (in_win(i - 1)) c = lw.w[i - 1 - lw.q];
          else { fill_top = i - 1; fill_newfrag = false; fill_step = true; kind = K_FILL; }
        }
      }
    } else if (is_kmer) {
      KJ_PM(PM_KMER);
      // InitialSI + (kk-1) UpdateSI in one lookup
      uint32_t hint = 32u;                                 // narrow: the BWT letter of a one-row interval (32 = unknown)
      bool escape = false;
      if constexpr (WIDE) { lo = (P)gv.x; hi = (P)(gv.x + gv.y); }
      else {
        const uint64_t e = kline_entry(gv, kidx);
        const uint32_t l16 = (uint32_t)(e >> 32);
        lo = (P)(uint32_t)e;
        if (l16 >= kKLineSingle) {
          if (l16 == kKLineEscape) escape = true;
          else { hi = lo + 1; hint = l16 & 31u; }
        } else hi = lo + (P)l16;
        // the k-mer that ends at j - 1 = w[j-kk] in front of the line's word: absent -> that end position is skipped
        // without a lookup (what the lookup would have led to: an empty interval, a "match" of one letter)
        skipj = j >= (int)kk && in_win(j - (int)kk) && ((cb >> ((uint32_t)lw.w[j - (int)kk - lw.q] - 1u)) & 1u) == 0u;
      }
      if (!WIDE && kind == K_PROBE) {
        // the k-mer that ends at e = j - (L - kk) (kMemProbe)
        const int e = j - (int)(L - kk);
        if (!escape && (lo >= hi || (kSpanRule && hi - lo == (kSpanEq ? last_sz : (P)1) && e - (int)kk + 1 >= i))) {
          // ... and the one that ends at e - 1, absent, passes end position e - 1 too
          const bool prev_absent = e >= (int)kk && in_win(e - (int)kk) && ((cb >> ((uint32_t)lw.w[e - (int)kk - lw.q] - 1u)) & 1u) == 0u;
          j = e - 1 - (prev_absent ? 1 : 0);
        } else noprobe = true;
        skipj = false;
        bk = BK_START_J;
      } else if (escape) {
        // an interval longer than the line's 16 bits can say: this search starts with InitialSI (bwt.c:146-152)
        c = lw.w[j - lw.q];
        lo = (P)ix.C[c]; hi = (P)ix.C[c + 1];
        i = j;                                             // (j >= kk - 1 >= 1)
        if (in_win(i - 1)) { c = lw.w[i - 1 - lw.q]; kind = K_STEP; }
        else { fill_top = i - 1; fill_newfrag = false; fill_step = true; kind = K_FILL; }
      } else if (lo >= hi) { i = j; bk = BK_END_MATCH; }   // match shorter than kk: never recorded, i > 1
      else if (kSpanRule && hi - lo == (kSpanEq ? last_sz : (P)1) && j - (int)kk + 1 >= i) bk = BK_END_MATCH;   // inside the last match (see kSpanRule, kSpanEq): i stays
      else {
        i = j - (int)kk + 1;
        KJ_HIST_SINGLE(hi - lo == 1, (int)kk);
        if (i == 0 || (WIDE && pj >= 0 && kk >= pw)) bk = BK_END_MATCH;   // (a probe of kk letters ends with the lookup)
        else if (in_win(i - 1)) {
          c = lw.w[i - 1 - lw.q];
          // one row whose BWT letter is not c: UpdateSI(c) finds nothing (bwt.c:160-173) - the match ends here
          if (hint != 32u && hint != c) bk = BK_END_MATCH; else kind = K_STEP;
        }
        else { fill_top = i - 1; fill_newfrag = false; fill_step = true; kind = K_FILL; }
      }
    } else if (kind == K_SAPOS) {
      // position in the text of the suffix of row lo = of fragment position i
      if constexpr (WIDE) {
        k = (P)(gv.x & kTposNone);                            // (k: free until the locate)
        if ((uint64_t)k == kTposNone) { c = lw.w[i - 1 - lw.q]; kind = K_STEP; }   // no position known for this row: the search steps on
        else kind = K_TEXT;
      } else {
        const uint32_t q = (uint32_t)lo & 3u;
        kidx = q == 0 ? (uint32_t)gv.x : q == 1 ? (uint32_t)(gv.x >> 32) : q == 2 ? (uint32_t)gv.y : (uint32_t)(gv.y >> 32);
        kind = K_TEXT;
        // (KAIJU_IDX_WARN_SA_SHORT: a row behind the missing sample keeps stepping - the match is then recorded through its
        //  own end row, whose locate decides as the lanes without the text arrays do; see DevIndex::beyond_lo)
        if (ix.beyond_n && kidx - ix.beyond_lo < ix.beyond_n) { c = lw.w[i - 1 - lw.q]; kind = K_STEP; }
      }
    } else if (kind == K_TEXT) {
      // UpdateSI on a one-row interval succeeds iff the letter in front of the suffix is the next letter of the read
      // (bwt.c:160-173 with hi - lo = 1): the match ends in front of the highest position x < i whose letter differs from the
      // text's (a terminator, 0, differs from every letter), or at the start of the fragment.  kTextCmp letters per round: text
      // dword q holds what fragment positions i-kTextCmp+4q .. +3 are compared with; the window's bytes are brought into that
      // alignment with one byte-align per dword.  Whatever stands below position 0 (the window words are clamped to the
      // window) may differ or not: a difference there means "down to the start of the fragment" as well
      const uint32_t *w32 = reinterpret_cast<const uint32_t *>(lw.w);
      const uint32_t t32[16] = {(uint32_t)w0.x, (uint32_t)(w0.x >> 32), (uint32_t)w0.y, (uint32_t)(w0.y >> 32),
                                (uint32_t)w1.x, (uint32_t)(w1.x >> 32), (uint32_t)w1.y, (uint32_t)(w1.y >> 32),
                                (uint32_t)w2.x, (uint32_t)(w2.x >> 32), (uint32_t)w2.y, (uint32_t)(w2.y >> 32),
                                (uint32_t)w3.x, (uint32_t)(w3.x >> 32), (uint32_t)w3.y, (uint32_t)(w3.y >> 32)};
      const int d0 = (i >> 2) - (kTextCmp >> 2);            // window dword of position i - 32 (rounded down)
      const uint32_t sh = (uint32_t)i & 3u;
      uint32_t lo32 = w32[d0 < 0 ? 0 : d0];
      int best = -0x10000;
#pragma unroll
      for (int q = 0; q < (kTextCmp >> 2); q++) {
        const int dq = d0 + q + 1;
        const uint32_t hi32 = w32[dq < 0 ? 0 : dq > (kWin >> 2) - 1 ? (kWin >> 2) - 1 : dq];
        const uint32_t d = t32[q] ^ kj_alignbyte(hi32, lo32, sh);
        // highest differing byte of this dword: 4q + 3 - clz / 8 (no difference: far below every other candidate)
        const int cand = d ? 4 * q + 3 - (int)((uint32_t)__builtin_clz(d) >> 3) : -0x10000;
        best = cand > best ? cand : best;
        lo32 = hi32;
      }
      if (best >= 0 || i <= kTextCmp) {
        const int x = best >= 0 ? i - kTextCmp + best : -1;
        const int i_new = x < 0 ? 0 : x + 1;
        if constexpr (!WIDE) {
          // (KAIJU_IDX_WARN_SA_SHORT: the suffix of the grown match itself lies behind the missing sample - its row gives no
          //  id, as for the lanes that step there: recorded through a row that says so)
          if (ix.beyond_n && kidx - (uint32_t)(i - i_new) - ix.beyond_lo < ix.beyond_n) { lo = (P)ix.beyond_row; hi = lo + 1; }
        }
        i = i_new;
        bk = BK_END_MATCH;                                  // (lo, hi still name the one row the search had reached: same sequence)
      } else {
        // all of them agree and there are more: one more round (fragments this long are rare: i <= kWin at the first one)
        i -= kTextCmp;
        if constexpr (WIDE) k -= (P)kTextCmp; else kidx -= (uint32_t)kTextCmp;
      }
    } else if (kind >= K_BK) {
      bk = kind - K_BK;                                      // a bookkeeping block left over from the last iteration
    } else if (kind == K_META) {
      KJ_PM(PM_META);
      pepoff = gv.x;
      fbase = (uint32_t)gv.y;
      nf = (uint32_t)(gv.y >> 32) & ~kNfragSegPending;
      f = 0; L = p.m; nsi = 0; found = false; ovf = false;
      multi = false;
      hit = b.hits + r;
      if (nf == 0) bk = BK_LOC_INIT; else kind = K_FRAG;
    } else if (kind == K_FRAG) {
      KJ_PM(PM_FRAG);
      dnext.start = (uint32_t)gv.x; dnext.len = (uint32_t)(gv.x >> 32);
      dnext.key = (uint32_t)gv.y; dnext.flags = (uint32_t)(gv.y >> 32);
      bk = BK_NEXT_FRAG;
    } else if (kind == K_FILL) {
      KJ_PM(PM_FILL);
      lw.q = fq;
      uint32_t *d32 = reinterpret_cast<uint32_t *>(lw.w);   // 4-byte aligned LDS: written as dwords
      d32[0] = (uint32_t)w0.x; d32[1] = (uint32_t)(w0.x >> 32); d32[2] = (uint32_t)w0.y; d32[3] = (uint32_t)(w0.y >> 32);
      d32[4] = (uint32_t)w1.x; d32[5] = (uint32_t)(w1.x >> 32); d32[6] = (uint32_t)w1.y; d32[7] = (uint32_t)(w1.y >> 32);
      d32[8] = (uint32_t)w2.x; d32[9] = (uint32_t)(w2.x >> 32); d32[10] = (uint32_t)w2.y; d32[11] = (uint32_t)(w2.y >> 32);
      d32[12] = (uint32_t)w3.x; d32[13] = (uint32_t)(w3.x >> 32); d32[14] = (uint32_t)w3.y; d32[15] = (uint32_t)(w3.y >> 32);
      if (fill_newfrag) {
        if (f < nf) {
          dnext.start = (uint32_t)gv.x; dnext.len = (uint32_t)(gv.x >> 32);
          dnext.key = (uint32_t)gv.y; dnext.flags = (uint32_t)(gv.y >> 32);
        }
        bk = BK_START_J;
      } else if (fill_step) { c = lw.w[i - 1 - lw.q]; kind = K_STEP; }
      else bk = BK_START_J;
    }

    if (parked) kind = kind_saved;
    // ---- (3) bookkeeping, blocks ordered along the usual flow (see mem_lane) ----
    KJ_PM(PM_TAIL);
    // ONE pass over the blocks, which stand in the order of the usual flow (a `while` around them made every variable of the
    // lane a loop-carried value: the compiler copied some thirty registers at the head of that loop and again at every join -
    // half of the kernel's VALU instructions were v_mov).
    {
      if (bk == BK_END_MATCH) {
        KJ_PM(PM_END_MATCH);
        const uint32_t l = (uint32_t)(j - i + 1);
        KJ_HIST_SINGLE_END(l);
        if constexpr (kSpanEq) last_sz = hi > lo ? (P)(hi - lo) : (P)1;   // (an absent k-mer ends its search at i = j: no k-mer lies inside that)
        bool probed = false;
        if constexpr (WIDE) {
          if (pj >= 0) {
            // the search from e was a probe: fewer than pw letters = no match of L letters ends at e .. pj (each would contain the
            // word j-L+1 .. e); otherwise the search from pj comes next, unprobed (and `i`, the end of a search from a SMALLER end
            // position, is no bound for the span rule there)
            probed = true;
            if (l < pw) { j--; pj = -1; }
            else { j = pj; pj = -2; i = flen; }             // (-2: until the lookup of that search is on its way - a window refill may come first)
            bk = BK_START_J;
          }
        }
        if (probed) {}
        else
        if (l >= L) {
          if (l > L) { nsi = 0; ovf = false; L = l; multi = false; }   // shorter matches are dropped (bwt.c:366-370, :577-582)
          const uint32_t ilen = (uint32_t)(int32_t)(hi - lo);
          if (nsi != 0 && fcur != s0frag) multi = true;      // (lazy SEG: longest matches in more than one fragment)
          if constexpr (VERBOSE) { if (nsi < (uint32_t)kVbMem) ls.vbm[(size_t)r * kVbAcc + nsi] = fcur << 16 | ((uint32_t)i & 0xffffu); }
          if (nsi == 0) { s0lo = lo; s0len = ilen; s0frag = fcur; }
          else if (nsi == 1) { s1lo = lo; s1len = ilen; s1frag = fcur; }
          else if (nsi < ls.si_cap) { SIEntry e; e.lo = lo; e.len = ilen; e.frag = fcur; ls.si[nsi] = e; if constexpr (COUNT) oc[kOpcSiSpill]++; }
          else ovf = true;
          nsi++;
          found = true;
        }
        if (probed) {}
        else if (i <= 1) bk = BK_NEXT_FRAG;                // bwt.c:376
        else { j--; bk = BK_START_J; }
      }
      if (bk == BK_START_J) {
        KJ_PM(PM_START_J);
        // for (j = len-1; j >= L-1; --j), L = max(m, longest) and growing (bwt.c:356)
        if (!WIDE && skipj && j >= (int)L - 1) {
          // the k-mer that ends here is not in the index (the line of end position j + 1 said so): as for an empty table
          // entry - i = j, nothing recorded, `if (i <= 1) break` (bwt.c:376), --j - and on to j - 1 in the same pass
          if (j <= 1) bk = BK_NEXT_FRAG; else j--;
        }
        skipj = false;
        if (bk == BK_NEXT_FRAG) {}
        else if (j < (int)L - 1 && !(WIDE && pj >= 0)) bk = BK_NEXT_FRAG;   // (a probe's end position lies below L - 1)
        else if (kk && j >= (int)kk - 1) {
          if constexpr (WIDE) {
            // wide: the search from e = j - (L - pw) stands in for the end positions e .. j (see pj above)
            if (kMemProbe && pj == -1 && L > pw) { pj = j; j -= (int)(L - pw); }
          }
          // narrow: the lookup is a probe (kMemProbe) unless one has just sent the lane here; e = the end position of its k-mer
          const bool probe = nprobe && !noprobe && L > kk && L <= (uint32_t)kWin;
          const int e = probe ? j - (int)(L - kk) : j;
          if (in_win(j) && in_win(e - (int)kk + 1)) {
            // (a rolling update of the index from end position j+1 was tried: it costs a register too many here)
            if constexpr (WIDE) {
              kidx = 0;
              for (uint32_t q = 0; q < kk; q++) kidx = kmer_index(kidx, lw.w[j - (int)q - lw.q]);
            } else {
              kidx = 0;
              for (uint32_t q = 1; q < kk; q++) kidx = kline_code(kidx, lw.w[e - (int)q - lw.q]);
              kidx = kline_ref(kidx, lw.w[e - lw.q]);
            }
            kind = probe ? K_PROBE : K_KMER; bk = BK_NONE;
            if (WIDE && pj == -2) pj = -1;
          } else { fill_top = j; fill_newfrag = false; fill_step = false; kind = K_FILL; bk = BK_NONE; }
        } else if (in_win(j)) {
          c = lw.w[j - lw.q];
          lo = (P)ix.C[c]; hi = (P)ix.C[c + 1];            // InitialSI, bwt.c:146-152
          i = j;
          if (i == 0) bk = BK_END_MATCH;
          else if (in_win(i - 1)) { c = lw.w[i - 1 - lw.q]; kind = K_STEP; bk = BK_NONE; }
          else { fill_top = i - 1; fill_newfrag = false; fill_step = true; kind = K_FILL; bk = BK_NONE; }
        } else { fill_top = j; fill_newfrag = false; fill_step = false; kind = K_FILL; bk = BK_NONE; }
      }
      if (bk == BK_NEXT_FRAG) {
        KJ_PM(PM_NEXT_FRAG);
        // getNextFragment(longest): stop when the best remaining key < longest (:550, :279);
        // dnext is the prefetched descriptor of fragment f
        if (f >= nf || (found && dnext.key < L)) bk = BK_LOC_INIT;
        else {
          fcur = f; f++;
          skipj = false;
          fsoff = pepoff + dnext.start; flen = (int)dnext.len;
          j = flen - 1;
          i = flen;                                        // (span rule: no earlier search in this fragment)
          pj = -1;
          fill_top = j; fill_newfrag = true; fill_step = false;
          kind = K_FILL; bk = BK_NONE;
        }
      }
      if (bk == BK_LOC_INIT) {
        KJ_PM(PM_LOC_INIT);
        nids = 0; flags = 0;
        hit->best = found ? L : 0u;
        // lazy SEG (kParamLazySeg, wave-uniform): the fragment that holds the longest matches, or the note that there
        // are several
        hit->reserved = ((p.flags & kParamLazySeg) && found) ? (ovf ? kWinForce : ((s0frag + 1u) | (multi ? kWinMulti : 0u))) : 0u;
        if (!found) bk = BK_FINISH;
        else if (ovf) {
          // (lazy SEG: the read takes the SEG pass first; the search behind it sends it to the retry pass)
          if (p.flags & kParamLazySeg) flags = kHitRetry;
          else if (wl.retry_list) { wl.retry_list[append_slot(wl.retry_count)] = r; flags = kHitRetry; }
          else flags = kHitInternalOverflow;
          bk = BK_FINISH;
        } else if (nsi <= 2u && (!WIDE || (s0len < kLocWideMaxLen && (nsi < 2u || s1len < kLocWideMaxLen)))) {
          // the locate walks of 64 different reads share nothing: they run with two or three lanes of a wavefront active
          // (a third of this kernel's time, profiles/r02_gprof).  The matches are noted in the order in which ids_from_SI
          // visits them (matches of one fragment: found for descending j, visited for ascending j; kaijux - XORDER - keeps
          // the order in which they were found) and k_mem_locate* walk them with every lane at work.
          const bool swap = !XORDER && nsi == 2u && s0frag == s1frag;
          // (narrow: row | length << 32; wide: row | length << 40, lengths below 2^24)
          const uint64_t e0 = WIDE ? ((uint64_t)s0lo | (uint64_t)s0len << kLocWideShift) : ((uint64_t)(uint32_t)s0lo | (uint64_t)s0len << 32);
          const uint64_t e1 = WIDE ? ((uint64_t)s1lo | (uint64_t)s1len << kLocWideShift) : ((uint64_t)(uint32_t)s1lo | (uint64_t)s1len << 32);
          hit->taxid[0] = swap ? e1 : e0;
          if (nsi == 2u) hit->taxid[1] = swap ? e0 : e1;
          nids = nsi; flags = kHitLocPending;
          bk = BK_FINISH;
        } else {
          // three and more longest matches (two reads in ten thousand on random data, a few in a hundred on a database of
          // protein families): all of them go into the record as well - nsi <= si_cap = 16 of its 21 slots - in visiting order:
          // the matches of one fragment stand together (found for descending j); greedyExact prepends and
          // ids_from_SI_recursive walks from the head (:835-845), so a fragment's matches are visited for ascending j.
          // XORDER (kaijux, whose classify_length searches with maxMatches(.., 1), ConsumerThreadx.cpp:135): the list head is
          // the match found FIRST, the others follow newest first (insert_SI_sorted, bwt.c:241-245): gs, ge-1, .., gs+1.
          bool big = false;
          if constexpr (WIDE) for (uint32_t e = 0; e < nsi; e++) big = big || si_len(e) >= kLocWideMaxLen;
          if (big || nsi > (uint32_t)kMaxIds) {
            // (an interval of 2^24 rows and more does not fit the record's entry: the retry pass walks such a read itself)
            if (p.flags & kParamLazySeg) flags = kHitRetry;
            else if (wl.retry_list) { wl.retry_list[append_slot(wl.retry_count)] = r; flags = kHitRetry; }
            else flags = kHitInternalOverflow;
          } else {
            uint32_t q = 0, gs = 0;
            while (gs < nsi) {
              const uint32_t fr = si_frag(gs);
              uint32_t ge = gs + 1;
              while (ge < nsi && si_frag(ge) == fr) ge++;
              if (XORDER) hit->taxid[q++] = WIDE ? ((uint64_t)si_lo(gs) | (uint64_t)si_len(gs) << kLocWideShift) : ((uint64_t)(uint32_t)si_lo(gs) | (uint64_t)si_len(gs) << 32);
              for (uint32_t e = ge; e-- > gs + (XORDER ? 1u : 0u);)
                hit->taxid[q++] = WIDE ? ((uint64_t)si_lo(e) | (uint64_t)si_len(e) << kLocWideShift) : ((uint64_t)(uint32_t)si_lo(e) | (uint64_t)si_len(e) << 32);
              gs = ge;
            }
            nids = nsi; flags = kHitLocPending;
          }
          bk = BK_FINISH;
        }
      }
      if (bk == BK_FINISH) {
        KJ_PM(PM_FINISH);
        hit->n_ids = nids; hit->flags = flags;
        if constexpr (COUNT) oc[kOpcHit]++;
        kind = K_IDLE; bk = BK_NONE;
      }
    }
    if (bk != BK_NONE) kind = K_BK + bk;
  }
  if constexpr (COUNT) opc_flush(opc_of(wl), oc);
#if defined(KJ_PROF) && defined(__HIP_DEVICE_COMPILE__)
  KJ_PM(PM_HEAD);
  if ((threadIdx.x & 63u) == 0 && ls.prof) {
    unsigned long long *dst = reinterpret_cast<unsigned long long *>(reinterpret_cast<uint8_t *>(wl.counter) + 1024);
    for (int x = 0; x < 3 * PM_N; x++) atomicAdd(dst + x, ls.prof[2 + x]);
  }
#endif
}

constexpr uint32_t kLocMaxEntries = 24;      // matches a hit record can hold for the locate kernels (MEM: 16 = the lane's match buffer; Greedy: max_matches_SI = 20; record: 21 slots)
// The ids of a read whose longest matches mem_lane2 left in its hit record (kHitLocPending): ids_from_SI for every match in
// turn (ConsumerThread.cpp:799-845; get_suffix bwt.c:105-121, FMindexCurrent compactfmi.c:312-336) - the same steps as
// BK_LOC_ROW / K_LF1 / K_LF2 / K_SA of the lane, one read per lane, narrow index.
// MANYROWS: the instantiation for reads whose matches hold many rows (a protein family: hundreds of rows of a dozen taxa).  The
// ids collected so far then also sit in registers (fully unrolled, no dynamic indexing) - looking every row's taxon up in the
// record in device memory, five dependent loads a row, was most of the post-search time on a database that is not i.i.d.
// (bench.py's hard leg).  42 registers: the common case (one row, one id) keeps the lean instantiation and hands reads with
// more than defer_rows rows on (returns false, the record untouched; defer_rows 0 = never).
template <bool WIDE, bool MANYROWS = false>
KJ_HD bool mem_locate_read(const DevIndex &ix, const Params &p, Hit *hit, uint32_t defer_rows = 0) {
  typedef typename std::conditional<WIDE, uint64_t, uint32_t>::type P;
  const uint32_t fl0 = hit->flags;
  if (!(fl0 & kHitLocPending)) return true;
  const uint32_t nsi = hit->n_ids;
  // the matches (row | length): the lean instantiation holds two in registers and hands reads with more on (they are rare:
  // two in ten thousand on random data) - the many-rows one reads every entry when its turn comes: it only writes the record
  // when it is through.  (Host emulation without a list: all of them copied first.)
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr uint32_t kEMax = MANYROWS ? 1u : 2u;              // (no array behind a run-time index: it would live in scratch memory)
#else
  constexpr uint32_t kEMax = MANYROWS ? 1u : kLocMaxEntries;
#endif
  uint64_t e[kEMax];
  e[0] = hit->taxid[0];
  if constexpr (!MANYROWS) {
    e[1] = nsi > 1u ? hit->taxid[1] : 0ull;
    if (nsi > 2u) {
#if defined(__HIP_DEVICE_COMPILE__)
      return false;                                          // (device: k_mem_locate always has a list - k_mem_locate_list - behind it)
#else
      if (defer_rows) return false;
      for (uint32_t q = 2; q < nsi && q < kEMax; q++) e[q] = hit->taxid[q];
#endif
    }
    if (defer_rows) {
      const uint64_t rows = (WIDE ? (e[0] >> kLocWideShift) : (e[0] >> 32)) + (nsi > 1u ? (WIDE ? (e[1] >> kLocWideShift) : (e[1] >> 32)) : 0ull);
      if (rows > defer_rows) return false;
    }
  }
  const P check = (P)((1ull << ix.chpt_exp) - 1ull);
  const RankBlock64 *const blk0 = ix.blocks64;
  uint32_t nids = 0, flags = fl0 & ~kHitLocPending;          // (a Greedy read may carry kHitSiCap already)
  uint64_t id0 = 0;
  uint32_t idr[MANYROWS ? kMaxIds : 1];                      // MANYROWS: the dense taxon indices collected so far
  if constexpr (MANYROWS) {
#pragma unroll
    for (int q = 0; q < kMaxIds; q++) idr[q] = 0;
  }
  // an id of the read: a taxon id - or, on the row_tax path, its dense index (translated when the record is written)
  auto add_tax = [&](uint64_t tax) {
    bool dup = false;
    if constexpr (MANYROWS) {
#pragma unroll
      for (int q = 0; q < kMaxIds; q++) dup = dup || (q < (int)nids && idr[q] == (uint32_t)tax);
      if (!dup && nids < (uint32_t)kMaxIds) {
#pragma unroll
        for (int q = 0; q < kMaxIds; q++) if (q == (int)nids) idr[q] = (uint32_t)tax;
        nids++;
      }
    } else {
      if (nids >= 1 && tax == id0) dup = true;
      for (uint32_t q = 1; q < nids && !dup; q++) if (hit->taxid[q] == tax) dup = true;
      if (!dup && nids < (uint32_t)kMaxIds) { if (nids == 0) id0 = tax; hit->taxid[nids++] = tax; }
    }
  };
  const bool dense = ix.row_tax != nullptr;                   // (wide indexes: where HBM had room for it, capi.hip)
  bool done = false;
  for (uint32_t s = 0; s < nsi && !done; s++) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint64_t es = MANYROWS ? hit->taxid[s] : (s == 0u ? e[0] : e[kEMax - 1u]);    // (device: nsi <= 2 here)
#else
    const uint64_t es = MANYROWS ? hit->taxid[s] : e[MANYROWS ? 0 : s];
#endif
    const P lo = WIDE ? (P)(es & ((1ull << kLocWideShift) - 1ull)) : (P)(uint32_t)es;
    const uint32_t len = WIDE ? (uint32_t)(es >> kLocWideShift) : (uint32_t)(es >> 32);
    const P rowend = lo + (P)(int32_t)len;
    if constexpr (MANYROWS) {
      // ids_from_SI's loop over the rows of the match (:803-844), four rows per 16-byte load of the row -> taxon table
      for (P row4 = lo & ~(P)3; row4 < rowend && !done; row4 += 4) {
        const u128 v = *reinterpret_cast<const u128 *>(ix.row_tax + row4);
        const uint32_t t4[4] = {(uint32_t)v.x, (uint32_t)(v.x >> 32), (uint32_t)v.y, (uint32_t)(v.y >> 32)};
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const P row = row4 + (P)q;
          if (row >= lo && row < rowend && !done) {
            if (nids > p.max_match_ids) { flags |= kHitIdCap; done = true; }     // :805-807, in front of every row
            else if (t4[q] != 0xffffffffu) add_tax(t4[q]);
          }
        }
      }
      continue;
    }
    for (P row = lo; row < rowend; row++) {
      if (nids > p.max_match_ids) { flags |= kHitIdCap; done = true; break; }     // :805-807
      if (dense) {
        // the walk below, precomputed for every row at index load (k_suffix_walk; wide: k_seq_walk_fill), down to the taxon
        const uint32_t t = ix.row_tax[row];
        if (t != 0xffffffffu) add_tax(t);
        continue;
      }
      P k = row;
      for (;;) {
        if ((k & check) == 0) {
          const uint64_t sa_idx = ((uint64_t)k >> ix.chpt_exp) - ix.sa_skip;
          if (sa_idx < ix.n_sa) {
            uint64_t tax;
            if constexpr (WIDE) { const uint32_t iseq = ix.sa_iseq[sa_idx]; tax = iseq < ix.nseq ? ix.seq_taxid[iseq] : ~0ull; }
            else tax = ix.sa_taxid[sa_idx];
            if (tax != ~0ull) add_tax(tax);
          }
          break;                                           // (beyond the samples the reference reads out of bounds: the row is skipped)
        }
        const RankBlock64 &rb = blk0[k >> 6];
        const uint32_t sft = (uint32_t)k & 63u;
        const uint32_t c = (uint32_t)((rb.plane[0] >> sft) & 1ull) | (uint32_t)((rb.plane[1] >> sft) & 1ull) << 1 |
                           (uint32_t)((rb.plane[2] >> sft) & 1ull) << 2 | (uint32_t)((rb.plane[3] >> sft) & 1ull) << 3 |
                           (uint32_t)((rb.plane[4] >> sft) & 1ull) << 4;
        if (c == 0) {
          // the walk ran into the start of a sequence: its number is the rank of the terminator (bwt.c:120)
          const uint32_t iseq = (uint32_t)rank_term(ix, k);
          if (iseq < ix.nseq) { const uint64_t tax = ix.seq_taxid[iseq]; if (tax != ~0ull) add_tax(tax); }
          break;
        }
        const uint64_t ia = (c & 1u) ? 0ull : ~0ull, ib = (c & 2u) ? 0ull : ~0ull, ic = (c & 4u) ? 0ull : ~0ull,
                       id = (c & 8u) ? 0ull : ~0ull, ie = (c & 16u) ? 0ull : ~0ull;
        const uint64_t m = (rb.plane[0] ^ ia) & (rb.plane[1] ^ ib) & (rb.plane[2] ^ ic) & (rb.plane[3] ^ id) & (rb.plane[4] ^ ie);
        // k = C[c] + rank(c, k): the counts are absolute, or (wide) relative to the base of the row's 2^mb_shift rows
        uint64_t base = 0;
        if constexpr (WIDE) base = ix.mb_base[(size_t)((uint64_t)k >> ix.mb_shift) * 20 + (c - 1u)];
        k = (P)(base + rb.cnt[c - 1u] + popc64(m & ((1ull << sft) - 1ull)));
      }
    }
  }
  if constexpr (MANYROWS) {
#pragma unroll
    for (int q = 0; q < kMaxIds; q++) if (q < (int)nids) hit->taxid[q] = ix.tax_of_dense[idr[q]];
  } else if (dense) {
    for (uint32_t q = 0; q < nids; q++) hit->taxid[q] = ix.tax_of_dense[q == 0 ? id0 : hit->taxid[q]];
  }
  for (uint32_t q = nids; q < nsi; q++) hit->taxid[q] = 0;   // (the slots that held the matches and got no id)
  hit->n_ids = nids; hit->flags = flags;
  return true;
}


// Columns 6 and 7 of kaiju -v for a read of the second-generation MEM lanes (ConsumerThread.cpp:580-590, :614-623, :820-824),
// from what the VERBOSE instantiation of mem_lane2 left behind: the matches in the hit record (visiting order, as for
// mem_locate_read, which runs BEHIND this and turns them into ids) and, in the read's row of VerboseOut::acc, where every one
// of them lies in the read (fragment << 16 | start, in the order found).  Column 7: classify_length pushes
// fragment.substr(si->qi, si->ql) of the list HEAD of every fragment that holds a longest match (greedyExact prepends: the
// match found last; kaijux' maxMatches(.., 1): the one found first), in the order the fragments were searched - for every
// fragment, whatever ids_from_SI's limit does later.  Column 6: ids_from_SI's loop over the rows with its limit on the number
// of distinct taxon ids in front of every row (:805-807), get_suffix's walk for the sequence number (the row -> taxon table
// does not know the sequence), the name noted before the id is (:822-824, :834).  One lane per read: this is the verbose
// path, a walk per row is what the reference pays too.
// TEXT = false: column 6 only - the reads of the VERBOSE Greedy lane (greedy_lane2), which writes column 7 itself: its record
// holds the best matches in list order (eval_match_scores appends, ids_from_SI walks from the head), up to max_matches_SI = 20.
template <bool WIDE, bool TEXT = true>
KJ_HD void mem_verbose_read(const DevIndex &ix, const Params &p, const Batch &b, uint32_t r, const VerboseOut &vb) {
  typedef typename std::conditional<WIDE, uint64_t, uint32_t>::type P;
  const Hit *hit = b.hits + r;
  if (!(hit->flags & kHitLocPending) || !vb.n_acc) return;
  const uint32_t ncap = TEXT ? (uint32_t)kVbMem : (uint32_t)kMaxIds;
  const uint32_t nsi = hit->n_ids < ncap ? hit->n_ids : ncap;
  if constexpr (TEXT) {
    const bool xo = (p.flags & kParamXOrder) != 0;
    uint32_t where[kVbMem];
    for (uint32_t q = 0; q < nsi; q++) where[q] = vb.acc[(size_t)r * kVbAcc + q];
    vb.text_len[r] = 0;
    const ReadMeta rm = b.meta[r];
    const Frag *F = b.frags + rm.frag;
    const uint8_t *pep = b.pep + rm.pep;
    const uint32_t L = hit->best;
    for (uint32_t gs = 0; gs < nsi;) {
      const uint32_t fr = where[gs] >> 16;
      uint32_t ge = gs + 1;
      while (ge < nsi && (where[ge] >> 16) == fr) ge++;
      vb_text(vb, r, pep + F[fr].start + (where[xo ? gs : ge - 1] & 0xffffu), L, 0, nullptr, nullptr, 0);
      gs = ge;
    }
  }
  vb.n_acc[r] = 0;
  const P check = (P)((1ull << ix.chpt_exp) - 1ull);
  const RankBlock64 *const blk0 = ix.blocks64;
  uint64_t ids[kMaxIds];
  uint32_t nids = 0;
  for (uint32_t s = 0; s < nsi; s++) {
    const uint64_t es = hit->taxid[s];
    const P lo = WIDE ? (P)(es & ((1ull << kLocWideShift) - 1ull)) : (P)(uint32_t)es;
    const uint32_t len = WIDE ? (uint32_t)(es >> kLocWideShift) : (uint32_t)(es >> 32);
    const P rowend = lo + (P)(int32_t)len;
    for (P row = lo; row < rowend; row++) {
      if (nids > p.max_match_ids) return;                    // :805-807 (and every later match breaks at its first row)
      P k = row;
      uint32_t iseq = 0xffffffffu;
      for (;;) {
        if ((k & check) == 0) {
          const uint64_t sa_idx = ((uint64_t)k >> ix.chpt_exp) - ix.sa_skip;
          if (sa_idx < ix.n_sa) iseq = ix.sa_iseq[sa_idx];   // (beyond the samples the reference reads out of bounds: the row is skipped)
          break;
        }
        const RankBlock64 &rb = blk0[k >> 6];
        const uint32_t sft = (uint32_t)k & 63u;
        const uint32_t c = (uint32_t)((rb.plane[0] >> sft) & 1ull) | (uint32_t)((rb.plane[1] >> sft) & 1ull) << 1 |
                           (uint32_t)((rb.plane[2] >> sft) & 1ull) << 2 | (uint32_t)((rb.plane[3] >> sft) & 1ull) << 3 |
                           (uint32_t)((rb.plane[4] >> sft) & 1ull) << 4;
        if (c == 0) { iseq = (uint32_t)rank_term(ix, k); break; }      // the start of a sequence: its number is the rank of the terminator (bwt.c:120)
        const uint64_t ia = (c & 1u) ? 0ull : ~0ull, ib = (c & 2u) ? 0ull : ~0ull, ic = (c & 4u) ? 0ull : ~0ull,
                       id = (c & 8u) ? 0ull : ~0ull, ie = (c & 16u) ? 0ull : ~0ull;
        const uint64_t m = (rb.plane[0] ^ ia) & (rb.plane[1] ^ ib) & (rb.plane[2] ^ ic) & (rb.plane[3] ^ id) & (rb.plane[4] ^ ie);
        uint64_t base = 0;
        if constexpr (WIDE) base = ix.mb_base[(size_t)((uint64_t)k >> ix.mb_shift) * 20 + (c - 1u)];
        k = (P)(base + rb.cnt[c - 1u] + popc64(m & ((1ull << sft) - 1ull)));
      }
      if (iseq >= ix.nseq || !ix.seq_valid[iseq]) continue;
      vb_acc(vb, ix, r, iseq);
      const uint64_t tax = ix.seq_taxid[iseq];
      bool dup = false;
      for (uint32_t q = 0; q < nids; q++) dup = dup || ids[q] == tax;
      if (!dup && nids < (uint32_t)kMaxIds) ids[nids++] = tax;
    }
  }
}

// The same for a TEAM of T lanes per read (T a power of two, the lanes of a team next to each other in the wavefront): the rows
// of a match are walked T at a time, one row per lane, and the team's first lane then takes their ids in row order - the
// order, the duplicates and the 21-id cut of the sequential loop (:799-845).  On an index without the row -> sequence table
// (wide indexes; narrow ones that had no room for it) a row costs a walk of up to 2^e dependent LF steps, and a read whose
// matches hold seven rows costs seven walks one after the other: 18.5 ms per 5 M pairs on a 28 G-row index with every protein
// seven times (profiles/r04_refseq_ref), a third of the step.  Team: how a lane learns what its team mates found -
//   TeamWave<T> (device): shuffles;  TeamSerial<T> (host emulation): the T walks run one after the other into an array.
template <int T>
struct TeamSerial {
  uint64_t vals[T], vals2[T];
  template <class F> KJ_HD void compute(F &&f) { for (int tl = 0; tl < T; tl++) vals[tl] = f(tl); }
  KJ_HD uint64_t get(int q) const { return vals[q]; }
  // two rows per lane (KJ_LOC_ILP): f(tl, a, b) fills the ids of rows tl and T + tl of the round
  template <class F> KJ_HD void compute2(F &&f) { for (int tl = 0; tl < T; tl++) f(tl, vals[tl], vals2[tl]); }
  KJ_HD uint64_t get2(int q) const { return vals2[q]; }
  KJ_HD bool leader() const { return true; }
  KJ_HD bool from_leader(bool v) const { return v; }
  // the matches of the read (row | length, up to kLocMaxEntries), copied before the first id is written over them
  uint64_t ent[kLocMaxEntries];
  KJ_HD void load_entries(const uint64_t *rec, uint32_t n) { for (uint32_t q = 0; q < kLocMaxEntries; q++) ent[q] = q < n ? rec[q] : 0ull; }
  KJ_HD uint64_t entry(uint32_t q) const { return ent[q]; }
};
#if defined(__HIPCC__)
template <int T>
struct TeamWave {
  uint64_t mine, mine2 = ~0ull;
  template <class F> __device__ __forceinline__ void compute(F &&f) { mine = f((int)(threadIdx.x & (T - 1))); }
  __device__ __forceinline__ uint64_t get(int q) const {
    const int src = (int)((threadIdx.x & 63u) & ~(uint32_t)(T - 1)) + q;
    return (uint64_t)(uint32_t)__shfl((int)(uint32_t)mine, src, 64) | (uint64_t)(uint32_t)__shfl((int)(uint32_t)(mine >> 32), src, 64) << 32;
  }
  template <class F> __device__ __forceinline__ void compute2(F &&f) { f((int)(threadIdx.x & (T - 1)), mine, mine2); }
  __device__ __forceinline__ uint64_t get2(int q) const {
    const int src = (int)((threadIdx.x & 63u) & ~(uint32_t)(T - 1)) + q;
    return (uint64_t)(uint32_t)__shfl((int)(uint32_t)mine2, src, 64) | (uint64_t)(uint32_t)__shfl((int)(uint32_t)(mine2 >> 32), src, 64) << 32;
  }
  __device__ __forceinline__ bool leader() const { return (threadIdx.x & (T - 1)) == 0; }
  __device__ __forceinline__ bool from_leader(bool v) const { return __shfl((int)v, (int)((threadIdx.x & 63u) & ~(uint32_t)(T - 1)), 64) != 0; }
  // the matches of the read (row | length, up to kLocMaxEntries): lane t of the team keeps entries t, t + T, .. in registers - the
  // leader writes ids over them later - and hands one out by a shuffle
  static constexpr uint32_t kEnt = (kLocMaxEntries + T - 1) / T;
  uint64_t ent[kEnt];
  __device__ __forceinline__ void load_entries(const uint64_t *rec, uint32_t n) {
    const uint32_t tl = threadIdx.x & (T - 1);
#pragma unroll
    for (uint32_t q = 0; q < kEnt; q++) ent[q] = tl + q * T < n ? rec[tl + q * T] : 0ull;
  }
  __device__ __forceinline__ uint64_t entry(uint32_t q) const {
    const int src = (int)((threadIdx.x & 63u) & ~(uint32_t)(T - 1)) + (int)(q % T);
    const uint64_t v = ent[q / T];
    return (uint64_t)(uint32_t)__shfl((int)(uint32_t)v, src, 64) | (uint64_t)(uint32_t)__shfl((int)(uint32_t)(v >> 32), src, 64) << 32;
  }
};
#endif
#ifndef KJ_LOC_ILP
#define KJ_LOC_ILP 1                      // rows a lane of a locate team walks side by side (2: DESIGN.md 6b, round 6)
#endif
template <bool WIDE, int T, class Team>
KJ_HD void mem_locate_read_team(const DevIndex &ix, const Params &p, Hit *hit, Team &team) {
  typedef typename std::conditional<WIDE, uint64_t, uint32_t>::type P;
  const uint32_t fl0 = hit->flags;                            // (every lane of the team reads the same record: team-uniform)
  if (!(fl0 & kHitLocPending)) return;
  const uint32_t nsi = hit->n_ids;
  // the matches (row | length): every lane of the team keeps its share before the leader writes the first id
  team.load_entries(hit->taxid, nsi);
  const P check = (P)((1ull << ix.chpt_exp) - 1ull);
  const RankBlock64 *const blk0 = ix.blocks64;
  uint32_t nids = 0, flags = fl0 & ~kHitLocPending;           // (the leader's; a Greedy read may carry kHitSiCap already)
  uint64_t id0 = 0;
  auto add_tax = [&](uint64_t tax) {
    bool dup = false;
    if (nids >= 1 && tax == id0) dup = true;
    for (uint32_t q = 1; q < nids && !dup; q++) if (hit->taxid[q] == tax) dup = true;
    if (!dup && nids < (uint32_t)kMaxIds) { if (nids == 0) id0 = tax; hit->taxid[nids++] = tax; }
  };
  // the id of one row: ~0 = none (a name without a usable id, or a row beyond the samples, where the reference reads out of bounds)
  auto walk = [&](P k) -> uint64_t {
    if (ix.row_tax) {                                         // the walk below, precomputed for every row at index load (k_suffix_walk)
      const uint32_t t = ix.row_tax[k];
      return t != 0xffffffffu ? ix.tax_of_dense[t] : ~0ull;
    }
    for (;;) {
      if ((k & check) == 0) {
        const uint64_t sa_idx = ((uint64_t)k >> ix.chpt_exp) - ix.sa_skip;
        if (sa_idx >= ix.n_sa) return ~0ull;
        if constexpr (WIDE) { const uint32_t iseq = ix.sa_iseq[sa_idx]; return iseq < ix.nseq ? ix.seq_taxid[iseq] : ~0ull; }
        else return ix.sa_taxid[sa_idx];
      }
      const RankBlock64 &rb = blk0[k >> 6];
      const uint32_t sft = (uint32_t)k & 63u;
      const uint32_t c = (uint32_t)((rb.plane[0] >> sft) & 1ull) | (uint32_t)((rb.plane[1] >> sft) & 1ull) << 1 |
                         (uint32_t)((rb.plane[2] >> sft) & 1ull) << 2 | (uint32_t)((rb.plane[3] >> sft) & 1ull) << 3 |
                         (uint32_t)((rb.plane[4] >> sft) & 1ull) << 4;
      if (c == 0) {
        // the walk ran into the start of a sequence: its number is the rank of the terminator (bwt.c:120)
        const uint32_t iseq = (uint32_t)rank_term(ix, k);
        return iseq < ix.nseq ? ix.seq_taxid[iseq] : ~0ull;
      }
      const uint64_t ia = (c & 1u) ? 0ull : ~0ull, ib = (c & 2u) ? 0ull : ~0ull, ic = (c & 4u) ? 0ull : ~0ull,
                     id = (c & 8u) ? 0ull : ~0ull, ie = (c & 16u) ? 0ull : ~0ull;
      const uint64_t m = (rb.plane[0] ^ ia) & (rb.plane[1] ^ ib) & (rb.plane[2] ^ ic) & (rb.plane[3] ^ id) & (rb.plane[4] ^ ie);
      uint64_t base = 0;
      if constexpr (WIDE) base = ix.mb_base[(size_t)((uint64_t)k >> ix.mb_shift) * 20 + (c - 1u)];
      k = (P)(base + rb.cnt[c - 1u] + popc64(m & ((1ull << sft) - 1ull)));
    }
  };
  bool done = false;                                          // team-uniform
#pragma unroll
  for (uint32_t s = 0; s < kLocMaxEntries; s++) {
    const uint64_t es = team.entry(s);                        // (fully unrolled: which register of which lane is known here)
    if (s >= nsi || done) continue;
    const P lo = WIDE ? (P)(es & ((1ull << kLocWideShift) - 1ull)) : (P)(uint32_t)es;
    const uint32_t len = WIDE ? (uint32_t)(es >> kLocWideShift) : (uint32_t)(es >> 32);
    const P rowend = lo + (P)(int32_t)len;
#if KJ_LOC_ILP == 2
    // two rows per lane and round, their walks interleaved step by step: the loads of the two are in flight together (a walk is
    // a chain of dependent loads, and matches of fifteen or twenty-three rows - a refseq-class index holds every protein in many
    // near-identical copies - took two or three rounds of T rows one after the other)
    for (P row0 = lo; row0 < rowend && !done; row0 += (P)(2 * T)) {
      team.compute2([&](int tl, uint64_t &ta, uint64_t &tb) {
        const P ra = row0 + (P)tl, rb2 = row0 + (P)T + (P)tl;
        if (ix.row_tax) { ta = ra < rowend ? walk(ra) : ~0ull; tb = rb2 < rowend ? walk(rb2) : ~0ull; return; }
        P k[2] = {ra, rb2};
        bool fin[2] = {!(ra < rowend), !(rb2 < rowend)};
        uint64_t res[2] = {~0ull, ~0ull};
        while (!(fin[0] && fin[1])) {
          // rows that stand on a sample (or beyond the samples) are through; the others load their rank block
          const RankBlock64 *rbp[2];
#pragma unroll
          for (int h = 0; h < 2; h++) {
            if (!fin[h] && (k[h] & check) == 0) {
              const uint64_t sa_idx = ((uint64_t)k[h] >> ix.chpt_exp) - ix.sa_skip;
              if (sa_idx < ix.n_sa) {
                if constexpr (WIDE) { const uint32_t iseq = ix.sa_iseq[sa_idx]; res[h] = iseq < ix.nseq ? ix.seq_taxid[iseq] : ~0ull; }
                else res[h] = ix.sa_taxid[sa_idx];
              }
              fin[h] = true;
            }
            rbp[h] = blk0 + (fin[h] ? (P)0 : (k[h] >> 6));
          }
          uint64_t pl[2][5];
#pragma unroll
          for (int h = 0; h < 2; h++)
#pragma unroll
            for (int x = 0; x < 5; x++) pl[h][x] = rbp[h]->plane[x];
          uint32_t cc[2];
#pragma unroll
          for (int h = 0; h < 2; h++) {
            const uint32_t sft = (uint32_t)k[h] & 63u;
            cc[h] = (uint32_t)((pl[h][0] >> sft) & 1ull) | (uint32_t)((pl[h][1] >> sft) & 1ull) << 1 | (uint32_t)((pl[h][2] >> sft) & 1ull) << 2 |
                    (uint32_t)((pl[h][3] >> sft) & 1ull) << 3 | (uint32_t)((pl[h][4] >> sft) & 1ull) << 4;
          }
          uint64_t base[2] = {0, 0};
          uint32_t cnt[2] = {0, 0};
#pragma unroll
          for (int h = 0; h < 2; h++) {
            if (fin[h] || cc[h] == 0) continue;
            if constexpr (WIDE) base[h] = ix.mb_base[(size_t)((uint64_t)k[h] >> ix.mb_shift) * 20 + (cc[h] - 1u)];
            cnt[h] = rbp[h]->cnt[cc[h] - 1u];
          }
#pragma unroll
          for (int h = 0; h < 2; h++) {
            if (fin[h]) continue;
            if (cc[h] == 0) {
              // the walk ran into the start of a sequence: its number is the rank of the terminator (bwt.c:120)
              const uint32_t iseq = (uint32_t)rank_term(ix, k[h]);
              res[h] = iseq < ix.nseq ? ix.seq_taxid[iseq] : ~0ull;
              fin[h] = true;
              continue;
            }
            const uint32_t c = cc[h], sft = (uint32_t)k[h] & 63u;
            const uint64_t ia = (c & 1u) ? 0ull : ~0ull, ib = (c & 2u) ? 0ull : ~0ull, ic = (c & 4u) ? 0ull : ~0ull,
                           id = (c & 8u) ? 0ull : ~0ull, ie = (c & 16u) ? 0ull : ~0ull;
            const uint64_t m = (pl[h][0] ^ ia) & (pl[h][1] ^ ib) & (pl[h][2] ^ ic) & (pl[h][3] ^ id) & (pl[h][4] ^ ie);
            k[h] = (P)(base[h] + cnt[h] + popc64(m & ((1ull << sft) - 1ull)));
          }
        }
        ta = res[0]; tb = res[1];
      });
#pragma unroll
      for (int hq = 0; hq < 2 * T; hq++) {
        const uint64_t tax = hq < T ? team.get(hq) : team.get2(hq - T);
        if (team.leader() && !done && row0 + (P)hq < rowend) {
          if (nids > p.max_match_ids) { flags |= kHitIdCap; done = true; }     // :805-807, tested in front of every row
          else if (tax != ~0ull) add_tax(tax);
        }
      }
      done = team.from_leader(done);
    }
#else
    for (P row0 = lo; row0 < rowend && !done; row0 += (P)T) {
      team.compute([&](int tl) -> uint64_t { const P row = row0 + (P)tl; return row < rowend ? walk(row) : ~0ull; });
#pragma unroll
      for (int q = 0; q < T; q++) {
        const uint64_t tax = team.get(q);
        if (team.leader() && !done && row0 + (P)q < rowend) {
          if (nids > p.max_match_ids) { flags |= kHitIdCap; done = true; }     // :805-807, tested in front of every row
          else if (tax != ~0ull) add_tax(tax);
        }
      }
      done = team.from_leader(done);
    }
#endif
  }
  if (!team.leader()) return;
  for (uint32_t q = nids; q < nsi; q++) hit->taxid[q] = 0;   // (the slots that held the matches and got no id)
  hit->n_ids = nids; hit->flags = flags;
}

// ----------------------------------------------------------------------------
// Greedy lane: classify_greedyblosum (ConsumerThread.cpp:424-541), maxMatches /
// maxMatches_withStart (bwt.c:261-336), addAllMismatchVariantsAtPosSI (:346-395),
// eval_match_scores (:751-797)
// ----------------------------------------------------------------------------
constexpr int kMaxMismatch = 8;

// (host-side workload histograms, tests/tools only: -DKJ_HIST, see KJ_HISTO above)

struct GItem {               // one queue entry: a fragment or a substitution variant (80 bytes)
  uint64_t si0, si1;         // resume interval (variants)
  uint32_t key, start, len;  // start: peptide offset of the underlying fragment
  int32_t diff;
  uint32_t matchlen;         // variants: residues already matched; unchecked originals: SEG slot + 1
  uint32_t tot;              // sum of the BLOSUM62 diagonal over the whole sequence
  uint32_t msum;             // ... over the already matched suffix (variants)
  uint8_t num_mm, segchecked;
  uint16_t pad;
  uint16_t sub_pos[kMaxMismatch];
  uint8_t sub_aa[kMaxMismatch];   // index-alphabet codes
  uint32_t pad2[2];
};
static_assert(sizeof(GItem) == 80, "GItem is 80 bytes");

struct GMatch {              // one SI of the current fragment (bwt.h:25-34), 32 bytes
  uint64_t lo;
  uint32_t len;
  int32_t qi, ql;
  uint32_t ord;              // quirk-walk order (index of the t-th visited match)
  uint32_t dsum;             // sum of the diagonal scores over the match
  uint32_t psum;             // ... over the sequence from its start to the end of the match
};

struct GBest { uint64_t lo; uint32_t len; uint32_t pad; };
struct GBestV {              // verbose: where the peptide of a best SI comes from
  uint32_t start, qi, ql, num_mm;
  uint16_t sub_pos[kMaxMismatch];
  uint8_t sub_aa[kMaxMismatch];
};

struct GreedyScratch {
  GItem *pool;  uint32_t pool_cap;       // append-only item pool of the current read
  uint16_t *ord;                         // queue order: indices into pool, [pool_cap]
  GMatch *matches; uint32_t match_cap;
  GBest *best;                           // [64]
  uint8_t *win;
  GBestV *bestv = nullptr;               // [64], verbose output only
};

struct GQueue { uint32_t head, tail, npool; bool overflow; };

// multimap emplace: behind every entry whose key is >= key
KJ_HD void gq_push(const GreedyScratch &gs, GQueue &q, const GItem &it) {
  if (q.npool >= gs.pool_cap || q.tail >= gs.pool_cap) { q.overflow = true; return; }
  const uint32_t id = q.npool++;
  gs.pool[id] = it;
  uint32_t pos = q.tail;
  KJ_HISTO(2, q.tail - q.head);
  KJ_HISTO(3, (q.tail > q.head && gs.pool[gs.ord[q.tail - 1]].key < it.key) ? 1 : 0);
  while (pos > q.head && gs.pool[gs.ord[pos - 1]].key < it.key) { gs.ord[pos] = gs.ord[pos - 1]; pos--; }
  gs.ord[pos] = (uint16_t)id;
  q.tail++;
}

KJ_HD GItem gitem_from_frag(const Frag &f) {
  GItem it; it.si0 = it.si1 = 0; it.key = f.key; it.start = f.start; it.len = f.len; it.diff = 0;
  it.matchlen = f.flags >> kFragSlotShift; it.tot = f.key; it.msum = 0;
  it.num_mm = 0; it.segchecked = (uint8_t)(f.flags & kFragChecked); it.pad = 0; it.pad2[0] = it.pad2[1] = 0;
  for (int x = 0; x < kMaxMismatch; x++) { it.sub_pos[x] = 0; it.sub_aa[x] = 0; }
  return it;
}
struct GSink {               // SEG pieces enter the queue as checked fragments (:302,316)
  const GreedyScratch *gs; GQueue *q;
  KJ_HD void operator()(const Frag &f) const { gq_push(*gs, *q, gitem_from_frag(f)); }
};

// residue `pos` of the (possibly substituted) fragment of item t
KJ_HD void gwin_fill(LaneWin &lw, const uint8_t *pep, const GItem &t, int top) {
  win_fill(lw, pep + t.start, (int)t.len, top);
  for (uint32_t x = 0; x < t.num_mm; x++) {
    const int pz = (int)t.sub_pos[x];
    if (pz >= lw.q && pz < lw.q + kWin && pz < (int)t.len) lw.w[pz - lw.q] = t.sub_aa[x];
  }
}
KJ_HD uint32_t gwin_get(LaneWin &lw, const uint8_t *pep, const GItem &t, int pos) {
  if (pos < lw.q || pos >= lw.q + kWin) gwin_fill(lw, pep, t, pos);
  return lw.w[pos - lw.q];
}

enum GState : int {
  GS_FETCH, GS_POP, GS_START_J, GS_END_MATCH, GS_AFTER_SEARCH, GS_VAR_NEXT_MATCH, GS_VAR_NEXT_SUB,
  GS_EVAL, GS_FINISH, GS_LOC_NEXT_SI, GS_LOC_ROW, GS_LF_CHECK, GS_ADD_ID, GS_LOC_DONE,
  GS_STEP, GS_VSTEP, GS_LF, GS_KMER, GS_EXIT
};

KJ_HD void greedy_lane(const DevIndex &ix, const ConstTables &ct, const Params &p, const SegQueue &sq,
                       const Batch &b, const WorkList &wl, const GreedyScratch &gs,
                       const VerboseOut &vb = VerboseOut{nullptr, nullptr, nullptr, nullptr, 0},
                       const BigSeg *big = nullptr) {
  // (big: the exact pass - `sq` is its queue, the regions of a fragment come from the pool)
  int state = GS_FETCH;
  uint32_t r = 0;
  const uint8_t *pep = nullptr;
  GQueue q{0, 0, 0, false};
  GItem t;                                   // the fragment being searched
  for (int x = 0; x < kMaxMismatch; x++) { t.sub_pos[x] = 0; t.sub_aa[x] = 0; }
  t.si0 = t.si1 = 0; t.key = t.start = t.len = 0; t.diff = 0; t.matchlen = 0; t.tot = t.msum = 0;
  t.num_mm = 0; t.segchecked = 0; t.pad = 0; t.pad2[0] = t.pad2[1] = 0;
  int flen = 0, j = 0, i = 0;
  uint64_t lo = 0, hi = 0;
  uint32_t nm = 0;                           // matches of the current fragment
  uint32_t acc = 0, tail = 0;                // running diagonal sums: over the match / right of j
  int last_qi = 0;
  bool m_ovf = false;
  uint32_t best = 0, nbest = 0, flags = 0;
  // variant generation
  uint32_t vw = 0, vmatch = 0, vsub = 0, vorig = 0, vscore = 0, vlen = 0, norder = 0;
  // locate
  uint32_t cur = 0, nids = 0, iseq = 0;
  uint64_t row = 0, rowend = 0, k = 0;
  Hit *hit = nullptr;
  LaneWin lw{gs.win, 0};
  const uint64_t check = (1ull << ix.chpt_exp) - 1;
  const uint32_t n_items = wl.n_items_ptr ? *wl.n_items_ptr : wl.n_items;
  // seeds shorter than seed_length are never recorded
  const uint32_t kk = (ix.kmer_k >= 2 && ix.kmer_k <= p.seed_length && p.seed_length >= 3) ? ix.kmer_k : 0;
  uint32_t kidx = 0;

  for (;;) {
    while (state < GS_STEP) {
      switch (state) {
        case GS_FETCH: {
          const uint32_t item = fetch_work(wl.counter);
          if (item >= n_items) { state = GS_EXIT; break; }
          r = wl.reads ? wl.reads[item] : item;
          const ReadMeta rm = b.meta[r];
          pep = b.pep + rm.pep;
          const Frag *F = b.frags + rm.frag;
          const uint32_t nf = rm.nfrag & ~kNfragSegPending;
          q.head = q.tail = q.npool = 0; q.overflow = false;
          for (uint32_t f = 0; f < nf; f++) gq_push(gs, q, gitem_from_frag(F[f]));   // already in queue order
          best = 0; nbest = 0; flags = 0; m_ovf = false;
          // (substitution positions are kept in 16 bits: no protein is that long, but say so if one is)
          for (uint32_t f = 0; f < nf; f++) if (F[f].len > 65535u) flags |= kHitInternalOverflow;
          vb_reset(vb, r);
          state = GS_POP;
          break;
        }
        case GS_POP: {
          // getNextFragment(best_match_score), ConsumerThread.cpp:272-342
          if (q.head == q.tail || gs.pool[gs.ord[q.head]].key < best) { state = GS_FINISH; break; }
          t = gs.pool[gs.ord[q.head]];
          q.head++;
          if (p.seg && !t.segchecked) {
            // SEG found regions in this fragment (computed by the SEG pass): the parent is dropped,
            // its unmasked pieces are queued, and the next fragment is popped (:291-334)
            Frag f; f.start = t.start; f.len = t.len; f.key = t.key; f.flags = 0;
            if (t.matchlen && big) {
              const uint2 ent = big->index[t.matchlen - 1];
              if (ent.y == kBigSegLost) flags |= kHitInternalOverflow;
              else seg_split_regs(ct, p, BigRegs{big->lr + 2 * (size_t)ent.x, (int)ent.y}, pep, f, GSink{&gs, &q});
            } else if (t.matchlen) {
              const SegRec rec = sq.recs[t.matchlen - 1];
              if (rec.overflow) flags |= kHitInternalOverflow;
              seg_split(ct, p, rec, pep, f, GSink{&gs, &q});
            }
            break;
          }
          flen = (int)t.len;
          nm = 0;
          if (t.num_mm == 0) {
            // maxMatches(seq, len, seed_length, 0), bwt.c:261-296
            j = flen - 1;
            tail = 0;
            gwin_fill(lw, pep, t, j);
            state = GS_START_J;
          } else {
            // maxMatches_withStart, bwt.c:298-336
            j = flen - 1;
            i = j - (int)t.matchlen + 1;
            lo = t.si0; hi = t.si1;
            acc = t.msum;
            gwin_fill(lw, pep, t, i > 0 ? i - 1 : 0);
            state = i > 0 ? GS_STEP : GS_END_MATCH;
          }
          break;
        }
        case GS_START_J: {
          if (j < (int)p.seed_length - 1) { state = GS_AFTER_SEARCH; break; }
          if (kk && j >= (int)kk - 1) {
            kidx = 0; acc = 0;
            for (uint32_t q = 0; q < kk; q++) {
              const uint32_t cq = gwin_get(lw, pep, t, j - (int)q);
              kidx = kmer_index(kidx, cq);
              acc += (uint32_t)ct.diag_idx[cq];
            }
            state = GS_KMER;
            break;
          }
          const uint32_t c = gwin_get(lw, pep, t, j);
          lo = ix.C[c]; hi = ix.C[c + 1];
          acc = (uint32_t)ct.diag_idx[c];
          i = j;
          state = i > 0 ? GS_STEP : GS_END_MATCH;
          break;
        }
        case GS_END_MATCH: {
          const int l = j - i + 1;
          if (t.num_mm == 0) {
            if (l >= (int)p.seed_length && (nm == 0 || i < last_qi)) {      // bwt.c:276-278
              if (nm < gs.match_cap) {
                GMatch mm; mm.lo = lo; mm.len = (uint32_t)(int32_t)(hi - lo); mm.qi = i; mm.ql = l; mm.ord = 0;
                mm.dsum = acc; mm.psum = t.tot - tail;
                gs.matches[nm] = mm;
              } else m_ovf = true;
              nm++;
              last_qi = i;
            }
            if (i <= 1) state = GS_AFTER_SEARCH;                            // bwt.c:292
            else {
              tail += (uint32_t)ct.diag_idx[gwin_get(lw, pep, t, j)];
              j--;
              state = GS_START_J;
            }
          } else {
            // :443-449: after the last allowed mismatch the match must reach min_fragment_length
            const int Lreq = (t.num_mm == p.mismatches) ? (int)p.m : (int)t.matchlen;
            if (l >= Lreq) {
              GMatch mm; mm.lo = lo; mm.len = (uint32_t)(int32_t)(hi - lo); mm.qi = i; mm.ql = l; mm.ord = 0;
              mm.dsum = acc; mm.psum = t.tot;
              if (gs.match_cap > 0) gs.matches[0] = mm; else m_ovf = true;
              nm = 1;
            }
            state = GS_AFTER_SEARCH;
          }
          break;
        }
        case GS_AFTER_SEARCH: {
          KJ_HISTO(t.num_mm == 0 ? 0 : 1, nm);
          if (nm == 0 || m_ovf) { state = GS_POP; break; }   // (overflow: the read is redone in the retry pass)
          // order in which `si_it = si_it->samelen ? si_it->samelen : si_it->next` (:477) visits the
          // sorted list built by insert_SI_sorted (bwt.c:225-252): heads of the length classes in
          // descending length until a class with a samelen chain is met; that chain (latest
          // insertion first) is walked and ends the traversal
          norder = 0;
          if (p.mismatches > 0 && t.num_mm < p.mismatches) {
            if (nm == 1) { gs.matches[0].ord = 0; norder = 1; }
            else {
              int v = -1;
              for (uint32_t x = 0; x < nm; x++) if (gs.matches[x].ql > v) v = gs.matches[x].ql;
              for (;;) {
                uint32_t head = nm, cnt = 0;
                for (uint32_t x = 0; x < nm; x++) if (gs.matches[x].ql == v) { if (head == nm) head = x; cnt++; }
                gs.matches[norder++].ord = head;
                if (cnt >= 2) {
                  for (uint32_t x = nm; x-- > head + 1;) if (gs.matches[x].ql == v) gs.matches[norder++].ord = x;
                  break;
                }
                int nv = -1;
                for (uint32_t x = 0; x < nm; x++) if (gs.matches[x].ql < v && gs.matches[x].ql > nv) nv = gs.matches[x].ql;
                if (nv < 0) break;
                v = nv;
              }
            }
          }
          vw = 0;
          state = GS_VAR_NEXT_MATCH;
          break;
        }
        case GS_VAR_NEXT_MATCH: {
          if (vw >= norder) { state = GS_EVAL; break; }
          vmatch = gs.matches[vw++].ord;
          const GMatch it = gs.matches[vmatch];
          const uint32_t mre = (uint32_t)(it.qi + it.ql - 1);
          if (!(it.qi > 0 && mre + 1 >= p.m)) break;                       // :469
          // addAllMismatchVariantsAtPosSI(t, qi-1, erase_pos, it), :346-395
          vlen = (mre < (uint32_t)flen - 1) ? mre + 1 : (uint32_t)flen;     // fragment.erase(erase_pos)
          vorig = ct.idx_to_aa[gwin_get(lw, pep, t, it.qi - 1)];
          {
            const int sc = (int)it.psum + t.diff;                           // calcScore(fragment, f->diff)
            const uint32_t cs = sc > 0 ? (uint32_t)sc : 0u;
            vscore = cs - (uint32_t)(int32_t)ct.b62[vorig][vorig];          // unsigned wrap as in :363
          }
          vsub = 0;
          state = GS_VAR_NEXT_SUB;
          break;
        }
        case GS_VAR_NEXT_SUB: {
          if (vsub >= 19) { state = GS_VAR_NEXT_MATCH; break; }
          const uint32_t s = ct.subst[vorig][vsub];
          const int32_t after = (int32_t)(vscore + (uint32_t)(int32_t)ct.b62[vorig][s]);
          if (after >= (int32_t)best && after >= (int32_t)p.min_score) state = GS_VSTEP;
          else state = GS_VAR_NEXT_MATCH;                                   // break at the first too-low score
          break;
        }
        case GS_EVAL: {
          // eval_match_scores(si, t), :751-797.  Head of the list = longest match.
          int v1 = -1;
          for (uint32_t x = 0; x < nm; x++) if (gs.matches[x].ql > v1) v1 = gs.matches[x].ql;
          if (v1 < (int)p.m) { state = GS_POP; break; }                     // :482
          // recursion order: the samelen chains of the classes (descending length, while >= m) in
          // insertion order, then the class heads in ascending length
          int v = v1, vlast = v1;
          for (int pass = 0; pass < 2; pass++) {
            v = pass == 0 ? v1 : vlast;
            for (;;) {
              uint32_t head = nm;
              for (uint32_t x = 0; x < nm; x++) if (gs.matches[x].ql == v) { head = x; break; }
              for (uint32_t x = (pass == 0 ? head + 1 : head); x < (pass == 0 ? nm : head + 1); x++) {
                if (gs.matches[x].ql != v) continue;
                const GMatch mm = gs.matches[x];
                const int sc = (int)mm.dsum + t.diff;                       // calcScore(seq, qi, ql, diff)
                const uint32_t score = sc > 0 ? (uint32_t)sc : 0u;
                if (score < p.min_score) continue;
                if (score > best) { best = score; nbest = 0; }
                if (score == best) {
                  if (nbest < p.max_matches_SI && nbest < 64) {
                    GBest gb; gb.lo = mm.lo; gb.len = mm.len; gb.pad = 0;
                    if (vb.text && gs.bestv) {             // verbose: what frag->seq.substr(qi, ql) will need (:783,:789)
                      GBestV bv; bv.start = t.start; bv.qi = (uint32_t)mm.qi; bv.ql = (uint32_t)mm.ql; bv.num_mm = t.num_mm;
                      for (int x = 0; x < kMaxMismatch; x++) { bv.sub_pos[x] = t.sub_pos[x]; bv.sub_aa[x] = t.sub_aa[x]; }
                      gs.bestv[nbest] = bv;
                    }
                    gs.best[nbest++] = gb;
                  }
                  else flags |= kHitSiCap;
                }
              }
              if (pass == 0) {
                int nv = -1;
                for (uint32_t x = 0; x < nm; x++) if (gs.matches[x].ql < v && gs.matches[x].ql > nv) nv = gs.matches[x].ql;
                if (nv < 0 || nv < (int)p.m) { vlast = v; break; }
                v = nv;
              } else {
                if (v == v1) break;
                int nv = 0x7fffffff;
                for (uint32_t x = 0; x < nm; x++) if (gs.matches[x].ql > v && gs.matches[x].ql < nv) nv = gs.matches[x].ql;
                v = nv;
              }
            }
          }
          KJ_HISTO(4, nbest);
          state = GS_POP;
          break;
        }
        case GS_FINISH: {
          KJ_HISTO(5, q.npool); KJ_HISTO(6, nbest);
          hit = b.hits + r;
          nids = 0;
          hit->reserved = 0;
          if (q.overflow || m_ovf) {
            hit->best = 0;
            if (wl.retry_list) { wl.retry_list[append_slot(wl.retry_count)] = r; flags = kHitRetry; }
            else flags = kHitInternalOverflow;
            state = GS_LOC_DONE; break;
          }
          hit->best = nbest ? best : 0u;
          if (vb.text && gs.bestv)
            for (uint32_t x = 0; x < nbest; x++) {
              const GBestV bv = gs.bestv[x];
              vb_text(vb, r, pep + bv.start + bv.qi, bv.ql, bv.num_mm < (uint32_t)kMaxMismatch ? bv.num_mm : (uint32_t)kMaxMismatch,
                      bv.sub_pos, bv.sub_aa, (int)bv.qi);
            }
          cur = 0;
          state = GS_LOC_NEXT_SI;
          break;
        }
        case GS_LOC_NEXT_SI: {
          if (cur >= nbest) { state = GS_LOC_DONE; break; }
          row = gs.best[cur].lo; rowend = row + (uint64_t)(int64_t)(int32_t)gs.best[cur].len;
          cur++;
          state = GS_LOC_ROW;
          break;
        }
        case GS_LOC_ROW: {
          if ((int64_t)row >= (int64_t)rowend) { state = GS_LOC_NEXT_SI; break; }
          if (nids > p.max_match_ids) { flags |= kHitIdCap; state = GS_LOC_DONE; break; }
          k = row;
          state = GS_LF_CHECK;
          break;
        }
        case GS_LF_CHECK: {
          if ((k & check) == 0) {
            if (sa_lookup(ix, k, iseq)) state = GS_ADD_ID; else { row++; state = GS_LOC_ROW; }
          } else state = GS_LF;
          break;
        }
        case GS_ADD_ID: {
          vb_acc(vb, ix, r, iseq);
          add_id(ix, hit, nids, iseq);
          row++;
          state = GS_LOC_ROW;
          break;
        }
        case GS_LOC_DONE: {
          hit->n_ids = nids; hit->flags = flags;
          state = GS_FETCH;
          break;
        }
      }
    }
    if (state == GS_EXIT) break;
    if (state == GS_STEP) {
      const uint32_t c = gwin_get(lw, pep, t, i - 1);
      const uint64_t nlo = rank_c(ix, c, lo), nhi = rank_c(ix, c, hi);
      if (nlo >= nhi) state = GS_END_MATCH;
      else { lo = nlo; hi = nhi; i--; acc += (uint32_t)ct.diag_idx[c]; if (i == 0) state = GS_END_MATCH; }
    } else if (state == GS_VSTEP) {
      // UpdateSI(trans[substitute]) on the match's interval (:372)
      const GMatch it = gs.matches[vmatch];
      const uint32_t s = ct.subst[vorig][vsub];
      const uint32_t c = ct.aa_to_idx[s];
      const uint64_t nlo = rank_c(ix, c, it.lo), nhi = rank_c(ix, c, it.lo + (uint64_t)(int64_t)(int32_t)it.len);
      if (nlo < nhi) {
        GItem nf = t;
        nf.len = vlen;
        nf.key = (uint32_t)(int32_t)(vscore + (uint32_t)(int32_t)ct.b62[vorig][s]);
        nf.diff = t.diff + (int)ct.b62[vorig][s] - (int)ct.b62[s][s];
        nf.si0 = nlo; nf.si1 = nhi;
        nf.matchlen = (uint32_t)it.ql + 1;
        nf.tot = it.psum - (uint32_t)ct.b62[vorig][vorig] + (uint32_t)ct.b62[s][s];
        nf.msum = it.dsum + (uint32_t)ct.b62[s][s];
        nf.segchecked = 1;
        if (t.num_mm < kMaxMismatch) { nf.sub_pos[t.num_mm] = (uint16_t)(it.qi - 1); nf.sub_aa[t.num_mm] = (uint8_t)c; }
        nf.num_mm = (uint8_t)(t.num_mm + 1);
        gq_push(gs, q, nf);
      }
      vsub++;
      state = GS_VAR_NEXT_SUB;
    } else if (state == GS_LF) {
      const uint32_t c = symbol_at(ix, k);
      if (c == 0) { iseq = (uint32_t)rank_term(ix, k); state = GS_ADD_ID; }
      else { k = rank_c(ix, c, k); state = GS_LF_CHECK; }
    } else {
      kmer_lookup(ix, kidx, lo, hi);
      if (lo >= hi) { i = j; state = GS_END_MATCH; }       // seed shorter than kk: never recorded, i > 1
      else { i = j - (int)kk + 1; state = i > 0 ? GS_STEP : GS_END_MATCH; }
    }
  }
}

// ----------------------------------------------------------------------------
// Greedy lane, second generation (indexes below 2^32 rows).  Same results as greedy_lane, built
// for the SIMT machine:
//   * every iteration of the lane loop has ONE branch-free load phase, then the arithmetic on
//     what arrived, then bookkeeping (mem_lane2);
//   * the work of a read falls into a FAST part - extending matches letter by letter and walking
//     to suffix-array samples, nine tenths of all iterations - and a SLOW part - queue, variants,
//     scores.  64 lanes in 64 different places of the slow part would make every iteration pay for
//     all of it, so the slow part only runs in every (gate+1)-th iteration ("heavy" iterations,
//     wave-uniform); a lane that reaches it in between parks (G_WAIT) until then;
//   * the 19 substitutions tried behind a match (addAllMismatchVariantsAtPosSI) all extend the
//     same interval, i.e. read the same two rank blocks: one G_VMULTI step ranks all 20 letters
//     and queues the surviving variants;
//   * lane memory: registers, an LDS row (peptide window with the substitutions of the current
//     variant patched in; the lengths of the matches of the current fragment, from which the
//     order in which the reference walks its sorted SI list is recomputed; the priorities of the
//     queued variants / SEG pieces, key << 16 | 0xffff - sequence number: the multimap of the
//     reference pops the largest key, earliest insertion first = the largest priority) and global
//     scratch (queued items of 128 bytes: state + ready-made window; match records).
//     The originals of the read are NOT queued: stage 1 wrote them in queue order, so they are
//     merged in from their list (an original precedes queued entries of the same key, having
//     been inserted before them).
// A read that does not fit the bounds (more than kGMaxMAll seeds in one fragment, more than
// kGSlotsAll live queue entries, keys or lengths >= 2^16) is sent to the retry pass (greedy_lane).
// ----------------------------------------------------------------------------
#ifdef KJ_G_SMALL                                  // tests: exercise the spill and retry paths
constexpr int kGMaxM = 1, kGMaxMAll = 2, kGSlots = 1, kGSlotsAll = 3;
#else
constexpr int kGMaxM = 8, kGMaxMAll = 256;
// queue slots: 12 priorities in LDS, 512 slots in all (64 KB of items per lane in device memory).  With 128, reads of the
// benchmark's viruses-size index never ran out, but on a 1 G-row index one push in ten found all slots handed out (the
// overflow area is append-only within a read: a dead slot then has to be searched for) and 2 reads in 10 000 had more than
// 128 LIVE entries - the retry pass they went to took five times as long as the whole batch (profiles/r03_l10).  The most
// slots a read of that workload asked for: 496.
constexpr int kGSlots = 12, kGSlotsAll = 512;
#endif
constexpr int kGSubStride = 17;                  // six words of substitutions + eleven of slow-part state
constexpr int kGExtPass = 4;                     // 16-byte loads in flight per pass over the queue's overflow area (4 priorities each;
                                                 // 8: slower)
// LDS rows (dwords): strides chosen odd (byte / dword accesses) or 4 x odd (16-byte accesses)
#ifndef KJ_G_SMALL
constexpr int kGWinStride = 17, kGMqStride = 4, kGPrioStride = 12;     // 132 bytes per lane + 68 (kGSubStride) = 200: three blocks of
                                                                        // 256 lanes + tables = 159 744 of the CU's 163 840 bytes of LDS
constexpr int kGreedyWavesPerSimd = 3;           // (168 VGPRs without a spill; with the slow-part state in registers instead of LDS: 247, two wavefronts)
#else
constexpr int kGWinStride = 17, kGMqStride = 13, kGPrioStride = 44;
constexpr int kGreedyWavesPerSimd = 2;
#endif

struct GMatch2 { uint32_t lo, len, qiql, dp; };          // qi | ql << 16, dsum | psum << 16
                                                         // (wide: lo 40 bits | qi 12 | ql 12, then len 32 | dsum 16 | psum 16)
struct GBest2 { uint32_t lo, len; };
struct GBest2W { uint64_t lo; uint32_t len, pad; };      // wide indexes
constexpr uint32_t kGWideMaxFrag = 4095u;                // wide: fragment positions are kept in 12 bits (longer: retry pass)
struct GreedyScratch2 {
  uint8_t *win;                // LDS: peptide window, kWin bytes, 4-byte aligned
  uint16_t *mq;                // LDS: lengths of the first kGMaxM matches
  uint32_t *prio;              // LDS: priorities of the first kGSlots queue slots, 16-byte aligned
  u128 *pool;                  // kGSlotsAll items of 128 bytes (+ 64 bytes of slack behind the last lane's)
  uint32_t *prio_ext;          // priorities of the slots kGSlots .. kGSlotsAll-1
  GMatch2 *matches;            // kGMaxMAll
  uint16_t *mq_ext;            // lengths of the matches kGMaxM .. kGMaxMAll-1
  GBest2 *best;                // 64
  GBest2W *bestw;              // 64, wide indexes (one of the two is used)
  uint32_t gate;               // heavy iterations: (iteration & gate) == 0
  unsigned long long *prof;    // -DKJ_PROF: the wavefront's LDS row (2 + 3 * PS_N)
  uint32_t lane;               // the device-memory pointers above are the bases of all lanes, this is the lane's number
  uint32_t *sub;               // LDS, kGSubStride words: the substitutions of the variant at hand + slow-part state
  // VERBOSE instantiation (kaiju -v): where the peptide of every best match comes from - base of all lanes, 64 per lane - and
  // the reads' column-7 rows (the accessions of column 6 follow from the record: mem_verbose_read<.., false>)
  GBestV *bestv = nullptr;
  VerboseOut vb{nullptr, nullptr, nullptr, nullptr, 0};
};

#ifdef KJ_NO_CHAIN_PRUNE                            // (A/B measurements, tests: the lane that queues every variant)
constexpr bool kChainPrune = false;
#else
constexpr bool kChainPrune = true;
#endif
// Can the variant item whose match [pz, j] (l0 = j - pz + 1 letters, nmm substitutions so far) lies on ONE database row, or any
// item that descends from it, ever hold a match of p.m letters?  (greedy_lane2: kChainPrune.)  Everything the reference does with
// such an item is an ungapped comparison of the fragment with the database text in front of the row's suffix: the match grows
// while the letters agree (UpdateSI on one row, bwt.c:160-173), a substitution (:346-395) takes the text's letter at a letter
// that differs and costs one of the p.mismatches, and the item dies at the next difference once they are spent (:443-449).  So
// the longest match the chain can reach ends in front of the (allowed + 1)-th difference - a matter of the MASK of differences
// between the sixteen letters in front of the match on both sides, no loop, no score.  Scores, thresholds, terminators and the
// order of the queue can only end the chain earlier.  tx: text[tp - 16, tp) of the match's text position tp; win / wq: the
// lane's window of the fragment (the letters in front of pz are the fragment's own: a chain substitutes from right to left).
// Conservative: what it cannot see (letters in front of the window, a chain longer than sixteen letters) counts as reachable.
// sc0: the score of the item's own match (eval_match_scores' m_dsum + diff).  A letter the chain gains adds at most 11 to it
// (the largest entry of the BLOSUM62 diagonal: W), a substitution at most kMaxSubstScore (the largest entry off the diagonal;
// host_tables.cpp checks both against the table): if even that stays below min_score, no item of the chain passes the gate.
constexpr int kMaxDiagScore = 11, kMaxSubstScore = 4;
#ifndef KJ_CHAIN_ROWS
#define KJ_CHAIN_ROWS 4
#endif
constexpr int kChainRows = KJ_CHAIN_ROWS;     // variants on up to this many rows are tested (every row's chain); 1: round 6's first form
#ifdef KJ_NO_WIDE_CHAIN_PRUNE
constexpr bool kWideChainPrune = false;       // (A/B measurements)
#else
constexpr bool kWideChainPrune = true;        // the test in greedy_lane2<.., WIDE> as well (24 bytes of scratch a lane at three wavefronts
                                              // per SIMD; k_greedy2_wide on an index without one-row matches: 47.8 ms with and without)
#endif
// thr: the score an item has to reach to matter - min_score, or the read's best score so far if that is higher (a match below
// `best` changes nothing, :757-775, and best only grows)
KJ_HD bool kj_chain_hopeless(const Params &p, const uint8_t *win, int wq, int pz, int j, const u128 &tx, uint32_t nmm, int sc0, int thr) {
  const int a = pz - wq;                                    // letters of the fragment in front of pz that the window holds
  if (a <= 0) return pz == 0 ? (j - pz + 1 < (int)p.m || sc0 < thr) : false;
  if (a < 16 && wq > 0) return false;
  // f: the sixteen letters of the fragment in front of pz, laid out like tx (the letter at distance k, k = 0 next to the match,
  // in byte 15 - k); letters in front of the fragment's start read as zeros, which differ from every letter
  const uint32_t *w32 = reinterpret_cast<const uint32_t *>(win);
  u128 f;
  if (a >= 16) {
    const int o = a - 16, d = o >> 2;
    const uint32_t sh = (uint32_t)(o & 3);
    const uint32_t q0 = w32[d], q1 = w32[d + 1], q2 = w32[d + 2], q3 = w32[d + 3], q4 = sh ? w32[d + 4] : 0u;
    const uint32_t r0 = kj_alignbyte(q1, q0, sh), r1 = kj_alignbyte(q2, q1, sh), r2 = kj_alignbyte(q3, q2, sh), r3 = kj_alignbyte(q4, q3, sh);
    f.x = r0 | (uint64_t)r1 << 32; f.y = r2 | (uint64_t)r3 << 32;
  } else {
    // the window starts with the fragment (wq = 0): its first a letters, moved up to bytes 16 - a .. 15
    const uint64_t lo = w32[0] | (uint64_t)w32[1] << 32, hi = w32[2] | (uint64_t)w32[3] << 32;
    const uint32_t sb = 8u * (uint32_t)(16 - a);            // 8 .. 120 bits up
    if (sb >= 64u) { f.x = 0; f.y = lo << (sb - 64u); }
    else { f.x = lo << sb; f.y = (hi << sb) | (lo >> (64u - sb)); }
    // (bytes of the window behind letter a - 1 that were shifted in: only the low 16 - a bytes are meant to be zero; the
    //  letters at and behind pz landed above byte 15 and fell out)
  }
  auto diff_bits = [](uint64_t x) -> uint32_t {             // bit i = byte i of x is not zero
    const uint64_t t = ((x & 0x7f7f7f7f7f7f7f7full) + 0x7f7f7f7f7f7f7f7full) | x;
    return (uint32_t)(((t & 0x8080808080808080ull) >> 7) * 0x0102040810204080ull >> 56);
  };
  // mask of differences by distance: byte 15 - k of f ^ tx <-> bit k
  const uint32_t mx = diff_bits(f.x ^ tx.x), my = diff_bits(f.y ^ tx.y);      // bit i <-> byte i (low half), byte 8 + i (high half)
  uint32_t m16 = 0;
#if defined(__HIP_DEVICE_COMPILE__)
  m16 = (__brev(my) >> 24) | (__brev(mx) >> 24) << 8;
#else
  for (int i = 0; i < 8; i++) { m16 |= ((my >> i) & 1u) << (7 - i); m16 |= ((mx >> i) & 1u) << (15 - i); }
#endif
  if (a < 16) m16 |= 0xffffu << a;                          // in front of the fragment's start nothing matches and nothing is substituted
  // the substitutions that are left take the first differences; the chain ends in front of the next one
  uint32_t m = m16;
  for (uint32_t q = nmm; q < p.mismatches && m; q++) m &= m - 1u;
  if (m == 0) return false;                                 // fewer differences than that within sixteen letters: it may go on
  int e = 0;
  while (!((m >> e) & 1u)) e++;
  if (a < 16 && e > a) e = a;
  if (j - pz + 1 + e < (int)p.m) return true;
  // of the e letters gained, those at a difference were substituted
  const int nsub = (int)popc64((uint64_t)(m16 & ((1u << e) - 1u)));
  return sc0 + kMaxDiagScore * (e - nsub) + kMaxSubstScore * nsub < thr;
}

enum GKind : int { G_STEP, G_KMER, G_PROBE,                                              // fast (G_PROBE: a k-mer lookup, kGreedyProbe)
                   G_VMULTI, G_META, G_FRAG, G_FILL, G_POPITEM, G_MLOAD, G_WAIT, G_IDLE,   // heavy iterations only
                   G_EXIT };
enum GBk : int { GB_NONE, GB_END_MATCH, GB_START_J,                                        // fast
                 GB_AFTER_SEARCH, GB_VAR_NEXT, GB_VAR_MATCH, GB_EVAL_NEXT, GB_EVAL_MATCH, GB_POP, GB_FINISH, GB_DONE };
enum GFillRet : int { FR_START_J, FR_STEP, FR_VARM };

// the lane's scratch in device memory: pointers kept per lane, or (three wavefronts per SIMD) recomputed from the lane number
#define GS_POOL (gs.pool + (size_t)gs.lane * (8 * kGSlotsAll))
#define GS_PRIO_EXT (gs.prio_ext + (size_t)gs.lane * (kGSlotsAll - kGSlots))
#define GS_MATCHES (gs.matches + (size_t)gs.lane * kGMaxMAll)
#define GS_MQ_EXT (gs.mq_ext + (size_t)gs.lane * (kGMaxMAll - kGMaxM))
#define GS_BEST (gs.best + (size_t)gs.lane * 64)
#define GS_BESTW (gs.bestw + (size_t)gs.lane * 64)
// WIDE: indexes of 2^32 rows and more - 64-bit positions, block counts relative to DevIndex::mb_base, the k-mer TABLE of
// 16-byte entries instead of the k-mer lines, sequence numbers instead of taxon ids at the sampled rows
// why a read left the lane for the retry pass (-DKJ_OVF_STATS builds): counters in the batch's counter block (words 40..47;
// kaiju_gpu_get_stats prints them under KAIJU_GPU_OVF_STATS=1).  0 key / sequence number beyond 16 bits, 1 more than kGSlotsAll live queue entries, 2 an
// original beyond the lane's length fields, 3 a SEG piece beyond them, 4 a variant beyond them, 5 more than kGMaxMAll matches in a
// fragment, 6 (wide) an interval of 2^32 rows and more
#if defined(__HIP_DEVICE_COMPILE__) && defined(KJ_OVF_STATS)         // (a diagnostic build: tests/tools/mem_variants.sh, variant ovf)
#define KJ_OVF(wl, why) atomicAdd(reinterpret_cast<uint32_t *>(reinterpret_cast<uintptr_t>((wl).counter) & ~(uintptr_t)255) + 40 + (why), 1u)
#else
#define KJ_OVF(wl, why) ((void)0)
#endif
template <bool COUNT = false, bool WIDE = false, bool VERBOSE = false>
KJ_HD void greedy_lane2(const DevIndex &ix, const ConstTables &ct, const Params &p, const SegQueue &sq,
                        const Batch &b, const WorkList &wl, const GreedyScratch2 &gs) {
  typedef typename std::conditional<WIDE, uint64_t, uint32_t>::type P;
  uint32_t oc[kOpcN];
  if constexpr (COUNT) for (int x = 0; x < kOpcN; x++) oc[x] = 0;
  int kind = G_IDLE, bk_pend = GB_NONE;
  // read
  uint32_t r = 0, nf = 0, fo = 0, fbase = 0;
  uint64_t pepoff = 0;
  // (state that only the slow part touches lives in the lane's LDS row: gs.sub[6..14])
  // (length and key of the prefetched original share a word, 16 bits each, saturating: a fragment that reaches 65535 in
  // either goes to the retry pass; the word this frees holds the largest priority in the queue's overflow area)
  uint32_t &on_start = gs.sub[6], &on_kl = gs.sub[7], &ext_max = gs.sub[8], &on_flags = gs.sub[9];
  auto on_pack = [](uint32_t len, uint32_t key) -> uint32_t { return (len > 0xffffu ? 0xffffu : len) | (key > 0xffffu ? 0xffffu : key) << 16; };
  uint32_t &b0lo = gs.sub[10], &b0len = gs.sub[11];
  on_start = on_kl = ext_max = on_flags = b0lo = b0len = 0;
  uint32_t best = 0, nbest = 0, flags = 0;
  bool ovf = false;
  // queue of variants and SEG pieces
  uint32_t &qseq = gs.sub[15], &pslot = gs.sub[16];
  qseq = pslot = 0;
  uint32_t qn = 0, qlive = 0;
  // the fragment being searched
  uint32_t t_start = 0, t_len = 0, t_matchlen = 0, t_tot = 0, t_msum = 0, t_nmm = 0;
  int32_t t_diff = 0;
  // substituted positions (16 bit) / letters (8 bit): six words of the lane's LDS row (touched by the slow part only)
  uint32_t &sp0 = gs.sub[0], &sp1 = gs.sub[1], &sp2 = gs.sub[2], &sp3 = gs.sub[3], &sa0 = gs.sub[4], &sa1 = gs.sub[5];
  sp0 = sp1 = sp2 = sp3 = sa0 = sa1 = 0;
  int flen = 0, j = 0, i = 0, last_qi = 0;
  P lo = 0, hi = 0;
  P sz_i = 1, sz_q = 1;                       // kSpanEq: rows of the interval the last search ended in / of the last recorded match
  uint32_t c = 1, cj = 1, acc = 0, tail = 0, nm = 0, kidx = 0;
  bool m_ovf = false;
  // the match at hand
  P m_lo = 0;
  uint32_t m_len = 0, m_qi = 0, m_ql = 0, m_dsum = 0, m_psum = 0;
  // walk over the matches for the substitution variants / for the scores
  int vi_v = -1, vi_x = 0, vi_phase = 2, vi_head = 0;
  int ev_pass = 0, ev_v = 0, ev_x = -1, ev_v1 = 0;
  bool ev_done = false;
  uint32_t mx = 0, ml_for = 0;
  // variant generation
  uint32_t &vorig = gs.sub[12], &vscore = gs.sub[13], &vlen = gs.sub[14];
  vorig = vscore = vlen = 0;
  // (the ids: a read leaves its best matches - up to max_matches_SI = 20 of them - in the hit record, k_mem_locate* walk them;
  //  until round 5 a read with more than one walked here, one lane of the wavefront at work)
  uint32_t nids = 0;
#define KJ_G_HIT (b.hits + r)                    /* (recomputed: two registers less than a pointer kept per lane) */
  int fill_top = 0, fill_ret = FR_START_J;
  bool fill_pref = false;

  uint8_t *const win = gs.win;
  uint16_t *const mq = gs.mq;
  uint32_t *const prio = gs.prio;
  int wq = 0;                                   // fragment position of win[0]
  const P check = (P)((1u << ix.chpt_exp) - 1);
  const uint32_t n_items = wl.n_items_ptr ? *wl.n_items_ptr : wl.n_items;
  const uint32_t ktab = WIDE ? ix.kmer_k : ix.kline_k;      // (narrow: the k-mer lines; wide: the table)
  const uint32_t kk = (ktab >= 2 && ktab <= p.seed_length && p.seed_length >= 3 &&
                       (WIDE ? (const void *)ix.kmer64 : (const void *)ix.kline)) ? ktab : 0;
  const uint32_t nwaves = kj_nwaves();
  uint32_t wnext = 0, wend = 0, itc = 0;
  const RankBlock64 *const blk0 = ix.blocks64;
  // k-mer lines (DevIndex::kline): kcode = the line of end position j (the word w[j-kk+1 .. j-1]), kidx = line and entry of
  // the lookup.  The line (and the diagonal sum of the k-mer) of end position j-1 follows from that of j
  uint32_t kpow = 1;
  for (uint32_t q = WIDE ? 1 : 2; q < kk; q++) kpow *= 20u; // 20^(kk-2): digit of w[j-1] in the line number (wide: 20^(kk-1), digit
                                                            // of w[j] in the table index)
  uint32_t kacc = 0;                            // (the line number lives in kidx >> 6)
  bool kroll = false;                           // kidx >> 6 / kacc / cj describe end position j+1 of this search
  bool skipj = false;                           // the k-mer that ends at the next end position (j - 1) is not in the index
  // BLOSUM62 diagonal by index-alphabet code, 4 bits each (values 4..11)
  uint64_t dg0 = 0, dg1 = 0;
  for (int x = 0; x < 16; x++) dg0 |= (uint64_t)((uint32_t)ct.diag_idx[x] & 15u) << (4 * x);
  for (int x = 16; x < 32; x++) dg1 |= (uint64_t)((uint32_t)ct.diag_idx[x] & 15u) << (4 * (x - 16));
#ifdef __HIP_DEVICE_COMPILE__
  // (the same in every lane: scalar registers)
  dg0 = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)dg0) |
        (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(dg0 >> 32)) << 32;
  dg1 = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)dg1) |
        (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(dg1 >> 32)) << 32;
  kpow = (uint32_t)__builtin_amdgcn_readfirstlane((int)kpow);
#endif

  auto diag = [&](uint32_t cc) -> uint32_t { return (uint32_t)(((cc & 16u) ? dg1 : dg0) >> (4u * (cc & 15u))) & 15u; };
  auto in_win = [&](int pos) -> bool { return pos >= wq && pos < wq + kWin; };
  auto mq_get = [&](uint32_t x) -> int { return (int)(x < (uint32_t)kGMaxM ? mq[x] : GS_MQ_EXT[x - kGMaxM]); };
  auto mq_max_below = [&](int bound) -> int {   // largest match length < bound, -1 if none
    int v = -1;
    for (uint32_t x = 0; x < nm; x++) { const int q = mq_get(x); if (q < bound && q > v) v = q; }
    return v;
  };
  auto mq_head = [&](int v) -> uint32_t {       // first match of length v (head of its class)
    uint32_t x = 0;
    while (x < nm && mq_get(x) != v) x++;
    return x;
  };
  // multimap emplace of a variant / SEG piece: returns the slot (or ~0)
  // The queue's priorities: slots 0 .. kGSlots-1 in LDS (freed slots are used again; qlive = the live ones among them), further
  // entries in the overflow area in device memory, APPEND-ONLY within a read (qn = slots handed out so far); ext_max = the
  // largest priority there, so that a pop only reads the overflow area when the best entry lies in it (live entries at a
  // push: 12 and more in 45 % of the cases - with a scan of the overflow area in every pop and every push, a fifth of this
  // lane's cycles went there, profiles/r03_l7).
  auto push_slot = [&](uint32_t key, uint32_t seq) -> uint32_t {
    KJ_HISTO(7, (qlive + (qn > (uint32_t)kGSlots ? qn - (uint32_t)kGSlots : 0u)) >> KJ_HIST_QSHIFT);   // (slots handed out: live entries + popped overflow slots)
    if (key > 0xffffu || seq >= 0xfffeu) { ovf = true; KJ_OVF(wl, 0); return ~0u; }
    const uint32_t pr = key << 16 | (0xffffu - seq);
    const uint32_t nl = qn < (uint32_t)kGSlots ? qn : (uint32_t)kGSlots;
    uint32_t slot;
    if (qlive < nl) {
      // a free slot among the LDS slots handed out so far
      slot = 0;
      if constexpr (kGSlots % 4 == 0) {
        bool got = false;
        for (uint32_t q = 0; q < nl && !got; q += 4) {
          const u128 v = *reinterpret_cast<const u128 *>(prio + q);
          const uint32_t e0 = (uint32_t)v.x, e1 = (uint32_t)(v.x >> 32), e2 = (uint32_t)v.y, e3 = (uint32_t)(v.y >> 32);
          const uint32_t x = e0 == 0 ? 0u : e1 == 0 ? 1u : e2 == 0 ? 2u : e3 == 0 ? 3u : 4u;
          if (x < 4u && q + x < nl) { slot = q + x; got = true; }
        }
      } else
        while (prio[slot] != 0) slot++;
      qlive++;
    } else if (qn < (uint32_t)kGSlots) { slot = qn++; qlive++; }
    else {
      if (qn < (uint32_t)kGSlotsAll) slot = qn++;
      else {
        // every slot has been handed out once (seven reads in 100 000 of the benchmark): a popped slot of the overflow area
        // is used again; none free = more than kGSlotsAll live entries: retry pass
        slot = kGSlots;
        while (slot < (uint32_t)kGSlotsAll && GS_PRIO_EXT[slot - kGSlots] != 0) slot++;
        if (slot >= (uint32_t)kGSlotsAll) { ovf = true; KJ_OVF(wl, 1); return ~0u; }
      }
      GS_PRIO_EXT[slot - kGSlots] = pr;
      if (pr > ext_max) ext_max = pr;
      if constexpr (COUNT) oc[kOpcPush]++;
      return slot;
    }
    prio[slot] = pr;
    if constexpr (COUNT) oc[kOpcPush]++;
    return slot;
  };
  // eval_match_scores on one match (ConsumerThread.cpp:751-797)
  auto eval_match = [&]() {
    const int sc = (int)m_dsum + t_diff;                    // calcScore(seq, qi, ql, diff)
    const uint32_t score = sc > 0 ? (uint32_t)sc : 0u;
    if (score < p.min_score) return;
    if (score > best) { best = score; nbest = 0; }
    if (score == best) {
      if (nbest < p.max_matches_SI && nbest < 64) {
        if constexpr (VERBOSE) {
          // what frag->seq.substr(qi, ql) will need (ConsumerThread.cpp:783, :789): the item's place in the read, the match's in
          // the item, the substitutions the variant carries (positions relative to the item's start)
          GBestV bv; bv.start = t_start; bv.qi = m_qi; bv.ql = m_ql; bv.num_mm = t_nmm;
          for (uint32_t x = 0; x < (uint32_t)kMaxMismatch; x++) {
            const uint32_t pw = x < 2 ? sp0 : x < 4 ? sp1 : x < 6 ? sp2 : sp3;
            bv.sub_pos[x] = (uint16_t)((pw >> ((x & 1u) * 16u)) & 0xffffu);
            bv.sub_aa[x] = (uint8_t)((x < 4 ? sa0 : sa1) >> ((x & 3u) * 8u));
          }
          gs.bestv[(size_t)gs.lane * 64 + nbest] = bv;
        }
        if constexpr (WIDE) { GBest2W gb; gb.lo = m_lo; gb.len = m_len; gb.pad = 0; GS_BESTW[nbest] = gb; }
        else if (nbest == 0) { b0lo = (uint32_t)m_lo; b0len = m_len; }
        else { GBest2 gb; gb.lo = (uint32_t)m_lo; gb.len = m_len; GS_BEST[nbest] = gb; }
        nbest++;
      } else flags |= kHitSiCap;
    }
  };

#if defined(KJ_STATS) && defined(__HIP_DEVICE_COMPILE__)
  unsigned long long st_t = clock64(), st_heavy = 0, st_load = 0, st_fast = 0, st_slow = 0, st_book = 0, st_nheavy = 0;
#define KJ_TICK(acc) { const unsigned long long _n = clock64(); acc += _n - st_t; st_t = _n; }
#else
#define KJ_TICK(acc)
#endif
  for (;;) {
    KJ_P(PS_HEAD);
    // wave-uniform: every (gate+1)-th iteration, or as soon as many lanes wait for the slow part
    bool heavy = (itc & (gs.gate & 0xffu)) == 0;
    if (!heavy && (gs.gate >> 8) != 0)
      heavy = popc64(kj_ballot(kind >= G_VMULTI && kind <= G_IDLE)) >= (gs.gate >> 8);
    itc = heavy ? 1u : itc + 1u;
    KJ_TICK(st_book)
    if (heavy) {
      // ---- (H0) the slow bookkeeping of the parked lanes, up to their next memory access ----
      int bk = GB_NONE;
      if (kind == G_WAIT) bk = bk_pend;
      while (bk != GB_NONE) {
        if (bk == GB_AFTER_SEARCH) {
          KJ_P(PS_AFTER_SEARCH);
          if (nm == 0 || m_ovf) bk = GB_POP;               // (overflow: the read is redone in the retry pass)
          else if (p.mismatches > 0 && t_nmm < p.mismatches) {
            // the order in which `si_it = si_it->samelen ? si_it->samelen : si_it->next` (:477) visits the
            // list of insert_SI_sorted (bwt.c:225-252): see greedy_lane
            vi_phase = 0;
            vi_v = nm == 1 ? (int)m_ql : mq_max_below(0x7fffffff);
            bk = GB_VAR_NEXT;
          } else { ev_pass = -1; bk = GB_EVAL_NEXT; }
        }
        if (bk == GB_VAR_NEXT) {
          KJ_P(PS_VAR_NEXT);
          bool have = false;
          if (nm == 1) {
            if (vi_phase == 0) { vi_phase = 2; have = true; }
          } else if (vi_phase == 0) {
            if (vi_v >= 0) {
              // one walk over the lengths: head of the class vi_v, its size, the next shorter class
              uint32_t head = nm, cnt = 0;
              int below = -1;
              for (uint32_t x = 0; x < nm; x++) {
                const int q = mq_get(x);
                if (q == vi_v) { if (head == nm) head = x; cnt++; }
                else if (q < vi_v && q > below) below = q;
              }
              mx = head; vi_head = (int)head; have = true;
              if (cnt >= 2) { vi_phase = 1; vi_x = (int)nm; }
              else { vi_v = below; if (vi_v < 0) vi_phase = 2; }
            }
          } else if (vi_phase == 1) {
            int x = vi_x - 1;
            while (x > vi_head && mq_get((uint32_t)x) != vi_v) x--;
            if (x > vi_head) { mx = (uint32_t)x; vi_x = x; have = true; }
            else vi_phase = 2;
          }
          if (!have) { ev_pass = -1; bk = GB_EVAL_NEXT; }
          else if (nm == 1) bk = GB_VAR_MATCH;
          else { ml_for = 0; kind = G_MLOAD; bk = GB_NONE; }
        }
        if (bk == GB_VAR_MATCH) {
          KJ_P(PS_VAR_MATCH);
          const uint32_t mre = m_qi + m_ql - 1u;
          if (!(m_qi > 0 && mre + 1u >= p.m)) { bk = GB_VAR_NEXT; continue; }          // :469
          else if (!in_win((int)m_qi - 1)) {
            fill_top = (int)m_qi - 1; fill_ret = FR_VARM; fill_pref = false; kind = G_FILL; bk = GB_NONE;
          } else {
            // addAllMismatchVariantsAtPosSI(t, qi-1, erase_pos, it), :346-395
            vlen = (mre < (uint32_t)flen - 1u) ? mre + 1u : (uint32_t)flen;    // fragment.erase(erase_pos)
            vorig = ct.idx_to_aa[win[(int)m_qi - 1 - wq]];
            const int sc = (int)m_psum + t_diff;                               // calcScore(fragment, f->diff)
            const uint32_t cs = sc > 0 ? (uint32_t)sc : 0u;
            vscore = cs - (uint32_t)(int32_t)ct.b62[vorig][vorig];             // unsigned wrap as in :363
            // the substitutes are tried in descending score and the loop ends at the first one below the threshold
            // (:368-369, 390-392): if the best one is, nothing is ranked and nothing queued
            const int32_t thr0 = (int32_t)best > (int32_t)p.min_score ? (int32_t)best : (int32_t)p.min_score;
            if ((int32_t)(vscore + (uint32_t)(int32_t)ct.b62[vorig][ct.subst[vorig][0]]) < thr0) { qseq += 19; bk = GB_VAR_NEXT; continue; }
            kind = G_VMULTI; bk = GB_NONE;
          }
        }
        if (bk == GB_EVAL_NEXT) {
          KJ_P(PS_EVAL_NEXT);
          // eval_match_scores(si, t), :751-797: the samelen chains of the classes (descending length,
          // while >= m) in insertion order, then the class heads in ascending length
          if (nm == 1) {
            if (m_ql >= p.m) eval_match();
            bk = GB_POP;
          } else {
            if (ev_pass < 0) {
              ev_v1 = mq_max_below(0x7fffffff);
              if (ev_v1 < (int)p.m) bk = GB_POP;                               // :482
              else { ev_pass = 0; ev_v = ev_v1; ev_x = -1; ev_done = false; }
            }
            while (bk == GB_EVAL_NEXT) {
              if (ev_pass == 0) {
                // one walk over the lengths: the next member of the class ev_v behind its head and behind ev_x, and the
                // next shorter class
                int head = -1, nxt = -1, nv = -1;
                for (uint32_t x = 0; x < nm; x++) {
                  const int q = mq_get(x);
                  if (q == ev_v) { if (head < 0) head = (int)x; else if (nxt < 0 && (int)x > ev_x) nxt = (int)x; }
                  else if (q < ev_v && q > nv) nv = q;
                }
                if (nxt >= 0) { ev_x = nxt; mx = (uint32_t)nxt; ml_for = 1; kind = G_MLOAD; bk = GB_NONE; }
                else if (nv < 0 || nv < (int)p.m) ev_pass = 1;                 // ev_v is the shortest class >= m
                else { ev_v = nv; ev_x = -1; }
              } else if (ev_done) bk = GB_POP;
              else {
                // the head of the class ev_v and the next longer class, in one walk
                uint32_t head = nm;
                int nv = 0x7fffffff;
                for (uint32_t x = 0; x < nm; x++) {
                  const int q = mq_get(x);
                  if (q == ev_v) { if (head == nm) head = x; }
                  else if (q > ev_v && q < nv) nv = q;
                }
                mx = head; ml_for = 1; kind = G_MLOAD; bk = GB_NONE;
                if (ev_v == ev_v1) ev_done = true; else ev_v = nv;
              }
            }
          }
        }
        if (bk == GB_EVAL_MATCH) { KJ_P(PS_EVAL_MATCH); eval_match(); bk = GB_EVAL_NEXT; continue; }
        if (bk == GB_POP) {
          KJ_P(PS_POP);
          // getNextFragment(best_match_score), ConsumerThread.cpp:272-342
          uint32_t dbest = 0, dslot = 0, ext_second = 0;
          {
            const uint32_t nl = qn < (uint32_t)kGSlots ? qn : (uint32_t)kGSlots;
            if constexpr (kGSlots % 4 == 0) {
              // four priorities per LDS read (slots at and above qn hold 0: G_META clears the row)
              for (uint32_t s = 0; s < nl; s += 4) {
                const u128 v = *reinterpret_cast<const u128 *>(prio + s);
                const uint32_t e0 = (uint32_t)v.x, e1 = (uint32_t)(v.x >> 32), e2 = (uint32_t)v.y, e3 = (uint32_t)(v.y >> 32);
                if (e0 > dbest) { dbest = e0; dslot = s; }
                if (e1 > dbest) { dbest = e1; dslot = s + 1u; }
                if (e2 > dbest) { dbest = e2; dslot = s + 2u; }
                if (e3 > dbest) { dbest = e3; dslot = s + 3u; }
              }
            } else
              for (uint32_t s = 0; s < nl; s++) { const uint32_t pr = prio[s]; if (pr > dbest) { dbest = pr; dslot = s; } }
            if (ext_max > dbest) {
              // the best entry lies in the overflow area: find it, and the runner-up there (the next ext_max).  Entries
              // behind qn are stale; the loads of a pass are independent (4 * kGExtPass priorities in flight at a time)
              const uint32_t next = qn - (uint32_t)kGSlots;
              uint32_t e1 = 0, s1 = 0, e2 = 0;
              for (uint32_t base = 0; base < next; base += 4 * kGExtPass) {
                const u128 *src = reinterpret_cast<const u128 *>(GS_PRIO_EXT + base);
                u128 v[kGExtPass];
#pragma unroll
                for (int q = 0; q < kGExtPass; q++) v[q] = (base + 4u * q < next) ? src[q] : u128{0, 0};
#pragma unroll
                for (int q = 0; q < kGExtPass; q++) {
                  const uint32_t w[4] = {(uint32_t)v[q].x, (uint32_t)(v[q].x >> 32), (uint32_t)v[q].y, (uint32_t)(v[q].y >> 32)};
#pragma unroll
                  for (int z = 0; z < 4; z++) {
                    const uint32_t idx = base + 4u * q + z;
                    const uint32_t pr = idx < next ? w[z] : 0u;
                    if (pr > e1) { e2 = e1; e1 = pr; s1 = idx; } else if (pr > e2) e2 = pr;
                  }
                }
              }
              dbest = e1; dslot = (uint32_t)kGSlots + s1; ext_second = e2;
            }
          }
          const bool have_o = fo < nf, have_d = dbest != 0;
          const uint32_t dkey = dbest >> 16;
          if ((!have_o && !have_d) || ovf || m_ovf) bk = GB_FINISH;
          else {
            const uint32_t on_key = on_kl >> 16, on_len = on_kl & 0xffffu;
            const bool pick_o = have_o && (!have_d || on_key >= dkey);
            if ((pick_o ? on_key : dkey) < best) bk = GB_FINISH;
            else if (!pick_o) {
              if (dslot < (uint32_t)kGSlots) { prio[dslot] = 0; qlive--; }
              else { GS_PRIO_EXT[dslot - kGSlots] = 0; ext_max = ext_second; }
              pslot = dslot; kind = G_POPITEM; bk = GB_NONE;
            }
            else {
              t_start = on_start; t_len = on_len; t_diff = 0; t_matchlen = 0; t_tot = on_key; t_msum = 0; t_nmm = 0;
              const uint32_t oflags = on_flags;
              if (on_key >= 0xffffu || on_len >= (WIDE ? kGWideMaxFrag : 0xffffu) || (WIDE && on_start >= (1u << 24))) { ovf = true; KJ_OVF(wl, 2); }
              fo++;
              if (p.seg && !(oflags & kFragChecked)) {
                // SEG found regions in this fragment (SEG pass): the parent is dropped, its unmasked
                // pieces are queued, and the next fragment is popped (:291-334)
                const uint32_t slot = oflags >> kFragSlotShift;
                KJ_P(PS_POP_SEG);
                if (slot) {
                  const SegRec &rec = sq.recs[slot - 1];
                  if (rec.overflow) flags |= kHitInternalOverflow;
                  Frag f; f.start = t_start; f.len = t_len; f.key = on_key; f.flags = 0;
                  seg_split(ct, p, rec, b.pep + pepoff, f, [&](const Frag &q) {
                    if (q.len > (WIDE ? kGWideMaxFrag : 0xffffu) || (WIDE && q.start >= (1u << 24))) { ovf = true; KJ_OVF(wl, 3); return; }
                    const uint32_t sl = push_slot(q.key, qseq);
                    if (sl == ~0u) return;
                    qseq++;
                    u128 *dst = GS_POOL + 8 * sl;
                    u128 v;
                    if constexpr (WIDE) {
                      // (wide items: interval ends 64 bit each; start of the fragment and key move to words 2 and 3)
                      v.x = v.y = 0; dst[0] = v;
                      v.x = q.len; v.y = q.key; dst[1] = v;
                      v.x = (uint64_t)q.start << 8; v.y = 0; dst[2] = v;
                      v.x = 0; v.y = (uint64_t)(q.key << 16) << 32; dst[3] = v;
                    } else {
                    v.x = 0; v.y = q.key | (uint64_t)q.start << 32; dst[0] = v;
                    v.x = q.len; v.y = q.key; dst[1] = v;
                    v.x = v.y = 0; dst[2] = v; dst[3] = v;   // no substitutions, no window
                    }
                  });
                }
                if (fo < nf) {
                  const Frag nx = b.frags[fbase + fo];
                  on_start = nx.start; on_kl = on_pack(nx.len, nx.key); on_flags = nx.flags;
                }
                continue;                                   // bk stays GB_POP
              }
              flen = (int)t_len; nm = 0; kroll = false; skipj = false;
              j = flen - 1; tail = 0;                       // maxMatches(seq, len, seed_length, 0), bwt.c:261-296
              i = flen;                                     // (span rule: no earlier search in this fragment)
              fill_top = j; fill_ret = FR_START_J; fill_pref = true; kind = G_FILL; bk = GB_NONE;
            }
          }
        }
        if (bk == GB_FINISH) {
          KJ_P(PS_FINISH);
          nids = 0;
          KJ_G_HIT->reserved = 0;
          if (ovf || m_ovf) {
            KJ_G_HIT->best = 0;
            if (wl.retry_list) { wl.retry_list[append_slot(wl.retry_count)] = r; flags = kHitRetry; }
            else flags = kHitInternalOverflow;
            bk = GB_DONE;
          } else {
            KJ_G_HIT->best = nbest ? best : 0u;
            // the best matches go into the hit record - row | length, in the order of the list (eval_match_scores appends,
            // ids_from_SI walks from the head: ConsumerThread.cpp:751-797, :799-845) - and k_mem_locate* turn them into ids
            // with every lane at work: nbest <= max_matches_SI = 20 of the record's 21 slots
            bool big = nbest > (uint32_t)kMaxIds;
            if constexpr (WIDE) for (uint32_t q = 0; q < nbest && !big; q++) big = GS_BESTW[q].len >= kLocWideMaxLen;
            if (big) {
              // (more matches than slots - max_matches_SI raised by a caller -, or a wide interval of 2^24 rows and more: the
              //  retry pass, whose lane walks itself)
              KJ_G_HIT->best = 0;
              if (wl.retry_list) { wl.retry_list[append_slot(wl.retry_count)] = r; flags = kHitRetry; }
              else flags = kHitInternalOverflow;
            } else if (nbest) {
              if constexpr (WIDE) {
                for (uint32_t q = 0; q < nbest; q++) { const GBest2W gb = GS_BESTW[q]; KJ_G_HIT->taxid[q] = gb.lo | (uint64_t)gb.len << kLocWideShift; }
              } else {
                KJ_G_HIT->taxid[0] = (uint64_t)b0lo | (uint64_t)b0len << 32;
                for (uint32_t q = 1; q < nbest; q++) { const GBest2 gb = GS_BEST[q]; KJ_G_HIT->taxid[q] = (uint64_t)gb.lo | (uint64_t)gb.len << 32; }
              }
              nids = nbest; flags |= kHitLocPending;
              if constexpr (VERBOSE) {
                // column 7: the peptide of every best match in list order with the variant's substitutions applied (:780-790)
                vb_reset(gs.vb, r);
                for (uint32_t x = 0; x < nbest; x++) {
                  const GBestV bv = gs.bestv[(size_t)gs.lane * 64 + x];
                  vb_text(gs.vb, r, b.pep + pepoff + bv.start + bv.qi, bv.ql, bv.num_mm < (uint32_t)kMaxMismatch ? bv.num_mm : (uint32_t)kMaxMismatch,
                          bv.sub_pos, bv.sub_aa, (int)bv.qi);
                }
              }
            }
            bk = GB_DONE;
          }
        }
        if (bk == GB_DONE) {
          KJ_G_HIT->n_ids = nids; KJ_G_HIT->flags = flags;
          if constexpr (COUNT) oc[kOpcHit]++;
          kind = G_IDLE; bk = GB_NONE;
        }
      }

      // ---- (H1) hand out reads to the lanes that finished one (see mem_lane2) ----
      KJ_P(PS_HANDOUT);
      {
        const bool need = kind == G_IDLE;
        const uint64_t mask = kj_ballot(need);
        if (mask) {
          const uint32_t n = popc64(mask);
          const uint32_t rank = kj_rank_below(mask);
          const uint32_t avail = wend - wnext;
          uint32_t newbase = 0, ch = 0;
          if (n > avail) {
            const uint32_t left = n_items > wend ? n_items - wend : 0;
            ch = left / (nwaves * 4u);
            if (ch > 128u) ch = 128u;
            if (ch < 8u) ch = 8u;
            if (ch < n - avail) ch = n - avail;
            const uint32_t leader = (uint32_t)__builtin_ctzll(mask);
            uint32_t got = 0;
            if (kj_lane() == leader) got = kj_fetch_chunk(wl.counter, ch);
            newbase = kj_bcast(got, leader);
          }
          if (need) {
            const uint32_t item = rank < avail ? wnext + rank : newbase + (rank - avail);
            if (item >= n_items) kind = G_EXIT;
            else { r = wl.reads ? wl.reads[item] : item; kind = G_META; }
          }
          if (n > avail) { wnext = newbase + (n - avail); wend = newbase + ch; }
          else wnext += n;
        }
        if (kj_ballot(kind != G_EXIT) == 0) break;
      }
    }

    // ---- (1) load phase ----
    KJ_P(PS_LOAD);
    KJ_TICK(st_heavy)
#if defined(KJ_STATS) && defined(__HIP_DEVICE_COMPILE__)
    if (heavy) st_nheavy++;
#endif
    KJ_HISTO(6, kind);
    const bool is_step = kind == G_STEP, is_vm = kind == G_VMULTI;
    const bool is_kmer = kind == G_KMER || (!WIDE && kind == G_PROBE);
    const P vlo = m_lo, vhi = m_lo + m_len;
    const P posA = is_step ? lo : is_vm ? vlo : 0;
    const P posB = is_step ? hi : is_vm ? vhi : posA;
    if constexpr (COUNT) {
      oc[kOpcLaneIters] += (kind != G_EXIT && kind != G_WAIT) ? 1u : 0u;
      if (kj_lane() == 0) oc[kOpcIters]++;
      if (is_kmer) oc[kOpcKmer]++;
      else if (kind == G_STEP) { oc[kOpcStep]++; oc[kOpcStepLines] += ((posA >> 6) != (posB >> 6)) ? 2u : 1u; }
      else if (heavy) {
        if (kind == G_VMULTI) { oc[kOpcVmulti]++; oc[kOpcStepLines] += ((posA >> 6) != (posB >> 6)) ? 2u : 1u; }
        else if (kind == G_META) oc[kOpcMeta]++;
        else if (kind == G_FRAG) oc[kOpcFrag]++;
        else if (kind == G_FILL) { oc[kOpcFill]++; if (fill_pref && fo < nf) oc[kOpcFrag]++; }
        else if (kind == G_POPITEM) oc[kOpcPopItem]++;
        else if (kind == G_MLOAD) oc[kOpcMload]++;
      }
    }
    const uint32_t cc = is_step ? c : 1u;
    const RankBlock64 *pa = blk0 + (posA >> 6), *pb = blk0 + (posB >> 6);
    const u128 a01 = *reinterpret_cast<const u128 *>(&pa->plane[0]);
    const u128 a23 = *reinterpret_cast<const u128 *>(&pa->plane[2]);
    const uint64_t a4 = pa->plane[4];
    const uint32_t ca = pa->cnt[cc - 1];
    const u128 b01 = *reinterpret_cast<const u128 *>(&pb->plane[0]);
    const u128 b23 = *reinterpret_cast<const u128 *>(&pb->plane[2]);
    const uint64_t b4 = pb->plane[4];
    // (narrow G_KMER: the second block is not needed - this load fetches the presence bits of the k-mer line instead)
    const bool kline_step = !WIDE && is_kmer;
    const uint32_t *cbp = &pb->cnt[cc - 1];
    if (kline_step) cbp = reinterpret_cast<const uint32_t *>(ix.kline + (size_t)(kidx >> 6) * kKLineBytes + kKLinePresent);
    const uint32_t cb = *cbp;
    uint64_t mba = 0, mbb = 0;                             // WIDE: counts at the start of the 2^mb_shift rows
    if constexpr (WIDE) {
      mba = ix.mb_base[(size_t)((uint64_t)posA >> ix.mb_shift) * 20 + (cc - 1)];
      mbb = ix.mb_base[(size_t)((uint64_t)posB >> ix.mb_shift) * 20 + (cc - 1)];
    }
    const uint8_t *gaddr = reinterpret_cast<const uint8_t *>(blk0);
    if (is_kmer) gaddr = WIDE ? reinterpret_cast<const uint8_t *>(ix.kmer64 + kidx) : ix.kline + (size_t)kidx * 2u;
    else if (kind == G_META) gaddr = reinterpret_cast<const uint8_t *>(b.meta + r);
    else if (kind == G_FRAG) gaddr = reinterpret_cast<const uint8_t *>(b.frags + fbase);
    else if (kind == G_FILL && fill_pref && fo < nf) gaddr = reinterpret_cast<const uint8_t *>(b.frags + fbase + fo);
    else if (kind == G_MLOAD) gaddr = reinterpret_cast<const uint8_t *>(GS_MATCHES + mx);
    // (kChainPrune: the text position of a one-row match comes with the rank lines of its substitution step - its one variant
    //  lies one letter in front of it in the text)
    const bool sa_hint = kChainPrune && !WIDE && is_vm && m_len == 1u && ix.sa_full && ix.text && (uint64_t)m_lo + 4u <= ix.bwtlen;   // (the aligned 16 bytes stay inside the array)
    if (sa_hint) gaddr = reinterpret_cast<const uint8_t *>(ix.sa_full + (uint32_t)m_lo);
    const uint32_t ghalf = (uint32_t)(reinterpret_cast<uintptr_t>(gaddr) >> 3) & 1u;
    const uint32_t gword = (uint32_t)(reinterpret_cast<uintptr_t>(gaddr) >> 2) & 3u;
    const u128 gv = *reinterpret_cast<const u128_unaligned *>(kline_step ? reinterpret_cast<uintptr_t>(gaddr)
                                                                          : reinterpret_cast<uintptr_t>(gaddr) & ~(uintptr_t)15);
    // ten 16-byte reads from two lane-chosen places: a window (G_FILL), a queued item (G_POPITEM)
    u128 xa0{0, 0}, xa1{0, 0}, xa2{0, 0}, xa3{0, 0}, xa4{0, 0}, xb0{0, 0}, xb1{0, 0}, xb2{0, 0}, xb3{0, 0}, xb4{0, 0};
    int fq = 0;
    // (three wavefronts per SIMD: these reads wait until the lane consumes them, behind the fast compute, when the rank lines
    // of this iteration are dead - a second, exposed wait in heavy iterations for forty registers less at the peak)
    if (false) {
      KJ_P(PS_LOAD10);
      fq = fill_top - (kWin - 1);
      if (fq < 0) fq = 0;
      const uint8_t *sa = reinterpret_cast<const uint8_t *>(blk0), *sb = sa;
      if (kind == G_FILL) sa = b.pep + pepoff + t_start + fq;
      else if (kind == G_POPITEM) { sa = reinterpret_cast<const uint8_t *>(GS_POOL + 8 * pslot); sb = sa + 80; }
      const u128_unaligned *pa16 = reinterpret_cast<const u128_unaligned *>(sa);
      const u128_unaligned *pb16 = reinterpret_cast<const u128_unaligned *>(sb);
      xa0 = pa16[0]; xa1 = pa16[1]; xa2 = pa16[2]; xa3 = pa16[3]; xa4 = pa16[4];
      xb0 = pb16[0]; xb1 = pb16[1]; xb2 = pb16[2]; xb3 = pb16[3]; xb4 = pb16[4];
    }

    // ---- (2) compute ----
    KJ_TICK(st_load)
    int bk = GB_NONE;
    if (is_step) {
      KJ_P(PS_STEP);
      KJ_HISTO(12, t_nmm == 0 ? 0 : (uint64_t)(hi - lo) == 1 ? 1 : 2);                              // (UpdateSI steps: originals / variants on one row / on more)
      const uint64_t ia = (cc & 1u) ? 0ull : ~0ull, ib = (cc & 2u) ? 0ull : ~0ull, ic = (cc & 4u) ? 0ull : ~0ull,
                     id = (cc & 8u) ? 0ull : ~0ull, ie = (cc & 16u) ? 0ull : ~0ull;
      const uint64_t ma = (a01.x ^ ia) & (a01.y ^ ib) & (a23.x ^ ic) & (a23.y ^ id) & (a4 ^ ie);
      const P ra = (P)(mba + ca + popc64(ma & ((1ull << (posA & 63u)) - 1ull)));
      {
        // UpdateSI(str[i-1]) (bwt.c:160-173)
        const uint64_t mb = (b01.x ^ ia) & (b01.y ^ ib) & (b23.x ^ ic) & (b23.y ^ id) & (b4 ^ ie);
        const P rb = (P)(mbb + cb + popc64(mb & ((1ull << (posB & 63u)) - 1ull)));
        if (ra >= rb) bk = GB_END_MATCH;
        else {
          lo = ra; hi = rb; i--; acc += diag(c);
          if (i == 0) bk = GB_END_MATCH;
          else if (kSpanRuleStep && t_nmm == 0 && nm != 0 && hi - lo == (kSpanEq ? sz_q : (P)1) && i >= last_qi) {
            // the span rule for an interval that shrinks to one row behind the k-mer lookup: the recorded match that reaches
            // furthest (last_qi, from a larger end position) contains it, the search ends where that one ended or, unrecorded, beyond
            i = last_qi; bk = GB_END_MATCH;
          }
          else if (in_win(i - 1)) c = win[i - 1 - wq];
          else { fill_top = i - 1; fill_ret = FR_STEP; fill_pref = false; kind = G_FILL; }
        }
      }
    } else if (is_kmer) {
      KJ_P(PS_KMER);
      if constexpr (WIDE) {
        // the k-mer table of 16-byte entries {lo, len}
        lo = (P)gv.x; hi = (P)(gv.x + gv.y);
        if (lo >= hi) { i = j; bk = GB_END_MATCH; }        // seed shorter than kk: never recorded, i > 1
        else if (kSpanRule && gv.y == (kSpanEq ? (uint64_t)sz_i : 1ull) && j - (int)kk + 1 >= i) bk = GB_END_MATCH;   // inside the last match (kSpanRule, kSpanEq): i stays
        else {
          i = j - (int)kk + 1;
          if (i == 0) bk = GB_END_MATCH;
          else if (in_win(i - 1)) { c = win[i - 1 - wq]; kind = G_STEP; }
          else { fill_top = i - 1; fill_ret = FR_STEP; fill_pref = false; kind = G_FILL; }
        }
      } else {
      const uint64_t e = kline_entry(gv, kidx);
      const uint32_t l16 = (uint32_t)(e >> 32);
      uint32_t hint = 32u;                                 // the BWT letter of a one-row interval (32 = unknown)
      lo = (P)(uint32_t)e;
      if (l16 >= kKLineSingle && l16 != kKLineEscape) { hi = lo + 1u; hint = l16 & 31u; }
      else hi = lo + l16;
      // the k-mer that ends at j - 1 = w[j-kk] in front of the line's word: absent -> that end position is passed without a
      // lookup (GB_START_J)
      skipj = j >= (int)kk && in_win(j - (int)kk) && ((cb >> ((uint32_t)win[j - (int)kk - wq] - 1u)) & 1u) == 0u;
      if (kind == G_PROBE) {
        // the k-mer q-1 .. q+kk-2 behind the recording of [q, J] (kGreedyProbe): absent = on to end position q+kk-3; acc holds the
        // diagonal sum of the letters passed
        if (l16 == 0u) { tail += acc; j = (int)last_qi + (int)kk - 3; }
        skipj = false; kroll = false;
        bk = GB_START_J;
      } else
      if (l16 == kKLineEscape) {
        // an interval longer than the line's 16 bits can say: this search starts with InitialSI (bwt.c:146-152)
        c = cj;
        lo = (P)ix.C[c]; hi = (P)ix.C[c + 1];
        acc = diag(c);
        i = j;                                             // (j >= kk - 1 >= 1)
        if (in_win(i - 1)) { c = win[i - 1 - wq]; kind = G_STEP; }
        else { fill_top = i - 1; fill_ret = FR_STEP; fill_pref = false; kind = G_FILL; }
      } else if (lo >= hi) { i = j; bk = GB_END_MATCH; }   // seed shorter than kk: never recorded, i > 1
      else if (kSpanRule && hi - lo == (kSpanEq ? sz_i : (P)1) && j - (int)kk + 1 >= i) bk = GB_END_MATCH;   // inside the last match (kSpanRule, kSpanEq): i stays
      else {
        i = j - (int)kk + 1;
        if (i == 0) bk = GB_END_MATCH;
        else if (in_win(i - 1)) {
          c = win[i - 1 - wq];
          // one row whose BWT letter is not c: UpdateSI(c) finds nothing (bwt.c:160-173) - the seed is the match
          if (hint != 32u && hint != c) bk = GB_END_MATCH; else kind = G_STEP;
        }
        else { fill_top = i - 1; fill_ret = FR_STEP; fill_pref = false; kind = G_FILL; }
      }
      }
    } else if (heavy) {
      KJ_TICK(st_fast)
      uint32_t vtp = 0;                                      // kChainPrune: text position of the one variant of a one-row match
      if (sa_hint) vtp = (gword == 0u ? (uint32_t)gv.x : gword == 1u ? (uint32_t)(gv.x >> 32) : gword == 2u ? (uint32_t)gv.y : (uint32_t)(gv.y >> 32)) - 1u;
      if (is_vm) {
        KJ_P(PS_VM_RANK);
        KJ_HISTO(8, m_len == 1 ? 0 : m_len <= 4 ? 1 : 2); KJ_HISTO(9, t_nmm);   // (rows of the interval the substitutes are tried on)
        // UpdateSI(trans[substitute]) on the interval of the match for all substitutes (ConsumerThread.cpp:366-392).
        // Only letters that OCCUR in BWT[lo, hi) extend it, and the interval of a match of eleven letters or more is a row
        // or a few: when both ends lie in the same or in neighbouring rank blocks the letters are read off the two lines
        // (a symbol at a time, each distinct letter once); an interval that spans more tries all twenty.
        const uint64_t lowA = (1ull << (posA & 63u)) - 1ull, lowB = (1ull << (posB & 63u)) - 1ull;
        const uint32_t dblk = (uint32_t)((posB >> 6) - (posA >> 6));
        // the reference stops at the first substitute whose score is too low (:368-369); the
        // substitutes are sorted by score (host_tables.cpp checks), so that is a threshold
        const int32_t thr = (int32_t)best > (int32_t)p.min_score ? (int32_t)best : (int32_t)p.min_score;
        const uint32_t corig = ct.aa_to_idx[vorig];
        auto match_of = [](const u128 &p01, const u128 &p23, uint64_t p4, uint32_t cx) -> uint64_t {
          const uint64_t ia = (cx & 1u) ? 0ull : ~0ull, ib = (cx & 2u) ? 0ull : ~0ull, ic = (cx & 4u) ? 0ull : ~0ull,
                         id = (cx & 8u) ? 0ull : ~0ull, ie = (cx & 16u) ? 0ull : ~0ull;
          return (p01.x ^ ia) & (p01.y ^ ib) & (p23.x ^ ic) & (p23.y ^ id) & (p4 ^ ie);
        };
        auto symbol_at = [](const u128 &p01, const u128 &p23, uint64_t p4, uint32_t t) -> uint32_t {
          return (uint32_t)((p01.x >> t) & 1ull) | (uint32_t)((p01.y >> t) & 1ull) << 1 | (uint32_t)((p23.x >> t) & 1ull) << 2 |
                 (uint32_t)((p23.y >> t) & 1ull) << 3 | (uint32_t)((p4 >> t) & 1ull) << 4;
        };
        uint32_t todo = 0x1ffffeu;                          // letters 1..20
        if (dblk <= 1u) {
          todo = 0;
          uint64_t ma = dblk == 0 ? (lowB & ~lowA) : ~lowA;  // rows of the interval in the first line
          while (ma) {
            const uint32_t cx = symbol_at(a01, a23, a4, (uint32_t)__builtin_ctzll(ma));
            ma &= ~match_of(a01, a23, a4, cx);
            todo |= 1u << cx;
          }
          uint64_t mb = dblk == 0 ? 0ull : lowB;             // ... and in the second
          while (mb) {
            const uint32_t cx = symbol_at(b01, b23, b4, (uint32_t)__builtin_ctzll(mb));
            mb &= ~match_of(b01, b23, b4, cx);
            todo |= 1u << cx;
          }
          todo &= 0x1ffffeu;                                 // (terminators and the padding behind the last row are no letters)
        }
        todo &= ~(1u << corig);
        uint32_t q0 = sp0, q1 = sp1, q2 = sp2, q3 = sp3;
        const uint32_t pz = m_qi - 1u;
        if (t_nmm < (uint32_t)kMaxMismatch) {
          const uint32_t hs = (t_nmm & 1u) * 16u;
          const uint32_t hm = ~(0xffffu << hs), pzz = (pz & 0xffffu) << hs;
          switch (t_nmm >> 1) {
            case 0: q0 = (q0 & hm) | pzz; break;
            case 1: q1 = (q1 & hm) | pzz; break;
            case 2: q2 = (q2 & hm) | pzz; break;
            default: q3 = (q3 & hm) | pzz; break;
          }
        }
        const int boo = (int)ct.b62[vorig][vorig];
        // the variants share everything but the letter: window of the fragment with the new letter
        const int need_fq = (int)m_qi - 2 - (kWin - 1) > 0 ? (int)m_qi - 2 - (kWin - 1) : 0;
        const bool win_ok = need_fq == wq;                  // the variant resumes at m_qi-2 (if m_qi > 1)
        while (todo) {
          const uint32_t cx = (uint32_t)__builtin_ctz(todo);
          todo &= todo - 1u;
          const int bos = (int)ct.b62_idx[vorig][cx - 1u];
          const uint32_t key = (uint32_t)(int32_t)(vscore + (uint32_t)(int32_t)bos);
          if ((int32_t)key < thr) continue;
          // the letter's counts in front of the two lines: read now that the letter is known (the lines have just been
          // fetched, so this is a cache hit - cheaper than holding all forty counts in registers for the few that are used)
          P ra = (P)(pa->cnt[cx - 1u] + popc64(match_of(a01, a23, a4, cx) & lowA));
          P rb = (P)(pb->cnt[cx - 1u] + popc64(match_of(b01, b23, b4, cx) & lowB));
          if constexpr (WIDE) {
            ra += (P)ix.mb_base[(size_t)((uint64_t)posA >> ix.mb_shift) * 20 + (cx - 1u)];
            rb += (P)ix.mb_base[(size_t)((uint64_t)posB >> ix.mb_shift) * 20 + (cx - 1u)];
          }
          if (ra >= rb) continue;
          KJ_P(PS_VM_PUSH);
          if (vlen > 0xffffu || m_ql + 1u > 0xffffu) { ovf = true; KJ_OVF(wl, 4); break; }
          const int bss = (int)diag(cx);
          if constexpr (kChainPrune && WIDE && kWideChainPrune) {
            // the same test on an index with 64-bit rows that keeps the text and the text position of EVERY row (tv_shift 0: up
            // to 2^34 rows where HBM has room); the position is read here - the load phase has no slot for it
            const int sc0 = (int)m_dsum + t_diff + bos;
            if (rb - ra <= (P)kChainRows && ix.sa_tpos5 && ix.tv_shift == 0u && ix.text && (m_ql + 1u < p.m || sc0 < thr)) {
              bool hopeless = true;
              for (P q = ra; q < rb && hopeless; q++) {              // (every row's chain, as in the narrow lane below)
                const uint64_t tp = *reinterpret_cast<const u64_unaligned *>(ix.sa_tpos5 + (size_t)q * 5u) & kTposNone;
                if constexpr (COUNT) oc[kOpcSa] += 2u;
                hopeless = tp != kTposNone && tp >= 16u + kTextPad &&
                           kj_chain_hopeless(p, win, wq, (int)pz, (int)m_qi + (int)m_ql - 1, *reinterpret_cast<const u128_unaligned *>(ix.text + tp - 16u), t_nmm + 1u, sc0, thr);
              }
              if (hopeless) {
                if constexpr (COUNT) oc[kOpcPruned]++;
                continue;
              }
            }
          }
          if constexpr (kChainPrune && !WIDE) {
            // A variant on ONE database row is the root of a CHAIN: UpdateSI on a one-row interval succeeds iff the letter in
            // front of the row's suffix is the letter asked for (bwt.c:160-173), the one substitute that exists at the next
            // mismatch is the text's letter there (ConsumerThread.cpp:366-392), and the row it leads to is one row again - so
            // everything the reference will do with this item and its descendants is an ungapped comparison of the fragment with
            // the database text in front of that suffix, with up to `mismatches` substitutions.  If NO item of the chain can
            // ever hold a match of m letters, none reaches eval_match_scores (:482: `if (m_ql >= min_fragment_length)`), and the
            // chain changes nothing: best, the list of best matches and the flags stay as they are, and the sequence numbers it
            // would have used keep the order of all other items (they only break ties, and they grow with time whether or not
            // some are skipped).  Such an item is not queued (round 6; kj_chain_hopeless).  19.6 of the 24 variant items a
            // benchmark read pops hang on one row, and 15 of them are of this kind: seeds of seven or eight letters in a wrong
            // frame that three substitutions cannot turn into a match of eleven.
            // (an item that passes the gate itself - m letters, min_score - is not looked at: on a database of protein families most
            //  one-row variants are of that kind, and the look costs a dependent load)
            const int sc0 = (int)m_dsum + t_diff + bos;           // the variant's own score (eval_match_scores: m_dsum + diff)
            // (round 6, later: a variant on a FEW rows - kChainRows; on the benchmark index 3.4 of the 5.5 items a read still
            //  queued hung on two rows, a protein and its mutated copy - is not queued if the chain of EVERY one of its rows is
            //  hopeless: whatever item descends from it ends on a non-empty subset of those rows, and for any row r of that
            //  subset the path that led there is a path of r's own chain - the fragment's letter where r's text has it, a
            //  substitution by r's letter where it has not -, with the same substitutions and the same score)
            const P nrows = rb - ra;
            if (nrows <= (P)kChainRows && ix.sa_full && ix.text && (m_ql + 1u < p.m || sc0 < thr)) {
              bool hopeless = true;
              for (P q = ra; q < rb && hopeless; q++) {
                // (the text position of the variant's match: one letter in front of its parent's when that had one row - that
                //  row's position came with the rank lines -, else the entry of the row)
#ifdef KJ_CHAIN_PRUNE_HINT_ONLY
                const uint32_t tp = nrows == (P)1 ? vtp : 0u;
#else
                const uint32_t tp = (nrows == (P)1 && vtp) ? vtp : ix.sa_full[(uint32_t)q];
#endif
                if constexpr (COUNT) oc[kOpcSa] += (nrows == (P)1 && vtp) ? 1u : 2u;    // (a line of text, and the row's entry of the full suffix array unless it came with the rank lines)
                hopeless = tp >= 16u + kTextPad &&
                           kj_chain_hopeless(p, win, wq, (int)pz, (int)m_qi + (int)m_ql - 1, *reinterpret_cast<const u128_unaligned *>(ix.text + tp - 16u), t_nmm + 1u, sc0, thr);
              }
              if (hopeless) {
                if constexpr (COUNT) oc[kOpcPruned]++;
                continue;
              }
            }
          }
          const uint32_t sl = push_slot(key, qseq + ct.subst_rank[vorig][cx - 1u]);
          if (sl == ~0u) break;
          uint32_t e0 = sa0, e1 = sa1;
          if (t_nmm < (uint32_t)kMaxMismatch) {
            const uint32_t bs = (t_nmm & 3u) * 8u, bm = ~(0xffu << bs);
            if (t_nmm < 4u) e0 = (e0 & bm) | cx << bs; else e1 = (e1 & bm) | cx << bs;
          }
          u128 *dst = GS_POOL + 8 * sl;
          u128 v;
          if constexpr (WIDE) { v.x = ra; v.y = rb; }
          else { v.x = ra | (uint64_t)rb << 32; v.y = key | (uint64_t)t_start << 32; }
          dst[0] = v;
          v.x = (vlen | (m_ql + 1u) << 16) | (uint64_t)(uint32_t)(t_diff + bos - bss) << 32;
          v.y = (m_psum - (uint32_t)boo + (uint32_t)bss) | (uint64_t)(m_dsum + (uint32_t)bss) << 32; dst[1] = v;
          // (wide: the fragment's start rides in word 2 next to the number of substitutions, the key in word 3 next to the tag)
          v.x = ((t_nmm + 1u) | (WIDE ? t_start << 8 : 0u)) | (uint64_t)q0 << 32; v.y = q1 | (uint64_t)q2 << 32; dst[2] = v;
          v.x = q3 | (uint64_t)e0 << 32;
          v.y = e1 | (uint64_t)((win_ok ? (uint32_t)wq + 1u : 0u) | (WIDE ? key << 16 : 0u)) << 32; dst[3] = v;
          if (win_ok) {
            const uint32_t *w32 = reinterpret_cast<const uint32_t *>(win);
            u128 wv;
            wv.x = w32[0] | (uint64_t)w32[1] << 32; wv.y = w32[2] | (uint64_t)w32[3] << 32; dst[4] = wv;
            wv.x = w32[4] | (uint64_t)w32[5] << 32; wv.y = w32[6] | (uint64_t)w32[7] << 32; dst[5] = wv;
            wv.x = w32[8] | (uint64_t)w32[9] << 32; wv.y = w32[10] | (uint64_t)w32[11] << 32; dst[6] = wv;
            wv.x = w32[12] | (uint64_t)w32[13] << 32; wv.y = w32[14] | (uint64_t)w32[15] << 32; dst[7] = wv;
            reinterpret_cast<uint8_t *>(dst + 4)[(int)pz - wq] = (uint8_t)cx;   // pz is in the window (GB_VAR_MATCH)
          }
        }
        qseq += 19;
        bk = GB_VAR_NEXT;
      } else if (kind == G_META) {
        KJ_P(PS_META);
        pepoff = gv.x;
        fbase = (uint32_t)gv.y;
        nf = (uint32_t)(gv.y >> 32) & ~kNfragSegPending;
        fo = 0; best = 0; nbest = 0; flags = 0; ovf = false; m_ovf = false;
        if constexpr (kGSlots % 4 == 0) {
          const u128 z{0, 0};
          for (uint32_t s = 0; s < (uint32_t)kGSlots; s += 4) *reinterpret_cast<u128 *>(prio + s) = z;
        } else
          for (uint32_t s = 0; s < (uint32_t)kGSlots; s++) prio[s] = 0;
        qn = qlive = qseq = 0; ext_max = 0;                  // (the overflow area is append-only within a read: nothing to clear)
        if (nf == 0) bk = GB_FINISH; else kind = G_FRAG;
      } else if (kind == G_FRAG) {
        KJ_P(PS_FRAG);
        on_start = (uint32_t)gv.x; on_kl = on_pack((uint32_t)(gv.x >> 32), (uint32_t)gv.y); on_flags = (uint32_t)(gv.y >> 32);
        bk = GB_POP;
      } else if (kind == G_FILL || kind == G_POPITEM) {
        KJ_P(PS_FILL);
        bool fill = kind == G_FILL;
        fq = fill_top - (kWin - 1);
        if (fq < 0) fq = 0;
        {
          const uint8_t *src = fill ? b.pep + pepoff + t_start + fq : reinterpret_cast<const uint8_t *>(GS_POOL + 8 * pslot);
          const u128_unaligned *s16 = reinterpret_cast<const u128_unaligned *>(src);
          xa0 = s16[0]; xa1 = s16[1]; xa2 = s16[2]; xa3 = s16[3];
        }
        u128 f0 = xa0, f1 = xa1, f2 = xa2, f3 = xa3;
        int newq = fq;
        if (!fill) {
          if constexpr (WIDE) { lo = (P)xa0.x; hi = (P)xa0.y; t_start = ((uint32_t)xa2.x >> 8) & 0xffffffu; }
          else { lo = (P)xa0.x; hi = (P)(xa0.x >> 32); t_start = (uint32_t)(xa0.y >> 32); }
          t_len = (uint32_t)xa1.x & 0xffffu; t_matchlen = ((uint32_t)xa1.x >> 16) & 0xffffu;
          t_diff = (int32_t)(uint32_t)(xa1.x >> 32);
          t_tot = (uint32_t)xa1.y; t_msum = (uint32_t)(xa1.y >> 32);
          t_nmm = WIDE ? ((uint32_t)xa2.x & 0xffu) : (uint32_t)xa2.x;
          sp0 = (uint32_t)(xa2.x >> 32); sp1 = (uint32_t)xa2.y; sp2 = (uint32_t)(xa2.y >> 32); sp3 = (uint32_t)xa3.x;
          sa0 = (uint32_t)(xa3.x >> 32); sa1 = (uint32_t)xa3.y;
          const uint32_t wtag = WIDE ? ((uint32_t)(xa3.y >> 32) & 0xffffu) : (uint32_t)(xa3.y >> 32);
          flen = (int)t_len; nm = 0; kroll = false; skipj = false;
          j = flen - 1;
          if (t_nmm != 0) { KJ_HISTO(10, (uint64_t)(hi - lo) == 1 ? 0 : (uint64_t)(hi - lo) <= 4 ? 1 : 2); KJ_HISTO(11, t_nmm); }   // (popped variants)
          if (t_nmm == 0) {
            // a SEG piece: maxMatches like an original
            tail = 0;
            i = flen;
            fill_top = j; fill_ret = FR_START_J; fill_pref = false; kind = G_FILL;
          } else {
            // maxMatches_withStart, bwt.c:298-336
            i = j - (int)t_matchlen + 1;
            acc = t_msum;
            if (i <= 0) bk = GB_END_MATCH;
            else if (wtag != 0) {                           // the item carries its window
              fill = true; fill_ret = FR_STEP; fill_pref = false;
              { const u128 *w16 = GS_POOL + 8 * pslot + 4; xa4 = w16[0]; xb0 = w16[1]; xb1 = w16[2]; xb2 = w16[3]; }
              f0 = xa4; f1 = xb0; f2 = xb1; f3 = xb2; newq = (int)wtag - 1;
            } else { fill_top = i - 1; fill_ret = FR_STEP; fill_pref = false; kind = G_FILL; }
          }
        }
        if (fill) {
          wq = newq;
          uint32_t *d32 = reinterpret_cast<uint32_t *>(win);
          d32[0] = (uint32_t)f0.x; d32[1] = (uint32_t)(f0.x >> 32); d32[2] = (uint32_t)f0.y; d32[3] = (uint32_t)(f0.y >> 32);
          d32[4] = (uint32_t)f1.x; d32[5] = (uint32_t)(f1.x >> 32); d32[6] = (uint32_t)f1.y; d32[7] = (uint32_t)(f1.y >> 32);
          d32[8] = (uint32_t)f2.x; d32[9] = (uint32_t)(f2.x >> 32); d32[10] = (uint32_t)f2.y; d32[11] = (uint32_t)(f2.y >> 32);
          d32[12] = (uint32_t)f3.x; d32[13] = (uint32_t)(f3.x >> 32); d32[14] = (uint32_t)f3.y; d32[15] = (uint32_t)(f3.y >> 32);
          if (kind == G_FILL) {
            // the substitutions of the variant (the reference edits the fragment string, :380)
            for (uint32_t x = 0; x < t_nmm && x < (uint32_t)kMaxMismatch; x++) {
              const uint32_t pw = x < 2 ? sp0 : x < 4 ? sp1 : x < 6 ? sp2 : sp3;
              const int pz = (int)((pw >> ((x & 1u) * 16u)) & 0xffffu);
              const uint32_t aw = x < 4 ? sa0 : sa1;
              if (pz >= wq && pz < wq + kWin && pz < (int)t_len) win[pz - wq] = (uint8_t)(aw >> ((x & 3u) * 8u));
            }
            if (fill_pref && fo < nf) {
              on_start = (uint32_t)gv.x; on_kl = on_pack((uint32_t)(gv.x >> 32), (uint32_t)gv.y); on_flags = (uint32_t)(gv.y >> 32);
            }
          }
          if (fill_ret == FR_STEP) { c = win[i - 1 - wq]; kind = G_STEP; }
          else if (fill_ret == FR_START_J) bk = GB_START_J;
          else bk = GB_VAR_MATCH;
        }
      } else if (kind == G_MLOAD) {
        KJ_P(PS_MLOAD);
        if constexpr (WIDE) {
          m_lo = (P)(gv.x & ((1ull << 40) - 1ull)); m_qi = (uint32_t)(gv.x >> 40) & 0xfffu; m_ql = (uint32_t)(gv.x >> 52) & 0xfffu;
          m_len = (uint32_t)gv.y; m_dsum = (uint32_t)(gv.y >> 32) & 0xffffu; m_psum = (uint32_t)(gv.y >> 48) & 0xffffu;
        } else {
          m_lo = (uint32_t)gv.x; m_len = (uint32_t)(gv.x >> 32);
          m_qi = (uint32_t)gv.y & 0xffffu; m_ql = ((uint32_t)gv.y >> 16) & 0xffffu;
          m_dsum = (uint32_t)(gv.y >> 32) & 0xffffu; m_psum = (uint32_t)(gv.y >> 48) & 0xffffu;
        }
        bk = ml_for == 0 ? GB_VAR_MATCH : GB_EVAL_MATCH;
      }
    }

    // ---- (3) fast bookkeeping; everything else waits for the next heavy iteration ----
    if (heavy) { KJ_TICK(st_slow) } else { KJ_TICK(st_fast) }
    KJ_P(PS_TAIL);
    bool probe_now = false;
    while (bk != GB_NONE) {
      if (bk == GB_END_MATCH) {
        KJ_P(PS_END_MATCH);
        const int l = j - i + 1;
        if (t_nmm == 0) {
          if constexpr (kSpanEq) sz_i = hi > lo ? (P)(hi - lo) : (P)1;
          bool recorded = false;
          if (l >= (int)p.seed_length && (nm == 0 || i < last_qi)) {        // bwt.c:276-278
            recorded = true;
            if (nm < (uint32_t)kGMaxMAll && (!WIDE || (uint64_t)(hi - lo) <= 0xffffffffull)) {
              m_lo = lo; m_len = (uint32_t)(hi - lo); m_qi = (uint32_t)i; m_ql = (uint32_t)l; m_dsum = acc; m_psum = t_tot - tail;
              if constexpr (WIDE) {
                u128 mm;
                mm.x = (uint64_t)m_lo | (uint64_t)m_qi << 40 | (uint64_t)m_ql << 52;
                mm.y = m_len | (uint64_t)m_dsum << 32 | (uint64_t)m_psum << 48;
                reinterpret_cast<u128 *>(GS_MATCHES)[nm] = mm;
              } else {
              GMatch2 mm; mm.lo = (uint32_t)m_lo; mm.len = m_len; mm.qiql = m_qi | m_ql << 16; mm.dp = m_dsum | m_psum << 16;
              GS_MATCHES[nm] = mm;
              }
              if (nm < (uint32_t)kGMaxM) mq[nm] = (uint16_t)l; else GS_MQ_EXT[nm - kGMaxM] = (uint16_t)l;
              if constexpr (COUNT) oc[kOpcMatchWr]++;
            } else { if (!m_ovf) KJ_OVF(wl, 5); m_ovf = true; }
            nm++;
            last_qi = i;
            if constexpr (kSpanEq) sz_q = (P)(hi - lo);
          }
          if (i <= 1) bk = GB_AFTER_SEARCH;                                 // bwt.c:292
          else {
            tail += diag(cj); j--; bk = GB_START_J;
            // kGreedyProbe: the k-mer i-1 .. i+kk-2 is looked up before the search from j (GB_START_J builds the lookup)
            probe_now = !WIDE && kGreedyProbe && recorded && kk && l > (int)kk && in_win(i - 1) && in_win(i + (int)kk - 2);
          }
        } else {
          // :443-449: after the last allowed mismatch the match must reach min_fragment_length
          const int Lreq = (t_nmm == p.mismatches) ? (int)p.m : (int)t_matchlen;
          if (l >= Lreq) {
            if (WIDE && (uint64_t)(hi - lo) > 0xffffffffull) { m_ovf = true; KJ_OVF(wl, 6); }
            m_lo = lo; m_len = (uint32_t)(hi - lo); m_qi = (uint32_t)i; m_ql = (uint32_t)l; m_dsum = acc; m_psum = t_tot;
            nm = 1;
          }
          bk = GB_AFTER_SEARCH;
        }
      }
      if (bk == GB_START_J) {
        KJ_P(PS_START_J);
        if (probe_now) skipj = false;                      // (the probe's answer may move j: what the last line said about j is dropped)
        if (skipj && j >= (int)p.seed_length - 1 && in_win(j) && in_win(j - (int)kk + 1)) {
          // the k-mer that ends here is not in the index (the line of end position j + 1 said so): as for an empty entry - i = j,
          // nothing recorded (GB_END_MATCH: l = 1 < seed_length), `if (i <= 1) break` (bwt.c:292), on to j - 1 in the same pass;
          // the line number and the diagonal sum roll over this end position too (kroll holds: the lookup at j + 1 set it)
          skipj = false;
          if (j <= 1) bk = GB_AFTER_SEARCH;
          else {
            const uint32_t cn = win[j - (int)kk + 1 - wq], c1 = win[j - wq];
            kidx = (((kidx >> 6) - (c1 - 1u) * kpow) * 20u + (cn - 1u)) << 6;
            kacc = kacc - diag(cj) + diag(cn);
            cj = c1;
            tail += diag(c1); j--;
          }
        }
        if (bk != GB_START_J) {}
        else if (j < (int)p.seed_length - 1) { skipj = false; bk = GB_AFTER_SEARCH; }
        else if (!probe_now && (!in_win(j) || (kk && j >= (int)kk - 1 && !in_win(j - (int)kk + 1)))) {
          fill_top = j; fill_ret = FR_START_J; fill_pref = false; kind = G_FILL; bk = GB_NONE;     // (skipj, if set, waits)
        } else if (WIDE && kk && j >= (int)kk - 1) {
          // wide: index of the word w[j-kk+1 .. j] in the k-mer table (w[j] = most significant digit), rolled from that of j + 1
          if (kroll) {
            const uint32_t cn = win[j - (int)kk + 1 - wq];
            kidx = (kidx - (cj - 1u) * kpow) * 20u + (cn - 1u);
            kacc = kacc - diag(cj) + diag(cn);
          } else {
            kidx = 0; kacc = 0;
            for (uint32_t q = 0; q < kk; q++) {
              const uint32_t cq = win[j - (int)q - wq];
              kidx = kmer_index(kidx, cq);
              kacc += diag(cq);
            }
          }
          cj = win[j - wq]; acc = kacc; kroll = true;
          kind = G_KMER; bk = GB_NONE;
        } else if (kk && j >= (int)kk - 1) {
          uint32_t kcode;
          const int ej = probe_now ? i + (int)kk - 2 : j;   // end position of the k-mer looked up
          if (kroll && !probe_now) {
            // from end position j + 1 (letter cj, line kidx >> 6 = w[j-kk+2 .. j]) to j: w[j] leaves the line's word at its
            // most significant digit, w[j-kk+1] enters at the least significant one
            const uint32_t cn = win[j - (int)kk + 1 - wq], c1 = win[j - wq];
            kcode = ((kidx >> 6) - (c1 - 1u) * kpow) * 20u + (cn - 1u);
            kacc = kacc - diag(cj) + diag(cn);
          } else {
            kcode = 0; kacc = diag(win[ej - wq]);
            for (uint32_t q = 1; q < kk; q++) {
              const uint32_t cq = win[ej - (int)q - wq];
              kcode = kline_code(kcode, cq);
              kacc += diag(cq);
            }
          }
          if (probe_now) {
            // what the end positions ej .. j add to `tail` if they are passed = the recorded match's diagonal sum (acc) less its
            // last letter (cj) and its first kk-2 (the k-mer's less its two outer letters); acc is free until the next search
            const uint32_t ce = win[ej - wq];
            acc = acc - diag(cj) - (kacc - diag(ce) - diag(win[i - 1 - wq]));
            kroll = false;
            kidx = kline_ref(kcode, ce);
            kind = G_PROBE; bk = GB_NONE;
          } else {
          cj = win[j - wq]; acc = kacc; kroll = true;
          kidx = kline_ref(kcode, cj);
          kind = G_KMER; bk = GB_NONE;
          }
        } else {
          c = cj = win[j - wq]; kroll = false;
          lo = (P)ix.C[c]; hi = (P)ix.C[c + 1];                              // InitialSI, bwt.c:146-152
          acc = diag(c);
          i = j;
          if (i == 0) { bk = GB_END_MATCH; continue; }
          else if (in_win(i - 1)) { c = win[i - 1 - wq]; kind = G_STEP; bk = GB_NONE; }
          else { fill_top = i - 1; fill_ret = FR_STEP; fill_pref = false; kind = G_FILL; bk = GB_NONE; }
        }
      }
      if (bk > GB_START_J) { bk_pend = bk; kind = G_WAIT; bk = GB_NONE; }
    }
  }
  if constexpr (COUNT) opc_flush(opc_of(wl), oc);
#if defined(KJ_PROF) && defined(__HIP_DEVICE_COMPILE__)
  KJ_P(PS_HEAD);
  if ((threadIdx.x & 63u) == 0) {
    unsigned long long *dst = reinterpret_cast<unsigned long long *>(reinterpret_cast<uint8_t *>(wl.counter) + 1024);
    for (int x = 0; x < 3 * PS_N; x++) atomicAdd(dst + x, gs.prof[2 + x]);
  }
#endif
#if defined(KJ_STATS) && defined(__HIP_DEVICE_COMPILE__)
  if ((threadIdx.x & 63u) == 0) {
    // experiment only: cycles per section of the lane loop, summed over the wavefronts
    unsigned long long *acc = reinterpret_cast<unsigned long long *>((reinterpret_cast<uintptr_t>(wl.counter) & ~(uintptr_t)63) + 32);
    atomicAdd(acc + 0, st_heavy); atomicAdd(acc + 1, st_load); atomicAdd(acc + 2, st_fast);
    atomicAdd(acc + 3, st_slow); atomicAdd(acc + 4, st_book); atomicAdd(acc + 5, (unsigned long long)itc << 32 | st_nheavy);
  }
#endif
}

// ----------------------------------------------------------------------------
// LCA on the device (lca_from_ids, util.cpp:194-263): the nodes.dmp tree as an open-addressing
// hash table taxon id -> slot, per slot the parent's id and slot and the depth as the reference
// computes it (1 + number of parent steps to a node that is its own parent or unknown).
// ----------------------------------------------------------------------------
struct DevTaxonomy {
  const uint64_t *key;       // [cap] taxon id of the slot, ~0 = empty
  const uint64_t *parent_id; // [cap]
  const uint32_t *parent_slot; // [cap] slot of the parent, ~0 if the parent is not in the tree
  const uint32_t *depth;     // [cap]
  uint32_t cap_mask;         // cap - 1 (cap is a power of two)
};
struct CompactHit {          // == kaiju_gpu_compact (16 bytes)
  uint64_t lca;              // LCA of the ids of the hit, 0: no hit / none of the ids is in the tree
  uint32_t best;
  uint32_t info;             // flags << 8 | n_ids
};
KJ_HD uint32_t tax_hash(uint64_t id) {
  id ^= id >> 33; id *= 0xff51afd7ed558ccdULL; id ^= id >> 33; id *= 0xc4ceb9fe1a85ec53ULL; id ^= id >> 33;
  return (uint32_t)id;
}
KJ_HD uint32_t tax_find(const DevTaxonomy &t, uint64_t id) {
  uint32_t s = tax_hash(id) & t.cap_mask;
  for (;;) {
    const uint64_t k = t.key[s];
    if (k == id) return s;
    if (k == ~0ull) return ~0u;
    s = (s + 1) & t.cap_mask;
  }
}
KJ_HD uint64_t tax_lca(const DevTaxonomy &t, const uint64_t *ids, uint32_t n) {
  if (n == 0) return 0;
  if (n == 1) return ids[0];                                  // util.cpp:197-199 (no lookup at all)
  uint64_t leaf[kMaxIds];
  uint32_t slot[kMaxIds], dep[kMaxIds];
  uint32_t m = 0, shallowest = 0xffffffffu;
  for (uint32_t i = 0; i < n && i < (uint32_t)kMaxIds; i++) {
    const uint32_t s = tax_find(t, ids[i]);
    if (s == ~0u) continue;                                   // not in the tree: dropped (:206-210)
    leaf[m] = ids[i]; slot[m] = s; dep[m] = t.depth[s];
    if (dep[m] < shallowest) shallowest = dep[m];
    m++;
  }
  if (m == 0) return 0;
  // a node outside the tree is its own parent (it is never looked up again)
  for (uint32_t i = 0; i < m; i++)
    for (uint32_t d = dep[i]; d > shallowest; d--)
      if (slot[i] != ~0u) { leaf[i] = t.parent_id[slot[i]]; slot[i] = t.parent_slot[slot[i]]; }
  // lock-step climb (:245-262).  After `shallowest` + 1 steps every path has reached its end, so
  // paths that have not met by then never will (the reference would loop forever; 0 here)
  for (uint32_t step = 0; step <= shallowest + 1; step++) {
    const uint64_t first = leaf[0];
    bool same = true;
    for (uint32_t i = 0; i < m; i++) {
      if (leaf[i] != first) same = false;
      if (slot[i] != ~0u) { leaf[i] = t.parent_id[slot[i]]; slot[i] = t.parent_slot[slot[i]]; }
    }
    if (same) return first;
  }
  return 0;
}
KJ_HD CompactHit compact_hit(const DevTaxonomy &t, const Hit &h) {
  CompactHit c;
  c.best = h.best; c.info = h.flags << 8 | (h.flags & kHitInternalOverflow) | (h.n_ids & 255u);
  c.lca = (h.n_ids == 0 || h.best == 0) ? 0 : tax_lca(t, h.taxid, h.n_ids);
  return c;
}

}  // namespace kj
