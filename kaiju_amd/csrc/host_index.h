// host_index.h — .fmi parsing and re-packing of the FM-index into the HBM layout.
//
// Reads the reference's on-disk index format byte for byte (bwt/bwt.c:51-88,
// bwt/suffixArray.c:282-321, bwt/fmicommon.h:190-217, bwt/compactfmi.c:165-171) or takes
// the same arrays from memory (kaiju_gpu_host_index), decodes the byte-coded BWT to plain
// letters and builds the rank blocks / superblocks / SA sample / taxon tables the kernels
// use (kj_core.h).  Pure host code; the arrays are uploaded by capi.hip.
#pragma once
#include <stdint.h>
#include <string>
#include <vector>

#include "kj_core.h"
#include "fmi_stream.h"

namespace kj {

// Allocator of the big arrays (hundreds of MB each): elements are default-initialised - resize() does not fill, so the
// pages of an array are first touched by the threads that write them, not by one thread zeroing it - and large blocks
// are 2 MB aligned with transparent huge pages asked for (one page fault per 2 MB instead of per 4 KB).
void *big_alloc(size_t bytes);
void big_free(void *p, size_t bytes);
template <class T> struct BigAlloc {
  typedef T value_type;
  BigAlloc() = default;
  template <class U> BigAlloc(const BigAlloc<U> &) {}
  T *allocate(size_t n) { return static_cast<T *>(big_alloc(n * sizeof(T))); }
  void deallocate(T *p, size_t n) { big_free(p, n * sizeof(T)); }
  template <class U> void construct(U *p) { ::new (static_cast<void *>(p)) U; }                      // default-, not value-initialised
  template <class U, class... A> void construct(U *p, A &&...a) { ::new (static_cast<void *>(p)) U(static_cast<A &&>(a)...); }
  template <class U> bool operator==(const BigAlloc<U> &) const { return true; }
  template <class U> bool operator!=(const BigAlloc<U> &) const { return false; }
};
template <class T> using BigVec = std::vector<T, BigAlloc<T>>;

struct HostIndexView {        // borrowed pointers (== kaiju_gpu_host_index)
  int64_t bwtlen = 0;
  int32_t nseq = 0, alen = 0;
  const char *alphabet = nullptr;
  const uint8_t *bwt = nullptr;
  const int32_t *startLcode = nullptr;
  const uint8_t *sa = nullptr;
  int64_t ncheck = 0;
  int32_t chpt_exp = 0, nbytes = 0, pbits = 0;
  const char *const *ids = nullptr;
};

// owns everything read from a .fmi file
struct FmiFile {
  int64_t len = 0;
  int32_t nseq = 0, alen = 0;
  std::string alphabet;
  int64_t salen = 0, ncheck = 0;
  int32_t chpt_exp = 0, nbytes = 0, sbits = 0, pbits = 0;
  int64_t mask = 0, check = 0;
  int32_t sa_nseq = 0;
  std::vector<std::string> ids;
  std::vector<const char *> id_ptrs;
  BigVec<uint8_t> sa;
  int32_t f_alen = 0;
  int64_t bwtlen = 0;
  int32_t N1 = 0, N2 = 0;
  BigVec<uint8_t> bwt;
  std::vector<int64_t> index1_last;   // index1[N1-1][*] = C[] as stored by the reference
  std::vector<int32_t> startLcode;
  // lazy load: where sa[] and bwt[] lie in the file (they are not read: fmi_stream.h streams them to the device)
  uint64_t sa_off = 0, bwt_off = 0;
  // returns 0 or a negative kaiju_gpu_status; msg receives details.  lazy: the two big arrays stay in the file
  int load(const char *path, std::string &msg, bool lazy = false);
  HostIndexView view() const;
};

// an array of an image file that was NOT read into host memory: element count and where its elements start in the file
// (capi.hip streams such arrays from the file to the device in pieces through page-locked buffers)
struct LazyArr { uint64_t off = 0, n = 0; };
struct ImageLazy {
  std::string path;            // empty: nothing is lazy
  LazyArr blocks64, sa_iseq, sa_pos, term_pos, kmer32, kmer64;
  LazyArr sa_taxid;            // (streamed .fmi only: an image's taxon ids of the samples are read into host memory)
};

// the packed index in host memory, ready for upload
struct PackedIndex {
  ImageLazy lazy;                   // read_image(.., lazy_big = true): the arrays that grow with the index stay in the file
  template <class V> static uint64_t count(const V &v, const LazyArr &l) { return v.empty() ? l.n : (uint64_t)v.size(); }
  BigVec<RankBlock64> blocks64;   // second-generation lanes: absolute counts (bwtlen < 2^32) or relative to mb_base
  std::vector<uint64_t> mb_base;       // wide layout: [nmb][20], counts at the start of every 2^mb_shift rows
  uint32_t mb_shift = 0;
  bool wide = false;                   // 64-bit positions (bwtlen >= 2^32, or forced for tests)
  BigVec<uint64_t> sa_taxid;      // taxon id per sampled SA row (~0: unusable name)
  uint64_t src_fmi_bytes = 0;        // size of the .fmi this was packed from (0: not from a file); kept in an image's header
  BigVec<uint32_t> sa_iseq;
  BigVec<uint32_t> sa_pos;        // offset (within its sequence) of every sampled row: what the text builder needs besides sa_iseq
                                  // (narrow indexes; empty where an offset does not fit 32 bits)
  // text verification (DevIndex::sa_full / text), built on the HOST for the test emulation only - the device builds its own
  BigVec<uint32_t> sa_full, row_seq;    // (row_seq: DevIndex::row_tax once build_text is through - dense taxon indices)
  std::vector<uint64_t> tax_of_dense;
  uint32_t beyond_lo = 0, beyond_n = 0, beyond_row = 0;   // DevIndex::beyond_*
  BigVec<uint8_t> text;
  BigVec<uint8_t> sa_tpos5;       // wide layout: DevIndex::sa_tpos5 (text position of every 2^tv_shift-th row)
  uint32_t tv_shift = 0;
  void build_text_wide(uint32_t shift);   // fills text / sa_tpos5 of a wide index (seq_walk_len / seq_walk_fill, kj_core.h)
  void build_text();                // fills sa_full / text (call after build / read_image); leaves them empty if not applicable
  std::vector<uint64_t> seq_taxid;
  std::vector<uint8_t> seq_valid;
  BigVec<uint64_t> term_pos;
  std::vector<std::string> names;   // sequence names (for the verbose columns)
  BigVec<uint2> kmer32;        // k-mer table (see DevIndex), one of the two is filled
  BigVec<ulonglong2> kmer64;
  uint32_t kmer_k = 0;
  BigVec<uint8_t> kline;       // k-mer lines of the host's table (DevIndex::kline; narrow indexes).  For the test emulation only:
                               // the device builds its own from the table it has grown (capi.hip), an image does not hold them
  // builds the k-mer table with k letters (0 = none); needs blocks/sb/C
  void build_kmer_table(uint32_t k);
  void build_klines();              // fills kline from kmer32 (call after build_kmer_table / read_image)
  uint64_t C[22] = {0};
  uint64_t bwtlen = 0, n_sa = 0, sa_skip = 0;
  uint32_t nseq = 0, chpt_exp = 0, alen = 0;
  uint32_t warnings = 0;
  std::string alphabet;
  uint8_t trans[128];               // translate2numbers table (sequence.c:68-97)
  // fills everything from a view; returns 0 or negative status
  int build(const HostIndexView &v, std::string &msg);
  // the parts of build() that do not touch bwt[] / sa[]: header fields, translation table, byte code -> letter (lcode[256]),
  // layout (narrow / wide) - and: sample geometry, warnings, names and taxon ids of the sequences
  int build_head(const HostIndexView &v, uint8_t *lcode, std::string &msg);
  void build_names(const HostIndexView &v, std::vector<std::string> *owned = nullptr);
  // A .fmi whose big arrays stay in the file (FmiFile::load(.., lazy)): the small parts are built here, `stream` says where the
  // rest is, `lazy` how many elements every device array will have; capi.hip lets fmi_stream_to_device pack them in HBM.
  FmiStreamSource stream;
  int build_streamed(FmiFile &f, const char *path, std::string &msg);     // (takes f's names)
  uint64_t bytes() const;
  // DevIndex whose pointers refer to THIS object's host vectors (used by the test emulation)
  DevIndex host_view() const;
  // kaijux / kaijup semantics (ConsumerThreadx.cpp:261-287): what a hit collects are database SEQUENCES, not taxa.
  // Afterwards the "taxon id" of sequence i is i itself and every name is usable.
  void to_sequence_ids();
  // the packed arrays as one file ("device image": written once, loaded instead of parsing and packing the .fmi)
  int write_image(const char *path, std::string &msg) const;
  // lazy_big: rank blocks, sampled sequence numbers / offsets, terminator rows and the k-mer table are not read - `lazy` says
  // where they are (a refseq-class image is 150 GB: eight ranks of a node cannot each hold a host copy of it)
  int read_image(const char *path, std::string &msg, bool lazy_big = false);
  static int image_source_bytes(const char *path, uint64_t &bytes, std::string &msg);   // header field of an image file
};

// the taxa of the sequences as dense indices in first-seen order (DevIndex::row_tax / tax_of_dense): seq_dense[i] = index of
// sequence i's taxon, 0xffffffff for a name without a usable id
void dense_taxa(const std::vector<uint64_t> &seq_taxid, const std::vector<uint8_t> &seq_valid, std::vector<uint32_t> &seq_dense,
                std::vector<uint64_t> &tax_of_dense);

// name -> taxon id with the rule of ids_from_SI (ConsumerThread.cpp:809-833)
bool parse_taxid(const char *name, uint64_t &id);

}  // namespace kj
