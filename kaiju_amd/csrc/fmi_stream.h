// fmi_stream.h — a .fmi straight to HBM: the file's byte-coded BWT and its sampled suffix array are streamed to the device in
// pieces and PACKED THERE into the layout the kernels read (kj_core.h: RankBlock64, sa_iseq / sa_taxid, term_pos, mb_base).
//
// Why: the host packer (host_index.cpp: PackedIndex::build) needs the parsed file AND the packed arrays in host memory - about
// twice the size of the .fmi (222 GB for refseq, where the reference itself needs 111: readIndexes bwt/bwt.c:78-88 reads the
// file into memory once, fmicommon.h:190-217).  Here the host holds the names of the sequences and two page-locked pieces; what
// the reference keeps in host memory goes through to the device.
//
// What is packed is exactly what PackedIndex::build packs (tests compare the arrays word for word): rank blocks are a running
// letter count over bwt[], done as a three-level prefix sum - inside a group of 256 blocks (k_pack_blocks), over the groups of
// a piece with the totals of the earlier pieces carried along (k_pack_scan), added back per block (k_pack_finish, which also
// writes the rows of the terminators).  C[] is only known when the last piece is through: narrow indexes get it added to every
// count by one more sweep over the blocks (k_pack_add_c), wide ones keep it in mb_base.
//
// The per-block / per-entry logic below is shared with the host emulation of the tests (tests/emu/kernel_emu.cpp).
#pragma once
#include <string>

#include "kj_core.h"

namespace kj {

constexpr uint32_t kPackGroupShift = 8;                         // rank blocks per group (= workgroup of k_pack_blocks): 256
constexpr uint32_t kPackGroupBlocks = 1u << kPackGroupShift;
constexpr uint32_t kPackGroupSymShift = kPackGroupShift + 6;    // symbols per group: 16384 (a divisor of every count base, 2^16 ..)
constexpr uint32_t kPackChannels = 21;                          // letter 0 (terminator) .. 20

// One rank block from `nsym` (<= 64) byte codes of the file's BWT: the five bit planes of the letters (padding = code 31, as
// the host packer writes it) and how often every letter 0..20 occurs.  lcode: byte code -> letter (fmi_fill_codes,
// compactfmi.c:75-89; 31 = a byte outside the table).  Returns false when such a byte occurs.
KJ_HD bool pack_block_letters(const uint8_t *raw, uint32_t nsym, const uint8_t *lcode, uint64_t pl[5], uint32_t cnt[kPackChannels]) {
  uint64_t p0 = 0, p1 = 0, p2 = 0, p3 = 0, p4 = 0;
  bool ok = true;
#pragma unroll
  for (uint32_t t = 0; t < 64; t++) {                  // (a fixed trip count: raw[] stays in registers on the device)
    if (t < nsym) {
      const uint32_t c = lcode[raw[t]];
      ok = ok && c != 31u;
      p0 |= (uint64_t)(c & 1u) << t; p1 |= (uint64_t)((c >> 1) & 1u) << t; p2 |= (uint64_t)((c >> 2) & 1u) << t;
      p3 |= (uint64_t)((c >> 3) & 1u) << t; p4 |= (uint64_t)((c >> 4) & 1u) << t;
    }
  }
  if (nsym < 64) {
    const uint64_t pad = nsym ? ~0ull << nsym : ~0ull;
    p0 |= pad; p1 |= pad; p2 |= pad; p3 |= pad; p4 |= pad;
  }
  pl[0] = p0; pl[1] = p1; pl[2] = p2; pl[3] = p3; pl[4] = p4;
#pragma unroll
  for (uint32_t c = 0; c < kPackChannels; c++) {
    uint64_t m = ~0ull;
    m &= (c & 1u) ? p0 : ~p0; m &= (c & 2u) ? p1 : ~p1; m &= (c & 4u) ? p2 : ~p2; m &= (c & 8u) ? p3 : ~p3; m &= (c & 16u) ? p4 : ~p4;
    cnt[c] = (uint32_t)popc64(m);
  }
  return ok;
}

// One entry of the file's sampled suffix array (suffixArray.h:37-51: nbytes bytes, big endian, sequence number above pbits
// bits of offset) -> what the kernels keep of it.
KJ_HD void pack_sa_entry(const uint8_t *e, int nbytes, int pbits, uint32_t &iseq, uint32_t &pos) {
  uint64_t val = 0;
  for (int q = 0; q < nbytes; q++) val = (val << 8) + e[q];
  iseq = (uint32_t)(val >> pbits);
  pos = (uint32_t)(val & (pbits >= 64 ? ~0ull : ((1ull << pbits) - 1ull)));
}

// where the two big arrays of a .fmi lie in the file and how they are coded (FmiFile::load(.., lazy))
struct FmiStreamSource {
  std::string path;           // empty: not a streamed load
  uint64_t sa_off = 0, bwt_off = 0;
  int32_t nbytes = 0, pbits = 0;
  uint8_t lcode[256];
};

// what a streamed load leaves on the device (every pointer from hipMalloc; the caller owns them)
struct FmiStreamResult {
  RankBlock64 *blocks64 = nullptr;
  uint64_t *mb_base = nullptr;        // wide only
  uint32_t *sa_iseq = nullptr;
  uint32_t *sa_pos = nullptr;         // narrow with pbits <= 32 only (the text builder's input; the caller frees it afterwards)
  uint64_t *sa_taxid = nullptr;       // narrow only, n_sa + 2 entries
  uint64_t *term_pos = nullptr;
  uint64_t C[22] = {0};
  uint64_t bytes_streamed = 0;
  double seconds = 0, seconds_reading = 0;
  size_t piece = 0;
};

struct PackedIndex;
// Host side (fmi_stream.hip).  pk: the small parts (PackedIndex::build_head / build_names done); d_seq_taxid / d_seq_valid:
// already on the device.  Returns 0 or a negative kaiju_gpu_status with `msg` set; on failure everything allocated is freed.
int fmi_stream_to_device(const FmiStreamSource &src, const PackedIndex &pk, const uint64_t *d_seq_taxid, const uint8_t *d_seq_valid,
                         FmiStreamResult &out, std::string &msg);

}  // namespace kj
