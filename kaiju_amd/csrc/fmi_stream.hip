// fmi_stream.hip — see fmi_stream.h: the BWT and the sampled suffix array of a .fmi go from the file to HBM in page-locked
// pieces and are packed by kernels; the host never holds either array (reference: readIndexes bwt/bwt.c:78-88 reads both into
// host memory, fmicommon.h:190-217, suffixArray.c:313-321).
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/kaiju_gpu.h"
#include "fmi_stream.h"
#include "host_index.h"

using namespace kj;

// ----------------------------------------------------------------------------------------
// kernels
// ----------------------------------------------------------------------------------------

// One thread per rank block, one workgroup per group of 256 blocks (16384 symbols of the piece).  Writes the block's planes
// and, in its count fields, how often every letter occurs IN FRONT OF the block INSIDE ITS GROUP (pad[0]: the terminators);
// the group's totals go to grp_tot.  raw: the piece's bytes in device memory (16-byte aligned), sym0: row of raw[0].
__global__ void __launch_bounds__(256)
k_pack_blocks(const uint8_t *__restrict__ raw, uint64_t sym0, uint64_t bwtlen, uint32_t nblk, const uint8_t *__restrict__ lcode_g,
              RankBlock64 *__restrict__ blocks, uint32_t *__restrict__ grp_tot, uint32_t *bad) {
  __shared__ uint8_t s_lcode[256];
  __shared__ uint32_t s_cnt[kPackChannels][kPackGroupBlocks + 1];
  const uint32_t tid = threadIdx.x;
  s_lcode[tid] = lcode_g[tid];
  __syncthreads();
  const uint32_t bi = blockIdx.x * kPackGroupBlocks + tid;
  uint64_t pl[5] = {~0ull, ~0ull, ~0ull, ~0ull, ~0ull};
  uint32_t cnt[kPackChannels];
#pragma unroll
  for (uint32_t c = 0; c < kPackChannels; c++) cnt[c] = 0;
  if (bi < nblk) {
    const uint64_t h0 = sym0 + (uint64_t)bi * 64;
    const uint32_t nsym = h0 >= bwtlen ? 0u : (uint32_t)min((uint64_t)64, bwtlen - h0);
    uint32_t w[16];
    const uint4 *src = reinterpret_cast<const uint4 *>(raw + (size_t)bi * 64);
#pragma unroll
    for (int q = 0; q < 4; q++) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if ((uint32_t)q * 16u < nsym) v = src[q];              // (bytes behind the end of the BWT are never looked at)
      w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
    }
    uint8_t by[64];
#pragma unroll
    for (int t = 0; t < 64; t++) by[t] = (uint8_t)(w[t >> 2] >> (8 * (t & 3)));
    if (!pack_block_letters(by, nsym, s_lcode, pl, cnt)) atomicOr(bad, 1u);
  }
#pragma unroll
  for (uint32_t c = 0; c < kPackChannels; c++) s_cnt[c][tid] = cnt[c];
  __syncthreads();
  if (tid < kPackChannels) {                                   // one thread per letter: exclusive prefix over the group's blocks
    uint32_t run = 0;
    for (uint32_t i = 0; i < kPackGroupBlocks; i++) { const uint32_t v = s_cnt[tid][i]; s_cnt[tid][i] = run; run += v; }
    grp_tot[(size_t)blockIdx.x * kPackChannels + tid] = run;
  }
  __syncthreads();
  if (bi < nblk) {
    RankBlock64 &r = blocks[bi];
#pragma unroll
    for (int q = 0; q < 5; q++) r.plane[q] = pl[q];
#pragma unroll
    for (uint32_t a = 0; a < 20; a++) r.cnt[a] = s_cnt[a + 1][tid];
    r.pad[0] = s_cnt[0][tid];
    r.pad[1] = 0;
  }
}

// One workgroup: exclusive prefix of the group totals of a piece on top of the totals of the earlier pieces (carry, updated),
// absolute and for every letter -> grp_abs; the count bases of the wide layout (every 2^mb_grp_shift groups; without C[]) ->
// mbcount.  Thread t owns ceil(ngroups / 1024) consecutive groups.
__global__ void __launch_bounds__(1024)
k_pack_scan(const uint32_t *__restrict__ grp_tot, uint32_t ngroups, uint64_t grp0, uint32_t mb_grp_shift, int wide, uint64_t *carry,
            uint64_t *__restrict__ mbcount, uint64_t *__restrict__ grp_abs) {
  __shared__ uint32_t s_part[kPackChannels][1024];
  const uint32_t tid = threadIdx.x, G = (ngroups + 1023u) / 1024u;
  const uint32_t g_lo = min(ngroups, tid * G), g_hi = min(ngroups, g_lo + G);
  uint32_t acc[kPackChannels];
#pragma unroll
  for (uint32_t c = 0; c < kPackChannels; c++) acc[c] = 0;
  for (uint32_t g = g_lo; g < g_hi; g++)
#pragma unroll
    for (uint32_t c = 0; c < kPackChannels; c++) acc[c] += grp_tot[(size_t)g * kPackChannels + c];
#pragma unroll
  for (uint32_t c = 0; c < kPackChannels; c++) s_part[c][tid] = acc[c];
  __syncthreads();
  const uint32_t wave = tid >> 6, lane = tid & 63u;
  for (uint32_t c = wave; c < kPackChannels; c += 16u) {        // a wavefront per letter: 1024 partial sums, 64 at a time
    uint32_t run = 0;
    for (uint32_t chunk = 0; chunk < 16u; chunk++) {
      const uint32_t v = s_part[c][chunk * 64u + lane];
      uint32_t incl = v;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(incl, (unsigned)d); if ((int)lane >= d) incl += t; }
      s_part[c][chunk * 64u + lane] = run + incl - v;
      run += __shfl(incl, 63);
    }
  }
  __syncthreads();
  uint64_t base[kPackChannels];
#pragma unroll
  for (uint32_t c = 0; c < kPackChannels; c++) base[c] = carry[c] + s_part[c][tid];
  __syncthreads();                                             // (every thread has read the carry before the last one replaces it)
  const uint64_t mbm = (1ull << mb_grp_shift) - 1ull;
  for (uint32_t g = g_lo; g < g_hi; g++) {
    const uint64_t gg = grp0 + g;
#pragma unroll
    for (uint32_t c = 0; c < kPackChannels; c++) grp_abs[(size_t)g * kPackChannels + c] = base[c];
    if (wide && (gg & mbm) == 0) {
      uint64_t *mc = mbcount + (size_t)(gg >> mb_grp_shift) * 20;
#pragma unroll
      for (uint32_t a = 0; a < 20; a++) mc[a] = base[a + 1];
    }
#pragma unroll
    for (uint32_t c = 0; c < kPackChannels; c++) base[c] += grp_tot[(size_t)g * kPackChannels + c];
  }
  if (tid == 1023u)
#pragma unroll
    for (uint32_t c = 0; c < kPackChannels; c++) carry[c] = base[c];
}

// One thread per rank block of the piece: the counts become "in front of the block since the start of the BWT" (narrow; C[]
// is added at the end, k_pack_add_c) or "... since the block's count base" (wide), and the rows of the block's terminators go
// to term_pos (they are met in ascending row order, so the n-th terminator of the BWT lands in entry n).
__global__ void __launch_bounds__(256)
k_pack_finish(RankBlock64 *__restrict__ blocks, uint32_t nblk, uint64_t sym0, uint64_t grp0, uint32_t mb_grp_shift, int wide,
              const uint64_t *__restrict__ grp_abs, const uint64_t *__restrict__ mbcount, uint64_t *__restrict__ term_pos, uint32_t nseq,
              uint32_t *bad) {
  const uint32_t bi = blockIdx.x * 256u + threadIdx.x;
  if (bi >= nblk) return;
  const uint32_t g = bi >> kPackGroupShift;
  const uint64_t *ab = grp_abs + (size_t)g * kPackChannels;
  RankBlock64 &r = blocks[bi];
  if (wide) {
    const uint64_t *mc = mbcount + (size_t)((grp0 + g) >> mb_grp_shift) * 20;
#pragma unroll
    for (uint32_t a = 0; a < 20; a++) r.cnt[a] += (uint32_t)(ab[a + 1] - mc[a]);
  } else {
#pragma unroll
    for (uint32_t a = 0; a < 20; a++) r.cnt[a] += (uint32_t)ab[a + 1];
  }
  uint64_t z = ab[0] + r.pad[0];
  uint64_t zm = ~(r.plane[0] | r.plane[1] | r.plane[2] | r.plane[3] | r.plane[4]);
  const uint64_t h0 = sym0 + (uint64_t)bi * 64;
  while (zm) {
    const uint32_t t = (uint32_t)__ffsll((unsigned long long)zm) - 1u;
    if (z < nseq) term_pos[z] = h0 + t; else atomicOr(bad, 2u);
    z++;
    zm &= zm - 1ull;
  }
  r.pad[0] = 0;
}

// narrow layout: C[letter] into every count (PackedIndex::build folds it in the same way)
__global__ void __launch_bounds__(256)
k_pack_add_c(RankBlock64 *__restrict__ blocks, uint64_t nb64, const uint64_t *__restrict__ c_of) {
  for (uint64_t bi = (uint64_t)blockIdx.x * 256 + threadIdx.x; bi < nb64; bi += (uint64_t)gridDim.x * 256) {
    RankBlock64 &r = blocks[bi];
#pragma unroll
    for (uint32_t a = 0; a < 20; a++) r.cnt[a] += (uint32_t)c_of[a];
  }
}
// wide layout: the count bases get it
__global__ void __launch_bounds__(256)
k_pack_mb_add_c(uint64_t *__restrict__ mb, uint64_t n_entries, const uint64_t *__restrict__ c_of) {
  for (uint64_t x = (uint64_t)blockIdx.x * 256 + threadIdx.x; x < n_entries; x += (uint64_t)gridDim.x * 256) mb[x] += c_of[x % 20];
}

// One thread per sampled suffix-array entry of the piece: the sequence number, (narrow) the offset for the text builder and the
// taxon id the locate ends with (PackedIndex::build does the same on the host).
__global__ void __launch_bounds__(256)
k_pack_sa(const uint8_t *__restrict__ raw, uint64_t i0, uint32_t n, int nbytes, int pbits, uint32_t nseq, const uint64_t *__restrict__ seq_taxid,
          const uint8_t *__restrict__ seq_valid, uint32_t *__restrict__ sa_iseq, uint32_t *__restrict__ sa_pos, uint64_t *__restrict__ sa_taxid) {
  const uint32_t x = blockIdx.x * 256u + threadIdx.x;
  if (x >= n) return;
  uint32_t is = 0, ps = 0;
  pack_sa_entry(raw + (size_t)x * (size_t)nbytes, nbytes, pbits, is, ps);
  sa_iseq[i0 + x] = is;
  if (sa_pos) sa_pos[i0 + x] = ps;
  if (sa_taxid) sa_taxid[i0 + x] = (is < nseq && seq_valid[is]) ? seq_taxid[is] : ~0ull;
}

// ----------------------------------------------------------------------------------------
// host
// ----------------------------------------------------------------------------------------
namespace {

double now_s() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

struct HipFail { std::string what; };
#define FS_HIP(call)                                                                          \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess) throw HipFail{std::string(#call) + ": " + hipGetErrorString(e_)};   \
  } while (0)

// file -> page-locked piece -> device staging buffer, two of each: the readers fill one piece while the other is on its way
// and being packed.  Copies and kernels share one stream, so a staging buffer is only overwritten when its kernels are done.
struct Streamer {
  int fd = -1;
  size_t piece = 0;
  void *host[2] = {nullptr, nullptr};
  uint8_t *dev[2] = {nullptr, nullptr};
  hipStream_t stream = nullptr;
  hipEvent_t copied[2] = {nullptr, nullptr};
  bool used[2] = {false, false};
  int k = 0;
  uint64_t bytes = 0;
  double t_read = 0;
  ~Streamer() {
    if (fd >= 0) close(fd);
    for (int q = 0; q < 2; q++) {
      if (host[q]) (void)hipHostFree(host[q]);
      if (dev[q]) (void)hipFree(dev[q]);
      if (copied[q]) (void)hipEventDestroy(copied[q]);
    }
    if (stream) (void)hipStreamDestroy(stream);
  }
  void open(const std::string &path, size_t piece_bytes) {
    fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) throw HipFail{"cannot open " + path};
    piece = piece_bytes;
    for (int q = 0; q < 2; q++) {
      FS_HIP(hipHostMalloc(&host[q], piece + 64, hipHostMallocDefault));
      FS_HIP(hipMalloc((void **)&dev[q], piece + 64));
      FS_HIP(hipEventCreateWithFlags(&copied[q], hipEventDisableTiming));
    }
    FS_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
  }
  // bytes [off, off + len) of the file into the next staging buffer (asynchronous from the copy on); returns that buffer
  const uint8_t *next(uint64_t off, size_t len) {
    if (used[k]) FS_HIP(hipEventSynchronize(copied[k]));               // the copy out of this page-locked piece has finished
    const double t0 = now_s();
    const unsigned nthreads = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    const size_t sub = std::max<size_t>((len + nthreads - 1) / nthreads, 1u << 20);
    std::atomic<bool> ok{true};
    std::vector<std::thread> th;
    uint8_t *d = static_cast<uint8_t *>(host[k]);
    auto rd = [&](size_t b, size_t e) {
      while (b < e) { const ssize_t r = pread(fd, d + b, e - b, (off_t)(off + b)); if (r <= 0) { ok = false; return; } b += (size_t)r; }
    };
    if (len <= sub) rd(0, len);
    else {
      for (size_t b = 0; b < len; b += sub) th.emplace_back(rd, b, std::min(len, b + sub));
      for (auto &x : th) x.join();
    }
    t_read += now_s() - t0;
    if (!ok.load()) throw HipFail{"short read from the .fmi file"};
    FS_HIP(hipMemcpyAsync(dev[k], host[k], len, hipMemcpyHostToDevice, stream));
    FS_HIP(hipEventRecord(copied[k], stream));
    used[k] = true;
    bytes += len;
    const uint8_t *p = dev[k];
    k ^= 1;
    return p;
  }
};

struct DevMem {                // frees what a failed load had allocated
  std::vector<void *> ptrs;
  bool keep = false;
  ~DevMem() { if (!keep) for (void *p : ptrs) (void)hipFree(p); }
  template <class T> T *alloc(size_t n, size_t slack_bytes = 32) {
    void *p = nullptr;
    FS_HIP(hipMalloc(&p, std::max<size_t>(n * sizeof(T), 16) + slack_bytes));
    ptrs.push_back(p);
    return static_cast<T *>(p);
  }
};

}  // namespace

namespace kj {

int fmi_stream_to_device(const FmiStreamSource &src, const PackedIndex &pk, const uint64_t *d_seq_taxid, const uint8_t *d_seq_valid,
                         FmiStreamResult &out, std::string &msg) {
  const double t_start = now_s();
  try {
    const uint64_t bwtlen = pk.bwtlen, nb64 = (bwtlen >> 6) + 1, n_sa = pk.n_sa;
    const bool wide = pk.wide;
    if (wide && (pk.mb_shift < kPackGroupSymShift + 1 || pk.mb_shift > 31)) { msg = "bad count-base shift"; return KAIJU_GPU_ERR_ARG; }
    const uint32_t mb_grp_shift = wide ? pk.mb_shift - kPackGroupSymShift : 0u;      // (narrow: no count bases; the kernels still shift by it)
    if (src.nbytes < 1 || src.nbytes > 8 || src.pbits < 0 || src.pbits > 62) { msg = "bad suffix array coding"; return KAIJU_GPU_ERR_FORMAT; }
    // piece size: KAIJU_GPU_STREAM_PIECE_MB / _KB (tests), default 256 MB, never much more than half of the larger array (a
    // small index does not pay for page-locking memory it does not fill); a multiple of the group size, at most 1 GB
    size_t piece = 256u << 20;
    if (const char *e = getenv("KAIJU_GPU_STREAM_PIECE_MB")) { const long v = atol(e); if (v >= 1 && v <= 1024) piece = (size_t)v << 20; }
    if (const char *e = getenv("KAIJU_GPU_STREAM_PIECE_KB")) { const long v = atol(e); if (v >= 16 && v <= (1024L << 10)) piece = (size_t)v << 10; }
    const uint64_t sa_bytes = n_sa * (uint64_t)src.nbytes;
    piece = (size_t)std::min<uint64_t>(piece, std::max(bwtlen, sa_bytes) / 2 + 1);
    const size_t gsym = (size_t)1 << kPackGroupSymShift;
    piece = std::max(gsym, (piece + gsym - 1) / gsym * gsym);
    out.piece = piece;

    Streamer st;
    st.open(src.path, piece);
    DevMem dm;
    RankBlock64 *blocks = dm.alloc<RankBlock64>((size_t)nb64);
    uint32_t *sa_iseq = dm.alloc<uint32_t>((size_t)n_sa);
    uint32_t *sa_pos = (!wide && src.pbits <= 32) ? dm.alloc<uint32_t>((size_t)n_sa) : nullptr;
    uint64_t *sa_taxid = !wide ? dm.alloc<uint64_t>((size_t)n_sa + 2) : nullptr;
    uint64_t *term_pos = dm.alloc<uint64_t>((size_t)pk.nseq);
    const uint64_t nmb = (bwtlen >> pk.mb_shift) + 1;
    uint64_t *mbcount = wide ? dm.alloc<uint64_t>((size_t)nmb * 20) : nullptr;
    // scratch of the pack kernels (freed at the end)
    DevMem tmp;
    const size_t max_groups = piece / gsym + 1;
    uint32_t *grp_tot = tmp.alloc<uint32_t>(max_groups * kPackChannels);
    uint64_t *grp_abs = tmp.alloc<uint64_t>(max_groups * kPackChannels);
    uint64_t *carry = tmp.alloc<uint64_t>(kPackChannels + 22);     // [21] running totals, then C[1..20] for the closing sweep
    uint8_t *d_lcode = tmp.alloc<uint8_t>(256);
    uint32_t *d_bad = tmp.alloc<uint32_t>(4);
    FS_HIP(hipMemsetAsync(carry, 0, (kPackChannels + 22) * 8, st.stream));
    FS_HIP(hipMemsetAsync(d_bad, 0, 16, st.stream));
    FS_HIP(hipMemcpyAsync(d_lcode, src.lcode, 256, hipMemcpyHostToDevice, st.stream));
    if (wide) FS_HIP(hipMemsetAsync(mbcount, 0, (size_t)nmb * 20 * 8, st.stream));
    if (sa_taxid) FS_HIP(hipMemsetAsync(sa_taxid + n_sa, 0xff, 16, st.stream));

    // ---- the sampled suffix array (it comes first in the file) ----
    {
      const uint64_t per_piece = piece / (uint64_t)src.nbytes;
      for (uint64_t i0 = 0; i0 < n_sa; i0 += per_piece) {
        const uint32_t n = (uint32_t)std::min<uint64_t>(per_piece, n_sa - i0);
        const uint8_t *raw = st.next(src.sa_off + i0 * (uint64_t)src.nbytes, (size_t)n * (size_t)src.nbytes);
        hipLaunchKernelGGL(k_pack_sa, dim3((n + 255u) / 256u), dim3(256), 0, st.stream, raw, i0, n, (int)src.nbytes, (int)src.pbits, pk.nseq,
                           d_seq_taxid, d_seq_valid, sa_iseq, sa_pos, sa_taxid);
        FS_HIP(hipGetLastError());
      }
    }
    // ---- the BWT ----
    for (uint64_t b = 0; b < bwtlen; b += piece) {
      const size_t len = (size_t)std::min<uint64_t>(piece, bwtlen - b);
      const bool last = b + len == bwtlen;
      const uint8_t *raw = st.next(src.bwt_off + b, len);
      const uint64_t blk0 = b >> 6;
      const uint32_t nblk = (uint32_t)((last ? nb64 : (b + len) >> 6) - blk0);      // (the last piece also writes the block behind the end)
      const uint32_t ngroups = (nblk + kPackGroupBlocks - 1) / kPackGroupBlocks;
      const uint64_t grp0 = b >> kPackGroupSymShift;
      hipLaunchKernelGGL(k_pack_blocks, dim3(ngroups), dim3(256), 0, st.stream, raw, b, bwtlen, nblk, d_lcode, blocks + blk0, grp_tot, d_bad);
      hipLaunchKernelGGL(k_pack_scan, dim3(1), dim3(1024), 0, st.stream, grp_tot, ngroups, grp0, mb_grp_shift, wide ? 1 : 0, carry, mbcount, grp_abs);
      hipLaunchKernelGGL(k_pack_finish, dim3((nblk + 255u) / 256u), dim3(256), 0, st.stream, blocks + blk0, nblk, b, grp0, mb_grp_shift, wide ? 1 : 0,
                         grp_abs, mbcount, term_pos, pk.nseq, d_bad);
      FS_HIP(hipGetLastError());
    }
    FS_HIP(hipStreamSynchronize(st.stream));
    // ---- totals -> C[]; checks of PackedIndex::build ----
    uint64_t total[kPackChannels];
    uint32_t bad[4] = {0, 0, 0, 0};
    FS_HIP(hipMemcpy(total, carry, sizeof total, hipMemcpyDeviceToHost));
    FS_HIP(hipMemcpy(bad, d_bad, sizeof bad, hipMemcpyDeviceToHost));
    uint64_t sum = 0;
    for (uint32_t a = 0; a < kPackChannels; a++) sum += total[a];
    if ((bad[0] & 1u) || sum != bwtlen) { msg = "BWT contains byte codes outside the code table"; return KAIJU_GPU_ERR_FORMAT; }
    if ((bad[0] & 2u) || total[0] != pk.nseq) { msg = "number of terminators in the BWT differs from nseq"; return KAIJU_GPU_ERR_FORMAT; }
    out.C[0] = 0;
    for (uint32_t a = 1; a < pk.alen; a++) out.C[a] = out.C[a - 1] + total[a - 1];
    for (uint32_t a = pk.alen; a < 22; a++) out.C[a] = bwtlen;
    FS_HIP(hipMemcpy(carry + kPackChannels, out.C + 1, 20 * 8, hipMemcpyHostToDevice));
    if (wide) {
      const uint64_t n = nmb * 20;
      hipLaunchKernelGGL(k_pack_mb_add_c, dim3((unsigned)std::min<uint64_t>((n + 255) / 256, 1u << 16)), dim3(256), 0, st.stream, mbcount, n, carry + kPackChannels);
    } else {
      hipLaunchKernelGGL(k_pack_add_c, dim3((unsigned)std::min<uint64_t>((nb64 + 255) / 256, 1u << 20)), dim3(256), 0, st.stream, blocks, nb64, carry + kPackChannels);
    }
    FS_HIP(hipGetLastError());
    FS_HIP(hipStreamSynchronize(st.stream));
    dm.keep = true;
    out.blocks64 = blocks; out.mb_base = mbcount; out.sa_iseq = sa_iseq; out.sa_pos = sa_pos; out.sa_taxid = sa_taxid; out.term_pos = term_pos;
    out.bytes_streamed = st.bytes;
    out.seconds_reading = st.t_read;
    out.seconds = now_s() - t_start;
    return KAIJU_GPU_OK;
  } catch (const HipFail &f) {
    (void)hipGetLastError();
    msg = f.what;
    if (f.what.compare(0, 11, "cannot open") == 0 || f.what.compare(0, 10, "short read") == 0) return KAIJU_GPU_ERR_IO;
    return f.what.find("out of memory") != std::string::npos ? KAIJU_GPU_ERR_NOMEM : KAIJU_GPU_ERR_HIP;
  }
}

}  // namespace kj
