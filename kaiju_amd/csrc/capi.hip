// capi.hip — HIP kernels (gfx950) and the C-ABI of include/kaiju_gpu.h.
//
// Kernels:
//   k_fragments  one lane per read: six-frame translation, fragment list in queue order,
//                eager SEG split for MEM (stage 1, kj_core.h:build_fragments)
//   k_fragments_fast  the same for mates up to 191 nt; k_trigcheck / k_segflag / k_seg / k_seg_apply*: SEG
//   k_mem        persistent lanes, one read at a time per lane: MEM search (+ locate for reads with many longest matches)
//   k_mem_locate one lane per read: the ids of the reads whose one or two longest matches k_mem left in the hit record
//   k_greedy2    persistent lanes: Greedy search (priority queue, substitutions) + locate
// Both search kernels are launched twice per batch: the main pass with small per-lane scratch
// and a retry pass (device-side work list, no host round trip) with worst-case scratch for the
// few reads whose match buffer / queue overflowed.
//
// There is no CPU fallback: without a HIP device every entry point that needs one fails with
// KAIJU_GPU_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <future>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>
#include <chrono>

#include "../../include/kaiju_gpu.h"
#include "fmi_stream.h"
#include "host_index.h"
#include "host_tables.h"
#include "kj_core.h"
#include "taxonomy.h"
#include "exact_pass.h"
#ifdef KJ_GREEDY3                    // the experimental row-pool Greedy lane: variant builds only
#include "kj_greedy3.h"
#endif

using namespace kj;

// ----------------------------------------------------------------------------------------
// kernels
// ----------------------------------------------------------------------------------------
constexpr int kBlock = 256;
constexpr uint32_t kLocDeferRows = 8;     // k_mem_locate / k_mem_post1: reads whose matches hold more rows go to the many-rows instantiation

__device__ __forceinline__ void load_tables(ConstTables &s_ct, const ConstTables *g_ct) {
  const uint32_t *src = reinterpret_cast<const uint32_t *>(g_ct);
  uint32_t *dst = reinterpret_cast<uint32_t *>(&s_ct);
  for (uint32_t i = threadIdx.x; i < sizeof(ConstTables) / 4; i += kBlock) dst[i] = src[i];
  __syncthreads();
}

// Stage 1, one lane per read, 64-thread blocks.  stage_bytes > 0: the six frame strings of every
// lane are staged in LDS (dword-interleaved over the wavefront) and copied out with 16-byte
// stores; 0: reads too long for LDS, strings are written in place.
constexpr int kFragBlock = 64;
__global__ void __launch_bounds__(kFragBlock)
k_fragments(const ConstTables *__restrict__ g_ct, Params p, SegTables st, Batch b, SegQueue sq, uint32_t *err,
            uint32_t stage_bytes) {
  __shared__ ConstTables s_ct;
  __shared__ int32_t s_entg[17];
  extern __shared__ __attribute__((aligned(16))) uint8_t s_stage[];
  if (threadIdx.x < 17) s_entg[threadIdx.x] = st.ent_g32[threadIdx.x];
  {
    const uint32_t *src = reinterpret_cast<const uint32_t *>(g_ct);
    uint32_t *dst = reinterpret_cast<uint32_t *>(&s_ct);
    for (uint32_t i = threadIdx.x; i < sizeof(ConstTables) / 4; i += kFragBlock) dst[i] = src[i];
    __syncthreads();
  }
  const uint32_t r = blockIdx.x * kFragBlock + threadIdx.x;
  if (r >= b.n_reads) return;
  uint32_t e = 0;
  build_fragments(s_ct, p, TrigCtx{s_entg, st.ent_locut32}, b, sq, r, &e,
                  stage_bytes ? s_stage + 4 * threadIdx.x : nullptr, 4 * kFragBlock, stage_bytes / 4);
  if (e) atomicOr(err, e);
}

// Stage 1, fast path (kj_core.h: build_fragments_fast): mates of up to kS1MaxLen nucleotides, one lane per read, frame
// strings stored unit by unit from registers.  LDS: the tables, the fragment list of every lane (dword-interleaved over the
// wavefront) and, TRIG, its letter-count rows.
constexpr int kS1Block = 64;
// UNITS: 16-residue units per frame string - kS1Units (mates up to 191 nt: the benchmark's reads) or kS1UnitsLong (up to 287 nt:
// 250-bp MiSeq reads; 128-bit masks)
template <bool TRIG, int UNITS = kS1Units>
__global__ void __launch_bounds__(kS1Block)
k_fragments_fast(const Stage1Tables *__restrict__ g_t, Params p, Batch b, SegQueue sq, uint32_t *err) {
  __shared__ __attribute__((aligned(16))) Stage1Tables s_t;
  __shared__ uint32_t s_codes[2 * kS1ListCap * kS1Block];
  __shared__ __attribute__((aligned(4))) uint8_t s_cnt[TRIG ? kS1Block * kS1CntStride : 4];
  {
    const uint4 *src = reinterpret_cast<const uint4 *>(g_t);
    uint4 *dst = reinterpret_cast<uint4 *>(&s_t);
    for (uint32_t i = threadIdx.x; i < sizeof(Stage1Tables) / 16; i += kS1Block) dst[i] = src[i];
    __syncthreads();
  }
  const uint32_t r = blockIdx.x * kS1Block + threadIdx.x;
  if (r >= b.n_reads) return;
  uint32_t e = 0;
  S1Lane ln;
  ln.codes = s_codes + threadIdx.x; ln.code_stride = kS1Block;
  ln.cnt = s_cnt + (TRIG ? threadIdx.x * kS1CntStride : 0);
  ln.tsbuf = nullptr;                                  // (the scan of whole strings takes its residues from registers)
  build_fragments_fast<TRIG, UNITS>(s_t, p, b, sq, r, &e, ln);
  if (e) atomicOr(err, e);
}

// ---- lazy SEG (MEM, kParamLazySeg) ----------------------------------------------------------------------
// The search lanes looked at the fragments unsplit and left, per read with a hit, the fragment(s) holding a longest match
// in Hit::reserved.  A read whose such fragments contain no 12-window at the SEG trigger entropy has the reference's
// result already: SEG would leave exactly those fragments alone, and the pieces of the others are substrings of their
// parents, which did not match longer.  The other reads are listed here and take the SEG pass.
__global__ void __launch_bounds__(256)
k_trigcheck(const Stage1Tables *__restrict__ g_t, Params p, Batch b, uint32_t *seglist, uint32_t *segcount) {
  __shared__ __attribute__((aligned(16))) Stage1Tables s_t;
  __shared__ __attribute__((aligned(4))) uint8_t s_cnt[256 * kS1CntStride];
  {
    const uint4 *src = reinterpret_cast<const uint4 *>(g_t);
    uint4 *dst = reinterpret_cast<uint4 *>(&s_t);
    for (uint32_t i = threadIdx.x; i < sizeof(Stage1Tables) / 16; i += 256) dst[i] = src[i];
    __syncthreads();
  }
  const uint32_t r = blockIdx.x * 256 + threadIdx.x;
  if (r >= b.n_reads) return;
  const uint32_t v = b.hits[r].reserved;
  if (v == 0) return;
  b.hits[r].reserved = 0;
  const bool need = lazy_seg_needed(s_t, p, b, r, b.hits + r, v, nullptr, s_cnt + threadIdx.x * kS1CntStride);
  if (need) seglist[atomicAdd(segcount, 1u)] = r;
}
// the listed reads: SEG trigger test of all their fragments (what stage 1 does eagerly elsewhere), hit record cleared
__global__ void __launch_bounds__(256)
k_segflag(Params p, SegTables st, Batch b, SegQueue sq, const uint32_t *seglist, const uint32_t *segcount, uint32_t *err) {
  __shared__ int64_t s_entg[13];
  if (threadIdx.x < 13) s_entg[threadIdx.x] = st.ent_g[threadIdx.x];
  __syncthreads();
  const SegCtx cx = seg_ctx(st, s_entg, nullptr);
  const uint32_t n = *segcount;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const uint32_t r = seglist[i];
    const ReadMeta rm = b.meta[r];
    Frag *F = b.frags + rm.frag;
    const uint8_t *pep = b.pep + rm.pep;
    const uint32_t nf = rm.nfrag & ~kNfragSegPending;
    uint32_t pending = 0;
    for (uint32_t k = 0; k < nf; k++) {
      const Frag f = F[k];
      uint32_t fl = kFragChecked;
      if (seg_triggers(cx, pep + f.start, (int)f.len)) {
        const uint32_t slot = atomicAdd(sq.count, 1u);
        if (slot >= sq.cap) atomicOr(err, 2u);
        else {
          SegWork wk; wk.read = r; wk.frag = k;
          sq.items[slot] = wk;
          pending = kNfragSegPending;
          fl = (slot + 1) << kFragSlotShift;
        }
      }
      F[k].flags = fl;
    }
    b.meta[r].nfrag = nf | pending;
    uint64_t *h8 = reinterpret_cast<uint64_t *>(b.hits + r);   // 184 bytes, 8-byte aligned
    for (int x = 0; x < (int)(sizeof(Hit) / 8); x++) h8[x] = 0;
  }
}
__global__ void __launch_bounds__(kBlock)
k_seg_apply_list(const ConstTables *__restrict__ g_ct, Params p, Batch b, SegQueue sq, const uint32_t *seglist,
                 const uint32_t *segcount, uint32_t *err) {
  __shared__ ConstTables s_ct;
  load_tables(s_ct, g_ct);
  const uint32_t n = *segcount;
  uint32_t e = 0;
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) seg_apply_mem(s_ct, p, b, sq, seglist[i], &e);
  if (e) atomicOr(err, e);
}

// ---- the post-search pass of the narrow MEM lanes, fused (round 6) ---------------------------------------------------------
// Until round 5 the records of a batch were streamed four times behind k_mem: k_trigcheck (4 bytes of 184 looked at),
// k_mem_locate, k_mem_locate_list and - a call of its own - k_lca.  k_mem_post1 is the three of them in ONE pass for the reads
// whose record is final after it (93 % of the benchmark's): the lazy-SEG look at the fragment that holds the longest matches
// (k_trigcheck's test), the ids of the matches (mem_locate_read, row -> taxon table) and, LCA, the 16-byte record
// (lca_from_ids util.cpp:194-263).  Reads that are not through - listed for the SEG pass, sent to the retry pass, matches of
// many rows - go on the `todo` list; k_mem_post2 finishes them behind the second search, the retry pass and the exact pass.
#ifdef KJ_POST_WAVES                      // (A/B measurements: any value makes the compiler aim at 68 registers and 96 more bytes of scratch)
#define KJ_POST_BOUNDS __launch_bounds__(256, KJ_POST_WAVES)
#else
#define KJ_POST_BOUNDS __launch_bounds__(256)
#endif
template <bool LCA>
__global__ void KJ_POST_BOUNDS
k_mem_post1(const Stage1Tables *__restrict__ g_t, DevIndex ix, Params p, Batch b, DevTaxonomy t, CompactHit *__restrict__ compact, int lazy,
            uint32_t *seglist, uint32_t *segcount, uint32_t *todo, uint32_t *todocount) {
  __shared__ __attribute__((aligned(16))) Stage1Tables s_t;
  __shared__ __attribute__((aligned(4))) uint8_t s_cnt[256 * kS1CntStride];
  if (lazy) {
    const uint4 *src = reinterpret_cast<const uint4 *>(g_t);
    uint4 *dst = reinterpret_cast<uint4 *>(&s_t);
    for (uint32_t i = threadIdx.x; i < sizeof(Stage1Tables) / 16; i += 256) dst[i] = src[i];
    __syncthreads();
  }
  const uint32_t r = blockIdx.x * 256 + threadIdx.x;
  bool need = false, later = false;
  if (r < b.n_reads) {
    Hit *h = b.hits + r;
    const uint32_t v = lazy ? h->reserved : 0u;
    if (v != 0) {
      h->reserved = 0;
#ifndef KJ_POST_NOTRIG                                       // (timing experiments only: wrong results)
      need = lazy_seg_needed(s_t, p, b, r, h, v, nullptr, s_cnt + threadIdx.x * kS1CntStride);
#endif
    }
    if (!need) {
      if (h->flags & kHitRetry) later = true;                               // (the retry pass writes this record)
#ifdef KJ_POST_NOLOCATE                                      // (timing experiments only: wrong results)
      else if (LCA) { CompactHit ch; ch.lca = 0; ch.best = h->best; ch.info = 0; compact[r] = ch; }
#else
      else if (!mem_locate_read<false>(ix, p, h, kLocDeferRows)) later = true;   // (matches of many rows: k_mem_post2's instantiation)
      else if (LCA) compact[r] = compact_hit(t, *h);
#endif
    }
  }
  // the two lists: one atomic per wavefront and list
  const uint32_t lane = threadIdx.x & 63u;
  {
    const uint64_t m = __ballot(need);
    if (m) {
      const uint32_t leader = (uint32_t)__builtin_ctzll(m);
      uint32_t base = 0;
      if (lane == leader) base = atomicAdd(segcount, (uint32_t)__popcll(m));
      base = (uint32_t)__shfl((int)base, (int)leader, 64);
      if (need) seglist[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = r;
    }
  }
  {
    const bool td = need || later;
    const uint64_t m = __ballot(td);
    if (m) {
      const uint32_t leader = (uint32_t)__builtin_ctzll(m);
      uint32_t base = 0;
      if (lane == leader) base = atomicAdd(todocount, (uint32_t)__popcll(m));
      base = (uint32_t)__shfl((int)base, (int)leader, 64);
      if (td) todo[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = r;
    }
  }
}
template <bool LCA>
__global__ void __launch_bounds__(256)
k_mem_post2(DevIndex ix, Params p, Batch b, DevTaxonomy t, CompactHit *__restrict__ compact, const uint32_t *__restrict__ todo,
            const uint32_t *__restrict__ todocount) {
  const uint32_t n = *todocount;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const uint32_t r = todo[i];
    Hit *h = b.hits + r;
    if (!mem_locate_read<false>(ix, p, h, kLocDeferRows)) mem_locate_read<false, true>(ix, p, h);
    if (LCA) compact[r] = compact_hit(t, *h);
  }
}

// Stage 1 for protein reads (kaiju -p, kaijup): one lane per read, peptides written in place
__global__ void __launch_bounds__(kFragBlock)
k_fragments_protein(const ConstTables *__restrict__ g_ct, Params p, SegTables st, Batch b, SegQueue sq, uint32_t *err) {
  __shared__ ConstTables s_ct;
  __shared__ int32_t s_entg[17];
  __shared__ uint8_t s_code[256];
  if (threadIdx.x < 17) s_entg[threadIdx.x] = st.ent_g32[threadIdx.x];
  {
    const uint32_t *src = reinterpret_cast<const uint32_t *>(g_ct);
    uint32_t *dst = reinterpret_cast<uint32_t *>(&s_ct);
    for (uint32_t i = threadIdx.x; i < sizeof(ConstTables) / 4; i += kFragBlock) dst[i] = src[i];
    for (uint32_t i = threadIdx.x; i < 256; i += kFragBlock) s_code[i] = 0;
    __syncthreads();
    if (threadIdx.x < 20) protein_code_entry(s_ct, threadIdx.x, s_code);
    __syncthreads();
  }
  const uint32_t r = blockIdx.x * kFragBlock + threadIdx.x;
  if (r >= b.n_reads) return;
  uint32_t e = 0;
  build_fragments_protein(s_ct, s_code, p, TrigCtx{s_entg, st.ent_locut32}, b, sq, r, &e);
  if (e) atomicOr(err, e);
}

// one wavefront per fragment: lanes share the sub-windows of s_Trim
struct CoopWave {
  uint64_t *prefix = nullptr;                     // LDS: 2 * (kSegPacked + 1) words (seg_trim's prefix counts)
  __device__ __forceinline__ uint64_t *pref() const { return prefix; }
  __device__ __forceinline__ void sync() const { __syncthreads(); }     // (the block is one wavefront)
  __device__ __forceinline__ int lane() const { return (int)(threadIdx.x & 63u); }
  __device__ __forceinline__ int width() const { return 64; }
  __device__ __forceinline__ void reduce_min(double &prob, int &t) const {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const double op = __shfl_xor(prob, o, 64);
      const int ot = __shfl_xor(t, o, 64);
      if (op < prob || (op == prob && ot < t)) { prob = op; t = ot; }
    }
  }
};

// a TEAM of T lanes per fragment (T a power of two below 64, the lanes of a team next to each other): the 64 / T teams of a
// wavefront work on fragments of their own
template <int T>
struct CoopTeam {
  uint64_t *prefix = nullptr;
  __device__ __forceinline__ uint64_t *pref() const { return prefix; }
  __device__ __forceinline__ void sync() const { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); }
  __device__ __forceinline__ int lane() const { return (int)(threadIdx.x & (uint32_t)(T - 1)); }
  __device__ __forceinline__ int width() const { return T; }
  __device__ __forceinline__ void reduce_min(double &prob, int &t) const {
#pragma unroll
    for (int o = T / 2; o > 0; o >>= 1) {
      const double op = __shfl_xor(prob, o, 64);
      const int ot = __shfl_xor(t, o, 64);
      if (op < prob || (op == prob && ot < t)) { prob = op; t = ot; }
    }
  }
};

// SEG pass: one wavefront (= one 64-thread block at a time) per fragment that stage 1 flagged; the
// fragment is copied to LDS first.  The number of fragments lives in device memory.
constexpr int kSegBlock = 64, kSegStage = 2048;
__global__ void __launch_bounds__(kSegBlock)
k_seg(Params p, SegTables st, Batch b, SegQueue sq) {
  __shared__ int64_t s_entg[13];
  __shared__ double s_lnf[kSegLnf];
  __shared__ __attribute__((aligned(16))) uint8_t s_frag[kSegStage];
  __shared__ int32_t s_work[4 * kSegMaxRegions];
  __shared__ uint8_t s_cls[kSegStage];
  if (threadIdx.x < 13) s_entg[threadIdx.x] = st.ent_g[threadIdx.x];
  if (threadIdx.x < kSegLnf) s_lnf[threadIdx.x] = st.lnfact[threadIdx.x];
  __syncthreads();
  const SegCtx cx = seg_ctx(st, s_entg, s_lnf);
  __shared__ uint64_t s_pref[2 * (kSegPacked + 1)];
  CoopWave coop;
  coop.prefix = s_pref;
  const uint32_t n = min(*sq.count, sq.cap);
  for (uint32_t s = blockIdx.x; s < n; s += gridDim.x)       // trip count is uniform over the block
    seg_compute(cx, coop, b, p, sq, s, s_frag, kSegStage, s_work, s_cls, [] { __syncthreads(); });
}
// The same with 64 / T fragments per wavefront.  A fragment of a 150-bp read has 35 residues on average: what a wavefront
// spends on it is mostly WAITING - the chain work item -> read -> fragment descriptor -> residues (four dependent loads), the
// window classes, the scan that runs the same in every lane - and only s_Trim's sub-windows (66 .. 200 of them for the usual
// raw segment of 12 .. 20 residues) use the lanes.  With teams the chains of 64 / T fragments overlap and s_Trim keeps the
// same number of lanes busy.  The teams of a wavefront run in lock step where their control flow agrees and one after the
// other where it does not; LDS: every team its own stage (kSegStage / teams bytes: longer fragments are read from device
// memory, seg_compute), classes and lists.  The team's "barrier" is a fence: its lanes belong to one wavefront.
#ifndef KJ_SEG_WAVES
#define KJ_SEG_WAVES 3                    // (131 registers: four wavefronts per SIMD would spill 36 bytes a lane)
#endif
template <int T>
__global__ void __launch_bounds__(kSegBlock, KJ_SEG_WAVES)
k_seg_teams(Params p, SegTables st, Batch b, SegQueue sq) {
  constexpr int NT = kSegBlock / T, kStage = kSegStage / NT;
  __shared__ int64_t s_entg[13];
  __shared__ double s_lnf[kSegLnf];
  __shared__ __attribute__((aligned(16))) uint8_t s_frag[kSegStage];
  __shared__ int32_t s_work[NT * 4 * kSegMaxRegions];
  __shared__ uint8_t s_cls[kSegStage];
  if (threadIdx.x < 13) s_entg[threadIdx.x] = st.ent_g[threadIdx.x];
  if (threadIdx.x < kSegLnf) s_lnf[threadIdx.x] = st.lnfact[threadIdx.x];
  __syncthreads();
  const SegCtx cx = seg_ctx(st, s_entg, s_lnf);
  __shared__ uint64_t s_pref[NT * 2 * (kSegPacked + 1)];
  const uint32_t team = threadIdx.x / (uint32_t)T;
  CoopTeam<T> coop;
  coop.prefix = s_pref + team * 2 * (kSegPacked + 1);
  const uint32_t n = min(*sq.count, sq.cap);
  for (uint32_t s0 = blockIdx.x * NT; s0 < n; s0 += gridDim.x * NT) {
    const uint32_t s = s0 + team;
    if (s < n)
      seg_compute(cx, coop, b, p, sq, s, s_frag + team * kStage, kStage, s_work + team * 4 * kSegMaxRegions, s_cls + team * kStage,
                  [] { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); });
  }
}

// MEM: apply the SEG records to the fragment lists
__global__ void __launch_bounds__(kBlock)
k_seg_apply(const ConstTables *__restrict__ g_ct, Params p, Batch b, SegQueue sq, uint32_t *err) {
  __shared__ ConstTables s_ct;
  load_tables(s_ct, g_ct);
  const uint32_t r = blockIdx.x * kBlock + threadIdx.x;
  if (r >= b.n_reads) return;
  uint32_t e = 0;
  seg_apply_mem(s_ct, p, b, sq, r, &e);
  if (e) atomicOr(err, e);
}

template <class P>
__device__ __forceinline__ void mem_body(const DevIndex &ix, const Params &p, const Batch &b, const WorkList &wl,
                                         SIEntry *si_all, uint32_t si_cap, const VerboseOut &vb) {
  __shared__ __attribute__((aligned(16))) uint8_t s_win[kBlock * kWinStride];
  const uint64_t lane = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  LaneScratch ls;
  ls.si = si_all + lane * si_cap;
  ls.si_cap = si_cap;
  ls.win = s_win + threadIdx.x * kWinStride;
  mem_lane<P>(ix, p, b, wl, ls, vb);
}
// second-generation lane (kj_core.h:mem_lane2): indexes below 2^32 symbols with a k-mer table
#ifndef KJ_MEM_WAVES
#define KJ_MEM_WAVES 4                             // wavefronts per SIMD the narrow MEM lane is compiled for (111 VGPRs since the locate left it)
#endif
#ifdef KJ_PROF
__global__ void __launch_bounds__(kBlock, 3)      // (the section marks need a few registers: no spills at three wavefronts per SIMD)
#else
__global__ void __launch_bounds__(kBlock, KJ_MEM_WAVES)
#endif
k_mem(DevIndex ix, Params p, Batch b, WorkList wl, SIEntry *si_all, uint32_t si_cap) {
  __shared__ __attribute__((aligned(16))) uint8_t s_win[kBlock * kWinStride];
  const uint64_t lane = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  LaneScratch ls;
  ls.si = si_all + lane * si_cap;
  ls.si_cap = si_cap;
  ls.win = s_win + threadIdx.x * kWinStride;
#ifdef KJ_PROF
  __shared__ unsigned long long s_prof[kBlock / 64][2 + 3 * PM_N];
  for (int x = threadIdx.x & 63; x < 2 + 3 * PM_N; x += 64) s_prof[threadIdx.x >> 6][x] = 0;
  ls.prof = s_prof[threadIdx.x >> 6];
  if ((threadIdx.x & 63) == 0) { ls.prof[0] = __builtin_readcyclecounter(); ls.prof[1] = PM_HEAD; }
#endif
  mem_lane2<false>(ix, p, b, wl, ls);
}
// the ids of the reads whose longest matches the lanes above left in their hit records (kParamDeferLocate): one lane per read
// (narrow index with the row -> sequence table: an id is two loads, and the rows of a match are neighbours in that table -
//  teams of lanes were SLOWER here even for matches of hundreds of rows, bench.py's hard leg: 12.2 -> 14.6 ms per 2 M reads.
//  What such matches cost was the look-up of every row's taxon among the ids collected so far: reads whose matches hold more
//  than kLocDeferRows rows are listed - one atomic per wavefront - and located by k_mem_locate_list, the instantiation that
//  keeps the collected ids in registers)
// WIDE: an index with 64-bit positions that had room for its row -> taxon table (112 GB at refseq_ref's 28 G rows: the 288 GB
// of an MI355X hold it next to the index; without it: k_mem_locate_wide below)
template <bool WIDE>
__global__ void __launch_bounds__(256)
k_mem_locate(DevIndex ix, Params p, Batch b, uint32_t *list, uint32_t *count) {
  const uint32_t r = blockIdx.x * 256 + threadIdx.x;
  const bool defer = r < b.n_reads && !mem_locate_read<WIDE>(ix, p, b.hits + r, list ? kLocDeferRows : 0u);
  const uint64_t m = __ballot(defer);
  if (m) {
    const uint32_t lane = threadIdx.x & 63u, leader = (uint32_t)__builtin_ctzll(m);
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(count, (uint32_t)__popcll(m));
    base = (uint32_t)__shfl((int)base, (int)leader, 64);
    if (defer) list[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = r;
  }
}
template <bool WIDE>
__global__ void __launch_bounds__(256)
k_mem_locate_list(DevIndex ix, Params p, Batch b, const uint32_t *__restrict__ list, const uint32_t *__restrict__ count) {
  const uint32_t n = *count;
  for (uint32_t t = blockIdx.x * 256 + threadIdx.x; t < n; t += gridDim.x * 256) mem_locate_read<WIDE, true>(ix, p, b.hits + list[t]);
}
// Indexes without the row -> sequence table (wide ones; narrow ones that had no room for the text arrays): a TEAM of kLocTeam
// lanes per read walks the rows of a match side by side (mem_locate_read_team)
#ifndef KJ_LOC_TEAM
#define KJ_LOC_TEAM 8
#endif
constexpr int kLocTeam = KJ_LOC_TEAM;             // (a power of two up to 64; 16 and 32 measured in round 6: DESIGN.md 6b)
__global__ void __launch_bounds__(256)
k_mem_locate_wide(DevIndex ix, Params p, Batch b) {
  const uint32_t r = (blockIdx.x * 256 + threadIdx.x) / kLocTeam;
  if (r >= b.n_reads) return;
  TeamWave<kLocTeam> team;
  mem_locate_read_team<true, kLocTeam>(ix, p, b.hits + r, team);
}
__global__ void __launch_bounds__(256)
k_mem_locate_team(DevIndex ix, Params p, Batch b) {
  const uint32_t r = (blockIdx.x * 256 + threadIdx.x) / kLocTeam;
  if (r >= b.n_reads) return;
  TeamWave<kLocTeam> team;
  mem_locate_read_team<false, kLocTeam>(ix, p, b.hits + r, team);
}
// the same kernel under a second name for the second search of the lazy SEG flow (the few reads whose fragments SEG had
// to cut), so that a kernel trace lists the full-size launches of k_mem by themselves
__global__ void __launch_bounds__(kBlock, KJ_MEM_WAVES)
k_mem_second(DevIndex ix, Params p, Batch b, WorkList wl, SIEntry *si_all, uint32_t si_cap) {
  __shared__ __attribute__((aligned(16))) uint8_t s_win[kBlock * kWinStride];
  const uint64_t lane = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  LaneScratch ls;
  ls.si = si_all + lane * si_cap;
  ls.si_cap = si_cap;
  ls.win = s_win + threadIdx.x * kWinStride;
  mem_lane2<false>(ix, p, b, wl, ls);
}
// the same lane with 64-bit positions: indexes of 2^32 rows and more (counts relative to mb_base, 16-byte k-mer entries)
__global__ void __launch_bounds__(kBlock, 3)
k_mem_wide2(DevIndex ix, Params p, Batch b, WorkList wl, SIEntry *si_all, uint32_t si_cap) {
  __shared__ __attribute__((aligned(16))) uint8_t s_win[kBlock * kWinStride];
  const uint64_t lane = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  LaneScratch ls;
  ls.si = si_all + lane * si_cap;
  ls.si_cap = si_cap;
  ls.win = s_win + threadIdx.x * kWinStride;
  __shared__ __attribute__((aligned(16))) uint8_t s_coop[(kBlock / 64) * kCoopBytesPerWave];
  ls.coop = s_coop + (threadIdx.x >> 6) * kCoopBytesPerWave;
  mem_lane2<true>(ix, p, b, wl, ls);
}
// counting instantiations (kaiju_gpu_set_count_ops): the same lanes adding up their memory steps (kj_core.h: OpCount);
// bench.py runs them once, untimed, for the algorithmic bytes of a launch
__global__ void __launch_bounds__(kBlock, 2)
k_mem_count(DevIndex ix, Params p, Batch b, WorkList wl, SIEntry *si_all, uint32_t si_cap) {
  __shared__ __attribute__((aligned(16))) uint8_t s_win[kBlock * kWinStride];
  const uint64_t lane = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  LaneScratch ls;
  ls.si = si_all + lane * si_cap;
  ls.si_cap = si_cap;
  ls.win = s_win + threadIdx.x * kWinStride;
  if (p.flags & kParamXOrder) mem_lane2<false, true, true>(ix, p, b, wl, ls);
  else mem_lane2<false, false, true>(ix, p, b, wl, ls);
}
__global__ void __launch_bounds__(kBlock, 2)
k_mem_wide2_count(DevIndex ix, Params p, Batch b, WorkList wl, SIEntry *si_all, uint32_t si_cap) {
  __shared__ __attribute__((aligned(16))) uint8_t s_win[kBlock * kWinStride];
  const uint64_t lane = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  LaneScratch ls;
  ls.si = si_all + lane * si_cap;
  ls.si_cap = si_cap;
  ls.win = s_win + threadIdx.x * kWinStride;
  __shared__ __attribute__((aligned(16))) uint8_t s_coop[(kBlock / 64) * kCoopBytesPerWave];
  ls.coop = s_coop + (threadIdx.x >> 6) * kCoopBytesPerWave;
  if (p.flags & kParamXOrder) mem_lane2<true, true, true>(ix, p, b, wl, ls);
  else mem_lane2<true, false, true>(ix, p, b, wl, ls);
}
// kaiju -v in MEM mode (kaiju_gpu_classify_batch_verbose): the same lanes noting where every recorded match lies in its read
// (kj_core.h: VERBOSE, LaneScratch::vbm = the reads' rows of VerboseOut::acc), and the pass that turns the records and those
// notes into columns 6 / 7 (mem_verbose_read) - in front of the locate kernels, which overwrite the matches with their ids
__global__ void __launch_bounds__(kBlock, 2)
k_mem_vb(DevIndex ix, Params p, Batch b, WorkList wl, SIEntry *si_all, uint32_t si_cap, uint32_t *vbm) {
  __shared__ __attribute__((aligned(16))) uint8_t s_win[kBlock * kWinStride];
  const uint64_t lane = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  LaneScratch ls;
  ls.si = si_all + lane * si_cap;
  ls.si_cap = si_cap;
  ls.win = s_win + threadIdx.x * kWinStride;
  ls.vbm = vbm;
  if (p.flags & kParamXOrder) mem_lane2<false, true, false, true>(ix, p, b, wl, ls);
  else mem_lane2<false, false, false, true>(ix, p, b, wl, ls);
}
__global__ void __launch_bounds__(kBlock, 2)
k_mem_wide2_vb(DevIndex ix, Params p, Batch b, WorkList wl, SIEntry *si_all, uint32_t si_cap, uint32_t *vbm) {
  __shared__ __attribute__((aligned(16))) uint8_t s_win[kBlock * kWinStride];
  const uint64_t lane = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  LaneScratch ls;
  ls.si = si_all + lane * si_cap;
  ls.si_cap = si_cap;
  ls.win = s_win + threadIdx.x * kWinStride;
  __shared__ __attribute__((aligned(16))) uint8_t s_coop[(kBlock / 64) * kCoopBytesPerWave];
  ls.coop = s_coop + (threadIdx.x >> 6) * kCoopBytesPerWave;
  ls.vbm = vbm;
  if (p.flags & kParamXOrder) mem_lane2<true, true, false, true>(ix, p, b, wl, ls);
  else mem_lane2<true, false, false, true>(ix, p, b, wl, ls);
}
template <bool WIDE, bool TEXT = true>
__global__ void __launch_bounds__(256)
k_mem_verbose(DevIndex ix, Params p, Batch b, VerboseOut vb) {
  const uint32_t r = blockIdx.x * 256 + threadIdx.x;
  if (r < b.n_reads) mem_verbose_read<WIDE, TEXT>(ix, p, b, r, vb);
}
// Column 7 on its way to the host: the lanes write a read's peptides as index-alphabet codes into its own row of text_cap bytes
// (1 KB per 150-bp read, of which a classified read uses ~40).  This pass turns the codes into letters and packs the rows of all
// reads into one string - a wavefront's reads next to each other, the wavefronts where one atomic per wavefront puts them -
// so that the host fetches the bytes that were written (tens of MB per 2 M reads) instead of every row (2 GB).
struct VbAlphabet { char c[32]; uint32_t n; };
__global__ void __launch_bounds__(256)
k_vb_pack(VerboseOut vb, uint32_t n, VbAlphabet al, uint8_t *__restrict__ packed, uint64_t *__restrict__ pos, unsigned long long *total) {
  const uint32_t r = blockIdx.x * 256 + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t w = 0;
  if (r < n) { w = vb.text_len[r]; if (w > vb.text_cap) w = vb.text_cap; }
  uint32_t incl = w;                                       // inclusive prefix sum over the wavefront
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)incl, d, 64); if (lane >= (uint32_t)d) incl += v; }
  const uint32_t wave_total = (uint32_t)__shfl((int)incl, 63, 64);
  unsigned long long base = 0;
  if (lane == 63u && wave_total) base = atomicAdd(total, (unsigned long long)wave_total);
  base = (unsigned long long)__shfl((long long)base, 63, 64);
  if (r >= n) return;
  const uint64_t at = base + incl - w;
  pos[r] = at;
  const uint8_t *src = vb.text + (size_t)r * vb.text_cap;
  for (uint32_t x = 0; x < w; x++) { const uint8_t c = src[x]; packed[at + x] = c == 255 ? (uint8_t)',' : (c < al.n ? (uint8_t)al.c[c] : (uint8_t)'?'); }
}
// kaijux (ids = database sequences): the matches of a fragment are visited in the list order of maxMatches(.., 1)
// (kj_core.h: XORDER); the first-generation lanes take the same switch from Params::flags
__global__ void __launch_bounds__(kBlock, 4)
k_mem_x(DevIndex ix, Params p, Batch b, WorkList wl, SIEntry *si_all, uint32_t si_cap) {
  __shared__ __attribute__((aligned(16))) uint8_t s_win[kBlock * kWinStride];
  const uint64_t lane = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  LaneScratch ls;
  ls.si = si_all + lane * si_cap;
  ls.si_cap = si_cap;
  ls.win = s_win + threadIdx.x * kWinStride;
  mem_lane2<false, true>(ix, p, b, wl, ls);
}
__global__ void __launch_bounds__(kBlock, 3)
k_mem_wide2_x(DevIndex ix, Params p, Batch b, WorkList wl, SIEntry *si_all, uint32_t si_cap) {
  __shared__ __attribute__((aligned(16))) uint8_t s_win[kBlock * kWinStride];
  const uint64_t lane = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  LaneScratch ls;
  ls.si = si_all + lane * si_cap;
  ls.si_cap = si_cap;
  ls.win = s_win + threadIdx.x * kWinStride;
  __shared__ __attribute__((aligned(16))) uint8_t s_coop[(kBlock / 64) * kCoopBytesPerWave];
  ls.coop = s_coop + (threadIdx.x >> 6) * kCoopBytesPerWave;
  mem_lane2<true, true>(ix, p, b, wl, ls);
}
// first-generation lane with 32-bit positions (kept for A/B measurements: KAIJU_GPU_MEM_LANE=v1)
__global__ void __launch_bounds__(kBlock)
k_mem_v1(DevIndex ix, Params p, Batch b, WorkList wl, SIEntry *si_all, uint32_t si_cap, VerboseOut vb) { mem_body<uint32_t>(ix, p, b, wl, si_all, si_cap, vb); }
// 64-bit positions (refseq-scale indexes)
__global__ void __launch_bounds__(kBlock)
k_mem_wide(DevIndex ix, Params p, Batch b, WorkList wl, SIEntry *si_all, uint32_t si_cap, VerboseOut vb) { mem_body<uint64_t>(ix, p, b, wl, si_all, si_cap, vb); }
// the same lanes over the device-side retry list, with worst-case scratch (separate symbol so
// that profiles list the two passes separately)
__global__ void __launch_bounds__(kBlock)
k_mem_retry(DevIndex ix, Params p, Batch b, WorkList wl, SIEntry *si_all, uint32_t si_cap, VerboseOut vb) { mem_body<uint64_t>(ix, p, b, wl, si_all, si_cap, vb); }

struct GreedyArrays {
  GItem *pool; uint16_t *ord; GMatch *matches; GBest *best;
  uint32_t pool_cap, match_cap;
  GBestV *bestv;                 // verbose output only (else nullptr)
};

__device__ __forceinline__ void greedy_body(const DevIndex &ix, const ConstTables *__restrict__ g_ct, const Params &p,
                                            const SegQueue &sq, const Batch &b, const WorkList &wl,
                                            const GreedyArrays &ga, const VerboseOut &vb) {
  __shared__ __attribute__((aligned(16))) uint8_t s_win[kBlock * kWinStride];
  __shared__ ConstTables s_ct;
  load_tables(s_ct, g_ct);
  const uint64_t lane = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  GreedyScratch gs;
  gs.pool = ga.pool + lane * ga.pool_cap; gs.pool_cap = ga.pool_cap;
  gs.ord = ga.ord + lane * ga.pool_cap;
  gs.matches = ga.matches + lane * ga.match_cap; gs.match_cap = ga.match_cap;
  gs.best = ga.best + lane * 64;
  gs.win = s_win + threadIdx.x * kWinStride;
  gs.bestv = ga.bestv ? ga.bestv + lane * 64 : nullptr;
  greedy_lane(ix, s_ct, p, sq, b, wl, gs, vb);
}
__global__ void __launch_bounds__(kBlock)
k_greedy(DevIndex ix, const ConstTables *__restrict__ g_ct, Params p, SegQueue sq, Batch b, WorkList wl, GreedyArrays ga,
         VerboseOut vb) {
  greedy_body(ix, g_ct, p, sq, b, wl, ga, vb);
}
__global__ void __launch_bounds__(kBlock)
k_greedy_retry(DevIndex ix, const ConstTables *__restrict__ g_ct, Params p, SegQueue sq, Batch b, WorkList wl, GreedyArrays ga,
               VerboseOut vb) {
  greedy_body(ix, g_ct, p, sq, b, wl, ga, vb);
}

// second-generation Greedy lane (kj_core.h:greedy_lane2): indexes below 2^32 symbols with a k-mer table.
// Dynamic LDS: per lane a window row, a match-length row and a priority row, then the constant tables.
struct GreedyArrays2 {
  u128 *pool; uint32_t *prio_ext; GMatch2 *matches; uint16_t *mq_ext; GBest2 *best;
  uint32_t gate;
  GBest2W *bestw;                // wide indexes: best is nullptr then
  GBestV *bestv;                 // kaiju -v (k_greedy2_vb / k_greedy2_wide_vb): 64 per lane; else nullptr
  VerboseOut vb;
};
constexpr size_t kGreedy2Lds = (size_t)kBlock * (kGWinStride + kGMqStride + kGPrioStride + kGSubStride) * 4 + sizeof(ConstTables);
template <bool COUNT, bool WIDE, bool VERBOSE = false>
__device__ __forceinline__ void greedy2_body(const DevIndex &ix, const ConstTables *__restrict__ g_ct, const Params &p, const SegQueue &sq,
                                             const Batch &b, const WorkList &wl, const GreedyArrays2 &ga) {
  extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];
  uint32_t *s_prio = s_dyn;                                   // 16-byte aligned rows
  uint32_t *s_win = s_prio + kBlock * kGPrioStride;
  uint32_t *s_mq = s_win + kBlock * kGWinStride;
  uint32_t *s_sub = s_mq + kBlock * kGMqStride;
  ConstTables &s_ct = *reinterpret_cast<ConstTables *>(s_sub + kBlock * kGSubStride);
  load_tables(s_ct, g_ct);
  const uint64_t lane = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  GreedyScratch2 gs;
  gs.sub = s_sub + threadIdx.x * kGSubStride;
  gs.win = reinterpret_cast<uint8_t *>(s_win + threadIdx.x * kGWinStride);
  gs.mq = reinterpret_cast<uint16_t *>(s_mq + threadIdx.x * kGMqStride);
  gs.prio = s_prio + threadIdx.x * kGPrioStride;
  // (the device-memory scratch as bases of all lanes: the lane computes its own pieces from its number)
  gs.pool = ga.pool; gs.prio_ext = ga.prio_ext; gs.matches = ga.matches; gs.mq_ext = ga.mq_ext; gs.best = ga.best; gs.bestw = ga.bestw;
  gs.lane = (uint32_t)lane;
  gs.gate = ga.gate;
  gs.prof = nullptr;
#ifdef KJ_PROF
  __shared__ unsigned long long s_prof[kBlock / 64][2 + 3 * PS_N];
  if constexpr (!COUNT) {
    for (int x = threadIdx.x & 63; x < 2 + 3 * PS_N; x += 64) s_prof[threadIdx.x >> 6][x] = 0;
    gs.prof = s_prof[threadIdx.x >> 6];
    if ((threadIdx.x & 63) == 0) { gs.prof[0] = __builtin_readcyclecounter(); gs.prof[1] = PS_HEAD; }
  }
#endif
  if constexpr (VERBOSE) { gs.bestv = ga.bestv; gs.vb = ga.vb; }
  greedy_lane2<COUNT, WIDE, VERBOSE>(ix, s_ct, p, sq, b, wl, gs);
}
__global__ void __launch_bounds__(kBlock, kGreedyWavesPerSimd)
k_greedy2(DevIndex ix, const ConstTables *__restrict__ g_ct, Params p, SegQueue sq, Batch b, WorkList wl, GreedyArrays2 ga) {
  greedy2_body<false, false>(ix, g_ct, p, sq, b, wl, ga);
}
__global__ void __launch_bounds__(kBlock, 1)
k_greedy2_count(DevIndex ix, const ConstTables *__restrict__ g_ct, Params p, SegQueue sq, Batch b, WorkList wl, GreedyArrays2 ga) {
  greedy2_body<true, false>(ix, g_ct, p, sq, b, wl, ga);
}
// the same lane with 64-bit positions: indexes of 2^32 rows and more (the k-mer table instead of the lines)
#ifndef KJ_GW_WAVES
#define KJ_GW_WAVES kGreedyWavesPerSimd             // (169 VGPRs since the locate left the lane: one register from three wavefronts per SIMD)
#endif
__global__ void __launch_bounds__(kBlock, KJ_GW_WAVES)
k_greedy2_wide(DevIndex ix, const ConstTables *__restrict__ g_ct, Params p, SegQueue sq, Batch b, WorkList wl, GreedyArrays2 ga) {
  greedy2_body<false, true>(ix, g_ct, p, sq, b, wl, ga);
}
__global__ void __launch_bounds__(kBlock, 1)
k_greedy2_wide_count(DevIndex ix, const ConstTables *__restrict__ g_ct, Params p, SegQueue sq, Batch b, WorkList wl, GreedyArrays2 ga) {
  greedy2_body<true, true>(ix, g_ct, p, sq, b, wl, ga);
}

// kaiju -v in Greedy mode: the same lanes keeping, per best match, where its peptide comes from (kj_core.h: VERBOSE, GBestV) and
// writing column 7 when a read is through; column 6 follows from the records (k_mem_verbose<.., false>, in front of the locate)
__global__ void __launch_bounds__(kBlock, 2)
k_greedy2_vb(DevIndex ix, const ConstTables *__restrict__ g_ct, Params p, SegQueue sq, Batch b, WorkList wl, GreedyArrays2 ga) {
  greedy2_body<false, false, true>(ix, g_ct, p, sq, b, wl, ga);
}
__global__ void __launch_bounds__(kBlock, 2)
k_greedy2_wide_vb(DevIndex ix, const ConstTables *__restrict__ g_ct, Params p, SegQueue sq, Batch b, WorkList wl, GreedyArrays2 ga) {
  greedy2_body<false, true, true>(ix, g_ct, p, sq, b, wl, ga);
}

#ifdef KJ_GREEDY3
// third-generation Greedy (kj_greedy3.h): ONE block of kG3Threads threads per CU owns kG3Pool reads as rows of LDS, its
// wavefronts pull rows by class - the lanes of a wavefront run the same piece of the algorithm.  Narrow indexes with k-mer lines.
template <bool COUNT>
__device__ __forceinline__ void greedy3_body(const DevIndex &ix, const ConstTables *__restrict__ g_ct, const Params &p, const SegQueue &sq,
                                             const Batch &b, const WorkList &wl, const GreedyArrays2 &ga, uint32_t split) {
  __shared__ __attribute__((aligned(16))) uint32_t s_all[kG3LdsBytes / 4];
  uint32_t *s_prio = s_all;
  uint32_t *s_mq = s_prio + kG3Pool * kG3PrioWords;
  uint32_t *s_win = s_mq + kG3Pool * kG3MqWords;
  uint32_t *s_st = s_win + kG3Pool * kG3WinWords;
  uint32_t *s_cls = s_st + kG3Pool * kG3StWords;
  uint32_t *s_cnt = s_cls + kG3Pool;
  uint16_t *s_tmp = reinterpret_cast<uint16_t *>(s_cnt + 32);
  ConstTables &s_ct = *reinterpret_cast<ConstTables *>(s_tmp + (kG3Threads / 64) * 64);
  for (uint32_t x = threadIdx.x; x < (uint32_t)(kG3Pool * (kG3RowBytes / 4)); x += blockDim.x) s_all[x] = 0;
  for (uint32_t x = threadIdx.x; x < (uint32_t)kG3Pool; x += blockDim.x) s_cls[x] = C3_IDLE;
  if (threadIdx.x < 32) s_cnt[threadIdx.x] = threadIdx.x == (uint32_t)C3_IDLE ? (uint32_t)kG3Pool : 0u;
  {
    const uint32_t *src = reinterpret_cast<const uint32_t *>(g_ct);
    uint32_t *dst = reinterpret_cast<uint32_t *>(&s_ct);
    for (uint32_t x = threadIdx.x; x < sizeof(ConstTables) / 4; x += blockDim.x) dst[x] = src[x];
  }
  __syncthreads();
  G3Ctx gx;
  gx.prio = s_prio; gx.win = s_win; gx.mq = s_mq; gx.st = s_st; gx.cls = s_cls; gx.cnt = s_cnt;
  gx.tmp = s_tmp + (threadIdx.x >> 6) * 64;
  gx.pool = ga.pool; gx.prio_ext = ga.prio_ext; gx.matches = ga.matches; gx.mq_ext = ga.mq_ext; gx.best = ga.best;
  gx.row0 = blockIdx.x * (uint32_t)kG3Pool; gx.npool = kG3Pool; gx.split = split;
  gx.prof = nullptr;
#ifdef KJ_PROF
  __shared__ unsigned long long s_prof[kG3Threads / 64][2 + 3 * PS_N];
  if constexpr (!COUNT) {
    for (int x = threadIdx.x & 63; x < 2 + 3 * PS_N; x += 64) s_prof[threadIdx.x >> 6][x] = 0;
    gx.prof = s_prof[threadIdx.x >> 6];
    if ((threadIdx.x & 63) == 0) { gx.prof[0] = __builtin_readcyclecounter(); gx.prof[1] = PS_HEAD; }
  }
#endif
  greedy_lane3<COUNT>(ix, s_ct, p, sq, b, wl, gx);
}
__global__ void __launch_bounds__(kG3Threads, 2)
k_greedy3(DevIndex ix, const ConstTables *__restrict__ g_ct, Params p, SegQueue sq, Batch b, WorkList wl, GreedyArrays2 ga, uint32_t split) {
  greedy3_body<false>(ix, g_ct, p, sq, b, wl, ga, split);
}
__global__ void __launch_bounds__(kG3Threads, 2)
k_greedy3_count(DevIndex ix, const ConstTables *__restrict__ g_ct, Params p, SegQueue sq, Batch b, WorkList wl, GreedyArrays2 ga, uint32_t split) {
  greedy3_body<true>(ix, g_ct, p, sq, b, wl, ga, split);
}

#endif

// Index load: the k-mer table one letter deeper.  child[idx * 20 + c - 1] = UpdateSI(parent[idx], c)
// (bwt.c:160-173) for all 20 letters from the two rank blocks at the ends of the parent's interval;
// empty intervals stay {0, 0}.  One thread per parent, 160 contiguous bytes of children each.
__global__ void __launch_bounds__(256)
k_kmer_extend(const RankBlock64 *__restrict__ blk, const uint2 *__restrict__ parent, uint2 *__restrict__ child,
              uint64_t n_parent) {
  for (uint64_t idx = (uint64_t)blockIdx.x * 256 + threadIdx.x; idx < n_parent; idx += (uint64_t)gridDim.x * 256) {
    const uint2 e = parent[idx];
    uint32_t out[40];
#pragma unroll
    for (int x = 0; x < 40; x++) out[x] = 0;
    if (e.y != 0) {
      const uint32_t lo = e.x, hi = e.x + e.y;
      const RankBlock64 &A = blk[lo >> 6], &B = blk[hi >> 6];
      const uint64_t lowA = (1ull << (lo & 63u)) - 1ull, lowB = (1ull << (hi & 63u)) - 1ull;
#pragma unroll
      for (int c = 1; c <= 20; c++) {
        uint64_t ma = lowA, mb = lowB;
#pragma unroll
        for (int bit = 0; bit < 5; bit++) {
          ma &= ((c >> bit) & 1) ? A.plane[bit] : ~A.plane[bit];
          mb &= ((c >> bit) & 1) ? B.plane[bit] : ~B.plane[bit];
        }
        const uint32_t ra = A.cnt[c - 1] + (uint32_t)__popcll(ma), rb = B.cnt[c - 1] + (uint32_t)__popcll(mb);
        if (ra < rb) { out[2 * (c - 1)] = ra; out[2 * (c - 1) + 1] = rb - ra; }
      }
    }
    uint4 *dst = reinterpret_cast<uint4 *>(child + idx * 20);
#pragma unroll
    for (int x = 0; x < 10; x++) dst[x] = make_uint4(out[4 * x], out[4 * x + 1], out[4 * x + 2], out[4 * x + 3]);
  }
}

// Index load, narrow indexes: the k-mer LINES of the table just grown (kj_core.h: DevIndex::kline, kline_build_one): one
// thread per line - twenty entries gathered with a stride of 20^(k-1) (coalesced over the threads), twenty consecutive ones
// for the presence bits, the BWT letter of every one-row interval from its rank block.
__global__ void __launch_bounds__(256)
k_kline_build(DevIndex ix, uint32_t k, uint64_t n_lines, uint8_t *__restrict__ lines) {
  for (uint64_t code = (uint64_t)blockIdx.x * 256 + threadIdx.x; code < n_lines; code += (uint64_t)gridDim.x * 256)
    kline_build_one(ix, k, code, lines + code * kKLineBytes);
}

// Index load, narrow indexes with room to spare: the database text and the full suffix array for the text verification of the
// MEM lane (kj_core.h: DevIndex::sa_full / text).  k_suffix_walk: (sequence, offset) of the suffix of every row by the
// reference's own walk (suffix_of_row = get_suffix, bwt.c:105-121); k_seq_lens: a sequence's length = the offset of its
// terminator suffix (rows 0 .. nseq-1); k_text_build: row r's suffix lies at g = off[sequence] + 1 + offset, and the letter
// in front of it is the row's BWT letter.
__global__ void __launch_bounds__(256)
k_suffix_walk(DevIndex ix, const uint32_t *__restrict__ smp_pos, uint32_t *__restrict__ row_seq, uint32_t *__restrict__ row_pos, uint32_t *bad,
              uint32_t *__restrict__ beyond_rows) {
  // bad[0]: a row could not be resolved; bad[1]: number of rows whose walk passed the missing sample of an index with the
  // reference's short sample array (beyond_rows[0 .. kBeyondRowsMax): those rows - k_rows_beyond unsets their sequence)
  for (uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x; r < ix.bwtlen; r += (uint64_t)gridDim.x * 256) {
    uint32_t sq = 0, ps = 0;
    bool by = false;
    if (!suffix_of_row(ix, smp_pos, r, sq, ps, &by) || sq >= ix.nseq) { atomicOr(bad, 1u); sq = 0; ps = 0; }
    if (by) { const uint32_t at = atomicAdd(bad + 1, 1u); if (at < kBeyondRowsMax) beyond_rows[at] = (uint32_t)r; }
    row_seq[r] = sq; row_pos[r] = ps;
  }
}
__global__ void __launch_bounds__(256)
k_rows_beyond(const uint32_t *__restrict__ beyond_rows, uint32_t n, uint32_t *__restrict__ row_seq) {
  const uint32_t x = blockIdx.x * 256 + threadIdx.x;
  if (x < n) row_seq[beyond_rows[x]] = 0xffffffffu;          // a locate of that row finds no sequence (the reference reads out of bounds there)
}
// row -> sequence becomes row -> dense taxon index (DevIndex::row_tax), in place; out[x] = src[idx[x]] (the text positions of the
// few rows behind the missing sample)
__global__ void __launch_bounds__(256)
k_row_tax(uint32_t *__restrict__ row_seq, uint64_t n, const uint32_t *__restrict__ seq_dense, uint32_t nseq) {
  for (uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x; r < n; r += (uint64_t)gridDim.x * 256) {
    const uint32_t q = row_seq[r];
    row_seq[r] = q < nseq ? seq_dense[q] : 0xffffffffu;
  }
}
__global__ void __launch_bounds__(256)
k_gather_u32(const uint32_t *__restrict__ src, const uint32_t *__restrict__ idx, uint32_t n, uint32_t *__restrict__ out) {
  const uint32_t x = blockIdx.x * 256 + threadIdx.x;
  if (x < n) out[x] = src[idx[x]];
}
__global__ void __launch_bounds__(256)
k_seq_lens(uint32_t nseq, const uint32_t *__restrict__ row_seq, const uint32_t *__restrict__ row_pos, uint32_t *__restrict__ len) {
  const uint32_t r = blockIdx.x * 256 + threadIdx.x;
  if (r < nseq) len[row_seq[r]] = row_pos[r];
}
__global__ void __launch_bounds__(256)
k_text_build(DevIndex ix, const uint32_t *__restrict__ row_seq, const uint32_t *__restrict__ row_pos, const uint32_t *__restrict__ off,
             uint32_t *__restrict__ sa_full, uint8_t *__restrict__ text) {
  for (uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x; r < ix.bwtlen; r += (uint64_t)gridDim.x * 256) {
    const uint32_t g = off[row_seq[r]] + 1u + row_pos[r];
    sa_full[r] = g;
    text[g - 1u] = (uint8_t)symbol_at(ix, r);
  }
}

// Index load, indexes with 64-bit positions that leave room: the database text and the text position of every 2^tv_shift-th
// row (DevIndex::sa_tpos5), by walking every sequence from its terminator row (seq_walk_len / seq_walk_fill, kj_core.h).  A lane
// that has finished a sequence takes the next one (sequences are 30 .. 30 000 letters long: a wave of 64 whole walks would wait
// for its longest), one LF step per loop iteration.
__global__ void __launch_bounds__(256)
k_seq_walk_len(DevIndex ix, uint32_t *next, uint32_t *__restrict__ t_seq, uint32_t *__restrict__ len, uint32_t *bad) {
  uint64_t k = 0, n = 0;
  uint32_t t = 0;
  bool active = false;
  for (;;) {
    if (!active) {
      t = atomicAdd(next, 1u);
      if (t >= ix.nseq) break;
      k = t; n = 0; active = true;
    }
    const uint32_t c = symbol_at(ix, k);
    // (no sequence is longer than the text, and its length must fit the 32 bits of len[]: a damaged image ends here - bad is set
    //  below - instead of spinning for billions of LF steps)
    if (c == 0 || n >= 0xfffffff0ull || n > ix.bwtlen) {
      const uint32_t q = c == 0 ? (uint32_t)rank_term(ix, k) : 0xffffffffu;
      if (q >= ix.nseq) { atomicOr(bad, 1u); t_seq[t] = 0; }
      else { t_seq[t] = q; len[q] = (uint32_t)n; }
      active = false;
    } else { k = rank_c(ix, c, k); n++; }
  }
}
__global__ void __launch_bounds__(256)
k_seq_walk_fill(DevIndex ix, uint32_t *next, const uint32_t *__restrict__ t_seq, const uint32_t *__restrict__ len, const uint64_t *__restrict__ off,
                uint8_t *__restrict__ text, uint8_t *__restrict__ tpos5, uint32_t tv_shift,
                uint32_t *__restrict__ row_tax, const uint32_t *__restrict__ seq_dense) {
  // text / tpos5 (both or neither) and row_tax are optional: the walk fills what the index had room for
  const uint64_t tvm = (1ull << tv_shift) - 1ull;
  uint64_t k = 0, g = 0, left = 0;
  uint32_t dq = 0xffffffffu;
  bool active = false;
  for (;;) {
    if (!active) {
      const uint32_t t = atomicAdd(next, 1u);
      if (t >= ix.nseq) break;
      const uint32_t q = t_seq[t];
      k = t; g = off[(size_t)q + 1]; left = (uint64_t)len[q] + 1; active = true;
      if (row_tax) dq = seq_dense[q];
    }
    if (row_tax) row_tax[k] = dq;
    const uint32_t c = symbol_at(ix, k);
    if (text) {
      if ((k & tvm) == 0) put_tpos5(tpos5, k >> tv_shift, g);
      text[g - 1] = (uint8_t)c;
    }
    if (c == 0 || --left == 0) active = false;
    else { k = rank_c(ix, c, k); g--; }
  }
}

// the same for indexes with 64-bit positions: 16-byte entries {lo, len}, counts relative to mb_base
__global__ void __launch_bounds__(256)
k_kmer_extend_wide(const RankBlock64 *__restrict__ blk, const uint64_t *__restrict__ mb_base, uint32_t mb_shift,
                   const ulonglong2 *__restrict__ parent, ulonglong2 *__restrict__ child, uint64_t n_parent) {
  for (uint64_t idx = (uint64_t)blockIdx.x * 256 + threadIdx.x; idx < n_parent; idx += (uint64_t)gridDim.x * 256) {
    const ulonglong2 e = parent[idx];
    ulonglong2 *dst = child + idx * 20;
    if (e.y == 0) {
      for (int c = 0; c < 20; c++) dst[c] = make_ulonglong2(0, 0);
      continue;
    }
    const uint64_t lo = e.x, hi = e.x + e.y;
    const RankBlock64 &A = blk[lo >> 6], &B = blk[hi >> 6];
    const uint64_t *ma0 = mb_base + (lo >> mb_shift) * 20, *mb0 = mb_base + (hi >> mb_shift) * 20;
    const uint64_t lowA = (1ull << (lo & 63u)) - 1ull, lowB = (1ull << (hi & 63u)) - 1ull;
#pragma unroll
    for (int c = 1; c <= 20; c++) {
      uint64_t ma = lowA, mb = lowB;
#pragma unroll
      for (int bit = 0; bit < 5; bit++) {
        ma &= ((c >> bit) & 1) ? A.plane[bit] : ~A.plane[bit];
        mb &= ((c >> bit) & 1) ? B.plane[bit] : ~B.plane[bit];
      }
      const uint64_t ra = ma0[c - 1] + A.cnt[c - 1] + (uint64_t)__popcll(ma), rb = mb0[c - 1] + B.cnt[c - 1] + (uint64_t)__popcll(mb);
      dst[c - 1] = ra < rb ? make_ulonglong2(ra, rb - ra) : make_ulonglong2(0, 0);
    }
  }
}

static_assert(sizeof(ConstTables) % 4 == 0, "ConstTables is copied as dwords");
static_assert(sizeof(Hit) == sizeof(kaiju_gpu_hit), "hit layout");

// ----------------------------------------------------------------------------------------
// error handling
// ----------------------------------------------------------------------------------------
static thread_local std::string tl_error;
static int fail(int code, const std::string &msg) { tl_error = msg; return code; }
#define KJ_HIP(call)                                                                         \
  do {                                                                                       \
    hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess)                                                                    \
      return fail(KAIJU_GPU_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));     \
  } while (0)

// no exception may cross the C ABI (std::bad_alloc / length_error from a vector sized by a damaged file, ...)
template <class F>
static int guarded(F &&f) {
  try { return f(); }
  catch (const std::bad_alloc &) { return fail(KAIJU_GPU_ERR_NOMEM, "out of host memory"); }
  catch (const std::length_error &) { return fail(KAIJU_GPU_ERR_FORMAT, "a size in the file or an argument is not plausible"); }
  catch (const std::exception &e) { return fail(KAIJU_GPU_ERR_ARG, std::string("unexpected error: ") + e.what()); }
}

extern "C" int kaiju_gpu_abi_version(void) { return KAIJU_GPU_ABI_VERSION; }
extern "C" const char *kaiju_gpu_last_error(void) { return tl_error.c_str(); }
extern "C" const char *kaiju_gpu_strerror(int status) {
  switch (status) {
    case KAIJU_GPU_OK: return "ok";
    case KAIJU_GPU_ERR_ARG: return "bad argument";
    case KAIJU_GPU_ERR_IO: return "I/O error";
    case KAIJU_GPU_ERR_FORMAT: return "unrecognised file content";
    case KAIJU_GPU_ERR_NO_DEVICE: return "no usable HIP device (this library has no CPU path)";
    case KAIJU_GPU_ERR_HIP: return "HIP runtime error";
    case KAIJU_GPU_ERR_NOMEM: return "out of memory";
    case KAIJU_GPU_ERR_UNSUPPORTED: return "parameter not supported by the kernels";
    case KAIJU_GPU_ERR_INDEX_BUG: return "index triggers a latent bug of the reference";
    default: return "unknown status";
  }
}
extern "C" int kaiju_gpu_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// ----------------------------------------------------------------------------------------
// index
// ----------------------------------------------------------------------------------------
struct kaiju_gpu_index {
  int device = 0;
  DevIndex dev{};                 // device pointers
  ConstTables ct_host{};
  ConstTables *d_ct = nullptr;
  const Stage1Tables *d_s1 = nullptr;
  SegTables st{};                 // lnfact points to device memory
  double *d_lnfact = nullptr;
  std::vector<void *> allocs;
  kaiju_gpu_index_info info{};
  kaiju_gpu_index_footprint fp{};
  uint64_t tpos_bytes = 0;        // bytes of DevIndex::sa_tpos5 (fp.sa_full also counts the row -> taxon table of a wide index)
  std::vector<std::string> names;
  int id_mode = 0;                // KAIJU_GPU_IDS_TAXON / KAIJU_GPU_IDS_SEQUENCE
  ~kaiju_gpu_index() {
    (void)hipSetDevice(device);
    for (void *p : allocs) (void)hipFree(p);
  }
};

static double LoadClockNow() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
// KAIJU_GPU_LOAD_TIMES=1: wall time of the phases of an index load / a context creation, on stderr
struct LoadClock {
  bool on;
  double t0, tl;
  static double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
  LoadClock() : on(getenv("KAIJU_GPU_LOAD_TIMES") != nullptr), t0(now()), tl(t0) {}
  void mark(const char *what) {
    if (!on) return;
    const double t = now();
    fprintf(stderr, "[kaiju_gpu load] %-34s %8.1f ms (at %8.1f ms)\n", what, (t - tl) * 1e3, (t - t0) * 1e3);
    tl = t;
  }
};

template <class T, class A>
static int upload(kaiju_gpu_index *ix, const std::vector<T, A> &v, const T **dst) {
  void *p = nullptr;
  const size_t bytes = std::max<size_t>(v.size() * sizeof(T), 16) + 32;   // slack: 16-byte loads of 8-byte entries
  KJ_HIP(hipMalloc(&p, bytes));
  ix->allocs.push_back(p);
  if (!v.empty()) KJ_HIP(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  *dst = static_cast<const T *>(p);
  return 0;
}

// An array that was left in its image file (PackedIndex::lazy): file -> page-locked pieces -> HBM, two pieces in flight (the
// readers fill one while the other is on its way to the device), never a host copy of the whole array - a refseq-class image
// is 150 GB and eight ranks of a node load it side by side (the reference maps the whole .fmi into every process:
// readIndexes bwt/bwt.c:78-88).  KAIJU_GPU_STREAM_PIECE_MB sets the piece size (default 256; tests use 1).
struct ImageStreamer {
  int fd = -1;
  size_t piece = 0;
  void *buf[2] = {nullptr, nullptr};
  hipStream_t stream = nullptr;
  hipEvent_t done[2] = {nullptr, nullptr};
  uint64_t bytes_streamed = 0;
  double t_read = 0, t_total = 0;
  ~ImageStreamer() {
    if (fd >= 0) close(fd);
    for (int k = 0; k < 2; k++) { if (buf[k]) (void)hipHostFree(buf[k]); if (done[k]) (void)hipEventDestroy(done[k]); }
    if (stream) (void)hipStreamDestroy(stream);
  }
  int open_file(const std::string &path) {
    if (fd >= 0) return 0;
    fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) return fail(KAIJU_GPU_ERR_IO, "cannot open " + path);
    size_t mb = 256;
    if (const char *e = getenv("KAIJU_GPU_STREAM_PIECE_MB")) { const long v = atol(e); if (v >= 1 && v <= 4096) mb = (size_t)v; }
    piece = mb << 20;
    if (const char *e = getenv("KAIJU_GPU_STREAM_PIECE_KB")) { const long v = atol(e); if (v >= 4 && v <= (4096L << 10)) piece = (size_t)v << 10; }   // (tests)
    for (int k = 0; k < 2; k++) {
      KJ_HIP(hipHostMalloc(&buf[k], piece, hipHostMallocDefault));
      KJ_HIP(hipEventCreateWithFlags(&done[k], hipEventDisableTiming));
    }
    KJ_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    return 0;
  }
  // bytes [off, off + n) of the file to device memory at dst
  int run(uint64_t off, uint64_t n, void *dst) {
    const double t0 = LoadClockNow();
    const unsigned nthreads = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    bool used[2] = {false, false};
    uint64_t at = 0;
    for (int k = 0; at < n; k ^= 1) {
      const size_t len = (size_t)std::min<uint64_t>(piece, n - at);
      if (used[k]) KJ_HIP(hipEventSynchronize(done[k]));               // the copy out of this buffer has finished
      const double tr = LoadClockNow();
      std::atomic<bool> ok{true};
      const size_t sub = std::max<size_t>((len + nthreads - 1) / nthreads, std::min<size_t>(1u << 20, std::max<size_t>(piece / 4, 4096)));
      std::vector<std::thread> th;
      for (size_t b = 0; b < len; b += sub) {
        const size_t e = std::min(len, b + sub);
        th.emplace_back([&, b, e]() {
          size_t q = b;
          uint8_t *d = static_cast<uint8_t *>(buf[k]);
          while (q < e) { const ssize_t r = pread(fd, d + q, e - q, (off_t)(off + at + q)); if (r <= 0) { ok = false; return; } q += (size_t)r; }
        });
      }
      for (auto &x : th) x.join();
      t_read += LoadClockNow() - tr;
      if (!ok.load()) return fail(KAIJU_GPU_ERR_IO, "short read from the index image");
      KJ_HIP(hipMemcpyAsync(static_cast<uint8_t *>(dst) + at, buf[k], len, hipMemcpyHostToDevice, stream));
      KJ_HIP(hipEventRecord(done[k], stream));
      used[k] = true;
      at += len;
    }
    KJ_HIP(hipStreamSynchronize(stream));
    bytes_streamed += n;
    t_total += LoadClockNow() - t0;
    return 0;
  }
};

// a host vector, or - when it is empty and the image reader left the array in its file - the file's bytes
template <class T, class A>
static int upload_arr(kaiju_gpu_index *ix, ImageStreamer &is, const std::string &path, const std::vector<T, A> &v, const LazyArr &l, const T **dst) {
  if (!v.empty() || l.n == 0) return upload(ix, v, dst);
  int rc = is.open_file(path);
  if (rc) return rc;
  void *p = nullptr;
  KJ_HIP(hipMalloc(&p, l.n * sizeof(T) + 32));
  ix->allocs.push_back(p);
  if ((rc = is.run(l.off, l.n * sizeof(T), p))) return rc;
  *dst = static_cast<const T *>(p);
  return 0;
}

static int index_from_packed(PackedIndex &pk, int device_id, kaiju_gpu_index **out, bool keep_packed = false);
static thread_local int tl_id_mode = 0;     // kaiju_gpu_index_load_ex: 1 = hits collect sequence numbers

// The first HIP call of a process starts the runtime (about 0.15 s on an MI355X host); reading and packing an index needs
// no device, so the check for one runs beside it.  Its verdict still comes first: without a device the load fails with
// KAIJU_GPU_ERR_NO_DEVICE whatever the file looks like.
static std::future<int> device_check_async(int device_id) {
  return std::async(std::launch::async, [device_id]() -> int {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return KAIJU_GPU_ERR_NO_DEVICE;
    if (device_id < 0 || device_id >= ndev) return KAIJU_GPU_ERR_ARG;
    if (hipSetDevice(device_id) != hipSuccess || hipFree(nullptr) != hipSuccess) return KAIJU_GPU_ERR_HIP;
    return KAIJU_GPU_OK;
  });
}
static int device_check_result(std::future<int> &f) {
  const int rc = f.get();
  if (rc == KAIJU_GPU_ERR_NO_DEVICE) return fail(rc, "hipGetDeviceCount found no device");
  if (rc == KAIJU_GPU_ERR_ARG) return fail(rc, "device_id out of range");
  if (rc) return fail(rc, "the HIP runtime did not start on that device");
  return KAIJU_GPU_OK;
}

static int index_from_view(const HostIndexView &v, int device_id, kaiju_gpu_index **out, std::future<int> *dev_started = nullptr) {
  if (!out) return fail(KAIJU_GPU_ERR_ARG, "out is NULL");
  *out = nullptr;
  std::future<int> own;
  if (!dev_started) { own = device_check_async(device_id); dev_started = &own; }
  std::string msg;
  PackedIndex pk;
  LoadClock lc;
  int rc = pk.build(v, msg);
  lc.mark("pack (host)");
  const int drc = device_check_result(*dev_started);
  lc.mark("wait for the HIP runtime");
  if (drc) return drc;
  if (rc) return fail(rc, msg);
  return index_from_packed(pk, device_id, out);
}

// upload of the packed arrays (from a freshly packed .fmi or from an image file)
static int index_from_packed(PackedIndex &pk, int device_id, kaiju_gpu_index **out, bool keep_packed) {
  std::string msg;
  int rc;
  LoadClock lc;
  if (tl_id_mode == 1) pk.to_sequence_ids();
  std::unique_ptr<kaiju_gpu_index> ix(new kaiju_gpu_index());
  ix->device = device_id;
  ix->id_mode = tl_id_mode;
  rc = build_const_tables(pk.trans, ix->ct_host, msg);
  if (rc) return fail(rc, msg);
  std::vector<double> lnfact;
  rc = build_seg_tables(lnfact, ix->st, msg);
  if (rc) return fail(rc, msg);
  lc.mark("host tables");
  KJ_HIP(hipSetDevice(device_id));
  DevIndex &d = ix->dev;
  // what goes to HBM per index row: rank blocks 2 B, SA sample (e = 3) 0.5 B of sequence numbers and, on a
  // narrow index, 1 B of taxon ids
  ImageStreamer is;
  const std::string &ipath = pk.lazy.path;
  const bool streamed = !pk.stream.path.empty();       // a .fmi whose BWT and samples are still in the file (fmi_stream.h)
  const uint32_t *d_stream_sa_pos = nullptr;
  if ((rc = upload(ix.get(), pk.seq_taxid, &d.seq_taxid))) return rc;
  if ((rc = upload(ix.get(), pk.seq_valid, &d.seq_valid))) return rc;
  d.mb_base = nullptr; d.mb_shift = pk.mb_shift;
  d.sa_taxid = nullptr;
  if (streamed) {
    // file -> page-locked pieces -> HBM, packed by kernels: no host copy of bwt[] / sa[], no packed arrays on the host
    FmiStreamResult sr;
    if ((rc = fmi_stream_to_device(pk.stream, pk, d.seq_taxid, d.seq_valid, sr, msg))) return fail(rc, msg);
    for (void *p : {(void *)sr.blocks64, (void *)sr.mb_base, (void *)sr.sa_iseq, (void *)sr.sa_taxid, (void *)sr.term_pos}) if (p) ix->allocs.push_back(p);
    d.blocks64 = sr.blocks64; d.mb_base = sr.mb_base; d.sa_iseq = sr.sa_iseq; d.sa_taxid = sr.sa_taxid; d.term_pos = sr.term_pos;
    d_stream_sa_pos = sr.sa_pos;                       // (freed behind the text builder below)
    if (sr.sa_pos) ix->allocs.push_back(sr.sa_pos);
    memcpy(pk.C, sr.C, sizeof pk.C);
    if (sr.mb_base) {                                  // (a few KB; kept on the host for the footprint and kaiju_gpu_index_load_devices)
      pk.mb_base.resize((size_t)((pk.bwtlen >> pk.mb_shift) + 1) * 20);
      KJ_HIP(hipMemcpy(pk.mb_base.data(), sr.mb_base, pk.mb_base.size() * 8, hipMemcpyDeviceToHost));
    }
    if (lc.on)
      fprintf(stderr, "[kaiju_gpu load]   streamed from the .fmi: %.2f GB in %.2f s (%.1f GB/s; %.2f s of it reading pieces into page-locked "
                      "memory, pieces of %.1f MB), packed on the device\n", sr.bytes_streamed * 1e-9, sr.seconds, sr.bytes_streamed * 1e-9 / std::max(sr.seconds, 1e-9),
              sr.seconds_reading, sr.piece / 1048576.0);
  } else {
    if ((rc = upload_arr(ix.get(), is, ipath, pk.blocks64, pk.lazy.blocks64, &d.blocks64))) return rc;
    if (!pk.mb_base.empty() && (rc = upload(ix.get(), pk.mb_base, &d.mb_base))) return rc;
    if (!pk.sa_taxid.empty() && (rc = upload(ix.get(), pk.sa_taxid, &d.sa_taxid))) return rc;
    if ((rc = upload_arr(ix.get(), is, ipath, pk.sa_iseq, pk.lazy.sa_iseq, &d.sa_iseq))) return rc;
    if ((rc = upload_arr(ix.get(), is, ipath, pk.term_pos, pk.lazy.term_pos, &d.term_pos))) return rc;
  }
  const double *dl = nullptr;
  if ((rc = upload(ix.get(), lnfact, &dl))) return rc;
  ix->st.lnfact = dl;
  {
    std::vector<Stage1Tables> s1(1);
    build_stage1_tables(ix->ct_host, ix->st, s1[0]);
    const Stage1Tables *d1 = nullptr;
    if ((rc = upload(ix.get(), s1, &d1))) return rc;
    ix->d_s1 = d1;
  }
  std::vector<ConstTables> ctv(1, ix->ct_host);
  const ConstTables *dct = nullptr;
  if ((rc = upload(ix.get(), ctv, &dct))) return rc;
  ix->d_ct = const_cast<ConstTables *>(dct);
  for (int a = 0; a < 22; a++) d.C[a] = pk.C[a];
  d.bwtlen = pk.bwtlen; d.n_sa = pk.n_sa; d.sa_skip = pk.sa_skip; d.nseq = pk.nseq; d.chpt_exp = pk.chpt_exp;
  d.kmer32 = nullptr; d.kmer64 = nullptr; d.kmer_k = pk.kmer_k; d.kline = nullptr; d.kline_k = 0;
  const uint64_t n_kmer32 = PackedIndex::count(pk.kmer32, pk.lazy.kmer32), n_kmer64 = PackedIndex::count(pk.kmer64, pk.lazy.kmer64);
  uint64_t kmer_bytes = 0;
  // the depth the HOST packer builds (PackedIndex::build: 5, KAIJU_GPU_KMER up to 6); a streamed .fmi has no rank blocks on the
  // host: its table starts on the device with the twenty one-letter intervals of InitialSI (bwt.c:146-152) and grows from there
  constexpr uint32_t kKmerResidentK = 5;          // depth of the table a narrow index keeps next to its k-mer lines
  void *kmer5 = nullptr;
  // (the five-letter level is in nobody's list while the deeper levels grow: an early return below must not leak it)
  struct FreeOnExit { void *&p; ~FreeOnExit() { if (p) (void)hipFree(p); } } kmer5_guard{kmer5};
  uint32_t host_k = pk.kmer_k;
  if (streamed && pk.alen == 21) {
    host_k = 5;
    if (const char *e = getenv("KAIJU_GPU_KMER")) { host_k = (uint32_t)atoi(e); if (host_k > 6) host_k = 6; }
    if (host_k < 2) host_k = 0;
    if (host_k) {
      if (!d.mb_base) {
        std::vector<uint2> k1(20);
        for (uint32_t c = 1; c <= 20; c++) { const uint64_t len = pk.C[c + 1] - pk.C[c]; k1[c - 1] = uint2{len ? (uint32_t)pk.C[c] : 0u, (uint32_t)len}; }
        if ((rc = upload(ix.get(), k1, &d.kmer32))) return rc;
      } else {
        std::vector<ulonglong2> k1(20);
        for (uint32_t c = 1; c <= 20; c++) { const uint64_t len = pk.C[c + 1] - pk.C[c]; k1[c - 1] = ulonglong2{len ? pk.C[c] : 0ull, len}; }
        if ((rc = upload(ix.get(), k1, &d.kmer64))) return rc;
      }
      d.kmer_k = 1;
    }
  }
  if (host_k) {
    if (streamed) {}
    else if (n_kmer32) { if ((rc = upload_arr(ix.get(), is, ipath, pk.kmer32, pk.lazy.kmer32, &d.kmer32))) return rc; }
    else if ((rc = upload_arr(ix.get(), is, ipath, pk.kmer64, pk.lazy.kmer64, &d.kmer64))) return rc;
    lc.mark("upload of the packed arrays");
    if (is.bytes_streamed && lc.on)
      fprintf(stderr, "[kaiju_gpu load]   streamed from the image file: %.2f GB in %.2f s (%.1f GB/s; %.2f s of it reading pieces into "
                      "page-locked memory, pieces of %zu MB)\n", is.bytes_streamed * 1e-9, is.t_total, is.bytes_streamed * 1e-9 / std::max(is.t_total, 1e-9),
              is.t_read, is.piece >> 20);
    // Deeper tables are grown on the device, one letter at a time.  Depth: KAIJU_GPU_KMER, or the
    // largest k <= 7 whose table (20^k entries of 8 bytes) has at most 8 entries per index row -
    // 10 GB for a viruses-size index, which is what 288 GB of HBM are for.
    uint32_t want = host_k;
    if (const char *e = getenv("KAIJU_GPU_KMER")) want = (uint32_t)atoi(e);
    else { uint64_t nn = 1; for (uint32_t q = 0; q < host_k; q++) nn *= 20; while (want < 7 && nn * 20 <= 8 * pk.bwtlen) { nn *= 20; want++; } }
    if (want > 7) want = 7;
    if (d.kmer32 && d.blocks64 && want > d.kmer_k) {
      uint64_t np = 1;
      for (uint32_t q = 0; q < d.kmer_k; q++) np *= 20;
      const uint2 *cur = d.kmer32;
      void *cur_alloc = ix->allocs.back();
      while (d.kmer_k < want) {
        void *child = nullptr;
        if (hipMalloc(&child, np * 20 * sizeof(uint2) + 64) != hipSuccess) { (void)hipGetLastError(); break; }   // keep the table we have
        const uint64_t blocks = std::min<uint64_t>((np + 255) / 256, 1u << 20);
        hipLaunchKernelGGL(k_kmer_extend, dim3((unsigned)blocks), dim3(256), 0, 0, d.blocks64, cur, static_cast<uint2 *>(child), np);
        KJ_HIP(hipGetLastError());
        KJ_HIP(hipDeviceSynchronize());
        // (the five-letter level stays: what remains resident once the lines of the deeper table exist, see below)
        if (d.kmer_k == kKmerResidentK && !d.mb_base && !getenv("KAIJU_GPU_KEEP_KMER_TABLE")) kmer5 = cur_alloc;
        else (void)hipFree(cur_alloc);
        ix->allocs.back() = child;
        cur_alloc = child; cur = static_cast<const uint2 *>(child);
        np *= 20; d.kmer_k++;
      }
      d.kmer32 = cur;
      kmer_bytes = np * sizeof(uint2) - n_kmer32 * sizeof(uint2);
    }
    d.kline = nullptr;
    if (d.kmer32 && d.blocks64 && !d.mb_base && d.kmer_k >= 2) {
      // the k-mer lines the second-generation lanes read (128 bytes per (k-1)-letter word; k = 7: 8.2 GB)
      uint64_t nl = 1;
      for (uint32_t q = 1; q < d.kmer_k; q++) nl *= 20;
      void *lines = nullptr;
      if (hipMalloc(&lines, nl * kKLineBytes + 256) == hipSuccess) {
        ix->allocs.push_back(lines);
        const uint64_t blocks = std::min<uint64_t>((nl + 255) / 256, 1u << 20);
        hipLaunchKernelGGL(k_kline_build, dim3((unsigned)blocks), dim3(256), 0, 0, d, d.kmer_k, nl, static_cast<uint8_t *>(lines));
        KJ_HIP(hipGetLastError());
        KJ_HIP(hipDeviceSynchronize());
        d.kline = static_cast<const uint8_t *>(lines);
        d.kline_k = d.kmer_k;
        kmer_bytes += nl * kKLineBytes;
        if (kmer5) {
          // The second-generation lanes read the LINES only; the table itself (20^7 entries: 10.2 GB on a viruses-size index,
          // half of its footprint) served the first-generation lanes - verbose output, the retry pass - which are as exact
          // with the five-letter level: that one stays (25.6 MB), the deep table goes
          ix->allocs.pop_back();                                        // (lines)
          (void)hipFree(ix->allocs.back());                             // (the deep table)
          ix->allocs.back() = kmer5;
          ix->allocs.push_back(lines);
          d.kmer32 = static_cast<const uint2 *>(kmer5); d.kmer_k = kKmerResidentK;
          kmer5 = nullptr;
        }
      } else (void)hipGetLastError();      // (no room: the lanes of the first generation serve, with the table)
    }
    if (kmer5) { (void)hipFree(kmer5); kmer5 = nullptr; }               // (no lines were built: the deep table stays as it is)
    if (d.kmer64 && d.blocks64 && d.mb_base && want > d.kmer_k) {
      uint64_t np = 1;
      for (uint32_t q = 0; q < d.kmer_k; q++) np *= 20;
      const ulonglong2 *cur = d.kmer64;
      void *cur_alloc = ix->allocs.back();
      while (d.kmer_k < want) {
        void *child = nullptr;
        if (hipMalloc(&child, np * 20 * sizeof(ulonglong2) + 64) != hipSuccess) { (void)hipGetLastError(); break; }
        const uint64_t blocks = std::min<uint64_t>((np + 255) / 256, 1u << 20);
        hipLaunchKernelGGL(k_kmer_extend_wide, dim3((unsigned)blocks), dim3(256), 0, 0, d.blocks64, d.mb_base, d.mb_shift, cur,
                           static_cast<ulonglong2 *>(child), np);
        KJ_HIP(hipGetLastError());
        KJ_HIP(hipDeviceSynchronize());
        (void)hipFree(cur_alloc);
        ix->allocs.back() = child;
        cur_alloc = child; cur = static_cast<const ulonglong2 *>(child);
        np *= 20; d.kmer_k++;
      }
      d.kmer64 = cur;
      kmer_bytes = np * sizeof(ulonglong2) - n_kmer64 * sizeof(ulonglong2);
    }
  }
  lc.mark("k-mer table (device)");
  // ---- text verification: the database text (1 B per row), the full suffix array and the sequence of every row (4 B per row
  //      each: 9 bytes per row in all, + 4 B per sequence for the text offsets; narrow indexes with room for it).  Room = twice
  //      the peak of the build (two temporaries of 4 B per row next to the arrays) plus what the classification contexts of
  //      two streams allocate later (scratch of the search lanes, peptides, fragment lists: up to ~4 GB for 10 M-read batches) -
  //      an index that does not leave that much goes without the text arrays rather than failing in kaiju_gpu_create ----
  d.sa_full = nullptr; d.text = nullptr; d.row_tax = nullptr; d.tax_of_dense = nullptr; d.n_dense = 0;
  d.beyond_lo = d.beyond_n = d.beyond_row = 0;
  uint64_t text_bytes = 0;
  {
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    const uint64_t need_peak = pk.bwtlen * 13 + ((uint64_t)pk.nseq << 3) + (64u << 20) + (4ull << 30);   // 9 B per row kept + one temporary of 4 B per row + contexts
    const bool want = !getenv("KAIJU_GPU_NO_TEXT") && !d.mb_base && d.blocks64 && PackedIndex::count(pk.sa_pos, pk.lazy.sa_pos) &&
                      pk.bwtlen + pk.nseq + 4 * (uint64_t)kTextPad < 0xffffffffull && need_peak < free_b / 2;
    if (want) {
      const uint32_t *d_smp = d_stream_sa_pos;
      if (!d_smp && (rc = upload_arr(ix.get(), is, ipath, pk.sa_pos, pk.lazy.sa_pos, &d_smp))) return rc;
      void *smp_alloc = const_cast<uint32_t *>(d_smp);
      uint32_t *row_seq = nullptr, *row_pos = nullptr, *d_len = nullptr, *d_off = nullptr, *d_bad = nullptr, *sa_full = nullptr, *d_beyond = nullptr;
      uint8_t *text = nullptr;
      bool ok = hipMalloc((void **)&row_seq, pk.bwtlen * 4 + 16) == hipSuccess && hipMalloc((void **)&row_pos, pk.bwtlen * 4) == hipSuccess &&
                hipMalloc((void **)&d_len, (size_t)pk.nseq * 4 + 16) == hipSuccess && hipMalloc((void **)&d_off, (size_t)pk.nseq * 4 + 16) == hipSuccess &&
                hipMalloc((void **)&d_bad, 16) == hipSuccess && hipMalloc((void **)&d_beyond, (size_t)kBeyondRowsMax * 4) == hipSuccess;
      if (ok) {
        (void)hipMemset(d_bad, 0, 16);
        (void)hipMemset(d_len, 0, (size_t)pk.nseq * 4);
        const unsigned blocks = (unsigned)std::min<uint64_t>((pk.bwtlen + 255) / 256, 1u << 20);
        hipLaunchKernelGGL(k_suffix_walk, dim3(blocks), dim3(256), 0, 0, d, d_smp, row_seq, row_pos, d_bad, d_beyond);
        hipLaunchKernelGGL(k_seq_lens, dim3((pk.nseq + 255) / 256), dim3(256), 0, 0, pk.nseq, row_seq, row_pos, d_len);
        std::vector<uint32_t> len(pk.nseq), off(pk.nseq);
        uint32_t bad[2] = {0, 0};                           // [1]: rows behind the missing sample of a KAIJU_IDX_WARN_SA_SHORT index
        ok = hipMemcpy(len.data(), d_len, (size_t)pk.nseq * 4, hipMemcpyDeviceToHost) == hipSuccess &&
             hipMemcpy(bad, d_bad, 8, hipMemcpyDeviceToHost) == hipSuccess && bad[0] == 0 && bad[1] <= kBeyondRowsMax;
        uint64_t at = kTextPad;
        for (uint32_t q = 0; ok && q < pk.nseq; q++) { off[q] = (uint32_t)at; at += (uint64_t)len[q] + 1; if (at + kTextPad >= 0xffffffffull) ok = false; }
        text_bytes = at + 2 * kTextPad;
        ok = ok && hipMemcpy(d_off, off.data(), (size_t)pk.nseq * 4, hipMemcpyHostToDevice) == hipSuccess &&
             hipMalloc((void **)&sa_full, pk.bwtlen * 4 + 64) == hipSuccess && hipMalloc((void **)&text, text_bytes) == hipSuccess;
        if (ok) {
          (void)hipMemset(text, 0, text_bytes);
          hipLaunchKernelGGL(k_text_build, dim3(blocks), dim3(256), 0, 0, d, row_seq, row_pos, d_off, sa_full, text);
          if (bad[1]) hipLaunchKernelGGL(k_rows_beyond, dim3((bad[1] + 255) / 256), dim3(256), 0, 0, d_beyond, bad[1], row_seq);
          ok = hipDeviceSynchronize() == hipSuccess && hipGetLastError() == hipSuccess;
          if (ok && bad[1]) {
            // the rows behind the missing sample lie next to each other in the text (one walk: DevIndex::beyond_lo)
            std::vector<uint32_t> rows(bad[1]), tp(bad[1]);
            hipLaunchKernelGGL(k_gather_u32, dim3((bad[1] + 255) / 256), dim3(256), 0, 0, sa_full, d_beyond, bad[1], row_pos);   // (row_pos: free now)
            ok = hipMemcpy(tp.data(), row_pos, (size_t)bad[1] * 4, hipMemcpyDeviceToHost) == hipSuccess &&
                 hipMemcpy(rows.data(), d_beyond, (size_t)bad[1] * 4, hipMemcpyDeviceToHost) == hipSuccess;
            if (ok) {
              const uint32_t lo = *std::min_element(tp.begin(), tp.end()), hi = *std::max_element(tp.begin(), tp.end());
              ok = hi - lo + 1 == bad[1];
              d.beyond_lo = lo; d.beyond_n = bad[1]; d.beyond_row = rows[0];
            }
          }
          if (ok) {
            // row -> sequence -> dense taxon index (the locate's scan over the rows of a match: contiguous loads only)
            std::vector<uint32_t> seq_dense;
            std::vector<uint64_t> tax_of_dense;
            dense_taxa(pk.seq_taxid, pk.seq_valid, seq_dense, tax_of_dense);
            const uint32_t *d_sd = nullptr;
            const uint64_t *d_td = nullptr;
            if (upload(ix.get(), seq_dense, &d_sd) || upload(ix.get(), tax_of_dense, &d_td)) ok = false;
            else {
              hipLaunchKernelGGL(k_row_tax, dim3(blocks), dim3(256), 0, 0, row_seq, pk.bwtlen, d_sd, pk.nseq);
              ok = hipDeviceSynchronize() == hipSuccess && hipGetLastError() == hipSuccess;
              ix->allocs.pop_back();                           // (tax_of_dense: re-registered below when everything worked)
              ix->allocs.pop_back();
              (void)hipFree(const_cast<uint32_t *>(d_sd));
              if (ok) { d.tax_of_dense = d_td; d.n_dense = (uint32_t)tax_of_dense.size(); }
              else (void)hipFree(const_cast<uint64_t *>(d_td));
            }
          }
          if (!ok) { d.beyond_lo = d.beyond_n = d.beyond_row = 0; }
        }
      }
      (void)hipGetLastError();
      for (void *q : {(void *)row_pos, (void *)d_len, (void *)d_off, (void *)d_bad, (void *)d_beyond}) if (q) (void)hipFree(q);
      if (!ok && row_seq) { (void)hipFree(row_seq); row_seq = nullptr; }     // (kept otherwise: DevIndex::row_seq)
      // (the sample offsets were only needed here)
      (void)hipFree(smp_alloc);
      ix->allocs.erase(std::find(ix->allocs.begin(), ix->allocs.end(), smp_alloc));
      d_stream_sa_pos = nullptr;
      if (ok) {
        ix->allocs.push_back(sa_full); ix->allocs.push_back(text); ix->allocs.push_back(row_seq);
        ix->allocs.push_back(const_cast<uint64_t *>(d.tax_of_dense));
        d.sa_full = sa_full; d.text = text; d.row_tax = row_seq;
      }
      else { if (sa_full) (void)hipFree(sa_full); if (text) (void)hipFree(text); text_bytes = 0; d.tax_of_dense = nullptr; d.n_dense = 0; }
    }
  }
  if (d_stream_sa_pos) {                                  // (streamed .fmi without text arrays: the sample offsets are not needed)
    (void)hipFree(const_cast<uint32_t *>(d_stream_sa_pos));
    ix->allocs.erase(std::find(ix->allocs.begin(), ix->allocs.end(), (void *)d_stream_sa_pos));
    d_stream_sa_pos = nullptr;
  }
  // ---- the same for an index with 64-bit positions: the text (1 B per row) and the text position of every 2^tv_shift-th row
  //      (5 B each), the densest sample that - with the temporaries of the build and 8 GB for the classification contexts -
  //      fits in 60 % of the free HBM: every row at 4 G rows (26 GB).  Indexes of 2^34 rows and more only on request
  //      (KAIJU_GPU_TV_SHIFT=s forces a sample; every second row at refseq_ref's 28 G rows would be 98 GB next to the index's
  //      92 GB, nothing fits at refseq_nr's 58 G): measured on 4.35 G rows only (profiles/r04_wide_text) ----
  d.sa_tpos5 = nullptr; d.tv_shift = 0;
  uint64_t tpos_bytes = 0, rowtax_bytes = 0;
  // ---- and the row -> taxon table (4 B per row, DevIndex::row_tax): with it the ids of a match are contiguous loads, as on a
  //      narrow index, instead of a walk of up to 2^e LF steps per row - k_mem_locate_wide was the largest kernel of a step
  //      wherever matches hold many rows (a database of protein families at 4.5 G rows: 35 of 47 ms per 2 M reads; refseq_ref's
  //      28 G rows with every protein seven times: 25 of 64 ms).  It comes FIRST when HBM is short (the text arrays save 6 % of
  //      the search kernel, section 5b of DESIGN.md): built when it fits - with 8 GB for the classification contexts - in 70 % of
  //      the free HBM, i.e. up to about 30 G rows on a 288 GB MI355X (112 GB at 28 G rows next to the index's 92 GB);
  //      KAIJU_GPU_ROW_TAX=0 / 1 overrides.  Filled by the same walk of every sequence that writes the text ----
  // (not on an index with the reference's short sample array: a text-grown match is recorded through the row where the text took
  //  over, the lanes without the arrays record its own end row - and whether that row lies behind the missing sample would then
  //  depend on whether the arrays fit; the narrow lane applies the skip to the grown match itself, DevIndex::beyond_lo.  The
  //  row -> taxon table likewise: the rows behind the missing sample are skipped by the walk, and known only to it)
  if (d.mb_base && d.blocks64 && d.term_pos && !getenv("KAIJU_GPU_NO_TEXT") && pk.bwtlen + 4 * (uint64_t)kTextPad < kTposNone &&
      !(pk.warnings & KAIJU_IDX_WARN_SA_SHORT)) {
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    const uint64_t tb_est = pk.bwtlen + 3 * (uint64_t)kTextPad, tmp = (uint64_t)pk.nseq * 20 + 64;
    const uint64_t rt_est = pk.bwtlen * 4 + 64;
    bool rt = (double)(rt_est + tmp + (8ull << 30)) <= 0.7 * (double)free_b;
    if (const char *e = getenv("KAIJU_GPU_ROW_TAX")) rt = atoi(e) != 0;
    const double free_tv = (double)free_b - (rt ? (double)rt_est : 0.0);
    int tv = -1;
    if (const char *e = getenv("KAIJU_GPU_TV_SHIFT")) { const int v = atoi(e); if (v >= 0 && v <= 8) tv = v; }
    else if (pk.bwtlen < (1ull << 34))
      for (int v = 0; v <= 3 && tv < 0; v++)
        if ((double)(tb_est + ((pk.bwtlen >> v) + 1) * 5 + tmp + (8ull << 30)) <= 0.6 * free_tv) tv = v;
    if (tv >= 0 || rt) {
      uint32_t *t_seq = nullptr, *d_len = nullptr, *d_cnt = nullptr, *row_tax = nullptr;
      uint64_t *d_off = nullptr;
      uint8_t *text = nullptr, *tpos = nullptr;
      const uint32_t *d_sd = nullptr;
      const uint64_t *d_td = nullptr;
      if (tv >= 0) tpos_bytes = ((pk.bwtlen >> tv) + 1) * 5 + 16;
      bool ok = hipMalloc((void **)&t_seq, (size_t)pk.nseq * 4 + 16) == hipSuccess && hipMalloc((void **)&d_len, (size_t)pk.nseq * 4 + 16) == hipSuccess &&
                hipMalloc((void **)&d_off, ((size_t)pk.nseq + 1) * 8) == hipSuccess && hipMalloc((void **)&d_cnt, 16) == hipSuccess;
      if (ok) {
        (void)hipMemset(d_cnt, 0, 16);
        (void)hipMemset(d_len, 0, (size_t)pk.nseq * 4);
        int n_cu = 256;
        { int dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev); }
        const unsigned blocks = (unsigned)std::min<uint64_t>(((uint64_t)pk.nseq + 255) / 256, (uint64_t)n_cu * 8);
        hipLaunchKernelGGL(k_seq_walk_len, dim3(blocks), dim3(256), 0, 0, d, d_cnt, t_seq, d_len, d_cnt + 1);
        std::vector<uint32_t> len(pk.nseq);
        std::vector<uint64_t> off((size_t)pk.nseq + 1);
        uint32_t cnt[2] = {0, 0};
        ok = hipMemcpy(len.data(), d_len, (size_t)pk.nseq * 4, hipMemcpyDeviceToHost) == hipSuccess &&
             hipMemcpy(cnt, d_cnt, 8, hipMemcpyDeviceToHost) == hipSuccess && cnt[1] == 0;
        off[0] = kTextPad;
        for (uint32_t q = 0; q < pk.nseq; q++) off[(size_t)q + 1] = off[q] + len[q] + 1;
        // (every row lies on exactly one walk: the lengths add up to the rows of the index, or the index is damaged)
        ok = ok && off[pk.nseq] - kTextPad == pk.bwtlen &&
             hipMemcpy(d_off, off.data(), off.size() * 8, hipMemcpyHostToDevice) == hipSuccess;
        if (ok && tv >= 0) {
          text_bytes = off[pk.nseq] + 2 * kTextPad;
          // (no room after all, or a text beyond 40-bit positions: the table below is still built)
          if (!(text_bytes < kTposNone && hipMalloc((void **)&text, text_bytes) == hipSuccess && hipMalloc((void **)&tpos, tpos_bytes) == hipSuccess)) {
            (void)hipGetLastError();
            if (text) (void)hipFree(text);
            text = nullptr; tpos = nullptr; text_bytes = 0; tpos_bytes = 0; tv = -1;
          }
        }
        if (ok && rt) {
          std::vector<uint32_t> seq_dense;
          std::vector<uint64_t> tax_of_dense;
          dense_taxa(pk.seq_taxid, pk.seq_valid, seq_dense, tax_of_dense);
          rowtax_bytes = pk.bwtlen * 4 + 64;                   // (+ slack: the many-rows locate reads 16 bytes at a time)
          if (upload(ix.get(), seq_dense, &d_sd) == 0) { ix->allocs.pop_back(); } else d_sd = nullptr;
          if (d_sd && upload(ix.get(), tax_of_dense, &d_td) == 0) { ix->allocs.pop_back(); } else d_td = nullptr;
          if (!(d_sd && d_td && hipMalloc((void **)&row_tax, rowtax_bytes) == hipSuccess)) {
            (void)hipGetLastError();
            row_tax = nullptr; rowtax_bytes = 0; rt = false;
          } else { d.n_dense = (uint32_t)tax_of_dense.size(); }
        }
        if (ok && (text || row_tax)) {
          if (text) { (void)hipMemset(text, 0, text_bytes); (void)hipMemset(tpos, 0xff, tpos_bytes); }
          if (row_tax) (void)hipMemset(row_tax, 0xff, rowtax_bytes);
          (void)hipMemset(d_cnt, 0, 16);
          hipLaunchKernelGGL(k_seq_walk_fill, dim3(blocks), dim3(256), 0, 0, d, d_cnt, t_seq, d_len, d_off, text, tpos, (uint32_t)std::max(tv, 0), row_tax, d_sd);
          ok = hipDeviceSynchronize() == hipSuccess && hipGetLastError() == hipSuccess;
        }
      }
      (void)hipGetLastError();
      for (void *q : {(void *)t_seq, (void *)d_len, (void *)d_off, (void *)d_cnt, (void *)d_sd}) if (q) (void)hipFree(q);
      if (ok && text) { ix->allocs.push_back(text); ix->allocs.push_back(tpos); d.text = text; d.sa_tpos5 = tpos; d.tv_shift = (uint32_t)tv; }
      else { if (text) (void)hipFree(text); if (tpos) (void)hipFree(tpos); text_bytes = 0; tpos_bytes = 0; }
      if (ok && row_tax) { ix->allocs.push_back(row_tax); ix->allocs.push_back(const_cast<uint64_t *>(d_td)); d.row_tax = row_tax; d.tax_of_dense = d_td; }
      else { if (row_tax) (void)hipFree(row_tax); if (d_td) (void)hipFree(const_cast<uint64_t *>(d_td)); rowtax_bytes = 0; d.n_dense = 0; }
    }
    lc.mark("text + text positions, row -> taxon table (device, 64-bit rows)");
  } else
  lc.mark("text + full suffix array (device)");
  kaiju_gpu_index_info &inf = ix->info;
  memset(&inf, 0, sizeof inf);
  inf.bwtlen = (int64_t)pk.bwtlen; inf.nseq = (int32_t)pk.nseq; inf.alen = (int32_t)pk.alen;
  inf.chpt_exp = (int32_t)pk.chpt_exp;
  inf.db_length = (double)((int64_t)pk.bwtlen - (int64_t)pk.nseq);   // Config.cpp:20
  inf.device_bytes = pk.bytes() + kmer_bytes;
  {
    kaiju_gpu_index_footprint &f = ix->fp;
    f.rank_blocks = PackedIndex::count(pk.blocks64, pk.lazy.blocks64) * sizeof(RankBlock64);
    f.count_bases = pk.mb_base.size() * 8;
    f.sa_seq = PackedIndex::count(pk.sa_iseq, pk.lazy.sa_iseq) * 4;
    f.sa_taxid = PackedIndex::count(pk.sa_taxid, pk.lazy.sa_taxid) * 8;
    f.seq_tables = pk.seq_taxid.size() * 8 + pk.seq_valid.size() + PackedIndex::count(pk.term_pos, pk.lazy.term_pos) * 8;
    uint64_t nw = 1;
    for (uint32_t q = 0; q < d.kmer_k; q++) nw *= 20;
    f.kmer_table = d.kmer_k ? nw * (d.kmer64 ? sizeof(ulonglong2) : sizeof(uint2)) : 0;
    uint64_t nlw = 1;
    for (uint32_t q = 1; q < d.kline_k; q++) nlw *= 20;
    f.kmer_lines = d.kline ? nlw * kKLineBytes : 0;
    f.other = sizeof(ConstTables) + sizeof(Stage1Tables) + lnfact.size() * 8;
    f.text = d.text ? text_bytes : 0;
    ix->tpos_bytes = d.sa_tpos5 ? tpos_bytes : 0;
    f.sa_full = d.sa_full ? pk.bwtlen * 8 : (d.sa_tpos5 ? tpos_bytes : 0) + (d.mb_base && d.row_tax ? rowtax_bytes : 0);   // (+ the taxon of every row, DevIndex::row_tax; wide: the text positions + that table)
    f.total = f.rank_blocks + f.count_bases + f.sa_seq + f.sa_taxid + f.seq_tables + f.kmer_table + f.kmer_lines + f.other + f.text + f.sa_full;
    f.kmer_k = std::max(d.kmer_k, d.kline_k); f.wide = d.mb_base ? 1u : 0u;
    inf.device_bytes = f.total;
    if (getenv("KAIJU_GPU_LOAD_TIMES"))
      fprintf(stderr, "[kaiju_gpu load] HBM: rank blocks %.2f GB, count bases %.3f GB, SA sample %.2f (sequence numbers) + %.2f (taxon ids) GB, "
                      "sequence tables %.2f GB, k = %u table %.2f GB + lines %.2f GB, text %.2f GB + full suffix array %.2f GB; %.2f B per index row "
                      "without the k-mer tables\n",
              f.rank_blocks * 1e-9, f.count_bases * 1e-9, f.sa_seq * 1e-9, f.sa_taxid * 1e-9, f.seq_tables * 1e-9, f.kmer_k,
              f.kmer_table * 1e-9, f.kmer_lines * 1e-9, f.text * 1e-9, f.sa_full * 1e-9,
              (double)(f.total - f.kmer_table - f.kmer_lines) / (double)(pk.bwtlen ? pk.bwtlen : 1));
  }
  inf.warnings = pk.warnings;
  snprintf(inf.alphabet, sizeof inf.alphabet, "%s", pk.alphabet.c_str());
  if (keep_packed) ix->names = pk.names; else ix->names.swap(pk.names);    // (keep_packed: further GPUs get the same arrays)
  KJ_HIP(hipDeviceSynchronize());
  *out = ix.release();
  return KAIJU_GPU_OK;
}

// An image of 2 GB and more is streamed to the device (no host copy); a smaller one is read into host memory and uploaded
// from there - allocating the page-locked pieces costs more than streaming saves (viruses-size image, 0.5 GB: 256 against 79 ms,
// profiles/r04_cli).  KAIJU_GPU_IMAGE_HOST_COPY=1: never stream; a piece size in the environment (tests): always.
static bool image_wants_streaming(const char *path) {
  if (getenv("KAIJU_GPU_IMAGE_HOST_COPY")) return false;
  if (getenv("KAIJU_GPU_STREAM_PIECE_KB") || getenv("KAIJU_GPU_STREAM_PIECE_MB")) return true;
  struct stat st;
  return stat(path, &st) == 0 && (uint64_t)st.st_size >= (2ull << 30);
}

// A .fmi of 1 GiB and more goes to the device in pieces and is packed there (fmi_stream.h): no host copy of the file, no packed
// arrays on the host - the loader needs the names of the sequences and two page-locked pieces where PackedIndex::build needs
// twice the file.  KAIJU_GPU_FMI_STREAM=1 / 0: always / never (tests run both ways and compare the device arrays).
static bool fmi_wants_streaming(const char *path) {
  if (const char *e = getenv("KAIJU_GPU_FMI_STREAM")) return atoi(e) != 0;
  struct stat st;
  return stat(path, &st) == 0 && (uint64_t)st.st_size >= (1ull << 30);
}

// does the file start with the magic of an index image?
static bool is_image_file(const char *path) {
  char m[8] = {0};
  FILE *fp = fopen(path, "rb");
  if (!fp) return false;
  const bool ok = fread(m, 1, 8, fp) == 8 && memcmp(m, "KJGPUIM", 7) == 0;
  fclose(fp);
  return ok;
}

extern "C" int kaiju_gpu_index_write_image(const char *fmi_path, const char *image_path) {
  return guarded([&]() -> int {
  if (!fmi_path || !image_path) return fail(KAIJU_GPU_ERR_ARG, "NULL argument");
  std::unique_ptr<FmiFile> f(new FmiFile());
  std::string msg;
  LoadClock lc;
  int rc = f->load(fmi_path, msg);
  lc.mark("read .fmi file");
  if (rc) return fail(rc, msg);
  PackedIndex pk;
  if ((rc = pk.build(f->view(), msg))) return fail(rc, msg);
  f.reset();                        // (a refseq-class .fmi is 60 GB in memory: gone before the image, as large again, is written)
  lc.mark("pack (host)");
  { struct stat st; if (stat(fmi_path, &st) == 0) pk.src_fmi_bytes = (uint64_t)st.st_size; }
  if ((rc = pk.write_image(image_path, msg))) return fail(rc, msg);
  lc.mark("write image file");
  return KAIJU_GPU_OK;
  });
}

extern "C" int kaiju_gpu_index_image_source_bytes(const char *image_path, uint64_t *fmi_bytes) {
  return guarded([&]() -> int {
  if (!image_path || !fmi_bytes) return fail(KAIJU_GPU_ERR_ARG, "NULL argument");
  std::string msg;
  const int rc = PackedIndex::image_source_bytes(image_path, *fmi_bytes, msg);
  return rc ? fail(rc, msg) : KAIJU_GPU_OK;
  });
}

// Host only: what an image file holds, read the way the loader reads it (header and the small arrays; the arrays that grow with
// the index are only located) - a caller can check an image and see how many bytes a load streams to the device without a GPU.
extern "C" int kaiju_gpu_index_image_info(const char *image_path, kaiju_gpu_index_info *info, uint64_t *streamed_bytes) {
  return guarded([&]() -> int {
  if (!image_path || !info) return fail(KAIJU_GPU_ERR_ARG, "NULL argument");
  if (!is_image_file(image_path)) return fail(KAIJU_GPU_ERR_FORMAT, "not a kaiju GPU index image");
  PackedIndex pk;
  std::string msg;
  const int rc = pk.read_image(image_path, msg, true);
  if (rc) return fail(rc, msg);
  memset(info, 0, sizeof *info);
  info->bwtlen = (int64_t)pk.bwtlen; info->nseq = (int32_t)pk.nseq; info->alen = (int32_t)pk.alen; info->chpt_exp = (int32_t)pk.chpt_exp;
  info->db_length = (double)((int64_t)pk.bwtlen - (int64_t)pk.nseq);
  info->device_bytes = pk.bytes();
  info->warnings = pk.warnings;
  snprintf(info->alphabet, sizeof info->alphabet, "%s", pk.alphabet.c_str());
  if (streamed_bytes) {
    const ImageLazy &l = pk.lazy;
    *streamed_bytes = l.blocks64.n * sizeof(RankBlock64) + l.sa_iseq.n * 4 + l.sa_pos.n * 4 + l.term_pos.n * 8 + l.kmer32.n * sizeof(uint2) +
                      l.kmer64.n * sizeof(ulonglong2);
  }
  return KAIJU_GPU_OK;
  });
}

extern "C" int kaiju_gpu_index_load(const char *fmi_path, int device_id, kaiju_gpu_index **out) {
  return guarded([&]() -> int {
  if (!fmi_path || !out) return fail(KAIJU_GPU_ERR_ARG, "NULL argument");
  *out = nullptr;
  std::future<int> dev = device_check_async(device_id);
  if (is_image_file(fmi_path)) {
    // a pre-packed image (kaiju_gpu_index_write_image): no parsing, no packing
    PackedIndex pk;
    std::string msg;
    LoadClock lc;
    // the arrays that grow with the index are streamed from the file to the device (kaijux ids rewrite the sampled ids on the
    // host: that mode reads everything; KAIJU_GPU_IMAGE_HOST_COPY=1 does so too, for comparison)
    const bool lazy = tl_id_mode == 0 && image_wants_streaming(fmi_path);
    const int rc = pk.read_image(fmi_path, msg, lazy);
    lc.mark(lazy ? "read image file (small arrays)" : "read image file");
    const int drc = device_check_result(dev);
    lc.mark("wait for the HIP runtime");
    if (drc) return drc;
    if (rc) return fail(rc, msg);
    return index_from_packed(pk, device_id, out);
  }
  FmiFile f;
  std::string msg;
  LoadClock lc;
  const bool streamed = fmi_wants_streaming(fmi_path);
  int rc = f.load(fmi_path, msg, streamed);
  lc.mark(streamed ? "read .fmi file (headers, names)" : "read .fmi file");
  if (rc) { const int drc = device_check_result(dev); return drc ? drc : fail(rc, msg); }
  if (!streamed) return index_from_view(f.view(), device_id, out, &dev);
  // the BWT and the sampled suffix array stay in the file: they are streamed to the device and packed there (fmi_stream.h)
  PackedIndex pk;
  rc = pk.build_streamed(f, fmi_path, msg);
  { std::vector<std::string>().swap(f.ids); std::vector<const char *>().swap(f.id_ptrs); }    // (pk has its own copy of the names)
  lc.mark("names, taxon ids (host)");
  const int drc = device_check_result(dev);
  lc.mark("wait for the HIP runtime");
  if (drc) return drc;
  if (rc) return fail(rc, msg);
  return index_from_packed(pk, device_id, out);
  });
}

extern "C" int kaiju_gpu_index_load_ex(const char *fmi_path, int device_id, int id_mode, kaiju_gpu_index **out) {
  if (id_mode != KAIJU_GPU_IDS_TAXON && id_mode != KAIJU_GPU_IDS_SEQUENCE) return fail(KAIJU_GPU_ERR_ARG, "id_mode");
  tl_id_mode = id_mode;
  const int rc = kaiju_gpu_index_load(fmi_path, device_id, out);
  tl_id_mode = 0;
  return rc;
}

// One index, several GPUs of a node: the file is parsed and packed (or its image read) ONCE, the packed arrays go to every
// device in turn; each device grows its own k-mer table, lines and text arrays.  out[k] belongs to devices[k].
extern "C" int kaiju_gpu_index_load_devices(const char *fmi_path, const int *devices, int n_devices, int id_mode, kaiju_gpu_index **out) {
  return guarded([&]() -> int {
  if (!fmi_path || !devices || !out || n_devices < 1) return fail(KAIJU_GPU_ERR_ARG, "bad argument");
  if (id_mode != KAIJU_GPU_IDS_TAXON && id_mode != KAIJU_GPU_IDS_SEQUENCE) return fail(KAIJU_GPU_ERR_ARG, "id_mode");
  for (int k = 0; k < n_devices; k++) out[k] = nullptr;
  std::future<int> dev = device_check_async(devices[0]);
  PackedIndex pk;
  std::string msg;
  int rc;
  if (is_image_file(fmi_path)) rc = pk.read_image(fmi_path, msg, id_mode == 0 && image_wants_streaming(fmi_path));
  else {
    FmiFile f;
    const bool streamed = fmi_wants_streaming(fmi_path);      // (every device then streams the file for itself: it is in the page cache)
    rc = f.load(fmi_path, msg, streamed);
    if (rc == 0) rc = streamed ? pk.build_streamed(f, fmi_path, msg) : pk.build(f.view(), msg);
  }
  const int drc = device_check_result(dev);
  if (drc) return drc;
  if (rc) return fail(rc, msg);
  tl_id_mode = id_mode;
  for (int k = 0; k < n_devices && rc == 0; k++) rc = index_from_packed(pk, devices[k], &out[k], k + 1 < n_devices);
  tl_id_mode = 0;
  if (rc) for (int k = 0; k < n_devices; k++) { delete out[k]; out[k] = nullptr; }
  return rc;
  });
}

extern "C" int kaiju_gpu_index_from_host(const kaiju_gpu_host_index *hv, int device_id, kaiju_gpu_index **out) {
  return guarded([&]() -> int {
  if (!hv || !out) return fail(KAIJU_GPU_ERR_ARG, "NULL argument");
  HostIndexView v;
  v.bwtlen = hv->bwtlen; v.nseq = hv->nseq; v.alen = hv->alen; v.alphabet = hv->alphabet; v.bwt = hv->bwt;
  v.startLcode = hv->startLcode; v.sa = hv->sa; v.ncheck = hv->ncheck; v.chpt_exp = hv->chpt_exp;
  v.nbytes = hv->nbytes; v.pbits = hv->pbits; v.ids = hv->ids;
  return index_from_view(v, device_id, out);
  });
}

extern "C" int kaiju_gpu_index_get_info(const kaiju_gpu_index *ix, kaiju_gpu_index_info *info) {
  if (!ix || !info) return fail(KAIJU_GPU_ERR_ARG, "NULL argument");
  *info = ix->info;
  return KAIJU_GPU_OK;
}
extern "C" int kaiju_gpu_index_get_footprint(const kaiju_gpu_index *ix, kaiju_gpu_index_footprint *out) {
  if (!ix || !out) return fail(KAIJU_GPU_ERR_ARG, "NULL argument");
  *out = ix->fp;
  return KAIJU_GPU_OK;
}
extern "C" void kaiju_gpu_index_free(kaiju_gpu_index *ix) { delete ix; }

// digest of `n` bytes: the sum over all 16-byte chunks of a mix of (chunk number, content); the last chunk is zero-padded
__global__ void __launch_bounds__(256)
k_digest(const uint8_t *__restrict__ p, uint64_t n, unsigned long long *out) {
  const uint64_t chunks = (n + 15) >> 4;
  uint64_t acc = 0;
  for (uint64_t c = (uint64_t)blockIdx.x * 256 + threadIdx.x; c < chunks; c += (uint64_t)gridDim.x * 256) {
    uint32_t w[4] = {0, 0, 0, 0};
    if ((c << 4) + 16 <= n && (reinterpret_cast<uintptr_t>(p) & 15u) == 0) {
      const uint4 v = reinterpret_cast<const uint4 *>(p)[c];
      w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
    } else {
      for (uint32_t b = 0; b < 16 && (c << 4) + b < n; b++) w[b >> 2] |= (uint32_t)p[(c << 4) + b] << (8 * (b & 3));
    }
    uint64_t h = (c + 1) * 0x9E3779B97F4A7C15ull;
#pragma unroll
    for (int q = 0; q < 4; q++) { h ^= w[q]; h *= 0xD6E8FEB86659FD93ull; h ^= h >> 29; }
    acc += h;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
  if ((threadIdx.x & 63u) == 0) atomicAdd(out, (unsigned long long)acc);
}

extern "C" int kaiju_gpu_index_digest(const kaiju_gpu_index *ix, uint64_t *out, uint32_t n_out) {
  return guarded([&]() -> int {
  if (!ix || !out || n_out < KAIJU_GPU_N_DIGESTS) return fail(KAIJU_GPU_ERR_ARG, "bad argument");
  KJ_HIP(hipSetDevice(ix->device));
  const DevIndex &d = ix->dev;
  uint64_t nw = 1, nlw = 1;
  for (uint32_t q = 0; q < d.kmer_k; q++) nw *= 20;
  for (uint32_t q = 1; q < d.kline_k; q++) nlw *= 20;
  struct Arr { const void *p; uint64_t bytes; };
  const Arr arrs[12] = {
      {d.blocks64, ((d.bwtlen >> 6) + 1) * sizeof(RankBlock64)},
      {d.mb_base, d.mb_base ? ((d.bwtlen >> d.mb_shift) + 1) * 20 * 8 : 0},
      {d.sa_iseq, d.n_sa * 4},
      {d.sa_taxid, d.sa_taxid ? (d.n_sa + 2) * 8 : 0},
      {d.term_pos, (uint64_t)d.nseq * 8},
      {d.seq_taxid, (uint64_t)d.nseq * 8},
      {d.seq_valid, (uint64_t)d.nseq},
      {d.kmer32 ? (const void *)d.kmer32 : (const void *)d.kmer64, d.kmer_k ? nw * (d.kmer32 ? sizeof(uint2) : sizeof(ulonglong2)) : 0},
      {d.kline, d.kline ? nlw * kKLineBytes : 0},
      {d.text, d.text ? ix->fp.text : 0},
      {d.sa_full ? (const void *)d.sa_full : (const void *)d.sa_tpos5, d.sa_full ? d.bwtlen * 4 : d.sa_tpos5 ? ix->tpos_bytes : 0},
      {d.row_tax, d.row_tax ? d.bwtlen * 4 : 0}};
  unsigned long long *acc = nullptr;
  KJ_HIP(hipMalloc((void **)&acc, 12 * 8));
  hipError_t e = hipMemset(acc, 0, 12 * 8);
  for (int a = 0; a < 12 && e == hipSuccess; a++) {
    if (!arrs[a].p || !arrs[a].bytes) continue;
    const uint64_t chunks = (arrs[a].bytes + 15) >> 4;
    hipLaunchKernelGGL(k_digest, dim3((unsigned)std::min<uint64_t>((chunks + 255) / 256, 1u << 16)), dim3(256), 0, 0,
                       static_cast<const uint8_t *>(arrs[a].p), arrs[a].bytes, acc + a);
    e = hipGetLastError();
  }
  unsigned long long h[12] = {0};
  if (e == hipSuccess) e = hipMemcpy(h, acc, sizeof h, hipMemcpyDeviceToHost);
  (void)hipFree(acc);
  if (e != hipSuccess) return fail(KAIJU_GPU_ERR_HIP, hipGetErrorString(e));
  for (int a = 0; a < 12; a++) out[a] = h[a];
  out[12] = d.kmer_k | (uint64_t)d.kline_k << 8;
  uint64_t hc = 0;
  for (int a = 0; a < 22; a++) hc = (hc ^ d.C[a]) * 0xD6E8FEB86659FD93ull + 1;
  out[13] = hc;
  return KAIJU_GPU_OK;
  });
}

struct kaiju_gpu_taxonomy {
  int device = 0;
  DevTaxonomy dev{};
  std::vector<void *> allocs;
  ~kaiju_gpu_taxonomy() {
    (void)hipSetDevice(device);
    for (void *p : allocs) (void)hipFree(p);
  }
};

__global__ void __launch_bounds__(256)
k_lca(DevTaxonomy t, const Hit *__restrict__ hits, uint32_t n, CompactHit *__restrict__ out) {
  const uint32_t r = blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  out[r] = compact_hit(t, hits[r]);
}

// ----------------------------------------------------------------------------------------
// context
// ----------------------------------------------------------------------------------------
struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
};

struct kaiju_gpu_ctx {
  const kaiju_gpu_index *ix = nullptr;
  kaiju_gpu_params params{};
  Params kp{};
  hipStream_t stream = nullptr;
  hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  bool ev_valid = false;
  int n_cu = 0, blocks_main = 0, blocks_retry = 0;
  DevBuf pep, frags, meta, counters, retry_list, seg_items, seg_recs;
  DevBuf scratch_main[10], scratch_retry[5], h_compact;
  DevBuf redo_bitmap, redo_list, redo_items, redo_index, redo_pool, redo_work, redo_cls;    // the exact pass
  bool greedy2 = false;
  bool greedy3 = false;            // KAIJU_GPU_GREEDY_LANE=v3: the row-pool lane (kj_greedy3.h; narrow index with k-mer lines)
  uint32_t g3_split = 1;           // KAIJU_GPU_G3_SPLIT
  uint32_t g3_threads = 512;       // KAIJU_GPU_G3_THREADS (a multiple of 64)
  uint32_t g1_pool = 192, g1_match = 64;   // first-generation Greedy lanes: queue slots and match records per lane of the main pass
  bool g1_pool_set = false;                // (KAIJU_GPU_G1_POOL given: also for -v)
  uint32_t greedy_gate = 1u | 32u << 8;   // heavy iteration every 2nd, or as soon as half the wavefront waits for one (measured: r02_gprof; round 3,
                                           // with the span rule and the probes thinning the fast iterations: every 2nd beats every 4th, profiles/r03_l14)
  bool verbose = false;            // kaiju_gpu_classify_batch_verbose: first-generation lanes + columns 6/7
  bool exact_pass = true;          // KAIJU_GPU_EXACT_PASS=0 switches the exact pass off (its reads stay flagged)
  bool count_ops = false;          // kaiju_gpu_set_count_ops: the main pass runs the counting instantiation of its lane
  bool mem_v1 = false;             // KAIJU_GPU_MEM_LANE=v1 (read once, at context creation)
  bool verbose_v1 = false;         // KAIJU_GPU_VERBOSE_LANE=v1: -v from the first-generation lanes (until round 6 the only way; A/B)
  bool stage1_old = false;         // KAIJU_GPU_STAGE1=old: build_fragments for every read length (A/B measurements)
  bool lazy_seg = true;            // KAIJU_GPU_LAZY_SEG=0: SEG pass over every flagged fragment in MEM mode too
  DevBuf seglist, loc_list, todo_list;
  int seg_team = 64;               // KAIJU_GPU_SEG_TEAM: lanes per fragment of the SEG pass (64 = one wavefront per fragment, k_seg; 8, 16, 32: k_seg_teams, slower - DESIGN.md 6b)
  bool fused_post = true;          // KAIJU_GPU_FUSED_POST=0: k_trigcheck / k_mem_locate / k_lca as separate passes (A/B measurements)
  const char *dump_frags = nullptr;// KAIJU_GPU_DUMP_FRAGS (developer aid; read once)
  uint32_t vb_text_cap = 0;
  DevBuf vb_nacc, vb_acc, vb_tlen, vb_text, vb_bestv, vb_bestv_retry;
  DevBuf vb_packed, vb_pos;          // column 7 packed for the way to the host (k_vb_pack)
  std::vector<uint8_t> vb_host;      // ... and where it arrives
  std::vector<uint32_t> vb_h_nacc, vb_h_acc;   // (host side of the accessions: kept, so that a call does not fault 80 bytes per read in again)
  DevBuf h_seqs, h_off, h_hits;      // staging for the host-buffer entry point
  kaiju_gpu_stats stats{};
  uint32_t last_n = 0;
  uint32_t max_read_len = 1024;
  ~kaiju_gpu_ctx() {
    if (!ix) return;
    (void)hipSetDevice(ix->device);
    DevBuf *all[] = {&pep, &frags, &meta, &counters, &retry_list, &seg_items, &seg_recs, &h_seqs, &h_off, &h_hits, &h_compact, &seglist, &loc_list, &todo_list,
                     &vb_nacc, &vb_acc, &vb_tlen, &vb_text, &vb_bestv, &vb_bestv_retry, &vb_packed, &vb_pos,
                     &redo_bitmap, &redo_list, &redo_items, &redo_index, &redo_pool, &redo_work, &redo_cls};
    for (DevBuf *b : all) if (b->p) (void)hipFree(b->p);
    for (int i = 0; i < 10; i++) if (scratch_main[i].p) (void)hipFree(scratch_main[i].p);
    for (int i = 0; i < 5; i++) if (scratch_retry[i].p) (void)hipFree(scratch_retry[i].p);
    for (auto &e : ev) if (e) (void)hipEventDestroy(e);
    if (stream) (void)hipStreamDestroy(stream);
  }
};

static int ensure(DevBuf &b, size_t bytes) {
  if (bytes <= b.cap) return 0;
  if (b.p) { KJ_HIP(hipFree(b.p)); b.p = nullptr; b.cap = 0; }
  const size_t want = bytes + bytes / 8 + 256;
  KJ_HIP(hipMalloc(&b.p, want));
  b.cap = want;
  return 0;
}

extern "C" void kaiju_gpu_default_params(kaiju_gpu_params *p, int mode) {
  if (!p) return;
  p->mode = mode ? 1 : 0;                 // Config.hpp:33-48
  p->min_fragment_length = 11;
  p->mismatches = 3;
  p->min_score = 65;
  p->seed_length = 7;
  p->seg = 1;
  p->use_evalue = mode ? 1 : 0;           // "-a mem" clears use_Evalue, kaiju.cpp:77-80
  p->min_evalue = 0.01;
  p->max_matches_SI = 20;
  p->max_match_ids = 20;
  p->input_is_protein = 0;
}

extern "C" int kaiju_gpu_create(kaiju_gpu_ctx **out, const kaiju_gpu_index *ix, const kaiju_gpu_params *p) {
  return guarded([&]() -> int {
  if (!out || !ix || !p) return fail(KAIJU_GPU_ERR_ARG, "NULL argument");
  *out = nullptr;
  if (p->mode != 0 && p->mode != 1) return fail(KAIJU_GPU_ERR_ARG, "mode must be 0 (MEM) or 1 (GREEDY)");
  if (p->min_fragment_length < 1 || p->min_fragment_length > 10000)
    return fail(KAIJU_GPU_ERR_UNSUPPORTED, "min_fragment_length out of range");
  if (p->mode == 1 && p->mismatches > (uint32_t)kMaxMismatch)
    return fail(KAIJU_GPU_ERR_UNSUPPORTED, "more than 8 mismatches are not supported");
  if (p->mode == 1 && p->seed_length < 1) return fail(KAIJU_GPU_ERR_ARG, "seed_length must be >= 1");
  if (p->max_match_ids > 20 || p->max_matches_SI > 64 || p->max_matches_SI < 1)
    return fail(KAIJU_GPU_ERR_UNSUPPORTED, "max_match_ids <= 20 and 1 <= max_matches_SI <= 64 required");
  std::unique_ptr<kaiju_gpu_ctx> c(new kaiju_gpu_ctx());
  c->ix = ix;
  c->params = *p;
  c->kp.mode = p->mode; c->kp.m = p->min_fragment_length; c->kp.mismatches = p->mismatches;
  c->kp.min_score = p->min_score; c->kp.seed_length = p->seed_length; c->kp.seg = p->seg ? 1 : 0;
  c->kp.max_matches_SI = p->max_matches_SI; c->kp.max_match_ids = p->max_match_ids;
  if (const char *e = getenv("KAIJU_GPU_DEBUG")) c->kp.debug = (uint32_t)atoi(e);
  if (const char *e = getenv("KAIJU_GPU_EXACT_PASS")) c->exact_pass = atoi(e) != 0;
  if (const char *e = getenv("KAIJU_GPU_MEM_LANE")) c->mem_v1 = !strcmp(e, "v1");
  if (const char *e = getenv("KAIJU_GPU_VERBOSE_LANE")) c->verbose_v1 = !strcmp(e, "v1");
  c->dump_frags = getenv("KAIJU_GPU_DUMP_FRAGS");
  if (const char *e = getenv("KAIJU_GPU_STAGE1")) c->stage1_old = !strcmp(e, "old");
  if (const char *e = getenv("KAIJU_GPU_LAZY_SEG")) c->lazy_seg = atoi(e) != 0;
  if (const char *e = getenv("KAIJU_GPU_FUSED_POST")) c->fused_post = atoi(e) != 0;
  if (const char *e = getenv("KAIJU_GPU_SEG_TEAM")) { const int v = atoi(e); if (v == 8 || v == 16 || v == 32 || v == 64) c->seg_team = v; }
  // kaijux: the MEM search of ConsumerThreadx.cpp:135 (maxMatches(.., 1)) finds the same longest matches as
  // greedyExact but lists them in another order, which shows where the id cap cuts and in the peptides of -v
  if (ix->id_mode == KAIJU_GPU_IDS_SEQUENCE && p->mode == 0) c->kp.flags |= kParamXOrder;
  if (p->input_is_protein) c->kp.flags |= kParamProtein;
  KJ_HIP(hipSetDevice(ix->device));
  hipDeviceProp_t prop;
  KJ_HIP(hipGetDeviceProperties(&prop, ix->device));
  c->n_cu = prop.multiProcessorCount;
  int occ = 0;
  if (p->mode == 0) KJ_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_mem, kBlock, 0));
  else {
    const bool g_wide = ix->dev.mb_base != nullptr;
    const uint32_t g_k = g_wide ? ix->dev.kmer_k : ix->dev.kline_k;
    c->greedy2 = ix->dev.blocks64 && (g_wide ? ix->dev.kmer64 != nullptr : ix->dev.kline != nullptr) && g_k >= 2 &&
                 g_k <= p->seed_length && p->seed_length >= 3;
    if (const char *e = getenv("KAIJU_GPU_GREEDY_LANE")) { if (!strcmp(e, "v1")) c->greedy2 = false; }
    // (the row-pool lane is parity-green but SLOWER than greedy_lane2 at the 480 rows a CU's LDS holds - DESIGN.md 6b, round 6:
    //  opt-in, KAIJU_GPU_GREEDY_LANE=v3)
#ifdef KJ_GREEDY3
    if (const char *e = getenv("KAIJU_GPU_GREEDY_LANE")) { if (!strcmp(e, "v3")) c->greedy3 = c->greedy2 && !g_wide; }
    if (const char *e = getenv("KAIJU_GPU_G3_SPLIT")) c->g3_split = (uint32_t)atoi(e);
    if (const char *e = getenv("KAIJU_GPU_G3_THREADS")) { const int v = atoi(e); if (v >= 64 && v <= kG3Threads && v % 64 == 0) c->g3_threads = (uint32_t)v; }
#else
    if (const char *e = getenv("KAIJU_GPU_GREEDY_LANE")) {
      if (!strcmp(e, "v3")) return fail(KAIJU_GPU_ERR_UNSUPPORTED, "KAIJU_GPU_GREEDY_LANE=v3: this library was built without -DKJ_GREEDY3 (tests/tools/mem_variants.sh, variant g3)");
    }
#endif
    if (const char *e = getenv("KAIJU_GPU_GREEDY_GATE")) { int v = atoi(e); if (v == 0 || v == 1 || v == 3 || v == 7 || v == 15) c->greedy_gate = (c->greedy_gate & ~0xffu) | (uint32_t)v; }
    // (bits 8..: heavy iteration as soon as that many lanes of the wavefront wait for the slow part; 0 = period only)
    if (const char *e = getenv("KAIJU_GPU_GREEDY_WAITERS")) { int v = atoi(e); if (v >= 0 && v <= 64) c->greedy_gate = (c->greedy_gate & 0xffu) | (uint32_t)v << 8; }
    if (c->greedy2) {
      KJ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_greedy2), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)kGreedy2Lds));
      KJ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_greedy2_count), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)kGreedy2Lds));
      KJ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_greedy2_wide), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)kGreedy2Lds));
      KJ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_greedy2_wide_count), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)kGreedy2Lds));
      KJ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_greedy2_vb), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGreedy2Lds));
      KJ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_greedy2_wide_vb), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGreedy2Lds));
      if (g_wide) KJ_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_greedy2_wide, kBlock, kGreedy2Lds));
      else KJ_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_greedy2, kBlock, kGreedy2Lds));
    } else KJ_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_greedy, kBlock, 0));
  }
  if (occ < 1) occ = 1;
  if (occ > 8) occ = 8;
  if (const char *e = getenv("KAIJU_GPU_BLOCKS_PER_CU")) { int v = atoi(e); if (v >= 1 && v <= 8) occ = v; }
  c->blocks_main = c->n_cu * occ;
  c->blocks_retry = p->mode == 0 ? 16 : 4;
  // (measurement knobs of the first-generation Greedy lanes, which serve -v: queue slots / match records per lane in the main
  //  pass, blocks of the retry pass - its lanes own 65535 slots each, 5.4 MB)
  if (const char *e = getenv("KAIJU_GPU_G1_POOL")) { int v = atoi(e); if (v >= 64 && v <= 65535) { c->g1_pool = (uint32_t)v; c->g1_pool_set = true; } }
  if (const char *e = getenv("KAIJU_GPU_G1_MATCH")) { int v = atoi(e); if (v >= 16 && v <= 65535) c->g1_match = (uint32_t)v; }
  if (const char *e = getenv("KAIJU_GPU_RETRY_BLOCKS")) { int v = atoi(e); if (v >= 1 && v <= 256) c->blocks_retry = v; }
  KJ_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  for (auto &e : c->ev) KJ_HIP(hipEventCreate(&e));
  *out = c.release();
  return KAIJU_GPU_OK;
  });
}

extern "C" void kaiju_gpu_destroy(kaiju_gpu_ctx *ctx) { delete ctx; }

// counters buffer layout (uint32): [0] main work counter, [1] retry work counter,
// [2] retry list length, [3] stage-1 error flags
// tax / d_compact: not null = the 16-byte records (LCA on the device) are written too - by the fused post-search pass where
// that serves the configuration (k_mem_post1 / k_mem_post2), by k_lca behind everything else otherwise
// the SEG pass over the fragments stage 1 (or k_segflag) queued
static void launch_seg(const kaiju_gpu_ctx *c, hipStream_t s, const Params &p, const SegTables &st, const Batch &b, const SegQueue &sq) {
  const dim3 grid(c->n_cu * 32), blk(kSegBlock);
  switch (c->seg_team) {
    case 8: hipLaunchKernelGGL(k_seg_teams<8>, grid, blk, 0, s, p, st, b, sq); break;
    case 16: hipLaunchKernelGGL(k_seg_teams<16>, grid, blk, 0, s, p, st, b, sq); break;
    case 32: hipLaunchKernelGGL(k_seg_teams<32>, grid, blk, 0, s, p, st, b, sq); break;
    default: hipLaunchKernelGGL(k_seg, grid, blk, 0, s, p, st, b, sq); break;
  }
}
// KAIJU_GPU_CALL_TIMES=1: host-side marks of a classification call on stderr (ms since the first mark; which thread) - where a
// call's wall time goes when it is not in the kernels (allocations of a context's first call, copies, waits)
static void call_mark(const char *what) {
  static const bool on = getenv("KAIJU_GPU_CALL_TIMES") != nullptr;
  if (!on) return;
  static const auto t0 = std::chrono::steady_clock::now();
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  fprintf(stderr, "[call %8.1f ms, thread %04x] %s\n", ms, (unsigned)(std::hash<std::thread::id>()(std::this_thread::get_id()) & 0xffffu), what);
}

static int launch_batch(kaiju_gpu_ctx *c, const void *d_seqs, uint64_t seq_bytes, const uint64_t *d_off,
                        uint32_t n, int paired, uint32_t max_read_len, kaiju_gpu_hit *d_out, hipStream_t s,
                        const kaiju_gpu_taxonomy *tax = nullptr, kaiju_gpu_compact *d_compact = nullptr) {
  const kaiju_gpu_index *ix = c->ix;
  const Params &p = c->kp;
  if (max_read_len == 0) max_read_len = 1024;
  const bool protein = (p.flags & kParamProtein) != 0;
  if (protein) {
    // a protein read is its own (single) frame: fragments are up to max_read_len long, not a third of it
    if (paired) return fail(KAIJU_GPU_ERR_ARG, "protein input has no paired mode (kaiju.cpp:201)");
    if (max_read_len > 0x10000000u) return fail(KAIJU_GPU_ERR_UNSUPPORTED, "protein read longer than 2^28");
    max_read_len *= 3;
  }
  const uint64_t max_pair = (uint64_t)max_read_len * (paired ? 2 : 1);
  call_mark("launch_batch: begin");
  // stage buffers
  const uint64_t pep_bytes = 2 * seq_bytes + kPepPerRead * n + 32 + 256;   // pep_base() + window over-read slack
  const uint64_t n_frag_slots = 2 * ((2 * seq_bytes) / (p.m + 1) + 7ull * n) + 8;
  int rc;
  if ((rc = ensure(c->pep, pep_bytes))) return rc;
  if ((rc = ensure(c->frags, n_frag_slots * sizeof(Frag)))) return rc;
  if (n_frag_slots >= 0xffffffffull) return fail(KAIJU_GPU_ERR_UNSUPPORTED, "batch too large: split it (fragment slots exceed 2^32)");
  if ((rc = ensure(c->meta, (size_t)n * sizeof(ReadMeta) + 16))) return rc;
  if ((rc = ensure(c->counters, 4096))) return rc;     // [0, 256) counters, [512, ..) totals of the counting lanes
  if ((rc = ensure(c->retry_list, (size_t)n * 4 + 16))) return rc;
  if ((rc = ensure(c->loc_list, (size_t)n * 4 + 16))) return rc;        // reads whose matches hold many rows (k_mem_locate_list)
  uint32_t *loc_list = static_cast<uint32_t *>(c->loc_list.p);
  Batch b;
  b.seqs = static_cast<const uint8_t *>(d_seqs); b.off = d_off; b.n_reads = n; b.paired = paired ? 1 : 0;
  b.pep = static_cast<uint8_t *>(c->pep.p); b.frags = static_cast<Frag *>(c->frags.p);
  b.meta = static_cast<ReadMeta *>(c->meta.p); b.hits = reinterpret_cast<Hit *>(d_out);
  uint32_t *cnt = static_cast<uint32_t *>(c->counters.p);
  // SEG work list: at most one entry per original fragment
  const uint64_t seg_cap = p.seg ? std::min<uint64_t>(n_frag_slots / 2 + 8, 0x00ffffffull) : 1;
  if ((rc = ensure(c->seg_items, seg_cap * sizeof(SegWork)))) return rc;
  if ((rc = ensure(c->seg_recs, seg_cap * sizeof(SegRec)))) return rc;
  SegQueue sq;
  sq.items = static_cast<SegWork *>(c->seg_items.p); sq.recs = static_cast<SegRec *>(c->seg_recs.p);
  sq.count = cnt + 4; sq.cap = (uint32_t)seg_cap;
#ifdef KJ_PROF
  KJ_HIP(hipMemsetAsync(cnt, 0, 4096, s));
#else
  KJ_HIP(hipMemsetAsync(cnt, 0, 1024, s));
#endif
  KJ_HIP(hipEventRecord(c->ev[0], s));
  const dim3 grid_reads((n + kBlock - 1) / kBlock), blk(kBlock);
  bool fused = false;              // the records (and, with a taxonomy, the 16-byte records) are finished by k_mem_post1 / _post2
  const dim3 grid_team((unsigned)(((uint64_t)n * kLocTeam + 255) / 256));          // k_mem_locate_wide / _team: kLocTeam lanes per read
  // which stage 1 / SEG flow: the fast stage 1 serves mates up to kS1MaxLenLong nucleotides (two instantiations); in MEM mode on the second-generation
  // lanes SEG is then looked at lazily (kj_core.h: kParamLazySeg), everywhere else stage 1 detects the SEG trigger itself
  const bool mem_narrow2 = ix->dev.blocks64 && ix->dev.kline && ix->dev.kline_k >= 2 && ix->dev.kline_k <= p.m;
  const bool mem_wide2 = ix->dev.blocks64 && ix->dev.mb_base && ix->dev.kmer64 && ix->dev.kmer_k >= 2 && ix->dev.kmer_k <= p.m;
  // kaiju -v in MEM mode: the VERBOSE instantiations of those lanes + k_mem_verbose - where a match's place in its read fits the
  // 16 + 16 bits of the lanes' notes (reads of 196 000 nt and more: the first-generation lanes, which also serve -v in Greedy
  // mode, the retry pass and the exact pass)
  const bool vb_v2 = c->verbose && !c->verbose_v1 && max_read_len / 3 + 4 < 65536 && 2 * max_pair / (p.m + 1) + 8 < 65536;
  const bool mem_v2 = p.mode == 0 && (mem_narrow2 || mem_wide2) && !c->mem_v1 && (!c->verbose || vb_v2);
  const bool fast1 = !protein && !c->stage1_old && max_read_len <= kS1MaxLenLong && p.m >= 1 && p.m <= 64;
  const bool long1 = max_read_len > kS1MaxLen;              // (192 .. 287 nt: the instantiation with six units per frame string)
  const bool lazy = fast1 && mem_v2 && p.seg && c->lazy_seg;
  const bool trig1 = fast1 && p.seg && !lazy;
  // the fused post-search pass (k_mem_post1 / _post2): narrow MEM lanes with the row -> taxon table, SEG lazily or not at all
  // (an eager SEG pass may send ANY read to the exact pass: nothing is final before that)
  fused = p.mode == 0 && mem_v2 && mem_narrow2 && ix->dev.row_tax && (lazy || !p.seg) && c->fused_post && n > 0 && !c->verbose;
  // unused id slots read as 0.  Not with the 16-byte records as the output on the fused path: the lanes write the header of every
  // record and the entries they announce in it, k_mem_post1 / _post2 read nothing else - d_hits is scratch there (1.84 GB less to
  // write per 10 M reads)
  if (n > 0 && !(fused && tax)) KJ_HIP(hipMemsetAsync(d_out, 0, (size_t)n * sizeof(kaiju_gpu_hit), s));
  if (n > 0) {
    // LDS staging area per lane: all frame strings of a read (or pair), rounded to 16 bytes
    uint32_t per_lane = (uint32_t)((2 * max_pair + 12 + 15) & ~15ull);
    if ((uint64_t)per_lane * kFragBlock > 60000) per_lane = 0;        // long reads: write in place
    if (protein)
      hipLaunchKernelGGL(k_fragments_protein, dim3((n + kFragBlock - 1) / kFragBlock), dim3(kFragBlock), 0, s,
                         ix->d_ct, p, ix->st, b, sq, cnt + 3);
    else if (fast1 && trig1 && long1)
      hipLaunchKernelGGL((k_fragments_fast<true, kS1UnitsLong>), dim3((n + kS1Block - 1) / kS1Block), dim3(kS1Block), 0, s, ix->d_s1, p, b, sq, cnt + 3);
    else if (fast1 && long1)
      hipLaunchKernelGGL((k_fragments_fast<false, kS1UnitsLong>), dim3((n + kS1Block - 1) / kS1Block), dim3(kS1Block), 0, s, ix->d_s1, p, b, sq, cnt + 3);
    else if (fast1 && trig1)
      hipLaunchKernelGGL(k_fragments_fast<true>, dim3((n + kS1Block - 1) / kS1Block), dim3(kS1Block), 0, s, ix->d_s1, p, b, sq, cnt + 3);
    else if (fast1)
      hipLaunchKernelGGL(k_fragments_fast<false>, dim3((n + kS1Block - 1) / kS1Block), dim3(kS1Block), 0, s, ix->d_s1, p, b, sq, cnt + 3);
    else
      hipLaunchKernelGGL(k_fragments, dim3((n + kFragBlock - 1) / kFragBlock), dim3(kFragBlock),
                         (size_t)per_lane * kFragBlock, s, ix->d_ct, p, ix->st, b, sq, cnt + 3, per_lane);
    KJ_HIP(hipGetLastError());
  }
  KJ_HIP(hipEventRecord(c->ev[1], s));
  if (n > 0 && p.seg && !lazy) {
    launch_seg(c, s, p, ix->st, b, sq);
    KJ_HIP(hipGetLastError());
    if (p.mode == 0) {
      hipLaunchKernelGGL(k_seg_apply, grid_reads, blk, 0, s, ix->d_ct, p, b, sq, cnt + 3);
      KJ_HIP(hipGetLastError());
    }
  }
  KJ_HIP(hipEventRecord(c->ev[2], s));
  WorkList wl_main;
  wl_main.counter = cnt + 0; wl_main.reads = nullptr; wl_main.n_items_ptr = nullptr; wl_main.n_items = n;
  wl_main.retry_list = static_cast<uint32_t *>(c->retry_list.p); wl_main.retry_count = cnt + 2;
  WorkList wl_retry;
  wl_retry.counter = cnt + 1; wl_retry.reads = static_cast<const uint32_t *>(c->retry_list.p);
  wl_retry.n_items_ptr = cnt + 2; wl_retry.n_items = 0; wl_retry.retry_list = nullptr; wl_retry.retry_count = nullptr;
  const uint64_t lanes_main = (uint64_t)c->blocks_main * kBlock;
  // the exact pass (kj_core.h: BigSeg; kernels in exact_pass.hip): reads with a fragment whose SEG regions did not fit
  // a SegRec are classified again behind the retry pass, with region lists of any length.  Counters: [5] listed reads,
  // [6] fragments of its queue, [7] its work counter, [20] pairs handed out of its pool
  constexpr uint32_t kRedoReads = 1u << 16, kRedoFrags = 1u << 18, kRedoPairs = 1u << 22;
  const bool exact_pass = n > 0 && p.seg && c->exact_pass;
  ExactPassLaunch xp{};
  if (exact_pass) {
    const uint64_t max_frag = protein ? max_read_len / 3 : max_read_len / 3 + 2;     // (max_read_len was tripled for protein reads)
    xp.cap_ints = (uint32_t)(2 * max_frag + 4);
    xp.cls_bytes = (uint32_t)((max_frag + 64) & ~63ull);
    const uint64_t per_block = 16ull * xp.cap_ints + xp.cls_bytes;
    xp.seg_blocks = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(32, (256ull << 20) / per_block));
    if ((rc = ensure(c->redo_bitmap, ((size_t)n / 32 + 2) * 4))) return rc;
    if ((rc = ensure(c->redo_list, (size_t)kRedoReads * 4))) return rc;
    if ((rc = ensure(c->redo_items, (size_t)kRedoFrags * sizeof(SegWork)))) return rc;
    if ((rc = ensure(c->redo_index, (size_t)kRedoFrags * sizeof(uint2)))) return rc;
    if ((rc = ensure(c->redo_pool, (size_t)kRedoPairs * 8))) return rc;
    if ((rc = ensure(c->redo_work, (size_t)xp.seg_blocks * 16 * xp.cap_ints))) return rc;
    if ((rc = ensure(c->redo_cls, (size_t)xp.seg_blocks * xp.cls_bytes))) return rc;
    KJ_HIP(hipMemsetAsync(c->redo_bitmap.p, 0, ((size_t)n / 32 + 2) * 4, s));
    xp.ix = ix->dev; xp.d_ct = ix->d_ct; xp.st = ix->st; xp.p = p; xp.b = b; xp.sq = sq; xp.cnt = cnt;
    xp.bitmap = static_cast<uint32_t *>(c->redo_bitmap.p); xp.list = static_cast<uint32_t *>(c->redo_list.p);
    xp.list_cap = kRedoReads;
    xp.sq2 = SegQueue{static_cast<SegWork *>(c->redo_items.p), nullptr, cnt + 6, kRedoFrags};
    xp.big = BigSeg{static_cast<uint2 *>(c->redo_index.p), static_cast<int32_t *>(c->redo_pool.p), cnt + 20, kRedoPairs};
    xp.work = static_cast<int32_t *>(c->redo_work.p); xp.cls = static_cast<uint8_t *>(c->redo_cls.p);
    xp.n_cu = c->n_cu; xp.stream = s;
  }
  VerboseOut vb{nullptr, nullptr, nullptr, nullptr, 0};
  if (c->verbose) {
    // columns 6/7: per read kVbAcc sequence numbers and room for 20 matched peptides
    c->vb_text_cap = kaiju_gpu_verbose_text_stride((uint32_t)std::min<uint64_t>(max_pair, 0xffffffffull), 0) - 1;   // (max_pair: tripled for protein reads above)
    if ((rc = ensure(c->vb_nacc, (size_t)n * 4 + 16))) return rc;
    if ((rc = ensure(c->vb_acc, (size_t)n * kVbAcc * 4 + 16))) return rc;
    if ((rc = ensure(c->vb_tlen, (size_t)n * 4 + 16))) return rc;
    if ((rc = ensure(c->vb_text, (size_t)n * c->vb_text_cap + 16))) return rc;
    vb.n_acc = static_cast<uint32_t *>(c->vb_nacc.p); vb.acc = static_cast<uint32_t *>(c->vb_acc.p);
    vb.text_len = static_cast<uint32_t *>(c->vb_tlen.p); vb.text = static_cast<uint8_t *>(c->vb_text.p);
    vb.text_cap = c->vb_text_cap;
    KJ_HIP(hipMemsetAsync(c->vb_nacc.p, 0, (size_t)n * 4, s));
    KJ_HIP(hipMemsetAsync(c->vb_tlen.p, 0, (size_t)n * 4, s));
  }
  if (p.mode == 0) {
    const uint32_t si_cap = 16;
    if ((rc = ensure(c->scratch_main[0], lanes_main * si_cap * sizeof(SIEntry)))) return rc;
    // every (fragment, end position) can yield at most one match
    const uint32_t si_cap_retry = (uint32_t)std::min<uint64_t>(2 * max_pair + 64, 1u << 24);
    int blocks_retry = c->blocks_retry;
    while (blocks_retry > 1 && (uint64_t)blocks_retry * kBlock * si_cap_retry * sizeof(SIEntry) > (1ull << 30)) blocks_retry /= 2;
    if ((rc = ensure(c->scratch_retry[0], (uint64_t)blocks_retry * kBlock * si_cap_retry * sizeof(SIEntry)))) return rc;
    if (n > 0) {
      const bool xo = (p.flags & kParamXOrder) != 0;
      SIEntry *si_main = static_cast<SIEntry *>(c->scratch_main[0].p);
      // the second-generation lane that serves this index (narrow: below 2^32 rows; wide: 64-bit positions)
      auto launch_v2 = [&](const Params &pp, const WorkList &wl, bool counting, bool second = false) {
        if (c->verbose) {
          if (mem_narrow2) hipLaunchKernelGGL(k_mem_vb, dim3(c->blocks_main), blk, 0, s, ix->dev, pp, b, wl, si_main, si_cap, vb.acc);
          else hipLaunchKernelGGL(k_mem_wide2_vb, dim3(c->blocks_main), blk, 0, s, ix->dev, pp, b, wl, si_main, si_cap, vb.acc);
        } else
        if (mem_narrow2) {
          if (second && !xo) hipLaunchKernelGGL(k_mem_second, dim3(c->blocks_main), blk, 0, s, ix->dev, pp, b, wl, si_main, si_cap);
          else if (counting) hipLaunchKernelGGL(k_mem_count, dim3(c->blocks_main), blk, 0, s, ix->dev, pp, b, wl, si_main, si_cap);
          else if (xo) hipLaunchKernelGGL(k_mem_x, dim3(c->blocks_main), blk, 0, s, ix->dev, pp, b, wl, si_main, si_cap);
          else hipLaunchKernelGGL(k_mem, dim3(c->blocks_main), blk, 0, s, ix->dev, pp, b, wl, si_main, si_cap);
        } else {
          if (counting) hipLaunchKernelGGL(k_mem_wide2_count, dim3(c->blocks_main), blk, 0, s, ix->dev, pp, b, wl, si_main, si_cap);
          else if (xo) hipLaunchKernelGGL(k_mem_wide2_x, dim3(c->blocks_main), blk, 0, s, ix->dev, pp, b, wl, si_main, si_cap);
          else hipLaunchKernelGGL(k_mem_wide2, dim3(c->blocks_main), blk, 0, s, ix->dev, pp, b, wl, si_main, si_cap);
        }
      };
      // the second-generation lanes leave the longest matches of every read in its hit record; k_mem_locate* behind the
      // searches turn them into ids (round 1-4: reads with three and more matches, and kaijux, were walked by the search lane)
      const bool defer = mem_v2;
      Params pd = p;
      if (defer) pd.flags |= kParamDeferLocate;
      if (mem_v2) {
        Params pm = pd;
        if (lazy) pm.flags |= kParamLazySeg;
        launch_v2(pm, wl_main, c->count_ops);
      } else if (!ix->dev.mb_base)
        hipLaunchKernelGGL(k_mem_v1, dim3(c->blocks_main), blk, 0, s, ix->dev, p, b, wl_main, si_main, si_cap, vb);
      else
        hipLaunchKernelGGL(k_mem_wide, dim3(c->blocks_main), blk, 0, s, ix->dev, p, b, wl_main, si_main, si_cap, vb);
      KJ_HIP(hipGetLastError());
      KJ_HIP(hipEventRecord(c->ev[3], s));
      uint32_t *todo = nullptr;
      const DevTaxonomy dt = tax ? tax->dev : DevTaxonomy{};
      CompactHit *const cmp = reinterpret_cast<CompactHit *>(d_compact);
      if (fused) {
        if ((rc = ensure(c->seglist, (size_t)n * 4 + 16))) return rc;
        if ((rc = ensure(c->todo_list, (size_t)n * 4 + 16))) return rc;
        todo = static_cast<uint32_t *>(c->todo_list.p);
        uint32_t *seglist = static_cast<uint32_t *>(c->seglist.p);
        if (tax) hipLaunchKernelGGL(k_mem_post1<true>, grid_reads, dim3(256), 0, s, ix->d_s1, ix->dev, p, b, dt, cmp, lazy ? 1 : 0, seglist, cnt + 22, todo, cnt + 26);
        else hipLaunchKernelGGL(k_mem_post1<false>, grid_reads, dim3(256), 0, s, ix->d_s1, ix->dev, p, b, dt, cmp, lazy ? 1 : 0, seglist, cnt + 22, todo, cnt + 26);
        KJ_HIP(hipGetLastError());
      }
      if (lazy) {
        // reads whose longest matches lie in fragments that SEG would cut: listed, SEG pass for their fragments, the
        // lists rewritten, searched again (counters: [22] listed reads, [23] work counter of that search)
        if ((rc = ensure(c->seglist, (size_t)n * 4 + 16))) return rc;
        uint32_t *seglist = static_cast<uint32_t *>(c->seglist.p);
        if (!fused) hipLaunchKernelGGL(k_trigcheck, grid_reads, dim3(256), 0, s, ix->d_s1, p, b, seglist, cnt + 22);
        hipLaunchKernelGGL(k_segflag, dim3(c->n_cu * 4), dim3(256), 0, s, p, ix->st, b, sq, seglist, cnt + 22, cnt + 3);
        launch_seg(c, s, p, ix->st, b, sq);
        hipLaunchKernelGGL(k_seg_apply_list, dim3(c->n_cu * 4), blk, 0, s, ix->d_ct, p, b, sq, seglist, cnt + 22, cnt + 3);
        WorkList wl_seg;
        wl_seg.counter = cnt + 23; wl_seg.reads = seglist; wl_seg.n_items_ptr = cnt + 22; wl_seg.n_items = 0;
        wl_seg.retry_list = wl_main.retry_list; wl_seg.retry_count = wl_main.retry_count;
        launch_v2(pd, wl_seg, false, true);
        KJ_HIP(hipGetLastError());
      }
      hipLaunchKernelGGL(k_mem_retry, dim3(blocks_retry), blk, 0, s, ix->dev, p, b, wl_retry,
                         static_cast<SIEntry *>(c->scratch_retry[0].p), si_cap_retry, vb);
      KJ_HIP(hipGetLastError());
      if (defer && c->verbose) {
        // columns 6 / 7 of the reads whose matches wait in their records (the retry pass above wrote its reads' own)
        if (mem_narrow2) hipLaunchKernelGGL(k_mem_verbose<false>, grid_reads, dim3(256), 0, s, ix->dev, p, b, vb);
        else hipLaunchKernelGGL(k_mem_verbose<true>, grid_reads, dim3(256), 0, s, ix->dev, p, b, vb);
        KJ_HIP(hipGetLastError());
      }
      if (defer && !fused) {
        if (mem_narrow2 && ix->dev.row_tax) {
          hipLaunchKernelGGL(k_mem_locate<false>, grid_reads, dim3(256), 0, s, ix->dev, p, b, loc_list, cnt + 24);
          hipLaunchKernelGGL(k_mem_locate_list<false>, dim3(c->n_cu * 8), dim3(256), 0, s, ix->dev, p, b, loc_list, cnt + 24);
        }
        else if (mem_narrow2) hipLaunchKernelGGL(k_mem_locate_team, grid_team, dim3(256), 0, s, ix->dev, p, b);
        else if (ix->dev.row_tax) {
          hipLaunchKernelGGL(k_mem_locate<true>, grid_reads, dim3(256), 0, s, ix->dev, p, b, loc_list, cnt + 24);
          hipLaunchKernelGGL(k_mem_locate_list<true>, dim3(c->n_cu * 8), dim3(256), 0, s, ix->dev, p, b, loc_list, cnt + 24);
        }
        else hipLaunchKernelGGL(k_mem_locate_wide, grid_team, dim3(256), 0, s, ix->dev, p, b);
        KJ_HIP(hipGetLastError());
      }
      if (exact_pass) {
        xp.si = static_cast<SIEntry *>(c->scratch_retry[0].p); xp.si_cap = si_cap_retry; xp.blocks_search = blocks_retry;
        xp.vb = vb;
        KJ_HIP(kj_launch_exact_pass(xp));
      }
      if (fused) {
        // the reads that were not through after k_mem_post1 (second search, retry pass, exact pass, matches of many rows)
        if (tax) hipLaunchKernelGGL(k_mem_post2<true>, dim3(c->n_cu * 8), dim3(256), 0, s, ix->dev, p, b, dt, cmp, todo, cnt + 26);
        else hipLaunchKernelGGL(k_mem_post2<false>, dim3(c->n_cu * 8), dim3(256), 0, s, ix->dev, p, b, dt, cmp, todo, cnt + 26);
        KJ_HIP(hipGetLastError());
      }
    } else KJ_HIP(hipEventRecord(c->ev[3], s));
  } else {
    const uint32_t frag_max = max_read_len / 3 + 4;
    GreedyArrays ga;
    // (-v: 512 slots.  With 192, 54 of 250 000 benchmark reads ran out of slots and went to the retry pass - whose few lanes
    //  then worked 180 ms on them, next to 100 ms for everything else: profiles/r06_l41/g1_probe.txt.  82 bytes a slot:
    //  11 GB of scratch for a -v run in Greedy mode instead of 4)
    ga.pool_cap = (c->verbose && !c->g1_pool_set) ? std::max<uint32_t>(c->g1_pool, 512u) : c->g1_pool; ga.match_cap = c->g1_match;
    // (-v: the VERBOSE instantiation of the second-generation lane unless KAIJU_GPU_VERBOSE_LANE=v1 - fragment positions must fit
    //  the 16 bits of a GBestV's substitution positions, as in the lane itself)
    const bool vb_g2 = c->verbose && !c->verbose_v1 && max_read_len / 3 + 4 < 65536;
    const bool use_g2 = c->greedy2 && (!c->verbose || vb_g2);
    const bool use_g3 = use_g2 && c->greedy3 && !c->verbose;
    // (the row-pool lane: one block per CU, kG3Pool rows each - its scratch in device memory is per ROW)
#ifdef KJ_GREEDY3
    const uint64_t lanes_g2 = use_g3 ? (uint64_t)c->n_cu * kG3Pool : lanes_main;
#else
    const uint64_t lanes_g2 = lanes_main;
#endif
    if (!use_g2) {
      if ((rc = ensure(c->scratch_main[0], lanes_main * ga.pool_cap * sizeof(GItem)))) return rc;
      if ((rc = ensure(c->scratch_main[1], lanes_main * ga.pool_cap * sizeof(uint16_t)))) return rc;
      if ((rc = ensure(c->scratch_main[2], lanes_main * ga.match_cap * sizeof(GMatch)))) return rc;
      if ((rc = ensure(c->scratch_main[4], lanes_main * 64 * sizeof(GBest)))) return rc;
    }
    ga.pool = static_cast<GItem *>(c->scratch_main[0].p); ga.ord = static_cast<uint16_t *>(c->scratch_main[1].p);
    ga.matches = static_cast<GMatch *>(c->scratch_main[2].p);
    ga.best = static_cast<GBest *>(c->scratch_main[4].p);
    ga.bestv = nullptr;
    if (c->verbose) {
      if ((rc = ensure(c->vb_bestv, lanes_main * 64 * sizeof(GBestV)))) return rc;
      ga.bestv = static_cast<GBestV *>(c->vb_bestv.p);
    }
    GreedyArrays gr;
    gr.pool_cap = 65535; gr.match_cap = frag_max + 8;
    const uint64_t lanes_retry = (uint64_t)c->blocks_retry * kBlock;
    if ((rc = ensure(c->scratch_retry[0], lanes_retry * gr.pool_cap * sizeof(GItem)))) return rc;
    if ((rc = ensure(c->scratch_retry[1], lanes_retry * gr.pool_cap * sizeof(uint16_t)))) return rc;
    if ((rc = ensure(c->scratch_retry[2], lanes_retry * gr.match_cap * sizeof(GMatch)))) return rc;
    if ((rc = ensure(c->scratch_retry[4], lanes_retry * 64 * sizeof(GBest)))) return rc;
    gr.pool = static_cast<GItem *>(c->scratch_retry[0].p); gr.ord = static_cast<uint16_t *>(c->scratch_retry[1].p);
    gr.matches = static_cast<GMatch *>(c->scratch_retry[2].p);
    gr.best = static_cast<GBest *>(c->scratch_retry[4].p);
    gr.bestv = nullptr;
    if (c->verbose) {
      if ((rc = ensure(c->vb_bestv_retry, lanes_retry * 64 * sizeof(GBestV)))) return rc;
      gr.bestv = static_cast<GBestV *>(c->vb_bestv_retry.p);
    }
    GreedyArrays2 g2{};
    if (use_g2) {
      if ((rc = ensure(c->scratch_main[5], (lanes_g2 * (8 * kGSlotsAll) + 4) * sizeof(u128)))) return rc;
      if ((rc = ensure(c->scratch_main[6], (lanes_g2 * (kGSlotsAll - kGSlots) + 16) * sizeof(uint32_t)))) return rc;   // (+ slack: read 16 bytes at a time)
      if ((rc = ensure(c->scratch_main[7], lanes_g2 * kGMaxMAll * sizeof(GMatch2)))) return rc;
      if ((rc = ensure(c->scratch_main[8], lanes_g2 * (kGMaxMAll - kGMaxM) * sizeof(uint16_t)))) return rc;
      const bool g_wide = ix->dev.mb_base != nullptr;
      if ((rc = ensure(c->scratch_main[9], lanes_g2 * 64 * (g_wide ? sizeof(GBest2W) : sizeof(GBest2))))) return rc;
      g2.pool = static_cast<u128 *>(c->scratch_main[5].p); g2.prio_ext = static_cast<uint32_t *>(c->scratch_main[6].p);
      g2.matches = static_cast<GMatch2 *>(c->scratch_main[7].p); g2.mq_ext = static_cast<uint16_t *>(c->scratch_main[8].p);
      g2.best = g_wide ? nullptr : static_cast<GBest2 *>(c->scratch_main[9].p);
      g2.bestw = g_wide ? static_cast<GBest2W *>(c->scratch_main[9].p) : nullptr;
      g2.gate = c->greedy_gate;
      g2.bestv = nullptr; g2.vb = vb;
      if (c->verbose) {
        if ((rc = ensure(c->vb_bestv, lanes_g2 * 64 * sizeof(GBestV)))) return rc;
        g2.bestv = static_cast<GBestV *>(c->vb_bestv.p);
      }
    }
    call_mark("launch_batch: Greedy scratch in place");
    if (n > 0) {
      Params pg = p;
      if (use_g2) pg.flags |= kParamDeferLocate;       // (greedy_lane2 leaves every read's best matches to k_mem_locate*)
      const bool g_wide = ix->dev.mb_base != nullptr;
#ifdef KJ_GREEDY3
      if (use_g3 && c->count_ops)
        hipLaunchKernelGGL(k_greedy3_count, dim3(c->n_cu), dim3(c->g3_threads), 0, s, ix->dev, ix->d_ct, pg, sq, b, wl_main, g2, c->g3_split);
      else if (use_g3)
        hipLaunchKernelGGL(k_greedy3, dim3(c->n_cu), dim3(c->g3_threads), 0, s, ix->dev, ix->d_ct, pg, sq, b, wl_main, g2, c->g3_split);
      else
#endif
      if (use_g2 && c->verbose && g_wide)
        hipLaunchKernelGGL(k_greedy2_wide_vb, dim3(c->blocks_main), blk, kGreedy2Lds, s, ix->dev, ix->d_ct, pg, sq, b, wl_main, g2);
      else if (use_g2 && c->verbose)
        hipLaunchKernelGGL(k_greedy2_vb, dim3(c->blocks_main), blk, kGreedy2Lds, s, ix->dev, ix->d_ct, pg, sq, b, wl_main, g2);
      else if (use_g2 && c->count_ops && g_wide)
        hipLaunchKernelGGL(k_greedy2_wide_count, dim3(c->blocks_main), blk, kGreedy2Lds, s, ix->dev, ix->d_ct, pg, sq, b, wl_main, g2);
      else if (use_g2 && g_wide)
        hipLaunchKernelGGL(k_greedy2_wide, dim3(c->blocks_main), blk, kGreedy2Lds, s, ix->dev, ix->d_ct, pg, sq, b, wl_main, g2);
      else if (use_g2 && c->count_ops)
        hipLaunchKernelGGL(k_greedy2_count, dim3(c->blocks_main), blk, kGreedy2Lds, s, ix->dev, ix->d_ct, pg, sq, b, wl_main, g2);
      else if (use_g2)
        hipLaunchKernelGGL(k_greedy2, dim3(c->blocks_main), blk, kGreedy2Lds, s, ix->dev, ix->d_ct, pg, sq, b, wl_main, g2);
      else
        hipLaunchKernelGGL(k_greedy, dim3(c->blocks_main), blk, 0, s, ix->dev, ix->d_ct, p, sq, b, wl_main, ga, vb);
      KJ_HIP(hipGetLastError());
      KJ_HIP(hipEventRecord(c->ev[3], s));
      hipLaunchKernelGGL(k_greedy_retry, dim3(c->blocks_retry), blk, 0, s, ix->dev, ix->d_ct, p, sq, b, wl_retry, gr, vb);
      KJ_HIP(hipGetLastError());
      if ((pg.flags & kParamDeferLocate) && c->verbose) {
        // column 6 of the reads whose best matches wait in their records (the retry pass above wrote its reads' own columns)
        if (g_wide) hipLaunchKernelGGL((k_mem_verbose<true, false>), grid_reads, dim3(256), 0, s, ix->dev, p, b, vb);
        else hipLaunchKernelGGL((k_mem_verbose<false, false>), grid_reads, dim3(256), 0, s, ix->dev, p, b, vb);
        KJ_HIP(hipGetLastError());
      }
      if (pg.flags & kParamDeferLocate) {
        if (g_wide && ix->dev.row_tax) {
          hipLaunchKernelGGL(k_mem_locate<true>, grid_reads, dim3(256), 0, s, ix->dev, p, b, loc_list, cnt + 25);
          hipLaunchKernelGGL(k_mem_locate_list<true>, dim3(c->n_cu * 8), dim3(256), 0, s, ix->dev, p, b, loc_list, cnt + 25);
        }
        else if (g_wide) hipLaunchKernelGGL(k_mem_locate_wide, grid_team, dim3(256), 0, s, ix->dev, p, b);
        else if (ix->dev.row_tax) {
          hipLaunchKernelGGL(k_mem_locate<false>, grid_reads, dim3(256), 0, s, ix->dev, p, b, loc_list, cnt + 25);
          hipLaunchKernelGGL(k_mem_locate_list<false>, dim3(c->n_cu * 8), dim3(256), 0, s, ix->dev, p, b, loc_list, cnt + 25);
        }
        else hipLaunchKernelGGL(k_mem_locate_team, grid_team, dim3(256), 0, s, ix->dev, p, b);
        KJ_HIP(hipGetLastError());
      }
      if (exact_pass) {
        xp.g_pool = gr.pool; xp.g_ord = gr.ord; xp.g_matches = gr.matches; xp.g_best = gr.best; xp.g_bestv = gr.bestv;
        xp.g_pool_cap = gr.pool_cap; xp.g_match_cap = gr.match_cap; xp.blocks_search = c->blocks_retry;
        xp.vb = vb;
        KJ_HIP(kj_launch_exact_pass(xp));
      }
    } else KJ_HIP(hipEventRecord(c->ev[3], s));
  }
  if (tax && !fused && n > 0) {
    hipLaunchKernelGGL(k_lca, dim3((n + 255) / 256), dim3(256), 0, s, tax->dev, reinterpret_cast<const Hit *>(d_out), n,
                       reinterpret_cast<CompactHit *>(d_compact));
    KJ_HIP(hipGetLastError());
  }
  KJ_HIP(hipEventRecord(c->ev[4], s));
  c->ev_valid = true;
  c->last_n = n;
  if (const char *dump = c->dump_frags) {
    // developer aid: the fragment lists as the search kernels saw them, "#" per read then "key:PEPTIDE flags"
    KJ_HIP(hipStreamSynchronize(s));
    std::vector<ReadMeta> hm(n);
    KJ_HIP(hipMemcpy(hm.data(), c->meta.p, (size_t)n * sizeof(ReadMeta), hipMemcpyDeviceToHost));
    std::vector<uint8_t> hp(c->pep.cap);
    KJ_HIP(hipMemcpy(hp.data(), c->pep.p, c->pep.cap, hipMemcpyDeviceToHost));
    std::vector<uint8_t> hf(c->frags.cap);
    KJ_HIP(hipMemcpy(hf.data(), c->frags.p, c->frags.cap, hipMemcpyDeviceToHost));
    const Frag *F = reinterpret_cast<const Frag *>(hf.data());
    if (FILE *fp = fopen(dump, "a")) {
      const char *alpha = ix->info.alphabet;
      for (uint32_t r = 0; r < n; r++) {
        fprintf(fp, "#\n");
        for (uint32_t k = 0; k < (hm[r].nfrag & ~kNfragSegPending); k++) {
          const Frag &f = F[hm[r].frag + k];
          fprintf(fp, "%u:", f.key);
          for (uint32_t x = 0; x < f.len; x++) fputc(alpha[hp[hm[r].pep + f.start + x] % 21], fp);
          fprintf(fp, "\n");
        }
      }
      fclose(fp);
    }
  }
  return KAIJU_GPU_OK;
}

extern "C" int kaiju_gpu_classify_batch_device(kaiju_gpu_ctx *ctx, const void *d_seqs, uint64_t seq_bytes,
                                               const uint64_t *d_off, uint32_t n_reads, int paired,
                                               kaiju_gpu_hit *d_out, void *stream) {
  return guarded([&]() -> int {
  if (!ctx || (!d_seqs && seq_bytes) || !d_off || (!d_out && n_reads)) return fail(KAIJU_GPU_ERR_ARG, "NULL argument");
  KJ_HIP(hipSetDevice(ctx->ix->device));
  hipStream_t s = stream ? static_cast<hipStream_t>(stream) : ctx->stream;
  return launch_batch(ctx, d_seqs, seq_bytes, d_off, n_reads, paired, ctx->max_read_len, d_out, s);
  });
}

// host buffers -> device, kernels queued on the context's stream; the hit records stay in ctx->h_hits
static int classify_host_buffers(kaiju_gpu_ctx *ctx, const char *seqs, const uint64_t *off, uint32_t n_reads, int paired,
                                 const kaiju_gpu_taxonomy *tax = nullptr) {
  KJ_HIP(hipSetDevice(ctx->ix->device));
  if (off[0] != 0) return fail(KAIJU_GPU_ERR_ARG, "off[0] must be 0");
  const uint64_t seq_bytes = off[2 * (uint64_t)n_reads];
  uint32_t max_len = 0;
  for (uint64_t i = 0; i < 2 * (uint64_t)n_reads; i++) {
    if (off[i + 1] < off[i]) return fail(KAIJU_GPU_ERR_ARG, "offsets must be non-decreasing");
    const uint64_t l = off[i + 1] - off[i];
    if (l > 0x3fffffffull) return fail(KAIJU_GPU_ERR_UNSUPPORTED, "read longer than 2^30");
    if (l > max_len) max_len = (uint32_t)l;
  }
  if (seq_bytes && !seqs) return fail(KAIJU_GPU_ERR_ARG, "seqs is NULL");
  int rc;
  if ((rc = ensure(ctx->h_seqs, seq_bytes + 64))) return rc;
  if ((rc = ensure(ctx->h_off, (2 * (size_t)n_reads + 1) * 8))) return rc;
  if ((rc = ensure(ctx->h_hits, (size_t)n_reads * sizeof(kaiju_gpu_hit)))) return rc;
  hipStream_t s = ctx->stream;
  if (seq_bytes) KJ_HIP(hipMemcpyAsync(ctx->h_seqs.p, seqs, seq_bytes, hipMemcpyHostToDevice, s));
  KJ_HIP(hipMemcpyAsync(ctx->h_off.p, off, (2 * (size_t)n_reads + 1) * 8, hipMemcpyHostToDevice, s));
  if (tax && (rc = ensure(ctx->h_compact, (size_t)n_reads * sizeof(kaiju_gpu_compact)))) return rc;
  return launch_batch(ctx, ctx->h_seqs.p, seq_bytes, static_cast<const uint64_t *>(ctx->h_off.p), n_reads, paired,
                      max_len ? max_len : 1, static_cast<kaiju_gpu_hit *>(ctx->h_hits.p), s, tax,
                      tax ? static_cast<kaiju_gpu_compact *>(ctx->h_compact.p) : nullptr);
}

extern "C" int kaiju_gpu_classify_batch(kaiju_gpu_ctx *ctx, const char *seqs, const uint64_t *off,
                                        uint32_t n_reads, int paired, kaiju_gpu_hit *out) {
  return guarded([&]() -> int {
  if (!ctx || !off || (!out && n_reads)) return fail(KAIJU_GPU_ERR_ARG, "NULL argument");
  if (n_reads == 0) return KAIJU_GPU_OK;
  const int rc = classify_host_buffers(ctx, seqs, off, n_reads, paired);
  if (rc) return rc;
  hipStream_t s = ctx->stream;
  KJ_HIP(hipMemcpyAsync(out, ctx->h_hits.p, (size_t)n_reads * sizeof(kaiju_gpu_hit), hipMemcpyDeviceToHost, s));
  KJ_HIP(hipStreamSynchronize(s));
  return KAIJU_GPU_OK;
  });
}

// kaiju_gpu_classify_batch_device and kaiju_gpu_lca_batch_device in one call: the 16-byte records are written by the search's
// own post-search pass where the configuration allows (no separate pass over the 184-byte records)
extern "C" int kaiju_gpu_classify_batch_device_compact(kaiju_gpu_ctx *ctx, const kaiju_gpu_taxonomy *t, const void *d_seqs, uint64_t seq_bytes,
                                                       const uint64_t *d_off, uint32_t n_reads, int paired,
                                                       kaiju_gpu_hit *d_hits, kaiju_gpu_compact *d_out, void *stream) {
  return guarded([&]() -> int {
  if (!ctx || !t || (!d_seqs && seq_bytes) || !d_off || (!d_hits && n_reads) || (!d_out && n_reads)) return fail(KAIJU_GPU_ERR_ARG, "NULL argument");
  if (t->device != ctx->ix->device) return fail(KAIJU_GPU_ERR_ARG, "taxonomy lives on another device");
  KJ_HIP(hipSetDevice(ctx->ix->device));
  hipStream_t s = stream ? static_cast<hipStream_t>(stream) : ctx->stream;
  return launch_batch(ctx, d_seqs, seq_bytes, d_off, n_reads, paired, ctx->max_read_len, d_hits, s, t, d_out);
  });
}

extern "C" int kaiju_gpu_set_max_read_length(kaiju_gpu_ctx *ctx, uint32_t max_read_len) {
  if (!ctx || max_read_len == 0 || max_read_len > 0x3fffffffu) return fail(KAIJU_GPU_ERR_ARG, "bad max_read_len");
  ctx->max_read_len = max_read_len;
  return KAIJU_GPU_OK;
}

// ---- LCA on the device ---------------------------------------------------------------------------
extern "C" int kaiju_gpu_taxonomy_upload(const kaiju_taxonomy *t, int device_id, kaiju_gpu_taxonomy **out) {
  return guarded([&]() -> int {
  if (!t || !out) return fail(KAIJU_GPU_ERR_ARG, "NULL argument");
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(KAIJU_GPU_ERR_NO_DEVICE, "hipGetDeviceCount found no device");
  if (device_id < 0 || device_id >= ndev) return fail(KAIJU_GPU_ERR_ARG, "device_id out of range");
  std::vector<uint64_t> key, parent_id;
  std::vector<uint32_t> parent_slot, depth;
  kj_taxonomy_table(t, key, parent_id, parent_slot, depth);
  std::unique_ptr<kaiju_gpu_taxonomy> g(new kaiju_gpu_taxonomy());
  g->device = device_id;
  KJ_HIP(hipSetDevice(device_id));
  auto up = [&](const void *src, size_t bytes, const void **dst) -> int {
    void *p = nullptr;
    KJ_HIP(hipMalloc(&p, bytes + 16));
    g->allocs.push_back(p);
    KJ_HIP(hipMemcpy(p, src, bytes, hipMemcpyHostToDevice));
    *dst = p;
    return 0;
  };
  int rc;
  if ((rc = up(key.data(), key.size() * 8, (const void **)&g->dev.key))) return rc;
  if ((rc = up(parent_id.data(), parent_id.size() * 8, (const void **)&g->dev.parent_id))) return rc;
  if ((rc = up(parent_slot.data(), parent_slot.size() * 4, (const void **)&g->dev.parent_slot))) return rc;
  if ((rc = up(depth.data(), depth.size() * 4, (const void **)&g->dev.depth))) return rc;
  g->dev.cap_mask = (uint32_t)(key.size() - 1);
  *out = g.release();
  return KAIJU_GPU_OK;
  });
}

extern "C" void kaiju_gpu_taxonomy_free(kaiju_gpu_taxonomy *t) { delete t; }

static_assert(sizeof(CompactHit) == sizeof(kaiju_gpu_compact) && sizeof(CompactHit) == 16, "compact record layout");

extern "C" int kaiju_gpu_lca_batch_device(kaiju_gpu_ctx *ctx, const kaiju_gpu_taxonomy *t, const kaiju_gpu_hit *d_hits,
                                          uint32_t n_reads, kaiju_gpu_compact *d_out, void *stream) {
  if (!ctx || !t || (!d_hits && n_reads) || (!d_out && n_reads)) return fail(KAIJU_GPU_ERR_ARG, "NULL argument");
  if (t->device != ctx->ix->device) return fail(KAIJU_GPU_ERR_ARG, "taxonomy lives on another device");
  if (n_reads == 0) return KAIJU_GPU_OK;
  KJ_HIP(hipSetDevice(ctx->ix->device));
  // NULL = the context's own stream, as in kaiju_gpu_classify_batch_device (the LCA must queue behind the search)
  hipStream_t s = stream ? static_cast<hipStream_t>(stream) : ctx->stream;
  hipLaunchKernelGGL(k_lca, dim3((n_reads + 255) / 256), dim3(256), 0, s, t->dev,
                     reinterpret_cast<const Hit *>(d_hits), n_reads, reinterpret_cast<CompactHit *>(d_out));
  KJ_HIP(hipGetLastError());
  return KAIJU_GPU_OK;
}

// (the part both entry points share: classification, k_vb_pack, the records / accessions / packed text fetched - the text
//  into ctx->vb_host, read r's at pos[r])
static int verbose_core(kaiju_gpu_ctx *ctx, const char *seqs, const uint64_t *off, uint32_t n_reads, int paired, kaiju_gpu_hit *out,
                        kaiju_gpu_verbose *vout, std::vector<uint64_t> &pos, std::vector<uint32_t> &tlen) {
  ctx->verbose = true;
  int rc = classify_host_buffers(ctx, seqs, off, n_reads, paired);
  ctx->verbose = false;
  if (rc) return rc;
  call_mark("verbose: launch_batch returned");
  hipStream_t s = ctx->stream;
  // column 7: letters, packed on the device (k_vb_pack) - the rows of text_cap bytes stay there
  const char *alpha = ctx->ix->info.alphabet;
  VbAlphabet al{};
  al.n = (uint32_t)std::min<size_t>(strlen(alpha), sizeof al.c);
  memcpy(al.c, alpha, al.n);
  if ((rc = ensure(ctx->vb_packed, (size_t)n_reads * ctx->vb_text_cap + 16))) return rc;
  if ((rc = ensure(ctx->vb_pos, (size_t)n_reads * 8 + 16))) return rc;
  VerboseOut vbd{static_cast<uint32_t *>(ctx->vb_nacc.p), static_cast<uint32_t *>(ctx->vb_acc.p), static_cast<uint32_t *>(ctx->vb_tlen.p),
                 static_cast<uint8_t *>(ctx->vb_text.p), ctx->vb_text_cap};
  unsigned long long *d_total = reinterpret_cast<unsigned long long *>(static_cast<uint8_t *>(ctx->vb_pos.p) + (size_t)n_reads * 8);
  KJ_HIP(hipMemsetAsync(d_total, 0, 8, s));
  hipLaunchKernelGGL(k_vb_pack, dim3((n_reads + 255) / 256), dim3(256), 0, s, vbd, n_reads, al, static_cast<uint8_t *>(ctx->vb_packed.p),
                     static_cast<uint64_t *>(ctx->vb_pos.p), d_total);
  KJ_HIP(hipGetLastError());
  std::vector<uint32_t> &nacc = ctx->vb_h_nacc, &acc = ctx->vb_h_acc;
  if (nacc.size() < n_reads) nacc.resize(n_reads);
  if (acc.size() < (size_t)n_reads * kVbAcc) acc.resize((size_t)n_reads * kVbAcc);
  pos.resize((size_t)n_reads + 1); tlen.resize(n_reads);
  KJ_HIP(hipMemcpyAsync(out, ctx->h_hits.p, (size_t)n_reads * sizeof(kaiju_gpu_hit), hipMemcpyDeviceToHost, s));
  KJ_HIP(hipMemcpyAsync(nacc.data(), ctx->vb_nacc.p, (size_t)n_reads * 4, hipMemcpyDeviceToHost, s));
  KJ_HIP(hipMemcpyAsync(tlen.data(), ctx->vb_tlen.p, (size_t)n_reads * 4, hipMemcpyDeviceToHost, s));
  KJ_HIP(hipMemcpyAsync(acc.data(), ctx->vb_acc.p, (size_t)n_reads * kVbAcc * 4, hipMemcpyDeviceToHost, s));
  KJ_HIP(hipMemcpyAsync(pos.data(), ctx->vb_pos.p, ((size_t)n_reads + 1) * 8, hipMemcpyDeviceToHost, s));
  call_mark("verbose: everything queued");
  KJ_HIP(hipStreamSynchronize(s));
  call_mark("verbose: stream drained (records, accessions, positions on the host)");
  const uint64_t total = pos[n_reads];
  if (total > (uint64_t)n_reads * ctx->vb_text_cap) return fail(KAIJU_GPU_ERR_HIP, "k_vb_pack: impossible total");
  if (ctx->vb_host.size() < total) ctx->vb_host.resize((size_t)total);
  if (total) KJ_HIP(hipMemcpy(ctx->vb_host.data(), ctx->vb_packed.p, (size_t)total, hipMemcpyDeviceToHost));
  for (uint32_t r = 0; r < n_reads; r++) {
    kaiju_gpu_verbose &v = vout[r];
    v.n_acc = nacc[r] > (uint32_t)kVbAcc ? (uint32_t)kVbAcc : nacc[r];
    for (uint32_t q = 0; q < (uint32_t)KAIJU_GPU_MAX_ACC; q++) v.acc_iseq[q] = q < v.n_acc ? acc[(size_t)r * kVbAcc + q] : 0;
    v.text_len = tlen[r] < ctx->vb_text_cap ? tlen[r] : ctx->vb_text_cap;
    v.truncated = tlen[r] > ctx->vb_text_cap ? 1u : 0u;
  }
  return KAIJU_GPU_OK;
}

// Verbose classification (the reference's -v): hit records plus, per read, the sequences that give
// column 6 and the text of column 7.  The second-generation lanes (MEM: k_mem_vb / k_mem_wide2_vb, Greedy: k_greedy2_vb /
// k_greedy2_wide_vb; + k_mem_verbose); the retry pass and the exact pass: the first-generation lanes.
extern "C" int kaiju_gpu_classify_batch_verbose(kaiju_gpu_ctx *ctx, const char *seqs, const uint64_t *off, uint32_t n_reads,
                                                int paired, kaiju_gpu_hit *out, kaiju_gpu_verbose *vout, char *text,
                                                uint32_t text_stride) {
  return guarded([&]() -> int {
  if (!ctx || !off || (n_reads && (!out || !vout || !text))) return fail(KAIJU_GPU_ERR_ARG, "NULL argument");
  if (n_reads == 0) return KAIJU_GPU_OK;
  std::vector<uint64_t> pos;
  std::vector<uint32_t> tlen;
  int rc = verbose_core(ctx, seqs, off, n_reads, paired, out, vout, pos, tlen);
  if (rc) return rc;
  for (uint32_t r = 0; r < n_reads; r++) {
    kaiju_gpu_verbose &v = vout[r];
    const uint32_t have = v.text_len;
    if (have + 1 > text_stride) v.truncated = 1u;
    const uint32_t w = have + 1 <= text_stride ? have : (text_stride ? text_stride - 1 : 0);
    char *dst = text + (size_t)r * text_stride;
    if (w) memcpy(dst, ctx->vb_host.data() + pos[r], w);
    if (text_stride) dst[w] = 0;
    v.text_len = w;
  }
  return KAIJU_GPU_OK;
  });
}

// The same with column 7 of the whole batch as ONE string that the context owns (no row of text_stride bytes per read for the
// caller to allocate, fault in and walk: at 150 bp a row is 1 KB, of which a classified read uses ~40 bytes)
extern "C" int kaiju_gpu_classify_batch_verbose_packed(kaiju_gpu_ctx *ctx, const char *seqs, const uint64_t *off, uint32_t n_reads,
                                                       int paired, kaiju_gpu_hit *out, kaiju_gpu_verbose *vout, uint64_t *text_pos,
                                                       const char **text, uint64_t *text_bytes) {
  return guarded([&]() -> int {
  if (!ctx || !off || !text || !text_bytes || (n_reads && (!out || !vout || !text_pos))) return fail(KAIJU_GPU_ERR_ARG, "NULL argument");
  *text = nullptr; *text_bytes = 0;
  if (n_reads == 0) return KAIJU_GPU_OK;
  std::vector<uint64_t> pos;
  std::vector<uint32_t> tlen;
  int rc = verbose_core(ctx, seqs, off, n_reads, paired, out, vout, pos, tlen);
  if (rc) return rc;
  memcpy(text_pos, pos.data(), (size_t)n_reads * 8);
  *text = reinterpret_cast<const char *>(ctx->vb_host.data());
  *text_bytes = pos[n_reads];
  return KAIJU_GPU_OK;
  });
}

// room for the text of column 7 of one read: up to 20 matched peptides (max_matches_SI), each at most a fragment long and
// followed by a comma; 64 KB at most (reads beyond ~9.8 kb with that many long matches are flagged `truncated`)
extern "C" uint32_t kaiju_gpu_verbose_text_stride(uint32_t max_pair_len, int input_is_protein) {
  const uint64_t frag = input_is_protein ? (uint64_t)max_pair_len : (uint64_t)max_pair_len / 3;
  return (uint32_t)std::min<uint64_t>(20ull * (frag + 2), 65536) + 1;
}

extern "C" const char *kaiju_gpu_index_seq_name(const kaiju_gpu_index *ix, uint32_t iseq) {
  if (!ix || iseq >= ix->names.size()) return nullptr;
  return ix->names[iseq].c_str();
}

// classify host buffers and return 16-byte records only (the 184-byte hit records never leave the device)
extern "C" int kaiju_gpu_classify_batch_compact(kaiju_gpu_ctx *ctx, const kaiju_gpu_taxonomy *t, const char *seqs,
                                                const uint64_t *off, uint32_t n_reads, int paired, kaiju_gpu_compact *out) {
  return guarded([&]() -> int {
  if (!ctx || !t || !off || (!out && n_reads)) return fail(KAIJU_GPU_ERR_ARG, "NULL argument");
  if (t->device != ctx->ix->device) return fail(KAIJU_GPU_ERR_ARG, "taxonomy lives on another device");
  if (n_reads == 0) return KAIJU_GPU_OK;
  int rc = classify_host_buffers(ctx, seqs, off, n_reads, paired, t);      // (the 16-byte records come with the search)
  if (rc) return rc;
  hipStream_t s = ctx->stream;
  KJ_HIP(hipMemcpyAsync(out, ctx->h_compact.p, (size_t)n_reads * sizeof(kaiju_gpu_compact), hipMemcpyDeviceToHost, s));
  KJ_HIP(hipStreamSynchronize(s));
  return KAIJU_GPU_OK;
  });
}

// host buffers in and out (blocking): upload the hit records, k_lca, download the compact records
extern "C" int kaiju_gpu_lca_batch(kaiju_gpu_ctx *ctx, const kaiju_gpu_taxonomy *t, const kaiju_gpu_hit *hits,
                                   uint32_t n_reads, kaiju_gpu_compact *out) {
  if (!ctx || !t || (!hits && n_reads) || (!out && n_reads)) return fail(KAIJU_GPU_ERR_ARG, "NULL argument");
  if (n_reads == 0) return KAIJU_GPU_OK;
  KJ_HIP(hipSetDevice(ctx->ix->device));
  void *d_in = nullptr, *d_out = nullptr;
  KJ_HIP(hipMalloc(&d_in, (size_t)n_reads * sizeof(kaiju_gpu_hit)));
  if (hipMalloc(&d_out, (size_t)n_reads * sizeof(kaiju_gpu_compact)) != hipSuccess) { (void)hipFree(d_in); return fail(KAIJU_GPU_ERR_NOMEM, "hipMalloc"); }
  int rc = KAIJU_GPU_OK;
  if (hipMemcpy(d_in, hits, (size_t)n_reads * sizeof(kaiju_gpu_hit), hipMemcpyHostToDevice) != hipSuccess) rc = KAIJU_GPU_ERR_HIP;
  if (rc == KAIJU_GPU_OK) rc = kaiju_gpu_lca_batch_device(ctx, t, static_cast<const kaiju_gpu_hit *>(d_in), n_reads,
                                                          static_cast<kaiju_gpu_compact *>(d_out), nullptr);
  if (rc == KAIJU_GPU_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = KAIJU_GPU_ERR_HIP;   // (k_lca ran on the context's stream)
  if (rc == KAIJU_GPU_OK && hipMemcpy(out, d_out, (size_t)n_reads * sizeof(kaiju_gpu_compact), hipMemcpyDeviceToHost) != hipSuccess) rc = KAIJU_GPU_ERR_HIP;
  (void)hipFree(d_in); (void)hipFree(d_out);
  return rc == KAIJU_GPU_OK ? rc : fail(rc, "kaiju_gpu_lca_batch");
}

extern "C" int kaiju_gpu_get_stream(kaiju_gpu_ctx *ctx, void **stream) {
  if (!ctx || !stream) return fail(KAIJU_GPU_ERR_ARG, "NULL argument");
  *stream = ctx->stream;
  return KAIJU_GPU_OK;
}
extern "C" int kaiju_gpu_set_count_ops(kaiju_gpu_ctx *ctx, int on) {
  if (!ctx) return fail(KAIJU_GPU_ERR_ARG, "NULL argument");
  ctx->count_ops = on != 0;
  return KAIJU_GPU_OK;
}
extern "C" int kaiju_gpu_get_op_counts(kaiju_gpu_ctx *ctx, uint64_t *out, uint32_t n_out) {
  if (!ctx || !out) return fail(KAIJU_GPU_ERR_ARG, "NULL argument");
  for (uint32_t x = 0; x < n_out; x++) out[x] = 0;
  if (!ctx->counters.p) return KAIJU_GPU_OK;
  KJ_HIP(hipSetDevice(ctx->ix->device));
  KJ_HIP(hipDeviceSynchronize());
  unsigned long long v[kOpcN];
  static_assert(kOpcOffsetBytes + sizeof v <= 1024, "the totals fit the counter block");
  KJ_HIP(hipMemcpy(v, static_cast<const uint8_t *>(ctx->counters.p) + kOpcOffsetBytes, sizeof v, hipMemcpyDeviceToHost));
  for (uint32_t x = 0; x < n_out && x < (uint32_t)kOpcN; x++) out[x] = v[x];
  return KAIJU_GPU_OK;
}

extern "C" int kaiju_gpu_synchronize(kaiju_gpu_ctx *ctx) {
  if (!ctx) return fail(KAIJU_GPU_ERR_ARG, "NULL argument");
  KJ_HIP(hipSetDevice(ctx->ix->device));
  KJ_HIP(hipDeviceSynchronize());
  return KAIJU_GPU_OK;
}

extern "C" int kaiju_gpu_get_stats(kaiju_gpu_ctx *ctx, kaiju_gpu_stats *stats) {
  if (!ctx || !stats) return fail(KAIJU_GPU_ERR_ARG, "NULL argument");
  memset(stats, 0, sizeof *stats);
  if (!ctx->ev_valid) return KAIJU_GPU_OK;
  KJ_HIP(hipSetDevice(ctx->ix->device));
  KJ_HIP(hipEventSynchronize(ctx->ev[4]));
  float t01 = 0, t12 = 0, t23 = 0, t34 = 0;
  KJ_HIP(hipEventElapsedTime(&t01, ctx->ev[0], ctx->ev[1]));
  KJ_HIP(hipEventElapsedTime(&t12, ctx->ev[1], ctx->ev[2]));
  KJ_HIP(hipEventElapsedTime(&t23, ctx->ev[2], ctx->ev[3]));
  KJ_HIP(hipEventElapsedTime(&t34, ctx->ev[3], ctx->ev[4]));
  uint32_t cnt[8] = {0};
  KJ_HIP(hipMemcpy(cnt, ctx->counters.p, sizeof cnt, hipMemcpyDeviceToHost));
  if (getenv("KAIJU_GPU_PRINT_STATS")) {
    unsigned long long acc[6] = {0};
    KJ_HIP(hipMemcpy(acc, static_cast<uint32_t *>(ctx->counters.p) + 8, sizeof acc, hipMemcpyDeviceToHost));
    if (ctx->params.mode == 0)
      fprintf(stderr, "[kj stats] lane-iters %llu step %llu kmer %llu lf %llu | sum over waves of max iters %llu, max passes %llu\n",
              acc[0], acc[1], acc[2], acc[3], acc[4], acc[5]);
    else
      fprintf(stderr, "[kj stats] greedy lane loop, cycles summed over waves: heavy chain %llu, load issue %llu, fast compute+book %llu, "
                      "slow compute %llu, loop head %llu | iterations (sum of last-wave counters) %llu heavy %llu\n",
              acc[0], acc[1], acc[2], acc[3], acc[4], acc[5] >> 32, acc[5] & 0xffffffffull);
  }
#ifdef KJ_PROF
  if (ctx->params.mode == 0) {
    static const char *names[PM_N] = {"HEAD", "LOAD", "LOADFILL", "STEP", "KMER", "LF1", "SA", "META", "FRAG", "FILL", "TAIL", "END_MATCH",
                                      "START_J", "NEXT_FRAG", "LOC_INIT", "LOC_NEXT_SI", "LOC_ROW", "FINISH"};
    unsigned long long pv[3 * PM_N];
    KJ_HIP(hipMemcpy(pv, static_cast<uint8_t *>(ctx->counters.p) + 1024, sizeof pv, hipMemcpyDeviceToHost));
    unsigned long long tot = 0;
    for (int x = 0; x < PM_N; x++) tot += pv[3 * x];
    fprintf(stderr, "[kj prof] section        cycles%%   entries/read  lanes/entry   cycles/entry   (k_mem main launch; n_reads %u, total wave-cycles %llu)\n", ctx->last_n, tot);
    for (int x = 0; x < PM_N; x++)
      fprintf(stderr, "[kj prof] %-13s %7.2f %12.3f %10.1f %12.1f\n", names[x], 100.0 * pv[3 * x] / (double)(tot ? tot : 1),
              (double)pv[3 * x + 1] / (ctx->last_n ? ctx->last_n : 1), (double)pv[3 * x + 2] / (double)(pv[3 * x + 1] ? pv[3 * x + 1] : 1),
              (double)pv[3 * x] / (double)(pv[3 * x + 1] ? pv[3 * x + 1] : 1));
  }
  if (ctx->params.mode == 1) {
    static const char *names[PS_N] = {"HEAD", "AFTER_SEARCH", "VAR_NEXT", "VAR_MATCH", "EVAL_NEXT", "EVAL_MATCH", "POP", "POP_SEG", "FINISH",
                                      "HANDOUT", "LOAD", "LOAD10", "STEP", "KMER", "LF1", "SA", "VM_RANK", "VM_PUSH", "META", "FRAG",
                                      "FILL", "MLOAD", "END_MATCH", "START_J", "LOC_ROW", "TAIL"};
    unsigned long long pv[3 * PS_N];
    KJ_HIP(hipMemcpy(pv, static_cast<uint8_t *>(ctx->counters.p) + 1024, sizeof pv, hipMemcpyDeviceToHost));
    unsigned long long tot = 0;
    for (int x = 0; x < PS_N; x++) tot += pv[3 * x];
    fprintf(stderr, "[kj prof] section        cycles%%   entries/read  lanes/entry   cycles/entry   (n_reads %u, total wave-cycles %llu)\n", ctx->last_n, tot);
    for (int x = 0; x < PS_N; x++)
      fprintf(stderr, "[kj prof] %-13s %7.2f %12.3f %10.1f %12.1f\n", names[x], 100.0 * pv[3 * x] / (double)(tot ? tot : 1),
              (double)pv[3 * x + 1] / (ctx->last_n ? ctx->last_n : 1), (double)pv[3 * x + 2] / (double)(pv[3 * x + 1] ? pv[3 * x + 1] : 1),
              (double)pv[3 * x] / (double)(pv[3 * x + 1] ? pv[3 * x + 1] : 1));
  }
#endif
  stats->n_reads = ctx->last_n;
  if (getenv("KAIJU_GPU_OVF_STATS") && ctx->params.mode == 1) {
    uint32_t why[8] = {0};
    KJ_HIP(hipMemcpy(why, static_cast<uint32_t *>(ctx->counters.p) + 40, sizeof why, hipMemcpyDeviceToHost));
    fprintf(stderr, "[kaiju_gpu] Greedy reads sent to the retry pass, by reason: key/sequence bits %u, queue full %u, original too long %u, "
                    "SEG piece too long %u, variant too long %u, matches per fragment %u, wide interval %u\n",
            why[0], why[1], why[2], why[3], why[4], why[5], why[6]);
  }
  stats->n_overflow_retries = cnt[2];
  stats->n_seg_fragments = cnt[4];
  // bit 0 (a SegRec overflowed in the MEM split of the main pass) is settled by the exact pass, which reports its own
  // failures: 4 = region pool exhausted, 8 = more reads than its list holds
  stats->error_flags = (ctx->kp.seg && ctx->exact_pass) ? (cnt[3] & ~1u) : cnt[3];
  stats->ms_translate = t01;
  stats->ms_seg = t12;
  stats->ms_search = t23;
  stats->ms_retry = t34;
  stats->ms_total = t01 + t12 + t23 + t34;
  return KAIJU_GPU_OK;
}
