// exact_pass.hip — kernels of the exact pass (see exact_pass.h and kj_core.h: BigSeg).
//
// A SegRec of the SEG pass holds 15 regions and 16-bit positions.  A fragment that needs more (long low-complexity-rich
// proteins or contigs) is marked `overflow` there; reads with such a fragment are classified again here, behind the main
// and retry passes: (1) list the reads, (2) stage 1 again into a queue of its own, (3) SEG with region lists of any
// length into a pool of (left, right) pairs, (4) MEM: the split from those, (5) the search by the first-generation lanes.
// All five launches find an empty list unless the batch holds such fragments and then cost a few microseconds.
#include "exact_pass.h"

using namespace kj;

namespace {
constexpr int kXBlock = 256, kXFragBlock = 64, kXSegBlock = 64;

__device__ __forceinline__ void x_load_tables(ConstTables &s_ct, const ConstTables *g_ct) {
  const uint32_t *src = reinterpret_cast<const uint32_t *>(g_ct);
  uint32_t *dst = reinterpret_cast<uint32_t *>(&s_ct);
  for (uint32_t i = threadIdx.x; i < sizeof(ConstTables) / 4; i += blockDim.x) dst[i] = src[i];
  __syncthreads();
}

// one wavefront per fragment: the lanes share the sub-windows of s_Trim (as capi.hip's SEG pass)
struct XCoopWave {
  __device__ __forceinline__ uint64_t *pref() const { return nullptr; }   // (not on the hot path: every sub-window counts its letters)
  __device__ __forceinline__ void sync() const {}
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
}  // namespace

// (1) the reads: every slot of the SEG pass marked `overflow` -> its read, once
__global__ void __launch_bounds__(256)
k_redo_collect(SegQueue sq, uint32_t *bitmap, uint32_t *list, uint32_t *count, uint32_t cap, uint32_t *err) {
  const uint32_t n = min(*sq.count, sq.cap);
  for (uint32_t s = blockIdx.x * 256 + threadIdx.x; s < n; s += gridDim.x * 256) {
    if (!sq.recs[s].overflow) continue;
    const uint32_t r = sq.items[s].read;
    const uint32_t bit = 1u << (r & 31u);
    if (atomicOr(bitmap + (r >> 5), bit) & bit) continue;
    const uint32_t k = atomicAdd(count, 1u);
    if (k < cap) list[k] = r;
    else { atomicSub(count, 1u); atomicOr(err, 8u); }         // (more such reads than the list holds: they stay flagged)
  }
}

// (2) stage 1 again for the listed reads (the MEM pass has replaced their fragment lists by the split ones); flagged
// fragments go to the queue of the exact pass; the hit record is cleared
__global__ void __launch_bounds__(kXFragBlock)
k_redo_fragments(const ConstTables *__restrict__ g_ct, Params p, SegTables st, Batch b, SegQueue sq2, uint32_t *err,
                 const uint32_t *list, const uint32_t *count) {
  __shared__ ConstTables s_ct;
  __shared__ int32_t s_entg[17];
  __shared__ uint8_t s_code[256];
  const uint32_t n = *count;
  if (n == 0) return;
  if (threadIdx.x < 17) s_entg[threadIdx.x] = st.ent_g32[threadIdx.x];
  for (uint32_t i = threadIdx.x; i < 256; i += kXFragBlock) s_code[i] = 0;
  x_load_tables(s_ct, g_ct);
  if (threadIdx.x < 20) protein_code_entry(s_ct, threadIdx.x, s_code);
  __syncthreads();
  uint32_t e = 0;
  for (uint32_t k = blockIdx.x * kXFragBlock + threadIdx.x; k < n; k += gridDim.x * kXFragBlock) {
    const uint32_t r = list[k];
    Hit *h = b.hits + r;
    h->best = h->n_ids = h->flags = h->reserved = 0;
    for (int q = 0; q < kMaxIds; q++) h->taxid[q] = 0;
    if (p.flags & kParamProtein) build_fragments_protein(s_ct, s_code, p, TrigCtx{s_entg, st.ent_locut32}, b, sq2, r, &e);
    else build_fragments(s_ct, p, TrigCtx{s_entg, st.ent_locut32}, b, sq2, r, &e, nullptr, 0, 0);
  }
  if (e) atomicOr(err, e);
}

// (3) SEG of the queued fragments with lists of any length: one wavefront per fragment, scratch in device memory
__global__ void __launch_bounds__(kXSegBlock)
k_redo_seg(SegTables st, Batch b, SegQueue sq2, BigSeg big, int32_t *work_all, uint8_t *cls_all, uint32_t cap_ints,
           uint32_t cls_bytes, uint32_t *err) {
  __shared__ int64_t s_entg[13];
  __shared__ double s_lnf[kSegLnf];
  const uint32_t n = min(*sq2.count, sq2.cap);
  if (n == 0) return;
  if (threadIdx.x < 13) s_entg[threadIdx.x] = st.ent_g[threadIdx.x];
  if (threadIdx.x < kSegLnf) s_lnf[threadIdx.x] = st.lnfact[threadIdx.x];
  __syncthreads();
  const SegCtx cx = seg_ctx(st, s_entg, s_lnf);
  const XCoopWave coop;
  int32_t *work = work_all + (size_t)blockIdx.x * 4 * cap_ints;
  uint8_t *cls = cls_all + (size_t)blockIdx.x * cls_bytes;
  uint32_t e = 0;
  for (uint32_t s = blockIdx.x; s < n; s += gridDim.x)        // trip count is uniform over the block
    seg_compute_big(cx, coop, b, sq2, big, s, work, (int)cap_ints, cls, &e, [] { __syncthreads(); });
  if (e && threadIdx.x == 0) atomicOr(err, e);
}

// (4) MEM: the split of the listed reads from the pool
__global__ void __launch_bounds__(kXFragBlock)
k_redo_apply(const ConstTables *__restrict__ g_ct, Params p, Batch b, BigSeg big, const uint32_t *list, const uint32_t *count) {
  __shared__ ConstTables s_ct;
  const uint32_t n = *count;
  if (n == 0) return;
  x_load_tables(s_ct, g_ct);
  for (uint32_t k = blockIdx.x * kXFragBlock + threadIdx.x; k < n; k += gridDim.x * kXFragBlock)
    seg_apply_mem_big(s_ct, p, b, big, list[k]);
}

// (5) the search of the listed reads by the first-generation lanes with worst-case scratch
__global__ void __launch_bounds__(kXBlock)
k_redo_mem(DevIndex ix, Params p, Batch b, WorkList wl, SIEntry *si_all, uint32_t si_cap, VerboseOut vb) {
  __shared__ __attribute__((aligned(16))) uint8_t s_win[kXBlock * kWinStride];
  if (*wl.n_items_ptr == 0) return;
  const uint64_t lane = (uint64_t)blockIdx.x * kXBlock + threadIdx.x;
  LaneScratch ls;
  ls.si = si_all + lane * si_cap;
  ls.si_cap = si_cap;
  ls.win = s_win + threadIdx.x * kWinStride;
  mem_lane<uint64_t>(ix, p, b, wl, ls, vb);
}
__global__ void __launch_bounds__(kXBlock)
k_redo_greedy(DevIndex ix, const ConstTables *__restrict__ g_ct, Params p, SegQueue sq2, Batch b, WorkList wl, GItem *pool,
              uint16_t *ord, GMatch *matches, GBest *best, GBestV *bestv, uint32_t pool_cap, uint32_t match_cap, VerboseOut vb,
              BigSeg big) {
  __shared__ __attribute__((aligned(16))) uint8_t s_win[kXBlock * kWinStride];
  __shared__ ConstTables s_ct;
  if (*wl.n_items_ptr == 0) return;
  x_load_tables(s_ct, g_ct);
  const uint64_t lane = (uint64_t)blockIdx.x * kXBlock + threadIdx.x;
  GreedyScratch gs;
  gs.pool = pool + lane * pool_cap; gs.pool_cap = pool_cap;
  gs.ord = ord + lane * pool_cap;
  gs.matches = matches + lane * match_cap; gs.match_cap = match_cap;
  gs.best = best + lane * 64;
  gs.win = s_win + threadIdx.x * kWinStride;
  gs.bestv = bestv ? bestv + lane * 64 : nullptr;
  greedy_lane(ix, s_ct, p, sq2, b, wl, gs, vb, &big);
}

hipError_t kj_launch_exact_pass(const ExactPassLaunch &a) {
  hipStream_t s = a.stream;
  uint32_t *cnt = a.cnt;
  hipLaunchKernelGGL(k_redo_collect, dim3(a.n_cu), dim3(256), 0, s, a.sq, a.bitmap, a.list, cnt + 5, a.list_cap, cnt + 3);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_redo_fragments, dim3(64), dim3(kXFragBlock), 0, s, a.d_ct, a.p, a.st, a.b, a.sq2, cnt + 3, a.list, cnt + 5);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  hipLaunchKernelGGL(k_redo_seg, dim3(a.seg_blocks), dim3(kXSegBlock), 0, s, a.st, a.b, a.sq2, a.big, a.work, a.cls, a.cap_ints,
                     a.cls_bytes, cnt + 3);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  WorkList wl;
  wl.counter = cnt + 7; wl.reads = a.list; wl.n_items_ptr = cnt + 5; wl.n_items = 0;
  wl.retry_list = nullptr; wl.retry_count = nullptr;
  if (a.p.mode == 0) {
    hipLaunchKernelGGL(k_redo_apply, dim3(64), dim3(kXFragBlock), 0, s, a.d_ct, a.p, a.b, a.big, a.list, cnt + 5);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    hipLaunchKernelGGL(k_redo_mem, dim3(a.blocks_search), dim3(kXBlock), 0, s, a.ix, a.p, a.b, wl, a.si, a.si_cap, a.vb);
  } else {
    hipLaunchKernelGGL(k_redo_greedy, dim3(a.blocks_search), dim3(kXBlock), 0, s, a.ix, a.d_ct, a.p, a.sq2, a.b, wl, a.g_pool,
                       a.g_ord, a.g_matches, a.g_best, a.g_bestv, a.g_pool_cap, a.g_match_cap, a.vb, a.big);
  }
  return hipGetLastError();
}
