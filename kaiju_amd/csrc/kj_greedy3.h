// kj_greedy3.h - Greedy search (kaiju -a greedy), third generation: the reads of a thread block live in LDS ROWS and the
// wavefronts of the block pull rows BY KIND, so that the 64 lanes of a wavefront run the same piece of the algorithm.
//
// Why (profiles/r05_gprof/greedy_prof.txt): greedy_lane2 (kj_core.h) binds a read to a lane.  A read is in one of some
// fifteen states (extend a match, look up a k-mer, pop the queue, walk the match list, rank the substitutes, queue a variant,
// fill the window ...), the 64 reads of a wavefront are in all of them at once, and every iteration of the wavefront pays for
// the code of every state with 6 - 18 of its 64 lanes at work: 2 160 wave instructions per iteration for about 47 lane steps.
// The lane steps themselves are few (184 per read); what they cost is the divergence.
//
// Here a read's whole state is a row of LDS (kG3RowBytes: search interval, match at hand, queue priorities, peptide window,
// the slow-part state greedy_lane2 already kept there) plus its scratch in device memory (queue items, match records), and
// nothing of it lives in registers across iterations.  Every row carries a CLASS - what the read needs next - in an LDS word.
// A wavefront (no barriers, the wavefronts of a block run independently):
//   1. picks the class with the most waiting rows (counts in LDS),
//   2. claims up to 64 rows of that class (scan of the class words, ballot + prefix count, one compare-and-swap per lane),
//   3. runs that class's piece of the algorithm on them - the code of greedy_lane2's sections, unchanged in what it computes -
//   4. gives the rows back under their new classes.
// One block of kG3Threads threads per CU owns kG3Pool rows: with fifteen classes a class holds some thirty rows on average
// and the fullest one, which is the one taken, usually a wavefront's worth.
//
// The per-read algorithm, its order and its quirks are those of greedy_lane2 (ConsumerThread.cpp:424-541 classify_greedyblosum,
// :346-395 addAllMismatchVariantsAtPosSI, :751-797 eval_match_scores, :272-342 getNextFragment; bwt.c:261-336 maxMatches /
// maxMatches_withStart): the reference's sequential best-first search per read, bit for bit.  Narrow indexes (below 2^32 rows)
// with k-mer lines; wide indexes keep greedy_lane2<.., true>.  Reads beyond the row's 15/16-bit fields (fragments of 2^15
// residues and more, 2^16 fragments) go to the retry pass like every other overflow.
#pragma once
#include "kj_core.h"

namespace kj {

#ifdef KJ_G_SMALL
constexpr int kG3PrioWords = 4, kG3MqWords = 1;
#else
constexpr int kG3PrioWords = 12, kG3MqWords = 4;          // kGSlots priorities (16-byte reads), kGMaxM lengths
#endif
static_assert(kG3PrioWords >= kGSlots && kG3PrioWords % 4 == 0 && kG3MqWords * 2 >= kGMaxM, "row pieces");
constexpr int kG3WinWords = 17;                            // 64 bytes of window + 4 (odd stride)
constexpr int kG3StWords = 49;                             // the state proper (odd stride: dword / halfword / byte accesses)
constexpr int kG3RowBytes = 4 * (kG3PrioWords + kG3WinWords + kG3MqWords + kG3StWords);
#ifndef KJ_G3_POOL
#ifdef KJ_PROF
#define KJ_G3_POOL 448                                     // (the section profiler's counters need LDS too)
#else
#define KJ_G3_POOL 480
#endif
#endif
constexpr int kG3Pool = KJ_G3_POOL;                        // rows per block (one block per CU)
constexpr int kG3Threads = 512;                            // most threads a block is launched with (KAIJU_GPU_G3_THREADS: fewer)
constexpr uint32_t kG3MaxLen = 0x7fffu;                    // fragment lengths and positions live in 15 bits here
// classes: what a row needs next
enum G3Cls : uint32_t { C3_FAST, C3_VMULTI, C3_DESC, C3_FILL, C3_MLOAD, C3_AFTER, C3_VARNEXT, C3_VARMATCH, C3_EVALNEXT, C3_EVALMATCH,
                        C3_POP, C3_FINISH, C3_IDLE, C3_EXIT, C3_N };
constexpr uint32_t kG3Busy = 0xffu;
constexpr size_t kG3LdsBytes = (size_t)kG3Pool * kG3RowBytes + (size_t)kG3Pool * 4 + 32 * 4 + (kG3Threads / 64) * 64 * 2 + sizeof(ConstTables);

struct G3Ctx {
  // LDS (device) / plain arrays (host emulation: one row)
  uint32_t *prio, *win, *mq, *st;            // [pool][words]
  uint32_t *cls;                             // [pool] class of the row, kG3Busy while a wavefront holds it
  uint32_t *cnt;                             // [32] rows per class (hints; C3_EXIT exact)
  uint16_t *tmp;                             // [64] this wavefront's compaction scratch
  // device memory, bases of all rows (greedy_lane2's scratch, one piece per row)
  u128 *pool; uint32_t *prio_ext; GMatch2 *matches; uint16_t *mq_ext; GBest2 *best;
  uint32_t row0;                             // global number of this block's row 0
  uint32_t npool;                            // rows of this block (kG3Pool; host: 1)
  uint32_t split;                            // 1: a wavefront runs one slow block per pull (rows re-enter the pool between blocks)
  unsigned long long *prof;
};

KJ_HD uint32_t g3_class_of(int kind, int bk_pend) {
  switch (kind) {
    case G_STEP: case G_KMER: case G_PROBE: return C3_FAST;
    case G_VMULTI: return C3_VMULTI;
    case G_META: case G_FRAG: return C3_DESC;
    case G_FILL: case G_POPITEM: return C3_FILL;
    case G_MLOAD: return C3_MLOAD;
    case G_IDLE: return C3_IDLE;
    case G_EXIT: return C3_EXIT;
    default: break;
  }
  switch (bk_pend) {
    case GB_AFTER_SEARCH: return C3_AFTER;
    case GB_VAR_NEXT: return C3_VARNEXT;
    case GB_VAR_MATCH: return C3_VARMATCH;
    case GB_EVAL_NEXT: return C3_EVALNEXT;
    case GB_EVAL_MATCH: return C3_EVALMATCH;
    case GB_POP: return C3_POP;
    default: return C3_FINISH;             // GB_FINISH, GB_DONE
  }
}

// ---- the pool (device) ---------------------------------------------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
// the class to serve: the fullest one (at most a wavefront's worth counts; ties go round with `rot`), C3_N if nothing waits
__device__ __forceinline__ uint32_t g3_choose(const G3Ctx &gx, uint32_t rot) {
  const uint32_t l = threadIdx.x & 63u;
  uint32_t key = 0;
  if (l < (uint32_t)C3_EXIT) {
    uint32_t n = __hip_atomic_load(gx.cnt + l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (n > 0x7fffffffu) n = 0;                          // (a count may dip below zero for a moment: claims are counted after the fact)
    if (n > 64u) n = 64u;
    key = n ? (n << 8 | ((l - rot) & 15u) << 4 | l) : 0u;
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) { const uint32_t ok = (uint32_t)__shfl_xor((int)key, o, 64); key = ok > key ? ok : key; }
  key = (uint32_t)__builtin_amdgcn_readfirstlane((int)key);
  return key ? (key & 15u) : (uint32_t)C3_N;
}
// claims up to 64 rows of class c; returns the lane's row or ~0
__device__ __forceinline__ uint32_t g3_pull(const G3Ctx &gx, uint32_t c, uint32_t start) {
  const uint32_t l = threadIdx.x & 63u;
  volatile uint16_t *tmp = gx.tmp;
  constexpr int kRounds = (kG3Pool + 63) / 64;
  uint32_t idx[kRounds];
  bool hit[kRounds];
#pragma unroll
  for (int k = 0; k < kRounds; k++) {
    uint32_t x = start + 64u * k + l;
    if (x >= (uint32_t)kG3Pool) x -= (uint32_t)kG3Pool;
    idx[k] = x;
    const bool in = 64u * k + l < (uint32_t)kG3Pool;
    hit[k] = in && __hip_atomic_load(gx.cls + (in ? x : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == c;
  }
  uint32_t total = 0;
#pragma unroll
  for (int k = 0; k < kRounds; k++) {
    const uint64_t m = __ballot(hit[k]);
    const uint32_t at = total + kj_rank_below(m);
    if (hit[k] && at < 64u) tmp[at] = (uint16_t)idx[k];
    total += popc64(m);
  }
  if (total > 64u) total = 64u;
  __builtin_amdgcn_wave_barrier();
  uint32_t v = ~0u;
  if (l < total) {
    const uint32_t cand = tmp[l];
    uint32_t expect = c;
    if (__hip_atomic_compare_exchange_strong(gx.cls + cand, &expect, kG3Busy, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))
      v = cand;
  }
  const uint64_t got = __ballot(v != ~0u);
  if (got && l == (uint32_t)__builtin_ctzll(got))
    __hip_atomic_fetch_sub(gx.cnt + c, popc64(got), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  return v;
}
// the rows of the active lanes go back under their new classes (one count update per class among them)
__device__ __forceinline__ void g3_release(const G3Ctx &gx, uint32_t v, uint32_t c) {
  uint64_t todo = __ballot(true);
  const uint32_t l = threadIdx.x & 63u;
  while (todo) {
    const uint32_t leader = (uint32_t)__builtin_ctzll(todo);
    const uint32_t cl = (uint32_t)__builtin_amdgcn_readlane((int)c, (int)leader);
    const uint64_t m = __ballot(c == cl);
    if (l == leader) __hip_atomic_fetch_add(gx.cnt + cl, popc64(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    todo &= ~m;
  }
  __hip_atomic_store(gx.cls + v, c, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
#endif

template <bool COUNT = false>
KJ_HD void greedy_lane3(const DevIndex &ix, const ConstTables &ct, const Params &p, const SegQueue &sq,
                        const Batch &b, const WorkList &wl, const G3Ctx &gx) {
  typedef uint32_t P;
  uint32_t oc[kOpcN];
  if constexpr (COUNT) for (int x = 0; x < kOpcN; x++) oc[x] = 0;
  const G3Ctx &gs = gx;                         // (KJ_P marks)
  (void)gs;
  const uint32_t n_items = wl.n_items_ptr ? *wl.n_items_ptr : wl.n_items;
  const uint32_t ktab = ix.kline_k;
  const uint32_t kk = (ktab >= 2 && ktab <= p.seed_length && p.seed_length >= 3 && ix.kline) ? ktab : 0;
  const RankBlock64 *const blk0 = ix.blocks64;
  uint32_t kpow = 1;
  for (uint32_t q = 2; q < kk; q++) kpow *= 20u;            // 20^(kk-2): digit of w[j-1] in the line number
  uint64_t dg0 = 0, dg1 = 0;                                // BLOSUM62 diagonal by index-alphabet code, 4 bits each (values 4..11)
  for (int x = 0; x < 16; x++) dg0 |= (uint64_t)((uint32_t)ct.diag_idx[x] & 15u) << (4 * x);
  for (int x = 16; x < 32; x++) dg1 |= (uint64_t)((uint32_t)ct.diag_idx[x] & 15u) << (4 * (x - 16));
#ifdef __HIP_DEVICE_COMPILE__
  dg0 = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)dg0) |
        (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(dg0 >> 32)) << 32;
  dg1 = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)dg1) |
        (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(dg1 >> 32)) << 32;
  kpow = (uint32_t)__builtin_amdgcn_readfirstlane((int)kpow);
  uint32_t rot = (threadIdx.x >> 6) * 2u, scan_at = ((threadIdx.x >> 6) * 64u) % (uint32_t)kG3Pool;
#endif
  auto diag = [&](uint32_t cc) -> uint32_t { return (uint32_t)(((cc & 16u) ? dg1 : dg0) >> (4u * (cc & 15u))) & 15u; };

  for (;;) {
    KJ_P(PS_HEAD);
    // ---- pull: a class, and up to 64 of its rows ----
    uint32_t C, v;
#if defined(__HIP_DEVICE_COMPILE__)
    C = g3_choose(gx, rot);
    rot = (rot + 1u) & 15u;
    if (C == (uint32_t)C3_N) {
      if (__hip_atomic_load(gx.cnt + C3_EXIT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= (uint32_t)kG3Pool) break;
      KJ_P(PS_LF1);                               // (profile: waiting for rows)
      __builtin_amdgcn_s_sleep(8);
      continue;
    }
    v = g3_pull(gx, C, scan_at);
    scan_at += 64u; if (scan_at >= (uint32_t)kG3Pool) scan_at -= (uint32_t)kG3Pool;
#else
    v = 0;
    C = gx.cls[0];
    if (C == (uint32_t)C3_EXIT) break;
#endif
    if (v == ~0u) continue;                      // (a lane without a row sits this iteration out; the wavefront goes on below)

    // ---- the row ----
    uint32_t *const prio = gx.prio + (size_t)v * kG3PrioWords;
    uint8_t *const win = reinterpret_cast<uint8_t *>(gx.win + (size_t)v * kG3WinWords);
    uint16_t *const mq = reinterpret_cast<uint16_t *>(gx.mq + (size_t)v * kG3MqWords);
    uint32_t *const st_ = gx.st + (size_t)v * kG3StWords;
    const size_t gl = (size_t)gx.row0 + v;      // the row's number among all rows: its scratch in device memory
    uint32_t &lo = st_[0], &hi = st_[1], &sz_i = st_[2], &sz_q = st_[3], &m_lo = st_[4], &m_len = st_[5], &kidx = st_[6], &r = st_[7],
             &fbase = st_[8], &pep16 = st_[9], &best = st_[10], &flags = st_[11], &t_start = st_[12];
    int32_t &t_diff = reinterpret_cast<int32_t &>(st_[13]);
    uint32_t &sp0 = st_[14], &sp1 = st_[15], &sp2 = st_[16], &sp3 = st_[17], &sa0 = st_[18], &sa1 = st_[19], &on_start = st_[20],
             &on_kl = st_[21], &ext_max = st_[22], &on_flags = st_[23], &b0lo = st_[24], &b0len = st_[25], &vscore = st_[26],
             &qseq = st_[27], &acc = st_[28], &tail = st_[29], &kacc = st_[30], &t_tot = st_[31], &t_msum = st_[32];
    int32_t &i = reinterpret_cast<int32_t &>(st_[33]), &j = reinterpret_cast<int32_t &>(st_[34]);
    uint16_t *const h_ = reinterpret_cast<uint16_t *>(st_ + 35);
    uint16_t &last_qi = h_[0], &t_len = h_[1], &fo = h_[2], &nf = h_[3], &t_matchlen = h_[4], &m_qi = h_[5], &m_ql = h_[6], &m_dsum = h_[7],
             &m_psum = h_[8], &nm = h_[9], &qn = h_[10], &qlive = h_[11], &mx = h_[12], &vlen = h_[13], &pslot = h_[14], &fill_top = h_[21],
             &wq = h_[22], &bits_ = h_[23];
    int16_t *const hs_ = reinterpret_cast<int16_t *>(h_);
    int16_t &vi_v = hs_[15], &vi_x = hs_[16], &vi_head = hs_[17], &ev_v = hs_[18], &ev_x = hs_[19], &ev_v1 = hs_[20];
    uint8_t *const b_ = reinterpret_cast<uint8_t *>(st_ + 47);
    uint8_t &kind = b_[0], &bk_pend = b_[1], &c = b_[2], &cj = b_[3], &nbest = b_[4], &t_nmm = b_[5], &vorig = b_[7];
    int8_t &ev_pass = reinterpret_cast<int8_t &>(b_[6]);
    // (flags and small enumerations: one halfword, unpacked here, packed again where the row is given back)
    const uint32_t bits0 = bits_;
    bool kroll = bits0 & 1u, skipj = bits0 & 2u, ovf = bits0 & 4u, m_ovf = bits0 & 8u, fill_pref = bits0 & 16u, ev_done = bits0 & 32u;
    uint32_t ml_for = (bits0 >> 6) & 1u;
    int fill_ret = (int)((bits0 >> 7) & 3u), vi_phase = (int)((bits0 >> 9) & 3u);
#define flen ((int)t_len)
    const uint8_t *const pepr = b.pep + ((uint64_t)pep16 << 4);

    u128 *const g_pool = gx.pool + gl * (8 * kGSlotsAll);
    uint32_t *const g_prio_ext = gx.prio_ext + gl * (kGSlotsAll - kGSlots);
    GMatch2 *const g_matches = gx.matches + gl * kGMaxMAll;
    uint16_t *const g_mq_ext = gx.mq_ext + gl * (kGMaxMAll - kGMaxM);
    GBest2 *const g_best = gx.best + gl * 64;
    Hit *const hit = b.hits + r;

    auto in_win = [&](int pos) -> bool { return pos >= (int)wq && pos < (int)wq + kWin; };
    auto mq_get = [&](uint32_t x) -> int { return (int)(x < (uint32_t)kGMaxM ? mq[x] : g_mq_ext[x - kGMaxM]); };
    auto mq_max_below = [&](int bound) -> int {
      int vv = -1;
      for (uint32_t x = 0; x < nm; x++) { const int q = mq_get(x); if (q < bound && q > vv) vv = q; }
      return vv;
    };
    // the queue's priorities: greedy_lane2's layout (slots 0 .. kGSlots-1 in the row, the others append-only in device memory)
    auto push_slot = [&](uint32_t key, uint32_t seq) -> uint32_t {
      if (key > 0xffffu || seq >= 0xfffeu) { ovf = true; KJ_OVF(wl, 0); return ~0u; }
      const uint32_t pr = key << 16 | (0xffffu - seq);
      const uint32_t nl = qn < (uint32_t)kGSlots ? qn : (uint32_t)kGSlots;
      uint32_t slot;
      if (qlive < nl) {
        slot = 0;
        if constexpr (kGSlots % 4 == 0) {
          bool got = false;
          for (uint32_t q = 0; q < nl && !got; q += 4) {
            const u128 vv = *reinterpret_cast<const u128 *>(prio + q);
            const uint32_t e0 = (uint32_t)vv.x, e1 = (uint32_t)(vv.x >> 32), e2 = (uint32_t)vv.y, e3 = (uint32_t)(vv.y >> 32);
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
          slot = kGSlots;
          while (slot < (uint32_t)kGSlotsAll && g_prio_ext[slot - kGSlots] != 0) slot++;
          if (slot >= (uint32_t)kGSlotsAll) { ovf = true; KJ_OVF(wl, 1); return ~0u; }
        }
        g_prio_ext[slot - kGSlots] = pr;
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
      const int sc = (int)m_dsum + t_diff;
      const uint32_t score = sc > 0 ? (uint32_t)sc : 0u;
      if (score < p.min_score) return;
      if (score > best) { best = score; nbest = 0; }
      if (score == best) {
        if (nbest < p.max_matches_SI && nbest < 64) {
          if (nbest == 0) { b0lo = m_lo; b0len = m_len; }
          else { GBest2 gb; gb.lo = m_lo; gb.len = m_len; g_best[nbest] = gb; }
          nbest++;
        } else flags |= kHitSiCap;
      }
    };

    if constexpr (COUNT) {
      oc[kOpcLaneIters]++;
#if defined(__HIP_DEVICE_COMPILE__)
      if ((threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(__ballot(true))) oc[kOpcIters]++;
#else
      oc[kOpcIters]++;
#endif
    }

    int bk = GB_NONE;
    bool probe_now = false;
    if (C >= (uint32_t)C3_AFTER && C <= (uint32_t)C3_FINISH) {
      // ---- the slow bookkeeping (greedy_lane2: H0), from the row's pending block up to its next memory access ----
      bk = bk_pend;
      while (bk != GB_NONE) {
        if (gx.split && g3_class_of(G_WAIT, bk) != C) { bk_pend = (uint8_t)bk; kind = G_WAIT; bk = GB_NONE; break; }
        if (bk == GB_AFTER_SEARCH) {
          KJ_P(PS_AFTER_SEARCH);
          if (nm == 0 || m_ovf) bk = GB_POP;
          else if (p.mismatches > 0 && t_nmm < p.mismatches) {
            vi_phase = 0;
            vi_v = (int16_t)(nm == 1 ? (int)m_ql : mq_max_below(0x7fffffff));
            bk = GB_VAR_NEXT;
          } else { ev_pass = -1; bk = GB_EVAL_NEXT; }
          continue;
        }
        if (bk == GB_VAR_NEXT) {
          KJ_P(PS_VAR_NEXT);
          bool have = false;
          if (nm == 1) {
            if (vi_phase == 0) { vi_phase = 2; have = true; }
          } else if (vi_phase == 0) {
            if (vi_v >= 0) {
              uint32_t head = nm, cntv = 0;
              int below = -1;
              for (uint32_t x = 0; x < nm; x++) {
                const int q = mq_get(x);
                if (q == vi_v) { if (head == nm) head = x; cntv++; }
                else if (q < vi_v && q > below) below = q;
              }
              mx = (uint16_t)head; vi_head = (int16_t)head; have = true;
              if (cntv >= 2) { vi_phase = 1; vi_x = (int16_t)nm; }
              else { vi_v = (int16_t)below; if (vi_v < 0) vi_phase = 2; }
            }
          } else if (vi_phase == 1) {
            int x = vi_x - 1;
            while (x > vi_head && mq_get((uint32_t)x) != vi_v) x--;
            if (x > vi_head) { mx = (uint16_t)x; vi_x = (int16_t)x; have = true; }
            else vi_phase = 2;
          }
          if (!have) { ev_pass = -1; bk = GB_EVAL_NEXT; }
          else if (nm == 1) bk = GB_VAR_MATCH;
          else { ml_for = 0; kind = G_MLOAD; bk = GB_NONE; }
          continue;
        }
        if (bk == GB_VAR_MATCH) {
          KJ_P(PS_VAR_MATCH);
          const uint32_t mre = (uint32_t)m_qi + m_ql - 1u;
          if (!(m_qi > 0 && mre + 1u >= p.m)) { bk = GB_VAR_NEXT; continue; }          // :469
          else if (!in_win((int)m_qi - 1)) {
            fill_top = (uint16_t)(m_qi - 1); fill_ret = FR_VARM; fill_pref = false; kind = G_FILL; bk = GB_NONE;
          } else {
            // addAllMismatchVariantsAtPosSI(t, qi-1, erase_pos, it), :346-395
            vlen = (uint16_t)((mre < (uint32_t)flen - 1u) ? mre + 1u : (uint32_t)flen);
            vorig = ct.idx_to_aa[win[(int)m_qi - 1 - (int)wq]];
            const int sc = (int)m_psum + t_diff;
            const uint32_t cs = sc > 0 ? (uint32_t)sc : 0u;
            vscore = cs - (uint32_t)(int32_t)ct.b62[vorig][vorig];             // unsigned wrap as in :363
            const int32_t thr0 = (int32_t)best > (int32_t)p.min_score ? (int32_t)best : (int32_t)p.min_score;
            if ((int32_t)(vscore + (uint32_t)(int32_t)ct.b62[vorig][ct.subst[vorig][0]]) < thr0) { qseq += 19; bk = GB_VAR_NEXT; continue; }
            kind = G_VMULTI; bk = GB_NONE;
          }
          continue;
        }
        if (bk == GB_EVAL_NEXT) {
          KJ_P(PS_EVAL_NEXT);
          if (nm == 1) {
            if (m_ql >= p.m) eval_match();
            bk = GB_POP;
          } else {
            if (ev_pass < 0) {
              ev_v1 = (int16_t)mq_max_below(0x7fffffff);
              if (ev_v1 < (int)p.m) bk = GB_POP;                               // :482
              else { ev_pass = 0; ev_v = ev_v1; ev_x = -1; ev_done = false; }
            }
            while (bk == GB_EVAL_NEXT) {
              if (ev_pass == 0) {
                int head = -1, nxt = -1, nv = -1;
                for (uint32_t x = 0; x < nm; x++) {
                  const int q = mq_get(x);
                  if (q == ev_v) { if (head < 0) head = (int)x; else if (nxt < 0 && (int)x > ev_x) nxt = (int)x; }
                  else if (q < ev_v && q > nv) nv = q;
                }
                if (nxt >= 0) { ev_x = (int16_t)nxt; mx = (uint16_t)nxt; ml_for = 1; kind = G_MLOAD; bk = GB_NONE; }
                else if (nv < 0 || nv < (int)p.m) ev_pass = 1;
                else { ev_v = (int16_t)nv; ev_x = -1; }
              } else if (ev_done) bk = GB_POP;
              else {
                uint32_t head = nm;
                int nv = 0x7fffffff;
                for (uint32_t x = 0; x < nm; x++) {
                  const int q = mq_get(x);
                  if (q == ev_v) { if (head == nm) head = x; }
                  else if (q > ev_v && q < nv) nv = q;
                }
                mx = (uint16_t)head; ml_for = 1; kind = G_MLOAD; bk = GB_NONE;
                if (ev_v == ev_v1) ev_done = true; else ev_v = (int16_t)nv;
              }
            }
          }
          continue;
        }
        if (bk == GB_EVAL_MATCH) { KJ_P(PS_EVAL_MATCH); eval_match(); bk = GB_EVAL_NEXT; continue; }
        if (bk == GB_POP) {
          KJ_P(PS_POP);
          // getNextFragment(best_match_score), ConsumerThread.cpp:272-342
          uint32_t dbest = 0, dslot = 0, ext_second = 0;
          {
            const uint32_t nl = qn < (uint32_t)kGSlots ? qn : (uint32_t)kGSlots;
            if constexpr (kGSlots % 4 == 0) {
              for (uint32_t s = 0; s < nl; s += 4) {
                const u128 vv = *reinterpret_cast<const u128 *>(prio + s);
                const uint32_t e0 = (uint32_t)vv.x, e1 = (uint32_t)(vv.x >> 32), e2 = (uint32_t)vv.y, e3 = (uint32_t)(vv.y >> 32);
                if (e0 > dbest) { dbest = e0; dslot = s; }
                if (e1 > dbest) { dbest = e1; dslot = s + 1u; }
                if (e2 > dbest) { dbest = e2; dslot = s + 2u; }
                if (e3 > dbest) { dbest = e3; dslot = s + 3u; }
              }
            } else
              for (uint32_t s = 0; s < nl; s++) { const uint32_t pr = prio[s]; if (pr > dbest) { dbest = pr; dslot = s; } }
            if (ext_max > dbest) {
              const uint32_t next = qn - (uint32_t)kGSlots;
              uint32_t e1 = 0, s1 = 0, e2 = 0;
              for (uint32_t base = 0; base < next; base += 4 * kGExtPass) {
                const u128 *src = reinterpret_cast<const u128 *>(g_prio_ext + base);
                u128 vv[kGExtPass];
#pragma unroll
                for (int q = 0; q < kGExtPass; q++) vv[q] = (base + 4u * q < next) ? src[q] : u128{0, 0};
#pragma unroll
                for (int q = 0; q < kGExtPass; q++) {
                  const uint32_t w[4] = {(uint32_t)vv[q].x, (uint32_t)(vv[q].x >> 32), (uint32_t)vv[q].y, (uint32_t)(vv[q].y >> 32)};
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
              else { g_prio_ext[dslot - kGSlots] = 0; ext_max = ext_second; }
              pslot = (uint16_t)dslot; kind = G_POPITEM; bk = GB_NONE;
            }
            else {
              t_start = on_start; t_len = (uint16_t)on_len; t_diff = 0; t_matchlen = 0; t_tot = on_key; t_msum = 0; t_nmm = 0;
              const uint32_t oflags = on_flags;
              if (on_key >= 0xffffu || on_len >= kG3MaxLen) { ovf = true; KJ_OVF(wl, 2); }
              fo++;
              if (p.seg && !(oflags & kFragChecked)) {
                // SEG found regions in this fragment: the parent is dropped, its unmasked pieces are queued, and the next
                // fragment is popped (:291-334)
                const uint32_t slot = oflags >> kFragSlotShift;
                KJ_P(PS_POP_SEG);
                if (slot) {
                  const SegRec &rec = sq.recs[slot - 1];
                  if (rec.overflow) flags |= kHitInternalOverflow;
                  Frag f; f.start = t_start; f.len = on_len; f.key = on_key; f.flags = 0;
                  seg_split(ct, p, rec, pepr, f, [&](const Frag &q) {
                    if (q.len >= kG3MaxLen) { ovf = true; KJ_OVF(wl, 3); return; }
                    const uint32_t sl = push_slot(q.key, qseq);
                    if (sl == ~0u) return;
                    qseq++;
                    u128 *dst = g_pool + 8 * sl;
                    u128 vv;
                    vv.x = 0; vv.y = q.key | (uint64_t)q.start << 32; dst[0] = vv;
                    vv.x = q.len; vv.y = q.key; dst[1] = vv;
                    vv.x = vv.y = 0; dst[2] = vv; dst[3] = vv;   // no substitutions, no window
                  });
                }
                if (fo < nf) {
                  const Frag nx = b.frags[fbase + fo];
                  on_start = nx.start; on_kl = (nx.len > 0xffffu ? 0xffffu : nx.len) | (nx.key > 0xffffu ? 0xffffu : nx.key) << 16; on_flags = nx.flags;
                }
                continue;                                   // bk stays GB_POP
              }
              nm = 0; kroll = false; skipj = false;
              j = flen - 1; tail = 0;                       // maxMatches(seq, len, seed_length, 0), bwt.c:261-296
              i = flen;
              fill_top = (uint16_t)j; fill_ret = FR_START_J; fill_pref = true; kind = G_FILL; bk = GB_NONE;
            }
          }
          continue;
        }
        if (bk == GB_FINISH) {
          KJ_P(PS_FINISH);
          uint32_t nids = 0;
          hit->reserved = 0;
          if (ovf || m_ovf) {
            hit->best = 0;
            if (wl.retry_list) { wl.retry_list[append_slot(wl.retry_count)] = r; flags = kHitRetry; }
            else flags = kHitInternalOverflow;
          } else {
            hit->best = nbest ? best : 0u;
            // the best matches go into the hit record (row | length, list order): k_mem_locate* turn them into ids
            if (nbest > (uint32_t)kMaxIds) {
              hit->best = 0;
              if (wl.retry_list) { wl.retry_list[append_slot(wl.retry_count)] = r; flags = kHitRetry; }
              else flags = kHitInternalOverflow;
            } else if (nbest) {
              hit->taxid[0] = (uint64_t)b0lo | (uint64_t)b0len << 32;
              for (uint32_t q = 1; q < nbest; q++) { const GBest2 gb = g_best[q]; hit->taxid[q] = (uint64_t)gb.lo | (uint64_t)gb.len << 32; }
              nids = nbest; flags |= kHitLocPending;
            }
          }
          hit->n_ids = nids; hit->flags = flags;
          if constexpr (COUNT) oc[kOpcHit]++;
          kind = G_IDLE; bk = GB_NONE;
          continue;
        }
        // (GB_DONE does not occur here: GB_FINISH writes the record itself)
        kind = G_IDLE; bk = GB_NONE;
      }
    } else if (C == (uint32_t)C3_IDLE) {
      // ---- reads for the rows that finished one: one atomic per wavefront ----
      KJ_P(PS_HANDOUT);
#if defined(__HIP_DEVICE_COMPILE__)
      const uint64_t mask = __ballot(true);
      const uint32_t leader = (uint32_t)__builtin_ctzll(mask);
      uint32_t got = 0;
      if ((threadIdx.x & 63u) == leader) got = atomicAdd(wl.counter, popc64(mask));
      const uint32_t item = (uint32_t)__shfl((int)got, (int)leader, 64) + kj_rank_below(mask);
#else
      const uint32_t item = fetch_work(wl.counter);
#endif
      if (item >= n_items) kind = G_EXIT;
      else { r = wl.reads ? wl.reads[item] : item; kind = G_META; }
    } else {
      // ---- a memory access and what follows from it ----
      KJ_P(PS_LOAD);
      const bool is_step = kind == G_STEP, is_vm = kind == G_VMULTI;
      const bool is_kmer = kind == G_KMER || kind == G_PROBE;
      const P posA = is_step ? lo : is_vm ? m_lo : 0;
      const P posB = is_step ? hi : is_vm ? m_lo + m_len : posA;
      if constexpr (COUNT) {
        if (is_kmer) oc[kOpcKmer]++;
        else if (kind == G_STEP) { oc[kOpcStep]++; oc[kOpcStepLines] += ((posA >> 6) != (posB >> 6)) ? 2u : 1u; }
        else if (kind == G_VMULTI) { oc[kOpcVmulti]++; oc[kOpcStepLines] += ((posA >> 6) != (posB >> 6)) ? 2u : 1u; }
        else if (kind == G_META) oc[kOpcMeta]++;
        else if (kind == G_FRAG) oc[kOpcFrag]++;
        else if (kind == G_FILL) { oc[kOpcFill]++; if (fill_pref && fo < nf) oc[kOpcFrag]++; }
        else if (kind == G_POPITEM) oc[kOpcPopItem]++;
        else if (kind == G_MLOAD) oc[kOpcMload]++;
      }
      if (C == (uint32_t)C3_FAST || C == (uint32_t)C3_VMULTI) {
        const uint32_t cc = is_step ? c : 1u;
        const RankBlock64 *pa = blk0 + (posA >> 6), *pb = blk0 + (posB >> 6);
        const u128 a01 = *reinterpret_cast<const u128 *>(&pa->plane[0]);
        const u128 a23 = *reinterpret_cast<const u128 *>(&pa->plane[2]);
        const uint64_t a4 = pa->plane[4];
        const uint32_t ca = pa->cnt[cc - 1];
        const u128 b01 = *reinterpret_cast<const u128 *>(&pb->plane[0]);
        const u128 b23 = *reinterpret_cast<const u128 *>(&pb->plane[2]);
        const uint64_t b4 = pb->plane[4];
        // (G_KMER: the second block is not needed - this load fetches the presence bits of the k-mer line instead)
        const uint32_t *cbp = &pb->cnt[cc - 1];
        if (is_kmer) cbp = reinterpret_cast<const uint32_t *>(ix.kline + (size_t)(kidx >> 6) * kKLineBytes + kKLinePresent);
        const uint32_t cb = *cbp;
        const uint8_t *gaddr = reinterpret_cast<const uint8_t *>(blk0);
        if (is_kmer) gaddr = ix.kline + (size_t)kidx * 2u;
        const u128 gv = *reinterpret_cast<const u128_unaligned *>(gaddr);
        if (is_step) {
          KJ_P(PS_STEP);
          const uint64_t ia = (cc & 1u) ? 0ull : ~0ull, ib = (cc & 2u) ? 0ull : ~0ull, ic = (cc & 4u) ? 0ull : ~0ull,
                         id = (cc & 8u) ? 0ull : ~0ull, ie = (cc & 16u) ? 0ull : ~0ull;
          const uint64_t ma = (a01.x ^ ia) & (a01.y ^ ib) & (a23.x ^ ic) & (a23.y ^ id) & (a4 ^ ie);
          const P ra = (P)(ca + popc64(ma & ((1ull << (posA & 63u)) - 1ull)));
          // UpdateSI(str[i-1]) (bwt.c:160-173)
          const uint64_t mb = (b01.x ^ ia) & (b01.y ^ ib) & (b23.x ^ ic) & (b23.y ^ id) & (b4 ^ ie);
          const P rb = (P)(cb + popc64(mb & ((1ull << (posB & 63u)) - 1ull)));
          if (ra >= rb) bk = GB_END_MATCH;
          else {
            lo = ra; hi = rb; i--; acc += diag(c);
            if (i == 0) bk = GB_END_MATCH;
            else if (kSpanRuleStep && t_nmm == 0 && nm != 0 && hi - lo == (kSpanEq ? sz_q : (P)1) && i >= (int)last_qi) {
              i = last_qi; bk = GB_END_MATCH;
            }
            else if (in_win(i - 1)) c = win[i - 1 - (int)wq];
            else { fill_top = (uint16_t)(i - 1); fill_ret = FR_STEP; fill_pref = false; kind = G_FILL; }
          }
        } else if (is_kmer) {
          KJ_P(PS_KMER);
          const uint64_t e = kline_entry(gv, kidx);
          const uint32_t l16 = (uint32_t)(e >> 32);
          uint32_t hint = 32u;                                 // the BWT letter of a one-row interval (32 = unknown)
          lo = (P)(uint32_t)e;
          if (l16 >= kKLineSingle && l16 != kKLineEscape) { hi = lo + 1u; hint = l16 & 31u; }
          else hi = lo + l16;
          skipj = j >= (int)kk && in_win(j - (int)kk) && ((cb >> ((uint32_t)win[j - (int)kk - (int)wq] - 1u)) & 1u) == 0u;
          if (kind == G_PROBE) {
            if (l16 == 0u) { tail += acc; j = (int)last_qi + (int)kk - 3; }
            skipj = false; kroll = false;
            bk = GB_START_J;
          } else
          if (l16 == kKLineEscape) {
            c = cj;
            lo = (P)ix.C[c]; hi = (P)ix.C[c + 1];
            acc = diag(c);
            i = j;
            if (in_win(i - 1)) { c = win[i - 1 - (int)wq]; kind = G_STEP; }
            else { fill_top = (uint16_t)(i - 1); fill_ret = FR_STEP; fill_pref = false; kind = G_FILL; }
          } else if (lo >= hi) { i = j; bk = GB_END_MATCH; }
          else if (kSpanRule && hi - lo == (kSpanEq ? sz_i : (P)1) && j - (int)kk + 1 >= i) bk = GB_END_MATCH;
          else {
            i = j - (int)kk + 1;
            if (i == 0) bk = GB_END_MATCH;
            else if (in_win(i - 1)) {
              c = win[i - 1 - (int)wq];
              if (hint != 32u && hint != c) bk = GB_END_MATCH; else kind = G_STEP;
            }
            else { fill_top = (uint16_t)(i - 1); fill_ret = FR_STEP; fill_pref = false; kind = G_FILL; }
          }
        } else if (is_vm) {
          KJ_P(PS_VM_RANK);
          // UpdateSI(trans[substitute]) on the interval of the match for all substitutes (ConsumerThread.cpp:366-392): see
          // greedy_lane2
          const uint64_t lowA = (1ull << (posA & 63u)) - 1ull, lowB = (1ull << (posB & 63u)) - 1ull;
          const uint32_t dblk = (uint32_t)((posB >> 6) - (posA >> 6));
          const int32_t thr = (int32_t)best > (int32_t)p.min_score ? (int32_t)best : (int32_t)p.min_score;
          const uint32_t corig = ct.aa_to_idx[vorig];
          auto match_of = [](const u128 &p01, const u128 &p23, uint64_t p4, uint32_t cx) -> uint64_t {
            const uint64_t ia = (cx & 1u) ? 0ull : ~0ull, ib = (cx & 2u) ? 0ull : ~0ull, ic = (cx & 4u) ? 0ull : ~0ull,
                           id = (cx & 8u) ? 0ull : ~0ull, ie = (cx & 16u) ? 0ull : ~0ull;
            return (p01.x ^ ia) & (p01.y ^ ib) & (p23.x ^ ic) & (p23.y ^ id) & (p4 ^ ie);
          };
          auto symbol_of = [](const u128 &p01, const u128 &p23, uint64_t p4, uint32_t t) -> uint32_t {
            return (uint32_t)((p01.x >> t) & 1ull) | (uint32_t)((p01.y >> t) & 1ull) << 1 | (uint32_t)((p23.x >> t) & 1ull) << 2 |
                   (uint32_t)((p23.y >> t) & 1ull) << 3 | (uint32_t)((p4 >> t) & 1ull) << 4;
          };
          uint32_t todo = 0x1ffffeu;                          // letters 1..20
          if (dblk <= 1u) {
            todo = 0;
            uint64_t ma = dblk == 0 ? (lowB & ~lowA) : ~lowA;
            while (ma) {
              const uint32_t cx = symbol_of(a01, a23, a4, (uint32_t)__builtin_ctzll(ma));
              ma &= ~match_of(a01, a23, a4, cx);
              todo |= 1u << cx;
            }
            uint64_t mb = dblk == 0 ? 0ull : lowB;
            while (mb) {
              const uint32_t cx = symbol_of(b01, b23, b4, (uint32_t)__builtin_ctzll(mb));
              mb &= ~match_of(b01, b23, b4, cx);
              todo |= 1u << cx;
            }
            todo &= 0x1ffffeu;
          }
          todo &= ~(1u << corig);
          uint32_t q0 = sp0, q1 = sp1, q2 = sp2, q3 = sp3;
          const uint32_t pz = (uint32_t)m_qi - 1u;
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
          const int need_fq = (int)m_qi - 2 - (kWin - 1) > 0 ? (int)m_qi - 2 - (kWin - 1) : 0;
          const bool win_ok = need_fq == (int)wq;             // the variant resumes at m_qi-2 (if m_qi > 1)
          while (todo) {
            const uint32_t cx = (uint32_t)__builtin_ctz(todo);
            todo &= todo - 1u;
            const int bos = (int)ct.b62_idx[vorig][cx - 1u];
            const uint32_t key = (uint32_t)(int32_t)(vscore + (uint32_t)(int32_t)bos);
            if ((int32_t)key < thr) continue;
            const P ra = (P)(pa->cnt[cx - 1u] + popc64(match_of(a01, a23, a4, cx) & lowA));
            const P rb = (P)(pb->cnt[cx - 1u] + popc64(match_of(b01, b23, b4, cx) & lowB));
            if (ra >= rb) continue;
            KJ_P(PS_VM_PUSH);
            if (vlen >= kG3MaxLen || (uint32_t)m_ql + 1u >= kG3MaxLen) { ovf = true; KJ_OVF(wl, 4); break; }
            const int bss = (int)diag(cx);
            const uint32_t sl = push_slot(key, qseq + ct.subst_rank[vorig][cx - 1u]);
            if (sl == ~0u) break;
            uint32_t e0 = sa0, e1 = sa1;
            if (t_nmm < (uint32_t)kMaxMismatch) {
              const uint32_t bs = (t_nmm & 3u) * 8u, bm = ~(0xffu << bs);
              if (t_nmm < 4u) e0 = (e0 & bm) | cx << bs; else e1 = (e1 & bm) | cx << bs;
            }
            u128 *dst = g_pool + 8 * sl;
            u128 vv;
            vv.x = ra | (uint64_t)rb << 32; vv.y = key | (uint64_t)t_start << 32;
            dst[0] = vv;
            vv.x = ((uint32_t)vlen | ((uint32_t)m_ql + 1u) << 16) | (uint64_t)(uint32_t)(t_diff + bos - bss) << 32;
            vv.y = ((uint32_t)m_psum - (uint32_t)boo + (uint32_t)bss) | (uint64_t)((uint32_t)m_dsum + (uint32_t)bss) << 32; dst[1] = vv;
            vv.x = ((uint32_t)t_nmm + 1u) | (uint64_t)q0 << 32; vv.y = q1 | (uint64_t)q2 << 32; dst[2] = vv;
            vv.x = q3 | (uint64_t)e0 << 32;
            vv.y = e1 | (uint64_t)(win_ok ? (uint32_t)wq + 1u : 0u) << 32; dst[3] = vv;
            if (win_ok) {
              const uint32_t *w32 = reinterpret_cast<const uint32_t *>(win);
              u128 wv;
              wv.x = w32[0] | (uint64_t)w32[1] << 32; wv.y = w32[2] | (uint64_t)w32[3] << 32; dst[4] = wv;
              wv.x = w32[4] | (uint64_t)w32[5] << 32; wv.y = w32[6] | (uint64_t)w32[7] << 32; dst[5] = wv;
              wv.x = w32[8] | (uint64_t)w32[9] << 32; wv.y = w32[10] | (uint64_t)w32[11] << 32; dst[6] = wv;
              wv.x = w32[12] | (uint64_t)w32[13] << 32; wv.y = w32[14] | (uint64_t)w32[15] << 32; dst[7] = wv;
              reinterpret_cast<uint8_t *>(dst + 4)[(int)pz - (int)wq] = (uint8_t)cx;   // pz is in the window (GB_VAR_MATCH)
            }
          }
          qseq += 19;
          bk = GB_VAR_NEXT;
        }
      } else if (C == (uint32_t)C3_DESC) {
        const uint8_t *gaddr = kind == G_META ? reinterpret_cast<const uint8_t *>(b.meta + r) : reinterpret_cast<const uint8_t *>(b.frags + fbase);
        const u128 gv = *reinterpret_cast<const u128 *>(gaddr);
        if (kind == G_META) {
          KJ_P(PS_META);
          pep16 = (uint32_t)(gv.x >> 4);
          fbase = (uint32_t)gv.y;
          const uint32_t nfr = (uint32_t)(gv.y >> 32) & ~kNfragSegPending;
          fo = 0; best = 0; nbest = 0; flags = 0; ovf = false; m_ovf = false;
          for (uint32_t s = 0; s < (uint32_t)kG3PrioWords; s++) prio[s] = 0;
          qn = 0; qlive = 0; qseq = 0; ext_max = 0;
          on_start = on_kl = on_flags = b0lo = b0len = 0;
          nf = (uint16_t)(nfr > 0xffffu ? 0xffffu : nfr);
          // (a peptide area that is not 16-byte aligned, 2^16 fragments and more: not this lane's read)
          if ((gv.x & 15ull) != 0 || nfr > 0xfffeu) { ovf = true; KJ_OVF(wl, 2); bk = GB_FINISH; }
          else if (nfr == 0) bk = GB_FINISH; else kind = G_FRAG;
        } else {
          KJ_P(PS_FRAG);
          const uint32_t fl = (uint32_t)(gv.x >> 32), fkey = (uint32_t)gv.y;
          on_start = (uint32_t)gv.x; on_kl = (fl > 0xffffu ? 0xffffu : fl) | (fkey > 0xffffu ? 0xffffu : fkey) << 16; on_flags = (uint32_t)(gv.y >> 32);
          bk = GB_POP;
        }
      } else if (C == (uint32_t)C3_FILL) {
        KJ_P(PS_FILL);
        bool fill = kind == G_FILL;
        int fq = (int)fill_top - (kWin - 1);
        if (fq < 0) fq = 0;
        u128 gv{0, 0};
        if (fill && fill_pref && fo < nf) gv = *reinterpret_cast<const u128 *>(b.frags + fbase + fo);
        u128 f0, f1, f2, f3;
        {
          const uint8_t *src = fill ? pepr + t_start + fq : reinterpret_cast<const uint8_t *>(g_pool + 8 * (uint32_t)pslot);
          const u128_unaligned *s16 = reinterpret_cast<const u128_unaligned *>(src);
          f0 = s16[0]; f1 = s16[1]; f2 = s16[2]; f3 = s16[3];
        }
        int newq = fq;
        if (!fill) {
          const u128 xa0 = f0, xa1 = f1, xa2 = f2, xa3 = f3;
          lo = (P)xa0.x; hi = (P)(xa0.x >> 32); t_start = (uint32_t)(xa0.y >> 32);
          t_len = (uint16_t)((uint32_t)xa1.x & 0xffffu); t_matchlen = (uint16_t)(((uint32_t)xa1.x >> 16) & 0xffffu);
          t_diff = (int32_t)(uint32_t)(xa1.x >> 32);
          t_tot = (uint32_t)xa1.y; t_msum = (uint32_t)(xa1.y >> 32);
          t_nmm = (uint8_t)(uint32_t)xa2.x;
          sp0 = (uint32_t)(xa2.x >> 32); sp1 = (uint32_t)xa2.y; sp2 = (uint32_t)(xa2.y >> 32); sp3 = (uint32_t)xa3.x;
          sa0 = (uint32_t)(xa3.x >> 32); sa1 = (uint32_t)xa3.y;
          const uint32_t wtag = (uint32_t)(xa3.y >> 32);
          nm = 0; kroll = false; skipj = false;
          j = flen - 1;
          if (t_nmm == 0) {
            // a SEG piece: maxMatches like an original
            tail = 0;
            i = flen;
            fill_top = (uint16_t)j; fill_ret = FR_START_J; fill_pref = false; kind = G_FILL;
          } else {
            // maxMatches_withStart, bwt.c:298-336
            i = j - (int)t_matchlen + 1;
            acc = t_msum;
            if (i <= 0) bk = GB_END_MATCH;
            else if (wtag != 0) {                           // the item carries its window
              fill = true; fill_ret = FR_STEP; fill_pref = false;
              const u128 *w16 = g_pool + 8 * (uint32_t)pslot + 4;
              f0 = w16[0]; f1 = w16[1]; f2 = w16[2]; f3 = w16[3]; newq = (int)wtag - 1;
            } else { fill_top = (uint16_t)(i - 1); fill_ret = FR_STEP; fill_pref = false; kind = G_FILL; }
          }
        }
        if (fill) {
          wq = (uint16_t)newq;
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
              if (pz >= (int)wq && pz < (int)wq + kWin && pz < (int)t_len) win[pz - (int)wq] = (uint8_t)(aw >> ((x & 3u) * 8u));
            }
            if (fill_pref && fo < nf) {
              const uint32_t fl = (uint32_t)(gv.x >> 32), fkey = (uint32_t)gv.y;
              on_start = (uint32_t)gv.x; on_kl = (fl > 0xffffu ? 0xffffu : fl) | (fkey > 0xffffu ? 0xffffu : fkey) << 16; on_flags = (uint32_t)(gv.y >> 32);
            }
          }
          if (fill_ret == FR_STEP) { c = win[i - 1 - (int)wq]; kind = G_STEP; }
          else if (fill_ret == FR_START_J) bk = GB_START_J;
          else bk = GB_VAR_MATCH;
        }
      } else if (C == (uint32_t)C3_MLOAD) {
        KJ_P(PS_MLOAD);
        const u128 gv = *reinterpret_cast<const u128 *>(g_matches + mx);
        m_lo = (uint32_t)gv.x; m_len = (uint32_t)(gv.x >> 32);
        m_qi = (uint16_t)((uint32_t)gv.y & 0xffffu); m_ql = (uint16_t)(((uint32_t)gv.y >> 16) & 0xffffu);
        m_dsum = (uint16_t)((uint32_t)(gv.y >> 32) & 0xffffu); m_psum = (uint16_t)((uint32_t)(gv.y >> 48) & 0xffffu);
        bk = ml_for == 0 ? GB_VAR_MATCH : GB_EVAL_MATCH;
      }

      // ---- the fast bookkeeping (END_MATCH, START_J); everything else is another class's ----
      KJ_P(PS_TAIL);
      while (bk != GB_NONE) {
        if (bk == GB_END_MATCH) {
          KJ_P(PS_END_MATCH);
          const int l = j - i + 1;
          if (t_nmm == 0) {
            if constexpr (kSpanEq) sz_i = hi > lo ? (P)(hi - lo) : (P)1;
            bool recorded = false;
            if (l >= (int)p.seed_length && (nm == 0 || i < (int)last_qi)) {        // bwt.c:276-278
              recorded = true;
              if (nm < (uint32_t)kGMaxMAll) {
                m_lo = lo; m_len = (uint32_t)(hi - lo); m_qi = (uint16_t)i; m_ql = (uint16_t)l; m_dsum = (uint16_t)acc; m_psum = (uint16_t)(t_tot - tail);
                GMatch2 mm; mm.lo = m_lo; mm.len = m_len; mm.qiql = (uint32_t)m_qi | (uint32_t)m_ql << 16; mm.dp = (uint32_t)m_dsum | (uint32_t)m_psum << 16;
                g_matches[nm] = mm;
                if (nm < (uint32_t)kGMaxM) mq[nm] = (uint16_t)l; else g_mq_ext[nm - kGMaxM] = (uint16_t)l;
                if constexpr (COUNT) oc[kOpcMatchWr]++;
              } else { if (!m_ovf) KJ_OVF(wl, 5); m_ovf = true; }
              nm++;
              last_qi = (uint16_t)i;
              if constexpr (kSpanEq) sz_q = (P)(hi - lo);
            }
            if (i <= 1) bk = GB_AFTER_SEARCH;                                 // bwt.c:292
            else {
              tail += diag(cj); j--; bk = GB_START_J;
              probe_now = kGreedyProbe && recorded && kk && l > (int)kk && in_win(i - 1) && in_win(i + (int)kk - 2);
            }
          } else {
            // :443-449: after the last allowed mismatch the match must reach min_fragment_length
            const int Lreq = (t_nmm == p.mismatches) ? (int)p.m : (int)t_matchlen;
            if (l >= Lreq) {
              m_lo = lo; m_len = (uint32_t)(hi - lo); m_qi = (uint16_t)i; m_ql = (uint16_t)l; m_dsum = (uint16_t)acc; m_psum = (uint16_t)t_tot;
              nm = 1;
            }
            bk = GB_AFTER_SEARCH;
          }
        }
        if (bk == GB_START_J) {
          KJ_P(PS_START_J);
          if (probe_now) skipj = false;
          if (skipj && j >= (int)p.seed_length - 1 && in_win(j) && in_win(j - (int)kk + 1)) {
            skipj = false;
            if (j <= 1) bk = GB_AFTER_SEARCH;
            else {
              const uint32_t cn = win[j - (int)kk + 1 - (int)wq], c1 = win[j - (int)wq];
              kidx = (((kidx >> 6) - (c1 - 1u) * kpow) * 20u + (cn - 1u)) << 6;
              kacc = kacc - diag(cj) + diag(cn);
              cj = (uint8_t)c1;
              tail += diag(c1); j--;
            }
          }
          if (bk != GB_START_J) {}
          else if (j < (int)p.seed_length - 1) { skipj = false; bk = GB_AFTER_SEARCH; }
          else if (!probe_now && (!in_win(j) || (kk && j >= (int)kk - 1 && !in_win(j - (int)kk + 1)))) {
            fill_top = (uint16_t)j; fill_ret = FR_START_J; fill_pref = false; kind = G_FILL; bk = GB_NONE;     // (skipj, if set, waits)
          } else if (kk && j >= (int)kk - 1) {
            uint32_t kcode;
            const int ej = probe_now ? i + (int)kk - 2 : j;   // end position of the k-mer looked up
            if (kroll && !probe_now) {
              const uint32_t cn = win[j - (int)kk + 1 - (int)wq], c1 = win[j - (int)wq];
              kcode = ((kidx >> 6) - (c1 - 1u) * kpow) * 20u + (cn - 1u);
              kacc = kacc - diag(cj) + diag(cn);
            } else {
              kcode = 0; kacc = diag(win[ej - (int)wq]);
              for (uint32_t q = 1; q < kk; q++) {
                const uint32_t cq = win[ej - (int)q - (int)wq];
                kcode = kline_code(kcode, cq);
                kacc += diag(cq);
              }
            }
            if (probe_now) {
              const uint32_t ce = win[ej - (int)wq];
              acc = acc - diag(cj) - (kacc - diag(ce) - diag(win[i - 1 - (int)wq]));
              kroll = false;
              kidx = kline_ref(kcode, ce);
              kind = G_PROBE; bk = GB_NONE;
            } else {
              cj = win[j - (int)wq]; acc = kacc; kroll = true;
              kidx = kline_ref(kcode, cj);
              kind = G_KMER; bk = GB_NONE;
            }
            probe_now = false;
          } else {
            c = cj = win[j - (int)wq]; kroll = false;
            lo = (P)ix.C[c]; hi = (P)ix.C[c + 1];                              // InitialSI, bwt.c:146-152
            acc = diag(c);
            i = j;
            if (i == 0) { bk = GB_END_MATCH; continue; }
            else if (in_win(i - 1)) { c = win[i - 1 - (int)wq]; kind = G_STEP; bk = GB_NONE; }
            else { fill_top = (uint16_t)(i - 1); fill_ret = FR_STEP; fill_pref = false; kind = G_FILL; bk = GB_NONE; }
          }
        }
        if (bk > GB_START_J) { bk_pend = (uint8_t)bk; kind = G_WAIT; bk = GB_NONE; }
      }
    }

    // ---- the row goes back under its new class ----
    bits_ = (uint16_t)((kroll ? 1u : 0u) | (skipj ? 2u : 0u) | (ovf ? 4u : 0u) | (m_ovf ? 8u : 0u) | (fill_pref ? 16u : 0u) | (ev_done ? 32u : 0u) |
                       (ml_for & 1u) << 6 | ((uint32_t)fill_ret & 3u) << 7 | ((uint32_t)vi_phase & 3u) << 9);
    const uint32_t nc = g3_class_of((int)kind, (int)bk_pend);
#if defined(__HIP_DEVICE_COMPILE__)
    KJ_P(PS_SA);                                  // (profile: giving the rows back)
    g3_release(gx, v, nc);
#else
    gx.cls[0] = nc;
#endif
#undef flen
  }
  if constexpr (COUNT) opc_flush(opc_of(wl), oc);
#if defined(KJ_PROF) && defined(__HIP_DEVICE_COMPILE__)
  KJ_P(PS_HEAD);
  if ((threadIdx.x & 63u) == 0) {
    unsigned long long *dst = reinterpret_cast<unsigned long long *>(reinterpret_cast<uint8_t *>(wl.counter) + 1024);
    for (int x = 0; x < 3 * PS_N; x++) atomicAdd(dst + x, gx.prof[2 + x]);
  }
#endif
}

}  // namespace kj
