// exact_pass.h — the exact pass (kj_core.h: BigSeg) as capi.hip launches it; kernels in exact_pass.hip.
//
// Its kernels live in a translation unit of their own on purpose: a second use of stage 1 (build_fragments) in the
// unit of the hot kernels changes how the compiler inlines and schedules k_fragments (seen in the device assembly,
// tests/tools/isa_dump.sh); apart, the kernels of the main pass stay exactly what was measured.
#pragma once
#include <hip/hip_runtime.h>

#include "kj_core.h"

struct ExactPassLaunch {
  kj::DevIndex ix;
  const kj::ConstTables *d_ct;
  kj::SegTables st;
  kj::Params p;
  kj::Batch b;
  kj::SegQueue sq;            // queue and records of the main SEG pass
  uint32_t *cnt;              // counters of the batch: [3] error flags, [5] listed reads, [6] queue of the exact pass,
                              // [7] its work counter, [20] pairs handed out of the pool
  uint32_t *bitmap, *list;    // one bit per read (zeroed); the listed reads
  uint32_t list_cap;
  kj::SegQueue sq2;           // queue of the exact pass (recs unused)
  kj::BigSeg big;
  int32_t *work; uint8_t *cls;    // SEG scratch: per block 4 * cap_ints ints and cls_bytes bytes
  uint32_t seg_blocks, cap_ints, cls_bytes;
  int n_cu;
  // search: scratch of the retry pass (the exact pass runs behind it on the same stream)
  int blocks_search;
  kj::SIEntry *si; uint32_t si_cap;                                                  // MEM
  kj::GItem *g_pool; uint16_t *g_ord; kj::GMatch *g_matches; kj::GBest *g_best; kj::GBestV *g_bestv;   // Greedy
  uint32_t g_pool_cap, g_match_cap;
  kj::VerboseOut vb;
  hipStream_t stream;
};

// collect the reads -> stage 1 -> SEG with lists of any length -> (MEM) split -> search; asynchronous on a.stream
hipError_t kj_launch_exact_pass(const ExactPassLaunch &a);
