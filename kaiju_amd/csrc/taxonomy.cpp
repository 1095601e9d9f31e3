// taxonomy.cpp — host side of the seam: nodes.dmp, LCA, E-value gate, C/U decision.
//
// Mirrors parseNodesDmp (util.cpp:79-99), lca_from_ids (util.cpp:194-263), the E-value gate of
// classify_greedyblosum (ConsumerThread.cpp:500-513) and the output decision of doWork
// (ConsumerThread.cpp:724-739).  Unlike the reference, node depths are computed once at load
// so that kaiju_taxonomy_lca is read-only and can be called from many threads.
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/kaiju_gpu.h"

#include "taxonomy.h"

extern "C" int kaiju_taxonomy_load(const char *path, kaiju_taxonomy **out) {
  if (!path || !out) return KAIJU_GPU_ERR_ARG;
  FILE *fp = fopen(path, "r");
  if (!fp) return KAIJU_GPU_ERR_IO;
  kaiju_taxonomy *t = new kaiju_taxonomy();
  t->nodes.reserve(2000000);
  char *line = nullptr;
  size_t cap = 0;
  ssize_t n;
  while ((n = getline(&line, &cap, fp)) > 0) {
    // first two unsigned integers of the line: node id, parent id (util.cpp:86-91)
    const char *s = line;
    if (!isdigit((unsigned char)*s)) continue;
    char *e = nullptr;
    const uint64_t node = strtoull(s, &e, 10);
    s = e;
    while (*s && !isdigit((unsigned char)*s)) s++;
    if (!*s) continue;
    const uint64_t parent = strtoull(s, nullptr, 10);
    t->nodes.emplace(node, kaiju_taxonomy::Node{parent, 0});   // emplace keeps the first entry
  }
  free(line);
  fclose(fp);
  // depth as computed in lca_from_ids (util.cpp:218-224): 1 + number of parent steps until a
  // node is its own parent or unknown
  std::vector<uint64_t> path_ids;
  for (auto &kv : t->nodes) {
    if (kv.second.depth) continue;
    path_ids.clear();
    uint64_t id = kv.first;
    uint32_t base = 0;
    for (;;) {
      auto it = t->nodes.find(id);
      if (it == t->nodes.end()) { base = 1; break; }          // unknown parent still costs one step (:220-223)
      if (it->second.depth) { base = it->second.depth; break; }
      path_ids.push_back(id);
      if (it->second.parent == id) { base = 0; break; }
      if (path_ids.size() > 100000) { base = 0; break; }      // cycle guard
      id = it->second.parent;
    }
    // nodes on the path: the last pushed has depth base+1 (1 for a root, 2 below an unknown parent)
    uint32_t d = base;
    for (size_t i = path_ids.size(); i-- > 0;) {
      d += 1;
      t->nodes[path_ids[i]].depth = d;
    }
  }
  *out = t;
  return KAIJU_GPU_OK;
}

extern "C" void kaiju_taxonomy_free(kaiju_taxonomy *t) { delete t; }

extern "C" uint64_t kaiju_taxonomy_lca(kaiju_taxonomy *t, const uint64_t *ids, uint32_t n) {
  if (!t || !ids || n == 0) return 0;
  if (n == 1) return ids[0];                                    // util.cpp:197-199
  uint64_t leafs[64];
  uint32_t depth[64];
  uint32_t m = 0, shallowest = 100000;
  for (uint32_t i = 0; i < n && m < 64; i++) {
    auto it = t->nodes.find(ids[i]);
    if (it == t->nodes.end()) continue;                         // not in the tree: dropped (:206-210)
    leafs[m] = ids[i]; depth[m] = it->second.depth;
    if (depth[m] < shallowest) shallowest = depth[m];
    m++;
  }
  if (m == 0) return 0;
  auto parent = [&](uint64_t id) {
    auto it = t->nodes.find(id);
    return it == t->nodes.end() ? id : it->second.parent;
  };
  for (uint32_t i = 0; i < m; i++)
    for (uint32_t d = depth[i]; d > shallowest; d--) leafs[i] = parent(leafs[i]);
  for (uint32_t guard = 0; guard < 200000; guard++) {           // :245-262
    const uint64_t first = leafs[0];
    bool same = true;
    for (uint32_t i = 0; i < m; i++) {
      if (leafs[i] != first) same = false;
      leafs[i] = parent(leafs[i]);
    }
    if (same) return first;
  }
  return 0;
}

extern "C" int kaiju_finalize_hits(kaiju_taxonomy *t, const kaiju_gpu_params *p, double db_length,
                                   const kaiju_gpu_hit *hits, const uint64_t *off, uint32_t n_reads,
                                   int paired, kaiju_result *out) {
  if (!t || !p || !hits || !off || !out) return KAIJU_GPU_ERR_ARG;
  // ConsumerThread.hpp:41-44
  const double LN_2 = 0.6931471805, LAMBDA = 0.3176, LN_K = -2.009915479;
  for (uint32_t r = 0; r < n_reads; r++) {
    const kaiju_gpu_hit &h = hits[r];
    kaiju_result &o = out[r];
    o.taxon = 0; o.best = h.best; o.classified = 0; o.pad[0] = o.pad[1] = o.pad[2] = 0;
    if (h.n_ids == 0 || h.best == 0) continue;
    if (p->mode == 1 && p->use_evalue) {
      const uint64_t len1 = off[2 * (uint64_t)r + 1] - off[2 * (uint64_t)r];
      const uint64_t len2 = off[2 * (uint64_t)r + 2] - off[2 * (uint64_t)r + 1];
      double query_len = static_cast<double>(len1) / 3.0;                    // :698
      if (paired) query_len += static_cast<double>(len2) / 3.0;             // :704
      if (p->input_is_protein) query_len = static_cast<double>(len1);       // :660
      const double bitscore = (LAMBDA * h.best - LN_K) / LN_2;
      const double Evalue = db_length * query_len * pow(2, -1 * bitscore);
      if (Evalue > p->min_evalue) continue;
    }
    const uint64_t lca = h.n_ids == 1 ? h.taxid[0] : kaiju_taxonomy_lca(t, h.taxid, h.n_ids);
    if (lca > 0) { o.taxon = lca; o.classified = 1; }
  }
  return KAIJU_GPU_OK;
}


// the tree as the open-addressing table of kj_core.h:DevTaxonomy (host arrays; capi.hip uploads them)
void kj_taxonomy_table(const kaiju_taxonomy *t, std::vector<uint64_t> &key, std::vector<uint64_t> &parent_id,
                       std::vector<uint32_t> &parent_slot, std::vector<uint32_t> &depth) {
  size_t cap = 16;
  while (cap < 2 * t->nodes.size() + 2) cap <<= 1;
  key.assign(cap, ~0ull); parent_id.assign(cap, 0); parent_slot.assign(cap, ~0u); depth.assign(cap, 0);
  const uint32_t mask = (uint32_t)(cap - 1);
  auto hash = [](uint64_t id) {
    id ^= id >> 33; id *= 0xff51afd7ed558ccdULL; id ^= id >> 33; id *= 0xc4ceb9fe1a85ec53ULL; id ^= id >> 33;
    return (uint32_t)id;
  };
  auto find = [&](uint64_t id) -> uint32_t {
    uint32_t s = hash(id) & mask;
    while (key[s] != ~0ull && key[s] != id) s = (s + 1) & mask;
    return s;
  };
  for (const auto &kv : t->nodes) {
    const uint32_t s = find(kv.first);
    key[s] = kv.first; parent_id[s] = kv.second.parent; depth[s] = kv.second.depth;
  }
  for (size_t s = 0; s < cap; s++) {
    if (key[s] == ~0ull) continue;
    const uint32_t ps = find(parent_id[s]);
    parent_slot[s] = key[ps] == parent_id[s] ? ps : ~0u;
  }
}

extern "C" int kaiju_finalize_compact(const kaiju_gpu_params *p, double db_length, const kaiju_gpu_compact *recs,
                                      const uint64_t *off, uint32_t n_reads, int paired, kaiju_result *out) {
  if (!p || !recs || !off || !out) return KAIJU_GPU_ERR_ARG;
  const double LN_2 = 0.6931471805, LAMBDA = 0.3176, LN_K = -2.009915479;   // ConsumerThread.hpp:41-44
  for (uint32_t r = 0; r < n_reads; r++) {
    const kaiju_gpu_compact &h = recs[r];
    kaiju_result &o = out[r];
    o.taxon = 0; o.best = h.best; o.classified = 0; o.pad[0] = o.pad[1] = o.pad[2] = 0;
    if ((h.info & 255u) == 0 || h.best == 0) continue;
    if (p->mode == 1 && p->use_evalue) {
      const uint64_t len1 = off[2 * (uint64_t)r + 1] - off[2 * (uint64_t)r];
      const uint64_t len2 = off[2 * (uint64_t)r + 2] - off[2 * (uint64_t)r + 1];
      double query_len = static_cast<double>(len1) / 3.0;                    // :698
      if (paired) query_len += static_cast<double>(len2) / 3.0;             // :704
      if (p->input_is_protein) query_len = static_cast<double>(len1);       // :660
      const double bitscore = (LAMBDA * h.best - LN_K) / LN_2;
      const double Evalue = db_length * query_len * pow(2, -1 * bitscore);
      if (Evalue > p->min_evalue) continue;
    }
    if (h.lca > 0) { o.taxon = h.lca; o.classified = 1; }
  }
  return KAIJU_GPU_OK;
}
