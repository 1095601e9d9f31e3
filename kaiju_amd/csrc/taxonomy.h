// taxonomy.h — the host-side nodes.dmp tree (taxonomy.cpp) as seen by capi.hip
#pragma once
#include <stdint.h>
#include <unordered_map>
#include <vector>

struct kaiju_taxonomy {
  struct Node { uint64_t parent; uint32_t depth; };
  std::unordered_map<uint64_t, Node> nodes;
};

// the tree as the open-addressing table of kj_core.h:DevTaxonomy
void kj_taxonomy_table(const kaiju_taxonomy *t, std::vector<uint64_t> &key, std::vector<uint64_t> &parent_id,
                       std::vector<uint32_t> &parent_slot, std::vector<uint32_t> &depth);
