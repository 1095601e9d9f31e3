// host_tables.h — constant tables of the classification path, built on the host.
#pragma once
#include <string>
#include <vector>

#include "kj_core.h"

namespace kj {

// BLOSUM62 / substitution order / genetic code / nucleotide codes (ConsumerThread.cpp:6-187)
// and the mapping between the reference's aa2int order and the index alphabet.
// Returns 0 or a negative status (alphabet lacking one of the 20 amino acids).
int build_const_tables(const uint8_t trans[128], ConstTables &t, std::string &msg);

// SEG tables (blast_seg.c): ln(n!) table and the integer entropy classification, which is
// verified here against the reference's floating-point expression for all 77 partitions
// of the window.  lnfact_host receives the table the device copy is made from.
int build_seg_tables(std::vector<double> &lnfact_host, SegTables &st, std::string &msg);

// tables of the fast stage 1 (kj_core.h: Stage1Tables) from the two above
void build_stage1_tables(const ConstTables &ct, const SegTables &st, Stage1Tables &t);

}  // namespace kj
