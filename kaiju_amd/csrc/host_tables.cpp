// host_tables.cpp — see host_tables.h
#include "host_tables.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/kaiju_gpu.h"

namespace kj {

namespace {
// the reference's aa2int order (ConsumerThread.cpp:40-60)
const char kAaOrder[21] = "ARNDCQEGHILKMFPSTWYV";
// BLOSUM62, rows/columns in kAaOrder (diagonal: ConsumerThread.cpp:61-80, rest :83-102)
const int8_t kB62[20][20] = {
  { 4,-1,-2,-2, 0,-1,-1, 0,-2,-1,-1,-1,-1,-2,-1, 1, 0,-3,-2, 0},
  {-1, 5, 0,-2,-3, 1, 0,-2, 0,-3,-2, 2,-1,-3,-2,-1,-1,-3,-2,-3},
  {-2, 0, 6, 1,-3, 0, 0, 0, 1,-3,-3, 0,-2,-3,-2, 1, 0,-4,-2,-3},
  {-2,-2, 1, 6,-3, 0, 2,-1,-1,-3,-4,-1,-3,-3,-1, 0,-1,-4,-3,-3},
  { 0,-3,-3,-3, 9,-3,-4,-3,-3,-1,-1,-3,-1,-2,-3,-1,-1,-2,-2,-1},
  {-1, 1, 0, 0,-3, 5, 2,-2, 0,-3,-2, 1, 0,-3,-1, 0,-1,-2,-1,-2},
  {-1, 0, 0, 2,-4, 2, 5,-2, 0,-3,-3, 1,-2,-3,-1, 0,-1,-3,-2,-2},
  { 0,-2, 0,-1,-3,-2,-2, 6,-2,-4,-4,-2,-3,-3,-2, 0,-2,-2,-3,-3},
  {-2, 0, 1,-1,-3, 0, 0,-2, 8,-3,-3,-1,-2,-1,-2,-1,-2,-2, 2,-3},
  {-1,-3,-3,-3,-1,-3,-3,-4,-3, 4, 2,-3, 1, 0,-3,-2,-1,-3,-1, 3},
  {-1,-2,-3,-4,-1,-2,-3,-4,-3, 2, 4,-2, 2, 0,-3,-2,-1,-2,-1, 1},
  {-1, 2, 0,-1,-3, 1, 1,-2,-1,-3,-2, 5,-1,-3,-1, 0,-1,-3,-2,-2},
  {-1,-1,-2,-3,-1, 0,-2,-3,-2, 1, 2,-1, 5, 0,-2,-1,-1,-1,-1, 1},
  {-2,-3,-3,-3,-2,-3,-3,-3,-1, 0, 0,-3, 0, 6,-4,-2,-2, 1, 3,-1},
  {-1,-2,-2,-1,-3,-1,-1,-2,-2,-3,-3,-1,-2,-4, 7,-1,-1,-4,-3,-2},
  { 1,-1, 1, 0,-1, 0, 0, 0,-1,-2,-2, 0,-1,-2,-1, 4, 1,-3,-2,-2},
  { 0,-1, 0,-1,-1,-1,-1,-2,-2,-1,-1,-1,-1,-2,-1, 1, 5,-2,-2, 0},
  {-3,-3,-4,-4,-2,-2,-3,-2,-2,-3,-2,-3,-1, 1,-4,-3,-2,11, 2,-3},
  {-2,-2,-2,-3,-2,-1,-2,-3, 2,-1,-1,-2,-1, 3,-3,-2,-2, 2, 7,-1},
  { 0,-3,-3,-3,-1,-2,-2,-3,-3, 3, 1,-2, 1,-1,-2,-2, 0,-3,-1, 4}};
// standard genetic code, index = n0*16 + n1*4 + n2 with A,C,G,T = 0..3 (codon2aa, :111-177)
const char kCodonAa[65] = "KNKNTTTTRSRSIIMIQHQHPPPPRRRRLLLLEDEDAAAAGGGGVVVV*Y*YSSSS*CWCLFLF";
}  // namespace

int build_const_tables(const uint8_t trans[128], ConstTables &t, std::string &msg) {
  memset(&t, 0, sizeof t);
  memcpy(t.b62, kB62, sizeof kB62);
  // blosum_subst (:10-30): the other 19 residues by descending score; equal scores in
  // descending aa2int order
  for (int a = 0; a < 20; a++) {
    int n = 0;
    for (int s = 11; s >= -4; s--)
      for (int b = 19; b >= 0; b--)
        if (b != a && kB62[a][b] == s) t.subst[a][n++] = (uint8_t)b;
    if (n != 19) { msg = "internal: substitution table"; return KAIJU_GPU_ERR_ARG; }
    // (the bounds of kj_chain_hopeless, kj_core.h: no diagonal entry above kMaxDiagScore, none off the diagonal above kMaxSubstScore)
    for (int b = 0; b < 20; b++)
      if (kB62[a][b] > (a == b ? kMaxDiagScore : kMaxSubstScore)) { msg = "internal: BLOSUM62 bounds of the chain test"; return KAIJU_GPU_ERR_ARG; }
  }
  uint8_t aa2int[128];
  memset(aa2int, 255, sizeof aa2int);
  for (int a = 0; a < 20; a++) aa2int[(int)kAaOrder[a]] = (uint8_t)a;
  for (int i = 0; i < 64; i++) t.codon_aa[i] = kCodonAa[i] == '*' ? 255 : aa2int[(int)kCodonAa[i]];
  memset(t.nuc, 255, sizeof t.nuc);
  t.nuc['A'] = t.nuc['a'] = 0; t.nuc['C'] = t.nuc['c'] = 1; t.nuc['G'] = t.nuc['g'] = 2;
  t.nuc['T'] = t.nuc['t'] = 3; t.nuc['U'] = t.nuc['u'] = 3;
  memset(t.idx_to_aa, 0, sizeof t.idx_to_aa);
  bool seen[32] = {false};
  for (int a = 0; a < 20; a++) {
    const uint8_t c = trans[(int)kAaOrder[a]];
    if (c < 1 || c > 20 || seen[c]) {
      msg = std::string("index alphabet does not hold the 20 amino acids as distinct letters (") + kAaOrder[a] + ")";
      return KAIJU_GPU_ERR_UNSUPPORTED;
    }
    seen[c] = true;
    t.aa_to_idx[a] = c;
    t.idx_to_aa[c] = (uint8_t)a;
    t.diag_idx[c] = kB62[a][a];
  }
  for (int i = 0; i < 64; i++) t.codon_idx[i] = t.codon_aa[i] == 255 ? 0 : t.aa_to_idx[t.codon_aa[i]];
  for (int a = 0; a < 20; a++) {
    for (int c = 1; c <= 20; c++) t.b62_idx[a][c - 1] = kB62[a][t.idx_to_aa[c]];
    memset(t.subst_rank[a], 255, 20);
    for (int k = 0; k < 19; k++) {
      t.subst_rank[a][t.aa_to_idx[t.subst[a][k]] - 1] = (uint8_t)k;
      // the Greedy kernel relies on the rows being sorted (probing stops at the first too-low score)
      if (k > 0 && kB62[a][t.subst[a][k]] > kB62[a][t.subst[a][k - 1]]) { msg = "internal: substitution order"; return KAIJU_GPU_ERR_ARG; }
    }
  }
  return 0;
}

namespace {
// the reference's s_Entropy (blast_seg.c:1596-1626) on a descending state vector
double ref_entropy(const int *sv, int n) {
  const double ln2 = 0.693147180559945309417232121458176568;   // NCBIMATH_LN2
  int total = 0;
  for (int i = 0; i < n; i++) total += sv[i];
  if (total == 0) return 0.;
  double ent = 0.0;
  for (int i = 0; i < n; i++) ent += ((double)sv[i]) * log(((double)sv[i]) / (double)total) / ln2;
  return fabs(ent / (double)total);
}
bool check_partitions(int remaining, int maxpart, int *sv, int n, const SegTables &st, std::string &msg) {
  if (remaining == 0) {
    if (n > 20) return true;                      // more than 20 distinct letters cannot occur
    const double H = ref_entropy(sv, n);
    int64_t score = 0;
    for (int i = 0; i < n; i++) score += st.ent_g[sv[i]];
    const bool lo_ref = H <= 2.2, hi_ref = H > 2.5;   // kSegLocut / kSegHicut, blast_seg.c:48-50
    const bool lo_int = score <= st.ent_locut, hi_int = score > st.ent_hicut;
    if (lo_ref != lo_int || hi_ref != hi_int) { msg = "integer SEG entropy classification disagrees with libm"; return false; }
    int32_t score32 = 0;
    for (int i = 0; i < n; i++) score32 += st.ent_g32[sv[i]];
    if (lo_ref != (score32 <= st.ent_locut32)) { msg = "32-bit SEG trigger classification disagrees with libm"; return false; }
    return true;
  }
  for (int p = remaining < maxpart ? remaining : maxpart; p >= 1; p--) {
    sv[n] = p;
    if (!check_partitions(remaining - p, p, sv, n + 1, st, msg)) return false;
  }
  return true;
}
}  // namespace

int build_seg_tables(std::vector<double> &lnfact_host, SegTables &st, std::string &msg) {
  // blast_seg.c:52-1308 tabulates ln(n!) for n = 0..10000 printed with six decimals
  const int N = 10001;
  lnfact_host.resize(N);
  char buf[64];
  for (int n = 0; n < N; n++) {
    snprintf(buf, sizeof buf, "%.6f", lgamma((double)n + 1.0));
    lnfact_host[n] = strtod(buf, nullptr);
  }
  if (lnfact_host[2] != 0.693147 || lnfact_host[20] != 42.335616 || lnfact_host[50] != 148.477767) {
    msg = "ln(n!) table self-check failed"; return KAIJU_GPU_ERR_ARG;
  }
  st.lnfact = lnfact_host.data();
  st.lnfact_n = (uint32_t)N;
  // fixed-point entropy: H = sum_letters (c/12) log2(12/c); scale 2^40
  const double scale = 1099511627776.0;
  st.ent_g[0] = 0;
  for (int c = 1; c <= 12; c++) st.ent_g[c] = (int64_t)llround(scale * ((double)c / 12.0) * log2(12.0 / (double)c));
  st.ent_locut = (int64_t)floor(scale * 2.2);
  st.ent_hicut = (int64_t)floor(scale * 2.5);
  const double scale32 = 67108864.0;          // 2^26: a window scores below 3.6 * 2^26 < 2^31
  memset(st.ent_g32, 0, sizeof st.ent_g32);
  for (int c = 1; c <= 12; c++) st.ent_g32[c] = (int32_t)llround(scale32 * ((double)c / 12.0) * log2(12.0 / (double)c));
  st.ent_locut32 = (int32_t)floor(scale32 * 2.2);
  int sv[16];
  if (!check_partitions(12, 12, sv, 0, st, msg)) return KAIJU_GPU_ERR_ARG;
  return 0;
}

void build_stage1_tables(const ConstTables &ct, const SegTables &st, Stage1Tables &t) {
  memset(&t, 0, sizeof t);
  for (int c = 0; c < 256; c++) t.nuc3[c] = ct.nuc[c] <= 3 ? ct.nuc[c] : 4;
  for (int idx = 0; idx < 512; idx++) {
    const int n0 = idx >> 6, n1 = (idx >> 3) & 7, n2 = idx & 7;     // n0: the first nucleotide of the codon
    if (n0 > 3 || n1 > 3 || n2 > 3) continue;                       // codon_to_int: a base that is not ACGTU -> stop
    t.tf[idx] = ct.codon_idx[n0 * 16 + n1 * 4 + n2];
    t.tr[idx] = ct.codon_idx[(63 - (n2 * 16 + n1 * 4 + n0)) & 63];  // revcomp_codon_to_int, ConsumerThread.cpp:873-875
  }
  // a window with a stop in it must never reach the trigger entropy: every stop adds more than the threshold
  const int32_t big = st.ent_locut32 + 1;
  for (int c = 0; c < 12; c++) t.dtab[c] = st.ent_g32[c + 1] - st.ent_g32[c];
  for (int c = 13; c < 26; c++) t.dtab[c] = big;
  for (int c = 0; c < 32; c++) t.diag[c] = c >= 1 && c <= 20 ? (uint8_t)ct.diag_idx[c] : 0;
  t.locut32 = st.ent_locut32;
}

}  // namespace kj
