// kernel_emu.cpp — TEST INFRASTRUCTURE ONLY.
//
// Compiles the per-lane kernel logic of kaiju_amd/csrc/kj_core.h with g++ and runs it as a
// single sequential "lane", so that `pytest -m "not gpu"` can check the kernel logic (packed
// index, fragment construction, SEG, MEM and Greedy state machines) against the oracle on a
// machine without a GPU.  This library is built and loaded by tests only; the product
// library (libkaiju_gpu.so) contains no CPU path and fails loudly without a HIP device.
#include <sys/stat.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/kaiju_gpu.h"
#include "../../kaiju_amd/csrc/host_index.h"
#include "../../kaiju_amd/csrc/host_tables.h"
#include "../../kaiju_amd/csrc/kj_core.h"
#include "../../kaiju_amd/csrc/kj_greedy3.h"
#include "../../kaiju_amd/csrc/fmi_stream.h"

using namespace kj;

struct EmuIndex {
  FmiFile file;
  PackedIndex packed;
  ConstTables ct;
  SegTables st;
  std::vector<double> lnfact;
  bool xmode = false;         // ids are sequence numbers (kaijux)
};

extern "C" {

static void *emu_index_load_mode(const char *path, char *err, int errlen, bool xmode) {
  EmuIndex *ix = new EmuIndex();
  std::string msg;
  int rc = ix->file.load(path, msg);
  if (rc == 0) rc = ix->packed.build(ix->file.view(), msg);
  if (rc == 0 && xmode) { ix->packed.to_sequence_ids(); ix->xmode = true; }     // (before the text arrays, as index_from_packed does)
  if (rc == 0) ix->packed.build_klines();          // (the device builds its k-mer lines in capi.hip: k_kline_build)
  if (rc == 0 && !getenv("KAIJU_EMU_NO_TEXT")) {
    // (... and its text arrays: k_suffix_walk, k_text_build; wide: k_seq_walk_len, k_seq_walk_fill - text positions of every
    //  2^KAIJU_EMU_TV_SHIFT-th row, default every second one so that searches both wait for such a row and stand on one)
    if (ix->packed.wide) ix->packed.build_text_wide(getenv("KAIJU_EMU_TV_SHIFT") ? (uint32_t)atoi(getenv("KAIJU_EMU_TV_SHIFT")) : 1u);
    else ix->packed.build_text();
  }
  if (rc == 0) rc = build_const_tables(ix->packed.trans, ix->ct, msg);
  if (rc == 0) rc = build_seg_tables(ix->lnfact, ix->st, msg);
  if (rc != 0) { snprintf(err, (size_t)errlen, "%s", msg.c_str()); delete ix; return nullptr; }
  // (tests/tools/big_rows_emu.py: an index of 2^33 rows and more next to the oracle's copy in 62 GB of host memory)
  if (getenv("KAIJU_EMU_DROP_FILE")) { BigVec<uint8_t>().swap(ix->file.bwt); BigVec<uint8_t>().swap(ix->file.sa); }
  return ix;
}
void *emu_index_load(const char *path, char *err, int errlen) { return emu_index_load_mode(path, err, errlen, false); }
// kaijux semantics: ids are sequence numbers
void *emu_index_load_x(const char *path, char *err, int errlen) { return emu_index_load_mode(path, err, errlen, true); }
void emu_index_free(void *h) { delete (EmuIndex *)h; }
uint32_t emu_index_warnings(void *h) { return ((EmuIndex *)h)->packed.warnings; }
// 1: the text arrays of text verification were built (sa_full / text / row_seq); number of rows whose row_seq says "no sequence"
int emu_has_text(void *h) { return ((EmuIndex *)h)->packed.sa_full.empty() && ((EmuIndex *)h)->packed.sa_tpos5.empty() ? 0 : 1; }
// the text arrays themselves (tests: the sequence walk of the wide layout builds what the per-row walk of the narrow one does)
const uint8_t *emu_text(void *h, uint64_t *n) { const auto &t = ((EmuIndex *)h)->packed.text; *n = t.size(); return t.data(); }
// text position of the suffix of row r: narrow sa_full[r]; wide the 5-byte entry of r (kTposNone-like ~0 when r is not a kept row)
uint64_t emu_text_pos(void *h, uint64_t r) {
  const PackedIndex &pk = ((EmuIndex *)h)->packed;
  if (!pk.sa_full.empty()) return r < pk.sa_full.size() ? pk.sa_full[(size_t)r] : ~0ull;
  if (pk.sa_tpos5.empty() || (r & ((1ull << pk.tv_shift) - 1ull)) || r >= pk.bwtlen) return ~0ull;
  const uint8_t *e = pk.sa_tpos5.data() + (size_t)(r >> pk.tv_shift) * 5;
  return (uint64_t)e[0] | (uint64_t)e[1] << 8 | (uint64_t)e[2] << 16 | (uint64_t)e[3] << 24 | (uint64_t)e[4] << 32;
}
uint64_t emu_rows_without_sequence(void *h) {
  uint64_t n = 0;
  const PackedIndex &pk = ((EmuIndex *)h)->packed;
  for (uint64_t r = 0; r < pk.bwtlen && r < pk.row_seq.size(); r++) n += pk.row_seq[(size_t)r] == 0xffffffffu;   // (behind the last row: padding)
  return n;
}
// the product's ln(n!) table (host_tables.cpp: build_seg_tables), entry n; -1 beyond it; emu_lnfact_n(): its length
uint32_t emu_lnfact_n(void *h) { return (uint32_t)((EmuIndex *)h)->lnfact.size(); }
double emu_lnfact(void *h, uint32_t n) { EmuIndex *ix = (EmuIndex *)h; return n < ix->lnfact.size() ? ix->lnfact[n] : -1.0; }

// rank primitives on the packed layout
uint64_t emu_rank(void *h, uint32_t c, uint64_t k) {
  EmuIndex *ix = (EmuIndex *)h;
  DevIndex d = ix->packed.host_view();
  if (c == 0) return rank_term(d, k);
  return rank_c(d, c, k);
}
uint32_t emu_symbol(void *h, uint64_t k) { DevIndex d = ((EmuIndex *)h)->packed.host_view(); return symbol_at(d, k); }

// SEG on an ASCII peptide (letters of the 20 amino acids)
// the SEG code's cooperation object of the emulation: one lane, with the prefix-count scratch of s_Trim (the device's k_seg
// has it in LDS) unless KAIJU_EMU_NO_SEG_PREFIX is set (every sub-window then counts its own letters: the two must agree)
static CoopSerial emu_coop() {
  static thread_local uint64_t pref[2 * (kSegPacked + 1)];
  CoopSerial c;
  c.prefix = getenv("KAIJU_EMU_NO_SEG_PREFIX") ? nullptr : pref;
  return c;
}
int emu_seg(void *h, const char *aa, int len, int32_t *left, int32_t *right) {
  EmuIndex *ix = (EmuIndex *)h;
  std::vector<uint8_t> codes((size_t)len + 1);
  for (int i = 0; i < len; i++) codes[(size_t)i] = ix->packed.trans[(unsigned char)aa[i] & 127];
  bool ov = false;
  const SegCtx cx = seg_ctx(ix->st, ix->st.ent_g, ix->st.lnfact);
  int32_t work[2 * kSegMaxRegions];
  std::vector<uint8_t> cls((size_t)len + 1);
  const bool use_cls = !getenv("KAIJU_EMU_SEG_NOCLS");
  int n = seg_regions(cx, emu_coop(), codes.data(), len, left, right, ov, work, use_cls ? cls.data() : nullptr, [] {});
  return ov ? -1 : n;
}

// verbose output (columns 6/7) of the next emu_classify calls: first-generation lanes write here
static VerboseOut g_vb{nullptr, nullptr, nullptr, nullptr, 0};
void emu_set_verbose(uint32_t *n_acc, uint32_t *acc, uint32_t *text_len, uint8_t *text, uint32_t text_cap) {
  g_vb = VerboseOut{n_acc, acc, text_len, text, text_cap};
}
const char *emu_seq_name(void *h, uint32_t iseq) {
  EmuIndex *ix = (EmuIndex *)h;
  return iseq < ix->packed.names.size() ? ix->packed.names[iseq].c_str() : nullptr;
}
const char *emu_alphabet(void *h) { return ((EmuIndex *)h)->packed.alphabet.c_str(); }

// fragments of one batch (stage 1): returns per-read lists as ASCII for comparison
// frag_dump receives, per read, a "#" line followed by one "key:PEPTIDE" line per fragment
int emu_classify(void *h, const kaiju_gpu_params *gp, const char *seqs, const uint64_t *off, uint32_t n,
                 int paired, kaiju_gpu_hit *out, uint32_t si_cap, uint32_t pool_cap, uint32_t match_cap,
                 uint32_t *n_retry_out, char *frag_dump, uint64_t frag_dump_cap) {
  EmuIndex *ix = (EmuIndex *)h;
  DevIndex d = ix->packed.host_view();
  Params p;
  p.mode = gp->mode; p.m = gp->min_fragment_length; p.mismatches = gp->mismatches; p.min_score = gp->min_score;
  p.seed_length = gp->seed_length; p.seg = gp->seg; p.max_matches_SI = gp->max_matches_SI; p.max_match_ids = gp->max_match_ids;
  if (p.mismatches > (uint32_t)kMaxMismatch) return KAIJU_GPU_ERR_UNSUPPORTED;
  // as kaiju_gpu_create does
  if (ix->xmode && p.mode == 0 && !getenv("KAIJU_EMU_NO_XORDER")) p.flags |= kParamXOrder;   // (switch: shows that a test sees the order)
  if (gp->input_is_protein) { p.flags |= kParamProtein; if (paired) return KAIJU_GPU_ERR_ARG; }
  Batch b;
  b.seqs = (const uint8_t *)seqs; b.off = off; b.n_reads = n; b.paired = paired;
  std::vector<uint8_t> pep((size_t)pep_base(off, n) + 512, 0);
  std::vector<Frag> frags((size_t)frag_base(off, n, p.m) + 8);
  std::vector<ReadMeta> meta(n);
  std::vector<Hit> hits(n);
  memset(hits.data(), 0, sizeof(Hit) * n);
  b.pep = pep.data(); b.frags = frags.data(); b.meta = meta.data(); b.hits = hits.data();
  uint32_t maxlen = 0;
  for (uint32_t r = 0; r < n; r++) {
    uint32_t l1 = (uint32_t)(off[2 * r + 1] - off[2 * r]), l2 = (uint32_t)(off[2 * r + 2] - off[2 * r + 1]);
    if (l1 > maxlen) maxlen = l1;
    if (l2 > maxlen) maxlen = l2;
  }
  uint32_t err = 0;
  // stage 1 -> SEG pass -> (MEM) apply, exactly the kernel sequence of capi.hip
  uint32_t seg_count = 0;
  const uint32_t seg_cap = (uint32_t)(frags.size() / 2 + 8);
  std::vector<SegWork> seg_items(seg_cap);
  std::vector<SegRec> seg_recs(seg_cap);
  SegQueue sq{seg_items.data(), seg_recs.data(), &seg_count, seg_cap};
  const SegCtx cx = seg_ctx(ix->st, ix->st.ent_g, ix->st.lnfact);
  // peptides are staged (here: linear scratch) and copied out, as the kernel does with its LDS area
  std::vector<uint8_t> stage((size_t)4 * maxlen + 256);
  const bool staged = !getenv("KAIJU_EMU_NOSTAGE");
  // which flow (capi.hip: launch_batch): the fast stage 1 for mates up to kS1MaxLen nucleotides; MEM on the second-generation
  // lanes then looks at SEG lazily
  const char *lane_env = getenv("KAIJU_EMU_LANE");
  // (verbose output: the VERBOSE instantiations of the second-generation lanes + mem_verbose_read, as capi.hip; KAIJU_EMU_VERBOSE_V1
  //  = the first-generation lanes, which wrote columns 6 / 7 until round 6)
  const bool vb_v2 = g_vb.n_acc && !getenv("KAIJU_EMU_VERBOSE_V1");
  const bool mem_v2 = p.mode == 0 && d.blocks64 && (d.kmer32 || (d.mb_base && d.kmer64)) && !lane_env && (!g_vb.n_acc || vb_v2);
  const bool fast1 = !(p.flags & kParamProtein) && !getenv("KAIJU_EMU_STAGE1_OLD") && maxlen <= kS1MaxLenLong && p.m >= 1 && p.m <= 64;
  const bool long1 = maxlen > kS1MaxLen;
  const bool lazy = fast1 && mem_v2 && p.seg && !getenv("KAIJU_EMU_LAZY_OFF");
  const bool trig1 = fast1 && p.seg && !lazy;
  Stage1Tables s1tab;
  build_stage1_tables(ix->ct, ix->st, s1tab);
  if (p.flags & kParamProtein) {
    uint8_t code[256];
    memset(code, 0, sizeof code);
    for (uint32_t a = 0; a < 20; a++) protein_code_entry(ix->ct, a, code);
    for (uint32_t r = 0; r < n; r++) build_fragments_protein(ix->ct, code, p, TrigCtx{ix->st.ent_g32, ix->st.ent_locut32}, b, sq, r, &err);
  } else if (fast1) {
    const Stage1Tables &s1 = s1tab;
    uint32_t codes[2 * kS1ListCap];
    alignas(4) uint8_t cnt[kS1CntStride];
    alignas(4) uint8_t tsbuf[kTsBufLong];
    S1Lane ln{codes, 1, cnt, tsbuf};
    for (uint32_t r = 0; r < n; r++) {
      for (auto &x : codes) x = 0xdeadbeefu;
      memset(cnt, 0xee, sizeof cnt);
      memset(tsbuf, 0xee, sizeof tsbuf);
      if (trig1 && long1) build_fragments_fast<true, kS1UnitsLong>(s1, p, b, sq, r, &err, ln);
      else if (long1) build_fragments_fast<false, kS1UnitsLong>(s1, p, b, sq, r, &err, ln);
      else if (trig1) build_fragments_fast<true>(s1, p, b, sq, r, &err, ln);
      else build_fragments_fast<false>(s1, p, b, sq, r, &err, ln);
    }
  } else
  for (uint32_t r = 0; r < n; r++) build_fragments(ix->ct, p, TrigCtx{ix->st.ent_g32, ix->st.ent_locut32}, b, sq, r, &err, staged ? stage.data() : nullptr, 4, (uint32_t)(stage.size() / 4));
  if (p.seg && !lazy) {
    if (getenv("KAIJU_EMU_SEG_COUNT")) {                 // how much work the eager SEG pass gets (tools: sizing of k_seg)
      uint64_t tl = 0;
      for (uint32_t s = 0; s < seg_count && s < seg_cap; s++) tl += b.frags[b.meta[seg_items[s].read].frag + seg_items[s].frag].len;
      fprintf(stderr, "[emu] SEG pass: %u fragments of %u reads, %.1f residues each\n", seg_count, n, seg_count ? (double)tl / seg_count : 0.0);
    }
    int32_t segwork[4 * kSegMaxRegions];
    std::vector<uint8_t> segstage(64);   // small on purpose: exercises both the staged and the direct path
    std::vector<uint8_t> segcls(64);
    for (uint32_t s = 0; s < seg_count && s < seg_cap; s++)
      seg_compute(cx, emu_coop(), b, p, sq, s, segstage.data(), (uint32_t)segstage.size(), segwork, segcls.data(), [] {});
    if (p.mode == 0) for (uint32_t r = 0; r < n; r++) seg_apply_mem(ix->ct, p, b, sq, r, &err);
  }
  if (frag_dump) {
    const char *alpha = ix->packed.alphabet.c_str();
    uint64_t w = 0;
    for (uint32_t r = 0; r < n; r++) {
      const Frag *F = frags.data() + meta[r].frag;
      const uint8_t *pp = pep.data() + meta[r].pep;
      if (w + 4 >= frag_dump_cap) return KAIJU_GPU_ERR_NOMEM;
      frag_dump[w++] = '#'; frag_dump[w++] = '\n';
      for (uint32_t f = 0; f < (meta[r].nfrag & ~kNfragSegPending); f++) {
        char tmp[32];
        int l = snprintf(tmp, sizeof tmp, "%u:", F[f].key);
        if (w + (uint64_t)l + F[f].len + 2 >= frag_dump_cap) return KAIJU_GPU_ERR_NOMEM;
        memcpy(frag_dump + w, tmp, (size_t)l); w += (uint64_t)l;
        for (uint32_t x = 0; x < F[f].len; x++) frag_dump[w++] = alpha[pp[F[f].start + x]];
        frag_dump[w++] = '\n';
      }
    }
    frag_dump[w] = 0;
  }
  // first pass with the given scratch sizes, overflowing reads go to the retry list
  std::vector<uint32_t> retry(n);
  uint32_t counter = 0, retry_count = 0;
  alignas(16) uint8_t win[kWin];
  std::vector<SIEntry> si(si_cap);
  std::vector<GItem> pool(pool_cap);
  std::vector<uint16_t> ord(pool_cap);
  std::vector<GMatch> matches(match_cap);
  std::vector<GBest> bestv(64);
  for (int pass = 0; pass < 2; pass++) {
    WorkList wl;
    wl.counter = &counter;
    uint32_t nitems = pass == 0 ? n : retry_count;
    wl.n_items = nitems; wl.n_items_ptr = nullptr;
    wl.reads = pass == 0 ? nullptr : retry.data();
    wl.retry_list = pass == 0 ? retry.data() : nullptr;
    wl.retry_count = &retry_count;
    counter = 0;
    if (pass == 1) {
      if (n_retry_out) *n_retry_out = retry_count;
      if (retry_count == 0) break;
      // retry with scratch sized for the worst case
      si.assign(2 * (size_t)maxlen + 64, SIEntry{});
      pool.assign(65535, GItem{}); ord.assign(65535, 0);
      matches.assign((size_t)maxlen + 8, GMatch{});
    }
    if (p.mode == 0) {
      LaneScratch ls{si.data(), (uint32_t)si.size(), win};
      const char *v = getenv("KAIJU_EMU_LANE");        // "v1", "wide" or default (v2 where possible)
      const bool xo = (p.flags & kParamXOrder) != 0;
      if (vb_v2) ls.vbm = g_vb.acc;
      auto lane_v2 = [&](const Params &pp, const WorkList &w2) {
        if (vb_v2) {                                     // (k_mem_vb / k_mem_wide2_vb)
          if (d.kmer32) { if (xo) mem_lane2<false, true, false, true>(d, pp, b, w2, ls); else mem_lane2<false, false, false, true>(d, pp, b, w2, ls); }
          else { if (xo) mem_lane2<true, true, false, true>(d, pp, b, w2, ls); else mem_lane2<true, false, false, true>(d, pp, b, w2, ls); }
        }
        else if (d.kmer32) { if (xo) mem_lane2<false, true>(d, pp, b, w2, ls); else mem_lane2<false>(d, pp, b, w2, ls); }
        else { if (xo) mem_lane2<true, true>(d, pp, b, w2, ls); else mem_lane2<true>(d, pp, b, w2, ls); }
      };
      if (mem_v2 && pass == 0) {
        // (capi.hip: the lanes leave the longest matches in the hit records, k_mem_locate* behind the searches walk them)
        Params pd = p;
        pd.flags |= kParamDeferLocate;
        Params pm = pd;
        if (lazy) pm.flags |= kParamLazySeg;
        lane_v2(pm, wl);
        if (lazy) {
          // k_trigcheck, k_segflag, k_seg, k_seg_apply_list and the search of the listed reads (capi.hip)
          std::vector<uint32_t> seglist;
          for (uint32_t r = 0; r < n; r++) {
            const uint32_t vv = hits[r].reserved;
            if (vv == 0) continue;
            hits[r].reserved = 0;
            alignas(4) uint8_t row[kS1CntStride], tb[kTsBuf];
            const bool need = lazy_seg_needed(s1tab, p, b, r, &hits[r], vv, nullptr, row);      // (the device's form: trig_fragment_units)
            if (need != lazy_seg_needed(s1tab, p, b, r, &hits[r], vv, tb, row)) { fprintf(stderr, "[emu] trig_fragment_units disagrees with the staged scan\n"); abort(); }
            if (vv != kWinForce && !(vv & kWinMulti)) {
              // (the check agrees with the SEG code's own trigger test)
              const Frag *F = frags.data() + meta[r].frag;
              const uint8_t *pp = pep.data() + meta[r].pep;
              const Frag &f = F[(vv & ~kWinMulti) - 1u];
              const bool a = trig_fragment(s1tab, pp, f.start, f.len, tb, row), bb = seg_triggers(cx, pp + f.start, (int)f.len);
              if (a != bb) { fprintf(stderr, "[emu] trig_fragment disagrees with seg_triggers\n"); abort(); }
              if (need != a) { fprintf(stderr, "[emu] lazy_seg_needed disagrees with trig_fragment\n"); abort(); }
            }
            if (need) seglist.push_back(r);
          }
          for (uint32_t r : seglist) {
            Frag *F = frags.data() + meta[r].frag;
            const uint8_t *pp = pep.data() + meta[r].pep;
            const uint32_t nf = meta[r].nfrag & ~kNfragSegPending;
            uint32_t pending = 0;
            for (uint32_t k = 0; k < nf; k++) {
              uint32_t fl = kFragChecked;
              if (seg_triggers(cx, pp + F[k].start, (int)F[k].len)) {
                const uint32_t slot = seg_count++;
                if (slot >= seg_cap) err |= 2u;
                else { seg_items[slot] = SegWork{r, k}; pending = kNfragSegPending; fl = (slot + 1) << kFragSlotShift; }
              }
              F[k].flags = fl;
            }
            meta[r].nfrag = nf | pending;
            memset(&hits[r], 0, sizeof(Hit));
          }
          int32_t segwork[4 * kSegMaxRegions];
          std::vector<uint8_t> segstage(64), segcls(64);
          for (uint32_t s2 = 0; s2 < seg_count && s2 < seg_cap; s2++)
            seg_compute(cx, emu_coop(), b, p, sq, s2, segstage.data(), (uint32_t)segstage.size(), segwork, segcls.data(), [] {});
          for (uint32_t r : seglist) seg_apply_mem(ix->ct, p, b, sq, r, &err);
          uint32_t counter2 = 0, nlist = (uint32_t)seglist.size();
          WorkList w2;
          w2.counter = &counter2; w2.n_items = nlist; w2.n_items_ptr = nullptr; w2.reads = seglist.data();
          w2.retry_list = retry.data(); w2.retry_count = &retry_count;
          if (nlist) lane_v2(pd, w2);
          if (getenv("KAIJU_EMU_PRINT_LAZY")) fprintf(stderr, "[emu] lazy SEG: %u of %u reads listed\n", nlist, n);
        }
      }
      else if (!d.mb_base && !(v && !strcmp(v, "wide")) && pass == 0) mem_lane<uint32_t>(d, p, b, wl, ls, g_vb);
      else mem_lane<uint64_t>(d, p, b, wl, ls, g_vb);
    } else {
      GreedyScratch gs;
      gs.pool = pool.data(); gs.pool_cap = (uint32_t)pool.size(); gs.ord = ord.data();
      gs.matches = matches.data(); gs.match_cap = (uint32_t)matches.size();
      gs.best = bestv.data(); gs.win = win;
      const char *v = getenv("KAIJU_EMU_LANE");        // "v1" or default (v2 where possible)
      std::vector<GBestV> bestvv(64);
      gs.bestv = g_vb.n_acc ? bestvv.data() : nullptr;
      if (d.blocks64 && (d.kline || (d.mb_base && d.kmer64)) && !v && pass == 0 && (!g_vb.n_acc || vb_v2)) {
        alignas(16) uint32_t lds_win[kGWinStride], lds_mq[kGMqStride], lds_prio[kGPrioStride];
        for (auto &x : lds_win) x = 0xdeadbeefu;
        for (auto &x : lds_mq) x = 0xdeadbeefu;
        for (auto &x : lds_prio) x = 0xdeadbeefu;
        std::vector<u128> pool2(8 * kGSlotsAll + 4);
        std::vector<uint32_t> prio_ext(kGSlotsAll - kGSlots + 16, 0xdeadbeefu);     // (+ slack: read 16 bytes at a time)
        std::vector<GMatch2> matches2(kGMaxMAll);
        std::vector<uint16_t> mq_ext(kGMaxMAll - kGMaxM, 0xdead);
        std::vector<GBest2> best2(64);
        std::vector<GBest2W> best2w(64);
        const char *ge = getenv("KAIJU_EMU_GATE");
        GreedyScratch2 g2{reinterpret_cast<uint8_t *>(lds_win), reinterpret_cast<uint16_t *>(lds_mq), lds_prio, pool2.data(),
                          prio_ext.data(), matches2.data(), mq_ext.data(), best2.data(), best2w.data(), ge ? (uint32_t)atoi(ge) : 3u};
        uint32_t lds_sub[24];
        for (auto &x : lds_sub) x = 0xdeadbeefu;
        g2.sub = lds_sub;
        Params pg = p;
        pg.flags |= kParamDeferLocate;
        const char *g3e = getenv("KAIJU_EMU_GREEDY");       // "3": the row-pool lane (kj_greedy3.h; narrow indexes) - the product's KAIJU_GPU_GREEDY_LANE=v3
        std::vector<GBestV> bestv2(64);
        if (vb_v2) { g2.bestv = bestv2.data(); g2.vb = g_vb; g2.lane = 0; }      // (k_greedy2_vb / k_greedy2_wide_vb)
        if (vb_v2 && d.mb_base) greedy_lane2<false, true, true>(d, ix->ct, pg, sq, b, wl, g2);
        else if (vb_v2) greedy_lane2<false, false, true>(d, ix->ct, pg, sq, b, wl, g2);
        else if (d.mb_base) greedy_lane2<false, true>(d, ix->ct, pg, sq, b, wl, g2);
        else if (!(g3e && !strcmp(g3e, "3"))) greedy_lane2(d, ix->ct, pg, sq, b, wl, g2);
        else {
          // third generation (kj_greedy3.h): the read's state in a row of "LDS"; here a pool of one row
          alignas(16) uint32_t r_prio[kG3PrioWords], r_win[kG3WinWords], r_mq[kG3MqWords], r_st[kG3StWords], r_cls[1], r_cnt[32];
          uint16_t r_tmp[64];
          for (auto &x : r_prio) x = 0;
          for (auto &x : r_win) x = 0;
          for (auto &x : r_mq) x = 0;
          for (auto &x : r_st) x = 0;
          for (auto &x : r_cnt) x = 0;
          r_cls[0] = C3_IDLE;
          const char *sp = getenv("KAIJU_EMU_G3_SPLIT");
          G3Ctx gx{r_prio, r_win, r_mq, r_st, r_cls, r_cnt, r_tmp, pool2.data(), prio_ext.data(), matches2.data(), mq_ext.data(), best2.data(),
                   0u, 1u, sp ? (uint32_t)atoi(sp) : 1u, nullptr};
          greedy_lane3(d, ix->ct, pg, sq, b, wl, gx);
        }
      } else greedy_lane(d, ix->ct, p, sq, b, wl, gs, g_vb);
    }
  }
  // k_mem_verbose (capi.hip): columns 6 / 7 of the reads whose matches wait in their records, in front of the locate
  if (mem_v2 && vb_v2) for (uint32_t r = 0; r < n; r++) { if (d.mb_base) mem_verbose_read<true>(d, p, b, r, g_vb); else mem_verbose_read<false>(d, p, b, r, g_vb); }
  // (Greedy: the VERBOSE lane wrote column 7 itself; column 6 from its records)
  if (p.mode != 0 && vb_v2) for (uint32_t r = 0; r < n; r++) { if (d.mb_base) mem_verbose_read<true, false>(d, p, b, r, g_vb); else mem_verbose_read<false, false>(d, p, b, r, g_vb); }
  // k_mem_locate (capi.hip): behind the main, the second and the retry search
  const bool locate_pass = true;
  // (indexes without the row -> sequence table are located by teams of lanes on the device: k_mem_locate_wide / _team; here a
  //  team of four whose walks run one after the other, KAIJU_EMU_LOCATE_SERIAL=1: the one-lane function)
  if (locate_pass) for (uint32_t r = 0; r < n; r++) {
    const bool serial = getenv("KAIJU_EMU_LOCATE_SERIAL") != nullptr;
    if (d.mb_base && d.row_tax && !serial) { if (!mem_locate_read<true>(d, p, &hits[r], 8)) mem_locate_read<true, true>(d, p, &hits[r]); }   // (k_mem_locate<true>, k_mem_locate_list<true>)
    else if (d.mb_base) { if (serial) mem_locate_read<true>(d, p, &hits[r]); else { TeamSerial<4> tm; mem_locate_read_team<true, 4>(d, p, &hits[r], tm); } }
    else if (serial) mem_locate_read<false>(d, p, &hits[r]);
    else if (d.row_tax) { if (!mem_locate_read<false>(d, p, &hits[r], 8)) mem_locate_read<false, true>(d, p, &hits[r]); }   // (k_mem_locate, k_mem_locate_list)
    else { TeamSerial<4> tm; mem_locate_read_team<false, 4>(d, p, &hits[r], tm); }
  }
  // the exact pass (kj_core.h: BigSeg), as capi.hip's k_redo_* kernels run it behind the retry pass
  if (p.seg && n > 0) {
    std::vector<uint32_t> redo;
    std::vector<char> seen(n, 0);
    for (uint32_t s2 = 0; s2 < seg_count && s2 < seg_cap; s2++) {
      if (!seg_recs[s2].overflow) continue;
      const uint32_t r = seg_items[s2].read;
      if (!seen[r]) { seen[r] = 1; redo.push_back(r); }
    }
    err &= ~1u;                                            // (kaiju_gpu_get_stats: settled by the exact pass)
    if (!redo.empty()) {
      const bool protein = (p.flags & kParamProtein) != 0;
      uint32_t count2 = 0, pairs = 0;
      const uint32_t cap2 = 1u << 12;
      const char *pe = getenv("KAIJU_EMU_REDO_POOL");       // (tests: a pool that is too small)
      const uint32_t pool_cap = pe ? (uint32_t)atoi(pe) : (1u << 20);
      std::vector<SegWork> items2(cap2);
      std::vector<uint2> index2(cap2);
      std::vector<int32_t> pool2((size_t)2 * pool_cap + 2);
      SegQueue sq2{items2.data(), nullptr, &count2, cap2};
      BigSeg big{index2.data(), pool2.data(), &pairs, pool_cap};
      uint8_t code[256];
      memset(code, 0, sizeof code);
      for (uint32_t a = 0; a < 20; a++) protein_code_entry(ix->ct, a, code);
      for (uint32_t r : redo) {
        memset(&hits[r], 0, sizeof(Hit));
        if (protein) build_fragments_protein(ix->ct, code, p, TrigCtx{ix->st.ent_g32, ix->st.ent_locut32}, b, sq2, r, &err);
        else build_fragments(ix->ct, p, TrigCtx{ix->st.ent_g32, ix->st.ent_locut32}, b, sq2, r, &err, nullptr, 4, 0);
      }
      const uint32_t max_frag = protein ? maxlen : maxlen / 3 + 2;
      const int cap_ints = (int)(2 * max_frag + 4);
      std::vector<int32_t> work2((size_t)4 * cap_ints);
      std::vector<uint8_t> cls2((size_t)max_frag + 64);
      for (uint32_t s2 = 0; s2 < count2 && s2 < cap2; s2++)
        seg_compute_big(cx, emu_coop(), b, sq2, big, s2, work2.data(), cap_ints, cls2.data(), &err, [] {});
      if (p.mode == 0) for (uint32_t r : redo) seg_apply_mem_big(ix->ct, p, b, big, r);
      WorkList wl;
      uint32_t rcount = (uint32_t)redo.size();
      counter = 0;
      wl.counter = &counter; wl.n_items = rcount; wl.n_items_ptr = nullptr; wl.reads = redo.data();
      wl.retry_list = nullptr; wl.retry_count = &retry_count;
      si.assign(2 * (size_t)maxlen + 64, SIEntry{});
      pool.assign(65535, GItem{}); ord.assign(65535, 0);
      matches.assign((size_t)maxlen + 8, GMatch{});
      if (p.mode == 0) {
        LaneScratch ls{si.data(), (uint32_t)si.size(), win};
        mem_lane<uint64_t>(d, p, b, wl, ls, g_vb);
      } else {
        GreedyScratch gs;
        gs.pool = pool.data(); gs.pool_cap = (uint32_t)pool.size(); gs.ord = ord.data();
        gs.matches = matches.data(); gs.match_cap = (uint32_t)matches.size();
        gs.best = bestv.data(); gs.win = win;
        std::vector<GBestV> bestvv(64);
        gs.bestv = g_vb.n_acc ? bestvv.data() : nullptr;
        greedy_lane(d, ix->ct, p, sq2, b, wl, gs, g_vb, &big);
      }
    }
  }
  static_assert(sizeof(Hit) == sizeof(kaiju_gpu_hit), "hit layout");
  memcpy(out, hits.data(), sizeof(Hit) * n);
  return err ? -100 : 0;
}

}  // extern "C"

// device image of an index: pack, write, read back; 0 if every array survives unchanged
extern "C" int emu_image_roundtrip(const char *fmi, const char *image) {
  FmiFile f; std::string msg;
  if (f.load(fmi, msg)) return -1;
  PackedIndex a, b;
  if (a.build(f.view(), msg)) return -2;
  { struct stat st; if (stat(fmi, &st) == 0) a.src_fmi_bytes = (uint64_t)st.st_size; }     // (as kaiju_gpu_index_write_image does)
  if (a.write_image(image, msg)) return -3;
  if (b.read_image(image, msg)) return -4;
  if (a.src_fmi_bytes != b.src_fmi_bytes) return 4;
  auto same = [](const auto &x, const auto &y) { return x.size() == y.size() && (x.empty() || !memcmp(x.data(), y.data(), x.size() * sizeof(x[0]))); };
  if (!same(a.blocks64, b.blocks64) || !same(a.sa_taxid, b.sa_taxid) ||
      !same(a.sa_iseq, b.sa_iseq) || !same(a.sa_pos, b.sa_pos) || !same(a.seq_taxid, b.seq_taxid) || !same(a.seq_valid, b.seq_valid) ||
      !same(a.term_pos, b.term_pos) || !same(a.kmer32, b.kmer32) || !same(a.kmer64, b.kmer64) || !same(a.mb_base, b.mb_base)) return 1;
  if (a.names != b.names || a.alphabet != b.alphabet || memcmp(a.C, b.C, sizeof a.C) || memcmp(a.trans, b.trans, 128)) return 2;
  if (a.bwtlen != b.bwtlen || a.n_sa != b.n_sa || a.sa_skip != b.sa_skip || a.nseq != b.nseq || a.chpt_exp != b.chpt_exp ||
      a.alen != b.alen || a.warnings != b.warnings || a.kmer_k != b.kmer_k || a.mb_shift != b.mb_shift || a.wide != b.wide) return 3;
  return 0;
}

// The streamed load of a .fmi (kaiju_amd/csrc/fmi_stream.h / .hip) on the host: the file's BWT and samples taken piece by piece,
// the steps of k_pack_sa / k_pack_blocks / k_pack_scan / k_pack_finish / k_pack_add_c one after the other with the shared
// per-block logic, and the outcome compared word for word with what PackedIndex::build packs.  piece: bytes per piece (a
// multiple of 16384).  0 = identical; > 0 names the array that differs; < 0 a failed step.
extern "C" int emu_stream_pack_check(const char *fmi, uint64_t piece) {
  FmiFile full, lz; std::string msg;
  if (full.load(fmi, msg)) return -1;
  if (lz.load(fmi, msg, true)) return -2;
  if (!lz.bwt.empty() || !lz.sa.empty() || lz.bwt_off == 0 || lz.sa_off == 0) return -3;
  PackedIndex a, b;
  if (a.build(full.view(), msg)) return -4;
  if (b.build_streamed(lz, fmi, msg)) return -5;
  if (b.wide != a.wide || b.mb_shift != a.mb_shift || b.n_sa != a.n_sa || b.sa_skip != a.sa_skip || b.warnings != a.warnings ||
      b.names != a.names || b.seq_taxid != a.seq_taxid || b.seq_valid != a.seq_valid || memcmp(a.trans, b.trans, 128)) return 20;
  FILE *fp = fopen(fmi, "rb");
  if (!fp) return -6;
  const FmiStreamSource &src = b.stream;
  const uint64_t bwtlen = b.bwtlen, nb64 = (bwtlen >> 6) + 1, n_sa = b.n_sa;
  const bool wide = b.wide;
  const uint32_t mb_grp_shift = b.mb_shift - kPackGroupSymShift;
  const uint64_t gsym = 1ull << kPackGroupSymShift;
  if (piece < gsym || piece % gsym) return -7;
  std::vector<uint8_t> raw((size_t)piece + 64);
  // k_pack_sa
  std::vector<uint32_t> sa_iseq((size_t)n_sa), sa_pos((!wide && src.pbits <= 32) ? (size_t)n_sa : 0);
  std::vector<uint64_t> sa_taxid(!wide ? (size_t)n_sa + 2 : 0, ~0ull);
  {
    const uint64_t per_piece = piece / (uint64_t)src.nbytes;
    for (uint64_t i0 = 0; i0 < n_sa; i0 += per_piece) {
      const uint64_t n = std::min<uint64_t>(per_piece, n_sa - i0);
      if (fseeko(fp, (off_t)(src.sa_off + i0 * src.nbytes), SEEK_SET) || fread(raw.data(), 1, (size_t)(n * src.nbytes), fp) != n * src.nbytes) { fclose(fp); return -8; }
      for (uint64_t x = 0; x < n; x++) {
        uint32_t is = 0, ps = 0;
        pack_sa_entry(raw.data() + x * src.nbytes, src.nbytes, src.pbits, is, ps);
        sa_iseq[(size_t)(i0 + x)] = is;
        if (!sa_pos.empty()) sa_pos[(size_t)(i0 + x)] = ps;
        if (!sa_taxid.empty()) sa_taxid[(size_t)(i0 + x)] = (is < b.nseq && b.seq_valid[is]) ? b.seq_taxid[is] : ~0ull;
      }
    }
  }
  // the BWT
  std::vector<RankBlock64> blocks((size_t)nb64);
  std::vector<uint64_t> term_pos(b.nseq, ~0ull), mbcount(wide ? (size_t)((bwtlen >> b.mb_shift) + 1) * 20 : 0, 0);
  uint64_t carry[kPackChannels] = {0};
  bool bad = false;
  for (uint64_t p0 = 0; p0 < bwtlen; p0 += piece) {
    const uint64_t len = std::min<uint64_t>(piece, bwtlen - p0);
    const bool last = p0 + len == bwtlen;
    if (fseeko(fp, (off_t)(src.bwt_off + p0), SEEK_SET) || fread(raw.data(), 1, (size_t)len, fp) != len) { fclose(fp); return -9; }
    const uint64_t blk0 = p0 >> 6;
    const uint32_t nblk = (uint32_t)((last ? nb64 : (p0 + len) >> 6) - blk0);
    const uint32_t ngroups = (nblk + kPackGroupBlocks - 1) / kPackGroupBlocks;
    const uint64_t grp0 = p0 >> kPackGroupSymShift;
    std::vector<uint32_t> grp_tot((size_t)ngroups * kPackChannels, 0);
    std::vector<uint64_t> grp_abs((size_t)ngroups * kPackChannels, 0);
    // k_pack_blocks: counts in front of the block inside its group
    for (uint32_t g = 0; g < ngroups; g++) {
      uint32_t run[kPackChannels] = {0};
      for (uint32_t t = 0; t < kPackGroupBlocks; t++) {
        const uint32_t bi = g * kPackGroupBlocks + t;
        if (bi >= nblk) break;
        const uint64_t h0 = p0 + (uint64_t)bi * 64;
        const uint32_t nsym = h0 >= bwtlen ? 0u : (uint32_t)std::min<uint64_t>(64, bwtlen - h0);
        uint64_t pl[5]; uint32_t cnt[kPackChannels];
        if (!pack_block_letters(raw.data() + (size_t)bi * 64, nsym, src.lcode, pl, cnt)) bad = true;
        RankBlock64 &r = blocks[(size_t)(blk0 + bi)];
        for (int q = 0; q < 5; q++) r.plane[q] = pl[q];
        for (uint32_t c = 0; c < 20; c++) r.cnt[c] = run[c + 1];
        r.pad[0] = run[0]; r.pad[1] = 0;
        for (uint32_t c = 0; c < kPackChannels; c++) run[c] += cnt[c];
      }
      for (uint32_t c = 0; c < kPackChannels; c++) grp_tot[(size_t)g * kPackChannels + c] = run[c];
    }
    // k_pack_scan
    for (uint32_t g = 0; g < ngroups; g++) {
      const uint64_t gg = grp0 + g;
      for (uint32_t c = 0; c < kPackChannels; c++) grp_abs[(size_t)g * kPackChannels + c] = carry[c];
      if (wide && (gg & ((1ull << mb_grp_shift) - 1ull)) == 0)
        for (uint32_t c = 0; c < 20; c++) mbcount[(size_t)(gg >> mb_grp_shift) * 20 + c] = carry[c + 1];
      for (uint32_t c = 0; c < kPackChannels; c++) carry[c] += grp_tot[(size_t)g * kPackChannels + c];
    }
    // k_pack_finish
    for (uint32_t bi = 0; bi < nblk; bi++) {
      const uint32_t g = bi >> kPackGroupShift;
      const uint64_t *ab = grp_abs.data() + (size_t)g * kPackChannels;
      RankBlock64 &r = blocks[(size_t)(blk0 + bi)];
      for (uint32_t c = 0; c < 20; c++) r.cnt[c] += (uint32_t)(ab[c + 1] - (wide ? mbcount[(size_t)((grp0 + g) >> mb_grp_shift) * 20 + c] : 0ull));
      uint64_t z = ab[0] + r.pad[0];
      uint64_t zm = ~(r.plane[0] | r.plane[1] | r.plane[2] | r.plane[3] | r.plane[4]);
      while (zm) {
        const uint32_t t = (uint32_t)__builtin_ctzll(zm);
        if (z < b.nseq) term_pos[(size_t)z] = p0 + (uint64_t)bi * 64 + t; else bad = true;
        z++; zm &= zm - 1ull;
      }
      r.pad[0] = 0;
    }
  }
  fclose(fp);
  if (bad) return -10;
  uint64_t C[22], sum = 0;
  for (uint32_t c = 0; c < kPackChannels; c++) sum += carry[c];
  if (sum != bwtlen || carry[0] != b.nseq) return -11;
  C[0] = 0;
  for (uint32_t c = 1; c < b.alen; c++) C[c] = C[c - 1] + carry[c - 1];
  for (uint32_t c = b.alen; c < 22; c++) C[c] = bwtlen;
  if (wide) for (size_t x = 0; x < mbcount.size(); x++) mbcount[x] += C[1 + x % 20];
  else for (auto &r : blocks) for (uint32_t c = 0; c < 20; c++) r.cnt[c] += (uint32_t)C[c + 1];
  if (memcmp(C, a.C, sizeof C)) return 1;
  if (blocks.size() != a.blocks64.size() || memcmp(blocks.data(), a.blocks64.data(), blocks.size() * sizeof(RankBlock64))) return 2;
  if (mbcount.size() != a.mb_base.size() || (!mbcount.empty() && memcmp(mbcount.data(), a.mb_base.data(), mbcount.size() * 8))) return 3;
  if (term_pos.size() != a.term_pos.size() || memcmp(term_pos.data(), a.term_pos.data(), term_pos.size() * 8)) return 4;
  if (sa_iseq.size() != a.sa_iseq.size() || (n_sa && memcmp(sa_iseq.data(), a.sa_iseq.data(), (size_t)n_sa * 4))) return 5;
  if (sa_pos.size() != a.sa_pos.size() || (!sa_pos.empty() && memcmp(sa_pos.data(), a.sa_pos.data(), sa_pos.size() * 4))) return 6;
  if (sa_taxid.size() != a.sa_taxid.size() || (!sa_taxid.empty() && memcmp(sa_taxid.data(), a.sa_taxid.data(), sa_taxid.size() * 8))) return 7;
  if (PackedIndex::count(b.blocks64, b.lazy.blocks64) != a.blocks64.size() || PackedIndex::count(b.sa_iseq, b.lazy.sa_iseq) != a.sa_iseq.size() ||
      PackedIndex::count(b.sa_pos, b.lazy.sa_pos) != a.sa_pos.size() || PackedIndex::count(b.sa_taxid, b.lazy.sa_taxid) != a.sa_taxid.size() ||
      PackedIndex::count(b.term_pos, b.lazy.term_pos) != a.term_pos.size()) return 8;
  return 0;
}

// LCA of the device path on the host arrays of the table (tests/test_capi.py)
#include "../../kaiju_amd/csrc/taxonomy.h"
extern "C" uint64_t emu_lca(kaiju_taxonomy *t, const uint64_t *ids, uint32_t n) {
  static kaiju_taxonomy *cached = nullptr;
  static std::vector<uint64_t> key, parent_id;
  static std::vector<uint32_t> parent_slot, depth;
  if (cached != t) { kj_taxonomy_table(t, key, parent_id, parent_slot, depth); cached = t; }
  DevTaxonomy d{key.data(), parent_id.data(), parent_slot.data(), depth.data(), (uint32_t)(key.size() - 1)};
  return tax_lca(d, ids, n);
}

#ifdef KJ_HIST
namespace kj { unsigned long long kj_hist[16][64]; }
extern "C" const unsigned long long *emu_hist() { return &kj::kj_hist[0][0]; }
#endif
