// ConsumerThread_gpu.cpp — the reference-side shim of BASELINE.json's north star: the reference's
// `kaiju` keeps its ingest (kaiju.cpp:288-394), its producer/consumer queue, Config, nodes.dmp / lca_from_ids
// (util.cpp:194-263), output formatting and flush_output(); ConsumerThread::doWork()'s per-read
// getAllFragmentsBits / classify_* loop (ConsumerThread.cpp:630-749) is replaced by this batching consumer
// over the C-ABI of include/kaiju_gpu.h.
//
// This file is OUR code.  integration/apply_shim.py compiles it together with a scratch copy of the
// reference's sources (three one-line edits: Config gets a `gpu_index` member, kaiju.cpp calls
// kaiju_gpu_attach() after config->init(), the CPU doWork() is renamed doWork_cpu()) and links
// -lkaiju_gpu.  Nothing of the reference is copied into this repository.
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "ConsumerThread.hpp"
#include "kaiju_gpu.h"

static void shim_die(const char *what, int rc) {
  std::cerr << "Error: " << what << ": " << kaiju_gpu_strerror(rc) << " (" << kaiju_gpu_last_error() << ")" << std::endl;
  exit(EXIT_FAILURE);                         // the reference's error convention (util.cpp:error + exit)
}

// kaiju.cpp, after readFMI() + config->init() (kaiju.cpp:236-238): hand the in-memory index structs that
// readIndexes() produced (bwt/bwt.h:10-22, fmi.h:9-18, suffixArray.h:10-33) to the GPU library
void kaiju_gpu_attach(Config *config) {
  kaiju_gpu_host_index hv;
  memset(&hv, 0, sizeof hv);
  hv.bwtlen = config->fmi->bwtlen;
  hv.nseq = config->bwt->nseq;
  hv.alen = config->bwt->alen;
  hv.alphabet = (const char *)config->bwt->alphabet;
  hv.bwt = (const uint8_t *)config->fmi->bwt;
  hv.startLcode = (const int32_t *)config->fmi->startLcode;
  hv.sa = (const uint8_t *)config->bwt->s->sa;
  hv.ncheck = config->bwt->s->ncheck;
  hv.chpt_exp = config->bwt->s->chpt_exp;
  hv.nbytes = config->bwt->s->nbytes;
  hv.pbits = config->bwt->s->pbits;
  hv.ids = (const char *const *)config->bwt->s->ids;
  int device = 0;
  if (const char *e = getenv("KAIJU_GPU_DEVICE")) device = atoi(e);
  const int rc = kaiju_gpu_index_from_host(&hv, device, &config->gpu_index);
  if (rc != 0) shim_die("kaiju_gpu_index_from_host", rc);
}

void ConsumerThread::doWork() {
  kaiju_gpu_params p;
  kaiju_gpu_default_params(&p, config->mode == GREEDY ? 1 : 0);
  p.min_fragment_length = config->min_fragment_length;
  p.mismatches = config->mismatches;
  p.min_score = config->min_score;
  p.seed_length = config->seed_length;
  p.seg = config->SEG ? 1 : 0;
  p.use_evalue = config->use_Evalue ? 1 : 0;
  p.min_evalue = config->min_Evalue;
  p.input_is_protein = config->input_is_protein ? 1 : 0;
  p.max_matches_SI = (uint32_t)config->max_matches_SI;
  p.max_match_ids = (uint32_t)config->max_match_ids;
  kaiju_gpu_ctx *ctx = NULL;
  int rc = kaiju_gpu_create(&ctx, config->gpu_index, &p);
  if (rc != 0) shim_die("kaiju_gpu_create", rc);
  size_t batch_reads = 1000000;
  if (const char *e = getenv("KAIJU_GPU_BATCH")) { long v = atol(e); if (v > 0) batch_reads = (size_t)v; }

  std::vector<ReadItem *> items;      // the batch in the order the queue delivered it
  std::vector<int64_t> slot;          // per item: its index in the device batch, -1 = stopped by the length gate
  std::string seqs;
  std::vector<uint64_t> off;
  std::vector<kaiju_gpu_hit> hits;
  std::vector<kaiju_gpu_verbose> vrec;
  std::vector<char> vtext;
  ReadItem *item = NULL;
  bool more = true;
  while (more) {
    items.clear(); slot.clear(); seqs.clear(); off.assign(1, 0);
    bool paired = false;
    size_t max_pair = 0;
    uint32_t n = 0;
    while (items.size() < batch_reads && (more = myWorkQueue->pop(&item))) {           // the reference's queue, unchanged
      assert(item != NULL);
      // length gates, ConsumerThread.cpp:640-654 (kept: such reads never reach the device)
      const bool gated = config->input_is_protein
                             ? item->sequence1.length() < config->min_fragment_length
                             : ((!item->paired && item->sequence1.length() < config->min_fragment_length * 3) ||
                                (item->paired && item->sequence1.length() < config->min_fragment_length * 3 &&
                                 item->sequence2.length() < config->min_fragment_length * 3));
      items.push_back(item);
      slot.push_back(gated ? -1 : (int64_t)n);
      if (gated) continue;
      n++;
      paired = paired || item->paired;
      seqs += item->sequence1;
      off.push_back(seqs.size());
      if (item->paired) seqs += item->sequence2;
      off.push_back(seqs.size());
      const size_t lp = item->sequence1.length() + (item->paired ? item->sequence2.length() : 0);
      if (lp > max_pair) max_pair = lp;
    }
    uint32_t vstride = 0;
    if (n > 0) {
      hits.resize(n);
      if (config->verbose) {
        // columns 6 and 7 of -v (ConsumerThread.cpp:527-536, :614-623) come from the library's verbose entry point
        vstride = kaiju_gpu_verbose_text_stride((uint32_t)max_pair, config->input_is_protein ? 1 : 0);
        vrec.resize(n);
        vtext.resize((size_t)n * vstride);
        rc = kaiju_gpu_classify_batch_verbose(ctx, seqs.data(), off.data(), n, paired ? 1 : 0, hits.data(), vrec.data(),
                                              vtext.data(), vstride);
      } else
        rc = kaiju_gpu_classify_batch(ctx, seqs.data(), off.data(), n, paired ? 1 : 0, hits.data());
      if (rc != 0) shim_die("kaiju_gpu_classify_batch", rc);
    }
    for (size_t it = 0; it < items.size(); it++) {
      item = items[it];
      if (slot[it] < 0) {                                                         // :640-654
        output << "U\t" << item->name << "\t0\n";
        delete item;
        continue;
      }
      const uint32_t r = (uint32_t)slot[it];
      const kaiju_gpu_hit &h = hits[r];
      if (h.flags & KAIJU_HIT_INEXACT)
        std::cerr << "Warning: a capacity bound of the GPU kernels was exceeded for read " << item->name << std::endl;
      uint64_t lca = 0;
      bool pass = h.n_ids > 0 && h.best > 0;
      if (pass && config->mode == GREEDY && config->use_Evalue) {
        // E-value gate, ConsumerThread.cpp:500-513 with query_len of :660,:698,:704
        query_len = config->input_is_protein ? static_cast<double>(item->sequence1.length())
                                             : static_cast<double>(item->sequence1.length()) / 3.0;
        if (!config->input_is_protein && item->paired) query_len += static_cast<double>(item->sequence2.length()) / 3.0;
        double bitscore = (LAMBDA * h.best - LN_K) / LN_2;
        double Evalue = config->db_length * query_len * pow(2, -1 * bitscore);
        if (Evalue > config->min_Evalue) pass = false;
      }
      if (pass) {
        match_ids.clear();
        match_ids.insert(h.taxid, h.taxid + h.n_ids);
        // ConsumerThread.cpp:538 / :625, the reference's own lca_from_ids (util.cpp:194-263)
        lca = (match_ids.size() == 1) ? *(match_ids.begin()) : lca_from_ids(config, node2depth, match_ids);
      }
      if (lca > 0) {                                                              // :724-739
        output << "C\t" << item->name << "\t" << lca;
        if (config->verbose) {
          std::stringstream ss;
          ss << h.best << "\t";
          for (auto it : match_ids) ss << it << ",";
          ss << "\t";
          match_dbnames.clear();
          for (uint32_t q = 0; q < vrec[r].n_acc; q++) {
            const char *nm = kaiju_gpu_index_seq_name(config->gpu_index, vrec[r].acc_iseq[q]);
            const char *pch = nm ? strrchr(nm, '_') : NULL;
            if (pch != NULL) match_dbnames.emplace(nm, pch - nm);
          }
          for (auto it : match_dbnames) ss << it << ",";
          ss << "\t";
          ss.write(vtext.data() + (size_t)r * vstride, vrec[r].text_len);
          output << "\t" << ss.str();
        }
        output << "\n";
      } else {
        output << "U\t" << item->name << "\t0\n";
      }
      delete item;
    }
    flush_output();
  }
  flush_output();
  kaiju_gpu_destroy(ctx);
}
