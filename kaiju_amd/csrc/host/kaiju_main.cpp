// kaiju_main.cpp — drop-in `kaiju` command line on top of the C-ABI (include/kaiju_gpu.h).
//
// Same options, input handling and output format as the reference driver
// (/root/reference/src/kaiju.cpp:52-452): FASTA/FASTQ (optionally gzip) single or paired input,
// names cut at the first of " /\t\r", sequences strip()'d of non-letters, one output line per
// read  "C\tname\ttaxon" / "U\tname\t0"  in input order (the reference's -z 1 order).  What
// differs is the mechanics: reads are parsed in blocks into the batch layout of the C-ABI by a
// producer thread and classified on the GPU batch-wise, instead of one ReadItem per mutex hand-off
// (kaiju.cpp:288-394, ProducerConsumerQueue.tpp:38-84).
//
// -v prints columns 4 (match length / score) and 5 (matching taxon ids); the accession and
// peptide columns of the reference's verbose mode are not produced yet.  -p (protein input)
// is not supported.
#include <getopt.h>
#include <zlib.h>

#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include <algorithm>

#include "../../../include/kaiju_gpu.h"

namespace {

void usage(const char *prog) {
  fprintf(stderr, "Kaiju (MI355X path) — classification of sequencing reads with a protein FM-index\n\n");
  fprintf(stderr, "Usage:\n   %s -t nodes.dmp -f kaiju_db.fmi -i reads.fastq [-j reads2.fastq]\n\n", prog);
  fprintf(stderr, "Mandatory arguments:\n");
  fprintf(stderr, "   -t FILENAME   Name of nodes.dmp file\n");
  fprintf(stderr, "   -f FILENAME   Name of database (.fmi) file\n");
  fprintf(stderr, "   -i FILENAME   Name of input file containing reads in FASTA or FASTQ format\n\n");
  fprintf(stderr, "Optional arguments:\n");
  fprintf(stderr, "   -j FILENAME   Name of second input file for paired-end reads\n");
  fprintf(stderr, "   -o FILENAME   Name of output file. If not specified, output will be printed to STDOUT\n");
  fprintf(stderr, "   -z INT        Accepted for compatibility (the search runs on the GPU)\n");
  fprintf(stderr, "   -a STRING     Run mode, either \"mem\"  or \"greedy\" (default: greedy)\n");
  fprintf(stderr, "   -e INT        Number of mismatches allowed in Greedy mode (default: 3)\n");
  fprintf(stderr, "   -m INT        Minimum match length (default: 11)\n");
  fprintf(stderr, "   -s INT        Minimum match score in Greedy mode (default: 65)\n");
  fprintf(stderr, "   -E FLOAT      Minimum E-value in Greedy mode (default: 0.01)\n");
  fprintf(stderr, "   -x            Enable SEG low complexity filter (enabled by default)\n");
  fprintf(stderr, "   -X            Disable SEG low complexity filter\n");
  fprintf(stderr, "   -v            Enable verbose output\n");
  exit(EXIT_FAILURE);
}

void die(const std::string &msg) {
  fprintf(stderr, "Error: %s\n\n", msg.c_str());
  exit(EXIT_FAILURE);
}

std::string now() {
  time_t t = time(nullptr);
  char buf[16] = {0};
  strftime(buf, sizeof buf, "%H:%M:%S", localtime(&t));
  return buf;
}

// line reader over zlib (reads plain files transparently)
struct LineReader {
  gzFile fp = nullptr;
  std::vector<char> buf;
  size_t pos = 0, len = 0;
  bool eof = false;
  explicit LineReader(const std::string &path) : buf(1 << 22) {
    fp = gzopen(path.c_str(), "rb");
    if (fp) gzbuffer(fp, 1 << 20);
  }
  ~LineReader() { if (fp) gzclose(fp); }
  bool fill() {
    if (eof) return false;
    const int n = gzread(fp, buf.data(), (unsigned)buf.size());
    if (n <= 0) { eof = true; len = pos = 0; return false; }
    len = (size_t)n; pos = 0;
    return true;
  }
  int peek() {
    if (pos >= len && !fill()) return EOF;
    return (unsigned char)buf[pos];
  }
  // std::getline semantics: false only if nothing could be read
  bool getline(std::string &line) {
    line.clear();
    bool any = false;
    for (;;) {
      if (pos >= len && !fill()) return any;
      any = true;
      const char *p = buf.data() + pos;
      const char *nl = (const char *)memchr(p, '\n', len - pos);
      if (nl) { line.append(p, (size_t)(nl - p)); pos = (size_t)(nl - buf.data()) + 1; return true; }
      line.append(p, len - pos);
      pos = len;
    }
  }
  void skipline() { std::string tmp; getline(tmp); }
};

struct Batch {
  std::vector<char> seqs;
  std::vector<uint64_t> off{0};
  std::vector<std::string> names;
  bool last = false;
  size_t n() const { return names.size(); }
};

inline void append_stripped(std::vector<char> &dst, const std::string &s) {   // strip(), util.cpp:25-32
  for (char c : s) if ((c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z')) dst.push_back(c);
}

// one record of one file, kaiju.cpp:288-331 / :333-386.  Returns false at end of file.
struct RecordReader {
  LineReader in;
  std::string path, line;
  bool first = true, fastq = false;
  explicit RecordReader(const std::string &p) : in(p), path(p) {}
  bool next(std::string &name, std::vector<char> &dst) {
    do { if (!in.getline(line)) return false; } while (line.empty());
    if (first) {
      if (line[0] == '@') fastq = true;
      else if (line[0] != '>') die("Auto-detection of file type for file " + path + " failed.");
      first = false;
    }
    line.erase(0, 1);
    const size_t cut = line.find_first_of(" /\t\r");
    if (cut != std::string::npos) line.erase(cut);
    name = line;
    if (fastq) {
      in.getline(line);
      append_stripped(dst, line);
      in.skipline();
      in.skipline();
    } else {
      for (;;) {
        const int c = in.peek();
        if (c == '>' || c == EOF) break;
        in.getline(line);
        append_stripped(dst, line);
      }
    }
    return true;
  }
};

struct Queue {
  std::mutex m;
  std::condition_variable cv_full, cv_empty;
  std::deque<std::unique_ptr<Batch>> q;
  size_t cap = 3;
  void push(std::unique_ptr<Batch> b) {
    std::unique_lock<std::mutex> lk(m);
    cv_full.wait(lk, [&] { return q.size() < cap; });
    q.push_back(std::move(b));
    cv_empty.notify_one();
  }
  std::unique_ptr<Batch> pop() {
    std::unique_lock<std::mutex> lk(m);
    cv_empty.wait(lk, [&] { return !q.empty(); });
    auto b = std::move(q.front());
    q.pop_front();
    cv_full.notify_one();
    return b;
  }
};

}  // namespace

int main(int argc, char **argv) {
  kaiju_gpu_params params;
  kaiju_gpu_default_params(&params, 1);
  std::string nodes_fn, fmi_fn, in1_fn, in2_fn, out_fn;
  bool verbose = false, paired = false, protein = false;
  int c;
  while ((c = getopt(argc, argv, "a:hdpxXvn:m:e:E:l:t:f:i:j:s:z:o:")) != -1) {
    switch (c) {
      case 'a':
        if (std::string(optarg) == "mem") { params.mode = 0; params.use_evalue = 0; }
        else if (std::string(optarg) == "greedy") params.mode = 1;
        else { fprintf(stderr, "-a must be a valid mode.\n"); usage(argv[0]); }
        break;
      case 'h': usage(argv[0]); break;
      case 'd': break;
      case 'v': verbose = true; break;
      case 'p': protein = true; break;
      case 'x': params.seg = 1; break;
      case 'X': params.seg = 0; break;
      case 'o': out_fn = optarg; break;
      case 'f': fmi_fn = optarg; break;
      case 't': nodes_fn = optarg; break;
      case 'i': in1_fn = optarg; break;
      case 'j': in2_fn = optarg; paired = true; break;
      case 'l': { int v = atoi(optarg); if (v < 7) die("Seed length must be >= 7."); params.seed_length = (uint32_t)v; break; }
      case 's': { int v = atoi(optarg); if (v <= 0) die("Min Score (-s) must be greater than 0."); params.min_score = (uint32_t)v; break; }
      case 'm': { int v = atoi(optarg); if (v <= 0) die("Min fragment length (-m) must be greater than 0."); params.min_fragment_length = (uint32_t)v; break; }
      case 'e': { int v = atoi(optarg); if (v < 0) die("Number of mismatches must be >= 0."); params.mismatches = (uint32_t)v; break; }
      case 'E': { double v = atof(optarg); if (v <= 0.0) die("E-value threshold must be greater than 0."); params.min_evalue = v; break; }
      case 'z': { if (atoi(optarg) <= 0) die("Number of threads (-z) must be greater than 0."); break; }
      case 'n': break;
      default: usage(argv[0]);
    }
  }
  if (nodes_fn.empty()) { fprintf(stderr, "Error: Please specify the location of the nodes.dmp file, using the -t option.\n\n"); usage(argv[0]); }
  if (fmi_fn.empty()) { fprintf(stderr, "Error: Please specify the location of the FMI file, using the -f option.\n\n"); usage(argv[0]); }
  if (in1_fn.empty()) { fprintf(stderr, "Error: Please specify the location of the input file, using the -i option.\n\n"); usage(argv[0]); }
  if (protein) die("Protein input (-p) is not supported by the GPU path.");
  if (params.use_evalue && params.mode == 0) die("E-value calculation is only possible in Greedy run mode.");

  if (verbose) fprintf(stderr, "%s Reading database\n", now().c_str());
  kaiju_taxonomy *tax = nullptr;
  if (kaiju_taxonomy_load(nodes_fn.c_str(), &tax) != 0) die("Could not open file " + nodes_fn);
  int device = 0;
  if (const char *e = getenv("KAIJU_GPU_DEVICE")) device = atoi(e);
  kaiju_gpu_index *index = nullptr;
  int rc = kaiju_gpu_index_load(fmi_fn.c_str(), device, &index);
  if (rc != 0) die(std::string("Could not load ") + fmi_fn + ": " + kaiju_gpu_strerror(rc) + " (" + kaiju_gpu_last_error() + ")");
  kaiju_gpu_index_info info;
  kaiju_gpu_index_get_info(index, &info);
  if (info.warnings && verbose) fprintf(stderr, " Warning: the index triggers a latent bug of the reference (flags %u)\n", info.warnings);
  kaiju_gpu_ctx *ctx = nullptr;
  rc = kaiju_gpu_create(&ctx, index, &params);
  if (rc != 0) die(std::string("kaiju_gpu_create: ") + kaiju_gpu_strerror(rc) + " (" + kaiju_gpu_last_error() + ")");

  FILE *out = stdout;
  if (!out_fn.empty()) {
    out = fopen(out_fn.c_str(), "w");
    if (!out) die("Could not open file " + out_fn + " for writing");
  }
  setvbuf(out, nullptr, _IOFBF, 1 << 22);

  size_t batch_reads = 1000000;
  if (const char *e = getenv("KAIJU_GPU_BATCH")) batch_reads = (size_t)std::max(1L, atol(e));
  if (verbose) fprintf(stderr, "%s Start classification on GPU %d\n", now().c_str(), device);

  Queue queue;
  std::thread producer([&]() {
    RecordReader r1(in1_fn);
    if (!r1.in.fp) die("Could not open file " + in1_fn);
    std::unique_ptr<RecordReader> r2;
    if (paired) { r2.reset(new RecordReader(in2_fn)); if (!r2->in.fp) die("Could not open file " + in2_fn); }
    std::unique_ptr<Batch> b(new Batch());
    std::string name, name2;
    for (;;) {
      if (!r1.next(name, b->seqs)) break;
      b->off.push_back(b->seqs.size());
      if (paired) {
        if (!r2->next(name2, b->seqs)) die("File " + in1_fn + " contains more reads then file " + in2_fn);
        if (name != name2) die("Read names are not identical between the two input files. Probably reads are not in the same order in both files.");
      }
      b->off.push_back(b->seqs.size());
      b->names.push_back(name);
      if (b->n() >= batch_reads) { queue.push(std::move(b)); b.reset(new Batch()); }
    }
    if (paired && r2->next(name2, b->seqs))
      fprintf(stderr, "Warning: File %s has more reads then file %s\n", in2_fn.c_str(), in1_fn.c_str());
    b->last = true;
    queue.push(std::move(b));
  });

  std::vector<kaiju_gpu_hit> hits;
  std::vector<kaiju_result> res;
  std::string text;
  for (;;) {
    std::unique_ptr<Batch> b = queue.pop();
    const uint32_t n = (uint32_t)b->n();
    if (n) {
      hits.resize(n);
      res.resize(n);
      rc = kaiju_gpu_classify_batch(ctx, b->seqs.data(), b->off.data(), n, paired ? 1 : 0, hits.data());
      if (rc != 0) die(std::string("classification failed: ") + kaiju_gpu_strerror(rc) + " (" + kaiju_gpu_last_error() + ")");
      kaiju_finalize_hits(tax, &params, info.db_length, hits.data(), b->off.data(), n, paired ? 1 : 0, res.data());
      text.clear();
      char num[32];
      for (uint32_t r = 0; r < n; r++) {
        if (res[r].classified) {
          text += "C\t"; text += b->names[r]; text += '\t';
          snprintf(num, sizeof num, "%llu", (unsigned long long)res[r].taxon); text += num;
          if (verbose) {
            snprintf(num, sizeof num, "\t%u\t", res[r].best); text += num;
            uint64_t ids[KAIJU_GPU_MAX_IDS];
            const uint32_t k = hits[r].n_ids;
            for (uint32_t q = 0; q < k; q++) ids[q] = hits[r].taxid[q];
            std::sort(ids, ids + k);                       // std::set iteration order, :527-536
            for (uint32_t q = 0; q < k; q++) { snprintf(num, sizeof num, "%llu,", (unsigned long long)ids[q]); text += num; }
          }
          text += '\n';
        } else { text += "U\t"; text += b->names[r]; text += "\t0\n"; }
      }
      fwrite(text.data(), 1, text.size(), out);
    }
    if (b->last) break;
  }
  producer.join();
  if (verbose) fprintf(stderr, "%s Finished.\n", now().c_str());
  fflush(out);
  if (out != stdout) fclose(out);
  kaiju_gpu_destroy(ctx);
  kaiju_gpu_index_free(index);
  kaiju_taxonomy_free(tax);
  return EXIT_SUCCESS;
}
