/*
 * kaiju_gpu.h — C-ABI of the MI355X-native Kaiju classification path.
 *
 * This is the drop-in boundary for the body of the reference's
 * ConsumerThread::doWork() between queue->pop() and the output line
 * (/root/reference/src/ConsumerThread.cpp:632-743): six-frame translation,
 * SEG, MEM / Greedy backward search on the protein FM-index, locate and
 * taxon-id collection run in HIP kernels on gfx950; the caller keeps ingest,
 * the work queue, LCA and output formatting (helpers for the latter two are
 * exported here too so that a host needs nothing else).
 *
 * Plain C: opaque handles, plain pointers and sizes, no C++/torch types, no
 * exit() inside the library.  Every function returns 0 on success or a
 * negative kaiju_gpu_status; kaiju_gpu_strerror() maps it to text.
 *
 * Reference interfaces replaced / mirrored (file:line in /root/reference/src):
 *   kaiju_gpu_index_load        readFMI util.cpp:265-276 -> readIndexes bwt/bwt.c:78-88
 *   kaiju_gpu_index_from_host   the in-memory BWT/FMI/suffixArray structs, bwt/bwt.h:10-22,
 *                               bwt/fmi.h:9-18, bwt/suffixArray.h:10-33
 *   kaiju_gpu_params            Config fields, Config.hpp:33-48 (+ kaiju.cpp:77-80 for -a mem)
 *   kaiju_gpu_create/destroy    ConsumerThread::ConsumerThread ConsumerThread.cpp:6-187
 *   kaiju_gpu_classify_batch    ConsumerThread::doWork ConsumerThread.cpp:630-749, i.e.
 *                               getAllFragmentsBits :190-270, getNextFragment :272-342,
 *                               classify_length :543-628, classify_greedyblosum :424-541,
 *                               ids_from_SI :799-845 (everything up to, not including, lca_from_ids)
 *   kaiju_taxonomy_load         parseNodesDmp util.cpp:79-99
 *   kaiju_taxonomy_lca          lca_from_ids util.cpp:194-263
 *   kaiju_gpu_lca_batch_device  the same on the device (kaiju_gpu_taxonomy_upload: the tree in HBM)
 *   kaiju_finalize_hits         E-value gate ConsumerThread.cpp:500-513, LCA call :538,:625 and the
 *                               C/U decision :724-739
 */
#ifndef KAIJU_GPU_H
#define KAIJU_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KAIJU_GPU_ABI_VERSION 1
#define KAIJU_GPU_MAX_IDS 21   /* ids_from_SI stops once the set holds > max_match_ids (20) ids */

typedef enum {
  KAIJU_GPU_OK = 0,
  KAIJU_GPU_ERR_ARG = -1,        /* bad argument                                  */
  KAIJU_GPU_ERR_IO = -2,         /* file could not be opened / short read          */
  KAIJU_GPU_ERR_FORMAT = -3,     /* .fmi / nodes.dmp content not understood        */
  KAIJU_GPU_ERR_NO_DEVICE = -4,  /* no usable HIP device (the path has NO CPU fallback) */
  KAIJU_GPU_ERR_HIP = -5,        /* a HIP runtime call failed (see strerror)       */
  KAIJU_GPU_ERR_NOMEM = -6,
  KAIJU_GPU_ERR_UNSUPPORTED = -7,/* parameter outside what the kernels implement   */
  KAIJU_GPU_ERR_INDEX_BUG = -8   /* index hits one of the reference's latent bugs  */
} kaiju_gpu_status;

typedef struct kaiju_gpu_index kaiju_gpu_index;   /* FM-index resident in HBM (shareable by contexts) */
typedef struct kaiju_gpu_ctx kaiju_gpu_ctx;       /* one classification stream on one GPU             */
typedef struct kaiju_taxonomy kaiju_taxonomy;     /* host-side nodes.dmp tree                          */

/* mirrors Config (Config.hpp:33-48) */
typedef struct {
  int32_t mode;                 /* 0 = MEM, 1 = GREEDY                               */
  uint32_t min_fragment_length; /* -m, default 11                                    */
  uint32_t mismatches;          /* -e, default 3 (Greedy)                            */
  uint32_t min_score;           /* -s, default 65 (Greedy)                           */
  uint32_t seed_length;         /* -l, default 7 (Greedy)                            */
  int32_t seg;                  /* -x / -X, default on                               */
  int32_t use_evalue;           /* Greedy default on; "-a mem" turns it off          */
  int32_t input_is_protein;     /* -p (Config.hpp:42) and kaijup: reads are protein sequences; unpaired only */
  double min_evalue;            /* -E, default 0.01                                  */
  uint32_t max_matches_SI;      /* 20                                                */
  uint32_t max_match_ids;       /* 20                                                */
} kaiju_gpu_params;

/* what the device produces per read (everything before lca_from_ids) */
typedef struct {
  uint32_t best;                /* MEM: longest match length; GREEDY: best score; 0 = no match */
  uint32_t n_ids;               /* distinct taxon ids collected, <= KAIJU_GPU_MAX_IDS          */
  uint32_t flags;               /* KAIJU_HIT_* bits                                            */
  uint32_t reserved;
  uint64_t taxid[KAIJU_GPU_MAX_IDS]; /* in the reference's traversal (first seen) order       */
} kaiju_gpu_hit;

#define KAIJU_HIT_ID_CAP 1u     /* the 21-id cap ended the traversal (order sensitive case)   */
#define KAIJU_HIT_SI_CAP 2u     /* Greedy: > max_matches_SI equal-score matches existed       */
#define KAIJU_HIT_INEXACT 0x80000000u /* a device-side capacity bound was exceeded for this read (search scratch exhausted
                                   even in the retry pass; the region pool of the exact pass for fragments with more
                                   than 15 SEG regions exhausted): the record is NOT guaranteed to equal the
                                   reference's.  kaiju_gpu_stats.error_flags reports the batch-wide cases: 2 = SEG queue
                                   full, 4 = region pool of the exact pass exhausted, 8 = more than 65536 reads needed
                                   the exact pass                                                              */

/* what the host seam turns a hit into: one output line "C/U \t name \t taxon" */
typedef struct {
  uint64_t taxon;               /* LCA, 0 when unclassified            */
  uint32_t best;                /* column 4 of the -v output           */
  uint8_t classified;           /* 1 = 'C', 0 = 'U'                    */
  uint8_t pad[3];
} kaiju_result;

/* In-memory view of an index as the reference's loader holds it.  All pointers are
   borrowed for the duration of the call only. */
typedef struct {
  int64_t bwtlen;               /* FMI::bwtlen                                              */
  int32_t nseq;                 /* BWT::nseq                                                */
  int32_t alen;                 /* BWT::alen (21 for Kaiju's protein indexes)               */
  const char *alphabet;         /* BWT::alphabet, alen chars, [0] = terminator              */
  const uint8_t *bwt;           /* FMI::bwt, byte-coded (letter, in-block count)            */
  const int32_t *startLcode;    /* FMI::startLcode[alen+1]                                  */
  const uint8_t *sa;            /* suffixArray::sa, ncheck * nbytes big-endian entries      */
  int64_t ncheck;               /* suffixArray::ncheck                                      */
  int32_t chpt_exp, nbytes, pbits; /* suffixArray::chpt_exp / nbytes / pbits                */
  const char *const *ids;       /* suffixArray::ids[nseq], "accession_taxid"                */
} kaiju_gpu_host_index;

typedef struct {
  int64_t bwtlen;
  int32_t nseq, alen, chpt_exp;
  double db_length;             /* Config::db_length = len - nseq (Config.cpp:20)           */
  uint64_t device_bytes;        /* HBM held by the packed index                             */
  uint32_t warnings;            /* KAIJU_IDX_WARN_* bits                                    */
  char alphabet[64];
} kaiju_gpu_index_info;

#define KAIJU_IDX_WARN_SA_SHORT 1u   /* header ncheck one short (suffixArray.c:160 vs bwt.c:115) */
#define KAIJU_IDX_WARN_RANK_BUG 2u   /* bwtlen hits the fmi_chpt_value_with_dir corner case      */

/* work / timing of the last batch of a context */
typedef struct {
  uint64_t n_reads;
  uint64_t n_seg_fragments;      /* fragments that went through the SEG pass              */
  uint64_t n_overflow_retries;   /* reads redone in the retry pass (scratch overflow)     */
  uint64_t error_flags;          /* 0 unless a device-side capacity bound was violated    */
  /* HIP-event times on the stream of the last batch */
  double ms_translate, ms_seg, ms_search, ms_retry, ms_total;
} kaiju_gpu_stats;

int kaiju_gpu_abi_version(void);
const char *kaiju_gpu_strerror(int status);
/* last detailed message of the calling thread (HIP error strings etc.) */
const char *kaiju_gpu_last_error(void);
int kaiju_gpu_device_count(void);

/* ---- index ---------------------------------------------------------- */
int kaiju_gpu_index_load(const char *fmi_path, int device_id, kaiju_gpu_index **out);
int kaiju_gpu_index_from_host(const kaiju_gpu_host_index *view, int device_id, kaiju_gpu_index **out);
/* kaijux / kaijup semantics (ConsumerThreadx.cpp:261-287, README "KaijuX and KaijuP"): a hit collects the
   database SEQUENCES it matches instead of their taxa.  With KAIJU_GPU_IDS_SEQUENCE the ids in kaiju_gpu_hit are
   sequence numbers (kaiju_gpu_index_seq_name() gives the names), first-seen order, capped like taxon ids. */
enum { KAIJU_GPU_IDS_TAXON = 0, KAIJU_GPU_IDS_SEQUENCE = 1 };
int kaiju_gpu_index_load_ex(const char *fmi_or_image_path, int device_id, int id_mode, kaiju_gpu_index **out);
/* The same index on several GPUs of the node (north star: "index replicated per GPU"): parsed and packed once, uploaded to
   devices[0 .. n_devices-1] in turn; out[k] is the replica on devices[k] (free each with kaiju_gpu_index_free). */
int kaiju_gpu_index_load_devices(const char *fmi_or_image_path, const int *devices, int n_devices, int id_mode,
                                 kaiju_gpu_index **out);

/* Device image of an index: the arrays of the HBM layout, packed once on the host and
   written to a file; kaiju_gpu_index_load() recognises such a file by its magic and uploads it without
   parsing or packing anything.  Needs no GPU. */
int kaiju_gpu_index_write_image(const char *fmi_path, const char *image_path);
/* Size in bytes of the .fmi an image was made from (its header remembers it), so that a caller can tell an image of ANOTHER
   index that merely has a newer time stamp (cp -p, rsync -t, restored backups) from a usable one.  Returns 0 or a negative
   status (not an image of this version: KAIJU_GPU_ERR_FORMAT). */
int kaiju_gpu_index_image_source_bytes(const char *image_path, uint64_t *fmi_bytes);

/* Host only: header fields of an image and the number of bytes a load streams from it to the device.  A load of an image
   reads its small arrays (taxon ids, names, count bases) into host memory and moves the arrays that grow with the index - rank
   blocks, sampled suffix-array entries, terminator rows, the k-mer table - from the file to HBM in page-locked pieces
   (KAIJU_GPU_STREAM_PIECE_MB, default 256), two in flight: no host copy of the index exists at any time (the reference maps
   the whole .fmi into every process, readIndexes bwt/bwt.c:78-88).  info->device_bytes = bytes of the packed arrays. */
int kaiju_gpu_index_image_info(const char *image_path, kaiju_gpu_index_info *info, uint64_t *streamed_bytes);

int kaiju_gpu_index_get_info(const kaiju_gpu_index *ix, kaiju_gpu_index_info *info);
/* Diagnostics: a position-sensitive 64-bit digest of every array the index holds in HBM (computed by a kernel, nothing is
   copied back), so that two ways of loading one index - packed on the host, streamed from an image, streamed from the .fmi
   and packed on the device (KAIJU_GPU_FMI_STREAM) - can be compared array by array at any size.  out[]: [0] rank blocks,
   [1] count bases, [2] sampled sequence numbers, [3] taxon ids of the samples, [4] terminator rows, [5] taxon id and [6]
   validity per sequence, [7] k-mer table, [8] k-mer lines, [9] text, [10] full suffix array / 40-bit text positions,
   [11] sequence of every row, [12] k (letters of the k-mer table), [13] C[] ; 0 for an array the index does not have. */
#define KAIJU_GPU_N_DIGESTS 14
int kaiju_gpu_index_digest(const kaiju_gpu_index *ix, uint64_t *out, uint32_t n_out);

/* What the index occupies in HBM, array by array (bytes).  Per index row: rank blocks 2 B; suffix-array sample at exponent e:
   4 / 2^e B of sequence numbers and - indexes below 2^32 rows only - 8 / 2^e B of taxon ids; the k-mer table and its lines
   do not grow with the index (20^k entries).  A refseq-class index (58 G rows, e = 3) is 2.5 B per row = 145 GB + tables. */
typedef struct kaiju_gpu_index_footprint {
  uint64_t rank_blocks;   /* RankBlock64 lines: five bit planes + twenty counts per 64 rows                                */
  uint64_t count_bases;   /* wide layout: the 64-bit counts at the start of every 2^31 rows                               */
  uint64_t sa_seq;        /* sequence number of every sampled suffix-array row                                            */
  uint64_t sa_taxid;      /* taxon id of every sampled row (narrow indexes)                                               */
  uint64_t seq_tables;    /* per database sequence: taxon id, validity, row of its terminator                             */
  uint64_t kmer_table;    /* suffix interval of every k-letter word                                                       */
  uint64_t kmer_lines;    /* the same as 128-byte lines for two end positions each (narrow indexes)                       */
  uint64_t text;          /* the database text (indexes that leave room for it: text verification of long matches)       */
  uint64_t sa_full;       /* ... and position + taxon of every row's suffix, 2 x 4 bytes per row (narrow indexes), or     */
                          /* the 40-bit text position of every 2^s-th row (wide indexes: s = 0 .. 3 by the room left)     */
                          /* + the taxon of every row, 4 bytes each (wide indexes that leave room: KAIJU_GPU_ROW_TAX)     */
  uint64_t other;         /* constant tables                                                                              */
  uint64_t total;
  uint32_t kmer_k, wide;  /* k of the k-mer lines (narrow: a five-letter table stays next to them) or of the table; 1 = 64-bit positions */
} kaiju_gpu_index_footprint;
int kaiju_gpu_index_get_footprint(const kaiju_gpu_index *ix, kaiju_gpu_index_footprint *out);
void kaiju_gpu_index_free(kaiju_gpu_index *ix);

/* ---- classification -------------------------------------------------- */
void kaiju_gpu_default_params(kaiju_gpu_params *p, int mode);
int kaiju_gpu_create(kaiju_gpu_ctx **out, const kaiju_gpu_index *ix, const kaiju_gpu_params *p);
void kaiju_gpu_destroy(kaiju_gpu_ctx *ctx);

/* Host buffers.  seqs: concatenated, already strip()'d ASCII nucleotides (protein letters, either case, when the
   context was created with input_is_protein; such reads have no mate and paired must be 0);
   off[2*n+1]: read r is seqs[off[2r], off[2r+1]) and its mate seqs[off[2r+1], off[2r+2])
   (empty when unpaired).  paired selects the length gate of ConsumerThread.cpp:647-654.
   Blocks until out[0..n) is filled. */
int kaiju_gpu_classify_batch(kaiju_gpu_ctx *ctx, const char *seqs, const uint64_t *off,
                             uint32_t n_reads, int paired, kaiju_gpu_hit *out);

/* Device-resident variant: all pointers are HIP device pointers on the context's
   GPU, work is enqueued on `stream` (a hipStream_t; NULL = the context's own
   stream) and the call returns without synchronising. */
int kaiju_gpu_classify_batch_device(kaiju_gpu_ctx *ctx, const void *d_seqs, uint64_t seq_bytes,
                                    const uint64_t *d_off, uint32_t n_reads, int paired,
                                    kaiju_gpu_hit *d_out, void *stream);
/* Upper bound of the read (mate) lengths of the batches handed to classify_batch_device; it sizes
   per-lane scratch and selects the LDS-staged translation kernel for short reads.  The host-buffer
   entry point measures it itself.  Default 1024. */
int kaiju_gpu_set_max_read_length(kaiju_gpu_ctx *ctx, uint32_t max_read_len);
int kaiju_gpu_synchronize(kaiju_gpu_ctx *ctx);
/* the context's own HIP stream (a hipStream_t): what the device-resident entry points use when they are given NULL; a
   caller that queues its own work behind a batch (a copy, a collective) orders it on this stream */
int kaiju_gpu_get_stream(kaiju_gpu_ctx *ctx, void **stream);
int kaiju_gpu_get_stats(kaiju_gpu_ctx *ctx, kaiju_gpu_stats *stats);
/* Accounting (bench.py's roofline, never a timed launch): with count_ops on, the main search pass of the following
   batches runs the counting instantiation of its lane, which adds up the memory steps of the search as THIS
   implementation performs them; kaiju_gpu_get_op_counts() returns the totals of the last batch:
   [0] k-mer table lookups, [1] UpdateSI steps past the table (bwt.c:160-173), [2] distinct 128-byte rank lines those
   steps (and the Greedy multi-letter steps) touched, [3] LF steps (compactfmi.c:312-336), [4] rank lines of those,
   [5] SA samples read, [6] read descriptors, [7] fragment descriptors, [8] 64-byte peptide windows, [9] terminator
   searches, [10] match records spilled (MEM), [11] hit records written, [12] Greedy multi-letter steps
   (ConsumerThread.cpp:346-395 at one position), [13] queue items read, [14] match records read, [15] queue items
   written, [16] match records written, [17] wave iterations, [18] lane iterations, [19] bytes of the state / queue / task
   records the third-generation Greedy kernels read and write. */
#define KAIJU_GPU_N_OP_COUNTS 20
int kaiju_gpu_set_count_ops(kaiju_gpu_ctx *ctx, int on);
int kaiju_gpu_get_op_counts(kaiju_gpu_ctx *ctx, uint64_t *out, uint32_t n_out);

/* ---- host side of the seam ------------------------------------------- */
int kaiju_taxonomy_load(const char *nodes_dmp_path, kaiju_taxonomy **out);
void kaiju_taxonomy_free(kaiju_taxonomy *t);
/* ids need not be sorted; returns 0 if none of them is in the tree */
uint64_t kaiju_taxonomy_lca(kaiju_taxonomy *t, const uint64_t *ids, uint32_t n);
/* E-value gate + LCA + C/U decision for a batch.  len1/len2 are the nucleotide
   lengths (query_len = len1/3.0 [+ len2/3.0], ConsumerThread.cpp:698,704); pass
   off as given to classify_batch.  db_length from kaiju_gpu_index_get_info. */
/* ---- verbose output: columns 6 and 7 of kaiju -v ------------------------- */
/* ConsumerThread.cpp:527-536 (Greedy), :614-623 (MEM), :820-824 (accessions).  Per read: the
   sequences whose names give the accession set of column 6 (first 20 distinct ones in the order the
   reference visits the rows; the caller prints the sorted set of kaiju_gpu_index_seq_name() prefixes
   up to the last '_') and the text of column 7 ("PEPTIDE,PEPTIDE,").  The search kernels of the
   default path in instantiations that note where every match lies in its read (MEM: k_mem_vb /
   k_mem_wide2_vb) or keep, per best match, its place and the substitutions of its variant (Greedy:
   k_greedy2_vb / k_greedy2_wide_vb), and a pass over the records in front of the locate
   (k_mem_verbose).  Reads that take the retry pass or the exact pass: first-generation kernels. */
#define KAIJU_GPU_MAX_ACC 20
typedef struct {
  uint32_t n_acc;
  uint32_t text_len;                      /* bytes at text + r * text_stride, 0-terminated (packed: at *text + text_pos[r]) */
  uint32_t truncated;                     /* the peptides did not fit */
  uint32_t acc_iseq[KAIJU_GPU_MAX_ACC];
} kaiju_gpu_verbose;
int kaiju_gpu_classify_batch_verbose(kaiju_gpu_ctx *ctx, const char *seqs, const uint64_t *off, uint32_t n_reads,
                                     int paired, kaiju_gpu_hit *out, kaiju_gpu_verbose *vout, char *text,
                                     uint32_t text_stride);
/* The same with column 7 of the whole batch as ONE string owned by the context: read r's text is
   (*text)[text_pos[r] .. text_pos[r] + vout[r].text_len), not terminated; *text_bytes = the length of
   the string.  *text stays valid until the next verbose call on this context (or its destruction).
   No row of text_stride bytes per read to allocate and walk: what the command line programs call. */
int kaiju_gpu_classify_batch_verbose_packed(kaiju_gpu_ctx *ctx, const char *seqs, const uint64_t *off, uint32_t n_reads,
                                            int paired, kaiju_gpu_hit *out, kaiju_gpu_verbose *vout, uint64_t *text_pos,
                                            const char **text, uint64_t *text_bytes);
/* text_stride that holds column 7 of every read of a batch whose longest read (both mates of a pair together) has
   max_pair_len letters: callers size `text` with it instead of guessing the library's bound */
uint32_t kaiju_gpu_verbose_text_stride(uint32_t max_pair_len, int input_is_protein);
/* name of database sequence iseq as stored in the .fmi ("accession_taxid"), NULL if out of range */
const char *kaiju_gpu_index_seq_name(const kaiju_gpu_index *index, uint32_t iseq);

/* ---- LCA on the device: 16-byte records instead of 184-byte ones ------- */
/* (what crosses PCIe / xGMI when the matched ids themselves are not needed, i.e. without -v;
   SURVEY.md 8f-3.  lca_from_ids util.cpp:194-263 on a device copy of the tree.) */
typedef struct kaiju_gpu_taxonomy kaiju_gpu_taxonomy;
typedef struct {
  uint64_t lca;        /* LCA of the ids of the hit; 0 = no hit, or none of its ids is in nodes.dmp */
  uint32_t best;       /* as kaiju_gpu_hit.best                                                    */
  uint32_t info;       /* kaiju_gpu_hit.flags << 8 | n_ids; bit 31 = KAIJU_HIT_INEXACT               */
} kaiju_gpu_compact;
int kaiju_gpu_taxonomy_upload(const kaiju_taxonomy *t, int device_id, kaiju_gpu_taxonomy **out);
void kaiju_gpu_taxonomy_free(kaiju_gpu_taxonomy *t);
/* d_hits: n records written by kaiju_gpu_classify_batch_device; asynchronous on stream (NULL = the context's own
   stream, i.e. behind a classify_batch_device call that was also given NULL) */
int kaiju_gpu_lca_batch_device(kaiju_gpu_ctx *ctx, const kaiju_gpu_taxonomy *t, const kaiju_gpu_hit *d_hits,
                               uint32_t n_reads, kaiju_gpu_compact *d_out, void *stream);
/* kaiju_gpu_classify_batch_device + kaiju_gpu_lca_batch_device in ONE call (round 6): where the configuration allows - MEM on an
   index below 2^32 rows with its row -> taxon table - the search's post-search pass writes the 16-byte records itself instead
   of a pass of its own over the 184-byte ones; everywhere else this is the two calls.  d_hits (n records) is the search's working
   set here, not an output: behind the call best / n_ids / flags and the first n_ids ids of every record are what
   kaiju_gpu_classify_batch_device writes, the id slots behind them are unspecified (not zeroed: 1.84 GB less to write per
   10 M reads).  What the reference does at this point: ids_from_SI + lca_from_ids per read in its
   ConsumerThread (src/ConsumerThread.cpp:591-612, src/util.cpp:194-263) */
int kaiju_gpu_classify_batch_device_compact(kaiju_gpu_ctx *ctx, const kaiju_gpu_taxonomy *t, const void *d_seqs, uint64_t seq_bytes,
                                            const uint64_t *d_off, uint32_t n_reads, int paired,
                                            kaiju_gpu_hit *d_hits, kaiju_gpu_compact *d_out, void *stream);
/* kaiju_gpu_classify_batch followed by the LCA on the device: host buffers in, 16-byte records out */
int kaiju_gpu_classify_batch_compact(kaiju_gpu_ctx *ctx, const kaiju_gpu_taxonomy *t, const char *seqs,
                                     const uint64_t *off, uint32_t n_reads, int paired, kaiju_gpu_compact *out);
/* LCA of host hit records through the device (blocking) */
int kaiju_gpu_lca_batch(kaiju_gpu_ctx *ctx, const kaiju_gpu_taxonomy *t, const kaiju_gpu_hit *hits,
                        uint32_t n_reads, kaiju_gpu_compact *out);
/* kaiju_finalize_hits for compact records (E-value gate, C/U decision; the LCA is already in them) */
int kaiju_finalize_compact(const kaiju_gpu_params *p, double db_length, const kaiju_gpu_compact *recs,
                           const uint64_t *off, uint32_t n_reads, int paired, kaiju_result *out);

/* ---- several processes of a node, one per GPU: the gather --------------- */
/* BASELINE north star: "reads shard embarrassingly across the GPUs of one node with the index replicated per GPU and per-GPU
   hit lists gathered with a single RCCL gather over xGMI".  The reference has no exchange (its threads append to one output
   stream under a mutex, ConsumerThread.cpp:847-856; the thread fan-out is kaiju.cpp:250-257); here every rank classifies its
   shard, the device LCA turns the hits into 16-byte records and ONE collective per batch brings them to the rank that
   writes the output.  kaiju_gpu_comm_create: rank 0 makes the communicator's id and hands it over through `rendezvous_path`
   (any path all ranks of the job see and may write next to, e.g. under /dev/shm; up to two minutes for everybody to arrive).
   The path need not be fresh: rank r > 0 announces itself with a random nonce in `<path>.r<r>`, takes the id only from a file
   that carries that nonce and removes its nonce file; rank 0 removes whatever lay at `<path>` before, and the file itself once
   every rank has acknowledged - a file left by an earlier job is never taken for this job's id.  Two jobs must not use ONE
   path at the same time.  librccl is opened on the first call (dlopen): nothing links it, a single-GPU run never loads it.
   kaiju_gpu_gather_compact: n records of EVERY rank (the same n everywhere) into d_recv on `root`, rank r's at d_recv + r * n;
   d_recv is ignored elsewhere.  Asynchronous on `stream` (a hipStream_t; kaiju_gpu_get_stream() of the context that wrote
   d_send orders it behind the batch).  Without a HIP device: KAIJU_GPU_ERR_NO_DEVICE. */
typedef struct kaiju_gpu_comm kaiju_gpu_comm;
int kaiju_gpu_comm_create(const char *rendezvous_path, int rank, int world, int device_id, kaiju_gpu_comm **out);
void kaiju_gpu_comm_destroy(kaiju_gpu_comm *comm);
int kaiju_gpu_comm_rank(const kaiju_gpu_comm *comm);
int kaiju_gpu_comm_world(const kaiju_gpu_comm *comm);
int kaiju_gpu_gather_compact(kaiju_gpu_comm *comm, const kaiju_gpu_compact *d_send, uint32_t n, kaiju_gpu_compact *d_recv,
                             int root, void *stream);
const char *kaiju_gpu_comm_last_error(void);
/* the librccl the gather runs on ("" before the first communicator): the one that lies next to the HIP runtime this process
   uses - a Python host may hold two ROCm stacks, the system's and the one bundled with torch */
const char *kaiju_gpu_comm_library(void);

int kaiju_finalize_hits(kaiju_taxonomy *t, const kaiju_gpu_params *p, double db_length,
                        const kaiju_gpu_hit *hits, const uint64_t *off, uint32_t n_reads,
                        int paired, kaiju_result *out);

#ifdef __cplusplus
}
#endif
#endif /* KAIJU_GPU_H */
