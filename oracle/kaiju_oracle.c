/*
 * kaiju_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE (see kaiju_oracle.h).
 *
 * Plain-C restatement of the reference's per-read classification path.  It
 * follows the reference's *sequential* control flow (lazy SEG inside the
 * fragment queue, dynamic pruning bound, byte-coded BWT scan, linked SI
 * lists) so that it can be checked function by function against the compiled
 * reference in oracle/_ref.  The HIP product path uses a different, parallel
 * formulation; the tests compare the two.
 *
 * Citations (file:line) are into /root/reference/src.
 */
#define _GNU_SOURCE
#include "kaiju_oracle.h"

#include <ctype.h>
#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* counters                                                            */
/* ------------------------------------------------------------------ */
static ko_counters g_cnt;
void ko_counters_reset(void) { memset(&g_cnt, 0, sizeof g_cnt); }
void ko_counters_get(ko_counters *c) { *c = g_cnt; }

/* ------------------------------------------------------------------ */
/* constant tables (ConsumerThread.cpp:6-187)                          */
/* ------------------------------------------------------------------ */

/* BLOSUM order used by the reference's aa2int (ConsumerThread.cpp:40-60) */
static const char AA_ORDER[21] = "ARNDCQEGHILKMFPSTWYV";

/* BLOSUM62, rows/cols in AA_ORDER.  Diagonal = blosum62diag (ConsumerThread.cpp:61-80),
   off-diagonal = b62 (ConsumerThread.cpp:83-102). */
static const int8_t B62[20][20] = {
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

/* standard genetic code, index = n0*16 + n1*4 + n2 with A,C,G,T/U = 0..3
   (codon2aa, ConsumerThread.cpp:111-177; every other index is '*') */
static const char CODON_AA[65] =
  "KNKNTTTTRSRSIIMIQHQHPPPPRRRRLLLLEDEDAAAAGGGGVVVV*Y*YSSSS*CWCLFLF";

static uint8_t nuc2int[256], compnuc2int[256];
static uint8_t aa2int[256];
static char codon2aa[256];
static char subst[20][19];        /* blosum_subst, ConsumerThread.cpp:10-30 */
static int tables_ready = 0;

static void init_tables(void) {
  if (tables_ready) return;
  memset(nuc2int, 255, sizeof nuc2int);
  memset(compnuc2int, 255, sizeof compnuc2int);
  const char *nt = "ACGT";
  for (int i = 0; i < 4; i++) {
    nuc2int[(uint8_t)nt[i]] = nuc2int[(uint8_t)tolower(nt[i])] = (uint8_t)i;
    compnuc2int[(uint8_t)nt[i]] = compnuc2int[(uint8_t)tolower(nt[i])] = (uint8_t)(3 - i);
  }
  nuc2int['U'] = nuc2int['u'] = 3;
  compnuc2int['U'] = compnuc2int['u'] = 0;
  memset(aa2int, 0, sizeof aa2int);
  for (int i = 0; i < 20; i++) aa2int[(uint8_t)AA_ORDER[i]] = (uint8_t)i;
  memset(codon2aa, '*', sizeof codon2aa);
  for (int i = 0; i < 64; i++) codon2aa[i] = CODON_AA[i];
  /* blosum_subst: the 19 other residues by descending BLOSUM62 score; equal
     scores appear in descending aa2int order (checked against the literal
     lists of ConsumerThread.cpp:10-30 by tests/test_oracle_tables.py). */
  for (int a = 0; a < 20; a++) {
    int n = 0;
    for (int s = 11; s >= -4; s--)
      for (int b = 19; b >= 0; b--)
        if (b != a && B62[a][b] == s) subst[a][n++] = AA_ORDER[b];
  }
  tables_ready = 1;
}

/* ------------------------------------------------------------------ */
/* index                                                               */
/* ------------------------------------------------------------------ */
struct ko_index {
  /* BWT header (bwt.c:51-61) */
  int64_t len;
  int32_t nseq, alen;
  char alphabet[64];
  /* suffix array (suffixArray.c:282-321) */
  int64_t salen, ncheck;
  int32_t chpt_exp, nbytes, sbits, pbits;
  int64_t mask, check;
  int32_t sa_nseq;
  char **ids;
  int32_t *seqTermOrder;
  int64_t *seqlengths;
  uint8_t *sa;
  /* FMI (fmicommon.h:190-217, compactfmi.c:165-171) */
  int32_t f_alen;
  int64_t bwtlen;
  int32_t N1, N2;
  uint8_t *bwt;
  int64_t *index1;   /* [N1][alen] */
  uint16_t *index2;  /* [N2][alen] */
  int32_t startLcode[65];
  uint8_t lcode[256], ncode[256];
  /* translate2numbers table (sequence.c:68-97) */
  signed char trans[128];
};

static int rd(void *dst, size_t sz, size_t n, FILE *fp) { return fread(dst, sz, n, fp) == n; }

ko_index *ko_load_fmi(const char *path) {
  init_tables();
  FILE *fp = fopen(path, "rb");
  if (!fp) return NULL;
  ko_index *ix = (ko_index *)calloc(1, sizeof *ix);
  int ok = 1;
  ok &= rd(&ix->len, 8, 1, fp);
  ok &= rd(&ix->nseq, 4, 1, fp);
  ok &= rd(&ix->alen, 4, 1, fp);
  if (!ok || ix->alen <= 0 || ix->alen > 60) { fclose(fp); free(ix); return NULL; }
  ok &= rd(ix->alphabet, 1, (size_t)ix->alen, fp);
  ok &= rd(&ix->salen, 8, 1, fp);
  ok &= rd(&ix->ncheck, 8, 1, fp);
  ok &= rd(&ix->chpt_exp, 4, 1, fp);
  ok &= rd(&ix->nbytes, 4, 1, fp);
  ok &= rd(&ix->sbits, 4, 1, fp);
  ok &= rd(&ix->pbits, 4, 1, fp);
  ok &= rd(&ix->mask, 8, 1, fp);
  ok &= rd(&ix->check, 8, 1, fp);
  ok &= rd(&ix->sa_nseq, 4, 1, fp);
  if (!ok) { fclose(fp); free(ix); return NULL; }
  ix->ids = (char **)calloc((size_t)ix->sa_nseq, sizeof(char *));
  for (int i = 0; i < ix->sa_nseq && ok; i++) {
    uint8_t l;
    ok &= rd(&l, 1, 1, fp);
    ix->ids[i] = (char *)malloc((size_t)l + 1);
    ok &= rd(ix->ids[i], 1, l, fp);
    ix->ids[i][l] = 0;
  }
  ix->seqTermOrder = (int32_t *)malloc(sizeof(int32_t) * (size_t)ix->sa_nseq);
  ix->seqlengths = (int64_t *)malloc(sizeof(int64_t) * (size_t)ix->sa_nseq);
  ok &= rd(ix->seqTermOrder, 4, (size_t)ix->sa_nseq, fp);
  ok &= rd(ix->seqlengths, 8, (size_t)ix->sa_nseq, fp);
  size_t sabytes = (size_t)ix->ncheck * (size_t)ix->nbytes;
  ix->sa = (uint8_t *)malloc(sabytes + 16);
  ok &= rd(ix->sa, 1, sabytes, fp);
  ok &= rd(&ix->f_alen, 4, 1, fp);
  ok &= rd(&ix->bwtlen, 8, 1, fp);
  ok &= rd(&ix->N1, 4, 1, fp);
  ok &= rd(&ix->N2, 4, 1, fp);
  if (!ok) { fclose(fp); return NULL; }
  ix->bwt = (uint8_t *)malloc((size_t)ix->bwtlen + 16);
  ok &= rd(ix->bwt, 1, (size_t)ix->bwtlen, fp);
  ix->index1 = (int64_t *)malloc(sizeof(int64_t) * (size_t)ix->N1 * (size_t)ix->f_alen);
  ok &= rd(ix->index1, 8, (size_t)ix->N1 * (size_t)ix->f_alen, fp);
  ix->index2 = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)ix->N2 * (size_t)ix->f_alen);
  ok &= rd(ix->index2, 2, (size_t)ix->N2 * (size_t)ix->f_alen, fp);
  ok &= rd(ix->startLcode, 4, (size_t)ix->f_alen + 1, fp);
  fclose(fp);
  if (!ok) return NULL;
  /* fmi_fill_codes, compactfmi.c:75-89 */
  for (int a = 0; a < ix->f_alen; a++) {
    int n = 0, k;
    for (k = ix->startLcode[a]; k < ix->startLcode[a + 1] - 1; k++) {
      ix->lcode[k] = (uint8_t)a;
      ix->ncode[k] = (uint8_t)n++;
    }
    ix->lcode[k] = (uint8_t)a;
    ix->ncode[k] = 255;
  }
  /* translation_table(alphabet, NULL, dummy = len-1, case-insensitive), sequence.c:68-97,132-141 */
  int l = ix->alen;
  ix->trans[0] = 0;
  for (int i = 1; i < 128; i++) ix->trans[i] = isalpha(i) ? (signed char)(l - 1) : (signed char)-1;
  for (int i = 0; i < l; i++) {
    ix->trans[toupper((unsigned char)ix->alphabet[i])] = (signed char)i;
    ix->trans[tolower((unsigned char)ix->alphabet[i])] = (signed char)i;
  }
  return ix;
}

void ko_free_index(ko_index *ix) {
  if (!ix) return;
  for (int i = 0; i < ix->sa_nseq; i++) free(ix->ids[i]);
  free(ix->ids); free(ix->seqTermOrder); free(ix->seqlengths); free(ix->sa);
  free(ix->bwt); free(ix->index1); free(ix->index2); free(ix);
}
int64_t ko_bwtlen(const ko_index *ix) { return ix->bwtlen; }
int32_t ko_nseq(const ko_index *ix) { return ix->nseq; }
int32_t ko_alen(const ko_index *ix) { return ix->alen; }
const char *ko_alphabet(const ko_index *ix) { return ix->alphabet; }
const char *ko_seq_name(const ko_index *ix, int32_t iseq) { return ix->ids[iseq]; }

/* ids_from_SI's name -> taxon id rule (ConsumerThread.cpp:809-833) */
uint64_t ko_seq_taxid(const ko_index *ix, int32_t iseq, int *ok) {
  const char *name = ix->ids[iseq];
  const char *pch = strrchr(name, '_');
  unsigned long id = strtoul(pch ? pch + 1 : name, NULL, 10);
  *ok = (id != ULONG_MAX);
  return (uint64_t)id;
}

/* ------------------------------------------------------------------ */
/* rank: FMindex / FMindexCurrent over the byte-coded BWT              */
/* ------------------------------------------------------------------ */
#define EX1 16
#define EX2 8

/* fmi_chpt_value_with_dir, fmicommon.h:59-73 */
static inline int64_t chpt_value(const ko_index *f, int64_t k, int c, int dir) {
  int64_t chpt2 = k >> EX2;
  if (dir > 0) chpt2 += 1;
  int64_t chpt1 = chpt2 >> (EX1 - EX2);
  return f->index1[chpt1 * f->f_alen + c] + f->index2[chpt2 * f->f_alen + c];
}

/* fmi_bwt2number, compactfmi.c:201-214: pos is AT a byte of letter c */
static inline int bwt2number(const ko_index *f, int c, int64_t pos, int dir) {
  int n, k = 0;
  while ((n = f->ncode[f->bwt[pos]]) == 255) {
    k += 1;
    pos += dir;
    while (f->lcode[f->bwt[pos]] != c) pos += dir;
  }
  return dir < 0 ? n + k : -n - k - 1;
}

/* FMindex, compactfmi.c:267-307 */
int64_t ko_fmindex(ko_index *f, int ct, int64_t k) {
  g_cnt.fmindex++;
  int64_t pos = k;           /* "bwt" pointer as an index; -1 plays NULL */
  int c = (k < f->bwtlen) ? f->lcode[f->bwt[k]] : 255;
  int dir = (k & ((int64_t)1 << (EX2 - 1))) ? 1 : -1;   /* fmi_direction, fmicommon.h:48-52 */
  int64_t fmi = chpt_value(f, k, ct, dir);
  int64_t delta = 0;
  if (c != ct) {
    int64_t stop = k & ~(((int64_t)1 << EX2) - 1);
    if (pos == stop) pos = -1;
    else {
      if (dir > 0) {
        stop += (int64_t)1 << EX2;
        if (stop >= f->bwtlen) {
          stop = f->bwtlen;
          if (k >= f->bwtlen) pos = stop - 1;
        }
        stop -= 1;
      } else delta = 1;
      if (pos == stop) pos = -1;
      else {
        /* find_closest_letter_with_bound, compactfmi.c:249-256 */
        int64_t p = pos + dir;
        for (;;) {
          g_cnt.bwt_scanned++;
          if (f->lcode[f->bwt[p]] == ct) { pos = p; break; }
          if (p == stop) { pos = -1; break; }
          p += dir;
        }
      }
    }
  }
  if (pos >= 0) fmi += delta + bwt2number(f, ct, pos, dir);
  return fmi;
}

/* FMindexCurrent / FMindexHere, compactfmi.c:312-336 */
int64_t ko_fmindex_current(ko_index *f, int64_t k, int *c) {
  g_cnt.fmindex_current++;
  *c = f->lcode[f->bwt[k]];
  int dir = (k & ((int64_t)1 << (EX2 - 1))) ? 1 : -1;
  int n = bwt2number(f, *c, k, dir);
  return n + chpt_value(f, k, *c, dir);
}

/* InitialSI, bwt.c:146-152 */
void ko_initial_si(ko_index *f, int ct, int64_t si[2]) {
  g_cnt.initial_si++;
  int64_t r = f->N1 - 1;
  si[0] = f->index1[r * f->f_alen + ct];
  if (ct < f->f_alen - 1) si[1] = f->index1[r * f->f_alen + ct + 1];
  else si[1] = f->bwtlen;
}

/* UpdateSI, bwt.c:160-173 (out may alias si) */
int64_t ko_update_si(ko_index *f, int ct, const int64_t si[2], int64_t out[2]) {
  g_cnt.update_si++;
  int64_t a = ko_fmindex(f, ct, si[0]);
  int64_t b = ko_fmindex(f, ct, si[1]);
  if (a >= b) return 0;
  out[0] = a; out[1] = b;
  return b - a;
}

/* uchar2long + suffixArray_decode_number, suffixArray.h:37-51 */
static inline void sa_decode(const ko_index *s, int64_t k, int32_t *iseq, int64_t *pos) {
  g_cnt.sa_decode++;
  const uint8_t *c = s->sa + k * s->nbytes;
  int n = s->nbytes;
  int64_t val = *c++;
  while (--n > 0) val = (val << 8) + *c++;
  *iseq = (int32_t)(val >> s->pbits);
  *pos = val & s->mask;
}

/* get_suffix, bwt.c:105-121 */
void ko_get_suffix(ko_index *f, int64_t i, int32_t *iseq, int64_t *pos) {
  g_cnt.get_suffix++;
  int64_t k = 0;
  int c = 1;
  while (c && (i & f->check)) {
    i = ko_fmindex_current(f, i, &c);
    ++k;
  }
  if (c) {
    sa_decode(f, (i >> f->chpt_exp) - ((f->sa_nseq - 1) >> f->chpt_exp) - 1, iseq, pos);
    *pos += k;
  } else { *iseq = (int32_t)i; *pos = k - 1; }
}

/* ------------------------------------------------------------------ */
/* SI lists (bwt.h:25-34, bwt.c:178-252)                               */
/* ------------------------------------------------------------------ */
typedef struct SI {
  int64_t start;
  int len, qi, ql;
  int count;                 /* rows in this node, its samelen chain and everything behind it (max_matches bookkeeping) */
  struct SI *next, *samelen;
} SI;

static SI *alloc_SI(const int64_t si[2], int qi, int ql) {
  SI *r = (SI *)malloc(sizeof(SI));
  r->start = si[0]; r->len = (int)(si[1] - si[0]);
  r->qi = qi; r->ql = ql; r->count = 0; r->next = NULL; r->samelen = NULL;
  return r;
}
static void free_SI_rec(SI *si) {
  if (!si) return;
  free_SI_rec(si->next); free_SI_rec(si->samelen); free(si);
}

/* insert_SI_sorted, bwt.c:225-252 (the ->count bookkeeping is unused when max_matches==0) */
static SI *insert_SI_sorted(SI *base, SI *nw) {
  if (!base) return nw;
  if (base->ql < nw->ql) { nw->next = base; return nw; }
  SI *tmp = base;
  while (tmp->next && tmp->next->ql >= nw->ql) tmp = tmp->next;
  if (tmp->ql == nw->ql) { nw->samelen = tmp->samelen; tmp->samelen = nw; }
  else { nw->next = tmp->next; tmp->next = nw; }
  return base;
}

/* backward extension shared by the three search routines: starting with interval
   si covering str[i+1..j], extend left while possible; returns start of match */
static int extend_left(ko_index *f, const uint8_t *str, int i, int64_t si[2]) {
  while (i-- > 0) {
    if (ko_update_si(f, str[i], si, si) == 0) break;
  }
  return i + 1;
}

/* greedyExact, bwt.c:347-380 (jump < 0 => delta = 1) */
static SI *greedyExact(ko_index *f, const uint8_t *str, int len, int L) {
  SI *first = NULL;
  int64_t si[2];
  for (int j = len - 1; j >= L - 1; j--) {
    ko_initial_si(f, str[j], si);
    int i = extend_left(f, str, j, si);
    int l = j - i + 1;
    if (l >= L) {
      if (l > L) { free_SI_rec(first); first = NULL; L = l; }
      SI *cur = first;
      first = alloc_SI(si, i, l);
      first->samelen = cur;
    }
    if (i <= 1) break;
  }
  return first;
}

/* maxMatches with max_matches == 0, bwt.c:261-296 */
static SI *maxMatches(ko_index *f, const uint8_t *str, int len, int L) {
  SI *first = NULL, *cur = NULL;
  int64_t si[2];
  for (int j = len - 1; j >= L - 1; j--) {
    ko_initial_si(f, str[j], si);
    int i = extend_left(f, str, j, si);
    int l = j - i + 1;
    if (l >= L) {
      if (!cur || i < cur->qi) {
        cur = alloc_SI(si, i, l);
        first = insert_SI_sorted(first, cur);
      }
    }
    if (i <= 1) break;
  }
  return first;
}

/* insert_SI_sorted with the ->count bookkeeping, bwt.c:225-252 */
static SI *insert_SI_sorted_cnt(SI *base, SI *nw) {
  nw->count = nw->len;
  if (!base) return nw;
  if (base->ql < nw->ql) { nw->next = base; nw->count += base->count; return nw; }
  SI *tmp = base;
  while (tmp->next && tmp->next->ql >= nw->ql) { tmp->count += nw->len; tmp = tmp->next; }
  tmp->count += nw->len;
  if (tmp->ql == nw->ql) {
    nw->samelen = tmp->samelen;
    if (tmp->samelen) nw->count += tmp->samelen->count;
    tmp->samelen = nw;
  } else {
    nw->next = tmp->next;
    if (tmp->next) nw->count += tmp->next->count;
    tmp->next = nw;
  }
  return base;
}
/* free_until_max_SI, bwt.c:204-219: drop the shortest length classes while at least max rows stay */
static int free_until_max_SI(SI *si, int max) {
  if (!si || si->count <= max) return 0;
  int n = si->count;
  SI *cur = si;
  while (cur->next && n - cur->next->count < max) cur = cur->next;
  if (cur->next) {
    n = cur->next->count;
    free_SI_rec(cur->next);
    cur->next = NULL;
    while (si) { si->count -= n; si = si->next; }
  }
  return cur->ql;
}
/* maxMatches with max_matches > 0, bwt.c:261-296 (kaijux's MEM search, ConsumerThreadx.cpp:135) */
static SI *maxMatches_limited(ko_index *f, const uint8_t *str, int len, int L, int max_matches) {
  SI *first = NULL, *cur = NULL;
  int64_t si[2];
  for (int j = len - 1; j >= L - 1; j--) {
    ko_initial_si(f, str[j], si);
    int i = extend_left(f, str, j, si);
    int l = j - i + 1;
    if (l >= L) {
      if (!cur || i < cur->qi) {
        cur = alloc_SI(si, i, l);
        first = insert_SI_sorted_cnt(first, cur);
        int k = free_until_max_SI(first, max_matches);
        if (k > L) L = k;
        if (l < k) cur = NULL;
      }
    }
    if (i <= 1) break;
  }
  return first;
}

/* maxMatches_withStart, bwt.c:298-336 */
static SI *maxMatches_withStart(ko_index *f, const uint8_t *str, int len, int L,
                                int64_t si0, int64_t si1, int offset) {
  int64_t si[2] = {si0, si1};
  int j = len - 1;
  int i = extend_left(f, str, j - offset + 1, si);
  int l = j - i + 1;
  if (l >= L) return alloc_SI(si, i, l);
  return NULL;
}

/* ------------------------------------------------------------------ */
/* SEG (blast_seg.c:1596-2332 with window 12, locut 2.2, hicut 2.5,    */
/* maxtrim 50, maxbogus 2, overlaps TRUE; alphabet = 20 aa, no bogus)  */
/* ------------------------------------------------------------------ */
#define SEG_WINDOW 12
static const double SEG_LOCUT = 2.2, SEG_HICUT = 2.5;   /* blast_seg.c:48-50 */
#define SEG_MAXTRIM 50
static const double kLn20 = 2.9957322735539909;          /* blast_seg.c:2193 */
static const double kLn2 = 0.693147180559945309417232121458176568; /* NCBIMATH_LN2 */

#define LNFACT_N 10001
static double lnfact_tab[LNFACT_N];   /* blast_seg.c:52-1308: ln(n!) printed with 6 decimals */
static int lnfact_ready = 0;
static void init_lnfact(void) {
  if (lnfact_ready) return;
  char buf[64];
  for (int n = 0; n < LNFACT_N; n++) {
    snprintf(buf, sizeof buf, "%.6f", lgamma((double)n + 1.0));
    lnfact_tab[n] = strtod(buf, NULL);
  }
  lnfact_ready = 1;
}
/* the table as this restatement derives it (tests/test_seg_tables_pin.py compares it with the reference's printed table) */
int ko_lnfact_n(void) { return LNFACT_N; }
double ko_lnfact(int n) { init_lnfact(); return (n >= 0 && n < LNFACT_N) ? lnfact_tab[n] : -1.0; }

/* s_lnfact, blast_seg.c:1850-1855 */
static double s_lnfact(int n) {
  if (n < LNFACT_N) return lnfact_tab[n];
  return ((n + 0.5) * log(n) - n + 0.9189385332);
}

typedef struct SSeg { int begin, end; struct SSeg *next; } SSeg;

/* a window over seq[start..start+length): composition + descending state vector */
typedef struct {
  const uint8_t *seq;      /* aa codes 0..19 of the PARENT sequence */
  int parent_len;
  int start, length;
  int comp[20];
  int state[22];
  double entropy;          /* > -2 => maintained on shift */
} Win;

static int cmp_desc(const void *a, const void *b) { return *(const int *)b - *(const int *)a; }

/* s_OpenWin + s_CompOn + s_StateOn, blast_seg.c:1490-1594 */
static void win_open(Win *w, const uint8_t *seq, int parent_len, int start, int length) {
  w->seq = seq; w->parent_len = parent_len; w->start = start; w->length = length;
  memset(w->comp, 0, sizeof w->comp);
  for (int i = 0; i < length; i++) w->comp[seq[start + i]]++;
  int nel = 0;
  memset(w->state, 0, sizeof w->state);
  for (int a = 0; a < 20; a++) if (w->comp[a]) w->state[nel++] = w->comp[a];
  qsort(w->state, (size_t)nel, sizeof(int), cmp_desc);
  w->entropy = -2.0;
}

/* s_Entropy, blast_seg.c:1596-1626 (total is never 10 here, so the log() branch) */
static double seg_entropy(const int *sv) {
  int total = 0;
  for (int i = 0; sv[i] != 0; i++) total += sv[i];
  if (total == 0) return 0.;
  double ent = 0.0;
  for (int i = 0; sv[i] != 0; i++)
    ent += ((double)sv[i]) * log(((double)sv[i]) / (double)total) / kLn2;
  return fabs(ent / (double)total);
}

/* s_DecrementSV / s_IncrementSV, blast_seg.c:1639-1666 */
static void sv_dec(int *sv, int cls) {
  int svi;
  while ((svi = *sv++) != 0) {
    if (svi == cls && *sv < cls) { sv[-1] = svi - 1; break; }
  }
}
static void sv_inc(int *sv, int cls) {
  for (;;) { if (*sv++ == cls) { sv[-1]++; break; } }
}

/* s_ShiftWin1, blast_seg.c:1673-1708 */
static int win_shift(Win *w) {
  if (w->start + 1 + w->length > w->parent_len) return 0;
  int out = w->seq[w->start], in = w->seq[w->start + w->length];
  w->start++;
  sv_dec(w->state, w->comp[out]--);
  sv_inc(w->state, w->comp[in]++);
  if (w->entropy > -2.) w->entropy = seg_entropy(w->state);
  return 1;
}

/* s_LnPerm, blast_seg.c:1864-1879 */
static double ln_perm(const int *sv, int window_length) {
  double ans = s_lnfact(window_length);
  for (int i = 0; sv[i] != 0; i++) ans -= s_lnfact(sv[i]);
  return ans;
}
/* s_LnAss, blast_seg.c:1890-1933 */
static double ln_ass(const int *sv, int alphasize) {
  double ans = lnfact_tab[alphasize];
  if (sv[0] == 0) return ans;
  int total = alphasize, cls = 1, svi = *sv, svim1 = sv[0];
  for (int i = 0;; svim1 = svi) {
    if (++i == alphasize) { ans -= s_lnfact(cls); break; }
    else if ((svi = *++sv) == svim1) { cls++; continue; }
    else {
      total -= cls;
      ans -= s_lnfact(cls);
      if (svi == 0) { ans -= s_lnfact(total); break; }
      else { cls = 1; continue; }
    }
  }
  return ans;
}
/* s_GetProb, blast_seg.c:1944-1967 */
static double get_prob(const int *sv, int total) {
  double totseq = ((double)total) * kLn20;
  double ans1 = ln_ass(sv, 20);
  double ans2 = ln_perm(sv, total);
  return ans1 + ans2 - totseq;
}

/* s_Trim, blast_seg.c:1971-2015: seq[0..len) is the raw segment */
static void seg_trim(const uint8_t *seq, int len, int *leftend, int *rightend) {
  int lend = 0, rend = len - 1, minlen = 1;
  if ((len - SEG_MAXTRIM) > minlen) minlen = len - SEG_MAXTRIM;
  double minprob = 1.;
  Win w;
  for (int l = len; l > minlen; l--) {
    int shift = 1, i = 0;
    win_open(&w, seq, len, 0, l);
    while (shift) {
      double prob = get_prob(w.state, l);
      if (prob < minprob) { minprob = prob; lend = i; rend = l + i - 1; }
      shift = win_shift(&w);
      i++;
    }
  }
  *leftend = *leftend + lend;
  *rightend = *rightend - (len - rend - 1);
}

/* s_SegSeq, blast_seg.c:2027-2113 (s_SeqEntropy :1751-1798, s_FindLow/High :1810-1845) */
static void seg_seq(const uint8_t *seq, int len, SSeg **segs, int offset) {
  const int window = SEG_WINDOW;
  const int downset = (window + 1) / 2 - 1, upset = window - downset;
  if (window > len) return;
  double *H = (double *)malloc(sizeof(double) * (size_t)len);
  for (int i = 0; i < len; i++) H[i] = -1.;
  Win w;
  win_open(&w, seq, len, 0, window);
  w.entropy = seg_entropy(w.state);
  int first = downset, last = len - upset;
  for (int i = first; i <= last; i++) { H[i] = w.entropy; win_shift(&w); }

  int lowlim = first;
  for (int i = first; i <= last; i++) {
    if (H[i] <= SEG_LOCUT && H[i] != -1.0) {
      int loi, hii, j;
      for (j = i; j >= lowlim; j--) { if (H[j] == -1.0) break; if (H[j] > SEG_HICUT) break; }
      loi = j + 1;
      for (j = i; j <= last; j++) { if (H[j] == -1.0) break; if (H[j] > SEG_HICUT) break; }
      hii = j - 1;
      int leftend = loi - downset, rightend = hii + upset - 1;
      seg_trim(seq + leftend, rightend - leftend + 1, &leftend, &rightend);
      if (i + upset - 1 < leftend) {
        int lend = loi - downset, rend = leftend - 1;
        SSeg *leftsegs = NULL;
        seg_seq(seq + lend, rend - lend + 1, &leftsegs, offset + lend);
        if (leftsegs != NULL) {
          /* blast_seg.c:2093-2097: only the first node survives; the rest leaks */
          SSeg *rest = leftsegs->next;
          while (rest) { SSeg *n = rest->next; free(rest); rest = n; }
          leftsegs->next = *segs;
          *segs = leftsegs;
        }
      }
      SSeg *seg = (SSeg *)calloc(1, sizeof(SSeg));
      seg->begin = leftend + offset;
      seg->end = rightend + offset;
      seg->next = *segs;
      *segs = seg;
      i = hii < rightend + downset ? hii : rightend + downset;
      lowlim = i + 1;
    }
  }
  free(H);
}

/* s_MergeSegs with hilenmin = 0, blast_seg.c:2122-2152 */
static void seg_merge(SSeg *segs) {
  if (!segs) return;
  SSeg *seg = segs, *nextseg = seg->next;
  while (nextseg != NULL) {
    if (seg->begin - nextseg->end - 1 < 0) {
      if (seg->end < nextseg->end) seg->end = nextseg->end;
      if (seg->begin > nextseg->begin) seg->begin = nextseg->begin;
      seg->next = nextseg->next;
      free(nextseg);
    } else seg = nextseg;
    nextseg = seg->next;
  }
}

/* SeqBufferSeg, blast_seg.c:2278-2332: codes = aa2int codes; returns regions ascending */
static int seg_codes(const uint8_t *codes, int len, int32_t *left, int32_t *right, int max_regions) {
  init_lnfact();
  g_cnt.seg_calls++;
  SSeg *segs = NULL;
  seg_seq(codes, len, &segs, 0);
  seg_merge(segs);
  /* s_SegsToBlastSeqLoc (blast_seg.c:2162-2171) reverses the list */
  int n = 0;
  for (SSeg *s = segs; s; s = s->next) n++;
  int k = n;
  for (SSeg *s = segs; s;) {
    --k;
    if (k < max_regions) { left[k] = s->begin; right[k] = s->end; }
    SSeg *nx = s->next; free(s); s = nx;
  }
  return n;
}

int ko_seg(const char *aa, int len, int32_t *left, int32_t *right, int max_regions) {
  init_tables();
  uint8_t *codes = (uint8_t *)malloc((size_t)len + 1);
  for (int i = 0; i < len; i++) codes[i] = aa2int[(uint8_t)aa[i]];
  int n = seg_codes(codes, len, left, right, max_regions);
  free(codes);
  return n;
}

/* ------------------------------------------------------------------ */
/* fragments (ConsumerThread.hpp:46-62) + the key-descending multimap   */
/* ------------------------------------------------------------------ */
typedef struct Fragment {
  char *seq;
  int len;
  unsigned num_mm;
  int diff;
  int64_t si0, si1;
  int matchlen;
  int segchecked;
} Fragment;

typedef struct { unsigned key; Fragment *f; } QItem;
typedef struct { QItem *a; int n, cap; } FragQueue;

static Fragment *frag_new(const char *s, int len) {
  Fragment *f = (Fragment *)calloc(1, sizeof *f);
  f->seq = (char *)malloc((size_t)len + 1);
  memcpy(f->seq, s, (size_t)len);
  f->seq[len] = 0;
  f->len = len;
  return f;
}
static void frag_free(Fragment *f) { if (f) { free(f->seq); free(f); } }

/* std::multimap<unsigned, Fragment*, std::greater>::emplace: after all keys >= key */
static void fq_push(FragQueue *q, unsigned key, Fragment *f) {
  if (q->n == q->cap) { q->cap = q->cap ? q->cap * 2 : 64; q->a = (QItem *)realloc(q->a, sizeof(QItem) * (size_t)q->cap); }
  int pos = q->n;
  while (pos > 0 && q->a[pos - 1].key < key) pos--;
  memmove(q->a + pos + 1, q->a + pos, sizeof(QItem) * (size_t)(q->n - pos));
  q->a[pos].key = key; q->a[pos].f = f;
  q->n++;
}
static Fragment *fq_pop(FragQueue *q, unsigned *key) {
  Fragment *f = q->a[0].f;
  if (key) *key = q->a[0].key;
  memmove(q->a, q->a + 1, sizeof(QItem) * (size_t)(q->n - 1));
  q->n--;
  return f;
}
static void fq_clear(FragQueue *q) {
  for (int i = 0; i < q->n; i++) frag_free(q->a[i].f);
  q->n = 0;
}

/* calcScore x3, ConsumerThread.cpp:397-421 */
static unsigned calc_score_plain(const char *s, int len) {
  unsigned score = 0;
  for (int i = 0; i < len; i++) score += (unsigned)B62[aa2int[(uint8_t)s[i]]][aa2int[(uint8_t)s[i]]];
  return score;
}
static unsigned calc_score_range(const char *s, int start, int len, int diff) {
  int score = 0;
  for (int i = start; i < start + len; i++) score += B62[aa2int[(uint8_t)s[i]]][aa2int[(uint8_t)s[i]]];
  score += diff;
  return score > 0 ? (unsigned)score : 0;
}

/* per-read state (ConsumerThread.hpp:64-106) */
typedef struct {
  ko_index *ix;
  const ko_params *p;
  FragQueue q;
} Ctx;

static void emit_fragment(Ctx *c, const char *s, int len) {
  if ((unsigned)len >= c->p->min_fragment_length) {
    if (c->p->mode == 1) {
      unsigned score = calc_score_plain(s, len);
      if (score >= c->p->min_score) fq_push(&c->q, score, frag_new(s, len));
    } else fq_push(&c->q, (unsigned)len, frag_new(s, len));
  }
}

/* getAllFragmentsBits, ConsumerThread.cpp:190-270 */
static void get_all_fragments(Ctx *c, const char *line, int len) {
  char *tr[3];
  int tl[3] = {0, 0, 0};
  for (int i = 0; i < 3; i++) tr[i] = (char *)malloc((size_t)len + 4);
  for (int count = 0; count < len - 2; count++) {
    const char *cd = line + count;
    uint8_t ci = (uint8_t)(nuc2int[(uint8_t)cd[0]] << 4 | nuc2int[(uint8_t)cd[1]] << 2 | nuc2int[(uint8_t)cd[2]]);
    char aa = codon2aa[ci];
    int idx = count % 3;
    if (aa == '*') { emit_fragment(c, tr[idx], tl[idx]); tl[idx] = 0; }
    else tr[idx][tl[idx]++] = aa;
  }
  for (int i = 0; i < 3; i++) { emit_fragment(c, tr[i], tl[i]); tl[i] = 0; }
  /* reverse strand: the first iteration (count = len-2) reads the string's NUL
     terminator as third base -> '*' on an empty frame, i.e. a no-op */
  for (int count = len - 3; count >= 0; count--) {
    const char *cd = line + count;
    uint8_t ci = (uint8_t)(compnuc2int[(uint8_t)cd[2]] << 4 | compnuc2int[(uint8_t)cd[1]] << 2 | compnuc2int[(uint8_t)cd[0]]);
    char aa = codon2aa[ci];
    int idx = count % 3;
    if (aa == '*') { emit_fragment(c, tr[idx], tl[idx]); tl[idx] = 0; }
    else tr[idx][tl[idx]++] = aa;
  }
  for (int i = 0; i < 3; i++) emit_fragment(c, tr[i], tl[i]);
  for (int i = 0; i < 3; i++) free(tr[i]);
}

/* protein input, ConsumerThread.cpp:659-696 (= ConsumerThreadp.cpp:22-63): upper case, split at every
   character that is not one of the 20 amino acids, runs of at least min_fragment_length (Greedy:
   scoring at least min_score) become fragments, left to right */
static void get_protein_fragments(Ctx *c, const char *line, int len) {
  static const char AA20[] = "ACDEFGHIKLMNPQRSTVWY";
  char *up = (char *)malloc((size_t)len + 1);
  for (int i = 0; i < len; i++) { char ch = line[i]; up[i] = (ch >= 'a' && ch <= 'z') ? (char)(ch - 32) : ch; }   /* toupper, C locale */
  int start = 0;
  for (int pos = 0; pos < len; pos++) {
    if (up[pos] != 0 && strchr(AA20, up[pos])) continue;
    emit_fragment(c, up + start, pos - start);              /* (pos-start >= min_fragment_length is tested inside) */
    start = pos + 1;
  }
  emit_fragment(c, up + start, len - start);                /* the remaining sequence */
  free(up);
}

int ko_fragments(const ko_params *p, const char *read, int len,
                 char *buf, int bufsize, uint32_t *keys, int max_frags) {
  init_tables();
  Ctx c; memset(&c, 0, sizeof c); c.p = p;
  if (p->protein) get_protein_fragments(&c, read, len);
  else if (len >= 3) get_all_fragments(&c, read, len);
  int n = 0, off = 0;
  for (int i = 0; i < c.q.n && n < max_frags; i++) {
    Fragment *f = c.q.a[i].f;
    if (off + f->len + 1 > bufsize) break;
    memcpy(buf + off, f->seq, (size_t)f->len + 1);
    off += f->len + 1;
    keys[n++] = c.q.a[i].key;
  }
  fq_clear(&c.q); free(c.q.a);
  return n;
}

/* getNextFragment, ConsumerThread.cpp:272-342 */
static Fragment *get_next_fragment(Ctx *c, unsigned min_score) {
  if (c->q.n == 0) return NULL;
  if (c->q.a[0].key < min_score) return NULL;
  Fragment *f = fq_pop(&c->q, NULL);
  const unsigned m = c->p->min_fragment_length;
  while (c->p->seg && f != NULL && !f->segchecked) {
    int32_t left[256], right[256];
    uint8_t *codes = (uint8_t *)malloc((size_t)f->len + 1);
    for (int i = 0; i < f->len; i++) codes[i] = aa2int[(uint8_t)f->seq[i]];
    int nreg = seg_codes(codes, f->len, left, right, 256);
    free(codes);
    if (nreg > 256) nreg = 256;
    if (nreg > 0) {
      size_t start = 0;
      for (int r = 0; r <= nreg; r++) {
        /* unsigned arithmetic as in :295,310 */
        size_t length = (r < nreg) ? (size_t)left[r] - start : (size_t)f->len - start;
        if (length > m) {
          size_t avail = (start <= (size_t)f->len) ? (size_t)f->len - start : 0;
          size_t take = length < avail ? length : avail;   /* std::string::substr clamps */
          if (c->p->mode == 1) {
            unsigned score = calc_score_range(f->seq, (int)start, (int)take, 0);
            if (score >= c->p->min_score) {
              Fragment *nf = frag_new(f->seq + start, (int)take); nf->segchecked = 1;
              fq_push(&c->q, score, nf);
            }
          } else {
            Fragment *nf = frag_new(f->seq + start, (int)take); nf->segchecked = 1;
            fq_push(&c->q, (unsigned)length, nf);
          }
        }
        if (r < nreg) start = (size_t)right[r] + 1;
      }
      frag_free(f);
      f = NULL;
      if (c->q.n > 0 && c->q.a[0].key >= min_score) f = fq_pop(&c->q, NULL);
    } else return f;
  }
  return f;
}

/* translate2numbers, sequence.c:151-154 */
static uint8_t *to_numbers(const ko_index *ix, const char *s, int len) {
  uint8_t *r = (uint8_t *)malloc((size_t)len + 1);
  for (int i = 0; i < len; i++) r[i] = (uint8_t)ix->trans[(uint8_t)s[i] & 127];
  return r;
}

/* ------------------------------------------------------------------ */
/* match-id collection (ids_from_SI, ConsumerThread.cpp:799-845)       */
/* ------------------------------------------------------------------ */
typedef struct { uint64_t id[KO_MAX_IDS + 1]; int n; int cap_hit; } IdSet;

static void ids_from_SI(Ctx *c, IdSet *ids, int64_t start, int len) {
  for (int64_t k = start; k < start + len; ++k) {
    if ((unsigned)ids->n > c->p->max_match_ids) { ids->cap_hit = 1; break; }
    int32_t iseq; int64_t pos;
    ko_get_suffix(c->ix, k, &iseq, &pos);
    int ok;
    uint64_t id = ko_seq_taxid(c->ix, iseq, &ok);
    if (c->p->kaijux) { id = (uint64_t)iseq; ok = 1; }      /* ConsumerThreadx.cpp:261-287: the sequence itself */
    if (!ok) continue;
    int seen = 0;
    for (int i = 0; i < ids->n; i++) if (ids->id[i] == id) { seen = 1; break; }
    if (!seen && ids->n < KO_MAX_IDS) ids->id[ids->n++] = id;
  }
}

/* ------------------------------------------------------------------ */
/* MEM: classify_length, ConsumerThread.cpp:543-628                    */
/* ------------------------------------------------------------------ */
static void classify_length(Ctx *c, ko_hit *out, IdSet *ids) {
  unsigned longest = 0;
  SI **heads = NULL; int nh = 0, caph = 0;
  for (;;) {
    Fragment *t = get_next_fragment(c, longest);
    if (!t) break;
    g_cnt.fragments_searched++;
    uint8_t *seq = to_numbers(c->ix, t->seq, t->len);
    unsigned L = c->p->min_fragment_length > longest ? c->p->min_fragment_length : longest;
    SI *si = c->p->kaijux ? maxMatches_limited(c->ix, seq, t->len, (int)L, 1) : greedyExact(c->ix, seq, t->len, (int)L);
    free(seq);
    frag_free(t);
    if (!si) continue;
    if ((unsigned)si->ql > longest) {
      for (int i = 0; i < nh; i++) free_SI_rec(heads[i]);
      nh = 0;
      longest = (unsigned)si->ql;
    } else if ((unsigned)si->ql != longest) { free_SI_rec(si); continue; }
    if (nh == caph) { caph = caph ? caph * 2 : 16; heads = (SI **)realloc(heads, sizeof(SI *) * (size_t)caph); }
    heads[nh++] = si;
  }
  out->best = longest;
  if (nh == 0) { free(heads); return; }
  for (int i = 0; i < nh; i++)
    for (SI *it = heads[i]; it; it = it->samelen) ids_from_SI(c, ids, it->start, it->len);
  for (int i = 0; i < nh; i++) free_SI_rec(heads[i]);
  free(heads);
}

/* ------------------------------------------------------------------ */
/* Greedy: classify_greedyblosum, ConsumerThread.cpp:424-541           */
/* ------------------------------------------------------------------ */
typedef struct { int64_t start; int len; } BestSI;
typedef struct {
  BestSI si[64]; int n; unsigned best; int si_cap_hit;
} BestSet;

/* eval_match_scores, ConsumerThread.cpp:751-797 */
static void eval_match_scores(Ctx *c, BestSet *b, SI *si, const Fragment *frag) {
  if (!si) return;
  if (si->samelen) eval_match_scores(c, b, si->samelen, frag);
  if (si->next && si->next->ql >= (int)c->p->min_fragment_length) eval_match_scores(c, b, si->next, frag);
  unsigned score = calc_score_range(frag->seq, si->qi, si->ql, frag->diff);
  if (score < c->p->min_score) return;
  if (score > b->best) {
    b->n = 0; b->best = score;
    b->si[b->n].start = si->start; b->si[b->n].len = si->len; b->n++;
  } else if (score == b->best) {
    if ((unsigned)b->n < c->p->max_matches_SI && b->n < 64) {
      b->si[b->n].start = si->start; b->si[b->n].len = si->len; b->n++;
    } else b->si_cap_hit = 1;
  }
}

/* addAllMismatchVariantsAtPosSI, ConsumerThread.cpp:346-395 */
static void add_mismatch_variants(Ctx *c, BestSet *b, const Fragment *f, unsigned pos, int erase_pos, const SI *si) {
  int flen = f->len;
  char *fragment = (char *)malloc((size_t)flen + 1);
  memcpy(fragment, f->seq, (size_t)flen + 1);
  char origchar = fragment[pos];
  int o = aa2int[(uint8_t)origchar];
  if (erase_pos >= 0 && erase_pos < flen) { fragment[erase_pos] = 0; flen = erase_pos; }
  /* :363 — unsigned wrap-around is part of the behaviour */
  uint32_t score = calc_score_range(fragment, 0, flen, f->diff) - (uint32_t)(int32_t)B62[o][o];
  int64_t siarray[2] = {si->start, si->start + (int64_t)si->len}, upd[2];
  for (int v = 0; v < 19; v++) {
    char itv = subst[o][v];
    int s = aa2int[(uint8_t)itv];
    int32_t after = (int32_t)(score + (uint32_t)(int32_t)B62[o][s]);
    if (after >= (int32_t)b->best && after >= (int32_t)c->p->min_score) {
      if (ko_update_si(c->ix, (uint8_t)c->ix->trans[(uint8_t)itv], siarray, upd) != 0) {
        fragment[pos] = itv;
        int diff = B62[o][s] - B62[s][s];
        Fragment *nf = frag_new(fragment, flen);
        nf->num_mm = f->num_mm + 1;
        nf->diff = f->diff + diff;
        nf->si0 = upd[0]; nf->si1 = upd[1];
        nf->matchlen = si->ql + 1;
        nf->segchecked = 1;
        fq_push(&c->q, (unsigned)after, nf);
      }
    } else break;
  }
  free(fragment);
}

static void classify_greedy(Ctx *c, ko_hit *out, IdSet *ids, double query_len) {
  BestSet b; memset(&b, 0, sizeof b);
  const ko_params *p = c->p;
  for (;;) {
    Fragment *t = get_next_fragment(c, b.best);
    if (!t) break;
    g_cnt.fragments_searched++;
    int length = t->len;
    unsigned num_mm = t->num_mm;
    uint8_t *seq = to_numbers(c->ix, t->seq, length);
    SI *si;
    if (num_mm > 0) {
      int L = (num_mm == p->mismatches) ? (int)p->min_fragment_length : t->matchlen;
      si = maxMatches_withStart(c->ix, seq, length, L, t->si0, t->si1, t->matchlen);
    } else si = maxMatches(c->ix, seq, length, (int)p->seed_length);
    free(seq);
    if (!si) { frag_free(t); continue; }
    if (p->mismatches > 0 && num_mm < p->mismatches) {
      SI *it = si;
      while (it) {
        unsigned mre = (unsigned)(it->qi + it->ql - 1);
        if (it->qi > 0 && mre + 1 >= p->min_fragment_length) {
          int erase_pos = (mre < (unsigned)length - 1) ? (int)mre + 1 : -1;
          add_mismatch_variants(c, &b, t, (unsigned)(it->qi - 1), erase_pos, it);
        }
        it = it->samelen ? it->samelen : it->next;   /* :477 */
      }
    }
    if ((unsigned)si->ql < p->min_fragment_length) { frag_free(t); free_SI_rec(si); continue; }
    eval_match_scores(c, &b, si, t);
    free_SI_rec(si);
    frag_free(t);
  }
  out->best = b.best;
  if (b.si_cap_hit) out->flags |= 2;
  if (b.n == 0) { out->best = 0; return; }
  if (p->use_evalue) {
    /* :500-513; constants ConsumerThread.hpp:41-44 */
    const double LN_2 = 0.6931471805, LAMBDA = 0.3176, LN_K = -2.009915479;
    double db_length = (double)(c->ix->len - c->ix->nseq);   /* Config.cpp:20 */
    double bitscore = (LAMBDA * b.best - LN_K) / LN_2;
    double Evalue = db_length * query_len * pow(2, -1 * bitscore);
    if (Evalue > p->min_evalue) { out->flags |= 4; return; }
  }
  for (int i = 0; i < b.n; i++) ids_from_SI(c, ids, b.si[i].start, b.si[i].len);
}

/* ------------------------------------------------------------------ */
/* taxonomy                                                            */
/* ------------------------------------------------------------------ */
struct ko_taxonomy {
  uint64_t *key, *val;   /* open-addressing hash: node -> parent */
  uint32_t *depth;       /* node2depth cache (0 = unknown) */
  size_t cap;            /* power of two */
  size_t n;
};

static size_t tx_slot(const ko_taxonomy *t, uint64_t k) {
  size_t h = (size_t)(k * 0x9E3779B97F4A7C15ull) & (t->cap - 1);
  while (t->key[h] != UINT64_MAX && t->key[h] != k) h = (h + 1) & (t->cap - 1);
  return h;
}
static int tx_has(const ko_taxonomy *t, uint64_t k) { return t->key[tx_slot(t, k)] == k; }
static uint64_t tx_parent(const ko_taxonomy *t, uint64_t k) {
  size_t h = tx_slot(t, k);
  return t->key[h] == k ? t->val[h] : k;   /* reference would throw; treat as root */
}
static void tx_insert(ko_taxonomy *t, uint64_t k, uint64_t v) {
  if ((t->n + 1) * 2 > t->cap) {
    ko_taxonomy old = *t;
    t->cap = old.cap * 2;
    t->key = (uint64_t *)malloc(sizeof(uint64_t) * t->cap);
    t->val = (uint64_t *)malloc(sizeof(uint64_t) * t->cap);
    t->depth = (uint32_t *)calloc(t->cap, sizeof(uint32_t));
    memset(t->key, 0xFF, sizeof(uint64_t) * t->cap);
    t->n = 0;
    for (size_t i = 0; i < old.cap; i++) if (old.key[i] != UINT64_MAX) tx_insert(t, old.key[i], old.val[i]);
    free(old.key); free(old.val); free(old.depth);
  }
  size_t h = tx_slot(t, k);
  if (t->key[h] == k) return;     /* unordered_map::emplace keeps the first */
  t->key[h] = k; t->val[h] = v; t->n++;
}

/* parseNodesDmp, util.cpp:79-99 */
ko_taxonomy *ko_load_nodes(const char *path) {
  FILE *fp = fopen(path, "r");
  if (!fp) return NULL;
  ko_taxonomy *t = (ko_taxonomy *)calloc(1, sizeof *t);
  t->cap = 1 << 12;
  t->key = (uint64_t *)malloc(sizeof(uint64_t) * t->cap);
  t->val = (uint64_t *)malloc(sizeof(uint64_t) * t->cap);
  t->depth = (uint32_t *)calloc(t->cap, sizeof(uint32_t));
  memset(t->key, 0xFF, sizeof(uint64_t) * t->cap);
  char *line = NULL; size_t cap = 0; ssize_t n;
  while ((n = getline(&line, &cap, fp)) > 0) {
    const char *s = line;
    if (!isdigit((unsigned char)*s)) continue;          /* stoul on "" throws -> line skipped */
    uint64_t node = strtoull(s, (char **)&s, 10);
    while (*s && !isdigit((unsigned char)*s)) s++;
    if (!*s) continue;
    uint64_t parent = strtoull(s, NULL, 10);
    tx_insert(t, node, parent);
  }
  free(line);
  fclose(fp);
  return t;
}
void ko_free_taxonomy(ko_taxonomy *t) { if (t) { free(t->key); free(t->val); free(t->depth); free(t); } }

/* lca_from_ids, util.cpp:194-263.  ids must be distinct (std::set). */
uint64_t ko_lca(ko_taxonomy *t, const uint64_t *ids, int n) {
  if (n == 1) return ids[0];
  uint64_t leafs[64];
  unsigned depths[64];
  unsigned shallowest = 100000;
  int m = 0;
  for (int i = 0; i < n && m < 64; i++) {
    uint64_t it = ids[i];
    if (!tx_has(t, it)) continue;
    size_t h = tx_slot(t, it);
    if (t->depth[h] == 0) {
      unsigned depth = 1;
      uint64_t id = it;
      while (tx_has(t, id) && id != tx_parent(t, id) && depth < 100000) { depth++; id = tx_parent(t, id); }
      t->depth[h] = depth;
    }
    leafs[m] = it; depths[m] = t->depth[h];
    if (depths[m] < shallowest) shallowest = depths[m];
    m++;
  }
  if (m <= 0) return 0;
  for (int i = 0; i < m; i++)
    for (int d = (int)depths[i] - (int)shallowest; d > 0; d--) leafs[i] = tx_parent(t, leafs[i]);
  for (int guard = 0; guard < 200000; guard++) {
    uint64_t first = leafs[0];
    int found = 1;
    for (int i = 0; i < m; i++) {
      if (first != leafs[i]) found = 0;
      leafs[i] = tx_parent(t, leafs[i]);
    }
    if (found) return first;
  }
  return 0;
}

/* ------------------------------------------------------------------ */
/* doWork framing, ConsumerThread.cpp:630-749 (nucleotide input only)  */
/* ------------------------------------------------------------------ */
void ko_default_params(ko_params *p, int mode) {
  /* Config.hpp:33-48; "-a mem" also clears use_Evalue (kaiju.cpp:77-80) */
  p->mode = mode;
  p->min_fragment_length = 11;
  p->mismatches = 3;
  p->min_score = 65;
  p->seed_length = 7;
  p->seg = 1;
  p->use_evalue = (mode == 1);
  p->min_evalue = 0.01;
  p->max_matches_SI = 20;
  p->max_match_ids = 20;
  p->kaijux = 0; p->protein = 0;
}

void ko_classify(ko_index *ix, ko_taxonomy *tax, const ko_params *p,
                 const char *seq1, int len1, const char *seq2, int len2, int paired,
                 ko_hit *out) {
  init_tables();
  memset(out, 0, sizeof *out);
  const unsigned m3 = p->min_fragment_length * 3;
  Ctx c; memset(&c, 0, sizeof c); c.ix = ix; c.p = p;
  double query_len;
  if (p->protein) {
    /* :640-646, :660; the mate is ignored (kaiju.cpp:201 refuses -j with -p) */
    if ((unsigned)len1 < p->min_fragment_length) return;
    query_len = (double)len1;
    get_protein_fragments(&c, seq1, len1);
  } else {
    if ((!paired && (unsigned)len1 < m3) || (paired && (unsigned)len1 < m3 && (unsigned)len2 < m3)) return;
    query_len = (double)len1 / 3.0;
    if ((unsigned)len1 >= m3) get_all_fragments(&c, seq1, len1);
    if (paired) {
      query_len += (double)len2 / 3.0;
      if ((unsigned)len2 >= m3) get_all_fragments(&c, seq2, len2);
    }
  }
  IdSet ids; memset(&ids, 0, sizeof ids);
  if (p->mode == 0) classify_length(&c, out, &ids);
  else classify_greedy(&c, out, &ids, query_len);
  fq_clear(&c.q); free(c.q.a);
  out->n_ids = (uint32_t)ids.n;
  if (ids.cap_hit) out->flags |= 1;
  for (int i = 0; i < ids.n; i++) out->taxid[i] = ids.id[i];
  if (ids.n == 0) { if (!(out->flags & 4) && p->mode == 0) { /* keep best */ } return; }
  if (p->kaijux) { out->classified = 1; return; }          /* no taxonomy: reported with its sequences */
  if (tax) {
    out->lca = (ids.n == 1) ? ids.id[0] : ko_lca(tax, ids.id, ids.n);
    out->classified = out->lca > 0;
  }
}

void ko_classify_batch(ko_index *ix, ko_taxonomy *tax, const ko_params *p,
                       const char *seqs, const uint64_t *off, uint32_t n, int paired,
                       ko_hit *out) {
  for (uint32_t r = 0; r < n; r++) {
    uint64_t a = off[2 * r], b = off[2 * r + 1], e = off[2 * r + 2];
    /* the reference reads one byte past the end of each read (a NUL); copy to be safe */
    int l1 = (int)(b - a), l2 = (int)(e - b);
    char *s1 = (char *)malloc((size_t)l1 + 1), *s2 = (char *)malloc((size_t)l2 + 1);
    memcpy(s1, seqs + a, (size_t)l1); s1[l1] = 0;
    memcpy(s2, seqs + b, (size_t)l2); s2[l2] = 0;
    ko_classify(ix, tax, p, s1, l1, s2, l2, paired, &out[r]);
    free(s1); free(s2);
  }
}
