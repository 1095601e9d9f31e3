/* gen_db.c - TEST / BENCHMARK TOOL: a synthetic protein database of refseq-class size (2^32 .. 2^36 index rows) in seconds.
 * Same recipe as kaiju_amd/synth.py:make_db: lengths clip(Gamma(2, 140), 30, 3000), residues i.i.d. from
 * UniProt-like background frequencies, 35 % of the sequences mutated copies (1 / 5 / 15 % substitutions) of an earlier
 * original, headers >WPnnnnnnnnn.1_<taxid> with the 5000 leaf taxa of synth.make_taxonomy().
 *   gen_db <nseq> <seed> <out.faa> <out.codes (uint8)> <out.offsets (int64, nseq+1)> <out.taxids (int64)>
 * Lengths, sources and taxa come from one serial generator; the residues of sequence i from a generator seeded with
 * (seed, i), so originals and then copies are filled by all cores (gcc -O2 -fopenmp; without OpenMP the same output, serially).
 * The FASTA file and the code file are written through shared mappings, sequence by sequence, in parallel as well.
 */
#define _GNU_SOURCE
#include <fcntl.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <unistd.h>

typedef struct { uint64_t s[2]; } rng_t;
static inline uint64_t rnd(rng_t *g) {            /* xorshift128+ */
  uint64_t a = g->s[0]; const uint64_t b = g->s[1];
  g->s[0] = b; a ^= a << 23; g->s[1] = a ^ b ^ (a >> 18) ^ (b >> 5);
  return g->s[1] + b;
}
static inline double uni(rng_t *g) { return ((rnd(g) >> 11) + 0.5) * (1.0 / 9007199254740992.0); }
static inline uint64_t mix(uint64_t x) {           /* splitmix64 finaliser */
  x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 27; x *= 0x94d049bb133111ebull; x ^= x >> 31; return x;
}
static inline void seed_for(rng_t *g, uint64_t seed, uint64_t i) {
  g->s[0] = mix(seed * 0x9E3779B97F4A7C15ull + 2 * i + 1); g->s[1] = mix(g->s[0] ^ 0xD1B54A32D192ED03ull) | 1ull;
  for (int k = 0; k < 4; k++) rnd(g);
}

static void *map_out(const char *path, size_t bytes) {
  const int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
  if (fd < 0) { perror(path); exit(1); }
  if (ftruncate(fd, (off_t)bytes) != 0) { perror("ftruncate"); exit(1); }
  void *p = mmap(NULL, bytes ? bytes : 1, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) { perror("mmap"); exit(1); }
  return p;
}

int main(int argc, char **argv) {
  if (argc < 7) { fprintf(stderr, "usage: gen_db nseq seed out.faa out.codes out.offsets out.taxids\n"); return 2; }
  const int64_t nseq = atoll(argv[1]);
  const uint64_t seed = (uint64_t)atoll(argv[2]);
  if (nseq < 1 || nseq > 999999999) { fprintf(stderr, "gen_db: nseq out of range (headers hold nine digits)\n"); return 2; }
  rng_t g0;
  g0.s[0] = 0x9E3779B97F4A7C15ull ^ seed; g0.s[1] = 0xD1B54A32D192ED03ull;
  for (int i = 0; i < 20; i++) rnd(&g0);
  static const char AA[] = "ACDEFGHIKLMNPQRSTVWY";
  static const double BG[20] = {8.25, 1.37, 5.45, 6.75, 3.86, 7.07, 2.27, 5.96, 5.84, 9.66,
                                2.42, 4.06, 4.70, 3.93, 5.53, 6.56, 5.34, 6.87, 1.08, 2.92};
  uint8_t *lut = malloc(65536);
  { double tot = 0, acc = 0; for (int a = 0; a < 20; a++) tot += BG[a];
    int a = 0; acc = BG[0] / tot;
    for (int v = 0; v < 65536; v++) { while (a < 19 && (v + 0.5) / 65536.0 > acc) { a++; acc += BG[a] / tot; } lut[v] = (uint8_t)a; } }
  int64_t *len = malloc(sizeof(int64_t) * nseq), *src = malloc(sizeof(int64_t) * nseq), *off = malloc(sizeof(int64_t) * (nseq + 1));
  int64_t *tax = malloc(sizeof(int64_t) * nseq), *orig = malloc(sizeof(int64_t) * nseq);
  if (!len || !src || !off || !tax || !orig) { fprintf(stderr, "out of memory\n"); return 1; }
  int64_t norig = 0, last_orig = 0;
  for (int64_t i = 0; i < nseq; i++) {
    double g = -140.0 * log(uni(&g0) * uni(&g0));
    if (g < 30) g = 30;
    if (g > 3000) g = 3000;
    len[i] = (int64_t)g; src[i] = -1;
    const int copy = uni(&g0) < 0.35 && i >= (nseq / 100 > 1 ? nseq / 100 : 1) && norig > 0;
    if (copy) { src[i] = orig[(int64_t)(uni(&g0) * norig)]; len[i] = len[src[i]]; }
    else { orig[norig++] = i; last_orig = i; }
    tax[i] = 100000 + (int64_t)(uni(&g0) * 5000);
    if (copy && uni(&g0) < 0.7) { const int64_t b = tax[src[i]]; tax[i] = b - b % 10 + (int64_t)(uni(&g0) * 10); }
  }
  int64_t total = 0;
  for (int64_t i = 0; i < nseq; i++) total += len[i];
  /* steer clear of the reference's rank bug for bwtlen % 65536 >= 65408 (or == 0), SURVEY.md 7 */
  while ((total + nseq) % 65536 >= 65408 || (total + nseq) % 65536 == 0) { len[last_orig]++; total++; }
  off[0] = 0;
  for (int64_t i = 0; i < nseq; i++) off[i + 1] = off[i] + len[i];
  uint8_t *codes = map_out(argv[4], (size_t)total);
  static const double RATE[3] = {0.01, 0.05, 0.15};
  /* originals, then the copies (a copy's source is an original) */
  for (int pass = 0; pass < 2; pass++) {
#pragma omp parallel for schedule(dynamic, 4096)
    for (int64_t i = 0; i < nseq; i++) {
      if ((src[i] < 0) != (pass == 0)) continue;
      rng_t g;
      seed_for(&g, seed, (uint64_t)i);
      uint8_t *d = codes + off[i];
      if (src[i] < 0) {
        int64_t k = 0;
        for (; k + 4 <= len[i]; k += 4) { const uint64_t r = rnd(&g); d[k] = lut[r & 65535]; d[k + 1] = lut[(r >> 16) & 65535]; d[k + 2] = lut[(r >> 32) & 65535]; d[k + 3] = lut[r >> 48]; }
        for (; k < len[i]; k++) d[k] = lut[rnd(&g) & 65535];
      } else {
        memcpy(d, codes + off[src[i]], (size_t)len[i]);
        const double rate = RATE[rnd(&g) % 3];
        /* geometric gaps between substitutions */
        for (double p = -log(uni(&g)) / rate; p < (double)len[i]; p += 1.0 - log(uni(&g)) / rate) d[(int64_t)p] = (uint8_t)(rnd(&g) % 20);
      }
    }
  }
  /* FASTA: ">WPnnnnnnnnn.1_tttttt\n" (22 bytes: the taxa have six digits) + residues + "\n" */
  const int64_t H = 22;
  const size_t fbytes = (size_t)(total + nseq * (H + 1));
  char *fa = map_out(argv[3], fbytes);
#pragma omp parallel for schedule(dynamic, 4096)
  for (int64_t i = 0; i < nseq; i++) {
    char *o = fa + off[i] + i * (H + 1);
    char hdr[32];
    snprintf(hdr, sizeof hdr, ">WP%09lld.1_%06lld\n", (long long)i, (long long)tax[i]);
    memcpy(o, hdr, (size_t)H);
    const uint8_t *d = codes + off[i];
    for (int64_t k = 0; k < len[i]; k++) o[H + k] = AA[d[k]];
    o[H + len[i]] = '\n';
  }
  munmap(fa, fbytes);
  munmap(codes, (size_t)total);
  FILE *f = fopen(argv[5], "wb"); fwrite(off, 8, (size_t)nseq + 1, f); fclose(f);
  f = fopen(argv[6], "wb"); fwrite(tax, 8, (size_t)nseq, f); fclose(f);
  fprintf(stderr, "gen_db: %lld sequences, %lld residues, bwtlen %lld\n", (long long)nseq, (long long)total, (long long)(total + nseq));
  return 0;
}
