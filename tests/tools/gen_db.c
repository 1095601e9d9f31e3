/* gen_db.c - TEST / BENCHMARK TOOL: a synthetic protein database of refseq-class size (>= 2^32 index rows) in seconds.
 * Same recipe as kaiju_amd/synth.py:make_db: lengths clip(Gamma(2, 140), 30, 3000), residues i.i.d. from
 * UniProt-like background frequencies, 35 % of the sequences mutated copies (1 / 5 / 15 % substitutions) of an earlier
 * original, headers >WPnnnnnnnnn.1_<taxid> with the 5000 leaf taxa of synth.make_taxonomy().
 *   gen_db <nseq> <seed> <out.faa> <out.codes (uint8)> <out.offsets (int64, nseq+1)> <out.taxids (int64)>
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static uint64_t s[2];
static inline uint64_t rnd(void) {            /* xorshift128+ */
  uint64_t a = s[0]; const uint64_t b = s[1];
  s[0] = b; a ^= a << 23; s[1] = a ^ b ^ (a >> 18) ^ (b >> 5);
  return s[1] + b;
}
static inline double uni(void) { return ((rnd() >> 11) + 0.5) * (1.0 / 9007199254740992.0); }

int main(int argc, char **argv) {
  if (argc < 7) { fprintf(stderr, "usage: gen_db nseq seed out.faa out.codes out.offsets out.taxids\n"); return 2; }
  const int64_t nseq = atoll(argv[1]);
  s[0] = 0x9E3779B97F4A7C15ull ^ (uint64_t)atoll(argv[2]); s[1] = 0xD1B54A32D192ED03ull;
  for (int i = 0; i < 20; i++) rnd();
  static const char AA[] = "ACDEFGHIKLMNPQRSTVWY";
  static const double BG[20] = {8.25, 1.37, 5.45, 6.75, 3.86, 7.07, 2.27, 5.96, 5.84, 9.66,
                                2.42, 4.06, 4.70, 3.93, 5.53, 6.56, 5.34, 6.87, 1.08, 2.92};
  uint8_t *lut = malloc(65536);
  { double tot = 0, acc = 0; for (int a = 0; a < 20; a++) tot += BG[a];
    int a = 0; acc = BG[0] / tot;
    for (int v = 0; v < 65536; v++) { while (a < 19 && (v + 0.5) / 65536.0 > acc) { a++; acc += BG[a] / tot; } lut[v] = (uint8_t)a; } }
  int64_t *len = malloc(sizeof(int64_t) * nseq), *src = malloc(sizeof(int64_t) * nseq), *off = malloc(sizeof(int64_t) * (nseq + 1));
  int64_t *tax = malloc(sizeof(int64_t) * nseq), *orig = malloc(sizeof(int64_t) * nseq);
  int64_t norig = 0, last_orig = 0;
  for (int64_t i = 0; i < nseq; i++) {
    double g = -140.0 * log(uni() * uni());
    if (g < 30) g = 30; if (g > 3000) g = 3000;
    len[i] = (int64_t)g; src[i] = -1;
    const int copy = uni() < 0.35 && i >= (nseq / 100 > 1 ? nseq / 100 : 1) && norig > 0;
    if (copy) { src[i] = orig[(int64_t)(uni() * norig)]; len[i] = len[src[i]]; }
    else { orig[norig++] = i; last_orig = i; }
    tax[i] = 100000 + (int64_t)(uni() * 5000);
    if (copy && uni() < 0.7) { const int64_t b = tax[src[i]]; tax[i] = b - b % 10 + (int64_t)(uni() * 10); }
  }
  int64_t total = 0;
  for (int64_t i = 0; i < nseq; i++) total += len[i];
  /* steer clear of the reference's rank bug for bwtlen % 65536 >= 65408 (or == 0), SURVEY.md 7 */
  while ((total + nseq) % 65536 >= 65408 || (total + nseq) % 65536 == 0) { len[last_orig]++; total++; }
  off[0] = 0;
  for (int64_t i = 0; i < nseq; i++) off[i + 1] = off[i] + len[i];
  uint8_t *codes = malloc((size_t)total);
  if (!codes) { fprintf(stderr, "out of memory\n"); return 1; }
  static const double RATE[3] = {0.01, 0.05, 0.15};
  for (int64_t i = 0; i < nseq; i++) {
    uint8_t *d = codes + off[i];
    if (src[i] < 0) {
      int64_t k = 0;
      for (; k + 4 <= len[i]; k += 4) { const uint64_t r = rnd(); d[k] = lut[r & 65535]; d[k + 1] = lut[(r >> 16) & 65535]; d[k + 2] = lut[(r >> 32) & 65535]; d[k + 3] = lut[r >> 48]; }
      for (; k < len[i]; k++) d[k] = lut[rnd() & 65535];
    } else {
      memcpy(d, codes + off[src[i]], (size_t)len[i]);
      const double rate = RATE[rnd() % 3];
      /* geometric gaps between substitutions */
      for (double p = -log(uni()) / rate; p < (double)len[i]; p += 1.0 - log(uni()) / rate) d[(int64_t)p] = (uint8_t)(rnd() % 20);
    }
  }
  FILE *f = fopen(argv[3], "wb");
  if (!f) { perror(argv[3]); return 1; }
  setvbuf(f, NULL, _IOFBF, 1 << 24);
  char *line = malloc(3100);
  for (int64_t i = 0; i < nseq; i++) {
    fprintf(f, ">WP%09lld.1_%lld\n", (long long)i, (long long)tax[i]);
    const uint8_t *d = codes + off[i];
    for (int64_t k = 0; k < len[i]; k++) line[k] = AA[d[k]];
    line[len[i]] = '\n';
    fwrite(line, 1, (size_t)len[i] + 1, f);
  }
  fclose(f);
  f = fopen(argv[4], "wb"); fwrite(codes, 1, (size_t)total, f); fclose(f);
  f = fopen(argv[5], "wb"); fwrite(off, 8, (size_t)nseq + 1, f); fclose(f);
  f = fopen(argv[6], "wb"); fwrite(tax, 8, (size_t)nseq, f); fclose(f);
  fprintf(stderr, "gen_db: %lld sequences, %lld residues, bwtlen %lld\n", (long long)nseq, (long long)total, (long long)(total + nseq));
  return 0;
}
