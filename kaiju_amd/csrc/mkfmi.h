/* mkfmi.h - index construction, TEST / BENCHMARK INFRASTRUCTURE (not part of the product boundary include/kaiju_gpu.h).
 *
 * Building a .fmi is the reference's off-line step (kaiju-mkbwt + kaiju-mkfmi) and out of scope of the classification path
 * (OUT OF SCOPE).  There is no network on the GPU box and the reference's binaries do not travel there, so
 * bench.py and the tests need a way to make the indexes they classify on: this builder writes byte-identical files
 * (tests/test_mkfmi_pin.py).  It is compiled into a library of its own, kaiju_amd/libkaiju_mkfmi.so (host only, no HIP);
 * libkaiju_gpu.so does not contain it.
 */
#ifndef KAIJU_MKFMI_H
#define KAIJU_MKFMI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status codes: the values of include/kaiju_gpu.h's enum (0 ok, -1 bad argument, -2 file error) */
#ifndef KAIJU_GPU_H
enum { KAIJU_GPU_OK = 0, KAIJU_GPU_ERR_ARG = -1, KAIJU_GPU_ERR_IO = -2 };
#endif

/* Protein FASTA -> .fmi, format-compatible with the reference's kaiju-mkbwt (-a
   ACDEFGHIKLMNPQRSTVWY -e chpt_exp, util/kaiju-makedb:373) followed by kaiju-mkfmi
   (bwt/mkbwt.c:922-1099, bwt/mkfmi.c:21-97); the reference binary reads the result.
   threads <= 0: all hardware threads.  Host only. */
int kaiju_build_fmi(const char *faa_path, const char *out_fmi_path, int threads, int chpt_exp);
/* The .fmi of the database in which every sequence of the FASTA occurs `copies` times in a row, without sorting it again (equal
   suffixes order by file position, so every row of the file's own index becomes `copies` rows); byte for byte what
   kaiju_build_fmi writes for the FASTA with the repeats spelled out.  Test / benchmark infrastructure: an index of 2^32 rows
   and more (the layout with 64-bit positions) from a small FASTA in seconds.  copy_taxids (may be NULL): copy t of sequence
   number i, named X_<id>, is named X_<copy_taxids[(i + t) % n_copy_taxids]>. */
int kaiju_build_fmi_replicated(const char *faa_path, const char *out_fmi_path, int threads, int chpt_exp, uint64_t copies,
                               const uint64_t *copy_taxids, uint32_t n_copy_taxids);
const char *kaiju_build_fmi_error(void);

#ifdef __cplusplus
}
#endif

#endif
