// randreach.hip — how does the rate of random 128-byte lines depend on the FOOTPRINT they are spread over?
// (VERDICT r02 "what's weak" 3: the wide-index MEM lane reached a third of the line rate of the narrow one on a 40 GB index;
// randbench.hip was calibrated at 128 MiB and 1 GiB only.)  ONE allocation of <alloc GiB>; the lanes draw hash-addressed
// lines from the first <window> GiB of it, for a list of windows - so the only thing that changes between the rows of the
// output is the reach of the accesses (address translation, channel interleave), not the allocation.
//   randreach <alloc GiB> <loads per lane> <mode: 1 = one 16-B load per line, 4 = the RankBlock64 query pattern> <blocks per CU>
//             <window GiB> [<window GiB> ...]     (window 0 = 256 MiB, fits the Infinity Cache)
// A second experiment (mode 14): every other load goes to a SECOND window at the far end of the allocation (k-mer table vs
// rank blocks of the wide index: two regions, alternating).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

__device__ __forceinline__ uint64_t mix(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return x;
}

template <int MODE>
__global__ void __launch_bounds__(256) k_reach(const uint4 *__restrict__ tab, uint64_t nlines, uint64_t far_line0, int iters,
                                               uint32_t *out) {
  uint64_t s = mix(blockIdx.x * 256ull + threadIdx.x + 1);
  uint32_t acc = 0;
  for (int i = 0; i < iters; ++i) {
    s = mix(s + 0x9e3779b97f4a7c15ULL);
    uint64_t b = (uint64_t)(((unsigned __int128)s * nlines) >> 64);
    if (MODE == 14 && (i & 1)) b += far_line0;
    const uint4 *p = tab + b * 8;
    if (MODE == 1) { const uint4 v = p[s & 7]; acc ^= v.x ^ v.w; }
    else if (MODE == 2) {            // two 16-byte loads of one lane in the same 64-byte half (k-mer line: entry + presence bits)
      const uint4 v0 = p[(s & 4) + 0], v1 = p[(s & 4) + 1 + (s & 1)]; acc ^= v0.x ^ v1.y;
    } else if (MODE == 3) {          // two loads of one lane in the two halves of the line
      const uint4 v0 = p[s & 3], v1 = p[4 + ((s >> 2) & 3)]; acc ^= v0.x ^ v1.y;
    } else if (MODE == 41 || MODE == 42) {
      // quad-cooperative: the four lanes of a quad fetch ONE line together (the line of the quad's lane 0: one DPP broadcast),
      // 41: the four chunks of one 64-byte half; 42: three chunks of the first half + one chunk of the rest (rank block: five
      // planes + one count).  Lines per wave instruction: 16
      const uint64_t b0 = (uint64_t)__shfl((unsigned long long)b, (int)(threadIdx.x & 60u), 64);
      const uint64_t s0 = (uint64_t)__shfl((unsigned long long)s, (int)(threadIdx.x & 60u), 64);
      const uint4 *q = tab + b0 * 8;
      const uint32_t l4 = threadIdx.x & 3u;
      const uint32_t ch = MODE == 41 ? (uint32_t)(s0 & 4) + l4 : (l4 < 3u ? l4 : 2u + (uint32_t)(s0 % 6u));
      const uint4 v = q[ch]; acc ^= v.x ^ v.w;
    } else { const uint4 v0 = p[0], v1 = p[1], v2 = p[2], v3 = p[3 + (s & 3) + ((s >> 2) & 1)]; acc ^= v0.x ^ v1.y ^ v2.z ^ v3.w; }
  }
  if (acc == 0x12345678u) out[0] = acc;
}

int main(int argc, char **argv) {
  if (argc < 6) { printf("usage: randreach <alloc GiB> <loads per lane> <mode 1|4|14> <blocks per CU> <window GiB>...\n"); return 2; }
  const size_t alloc = (size_t)atol(argv[1]) << 30;
  const int iters = atoi(argv[2]), mode = atoi(argv[3]), bpc = atoi(argv[4]);
  uint4 *tab; uint32_t *out;
  if (hipMalloc(&tab, alloc) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) { printf("alloc of %zu GiB failed\n", alloc >> 30); return 1; }
  hipMemset(tab, 1, alloc);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256 * bpc;
  for (int a = 5; a < argc; a++) {
    const double wg = atof(argv[a]);
    size_t win = wg <= 0 ? ((size_t)256 << 20) : (size_t)(wg * (double)(1ull << 30));
    if (win > alloc) win = alloc;
    if (mode == 14 && 2 * win > alloc) win = alloc / 2;
    const uint64_t nlines = win / 128, far0 = (alloc - win) / 128;
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0, 0);
      if (mode == 1) k_reach<1><<<blocks, 256>>>(tab, nlines, far0, iters, out);
      else if (mode == 2) k_reach<2><<<blocks, 256>>>(tab, nlines, far0, iters, out);
      else if (mode == 3) k_reach<3><<<blocks, 256>>>(tab, nlines, far0, iters, out);
      else if (mode == 41) k_reach<41><<<blocks, 256>>>(tab, nlines, far0, iters, out);
      else if (mode == 42) k_reach<42><<<blocks, 256>>>(tab, nlines, far0, iters, out);
      else if (mode == 14) k_reach<14><<<blocks, 256>>>(tab, nlines, far0, iters, out);
      else k_reach<4><<<blocks, 256>>>(tab, nlines, far0, iters, out);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    const double q = (double)blocks * 256 * iters / ((mode == 41 || mode == 42) ? 4.0 : 1.0);    // (quad modes: a line per quad and load)
    printf("alloc %zu GiB mode %d window %8.2f GiB lanes %d loads/lane %d: %8.3f ms  %6.2f G lines/s  %7.1f GB/s at 128 B/line\n",
           alloc >> 30, mode, (double)win / (double)(1ull << 30), blocks * 256, iters, best, q / best * 1e-6, q * 128 / best * 1e-6);
    fflush(stdout);
  }
  return 0;
}
