// randbench.hip — calibration microbenchmark for the roofline of the search kernels:
// how many random 128-byte index blocks per second can one MI355X fetch, and what do the
// FETCH_SIZE / TCC_EA0_RDREQ counters report per block?  Each lane issues independent
// (hash-addressed) loads in the shape of one rank query of kj_core.h: MODE 1 = one 16-B load,
// MODE 4 = four 16-B loads spread over both 64-B halves of the block (RankBlock64 pattern),
// MODE 8 = the whole 128-B block.   Usage: randbench <table MiB> <loads per lane> <mode>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

__device__ __forceinline__ uint64_t mix(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return x;
}

template <int MODE>
__global__ void __launch_bounds__(256) k_rand(const uint4 *__restrict__ tab, uint64_t nblocks, int iters,
                                              uint32_t *out) {
  uint64_t s = mix(blockIdx.x * 256ull + threadIdx.x + 1);
  uint32_t acc = 0;
  for (int i = 0; i < iters; ++i) {
    s = mix(s + 0x9e3779b97f4a7c15ULL);
    const uint64_t b = (uint64_t)(((unsigned __int128)s * nblocks) >> 64);
    const uint4 *p = tab + b * 8;
    if (MODE == 1) {
      uint4 v = p[s & 7]; acc ^= v.x ^ v.w;
    } else if (MODE == 4) {
      uint4 v0 = p[0], v1 = p[1], v2 = p[2], v3 = p[3 + (s & 3) + ((s >> 2) & 1)];
      acc ^= v0.x ^ v1.y ^ v2.z ^ v3.w;
    } else if (MODE == 2) {   // both loads inside one 64-byte half
      uint4 v0 = p[(s & 4) + 0], v1 = p[(s & 4) + 1 + (s & 1)];
      acc ^= v0.x ^ v1.y;
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) { uint4 v = p[k]; acc ^= v.x + v.y; }
    }
  }
  if (acc == 0x12345678u) out[0] = acc;
}

int main(int argc, char **argv) {
  size_t mib = argc > 1 ? atol(argv[1]) : 1024;
  int iters = argc > 2 ? atoi(argv[2]) : 256;
  int mode = argc > 3 ? atoi(argv[3]) : 4;
  int bpc = argc > 4 ? atoi(argv[4]) : 8;
  size_t bytes = mib << 20;
  uint4 *tab; uint32_t *out;
  if (hipMalloc(&tab, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(tab, 1, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256 * bpc;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0, 0);
    if (mode == 1) k_rand<1><<<blocks, 256>>>(tab, bytes / 128, iters, out);
    else if (mode == 2) k_rand<2><<<blocks, 256>>>(tab, bytes / 128, iters, out);
    else if (mode == 4) k_rand<4><<<blocks, 256>>>(tab, bytes / 128, iters, out);
    else k_rand<8><<<blocks, 256>>>(tab, bytes / 128, iters, out);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double q = (double)blocks * 256 * iters;
    printf("mode %d table %zu MiB lanes %d iters %d: %.3f ms, %.2f G blocks/s, %.1f GB/s at 128 B/block\n", mode, mib,
           blocks * 256, iters, ms, q / ms * 1e-6, q * 128 / ms * 1e-6);
  }
  return 0;
}
