// Microbenchmark (development aid, not part of the product): random aligned 128-byte line reads from a buffer of
// S bytes -- the access pattern of the coordinate-CV lookup on the replica.  Usage: randline <GiB> <lines> <mode>
//   mode 1: one lane per line (4 x 32-B loads per lane); mode 4: four lanes per line (one 32-B load each)
//   mode 2: two lines per lane, second line `far` bytes after the first (3-D: block(z), block(z+1))
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__device__ __forceinline__ unsigned long long mix(unsigned long long z) {
  z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}
template <int MODE>
__global__ void __launch_bounds__(256) k(const double4 *__restrict__ buf, unsigned long long nlines_buf, long long n, double *__restrict__ out, long long far_lines, int sorted) {
  const long long stride = (long long)gridDim.x * 256;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n * (MODE == 4 ? 4 : 1); t += stride) {
    const long long i = (MODE == 4) ? (t >> 2) : t;
    unsigned long long line = sorted ? ((unsigned long long)i * (nlines_buf / (unsigned long long)n)) : (mix((unsigned long long)i) % nlines_buf);
    if (MODE == 4) {
      const double4 v = buf[line * 4 + (t & 3)];
      double s = v.x + v.y + v.z + v.w;
      s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64);
      if ((t & 3) == 0) out[i] = s;
    } else {
      double s = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) { const double4 v = buf[line * 4 + k]; s += v.x + v.y + v.z + v.w; }
      if (MODE == 2) {
        unsigned long long l2 = line + (unsigned long long)far_lines; if (l2 >= nlines_buf) l2 -= nlines_buf;
#pragma unroll
        for (int k = 0; k < 4; k++) { const double4 v = buf[l2 * 4 + k]; s += v.x + v.y + v.z + v.w; }
      }
      out[i] = s;
    }
  }
}
int main(int argc, char **argv) {
  const double gib = argc > 1 ? atof(argv[1]) : 0.5;
  const long long n = argc > 2 ? atoll(argv[2]) : 262144;
  const int mode = argc > 3 ? atoi(argv[3]) : 1;
  const long long far = argc > 4 ? atoll(argv[4]) : 262144;
  const int sorted = argc > 5 ? atoi(argv[5]) : 0;
  const size_t bytes = (size_t)(gib * (1ull << 30)) & ~(size_t)127;
  double4 *buf; double *out;
  CK(hipMalloc((void **)&buf, bytes)); CK(hipMemset(buf, 0, bytes)); CK(hipMalloc((void **)&out, sizeof(double) * n));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const unsigned blocks = (unsigned)((n * (mode == 4 ? 4 : 1) + 255) / 256);
  float best = 1e9, tot = 0;
  for (int rep = 0; rep < 12; rep++) {
    CK(hipEventRecord(e0, 0));
    if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, buf, bytes / 128, n, out, far, sorted);
    else if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, buf, bytes / 128, n, out, far, sorted);
    else hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, buf, bytes / 128, n, out, far, sorted);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep >= 2) { tot += ms; if (ms < best) best = ms; }
  }
  const double lines = (double)n * (mode == 2 ? 2 : 1);
  printf("buf %.2f GiB  n %lld  mode %d far %lld sorted %d: avg %.1f us  best %.1f us  -> %.2f TB/s of 128-B lines (best)\n", gib, n, mode, far, sorted, tot / 10 * 1e3, best * 1e3,
         lines * 128 / (best * 1e-3) / 1e12);
  return 0;
}
