// Development aid: does hipExtAnyOrderLaunch let a kernel start while the previous kernel of the SAME stream still runs
// (AQL barrier bit cleared)?  Two spinning kernels, in-kernel wall clocks.
//   hipcc --offload-arch=gfx950 -O2 tools/anyorder.hip -o tools/anyorder && tools/anyorder
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <chrono>

__global__ void spin(unsigned long long *out, unsigned long long ticks) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(10);
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    out[0] = t0;
    out[1] = wall_clock64();
  }
}

struct BigArg { double a[88]; };
__global__ void spin_big(unsigned long long *out, unsigned long long ticks, BigArg b) {
  if (ticks == 12345ull && threadIdx.x == 0) out[0] = (unsigned long long)b.a[3];
}

int main() {
  unsigned long long *d = nullptr, h[8];
  hipMalloc(&d, 64);
  hipStream_t s;
  hipStreamCreate(&s);
  for (int flags = 0; flags < 2; flags++) {
    for (int rep = 0; rep < 3; rep++) {
      hipMemsetAsync(d, 0, 64, s);
      hipStreamSynchronize(s);
      hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, d, 5000ull);       // 50 us at 100 MHz
      hipExtLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, nullptr, nullptr, flags, d + 2, 5000ull);
      hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, d + 4, 1000ull);   // ordinary launch behind both
      hipStreamSynchronize(s);
      hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
      printf("flags %d: A %.1f..%.1f us  B %.1f..%.1f us  C %.1f..%.1f us\n", flags, 0.0, (h[1] - h[0]) * 0.01, (h[2] - h[0]) * 0.01,
             (h[3] - h[0]) * 0.01, (h[4] - h[0]) * 0.01, (h[5] - h[0]) * 0.01);
    }
  }
  // host cost of a launch call (no arguments / ~700 bytes of arguments), queue kept short
  {
    BigArg big{};
    const int N = 2000;
    for (int variant = 0; variant < 2; variant++) {
      hipStreamSynchronize(s);
      double tot = 0;
      for (int i = 0; i < N; i++) {
        if ((i & 15) == 15) hipStreamSynchronize(s);
        const auto t0 = std::chrono::steady_clock::now();
        if (variant == 0) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, d + 6, 0ull);
        else hipLaunchKernelGGL(spin_big, dim3(1), dim3(64), 0, s, d + 6, 0ull, big);
        tot += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      }
      hipStreamSynchronize(s);
      printf("host time per launch call (%s): %.2f us\n", variant ? "704 B of arguments" : "16 B of arguments", tot / N);
    }
  }
  // two streams: A on s, B on s2, C on s behind an event recorded on s2 after B
  hipStream_t s2;
  hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  hipEvent_t ev;
  hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  for (int rep = 0; rep < 4; rep++) {
    hipMemsetAsync(d, 0, 64, s);
    hipStreamSynchronize(s);
    hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, d, 5000ull);
    hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s2, d + 2, 3000ull);
    hipEventRecord(ev, s2);
    hipStreamWaitEvent(s, ev, 0);
    hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, d + 4, 1000ull);
    hipStreamSynchronize(s);
    hipStreamSynchronize(s2);
    hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
    printf("two streams: A %.1f..%.1f us  B %.1f..%.1f us  C (waits for both) %.1f..%.1f us\n", 0.0, (h[1] - h[0]) * 0.01,
           ((long long)h[2] - (long long)h[0]) * 0.01, ((long long)h[3] - (long long)h[0]) * 0.01, (h[4] - h[0]) * 0.01, (h[5] - h[0]) * 0.01);
  }
  return 0;
}
