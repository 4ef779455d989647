// What does a fresh process pay before its first kernel has run?  The floor under every CLI search (xsgrep, the README's
// my_grep): HIP start-up, one stream, one empty kernel, one pinned and one device allocation -- nothing of ours.
// build: hipcc --offload-arch=gfx950 -O2 hip_start.hip -o build/hip_start ; prints milliseconds since main().
#include <hip/hip_runtime.h>
#include <stdio.h>

#include <chrono>

static double ms_since(std::chrono::steady_clock::time_point t0) {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
__global__ void k_empty() {}

int main() {
  const auto t0 = std::chrono::steady_clock::now();
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  printf("hipGetDeviceCount -> %d (%s)   %9.3f ms\n", n, hipGetErrorString(e), ms_since(t0));
  e = hipSetDevice(0);
  printf("hipSetDevice                      %9.3f ms\n", ms_since(t0));
  hipDeviceProp_t p;
  e = hipGetDeviceProperties(&p, 0);
  printf("hipGetDeviceProperties            %9.3f ms\n", ms_since(t0));
  hipStream_t s;
  e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  printf("hipStreamCreate                   %9.3f ms\n", ms_since(t0));
  hipLaunchKernelGGL(k_empty, dim3(1), dim3(1), 0, s);
  e = hipStreamSynchronize(s);
  printf("first kernel done                 %9.3f ms\n", ms_since(t0));
  void *h = nullptr, *d = nullptr;
  e = hipHostMalloc(&h, 16u << 20, hipHostMallocDefault);
  printf("hipHostMalloc 16 MiB              %9.3f ms\n", ms_since(t0));
  e = hipHostMalloc(&h, 16u << 20, hipHostMallocDefault);
  printf("hipHostMalloc 16 MiB (second)     %9.3f ms\n", ms_since(t0));
  e = hipMalloc(&d, 16u << 20);
  printf("hipMalloc 16 MiB                  %9.3f ms\n", ms_since(t0));
  e = hipMemcpyAsync(d, h, 16u << 20, hipMemcpyHostToDevice, s);
  e = hipStreamSynchronize(s);
  printf("first H2D 16 MiB done             %9.3f ms\n", ms_since(t0));
  e = hipMemcpyAsync(d, h, 16u << 20, hipMemcpyHostToDevice, s);
  e = hipStreamSynchronize(s);
  printf("second H2D 16 MiB done            %9.3f ms\n", ms_since(t0));
  (void)e;
  return 0;
}
