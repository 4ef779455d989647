// Can the reader skip its memcpy?  mmap a page-cache/tmpfs resident file, pin the mapping with
// hipHostRegister and let the copy engine read the page-cache pages directly.
// Prints: registration rate, H2D rate from the registered mapping, H2D rate from a plain
// (unregistered) mapping, and for scale pread-into-pinned + H2D.
// build: hipcc --offload-arch=gfx950 -O2 mmap_h2d.hip -o mmap_h2d ; run: ./mmap_h2d /dev/shm/file GiB
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e = (x);                                                             \
    if (e != hipSuccess) {                                                          \
      printf("%s -> %s\n", #x, hipGetErrorString(e));                               \
      fflush(stdout);                                                               \
    }                                                                               \
  } while (0)

int main(int argc, char** argv) {
  const char* path = argv[1];
  const size_t size = (size_t)(atof(argv[2]) * (1ull << 30));
  {  // make the file
    int fd = open(path, O_CREAT | O_TRUNC | O_WRONLY, 0600);
    std::vector<char> buf(64 << 20, 'x');
    for (size_t i = 0; i < buf.size(); i += 61) buf[i] = '\n';
    for (size_t w = 0; w < size; w += buf.size()) (void)!write(fd, buf.data(), buf.size());
    close(fd);
  }
  int fd = open(path, O_RDONLY);
  void* dev = nullptr;
  CK(hipMalloc(&dev, size));
  hipStream_t s;
  CK(hipStreamCreate(&s));
  const size_t chunk = 16u << 20;

  for (int variant = 0; variant < 3; ++variant) {
    const int prot = variant == 2 ? (PROT_READ | PROT_WRITE) : PROT_READ;
    const int flags = variant == 2 ? MAP_PRIVATE : MAP_SHARED;
    void* m = mmap(nullptr, size, prot, flags | MAP_POPULATE, fd, 0);
    if (m == MAP_FAILED) {
      printf("variant %d: mmap failed\n", variant);
      continue;
    }
    if (variant == 0) {  // unregistered mapping: the runtime stages pageable memory itself
      double t0 = now();
      for (size_t o = 0; o < size; o += chunk) CK(hipMemcpyAsync((char*)dev + o, (char*)m + o, chunk, hipMemcpyHostToDevice, s));
      CK(hipStreamSynchronize(s));
      double dt = now() - t0;
      printf("pageable mmap -> H2D: %.2f GiB/s\n", size / dt / (1 << 30));
    } else {
      double t0 = now();
      hipError_t e = hipHostRegister(m, size, variant == 1 ? hipHostRegisterDefault : hipHostRegisterDefault);
      double treg = now() - t0;
      printf("variant %d (%s): hipHostRegister -> %s, %.3f s = %.2f GiB/s\n", variant,
             variant == 1 ? "PROT_READ MAP_SHARED" : "PROT_RW MAP_PRIVATE", hipGetErrorString(e), treg, size / treg / (1 << 30));
      if (e == hipSuccess) {
        for (int rep = 0; rep < 2; ++rep) {
          t0 = now();
          for (size_t o = 0; o < size; o += chunk)
            CK(hipMemcpyAsync((char*)dev + o, (char*)m + o, chunk, hipMemcpyHostToDevice, s));
          CK(hipStreamSynchronize(s));
          double dt = now() - t0;
          printf("  registered mmap -> H2D: %.2f GiB/s\n", size / dt / (1 << 30));
        }
        // per-chunk registration (what a streaming reader would do): register 16 MiB, copy, unregister
        CK(hipHostUnregister(m));
        t0 = now();
        const size_t n = size / chunk < 64 ? size / chunk : 64;
        for (size_t k = 0; k < n; ++k) {
          CK(hipHostRegister((char*)m + k * chunk, chunk, hipHostRegisterDefault));
          CK(hipMemcpyAsync((char*)dev + k * chunk, (char*)m + k * chunk, chunk, hipMemcpyHostToDevice, s));
          CK(hipStreamSynchronize(s));
          CK(hipHostUnregister((char*)m + k * chunk));
        }
        double dt = now() - t0;
        printf("  per-chunk register+copy+unregister: %.2f GiB/s\n", n * chunk / dt / (1 << 30));
      } else {
        (void)hipGetLastError();
      }
    }
    munmap(m, size);
    fflush(stdout);
  }
  for (int populate = 0; populate < 2; ++populate) {  // the streaming pattern on a fresh mapping
    for (size_t seg : {chunk, 4 * chunk}) {
      void* m = mmap(nullptr, size, PROT_READ, MAP_SHARED | (populate ? MAP_POPULATE : 0), fd, 0);
      double t0 = now();
      double treg = 0, tun = 0;
      for (size_t o = 0; o < size; o += seg) {
        double a = now();
        CK(hipHostRegister((char*)m + o, seg, hipHostRegisterDefault));
        treg += now() - a;
        for (size_t q = 0; q < seg; q += chunk)
          CK(hipMemcpyAsync((char*)dev + o + q, (char*)m + o + q, chunk, hipMemcpyHostToDevice, s));
        CK(hipStreamSynchronize(s));
        a = now();
        CK(hipHostUnregister((char*)m + o));
        tun += now() - a;
      }
      double dt = now() - t0;
      printf("fresh mmap%s, segments of %zu MiB: register+copy+unregister serial %.2f GiB/s (register %.3f s, unregister %.3f s of %.3f s)\n",
             populate ? " (populated)" : "", seg >> 20, size / dt / (1 << 30), treg, tun, dt);
      munmap(m, size);
      fflush(stdout);
    }
  }
  for (int T : {1, 2, 4, 8}) {  // T threads, each: register its next 16 MiB chunk, copy on its own stream, unregister
    void* m = mmap(nullptr, size, PROT_READ, MAP_SHARED, fd, 0);
    std::atomic<size_t> next{0};
    const size_t nchunks = size / chunk;
    double t0 = now();
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
      th.emplace_back([&] {
        CK(hipSetDevice(0));
        hipStream_t st;
        CK(hipStreamCreate(&st));
        for (;;) {
          const size_t k = next.fetch_add(1);
          if (k >= nchunks) break;
          CK(hipHostRegister((char*)m + k * chunk, chunk, hipHostRegisterPortable));
          CK(hipMemcpyAsync((char*)dev + k * chunk, (char*)m + k * chunk, chunk, hipMemcpyHostToDevice, st));
          CK(hipStreamSynchronize(st));
          CK(hipHostUnregister((char*)m + k * chunk));
        }
        CK(hipStreamDestroy(st));
      });
    for (auto& x : th) x.join();
    double dt = now() - t0;
    printf("fresh mmap, %d threads register+copy+unregister: %.2f GiB/s\n", T, size / dt / (1 << 30));
    munmap(m, size);
    fflush(stdout);
  }
  for (int T : {1, 2, 4, 8}) {  // the same with pread into a per-thread pinned buffer (what the pipeline does today)
    std::atomic<size_t> next{0};
    const size_t nchunks = size / chunk;
    double t0 = now();
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
      th.emplace_back([&] {
        CK(hipSetDevice(0));
        hipStream_t st;
        CK(hipStreamCreate(&st));
        void* pin = nullptr;
        CK(hipHostMalloc(&pin, chunk, hipHostMallocDefault));
        for (;;) {
          const size_t k = next.fetch_add(1);
          if (k >= nchunks) break;
          (void)!pread(fd, pin, chunk, (off_t)(k * chunk));
          CK(hipMemcpyAsync((char*)dev + k * chunk, pin, chunk, hipMemcpyHostToDevice, st));
          CK(hipStreamSynchronize(st));
        }
        CK(hipHostFree(pin));
        CK(hipStreamDestroy(st));
      });
    for (auto& x : th) x.join();
    double dt = now() - t0;
    printf("%d threads pread->pinned->H2D (serial per thread): %.2f GiB/s\n", T, size / dt / (1 << 30));
    fflush(stdout);
  }
  {  // pread into a pinned buffer, then H2D (what the reader threads do now), one thread
    void* pin = nullptr;
    CK(hipHostMalloc(&pin, 2 * chunk, hipHostMallocDefault));
    double t0 = now();
    int b = 0;
    for (size_t o = 0; o < size; o += chunk, b ^= 1) {
      (void)!pread(fd, (char*)pin + b * chunk, chunk, (off_t)o);
      CK(hipMemcpyAsync((char*)dev + o, (char*)pin + b * chunk, chunk, hipMemcpyHostToDevice, s));
      if (o >= chunk) { /* previous copy of this half finished long ago at these rates; sync to be safe */ }
      CK(hipStreamSynchronize(s));
    }
    double dt = now() - t0;
    printf("1 thread pread -> pinned -> H2D (serial): %.2f GiB/s\n", size / dt / (1 << 30));
  }
  unlink(path);
  return 0;
}
