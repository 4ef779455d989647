// xsg_diag.hip -- libxsg_diag.so: read-only probes of the HBM stream, NOT part of the product library.
//
// Kernels with k_scan's load shape (LOADS x global_load_dwordx4 per lane, 1 KiB per wave-instruction, 4 waves
// per workgroup) and no work on the bytes: what this device delivers for that access pattern, the yardstick next
// to which k_scan's rate is read (scripts/read_variants.py, scripts/read_exp.py).  They take
// raw device pointers and sizes -- nothing of libxsg's objects -- so the product library carries no code that
// is not on the search path.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int kBlock = 256;
constexpr int kWaves = 4;
constexpr uint32_t kUnit = 16;
constexpr uint32_t kWaveLoad = 64 * kUnit;

thread_local char g_err[256] = "";

// VARIANT 0: tile = block index (k_scan's mapping)      1: + non-temporal loads
//         2: XCD-contiguous (blocks b, b+8, ... walk one eighth of the span)
//         3: waves of a block interleave their KiBs instead of owning contiguous spans
//         4..6: variant 1 with other cache-policy bits on the load: "sc1 nt", "sc0 sc1 nt", "sc0 nt" (inline asm)
//         7..8: the same through buffer loads the compiler counts: raw_buffer_load_b128 with aux 18 (sc1 nt), 19 (sc0 sc1 nt)
template <int LOADS, int VARIANT>
__global__ __launch_bounds__(kBlock) void k_read_ceiling(const uint8_t* base, uint64_t ntiles, uint32_t* sink) {
  uint64_t tile = (uint64_t)blockIdx.x + (uint64_t)blockIdx.y * gridDim.x;
  if (tile >= ntiles) return;
  if (VARIANT == 2) {
    const uint64_t per = ntiles / 8;
    if (tile < per * 8) tile = (tile % 8) * per + tile / 8;
  }
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint8_t* tbase = base + tile * (uint64_t)(kWaveLoad * LOADS * kWaves);
  uint4 v[LOADS];
#pragma unroll
  for (int j = 0; j < LOADS; ++j) {
    const uint64_t off = VARIANT == 3 ? ((uint64_t)(j * kWaves + wave) * kWaveLoad + (uint64_t)lane * kUnit)
                                      : ((uint64_t)wave * (kWaveLoad * LOADS) + (uint64_t)j * kWaveLoad + (uint64_t)lane * kUnit);
    if (VARIANT == 1) {
      const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(tbase + off));
      v[j] = make_uint4(t.x, t.y, t.z, t.w);
    } else if (VARIANT >= 7) {
      // one descriptor per wave (wave-uniform base), the lane's part in voffset
      const uint8_t* wb = tbase + (VARIANT == 3 ? 0 : (uint64_t)__builtin_amdgcn_readfirstlane(wave) * (kWaveLoad * LOADS));
      const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)wb);
      const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)wb >> 32));
      auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(((uintptr_t)hi << 32) | lo), 0, 0x7fffffff, 0x00020000);
      const uint32_t voff = (uint32_t)j * kWaveLoad + lane * kUnit;
      const u32x4 t = VARIANT == 7 ? __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 18)
                                   : __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 19);
      v[j] = make_uint4(t.x, t.y, t.z, t.w);
    } else if (VARIANT >= 4) {
      u32x4 t;
      const uint8_t* p = tbase + off;
      if (VARIANT == 4) asm volatile("global_load_dwordx4 %0, %1, off sc1 nt" : "=v"(t) : "v"(p) : "memory");
      if (VARIANT == 5) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt" : "=v"(t) : "v"(p) : "memory");
      if (VARIANT == 6) asm volatile("global_load_dwordx4 %0, %1, off sc0 nt" : "=v"(t) : "v"(p) : "memory");
      v[j] = make_uint4(t.x, t.y, t.z, t.w);
    } else {
      v[j] = *reinterpret_cast<const uint4*>(tbase + off);
    }
  }
  if (VARIANT >= 4 && VARIANT <= 6) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the compiler does not count asm loads
  uint32_t x = 0;
#pragma unroll
  for (int j = 0; j < LOADS; ++j) x ^= v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
  if (x == 0xdeadbeefu) sink[0] = x;  // keeps the loads alive; practically never true
}

// Pure nt read of a flat span with the knobs of the burst experiments: LOADS units per lane, any workgroup size,
// wave stagger (wave w sleeps w * stagger * 64 clocks first), a pause between the loads of one wave.
template <int LOADS>
__global__ void k_read_exp(const uint8_t* base, uint64_t ntiles, uint32_t stagger, uint32_t gap, uint32_t* sink) {
  const uint64_t tile = (uint64_t)blockIdx.x + (uint64_t)blockIdx.y * gridDim.x;
  if (tile >= ntiles) return;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t nwaves = blockDim.x >> 6;
  for (uint32_t i = 0; i < stagger * wave; ++i) __builtin_amdgcn_s_sleep(1);
  const uint8_t* p = base + tile * (uint64_t)(kWaveLoad * LOADS) * nwaves + (uint64_t)wave * (kWaveLoad * LOADS) +
                     (uint64_t)lane * kUnit;
  uint4 v[LOADS];
#pragma unroll
  for (int j = 0; j < LOADS; ++j) {
    const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p + (uint64_t)j * kWaveLoad));
    v[j] = make_uint4(t.x, t.y, t.z, t.w);
    if (gap) {
      __builtin_amdgcn_sched_barrier(0);
      for (uint32_t i = 0; i < gap; ++i) __builtin_amdgcn_s_sleep(1);
    }
  }
  uint32_t x = 0;
#pragma unroll
  for (int j = 0; j < LOADS; ++j) x ^= v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
  if (x == 0xdeadbeefu) sink[0] = x;
}

dim3 grid_for(uint64_t ntiles) {
  const uint64_t maxx = 1u << 30;
  return ntiles <= maxx ? dim3((unsigned)ntiles) : dim3((unsigned)maxx, (unsigned)((ntiles + maxx - 1) / maxx));
}

template <int LOADS>
void launch_rc(int variant, dim3 grid, hipStream_t s, const uint8_t* base, uint64_t ntiles, uint32_t* sink) {
  switch (variant) {
    case 1: hipLaunchKernelGGL((k_read_ceiling<LOADS, 1>), grid, dim3(kBlock), 0, s, base, ntiles, sink); break;
    case 2: hipLaunchKernelGGL((k_read_ceiling<LOADS, 2>), grid, dim3(kBlock), 0, s, base, ntiles, sink); break;
    case 3: hipLaunchKernelGGL((k_read_ceiling<LOADS, 3>), grid, dim3(kBlock), 0, s, base, ntiles, sink); break;
    case 4: hipLaunchKernelGGL((k_read_ceiling<LOADS, 4>), grid, dim3(kBlock), 0, s, base, ntiles, sink); break;
    case 5: hipLaunchKernelGGL((k_read_ceiling<LOADS, 5>), grid, dim3(kBlock), 0, s, base, ntiles, sink); break;
    case 6: hipLaunchKernelGGL((k_read_ceiling<LOADS, 6>), grid, dim3(kBlock), 0, s, base, ntiles, sink); break;
    case 7: hipLaunchKernelGGL((k_read_ceiling<LOADS, 7>), grid, dim3(kBlock), 0, s, base, ntiles, sink); break;
    case 8: hipLaunchKernelGGL((k_read_ceiling<LOADS, 8>), grid, dim3(kBlock), 0, s, base, ntiles, sink); break;
    default: hipLaunchKernelGGL((k_read_ceiling<LOADS, 0>), grid, dim3(kBlock), 0, s, base, ntiles, sink); break;
  }
}

int fail(const char* what, hipError_t e) {
  snprintf(g_err, sizeof g_err, "%s: %s", what, hipGetErrorString(e));
  return -1;
}

template <typename Launch>
int timed(int iters, float* avg_ms, Launch&& launch) {
  if (iters <= 0 || !avg_ms) {
    snprintf(g_err, sizeof g_err, "bad iters/avg_ms");
    return -1;
  }
  hipEvent_t e0, e1;
  hipError_t e;
  if ((e = hipEventCreate(&e0)) != hipSuccess || (e = hipEventCreate(&e1)) != hipSuccess) return fail("hipEventCreate", e);
  launch();  // warm-up
  (void)hipEventRecord(e0, nullptr);
  for (int i = 0; i < iters; ++i) launch();
  (void)hipEventRecord(e1, nullptr);
  if ((e = hipEventSynchronize(e1)) != hipSuccess) return fail("hipEventSynchronize", e);
  if ((e = hipGetLastError()) != hipSuccess) return fail("kernel launch", e);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *avg_ms = ms / (float)iters;
  return 0;
}

}  // namespace

extern "C" const char* xsg_diag_last_error(void) { return g_err; }

// Reads bytes [0, floor(bytes / tile_bytes) * tile_bytes) of d_base (device memory, 16-byte aligned) `iters` times on
// the NULL stream; tile_bytes in {4096, 8192, 16384, 32768}; d_sink: 4 writable device bytes.  0 / -1.
extern "C" int xsg_diag_read(const void* d_base, uint64_t bytes, uint32_t tile_bytes, int variant, int iters, void* d_sink,
                             float* avg_ms, uint64_t* bytes_per_launch) {
  const uint64_t ntiles = tile_bytes ? bytes / tile_bytes : 0;
  if (!d_base || !d_sink || !ntiles) {
    snprintf(g_err, sizeof g_err, "bad argument");
    return -1;
  }
  const uint8_t* base = static_cast<const uint8_t*>(d_base);
  uint32_t* sink = static_cast<uint32_t*>(d_sink);
  const dim3 grid = grid_for(ntiles);
  if (tile_bytes != 4096u && tile_bytes != 8192u && tile_bytes != 16384u && tile_bytes != 32768u) {
    snprintf(g_err, sizeof g_err, "tile_bytes must be 4096/8192/16384/32768");
    return -1;
  }
  if (bytes_per_launch) *bytes_per_launch = ntiles * tile_bytes;
  return timed(iters, avg_ms, [&] {
    switch (tile_bytes) {
      case 4096u: launch_rc<1>(variant, grid, nullptr, base, ntiles, sink); break;
      case 8192u: launch_rc<2>(variant, grid, nullptr, base, ntiles, sink); break;
      case 16384u: launch_rc<4>(variant, grid, nullptr, base, ntiles, sink); break;
      default: launch_rc<8>(variant, grid, nullptr, base, ntiles, sink); break;
    }
  });
}

extern "C" int xsg_diag_read_exp(const void* d_base, uint64_t bytes, int loads, int block, uint32_t stagger, uint32_t gap,
                                 int iters, void* d_sink, float* avg_ms, uint64_t* bytes_per_launch) {
  if (!d_base || !d_sink || block % 64 || block < 64 || block > 1024 || (loads != 1 && loads != 2 && loads != 4 && loads != 8)) {
    snprintf(g_err, sizeof g_err, "bad argument");
    return -1;
  }
  const uint64_t tile_bytes = (uint64_t)kWaveLoad * (uint64_t)loads * (uint64_t)(block / 64);
  const uint64_t ntiles = bytes / tile_bytes;
  if (!ntiles) {
    snprintf(g_err, sizeof g_err, "span shorter than one tile");
    return -1;
  }
  const uint8_t* base = static_cast<const uint8_t*>(d_base);
  uint32_t* sink = static_cast<uint32_t*>(d_sink);
  const dim3 grid = grid_for(ntiles);
  if (bytes_per_launch) *bytes_per_launch = ntiles * tile_bytes;
  return timed(iters, avg_ms, [&] {
    switch (loads) {
      case 1: hipLaunchKernelGGL((k_read_exp<1>), grid, dim3(block), 0, nullptr, base, ntiles, stagger, gap, sink); break;
      case 2: hipLaunchKernelGGL((k_read_exp<2>), grid, dim3(block), 0, nullptr, base, ntiles, stagger, gap, sink); break;
      case 4: hipLaunchKernelGGL((k_read_exp<4>), grid, dim3(block), 0, nullptr, base, ntiles, stagger, gap, sink); break;
      default: hipLaunchKernelGGL((k_read_exp<8>), grid, dim3(block), 0, nullptr, base, ntiles, stagger, gap, sink); break;
    }
  });
}
