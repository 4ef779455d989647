// Issue cost of the VALU/SALU instructions k_scan's hot loop is made of (gfx950).
// 8 waves per SIMD, 8 independent instructions per loop iteration.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define BODY8(ASM)                                                                     \
  asm volatile(ASM "\n" ASM "\n" ASM "\n" ASM "\n" ASM "\n" ASM "\n" ASM "\n" ASM      \
               : "+v"(a), "+v"(b), "+v"(c), "+v"(d)                                    \
               : "s"(ref)                                                              \
               : "vcc", "s20", "s21", "scc");

#define KERNEL(NAME, ASM)                                                             \
  __global__ void NAME(const uint32_t* in, uint32_t* out, int n, uint32_t ref) {      \
    uint32_t a = in[threadIdx.x & 63], b = a ^ 0x0101, c = a ^ 0x020002, d = a + 77;  \
    for (int i = 0; i < n; ++i) { BODY8(ASM) }                                        \
    out[threadIdx.x + blockIdx.x * blockDim.x] = a ^ b ^ c ^ d;                       \
  }

KERNEL(k_add, "v_add_u32 %0, %1, %2")
KERNEL(k_xor, "v_xor_b32 %0, %1, %2")
KERNEL(k_alignbyte, "v_alignbyte_b32 %0, %1, %2, 1")
KERNEL(k_alignbit, "v_alignbit_b32 %0, %1, %2, 8")
KERNEL(k_perm, "v_perm_b32 %0, %1, %2, %3")
KERNEL(k_lshl_or, "v_lshl_or_b32 %0, %1, 8, %2")
KERNEL(k_or3, "v_or3_b32 %0, %1, %2, %3")
KERNEL(k_bfe, "v_bfe_u32 %0, %1, 8, 16")
KERNEL(k_cmp_vcc, "v_cmp_eq_u32 vcc, %4, %1")
KERNEL(k_cmp_sgpr, "v_cmp_eq_u32 s[20:21], %4, %1")
KERNEL(k_cndmask, "v_cndmask_b32 %0, %1, %2, vcc")
KERNEL(k_bcnt, "v_bcnt_u32_b32 %0, %1, %2")
KERNEL(k_dpp, "v_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf")
KERNEL(k_salu_or, "s_or_b64 s[20:21], s[20:21], vcc")

typedef void (*kern_t)(const uint32_t*, uint32_t*, int, uint32_t);

__global__ void k_lshr64_real(const uint32_t* in, uint32_t* out, int n, uint32_t ref) {
  uint64_t a = in[threadIdx.x & 63], b = a * 0x10001;
  for (int i = 0; i < n; ++i) {
    asm volatile("v_lshrrev_b64 %0, 8, %1\nv_lshrrev_b64 %1, 8, %0\nv_lshrrev_b64 %0, 8, %1\nv_lshrrev_b64 %1, 8, %0\n"
                 "v_lshrrev_b64 %0, 8, %1\nv_lshrrev_b64 %1, 8, %0\nv_lshrrev_b64 %0, 8, %1\nv_lshrrev_b64 %1, 8, %0"
                 : "+v"(a), "+v"(b));
  }
  out[threadIdx.x + blockIdx.x * blockDim.x] = (uint32_t)(a ^ b);
}

int main() {
  const int nblk = 256 * 8, nthr = 256, iters = 2048;
  uint32_t *din, *dout;
  CK(hipMalloc(&din, 64 * 4));
  CK(hipMalloc(&dout, (size_t)nblk * nthr * 4));
  CK(hipMemset(din, 1, 64 * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  struct { const char* name; kern_t k; } ks[] = {
      {"v_add_u32", k_add}, {"v_xor_b32", k_xor}, {"v_alignbyte_b32", k_alignbyte}, {"v_alignbit_b32", k_alignbit},
      {"v_perm_b32", k_perm}, {"v_lshl_or_b32", k_lshl_or}, {"v_or3_b32", k_or3}, {"v_bfe_u32", k_bfe},
      {"v_cmp_eq_u32 -> vcc", k_cmp_vcc}, {"v_cmp_eq_u32 -> sgpr", k_cmp_sgpr}, {"v_cndmask_b32", k_cndmask},
      {"v_bcnt_u32_b32", k_bcnt}, {"v_mov_b32_dpp wave_shl", k_dpp}, {"s_or_b64", k_salu_or},
      {"v_lshrrev_b64", k_lshr64_real}};
  for (auto& e : ks) {
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(e.k, dim3(nblk), dim3(nthr), 0, 0, din, dout, iters, 0x6b636f6cu);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    const double waves_per_simd = (double)nblk * 4 / (256.0 * 4);
    const double instr = waves_per_simd * iters * 8;
    printf("%-26s %.3f ms  %.2f ns per wave-instruction per SIMD\n", e.name, best, best * 1e6 / instr);
    fflush(stdout);
  }
  return 0;
}
