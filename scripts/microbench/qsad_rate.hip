// Microbenchmark + semantics check of v_qsad_pk_u16_u8 on gfx950: is the packed
// sliding SAD cheap enough to replace 3 x v_alignbyte + 8 x v_cmp per 4 positions?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_qsad(const uint64_t* in, uint64_t* out, int n, uint32_t ref) {
  uint64_t a = in[threadIdx.x & 63], b = a ^ 0x0101, c = a ^ 0x020002, d = a ^ 0x3000003;
  uint64_t r0 = 0, r1 = 0, r2 = 0, r3 = 0;
  for (int i = 0; i < n; ++i) {
    r0 = __builtin_amdgcn_qsad_pk_u16_u8(a, ref, r0);
    r1 = __builtin_amdgcn_qsad_pk_u16_u8(b, ref, r1);
    r2 = __builtin_amdgcn_qsad_pk_u16_u8(c, ref, r2);
    r3 = __builtin_amdgcn_qsad_pk_u16_u8(d, ref, r3);
  }
  out[threadIdx.x + blockIdx.x * blockDim.x] = r0 ^ r1 ^ r2 ^ r3;
}
__global__ void k_align(const uint64_t* in, uint64_t* out, int n, uint32_t ref) {
  uint32_t a = (uint32_t)in[threadIdx.x & 63], b = a ^ 0x0101, c = a ^ 0x020002, d = a ^ 0x3000003;
  for (int i = 0; i < n; ++i) {
    a = __builtin_amdgcn_alignbyte(b, a, 1) + ref;
    b = __builtin_amdgcn_alignbyte(c, b, 2) + ref;
    c = __builtin_amdgcn_alignbyte(d, c, 3) + ref;
    d = __builtin_amdgcn_alignbyte(a, d, 1) + ref;
  }
  out[threadIdx.x + blockIdx.x * blockDim.x] = a ^ b ^ c ^ d;
}
__global__ void k_sem(const uint64_t* in, const uint32_t* refs, uint64_t* out, int n) {
  int i = threadIdx.x + blockIdx.x * blockDim.x;
  if (i < n) out[i] = __builtin_amdgcn_qsad_pk_u16_u8(in[i], refs[i], 0x0001000200030004ull);
}
int main() {
  const int nblk = 256 * 8, nthr = 256, iters = 4096;
  uint64_t *din, *dout;
  CK(hipMalloc(&din, 64 * 8));
  CK(hipMalloc(&dout, (size_t)nblk * nthr * 8));
  std::vector<uint64_t> h(64);
  for (int i = 0; i < 64; ++i) h[i] = 0x0123456789abcdefull * (i + 1);
  CK(hipMemcpy(din, h.data(), 64 * 8, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int which = 0; which < 2; ++which) {
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0));
      if (which == 0) hipLaunchKernelGGL(k_qsad, dim3(nblk), dim3(nthr), 0, 0, din, dout, iters, 0x6b636f6cu);
      else hipLaunchKernelGGL(k_align, dim3(nblk), dim3(nthr), 0, 0, din, dout, iters, 0x6b636f6cu);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      // instructions per SIMD: nblk*4 waves / (256 CUs * 4 SIMDs) * iters * (4 qsad | 8 valu)
      const double waves_per_simd = (double)nblk * 4 / (256.0 * 4);
      const double instr = waves_per_simd * iters * (which == 0 ? 4 : 8);
      printf("%s rep %d: %.3f ms, %.2f ns per wave-instruction per SIMD\n", which == 0 ? "v_qsad_pk_u16_u8" : "v_alignbyte+v_add", rep, ms,
             ms * 1e6 / instr);
    }
  }
  // semantics: D[16k+15:16k] = SAD(bytes k..k+3 of S0, bytes of S1) + S2[16k+15:16k]
  const int n = 4096;
  std::vector<uint64_t> in(n), out(n);
  std::vector<uint32_t> refs(n);
  srand(1);
  for (int i = 0; i < n; ++i) {
    in[i] = ((uint64_t)rand() << 33) ^ ((uint64_t)rand() << 11) ^ rand();
    refs[i] = (i % 3 == 0) ? (uint32_t)(in[i] >> (8 * (i % 4))) : (uint32_t)rand() * 2654435761u;
  }
  uint64_t* d_in; uint32_t* d_ref; uint64_t* d_out;
  CK(hipMalloc(&d_in, n * 8)); CK(hipMalloc(&d_ref, n * 4)); CK(hipMalloc(&d_out, n * 8));
  CK(hipMemcpy(d_in, in.data(), n * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_ref, refs.data(), n * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_sem, dim3(n / 256), dim3(256), 0, 0, d_in, d_ref, d_out, n);
  CK(hipMemcpy(out.data(), d_out, n * 8, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int i = 0; i < n; ++i) {
    uint64_t want = 0;
    for (int k = 0; k < 4; ++k) {
      uint32_t sad = 4 - k;  // S2 lanes: 0x0004 at lane 0 ... 0x0001 at lane 3
      for (int j = 0; j < 4; ++j) {
        int x = (int)((in[i] >> (8 * (k + j))) & 0xff), y = (int)((refs[i] >> (8 * j)) & 0xff);
        sad += (uint32_t)abs(x - y);
      }
      want |= (uint64_t)(sad & 0xffff) << (16 * k);
    }
    if (want != out[i]) { if (bad < 3) printf("sem mismatch %d: got %016llx want %016llx\n", i, (unsigned long long)out[i], (unsigned long long)want); ++bad; }
  }
  printf("semantics mismatches: %d of %d\n", bad, n);
  return bad != 0;
}
