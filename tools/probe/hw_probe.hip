// Hardware-assumption probe for gfx950: checks the MFMA fragment maps, the
// ds_read_b64_tr_b16 transposed read and global_load_lds semantics the kernels rely on.
// Build: hipcc --offload-arch=gfx950 -O2 hw_probe.hip -o hw_probe ; run on an MI355X.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef unsigned short u16;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(2);} } while (0)

static inline u16 f2bf(float f) { unsigned u; memcpy(&u, &f, 4); return (u16)((u + 0x7FFF + ((u >> 16) & 1)) >> 16); }
static inline float bf2f(u16 h) { unsigned u = ((unsigned)h) << 16; float f; memcpy(&f, &u, 4); return f; }

// C[16x16] = A[16x32] * B[32x16]; A row-major [16][32], Bt row-major [16 cols][32 k]
__global__ void k_mfma16(const u16* A, const u16* Bt, float* C) {
  int l = threadIdx.x;
  bf16x8 a = *(const bf16x8*)(A + (l & 15) * 32 + 8 * (l >> 4));
  bf16x8 b = *(const bf16x8*)(Bt + (l & 15) * 32 + 8 * (l >> 4));
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; r++) C[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}
// C[32x32] = A[32x16] * B[16x32]; A [32][16], Bt [32 cols][16 k]
__global__ void k_mfma32(const u16* A, const u16* Bt, float* C) {
  int l = threadIdx.x;
  bf16x8 a = *(const bf16x8*)(A + (l & 31) * 16 + 8 * (l >> 5));
  bf16x8 b = *(const bf16x8*)(Bt + (l & 31) * 16 + 8 * (l >> 5));
  f32x16 c = {};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; r++) C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}
// tr read: LDS image M[16 rows][64 cols] u16 (row stride 64 elements). Each 16-lane group g reads the
// 4x16 block rows 4g..4g+3, cols 0..15. Expect lane i of group: {M[4g+0][i], M[4g+1][i], M[4g+2][i], M[4g+3][i]}.
__global__ void k_tr(const u16* in, u16* out) {
  __shared__ __attribute__((aligned(16))) u16 lds[16 * 64];
  for (int i = threadIdx.x; i < 16 * 64; i += 64) lds[i] = in[i];
  __syncthreads();
  int l = threadIdx.x, g = l >> 4, i = l & 15, q = i >> 2, p = i & 3;
  const u16* addr = lds + (4 * g + q) * 64 + 4 * p;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)addr);
  for (int e = 0; e < 4; e++) out[l * 4 + e] = (u16)v[e];
}
// global_load_lds 16B: lane i loads 16 B from src + perm(i)*16 and LDS gets it at base + i*16.
__global__ void k_glds(const unsigned* src, unsigned* out) {
  __shared__ __attribute__((aligned(16))) unsigned lds[4 * 256];
  int l = threadIdx.x;
  int w = l >> 6, lane = l & 63;
  int srcchunk = (lane ^ 5) + 64 * w;  // permuted source
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + srcchunk * 4),
                                   (__attribute__((address_space(3))) void*)(lds + w * 256), 16, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);  // drain everything
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int e = 0; e < 4; e++) out[l * 4 + e] = lds[l * 4 + e];
}
// raw_buffer_load_lds 16B with bounds: records beyond num_records must land as zeros
__global__ void k_blds(const unsigned* src, unsigned* out, int valid_bytes) {
  __shared__ __attribute__((aligned(16))) unsigned lds[4 * 256];
  int l = threadIdx.x;
  int w = l >> 6, lane = l & 63;
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, valid_bytes, 0x00020000);
  int srcchunk = (lane ^ 3) + 64 * w;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds + w * 256), 16, srcchunk * 16, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int e = 0; e < 4; e++) out[l * 4 + e] = lds[l * 4 + e];
}
// accumulator-as-operand: X = A1[32x16]*B1[16x32] (32x32, small ints), then Y = X^T * B2 where
// B2 is [32 k][32 cols]; X^T*B2 uses X regs as the A operand with permuted k order.
__global__ void k_accop(const u16* A1, const u16* B1t, const u16* B2t, float* Y) {
  int l = threadIdx.x, r = l & 31, h = l >> 5;
  bf16x8 a = *(const bf16x8*)(A1 + r * 16 + 8 * h);
  bf16x8 b = *(const bf16x8*)(B1t + r * 16 + 8 * h);
  f32x16 x = {};
  x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, x, 0, 0, 0);   // x: col=lane&31, row=(reg&3)+8*(reg>>2)+4h
  f32x16 y = {};
  for (int s = 0; s < 2; s++) {
    bf16x8 xa, b2;
    for (int j = 0; j < 8; j++) {
      xa[j] = (__bf16)x[8 * s + j];
      int k = 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);     // row of X this element holds
      unsigned short bits = B2t[r * 32 + k];               // B2[k][col r] stored as B2t[col][k]
      b2[j] = __builtin_bit_cast(__bf16, bits);
    }
    y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa, b2, y, 0, 0, 0);   // Y = X^T * B2  -> [32 (X cols)][32]
  }
  for (int q = 0; q < 16; q++) Y[((q & 3) + 8 * (q >> 2) + 4 * h) * 32 + r] = y[q];
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device: %s arch=%s CUs=%d clock=%d kHz LDS/block=%zu maxLDS/CU=%zu L2=%d HBM=%.1f GB\n", p.name, p.gcnArchName,
         p.multiProcessorCount, p.clockRate, p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor, p.l2CacheSize,
         p.totalGlobalMem / 1e9);
  int fails = 0;
  srand(1);
  {  // mfma16
    std::vector<u16> A(16 * 32), Bt(16 * 32); std::vector<float> Af(16 * 32), Bf(16 * 32);
    for (int i = 0; i < 16 * 32; i++) { Af[i] = (float)(rand() % 7 - 3); Bf[i] = (float)(rand() % 5 - 2); A[i] = f2bf(Af[i]); Bt[i] = f2bf(Bf[i]); }
    u16 *dA, *dB; float* dC; CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1024)); CK(hipMalloc(&dC, 1024));
    CK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, Bt.data(), 1024, hipMemcpyHostToDevice));
    k_mfma16<<<1, 64>>>(dA, dB, dC); CK(hipDeviceSynchronize());
    std::vector<float> C(256); CK(hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int m = 0; m < 16; m++) for (int n = 0; n < 16; n++) { float s = 0; for (int k = 0; k < 32; k++) s += Af[m * 32 + k] * Bf[n * 32 + k]; if (s != C[m * 16 + n]) bad++; }
    printf("mfma_f32_16x16x32_bf16 layout: %s (%d bad)\n", bad ? "FAIL" : "PASS", bad); fails += bad != 0;
  }
  {  // mfma32
    std::vector<u16> A(32 * 16), Bt(32 * 16); std::vector<float> Af(512), Bf(512);
    for (int i = 0; i < 512; i++) { Af[i] = (float)(rand() % 7 - 3); Bf[i] = (float)(rand() % 5 - 2); A[i] = f2bf(Af[i]); Bt[i] = f2bf(Bf[i]); }
    u16 *dA, *dB; float* dC; CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1024)); CK(hipMalloc(&dC, 4096));
    CK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, Bt.data(), 1024, hipMemcpyHostToDevice));
    k_mfma32<<<1, 64>>>(dA, dB, dC); CK(hipDeviceSynchronize());
    std::vector<float> C(1024); CK(hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int m = 0; m < 32; m++) for (int n = 0; n < 32; n++) { float s = 0; for (int k = 0; k < 16; k++) s += Af[m * 16 + k] * Bf[n * 16 + k]; if (s != C[m * 32 + n]) bad++; }
    printf("mfma_f32_32x32x16_bf16 layout: %s (%d bad)\n", bad ? "FAIL" : "PASS", bad); fails += bad != 0;
  }
  {  // tr read
    std::vector<u16> M(16 * 64); for (int i = 0; i < 16 * 64; i++) M[i] = (u16)i;
    u16 *dI, *dO; CK(hipMalloc(&dI, 2048)); CK(hipMalloc(&dO, 512));
    CK(hipMemcpy(dI, M.data(), 2048, hipMemcpyHostToDevice));
    k_tr<<<1, 64>>>(dI, dO); CK(hipDeviceSynchronize());
    std::vector<u16> O(256); CK(hipMemcpy(O.data(), dO, 512, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; l++) for (int e = 0; e < 4; e++) { int g = l >> 4, i = l & 15; u16 exp = M[(4 * g + e) * 64 + i]; if (O[l * 4 + e] != exp) bad++; }
    printf("ds_read_b64_tr_b16 semantics: %s (%d bad)\n", bad ? "FAIL" : "PASS", bad); fails += bad != 0;
    if (bad) { for (int l = 0; l < 20; l++) printf("  lane %d got %d %d %d %d\n", l, O[l*4], O[l*4+1], O[l*4+2], O[l*4+3]); }
  }
  {  // glds
    std::vector<unsigned> S(256 * 4); for (int i = 0; i < 1024; i++) S[i] = i * 3 + 1;
    unsigned *dS, *dO; CK(hipMalloc(&dS, 4096)); CK(hipMalloc(&dO, 4096));
    CK(hipMemcpy(dS, S.data(), 4096, hipMemcpyHostToDevice));
    k_glds<<<1, 256>>>(dS, dO); CK(hipDeviceSynchronize());
    std::vector<unsigned> O(1024); CK(hipMemcpy(O.data(), dO, 4096, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 256; l++) for (int e = 0; e < 4; e++) { int w = l >> 6, lane = l & 63; int sc = (lane ^ 5) + 64 * w; if (O[l * 4 + e] != S[sc * 4 + e]) bad++; }
    printf("global_load_lds dwordx4 semantics: %s (%d bad)\n", bad ? "FAIL" : "PASS", bad); fails += bad != 0;
  }
  {  // buffer_load lds with bounds
    std::vector<unsigned> S(256 * 4); for (int i = 0; i < 1024; i++) S[i] = i * 7 + 5;
    unsigned *dS, *dO; CK(hipMalloc(&dS, 4096)); CK(hipMalloc(&dO, 4096));
    CK(hipMemcpy(dS, S.data(), 4096, hipMemcpyHostToDevice));
    CK(hipMemset(dO, 0xff, 4096));
    const int valid = 3000;  // bytes; chunks at offset >= 3000-15 are (partly) out of range
    k_blds<<<1, 256>>>(dS, dO, valid); CK(hipDeviceSynchronize());
    std::vector<unsigned> O(1024); CK(hipMemcpy(O.data(), dO, 4096, hipMemcpyDeviceToHost));
    int bad = 0, zeros = 0;
    for (int l = 0; l < 256; l++) for (int e = 0; e < 4; e++) { int w = l >> 6, lane = l & 63; int sc = (lane ^ 3) + 64 * w; int byte = sc * 16 + e * 4;
      unsigned exp = (byte + 4 <= valid) ? S[sc * 4 + e] : 0u; if (O[l * 4 + e] != exp) bad++; if (byte + 4 > valid) zeros++; }
    printf("raw_buffer_load_lds dwordx4 + bounds->0 semantics: %s (%d bad, %d expected zeros)\n", bad ? "FAIL" : "PASS", bad, zeros); fails += bad != 0;
  }
  {  // acc as operand
    std::vector<u16> A1(512), B1t(512), B2t(1024); std::vector<float> A1f(512), B1f(512), B2f(1024);
    for (int i = 0; i < 512; i++) { A1f[i] = (float)(rand() % 3 - 1); B1f[i] = (float)(rand() % 3 - 1); A1[i] = f2bf(A1f[i]); B1t[i] = f2bf(B1f[i]); }
    for (int i = 0; i < 1024; i++) { B2f[i] = (float)(rand() % 5 - 2); B2t[i] = f2bf(B2f[i]); }
    u16 *dA, *dB, *dB2; float* dY; CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1024)); CK(hipMalloc(&dB2, 2048)); CK(hipMalloc(&dY, 4096));
    CK(hipMemcpy(dA, A1.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B1t.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB2, B2t.data(), 2048, hipMemcpyHostToDevice));
    k_accop<<<1, 64>>>(dA, dB, dB2, dY); CK(hipDeviceSynchronize());
    std::vector<float> Y(1024); CK(hipMemcpy(Y.data(), dY, 4096, hipMemcpyDeviceToHost));
    // X[m][n] = sum_k A1[m][k] B1t[n][k]; Y[n][c] = sum_m X[m][n] * B2[m][c], B2[m][c] = B2t[c][m]
    std::vector<float> X(1024);
    for (int m = 0; m < 32; m++) for (int n = 0; n < 32; n++) { float s = 0; for (int k = 0; k < 16; k++) s += A1f[m * 16 + k] * B1f[n * 16 + k]; X[m * 32 + n] = s; }
    int bad = 0;
    for (int n = 0; n < 32; n++) for (int c = 0; c < 32; c++) { float s = 0; for (int m = 0; m < 32; m++) s += X[m * 32 + n] * B2f[c * 32 + m]; if (s != Y[n * 32 + c]) bad++; }
    printf("accumulator-as-A-operand (X^T*B) k-permutation: %s (%d bad)\n", bad ? "FAIL" : "PASS", bad); fails += bad != 0;
  }
  printf("probe done: %d failing checks\n", fails);
  return fails ? 1 : 0;
}
