// would the attention backward's key-owner block be faster at ONE wave per SIMD with 64 keys per wave and its dK^T / dV^T
// accumulators in AGPRs?  (Round 3 called that form "not buildable from HIP source": hipcc picks the AGPR form or - with -mllvm -amdgpu-mfma-vgpr-form -
// the VGPR form for ALL of a kernel's MFMA builtins.  It IS buildable: builtins in VGPR form for the S' / dP' chains whose results the VALU reads,
// inline-asm v_mfma with "+a" operands for the two accumulating chains - hipcc allocates 128 AGPRs for them and leaves them alone.)
//
// The probe runs the loop body of attn_dkv2_block (attention_bwd.hip) without staging, as tools/probe/mfma_shape_probe.hip does, in two forms:
//   K32  the shipped shape: 32 keys per wave, 2 blocks of 4 waves per CU (two waves per SIMD), all accumulators in VGPRs, compiler-scheduled
//   K64  64 keys per wave (two 32-key tiles sharing every Q / dO fragment, row constant and transposed fragment read from LDS: half the LDS bytes per
//        MFMA), ONE block of 4 waves per CU (one wave per SIMD: the block declares 96 KiB of LDS), dK^T / dV^T in 128 AGPRs through inline asm
// on random and on all-zero operands, and prints the executed TFLOP/s of each.  Results are not checked against anything.
// Build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form kv64_probe.hip -o kv64_probe ; run on an MI355X.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(2);} } while (0)

__device__ __forceinline__ s16x4 tr16(const void* p) { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)p); }
__device__ __forceinline__ bf16x8 tr_pair(const unsigned char* a, const unsigned char* b) {
  const s16x4 x = tr16(a), y = tr16(b);
  const s16x8 v = {x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
  return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ int swz128(int row, int c16) {
  const int x = (row >> 1) & 7;
  return row * 128 + ((c16 ^ (((x & 1) << 2) | (x >> 1))) << 4);
}
// accumulate into an AGPR block: the operands are VGPRs the compiler has just written (s_nop 1: MI355X guide 5.7 item 2)
__device__ __forceinline__ void mfma_a(f32x16& acc, bf16x8 a, bf16x8 b) {
  asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}

constexpr int IMG = 128 * 128;               // one 128-row x 128-byte image (Q or dO tile)
constexpr int BUF = 2 * IMG + 2 * 128 * 4;   // Q | dO | row constants (lse, delta)

__device__ __forceinline__ void fill(unsigned char* smem, const unsigned short* src, int t) {
  for (int i = t; i < 2 * BUF / 2; i += 256) ((unsigned short*)smem)[i] = src[(blockIdx.x * 977 + i) & 0xfffff];
  for (int i = t; i < 512; i += 256) {
    const int b = i >> 8, j = i & 255;
    ((float*)(smem + b * BUF + 2 * IMG))[j] = -8.f - (float)((src[i] & 1023)) * (1.f / 256.f);
  }
}

// ---- K32: the shipped shape (two blocks per CU) ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void k32(const unsigned short* __restrict__ src, float* __restrict__ sink, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BUF];
  const int t = threadIdx.x, l = t & 63;
  fill(smem, src, t);
  bf16x8 kf[4], vf[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    kf[i] = *(const bf16x8*)(src + (((blockIdx.x * 256 + t) * 64) & 0xfffff) + i * 8);
    vf[i] = *(const bf16x8*)(src + (((blockIdx.x * 256 + t) * 64) & 0xfffff) + 32 + i * 8);
  }
  __syncthreads();
  const int r = l & 31, h2 = l >> 5, g1 = (l >> 4) & 1, qq = (l >> 2) & 3, pp = l & 3;
  f32x16 dkt[2], dvt[2];
#pragma unroll
  for (int i = 0; i < 16; i++) { dkt[0][i] = dkt[1][i] = dvt[0][i] = dvt[1][i] = 0.f; }
  for (int it = 0; it < iters; it++) {
    const unsigned char* sq = smem + (it & 1) * BUF;
    const unsigned char* sdo = sq + IMG;
    const int rb = 32 * ((it >> 1) & 3);
    bf16x8 qa[4], da[4], ta[2][2], tb[2][2], pf[2], dsf[2];
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      qa[ks] = *(const bf16x8*)(sq + swz128(rb + r, 2 * ks + h2));
      da[ks] = *(const bf16x8*)(sdo + swz128(rb + r, 2 * ks + h2));
    }
    f32x16 s = *(const f32x16*)((const float*)(sq + 2 * IMG) + rb + 16 * h2);
    f32x16 dp = *(const f32x16*)((const float*)(sq + 2 * IMG) + 128 + rb + 16 * h2);
#pragma unroll
    for (int ks = 0; ks < 4; ks++) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[ks], kf[ks], s, 0, 0, 0);
#pragma unroll
    for (int ks = 0; ks < 4; ks++) dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da[ks], vf[ks], dp, 0, 0, 0);
#pragma unroll
    for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
      for (int dt = 0; dt < 2; dt++) {
        const int qb = rb + 16 * s2 + 4 * h2 + qq, c16 = 4 * dt + 2 * g1 + (pp >> 1), sb8 = 8 * (pp & 1);
        ta[s2][dt] = tr_pair(sdo + swz128(qb, c16) + sb8, sdo + swz128(qb + 8, c16) + sb8);
        tb[s2][dt] = tr_pair(sq + swz128(qb, c16) + sb8, sq + swz128(qb + 8, c16) + sb8);
      }
#pragma unroll
    for (int i = 0; i < 16; i++) s[i] = __builtin_amdgcn_exp2f(s[i]);
#pragma unroll
    for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
      for (int j = 0; j < 8; j++) pf[s2][j] = (__bf16)s[8 * s2 + j];
#pragma unroll
    for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
      for (int dt = 0; dt < 2; dt++) dvt[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ta[s2][dt], pf[s2], dvt[dt], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 16; i++) dp[i] *= s[i];
#pragma unroll
    for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
      for (int j = 0; j < 8; j++) dsf[s2][j] = (__bf16)dp[8 * s2 + j];
#pragma unroll
    for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
      for (int dt = 0; dt < 2; dt++) dkt[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tb[s2][dt], dsf[s2], dkt[dt], 0, 0, 0);
  }
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < 16; i++) acc += dkt[0][i] + dkt[1][i] + dvt[0][i] + dvt[1][i];
  sink[blockIdx.x * 256 + t] = acc;
}

// ---- K64: 64 keys per wave, one wave per SIMD, dK^T / dV^T in AGPRs -------------------------------------------------------------------------------
template <int PHASED>
__global__ __launch_bounds__(256, 1) void k64(const unsigned short* __restrict__ src, float* __restrict__ sink, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[96 * 1024];   // > 80 KiB: one block per CU
  const int t = threadIdx.x, l = t & 63;
  fill(smem, src, t);
  bf16x8 kf[2][4], vf[2][4];
#pragma unroll
  for (int kt = 0; kt < 2; kt++)
#pragma unroll
    for (int i = 0; i < 4; i++) {
      kf[kt][i] = *(const bf16x8*)(src + (((blockIdx.x * 256 + t) * 128 + 64 * kt) & 0xfffff) + i * 8);
      vf[kt][i] = *(const bf16x8*)(src + (((blockIdx.x * 256 + t) * 128 + 64 * kt) & 0xfffff) + 32 + i * 8);
    }
  __syncthreads();
  const int r = l & 31, h2 = l >> 5, g1 = (l >> 4) & 1, qq = (l >> 2) & 3, pp = l & 3;
  f32x16 dkt[2][2], dvt[2][2];
#pragma unroll
  for (int kt = 0; kt < 2; kt++)
#pragma unroll
    for (int dt = 0; dt < 2; dt++)
#pragma unroll
      for (int i = 0; i < 16; i++) { dkt[kt][dt][i] = 0.f; dvt[kt][dt][i] = 0.f; }
  for (int it = 0; it < iters; it++) {
    const unsigned char* sq = smem + (it & 1) * BUF;
    const unsigned char* sdo = sq + IMG;
    const int rb = 32 * ((it >> 1) & 3);
    bf16x8 qa[4], da[4], ta[2][2], tb[2][2];
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      qa[ks] = *(const bf16x8*)(sq + swz128(rb + r, 2 * ks + h2));
      da[ks] = *(const bf16x8*)(sdo + swz128(rb + r, 2 * ks + h2));
    }
    const f32x16 s0 = *(const f32x16*)((const float*)(sq + 2 * IMG) + rb + 16 * h2);
    const f32x16 d0 = *(const f32x16*)((const float*)(sq + 2 * IMG) + 128 + rb + 16 * h2);
#pragma unroll
    for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
      for (int dt = 0; dt < 2; dt++) {
        const int qb = rb + 16 * s2 + 4 * h2 + qq, c16 = 4 * dt + 2 * g1 + (pp >> 1), sb8 = 8 * (pp & 1);
        ta[s2][dt] = tr_pair(sdo + swz128(qb, c16) + sb8, sdo + swz128(qb + 8, c16) + sb8);
        tb[s2][dt] = tr_pair(sq + swz128(qb, c16) + sb8, sq + swz128(qb + 8, c16) + sb8);
      }
    f32x16 s[2], dp[2];
    bf16x8 pf[2][2], dsf[2][2];
    // both key tiles' S' and dP' chains first (16 MFMAs: the exponentials of tile 0 can start under tile 1's chains), then the four accumulating chains
#pragma unroll
    for (int kt = 0; kt < 2; kt++) {
#pragma unroll
      for (int ks = 0; ks < 4; ks++) s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[ks], kf[kt][ks], ks == 0 ? s0 : s[kt], 0, 0, 0);
#pragma unroll
      for (int ks = 0; ks < 4; ks++) dp[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da[ks], vf[kt][ks], ks == 0 ? d0 : dp[kt], 0, 0, 0);
    }
#pragma unroll
    for (int kt = 0; kt < 2; kt++) {
#pragma unroll
      for (int i = 0; i < 16; i++) s[kt][i] = __builtin_amdgcn_exp2f(s[kt][i]);
#pragma unroll
      for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
        for (int j = 0; j < 8; j++) pf[kt][s2][j] = (__bf16)s[kt][8 * s2 + j];
#pragma unroll
      for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
        for (int dt = 0; dt < 2; dt++) mfma_a(dvt[kt][dt], ta[s2][dt], pf[kt][s2]);
#pragma unroll
      for (int i = 0; i < 16; i++) dp[kt][i] *= s[kt][i];
#pragma unroll
      for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
        for (int j = 0; j < 8; j++) dsf[kt][s2][j] = (__bf16)dp[kt][8 * s2 + j];
#pragma unroll
      for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
        for (int dt = 0; dt < 2; dt++) mfma_a(dkt[kt][dt], tb[s2][dt], dsf[kt][s2]);
      if (PHASED) __builtin_amdgcn_sched_barrier(0);
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the last accumulating MFMAs have written their AGPRs before the compiler reads them
  float acc = 0.f;
#pragma unroll
  for (int kt = 0; kt < 2; kt++)
#pragma unroll
    for (int dt = 0; dt < 2; dt++)
#pragma unroll
      for (int i = 0; i < 16; i++) acc += dkt[kt][dt][i] + dvt[kt][dt][i];
  sink[blockIdx.x * 256 + t] = acc;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  unsigned short* src; float* sink;
  CK(hipMalloc(&src, (1 << 20) * 2 + 4096 * 128 * 2));
  CK(hipMalloc(&sink, 512 * 256 * 4));
  std::vector<unsigned short> h((1 << 20) + 4096 * 128);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int data = 0; data < 2; data++) {
    srand(1);
    for (auto& x : h) x = data == 0 ? (unsigned short)(((rand() & 1) << 15) | ((124 + (rand() & 3)) << 7) | (rand() & 127)) : 0;
    CK(hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 2; rep++)
      for (int form = 0; form < 3; form++) {
        float ms = 0.f;
        const int blocks = form == 0 ? 512 : 256;
        for (int pass = 0; pass < 2; pass++) {   // the first pass warms the clock governor up
          CK(hipEventRecord(e0));
          if (form == 0) hipLaunchKernelGGL(k32, dim3(blocks), dim3(256), 0, 0, src, sink, iters);
          else if (form == 1) hipLaunchKernelGGL(k64<0>, dim3(blocks), dim3(256), 0, 0, src, sink, iters);
          else hipLaunchKernelGGL(k64<1>, dim3(blocks), dim3(256), 0, 0, src, sink, iters);
          CK(hipEventRecord(e1));
          CK(hipEventSynchronize(e1));
          CK(hipEventElapsedTime(&ms, e0, e1));
        }
        const double flop = 2.0 * 32 * 32 * 64 * 4 * (double)iters * 4 * blocks * (form == 0 ? 1 : 2);   // four products per 32 x 32 sub-tile, four waves per block
        printf("%s operands, %-58s: %8.2f ms  %7.1f TFLOP/s executed\n", data == 0 ? "random" : "zero  ",
               form == 0 ? "K32: 32 keys / wave, two waves per SIMD, VGPR accumulators" : (form == 1 ? "K64: 64 keys / wave, one wave per SIMD, AGPR accumulators" : "K64, one scheduling region per key tile"),
               ms, flop / (ms * 1e-3) / 1e12);
      }
  }
  return 0;
}
