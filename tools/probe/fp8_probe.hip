// Hardware-assumption probe for the block-scaled fp8 MFMA on gfx950 (v_mfma_scale_f32_32x32x64_f8f6f4, OCP e4m3 operands):
// which (row, k) each byte of a lane's 32-byte A / B operand holds, and what the E8M0 scale operand does.
// The attention kernel of BASELINE.json config 5 (attention_fp8.hip) is written against what this prints.
// Build: hipcc --offload-arch=gfx950 -O2 fp8_probe.hip -o fp8_probe ; run on an MI355X.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(2);} } while (0)

// OCP e4m3 (bias 7, no infinities): exact encoder for small integers / powers of two
static unsigned char enc_e4m3(float f) {
  if (f == 0.f) return 0;
  unsigned char s = f < 0 ? 0x80 : 0;
  f = fabsf(f);
  int e;
  float m = frexpf(f, &e);  // f = m * 2^e, m in [0.5, 1)
  int E = e - 1 + 7;        // f = (2m) * 2^(e-1)
  int M = (int)lrintf((2 * m - 1) * 8);
  if (E <= 0 || E > 15 || M > 7) { printf("value %g not handled by the probe encoder\n", f); exit(3); }
  return s | (unsigned char)(E << 3) | (unsigned char)M;
}

// per-lane scales: lane (r, h) supplies the scale of row / column r, contraction block h
__global__ void k_scaled_lane(const unsigned char* abytes, const unsigned char* bbytes, float* C, const int* sa, const int* sb) {
  const int l = threadIdx.x;
  v8i a = *(const v8i*)(abytes + l * 32);
  v8i b = *(const v8i*)(bbytes + l * 32);
  v16f c = {};
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa[l], 0, sb[l]);
  for (int r = 0; r < 16; r++) C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}

__global__ void k_scaled(const unsigned char* abytes, const unsigned char* bbytes, float* C, int sa, int sb) {
  const int l = threadIdx.x;
  v8i a = *(const v8i*)(abytes + l * 32);
  v8i b = *(const v8i*)(bbytes + l * 32);
  v16f c = {};
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
  for (int r = 0; r < 16; r++) C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];   // C/D map is shape-determined
}

// candidate maps: byte b (0..31) of lane (r = l & 31, h = l >> 5) -> k
static int kmap(int cand, int h, int b) {
  switch (cand) {
    case 0: return 32 * h + b;                              // 32 consecutive k per lane half
    case 1: return 32 * (b >> 4) + 16 * h + (b & 15);       // two 16-wide chunks per lane, halves interleaved
    case 2: return 16 * (b >> 3) + 8 * h + (b & 7);         // four 8-wide chunks (the bf16 32x32x16 pattern, four times)
    default: return 0;
  }
}

int main() {
  const int M = 32, N = 32, K = 64;
  std::vector<float> A(M * K), B(K * N), ref(M * N, 0.f);
  srand(1);
  for (auto& v : A) v = (float)(rand() % 5 - 2);
  for (auto& v : B) v = (float)(rand() % 5 - 2);
  for (int i = 0; i < M; i++)
    for (int j = 0; j < N; j++) {
      float s = 0;
      for (int k = 0; k < K; k++) s += A[i * K + k] * B[k * N + j];
      ref[i * N + j] = s;
    }
  unsigned char *da, *db;
  float* dc;
  CK(hipMalloc(&da, 64 * 32));
  CK(hipMalloc(&db, 64 * 32));
  CK(hipMalloc(&dc, M * N * 4));
  std::vector<unsigned char> ha(64 * 32), hb(64 * 32);
  std::vector<float> hc(M * N);
  int found = -1;
  for (int cand = 0; cand < 3; cand++) {
    for (int l = 0; l < 64; l++)
      for (int b = 0; b < 32; b++) {
        const int k = kmap(cand, l >> 5, b);
        ha[l * 32 + b] = enc_e4m3(A[(l & 31) * K + k]);   // A[row = l & 31][k]
        hb[l * 32 + b] = enc_e4m3(B[k * N + (l & 31)]);   // B[k][col = l & 31]
      }
    CK(hipMemcpy(da, ha.data(), ha.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), hb.size(), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_scaled, dim3(1), dim3(64), 0, 0, da, db, dc, 127, 127);   // E8M0 127 = 2^0
    CK(hipMemcpy(hc.data(), dc, M * N * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < M * N; i++) bad += hc[i] != ref[i];
    printf("k-map candidate %d: %d / %d mismatches\n", cand, bad, M * N);
    if (!bad && found < 0) found = cand;
  }
  if (found < 0) { printf("NO candidate k-map matches\n"); return 1; }
  // scales: E8M0 byte e -> 2^(e-127); byte 0 of the VGPR with op_sel 0; per lane = per (row, 32-k block)
  for (int l = 0; l < 64; l++)
    for (int b = 0; b < 32; b++) {
      const int k = kmap(found, l >> 5, b);
      ha[l * 32 + b] = enc_e4m3(A[(l & 31) * K + k]);
      hb[l * 32 + b] = enc_e4m3(B[k * N + (l & 31)]);
    }
  CK(hipMemcpy(da, ha.data(), ha.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(db, hb.data(), hb.size(), hipMemcpyHostToDevice));
  struct { int sa, sb; float f; } cases[] = {{128, 127, 2.f}, {127, 129, 4.f}, {126, 126, 0.25f}, {127 | (130 << 8), 127, 1.f}};
  for (auto& cs : cases) {
    hipLaunchKernelGGL(k_scaled, dim3(1), dim3(64), 0, 0, da, db, dc, cs.sa, cs.sb);
    CK(hipMemcpy(hc.data(), dc, M * N * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < M * N; i++) bad += hc[i] != ref[i] * cs.f;
    printf("scale_a 0x%x scale_b 0x%x expect x%g: %d mismatches\n", cs.sa, cs.sb, cs.f, bad);
  }
  {  // per-lane scales with candidate map 0 (k = 32 h + b): reference applies 2^(sa[row][h] - 127) * 2^(sb[col][h] - 127) per 32-k block
    std::vector<int> hsa(64), hsb(64);
    for (int l = 0; l < 64; l++) { hsa[l] = 124 + (l * 7) % 6; hsb[l] = 125 + (l * 5) % 5; }
    int *dsa, *dsb;
    CK(hipMalloc(&dsa, 256)); CK(hipMalloc(&dsb, 256));
    CK(hipMemcpy(dsa, hsa.data(), 256, hipMemcpyHostToDevice));
    CK(hipMemcpy(dsb, hsb.data(), 256, hipMemcpyHostToDevice));
    for (int l = 0; l < 64; l++)
      for (int b = 0; b < 32; b++) {
        const int k = 32 * (l >> 5) + b;
        ha[l * 32 + b] = enc_e4m3(A[(l & 31) * K + k]);
        hb[l * 32 + b] = enc_e4m3(B[k * N + (l & 31)]);
      }
    CK(hipMemcpy(da, ha.data(), ha.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), hb.size(), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_scaled_lane, dim3(1), dim3(64), 0, 0, da, db, dc, dsa, dsb);
    CK(hipMemcpy(hc.data(), dc, M * N * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < M; i++)
      for (int j = 0; j < N; j++) {
        float s = 0;
        for (int h = 0; h < 2; h++) {
          float part = 0;
          for (int k = 32 * h; k < 32 * h + 32; k++) part += A[i * K + k] * B[k * N + j];
          s += part * ldexpf(1.f, hsa[i + 32 * h] - 127) * ldexpf(1.f, hsb[j + 32 * h] - 127);
        }
        bad += hc[i * N + j] != s;
      }
    printf("per-lane scales (lane (r,h) -> row r, block h): %d mismatches\n", bad);
  }
  {  // which (row, 32-k block) does lane L's A-scale byte act on?  (data placed with map 0; B scales all 2^0)
    std::vector<int> hsa(64), hsb(64, 127);
    int *dsa, *dsb;
    CK(hipMalloc(&dsa, 256)); CK(hipMalloc(&dsb, 256));
    CK(hipMemcpy(dsb, hsb.data(), 256, hipMemcpyHostToDevice));
    const int probes[] = {0, 1, 5, 31, 32, 33, 37, 63};
    for (int which = 0; which < 2; which++)       // 0: vary A scales, 1: vary B scales
      for (int L : probes) {
        for (int l = 0; l < 64; l++) hsa[l] = 127;
        hsa[L] = 128;
        CK(hipMemcpy(which ? dsb : dsa, hsa.data(), 256, hipMemcpyHostToDevice));
        for (int l = 0; l < 64; l++) hsa[l] = 127;
        CK(hipMemcpy(which ? dsa : dsb, hsa.data(), 256, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_scaled_lane, dim3(1), dim3(64), 0, 0, da, db, dc, dsa, dsb);
        CK(hipMemcpy(hc.data(), dc, M * N * 4, hipMemcpyDeviceToHost));
        printf("%c scale of lane %2d doubled ->", which ? 'B' : 'A', L);
        const int x = L & 31;
        // partial sums per 8-byte group of the contraction as the data was placed (map 0: k = 8 q8 + e; half = q8 >> 2)
        int subset = -1, nmatch = 0;
        for (int sub = 1; sub < 256; sub++) {
          bool ok = true;
          for (int y = 0; y < 32 && ok; y++) {
            const int i = which ? y : x, j = which ? x : y;
            float want = hc[i * N + j] - ref[i * N + j], got = 0;
            for (int q8 = 0; q8 < 8; q8++)
              if (sub >> q8 & 1)
                for (int k = 8 * q8; k < 8 * q8 + 8; k++) got += A[i * K + k] * B[k * N + j];
            ok = got == want;
          }
          if (ok) { subset = sub; nmatch++; }
        }
        printf(" %s %d: groups of 8 k (bit q = k 8q..8q+7) 0x%02x (%d subsets fit)", which ? "col" : "row", x, subset, nmatch);
        printf("\n");
      }
  }
  printf("RESULT kmap=%d\n", found);
  return 0;
}
