// Small HBM-bound helpers of the MIM step: weight casts / transposes, decoder-input assembly, row gathers, column sums.
#include "smbv_common.h"
#include "smbv_kernels.h"

namespace smbv {

// fp32 -> bf16 (n multiple of 4)
__global__ void cast_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, size_t n4) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const f32x4 v = ((const f32x4*)src)[i];
    u32x2 pk = {pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
    ((u32x2*)dst)[i] = pk;
  }
}

int cast_f32_bf16(const float* src, bf16_t* dst, size_t n, hipStream_t st) {
  if (n == 0) return SMBV_OK;
  if ((n & 3) || ((uintptr_t)src & 15) || ((uintptr_t)dst & 7)) { set_error("cast_f32_bf16: n %% 4 / alignment"); return SMBV_ERR_ALIGN; }
  const size_t n4 = n >> 2;
  hipLaunchKernelGGL(cast_kernel, dim3((unsigned)std::min<size_t>(4096, (n4 + 255) / 256)), dim3(256), 0, st, src, dst, n4);
  return check_launch("cast_f32_bf16");
}

// dst bf16 [C][ldd >= R] = transpose(src fp32 [R][C]); 32x32 tiles through LDS
__global__ __launch_bounds__(256) void transpose_cast_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int R, int C, int ldd) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int r = r0 + ty + 8 * i, c = c0 + tx;
    tile[ty + 8 * i][tx] = (r < R && c < C) ? src[(size_t)r * C + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int c = c0 + ty + 8 * i, r = r0 + tx;
    if (c < C && r < R) dst[(size_t)c * ldd + r] = f2bf(tile[tx][ty + 8 * i]);
  }
}

int transpose_cast_f32_bf16(const float* src, bf16_t* dst, int R, int C, int ldd, hipStream_t st) {
  if (R <= 0 || C <= 0 || ldd < R) { set_error("transpose_cast: empty"); return SMBV_ERR_SHAPE; }
  hipLaunchKernelGGL(transpose_cast_kernel, dim3(cdiv(C, 32), cdiv(R, 32)), dim3(256), 0, st, src, dst, R, C, ldd);
  return check_launch("transpose_cast");
}

// All per-step weight preparation in a handful of launches: the host sequencer lists, per fp32 master tensor, a bf16 cast (optionally
// transposed, with a destination leading dimension), an fp32 copy or a zero fill; the list travels BY VALUE in the kernel arguments,
// 64 jobs (2.5 KiB) per launch - no device-resident table, no host-to-device copy (a pageable hipMemcpyAsync would block the caller
// behind everything queued on the stream), no state; a block handles one 32x32 tile of one job (binary search over the first-tile
// prefix).  Replaces ~200 tiny cast / transpose / copy launches by ~5.
struct PrepBatch {
  PrepJob jobs[64];
  int n;
};

__global__ __launch_bounds__(256) void prep_weights_kernel(const PrepBatch batch) {
  __shared__ float tile64[64][65];
  float (*tile)[33] = (float (*)[33])&tile64[0][0];   // the scalar kinds' 32 x 33 view of the same memory
  int lo = 0, hi = batch.n - 1;
  const int b = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (batch.jobs[mid].tile0 <= b) lo = mid; else hi = mid - 1;
  }
  const PrepJob j = batch.jobs[lo];
  const int tl = b - j.tile0;
  if (j.kind == 4) {
    // ONE read of the fp32 master per step: a 64 x 64 tile as 16-byte loads (256-byte row segments), the bf16 copy as 8-byte stores, the transposed
    // copy through LDS (row stride 65: both the row writes and the column reads spread over all banks) as 8-byte stores along the transposed rows
    const int tc64 = (j.C + 63) >> 6;
    const int r0 = (tl / tc64) * 64, c0 = (tl % tc64) * 64;
    const int lr = threadIdx.x >> 4, lc = (threadIdx.x & 15) * 4;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int r = r0 + lr + 16 * i, c = c0 + lc;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      const bool in = r < j.R && c < j.C;   // (C % 4 == 0: a 4-column group is inside or outside as a whole)
      if (in) v = *(const f32x4*)(j.src + (size_t)r * j.lds + c);
      if (in && j.dst) {
        u32x2 o;
        o[0] = pack_bf16(v[0], v[1]);
        o[1] = pack_bf16(v[2], v[3]);
        *(u32x2*)((bf16_t*)j.dst + (size_t)r * j.ldd + c) = o;
      }
      if (j.dst2) {
#pragma unroll
        for (int e = 0; e < 4; e++) tile64[lr + 16 * i][lc + e] = v[e];
      }
    }
    if (j.dst2) {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int c = lr + 16 * i, r = lc;   // transposed row c0 + c, its columns r0 + r .. r0 + r + 3
        if (c0 + c < j.C && r0 + r < j.R) {  // (R % 4 == 0)
          u32x2 o;
          o[0] = pack_bf16(tile64[r][c], tile64[r + 1][c]);
          o[1] = pack_bf16(tile64[r + 2][c], tile64[r + 3][c]);
          *(u32x2*)((bf16_t*)j.dst2 + (size_t)(c0 + c) * j.ldd2 + r0 + r) = o;
        }
      }
    }
    return;
  }
  const int tiles_c = (j.C + 31) >> 5;
  const int r0 = (tl / tiles_c) * 32, c0 = (tl % tiles_c) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  if (j.kind == 0) {  // bf16 cast, same layout (dst leading dimension ldd)
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int r = r0 + ty + 8 * i, c = c0 + tx;
      if (r < j.R && c < j.C) ((bf16_t*)j.dst)[(size_t)r * j.ldd + c] = f2bf(j.src[(size_t)r * j.lds + c]);
    }
  } else if (j.kind == 1) {  // bf16 cast, transposed: dst[c][r]
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int r = r0 + ty + 8 * i, c = c0 + tx;
      tile[ty + 8 * i][tx] = (r < j.R && c < j.C) ? j.src[(size_t)r * j.lds + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int c = c0 + ty + 8 * i, r = r0 + tx;
      if (c < j.C && r < j.R) ((bf16_t*)j.dst)[(size_t)c * j.ldd + r] = f2bf(tile[tx][ty + 8 * i]);
    }
  } else {  // fp32 copy (kind 2) or zero fill (kind 3) of a [1, C] row
    for (int c = c0 + threadIdx.x; c < min(j.C, c0 + 32); c += 256) ((float*)j.dst)[c] = j.kind == 2 ? j.src[c] : 0.f;
  }
}

int prep_weights(const PrepJob* jobs, int njobs, hipStream_t st) {
  for (int j0 = 0; j0 < njobs; j0 += 64) {
    PrepBatch b;
    b.n = std::min(64, njobs - j0);
    const int base = jobs[j0].tile0;
    int tiles = 0;
    for (int i = 0; i < b.n; i++) {
      b.jobs[i] = jobs[j0 + i];
      b.jobs[i].tile0 -= base;
      tiles = b.jobs[i].tile0 + prep_job_tiles(b.jobs[i]);
    }
    for (int i = b.n; i < 64; i++) b.jobs[i] = b.jobs[b.n - 1];
    if (tiles <= 0) continue;
    hipLaunchKernelGGL(prep_weights_kernel, dim3(tiles), dim3(256), 0, st, b);
    int rc = check_launch("prep_weights");
    if (rc) return rc;
  }
  return SMBV_OK;
}

// decoder input, masked part (src/models/videomae/modeling_videomae.py:811-815): x_full[b, Nvis + i, :] = mask_token + pos[mask_idx[b,i], :]
__global__ void fill_mask_rows_kernel(float* __restrict__ xfull, const float* __restrict__ mask_token, const float* __restrict__ pos,
                                      const int* __restrict__ mask_idx, int nmask, int nvis, int D) {
  const int row = blockIdx.x;  // over B*nmask
  const int b = row / nmask, i = row % nmask;
  const float* p = pos + (size_t)mask_idx[row] * D;
  float* o = xfull + ((size_t)b * (nvis + nmask) + nvis + i) * D;
  for (int c = threadIdx.x; c < (D >> 2); c += blockDim.x) ((f32x4*)o)[c] = ((const f32x4*)mask_token)[c] + ((const f32x4*)p)[c];
}

int fill_mask_rows(float* xfull, const float* mask_token, const float* pos, const int* mask_idx, int B, int nmask, int nvis, int D, hipStream_t st) {
  if (B <= 0 || nmask <= 0 || nvis < 0 || (D & 3)) { set_error("fill_mask_rows: bad shape"); return SMBV_ERR_SHAPE; }
  hipLaunchKernelGGL(fill_mask_rows_kernel, dim3(B * nmask), dim3(128), 0, st, xfull, mask_token, pos, mask_idx, nmask, nvis, D);
  return check_launch("fill_mask_rows");
}

// out bf16 [rows][D] = in bf16 [map(row)][D], map(row) = (row / grp_in) * grp_out + row % grp_in + row_off
__global__ void gather_rows_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int D8, long grp_in, long grp_out, long row_off) {
  const long row = blockIdx.x;
  const long src = (row / grp_in) * grp_out + (row % grp_in) + row_off;
  for (int c = threadIdx.x; c < D8; c += blockDim.x) ((u32x4*)(out + row * D8 * 8))[c] = ((const u32x4*)(in + src * D8 * 8))[c];
}

int gather_rows_bf16(const bf16_t* in, bf16_t* out, int rows, int D, long grp_in, long grp_out, long row_off, hipStream_t st) {
  if (rows <= 0 || (D & 7) || grp_in <= 0) { set_error("gather_rows_bf16: bad shape"); return SMBV_ERR_SHAPE; }
  if (((uintptr_t)in | (uintptr_t)out) & 15) { set_error("gather_rows_bf16: alignment"); return SMBV_ERR_ALIGN; }
  hipLaunchKernelGGL(gather_rows_kernel, dim3(rows), dim3(64), 0, st, in, out, D >> 3, grp_in, grp_out, row_off);
  return check_launch("gather_rows_bf16");
}

// Deterministic column reduction of a partial-sum matrix part[nrows][ncols] -> out0[0:split) | out1[split:ncols).
// 256 threads = 8 column lanes (float4 each -> 32 columns per block) x 32 row lanes; a thread's rows (r, r+32, ...) are loaded four at
// a time (the partial matrices have <= 64..128 rows: with one load in flight this kernel was pure latency), combined through LDS in a
// fixed order.
__global__ __launch_bounds__(256) void reduce_cols_kernel(const float* __restrict__ part, int nrows, int ncols, float* __restrict__ out0,
                                                          float* __restrict__ out1, int split) {
  __shared__ f32x4 red[32][8];
  const int cl = threadIdx.x & 7, rl = threadIdx.x >> 3;
  const int c4 = blockIdx.x * 8 + cl;  // float4 column index
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (c4 * 4 < ncols) {
    const float* p = part + 4 * c4;
    int r = rl;
    for (; r + 96 < nrows; r += 128) {
      const f32x4 a = *(const f32x4*)(p + (size_t)r * ncols), b = *(const f32x4*)(p + (size_t)(r + 32) * ncols);
      const f32x4 c = *(const f32x4*)(p + (size_t)(r + 64) * ncols), d = *(const f32x4*)(p + (size_t)(r + 96) * ncols);
      acc += a; acc += b; acc += c; acc += d;
    }
    for (; r < nrows; r += 32) acc += *(const f32x4*)(p + (size_t)r * ncols);
  }
  red[rl][cl] = acc;
  __syncthreads();
  if (rl == 0 && c4 * 4 < ncols) {
#pragma unroll
    for (int i = 1; i < 32; i++) acc += red[i][cl];
    const int c = 4 * c4;
    if (c < split) *(f32x4*)(out0 + c) = acc;
    else *(f32x4*)(out1 + (c - split)) = acc;
  }
}

int reduce_cols(const float* part, int nrows, int ncols, float* out0, float* out1, int split, hipStream_t st) {
  if ((ncols & 3) || (split & 3)) { set_error("reduce_cols: ncols/split must be multiples of 4"); return SMBV_ERR_SHAPE; }
  hipLaunchKernelGGL(reduce_cols_kernel, dim3(cdiv(ncols, 32)), dim3(256), 0, st, part, nrows, ncols, out0, out1, split);
  return check_launch("reduce_cols");
}

// column sums over rows (bias gradients, mask_token gradient): out[n] = sum_m X[map(m)][n]; two deterministic stages.
// Stage 1: block = 16 column groups (8 columns = 16 B of bf16 / 2 x 16 B of fp32 each -> 128 columns) x 16 row lanes.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ X, int ld, int M, int N, float* __restrict__ part, int mchunk,
                                                     long grp_in, long grp_out, long row_off) {
  __shared__ float red[16][129];
  const int cg = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int n0 = blockIdx.x * 128 + cg * 8;
  const int z = blockIdx.y;
  const int m0 = z * mchunk, m1 = min(M, m0 + mchunk);
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (n0 < N) {
    auto src = [&](int m) -> const T* {
      size_t r = m;
      if (grp_in > 0) r = (size_t)(m / grp_in) * grp_out + (m % grp_in) + row_off;
      return X + r * ld + n0;
    };
    auto add = [&](const u32x4* v) {
      if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int i = 0; i < 4; i++) { a[2 * i] += bf_lo(v[0][i]); a[2 * i + 1] += bf_hi(v[0][i]); }
      } else {
        const f32x4 v0 = __builtin_bit_cast(f32x4, v[0]), v1 = __builtin_bit_cast(f32x4, v[1]);
#pragma unroll
        for (int i = 0; i < 4; i++) { a[i] += v0[i]; a[4 + i] += v1[i]; }
      }
    };
    constexpr int W = sizeof(T) == 2 ? 1 : 2;  // 16-byte words per 8 columns
    constexpr int U = 8;                       // rows in flight per thread (the loop is latency-bound otherwise)
    int m = m0 + rl;
    for (; m + 16 * (U - 1) < m1; m += 16 * U) {
      u32x4 v[U][W];
#pragma unroll
      for (int u = 0; u < U; u++)
#pragma unroll
        for (int k = 0; k < W; k++) v[u][k] = ((const u32x4*)src(m + 16 * u))[k];
#pragma unroll
      for (int u = 0; u < U; u++) add(v[u]);  // fixed order: the sum is reproducible
    }
    for (; m < m1; m += 16) {
      u32x4 v[W];
#pragma unroll
      for (int k = 0; k < W; k++) v[k] = ((const u32x4*)src(m))[k];
      add(v);
    }
  }
#pragma unroll
  for (int i = 0; i < 8; i++) red[rl][cg * 8 + i] = a[i];
  __syncthreads();
  if (threadIdx.x < 128) {
    const int n = blockIdx.x * 128 + threadIdx.x;
    if (n < N) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 16; i++) s += red[i][threadIdx.x];
      part[(size_t)z * N + n] = s;
    }
  }
}

static int colsum_splits(int M, int N) {
  const int cb = cdiv(N, 128);
  return std::max(1, std::min(std::min(cdiv(M, 128), 64), cdiv(768, cb)));
}
size_t colsum_workspace_bytes(int M, int N) { return (size_t)colsum_splits(M, N) * N * sizeof(float); }

template <typename T>
static int colsum_impl(const T* X, int ld, int M, int N, float* out, long grp_in, long grp_out, long row_off, void* ws, size_t ws_bytes, hipStream_t st) {
  if (M <= 0 || N <= 0 || (N & 7) || (ld & 7)) { set_error("colsum: need M,N > 0 and N, ld multiples of 8 (M=%d N=%d ld=%d)", M, N, ld); return SMBV_ERR_SHAPE; }
  if ((uintptr_t)X & 15) { set_error("colsum: alignment"); return SMBV_ERR_ALIGN; }
  const int nz = colsum_splits(M, N);
  if (!ws || ws_bytes < (size_t)nz * N * sizeof(float)) { set_error("colsum: workspace too small"); return SMBV_ERR_WORKSPACE; }
  const int mchunk = cdiv(M, nz);
  hipLaunchKernelGGL(colsum_kernel<T>, dim3(cdiv(N, 128), nz), dim3(256), 0, st, X, ld, M, N, (float*)ws, mchunk, grp_in, grp_out, row_off);
  int rc = check_launch("colsum");
  if (rc) return rc;
  return reduce_cols((const float*)ws, nz, N, out, out, N, st);
}

int colsum_bf16(const bf16_t* X, int ld, int M, int N, float* out, long grp_in, long grp_out, long row_off, void* ws, size_t ws_bytes, hipStream_t st) {
  return colsum_impl<bf16_t>(X, ld, M, N, out, grp_in, grp_out, row_off, ws, ws_bytes, st);
}
int colsum_f32(const float* X, int ld, int M, int N, float* out, long grp_in, long grp_out, long row_off, void* ws, size_t ws_bytes, hipStream_t st) {
  return colsum_impl<float>(X, ld, M, N, out, grp_in, grp_out, row_off, ws, ws_bytes, st);
}

}  // namespace smbv
