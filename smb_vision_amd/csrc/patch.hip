// Brick (3-D patch) readers over the volume pixel_values fp32 [B, T, 1, H, W]  (HBM-bound).
//
//   gather_bricks : selected tokens' ts x ps x ps bricks -> bf16 rows [rows, K] in Conv3d-weight order
//                   k = (kt*ps + kh)*ps + kw  (src/models/videomae/modeling_videomae.py:172-177,190-191: Conv3d with
//                   kernel == stride is a GEMM over bricks; token order (T',H',W') row-major = flatten(2)).
//                   Only the visible bricks are read: the reference computes all N and discards the masked rows (:134-137).
//   brick_stats   : per-brick mean and 1/(sqrt(var_unbiased)+1e-6) of the masked bricks (:859-861, norm_pix_loss).
//   mse_fwd/bwd   : mean((logits - label)^2) over all B*N_mask*K elements (:896-897) with the label
//                   (x - mean) * rinv recomputed from the fp32 volume (per-patch order (ts,ps,ps,C) == k for C = 1), so the
//                   218 MB label tensor is never materialised.  Deterministic two-stage reduction.
// A brick row (fixed kt,kh) is ps contiguous floats (64 B at ps = 16): every thread moves one float4, and a block sweeps TWO list-adjacent bricks
// row by row so that W-adjacent bricks are read as whole 128-B lines (below).
#include "smbv_common.h"
#include "smbv_kernels.h"

namespace smbv {

struct BrickGeom {
  int T, H, W, ts, ps, gh, gw, N, K;
};

__device__ __forceinline__ size_t brick_elem_off(const BrickGeom& g, int b, int tok, int k) {
  const int dq = tok / (g.gh * g.gw), hq = (tok / g.gw) % g.gh, wq = tok % g.gw;
  const int kw = k % g.ps, kh = (k / g.ps) % g.ps, kt = k / (g.ps * g.ps);
  return (((size_t)b * g.T + dq * g.ts + kt) * g.H + hq * g.ps + kh) * g.W + wq * g.ps + kw;
}

// Two bricks per block (round 5).  A brick row (fixed kt, kh) is ps contiguous floats - 64 B at ps = 16, HALF a 128-B line - and the token lists are
// ascending, so list neighbours are usually W-neighbours in the volume (always, in pairs, under the 32-voxel mask blocks of the shipped recipe: a mask
// block covers 2 x 2 x 2 bricks).  Slot s of a block's sweep is (row pair-row s / 2q, brick (s % 2q) / q, float4 (s % 2q) % q), q = ps / 4: the 2q lanes
// of one pair-row read brick A's row then brick B's - ONE whole line when B is A's right neighbour, two half lines (what every request was before)
// when it is not.  Works for any list; nothing is assumed about adjacency.
struct PairSlot { int brick, k; };
__device__ __forceinline__ PairSlot pair_slot(int s, int q, int ps) {
  const int q2 = 2 * q, u = s % q2;
  PairSlot r;
  r.brick = u >= q ? 1 : 0;
  r.k = (s / q2) * ps + (u - r.brick * q) * 4;
  return r;
}
__device__ __forceinline__ void block_sum2_256(float& a, float& b, float (*red)[2]) {
  a = wave_sum(a);
  b = wave_sum(b);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { red[w][0] = a; red[w][1] = b; }
  __syncthreads();
  a = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
  b = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
}

__global__ __launch_bounds__(256) void gather_bricks_kernel(const float* __restrict__ pix, const int* __restrict__ idx, int rows, int per_batch,
                                                            bf16_t* __restrict__ out, BrickGeom g) {
  const int row0 = 2 * blockIdx.x;
  const bool two = row0 + 1 < rows;
  const int tok[2] = {idx[row0], two ? idx[row0 + 1] : 0};
  const int bb[2] = {row0 / per_batch, (row0 + 1) / per_batch};
  const int q = g.ps >> 2, slots = 2 * (g.K >> 2);
  for (int s = threadIdx.x; s < slots; s += 256) {
    const PairSlot ps = pair_slot(s, q, g.ps);
    if (ps.brick && !two) continue;
    const f32x4 v = *(const f32x4*)(pix + brick_elem_off(g, bb[ps.brick], tok[ps.brick], ps.k));
    u32x2 pk = {pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
    *(u32x2*)(out + (size_t)(row0 + ps.brick) * g.K + ps.k) = pk;
  }
}

__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// One pass over the bricks: a thread keeps its float4s in registers between the mean and the variance sweep (2 K / 4 <= 256 * RES slots: the path's
// 16^3 bricks fit RES = 8; larger bricks re-read the tail from L2).  Two-pass arithmetic (mean, then centred squares), per brick.
template <int RES>
__global__ __launch_bounds__(256) void brick_stats_kernel(const float* __restrict__ pix, const int* __restrict__ idx, int rows, int per_batch,
                                                          float* __restrict__ mean, float* __restrict__ rinv, BrickGeom g) {
  __shared__ float red[4][2];
  const int row0 = 2 * blockIdx.x;
  const bool two = row0 + 1 < rows;
  const int tok[2] = {idx[row0], two ? idx[row0 + 1] : 0};
  const int bb[2] = {row0 / per_batch, (row0 + 1) / per_batch};
  const int q = g.ps >> 2, slots = 2 * (g.K >> 2);
  f32x4 keep[RES];
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int i = 0; i < RES; i++) {
    const int s = threadIdx.x + 256 * i;
    const PairSlot ps = pair_slot(s, q, g.ps);
    const bool on = s < slots && (two || !ps.brick);
    keep[i] = on ? *(const f32x4*)(pix + brick_elem_off(g, bb[ps.brick], tok[ps.brick], ps.k)) : (f32x4){0.f, 0.f, 0.f, 0.f};
    const float t = (keep[i][0] + keep[i][1]) + (keep[i][2] + keep[i][3]);
    s0 += ps.brick ? 0.f : t;
    s1 += ps.brick ? t : 0.f;
  }
  for (int s = threadIdx.x + 256 * RES; s < slots; s += 256) {
    const PairSlot ps = pair_slot(s, q, g.ps);
    if (ps.brick && !two) continue;
    const f32x4 v = *(const f32x4*)(pix + brick_elem_off(g, bb[ps.brick], tok[ps.brick], ps.k));
    const float t = (v[0] + v[1]) + (v[2] + v[3]);
    s0 += ps.brick ? 0.f : t;
    s1 += ps.brick ? t : 0.f;
  }
  block_sum2_256(s0, s1, red);
  const float mu[2] = {s0 / g.K, s1 / g.K};
  float q0 = 0.f, q1 = 0.f;
#pragma unroll
  for (int i = 0; i < RES; i++) {
    const int s = threadIdx.x + 256 * i;
    const PairSlot ps = pair_slot(s, q, g.ps);
    if (s < slots && (two || !ps.brick)) {
      const f32x4 v = keep[i] - mu[ps.brick];
      const float t = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
      q0 += ps.brick ? 0.f : t;
      q1 += ps.brick ? t : 0.f;
    }
  }
  for (int s = threadIdx.x + 256 * RES; s < slots; s += 256) {
    const PairSlot ps = pair_slot(s, q, g.ps);
    if (ps.brick && !two) continue;
    const f32x4 v = *(const f32x4*)(pix + brick_elem_off(g, bb[ps.brick], tok[ps.brick], ps.k)) - mu[ps.brick];
    const float t = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    q0 += ps.brick ? 0.f : t;
    q1 += ps.brick ? t : 0.f;
  }
  block_sum2_256(q0, q1, red);
  if (threadIdx.x < 2 && (threadIdx.x == 0 || two)) {
    const float var = (threadIdx.x ? q1 : q0) / (g.K - 1);  // unbiased
    mean[row0 + threadIdx.x] = mu[threadIdx.x];
    rinv[row0 + threadIdx.x] = 1.0f / (sqrtf(var) + 1e-6f);
  }
}

template <bool BWD>
__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ pix, const int* __restrict__ idx, int rows, int per_batch,
                                                  const float* __restrict__ mean, const float* __restrict__ rinv,
                                                  const bf16_t* __restrict__ logits, float* __restrict__ row_partial,
                                                  bf16_t* __restrict__ dlogits, const float* __restrict__ gscale, float coef,
                                                  BrickGeom g) {
  __shared__ float red[4][2];
  const int row0 = 2 * blockIdx.x;
  const bool two = row0 + 1 < rows;
  const int tok[2] = {idx[row0], two ? idx[row0 + 1] : 0};
  const int bb[2] = {row0 / per_batch, (row0 + 1) / per_batch};
  const float mu[2] = {mean[row0], two ? mean[row0 + 1] : 0.f}, ri[2] = {rinv[row0], two ? rinv[row0 + 1] : 0.f};
  const int q = g.ps >> 2, slots = 2 * (g.K >> 2);
  float s0 = 0.f, s1 = 0.f;
  float gs = coef;
  if (BWD && gscale) gs *= *gscale;
  for (int s = threadIdx.x; s < slots; s += 256) {
    const PairSlot ps = pair_slot(s, q, g.ps);
    if (ps.brick && !two) continue;
    const size_t o = (size_t)(row0 + ps.brick) * g.K + ps.k;
    const f32x4 v = (*(const f32x4*)(pix + brick_elem_off(g, bb[ps.brick], tok[ps.brick], ps.k)) - mu[ps.brick]) * ri[ps.brick];
    const u32x2 pk = *(const u32x2*)(logits + o);
    const f32x4 d = (f32x4){bf_lo(pk[0]), bf_hi(pk[0]), bf_lo(pk[1]), bf_hi(pk[1])} - v;
    if (BWD) {
      const f32x4 od = d * gs;
      u32x2 po = {pack_bf16(od[0], od[1]), pack_bf16(od[2], od[3])};
      *(u32x2*)(dlogits + o) = po;
    } else {
      const float t = (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
      s0 += ps.brick ? 0.f : t;
      s1 += ps.brick ? t : 0.f;
    }
  }
  if (!BWD) {
    block_sum2_256(s0, s1, red);
    if (threadIdx.x == 0) row_partial[row0] = s0;
    if (threadIdx.x == 1 && two) row_partial[row0 + 1] = s1;
  }
}

// one block of 1,024 threads, four independent loads in flight per thread (a 256-thread block walking 13,312 partials one dependent load at a time took 17 us
// on the step's critical path); fixed summation order: run-to-run identical
__global__ __launch_bounds__(1024) void sum_scale_kernel(const float* __restrict__ part, int n, float coef, float* __restrict__ out) {
  __shared__ float red[16];
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int i = threadIdx.x;
  for (; i + 3072 < n; i += 4096) {
    const float v0 = part[i], v1 = part[i + 1024], v2 = part[i + 2048], v3 = part[i + 3072];
    a0 += v0; a1 += v1; a2 += v2; a3 += v3;
  }
  for (; i < n; i += 1024) a0 += part[i];
  const float v = wave_sum((a0 + a1) + (a2 + a3));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < 16; w++) tot += red[w];
    *out = tot * coef;
  }
}

static int make_geom(BrickGeom& g, int T, int H, int W, int ts, int ps, const char* who) {
  if (T <= 0 || H <= 0 || W <= 0 || ts <= 0 || ps <= 0 || T % ts || H % ps || W % ps || (ps & 3)) {
    set_error("%s: volume %dx%dx%d not divisible by tubelet %d / patch %d (patch must be a multiple of 4)", who, T, H, W, ts, ps);
    return SMBV_ERR_SHAPE;
  }
  g.T = T; g.H = H; g.W = W; g.ts = ts; g.ps = ps; g.gh = H / ps; g.gw = W / ps;
  g.N = (T / ts) * g.gh * g.gw;
  g.K = ts * ps * ps;
  return SMBV_OK;
}

int gather_bricks(const float* pix, const int* idx, int rows, int per_batch, bf16_t* out, int T, int H, int W, int ts, int ps, hipStream_t st) {
  BrickGeom g;
  int rc = make_geom(g, T, H, W, ts, ps, "gather_bricks");
  if (rc) return rc;
  if (rows <= 0 || per_batch <= 0 || rows % per_batch) { set_error("gather_bricks: rows=%d per_batch=%d", rows, per_batch); return SMBV_ERR_SHAPE; }
  if (((uintptr_t)pix & 15) || ((uintptr_t)out & 7)) { set_error("gather_bricks: alignment"); return SMBV_ERR_ALIGN; }
  hipLaunchKernelGGL(gather_bricks_kernel, dim3((rows + 1) / 2), dim3(256), 0, st, pix, idx, rows, per_batch, out, g);
  return check_launch("gather_bricks");
}

int brick_stats(const float* pix, const int* idx, int rows, int per_batch, float* mean, float* rinv, int T, int H, int W, int ts, int ps, hipStream_t st) {
  BrickGeom g;
  int rc = make_geom(g, T, H, W, ts, ps, "brick_stats");
  if (rc) return rc;
  if (rows <= 0 || per_batch <= 0 || rows % per_batch) { set_error("brick_stats: rows=%d per_batch=%d", rows, per_batch); return SMBV_ERR_SHAPE; }
  hipLaunchKernelGGL(brick_stats_kernel<8>, dim3((rows + 1) / 2), dim3(256), 0, st, pix, idx, rows, per_batch, mean, rinv, g);
  return check_launch("brick_stats");
}

int mse_fwd(const float* pix, const int* idx, int rows, int per_batch, const float* mean, const float* rinv, const bf16_t* logits,
            float* row_partial, float* loss, int T, int H, int W, int ts, int ps, hipStream_t st) {
  BrickGeom g;
  int rc = make_geom(g, T, H, W, ts, ps, "mse_fwd");
  if (rc) return rc;
  hipLaunchKernelGGL(mse_kernel<false>, dim3((rows + 1) / 2), dim3(256), 0, st, pix, idx, rows, per_batch, mean, rinv, logits, row_partial, (bf16_t*)nullptr,
                     (const float*)nullptr, 0.f, g);
  rc = check_launch("mse_fwd");
  if (rc) return rc;
  hipLaunchKernelGGL(sum_scale_kernel, dim3(1), dim3(1024), 0, st, row_partial, rows, 1.0f / ((float)rows * (float)g.K), loss);
  return check_launch("mse_reduce");
}

int mse_bwd(const float* pix, const int* idx, int rows, int per_batch, const float* mean, const float* rinv, const bf16_t* logits,
            bf16_t* dlogits, const float* gscale, int T, int H, int W, int ts, int ps, hipStream_t st) {
  BrickGeom g;
  int rc = make_geom(g, T, H, W, ts, ps, "mse_bwd");
  if (rc) return rc;
  hipLaunchKernelGGL(mse_kernel<true>, dim3((rows + 1) / 2), dim3(256), 0, st, pix, idx, rows, per_batch, mean, rinv, logits, (float*)nullptr, dlogits, gscale,
                     2.0f / ((float)rows * (float)g.K), g);
  return check_launch("mse_bwd");
}

}  // namespace smbv
