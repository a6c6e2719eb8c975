// Small kernels of the V-JEPA step (src/run_vjepa.py / src/models/vjepa/modeling_vjepa.py) besides the transformer blocks:
//   rope3d          3-axis rotary embedding of the q and k sections of the fused QKV matrix, in place (modeling_vjepa.py:204-228,
//                   :277-326), and its transpose for the backward.  NOTE the reference's quirk, reproduced exactly: pairs are
//                   interleaved (2i, 2i+1) but cos/sin are TILED ([f0..f_{n-1}, f0..f_{n-1}]), so the two elements of a pair use
//                   different angles and the map is not a rotation: the backward is the transposed map, not rotate(-theta).
//   gather/scatter  apply_masks (:542-556), predictor token assembly in position order (:602-619, :718-719) and its inverse (:737-738)
//   l1              nn.L1Loss (run_vjepa.py:108,136) forward + backward
//   ema             MomentumEncoder.update (run_vjepa.py:94-98) as one multi-tensor launch
#include "smbv_common.h"
#include "smbv_kernels.h"
#include "../../include/smbv.h"

namespace smbv {

// One thread per (token, head, 16-byte chunk): 8 elements = 4 rotation pairs of q and the same 4 of k (one table lookup serves both).
// qkv: bf16 [rows, ld]; q at column 0, k at column k_off; head stride hs (64 when padded); elements >= 3 * seg of a head pass through.
// pos: int32 token position ids [rows] (null -> row % S).  table: fp32 [P][2][seg/2] cos|sin for positions 0..P-1.
// (Round 3: the first form - one thread per pair, 4-byte accesses - moved 1.7 TB/s; V-JEPA runs it 36 times per step.)
__global__ __launch_bounds__(256) void rope3d_kernel(bf16_t* __restrict__ qkv, int ld, int k_off, long rows, int S, int H, int hs, int seg, int grid,
                                                     const int* __restrict__ pos, const float* __restrict__ table, int transpose, int nchunk) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int tpr = H * nchunk;  // threads per token
  const long row = gid / tpr;
  if (row >= rows) return;
  const int rem = (int)(gid - row * tpr);
  const int h = rem / nchunk, ch = rem - h * nchunk;
  const int p = pos ? pos[row] : (int)(row % S);
  const int tpf = grid * grid;
  const int frame = p / tpf, height = (p - tpf * frame) / grid, width = (p - tpf * frame) - grid * height;
  const int half = seg >> 1;
  u32x4* pq = (u32x4*)(qkv + row * ld + h * hs + 8 * ch);
  u32x4* pk = (u32x4*)(qkv + row * ld + k_off + h * hs + 8 * ch);
  u32x4 q = *pq, k = *pk;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int e0 = 8 * ch + 2 * j;                   // the pair's first element inside the head
    if (e0 >= 3 * seg) break;                        // pass-through tail (seg is even: a pair never straddles the end)
    const int axis = e0 / seg;                       // 0 frame, 1 height, 2 width
    const int t0 = e0 - axis * seg;                  // element index inside its segment
    const int pid = axis == 0 ? frame : (axis == 1 ? height : width);
    const float* tb = table + (size_t)pid * seg;     // [cos half | sin half]
    const float c0 = tb[t0 % half], s0 = tb[half + t0 % half];
    const float c1 = tb[(t0 + 1) % half], s1 = tb[half + (t0 + 1) % half];
#pragma unroll
    for (int sec = 0; sec < 2; sec++) {
      const unsigned u = sec ? k[j] : q[j];
      const float x0 = bf_lo(u), x1 = bf_hi(u);
      float y0, y1;
      rope_pair(x0, x1, c0, s0, c1, s1, transpose != 0, y0, y1);
      if (sec) k[j] = pack_bf16(y0, y1); else q[j] = pack_bf16(y0, y1);
    }
  }
  *pq = q;
  *pk = k;
}

int rope3d(bf16_t* qkv, int ld, int k_off, long rows, int S, int H, int hs, int Dh, int grid, const int* pos, const float* table, int transpose,
           hipStream_t st) {
  const int seg = 2 * ((Dh / 3) / 2);
  if (seg <= 0) return SMBV_OK;
  if (rows <= 0 || H <= 0 || grid <= 0 || (ld & 7) || (k_off & 7) || (hs & 7) || !table || ((uintptr_t)qkv & 15)) { set_error("rope3d: bad argument"); return SMBV_ERR_ARG; }
  const int nchunk = cdiv(3 * seg, 8);               // 16-byte chunks of a head that hold rotated elements (<= hs / 8)
  if (8 * nchunk > hs) { set_error("rope3d: head slot %d narrower than the rotated width", hs); return SMBV_ERR_ARG; }
  const long total = rows * H * nchunk;
  hipLaunchKernelGGL(rope3d_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, qkv, ld, k_off, rows, S, H, hs, seg, grid, pos, table, transpose, nchunk);
  return check_launch("rope3d");
}

// dst row r <- src row idx[r]  (gather);  scatter: dst row idx[r] <- src row r.  T = float or bf16_t; D multiple of 4 (f32) / 8 (bf16)
template <typename T, bool SCATTER>
__global__ void rows_idx_kernel(const T* __restrict__ src, T* __restrict__ dst, const int* __restrict__ idx, int per_batch_idx, long src_batch_rows,
                                long dst_batch_rows, int D16) {
  const long r = blockIdx.x;
  const long b = r / per_batch_idx;
  const long i = idx[r];
  const u32x4* s = (const u32x4*)(src + (SCATTER ? r : b * src_batch_rows + i) * (size_t)D16 * (16 / sizeof(T)));
  u32x4* d = (u32x4*)(dst + (SCATTER ? b * dst_batch_rows + i : r) * (size_t)D16 * (16 / sizeof(T)));
  for (int c = threadIdx.x; c < D16; c += blockDim.x) d[c] = s[c];
}

// dst row idx[r] += src row r, bf16 (one row per block, rows distinct within a launch: plain read-modify-write, reproducible)
__global__ void rows_idx_add_bf16_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, const int* __restrict__ idx, int D8) {
  const long r = blockIdx.x;
  const u32x4* s = (const u32x4*)(src + r * (size_t)D8 * 8);
  u32x4* d = (u32x4*)(dst + (size_t)idx[r] * D8 * 8);
  for (int c = threadIdx.x; c < D8; c += blockDim.x) {
    const u32x4 a = s[c], b = d[c];
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; e++) o[e] = pack_bf16(bf_lo(a[e]) + bf_lo(b[e]), bf_hi(a[e]) + bf_hi(b[e]));
    d[c] = o;
  }
}

// dst (bf16) += src (fp32), elementwise, n % 8 == 0
__global__ void add_f32_into_bf16_kernel(bf16_t* __restrict__ dst, const float* __restrict__ src, size_t n8) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    const u32x4 d = ((const u32x4*)dst)[i];
    const f32x4 a = ((const f32x4*)src)[2 * i], b = ((const f32x4*)src)[2 * i + 1];
    const u32x4 o = {pack_bf16(bf_lo(d[0]) + a[0], bf_hi(d[0]) + a[1]), pack_bf16(bf_lo(d[1]) + a[2], bf_hi(d[1]) + a[3]),
                     pack_bf16(bf_lo(d[2]) + b[0], bf_hi(d[2]) + b[1]), pack_bf16(bf_lo(d[3]) + b[2], bf_hi(d[3]) + b[3])};
    ((u32x4*)dst)[i] = o;
  }
}
int add_f32_into_bf16(bf16_t* dst, const float* src, size_t n, hipStream_t st) {
  if ((n & 7) || (((uintptr_t)dst | (uintptr_t)src) & 15)) { set_error("add_f32_into_bf16: n %% 8 / alignment"); return SMBV_ERR_ALIGN; }
  hipLaunchKernelGGL(add_f32_into_bf16_kernel, dim3((int)std::min<size_t>(2048, (n / 8 + 255) / 256)), dim3(256), 0, st, dst, src, n / 8);
  return check_launch("add_f32_into_bf16");
}

int scatter_add_rows_idx_bf16(const bf16_t* src, bf16_t* dst, const int* idx, long nidx, int D, hipStream_t st) {
  if (nidx <= 0 || (D & 7) || (((uintptr_t)src | (uintptr_t)dst) & 15)) { set_error("scatter_add_rows_idx_bf16: bad shape/alignment"); return SMBV_ERR_SHAPE; }
  hipLaunchKernelGGL(rows_idx_add_bf16_kernel, dim3(nidx), dim3(64), 0, st, src, dst, idx, D / 8);
  return check_launch("rows_idx_add");
}

template <typename T>
static int rows_idx(const T* src, T* dst, const int* idx, long nidx, int per_batch, long src_batch_rows, long dst_batch_rows, int D, int scatter,
                    hipStream_t st) {
  const int per16 = 16 / (int)sizeof(T);
  if (nidx <= 0 || per_batch <= 0 || D % per16 || (((uintptr_t)src | (uintptr_t)dst) & 15)) { set_error("rows_idx: bad shape/alignment"); return SMBV_ERR_SHAPE; }
  if (scatter) hipLaunchKernelGGL((rows_idx_kernel<T, true>), dim3(nidx), dim3(64), 0, st, src, dst, idx, per_batch, src_batch_rows, dst_batch_rows, D / per16);
  else hipLaunchKernelGGL((rows_idx_kernel<T, false>), dim3(nidx), dim3(64), 0, st, src, dst, idx, per_batch, src_batch_rows, dst_batch_rows, D / per16);
  return check_launch("rows_idx");
}

int gather_rows_idx_bf16(const bf16_t* src, bf16_t* dst, const int* idx, long nidx, int per_batch, long src_batch_rows, int D, hipStream_t st) {
  return rows_idx<bf16_t>(src, dst, idx, nidx, per_batch, src_batch_rows, 0, D, 0, st);
}
int scatter_rows_idx_bf16(const bf16_t* src, bf16_t* dst, const int* idx, long nidx, int per_batch, long dst_batch_rows, int D, hipStream_t st) {
  return rows_idx<bf16_t>(src, dst, idx, nidx, per_batch, 0, dst_batch_rows, D, 1, st);
}
int scatter_rows_idx_f32(const float* src, float* dst, const int* idx, long nidx, int per_batch, long dst_batch_rows, int D, hipStream_t st) {
  return rows_idx<float>(src, dst, idx, nidx, per_batch, 0, dst_batch_rows, D, 1, st);
}
int gather_rows_idx_f32(const float* src, float* dst, const int* idx, long nidx, int per_batch, long src_batch_rows, int D, hipStream_t st) {
  return rows_idx<float>(src, dst, idx, nidx, per_batch, src_batch_rows, 0, D, 0, st);
}

// dst row idx[r] (per batch) <- vec  (mask-token rows of the predictor input)
__global__ void fill_rows_idx_kernel(float* __restrict__ dst, const float* __restrict__ vec, const int* __restrict__ idx, int per_batch,
                                     long dst_batch_rows, int D4) {
  const long r = blockIdx.x;
  f32x4* d = (f32x4*)(dst + ((r / per_batch) * dst_batch_rows + idx[r]) * (size_t)D4 * 4);
  for (int c = threadIdx.x; c < D4; c += blockDim.x) d[c] = ((const f32x4*)vec)[c];
}

int fill_rows_idx_f32(float* dst, const float* vec, const int* idx, long nidx, int per_batch, long dst_batch_rows, int D, hipStream_t st) {
  if (!dst || !vec || !idx || nidx <= 0 || (D & 3)) { set_error("fill_rows_idx_f32: bad argument"); return SMBV_ERR_ARG; }
  hipLaunchKernelGGL(fill_rows_idx_kernel, dim3(nidx), dim3(64), 0, st, dst, vec, idx, per_batch, dst_batch_rows, D / 4);
  return check_launch("fill_rows_idx");
}

// dst[i] = src[(i / grp_in) * grp_out + i % grp_in]   (un-padding of head-padded gradients)
__global__ void map_copy_kernel(float* __restrict__ dst, const float* __restrict__ src, size_t n, long grp_in, long grp_out) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[(i / grp_in) * grp_out + i % grp_in];
}
int map_copy_f32(float* dst, const float* src, size_t n, long grp_in, long grp_out, hipStream_t st) {
  if (!n) return SMBV_OK;
  hipLaunchKernelGGL(map_copy_kernel, dim3((unsigned)std::min<size_t>(2048, (n + 255) / 256)), dim3(256), 0, st, dst, src, n, grp_in, grp_out);
  return check_launch("map_copy");
}

template <bool BWD>
__global__ __launch_bounds__(256) void l1_kernel(const float* __restrict__ pred, const float* __restrict__ tgt, size_t n4, float* __restrict__ part,
                                                 float* __restrict__ dpred, const float* __restrict__ gscale, float coef) {
  __shared__ float red[4];
  float s = 0.f;
  float gs = coef;
  if (BWD && gscale) gs *= *gscale;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const f32x4 d = ((const f32x4*)pred)[i] - ((const f32x4*)tgt)[i];
    if (BWD) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; e++) o[e] = d[e] > 0.f ? gs : (d[e] < 0.f ? -gs : 0.f);   // sign(d) * g / n, sign(0) = 0 as torch
      ((f32x4*)dpred)[i] = o;
    } else {
      s += (fabsf(d[0]) + fabsf(d[1])) + (fabsf(d[2]) + fabsf(d[3]));
    }
  }
  if (!BWD) {
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
  }
}

__global__ __launch_bounds__(256) void sum_scale2_kernel(const float* __restrict__ part, int n, float coef, float* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += part[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) *out = (red[0] + red[1] + red[2] + red[3]) * coef;
}

__global__ __launch_bounds__(256) void ema_kernel(const smbv_adam_chunk* __restrict__ chunks, float momentum) {
  const smbv_adam_chunk c = chunks[blockIdx.x];   // p = target (k), g = online (q)
  const float om = 1.0f - momentum;
  const int n4 = c.n >> 2;
  for (int i = threadIdx.x; i < n4; i += 256) ((f32x4*)c.p)[i] = ((f32x4*)c.p)[i] * momentum + ((const f32x4*)c.g)[i] * om;
  for (int i = (n4 << 2) + threadIdx.x; i < c.n; i += 256) c.p[i] = c.p[i] * momentum + c.g[i] * om;
}

}  // namespace smbv

using namespace smbv;

extern "C" {

int smbv_rope3d(smbv_bf16* qkv, int ld, int k_off, long rows, int S, int H, int head_stride, int Dh, int grid, const int32_t* pos, const float* table,
                int transpose, smbv_stream stream) {
  if (!qkv) { set_error("smbv_rope3d: null operand"); return SMBV_ERR_ARG; }
  return rope3d(qkv, ld, k_off, rows, S, H, head_stride, Dh, grid, pos, table, transpose, (hipStream_t)stream);
}
int smbv_gather_rows_idx_f32(const float* src, float* dst, const int32_t* idx, long nidx, int per_batch, long src_batch_rows, int D, smbv_stream stream) {
  return rows_idx<float>(src, dst, idx, nidx, per_batch, src_batch_rows, 0, D, 0, (hipStream_t)stream);
}
int smbv_scatter_rows_idx_f32(const float* src, float* dst, const int32_t* idx, long nidx, int per_batch, long dst_batch_rows, int D, smbv_stream stream) {
  return rows_idx<float>(src, dst, idx, nidx, per_batch, 0, dst_batch_rows, D, 1, (hipStream_t)stream);
}
int smbv_fill_rows_idx_f32(float* dst, const float* vec, const int32_t* idx, long nidx, int per_batch, long dst_batch_rows, int D, smbv_stream stream) {
  return fill_rows_idx_f32(dst, vec, idx, nidx, per_batch, dst_batch_rows, D, (hipStream_t)stream);
}
int smbv_l1_fwd(const float* pred, const float* target, size_t n, float* partial /*[1024]*/, float* loss, smbv_stream stream) {
  if (!pred || !target || !partial || !loss || n == 0 || (n & 3)) { set_error("smbv_l1_fwd: bad argument (n %% 4)"); return SMBV_ERR_ARG; }
  const int nb = (int)std::min<size_t>(1024, (n / 4 + 255) / 256);
  hipLaunchKernelGGL(l1_kernel<false>, dim3(nb), dim3(256), 0, (hipStream_t)stream, pred, target, n / 4, partial, (float*)nullptr, (const float*)nullptr, 0.f);
  int rc = check_launch("l1_fwd");
  if (rc) return rc;
  hipLaunchKernelGGL(sum_scale2_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)partial, nb, 1.0f / (float)n, loss);
  return check_launch("l1_reduce");
}
int smbv_l1_bwd(const float* pred, const float* target, size_t n, float* dpred, const float* gscale, smbv_stream stream) {
  if (!pred || !target || !dpred || n == 0 || (n & 3)) { set_error("smbv_l1_bwd: bad argument"); return SMBV_ERR_ARG; }
  const int nb = (int)std::min<size_t>(2048, (n / 4 + 255) / 256);
  hipLaunchKernelGGL(l1_kernel<true>, dim3(nb), dim3(256), 0, (hipStream_t)stream, pred, target, n / 4, (float*)nullptr, dpred, gscale, 1.0f / (float)n);
  return check_launch("l1_bwd");
}
int smbv_ema_update(const smbv_adam_chunk* chunks_dev, int nchunks, float momentum, smbv_stream stream) {
  if (!chunks_dev || nchunks <= 0) { set_error("smbv_ema_update: bad argument"); return SMBV_ERR_ARG; }
  hipLaunchKernelGGL(ema_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, chunks_dev, momentum);
  return check_launch("ema");
}

}  // extern "C"
