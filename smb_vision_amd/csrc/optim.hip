// Fused optimizer tail of the train step: global-L2 gradient norm + clip + AdamW as two multi-tensor
// launches over a chunk table, instead of ~10 foreach launches and an in-place gradient rescale.  HBM-bound: 4 B/param for
// the norm, 28 B/param for the update (read p,g,m,v; write p,m,v).
// Semantics: torch.nn.utils.clip_grad_norm_ (coef = min(1, max_norm / (norm + 1e-6))) followed by torch.optim.AdamW
// (decoupled weight decay p *= 1 - lr*wd; m, v EMAs; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)) — what HF Trainer runs at
// site-packages/transformers/trainer.py:2539 and :1785 for the reference.
#include "smbv_common.h"
#include "smbv_kernels.h"
#include "../../include/smbv.h"

namespace smbv {
extern int g_ab;

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void sqnorm_kernel(const smbv_adam_chunk* __restrict__ chunks, float* __restrict__ partial) {
  __shared__ float red[4];
  const smbv_adam_chunk c = chunks[blockIdx.x];
  float s = 0.f;
  const int n4 = c.n >> 2;
  for (int i = threadIdx.x; i < n4; i += 256) {
    const f32x4 g = ((const f32x4*)c.g)[i];
    s += (g[0] * g[0] + g[1] * g[1]) + (g[2] * g[2] + g[3] * g[3]);
  }
  for (int i = (n4 << 2) + threadIdx.x; i < c.n; i += 256) s += c.g[i] * c.g[i];
  const float tot = block_sum(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ part, int n, float* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += part[i];
  const float tot = block_sum(s, red);
  if (threadIdx.x == 0) *out = tot;
}

// (rev: the chunks in REVERSE table order.  The squared-norm pass in front of this kernel streams the gradient arena front to back, so what the 256-MiB
// Infinity Cache still holds when the update starts is the arena's END: read first, those ~2/3 of the gradients come from the cache instead of HBM while the
// other three streams - p, m, v - take the memory channels; profiles/r05_adamw_order_ab.txt)
__global__ __launch_bounds__(256) void adamw_kernel(const smbv_adam_chunk* __restrict__ chunks, smbv_adam_hyper h, const float* __restrict__ gnorm_sq, int rev) {
  const smbv_adam_chunk c = chunks[rev ? gridDim.x - 1 - blockIdx.x : blockIdx.x];
  float coef = 1.0f;
  if (gnorm_sq && h.max_grad_norm > 0.f) coef = fminf(1.0f, h.max_grad_norm / (sqrtf(*gnorm_sq) + 1e-6f));
  const float lr = h.lr[c.group], wd = h.weight_decay[c.group];
  const float decay = 1.0f - lr * wd, step = lr / h.bias_corr1, rbc2 = rsqrtf(h.bias_corr2);
  const float b1 = h.beta1, b2 = h.beta2, ob1 = 1.0f - h.beta1, ob2 = 1.0f - h.beta2;
  const int n4 = c.n >> 2;
  for (int i = threadIdx.x; i < n4; i += 256) {
    f32x4 p = ((f32x4*)c.p)[i], m = ((f32x4*)c.m)[i], v = ((f32x4*)c.v)[i];
    const f32x4 g = ((const f32x4*)c.g)[i] * coef;
    m = m * b1 + g * ob1;
    v = v * b2 + g * g * ob2;
#pragma unroll
    for (int e = 0; e < 4; e++) p[e] = p[e] * decay - step * m[e] / (sqrtf(v[e]) * rbc2 + h.eps);
    ((f32x4*)c.p)[i] = p;
    ((f32x4*)c.m)[i] = m;
    ((f32x4*)c.v)[i] = v;
  }
  for (int i = (n4 << 2) + threadIdx.x; i < c.n; i += 256) {
    const float g = c.g[i] * coef;
    const float m = c.m[i] * b1 + g * ob1, v = c.v[i] * b2 + g * g * ob2;
    c.p[i] = c.p[i] * decay - step * m / (sqrtf(v) * rbc2 + h.eps);
    c.m[i] = m;
    c.v[i] = v;
  }
}

}  // namespace smbv

using namespace smbv;

extern "C" {

int smbv_grad_sqnorm(const smbv_adam_chunk* chunks_dev, int nchunks, float* partial, float* out_sq, smbv_stream stream) {
  if (!chunks_dev || nchunks <= 0 || !partial || !out_sq) { set_error("smbv_grad_sqnorm: bad argument"); return SMBV_ERR_ARG; }
  hipLaunchKernelGGL(sqnorm_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, chunks_dev, partial);
  int rc = check_launch("sqnorm");
  if (rc) return rc;
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)partial, nchunks, out_sq);
  return check_launch("sum_partials");
}

int smbv_adamw_step(const smbv_adam_chunk* chunks_dev, int nchunks, const smbv_adam_hyper* h, const float* gnorm_sq, smbv_stream stream) {
  if (!chunks_dev || nchunks <= 0 || !h) { set_error("smbv_adamw_step: bad argument"); return SMBV_ERR_ARG; }
  if (h->bias_corr1 <= 0.f || h->bias_corr2 <= 0.f) { set_error("smbv_adamw_step: bias corrections must be positive"); return SMBV_ERR_ARG; }
  hipLaunchKernelGGL(adamw_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, chunks_dev, *h, gnorm_sq, (gnorm_sq && !(g_ab & 2)) ? 1 : 0);
  return check_launch("adamw");
}

}  // extern "C"
