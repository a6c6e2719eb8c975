// Device side of the input feed: the tail of the reference's MONAI pipeline that runs on every volume before
// the model sees it - ScaleIntensityRanged(a_min, a_max, b_min, b_max, clip) -> SpatialPadd(H, W, D) -> CenterSpatialCropd(H, W, D)
// -> PermuteImage (src/dataloader/mim.py:154-169, 88-93) - as ONE pass from the uploaded raw volume (int16 / uint16 HU, bf16 or fp32,
// memory order (H0, W0, D0) with D0 fastest, as EnsureChannelFirstd leaves a NIfTI array) to the model's fp32 [D, 1, H, W] layout.
// Uploading int16 instead of the 335.5 MB fp32 volume cuts the PCIe bytes per volume 4x; nothing of the pipeline's arithmetic changes:
//   v = (x - a_min) / (a_max - a_min) * (b_max - b_min) + b_min   (fp32, IEEE division), clipped to [b_min, b_max] when clip != 0;
//   voxels the symmetric pad adds are 0 (the pad follows the scaling); the centre crop starts at size/2 - roi/2.
// The H0/W0/D0 -> D/H/W permutation is a transpose of the (w, d) plane for every h: 64 x 64 tiles go through LDS so that both the
// reads (128 B of int16 along d) and the writes (256 B of fp32 along w) are whole cache lines.
#include "smbv_common.h"
#include "smbv_kernels.h"

namespace smbv {

template <typename T>
__device__ __forceinline__ float raw_to_float(T v);
template <> __device__ __forceinline__ float raw_to_float<short>(short v) { return (float)v; }
template <> __device__ __forceinline__ float raw_to_float<unsigned short>(unsigned short v) { return (float)v; }
template <> __device__ __forceinline__ float raw_to_float<float>(float v) { return v; }
struct bf16_raw { unsigned short bits; };
template <> __device__ __forceinline__ float raw_to_float<bf16_raw>(bf16_raw v) { return __builtin_bit_cast(float, (unsigned)v.bits << 16); }

struct PrepGeom {
  int H0, W0, D0;     // source extents
  int H, W, D;        // target extents (model: image_size, image_size, num_frames)
  int oh, ow, od;     // source index = target index + offset (negative inside the left pad)
  float a_min, b_min, b_max, a_rng;
  int clip;
};

template <typename T>
__global__ __launch_bounds__(256) void prepare_volume_kernel(const T* __restrict__ src, float* __restrict__ dst, PrepGeom g) {
  __shared__ float tile[64][65];
  const int t = threadIdx.x;
  const int w0 = blockIdx.x * 64, d0 = blockIdx.y * 64, h = blockIdx.z % g.H, b = blockIdx.z / g.H;
  const int hs = h + g.oh;
  const T* sp = src + (size_t)b * g.H0 * g.W0 * g.D0;
  // read: 16 threads per w-row (4 consecutive d each), 16 rows per pass
  const int rd = (t & 15) * 4, rw = t >> 4;
#pragma unroll
  for (int pass = 0; pass < 4; pass++) {
    const int wl = rw + 16 * pass;
    const int ws = w0 + wl + g.ow;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const int ds = d0 + rd + e + g.od;
      float v = 0.f;  // SpatialPadd: constant 0 after the intensity scaling
      if (hs >= 0 && hs < g.H0 && ws >= 0 && ws < g.W0 && ds >= 0 && ds < g.D0) {
        const float x = raw_to_float<T>(sp[((size_t)hs * g.W0 + ws) * g.D0 + ds]);
        // each step rounded on its own, as the reference's tensor ops are.  hipcc's __fmul_rn / __fadd_rn are plain * and +, which -ffp-contract=fast
        // fuses into one FMA (invisible while b_max - b_min is a power of two, one ulp off at 3.5): the product is pinned in a register between them
        v = __fdiv_rn(__fsub_rn(x, g.a_min), g.a_rng);
        v = __fmul_rn(v, __fsub_rn(g.b_max, g.b_min));
        asm volatile("" : "+v"(v));
        v = __fadd_rn(v, g.b_min);
        if (g.clip) v = fminf(fmaxf(v, g.b_min), g.b_max);
      }
      tile[wl][rd + e] = v;
    }
  }
  __syncthreads();
  // write: 16 threads per d-row (4 consecutive w each), 16 rows per pass
  const int ww = (t & 15) * 4, wd = t >> 4;
  float* dp = dst + (size_t)b * g.D * g.H * g.W;
#pragma unroll
  for (int pass = 0; pass < 4; pass++) {
    const int dl = wd + 16 * pass;
    const int d = d0 + dl;
    if (d >= g.D) continue;
    float* o = dp + ((size_t)d * g.H + h) * g.W + w0 + ww;
    if (w0 + ww + 3 < g.W && ((g.W & 3) == 0)) {
      f32x4 v = {tile[ww][dl], tile[ww + 1][dl], tile[ww + 2][dl], tile[ww + 3][dl]};
      *(f32x4*)o = v;
    } else {
#pragma unroll
      for (int e = 0; e < 4; e++)
        if (w0 + ww + e < g.W) o[e] = tile[ww + e][dl];
    }
  }
}

// MONAI semantics (published behaviour; monai itself is absent from the build image, see oracle/feed_oracle.py):
//   SpatialPad(method="symmetric"): pad = max(target - size, 0), left = pad / 2, right = pad - left;
//   CenterSpatialCrop(roi): start = size' / 2 - roi / 2 on the padded size'.
static void axis_offset(int size, int target, int& off) {
  const int pad = target > size ? target - size : 0;
  const int left = pad / 2;
  const int padded = size + pad;
  const int start = padded / 2 - target / 2;  // 0 whenever padding happened
  off = start - left;                         // source index = target index + off
}

int prepare_volume(const void* src, int src_dtype, float* dst, int B, int H0, int W0, int D0, int H, int W, int D, float a_min, float a_max,
                   float b_min, float b_max, int clip, hipStream_t st) {
  if (B <= 0 || H0 <= 0 || W0 <= 0 || D0 <= 0 || H <= 0 || W <= 0 || D <= 0) { set_error("prepare_volume: empty volume"); return SMBV_ERR_SHAPE; }
  if (!(a_max > a_min)) { set_error("prepare_volume: need a_max > a_min"); return SMBV_ERR_ARG; }
  if ((size_t)B * H > 65535u * 1ull) { set_error("prepare_volume: B*H too large for one launch"); return SMBV_ERR_SHAPE; }
  if ((uintptr_t)dst & 15) { set_error("prepare_volume: dst must be 16-byte aligned"); return SMBV_ERR_ALIGN; }
  PrepGeom g;
  g.H0 = H0; g.W0 = W0; g.D0 = D0; g.H = H; g.W = W; g.D = D;
  axis_offset(H0, H, g.oh);
  axis_offset(W0, W, g.ow);
  axis_offset(D0, D, g.od);
  g.a_min = a_min; g.a_rng = a_max - a_min; g.b_min = b_min; g.b_max = b_max; g.clip = clip;
  const dim3 grid(cdiv(W, 64), cdiv(D, 64), B * H), blk(256);
  switch (src_dtype) {
    case 0: hipLaunchKernelGGL(prepare_volume_kernel<short>, grid, blk, 0, st, (const short*)src, dst, g); break;
    case 1: hipLaunchKernelGGL(prepare_volume_kernel<unsigned short>, grid, blk, 0, st, (const unsigned short*)src, dst, g); break;
    case 2: hipLaunchKernelGGL(prepare_volume_kernel<bf16_raw>, grid, blk, 0, st, (const bf16_raw*)src, dst, g); break;
    case 3: hipLaunchKernelGGL(prepare_volume_kernel<float>, grid, blk, 0, st, (const float*)src, dst, g); break;
    default: set_error("prepare_volume: unknown source dtype %d (0 int16, 1 uint16, 2 bf16, 3 fp32)", src_dtype); return SMBV_ERR_ARG;
  }
  return check_launch("prepare_volume");
}

}  // namespace smbv
