// Training-input augmentation ops of the reference's preprocessing plugin (src/ops/preprocessing):
//   DataAugmentation  -> fn2_augment_f32          (kernels/data_augmentation.cc:30-150, .cu.cc:22-70)
//   FlowAugmentation  -> fn2_flow_augmentation_f32 (kernels/flow_augmentation.cc:19-66, _gpu.cu.cc:22-70)
// Both are one pass over the output: HBM-bound, a lane per output pixel (all channels of the pixel, so the
// chromatic transform sees r, g and b together).  The random coefficients and their composition into the
// 2x3 matrices are host work (src/preprocessing.py), as in the reference (the op keeps them in host memory).
#include "fn2_common.h"

namespace fn2 {

static inline int aug_grid(long work_items) {
  long g = (work_items + 255) / 256;
  if (g > 256L * 32) g = 256L * 32;
  return (int)(g < 1 ? 1 : g);
}

__device__ __forceinline__ float clampf(float f, float a, float b) { return fmaxf(a, fminf(f, b)); }

// out[n,y,x,:] = bilinear(src[n], T_n (x, y)) with the sample position clamped to [0, size - 1.05]
// (data_augmentation.cc:74-115), then -- when chroma != nullptr, C == 3 -- the chromatic chain per pixel
// (:117-147): colour gains, brightness compensation mean_in / (mean_out + 0.01), clamp, gamma, brightness,
// contrast around 0.5, clamp.  chroma[n] = (gamma, brightness, contrast, color1, color2, color3).
template <int C>
__global__ void __launch_bounds__(256) augment_kernel(const float* __restrict__ src, float* __restrict__ out,
                                                      const float* __restrict__ trans, const float* __restrict__ chroma,
                                                      int N, int SH, int SW, int OH, int OW, int Cdyn) {
  const int Cn = C > 0 ? C : Cdyn;
  const long total = (long)N * OH * OW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % OW), y = (int)((i / OW) % OH), n = (int)(i / OW / OH);
    const float* t = trans + n * 6;
    float xpos = x * t[0] + y * t[1] + t[2];
    float ypos = x * t[3] + y * t[4] + t[5];
    xpos = clampf(xpos, 0.0f, (float)SW - 1.05f);
    ypos = clampf(ypos, 0.0f, (float)SH - 1.05f);
    const float tlx = floorf(xpos), tly = floorf(ypos);
    const float xd = xpos - tlx, yd = ypos - tly;
    const float* tl = src + (((long)n * SH + (int)tly) * SW + (int)tlx) * Cn;
    const float* tr = tl + Cn;
    const float* bl = tl + (long)Cn * SW;
    const float* br = bl + Cn;
    float* o = out + i * Cn;
    if constexpr (C == 3) {
      if (chroma != nullptr) {
        const float* cc = chroma + n * 6;
        float rgb[3], mean_in = 0.f, mean_out = 0.f;
        // 12-byte tap loads / result store (fn2_common.h rgb3_t): a third of the memory instructions
        const rgb3_t ptl = load_rgb(tl), ptr_ = load_rgb(tr), pbl = load_rgb(bl), pbr = load_rgb(br);
        const float vtl[3] = {ptl.r, ptl.g, ptl.b}, vtr[3] = {ptr_.r, ptr_.g, ptr_.b};
        const float vbl[3] = {pbl.r, pbl.g, pbl.b}, vbr[3] = {pbr.r, pbr.g, pbr.b};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float v = (1 - xd) * (1 - yd) * vtl[c] + xd * yd * vbr[c] + (1 - xd) * yd * vbl[c] + xd * (1 - yd) * vtr[c];
          mean_in += v;
          rgb[c] = v * cc[3 + c];
          mean_out += rgb[c];
        }
        const float comp = mean_in / (mean_out + 0.01f);
        float res[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          float v = clampf(rgb[c] * comp, 0.0f, 1.0f);
          v = powf(v, cc[0]);
          v = v + cc[1];
          v = 0.5f + (v - 0.5f) * cc[2];
          res[c] = clampf(v, 0.0f, 1.0f);
        }
        store_rgb(o, res[0], res[1], res[2]);
        continue;
      }
    }
    for (int c = 0; c < Cn; ++c)
      o[c] = (1 - xd) * (1 - yd) * tl[c] + xd * yd * br[c] + (1 - xd) * yd * bl[c] + xd * (1 - yd) * tr[c];
  }
}

// flow_augmentation.cc:30-66: out[n,y,x] = T_b^-1 (p1 + flow[n, round(p1)]) - (x, y), p1 = T_a (x, y); the
// flow is read at (int)(p + 0.5) through a FLAT index clamped to the tensor (the reference's clamp, :47-52).
__global__ void __launch_bounds__(256) flow_augmentation_kernel(const float* __restrict__ flow, const float* __restrict__ ta,
                                                                const float* __restrict__ itb, float* __restrict__ out,
                                                                int N, int SH, int SW, int OH, int OW) {
  const long total = (long)N * OH * OW;
  const long src_total = (long)N * SH * SW * 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const float x = (float)(i % OW), y = (float)((i / OW) % OH);
    const int n = (int)(i / OW / OH);
    const float* a = ta + n * 6;
    const float* b = itb + n * 6;
    const float x1 = x * a[0] + y * a[1] + a[2];
    const float y1 = x * a[3] + y * a[4] + a[5];
    const long ix = (((long)n * SH + (int)(y1 + 0.5f)) * SW + (int)(x1 + 0.5f)) * 2;
    const long cx = ix < 0 ? 0 : (ix > src_total - 1 ? src_total - 1 : ix);
    const long iy = ix + 1;
    const long cy = iy < 0 ? 0 : (iy > src_total - 1 ? src_total - 1 : iy);
    const float x2 = x1 + flow[cx], y2 = y1 + flow[cy];
    const float x3 = x2 * b[0] + y2 * b[1] + b[2];
    const float y3 = x2 * b[3] + y2 * b[4] + b[5];
    *reinterpret_cast<float2*>(out + i * 2) = make_float2(x3 - x, y3 - y);
  }
}

}  // namespace fn2

using namespace fn2;

extern "C" {

int fn2_augment_f32(const float* src, const float* transforms, const float* chromatic, float* out, int n, int src_h,
                    int src_w, int c, int out_h, int out_w, void* stream) {
  FN2_REQUIRE(src && transforms && out, "augment: null pointer");
  FN2_REQUIRE(n >= 1 && src_h >= 2 && src_w >= 2 && c >= 1 && out_h >= 1 && out_w >= 1, "augment: bad dims");
  FN2_REQUIRE(chromatic == nullptr || c == 3, "augment: the chromatic transform needs 3 channels");
  const long total = (long)n * out_h * out_w;
  hipStream_t s = (hipStream_t)stream;
  if (c == 3)
    hipLaunchKernelGGL(augment_kernel<3>, dim3(aug_grid(total)), dim3(256), 0, s, src, out, transforms, chromatic, n, src_h,
                       src_w, out_h, out_w, 3);
  else
    hipLaunchKernelGGL(augment_kernel<0>, dim3(aug_grid(total)), dim3(256), 0, s, src, out, transforms, chromatic, n, src_h,
                       src_w, out_h, out_w, c);
  FN2_CHECK_LAUNCH("augment");
  return FN2_OK;
}

int fn2_flow_augmentation_f32(const float* flows, const float* transforms_from_a, const float* inv_transforms_from_b,
                              float* out, int n, int src_h, int src_w, int out_h, int out_w, void* stream) {
  FN2_REQUIRE(flows && transforms_from_a && inv_transforms_from_b && out, "flow_augmentation: null pointer");
  FN2_REQUIRE(n >= 1 && src_h >= 1 && src_w >= 1 && out_h >= 1 && out_w >= 1, "flow_augmentation: bad dims");
  hipLaunchKernelGGL(flow_augmentation_kernel, dim3(aug_grid((long)n * out_h * out_w)), dim3(256), 0, (hipStream_t)stream,
                     flows, transforms_from_a, inv_transforms_from_b, out, n, src_h, src_w, out_h, out_w);
  FN2_CHECK_LAUNCH("flow_augmentation");
  return FN2_OK;
}

}  // extern "C"
