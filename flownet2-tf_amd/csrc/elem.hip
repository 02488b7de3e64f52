// HBM-bound fused elementwise kernels between the sub-networks of the stacked models.
// One lane per pixel; the 16-channel (padded) input row of the next network is assembled in
// registers and written with 16-byte stores, so flow_warp + brightness error + concat
// (flownet_cs.py:21-36) and ChannelNorm x4 + flow_warp x2 + concat (flownet2.py:25-47) are one
// pass over the images instead of 6-10 TF nodes.
#include "fn2_common.h"

namespace fn2 {

// bilinear warp of a 3-channel fp32 image, exact reference rules (flow_warp.cu.cc:44-95)
__device__ __forceinline__ void warp3(const float* __restrict__ img, long nb, int x, int y, float u, float v,
                                      int W, int H, float o[3]) {
  const float x2 = (float)x + u, y2 = (float)y + v;
  o[0] = o[1] = o[2] = 0.f;
  if (!((x2 >= 0.f) && (y2 >= 0.f) && (x2 < (float)W) && (y2 < (float)H))) return;
  const int xL = (int)x2, yT = (int)y2;
  const int xR = min(xL + 1, W - 1), yB = min(yT + 1, H - 1);
  const float al = x2 - (float)xL, be = y2 - (float)yT;
  const float cTL = (1.f - al) * (1.f - be), cTR = al * (1.f - be), cBL = (1.f - al) * be, cBR = al * be;
  // 12-byte tap loads (fn2_common.h rgb3_t); same expression per channel as flow_warp_kernel
  const rgb3_t tl = load_rgb(img + (nb + (long)yT * W + xL) * 3), tr = load_rgb(img + (nb + (long)yT * W + xR) * 3);
  const rgb3_t bl = load_rgb(img + (nb + (long)yB * W + xL) * 3), br = load_rgb(img + (nb + (long)yB * W + xR) * 3);
  o[0] = cTL * tl.r + cTR * tr.r + cBL * bl.r + cBR * br.r;
  o[1] = cTL * tl.g + cTR * tr.g + cBL * bl.g + cBR * br.g;
  o[2] = cTL * tl.b + cTR * tr.b + cBL * bl.b + cBR * br.b;
}

template <typename OutT>
__global__ void __launch_bounds__(256) stack_input_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          const float* __restrict__ flow, OutT* __restrict__ out,
                                                          int N, int H, int W, int out_cs, int out_c0, int pad) {
  const long npix = (long)N * H * W;
  for (long pix = (long)blockIdx.x * blockDim.x + threadIdx.x; pix < npix; pix += (long)gridDim.x * blockDim.x) {
    const int x = (int)(pix % W), y = (int)((pix / W) % H);
    const long nb = (pix / W / H) * (long)H * W;
    const float2 f = *reinterpret_cast<const float2*>(flow + pix * 2);
    float wv[3];
    warp3(b, nb, x, y, f.x, f.y, W, H, wv);
    float v[16];
    float e = 0.f;
    const rgb3_t pa = load_rgb(a + pix * 3), pb = load_rgb(b + pix * 3);
    const float av3[3] = {pa.r, pa.g, pa.b}, bv3[3] = {pb.r, pb.g, pb.b};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float av = av3[c];
      v[c] = av;
      v[3 + c] = bv3[c];
      v[6 + c] = wv[c];
      const float d = av - wv[c];
      e += d * d;
    }
    v[9] = f.x * 0.05f;   // flownet_cs.py:34
    v[10] = f.y * 0.05f;
    v[11] = sqrtf(e);     // brightness error, flownet_cs.py:24-27
    v[12] = v[13] = v[14] = v[15] = 0.f;
    const size_t opix = ((size_t)(pix / W / H) * (H + 2 * pad) + y + pad) * (W + 2 * pad) + x + pad;
    store16<OutT>(out + opix * out_cs + out_c0, v);
  }
}

template <typename OutT>
__global__ void __launch_bounds__(256) fusion_input_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                           const float* __restrict__ fsd,
                                                           const float* __restrict__ fcss, OutT* __restrict__ out,
                                                           int N, int H, int W, int out_cs, int out_c0, int pad) {
  const long npix = (long)N * H * W;
  for (long pix = (long)blockIdx.x * blockDim.x + threadIdx.x; pix < npix; pix += (long)gridDim.x * blockDim.x) {
    const int x = (int)(pix % W), y = (int)((pix / W) % H);
    const long nb = (pix / W / H) * (long)H * W;
    const float2 sd = *reinterpret_cast<const float2*>(fsd + pix * 2);
    const float2 cs = *reinterpret_cast<const float2*>(fcss + pix * 2);
    float wsd[3], wcs[3];
    warp3(b, nb, x, y, sd.x, sd.y, W, H, wsd);  // flownet2.py:33
    warp3(b, nb, x, y, cs.x, cs.y, W, H, wcs);  // flownet2.py:37
    float v[16];
    float esd = 0.f, ecs = 0.f;
    const rgb3_t pa = load_rgb(a + pix * 3);
    const float av3[3] = {pa.r, pa.g, pa.b};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float av = av3[c];
      v[c] = av;
      const float d1 = av - wsd[c], d2 = av - wcs[c];
      esd += d1 * d1;
      ecs += d2 * d2;
    }
    v[3] = sd.x; v[4] = sd.y; v[5] = cs.x; v[6] = cs.y;           // flownet2.py:41-43
    v[7] = sqrtf(sd.x * sd.x + sd.y * sd.y);                       // ChannelNorm(flow_sd), :30
    v[8] = sqrtf(cs.x * cs.x + cs.y * cs.y);                       // ChannelNorm(flow_css), :31
    v[9] = sqrtf(esd);                                             // :35
    v[10] = sqrtf(ecs);                                            // :39
    v[11] = v[12] = v[13] = v[14] = v[15] = 0.f;
    const size_t opix = ((size_t)(pix / W / H) * (H + 2 * pad) + y + pad) * (W + 2 * pad) + x + pad;
    store16<OutT>(out + opix * out_cs + out_c0, v);
  }
}

static inline int grid_for(long work_items, int block) {
  long g = (work_items + block - 1) / block;
  if (g > 256L * 16) g = 256L * 16;
  if (g < 1) g = 1;
  return (int)g;
}

static int check_out16(const fn2_tensor* out, int c, const char* what) {
  FN2_REQUIRE(out && out->data, "%s: null output", what);
  FN2_REQUIRE(out->c == c, "%s: output view must have %d channels", what, c);
  FN2_REQUIRE(out->cs % 8 == 0 && out->c0 % 8 == 0 && out->c0 + 16 <= out->cs,
              "%s: output needs a 16-channel, 8-aligned slot", what);
  FN2_REQUIRE(out->dtype >= FN2_F32 && out->dtype <= FN2_F16X2, "%s: bad dtype", what);
  return FN2_OK;
}

}  // namespace fn2

using namespace fn2;

// uint8 image bytes -> fp32 through a 256-entry table: what Net.adapt_x does on the host (src/net.py:338-345:
// `x / 255.0` in float64 when the image's max exceeds 1, else the values as they are, then the float32 feed) done
// after the copy, so the host link carries one byte per channel instead of four.  The table IS the host arithmetic
// (lut[i] = float32(float64(i) / 255.0) or float32(i)), hence the result is byte-identical to the fp32 path.
// HBM-bound: 16 bytes in, 64 bytes out per lane.
__global__ void __launch_bounds__(256) u8_to_f32_lut_kernel(const unsigned char* __restrict__ src,
                                                            const float* __restrict__ lut, float* __restrict__ dst,
                                                            long count) {
  __shared__ float t[256];
  t[threadIdx.x] = lut[threadIdx.x];
  __syncthreads();
  const long n16 = count >> 4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x) {
    const uint4 v = reinterpret_cast<const uint4*>(src)[i];
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
    float4* o = reinterpret_cast<float4*>(dst) + i * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      o[j] = make_float4(t[w[j] & 255u], t[(w[j] >> 8) & 255u], t[(w[j] >> 16) & 255u], t[w[j] >> 24]);
  }
  if (blockIdx.x == 0)  // ragged tail (counts that are not multiples of 16)
    for (long i = (n16 << 4) + threadIdx.x; i < count; i += blockDim.x) dst[i] = t[src[i]];
}

extern "C" {

int fn2_u8_to_f32_lut(const unsigned char* src, const float* lut256, float* dst, long count, void* stream) {
  FN2_REQUIRE(src && lut256 && dst && count >= 0, "u8_to_f32_lut: null pointer / negative count");
  FN2_REQUIRE(((size_t)src & 15) == 0 && ((size_t)dst & 15) == 0, "u8_to_f32_lut: src and dst must be 16-byte aligned");
  if (count == 0) return FN2_OK;
  hipLaunchKernelGGL(u8_to_f32_lut_kernel, dim3(grid_for(count >> 4, 256)), dim3(256), 0, (hipStream_t)stream, src,
                     lut256, dst, count);
  FN2_CHECK_LAUNCH("u8_to_f32_lut");
  return FN2_OK;
}

int fn2_stack_input(const float* a, const float* b, const float* flow, const fn2_tensor* out, int pad, void* stream) {
  FN2_REQUIRE(a && b && flow, "stack_input: null pointer");
  int rc = check_out16(out, 12, "stack_input");
  if (rc) return rc;
  FN2_REQUIRE(pad >= 0 && out->h > 2 * pad && out->w > 2 * pad, "bad border");
  const long npix = (long)out->n * (out->h - 2 * pad) * (out->w - 2 * pad);
  if (out->dtype == FN2_F32)
    hipLaunchKernelGGL(stack_input_kernel<float>, dim3(grid_for(npix, 256)), dim3(256), 0, (hipStream_t)stream, a,
                       b, flow, (float*)out->data, out->n, out->h - 2 * pad, out->w - 2 * pad, out->cs, out->c0, pad);
  else if (out->dtype == FN2_F16X2)
    hipLaunchKernelGGL(stack_input_kernel<x2_t>, dim3(grid_for(npix, 256)), dim3(256), 0, (hipStream_t)stream, a,
                       b, flow, (x2_t*)out->data, out->n, out->h - 2 * pad, out->w - 2 * pad, out->cs, out->c0, pad);
  else if (out->dtype == FN2_BF16)
    hipLaunchKernelGGL(stack_input_kernel<bf16_t>, dim3(grid_for(npix, 256)), dim3(256), 0, (hipStream_t)stream, a,
                       b, flow, (bf16_t*)out->data, out->n, out->h - 2 * pad, out->w - 2 * pad, out->cs, out->c0, pad);
  else
    hipLaunchKernelGGL(stack_input_kernel<f16_t>, dim3(grid_for(npix, 256)), dim3(256), 0, (hipStream_t)stream, a,
                       b, flow, (f16_t*)out->data, out->n, out->h - 2 * pad, out->w - 2 * pad, out->cs, out->c0, pad);
  FN2_CHECK_LAUNCH("stack_input");
  return FN2_OK;
}

int fn2_fusion_input(const float* a, const float* b, const float* flow_sd, const float* flow_css,
                     const fn2_tensor* out, int pad, void* stream) {
  FN2_REQUIRE(a && b && flow_sd && flow_css, "fusion_input: null pointer");
  int rc = check_out16(out, 11, "fusion_input");
  if (rc) return rc;
  FN2_REQUIRE(pad >= 0 && out->h > 2 * pad && out->w > 2 * pad, "bad border");
  const long npix = (long)out->n * (out->h - 2 * pad) * (out->w - 2 * pad);
  if (out->dtype == FN2_F32)
    hipLaunchKernelGGL(fusion_input_kernel<float>, dim3(grid_for(npix, 256)), dim3(256), 0, (hipStream_t)stream, a,
                       b, flow_sd, flow_css, (float*)out->data, out->n, out->h - 2 * pad, out->w - 2 * pad, out->cs, out->c0, pad);
  else if (out->dtype == FN2_F16X2)
    hipLaunchKernelGGL(fusion_input_kernel<x2_t>, dim3(grid_for(npix, 256)), dim3(256), 0, (hipStream_t)stream,
                       a, b, flow_sd, flow_css, (x2_t*)out->data, out->n, out->h - 2 * pad, out->w - 2 * pad, out->cs, out->c0, pad);
  else if (out->dtype == FN2_BF16)
    hipLaunchKernelGGL(fusion_input_kernel<bf16_t>, dim3(grid_for(npix, 256)), dim3(256), 0, (hipStream_t)stream,
                       a, b, flow_sd, flow_css, (bf16_t*)out->data, out->n, out->h - 2 * pad, out->w - 2 * pad, out->cs, out->c0, pad);
  else
    hipLaunchKernelGGL(fusion_input_kernel<f16_t>, dim3(grid_for(npix, 256)), dim3(256), 0, (hipStream_t)stream,
                       a, b, flow_sd, flow_css, (f16_t*)out->data, out->n, out->h - 2 * pad, out->w - 2 * pad, out->cs, out->c0, pad);
  FN2_CHECK_LAUNCH("fusion_input");
  return FN2_OK;
}

}  // extern "C"
