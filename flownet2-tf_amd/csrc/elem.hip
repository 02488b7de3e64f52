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

// A full-resolution flow field given either as itself (ph == 0: `p` is [N,H,W,2]) or as the quarter-resolution
// predict_flow2 it is resized from (`flow = resize_bilinear(scale * predict_flow2)`, align_corners, flownet_s.py:105-109):
// then the consumer interpolates its pixel itself -- resize_bilinear_c2_kernel's arithmetic, operation for operation --
// and, when `keep` is set, stores it there as the resize launch would have.
struct FlowSrc { const float* p; int ph, pw; float sy, sx, scale; float* keep; };

__device__ __forceinline__ float2 flow_at(const FlowSrc& f, int n, int y, int x, long pix) {
  if (f.ph == 0) return *reinterpret_cast<const float2*>(f.p + pix * 2);
  const float fy = (float)y * f.sy;
  const int y0 = (int)floorf(fy);
  const int y1 = min(y0 + 1, f.ph - 1);
  const float ly = fy - (float)y0;
  const float2* r0 = reinterpret_cast<const float2*>(f.p) + ((long)n * f.ph + y0) * f.pw;
  const float2* r1 = reinterpret_cast<const float2*>(f.p) + ((long)n * f.ph + y1) * f.pw;
  const float fx = (float)x * f.sx;
  const int x0 = (int)floorf(fx);
  const int x1 = min(x0 + 1, f.pw - 1);
  const float lx = fx - (float)x0;
  float2 r = bilerp_c2(r0[x0], r0[x1], r1[x0], r1[x1], lx, ly, f.scale);
  // the vector is used as the resize launch would have STORED it: without this the final multiply by `scale` contracts
  // with the consumer's first addition (x + u in the warp) and the results differ in the last bit
  asm volatile("" : "+v"(r.x), "+v"(r.y));
  if (f.keep != nullptr) *reinterpret_cast<float2*>(f.keep + pix * 2) = r;
  return r;
}

template <typename OutT>
__global__ void __launch_bounds__(256) stack_input_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          const FlowSrc flow, OutT* __restrict__ out,
                                                          int N, int H, int W, int out_cs, int out_c0, int pad) {
  const long npix = (long)N * H * W;
  for (long pix = (long)blockIdx.x * blockDim.x + threadIdx.x; pix < npix; pix += (long)gridDim.x * blockDim.x) {
    const int x = (int)(pix % W), y = (int)((pix / W) % H);
    const long nb = (pix / W / H) * (long)H * W;
    const float2 f = flow_at(flow, (int)(pix / W / H), y, x, pix);
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
                                                           const FlowSrc fsd, const FlowSrc fcss, OutT* __restrict__ out,
                                                           int N, int H, int W, int out_cs, int out_c0, int pad) {
  const long npix = (long)N * H * W;
  for (long pix = (long)blockIdx.x * blockDim.x + threadIdx.x; pix < npix; pix += (long)gridDim.x * blockDim.x) {
    const int x = (int)(pix % W), y = (int)((pix / W) % H);
    const long nb = (pix / W / H) * (long)H * W;
    const float2 sd = flow_at(fsd, (int)(pix / W / H), y, x, pix);
    const float2 cs = flow_at(fcss, (int)(pix / W / H), y, x, pix);
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

static FlowSrc flow_src(const float* p, int ph, int pw, float scale, float* keep, int out_h, int out_w) {
  FlowSrc f{p, ph, pw, 0.f, 0.f, scale, keep};
  if (ph > 0) {   // fn2_resize_bilinear_f32's scales
    f.sy = out_h > 1 ? (float)(ph - 1) / (float)(out_h - 1) : 0.f;
    f.sx = out_w > 1 ? (float)(pw - 1) / (float)(out_w - 1) : 0.f;
  }
  return f;
}

static int stack_input_launch(const float* a, const float* b, const FlowSrc& f, const fn2_tensor* out, int pad, void* stream) {
  const long npix = (long)out->n * (out->h - 2 * pad) * (out->w - 2 * pad);
  if (out->dtype == FN2_F32)
    hipLaunchKernelGGL(stack_input_kernel<float>, dim3(grid_for(npix, 256)), dim3(256), 0, (hipStream_t)stream, a,
                       b, f, (float*)out->data, out->n, out->h - 2 * pad, out->w - 2 * pad, out->cs, out->c0, pad);
  else if (out->dtype == FN2_F16X2)
    hipLaunchKernelGGL(stack_input_kernel<x2_t>, dim3(grid_for(npix, 256)), dim3(256), 0, (hipStream_t)stream, a,
                       b, f, (x2_t*)out->data, out->n, out->h - 2 * pad, out->w - 2 * pad, out->cs, out->c0, pad);
  else if (out->dtype == FN2_BF16)
    hipLaunchKernelGGL(stack_input_kernel<bf16_t>, dim3(grid_for(npix, 256)), dim3(256), 0, (hipStream_t)stream, a,
                       b, f, (bf16_t*)out->data, out->n, out->h - 2 * pad, out->w - 2 * pad, out->cs, out->c0, pad);
  else
    hipLaunchKernelGGL(stack_input_kernel<f16_t>, dim3(grid_for(npix, 256)), dim3(256), 0, (hipStream_t)stream, a,
                       b, f, (f16_t*)out->data, out->n, out->h - 2 * pad, out->w - 2 * pad, out->cs, out->c0, pad);
  FN2_CHECK_LAUNCH("stack_input");
  return FN2_OK;
}

int fn2_stack_input(const float* a, const float* b, const float* flow, const fn2_tensor* out, int pad, void* stream) {
  FN2_REQUIRE(a && b && flow, "stack_input: null pointer");
  int rc = check_out16(out, 12, "stack_input");
  if (rc) return rc;
  FN2_REQUIRE(pad >= 0 && out->h > 2 * pad && out->w > 2 * pad, "bad border");
  return stack_input_launch(a, b, flow_src(flow, 0, 0, 1.f, nullptr, 0, 0), out, pad, stream);
}

int fn2_stack_input_pf(const float* a, const float* b, const float* pf, int pf_h, int pf_w, float scale, float* flow_out,
                       const fn2_tensor* out, int pad, void* stream) {
  FN2_REQUIRE(a && b && pf && pf_h >= 1 && pf_w >= 1, "stack_input_pf: null pointer / bad size");
  int rc = check_out16(out, 12, "stack_input_pf");
  if (rc) return rc;
  FN2_REQUIRE(pad >= 0 && out->h > 2 * pad && out->w > 2 * pad, "bad border");
  return stack_input_launch(a, b, flow_src(pf, pf_h, pf_w, scale, flow_out, out->h - 2 * pad, out->w - 2 * pad), out, pad, stream);
}

static int fusion_input_launch(const float* a, const float* b, const FlowSrc& fs, const FlowSrc& fc, const fn2_tensor* out,
                               int pad, void* stream) {
  const long npix = (long)out->n * (out->h - 2 * pad) * (out->w - 2 * pad);
  if (out->dtype == FN2_F32)
    hipLaunchKernelGGL(fusion_input_kernel<float>, dim3(grid_for(npix, 256)), dim3(256), 0, (hipStream_t)stream, a,
                       b, fs, fc, (float*)out->data, out->n, out->h - 2 * pad, out->w - 2 * pad, out->cs, out->c0, pad);
  else if (out->dtype == FN2_F16X2)
    hipLaunchKernelGGL(fusion_input_kernel<x2_t>, dim3(grid_for(npix, 256)), dim3(256), 0, (hipStream_t)stream,
                       a, b, fs, fc, (x2_t*)out->data, out->n, out->h - 2 * pad, out->w - 2 * pad, out->cs, out->c0, pad);
  else if (out->dtype == FN2_BF16)
    hipLaunchKernelGGL(fusion_input_kernel<bf16_t>, dim3(grid_for(npix, 256)), dim3(256), 0, (hipStream_t)stream,
                       a, b, fs, fc, (bf16_t*)out->data, out->n, out->h - 2 * pad, out->w - 2 * pad, out->cs, out->c0, pad);
  else
    hipLaunchKernelGGL(fusion_input_kernel<f16_t>, dim3(grid_for(npix, 256)), dim3(256), 0, (hipStream_t)stream,
                       a, b, fs, fc, (f16_t*)out->data, out->n, out->h - 2 * pad, out->w - 2 * pad, out->cs, out->c0, pad);
  FN2_CHECK_LAUNCH("fusion_input");
  return FN2_OK;
}

int fn2_fusion_input(const float* a, const float* b, const float* flow_sd, const float* flow_css,
                     const fn2_tensor* out, int pad, void* stream) {
  FN2_REQUIRE(a && b && flow_sd && flow_css, "fusion_input: null pointer");
  int rc = check_out16(out, 11, "fusion_input");
  if (rc) return rc;
  FN2_REQUIRE(pad >= 0 && out->h > 2 * pad && out->w > 2 * pad, "bad border");
  return fusion_input_launch(a, b, flow_src(flow_sd, 0, 0, 1.f, nullptr, 0, 0), flow_src(flow_css, 0, 0, 1.f, nullptr, 0, 0),
                             out, pad, stream);
}

int fn2_fusion_input_pf(const float* a, const float* b, const float* pf_sd, float scale_sd, float* flow_sd_out,
                        const float* pf_css, float scale_css, float* flow_css_out, int pf_h, int pf_w,
                        const fn2_tensor* out, int pad, void* stream) {
  FN2_REQUIRE(a && b && pf_sd && pf_css && pf_h >= 1 && pf_w >= 1, "fusion_input_pf: null pointer / bad size");
  int rc = check_out16(out, 11, "fusion_input_pf");
  if (rc) return rc;
  FN2_REQUIRE(pad >= 0 && out->h > 2 * pad && out->w > 2 * pad, "bad border");
  const int oh = out->h - 2 * pad, ow = out->w - 2 * pad;
  return fusion_input_launch(a, b, flow_src(pf_sd, pf_h, pf_w, scale_sd, flow_sd_out, oh, ow),
                             flow_src(pf_css, pf_h, pf_w, scale_css, flow_css_out, oh, ow), out, pad, stream);
}

}  // extern "C"
