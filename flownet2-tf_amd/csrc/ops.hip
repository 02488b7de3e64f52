// Op-surface kernels (float32, dense NHWC) and small HBM-bound engine kernels.
// gfx950 only.  Semantics follow the reference kernels cited per function; the
// implementation does not (no materialised padding, no 1024-thread transposes,
// no default-stream memsets).
#include "fn2_common.h"

namespace fn2 {

static thread_local char g_err[512] = "";
char* err_buf() { return g_err; }
int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

// ---------------------------------------------------------------------------
// correlation, generic attributes (any odd k, s1, s2, pad): one thread per
// output element.  Fallback only -- the model's call site (k=1, s1=1, pad=md)
// goes through the MFMA kernel in corr.hip.
//   out[n,y,x,d] = 1/(k*k*C) sum_{j,i,c} A0[n,y1+j,x1+i,c]*B0[n,y1+s2p+j,x1+s2o+i,c]
//   (correlation_kernel.cu.cc:45-110), A0/B0 = inputs zero-extended by `pad`.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) correlation_generic_kernel(
    const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int N, int H,
    int W, int C, int k, int md, int s1, int s2, int pad, int oh, int ow, int gr, int gw) {
  const int D = gw * gw;
  const long total = (long)N * oh * ow * D;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    const int d = (int)(idx % D);
    const int x = (int)((idx / D) % ow);
    const int y = (int)((idx / D / ow) % oh);
    const int n = (int)(idx / D / ow / oh);
    const int s2o = (d % gw - gr) * s2;
    const int s2p = (d / gw - gr) * s2;
    // top-left of the patch in UNPADDED coordinates
    const int ya = y * s1 + md - pad, xa = x * s1 + md - pad;
    float acc = 0.f;
    for (int j = 0; j < k; ++j) {
      const int y1 = ya + j, y2 = ya + j + s2p;
      if (y1 < 0 || y1 >= H || y2 < 0 || y2 >= H) continue;
      for (int i = 0; i < k; ++i) {
        const int x1 = xa + i, x2 = xa + i + s2o;
        if (x1 < 0 || x1 >= W || x2 < 0 || x2 >= W) continue;
        const float* pa = a + (((long)n * H + y1) * W + x1) * C;
        const float* pb = b + (((long)n * H + y2) * W + x2) * C;
        int c = 0;
        if ((C & 3) == 0) {
          for (; c < C; c += 4) {
            const float4 va = *reinterpret_cast<const float4*>(pa + c);
            const float4 vb = *reinterpret_cast<const float4*>(pb + c);
            acc += va.x * vb.x + va.y * vb.y + va.z * vb.z + va.w * vb.w;
          }
        } else {
          for (; c < C; ++c) acc += pa[c] * pb[c];
        }
      }
    }
    out[idx] = acc / (float)(k * k * C);
  }
}

// ---------------------------------------------------------------------------
// correlation backward (correlation_grad_kernel.cu.cc:20-189): one thread per
// input element, both gradients in one pass over the displacement grid.
// ---------------------------------------------------------------------------
__device__ __forceinline__ int ceil_div_any(int num, int den) {  // ROUND_OFF trick, :5,:51-63
  return (num >= 0) ? (num + den - 1) / den : -((-num) / den);
}
__device__ __forceinline__ int floor_div_any(int num, int den) {
  return (num >= 0) ? num / den : -((-num + den - 1) / den);
}

__global__ void __launch_bounds__(256) correlation_grad_kernel(
    const float* __restrict__ g, const float* __restrict__ a, const float* __restrict__ b,
    float* __restrict__ da, float* __restrict__ db, int N, int H, int W, int C, int k, int md, int s1,
    int s2, int pad, int oh, int ow, int gr, int gw) {
  const int kr = (k - 1) / 2;
  const int D = gw * gw;
  const long total = (long)N * H * W * C;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % C);
    const int xx = (int)((idx / C) % W);
    const int yy = (int)((idx / C / W) % H);
    const int n = (int)(idx / C / W / H);
    const int x = xx + pad, y = yy + pad;  // padded coordinates, :44-45
    float sa = 0.f, sb = 0.f;
    // ranges for grad A do not depend on the displacement (:51-72)
    int axmin = ceil_div_any(x - 2 * kr - md, s1), aymin = ceil_div_any(y - 2 * kr - md, s1);
    int axmax = floor_div_any(x - md, s1), aymax = floor_div_any(y - md, s1);
    const bool a_ok = (axmax >= 0) && (aymax >= 0) && (axmin <= ow - 1) && (aymin <= oh - 1);
    axmin = max(0, axmin); axmax = min(ow - 1, axmax);
    aymin = max(0, aymin); aymax = min(oh - 1, aymax);
    for (int p = -gr; p <= gr; ++p) {
      for (int o = -gr; o <= gr; ++o) {
        const int s2o = s2 * o, s2p = s2 * p;
        const int op = (p + gr) * gw + (o + gr);
        if (a_ok) {
          const int yb = y + s2p - pad, xb = x + s2o - pad;  // B0 read, zero outside the image
          if (yb >= 0 && yb < H && xb >= 0 && xb < W) {
            const float bv = b[(((long)n * H + yb) * W + xb) * C + c];
            float gs = 0.f;
            for (int y2 = aymin; y2 <= aymax; ++y2)
              for (int x2 = axmin; x2 <= axmax; ++x2) gs += g[(((long)n * oh + y2) * ow + x2) * D + op];
            sa += gs * bv;
          }
        }
        int bxmin = ceil_div_any(x - 2 * kr - md - s2o, s1), bymin = ceil_div_any(y - 2 * kr - md - s2p, s1);
        int bxmax = floor_div_any(x - md - s2o, s1), bymax = floor_div_any(y - md - s2p, s1);
        if ((bxmax >= 0) && (bymax >= 0) && (bxmin <= ow - 1) && (bymin <= oh - 1)) {
          bxmin = max(0, bxmin); bxmax = min(ow - 1, bxmax);
          bymin = max(0, bymin); bymax = min(oh - 1, bymax);
          const int ya = y - s2p - pad, xa = x - s2o - pad;  // A0 read (:165-168)
          if (ya >= 0 && ya < H && xa >= 0 && xa < W) {
            const float av = a[(((long)n * H + ya) * W + xa) * C + c];
            float gs = 0.f;
            for (int y2 = bymin; y2 <= bymax; ++y2)
              for (int x2 = bxmin; x2 <= bxmax; ++x2) gs += g[(((long)n * oh + y2) * ow + x2) * D + op];
            sb += gs * av;
          }
        }
      }
    }
    const float sumelems = (float)((2 * kr + 1) * (2 * kr + 1) * C);
    da[idx] = sa / sumelems;
    db[idx] = sb / sumelems;
  }
}

// Tiled form for the FlowNetC call site (kernel_size 1, stride_1 1, pad = max_displacement: flownet_c.py:40), where
// the reference's window sums collapse to one element:
//   dA[n,y,x,c] = (1/C) sum_{p,o} g[n,y,x,(p,o)]           * B0[n, y+s2 p, x+s2 o, c]
//   dB[n,y,x,c] = (1/C) sum_{p,o} g[n,y-s2 p,x-s2 o,(p,o)] * A0[n, y-s2 p, x-s2 o, c]      (zero outside the image)
// A block owns 16 pixels of a row, a lane one channel.  Per displacement row the lane loads its channel of the 56
// window pixels ONCE (coalesced over channels) and reuses it for the 16 x 21 (pixel, o) products; the gradient
// values are block-uniform LDS broadcasts.  The thread-per-element kernel above re-loads the window for every
// output pixel: 7.3 ms at batch 8 against 0.16 ms for the forward op.  Same accumulation order per output.
template <bool DA, int TILE, int GR, int S2>
__global__ void __launch_bounds__(256) correlation_grad_tiled_kernel(const float* __restrict__ g,
                                                                     const float* __restrict__ src,
                                                                     float* __restrict__ dst, int N, int H, int W,
                                                                     int C) {
  constexpr int GW = 2 * GR + 1, D = GW * GW, MD = GR * S2, WIN = TILE + 2 * MD;
  __shared__ float gs[DA ? TILE * D : WIN * GW];
  const int cblocks = (C + 255) / 256;
  const int n = blockIdx.z / cblocks, c = (blockIdx.z % cblocks) * 256 + threadIdx.x;
  const int y = blockIdx.y, x0 = blockIdx.x * TILE;
  const bool c_ok = c < C;
  float acc[TILE];
#pragma unroll
  for (int i = 0; i < TILE; ++i) acc[i] = 0.f;
  if constexpr (DA) {
    for (int i = threadIdx.x; i < TILE * D; i += 256) {
      const int px = i / D, d = i - px * D;
      gs[i] = (x0 + px < W) ? g[(((long)n * H + y) * W + x0 + px) * D + d] : 0.f;
    }
    __syncthreads();
  }
  for (int p = 0; p < GW; ++p) {
    const int ys = DA ? y + (p - GR) * S2 : y - (p - GR) * S2;  // block-uniform
    if (ys < 0 || ys >= H) continue;
    if constexpr (!DA) {
      __syncthreads();
      for (int i = threadIdx.x; i < WIN * GW; i += 256) {
        const int j = i / GW, o = i - j * GW;
        const int xa = x0 - MD + j;
        gs[i] = (xa >= 0 && xa < W) ? g[(((long)n * H + ys) * W + xa) * D + p * GW + o] : 0.f;
      }
      __syncthreads();
    }
    float win[WIN];
#pragma unroll
    for (int j = 0; j < WIN; ++j) {
      const int xw = x0 - MD + j;
      win[j] = (c_ok && xw >= 0 && xw < W) ? src[(((long)n * H + ys) * W + xw) * C + c] : 0.f;
    }
#pragma unroll
    for (int o = 0; o < GW; ++o)
#pragma unroll
      for (int px = 0; px < TILE; ++px) {
        if constexpr (DA) acc[px] += gs[px * D + p * GW + o] * win[px + S2 * o];
        else acc[px] += gs[(px + 2 * MD - S2 * o) * GW + o] * win[px + 2 * MD - S2 * o];
      }
  }
  if (c_ok)
#pragma unroll
    for (int px = 0; px < TILE; ++px)
      if (x0 + px < W) dst[(((long)n * H + y) * W + x0 + px) * C + c] = acc[px] / (float)C;
}

// ---------------------------------------------------------------------------
// flow_warp (flow_warp.cu.cc:44-95): one lane per pixel, all channels in the lane.
// ---------------------------------------------------------------------------
struct WarpTaps {
  bool valid;
  int xL, xR, yT, yB;
  float cTL, cTR, cBL, cBR, alpha, beta, x2, y2;
};
__device__ __forceinline__ WarpTaps warp_taps(int x, int y, float u, float v, int W, int H) {
  WarpTaps t;
  t.x2 = (float)x + u;  // :45
  t.y2 = (float)y + v;  // :46
  t.valid = (t.x2 >= 0.f) && (t.y2 >= 0.f) && (t.x2 < (float)W) && (t.y2 < (float)H);  // NaN fails, :80
  const float xs = t.valid ? t.x2 : 0.f, ys = t.valid ? t.y2 : 0.f;
  t.xL = (int)xs;  // truncation, :54
  t.yT = (int)ys;
  t.xR = min(t.xL + 1, W - 1);  // :56
  t.yB = min(t.yT + 1, H - 1);
  t.alpha = xs - (float)t.xL;  // :64
  t.beta = ys - (float)t.yT;
  t.cTL = (1.f - t.alpha) * (1.f - t.beta);  // :66-69
  t.cTR = t.alpha * (1.f - t.beta);
  t.cBL = (1.f - t.alpha) * t.beta;
  t.cBR = t.alpha * t.beta;
  return t;
}

template <int CFIX>
__global__ void __launch_bounds__(256) flow_warp_kernel(const float* __restrict__ image,
                                                        const float* __restrict__ flow,
                                                        float* __restrict__ out, int N, int H, int W, int C) {
  const long npix = (long)N * H * W;
  for (long pix = (long)blockIdx.x * blockDim.x + threadIdx.x; pix < npix;
       pix += (long)gridDim.x * blockDim.x) {
    const int x = (int)(pix % W);
    const int y = (int)((pix / W) % H);
    const long nb = (pix / W / H) * (long)H * W;
    const float2 f = *reinterpret_cast<const float2*>(flow + pix * 2);
    const WarpTaps t = warp_taps(x, y, f.x, f.y, W, H);
    const int CC = CFIX > 0 ? CFIX : C;
    const float* pTL = image + (nb + (long)t.yT * W + t.xL) * CC;
    const float* pTR = image + (nb + (long)t.yT * W + t.xR) * CC;
    const float* pBL = image + (nb + (long)t.yB * W + t.xL) * CC;
    const float* pBR = image + (nb + (long)t.yB * W + t.xR) * CC;
    float* po = out + pix * CC;
#pragma unroll
    for (int c = 0; c < CC; ++c) {
      float v = 0.f;
      if (t.valid) v = t.cTL * pTL[c] + t.cTR * pTR[c] + t.cBL * pBL[c] + t.cBR * pBR[c];  // :82-86
      po[c] = v;
    }
  }
}

// Three-channel form (every call site of the networks warps an RGB image): a lane still owns a pixel, but its three
// channels move as ONE 12-byte access -- a 4-byte-aligned packed struct compiles to global_load/store_dwordx3 -- so a
// pixel costs 6 memory instructions (flow, four taps, store) instead of 13 (the scalar form above: two loads per tap
// and three stores).  The pass is bound by vector-memory instruction issue, not by bytes or arithmetic: giving every
// lane ONE float (coalesced dwords, 6 instructions per 4 output bytes) and dropping the 64-bit index divisions both
// measured no faster (0.26 ms at batch 64 either way).  UNR pixels per lane are in flight.
typedef rgb3_t WarpRGB;
template <int UNR>
__global__ void __launch_bounds__(256) flow_warp_rgb_kernel(const float* __restrict__ image,
                                                            const float* __restrict__ flow,
                                                            float* __restrict__ out, int N, int H, int W) {
  const long npix = (long)N * H * W;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long p0 = (long)blockIdx.x * blockDim.x + threadIdx.x; p0 < npix; p0 += stride * UNR) {
    float2 f[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const long pix = p0 + u * stride;
      f[u] = pix < npix ? *reinterpret_cast<const float2*>(flow + pix * 2) : make_float2(0.f, 0.f);
    }
    WarpTaps t[UNR];
    WarpRGB tl[UNR], tr[UNR], bl[UNR], br[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const long pix = p0 + u * stride;
      const long pc = pix < npix ? pix : 0;
      const int x = (int)(pc % W);
      const int y = (int)((pc / W) % H);
      const long nb = (pc / W / H) * (long)H * W;
      t[u] = warp_taps(x, y, f[u].x, f[u].y, W, H);
      tl[u] = *reinterpret_cast<const WarpRGB*>(image + (nb + (long)t[u].yT * W + t[u].xL) * 3);
      tr[u] = *reinterpret_cast<const WarpRGB*>(image + (nb + (long)t[u].yT * W + t[u].xR) * 3);
      bl[u] = *reinterpret_cast<const WarpRGB*>(image + (nb + (long)t[u].yB * W + t[u].xL) * 3);
      br[u] = *reinterpret_cast<const WarpRGB*>(image + (nb + (long)t[u].yB * W + t[u].xR) * 3);
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const long pix = p0 + u * stride;
      if (pix >= npix) continue;
      WarpRGB o = {0.f, 0.f, 0.f};
      if (t[u].valid) {  // :82-86, same expression per channel as the scalar form
        o.r = t[u].cTL * tl[u].r + t[u].cTR * tr[u].r + t[u].cBL * bl[u].r + t[u].cBR * br[u].r;
        o.g = t[u].cTL * tl[u].g + t[u].cTR * tr[u].g + t[u].cBL * bl[u].g + t[u].cBR * br[u].g;
        o.b = t[u].cTL * tl[u].b + t[u].cTR * tr[u].b + t[u].cBL * bl[u].b + t[u].cBR * br[u].b;
      }
      *reinterpret_cast<WarpRGB*>(out + pix * 3) = o;
    }
  }
}

// flow_warp backward (flow_warp_grad.cu.cc:30-86).  image_grad must be zeroed by the caller
// (hipMemsetAsync on the same stream in the launcher); flow_grad is written for every pixel.
__global__ void __launch_bounds__(256) flow_warp_grad_kernel(const float* __restrict__ image,
                                                             const float* __restrict__ flow,
                                                             const float* __restrict__ grad,
                                                             float* __restrict__ image_grad,
                                                             float* __restrict__ flow_grad, int N, int H,
                                                             int W, int C) {
  const long npix = (long)N * H * W;
  for (long pix = (long)blockIdx.x * blockDim.x + threadIdx.x; pix < npix;
       pix += (long)gridDim.x * blockDim.x) {
    const int x = (int)(pix % W);
    const int y = (int)((pix / W) % H);
    const long nb = (pix / W / H) * (long)H * W;
    const float2 f = *reinterpret_cast<const float2*>(flow + pix * 2);
    const WarpTaps t = warp_taps(x, y, f.x, f.y, W, H);
    float du = 0.f, dv = 0.f;
    if (t.valid) {
      const long oTL = (nb + (long)t.yT * W + t.xL) * C, oTR = (nb + (long)t.yT * W + t.xR) * C;
      const long oBL = (nb + (long)t.yB * W + t.xL) * C, oBR = (nb + (long)t.yB * W + t.xR) * C;
      const float gy = (float)t.yB - t.y2;  // gamma = iy2_B - y2, :54
      const float gx = (float)t.xR - t.x2;  // gamma = ix2_R - x2, :71
      for (int c = 0; c < C; ++c) {
        const float g = grad[pix * C + c];
        atomicAdd(image_grad + oTL + c, g * t.cTL);  // :43-52
        atomicAdd(image_grad + oTR + c, g * t.cTR);
        atomicAdd(image_grad + oBL + c, g * t.cBL);
        atomicAdd(image_grad + oBR + c, g * t.cBR);
        const float TL = image[oTL + c], TR = image[oTR + c], BL = image[oBL + c], BR = image[oBR + c];
        du += g * (gy * (TR - TL) + (1.f - gy) * (BR - BL));  // :57-69
        dv += g * (gx * (BL - TL) + (1.f - gx) * (BR - TR));  // :74-85
      }
    }
    *reinterpret_cast<float2*>(flow_grad + pix * 2) = make_float2(du, dv);
  }
}

// ---------------------------------------------------------------------------
// downsample (downsample_kernel_gpu.cu.cc:35-76): one wavefront per output
// pixel (all channels, C <= 4 per pass), lanes stride the tap window, wave64
// shuffle reduction of the three accumulators.
// ---------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__global__ void __launch_bounds__(256) downsample_kernel(const float* __restrict__ in,
                                                         float* __restrict__ out, int N, int Hin, int Win,
                                                         int C, int oh, int ow, float wscale, float hscale,
                                                         int wr, int hr, float in_scale) {
  const int lane = threadIdx.x & 63;
  const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
  const long nout = (long)N * oh * ow;
  for (long o = wave; o < nout; o += nwaves) {
    const int dx = (int)(o % ow);
    const int dy = (int)((o / ow) % oh);
    const int n = (int)(o / ow / oh);
    const float srcx = ((float)dx / (float)(ow - 1)) * (float)(Win - 1);  // :41
    const float srcy = ((float)dy / (float)(oh - 1)) * (float)(Hin - 1);  // :42
    const int ix = (int)roundf(srcx), iy = (int)roundf(srcy);             // :44-45
    const int ww = 2 * wr + 1, wh = 2 * hr + 1;
    for (int c0 = 0; c0 < C; c0 += 4) {
      float av[4] = {0, 0, 0, 0}, aw[4] = {0, 0, 0, 0}, an[4] = {0, 0, 0, 0};
      for (int t = lane; t < ww * wh; t += 64) {
        const int xo = ix - wr + t % ww, yo = iy - hr + t / ww;
        if (xo >= 0 && yo >= 0 && xo < Win && yo < Hin) {
          const float wgt = fmaxf(0.f, 1.f - fabsf((float)xo - srcx) / wscale) *
                            fmaxf(0.f, 1.f - fabsf((float)yo - srcy) / hscale);  // :57-58
          const float* p = in + (((long)n * Hin + yo) * Win + xo) * C + c0;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            if (c0 + c < C) {
              const float s = p[c] * in_scale;
              if (s != s) {  // NaN sample: counts as NaN weight only, :59-63
                an[c] += wgt;
              } else {
                av[c] += s * wgt;
                aw[c] += wgt;
              }
            }
          }
        }
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float v = wave_sum(av[c]), w = wave_sum(aw[c]), nn = wave_sum(an[c]);
        if (lane == 0 && c0 + c < C)
          out[o * C + c0 + c] = (nn / w > 0.5f) ? __int_as_float(0x7fffffff) : v / w;  // :68-72
      }
    }
  }
}

// Separable form of the same op, one block per output ROW (n, dy).  The tap weight is a product wx(xo) * wy(yo)
// and the three sums (value, weight, NaN weight) are linear in it, so
//   sum_{yo,xo} wy wx f(xo,yo) = sum_yo wy ( sum_xo wx f(xo,yo) ).
// Phase 1 writes the horizontal sums of every window row to LDS ([row][dx][c] x 3 floats), phase 2 combines them
// vertically.  O(window) instead of O(window^2) work per output; only the summation ORDER differs from the
// reference's serial loop (fp32 rounding, ~1e-7 relative).  Used when the LDS image fits (host check).
// Block per output row (n, dy), a thread per (window row, dx, c) walks its few taps.
__global__ void __launch_bounds__(256) downsample_sep_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                             int N, int Hin, int Win, int C, int oh, int ow,
                                                             float wscale, float hscale, int wr, int hr, float in_scale) {
  extern __shared__ float rows[];  // [wh][ow*C][3]
  const int dy = blockIdx.x % oh, n = blockIdx.x / oh;
  const float srcy = ((float)dy / (float)(oh - 1)) * (float)(Hin - 1);
  const int iy = (int)roundf(srcy);
  const int wh = 2 * hr + 1, ww = 2 * wr + 1, rowlen = ow * C;
  if (C == 2) {
    // two-channel tensors (the flow fields the loss downsamples): a lane owns both channels of a (window row, dx)
    // and reads each tap as ONE 8-byte load -- the pass is bound by memory-instruction issue (see flow_warp)
    for (int item = threadIdx.x; item < wh * ow; item += blockDim.x) {
      const int r = item / ow, dx = item - r * ow;
      const int yo = iy - hr + r;
      float av0 = 0.f, aw0 = 0.f, an0 = 0.f, av1 = 0.f, aw1 = 0.f, an1 = 0.f;
      if (yo >= 0 && yo < Hin) {
        const float srcx = ((float)dx / (float)(ow - 1)) * (float)(Win - 1);
        const int ix = (int)roundf(srcx);
        const float2* p = reinterpret_cast<const float2*>(in) + ((long)n * Hin + yo) * Win;
        for (int k = 0; k < ww; ++k) {
          const int xo = ix - wr + k;
          if (xo < 0 || xo >= Win) continue;
          const float wgt = fmaxf(0.f, 1.f - fabsf((float)xo - srcx) / wscale);
          const float2 sv = make_float2(p[xo].x * in_scale, p[xo].y * in_scale);
          if (sv.x != sv.x) an0 += wgt; else { av0 += sv.x * wgt; aw0 += wgt; }
          if (sv.y != sv.y) an1 += wgt; else { av1 += sv.y * wgt; aw1 += wgt; }
        }
      }
      float* q = rows + ((long)r * rowlen + dx * 2) * 3;
      q[0] = av0; q[1] = aw0; q[2] = an0; q[3] = av1; q[4] = aw1; q[5] = an1;
    }
  } else
  for (int item = threadIdx.x; item < wh * rowlen; item += blockDim.x) {
    const int r = item / rowlen, e = item - r * rowlen;
    const int dx = e / C, c = e - dx * C;
    const int yo = iy - hr + r;
    float av = 0.f, aw = 0.f, an = 0.f;
    if (yo >= 0 && yo < Hin) {
      const float srcx = ((float)dx / (float)(ow - 1)) * (float)(Win - 1);
      const int ix = (int)roundf(srcx);
      const float* p = in + ((long)n * Hin + yo) * Win * C + c;
      for (int k = 0; k < ww; ++k) {
        const int xo = ix - wr + k;
        if (xo < 0 || xo >= Win) continue;
        const float wgt = fmaxf(0.f, 1.f - fabsf((float)xo - srcx) / wscale);
        const float sv = p[(long)xo * C] * in_scale;
        if (sv != sv) an += wgt; else { av += sv * wgt; aw += wgt; }
      }
    }
    rows[item * 3 + 0] = av; rows[item * 3 + 1] = aw; rows[item * 3 + 2] = an;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < rowlen; e += blockDim.x) {
    float av = 0.f, aw = 0.f, an = 0.f;
    for (int r = 0; r < wh; ++r) {
      const int yo = iy - hr + r;
      if (yo < 0 || yo >= Hin) continue;
      const float wy = fmaxf(0.f, 1.f - fabsf((float)yo - srcy) / hscale);
      const float* q = rows + ((long)r * rowlen + e) * 3;
      av += wy * q[0]; aw += wy * q[1]; an += wy * q[2];
    }
    out[((long)n * oh + dy) * rowlen + e] = (an / aw > 0.5f) ? __int_as_float(0x7fffffff) : av / aw;
  }
}

// Wide windows (>= 32 taps per row, i.e. the coarse loss scales: few outputs, thousands of taps each): a block of
// 16 wavefronts per output PIXEL.  A wavefront per window row, lanes stride the taps of the row (coalesced), all
// C <= 4 channels per tap, shuffle reduction, vertical weights applied per row, then one LDS reduction over rows.
__global__ void __launch_bounds__(1024) downsample_wide_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                               int N, int Hin, int Win, int C, int oh, int ow,
                                                               float wscale, float hscale, int wr, int hr, float in_scale) {
  __shared__ float part[16][4][3];
  const int dx = blockIdx.x % ow, dy = (blockIdx.x / ow) % oh, n = blockIdx.x / ow / oh;
  const float srcx = ((float)dx / (float)(ow - 1)) * (float)(Win - 1);
  const float srcy = ((float)dy / (float)(oh - 1)) * (float)(Hin - 1);
  const int ix = (int)roundf(srcx), iy = (int)roundf(srcy);
  const int wh = 2 * hr + 1, ww = 2 * wr + 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
  float av[4] = {0, 0, 0, 0}, aw[4] = {0, 0, 0, 0}, an[4] = {0, 0, 0, 0};
  for (int r = wave; r < wh; r += nwave) {
    const int yo = iy - hr + r;
    if (yo < 0 || yo >= Hin) continue;
    const float wy = fmaxf(0.f, 1.f - fabsf((float)yo - srcy) / hscale);
    const float* p = in + ((long)n * Hin + yo) * Win * C;
    for (int k = lane; k < ww; k += 64) {
      const int xo = ix - wr + k;
      if (xo < 0 || xo >= Win) continue;
      const float wgt = wy * fmaxf(0.f, 1.f - fabsf((float)xo - srcx) / wscale);
      if (C == 2) {  // both channels of a tap as one 8-byte load
        const float2 raw = reinterpret_cast<const float2*>(p)[xo];
        const float2 sv = make_float2(raw.x * in_scale, raw.y * in_scale);
        if (sv.x != sv.x) an[0] += wgt; else { av[0] += sv.x * wgt; aw[0] += wgt; }
        if (sv.y != sv.y) an[1] += wgt; else { av[1] += sv.y * wgt; aw[1] += wgt; }
        continue;
      }
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (c < C) {
          const float sv = p[(long)xo * C + c] * in_scale;
          if (sv != sv) an[c] += wgt; else { av[c] += sv * wgt; aw[c] += wgt; }
        }
    }
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float a = wave_sum(av[c]), b = wave_sum(aw[c]), d = wave_sum(an[c]);
    if (lane == 0) { part[wave][c][0] = a; part[wave][c][1] = b; part[wave][c][2] = d; }
  }
  __syncthreads();
  if (threadIdx.x < C) {
    float a = 0.f, b = 0.f, d = 0.f;
    for (int w = 0; w < nwave; ++w) { a += part[w][threadIdx.x][0]; b += part[w][threadIdx.x][1]; d += part[w][threadIdx.x][2]; }
    out[(long)blockIdx.x * C + threadIdx.x] = (d / b > 0.5f) ? __int_as_float(0x7fffffff) : a / b;
  }
}

// ---------------------------------------------------------------------------
// resize_bilinear(align_corners=True) of scale*in (flownet_s.py:107-111)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) resize_bilinear_kernel(const float* __restrict__ in,
                                                              float* __restrict__ out, int N, int Hin,
                                                              int Win, int C, int oh, int ow, float sy,
                                                              float sx, float scale) {
  const long total = (long)N * oh * ow;
  for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
    const int x = (int)(o % ow);
    const int y = (int)((o / ow) % oh);
    const int n = (int)(o / ow / oh);
    const float fy = (float)y * sy, fx = (float)x * sx;
    const int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
    const int y1 = min(y0 + 1, Hin - 1), x1 = min(x0 + 1, Win - 1);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float* b = in + (long)n * Hin * Win * C;
    for (int c = 0; c < C; ++c) {
      const float tl = b[((long)y0 * Win + x0) * C + c], tr = b[((long)y0 * Win + x1) * C + c];
      const float bl = b[((long)y1 * Win + x0) * C + c], br = b[((long)y1 * Win + x1) * C + c];
      const float top = tl + (tr - tl) * lx, bot = bl + (br - bl) * lx;
      out[o * C + c] = (top + (bot - top) * ly) * scale;
    }
  }
}

// Two-channel form (the flow fields: every caller in the networks): a lane owns two adjacent output pixels -- one
// 16-byte store, eight 8-byte tap loads (neighbouring outputs share taps: L1 hits).  Same arithmetic per element.
__global__ void __launch_bounds__(256) resize_bilinear_c2_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                                 int N, int Hin, int Win, int oh, int ow, float sy,
                                                                 float sx, float scale) {
  // block = (output row blockIdx.y, 256 pixel pairs of it): no per-element 64-bit divisions
  const int y = blockIdx.y % oh, n = blockIdx.y / oh;
  const int xp = blockIdx.x * 256 + threadIdx.x;
  if (2 * xp >= ow) return;
  const float fy = (float)y * sy;
  const int y0 = (int)floorf(fy);
  const int y1 = min(y0 + 1, Hin - 1);
  const float ly = fy - (float)y0;
  const float2* r0 = reinterpret_cast<const float2*>(in) + ((long)n * Hin + y0) * Win;
  const float2* r1 = reinterpret_cast<const float2*>(in) + ((long)n * Hin + y1) * Win;
  float res[4];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const float fx = (float)(2 * xp + j) * sx;
    const int x0 = (int)floorf(fx);
    const int x1 = min(x0 + 1, Win - 1);
    const float lx = fx - (float)x0;
    const float2 r = bilerp_c2(r0[x0], r0[x1], r1[x0], r1[x1], lx, ly, scale);
    res[2 * j] = r.x;
    res[2 * j + 1] = r.y;
  }
  *reinterpret_cast<float4*>(out + ((long)blockIdx.y * ow + 2 * xp) * 2) = make_float4(res[0], res[1], res[2], res[3]);
}

static inline int grid_for(long work_items, int block) {
  long g = (work_items + block - 1) / block;
  if (g > 256L * 16) g = 256L * 16;  // 256 CUs x 16 blocks, grid-stride the rest
  if (g < 1) g = 1;
  return (int)g;
}

int correlation_geometry(int h, int w, int k, int md, int s1, int s2, int pad, int* oh, int* ow, int* gr,
                         int* gw) {
  FN2_REQUIRE(k > 0 && (k % 2) != 0, "kernel_size must be odd");  // correlation_kernel.cc:23
  FN2_REQUIRE(md >= 0 && s1 >= 1 && s2 >= 1 && pad >= 0, "correlation: bad attributes");
  const int kr = (k - 1) / 2, border = md + kr;
  const int hp = h + 2 * pad, wp = w + 2 * pad;
  *oh = (int)ceilf((float)(hp - border * 2) / (float)s1);  // :50
  *ow = (int)ceilf((float)(wp - border * 2) / (float)s1);  // :51
  FN2_REQUIRE(*oh >= 1, "Neighborhood and kernel don't fit in input height.");  // :53
  FN2_REQUIRE(*ow >= 1, "Neighborhood and kernel don't fit in input width.");   // :55
  *gr = md / s2;
  *gw = 2 * (*gr) + 1;
  // The reference never bounds-checks the displaced read; outside pad >= gr*s2 - ... it is UB.
  const int lo = md - (*gr) * s2;
  const int hiy = (*oh - 1) * s1 + md + (*gr) * s2 + k - 1, hix = (*ow - 1) * s1 + md + (*gr) * s2 + k - 1;
  FN2_REQUIRE(lo >= 0 && hiy < hp && hix < wp,
              "displacement window leaves the padded input (undefined in the reference)");
  return FN2_OK;
}

}  // namespace fn2

using namespace fn2;

extern "C" {

const char* fn2_last_error(void) { return fn2::err_buf(); }
int fn2_version(void) { return 1; }

int fn2_device_info(char* name, int cap, int* cus) {
  int dev = 0;
  FN2_HIP(hipGetDevice(&dev));
  hipDeviceProp_t p;
  FN2_HIP(hipGetDeviceProperties(&p, dev));
  if (name && cap > 0) snprintf(name, cap, "%s", p.gcnArchName);
  if (cus) *cus = p.multiProcessorCount;
  return FN2_OK;
}

int fn2_correlation_out_shape(int h, int w, int k, int md, int s1, int s2, int pad, int* oh, int* ow,
                              int* oc) {
  int gr, gw;
  int rc = correlation_geometry(h, w, k, md, s1, s2, pad, oh, ow, &gr, &gw);
  if (rc) return rc;
  *oc = gw * gw;
  return FN2_OK;
}

int fn2_correlation_generic_f32(const float* a, const float* b, float* out, int n, int h, int w, int c,
                                int k, int md, int s1, int s2, int pad, void* stream) {
  FN2_REQUIRE(a && b && out, "correlation: null pointer");
  FN2_REQUIRE(n >= 1 && h >= 1 && w >= 1 && c >= 1, "correlation: inputs must have rank 4 with positive dims");
  int oh, ow, gr, gw;
  int rc = correlation_geometry(h, w, k, md, s1, s2, pad, &oh, &ow, &gr, &gw);
  if (rc) return rc;
  const long total = (long)n * oh * ow * gw * gw;
  hipLaunchKernelGGL(correlation_generic_kernel, dim3(grid_for(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, a, b, out, n, h, w, c, k, md, s1, s2, pad, oh, ow, gr, gw);
  FN2_CHECK_LAUNCH("correlation_generic");
  return FN2_OK;
}

int fn2_correlation_grad_f32(const float* g, const float* a, const float* b, float* da, float* db, int n,
                             int h, int w, int c, int k, int md, int s1, int s2, int pad, void* stream) {
  FN2_REQUIRE(g && a && b && da && db, "correlation_grad: null pointer");
  FN2_REQUIRE(n >= 1 && h >= 1 && w >= 1 && c >= 1, "correlation_grad: bad dims");
  int oh, ow, gr, gw;
  int rc = correlation_geometry(h, w, k, md, s1, s2, pad, &oh, &ow, &gr, &gw);
  if (rc) return rc;
  const long total = (long)n * h * w * c;
  if (k == 1 && s1 == 1 && s2 == 2 && md == 20 && pad == md && (long)n * ((c + 255) / 256) <= 65535 && h <= 65535) {
    // the FlowNetC attribute set: tiled kernels, one launch per gradient
    const dim3 grid((w + 15) / 16, h, n * ((c + 255) / 256));
    hipLaunchKernelGGL((correlation_grad_tiled_kernel<true, 16, 10, 2>), grid, dim3(256), 0, (hipStream_t)stream, g, b, da,
                       n, h, w, c);
    hipLaunchKernelGGL((correlation_grad_tiled_kernel<false, 16, 10, 2>), grid, dim3(256), 0, (hipStream_t)stream, g, a, db,
                       n, h, w, c);
    FN2_CHECK_LAUNCH("correlation_grad_tiled");
    return FN2_OK;
  }
  hipLaunchKernelGGL(correlation_grad_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     g, a, b, da, db, n, h, w, c, k, md, s1, s2, pad, oh, ow, gr, gw);
  FN2_CHECK_LAUNCH("correlation_grad");
  return FN2_OK;
}

int fn2_flow_warp_f32(const float* image, const float* flow, float* out, int n, int h, int w, int c,
                      void* stream) {
  FN2_REQUIRE(image && flow && out, "flow_warp: null pointer");
  FN2_REQUIRE(n >= 1 && h >= 1 && w >= 1 && c >= 1, "Input images must have rank 4");  // flow_warp.cc:22
  const long npix = (long)n * h * w;
  if (c == 3)
    hipLaunchKernelGGL(flow_warp_rgb_kernel<2>, dim3(grid_for(npix / 2, 256)), dim3(256), 0, (hipStream_t)stream, image, flow,
                       out, n, h, w);
  else
    hipLaunchKernelGGL(flow_warp_kernel<0>, dim3(grid_for(npix, 256)), dim3(256), 0, (hipStream_t)stream, image,
                       flow, out, n, h, w, c);
  FN2_CHECK_LAUNCH("flow_warp");
  return FN2_OK;
}

int fn2_flow_warp_grad_f32(const float* image, const float* flow, const float* grad, float* image_grad,
                           float* flow_grad, int n, int h, int w, int c, void* stream) {
  FN2_REQUIRE(image && flow && grad && image_grad && flow_grad, "flow_warp_grad: null pointer");
  FN2_REQUIRE(n >= 1 && h >= 1 && w >= 1 && c >= 1, "flow_warp_grad: bad dims");
  const long npix = (long)n * h * w;
  FN2_HIP(hipMemsetAsync(image_grad, 0, sizeof(float) * npix * c, (hipStream_t)stream));  // same stream
  hipLaunchKernelGGL(flow_warp_grad_kernel, dim3(grid_for(npix, 256)), dim3(256), 0, (hipStream_t)stream, image,
                     flow, grad, image_grad, flow_grad, n, h, w, c);
  FN2_CHECK_LAUNCH("flow_warp_grad");
  return FN2_OK;
}

int fn2_downsample_f32(const float* in, float* out, int n, int in_h, int in_w, int c, int out_h, int out_w,
                       void* stream) {
  return fn2_downsample_scaled_f32(in, 1.0f, out, n, in_h, in_w, c, out_h, out_w, stream);  // (x * 1.0f == x, NaN included)
}

int fn2_downsample_scaled_f32(const float* in, float in_scale, float* out, int n, int in_h, int in_w, int c, int out_h,
                              int out_w, void* stream) {
  FN2_REQUIRE(in && out, "downsample: null pointer");
  FN2_REQUIRE(n >= 1 && in_h >= 1 && in_w >= 1 && c >= 1, "Input images must have rank 4");  // downsample_kernel.cc:25
  FN2_REQUIRE(out_h >= 1 && out_w >= 1, "downsample: size must be positive");
  const float wscale = (float)(in_w - 1) / (float)(out_w - 1);  // :92 (division by zero not guarded, as the reference)
  const float hscale = (float)(in_h - 1) / (float)(out_h - 1);
  FN2_REQUIRE(out_h > 1 && out_w > 1, "downsample: output size 1 divides by zero in the reference kernel");
  const int wr = (int)ceilf(wscale), hr = (int)ceilf(hscale);
  const long nout = (long)n * out_h * out_w;
  const size_t lds = (size_t)(2 * hr + 1) * out_w * c * 3 * sizeof(float);
  if (2 * wr + 1 >= 32 && c <= 4 && nout < (1L << 30))
    hipLaunchKernelGGL(downsample_wide_kernel, dim3((unsigned)nout), dim3(1024), 0, (hipStream_t)stream, in, out, n, in_h,
                       in_w, c, out_h, out_w, wscale, hscale, wr, hr, in_scale);
  else if (lds <= 64 * 1024 && (long)n * out_h < (1L << 30))
    hipLaunchKernelGGL(downsample_sep_kernel, dim3(n * out_h), dim3(256), lds, (hipStream_t)stream, in, out, n,
                       in_h, in_w, c, out_h, out_w, wscale, hscale, wr, hr, in_scale);
  else
    hipLaunchKernelGGL(downsample_kernel, dim3(grid_for(nout * 64, 256)), dim3(256), 0, (hipStream_t)stream, in,
                       out, n, in_h, in_w, c, out_h, out_w, wscale, hscale, wr, hr, in_scale);
  FN2_CHECK_LAUNCH("downsample");
  return FN2_OK;
}

int fn2_resize_bilinear_f32(const float* in, float* out, int n, int in_h, int in_w, int c, int out_h,
                            int out_w, float scale, void* stream) {
  FN2_REQUIRE(in && out, "resize_bilinear: null pointer");
  FN2_REQUIRE(n >= 1 && in_h >= 1 && in_w >= 1 && c >= 1 && out_h >= 1 && out_w >= 1, "resize_bilinear: bad dims");
  const float sy = out_h > 1 ? (float)(in_h - 1) / (float)(out_h - 1) : 0.f;
  const float sx = out_w > 1 ? (float)(in_w - 1) / (float)(out_w - 1) : 0.f;
  const long total = (long)n * out_h * out_w;
  if (c == 2 && out_w % 2 == 0 && (long)n * out_h <= 65535)
    hipLaunchKernelGGL(resize_bilinear_c2_kernel, dim3((out_w / 2 + 255) / 256, n * out_h), dim3(256), 0,
                       (hipStream_t)stream, in, out, n, in_h, in_w, out_h, out_w, sy, sx, scale);
  else
    hipLaunchKernelGGL(resize_bilinear_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, in,
                       out, n, in_h, in_w, c, out_h, out_w, sy, sx, scale);
  FN2_CHECK_LAUNCH("resize_bilinear");
  return FN2_OK;
}

int fn2_capture_begin(void* stream) {
  FN2_HIP(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
  return FN2_OK;
}
int fn2_capture_end(void* stream, void** graph_exec) {
  hipGraph_t g = nullptr;
  FN2_HIP(hipStreamEndCapture((hipStream_t)stream, &g));
  hipGraphExec_t ge = nullptr;
  hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (e != hipSuccess) return fn2::fail(FN2_ERR_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e));
  *graph_exec = (void*)ge;
  return FN2_OK;
}
int fn2_graph_launch(void* graph_exec, void* stream) {
  FN2_HIP(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream));
  return FN2_OK;
}
int fn2_graph_destroy(void* graph_exec) {
  FN2_HIP(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
  return FN2_OK;
}

}  // extern "C"
