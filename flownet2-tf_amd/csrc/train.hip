// Training-step kernels (fp32): what the reference gets from tf.gradients + tf.train.AdamOptimizer
// (src/net.py:1290-1295, :1386-1392) for the FlowNetS loss (src/flownet_s/flownet_s.py:122-161).
//   * input gradients of conv / transposed-conv layers reuse fn2_conv2d (kind 0 on rotated weights,
//     kind 3 = transpose of a stride-2 conv, kind 0 k4 s2 for the transposed convs) with accumulate;
//   * this file: the filter-gradient GEMM on the fp32 matrix cores, the flow-head filter gradient,
//     bias gradient, LeakyReLU backward, the multiscale EPE loss + its gradient, upsample_flow
//     backward, the weight re-layout gather and the Adam update.
#include "fn2_common.h"

namespace fn2 {

typedef __attribute__((address_space(3))) void* lptr_t;
constexpr unsigned kOobT = 0x80000000u;

static inline int grid_for(long work_items, int block) {
  long g = (work_items + block - 1) / block;
  if (g > 256L * 16) g = 256L * 16;
  if (g < 1) g = 1;
  return (int)g;
}

__device__ __forceinline__ float wave_sum_t(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// ---------------------------------------------------------------------------
// average_endpoint_error (utils.py:209-224) of one scale and its gradient:
//   L = weight * sum_{n,y,x} ||pred - label||_2 / N ;  dpred = weight / N * (pred - label) / ||pred - label||
// (0 where the difference is exactly 0).  loss_accum += L (one atomic per block).
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) epe_loss_grad_kernel(const float* __restrict__ pred,
                                                            const float* __restrict__ label, float* __restrict__ dpred,
                                                            float* __restrict__ loss_accum, long npix, float scale) {
  float part = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
    const float2 p = *reinterpret_cast<const float2*>(pred + 2 * i);
    const float2 l = *reinterpret_cast<const float2*>(label + 2 * i);
    const float du = p.x - l.x, dv = p.y - l.y;
    const float e = sqrtf(du * du + dv * dv);
    part += e;
    const float inv = e > 0.f ? scale / e : 0.f;
    *reinterpret_cast<float2*>(dpred + 2 * i) = make_float2(du * inv, dv * inv);
  }
  part = wave_sum_t(part);
  __shared__ float s[4];
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = part;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(loss_accum, (s[0] + s[1] + s[2] + s[3]) * scale);
}

// LeakyReLU backward in place on a channel slice, fused with the bias gradient of the same layer:
//   g *= d/dx (0.55 x + 0.45 |x|) evaluated from the layer OUTPUT y (same sign as x): 1 for y > 0, 0.1 for
//   y < 0, 0.55 at 0 (tf.abs' = sign);   db[c] += sum over pixels of the new g[pixel][c].
// ACT = false: bias gradient only (g is not written).  db == nullptr: activation only.
// Block = a pixel range x GPB float4 channel groups (GPB a power of two <= 256, threads beyond c/4 idle);
// consecutive threads read consecutive 16-byte groups of a pixel, the 256/GPB thread rows stride the pixels.
// HBM-bound: reads y and g once, writes g once.
template <bool ACT>
__global__ void __launch_bounds__(256) act_bias_bwd_kernel(const float* __restrict__ y, float* __restrict__ g,
                                                           float* __restrict__ db, long npix, int c, int y_cs, int y_c0,
                                                           int g_cs, int g_c0, int gpb_log2) {
  const int gpb = 1 << gpb_log2, rows = 256 >> gpb_log2;
  const int tc = threadIdx.x & (gpb - 1), tr = threadIdx.x >> gpb_log2;
  const int ch = (blockIdx.x * gpb + tc) * 4;
  const long per = (npix + gridDim.y - 1) / gridDim.y;
  const long p0 = (long)blockIdx.y * per, p1 = min(npix, p0 + per);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  auto slope = [](float yv) { return yv > 0.f ? 1.f : (yv < 0.f ? 0.1f : 0.55f); };
  if (ch < c) {
    // four pixels per trip: all eight loads are issued before the first use (one wave keeps 8 x 1 KB in flight;
    // with one pixel per trip the pass ran at a third of the bandwidth)
    long pix = p0 + tr;
    for (; pix + 3L * rows < p1; pix += 4L * rows) {
      float4 gv[4], yv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        gv[u] = *reinterpret_cast<const float4*>(g + (pix + (long)u * rows) * g_cs + g_c0 + ch);
        if constexpr (ACT) yv[u] = *reinterpret_cast<const float4*>(y + (pix + (long)u * rows) * y_cs + y_c0 + ch);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if constexpr (ACT) {
          gv[u].x *= slope(yv[u].x); gv[u].y *= slope(yv[u].y); gv[u].z *= slope(yv[u].z); gv[u].w *= slope(yv[u].w);
          *reinterpret_cast<float4*>(g + (pix + (long)u * rows) * g_cs + g_c0 + ch) = gv[u];
        }
        acc.x += gv[u].x; acc.y += gv[u].y; acc.z += gv[u].z; acc.w += gv[u].w;
      }
    }
    for (; pix < p1; pix += rows) {
      float4* gp = reinterpret_cast<float4*>(g + pix * g_cs + g_c0 + ch);
      float4 gv = *gp;
      if constexpr (ACT) {
        const float4 yv = *reinterpret_cast<const float4*>(y + pix * y_cs + y_c0 + ch);
        gv.x *= slope(yv.x); gv.y *= slope(yv.y); gv.z *= slope(yv.z); gv.w *= slope(yv.w);
        *gp = gv;
      }
      acc.x += gv.x; acc.y += gv.y; acc.z += gv.z; acc.w += gv.w;
    }
  }
  if (db == nullptr) return;
  __shared__ float4 s[256];
  s[threadIdx.x] = acc;
  __syncthreads();
  if (tr == 0 && ch < c) {
    for (int r = 1; r < rows; ++r) {
      const float4 t = s[r * gpb + tc];
      acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
    }
    atomicAdd(db + ch, acc.x); atomicAdd(db + ch + 1, acc.y); atomicAdd(db + ch + 2, acc.z); atomicAdd(db + ch + 3, acc.w);
  }
}

// bias gradient of a dense 2-channel tensor (the flow heads): db[0..1] += sum of g over pixels.
__global__ void __launch_bounds__(256) bias_grad2_kernel(const float* __restrict__ g, float* __restrict__ db, long npix) {
  float a0 = 0.f, a1 = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
    const float2 v = *reinterpret_cast<const float2*>(g + 2 * i);
    a0 += v.x; a1 += v.y;
  }
  a0 = wave_sum_t(a0); a1 = wave_sum_t(a1);
  if ((threadIdx.x & 63) == 0) { atomicAdd(db, a0); atomicAdd(db + 1, a1); }
}

// generic fallback (channel count not a multiple of 4): block = 64 channels x a pixel range
__global__ void __launch_bounds__(256) bias_grad_kernel(const float* __restrict__ g, float* __restrict__ db, long npix,
                                                        int c, int g_cs, int g_c0) {
  const int ch = blockIdx.x * 64 + (threadIdx.x & 63);
  const int sub = threadIdx.x >> 6;
  const long per = (npix + gridDim.y - 1) / gridDim.y;
  const long p0 = (long)blockIdx.y * per, p1 = min(npix, p0 + per);
  float acc = 0.f;
  if (ch < c)
    for (long pix = p0 + sub; pix < p1; pix += 4) acc += g[pix * g_cs + g_c0 + ch];
  __shared__ float s[4][64];
  s[sub][threadIdx.x & 63] = acc;
  __syncthreads();
  if (sub == 0 && ch < c) atomicAdd(db + ch, s[0][threadIdx.x] + s[1][threadIdx.x] + s[2][threadIdx.x] + s[3][threadIdx.x]);
}

// dst[i] = map[i] >= 0 ? src[map[i]] : 0  -- derives the transposed / phase-decomposed weights that the
// input-gradient convolutions read from the master (forward-layout) weights once per step.
__global__ void __launch_bounds__(256) gather_kernel(float* __restrict__ dst, const float* __restrict__ src,
                                                     const int* __restrict__ map, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int m = map[i];
    dst[i] = m >= 0 ? src[m] : 0.f;
  }
}

// Adam (tf.train.AdamOptimizer form: lr_t = lr*sqrt(1-b2^t)/(1-b1^t), eps outside the sqrt) with the
// slim L2 regulariser folded in: g' = g + l2*w  (flownet_s.py:36-37; training_schedules.py:46-53).
__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ w, float* __restrict__ m, float* __restrict__ v,
                                                   const float* __restrict__ g, long n, float lr_t, float b1, float b2,
                                                   float eps, float l2, float gscale) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float wi = w[i];
    const float gi = g[i] * gscale + l2 * wi;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    w[i] = wi - lr_t * mi / (sqrtf(vi) + eps);
  }
}

// upsample_flowXtoY backward (forward: conv-transpose 2->2, k4 s2 crop 1, upsample_flow_kernel in conv.hip):
//   dpf[n,y,x,i] (+)= sum_{ky,kx,o} g[n,2y+ky-1,2x+kx-1,o] * w[ky,kx,o,i];  dw[ky,kx,o,i] += sum g * pf
__global__ void __launch_bounds__(256) upsample_flow_bwd_kernel(const float* __restrict__ g, int g_cs, int g_c0,
                                                                const float* __restrict__ pf, const float* __restrict__ w,
                                                                float* __restrict__ dpf, float* __restrict__ dw, int N,
                                                                int H, int W, int accum) {
  __shared__ float sw[64];
  __shared__ float sdw[64];
  if (threadIdx.x < 64) { sw[threadIdx.x] = w[threadIdx.x]; sdw[threadIdx.x] = 0.f; }
  __syncthreads();
  float ldw[64];
#pragma unroll
  for (int q = 0; q < 64; ++q) ldw[q] = 0.f;
  const long total = (long)N * H * W;
  for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
    const int x = (int)(o % W), y = (int)((o / W) % H), n = (int)(o / W / H);
    const float2 pv = *reinterpret_cast<const float2*>(pf + o * 2);
    float r0 = 0.f, r1 = 0.f;
#pragma unroll
    for (int ky = 0; ky < 4; ++ky) {
      const int gy = 2 * y + ky - 1;
      if (gy < 0 || gy >= 2 * H) continue;
#pragma unroll
      for (int kx = 0; kx < 4; ++kx) {
        const int gx = 2 * x + kx - 1;
        if (gx < 0 || gx >= 2 * W) continue;
        const float* gp = g + (((size_t)n * 2 * H + gy) * 2 * W + gx) * g_cs + g_c0;
        const float g0 = gp[0], g1 = gp[1];
        const float* ww = sw + (ky * 4 + kx) * 4;  // [o][i]
        r0 += g0 * ww[0] + g1 * ww[2];
        r1 += g0 * ww[1] + g1 * ww[3];
        float* d = ldw + (ky * 4 + kx) * 4;
        d[0] += g0 * pv.x; d[1] += g0 * pv.y; d[2] += g1 * pv.x; d[3] += g1 * pv.y;
      }
    }
    float2* dp = reinterpret_cast<float2*>(dpf + o * 2);
    if (accum) { const float2 e = *dp; r0 += e.x; r1 += e.y; }
    *dp = make_float2(r0, r1);
  }
#pragma unroll
  for (int q = 0; q < 64; ++q) {
    const float s = wave_sum_t(ldw[q]);
    if ((threadIdx.x & 63) == 0) atomicAdd(&sdw[q], s);
  }
  __syncthreads();
  if (threadIdx.x < 64) atomicAdd(dw + threadIdx.x, sdw[threadIdx.x]);
}

// Flow-head filter gradient (3x3, stride 1, pad 1, Cout = 2):
//   dw[co][tap*cin_pad + ci] += sum_{iy,ix} x[iy][ix][ci] * g[iy - ky + 1][ix - kx + 1][co]
// x is read ONCE: a thread owns 4 channels (one float4 per pixel) and all 9 taps x 2 outputs = 72 accumulators;
// the 18 g values of a pixel are wave-uniform per 16-lane group (L1 broadcast).  Block = 64 channels x a pixel
// range; lane = (16 channel groups) x (4 pixel lanes), 4 waves stride the pixels further.  HBM-bound on x.
__global__ void __launch_bounds__(256) head_bwd_filter_kernel(const float* __restrict__ x, int x_cs, int x_c0, int cin,
                                                              const float* __restrict__ g, float* __restrict__ dw,
                                                              int cin_pad, int kpad, int N, int H, int W) {
  const int cg = threadIdx.x & 15, pl = threadIdx.x >> 4;  // 16 pixel lanes per block
  const int ci = blockIdx.x * 64 + cg * 4;
  const long npix = (long)N * H * W;
  const long per = (npix + gridDim.y - 1) / gridDim.y;
  const long p0 = (long)blockIdx.y * per, p1 = min(npix, p0 + per);
  float acc[9][2][4];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[t][o][q] = 0.f;
  if (ci < cin)
    for (long pix = p0 + pl; pix < p1; pix += 16) {
      const int ix = (int)(pix % W), iy = (int)((pix / W) % H);
      const float4 xv = *reinterpret_cast<const float4*>(x + pix * x_cs + x_c0 + ci);
      const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int oy = iy - ky + 1;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int ox = ix - kx + 1;
          float2 gv = make_float2(0.f, 0.f);
          if (oy >= 0 && oy < H && ox >= 0 && ox < W)
            gv = *reinterpret_cast<const float2*>(g + (pix + (long)(1 - ky) * W + (1 - kx)) * 2);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            acc[ky * 3 + kx][0][q] += xs[q] * gv.x;
            acc[ky * 3 + kx][1][q] += xs[q] * gv.y;
          }
        }
      }
    }
  // reduce the 4 pixel lanes of a wave by shuffles, the 4 waves through LDS
  __shared__ float s[4][16][72];
  const int wave = threadIdx.x >> 6;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float v = acc[t][o][q];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if ((threadIdx.x & 63) < 16) s[wave][cg][(t * 2 + o) * 4 + q] = v;
      }
  __syncthreads();
  for (int e = threadIdx.x; e < 16 * 72; e += 256) {
    const int g16 = e / 72, r = e - g16 * 72, t = r >> 3, o = (r >> 2) & 1, q = r & 3;
    const int c = blockIdx.x * 64 + g16 * 4 + q;
    if (c < cin) atomicAdd(dw + (size_t)o * kpad + (size_t)t * cin_pad + c, s[0][g16][r] + s[1][g16][r] + s[2][g16][r] + s[3][g16][r]);
  }
}

// Flow-head input gradient: dx[pix][ci] += sum_{tap,co<2} g[pix - (tap - 1)][co] * w[co][tap*cin_pad + ci]
// (3x3, stride 1, pad 1; the transpose of the head).  One thread per (pixel, 4 channels); reads the head's own
// packed weight, so no transposed copy exists.  HBM-bound on the read-modify-write of dx.
__global__ void __launch_bounds__(256) head_bwd_data_kernel(const float* __restrict__ g, const float* __restrict__ w,
                                                            float* __restrict__ dx, int dx_cs, int dx_c0, int cin,
                                                            int cin_pad, int kpad, int N, int H, int W) {
  const int c4 = (cin + 3) >> 2;
  const long total = (long)N * H * W * c4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long pix = i / c4;
    const int ci = (int)(i - pix * c4) * 4;
    const int x = (int)(pix % W), y = (int)((pix / W) % H);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int gy = y + 1 - ky;
      if (gy < 0 || gy >= H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int gx = x + 1 - kx;
        if (gx < 0 || gx >= W) continue;
        const float2 gv = *reinterpret_cast<const float2*>(g + (pix + (long)(1 - ky) * W + (1 - kx)) * 2);
        const float* w0 = w + (size_t)(ky * 3 + kx) * cin_pad + ci;
        const float* w1 = w0 + kpad;
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] += gv.x * w0[q] + gv.y * w1[q];
      }
    }
    float* d = dx + pix * dx_cs + dx_c0 + ci;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (ci + q < cin) d[q] += acc[q];
  }
}

// ---------------------------------------------------------------------------
// Filter gradient on the fp32 matrix cores.
//   D[i][j] (tap) = sum_pix  Dn[pix][i] * Sm[tap-shifted pix][j]
// Dn ("dense") is walked pixel by pixel; Sm ("sampled") is read at (y*s + ky - pad, x*s + kx - pad), zero
// outside.  For a convolution Dn = dY, Sm = X (D = dW[co][ci]); for a transposed convolution Dn = X, Sm = dY
// with s = 2, pad = 1 (D = dWt[ci][co]).  Block = 128 (i) x 128 (j) x one tap x one pixel range;
// 4 waves x (64 x 64) = 2 x 2 v_mfma_f32_32x32x2_f32 tiles; the reduction index of the MFMA is the pixel.
// Operands go L2 -> LDS by buffer LDS-DMA as [32 pixels][128 channels] fp32 tiles (512-byte rows, read back
// with ds_read_b32: lane = channel, conflict-free without swizzle); results are added to the packed weight
// gradient with fp32 atomics (summation order across pixel ranges is not deterministic, like cuDNN's default).
// ---------------------------------------------------------------------------
struct BwdwArgs {
  const float* dn; const float* sm; float* dw;
  int N, DH, DW_, dn_cs, dn_c0, Ci;   // dense tensor: [N, DH, DW] pixels, Ci channels of interest
  int SH, SW, sm_cs, sm_c0, Cj;       // sampled tensor
  int KH, KW, stride, pad;
  int dn_bytes, sm_bytes;
  int P;                               // N*DH*DW
  int pix_per_split;                   // multiple of 32
  long stride_i, stride_j;             // element strides of D[i][j] inside dw
  int perm_i, perm_j;                  // apply the 32-row permutation of the LDS-DMA weight layout to this index
  int tap_base[49];                    // element offset of the tap inside dw
};

__device__ __forceinline__ int perm32(int r) {  // packed row of output channel r (fn2_conv_plan.layout == 1)
  const int g = r & ~31, q = r & 31, h = q >> 4, rr = q & 15;
  return g + (rr & 3) + 8 * (rr >> 2) + 4 * h;
}

// SWAP: the MFMA computes D^T (lane = i, registers = j) so that the atomics of a wave instruction are
// contiguous when i is the fastest index of dw (the transposed convolutions, stride_i == 1).
// NI x NJ = 32x32 MFMA tiles per wave (block tile 64*NI x 64*NJ): layers with <= 64 channels on a side (the
// stems, conv2, deconv2) would waste half or three quarters of a 128-wide tile's matrix work.
template <bool SWAP, int NI, int NJ>
__global__ void __launch_bounds__(256) bwd_filter_kernel(const BwdwArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int PK = 32;  // pixels per stage
  // [dense | sampled][pixel][channel] x 2 stages; separate LDS objects so the waitcnt pass lets the ds_reads of one
  // stage run while the LDS-DMA of the next is in flight (see conv2.hip)
  __shared__ float lds0[2][PK * 128];
  __shared__ float lds1[2][PK * 128];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wave >> 1, wj = wave & 1;
  constexpr int TI = 64 * NI, TJ = 64 * NJ;
  const int i0 = blockIdx.x * TI;
  const int njt = (p.Cj + TJ - 1) / TJ;
  const int tap = blockIdx.y / njt, j0 = (blockIdx.y - tap * njt) * TJ;
  const int ky = tap / p.KW, kx = tap - ky * p.KW;
  const int pbeg = blockIdx.z * p.pix_per_split, pend = min(p.P, pbeg + p.pix_per_split);
  if (pbeg >= pend) return;
  const int nstage = (pend - pbeg + PK - 1) / PK;

  const auto rs_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dn), 0, p.dn_bytes, 0x00020000);
  const auto rs_s = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.sm), 0, p.sm_bytes, 0x00020000);

  // DMA pieces: one wave instruction = 2 pixel rows x 512 B.  Per stage 16 pieces per tensor, 4 per wave each.
  // lane -> (row lane>>5 of the piece, 16-byte chunk lane&31 = channels 4*(lane&31)..+3)
  const int lrow = lane >> 5, lch = (lane & 31) * 4;
  const bool ch_ok_d = lch < TI && i0 + lch < p.Ci, ch_ok_s = lch < TJ && j0 + lch < p.Cj;
  // sampled-tensor pixel state of this lane's 4 rows: r = (wave*4 + k)*2 + lrow within the stage
  int sn[4], sy[4], sx[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int pix = pbeg + (wave * 4 + k) * 2 + lrow;
    const int n = pix / (p.DH * p.DW_), rem = pix - n * (p.DH * p.DW_);
    sn[k] = n; sy[k] = rem / p.DW_; sx[k] = rem - sy[k] * p.DW_;
  }
  auto issue = [&](int st, float (*lds)[PK * 128]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int r = (wave * 4 + k) * 2;          // first of the 2 rows of this piece (wave-uniform)
      const int pix = pbeg + st * PK + r + lrow;  // this lane's pixel
      const bool pv = pix < pend;
      const unsigned vd = (pv && ch_ok_d) ? (unsigned)((pix * p.dn_cs + p.dn_c0 + i0 + lch) * 4) : kOobT;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_d, (lptr_t)&lds[0][r * 128], 16, vd, 0, 0, 0);
      const int iy = sy[k] * p.stride + ky - p.pad, ix = sx[k] * p.stride + kx - p.pad;
      const bool sv = pv && ch_ok_s && iy >= 0 && iy < p.SH && ix >= 0 && ix < p.SW;
      const unsigned vs = sv ? (unsigned)((((sn[k] * p.SH + iy) * p.SW + ix) * p.sm_cs + p.sm_c0 + j0 + lch) * 4) : kOobT;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_s, (lptr_t)&lds[1][r * 128], 16, vs, 0, 0, 0);
      // advance this row by PK pixels for the next stage
      sx[k] += PK;
      while (sx[k] >= p.DW_) { sx[k] -= p.DW_; if (++sy[k] == p.DH) { sy[k] = 0; ++sn[k]; } }
    }
  };

  f32x16 acc[NI][NJ];
#pragma unroll
  for (int a = 0; a < NI; ++a)
#pragma unroll
    for (int b = 0; b < NJ; ++b)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.f;

  const int fr = lane & 31, fk = lane >> 5;
  auto compute = [&](const float (*lds)[PK * 128]) {
    const float* Dn = &lds[0][wi * NI * 32 + fr];
    const float* Sm = &lds[1][wj * NJ * 32 + fr];
#pragma unroll
    for (int kk = 0; kk < PK; kk += 2) {
      const int row = (kk + fk) * 128;
      float av[NI], bv[NJ];
#pragma unroll
      for (int t = 0; t < NI; ++t) av[t] = Dn[row + 32 * t];
#pragma unroll
      for (int t = 0; t < NJ; ++t) bv[t] = Sm[row + 32 * t];
#pragma unroll
      for (int ti = 0; ti < NI; ++ti)
#pragma unroll
        for (int tj = 0; tj < NJ; ++tj) {
          if constexpr (SWAP) acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[tj], av[ti], acc[ti][tj], 0, 0, 0);
          else acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ti], bv[tj], acc[ti][tj], 0, 0, 0);
        }
    }
  };
  issue(0, lds0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // two stages per trip, no branch around the MFMAs: a stage past the pixel range loads zeros (pix >= pend)
  const int nstage2 = (nstage + 1) & ~1;
  for (int st = 0; st < nstage2; st += 2) {
    issue(st + 1, lds1);
    compute(lds0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (st + 2 < nstage2) issue(st + 2, lds0);
    compute(lds1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  // D layout (32x32): col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5); (row, col) = (i, j), or (j, i) if SWAP
  float* base = p.dw + p.tap_base[tap];
#pragma unroll
  for (int ti = 0; ti < NI; ++ti)
#pragma unroll
    for (int tj = 0; tj < NJ; ++tj) {
      if constexpr (SWAP) {
        const int i = i0 + wi * NI * 32 + ti * 32 + fr;
        if (i >= p.Ci) continue;
        const long oi = (long)(p.perm_i ? perm32(i) : i) * p.stride_i;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int j = j0 + wj * NJ * 32 + tj * 32 + (q & 3) + 8 * (q >> 2) + 4 * fk;
          if (j < p.Cj) atomicAdd(base + (long)(p.perm_j ? perm32(j) : j) * p.stride_j + oi, acc[ti][tj][q]);
        }
      } else {
        const int j = j0 + wj * NJ * 32 + tj * 32 + fr;
        if (j >= p.Cj) continue;
        const long oj = (long)(p.perm_j ? perm32(j) : j) * p.stride_j;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int i = i0 + wi * NI * 32 + ti * 32 + (q & 3) + 8 * (q >> 2) + 4 * fk;
          if (i < p.Ci) atomicAdd(base + (long)(p.perm_i ? perm32(i) : i) * p.stride_i + oj, acc[ti][tj][q]);
        }
      }
    }
#endif
}

}  // namespace fn2

using namespace fn2;

extern "C" {

int fn2_epe_loss_grad(const float* pred, const float* label, float* dpred, float* loss_accum, int n, int h, int w,
                      float weight, void* stream) {
  FN2_REQUIRE(pred && label && dpred && loss_accum, "epe_loss_grad: null pointer");
  FN2_REQUIRE(n >= 1 && h >= 1 && w >= 1, "epe_loss_grad: bad dims");
  const long npix = (long)n * h * w;
  hipLaunchKernelGGL(epe_loss_grad_kernel, dim3(grid_for(npix, 256)), dim3(256), 0, (hipStream_t)stream, pred, label,
                     dpred, loss_accum, npix, weight / (float)n);
  FN2_CHECK_LAUNCH("epe_loss_grad");
  return FN2_OK;
}

static int gpb_log2_for(int c4) {
  int l = 0;
  while ((1 << l) < c4 && l < 8) ++l;
  return l;
}

int fn2_leaky_bwd(const fn2_tensor* y, const fn2_tensor* g, float* db, void* stream) {
  FN2_REQUIRE(y && g && y->data && g->data, "leaky_bwd: null tensor");
  FN2_REQUIRE(y->dtype == FN2_F32 && g->dtype == FN2_F32, "leaky_bwd: fp32 only");
  FN2_REQUIRE(y->n == g->n && y->h == g->h && y->w == g->w && y->c == g->c, "leaky_bwd: shape mismatch");
  FN2_REQUIRE(y->c % 4 == 0 && y->cs % 4 == 0 && y->c0 % 4 == 0 && g->cs % 4 == 0 && g->c0 % 4 == 0,
              "leaky_bwd: channel slice must be 4-aligned");
  const long npix = (long)y->n * y->h * y->w;
  const int l2 = gpb_log2_for(y->c / 4);
  // few, fat blocks: every block ends with one atomic per channel on the SAME db[c] addresses, and 2048 blocks
  // serialised on them cost more than the pass itself (conv1: 0.22 ms for a 300 MB pass)
  long splits = (npix + 255) / 256;
  if (splits > 512) splits = 512;
  hipLaunchKernelGGL(act_bias_bwd_kernel<true>, dim3((y->c / 4 + (1 << l2) - 1) >> l2, (int)splits), dim3(256), 0,
                     (hipStream_t)stream, (const float*)y->data, (float*)g->data, db, npix, y->c, y->cs, y->c0, g->cs,
                     g->c0, l2);
  FN2_CHECK_LAUNCH("leaky_bwd");
  return FN2_OK;
}

int fn2_bias_grad(const fn2_tensor* g, float* db, void* stream) {
  FN2_REQUIRE(g && g->data && db, "bias_grad: null pointer");
  FN2_REQUIRE(g->dtype == FN2_F32, "bias_grad: fp32 only");
  const long npix = (long)g->n * g->h * g->w;
  hipStream_t st = (hipStream_t)stream;
  if (g->c == 2 && g->cs == 2 && g->c0 == 0) {
    hipLaunchKernelGGL(bias_grad2_kernel, dim3(grid_for(npix, 1024)), dim3(256), 0, st, (const float*)g->data, db, npix);
  } else if (g->c % 4 == 0 && g->cs % 4 == 0 && g->c0 % 4 == 0) {
    const int l2 = gpb_log2_for(g->c / 4);
    long splits = (npix + 255) / 256;
    if (splits > 512) splits = 512;
    hipLaunchKernelGGL(act_bias_bwd_kernel<false>, dim3((g->c / 4 + (1 << l2) - 1) >> l2, (int)splits), dim3(256), 0, st,
                       (const float*)nullptr, (float*)g->data, db, npix, g->c, 0, 0, g->cs, g->c0, l2);
  } else {
    int splits = (int)((npix + 4095) / 4096);
    if (splits > 256) splits = 256;
    hipLaunchKernelGGL(bias_grad_kernel, dim3((g->c + 63) / 64, splits), dim3(256), 0, st, (const float*)g->data, db,
                       npix, g->c, g->cs, g->c0);
  }
  FN2_CHECK_LAUNCH("bias_grad");
  return FN2_OK;
}

int fn2_gather_f32(float* dst, const float* src, const int32_t* map, int64_t n, void* stream) {
  FN2_REQUIRE(dst && src && map && n >= 0, "gather: bad arguments");
  if (n == 0) return FN2_OK;
  hipLaunchKernelGGL(gather_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, dst, src, map, (long)n);
  FN2_CHECK_LAUNCH("gather");
  return FN2_OK;
}

int fn2_adam_step(float* w, float* m, float* v, const float* g, int64_t n, float lr, float beta1, float beta2,
                  float eps, int step, float l2, float grad_scale, void* stream) {
  FN2_REQUIRE(w && m && v && g && n >= 0 && step >= 1, "adam_step: bad arguments");
  if (n == 0) return FN2_OK;
  const float lr_t = lr * sqrtf(1.f - powf(beta2, (float)step)) / (1.f - powf(beta1, (float)step));
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, w, m, v, g, (long)n, lr_t,
                     beta1, beta2, eps, l2, grad_scale);
  FN2_CHECK_LAUNCH("adam");
  return FN2_OK;
}

int fn2_upsample_flow_bwd(const fn2_tensor* g, const float* pf, const float* w, float* dpf, float* dw, int accumulate,
                          void* stream) {
  FN2_REQUIRE(g && g->data && pf && w && dpf && dw, "upsample_flow_bwd: null pointer");
  FN2_REQUIRE(g->dtype == FN2_F32 && g->c == 2 && g->h % 2 == 0 && g->w % 2 == 0, "upsample_flow_bwd: g must be a 2-channel fp32 view of even size");
  const int H = g->h / 2, W = g->w / 2;
  hipLaunchKernelGGL(upsample_flow_bwd_kernel, dim3(grid_for((long)g->n * H * W, 256)), dim3(256), 0,
                     (hipStream_t)stream, (const float*)g->data, g->cs, g->c0, pf, w, dpf, dw, g->n, H, W, accumulate);
  FN2_CHECK_LAUNCH("upsample_flow_bwd");
  return FN2_OK;
}

int fn2_head_bwd_filter(const fn2_tensor* x, const float* g, float* dw, int cin_pad, int kpad, void* stream) {
  FN2_REQUIRE(x && x->data && g && dw, "head_bwd_filter: null pointer");
  FN2_REQUIRE(x->dtype == FN2_F32 && cin_pad >= x->c && kpad >= 9 * cin_pad, "head_bwd_filter: bad layout");
  FN2_REQUIRE(x->cs % 4 == 0 && x->c0 % 4 == 0 && (x->c + 3) / 4 * 4 <= x->cs - x->c0, "head_bwd_filter: x view must be 16-byte aligned and padded to 4 channels");
  const long npix = (long)x->n * x->h * x->w;
  int splits = (int)((npix + 511) / 512);
  if (splits > 512) splits = 512;
  hipLaunchKernelGGL(head_bwd_filter_kernel, dim3((x->c + 63) / 64, splits), dim3(256), 0, (hipStream_t)stream,
                     (const float*)x->data, x->cs, x->c0, x->c, g, dw, cin_pad, kpad, x->n, x->h, x->w);
  FN2_CHECK_LAUNCH("head_bwd_filter");
  return FN2_OK;
}

int fn2_head_bwd_data(const float* g, const float* w, const fn2_tensor* dx, int cin_pad, int kpad, void* stream) {
  FN2_REQUIRE(g && w && dx && dx->data, "head_bwd_data: null pointer");
  FN2_REQUIRE(dx->dtype == FN2_F32 && cin_pad >= dx->c && cin_pad % 4 == 0 && kpad >= 9 * cin_pad, "head_bwd_data: bad layout");
  const long total = (long)dx->n * dx->h * dx->w * ((dx->c + 3) / 4);
  hipLaunchKernelGGL(head_bwd_data_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, g, w,
                     (float*)dx->data, dx->cs, dx->c0, dx->c, cin_pad, kpad, dx->n, dx->h, dx->w);
  FN2_CHECK_LAUNCH("head_bwd_data");
  return FN2_OK;
}

int fn2_conv2d_bwd_filter(const fn2_bwdw_desc* d, void* stream) {
  FN2_REQUIRE(d && d->x.data && d->dy.data && d->dw, "bwd_filter: null pointer");
  FN2_REQUIRE(d->x.dtype == FN2_F32 && d->dy.dtype == FN2_F32, "bwd_filter: fp32 only");
  FN2_REQUIRE(d->kind >= 0 && d->kind <= 2, "bwd_filter: kind 0 (conv), 1 (deconv k4 s2 crop 1) or 2 (stem row-run conv)");
  FN2_REQUIRE(d->x.n == d->dy.n, "bwd_filter: batch mismatch");
  FN2_REQUIRE(d->cin_pad % 8 == 0 && d->cout_pad >= d->dy.c && d->kpad > 0, "bwd_filter: bad packed sizes");
  FN2_REQUIRE((d->x.cs % 4) == 0 && (d->x.c0 % 4) == 0 && (d->dy.cs % 4) == 0 && (d->dy.c0 % 4) == 0,
              "bwd_filter: views must be 16-byte aligned");
  BwdwArgs a;
  const fn2_tensor* dn;
  const fn2_tensor* sm;
  if (d->kind == 0 || d->kind == 2) {
    FN2_REQUIRE(d->kh >= 1 && d->kh * d->kw <= 49 && d->stride >= 1 && d->pad >= 0, "bwd_filter: bad geometry");
    if (d->kind == 2) FN2_REQUIRE(d->pad == 0 && d->x.c0 == 0 && d->cin_pad >= d->kw * d->x.cs, "bwd_filter: stem layout");
    FN2_REQUIRE(d->dy.h == (d->x.h + 2 * d->pad - d->kh) / d->stride + 1 && d->dy.w == (d->x.w + 2 * d->pad - d->kw) / d->stride + 1,
                "bwd_filter: dy spatial size does not match the convolution");
    dn = &d->dy; sm = &d->x;   // D[i = co][j = ci]
    a.KH = d->kh; a.KW = d->kw; a.stride = d->stride; a.pad = d->pad;
    a.stride_i = d->kpad; a.stride_j = 1; a.perm_i = d->wgt_layout == 1; a.perm_j = 0;
    if (d->kind == 2) {
      // row-run stem: the kw taps x cs channels of a kernel row are ONE contiguous run of the pre-padded input,
      // so a kernel row is a single "tap" of kw*cs sampled channels (same trick as the forward kind-2 kernel)
      a.KW = 1;
      for (int t = 0; t < d->kh; ++t) a.tap_base[t] = t * d->cin_pad;
    } else {
      for (int t = 0; t < d->kh * d->kw; ++t) a.tap_base[t] = t * d->cin_pad;
    }
  } else {
    FN2_REQUIRE(d->kh == 4 && d->kw == 4 && d->stride == 2, "bwd_filter: deconv is k4 s2 crop 1");
    FN2_REQUIRE(d->dy.h == 2 * d->x.h && d->dy.w == 2 * d->x.w, "bwd_filter: deconv dy must be 2H x 2W");
    dn = &d->x; sm = &d->dy;   // D[i = ci][j = co]; dWt[ky,kx,co,ci] = sum x[y,x,ci] * dy[2y+ky-1, 2x+kx-1, co]
    a.KH = 4; a.KW = 4; a.stride = 2; a.pad = 1;
    a.stride_i = 1; a.stride_j = d->kpad; a.perm_i = 0; a.perm_j = d->wgt_layout == 1;
    for (int ky = 0; ky < 4; ++ky)
      for (int kx = 0; kx < 4; ++kx) {
        const int pa = (3 - ky) & 1, pb = (3 - kx) & 1, ty = (3 - pa - ky) / 2, tx = (3 - pb - kx) / 2;
        a.tap_base[ky * 4 + kx] = (pa * 2 + pb) * d->cout_pad * d->kpad + (ty * 2 + tx) * d->cin_pad;
      }
  }
  a.dn = (const float*)dn->data; a.sm = (const float*)sm->data; a.dw = d->dw;
  a.N = dn->n; a.DH = dn->h; a.DW_ = dn->w; a.dn_cs = dn->cs; a.dn_c0 = dn->c0; a.Ci = dn->c;
  a.SH = sm->h; a.SW = sm->w; a.sm_cs = sm->cs; a.sm_c0 = sm->c0; a.Cj = d->kind == 2 ? d->kw * sm->cs : sm->c;
  const long dnb = (long)dn->n * dn->h * dn->w * dn->cs * 4, smb = (long)sm->n * sm->h * sm->w * sm->cs * 4;
  FN2_REQUIRE(dnb < (1L << 31) && smb < (1L << 31), "bwd_filter: tensors >= 2 GiB are not addressable");
  a.dn_bytes = (int)dnb; a.sm_bytes = (int)smb;
  a.P = dn->n * dn->h * dn->w;
  const int taps = a.KH * a.KW;
  const int ni = a.Ci <= 64 ? 1 : 2, nj = a.Cj <= 64 ? 1 : 2;
  const int it = (a.Ci + 64 * ni - 1) / (64 * ni), jt = (a.Cj + 64 * nj - 1) / (64 * nj);
  // pixel splits: aim at >= ~1024 blocks, ranges multiples of 32 pixels
  long blocks = (long)it * jt * taps;
  int splits = (int)((1024 + blocks - 1) / blocks);
  const int max_splits = (a.P + 255) / 256;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  a.pix_per_split = (((a.P + splits - 1) / splits) + 31) / 32 * 32;
  splits = (a.P + a.pix_per_split - 1) / a.pix_per_split;
  const dim3 grid(it, jt * taps, splits), block(256);
  hipStream_t st = (hipStream_t)stream;
#define FN2_BWF(SW_, NI_, NJ_) hipLaunchKernelGGL((bwd_filter_kernel<SW_, NI_, NJ_>), grid, block, 0, st, a)
  if (d->kind == 1) {
    if (ni == 2 && nj == 2) FN2_BWF(true, 2, 2); else if (ni == 2) FN2_BWF(true, 2, 1);
    else if (nj == 2) FN2_BWF(true, 1, 2); else FN2_BWF(true, 1, 1);
  } else {
    if (ni == 2 && nj == 2) FN2_BWF(false, 2, 2); else if (ni == 2) FN2_BWF(false, 2, 1);
    else if (nj == 2) FN2_BWF(false, 1, 2); else FN2_BWF(false, 1, 1);
  }
#undef FN2_BWF
  FN2_CHECK_LAUNCH("bwd_filter");
  return FN2_OK;
}

}  // extern "C"
