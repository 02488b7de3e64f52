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
// timing ablations of bwd_filter_x2_kernel (FN2_BWF_DBG: 1 no sampled-operand DMA, 2 no dense DMA, 4 no LDS reads / MFMAs,
// 8 no atomics): compiled in by tools/build_variant.sh abl "-DFN2_BWF_ABLATE=1" only
#ifndef FN2_BWF_ABLATE
#define FN2_BWF_ABLATE 0
#endif

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
// pixw != nullptr: per-pixel weights of the hard-flow-example-mining losses (utils.py:227-339): the 0 / (1 + lambda) *
// #pixels / #hard mask of 'hard', or 1 + lambda * edges of 'edges';  L = weight / N * sum w_pix ||pred - label||.
__global__ void __launch_bounds__(256) epe_loss_grad_kernel(const float* __restrict__ pred,
                                                            const float* __restrict__ label,
                                                            const float* __restrict__ pixw, float* __restrict__ dpred,
                                                            float* __restrict__ loss_accum, long npix, float scale,
                                                            float grad_mult) {
  float part = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
    const float2 p = *reinterpret_cast<const float2*>(pred + 2 * i);
    const float2 l = *reinterpret_cast<const float2*>(label + 2 * i);
    const float du = p.x - l.x, dv = p.y - l.y;
    const float wp = pixw != nullptr ? pixw[i] : 1.f;
    const float e = sqrtf(du * du + dv * dv);
    part += wp * e;
    const float inv = e > 0.f ? wp * scale * grad_mult / e : 0.f;
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

// Split-fp16 form of the same pass (gradient and activation buffers of the f16x2 trainer): a thread owns one group
// of 8 channels of a pixel (16 B of hi parts + 16 B of lo parts per tensor).
template <bool ACT>
__global__ void __launch_bounds__(256) act_bias_bwd_x2_kernel(const x2_t* __restrict__ y, x2_t* __restrict__ g,
                                                              float* __restrict__ db, long npix, int c, int y_cs,
                                                              int y_c0, int g_cs, int g_c0, int gpb_log2) {
  const int gpb = 1 << gpb_log2, rows = 256 >> gpb_log2;
  const int tc = threadIdx.x & (gpb - 1), tr = threadIdx.x >> gpb_log2;
  const int ch = (blockIdx.x * gpb + tc) * 8;
  const long per = (npix + gridDim.y - 1) / gridDim.y;
  const long p0 = (long)blockIdx.y * per, p1 = min(npix, p0 + per);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (ch < c) {
    auto one = [&](uint4* gp, const uint4 (&gq)[2], const uint4 (&yq)[2]) {
      float gv[8];
      join8(gq[0], gq[1], gv);
      if constexpr (ACT) {
        float yv[8];
        join8(yq[0], yq[1], yv);
#pragma unroll
        for (int j = 0; j < 8; ++j) gv[j] *= yv[j] > 0.f ? 1.f : (yv[j] < 0.f ? 0.1f : 0.55f);
        split8(gv, gp[0], gp[1]);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += gv[j];
    };
    long pix = p0 + tr;
    for (; pix + 3L * rows < p1; pix += 4L * rows) {  // four pixels per trip: 16 x 16-byte loads in flight per lane
      uint4 gq[4][2], yq[4][2];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint4* gp = reinterpret_cast<const uint4*>(g + (pix + (long)u * rows) * g_cs + g_c0 + ch);
        gq[u][0] = gp[0]; gq[u][1] = gp[1];
        if constexpr (ACT) {
          const uint4* yp = reinterpret_cast<const uint4*>(y + (pix + (long)u * rows) * y_cs + y_c0 + ch);
          yq[u][0] = yp[0]; yq[u][1] = yp[1];
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        one(reinterpret_cast<uint4*>(g + (pix + (long)u * rows) * g_cs + g_c0 + ch), gq[u], yq[u]);
    }
    for (; pix < p1; pix += rows) {
      uint4* gp = reinterpret_cast<uint4*>(g + pix * g_cs + g_c0 + ch);
      uint4 gq[2] = {gp[0], gp[1]}, yq[2] = {gq[0], gq[1]};
      if constexpr (ACT) {
        const uint4* yp = reinterpret_cast<const uint4*>(y + pix * y_cs + y_c0 + ch);
        yq[0] = yp[0]; yq[1] = yp[1];
      }
      one(gp, gq, yq);
    }
  }
  if (db == nullptr) return;
  __shared__ float s[256][9];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[threadIdx.x][j] = acc[j];
  __syncthreads();
  if (tr == 0 && ch < c) {
    for (int r = 1; r < rows; ++r)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += s[r * gpb + tc][j];
#pragma unroll
    for (int j = 0; j < 8; ++j) atomicAdd(db + ch + j, acc[j]);
  }
}

// fp32 -> split fp16 with a power-of-two scale (the forward / backward weight copies of the f16x2 trainer are
// derived from the fp32 master every step); map != nullptr: gathered, dst[i] = map[i] >= 0 ? src[map[i]] : 0.
// Where the (hi, lo) 16-byte chunks of 8-element group g of a packed [rows][K] weight (K % 32 == 0) live in the split-fp16
// image: row-major (fn2_conv2d wgt_layout 1: hi at chunk 2 g, lo right behind it) or MFMA-fragment order (wgt_layout 2,
// conv2.hip WREG: per 32-row tile and 128-byte stage four 1 KiB blocks f = 2 q + part, lane 32 h + r of a block holds
// chunk 4 q + 2 h + part of row r; the lo chunk is the same lane of the next block).  Returns 16-byte chunk indices.
__device__ __forceinline__ void x2_chunks(long g, int frag_k, long& hi, long& lo) {
  if (frag_k == 0) { hi = 2 * g; lo = 2 * g + 1; return; }
  const int gpr = frag_k >> 3;                      // groups per row
  const long row = g / gpr;
  const int gk = (int)(g - row * gpr), st = gk >> 2, gs = gk & 3;
  hi = (((row >> 5) * (frag_k >> 5) + st) * 4 + (gs >> 1) * 2) * 64 + (gs & 1) * 32 + (row & 31);
  lo = hi + 64;
}

__global__ void __launch_bounds__(256) to_x2_kernel(const float* __restrict__ src, const int* __restrict__ map,
                                                    x2_t* __restrict__ dst, long ngroups, float scale, int frag_k) {
  for (long gi = (long)blockIdx.x * blockDim.x + threadIdx.x; gi < ngroups; gi += (long)gridDim.x * blockDim.x) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (map != nullptr) { const int m = map[gi * 8 + j]; v[j] = m >= 0 ? src[m] * scale : 0.f; }
      else v[j] = src[gi * 8 + j] * scale;
    }
    long hi, lo;
    x2_chunks(gi, frag_k, hi, lo);
    uint4* q = reinterpret_cast<uint4*>(dst);
    split8(v, q[hi], q[lo]);
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
// dst += src (fp32, n % 4 == 0 elements as 16-byte accesses; the tail scalar)
__global__ void __launch_bounds__(256) add_inplace_kernel(float* __restrict__ dst, const float* __restrict__ src, long n) {
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    float4 d = reinterpret_cast<float4*>(dst)[i];
    const float4 a = reinterpret_cast<const float4*>(src)[i];
    d.x += a.x; d.y += a.y; d.z += a.z; d.w += a.w;
    reinterpret_cast<float4*>(dst)[i] = d;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) dst[(n4 << 2) + threadIdx.x] += src[(n4 << 2) + threadIdx.x];
}

// dst[pix][0..c) = src[pix][c0..c0+c) of an fp32 NHWC buffer with channel stride cs (dense destination)
__global__ void __launch_bounds__(256) slice_copy_kernel(const float* __restrict__ src, float* __restrict__ dst, long npix,
                                                         int c, int cs, int c0) {
  const long total = npix * c;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long pix = i / c;
    const int ch = (int)(i - pix * c);
    dst[i] = src[pix * cs + c0 + ch];
  }
}

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

// All parameter tensors in one launch: blockIdx.y = tensor, blocks of a row stride its elements four at a time
// (16-byte accesses; every tensor starts 16-byte aligned).  tab[t] = {w, m, v, g} pointers + the split-fp16 forward
// copy of the weight (or null) and its power-of-two scale: the copy the convolutions read is written by the same pass
// that updates the fp32 master.  n[t] elements, l2[t] regulariser (0 for biases / transposed convs).
struct AdamTensor { float* w; float* m; float* v; const float* g; x2_t* wx2; float scale; int frag_k; };  // frag_k: see x2_chunks
static_assert(sizeof(AdamTensor) == 48, "the host builds the table as six 64-bit words per tensor");
__device__ __forceinline__ void adam_tensor(const AdamTensor& t, long cnt, float reg, float lr_t, float b1, float b2,
                                            float eps, float gscale) {
  auto one = [&](float wi, float gi, float& mi, float& vi) {
    gi = gi * gscale + reg * wi;
    mi = b1 * mi + (1.f - b1) * gi;
    vi = b2 * vi + (1.f - b2) * gi * gi;
    return wi - lr_t * mi / (sqrtf(vi) + eps);
  };
  const long n4 = cnt >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    float4 w = reinterpret_cast<float4*>(t.w)[i], m = reinterpret_cast<float4*>(t.m)[i], v = reinterpret_cast<float4*>(t.v)[i];
    const float4 g = reinterpret_cast<const float4*>(t.g)[i];
    w.x = one(w.x, g.x, m.x, v.x); w.y = one(w.y, g.y, m.y, v.y);
    w.z = one(w.z, g.z, m.z, v.z); w.w = one(w.w, g.w, m.w, v.w);
    reinterpret_cast<float4*>(t.m)[i] = m;
    reinterpret_cast<float4*>(t.v)[i] = v;
    reinterpret_cast<float4*>(t.w)[i] = w;
    if (t.wx2 != nullptr) {
      const float o[4] = {w.x * t.scale, w.y * t.scale, w.z * t.scale, w.w * t.scale};
      if (t.frag_k == 0) {
        store_vec<x2_t, 4>(t.wx2 + 4 * i, o);
      } else {  // elements 4 i .. 4 i + 3 = half of group i / 2: 8 bytes of its hi chunk, 8 bytes of its lo chunk
        typedef __attribute__((ext_vector_type(4))) _Float16 h4;
        h4 h, l;
#pragma unroll
        for (int j = 0; j < 4; ++j) { h[j] = (_Float16)o[j]; l[j] = (_Float16)(o[j] - (float)h[j]); }
        long ch, cl;
        x2_chunks(i >> 1, t.frag_k, ch, cl);
        char* base = reinterpret_cast<char*>(t.wx2);
        *reinterpret_cast<h4*>(base + ch * 16 + (i & 1) * 8) = h;
        *reinterpret_cast<h4*>(base + cl * 16 + (i & 1) * 8) = l;
      }
    }
  }
  for (long i = 4 * n4 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += (long)gridDim.x * blockDim.x) {
    float mi = t.m[i], vi = t.v[i];
    const float wi = one(t.w[i], t.g[i], mi, vi);
    t.m[i] = mi; t.v[i] = vi; t.w[i] = wi;
    if (t.wx2 != nullptr) store_elem<x2_t>(t.wx2 + i, wi * t.scale);
  }
}
__global__ void __launch_bounds__(256) adam_multi_kernel(const AdamTensor* __restrict__ tab, const long* __restrict__ n,
                                                         const float* __restrict__ l2, float lr_t, float b1, float b2,
                                                         float eps, float gscale) {
  adam_tensor(tab[blockIdx.y], n[blockIdx.y], l2[blockIdx.y], lr_t, b1, b2, eps, gscale);
}

// The same with the per-step scalars read from device memory {lr_t, beta1, beta2, eps, grad_scale}: the launch
// arguments no longer change from step to step, so the whole train step can be replayed as a hipGraph.
__global__ void __launch_bounds__(256) adam_multi_dev_kernel(const AdamTensor* __restrict__ tab, const long* __restrict__ n,
                                                             const float* __restrict__ l2, const float* __restrict__ hyper) {
  adam_tensor(tab[blockIdx.y], n[blockIdx.y], l2[blockIdx.y], hyper[0], hyper[1], hyper[2], hyper[3], hyper[4]);
}

// upsample_flowXtoY backward (forward: conv-transpose 2->2, k4 s2 crop 1, upsample_flow_kernel in conv.hip):
//   dpf[n,y,x,i] (+)= sum_{ky,kx,o} g[n,2y+ky-1,2x+kx-1,o] * w[ky,kx,o,i];  dw[ky,kx,o,i] += sum g * pf
template <typename T>
__global__ void __launch_bounds__(256) upsample_flow_bwd_kernel(const T* __restrict__ g, int g_cs, int g_c0,
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
        const T* gp = g + (((size_t)n * 2 * H + gy) * 2 * W + gx) * g_cs + g_c0;
        const float g0 = load_elem<T>(gp), g1 = load_elem<T>(gp + 1);
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

// The gradient of a flow head's output (3x3, stride 1, pad 1, Cout = 2) as an 18-"channel" tensor:
//   G18[pix][tap * 2 + o] = g[pix + (1 - ky) W + (1 - kx)][o]   (tap = ky * 3 + kx; 0 where that pixel is outside)
// With it both gradients of the head are ordinary matrix products on the kernels the other layers use:
//   dW[o][tap][ci] = sum_pix G18[pix][tap, o] x[pix][ci]   -> bwd_filter (kind 4: 1x1, rows mapped into the head's layout)
//   dx[pix][ci]   += sum_c18 G18[pix][c18] W[c18][ci]      -> fn2_conv2d, 1x1, accumulate
// Split fp16, channel stride >= 24: a thread writes one 8-channel group (4 taps) of a pixel; groups past channel 23
// are never written (they stay the zeros of the allocation, their weights are zero rows).
__global__ void __launch_bounds__(256) head_g18_kernel(const float* __restrict__ g, x2_t* __restrict__ out, int out_cs,
                                                       int out_c0, int N, int H, int W) {
  const long total = (long)N * H * W * 3;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long pix = i / 3;
    const int grp = (int)(i - pix * 3);
    const int x = (int)(pix % W), y = (int)((pix / W) % H);
    float v[8];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int tap = grp * 4 + t, ky = tap / 3, kx = tap - ky * 3;
      const int oy = y + 1 - ky, ox = x + 1 - kx;
      float2 gv = make_float2(0.f, 0.f);
      if (tap < 9 && oy >= 0 && oy < H && ox >= 0 && ox < W)
        gv = *reinterpret_cast<const float2*>(g + (pix + (long)(1 - ky) * W + (1 - kx)) * 2);
      v[2 * t] = gv.x; v[2 * t + 1] = gv.y;
    }
    uint4* q = reinterpret_cast<uint4*>(out + pix * out_cs + out_c0 + grp * 8);
    split8(v, q[0], q[1]);
  }
}

// Flow-head filter gradient (3x3, stride 1, pad 1, Cout = 2):
//   dw[co][tap*cin_pad + ci] += sum_{iy,ix} x[iy][ix][ci] * g[iy - ky + 1][ix - kx + 1][co]
// x is read ONCE: a thread owns 4 channels (one float4 per pixel) and all 9 taps x 2 outputs = 72 accumulators;
// the 18 g values of a pixel are wave-uniform per 16-lane group (L1 broadcast).  Block = 64 channels x a pixel
// range; lane = (16 channel groups) x (4 pixel lanes), 4 waves stride the pixels further.  HBM-bound on x.
template <typename T>
__global__ void __launch_bounds__(256) head_bwd_filter_kernel(const T* __restrict__ x, int x_cs, int x_c0, int cin,
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
  if (ci < cin) {
    // (more loads in flight were tried both ways: four pixels per trip with their loads up front took 256 VGPRs, one wave
    // per SIMD and twice the time; the next pixel's x loaded before this pixel's FMAs was 10-30 % slower, capping the
    // kernel at 128 VGPRs for a fourth wave per SIMD spilled and was 25 % slower)
    for (long pix = p0 + pl; pix < p1; pix += 16) {
      const int ix = (int)(pix % W), iy = (int)((pix / W) % H);
      float xs[4];
      load4<T>(x + pix * x_cs + x_c0 + ci, xs);
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
template <typename T>
__global__ void __launch_bounds__(256) head_bwd_data_kernel(const float* __restrict__ g, const float* __restrict__ w,
                                                            T* __restrict__ dx, int dx_cs, int dx_c0, int cin,
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
    T* d = dx + pix * dx_cs + dx_c0 + ci;
    if constexpr (is_x2<T>::value) {  // whole 4-channel half-groups (the pad channels of the view stay 0 + 0)
      float cur[4];
      load4<T>(d, cur);
#pragma unroll
      for (int q = 0; q < 4; ++q) cur[q] += (ci + q < cin) ? acc[q] : 0.f;
      store_vec<T, 4>(d, cur);
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (ci + q < cin) d[q] += acc[q];
    }
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
  int head_kpad, head_cin_pad;         // kind 4 (flow head from G18): row i = (tap, o) goes to (i & 1) * kpad + (i >> 1) * cin_pad; else 0
  int xcd;                             // bwd_filter_x2_kernel: XCD-aware block order (FN2_BWF_XCD=0: plain)
  int dbg;                             // FN2_BWF_DBG ablation bits (timing experiments; FN2_CONV_ABLATE builds only)
  float* db;                           // bias gradient db[i] += sum_pix Dn[pix][i] (convolutions: Dn = dY), or nullptr
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
// X2: both operands are split-fp16 tensors; the DMA moves their rows verbatim (a 16-byte chunk is the hi or the lo
// half of an 8-channel group) and an operand is rebuilt as hi + lo from two ds_read_u16 when it is fed to the fp32
// MFMA -- the matrix rate, not LDS, bounds this kernel.
template <bool SWAP, int NI, int NJ, bool X2 = false>
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
  // split fp16: chunk lane&31 belongs to the group of channels (lch & ~7)..+7
  const int lgrp = X2 ? (lch & ~7) : lch;
  const bool ch_ok_d = lch < TI && i0 + lgrp < p.Ci, ch_ok_s = lch < TJ && j0 + lgrp < p.Cj;
  // sampled-tensor pixel state of this lane's 4 rows: r = (wave*4 + k)*2 + lrow within the stage
  // walking state of the next stage to issue, advanced by additions only (see bwd_filter_x2_kernel)
  int pixd[4], sy[4], sx[4], iy[4], ix[4], offd[4], offs[4];
  const int dstep = PK * p.dn_cs * 4, xstep = PK * p.stride * p.sm_cs * 4;
  const int rowjump = (p.stride * p.SW - p.DW_ * p.stride) * p.sm_cs * 4;
  const int imgjump = (p.SH - p.DH * p.stride) * p.SW * p.sm_cs * 4;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int pix = pbeg + (wave * 4 + k) * 2 + lrow;
    const int n = pix / (p.DH * p.DW_), rem = pix - n * (p.DH * p.DW_);
    pixd[k] = pix; sy[k] = rem / p.DW_; sx[k] = rem - sy[k] * p.DW_;
    iy[k] = sy[k] * p.stride + ky - p.pad; ix[k] = sx[k] * p.stride + kx - p.pad;
    offd[k] = (pix * p.dn_cs + p.dn_c0 + i0 + lch) * 4;
    offs[k] = (((n * p.SH + iy[k]) * p.SW + ix[k]) * p.sm_cs + p.sm_c0 + j0 + lch) * 4;
  }
  auto issue = [&](int st, float (*lds)[PK * 128]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int r = (wave * 4 + k) * 2;          // first of the 2 rows of this piece (wave-uniform)
      // branch-free (see bwd_filter_x2_kernel): a valid offset is < 2^31, every reason to skip a row sets bit 31
      const unsigned pvm = (unsigned)(pend - 1 - pixd[k]) & kOobT;
      const unsigned vd = (unsigned)offd[k] | pvm | (ch_ok_d ? 0u : kOobT);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_d, (lptr_t)&lds[0][r * 128], 16, vd, 0, 0, 0);
      const unsigned padm = (unsigned)(iy[k] | (p.SH - 1 - iy[k]) | ix[k] | (p.SW - 1 - ix[k])) & kOobT;
      const unsigned vs = ((unsigned)offs[k] & ~kOobT) | pvm | padm | (ch_ok_s ? 0u : kOobT);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_s, (lptr_t)&lds[1][r * 128], 16, vs, 0, 0, 0);
      // advance this row by PK pixels for the next stage
      pixd[k] += PK; offd[k] += dstep;
      sx[k] += PK; ix[k] += PK * p.stride; offs[k] += xstep;
      while (sx[k] >= p.DW_) {
        sx[k] -= p.DW_; ix[k] -= p.DW_ * p.stride; iy[k] += p.stride; offs[k] += rowjump;
        if (++sy[k] == p.DH) { sy[k] = 0; iy[k] -= p.DH * p.stride; offs[k] += imgjump; }
      }
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
    // split fp16: channel ch of a row sits at half-word (ch >> 3) * 16 + (ch & 7) (hi) and + 8 (lo)
    const int chd = wi * NI * 32 + fr, chs = wj * NJ * 32 + fr;
    const _Float16* Dh = reinterpret_cast<const _Float16*>(&lds[0][0]) + (chd >> 3) * 16 + (chd & 7);
    const _Float16* Sh = reinterpret_cast<const _Float16*>(&lds[1][0]) + (chs >> 3) * 16 + (chs & 7);
#pragma unroll
    for (int kk = 0; kk < PK; kk += 2) {
      const int row = (kk + fk) * 128;
      float av[NI], bv[NJ];
      if constexpr (X2) {
#pragma unroll
        for (int t = 0; t < NI; ++t) av[t] = (float)Dh[row * 2 + t * 64] + (float)Dh[row * 2 + t * 64 + 8];
#pragma unroll
        for (int t = 0; t < NJ; ++t) bv[t] = (float)Sh[row * 2 + t * 64] + (float)Sh[row * 2 + t * 64 + 8];
      } else {
#pragma unroll
        for (int t = 0; t < NI; ++t) av[t] = Dn[row + 32 * t];
#pragma unroll
        for (int t = 0; t < NJ; ++t) bv[t] = Sm[row + 32 * t];
      }
#pragma unroll
      for (int ti = 0; ti < NI; ++ti)
#pragma unroll
        for (int tj = 0; tj < NJ; ++tj) {
          if constexpr (SWAP) acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[tj], av[ti], acc[ti][tj], 0, 0, 0);
          else acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ti], bv[tj], acc[ti][tj], 0, 0, 0);
        }
    }
  };
  // Bias gradient in the same pass (convolutions: Dn = dY): the blocks of tap 0, j tile 0 see every dense pixel of
  // their (i tile, pixel range) exactly once; thread t sums channel t & 127 over 16 of a stage's 32 pixel rows.
  const bool do_bias = p.db != nullptr && tap == 0 && j0 == 0;
  float bsum = 0.f;
  auto bias_stage = [&](const float (*lds)[PK * 128]) {
    const int c = tid & 127, r0 = (tid >> 7) * 16;
    if (!do_bias || c >= TI) return;
    if constexpr (X2) {
      const _Float16* h = reinterpret_cast<const _Float16*>(&lds[0][0]) + (c >> 3) * 16 + (c & 7);
#pragma unroll
      for (int r = 0; r < 16; ++r) bsum += (float)h[(r0 + r) * 256] + (float)h[(r0 + r) * 256 + 8];
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) bsum += lds[0][(r0 + r) * 128 + c];
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
    bias_stage(lds0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (st + 2 < nstage2) issue(st + 2, lds0);
    compute(lds1);
    bias_stage(lds1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (do_bias) {  // block-uniform: the two pixel halves of a channel meet in LDS (the stage buffers are free)
    if (tid >= 128) lds0[0][tid - 128] = bsum;
    __syncthreads();
    const int c = tid;
    if (tid < 128 && c < TI && i0 + c < p.Ci) atomicAdd(p.db + i0 + c, bsum + lds0[0][c]);
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
          if (i < p.Ci) atomicAdd(base + (p.head_kpad ? (long)(i & 1) * p.head_kpad + (long)(i >> 1) * p.head_cin_pad
                                                   : (long)(p.perm_i ? perm32(i) : i) * p.stride_i) + oj, acc[ti][tj][q]);
        }
      }
    }
#endif
}

// ---------------------------------------------------------------------------
// Filter gradient of the split-fp16 trainer on the FP16 matrix cores.  Same tiling, DMA and epilogue as
// bwd_filter_kernel, but the reduction index of v_mfma_f32_32x32x16_f16 is 16 PIXELS, so an operand is "8
// consecutive pixels of one channel": a column of the [pixel][channel] LDS tile.  ds_read_b64_tr_b16 delivers
// exactly that (a 4-pixel x 16-channel block, column-major to the lanes), two per operand half; hi and lo parts
// are separate 16-byte chunks of a group and give the three products hi*hi + hi*lo + lo*hi.  The 512-byte pixel
// rows would put the 4 rows of a transposed block on the same banks; the DMA therefore stores logical chunk L of
// row r at chunk L ^ ((r & 1) | ((r & 2) << 2)) -- the 16 (row, chunk) pairs of a 32-lane read then cover 16
// distinct chunk positions mod 16.  5x the matrix rate of the fp32 form (96 vs 512 cycles per 16 pixels of a
// 32 x 32 tile).
// ---------------------------------------------------------------------------
// UNI ("uniform walk"): every stage's 32 dense pixels are one piece of an image row (DW % 32 == 0) or whole rows of one
// image (32 % DW == 0, DH % (32 / DW) == 0) -- all the reference's resolutions.  Then a lane's pixel keeps its place
// inside the stage: its byte offsets are constants, the stage's position is a handful of scalars (buffer soffset) and
// the per-stage vector work is one zero-pad test per sampled piece.  The general walk (per-lane pixel state) spent more
// issue cycles on bookkeeping than on the MFMAs: 56 % of the filter-gradient time ran with neither DMA nor MFMA
// (FN2_BWF_DBG ablations).
template <bool SWAP, int NI, int NJ, bool UNI = false>
__global__ void __launch_bounds__(256) bwd_filter_x2_kernel(const BwdwArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int PK = 32;
  __shared__ float lds0[2][PK * 128];
  __shared__ float lds1[2][PK * 128];
  typedef short s4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) s4* lds_s4_t;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wave >> 1, wj = wave & 1;
  constexpr int TI = 64 * NI, TJ = 64 * NJ;
  // XCD-aware block order (as conv_igemm2_kernel): workgroup L runs on XCD L & 7; give every XCD a contiguous band of the
  // (pixel range, tap / j tile, i tile) space, pixel range slowest: the blocks of a band read the same pixels (every tap and
  // tile of a pixel range walks the same dense rows and neighbouring sampled rows), which then meet in that XCD's L2
  int bxi = blockIdx.x, byi = blockIdx.y, bzi = blockIdx.z;
  if (p.xcd) {
    const int NT = gridDim.x * gridDim.y * gridDim.z;
    const int L = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const int xcd = L & 7, chunk = NT >> 3, rem = NT & 7;
    int Lp = xcd * chunk + min(xcd, rem) + (L >> 3);
    bxi = Lp % (int)gridDim.x; Lp /= (int)gridDim.x;
    byi = Lp % (int)gridDim.y; bzi = Lp / (int)gridDim.y;
  }
  const int i0 = bxi * TI;
  const int njt = (p.Cj + TJ - 1) / TJ;
  const int tap = byi / njt, j0 = (byi - tap * njt) * TJ;
  const int ky = tap / p.KW, kx = tap - ky * p.KW;
  const int pbeg = bzi * p.pix_per_split, pend = min(p.P, pbeg + p.pix_per_split);
  if (pbeg >= pend) return;
  const int nstage = (pend - pbeg + PK - 1) / PK;

  const auto rs_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dn), 0, p.dn_bytes, 0x00020000);
  const auto rs_s = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.sm), 0, p.sm_bytes, 0x00020000);

  // DMA: lane -> (row lane>>5 of the 2-row piece, physical chunk lane&31); the logical chunk depends on the row
  const int lrow = lane >> 5, lphys = lane & 31;
  const int dstep = PK * p.dn_cs * 4, xstep = PK * p.stride * p.sm_cs * 4;
  const int rowjump = (p.stride * p.SW - p.DW_ * p.stride) * p.sm_cs * 4;  // first pixel of the next dense row
  const int imgjump = (p.SH - p.DH * p.stride) * p.SW * p.sm_cs * 4;       // ... of the next image
  // ---- UNI: lane constants + scalar stage position
  const int sbias = (p.pad * p.SW + p.pad) * p.sm_cs * 4;  // the sampled descriptor starts this far in front of the tensor, so that lane offsets are >= 0
  const auto rs_su = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.sm)) - sbias, 0, p.sm_bytes + sbias, 0x00020000);
  unsigned uvd[4], uvs[4];
  int uiy[4], uix[4];
  int u_soff_d = 0, u_soff_s = 0, u_yb = 0, u_xb = 0, u_sy = 0, u_sx = 0, u_pix = pbeg;
  const int rps = p.DW_ >= PK ? 1 : PK / p.DW_;            // image rows per stage
  const int u_rowstep = rps * p.stride * p.SW * p.sm_cs * 4;
  if constexpr (UNI) {
    const int n0 = pbeg / (p.DH * p.DW_), rem0 = pbeg - n0 * (p.DH * p.DW_);
    u_yb = rem0 / p.DW_; u_xb = rem0 - u_yb * p.DW_;
    u_sy = u_yb * p.stride; u_sx = u_xb * p.stride;
    u_soff_d = pbeg * p.dn_cs * 4;
    u_soff_s = ((n0 * p.SH + u_sy) * p.SW + u_sx) * p.sm_cs * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int r = (wave * 4 + k) * 2 + lrow;
      const int dy = p.DW_ >= PK ? 0 : r / p.DW_, dx = p.DW_ >= PK ? r : r - dy * p.DW_;
      uiy[k] = dy * p.stride + ky - p.pad; uix[k] = dx * p.stride + kx - p.pad;
      const int L = lphys ^ ((r & 1) | ((r & 2) << 2));
      const int grp_ch = (L >> 1) * 8;
      uvd[k] = (grp_ch < TI && i0 + grp_ch < p.Ci) ? (unsigned)((r * p.dn_cs + p.dn_c0 + i0) * 4 + L * 16) : kOobT;
      uvs[k] = (grp_ch < TJ && j0 + grp_ch < p.Cj)
                   ? (unsigned)(((uiy[k] * p.SW + uix[k]) * p.sm_cs + p.sm_c0 + j0) * 4 + L * 16 + sbias) : kOobT;
    }
  }
  auto issue_uni = [&](float (*lds)[PK * 128]) {
    // branch-free: offsets are < 2^31 or kOobT = 2^31, so "out of range" is OR-ing the sign bit in.  (Written with ?: and
    // &&, the compiler built a tree of uniform and exec-mask branches around duplicated DMA instructions.)
    const unsigned okm = u_pix < pend ? 0u : kOobT;   // uniform: ranges and the tensor are whole stages
    const int shm1 = p.SH - 1, swm1 = p.SW - 1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int r = (wave * 4 + k) * 2;
      if (!(FN2_BWF_ABLATE && (p.dbg & 2)))
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_d, (lptr_t)&lds[0][r * 128], 16, uvd[k] | okm, u_soff_d, 0, 0);
      const int ty = u_sy + uiy[k], tx = u_sx + uix[k];
      const unsigned pad = (unsigned)(ty | (shm1 - ty) | tx | (swm1 - tx)) & kOobT;  // sign bit: a coordinate < 0 or > size - 1
      if (!(FN2_BWF_ABLATE && (p.dbg & 1)))
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_su, (lptr_t)&lds[1][r * 128], 16, uvs[k] | okm | pad, u_soff_s, 0, 0);
    }
    u_pix += PK; u_soff_d += dstep;
    if (p.DW_ >= PK) {
      u_xb += PK; u_sx += PK * p.stride; u_soff_s += xstep;
      if (u_xb == p.DW_) {
        u_xb = 0; u_sx = 0; u_soff_s += rowjump; u_sy += p.stride;
        if (++u_yb == p.DH) { u_yb = 0; u_sy = 0; u_soff_s += imgjump; }
      }
    } else {
      u_yb += rps; u_sy += rps * p.stride; u_soff_s += u_rowstep;
      if (u_yb == p.DH) { u_yb = 0; u_sy = 0; u_soff_s += imgjump; }
    }
  };
  // ---- general walk.  Per-row state: every quantity of the NEXT stage to issue is kept up to date by additions (the products
  // pixel * stride, row * width ... of the first version were 32 quarter-rate integer multiplies per two stages and wave)
  int pixd[4], sy[4], sx[4], iy[4], ix[4], offd[4], offs[4];
  unsigned coff_d[4], coff_s[4];  // byte offset of this lane's logical chunk inside the tile's 512-byte row, or OOB
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = (wave * 4 + k) * 2 + lrow;
    const int pix = pbeg + r;
    const int n = pix / (p.DH * p.DW_), rem = pix - n * (p.DH * p.DW_);
    pixd[k] = pix; sy[k] = rem / p.DW_; sx[k] = rem - sy[k] * p.DW_;
    iy[k] = sy[k] * p.stride + ky - p.pad; ix[k] = sx[k] * p.stride + kx - p.pad;
    offd[k] = (pix * p.dn_cs + p.dn_c0 + i0) * 4;
    offs[k] = (((n * p.SH + iy[k]) * p.SW + ix[k]) * p.sm_cs + p.sm_c0 + j0) * 4;
    const int L = lphys ^ ((r & 1) | ((r & 2) << 2));
    const int grp_ch = (L >> 1) * 8;  // first channel of the group this chunk belongs to
    coff_d[k] = (grp_ch < TI && i0 + grp_ch < p.Ci) ? (unsigned)(L * 16) : kOobT;
    coff_s[k] = (grp_ch < TJ && j0 + grp_ch < p.Cj) ? (unsigned)(L * 16) : kOobT;
  }
  auto issue = [&](int st, float (*lds)[PK * 128]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int r = (wave * 4 + k) * 2;
      // branch-free (see issue_uni): a valid offset is < 2^31, every reason to skip a row sets bit 31
      const unsigned pvm = (unsigned)(pend - 1 - pixd[k]) & kOobT;   // pixel past the range
      const unsigned vd = ((unsigned)offd[k] + coff_d[k]) | pvm;      // (coff = 2^31 for a padding chunk)
      if (!(FN2_BWF_ABLATE && (p.dbg & 2))) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_d, (lptr_t)&lds[0][r * 128], 16, vd, 0, 0, 0);
      const unsigned padm = (unsigned)(iy[k] | (p.SH - 1 - iy[k]) | ix[k] | (p.SW - 1 - ix[k])) & kOobT;
      const unsigned vs = (((unsigned)offs[k] & ~kOobT) + coff_s[k]) | pvm | padm;
      if (!(FN2_BWF_ABLATE && (p.dbg & 1))) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_s, (lptr_t)&lds[1][r * 128], 16, vs, 0, 0, 0);
      pixd[k] += PK; offd[k] += dstep;
      sx[k] += PK; ix[k] += PK * p.stride; offs[k] += xstep;
      while (sx[k] >= p.DW_) {
        sx[k] -= p.DW_; ix[k] -= p.DW_ * p.stride; iy[k] += p.stride; offs[k] += rowjump;
        if (++sy[k] == p.DH) { sy[k] = 0; iy[k] -= p.DH * p.stride; offs[k] += imgjump; }
      }
    }
  };

  f32x16 acc[NI][NJ];
#pragma unroll
  for (int a = 0; a < NI; ++a)
#pragma unroll
    for (int b = 0; b < NJ; ++b)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.f;

  // transposed-read geometry of this lane: 16-lane group gq = channels 16 gq .. +15 of a 32-channel tile, lane
  // 4q + pq of the group addresses pixel row q, channels 4 pq .. +3; the lane half (lane >> 5) = pixels + 8
  const int kg = lane >> 5, gq = (lane >> 4) & 1, tq = (lane & 15) >> 2, pq = lane & 3;
  const int ch_in_tile = 16 * gq + 4 * pq;  // multiple of 4
  auto frag = [&](const float* tile, int ch0, int part, int pb) -> uint4 {
    // 8 pixels pb + 8 kg .. +7 of channels ch0 .. (column-major to the lanes): hi (part 0) or lo (part 1) halves
    const int ch = ch0 + ch_in_tile;
    const int L = 2 * (ch >> 3) + part;
    const char* base = reinterpret_cast<const char*>(tile);
    s4 v[2];
#pragma unroll
    for (int sblk = 0; sblk < 2; ++sblk) {
      const int row = pb + 8 * kg + 4 * sblk + tq;
      const int phys = L ^ ((row & 1) | ((row & 2) << 2));
      v[sblk] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t)(base + row * 512 + phys * 16 + (ch & 4) * 2));
    }
    uint4 o;
    o.x = __builtin_bit_cast(uint2, v[0]).x; o.y = __builtin_bit_cast(uint2, v[0]).y;
    o.z = __builtin_bit_cast(uint2, v[1]).x; o.w = __builtin_bit_cast(uint2, v[1]).y;
    return o;
  };
  auto compute = [&](const float (*lds)[PK * 128]) {
    if (FN2_BWF_ABLATE && (p.dbg & 4)) return;
#pragma unroll
    for (int pb = 0; pb < PK; pb += 16) {
      uint4 ah[NI], al[NI], bh[NJ], bl[NJ];
#pragma unroll
      for (int t = 0; t < NI; ++t) { ah[t] = frag(lds[0], wi * NI * 32 + t * 32, 0, pb); al[t] = frag(lds[0], wi * NI * 32 + t * 32, 1, pb); }
#pragma unroll
      for (int t = 0; t < NJ; ++t) { bh[t] = frag(lds[1], wj * NJ * 32 + t * 32, 0, pb); bl[t] = frag(lds[1], wj * NJ * 32 + t * 32, 1, pb); }
#pragma unroll
      for (int ti = 0; ti < NI; ++ti)
#pragma unroll
        for (int tj = 0; tj < NJ; ++tj) {
          if constexpr (SWAP) {
            acc[ti][tj] = mfma_32x32x16<f16_t>(bl[tj], ah[ti], acc[ti][tj]);
            acc[ti][tj] = mfma_32x32x16<f16_t>(bh[tj], al[ti], acc[ti][tj]);
            acc[ti][tj] = mfma_32x32x16<f16_t>(bh[tj], ah[ti], acc[ti][tj]);
          } else {
            acc[ti][tj] = mfma_32x32x16<f16_t>(al[ti], bh[tj], acc[ti][tj]);
            acc[ti][tj] = mfma_32x32x16<f16_t>(ah[ti], bl[tj], acc[ti][tj]);
            acc[ti][tj] = mfma_32x32x16<f16_t>(ah[ti], bh[tj], acc[ti][tj]);
          }
        }
    }
  };
  // Bias gradient in the same pass (see bwd_filter_kernel); channel c of pixel row r: hi half-word (c & 7) of chunk
  // 2 (c >> 3) ^ swz(r), lo of the chunk after it (before the swizzle)
  const bool do_bias = p.db != nullptr && tap == 0 && j0 == 0;
  float bsum = 0.f;
  auto bias_stage = [&](const float (*lds)[PK * 128]) {
    const int c = tid & 127, r0 = (tid >> 7) * 16;
    if (!do_bias || c >= TI) return;
    const char* base = reinterpret_cast<const char*>(&lds[0][0]) + (c & 7) * 2;
    const int L = 2 * (c >> 3);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = r0 + r, sw = (row & 1) | ((row & 2) << 2);
      bsum += (float)*reinterpret_cast<const _Float16*>(base + row * 512 + (L ^ sw) * 16) +
              (float)*reinterpret_cast<const _Float16*>(base + row * 512 + ((L + 1) ^ sw) * 16);
    }
  };
  auto issue_any = [&](int st, float (*lds)[PK * 128]) {
    if constexpr (UNI) issue_uni(lds);
    else issue(st, lds);
  };
  issue_any(0, lds0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int nstage2 = (nstage + 1) & ~1;
  for (int st = 0; st < nstage2; st += 2) {
    issue_any(st + 1, lds1);
    compute(lds0);
    bias_stage(lds0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (st + 2 < nstage2) issue_any(st + 2, lds0);
    compute(lds1);
    bias_stage(lds1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (do_bias) {
    if (tid >= 128) lds0[0][tid - 128] = bsum;
    __syncthreads();
    if (tid < 128 && tid < TI && i0 + tid < p.Ci) atomicAdd(p.db + i0 + tid, bsum + lds0[0][tid]);
  }
  const int fr = lane & 31, fk = lane >> 5;
  float* base = p.dw + p.tap_base[tap];
  if (FN2_BWF_ABLATE && (p.dbg & 8)) return;
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
          if (i < p.Ci) atomicAdd(base + (p.head_kpad ? (long)(i & 1) * p.head_kpad + (long)(i >> 1) * p.head_cin_pad
                                                   : (long)(p.perm_i ? perm32(i) : i) * p.stride_i) + oj, acc[ti][tj][q]);
        }
      }
    }
#endif
}

}  // namespace fn2

using namespace fn2;

extern "C" {

int fn2_epe_loss_grad(const float* pred, const float* label, float* dpred, float* loss_accum, int n, int h, int w,
                      float weight, float grad_mult, void* stream) {
  FN2_REQUIRE(pred && label && dpred && loss_accum, "epe_loss_grad: null pointer");
  FN2_REQUIRE(n >= 1 && h >= 1 && w >= 1, "epe_loss_grad: bad dims");
  const long npix = (long)n * h * w;
  hipLaunchKernelGGL(epe_loss_grad_kernel, dim3(grid_for(npix, 256)), dim3(256), 0, (hipStream_t)stream, pred, label,
                     (const float*)nullptr, dpred, loss_accum, npix, weight / (float)n, grad_mult);
  FN2_CHECK_LAUNCH("epe_loss_grad");
  return FN2_OK;
}

int fn2_epe_loss_grad_weighted(const float* pred, const float* label, const float* pixel_weight, float* dpred,
                               float* loss_accum, int n, int h, int w, float weight, float grad_mult, void* stream) {
  FN2_REQUIRE(pred && label && pixel_weight && dpred && loss_accum, "epe_loss_grad_weighted: null pointer");
  FN2_REQUIRE(n >= 1 && h >= 1 && w >= 1, "epe_loss_grad_weighted: bad dims");
  const long npix = (long)n * h * w;
  hipLaunchKernelGGL(epe_loss_grad_kernel, dim3(grid_for(npix, 256)), dim3(256), 0, (hipStream_t)stream, pred, label,
                     pixel_weight, dpred, loss_accum, npix, weight / (float)n, grad_mult);
  FN2_CHECK_LAUNCH("epe_loss_grad_weighted");
  return FN2_OK;
}

// pixel ranges per channel block: <= 512 blocks (each ends with same-address atomics on db), but never fewer than
// ~16 pixels per thread row, so that small tensors still spread over the chip
static int bias_splits(long npix, int gpb_log2) {
  const int rows = 256 >> gpb_log2;
  long s = npix / (rows * 8 > 16 ? rows * 8 : 16);
  if (s > 512) s = 512;
  return (int)(s < 1 ? 1 : s);
}

static int gpb_log2_for(int c4) {
  int l = 0;
  while ((1 << l) < c4 && l < 8) ++l;
  return l;
}

int fn2_leaky_bwd(const fn2_tensor* y, const fn2_tensor* g, float* db, void* stream) {
  FN2_REQUIRE(y && g && y->data && g->data, "leaky_bwd: null tensor");
  FN2_REQUIRE((y->dtype == FN2_F32 || y->dtype == FN2_F16X2) && g->dtype == y->dtype, "leaky_bwd: fp32 or split fp16, both alike");
  FN2_REQUIRE(y->n == g->n && y->h == g->h && y->w == g->w && y->c == g->c, "leaky_bwd: shape mismatch");
  if (y->dtype == FN2_F16X2) {
    FN2_REQUIRE(y->c % 8 == 0 && y->cs % 8 == 0 && y->c0 % 8 == 0 && g->cs % 8 == 0 && g->c0 % 8 == 0,
                "leaky_bwd: split-fp16 slices are group (8) aligned");
    const long npx = (long)y->n * y->h * y->w;
    const int l2x = gpb_log2_for(y->c / 8);
    const int sp = bias_splits(npx, l2x);
    hipLaunchKernelGGL(act_bias_bwd_x2_kernel<true>, dim3((y->c / 8 + (1 << l2x) - 1) >> l2x, (int)sp), dim3(256), 0,
                       (hipStream_t)stream, (const x2_t*)y->data, (x2_t*)g->data, db, npx, y->c, y->cs, y->c0, g->cs,
                       g->c0, l2x);
    FN2_CHECK_LAUNCH("leaky_bwd");
    return FN2_OK;
  }
  FN2_REQUIRE(y->c % 4 == 0 && y->cs % 4 == 0 && y->c0 % 4 == 0 && g->cs % 4 == 0 && g->c0 % 4 == 0,
              "leaky_bwd: channel slice must be 4-aligned");
  const long npix = (long)y->n * y->h * y->w;
  const int l2 = gpb_log2_for(y->c / 4);
  // few, fat blocks: every block ends with one atomic per channel on the SAME db[c] addresses, and 2048 blocks
  // serialised on them cost more than the pass itself (conv1: 0.22 ms for a 300 MB pass)
  const int splits = bias_splits(npix, l2);
  hipLaunchKernelGGL(act_bias_bwd_kernel<true>, dim3((y->c / 4 + (1 << l2) - 1) >> l2, (int)splits), dim3(256), 0,
                     (hipStream_t)stream, (const float*)y->data, (float*)g->data, db, npix, y->c, y->cs, y->c0, g->cs,
                     g->c0, l2);
  FN2_CHECK_LAUNCH("leaky_bwd");
  return FN2_OK;
}

// bias gradient of a narrow split-fp16 slice (c < 8: the two upsample_flow channels of a concat gradient buffer,
// which carry a bias only in the FlowNet2 fusion net): lane per pixel, wave reduction, one atomic per wave and channel
__global__ void __launch_bounds__(256) bias_grad_x2_narrow_kernel(const x2_t* __restrict__ g, float* __restrict__ db, long npix,
                                                                  int c, int cs, int c0) {
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x)
    for (int j = 0; j < c; ++j) acc[j] += load_elem<x2_t>(g + i * cs + c0 + j);
  for (int j = 0; j < c; ++j) {
    float v = acc[j];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(db + j, v);
  }
}

int fn2_bias_grad(const fn2_tensor* g, float* db, void* stream) {
  FN2_REQUIRE(g && g->data && db, "bias_grad: null pointer");
  FN2_REQUIRE(g->dtype == FN2_F32 || g->dtype == FN2_F16X2, "bias_grad: fp32 or split-fp16 gradients");
  const long npix = (long)g->n * g->h * g->w;
  hipStream_t st = (hipStream_t)stream;
  if (g->dtype == FN2_F16X2) {
    // linear layers with a bias in the split-fp16 trainer (FlowNetSD's interconvN): the LeakyReLU pass without the
    // activation factor
    if (g->c < 8) {
      hipLaunchKernelGGL(bias_grad_x2_narrow_kernel, dim3(grid_for(npix, 1024)), dim3(256), 0, st, (const x2_t*)g->data, db,
                         npix, g->c, g->cs, g->c0);
      FN2_CHECK_LAUNCH("bias_grad");
      return FN2_OK;
    }
    FN2_REQUIRE(g->c % 8 == 0 && g->cs % 8 == 0 && g->c0 % 8 == 0, "bias_grad: split-fp16 slices are group (8) aligned");
    const int l2x = gpb_log2_for(g->c / 8);
    const int sp = bias_splits(npix, l2x);
    hipLaunchKernelGGL(act_bias_bwd_x2_kernel<false>, dim3((g->c / 8 + (1 << l2x) - 1) >> l2x, (int)sp), dim3(256), 0, st,
                       (const x2_t*)nullptr, (x2_t*)g->data, db, npix, g->c, 0, 0, g->cs, g->c0, l2x);
    FN2_CHECK_LAUNCH("bias_grad");
    return FN2_OK;
  }
  if (g->c == 2 && g->cs == 2 && g->c0 == 0) {
    hipLaunchKernelGGL(bias_grad2_kernel, dim3(grid_for(npix, 1024)), dim3(256), 0, st, (const float*)g->data, db, npix);
  } else if (g->c % 4 == 0 && g->cs % 4 == 0 && g->c0 % 4 == 0) {
    const int l2 = gpb_log2_for(g->c / 4);
    const int splits = bias_splits(npix, l2);
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

int fn2_to_f16x2(void* dst, const float* src, const int32_t* map, int64_t n, float scale, void* stream) {
  FN2_REQUIRE(dst && src && n >= 0 && n % 8 == 0, "to_f16x2: n must be a multiple of 8 (whole groups)");
  if (n == 0) return FN2_OK;
  hipLaunchKernelGGL(to_x2_kernel, dim3(grid_for(n / 8, 256)), dim3(256), 0, (hipStream_t)stream, src, map, (x2_t*)dst,
                     (long)(n / 8), scale, 0);
  FN2_CHECK_LAUNCH("to_f16x2");
  return FN2_OK;
}

int fn2_to_f16x2_frag(void* dst, const float* src, const int32_t* map, int64_t n, float scale, int k, void* stream) {
  FN2_REQUIRE(dst && src && n >= 0 && k > 0 && k % 32 == 0 && n % ((int64_t)k * 32) == 0,
              "to_f16x2_frag: a packed [rows][k] weight with k %% 32 == 0 and rows %% 32 == 0");
  if (n == 0) return FN2_OK;
  hipLaunchKernelGGL(to_x2_kernel, dim3(grid_for(n / 8, 256)), dim3(256), 0, (hipStream_t)stream, src, map, (x2_t*)dst,
                     (long)(n / 8), scale, k);
  FN2_CHECK_LAUNCH("to_f16x2_frag");
  return FN2_OK;
}

int fn2_fill_zero(void* dst, int64_t bytes, void* stream) {
  FN2_REQUIRE(dst && bytes >= 0, "fill_zero: bad arguments");
  if (bytes == 0) return FN2_OK;
  FN2_HIP(hipMemsetAsync(dst, 0, (size_t)bytes, (hipStream_t)stream));
  return FN2_OK;
}

int fn2_add_f32(float* dst, const float* src, int64_t n, void* stream) {
  FN2_REQUIRE(dst && src && n >= 0, "add: bad arguments");
  if (n == 0) return FN2_OK;
  hipLaunchKernelGGL(add_inplace_kernel, dim3(grid_for((n + 3) / 4, 256)), dim3(256), 0, (hipStream_t)stream, dst, src, (long)n);
  FN2_CHECK_LAUNCH("add");
  return FN2_OK;
}

int fn2_slice_copy_f32(const fn2_tensor* src, float* dst, void* stream) {
  FN2_REQUIRE(src && src->data && dst && src->dtype == FN2_F32, "slice_copy: fp32 source view");
  FN2_REQUIRE(src->c >= 1 && src->c0 >= 0 && src->c0 + src->c <= src->cs, "slice_copy: channel slice outside the buffer");
  const long npix = (long)src->n * src->h * src->w;
  hipLaunchKernelGGL(slice_copy_kernel, dim3(grid_for(npix * src->c, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)src->data, dst, npix, src->c, src->cs, src->c0);
  FN2_CHECK_LAUNCH("slice_copy");
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

int fn2_adam_step_multi(const void* table, const int64_t* counts, const float* l2, int n_tensors, float lr, float beta1,
                        float beta2, float eps, int step, float grad_scale, void* stream) {
  FN2_REQUIRE(table && counts && l2 && n_tensors >= 1 && n_tensors <= 65535 && step >= 1, "adam_step_multi: bad arguments");
  const float lr_t = lr * sqrtf(1.f - powf(beta2, (float)step)) / (1.f - powf(beta1, (float)step));
  hipLaunchKernelGGL(adam_multi_kernel, dim3(128, n_tensors), dim3(256), 0, (hipStream_t)stream,
                     (const AdamTensor*)table, (const long*)counts, l2, lr_t, beta1, beta2, eps, grad_scale);
  FN2_CHECK_LAUNCH("adam_multi");
  return FN2_OK;
}

int fn2_adam_step_multi_dev(const void* table, const int64_t* counts, const float* l2, int n_tensors, const float* hyper,
                            void* stream) {
  FN2_REQUIRE(table && counts && l2 && hyper && n_tensors >= 1 && n_tensors <= 65535, "adam_step_multi_dev: bad arguments");
  hipLaunchKernelGGL(adam_multi_dev_kernel, dim3(128, n_tensors), dim3(256), 0, (hipStream_t)stream,
                     (const AdamTensor*)table, (const long*)counts, l2, hyper);
  FN2_CHECK_LAUNCH("adam_multi_dev");
  return FN2_OK;
}

int fn2_upsample_flow_bwd(const fn2_tensor* g, const float* pf, const float* w, float* dpf, float* dw, int accumulate,
                          void* stream) {
  FN2_REQUIRE(g && g->data && pf && w && dpf && dw, "upsample_flow_bwd: null pointer");
  FN2_REQUIRE((g->dtype == FN2_F32 || g->dtype == FN2_F16X2) && g->c == 2 && g->h % 2 == 0 && g->w % 2 == 0,
              "upsample_flow_bwd: g must be a 2-channel fp32 / split-fp16 view of even size");
  const int H = g->h / 2, W = g->w / 2;
  if (g->dtype == FN2_F16X2)
    hipLaunchKernelGGL(upsample_flow_bwd_kernel<x2_t>, dim3(grid_for((long)g->n * H * W, 256)), dim3(256), 0,
                       (hipStream_t)stream, (const x2_t*)g->data, g->cs, g->c0, pf, w, dpf, dw, g->n, H, W, accumulate);
  else
    hipLaunchKernelGGL(upsample_flow_bwd_kernel<float>, dim3(grid_for((long)g->n * H * W, 256)), dim3(256), 0,
                       (hipStream_t)stream, (const float*)g->data, g->cs, g->c0, pf, w, dpf, dw, g->n, H, W, accumulate);
  FN2_CHECK_LAUNCH("upsample_flow_bwd");
  return FN2_OK;
}

int fn2_head_g18(const float* g, const fn2_tensor* out, void* stream) {
  FN2_REQUIRE(g && out && out->data, "head_g18: null pointer");
  FN2_REQUIRE(out->dtype == FN2_F16X2 && out->c == 18 && out->cs % 8 == 0 && out->c0 % 8 == 0 && out->cs - out->c0 >= 24,
              "head_g18: the output is an 18-channel split-fp16 view with room for three 8-channel groups");
  const long total = (long)out->n * out->h * out->w * 3;
  hipLaunchKernelGGL(head_g18_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, g, (x2_t*)out->data,
                     out->cs, out->c0, out->n, out->h, out->w);
  FN2_CHECK_LAUNCH("head_g18");
  return FN2_OK;
}

int fn2_head_bwd_filter(const fn2_tensor* x, const float* g, float* dw, int cin_pad, int kpad, void* stream) {
  FN2_REQUIRE(x && x->data && g && dw, "head_bwd_filter: null pointer");
  FN2_REQUIRE((x->dtype == FN2_F32 || x->dtype == FN2_F16X2) && cin_pad >= x->c && kpad >= 9 * cin_pad, "head_bwd_filter: bad layout");
  FN2_REQUIRE(x->cs % 4 == 0 && x->c0 % 4 == 0 && (x->c + 3) / 4 * 4 <= x->cs - x->c0, "head_bwd_filter: x view must be 16-byte aligned and padded to 4 channels");
  const long npix = (long)x->n * x->h * x->w;
  // pixel ranges: ~512 pixels per block on large maps, but at least ~256 blocks in total on the small ones
  // (the 6x8 head had 16 blocks walking 384 pixels each: 45 us for a 1.5 MB tensor)
  const int cblocks = (x->c + 63) / 64;
  int splits = (int)((npix + 511) / 512);
  const int want = (256 + cblocks - 1) / cblocks;
  if (splits < want) splits = (int)((npix + 31) / 32 < want ? (npix + 31) / 32 : want);
  if (splits > 512) splits = 512;
  if (splits < 1) splits = 1;
  if (x->dtype == FN2_F16X2)
    hipLaunchKernelGGL(head_bwd_filter_kernel<x2_t>, dim3((x->c + 63) / 64, splits), dim3(256), 0, (hipStream_t)stream,
                       (const x2_t*)x->data, x->cs, x->c0, x->c, g, dw, cin_pad, kpad, x->n, x->h, x->w);
  else
    hipLaunchKernelGGL(head_bwd_filter_kernel<float>, dim3((x->c + 63) / 64, splits), dim3(256), 0, (hipStream_t)stream,
                       (const float*)x->data, x->cs, x->c0, x->c, g, dw, cin_pad, kpad, x->n, x->h, x->w);
  FN2_CHECK_LAUNCH("head_bwd_filter");
  return FN2_OK;
}

int fn2_head_bwd_data(const float* g, const float* w, const fn2_tensor* dx, int cin_pad, int kpad, void* stream) {
  FN2_REQUIRE(g && w && dx && dx->data, "head_bwd_data: null pointer");
  FN2_REQUIRE((dx->dtype == FN2_F32 || dx->dtype == FN2_F16X2) && cin_pad >= dx->c && cin_pad % 4 == 0 && kpad >= 9 * cin_pad,
              "head_bwd_data: bad layout");
  FN2_REQUIRE(dx->cs % 4 == 0 && dx->c0 % 4 == 0, "head_bwd_data: dx view must be 16-byte aligned");
  const long total = (long)dx->n * dx->h * dx->w * ((dx->c + 3) / 4);
  if (dx->dtype == FN2_F16X2)
    hipLaunchKernelGGL(head_bwd_data_kernel<x2_t>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, g, w,
                       (x2_t*)dx->data, dx->cs, dx->c0, dx->c, cin_pad, kpad, dx->n, dx->h, dx->w);
  else
    hipLaunchKernelGGL(head_bwd_data_kernel<float>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, g, w,
                       (float*)dx->data, dx->cs, dx->c0, dx->c, cin_pad, kpad, dx->n, dx->h, dx->w);
  FN2_CHECK_LAUNCH("head_bwd_data");
  return FN2_OK;
}

int fn2_conv2d_bwd_filter(const fn2_bwdw_desc* d, void* stream) {
  FN2_REQUIRE(d && d->x.data && d->dy.data && d->dw, "bwd_filter: null pointer");
  FN2_REQUIRE((d->x.dtype == FN2_F32 || d->x.dtype == FN2_F16X2) && d->dy.dtype == d->x.dtype,
              "bwd_filter: x and dy must both be fp32 or both split fp16");
  if (d->x.dtype == FN2_F16X2)
    FN2_REQUIRE(d->x.cs % 8 == 0 && d->x.c0 % 8 == 0 && d->dy.cs % 8 == 0 && d->dy.c0 % 8 == 0, "bwd_filter: split-fp16 views are group (8) aligned");
  FN2_REQUIRE((d->kind >= 0 && d->kind <= 2) || d->kind == 4,
              "bwd_filter: kind 0 (conv), 1 (deconv k4 s2 crop 1), 2 (stem row-run conv) or 4 (flow head from fn2_head_g18)");
  FN2_REQUIRE(d->x.n == d->dy.n, "bwd_filter: batch mismatch");
  FN2_REQUIRE(d->cin_pad % 8 == 0 && d->cout_pad >= d->dy.c && d->kpad > 0, "bwd_filter: bad packed sizes");
  FN2_REQUIRE((d->x.cs % 4) == 0 && (d->x.c0 % 4) == 0 && (d->dy.cs % 4) == 0 && (d->dy.c0 % 4) == 0,
              "bwd_filter: views must be 16-byte aligned");
  BwdwArgs a;
  a.head_kpad = a.head_cin_pad = 0;
  const fn2_tensor* dn;
  const fn2_tensor* sm;
  if (d->kind == 4) {
    // flow head: dy = G18 (fn2_head_g18: 18 channels = 9 taps x 2 outputs, already shifted), a 1x1 product with x; the
    // rows land in the head's natural layout dw[o][tap * cin_pad + ci]
    FN2_REQUIRE(d->dy.c == 18 && d->dy.h == d->x.h && d->dy.w == d->x.w && d->kpad >= 9 * d->cin_pad && d->wgt_layout == 0 && d->db == nullptr,
                "bwd_filter: kind 4 takes the 18-channel tensor of fn2_head_g18 and the head's natural weight layout");
    dn = &d->dy; sm = &d->x;
    a.KH = 1; a.KW = 1; a.stride = 1; a.pad = 0;
    a.stride_i = 0; a.stride_j = 1; a.perm_i = 0; a.perm_j = 0;
    a.tap_base[0] = 0;
    a.head_kpad = d->kpad; a.head_cin_pad = d->cin_pad;
  } else if (d->kind == 0 || d->kind == 2) {
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
  FN2_REQUIRE(d->db == nullptr || d->kind != 1, "bwd_filter: the fused bias gradient sums dy, the dense operand of convolutions only");
  a.db = d->db;
  { const char* e = getenv("FN2_BWF_DBG"); a.dbg = e ? atoi(e) : 0; }
  { const char* e = getenv("FN2_BWF_XCD"); a.xcd = e ? atoi(e) : 1; }
  a.N = dn->n; a.DH = dn->h; a.DW_ = dn->w; a.dn_cs = dn->cs; a.dn_c0 = dn->c0; a.Ci = dn->c;
  a.SH = sm->h; a.SW = sm->w; a.sm_cs = sm->cs; a.sm_c0 = sm->c0; a.Cj = d->kind == 2 ? d->kw * sm->cs : sm->c;
  const long dnb = (long)dn->n * dn->h * dn->w * dn->cs * 4, smb = (long)sm->n * sm->h * sm->w * sm->cs * 4;
  FN2_REQUIRE(dnb < (1L << 31) && smb < (1L << 31), "bwd_filter: tensors >= 2 GiB are not addressable");
  a.dn_bytes = (int)dnb; a.sm_bytes = (int)smb;
  a.P = dn->n * dn->h * dn->w;
  const int taps = a.KH * a.KW;
  const int ni = a.Ci <= 64 ? 1 : 2, nj = a.Cj <= 64 ? 1 : 2;
  const int it = (a.Ci + 64 * ni - 1) / (64 * ni), jt = (a.Cj + 64 * nj - 1) / (64 * nj);
  // pixel splits: aim at >= ~384 blocks, ranges multiples of 32 pixels
  long blocks = (long)it * jt * taps;
  // (measured on the FlowNetS step: 1024 -> 5.79 ms, 512 -> 5.77, 384 -> 5.64, 256 -> 5.75, 192 -> 6.0: a quarter of the
  // kernel's time is its fp32 atomics, whose count grows with the pixel splits)
  static const int want_blocks = [] { const char* e = getenv("FN2_BWF_BLOCKS"); return e ? atoi(e) : 384; }();
  int splits = (int)((want_blocks + blocks - 1) / blocks);
  const int max_splits = (a.P + 255) / 256;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  a.pix_per_split = (((a.P + splits - 1) / splits) + 31) / 32 * 32;
  splits = (a.P + a.pix_per_split - 1) / a.pix_per_split;
  const dim3 grid(it, jt * taps, splits), block(256);
  hipStream_t st = (hipStream_t)stream;
#define FN2_BWF(SW_, NI_, NJ_)                                                                       \
  do {                                                                                               \
    if (x2 && x2_mfma && uni) hipLaunchKernelGGL((bwd_filter_x2_kernel<SW_, NI_, NJ_, true>), grid, block, 0, st, a); \
    else if (x2 && x2_mfma) hipLaunchKernelGGL((bwd_filter_x2_kernel<SW_, NI_, NJ_, false>), grid, block, 0, st, a); \
    else if (x2) hipLaunchKernelGGL((bwd_filter_kernel<SW_, NI_, NJ_, true>), grid, block, 0, st, a);  \
    else hipLaunchKernelGGL((bwd_filter_kernel<SW_, NI_, NJ_, false>), grid, block, 0, st, a);        \
  } while (0)
  const bool x2 = d->x.dtype == FN2_F16X2;
  // uniform walk (bwd_filter_x2_kernel<.., UNI>): a stage = 32 pixels of one image row, or whole rows of one image.
  // FN2_BWF_UNI=0: the general per-lane walk everywhere (A/B)
  static const bool uni_on = [] { const char* e = getenv("FN2_BWF_UNI"); return !e || atoi(e) != 0; }();
  const bool uni = uni_on && (a.DW_ % 32 == 0 || (32 % a.DW_ == 0 && a.DH % (32 / a.DW_) == 0));
  const char* dbg_env = getenv("FN2_CONV_DBG");
  const bool x2_mfma = !(dbg_env && (atoi(dbg_env) & 128));  // bit 128: the fp32-MFMA form on split-fp16 tensors (A/B)
  if (d->kind == 1) {  // (kinds 0, 2, 4 below)
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
