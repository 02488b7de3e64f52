// Convolution / transposed-convolution layers as implicit GEMM on the gfx950 matrix cores.
//
//   D[cout][pixel] += W[cout][k] * X[pixel][k],  k = (tap, channel)
//
// fp32 path : v_mfma_f32_16x16x4_f32  (exact fp32 FMA chains -- the parity path)
// bf16 path : v_mfma_f32_16x16x32_bf16 (fp32 accumulate)
//
// Both paths share one byte geometry: a "chunk" is 16 bytes of channels (4 fp32 or 8 bf16), a
// k-step is 4 chunks = one 64-byte LDS row per operand row.  Lane l of a wave reads row (l&15),
// chunk (l>>4) with one ds_read_b128; in bf16 that IS the 16x16x32 operand (k = 8*(l>>4)+j), in
// fp32 the four floats feed four 16x16x4 MFMAs (a permutation of k shared by both operands).
//
// Fused into the kernel: the explicit zero padding of the reference (utils.py:408-412 pad() ->
// predicated loads), bias, LeakyReLU (utils.py:401-405), the concat (tf.concat axis=3 -> the
// output is written into a channel slice of the consumer's buffer), the antipad crop of the
// transposed convolution (utils.py:415-421) and its zero-insertion (phase decomposition: four
// 2x2 stride-1 convolutions, blockIdx.z = phase).
//
// Replaces, for the reference, slim.conv2d / slim.conv2d_transpose (cuDNN) at
// src/flownet_s/flownet_s.py:39-104 and the same call sites in flownet_c/sd/2.
#include "conv_common.h"
#include <cstdlib>

namespace fn2 {

// TC: 16-cout MFMA tiles per wave; WC x WP waves over (cout, pixels); each wave owns 64 pixels.
template <typename T, typename OutT, int TC, int WC, int WP>
__global__ void __launch_bounds__(256) conv_igemm_kernel(const ConvArgs p) {
  constexpr int CH = 16 / (int)sizeof(T);
  constexpr int BC = WC * TC * 16;
  constexpr int BP = WP * 64;
  constexpr int NT = 256;
  static_assert(WC * WP == 4, "4 waves per block");
  constexpr int NPR = BP / 64;                 // pixel-row chunks per thread per k-step
  constexpr int NWR = (BC * 4 + NT - 1) / NT;  // weight-row chunks per thread per k-step
  constexpr int WSWZ_BIT = (TC == 4) ? 5 : (TC == 2) ? 4 : 3;
  __shared__ uint4 lds[2][(BC + BP) * 4];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wc = wave / WP, wp = wave % WP;
  const int cid = tid & 3;  // chunk column this thread stages

  int pad_y = p.pad, pad_x = p.pad, oy_off = 0, ox_off = 0, osc = 1;
  const T* wgt = reinterpret_cast<const T*>(p.wgt);
  const int phase = blockIdx.z / p.splitk, split = blockIdx.z - phase * p.splitk;
  if (p.deconv) {
    const int a = phase >> 1, b = phase & 1;
    pad_y = a ? p.ph_pad1 : p.ph_pad0; pad_x = b ? p.ph_pad1 : p.ph_pad0; oy_off = a; ox_off = b; osc = 2;
    wgt += (size_t)phase * p.cout_pad * p.ksteps * 4 * CH;
  }
  const int kt0 = split * p.kper;
  const int kt1 = min(p.ksteps, kt0 + p.kper);
  const int m0 = blockIdx.x * BP;
  const int c0 = blockIdx.y * BC;
  const T* in = reinterpret_cast<const T*>(p.in);

  // ---- per-thread staging state
  int iy0[NPR], ix0[NPR];
  size_t pbase[NPR];
  bool pvalid[NPR];
#pragma unroll
  for (int q = 0; q < NPR; ++q) {
    const int m = m0 + (tid >> 2) + 64 * q;
    pvalid[q] = m < p.M;
    const int mm = pvalid[q] ? m : 0;
    const int n = mm / (p.OH * p.OW);
    const int rem = mm - n * (p.OH * p.OW);
    const int oy = rem / p.OW, ox = rem - oy * p.OW;
    iy0[q] = oy * p.stride - pad_y;
    ix0[q] = ox * p.stride - pad_x;
    pbase[q] = (size_t)n * p.H * p.W;
  }
  const size_t wrow_elems = (size_t)p.ksteps * 4 * CH;
  // tap state of chunk column `cid`
  const int q0 = kt0 * 4 + cid;
  int cc = q0 % p.cin_chunks;
  int tap = q0 / p.cin_chunks;
  int ky = tap / p.KW, kx = tap - ky * p.KW;

  uint4 rp[NPR], rw[NWR];
  auto load_step = [&](int kt) {
    const bool tap_ok = ky < p.KH;
#pragma unroll
    for (int q = 0; q < NPR; ++q) {
      const int iy = iy0[q] + ky, ix = ix0[q] + kx;
      const bool ok = pvalid[q] && tap_ok && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (ok) {
        const T* src = in + (pbase[q] + (size_t)iy * p.W + ix) * p.in_cs + p.in_c0 + cc * CH;
        v = *reinterpret_cast<const uint4*>(src);
      }
      rp[q] = v;
    }
#pragma unroll
    for (int q = 0; q < NWR; ++q) {
      const int e = tid + NT * q;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (e < BC * 4) {
        const T* src = wgt + (size_t)(c0 + (e >> 2)) * wrow_elems + ((size_t)kt * 4 + cid) * CH;
        v = *reinterpret_cast<const uint4*>(src);
      }
      rw[q] = v;
    }
  };
  auto advance = [&]() {
    cc += 4;
    while (cc >= p.cin_chunks) {
      cc -= p.cin_chunks;
      if (++kx == p.KW) { kx = 0; ++ky; }
    }
  };
  auto store_step = [&](int buf) {
#pragma unroll
    for (int q = 0; q < NWR; ++q) {
      const int e = tid + NT * q;
      if (e < BC * 4) {
        const int row = e >> 2;
        lds[buf][row * 4 + (cid ^ (((row >> WSWZ_BIT) & 1) * 3))] = rw[q];
      }
    }
#pragma unroll
    for (int q = 0; q < NPR; ++q) {
      const int row = (tid >> 2) + 64 * q;
      lds[buf][(BC + row) * 4 + (cid ^ (((row >> 3) & 1) * 3))] = rp[q];
    }
  };

  // ---- fragment addresses (constant over k)
  const int fi = lane & 15, fchunk = (lane >> 4) ^ (((fi >> 3) & 1) * 3);
  int a_off[TC], b_off[4];
#pragma unroll
  for (int t = 0; t < TC; ++t)
    a_off[t] = (wc * TC * 16 + (fi >> 2) * (TC * 4) + t * 4 + (fi & 3)) * 4 + fchunk;
#pragma unroll
  for (int pt = 0; pt < 4; ++pt) b_off[pt] = (BC + wp * 64 + pt * 16 + fi) * 4 + fchunk;

  f32x4 acc[TC][4];
#pragma unroll
  for (int t = 0; t < TC; ++t)
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) acc[t][pt] = f32x4{0.f, 0.f, 0.f, 0.f};

  load_step(kt0);
  store_step(0);
  __syncthreads();

  for (int kt = kt0; kt < kt1; ++kt) {
    const int buf = (kt - kt0) & 1;
    const bool more = kt + 1 < kt1;
    if (more) {
      advance();
      load_step(kt + 1);
    }
    uint4 fa[TC], fb[4];
#pragma unroll
    for (int t = 0; t < TC; ++t) fa[t] = lds[buf][a_off[t]];
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) fb[pt] = lds[buf][b_off[pt]];
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int t = 0; t < TC; ++t)
#pragma unroll
        for (int pt = 0; pt < 4; ++pt)
          acc[t][pt] = mfma_16x16x32<T>(fa[t], fb[pt], acc[t][pt]);
    } else {
      // independent accumulators back to back (fp32 MFMA: 40-cycle dependent latency, 32-cycle issue)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < TC; ++t)
#pragma unroll
          for (int pt = 0; pt < 4; ++pt)
            acc[t][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(f32x4, fa[t])[j],
                                                             __builtin_bit_cast(f32x4, fb[pt])[j], acc[t][pt], 0, 0, 0);
    }
    if (more) store_step(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: lane holds, per pixel tile, TC*4 consecutive couts of one pixel
  OutT* out = reinterpret_cast<OutT*>(p.out);
  const int cout_base = c0 + wc * TC * 16 + (lane >> 4) * (TC * 4);
  if (p.splitk > 1) {
    float* slab = p.ws + (size_t)split * p.N * p.out_H * p.out_W * p.ws_cs;
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) {
      const int m = m0 + wp * 64 + pt * 16 + fi;
      if (m >= p.M) continue;
      const int n = m / (p.OH * p.OW);
      const int rem = m - n * (p.OH * p.OW);
      const int oy = rem / p.OW, ox = rem - oy * p.OW;
      float* po = slab + (((size_t)n * p.out_H + (oy * osc + oy_off)) * p.out_W + (ox * osc + ox_off)) * p.ws_cs;
#pragma unroll
      for (int t = 0; t < TC; ++t) {
        const int co = cout_base + t * 4;
        if (co < p.ws_cs)
          *reinterpret_cast<float4*>(po + co) =
              make_float4(acc[t][pt][0], acc[t][pt][1], acc[t][pt][2], acc[t][pt][3]);
      }
    }
    return;
  }
  float bias[TC][4];
#pragma unroll
  for (int t = 0; t < TC; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = cout_base + t * 4 + r;
      bias[t][r] = (p.bias != nullptr && co < p.Cout) ? p.bias[co] : 0.f;
    }
#pragma unroll
  for (int pt = 0; pt < 4; ++pt) {
    const int m = m0 + wp * 64 + pt * 16 + fi;
    if (m >= p.M) continue;
    const int n = m / (p.OH * p.OW);
    const int rem = m - n * (p.OH * p.OW);
    const int oy = rem / p.OW, ox = rem - oy * p.OW;
    OutT* po = out + (((size_t)n * p.out_H + (oy * osc + oy_off)) * p.out_W + (ox * osc + ox_off)) * p.out_cs +
               p.out_c0;
#pragma unroll
    for (int t = 0; t < TC; ++t) {
      float v[4];
      const int co = cout_base + t * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float x = acc[t][pt][r] * p.out_scale + bias[t][r];
        if constexpr (sizeof(OutT) == 4) {
          if (p.accum && co + r < p.Cout) x += load_elem<OutT>(po + co + r);
          if (co + r < p.Cout) x = act_grad<OutT>(p, out, po + co + r, co + r, x);
        }
        if (p.act == FN2_ACT_LEAKY) x = leaky(x);
        v[r] = x;
      }
      if (p.vec_ok && co + 3 < p.Cout) {
        store4<OutT>(po + co, v[0], v[1], v[2], v[3]);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (co + r < p.Cout) store_elem<OutT>(po + co + r, v[r]);
      }
    }
  }
}

// Split-K finalize: out = act(bias + sum_s slab[s]), one thread per (output pixel, 4 couts).
template <typename OutT>
__global__ void __launch_bounds__(256) splitk_finalize_kernel(const float* __restrict__ ws, const float* __restrict__ bias,
                                                              OutT* __restrict__ out, long npix, int ws_cs, int splitk,
                                                              int Cout, int out_cs, int out_c0, int act, int vec_ok,
                                                              float out_scale, int accum, const OutT* __restrict__ mask_y,
                                                              int mask_c0, int mask_c1, const float* __restrict__ up_src,
                                                              const float* __restrict__ up_w, const float* __restrict__ up_bias,
                                                              int up_c0, int out_H, int out_W) {
  const int groups = ws_cs / 4;
  const int groups_all = groups + (up_src != nullptr ? 1 : 0);
  const long total = npix * groups_all;
  const size_t slab = (size_t)npix * ws_cs;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long pix = i / groups_all;
    const int co = (int)(i - pix * groups_all) * 4;
    if (co == groups * 4) {
      // fn2_conv_desc.up_src: upsample_flowXtoY for this output pixel (upsample_flow_kernel's arithmetic, tap for tap)
      const int ox = (int)(pix % out_W), oy = (int)((pix / out_W) % out_H), n = (int)(pix / out_W / out_H);
      const int H = out_H >> 1, W = out_W >> 1;
      const int a = oy & 1, b = ox & 1, y = oy >> 1, x = ox >> 1;
      float r0 = up_bias ? up_bias[0] : 0.f, r1 = up_bias ? up_bias[1] : 0.f;
#pragma unroll
      for (int ty = 0; ty < 2; ++ty) {
        const int iy = y - 1 + a + ty, ky = 3 - a - 2 * ty;
        if (iy < 0 || iy >= H) continue;
#pragma unroll
        for (int tx = 0; tx < 2; ++tx) {
          const int ix = x - 1 + b + tx, kx = 3 - b - 2 * tx;
          if (ix < 0 || ix >= W) continue;
          const float2 v = *reinterpret_cast<const float2*>(up_src + (((long)n * H + iy) * W + ix) * 2);
          const float* ww = up_w + (ky * 4 + kx) * 4;
          r0 += v.x * ww[0] + v.y * ww[1];
          r1 += v.x * ww[2] + v.y * ww[3];
        }
      }
      OutT* pu = out + (size_t)pix * out_cs + up_c0;
      store_elem<OutT>(pu, r0);
      store_elem<OutT>(pu + 1, r1);
      continue;
    }
    float4 v = *reinterpret_cast<const float4*>(ws + pix * ws_cs + co);
    int s = 1;
    for (; s + 3 < splitk; s += 4) {  // four slabs in flight; added in split order
      float4 t[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) t[u] = *reinterpret_cast<const float4*>(ws + (s + u) * slab + pix * ws_cs + co);
#pragma unroll
      for (int u = 0; u < 4; ++u) { v.x += t[u].x; v.y += t[u].y; v.z += t[u].z; v.w += t[u].w; }
    }
    for (; s < splitk; ++s) {
      const float4 t = *reinterpret_cast<const float4*>(ws + s * slab + pix * ws_cs + co);
      v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    float r[4] = {v.x, v.y, v.z, v.w};
    OutT* po = out + (size_t)pix * out_cs + out_c0 + co;
    const bool full = vec_ok && co + 3 < Cout;
    float old[4] = {0.f, 0.f, 0.f, 0.f}, yv[4] = {1.f, 1.f, 1.f, 1.f};
    const bool mask = mask_y != nullptr && co < mask_c1 && co + 4 > mask_c0;  // fused LeakyReLU backward (act_grad, conv_common.h)
    if constexpr (sizeof(OutT) == 4) {
      if (full) {
        if (accum) load4<OutT>(po, old);
        if (mask) load4<OutT>(mask_y + (size_t)pix * out_cs + out_c0 + co, yv);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (co + j < Cout) {
            if (accum) old[j] = load_elem<OutT>(po + j);
            if (mask) yv[j] = load_elem<OutT>(mask_y + (size_t)pix * out_cs + out_c0 + co + j);
          }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      r[j] *= out_scale;
      if (bias != nullptr && co + j < Cout) r[j] += bias[co + j];
      if constexpr (sizeof(OutT) == 4) {
        if (accum) r[j] += old[j];
        if (mask && co + j >= mask_c0 && co + j < mask_c1) r[j] *= yv[j] > 0.f ? 1.f : (yv[j] < 0.f ? 0.1f : 0.55f);
      }
      if (act == FN2_ACT_LEAKY) r[j] = leaky(r[j]);
    }
    if (vec_ok && co + 3 < Cout) {
      store4<OutT>(po, r[0], r[1], r[2], r[3]);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (co + j < Cout) store_elem<OutT>(po + j, r[j]);
    }
  }
}

// ---------------------------------------------------------------------------
// Flow heads (predict_flowN: 3x3, stride 1, pad 1, Cout = 2, linear; flownet_s.py:54-56).
// Two output channels cannot feed a matrix core: HBM/L2-bound dot products.  One wavefront per
// output pixel; the 64 lanes stride the (tap, 16-byte channel chunk) space, wave64 shuffle
// reduction at the end.  Reads rows 0 and 1 of the same packed weight as the MFMA path.
// ---------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void dot_chunk(const uint4& x, const uint4& w0, const uint4& w1, float& a0, float& a1);
template <>
__device__ __forceinline__ void dot_chunk<float>(const uint4& x, const uint4& w0, const uint4& w1, float& a0,
                                                 float& a1) {
  const float4 xv = __builtin_bit_cast(float4, x), u = __builtin_bit_cast(float4, w0), v = __builtin_bit_cast(float4, w1);
  a0 += xv.x * u.x + xv.y * u.y + xv.z * u.z + xv.w * u.w;
  a1 += xv.x * v.x + xv.y * v.y + xv.z * v.z + xv.w * v.w;
}
template <>
__device__ __forceinline__ void dot_chunk<bf16_t>(const uint4& x, const uint4& w0, const uint4& w1, float& a0,
                                                  float& a1) {
  const unsigned xs[4] = {x.x, x.y, x.z, x.w}, us[4] = {w0.x, w0.y, w0.z, w0.w}, vs[4] = {w1.x, w1.y, w1.z, w1.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float xl = __uint_as_float(xs[j] << 16), xh = __uint_as_float(xs[j] & 0xffff0000u);
    a0 += xl * __uint_as_float(us[j] << 16) + xh * __uint_as_float(us[j] & 0xffff0000u);
    a1 += xl * __uint_as_float(vs[j] << 16) + xh * __uint_as_float(vs[j] & 0xffff0000u);
  }
}

template <>
__device__ __forceinline__ void dot_chunk<x2_t>(const uint4&, const uint4&, const uint4&, float&, float&) {}
template <>
__device__ __forceinline__ void dot_chunk<f16_t>(const uint4& x, const uint4& w0, const uint4& w1, float& a0,
                                                 float& a1) {
  typedef __attribute__((ext_vector_type(8))) _Float16 h8;
  const h8 xv = __builtin_bit_cast(h8, x), u = __builtin_bit_cast(h8, w0), v = __builtin_bit_cast(h8, w1);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    a0 += (float)xv[j] * (float)u[j];
    a1 += (float)xv[j] * (float)v[j];
  }
}

template <typename T>
__global__ void __launch_bounds__(256) flow_head_kernel(const ConvArgs p) {
  constexpr int CH = 16 / (int)sizeof(T);
  const int lane = threadIdx.x & 63;
  const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
  const T* in = reinterpret_cast<const T*>(p.in);
  const T* w0 = reinterpret_cast<const T*>(p.wgt);
  const T* w1 = w0 + (size_t)p.ksteps * 4 * CH;
  float* out = reinterpret_cast<float*>(p.out);
  if constexpr (is_x2<T>::value) {
    // split-fp16 activations, fp32 weights: an item is one group of 8 channels of one tap
    const float* wf0 = reinterpret_cast<const float*>(p.wgt);
    const float* wf1 = wf0 + (size_t)p.ksteps * 4 * CH;
    const int groups = p.cin_chunks >> 1;
    const int nitems = 9 * groups;
    if (p.dbg & 16384) {
      // block per pixel (small maps: the 6x8 .. 24x32 levels, 192..3072 pixels at batch 4): the four waves split the
      // 9 x C/8 items of the pixel and the partial sums meet in LDS -- four times the loads in flight per pixel of
      // the wave-per-pixel form below, whose single wave walks up to 18 dependent iterations
      __shared__ float part[8];
      for (long m = blockIdx.x; m < p.M; m += gridDim.x)
        fh_pixel(in, p.H, p.W, p.in_cs, p.in_c0, groups, wf0, wf1, p.bias, p.out_scale, out + (size_t)m * p.out_cs + p.out_c0, m, part);
      return;
    }
    for (long m = wave; m < p.M; m += nwaves) {
      const int x = (int)(m % p.W), y = (int)((m / p.W) % p.H);
      const size_t nb = (size_t)(m / p.W / p.H) * p.H * p.W;
      float a0 = 0.f, a1 = 0.f;
      for (int q = lane; q < nitems; q += 64) {
        const int tap = q / groups, gi = q - tap * groups;
        const int ky = tap / 3, kx = tap - ky * 3;
        const int iy = y + ky - 1, ix = x + kx - 1;
        if (iy < 0 || iy >= p.H || ix < 0 || ix >= p.W) continue;
        const uint4* src = reinterpret_cast<const uint4*>(in + (nb + (size_t)iy * p.W + ix) * p.in_cs + p.in_c0 + gi * 8);
        float xv[8];
        join8(src[0], src[1], xv);
        const float* u = wf0 + (size_t)q * 8;
        const float* v = wf1 + (size_t)q * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          a0 += xv[j] * u[j];
          a1 += xv[j] * v[j];
        }
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        a0 += __shfl_xor(a0, off, 64);
        a1 += __shfl_xor(a1, off, 64);
      }
      if (lane == 0) {
        float* po = out + (size_t)m * p.out_cs + p.out_c0;
        po[0] = a0 * p.out_scale + (p.bias ? p.bias[0] : 0.f);
        po[1] = a1 * p.out_scale + (p.bias ? p.bias[1] : 0.f);
      }
    }
    return;
  }
  const int items = 9 * p.cin_chunks;
  for (long m = wave; m < p.M; m += nwaves) {
    const int x = (int)(m % p.W), y = (int)((m / p.W) % p.H);
    const size_t nb = (size_t)(m / p.W / p.H) * p.H * p.W;
    float a0 = 0.f, a1 = 0.f;
    for (int q = lane; q < items; q += 64) {
      const int tap = q / p.cin_chunks, cc = q - tap * p.cin_chunks;
      const int ky = tap / 3, kx = tap - ky * 3;
      const int iy = y + ky - 1, ix = x + kx - 1;
      if (iy < 0 || iy >= p.H || ix < 0 || ix >= p.W) continue;
      const uint4 xv = *reinterpret_cast<const uint4*>(in + (nb + (size_t)iy * p.W + ix) * p.in_cs + p.in_c0 + cc * CH);
      const size_t wo = (size_t)q * CH;  // k index = tap*cin_pad + cc*CH
      dot_chunk<T>(xv, *reinterpret_cast<const uint4*>(w0 + wo), *reinterpret_cast<const uint4*>(w1 + wo), a0, a1);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      a0 += __shfl_xor(a0, off, 64);
      a1 += __shfl_xor(a1, off, 64);
    }
    if (lane == 0) {
      float* po = out + (size_t)m * p.out_cs + p.out_c0;
      po[0] = a0 * p.out_scale + (p.bias ? p.bias[0] : 0.f);
      po[1] = a1 * p.out_scale + (p.bias ? p.bias[1] : 0.f);
    }
  }
}

template <typename T, typename OutT>
static int launch_conv(const ConvArgs& a, int tile, int phases_, hipStream_t s) {
  dim3 block(256);
  const int phases = phases_ * a.splitk;
  if (tile == 128) {
    dim3 grid(cdiv(a.M, 128), a.cout_pad / 128, phases);
    hipLaunchKernelGGL((conv_igemm_kernel<T, OutT, 4, 2, 2>), grid, block, 0, s, a);
  } else if (tile == 64) {
    dim3 grid(cdiv(a.M, 256), a.cout_pad / 64, phases);
    hipLaunchKernelGGL((conv_igemm_kernel<T, OutT, 4, 1, 4>), grid, block, 0, s, a);
  } else if (tile == 32) {
    dim3 grid(cdiv(a.M, 256), a.cout_pad / 32, phases);
    hipLaunchKernelGGL((conv_igemm_kernel<T, OutT, 2, 1, 4>), grid, block, 0, s, a);
  } else {
    dim3 grid(cdiv(a.M, 256), a.cout_pad / 16, phases);
    hipLaunchKernelGGL((conv_igemm_kernel<T, OutT, 1, 1, 4>), grid, block, 0, s, a);
  }
  FN2_CHECK_LAUNCH("conv_igemm");
  return FN2_OK;
}

// Second half of a flow head computed as a GEMM: fn2_conv2d first runs the head as a 1x1 convolution with
// 18 outputs, t[pix][tap*2 + co] = sum_ci x[pix][ci] * w[tap][ci][co] (the activations are read ONCE, on the
// matrix cores, instead of nine times by the dot-product kernel); this kernel adds the nine shifted partials:
//   out[n,y,x,co] = bias[co] + sum_{ky,kx} t[n, y+ky-1, x+kx-1][(ky*3+kx)*2 + co]    (zero outside the image)
// Small heads (the 6x8 .. 96x128 levels: a handful of blocks, latency-bound): one thread per output pixel, nine
// 8-byte loads each.
__global__ void __launch_bounds__(256) flow_head_gather_small_kernel(const float* __restrict__ t, int t_cs,
                                                                     const float* __restrict__ bias,
                                                                     float* __restrict__ out, int N, int H, int W) {
  const long total = (long)N * H * W;
  const float b0 = bias ? bias[0] : 0.f, b1 = bias ? bias[1] : 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % W), y = (int)((i / W) % H);
    float a0 = b0, a1 = b1;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = y + ky - 1;
      if (iy < 0 || iy >= H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = x + kx - 1;
        if (ix < 0 || ix >= W) continue;
        const float2 v = *reinterpret_cast<const float2*>(t + (i + (long)(ky - 1) * W + (kx - 1)) * t_cs + (ky * 3 + kx) * 2);
        a0 += v.x; a1 += v.y;
      }
    }
    *reinterpret_cast<float2*>(out + 2 * i) = make_float2(a0, a1);
  }
}

// One block = a GT_Y x GT_X tile of output pixels of one image: the (GT_Y+2) x (GT_X+2) halo of partials goes
// through LDS with coalesced 72-byte runs per pixel, every lane then sums its nine taps from LDS (lane stride 72 B:
// conflict-free for ds_read_b64).  The first form (one thread per pixel, nine 8-byte loads at 128-byte lane stride
// straight from memory) ran at 0.5 TB/s on big heads: 0.122 ms for FlowNet2's full-resolution one, now 0.029 ms.
constexpr int GT_X = 64, GT_Y = 8;
__global__ void __launch_bounds__(256) flow_head_gather_kernel(const float* __restrict__ t, int t_cs,
                                                               const float* __restrict__ bias, float* __restrict__ out,
                                                               int N, int H, int W) {
  constexpr int HX = GT_X + 2, HY = GT_Y + 2;
  __shared__ float2 sm[HY * HX * 9];
  const int tiles_x = (W + GT_X - 1) / GT_X, tiles_y = (H + GT_Y - 1) / GT_Y;
  const float b0 = bias ? bias[0] : 0.f, b1 = bias ? bias[1] : 0.f;
  const int tid = threadIdx.x;
  for (int blk = blockIdx.x; blk < N * tiles_y * tiles_x; blk += gridDim.x) {
    const int tx = blk % tiles_x, ty = (blk / tiles_x) % tiles_y, n = blk / (tiles_x * tiles_y);
    const int x0 = tx * GT_X - 1, y0 = ty * GT_Y - 1;
    for (int idx = tid; idx < HY * HX * 9; idx += 256) {
      const int p = idx / 9, j = idx - p * 9;
      const int hy = p / HX, hx = p - hy * HX;
      const int gy = y0 + hy, gx = x0 + hx;
      float2 v = make_float2(0.f, 0.f);
      if (gy >= 0 && gy < H && gx >= 0 && gx < W)
        v = *reinterpret_cast<const float2*>(t + (((long)n * H + gy) * W + gx) * t_cs + 2 * j);
      sm[idx] = v;
    }
    __syncthreads();
    const int px = tid & (GT_X - 1);
#pragma unroll
    for (int r = 0; r < GT_Y / 4; ++r) {
      const int py = (tid >> 6) + 4 * r;
      const int oy = ty * GT_Y + py, ox = tx * GT_X + px;
      if (oy < H && ox < W) {
        float a0 = b0, a1 = b1;  // same summation order as the reference order of taps: ky outer, kx inner
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const float2 v = sm[((py + ky) * HX + px + kx) * 9 + ky * 3 + kx];
            a0 += v.x; a1 += v.y;
          }
        *reinterpret_cast<float2*>(out + (((long)n * H + oy) * W + ox) * 2) = make_float2(a0, a1);
      }
    }
    __syncthreads();
  }
}

// Where a tail reads its partials: one tensor [pixel][t_cs], or the `nslab` raw split-K slabs of the GEMM that made them
// (slab stride in floats; their sum, in split order, times `scale` is what the finalize pass would have stored).
struct TailSrc { const float* t; int t_cs; int nslab; long slab; float scale; };

// Flow-head tail in ONE launch: the gather above generalised to NT x NT taps (3: predict_flowN; 5: a linear interconvN
// composed with its predict_flowN, see fn2_flow_head_tail in flownet2_hip.h) and followed, in the same block, by
// upsample_flowXtoY (4x4 stride-2 transposed conv on the two flow channels, flownet_s.py:60-63) into the 2-channel
// slice of the next concat buffer.  A block owns TY x TX flow pixels: the partials of the tile + (NT/2 + UP) halo go
// through LDS in coalesced runs, the flow is formed on the tile + UP halo (the transposed conv reads a 1-pixel
// neighbourhood; halo pixels are recomputed by the neighbouring blocks rather than exchanged), the tile itself is
// written as fp32 [n,h,w,2], and the 2TY x 2TX upsampled pixels leave as one (hi, lo) pair of 4-byte stores each.
// `ring`: the image's outermost pixel ring was written to `pf` beforehand by head_ring_kernel (a composed head's
// border pixels use other weights) -- it is read from there instead of being gathered.
// Tap order of the sums = the existing kernels': ky outer, kx inner from the bias; ty outer, tx inner for the upsample.
template <int NT, int TY, int TX, bool UP, typename OutT>
__global__ void __launch_bounds__(256) head_tail_kernel(const TailSrc ts, const float* __restrict__ bias,
                                                        float* __restrict__ pf, int N, int H, int W, int ring,
                                                        const float* __restrict__ up_w, const float* __restrict__ up_bias,
                                                        OutT* __restrict__ up, int up_cs, int up_c0) {
  constexpr int R = NT / 2, E = UP ? 1 : 0, HALO = R + E;
  constexpr int RW = TX + 2 * HALO, RH = TY + 2 * HALO, NTAP = NT * NT;
  constexpr int EW = TX + 2 * E, EH = TY + 2 * E;
  __shared__ float2 sm[RH * RW * NTAP];
  __shared__ float2 spf[EH * EW];
  __shared__ float sw[64];
  const int tid = threadIdx.x;
  if (UP && tid < 64) sw[tid] = up_w[tid];  // [ky][kx][o][i]
  const int tiles_x = (W + TX - 1) / TX, tiles_y = (H + TY - 1) / TY;
  const float b0 = bias ? bias[0] : 0.f, b1 = bias ? bias[1] : 0.f;
  for (int blk = blockIdx.x; blk < N * tiles_y * tiles_x; blk += gridDim.x) {
    const int tx = blk % tiles_x, ty = (blk / tiles_x) % tiles_y, n = blk / (tiles_x * tiles_y);
    const int x0 = tx * TX - HALO, y0 = ty * TY - HALO;
    for (int idx = tid; idx < RH * RW * NTAP; idx += 256) {
      const int p = idx / NTAP, j = idx - p * NTAP;
      const int hy = p / RW, hx = p - hy * RW;
      const int gy = y0 + hy, gx = x0 + hx;
      float2 v = make_float2(0.f, 0.f);
      if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
        const float* src = ts.t + (((long)n * H + gy) * W + gx) * ts.t_cs + 2 * j;
        v = *reinterpret_cast<const float2*>(src);
        for (int sl = 1; sl < ts.nslab; ++sl) {   // raw split-K slabs of the head GEMM (fn2_conv_desc.raw_partials): summed in split order
          const float2 u = *reinterpret_cast<const float2*>(src + sl * ts.slab);
          v.x += u.x; v.y += u.y;
        }
        v.x *= ts.scale; v.y *= ts.scale;
      }
      sm[idx] = v;
    }
    __syncthreads();
    // flow on the tile + E halo
    for (int e = tid; e < EH * EW; e += 256) {
      const int ey = e / EW, ex = e - ey * EW;
      const int oy = ty * TY - E + ey, ox = tx * TX - E + ex;
      float a0 = 0.f, a1 = 0.f;
      if (oy >= 0 && oy < H && ox >= 0 && ox < W) {
        const bool inner = ey >= E && ey < E + TY && ex >= E && ex < E + TX;  // this block's own pixels
        float2* po = reinterpret_cast<float2*>(pf + (((long)n * H + oy) * W + ox) * 2);
        if (ring && (oy == 0 || oy == H - 1 || ox == 0 || ox == W - 1)) {
          const float2 v = *po;
          a0 = v.x; a1 = v.y;
        } else {
          a0 = b0; a1 = b1;
#pragma unroll
          for (int ky = 0; ky < NT; ++ky)
#pragma unroll
            for (int kx = 0; kx < NT; ++kx) {
              const float2 v = sm[((ey + ky) * RW + ex + kx) * NTAP + ky * NT + kx];
              a0 += v.x; a1 += v.y;
            }
          if (inner) *po = make_float2(a0, a1);
        }
      }
      spf[e] = make_float2(a0, a1);  // zeros outside the image: the transposed conv has no taps there
    }
    if constexpr (UP) {
      __syncthreads();
      const float ub0 = up_bias ? up_bias[0] : 0.f, ub1 = up_bias ? up_bias[1] : 0.f;
      for (int o = tid; o < 4 * TY * TX; o += 256) {
        const int uy = o / (2 * TX), ux = o - uy * (2 * TX);
        const int oy = 2 * ty * TY + uy, ox = 2 * tx * TX + ux;
        if (oy >= 2 * H || ox >= 2 * W) continue;
        const int a = uy & 1, b = ux & 1, y = uy >> 1, x = ux >> 1;   // tile-local flow pixel
        float r0 = ub0, r1 = ub1;
#pragma unroll
        for (int tty = 0; tty < 2; ++tty) {
          const int iy = y - 1 + a + tty, ky = 3 - a - 2 * tty;
          const int gy = ty * TY + iy;
          if (gy < 0 || gy >= H) continue;
#pragma unroll
          for (int ttx = 0; ttx < 2; ++ttx) {
            const int ix = x - 1 + b + ttx, kx = 3 - b - 2 * ttx;
            const int gx = tx * TX + ix;
            if (gx < 0 || gx >= W) continue;
            const float2 v = spf[(iy + E) * EW + ix + E];
            const float* ww = sw + (ky * 4 + kx) * 4;
            r0 += v.x * ww[0] + v.y * ww[1];
            r1 += v.x * ww[2] + v.y * ww[3];
          }
        }
        OutT* po = up + (((size_t)n * 2 * H + oy) * 2 * W + ox) * up_cs + up_c0;
        if constexpr (is_x2<OutT>::value) {
          // channels (c0, c0 + 1) of one 8-channel group, c0 even: the two hi halves are 4 bytes, the two lo halves too
          const size_t addr = reinterpret_cast<size_t>(po);
          _Float16* g = reinterpret_cast<_Float16*>(addr & ~size_t(31));
          const int j = (int)((addr & 31) >> 2);
          typedef __attribute__((ext_vector_type(2))) _Float16 h2;
          const _Float16 h0 = (_Float16)r0, h1 = (_Float16)r1;
          *reinterpret_cast<h2*>(g + j) = h2{h0, h1};
          *reinterpret_cast<h2*>(g + 8 + j) = h2{(_Float16)(r0 - (float)h0), (_Float16)(r1 - (float)h1)};
        } else {
          store_elem<OutT>(po, r0);
          store_elem<OutT>(po + 1, r1);
        }
      }
    }
    __syncthreads();
  }
}

// Border ring of a COMPOSED head (fn2_flow_head_tail, taps = 5).  With ic = interconv(x) (3x3, pad 1, bias b1, linear) and
// pf = predict_flow(ic) (3x3, pad 1, bias b2, linear), the reference zero-pads ic before the second conv
// (flownet_sd.py:60-63: pad(interconv)), so for a pixel p on the image's outermost ring the taps t of the second conv
// with p + t outside the image must NOT see interconv evaluated there.  The composed 5x5 weights therefore depend on
// which borders p touches: case = 3 * cy + cx, c in {0: low border, 1: interior, 2: high border}, and
//   pf(p) = bc[case] + sum_{u in 5x5, ci} wc[case][u][ci][o] x(p + u)[ci]        (x zero outside the image)
// One wave per ring pixel, lanes stride the (tap, 8-channel group) items, shuffle reduction: the flow_head_kernel form.
// (Images with H or W < 3 make a pixel touch both borders of an axis; the host refuses those.)
__global__ void __launch_bounds__(256) head_ring_kernel(const x2_t* __restrict__ in, int in_cs, int in_c0, int cin_groups,
                                                        const float* __restrict__ wc, const float* __restrict__ bc,
                                                        float* __restrict__ pf, int N, int H, int W) {
  const int lane = threadIdx.x & 63;
  const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
  const int ring = 2 * W + 2 * (H - 2);
  const int nitems = 25 * cin_groups;
  const size_t case_stride = (size_t)nitems * 8 * 2;  // floats per case: [tap][ci][o]
  for (long m = wave; m < (long)N * ring; m += nwaves) {
    const int n = (int)(m / ring), r = (int)(m - (long)n * ring);
    int y, x;
    if (r < W) { y = 0; x = r; }
    else if (r < 2 * W) { y = H - 1; x = r - W; }
    else { const int q = r - 2 * W; y = 1 + (q >> 1); x = (q & 1) ? W - 1 : 0; }
    const int cy = y == 0 ? 0 : (y == H - 1 ? 2 : 1), cx = x == 0 ? 0 : (x == W - 1 ? 2 : 1);
    const int cs = 3 * cy + cx;
    const float* w = wc + (size_t)cs * case_stride;
    float a0 = 0.f, a1 = 0.f;
    for (int q = lane; q < nitems; q += 64) {
      const int tap = q / cin_groups, gi = q - tap * cin_groups;
      const int ky = tap / 5, kx = tap - ky * 5;
      const int iy = y + ky - 2, ix = x + kx - 2;
      if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
      const uint4* src = reinterpret_cast<const uint4*>(in + (((size_t)n * H + iy) * W + ix) * in_cs + in_c0 + gi * 8);
      float xv[8];
      join8(src[0], src[1], xv);
      const float* u = w + (size_t)q * 16;   // [ci 0..7][o 0..1]
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        a0 += xv[j] * u[2 * j];
        a1 += xv[j] * u[2 * j + 1];
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      a0 += __shfl_xor(a0, off, 64);
      a1 += __shfl_xor(a1, off, 64);
    }
    if (lane == 0)
      *reinterpret_cast<float2*>(pf + (((size_t)n * H + y) * W + x) * 2) = make_float2(a0 + bc[2 * cs], a1 + bc[2 * cs + 1]);
  }
}

// ---------------------------------------------------------------------------
// upsample_flowXtoY: 2 -> 2 channel transposed conv 4x4 s2 crop 1, linear (flownet_s.py:60-63; bias only in
// the FlowNet2 fusion net, flownet2.py:70-73, :86-89).  HBM-bound: one lane per output pixel.
// ---------------------------------------------------------------------------
template <typename OutT>
__global__ void __launch_bounds__(256) upsample_flow_kernel(const float* __restrict__ in,
                                                            const float* __restrict__ w, const float* __restrict__ bias,
                                                            OutT* __restrict__ out,
                                                            int N, int H, int W, int out_cs, int out_c0) {
  __shared__ float sw[64];
  if (threadIdx.x < 64) sw[threadIdx.x] = w[threadIdx.x];  // [ky][kx][o][i]
  __syncthreads();
  const float b0 = bias ? bias[0] : 0.f, b1 = bias ? bias[1] : 0.f;
  const long total = (long)N * 4 * H * W;
  for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
    const int ox = (int)(o % (2 * W));
    const int oy = (int)((o / (2 * W)) % (2 * H));
    const int n = (int)(o / (2 * W) / (2 * H));
    const int a = oy & 1, b = ox & 1, y = oy >> 1, x = ox >> 1;
    float r0 = b0, r1 = b1;
#pragma unroll
    for (int ty = 0; ty < 2; ++ty) {
      const int iy = y - 1 + a + ty, ky = 3 - a - 2 * ty;
      if (iy < 0 || iy >= H) continue;
#pragma unroll
      for (int tx = 0; tx < 2; ++tx) {
        const int ix = x - 1 + b + tx, kx = 3 - b - 2 * tx;
        if (ix < 0 || ix >= W) continue;
        const float2 v = *reinterpret_cast<const float2*>(in + (((long)n * H + iy) * W + ix) * 2);
        const float* ww = sw + (ky * 4 + kx) * 4;
        r0 += v.x * ww[0] + v.y * ww[1];
        r1 += v.x * ww[2] + v.y * ww[3];
      }
    }
    OutT* po = out + (size_t)o * out_cs + out_c0;
    store_elem<OutT>(po, r0);
    store_elem<OutT>(po + 1, r1);
  }
}

// images fp32 [n,h,w,3] (x NIMG) -> the first 8 channels [img0 rgb | img1 rgb | 0 0] (NIMG = 2) or
// [rgb | 0 x5] (NIMG = 1) of the interior of a view padded by `pad` pixels: one lane per pixel, one
// 8-channel group store (the zero channels are the buffer's own padding channels).
template <typename OutT, int NIMG>
__global__ void __launch_bounds__(256) pack_image_kernel(const float* __restrict__ img0, const float* __restrict__ img1,
                                                         OutT* __restrict__ out, int n, int h, int w, int n0, int pad,
                                                         int out_cs, int c0) {
  const long npix = (long)n * h * w;
  const int hp = h + 2 * pad, wp = w + 2 * pad;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % w), y = (int)((i / w) % h), nn = (int)(i / w / h);
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const rgb3_t p0 = load_rgb(img0 + i * 3);  // one 12-byte load per pixel (fn2_common.h)
    v[0] = p0.r; v[1] = p0.g; v[2] = p0.b;
    if (NIMG == 2) { const rgb3_t p1 = load_rgb(img1 + i * 3); v[3] = p1.r; v[4] = p1.g; v[5] = p1.b; }
    OutT* d = out + (((size_t)(n0 + nn) * hp + y + pad) * wp + x + pad) * out_cs + c0;
    if constexpr (is_x2<OutT>::value) {
      uint4* q = reinterpret_cast<uint4*>(d);
      split8(v, q[0], q[1]);
    } else {
      store_vec<OutT, 4>(d, v);
      store_vec<OutT, 4>(d + 4, v + 4);
    }
  }
}

// Space-to-depth image packing for a stride-2 stem: out[n, sy, sx, (py*2+px)*4 + c] = padded_img[2sy+py, 2sx+px, c]
// (c < 3, 4th channel 0), padded_img = img with a zero border of `pad` pixels.  A k x k stride-2 convolution
// on the image is then a ceil(k/2) x ceil(k/2) stride-1 convolution on these 16-channel super-pixels: a kernel
// row is a run of 4 x 16 channels = whole 128-byte lines with NO padding taps (7x7x3: 8 stages instead of 14).
// One lane per super-pixel, border super-pixels included (the buffer needs no pre-zeroing).
template <typename OutT>
__global__ void __launch_bounds__(256) pack_image_s2d_kernel(const float* __restrict__ img, OutT* __restrict__ out,
                                                             int n, int h, int w, int n0, int pad, int out_cs) {
  const int hs = (h + 2 * pad) / 2, ws = (w + 2 * pad) / 2;
  const long total = (long)n * hs * ws;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int sx = (int)(i % ws), sy = (int)((i / ws) % hs), nn = (int)(i / ws / hs);
    float v[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int y = 2 * sy + (q >> 1) - pad, x = 2 * sx + (q & 1) - pad;
      const bool ok = y >= 0 && y < h && x >= 0 && x < w;
      const float* p = img + (((long)nn * h + (ok ? y : 0)) * w + (ok ? x : 0)) * 3;
      const rgb3_t px = load_rgb(p);  // one 12-byte load per pixel (fn2_common.h)
      v[q * 4 + 0] = ok ? px.r : 0.f; v[q * 4 + 1] = ok ? px.g : 0.f; v[q * 4 + 2] = ok ? px.b : 0.f; v[q * 4 + 3] = 0.f;
    }
    OutT* d = out + (((size_t)(n0 + nn) * hs + sy) * ws + sx) * out_cs;
    if constexpr (is_x2<OutT>::value) {
      uint4* qd = reinterpret_cast<uint4*>(d);
      split8(v, qd[0], qd[1]);
      split8(v + 8, qd[2], qd[3]);
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) store_vec<OutT, 4>(d + 4 * k, v + 4 * k);
    }
  }
}

static inline int grid_for(long work_items, int block) {
  long g = (work_items + block - 1) / block;
  if (g > 256L * 16) g = 256L * 16;
  if (g < 1) g = 1;
  return (int)g;
}

static int check_view(const fn2_tensor* t, const char* what) {
  FN2_REQUIRE(t && t->data, "%s: null tensor", what);
  FN2_REQUIRE(t->dtype >= FN2_F32 && t->dtype <= FN2_F16X2, "%s: bad dtype", what);
  FN2_REQUIRE(t->n >= 1 && t->h >= 1 && t->w >= 1 && t->c >= 1, "%s: bad dims", what);
  FN2_REQUIRE(t->c0 >= 0 && t->c0 + t->c <= t->cs, "%s: channel slice outside the buffer", what);
  return FN2_OK;
}

template <int NIMG>
static int pack_imgs(const float* i0, const float* i1, const fn2_tensor* out, int n, int n0, int pad, void* stream) {
  FN2_REQUIRE(pad >= 0 && out->h > 2 * pad && out->w > 2 * pad, "pack: bad border");
  FN2_REQUIRE(out->c0 % 8 == 0 && out->cs % 8 == 0 && out->c0 + 8 <= out->cs, "pack: needs an 8-channel aligned slot");
  const int h = out->h - 2 * pad, w = out->w - 2 * pad;
  const long npix = (long)n * h * w;
  const dim3 g(grid_for(npix, 256)), b(256);
  hipStream_t s = (hipStream_t)stream;
  if (out->dtype == FN2_F32)
    hipLaunchKernelGGL((pack_image_kernel<float, NIMG>), g, b, 0, s, i0, i1, (float*)out->data, n, h, w, n0, pad, out->cs, out->c0);
  else if (out->dtype == FN2_F16X2)
    hipLaunchKernelGGL((pack_image_kernel<x2_t, NIMG>), g, b, 0, s, i0, i1, (x2_t*)out->data, n, h, w, n0, pad, out->cs, out->c0);
  else if (out->dtype == FN2_BF16)
    hipLaunchKernelGGL((pack_image_kernel<bf16_t, NIMG>), g, b, 0, s, i0, i1, (bf16_t*)out->data, n, h, w, n0, pad, out->cs, out->c0);
  else
    hipLaunchKernelGGL((pack_image_kernel<f16_t, NIMG>), g, b, 0, s, i0, i1, (f16_t*)out->data, n, h, w, n0, pad, out->cs, out->c0);
  FN2_CHECK_LAUNCH("pack_image");
  return FN2_OK;
}

template <int NT, int TY, int TX, typename OutT>
void launch_head_tail(bool up, const TailSrc ts, const float* bias, float* pf, int n, int h, int w, int ring,
                      const float* up_w, const float* up_bias, void* up_data, int up_cs, int up_c0, hipStream_t s) {
  const long tiles = (long)n * ((h + TY - 1) / TY) * ((w + TX - 1) / TX);
  const dim3 grid((unsigned)std::min<long>(tiles, 1 << 16)), block(256);
  if (up)
    hipLaunchKernelGGL((head_tail_kernel<NT, TY, TX, true, OutT>), grid, block, 0, s, ts, bias, pf, n, h, w, ring, up_w,
                       up_bias, (OutT*)up_data, up_cs, up_c0);
  else
    hipLaunchKernelGGL((head_tail_kernel<NT, TY, TX, false, float>), grid, block, 0, s, ts, bias, pf, n, h, w, ring,
                       (const float*)nullptr, (const float*)nullptr, (float*)nullptr, 0, 0);
}
template <typename OutT>
void launch_head_tail_t(int taps, bool up, const TailSrc ts, const float* bias, float* pf, int n, int h, int w,
                        int ring, const float* up_w, const float* up_bias, void* up_data, int up_cs, int up_c0,
                        hipStream_t s) {
  // big maps: 8-row tiles (less halo per flow pixel); small ones: 4 x 32 tiles so that a 48 x 64 level still has ~100 blocks
  const long big = (long)n * ((h + 7) / 8) * ((w + 63) / 64);
  if (taps == 3) {
    if (big >= 256) launch_head_tail<3, 8, 64, OutT>(up, ts, bias, pf, n, h, w, ring, up_w, up_bias, up_data, up_cs, up_c0, s);
    else launch_head_tail<3, 4, 32, OutT>(up, ts, bias, pf, n, h, w, ring, up_w, up_bias, up_data, up_cs, up_c0, s);
  } else {
    if (big >= 256) launch_head_tail<5, 8, 32, OutT>(up, ts, bias, pf, n, h, w, ring, up_w, up_bias, up_data, up_cs, up_c0, s);
    else launch_head_tail<5, 4, 32, OutT>(up, ts, bias, pf, n, h, w, ring, up_w, up_bias, up_data, up_cs, up_c0, s);
  }
}

}  // namespace fn2

using namespace fn2;

extern "C" {

int fn2_conv2d_plan(int in_dtype, int cin_pad, int cout, fn2_conv_plan* plan) {
  FN2_REQUIRE(plan, "conv2d_plan: null plan");
  FN2_REQUIRE(in_dtype >= FN2_F32 && in_dtype <= FN2_F16X2, "conv2d_plan: bad dtype");
  FN2_REQUIRE(cin_pad > 0 && cin_pad % 8 == 0 && cout >= 1, "conv2d_plan: cin_pad must be a positive multiple of 8");
  const int esz = dtype_size(in_dtype);
  if (cout == 2) {  // flow heads: dedicated dot-product kernel reading rows 0 and 1 of a natural-order weight
    plan->layout = 0; plan->cout_tile = 16; plan->kstep_elems = 64 / esz;
    plan->wgt_dtype = in_dtype == FN2_F16X2 ? FN2_F32 : in_dtype;
    return FN2_OK;
  }
  if (conv_fast_ok(in_dtype, cin_pad, cout)) {
    plan->layout = 1; plan->cout_tile = cout > 64 ? 128 : cout > 32 ? 64 : 32; plan->kstep_elems = 128 / esz;
    plan->wgt_dtype = in_dtype;
    return FN2_OK;
  }
  if (in_dtype == FN2_F16X2)
    return fail(FN2_ERR_UNSUPPORTED, "split-fp16 inputs need cin_pad %% 32 == 0 (LDS-DMA kernel); got %d", cin_pad);
  plan->layout = 0; plan->cout_tile = cout > 64 ? 128 : cout > 32 ? 64 : cout > 16 ? 32 : 16;
  plan->kstep_elems = 64 / esz; plan->wgt_dtype = in_dtype;
  return FN2_OK;
}

}  // extern "C"

namespace fn2 {

static bool is_flow_head(const fn2_conv_desc* d) {
  return d->kind == 0 && d->out.c == 2 && d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1 &&
         d->out.dtype == FN2_F32 && d->act == FN2_ACT_NONE;
}
// Cout == 2 layers that are not 3x3/s1/p1 fp32-out heads run on the generic kernel (16-cout tile)

// Block tiles of the LDS-DMA kernel are HALF as wide in pixels as the first version's (128 x 64, 64 x 128,
// 32 x 128 instead of 128 x 128, 64 x 256, 32 x 256): 40-48 KB of LDS = three or four resident blocks per CU instead
// of two.  Measured (tools/ab_conv.py, tools/ab_e2e.sh on one box): large 128-cout layers +1..13 % (the extra
// resident block covers more of the DMA / barrier waits than the extra weight fetches cost), 24x32 / 12x16 levels
// +17-19 % (no split-K slabs, no finalize launch), 64- and 32-cout layers +10..17 %; FlowNetC b8 -3.5 %,
// FlowNet2 b4 -8 % end to end.  Exception: 128-cout layers whose half-width grid is still under 96 blocks (the 6x8
// level) -- there the split-K count is what matters and the narrow tile loses 10 % (at 12x16, 96 blocks: +14 %).  FN2_CONV_DBG bit 32 = the
// full-width tiles (A/B).
static bool wants_bp64(const ConvArgs& a, int tile, int phases, int layout) {
  if (layout != 1 || (a.dbg & 32)) return false;
  if (tile < 128) return true;
  const char* e = getenv("FN2_BP64_MIN");  // tuning knob of the experiments behind the default
  const int min_blocks = e ? atoi(e) : 96;
  return cdiv(a.M, 64) * (long)(a.cout_pad / 128) * phases >= min_blocks;
}

// validate + fill everything except the split-K fields
static int build_args(const fn2_conv_desc* d, ConvArgs* out, int* tile_out, int* phases_out) {
  FN2_REQUIRE(d, "conv2d: null descriptor");
  int rc = check_view(&d->in, "conv2d input");
  if (rc) return rc;
  rc = check_view(&d->out, "conv2d output");
  if (rc) return rc;
  FN2_REQUIRE(d->wgt, "conv2d: null weights");
  FN2_REQUIRE((d->kind >= 0 && d->kind <= 3) || d->kind == 5,
              "conv2d: kind must be 0 (conv), 1 (deconv k4 s2 crop 1), 2 (stem row-run conv), 3 (transpose of a stride-2 conv) "
              "or 5 (deconv k4 s2 crop 1, column phases merged)");
  FN2_REQUIRE(d->in.n == d->out.n, "conv2d: batch mismatch");
  FN2_REQUIRE(d->up_src == nullptr || ((d->kind == 1 || d->kind == 5) && d->up_w != nullptr && d->up_c0 >= 0 &&
                                       d->up_c0 + 2 <= d->out.cs && !d->accumulate),
              "conv2d: up_src rides on a kind-1 / kind-5 transposed conv (up_w set, [up_c0, up_c0 + 2) inside the out buffer)");
  FN2_REQUIRE(d->cin_pad % 8 == 0 && d->cin_pad >= d->in.c, "conv2d: cin_pad must be a multiple of 8 >= Cin");
  FN2_REQUIRE(d->in.cs % 8 == 0 && d->in.c0 % 8 == 0, "conv2d: input channel stride/offset must be multiples of 8");
  if (d->kind != 2)
    FN2_REQUIRE(d->in.c0 + d->cin_pad <= d->in.cs, "conv2d: padded input channels exceed the buffer stride");
  const int esz = dtype_size(d->in.dtype);
  const int CH = 16 / esz;
  FN2_REQUIRE(d->kpad > 0 && d->kpad % (4 * CH) == 0, "conv2d: kpad must be a multiple of one k-step");
  fn2_conv_plan plan;
  rc = fn2_conv2d_plan(d->in.dtype, d->cin_pad, d->out.c, &plan);
  if (rc) return rc;
  // wgt_layout 2 = layout 1's packed matrix re-tiled into MFMA-fragment order (conv2.hip, WREG): 128-cout split-fp16 layers
  const bool wfrag = d->wgt_layout == 2;
  if (wfrag)
    FN2_REQUIRE(plan.layout == 1 && plan.cout_tile >= 64 && d->in.dtype == FN2_F16X2 && d->kind != 2,
                "conv2d: wgt_layout 2 (fragment order) is for split-fp16 layers with more than 32 output channels");
  else
    FN2_REQUIRE(d->wgt_layout == plan.layout, "conv2d: wgt_layout %d does not match fn2_conv2d_plan (%d)", d->wgt_layout,
                plan.layout);
  const int tile = plan.cout_tile;
  FN2_REQUIRE(d->cout_pad % tile == 0 && d->cout_pad >= d->out.c, "conv2d: cout_pad must be a multiple of the cout tile");
  FN2_REQUIRE(d->act == FN2_ACT_NONE || d->act == FN2_ACT_LEAKY, "conv2d: bad activation");
  FN2_REQUIRE(d->out.dtype == d->in.dtype || d->out.dtype == FN2_F32 ||
                  (d->in.dtype == FN2_F32 && d->out.dtype == FN2_F16X2),
              "conv2d: output dtype must be the input dtype, fp32, or split-fp16 from an fp32 stem");
  if (d->in.dtype == FN2_F16X2) FN2_REQUIRE(d->in.cs % 8 == 0 && d->in.c0 % 8 == 0, "conv2d: split-fp16 views are group (8) aligned");
  if (d->out.dtype == FN2_F16X2) FN2_REQUIRE(d->out.cs % 8 == 0 && d->out.c0 % 8 == 0, "conv2d: split-fp16 views are group (8) aligned");

  ConvArgs& a = *out;
  a = ConvArgs{};   // (optional parts -- head blocks, ring blocks -- are off unless a caller sets them afterwards)
  a.KH_KW_hint = 0;
  a.wfrag = wfrag ? 1 : 0;   // (2 = the 3-slot ring form, chosen below)
  a.merged = d->kind == 5 ? 1 : 0;
  a.in = d->in.data; a.wgt = d->wgt; a.bias = d->bias; a.out = d->out.data;
  a.N = d->in.n; a.H = d->in.h; a.W = d->in.w; a.in_cs = d->in.cs; a.in_c0 = d->in.c0;
  a.cin_chunks = d->cin_pad / CH;
  int phases = 1;
  int trim[2] = {0, 0};  // kind 3: taps per phase parity (0 = all KH x KW)
  if (d->kind == 0) {
    FN2_REQUIRE(d->kh >= 1 && d->kw >= 1 && d->stride >= 1 && d->pad >= 0, "conv2d: bad kernel geometry");
    a.KH = d->kh; a.KW = d->kw; a.stride = d->stride; a.pad = d->pad;
    a.OH = (d->in.h + 2 * d->pad - d->kh) / d->stride + 1;  // VALID on the padded input
    a.OW = (d->in.w + 2 * d->pad - d->kw) / d->stride + 1;
    FN2_REQUIRE(a.OH >= 1 && a.OW >= 1, "conv2d: kernel does not fit");
    FN2_REQUIRE(d->out.h == a.OH && d->out.w == a.OW, "conv2d: output spatial size %dx%d != expected %dx%d",
                d->out.h, d->out.w, a.OH, a.OW);
    a.deconv = 0; a.ph_pad0 = a.ph_pad1 = 0;
  } else if (d->kind == 2) {
    // stem row-run conv: the kw taps x cs channels of a kernel row are one contiguous run of the
    // pre-padded NHWC input; runs past the row end read the next row / out-of-range zeros under zero weights
    FN2_REQUIRE(d->kh >= 1 && d->kw >= 1 && d->stride >= 1 && d->pad == 0, "stem conv: pad must be baked into the input buffer");
    FN2_REQUIRE(d->in.c0 == 0 && d->in.c == d->in.cs, "stem conv: the input view must be the whole buffer");
    FN2_REQUIRE(d->cin_pad >= d->kw * d->in.cs, "stem conv: run (cin_pad) shorter than kw*cs");
    FN2_REQUIRE(d->wgt_layout == 1, "stem conv: runs on the LDS-DMA kernel only (run must be whole 128-byte lines)");
    a.KH = d->kh; a.KW = 1; a.stride = d->stride; a.pad = 0;
    a.KH_KW_hint = d->kw;
    a.OH = (d->in.h - d->kh) / d->stride + 1;
    a.OW = (d->in.w - d->kw) / d->stride + 1;
    FN2_REQUIRE(a.OH >= 1 && a.OW >= 1, "stem conv: kernel does not fit");
    FN2_REQUIRE(d->out.h == a.OH && d->out.w == a.OW, "stem conv: output spatial size %dx%d != expected %dx%d",
                d->out.h, d->out.w, a.OH, a.OW);
    a.deconv = 0; a.ph_pad0 = a.ph_pad1 = 0;
  } else if (d->kind == 5) {
    // the same transposed convolution as kind 1 for 16 / 32 output channels on split fp16 (the fusion net's fuse_deconv0 /
    // fuse_deconv1, flownet2.py:66-84): blockIdx.z = output row phase a, both column phases in one block as a 2 x 3-tap
    // convolution with 2 Cout packed rows (weights.pack_deconv_merged)
    FN2_REQUIRE(d->kh == 4 && d->kw == 4 && d->stride == 2, "deconv (kind 5): only k=4 s=2 crop 1");
    FN2_REQUIRE((d->out.c == 16 || d->out.c == 32) && d->in.dtype == FN2_F16X2 && d->wgt_layout == 1 && d->cout_pad == 2 * d->out.c,
                "deconv (kind 5): 16 or 32 output channels, split-fp16 input, cout_pad = 2 Cout");
    FN2_REQUIRE(d->in.w % 128 == 0, "deconv (kind 5): input width must be a multiple of 128 (halo kernel tiles)");
    a.KH = 2; a.KW = 3; a.stride = 1; a.pad = 0;
    a.OH = d->in.h; a.OW = d->in.w;
    FN2_REQUIRE(d->out.h == 2 * d->in.h && d->out.w == 2 * d->in.w, "deconv: output must be 2H x 2W");
    a.deconv = 1;
    a.ph_pad0 = 1; a.ph_pad1 = 0;
    phases = 2;
  } else if (d->kind == 1) {
    FN2_REQUIRE(d->kh == 4 && d->kw == 4 && d->stride == 2, "deconv: only k=4 s=2 crop 1 (flownet_s.py:53-63)");
    // bias: nullptr inside the refinement scopes (biases_initializer=None, flownet_s.py:53); the FlowNet2 fusion
    // net's fuse_deconv1 / fuse_deconv0 carry one (flownet2.py:50-57 opens no such scope, :66-84)
    a.KH = 2; a.KW = 2; a.stride = 1; a.pad = 0;
    a.OH = d->in.h; a.OW = d->in.w;
    FN2_REQUIRE(d->out.h == 2 * d->in.h && d->out.w == 2 * d->in.w, "deconv: output must be 2H x 2W");
    a.deconv = 1;
    a.ph_pad0 = 1; a.ph_pad1 = 0;  // phase a reads rows m-1+a+t, t in {0,1}
    phases = 4;
  } else {
    // kind 3: gradient wrt the input of a stride-2 convolution (kernel k, pad p) = transposed convolution.
    // Output row 2m+a receives dY rows m+lo_a+t, t in [0,T), T = ceil(k/2), lo_a = ceil((a+p-k+1)/2); the
    // packed phase weight holds W[ky = a+p-2lo_a-2t] (zero where ky falls outside [0,k)).
    FN2_REQUIRE(d->stride == 2 && d->kh == d->kw && d->kh >= 1 && d->pad >= 0 && d->bias == nullptr && d->act == FN2_ACT_NONE,
                "conv-transpose (kind 3): stride 2, square kernel, no bias, no activation");
    const int T = (d->kh + 1) / 2;
    a.KH = T; a.KW = T; a.stride = 1; a.pad = 0;
    a.OH = d->in.h; a.OW = d->in.w;
    FN2_REQUIRE(d->out.h == 2 * d->in.h && d->out.w == 2 * d->in.w, "conv-transpose: output must be 2H x 2W");
    a.deconv = 1;
    auto ceil_half = [](int v) { return v >= 0 ? (v + 1) / 2 : -((-v) / 2); };
    a.ph_pad0 = -ceil_half(0 + d->pad - d->kh + 1);
    a.ph_pad1 = -ceil_half(1 + d->pad - d->kh + 1);
    phases = 4;
    // phase a holds ky = ky0, ky0 - 2, ... >= 0 with ky0 = a + p + 2 ph_pad_a: a prefix of its T tap slots
    const char* e_trim = getenv("FN2_TRIM_TAPS");  // 0: walk the zero slots as well (A/B)
    if (!e_trim || atoi(e_trim) != 0)
      for (int ph = 0; ph < 2; ++ph) {
        const int ky0 = ph + d->pad + 2 * (ph ? a.ph_pad1 : a.ph_pad0);
        const int t = ky0 < 0 ? 0 : ky0 / 2 + 1;
        trim[ph] = t < T ? t : T;
      }
  }
  for (int ph = 0; ph < 2; ++ph) { a.kh_ph[ph] = trim[ph] > 0 ? trim[ph] : a.KH; a.kw_ph[ph] = trim[ph] > 0 ? trim[ph] : a.KW; }
  a.accum = d->accumulate ? 1 : 0;
  a.mask_y = d->act_grad_y; a.mask_c0 = d->act_grad_c0; a.mask_c1 = d->act_grad_c1;
  if (a.mask_y != nullptr) {
    FN2_REQUIRE(d->out.dtype == FN2_F32 || d->out.dtype == FN2_F16X2, "conv2d: act_grad_y needs an fp32 or split-fp16 output");
    FN2_REQUIRE(d->act == FN2_ACT_NONE && d->kind != 2, "conv2d: act_grad_y goes with a linear gradient layer (kinds 0, 1, 3)");
    FN2_REQUIRE(0 <= a.mask_c0 && a.mask_c0 < a.mask_c1 && a.mask_c1 <= d->out.c, "conv2d: act_grad channel range outside the output view");
    FN2_REQUIRE(!is_flow_head(d), "conv2d: act_grad_y is not available on the flow-head path");
  }
  if (a.accum) FN2_REQUIRE(d->out.dtype == FN2_F32 || d->out.dtype == FN2_F16X2,
                           "conv2d: accumulate needs an fp32 or split-fp16 output");
  const long M = (long)a.N * a.OH * a.OW;
  FN2_REQUIRE(M < (1L << 31), "conv2d: too many output pixels");
  a.M = (int)M;
  a.out_H = d->out.h; a.out_W = d->out.w; a.out_cs = d->out.cs; a.out_c0 = d->out.c0; a.Cout = d->out.c;
  FN2_REQUIRE(d->kpad >= a.KH * a.KW * d->cin_pad, "conv2d: kpad smaller than taps*cin_pad");
  a.ksteps = d->kpad / (4 * CH);
  a.cout_pad = d->cout_pad;
  a.act = d->act;
  a.out_scale = d->out_scale == 0.f ? 1.f : d->out_scale;
  a.vec_ok = (d->out.cs % 4 == 0) && (d->out.c0 % 4 == 0);
  a.splitk = 1; a.kper = a.ksteps; a.ws = nullptr; a.ws_cs = (a.Cout + 3) / 4 * 4;
  { const char* e = getenv("FN2_CONV_DBG"); a.dbg = e ? atoi(e) : 0; }
  a.in_bytes = 0;
  if (d->wgt_layout >= 1) {
    // the LDS-DMA kernel addresses both operands through buffer descriptors with 32-bit byte offsets
    const long in_bytes = (long)d->in.n * d->in.h * d->in.w * d->in.cs * esz;
    FN2_REQUIRE(in_bytes < (1L << 31), "conv2d: input buffer >= 2 GiB is not addressable by the LDS-DMA kernel");
    FN2_REQUIRE((long)d->cout_pad * d->kpad * esz < (1L << 31), "conv2d: packed weight >= 2 GiB per phase");
    a.in_bytes = (int)in_bytes;
  }
  {
    // order of the (pixel tile, cout tile) bands over the XCDs: fabric bytes ~ A min(8, X) + B with the pixel-major
    // order (every XCD that holds a pixel tile pulls the weights), A + B min(8, Y) with the weight-major one
    const double A = (double)d->cout_pad * d->kpad * esz, B = (double)d->in.n * d->in.h * d->in.w * d->in.cs * esz;
    const long X = cdiv(M, 64), Y = cdiv(d->cout_pad, 128);
    const char* e = getenv("FN2_WMAJOR");
    const int mode = e ? atoi(e) : 2;
    a.wmajor = mode == 2 ? (A * (double)(X < 8 ? X : 8) + B > A + B * (double)(Y < 8 ? Y : 8)) : mode;
  }
  a.bp64 = wfrag ? 1 : wants_bp64(a, tile, phases, d->wgt_layout) ? 1 : 0;
  // experiment (bit 64): weight-streaming 128-cout layers (the ones left on 128 x 128 + split-K) on 128 x 64 tiles
  // with the 3-slot ring
  if ((a.dbg & 64) && d->wgt_layout == 1 && tile == 128 && !a.bp64) a.bp64 = 2;
  // (fragment-order weights, wgt_layout 2: always the plain 128 x 64 grid of the WREG kernel -- every choice below is for layout 1)
  if (wfrag) {
    // FN2_WREG_RING: 0 = two-stage loop; 1 (default) = the 3-slot ring on one-round grids (384 .. FN2_RING_MAX blocks, as
    // for layout 1); 2 = the ring everywhere
    const char* e_wr = getenv("FN2_WREG_RING");
    const char* e_ring = getenv("FN2_RING_MAX");
    const int mode = e_wr ? atoi(e_wr) : 1;
    const long blocks = (long)cdiv(a.M, 64) * (a.cout_pad / 128) * phases;
    if (tile == 128 && (mode == 2 || (mode == 1 && blocks >= 384 && blocks <= (e_ring ? atoi(e_ring) : 512)))) a.wfrag = 2;
  }
  // 128 x 64 layers whose grid is at most two blocks per CU and takes no split-K (384..512 blocks: conv4_1 / deconv3
  // at batch 8, conv3 / conv3_1 at batch 4): the 3-slot ring's 72 KB of LDS costs them no resident block and keeps
  // two stages of DMA in flight instead of one (+4..7 %; with a third resident block to lose it is 5-12 % slower).
  // FN2_RING_MAX = largest such grid (0 = never); FN2_CONV_DBG bit 256 = ring on every 128 x 64 layer (A/B).
  // (the same rule on the 64- and 32-cout tiles measured neutral: not instantiated)
  // K groups (conv2.hip): 128-cout layers whose 128 x 64 grid is under 96 blocks -- the 6x8 / 12x16 levels, which used
  // 128 x 128 tiles + up to 16 partial-sum slabs before -- run two groups of four waves per block (in-block split-K)
  // and half the slabs: conv5..deconv5 10-18 % faster, FlowNet2 b4 -2 % end to end.  Measured and NOT taken: groups on
  // the 96..256-block layers (conv4, conv4_1, deconv3/4 at batch 4).  The groups of a block share its barriers, so
  // they issue their DMA pieces together and compute together; separate split-K blocks drift apart and overlap:
  // conv4_1 200 -> 157 TFLOP/s, deconv4 161 -> 83 with groups instead of splits.  Three groups (144 KB of LDS) lose
  // to two on the small levels as well.  FN2_KG=0 switches groups off (A/B); FN2_KG3_MAX / FN2_KG2_MAX = largest
  // grids that take 3 / 2 groups.
  a.kg = 1;
  if (d->wgt_layout == 1 && tile == 128 && !(a.dbg & 32)) {
    const char* e_kg = getenv("FN2_KG");
    if (!e_kg || atoi(e_kg) != 0) {
      const char* e3 = getenv("FN2_KG3_MAX");
      const char* e2 = getenv("FN2_KG2_MAX");
      const long blocks = (long)cdiv(a.M, 64) * (a.cout_pad / 128) * phases;
      if (blocks <= (e3 ? atoi(e3) : 0)) { a.bp64 = 1; a.kg = 3; }
      else if (blocks <= (e2 ? atoi(e2) : 95)) { a.bp64 = 1; a.kg = 2; }
    }
  }
  // Deep ring (conv2.hip, STAGES = 6): 128 x 64 grids of about one block per CU or less.  A single resident block has no
  // neighbour to hide its DMA round trips, so the two-stage loop runs at one round trip per stage there and the layer
  // was split over K to get more blocks (slabs + a finalize launch); five stages of DMA in flight per block stream at the
  // L2 -> LDS rate instead.  FN2_DEEP_MIN / FN2_DEEP_MAX = the grid sizes (blocks) that take it.
  if (d->wgt_layout == 1 && tile == 128 && !(a.dbg & 32) && d->in.dtype == FN2_F16X2) {
    const char* e_lo = getenv("FN2_DEEP_MIN");
    const char* e_hi = getenv("FN2_DEEP_MAX");
    const long blocks = (long)cdiv(a.M, 64) * (a.cout_pad / 128) * phases;
    const int lo = e_lo ? atoi(e_lo) : 0, hi = e_hi ? atoi(e_hi) : -1;
    if (blocks >= lo && blocks <= hi) { a.bp64 = 3; a.kg = 1; }
  }
  if (d->wgt_layout == 1 && tile == 128 && a.bp64 == 1 && a.kg == 1) {
    const char* e_ring = getenv("FN2_RING_MAX");  // tuning knob (read per launch: tools/ab_conv.py toggles it in-process)
    const int ring_max = e_ring ? atoi(e_ring) : 512;
    const long blocks = (long)cdiv(a.M, 64) * (a.cout_pad / 128) * phases;
    if ((a.dbg & 256) || (blocks >= 384 && blocks <= ring_max)) a.bp64 = 2;
    // experiment (FN2_RING_SPLIT = smallest grid): split-K layers on the ring as well, split so that the grid is one round
    // of two resident blocks per CU (512 slots) instead of three two-stage blocks (640 slots)
    const char* e_rs = getenv("FN2_RING_SPLIT");
    if (e_rs && blocks >= atoi(e_rs) && blocks < 384) a.bp64 = 2;
  }
  *tile_out = tile;
  *phases_out = phases;
  return FN2_OK;
}

// Preferred split-K factor: fill >= ~2 blocks per CU on layers whose output grid is small
// (the 6x8 .. 24x32 resolution layers: weight-bandwidth bound, SURVEY.md section 7 "hard parts").
static int preferred_split(const ConvArgs& a, int tile, int phases) {
  if (a.merged) return 1;  // kind 5 runs on the halo kernel: no K split
  const int bp = tile == 128 ? (a.bp64 ? 64 : 128) : (a.bp64 ? 128 : 256);
  const long blocks = (long)cdiv(a.M, bp) * (a.cout_pad / tile) * phases;
  const char* e_min = getenv("FN2_SPLIT_MINBLOCKS");  // tuning knob: grids from this many blocks up take no split-K
  if (blocks >= (e_min ? atoi(e_min) : 384)) return 1;
  if (a.bp64 == 3) {
    // deep ring: one block per CU streams by itself; split only to reach about FN2_DEEP_SLOTS blocks
    const char* e_sl = getenv("FN2_DEEP_SLOTS");
    int s = (int)((e_sl ? atoi(e_sl) : 256) / blocks);
    const int maxs = a.ksteps / 16;
    if (s > maxs) s = maxs;
    if (s > 16) s = 16;
    if (s < 2) return 1;
    const int kper = cdiv(a.ksteps, s);
    return cdiv(a.ksteps, kper);
  }
  if (a.kg > 1) {
    // K-group blocks hold a CU each (96 / 144 KB of LDS): one round = 256 blocks; a group wants >= 6 stages
    const char* e_nos = getenv("FN2_KG_NOSPLIT");  // grids from this many blocks up take no split
    if (blocks >= (e_nos ? atoi(e_nos) : 96)) return 1;
    const char* e_sl = getenv("FN2_KG_SLOTS");
    int s = (int)((e_sl ? atoi(e_sl) : 256) / blocks);
    const int maxs = a.ksteps / (12 * a.kg);
    if (s > maxs) s = maxs;
    if (s > 16) s = 16;
    if (s < 2) return 1;
    const int kper = cdiv(a.ksteps, s);
    return cdiv(a.ksteps, kper);
  }
  // as many splits as still give ONE round of resident blocks (2-3 per CU): rounding up put 528 blocks on 512
  // slots for the 12x16-level layers and a second round of 16 stragglers doubled the kernel time
  const char* e = getenv("FN2_SPLIT_SLOTS");  // tuning knob of the experiments behind the default
  const char* e_rsl = getenv("FN2_RING_SPLIT_SLOTS");
  const int slots = (tile == 128 && a.bp64 == 2) ? (e_rsl ? atoi(e_rsl) : 512) : e ? atoi(e) : 640;  // (512 until the slab stores were coalesced; FlowNet2 b4: 512 -> 4.29, 640 -> 4.24, 768 / 1024 -> 4.31 / 4.33 ms)
  int s = (int)(slots / blocks);
  const int maxs = a.ksteps / 8;
  if (s > maxs) s = maxs;
  if (s > 16) s = 16;
  if (s < 2) return 1;
  const int kper = cdiv(a.ksteps, s);
  return cdiv(a.ksteps, kper);  // no empty splits
}

static int64_t split_bytes(const ConvArgs& a, int s) {
  return (int64_t)s * a.N * a.out_H * a.out_W * a.ws_cs * (int64_t)sizeof(float);
}

}  // namespace fn2

extern "C" {

int64_t fn2_conv2d_workspace_bytes(const fn2_conv_desc* d) {
  ConvArgs a;
  int tile, phases;
  if (build_args(d, &a, &tile, &phases) != FN2_OK) return 0;
  if (is_flow_head(d)) return 0;
  const int s = preferred_split(a, tile, phases);
  return s > 1 ? split_bytes(a, s) : 0;
}

int fn2_conv2d_splits(const fn2_conv_desc* d) {
  ConvArgs a;
  int tile, phases;
  if (build_args(d, &a, &tile, &phases) != FN2_OK || is_flow_head(d)) return 1;
  int sk = preferred_split(a, tile, phases);
  while (sk > 1 && (d->workspace == nullptr || split_bytes(a, sk) > d->workspace_bytes)) --sk;
  if ((long)a.N * a.out_H * a.out_W * a.ws_cs >= (1L << 31)) sk = 1;
  if (sk > 1) {   // as fn2_conv2d rounds it: whole stages, no empty split
    const int kper = cdiv(a.ksteps, sk);
    sk = cdiv(a.ksteps, kper);
    if (d->wgt_layout >= 1) sk = cdiv(a.ksteps / 2, cdiv(kper, 2));
  }
  return sk < 1 ? 1 : sk;
}

int fn2_conv2d_kernel_name(const fn2_conv_desc* d, char* name, int cap) {
  FN2_REQUIRE(name && cap > 0, "conv2d_kernel_name: no buffer");
  name[0] = 0;
  if (!d || d->wgt_layout < 1 || is_flow_head(d)) return FN2_OK;  // only the LDS-DMA launchers report their choice
  conv_name_sink() = ConvNameSink{name, cap};
  const int rc = fn2_conv2d(d, nullptr);
  conv_name_sink() = ConvNameSink{nullptr, 0};
  return rc;
}

int fn2_conv2d(const fn2_conv_desc* d, void* stream) {
  ConvArgs a;
  int tile, phases;
  int rc = build_args(d, &a, &tile, &phases);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (is_flow_head(d)) {
    int blocks = grid_for((long)a.M * 64, 256);
    if (d->in.dtype == FN2_F16X2 && a.M <= 4096 && !(a.dbg & 32768)) {  // block per pixel on the small maps
      a.dbg |= 16384;
      blocks = a.M;
    } else {
      a.dbg &= ~16384;
    }
    if (d->in.dtype == FN2_F32)
      hipLaunchKernelGGL(flow_head_kernel<float>, dim3(blocks), dim3(256), 0, s, a);
    else if (d->in.dtype == FN2_BF16)
      hipLaunchKernelGGL(flow_head_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, a);
    else if (d->in.dtype == FN2_F16)
      hipLaunchKernelGGL(flow_head_kernel<f16_t>, dim3(blocks), dim3(256), 0, s, a);
    else
      hipLaunchKernelGGL(flow_head_kernel<x2_t>, dim3(blocks), dim3(256), 0, s, a);
    FN2_CHECK_LAUNCH("flow_head");
    return FN2_OK;
  }
  int sk = preferred_split(a, tile, phases);
  while (sk > 1 && (d->workspace == nullptr || split_bytes(a, sk) > d->workspace_bytes)) --sk;
  if ((long)a.N * a.out_H * a.out_W * a.ws_cs >= (1L << 31)) sk = 1;  // slab offsets are 31-bit (conv2.hip, partial stores)
  if (sk > 1) {
    a.kper = cdiv(a.ksteps, sk);
    sk = cdiv(a.ksteps, a.kper);
  }
  if (sk > 1) {
    a.splitk = sk;
    a.ws = reinterpret_cast<float*>(d->workspace);
  }
  if (d->up_src != nullptr && d->kind == 5 && 2 * d->out.c == 32) {   // one lane per output pixel holds all its channels
    a.up_src = d->up_src; a.up_w = d->up_w; a.up_bias = d->up_bias; a.up_c0 = d->up_c0;
  }
  if (d->head != nullptr && !conv_name_sink().buf) {
    // fn2_conv_desc.head: ride on this launch when it is the split-K split-fp16 launch on fragment-order weights (conv2.hip:
    // the head pixels are extra z slices of its grid), else run in front of it
    const fn2_conv_desc* h = reinterpret_cast<const fn2_conv_desc*>(d->head);
    FN2_REQUIRE(d->kind == 1 && is_flow_head(h) && h->head == nullptr && h->up_src == nullptr,
                "conv2d: head rides on a kind-1 transposed conv and is a 3x3 two-output fp32 flow head");
    FN2_REQUIRE(h->in.data == d->in.data && h->in.n == d->in.n && h->in.h == d->in.h && h->in.w == d->in.w &&
                    h->in.cs == d->in.cs && h->in.c0 == d->in.c0 && h->in.dtype == d->in.dtype,
                "conv2d: head and transposed conv must read the same view");
    ConvArgs ha;
    int htile, hphases;
    rc = build_args(h, &ha, &htile, &hphases);
    if (rc) return rc;
    const char* e = getenv("FN2_HEAD_RIDE");
    const bool ride = d->in.dtype == FN2_F16X2 && d->wgt_layout == 2 && tile == 128 && sk > 1 && !(e && atoi(e) == 0);
    if (ride) {
      a.fh_M = ha.M;
      a.fh_w = reinterpret_cast<const float*>(ha.wgt);
      a.fh_w1 = ha.ksteps * 4 * 4;          // floats per output plane (flow_head_kernel: ksteps * 4 chunks of 4 floats)
      a.fh_groups = ha.cin_chunks >> 1;
      a.fh_bias = ha.bias;
      a.fh_out = reinterpret_cast<float*>(ha.out);
      a.fh_out_cs = ha.out_cs; a.fh_out_c0 = ha.out_c0;
      a.fh_scale = ha.out_scale;
    } else {
      rc = fn2_conv2d(h, stream);
      if (rc) return rc;
    }
  }
  if (d->wgt_layout >= 1) {
    // LDS-DMA kernel: a stage is two generic k-steps (the tap never changes inside a stage)
    a.ksteps = a.ksteps / 2;
    a.kper = cdiv(a.kper, 2);
    a.splitk = cdiv(a.ksteps, a.kper);
    if (a.splitk == 1) a.ws = nullptr;
    rc = launch_conv_fast(a, d->in.dtype, d->out.dtype, tile, phases, s);
  } else if (d->in.dtype == FN2_F32 && d->out.dtype == FN2_F16X2) rc = launch_conv<float, x2_t>(a, tile, phases, s);
  else if (d->in.dtype == FN2_F32) rc = launch_conv<float, float>(a, tile, phases, s);
  else if (d->in.dtype == FN2_BF16 && d->out.dtype == FN2_BF16) rc = launch_conv<bf16_t, bf16_t>(a, tile, phases, s);
  else if (d->in.dtype == FN2_BF16) rc = launch_conv<bf16_t, float>(a, tile, phases, s);
  else if (d->out.dtype == FN2_F16) rc = launch_conv<f16_t, f16_t>(a, tile, phases, s);
  else rc = launch_conv<f16_t, float>(a, tile, phases, s);
  if (rc || conv_name_sink().buf) return rc;
  if (a.splitk == 1) {
    if (d->up_src == nullptr || a.up_src != nullptr) return rc;   // (a.up_src: the kind-5 epilogue did it)
    // no finalize pass to ride on: the stand-alone upsample_flow kernel behind the convolution
    fn2_tensor uv = d->out;
    uv.c = 2; uv.c0 = d->up_c0;
    return fn2_upsample_flow(d->up_src, d->up_w, d->up_bias, &uv, d->out.n, d->out.h / 2, d->out.w / 2, stream);
  }
  if ((a.dbg & 524288) || d->raw_partials) return rc;  // (524288: ablation, no finalize pass; raw_partials: the caller sums the slabs)
  const long npix = (long)a.N * a.out_H * a.out_W;
  const int fgrid = grid_for(npix * (a.ws_cs / 4 + (d->up_src != nullptr ? 1 : 0)), 256);
  if (d->out.dtype == FN2_F32)
    hipLaunchKernelGGL(splitk_finalize_kernel<float>, dim3(fgrid), dim3(256), 0, s, a.ws, a.bias, (float*)a.out, npix,
                       a.ws_cs, a.splitk, a.Cout, a.out_cs, a.out_c0, a.act, a.vec_ok, a.out_scale, a.accum, (const float*)a.mask_y, a.mask_c0, a.mask_c1, d->up_src, d->up_w, d->up_bias, d->up_c0, a.out_H, a.out_W);
  else if (d->out.dtype == FN2_F16X2)
    hipLaunchKernelGGL(splitk_finalize_kernel<x2_t>, dim3(fgrid), dim3(256), 0, s, a.ws, a.bias, (x2_t*)a.out,
                       npix, a.ws_cs, a.splitk, a.Cout, a.out_cs, a.out_c0, a.act, a.vec_ok, a.out_scale, a.accum, (const x2_t*)a.mask_y, a.mask_c0, a.mask_c1, d->up_src, d->up_w, d->up_bias, d->up_c0, a.out_H, a.out_W);
  else if (d->out.dtype == FN2_BF16)
    hipLaunchKernelGGL(splitk_finalize_kernel<bf16_t>, dim3(fgrid), dim3(256), 0, s, a.ws, a.bias, (bf16_t*)a.out,
                       npix, a.ws_cs, a.splitk, a.Cout, a.out_cs, a.out_c0, a.act, a.vec_ok, a.out_scale, a.accum, (const bf16_t*)a.mask_y, a.mask_c0, a.mask_c1, d->up_src, d->up_w, d->up_bias, d->up_c0, a.out_H, a.out_W);
  else
    hipLaunchKernelGGL(splitk_finalize_kernel<f16_t>, dim3(fgrid), dim3(256), 0, s, a.ws, a.bias, (f16_t*)a.out,
                       npix, a.ws_cs, a.splitk, a.Cout, a.out_cs, a.out_c0, a.act, a.vec_ok, a.out_scale, a.accum, (const f16_t*)a.mask_y, a.mask_c0, a.mask_c1, d->up_src, d->up_w, d->up_bias, d->up_c0, a.out_H, a.out_W);
  FN2_CHECK_LAUNCH("splitk_finalize");
  return FN2_OK;
}

int fn2_flow_head_gather(const float* t, int t_cs, const float* bias, float* out, int n, int h, int w, void* stream) {
  FN2_REQUIRE(t && out, "flow_head_gather: null pointer");
  FN2_REQUIRE(t_cs >= 18 && t_cs % 2 == 0 && n >= 1 && h >= 1 && w >= 1, "flow_head_gather: t must hold 18 partials per pixel");
  const long tiles = (long)n * ((h + GT_Y - 1) / GT_Y) * ((w + GT_X - 1) / GT_X);
  if (tiles >= 1024)  // >= 4 tiles per CU: the LDS-tiled form pays (full-resolution heads)
    hipLaunchKernelGGL(flow_head_gather_kernel, dim3((unsigned)std::min<long>(tiles, 1 << 16)), dim3(256), 0,
                       (hipStream_t)stream, t, t_cs, bias, out, n, h, w);
  else
    hipLaunchKernelGGL(flow_head_gather_small_kernel, dim3(grid_for((long)n * h * w, 256)), dim3(256), 0,
                       (hipStream_t)stream, t, t_cs, bias, out, n, h, w);
  FN2_CHECK_LAUNCH("flow_head_gather");
  return FN2_OK;
}

int fn2_flow_head_tail(const float* t, int t_cs, int taps, const float* bias, float* pf, int n, int h, int w, int ring,
                       const float* up_w, const float* up_bias, const fn2_tensor* up_out, void* stream) {
  return fn2_flow_head_tail_slabs(t, t_cs, 1, 0, 1.f, taps, bias, pf, n, h, w, ring, up_w, up_bias, up_out, stream);
}

int fn2_flow_head_tail_slabs(const float* t, int t_cs, int nslab, int64_t slab_stride, float scale, int taps, const float* bias,
                             float* pf, int n, int h, int w, int ring, const float* up_w, const float* up_bias,
                             const fn2_tensor* up_out, void* stream) {
  FN2_REQUIRE(t && pf, "flow_head_tail: null pointer");
  FN2_REQUIRE(nslab >= 1 && (nslab == 1 || slab_stride >= (int64_t)n * h * w * t_cs), "flow_head_tail: bad slab geometry");
  const TailSrc ts{t, t_cs, nslab, (long)slab_stride, scale};
  FN2_REQUIRE(taps == 3 || taps == 5, "flow_head_tail: taps must be 3 or 5");
  FN2_REQUIRE(t_cs >= 2 * taps * taps && t_cs % 2 == 0 && n >= 1 && h >= 1 && w >= 1,
              "flow_head_tail: t must hold 2 * taps^2 partials per pixel");
  FN2_REQUIRE(!ring || (h >= 3 && w >= 3), "flow_head_tail: a border ring needs h, w >= 3");
  const bool up = up_w != nullptr;
  hipStream_t s = (hipStream_t)stream;
  if (!up) {
    launch_head_tail_t<float>(taps, false, ts, bias, pf, n, h, w, ring, nullptr, nullptr, nullptr, 0, 0, s);
  } else {
    int rc = check_view(up_out, "flow_head_tail upsample output");
    if (rc) return rc;
    FN2_REQUIRE(up_out->c == 2 && up_out->n == n && up_out->h == 2 * h && up_out->w == 2 * w,
                "flow_head_tail: the upsample view must be [n, 2h, 2w, 2]");
    if (up_out->dtype == FN2_F16X2) FN2_REQUIRE(up_out->c0 % 2 == 0, "flow_head_tail: split-fp16 slice must start at an even channel");
    void* d = up_out->data;
    if (up_out->dtype == FN2_F32) launch_head_tail_t<float>(taps, true, ts, bias, pf, n, h, w, ring, up_w, up_bias, d, up_out->cs, up_out->c0, s);
    else if (up_out->dtype == FN2_F16X2) launch_head_tail_t<x2_t>(taps, true, ts, bias, pf, n, h, w, ring, up_w, up_bias, d, up_out->cs, up_out->c0, s);
    else if (up_out->dtype == FN2_BF16) launch_head_tail_t<bf16_t>(taps, true, ts, bias, pf, n, h, w, ring, up_w, up_bias, d, up_out->cs, up_out->c0, s);
    else launch_head_tail_t<f16_t>(taps, true, ts, bias, pf, n, h, w, ring, up_w, up_bias, d, up_out->cs, up_out->c0, s);
  }
  FN2_CHECK_LAUNCH("flow_head_tail");
  return FN2_OK;
}

int fn2_flow_head5(const fn2_tensor* x, const void* wgt, int cin_pad, int kpad, float out_scale, const float* bias, float* pf,
                   int ring, const float* ring_w, const float* ring_b, void* stream) {
  int rc = check_view(x, "flow_head5 input");
  if (rc) return rc;
  FN2_REQUIRE(wgt && pf, "flow_head5: null pointer");
  FN2_REQUIRE(x->dtype == FN2_F16X2 && x->cs % 8 == 0 && x->c0 % 8 == 0, "flow_head5: split-fp16 input, 8-aligned view");
  FN2_REQUIRE(cin_pad % 32 == 0 && cin_pad >= x->c && x->c0 + cin_pad <= x->cs && kpad == cin_pad,
              "flow_head5: the channel run must be whole 128-byte lines inside the buffer (kpad = cin_pad)");
  FN2_REQUIRE(!ring || (x->h >= 3 && x->w >= 3), "flow_head5: a border ring needs h, w >= 3");
  const long in_bytes = (long)x->n * x->h * x->w * x->cs * 4;
  FN2_REQUIRE(in_bytes < (1L << 31), "flow_head5: input buffer >= 2 GiB is not addressable by the LDS-DMA kernel");
  ConvArgs a = {};
  a.in = x->data; a.wgt = wgt; a.bias = bias; a.out = pf;
  a.N = x->n; a.H = x->h; a.W = x->w; a.in_cs = x->cs; a.in_c0 = x->c0;
  a.cin_chunks = cin_pad / 4;            // 16-byte chunks per tap (split fp16: 4 channels each)
  a.KH = a.KW = 1; a.stride = 1; a.pad = 0;
  a.OH = x->h; a.OW = x->w;
  a.h5_tx = cdiv(x->w, 28); a.h5_ty = cdiv(x->h, 4); a.h5_ring = ring ? 1 : 0;
  const long blocks = (long)x->n * a.h5_ty * a.h5_tx;
  FN2_REQUIRE(blocks * 256 < (1L << 31), "flow_head5: too many tiles");
  a.M = (int)(blocks * 256);
  a.out_H = x->h; a.out_W = x->w; a.out_cs = 2; a.out_c0 = 0; a.Cout = 2;
  a.ksteps = cin_pad / 32;               // 128-byte stages per packed row
  a.cout_pad = 64;
  a.act = FN2_ACT_NONE;
  a.kh_ph[0] = a.kh_ph[1] = a.kw_ph[0] = a.kw_ph[1] = 1;
  a.kg = 1; a.splitk = 1; a.kper = a.ksteps; a.ws = nullptr; a.ws_cs = 4;
  a.out_scale = out_scale == 0.f ? 1.f : out_scale;
  a.in_bytes = (int)in_bytes;
  { const char* e = getenv("FN2_H5_DBG"); a.dbg = e ? atoi(e) : 0; }   // 0: XCD-aware tile order; ablation bits 2 / 8 / 4 (timing only)
  a.h5_tiles = (int)blocks;
  long total = blocks;
  if (ring && ring_w != nullptr) {       // the ring as extra blocks of the same launch, 16 ring pixels each
    FN2_REQUIRE(ring_b != nullptr, "flow_head5: ring_w without ring_b");
    a.h5_wc = ring_w; a.h5_bc = ring_b; a.h5_groups = (x->c + 7) / 8;
    FN2_REQUIRE(x->c0 + a.h5_groups * 8 <= x->cs, "flow_head5: the last 8-channel group reaches past the buffer");
    total += cdiv((long)x->n * (2 * x->w + 2 * (x->h - 2)), 16);
  }
  // Strip form (conv2.hip: head5_strip_kernel) for 96- / 192-channel runs: a block walks down a strip of 60 output
  // columns; the row segments are cut so that the grid is one round of the chip (<= 256 blocks, 1 per CU: 112 / 136 KB
  // of LDS).  FN2_H5_STRIP=0: the tile form (A/B).
  {
    const char* e = getenv("FN2_H5_STRIP");
    if ((a.ksteps == 3 || a.ksteps == 6) && !(e && atoi(e) == 0) && !(a.dbg & 10)) {
      const int nstrip = cdiv(x->w, 60);
      const char* eb = getenv("FN2_H5_BLOCKS");
      const int target = eb ? atoi(eb) : 256;
      const int segs = std::max(1, std::min(x->h, target / std::max(1, x->n * nstrip)));
      const int rows = cdiv(x->h, segs);
      a.h5_tx = nstrip; a.h5_ty = cdiv(x->h, rows); a.h5_rows = rows;
      return launch_head5_strip(a, x->n * a.h5_ty * a.h5_tx, (hipStream_t)stream);
    }
  }
  return launch_head5(a, (int)total, (hipStream_t)stream);
}

int fn2_flow_head_ring(const fn2_tensor* x, const float* wc, const float* bc, float* pf, void* stream) {
  int rc = check_view(x, "flow_head_ring input");
  if (rc) return rc;
  FN2_REQUIRE(wc && bc && pf, "flow_head_ring: null pointer");
  FN2_REQUIRE(x->dtype == FN2_F16X2 && x->cs % 8 == 0 && x->c0 % 8 == 0, "flow_head_ring: split-fp16 input, 8-aligned view");
  FN2_REQUIRE(x->h >= 3 && x->w >= 3, "flow_head_ring: h, w >= 3");
  const int groups = (x->c + 7) / 8;
  FN2_REQUIRE(x->c0 + groups * 8 <= x->cs, "flow_head_ring: the last 8-channel group reaches past the buffer");
  const long ring = (long)x->n * (2 * x->w + 2 * (x->h - 2));
  hipLaunchKernelGGL(head_ring_kernel, dim3(grid_for(ring * 64, 256)), dim3(256), 0, (hipStream_t)stream, (const x2_t*)x->data,
                     x->cs, x->c0, groups, wc, bc, pf, x->n, x->h, x->w);
  FN2_CHECK_LAUNCH("flow_head_ring");
  return FN2_OK;
}

int fn2_upsample_flow(const float* in, const float* w, const float* bias, const fn2_tensor* out, int n, int h, int wd,
                      void* stream) {
  FN2_REQUIRE(in && w, "upsample_flow: null pointer");
  int rc = check_view(out, "upsample_flow output");
  if (rc) return rc;
  FN2_REQUIRE(out->c == 2 && out->n == n && out->h == 2 * h && out->w == 2 * wd,
              "upsample_flow: output view must be [n, 2h, 2w, 2]");
  const long total = (long)n * 4 * h * wd;
  if (out->dtype == FN2_F32)
    hipLaunchKernelGGL(upsample_flow_kernel<float>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       in, w, bias, (float*)out->data, n, h, wd, out->cs, out->c0);
  else if (out->dtype == FN2_F16X2)
    hipLaunchKernelGGL(upsample_flow_kernel<x2_t>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       in, w, bias, (x2_t*)out->data, n, h, wd, out->cs, out->c0);
  else if (out->dtype == FN2_BF16)
    hipLaunchKernelGGL(upsample_flow_kernel<bf16_t>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       in, w, bias, (bf16_t*)out->data, n, h, wd, out->cs, out->c0);
  else
    hipLaunchKernelGGL(upsample_flow_kernel<f16_t>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       in, w, bias, (f16_t*)out->data, n, h, wd, out->cs, out->c0);
  FN2_CHECK_LAUNCH("upsample_flow");
  return FN2_OK;
}

int fn2_pack_pair(const float* a, const float* b, const fn2_tensor* out, int pad, void* stream) {
  FN2_REQUIRE(a && b, "pack_pair: null pointer");
  int rc = check_view(out, "pack_pair output");
  if (rc) return rc;
  FN2_REQUIRE(out->c == 6, "pack_pair: output view must have 6 channels");
  return pack_imgs<2>(a, b, out, out->n, 0, pad, stream);
}

int fn2_pack_image_s2d(const float* img, int n_img, int h, int w, const fn2_tensor* out, int n0, int pad, void* stream) {
  FN2_REQUIRE(img, "pack_image_s2d: null pointer");
  int rc = check_view(out, "pack_image_s2d output");
  if (rc) return rc;
  FN2_REQUIRE(pad >= 0 && (h + 2 * pad) % 2 == 0 && (w + 2 * pad) % 2 == 0, "pack_image_s2d: padded size must be even");
  FN2_REQUIRE(out->h == (h + 2 * pad) / 2 && out->w == (w + 2 * pad) / 2 && out->c == 16 && out->c0 == 0 && out->cs == 16,
              "pack_image_s2d: output must be the whole [n, (h+2pad)/2, (w+2pad)/2, 16] buffer");
  FN2_REQUIRE(n_img >= 1 && n0 >= 0 && n0 + n_img <= out->n, "pack_image_s2d: rows [n0, n0+n_img) outside the buffer");
  const long total = (long)n_img * out->h * out->w;
  const dim3 g(grid_for(total, 256)), b(256);
  hipStream_t s = (hipStream_t)stream;
  if (out->dtype == FN2_F32)
    hipLaunchKernelGGL(pack_image_s2d_kernel<float>, g, b, 0, s, img, (float*)out->data, n_img, h, w, n0, pad, out->cs);
  else if (out->dtype == FN2_F16X2)
    hipLaunchKernelGGL(pack_image_s2d_kernel<x2_t>, g, b, 0, s, img, (x2_t*)out->data, n_img, h, w, n0, pad, out->cs);
  else if (out->dtype == FN2_BF16)
    hipLaunchKernelGGL(pack_image_s2d_kernel<bf16_t>, g, b, 0, s, img, (bf16_t*)out->data, n_img, h, w, n0, pad, out->cs);
  else
    hipLaunchKernelGGL(pack_image_s2d_kernel<f16_t>, g, b, 0, s, img, (f16_t*)out->data, n_img, h, w, n0, pad, out->cs);
  FN2_CHECK_LAUNCH("pack_image_s2d");
  return FN2_OK;
}

int fn2_pack_image(const float* img, int n_img, const fn2_tensor* out, int n0, int pad, void* stream) {
  FN2_REQUIRE(img, "pack_image: null pointer");
  int rc = check_view(out, "pack_image output");
  if (rc) return rc;
  FN2_REQUIRE(out->c == 3, "pack_image: output view must have 3 channels");
  FN2_REQUIRE(n_img >= 1 && n0 >= 0 && n0 + n_img <= out->n, "pack_image: rows [n0, n0+n_img) outside the buffer");
  return pack_imgs<1>(img, nullptr, out, n_img, n0, pad, stream);
}

}  // extern "C"
