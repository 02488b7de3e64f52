// Shared between conv.hip (generic implicit-GEMM kernel) and conv2.hip (LDS-DMA fast path).
#pragma once
#include "fn2_common.h"

namespace fn2 {

struct ConvArgs {
  const void* in;
  const void* wgt;
  const float* bias;
  void* out;
  int N, H, W, in_cs, in_c0;
  int cin_chunks;  // chunks per tap
  int KH, KW, stride, pad;
  int OH, OW;  // pixel grid of the GEMM (per phase for deconv)
  int M;       // N*OH*OW
  int out_H, out_W, out_cs, out_c0, Cout;
  int ksteps;  // packed row length / 4 chunks
  int cout_pad;
  int act;
  int deconv;   // 1: four output phases in blockIdx.z (transposed convolutions), output pixel (2m+a, 2m'+b)
  int ph_pad0, ph_pad1;  // tap origin of phase a = 0 / 1: input row = m - ph_pad + t
  int kh_ph[2], kw_ph[2];  // taps a phase really has (rows: phase >> 1, columns: phase & 1); the packed weight keeps KH x KW slots per phase.
                           // kind 3 (transpose of a k x k stride-2 conv): 9 of 16 slots (k = 3), 25 of 36 (k = 5) are non-zero
  int merged;   // kind 5 (halo kernel): blockIdx.z = output row phase a, packed row b Cout + co = column phase b, channel co
  int wmajor;   // LDS-DMA kernel: XCD bands run over the pixel tiles of ONE cout tile (weight-streaming layers), see conv.hip
  int bp64;     // LDS-DMA kernel: half-width pixel tiles (128x64 / 64x128 / 32x128), see wants_bp64 in conv.hip
  int KH_KW_hint;  // kind 2 (row-run stems): the real kernel width KW (KW itself is 1 there: a kernel row is one "tap"); else 0
  int kg;       // LDS-DMA kernel, 128 x 64 tiles: K groups per block (1, 2 or 3; conv2.hip), see build_args in conv.hip
  int h5_tx, h5_ty, h5_ring;  // composed 5x5 flow head (conv2.hip, HEAD5): tiles per image row / column, ring pixels handled apart
  int h5_tiles, h5_groups;    // ... blocks [h5_tiles, gridDim.x) compute the border ring from these (nullptr: no ring blocks):
  const float* h5_wc;         //     fp32 [9 cases][25 taps][8 h5_groups][2]
  const float* h5_bc;         //     fp32 [9][2]
  int h5_rows;                // strip form (head5_strip_kernel): h5_tx = strips of 60 output columns, h5_ty = row segments, h5_rows = output rows per segment
  // predict_flow(N+1) riding on the transposed conv of level N (fn2_conv_desc.head; conv2.hip, the WREG SLAB launch): both read
  // the same concat buffer.  Blocks with blockIdx.z >= fh_z0 compute head pixels (block per pixel, fh_pixel below).
  int fh_M, fh_z0;            // head pixels N * H * W (0: nothing rides); first z slice of the head part of the grid
  const float* fh_w;          // the head's fp32 weight: [o][tap][8 * fh_groups], plane o = 1 at + fh_w1 floats
  int fh_w1, fh_groups;
  const float* fh_bias;       // [2] or nullptr
  float* fh_out;              // fp32 view [n, h, w, fh_out_cs] + fh_out_c0
  int fh_out_cs, fh_out_c0;
  float fh_scale;
  // fn2_conv_desc.up_src on a kind-5 launch (conv_halo_kernel, merged column phases): the lane that holds an output pixel's
  // Cout channels also writes upsample_flow(N+1)toN of that pixel to channels [up_c0, up_c0 + 2) of the out buffer
  const float* up_src; const float* up_w; const float* up_bias; int up_c0;
  int wfrag;    // 1: the weight is stored in MFMA-fragment order (wgt_layout 2) and loaded straight into registers (conv2.hip, WREG)
  int accum;    // 1: out += result (fp32 outputs; gradient accumulation into shared buffers)
  int vec_ok;  // out_cs % 4 == 0 && out_c0 % 4 == 0
  int splitk;  // K splits (blockIdx.z = phase*splitk + split); > 1 -> raw fp32 partials go to `ws`
  int kper;    // k-steps per split
  float* ws;   // [splitk][N*out_H*out_W][ws_cs] fp32 partial sums
  int ws_cs;   // Cout rounded up to 4
  float out_scale;  // accumulator scale before bias (power of two chosen by the weight packer)
  int in_bytes;  // byte size of the input buffer (LDS-DMA kernel: buffer descriptor num_records)
  const void* mask_y;  // fused LeakyReLU backward (fn2_conv_desc.act_grad_y): forward activations laid out like `out`, or nullptr
  int mask_c0, mask_c1;  // ... applied to channels [mask_c0, mask_c1) of the output view
  int dbg;     // FN2_CONV_DBG ablation bits (timing experiments only; results are wrong when set)
};

// One pixel of a 3x3 two-output flow head on a split-fp16 input, by one block of 256 threads: the four waves split the
// 9 x C/8 (tap, 8-channel group) items, partial sums meet in `part` (8 floats of LDS).  flow_head_kernel's block-per-pixel
// form (conv.hip) and the head blocks of a transposed conv's launch (conv2.hip) are this function.
__device__ __forceinline__ void fh_pixel(const x2_t* __restrict__ in, int H, int W, int in_cs, int in_c0, int groups,
                                         const float* __restrict__ wf0, const float* __restrict__ wf1,
                                         const float* __restrict__ bias, float out_scale, float* __restrict__ po, long m,
                                         float* part) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int nitems = 9 * groups;
  const int x = (int)(m % W), y = (int)((m / W) % H);
  const size_t nb = (size_t)(m / W / H) * H * W;
  float a0 = 0.f, a1 = 0.f;
  for (int q = threadIdx.x; q < nitems; q += 256) {
    const int tap = q / groups, gi = q - tap * groups;
    const int ky = tap / 3, kx = tap - ky * 3;
    const int iy = y + ky - 1, ix = x + kx - 1;
    if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
    const uint4* src = reinterpret_cast<const uint4*>(in + (nb + (size_t)iy * W + ix) * in_cs + in_c0 + gi * 8);
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
  if (lane == 0) { part[wv * 2] = a0; part[wv * 2 + 1] = a1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    po[0] = (part[0] + part[2] + part[4] + part[6]) * out_scale + (bias ? bias[0] : 0.f);
    po[1] = (part[1] + part[3] + part[5] + part[7]) * out_scale + (bias ? bias[1] : 0.f);
  }
  __syncthreads();
}

// Backward of LeakyReLU fused into the epilogue of the layer that completes a gradient slice: x is the finished
// gradient wrt the activation OUTPUT at `elem`; the factor d/dx (0.55 x + 0.45 |x|) is read off the forward output y at
// the same position of the mirrored buffer (1 for y > 0, 0.1 for y < 0, 0.55 at 0: train.hip, act_bias_bwd_kernel).
template <typename OutT>
__device__ __forceinline__ float act_grad(const ConvArgs& p, const OutT* out_base, const OutT* elem, int co, float x) {
  if (p.mask_y != nullptr && co >= p.mask_c0 && co < p.mask_c1) {
    const float y = load_elem<OutT>(reinterpret_cast<const OutT*>(p.mask_y) + (elem - out_base));
    x *= y > 0.f ? 1.f : (y < 0.f ? 0.1f : 0.55f);
  }
  return x;
}

// The 16 consecutive couts [cout_base, cout_base + 16) of one pixel in an LDS-DMA kernel's epilogue: v (+)= what the
// output holds (accumulate), then the fused LeakyReLU backward; `full`: all 16 inside the view and 32-byte aligned.
template <typename OutT>
__device__ __forceinline__ void accum_act_grad16(const ConvArgs& p, const OutT* out_base, const OutT* po, int cout_base,
                                                 bool full, float (&v)[16]) {
  if constexpr (sizeof(OutT) == 4) {
    const bool mask = p.mask_y != nullptr && cout_base < p.mask_c1 && cout_base + 16 > p.mask_c0;
    if (!p.accum && !mask) return;
    if (full) {
      if (p.accum) {
        float o[16];
        load16<OutT>(po, o);
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] += o[q];
      }
      if (mask) {
        float y[16];
        load16<OutT>(reinterpret_cast<const OutT*>(p.mask_y) + (po - out_base), y);
#pragma unroll
        for (int q = 0; q < 16; ++q)
          if (cout_base + q >= p.mask_c0 && cout_base + q < p.mask_c1) v[q] *= y[q] > 0.f ? 1.f : (y[q] < 0.f ? 0.1f : 0.55f);
      }
    } else {
#pragma unroll
      for (int q = 0; q < 16; ++q)
        if (cout_base + q < p.Cout) {
          if (p.accum) v[q] += load_elem<OutT>(po + q);
          v[q] = act_grad<OutT>(p, out_base, po + q, cout_base + q, v[q]);
        }
    }
  }
}

template <typename OutT>
__device__ __forceinline__ void store4(OutT* p, float a, float b, float c, float d) {
  const float v[4] = {a, b, c, d};
  store_vec<OutT, 4>(p, v);
}

// Name sink: while set (fn2_conv2d_kernel_name), the launchers write the instantiation they would launch -- as
// rocprofv3 prints it, without the "void fn2::" prefix and the argument list -- and launch nothing.
struct ConvNameSink { char* buf; int cap; };
ConvNameSink& conv_name_sink();

// conv2.hip: the fast path.  Returns FN2_ERR_UNSUPPORTED when (dtype, tile) has no instantiation.
int launch_conv_fast(const ConvArgs& a, int in_dtype, int out_dtype, int tile, int phases, hipStream_t s);
// conv2.hip: the composed 5x5 flow head (HEAD5 instantiation); `blocks` = n * tiles_y * tiles_x
int launch_head5(const ConvArgs& a, int blocks, hipStream_t s);
int launch_head5_strip(const ConvArgs& a, int blocks, hipStream_t s);
// true when the fast kernel covers this geometry (then the packed weight must use the permuted-64 row order)
bool conv_fast_ok(int in_dtype, int cin_pad, int cout);

}  // namespace fn2
