// FlowNetC cost volume, second-generation kernel: the same banded GEMM per (n, y, 64-pixel block) as
// corr.hip, with the operand movement of the fast convolution kernel:
//   * the displaced B-row window (112 pixels x 128-byte channel lines) goes L2 -> LDS by LDS-DMA through a
//     buffer descriptor; pixels left/right of the image (the reference's zero Pad, pad.cu.cc:46-74) are an
//     out-of-range offset = zeros, so a lane's offset is computed ONCE and the per-stage motion (displaced
//     row, channel line) is a scalar offset: no address VALU, no staging registers, no ds_write;
//   * 128-byte LDS rows, XOR swizzle on the source chunk and on the ds_read_b128 fragment reads;
//   * split-fp16 features (FN2_F16X2) are multiplied on the fp16 matrix cores (hi*hi + hi*lo + lo*hi)
//     instead of being converted to fp32 operands: the DMA source permutation puts the four hi chunks
//     of a line in LDS chunks 0..3 and the four lo chunks in 4..7.
// out[n,y,x,(p+gr)*gw+(o+gr)] = (1/C) sum_c A[n,y,x,c] * B0[n,y+s2p,x+s2o,c]   (correlation_kernel.cu.cc:45-110)
#include "fn2_common.h"

namespace fn2 {

typedef __attribute__((address_space(3))) void* lptr_t;

struct CorrArgs {
  const void* a;
  const void* b;
  void* out;
  int N, H, W;
  int a_cs, a_c0, b_cs, b_c0, out_cs, out_c0;
  int md, s2, gr, gw, lo;
  int act;
  float c_f;
};

constexpr int C2_NBT = 4;             // window tiles per A tile
constexpr int C2_WIN = 4 + C2_NBT - 1;  // 7 window tiles = 112 pixels
constexpr unsigned kOob = 0x80000000u;

// NL = 128-byte channel lines per pixel (C * sizeof(T) / 128)
template <typename T, typename OutT, int NL>
__global__ void __launch_bounds__(256) corr2_kernel(const CorrArgs p, const int b_bytes) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int ESZ = (int)sizeof(T);
  constexpr bool X2 = is_x2<T>::value;
  constexpr int SG = NL < 2 ? 1 : 2;  // lines per LDS stage
  constexpr int NSTAGE = NL / SG;
  constexpr int WPX = C2_WIN * 16;
  constexpr int NPIECE = WPX * SG / 8;  // DMA wave-instructions per stage (8 rows of 128 B each)
  constexpr int PPW = (NPIECE + 3) / 4;
  // two LDS objects (not lds[2][..]): lets the waitcnt pass prove that a stage's ds_reads do not touch the object the
  // next stage's LDS-DMA is filling (see conv2.hip); the buffer choice is static when NSTAGE is even
  __shared__ uint4 lds0[SG][WPX * 8];
  __shared__ uint4 lds1[SG][WPX * 8];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware order (see conv2.hip): workgroups go round-robin over the 8 XCDs by linear id, so with the plain
  // (x, y, n) mapping the 21 output rows that share a displaced B row run on 8 different L2s.  Each XCD gets a
  // contiguous band of (n, y, x-block) instead: for batch 8 exactly one image, whose B features (3 MB in split
  // fp16) stay in that XCD's 4 MB L2 across the 21 re-reads.
  const int gx = gridDim.x, gy = gridDim.y;
  const int NT = gx * gy * (int)gridDim.z, L = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
  const int xcd = L & 7, chunk = NT >> 3, rem = NT & 7;
  const int Lp = xcd * chunk + min(xcd, rem) + (L >> 3);
  const int xb = (Lp % gx) * 64, y = (Lp / gx) % gy, n = Lp / (gx * gy);
  OutT* out = reinterpret_cast<OutT*>(p.out);
  const int fi = lane & 15, fg = lane >> 4;

  // ---- A tile (16 pixels x all channels) -> registers as MFMA operands, two chunks per line
  uint4 fa[NL][2];
  {
    const int x = xb + wave * 16 + fi;
    const bool ok = x < p.W;
    const char* src = reinterpret_cast<const char*>(p.a) +
                      ((((size_t)n * p.H + y) * p.W + (ok ? x : 0)) * p.a_cs + p.a_c0) * ESZ;
#pragma unroll
    for (int l = 0; l < NL; ++l)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        // 16-bit: k-step q, chunk 4q+fg;  fp32: chunk 4q+fg (4 MFMAs each);  split fp16: q=0 hi, q=1 lo of group fg
        const int chunk = X2 ? (2 * fg + q) : (4 * q + fg);
        fa[l][q] = ok ? *reinterpret_cast<const uint4*>(src + l * 128 + chunk * 16) : make_uint4(0, 0, 0, 0);
      }
  }

  // valid displaced rows: yb = y + (pi - gr)*s2 in [0, H)
  int p_lo = 0, p_hi = p.gw - 1;
  while (p_lo <= p_hi && y + (p_lo - p.gr) * p.s2 < 0) ++p_lo;
  while (p_hi >= p_lo && y + (p_hi - p.gr) * p.s2 >= p.H) --p_hi;
  for (int pi = 0; pi < p.gw; ++pi) {  // rows displaced fully into the zero padding: exactly 0
    if (pi >= p_lo && pi <= p_hi) continue;
    for (int idx = lane; idx < 16 * p.gw; idx += 64) {
      const int px = idx / p.gw, o = idx - px * p.gw;
      const int x = xb + wave * 16 + px;
      if (x < p.W) store_elem<OutT>(out + (((size_t)n * p.H + y) * p.W + x) * p.out_cs + p.out_c0 + pi * p.gw + o, 0.f);
    }
  }
  const int np = p_hi - p_lo + 1;
  if (np <= 0) return;

  // ---- DMA: per-lane byte offset of (window pixel, line-in-stage, source chunk), computed once
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.b), 0, b_bytes, 0x00020000);
  unsigned voff[PPW];
#pragma unroll
  for (int k = 0; k < PPW; ++k) {
    const int q = wave + 4 * k;  // piece
    const int sl = q / (WPX / 8), rg = q - sl * (WPX / 8);
    const int px = rg * 8 + (lane >> 3);
    const int L = (lane & 7) ^ ((px >> 1) & 7);          // logical chunk held at this physical position
    const int G = X2 ? ((L & 3) * 2 + (L >> 2)) : L;      // split fp16: LDS [hi0..hi3 | lo0..lo3] <- global [hi0 lo0 hi1 lo1 ..]
    const int xw = xb - p.md + px;
    const bool ok = q < NPIECE && xw >= 0 && xw < p.W;
    voff[k] = ok ? (unsigned)((xw * p.b_cs + p.b_c0) * ESZ + sl * 128 + G * 16) : kOob;
  }
  const unsigned row_bytes = (unsigned)(p.W * p.b_cs * ESZ);
  auto issue_stage = [&](int it, uint4 (*lds)[WPX * 8]) {
    const int pi = p_lo + it / NSTAGE, st = it - (it / NSTAGE) * NSTAGE;
    const int yb = y + (pi - p.gr) * p.s2;
    const int soff = (n * p.H + yb) * (int)row_bytes + st * SG * 128;
#pragma unroll
    for (int k = 0; k < PPW; ++k) {
      const int q = wave + 4 * k;
      if (q < NPIECE) {
        const int sl = q / (WPX / 8), rg = q - sl * (WPX / 8);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)&lds[sl][rg * 64], 16, voff[k], soff, 0, 0);
      }
    }
  };

  const int tb0 = wave + p.lo / 16;
  const int fsw = (fi >> 1) & 7;
  f32x4 acc[C2_NBT];
  const int niter = np * NSTAGE;
  issue_stage(0, lds0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int pidx = 0; pidx < np; ++pidx) {
#pragma unroll
    for (int tb = 0; tb < C2_NBT; ++tb) acc[tb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int st = 0; st < NSTAGE; ++st) {
      const int it = pidx * NSTAGE + st;
      const int buf = (NSTAGE % 2 == 0) ? (st & 1) : (it & 1);
      uint4 (*lds)[WPX * 8] = buf ? lds1 : lds0;
      if (it + 1 < niter) issue_stage(it + 1, buf ? lds0 : lds1);
#pragma unroll
      for (int sl = 0; sl < SG; ++sl) {
        const int l = st * SG + sl;
#pragma unroll
        for (int tb = 0; tb < C2_NBT; ++tb) {
          const uint4* row = &lds[sl][((tb0 + tb) * 16 + fi) * 8];
          const uint4 b0 = row[fg ^ fsw], b1 = row[(4 + fg) ^ fsw];
          if constexpr (X2) {  // b0 = hi, b1 = lo of group fg; fa[l][0] = hi, fa[l][1] = lo
            acc[tb] = mfma_16x16x32<f16_t>(fa[l][1], b0, acc[tb]);
            acc[tb] = mfma_16x16x32<f16_t>(fa[l][0], b1, acc[tb]);
            acc[tb] = mfma_16x16x32<f16_t>(fa[l][0], b0, acc[tb]);
          } else if constexpr (sizeof(T) == 2) {
            acc[tb] = mfma_16x16x32<T>(fa[l][0], b0, acc[tb]);
            acc[tb] = mfma_16x16x32<T>(fa[l][1], b1, acc[tb]);
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              acc[tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(f32x4, fa[l][0])[j],
                                                           __builtin_bit_cast(f32x4, b0)[j], acc[tb], 0, 0, 0);
              acc[tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(f32x4, fa[l][1])[j],
                                                           __builtin_bit_cast(f32x4, b1)[j], acc[tb], 0, 0, 0);
            }
          }
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    // D layout: row (A pixel) = fg*4 + r, col (window pixel) = fi
    const int pi = p_lo + pidx;
#pragma unroll
    for (int tb = 0; tb < C2_NBT; ++tb) {
      const int j = (tb0 + tb) * 16 + fi;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = fg * 4 + r;
        const int x = xb + wave * 16 + i;
        const int delta = j - (wave * 16 + i) - p.lo;  // = (o + gr) * s2
        if (x < p.W && delta >= 0 && delta <= 2 * p.gr * p.s2 && (delta % p.s2) == 0) {
          float v = acc[tb][r] / p.c_f;
          if (p.act == FN2_ACT_LEAKY) v = leaky(v);
          store_elem<OutT>(out + (((size_t)n * p.H + y) * p.W + x) * p.out_cs + p.out_c0 + pi * p.gw + delta / p.s2, v);
        }
      }
    }
  }
#endif  // __HIP_DEVICE_COMPILE__
}

template <typename T, typename OutT>
static int launch2(const CorrArgs& a, int C, int b_bytes, hipStream_t s) {
  const int nl = C * (int)sizeof(T) / 128;
  dim3 grid(cdiv(a.W, 64), a.H, a.N), block(256);
  switch (nl) {
#define FN2_C2_CASE(NL_)                                                               \
  case NL_:                                                                            \
    hipLaunchKernelGGL((corr2_kernel<T, OutT, NL_>), grid, block, 0, s, a, b_bytes);   \
    break;
    FN2_C2_CASE(1)
    FN2_C2_CASE(2)
    FN2_C2_CASE(4)
    FN2_C2_CASE(8)
#undef FN2_C2_CASE
    default:
      return fail(FN2_ERR_UNSUPPORTED, "correlation: unsupported channel count %d", C);
  }
  FN2_CHECK_LAUNCH("corr2");
  return FN2_OK;
}

// Attributes / sizes the DMA kernel covers (otherwise corr.hip's register-staged kernel or the generic one)
bool corr2_ok(int C, int dtype, int md, int s2, long b_bytes) {
  const int esz = dtype_size(dtype);
  if ((C * esz) % 128 != 0) return false;
  const int nl = C * esz / 128;
  if (!(nl == 1 || nl == 2 || nl == 4 || nl == 8)) return false;
  const int gr = md / s2, lo = md - gr * s2;
  const int nbt = (15 + lo + 2 * gr * s2) / 16 - lo / 16 + 1;
  return nbt <= C2_NBT && (64 + 2 * md) <= C2_WIN * 16 && lo / 16 == 0 && b_bytes < (1L << 31);
}

// a/b views given as (ptr, cs, c0); dtype codes as fn2_dtype
int launch_corr2(const void* a, int a_cs, int a_c0, const void* b, int b_cs, int b_c0, void* out, int out_cs,
                 int out_c0, int in_dtype, int out_dtype, int N, int H, int W, int C, int md, int s2, int gr, int gw,
                 int act, hipStream_t s) {
  CorrArgs g;
  g.a = a; g.b = b; g.out = out;
  g.N = N; g.H = H; g.W = W;
  g.a_cs = a_cs; g.a_c0 = a_c0; g.b_cs = b_cs; g.b_c0 = b_c0; g.out_cs = out_cs; g.out_c0 = out_c0;
  g.md = md; g.s2 = s2; g.gr = gr; g.gw = gw; g.lo = md - gr * s2;
  g.act = act;
  g.c_f = (float)C;
  const int bb = (int)((long)N * H * W * b_cs * dtype_size(in_dtype));
  if (in_dtype == FN2_F32) return launch2<float, float>(g, C, bb, s);
  if (in_dtype == FN2_BF16) return out_dtype == FN2_BF16 ? launch2<bf16_t, bf16_t>(g, C, bb, s) : launch2<bf16_t, float>(g, C, bb, s);
  if (in_dtype == FN2_F16) return out_dtype == FN2_F16 ? launch2<f16_t, f16_t>(g, C, bb, s) : launch2<f16_t, float>(g, C, bb, s);
  return out_dtype == FN2_F16X2 ? launch2<x2_t, x2_t>(g, C, bb, s) : launch2<x2_t, float>(g, C, bb, s);
}

}  // namespace fn2
