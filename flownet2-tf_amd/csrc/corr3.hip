// FlowNetC cost volume, third-generation kernel for the call-site attributes (flownet_c.py:40: kernel 1, max
// displacement 20, stride_2 = 2, pad 20 -> 21 x 21 displacements of even offsets):
//   out[n,y,x,(p+10)*21+(o+10)] = (1/C) sum_c A[n,y,x,c] * B0[n,y+2p,x+2o,c]       (correlation_kernel.cu.cc:45-110)
//
// corr2.hip is bound by three things, none of them arithmetic: (1) every displaced B row is fetched L2 -> LDS once
// per A row that uses it (21 times) -- 0.9 GB at batch 8; (2) a result run of 21 channels is written as 2-byte pieces
// of 128-byte lines that other displacement rows complete much later; (3) a block walks 21 displacement rows x 4
// stages one after the other with one stage in flight.  This kernel:
//   * same-parity row quads: a block owns A rows y, y + 2, y + 4, y + 6 of a 32-pixel column block (8 waves = 4 rows
//     x 2 column parities), which use the SAME displaced B rows but for three at each end -- 11 B rows serve 4 x 8
//     (row, displacement-row) pairs instead of 32;
//   * same-parity columns: a wave's 16 A pixels are x0 + 2i, the B window is de-interleaved by column parity at DMA
//     time (the source address is per lane, so any permutation is free), and a 16 x 48 band holds the 21 even
//     displacements of 16 pixels: 3 MFMA column tiles instead of 4 of 16 x 64 (21 of 48 computed columns are used);
//   * displacement rows in chunks of 8 (grid.z x 3): 21 * 8 = 168 channels is a whole number of 8-channel groups, so
//     a block accumulates its 128 pixels x 168 channels in LDS and stores them once, as whole 32-byte split-fp16
//     groups / whole lines;
//   * a three-slot ring of two-line stages with the LDS-DMA two stages ahead (counted vmcnt, bare s_barrier).
// 1/C, LeakyReLU and the write into the 473-channel concat are fused as before.  Rows displaced out of the image
// are never fetched: their outputs are the zeros the LDS tile starts with (the reference's Pad, pad.cu.cc:46-74).
#include "fn2_common.h"

namespace fn2 {

typedef __attribute__((address_space(3))) void* lptr3_t;

struct CorrArgs {  // same layout as corr2.hip / corr.hip
  const void* a;
  const void* b;
  void* out;
  int N, H, W;
  int a_cs, a_c0, b_cs, b_c0, out_cs, out_c0;
  int md, s2, gr, gw, lo;
  int act;
  float c_f;
};

#ifndef FN2_C3_XS
#define FN2_C3_XS 2
#endif
constexpr int C3_XS = FN2_C3_XS;  // 32-pixel column spans per block (waves per row and parity)
constexpr int C3_HALF = C3_XS == 2 ? 64 : 48;  // LDS rows per column parity: window half-indices (16 (XS-1) + 36 needed, 3 tiles of 16 read per span)
constexpr int C3_ROWS = 2 * C3_HALF;
constexpr int C3_OTS = 168;    // channels of a chunk of 8 displacement rows
#ifndef FN2_C3_SLOTS
#define FN2_C3_SLOTS 3
#endif
constexpr int C3_SLOTS = FN2_C3_SLOTS;   // ring slots; C3_SLOTS - 1 stages of DMA in flight
constexpr int C3_SG = 1;       // 128-byte channel lines per stage
constexpr int C3_TY = 2;       // same-parity A rows per block
constexpr int C3_NW = 2 * C3_TY * C3_XS;  // waves per block: rows x column parities x spans
constexpr int C3_PX = 32 * C3_TY * C3_XS; // output pixels per block
constexpr unsigned kOob3 = 0x80000000u;

typedef int v4i3_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16_3(const v4i3_t rsrc, const void* lds, unsigned voff, int soff) {
  const unsigned la = (unsigned)(unsigned long long)(lptr3_t)lds;
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :: "s"(la), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

// NL = 128-byte channel lines per pixel (C * sizeof(T) / 128), a multiple of C3_SG
template <typename T, typename OutT, int NL>
__global__ void __launch_bounds__(64 * C3_NW) corr3_kernel(const CorrArgs p, const int b_bytes, const int dbg) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int ESZ = (int)sizeof(T);
  constexpr bool X2 = is_x2<T>::value;
  constexpr int SG = NL < C3_SG ? 1 : C3_SG;
  constexpr int NST = NL / SG;                      // stages per B row
  constexpr int NPIECE = C3_ROWS * SG / 8;          // DMA wave-instructions per stage (24)
  constexpr int PPW = (NPIECE + C3_NW - 1) / C3_NW;  // per wave (3)
  __shared__ uint4 ring[C3_SLOTS][C3_SG][C3_ROWS * 8];   // 3 x 12 KB
  __shared__ float ot[C3_PX * C3_OTS];                   // 42 KB: [2 rows x 32 pixels][168 channels] -> two blocks per CU

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware order (conv2.hip): each XCD walks a contiguous band of (n, chunk, row quad, x block), so the B rows of
  // one image stay in that XCD's L2 across the blocks that share them
  const int gx = gridDim.x, gy = gridDim.y;
  const int NT = gx * gy * (int)gridDim.z, L = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
  const int xcd = L & 7, chk = NT >> 3, rem = NT & 7;
  const int Lp = xcd * chk + min(xcd, rem) + (L >> 3);
  const int xb = (Lp % gx) * (32 * C3_XS);
  const int g = (Lp / gx) % gy;
  const int zc = Lp / (gx * gy);
  const int n = zc / 3, c = zc - n * 3;
  const int y0 = 2 * C3_TY * (g >> 1) + (g & 1);    // A rows y0, y0 + 2, ..
  const int pi0 = 8 * c, npi = min(8, p.gw - pi0);  // displacement rows [pi0, pi0 + npi)

  const int fi = lane & 15, fg = lane >> 4;
  const int hh = wave % C3_XS, par = (wave / C3_XS) & 1, r = wave / (2 * C3_XS);  // span, column parity, A row
  const int ya = y0 + 2 * r;
  const int xa = xb + 32 * hh + par + 2 * fi;       // this lane's A pixel as an MFMA row / column index fi

  // ---- zero the output tile
  for (int i = tid; i < C3_PX * C3_OTS / 4; i += 64 * C3_NW) reinterpret_cast<float4*>(ot)[i] = make_float4(0.f, 0.f, 0.f, 0.f);

  // ---- A fragments -> registers (16 pixels x all channels per wave)
  uint4 fa[NL][2];
  {
    const bool ok = xa < p.W && ya < p.H;
    const char* src = reinterpret_cast<const char*>(p.a) +
                      ((((size_t)n * p.H + (ok ? ya : 0)) * p.W + (ok ? xa : 0)) * p.a_cs + p.a_c0) * ESZ;
#pragma unroll
    for (int l = 0; l < NL; ++l)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int chunk = X2 ? (2 * fg + q) : (4 * q + fg);  // split fp16: q = 0 hi, 1 lo of group fg; 16-bit: k-step q
        fa[l][q] = ok ? *reinterpret_cast<const uint4*>(src + l * 128 + chunk * 16) : make_uint4(0, 0, 0, 0);
      }
  }

  // ---- B rows of the block: yb(k) = y0 + 2 (pi0 - 10) + 2 k, k in [0, npi + TY - 2]; row r uses k as pi = pi0 + k - r
  const int yb0 = y0 + 2 * (pi0 - p.gr);
  int k_lo = 0, k_hi = npi + C3_TY - 2;
  while (k_lo <= k_hi && yb0 + 2 * k_lo < 0) ++k_lo;
  while (k_hi >= k_lo && yb0 + 2 * k_hi >= p.H) --k_hi;
  const int nrow = k_hi - k_lo + 1;
  const int S = nrow > 0 ? nrow * NST : 0;  // stages

  // ---- DMA offsets: NPIECE pieces of 8 rows per stage, PPW per wave.  LDS row rho of a line <- window pixel 2 rho
  // (rho < 48) or 2 (rho - 48) + 1; pixels outside the image read zeros (range check)
  const v4i3_t rsrc = v4i3_t{(int)((unsigned long long)p.b & 0xffffffffu), (int)(((unsigned long long)p.b >> 32) & 0xffffu),
                             b_bytes, 0x00020000};
  unsigned voff[PPW];
#pragma unroll
  for (int k = 0; k < PPW; ++k) {
    const int q = wave * PPW + k;                        // piece
    const int sl = q / (C3_ROWS / 8), rg = q - sl * (C3_ROWS / 8);
    const int rho = rg * 8 + (lane >> 3);
    const int Lc = (lane & 7) ^ ((rho >> 1) & 7);        // logical chunk held at this physical position
    const int G = X2 ? ((Lc & 3) * 2 + (Lc >> 2)) : Lc;  // split fp16: LDS [hi0..hi3 | lo0..lo3] <- global [hi0 lo0 hi1 lo1 ..]
    const int px = rho < C3_HALF ? 2 * rho : 2 * (rho - C3_HALF) + 1;
    const int xw = xb - p.md + px;
    const bool ok = q < NPIECE && xw >= 0 && xw < p.W;
    voff[k] = ok ? (unsigned)((xw * p.b_cs + p.b_c0) * ESZ + sl * 128 + G * 16) : kOob3;
  }
  const int row_bytes = p.W * p.b_cs * ESZ;
  // issue state, advanced incrementally (a stage is ~150 cycles of matrix work: no divisions on this path)
  int is_s = 0, is_st = 0, is_slot = 0;
  int is_soff = (n * p.H + yb0 + 2 * k_lo) * row_bytes;
  auto issue_stage = [&]() {  // next stage = (B row, lines SG * st ..) into the next ring slot; past the end: zeros
    uint4 (*slot)[C3_ROWS * 8] = ring[is_slot];
    const bool live = is_s < S;
#pragma unroll
    for (int k = 0; k < PPW; ++k) {
      const int q = wave * PPW + k;
      const int sl = q / (C3_ROWS / 8), rg = q - sl * (C3_ROWS / 8);
      if (q < NPIECE && !(dbg & 1)) dma16_3(rsrc, &slot[sl][rg * 64], live ? voff[k] : kOob3, live ? is_soff : 0);
    }
    ++is_s;
    is_slot = is_slot == C3_SLOTS - 1 ? 0 : is_slot + 1;
    if (++is_st == NST) { is_st = 0; is_soff += 2 * row_bytes - (NST - 1) * SG * 128; }
    else is_soff += SG * 128;
  };

  const int fsw = (fi >> 1) & 7;                    // ((LDS row) >> 1) & 7 of the rows this lane reads (48 / 2 = 0 mod 8)
  const int rbase = par * C3_HALF + 16 * hh + fi;   // + 16 tb
  // 1 / C is exact when C is a power of two (the call site's 256): multiply; otherwise divide as the reference does
  const int ci = (int)p.c_f;
  const bool pow2 = (ci & (ci - 1)) == 0;
  const float rcp = 1.f / p.c_f;
  f32x4 acc[3];
#pragma unroll
  for (int tb = 0; tb < 3; ++tb) acc[tb] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (S > 0) {
#pragma unroll
    for (int d = 0; d < C3_SLOTS - 1; ++d) issue_stage();
    __syncthreads();  // the zeroed output tile is visible to every wave (DMA still in flight: waited below)
    int cslot = 0;
    for (int kk = 0; kk < nrow; ++kk) {
#pragma unroll
      for (int st = 0; st < NST; ++st) {  // static st: fa[] indices are compile-time
        // all but the newest stage (PPW pieces of this wave) have landed; every wave is done reading slot (s-1)%3
        // (all but the C3_SLOTS - 2 newest stages of this wave's pieces)
        static_assert(PPW * (C3_SLOTS - 2) <= 63, "vmcnt field");
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(PPW * (C3_SLOTS - 2)) : "memory");
        issue_stage();
        uint4 (*slot)[C3_ROWS * 8] = ring[cslot];
        cslot = cslot == C3_SLOTS - 1 ? 0 : cslot + 1;
        const int pil_s = k_lo + kk - r;  // this B row is displacement row pil_s of the wave's A row: used or not (wave-uniform)
        if (!(dbg & 2) && pil_s >= 0 && pil_s < npi)
#pragma unroll
        for (int sl = 0; sl < SG; ++sl) {
          const int l = st * SG + sl;
#pragma unroll
          for (int tb = 0; tb < 3; ++tb) {
            const uint4* row = &slot[sl][(rbase + 16 * tb) * 8];
            const uint4 b0 = row[fg ^ fsw], b1 = row[(4 + fg) ^ fsw];
            if constexpr (X2) {  // b0 = hi, b1 = lo of group fg; fa[l][0] = hi, fa[l][1] = lo
              acc[tb] = mfma_16x16x32<f16_t>(fa[l][1], b0, acc[tb]);
              acc[tb] = mfma_16x16x32<f16_t>(fa[l][0], b1, acc[tb]);
              acc[tb] = mfma_16x16x32<f16_t>(fa[l][0], b0, acc[tb]);
            } else {
              acc[tb] = mfma_16x16x32<T>(fa[l][0], b0, acc[tb]);
              acc[tb] = mfma_16x16x32<T>(fa[l][1], b1, acc[tb]);
            }
          }
        }
      }
      // B row done: D[i][j] (i = fg*4 + q: A pixel, j = fi: window column of tile tb) is displacement o' = 16 tb + j - i
      const int pil = k_lo + kk - r;  // displacement row of this wave's A row, relative to pi0
      const bool rowok = pil >= 0 && pil < npi;
#pragma unroll
      for (int tb = 0; tb < 3; ++tb) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int i = fg * 4 + q;
          const int o = 16 * tb + fi - i;
          if (rowok && o >= 0 && o <= 2 * p.gr && !(dbg & 4)) {
            float v = pow2 ? acc[tb][q] * rcp : acc[tb][q] / p.c_f;
            if (p.act == FN2_ACT_LEAKY) v = leaky(v);
            ot[(r * 32 * C3_XS + 32 * hh + par + 2 * i) * C3_OTS + pil * p.gw + o] = v;
          }
        }
        acc[tb] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the look-ahead (zero) stages before the block ends
  }
  __syncthreads();

  if (dbg & 8) return;
  // ---- store the tile: pixel (r, x) -> channels [pi0 * 21, pi0 * 21 + npi * 21) of its output row
  OutT* out = reinterpret_cast<OutT*>(p.out);
  const int nch = npi * p.gw;
  const int ch0 = p.out_c0 + pi0 * p.gw;
  if constexpr (is_x2<OutT>::value) {
    // whole 8-channel groups as 32 contiguous bytes (hi | lo); a last ragged group goes channel by channel
    const int ngrp = nch >> 3;
    for (int idx = tid; idx < C3_PX * ngrp; idx += 64 * C3_NW) {
      const int px = idx / ngrp, gi = idx - px * ngrp;
      const int rr = px / (32 * C3_XS), x = xb + px % (32 * C3_XS), y = y0 + 2 * rr;
      if (x >= p.W || y >= p.H) continue;
      const float* src = &ot[px * C3_OTS + gi * 8];
      float v[8];
      *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(src);
      *reinterpret_cast<float4*>(v + 4) = *reinterpret_cast<const float4*>(src + 4);
      uint4* q = reinterpret_cast<uint4*>(out + (((size_t)n * p.H + y) * p.W + x) * p.out_cs + ch0 + gi * 8);
      split8(v, q[0], q[1]);
    }
    const int tail = nch - (ngrp << 3);
    for (int idx = tid; idx < C3_PX * tail; idx += 64 * C3_NW) {
      const int px = idx / tail, j = (ngrp << 3) + (idx - px * tail);
      const int rr = px / (32 * C3_XS), x = xb + px % (32 * C3_XS), y = y0 + 2 * rr;
      if (x < p.W && y < p.H)
        store_elem<OutT>(out + (((size_t)n * p.H + y) * p.W + x) * p.out_cs + ch0 + j, ot[px * C3_OTS + j]);
    }
  } else {
    for (int idx = tid; idx < C3_PX * nch; idx += 64 * C3_NW) {
      const int px = idx / nch, j = idx - px * nch;
      const int rr = px / (32 * C3_XS), x = xb + px % (32 * C3_XS), y = y0 + 2 * rr;
      if (x < p.W && y < p.H)
        store_elem<OutT>(out + (((size_t)n * p.H + y) * p.W + x) * p.out_cs + ch0 + j, ot[px * C3_OTS + j]);
    }
  }
#endif  // __HIP_DEVICE_COMPILE__
}

template <typename T, typename OutT>
static int launch3(const CorrArgs& a, int C, int b_bytes, hipStream_t s) {
  const int nl = C * (int)sizeof(T) / 128;
  dim3 grid(cdiv(a.W, 32 * C3_XS), a.H / C3_TY, a.N * 3), block(64 * C3_NW);
  const char* e_dbg = getenv("FN2_CORR3_DBG");  // ablation bits (timing experiments; results are wrong when set)
  const int dbg = e_dbg ? atoi(e_dbg) : 0;
  switch (nl) {
#define FN2_C3_CASE(NL_)                                                               \
  case NL_:                                                                            \
    hipLaunchKernelGGL((corr3_kernel<T, OutT, NL_>), grid, block, 0, s, a, b_bytes, dbg);   \
    break;
    FN2_C3_CASE(1)
    FN2_C3_CASE(2)
    FN2_C3_CASE(4)
    FN2_C3_CASE(8)
#undef FN2_C3_CASE
    default:
      return fail(FN2_ERR_UNSUPPORTED, "correlation: unsupported channel count %d", C);
  }
  FN2_CHECK_LAUNCH("corr3");
  return FN2_OK;
}

// The FlowNetC attribute set on 16-bit / split-fp16 features, split-fp16
// outputs group-aligned.  FN2_CORR3=0 keeps corr2 (A/B).  Rows in same-parity quads: H % 8 == 0.
bool corr3_ok(int C, int in_dtype, int out_dtype, int md, int s2, int H, int out_cs, int out_c0, long b_bytes) {
  const char* e = getenv("FN2_CORR3");
  if (e && atoi(e) == 0) return false;
  if (in_dtype == FN2_F32) return false;
  const int esz = dtype_size(in_dtype);
  if ((C * esz) % 128 != 0) return false;
  const int nl = C * esz / 128;
  if (!(nl == 1 || nl == 2 || nl == 4 || nl == 8)) return false;
  if (md != 20 || s2 != 2 || H % (2 * C3_TY) != 0 || b_bytes >= (1L << 31)) return false;
  if (out_dtype == FN2_F16X2 && (out_cs % 8 != 0 || out_c0 % 8 != 0)) return false;
  return true;
}

int launch_corr3(const void* a, int a_cs, int a_c0, const void* b, int b_cs, int b_c0, void* out, int out_cs,
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
  if (in_dtype == FN2_BF16) return out_dtype == FN2_BF16 ? launch3<bf16_t, bf16_t>(g, C, bb, s) : launch3<bf16_t, float>(g, C, bb, s);
  if (in_dtype == FN2_F16) return out_dtype == FN2_F16 ? launch3<f16_t, f16_t>(g, C, bb, s) : launch3<f16_t, float>(g, C, bb, s);
  return out_dtype == FN2_F16X2 ? launch3<x2_t, x2_t>(g, C, bb, s) : launch3<x2_t, float>(g, C, bb, s);
}

}  // namespace fn2
