// Fast path of the implicit-GEMM convolution (gfx950): layers whose per-tap channel run is a
// multiple of 128 bytes (Cin_pad % 64 == 0 in bf16, % 32 in fp32) and Cout > 32.
//
//   D[cout][pixel] += W[cout][k] * X[pixel][k],  k = (tap, channel);  block tile 128 cout x 128 px
//   (or 64 x 256), 4 waves, each a 64 x 64 tile = 2 x 2 MFMA tiles of 32 x 32
//   (v_mfma_f32_32x32x16_bf16 / v_mfma_f32_32x32x2_f32).
//
// What the in-process ablations (tools/ab_conv.py, FN2_CONV_DBG) showed about the first version:
// the K loop was bound by vector-ALU ISSUE, not by the matrix pipe or by bytes -- ~100 address
// VALU per stage for the operand fetches plus 8 of every 16 cycles of each 16x16x32 MFMA.  Hence:
//   * operands go HBM/L2 -> LDS by LDS-DMA through BUFFER descriptors (buffer_load_dwordx4 ... lds):
//     a lane's byte offset inside the tensor is computed once, the per-stage motion (tap, channel
//     block) is a scalar offset, and out-of-image taps (the reference's explicit zero pad,
//     utils.py:408-412) / rows past the end of the pixel grid are an out-of-range offset that the
//     descriptor's range check turns into zeros: 4 VALU per pixel piece, 0 per weight piece;
//   * 32x32 MFMA tiles: half the matrix instructions per FLOP of the 16x16 shape;
//   * a stage is 128 bytes of channels per operand row, so it never straddles a filter tap (the tap
//     is wave-uniform scalar state) and every row is fetched as one full 128-byte line;
//   * the LDS image is lane-linear (8 rows x 128 B per wave instruction); the bank-conflict XOR
//     swizzle (phys chunk = chunk ^ ((row >> 1) & 7)) is applied to the SOURCE offset and again on
//     the ds_read_b128 fragment reads (conflict-free for the 32-row x 2-chunk operand pattern);
//   * 2-stage pipeline: the next stage's DMA is issued in front of this stage's MFMAs (issuing the
//     pieces between MFMA groups measured slower for both dtypes once the loop was clean), MFMA
//     fragments double-buffered in registers, one s_waitcnt vmcnt(0) + barrier per stage;
//   * weight rows are permuted by the host inside every group of 32 (packed row (r&3)+8(r>>2)+4h
//     <-> cout 16h+r) so that the 16 accumulator registers of a lane are 16 CONSECUTIVE output
//     channels: the epilogue stores 32-byte (bf16) / 64-byte (fp32) runs per pixel.
#include "conv_common.h"

#include <type_traits>
#include <algorithm>
#include <utility>

#ifndef FN2_X2_ORDER
#define FN2_X2_ORDER 0
#endif
// 1: the FN2_CONV_DBG ablation switches of the main loop (bits 1, 2: skip the pixel / weight DMA; 16: tap-outer K order)
// are compiled in (tools/build_variant.sh abl "-DFN2_CONV_ABLATE=1" for the experiments of DESIGN.md section 7);
// 0 (default): compiled out -- no scalar branches between the DMA pieces
#ifndef FN2_CONV_ABLATE
#define FN2_CONV_ABLATE 0
#endif
// 1 (default): raise the wave priority over the MFMA section of a stage (s_setprio): a wave that has its fragments
// issues its matrix instructions ahead of co-resident waves still issuing DMA pieces.  Together with the switch above
// +0.6..1.6 % on the large layers, +4.5 % on conv5_1, neutral on the ring layers (same-process A/B, tools/ab_conv.py)
#ifndef FN2_SETPRIO
#define FN2_SETPRIO 1
#endif

namespace fn2 {

typedef __attribute__((address_space(3))) void* lptr_t;

constexpr unsigned kOobOffset = 0x80000000u;  // >= num_records of every descriptor (tensors < 2 GiB)

typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));   // a 16-byte register quad as inline-asm operand

// raw buffer descriptor: base, stride 0, num_records bytes, raw dword format
__device__ __forceinline__ v4i_t make_rsrc(const void* base, int bytes) {
  const unsigned long long a = (unsigned long long)base;
  return v4i_t{(int)(a & 0xffffffffu), (int)((a >> 32) & 0xffffu), bytes, 0x00020000};
}

// 16 bytes per lane, global/L2 -> LDS (lane-linear from `lds`), as inline asm: the compiler's waitcnt pass then
// knows nothing about these LDS writes, and every wait on them is the explicit s_waitcnt in the pipeline code
// below.  (With the builtin, the pass protects each later ds_read itself; it proves independence only for
// distinct LDS objects, and in the 3-slot ring it still put a vmcnt(0) -- a wait for the stage issued a few
// instructions earlier -- at the loop header.)
__device__ __forceinline__ void dma16(const v4i_t rsrc, const void* lds, unsigned voff, int soff) {
  const unsigned la = (unsigned)(unsigned long long)(lptr_t)lds;
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :: "s"(la), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

// Border ring of a composed head (fn2_flow_head_ring, conv.hip head_ring_kernel) as extra blocks of the HEAD5 launch:
// block `rb` of the ring part takes 16 ring pixels, a wave four of them one after the other (wave per pixel: lanes stride
// the (tap, 8-channel group) items, shuffle reduction).
__device__ __forceinline__ void h5_ring_pixel(const ConvArgs& p, long m) {   // one wave: ring pixel m of N * ring
  const int lane = threadIdx.x & 63;
  const int H = p.OH, W = p.OW, ring = 2 * W + 2 * (H - 2);
  const int nitems = 25 * p.h5_groups;
  const size_t case_stride = (size_t)nitems * 16;
  const x2_t* in = reinterpret_cast<const x2_t*>(p.in);
  float* pf = reinterpret_cast<float*>(p.out);
  const int n = (int)(m / ring), r = (int)(m - (long)n * ring);
  int y, x;
  if (r < W) { y = 0; x = r; }
  else if (r < 2 * W) { y = H - 1; x = r - W; }
  else { const int q = r - 2 * W; y = 1 + (q >> 1); x = (q & 1) ? W - 1 : 0; }
  const int cy = y == 0 ? 0 : (y == H - 1 ? 2 : 1), cx = x == 0 ? 0 : (x == W - 1 ? 2 : 1);
  const int cs = 3 * cy + cx;
  const float* w = p.h5_wc + (size_t)cs * case_stride;
  float a0 = 0.f, a1 = 0.f;
  constexpr int U = 5;   // items per lane whose loads are in flight together (a pixel is 5 / 9 of them on 82 / 162 channels)
  for (int q0 = lane; q0 < nitems; q0 += 64 * U) {
    uint4 xs[U][2];
    float4 ws[U][4];
    bool ok[U];
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const int q = min(q0 + 64 * k, nitems - 1);
      const int tap = q / p.h5_groups, gi = q - tap * p.h5_groups;
      const int ky = tap / 5, kx = tap - ky * 5;
      const int iy = y + ky - 2, ix = x + kx - 2;
      ok[k] = q0 + 64 * k < nitems && iy >= 0 && iy < H && ix >= 0 && ix < W;
      const int cy2 = min(max(iy, 0), H - 1), cx2 = min(max(ix, 0), W - 1);
      const uint4* src = reinterpret_cast<const uint4*>(in + (((size_t)n * H + cy2) * W + cx2) * p.in_cs + p.in_c0 + gi * 8);
      xs[k][0] = src[0];
      xs[k][1] = src[1];
      const float4* u = reinterpret_cast<const float4*>(w + (size_t)q * 16);   // [ci 0..7][o 0..1]
#pragma unroll
      for (int j = 0; j < 4; ++j) ws[k][j] = u[j];
    }
#pragma unroll
    for (int k = 0; k < U; ++k) {
      float xv[8];
      join8(xs[k][0], xs[k][1], xv);
      if (ok[k]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          a0 += xv[2 * j] * ws[k][j].x;
          a1 += xv[2 * j] * ws[k][j].y;
          a0 += xv[2 * j + 1] * ws[k][j].z;
          a1 += xv[2 * j + 1] * ws[k][j].w;
        }
      }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    a0 += __shfl_xor(a0, off, 64);
    a1 += __shfl_xor(a1, off, 64);
  }
  if (lane == 0)
    *reinterpret_cast<float2*>(pf + (((size_t)n * H + y) * W + x) * 2) = make_float2(a0 + p.h5_bc[2 * cs], a1 + p.h5_bc[2 * cs + 1]);
}

__device__ __forceinline__ void h5_ring_block(const ConvArgs& p, int rb) {
  const int wave = threadIdx.x >> 6;
  const long total = (long)p.N * (2 * p.OW + 2 * (p.OH - 2));
  for (int k = 0; k < 4; ++k) {
    const long m = (long)rb * 16 + wave * 4 + k;
    if (m >= total) return;
    h5_ring_pixel(p, m);
  }
}

// HEAD5 epilogue: word w of the partial-sum table T, which spans the two stage buffers (two LDS objects of `half` bytes)
__device__ __forceinline__ float* h5_t(uint4* a, uint4* b, int half, int w) {
  const int byte = w * 4;
  return byte < half ? reinterpret_cast<float*>(reinterpret_cast<char*>(a) + byte)
                     : reinterpret_cast<float*>(reinterpret_cast<char*>(b) + (byte - half));
}

// TCN = 32-cout MFMA tiles per wave (2: the 128x128 / 64x256 blocks; 1: a 32-cout x 256-pixel block for the
// Cout <= 32 layers -- full-resolution fusion layers, conv_redir -- whose 64-cout tile was half or more padding).
// TPN = 32-pixel MFMA tiles per wave (1: a 128-cout x 64-pixel block for mid-size layers whose 128 x 128 grid
// would not fill the chip: twice the blocks instead of split-K slabs + a finalize launch).
// STAGES = 3: a 3-slot LDS ring with the DMA TWO stages ahead, for the weight-streaming layers (6x8 / 12x16
// levels): there every stage waits a full HBM round trip for weights nobody has touched yet, and one stage of
// look-ahead per block leaves the chip latency-bound (conv6_1: 38 MB of weights in 46 us).
// KG = K groups: a block of KG x 4 waves in which group g runs the stages kt0 + g, kt0 + g + KG, ... of the tile on
// its own pair of stage buffers, and the groups' accumulators are summed through LDS (fixed order) in front of the
// epilogue -- split-K inside the workgroup.  Used on the 6x8 / 12x16 levels (grids under 96 blocks of 128 x 64), where
// it halves the number of partial-sum slabs the split-K finalize pass has to read (policy and measurements: conv.hip,
// build_args).
// M16 = the 16x16x32 matrix instruction instead of 32x32x16 (split-fp16 operands only; A/B experiment of the guide's
// "the chip can hold a higher clock on one MFMA shape than on the other"): same LDS image, same fragment reads per
// FLOP; a lane then holds runs of 4 consecutive output channels instead of 16.
// WREG = the weight operand never touches LDS: the host stores it in MFMA-FRAGMENT order (wgt_layout 2: per 32-cout tile
// and 128-byte stage four 1 KiB blocks {q = 0, 1} x {hi, lo}, lane-linear, so a wave's fragment is ONE fully coalesced
// 16-byte-per-lane load) and every wave loads the fragments of its own 32 couts straight into registers one stage
// ahead; only the pixel rows go through LDS-DMA.  The wave layout that goes with it is WC = 4 x WP = 1 (a wave owns 32
// couts x the whole pixel tile: no weight byte is fetched twice).  Against the 2 x 2 layout of the same 128 x 64 tile:
// LDS-DMA pieces per wave and stage 6 -> 2 (their issue cost, ~60-180 cycles each, is what a wave spends besides its
// 384 cycles of MFMAs), LDS writes 24 -> 8 KB and fragment reads 48 -> 32 KB per stage, the same 24 KB through L1.
// HEAD5 = the composed 5x5 two-output flow head (interconvN + predict_flowN as one convolution, fn2_flow_head5 in
// flownet2_hip.h) on the 64-cout x 256-pixel tile: the block's 256 "pixels" are the 8 x 32 positions of a 4 x 28 output
// tile + 2 halo (out-of-image positions are zero rows), the GEMM is the 1x1 product to the 50 (tap, output) partials of
// every position, and the epilogue -- instead of storing them -- puts the partials in LDS (the stage buffers are free),
// sums the 25 shifted partials of each output pixel and writes the two flow channels.
// SLAB (WREG instantiations only) = the launch is a K split: the epilogue is the partial-sum slab store and nothing else
// (without it a WREG instantiation has no slab code at all); the other instantiations decide at run time.
template <typename T, typename OutT, int WC, int WP, int TCN = 2, int TPN = 2, int STAGES = 2, int KG = 1, bool M16 = false,
          bool WREG = false, bool SLAB = false, bool HEAD5 = false, bool FEED = false>
__global__ void __launch_bounds__(256 * KG + (FEED ? 256 : 0)) conv_igemm2_kernel(const ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)  // the buffer-descriptor type only exists in the device pass; the host pass needs just the stub
  constexpr int CH = 16 / (int)sizeof(T);
  constexpr int ESZ = (int)sizeof(T);
  constexpr int BC = WC * TCN * 32, BP = WP * TPN * 32;
  static_assert(WC * WP == 4, "4 waves per block");
  constexpr int NWI = BC / 32;  // weight-row DMA pieces per wave per stage (8 rows each)
  constexpr int NPI = BP / 32;  // pixel-row DMA pieces per wave per stage
  static_assert(!WREG || (is_x2<T>::value && TCN == 1 && KG == 1 && (STAGES == 2 || STAGES == 3) && !M16),
                "WREG: split fp16, one cout tile per wave, two-stage loop or 3-slot ring");
  constexpr int BOFF = WREG ? 0 : BC;           // first pixel row inside a stage buffer
  // WREG: the stage buffers hold pixel rows only (BP x 128 B each) and all of them live in ONE pool together with the
  // epilogue's wave tiles ([32 TPN pixels][TCN 32 + 4 words] + offsets per wave, two waves per half of the pool): the
  // pool is max(STAGES stage buffers, 4 wave tiles); ROWS = rows of HALF the pool (what the epilogue's region test sees)
  constexpr int kWTileBytes = (32 * TPN * (TCN * 32 + 4) + 32 * TPN) * 4;
  constexpr int kPoolHalfRows = (STAGES * BP * 128 / 2 > 2 * kWTileBytes ? STAGES * BP * 128 / 2 : 2 * kWTileBytes) / 128 + 1;
  constexpr int ROWS = WREG ? kPoolHalfRows : BC + BP;
  // two separate LDS objects, not lds[2][..]: the waitcnt pass only lets a ds_read run ahead of an in-flight
  // LDS-DMA when it can prove (alias scopes of distinct LDS variables) that they touch different objects; with
  // one array and a runtime buffer index it put s_waitcnt vmcnt(0) in front of every stage's first ds_read,
  // i.e. the "prefetch" of stage s+1 was waited for BEFORE the MFMAs of stage s.
  static_assert(KG == 1 || STAGES == 2, "K groups run the two-stage loop");
  // STAGES > 3: one ring of STAGES slots (the DMA is inline asm, so the compiler's waitcnt pass sees no LDS writes and a
  // runtime slot index costs nothing); the epilogue's scratch tiles then live in its first two slots
  constexpr bool kDeep = STAGES > 3;
  __shared__ uint4 lds0_all[(kDeep ? STAGES : WREG ? 2 : KG) * ROWS * 8];
  __shared__ uint4 lds1_all[(kDeep || WREG) ? 1 : KG * ROWS * 8];
  __shared__ uint4 lds2_own[(STAGES == 3 && !WREG) ? ROWS * 8 : 1];

  if constexpr (HEAD5) {
    if ((int)blockIdx.x >= p.h5_tiles) {  // the ring part of the grid (block-uniform)
      h5_ring_block(p, (int)blockIdx.x - p.h5_tiles);
      return;
    }
  }
  if constexpr (WREG && SLAB && is_x2<T>::value) {
    if (p.fh_M > 0 && (int)blockIdx.z >= p.fh_z0) {
      // predict_flow(N+1) riding on deconvN (fn2_conv_desc.head): the z slices behind the convolution's own are head
      // blocks, a pixel at a time each (they are dispatched last and fill the slots the convolution leaves free)
      if (FEED && threadIdx.x >= 256) return;   // (fh_pixel is written for 256 threads; ended waves do not count at its barriers)
      float* part = reinterpret_cast<float*>(lds0_all);
      const long e = (((long)blockIdx.z - p.fh_z0) * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
      const long total = (long)(gridDim.z - p.fh_z0) * gridDim.y * gridDim.x;
      for (long m = e; m < p.fh_M; m += total)
        fh_pixel(reinterpret_cast<const x2_t*>(p.in), p.H, p.W, p.in_cs, p.in_c0, p.fh_groups, p.fh_w, p.fh_w + p.fh_w1,
                 p.fh_bias, p.fh_scale, p.fh_out + (size_t)m * p.fh_out_cs + p.fh_out_c0, m, part);
      return;
    }
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = KG > 1 ? wave_all >> 2 : 0;   // K group of this wave
  // FEED (experiment, WREG ring only): waves 4 .. 7 issue the pixel DMA pieces of wave (w - 4)'s rows and nothing else; waves
  // 0 .. 3 load their weight fragments and multiply (the ~150 cycles a wave spends per LDS-DMA piece then run beside the
  // MFMAs on the SIMD's other wave slot, as in head5_strip_kernel)
  static_assert(!FEED || (WREG && STAGES == 3 && KG == 1), "FEED: the WREG ring");
  const bool feeder = FEED && wave_all >= 4;
  const int wave = (KG > 1 || FEED) ? wave_all & 3 : wave_all;
  uint4* const lds0 = lds0_all + grp * (ROWS * 8);
  uint4* const lds1 = WREG ? lds0_all + BP * 8 : kDeep ? lds0_all + ROWS * 8 : lds1_all + grp * (ROWS * 8);
  uint4* const lds2 = WREG ? lds0_all + 2 * BP * 8 : lds2_own;
  uint4* const ep1 = (kDeep || WREG) ? lds0_all + ROWS * 8 : lds1_all;   // second scratch region of the epilogue
  const int wc = wave / WP, wp = wave % WP;

  int pad_y = p.pad, pad_x = p.pad, oy_off = 0, ox_off = 0, osc = 1;
  const T* wgt = reinterpret_cast<const T*>(p.wgt);
  const int phase = blockIdx.z / p.splitk, split = blockIdx.z - phase * p.splitk;
  const unsigned wrow_bytes = (unsigned)p.ksteps * 128u;
  if (p.deconv) {
    const int a = phase >> 1, b = phase & 1;
    pad_y = a ? p.ph_pad1 : p.ph_pad0; pad_x = b ? p.ph_pad1 : p.ph_pad0; oy_off = a; ox_off = b; osc = 2;
    wgt += (size_t)phase * p.cout_pad * (wrow_bytes / ESZ);
  }
  // taps this block walks: the phases of a transposed stride-2 conv have fewer than KH x KW (zero slots are skipped)
  const int khi = p.deconv ? p.kh_ph[phase >> 1] : p.KH, kwi = p.deconv ? p.kw_ph[phase & 1] : p.KW;
  const int nst_all = (p.cin_chunks >> 3) * khi * kwi;
  const int kt0 = split * p.kper + grp;  // first stage of this K group
  const int kt1 = (FN2_CONV_ABLATE && (p.dbg & 1048576)) ? kt0 : min(nst_all, split * p.kper + p.kper);  // ablation: no K loop
  // XCD-aware tile order.  Workgroups are dealt round-robin over the 8 XCDs (each with its own L2) in linear id
  // order, so with the plain mapping a pixel tile's vertical neighbours and its other cout tiles run on other
  // XCDs / much later: every 3x3 halo row and every extra cout tile re-read the activations from HBM.  Give
  // XCD i a contiguous band of the (pixel tile, cout tile) space, cout tile fastest: consecutive blocks of one
  // XCD share the input tile, then move to the adjacent one (halo rows still in that XCD's L2).
  int bx = blockIdx.x, by = blockIdx.y;
  if (!(p.dbg & 4)) {
    const int NT = HEAD5 ? p.h5_tiles : gridDim.x * gridDim.y, L = blockIdx.x + gridDim.x * blockIdx.y;  // (HEAD5: the tile part of the grid)
    const int xcd = L & 7, chunk = NT >> 3, rem = NT & 7;
    const int Lp = xcd * chunk + min(xcd, rem) + (L >> 3);
    if (p.wmajor) {
      // weight-streaming layers (the 6x8 / 12x16 levels: tens of MB of weights, under 2 MB of activations): the
      // pixel tiles of one cout tile run side by side on one XCD, so a weight line crosses the fabric once instead
      // of once per XCD that holds one of its pixel tiles; the small input is what gets re-read by every XCD
      by = Lp / (int)gridDim.x;
      bx = Lp - by * (int)gridDim.x;
    } else {
      bx = Lp / (int)gridDim.y;
      by = Lp - bx * (int)gridDim.y;
    }
  }
  const int m0 = bx * BP;
  const int c0 = by * BC;

  // buffer descriptors: base, stride 0, num_records bytes, raw dword format
  const v4i_t rsrc_w = make_rsrc(wgt, (int)(p.cout_pad * wrow_bytes));
  const v4i_t rsrc_x = make_rsrc(p.in, p.in_bytes);

  // ---- per-lane DMA offsets.  Lane -> (row lane>>3 of the piece's 8 rows, physical chunk lane&7)
  const int lrow = lane >> 3, lphys = lane & 7;
  unsigned woff[NWI];
#pragma unroll
  for (int j = 0; j < NWI; ++j) {
    const int row = wave * (BC / 4) + j * 8 + lrow;
    woff[j] = (unsigned)(c0 + row) * wrow_bytes + (unsigned)((lphys ^ ((row >> 1) & 7)) * 16);
  }
  // pixel rows: byte offset of (n, iy0, ix0, chunk) -- may be "negative" while the tap is out of the
  // image -- and a validity mask: bit ky: 0 <= iy0+ky < H; bit 8+kx: 0 <= ix0+kx < W
  int roff[NPI];
  unsigned vmask[NPI];
#pragma unroll
  for (int j = 0; j < NPI; ++j) {
    const int row = wave * (BP / 4) + j * 8 + lrow;
    const int m = m0 + row;
    bool v = m < p.M;
    const int mm = v ? m : 0;
    int n = mm / (p.OH * p.OW);
    const int rem = mm - n * (p.OH * p.OW);
    int oy = rem / p.OW, ox = rem - oy * p.OW;
    if constexpr (HEAD5) {  // row = position (ry, rx) of the block's 8 x 32 window; bx = (n, tile row, tile column)
      static_assert(BP == 256, "HEAD5: 8 x 32 positions per block");
      const int tx = bx % p.h5_tx, tyn = bx / p.h5_tx;
      n = tyn / p.h5_ty;
      oy = (tyn - n * p.h5_ty) * 4 - 2 + (row >> 5);
      ox = tx * 28 - 2 + (row & 31);
      v = true;  // (positions outside the image fail the range test below: zero rows)
    }
    const int iy0 = oy * p.stride - pad_y, ix0 = ox * p.stride - pad_x;
    roff[j] = (((n * p.H + iy0) * p.W + ix0) * p.in_cs + p.in_c0) * ESZ + (lphys ^ ((row >> 1) & 7)) * 16;
    unsigned mk = 0;
    if (v) {
      for (int k = 0; k < p.KH; ++k) mk |= (unsigned)(iy0 + k >= 0 && iy0 + k < p.H) << k;
      for (int k = 0; k < p.KW; ++k) mk |= (unsigned)(ix0 + k >= 0 && ix0 + k < p.W) << (8 + k);
    }
    vmask[j] = mk;
  }
  // wave-uniform state of stage kt0: tap (ky, kx) and 128-byte block `sc` inside the tap's channel run
  // K order: 128-byte channel block OUTER, filter tap INNER.  The taps of one channel block re-touch the same
  // 128-byte lines of neighbouring pixels within KH*KW consecutive stages, so a block's live set is one line per
  // halo pixel (tens of KB) and stays in its XCD's L2; with the tap outer (the first version) the re-touch
  // distance was a whole pass over the channels and every tap missed L2: FETCH_SIZE showed the 3x3 layers
  // fetching their input 6-9 times (conv3_1: 449 MB for a 47 MB input).  Only the summation order changes.
  const int spt = p.cin_chunks >> 3;  // 128-byte stages per tap
  const int ntap = khi * kwi;
  const bool tap_outer = FN2_CONV_ABLATE && (p.dbg & 16);  // A/B switch: the first version's order
  int sc = tap_outer ? kt0 % spt : kt0 / ntap;  // channel block of stage kt0
  int tap = tap_outer ? kt0 / spt : kt0 - sc * ntap;
  int ky = tap / kwi, kx = tap - ky * kwi;
  int wstage = kt0;  // position in this order; the weight row offset of the stage is ((ky*KW + kx)*spt + sc) * 128

  auto issue_piece = [&](auto piece_c, uint4* lds) {
    constexpr int i = decltype(piece_c)::value;
    if constexpr (i < NWI) {
      if constexpr (!WREG) {
        if (!(FN2_CONV_ABLATE && (p.dbg & 2)))
          dma16(rsrc_w, &lds[(wave * (BC / 4) + i * 8) * 8], woff[i], ((ky * p.KW + kx) * spt + sc) * 128);
      }
    } else {
      constexpr int j = i - NWI;
      const unsigned tbit = (1u << ky) | (1u << (8 + kx));
      const int toff = ((ky * p.W + kx) * p.in_cs + sc * 8 * CH) * ESZ;
      const unsigned voff = ((vmask[j] & tbit) == tbit && wstage < kt1) ? (unsigned)(roff[j] + toff) : kOobOffset;
      if (!(FN2_CONV_ABLATE && (p.dbg & 1)))
        dma16(rsrc_x, &lds[(BOFF + wave * (BP / 4) + j * 8) * 8], voff, 0);
    }
  };
  // WREG: this wave's weight fragments of one stage, loaded by inline asm (the compiler must neither wait for them nor
  // count them: guide 5.7 item 1, form ii -- every consumer sits behind a wait statement that names the registers)
  u32x4_t wfa[WREG ? 4 : 1], wfb[WREG ? 4 : 1], wfc[(WREG && STAGES == 3) ? 4 : 1];
  const unsigned wlane = (unsigned)lane * 16u;
  const int wtile_off = WREG ? ((c0 >> 5) + wc) * (p.ksteps * 4096) : 0;   // byte offset of this wave's 32-cout tile
  auto load_w = [&](u32x4_t (&wf)[WREG ? 4 : 1]) {
    if constexpr (WREG) {
      const int soff = wtile_off + ((ky * p.KW + kx) * spt + sc) * 4096;
      asm volatile("buffer_load_dwordx4 %0, %4, %5, %6 offen\n\t"
                   "buffer_load_dwordx4 %1, %4, %5, %6 offen offset:1024\n\t"
                   "buffer_load_dwordx4 %2, %4, %5, %6 offen offset:2048\n\t"
                   "buffer_load_dwordx4 %3, %4, %5, %6 offen offset:3072"
                   : "=&v"(wf[0]), "=&v"(wf[1]), "=&v"(wf[2]), "=&v"(wf[3])
                   : "v"(wlane), "s"(rsrc_w), "s"(soff) : "memory");
    }
  };
  auto advance1 = [&]() {
    ++wstage;
    if (tap_outer) {
      if (++sc == spt) {
        sc = 0;
        if (++kx == kwi) { kx = 0; ++ky; }
      }
      if (ky == khi) { ky = 0; sc = spt; }  // past the end: park on an always-masked state
    } else if (++kx == kwi) {
      kx = 0;
      if (++ky == khi) { ky = 0; ++sc; }
    }
  };
  auto advance = [&]() {
#pragma unroll
    for (int g = 0; g < KG; ++g) advance1();  // this group's next stage
  };
  auto issue_stage = [&](uint4* lds) {
    [&]<int... I>(std::integer_sequence<int, I...>) {
      (issue_piece(std::integral_constant<int, I>{}, lds), ...);
    }(std::make_integer_sequence<int, NWI + NPI>{});
    advance();
  };
  auto issue_stage_w = [&](uint4* lds, u32x4_t (&wf)[WREG ? 4 : 1]) {   // WREG: pixel DMA + this wave's weight fragments
    [&]<int... I>(std::integer_sequence<int, I...>) {
      (issue_piece(std::integral_constant<int, I>{}, lds), ...);
    }(std::make_integer_sequence<int, NWI + NPI>{});
    load_w(wf);
    advance();
  };
  auto wait_w = [&](u32x4_t (&wf)[WREG ? 4 : 1]) {   // everything in flight has landed; the fragments may be read from here on
    if constexpr (WREG)
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(wf[0]), "+v"(wf[1]), "+v"(wf[2]), "+v"(wf[3]) :: "memory");
  };

  // ---- fragment addresses: row r = l&31 of a 32-row tile, chunk 2*ks + (l>>5), swizzled by the row
  const int fr = lane & 31, fh = lane >> 5, fsw = (fr >> 1) & 7;
  f32x16 acc[TCN][TPN];
#pragma unroll
  for (int tc = 0; tc < TCN; ++tc)
#pragma unroll
    for (int tp = 0; tp < TPN; ++tp)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[tc][tp][q] = 0.f;
  // M16: 16 x 16 sub-tiles (sc, sp) of the wave's (TCN*32) x (TPN*32) tile
  f32x4 acc16[M16 ? TCN * 2 : 1][M16 ? TPN * 2 : 1];
  if constexpr (M16) {
#pragma unroll
    for (int a = 0; a < TCN * 2; ++a)
#pragma unroll
      for (int b = 0; b < TPN * 2; ++b) acc16[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // one stage of MFMAs on the LDS object `lds`
  auto compute = [&](const uint4* lds) {
    if (FN2_CONV_ABLATE && (p.dbg & 131072)) return;  // ablation: no LDS reads, no MFMAs
#if FN2_SETPRIO
    __builtin_amdgcn_s_setprio(1);
#endif
    const uint4* A = &lds[(wc * TCN * 32 + fr) * 8];
    const uint4* B = &lds[(BC + wp * TPN * 32 + fr) * 8];
    if constexpr (is_x2<T>::value && M16) {
      // 16x16x32: lane (row l & 15, channel group g = l >> 4) reads the hi chunk 2g and the lo chunk 2g + 1 of its
      // row; one instruction covers the stage's 32 channels
      const int r16 = lane & 15, g16 = lane >> 4;
      uint4 ah[TCN * 2], al[TCN * 2], bh[TPN * 2], bl[TPN * 2];
#pragma unroll
      for (int t = 0; t < TCN * 2; ++t) {
        const int row = wc * TCN * 32 + t * 16 + r16, sw = (row >> 1) & 7;
        ah[t] = lds[row * 8 + ((2 * g16) ^ sw)];
        al[t] = lds[row * 8 + ((2 * g16 + 1) ^ sw)];
      }
#pragma unroll
      for (int t = 0; t < TPN * 2; ++t) {
        const int row = BC + wp * TPN * 32 + t * 16 + r16, sw = (row >> 1) & 7;
        bh[t] = lds[row * 8 + ((2 * g16) ^ sw)];
        bl[t] = lds[row * 8 + ((2 * g16 + 1) ^ sw)];
      }
#pragma unroll
      for (int a = 0; a < TCN * 2; ++a)
#pragma unroll
        for (int b = 0; b < TPN * 2; ++b) {
          acc16[a][b] = mfma_16x16x32<f16_t>(al[a], bh[b], acc16[a][b]);
          acc16[a][b] = mfma_16x16x32<f16_t>(ah[a], bl[b], acc16[a][b]);
          acc16[a][b] = mfma_16x16x32<f16_t>(ah[a], bh[b], acc16[a][b]);
        }
    } else if constexpr (is_x2<T>::value) {
      // split fp16: the 128-byte row is [hi g0 | lo g0 | hi g1 | lo g1 | hi g2 | lo g2 | hi g3 | lo g3]
      // (4 groups of 8 channels).  One 32x32x16 product covers groups (2q, 2q+1): lane half h owns
      // group 2q+h, i.e. chunks 4q+2h (hi) and 4q+2h+1 (lo);  x*w = hi*hi + hi*lo + lo*hi.
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int chh = (4 * q + 2 * fh) ^ fsw, chl = (4 * q + 2 * fh + 1) ^ fsw;
        uint4 ah[TCN], al[TCN], bh[TPN], bl[TPN];
#pragma unroll
        for (int t = 0; t < TCN; ++t) { ah[t] = A[t * 32 * 8 + chh]; al[t] = A[t * 32 * 8 + chl]; }
#pragma unroll
        for (int t = 0; t < TPN; ++t) { bh[t] = B[t * 32 * 8 + chh]; bl[t] = B[t * 32 * 8 + chl]; }
#pragma unroll
        for (int tc = 0; tc < TCN; ++tc)
#pragma unroll
          for (int tp = 0; tp < TPN; ++tp) {
            acc[tc][tp] = mfma_32x32x16<f16_t>(al[tc], bh[tp], acc[tc][tp]);
            acc[tc][tp] = mfma_32x32x16<f16_t>(ah[tc], bl[tp], acc[tc][tp]);
            acc[tc][tp] = mfma_32x32x16<f16_t>(ah[tc], bh[tp], acc[tc][tp]);
          }
      }
    } else {
      // fragments of k-step ks+1 are read while the MFMAs of k-step ks run (register double buffer)
      uint4 fa[2][TCN], fb[2][TPN];
      {
        const int ch = fh ^ fsw;
#pragma unroll
        for (int t = 0; t < TCN; ++t) fa[0][t] = A[t * 32 * 8 + ch];
#pragma unroll
        for (int t = 0; t < TPN; ++t) fb[0][t] = B[t * 32 * 8 + ch];
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int cur = ks & 1, nxt = cur ^ 1;
        if (ks + 1 < 4) {
          const int ch = ((ks + 1) * 2 + fh) ^ fsw;
#pragma unroll
          for (int t = 0; t < TCN; ++t) fa[nxt][t] = A[t * 32 * 8 + ch];
#pragma unroll
          for (int t = 0; t < TPN; ++t) fb[nxt][t] = B[t * 32 * 8 + ch];
        }
        if constexpr (sizeof(T) == 2) {
#pragma unroll
          for (int tc = 0; tc < TCN; ++tc)
#pragma unroll
            for (int tp = 0; tp < TPN; ++tp)
              acc[tc][tp] = mfma_32x32x16<T>(fa[cur][tc], fb[cur][tp], acc[tc][tp]);
        } else {
          // chunk = 4 floats; MFMA j takes element j of both operands (a permutation of k shared by both)
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int tc = 0; tc < TCN; ++tc)
#pragma unroll
              for (int tp = 0; tp < TPN; ++tp)
                acc[tc][tp] = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(f32x4, fa[cur][tc])[j],
                                                                   __builtin_bit_cast(f32x4, fb[cur][tp])[j],
                                                                   acc[tc][tp], 0, 0, 0);
        }
      }
    }
#if FN2_SETPRIO
    __builtin_amdgcn_s_setprio(0);
#endif
  };

  // WREG: the A fragments of the stage are the registers wf[2 q + {0: hi, 1: lo}]; B as above
  auto compute_w = [&](const uint4* lds, const u32x4_t (&wf)[WREG ? 4 : 1]) {
    if constexpr (WREG) {
#if FN2_SETPRIO
      __builtin_amdgcn_s_setprio(1);
#endif
      const uint4* B = &lds[(BOFF + wp * TPN * 32 + fr) * 8];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int chh = (4 * q + 2 * fh) ^ fsw, chl = (4 * q + 2 * fh + 1) ^ fsw;
        uint4 bh[TPN], bl[TPN];
#pragma unroll
        for (int t = 0; t < TPN; ++t) { bh[t] = B[t * 32 * 8 + chh]; bl[t] = B[t * 32 * 8 + chl]; }
#pragma unroll
        for (int tp = 0; tp < TPN; ++tp) {
          const uint4 ah = __builtin_bit_cast(uint4, wf[2 * q]), al = __builtin_bit_cast(uint4, wf[2 * q + 1]);
          acc[0][tp] = mfma_32x32x16<f16_t>(al, bh[tp], acc[0][tp]);
          acc[0][tp] = mfma_32x32x16<f16_t>(ah, bl[tp], acc[0][tp]);
          acc[0][tp] = mfma_32x32x16<f16_t>(ah, bh[tp], acc[0][tp]);
        }
      }
#if FN2_SETPRIO
      __builtin_amdgcn_s_setprio(0);
#endif
    }
  };

  // NOTE: nothing conditional may wrap the MFMAs: an `if` around them made hipcc shuttle all 64
  // accumulators between VGPRs and AGPRs four times per stage (256 v_accvgpr moves, the dominant VALU
  // cost of the first version of this loop -- found with SQ_INSTS_VALU and the .s).
  // Two stages per trip so that every LDS access names its object statically (see the lds0/lds1 note), and no
  // branch between them: an odd stage count is rounded up with a stage whose pixel rows are all zero (the
  // validity test in issue_piece fails for stages >= kt1), so its MFMAs add 0 * stale finite weights.
  if constexpr (FEED) {
    const int nst3 = (kt1 - kt0 + 2) / 3 * 3;
    auto pix_stage = [&](uint4* lds) {   // this feeder's pixel pieces of the next stage
      [&]<int... I>(std::integer_sequence<int, I...>) {
        (issue_piece(std::integral_constant<int, NWI + I>{}, lds), ...);
      }(std::make_integer_sequence<int, NPI>{});
      advance();
    };
    if (feeder) {
      static_assert(NPI == 2, "vmcnt literal below");
      pix_stage(lds0);
      pix_stage(lds1);
      for (int s = 0; s < nst3; s += 3) {
        asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory"); pix_stage(lds2);
        asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory"); pix_stage(lds0);
        asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory"); pix_stage(lds1);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      auto w_stage = [&](u32x4_t (&wf)[4]) { load_w(wf); advance(); };
      auto sync_c = [&](u32x4_t (&wf)[4]) {   // this wave's fragments of the stage have landed (the next stage's four loads may be in flight)
        asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" : "+v"(wf[0]), "+v"(wf[1]), "+v"(wf[2]), "+v"(wf[3]) :: "memory");
      };
      w_stage(wfa);
      w_stage(wfb);
      for (int s = 0; s < nst3; s += 3) {
        sync_c(wfa); w_stage(wfc); compute_w(lds0, wfa);
        sync_c(wfb); w_stage(wfa); compute_w(lds1, wfb);
        sync_c(wfc); w_stage(wfb); compute_w(lds2, wfc);
      }
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(wfa[0]), "+v"(wfa[1]), "+v"(wfa[2]), "+v"(wfa[3]), "+v"(wfb[0]), "+v"(wfb[1]),
                   "+v"(wfb[2]), "+v"(wfb[3]) :: "memory");
    }
  } else if constexpr (WREG && STAGES == 3) {
    // 3-slot ring of pixel stages (8 KB each) + three register sets of weight fragments: per stage, wait until all but the
    // newest stage's 6 vector-memory instructions of this wave (2 DMA pieces + 4 fragment loads) are done, bare barrier,
    // refill the slot and the register set of the stage computed last with the stage two ahead
    static_assert(NPI + 4 == 6, "vmcnt literal below");
    auto ring_sync_w = [&](u32x4_t (&wf)[4]) {
      asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" : "+v"(wf[0]), "+v"(wf[1]), "+v"(wf[2]), "+v"(wf[3]) :: "memory");
    };
    issue_stage_w(lds0, wfa);
    issue_stage_w(lds1, wfb);
    const int nst3 = (kt1 - kt0 + 2) / 3 * 3;
    for (int s = 0; s < nst3; s += 3) {
      ring_sync_w(wfa); issue_stage_w(lds2, wfc); compute_w(lds0, wfa);
      ring_sync_w(wfb); issue_stage_w(lds0, wfa); compute_w(lds1, wfb);
      ring_sync_w(wfc); issue_stage_w(lds1, wfb); compute_w(lds2, wfc);
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(wfa[0]), "+v"(wfa[1]), "+v"(wfa[2]), "+v"(wfa[3]), "+v"(wfb[0]), "+v"(wfb[1]),
                 "+v"(wfb[2]), "+v"(wfb[3]) :: "memory");  // the look-ahead stages (zero pixel rows) have landed before the registers die
  } else if constexpr (WREG) {
    issue_stage_w(lds0, wfa);
    wait_w(wfa);
    __syncthreads();
    const int nst2 = (kt1 - kt0 + 1) & ~1;
    for (int s = 0; s < nst2; s += 2) {
      issue_stage_w(lds1, wfb);
      compute_w(lds0, wfa);
      wait_w(wfb);
      __syncthreads();
      if (s + 2 < nst2) issue_stage_w(lds0, wfa);
      compute_w(lds1, wfb);
      wait_w(wfa);
      __syncthreads();
    }
  } else if constexpr (STAGES == 2) {
    // (On the MFMA-bound layers a 3-slot ring was measured 20-30 % slower with 128 x 128 tiles: 96 KB of LDS = one
    // block per CU, and losing the second block's MFMAs under this block's waits costs more than the deeper
    // prefetch gains.)
    issue_stage(lds0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // trip count of group 0 (the longest), rounded up to even; groups that run out of stages add zero stages
    const int nst2 = ((kt1 - (kt0 - grp) + KG - 1) / KG + 1) & ~1;
    for (int s = 0; s < nst2; s += 2) {
      issue_stage(lds1);  // next stage's DMA in flight under this stage's MFMAs
      compute(lds0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (s + 2 < nst2) issue_stage(lds0);
      compute(lds1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  } else if constexpr (kDeep) {
    // Deep ring for grids of about one block per CU (a single resident block has nothing else to hide its DMA round
    // trips behind): STAGES slots, D = STAGES - 1 stages of DMA in flight, counted vmcnt -- never 0 inside the loop --
    // and a bare s_barrier per stage.  Iteration s: wait until this wave's pieces of stage s have landed (all but the
    // 6 (D - 1) youngest), barrier (every wave's pieces of stage s are in LDS, and every wave has finished reading the
    // slot of stage s - 1), refill that slot with stage s + D, compute stage s.  Stages past the end are zero stages
    // (out-of-range pixel rows), so the loop has no tail code; they are drained after it.
    constexpr int ND = NWI + NPI, D = STAGES - 1;
    static_assert(ND * (D - 1) <= 63, "vmcnt field");
    auto slot_ptr = [&](int sl) { return lds0_all + sl * (ROWS * 8); };
#pragma unroll
    for (int d = 0; d < D; ++d) issue_stage(slot_ptr(d));
    const int nst = kt1 - kt0;
    int rd = 0, wr = D;   // slot read by this iteration / refilled in it
    for (int s = 0; s < nst; ++s) {
      asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(ND * (D - 1)) : "memory");
      issue_stage(slot_ptr(wr));
      compute(slot_ptr(rd));
      rd = rd + 1 == STAGES ? 0 : rd + 1;
      wr = wr + 1 == STAGES ? 0 : wr + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    // Ring: per stage, wait until all but the newest stage's ND DMA instructions of this wave have landed, bare
    // s_barrier (every wave's rows of stage s are in LDS, and every wave is done reading slot (s+2)%3, the slot of
    // stage s-1: its ds_reads were consumed by MFMAs before the barrier), refill that slot two stages ahead.
    // __syncthreads() would not do: its fence drains vmcnt to 0, i.e. waits for the look-ahead stage as well.
    constexpr int ND = NWI + NPI;
    static_assert(ND == 5 || ND == 6 || ND == 8, "vmcnt literal below");
    auto ring_sync = [] {
      if constexpr (ND == 5) asm volatile("s_waitcnt vmcnt(5)\n\ts_barrier" ::: "memory");
      else if constexpr (ND == 6) asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
    };
    issue_stage(lds0);
    issue_stage(lds1);
    const int nst3 = (kt1 - kt0 + 2) / 3 * 3;
    for (int s = 0; s < nst3; s += 3) {
      ring_sync(); issue_stage(lds2); compute(lds0);
      ring_sync(); issue_stage(lds0); compute(lds1);
      ring_sync(); issue_stage(lds1); compute(lds2);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the look-ahead stages (all-zero pixel rows past kt1)
  }

  if constexpr (KG > 1) {
    // sum the K groups through LDS (every wave is past the loop's last barrier: the stage buffers are free), group 1
    // first, then group 2: a fixed order, so the result does not depend on timing
    constexpr int NA4 = TCN * TPN * 4;  // float4 of accumulators per lane (the same 16 x TCN x TPN floats in either MFMA shape)
    static_assert((KG - 1) * 4 * NA4 * 64 <= KG * ROWS * 8, "reduction scratch must fit the stage buffers");
    float4* red = reinterpret_cast<float4*>(lds0_all);
    auto acc4 = [&](int idx) -> float4 {  // idx-th float4 of this lane's accumulators
      if constexpr (M16) {
        const f32x4 v = acc16[idx / (TPN * 2)][idx % (TPN * 2)];
        return make_float4(v[0], v[1], v[2], v[3]);
      } else {
        const int t = idx >> 2, q = idx & 3;
        return make_float4(acc[t / TPN][t % TPN][4 * q], acc[t / TPN][t % TPN][4 * q + 1], acc[t / TPN][t % TPN][4 * q + 2],
                           acc[t / TPN][t % TPN][4 * q + 3]);
      }
    };
    if (grp > 0) {
#pragma unroll
      for (int idx = 0; idx < NA4; ++idx) red[(((grp - 1) * 4 + wave) * NA4 + idx) * 64 + lane] = acc4(idx);
    }
    __syncthreads();
    if (grp > 0) return;
#pragma unroll
    for (int g = 0; g < KG - 1; ++g)
#pragma unroll
      for (int idx = 0; idx < NA4; ++idx) {
        const float4 v = red[((g * 4 + wave) * NA4 + idx) * 64 + lane];
        if constexpr (M16) {
          f32x4& d = acc16[idx / (TPN * 2)][idx % (TPN * 2)];
          d[0] += v.x; d[1] += v.y; d[2] += v.z; d[3] += v.w;
        } else {
          const int t = idx >> 2, q = idx & 3;
          acc[t / TPN][t % TPN][4 * q] += v.x; acc[t / TPN][t % TPN][4 * q + 1] += v.y;
          acc[t / TPN][t % TPN][4 * q + 2] += v.z; acc[t / TPN][t % TPN][4 * q + 3] += v.w;
        }
      }
  }

  if constexpr (HEAD5) {
    if (p.dbg & 2) return;  // (ablation: K loop only)
    // partials -> LDS as T[position][52] (50 used: (uy * 5 + ux) * 2 + o), then the 25-tap sums.  The two-stage loop ended
    // with a barrier, so the stage buffers (2 x (64 + 256) x 128 B) are free.
    constexpr int TS = 52;
    static_assert(256 * TS * 4 <= 2 * ROWS * 128, "T fits the stage buffers");
#pragma unroll
    for (int tc = 0; tc < TCN; ++tc)
#pragma unroll
      for (int tp = 0; tp < TPN; ++tp) {
        const int pos = wp * TPN * 32 + tp * 32 + fr;
        const int cb = tc * 32 + fh * 16;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (cb + 4 * q < TS)
            *reinterpret_cast<float4*>(h5_t(lds0_all, lds1_all, ROWS * 128, pos * TS + cb + 4 * q)) =
                make_float4(acc[tc][tp][4 * q] * p.out_scale, acc[tc][tp][4 * q + 1] * p.out_scale,
                            acc[tc][tp][4 * q + 2] * p.out_scale, acc[tc][tp][4 * q + 3] * p.out_scale);
      }
    __syncthreads();
    if (p.dbg & 8) return;  // (ablation: no gather)
    const int tx = bx % p.h5_tx, tyn = bx / p.h5_tx;
    const int n = tyn / p.h5_ty, ty = tyn - n * p.h5_ty;
    float* pf = reinterpret_cast<float*>(p.out);
    if (tid < 224) {  // 4 x 28 output pixels x 2 flow channels
      const int o = tid & 1, px = tid >> 1;
      const int py = px / 28, pxx = px - py * 28;
      const int oy = ty * 4 + py, ox = tx * 28 + pxx;
      if (oy < p.OH && ox < p.OW && !(p.h5_ring && (oy == 0 || oy == p.OH - 1 || ox == 0 || ox == p.OW - 1))) {
        float a = p.bias ? p.bias[o] : 0.f;
#pragma unroll
        for (int uy = 0; uy < 5; ++uy)
#pragma unroll
          for (int ux = 0; ux < 5; ++ux)
            a += *h5_t(lds0_all, lds1_all, ROWS * 128, ((py + uy) * 32 + pxx + ux) * TS + (uy * 5 + ux) * 2 + o);
        pf[(((size_t)n * p.OH + oy) * p.OW + ox) * 2 + o] = a;
      }
    }
    return;
  }
  if (FN2_CONV_ABLATE && (p.dbg & 262144)) return;  // ablation: no epilogue
  if constexpr (M16) {
    // D of a 16 x 16 sub-tile: column (pixel) = lane & 15, rows (packed weight rows) 4 (lane >> 4) + j.  With the row
    // permutation of the 32 x 32 layout (packed row (r & 3) + 8 (r >> 2) + 4 h <-> cout 16 h + r) the four rows of a
    // lane are the consecutive couts 16 (g & 1) + 8 s + 4 (g >> 1) + j of their 32-cout group (s = sub-tile parity).
    OutT* out16 = reinterpret_cast<OutT*>(p.out);
    const int r16 = lane & 15, g16 = lane >> 4;
    float* slab = p.splitk > 1 ? p.ws + (size_t)split * p.N * p.out_H * p.out_W * p.ws_cs : nullptr;
#pragma unroll
    for (int b = 0; b < TPN * 2; ++b) {
      const int m = m0 + wp * TPN * 32 + b * 16 + r16;
      if (m >= p.M) continue;
      const int n = m / (p.OH * p.OW);
      const int rem = m - n * (p.OH * p.OW);
      const int oy = rem / p.OW, ox = rem - oy * p.OW;
      const size_t opix = ((size_t)n * p.out_H + (oy * osc + oy_off)) * p.out_W + (ox * osc + ox_off);
      OutT* po = out16 + opix * p.out_cs + p.out_c0;
#pragma unroll
      for (int a = 0; a < TCN * 2; ++a) {
        const int co = c0 + wc * TCN * 32 + (a >> 1) * 32 + 16 * (g16 & 1) + 8 * (a & 1) + 4 * (g16 >> 1);
        if (slab != nullptr) {  // raw fp32 partial sums; bias / activation in the finalize pass
          if (co < p.ws_cs)
            *reinterpret_cast<float4*>(slab + opix * p.ws_cs + co) =
                make_float4(acc16[a][b][0], acc16[a][b][1], acc16[a][b][2], acc16[a][b][3]);
          continue;
        }
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float x = acc16[a][b][j] * p.out_scale + ((p.bias != nullptr && co + j < p.Cout) ? p.bias[co + j] : 0.f);
          if constexpr (sizeof(OutT) == 4) {
            if (p.accum && co + j < p.Cout) x += load_elem<OutT>(po + co + j);
            if (co + j < p.Cout) x = act_grad<OutT>(p, out16, po + co + j, co + j, x);
          }
          if (p.act == FN2_ACT_LEAKY) x = leaky(x);
          v[j] = x;
        }
        if (p.vec_ok && co + 3 < p.Cout) store_vec<OutT, 4>(po + co, v);
        else
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (co + j < p.Cout) store_elem<OutT>(po + co + j, v[j]);
      }
    }
    return;
  }

  // ---- epilogue.  Lane (pixel fr of tile tp, half fh) holds couts [16 fh, 16 fh + 16) of cout tile tc.
  OutT* out = reinterpret_cast<OutT*>(p.out);
  // wave-private LDS tile [pixel][cout] of the epilogues below (the stage buffers are free by now)
  constexpr int WCOLS = TCN * 32, RS = WCOLS + 4, WROWS = 32 * TPN;   // row stride padded: 16 lanes x float4 hit 16 bank groups
  constexpr int WTW = WROWS * RS + WROWS;                             // 4-byte words per wave: tile + one offset per row
  constexpr bool kLdsT = (KG > 1 ? 4 : 2) * WTW * 4 <= KG * ROWS * 128;
  if constexpr (FEED && !kLdsT) { if (feeder) return; }   // (no barrier below in that case)
  // K groups: the reduction scratch lives in lds0_all and other waves may still be reading it -> all four in lds1_all
  float* const tl = reinterpret_cast<float*>(KG > 1 ? ep1 : (wave < 2 ? lds0_all : ep1)) + (KG > 1 ? wave : (wave & 1)) * WTW;
  constexpr int CPR = WCOLS / 4, RPI = 64 / CPR;   // 16-byte chunks per tile row, rows per wave instruction
  static_assert(!SLAB || WREG, "SLAB is a specialisation of the WREG instantiations");
  if ((SLAB || !WREG) && p.splitk > 1) {
    float* slab = p.ws + (size_t)split * p.N * p.out_H * p.out_W * p.ws_cs;
    // The accumulators of a lane are 16 consecutive couts of ONE pixel: stored directly, a wave instruction writes 64
    // separate 16-byte pieces (one per pixel row of the slab).  Through a wave-private LDS tile [pixel][cout] the same
    // instruction count writes whole 256-byte runs (16 lanes per pixel row): measured 8-11 % of the kernel on the
    // split layers (tools/ab_conv.py, FN2_CONV_DBG bit 2097152 of the ablation build).
    if constexpr (kLdsT) {
      if constexpr (STAGES >= 3) __syncthreads();  // the ring ends without a barrier: other waves may still read their last slot
      if constexpr (FEED) { if (feeder) return; }
      int* rowoff = reinterpret_cast<int*>(tl + WROWS * RS);
#pragma unroll
      for (int tp = 0; tp < TPN; ++tp) {
        const int m = m0 + wp * TPN * 32 + tp * 32 + fr;
        const int mm = m < p.M ? m : 0;
        const int n = mm / (p.OH * p.OW);
        const int rem = mm - n * (p.OH * p.OW);
        const int oy = rem / p.OW, ox = rem - oy * p.OW;
        // slab offsets fit 31 bits: the slabs of a launch are < 2 GiB (split_bytes, conv.hip)
        if (fh == 0) rowoff[tp * 32 + fr] = m < p.M ? (int)((((size_t)n * p.out_H + (oy * osc + oy_off)) * p.out_W + (ox * osc + ox_off)) * p.ws_cs) : -1;
#pragma unroll
        for (int tc = 0; tc < TCN; ++tc)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(tl + (tp * 32 + fr) * RS + tc * 32 + fh * 16 + q * 4) =
                make_float4(acc[tc][tp][4 * q], acc[tc][tp][4 * q + 1], acc[tc][tp][4 * q + 2], acc[tc][tp][4 * q + 3]);
      }
      const int rr = lane / CPR, ch = lane - rr * CPR;
      const int co = c0 + wc * WCOLS + ch * 4;
#pragma unroll
      for (int i = 0; i < WROWS / RPI; ++i) {
        const int row = i * RPI + rr;
        const float4 v = *reinterpret_cast<const float4*>(tl + row * RS + ch * 4);
        const int off = rowoff[row];
        if (off >= 0 && co < p.ws_cs) *reinterpret_cast<float4*>(slab + (size_t)off + co) = v;
      }
      return;
    }
#pragma unroll
    for (int tp = 0; tp < TPN; ++tp) {
      const int m = m0 + wp * TPN * 32 + tp * 32 + fr;
      if (m >= p.M) continue;
      const int n = m / (p.OH * p.OW);
      const int rem = m - n * (p.OH * p.OW);
      const int oy = rem / p.OW, ox = rem - oy * p.OW;
      float* po = slab + (((size_t)n * p.out_H + (oy * osc + oy_off)) * p.out_W + (ox * osc + ox_off)) * p.ws_cs;
#pragma unroll
      for (int tc = 0; tc < TCN; ++tc)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int co = c0 + wc * TCN * 32 + tc * 32 + fh * 16 + q * 4;
          if (co < p.ws_cs)
            *reinterpret_cast<float4*>(po + co) = make_float4(acc[tc][tp][4 * q], acc[tc][tp][4 * q + 1],
                                                              acc[tc][tp][4 * q + 2], acc[tc][tp][4 * q + 3]);
        }
    }
    return;
  }
  if constexpr (SLAB) return;
  const bool vec16 = (p.out_cs % 8 == 0) && (p.out_c0 % 8 == 0);
  // 4-byte outputs whose whole wave tile lies inside the view: the finished 16-byte chunks go through the wave's LDS
  // tile and leave as whole (TCN x 128)-byte runs of a pixel, like the split-K slabs above (wave-uniform choice)
  const bool wide = kLdsT && sizeof(OutT) == 4 && vec16 && c0 + wc * WCOLS + WCOLS <= p.Cout && !(p.dbg & 4194304);
  int* const rowoff = reinterpret_cast<int*>(tl + WROWS * RS);
  if constexpr (STAGES >= 3 && kLdsT) __syncthreads();  // (every wave: `wide` may differ between the cout halves of a block)
  if constexpr (FEED) { if (feeder) return; }             // (the feeding waves took part in that barrier and have nothing to store)
#pragma unroll
  for (int tc = 0; tc < TCN; ++tc) {
    const int cout_base = c0 + wc * TCN * 32 + tc * 32 + fh * 16;
    float bias[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) bias[q] = (p.bias != nullptr && cout_base + q < p.Cout) ? p.bias[cout_base + q] : 0.f;
#pragma unroll
    for (int tp = 0; tp < TPN; ++tp) {
      const int m = m0 + wp * TPN * 32 + tp * 32 + fr;
      if (wide && tc == 0 && fh == 0) rowoff[tp * 32 + fr] = m < p.M ? 0 : -1;
      if (m >= p.M) continue;
      const int n = m / (p.OH * p.OW);
      const int rem = m - n * (p.OH * p.OW);
      const int oy = rem / p.OW, ox = rem - oy * p.OW;
      const size_t opix = (((size_t)n * p.out_H + (oy * osc + oy_off)) * p.out_W + (ox * osc + ox_off));
      OutT* po = out + opix * p.out_cs + p.out_c0 + cout_base;
      float v[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) v[q] = acc[tc][tp][q] * p.out_scale + bias[q];
      // fp32 and split-fp16 outputs can accumulate (gradient buffers) and take the fused LeakyReLU backward
      accum_act_grad16<OutT>(p, out, po, cout_base, vec16 && cout_base + 15 < p.Cout, v);
      if (p.act == FN2_ACT_LEAKY) {
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = leaky(v[q]);
      }
      if (wide) {
        if (tc == 0 && fh == 0) rowoff[tp * 32 + fr] = (int)opix;  // pixel index (< 2^31: conv.hip requires M < 2^31 outputs... per view)
        store16<OutT>(reinterpret_cast<OutT*>(tl + (tp * 32 + fr) * RS + tc * 32 + fh * 16), v);
      } else if (vec16 && cout_base + 15 < p.Cout) {
        store16<OutT>(po, v);
      } else {
#pragma unroll
        for (int q = 0; q < 16; ++q)
          if (cout_base + q < p.Cout) store_elem<OutT>(po + q, v[q]);
      }
    }
  }
  if (wide) {
    const int rr = lane / CPR, ch = lane - rr * CPR;
    OutT* ob = out + p.out_c0 + c0 + wc * WCOLS + ch * 4;
#pragma unroll
    for (int i = 0; i < WROWS / RPI; ++i) {
      const int row = i * RPI + rr;
      const uint4 v = *reinterpret_cast<const uint4*>(tl + row * RS + ch * 4);
      const int opix = rowoff[row];
      if (opix >= 0) *reinterpret_cast<uint4*>(ob + (size_t)opix * p.out_cs) = v;
    }
  }
#endif  // __HIP_DEVICE_COMPILE__
}

// ---------------------------------------------------------------------------------------------------------------
// Halo variant for stride-1 layers (3x3 convolutions and the 2x2 phase convolutions of the transposed convs) on
// split-fp16 operands.  In the kernel above every filter tap fetches its own copy of the tile's pixel rows, although
// the taps kx = 0..KW-1 of one kernel row read the SAME input pixels shifted by one: a tile of BP consecutive pixels
// of an image row needs BP + KW - 1 input pixels per (channel block, ky), not KW * BP.  Here the pixel operand is
// fetched once per (channel block, ky) WITH its halo -- LDS row e <-> input pixel ix0 + e, so a row is valid or zero
// by the pixel it is, whatever tap reads it -- and tap kx reads the fragment rows r + kx; the weights still move per
// tap.  L2 -> LDS bytes per (channel block, ky): (KW*BC + BP + KW - 1) rows instead of KW * (BC + BP): -22 % for the
// 128 x 64 tile, -44 % for 64 x 128, -53 % for 32 x 128 (KW = 3) -- the layers with few output channels (the
// full-resolution fusion layers, deconv2, interconvN) are bound by exactly that stream.
// Requires: stride 1, no split-K, tiles that do not straddle image rows (OW % BP == 0).
template <typename OutT, int WC, int WP, int TCN, int TPN, int KW>
__global__ void __launch_bounds__(256) conv_halo_kernel(const ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  using T = x2_t;
  constexpr int ESZ = 4;
  constexpr int BC = WC * TCN * 32, BP = WP * TPN * 32;
  static_assert(WC * WP == 4, "4 waves per block");
  constexpr int NWI = BC / 32;                // weight pieces (8 rows each) per wave per tap stage
  constexpr int BROWS = BP + 8;               // BP + KW - 1 pixel rows, whole pieces
  constexpr int NBP = BROWS / 8;              // pixel pieces per (channel block, ky)
  constexpr int NBW = (NBP + 3) / 4;          // ... per wave
  __shared__ uint4 ldsA[2][BC * 8];
  __shared__ uint4 ldsB[2][BROWS * 8];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave / WP, wp = wave % WP;

  int pad_y = p.pad, pad_x = p.pad, oy_off = 0, ox_off = 0, osc = 1;
  const T* wgt = reinterpret_cast<const T*>(p.wgt);
  const int phase = blockIdx.z;
  const unsigned wrow_bytes = (unsigned)p.ksteps * 128u;
  if (p.merged) {  // kind 5: the two column phases of output row 2y + phase in one block (columns x - 1 .. x + 1 of the input)
    pad_y = phase ? p.ph_pad1 : p.ph_pad0; pad_x = 1; oy_off = phase; ox_off = 0; osc = 2;
    wgt += (size_t)phase * p.cout_pad * (wrow_bytes / ESZ);
  } else if (p.deconv) {
    const int a = phase >> 1, b = phase & 1;
    pad_y = a ? p.ph_pad1 : p.ph_pad0; pad_x = b ? p.ph_pad1 : p.ph_pad0; oy_off = a; ox_off = b; osc = 2;
    wgt += (size_t)phase * p.cout_pad * (wrow_bytes / ESZ);
  }
  int bx = blockIdx.x, by = blockIdx.y;
  {  // XCD-aware tile order (see conv_igemm2_kernel)
    const int NT = gridDim.x * gridDim.y, L = blockIdx.x + gridDim.x * blockIdx.y;
    const int xcd = L & 7, chunk = NT >> 3, rem = NT & 7;
    const int Lp = xcd * chunk + min(xcd, rem) + (L >> 3);
    bx = Lp / (int)gridDim.y;
    by = Lp - bx * (int)gridDim.y;
  }
  const int m0 = bx * BP, c0 = by * BC;
  // the tile lies inside one image row: (n, oy) are block-uniform
  const int tn = m0 / (p.OH * p.OW);
  const int trem = m0 - tn * (p.OH * p.OW);
  const int toy = trem / p.OW, tox = trem - toy * p.OW;
  const int iy_base = toy - pad_y, ix_base = tox - pad_x;

  const v4i_t rsrc_w = make_rsrc(wgt, (int)(p.cout_pad * wrow_bytes));
  const v4i_t rsrc_x = make_rsrc(p.in, p.in_bytes);
  const int lrow = lane >> 3, lphys = lane & 7;
  unsigned woff[NWI];
#pragma unroll
  for (int j = 0; j < NWI; ++j) {
    const int row = wave * (BC / 4) + j * 8 + lrow;
    woff[j] = (unsigned)(c0 + row) * wrow_bytes + (unsigned)((lphys ^ ((row >> 1) & 7)) * 16);
  }
  // pixel rows with halo: LDS row e <-> input pixel (tn, iy_base + ky, ix_base + e); x validity is the lane's own
  int boff[NBW];
  bool bok[NBW];
#pragma unroll
  for (int j = 0; j < NBW; ++j) {
    const int q = wave + 4 * j;
    const int e = q * 8 + lrow;
    const int ix = ix_base + e;
    bok[j] = q < NBP && e < BP + KW - 1 && ix >= 0 && ix < p.W;
    boff[j] = (((tn * p.H + iy_base) * p.W + ix) * p.in_cs + p.in_c0) * ESZ + (lphys ^ ((e >> 1) & 7)) * 16;
  }
  const int spt = p.cin_chunks >> 3;       // 128-byte channel blocks per tap
  const int nsup = spt * p.KH;             // (channel block, ky) super-stages; K order: channel block, ky, kx
  auto issue_B = [&](int sup, uint4* lds) {
    const int sc = sup / p.KH, ky = sup - sc * p.KH;
    const bool yok = (unsigned)(iy_base + ky) < (unsigned)p.H;
    const int toff = (ky * p.W * p.in_cs) * ESZ + sc * 128;
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
      const int q = wave + 4 * j;
      if (q < NBP) dma16(rsrc_x, &lds[q * 64], (bok[j] && yok) ? (unsigned)(boff[j] + toff) : kOobOffset, 0);
    }
  };
  auto issue_A = [&](int sup, int kx, uint4* lds) {
    const int sc = sup / p.KH, ky = sup - sc * p.KH;
    const int soff = ((ky * KW + kx) * spt + sc) * 128;
#pragma unroll
    for (int j = 0; j < NWI; ++j) dma16(rsrc_w, &lds[(wave * (BC / 4) + j * 8) * 8], woff[j], soff);
  };

  const int fr = lane & 31, fh = lane >> 5, fsw = (fr >> 1) & 7;
  f32x16 acc[TCN][TPN];
#pragma unroll
  for (int tc = 0; tc < TCN; ++tc)
#pragma unroll
    for (int tp = 0; tp < TPN; ++tp)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[tc][tp][q] = 0.f;

  auto compute = [&](const uint4* la, const uint4* lb, int kx) {
#if FN2_SETPRIO
    __builtin_amdgcn_s_setprio(1);
#endif
    const uint4* A = &la[(wc * TCN * 32 + fr) * 8];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int chh = (4 * q + 2 * fh) ^ fsw, chl = (4 * q + 2 * fh + 1) ^ fsw;
      uint4 ah[TCN], al[TCN], bh[TPN], bl[TPN];
#pragma unroll
      for (int t = 0; t < TCN; ++t) { ah[t] = A[t * 32 * 8 + chh]; al[t] = A[t * 32 * 8 + chl]; }
#pragma unroll
      for (int t = 0; t < TPN; ++t) {
        const int rb = wp * TPN * 32 + t * 32 + fr + kx, sw = (rb >> 1) & 7;  // tap kx = the same pixels one row on
        bh[t] = lb[rb * 8 + ((4 * q + 2 * fh) ^ sw)];
        bl[t] = lb[rb * 8 + ((4 * q + 2 * fh + 1) ^ sw)];
      }
#pragma unroll
      for (int tc = 0; tc < TCN; ++tc)
#pragma unroll
        for (int tp = 0; tp < TPN; ++tp) {
          acc[tc][tp] = mfma_32x32x16<f16_t>(al[tc], bh[tp], acc[tc][tp]);
          acc[tc][tp] = mfma_32x32x16<f16_t>(ah[tc], bl[tp], acc[tc][tp]);
          acc[tc][tp] = mfma_32x32x16<f16_t>(ah[tc], bh[tp], acc[tc][tp]);
        }
    }
#if FN2_SETPRIO
    __builtin_amdgcn_s_setprio(0);
#endif
  };

  issue_B(0, ldsB[0]);
  issue_A(0, 0, ldsA[0]);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int abuf = 0;
  for (int sup = 0; sup < nsup; ++sup) {
#pragma unroll
    for (int kx = 0; kx < KW; ++kx) {
      // the next tap's weights, and with the first tap of a super-stage the NEXT super-stage's pixel rows (two more
      // tap stages pass before they are read), in flight under this stage's MFMAs
      if (kx + 1 < KW) issue_A(sup, kx + 1, ldsA[abuf ^ 1]);
      else if (sup + 1 < nsup) issue_A(sup + 1, 0, ldsA[abuf ^ 1]);
      if (kx == 0 && sup + 1 < nsup) issue_B(sup + 1, ldsB[(sup + 1) & 1]);
      compute(ldsA[abuf], ldsB[sup & 1], kx);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      abuf ^= 1;
    }
  }

  // ---- epilogue (as conv_igemm2_kernel): lane (pixel fr of tile tp, half fh) holds couts [16 fh, 16 fh + 16) of tile tc
  OutT* out = reinterpret_cast<OutT*>(p.out);
  const bool vec16 = (p.out_cs % 8 == 0) && (p.out_c0 % 8 == 0);
  constexpr int ERS = 36;   // floats per epilogue tile row: 32 couts + 16 bytes of padding
  static_assert(sizeof(ldsA) >= 4 * 32 * ERS * 4 || sizeof(ldsB) >= 4 * 32 * ERS * 4, "epilogue tiles fit a stage array");
  float* const etl = reinterpret_cast<float*>(sizeof(ldsA) >= 4 * 32 * ERS * 4 ? &ldsA[0][0] : &ldsB[0][0]) + wave * (32 * ERS);
#pragma unroll
  for (int tc = 0; tc < TCN; ++tc) {
    int cout_base = c0 + wc * TCN * 32 + tc * 32 + fh * 16;
    int bsel = 0;   // merged (kind 5): packed row b Cout + co -> column phase b = output pixel 2 x + b, channel co
    if (p.merged) { bsel = cout_base / p.Cout; cout_base -= bsel * p.Cout; if (bsel > 1) continue; }
    float bias[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) bias[q] = (p.bias != nullptr && cout_base + q < p.Cout) ? p.bias[cout_base + q] : 0.f;
#pragma unroll
    for (int tp = 0; tp < TPN; ++tp) {
      const int ox = tox + wp * TPN * 32 + tp * 32 + fr;
      OutT* po = out + (((size_t)tn * p.out_H + (toy * osc + oy_off)) * p.out_W + (ox * osc + ox_off + bsel)) * p.out_cs +
                 p.out_c0 + cout_base;
      float v[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) v[q] = acc[tc][tp][q] * p.out_scale + bias[q];
      accum_act_grad16<OutT>(p, out, po, cout_base, vec16 && cout_base + 15 < p.Cout, v);
      if (p.act == FN2_ACT_LEAKY) {
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = leaky(v[q]);
      }
      if (sizeof(OutT) == 4 && vec16 && !p.merged && c0 + wc * TCN * 32 + tc * 32 + 32 <= p.Cout && !(p.dbg & 4194304)) {
        // 128-byte runs through a wave-private LDS tile (the loop's last barrier has passed), as in conv_rowrun_kernel
        store16<OutT>(reinterpret_cast<OutT*>(etl + fr * ERS + fh * 16), v);
        const int rr = lane >> 3, ch = lane & 7;
        const int ox0 = tox + wp * TPN * 32 + tp * 32;
        OutT* ob = out + (((size_t)tn * p.out_H + (toy * osc + oy_off)) * p.out_W + (ox0 * osc + ox_off)) * p.out_cs +
                   p.out_c0 + c0 + wc * TCN * 32 + tc * 32 + ch * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = i * 8 + rr;
          *reinterpret_cast<uint4*>(ob + (size_t)row * osc * p.out_cs) = *reinterpret_cast<const uint4*>(etl + row * ERS + ch * 4);
        }
      } else if (vec16 && cout_base + 15 < p.Cout) {
        store16<OutT>(po, v);
      } else {
#pragma unroll
        for (int q = 0; q < 16; ++q)
          if (cout_base + q < p.Cout) store_elem<OutT>(po + q, v[q]);
      }
      if (p.merged && p.up_src != nullptr && cout_base == 0 && ox < p.OW) {
        // upsample_flowXtoY of this output pixel (upsample_flow_kernel's arithmetic, tap for tap; conv.hip)
        const int uy = toy * osc + oy_off, ux = ox * osc + ox_off + bsel;
        const int H2 = p.out_H >> 1, W2 = p.out_W >> 1;
        const int a = uy & 1, b = ux & 1, y = uy >> 1, x = ux >> 1;
        float r0 = p.up_bias ? p.up_bias[0] : 0.f, r1 = p.up_bias ? p.up_bias[1] : 0.f;
#pragma unroll
        for (int ty = 0; ty < 2; ++ty) {
          const int iy = y - 1 + a + ty, ky = 3 - a - 2 * ty;
          if (iy < 0 || iy >= H2) continue;
#pragma unroll
          for (int tx = 0; tx < 2; ++tx) {
            const int ix = x - 1 + b + tx, kx = 3 - b - 2 * tx;
            if (ix < 0 || ix >= W2) continue;
            const float2 sv = *reinterpret_cast<const float2*>(p.up_src + (((long)tn * H2 + iy) * W2 + ix) * 2);
            const float* ww = p.up_w + (ky * 4 + kx) * 4;
            r0 += sv.x * ww[0] + sv.y * ww[1];
            r1 += sv.x * ww[2] + sv.y * ww[3];
          }
        }
        OutT* pu = out + (((size_t)tn * p.out_H + uy) * p.out_W + ux) * p.out_cs + p.up_c0;
        store_elem<OutT>(pu, r0);
        store_elem<OutT>(pu + 1, r1);
      }
    }
  }
#endif  // __HIP_DEVICE_COMPILE__
}

// ---------------------------------------------------------------------------------------------------------------
// Persistent short-K kernel for the 3x3 stems at full resolution (FlowNetSD conv0, the fusion net's fuse_conv0: 6 / 11
// input channels -> 64, flownet_sd.py:29, flownet2.py:61): as kind-2 row-run convolutions their whole K is NST = 3
// stages, so in the generic kernel a block spends its life in the prologue (per-lane pixel arithmetic), three
// dependent DMA round trips and the epilogue -- 6144 blocks of that at batch 4.  Here a block keeps the 64 x K weight
// matrix in LDS for its whole life (fetched once, not once per pixel tile) and walks pixel tiles blockIdx.x,
// blockIdx.x + gridDim.x, ...: ALL the stages of the next tile are in flight under the current tile's MFMAs and
// stores.  One block of four waves per CU (120 KB of LDS), 64 cout x 128 pixel tiles as conv_igemm2<.., 1, 4, 2, 1>.
template <typename OutT, int NST>
__global__ void __launch_bounds__(256) conv_stem_kernel(const ConvArgs p, const int ntiles) {
#if defined(__HIP_DEVICE_COMPILE__)
  using T = x2_t;
  constexpr int ESZ = 4;
  constexpr int BC = 64, BP = 128, TCN = 2;
  __shared__ uint4 ldsA[NST][BC * 8];
  __shared__ uint4 ldsB[2][NST][BP * 8];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // = pixel quarter wp (WC = 1, WP = 4)
  const unsigned wrow_bytes = (unsigned)p.ksteps * 128u;
  const v4i_t rsrc_w = make_rsrc(p.wgt, (int)(p.cout_pad * wrow_bytes));
  const v4i_t rsrc_x = make_rsrc(p.in, p.in_bytes);
  const int lrow = lane >> 3, lphys = lane & 7;
  const int spt = p.cin_chunks >> 3;  // lines per run
  // stage s: channel line sc = s / KH of kernel row ky = s % KH (the K order of conv_igemm2_kernel: line outer, row inner)
  // ---- weights: once
#pragma unroll
  for (int st = 0; st < NST; ++st) {
    const int sc = st / p.KH, ky = st - sc * p.KH;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = wave * 16 + j * 8 + lrow;
      const unsigned woff = (unsigned)row * wrow_bytes + (unsigned)((lphys ^ ((row >> 1) & 7)) * 16);
      dma16(rsrc_w, &ldsA[st][(wave * 16 + j * 8) * 8], woff, (ky * spt + sc) * 128);
    }
  }
  auto issue_B = [&](int tile, int buf) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = wave * 32 + j * 8 + lrow;
      const int m = tile * BP + row;
      const bool v = m < p.M && tile < ntiles;
      const int mm = v ? m : 0;
      const int n = mm / (p.OH * p.OW);
      const int rem = mm - n * (p.OH * p.OW);
      const int oy = rem / p.OW, ox = rem - oy * p.OW;
      const int roff = (((n * p.H + oy * p.stride) * p.W + ox * p.stride) * p.in_cs + p.in_c0) * ESZ +
                       (lphys ^ ((row >> 1) & 7)) * 16;
#pragma unroll
      for (int st = 0; st < NST; ++st) {
        const int sc = st / p.KH, ky = st - sc * p.KH;
        const int toff = (ky * p.W * p.in_cs) * ESZ + sc * 128;
        dma16(rsrc_x, &ldsB[buf][st][(wave * 32 + j * 8) * 8], v ? (unsigned)(roff + toff) : kOobOffset, 0);
      }
    }
  };
  const int fr = lane & 31, fh = lane >> 5, fsw = (fr >> 1) & 7;
  OutT* out = reinterpret_cast<OutT*>(p.out);
  const bool vec16 = (p.out_cs % 8 == 0) && (p.out_c0 % 8 == 0);
  float bias[TCN][16];
#pragma unroll
  for (int tc = 0; tc < TCN; ++tc)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int co = tc * 32 + fh * 16 + q;
      bias[tc][q] = (p.bias != nullptr && co < p.Cout) ? p.bias[co] : 0.f;
    }

  int tile = blockIdx.x, cur = 0;
  issue_B(tile, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (; tile < ntiles; tile += gridDim.x, cur ^= 1) {
    issue_B(tile + gridDim.x, cur ^ 1);  // every stage of the next tile (zeros past the last one)
    f32x16 acc[TCN];
#pragma unroll
    for (int tc = 0; tc < TCN; ++tc)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[tc][q] = 0.f;
#if FN2_SETPRIO
    __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
    for (int st = 0; st < NST; ++st) {
      const uint4* A = &ldsA[st][fr * 8];
      const uint4* B = &ldsB[cur][st][(wave * 32 + fr) * 8];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int chh = (4 * q + 2 * fh) ^ fsw, chl = (4 * q + 2 * fh + 1) ^ fsw;
        const uint4 bh = B[chh], bl = B[chl];
#pragma unroll
        for (int tc = 0; tc < TCN; ++tc) {
          const uint4 ah = A[tc * 32 * 8 + chh], al = A[tc * 32 * 8 + chl];
          acc[tc] = mfma_32x32x16<f16_t>(al, bh, acc[tc]);
          acc[tc] = mfma_32x32x16<f16_t>(ah, bl, acc[tc]);
          acc[tc] = mfma_32x32x16<f16_t>(ah, bh, acc[tc]);
        }
      }
    }
#if FN2_SETPRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    const int m = tile * BP + wave * 32 + fr;
    if (m < p.M) {
      const int n = m / (p.OH * p.OW);
      const int rem = m - n * (p.OH * p.OW);
      const int oy = rem / p.OW, ox = rem - oy * p.OW;
#pragma unroll
      for (int tc = 0; tc < TCN; ++tc) {
        const int cout_base = tc * 32 + fh * 16;
        OutT* po = out + (((size_t)n * p.out_H + oy) * p.out_W + ox) * p.out_cs + p.out_c0 + cout_base;
        float v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          float x = acc[tc][q] * p.out_scale + bias[tc][q];
          if (p.act == FN2_ACT_LEAKY) x = leaky(x);
          v[q] = x;
        }
        if (vec16 && cout_base + 15 < p.Cout) {
          store16<OutT>(po, v);
        } else {
#pragma unroll
          for (int q = 0; q < 16; ++q)
            if (cout_base + q < p.Cout) store_elem<OutT>(po + q, v[q]);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
#endif  // __HIP_DEVICE_COMPILE__
}

// ---------------------------------------------------------------------------------------------------------------
// Composed 5x5 flow head, STRIP form (fn2_flow_head5; flownet2.py:74-77 / :90-93 fuse_interconvN + predict_flowN, the
// two full- / half-resolution heads of the fusion net).  The tile form above (HEAD5) computes the 50 (tap, output)
// partials on an 8 x 32 window to get 4 x 28 outputs: 2.3x the input bytes and MFMAs of the layer.  Here a block walks
// DOWN a strip of 64 input columns (60 outputs): per input row r it forms the partial row T[50][64] (64 pixels x 64
// packed couts x K on the matrix cores, weights resident in LDS as in conv_stem_kernel), and adds, for every tap row ky,
// sum_kx T[(ky, kx, o)][x + kx - 2] to the accumulator of output row r + 2 - ky (a 5-row ring in LDS); output row r - 2
// is complete when row r has been added and leaves as one 480-byte run.  Input bytes per output: 64/60 x (R + 4)/R.
// Eight waves: 0-3 multiply (wave = 32 couts x 32 pixels) and write T; 4-7 only feed the 4-slot LDS-DMA ring of
// 96-channel chunks (CPR chunks per row: 1 for the 82-channel concat0, 2 for the 162-channel concat1) -- the ~150
// cycles each DMA piece costs its wave to issue (24 pieces per chunk) run beside the MFMAs on the SIMD's other wave
// slot instead of in front of them; counted vmcnt + raw barriers.  All eight gather.  Measured on the first form (four
// waves doing everything, 3 slots; tools/diag/h5_ablate.py, 4x384x512x82): DMA + barriers 51 us, + MFMA 67, + gather 99
// (three dependent LDS round trips per item behind branches: batched here), + ring share 126.
// The border ring (other weights per border case, h5_ring_pixel) is shared out evenly: every block ends with its share.
template <int CPR>
__global__ void __launch_bounds__(512) head5_strip_kernel(const ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int RD = 4, CL = 3, NL = CPR * CL;   // ring slots (five for the 96-channel form measured the same)
  constexpr int TS = 68;   // floats per row of T (64 columns; + 4 moves consecutive rows onto other banks)
  __shared__ uint4 ldsW[NL][64 * 8];
  __shared__ uint4 ldsX[RD][CL][64 * 8];
  __shared__ float T[50 * TS];
  __shared__ float accr[5 * 64 * 2];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool feeder = wave >= 4;
  const int w4 = wave & 3;
  const int wc = w4 >> 1, wp = w4 & 1;
  const int H = p.OH, W = p.OW;
  const int nstrip = p.h5_tx, nseg = p.h5_ty, R = p.h5_rows;
  const int b = blockIdx.x;
  const int sx = b % nstrip, sg = (b / nstrip) % nseg, n = b / (nstrip * nseg);
  const int x0 = sx * 60 - 2;
  const int y0 = sg * R, y1 = min(H, y0 + R);
  const int r0 = max(0, y0 - 2), r1 = min(H - 1, y1 + 1);
  const int nch = (r1 - r0 + 1) * CPR;
  const float bias_o[2] = {p.bias ? p.bias[0] : 0.f, p.bias ? p.bias[1] : 0.f};
  const unsigned wrow_bytes = (unsigned)p.ksteps * 128u;
  const v4i_t rsrc_w = make_rsrc(p.wgt, (int)(p.cout_pad * wrow_bytes));
  const v4i_t rsrc_x = make_rsrc(p.in, p.in_bytes);
  const int lrow = lane >> 3, lphys = lane & 7;
  for (int i = tid; i < 5 * 64 * 2; i += 512) accr[i] = 0.f;
  // chunk c = (input row r0 + c / CPR, channel lines [CL (c % CPR), +CL)); 6 DMA pieces per feeder wave, zeros past the end
  auto issue = [&](int c, int slot) {
    const int rr = c / CPR, kc = c - rr * CPR;
    const int r = r0 + rr;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int px = w4 * 16 + j * 8 + lrow;
      const int x = x0 + px;
      const bool ok = c < nch && x >= 0 && x < W;
      const unsigned off = ok ? (unsigned)((((n * H + r) * W + x) * p.in_cs + p.in_c0) * 4 + (lphys ^ ((px >> 1) & 7)) * 16 +
                                           kc * (CL * 128))
                              : kOobOffset;
#pragma unroll
      for (int ln = 0; ln < CL; ++ln)
        dma16(rsrc_x, &ldsX[slot][ln][(w4 * 16 + j * 8) * 8], ok ? off + ln * 128 : kOobOffset, 0);
    }
  };
  if (feeder) {
    // ---- weights: once
#pragma unroll
    for (int ln = 0; ln < NL; ++ln)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = w4 * 16 + j * 8 + lrow;
        const unsigned woff = (unsigned)row * wrow_bytes + (unsigned)((lphys ^ ((row >> 1) & 7)) * 16);
        dma16(rsrc_w, &ldsW[ln][(w4 * 16 + j * 8) * 8], woff, ln * 128);
      }
#pragma unroll
    for (int c = 0; c < RD - 1; ++c) issue(c, c);
  }
  const int fr = lane & 31, fh = lane >> 5, fsw = (fr >> 1) & 7;
  float* pf = reinterpret_cast<float*>(p.out);
  f32x16 acc;
  int slot = 0;
  for (int c = 0; c < nch; ++c) {
    const int rr = c / CPR, kc = c - rr * CPR;
    const int r = r0 + rr;
    if (feeder) {   // chunks c + 1 .. c + RD - 2 stay in flight (6 pieces each)
      if constexpr (RD == 5) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // (the gather's accumulator writes)
    __builtin_amdgcn_s_barrier();
    const bool last = kc == CPR - 1;
    if (feeder) {
      issue(c + RD - 1, slot == 0 ? RD - 1 : slot - 1);   // the slot of chunk c - 1: every wave passed its MFMAs a barrier ago
    } else {
      if (kc == 0) {
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.f;
      }
      if (!(p.dbg & 64))   // (ablation bits 64 / 32 / 16: no MFMA / no gather / no ring share; timing only)
#pragma unroll
      for (int ln = 0; ln < CL; ++ln) {
        const uint4* A = &ldsW[kc * CL + ln][(wc * 32 + fr) * 8];
        const uint4* B = &ldsX[slot][ln][(wp * 32 + fr) * 8];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int chh = (4 * q + 2 * fh) ^ fsw, chl = (4 * q + 2 * fh + 1) ^ fsw;
          const uint4 bh = B[chh], bl = B[chl];
          const uint4 ah = A[chh], al = A[chl];
          acc = mfma_32x32x16<f16_t>(al, bh, acc);
          acc = mfma_32x32x16<f16_t>(ah, bl, acc);
          acc = mfma_32x32x16<f16_t>(ah, bh, acc);
        }
      }
      if (last) {   // T[(ky * 5 + kx) * 2 + o][column]: lane (column wp 32 + fr, half fh) holds packed couts [wc 32 + fh 16, +16)
        const int cb = wc * 32 + fh * 16;
#pragma unroll
        for (int q = 0; q < 16; ++q)
          if (cb + q < 50) T[(cb + q) * TS + wp * 32 + fr] = acc[q] * p.out_scale;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    slot = slot == RD - 1 ? 0 : slot + 1;
    if (last && !(p.dbg & 32)) {
      // all eight waves add: wave g the (ky, o) items g and g + 8 (waves 0, 1) for column x = lane (measured: the four
      // multiplying waves alone, three items each, 89 against 81 us).  Every LDS word is fetched before the first use, no
      // branch in front of the loads (clamped addresses, results selected)
      const int x = lane;
      const int xc = min(max(x, 2), 61);
      float tv[2][5], av[2];
      int ai[2];
      bool ok[2], fin[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int it = min(wave + 8 * k, 9);
        const int ky = it >> 1, o = it & 1;
        const int y = r + 2 - ky;
        ok[k] = wave + 8 * k < 10 && y >= y0 && y < y1 && x >= 2 && x < 62;
        fin[k] = ky == 4 || r == H - 1;   // no later input row reaches output row y
        const int ys = ((y % 5) + 5) % 5;
        ai[k] = (ys * 64 + xc) * 2 + o;
        const float* t = &T[(ky * 10 + o) * TS + xc - 2];
#pragma unroll
        for (int kx = 0; kx < 5; ++kx) tv[k][kx] = t[2 * kx * TS + kx];
        av[k] = accr[ai[k]];
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int it = min(wave + 8 * k, 9);
        const int ky = it >> 1, o = it & 1;
        const int y = r + 2 - ky;
        float sum = tv[k][0];
        sum += tv[k][1];
        sum += tv[k][2];
        sum += tv[k][3];
        sum += tv[k][4];
        const float v = av[k] + sum;
        if (ok[k]) {
          accr[ai[k]] = fin[k] ? 0.f : v;
          const int ox = x0 + x;
          if (fin[k] && ox < W && !(p.h5_ring && (y == 0 || y == H - 1 || ox == 0 || ox == W - 1)))
            pf[(((size_t)n * H + y) * W + ox) * 2 + o] = v + bias_o[o];
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (p.h5_wc != nullptr && !(p.dbg & 16)) {   // this block's share of the border ring, a wave per pixel
    const long total = (long)p.N * (2 * W + 2 * (H - 2));
    const long per = (total + gridDim.x - 1) / gridDim.x;
    const long m1 = min(total, (long)(b + 1) * per);
    for (long m = (long)b * per + wave; m < m1; m += 8) h5_ring_pixel(p, m);
  }
#endif  // __HIP_DEVICE_COMPILE__
}


// ---------------------------------------------------------------------------------------------------------------
// Row-run stems from the RAW image row (kind 2: the first layer of every network on its pre-padded few-channel input;
// flownet_s.py:39 conv1 7x7 s2 on 12 / 6 channels, flownet_c.py:30-34 as 4x4 s1 on 2x2 super-pixels, flownet_sd.py:29
// conv0 and flownet2.py:61 fuse_conv0 3x3 s1).  In conv_igemm2_kernel a pixel's run of KW*cs channels is fetched as
// whole 128-byte lines per OUTPUT pixel although neighbouring runs overlap by (KW - stride) / KW: conv1 of FlowNetS
// moves 3.5 KB per output pixel and kernel row.  Here the block fetches the input row SEGMENT its 128 output pixels
// read -- (127 stride + KW) pixels x cs channels, once per kernel row -- and the MFMA operand of output pixel r,
// 8-channel group g is simply LDS slot (r stride) (cs / 4) + 2 g (+1 for the lo half): the run IS contiguous in the
// raw image.  Slots are XOR-swizzled (slot ^ ((slot >> 4) & 15), applied to the DMA source and to the reads) so that 16
// lanes at a stride of 2, 4 or 8 slots land on 16 different 16-byte slots of a bank row.  About half the L2 -> LDS
// bytes of the line form for every stem (weights still move per 128-byte line of the run).
// Tile 64 cout x 128 pixels of one output row; weights double-buffered per line, the segment per kernel row.
template <typename OutT, int SPP>
__global__ void __launch_bounds__(256) conv_rowrun_kernel(const ConvArgs p, const int npieces) {
#if defined(__HIP_DEVICE_COMPILE__)
  using T = x2_t;
  constexpr int BC = 64, BP = 128, TCN = 2;
  constexpr int MAXP = SPP == 4 ? 17 : 9;      // 1 KB pieces of a segment: 7x7 s2 on cs 16 reads 1048 slots, on cs 8 524
  __shared__ uint4 ldsA[2][BC * 8];
  __shared__ uint4 ldsB[2][MAXP * 64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const T* wgt = reinterpret_cast<const T*>(p.wgt);
  const unsigned wrow_bytes = (unsigned)p.ksteps * 128u;
  int bx = blockIdx.x, by = blockIdx.y;
  {  // XCD-aware tile order (see conv_igemm2_kernel): vertical neighbours share KH - stride input rows
    const int NT = gridDim.x * gridDim.y, L = blockIdx.x + gridDim.x * blockIdx.y;
    const int xcd = L & 7, chunk = NT >> 3, rem = NT & 7;
    const int Lp = xcd * chunk + min(xcd, rem) + (L >> 3);
    bx = Lp / (int)gridDim.y;
    by = Lp - bx * (int)gridDim.y;
  }
  const int m0 = bx * BP, c0 = by * BC;
  const int tn = m0 / (p.OH * p.OW);
  const int trem = m0 - tn * (p.OH * p.OW);
  const int toy = trem / p.OW, tox = trem - toy * p.OW;   // the tile lies inside one output row

  const v4i_t rsrc_w = make_rsrc(wgt, (int)(p.cout_pad * wrow_bytes));
  const v4i_t rsrc_x = make_rsrc(p.in, p.in_bytes);
  const int lrow = lane >> 3, lphys = lane & 7;
  unsigned woff[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = wave * 16 + j * 8 + lrow;
    woff[j] = (unsigned)(c0 + row) * wrow_bytes + (unsigned)((lphys ^ ((row >> 1) & 7)) * 16);
  }
  // segment DMA: physical slot P of piece q <- logical slot P ^ ((P >> 4) & 15) of the row segment
  const int nslots = ((BP - 1) * p.stride + p.KH_KW_hint) * SPP;  // slots holding real pixels
  const int seg0 = (((tn * p.H + toy * p.stride) * p.W + tox * p.stride) * p.in_cs + p.in_c0) * 4;  // byte offset at ky = 0
  constexpr int NBW = (MAXP + 3) / 4;
  unsigned boff[NBW];
#pragma unroll
  for (int j = 0; j < NBW; ++j) {
    const int q = wave + 4 * j;
    const int P = q * 64 + lane;
    const int Ls = P ^ ((P >> 4) & 15);
    boff[j] = (q < npieces && Ls < nslots) ? (unsigned)(seg0 + Ls * 16) : kOobOffset;
  }
  const int nl = p.cin_chunks >> 3;   // 128-byte lines per kernel row
  const int ngroups = (p.KH_KW_hint * p.in_cs + 7) >> 3;   // 8-channel groups of a run that hold pixels (7x7 on 16 channels: 14 of 16)
  auto issue_B = [&](int ky, uint4* lds) {
    const int toff = ky * p.W * p.in_cs * 4;
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
      const int q = wave + 4 * j;
      if (q < npieces) dma16(rsrc_x, &lds[q * 64], boff[j] == kOobOffset ? kOobOffset : boff[j] + (unsigned)toff, 0);
    }
  };
  auto issue_A = [&](int ky, int l, uint4* lds) {   // weight k index = ky * run_pad + channel (pack_stem)
    const int soff = (ky * nl + l) * 128;
#pragma unroll
    for (int j = 0; j < 2; ++j) dma16(rsrc_w, &lds[(wave * 16 + j * 8) * 8], woff[j], soff);
  };

  const int fr = lane & 31, fh = lane >> 5, fsw = (fr >> 1) & 7;
  const int pslot = (wave * 32 + fr) * p.stride * SPP;   // first slot of this lane's pixel run
  f32x16 acc[TCN];
#pragma unroll
  for (int tc = 0; tc < TCN; ++tc)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[tc][q] = 0.f;

  issue_B(0, ldsB[0]);
  issue_A(0, 0, ldsA[0]);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int abuf = 0;
  for (int ky = 0; ky < p.KH; ++ky) {
    const uint4* lb = ldsB[ky & 1];
    for (int l = 0; l < nl; ++l) {
      if (l + 1 < nl) issue_A(ky, l + 1, ldsA[abuf ^ 1]);
      else if (ky + 1 < p.KH) issue_A(ky + 1, 0, ldsA[abuf ^ 1]);
      if (l == 0 && ky + 1 < p.KH) issue_B(ky + 1, ldsB[(ky + 1) & 1]);
#if FN2_SETPRIO
      __builtin_amdgcn_s_setprio(1);
#endif
      const uint4* A = &ldsA[abuf][fr * 8];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        if (4 * l + 2 * q >= ngroups) continue;         // a k-step wholly in the zero padding of the run's last line
        const int chh = (4 * q + 2 * fh) ^ fsw, chl = (4 * q + 2 * fh + 1) ^ fsw;
        const int g = 4 * l + 2 * q + fh;               // 8-channel group of the run
        const int sh = pslot + 2 * g, sl = sh + 1;
        const uint4 bh = lb[sh ^ ((sh >> 4) & 15)], bl = lb[sl ^ ((sl >> 4) & 15)];
#pragma unroll
        for (int tc = 0; tc < TCN; ++tc) {
          const uint4 ah = A[tc * 32 * 8 + chh], al = A[tc * 32 * 8 + chl];
          acc[tc] = mfma_32x32x16<f16_t>(al, bh, acc[tc]);
          acc[tc] = mfma_32x32x16<f16_t>(ah, bl, acc[tc]);
          acc[tc] = mfma_32x32x16<f16_t>(ah, bh, acc[tc]);
        }
      }
#if FN2_SETPRIO
      __builtin_amdgcn_s_setprio(0);
#endif
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      abuf ^= 1;
    }
  }

  OutT* out = reinterpret_cast<OutT*>(p.out);
  const bool vec16 = (p.out_cs % 8 == 0) && (p.out_c0 % 8 == 0);
  const int ox = tox + wave * 32 + fr;
  // the stems write the largest tensors of a step (SD conv0 / fuse_conv0: 200 MB at batch 4): a lane's 16 couts go through
  // a wave-private LDS tile [32 pixels][32 couts] (the loop's buffers are free, its last barrier has passed) and leave as
  // whole 128-byte runs, eight pixels per wave instruction (see the epilogue of conv_igemm2_kernel)
  constexpr int ERS = 36;   // floats per tile row: 32 couts + 16 bytes of padding
  float* const tl = reinterpret_cast<float*>(&ldsB[0][0]) + wave * (32 * ERS);
  static_assert(4 * 32 * ERS * 4 <= (int)sizeof(ldsB), "epilogue tiles fit the segment buffers");
#pragma unroll
  for (int tc = 0; tc < TCN; ++tc) {
    const int cout_base = c0 + tc * 32 + fh * 16;
    OutT* po = out + (((size_t)tn * p.out_H + toy) * p.out_W + ox) * p.out_cs + p.out_c0 + cout_base;
    float v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      float x = acc[tc][q] * p.out_scale + ((p.bias != nullptr && cout_base + q < p.Cout) ? p.bias[cout_base + q] : 0.f);
      if (p.act == FN2_ACT_LEAKY) x = leaky(x);
      v[q] = x;
    }
    if (sizeof(OutT) == 4 && vec16 && c0 + tc * 32 + 32 <= p.Cout && !(p.dbg & 4194304)) {  // (wave-uniform)
      store16<OutT>(reinterpret_cast<OutT*>(tl + fr * ERS + fh * 16), v);
      const int rr = lane >> 3, ch = lane & 7;
      OutT* ob = out + (((size_t)tn * p.out_H + toy) * p.out_W + tox + wave * 32) * p.out_cs + p.out_c0 + c0 + tc * 32 + ch * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = i * 8 + rr;
        *reinterpret_cast<uint4*>(ob + (size_t)row * p.out_cs) = *reinterpret_cast<const uint4*>(tl + row * ERS + ch * 4);
      }
    } else if (vec16 && cout_base + 15 < p.Cout) {
      store16<OutT>(po, v);
    } else {
#pragma unroll
      for (int q = 0; q < 16; ++q)
        if (cout_base + q < p.Cout) store_elem<OutT>(po + q, v[q]);
    }
  }
#endif  // __HIP_DEVICE_COMPILE__
}

// kind-2 stems on split fp16 whose 128-pixel tiles stay inside output rows; FN2_CONV_DBG bit 65536 = off (A/B)
template <typename OutT>
static bool launch_rowrun(const ConvArgs& a, int tile, int phases, hipStream_t s) {
  if ((a.dbg & 65536) || tile != 64 || phases != 1 || a.deconv || a.KW != 1 || a.KH_KW_hint < 1 || a.splitk != 1 || a.accum || a.wfrag)
    return false;
  if (a.OW % 128 != 0 || (a.in_cs != 8 && a.in_cs != 16)) return false;
  const int spp = a.in_cs / 4;
  const int nl = a.cin_chunks >> 3;
  // slots read: the last pixel's run end, rounded up to whole 8-channel groups of whole lines
  const int max_slot = 127 * a.stride * spp + 2 * (4 * nl - 1) + 1;
  const int npieces = max_slot / 64 + 1;
  if (npieces > (spp == 4 ? 17 : 9)) return false;
  dim3 grid(a.M / 128, a.cout_pad / 64, 1), block(256);
  if (conv_name_sink().buf) {
    snprintf(conv_name_sink().buf, conv_name_sink().cap, "conv_rowrun_kernel<%s, %d>", is_x2<OutT>::value ? "fn2::x2_t" : "float", spp);
    return true;
  }
  if (spp == 4) hipLaunchKernelGGL((conv_rowrun_kernel<OutT, 4>), grid, block, 0, s, a, npieces);
  else hipLaunchKernelGGL((conv_rowrun_kernel<OutT, 2>), grid, block, 0, s, a, npieces);
  return true;
}

// kind-2 stems with at most 4 stages and at most 64 output channels; FN2_CONV_DBG bit 8192 = off (A/B)
template <typename OutT>
static bool launch_stem(const ConvArgs& a, int tile, int phases, hipStream_t s) {
  if ((a.dbg & 8192) || tile != 64 || phases != 1 || a.deconv || a.KW != 1 || a.splitk != 1 || a.accum ||
      a.ksteps < 1 || a.ksteps > 4 || a.cout_pad != 64)
    return false;
  const int ntiles = cdiv(a.M, 128);
  if (ntiles < 512) return false;  // the persistent form needs a few tiles per CU to amortise the weight fill
  const char* e = getenv("FN2_STEM_BLOCKS");
  const int blocks = std::min(ntiles, e ? atoi(e) : 256);
  if (conv_name_sink().buf) {
    snprintf(conv_name_sink().buf, conv_name_sink().cap, "conv_stem_kernel<%s, %d>",
             is_x2<OutT>::value ? "fn2::x2_t" : "float", a.ksteps);
    return true;
  }
  switch (a.ksteps) {
    case 1: hipLaunchKernelGGL((conv_stem_kernel<OutT, 1>), dim3(blocks), dim3(256), 0, s, a, ntiles); break;
    case 2: hipLaunchKernelGGL((conv_stem_kernel<OutT, 2>), dim3(blocks), dim3(256), 0, s, a, ntiles); break;
    case 3: hipLaunchKernelGGL((conv_stem_kernel<OutT, 3>), dim3(blocks), dim3(256), 0, s, a, ntiles); break;
    default: hipLaunchKernelGGL((conv_stem_kernel<OutT, 4>), dim3(blocks), dim3(256), 0, s, a, ntiles); break;
  }
  return true;
}

// stride-1 split-fp16 layers whose tiles stay inside image rows run the halo kernel; FN2_CONV_DBG bit 2048 = off (A/B)
template <typename OutT>
static bool launch_halo(const ConvArgs& a, int tile, int phases, hipStream_t s) {
  if (((a.dbg & 2048) && !a.merged) || a.stride != 1 || a.splitk != 1 || a.kg != 1 || !a.bp64 || a.wfrag || (a.KW != 2 && a.KW != 3)) return false;
  if (a.deconv && (a.kh_ph[0] != a.KH || a.kh_ph[1] != a.KH || a.kw_ph[0] != a.KW || a.kw_ph[1] != a.KW)) return false;  // trimmed phases: conv_igemm2_kernel
  // one-round 128 x 64 grids (384..512 blocks) keep the 3-slot ring: two stages of DMA in flight beat the smaller
  // stream there (conv3_1 at batch 4: ring 0.196 ms per 3 launches, halo 0.213)
  if (tile == 128 && a.bp64 == 2 && !(a.dbg & 4096)) return false;
  if (a.bp64 == 3) return false;  // deep ring chosen (conv.hip: build_args)
  const int bp = tile == 128 ? 64 : 128;
  if (a.OW % bp != 0 || a.M % bp != 0) return false;
  dim3 block(256);
#define FN2_HALO(GY, ...)                                                                         \
  do {                                                                                            \
    dim3 grid(a.M / bp, GY, phases);                                                              \
    if (conv_name_sink().buf)                                                                     \
      snprintf(conv_name_sink().buf, conv_name_sink().cap, "conv_halo_kernel<%s, %s, %d>",       \
               is_x2<OutT>::value ? "fn2::x2_t" : "float", #__VA_ARGS__, a.KW);                    \
    else if (a.KW == 3) hipLaunchKernelGGL((conv_halo_kernel<OutT, __VA_ARGS__, 3>), grid, block, 0, s, a); \
    else hipLaunchKernelGGL((conv_halo_kernel<OutT, __VA_ARGS__, 2>), grid, block, 0, s, a);      \
  } while (0)
  if (tile == 128) FN2_HALO(a.cout_pad / 128, 2, 2, 2, 1);
  else if (tile == 64) FN2_HALO(a.cout_pad / 64, 1, 4, 2, 1);
  else FN2_HALO(a.cout_pad / 32, 1, 4, 1, 1);
#undef FN2_HALO
  return true;
}

template <typename T> static const char* type_name();
template <> const char* type_name<float>() { return "float"; }
template <> const char* type_name<bf16_t>() { return "__bf16"; }
template <> const char* type_name<f16_t>() { return "_Float16"; }
template <> const char* type_name<x2_t>() { return "fn2::x2_t"; }

ConvNameSink& conv_name_sink() {
  static thread_local ConvNameSink sink{nullptr, 0};
  return sink;
}

template <typename T, typename OutT>
static int launch2(const ConvArgs& a, int tile, int phases, hipStream_t s) {
  const int z = phases * a.splitk;
  // M16 = the 16x16x32 instruction for split-fp16 operands (same LDS traffic per FLOP).  On single layers that are
  // matrix-bound it is +5..7 % (conv2 b8 312 -> 327, conv3_1 b8 308 -> 324, conv3_1 at batch 64 354 -> 380 TFLOP/s;
  // tools/ab_conv.py variants 0,1024) -- the guide's "the chip holds a higher clock on this shape".  End to end at the
  // BASELINE batch sizes it does not pay: FlowNet2 b4 4.34 -> 4.38 ms, FlowNetC b8 1.570 -> 1.562 ms with it on the
  // plain 128 x 64 layers of >= 768 blocks; on the ring +4 % time, on the 64- / 32-cout tiles +8 % / +5.5 % (runs of 4
  // output channels per lane double the store instructions of layers bound by stores and L2 -> LDS traffic).  So it
  // is OFF by default: FN2_M16_MIN = smallest plain 128 x 64 grid that takes it (e.g. 768 for large-batch serving),
  // FN2_CONV_DBG bit 1024 = on every split-fp16 tile (A/B).
  const long blocks64 = (long)cdiv(a.M, 64) * (a.cout_pad / 128) * z;
  const char* e_m16 = getenv("FN2_M16_MIN");
  const bool m16 = is_x2<T>::value && !(a.dbg & 512) &&
                   ((a.dbg & 1024) || (e_m16 && tile == 128 && a.bp64 == 1 && a.kg == 1 && a.splitk == 1 &&
                                       blocks64 >= atoi(e_m16)));
#define FN2_LAUNCH2(GX, GY, THREADS, ...)                                                              \
  do {                                                                                                 \
    dim3 grid(GX, GY, z);                                                                              \
    if (conv_name_sink().buf)                                                                          \
      snprintf(conv_name_sink().buf, conv_name_sink().cap, "conv_igemm2_kernel<%s, %s, %s, %s>", type_name<T>(), \
               type_name<OutT>(), #__VA_ARGS__, m16 ? "true" : "false");                              \
    else if (m16) hipLaunchKernelGGL((conv_igemm2_kernel<T, OutT, __VA_ARGS__, is_x2<T>::value>), grid, dim3(THREADS), 0, s, a); \
    else hipLaunchKernelGGL((conv_igemm2_kernel<T, OutT, __VA_ARGS__, false>), grid, dim3(THREADS), 0, s, a);            \
  } while (0)
  if (a.wfrag) {
    if constexpr (is_x2<T>::value) {
      if (tile == 64) {  // 64 couts x 128 pixels: two waves per 32-cout tile (each loads that tile's fragments), two pixel halves
        dim3 grid(cdiv(a.M, 128), a.cout_pad / 64, z);
        if (conv_name_sink().buf)
          snprintf(conv_name_sink().buf, conv_name_sink().cap, "conv_igemm2_kernel<%s, %s, 2, 2, 1, 2, 2, 1, false, true, %s>",
                   type_name<T>(), type_name<OutT>(), a.splitk > 1 ? "true" : "false");
        else if (a.splitk > 1)
          hipLaunchKernelGGL((conv_igemm2_kernel<T, OutT, 2, 2, 1, 2, 2, 1, false, true, true>), grid, dim3(256), 0, s, a);
        else
          hipLaunchKernelGGL((conv_igemm2_kernel<T, OutT, 2, 2, 1, 2, 2, 1, false, true, false>), grid, dim3(256), 0, s, a);
      } else {
        dim3 grid(cdiv(a.M, 64), a.cout_pad / 128, z);
        const bool slab = a.splitk > 1;
        ConvArgs ah = a;
        if (a.fh_M > 0 && slab && !conv_name_sink().buf) {
          // head pixels as extra z slices: about FN2_FH_BLOCKS (1024; 256: +30 us, 512: +10 us on FlowNet2 b4) blocks, a pixel at a time each
          const char* e = getenv("FN2_FH_BLOCKS");
          const long per_z = (long)grid.x * grid.y;
          const long want = std::min<long>(a.fh_M, e ? atoi(e) : 1024);
          ah.fh_z0 = z;
          grid.z = z + (unsigned)std::max<long>(1, (want + per_z - 1) / per_z);
        } else {
          ah.fh_M = 0;
        }
        const ConvArgs& a = ah;   // (the launches below take the copy)
        // FN2_WREG_FEED (default 1): the ring launches WITHOUT split-K take the eight-wave form (four waves feed the pixel ring,
        // four multiply): FlowNet2 b4 3.657 -> 3.60 ms.  2: the K-split ring launches as well (measured with the ring on
        // every WREG launch: 3.52 -> 3.63 ms, so not the default).  0: off.
        const char* e_feed = getenv("FN2_WREG_FEED");
        const int feed = e_feed ? atoi(e_feed) : 1;
        if (conv_name_sink().buf)
          snprintf(conv_name_sink().buf, conv_name_sink().cap, "conv_igemm2_kernel<%s, %s, 4, 1, 1, 2, %d, 1, false, true, %s%s>",
                   type_name<T>(), type_name<OutT>(), a.wfrag == 2 ? 3 : 2, slab ? "true" : "false",
                   (a.wfrag == 2 && (slab ? feed >= 2 : feed >= 1)) ? ", false, true" : "");
        else if (a.wfrag == 2 && slab && feed >= 2)
          hipLaunchKernelGGL((conv_igemm2_kernel<T, OutT, 4, 1, 1, 2, 3, 1, false, true, true, false, true>), grid, dim3(512), 0, s, a);
        else if (a.wfrag == 2 && slab)
          hipLaunchKernelGGL((conv_igemm2_kernel<T, OutT, 4, 1, 1, 2, 3, 1, false, true, true>), grid, dim3(256), 0, s, a);
        else if (a.wfrag == 2 && feed >= 1)
          hipLaunchKernelGGL((conv_igemm2_kernel<T, OutT, 4, 1, 1, 2, 3, 1, false, true, false, false, true>), grid, dim3(512), 0, s, a);
        else if (a.wfrag == 2)
          hipLaunchKernelGGL((conv_igemm2_kernel<T, OutT, 4, 1, 1, 2, 3, 1, false, true, false>), grid, dim3(256), 0, s, a);
        else if (slab)
          hipLaunchKernelGGL((conv_igemm2_kernel<T, OutT, 4, 1, 1, 2, 2, 1, false, true, true>), grid, dim3(256), 0, s, a);
        else
          hipLaunchKernelGGL((conv_igemm2_kernel<T, OutT, 4, 1, 1, 2, 2, 1, false, true, false>), grid, dim3(256), 0, s, a);
      }
    } else {
      return fail(FN2_ERR_UNSUPPORTED, "conv fast path: fragment-order weights are split fp16 only");
    }
  } else if (tile == 128 && a.bp64 && a.kg == 3) FN2_LAUNCH2(cdiv(a.M, 64), a.cout_pad / 128, 768, 2, 2, 2, 1, 2, 3);
  else if (tile == 128 && a.bp64 && a.kg == 2) FN2_LAUNCH2(cdiv(a.M, 64), a.cout_pad / 128, 512, 2, 2, 2, 1, 2, 2);
  else if (tile == 128 && a.bp64 == 3) FN2_LAUNCH2(cdiv(a.M, 64), a.cout_pad / 128, 256, 2, 2, 2, 1, 6, 1);
  else if (tile == 128 && a.bp64 == 2) FN2_LAUNCH2(cdiv(a.M, 64), a.cout_pad / 128, 256, 2, 2, 2, 1, 3, 1);
  else if (tile == 128 && a.bp64) FN2_LAUNCH2(cdiv(a.M, 64), a.cout_pad / 128, 256, 2, 2, 2, 1, 2, 1);
  else if (tile == 128) FN2_LAUNCH2(cdiv(a.M, 128), a.cout_pad / 128, 256, 2, 2, 2, 2, 2, 1);
  else if (tile == 64 && a.bp64) FN2_LAUNCH2(cdiv(a.M, 128), a.cout_pad / 64, 256, 1, 4, 2, 1, 2, 1);
  else if (tile == 64) FN2_LAUNCH2(cdiv(a.M, 256), a.cout_pad / 64, 256, 1, 4, 2, 2, 2, 1);
  else if (a.bp64) FN2_LAUNCH2(cdiv(a.M, 128), a.cout_pad / 32, 256, 1, 4, 1, 1, 2, 1);
  else FN2_LAUNCH2(cdiv(a.M, 256), a.cout_pad / 32, 256, 1, 4, 1, 2, 2, 1);
#undef FN2_LAUNCH2
  FN2_CHECK_LAUNCH("conv_igemm2");
  return FN2_OK;
}

int launch_head5_strip(const ConvArgs& a, int blocks, hipStream_t s) {
  if (a.ksteps == 3) hipLaunchKernelGGL((head5_strip_kernel<1>), dim3(blocks), dim3(512), 0, s, a);
  else if (a.ksteps == 6) hipLaunchKernelGGL((head5_strip_kernel<2>), dim3(blocks), dim3(512), 0, s, a);
  else return fail(FN2_ERR_UNSUPPORTED, "flow_head5 strip form: %d channel lines (3 or 6)", a.ksteps);
  FN2_CHECK_LAUNCH("flow_head5_strip");
  return FN2_OK;
}

int launch_head5(const ConvArgs& a, int blocks, hipStream_t s) {
  hipLaunchKernelGGL((conv_igemm2_kernel<x2_t, float, 1, 4, 2, 2, 2, 1, false, false, false, true>), dim3(blocks, 1, 1), dim3(256), 0,
                     s, a);
  FN2_CHECK_LAUNCH("flow_head5");
  return FN2_OK;
}

bool conv_fast_ok(int in_dtype, int cin_pad, int cout) {
  const int esz = dtype_size(in_dtype);
  return cout > 2 && (cin_pad * esz) % 128 == 0;
}

int launch_conv_fast(const ConvArgs& a, int in_dtype, int out_dtype, int tile, int phases, hipStream_t s) {
  if (tile != 128 && tile != 64 && tile != 32) return fail(FN2_ERR_UNSUPPORTED, "conv fast path: cout tile %d", tile);
  if (in_dtype == FN2_F32) return launch2<float, float>(a, tile, phases, s);
  if (in_dtype == FN2_BF16) {
    if (out_dtype == FN2_BF16) return launch2<bf16_t, bf16_t>(a, tile, phases, s);
    return launch2<bf16_t, float>(a, tile, phases, s);
  }
  if (in_dtype == FN2_F16) {
    if (out_dtype == FN2_F16) return launch2<f16_t, f16_t>(a, tile, phases, s);
    return launch2<f16_t, float>(a, tile, phases, s);
  }
  if (out_dtype == FN2_F16X2) {
    if (launch_rowrun<x2_t>(a, tile, phases, s)) { FN2_CHECK_LAUNCH("conv_rowrun"); return FN2_OK; }
    if (launch_stem<x2_t>(a, tile, phases, s)) { FN2_CHECK_LAUNCH("conv_stem"); return FN2_OK; }
    if (launch_halo<x2_t>(a, tile, phases, s)) { FN2_CHECK_LAUNCH("conv_halo"); return FN2_OK; }
    if (a.merged) return fail(FN2_ERR_UNSUPPORTED, "conv2d kind 5: needs the halo kernel (input width a multiple of 128, no split-K)");
    return launch2<x2_t, x2_t>(a, tile, phases, s);
  }
  if (launch_halo<float>(a, tile, phases, s)) { FN2_CHECK_LAUNCH("conv_halo"); return FN2_OK; }
  if (a.merged) return fail(FN2_ERR_UNSUPPORTED, "conv2d kind 5: needs the halo kernel (input width a multiple of 128, no split-K)");
  return launch2<x2_t, float>(a, tile, phases, s);
}

}  // namespace fn2
