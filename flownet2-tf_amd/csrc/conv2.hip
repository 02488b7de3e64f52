// Fast path of the implicit-GEMM convolution (gfx950): layers whose per-tap channel run is a
// multiple of 128 bytes (Cin_pad % 64 == 0 in bf16, % 32 in fp32) and Cout > 32.
//
// Differences from the generic kernel in conv.hip:
//   * a stage is 128 bytes of channels per operand row (two bf16 MFMA k-steps), so a stage never
//     straddles a filter tap: the tap (ky, kx) is wave-uniform scalar state and every pixel row is
//     fetched as one full 128-byte line (8 lanes x 16 B);
//   * operands go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4), no staging VGPRs and no
//     ds_write: one wave instruction fills 8 LDS rows.  The LDS image is lane-linear, so the
//     bank-conflict XOR swizzle is applied to the per-lane SOURCE address and again on the read
//     (phys chunk = chunk ^ ((row >> 1) & 7): conflict-free ds_read_b128 fragment reads);
//     out-of-image taps (the reference's explicit zero pad, utils.py:408-412) and rows past the end
//     of the pixel grid read a 16-byte zero page instead of being predicated;
//   * 2-stage software pipeline: the DMA of stage s+1 is in flight under the 32 (bf16) / 128
//     (fp32) MFMAs of stage s; one s_waitcnt vmcnt(0) + barrier per stage;
//   * weight rows are stored by the host in the permuted order that makes each lane's 16 accumulator
//     registers 16 CONSECUTIVE output channels (packed row t*16+g*4+r <-> cout g*16+t*4+r inside every
//     64-row group), so the epilogue writes 32-byte (bf16) / 64-byte (fp32) runs per pixel.
#include "conv_common.h"

namespace fn2 {

__device__ uint4 g_zero_page[4];  // all-zero source for padded taps / tail rows

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <typename OutT>
__device__ __forceinline__ void store16v(OutT* p, const float* v);
template <>
__device__ __forceinline__ void store16v<float>(float* p, const float* v) {
#pragma unroll
  for (int q = 0; q < 4; ++q)
    reinterpret_cast<float4*>(p)[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
}
template <>
__device__ __forceinline__ void store16v<bf16_t>(bf16_t* p, const float* v) {
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    bf16x8 t;
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = (bf16_t)v[8 * q + j];
    reinterpret_cast<bf16x8*>(p)[q] = t;
  }
}

template <typename T, typename OutT, int WC, int WP>
__global__ void __launch_bounds__(256) conv_igemm2_kernel(const ConvArgs p) {
  constexpr int CH = 16 / (int)sizeof(T);
  constexpr int BC = WC * 64, BP = WP * 64;
  static_assert(WC * WP == 4, "4 waves per block");
  constexpr int NWI = BC / 32;  // weight-row DMA instructions per wave per stage (8 rows each)
  constexpr int NPI = BP / 32;  // pixel-row DMA instructions per wave per stage
  constexpr int ROWS = BC + BP;
  __shared__ uint4 lds[2][ROWS * 8];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave / WP, wp = wave % WP;

  int pad_y = p.pad, pad_x = p.pad, oy_off = 0, ox_off = 0, osc = 1;
  const T* wgt = reinterpret_cast<const T*>(p.wgt);
  const int phase = blockIdx.z / p.splitk, split = blockIdx.z - phase * p.splitk;
  const size_t wrow_elems = (size_t)p.ksteps * 8 * CH;
  if (p.deconv) {
    const int a = phase >> 1, b = phase & 1;
    pad_y = 1 - a; pad_x = 1 - b; oy_off = a; ox_off = b; osc = 2;
    wgt += (size_t)phase * p.cout_pad * wrow_elems;
  }
  const int kt0 = split * p.kper;
  const int kt1 = min(p.ksteps, kt0 + p.kper);
  const int m0 = blockIdx.x * BP;
  const int c0 = blockIdx.y * BC;
  const T* in = reinterpret_cast<const T*>(p.in);

  // ---- DMA source state.  Lane -> (row lane>>3 of the instruction's 8 rows, physical chunk lane&7)
  const int lrow = lane >> 3, lphys = lane & 7;
  const T* wsrc[NWI];
#pragma unroll
  for (int j = 0; j < NWI; ++j) {
    const int row = wave * (BC / 4) + j * 8 + lrow;
    const int c = lphys ^ ((row >> 1) & 7);
    wsrc[j] = wgt + (size_t)(c0 + row) * wrow_elems + ((size_t)kt0 * 8 + c) * CH;
  }
  int iy0[NPI], ix0[NPI], pc[NPI];
  size_t pbase[NPI];
#pragma unroll
  for (int j = 0; j < NPI; ++j) {
    const int row = wave * (BP / 4) + j * 8 + lrow;
    const int m = m0 + row;
    const bool v = m < p.M;
    const int mm = v ? m : 0;
    const int n = mm / (p.OH * p.OW);
    const int rem = mm - n * (p.OH * p.OW);
    const int oy = rem / p.OW, ox = rem - oy * p.OW;
    iy0[j] = v ? oy * p.stride - pad_y : -(1 << 20);  // tail rows: never in range -> zero page
    ix0[j] = ox * p.stride - pad_x;
    pbase[j] = (size_t)n * p.H * p.W;
    pc[j] = lphys ^ ((row >> 1) & 7);
  }
  // wave-uniform tap state of stage kt0
  const int spt = p.cin_chunks >> 3;  // stages per tap
  int tap = kt0 / spt;
  int sc = kt0 - tap * spt;
  int ky = tap / p.KW, kx = tap - ky * p.KW;

  auto issue_stage = [&](int buf) {
    if (!(p.dbg & 2))
#pragma unroll
    for (int j = 0; j < NWI; ++j) {
      __builtin_amdgcn_global_load_lds((gptr_t)wsrc[j], (lptr_t)&lds[buf][(wave * (BC / 4) + j * 8) * 8], 16, 0, 0);
      wsrc[j] += 8 * CH;
    }
    const bool tap_ok = ky < p.KH;
    const int cbase = p.in_c0 + sc * 8 * CH;
    if (!(p.dbg & 1))
#pragma unroll
    for (int j = 0; j < NPI; ++j) {
      const int iy = iy0[j] + ky, ix = ix0[j] + kx;
      const bool ok = tap_ok && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      const T* src = ok ? in + (pbase[j] + (size_t)iy * p.W + ix) * p.in_cs + cbase + pc[j] * CH
                        : reinterpret_cast<const T*>(g_zero_page);
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)&lds[buf][(BC + wave * (BP / 4) + j * 8) * 8], 16, 0, 0);
    }
    if (++sc == spt) {
      sc = 0;
      if (++kx == p.KW) { kx = 0; ++ky; }
    }
  };

  // ---- fragment addresses: row (l&15) of a 16-row tile, chunk ks*4 + (l>>4), swizzled
  const int fi = lane & 15, fg = lane >> 4, fsw = (fi >> 1) & 7;
  f32x4 acc[4][4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) acc[t][pt] = f32x4{0.f, 0.f, 0.f, 0.f};

  issue_stage(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int s = kt0; s < kt1; ++s) {
    const int buf = (s - kt0) & 1;
    if (s + 1 < kt1) issue_stage(buf ^ 1);
    if (!(p.dbg & 4))
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ch = (ks * 4 + fg) ^ fsw;
      uint4 fa[4], fb[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) fa[t] = lds[buf][(wc * 64 + t * 16 + fi) * 8 + ch];
#pragma unroll
      for (int pt = 0; pt < 4; ++pt) fb[pt] = lds[buf][(BC + wp * 64 + pt * 16 + fi) * 8 + ch];
      if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int pt = 0; pt < 4; ++pt)
            acc[t][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[t]),
                                                                __builtin_bit_cast(bf16x8, fb[pt]),
                                                                acc[t][pt], 0, 0, 0);
      } else {
        // fp32 16x16x4: 40-cycle dependent latency vs 32-cycle issue -> walk the 16 accumulators
        // for each k component instead of chaining 4 MFMAs on one accumulator
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int pt = 0; pt < 4; ++pt)
              acc[t][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(f32x4, fa[t])[j],
                                                               __builtin_bit_cast(f32x4, fb[pt])[j], acc[t][pt], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- epilogue.  Packed row t*16 + g*4 + r holds cout g*16 + t*4 + r of this wave's 64-cout group,
  // so acc[0..3][pt][0..3] of a lane are 16 consecutive output channels of one pixel.
  OutT* out = reinterpret_cast<OutT*>(p.out);
  const int cout_base = c0 + wc * 64 + fg * 16;
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
      for (int t = 0; t < 4; ++t) {
        const int co = cout_base + t * 4;
        if (co < p.ws_cs)
          *reinterpret_cast<float4*>(po + co) =
              make_float4(acc[t][pt][0], acc[t][pt][1], acc[t][pt][2], acc[t][pt][3]);
      }
    }
    return;
  }
  float bias[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) bias[q] = (p.bias != nullptr && cout_base + q < p.Cout) ? p.bias[cout_base + q] : 0.f;
  const bool full16 = (cout_base + 15 < p.Cout) && (p.out_cs % 8 == 0) && (p.out_c0 % 8 == 0);
#pragma unroll
  for (int pt = 0; pt < 4; ++pt) {
    const int m = m0 + wp * 64 + pt * 16 + fi;
    if (m >= p.M) continue;
    const int n = m / (p.OH * p.OW);
    const int rem = m - n * (p.OH * p.OW);
    const int oy = rem / p.OW, ox = rem - oy * p.OW;
    OutT* po = out + (((size_t)n * p.out_H + (oy * osc + oy_off)) * p.out_W + (ox * osc + ox_off)) * p.out_cs +
               p.out_c0 + cout_base;
    float v[16];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float x = acc[t][pt][r] + bias[t * 4 + r];
        if (p.act == FN2_ACT_LEAKY) x = leaky(x);
        v[t * 4 + r] = x;
      }
    if (full16) {
      store16v<OutT>(po, v);
    } else {
#pragma unroll
      for (int q = 0; q < 16; ++q)
        if (cout_base + q < p.Cout) po[q] = from_f32<OutT>(v[q]);
    }
  }
}

template <typename T, typename OutT>
static int launch2(const ConvArgs& a, int tile, int phases, hipStream_t s) {
  dim3 block(256);
  const int z = phases * a.splitk;
  if (tile == 128) {
    dim3 grid(cdiv(a.M, 128), a.cout_pad / 128, z);
    hipLaunchKernelGGL((conv_igemm2_kernel<T, OutT, 2, 2>), grid, block, 0, s, a);
  } else {
    dim3 grid(cdiv(a.M, 256), a.cout_pad / 64, z);
    hipLaunchKernelGGL((conv_igemm2_kernel<T, OutT, 1, 4>), grid, block, 0, s, a);
  }
  FN2_CHECK_LAUNCH("conv_igemm2");
  return FN2_OK;
}

bool conv_fast_ok(int in_dtype, int cin_pad, int cout) {
  const int esz = in_dtype == FN2_BF16 ? 2 : 4;
  return cout > 32 && (cin_pad * esz) % 128 == 0;
}

int launch_conv_fast(const ConvArgs& a, int in_dtype, int out_dtype, int tile, int phases, hipStream_t s) {
  if (tile != 128 && tile != 64) return fail(FN2_ERR_UNSUPPORTED, "conv fast path: cout tile %d", tile);
  if (in_dtype == FN2_F32) return launch2<float, float>(a, tile, phases, s);
  if (out_dtype == FN2_BF16) return launch2<bf16_t, bf16_t>(a, tile, phases, s);
  return launch2<bf16_t, float>(a, tile, phases, s);
}

}  // namespace fn2
