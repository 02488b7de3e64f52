// FlowNetC cost volume on the matrix cores (gfx950).
//
// For the structure the model uses (kernel_size 1, stride_1 1, pad == max_displacement;
// src/flownet_c/flownet_c.py:40) the correlation of output row y with displaced row y+s2*p is a
// banded GEMM over channels:  G[x][x'] = sum_c A[y,x,c] * B[y+s2p, x', c],  |x'-x| <= gr*s2, and
//   out[n,y,x,(p+gr)*gw + (o+gr)] = G[x][x+s2*o] / C        (correlation_kernel.cu.cc:45-110).
//
// One block = one (n, y, 64-pixel x block).  Each of the 4 waves keeps its 16-pixel A tile in
// registers as MFMA operands for ALL channels; the displaced B row window (64 + 2*md pixels,
// zero outside the image = the reference's Pad, pad.cu.cc:46-74) is staged through LDS once
// per (p, channel group) and shared by the 4 waves; each wave multiplies its A tile with the 4
// window tiles that intersect its band.  fp32 in -> v_mfma_f32_16x16x4_f32 (exact fp32),
// bf16 in -> v_mfma_f32_16x16x32_bf16.  No padded copies, no 441-iteration serial loop, no
// lane-0 reduction.  1/C scaling, optional LeakyReLU and the write
// into a channel slice of the consumer's concat buffer (flownet_c.py:41-46) are fused.
#include "fn2_common.h"

namespace fn2 {


int correlation_geometry(int h, int w, int k, int md, int s1, int s2, int pad, int* oh, int* ow, int* gr,
                         int* gw);

struct CorrArgs {
  const void* a;
  const void* b;
  void* out;
  int N, H, W;
  int a_cs, a_c0, b_cs, b_c0, out_cs, out_c0;
  int md, s2, gr, gw, lo;  // lo = md - gr*s2
  int act;
  float c_f;  // (float)C: the reference divides by sumelems (correlation_kernel.cu.cc:105-110)
};

constexpr int CORR_NBT = 4;  // window tiles per A tile
constexpr int CORR_WIN = 4 + CORR_NBT - 1;  // window tiles per block (7 -> 112 pixels)

// NS = 64-byte channel slabs = C*sizeof(T)/64
template <typename T, typename OutT, int NS>
__global__ void __launch_bounds__(256) corr_mfma_kernel(const CorrArgs p) {
  constexpr int CH = 16 / (int)sizeof(T);
  constexpr int SG = NS < 4 ? NS : 4;  // slabs per LDS stage
  constexpr int NSTAGE = NS / SG;
  constexpr int WPX = CORR_WIN * 16;
  constexpr int STAGE_CHUNKS = WPX * SG * 4;
  constexpr int NLD = (STAGE_CHUNKS + 255) / 256;
  __shared__ uint4 lds[2][SG][WPX * 4];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int xb = blockIdx.x * 64, y = blockIdx.y, n = blockIdx.z;
  const T* A = reinterpret_cast<const T*>(p.a);
  const T* B = reinterpret_cast<const T*>(p.b);
  OutT* out = reinterpret_cast<OutT*>(p.out);
  const int fi = lane & 15, fg = lane >> 4;

  // ---- A tile -> registers, all channels (MFMA operand layout: row fi, chunk fg of each slab)
  uint4 afrag[NS];
  {
    const int x = xb + wave * 16 + fi;
    const bool ok = x < p.W;
    if constexpr (is_x2<T>::value) {
      // split fp16 -> fp32 operands: slab s = 16 channels = groups 2s, 2s+1; this lane's 4 channels
      // are j0..j0+3 of group 2s + (fg>>1): 8 bytes of hi parts and 8 bytes of lo parts
      const char* src = reinterpret_cast<const char*>(A + (((size_t)n * p.H + y) * p.W + (ok ? x : 0)) * p.a_cs + p.a_c0) +
                        (fg >> 1) * 32 + (fg & 1) * 8;
      typedef __attribute__((ext_vector_type(4))) _Float16 h4;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) {
          const h4 h = *reinterpret_cast<const h4*>(src + s * 64), l = *reinterpret_cast<const h4*>(src + s * 64 + 16);
          v = make_float4((float)h[0] + (float)l[0], (float)h[1] + (float)l[1], (float)h[2] + (float)l[2],
                          (float)h[3] + (float)l[3]);
        }
        afrag[s] = __builtin_bit_cast(uint4, v);
      }
    } else {
      const T* src = A + (((size_t)n * p.H + y) * p.W + (ok ? x : 0)) * p.a_cs + p.a_c0 + fg * CH;
#pragma unroll
      for (int s = 0; s < NS; ++s)
        afrag[s] = ok ? *reinterpret_cast<const uint4*>(src + s * 4 * CH) : make_uint4(0, 0, 0, 0);
    }
  }

  // valid displaced rows: yb = y + (pi - gr)*s2 in [0, H)
  int p_lo = 0, p_hi = p.gw - 1;
  while (p_lo <= p_hi && y + (p_lo - p.gr) * p.s2 < 0) ++p_lo;
  while (p_hi >= p_lo && y + (p_hi - p.gr) * p.s2 >= p.H) --p_hi;

  // rows displaced fully into the zero padding: the correlation is exactly 0
  for (int pi = 0; pi < p.gw; ++pi) {
    if (pi >= p_lo && pi <= p_hi) continue;
    for (int idx = lane; idx < 16 * p.gw; idx += 64) {
      const int px = idx / p.gw, o = idx - px * p.gw;
      const int x = xb + wave * 16 + px;
      if (x < p.W)
        store_elem<OutT>(out + (((size_t)n * p.H + y) * p.W + x) * p.out_cs + p.out_c0 + pi * p.gw + o, 0.f);
    }
  }
  const int niter = (p_hi - p_lo + 1) * NSTAGE;
  if (niter <= 0) return;

  // split fp16: a work item is one (pixel, group of 8 channels) = 32 bytes in, two fp32 chunks out
  constexpr int X2_ITEMS = WPX * SG * 2;
  constexpr int NLD2 = (X2_ITEMS + 255) / 256;
  uint4 regs[is_x2<T>::value ? 2 * NLD2 : NLD];
  auto load_stage = [&](int it) {
    if constexpr (is_x2<T>::value) {
      const int pi = p_lo + it / NSTAGE, st = it % NSTAGE;
      const int yb = y + (pi - p.gr) * p.s2;
      const T* rowb = B + ((size_t)n * p.H + yb) * p.W * p.b_cs + p.b_c0 + st * SG * 4 * CH;
#pragma unroll
      for (int q = 0; q < NLD2; ++q) {
        const int e = tid + 256 * q;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (e < X2_ITEMS) {
          const int px = e / (SG * 2), gi = e % (SG * 2);
          const int xw = xb - p.md + px;
          if (xw >= 0 && xw < p.W) {
            const uint4* src = reinterpret_cast<const uint4*>(rowb + (size_t)xw * p.b_cs + gi * 8);
            join8(src[0], src[1], v);
          }
        }
        regs[2 * q] = __builtin_bit_cast(uint4, make_float4(v[0], v[1], v[2], v[3]));
        regs[2 * q + 1] = __builtin_bit_cast(uint4, make_float4(v[4], v[5], v[6], v[7]));
      }
      return;
    }
    const int pi = p_lo + it / NSTAGE, st = it % NSTAGE;
    const int yb = y + (pi - p.gr) * p.s2;
    const T* rowb = B + ((size_t)n * p.H + yb) * p.W * p.b_cs + p.b_c0 + st * SG * 4 * CH;
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
      const int e = tid + 256 * q;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (e < STAGE_CHUNKS) {
        const int px = e / (SG * 4), c16 = e % (SG * 4);
        const int xw = xb - p.md + px;
        if (xw >= 0 && xw < p.W) v = *reinterpret_cast<const uint4*>(rowb + (size_t)xw * p.b_cs + c16 * CH);
      }
      regs[q] = v;
    }
  };
  auto store_stage = [&](int buf) {
    if constexpr (is_x2<T>::value) {
#pragma unroll
      for (int q = 0; q < NLD2; ++q) {
        const int e = tid + 256 * q;
        if (e < X2_ITEMS) {
          const int px = e / (SG * 2), gi = e % (SG * 2);
          const int slab = gi >> 1, c0 = (gi & 1) * 2, sw = ((px >> 3) & 1) * 3;
          lds[buf][slab][px * 4 + (c0 ^ sw)] = regs[2 * q];
          lds[buf][slab][px * 4 + ((c0 + 1) ^ sw)] = regs[2 * q + 1];
        }
      }
      return;
    }
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
      const int e = tid + 256 * q;
      if (e < STAGE_CHUNKS) {
        const int px = e / (SG * 4), c16 = e % (SG * 4);
        const int slab = c16 >> 2, cid = c16 & 3;
        lds[buf][slab][px * 4 + (cid ^ (((px >> 3) & 1) * 3))] = regs[q];
      }
    }
  };

  // first window tile this wave needs, and the per-lane fragment offset
  const int tb0 = wave + p.lo / 16;
  const int fchunk = fg ^ (((fi >> 3) & 1) * 3);

  f32x4 acc[CORR_NBT];
  load_stage(0);
  store_stage(0);
  __syncthreads();
  for (int it = 0; it < niter; ++it) {
    const int buf = it & 1;
    const int st = it % NSTAGE;
    const bool more = it + 1 < niter;
    if (more) load_stage(it + 1);
    if (st == 0) {
#pragma unroll
      for (int tb = 0; tb < CORR_NBT; ++tb) acc[tb] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int s = 0; s < SG; ++s) {
      // afrag index must be compile-time: unroll over stages
      uint4 fa = make_uint4(0, 0, 0, 0);
#pragma unroll
      for (int ss = 0; ss < NSTAGE; ++ss)
        if (ss == st) fa = afrag[ss * SG + s];
#pragma unroll
      for (int tb = 0; tb < CORR_NBT; ++tb) {
        const int tile = tb0 + tb;
        uint4 fb = make_uint4(0, 0, 0, 0);
        if (tile < CORR_WIN) fb = lds[buf][s][(tile * 16 + fi) * 4 + fchunk];
        if constexpr (sizeof(T) == 2) {
          acc[tb] = mfma_16x16x32<T>(fa, fb, acc[tb]);
        } else {
          const float4 va = __builtin_bit_cast(float4, fa), vb = __builtin_bit_cast(float4, fb);
          acc[tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(va.x, vb.x, acc[tb], 0, 0, 0);
          acc[tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(va.y, vb.y, acc[tb], 0, 0, 0);
          acc[tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(va.z, vb.z, acc[tb], 0, 0, 0);
          acc[tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(va.w, vb.w, acc[tb], 0, 0, 0);
        }
      }
    }
    if (st == NSTAGE - 1) {
      // D layout: row (A pixel) = fg*4 + r, col (window pixel) = fi
      const int pi = p_lo + it / NSTAGE;
#pragma unroll
      for (int tb = 0; tb < CORR_NBT; ++tb) {
        const int j = (tb0 + tb) * 16 + fi;  // window column
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
    if (more) store_stage(buf ^ 1);
    __syncthreads();
  }
}

template <typename T, typename OutT>
static int launch_corr(const CorrArgs& a, int C, hipStream_t s) {
  const int ns = C * (int)sizeof(T) / 64;
  dim3 grid(cdiv(a.W, 64), a.H, a.N), block(256);
  switch (ns) {
#define FN2_CORR_CASE(NS_)                                                                   \
  case NS_:                                                                                  \
    hipLaunchKernelGGL((corr_mfma_kernel<T, OutT, NS_>), grid, block, 0, s, a);              \
    break;
    FN2_CORR_CASE(1)
    FN2_CORR_CASE(2)
    FN2_CORR_CASE(4)
    FN2_CORR_CASE(8)
    FN2_CORR_CASE(16)
#undef FN2_CORR_CASE
    default:
      return fail(FN2_ERR_UNSUPPORTED, "correlation (MFMA path): unsupported channel count %d", C);
  }
  FN2_CHECK_LAUNCH("corr_mfma");
  return FN2_OK;
}

// Does the MFMA kernel cover these attributes?
static bool corr_fast_ok(int C, int esz, int k, int md, int s1, int s2, int pad) {
  if (k != 1 || s1 != 1 || pad != md) return false;
  const int ns = C * esz / 64;
  if (ns * 64 != C * esz) return false;
  if (!(ns == 1 || ns == 2 || ns == 4 || ns == 8 || ns == 16)) return false;
  const int gr = md / s2, lo = md - gr * s2;
  const int nbt = (15 + lo + 2 * gr * s2) / 16 - lo / 16 + 1;
  return nbt <= CORR_NBT && (64 + 2 * md) <= CORR_WIN * 16 && lo / 16 == 0;
}

// corr2.hip
bool corr2_ok(int C, int dtype, int md, int s2, long b_bytes);
// corr3.hip: the FlowNetC attribute set on 16-bit / split-fp16 features (same-parity row pairs and columns, LDS output tile)
bool corr3_ok(int C, int in_dtype, int out_dtype, int md, int s2, int H, int out_cs, int out_c0, long b_bytes);
int launch_corr3(const void* a, int a_cs, int a_c0, const void* b, int b_cs, int b_c0, void* out, int out_cs,
                 int out_c0, int in_dtype, int out_dtype, int N, int H, int W, int C, int md, int s2, int gr, int gw,
                 int act, hipStream_t s);
int launch_corr2(const void* a, int a_cs, int a_c0, const void* b, int b_cs, int b_c0, void* out, int out_cs,
                 int out_c0, int in_dtype, int out_dtype, int N, int H, int W, int C, int md, int s2, int gr, int gw,
                 int act, hipStream_t s);

}  // namespace fn2

using namespace fn2;

extern "C" {

int fn2_correlation_generic_f32(const float* a, const float* b, float* out, int n, int h, int w, int c,
                                int k, int md, int s1, int s2, int pad, void* stream);

int fn2_correlation_f32(const float* a, const float* b, float* out, int n, int h, int w, int c, int k, int md,
                        int s1, int s2, int pad, void* stream) {
  FN2_REQUIRE(a && b && out, "correlation: null pointer");
  FN2_REQUIRE(n >= 1 && h >= 1 && w >= 1 && c >= 1, "input_a must have rank 4");  // correlation_kernel.cc:31
  int oh, ow, gr, gw;
  int rc = correlation_geometry(h, w, k, md, s1, s2, pad, &oh, &ow, &gr, &gw);
  if (rc) return rc;
  if (k == 1 && s1 == 1 && pad == md && corr2_ok(c, FN2_F32, md, s2, (long)n * h * w * c * 4))
    return launch_corr2(a, c, 0, b, c, 0, out, gw * gw, 0, FN2_F32, FN2_F32, n, h, w, c, md, s2, gr, gw, FN2_ACT_NONE,
                        (hipStream_t)stream);
  if (!corr_fast_ok(c, 4, k, md, s1, s2, pad))
    return fn2_correlation_generic_f32(a, b, out, n, h, w, c, k, md, s1, s2, pad, stream);
  CorrArgs g;
  g.a = a; g.b = b; g.out = out;
  g.N = n; g.H = h; g.W = w;
  g.a_cs = c; g.a_c0 = 0; g.b_cs = c; g.b_c0 = 0; g.out_cs = gw * gw; g.out_c0 = 0;
  g.md = md; g.s2 = s2; g.gr = gr; g.gw = gw; g.lo = md - gr * s2;
  g.act = FN2_ACT_NONE;
  g.c_f = (float)c;
  return launch_corr<float, float>(g, c, (hipStream_t)stream);
}

// The op surface on the split-fp16 matrix-core kernel (corr3.hip): fp32 features are rewritten once as split fp16
// (x = hi + lo, 22 bits; the same 4 bytes per value) into a caller-provided workspace and multiplied with 3 fp16 MFMAs
// per product instead of fp32 MFMAs at a sixteenth of the rate.  0 bytes = this geometry stays on fn2_correlation_f32.
static bool corr_ws_path(int n, int h, int w, int c, int k, int md, int s1, int s2, int pad) {
  return k == 1 && s1 == 1 && pad == md && c % 32 == 0 &&
         corr3_ok(c, FN2_F16X2, FN2_F32, md, s2, h, 441, 0, (long)n * h * w * c * 4);
}

int64_t fn2_correlation_workspace_bytes(int n, int h, int w, int c, int k, int md, int s1, int s2, int pad) {
  if (n < 1 || h < 1 || w < 1 || c < 1) return 0;
  return corr_ws_path(n, h, w, c, k, md, s1, s2, pad) ? 2 * (int64_t)n * h * w * c * 4 : 0;
}

int fn2_correlation_f32_ws(const float* a, const float* b, float* out, int n, int h, int w, int c, int k, int md,
                           int s1, int s2, int pad, void* workspace, int64_t workspace_bytes, void* stream) {
  FN2_REQUIRE(a && b && out, "correlation: null pointer");
  FN2_REQUIRE(n >= 1 && h >= 1 && w >= 1 && c >= 1, "input_a must have rank 4");  // correlation_kernel.cc:31
  const int64_t need = fn2_correlation_workspace_bytes(n, h, w, c, k, md, s1, s2, pad);
  if (need == 0 || workspace == nullptr || workspace_bytes < need)
    return fn2_correlation_f32(a, b, out, n, h, w, c, k, md, s1, s2, pad, stream);
  int oh, ow, gr, gw;
  int rc = correlation_geometry(h, w, k, md, s1, s2, pad, &oh, &ow, &gr, &gw);
  if (rc) return rc;
  const int64_t cnt = (int64_t)n * h * w * c;
  char* wa = reinterpret_cast<char*>(workspace);
  char* wb = wa + cnt * 4;
  rc = fn2_to_f16x2(wa, a, nullptr, cnt, 1.f, stream);
  if (rc) return rc;
  rc = fn2_to_f16x2(wb, b, nullptr, cnt, 1.f, stream);
  if (rc) return rc;
  return launch_corr3(wa, c, 0, wb, c, 0, out, gw * gw, 0, FN2_F16X2, FN2_F32, n, h, w, c, md, s2, gr, gw, FN2_ACT_NONE,
                      (hipStream_t)stream);
}

int fn2_correlation_fused(const fn2_tensor* a, const fn2_tensor* b, const fn2_tensor* out, int md, int s2,
                          int act, void* stream) {
  FN2_REQUIRE(a && b && out && a->data && b->data && out->data, "correlation_fused: null tensor");
  FN2_REQUIRE(a->dtype == b->dtype, "correlation_fused: a/b dtype mismatch");
  FN2_REQUIRE(a->n == b->n && a->h == b->h && a->w == b->w && a->c == b->c,
              "input_a and input_b must have the same shape");
  FN2_REQUIRE(out->n == a->n && out->h == a->h && out->w == a->w, "correlation_fused: output spatial mismatch");
  int oh, ow, gr, gw;
  int rc = correlation_geometry(a->h, a->w, 1, md, 1, s2, md, &oh, &ow, &gr, &gw);
  if (rc) return rc;
  FN2_REQUIRE(out->c == gw * gw, "correlation_fused: output view must have %d channels", gw * gw);
  FN2_REQUIRE(out->c0 + out->c <= out->cs && a->c0 + a->c <= a->cs && b->c0 + b->c <= b->cs,
              "correlation_fused: channel slice outside the buffer");
  const int esz = dtype_size(a->dtype);
  FN2_REQUIRE((a->cs * esz) % 16 == 0 && (a->c0 * esz) % 16 == 0 && (b->cs * esz) % 16 == 0 &&
                  (b->c0 * esz) % 16 == 0,
              "correlation_fused: feature views must be 16-byte aligned");
  FN2_REQUIRE(out->dtype == a->dtype || out->dtype == FN2_F32, "correlation_fused: bad output dtype");
  if (corr3_ok(a->c, a->dtype, out->dtype, md, s2, a->h, out->cs, out->c0, (long)b->n * b->h * b->w * b->cs * esz))
    return launch_corr3(a->data, a->cs, a->c0, b->data, b->cs, b->c0, out->data, out->cs, out->c0, a->dtype,
                        out->dtype, a->n, a->h, a->w, a->c, md, s2, gr, gw, act, (hipStream_t)stream);
  if (corr2_ok(a->c, a->dtype, md, s2, (long)b->n * b->h * b->w * b->cs * esz))
    return launch_corr2(a->data, a->cs, a->c0, b->data, b->cs, b->c0, out->data, out->cs, out->c0, a->dtype,
                        out->dtype, a->n, a->h, a->w, a->c, md, s2, gr, gw, act, (hipStream_t)stream);
  if (!corr_fast_ok(a->c, esz, 1, md, 1, s2, md))
    return fail(FN2_ERR_UNSUPPORTED, "correlation_fused: C=%d md=%d s2=%d not covered by the MFMA kernel", a->c, md, s2);
  FN2_REQUIRE(out->dtype == a->dtype || out->dtype == FN2_F32, "correlation_fused: bad output dtype");
  if (a->dtype == FN2_F16X2)
    FN2_REQUIRE(a->cs % 8 == 0 && a->c0 % 8 == 0 && b->cs % 8 == 0 && b->c0 % 8 == 0 && out->cs % 8 == 0,
                "correlation_fused: split-fp16 views are group (8) aligned");
  CorrArgs g;
  g.a = a->data; g.b = b->data; g.out = out->data;
  g.N = a->n; g.H = a->h; g.W = a->w;
  g.a_cs = a->cs; g.a_c0 = a->c0; g.b_cs = b->cs; g.b_c0 = b->c0; g.out_cs = out->cs; g.out_c0 = out->c0;
  g.md = md; g.s2 = s2; g.gr = gr; g.gw = gw; g.lo = md - gr * s2;
  g.act = act;
  g.c_f = (float)a->c;
  hipStream_t s = (hipStream_t)stream;
  if (a->dtype == FN2_F32) return launch_corr<float, float>(g, a->c, s);
  if (a->dtype == FN2_BF16) {
    if (out->dtype == FN2_BF16) return launch_corr<bf16_t, bf16_t>(g, a->c, s);
    return launch_corr<bf16_t, float>(g, a->c, s);
  }
  if (a->dtype == FN2_F16) {
    if (out->dtype == FN2_F16) return launch_corr<f16_t, f16_t>(g, a->c, s);
    return launch_corr<f16_t, float>(g, a->c, s);
  }
  // split fp16 features: converted to fp32 operands on the way in (fp32 MFMA), split again on the way out
  if (out->dtype == FN2_F16X2) return launch_corr<x2_t, x2_t>(g, a->c, s);
  return launch_corr<x2_t, float>(g, a->c, s);
}

}  // extern "C"
