/* flownet2_hip.h -- C ABI of libflownet2_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the FlowNet2 hot path of fperezgamonal/flownet2-tf
 * (SURVEY.md section 8b).  The reference exposes its native code through the
 * TensorFlow plugin ABI (tf.load_op_library + OpKernel::Compute); the plain
 * functions under that ABI are what each entry point below replaces:
 *
 *   fn2_correlation_f32        <- Correlation(...)      src/ops/correlation/correlation_kernel.h:11-29
 *                                 + Pad(...)            src/ops/correlation/pad.h:9-17
 *                                 (op: src/correlation.py:7-14)
 *   fn2_correlation_grad_f32   <- CorrelationGradA/B    src/ops/correlation/correlation_kernel.h:32-74
 *                                 (op: src/correlation.py:17-35)
 *   fn2_flow_warp_f32          <- FlowWarp(...)         src/ops/flow_warp/flow_warp.h:15-18
 *                                 (op: src/flow_warp.py:7-8)
 *   fn2_flow_warp_grad_f32     <- FlowWarpGrad(...)     src/ops/flow_warp/flow_warp.h:20-25
 *                                 (op: src/flow_warp.py:11-15)
 *   fn2_downsample_f32         <- Downsample(...)       src/ops/downsample/downsample_kernel.h:12-14
 *                                 (op: src/downsample.py:7-8)
 *   fn2_resize_bilinear_f32,
 *   fn2_conv2d, fn2_upsample_flow, fn2_* fused elementwise
 *                              <- the TensorFlow builtins the model files call
 *                                 (slim.conv2d / conv2d_transpose / resize_bilinear /
 *                                 concat / LeakyReLU: src/flownet_s/flownet_s.py:26-111,
 *                                 src/utils.py:401-421), which the reference gets from
 *                                 TensorFlow + cuDNN (not vendored).
 *
 * Conventions
 *   - Every pointer is a DEVICE pointer owned by the caller (the reference lets
 *     TensorFlow own all memory: correlation_kernel.cc:61-80).  The library
 *     allocates nothing and keeps no mutable global state except the
 *     thread-local last-error string.
 *   - Tensors are NHWC.  Op-surface entry points (suffix _f32) take dense
 *     float32 tensors exactly like the reference ops.  Engine entry points take
 *     an fn2_tensor view: dtype + channel stride + first channel, so that a
 *     layer can read/write a channel slice of a concat buffer without copies.
 *   - `stream` is a hipStream_t (NULL = default stream); all work is enqueued
 *     on it, nothing synchronises.
 *   - Return value: FN2_OK (0) or a negative fn2_status; fn2_last_error() gives
 *     the message.  Argument conditions mirror the reference's OP_REQUIRES
 *     checks (correlation_kernel.cc:23,31-32,50-53; flow_warp.cc:22-30;
 *     downsample_kernel.cc:19,25-26).
 */
#ifndef FLOWNET2_HIP_H_
#define FLOWNET2_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  FN2_OK = 0,
  FN2_ERR_INVALID_ARGUMENT = -1, /* shape / attribute check failed (errors::InvalidArgument) */
  FN2_ERR_UNSUPPORTED = -2,      /* valid but not implemented for this dtype/shape */
  FN2_ERR_HIP = -3               /* a HIP runtime call or kernel launch failed */
} fn2_status;

/* FN2_F16X2: "split fp16" storage of fp32-grade values: every group of 8 consecutive channels is 16 bytes
 * of fp16 hi parts followed by 16 bytes of fp16 lo parts (x = hi + lo, 22 significant bits, 4 bytes per
 * channel like fp32).  Products are formed on the fp16 matrix cores as hi*hi + hi*lo + lo*hi with fp32
 * accumulation: fp32-grade results at 3/16 of the fp32-MFMA cost.  Channel offsets/strides of such views
 * must be multiples of 8. */
typedef enum { FN2_F32 = 0, FN2_BF16 = 1, FN2_F16 = 2, FN2_F16X2 = 3 } fn2_dtype;

typedef enum { FN2_ACT_NONE = 0, FN2_ACT_LEAKY = 1 /* 0.55x + 0.45|x|, utils.py:401-405 */ } fn2_act;

/* NHWC view of (a channel slice of) a device buffer. */
typedef struct {
  void* data;     /* element (n=0,y=0,x=0,channel 0 of the BUFFER) */
  int32_t dtype;  /* fn2_dtype */
  int32_t n, h, w;
  int32_t c;      /* logical channels of the view */
  int32_t cs;     /* channel stride of the buffer in elements (>= c0 + c) */
  int32_t c0;     /* first channel of the view inside the buffer */
} fn2_tensor;

const char* fn2_last_error(void);
int fn2_version(void);
/* CRC-32C (Castagnoli, reflected 0x82F63B78) of n bytes of HOST memory, continuing from `crc` (0 to start): the
 * checksum of the TensorFlow checkpoint bundle and TFRecord containers the reference reads and writes through
 * tf.train.Saver (src/net.py:566-569, :1386-1392) and tf.data.TFRecordDataset (src/dataloader.py).  Host-only. */
uint32_t fn2_crc32c(const void* data, int64_t n, uint32_t crc);
/* Fills name (<= cap bytes) with the gcnArchName of the current device, *cus with its CU count. */
int fn2_device_info(char* name, int cap, int* cus);

/* ---------------------------------------------------------------- op surface (float32, dense NHWC) */

/* Output geometry of the correlation op (shape fn correlation_op.cc:9-51). */
int fn2_correlation_out_shape(int h, int w, int kernel_size, int max_displacement, int stride_1,
                              int stride_2, int pad, int* out_h, int* out_w, int* out_c);

/* out[n,y,x,d] = 1/(k*k*C) * sum_{j,i,c} A0[n,y1+j,x1+i,c] * B0[n,y1+s2p+j,x1+s2o+i,c]
 * (correlation_kernel.cu.cc:45-110); the zero padding is fused (no padded copy). */
int fn2_correlation_f32(const float* a, const float* b, float* out, int n, int h, int w, int c,
                        int kernel_size, int max_displacement, int stride_1, int stride_2, int pad,
                        void* stream);
/* The same op with a caller-provided scratch buffer (the reference allocates its padded copies through
 * ctx->allocate_temp, correlation_kernel.cc:61-80): for the FlowNetC attribute set (kernel 1, max_displacement 20,
 * stride_2 2, pad 20; C % 32 == 0, H % 8 == 0) the features are rewritten as split fp16 (hi + lo, 22 bits) in the
 * workspace and the cost volume runs on the fp16 matrix cores (3 MFMAs per product); other geometries, a NULL or a too
 * small workspace fall through to fn2_correlation_f32.  fn2_correlation_workspace_bytes returns 0 for those. */
int64_t fn2_correlation_workspace_bytes(int n, int h, int w, int c, int kernel_size, int max_displacement, int stride_1,
                                        int stride_2, int pad);
int fn2_correlation_f32_ws(const float* a, const float* b, float* out, int n, int h, int w, int c, int kernel_size,
                           int max_displacement, int stride_1, int stride_2, int pad, void* workspace,
                           int64_t workspace_bytes, void* stream);

/* grad_a, grad_b (same shape as a) from grad_out (correlation_grad_kernel.cu.cc:20-189). */
int fn2_correlation_grad_f32(const float* grad_out, const float* a, const float* b, float* grad_a,
                             float* grad_b, int n, int h, int w, int c, int kernel_size,
                             int max_displacement, int stride_1, int stride_2, int pad, void* stream);

/* out[n,y,x,:] = bilinear(image[n], x+u, y+v) inside the image else 0 (flow_warp.cu.cc:44-95). */
int fn2_flow_warp_f32(const float* image, const float* flow, float* out, int n, int h, int w, int c,
                      void* stream);

/* image_grad (zeroed then scatter-added) and flow_grad (flow_warp_grad.cu.cc:30-86). */
int fn2_flow_warp_grad_f32(const float* image, const float* flow, const float* grad_out,
                           float* image_grad, float* flow_grad, int n, int h, int w, int c,
                           void* stream);

/* NaN-aware triangle-weighted area downsample (downsample_kernel_gpu.cu.cc:35-76). */
int fn2_downsample_f32(const float* in, float* out, int n, int in_h, int in_w, int c, int out_h,
                       int out_w, void* stream);

/* The same op on (in_scale * in), the product formed per sample in fp32 before it is weighted: FlowNetS.loss downsamples
 * `flow = flow * 0.05` (flownet_s.py:123-129; 20 * flow in flownet_sd.py:122) -- one pass instead of a scaling pass + the op. */
int fn2_downsample_scaled_f32(const float* in, float in_scale, float* out, int n, int in_h, int in_w, int c, int out_h,
                              int out_w, void* stream);

/* tf.image.resize_bilinear(align_corners=True) of (scale * in). */
int fn2_resize_bilinear_f32(const float* in, float* out, int n, int in_h, int in_w, int c, int out_h,
                            int out_w, float scale, void* stream);

/* ---------------------------------------------------------------- engine layer kernels */

/* One convolution layer as an implicit GEMM on MFMA (fp32: v_mfma_f32_16x16x4_f32,
 * bf16: v_mfma_f32_16x16x32_bf16), zero padding / bias / LeakyReLU / concat-slice write fused.
 *   kind 0: slim.conv2d(pad(x,p), Cout, k, stride, 'VALID')        (flownet_s.py:39-50)
 *   kind 1: antipad(slim.conv2d_transpose(x, Cout, 4, 2, 'VALID')) (flownet_s.py:53-63), computed as
 *           four 2x2 stride-1 phase convolutions; bias NULL inside the refinement scopes (biases_initializer=None,
 *           flownet_s.py:53), non-NULL for the FlowNet2 fusion net's fuse_deconv1 / fuse_deconv0 (flownet2.py:66-84).
 *   kind 2: the same convolution as kind 0 for the few-channel network inputs (3/6/11/12 channels): the
 *           input view is the WHOLE pre-padded buffer [n, H+2p, W+2p, cs] (in.c == in.cs, in.c0 == 0,
 *           desc.pad == 0), and the kw horizontal taps x cs channels of one kernel row are read as ONE
 *           contiguous run of cin_pad >= kw*cs channels (packed weight k = ky*cin_pad + kx*cs + c, zero
 *           beyond kw*cs), so a 7x7x3 stem costs K = 7*64 instead of 49*8..64 and runs on the LDS-DMA kernel.
 *   kind 3: the transposed convolution that is the input-gradient of a stride-2 kind-0 layer (kh = kw = k,
 *           pad = p of THAT layer): four phases of ceil(k/2)^2 taps; packed weight [4][cout_pad][T*T*cin_pad]
 *           with tap (ty,tx) of phase (a,b) = W[ky = a+p-2*lo_a-2*ty][kx likewise], lo_a = ceil((a+p-k+1)/2).
 *   kind 5: kind 1 for 16 / 32 output channels on split fp16 over maps whose width is a multiple of 128 (the fusion net's
 *           fuse_deconv1 / fuse_deconv0, flownet2.py:66-84): two row phases; each holds both column phases of its output
 *           rows as one 2 x 3-tap convolution with 2 Cout packed rows (row b Cout + co -> pixel (2y+a, 2x+b), channel co;
 *           tap (ty, kx3) reads input (y-1+a+ty, x-1+kx3); phase b has W[3-a-2ty][3-b-2(kx3-b)] at kx3 in {b, b+1}, zero in
 *           its third slot); cout_pad = 2 Cout, kpad >= 6 cin_pad.  Half the passes over the input of kind 1.
 * `wgt` is the layer's weight pre-packed by fn2_pack_* layout rules (see DESIGN.md "weights"):
 *   [phase][cout_pad][kpad] elements of in.dtype, k = (tap, channel) with channels padded to a
 *   multiple of 8, kpad a multiple of the k-step; cout_pad a multiple of the block's cout tile (fn2_conv2d_plan). */
typedef struct {
  fn2_tensor in;       /* in.c = logical Cin; in.cs and in.c0 multiples of 8 */
  fn2_tensor out;      /* out.c = Cout; out.dtype may be FN2_F32 while in is bf16 (flow heads) */
  const void* wgt;
  const float* bias;   /* [Cout] fp32 or NULL */
  int32_t kind;        /* 0 conv, 1 deconv 4x4 s2 crop 1, 2 stem row-run conv, 3 transpose of a stride-2 conv, 5 = 1 with merged column phases */
  int32_t kh, kw, stride, pad;
  int32_t act;         /* fn2_act */
  int32_t cin_pad;     /* channels per tap in the packed weight (multiple of 8, >= in.c) */
  int32_t cout_pad;    /* rows per phase in the packed weight */
  int32_t kpad;        /* elements per packed row */
  int32_t wgt_layout;  /* row order of the packed weight: fn2_conv_plan.layout -- or 2 where the plan says 1, the input is
                          split fp16 and the cout tile is 128 (64): layout 1's matrix re-tiled into MFMA-FRAGMENT order,
                          per 32-row tile t and 128-byte stage s four 1 KiB blocks f = 2 q + part at byte
                          ((t * kpad / 32 + s) * 4 + f) * 1024, lane 32 h + r of a block = the 16-byte chunk
                          4 q + 2 h + part of row 32 t + r (part 0 = hi, 1 = lo halves of an 8-channel group).  The launch
                          then loads the weight operand straight into registers (conv2.hip, WREG); the same values,
                          the same result up to fp32 summation order.  fn2_to_f16x2_frag writes this layout. */
  int32_t accumulate;  /* 1: out += result (fp32 outputs only): gradient accumulation */
  float out_scale;     /* accumulator scale applied before the bias (0 = 1): lets the packer store
                          2^k-scaled weights so that split-fp16 lo parts stay normal fp16 numbers */
  void* workspace;     /* fp32 scratch for split-K partial sums, or NULL (then no split-K) */
  int64_t workspace_bytes;
  /* Backward of LeakyReLU fused into the layer that COMPLETES a gradient slice (training; the reference gets the same
   * product from tf.gradients of utils.py:401-405): after out (+)= result, channels [act_grad_c0, act_grad_c1) of the
   * out view are multiplied by 1 / 0.1 / 0.55 for a forward output y > 0 / < 0 / == 0, y read from act_grad_y -- the
   * forward activation buffer the gradient buffer mirrors: same dtype, geometry and strides as out.data, element for
   * element.  NULL: off.  fp32 and split-fp16 outputs, act == FN2_ACT_NONE. */
  const void* act_grad_y;
  int32_t act_grad_c0, act_grad_c1;
  /* upsample_flowXtoY riding on the transposed conv of the same decoder level (flownet_s.py:57-63: deconvN and
   * upsample_flow(N+1)toN write neighbouring channel slices of one concat tensor): when up_src is not NULL the call also
   * writes fn2_upsample_flow(up_src, up_w, up_bias) -- up_src = predict_flow(N+1), dense fp32 [n, out.h/2, out.w/2, 2] --
   * to channels [up_c0, up_c0 + 2) of out's buffer.  A split-K launch does it in its finalize pass (one launch less on the
   * decoder chain); any other launch is followed by the stand-alone kernel.  Same values either way.  kind 1 only. */
  const float* up_src;
  const float* up_w;     /* [4][4][2][2] HW-O-I */
  const float* up_bias;  /* [2] or NULL */
  int32_t up_c0;
  /* predict_flow(N+1) riding on the same launch (flownet_s.py:54-59: the head and deconvN read the same tensor): `head` is
   * the complete descriptor of a 3x3 two-output fp32 flow head (kind 0, k 3, s 1, p 1, no activation) whose `in` view is
   * this descriptor's `in` view.  The call computes it as well, BEFORE up_src is read (so up_src may be head->out.data):
   * as extra blocks of a split-K split-fp16 launch on fragment-order weights, else as fn2_conv2d(head) in front of this
   * launch.  Same values either way.  kind 1 only; NULL: off. */
  const void* head;      /* const fn2_conv_desc* */
  /* 1: a launch that splits K leaves its S raw fp32 partial-sum slabs in `workspace` -- [S][n*h*w][round_up(Cout, 4)], no
   * bias, no out_scale -- and writes nothing to `out`: the consumer sums them (fn2_flow_head_tail_slabs does, for the
   * GEMM-form flow heads: one launch less per head).  S = fn2_conv2d_splits(desc); with S == 1 the call is the ordinary one. */
  int32_t raw_partials;
} fn2_conv_desc;

/* How fn2_conv2d runs a layer of this (input dtype, padded Cin, Cout), i.e. how its weight must be packed:
 *   cout_tile   rows per block: cout_pad must be a multiple;
 *   kstep_elems kpad must be a multiple (elements);
 *   wgt_dtype   element type of the packed weight (FN2_F16X2 activations: split pairs on the LDS-DMA
 *               kernel, plain fp32 on the generic kernel and the flow heads);
 *   layout      0: row r = output channel r;
 *               1: LDS-DMA kernel: inside every group of 32 rows, row (r&3)+8(r>>2)+4h = output channel
 *                  16h+r (h in 0..1, r in 0..15), so that a lane's 16 accumulators are 16 consecutive
 *                  channels. */
typedef struct {
  int32_t layout, cout_tile, kstep_elems, wgt_dtype;
} fn2_conv_plan;
int fn2_conv2d_plan(int in_dtype, int cin_pad, int cout, fn2_conv_plan* plan);
/* Bytes of workspace with which this layer would use its preferred split-K factor (0 = none needed).
 * The caller owns the workspace (the reference: ctx->allocate_temp, correlation_kernel.cc:66-80);
 * one buffer of the maximum over layers can be shared by all launches on a stream. */
int64_t fn2_conv2d_workspace_bytes(const fn2_conv_desc* d);
int fn2_conv2d(const fn2_conv_desc* d, void* stream);
/* The device kernel (template instantiation, as rocprofv3 prints it without "void fn2::" and the argument list) that
 * fn2_conv2d would launch for this descriptor under the current tuning knobs -- the tile / ring / K-group / halo choice is
 * made inside the library; profiles and the bench's per-kernel table name launches by it.  "" for the paths that do
 * not report (generic kernel, flow heads).  Launches nothing. */
int fn2_conv2d_kernel_name(const fn2_conv_desc* desc, char* name, int cap);
/* Number of K splits fn2_conv2d(desc) takes with desc's workspace (1 = no split-K: no slabs, no finalize pass). */
int fn2_conv2d_splits(const fn2_conv_desc* desc);

/* Flow head (predict_flowN: 3x3, stride 1, pad 1, 2 outputs; flownet_s.py:54-56) as a GEMM: run fn2_conv2d as a
 * 1x1 convolution with 18 outputs t[pix][tap*2+co] (weight w1x1[ci][tap*2+co] = w[ky][kx][ci][co]) into an fp32
 * scratch tensor, then this call: out[n,y,x,co] = bias[co] + sum_taps t[n,y+ky-1,x+kx-1][tap*2+co].  out dense [n,h,w,2]. */
int fn2_flow_head_gather(const float* t, int t_cs, const float* bias, float* out, int n, int h, int w, void* stream);
/* The tail of a flow head in one launch: fn2_flow_head_gather generalised to taps x taps (3 or 5) shifted partials
 *   pf[n,y,x,co] = bias[co] + sum_{ky,kx < taps} t[n, y+ky-taps/2, x+kx-taps/2][(ky*taps+kx)*2 + co]   (zero outside),
 * written dense fp32 [n,h,w,2], followed -- when up_w is not NULL -- by upsample_flowXtoY on it (fn2_upsample_flow:
 * flownet_s.py:60-63; up_bias NULL except in the fusion net, flownet2.py:70-73, :86-89) into the [n,2h,2w,2] view up_out.
 * taps = 5 serves a COMPOSED head: where the reference applies a linear 3x3 interconvN and then the linear 3x3
 * predict_flowN to it (flownet_sd.py:60-64 ..., flownet2.py:74-77, :90-93; activation_fn=None on both), the two are
 * one 5x5 convolution of the concat buffer with weights w5[u] = sum_{t+s=u} w2[t] w1[s], except on the image's outermost
 * pixel ring (the reference zero-pads the interconv OUTPUT): ring != 0 says those pixels were already written to pf by
 * fn2_flow_head_ring and must be taken from there. */
int fn2_flow_head_tail(const float* t, int t_cs, int taps, const float* bias, float* pf, int n, int h, int w, int ring,
                       const float* up_w, const float* up_bias, const fn2_tensor* up_out, void* stream);
/* The same with t given as `nslab` raw split-K slabs (fn2_conv_desc.raw_partials): slab s at t + s * slab_stride floats,
 * [n*h*w][t_cs] each; a partial = scale * (slab 0 + slab 1 + ...), the sum in split order -- bit for bit what the GEMM's
 * finalize pass would have stored. */
int fn2_flow_head_tail_slabs(const float* t, int t_cs, int nslab, int64_t slab_stride, float scale, int taps, const float* bias,
                             float* pf, int n, int h, int w, int ring, const float* up_w, const float* up_bias,
                             const fn2_tensor* up_out, void* stream);
/* The interior of a composed head in ONE launch, without the partials ever leaving the chip: for every pixel not on the
 * outermost ring (all pixels when ring == 0)
 *   pf[n,y,x,o] = bias[o] + sum_{u in 5x5} sum_ci w5[u][ci][o] x[n, y+uy-2, x+ux-2, ci]            (x zero outside the image)
 * x: split-fp16 view whose channel run is whole 128-byte lines (cin_pad % 32 == 0, c0 + cin_pad <= cs); wgt: the 1x1
 * matrix [64 rows (50 used: (uy*5+ux)*2+o)][cin_pad] in fn2_conv2d's wgt_layout 1 for a 64-row tile, split fp16, scaled by
 * 1 / out_scale; pf dense fp32 [n,h,w,2].  Two forms, chosen by the library: cin_pad = 96 or 192 -> a block walks down a
 * strip of 60 output columns, forms the 50 partials of one input row at a time on the matrix cores and adds them, per tap
 * row, into a five-row accumulator ring in LDS (input read ~1.15 times); else a block forms the partials of the 8 x 32
 * positions around a 4 x 28 output tile, keeps them in LDS and sums the 25 shifted ones per output.  The order of the 25
 * additions differs between the two forms (strip: kx inside ky, bias last), so they agree to fp32 rounding, not bit for bit.
 * ring != 0 with ring_w / ring_b (fn2_flow_head_ring's wc / bc): the same launch computes the ring pixels (extra blocks
 * of the tile form; an even share at the end of every block of the strip form);
 * with ring_w == NULL they are left as they are (written by fn2_flow_head_ring, or not needed). */
int fn2_flow_head5(const fn2_tensor* x, const void* wgt, int cin_pad, int kpad, float out_scale, const float* bias, float* pf,
                   int ring, const float* ring_w, const float* ring_b, void* stream);
/* Border ring of a composed head: pf[n,y,x,o] = bc[case][o] + sum_{u in 5x5} sum_ci wc[case][u][ci][o] x[n, y+uy-2, x+ux-2, ci]
 * for the pixels with y in {0, h-1} or x in {0, w-1}; case = 3*cy + cx with c = 0 / 1 / 2 for the low border / interior /
 * high border of that axis.  x: split-fp16 view; wc: fp32 [9][25][8*ceil(c/8)][2], zero on the pad channels; bc: fp32 [9][2]. */
int fn2_flow_head_ring(const fn2_tensor* x, const float* wc, const float* bc, float* pf, void* stream);
/* upsample_flowXtoY: 2->2 channel conv-transpose 4x4 s2 crop 1, linear (flownet_s.py:60-63).
 * in: fp32 [n,h,w,2] dense; w: fp32 [4][4][2 out][2 in] (reference HW-O-I layout); out: view with c=2.
 * bias: fp32 [2] or NULL.  NULL inside the refinement scopes of S / C / SD (biases_initializer=None, flownet_s.py:53,
 * flownet_c.py:58, flownet_sd.py:44); the FlowNet2 fusion net's fuse_upsample_flow2to1 / 1to0 carry one
 * (flownet2.py:50-57 opens no such scope; :70-73, :86-89). */
int fn2_upsample_flow(const float* in, const float* w, const float* bias, const fn2_tensor* out, int n, int h, int wd,
                      void* stream);

/* uint8 image bytes -> fp32 [0,1] after the host-to-device copy: replaces the host side of Net.adapt_x's
 * normalisation (src/net.py:338-345: `x / 255.0` when the image's max exceeds 1) so that images cross PCIe as one
 * byte per channel.  dst[i] = lut256[src[i]]; the caller supplies the table (float32(float64(i) / 255.0), or
 * float32(i) for an image whose max is <= 1), which makes the result byte-identical to the host arithmetic.
 * src, dst: device, 16-byte aligned; count = number of bytes. */
int fn2_u8_to_f32_lut(const unsigned char* src, const float* lut256, float* dst, long count, void* stream);

/* The four "network input" builders below write the INTERIOR of a view whose h, w include a zero border of
 * `pad` pixels on every side (allocated zero by the caller, never written): the reference's pad() in front
 * of the stem convolution (utils.py:408-412) is baked into the buffer, so the stem can run as a kind-2
 * row-run convolution.  pad = 0 gives a dense tensor.
 * images fp32 [n,h,w,3] x2 -> out view with 6 (pad 8) channels [a | b | 0 0]  (flownet_s.py:24) */
int fn2_pack_pair(const float* a, const float* b, const fn2_tensor* out, int pad, void* stream);
/* image fp32 [n_img,h,w,3] -> batch rows [n0, n0+n_img) of the out view, 3 (pad 8) channels
 * (the siamese towers of FlowNetC run as one 2N batch, flownet_c.py:30-37) */
int fn2_pack_image(const float* img, int n_img, const fn2_tensor* out, int n0, int pad, void* stream);
/* Space-to-depth variant for a stride-2 stem on a 3-channel image (FlowNetC's pad(.., 3) + conv1 7x7 stride 2 on
 * each image, flownet_c.py:30-37): out[n, sy, sx, (py*2+px)*4 + c] =
 * zero-padded img[2sy+py-pad, 2sx+px-pad, c]; out is the WHOLE [n_total, (h+2pad)/2, (w+2pad)/2, 16] buffer.  The
 * k x k stride-2 convolution becomes a ceil(k/2)^2 stride-1 kind-2 convolution with the weights re-indexed
 * w'[ky', kx', (py*2+px)*4+c] = w[2ky'+py, 2kx'+px, c] (zero past k). */
int fn2_pack_image_s2d(const float* img, int n_img, int h, int w, const fn2_tensor* out, int n0, int pad, void* stream);

/* FlowNetC correlation inside the engine: a, b views over conv3 features (C multiple of 32),
 * out = LeakyReLU(correlation(a, b, 1, md, 1, s2, md)) written into a channel slice
 * (flownet_c.py:40-46). */
int fn2_correlation_fused(const fn2_tensor* a, const fn2_tensor* b, const fn2_tensor* out,
                          int max_displacement, int stride_2, int act, void* stream);

/* Stacked-net input (flownet_cs.py:21-36): out view (16-channel padded) =
 * [a(3) | b(3) | warp(b, flow)(3) | flow*0.05(2) | sqrt(sum_c (a-warp)^2)(1) | 0...]. */
int fn2_stack_input(const float* a, const float* b, const float* flow, const fn2_tensor* out, int pad,
                    void* stream);

/* FlowNet2 fusion input (flownet2.py:25-47): out view (16-channel padded) =
 * [a(3) | flow_sd(2) | flow_css(2) | |sd| | |css| | |a-warp(b,sd)| | |a-warp(b,css)| | 0...]. */
int fn2_fusion_input(const float* a, const float* b, const float* flow_sd, const float* flow_css,
                     const fn2_tensor* out, int pad, void* stream);
/* The same two ops fed with the quarter-resolution predict_flow2 tensors ([n, pf_h, pf_w, 2] fp32) the flows are resized
 * from -- flow = resize_bilinear(scale * predict_flow2) (align_corners; flownet_s.py:105-109, flownet_c.py:113-118,
 * flownet_sd.py:104-108): every pixel interpolates its own flow vector with fn2_resize_bilinear_f32's arithmetic, so the
 * resize launch in front of the op disappears from a stack's chain; flow_out (NULL: skip) receives the full-resolution
 * flow as that launch would have written it.  Bit-identical to resize + op. */
int fn2_stack_input_pf(const float* a, const float* b, const float* pf, int pf_h, int pf_w, float scale, float* flow_out,
                       const fn2_tensor* out, int pad, void* stream);
int fn2_fusion_input_pf(const float* a, const float* b, const float* pf_sd, float scale_sd, float* flow_sd_out,
                        const float* pf_css, float scale_css, float* flow_css_out, int pf_h, int pf_w,
                        const fn2_tensor* out, int pad, void* stream);

/* ---------------------------------------------------------------- training step (fp32)
 * What the reference obtains from tf.gradients + tf.train.AdamOptimizer over the FlowNetS loss
 * (src/net.py:1290-1295, :1386-1392; src/flownet_s/flownet_s.py:122-161; src/utils.py:209-224).
 * Input gradients of the conv layers are fn2_conv2d launches (kind 0 on rotated weights, kind 3,
 * kind 0 k4 s2 for the transposed convs) with `accumulate`. */

/* L = weight * sum_pixels ||pred - label||_2 / n added to *loss_accum (device scalar);
 * dpred = grad_mult * weight / n * (pred - label) / ||pred - label||  (average_endpoint_error, utils.py:209-224).
 * grad_mult: loss scaling of the split-fp16 trainer (a power of two that keeps the activation gradients inside
 * the fp16 exponent range; removed again by fn2_adam_step's grad_scale), 1 otherwise. */
int fn2_epe_loss_grad(const float* pred, const float* label, float* dpred, float* loss_accum, int n, int h, int w,
                      float weight, float grad_mult, void* stream);
/* The same with a per-pixel weight [n,h,w] (fp32): the hard-flow-example-mining variants of
 * average_endpoint_error_hfem (src/utils.py:227-339) -- 'hard': weight = (1 + lambda) * #pixels / #hard on the top-k
 * EPE pixels of the batch and 0 elsewhere; 'edges': 1 + lambda * edges.  L = weight / n * sum w_pix ||pred - label||. */
int fn2_epe_loss_grad_weighted(const float* pred, const float* label, const float* pixel_weight, float* dpred,
                               float* loss_accum, int n, int h, int w, float weight, float grad_mult, void* stream);
/* g *= LeakyReLU'(.) evaluated from the layer output y (utils.py:401-405); y, g fp32 channel-slice views.
 * db != NULL: the bias gradient of the same layer in the same pass, db[c] += sum over pixels of the new g. */
int fn2_leaky_bwd(const fn2_tensor* y, const fn2_tensor* g, float* db, void* stream);
/* db[c] += sum over pixels of g (db zeroed by the caller). */
int fn2_bias_grad(const fn2_tensor* g, float* db, void* stream);
/* dst (split fp16, n logical elements, n % 8 == 0) = scale * (map ? src[map[i]] (0 where map[i] < 0) : src[i]): the
 * forward and backward-data weight copies of the split-fp16 trainer, re-derived from the fp32 master every step
 * (scale = the power of two of the layer's descriptor, out_scale = 1 / scale). */
int fn2_to_f16x2(void* dst, const float* src, const int32_t* map, int64_t n, float scale, void* stream);
/* The same into fn2_conv2d's wgt_layout 2 (MFMA-fragment order; see fn2_conv_desc.wgt_layout): src is the packed
 * [rows][k] matrix (rows % 32 == 0, k % 32 == 0, n = rows * k); only WHERE each 16-byte (hi / lo) chunk lands differs. */
int fn2_to_f16x2_frag(void* dst, const float* src, const int32_t* map, int64_t n, float scale, int k, void* stream);
/* Housekeeping of the train step on the caller's stream (what tf.gradients' accumulators and tf.zeros do in the
 * reference graph): zero a gradient buffer; dst += src (the correlation's two input gradients joining the towers'
 * gradient buffers, correlation.py:17-35); a dense copy of a channel slice of an fp32 NHWC buffer. */
int fn2_fill_zero(void* dst, int64_t bytes, void* stream);
int fn2_add_f32(float* dst, const float* src, int64_t n, void* stream);
int fn2_slice_copy_f32(const fn2_tensor* src, float* dst, void* stream);
/* dst[i] = map[i] >= 0 ? src[map[i]] : 0: derives the weight layouts of the input-gradient convolutions. */
int fn2_gather_f32(float* dst, const float* src, const int32_t* map, int64_t n, void* stream);
/* tf.train.AdamOptimizer update of n parameters with g' = grad_scale*g + l2*w (slim l2_regularizer). */
int fn2_adam_step(float* w, float* m, float* v, const float* g, int64_t n, float lr, float beta1, float beta2,
                  float eps, int step, float l2, float grad_scale, void* stream);
/* The same update (tf.train.AdamOptimizer.apply_gradients, net.py:1290-1295) for n_tensors parameter tensors in
 * ONE launch.  table: device array of six 64-bit words per tensor: {float* w, float* m, float* v, const float* g,
 * void* w_f16x2, float scale (low 32 bits) | int frag_k (high 32 bits: 0 = row-major copy, else the packed row length k of a
 * wgt_layout-2 copy)} -- w_f16x2: the split-fp16 copy of w the convolutions read (same packed
 * geometry), rewritten as w * scale by the same pass, or NULL; every pointer 16-byte aligned.  counts / l2: device
 * arrays of element counts (int64) and L2 coefficients (0 where the reference does not regularise). */
int fn2_adam_step_multi(const void* table, const int64_t* counts, const float* l2, int n_tensors, float lr, float beta1,
                        float beta2, float eps, int step, float grad_scale, void* stream);
/* fn2_adam_step_multi with the per-step scalars in device memory: hyper = {lr_t, beta1, beta2, eps, grad_scale} with
 * lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t) (tf.train.AdamOptimizer's form, net.py:1290-1295).  The launch
 * arguments are then the same every step, which lets the train step be captured once and replayed as a hipGraph. */
int fn2_adam_step_multi_dev(const void* table, const int64_t* counts, const float* l2, int n_tensors, const float* hyper,
                            void* stream);
/* upsample_flowXtoY backward: g = gradient view [n,2h,2w,2] of its output slice, pf its fp32 input [n,h,w,2],
 * w [4][4][2][2]; dpf (+)= input gradient, dw += filter gradient. */
int fn2_upsample_flow_bwd(const fn2_tensor* g, const float* pf, const float* w, float* dpf, float* dw,
                          int accumulate, void* stream);
/* flow head (3x3 s1 p1, 2 outputs) filter gradient into rows 0,1 of its natural-order packed weight. */
int fn2_head_bwd_filter(const fn2_tensor* x, const float* g, float* dw, int cin_pad, int kpad, void* stream);
/* flow head input gradient, ADDED into the view dx: dx[pix][ci] += sum_{tap,co} g[pix-(tap-1)][co] * w[co][tap][ci]. */
int fn2_head_bwd_data(const float* g, const float* w, const fn2_tensor* dx, int cin_pad, int kpad, void* stream);

/* Filter gradient of one conv layer on the matrix cores (fp32 tensors: fp32 MFMA; split-fp16 tensors: 3 fp16 MFMAs per
 * product), ADDED into `dw` (fp32, atomics), which has the layout of the layer's packed forward weight
 * (fn2_conv_desc.wgt: same cin_pad / cout_pad / kpad / wgt_layout).
 *   kind 0: conv            dW[co][tap][ci] += sum x[pix*s + tap - pad][ci] * dy[pix][co]
 *   kind 1: deconv k4 s2    dWt (4-phase packing) += sum x[pix][ci] * dy[2 pix + tap - 1][co]
 *   kind 2: stem row-run conv (x = the pre-padded buffer, pad = 0, cin_pad = run length)
 *   kind 4: flow head from fn2_head_g18 (dy = the 18-channel tensor, kh = kw = 3, wgt_layout = 0, cout_pad >= 18):
 *           dW[o][tap * cin_pad + ci] += sum x[pix][ci] * G18[pix][tap * 2 + o] */
typedef struct {
  fn2_tensor x;        /* layer input (fp32 or split fp16; dy has the same dtype) */
  fn2_tensor dy;       /* gradient wrt the layer's pre-activation output */
  float* dw;
  int32_t kind, kh, kw, stride, pad;
  int32_t cin_pad, cout_pad, kpad, wgt_layout;
  float* db;           /* kinds 0 / 2: db[co] += sum over pixels of dy[.., co] in the same pass (the layer's bias gradient), or NULL */
} fn2_bwdw_desc;
int fn2_conv2d_bwd_filter(const fn2_bwdw_desc* d, void* stream);
/* Flow heads (predict_flowN: 3x3, pad 1, 2 outputs) through the matrix-core kernels: fn2_head_g18 writes the head's
 * output gradient g [n, h, w, 2] (fp32) as the 18-channel split-fp16 tensor G18[pix][tap * 2 + o] = g[pix shifted by the
 * tap][o] (zero outside the image) into `out` (c == 18, at least 24 channels of room; channels >= 24 of a wider buffer
 * must be zero and stay untouched).  Then fn2_conv2d_bwd_filter with kind 4 (x = head input, dy = G18, dw = the head's
 * natural-order weight gradient [2][kpad], cin_pad, kpad of the head) is the filter gradient, and a 1x1 fn2_conv2d from
 * G18 with accumulate the input gradient.  (fn2_head_bwd_filter / fn2_head_bwd_data are the fp32-tensor forms.) */
int fn2_head_g18(const float* g, const fn2_tensor* out, void* stream);

/* ---------------------------------------------------------------- training-input augmentation
 * The reference's preprocessing plugin (src/ops/preprocessing/preprocessing.cc:24-95).  The random coefficients
 * and their 2x3 matrices are made on the host (as the op does, HostMemory transforms); these are the per-pixel
 * passes.  All pointers are device pointers; transforms / chromatic are [n][6] fp32. */
/* DataAugmentation (kernels/data_augmentation.cc:30-150): out[n,y,x,:] = bilinear(src[n], T_n(x,y)), position
 * clamped to [0, size-1.05]; chromatic != NULL (c == 3): colour gains, brightness compensation, gamma,
 * brightness, contrast, clamp to [0,1]; chromatic[n] = (gamma, brightness, contrast, color1..3). */
int fn2_augment_f32(const float* src, const float* transforms, const float* chromatic, float* out, int n, int src_h,
                    int src_w, int c, int out_h, int out_w, void* stream);
/* FlowAugmentation (kernels/flow_augmentation.cc:19-66): out = T_b^-1(T_a(x,y) + flow[round(T_a(x,y))]) - (x,y). */
int fn2_flow_augmentation_f32(const float* flows, const float* transforms_from_a, const float* inv_transforms_from_b,
                              float* out, int n, int src_h, int src_w, int out_h, int out_w, void* stream);

/* ---------------------------------------------------------------- launch-graph helpers (hipGraph) */
int fn2_capture_begin(void* stream);
int fn2_capture_end(void* stream, void** graph_exec);
int fn2_graph_launch(void* graph_exec, void* stream);
int fn2_graph_destroy(void* graph_exec);

#ifdef __cplusplus
}
#endif
#endif /* FLOWNET2_HIP_H_ */
