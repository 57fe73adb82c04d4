/*
 * ganlab_hip.h - C ABI of the MI355X (gfx950) kernel library for the gan-lab G+D training step.
 *
 * The reference (sidward14/gan-lab v0.4.2) is pure Python/PyTorch: it has no FFI, plugin or
 * operator interface of its own.  The boundary this library replaces is therefore
 * the set of ATen call sites inside the reference's layer ops; each entry point below cites the
 * reference file:line (relative to /root/reference/gan_lab) whose math it implements.  A
 * maintainer binds these with ctypes from `torch.autograd.Function`s - see INTEGRATION.md and
 * gan_lab_amd/_lib.py.
 *
 * Conventions
 *  - all tensors are fp32, contiguous NCHW (weights OIHW), device pointers owned by the caller;
 *  - nothing is allocated inside: scratch is a caller-provided workspace;
 *  - every call is asynchronous on `stream` (a hipStream_t passed as void*);
 *  - return value: 0 on success, negative GANLAB_E* on error (never throws across the ABI);
 *  - re-entrant: no global mutable state.
 */
#ifndef GANLAB_HIP_H
#define GANLAB_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GANLAB_OK 0
#define GANLAB_EINVAL (-1)   /* bad argument (null pointer, non-positive dim, unsupported ks) */
#define GANLAB_EWORKSPACE (-2) /* workspace too small */
#define GANLAB_ELAUNCH (-3)  /* hipGetLastError() != hipSuccess after the launch */
#define GANLAB_EUNSUPPORTED (-4) /* geometry outside what the fused kernel handles (see *_supported) */

#define GANLAB_ACT_NONE 0
#define GANLAB_ACT_LRELU 1

#define GANLAB_PACK_FWD 0    /* out[tap][ci][co] = scale * w[co][ci][tap]                */
#define GANLAB_PACK_DGRAD 1  /* out[tap][co][ci] = scale * w[co][ci][KK-1-tap] (flipped) */

int ganlab_abi_version(void);

/* Diagnostic (no reference counterpart): demangled symbol and workgroup count of the calling thread's most recent
 * kernel launch made by this library.  Returns the symbol's length (it is truncated to cap-1 characters), 0 when
 * nothing was launched yet.  bench.py uses it to name the kernel its `roofline` prices from what was dispatched. */
int ganlab_last_launch(char* name, int cap, unsigned* grid);
/* Diagnostics for tests that assert WHICH kernels a step dispatched (an entry point may launch several: a weight
 * gradient and its slot reduction): the number of kernels the calling thread has launched through this library so far,
 * and the launch `back` positions before its most recent one (0 = what ganlab_last_launch reports; the library keeps
 * the last 16; returns 0 beyond that). */
unsigned long long ganlab_launch_count(void);
int ganlab_launch_history(int back, char* name, int cap, unsigned* grid);

/* Geometry of one convolution C(x, w) = scale * conv2d(up2?(x), w, stride 1, padding pad).
 * Replaces Conv2dEx.forward / LinearEx.forward (utils/custom_layers.py:202-211, :282-291) with the
 * eq-LR runtime scale folded into `scale` (the reference multiplies the input, :204), and the
 * nearest-neighbour 2x upsample in front of a generator conv
 * (stylegan/architectures.py:292-334, progan/architectures.py:119-148) folded in as `up`. */
typedef struct {
  int N, Cin, Hin, Win; /* physical input (before the optional 2x nearest upsample)     */
  int Cout, ks, pad;    /* square kernel ks in {1,3}; a 4x4 valid conv is run as a linear */
  int up;               /* 1: input is nearest-upsampled 2x on the fly                   */
  int pool;             /* 1: output is 2x2 average-pooled (ganlab_conv_s2_* entry points only) */
} ganlab_conv_geom;

/* sizeof(ganlab_conv_geom) as this library was compiled: a binding asserts its mirror struct against it once. */
int ganlab_conv_geom_size(void);

/* Output spatial size of a geometry: Hout = (up ? 2*Hin : Hin) + 2*pad - ks + 1. */
int ganlab_conv_out_hw(const ganlab_conv_geom* g, int* Hout, int* Wout);

/* Re-layout OIHW weights for the implicit-GEMM kernels; `mode` is GANLAB_PACK_*.
 * Returns the number of floats the packed buffer needs when `out` is NULL. */
long long ganlab_conv_pack_f32(const float* w, float* out, int Cout, int Cin, int ks, int mode,
                               float scale, void* stream);

/* y = act( conv(x, wp) + bias * bias_scale ), fp32 MFMA implicit GEMM.
 * wp: GANLAB_PACK_FWD-packed weights (scale already folded in).  bias may be NULL.
 * custom_layers.py:202-211 (+ Conv2dBias :222-226, LeakyReLU) */
int ganlab_conv_fwd_f32(const float* x, const float* wp, const float* bias, float* y,
                        const ganlab_conv_geom* g, float bias_scale, int act, float slope,
                        void* stream);

/* gx = conv_transpose(gy, w) wrt the conv input (the autograd rule of custom_layers.py:204-206).
 * wp: GANLAB_PACK_DGRAD-packed weights.  When g->up == 1 the result is the gradient w.r.t. the
 * *virtual* (upsampled) input, shape (N, Cin, 2*Hin, 2*Win); fold it with ganlab_pool2_f32. */
int ganlab_conv_dgrad_f32(const float* gy, const float* wp, float* gx_virtual,
                          const ganlab_conv_geom* g, void* stream);

/* Split-K variants for the small, channel-heavy layers (512-channel 4x4 / 8x8 maps; linear layers at small batch), whose
 * plain launch is a few dozen workgroups walking a long contraction: ganlab_conv_splitk_plan returns S (0 / 1: use the plain
 * entry point); S >= 2: S workgroups share one output tile, raw partial sums go to the workspace (S * output elements
 * floats) and a finishing kernel adds them in a fixed order, + bias, + activation.  No upsample. */
int ganlab_conv_splitk_plan(const ganlab_conv_geom* g, int dgrad);
int ganlab_conv_fwd_splitk_f32(const float* x, const float* wp, const float* bias, float* y, const ganlab_conv_geom* g,
                               float bias_scale, int act, float slope, void* workspace, size_t workspace_bytes,
                               void* stream);
int ganlab_conv_dgrad_splitk_f32(const float* gy, const float* wp, float* gx, const ganlab_conv_geom* g, void* workspace,
                                 size_t workspace_bytes, void* stream);

/* gx = dgrad(gy, w) * lrelu'(x) where x, the conv's input (shaped like gx), is the LeakyReLU output of the layer in
 * front (D block k's pooled conv -> block k+1's first conv, progan/architectures.py:280-293): that layer's backward then
 * takes gx as the gradient of its PRE-activation and skips its own gy * lrelu'(y) pass.  3x3 pad-1 convs, no
 * upsample, rows of >= 16 4-aligned pixels (*_supported). */
int ganlab_conv_dgrad_mask_supported(const ganlab_conv_geom* g);
int ganlab_conv_dgrad_mask_f32(const float* gy, const float* wp, const float* x, float* gx, const ganlab_conv_geom* g,
                               float slope, void* stream);

/* fromRGB (1x1 conv from <= 3 channels + bias + LeakyReLU, progan/architectures.py:232-237) at >= 64x64 runs on
 * HBM-streaming kernels that can fold the activation's backward into the conv's own gradient kernels, saving the
 * separate  gz = gy * lrelu'(y)  pass over the widest tensor of the discriminator:
 *   dgrad_act : gx = dgrad(gy * lrelu'(y), w)             fwd_mask : out = conv(x, w) * lrelu'(y)  (its adjoint)
 *   wgrad_act : gw = scale * wgrad(gy * lrelu'(y), x) and gb = bias_scale * sum(gy * lrelu'(y))
 * y is the conv's activated OUTPUT.  *_supported() == 0 -> GANLAB_EUNSUPPORTED, compose the plain entry points. */
int ganlab_conv_act_bwd_fused_supported(const ganlab_conv_geom* g);
int ganlab_conv_dgrad_act_f32(const float* gy, const float* y, const float* wp, float* gx, const ganlab_conv_geom* g,
                              float slope, void* stream);
int ganlab_conv_fwd_mask_f32(const float* x, const float* wp, const float* y, float* out, const ganlab_conv_geom* g,
                             float slope, void* stream);
int ganlab_conv_wgrad_act_f32(const float* gy, const float* y, const float* x, float* gw, float* gb,
                              const ganlab_conv_geom* g, float scale, float bias_scale, float slope, void* workspace,
                              size_t workspace_bytes, void* stream);

/* gw[co][ci][ky][kx] = scale * sum_{n,y,x} gy[n,co,y,x] * up2?(x)[n,ci,y+ky-pad,x+kx-pad].
 * Deterministic two-stage reduction through `workspace`; query the size with
 * ganlab_conv_wgrad_workspace (bytes). */
size_t ganlab_conv_wgrad_workspace(const ganlab_conv_geom* g);
int ganlab_conv_wgrad_f32(const float* gy, const float* x, float* gw, const ganlab_conv_geom* g,
                          float scale, void* workspace, size_t workspace_bytes, void* stream);

/* ---- stride-2 fused layers (csrc/conv_s2.hip) -------------------------------------------------------
 * D "down" layer AvgPool2d(2)(conv3x3(x)) (progan/architectures.py:261-284; geom.pool = 1) and G "up"
 * layer conv3x3(Upsample2x(x)) (stylegan/architectures.py:292-334; geom.up = 1) collapse exactly to a
 * 4x4 stride-2 kernel K4 = M W M^T: 16 instead of 36 MACs per low-res pixel, no full-resolution
 * intermediate.  Same calling convention as the plain entry points; `geom` describes the ORIGINAL 3x3
 * layer (ks = 3, pad = 1, exactly one of up / pool set).  ganlab_conv_s2_supported() tells whether the
 * shape qualifies (even sizes, low-res width >= 16 and a multiple of 4); otherwise compose the plain
 * kernels.  pack: `up` selects the combination matrix, `transpose` = 1 packs for the operator that
 * consumes the output gradient (dgrad). */
int ganlab_conv_s2_supported(const ganlab_conv_geom* g);
long long ganlab_conv_s2_pack_f32(const float* w, float* out, int Cout, int Cin, int up, int transpose,
                                  float scale, void* stream);
int ganlab_conv_s2_fwd_f32(const float* x, const float* wp, const float* bias, float* y,
                           const ganlab_conv_geom* g, float bias_scale, int act, float slope, void* stream);
int ganlab_conv_s2_dgrad_f32(const float* gy, const float* wp, float* gx, const ganlab_conv_geom* g,
                             void* stream);
size_t ganlab_conv_s2_wgrad_workspace(const ganlab_conv_geom* g);
int ganlab_conv_s2_wgrad_f32(const float* gy, const float* x, float* gw, const ganlab_conv_geom* g,
                             float scale, void* workspace, size_t workspace_bytes, void* stream);

/* ---- bf16-compute convolutions (csrc/conv_bf16.hip; BASELINE config #2 "bf16 compute / fp32 master") ---------
 * Same math and call sites as the fp32 entry points above (custom_layers.py:202-211 and its autograd rules), but
 * the operands are rounded to bf16 (RNE) on their way into LDS and multiplied on v_mfma_f32_16x16x32_bf16 with fp32
 * accumulation; every tensor in HBM stays fp32 (the reference's storage type, fp32 master weights).  The reference
 * itself is fp32 only, so this path is an extension selected by the caller (gan_lab_amd.ops.compute_dtype).
 * Supported (ganlab_conv_bf16_supported; forward and input gradient): 3x3, pad 1, Cin % 64 == 0, Cout % 64 == 0 and, at
 * the resolution the taps run on (2 Hin x 2 Win with up), either H % 8 == 0, W % 32 == 0 or H % 16 == 0, W % 16 == 0;
 * up (nearest 2x upsample in front, progan/architectures.py:160-186) or pool (2x2 average pool behind, :261-284) - not
 * both - fold into the kernels: x / gx are Hin x Win, y / gy are Ho x Wo as for ganlab_conv_s2_*; everything else stays
 * on the exact fp32 kernels.  The weight gradient needs W % 32 == 0, H % 8 == 0 at the taps' resolution (with up it reads
 * the Hin x Win input in place, with pool the pooled gy): ganlab_conv_wgrad_bf16_workspace returns 0 for a geometry it
 * does not take (ganlab_conv_wgrad_bf16 then returns GANLAB_EUNSUPPORTED) and the caller uses ganlab_conv_wgrad_f32 on
 * the plain geometry with the materialised upsample of x, resp. of gy / 4.
 * pack: returns the number of bf16 elements (9*Cout*Cin) when `out` is NULL; mode is GANLAB_PACK_*. */
int ganlab_conv_bf16_supported(const ganlab_conv_geom* g);
long long ganlab_conv_pack_bf16(const float* w, void* out, int Cout, int Cin, int mode, float scale, void* stream);
int ganlab_conv_fwd_bf16(const float* x, const void* wp, const float* bias, float* y, const ganlab_conv_geom* g,
                         float bias_scale, int act, float slope, void* stream);
int ganlab_conv_dgrad_bf16(const float* gy, const void* wp, float* gx, const ganlab_conv_geom* g, void* stream);
/* split-K forms of the two for layers with few output tiles (the 512-channel 16x16 layers at batch 8: 64 workgroups on 256
 * CUs): ganlab_conv_bf16_splitk_plan = number of K-splits S (1: nothing to split); workspace: S * N * C * H * W floats at the
 * taps' resolution (C: the operator's output channels), raw partial sums + a fixed-order finish (pool / scale / bias / act) */
int ganlab_conv_bf16_splitk_plan(const ganlab_conv_geom* g, int dgrad);
int ganlab_conv_fwd_bf16_splitk(const float* x, const void* wp, const float* bias, float* y, const ganlab_conv_geom* g,
                                float bias_scale, int act, float slope, void* workspace, size_t workspace_bytes,
                                void* stream);
int ganlab_conv_dgrad_bf16_splitk(const float* gy, const void* wp, float* gx, const ganlab_conv_geom* g, void* workspace,
                                  size_t workspace_bytes, void* stream);
size_t ganlab_conv_wgrad_bf16_workspace(const ganlab_conv_geom* g);
int ganlab_conv_wgrad_bf16(const float* gy, const float* x, float* gw, const ganlab_conv_geom* g, float scale,
                           void* workspace, size_t workspace_bytes, void* stream);

/* ---- fp32 convolutions as split products on the bf16 matrix cores (csrc/conv_x3.hip) ---------------------------------
 * Same math and call sites as ganlab_conv_fwd_f32 / ganlab_conv_dgrad_f32 (custom_layers.py:202-211), fp32 in, fp32 out:
 * every operand is cut into three bf16 planes that sum to it EXACTLY, a product is six v_mfma_f32_16x16x32_bf16 products
 * (the three dropped cross terms are <= 2^-26 of it) and the sums are kept in fp32 chains of one 32-channel chunk each,
 * so the result is as close to float64 as the exact-fp32 kernels' (tools/op_error_probe.py) at 16/6 of their MFMA rate.
 * Supported (ganlab_conv_x3_supported): 3x3, pad 1, no up / pool, H % 16 == 0, W % 16 == 0, contraction channels % 64 == 0,
 * output channels % 64 == 0.  pack: OIHW -> three-plane k-step images, returns the number of bf16 elements
 * (27*Cout*Cin) when `out` is NULL; mode is GANLAB_PACK_*. */
int ganlab_conv_x3_supported(const ganlab_conv_geom* g, int dgrad);
long long ganlab_conv_x3_pack(const float* w, void* out, int Cout, int Cin, int mode, float scale, void* stream);
int ganlab_conv_fwd_x3(const float* x, const void* wp, const float* bias, float* y, const ganlab_conv_geom* g,
                       float bias_scale, int act, float slope, void* stream);
int ganlab_conv_dgrad_x3(const float* gy, const void* wp, float* gx, const ganlab_conv_geom* g, void* stream);
/* the forms of the exact-fp32 entry points the thick layers use, same arguments except the packed weights:
 * ganlab_conv_dgrad_mask_f32, ganlab_conv_fwd_aff_f32, ganlab_conv_fwd_aff_tail_f32 (+ its tiles-per-plane query) */
int ganlab_conv_dgrad_mask_x3(const float* gy, const void* wp, const float* x, float* gx, const ganlab_conv_geom* g,
                              float slope, void* stream);
int ganlab_conv_fwd_aff_x3(const float* x, const void* wp, const float* aff_s, const float* aff_t, const float* bias,
                           float* y, const ganlab_conv_geom* g, float bias_scale, int act, float slope, void* stream);
int ganlab_conv_fwd_aff_tail_x3_chunks(const ganlab_conv_geom* g);
/* the stride-2 fused layers' transposed form (the exact-fp32 T kernel of csrc/conv_s2.hip): ganlab_conv_s2_fwd_f32 /
 * ganlab_conv_s2_fwd_aff_f32 of an up layer (geom.up = 1; dgrad = 0) and ganlab_conv_s2_dgrad_f32 of a pooled layer
 * (geom.pool = 1; dgrad = 1), low resolution H % 8 == 0, W % 16 == 0, contraction channels % 64 == 0, output channels
 * % 32 == 0 (not a multiple of 64: the 32-channel form - both row parities per workgroup, its own packed layout).  pack: `up` = 1 packs an up
 * layer's forward weights, 0 a pooled layer's input-gradient weights (48*Cout*Cin bf16 elements; in a ganlab_pack_desc:
 * kind GANLAB_PACKKIND_X3 with ks = 4 and the same `up`). */
int ganlab_conv_s2_x3_supported(const ganlab_conv_geom* g, int dgrad);
long long ganlab_conv_s2_x3_pack(const float* w, void* out, int Cout, int Cin, int up, float scale, void* stream);
int ganlab_conv_s2_fwd_x3(const float* x, const void* wp, const float* bias, float* y, const ganlab_conv_geom* g,
                          float bias_scale, int act, float slope, void* stream);
int ganlab_conv_s2_fwd_aff_x3(const float* x, const void* wp, const float* aff_s, const float* aff_t, const float* bias,
                              float* y, const ganlab_conv_geom* g, float bias_scale, int act, float slope, void* stream);
int ganlab_conv_s2_dgrad_x3(const float* gy, const void* wp, float* gx, const ganlab_conv_geom* g, void* stream);
/* ... and their strided form (csrc/conv_x3_down.hip; the exact-fp32 S kernel of csrc/conv_s2.hip): ganlab_conv_s2_fwd_f32 of a
 * pooled layer (geom.pool = 1; dgrad = 0) and ganlab_conv_s2_dgrad_f32 of an up layer (geom.up = 1; dgrad = 1); low resolution
 * H % 8 == 0, W % 16 == 0, contraction channels % 16 == 0, output channels % 128 == 0.  pack: `up` = 0 a pooled layer's forward
 * weights, 1 an up layer's input-gradient weights (48*Cout*Cin bf16 elements; ganlab_pack_desc: kind X3, ks = 5, same `up`). */
int ganlab_conv_s2_down_x3_supported(const ganlab_conv_geom* g, int dgrad);
long long ganlab_conv_s2_down_x3_pack(const float* w, void* out, int Cout, int Cin, int up, float scale, void* stream);
int ganlab_conv_s2_down_fwd_x3(const float* x, const void* wp, const float* bias, float* y, const ganlab_conv_geom* g,
                               float bias_scale, int act, float slope, void* stream);
int ganlab_conv_s2_down_dgrad_x3(const float* gy, const void* wp, float* gx, const ganlab_conv_geom* g, void* stream);
/* weight gradient of a plain 3x3 layer (csrc/conv_x3_wgrad.hip): ganlab_conv_wgrad_f32, or with aff_s / aff_t non-null
 * ganlab_conv_wgrad_aff_f32, as split products; H a power of two, W % 32 == 0, Cin % 64 == 0, Cout % 32 == 0.  Deterministic
 * (per-workgroup slots in the workspace, fixed-order reduction). */
int ganlab_conv_wgrad_x3_supported(const ganlab_conv_geom* g);
size_t ganlab_conv_wgrad_x3_workspace(const ganlab_conv_geom* g);
int ganlab_conv_wgrad_x3(const float* gy, const float* x, const float* aff_s, const float* aff_t, float* gw,
                         const ganlab_conv_geom* g, float scale, void* workspace, size_t workspace_bytes, void* stream);
/* weight gradient of a stride-2 3x3 layer (csrc/conv_x3_s2_wgrad.hip; geom.pool = 1: avgpool2(conv3(x)), geom.up = 1:
 * conv3(up2(x))): ganlab_conv_s2_wgrad_f32, or with aff_s / aff_t non-null (up layers) ganlab_conv_s2_wgrad_aff_f32, as split
 * products over the 2x2 box sums of the high-resolution operand; low resolution H a power of two >= 4, W % 32 == 0 or W == 16, channels of
 * the low-resolution operand (pool: Cout, up: Cin) % 64 == 0, of the high-resolution one % 32 == 0.  Deterministic. */
int ganlab_conv_s2_wgrad_x3_supported(const ganlab_conv_geom* g);
size_t ganlab_conv_s2_wgrad_x3_workspace(const ganlab_conv_geom* g);
int ganlab_conv_s2_wgrad_x3(const float* gy, const float* x, const float* aff_s, const float* aff_t, float* gw,
                            const ganlab_conv_geom* g, float scale, void* workspace, size_t workspace_bytes, void* stream);
int ganlab_conv_fwd_aff_tail_x3(const float* x, const void* wp, const float* aff_s, const float* aff_t, const float* bias,
                                const float* noise, const float* noise_w, float* y, float* mean, float* rstd,
                                const ganlab_conv_geom* g, float bias_scale, int act, float slope, float eps, void* workspace,
                                size_t workspace_bytes, void* stream);

/* ---- depthwise / resampling (custom_layers.py:36-53; nn.Upsample / nn.AvgPool2d call sites) ---- */
/* y = depthwise [1 2 1]x[1 2 1]/16 blur, zero padding (self-adjoint: also its own backward). */
int ganlab_blur3x3_f32(const float* x, float* y, long long planes, int H, int W, void* stream);
/* y[2h+i,2w+j] = scale * x[h,w]  (nearest upsample; backward of pool2) */
int ganlab_up2_f32(const float* x, float* y, long long planes, int H, int W, float scale, void* stream);
/* y[h,w] = scale * sum_{i,j<2} x[2h+i,2w+j]  (avg-pool with scale .25; backward of up2) */
int ganlab_pool2_f32(const float* x, float* y, long long planes, int Hout, int Wout, float scale,
                     void* stream);
/* The other resamplers of custom_layers.py:59-75 (NearestPool2d, BilinearPool2d) and of nn.Upsample(mode='bilinear')
 * (resnetgan/learner.py:147-170), forward and adjoint: a separable table-driven gather
 *   y[p,oy,ox] = sum_{a<Ty} sum_{b<Tx} wy[oy*Ty+a] * wx[ox*Tx+b] * x[p, iy[oy*Ty+a], ix[ox*Tx+b]]
 * with int32 source indices / float weights per output row and column (1 <= Ty, Tx <= 6; pad with weight 0). */
int ganlab_resample2d_f32(const float* x, float* y, const int* iy, const float* wy, const int* ix, const float* wx,
                          long long planes, int Hi, int Wi, int Ho, int Wo, int Ty, int Tx, void* stream);

/* ---- bias / activation / noise (custom_layers.py:213-226, stylegan/architectures.py:105-119) ---- */
/* y = act(x + noise_w[c]*noise[n,hw] + bias[c]*bias_scale); noise/noise_w and bias may be NULL. */
int ganlab_bias_act_f32(const float* x, const float* bias, const float* noise, const float* noise_w,
                        float* y, int N, int C, long long HW, float bias_scale, int act, float slope,
                        void* stream);
/* gz = gy * (y > 0 ? 1 : slope)   (LeakyReLU backward from the saved OUTPUT) */
int ganlab_act_bwd_f32(const float* gy, const float* y, float* gz, long long n, float slope, void* stream);
/* gz = gy * (y > 0 ? 1 : slope) and gb[c] = scale * sum_{n,hw} gz[n,c,hw] in ONE pass (the backward of
 * Conv2dBias + LeakyReLU, custom_layers.py:222-226); workspace as ganlab_channel_sum_workspace */
int ganlab_act_bwd_bias_f32(const float* gy, const float* y, float* gz, float* gb, int N, int C, long long HW,
                            float slope, float scale, void* workspace, size_t workspace_bytes, void* stream);
/* ---- blur fused with its pointwise neighbours (H even, W % 4 == 0: ganlab_blur_fused_supported) --------
 * The binomial blur of a G layer sits between the up-conv and noise/bias/LeakyReLU
 * (stylegan/architectures.py:331-360 -> :105-119, progan/architectures.py:95-107), that of a D block
 * between LeakyReLU and the down-conv (progan/architectures.py:280-293): one pass instead of two.
 *   blur_bias_act:  y   = act(blur(x) + noise_w[c]*noise[n,hw] + bias[c]*bias_scale)
 *   blur_act_bwd:   out = lrelu'(y) * blur(g),  gb[c] = bias_scale * sum out        (gb nullable)
 *   act_bwd_blur:   out = blur(lrelu'(y) * g),  gb[c] = bias_scale * sum lrelu'(y)*g,
 *                   gnw[c] = sum lrelu'(y)*g*noise[n,hw]                             (gb, gnw nullable)
 * blur_act_bwd and act_bwd_blur are adjoint: each is the other's backward (R1 double backward). */
int ganlab_blur_fused_supported(int H, int W);
size_t ganlab_blur_fused_workspace(int N, int C, int H, int W);
int ganlab_blur_bias_act_f32(const float* x, const float* bias, const float* noise, const float* noise_w, float* y,
                             int N, int C, int H, int W, float bias_scale, int act, float slope, void* stream);
int ganlab_blur_act_bwd_f32(const float* g, const float* y, float* out, float* gb, int N, int C, int H, int W,
                            float slope, float bias_scale, void* workspace, size_t workspace_bytes, void* stream);
int ganlab_act_bwd_blur_f32(const float* g, const float* y, const float* noise, float* out, float* gb, float* gnw,
                            int N, int C, int H, int W, float slope, float bias_scale, void* workspace,
                            size_t workspace_bytes, void* stream);
/* The same two forward ops, also producing the InstanceNorm statistics of their OUTPUT in the same pass (the generator
 * layer is conv -> [blur] -> noise + bias + LeakyReLU -> InstanceNorm + style, stylegan/architectures.py:497-526: the
 * statistics pass of ganlab_instnorm_stats_f32 over the tensor just written is saved).  mean / rstd: (N*C,), rstd =
 * 1/sqrt(biased var + eps); fp64 sums; workspace ganlab_act_stats_workspace bytes.  HW % 4 == 0. */
size_t ganlab_act_stats_workspace(int N, int C, long long HW);
int ganlab_bias_act_stats_f32(const float* x, const float* bias, const float* noise, const float* noise_w, float* y,
                              float* mean, float* rstd, int N, int C, long long HW, float bias_scale, int act,
                              float slope, float eps, void* workspace, size_t workspace_bytes, void* stream);
int ganlab_blur_bias_act_stats_f32(const float* x, const float* bias, const float* noise, const float* noise_w,
                                   float* y, float* mean, float* rstd, int N, int C, int H, int W, float bias_scale,
                                   int act, float slope, float eps, void* workspace, size_t workspace_bytes,
                                   void* stream);
/* ... and the matching backward: InstanceNorm+style backward fused with the LeakyReLU / bias / noise backward in front
 * of it (x = that layer's activated output = the InstanceNorm input).  gz = dL/d(pre-activation), gb (C,) =
 * bias_scale * sum gz or NULL, gnw (C,) = sum gz*noise or NULL; s1, s2 from ganlab_instnorm_style_bwd_reduce_f32. */
size_t ganlab_instnorm_bwd_act_workspace(int N, int C, long long HW);
int ganlab_instnorm_style_bwd_act_f32(const float* gy, const float* x, const float* mean, const float* rstd,
                                      const float* style, const float* s1, const float* s2, const float* noise,
                                      float* gz, float* gb, float* gnw, int N, int C, long long HW, int act,
                                      float slope, float bias_scale, void* workspace, size_t workspace_bytes,
                                      void* stream);
/* ... followed by the (self-adjoint) blur of a BLURRED generator layer in the same pass: out = blur(gz), gz is not written
 * (blur -> noise/bias/LeakyReLU -> InstanceNorm/style backward of stylegan/architectures.py:497-526 behind :331-360) */
int ganlab_instnorm_style_bwd_act_blur_f32(const float* gy, const float* x, const float* mean, const float* rstd,
                                           const float* style, const float* s1, const float* s2, const float* noise,
                                           float* out, float* gb, float* gnw, int N, int C, int H, int W, int act,
                                           float slope, float bias_scale, void* workspace, size_t workspace_bytes,
                                           void* stream);
/* The same two passes (ganlab_instnorm_style_bwd_reduce_f32, ganlab_instnorm_style_bwd_act_f32) for the generator's LAST
 * layer, whose only reader is toRGB (stylegan/architectures.py:170-172, :398-402 - a 1x1 conv to crgb <= 4 channels): the
 * incoming gradient  gy[n, c] = sum_k wp[k * round_up(C, 64) + c] * grgb[n, k]  is recomputed from the image gradient grgb
 * (N, crgb, HW) and toRGB's input-gradient pack wp (scale included) instead of being written by ganlab_conv_dgrad_f32 and
 * read back twice.  gz / gb / gnw carry the bits of that sequence given the same s1 / s2; s1 / s2 are fp64 sums in another
 * order (equal after rounding to fp32 up to ties).  C <= 16, HW % 4 == 0, HW >= 1024. */
int ganlab_instnorm_bwd_rgb_supported(int N, int C, int crgb, long long HW);
size_t ganlab_instnorm_bwd_reduce_rgb_workspace(int N, int C, long long HW);
int ganlab_instnorm_style_bwd_reduce_rgb_f32(const float* grgb, const float* wp, int crgb, const float* x,
                                             const float* mean, const float* rstd, float* s1, float* s2, int N, int C,
                                             long long HW, void* workspace, size_t workspace_bytes, void* stream);
int ganlab_instnorm_style_bwd_act_rgb_f32(const float* grgb, const float* wp, int crgb, const float* x, const float* mean,
                                          const float* rstd, const float* style, const float* s1, const float* s2,
                                          const float* noise, float* gz, float* gb, float* gnw, int N, int C, long long HW,
                                          int act, float slope, float bias_scale, void* workspace, size_t workspace_bytes,
                                          void* stream);
/* out[c] = scale * sum_{n,hw} a[n,c,hw] * (b ? b[n,hw] : 1)   (bias / noise-weight gradients) */
int ganlab_channel_sum_f32(const float* a, const float* b_n1hw, float* out, int N, int C, long long HW,
                           float scale, void* workspace, size_t workspace_bytes, void* stream);
size_t ganlab_channel_sum_workspace(int N, int C, long long HW);

/* ---- normalisation ---- */
/* InstanceNorm2d(eps, biased var) statistics per (n,c) plane: mean, rstd (custom_layers.py:98-99) */
int ganlab_instnorm_stats_f32(const float* x, float* mean, float* rstd, long long planes, long long HW,
                              float eps, void* stream);
/* the same statistics for few, long rows (LayerNorm([C,R,R]) of the ResNet critics, custom_layers.py:100-107: `rows` =
 * batch, M = C*R*R): several blocks per row, fp64 partial sums, fixed-order finish (deterministic). */
size_t ganlab_row_stats_workspace(long long rows, long long M);
int ganlab_row_stats_f32(const float* x, float* mean, float* rstd, long long rows, long long M, float eps,
                         void* workspace, size_t workspace_bytes, void* stream);
/* y = (x-mean)*rstd*(ys+1)+yb with style (N,2,C) (stylegan/architectures.py:524-526); style may be
 * NULL (plain InstanceNorm). */
int ganlab_instnorm_style_fwd_f32(const float* x, const float* mean, const float* rstd,
                                  const float* style, float* y, int N, int C, long long HW, void* stream);
/* s1[n,c] = sum gy ; s2[n,c] = sum gy * xhat */
int ganlab_instnorm_style_bwd_reduce_f32(const float* gy, const float* x, const float* mean,
                                         const float* rstd, float* s1, float* s2, long long planes,
                                         long long HW, void* stream);
/* gx = rstd*(ys+1)*(gy - s1/HW - xhat*s2/HW) */
int ganlab_instnorm_style_bwd_apply_f32(const float* gy, const float* x, const float* mean,
                                        const float* rstd, const float* style, const float* s1,
                                        const float* s2, float* gx, int N, int C, long long HW,
                                        void* stream);
/* PixelNorm over the channel dim (custom_layers.py:81-86): y = x*rsqrt(mean_c x^2 + eps) */
int ganlab_pixelnorm_fwd_f32(const float* x, float* y, int N, int C, long long HW, float eps, void* stream);
int ganlab_pixelnorm_bwd_f32(const float* gy, const float* x, float* gx, int N, int C, long long HW,
                             float eps, void* stream);

/* building blocks of BatchNorm2d / LayerNorm (ResNet GAN path, custom_layers.py:100-107): the
 * normalisations are composed from y = x*scale[c] + shift[c], elementwise products and channel sums,
 * each closed under differentiation, so the WGAN-GP double backward works through LayerNorm */
int ganlab_chan_affine_f32(const float* x, const float* scale, const float* shift, float* y, int N, int C,
                           long long HW, void* stream);
int ganlab_mul_f32(const float* a, const float* b, float* out, long long n, void* stream);
/* ---- fused LayerNorm([C,R,R]) of the ResNet critics, first and second order (csrc/norm.hip) ----------------
 * resnetgan/resblocks.py:15-121 -> NormalizeLayer('LayerNorm') = nn.LayerNorm (custom_layers.py:100-107).  A sample
 * is a row of M = C*R*R elements; mean / rstd come from ganlab_instnorm_stats_f32(planes = N, HW = M);
 * P_x(g) = rstd * (g - mean(g) - xhat * mean(g * xhat)) per row.
 *   ln_affine_fwd:   y = act((x - mean[n]) * rstd[n] * w[m] + b[m])                     (w, b nullable; act = GANLAB_ACT_LRELU
 *                    folds in the LeakyReLU / ReLU that follows the normalisation in every residual block,
 *                    resnetgan/resblocks.py:48-49)
 *   colscale:        out[n,m] = a[n,m] * w[m]                                           (ghat = gy * w)
 *   coldot:          o1[m] = sum_n a * f,  f = (x - mean[n]) * rstd[n] (mean given) or x;  o2[m] = sum_n a (nullable)
 *                    -> gw, gb of the backward; d/dw of the double backward
 *   ln_rowsums:      per row n, several blocks per row + a fixed-order finish (deterministic), out = [N][3]:
 *                    t = a*(wa ? wa[m] : 1):  s0 = sum t,  s1 = sum t*xhat,  s2 = sum a*b2*(w2 ? w2[m] : 1) (b2 nullable);
 *                    with yact (the fused layer's OUTPUT) the backward of that LeakyReLU is applied on load,
 *                    a <- a * lrelu'(yact), and the masked gradient is also stored to gz for the passes that follow
 *   ln_project:      out = (wo ? wo[m] : 1) * rstd[n] * (a*(wa ? wa[m] : 1) - s0/M - xhat*s1/M)   (= wo * P_x(a*wa))
 *   ln_bwdbwd_apply: out = c1[n] * xhat + c2[n] * pu + c3[n] * gx    (d/dx of the double backward); the row coefficients
 *                    are formed in the kernel from sums = ln_rowsums(gy, w) and usums = ln_rowsums(u, b2 = gy, w2 = w):
 *                    c1 = -rstd^2 (u2 - s0 u0 - s1 u1), c2 = -rstd s1, c3 = -rstd u1, every sum divided by M */
int ganlab_ln_affine_fwd_f32(const float* x, const float* mean, const float* rstd, const float* w, const float* b,
                             float* y, int N, long long M, int act, float slope, void* stream);
int ganlab_colscale_f32(const float* a, const float* w, float* out, int N, long long M, void* stream);
int ganlab_coldot_f32(const float* a, const float* x, const float* mean, const float* rstd, float* o1, float* o2,
                      int N, long long M, void* stream);
size_t ganlab_ln_rowsums_workspace(int N, long long M);
int ganlab_ln_rowsums_f32(const float* a, const float* wa, const float* x, const float* mean, const float* rstd,
                          const float* b2, const float* w2, float* out, int N, long long M, void* workspace,
                          size_t workspace_bytes, const float* yact, float* gz, float slope, void* stream);
int ganlab_ln_project_f32(const float* a, const float* wa, const float* x, const float* mean, const float* rstd,
                          const float* sums, const float* wo, float* out, int N, long long M, void* stream);
/* ln_bwd_cols: the first-order backward's projection AND parameter gradients in one pass over (a, x), thread per column:
 * gx = rstd[n] * (a*w[m] - s0[n]/M - xhat*s1[n]/M) (= ln_project with wa = w), gw[m] = sum_n a*xhat, gb[m] = sum_n a (= coldot) */
int ganlab_ln_bwd_cols_f32(const float* a, const float* w, const float* x, const float* mean, const float* rstd,
                           const float* sums, float* gx, float* gw, float* gb, int N, long long M, void* stream);
int ganlab_ln_bwdbwd_apply_f32(const float* x, const float* mean, const float* rstd, const float* pu, const float* gx,
                               const float* sums, const float* usums, float* out, int N, long long M, void* stream);
/* ---- fused BatchNorm2d (training mode, first order) of the ResNet generators (csrc/norm.hip) --------------------
 * resnetgan/resblocks.py:15-121 -> NormalizeLayer('BatchNorm') = nn.BatchNorm2d (custom_layers.py:100-107).  A row is
 * a channel (N segments of HW elements); workspace as ganlab_ln_rowsums_workspace(C, N*HW).
 *   bn_stats:     out[c] = {batch mean, biased batch variance, 0}    (fp64 partial sums, evaluated in fp64)
 *   bn_bwd_sums:  out[c] = {sum gy (= d/d bias), sum gy * xhat (= d/d weight), 0}; yact / gz / slope as in ln_rowsums
 *   bn_bwd_apply: gx = pre[c] * (gy - s0/L - xhat * s1/L), pre = rstd * weight, L = N*HW
 *   bn_apply:     y = act((x - mean[c]) * scale[c] + shift[c])  (centred form: keeps the rounding error at eps*|y|)
 *   bn_finalize:  from bn_stats' out: {mean[C], var[C], rstd[C] = rsqrt(var + eps), scale[C] = rstd * weight} into
 *                 out[4][C]; running_mean / running_var (may be NULL) move by ``momentum`` towards the batch mean /
 *                 UNBIASED variance (var * unbias, unbias = L/(L-1)), *batches (may be NULL) += 1 - everything
 *                 nn.BatchNorm2d.forward does besides normalising (torch/nn/modules/batchnorm.py), one launch */
int ganlab_bn_stats_f32(const float* x, float* out, int N, int C, long long HW, void* workspace, size_t workspace_bytes,
                        void* stream);
int ganlab_bn_apply_f32(const float* x, const float* mean, const float* scale, const float* shift, float* y, int N,
                        int C, long long HW, int act, float slope, void* stream);
int ganlab_bn_finalize_f32(const float* mom, const float* weight, float* running_mean, float* running_var,
                           long long* batches, float* out, int C, float eps, float momentum, float unbias,
                           void* stream);
int ganlab_bn_bwd_sums_f32(const float* gy, const float* x, const float* mean, const float* rstd, float* out, int N,
                           int C, long long HW, void* workspace, size_t workspace_bytes, const float* yact, float* gz,
                           float slope, void* stream);
int ganlab_bn_bwd_apply_f32(const float* gy, const float* x, const float* mean, const float* rstd, const float* sums,
                            const float* pre, float* gx, int N, int C, long long HW, void* stream);
/* nn.Tanh of the ResNet generators (resnetgan/architectures.py:55, :93) */
int ganlab_tanh_fwd_f32(const float* x, float* y, long long n, void* stream);
int ganlab_tanh_bwd_f32(const float* gy, const float* y, float* gx, long long n, void* stream);

/* ---- minibatch stddev statistic (custom_layers.py:117-140), contiguous groups of gs samples ---- */
/* stat[g] = mean_f sqrt(var_unbiased_i(x[g,i,f]) + eps),  f over F = C*H*W features */
int ganlab_mbstd_fwd_f32(const float* x, float* stat, int G, int gs, long long F, float eps, void* stream);
/* gx[g,i,f] = gstat[g]/(F(gs-1)) * (x-mu)/s */
int ganlab_mbstd_bwd_f32(const float* x, const float* gstat, float* gx, int G, int gs, long long F,
                         float eps, void* stream);
/* backward of ganlab_mbstd_bwd: given ggx (cotangent of gx) -> g_gstat[g] and g_x[g,i,f] */
int ganlab_mbstd_bwdbwd_f32(const float* x, const float* gstat, const float* ggx, float* g_gstat,
                            float* g_x, int G, int gs, long long F, float eps, void* stream);

/* ---- elementwise / reductions ---- */
/* out = a*x + b*y (y may be NULL -> out = a*x); fade-in blends progan/architectures.py:163-167,311-315 */
int ganlab_axpby_f32(const float* x, const float* y, float* out, long long n, float a, float b, void* stream);
/* out = a * gout[0] * (x ? x : 1): backward of the scalar reductions, gout stays on the device */
int ganlab_scale_dev_f32(const float* x, const float* gout, float* out, long long n, float a, void* stream);
/* out[n,:] = t[n]*a[n,:] + (1-t[n])*b[n,:]  (WGAN-GP interpolates, resnetgan/learner.py:793-796) */
int ganlab_lerp_rows_f32(const float* a, const float* b, const float* t, float* out, long long N, long long M,
                         void* stream);
/* out[0] = scale * sum x  |  scale * sum x^2 */
int ganlab_sum_f32(const float* x, float* out, long long n, float scale, int squared, void* workspace,
                   size_t workspace_bytes, void* stream);
size_t ganlab_sum_workspace(long long n);
/* BCE-with-logits vs a constant target t, mean over n (progan/learner.py:793-800, :886-896) */
int ganlab_bce_logits_fwd_f32(const float* x, float* out, int n, float target, void* stream);
int ganlab_bce_logits_bwd_f32(const float* x, const float* gout, float* gx, int n, float target, void* stream);
/* WGAN-GP style penalty on channel-norms (resnetgan/learner.py:817-825):
 * out[0] = scale * sum_{n,hw} (sqrt(sum_c g^2) - gamma)^2 ;  bwd: gg = gout*scale*2*(s-gamma)*g/s */
int ganlab_chnorm_penalty_fwd_f32(const float* g, float* out, int N, int C, long long HW, float gamma,
                                  float scale, void* workspace, size_t workspace_bytes, void* stream);
int ganlab_chnorm_penalty_bwd_f32(const float* g, const float* gout, float* gg, int N, int C, long long HW,
                                  float gamma, float scale, void* stream);

/* ---- optimiser (torch.optim.Adam as configured by backprop_utils.py:109-120; EWMA progan/learner.py:909-916) */
/* One fused Adam step over a flat parameter arena; bias corrections are passed in by the host. */
int ganlab_adam_f32(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1,
                    float beta2, float eps, float wd, float bc1, float bc2, void* stream);
/* lagged = p*(1-beta) + lagged*beta */
int ganlab_ewma_f32(float* lagged, const float* p, long long n, float beta, void* stream);
/* counter-based N(0,1) generator (Philox4x32-10 + Box-Muller) for latents / per-layer noise */
int ganlab_randn_f32(float* out, long long n, uint64_t seed, uint64_t offset, void* stream);

/* ---- step-graph forms (gan_lab_amd/graphs.py GraphedStep; progan/learner.py:734-943 replayed as HIP graphs) ----
 * A captured launch replays with frozen arguments, so the per-step scalars live in a GANLAB_STEP_SCALARS_BYTES device
 * block: [0,8) the Philox stream position at the start of the replay, floats at byte 16: (lr, 1-beta1^t, 1-beta2^t) of
 * the critic's optimiser, then of the generator's.  One ordinary launch rewrites it before each replay. */
#define GANLAB_STEP_SCALARS_BYTES 64
int ganlab_step_scalars_size(void);
int ganlab_set_step_scalars(void* block, uint64_t rng_base, float f0, float f1, float f2, float f3, float f4, float f5,
                            void* stream);
/* ganlab_randn_f32 at stream position *base + delta (base: device uint64) */
int ganlab_randn_dev_f32(float* out, long long n, uint64_t seed, const void* base, uint64_t delta, void* stream);
/* ganlab_adam_f32 with (lr, bc1, bc2) read from device memory */
int ganlab_adam_dev_f32(float* p, const float* g, float* m, float* v, long long n, const float* lr_bc1_bc2, float beta1,
                        float beta2, float eps, float wd, void* stream);


/* ---- real-image input path -----------------------------------------------------------
 * uint8 NHWC dataset images -> 2^k box downsample -> fp32 NCHW ((v/255 - mean[c]) / std[c]); replaces the host
 * chain PIL Image.resize(BOX) -> ToTensor -> Normalize (data_config.py:307-341, progan/learner.py:1099-1112).
 * The uint8 stage is bit-exact with PIL's two-pass rounding for power-of-two factors; `flip` (nullable,
 * one byte per image) mirrors the image horizontally (RandomHorizontalFlip, data_config.py:332-333).
 * GANLAB_EUNSUPPORTED unless factor is a power of two dividing Hs and Ws. */
int ganlab_u8_box_decode_f32(const unsigned char* in_nhwc, float* out_nchw, int N, int Hs, int Ws, int C, int factor,
                             const float* mean, const float* stdv, const unsigned char* flip, void* stream);

/* ---- LeakyReLU masks as bits ------------------------------------------------------------------------------------------
 * The critic's block conv -> bias -> LeakyReLU -> blur (progan/architectures.py:261-284) needs only sign(y) of the LeakyReLU
 * output y in its backward (nn.LeakyReLU backward: grad * (y > 0 ? 1 : slope)): the blur pass that reads y emits the sign
 * bits (bit e of the NCHW-linear element index, 32 per word; W % 32 == 0), the backward passes read 1/32 of the bytes and y
 * itself is not kept. */
int ganlab_mask_bits_supported(int H, int W);
int ganlab_blur3x3_bits_f32(const float* x, float* y, unsigned* bits, long long planes, int H, int W, void* stream);
int ganlab_blur_act_bwd_bits_f32(const float* g, const unsigned* ybits, float* out, float* gb, int N, int C, int H, int W,
                                 float slope, float bias_scale, void* workspace, size_t workspace_bytes, void* stream);
int ganlab_act_bwd_blur_bits_f32(const float* g, const unsigned* ybits, const float* noise, float* out, float* gb,
                                 float* gnw, int N, int C, int H, int W, float slope, float bias_scale, void* workspace,
                                 size_t workspace_bytes, void* stream);
/* fromRGB (1x1, <= 3 input channels, progan/architectures.py:286-292) with its LeakyReLU mask as bits: the forward writes y and
 * the sign bits of y, the gradient kernels of ganlab_conv_{dgrad_act,fwd_mask,wgrad_act}_f32 read the bits instead of y */
int ganlab_conv_fwd_bits_f32(const float* x, const float* wp, const float* bias, float* y, unsigned* ybits,
                             const ganlab_conv_geom* g, float bias_scale, int act, float slope, void* stream);
/* The critic's  conv3x3 -> +bias -> LeakyReLU -> binomial blur  (progan/architectures.py:254-284) of a thin layer (<= 16 -> 16
 * channels, W % 64 == 0, H % 4 == 0) in ONE rolling-window kernel (csrc/conv_roll_blur.hip): y = blur(lrelu(conv + b)) and
 * the sign bits of the un-blurred activation (ybits may be NULL).  wp: GANLAB_PACK_FWD-packed weights. */
int ganlab_conv_fwd_blur_supported(const ganlab_conv_geom* g, const void* x, const void* y);
int ganlab_conv_fwd_blur_bits_f32(const float* x, const float* wp, const float* bias, float* y, unsigned* ybits,
                                  const ganlab_conv_geom* g, float bias_scale, float slope, void* stream);
/* The thin transposed stride-2 conv (32 low-resolution channels -> <= 16 high-resolution ones, Wl % 32 == 0) together with
 * the blur that follows it and what follows the blur (csrc/conv_s2_roll_blur.hip).  `g` is the UP layer (g->up) for the
 * generator's forward tail (stylegan/architectures.py:292-334, 497-526):
 *   y = act(blur(conv(up2(x * aff_s + aff_t), w)) + noise_w[c] * noise[n,hw] + bias[c] * bias_scale), mean / rstd = the
 *   InstanceNorm statistics of y (aff_s / aff_t NULL: plain input; wp: ganlab_conv_s2_pack_f32(up 1, transposed 0));
 * and the POOLED layer (g->pool) for the critic's backward (progan/architectures.py:254-284):
 *   gz = lrelu'(ybits) * blur(dgrad(gy, w)), gb[c] = bias_scale * sum gz (gb may be NULL) - the pooled conv's input gradient
 *   fused with the backward of the LeakyReLU -> blur in front of it (wp: what ganlab_conv_s2_dgrad_f32 takes). */
int ganlab_conv_s2_blur_supported(const ganlab_conv_geom* g);
size_t ganlab_conv_s2_blur_workspace(const ganlab_conv_geom* g);
int ganlab_conv_s2_fwd_blur_tail_f32(const float* x, const float* wp, const float* aff_s, const float* aff_t,
                                     const float* bias, const float* noise, const float* noise_w, float* y, float* mean,
                                     float* rstd, const ganlab_conv_geom* g, float bias_scale, int act, float slope, float eps,
                                     void* workspace, size_t workspace_bytes, void* stream);
int ganlab_conv_s2_dgrad_blur_act_bits_f32(const float* gy, const float* wp, const unsigned* ybits, float* gz, float* gb,
                                           const ganlab_conv_geom* g, float slope, float bias_scale, void* workspace,
                                           size_t workspace_bytes, void* stream);
/* The input gradient of the critic's first 3x3 conv (geometry g, thin: csrc/conv_roll_blur.hip) met by the backward of the
 * fromRGB layer in front of it (1x1 conv of the img_c <= 3 channel image + LeakyReLU, progan/architectures.py:232-237) inside
 * the kernel, for callers that do not need fromRGB's own input gradient: gz = dgrad(gy, w) * lrelu'(ybits) is never written;
 * gw_rgb[co][c] = scale * sum gz[co] * img[c], gb_rgb[co] = bias_scale * sum gz[co] (either may be NULL); acc_w / acc_b: add
 * to what they hold.  wp: what ganlab_conv_dgrad_f32 takes. */
int ganlab_conv_dgrad_rgb_sums_supported(const ganlab_conv_geom* g, int img_c);
size_t ganlab_conv_dgrad_rgb_sums_workspace(const ganlab_conv_geom* g);
int ganlab_conv_dgrad_rgb_sums_f32(const float* gy, const float* wp, const unsigned* ybits, const float* img, float* gw_rgb,
                                   float* gb_rgb, const ganlab_conv_geom* g, int img_c, float scale, float bias_scale,
                                   float slope, int acc_w, int acc_b, void* workspace, size_t workspace_bytes, void* stream);
int ganlab_conv_dgrad_act_bits_f32(const float* gy, const unsigned* ybits, const float* wp, float* gx,
                                   const ganlab_conv_geom* g, float slope, void* stream);
int ganlab_conv_fwd_mask_bits_f32(const float* x, const float* wp, const unsigned* ybits, float* out,
                                  const ganlab_conv_geom* g, float slope, void* stream);
int ganlab_conv_wgrad_act_bits_f32(const float* gy, const unsigned* ybits, const float* x, float* gw, float* gb,
                                   const ganlab_conv_geom* g, float scale, float bias_scale, float slope, void* workspace,
                                   size_t workspace_bytes, void* stream);

/* ---- all weight re-layouts of a network in ONE launch (csrc/pack.hip) -------------------------------------------------
 * The packed forms ganlab_conv_pack_f32 / ganlab_conv_s2_pack_f32 / ganlab_conv_pack_bf16 produce, rebuilt for a whole
 * table of weights after the optimiser rewrote them (Conv2dEx.forward multiplies by wscale on every call,
 * custom_layers.py:202-211; here the scaled re-layout is cached between optimiser steps).  `descs_device`: device copy of
 * n_desc descriptors sorted by block0.  One thread re-lays out one weight position with all of its taps (36 contiguous
 * source bytes), so descriptor i owns blocks [block0, block0 + ceil(total / taps / 256)) with taps = ks*ks (PLAIN), 16
 * (S2: the K4 taps of the stride-2 kernels) or 9 (BF16). */
#define GANLAB_PACKKIND_PLAIN 0
#define GANLAB_PACKKIND_S2 1
#define GANLAB_PACKKIND_BF16 2
#define GANLAB_PACKKIND_X3 3     /* three-plane bf16 k-step images of csrc/conv_x3.hip (27*Cout*Cin bf16 elements) */
typedef struct ganlab_pack_desc {
  const float* src;      /* OIHW parameter */
  void* dst;             /* packed buffer of `total` elements (float, or bf16 for GANLAB_PACKKIND_BF16 / _X3) */
  int kind;              /* GANLAB_PACKKIND_* */
  int Cout, Cin, ks;
  int mode;              /* PLAIN / BF16: GANLAB_PACK_FWD or _DGRAD; S2: transpose flag */
  int up;                /* S2: 1 = up layer, 0 = down (pooled) layer */
  float scale;
  int reserved;
  long long total;
  long long block0;
} ganlab_pack_desc;
int ganlab_pack_desc_size(void);
int ganlab_pack_many(const ganlab_pack_desc* descs_device, int n_desc, long long total_blocks, void* stream);

/* ---- deferred InstanceNorm: "affine on load" consumers (csrc/mod.hip, template flag AFF in the conv kernels) ------------
 * The generator layer of stylegan/architectures.py:497-526 ends in b = a*s[n,c] + t[n,c] (InstanceNorm + AdaIN of the
 * activated tensor a; custom_layers.py:98-99 + stylegan/architectures.py:524-526).  These entry points let the CONSUMER of
 * b read a instead and apply the per-(sample, channel) affine while the patch goes from registers to LDS - elements in the
 * zero padding of b stay zero.  aff_s / aff_t: [N][Cin] floats.  Replaces the second pass of ganlab_instnorm_style_fwd_f32
 * followed by the plain kernel; the input gradient of such a layer is the plain ganlab_conv_dgrad_f32 / _s2_dgrad_f32 (it is
 * d loss / d b, which the InstanceNorm backward takes). */
/* s = rstd*(ys+1), t = yb - mean*s per (n, c); style = (N, 2C) = [ys | yb] or NULL */
int ganlab_in_affine_f32(const float* mean, const float* rstd, const float* style, float* s, float* t, int N, int C,
                         void* stream);
/* plain 3x3 'same' layer on planes >= 32 wide with more than 16 output channels: bit 0 = forward, bit 1 = weight gradient */
int ganlab_conv_aff_supported(const ganlab_conv_geom* g);
/* ganlab_conv_fwd_f32 on b = a*s + t */
int ganlab_conv_fwd_aff_f32(const float* x, const float* wp, const float* aff_s, const float* aff_t, const float* bias,
                            float* y, const ganlab_conv_geom* g, float bias_scale, int act, float slope, void* stream);
/* ganlab_conv_wgrad_f32 with b = a*s + t as the x operand (workspace: ganlab_conv_wgrad_workspace) */
/* The same layer one size up: a plain 3x3 generator layer (Cout > 16, W % 32 == 0, H % 8 == 0) with a deferred-InstanceNorm
 * input and its whole tail in the conv kernel's epilogue (stylegan/architectures.py:497-526): y = act(conv(x * aff_s + aff_t, w)
 * + noise_w * noise + bias * bias_scale), mean / rstd = InstanceNorm statistics of y.  ganlab_conv_fwd_aff_tail_chunks: tiles
 * per plane (workspace: N * Cout * tiles * 2 doubles), 0 where the form does not apply. */
int ganlab_conv_fwd_aff_tail_chunks(const ganlab_conv_geom* g);
int ganlab_conv_fwd_aff_tail_f32(const float* x, const float* wp, const float* aff_s, const float* aff_t, const float* bias,
                                 const float* noise, const float* noise_w, float* y, float* mean, float* rstd,
                                 const ganlab_conv_geom* g, float bias_scale, int act, float slope, float eps, void* workspace,
                                 size_t workspace_bytes, void* stream);
int ganlab_conv_wgrad_aff_f32(const float* gy, const float* x, const float* aff_s, const float* aff_t, float* gw,
                              const ganlab_conv_geom* g, float scale, void* workspace, size_t workspace_bytes, void* stream);
/* Upsample + conv3x3 (up = 1) with a deferred low-resolution input: bit 0 = forward, bit 1 = weight gradient */
int ganlab_conv_s2_aff_supported(const ganlab_conv_geom* g);
int ganlab_conv_s2_fwd_aff_f32(const float* x, const float* wp, const float* aff_s, const float* aff_t, const float* bias,
                               float* y, const ganlab_conv_geom* g, float bias_scale, int act, float slope, void* stream);
int ganlab_conv_s2_wgrad_aff_f32(const float* gy, const float* x, const float* aff_s, const float* aff_t, float* gw,
                                 const ganlab_conv_geom* g, float scale, void* workspace, size_t workspace_bytes,
                                 void* stream);
/* thin 3x3 layer (Cin, Cout <= 16, W % 64 == 0, H % 4 == 0) that also finishes itself */
int ganlab_mod_conv_supported(const ganlab_conv_geom* g);
/* statistics chunks per (n, co) plane written by the forward: workspace = N*Cout*chunks*2 doubles */
int ganlab_mod_conv_stat_chunks(const ganlab_conv_geom* g);
/* y = act(conv(a*s + t, w) + bias*bias_scale + noise_w*noise) and, when mean / rstd are given, the InstanceNorm statistics
 * (biased variance, eps) of y from the same pass.  wp: ganlab_conv_pack_f32(GANLAB_PACK_FWD); aff_s / aff_t NULL: plain input. */
int ganlab_mod_conv_fwd_f32(const float* x, const float* wp, const float* aff_s, const float* aff_t, const float* bias,
                            const float* noise, const float* noise_w, float* y, float* mean, float* rstd,
                            const ganlab_conv_geom* g, float bias_scale, int act, float slope, float eps,
                            void* workspace, size_t workspace_bytes, void* stream);
/* toRGB (1x1, linear) of a deferred tensor: weff[n][ci][4] = scale*w[co][ci]*s[n,ci], beff[n][4] = bias*bias_scale + scale*w.t */
int ganlab_mod_torgb_prep_f32(const float* w, const float* bias, const float* s, const float* t, float* weff, float* beff,
                              int N, int Cin, int Cout, float scale, float bias_scale, void* stream);
/* gw (Cout, Cin) / gb (Cout) (either may be NULL) from ganlab_mod_torgb_cross_f32's N x 68 sums */
int ganlab_mod_torgb_wgrad_f32(const float* cross, const float* s, const float* t, float* gw, float* gb, int N, int Cin,
                               int Cout, float scale, float bias_scale, void* stream);
/* y[n,co] = sum_ci weff[n][ci][4] * x[n,ci] + beff[n][4]   (Cout <= 4) */
int ganlab_mod_torgb_fwd_f32(const float* x, const float* weff, const float* beff, float* y, int N, int Cin, int Cout,
                             long long HW, void* stream);
size_t ganlab_mod_torgb_cross_workspace(int N);
/* out[n][68]: [ci 16][4] = sum_px x[n,ci]*gy[n,co], then [64 + co] = sum_px gy[n,co]   (Cin <= 16, Cout <= 4) */
int ganlab_mod_torgb_cross_f32(const float* x, const float* gy, float* out, int N, int Cin, int Cout, long long HW,
                               void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GANLAB_HIP_H */
