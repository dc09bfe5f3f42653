/*
 * dcvgan_hip.h — C ABI of libdcvgan_hip.so: the hand-written gfx950 (MI355X)
 * kernels under the DCVGAN generator/discriminator training step.
 *
 * The reference (raahii/dcvgan) is pure Python over torch.nn; it has no FFI of
 * its own.  Each entry point below replaces one torch.nn / torch.optim call
 * site of the reference's hot path (file:line given per function); the Python
 * host side (dcvgan_amd/native.py) binds them with ctypes — see INTEGRATION.md.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless
 *     the name ends in _host; all tensors are fp32;
 *   - tensors are described as 5-D N,C,D,H,W sizes + element strides
 *     (2-D layers use D = 1), so the non-contiguous NCDHW views the reference
 *     produces (generator.py:139,433) are consumed without a copy;
 *   - `stream` is a hipStream_t passed as void*; nothing synchronises the host;
 *   - the caller owns every buffer, including `ws` scratch (size from the
 *     matching *_workspace_bytes call; 256-byte aligned);
 *   - return value: 0 on success, a negative DCV_E* code otherwise; nothing
 *     throws across the ABI.  dcv_last_error() gives a text for the last
 *     failure on the calling thread.
 *   - re-entrant: autograd runs backward kernels from its own thread.
 */
#ifndef DCVGAN_HIP_H
#define DCVGAN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DCV_OK 0
#define DCV_EINVAL (-1)   /* bad shape / geometry / null pointer          */
#define DCV_EWORKSPACE (-2) /* ws too small                                */
#define DCV_EHIP (-3)     /* a HIP runtime call failed                     */
#define DCV_EUNSUPPORTED (-4)

/* activation codes for fused epilogues / elementwise kernels */
#define DCV_ACT_NONE 0
#define DCV_ACT_LEAKY 1   /* LeakyReLU(slope); slope 0 = ReLU              */
#define DCV_ACT_TANH 2

typedef struct dcv_dims5 {
    int32_t n, c, d, h, w;        /* sizes                                  */
    int64_t sn, sc, sd, sh, sw;   /* element strides                        */
} dcv_dims5;

/* Geometry of one nn.Conv2d / nn.Conv3d / nn.ConvTranspose2d module.
 * cin/cout are the MODULE's in/out channels; weight layout is torch's:
 * (cout, cin, kd, kh, kw) for conv, (cin, cout, kd, kh, kw) for transposed. */
typedef struct dcv_conv_geom {
    int32_t kd, kh, kw;
    int32_t sd, sh, sw;
    int32_t pd, ph, pw;
    int32_t transposed;
    int32_t cin, cout;
    /* precision of the MFMA products for THIS module's three passes: 0 = the process default (dcv_set_precision), 1 = fp32, 2 = bf16 products,
     * 3 = fp32 emulated on the bf16 matrix pipe (each operand split into three bf16 pieces, 6 products, fp32 accumulation — see dcv_set_precision)
     * (a per-module switch; dcvgan_amd.util.set_precision(module, "bf16") sets it on a module's convolutions) */
    int32_t mfma;
} dcv_conv_geom;

const char* dcv_last_error(void);
/* ABI version.  4 (round 6): dcv_scale_dev, dcv_conv_backward_data_bn(_workspace_bytes), dcv_conv_forward_bn, dcv_conv_backward_weight_bn, dcv_bn_forward_stats_only, dcv_bn_apply exist (no struct changed).  3 (round 5): dcv_conv_backward_weight_acc / dcv_cl_conv_backward_weight_acc, dcv_cl_conv_backward_data_gated, dcv_clf16_*, dcv_normal_fill_many exist (no struct changed).
 * 2 (round 4): dcv_conv_geom has the 13th field `mfma`, dcv_wpack the 4th field `precision`, dcv_abi_struct_sizes exists.
 * A host compares dcv_version() and dcv_abi_struct_sizes() with its own declarations BEFORE the first call that passes a struct
 * (dcvgan_amd/native.py does, and refuses to load on a mismatch): the library cannot see the size of what a pointer points to. */
int dcv_version(void);
/* out[0..2] = sizeof(dcv_dims5), sizeof(dcv_conv_geom), sizeof(dcv_wpack) as this library was compiled */
void dcv_abi_struct_sizes(size_t out[3]);
/* number of kernel launches issued through this library so far (tests use it to
 * prove the HIP path, not a fallback, did the work) */
uint64_t dcv_launch_count(void);
/* DEFAULT precision of the MFMA products in the large GEMM kernels, for modules whose dcv_conv_geom.mfma is 0: 0 = fp32 (default; the mode every parity claim
 * and the headline benchmark refer to), 1 = bf16 products with fp32 accumulation (v_mfma_f32_32x32x16_bf16): tensors,
 * weights, BatchNorm statistics and optimiser state stay fp32, only the MFMA fragments are rounded (RNE) as they are read
 * from LDS.  A throughput mode for BASELINE.json's bf16 / fp16 configs; the reference itself is fp32-only. */
/* mode 2 (round 4, experimental, never the default): fp32 EMULATED on the bf16 matrix pipe.  Every fp32 operand is split exactly into three bf16 pieces
 * x = hi + mid + lo (RNE at each level); a bf16 x bf16 product is exact in fp32, and the six products of total order <= 2 (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi,
 * mid*mid) are accumulated in fp32 by v_mfma_f32_32x32x16_bf16 — the dropped terms are <= 2^-23 of a product, the size of fp32's own rounding of it.  6 bf16
 * MFMAs (192 cycles) replace 8 fp32 ones (512 cycles) per 32 x 32 x 16 block; error against fp64 measured beside the native kernel's in profiles/r04_f32x6_*. */
int dcv_set_precision(int mode);
int dcv_get_precision(void);
/* the precision code (1 fp32, 2 bf16 products, 3 fp32-on-bf16) a dcv_conv_* call with this geometry would run at right now: g->mfma, or the process default
 * when that is 0.  A caller that owns packed weights (dcv_wpack) stamps them with it. */
int dcv_conv_effective_precision(const dcv_conv_geom* g);
/* diagnostics: which GEMM kernel instance the calling thread's last dcv_conv_* call launched (bench.py / tools label
 * their per-layer timings with it) */
const char* dcv_debug_last_kernel(void);
/* diagnostics text: resident workgroups/CU, registers, LDS of every GEMM kernel (needs a GPU) */
int dcv_debug_kernel_info(char* buf, size_t n);

/* ---- convolutions ------------------------------------------------------- *
 * Replace nn.Conv2d (generator.py:174,204; discriminator.py:83,89,95-101),
 * nn.Conv3d (discriminator.py:181-206,288-305) and nn.ConvTranspose2d
 * (generator.py:61-73,239-241,273-275) forward and their autograd backward.
 * Implicit GEMM on v_mfma_f32_32x32x2_f32.
 *   forward        : y = act(conv(x, w))            (act fused in the epilogue)
 *   backward_data  : dx (+)= conv^T(dy, w)
 *   backward_weight: dw  = corr(x, dy)              (deterministic split-K)
 */
size_t dcv_conv_workspace_bytes(const dcv_conv_geom* g, const dcv_dims5* x, const dcv_dims5* y, int which /*0 fwd,1 bwd-data,2 bwd-weight*/);
/* forward / backward_data read the weights K-major ("packed"); by default they re-pack them into `ws` on every call.
 * Weights only change at optimiser steps (trainer.py:320-322,357-359) while each layer runs 2-3 times per phase, so a
 * caller may own the packed copy instead: a buffer of dcv_conv_packed_bytes(g, x, y, which) bytes per (layer, which,
 * input geometry), passed with ready = 0 the first time after the weights changed (the call packs into it) and
 * ready = 1 afterwards (the packing launches are skipped).  pack = NULL keeps the default.
 * A packed copy is valid for ONE (weights, effective precision) pair — the packed FORMAT depends on the precision (fp32 [k][OCp]; bf16 products
 * [k/8][OCp][8] bf16; fp32-on-bf16: three such planes).  `precision` carries that pair's second half across the ABI: the caller sets it to
 * dcv_conv_effective_precision(g) when it hands the buffer over with ready = 0, keeps it with the buffer, and passes it back with ready = 1; a call whose
 * own effective precision differs from a ready pack's returns DCV_EINVAL before any launch (it never reads a pack of the other format), and a
 * ready = 0 call whose `precision` is not the call's own is refused the same way.  precision = 0 is refused too: there is no "unchecked" pack. */
typedef struct dcv_wpack {
    float* buf;
    size_t bytes;
    int32_t ready;
    int32_t precision;   /* 1 fp32, 2 bf16 products, 3 fp32-on-bf16: what `buf` was / is to be packed for */
} dcv_wpack;
size_t dcv_conv_packed_bytes(const dcv_conv_geom* g, const dcv_dims5* x, const dcv_dims5* y, int which /*0 fwd,1 bwd-data*/);
int dcv_conv_forward(const dcv_conv_geom* g, const float* x, const dcv_dims5* xd, const float* w,
                     float* y, const dcv_dims5* yd, int act, float slope, const dcv_wpack* pack,
                     void* ws, size_t ws_bytes, void* stream);
/* conv forward that also leaves per-tile BatchNorm partial sums of y (a conv -> BatchNorm pair, generator.py /
 * discriminator.py blocks): stat[part][pitch][2] = {sum, sum of squares}; *nparts = 0 when this geometry's
 * kernel cannot produce them (the caller then runs the plain statistics pass).  No activation, no accumulate. */
size_t dcv_conv_stats_bytes(const dcv_conv_geom* g, const dcv_dims5* x, const dcv_dims5* y);
int dcv_conv_forward_stats(const dcv_conv_geom* g, const float* x, const dcv_dims5* xd, const float* w, float* y, const dcv_dims5* yd,
                           float* stat, size_t stat_bytes, int* nparts, int* pitch, const dcv_wpack* pack, void* ws, size_t ws_bytes, void* stream);
int dcv_conv_backward_data(const dcv_conv_geom* g, const float* dy, const dcv_dims5* dyd, const float* w,
                           float* dx, const dcv_dims5* dxd, int accumulate, const dcv_wpack* pack,
                           void* ws, size_t ws_bytes, void* stream);
/* backward_data followed by the (Leaky)ReLU derivative of the layer that PRODUCED this conv's input, read off that
 * input itself: dx = (accumulate ? dx : 0) + conv^T(dy, w), then dx *= (x > 0 ? 1 : slope).  x must have dx's shape and
 * strides.  Replaces the separate derivative pass of an nn.Conv2d + nn.LeakyReLU pair whose output feeds this conv
 * (Inconv -> DownBlock 0, generator.py:173-176,203-207).  DCV_EUNSUPPORTED (before any launch) for geometries whose
 * kernels lack the gated epilogue: the caller then runs the two steps separately. */
int dcv_conv_backward_data_gated(const dcv_conv_geom* g, const float* dy, const dcv_dims5* dyd, const float* w,
                                 float* dx, const dcv_dims5* dxd, int accumulate,
                                 const float* x, const dcv_dims5* xd, int act, float slope, const dcv_wpack* pack,
                                 void* ws, size_t ws_bytes, void* stream);
/* A BatchNorm (+ activation) group whose output is NEVER WRITTEN (round 6; the colour generator's UpBlock 5, generator.py:238-250: 1.17 GB per pass at B = 70 that
 * only the RGB head reads): dcv_bn_forward_stats_only finalises the statistics the producing convolution's epilogue left (dcv_conv_forward_stats) and updates the running
 * statistics; the head's forward (dcv_conv_forward_bn) and weight gradient (dcv_conv_backward_weight_bn) then read the BatchNorm INPUT for the operand's first cbn
 * channels and apply act(x * gamma * invstd + beta - mean * gamma * invstd) on load, and dcv_conv_backward_data_bn (below) is the matching backward.
 * The *_bn conv entries take only the head's geometry (3x3 / 1 / 1, 3 output channels, 64-wide rows, fp32) and return DCV_EUNSUPPORTED for anything else BEFORE running
 * anything; a caller that gets that materialises the output with dcv_bn_apply and uses the plain entries. */
int dcv_bn_forward_stats_only(const float* x, const dcv_dims5* xd, float* running_mean, float* running_var, int64_t* num_batches_tracked, float* save_mean, float* save_invstd,
                              float momentum, float eps, const float* stat, int nparts, int pitch, void* stream);
int dcv_bn_apply(const float* x, const dcv_dims5* xd, float* y, const dcv_dims5* yd, const float* gamma, const float* beta, const float* save_mean, const float* save_invstd,
                 const float* mask, int act, float slope, void* stream);
int dcv_conv_forward_bn(const dcv_conv_geom* g, const float* x, const dcv_dims5* xd, const float* w, float* y, const dcv_dims5* yd, int act, float slope,
                        const dcv_wpack* pack, void* ws, size_t ws_bytes, int cbn, const float* bn_x, const dcv_dims5* bn_xd, const float* gamma, const float* beta,
                        const float* save_mean, const float* save_invstd, int bn_act, float bn_slope, void* stream);
int dcv_conv_backward_weight_bn(const dcv_conv_geom* g, const float* x, const dcv_dims5* xd, const float* dy, const dcv_dims5* dyd, float* dw, int accumulate,
                                void* ws, size_t ws_bytes, int cbn, const float* bn_x, const dcv_dims5* bn_xd, const float* gamma, const float* beta,
                                const float* save_mean, const float* save_invstd, int bn_act, float bn_slope, void* stream);
/* The data gradient of a convolution whose first `cbn` input channels are the output of a BatchNorm (training mode, no dropout mask) + (Leaky)ReLU | identity, FUSED
 * with that BatchNorm's backward — the last stage of the colour generator: `UpBlock` 5 -> torch.cat with the stem's skip -> `Outconv`
 * (generator.py:238-250,272-277,393-400).  Where the geometry is the RGB head's (3x3 / stride 1 / pad 1 transposed, 3 -> 128 channels on 64-wide rows, fp32) the gradient
 * of those cbn channels is never written: it is recomputed from dy inside the BatchNorm reduction and inside the kernel that writes the gradient of the BatchNorm INPUT
 * (4.7 GB of HBM traffic per call instead of 8.2 GB at B = 70).  On return *fused = 1: dx[:, cbn:] , bn_dx, dgamma, dbeta are written and dx[:, :cbn] is NOT;
 * *fused = 0: the plain data gradient ran (any other geometry), dx is complete and the caller runs dcv_bn_act_backward itself.  ws / pack: as dcv_conv_backward_data;
 * ws2: dcv_conv_backward_data_bn_workspace_bytes(dxd, cbn) bytes.  act: DCV_ACT_NONE or DCV_ACT_LEAKY (slope 0 = ReLU) of the BatchNorm group. */
size_t dcv_conv_backward_data_bn_workspace_bytes(const dcv_dims5* dxd, int cbn);
int dcv_conv_backward_data_bn(const dcv_conv_geom* g, const float* dy, const dcv_dims5* dyd, const float* w, float* dx, const dcv_dims5* dxd, const dcv_wpack* pack,
                              void* ws, size_t ws_bytes, int cbn, const float* bn_x, const dcv_dims5* bn_xd, const float* gamma, const float* beta,
                              const float* save_mean, const float* save_invstd, int act, float slope, float* bn_dx, const dcv_dims5* bn_dxd, float* dgamma, float* dbeta,
                              void* ws2, size_t ws2_bytes, int* fused, void* stream);
int dcv_conv_backward_weight(const dcv_conv_geom* g, const float* x, const dcv_dims5* xd,
                             const float* dy, const dcv_dims5* dyd, float* dw,
                             void* ws, size_t ws_bytes, void* stream);
/* dw = (accumulate ? dw : 0) + corr(x, dy)  (ABI 3).  Replaces the elementwise sums autograd forms when a weight is used twice in one backward (every discriminator
 * parameter: D on the real and on the fake batch, trainer.py:299-309 -> :319) or when .grad already holds an earlier backward's gradient (the discriminators in the
 * G phase, trainer.py:356 on top of :319; discriminator.zero_grad() only at :288-290): the slab reduce adds its fixed-order sum to dw — the same two operands and
 * one rounding as torch's add, so the result is bit-identical. */
int dcv_conv_backward_weight_acc(const dcv_conv_geom* g, const float* x, const dcv_dims5* xd,
                                 const float* dy, const dcv_dims5* dyd, float* dw, int accumulate,
                                 void* ws, size_t ws_bytes, void* stream);

/* ---- BatchNorm{2,3}d (+ Dropout2d) (+ activation) ------------------------ *
 * Replace nn.BatchNorm2d/3d + nn.Dropout2d + nn.(Leaky)ReLU chains
 * (generator.py:62-72,205-211,242-248; discriminator.py:96-100,191-202,290-302).
 * Training mode: batch statistics over (N, D, H, W), biased variance for the
 * normalisation, unbiased for running_var, momentum 0.1 (torch defaults).
 *   y = act( mask[n,c] * ( gamma * (x - mean) * invstd + beta ) )
 * `mask` (N*C floats, 0 or 1/(1-p)) may be NULL (no dropout).
 * save_mean / save_invstd (C floats each) are outputs in training mode and
 * inputs to the backward.  In eval mode (training = 0) running stats are used.
 */
size_t dcv_bn_workspace_bytes(int channels);
int dcv_bn_act_forward(const float* x, const dcv_dims5* xd, float* y, const dcv_dims5* yd,
                       const float* gamma, const float* beta,
                       float* running_mean, float* running_var, int64_t* num_batches_tracked /* may be NULL; += 1 when training */,
                       float* save_mean, float* save_invstd,
                       const float* mask, int training, float momentum, float eps,
                       int act, float slope, void* ws, size_t ws_bytes, void* stream);
/* dx, dgamma, dbeta from dy; x is the BN input, y unused. dgamma/dbeta are
 * OVERWRITTEN (C floats each). */
/* dcv_bn_act_forward with the batch statistics taken from dcv_conv_forward_stats' partial sums (training mode) */
int dcv_bn_act_forward_stats(const float* x, const dcv_dims5* xd, float* y, const dcv_dims5* yd, const float* gamma, const float* beta,
                             float* running_mean, float* running_var, int64_t* num_batches_tracked, float* save_mean, float* save_invstd, const float* mask,
                             float momentum, float eps, int act, float slope, const float* stat, int nparts, int pitch,
                             void* ws, size_t ws_bytes, void* stream);
int dcv_bn_act_backward(const float* dy, const dcv_dims5* dyd, const float* x, const dcv_dims5* xd,
                        float* dx, const dcv_dims5* dxd,
                        const float* gamma, const float* beta, const float* save_mean, const float* save_invstd,
                        const float* mask, int training, int act, float slope,
                        float* dgamma, float* dbeta, void* ws, size_t ws_bytes, void* stream);

/* ---- elementwise --------------------------------------------------------- */
/* y = act(x)  /  dx = dy * act'(.) evaluated from the OUTPUT y
 * (nn.LeakyReLU generator.py:175, discriminator.py:84,90,186; nn.Tanh generator.py:78,276) */
int dcv_act_forward(const float* x, const dcv_dims5* xd, float* y, const dcv_dims5* yd, int act, float slope, void* stream);
int dcv_act_backward(const float* dy, const dcv_dims5* dyd, const float* y, const dcv_dims5* yd,
                     float* dx, const dcv_dims5* dxd, int act, float slope, void* stream);
/* y = a*x + b*z   (z may be NULL): Noise add with an injected sample
 * (discriminator.py:30-39), temporal difference of gdis (discriminator.py:330-331),
 * gradient accumulation, strided copies (torch.cat at generator.py:393-400,
 * discriminator.py:124,228 is two such copies into channel slices). */
int dcv_axpby(const float* x, const dcv_dims5* xd, float a, const float* z, const dcv_dims5* zd, float b,
              float* y, const dcv_dims5* yd, void* stream);
/* y = x + sigma * N(0,1) drawn on the device (Philox4x32-10 + Box-Muller), the
 * production form of discriminator.py:30-39.  (seed, offset) select the stream. */
int dcv_noise_add(const float* x, const dcv_dims5* xd, float* y, const dcv_dims5* yd,
                  float sigma, uint64_t seed, uint64_t offset, void* stream);
/* out[i] = N(0,1), i < n   (latents: generator.py:85,88,104,356) */
int dcv_normal_fill(float* out, int64_t n, uint64_t seed, uint64_t offset, void* stream);
/* `count` consecutive draws of n values each, out[j * n + i]: draw j holds exactly what dcv_normal_fill(.., seed, offset + j) would write (one launch for the
 * generator's per-frame motion noise, models.py / generator.py:57-62 of the reference: video_length draws of (batch, dim_z_motion)). */
int dcv_normal_fill_many(float* out, int64_t n, int64_t count, uint64_t seed, uint64_t offset, void* stream);
/* Dropout2d(p) plane mask: mask[i] = Bernoulli(1-p) / (1-p), i < n = N*C (generator.py:211,248) */
int dcv_dropout_mask(float* mask, int64_t n, float p, uint64_t seed, uint64_t offset, void* stream);

/* ---- input pipeline (dataset.py:125-186): disk-order frames -> training tensors ----------- *
 * out (B,C,T,H,W) fp32 = float(in (B,T,H,W,C)) / div - sub, numpy's fp32 operation order:
 * colour / depth PNG frames: in uint8, div 127.5, sub 1.0 (dataset.py:131, 168);
 * optical flow: in fp32, div image_size, sub 0 (dataset.py:174).                              */
int dcv_decode_video(const void* in, int in_is_u8, int B, int T, int H, int W, int C, float div, float sub, float* out, void* stream);
/* SURREAL depth (dataset.py:137-156): depth (B,T,H,W) fp32 with background >= 1e10 -> out (B,1,T,H,W):
 * foreground min-max normalised per clip to [-1, 0.8], background 1.0; ws_minmax: 2*B floats.  */
int dcv_surreal_depth(const float* depth, int B, int T, int H, int W, float* out, float* ws_minmax, void* stream);

/* ---- segmentation branch (surreal-segm.yml, 25 body-part channels) ---------- *
 * softmax over the channel axis = the geometry generator's nn.Softmax(dim=1) head (generator.py:75-76)  */
int dcv_softmax_channels_forward(const float* x, const dcv_dims5* xd, float* y, const dcv_dims5* yd, void* stream);
int dcv_softmax_channels_backward(const float* dy, const dcv_dims5* dyd, const float* y, const dcv_dims5* yd, float* dx, const dcv_dims5* dxd, void* stream);
/* one-hot / softmax maps -> {-1,+1} maps: argmax over channels (first maximum), scatter (generator.py:378-385) */
int dcv_segm_onehot(const float* x, const dcv_dims5* xd, float* y, const dcv_dims5* yd, void* stream);
/* argmax -> part colour (util.py:236-246); palette = C x 3 bytes on the device; out uint8 (N,3,D,H,W)  */
int dcv_segm_to_rgb(const float* x, const dcv_dims5* xd, const uint8_t* palette, uint8_t* out, void* stream);
/* dataset.py:176-181: label frames uint8 (B,T,H,W) -> one-hot fp32 (B,C,T,H,W)                       */
int dcv_decode_segmentation(const uint8_t* labels, int B, int T, int H, int W, int C, float* out, void* stream);

/* ---- sampling path: float videos -> uint8 on the device --------------------- *
 * util.videos_to_numpy (util.py:58-79) and the depth branch of
 * util.geometric_info_in_color_format (util.py:219-222): out = uint8((clip(x,-1,1)+1)/2*255),
 * same fp32 operation order (bytes identical); out is contiguous (N, C*channel_repeat, D, H, W).  */
int dcv_videos_to_uint8(const float* x, const dcv_dims5* xd, uint8_t* out, int channel_repeat, void* stream);
/* util.visualize_optical_flow (util.py:143-170) for a (B,2,T,H,W) flow video scaled by `scale`
 * (= H, util.py:227): hue = direction, value = per-frame min-max normalised magnitude; out uint8
 * (B,3,T,H,W); ws_minmax: 2*B*T floats of scratch. */
int dcv_flow_to_rgb(const float* flow, const dcv_dims5* fd, float scale, uint8_t* out, float* ws_minmax, void* stream);

/* ---- GAN losses (fused value + gradient) --------------------------------- *
 * loss.py:91-99,123-131 (BCE-with-logits, sum / numel) and loss.py:163-164,
 * 190-191 (hinge / softplus).  kind: 0 = BCE target 1, 1 = BCE target 0,
 * 2 = mean(relu(1 - y)), 3 = mean(relu(1 + y)), 4 = mean(softplus(-y)).
 * *loss_out (+)= value ; dy_out[i] = d value / d y[i].                        */
int dcv_gan_loss(const float* y, int64_t n, int kind, float* loss_out, int accumulate, float* dy_out, void* stream);
/* y[i] = x[i] * *s, s a 0-d DEVICE scalar: the backward of a loss term — `loss.backward()` (trainer.py:319,356) hands every
 * term of loss.py:99,131,164,191 the upstream cotangent as a device tensor; dy_out of dcv_gan_loss times it, without a host read. */
int dcv_scale_dev(const float* x, int64_t n, const float* s, float* y, void* stream);

/* ---- GRUCell (generator.py:58,94) --------------------------------------- *
 * The whole T-step motion-latent recurrence in one launch: h_t = GRU(e_t, h_{t-1}).
 * e: (T, B, dm) noise, h0: (B, dm); out: (B, T, dm) (= torch.stack(h[1:], 1));
 * gates: (T, B, 4*dm) saved [r, z, n, hn_pre] for the backward.                */
size_t dcv_gru_workspace_bytes(int B, int dm);
int dcv_gru_forward(const float* e, const float* h0, const float* w_ih, const float* w_hh,
                    const float* b_ih, const float* b_hh, float* out, float* gates,
                    int T, int B, int dm, void* stream);
int dcv_gru_backward(const float* dout, const float* e, const float* h0, const float* out, const float* gates,
                     const float* w_ih, const float* w_hh,
                     float* dw_ih, float* dw_hh, float* db_ih, float* db_hh,
                     int T, int B, int dm, void* ws, size_t ws_bytes, void* stream);

/* ---- Adam (train.py:171-176: betas (0.5, 0.999), eps 1e-8, L2 weight decay) */
/* torch.optim.Adam's update operation for operation; the bias corrections 1 - beta^step and the step size
 * lr / (1 - beta1^step) are formed in double on the host (torch forms them as Python floats) and rounded once.
 * g is scaled by grad_scale first (1 / world size under data parallelism, 1 otherwise). */
int dcv_adam_step(float* p, const float* g, float* m, float* v, int64_t n,
                  double lr, double beta1, double beta2, double eps, double weight_decay, int step,
                  double grad_scale, void* stream);
/* the same update for n_tensors parameter tensors of one optimiser (one shared step count) in ceil(n/24) launches */
int dcv_adam_step_multi(int n_tensors, float* const* p, const float* const* g, float* const* m, float* const* v, const int64_t* numel,
                        double lr, double beta1, double beta2, double eps, double weight_decay, int step, double grad_scale, void* stream);

/* ---- bf16 channels-last ("CL16") data path -------------------------------------------------- *
 * BASELINE.json configs[2] ("surreal-depth1, bf16 MFMA") and configs[4] ("fp16 MFMA") name 16-bit variants of the same step
 * (config/surreal-depth1.yml:5,47-76, config/isogd-flow.yml; the reference itself is fp32-only).  This is that path as a DATA path:
 * activations and their gradients are bf16 in HBM with the channel innermost (memory order n, d, h, w, c: `sc` = 1, the pixel pitch
 * a multiple of 8 elements and >= the channel count rounded up to 8 — padding channels hold zeros), so an MFMA operand fragment
 * (8 consecutive k = channels of one tap) is one 16-byte read for every stride / padding; weights stay fp32 masters in torch layout
 * (packed to bf16 K-major tiles by dcv_cl_pack_weights, once per optimiser step), accumulators, BatchNorm statistics, weight gradients and
 * the optimiser stay fp32.  v_mfma_f32_32x32x16_bf16.  Same call sites as the fp32 entry points above; `void*` tensors are bf16,
 * described by dcv_dims5 with element strides.  A throughput path with its own tolerance (tests/test_cl16_gpu.py), never the default. */
size_t dcv_cl_packed_bytes(const dcv_conv_geom* g, const dcv_dims5* x, const dcv_dims5* y, int which /*0 fwd, 1 bwd-data*/);
int dcv_cl_pack_weights(const dcv_conv_geom* g, const dcv_dims5* x, const dcv_dims5* y, int which, const float* w, void* packed, size_t bytes, void* stream);
/* scratch of a forward / backward-data call (thin destinations — <= 8 channels fed by a wide source — run as a 1x1 GEMM over the source followed by a gather
 * of each destination pixel's taps, and keep the GEMM's result there) */
size_t dcv_cl_conv_workspace_bytes(const dcv_conv_geom* g, const dcv_dims5* x, const dcv_dims5* y, int which);
int dcv_cl_conv_forward(const dcv_conv_geom* g, const void* x, const dcv_dims5* xd, const void* packed, void* y, const dcv_dims5* yd,
                        int act, float slope, void* ws, size_t ws_bytes, void* stream);
/* conv -> BatchNorm pairs in training mode (as dcv_conv_forward_stats on the fp32 path): the epilogue also leaves {sum, sum of squares} of the STORED bf16 values per position tile
 * and output channel: *nparts rows of *pitch channels x 2 floats in `stat` (dcv_cl_conv_stats_bytes; 0 = this geometry's form produces none, and *nparts stays 0: the BatchNorm op
 * then makes its own pass).  dcv_cl_bn_act_forward_stats consumes them. */
size_t dcv_cl_conv_stats_bytes(const dcv_conv_geom* g, const dcv_dims5* x, const dcv_dims5* y);
int dcv_cl_conv_forward_stats(const dcv_conv_geom* g, const void* x, const dcv_dims5* xd, const void* packed, void* y, const dcv_dims5* yd,
                              float* stat, size_t stat_bytes, int* nparts, int* pitch, void* ws, size_t ws_bytes, void* stream);
int dcv_cl_conv_backward_data(const dcv_conv_geom* g, const void* dy, const dcv_dims5* dyd, const void* packed, void* dx, const dcv_dims5* dxd,
                              int accumulate, void* ws, size_t ws_bytes, void* stream);
/* backward_data + the (Leaky)ReLU derivative of the layer that produced this convolution's input, read off that input `xg` (dx's shape and strides), in one epilogue
 * (ABI 3): dx = (accumulate ? dx : 0) + conv^T(dy, w); dx *= (xg > 0 ? 1 : slope).  The bf16 counterpart of dcv_conv_backward_data_gated (Inconv -> DownBlock 0,
 * generator.py:173-176,203-207).  DCV_EUNSUPPORTED before any launch where the form has no such epilogue. */
int dcv_cl_conv_backward_data_gated(const dcv_conv_geom* g, const void* dy, const dcv_dims5* dyd, const void* packed, void* dx, const dcv_dims5* dxd,
                                    int accumulate, const void* xg, const dcv_dims5* xgd, int act, float slope, void* ws, size_t ws_bytes, void* stream);
size_t dcv_cl_wgrad_workspace_bytes(const dcv_conv_geom* g, const dcv_dims5* x, const dcv_dims5* y);
int dcv_cl_conv_backward_weight(const dcv_conv_geom* g, const void* x, const dcv_dims5* xd, const void* dy, const dcv_dims5* dyd, float* dw,
                                void* ws, size_t ws_bytes, void* stream);
/* dw = (accumulate ? dw : 0) + corr(x, dy)  (ABI 3; as dcv_conv_backward_weight_acc) */
int dcv_cl_conv_backward_weight_acc(const dcv_conv_geom* g, const void* x, const dcv_dims5* xd, const void* dy, const dcv_dims5* dyd, float* dw, int accumulate,
                                    void* ws, size_t ws_bytes, void* stream);
/* module boundary: fp32 NCDHW (any strides) <-> bf16 channels-last (same shape; padding channels are written as zeros).
 * dcv_cl_to_f32 with accumulate = 1 adds into y (a gradient arriving at an fp32 leaf). */
int dcv_cl_from_f32(const float* x, const dcv_dims5* xd, void* y, const dcv_dims5* yd, void* stream);
int dcv_cl_to_f32(const void* x, const dcv_dims5* xd, float* y, const dcv_dims5* yd, int accumulate, void* stream);
/* kind 0: y = x   1: y = a x + b z   2: y = x + a N(0,1) (Philox; discriminator.py:30-39)   3: dx = x * lrelu'(z; slope a) (x = dy, z = the activation's output)
 * 4: dx = x * (1 - z^2) (tanh, z = output)   5: y = lrelu(x; a)   6: y = tanh(x) */
int dcv_cl_elementwise(int kind, const void* x, const dcv_dims5* xd, const void* z, const dcv_dims5* zd, void* y, const dcv_dims5* yd,
                       float a, float b, uint64_t seed, uint64_t offset, void* stream);
/* BatchNorm{2,3}d (+ Dropout2d mask) (+ (Leaky)ReLU) on bf16 channels-last tensors; statistics, parameters and their gradients fp32
 * (generator.py:62-72,205-211,242-248; discriminator.py:96-100,191-202,290-302).  Same semantics as dcv_bn_act_forward / _backward. */
size_t dcv_cl_bn_workspace_bytes(int channels);
int dcv_cl_bn_act_forward(const void* x, const dcv_dims5* xd, void* y, const dcv_dims5* yd, const float* gamma, const float* beta,
                          float* running_mean, float* running_var, int64_t* num_batches_tracked, float* save_mean, float* save_invstd,
                          const float* mask, int training, float momentum, float eps, int act, float slope, void* ws, size_t ws_bytes, void* stream);
int dcv_cl_bn_act_forward_stats(const void* x, const dcv_dims5* xd, void* y, const dcv_dims5* yd, const float* gamma, const float* beta,
                                float* running_mean, float* running_var, int64_t* num_batches_tracked, float* save_mean, float* save_invstd,
                                const float* mask, float momentum, float eps, int act, float slope, const float* stat, int nparts, int pitch,
                                void* ws, size_t ws_bytes, void* stream);
int dcv_cl_bn_act_backward(const void* dy, const dcv_dims5* dyd, const void* x, const dcv_dims5* xd, void* dx, const dcv_dims5* dxd,
                           const float* gamma, const float* beta, const float* save_mean, const float* save_invstd, const float* mask,
                           int training, int act, float slope, float* dgamma, float* dbeta, void* ws, size_t ws_bytes, void* stream);


/* ---- the same 16-bit channels-last path with fp16 as its element type (ABI 3) ------------ *
 * BASELINE configs[4] names "fp16 MFMA" (config/isogd-flow.yml on 32 x 128 x 128 clips: the size-agnostic discriminators, discriminator.py:181-206,288-305).
 * conv_cl16.hip / cl_elementwise.hip are compiled a second time with _Float16 as the element type and v_mfma_f32_32x32x16_f16: identical layout, kernels and
 * semantics, 10 mantissa bits instead of 7, largest finite value 65504 (a pre-BatchNorm sum beyond it becomes inf: the stress leg of bench.py reports the
 * largest magnitude it saw).  Same signatures as the dcv_cl_* entry points above; tensors are torch.float16 on the host side. */
size_t dcv_clf16_packed_bytes(const dcv_conv_geom* g, const dcv_dims5* x, const dcv_dims5* y, int which );
size_t dcv_clf16_conv_workspace_bytes(const dcv_conv_geom* g, const dcv_dims5* x, const dcv_dims5* y, int which);
int dcv_clf16_pack_weights(const dcv_conv_geom* g, const dcv_dims5* x, const dcv_dims5* y, int which, const float* w, void* packed, size_t bytes, void* stream);
int dcv_clf16_conv_forward(const dcv_conv_geom* g, const void* x, const dcv_dims5* xd, const void* packed, void* y, const dcv_dims5* yd, int act, float slope, void* ws, size_t ws_bytes, void* stream);
size_t dcv_clf16_conv_stats_bytes(const dcv_conv_geom* g, const dcv_dims5* x, const dcv_dims5* y);
int dcv_clf16_conv_forward_stats(const dcv_conv_geom* g, const void* x, const dcv_dims5* xd, const void* packed, void* y, const dcv_dims5* yd, float* stat, size_t stat_bytes, int* nparts, int* pitch, void* ws, size_t ws_bytes, void* stream);
int dcv_clf16_conv_backward_data(const dcv_conv_geom* g, const void* dy, const dcv_dims5* dyd, const void* packed, void* dx, const dcv_dims5* dxd, int accumulate, void* ws, size_t ws_bytes, void* stream);
int dcv_clf16_conv_backward_data_gated(const dcv_conv_geom* g, const void* dy, const dcv_dims5* dyd, const void* packed, void* dx, const dcv_dims5* dxd, int accumulate, const void* xg, const dcv_dims5* xgd, int act, float slope, void* ws, size_t ws_bytes, void* stream);
size_t dcv_clf16_wgrad_workspace_bytes(const dcv_conv_geom* g, const dcv_dims5* x, const dcv_dims5* y);
int dcv_clf16_conv_backward_weight(const dcv_conv_geom* g, const void* x, const dcv_dims5* xd, const void* dy, const dcv_dims5* dyd, float* dw, void* ws, size_t ws_bytes, void* stream);
int dcv_clf16_conv_backward_weight_acc(const dcv_conv_geom* g, const void* x, const dcv_dims5* xd, const void* dy, const dcv_dims5* dyd, float* dw, int accumulate, void* ws, size_t ws_bytes, void* stream);
int dcv_clf16_from_f32(const float* x, const dcv_dims5* xd, void* y, const dcv_dims5* yd, void* stream);
int dcv_clf16_to_f32(const void* x, const dcv_dims5* xd, float* y, const dcv_dims5* yd, int accumulate, void* stream);
int dcv_clf16_elementwise(int kind, const void* x, const dcv_dims5* xd, const void* z, const dcv_dims5* zd, void* y, const dcv_dims5* yd, float a, float b, uint64_t seed, uint64_t offset, void* stream);
size_t dcv_clf16_bn_workspace_bytes(int channels);
int dcv_clf16_bn_act_forward(const void* x, const dcv_dims5* xd, void* y, const dcv_dims5* yd, const float* gamma, const float* beta, float* running_mean, float* running_var, int64_t* num_batches_tracked, float* save_mean, float* save_invstd, const float* mask, int training, float momentum, float eps, int act, float slope, void* ws, size_t ws_bytes, void* stream);
int dcv_clf16_bn_act_forward_stats(const void* x, const dcv_dims5* xd, void* y, const dcv_dims5* yd, const float* gamma, const float* beta, float* running_mean, float* running_var, int64_t* num_batches_tracked, float* save_mean, float* save_invstd, const float* mask, float momentum, float eps, int act, float slope, const float* stat, int nparts, int pitch, void* ws, size_t ws_bytes, void* stream);
int dcv_clf16_bn_act_backward(const void* dy, const dcv_dims5* dyd, const void* x, const dcv_dims5* xd, void* dx, const dcv_dims5* dxd, const float* gamma, const float* beta, const float* save_mean, const float* save_invstd, const float* mask, int training, int act, float slope, float* dgamma, float* dbeta, void* ws, size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DCVGAN_HIP_H */
