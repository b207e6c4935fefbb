/* ustrun.h -- C ABI of libustrun.so: the MI355X (gfx950) implementation of UST-RUN's
 * 2D U-Net training hot path.
 *
 * The reference has no FFI: its boundary for this path is the torch.nn.Module API of
 * networks/unet_model.py + utils/losses.py and the inline step in train.py.  Each entry point
 * below names the reference code it replaces (file:line under the reference tree).
 *
 * Conventions
 *   - every function returns 0 on success, nonzero on error; ustrun_last_error() gives the
 *     message (thread-local).  No exceptions cross the ABI, nothing is allocated or freed here,
 *     no pointer is retained after return, nothing synchronises the device: all work is
 *     enqueued on `stream` (a hipStream_t; NULL = default stream).
 *   - all pointers are DEVICE pointers unless named host_*.
 *   - activations inside the network are NHWC ("pixel-major": channels of one pixel are
 *     contiguous); the network input and the logits are NCHW as in the reference.
 *   - dtype selects the storage type of activations, activation gradients and packed weights:
 *     USTRUN_F32 (exact f32 MFMA, the parity path), USTRUN_BF16 (bf16 tensors in HBM, bf16 MFMA,
 *     f32 accumulate / BatchNorm statistics / losses / optimizer) or USTRUN_F16 (the same kernels
 *     built for IEEE half: the reference's own mixed-precision type, torch.cuda.amp autocast +
 *     GradScaler, train.py:30,54,551-552,842-845; activation GRADIENTS in half need the caller's loss
 *     scale: ustrun_seg_loss_bwd's gscale carries it in, ustrun_sgd_ema_scaled takes it out and skips
 *     the update when the scaled gradient is not finite).  `void*` activation arguments follow
 *     dtype; the network input, logits, parameters and gradients of parameters are always f32.
 *     USTRUN_F32X3: every tensor, loader and epilogue of USTRUN_F32, with the products of the 3x3 convolutions (and the
 *     ConvTranspose pair) on the BF16 matrix cores over three-term operand splits x = x0 + x1 + x2, six MFMAs per product
 *     (csrc/x3.hip): f32-level results (the north_star tolerance: 1e-4 on logits, arg-max bit-exact outside 1e-4 margins) at a
 *     multiple of the f32 matrix rate.  Its packed weights are the f32 pack followed by the three bf16 planes: 2.5x the floats.
 */
#ifndef USTRUN_H
#define USTRUN_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define USTRUN_VERSION 100
enum { USTRUN_F32 = 0, USTRUN_BF16 = 1, USTRUN_F16 = 2, USTRUN_F32X3 = 3 };
enum { USTRUN_LOSS_SOFTMAX = 0, USTRUN_LOSS_SIGMOID = 1 };

typedef void* ustrun_stream_t;

int ustrun_version(void);
const char* ustrun_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Activation source of an implicit-GEMM loader.  The loader evaluates, per element,
 *     a = relu?( x*scale[c] + shift[c] )   then, if pool, max over the 2x2 window,
 * and zero outside the stored extent (conv zero padding, F.pad of the up path).  This is how
 * BatchNorm+ReLU (unet_parts.py:17-18), MaxPool2d(2) (:34), F.pad (:62-63) and torch.cat (:67)
 * are folded into the consumer's loads instead of being materialised.
 * ---------------------------------------------------------------------------------------- */
typedef struct ustrun_src {
    const void*  ptr;        /* stored tensor                                              */
    const float* scale;      /* per-channel affine, NULL = identity                        */
    const float* shift;
    int32_t C, H, W;         /* stored channels / extent                                   */
    int64_t sN, sH, sW, sC;  /* element strides (NHWC: sC = 1; NCHW input: sW = 1)         */
    int32_t relu;            /* max(0, .) after the affine                                 */
    int32_t pool;            /* logical pixel (y,x) = max of stored (2y..2y+1, 2x..2x+1)   */
    int32_t off_y, off_x;    /* logical (y,x) reads stored (y-off_y, x-off_x)              */
    int32_t f32;             /* 1: this tensor is f32 even when dtype is USTRUN_BF16 (the    */
                             /*    network input); otherwise its element type follows dtype  */
    int32_t gN;              /* > 0: the batch is several independent forward passes laid end to end, gN images */
    int64_t gstride;         /*      each, with their own BatchNorm statistics: image n uses scale/shift +      */
                             /*      (n / gN) * gstride; statistics rows of a forward never mix two passes       */
} ustrun_src_t;

/* ---- weight packing (done once per optimizer step) ---------------------------------------
 * conv3x3: torch [Cout][Cin][3][3] -> fwd [9][Cin][Cout] and dgrad [9][Cout][Cin]
 * convT2x2: torch [Cin][Cout][2][2] -> fwd [4][Cin][Cout] and dgrad [4][Cout][Cin]
 * (USTRUN_F32X3: each of the two buffers holds 2.5 x that many floats -- the f32 pack, then its bf16 planes)  */
int ustrun_pack_conv3x3(const float* w, int Cout, int Cin, void* w_fwd, void* w_dgrad, int dtype, ustrun_stream_t s);
int ustrun_pack_convT2x2(const float* w, int Cin, int Cout, void* w_fwd, void* w_dgrad, int dtype, ustrun_stream_t s);

/* ---- conv3x3 (pad 1, no bias) forward: replaces nn.Conv2d at unet_parts.py:16,19 ----------
 * y[N,H,W,Cout] (raw, pre-BN) = conv(loader(srcs)), plus per-channel partial sums of y and
 * y*y per M-tile for the train-mode BatchNorm that follows: stat[mtile][2][Cout] (NULL = skip).
 * ustrun_conv_mtiles gives the number of M-tiles the launch will use.                        */
int ustrun_conv_mtiles(int N, int H, int W, int Cout);
int ustrun_conv3x3_fwd(const ustrun_src_t* srcs, int nsrc, const void* w_fwd, int N, int H, int W,
                       int Cout, void* y, float* stat, int dtype, ustrun_stream_t s);
/* Same launch, but reports in *stat_rows how many rows of stat it wrote (<= ustrun_conv_mtiles) instead of
 * zero-filling the rest: pass that count to ustrun_bn_finalize and no memset is enqueued.            */
int ustrun_conv3x3_fwd_rows(const ustrun_src_t* srcs, int nsrc, const void* w_fwd, int N, int H, int W,
                            int Cout, void* y, float* stat, int* stat_rows, int dtype, ustrun_stream_t s);

/* ---- train-mode BatchNorm statistics: replaces nn.BatchNorm2d at unet_parts.py:17,20 -------
 * Reduces stat[mtiles][2][C] (fixed order, f64) to batch mean / biased variance, writes
 * scale = gamma*rstd, shift = beta - mean*scale (consumed by the next loader), mean, rstd
 * (kept for backward), and updates running_mean/var (unbiased var, momentum) and
 * num_batches_tracked when update_running != 0.  eval mode: ustrun_bn_eval_affine.           */
int ustrun_bn_finalize(float* stat /* scratch: may be overwritten */, int mtiles, int C, int64_t count, const float* gamma,
                       const float* beta, float* running_mean, float* running_var,
                       int64_t* num_batches_tracked, float momentum, float eps, int update_running,
                       float* scale, float* shift, float* mean, float* rstd, ustrun_stream_t s);
int ustrun_bn_eval_affine(int C, const float* gamma, const float* beta, const float* running_mean,
                          const float* running_var, float eps, float* scale, float* shift,
                          ustrun_stream_t s);
/* materialise a = relu(y*scale+shift) as NHWC or NCHW f32 (block-level API and feature=True) */
int ustrun_bn_relu_apply(const void* y, const float* scale, const float* shift, int64_t npix, int C,
                         int HW, float* out, int out_nchw, int dtype, ustrun_stream_t s);
/* materialise the Down block's pooled input: out[N,H/2,W/2,C] = MaxPool2d(2)(relu(src*scale+shift)) in the storage
 * dtype (replaces nn.MaxPool2d at unet_parts.py:34 together with the producer's BatchNorm+ReLU); src is a plain
 * contiguous NHWC activation, its pass groups are honoured                                                     */
/* a[N,H,W,C] = relu(src*scale+shift) in the 16-bit storage dtype (src: plain contiguous NHWC with its BatchNorm constants, pass
 * groups honoured): the input of a DoubleConv's second convolution (unet_parts.py:16-21) written out once, for the layers where
 * that costs less than applying BatchNorm + ReLU per staged item in the convolution AND its weight gradient                */
int ustrun_act16(const ustrun_src_t* src, int N, void* out, int dtype, ustrun_stream_t s);
int ustrun_pool_act(const ustrun_src_t* src, int N, void* out, int dtype, ustrun_stream_t s);
/* the same pass, also writing the un-pooled activation act[N,H,W,C] = relu(src*scale+shift) in the storage dtype (act may be
 * NULL: then exactly ustrun_pool_act): the encoder output that unet_model.py:33-36 hands to the decoder as the skip operand of
 * torch.cat (unet_parts.py:66) -- the concat convolution and its weight gradient then read a plain tensor instead of applying
 * BatchNorm + ReLU per staged item; 16-bit storage and even H, W only                                                     */
int ustrun_pool_act2(const ustrun_src_t* src, int N, void* out, void* act, int dtype, ustrun_stream_t s);
/* plain MaxPool2d(2) backward on NHWC tensors (stand-alone Down block): dx[N,H,W,C] from dp[N,H/2,W/2,C] and the
 * pooled input x; first maximum of the window wins (torch's rule)                                              */
int ustrun_maxpool_bwd(const void* dp, const void* x, int N, int H, int W, int C, void* dx, int dtype,
                       ustrun_stream_t s);

/* ---- ConvTranspose2d(k=2,s=2,bias): replaces unet_parts.py:53 ------------------------------
 * u[N,2H,2W,Cout] = convT(loader(src)) + bias.                                              */
int ustrun_convT2x2_fwd(const ustrun_src_t* src, const void* w_fwd, const float* bias, int N, int H,
                        int W, int Cout, void* u, int dtype, ustrun_stream_t s);

/* ---- 1x1 head with bias: replaces OutConv, unet_parts.py:71-76 -----------------------------
 * logits NCHW f32 [N,K,H,W] from y[N,H,W,C] through the loader affine (scale/shift/relu).    */
int ustrun_head_fwd(const void* y, const float* scale, const float* shift, int64_t npix, int HW,
                    int C, int K, const float* w, const float* bias, float* logits, int dtype,
                    ustrun_stream_t s);
/* backward: da[N,H,W,C] (grad wrt the activated input), dw[K][C], db[K] (accumulate != 0 adds) */
int ustrun_head_bwd(const float* dlogits, const void* y, const float* scale, const float* shift,
                    int64_t npix, int HW, int C, int K, const float* w, void* da, float* dw, float* db,
                    int accumulate, float* partials, int64_t partials_bytes, int dtype, ustrun_stream_t s);

/* ---- BatchNorm+ReLU (+MaxPool) backward: autograd of unet_parts.py:17-18,34 ------------------
 * Inputs: da (grad wrt the activated output, may be NULL), dp (grad wrt the 2x2-pooled
 * activated output, may be NULL; routed to the window arg-max recomputed from y), y and the
 * forward scale/shift/mean/rstd.  Pass 1 (reduce) produces dgamma/dbeta and the coefficient
 * table coef[3][C]; pass 2 (apply) writes dy = coef0*dz + coef1*y + coef2, dz = da_total*(a>0). */
int64_t ustrun_bn_bwd_partials_bytes(int64_t npix, int C);
int ustrun_bn_bwd_reduce(const void* da, const void* dp, const void* y, const float* scale,
                         const float* shift, const float* mean, const float* rstd, const float* gamma,
                         int N, int H, int W, int C, float* dgamma, float* dbeta, int accumulate,
                         float* coef, float* partials, int64_t partials_bytes, int dtype, ustrun_stream_t s);
int ustrun_bn_bwd_apply(const void* da, const void* dp, const void* y, const float* scale,
                        const float* shift, const float* coef, int N, int H, int W, int C, void* dy,
                        int dtype, ustrun_stream_t s);

/* ---- conv3x3 input gradient (autograd of unet_parts.py:16,19) ------------------------------
 * da = conv3x3(dy, flipped/transposed w).  Channels [0,C0) go to da0[N,H,W,C0]; channels
 * [C0,Cin) go to da1[N,H1,W1,Cin-C0] at pixel (y-o1y, x-o1x) (the cat/pad split of the up path). */
int ustrun_conv3x3_dgrad(const void* dy, const void* w_dgrad, int N, int H, int W, int Cout, int Cin,
                         void* da0, int C0, void* da1, int H1, int W1, int o1y, int o1x, int dtype,
                         ustrun_stream_t s);
/* the same input gradient (single destination) which ALSO forms the BatchNorm-backward sums of the layer whose da it writes
 * (unet_parts.py:17-21 backward: the BatchNorm2d + ReLU between the two convolutions of a DoubleConv): y = that layer's
 * pre-BatchNorm output [N,H,W,Cin] (the layout of da), scale / shift = its forward constants (gN images per pass, gstride floats
 * between the passes' constants; gN = 0: one pass); stat receives *stat_rows rows of [2][Cin] = {sum(da mask), sum(da mask y)},
 * mask = y scale + shift > 0, over the da values as stored -- what ustrun_bn_bwd_reduce would form from a second read of both
 * tensors; ustrun_bn_bwd_finalize_stat turns the rows into dgamma / dbeta / coefficients.  *stat_rows = 0 and NO launch when
 * the shape is not one the fused epilogue covers (16-bit storage, >= 128 input channels, the 256-pixel tiles): the caller then
 * runs ustrun_conv3x3_dgrad and ustrun_bn_bwd_reduce.  stat: ustrun_conv_mtiles(N, H, W, Cin) rows. */
int ustrun_conv3x3_dgrad_bnsum(const void* dy, const void* w_dgrad, int N, int H, int W, int Cout, int Cin, void* da,
                               const void* y, const float* scale, const float* shift, int gN, int64_t gstride,
                               float* stat, int* stat_rows, int dtype, ustrun_stream_t s);
/* rows of ustrun_conv3x3_dgrad_bnsum (rows_per_pass x passes rows of [2][C], pass after pass) -> dgamma, dbeta (the passes'
 * contributions added in order; accumulate != 0 adds to what is there) and coef[pass][3][C] for ustrun_bn_bwd_apply; the table
 * is reduced in place (it is scratch afterwards)                                                                          */
int ustrun_bn_bwd_finalize_stat(float* stat, int rows_per_pass, int passes, int C, int64_t count, const float* gamma,
                                const float* mean, const float* rstd, int64_t aff_stride, float* dgamma, float* dbeta,
                                int accumulate, float* coef, ustrun_stream_t s);
/* ---- conv3x3 weight gradient: dw[Cout][Cin][3][3] (torch layout, f32) ---------------------- */
int64_t ustrun_wgrad_partials_bytes(int nseg, int Cin, int Cout, int64_t npix);
int ustrun_conv3x3_wgrad(const ustrun_src_t* srcs, int nsrc, const void* dy, int N, int H, int W,
                         int Cout, float* dw, int accumulate, float* partials, int64_t partials_bytes,
                         int dtype, ustrun_stream_t s);
/* ---- ConvTranspose2d backward --------------------------------------------------------------- */
int ustrun_convT2x2_dgrad(const void* du, const void* w_dgrad, int N, int H, int W, int Cout, int Cin,
                          void* da, int dtype, ustrun_stream_t s);
/* the same, also forming the BatchNorm-backward sums of the layer whose da it writes -- the conv2 under this ConvTranspose
 * (unet_parts.py:56-68 backward into unet_parts.py:17-21): arguments and rows as ustrun_conv3x3_dgrad_bnsum, one row per 128
 * pixels; *stat_rows = 0 and no launch on a shape the fused epilogue does not cover                                       */
int ustrun_convT2x2_dgrad_bnsum(const void* du, const void* w_dgrad, int N, int H, int W, int Cout, int Cin, void* da,
                                const void* y, const float* scale, const float* shift, int gN, int64_t gstride,
                                float* stat, int* stat_rows, int dtype, ustrun_stream_t s);
int ustrun_convT2x2_wgrad(const ustrun_src_t* src, const void* du, int N, int H, int W, int Cout,
                          float* dw, float* db, int accumulate, float* partials, int64_t partials_bytes,
                          int dtype, ustrun_stream_t s);

/* ---- bilinear Up block: nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True), unet_parts.py:48-50 ----
 * x[N,H,W,C] -> y[N,2H,2W,C] (NHWC f32, C % 4 == 0); _bwd is its exact adjoint (dy -> dx), fixed summation order. */
int ustrun_upsample2x_fwd(const float* x, int N, int H, int W, int C, float* y, ustrun_stream_t s);
int ustrun_upsample2x_bwd(const float* dy, int N, int H, int W, int C, float* dx, ustrun_stream_t s);
/* the same two passes in a plan's storage dtype (the fused U-Net with bilinear = 1): the forward reads a plain contiguous NHWC
 * source through its BatchNorm constants + ReLU (pass groups honoured) and stores the interpolated activation; the backward is the
 * adjoint over dy [N,2H,2W,C] -> dx [N,H,W,C]                                                                                   */
int ustrun_upsample2x_act(const ustrun_src_t* src, int N, void* out, int dtype, ustrun_stream_t s);
int ustrun_upsample2x_bwd_t(const void* dy, int N, int H, int W, int C, void* dx, int dtype, ustrun_stream_t s);

/* ---- pseudo-labels: replaces train.py:648-667 (teacher) and :669-674 (student) ---------------
 * softmax: conf,label = max(softmax(logits,1),1); mask = conf > th   (label int64, mask f32)
 * sigmoid: label = (p >= .5); mask = (p >= th) + (p <= 1-th)          (both f32 [N,K,H,W])   */
int ustrun_pseudo_label(const float* logits, int N, int K, int HW, float threshold, int mode,
                        void* label, float* mask, ustrun_stream_t s);

/* ---- ensemble/CutMix target algebra: replaces train.py:677-697 -------------------------------
 * box[N,HW] in {0,1}; cut_label/cut_mask are already gathered by `choice`.                    */
int ustrun_mix_targets(int mode, int N, int K, int HW, const float* box, const void* pl,
                       const float* mask, const void* pl_w_ul, const float* mask_w_ul,
                       const void* pl_w_lu, const float* mask_w_lu, const void* cut_label,
                       const float* cut_mask, void* pl_w, float* mask_w, void* pl_ul, float* mask_ul,
                       void* pl_lu, float* mask_lu, ustrun_stream_t s);
/* CutMix rectangles -> {0,1} maps on the device: box[n,y,x] = (y0<=y<y1 && x0<=x<x1).  Replaces the host-built
 * maps of train.py:222-251 (cutmix_box, all_cover_box) that the reference uploads with .cuda(): only the N x 4 corner
 * ints {y0,y1,x0,x1} (HOST pointer) travel, inside the launch's argument block.                 */
#define USTRUN_MAX_RECTS 64
int ustrun_rect_masks(const int32_t* rects_host, int N, int H, int W, float* box, ustrun_stream_t s);
/* Stream-ordered upload of a few host bytes (indices, ratios) through the launch's argument block; replaces the
 * small torch.tensor(...).cuda() uploads of train.py:612-636,749-782.                          */
#define USTRUN_UPLOAD_MAX 2048
int ustrun_upload_small(void* dst, const void* src_host, int nbytes, ustrun_stream_t s);
/* CutMix image compositing out = a*(1-box) + b*box on NCHW images: train.py:644-646,688,691   */
int ustrun_box_mix(const float* a, const float* b, const float* box, int N, int C, int HW,
                   float* out, ustrun_stream_t s);
/* One launch that ASSEMBLES a batch from rows that live anywhere on the device: out row r (row_bytes bytes each, a multiple of
 * 16; every pointer 16-byte aligned) = a byte copy of rows[r].a when rows[r].b is NULL, else the CutMix composite of f32 rows
 * a*(1-box) + b*box with box[HW] broadcast over the row's channels (the arithmetic of ustrun_box_mix).  The row table is a HOST
 * array (device ADDRESSES computed by the caller: base + index * stride) and travels in the launch's argument block.  Replaces
 * the reference's fancy indexing / torch.cat / clone chains around its forward calls: cut_img[choice], cut_label[choice],
 * cut_mask[choice] (train.py:627,690-697), the three teacher inputs (:643-647), the student's inputs (:689-702,734), the
 * memory bank's torch.cat (:764-780) -- without materialising the concatenations or the index tensors.              */
typedef struct ustrun_asm_row { const void* a; const void* b; const float* box; } ustrun_asm_row_t;
#define USTRUN_ASM_MAX 128
int ustrun_assemble(const ustrun_asm_row_t* rows_host, int nrows, int64_t row_bytes, int HW, void* out, ustrun_stream_t s);
/* Label decoding of the loop head, one pass: train.py:590-608 / train_mnms.py:549-556.  y: the loader's f32 label tensor.
 * kind 0 (fundus):   y [N,HW]   -> out f32 [N,2,HW] = {y == 0, y <= 128}
 * kind 1 (prostate): y [N,HW]   -> out i64 [N,HW]   = (y == 0)
 * kind 2 (BUSI):     y [N,HW]   -> out i64 [N,HW]   = (y == 255)
 * kind 3 (MNMS):     y [N,HW,3] -> out i64 [N,HW]   = 3 if y[..,2]==255 else 2 if y[..,1]==255 else 1 if y[..,0]==255 else 0 */
int ustrun_decode_labels(const float* y, int kind, int N, int HW, void* out, ustrun_stream_t s);
/* Bounding rectangle of the union of up to four planes' non-zero pixels (train.py:722-730 + obtain_all_cover_box :242-251: the
 * low-quality sample's region = its pseudo-label planes united with the picked labelled sample's mask planes): plane k is
 * f32 or i64 (is_i64 bit k) of H*W elements; partial[USTRUN_BBOX_BLOCKS][4] int32 = per-block {min y, max y, min x, max x}
 * ({H, -1, W, -1} for a block that saw none) -- 1 KB that the host copies and folds: the 256 KB region map never exists.  */
#define USTRUN_BBOX_BLOCKS 64
int ustrun_region_bbox(const void* const* planes_host, int nplanes, int is_i64_bits, int H, int W, int32_t* partial,
                       ustrun_stream_t s);

/* ---- FFT low-frequency amplitude mix: replaces train.py:158-207,628-636 (numpy on the host) --------
 * src/trg/out are normalised NCHW images (k/127.5-1); per image n the (2b+1)^2 window of src's
 * fft-shifted amplitude is blended towards trg's with ratio ratios[n] (device array), phase kept;
 * out = clip(result,0,255)/127.5-1.  b = floor(min(H,W)*L).                                       */
int64_t ustrun_freq_mix_work_bytes(int n, int C, int b);
int ustrun_freq_mix(const float* src, const float* trg, const float* ratios, int n, int C, int H, int W,
                    int b, float* out, void* work, int64_t work_bytes, ustrun_stream_t s);

/* ---- loss term ce + dice: replaces train.py:816-817,829-836 with utils/losses.py:236-268 -----
 * softmax mode: target int64 [N,HW], mask f32 [N,HW] or NULL;  CE(reduction none)*mask, mean over
 *   all pixels (Q5) + per-class masked Dice with the class-0 mask all ones (Q4).
 * sigmoid mode: target/mask f32 [N,K,HW];  BCE-with-logits*mask mean + ONE global masked Dice.
 * fwd writes out[0]=ce, out[1]=dice, out[2..] = the reduced sums the backward needs
 * (USTRUN_LOSS_NSUMS(K) floats in all).  bwd writes
 *   dlogits = gscale * (ce_weight*g[0] * dce/dlogits + dice_weight*g[1] * ddice/dlogits),
 * g = gscale_dev (two device floats: upstream grads of ce and dice) or {1,1} when NULL.          */
#define USTRUN_LOSS_NSUMS(K) (4 + 3 * (K))
int64_t ustrun_loss_partials_bytes(int N, int K, int HW);
int ustrun_seg_loss_fwd(const float* logits, const void* target, const float* mask, int N, int K,
                        int HW, int mode, float* out, float* partials, int64_t partials_bytes,
                        ustrun_stream_t s);
int ustrun_seg_loss_bwd(const float* logits, const void* target, const float* mask, int N, int K,
                        int HW, int mode, const float* sums, const float* gscale_dev, float gscale,
                        float ce_weight, float dice_weight, float* dlogits, ustrun_stream_t s);

/* ---- DiceLossWithMask.forward in every mode combination of utils/losses.py:236-268 (the two the step uses run fused with
 * their CE / BCE terms in ustrun_seg_loss_*; this pair serves the rest of the signature: class weights, sigmoid per class,
 * softmax + multi, raw inputs).  act: 0 none, 1 softmax over classes, 2 sigmoid.  multi = 0: per-class Dice on the one-hot of a
 * class-index target [N,HW] (int64 or f32: target_is_i64), mask f32 [N,HW] or NULL encoded as the reference does (class 0:
 * all ones, class k >= 1: mask == 1 -- Q4), mean over classes of weight[k] * dice_k (weight_host: K host floats or NULL = 1).
 * multi = 1: ONE Dice over all of [N,K,HW]; target / mask f32 [N,K,HW], or [N,HW] broadcast over the classes when
 * target_per_pixel / mask_per_pixel is set.  fwd: out[0] = loss, out[1..3K] = the sums the backward needs (1 + 3K floats);
 * partials as ustrun_loss_partials_bytes.  bwd: dlogits = gscale * gscale_dev[0] * dloss/dlogits.                     */
int ustrun_dice_fwd(const float* logits, const void* target, int target_is_i64, int target_per_pixel, const float* mask,
                    int mask_per_pixel, int N, int K, int HW, int act, int multi, const float* weight_host, float* out,
                    float* partials, int64_t partials_bytes, ustrun_stream_t s);
int ustrun_dice_bwd(const float* logits, const void* target, int target_is_i64, int target_per_pixel, const float* mask,
                    int mask_per_pixel, int N, int K, int HW, int act, int multi, const float* weight_host, const float* sums,
                    const float* gscale_dev, float gscale, float* dlogits, ustrun_stream_t s);

/* ---- per-sample binary overlap counts for the numpy Dice of utils/metrics.py:114-146 ---------
 * counts[n][c] = {|pred|, |gt|, |pred & gt|} with pred/gt binarised as (x == cls[c]) or (x != 0). */
int ustrun_dice_counts(const void* pred, const void* gt, int pred_is_i64, int gt_is_i64, int N,
                       int K, int HW, int by_class, int32_t* counts, ustrun_stream_t s);

/* ---- SGD(momentum, weight decay) + EMA teacher over flat buffers: train.py:512,848,87-93 ------
 * g += wd*p; v = first ? g : mu*v + g; p -= lr*v; t = alpha*t + (1-alpha)*p                    */
/* ---- dynamic loss scale of the IEEE-half path: torch.cuda.amp.GradScaler as train.py:552,842-845 uses it, on the device.
 * amp_state = 8 floats {scale, scale, growth_tracker, found_inf, steps skipped so far, steps seen so far, 0, 0} (initialise
 * to {65536, 65536, 0, 0, 0, 0, 0, 0}); its address is what
 * ustrun_seg_loss_bwd / ustrun_dice_bwd take as gscale_dev, which scales the backward.  Per step, after the gradient
 * all-reduce: ustrun_amp_check (found_inf = 1 if any element of g is not finite) -> ustrun_sgd_ema_scaled (gradient x
 * grad_scale / scale; found_inf: parameters and momentum untouched = GradScaler.step skipping optimizer.step, the EMA line
 * still runs) -> ustrun_amp_update (scale x backoff after a skipped step, x growth after growth_interval clean ones --
 * GradScaler defaults 2, 0.5, 2000 --, found_inf cleared).  Nothing here waits for the host.                           */
int ustrun_amp_check(const float* g, int64_t n, float* amp_state, ustrun_stream_t s);
int ustrun_sgd_ema_scaled(float* p, const float* g, float* v, float* t, int64_t n, float lr, float mu, float wd,
                          int first, float alpha, float grad_scale, const float* amp_state, ustrun_stream_t s);
int ustrun_amp_update(float* amp_state, float growth_factor, float backoff_factor, int growth_interval, ustrun_stream_t s);
int ustrun_sgd_ema(float* p, const float* g, float* v, float* t, int64_t n, float lr, float mu,
                   float wd, int first, float alpha, float grad_scale, ustrun_stream_t s);

/* ---- whole-network plan: replaces UNet.forward / autograd backward, unet_model.py:25-39 -------
 * The plan is a host-side description (shapes + parameter pointers); the caller owns all device
 * memory: params, packed-weight arena, forward workspace (kept for backward) and scratch.      */
typedef struct ustrun_unet_desc {
    int32_t N, C, H, W, K;       /* batch, input channels, extent, classes                    */
    int32_t base;                /* 64 in the reference (unet_model.py:13)                    */
    int32_t dtype;               /* storage dtype of activations                              */
    int32_t train;               /* batch statistics (1) or running statistics (0)            */
    int32_t update_running;      /* update BN running buffers (train mode)                    */
    int32_t groups;              /* > 1: N = groups * n images of `groups` independent forward passes batched into one
                                  * call; BatchNorm statistics (and running-buffer updates, in order) are per pass  */
    int32_t tail;                /* > 0: N = groups * n + tail: `tail` (< n) more images follow as one further, shorter forward
                                  * pass (its own BatchNorm statistics, its running-buffer update after the others') whose
                                  * logits nobody reads: the head skips it and ustrun_unet_backward covers the first
                                  * groups * n images only (train.py:740: the low-quality sample's forward, Q2)      */
    int32_t lead;                /* > 0: the first `lead` (< groups) passes carry no gradient: the forward treats them as any other pass
                                  * (statistics, running-buffer updates in order, logits), ustrun_unet_backward starts behind them --
                                  * dlogits keeps the forward's layout (N - tail images; the leading passes' rows are ignored).
                                  * train.py:668: the student's forward on the weak view, whose arg-max alone is used (Q3), runs as
                                  * the first pass of the call that carries the four gradient passes of :699-702          */
    float   momentum, eps;
    /* parameters/buffers, torch layouts, in state_dict order:                 */
    const float* conv_w[18];     /* inc.0, inc.3, down1..4 (.0,.3), up1..4.conv (.0,.3)        */
    const float* bn_w[18];  const float* bn_b[18];
    float* bn_rm[18];       float* bn_rv[18];   int64_t* bn_nbt[18];
    const float* up_w[4];   const float* up_b[4];
    const float* head_w;    const float* head_b;
    void* packed;                /* packed-weight arena, ustrun_unet_packed_bytes()           */
    int32_t bilinear;            /* 1: the reference's UNet(bilinear=True) (unet_model.py:17-22, unet_parts.py:48-51): Up upsamples its
                                  * input 2x (bilinear, align_corners=True) instead of a ConvTranspose2d -- no up_w / up_b, down4 and the
                                  * decoder on the halved channel plan (DoubleConv(in, out, in // 2)), 56 parameter tensors           */
} ustrun_unet_desc_t;

int64_t ustrun_unet_packed_bytes(const ustrun_unet_desc_t* d);
int64_t ustrun_unet_fwd_workspace_bytes(const ustrun_unet_desc_t* d);
int64_t ustrun_unet_bwd_scratch_bytes(const ustrun_unet_desc_t* d);
int ustrun_unet_pack(const ustrun_unet_desc_t* d, ustrun_stream_t s);
/* x NCHW f32 [N,C,H,W] -> logits NCHW f32 [N,K,H,W]; feat (optional) NCHW f32 [N,base,H,W]  */
int ustrun_unet_forward(const ustrun_unet_desc_t* d, const float* x, float* logits, float* feat,
                        void* workspace, ustrun_stream_t s);
/* grads[] follow the parameter order of the reference's model.parameters() (64 tensors for
 * bilinear=False, 56 for bilinear=True): accumulate != 0 adds into them.                     */
int ustrun_unet_backward(const ustrun_unet_desc_t* d, const float* x, const float* dlogits,
                         void* workspace, void* scratch, float* const* grads, int accumulate,
                         ustrun_stream_t s);
/* the same in pieces sharing one scratch: part 1 = head + decoder (afterwards the gradients of up1..up4 and outc,
 * the contiguous tail of the parameter order, are final: their all-reduce can start), part 2 = encoder -- or part 3 =
 * down4 (parameters 24..29 final afterwards) followed by part 4 = down3..inc; part 0 = all */
int ustrun_unet_backward_part(const ustrun_unet_desc_t* d, const float* x, const float* dlogits,
                              void* workspace, void* scratch, float* const* grads, int accumulate, int part,
                              ustrun_stream_t s);

/* ---- DeepLabV2-ResNet operators (reference networks/deeplabv2.py:10-33, networks/backbone/resnet.py:55-176; SURVEY.md 8f
 * row 4), forward and backward.  Activations NHWC in the compute dtype like the U-Net's.                                   */
/* torch conv weight [Cout][Cin][kh*kw] -> the forward pack of `taps` slices (bf16 [tap][Cin/8][Cout][8], f32 [tap][Cin][Cout]);
 * w_fwd holds ustrun_pack_conv_elems() elements of the compute dtype                                                      */
int ustrun_pack_conv(const float* w, int Cout, int Cin, int taps, void* w_fwd, int dtype, ustrun_stream_t s);
int64_t ustrun_pack_conv_elems(int Cout, int Cin, int taps);
/* k x k convolution (k = 1, 3, 5, 7), stride 1 or 2, dilation d, padding d * (k / 2) (nn.Conv2d of resnet.py:8-15,124 and
 * deeplabv2.py:15-17) of the concatenated sources -> y [N, Ho, Wo, Cout] (compute dtype, or f32 when y_f32); bias optional;
 * stat (optional): BatchNorm-statistics partial rows [rows][2][Cout] -- the buffer must hold ustrun_conv_mtiles(N, Ho, Wo,
 * Cout) rows (which kernel serves the launch decides how many are written), *stat_rows receives the count written       */
int ustrun_conv2d_fwd(const ustrun_src_t* srcs, int nsrc, const void* w_fwd, const float* bias, int N, int Ho, int Wo, int Cout,
                      int k, int stride, int dilation, void* y, int y_f32, float* stat, int* stat_rows, int dtype, ustrun_stream_t s);
/* a k x k convolution over FEW input channels (the 7x7 / stride-2 stem, resnet.py:124) as `nrows` row segments of a source the
 * caller has zero-padded and describes window-wise: src->C = the k * Cin contiguous NHWC elements of one kernel row rounded up
 * to a multiple of 8, src->sW = Cin (pixel stride in elements), src->sH / sN = the padded tensor's strides, src->H / W = the
 * rows / window starts available; output pixel (y, x), segment s reads the window at (stride*y + s, stride*x); weights packed
 * by ustrun_pack_conv as [Cout][src->C]["taps" = nrows]                                                                    */
int ustrun_conv_rowwin_fwd(const ustrun_src_t* src, const void* w_fwd, int N, int Ho, int Wo, int Cout, int nrows, int stride, void* y,
                           float* stat, int* stat_rows, int dtype, ustrun_stream_t s);
/* MaxPool2d(3, stride 2, padding 1) of relu(y * scale + shift) (resnet.py:127): [N,H,W,C] -> [N,(H+1)/2,(W+1)/2,C]       */
int ustrun_maxpool3x3s2(const void* y, const float* scale, const float* shift, int N, int H, int W, int C, void* out, int dtype,
                        ustrun_stream_t s);
/* the bottleneck's join (resnet.py:97-103): out = relu(y * scale + shift + identity), identity = idn * iscale + ishift
 * (the downsample branch's BatchNorm) or idn itself when iscale is NULL                                                   */
int ustrun_bn_add_relu(const void* y, const float* scale, const float* shift, const void* idn, const float* iscale,
                       const float* ishift, int64_t npix, int C, void* out, int dtype, ustrun_stream_t s);
/* the classifier's dilated 3x3 branches (deeplabv2.py:15-17,26-28) evaluated as ONE 1x1 GEMM z [N,h,w,nrates*9*K] (column
 * (r*9+tap)*K+k = sum_c x[c] w_r[k][c][tap], from ustrun_conv2d_fwd with k = 1) followed by this shifted add:
 * out[N,h,w,K] = bias_sum[k] + sum_{r,tap} z[p + rates[r]*(tap/3-1, tap%3-1)][(r*9+tap)*K+k], zero outside the image;
 * rates: host array of nrates (<= 4) ints                                                                                  */
int ustrun_aspp_gather(const float* z, int N, int h, int w, int K, int nrates, const int* host_rates, const float* bias_sum,
                       float* out, ustrun_stream_t s);
/* sum of up to four f32 NHWC maps [N,h,w,K] (the classifier's dilated branches, deeplabv2.py:26-28), resized bilinearly with
 * align_corners=True to NCHW f32 [N,K,H,W] (deeplabv2.py:30)                                                               */
int ustrun_sum_resize_bilinear(const float* const* maps, int nmaps, int N, int h, int w, int K, int H, int W, float* out,
                               ustrun_stream_t s);

/* the same row windows written out as GEMM rows: out [N*Ho*Wo][k_padded] in the compute dtype, column s * src->C + j = element j of
 * the window of kernel row s, columns >= nrows * src->C zero (k_padded % 8 == 0).  With it the stem is ONE 1x1 convolution over
 * k_padded "channels" (ustrun_conv2d_fwd / ustrun_conv2d_wgrad with k = 1) on the fast GEMM kernels                             */
int ustrun_rowwin_patches(const ustrun_src_t* src, int N, int Ho, int Wo, int nrows, int stride, int k_padded, void* out, int dtype,
                          ustrun_stream_t s);

/* ---- DeepLabV2-ResNet backward (autograd of the modules above).  Input gradients of the stride-1 convolutions are
 * ustrun_conv2d_fwd launches over dy with the flipped / transposed weights (packed by ustrun_pack_conv from
 * w.flip(2,3).transpose(0,1)); BatchNorm backward is ustrun_bn_bwd_reduce / _apply (a BatchNorm that is NOT followed by a ReLU
 * -- bn3, the projection shortcut's -- passes scale = 0, shift = 1 so that the kernels' ReLU mask is all ones).                */
/* weight gradient of ustrun_conv2d_fwd's convolution (k = 1 or 3): dw[Cout][Cin][k][k] (torch layout, f32); partials: scratch
 * of ustrun_wgrad_partials_bytes(k*k, Cin, Cout, N*Ho*Wo) bytes                                                              */
int ustrun_conv2d_wgrad(const ustrun_src_t* srcs, int nsrc, const void* dy, int N, int Ho, int Wo, int Cout, int k, int stride,
                        int dilation, float* dw, int accumulate, float* partials, int64_t partials_bytes, int dtype, ustrun_stream_t s);
/* weight gradient of ustrun_conv_rowwin_fwd: dw[Cout][src->C][nrows] (the layout its weights were packed from)               */
int ustrun_conv_rowwin_wgrad(const ustrun_src_t* src, const void* dy, int N, int Ho, int Wo, int Cout, int nrows, int stride, float* dw,
                             int accumulate, float* partials, int64_t partials_bytes, int dtype, ustrun_stream_t s);
/* the input gradient of a 1x1, stride-1 convolution with what follows it in a bottleneck's backward (resnet.py:87-105 under
 * autograd), in one launch: g[N,H,W,Cin] = (dy (*) w_dgrad + add) * (ref > 0) -- what ustrun_conv2d_fwd over dy followed by
 * ustrun_relu_bwd_add computes, rounding included (add / ref may be NULL as there: conv1's input gradient and the residual join)
 * -- and, with y != NULL, the BatchNorm-backward sums of the layer whose OUTPUT gradient g is, as *stat_rows rows of [2][Cin] =
 * {sum(g mask), sum(g mask y)} for ustrun_bn_bwd_finalize_stat (stat: ustrun_conv_mtiles(N, H, W, Cin) rows): mask = y scale +
 * shift > 0 (a BatchNorm + ReLU layer: conv3's input gradient feeding bn2), or all ones with scale = shift = NULL (a BatchNorm no
 * ReLU follows: the previous block's bn3, resnet.py:96, behind the join).  *fused = 0 and NO launch when the shape is not covered
 * (16-bit storage, Cin % 128 == 0, Cout % 64 == 0): the caller then runs the separate calls.                                   */
int ustrun_conv1x1_dgrad_join(const void* dy, const void* w_dgrad, int N, int H, int W, int Cout, int Cin, const void* add,
                              const void* ref, void* g, const void* y, const float* scale, const float* shift, float* stat,
                              int* stat_rows, int* fused, int dtype, ustrun_stream_t s);
/* space-to-batch for a dilated 3x3 convolution's weight gradient: out[(n r + a) r + b][i][j][c] = act(src[n][i r + a][j r + b][c]),
 * every residue class (a, b) of the r x r pixel lattice as an image of its own, all padded with ZEROS to ceil(H / r) x ceil(W / r)
 * (out: N r r images); act = relu(x scale + shift) when the source carries constants, else a copy (src: plain contiguous NHWC,
 * 16-bit storage, C % 8 == 0).  With BOTH operands laid out this way ustrun_conv2d_wgrad(k = 3, dilation = 1) over N r r images
 * returns the weight gradient of the dilation-r convolution (networks/backbone/resnet.py:92-93 with replace_stride_with_dilation)
 * on the all-taps kernel.                                                                                                        */
int ustrun_space_to_batch(const ustrun_src_t* src, int N, int r, void* out, int dtype, ustrun_stream_t s);
/* the input gradient of a (dilated) 3x3, stride-1 convolution as ustrun_conv2d_fwd forms it (a convolution of dy with the pack of
 * w.flip(2,3).transpose(0,1)) that also forms the BatchNorm-backward sums of the BatchNorm + ReLU layer whose output gradient da
 * is -- conv2's input gradient feeding bn1 (resnet.py:89-93 under autograd): rows and mask as ustrun_conv3x3_dgrad_bnsum.
 * *stat_rows = 0 and NO launch when the fused epilogue does not cover the shape (16-bit storage, Cin % 128 == 0, dilation 1 / 2 / 4). */
int ustrun_conv2d_dgrad_bnsum(const void* dy, const void* w_flipped, int N, int H, int W, int Cout, int Cin, int dilation, void* da,
                              const void* y, const float* scale, const float* shift, float* stat, int* stat_rows, int dtype,
                              ustrun_stream_t s);
/* gradient of the bottleneck's join with respect to its pre-ReLU sum: g = (a + b) * (ref > 0) over n elements (n % 4 == 0);
 * b = NULL: one contribution, ref = NULL: no ReLU (the max-pool output feeding layer1)                                       */
int ustrun_relu_bwd_add(const void* a, const void* b, const void* ref, int64_t n, void* g, int dtype, ustrun_stream_t s);
/* MaxPool2d(3, 2, 1) backward: dp [N,(H+1)/2,(W+1)/2,C] -> da [N,H,W,C] (gradient of the ACTIVATED tensor relu(y*scale+shift),
 * whose ReLU the following ustrun_bn_bwd_* applies); the first maximum of a window wins (torch's rule); gather, fixed order   */
int ustrun_maxpool3x3s2_bwd(const void* dp, const void* y, const float* scale, const float* shift, int N, int H, int W, int C, void* da,
                            int dtype, ustrun_stream_t s);
/* adjoint of ustrun_sum_resize_bilinear for one map: dout NCHW f32 [N,K,H,W] -> dlow NHWC f32 [N,h,w,K]                      */
int ustrun_sum_resize_bilinear_bwd(const float* dout, int N, int h, int w, int K, int H, int W, float* dlow, ustrun_stream_t s);
/* adjoint of ustrun_aspp_gather: dz [N,h,w,zc_padded] in the compute dtype (columns >= nrates*9*K are zero)                  */
int ustrun_aspp_scatter(const float* dlow, int N, int h, int w, int K, int nrates, const int* host_rates, int zc_padded, void* dz,
                        int dtype, ustrun_stream_t s);
/* out[c] (+)= sum over rows of x[row][c] (f32, f64 accumulation, fixed order): the classifier biases' gradient              */
int ustrun_colsum(const float* x, int64_t rows, int C, float* out, int accumulate, ustrun_stream_t s);

/* test aid: tile configuration of the last halo-tiled bf16 3x3 convolution launched by this process, as
 * TH<<24 | TW<<16 | BN<<8 | MI<<4 | NT<<2 | POOL<<1 | XF (0 before any) -- lets a parity test assert that its shape
 * reached the production tile it was written for                                                   */
int ustrun_debug_last_conv_variant(void);
/* the 1x1 / ConvTranspose GEMM kernel (convT_bf16.hip) reports 0x43540000 | (BN / 32) << 8 | (BK / 32) << 4 | MODE there
 * (MODE 0 ConvTranspose forward, 1 its input gradient, 2 a 1x1 convolution).
 * Same for the last weight-gradient launch: the one-tap-per-block kernel (wgrad_tap_bf16.hip) reports
 * 0x54000000 | (TM / 64) << 20 | (TN / 64) << 16 | loader << 12 | ksplit (loader 2 = pixel-linear 1x1, 1 = one shifted tap,
 * 0 = k x k taps); the all-taps halo kernel 0x48000000 | build << 20 | ksplit (build 0 = round-2 kernel, 1 = buffer-addressed
 * transfers, 2 = two wave groups in opposite phases); 0 before any / for the generic kernels.
 * ustrun_debug_last_conv_variant of the ConvTranspose / 1x1 GEMM kernel: 0x43540000 | (weights through registers ? 0x1000 : 0) |
 * (BN / 32) << 8 | (BK / 32) << 4 | mode (0 forward, 1 input gradient, 2 plain 1x1)                              */
int ustrun_debug_last_wgrad_variant(void);
/* the last BatchNorm-backward launch (ustrun_bn_bwd_reduce / _apply and the whole-network backward):
 * 0x424E0000 | pass << 8 (0 reduce, 1 apply) | element bytes << 4 | even-sized pooled windows << 2 | pooled << 1 |
 * eight channels (16 bytes) per lane; 0 before any                                                              */
int ustrun_debug_last_bn_variant(void);
/* test aid, host only (no launch, no device access): the number of BatchNorm-statistics rows the kernel that would serve a
 * k x k convolution of a dense NHWC source [N, Cin] -> [N, Ho, Wo, Cout] writes.  tests/test_host_logic.py sweeps shapes
 * with it: the count must never exceed ustrun_conv_mtiles(N, Ho, Wo, Cout), which is what callers allocate           */
int ustrun_debug_conv_stat_rows(int N, int Ho, int Wo, int Cin, int Cout, int k, int stride, int dilation, int pooled, int dtype);
/* test / tuning aid: process-wide kernel-selection flags, returns the previous value.  bit 0: run the 64 -> 64 channel
 * full-resolution convolutions on the halo-tiled kernel instead of the weight-stationary row-streaming one (A/B timing
 * inside one process); bit 1: the four-wave build of the streaming kernel for every launch, bit 2: the eight-wave build for every
 * launch; bit 4 (16): no consumer / producer build (then eight waves for plain sources without statistics, four waves otherwise),
 * bit 5 (32): the consumer / producer build for every launch -- it is the default since round 3.
 * bit 3 (8): halo-tiled kernel with weight tiles in LDS (round 2) instead of weights through registers;
 * bit 6 (64): all-taps weight gradient with per-item pointers (round 2) instead of buffer-addressed transfers;
 * bit 7 (128): that kernel with extra LDS so that one block fits a CU (occupancy experiment);
 * bit 8 (256): no two-group weight-gradient kernel; bit 9 (512): ConvTranspose / 1x1 GEMM with weight tiles in LDS (round 2).
 * bits 10-11: force the halo kernel's tile in the 128-column case (1: 8 x 32 px, 2: 16 x 16, 3: 8 x 16; 0: chosen by padding).
 * bit 12 (4096): BatchNorm backward (plain, bf16) on the 4-channel-per-lane kernels instead of the 8-channel ones.
 * The last-variant code of the streaming kernel is 0x57530000 | (eight waves ? 0x100 : consumer / producer ? 0x200 : 0) | XF.
 * The halo-tiled kernel runs plain sources (the input gradients) on its 256-pixel x 128-channel tiles on v_mfma_f32_16x16x32
 *   (ustrun_debug_last_conv_variant then carries bit 7, 0x80); bit 15 (32768): on every tile; bit 21 (2097152): on none.
 * bit 14 (16384): first convolution (C <= 4 -> 64) on the tile-per-block kernel of rounds 1-3 instead of the streaming one.
 * bits 16-20: layer + 1 at which ustrun_unet_backward stops early (tests/diag_grad.py; 0: runs through).
 * bit 22 (4194304): ustrun_unet_forward finalizes each layer's BatchNorm statistics in ONE launch (last-ticket pattern) instead of
 *   two; bit-identical results, measured no faster (profiles/r04_ab_bn_fused_finalize.log), off by default.
 * bit 23 (8388608): ustrun_unet_backward runs the BatchNorm-backward reduce pass of the layer under the head (rounds 1-3)
 *   instead of taking its two sums from the head kernel's partial rows.
 * bit 24 (16777216), ENVIRONMENT ONLY (USTRUN_DEBUG_FLAGS at load; ustrun_debug_flags does not change it): the decoder reads
 *   its skip operands through BatchNorm + ReLU on load (rounds 1-3) instead of the activation tensors ustrun_pool_act2
 *   materialises -- the switch shapes the workspace that ustrun_unet_forward and _backward share, so it must not differ
 *   between the threads that call them.
 * bit 25 (33554432): ustrun_unet_backward never asks an input gradient for the BatchNorm-backward sums of the layer it feeds
 *   (ustrun_conv3x3_dgrad_bnsum): every BatchNorm backward runs its reduce pass, as in rounds 1-3.
 * bit 26 (67108864), ENVIRONMENT ONLY like bit 24: the second convolution of every DoubleConv reads its operand through
 *   BatchNorm + ReLU on load (rounds 1-3) instead of the activation ustrun_act16 writes out on the levels from 256 channels.
 * bit 27 (134217728): ustrun_head_fwd on its generic 16-bit kernel instead of the 64-channel one.
 * bit 13 (8192): 64-output-channel 3x3 layers on >= 32-wide maps on the 16 x 32-pixel tile (one block per CU; A/B runs).
 * A caller that runs a forward and its backward on different threads sets the same value on both (the Python host does:
 * ustrun/engine.py hands the forward's flags to autograd's backward thread).
 * The value is PER CALLING THREAD (as are the last-variant codes and the stamp buffer below): a thread that sets it changes
 * kernel selection for the launches it issues itself and for nobody else, so the library keeps no process-wide mutable
 * state.  Every thread starts from the value the environment variable USTRUN_DEBUG_FLAGS had when the library was loaded
 * (read once at load, never on a launch path).                                                                    */
int ustrun_debug_flags(int flags);
/* a second word of the same kind (the first is full), same rules: per calling thread, returns the previous value, starts from the
 * environment variable USTRUN_DEBUG_FLAGS2 as read at load.
 * bit 0 (1): the halo-tiled 3x3 kernel never runs on its linear tiles (round 5: maps with 18 / 24 / 36 / 72-pixel rows; DESIGN.md
 *   10.9) -- the rectangular tile of the padding rule instead (A/B runs, and the tests that pin those tiles).
 * bit 1 (2): dtype USTRUN_F32X3 launches its halo-tiled convolution and its weight gradient once per pass of a batched call, as
 *   before they took the pass's constants per image (A/B runs).
 * bit 2 (4): the 64 -> 64 streaming kernel keeps its uniform strip split (round 6: the flat plan gives every block the same number
 *   of row steps at any image count) -- A/B runs.
 * bit 3 (8): ustrun_conv1x1_dgrad_join never fuses (*fused = 0) -- A/B runs and the tests of the unfused path.                 */
int ustrun_debug_flags2(int flags);
/* Operator-level declaration (per calling thread; returns the previous value): while `allow` is nonzero, the convolution entry
 * points accept a batch whose LAST pass is shorter than ustrun_src_t::gN (N % gN != 0 -- the caller then owns a constants table of
 * ceil(N / gN) entries and splits the statistics rows accordingly).  Otherwise N % gN != 0 is an error ("inconsistent pass
 * groups"): a mis-sized batch is not a tail.  ustrun_unet_forward / _backward set it themselves from ustrun_unet_desc_t::tail.   */
int ustrun_short_last_pass(int allow);
/* development aid: while a device buffer is set here (per calling thread), the 64 -> 64 streaming kernel and the two-group
 * all-taps weight gradient run their phase-stamping diagnostic builds and write per-wave cycle sums there as
 * [workgroup][8 waves][8] u64 = 64 u64 per workgroup (tools/ab_ws64.py --diag, tools/diag_wgrad.py).  n_u64 = the buffer's
 * length: a launch whose grid needs more (grid x 64) fails with an error instead of writing past the end.  NULL restores
 * the product kernels                                                                                              */
int ustrun_debug_buffer(void* device_u64, int64_t n_u64);
/* measurement aid (tools/clock_probe.py, bench.py): `blocks` workgroups each write 4 u64 {s_memtime (shader-clock ticks),
 * s_memrealtime (100 MHz ticks), XCC id, HW_ID} to device_u64[blocks][4].  Two probes on one stream around a series of
 * launches give the shader clock the chip held meanwhile: d(memtime) / d(memrealtime) x 100 MHz, per XCD.          */
int ustrun_debug_clock_probe(void* device_u64, int blocks, ustrun_stream_t s);

/* ---- optional launch profiler (bench.py): HIP events recorded on the launch stream around every
 * implicit-GEMM (kind 0) / weight-gradient (kind 1) launch while enabled; collect synchronises on
 * the recorded events, returns the sums and resets that kind.                                   */
int ustrun_profile_enable(int on);
int ustrun_profile_collect(int kind, double* host_ms, double* host_flops, double* host_bytes,
                           int64_t* host_launches);
/* only != 0: time only the launches issued on stream s (a launch on a side stream overlaps the profiled stream's
 * kernels, its event pair would measure contention); only == 0: every stream (the default)      */
int ustrun_profile_stream(ustrun_stream_t s, int only);
/* one record per timed launch, in launch order, WITHOUT resetting (call before ustrun_profile_collect).  tag: the
 * whole-network calls label their launches -- 3x3 conv i (0..17, network order) forward = i, ConvTranspose j (0..3)
 * forward = 20 + j, the input gradients of the same layers + 100, the weight gradients + 200; -1 for operator-level
 * calls.  n = images in the launch.  flops / bytes = the launch's ALGORITHMIC cost (every stored input, output and
 * weight element touched once at its stored size).  Returns the number of records written, -1 on error.        */
typedef struct ustrun_prof_rec {
    int32_t kind, tag, n, pad;
    double  ms, flops, bytes;
} ustrun_prof_rec_t;
int64_t ustrun_profile_records(ustrun_prof_rec_t* host_out, int64_t max_records);

#ifdef __cplusplus
}
#endif
#endif /* USTRUN_H */
