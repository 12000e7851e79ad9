/* din_hip.h -- C ABI of libdin_hip.so: the MI355X (gfx950) kernels behind the DIN stage-2 hot path.
 *
 * The reference (JacobYuan7/DIN-Group-Activity-Recognition-Benchmark) is pure Python; its only native
 * boundary is the third-party `roi_align` torch extension.  This header therefore DEFINES the native
 * boundary a maintainer would bind (ctypes stub shown in INTEGRATION.md); each entry cites the reference
 * interface (file:line, relative to the reference root) whose arithmetic it replaces.
 *
 * Conventions
 *  - plain C types only; every pointer is a DEVICE pointer owned by the caller (PyTorch-ROCm allocates);
 *    the library never allocates, never frees, never synchronises, never throws.
 *  - every function returns 0 on success or a negative DIN_E_* code; din_last_error_string() gives the
 *    thread-local message.  `stream` is a hipStream_t passed as void* (NULL = default stream).
 *  - activations are NHWC ("pixel-major"): element (n,y,x,c) of a tensor with pixel stride `ld` and
 *    channel offset `coff` lives at ((n*H+y)*W+x)*ld + coff + c.  ld/coff let several producers write
 *    disjoint channel ranges of one buffer (torch.cat(dim=1) at infer_model.py:172 and in the Inception
 *    blocks costs nothing).
 *  - dtype: DIN_F32 (parity mode, fp32 storage + fp32 MFMA accumulate) or DIN_BF16 (throughput mode:
 *    bf16 storage, fp32 MFMA accumulate).  Everything after RoIAlign is always fp32.
 *  - stateless and re-entrant: safe from autograd worker threads, concurrently on different streams.  The
 *    library never reads the process environment: kernel-selection switches used by the tests and the
 *    tuning tools are process-wide OPTIONS set through din_set_option (unset = the shipped choice).
 */
#ifndef DIN_HIP_H
#define DIN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DIN_ABI_VERSION 9   /* 9: din_conv_wgrad_group (the weight gradients of several layers in one launch).   8: din_set_option / din_get_option replace every getenv() of the library (tests select kernels through the ABI; a stray environment variable can no longer change a launch).   7: din_conv_dgrad_x (a strided dgrad that carries the 1x1 / stride-1 dgrad of a sibling conv reading the same view).   6: batch-statistics BatchNorm is deterministic: din_bn_stats / din_bn_bwd_stats write per-workgroup fp64 slabs into a workspace (din_bn_workspace), din_bn_finalize / din_bn_reduce add them in slab order -- no atomics, nothing for the caller to zero.   5: dropout seeds take an optional device-side offset (din_layernorm_*, din_act_dropout_*), din_counter_add, din_conv1x1_wgrad_multi.   2: din_walk_* take plain / clamp / n_per_clip; + bn, mask_actors.  3: din_roi_align_* take the box grid and a crop channel range.  4: din_conv_desc.in_u8, context-encoding entry points */

enum { DIN_F32 = 0, DIN_BF16 = 1 };

enum {
    DIN_OK = 0,
    DIN_E_ARG = -1,      /* bad argument (shape, alignment, null pointer)            */
    DIN_E_LAUNCH = -2,   /* hipLaunch / hipGetLastError failure                       */
    DIN_E_WORKSPACE = -3,/* caller-provided workspace too small                       */
    DIN_E_UNSUPPORTED = -4
};

int din_abi_version(void);
const char* din_last_error_string(void);
/* name of the device code object's target ("gfx950"); lets the host fail loudly on a wrong build */
const char* din_build_arch(void);
/* Process-wide tuning / test options ("DIN_CONV_HALO" = "2", ...; value NULL = unset).  They select between kernels that
 * compute the same result (tests cover every instantiation through them; tools/ A/B them); production code never sets any.
 * Thread-safe; a launch reads each option it consults exactly once.  The reference has no counterpart (its only knobs are
 * the Config fields, config.py:10-104, which stay in the Python layer). */
int din_set_option(const char* name, const char* value);
/* copies the current value into buf (empty string when unset); returns 1 if set, 0 if unset, negative on error */
int din_get_option(const char* name, char* buf, int buf_bytes);

/* ------------------------------------------------------------------------------------------------
 * Row P  prep_images  (utils.py:8-19): y = ((x/255) - 0.5) * 2, same three fp32 roundings.
 * ---------------------------------------------------------------------------------------------- */
/* NCHW fp32 -> NCHW fp32, API-parity form of utils.prep_images */
int din_prep_images_f32(const float* in, float* out, int64_t n, void* stream);
/* fused loader for the backbone: NCHW (uint8 or fp32, values 0..255) -> normalised NHWC with the channel
 * dimension zero-padded to `cpad` (4 for DIN_F32, 8 for DIN_BF16) so conv1 reads 16-byte pixels.      */
int din_prep_images_nhwc(const void* in, int in_is_u8, void* out, int out_dtype,
                         int nb, int h, int w, int cpad, void* stream);
/* backward of the loader is never needed (images carry no gradient: train_net_dynamic.py:175). */

/* ------------------------------------------------------------------------------------------------
 * Rows V, I, E, L, D1  dense contractions: implicit-GEMM convolution on MFMA.
 * Replaces torch.nn.Conv2d / nn.Linear arithmetic reached from backbone/backbone.py:44-99,
 * infer_model.py:184 (fc_emb_1), :190 (point_conv), :226 (fc_activities),
 * dynamic_infer_module.py:149 (hidden_weight), :191 (p_conv), :195 (scale_conv).
 * A Linear layer is the 1x1 case with NB=1, H=1, W=rows.
 * ---------------------------------------------------------------------------------------------- */
typedef struct din_conv_desc {
    int32_t nb, h, w, cin;          /* input  tensor [nb,h,w,cin], pixel stride ldi, channel offset cioff   */
    int32_t oh, ow, cout;           /* output tensor [nb,oh,ow,cout], pixel stride ldo, channel offset cooff */
    int32_t kh, kw, sh, sw, ph, pw, dh, dw;
    int32_t ldi, cioff, ldo, cooff;
    int32_t dtype;                  /* DIN_F32 / DIN_BF16 : storage type of in, out and packed weights       */
    int32_t in_u8;                  /* 1: `in` is the raw uint8 clip batch [nb][3][h][w] (volleyball.py:223-275 frames before
                                       utils.prep_images) and the image layer normalises on load -- (x/255 - 0.5)*2 in the reference's
                                       three fp32 roundings, utils.py:8-19 -- instead of reading a prepared NHWC tensor; ldi / cioff are
                                       ignored.  Only din_conv_fwd / din_conv_wgrad, only where din_conv_accepts_u8() says so. */
} din_conv_desc;

enum {
    DIN_CONV_BIAS = 1,   /* out += bias[cout]  (fp32 vector)                                    */
    DIN_CONV_RELU = 2,   /* out = max(out, 0)                                                    */
    DIN_CONV_ACCUM = 4,  /* dgrad only: out += result (several consumers of one tensor)          */
    DIN_CONV_MASK = 8    /* dgrad only: result *= (mask > 0)  -- fused ReLU backward, mask has the
                            layout of the dgrad output (pixel stride ldm, channel offset moff)    */
};

/* number of elements (of desc->dtype) of the packed filter bank for fwd (transposed=0) / dgrad (=1) */
int64_t din_conv_packed_elems(const din_conv_desc* d, int transposed);
/* w: reference layout [cout][cin][kh][kw] fp32.  scale (nullable, [cout] fp32) is folded into the filters
 * (BatchNorm-eval: backbone.py BasicConv2d).  transposed=0 -> [cout_pad][(r,s,ci)], =1 -> [cin_pad][(r,s,co)] */
int din_conv_pack_weights(const din_conv_desc* d, const float* w, const float* scale, void* wpk,
                          int transposed, void* stream);
/* All filter banks of a backbone in ONE launch.  din_conv_pack_desc fills one table entry on the host (same layout rules as
 * din_conv_pack_weights); the caller uploads the table once and then calls din_conv_pack_multi every step with, per workgroup, the
 * bank it packs (layer_of) and which chunk_elems-sized chunk of it (chunk_index).  All three arrays live on the device. */
typedef struct din_pack_desc {
    uint64_t w, scale, out;          /* device addresses: fp32 [cout][cin][kh][kw] filters, fp32 [cout] scale (0: none), packed bank */
    int32_t cout, cin, kh, kw;
    int32_t rows, rows_pad, inner, inner_pad, kelems, transposed, dtype;
    int32_t reserved;
} din_pack_desc;
int din_conv_pack_desc(const din_conv_desc* d, const float* w, const float* scale, void* wpk, int transposed, din_pack_desc* out);
int din_conv_pack_multi(const din_pack_desc* table, const int32_t* layer_of, const int32_t* chunk_index, int nblocks,
                        int chunk_elems, void* stream);
/* Forward conv with TWO destinations: produced channels [0, csplit) go to `out` (desc->ldo / desc->cooff), channels [csplit, desc->cout) to
 * `out2` (pixel stride ldo2, channel offset cooff2).  For sibling convs that read the same tensor (torchvision InceptionA: branch1x1,
 * branch5x5_1, branch3x3dbl_1 -- backbone.py MyInception_v3 via torchvision inception.py InceptionA.forward): one launch over the
 * concatenated filter bank reads the input once and runs on a wide tile.  wpk = the banks of the siblings packed row after row (same cin,
 * kh, kw); bias = their shifts concatenated.  craw > 0: channels >= craw are stored raw (no bias, no ReLU) -- the branch_pool conv whose
 * bias + ReLU follow its average pool.  Not available for shapes that need split-K (returns DIN_E_ARG: launch them separately). */
int din_conv_fwd2(const din_conv_desc* d, const void* in, const void* wpk, const float* bias, void* out, void* out2, int ldo2, int cooff2,
                  int csplit, int craw, int flags, void* workspace, int64_t workspace_bytes, void* stream);
/* which tile variant the planner picks (which: 0 fwd, 1 dgrad -> pixels x filters of conv_gather_*_kernel; 2 wgrad -> filter rows x
 * k columns of conv_wgrad_*_kernel, bn = 1000 + k columns for conv_wgrad_ring_kernel; bm = 0 -> the stationary-filter stem kernels
 * conv_small_kernel / conv_wgrad_small_kernel with bn filters; bm = 1 -> conv_halo_kernel; bm = 2 -> conv_gather_pipe_kernel; bm = 4 ->
 * conv1x1_stream_kernel; bm = 5 -> conv1x1_regw_kernel, classes of bn = 192 filters resident in registers): lets a profiler-side caller
 * name the kernel a launch resolves to */
int din_conv_kernel_tile(const din_conv_desc* d, int which, int32_t* bm, int32_t* bn);
/* which instantiation of conv_gather_fast_kernel a fwd (0) / dgrad (1) launch resolves to: flags bit 0 = FASTK (scalar k-walk), bit 1 = 8 waves
 * (4 x 2) instead of 4 (2 x 2) -- so that a profiler-side caller can spell the exact kernel name rocprofv3 prints */
int din_conv_kernel_variant(const din_conv_desc* d, int which, int32_t* flags);
/* workspace bytes needed by fwd / dgrad / wgrad for this descriptor (split-K partial sums) */
int64_t din_conv_workspace_bytes(const din_conv_desc* d, int which /*0 fwd,1 dgrad,2 wgrad*/);

/* 1 when both the forward and the weight-gradient launch of this layer run the image-layer kernels that can read uint8 frames directly
 * (bf16, 3x3 stride 2, <= 8 input channels, <= 32 filters, >= 256 Ki output pixels); 0 otherwise (prepare the input with
 * din_prep_images_nhwc).  Host-only planning call. */
int din_conv_accepts_u8(const din_conv_desc* d);

int din_conv_fwd(const din_conv_desc* d, const void* in, const void* wpk, const float* bias, void* out,
                 int flags, void* workspace, int64_t workspace_bytes, void* stream);
/* dout has the OUTPUT geometry of d (pixel stride ldo/cooff); din gets the INPUT geometry (ldi/cioff).   */
int din_conv_dgrad(const din_conv_desc* d, const void* dout, const void* wpk_t, void* din,
                   const void* mask, int ldm, int moff, int flags,
                   void* workspace, int64_t workspace_bytes, void* stream);
/* Fused dgrad of up to four 1x1 / stride-1 convolutions that read the SAME tensor (the branch-entry convs of torchvision's
 * InceptionA/C blocks, backbone.py:61-77): din = sum_b dout_b . W_b^T computed as ONE contraction over the concatenated
 * channels, so the (write-bound) gradient tensor is written once instead of once per branch with read-modify-write.
 * Source b: dout_b [nb,h,w,cout_b] (pixel stride ldo, offset cooff), wpk_t = its bank packed with transposed=1.       */
typedef struct din_conv_src {
    const void* dout;
    const void* wpk_t;
    int32_t cout, ldo, cooff;
} din_conv_src;
int din_conv1x1_dgrad_multi(int nsrc, const din_conv_src* srcs, int dtype, int nb, int h, int w, int cin, int ldi,
                            int cioff, void* din, const void* mask, int ldm, int moff, int flags, void* stream);
/* din_conv_dgrad of a STRIDED conv plus the dgrad of ONE 1x1 / stride-1 conv `x` that reads the same input view, in the same launches:
 * din (op)= mask(conv^T(dout) + conv1x1^T(x->dout)).  torchvision's InceptionB (backbone.py MyInception_v3 Mixed_6a, through torchvision
 * inception.py InceptionB.forward: branch3x3 = 3x3 stride 2 and branch3x3dbl_1 = 1x1 both read the block input): the 1x1's 64 channels ride
 * as one more k-step of the four parity-class launches of the 3x3, so the 288-channel gradient map is read-modify-written once, not twice.
 * x->wpk_t = the 1x1's bank packed with transposed=1 (as for din_conv1x1_dgrad_multi).  Shapes the fused kernel does not serve (fp32,
 * stride 1, tiles other than 128 x 96, split-K; DIN_DGRAD_X=0) run as din_conv_dgrad followed by din_conv1x1_dgrad_multi with ACCUM --
 * same result up to the rounding of the intermediate sum.                                                                      */
int din_conv_dgrad_x(const din_conv_desc* d, const void* dout, const void* wpk_t, void* din, const void* mask, int ldm, int moff, int flags,
                     const din_conv_src* x, void* workspace, int64_t workspace_bytes, void* stream);
/* 1 when din_conv_dgrad_x serves this descriptor with the fused kernel, 0 when it runs the two-launch form.  Host-only planning call. */
int din_conv_dgrad_x_fused(const din_conv_desc* d);
/* dw: [cout][cin][kh][kw] fp32, overwritten (or += when accumulate!=0), multiplied by scale[cout] when scale
 * is given.  dbias (nullable) [cout] fp32 = column sums of dout.  wdot (nullable) [cout] fp32 =
 * <w[co,:], dw_raw[co,:]> (needs w) -- the BatchNorm-eval scale gradient.  accumulate bit 1 (value 2): dbias / wdot were zeroed by
 * the caller (they are accumulated into with atomics; a backbone zeroes one flat buffer for all its layers instead of 2 memsets per
 * layer).                                                                                                   */
int din_conv_wgrad(const din_conv_desc* d, const void* in, const void* dout, float* dw, float* dbias,
                   const float* scale, const float* w, float* wdot, int accumulate,
                   void* workspace, int64_t workspace_bytes, void* stream);

/* Weight gradients of 2..4 1x1 / stride-1 convs that read the SAME tensor view (the block-entry convs of an InceptionA block, reference
 * backbone/backbone.py:44-58 through torchvision: branch1x1, branch5x5_1 + branch3x3dbl_1, branch_pool) in ONE launch: the input is read
 * once instead of once per layer, every layer's dW block stays in registers of persistent workgroups (csrc/conv_wgrad_1x1.hip).
 * Source b: dout_b [pixels][ldo] at channel cooff (its gradient operand), dw_b [cout_b][cin] fp32, and din_conv_wgrad's optional
 * dbias / scale / w / wdot of that layer.  din_conv1x1_wgrad_multi_workspace returns the workspace bytes, or 0 when the group does not
 * fit the kernel (bf16, cin in {192, 256, 288}, couts multiples of 16 that can be dealt to two classes of <= 128 rows, >= 128K pixels):
 * the caller then runs din_conv_wgrad per layer.  accumulate as in din_conv_wgrad.                                              */
typedef struct din_conv_wsrc {
    const void* dout;
    float* dw;
    float* dbias;
    const float* scale;
    const float* w;
    float* wdot;
    int32_t cout, ldo, cooff;
} din_conv_wsrc;
int64_t din_conv1x1_wgrad_multi_workspace(int nsrc, const din_conv_wsrc* srcs, int dtype, int64_t pixels, int cin);
int din_conv1x1_wgrad_multi(int nsrc, const din_conv_wsrc* srcs, int dtype, int64_t pixels, int cin, int ldi, int cioff, const void* in,
                            int accumulate, void* workspace, int64_t workspace_bytes, void* stream);

/* The weight gradients of up to 16 LAYERS in one launch (autograd of the torch.nn.Conv2d layers of reference backbone/backbone.py:44-99 --
 * torchvision's InceptionC blocks hold ten 7-tap / 1x1 layers of equal map size each).  A weight-gradient launch of the pipelined kernel puts
 * one workgroup on every CU and each writes a full fp32 partial tile: slices x |dW| = ~50 MB per layer whatever the batch, read back by the
 * reduce launch.  Layers whose gradients are due at about the same time share one launch and its 256 workgroups instead: a layer of a group
 * of six is cut into 7 instead of 42 pixel slices (8 MB of partials), and the launch has one tail instead of six.
 * An item = the arguments of one din_conv_wgrad call.  din_conv_wgrad_group_key: 0 = this layer's gradient does not run on the pipelined
 * kernel (it cannot join a group); items with EQUAL non-zero keys may share a launch.  din_conv_wgrad_group_workspace: bytes for this group,
 * 0 = not a valid group (the caller runs din_conv_wgrad per item; din_conv_wgrad_group itself does the same when handed such a list, with
 * a workspace that must then hold the largest single-item need).  Results equal din_conv_wgrad's up to the order of the slice sums. */
typedef struct din_conv_wgrad_item {
    din_conv_desc desc;
    const void* in;
    const void* dout;
    float* dw;
    float* dbias;
    const float* scale;
    const float* w;
    float* wdot;
    int32_t accumulate, reserved;
} din_conv_wgrad_item;
int din_conv_wgrad_group_key(const din_conv_desc* d);
int64_t din_conv_wgrad_group_workspace(int n, const din_conv_wgrad_item* items);
int din_conv_wgrad_group(int n, const din_conv_wgrad_item* items, void* workspace, int64_t workspace_bytes, void* stream);

/* out[c] = sum over rows of g[row*ld + coff + c] (fp32 result; BatchNorm shift gradient of a conv whose epilogue ran in the pool) */
int din_colsum(const void* g, int dtype, int64_t rows, int c, int ld, int coff, float* out, void* stream);

/* BatchNorm(eval) folding helpers (torchvision BasicConv2d, eps=1e-3; train_net_dynamic.py:17-20 set_bn_eval)
 * scale = gamma*rsqrt(var+eps); shift = beta - mean*scale                                                */
int din_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                float* scale, float* shift, int c, void* stream);
/* dgamma = (wdot - dshift*mean)*rsqrt(var+eps); dbeta = dshift                                            */
int din_bn_fold_bwd(const float* wdot, const float* dshift, const float* mean, const float* var, float eps,
                    float* dgamma, float* dbeta, int c, void* stream);
/* The same two maps for ALL BatchNorm layers of a backbone in one launch each.  Device tables: ptrs [n][4] = addresses of
 * {gamma, beta, running_mean, running_var} per layer; offs [n+1] = prefix sums of the channel counts (offs[n] = total).  scale / shift /
 * wdot / dshift / dgamma / dbeta are flat [total] fp32 arrays, layer l owning [offs[l], offs[l+1]). */
int din_bn_fold_multi(const uint64_t* ptrs, const int32_t* offs, int n, int total, float eps, float* scale, float* shift,
                      void* stream);
int din_bn_fold_bwd_multi(const uint64_t* ptrs, const int32_t* offs, int n, int total, float eps, const float* wdot,
                          const float* dshift, float* dgamma, float* dbeta, void* stream);

/* BatchNorm with BATCH statistics: the reference's default for the Inception-v3 backbone in stage 2 -- model.train() without
 * set_bn_eval (train_net_dynamic.py:98-100,170-172; config.py:80) -> torch.nn.functional.batch_norm(training=True) inside torchvision's
 * BasicConv2d.  Views are [rows][c] with pixel stride ld and channel offset coff (elements; multiples of 4 fp32 / 8 bf16).
 * Workspace ws (fp64, din_bn_workspace(rows, c) bytes, NOT zeroed by the caller): ws[0..2c) = the reduced sums, then din_bn_parts(rows)
 * slabs of 2c partial sums, one per workgroup of the statistics kernel.  Everything is summed in a fixed order (fp64 per-thread
 * accumulators with exact products, lane-ordered LDS sum, slab-ordered reduce): the same input gives the same bits on every run.
 *   din_bn_stats     : slab[p][0..c) = sum over the rows of part p of (x - shift), slab[p][c..2c) = sum (x - shift)^2   (shift [c] nullable
 *                      = 0; fp64 accumulation makes it unnecessary -- kept for callers that have one; din_bn_finalize must get the SAME)
 *   din_bn_finalize  : ws[0..2c) = slab sums in slab order (nparts > 0; nparts == 0: ws[0..2c) already holds them); mean,
 *                      rstd = 1/sqrt(biased var + eps); a = gamma*rstd, b = beta - mean*a; running_mean/var (may be NULL) updated
 *                      in place with `momentum` and the unbiased variance, as torch does
 *   din_bn_apply     : y = a*x + b (relu != 0: max(.,0)) -- x and y may be views of different tensors
 *   din_bn_bwd_stats : ws[0..c) = sum gz, ws[c..2c) = sum gz*xhat, xhat = (x - mean)*rstd   (gz: gradient at the BN output,
 *                      already masked by the ReLU that follows); slabs + din_bn_reduce inside
 *   din_bn_reduce    : ws[0..2c) = sum of nparts slabs at ws + 2c, in slab order (for producers that fill the slabs themselves)
 *   din_bn_bwd_apply : dy = gamma*rstd*(gz - s1/rows - xhat*s2/rows); dgamma = s2, dbeta = s1   (sums = ws of din_bn_bwd_stats)        */
int din_bn_parts(int64_t rows);
int64_t din_bn_workspace(int64_t rows, int c);
int din_bn_stats(const void* x, int dtype, int64_t rows, int c, int ld, int coff, const float* shift, double* ws, void* stream);
int din_bn_reduce(double* ws, int nparts, int c, void* stream);
int din_bn_finalize(double* ws, int nparts, int64_t rows, int c, const float* gamma, const float* beta, float eps, float momentum,
                    float* running_mean, float* running_var, float* a, float* b, float* mean, float* rstd, const float* shift,
                    void* stream);
int din_bn_apply(const void* x, int dtype, int64_t rows, int c, int ldx, int cxoff, const float* a, const float* b, int relu,
                 void* y, int ldy, int cyoff, void* stream);
int din_bn_bwd_stats(const void* gz, int ldg, int cgoff, const void* x, int ldx, int cxoff, int dtype, int64_t rows, int c,
                     const float* mean, const float* rstd, double* ws, void* stream);
int din_bn_bwd_apply(const void* gz, int ldg, int cgoff, const void* x, int ldx, int cxoff, int dtype, int64_t rows, int c,
                     const float* gamma, const float* mean, const float* rstd, const double* sums, void* dy, int ldy, int cyoff,
                     float* dgamma, float* dbeta, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Pools / resize (backbone.py:51,57 max_pool2d; torchvision InceptionA/C avg_pool2d(3,1,1), InceptionB
 * max_pool2d(3,2); vgg.features MaxPool2d(2,2); infer_model.py:169 F.interpolate bilinear align_corners)
 * ---------------------------------------------------------------------------------------------- */
typedef struct din_pool_desc {
    int32_t nb, h, w, c, oh, ow;
    int32_t k, stride, pad;
    int32_t ldi, cioff, ldo, cooff;
    int32_t dtype;
} din_pool_desc;
/* argmax (nullable): uint8 [nb,oh,ow,c] map of the winning tap r*k+s (first maximum, PyTorch's tie rule), 255 when the
 * winner is <= 0 -- i.e. the map already carries the fused ReLU-backward mask of the tensor being pooled.            */
int din_maxpool_fwd(const din_pool_desc* d, const void* in, void* out, uint8_t* argmax, void* stream);
/* din = scatter of dout to the first maximal element of each window; relu_mask!=0 additionally multiplies by
 * (in > 0) -- the fused backward of the ReLU that produced `in`.  accumulate!=0: din += ...
 * With argmax (saved by the forward; requires relu_mask!=0) `in` is not read at all; without it the window arg-max
 * is recomputed from `in`.                                                                                          */
int din_maxpool_bwd(const din_pool_desc* d, const void* in, const uint8_t* argmax, const void* dout, void* din_,
                    int relu_mask, int accumulate, void* stream);
/* F.avg_pool2d(x, k, stride, pad) with count_include_pad (backbone: torchvision InceptionA/C branch_pool).  flags: DIN_CONV_BIAS adds
 * bias[c] (fp32) after the average, DIN_CONV_RELU clamps: the epilogue of a 1x1 conv commuted in front of the pool
 * (avgpool(conv1x1(x)) == conv1x1(avgpool(x)), both linear, zero padding maps to zero) -- the pool then runs on cout channels */
int din_avgpool_fwd(const din_pool_desc* d, const void* in, void* out, const float* bias, int flags, void* stream);
int din_avgpool_bwd(const din_pool_desc* d, const void* dout, void* din_, const void* mask, int accumulate,
                    void* stream);
int din_bilinear_fwd(const din_pool_desc* d, const void* in, void* out, void* stream); /* align_corners=True */
int din_bilinear_bwd(const din_pool_desc* d, const void* dout, void* din_, const void* mask, int accumulate,
                     void* stream);

/* ------------------------------------------------------------------------------------------------
 * Row R  RoIAlign(K,K) = TF crop_and_resize, transform_fpcoor=True (third-party longcw/RoIAlign.pytorch;
 * call site infer_model.py:178-180), optionally composed with the multi-scale fuse in front of it
 * (infer_model.py:165-172: F.interpolate(size=(OH,OW), mode='bilinear', align_corners=True) + torch.cat).
 * fm: the STORED map, NHWC [nb,hf,wf,c] (dtype fm_dtype, pixel stride ldf).  (gh,gw): the grid the boxes are expressed on.
 * gh==hf && gw==wf is the plain RoIAlign; a larger grid means "fm virtually resized to gh x gw with align_corners": each sample
 * is then a 3x3 weighted sum of stored pixels (both operations are separable and linear) and the resized map is never materialised.
 * boxes [m,4]=(x1,y1,x2,y2) grid px fp32; box_ind [m] int32; out fp32 [m][out_c][k][k] (the reference's flatten order
 * d,ky,kx: infer_model.py:181), this map's c channels at [out_coff, out_coff+c) -- one call per source of the torch.cat.
 * idx_out (nullable, plain sampling only) int32 [m][k][6] = (top,bottom,left,right,oob_y,oob_x) per sample row/col for
 * bit-exact index tests.
 * ---------------------------------------------------------------------------------------------- */
int din_roi_align_fwd(const void* fm, int fm_dtype, int nb, int hf, int wf, int c, int ldf, int gh, int gw,
                      const float* boxes, const int32_t* box_ind, int m, int k,
                      float* out, int out_c, int out_coff, int32_t* idx_out, void* stream);
/* dfm fp32 [nb,hf,wf,c] must be zeroed by the caller; 4-corner atomic scatter; no gradient to boxes */
int din_roi_align_bwd(const float* dout, int nb, int hf, int wf, int c,
                      const float* boxes, const int32_t* box_ind, int m, int k,
                      float* dfm, void* stream);
/* crop gradient [m][c][k*k] (the reference's flatten order = the column order of fc_emb_1) -> channel-contiguous [m][k*k][c]: the
 * gather backward then reads 16/32-byte runs instead of 4-byte loads 4*k*k bytes apart.  One call serves every source map. */
int din_roi_crop_grad_transpose(const float* dout, int m, int c, int k, float* out, void* stream);
/* Gather form of the gradient, written once into the stored map's gradient view gfm [nb,hf,wf,ldg] (storage `dtype`, channels
 * [0,c), every pixel -- zeros outside the boxes): no atomics, no fp32 staging tensor, no zero-fill, deterministic summation order.
 * dout: the crop gradient with dout_c channels per crop, this map's at [dout_coff, dout_coff+c); layout [m][dout_c][k*k], or
 * [m][k*k][dout_c] when `transposed` (din_roi_crop_grad_transpose).  (gh,gw) as in din_roi_align_fwd: with a larger grid this is the
 * fused backward of RoIAlign AND the bilinear resize.
 * fm_mask (nullable, same dtype, pixel stride ldf): the stored map; gradients are multiplied by (fm_mask > 0) -- the fused
 * backward of the ReLU that produced it.  box_ind may be in any order. */
int din_roi_align_bwd_nhwc(const float* dout, int dout_c, int dout_coff, int transposed, int nb, int hf, int wf, int c,
                           int gh, int gw, const float* boxes, const int32_t* box_ind, int m, int k,
                           const void* fm_mask, int dtype, int ldf, void* gfm, int ldg, void* stream);
/* fp32 gradient map -> backbone dtype, fused with the ReLU mask of the feature map that was cropped */
int din_grad_cast_mask(const float* g, const void* y, void* out, int dtype, int64_t pixels, int c,
                       int ldy, int yoff, int ldo, int ooff, int use_mask, void* stream);
/* builds box_ind[i*n+j] = i (infer_model.py:155-157) */
int din_boxes_frame_index(int32_t* out, int bt, int n, void* stream);

/* ------------------------------------------------------------------------------------------------
 * SURVEY 8(f)-4  Context-encoding transformer of Dynamic_TCE_volleyball (infer_model.py:404-410;
 * infer_module/TCE_STBiP_module.py:252-286 EmbfeatureContextEncodingTransformer, :300-313 the 4-head layer).  fp32.
 * q [bt][n][heads*c] (emb_roi of every head), kf [bt][p][heads*c] (downsample2 of every head over the p = OH*OW context pixels),
 * s / a [bt][heads][n][p].  n <= 16 (the reference asserts 12), c a power of two <= 256 (128 in the reference).
 * ---------------------------------------------------------------------------------------------- */
/* s = <q, kf> per (frame, head, box, pixel): torch.matmul(emb_roi_feature, image_feature) (:271).  Also the backward's dA = <dctx, kf>. */
int din_ctx_scores(const float* q, const float* kf, float* s, int bt, int n, int p, int heads, int c, void* stream);
/* F.softmax(a, dim=2) (:273) in place over rows of `len`; backward in place on da: ds = a * (da - sum(a * da)) */
int din_softmax_rows(float* s, int64_t rows, int len, void* stream);
int din_softmax_rows_bwd(const float* a, float* da, int64_t rows, int len, void* stream);
/* out[bt][n][heads*c] = sum_p a * kf: torch.matmul(A, image_feature) (:277).  Also the backward's dq = sum_p ds * kf. */
int din_ctx_apply(const float* a, const float* kf, float* out, int bt, int n, int p, int heads, int c, void* stream);
/* dkf = a^T dctx + ds^T q: the gradient of both uses of the keys, written once */
int din_ctx_keys_grad(const float* a, const float* ds, const float* dctx, const float* q, float* dkf, int bt, int n, int p, int heads,
                      int c, void* stream);
/* Context_PositionEmbeddingSine.forward's last line (positional_encoding.py:91): y[f][i] = float(x[f][i]) + pos[i]; x in the backbone's
 * storage type (`dtype`), `per_frame` = OH*OW*C elements.  Backward: gx = cast(gy) (* (x > 0) when use_mask: the backbone graph takes
 * gradients that are already multiplied by the ReLU mask of its output) */
int din_add_position(const void* x, int dtype, const float* pos, float* y, int64_t frames, int64_t per_frame, void* stream);
int din_add_position_bwd(const float* gy, const void* x, int dtype, void* gx, int64_t elems, int use_mask, void* stream);
/* y = dropout(relu?(x)) (nn.ReLU + nn.Dropout of the FFN, TCE_STBiP_module.py:243-247; nn.Dropout on the attended context :277) with the
 * counter-based keep mask of din_layernorm_* (same seed + element index -> same mask in the backward) */
int din_act_dropout_fwd(const float* x, float* y, int64_t n, int relu, float drop_p, uint64_t seed, const uint64_t* seed_offset, void* stream);
int din_act_dropout_bwd(const float* gy, const float* x, float* gx, int64_t n, int relu, float drop_p, uint64_t seed,
                        const uint64_t* seed_offset, void* stream);

/* ------------------------------------------------------------------------------------------------
 * LayerNorm (+residual, +ReLU, +dropout): nl_emb_1 (infer_model.py:185), point_ln (:192), dpi_nl (:214),
 * hier_LN (dynamic_infer_module.py:493).  x, res: [rows][len]; gamma/beta [len]; eps 1e-5.
 * y = dropout(relu?(LN(x + res?)*gamma + beta)).  stats: [rows][2] = (mean, rstd) saved for backward.
 * dropout: keep-mask from a counter-based hash of (seed, element index), scale 1/(1-p); p=0 disables.
 * seed_offset (nullable, device memory): the mask seed is (seed + *seed_offset) mod 2^63, read by the kernel -- a training step captured
 * in a HIP graph bakes `seed` into the launch, so the part that changes per step lives in device memory (din_counter_add advances it
 * inside the graph); forward and backward of one step see the same value.
 * ---------------------------------------------------------------------------------------------- */
int din_layernorm_fwd(const float* x, const float* res, const float* gamma, const float* beta, float eps,
                      float* y, float* stats, int64_t rows, int64_t len, int relu, float drop_p,
                      uint64_t seed, const uint64_t* seed_offset, void* stream);
/* dx [rows][len] (also the gradient of res); dgamma/dbeta [len] are ACCUMULATED atomically (caller zeroes) */
int din_layernorm_bwd(const float* dy, const float* x, const float* res, const float* gamma,
                      const float* y, const float* stats, float* dx, float* dgamma, float* dbeta,
                      int64_t rows, int64_t len, int relu, float drop_p, uint64_t seed, const uint64_t* seed_offset, void* stream);
/* *counter += delta (one thread; stream-ordered): the per-step part of the dropout seeds of a graph-captured training step */
int din_counter_add(uint64_t* counter, uint64_t delta, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Rows D2-D4  Dynamic Relation + Dynamic Walk (dynamic_infer_module.py:184-282, 344-404), one module,
 * one sampling ratio.  x fp32 [b,t,n,c].  pred fp32 [b,t,n,cp] is the output of the fused
 * p_conv/scale_conv contraction (din_conv_fwd with cout = 3*k2: channels [0,k2) y-offsets, [k2,2k2)
 * x-offsets, [2k2,3k2) relation logits; cp = its pixel stride).  scale_factor=0 -> mean over k2, no logits.
 *   z   [b,t,n,c]      = sum_k a_k * S_k                   (:278 / :280)
 *   a   [b,t,n,k2]     = softmax_k(logits)  (saved)        (:196)
 *   idx [b,t,n,k2,4]   int32 (ly,ry,lx,rx) clamped corners (:208-223)  -- bit-exact row
 *   mad (nullable) [b,t,n,k2,c] = S (ft_infer_MAD, :259); not materialised when NULL.
 * ---------------------------------------------------------------------------------------------- */
/*   n_per_clip (nullable) int32 [b]: clip i is a T x n_per_clip[i] grid held in the first columns of its T x n slab (Dynamic_collective,
 *   infer_model.py:1286-1293 runs the module per clip on boxes_features_all[b, :, :N]); columns beyond it are zero padding: they are
 *   read as zeros, get z = 0 and no gradient, and the x clamp range is the clip's own padded width.  x must already be zero there.
 *   The forward needs ONE padded (t+2pt) x (n+2pl) x 64-channel fp32 tile in LDS (<= 160 KiB, else DIN_E_ARG); the backward keeps a
 *   second one for the feature gradient when it fits and scatters into dx with global atomics when it does not.                 */
/*   plain != 0: no walk -- S_k is the feature at lattice point k itself (plain_infer_ratio :154-181 and the relation half of
 *   parallel_infer :285-298); the offset channels of pred are ignored and receive no gradient.
 *   clamp (nullable HOST array of 4 ints {iy_max, ix_max, py_max, px_max}): clamp maxima of the corner indices and of the sampling
 *   position, for parallel_infer's walk half, which clamps with person_mat_shape (T + 2 ratio - 1, N + 2 ratio - 1; T + 2 ratio,
 *   N + 2 ratio: :307-317) instead of the padded grid; index maxima are additionally held inside the padded grid.              */
int din_walk_fwd(const float* x, const float* pred, int cp, int b, int t, int n, int c,
                 int kh, int kw, int ratio, int scale_factor, int plain, const int32_t* clamp, const int32_t* n_per_clip,
                 float* z, float* a, int32_t* idx, float* mad, void* stream);
/* gz [b,t,n,c] -> dx_walk [b,t,n,c] (overwritten), dpred [b,t,n,cp] (first 3*k2 channels written):
 * d offset through the |.| coefficients with detached floor (Q4), inclusive clamp pass-through (Q9),
 * sign(0)=0 (Q8); d logits through the softmax.  scratch: fp32 [ceil(c/64)][b,t,n,3*k2] (per-channel-chunk partial sums, written by the call).     */
int din_walk_bwd(const float* x, const float* pred, int cp, const float* a, const float* gz,
                 int b, int t, int n, int c, int kh, int kw, int ratio, int scale_factor, int plain, const int32_t* clamp,
                 const int32_t* n_per_clip, float* dx, float* dpred, float* scratch, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Row H  head (infer_model.py:224-232): max over actors -> fc_activities -> mean over frames.
 * s fp32 [b,t,n,c]; w [a,c]; bias [a]; argmax int32 [b,t,c] saved for backward.
 * scores: caller allocates b*a + b*t*a floats: [0, b*a) = clip scores [b,a], the tail = per-frame scores [b,t,a].
 * n_per_clip (nullable, int32 [b]) = number of valid actors of each clip (Collective: variable N).
 * ---------------------------------------------------------------------------------------------- */
int din_head_fwd(const float* s, const float* w, const float* bias, const int32_t* n_per_clip,
                 int b, int t, int n, int c, int a, float* scores, int32_t* argmax, void* stream);
/* ds overwritten; dw, dbias ACCUMULATED atomically (caller zeroes) */
int din_head_bwd(const float* dscores, const float* s, const float* w, const int32_t* argmax,
                 int b, int t, int n, int c, int a, float* ds, float* dw, float* dbias, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Small helpers used by the host mirror
 * ---------------------------------------------------------------------------------------------- */
/* out = alpha*x + beta*y (fp32, elementwise) -- ratio mean / beta-weighted sum (:144-147), residual sums */
int din_axpby(const float* x, const float* y, float* out, float alpha, float beta, int64_t n, void* stream);
/* out[b,t,i,:] = i < n_per_clip[b] ? x[b,t,i,:] : 0 -- zeroes the padding actors of Collective clips so that the batched T x MAX_N grid
 * behaves like the reference's per-clip slices boxes_features_all[b, :, :N] (infer_model.py:1286-1288); its own backward */
int din_mask_actors(const float* x, const int32_t* n_per_clip, int b, int t, int n, int c, float* out, void* stream);
/* out (+)= x * scalar[idx] with a DEVICE scalar (learnable beta, :42-44,145); out[idx] += <x,y> for its gradient */
int din_scale_by_param(const float* x, const float* scalar, int idx, float* out, int accumulate, int64_t n, void* stream);
int din_dot_accum(const float* x, const float* y, float* out, int idx, int64_t n, void* stream);
/* dst[i] = (dtype)src[i] conversions between fp32 and bf16 */
int din_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, void* stream);
/* NHWC(dtype, ld/coff) -> NCHW fp32 and back: API-parity views for MyVGG16/MyInception_v3 outputs */
int din_nhwc_to_nchw_f32(const void* in, int dtype, int nb, int h, int w, int c, int ld, int coff,
                         float* out, void* stream);
int din_nchw_f32_to_nhwc(const float* in, int nb, int h, int w, int c, void* out, int dtype, int ld,
                         int coff, void* stream);
/* fused Adam step over a flat fp32 parameter/gradient buffer (train_net_dynamic.py:104,224) */
int din_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream);
/* The same update for a whole parameter list in ONE launch.  Device tables: ptrs [nt][4] = {p, g, m, v} addresses, sizes [nt]
 * element counts, and one (tensor, chunk index) pair per workgroup: workgroup b updates elements
 * [chunk_index[b]*chunk_elems, +chunk_elems) of tensor chunk_tensor[b]. */
int din_adam_step_multi(const uint64_t* ptrs, const int64_t* sizes, const int32_t* chunk_tensor, const int32_t* chunk_index,
                        int nchunks, int chunk_elems, float lr, float beta1, float beta2, float eps, float weight_decay,
                        int step, float grad_scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DIN_HIP_H */
