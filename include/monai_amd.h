/* monai_amd -- C ABI of the MI355X (gfx950) kernels behind MONAI's 3-D sliding-window segmentation
 * hot path.  This library takes the place of the reference's native layer (`monai._C`, built from
 * monai/csrc by setup.py:81-125, and the JIT loader monai/_extensions/loader.py:49-93) for this path,
 * and of the ATen operators the reference's Python reaches on it.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory owned by the caller (in practice
 *     torch-ROCm tensors); nothing is allocated, freed or synchronised inside the library;
 *   - every call enqueues its kernels on `stream` (a hipStream_t passed as void*) and returns at once;
 *   - return value 0 = ok, negative = error; mh_last_error() gives the thread-local message.  This is
 *     the AT_ERROR -> RuntimeError behaviour of the reference (monai/csrc/resample/pushpull.h:64-78);
 *   - all data is fp32 ("compute_dtype = inputs.dtype", monai/inferers/utils.py:148), laid out NCDHW
 *     with W contiguous.  2-D problems are passed with D = 1.
 */
#ifndef MONAI_AMD_H
#define MONAI_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MH_OK 0
#define MH_ERR_ARG (-1)
#define MH_ERR_LAUNCH (-2)
#define MH_ERR_UNSUPPORTED (-3)

/* Activation tensor view.  `nrm` is NULL or points at one {alpha, beta, slope, bound} float4 per (n, c):
 * consumers see  y = fma(x, alpha, beta); y = y > 0 ? y : y*slope  (InstanceNorm3d(affine) followed by
 * LeakyReLU -- the ADN("NDA") block of monai/networks/blocks/acti_norm.py:69-101 -- applied on load
 * instead of in a pass of its own).  `bound` >= max |y| over the (n, c) plane when the producer of the tensor knows
 * one, 0 when none is given, inf / NaN when the plane (or its statistics) holds a non-finite value:
 *   - mh_instnorm_finalize_f32 / mh_groupnorm_finalize_f32 write (|gamma| sqrt(count) + |beta|) max(1, |slope|);
 *   - on an OUTPUT view of mh_deconv_k2s2_f32 / mh_deconv_ks_f32 / mh_add_act_f32 / mh_pixelshuffle_f32, `nrm` (if not NULL) names identity
 *     records prepared by mh_nrm_identity_f32; the kernel folds max |value written| into their `bound`;
 *     mh_maxpool2_f32 / mh_pad_replicate_f32 write {1, 0, 1, bound of their input} into it.
 * The split-precision convolution (mh_conv3d_k3_h2_config) needs the bounds of its input; nothing else reads them.
 * n_stride / nrm_n_stride are in floats. */
typedef struct mh_tensor5 {
    float* data;
    int64_t n_stride;
    const float* nrm;
    int64_t nrm_n_stride;
    int32_t N, C, D, H, W;
} mh_tensor5;

int mh_version(void);
const char* mh_last_error(void);

/* ---- sliding-window inferer (monai/inferers/utils.py:42-321) -------------------------------------- */

/* The dense window grid of dense_patch_slices (monai/data/utils.py:166-206) is described by its per-axis
 * start lists sz/sy/sx (ascending HOST int32 arrays); window w = (iz*ny + iy)*nx + ix, i.e. row-major with
 * the last axis fastest.  Lists of dense_patch_slices' own form -- i*interval, the last one clipped to
 * image - roi -- are recognised and evaluated in closed form inside the kernels (any number of windows per
 * axis, e.g. SliceInferer's one window per slice); any other ascending list is copied into the kernel
 * arguments (at most 160 windows per axis). */

/* Window gather, utils.py:217-224 (`torch.cat([inputs[win_slice] ...])`).  vol is one image [C,D,H,W];
 * windows w0 .. w0+nwin-1 of the grid are written to out [nwin, C, rd, rh, rw] (dense). */
int mh_window_extract_f32(const float* vol, int C, int D, int H, int W, const int32_t* sz, int nz,
                          const int32_t* sy, int ny, const int32_t* sx, int nx, int w0, int nwin, int rd, int rh,
                          int rw, float* out, void* stream);

/* Importance-weighted blend + normalise, utils.py:264-298 (`seg *= w; out[win] += seg; cnt[win] += w;
 * out /= cnt`) as ONE gather pass: for every output voxel the covering windows are visited in
 * ascending window index, acc += fp32(logit*w), cnt += w, out = acc/cnt -- bit-identical to the
 * reference's scatter order.
 *   logits  [nz*ny*nx][K][rd][rh][rw]   all windows of the grid, in window order; consecutive windows are `window_stride`
 *                                       floats apart (0 = dense, K*rd*rh*rw).  A stride that is not a multiple of a large power
 *                                       of two spreads the concurrently read (window, class) streams over the HBM channels.
 *   imp     [rd][rh][rw]                importance map (monai/data/utils.py:1084-1134)
 *   out     [K][D][H][W]
 * Every voxel must be covered by at least one window (true for dense_patch_slices).
 * premultiplied != 0: `logits` already hold logit*w (the `process_fn` path, utils.py:232-234, where the weight may change
 * per window batch): acc += logit, cnt += imp -- the reference's `seg *= w_t` followed by `+=`, with its count map. */
int mh_sw_blend_f32(const float* logits, int64_t window_stride, const float* imp, float* out, int K, int D, int H, int W,
                    int rd, int rh, int rw, const int32_t* sz, int nz, const int32_t* sy, int ny, const int32_t* sx,
                    int nx, int premultiplied, void* stream);

/* The blend in the summation order of the reference's BUFFERED schedule (monai/inferers/utils.py:239-253, 276-284, 324-348: `buffer_steps` > 0, `buffer_dim`):
 * windows stably sorted by their start along the buffered axis, groups of `buffer_steps` distinct starts accumulated from zero and then added to the output, the
 * count map in the sorted order -- bit-identical to the reference's buffered run (its non-buffered run differs from it by roundings).  Window-major logits as
 * mh_sw_blend_f32; `buffer_axis` 0 / 1 / 2 = z / y / x of the 3-D view; premultiplied != 0: `logits` hold logit * weight already (process_fn), `imp` is the count's map.  At most 160 windows per axis. */
int mh_sw_blend_buffered_f32(const float* logits, int64_t window_stride, const float* imp, float* out, int K, int D, int H, int W, int rd, int rh, int rw,
                             const int32_t* sz, int nz, const int32_t* sy, int ny, const int32_t* sx, int nx, int buffer_axis, int buffer_steps, int premultiplied, void* stream);

/* The blend over the MOSAIC logits layout (single-GPU fused path).  Windows i and i + m of an axis do not overlap when m * step >= roi, so the
 * windows of one residue class per axis (i mod m, m = 2^log2m in {1, 2, 4}; the last -- clipped -- window of an axis is a class of its own, index m)
 * tile space: the logits of class (cz, cy, cx) are ONE dense array [K][cnt_z*rd][cnt_y*rh][cnt_x*rw] at float offset class_base[(cz*5 + cy)*5 + cx]
 * (HOST int64[125]) from `logits`, window (jz, jy, jx) of the class (j = i >> log2m) at its mosaic position; mh_sw_mosaic_class_counts gives cnt per
 * class of an axis with n windows.  At overlap 0.5 the blend then reads 1 KB runs per wave from <= 8 (27) arrays instead of 384-byte pieces of
 * 8 (27) x K window blocks.  Same arithmetic in the same (ascending window) order as mh_sw_blend_f32: identical bits.  Regular grids, K <= 8, extents
 * divisible by 4; MH_ERR_UNSUPPORTED otherwise (use the window-major layout).  mh_conv1x1_windows_f32 writes the layout.
 * imp_factored != 0: `imp` is not the [rd][rh][rw] map but its factors [gz (rd) | gy (rh) | gx (rw) | floor] with map[z][y][x] = max(fl(fl(gz[z] gy[y]) gx[x]),
 * floor) -- how compute_importance_map (monai/data/utils.py:1084-1134) forms the gaussian and the constant map; the kernel re-forms the same fp32 values in
 * registers ((rd + rh) % 4 == 0 required). */
int mh_sw_mosaic_class_counts(int n, int log2m, int32_t* counts5);
int mh_sw_blend_mosaic_f32(const float* logits, const int64_t* class_base, int log2m_z, int log2m_y, int log2m_x, const float* imp, int imp_factored, float* out, int K,
                           int D, int H, int W, int rd, int rh, int rw, const int32_t* sz, int nz, const int32_t* sy, int ny, const int32_t* sx, int nx,
                           void* stream);

/* The same blend with the post-processing step that follows it in segmentation bundles fused into its epilogue:
 * AsDiscrete(argmax=True) (monai/transforms/post/array.py:132-237, `torch.argmax(img, dim=0, keepdim=True)`).  Every voxel's K
 * blended values are formed in registers exactly as mh_sw_blend_f32 forms them and only the label -- index of the first maximal
 * value, NaN counts as maximal -- is written: labels [D][H][W] as float32 (labels_u8 = 0, AsDiscrete's default output dtype) or
 * uint8 (labels_u8 = 1, K <= 256).  Output traffic K*4 B per voxel -> 4 B or 1 B.  Regular (dense_patch_slices) grids only;
 * MH_ERR_UNSUPPORTED otherwise (blend, then mh_channel_reduce_f32). */
int mh_sw_blend_argmax_f32(const float* logits, int64_t window_stride, const float* imp, void* labels, int labels_u8, int K, int D,
                           int H, int W, int rd, int rh, int rw, const int32_t* sz, int nz, const int32_t* sy, int ny,
                           const int32_t* sx, int nx, int premultiplied, void* stream);

/* AvgMerger of the PatchInferer family (monai/inferers/merger.py:103-205): one patch, `values[slice] += patch; counts[slice] += 1`
 * (values / patch [NC][...] fp32 dense, counts uint8 like the reference's default count_dtype; the patch must lie inside the
 * merged volume), and the final in-place `values /= counts`.  2-D / 1-D problems pad with leading size-1 axes. */
int mh_patch_accumulate_f32(float* values, uint8_t* counts, const float* patch, int NC, int D, int H, int W, int pd, int ph,
                            int pw, int z0, int y0, int x0, void* stream);
/* The same for a whole batch of `npatch` equally sized patches (patches [npatch][NC][pd][ph][pw] dense, loc [npatch][3] HOST int32 (z, y, x)) in one
 * launch: every merged element adds the patches that cover it in batch order -- the bits of npatch single-patch calls, also where patches of the batch
 * overlap each other. */
int mh_patch_accumulate_batch_f32(float* values, uint8_t* counts, const float* patches, int npatch, const int32_t* loc, int NC, int D, int H, int W,
                                  int pd, int ph, int pw, void* stream);
int mh_avg_finalize_f32(float* values, const uint8_t* counts, int64_t n, void* stream);

/* ---- network blocks (BasicUNet: monai/networks/nets/basic_unet.py:27-279) -------------------------- */

/* Conv3d k=3, stride 1, padding 1 (+bias) -- the conv of `Convolution`, blocks/convolutions.py:98-171.
 * Several kernel configurations exist; mh_conv3d_k3_select picks one:
 *   0                      direct VALU kernel, any channel counts
 *   1 .. n-1               fp32-MFMA implicit-GEMM tiles (v_mfma_f32_32x32x2_f32), exact fp32
 *   n = num_configs()      Winograd F(2x2, 3x3) in (y, x) + three direct z taps, z-streaming, fp32 (Cin % 8 == 0, Cout % 16 == 0,
 *                          even H, W % 8 == 0); selected for full 16 x 16 regions of planes >= 48^2 with D >= 48
 *   h2_config()            fp16 matrix cores, two-piece split precision, fp32-equivalent (below)
 *   c1_config()            one input channel, packed fp32 vector arithmetic (below)
 * `algo` restricts the choice to an arithmetic family -- it is an ARGUMENT: the library reads no environment variable and keeps
 * no state, what a call computes depends on its arguments alone.  `input_bounded` != 0 promises that every record of the input
 * view the convolution will be given carries a magnitude bound (mh_tensor5 above); without it the split-precision
 * configuration is never returned.  Weights are repacked (Winograd: transformed) once per configuration. */
#define MH_ALGO_AUTO 0     /* fastest configuration that is fp32-equivalent for the given input: h2 for bounded inputs, fp32 otherwise */
#define MH_ALGO_DIRECT 1   /* exact-fp32 matrix-core tiles only (no Winograd, no split precision, no one-channel kernel) */
#define MH_ALGO_WINO2D 2   /* the in-plane Winograd configuration wherever its shape rules allow, fp32 tiles elsewhere */
#define MH_ALGO_H2 3       /* split precision wherever it fits and the input is bounded, fp32 tiles elsewhere (no Winograd) */
#define MH_ALGO_FP32 4     /* AUTO among the exact-fp32 kernels (tiles, in-plane Winograd, one-channel kernel) */
int mh_conv3d_k3_select(int algo, int input_bounded, int Cin, int Cout, int D, int H, int W);
int mh_conv3d_k3_num_configs(void);                    /* highest configuration id of the exact-fp32 matrix-core family */
/* Configuration outside 0 .. num_configs(): z-streaming direct convolution on the fp16 matrix cores in two-piece split
 * precision (kernels/conv3d_h2.h) -- every fp32 operand as hi + lo fp16 pieces, products hi*hi + lo*hi + hi*lo accumulated in
 * fp32: fp32-equivalent results (oracle BasicUNet: max |logit difference| 4e-6, the level of two fp32 summation orders) at 3/16
 * of the fp32 matrix-core cycles.  Needs Cin % 16 == 0, Cin <= 256, Cout % 16 == 0 with Cout >= 32 (round 4: a half-filled last group of 32 couts -- zero weight columns, dropped stores and statistics), W % 4 == 0 and D * H * W < 2^24 voxels (the result stores address the
 * 32 output planes of a workgroup through one raw buffer with 31-bit byte offsets; mh_conv3d_k3_select does not return the configuration beyond,
 * mh_conv3d_k3_f32 answers MH_ERR_UNSUPPORTED).  The activated input of sample n is scaled by the
 * power of two that puts the largest `bound` of its records just below 2^15 (undone exactly in the epilogue), so any finite
 * magnitude is in range; a record without a bound, or with a non-finite one, makes that sample's output NaN (what the reference
 * computes behind a normalisation whose statistics are non-finite, and a loud failure for a caller that broke the contract).
 * With in->nrm == NULL there is nothing to take a scale from: the input is used as it is (|x| must stay below 65504). */
int mh_conv3d_k3_h2_config(void);
/* mh_conv3d_k3_h2_config's kernel with output channel groups of 16 (round 4): a 32-column matrix instruction carries two z-taps of 16 couts instead of one tap of 32
 * ([kz 0 | kz 1] and [kz 2 | 0]: 6 instead of 9 instructions per in-plane tap), a completed plane is the sum of three partial planes.  Same reference op, record / bound
 * contract and tolerance class; Cin % 16 == 0, Cin <= 256, Cout % 16 == 0, W % 4 == 0.  mh_conv3d_k3_select returns it for Cout == 16 (UNETR's / SwinUNETR's full-resolution
 * levels), where a 32-couts group would be half zero weights. */
int mh_conv3d_k3_h2c_config(void);
/* Configuration outside 0 .. num_configs(): ONE input channel (the first layer of the networks), packed fp32 vector arithmetic
 * (kernels/conv3d_c1.h) -- exact fp32 like the matrix-core tiles, bound by writing the result instead of by multiplying a zero-padded
 * channel pair.  Needs Cin == 1, Cout % 8 == 0, W % 4 == 0; chosen by mh_conv3d_k3_select for such layers. */
int mh_conv3d_k3_c1_config(void);
/* mh_conv3d_k3_h2_config's arithmetic for SMALL volumes (round 5, kernels/conv3d_vol_h2.h): one sample's whole D x H x W volume is the workgroup's M tile (no z-march,
 * no halo recomputation, one statistics record per (n, cout)) -- the 6^3 level of a 96^3 window, where 16 x 16 regions would be 14 % full.  Same reference op, record /
 * bound contract and tolerance class; Cin % 16 == 0 (<= 768), Cout % 32 == 0, 64 <= D*H*W <= 256, (D+2)(H+2)(W+2) <= 512.  mh_conv3d_k3_select returns it for bounded
 * inputs of such volumes that mh_conv3d_k3_h2_config does not take (H or W below 8). */
int mh_conv3d_k3_h2v_config(void);
/* mh_conv3d_k3_h2_config's arithmetic behind an in-plane Winograd F(2x2, 3x3) transform (round 6, kernels/conv3d_wino_h2.h): 2.25x fewer matrix instructions per output
 * voxel, the transformed weights resident in registers (one wave per SIMD, each owning one row of the 4 x 4 transform positions), z-streaming.  Same record / bound contract
 * and tolerance class; Cin == 32, Cout % 32 == 0, H % 4 == 0, W % 16 == 0, D >= 2 (mh_conv3d_k3_h2w_fits).  Takes mh_conv3d_k3_accumulate_f32 and mh_conv3d_k3_pool_f32
 * (even D) as well.  Under MH_ALGO_AUTO mh_conv3d_k3_select returns it in place of mh_conv3d_k3_h2_config for such layers with D >= 24 (measured 1.17x at 96^3, 1.3x at 48^3:
 * profiles/r06_h2w_ab.txt); MH_ALGO_H2 by name keeps the direct kernel. */
int mh_conv3d_k3_h2w_config(void);
int mh_conv3d_k3_h2w_fits(int D, int H, int W);
int mh_conv3d_k3_accepts(int cfg, int Cin, int Cout);  /* 1 if `cfg` can run these channel counts */
int64_t mh_conv3d_k3_packed_floats(int cfg, int Cin, int Cout);
/* w: torch layout [Cout][Cin][3][3][3] */
int mh_conv3d_k3_pack_f32(int cfg, const float* w, int Cin, int Cout, float* packed, void* stream);
/* number of statistics records per (n, c) this configuration emits (0: none, use mh_instnorm_stats) */
int mh_conv3d_k3_stat_tiles(int cfg, int D, int H, int W);
/* out = conv(act(in)) + bias.  `stats`: NULL or [N][Cout][tiles][3] {count, mean, M2} partial records
 * of the values written (fused InstanceNorm statistics). */
int mh_conv3d_k3_f32(int cfg, const mh_tensor5* in, const float* packed_w, const float* bias,
                     const mh_tensor5* out, float* stats, void* stream);
/* mh_conv3d_k3_f32 that also leaves MaxPool3d(kernel_size=2) of its output (`Down`, monai/networks/nets/basic_unet.py:61-89: the pooling behind an encoder block) without
 * a pass over the full-resolution tensor: pool_max / pool_min [N][>= Cout][D/2][H/2][W/2] (batch stride pool_n_stride floats) receive the 2 x 2 x 2 maxima and minima of the
 * RAW output; the consumer reads pool_max under `out`'s records after mh_pool_select_f32(pool_max, pool_min, out's records) has copied the minima over the channels whose
 * alpha is negative (normalise + LeakyReLU is monotone in the raw value: max of the activated values = activation of the raw max for alpha >= 0, of the raw min otherwise).
 * The split-precision configuration on even extents whose regions are 16 x 16 (mh_conv3d_k3_pool_accepts); input records and statistics required. */
int mh_conv3d_k3_pool_accepts(int cfg, int Cin, int Cout, int D, int H, int W);
int mh_conv3d_k3_pool_f32(int cfg, const mh_tensor5* in, const float* packed_w, const float* bias, const mh_tensor5* out, float* stats, float* pool_max, float* pool_min,
                          int64_t pool_n_stride, void* stream);
int mh_pool_select_f32(float* pool_max, const float* pool_min, const float* nrm, int64_t nrm_n_stride, int N, int C, int64_t n_stride, int64_t vol, void* stream);
/* mh_conv3d_k3_f32 whose result is ADDED to what `out` holds (out += conv + bias; `stats` = the statistics of the sum): the split-precision configuration
 * (mh_conv3d_k3_h2_config, its 16-cout form mh_conv3d_k3_h2c_config, mh_conv3d_k3_h2w_config) with input records and statistics only -- MH_ERR_UNSUPPORTED otherwise.
 * One half of the UpCat path (mh_upconv_k4s2_f32), a convolution evaluated in halves of its input channels, a residual join x += conv(...) (SegResNet's ResBlock,
 * monai/networks/blocks/segresnet_block.py:62-97). */
int mh_conv3d_k3_accumulate_f32(int cfg, const mh_tensor5* in, const float* packed_w, const float* bias, const mh_tensor5* out, float* stats, void* stream);

/* InstanceNorm3d statistics (nn.InstanceNorm3d(affine=True, eps) via layers/factories.py:228-241):
 * partial {count, mean, M2} records per 4096-element chunk of each (n, c) plane, then a finalize that
 * merges `tiles` records per (n, c) in fp64 and writes the consumer-side float4
 * {alpha = gamma/sqrt(var+eps), beta = bias - mean*alpha, slope, bound} (biased variance; bound: see mh_tensor5). */
int mh_instnorm_stat_tiles(int D, int H, int W);
int mh_instnorm_stats_f32(const mh_tensor5* x, float* stats, void* stream);
int mh_instnorm_finalize_f32(const float* stats, int tiles, int N, int C, const float* gamma, const float* beta,
                             float eps, float slope, float* nrm, int64_t nrm_n_stride, void* stream);
/* GroupNorm (nn.GroupNorm(groups, C, eps, affine), monai/networks/layers/utils.py get_norm_layer("group")): the same
 * per-(n, c) records, merged over the C / groups consecutive channels of each group; every channel of a group gets the
 * group's mean / variance with its own gamma / beta.  groups == C is InstanceNorm. */
int mh_groupnorm_finalize_f32(const float* stats, int tiles, int N, int C, int groups, const float* gamma, const float* beta, float eps,
                              float slope, float* nrm, int64_t nrm_n_stride, void* stream);

/* Identity records {1, 0, 1, bound = FLT_MIN ("known: nothing written yet")} for N x C channels, nrm_n_stride floats between samples: what a raw
 * producer's output view must point at so that the kernel can leave max |value| in `bound` (mh_tensor5 above). */
int mh_nrm_identity_f32(float* nrm, int N, int C, int64_t nrm_n_stride, void* stream);

/* MaxPool3d(kernel_size=2) of act(in) -- `Down`, basic_unet.py:61-89.  out dims = floor(in/2); out->D == in->D pools plane by plane
 * (MaxPool2d of a 2-D network that runs as one plane of this engine). */
int mh_maxpool2_f32(const mh_tensor5* in, const mh_tensor5* out, void* stream);

/* ConvTranspose3d(k=2, s=2) (+bias) of act(in) -- UpSample(mode="deconv"), blocks/upsample.py:102-116.
 * w: torch layout [Cin][Cout][2][2][2].  out dims = 2*in. */
int mh_deconv_k2s2_f32(const mh_tensor5* in, const float* w, const float* bias, const mh_tensor5* out,
                       void* stream);

/* The same transposed convolution (k = 2, s = 2) on the fp16 matrix cores in the split precision of mh_conv3d_k3_h2_config (kernels/deconv_h2.h): one GEMM with
 * (cout, parity) rows whose result is stored pixel-shuffled; the input is read once per 32 (16) output channels instead of once per 4.  `in` must carry records
 * with magnitude bounds; Cin % 16 == 0 (<= 1024), Cout % 16 == 0 (mh_deconv_k2s2_h2_accepts); w [Cin][Cout][2][2][2] packed once by mh_deconv_k2s2_h2_pack_f32 into
 * mh_deconv_k2s2_h2_packed_floats(Cin, Cout) floats.  Output records (when the view has them) receive the magnitude bound as in mh_deconv_k2s2_f32. */
int mh_deconv_k2s2_h2_accepts(int Cin, int Cout, int D, int H, int W);
int64_t mh_deconv_k2s2_h2_packed_floats(int Cin, int Cout);
int mh_deconv_k2s2_h2_pack_f32(const float* w, int Cin, int Cout, float* packed, void* stream);
int mh_deconv_k2s2_h2_f32(const mh_tensor5* in, const float* packed, const float* bias, const mh_tensor5* out, void* stream);

/* UpCat (monai/networks/nets/basic_unet.py:130-178) without its up-sampled intermediate: the `x_0 = ConvTranspose3d(k2, s2)(x)` half of
 * `Conv3d(k3, p1)(cat([x_e, x_0]))` is the composite transposed convolution k4 s2 p1 of x itself (kernels/upconv_h2.h) -- 8 instead of 27 taps per output voxel, no
 * full-resolution x_0 written or read.  w4 [Cin][Cout][4][4][4] = the composite weights (host: sum over the up channels of deconv x conv weight products),
 * bias_table [27][Cout] = what the deconvolution's bias contributes per (first / interior / last) position class of (z, y, x).
 * mh_upconv_k4s2_f32 with accumulate == 0 WRITES  convT4(act(low)) + bias_table  to `out` (mh_conv3d_k3_accumulate_f32 on the skip channels then adds the
 * convolution's x_e half and leaves the statistics of the sum); with accumulate != 0 it ADDS them to `out` in place -- `out` holding the x_e half, raw, written by
 * mh_conv3d_k3_f32 -- and, with stats != NULL, leaves one {count, mean, M2} record per (n, cout, tile) of the SUM (mh_upconv_k4s2_stat_tiles of them per (n, c),
 * merged by mh_instnorm_finalize_f32).  Split-precision arithmetic and record / bound contract of mh_conv3d_k3_h2_config.  Cin == 32, Cout % 32 == 0,
 * low W % 4 == 0, out = 2 x low; mh_upconv_k4s2_accepts says so. */
int mh_upconv_k4s2_accepts(int Cin, int Cout, int Dl, int Hl, int Wl);
int64_t mh_upconv_k4s2_packed_floats(int Cin, int Cout);
int mh_upconv_k4s2_pack_f32(const float* w4, int Cin, int Cout, float* packed, void* stream);
int mh_upconv_k4s2_stat_tiles(int Dl, int Hl, int Wl);
int mh_upconv_k4s2_f32(const mh_tensor5* low, const float* packed, const float* bias_table, const mh_tensor5* out, int accumulate, float* stats, void* stream);

/* Conv3d k=3, stride 2, padding 1 (+bias) of act(in) on the fp16 matrix cores in the split precision of mh_conv3d_k3_h2_config (kernels/conv3d_s2_h2.h): the
 * down-sampling convolutions of DynUNet (monai/networks/blocks/dynunet_block.py:135-186), SegResNet (monai/networks/nets/segresnet.py:111-133) and UNet
 * (monai/networks/nets/unet.py:197-237) as 8 dense stride-1 sub-convolutions over the input's parity phases (27 taps in all, no zero taps).  w: [Cout][Cin][3][3][3];
 * `in` must carry records with magnitude bounds; extents even; Cin % 16 == 0, Cout % 32 == 0 (mh_conv3d_k3s2_accepts).  `workspace`:
 * mh_conv3d_k3s2_workspace_floats(N, Cin, D, H, W) floats, 16-byte aligned, caller-owned scratch (the activated, scaled, phase-split fp16 pieces of the input).
 * With stats != NULL one {count, mean, M2} record per (n, cout, tile): mh_conv3d_k3s2_stat_tiles(D, H, W) of them per (n, c), merged by mh_instnorm_finalize_f32.
 * fused != 0: no split pass and no workspace (may be NULL) -- the GEMM's staging converts the fp32 input itself, once per group of 64 output channels (Cin <= 512):
 * the better form for layers with one or two such groups, where the split pass costs as much as the GEMM. */
int mh_conv3d_k3s2_accepts(int Cin, int Cout, int D, int H, int W);
int64_t mh_conv3d_k3s2_packed_floats(int Cin, int Cout);
int64_t mh_conv3d_k3s2_workspace_floats(int N, int Cin, int D, int H, int W);
int mh_conv3d_k3s2_stat_tiles(int D, int H, int W);
int mh_conv3d_k3s2_pack_f32(const float* w, int Cin, int Cout, float* packed, void* stream);
int mh_conv3d_k3s2_f32(const mh_tensor5* in, const float* packed, const float* bias, const mh_tensor5* out, float* workspace, float* stats, int fused, void* stream);

/* Conv3d k=1 (+bias) of act(in) -- `final_conv`, basic_unet.py:252.  w: [Cout][Cin]. */
int mh_conv1x1_f32(const mh_tensor5* in, const float* w, const float* bias, const mh_tensor5* out,
                   void* stream);
/* The same convolution leaving the InstanceNorm statistics of its output as it writes it (the 1x1x1 shortcut of UnetResBlock, `conv3` -> `norm3`,
 * blocks/dynunet_block.py:72-111): stats [N][Cout][tiles][3] = {count, mean, M2} per workgroup tile, tiles = mh_conv1x1_stat_tiles(D, H, W); the
 * records go to mh_instnorm_finalize_f32 like those of mh_instnorm_stats_f32 (no pass of its own over the tensor). */
int mh_conv1x1_stat_tiles(int D, int H, int W);
int mh_conv1x1_stats_f32(const mh_tensor5* in, const float* w, const float* bias, const mh_tensor5* out, float* stats, void* stream);

/* The same 1x1x1 convolution with ALL output channels from one read of the input, on the fp16 matrix cores in two-piece split precision (fp32-equivalent;
 * kernels/conv1x1_h2.h): UnetResBlock.conv3 (monai/networks/blocks/dynunet_block.py:72-111) where Cout > 16 made mh_conv1x1_f32 re-read the input per 16 couts.
 * The input's records must carry magnitude bounds (as for mh_conv3d_k3_f32's split-precision configuration).  packed = mh_conv1x1_h2_pack_f32(w [Cout][Cin]);
 * stats (or NULL): the records of mh_conv1x1_stats_f32, same tile count (mh_conv1x1_stat_tiles).  _accepts: Cin <= 1536, D*H*W % 4 == 0; 16-byte aligned tensors. */
int mh_conv1x1_h2_accepts(int Cin, int Cout, int D, int H, int W);
int64_t mh_conv1x1_h2_packed_floats(int Cin, int Cout);
int mh_conv1x1_h2_pack_f32(const float* w, int Cout, int Cin, float* packed, void* stream);
int mh_conv1x1_h2_f32(const mh_tensor5* in, const float* packed, const float* bias, const mh_tensor5* out, float* stats, void* stream);

/* The same 1x1 convolution with one strided destination per batch element (the windows of a sliding-window launch written straight into the mosaic
 * logits layout): place [N][4] HOST int64 = {float offset from `base`, channel stride, z stride, y stride}; x stays contiguous.  Cout <= 8. */
int mh_conv1x1_windows_f32(const mh_tensor5* in, const float* w, const float* bias, float* base, int Cout, const int64_t* place, void* stream);

/* ---- UNETR pieces (monai/networks/nets/unetr.py, blocks/selfattention.py, blocks/dynunet_block.py) ------------ */

/* out = lrelu_slope(act(a) + act(b)): the residual join of UnetResBlock (dynunet_block.py:96-111); a, b carry their
 * deferred InstanceNorm records (b->nrm NULL = identity shortcut; b NULL = no second operand: materialises act(a)). */
int mh_add_act_f32(const mh_tensor5* a, const mh_tensor5* b, float slope, const mh_tensor5* out, void* stream);

/* The residual join fused into the 1x1x1 output convolution behind it: out[co] = bias[co] + sum_ci w[co][ci] lrelu(act(a[ci]) + act(b[ci]), slope) -- UnetResBlock's
 * join (dynunet_block.py:96-111) + UnetOutBlock (:251-268) without the joined tensor in HBM; the same bits as mh_add_act_f32 followed by mh_conv1x1_f32.
 * w [Cout][Cin]; _accepts: 1 .. 8 output channels, D*H*W % 4 == 0; 16-byte aligned tensors. */
int mh_conv1x1_sum2_accepts(int Cout, int D, int H, int W);
int mh_conv1x1_sum2_f32(const mh_tensor5* a, const mh_tensor5* b, float slope, const float* w, const float* bias, const mh_tensor5* out, void* stream);

/* Replicate padding at the far end of each axis (out extents = in extents + 0 or 1): `UpCat`'s
 * F.pad(x_0, sp, "replicate") for odd encoder extents -- monai/networks/nets/basic_unet.py:163-170.  Raw copy. */
int mh_pad_replicate_f32(const mh_tensor5* in, const mh_tensor5* out, void* stream);

/* SubpixelUpsample behind its convolution, scale factor 2 -- monai/networks/blocks/upsample.py:186-288 (`pixelshuffle`: monai/networks/utils.py:370-412, `pad_pool`:
 * upsample.py:262-272): in [N][C * fz * 4][D][H][W] (the convolution's raw output) -> out [N][C][fz D][2 H][2 W], sub-voxel (i, j, k) of channel c from input channel
 * c * fz * 4 + i * 4 + j * 2 + k; pad_pool != 0: followed by ConstantPad(1 in front of every spatial axis) + AvgPool(2, stride 1).  fz = 2: three spatial dimensions,
 * fz = 1: two (one plane).  `out->nrm` (if not NULL): identity records, max |value written| goes into their bound. */
int mh_pixelshuffle_f32(const mh_tensor5* in, const mh_tensor5* out, int fz, int pad_pool, void* stream);

/* Self-attention core of SABlock.forward (selfattention.py:156-218): qkv [B][S][3*heads*HD] (the qkv Linear's output, feature index =
 * which*heads*HD + head*HD + d) -> out [B][S][heads*HD] = softmax(Q K^T * scale) V per head, on the fp16 matrix cores in two-piece split precision
 * (fp32-equivalent: hi*hi + lo*hi + hi*lo, fp32 accumulation; |q|, |k|, |v| < 65504), keys / values streamed through LDS in tiles of 32 with an online
 * softmax: ANY sequence length.  head_dim 32, 64, 96 or 128; 16-byte aligned tensors. */
int mh_attention_f32(const float* qkv, float* out, int B, int S, int heads, int head_dim, float scale, void* stream);

/* Window attention of SwinUNETR's Swin transformer (WindowAttention.forward, monai/networks/nets/swin_unetr.py:519-541):
 * softmax((q * scale) k^T + relative_position_bias[head] + mask[window % nW]) v per (window, head).
 *   qkv     [BW][S][3][heads][head_dim]   the qkv Linear's output of BW windows of S tokens
 *   bias_t  [heads][S][S] or NULL         relative position bias, TRANSPOSED: bias_t[h][key][query] = bias[h][query][key]
 *   mask    [nW][S][S] or NULL            shifted-window mask (0 / -100, symmetric); window w uses mask[w % nW]
 *   out     [BW][S][heads * head_dim]
 * head_dim 8 / 16 / 32 (feature sizes 24 / 48 / 96).  Head dims 16 / 32 run on the fp16 matrix cores in two-piece split precision (fp32-equivalent for |q|, |k|, |v| < 65504:
 * no input scaling -- hidden states behind a LayerNorm); exact_fp32 != 0 keeps the exact-fp32 VALU kernel wherever the window's tokens fit its LDS-resident form. */
int mh_window_attention_f32(const float* qkv, const float* bias_t, const float* mask, float* out, int BW, int nW, int S, int heads,
                            int head_dim, float scale, int exact_fp32, void* stream);

/* The same attention with the bias and the mask EVALUATED instead of read from S x S tables (the split-precision kernel, head dims 16 / 32):
 * bias[h][q][k] = rel_table[(coord[q] - coord[k] + coord_off) * heads + h], the reference's relative_position_bias_table gathered through its
 * relative_position_index (swin_unetr.py:492-528; the index is linear in the token coordinates: coord[t] = index[t][0], coord_off = index[0][0]);
 * mask[w][q][k] = region[w % nW][q] == region[w % nW][k] ? 0 : -100, the region ids of compute_mask (swin_unetr.py:774-812) before their pairwise difference.
 *   rel_table [table_rows][heads]; coord [S] (values in [0, 65536)); region [nW][S] (ids in [0, 65536)) or NULL.
 * Bit-identical to mh_window_attention_f32 on the materialised tables.  _accepts: head_dim 16 / 32, S <= 1024, table_rows <= 4096. */
int mh_window_attention_rel_accepts(int S, int head_dim, int table_rows);
int mh_window_attention_rel_f32(const float* qkv, const float* rel_table, int table_rows, const int32_t* coord, int coord_off, const int32_t* region,
                                float* out, int BW, int nW, int S, int heads, int head_dim, float scale, void* stream);

/* nn.Linear of the transformer blocks: y[M][N] = act(x[M][K] . w[N][K]^T + bias) (+ residual[M][N]) -- SABlock.qkv / out_proj
 * (monai/networks/blocks/selfattention.py:105-218), MLPBlock.linear1 -> GELU -> linear2 (mlp.py:56-80), the residual sums of
 * TransformerBlock.forward (transformerblock.py:88-105), PatchEmbeddingBlock's projection of the flattened patches
 * (patchembedding.py:32-142), SwinUNETR's WindowAttention.qkv / proj and Mlp (nets/swin_unetr.py:426-532).  fp32 in and out; the
 * products run on the fp16 matrix cores in two-piece split precision (fp32-equivalent, see mh_conv3d_k3_h2_config).  The weight
 * matrix is packed once per layer (mh_linear_pack_f32 into mh_linear_packed_floats(N, K) floats, 16-byte aligned).
 * act: 0 none, 1 GELU (erf form).  K % 4 == 0; x, packed_w 16-byte aligned; bias and residual may be null. */
int64_t mh_linear_packed_floats(int N, int K);
int mh_linear_pack_f32(const float* w, int N, int K, float* packed, void* stream);
int mh_linear_f32(const float* x, const float* packed_w, const float* bias, const float* residual, float* y, int64_t M, int N, int K, int act,
                  void* stream);
/* The same map with the workgroup tile as an argument: 0 = mh_linear_f32's choice (by how many tiles the problem has), 64 = 128 rows x 64 columns (4 waves),
 * 128 = 128 x 128 (8 waves: 2/3 of the L2 traffic per flop, for the token counts of the ViT blocks).  Both tiles give the same bits. */
int mh_linear_tile_f32(const float* x, const float* packed_w, const float* bias, const float* residual, float* y, int64_t M, int N, int K, int act,
                       int tile, void* stream);

/* nn.LayerNorm over the last dimension (TransformerBlock.norm1 / norm2, ViT.norm, SwinTransformerBlock.norm1 / norm2, PatchMerging.norm):
 * y = (x - mean) * rsqrt(var + eps) * gamma + beta per row, biased variance.  K <= 4096; gamma / beta may be null. */
int mh_layernorm_f32(const float* x, const float* gamma, const float* beta, float eps, float* y, int64_t M, int K, void* stream);

/* SwinTransformerBlock's data movement folded into the two token-wise kernels around the attention (monai/networks/nets/swin_unetr.py:624-672): with
 * row_of[(window, token)] = the voxel row of x [B*D*H*W][C] that (window, token) of the padded, cyclically shifted, window-partitioned volume holds (-1 = padding),
 *   mh_layernorm_gather_f32: y[r] = LayerNorm(x[src_row[r]]) (zeros where src_row[r] < 0) = window_partition(roll(pad(norm1(x)))) in one pass;
 *   mh_linear_scatter_f32:   y[dst_row[m]] = act(x[m] . w^T + bias) + residual[dst_row[m]] (rows with dst_row[m] < 0 dropped)
 *                            = shortcut + crop(roll_back(window_reverse(proj(attention)))) in the projection's epilogue.
 * _gather_accepts: K a multiple of 4, <= 1024; 16-byte aligned tensors.  M_out = rows of y (= entries of src_row); M = rows of x (= entries of dst_row). */
int mh_layernorm_gather_accepts(int K);
int mh_layernorm_gather_f32(const float* x, const float* gamma, const float* beta, float eps, float* y, int64_t M_out, int K, const int32_t* src_row, void* stream);
int mh_linear_scatter_f32(const float* x, const float* packed_w, const float* bias, const float* residual, float* y, int64_t M, int N, int K, int act,
                          const int32_t* dst_row, void* stream);

/* ---- UNet pieces (monai/networks/nets/unet.py:106-298) --------------------------------------------------------- */

/* Conv3d k=3, stride s, padding 1 (+bias) of act(in): the strided `Convolution` / `ResidualUnit` convs of the down path
 * (unet.py:197-237).  packed_w is the configuration-0 layout of mh_conv3d_k3_pack_f32 ([Cin][27][Cout]). */
int mh_conv3d_k3_strided_f32(const mh_tensor5* in, const float* packed_w, const float* bias, const mh_tensor5* out,
                             int stride, void* stream);
/* the same with one stride per axis (DynUNet on anisotropic nnU-Net plans, e.g. strides (1, 2, 2); a kernel extent of 1 along an
 * axis is a 3-tap kernel whose outer taps are zero -- the host expands the weights) */
int mh_conv3d_k3_strided3_f32(const mh_tensor5* in, const float* packed_w, const float* bias, const mh_tensor5* out, int sz, int sy, int sx,
                              void* stream);
/* ConvTranspose3d with kernel == stride == (fz, fy, fx), each 1 or 2 (UnetUpBlock.transp_conv, monai/networks/blocks/dynunet_block.py:188-201);
 * w in torch layout [Cin][Cout][fz][fy][fx]; out = factor * in per axis */
int mh_deconv_ks_f32(const mh_tensor5* in, const float* w, const float* bias, const mh_tensor5* out, int fz, int fy, int fx, void* stream);

/* ConvTranspose3d k=3, stride s, padding 1, output_padding s-1 (+bias) of act(in): the up path (unet.py:249-294).
 * w: torch layout [Cin][Cout][3][3][3].  out dims = s * in. */
int mh_deconv_k3_f32(const mh_tensor5* in, const float* w, const float* bias, const mh_tensor5* out, int stride,
                     void* stream);

/* ---- resampling (Spacingd / SpatialResample / Resample / AffineTransform / grid_pull) -------------------- */

/* Output voxel (oz, oy, ox) samples the source at index  m[row*4+0..2] . (oz, oy, ox) + m[row*4+3]  (rows z, y, x;
 * `m` is 12 HOST doubles: the voxel-space composition of to_norm_affine + F.affine_grid + F.grid_sample's
 * unnormalisation, monai/networks/layers/spatial_transforms.py:564-591, monai/networks/utils.py:243-326).  All NC
 * channel volumes [NC][Di][Hi][Wi] share the coordinates.  mode: 0 nearest, 1 (tri)linear; pad: 0 zeros, 1 border,
 * 2 reflection, with ATen grid_sampler semantics (the rule acts on the coordinate; `align_corners` selects the
 * reflection interval).  compute_f64: interpolation arithmetic in fp64 (the reference's default dtype=float64,
 * monai/transforms/spatial/array.py:141,355) or fp32; input and output are fp32 (functional.py:183).
 * `workspace` (DEVICE, mh_affine_resample_workspace_bytes(Do,Ho,Wo) bytes, may be NULL) enables the separable fast
 * path for axis-aligned matrices: per-axis tap tables are built on the device, then every voxel is 8 loads + 7 blends. */
int64_t mh_affine_resample_workspace_bytes(int Do, int Ho, int Wo);
int mh_affine_resample_f32(const float* src, int NC, int Di, int Hi, int Wi, float* dst, int Do, int Ho, int Wo,
                           const double* m, int mode, int pad, int align_corners, int compute_f64, void* workspace,
                           void* stream);

/* Same sampler with an explicit coordinate field `coords` [3][Do][Ho][Wo] (planes z, y, x; fp32 or fp64 DEVICE
 * memory) -- the torch branch of `Resample.__call__` (monai/transforms/spatial/array.py:2101-2116).  The source index
 * along axis a is  scale3[a] * coord + offset3[a]  (HOST doubles, NULL = 1 / 0): the reference's `norm_coords`
 * scaling and grid_sample's unnormalisation folded into one affine per axis, so the grid is never rewritten. */
int mh_grid_resample_f32(const float* src, int NC, int Di, int Hi, int Wi, const void* coords, int coords_f64,
                         const double* scale3, const double* offset3, float* dst, int Do, int Ho, int Wo, int mode,
                         int pad, int align_corners, int compute_f64, void* stream);

/* monai._C.grid_pull (monai/csrc/ext.cpp:67, monai/csrc/resample/pushpull.h:58-110) for 3-D volumes: B-spline
 * interpolation orders 0-7 and boundary conditions replicate(0) dct1(1) dct2(2) dst1(3) dst2(4) dft(5) zero(7) per
 * axis, `extrapolate` as in the reference (out-of-FOV samples, tolerance 5e-2, are zero when false).  src (B,C,X,Y,Z),
 * grid (B,Xo,Yo,Zo,3) with the last axis holding voxel coordinates in tensor-axis order, out (B,C,Xo,Yo,Zo); all three
 * fp32 or all three fp64 (is_f64), dense.  bound3 / interp3 are HOST int32[3].  Shorthand for mh_pushpull(ndim = 3,
 * do_pull); 1-D / 2-D problems go through mh_pushpull, which reproduces the reference's lower-dimensional arithmetic. */
int mh_grid_pull(const void* src, const void* grid, void* out, int is_f64, int B, int C, int X, int Y, int Z, int Xo,
                 int Yo, int Zo, const int32_t* bound3, const int32_t* interp3, int extrapolate, void* stream);

/* The dispatcher behind every other monai._C resampling entry point -- grid_pull_backward, grid_push(_backward),
 * grid_count(_backward), grid_grad(_backward) (monai/csrc/ext.cpp:67-74, monai/csrc/resample/pushpull.h:112-509) -- with
 * the reference's own flag set (`pushpull(source, grid, target, bound, interpolation, extrapolate, do_pull, do_push,
 * do_count, do_grad, do_sgrad)`, pushpull.h:24-50) and B-spline interpolation orders 0-7 per axis.
 *   ndim     real spatial axes (1-3); tensors are passed as 3-D with trailing size-1 axes, vector axes have ndim entries
 *   source   (B,C,X,Y,Z) for pull / sgrad / grad, NULL for push / count (the volume being splatted into is `out`)
 *   grid     (B,Xo,Yo,Zo,ndim) voxel coordinates in tensor-axis order
 *   target   NULL, (B,C,Xo,Yo,Zo) or -- target_k != 0 -- (B,C,Xo,Yo,Zo,ndim)
 *   out      do_pull: (B,C,Xo,Yo,Zo); do_sgrad: (B,C,Xo,Yo,Zo,ndim); do_push: (B,C,X,Y,Z); do_count: (B,1,X,Y,Z) (C is
 *            ignored); push / count outputs are zero-filled by the call; NULL when none of the four is set
 *   grad     do_grad: (B,Xo,Yo,Zo,ndim) gradient with respect to the grid, else NULL
 * At most one of do_pull / do_sgrad / do_push / do_count, optionally combined with do_grad, as the reference calls it.
 * All tensors fp32 or all fp64 (is_f64), dense.  bound3 / interp3: HOST int32[3]; entries beyond ndim only take part in
 * the reference's "all three orders equal" test that selects the nearest / linear fast paths (pushpull_cpu.cpp:136);
 * like the reference's kernels (pushpull_cpu.cpp:501) the third axis is interpolated with the SECOND entry's order.
 * push / count scatter with floating-point atomics (summation order undefined, as in the reference's GPU path). */
int mh_pushpull(const void* source, const void* grid, const void* target, void* out, void* grad, int is_f64, int ndim, int B,
                int C, int X, int Y, int Z, int Xo, int Yo, int Zo, const int32_t* bound3, const int32_t* interp3,
                int extrapolate, int do_pull, int do_push, int do_count, int do_grad, int do_sgrad, int target_k, void* stream);

/* ---- post-processing (Activations / AsDiscrete, monai/transforms/post/array.py:61-237) ----------------------- */

/* element-wise: op 0 = sigmoid, 1 = (x >= param) as 0/1, 2 = round half to even (torch.round) */
int mh_pointwise_f32(int op, const float* src, float* dst, int64_t n, float param, void* stream);
/* over the channel axis of a channel-first tensor [C][n]: op 0 = argmax (first maximal index, NaN maximal; written as
 * float into dst[n]), 1 = softmax (dst [C][n]) */
int mh_channel_reduce_f32(int op, const float* src, float* dst, int C, int64_t n, void* stream);
/* one_hot (monai/networks/utils.py:170-221): float labels [n] -> dst [K][n] of 0/1 */
int mh_onehot_f32(const float* labels, float* dst, int K, int64_t n, void* stream);

/* ---- pre-processing in front of the path ------------------------------------------------- */

/* ScaleIntensityRange (monai/transforms/intensity/array.py:958-1012):
 *   y = (x - a_min) / a_div;  if (rescale) y = y * b_scale + b_min;  if (clip_lo) y = max(y, lo);  if (clip_hi) y = min(y, hi)
 * with the reference's roundings (true division, separate multiply and add; NaN passes the clamp like torch.clamp).
 * The a_min == a_max branch of the reference (no division) is a_div = 1, b_scale = 1 by the host side. */
int mh_scale_intensity_range_f32(const float* src, float* dst, int64_t n, float a_min, float a_div, int rescale, float b_scale, float b_min,
                                 int clip_lo, float lo, int clip_hi, float hi, void* stream);
/* generate_spatial_bounding_box with the default select_fn (monai/transforms/utils.py:1069-1129): box6 (DEVICE int32[6]) =
 * {zmin, ymin, xmin, zmax, ymax, xmax} (inclusive) of the voxels of src [C][D][H][W] where any channel is > 0; zmax == -1
 * when there is no foreground.  workspace: DEVICE int32[mh_foreground_bbox_workspace_ints(D, H)], caller-owned. */
int mh_foreground_bbox_workspace_ints(int D, int H);
int mh_foreground_bbox_f32(const float* src, int C, int D, int H, int W, int32_t* workspace, int32_t* box6, void* stream);
/* CropForeground.crop_pad (monai/transforms/croppad/array.py:884-927), constant mode: dst [C][Do][Ho][Wo],
 * dst[c][z][y][x] = src[c][z+sz][y+sy][x+sx] inside src [C][D][H][W], `value` outside (starts may be negative). */
int mh_crop_pad_f32(const float* src, float* dst, int C, int D, int H, int W, int Do, int Ho, int Wo, int sz, int sy, int sx, float value,
                    void* stream);

/* NormalizeIntensity (monai/transforms/intensity/array.py:816-907).  src = C runs of n contiguous values (C = 1: whole image;
 * C = channels: channel_wise).  _stats: subdiv (DEVICE float[C][2]) = {mean, population std (0 -> 1)} of all / the non-zero
 * values of each run, fp64 accumulation; workspace: DEVICE double[mh_normalize_stats_workspace_doubles(C, n)], caller-owned.
 * _apply: dst = (src - subdiv[c][0]) / subdiv[c][1]; with nonzero != 0 zeros pass through unchanged.  No host synchronisation. */
int64_t mh_normalize_stats_workspace_doubles(int C, int64_t n);
int mh_normalize_stats_f32(const float* src, int C, int64_t n, int nonzero, double* workspace, float* subdiv, void* stream);
int mh_normalize_apply_f32(const float* src, float* dst, int C, int64_t n, int nonzero, const float* subdiv, void* stream);
/* ScaleIntensity (monai/transforms/intensity/array.py:445-491; rescale_array, monai/transforms/utils.py:229-251).  src = C runs of n values.
 * mh_minmax_f32: table (DEVICE float[C][2]) = {min, max} of each run (NaN if the run holds one, as torch.min / max); workspace: DEVICE
 * float[mh_minmax_workspace_floats(C, n)].  mh_minmax_scale_f32: dst = (src - min) / (max - min) [* b_scale + b_min when rescale]; a run
 * with min == max takes the reference's branch instead: src * flat_mul when flat_has_mul, else src.  No host synchronisation. */
int64_t mh_minmax_workspace_floats(int C, int64_t n);
int mh_minmax_f32(const float* src, int C, int64_t n, float* workspace, float* table, void* stream);
int mh_minmax_scale_f32(const float* src, float* dst, int C, int64_t n, const float* table, int rescale, float b_scale, float b_min,
                        int flat_has_mul, float flat_mul, void* stream);
/* Orientation (monai/transforms/spatial/functional.py:187-229: torch.flip over the reversed axes, then permute):
 * src [C][in_size3] -> dst [C][out], out axis k = input axis perm3[k] (HOST int32[3], a permutation of 0..2), input axis a
 * read backwards when flip3[a] != 0 (HOST int32[3]).  Images with fewer spatial axes pass leading extents of 1. */
int mh_flip_permute_f32(const float* src, float* dst, int C, const int32_t* in_size3, const int32_t* perm3, const int32_t* flip3, void* stream);

/* ---- Gaussian smoothing (GaussianSmooth / GaussianFilter / separable_filtering) ----------------------------- */

/* dst = src convolved with kz (x) ky (x) kx, zero padding, per channel volume [NC][D][H][W]
 * (monai/networks/layers/simplelayers.py:170-249; kernels from gaussian_1d, layers/convutils.py:78-131).  The three
 * 1-D kernels are HOST float arrays with odd tap counts (at most 33 each; a single tap 1.0 skips an axis, 2-D
 * images are passed with D = 1 and kz = {1}).  One fused pass: every voxel is read once and written once. */
int mh_separable_filter3d_f32(const float* src, float* dst, int NC, int D, int H, int W, const float* kz, int kz_n,
                              const float* ky, int ky_n, const float* kx, int kx_n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MONAI_AMD_H */
