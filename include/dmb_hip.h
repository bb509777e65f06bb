/*
 * dmb_hip.h -- C ABI of libdmb_hip.so: the MI355X (gfx950) cost-volume -> 3-D aggregation
 * -> disparity-regression path of DenseMatchingBenchmark's dmb.modeling.stereo pipeline.
 *
 * Boundary rules:
 *   - every pointer is a DEVICE pointer owned by the caller (the PyTorch-ROCm caching
 *     allocator in the Python host layer) unless the parameter name ends in `_host`;
 *   - all tensors are FP32, contiguous, NC(D)HW exactly as the reference lays them out;
 *   - `stream` is the caller's hipStream_t (NULL = the legacy default stream); every entry
 *     point only enqueues kernel launches on that stream: no synchronisation, no allocation, no
 *     memset / memcpy -- so a sequence of calls can be captured into a HIP graph and replayed
 *     (tests/test_graph_gpu.py captures a whole PSMNet step).  Scratch memory a kernel needs is a
 *     caller-provided workspace argument (dmb_*_workspace_bytes / DMB_*_WORKSPACE_BYTES), as the
 *     reference's own native op takes caller-allocated tensors only
 *     (dmb/ops/spn/functions/gaterecurrent2dnoind.py:8-39);
 *   - return value 0 = success, otherwise the hipError_t value of the failing launch or one of
 *     the DMB_E* codes below.  Nothing here calls exit() (the reference's only native op,
 *     dmb/ops/spn/src/gaterecurrent2dnoind_kernel.cu:544-549, does; this library does not);
 *   - host-side state, all of it: a thread-local pointer to the last error string, and two
 *     per-device read-mostly caches filled on first use (the device's compute-unit count; which
 *     kernels have had their dynamic-LDS limit raised on which device, a per-device attribute
 *     that hipFuncSetAttribute requires before a launch with more than 64 KB of LDS).  No
 *     option table, no device allocations, nothing a launch depends on besides its arguments:
 *     calls from several host threads on distinct streams are safe (the workspace of
 *     dmb_deconv3d_k3s2_f32 must then be distinct per stream, see there).  The release library
 *     exports exactly the functions declared here (tests/test_host_logic.py compares the two lists);
 *     kernel-variant switches exist only in the development build (build.py dev=True ->
 *     lib/libdmb_hip_dev.so, used by scripts/, never by the package, the tests or bench.py).
 *
 * Each entry point names the reference interface it replaces (file:line under the reference
 * tree).  The reference is pure PyTorch on this path, so "replaces" means: the same
 * arithmetic that those torch calls perform, as one hand-written HIP kernel launch.
 */
#ifndef DMB_HIP_H
#define DMB_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define DMB_OK 0
#define DMB_EINVAL 100001   /* bad argument (NULL pointer, non-positive size, D too large ...) */
#define DMB_EUNSUPPORTED 100002 /* shape outside what the kernels are instantiated for */

#define DMB_MAX_DISP_SAMPLES 256 /* upper bound on the number of disparity samples D */

/* ABI version: bumped whenever a signature below changes (8: dmb_bn_train_fwd_f32, dmb_catconv_pack_weights_f32, dmb_conv3d_pack_weights_multi_f32, dmb_cat_first_wgrad_maps_f32, dmb_conv3d_k3_bnstats_f32 and dmb_bn_train_act_f32 added, dmb_bn_act_bwd_f32 takes the gradient its
 * skip operand already holds (`dres_acc`); 7: DMB_CONV_SINGLE_CHAIN in the `relu` argument of the convolution
 * entry points, `flags` argument of dmb_conv3d_k3_c1_f32; 6: dmb_stereo_pad_normalize_f32 / _u8 added; 5: dmb_fast_fms_bwd_f32 takes a mode, the forward's norm and an
 * optional gradient buffer for per-pixel samples; 4: workspace argument of dmb_deconv3d_k3s2_f32, the merged-heads entry
 * points of version 3 removed). */
int dmb_abi_version(void);
/* Static string describing the last DMB_E* code returned on this thread ("" if none). */
const char* dmb_last_error(void);
/* sha256 (hex) of the sources this binary was built from: every translation unit, the shared headers, this file and the
 * compiler flags (densematchingbenchmark_amd/build.py::sources_digest).  The Python binding refuses a library whose id does
 * not match the sources next to it -- a prebuilt library that travels with a snapshot is then PROVEN to be built from that
 * snapshot's sources, not merely assumed to be. */
const char* dmb_build_id(void);

/* ------------------------------------------------------------------------------------------
 * Cost-volume builders
 * ---------------------------------------------------------------------------------------- */

/* cat_fms: dmb/modeling/stereo/cost_processors/utils/cat_fms.py:7-48.
 *   out[b, c,   k, y, x] = L[b, c, y, x]        if in range else 0
 *   out[b, C+c, k, y, x] = R[b, c, y, x - d_k]  if in range else 0
 * "in range": d_k > 0 -> x >= d_k;  d_k == 0 -> all x;  d_k < 0 -> x < W + d_k  (cat_fms.py:36-44)
 * L, R: [B, C, H, W]; out: [B, 2C, D, H, W]; disp_idx_host: D ints on the HOST, d_k as produced by
 * int(linspace(start, start+max_disp-1, D)[k]) (cat_fms.py:28-35).  Pure copy: bit-exact. */
int dmb_cat_fms_f32(const float* L, const float* R, float* out, int B, int C, int H, int W, int D,
                    const int* disp_idx_host, void* stream);

/* dif_fms: dmb/modeling/stereo/cost_processors/utils/dif_fms.py:7-46.
 *   out[b, c, k, y, x] = L[b, c, y, x] - R[b, c, y, x - d_k] if in range else 0;  out: [B, C, D, H, W]. */
int dmb_dif_fms_f32(const float* L, const float* R, float* out, int B, int C, int H, int W, int D,
                    const int* disp_idx_host, void* stream);

/* fast_cat_fms: dmb/modeling/stereo/cost_processors/utils/cat_fms.py:51-82 (CAT_FUNCS["fast_mode"]; the builder
 * DeepPruner.py:192 calls with per-pixel samples) on layers/inverse_warp_3d.py:4-52.
 *   T[b, c, k, y, x]     = tri-linear blend of R[b, c] around ix = (x - s) * W / (W - 1) - 0.5, iy = y * H / (H - 1) - 0.5,
 *                          iz = k * D / (D - 1) - 0.5, zero padding (F.grid_sample on a (size - 1)-normalised grid with
 *                          align_corners=False: what the reference computes on torch >= 1.3), s = disp_sample
 *   out[b, c,     k, y, x] = L[b, c, y, x] * (T > 0)        (cat_fms.py:77)
 *   out[b, C + c, k, y, x] = T
 * disp_sample (DEVICE): [B, D, H, W] if per_pixel else [D] (cat_fms.py:55-63: linspace(start, end, D), not truncated).
 * FP32 operation for operation as the reference on CPU: bit-exact.  D, H, W >= 2 (the reference divides by size - 1). */
int dmb_fast_cat_fms_f32(const float* L, const float* R, const float* disp_sample, float* out, int B, int C, int D, int H,
                         int W, int per_pixel, void* stream);

/* fast_dif_fms: dif_fms.py:49-86 (DIF_FUNCS["fast_mode"]; AnyNet.py:73).  out[b, c, k, y, x] = L * (T > 0) - T with T as
 * above; normalize != 0: out[b, k, y, x] = || . ||_p over the channels (dif_fms.py:80-82), out is [B, D, H, W]. */
int dmb_fast_dif_fms_f32(const float* L, const float* R, const float* disp_sample, float* out, int B, int C, int D, int H,
                         int W, int per_pixel, int normalize, float p, void* stream);

/* Backward of fast_cat_fms (mode 0), fast_dif_fms (mode 1) and fast_dif_fms(normalize=True) (mode 2): what the reference
 * obtains from torch.autograd through F.grid_sample, the expand of inverse_warp_3d.py:19-20 and torch.norm (cat_fms.py:51-82,
 * dif_fms.py:49-86).
 *   dvol: gradient of the builder's output ([B, 2C, D, H, W] / [B, C, D, H, W] / [B, D, H, W]);  dL, dR: [B, C, H, W];
 *   norm_out: mode 2 only (else NULL): the forward's output [B, D, H, W], with its p;
 *   d_samples: NULL, or [B, D, H, W] = the gradient with respect to per-pixel samples (per_pixel != 0 only: the sampler's
 *     column derivative times -(W / 2) * 2 / (W - 1), summed over the channels -- the path AnyNet.py:60-73 and
 *     DeepPruner.py:192 train through);
 *   partial: workspace of B * C * H * 2 * W floats (per output row the gradient rows of its two source rows).
 *   dL = sum_k dvol_ref * (T > 0);  dR = the sampler's adjoint of dvol_tgt (cat) or of -dvol (dif); the mask (T > 0) is a
 *   constant, as in the reference (cat_fms.py:77).
 * Sums along x go through LDS atomics (their order is the hardware's, as in the reference's own GPU backward).  Same
 * D, H, W >= 2 rule as the forward; W <= 1024 (two groups of 8 channel rows in 64 KiB of LDS: DMB_EUNSUPPORTED beyond -- the
 * forward has no such limit, the host layer refuses wider maps under autograd up front). */
int dmb_fast_fms_bwd_f32(const float* L, const float* R, const float* disp_sample, const float* dvol, const float* norm_out,
                         float* dL, float* dR, float* d_samples, float* partial, int B, int C, int D, int H, int W,
                         int per_pixel, int mode, float p, void* stream);

/* Spatial propagation scan: dmb/ops/spn (GateRecurrent2dnoind: functions/gaterecurrent2dnoind.py:10-44 on
 * src/gaterecurrent2dnoind_kernel.cu:10-166,288-345,535-552), the reference's only native op (CUDA; one kernel launch per
 * scanned line there, ONE launch here).  X, G1, G2, G3, H: [N, C, H, W].
 *   H[s, t] = (1 - g1 - g2 - g3) * X[s, t] + g1 * H[s', t - 1] + g2 * H[s', t] + g3 * H[s', t + 1],   g_k = G_k[s, t] where
 *   the neighbour lies inside the image, else 0; s along the columns if horizontal else along the rows, s' = s - 1 (s + 1 if
 *   reverse).  The line across the scan direction may hold at most 2046 positions (forward and backward alike).
 * Backward: dH = gradient of the output; writes dX, dG1, dG2, dG3 (zero where a link leaves the image).  Unlike the reference
 * (kernel.cu:317) it does not overwrite dH.  Parity UNPINNED: the reference op cannot be built here (CUDA). */
int dmb_spn_gaterecurrent2d_f32(const float* X, const float* G1, const float* G2, const float* G3, float* H_out, int N, int C,
                                int H, int W, int horizontal, int reverse, void* stream);
int dmb_spn_gaterecurrent2d_bwd_f32(const float* X, const float* G1, const float* G2, const float* G3, const float* H_fwd,
                                    const float* dH, float* dX, float* dG1, float* dG2, float* dG3, int N, int C, int H, int W,
                                    int horizontal, int reverse, void* stream);

/* Group-wise correlation volume (GwcNet).  ABSENT from the reference (README.md:16 only names it);
 * occupies the COR_FUNCS slot of cost_processors/utils/correlation1d_cost.py:29-31.  Spec:
 *   out[b, g, k, y, x] = (1/(C/G)) * sum_{c in group g} L[b,c,y,x] * R[b,c,y,x-d_k]  if in range else 0
 * L, R: [B, C, H, W], C % G == 0; out: [B, G, D, H, W] written into a tensor whose channel count is
 * `out_channels` at channel offset `out_ch_offset` (so the volume can be produced directly inside a
 * wider concatenated volume).  Sum order over the group's channels: ascending c, FP32 fma chain. */
int dmb_gwc_fms_f32(const float* L, const float* R, float* out, int B, int C, int G, int H, int W, int D,
                    const int* disp_idx_host, int out_channels, int out_ch_offset, void* stream);

/* First convolution of the aggregators (aggregators/PSMNet.py:31-33, AcfNet.py:28-31: dres0[0] = Conv3d(2C, Co, 3,
 * padding 1) + BatchNorm3d + ReLU) applied to the concatenation volume of cat_fms (cat_fms.py:7-48) with d_k = k, WITHOUT
 * the volume: the left half of that volume does not depend on z and the right half depends on (z, x) through x - z, so the
 * layer is a sum of 2-D maps (3x3 convolutions of the two feature maps with the dz slices of the weight: dmb_conv2d_f32)
 * looked up at x and at x - (z + dz - 1); see csrc/catconv.hip.  Same FP32 products, summed per dz.
 *
 * dmb_copy_window_f32: dst[r, j] = src[r, j + xs] (0 outside [0, W)), j in [0, Wd): the zero-extended / cropped feature
 *   rows the 2-D convolutions run on.
 * dmb_catconv_finalize_f32: per-dz maps -> sums over dz.  FA [B, CA, H, W] (channel dz*Co + co: left half, all taps),
 *   FB [B, CB, H, Wc] (channel (m-1)*CB/2 + dz*Co + co: left half, taps dx >= m, m = 1, 2, columns [0, Wc)),
 *   HC [B, CA, H, W+4] (right half at n = j - 4), HD [B, CA, H, Wc] (right half without the dx = 2 tap at n = j + W - Wc)
 *   -> FM [B, Co, H, W] and GM [B, Co, H, W+4] (interior planes), BAND [B, Co, H, D, 4] (left half at x = z - 2 .. z + 1 of
 *   plane z), GB [B, Co, H, D] (right half at x = W - 1 of plane z).  W % 4 == 0, D >= 3, D + 2 <= Wc <= W.
 * dmb_catconv_combine_f32: out[b, co, z, y, x] = act(scale[co] * (f + g) + shift[co]) -> [B, Co, D, H, W], one pass
 *   (planes 0 and D-1 are summed from FA / HC on the fly).  D % 4 == 0, W >= D + 8, Wc = D + 4.
 * dmb_catconv_pack_weights_f32 (ABI 8): the five conv2d weight packs of that form from the layer's weight [Co, 2C, 3, 3, 3]
 *   (dif != 0: [Co, C, 3, 3, 3] of a layer on the difference volume of dif_fms.py:7-46, right half = -w) in one launch: `packs`
 *   = 5 consecutive packs of dmb_conv2d_packed_floats(CA, C, 3) floats in the order A, B1, B2 (left half, dx taps from 0 / 1 /
 *   2), HC, HD (right half, dx taps up to 2 / 1), rows dz * Co + co, zero rows from 3 Co on (3 Co <= CA in {32, 64, 128}).
 *   A training step re-packs after every optimizer update: as torch slicing this was 25 launches. */
int dmb_catconv_pack_weights_f32(const float* w, float* packs, int Co, int C, int CA, int dif, void* stream);
int dmb_copy_window_f32(const float* src, float* dst, long long rows, int W, int Wd, int xs, void* stream);
/* t[r, x0 .. pitch) = 0 for rows r of `pitch` floats: re-zeroes the padding columns of a row-padded tensor after a convolution has
 * run over it as if they were image columns (see dmb_deconv3d_k3s2_f32, `Wout`). */
int dmb_zero_columns_f32(float* t, long long rows, int pitch, int x0, void* stream);
int dmb_catconv_finalize_f32(const float* FA, const float* FB, const float* HC, const float* HD, float* FM, float* BAND,
                             float* GM, float* GB, int B, int Co, int CA, int CB, int D, int H, int W, int Wc, void* stream);
int dmb_catconv_combine_f32(const float* FA, const float* HC, const float* FM, const float* BAND, const float* GM,
                            const float* GB, const float* scale, const float* shift, float* out, int B, int Co, int CA, int D,
                            int H, int W, int relu, void* stream);

/* correlation1d_cost (cost_processors/utils/correlation1d_cost.py:7-27), the reference's COR_FUNCS['default']:
 *   out[b, j, y, x] = leaky_relu( sum_c L[b,c,y,x] * R[b,c,y, x + j - (D-1)], negative_slope ),  0 <= j < D
 * i.e. the first D of the 2D-1 patch offsets of SpatialCorrelationSampler(kernel_size 1, patch_size (1, 2D-1), stride 1,
 * padding 0, dilation_patch 1) -- channel j is disparity D-1-j, R is 0 outside the image, no 1/C normalisation.
 * L, R: [B, C, H, W]; out: [B, D, H, W] (4-D: this volume feeds 2-D aggregators).  The sampler is a third-party
 * package absent from the reference tree (ClementPinard/Pytorch-Correlation-extension, branch fix_1.7 per
 * INSTALL.md:60-66, unpinned in requirements.txt): restated from its published semantics, parity UNPINNED. */
int dmb_correlation1d_f32(const float* L, const float* R, float* out, int B, int C, int H, int W, int D,
                          float negative_slope, void* stream);

/* Same as dmb_cat_fms_f32 but writing into channels [out_ch_offset, out_ch_offset+2C) of a volume with
 * `out_channels` channels (GwcNet's gwc+concat volume). */
int dmb_cat_fms_into_f32(const float* L, const float* R, float* out, int B, int C, int H, int W, int D,
                         const int* disp_idx_host, int out_channels, int out_ch_offset, void* stream);

/* ------------------------------------------------------------------------------------------
 * 3-D convolution family (the aggregators' layers)
 *
 * Replaces nn.Conv3d / nn.ConvTranspose3d + nn.BatchNorm3d (eval) + residual add + ReLU as composed by
 * dmb/modeling/stereo/layers/basic_layers.py:68-100,160-177, cost_processors/utils/hourglass.py:62-86 and
 * cost_processors/aggregators/{PSMNet.py:55-95, AcfNet.py:59-90, StereoNet.py:42-55}.
 *
 * Epilogue, in this order (matches the reference's op order):
 *     v = acc * scale[co] + shift[co]      (scale/shift may be NULL -> 1 / 0; BN and bias folded by caller)
 *     v = max(v, 0)                        (if relu == 2: GC-Net adds its skips to the ACTIVATED output,
 *                                           aggregators/GCNet.py:108-116)
 *     v = v + residual[...]                (residual may be NULL; same shape as y)
 *     v = max(v, 0)                        (if relu == 1: hourglass.py:67-81 activates after the skip add)
 * acc is an FP32 fma chain over (ci, kd, kh, kw) -- ONE chain per output voxel in the kernels a launch that fills the chip
 * takes.  A launch that would leave most of the chip idle (one small stereo pair per call: dmb/apis/inference.py:191-225)
 * takes a split-K form instead (csrc/conv3d_sk.hip, conv3d_c1s_kernel): the input channels of a voxel are split over the waves
 * of a workgroup and the partial chains added in a fixed order -- reproducible run to run, but the last bits then depend on which
 * form the launch's SIZE selects (batch 1 and batch 4 of the same pair may differ by an FP32 rounding of the sum).
 * DMB_CONV_SINGLE_CHAIN, or-ed into the `relu` argument of dmb_conv3d_k3_f32 / dmb_deconv3d_k3s2_f32 / dmb_conv2d_f32 (bits 0-7
 * stay the activation mode) or passed as `flags` of dmb_conv3d_k3_c1_f32, keeps a launch on the single-chain kernels whatever its size:
 * results are then bit-identical across batch sizes (slower for small launches: 31-35 us instead of 13-15 us per layer of the
 * deepest hourglass level of one 256x512 pair).
 * ---------------------------------------------------------------------------------------- */
#define DMB_CONV_SINGLE_CHAIN 0x100

/* Number of floats of the packed-weight buffer for a k=3 convolution / transposed convolution (input channels are
 * zero-padded to a multiple of 8 inside the packed stream, so any Ci >= 1 is accepted). */
long long dmb_conv3d_packed_floats(int Co, int Ci);
long long dmb_deconv3d_packed_floats(int Ci, int Co);

/* Re-order nn.Conv3d weights [Co, Ci, 3, 3, 3] into the MFMA B-fragment stream the kernels read. */
int dmb_conv3d_pack_weights_f32(const float* w, float* wpack, int Co, int Ci, void* stream);
/* Re-order nn.ConvTranspose3d weights [Ci, Co, 3, 3, 3] (stride 2, pad 1, output_padding 1). */
int dmb_deconv3d_pack_weights_f32(const float* w, float* wpack, int Ci, int Co, void* stream);

/* Conv3d, kernel 3, padding 1, stride 1 or 2 (all three axes).  x: [B, Ci, D, H, W];
 * y: [B, Co, Do, Ho, Wo] with Do = (D - 1) / stride + 1 etc.  Co in {32, 64, 128}: MFMA implicit GEMM using
 * wpack from dmb_conv3d_pack_weights_f32.  Any Ci >= 1.  8 channels of one batch item must stay below 2 GiB
 * (buffer resources cover one channel chunk at a time, so the tensors themselves may be larger). */
int dmb_conv3d_k3_f32(const float* x, const float* wpack, const float* scale, const float* shift,
                      const float* residual, float* y, int B, int Ci, int Co, int D, int H, int W,
                      int stride, int relu, void* stream);

/* Conv3d kernel 3 pad 1 stride 1 with ONE output channel (classifier heads: PSMNet.py:46,50,54,
 * StereoNet.py:39).  w: raw [1, Ci, 3, 3, 3]; bias_host: scalar added to every output; residual may be
 * NULL (PSMNet.py:71-72 adds the previous level's cost).  y: [B, 1, D, H, W].  flags: 0 or DMB_CONV_SINGLE_CHAIN. */
int dmb_conv3d_k3_c1_f32(const float* x, const float* w, float bias, const float* residual, float* y,
                         int B, int Ci, int D, int H, int W, int flags, void* stream);

/* ConvTranspose3d kernel 3, stride 2, padding 1, output_padding 1 (hourglass.py:52-60):
 * x: [B, Ci, D, H, W] -> y: [B, Co, 2D, 2H, Wout];  y[o] += x[i] * w[k] with o = 2i - 1 + k.  Co = 64 or any Co <= 32
 * (fewer than 32: zero-padded weight rows, e.g. GC-Net's 1-channel output layer, aggregators/GCNet.py:63-67).
 * Wout: the output (and residual) row length: 2W -- or, for an input whose rows are ZERO-PADDED on the right to a 16-byte multiple
 * (an image of Wi columns stored with W = Wi rounded up to a multiple of 4: the hourglass's deepest level at the KITTI shape is
 * 78 columns wide), Wout = 2 Wi < 2W with Wout % 4 == 0: only output columns below Wout exist.  Such a padded input must
 * hold zeros in its padding columns (dmb_copy_window_f32 / dmb_zero_columns_f32); it keeps the layer on the 16-byte paths.
 * Wout < 2W needs the workspace form (Co = 32 or 64, Ci % 16 == 0, aligned operands), otherwise DMB_EUNSUPPORTED.
 * workspace: DMB_DECONV3D_WORKSPACE_BYTES of device memory, 4-byte aligned, holding ZEROS (the caller zeroes it once, e.g.
 * at allocation); the launch uses it for its work-item counters and leaves it zeroed again, so the same workspace serves
 * every later call on that stream, eager or replayed from a captured graph.  Launches that may be in flight at the same
 * time (different streams) need different workspaces.  NULL selects the kernel form without counters (static tile walk:
 * same results bit for bit, slower). */
#define DMB_DECONV3D_WORKSPACE_BYTES 2048
int dmb_deconv3d_k3s2_f32(const float* x, const float* wpack, const float* scale, const float* shift,
                          const float* residual, float* y, int B, int Ci, int Co, int D, int H, int W, int Wout,
                          int relu, void* workspace, void* stream);

/* ------------------------------------------------------------------------------------------
 * Cost up-sampling
 * ---------------------------------------------------------------------------------------- */

/* F.interpolate(mode='trilinear', align_corners=True) of a 1-channel volume (PSMNet.py:77-93):
 * x: [B, Di, Hi, Wi] -> y: [B, Do, Ho, Wo];  src = dst * (in-1)/(out-1) per axis. */
int dmb_trilinear_ac_f32(const float* x, float* y, int B, int Di, int Hi, int Wi, int Do, int Ho, int Wo,
                         void* stream);

/* nn.ConvTranspose3d(1, 1, kernel 8, stride 4, padding 2, bias=False) called with output_size =
 * [4D, 4H, 4W] (AcfNet.py:55-57,81-83): y[o] = sum x[i] * w[k], o = 4i - 2 + k.  w: [8, 8, 8]. */
int dmb_deconv3d_k8s4_c1_f32(const float* x, const float* w, float* y, int B, int D, int H, int W,
                             void* stream);

/* The same up-sampling with the standard soft-argmin of the volume it writes folded in (what dmb_trilinear_ac_soft_argmin_f32 is
 * to the tri-linear up-sampling): y [B, 4D, 4H, 4W] and disp [B, 1, 4H, 4W] = soft_argmin(y, samples, alpha, normalize) bit
 * for bit, one pass.  disp may be NULL (up-sampling only, z-column form).  Per-output arithmetic = dmb_deconv3d_k8s4_c1_f32. */
int dmb_deconv3d_k8s4_c1_soft_argmin_f32(const float* x, const float* w, float* y, float* disp, int B, int D, int H, int W,
                                         float alpha, const float* disp_sample_host, void* stream);

/* ------------------------------------------------------------------------------------------
 * Disparity regression
 * ---------------------------------------------------------------------------------------- */

/* SoftArgmin / FasterSoftArgmin with the module's own linspace samples
 * (disp_predictors/soft_argmin.py:45-75, faster_soft_argmin.py:51-75):
 *   p = softmax(alpha * cost, dim=1) if normalize else alpha * cost;  disp = sum_k p_k * (start + k*step)
 * with the sample values passed as a host array (they are linspace(start, end, D) in FP32).
 * cost: [B, D, H, W] -> disp: [B, 1, H, W]. */
int dmb_soft_argmin_f32(const float* cost, float* disp, int B, int D, int H, int W, float alpha,
                        int normalize, const float* disp_sample_host, void* stream);

/* SoftArgmin with a per-pixel disparity-sample tensor (soft_argmin.py:67-72): sample: [B, D, H, W]. */
int dmb_soft_argmin_sampled_f32(const float* cost, const float* sample, float* disp, int B, int D, int H,
                                int W, float alpha, int normalize, void* stream);

/* LocalSoftArgmin (disp_predictors/local_soft_argmin.py:48-105).  argidx (may be NULL) receives the
 * arg-max index per pixel as int64 [B, 1, H, W] -- the bit-exact "index path" (first maximal index,
 * torch.argmax semantics). */
int dmb_local_soft_argmin_f32(const float* cost, float* disp, long long* argidx, int B, int D, int H, int W,
                              int radius, int radius_dilation, int start_disp, int dilation, float alpha,
                              void* stream);

/* Producer-fused form of the two calls above: F.interpolate(..., 'trilinear', align_corners=True) AND the soft-argmin
 * (normalize=True, disp_sample given per plane) of the up-sampled volume in one pass: y [B, Do, Ho, Wo] is written as
 * by dmb_trilinear_ac_f32 and disp [B, 1, Ho, Wo] equals dmb_soft_argmin_f32(y) bit for bit, without re-reading y
 * (aggregators/PSMNet.py:75-88 followed by disp_predictors/faster_soft_argmin.py:51-71). */
int dmb_trilinear_ac_soft_argmin_f32(const float* x, float* y, float* disp, int B, int Di, int Hi, int Wi, int Do,
                                     int Ho, int Wo, float alpha, const float* disp_sample_host, void* stream);

/* Opt-in fused fast path: trilinear(align_corners=True) up-sampling of the 1/4-resolution cost
 * [B, Di, Hi, Wi] to [B, Do, Ho, Wo] + soft-argmin, without materialising the full-resolution volume.
 * Legal only when the caller does not need `costs` back (SURVEY 7.3 "costs are part of the return
 * contract"). */
int dmb_trilinear_soft_argmin_f32(const float* x, float* disp, int B, int Di, int Hi, int Wi, int Do, int Ho,
                                  int Wo, float alpha, const float* disp_sample_host, void* stream);

/* ------------------------------------------------------------------------------------------
 * AcfNet confidence head (cmn/cmn.py:10-36,57-69)
 *   h = relu(conv2d_3x3(cost, w1) * scale + shift)   w1: [Cm, D, 3, 3], pad 1, no bias, BN folded
 *   conf = sigmoid(sum_m h_m * w2[m])                w2: [Cm]
 * cost: [B, D, H, W] -> conf: [B, 1, H, W].  w1pack from dmb_conf_head_pack_weights_f32.
 * ---------------------------------------------------------------------------------------- */
long long dmb_conf_head_packed_floats(int Cm, int D);
int dmb_conf_head_pack_weights_f32(const float* w1, float* w1pack, int Cm, int D, void* stream);
int dmb_conf_head_f32(const float* cost, const float* w1pack, const float* scale, const float* shift,
                      const float* w2, float* conf, int B, int D, int Cm, int H, int W, void* stream);

/* The same head when `cost` is AcfNet's learned 4x up-sampling (ConvTranspose3d(1, 1, 8, 4, 2), aggregators/AcfNet.py:
 * 55-57,81-83) of a quarter-resolution volume c [B, Dq, Hq, Wq]: the head's 3x3 convolution composed with the up-sampling
 * is, per output phase (Y mod 4, X mod 4), a 3x3 convolution of c (Dq channels) -- 4x fewer multiplications, run on
 * dmb_conv2d_f32 with the composed weights (host side: ops.conf_head_k8s4_pack) into hq [B, 16*M, Hq, Wq], channel =
 * (phase_y*4 + phase_x)*M + m, BatchNorm + ReLU applied.
 * dmb_conf_gather_f32: conf[b, 0, 4y'+py, 4x'+px] = sigmoid(sum_m hq[b, (py*4+px)*M + m, y', x'] * w2[m]).
 * dmb_conf_ring_f32: the outermost pixel ring of conf recomputed directly from the up-sampled volume cost [B, D, H, W]
 *   (there the head zero-pads the volume, which the composed form cannot see); w1t = w1 transposed to [D, 3, 3, M], M = 64. */
int dmb_conf_gather_f32(const float* hq, const float* w2, float* conf, int B, int M, int Hq, int Wq, void* stream);

/* The same composed head WITHOUT the hidden tensor, in one launch: for each of the nsets weight sets laid out one after the
 * other in wpack (set p = the sub-pixel phases 2 p and 2 p + 1 x 64 hidden channels, weights of ops.conf_head_k8s4_pack), the
 * 3x3 convolution Ci -> 128 of the quarter-resolution volume c [B, Ci, Hq, Wq] with BN scale/shift and ReLU, each pixel's
 * 64-vector reduced against w2 [64] in the kernel's epilogue, sigmoid, scattered to conf[b, 0, 4 y + by, 4 x + bx]
 * (phase = 4 by + bx).  Replaces conv2d + dmb_conf_gather_f32 on the composed path (cmn/cmn.py:27-43 on
 * aggregators/AcfNet.py:55-57,81-83); nsets = 8 covers all 16 phases. */
int dmb_conf_phase_conv2d_f32(const float* c, const float* wpack, const float* scale, const float* shift, const float* w2,
                              float* conf, int B, int Ci, int Hq, int Wq, int nsets, void* stream);
int dmb_conf_ring_f32(const float* cost, const float* w1t, const float* scale, const float* shift, const float* w2,
                      float* conf, int B, int D, int M, int H, int W, void* stream);

/* ------------------------------------------------------------------------------------------
 * Evaluation accumulator (data/datasets/evaluation/stereo/pixel_error.py:6-73 + eval.py:12-31 crop +
 * tools/test.py:304-307 averaging).  For every image b: crop est/gt [Hp, Wp] to rows [Hp-H0, Hp) and
 * columns [0, W0); mask = gt > lb && gt < ub; if the mask is non-empty
 *     acc[0] += 1; acc[1] += mean|gt-est|; acc[2..5] += 100 * mean(|gt-est| > {1,2,3,5})
 * (an empty mask contributes an image with all-zero errors, pixel_error.py:48-55).  acc: 6 doubles on
 * the device, updated in stream order (no atomics); the caller zeroes it and all-reduces it across
 * ranks.  workspace: DMB_EPE_WORKSPACE_DOUBLES * B doubles of caller-owned device scratch, fully overwritten by the
 * call: every image is reduced in 64 slices whose sums are added in a fixed order, so the per-image means are
 * reproducible bit for bit (since ABI version 3; version 2 took 6*B doubles and summed the slices atomically). */
#define DMB_EPE_WORKSPACE_DOUBLES 384
int dmb_epe_accum_f64(const float* est, const float* gt, double* acc, double* workspace, int B, int Hp, int Wp,
                      int H0, int W0, float lb, float ub, void* stream);

/* The same for up to 4 estimates against ONE ground truth in a single pass (the disparity maps of one forward:
 * tools/test.py evaluates every entry of results['disps'] against the same batch['leftDisp']): est: HOST array of
 * nmaps device pointers; acc: [nmaps, 6] doubles, row i accumulated from est[i]; workspace: DMB_EPE_WORKSPACE_DOUBLES*B*nmaps doubles.
 * Per estimate the arithmetic is dmb_epe_accum_f64's. */
int dmb_epe_accum_multi_f64(int nmaps, const float* const* est, const float* gt, double* acc, double* workspace, int B,
                            int Hp, int Wp, int H0, int W0, float lb, float ub, void* stream);

/* ------------------------------------------------------------------------------------------
 * Data-side conventions in front of the path: dmb/data/transforms/stereo_trans.py:20-44 (CenterCrop), :92-119 (StereoPad),
 * :78-90 (Normalize), composed in the order of dmb/data/datasets/stereo/builder.py:22-28 and dmb/apis/inference.py:120-129
 * (pad BEFORE normalise: a padded pixel holds (0 - mean[c]) / std[c]).  One pass:
 *   dst[b, c, y, x] = n(src[b, c, y - (th - h) + y0, x + x0])   for y >= th - h and x < w,   n(0) elsewhere;
 *   n(v) = (v - mean[c]) / std[c]  (FP32 subtract, correctly rounded FP32 divide -- torchvision's sub_().div_()), or v when
 *   mean_host / std_host are NULL (pad / crop only).
 * src: planar FP32 [B, Cs, sh, sw] (_f32) or the decoder's interleaved bytes [B, sh, sw, Cs] (_u8; imread's layout, the first
 * C <= Cs channels are taken as stereo/scene_flow/base.py:17-23 does); (y0, x0, h, w) = the window of the source that is kept
 * (the whole image: 0, 0, sh, sw); dst [B, C, th, tw] with th >= h, tw >= w, tw % 4 == 0, C <= 4. */
int dmb_stereo_pad_normalize_f32(const float* src, float* dst, int B, int C, int Cs, int sh, int sw, int y0, int x0, int h, int w,
                                 int th, int tw, const float* mean_host, const float* std_host, void* stream);
int dmb_stereo_pad_normalize_u8(const unsigned char* src_hwc, float* dst, int B, int C, int Cs, int sh, int sw, int y0, int x0,
                                int h, int w, int th, int tw, const float* mean_host, const float* std_host, void* stream);

/* ------------------------------------------------------------------------------------------
 * "Next" rows: the 2-D networks either side of the path.
 * dmb/modeling/stereo/backbones/PSMNet.py:8-129 and backbones/StereoNet.py:7-106 with
 * layers/basic_layers.py:31-46,105-123,219-243; disp_refinement/StereoNet.py:7-62 with
 * disp_refinement/utils/edge_aware.py:8-70.
 * ---------------------------------------------------------------------------------------- */

/* nn.Conv2d weights [Co, Ci, k, k] (k = 1, 3 or 5) -> MFMA A-fragment stream; Co <= 128, any Ci. */
long long dmb_conv2d_packed_floats(int Co, int Ci, int ksize);
int dmb_conv2d_pack_weights_f32(const float* w, float* wpack, int Co, int Ci, int ksize, void* stream);

/* Conv2d with padding = dilation * (k/2) + folded BatchNorm / bias (scale, shift: each may be NULL).  Supported:
 * stride 1 with kernel 1 | 3, dilation 1 | 2 (4 | 8 when Co <= 32), Co <= 128; stride 2 with kernel 1 | 3 (Co <= 64) or
 * 5 (Co <= 32), dilation 1; anything else returns DMB_EUNSUPPORTED.  Then + residual (may be NULL; added after the affine,
 * basic_layers.py:236-241) + ReLU.  x, y and residual may be channel windows of wider tensors: x points at the first input channel of batch item 0 of
 * a tensor with in_channels_total channels per item (likewise y / out_channels_total, residual / res_channels_total).
 * x: [B, Ci (of in_channels_total), H, W] -> y: [B, Co (of out_channels_total), Ho, Wo], Ho = (H - 1) / stride + 1. */
int dmb_conv2d_f32(const float* x, const float* wpack, const float* scale, const float* shift, const float* residual,
                   float* y, int B, int Ci, int Co, int H, int W, int ksize, int stride, int dilation, int relu,
                   int in_channels_total, int out_channels_total, int res_channels_total, void* stream);

/* njobs (<= 6) independent 3x3 stride-1 convolutions of ONE layer shape (B, Ci -> Co, H) in one launch: job q convolves
 * x[q] [B, Ci, H, W[q]] with wpack[q] into y[q] (a channel window of a tensor with out_channels_total[q] channels); no affine,
 * no residual, no ReLU; W[q] % 4 == 0 and 16-byte aligned bases.  The arrays are HOST arrays of device pointers / sizes.
 * Used by the volume-free first layer (cat_fms.py:7-48 + aggregators/PSMNet.py:31-33 without the volume): its five 2-D
 * convolutions, three of them 52-column border maps that would each take a round of the chip on their own. */
int dmb_conv2d_k3_multi_f32(int njobs, const float* const* x, const float* const* wpack, float* const* y, const int* W,
                            const int* out_channels_total, int B, int Ci, int Co, int H, void* stream);

/* nn.AvgPool2d(k, stride=k) (PSMNet.py:43-58): channels [in_ch_offset, in_ch_offset + C) of x [B, in_channels_total,
 * H, W] -> y [B, C, H/k, W/k]. */
int dmb_avgpool2d_f32(const float* x, float* y, int B, int C, int H, int W, int k, int in_channels_total,
                      int in_ch_offset, void* stream);

/* F.interpolate(mode='bilinear', align_corners=True) (PSMNet.py:95-117): x [B, C, Hi, Wi] -> channels
 * [out_ch_offset, out_ch_offset + C) of y [B, out_channels_total, Ho, Wo]. */
int dmb_bilinear_ac_f32(const float* x, float* y, int B, int C, int Hi, int Wi, int Ho, int Wo, int out_channels_total,
                        int out_ch_offset, void* stream);

/* F.interpolate(mode='bilinear', align_corners=False) * mult: the coarse disparity map brought to image size and
 * rescaled by the resolution ratio (disp_refinement/StereoNet.py:49-50, disp_refinement/utils/edge_aware.py:49-50).
 * x [B, C, Hi, Wi] -> channels [out_ch_offset, out_ch_offset + C) of y [B, out_channels_total, Ho, Wo]. */
int dmb_bilinear_scale_f32(const float* x, float* y, int B, int C, int Hi, int Wi, int Ho, int Wo, float mult,
                           int out_channels_total, int out_ch_offset, void* stream);

/* ------------------------------------------------------------------------------------------
 * "Next" row (first part): the training-side loss terms of AcfNet's cost filtering, forward and
 * backward, each ONE pass over what it reads.
 * ---------------------------------------------------------------------------------------- */

/* Doubles of reduction workspace for a loss over n_elements pixels (per-block partial sums, deterministic). */
long long dmb_loss_workspace_doubles(long long n_elements);

/* StereoFocalLoss.loss_per_level with LaplaceDisp2Prob (losses/stereo_focal_loss.py:63-101, losses/utils/
 * disp2prob.py:107-173) at the cost volume's own resolution:
 *   m1 = lower < gt < upper;  g = gt*m1;  m2 = start_disp < g < end_disp;  p = softmax_d(-|s_d - g*m2| / variance);
 *   P = p*m2 + 1e-40;  loss = -sum(P * (1-P)^(-focal_coefficient) * log_softmax_d(cost) * m1) / max(sum(m1), 1).
 * cost [B, D, H, W]; gt [B, 1, H, W]; variance: per-pixel map [B, 1, H, W] or NULL (then variance_scalar);
 * disp_sample_host: D host floats (s_d).  stats: 2*B*H*W floats saved for the backward pass; loss_out: 2 floats
 * (the loss, the divisor). */
int dmb_stereo_focal_loss_fwd_f32(const float* cost, const float* gt, const float* variance, float variance_scalar,
                                  const float* disp_sample_host, float* stats, double* workspace, float* loss_out,
                                  int B, int D, int H, int W, float lower, float upper, float start_disp,
                                  float end_disp, float focal_coefficient, void* stream);
/* d loss / d cost [B, D, H, W] and (if grad_variance != NULL) d loss / d variance [B, 1, H, W], scaled by
 * grad_out[0] (device scalar, may be NULL = 1) * grad_scale. */
int dmb_stereo_focal_loss_bwd_f32(const float* cost, const float* gt, const float* variance, float variance_scalar,
                                  const float* disp_sample_host, const float* stats, const float* loss_out,
                                  const float* grad_out, float grad_scale, float* grad_cost, float* grad_variance,
                                  int B, int D, int H, int W, float lower, float upper, float start_disp,
                                  float end_disp, float focal_coefficient, void* stream);

/* Masked mean over a [B, 1, H, W] map, mask = lower < gt < upper.  mode 0: ConfidenceNllLoss, -logsigmoid(x)
 * (losses/conf_nll_loss.py:35-56, x = confidence logits); mode 1: DispSmoothL1Loss, smooth_l1(x - gt)
 * (losses/smooth_l1_loss.py:36-58, x = estimated disparity).  loss_out: 2 floats (loss, divisor). */
int dmb_map_loss_fwd_f32(const float* x, const float* gt, double* workspace, float* loss_out, long long n, float lower,
                         float upper, int mode, void* stream);
int dmb_map_loss_bwd_f32(const float* x, const float* gt, const float* loss_out, const float* grad_out,
                         float grad_scale, float* grad_x, long long n, float lower, float upper, int mode,
                         void* stream);

/* ------------------------------------------------------------------------------------------
 * "Next" row (second part): backward passes of the 3-D convolution units (autograd of the
 * nn.Sequential(Conv3d | ConvTranspose3d, BatchNorm3d[, ReLU]) factories, layers/basic_layers.py:68-100,160-177).
 * "dc" is the gradient with respect to the raw convolution output (before BatchNorm).
 *
 * Data gradients reuse the forward kernels (scale = shift = residual = NULL, relu = 0) on re-packed weights:
 *   stride-1 Conv3d   dx = dmb_conv3d_k3_f32(dc, pack_dgrad(w), stride 1)           (channels Co -> Ci)
 *   stride-2 Conv3d   dx = dmb_deconv3d_k3s2_f32(dc, dmb_deconv3d_pack_weights_f32(w, Co, Ci))   (even D, H, W of x)
 *   ConvTranspose3d   dx = dmb_conv3d_k3_f32(dy, dmb_conv3d_pack_weights_f32(w, Ci, Co), stride 2)
 * ---------------------------------------------------------------------------------------- */

/* Weights of the data-gradient convolution of a stride-1 nn.Conv3d with weight w [Co, Ci, 3, 3, 3]: channel roles
 * exchanged, taps mirrored.  wpack holds dmb_conv3d_packed_floats(Ci, Co) floats. */
int dmb_conv3d_pack_dgrad_weights_f32(const float* w, float* wpack, int Co, int Ci, void* stream);

/* Many packs in ONE launch (ABI 8).  `jobs_device`: njobs dmb_pack_job records IN DEVICE MEMORY (the caller keeps the table as
 * long as its parameters and pack buffers live; a training step re-packs every unit's forward and data-gradient weights after
 * each optimizer update -- 52 launches of 4.8 us in a PSMNet step).  mode 0: dmb_conv3d_pack_weights_f32(w, wpack, Co, Ci);
 * 1: dmb_deconv3d_pack_weights_f32(w, wpack, Ci, Co) (w = [Ci, Co, 27]); 2: dmb_conv3d_pack_dgrad_weights_f32 of a layer
 * with weight [Ci, Co, 27] (here Co = the DATA GRADIENT's output channels).  wpack: dmb_conv3d_packed_floats(Co, Ci) floats. */
typedef struct {
  const float* w;
  float* wpack;
  int Co, Ci, mode, reserved;
} dmb_pack_job;
int dmb_conv3d_pack_weights_multi_f32(const void* jobs_device, int njobs, void* stream);

/* Weight gradient of a stride-1 nn.Conv3d (kernel 3, padding 1): dw[co, ci, tap] = sum_{b, v} dc[b, co, v] *
 * x[b, ci, v + tap - 1] (torch.nn.grad.conv3d_weight).  x [B, Ci, D, H, W], dc [B, Co, D, H, W], dw [Co, Ci, 27].
 * FP32 MFMA chains per workgroup, partial results added in a fixed order (bit-reproducible, no atomics).
 * workspace: dmb_conv3d_wgrad_workspace_floats(Co, Ci) floats.  32 channels of one batch item must stay below 2 GiB. */
long long dmb_conv3d_wgrad_workspace_floats(int Co, int Ci);
int dmb_conv3d_k3_wgrad_f32(const float* x, const float* dc, float* dw, float* workspace, int B, int Ci, int Co, int D,
                            int H, int W, void* stream);
/* Weight gradient of the stride-2 units: out[cs, cb, tap] = sum_{b, o} small[b, cs, o] * big[b, cb, 2 o + tap - 1].
 *   nn.Conv3d(k 3, s 2, p 1):                 small = dc [B, Co, Ds, Hs, Ws], big = x  [B, Ci, Db, Hb, Wb] -> dW [Co, Ci, 27]
 *   nn.ConvTranspose3d(k 3, s 2, p 1, op 1):  small = x  [B, Ci, ...],        big = dy [B, Co, 2Ds, ...]  -> dW [Ci, Co, 27]
 * Each big extent is 2n or 2n - 1 (16-byte staging when both widths are multiples of 4 and the tensors aligned).
 * workspace: dmb_conv3d_wgrad_workspace_floats(Cs, Cb) floats. */
int dmb_conv3d_k3s2_wgrad_f32(const float* small, const float* big, float* dw, float* workspace, int B, int Cs, int Cb,
                              int Ds, int Hs, int Ws, int Db, int Hb, int Wb, void* stream);

/* Weight gradient of a stride-1 nn.Conv2d with kernel 1, or kernel 3 with dilation 1 | 2 | 4 | 8 (padding = dilation * (k / 2)):
 * AcfNet's confidence heads (cmn/cmn.py:21-36) and the stride-1 layers of the 2-D networks.  x [B, Ci, H, W],
 * dc [B, Co, H, W] -> dw [Co, Ci, k*k].  W a multiple of 4, tensors 16-byte aligned.  The data gradient is dmb_conv2d_f32 on
 * mirrored, channel-exchanged weights.  workspace: dmb_conv2d_wgrad_workspace_floats(Co, Ci) floats. */
long long dmb_conv2d_wgrad_workspace_floats(int Co, int Ci);
int dmb_conv2d_wgrad_f32(const float* x, const float* dc, float* dw, float* workspace, int B, int Ci, int Co, int H, int W,
                         int ksize, int dilation, void* stream);

/* BatchNorm (training mode) + skip add + ReLU of a convolution unit, layout [B, C, S] (S = voxels or pixels per channel).
 * relu: 0 none, 1 after the skip add, 2 before it -- the same epilogue the inference kernels fuse.
 *
 * dmb_bn_train_stats_f32: batch mean / biased variance per channel (FP64 sums, fixed order) ->
 *   mean_out, invstd_out = 1/sqrt(var + eps), scale_out = gamma*invstd, shift_out = beta - mean*scale_out (C floats each);
 *   running_mean / running_var (may be NULL) are updated as nn.BatchNorm does (momentum, unbiased variance).
 *   gamma / beta may be NULL (1 / 0).  workspace: dmb_bn_workspace_doubles(C, S) doubles.
 * dmb_bn_act_f32: y = act(c*scale + shift (+ residual)).
 * dmb_bn_train_fwd_f32 (ABI 8): both of the above for a batch-statistics unit in TWO launches instead of four -- the block
 *   sums, then one kernel whose workgroups finish their channel's statistics themselves (same arithmetic, same bits as
 *   dmb_bn_train_stats_f32), write mean / invstd / scale / shift, update the running buffers, add 1 to *num_batches_tracked (an
 *   int64 on the device, may be NULL: nn.BatchNorm's counter, basic_layers.py:68-83 under train()) and normalise.
 * dmb_bn_act_bwd_f32: dpre = dy * [ReLU mask];  dbeta = sum dpre;  dgamma = sum dpre * (c - mean)*invstd;
 *   dc = scale*(dpre - dbeta/N - xhat*dgamma/N) if training else scale*dpre;  dres (may be NULL) = the gradient flowing
 *   into the skip branch (dpre for relu 1, dy otherwise) + dres_acc (may be NULL; ABI 8: what the skip operand has already
 *   collected from its other consumers, so that autograd's own addition -- three tensor passes -- is one extra read here).
 *   y (the unit's output) is only read for relu == 1.  Two launches (block sums; apply, which finishes the sums itself). */
/* Batch statistics from the convolution's epilogue (ABI 8): dmb_conv3d_k3_bnstats_f32 is the RAW stride-1 convolution Ci -> 32
 * (no affine, skip or ReLU: what a training-mode unit computes before its BatchNorm, basic_layers.py:68-83 under train()) that also
 * writes, per workgroup and channel, the FP64 sum of its outputs and of their squares: stats[(ch * P + p) * 2 + {0, 1}], P =
 * dmb_conv3d_k3_bnstats_partials(...) (0: the shape is not covered -- call dmb_conv3d_k3_f32 and dmb_bn_train_fwd_f32).
 * dmb_bn_train_act_f32 then is dmb_bn_train_fwd_f32 without its pass over c for the block sums: it finishes the statistics from
 * those partials (mean = sum / N, var = sumsq / N - mean^2 in FP64), updates the buffers and the counter, and normalises. */
long long dmb_conv3d_k3_bnstats_partials(int B, int Ci, int Co, int D, int H, int W);
int dmb_conv3d_k3_bnstats_f32(const float* x, const float* wpack, float* y, double* stats, int B, int Ci, int Co, int D, int H, int W,
                              void* stream);
int dmb_bn_train_act_f32(const float* c, const double* partials, int nparts, const float* gamma, const float* beta,
                         float* running_mean, float* running_var, long long* num_batches_tracked, float momentum, float eps,
                         float* mean_out, float* invstd_out, float* scale_out, float* shift_out, const float* residual, float* y,
                         int B, int C, long long S, int relu, void* stream);
long long dmb_bn_workspace_doubles(int C, long long S);
int dmb_bn_train_stats_f32(const float* c, const float* gamma, const float* beta, float* running_mean,
                           float* running_var, float momentum, float eps, float* mean_out, float* invstd_out,
                           float* scale_out, float* shift_out, double* workspace, int B, int C, long long S, void* stream);
int dmb_bn_act_f32(const float* c, const float* scale, const float* shift, const float* residual, float* y, int B, int C,
                   long long S, int relu, void* stream);
int dmb_bn_train_fwd_f32(const float* c, const float* gamma, const float* beta, float* running_mean, float* running_var,
                         long long* num_batches_tracked, float momentum, float eps, float* mean_out, float* invstd_out,
                         float* scale_out, float* shift_out, const float* residual, float* y, double* workspace, int B, int C,
                         long long S, int relu, void* stream);
int dmb_bn_act_bwd_f32(const float* dy, const float* c, const float* y, const float* scale, const float* shift,
                       const float* mean, const float* invstd, double* workspace, float* dgamma, float* dbeta, float* dc,
                       float* dres, const float* dres_acc, int B, int C, long long S, int relu, int training, void* stream);

/* out[c] = sum_{b, s} a[b, c, s] * g[b, 0, s] (FP64 sums): the weight gradient of a 1x1 convolution with one output channel
 * (second layer of AcfNet's confidence heads, cmn/cmn.py:30).  workspace: dmb_bn_workspace_doubles(C, S) doubles. */
int dmb_channel_dot_f32(const float* a, const float* g, double* workspace, float* out, int B, int C, long long S, void* stream);

/* Weight gradient of the aggregators' first convolution (aggregators/PSMNet.py:31-33 on cat_fms.py:7-48, d_k = k) WITHOUT the volume
 * (ABI 8; the backward twin of dmb_catconv_*): one pass over dc [B, Co, D, H, W] folds z into 2 x 9 maps per output channel,
 *   maps_left [B, 9 Co, H, W] (channel (dz*3 + dx)*Co + co) = sum_z dc[co, z, y, x]               over 0 <= z+dz-1 < D, x+dx-1 >= z+dz-1
 *   maps_right[B, 9 Co, H, W]                               = sum_z dc[co, z, y, u + z + dz - dx]  over 0 <= z+dz-1 < D, u+z+dz-1 < W
 * and the layer's weight gradient is two 2-D weight gradients against them (dmb_conv2d_wgrad_f32 with x = the left / right feature
 * map): dW[co, ci, dz, dy, dx] = dWL[(dz*3+dx)*Co + co, ci, dy, dx], dW[co, C + ci, dz, dy, dx] = dWR[(dz*3+dx)*Co + co, ci, dy, 1]
 * (a difference volume, dif_fms.py:7-46: dW = the left term minus the right one). */
int dmb_cat_first_wgrad_maps_f32(const float* dc, float* maps_left, float* maps_right, int B, int Co, int D, int H, int W, void* stream);

/* Backward of the cost-volume builders: dvol [B, 2C (cat) or C (dif), D, H, W] -> dL, dR [B, C, H, W]; sums over the
 * valid columns of every disparity plane (cat_fms.py:36-44), FP32 in ascending plane order. */
int dmb_cat_fms_bwd_f32(const float* dvol, float* dL, float* dR, int B, int C, int H, int W, int D,
                        const int* disp_idx_host, void* stream);
int dmb_dif_fms_bwd_f32(const float* dvol, float* dL, float* dR, int B, int C, int H, int W, int D,
                        const int* disp_idx_host, void* stream);

/* Backward of the soft-argmin (normalize = True): grad_cost[k] = grad_disp * alpha * p_k * (s_k - disp),
 * p = softmax(alpha * cost).  cost, grad_cost [B, D, H, W]; disp (the forward result), grad_disp [B, 1, H, W]. */
int dmb_soft_argmin_bwd_f32(const float* cost, const float* disp, const float* grad_disp, float* grad_cost, int B,
                            int D, int H, int W, float alpha, const float* disp_sample_host, void* stream);

/* Backward of dmb_trilinear_ac_soft_argmin_f32 with respect to the low-resolution cost x [B, Di, Hi, Wi]: the gradient of
 * the disparity is propagated without the [B, Do, Ho, Wo] volume (re-created per pixel in registers); grad_y (may be NULL) is
 * a gradient that arrives on the volume itself (a loss on the costs) and is added.  scratch: B*Di*Ho*Wo floats. */
int dmb_trilinear_ac_soft_argmin_bwd_f32(const float* x, const float* disp, const float* grad_disp, const float* grad_y,
                                         float* scratch, float* grad_x, int B, int Di, int Hi, int Wi, int Do, int Ho, int Wo,
                                         float alpha, const float* disp_sample_host, void* stream);
/* Backward of dmb_trilinear_ac_f32 alone (F.interpolate(trilinear, align_corners=True)): grad_y [B, Do, Ho, Wo] ->
 * grad_x [B, Di, Hi, Wi].  scratch: B*Di*Ho*Wo floats. */
int dmb_trilinear_ac_bwd_f32(const float* grad_y, float* scratch, float* grad_x, int B, int Di, int Hi, int Wi, int Do, int Ho,
                             int Wo, void* stream);

/* Backward of dmb_avgpool2d_f32 and dmb_bilinear_ac_f32 on plain [B, C, ., .] tensors (the SPP branches of the PSMNet
 * backbone under training): grad_y -> grad_x. */
int dmb_avgpool2d_bwd_f32(const float* grad_y, float* grad_x, int B, int C, int H, int W, int k, void* stream);
int dmb_bilinear_ac_bwd_f32(const float* grad_y, float* grad_x, int B, int C, int Hi, int Wi, int Ho, int Wo, void* stream);
/* Backward of dmb_bilinear_scale_f32 (half-pixel bilinear * mult, edge_aware.py:49-50) on plain tensors. */
int dmb_bilinear_scale_bwd_f32(const float* grad_y, float* grad_x, int B, int C, int Hi, int Wi, int Ho, int Wo, float mult,
                               void* stream);

/* Backward of dmb_deconv3d_k8s4_c1_f32 (AcfNet's learned up-sampling, aggregators/AcfNet.py:55-57): dx [B, D, H, W] =
 * sum_k dy[4 i - 2 + k] w[k], dw [8, 8, 8] = sum_{b, i} x[b, i] dy[b, 4 i - 2 + k]; dy [B, 4D, 4H, 4W].  Either output may be
 * NULL.  workspace: dmb_deconv3d_k8s4_bwd_workspace_doubles() doubles (needed for dw). */
long long dmb_deconv3d_k8s4_bwd_workspace_doubles(void);
int dmb_deconv3d_k8s4_c1_bwd_f32(const float* x, const float* w, const float* dy, float* dx, float* dw,
                                 double* workspace, int B, int D, int H, int W, void* stream);

/* ------------------------------------------------------------------------------------------
 * EXPERIMENTAL, OPT-IN (never selected by the default path): the stride-1
 * convolution (32 or 64 output channels) with every FP32 operand split exactly into three bf16 pieces and the six largest cross products issued on
 * the bf16 matrix cores with FP32 accumulation.  At least as close to the real-number result as the FP32 fma chain of
 * dmb_conv3d_k3_f32 (tests compare both with FP64), but not bit-identical to it.  Same arguments as dmb_conv3d_k3_f32
 * with stride 1; Co = 32 with W a multiple of 48, or Co = 64 with W a multiple of 24; tensors 16-byte aligned.
 * ---------------------------------------------------------------------------------------- */
long long dmb_conv3d_x6_packed_bytes(int Co, int Ci);
int dmb_conv3d_x6_pack_weights_f32(const float* w, void* wpack, int Co, int Ci, void* stream);
int dmb_conv3d_k3_x6_f32(const float* x, const void* wpack, const float* scale, const float* shift,
                         const float* residual, float* y, int B, int Ci, int Co, int D, int H, int W, int relu,
                         void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DMB_HIP_H */
