/* libsuo_hip.so -- C ABI of the MI355X-native (gfx950) hot path of rpng/suo_slam.
 *
 * Drop-in boundary.  Today the reference reaches its native code through
 * two pybind11 modules imported at /root/reference/lib/object_slam.py:9-10:
 *     lambdatwist.pnp(xs, ys, threshold)     thirdparty/lambdatwist/pnp_python_binding.cpp:57-62
 *     g2o.* graph API used by optimize()     lib/object_slam.py:703-903 (python/types/object_slam/types_object_slam.h:17-52)
 * and runs its network as stock PyTorch ops (PkpNet.forward, lib/models/pkpnet.py:80-119).
 * This header is what a ctypes / pybind stub binds instead (INTEGRATION.md shows the stubs).
 *
 * Conventions: plain C, caller-owned buffers, no exceptions; every function returns 0 on success
 * or a SUO_ERR_* code (suo_last_error() gives the message).  Pointers named *_dev are device (HBM)
 * pointers; `stream` is a hipStream_t passed as void* (NULL = internal stream + blocking).
 * All work is stream-ordered; the library never reads or writes host memory behind *_dev names.
 *
 * Runtime switches.  The library reads NINE environment variables, all of them selections between forms whose results the test suite holds against each other
 * (csrc/tune.h: env_switch); nothing else in the environment changes what it does:
 *     SUO_WINO_BF16X3=0      networks are built on the fp32 matrix pipe (SUO_PIPE_F32)                      read when a network is created
 *     SUO_F16X2=0            networks are built on three bf16 terms per operand (SUO_PIPE_BF16X3)             read when a network is created
 *     SUO_STEM_X3=0          prior-less pass: RoIAlign and the stem as two launches instead of the fused one  first use, per process
 *     SUO_FUSE_UPSAMPLE=0    Hourglass up-sample add as its own launch instead of the fused tail's epilogue   first use, per process
 *     SUO_FUSE_POOL=0        2x2 max-pool as its own launch instead of the producing GEMM's epilogue          first use, per process
 *     SUO_NET_SIDE_STREAMS=n the Hourglass up1 branches on n side streams (default 0)                          first use, per process
 *     SUO_SERIAL             (set) one stream, kernels back to back: per-kernel profiling                      per call
 *     SUO_LM_CAM2=0          camera tracking on lm_cam_kernel instead of lm_cam2_kernel                        first use, per process
 *     SUO_WINO_W8=0          small launches keep the four-wave Winograd kernels (see suo_conv3x3_wino_f16x2*)  per launch
 * The thresholds and A/B knobs behind the measurements of DESIGN.md / profiles/REJECTED.md (SUO_TUNE in csrc/tune.h: launch-size thresholds, tile choices, kernel
 * selections) are compiled to their defaults; only the variant builds of tools/build_variant.sh (-DSUO_TUNING) read them from the environment.  The host side above
 * the ABI has its own, in suo_slam_amd/: SUO_HIP_LIB (path of the library), SUO_BA_GRAPH / SUO_BA_HOST_SCHEDULE / SUO_FORCE_COLLECTIVES (ba_dist.py),
 * SUO_SLAM_STORE_SLOTS (slam_score.py), SUO_SLAM_VOTE_CHAIN (object_slam.py: 0 = the host votes between the passes of a SLAM view).
 */
#ifndef SUO_HIP_H
#define SUO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SUO_NUM_KP 41     /* lib/labeling/kp_config.py:82-94 */
#define SUO_HEAT 64       /* heat-map side */
#define SUO_CROP 256      /* network input side (lib/datasets/bop.py:21) */

const char* suo_last_error(void);
int suo_version(void);
/* number of visible HIP devices; <0 on error.  The product path requires one. */
int suo_device_count(void);

/* ---- keypoint network: replaces PkpNet.forward (lib/models/pkpnet.py:80-119) -------------------- */
typedef struct suo_net suo_net;

/* Build from a reference-layout state_dict (checkpoint['model'], lib/object_slam.py:92-97):
 * n float32 host tensors, names[i] the reference key ("backbone.r1.conv1.weight", ...), shapes[i]
 * its dims.  BatchNorm is folded and weights are packed for MFMA on the host, then uploaded. */
int suo_net_create(int n, const char* const* names, const float* const* data, const int64_t* const* shapes,
                   const int* ndims, int max_crops, suo_net** out);
void suo_net_destroy(suo_net* net);
int suo_net_set_graph(suo_net* net, int enable);          /* hipGraph replay of the backbone (default on) */
/* Capture the backbone graph for L crops ahead of time (with_priors: the 48-channel staging layout of the SLAM prior pass, else
 * the image-only layout).  Graphs are otherwise captured on the first call that sees a crop count -- inside whatever that call is
 * timing; a harness feeding frames with 3..max_crops detections calls this once per count at start-up.  `stream` as in the forwards
 * the graphs will be replayed on (capture is stream-agnostic; NULL = internal). */
int suo_net_prepare(suo_net* net, int L, int with_priors, void* stream);
size_t suo_net_workspace_bytes(const suo_net* net);
/* Measurement aid (nothing is launched): the launch schedule of ONE call of L crops cut from n_frames frames of H x W pixels, walked as a dry run on the
 * network's current matrix pipe.  bytes[0..5] = ALGORITHMIC HBM bytes (every operand read once, every result written once, the weights once) of the
 * staging / stem launch, the 3x3 convolutions (+ fused Residual tails), the 1x1 GEMMs, the one-launch Residual blocks, the pool / up-sample launches,
 * decode + classifier; *n_launches = kernels of the call.  bench.py's `roofline_all.whole_call` divides their sum by the measured step time. */
int suo_net_schedule_bytes(suo_net* net, int L, int n_frames, int H, int W, int with_priors, double* bytes6, int* n_launches);

/* Matrix pipe of the network's large launches (the reference runs cuDNN's fp32 convolutions, lib/models/layers/Residual.py:20-35; all three forms are held
 * to its outputs at 1e-5, tests/test_gpu_cnn.py).  Chosen when the network is created -- SUO_WINO_BF16X3=0: SUO_PIPE_F32; SUO_F16X2=0: SUO_PIPE_BF16X3;
 * default SUO_PIPE_F16X2 -- and lowered at run time by suo_net_set_pipe (F16X2 -> BF16X3; the fp32 form only when created with it).
 *   SUO_PIPE_F32     v_mfma_f32_32x32x2_f32, exact fp32 products
 *   SUO_PIPE_BF16X3  operands as three bf16 terms, 6 of 9 cross products, fp32 accumulate (csrc/bf16x3.h): fp32's range
 *   SUO_PIPE_F16X2   operands as two fp16 terms, 3 of 4 cross products, fp32 accumulate (csrc/f16x2.h): half the matrix-pipe work, fp16's RANGE --
 *                    activations enter times 16, so a forward in which some |activation| >= 4094 is not computable in this form.  The kernels detect
 *                    that (they never write inf silently) and raise a flag:
 * suo_net_range_exceeded() returns 1 when a forward since its last call left the range -- the outputs of that forward are INVALID -- and clears the flag;
 * the network has then already been moved to SUO_PIPE_BF16X3 and the caller re-issues the call.  Call it after synchronising the stream and before
 * using the outputs.  Forwards on the NULL stream (blocking) check and re-issue by themselves. */
#define SUO_PIPE_F32 0
#define SUO_PIPE_BF16X3 1
#define SUO_PIPE_F16X2 2
int suo_net_get_pipe(const suo_net* net);
int suo_net_set_pipe(suo_net* net, int pipe);
int suo_net_range_exceeded(suo_net* net);

/* One frame: image either SUO_IMG_U8_HWC = uint8 [H,W,3] as cv2.imread gives it (scaled by 1/255 on
 * device: object_slam.py:1092 fused) or SUO_IMG_F32_CHW = float32 [3,H,W] already scaled (the tensor
 * PkpNet.forward receives); boxes float32 [L,4] xyxy pixels, priors float32 [L,41,256,256] or NULL (= zeros, pkpnet.py:95-97).
 * Outputs (device): uv [L,41,2], cov [L,41,2,2], kp_mask [L,41] (sigmoid prob), optional
 * kp_mask_logits [L,41] and prob_logits [L,41,64,64] (NCHW, the reference's ret["prob_logits"]). */
#define SUO_IMG_U8_HWC 0
#define SUO_IMG_F32_CHW 1
int suo_net_forward(suo_net* net, const void* img_dev, int img_format, int H, int W, const float* boxes_dev, int L,
                    const float* priors_dev, float* uv_dev, float* cov_dev, float* kp_mask_dev,
                    float* kp_mask_logits_dev, float* prob_logits_dev, void* stream);

/* Several independent frames in one call (they batch like crops do): imgs_dev is a contiguous stack of frames
 * [B,H,W,3] uint8 (or [B,3,H,W] float32), box_img_dev[l] the frame index of crop l; everything else as above. */
int suo_net_forward_frames(suo_net* net, const void* imgs_dev, int img_format, int H, int W, const float* boxes_dev,
                           const int* box_img_dev, int L, const float* priors_dev, float* uv_dev, float* cov_dev,
                           float* kp_mask_dev, float* kp_mask_logits_dev, float* prob_logits_dev, void* stream);

/* The SLAM pass with priors (lib/object_slam.py:486-519 -> __run_kp_model): instead of the dense prior tensor that the
 * reference renders on the host with make_prior_kp_input (lib/utils/utils.py:356-411) and uploads -- 10.7 MB per crop --
 * pass the projected keypoints themselves: prior_uv_dev float32 [L,41,2] NDC, prior_mask_dev uint8 [L,41] (0 = channel
 * stays zero; non-finite keypoints are skipped like utils.py:402).  The 91x91 Gaussian stamps are evaluated while the
 * crop is staged.  box_img_dev may be NULL (all crops from frame 0).  Everything else as suo_net_forward_frames. */
int suo_net_forward_prior_kp(suo_net* net, const void* imgs_dev, int img_format, int H, int W, const float* boxes_dev,
                             const int* box_img_dev, int L, const float* prior_uv_dev, const uint8_t* prior_mask_dev,
                             float* uv_dev, float* cov_dev, float* kp_mask_dev, float* kp_mask_logits_dev,
                             float* prob_logits_dev, void* stream);
/* make_prior_kp_input on the device: the dense [L,41,256,256] tensor the reference builds (parity / interoperability) */
int suo_render_priors(const float* prior_uv_dev, const uint8_t* prior_mask_dev, int L, float* out_dev, void* stream);

/* Backbone only (HourglassNet.forward, lib/models/hg.py:95-119): staged NHWC input [L,256,256,48]
 * (44 channels zero padded to 48; NULL = re-use the input staged by the previous call) -> logits
 * [L,41,64,64] NCHW (may be NULL).  Test / profiling entry for the conv stack. */
int suo_net_backbone(suo_net* net, const float* staged_dev, int L, float* prob_logits_dev, void* stream);

/* Frame upload (the H2D of lib/object_slam.py:1096-1098) as a stream-ordered kernel: src is PINNED host memory (hipHostMalloc /
 * torch pin_memory, i.e. mapped into the device's address space; 16-byte aligned like dst_dev), read over the host link with
 * coalesced 16-byte loads.  Use instead of hipMemcpyAsync in front of a network call: see csrc/misc.hip for the measured reason. */
int suo_upload(void* dst_dev, const void* src_pinned_host, size_t bytes, void* stream);

/* ---- stand-alone stages (each is also a parity-test entry point) --------------------------------- */
/* spatial_softmax + post_process_kp (pkpnet.py:13-63): logits [L,41,64,64] -> uv, cov, mean logit.
 * Optional outputs (NULL = skipped): argmax_idx_dev int32 [L,41] = flat index h*64+w of the first maximum of each heat-map, exactly what
 * torch.argmax(logits.flatten(2), -1) returns -- the reference decodes with a SOFT arg-max only (pkpnet.py:28-63), this diagnostic hard
 * arg-max is the bit-exact integer keypoint output; prob_dev float32 [L,41,64,64] = the soft-max itself, ret["prob"]
 * (pkpnet.py:111). */
int suo_decode_heatmaps(const float* logits_dev, int L, float* uv_dev, float* cov_dev, float* mean_logit_dev, int32_t* argmax_idx_dev,
                        float* prob_dev, void* stream);
/* classifier (pkpnet.py:74-78,116-118): sigmoid(W relu(mean_logit) + b) */
int suo_classifier(const float* mean_logit_dev, const float* w_dev, const float* b_dev, int L,
                   float* kp_logit_dev, float* kp_prob_dev, void* stream);
/* keypoint validity masks (object_slam.py:1100-1115); model_mask may be NULL (= all true) */
int suo_keypoint_masks(const float* uv_dev, const float* cov_dev, const float* kp_prob_dev, const uint8_t* model_mask_dev,
                       int L, float bbox_thresh, float kp_var_thresh, uint8_t* mask_dev, void* stream);
/* roi_align + concat (pkpnet.py:93-101) -> NHWC [L,256,256,48] (44 channels, zero padded to 48) */
int suo_roi_align_concat(const void* img_dev, int img_format, int H, int W, const float* boxes_dev, int L, const float* priors_dev,
                         float* out_dev, void* stream);

/* Weight packers (host): GEMM weight W[N][K] -> the two MFMA B-operand layouts back to back, `out` holds
 * 2*Np*Kp floats: [Kp/8][Np/32][64][4] for v_mfma_f32_32x32x2 and [Kp/16][Np/16][64][4] for v_mfma_f32_16x16x4
 * (small feature maps); conv weight W[N][C][KS][KS] -> same with K' = [chunk][ky][kx][kk]. */
int suo_pack_gemm_weight(const float* w, int N, int K, int Np, int Kp, float* out);
int suo_pack_conv_weight(const float* w, int N, int C, int KS, int Np, int Cp, int CK, float* out);

/* 1x1 convolution on NHWC pixels: out = epi(pro(A1) W1 + A2 W2 + bias (+R)); see csrc/conv.hip */
int suo_conv1x1(const float* a1_dev, int lda1, int K1, const float* pro_scale_dev, const float* pro_shift_dev,
                const float* a2_dev, int lda2, int K2, const float* wp_dev, const float* bias_dev,
                const float* r_dev, int ldr, float* out_dev, int ldo, int M, int N, int n_valid, int relu,
                int nchw_hw, void* stream);
/* ... followed by nn.MaxPool2d(2, 2) in the same launch (the stem's `pool(r1(x))`, pkpnet/hg.py; Hourglass `low1(max_pool(x))`, hg.py:41):
 * the M pixels are images of H x W (W a multiple of 64, H even, N a multiple of 128); pool_out_dev is [M/4, ldo]; out_dev may be NULL
 * when only the pooled tensor is wanted.  Bit-identical to suo_conv1x1 followed by suo_maxpool2. */
int suo_conv1x1_pool(const float* a1_dev, int lda1, int K1, const float* pro_scale_dev, const float* pro_shift_dev,
                     const float* a2_dev, int lda2, int K2, const float* wp_dev, const float* bias_dev,
                     const float* r_dev, int ldr, float* out_dev, int ldo, int M, int N, int relu, int H, int W,
                     float* pool_out_dev, void* stream);
/* The same 1x1 convolution (N = 128 or 64, K a multiple of 64 up to 512, optional BN + ReLU prologue, bias, optional ReLU) at fp32 accuracy on the
 * bf16 matrix pipe: both operands are split into three bf16 terms and 6 of the 9 cross products are accumulated in fp32
 * (csrc/gemm_bf16x3.hip; what suo_net_forward launches for conv1 of its Residual blocks at >= 4096 pixels on the fp16 pipe, >= 32768 on the bf16x3 pipe (csrc/net.hip: x3_min_rows) unless SUO_WINO_BF16X3=0).
 * wp3 = suo_pack_gemm_weight_bf16x3(W[N][K]) -> 3*N*K uint16 (MFMA B-operand order). */
int suo_pack_gemm_weight_bf16x3(const float* w, int N, int K, uint16_t* out);
int suo_conv1x1_bf16x3(const float* a_dev, int lda, int K, const float* pro_scale_dev, const float* pro_shift_dev, const uint16_t* wp3_dev,
                       const float* bias_dev, float* out_dev, int ldo, int M, int N, int relu, void* stream);
/* The general form: N a multiple of 128 (or 64), an optional second K segment (a2, K2: the skip conv of a Residual block; then no prologue), an optional
 * residual operand r_dev [M, ldr] added after the bias; K1, K2 multiples of 64; wp3 = suo_pack_gemm_weight_bf16x3 of the row-wise concatenation
 * [W1 | W2] (N rows, K1 + K2 columns).  What suo_net_forward launches for its N = 256 1x1 convolutions at 64x64 (lin, re-injection, conv3 + conv4). */
int suo_conv1x1_bf16x3_ex(const float* a1_dev, int lda1, int K1, const float* pro_scale_dev, const float* pro_shift_dev, const float* a2_dev,
                          int lda2, int K2, const uint16_t* wp3_dev, const float* bias_dev, const float* r_dev, int ldr, float* out_dev, int ldo,
                          int M, int N, int relu, void* stream);
/* ... with the 2x2 max-pool of the result written as well (suo_conv1x1_pool's contract: the M pixels are images of H x W, W a multiple of 64, H even;
 * pool_out_dev is [M/4, ldo]; out_dev may be NULL when only the pooled tensor is wanted). */
int suo_conv1x1_bf16x3_pool(const float* a1_dev, int lda1, int K1, const float* pro_scale_dev, const float* pro_shift_dev, const float* a2_dev,
                            int lda2, int K2, const uint16_t* wp3_dev, const float* bias_dev, const float* r_dev, int ldr, float* out_dev, int ldo,
                            int M, int N, int relu, int H, int W, float* pool_out_dev, void* stream);
/* The same kernel on TWO fp16 terms per operand, three MFMAs per product block (csrc/f16x2.h; what suo_net_forward launches by default).
 * w16 = suo_pack_gemm_weight_f16x2(W[N][K]) -> 2*N*K uint16 (row n times 2^t_n, max |row| in [2^12, 2^13)) and oscale[N] = 2^-(t_n + 4), the factor the
 * kernel's epilogue applies (activations enter times 2^4).  range_flag_dev: a device-visible word the kernel sets to 1 when a scaled activation reaches
 * 65504 (the launch's results are then invalid: re-issue on the bf16x3 entry); never cleared by the library. */
int suo_pack_gemm_weight_f16x2(const float* w, int N, int K, uint16_t* out, float* oscale_out);
int suo_conv1x1_f16x2_ex(const float* a1_dev, int lda1, int K1, const float* pro_scale_dev, const float* pro_shift_dev, const float* a2_dev, int lda2, int K2,
                         const uint16_t* w16_dev, const float* oscale_dev, const float* bias_dev, const float* r_dev, int ldr, float* out_dev, int ldo, int M, int N,
                         int relu, unsigned* range_flag_dev, void* stream);
int suo_conv1x1_f16x2_pool(const float* a1_dev, int lda1, int K1, const float* pro_scale_dev, const float* pro_shift_dev, const float* a2_dev, int lda2, int K2,
                           const uint16_t* w16_dev, const float* oscale_dev, const float* bias_dev, const float* r_dev, int ldr, float* out_dev, int ldo, int M, int N,
                           int relu, int H, int W, float* pool_out_dev, unsigned* range_flag_dev, void* stream);
/* KxK convolution, NHWC: KS=3 (stride 1, pad 1) or KS=7 (stride 2, pad 3) */
int suo_conv_kxk(int KS, const float* in_dev, int L, int H, int W, int C, const float* wp_dev, const float* bias_dev,
                 float* out_dev, int N, int relu, void* stream);
/* 3x3 convolution (stride 1, pad 1, N = 128 or 64 output channels, C a multiple of 16) in Winograd F(2x2,3x3) form: weight W[N][C][3][3] ->
 * out[16 * Np * Cp] floats (U = G g G^T per channel pair, computed in fp64, MFMA B-operand order); same tensors as suo_conv_kxk. */
int suo_pack_wino_weight(const float* w, int N, int C, int Np, int Cp, float* out);
int suo_conv3x3_wino(const float* in_dev, int L, int H, int W, int C, const float* wp_dev, const float* bias_dev, float* out_dev, int N,
                     int relu, void* stream);
/* The tail of a Residual block in ONE launch (layers/Residual.py:27-35): out = W3 relu(conv3x3(in) + bias2) + bias3 + skip with
 * in [L,H,W,128], out / skip [L,H,W,256] NHWC; the 128-channel tensor between the two convolutions stays in the CU.  Large
 * launches only (>= 1024 tiles of 128 pixels, the shapes the network uses it for); bit-identical to suo_conv_kxk + suo_conv1x1. */
int suo_conv3x3_conv1x1_skip(const float* in_dev, int L, int H, int W, const float* wp2_dev, const float* bias2_dev, const float* wp3_dev,
                             const float* bias3_dev, const float* skip_dev, float* out_dev, void* stream);
/* csrc/conv_wino_x3.hip (what suo_net_forward launches for the Residual blocks' 3x3 + tail unless SUO_WINO_BF16X3=0): the Winograd 3x3
 * convolution (128 -> 128 channels) with its element-wise products on the bf16 matrix pipe at fp32 accuracy (3-way split of both operands,
 * 6 of 9 cross terms, fp32 accumulate).
 * wq3 = suo_pack_wino_weight_bf16x3(W[128][128][3][3]) -> 3 * 16 * N * C uint16; same tensors as suo_conv3x3_wino / .._conv1x1_skip_up. */
int suo_pack_wino_weight_bf16x3(const float* w, int N, int C, uint16_t* out);
int suo_conv3x3_wino_x3(const float* in_dev, int L, int H, int W, const uint16_t* wq3_dev, const float* bias_dev, float* out_dev, int relu,
                        void* stream);
/* the same with channels = 128 or 64 (64 -> 64: the first two Residual blocks; wq3 = suo_pack_wino_weight_bf16x3(w, 64, 64, ...)) */
int suo_conv3x3_wino_x3_n(const float* in_dev, int L, int H, int W, int channels, const uint16_t* wq3_dev, const float* bias_dev, float* out_dev,
                          int relu, void* stream);
/* wp3_dev: conv3 weight packed by suo_pack_gemm_weight (tail_bf16x3 = 0: conv3 on the fp32 pipe) or by suo_pack_tail_weight_bf16x3
 * (tail_bf16x3 = 1: W3[256][128] -> 3 * 256 * 128 uint16, conv3 on the bf16 pipe as well). */
int suo_pack_tail_weight_bf16x3(const float* w3, int N2, int K, uint16_t* out);
int suo_conv3x3_wino_x3_conv1x1_skip_up(const float* in_dev, int L, int H, int W, const uint16_t* wq3_dev, const float* bias2_dev,
                                        const void* wp3_dev, int tail_bf16x3, const float* bias3_dev, const float* skip_dev,
                                        const float* up_dev, float* out_dev, void* stream);
/* csrc/conv_wino_x3.hip on two fp16 terms per operand (csrc/f16x2.h; the network's default for the Residual blocks' 3x3 + tail): packers return the planes
 * (2 * 16 * N * C / 2 * N2 * K uint16) and the per-output-channel factors; range_flag_dev as suo_conv1x1_f16x2_ex.
 * Launches of at most 256 tiles of 8 x 16 pixels (one workgroup per CU: a one-frame call, the passes of a SLAM view) run the 128-channel kernels with EIGHT waves
 * per tile (two per 32-channel slice, half the Winograd components each, no per-chunk fold; conv3 one 32-channel tile per wave): the same products, the last
 * additions of the output transform paired differently -- results within fp32 rounding of the four-wave form's, which SUO_WINO_W8=0 keeps everywhere. */
int suo_pack_wino_weight_f16x2(const float* w, int N, int C, uint16_t* out, float* oscale_out);
int suo_conv3x3_wino_f16x2_n(const float* in_dev, int L, int H, int W, int channels, const uint16_t* wq16_dev, const float* oscale_dev, const float* bias_dev,
                             float* out_dev, int relu, unsigned* range_flag_dev, void* stream);
int suo_pack_tail_weight_f16x2(const float* w3, int N2, int K, uint16_t* out, float* oscale_out);
int suo_conv3x3_wino_f16x2_conv1x1_skip_up(const float* in_dev, int L, int H, int W, const uint16_t* wq16_dev, const float* oscale2_dev, const float* bias2_dev,
                                           const uint16_t* w3p16_dev, const float* oscale3_dev, const float* bias3_dev, const float* skip_dev, const float* up_dev,
                                           float* out_dev, unsigned* range_flag_dev, void* stream);
/* ... and with the NEXT Residual block's conv1 in the same launch (layers/Residual.py:22-24 of the block that consumes out_dev): next_out_dev [L,H,W,128] =
 * relu(W1' relu(out * next_scale + next_shift) + next_b1) with W1' [128][256] (bn1 folded) as suo_pack_gemm_weight_f16x2 planes + factors.  out_dev is written as
 * before; the 256-channel tensor is not re-read by a separate 1x1 launch.  Bit-identical to suo_conv3x3_wino_f16x2_conv1x1_skip_up followed by
 * suo_conv1x1_f16x2_ex (same products, same order).  What suo_net_forward launches wherever a 256 -> 256 block on a launch of more than 256 tiles feeds another
 * (always the four-wave kernel: the eight-wave form of small launches does not carry the next block's conv1). */
int suo_conv3x3_wino_f16x2_conv1x1_skip_up_next(const float* in_dev, int L, int H, int W, const uint16_t* wq16_dev, const float* oscale2_dev, const float* bias2_dev,
                                                const uint16_t* w3p16_dev, const float* oscale3_dev, const float* bias3_dev, const float* skip_dev, const float* up_dev,
                                                float* out_dev, const float* next_scale_dev, const float* next_shift_dev, const uint16_t* next_w1h_dev,
                                                const float* next_osc1_dev, const float* next_b1_dev, float* next_out_dev, unsigned* range_flag_dev, void* stream);
/* The same with the 3x3 convolution in Winograd form (wq2 from suo_pack_wino_weight): what the network launches for its 256 -> 256 blocks */
int suo_conv3x3_wino_conv1x1_skip(const float* in_dev, int L, int H, int W, const float* wq2_dev, const float* bias2_dev, const float* wp3_dev,
                                  const float* bias3_dev, const float* skip_dev, float* out_dev, void* stream);
/* ... and with the Hourglass's "up1 + up2(low3)" (hg.py:56-58) folded in: out += nearest-neighbour 2x up-sampling of up_dev [L,H/2,W/2,256] */
int suo_conv3x3_wino_conv1x1_skip_up(const float* in_dev, int L, int H, int W, const float* wq2_dev, const float* bias2_dev, const float* wp3_dev,
                                     const float* bias3_dev, const float* skip_dev, const float* up_dev, float* out_dev, void* stream);
/* csrc/res_small.hip (what suo_net_forward launches for the Hourglass blocks at 32x32 and below when a call holds few crops -- the reference's
 * call shape, one frame per call, lib/object_slam.py:1099): a WHOLE 256 -> 256 Residual block (layers/Residual.py:20-35) in one launch,
 *   out = W3 relu(conv3x3(relu(W1 relu(x * pro_scale + pro_shift) + b1)) + b2) + b3 + x [+ 2x up-sampling of up_dev]
 * x_dev [L,H,W,256] NHWC -- or, pool_in = 1, [L,2H,2W,256] whose 2x2 max-pool is the block's input (hg.py:41).  Weights with their
 * BatchNorms folded by the caller: w1 [128][256], w2 [128][128][3][3] (times scale2[n] if given), w3 [256][128], packed by
 * suo_pack_res_block into w1p [128*256], w2p [128*128*9], w3p [256*128] floats.  Bit-identical to suo_conv1x1 (prologue, ReLU) ->
 * suo_conv_kxk(3) (ReLU) -> suo_conv1x1 (+ skip) -> suo_upsample2_add on the same tensors. */
int suo_pack_res_block(const float* w1, const float* w2, const float* scale2, const float* w3, float* w1p, float* w2p, float* w3p);
int suo_res_block(const float* x_dev, int L, int H, int W, int pool_in, const float* pro_scale_dev, const float* pro_shift_dev, const float* w1p_dev,
                  const float* b1_dev, const float* w2p_dev, const float* b2_dev, const float* w3p_dev, const float* b3_dev, const float* up_dev,
                  float* out_dev, void* stream);
/* csrc/res_small_x3.hip: the same block with every product on the bf16 matrix pipe at fp32 accuracy (3-way operand split, csrc/bf16x3.h), 4 x 8 pixel
 * tiles -- the network's default for the 32x32 and 16x16 levels of a one-frame call.  Weights as uint16 planes: w1x [3*128*256], w2x [3*128*128*9],
 * w3x [3*256*128] from suo_pack_res_block_bf16x3 (same inputs as suo_pack_res_block). */
int suo_pack_res_block_bf16x3(const float* w1, const float* w2, const float* scale2, const float* w3, uint16_t* w1x, uint16_t* w2x, uint16_t* w3x);
int suo_res_block_bf16x3(const float* x_dev, int L, int H, int W, int pool_in, const float* pro_scale_dev, const float* pro_shift_dev,
                         const uint16_t* w1x_dev, const float* b1_dev, const uint16_t* w2x_dev, const float* b2_dev, const uint16_t* w3x_dev,
                         const float* b3_dev, const float* up_dev, float* out_dev, void* stream);
/* ... and on two fp16 terms per operand (csrc/f16x2.h; the default of suo_net_forward for these levels): planes w1h [2*128*256], w2h [2*128*128*9], w3h [2*256*128]
 * and the per-channel factors osc1 [128], osc2 [128], osc3 [256] from suo_pack_res_block_f16x2; range_flag_dev as suo_conv1x1_f16x2_ex (here it also covers the
 * two activations the caller never sees, relu(conv1) and relu(conv2)). */
int suo_pack_res_block_f16x2(const float* w1, const float* w2, const float* scale2, const float* w3, uint16_t* w1h, uint16_t* w2h, uint16_t* w3h, float* osc1, float* osc2,
                             float* osc3);
int suo_res_block_f16x2(const float* x_dev, int L, int H, int W, int pool_in, const float* pro_scale_dev, const float* pro_shift_dev, const uint16_t* w1h_dev,
                        const float* osc1_dev, const float* b1_dev, const uint16_t* w2h_dev, const float* osc2_dev, const float* b2_dev, const uint16_t* w3h_dev,
                        const float* osc3_dev, const float* b3_dev, const float* up_dev, float* out_dev, unsigned* range_flag_dev, void* stream);
/* csrc/gemm_bf16x3.hip (gemm_chain_head_kernel): TWO 1x1 convolutions in one launch -- the last stack's  logits = tmpOut(relu(bn(lin(x))))  (hg.py:106-111) -- with the
 * 256-channel tensor between them kept in LDS instead of written and read back (what suo_net_forward launches for it at >= 4096 pixels on the fp16 pipe (csrc/net.hip: x3_min_rows); SUO_CHAIN_HEAD=0: two launches).
 * a_dev [M, lda >= 256] rows; w1h / osc1 = suo_pack_gemm_weight_f16x2 of W1 [256][256] (BatchNorm folded), bias1 [256]; w2h / osc2 of W2 [64][256] (rows >= n_valid zero),
 * bias2 [64]; out_nchw_dev [M / hw, n_valid, hw].  M a multiple of 64, hw a multiple of 64 dividing M.  Bit-identical to suo_conv1x1_f16x2_ex twice. */
int suo_conv1x1_chain_head_f16x2(const float* a_dev, int lda, int M, const uint16_t* w1h_dev, const float* osc1_dev, const float* bias1_dev, const uint16_t* w2h_dev,
                                 const float* osc2_dev, const float* bias2_dev, float* out_nchw_dev, int n_valid, int hw, unsigned* range_flag_dev, void* stream);
/* csrc/stem_x3.hip (what suo_net_forward launches for the prior-less pass unless SUO_STEM_X3=0): RoIAlign of the frame (pkpnet.py:93) + the stem
 * conv1_ 7x7 / stride 2 over the 3 image channels + bn1 + ReLU (hg.py:67-69,96-98) in one launch, products on the bf16 matrix pipe (3-way split);
 * the staged [L,256,256,*] crop tensor is never written.  wx = suo_pack_stem_weight_bf16x3(W[64][Cw][7][7], Cw, bn scale[64] or NULL) ->
 * 14 * 2 * 3 * 64 * 8 uint16; bias [64] (folded); frame / boxes / box_img as suo_roi_align_concat; out_dev [L,128,128,64] NHWC. */
int suo_pack_stem_weight_bf16x3(const float* w, int Cw, const float* scale, uint16_t* out);
int suo_stem_x3(const void* img_dev, int fmt, int H, int W, const float* boxes_dev, const int* box_img_dev, int L, const uint16_t* wx_dev,
                const float* bias_dev, float* out_dev, void* stream);
/* ... on two fp16 terms per operand (csrc/f16x2.h; the default of suo_net_forward): wh = suo_pack_stem_weight_f16x2 -> 14 * 2 * 2 * 64 * 8 uint16 and oscale [64];
 * range_flag_dev as suo_conv1x1_f16x2_ex (a uint8 frame cannot raise it; a float frame holding values beyond 4094 does). */
int suo_pack_stem_weight_f16x2(const float* w, int Cw, const float* scale, uint16_t* out, float* oscale_out);
int suo_stem_f16x2(const void* img_dev, int fmt, int H, int W, const float* boxes_dev, const int* box_img_dev, int L, const uint16_t* wh_dev, const float* oscale_dev,
                   const float* bias_dev, float* out_dev, unsigned* range_flag_dev, void* stream);
/* ... with the first Residual block's conv1 computed on the tile as well (what suo_net_forward launches; SUO_STEM_NEXT=0: conv1 as its own launch):
 * n_out [L,128,128,64] = relu(W1 relu(n_scale * out + n_shift) * n_osc1 + n_b1), n_w1h / n_osc1 = suo_pack_gemm_weight_f16x2 of W1 [64][64] (BatchNorm folded).
 * Bit-identical to suo_stem_f16x2 followed by suo_conv1x1_f16x2_ex on its output. */
int suo_stem_f16x2_next(const void* img_dev, int fmt, int H, int W, const float* boxes_dev, const int* box_img_dev, int L, const uint16_t* wh_dev, const float* oscale_dev,
                        const float* bias_dev, float* out_dev, const float* n_scale_dev, const float* n_shift_dev, const uint16_t* n_w1h_dev, const float* n_osc1_dev,
                        const float* n_b1_dev, float* n_out_dev, unsigned* range_flag_dev, void* stream);
int suo_maxpool2(const float* in_dev, float* out_dev, int L, int H, int W, int C, void* stream);
int suo_upsample2_add(const float* up1_dev, const float* low_dev, float* out_dev, int L, int H, int W, int C, void* stream);

/* ---- PnP: replaces lambdatwist.pnp (thirdparty/lambdatwist/pnp_python_binding.cpp:32-62) ---------- */
/* One call per object in the reference (lib/object_slam.py:1144); here all objects of a frame go in one
 * launch (one wavefront each).  Host buffers: n_pts[n_obj]; xs [sum n_pts][3] model points; ys
 * [sum n_pts][2] NORMALISED image points (object_slam.py:34-36); threshold as PnpParams (1e-3).
 * T_out [n_obj][16] row-major 4x4.  status[o] = 1 <=> identity pose = total failure, the reference's
 * error convention (pnp_ransac.cpp:231, object_slam.py:38).  Objects with n_pts < 4 return identity.
 * seed keys the counter-based sampler (object o uses seed + o * 0x9E3779B97F4A7C15).  Never throws. */
int suo_pnp_batch(int n_obj, const int* n_pts, const double* xs, const double* ys, double threshold, uint64_t seed,
                  int do_refine, double* T_out, int* status, int* best_inliers, int* iterations);
/* Index-work parity with the reference's sampler (legacy entry, not the batched fast path): the same RANSAC with hypothesis i of object o sampling the four
 * points draws[(o * n_draws + i) * 4 .. + 3] (ascending indices into the object's points) instead of the counter-based generator.  Fed with the sequence the
 * reference's process-global std::default_random_engine (seed 0) gives through get4RandomInRange0 (thirdparty/lambdatwist/pnp_ransac.cpp:161-183,
 * utils/random.h:65-116; restated host-side in suo_slam_amd/lambdatwist.py: ReferenceSampler) the chosen sample and the consensus set are the reference's.
 * n_draws >= 1000 (the iteration cap, parameters.h:76-102); winner[o] = index of the hypothesis that became the result, -1 if none; host buffers. */
int suo_pnp_replay(int n_obj, const int* n_pts, const double* xs, const double* ys, double threshold, const int* draws, int n_draws, int do_refine,
                   double* T_out, int* status, int* best_inliers, int* iterations, int* winner);
/* exact legacy signature: pnp(xs[N,3], ys[N,2], threshold) -> 4x4 */
int suo_pnp(const double* xs, const double* ys, int n, double threshold, double* T_out);

/* ---- device-resident frame geometry: network outputs -> poses in one stream-ordered chain ----------------------
 * What the reference does on the host between the network and the poses of a single-view frame -- the three .cpu() synchronisations
 * and mask logic (lib/object_slam.py:1100-1115), `exp_uv[k][kp_mask]` compaction (:1118-1135), the K^-T normalisation of pnp() (:34-36),
 * one lambdatwist.pnp per object (:1144), acceptance (:1147-1148), then optimize(): one g2o edge per keypoint with inv(cov) as
 * information (:795-837) and the robust LM rounds (:842-896) with the camera fixed at identity (:383-385, :774) -- as four kernels on the
 * caller's stream behind suo_net_forward* + suo_keypoint_masks, and ONE device-to-host copy of everything the host keeps.
 *   suo_frame_geom_launch: asynchronous.  n_frames frames, frame f = crops [frame_first[f], frame_first[f+1]) of the device arrays
 *     uv_dev [L,41,2], cov_dev [L,41,2,2], mask_dev [L,41] (the network's outputs and suo_keypoint_masks' result, same stream),
 *     model_kps_dev float32 [L,41,3].  Host arrays (copied before returning): kinv [L][6] = KinvT[0][0],[1][0],[2][0],[0][1],[1][1],[2][1]
 *     of inv(K_bbox).T (:34), camk [L][4] = fx, fy, cx, cy of K_bbox (:799), min_depth [L] = 0.5 * diameter (:1147).
 *     PnP sampler keys as if suo_pnp_batch were called once per frame with that frame's solvable crops (>= 4 keypoints) only and the
 *     caller advanced `seed` by their number from frame to frame (what ObjectSLAM does between process_view calls).
 *     do_lm = 0 stops after PnP + acceptance (SLAM tracking continues on the host: camera hypotheses :975-1072).
 *     At most 16 crops per frame when do_lm (one wave per object, csrc/lm_frame.hip).
 *   suo_frame_geom_fetch: waits for that launch and points `out` into the context's pinned read-back block (valid until the next launch):
 *     per crop T_pnp [16] (row-major 4x4; pnp_status 1 = identity = failure), accepted, T_opt [12] (refined T_OtoC, = PnP pose when not
 *     refined), n_kp, and per keypoint SLOT (crop * 41 + position among the crop's valid keypoints, i.e. the order of the reference's
 *     boolean indexing) inlier / chi2; uv / cov / mask as the network and suo_keypoint_masks wrote them; lm_stats [n_frames][4]. */
typedef struct suo_frame_geom suo_frame_geom;
typedef struct suo_frame_geom_params {
    double pnp_threshold;          /* 1e-3 (parameters.h:35) */
    uint64_t seed;
    int use_cov;                   /* 0: information = identity (no_network_cov, :825) */
    int do_lm;
    int its[4]; int n_rounds;      /* {10,10,40,40}, 4 (:843-846) */
    double chi2_thr;               /* 5.991 */
    double huber_delta;            /* sqrt(5.991) */
    uint64_t* seed_dev;            /* NULL, or a device-resident running key: the launch samples with seed + *seed_dev (read when its PnP kernel runs) and
                                    * adds its number of solvable problems (crops with >= 4 keypoints) to *seed_dev when done -- what a host caller adds to
                                    * `seed` between launches, without waiting for the previous launch's read-back (a second batch in flight) */
} suo_frame_geom_params;
typedef struct suo_frame_geom_result {
    int n_frames, n_crops;
    const double* T_pnp; const double* T_opt; const double* chi2;
    const int* pnp_status; const int* pnp_best_inliers; const int* pnp_iterations; const int* n_kp; const int* lm_stats;
    const uint8_t* accepted; const uint8_t* inlier;
    const float* uv; const float* cov; const uint8_t* mask;
} suo_frame_geom_result;
int suo_frame_geom_create(int max_crops, int max_frames, suo_frame_geom** out);
void suo_frame_geom_destroy(suo_frame_geom* g);
int suo_frame_geom_launch(suo_frame_geom* g, int n_frames, const int* frame_first, const float* uv_dev, const float* cov_dev,
                          const uint8_t* mask_dev, const float* model_kps_dev, const double* kinv, const double* camk, const double* min_depth,
                          const suo_frame_geom_params* params, void* stream);
int suo_frame_geom_fetch(suo_frame_geom* g, suo_frame_geom_result* out);
int suo_frame_geom_ready(suo_frame_geom* g);      /* 1 when the last launch has completed (or nothing is in flight), 0 otherwise; never blocks */
/* The result block's DEVICE pointers (same fields): valid, stream-ordered, for work enqueued on the launch's stream behind it, until the context's next launch. */
int suo_frame_geom_device_result(suo_frame_geom* g, suo_frame_geom_result* out);

/* ---- SLAM tracking between the two network passes of a view (round 6): camera-hypothesis vote + prior projection on the device -------------
 * Replaces, on the caller's stream and without a host round trip, ObjectSLAM.__estimate_camera_pose (lib/object_slam.py:975-1072) on the detections of the view's
 * FIRST pass and the projection of the prior keypoints of the SECOND pass's objects (:486-514).  Inputs on the device: pass A's chain results
 * (suo_frame_geom_device_result: T_pnp [n_a][16], accepted [n_a], n_kp [n_a]), pass A's network outputs uv / cov, its masks and model keypoints ([n_a,41,...]), pass B's
 * model keypoints [n_b,41,3] float32 and class masks [n_b,41]; block_dev = SUO_SLAM_VOTE_BLOCK doubles staged by the caller (e.g. with its other small arrays):
 *   a_in_map [16] | a_T_OtoG [16][12] | a_K_bbox [16][9] (the float32 container widened, :1082) | b_in_map [16] | b_T_OtoG [16][12] | b_K_bbox [16][9] (fix_K_for_bbox_ndc, double)
 * (rows of objects that are not in the map: in_map 0, the rest ignored).  has_cov / kp_std2 / chi2_max as suo_slam_score; min_inliers = 4 (:975).
 * Outputs on the device: prior_uv_dev [n_b][41][2] float32 + prior_mask_dev [n_b][41] -- what suo_net_forward_prior_kp takes (rows of objects without a prior: zeros) --
 * and out_dev [SUO_SLAM_VOTE_OUT] doubles: T_GtoC [3][4] | best crop (-1: no hypothesis reached min_inliers -- the caller falls back to
 * __backup_estimate_camera_pose and issues pass B again) | number of hypotheses | best count | counts [16] (-1: not a hypothesis) | NaN flag. */
#define SUO_SLAM_VOTE_BLOCK 704
#define SUO_SLAM_VOTE_OUT 32
int suo_slam_vote(int n_a, const double* T_pnp_dev, const uint8_t* accepted_dev, const int* n_kp_dev, const float* uv_dev, const float* cov_dev,
                  const uint8_t* mask_dev, const float* model_kps_a_dev, const double* block_dev, int n_b, const float* model_kps_b_dev,
                  const uint8_t* model_mask_b_dev, int has_cov, double kp_std2, double chi2_max, int min_inliers, float* prior_uv_dev,
                  uint8_t* prior_mask_dev, double* out_dev, void* stream);

/* ---- pose refinement / bundle adjustment: replaces the g2o calls of ObjectSLAM.optimize -------------
 * (lib/object_slam.py:703-903; g2o surface listed in SURVEY.md 8b).  A problem is the flat SoA of the graph
 * the reference builds edge by edge (:746-839): vertices = cameras (T_GtoC) and objects (T_OtoG) as
 * row-major 3x4; one edge per keypoint measurement with cam_k = [fx,fy,cx,cy] of K_bbox, model point,
 * measured uv (NDC) and information (xx,xy,yy).  obj_fixed all 1 reproduces curr_only mode
 * (EdgeSE3ProjectFromFixedObject); cam_fixed[0]=1 the global / single-view gauge (:774).
 * Runs the robust rounds of :842-896 (its[], chi2 gate, Huber drop at round max(1,n_rounds/2)) on-device. */
typedef struct suo_ba_problem {
    int n_cam, n_obj, n_edge;
    double* cam_T;                 /* [n_cam][12] in/out */
    const uint8_t* cam_fixed;      /* [n_cam] */
    double* obj_T;                 /* [n_obj][12] in/out */
    const uint8_t* obj_fixed;      /* [n_obj] */
    const int32_t* edge_cam;       /* [n_edge] */
    const int32_t* edge_obj;       /* [n_edge] */
    const double* edge_camk;       /* [n_edge][4] */
    const double* edge_p;          /* [n_edge][3] */
    const double* edge_uv;         /* [n_edge][2] */
    const double* edge_info;       /* [n_edge][3] = (xx, xy, yy) */
    uint8_t* edge_inlier;          /* [n_edge] in/out (detections[...]["inliers"]) */
    double* edge_chi2;             /* [n_edge] out, may be NULL */
    int its[8]; int n_rounds;      /* e.g. {10,10,40,40}, 4 */
    int init_with_outliers;        /* opt_init_with_outliers and curr_only (:848-852) */
    double chi2_thr;               /* 5.991 */
    double huber_delta;            /* sqrt(5.991) */
    int stats[4];                  /* out: rounds, LM iterations, LM trials, final num_good */
} suo_ba_problem;
int suo_optimize(suo_ba_problem* problem);
/* many independent problems (frames) in one launch, one workgroup each.
 * Which kernels run (all of them g2o's Levenberg-Marquardt on the same graph, optimization_algorithm_levenberg.cpp:58-150; they differ in summation order only):
 *   frame-sized graphs (the per-view refinements)                       one 256-thread workgroup per problem (csrc/lm.hip)
 *   ONE graph of >= 512 edges with free cameras and free objects        the phase kernels of the multi-GPU adjustment below on one rank under the device-resident
 *   (ObjectSLAM.optimize's global adjustment, lib/object_slam.py:746)   schedule, driven from C: ~12 launches per LM trial, the host reads 16 doubles once per <= 12 trials
 *                                                                       (round 6; 60 cameras x 8 objects: 6.8 ms, 10.5 through the grid-barrier kernel of rounds 4-5)
 *   more than 16 free objects next to free cameras                      the same phases under the host-driven schedule (the reduced system then lives in HBM) */
int suo_optimize_batch(suo_ba_problem* problems, int n_problems);

/* ---- phase-wise bundle adjustment for the multi-GPU global pose graph ------------------------
 * Cameras are partitioned across GPUs; each rank builds a context over ITS cameras' edges and ALL objects, and
 * a host driver (suo_slam_amd/ba_dist.py) runs g2o's LM schedule with two all-reduces per trial (RCCL):
 *   suo_ba_linearize   -> [chi2_local | (Hoo 21 + bo 6) per object | max diag Hcc]         (sum / max over ranks)
 *   suo_ba_schur       -> [S_g (ns x ns) | r_g (ns) | ok]   S_g = sum_c Hco^T (Hcc + lambda I)^-1 Hco  (sum)
 *   suo_ba_solve_update(in = [Hoo+bo totals | S_total | r_total]) -> [chi2_local | scale_cams | scale_objs | ok]
 *   suo_ba_restore      = pop() after a rejected trial;  suo_ba_classify = chi2 re-classification of own edges.
 * ns = 6 * (#free objects); beyond 96 rows (16 free objects) the reduced system is factorised in global memory. */
typedef struct suo_ba_ctx suo_ba_ctx;
int suo_ba_ctx_create(suo_ba_problem* local_problem, suo_ba_ctx** out);
void suo_ba_ctx_destroy(suo_ba_ctx* ctx);
int suo_ba_ctx_ns(const suo_ba_ctx* ctx);
int suo_ba_classify(suo_ba_ctx* ctx, int keep_all, double* num_good_local);
int suo_ba_linearize(suo_ba_ctx* ctx, int robust_on, double* out);
int suo_ba_schur(suo_ba_ctx* ctx, double lambda, double* out);
int suo_ba_solve_update(suo_ba_ctx* ctx, double lambda, int robust_on, const double* in, double* out);
int suo_ba_restore(suo_ba_ctx* ctx);
int suo_ba_ctx_download(suo_ba_ctx* ctx, suo_ba_problem* local_problem);
/* The same phases on caller-owned DEVICE buffers: stream-ordered on `stream` (the caller's, e.g. the one RCCL orders its
 * collectives against), no host synchronisation, nothing staged through host memory -- the buffers are all-reduced in place.
 *   lin_dev [1 + 27 n_obj + world] = [chi2_local | (Hoo 21 + bo 6) per object | max |diag Hcc| in slot `rank`, 0 elsewhere]  (SUM)
 *   sch_dev [ns*ns + ns + 1]       = [S_g | r_g | ok]                                                                    (SUM)
 *   red_dev [4]                    = [chi2_local | scale over own cameras | ok | scale over objects (identical on every rank)]
 *                                    (SUM over the first three)
 * suo_ba_solve_update_dev takes the REDUCED lin_dev / sch_dev and treats the trial as failed unless sch_dev's ok count == world.
 * Before suo_ba_ctx_download the caller synchronises `stream`. */
int suo_ba_classify_dev(suo_ba_ctx* ctx, int keep_all, double* num_good_dev, void* stream);
int suo_ba_linearize_dev(suo_ba_ctx* ctx, int robust_on, int rank, int world, double* lin_dev, void* stream);
int suo_ba_schur_dev(suo_ba_ctx* ctx, double lambda, double* sch_dev, void* stream);
int suo_ba_solve_update_dev(suo_ba_ctx* ctx, double lambda, int robust_on, int world, const double* lin_dev, const double* sch_dev,
                            double* red_dev, void* stream);
int suo_ba_restore_dev(suo_ba_ctx* ctx, void* stream);
/* The same phases under a DEVICE-RESIDENT LM schedule: g2o's accept / reject arithmetic (optimization_algorithm_levenberg.cpp:88-148) runs in
 * one-thread kernels on the reduced scalars and keeps its state in ctl_dev [16 doubles]:
 *   [0] lambda [1] ni [2] current chi2 [3] state (0: the next unit linearises, 1: the next unit is a trial on the standing linearisation,
 *   2: the round is over) [4] iterations done this round [5] iteration budget [6] qmax [7] LM iterations [8] LM trials (all rounds; zero them
 *   once) [9] rho [10] restore flag [11] ranks.
 * One UNIT = suo_ba_lm_linearize_dev (live in state 0; always leaves this rank's totals in lin_dev) -> all-reduce(lin_dev) ->
 * suo_ba_lm_schur_dev (chi2 / lambda init of a fresh linearisation, then the Schur phase for ctl's lambda) -> all-reduce(sch_dev) ->
 * suo_ba_lm_solve_update_dev -> all-reduce(red_dev[0:3]) -> suo_ba_lm_decide_dev (gain ratio, lambda / ni, restore of a rejected step,
 * next state).  Phases that are not live return at once, so units may be enqueued blindly -- the host reads ctl_dev once per batch of
 * units instead of four doubles per trial (suo_slam_amd/ba_dist.py: optimize_distributed).  suo_ba_lm_begin_dev starts a round of `its`
 * iterations (optimizer.optimize(its), lib/object_slam.py:873-875). */
int suo_ba_lm_begin_dev(suo_ba_ctx* ctx, double* ctl_dev, int its, int world, void* stream);
int suo_ba_lm_linearize_dev(suo_ba_ctx* ctx, int robust_on, int rank, int world, const double* ctl_dev, double* lin_local_dev, double* lin_dev,
                            void* stream);
int suo_ba_lm_schur_dev(suo_ba_ctx* ctx, double* ctl_dev, const double* lin_dev, double* sch_dev, void* stream);
int suo_ba_lm_solve_update_dev(suo_ba_ctx* ctx, int robust_on, int world, const double* ctl_dev, const double* lin_dev, const double* sch_dev,
                               double* red_dev, void* stream);
int suo_ba_lm_decide_dev(suo_ba_ctx* ctx, double* ctl_dev, const double* red_dev, void* stream);
/* The same unit on ONE rank, where nothing is exchanged between its phases: one call, the control steps folded into the kernels in front of them (12 launches
 * instead of 14), bit-identical to the four calls above back to back. */
int suo_ba_lm_unit_one_rank_dev(suo_ba_ctx* ctx, int robust_on, double* ctl_dev, double* lin_local_dev, double* lin_dev, double* sch_dev, double* red_dev, void* stream);
/* Test entry (finite-difference checks of what the KERNELS linearise, not of the oracle): after suo_ba_linearize, per edge in
 * the caller's order jac_out[e][29] = [J_cam 2x6 | J_obj 2x6 | w*info (xx,xy,yy) | -w*info*err (2)] (EdgeSE3ProjectFromObject::
 * linearizeOplus, types_object_slam.cpp:70-123, columns = [omega, upsilon]) and err_out[e][2] (computeError, :45-60). */
int suo_debug_ba_jacobians(suo_ba_ctx* ctx, int n_edge, double* jac_out, double* err_out);
/* Test entry: the reduced (object) system's solve as the LM kernels run it -- block-6 Cholesky of one workgroup, trailing updates on v_mfma_f64_16x16x4_f64 tiles, the
 * right-hand side eliminated inside the factorisation (csrc/lm_device.h: wg_cholesky_solve; g2o solves the same system with a sparse LL^T, block_solver.hpp:464-566 /
 * linear_solver_cholmod.h) -- on a dense symmetric A [ns][ns] (HOST, row-major, lower triangle read; ns a multiple of 6, at most 96) and b [ns]: x_out [ns] solves
 * A x = b; *ok_out = 0 when a pivot was not positive (the LM kernels then reject the trial). */
int suo_debug_cholesky_solve(const double* A, const double* b, int ns, double* x_out, int* ok_out);

/* ---- evaluation meter: ADD / ADD-S pose errors ------------------------------------
 * Replaces the distance part of EvalMeter.update (lib/utils/eval_meter.py:126-155,233-242):
 *   ADD = mean_i |T_gt p_i - T_pred p_i|,  ADD-S = mean_i min_j |T_gt p_i - T_pred p_j|   (fp32, mesh units = mm).
 * suo_mesh_db_create uploads the model point clouds once (mesh_db[obj]["points"], lib/utils/mesh_database.py:31-40):
 * n_pts[n_models], pts = the clouds concatenated, [sum(n_pts)][3] floats, host memory.
 * suo_pose_errors evaluates n (model, pose pair) items in one launch set: model_index[n] into the database,
 * T_pred / T_gt [n][12] = row-major 3x4 [R|t] (host), results add[n], adds[n] (host).  Blocking. */
int suo_mesh_db_create(int n_models, const int* n_pts, const float* pts, void** mesh_db_out);
void suo_mesh_db_destroy(void* mesh_db);
int suo_pose_errors(void* mesh_db, int n, const int* model_index, const float* T_pred, const float* T_gt, float* add, float* adds);

/* ---- SLAM-mode hypothesis scoring ------------------------------------------
 * Replaces the per-pair numpy of ObjectSLAM.__estimate_camera_pose's hypothesis loop (lib/object_slam.py:1000-1066) and of
 * __maybe_reinit_objects' inlier counts (:619-690): count_k [ z_k > 0 and chi2_k <= chi2_max ] for n_pairs (pose, detection) pairs in one launch.
 * A detection is written once into a device-resident store (suo_slam_store_put; host rows [n][SUO_SLAM_ROW] doubles:
 *   model keypoints [41][3] | predicted uv [41][2] | covariances [41][2][2] | K_bbox [3][3] row-major | n | has_cov (0 / 1), unused keypoints zero);
 * a pair is SUO_SLAM_PAIR doubles: T [3][4] row-major (detection's object -> camera), the keypoint selection as the 64 bits of a uint64
 * (bit k = keypoint k counts: k < n, and its inlier flag where the reference scores inliers only, :1040), the store slot as the 64 bits of an int64.
 * chi2 = r^T Sigma^-1 r with Sigma's diagonal clamped at 1e-4 (:669,:1054), or |r|^2 / kp_std2 for a detection without covariances.
 * counts_host [n_pairs + 1]: the counts, then a flag: some selected keypoint's chi2 was NaN (the reference asserts on it).  Blocking; host buffers. */
#define SUO_SLAM_ROW 380
#define SUO_SLAM_PAIR 14
void* suo_slam_store_create(int capacity);
void suo_slam_store_destroy(void* store);
int suo_slam_store_put(void* store, int first_slot, int n, const double* rows_host);
int suo_slam_score(void* store, int n_pairs, const double* pairs_host, double chi2_max, double kp_std2, int32_t* counts_host);

#ifdef __cplusplus
}
#endif
#endif /* SUO_HIP_H */
