// Internal declarations shared by the HIP translation units of libsuo_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#define SUO_OK 0
#define SUO_ERR_ARG 1
#define SUO_ERR_HIP 2
#define SUO_ERR_MISSING 3

#define SUO_HIP_CHECK(expr)                                                            \
    do {                                                                               \
        hipError_t _e = (expr);                                                        \
        if (_e != hipSuccess) {                                                        \
            suo_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return SUO_ERR_HIP;                                                        \
        }                                                                              \
    } while (0)

void suo_set_error(const char* fmt, ...);

namespace suo {

constexpr int NUM_KP = 41;
constexpr int HEAT = 64;          // heat-map side
constexpr int CROP = 256;         // network input side
constexpr int IN_C = 48;          // 3 + 41 = 44 input channels, padded to 48 in the NHWC staging buffer
#ifndef SUO_IMG_C
#define SUO_IMG_C 4
#endif
constexpr int IMG_C = SUO_IMG_C;  // staging without priors: 3 image channels + 1 pad.  The stem then runs a DENSE K axis (tap * 3 +
                                  // channel, two consecutive terms per MFMA k-step: csrc/conv.hip, PAIR mode) -- 76 instead of 196
                                  // MFMAs per accumulator, and the nonzero products are summed in the order of the 48-channel stem, so
                                  // the prior-less pass stays bit-identical to feeding zero priors (tests/test_gpu_cnn.py checks it).
                                  // -DSUO_IMG_C=8: the older form, one 8-channel chunk with 5 zero channels.

// ---- packed-weight geometry (B operand of v_mfma_f32_32x32x2_f32) -------------------------------
// A GEMM weight W[N][K] (row = output channel) is stored as  Wp[K/8][N/32][64 lanes][4]  with
//   Wp[kb][nb][lane][t] = W[nb*32 + (lane&31)][kb*8 + (lane>>5)*4 + t]
// so one wave reads its four next MFMA B operands with ONE coalesced 16-byte-per-lane load.
size_t packed_weight_floats(int n_pad, int k_pad);

// 1x1 convolution (= GEMM over NHWC pixels) with fused prologue / epilogue.
struct GemmArgs {
    const float* A1; int lda1; int K1;                   // [M, K1] activations, row stride lda1
    const float* pro_scale; const float* pro_shift;      // optional: A1 <- relu(A1*scale[k]+shift[k])
    const float* A2; int lda2; int K2;                   // optional second operand (skip conv4 / tmpOut_)
    const float* Wp;                                     // packed [(K1+K2)/8][N/32][64][4]
    const float* bias;                                   // [N] (padded)
    const float* R; int ldr;                             // optional residual [M, n_valid]
    float* out; int ldo;                                 // [M, ldo]   (NHWC)   or NCHW when nchw_hw>0
    int M; int N; int n_valid; int relu; int nchw_hw;    // nchw_hw = H*W of one crop for NCHW output
    // optional fused 2x2 max-pool of the result (nn.MaxPool2d(2, 2) after a block, hg.py:41 / pkpnet.py stem): the M pixels are
    // images of pool_H x pool_W, pool_out is [M / 4, ldo]; `out` may then be nullptr (only the pooled tensor is wanted)
    float* pool_out; int pool_H; int pool_W;
    // f16x2 form only (csrc/f16x2.h): per-column factor 2^-(t_n + S2_XSHIFT) that brings the accumulator back to scale, and the range-guard flag
    const float* oscale; unsigned* range_flag;
};
int launch_gemm1x1(const GemmArgs& a, hipStream_t s);
bool gemm1x1_can_pool(const GemmArgs& a);                // shapes the fused pool takes (128 x 128 tiles of 2 image rows x 64 columns)

// KxK convolution (3x3 s1 p1 or 7x7 s2 p3), NHWC, input already activated, zero padding.
struct ConvArgs {
    const float* in; int L, H, W, C;                     // [L,H,W,C]
    const float* Wp;                                     // packed with K' = [chunk][ky][kx][kk]
    const float* bias; float* out; int OH, OW, N;        // out [L,OH,OW,N]
    int relu;
    // fused Residual tail (launch_conv3x3_fused only): out2 = W3 relu(conv + bias) + bias3 + R, [L,OH,OW,N2]; `out` is not written
    const float* W3p; const float* bias3; const float* R; float* out2; int N2;
    const float* up;                                     // optional [L,OH/2,OW/2,N2]: out2 += nearest-neighbour 2x up-sampling of it (hg.py:56-58)
    int w3_bf16x3;                                       // csrc/conv_wino_x3.hip only: W3p holds pack_tail_weight_bf16x3's uint16 planes
    // f16x2 form only (csrc/f16x2.h): per-channel factors 2^-(t_n + S2_XSHIFT) of the 3x3 convolution [N] and of the tail's conv3 [N2], range-guard flag
    const float* oscale; const float* oscale3; unsigned* range_flag;
    // ... and, optional, the NEXT block's conv1 in the same launch (csrc/conv_wino_x3.hip, NEXT): its BatchNorm prologue [256], its weights as two fp16 planes
    // (pack_gemm_weight_f16x2 of W1 [128][256]) with their factors [128], its folded bias [128]; n_out [L,OH,OW,128] = relu(bn1(conv1(relu(bn(out2)))))
    const float* n_scale; const float* n_shift; const float* n_W1; const float* n_osc1; const float* n_b1; float* n_out;
};
int launch_conv3x3(const ConvArgs& a, hipStream_t s);
bool conv3x3_fusable(const ConvArgs& a);
// Winograd F(2x2,3x3) form (csrc/conv_wino.hip): weights packed by pack_wino_weight (16 floats per (n, c))
void pack_wino_weight(const float* W, int N, int C, int Np, int Cp, const float* out_scale, float* out);
int launch_conv3x3_wino(const ConvArgs& a, hipStream_t s);
int launch_conv3x3_wino_fused(const ConvArgs& a, hipStream_t s);
bool conv3x3_wino_pays(const ConvArgs& a, long min_tiles = -1);      // enough 8 x 16-pixel tiles (min_tiles < 0: SUO_CONV_WINO_TILES, default 256 -- measured on the fp32-pipe kernel)
int launch_conv3x3_fused(const ConvArgs& a, hipStream_t s);
int launch_conv3x3_small(const ConvArgs& a, hipStream_t s);
int launch_gemm_small(const GemmArgs& a, hipStream_t s);
int launch_gemm_persist(const GemmArgs& a, int cfg, hipStream_t s);    // cfg 1: 128x128, 2: 128x64, 3: 64x64 tiles
int launch_conv7x7s2(const ConvArgs& a, hipStream_t s);

// A whole 256 -> 256 Residual block in one launch on small maps (csrc/res_small.hip): out = W3 relu(conv3x3(relu(W1 relu(x * scale + shift) + b1)) + b2) + b3 + x [+ up]
struct ResBlockArgs {
    const float* x; int L, H, W;                         // [L,H,W,256] -- or, pool_in: [L,2H,2W,256] whose 2x2 max-pool is the block's input
    int pool_in;
    const float* pro_scale; const float* pro_shift;      // the block's leading BatchNorm, [256]
    const float* W1; const float* b1;                    // pack_res16_gemm(W1 * bn1 scale [128][256]), [128]
    const float* W2; const float* b2;                    // pack_res16_conv3x3(W2 [128][128][3][3], bn2 scale), [128]
    const float* W3; const float* b3;                    // pack_res16_gemm(W3 [256][128]), [256]
    const float* up;                                     // optional [L,H/2,W/2,256]: out += its nearest-neighbour 2x up-sampling (hg.py:56-58)
    float* out;                                          // [L,H,W,256]
    // f16x2 form only (csrc/f16x2.h): per-channel factors 2^-(t_n + S2_XSHIFT) of conv1 [128], conv2 [128], conv3 [256]; the range-guard flag
    const float* osc1; const float* osc2; const float* osc3; unsigned* range_flag;
};
void pack_res16_gemm(const float* W, int N, int K, float* out);
void pack_res16_conv3x3(const float* W, int N, int C, const float* out_scale, float* out);
bool res_block_takes(const ResBlockArgs& a);
int launch_res_block(const ResBlockArgs& a, hipStream_t s);
// the same block with its products on the bf16 matrix pipe (csrc/res_small_x3.hip; 4 x 8 pixel tiles): W1 / W3 = pack_gemm_weight_bf16x3 planes,
// W2 = pack_res_conv3x3_bf16x3 planes (uint16, passed through the float pointers of ResBlockArgs)
void pack_res_conv3x3_bf16x3(const float* W, const float* out_scale, uint16_t* out);
int launch_res_block_x3(const ResBlockArgs& a, hipStream_t s);
void pack_res_conv3x3_f16x2(const float* W, const float* out_scale, uint16_t* out, float* oscale_out);
int launch_res_block_f16x2(const ResBlockArgs& a, hipStream_t s);

int launch_maxpool2(const float* in, float* out, int L, int H, int W, int C, hipStream_t s);
int launch_upsample2_add(const float* up1, const float* low, float* out, int L, int H, int W, int C, hipStream_t s);
int launch_roi_align_concat(const void* img, int fmt, int H, int W, const float* boxes, const int* box_img, int L, int out_c,
                            const float* priors, const float* prior_uv, const uint8_t* prior_mask, float* out, hipStream_t s);
int launch_render_priors(const float* uv, const uint8_t* mask, int L, float* out, hipStream_t s);
// experimental: fp32-accurate 1x1 convolution on the bf16 matrix pipe (csrc/gemm_bf16x3.hip)
void pack_gemm_weight_bf16x3(const float* W, int N, int K, uint16_t* out);
bool gemm_bf16x3_takes(const GemmArgs& g);
int launch_gemm_bf16x3_args(const GemmArgs& g, const uint16_t* Wx3, hipStream_t s);
int launch_gemm_bf16x3(const float* A, int lda, int K, const float* pro_scale, const float* pro_shift, const uint16_t* Wp, const float* bias,
                       float* out, int ldo, int M, int N, int relu, hipStream_t s);
// the same kernel on two fp16 terms per operand, three MFMAs per product block (csrc/f16x2.h): oscale_out[N] = 2^-(t_n + S2_XSHIFT) goes into GemmArgs.oscale,
// GemmArgs.range_flag must name the guard flag; out = 2 * N * K uint16
void pack_gemm_weight_f16x2(const float* W, int N, int K, uint16_t* out, float* oscale_out);
int launch_gemm_f16x2_args(const GemmArgs& g, const uint16_t* W16, hipStream_t s);
// two 1x1 convolutions in one launch, the 256-channel tensor between them kept in LDS (csrc/gemm_bf16x3.hip: gemm_chain_head_kernel): NCHW logits = W2 relu(W1 A + b1) + b2
bool gemm_chain_head_takes(int M, int lda, int n_valid, int hw);
int launch_gemm_chain_head(const float* A, int lda, int M, const uint16_t* W1h, const float* osc1, const float* bias1, const uint16_t* W2h, const float* osc2, const float* bias2,
                           float* out, int n_valid, int hw, unsigned* range_flag, hipStream_t s);
// experimental: Winograd 3x3 with its products on the bf16 matrix pipe at fp32 accuracy (csrc/conv_wino_x3.hip); ConvArgs.Wp = packed uint16
void pack_tail_weight_bf16x3(const float* W3, int N2, int K, uint16_t* out);
void pack_wino_weight_bf16x3(const float* W, int N, int C, int Np, int Cp, const float* out_scale, uint16_t* out);
int launch_conv3x3_wino_x3(const ConvArgs& a, hipStream_t s);
int launch_conv3x3_wino_x3_fused(const ConvArgs& a, hipStream_t s);
// ... and on two fp16 terms per operand (csrc/f16x2.h; ConvArgs.oscale / oscale3 / range_flag set)
void pack_tail_weight_f16x2(const float* W3, int N2, int K, uint16_t* out, float* oscale_out);
void pack_wino_weight_f16x2(const float* W, int N, int C, int Np, int Cp, const float* out_scale, uint16_t* out, float* oscale_out);
int launch_conv3x3_wino_f16x2(const ConvArgs& a, hipStream_t s);
int launch_conv3x3_wino_f16x2_fused(const ConvArgs& a, hipStream_t s);
bool conv3x3_wino_f16x2_w8(long tiles);                // launches of <= one workgroup per CU: the eight-wave form (no next-block conv1 there)
// RoIAlign + stem 7x7 / 2 (+ BN + ReLU) of the prior-less pass in one launch on the bf16 pipe (csrc/stem_x3.hip)
void pack_stem_weight_bf16x3(const float* W, int Cw, const float* out_scale, uint16_t* out);
// osc / range_flag: both null = Wx holds three bf16 planes; both set = two fp16 planes (pack_stem_weight_f16x2) with their per-channel factors and the guard flag
// next (fp16 form only): the first Residual block's conv1 on the tile -- relu(scale x + shift) W1^T (64 -> 64, pack_gemm_weight_f16x2 planes) * osc1 + b1, ReLU -> out
struct StemNext { const float* scale; const float* shift; const uint16_t* W1; const float* osc1; const float* b1; float* out; };
int launch_stem_x3(const void* img, int fmt, int H, int W, const float* boxes, const int* box_img, int L, const uint16_t* Wx, const float* bias,
                   float* out, hipStream_t s, const float* osc = nullptr, unsigned* range_flag = nullptr, const StemNext* next = nullptr);
void pack_stem_weight_f16x2(const float* W, int Cw, const float* out_scale, uint16_t* out, float* oscale_out);
int launch_upload(void* dst_dev, const void* src_host, size_t bytes, hipStream_t s);
int launch_decode(const float* logits, int L, float* uv, float* cov, float* mean_logit, int* argmax_idx, float* prob, hipStream_t s);
int launch_classifier(const float* mean_logit, const float* Wc, const float* bc, int L,
                      float* kp_logit, float* kp_prob, hipStream_t s);
int launch_kp_masks(const float* uv, const float* cov, const float* kp_prob, const uint8_t* model_mask,
                    int L, float bbox_thresh, float kp_var_thresh, uint8_t* out_mask, hipStream_t s);

}  // namespace suo
