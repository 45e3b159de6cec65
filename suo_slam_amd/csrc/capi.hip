// extern "C" surface of libsuo_hip.so (declared in include/suo_hip.h).
#include <stdarg.h>
#include <stdio.h>

#include "../../include/suo_hip.h"
#include "net.h"

static thread_local char g_err[1024] = "";

void suo_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

struct suo_net { suo::Net* impl; };

extern "C" {

const char* suo_last_error(void) { return g_err; }
int suo_version(void) { return 100; }

int suo_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return -1;
    return n;
}

int suo_net_create(int n, const char* const* names, const float* const* data, const int64_t* const* shapes,
                   const int* ndims, int max_crops, suo_net** out) {
    if (!out || n <= 0 || max_crops <= 0) { suo_set_error("suo_net_create: bad arguments"); return SUO_ERR_ARG; }
    try {
        suo_net* h = new suo_net();
        h->impl = new suo::Net(n, names, data, shapes, ndims, max_crops);
        *out = h;
    } catch (const std::exception& e) {
        suo_set_error("suo_net_create: %s", e.what());
        return SUO_ERR_MISSING;
    }
    return SUO_OK;
}

void suo_net_destroy(suo_net* net) {
    if (!net) return;
    delete net->impl;
    delete net;
}

int suo_net_set_graph(suo_net* net, int enable) {
    if (!net) return SUO_ERR_ARG;
    net->impl->set_use_graph(enable != 0);
    return SUO_OK;
}

int suo_net_prepare(suo_net* net, int L, int with_priors, void* stream) {
    if (!net) return SUO_ERR_ARG;
    return net->impl->prepare(L, with_priors, (hipStream_t)stream);
}

size_t suo_net_workspace_bytes(const suo_net* net) { return net ? net->impl->workspace_bytes() : 0; }
int suo_net_schedule_bytes(suo_net* net, int L, int n_frames, int H, int W, int with_priors, double* bytes6, int* n_launches) {
    if (!net) return SUO_ERR_ARG;
    return net->impl->schedule_bytes(L, n_frames, H, W, with_priors, bytes6, n_launches);
}

int suo_net_get_pipe(const suo_net* net) { return net ? net->impl->pipe() : -1; }
int suo_net_set_pipe(suo_net* net, int pipe) { return net ? net->impl->set_pipe(pipe) : SUO_ERR_ARG; }
int suo_net_range_exceeded(suo_net* net) { return net ? net->impl->range_exceeded() : 0; }

int suo_net_forward(suo_net* net, const void* img, int img_format, int H, int W, const float* boxes, int L, const float* priors,
                    float* uv, float* cov, float* kp_mask, float* kp_logits, float* logits, void* stream) {
    if (!net || !img || !boxes || !uv || !cov || !kp_mask) { suo_set_error("suo_net_forward: null argument"); return SUO_ERR_ARG; }
    return net->impl->forward(img, img_format, H, W, boxes, nullptr, L, priors, nullptr, nullptr, uv, cov, kp_mask, kp_logits, logits, (hipStream_t)stream);
}

int suo_net_forward_frames(suo_net* net, const void* imgs, int img_format, int H, int W, const float* boxes, const int* box_img, int L,
                           const float* priors, float* uv, float* cov, float* kp_mask, float* kp_logits, float* logits, void* stream) {
    if (!net || !imgs || !boxes || !box_img || !uv || !cov || !kp_mask) { suo_set_error("suo_net_forward_frames: null argument"); return SUO_ERR_ARG; }
    return net->impl->forward(imgs, img_format, H, W, boxes, box_img, L, priors, nullptr, nullptr, uv, cov, kp_mask, kp_logits, logits, (hipStream_t)stream);
}

int suo_net_forward_prior_kp(suo_net* net, const void* imgs, int img_format, int H, int W, const float* boxes, const int* box_img, int L,
                             const float* prior_uv, const uint8_t* prior_mask, float* uv, float* cov, float* kp_mask, float* kp_logits,
                             float* logits, void* stream) {
    if (!net || !imgs || !boxes || !prior_uv || !prior_mask || !uv || !cov || !kp_mask) { suo_set_error("suo_net_forward_prior_kp: null argument"); return SUO_ERR_ARG; }
    return net->impl->forward(imgs, img_format, H, W, boxes, box_img, L, nullptr, prior_uv, prior_mask, uv, cov, kp_mask, kp_logits, logits, (hipStream_t)stream);
}

int suo_render_priors(const float* prior_uv, const uint8_t* prior_mask, int L, float* out, void* stream) {
    return suo::launch_render_priors(prior_uv, prior_mask, L, out, (hipStream_t)stream);
}

int suo_net_backbone(suo_net* net, const float* staged, int L, float* logits, void* stream) {
    if (!net) { suo_set_error("suo_net_backbone: null net"); return SUO_ERR_ARG; }
    return net->impl->forward_staged(staged, L, logits, (hipStream_t)stream);
}

int suo_pack_gemm_weight_bf16x3(const float* w, int N, int K, uint16_t* out) {
    if (!w || !out || N <= 0 || (N % 32) || K <= 0 || (K % 16)) { suo_set_error("suo_pack_gemm_weight_bf16x3: bad arguments"); return SUO_ERR_ARG; }
    suo::pack_gemm_weight_bf16x3(w, N, K, out);
    return SUO_OK;
}

int suo_conv1x1_bf16x3(const float* a_dev, int lda, int K, const float* pro_scale_dev, const float* pro_shift_dev, const uint16_t* wp3_dev,
                       const float* bias_dev, float* out_dev, int ldo, int M, int N, int relu, void* stream) {
    if (!a_dev || !wp3_dev || !out_dev) { suo_set_error("suo_conv1x1_bf16x3: null argument"); return SUO_ERR_ARG; }
    return suo::launch_gemm_bf16x3(a_dev, lda, K, pro_scale_dev, pro_shift_dev, wp3_dev, bias_dev, out_dev, ldo, M, N, relu, (hipStream_t)stream);
}

int suo_conv1x1_bf16x3_pool(const float* a1_dev, int lda1, int K1, const float* pro_scale_dev, const float* pro_shift_dev, const float* a2_dev, int lda2, int K2,
                            const uint16_t* wp3_dev, const float* bias_dev, const float* r_dev, int ldr, float* out_dev, int ldo, int M, int N, int relu,
                            int H, int W, float* pool_out_dev, void* stream) {
    if (!a1_dev || !wp3_dev || (!out_dev && !pool_out_dev)) { suo_set_error("suo_conv1x1_bf16x3: null argument"); return SUO_ERR_ARG; }
    suo::GemmArgs g = {};
    g.A1 = a1_dev; g.lda1 = lda1; g.K1 = K1; g.pro_scale = pro_scale_dev; g.pro_shift = pro_shift_dev; g.A2 = a2_dev; g.lda2 = lda2; g.K2 = K2;
    g.bias = bias_dev; g.R = r_dev; g.ldr = ldr; g.out = out_dev; g.ldo = ldo; g.M = M; g.N = N; g.n_valid = N; g.relu = relu;
    g.pool_out = pool_out_dev; g.pool_H = H; g.pool_W = W;
    return suo::launch_gemm_bf16x3_args(g, wp3_dev, (hipStream_t)stream);
}

int suo_conv1x1_bf16x3_ex(const float* a1_dev, int lda1, int K1, const float* pro_scale_dev, const float* pro_shift_dev, const float* a2_dev, int lda2, int K2,
                          const uint16_t* wp3_dev, const float* bias_dev, const float* r_dev, int ldr, float* out_dev, int ldo, int M, int N, int relu, void* stream) {
    return suo_conv1x1_bf16x3_pool(a1_dev, lda1, K1, pro_scale_dev, pro_shift_dev, a2_dev, lda2, K2, wp3_dev, bias_dev, r_dev, ldr, out_dev, ldo, M, N, relu, 0, 0,
                                   nullptr, stream);
}

int suo_upload(void* dst_dev, const void* src_pinned_host, size_t bytes, void* stream) {
    if (!dst_dev || !src_pinned_host) { suo_set_error("suo_upload: null argument"); return SUO_ERR_ARG; }
    return suo::launch_upload(dst_dev, src_pinned_host, bytes, (hipStream_t)stream);
}

int suo_decode_heatmaps(const float* logits, int L, float* uv, float* cov, float* mean_logit, int32_t* argmax_idx, float* prob, void* stream) {
    return suo::launch_decode(logits, L, uv, cov, mean_logit, argmax_idx, prob, (hipStream_t)stream);
}

int suo_classifier(const float* mean_logit, const float* w, const float* b, int L, float* kp_logit, float* kp_prob, void* stream) {
    return suo::launch_classifier(mean_logit, w, b, L, kp_logit, kp_prob, (hipStream_t)stream);
}

int suo_keypoint_masks(const float* uv, const float* cov, const float* kp_prob, const uint8_t* model_mask, int L,
                       float bbox_thresh, float kp_var_thresh, uint8_t* mask, void* stream) {
    return suo::launch_kp_masks(uv, cov, kp_prob, model_mask, L, bbox_thresh, kp_var_thresh, mask, (hipStream_t)stream);
}

int suo_roi_align_concat(const void* img, int img_format, int H, int W, const float* boxes, int L, const float* priors, float* out, void* stream) {
    return suo::launch_roi_align_concat(img, img_format, H, W, boxes, nullptr, L, suo::IN_C, priors, nullptr, nullptr, out, (hipStream_t)stream);
}

int suo_pack_gemm_weight(const float* w, int N, int K, int Np, int Kp, float* out) {
    if (Np % 32 || Kp % 16 || N > Np || K > Kp) { suo_set_error("suo_pack_gemm_weight: bad padding"); return SUO_ERR_ARG; }
    suo::pack_gemm_weight(w, N, K, K, Np, Kp, out);
    return SUO_OK;
}

int suo_pack_conv_weight(const float* w, int N, int C, int KS, int Np, int Cp, int CK, float* out) {
    // (CK = 4: paired taps of the image-only stem)
    if (Np % 32 || Cp % CK || (CK % 8 && CK != 4) || N > Np || C > Cp) { suo_set_error("suo_pack_conv_weight: bad padding"); return SUO_ERR_ARG; }
    suo::pack_conv_weight(w, N, C, KS, Np, Cp, CK, nullptr, out);
    return SUO_OK;
}

int suo_conv1x1(const float* a1, int lda1, int K1, const float* pro_scale, const float* pro_shift, const float* a2, int lda2,
                int K2, const float* wp, const float* bias, const float* r, int ldr, float* out, int ldo, int M, int N,
                int n_valid, int relu, int nchw_hw, void* stream) {
    suo::GemmArgs g = {};
    g.A1 = a1; g.lda1 = lda1; g.K1 = K1; g.pro_scale = pro_scale; g.pro_shift = pro_shift;
    g.A2 = a2; g.lda2 = lda2; g.K2 = a2 ? K2 : 0; g.Wp = wp; g.bias = bias; g.R = r; g.ldr = ldr;
    g.out = out; g.ldo = ldo; g.M = M; g.N = N; g.n_valid = n_valid; g.relu = relu; g.nchw_hw = nchw_hw;
    return suo::launch_gemm1x1(g, (hipStream_t)stream);
}

int suo_conv1x1_pool(const float* a1, int lda1, int K1, const float* pro_scale, const float* pro_shift, const float* a2, int lda2,
                     int K2, const float* wp, const float* bias, const float* r, int ldr, float* out, int ldo, int M, int N, int relu,
                     int H, int W, float* pool_out, void* stream) {
    if (!pool_out) { suo_set_error("suo_conv1x1_pool: pool_out is NULL"); return SUO_ERR_ARG; }
    suo::GemmArgs g = {};
    g.A1 = a1; g.lda1 = lda1; g.K1 = K1; g.pro_scale = pro_scale; g.pro_shift = pro_shift;
    g.A2 = a2; g.lda2 = lda2; g.K2 = a2 ? K2 : 0; g.Wp = wp; g.bias = bias; g.R = r; g.ldr = ldr;
    g.out = out; g.ldo = ldo; g.M = M; g.N = N; g.n_valid = N; g.relu = relu;
    g.pool_out = pool_out; g.pool_H = H; g.pool_W = W;
    return suo::launch_gemm1x1(g, (hipStream_t)stream);
}

int suo_pack_gemm_weight_f16x2(const float* w, int N, int K, uint16_t* out, float* oscale_out) {
    if (!w || !out || !oscale_out || N <= 0 || (N % 32) || K <= 0 || (K % 16)) { suo_set_error("suo_pack_gemm_weight_f16x2: bad arguments"); return SUO_ERR_ARG; }
    suo::pack_gemm_weight_f16x2(w, N, K, out, oscale_out);
    return SUO_OK;
}

int suo_conv1x1_f16x2_pool(const float* a1_dev, int lda1, int K1, const float* pro_scale_dev, const float* pro_shift_dev, const float* a2_dev, int lda2, int K2,
                           const uint16_t* w16_dev, const float* oscale_dev, const float* bias_dev, const float* r_dev, int ldr, float* out_dev, int ldo, int M, int N,
                           int relu, int H, int W, float* pool_out_dev, unsigned* range_flag_dev, void* stream) {
    if (!a1_dev || !w16_dev || !oscale_dev || !range_flag_dev || (!out_dev && !pool_out_dev)) { suo_set_error("suo_conv1x1_f16x2: null argument"); return SUO_ERR_ARG; }
    suo::GemmArgs g = {};
    g.A1 = a1_dev; g.lda1 = lda1; g.K1 = K1; g.pro_scale = pro_scale_dev; g.pro_shift = pro_shift_dev; g.A2 = a2_dev; g.lda2 = lda2; g.K2 = a2_dev ? K2 : 0;
    g.bias = bias_dev; g.R = r_dev; g.ldr = ldr; g.out = out_dev; g.ldo = ldo; g.M = M; g.N = N; g.n_valid = N; g.relu = relu;
    g.pool_out = pool_out_dev; g.pool_H = H; g.pool_W = W; g.oscale = oscale_dev; g.range_flag = range_flag_dev;
    return suo::launch_gemm_f16x2_args(g, w16_dev, (hipStream_t)stream);
}

int suo_conv1x1_f16x2_ex(const float* a1_dev, int lda1, int K1, const float* pro_scale_dev, const float* pro_shift_dev, const float* a2_dev, int lda2, int K2,
                         const uint16_t* w16_dev, const float* oscale_dev, const float* bias_dev, const float* r_dev, int ldr, float* out_dev, int ldo, int M, int N,
                         int relu, unsigned* range_flag_dev, void* stream) {
    return suo_conv1x1_f16x2_pool(a1_dev, lda1, K1, pro_scale_dev, pro_shift_dev, a2_dev, lda2, K2, w16_dev, oscale_dev, bias_dev, r_dev, ldr, out_dev, ldo, M, N, relu, 0, 0,
                                  nullptr, range_flag_dev, stream);
}

int suo_pack_wino_weight_f16x2(const float* w, int N, int C, uint16_t* out, float* oscale_out) {
    if (!w || !out || !oscale_out || N % 32 || C % 16) { suo_set_error("suo_pack_wino_weight_f16x2: bad arguments"); return SUO_ERR_ARG; }
    suo::pack_wino_weight_f16x2(w, N, C, N, C, nullptr, out, oscale_out);
    return SUO_OK;
}

int suo_conv3x3_wino_f16x2_n(const float* in, int L, int H, int W, int channels, const uint16_t* wq16, const float* oscale, const float* bias, float* out, int relu,
                             unsigned* range_flag_dev, void* stream) {
    suo::ConvArgs c = {};
    c.in = in; c.L = L; c.H = H; c.W = W; c.C = channels; c.Wp = (const float*)wq16; c.bias = bias; c.out = out; c.OH = H; c.OW = W; c.N = channels; c.relu = relu;
    c.oscale = oscale; c.range_flag = range_flag_dev;
    return suo::launch_conv3x3_wino_f16x2(c, (hipStream_t)stream);
}

int suo_pack_tail_weight_f16x2(const float* w3, int N2, int K, uint16_t* out, float* oscale_out) {
    if (!w3 || !out || !oscale_out || N2 % 32 || K % 16) { suo_set_error("suo_pack_tail_weight_f16x2: bad arguments"); return SUO_ERR_ARG; }
    suo::pack_tail_weight_f16x2(w3, N2, K, out, oscale_out);
    return SUO_OK;
}

int suo_conv3x3_wino_f16x2_conv1x1_skip_up(const float* in, int L, int H, int W, const uint16_t* wq16, const float* oscale2, const float* bias2, const uint16_t* w3p16,
                                           const float* oscale3, const float* bias3, const float* skip, const float* up, float* out, unsigned* range_flag_dev,
                                           void* stream) {
    suo::ConvArgs c = {};
    c.in = in; c.L = L; c.H = H; c.W = W; c.C = 128; c.Wp = (const float*)wq16; c.bias = bias2; c.out = nullptr; c.OH = H; c.OW = W; c.N = 128; c.relu = 1;
    c.W3p = (const float*)w3p16; c.bias3 = bias3; c.R = skip; c.out2 = out; c.N2 = 256; c.up = up; c.oscale = oscale2; c.oscale3 = oscale3; c.range_flag = range_flag_dev;
    if (up && ((H | W) & 1)) { suo_set_error("suo_conv3x3_wino_f16x2_conv1x1_skip_up: odd map size"); return SUO_ERR_ARG; }
    return suo::launch_conv3x3_wino_f16x2_fused(c, (hipStream_t)stream);
}

int suo_conv3x3_wino_f16x2_conv1x1_skip_up_next(const float* in, int L, int H, int W, const uint16_t* wq16, const float* oscale2, const float* bias2, const uint16_t* w3p16,
                                                const float* oscale3, const float* bias3, const float* skip, const float* up, float* out, const float* next_scale,
                                                const float* next_shift, const uint16_t* next_w1h, const float* next_osc1, const float* next_b1, float* next_out,
                                                unsigned* range_flag_dev, void* stream) {
    suo::ConvArgs c = {};
    c.in = in; c.L = L; c.H = H; c.W = W; c.C = 128; c.Wp = (const float*)wq16; c.bias = bias2; c.out = nullptr; c.OH = H; c.OW = W; c.N = 128; c.relu = 1;
    c.W3p = (const float*)w3p16; c.bias3 = bias3; c.R = skip; c.out2 = out; c.N2 = 256; c.up = up; c.oscale = oscale2; c.oscale3 = oscale3; c.range_flag = range_flag_dev;
    c.n_scale = next_scale; c.n_shift = next_shift; c.n_W1 = (const float*)next_w1h; c.n_osc1 = next_osc1; c.n_b1 = next_b1; c.n_out = next_out;
    if (!next_w1h) { suo_set_error("suo_conv3x3_wino_f16x2_conv1x1_skip_up_next: next_w1h is NULL"); return SUO_ERR_ARG; }
    if (up && ((H | W) & 1)) { suo_set_error("suo_conv3x3_wino_f16x2_conv1x1_skip_up_next: odd map size"); return SUO_ERR_ARG; }
    return suo::launch_conv3x3_wino_f16x2_fused(c, (hipStream_t)stream);
}

int suo_conv_kxk(int KS, const float* in, int L, int H, int W, int C, const float* wp, const float* bias, float* out, int N,
                 int relu, void* stream) {
    suo::ConvArgs c = {};
    c.in = in; c.L = L; c.H = H; c.W = W; c.C = C; c.Wp = wp; c.bias = bias; c.out = out; c.N = N; c.relu = relu;
    if (KS == 3) { c.OH = H; c.OW = W; return suo::launch_conv3x3(c, (hipStream_t)stream); }
    if (KS == 7) { c.OH = H / 2; c.OW = W / 2; return suo::launch_conv7x7s2(c, (hipStream_t)stream); }
    suo_set_error("suo_conv_kxk: KS=%d unsupported", KS);
    return SUO_ERR_ARG;
}

int suo_pack_wino_weight(const float* w, int N, int C, int Np, int Cp, float* out) {
    if (Np % 32 || Cp % 16 || N > Np || C > Cp) { suo_set_error("suo_pack_wino_weight: bad padding"); return SUO_ERR_ARG; }
    suo::pack_wino_weight(w, N, C, Np, Cp, nullptr, out);
    return SUO_OK;
}

int suo_conv3x3_wino(const float* in, int L, int H, int W, int C, const float* wp, const float* bias, float* out, int N, int relu, void* stream) {
    suo::ConvArgs c = {};
    c.in = in; c.L = L; c.H = H; c.W = W; c.C = C; c.Wp = wp; c.bias = bias; c.out = out; c.OH = H; c.OW = W; c.N = N; c.relu = relu;
    return suo::launch_conv3x3_wino(c, (hipStream_t)stream);
}

int suo_conv3x3_wino_conv1x1_skip_up(const float* in, int L, int H, int W, const float* wq2, const float* bias2, const float* wp3, const float* bias3,
                                     const float* skip, const float* up, float* out, void* stream) {
    suo::ConvArgs c = {};
    c.in = in; c.L = L; c.H = H; c.W = W; c.C = 128; c.Wp = wq2; c.bias = bias2; c.out = nullptr; c.OH = H; c.OW = W; c.N = 128; c.relu = 1;
    c.W3p = wp3; c.bias3 = bias3; c.R = skip; c.out2 = out; c.N2 = 256; c.up = up;
    if (up && ((H | W) & 1)) { suo_set_error("suo_conv3x3_wino_conv1x1_skip_up: odd map size"); return SUO_ERR_ARG; }
    return suo::launch_conv3x3_wino_fused(c, (hipStream_t)stream);
}

int suo_pack_wino_weight_bf16x3(const float* w, int N, int C, uint16_t* out) {
    if (!w || !out || N % 32 || C % 16) { suo_set_error("suo_pack_wino_weight_bf16x3: bad arguments"); return SUO_ERR_ARG; }
    suo::pack_wino_weight_bf16x3(w, N, C, N, C, nullptr, out);
    return SUO_OK;
}

int suo_conv3x3_wino_x3_n(const float* in, int L, int H, int W, int channels, const uint16_t* wq3, const float* bias, float* out, int relu, void* stream) {
    suo::ConvArgs c = {};
    c.in = in; c.L = L; c.H = H; c.W = W; c.C = channels; c.Wp = (const float*)wq3; c.bias = bias; c.out = out; c.OH = H; c.OW = W; c.N = channels; c.relu = relu;
    return suo::launch_conv3x3_wino_x3(c, (hipStream_t)stream);
}

int suo_conv3x3_wino_x3(const float* in, int L, int H, int W, const uint16_t* wq3, const float* bias, float* out, int relu, void* stream) {
    return suo_conv3x3_wino_x3_n(in, L, H, W, 128, wq3, bias, out, relu, stream);
}

int suo_pack_tail_weight_bf16x3(const float* w3, int N2, int K, uint16_t* out) {
    if (!w3 || !out || N2 % 32 || K % 16) { suo_set_error("suo_pack_tail_weight_bf16x3: bad arguments"); return SUO_ERR_ARG; }
    suo::pack_tail_weight_bf16x3(w3, N2, K, out);
    return SUO_OK;
}

int suo_conv3x3_wino_x3_conv1x1_skip_up(const float* in, int L, int H, int W, const uint16_t* wq3, const float* bias2, const void* wp3, int tail_bf16x3,
                                        const float* bias3, const float* skip, const float* up, float* out, void* stream) {
    suo::ConvArgs c = {};
    c.w3_bf16x3 = tail_bf16x3;
    c.in = in; c.L = L; c.H = H; c.W = W; c.C = 128; c.Wp = (const float*)wq3; c.bias = bias2; c.out = nullptr; c.OH = H; c.OW = W; c.N = 128; c.relu = 1;
    c.W3p = (const float*)wp3; c.bias3 = bias3; c.R = skip; c.out2 = out; c.N2 = 256; c.up = up;
    if (up && ((H | W) & 1)) { suo_set_error("suo_conv3x3_wino_x3_conv1x1_skip_up: odd map size"); return SUO_ERR_ARG; }
    return suo::launch_conv3x3_wino_x3_fused(c, (hipStream_t)stream);
}

int suo_conv3x3_wino_conv1x1_skip(const float* in, int L, int H, int W, const float* wq2, const float* bias2, const float* wp3, const float* bias3,
                                  const float* skip, float* out, void* stream) {
    return suo_conv3x3_wino_conv1x1_skip_up(in, L, H, W, wq2, bias2, wp3, bias3, skip, nullptr, out, stream);
}

int suo_conv3x3_conv1x1_skip(const float* in, int L, int H, int W, const float* wp2, const float* bias2, const float* wp3, const float* bias3,
                             const float* skip, float* out, void* stream) {
    suo::ConvArgs c = {};
    c.in = in; c.L = L; c.H = H; c.W = W; c.C = 128; c.Wp = wp2; c.bias = bias2; c.out = nullptr; c.OH = H; c.OW = W; c.N = 128; c.relu = 1;
    c.W3p = wp3; c.bias3 = bias3; c.R = skip; c.out2 = out; c.N2 = 256;
    return suo::launch_conv3x3_fused(c, (hipStream_t)stream);
}

int suo_pack_res_block(const float* w1, const float* w2, const float* scale2, const float* w3, float* w1p, float* w2p, float* w3p) {
    if (!w1 || !w2 || !w3 || !w1p || !w2p || !w3p) { suo_set_error("suo_pack_res_block: null argument"); return SUO_ERR_ARG; }
    suo::pack_res16_gemm(w1, 128, 256, w1p);
    suo::pack_res16_conv3x3(w2, 128, 128, scale2, w2p);
    suo::pack_res16_gemm(w3, 256, 128, w3p);
    return SUO_OK;
}

int suo_res_block(const float* x, int L, int H, int W, int pool_in, const float* pro_scale, const float* pro_shift, const float* w1p, const float* b1,
                  const float* w2p, const float* b2, const float* w3p, const float* b3, const float* up, float* out, void* stream) {
    suo::ResBlockArgs a = {};
    a.x = x; a.L = L; a.H = H; a.W = W; a.pool_in = pool_in; a.pro_scale = pro_scale; a.pro_shift = pro_shift; a.W1 = w1p; a.b1 = b1; a.W2 = w2p; a.b2 = b2;
    a.W3 = w3p; a.b3 = b3; a.up = up; a.out = out;
    return suo::launch_res_block(a, (hipStream_t)stream);
}

int suo_pack_res_block_bf16x3(const float* w1, const float* w2, const float* scale2, const float* w3, uint16_t* w1x, uint16_t* w2x, uint16_t* w3x) {
    if (!w1 || !w2 || !w3 || !w1x || !w2x || !w3x) { suo_set_error("suo_pack_res_block_bf16x3: null argument"); return SUO_ERR_ARG; }
    suo::pack_gemm_weight_bf16x3(w1, 128, 256, w1x);
    suo::pack_res_conv3x3_bf16x3(w2, scale2, w2x);
    suo::pack_gemm_weight_bf16x3(w3, 256, 128, w3x);
    return SUO_OK;
}

int suo_res_block_bf16x3(const float* x, int L, int H, int W, int pool_in, const float* pro_scale, const float* pro_shift, const uint16_t* w1x, const float* b1,
                         const uint16_t* w2x, const float* b2, const uint16_t* w3x, const float* b3, const float* up, float* out, void* stream) {
    suo::ResBlockArgs a = {};
    a.x = x; a.L = L; a.H = H; a.W = W; a.pool_in = pool_in; a.pro_scale = pro_scale; a.pro_shift = pro_shift; a.W1 = (const float*)w1x; a.b1 = b1;
    a.W2 = (const float*)w2x; a.b2 = b2; a.W3 = (const float*)w3x; a.b3 = b3; a.up = up; a.out = out;
    return suo::launch_res_block_x3(a, (hipStream_t)stream);
}

int suo_pack_res_block_f16x2(const float* w1, const float* w2, const float* scale2, const float* w3, uint16_t* w1h, uint16_t* w2h, uint16_t* w3h, float* osc1, float* osc2,
                             float* osc3) {
    if (!w1 || !w2 || !w3 || !w1h || !w2h || !w3h || !osc1 || !osc2 || !osc3) { suo_set_error("suo_pack_res_block_f16x2: null argument"); return SUO_ERR_ARG; }
    suo::pack_gemm_weight_f16x2(w1, 128, 256, w1h, osc1);
    suo::pack_res_conv3x3_f16x2(w2, scale2, w2h, osc2);
    suo::pack_gemm_weight_f16x2(w3, 256, 128, w3h, osc3);
    return SUO_OK;
}

int suo_res_block_f16x2(const float* x, int L, int H, int W, int pool_in, const float* pro_scale, const float* pro_shift, const uint16_t* w1h, const float* osc1,
                        const float* b1, const uint16_t* w2h, const float* osc2, const float* b2, const uint16_t* w3h, const float* osc3, const float* b3, const float* up,
                        float* out, unsigned* range_flag_dev, void* stream) {
    suo::ResBlockArgs a = {};
    a.x = x; a.L = L; a.H = H; a.W = W; a.pool_in = pool_in; a.pro_scale = pro_scale; a.pro_shift = pro_shift;
    a.W1 = (const float*)w1h; a.b1 = b1; a.W2 = (const float*)w2h; a.b2 = b2; a.W3 = (const float*)w3h; a.b3 = b3; a.up = up; a.out = out;
    a.osc1 = osc1; a.osc2 = osc2; a.osc3 = osc3; a.range_flag = range_flag_dev;
    return suo::launch_res_block_f16x2(a, (hipStream_t)stream);
}

int suo_conv1x1_chain_head_f16x2(const float* a, int lda, int M, const uint16_t* w1h, const float* osc1, const float* bias1, const uint16_t* w2h, const float* osc2,
                                 const float* bias2, float* out_nchw, int n_valid, int hw, unsigned* range_flag_dev, void* stream) {
    return suo::launch_gemm_chain_head(a, lda, M, w1h, osc1, bias1, w2h, osc2, bias2, out_nchw, n_valid, hw, range_flag_dev, (hipStream_t)stream);
}

int suo_pack_stem_weight_bf16x3(const float* w, int Cw, const float* scale, uint16_t* out) {
    if (!w || !out || Cw < 3) { suo_set_error("suo_pack_stem_weight_bf16x3: bad arguments"); return SUO_ERR_ARG; }
    suo::pack_stem_weight_bf16x3(w, Cw, scale, out);
    return SUO_OK;
}

int suo_pack_stem_weight_f16x2(const float* w, int Cw, const float* scale, uint16_t* out, float* oscale_out) {
    if (!w || !out || !oscale_out || Cw < 3) { suo_set_error("suo_pack_stem_weight_f16x2: bad arguments"); return SUO_ERR_ARG; }
    suo::pack_stem_weight_f16x2(w, Cw, scale, out, oscale_out);
    return SUO_OK;
}

int suo_stem_f16x2(const void* img, int fmt, int H, int W, const float* boxes, const int* box_img, int L, const uint16_t* wh, const float* oscale, const float* bias,
                   float* out, unsigned* range_flag_dev, void* stream) {
    if (!oscale || !range_flag_dev) { suo_set_error("suo_stem_f16x2: null argument"); return SUO_ERR_ARG; }
    return suo::launch_stem_x3(img, fmt, H, W, boxes, box_img, L, wh, bias, out, (hipStream_t)stream, oscale, range_flag_dev);
}

int suo_stem_f16x2_next(const void* img, int fmt, int H, int W, const float* boxes, const int* box_img, int L, const uint16_t* wh, const float* oscale, const float* bias,
                        float* out, const float* n_scale, const float* n_shift, const uint16_t* n_w1h, const float* n_osc1, const float* n_b1, float* n_out,
                        unsigned* range_flag_dev, void* stream) {
    if (!oscale || !range_flag_dev) { suo_set_error("suo_stem_f16x2_next: null argument"); return SUO_ERR_ARG; }
    const suo::StemNext nx = {n_scale, n_shift, n_w1h, n_osc1, n_b1, n_out};
    return suo::launch_stem_x3(img, fmt, H, W, boxes, box_img, L, wh, bias, out, (hipStream_t)stream, oscale, range_flag_dev, &nx);
}

int suo_stem_x3(const void* img, int fmt, int H, int W, const float* boxes, const int* box_img, int L, const uint16_t* wx, const float* bias, float* out,
                void* stream) {
    return suo::launch_stem_x3(img, fmt, H, W, boxes, box_img, L, wx, bias, out, (hipStream_t)stream);
}

int suo_maxpool2(const float* in, float* out, int L, int H, int W, int C, void* stream) {
    return suo::launch_maxpool2(in, out, L, H, W, C, (hipStream_t)stream);
}

int suo_upsample2_add(const float* up1, const float* low, float* out, int L, int H, int W, int C, void* stream) {
    return suo::launch_upsample2_add(up1, low, out, L, H, W, C, (hipStream_t)stream);
}

}  // extern "C"
