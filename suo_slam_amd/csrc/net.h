// Host-side runtime object behind suo_net_* (see include/suo_hip.h).
#pragma once
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "suo_internal.h"

namespace suo {

struct HostTensor {
    const float* data = nullptr;
    std::vector<int64_t> shape;
    size_t numel = 0;
};

// Wx3: bf16x3 planes (N a multiple of 128, K segments multiples of 64; csrc/gemm_bf16x3.hip); W16 / osc16: the two fp16 planes and the per-column factors (csrc/f16x2.h)
struct GemmW { float* Wp = nullptr; float* bias = nullptr; int N = 0, n_valid = 0, K1 = 0, K2 = 0; float* Wx3 = nullptr; float* W16 = nullptr; float* osc16 = nullptr; };
// Wq: Winograd-packed (3x3, 128 -> 128 / 64 -> 64); Wq3: its bf16x3 planes (csrc/conv_wino_x3.hip); Wq16 / osc16: its fp16 planes and per-channel factors
struct ConvW { float* Wp = nullptr; float* bias = nullptr; int N = 0, C = 0, KS = 0; float* Wq = nullptr; float* Wq3 = nullptr; float* Wq16 = nullptr; float* osc16 = nullptr; };
struct ResidualW {
    float* pro_scale = nullptr; float* pro_shift = nullptr;
    GemmW c1; ConvW c2; GemmW c3;
    int cin = 0, cout = 0; bool has_skip_conv = false;
    float* c3x = nullptr;                                // conv3 as bf16x3 planes for the fused Winograd tail on the bf16 pipe (256 <- 128 only)
    float* c3x16 = nullptr; float* c3osc16 = nullptr;    // ... as two fp16 planes + per-column factors (csrc/f16x2.h)
    // 256 -> 256 blocks: the whole block in one launch on small maps (csrc/res_small.hip fp32 pipe; csrc/res_small_x3.hip bf16 pipe)
    float* rb_w[3] = {nullptr, nullptr, nullptr};        // pack_res16_gemm(W1 bn1-folded) | pack_res16_conv3x3(W2, bn2 scale) | pack_res16_gemm(W3)
    float* rbx_w[3] = {nullptr, nullptr, nullptr};       // the same as bf16x3 planes (uint16)
    float* rbh_w[3] = {nullptr, nullptr, nullptr};       // ... as two fp16 planes (csrc/f16x2.h)
    float* rbh_osc[3] = {nullptr, nullptr, nullptr};     // and their per-channel factors [128], [128], [256]
};
struct HourglassW {
    int n = 0;
    ResidualW up1[2], low1[2], low2[2], low3[2];
    std::unique_ptr<HourglassW> inner;
};

void pack_gemm_weight(const float* W, int N, int K, int ldw, int Np, int Kp, float* out);
void pack_conv_weight(const float* W, int N, int C, int KS, int Np, int Cp, int CK, const float* out_scale, float* out);

class Net {
public:
    Net(int n, const char* const* names, const float* const* data, const int64_t* const* shapes, const int* ndims, int max_crops);
    ~Net();
    int forward(const void* img, int fmt, int H, int W, const float* boxes, const int* box_img, int L, const float* priors,
                const float* prior_uv, const uint8_t* prior_mask, float* uv, float* cov,
                float* kp_prob, float* kp_logit, float* logits_out, hipStream_t s);
    int forward_staged(const float* in0_user, int L, float* logits_out, hipStream_t s);
    int prepare(int L, int with_priors, hipStream_t s);
    void set_use_graph(bool v) { use_graph_ = v; }
    // matrix pipe of the large launches: 0 = fp32 MFMA, 1 = three bf16 terms (6 MFMAs per product block), 2 = two fp16 terms (3 MFMAs; range-guarded)
    int pipe() const { return pipe_; }
    int set_pipe(int p);
    // 1 when a forward since the last call of this function left the fp16 range (its outputs are invalid); clears the flag.  The caller has synchronised.
    int range_exceeded();
    int max_crops() const { return max_crops_; }
    size_t workspace_bytes() const { return ws_floats_ * sizeof(float); }
    int schedule_bytes(int L, int n_frames, int H, int W, int with_priors, double* out, int* n_launches);
    const HostTensor& T(const std::string& name) const;

private:
    static constexpr int kNumSide = 4, kNumEvents = 32;
    float* upload(const std::vector<float>& v);
    void make_gemm(const std::string& conv, const std::string& bn_after, const std::string& conv2, GemmW& g);
    void make_conv(const std::string& conv, const std::string& bn_after, int CK, ConvW& c, int c_used = 0);
    void make_residual(const std::string& p, ResidualW& r);
    void make_hourglass(const std::string& p, int n, HourglassW& h);
    float* alloc(size_t floats);
    // pool_out: also produce max_pool2d(out, 2, 2) (fused into the last GEMM where it can be, else a separate launch); `out` may then be nullptr
    // next: the Residual block that consumes `out` next at this resolution (or nullptr): where this block ends in the fused fp16 Winograd tail, that launch
    // also computes next's conv1 (csrc/conv_wino_x3.hip, NEXT) and residual(*next, out, ...) finds it done (pre_*)
    int residual(const ResidualW& r, const float* x, float* out, int L, int H, int W, hipStream_t s, const float* up = nullptr, float* pool_out = nullptr,
                 const ResidualW* next = nullptr);
    long wino_min_tiles() const;
    long x3_min_rows() const;
    bool next_conv1_fusable(const ResidualW& next, int L, int H, int W) const;
    int gemm_maybe_pooled(GemmArgs& g, int L, int H, int W, float* pool_out, hipStream_t s, const GemmW* gw = nullptr);
    bool residual_tail_is_fused(const ResidualW& r, int L, int H, int W) const;
    int residual_in_one_launch(const ResidualW& r, int L, int H, int W) const;      // 0: no; 1: csrc/res_small.hip; 2: csrc/res_small_x3.hip
    int residual_one_launch(const ResidualW& r, const float* x, float* out, int L, int H, int W, hipStream_t s, const float* up, bool pool_in);
    int hourglass(const HourglassW& h, const float* x, float* out, int L, int H, int W, hipStream_t s, int depth_idx, const float* x_pooled = nullptr);
    int backbone(const float* in0, int in_c, float* logits, int L, hipStream_t s, bool stem_done = false);
    int run_backbone(float* in0, int in_c, float* logits, int L, hipStream_t s, bool stem_done = false);
    int ensure_graph(float* in0, int in_c, float* logits, int L, hipStream_t s, hipGraphExec_t* exec, bool stem_done = false);
    bool fused_stem() const;
    int follow_null_stream();

    std::map<std::string, HostTensor> tensors_;
    std::vector<float*> owned_;
    ConvW stem_, stem_img_;      // all 44 input channels | the 3 image channels only (no priors: SLAM_C = 16)
    float* stem_h2_w_ = nullptr; float* stem_h2_osc_ = nullptr;       // ... as two fp16 planes + per-channel factors (csrc/f16x2.h)
    float* stem_x3_w_ = nullptr; float* stem_x3_bias_ = nullptr;      // the image-only stem as bf16x3 planes for the fused RoIAlign + stem launch (csrc/stem_x3.hip)
    float* stem_slab_ = nullptr;                                      // [max_crops,128,128,64] persistent slab of the stem's output
    float* stem_mid1_slab_ = nullptr;                                 // ... and of r1's conv1 when the fused stem launch computes it (csrc/stem_x3.hip: NEXT)
    bool stem_computes_r1_conv1() const;
    ResidualW r1_, r4_, r5_, post_[2][2];
    HourglassW hg_[2];
    GemmW lin_[2], head_[2], reinject_;
    float* cls_w_ = nullptr; float* cls_b_ = nullptr;
    int max_crops_;
    float* ws_ = nullptr; size_t ws_floats_ = 0, ws_used_ = 0, ws_mark_ = 0;
    float* d_mean_logit_ = nullptr;
    hipStream_t side_[kNumSide] = {}; hipStream_t own_stream_ = nullptr;
    hipEvent_t ev_[kNumEvents] = {}; int ev_next_ = 0;
    bool use_graph_ = true; bool dry_run_ = false;
    bool acct_on_ = false; double acct_[8] = {}; int acct_launches_ = 0;      // schedule_bytes: algorithmic HBM bytes per kind of launch, summed over a dry run
    void acct(int kind, double bytes) { if (acct_on_) acct_[kind] += bytes; }
    int maxpool(const float* in, float* out, int L, int H, int W, int C, hipStream_t s);
    struct PreConv1 { const float* x; const ResidualW* r; float* mid1; };      // conv1 of block r on input x, already computed by the producer's tail
    std::vector<PreConv1> pre_;                                              // (several can be pending: up1[0]'s for up1[1] waits while the low branch runs)
    int pipe_ = 1, pipe_built_ = 1;                      // the pipe in use / the best one the weights were packed for
    unsigned* range_flag_ = nullptr;                     // mapped host memory: the f16x2 kernels raise it, the host reads it after any synchronisation
    // two executables per captured graph, launched alternately: hipGraphLaunch of an executable whose previous launch is still running blocks the host until
    // that one ends (measured: 14 ms per call with a second batch in flight behind ObjectSLAM.submit_views_single) -- with two, the host runs ahead by one call
    struct GraphEntry { hipGraph_t graph = nullptr; hipGraphExec_t exec[2] = {nullptr, nullptr}; int next = 0; };
    std::map<int, GraphEntry> graphs_;
};

}  // namespace suo
