// Host-side runtime of the keypoint CNN: weight folding/packing, workspace, launch schedule,
// multi-stream hourglass branches and hipGraph capture of the backbone.
//
// Mirrors PkpNet.forward (/root/reference/lib/models/pkpnet.py:80-119) over a state_dict with the
// reference's key names (backbone.* / classifier.2.*), see suo_slam_amd/weights.py.
#include "net.h"
#include "f16x2.h"

#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <string>
#include "tune.h"

namespace suo {

static constexpr float BN_EPS = 1e-5f;

// ------------------------------------------------------------------------------------------------
// weight lookup helpers
const HostTensor& Net::T(const std::string& name) const {
    auto it = tensors_.find(name);
    if (it == tensors_.end()) throw std::runtime_error("missing tensor: " + name);
    return it->second;
}

// A checkpoint tensor must have exactly the shape the architecture (lib/models/hg.py:61-93, layers/Residual.py:7-18)
// gives it: anything else would be read out of bounds or silently zero-padded.
static void expect_shape(const Net& net, const std::string& name, std::initializer_list<int64_t> dims) {
    const HostTensor& t = net.T(name);
    if (t.shape.size() == dims.size() && std::equal(dims.begin(), dims.end(), t.shape.begin())) return;
    std::string got, want;
    for (int64_t d : t.shape) got += (got.empty() ? "" : ",") + std::to_string(d);
    for (int64_t d : dims) want += (want.empty() ? "" : ",") + std::to_string(d);
    throw std::runtime_error("tensor " + name + " has shape [" + got + "], expected [" + want + "]");
}
static void expect_bn(const Net& net, const std::string& p, int64_t c) {
    for (const char* f : {".weight", ".bias", ".running_mean", ".running_var"}) expect_shape(net, p + f, {c});
}
static void expect_conv(const Net& net, const std::string& p, int64_t cout, int64_t cin, int64_t k) {
    expect_shape(net, p + ".weight", {cout, cin, k, k});
    expect_shape(net, p + ".bias", {cout});
}

// BN(eval) as y = x*scale + shift
static void bn_affine(const Net& net, const std::string& p, std::vector<float>& scale, std::vector<float>& shift) {
    const HostTensor& g = net.T(p + ".weight");
    const HostTensor& b = net.T(p + ".bias");
    const HostTensor& m = net.T(p + ".running_mean");
    const HostTensor& v = net.T(p + ".running_var");
    const size_t c = g.numel;
    scale.resize(c);
    shift.resize(c);
    for (size_t i = 0; i < c; ++i) {
        const float s = g.data[i] / sqrtf(v.data[i] + BN_EPS);
        scale[i] = s;
        shift[i] = b.data[i] - m.data[i] * s;
    }
}

// Pack W[n][k] (n < N, k < K; zero beyond) into the MFMA B-operand layouts.  `out` holds 2*Np*Kp floats:
//   [0, Np*Kp)        32x32x2 form  [Kp/8][Np/32][64][4]:  W[nb*32+(lane&31)][kb*8 +(lane>>5)*4+t]
//   [Np*Kp, 2*Np*Kp)  16x16x4 form  [Kp/16][Np/16][64][4]: W[nb*16+(lane&15)][kg*16+(lane>>4)*4+t]   (small-map kernels)
void pack_gemm_weight(const float* W, int N, int K, int ldw, int Np, int Kp, float* out) {
    {
        float* o16 = out + (size_t)Np * Kp;
        const int NB16 = Np / 16;
        for (int kg = 0; kg < Kp / 16; ++kg)
            for (int nb = 0; nb < NB16; ++nb)
                for (int lane = 0; lane < 64; ++lane)
                    for (int t = 0; t < 4; ++t) {
                        const int n = nb * 16 + (lane & 15), k = kg * 16 + (lane >> 4) * 4 + t;
                        o16[(((size_t)kg * NB16 + nb) * 64 + lane) * 4 + t] = (n < N && k < K) ? W[(size_t)n * ldw + k] : 0.f;
                    }
    }
    const int NB = Np / 32;
    for (int kb = 0; kb < Kp / 8; ++kb)
        for (int nb = 0; nb < NB; ++nb)
            for (int lane = 0; lane < 64; ++lane)
                for (int t = 0; t < 4; ++t) {
                    const int n = nb * 32 + (lane & 31), k = kb * 8 + (lane >> 5) * 4 + t;
                    out[(((size_t)kb * NB + nb) * 64 + lane) * 4 + t] = (n < N && k < K) ? W[(size_t)n * ldw + k] : 0.f;
                }
}

// Conv weight W[n][c][ky][kx] -> GEMM weight with K' = [chunk][ky][kx][kk] (kk < CK), then packed.
void pack_conv_weight(const float* W, int N, int C, int KS, int Np, int Cp, int CK, const float* out_scale, float* out) {
    const int nch = Cp / CK, Kraw = nch * KS * KS * CK, Kp = (Kraw + 15) / 16 * 16;
    std::vector<float> g((size_t)Np * Kp, 0.f);
    if (CK == 4) {
        // image-only stem (csrc/conv.hip, PAIR mode): dense K axis kk = tap * 3 + channel; MFMA k-step t of group kb multiplies
        // kk = 8 kb + 2 t (lanes 0-31) and 8 kb + 2 t + 1 (lanes 32-63), which pack_gemm_weight reads from column kb*8 + half*4 + t
        for (int n = 0; n < N; ++n)
            for (int kk = 0; kk < KS * KS * 3; ++kk) {
                const int tap = kk / 3, c = kk % 3;
                if (c >= C) continue;
                const int kb = kk / 8, r = kk % 8, t = r / 2, half = r % 2;
                const float s = out_scale ? out_scale[n] : 1.f;
                g[(size_t)n * Kp + kb * 8 + half * 4 + t] = W[(((size_t)n * C + c) * KS + tap / KS) * KS + tap % KS] * s;
            }
        pack_gemm_weight(g.data(), Np, Kp, Kp, Np, Kp, out);
        return;
    }
    for (int n = 0; n < N; ++n)
        for (int ch = 0; ch < nch; ++ch)
            for (int ky = 0; ky < KS; ++ky)
                for (int kx = 0; kx < KS; ++kx)
                    for (int kk = 0; kk < CK; ++kk) {
                        const int c = ch * CK + kk;
                        if (c >= C) continue;
                        const float s = out_scale ? out_scale[n] : 1.f;
                        g[(size_t)n * Kp + ((ch * KS + ky) * KS + kx) * CK + kk] = W[(((size_t)n * C + c) * KS + ky) * KS + kx] * s;
                    }
    pack_gemm_weight(g.data(), Np, Kp, Kp, Np, Kp, out);
}

float* Net::upload(const std::vector<float>& v) {
    float* d = nullptr;
    if (hipMalloc(&d, v.size() * sizeof(float)) != hipSuccess) throw std::runtime_error("hipMalloc(weights) failed");
    if (hipMemcpy(d, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
        throw std::runtime_error("hipMemcpy(weights) failed");
    owned_.push_back(d);
    return d;
}

#ifndef SUO_WINO_BF16X3_DEFAULT
#define SUO_WINO_BF16X3_DEFAULT 1
#endif
static int round_up(int v, int m) { return (v + m - 1) / m * m; }

// SUO_WINO_BF16X3=0: the Residual blocks' 3x3 convolution + fused tail on the fp32 matrix pipe (csrc/conv_wino.hip) instead of the bf16 pipe
// with 3-way split operands (csrc/conv_wino_x3.hip: same accuracy, ~1.2x faster)
// (read when a network is built, not cached: one process can hold networks of both kinds)
static bool wino_bf16x3() {
    return env_switch("SUO_WINO_BF16X3", SUO_WINO_BF16X3_DEFAULT) != 0;
}
// SUO_F16X2=0: stay on the three-term bf16 form; default: the large launches run the two-term fp16 form (csrc/f16x2.h: half the MFMAs per product, range-guarded --
// a call that leaves fp16's range is reported by Net::range_exceeded and the network falls back to the bf16 form, whose planes are packed as well)
static bool pipe_f16x2() {
    return wino_bf16x3() && env_switch("SUO_F16X2", 1) != 0;
}

// 1x1 conv W[N][K] with optional per-output scale (BN folded) -> device packed weight + bias
void Net::make_gemm(const std::string& conv, const std::string& bn_after, const std::string& conv2, GemmW& g) {
    const HostTensor& w = T(conv + ".weight");
    const HostTensor& b = T(conv + ".bias");
    const int N = (int)w.shape[0], K1 = (int)w.shape[1];
    const int Np = round_up(N, 64), K1p = round_up(K1, 32);
    int K2 = 0, K2p = 0;
    if (!conv2.empty()) { K2 = (int)T(conv2 + ".weight").shape[1]; K2p = round_up(K2, 32); }
    std::vector<float> scale, shift;
    if (!bn_after.empty()) bn_affine(*this, bn_after, scale, shift);
    const int Kp = K1p + K2p;
    std::vector<float> full((size_t)Np * Kp, 0.f), bias(Np, 0.f);
    for (int n = 0; n < N; ++n) {
        const float s = scale.empty() ? 1.f : scale[n];
        for (int k = 0; k < K1; ++k) full[(size_t)n * Kp + k] = w.data[(size_t)n * K1 + k] * s;
        bias[n] = scale.empty() ? b.data[n] : b.data[n] * s + shift[n];
    }
    if (K2) {
        const HostTensor& w2 = T(conv2 + ".weight");
        const HostTensor& b2 = T(conv2 + ".bias");
        for (int n = 0; n < N; ++n) {
            for (int k = 0; k < K2; ++k) full[(size_t)n * Kp + K1p + k] = w2.data[(size_t)n * K2 + k];
            bias[n] += b2.data[n];
        }
    }
    std::vector<float> packed(2 * (size_t)Np * Kp);
    pack_gemm_weight(full.data(), Np, Kp, Kp, Np, Kp, packed.data());
    g.Wp = upload(packed);
    g.bias = upload(bias);
    g.N = Np; g.n_valid = N; g.K1 = K1p; g.K2 = K2p;
    // (N = 64 -- conv1 of r1 / r4 -- stays on the persistent fp32-pipe kernel: these launches are HBM-streaming, 430 / 180 us there against 563 / 194
    //  in the bf16x3 kernel's 64-column tiles at 256 crops)
    if (N % 128 == 0 && Np == N && K1 == K1p && K2 == K2p && K1 % 64 == 0 && K2 % 64 == 0 && K1 <= 512 && wino_bf16x3()) {      // the same operator on the bf16 pipe at fp32 accuracy
        std::vector<float> x3((size_t)3 * N * Kp / 2);                         // uint16 planes
        pack_gemm_weight_bf16x3(full.data(), N, Kp, reinterpret_cast<uint16_t*>(x3.data()));
        g.Wx3 = upload(x3);
        if (pipe_f16x2()) {
            std::vector<float> h2((size_t)N * Kp), osc(N);                    // uint16 planes: 2 * N * Kp
            pack_gemm_weight_f16x2(full.data(), N, Kp, reinterpret_cast<uint16_t*>(h2.data()), osc.data());
            g.W16 = upload(h2);
            g.osc16 = upload(osc);
        }
    } else if (Np == 64 && N == 64 && K1 == 64 && K1p == K1 && K2 == 0 && pipe_f16x2()) {
        // r1's conv1 (64 -> 64): fp16 planes for the stem launch that computes it on the tile (csrc/stem_x3.hip: NEXT)
        std::vector<float> h2((size_t)Np * Kp), osc(Np);
        pack_gemm_weight_f16x2(full.data(), Np, Kp, reinterpret_cast<uint16_t*>(h2.data()), osc.data());
        g.W16 = upload(h2);
        g.osc16 = upload(osc);
    } else if (Np == 64 && K1 == 256 && K1p == K1 && K2 == 0 && pipe_f16x2()) {
        // the output head (tmpOut: 256 -> 41, padded to 64 rows of zeros): fp16 planes for the lin + head launch (csrc/gemm_bf16x3.hip: gemm_chain_head_kernel)
        std::vector<float> h2((size_t)Np * Kp), osc(Np);
        pack_gemm_weight_f16x2(full.data(), Np, Kp, reinterpret_cast<uint16_t*>(h2.data()), osc.data());
        g.W16 = upload(h2);
        g.osc16 = upload(osc);
    }
}

// c_used > 0: keep only the first c_used input channels of the filter (the others multiply structural zeros)
void Net::make_conv(const std::string& conv, const std::string& bn_after, int CK, ConvW& c, int c_used) {
    const HostTensor& w = T(conv + ".weight");
    const HostTensor& b = T(conv + ".bias");
    const int N = (int)w.shape[0], Cw = (int)w.shape[1], KS = (int)w.shape[2];
    const int C = c_used > 0 ? c_used : Cw;
    const int Np = round_up(N, 64), Cp = round_up(C, CK);
    std::vector<float> scale, shift;
    if (!bn_after.empty()) bn_affine(*this, bn_after, scale, shift);
    std::vector<float> packed(2 * (size_t)Np * ((Cp * KS * KS + 15) / 16 * 16)), bias(Np, 0.f);
    std::vector<float> sliced;
    const float* wdata = w.data;
    if (C != Cw) {
        sliced.resize((size_t)N * C * KS * KS);
        for (int n = 0; n < N; ++n)
            memcpy(&sliced[(size_t)n * C * KS * KS], w.data + (size_t)n * Cw * KS * KS, (size_t)C * KS * KS * sizeof(float));
        wdata = sliced.data();
    }
    pack_conv_weight(wdata, N, C, KS, Np, Cp, CK, scale.empty() ? nullptr : scale.data(), packed.data());
    for (int n = 0; n < N; ++n) bias[n] = scale.empty() ? b.data[n] : b.data[n] * scale[n] + shift[n];
    c.Wp = upload(packed);
    c.bias = upload(bias);
    c.N = Np; c.C = Cp; c.KS = KS;
    if (KS == 3 && N == C && (N == 128 || N == 64) && c_used <= 0) {       // the Residual blocks' 128 -> 128 / 64 -> 64 convolutions: also in Winograd form
        std::vector<float> wq((size_t)16 * N * C);
        pack_wino_weight(w.data, N, C, N, C, scale.empty() ? nullptr : scale.data(), wq.data());
        c.Wq = upload(wq);
        if (wino_bf16x3()) {                                  // the same on the bf16 pipe at fp32 accuracy (csrc/conv_wino_x3.hip; 128 -> 128 and 64 -> 64)
            std::vector<float> wq3((size_t)3 * 16 * N * C / 2);                  // uint16 planes
            pack_wino_weight_bf16x3(w.data, N, C, N, C, scale.empty() ? nullptr : scale.data(), reinterpret_cast<uint16_t*>(wq3.data()));
            c.Wq3 = upload(wq3);
            if (pipe_f16x2()) {
                std::vector<float> wq16((size_t)16 * N * C), osc(N);             // uint16 planes: 2 * 16 * N * C
                pack_wino_weight_f16x2(w.data, N, C, N, C, scale.empty() ? nullptr : scale.data(), reinterpret_cast<uint16_t*>(wq16.data()), osc.data());
                c.Wq16 = upload(wq16);
                c.osc16 = upload(osc);
            }
        }
    }
}

void Net::make_residual(const std::string& p, ResidualW& r) {
    {   // Residual(cin, cout): bn[cin] conv1[cout/2,cin,1] bn1 conv2[cout/2,cout/2,3] bn2 conv3[cout,cout/2,1] (+conv4[cout,cin,1] iff cin != cout)
        const int64_t cin = (int64_t)T(p + ".bn.weight").numel, cout = T(p + ".conv3.weight").shape.empty() ? 0 : T(p + ".conv3.weight").shape[0];
        const int64_t h = cout / 2;
        expect_bn(*this, p + ".bn", cin);
        expect_conv(*this, p + ".conv1", h, cin, 1);
        expect_bn(*this, p + ".bn1", h);
        expect_conv(*this, p + ".conv2", h, h, 3);
        expect_bn(*this, p + ".bn2", h);
        expect_conv(*this, p + ".conv3", cout, h, 1);
        if (tensors_.count(p + ".conv4.weight")) expect_conv(*this, p + ".conv4", cout, cin, 1);
        else if (cin != cout) throw std::runtime_error("residual " + p + ": " + std::to_string(cin) + " -> " + std::to_string(cout) + " channels needs conv4");
    }
    std::vector<float> sc, sh;
    bn_affine(*this, p + ".bn", sc, sh);
    r.cin = (int)sc.size();
    r.pro_scale = upload(sc);
    r.pro_shift = upload(sh);
    make_gemm(p + ".conv1", p + ".bn1", "", r.c1);
    make_conv(p + ".conv2", p + ".bn2", 32, r.c2);
    r.has_skip_conv = tensors_.count(p + ".conv4.weight") > 0;
    // conv3 (+ conv4 on the raw input as a second K segment: out = W3*mid + W4*x + b3 + b4)
    make_gemm(p + ".conv3", "", r.has_skip_conv ? p + ".conv4" : "", r.c3);
    r.cout = r.c3.n_valid;
    if (r.cin == 256 && r.cout == 256 && !r.has_skip_conv) {
        // the block in one launch (small maps): bn1 folded into W1's rows and bn2 into W2's exactly as make_gemm / make_conv fold them
        // (float products w * s), so the fp32 form is bit-identical to the per-layer launches
        const HostTensor& w1 = T(p + ".conv1.weight");
        const HostTensor& w2 = T(p + ".conv2.weight");
        const HostTensor& w3 = T(p + ".conv3.weight");
        std::vector<float> s1, t1, s2, t2;
        bn_affine(*this, p + ".bn1", s1, t1);
        bn_affine(*this, p + ".bn2", s2, t2);
        std::vector<float> w1f((size_t)128 * 256);
        for (int n = 0; n < 128; ++n)
            for (int k = 0; k < 256; ++k) w1f[(size_t)n * 256 + k] = w1.data[(size_t)n * 256 + k] * s1[n];
        std::vector<float> p1((size_t)128 * 256), p2((size_t)128 * 128 * 9), p3((size_t)256 * 128);
        pack_res16_gemm(w1f.data(), 128, 256, p1.data());
        pack_res16_conv3x3(w2.data, 128, 128, s2.data(), p2.data());
        pack_res16_gemm(w3.data, 256, 128, p3.data());
        r.rb_w[0] = upload(p1); r.rb_w[1] = upload(p2); r.rb_w[2] = upload(p3);
        if (wino_bf16x3()) {
            std::vector<float> x1((size_t)3 * 128 * 256 / 2), x2((size_t)3 * 128 * 128 * 9 / 2), x3((size_t)3 * 256 * 128 / 2);      // uint16 planes
            pack_gemm_weight_bf16x3(w1f.data(), 128, 256, reinterpret_cast<uint16_t*>(x1.data()));
            pack_res_conv3x3_bf16x3(w2.data, s2.data(), reinterpret_cast<uint16_t*>(x2.data()));
            pack_gemm_weight_bf16x3(w3.data, 256, 128, reinterpret_cast<uint16_t*>(x3.data()));
            r.rbx_w[0] = upload(x1); r.rbx_w[1] = upload(x2); r.rbx_w[2] = upload(x3);
            if (pipe_f16x2()) {
                std::vector<float> h1((size_t)128 * 256), h2((size_t)128 * 128 * 9), h3((size_t)256 * 128), o1(128), o2(128), o3(256);      // uint16 planes: 2 per entry
                pack_gemm_weight_f16x2(w1f.data(), 128, 256, reinterpret_cast<uint16_t*>(h1.data()), o1.data());
                pack_res_conv3x3_f16x2(w2.data, s2.data(), reinterpret_cast<uint16_t*>(h2.data()), o2.data());
                pack_gemm_weight_f16x2(w3.data, 256, 128, reinterpret_cast<uint16_t*>(h3.data()), o3.data());
                r.rbh_w[0] = upload(h1); r.rbh_w[1] = upload(h2); r.rbh_w[2] = upload(h3);
                r.rbh_osc[0] = upload(o1); r.rbh_osc[1] = upload(o2); r.rbh_osc[2] = upload(o3);
            }
        }
    }
    if (r.c2.Wq3 && !r.has_skip_conv) {
        const HostTensor& w3 = T(p + ".conv3.weight");
        if (w3.shape[0] == 256 && w3.shape[1] == 128) {
            std::vector<float> x3((size_t)3 * 256 * 128 / 2);
            pack_tail_weight_bf16x3(w3.data, 256, 128, reinterpret_cast<uint16_t*>(x3.data()));
            r.c3x = upload(x3);
            if (r.c2.Wq16) {
                std::vector<float> h2((size_t)256 * 128), osc(256);
                pack_tail_weight_f16x2(w3.data, 256, 128, reinterpret_cast<uint16_t*>(h2.data()), osc.data());
                r.c3x16 = upload(h2);
                r.c3osc16 = upload(osc);
            }
        }
    }
}

void Net::make_hourglass(const std::string& p, int n, HourglassW& h) {
    h.n = n;
    auto check256 = [&](const ResidualW& r, const std::string& name) {
        if (r.cin != 256 || r.cout != 256) throw std::runtime_error("residual " + name + " must be 256 -> 256 channels");
    };
    for (int j = 0; j < 2; ++j) {
        make_residual(p + ".up1_." + std::to_string(j), h.up1[j]);
        make_residual(p + ".low1_." + std::to_string(j), h.low1[j]);
        make_residual(p + ".low3_." + std::to_string(j), h.low3[j]);
        check256(h.up1[j], p + ".up1_"); check256(h.low1[j], p + ".low1_"); check256(h.low3[j], p + ".low3_");
    }
    if (n > 1) {
        h.inner.reset(new HourglassW());
        make_hourglass(p + ".low2", n - 1, *h.inner);
    } else {
        for (int j = 0; j < 2; ++j) make_residual(p + ".low2_." + std::to_string(j), h.low2[j]);
    }
}

// ------------------------------------------------------------------------------------------------
Net::Net(int n, const char* const* names, const float* const* data, const int64_t* const* shapes, const int* ndims, int max_crops)
    : max_crops_(max_crops) {
    for (int i = 0; i < n; ++i) {
        HostTensor t;
        t.data = data[i];
        t.numel = 1;
        for (int d = 0; d < ndims[i]; ++d) { t.shape.push_back(shapes[i][d]); t.numel *= (size_t)shapes[i][d]; }
        tensors_[names[i]] = t;
    }
    const std::string b = "backbone";
    auto expect_io = [&](const std::string& p, int cin, int cout) {
        expect_shape(*this, p + ".bn.weight", {cin});
        expect_shape(*this, p + ".conv3.bias", {cout});
    };
    expect_conv(*this, b + ".conv1_", 64, 3 + NUM_KP, 7);
    expect_bn(*this, b + ".bn1", 64);
    expect_io(b + ".r1", 64, 128);
    expect_io(b + ".r4", 128, 128);
    expect_io(b + ".r5", 128, 256);
    for (int i = 0; i < 2; ++i) {
        const std::string si = std::to_string(i);
        expect_conv(*this, b + ".lin_." + si + ".0", 256, 256, 1);
        expect_bn(*this, b + ".lin_." + si + ".1", 256);
        expect_conv(*this, b + ".tmpOut." + si, NUM_KP, 256, 1);
    }
    expect_conv(*this, b + ".ll_.0", 256, 256, 1);
    expect_conv(*this, b + ".tmpOut_.0", 256, NUM_KP, 1);
    expect_shape(*this, "classifier.2.weight", {NUM_KP, NUM_KP});
    expect_shape(*this, "classifier.2.bias", {NUM_KP});
    make_conv(b + ".conv1_", b + ".bn1", 16, stem_);
    // Without priors (every single-view pass and the first SLAM pass, lib/object_slam.py:1094-1097) the 41 prior
    // channels are zeros: multiply only the 3 image channels.  Same taps, same order, so the result is bit-identical
    // to feeding zero priors through the full filter; 13 % of the network's MACs (41/44 of the stem) are never issued.
    make_conv(b + ".conv1_", b + ".bn1", IMG_C, stem_img_, 3);
    if (wino_bf16x3()) {                                      // ... and for the fused RoIAlign + stem launch on the bf16 pipe (csrc/stem_x3.hip)
        const HostTensor& w = T(b + ".conv1_.weight");
        const HostTensor& bb = T(b + ".conv1_.bias");
        std::vector<float> sc, sh;
        bn_affine(*this, b + ".bn1", sc, sh);
        std::vector<float> wx((size_t)14 * 2 * 3 * 64 * 8 / 2), bias(64);        // uint16 planes
        pack_stem_weight_bf16x3(w.data, (int)w.shape[1], sc.data(), reinterpret_cast<uint16_t*>(wx.data()));
        for (int n = 0; n < 64; ++n) bias[n] = bb.data[n] * sc[n] + sh[n];
        stem_x3_w_ = upload(wx);
        stem_x3_bias_ = upload(bias);
        if (pipe_f16x2()) {
            std::vector<float> wh((size_t)14 * 2 * 2 * 64 * 8 / 2), osc(64);    // uint16 planes
            pack_stem_weight_f16x2(w.data, (int)w.shape[1], sc.data(), reinterpret_cast<uint16_t*>(wh.data()), osc.data());
            stem_h2_w_ = upload(wh);
            stem_h2_osc_ = upload(osc);
        }
    }
    make_residual(b + ".r1", r1_);
    make_residual(b + ".r4", r4_);
    make_residual(b + ".r5", r5_);
    for (int i = 0; i < 2; ++i) {
        make_hourglass(b + ".hourglass." + std::to_string(i), 4, hg_[i]);
        for (int j = 0; j < 2; ++j) make_residual(b + ".Residual." + std::to_string(i * 2 + j), post_[i][j]);
        make_gemm(b + ".lin_." + std::to_string(i) + ".0", b + ".lin_." + std::to_string(i) + ".1", "", lin_[i]);
        make_gemm(b + ".tmpOut." + std::to_string(i), "", "", head_[i]);
    }
    // inter-stack re-injection (hg.py:112-117): x + ll_(ll) + tmpOut_(tmpOut(ll)).  tmpOut and tmpOut_ are both plain 1x1
    // convolutions with nothing between them, so the sum is ONE 256 -> 256 GEMM on ll with
    //     W' = W_ll + W_tmpOut_ W_tmpOut ,   b' = b_ll + b_tmpOut_ + W_tmpOut_ b_tmpOut      (folded here in fp64, rounded once)
    // + the residual x: the 41-channel stack-0 heat-maps -- which nothing else reads (only the last stack is returned,
    // pkpnet.py:103-105) -- are never materialised and the 64 extra K columns of the dual-operand form are not multiplied.
    {
        const HostTensor& wl = T(b + ".ll_.0.weight");   const HostTensor& bl = T(b + ".ll_.0.bias");
        const HostTensor& wt_ = T(b + ".tmpOut_.0.weight"); const HostTensor& bt_ = T(b + ".tmpOut_.0.bias");
        const HostTensor& wh = T(b + ".tmpOut.0.weight");  const HostTensor& bh = T(b + ".tmpOut.0.bias");
        const int N = (int)wl.shape[0], K = (int)wl.shape[1], J = (int)wh.shape[0];      // 256, 256, 41
        if ((int)wt_.shape[0] != N || (int)wt_.shape[1] != J || (int)wh.shape[1] != K) throw std::runtime_error("re-injection convolutions have unexpected shapes");
        std::vector<float> full((size_t)N * K), bias(N);
        for (int n = 0; n < N; ++n) {
            double bb = (double)bl.data[n] + (double)bt_.data[n];
            for (int j = 0; j < J; ++j) bb += (double)wt_.data[(size_t)n * J + j] * (double)bh.data[j];
            bias[n] = (float)bb;
            for (int k = 0; k < K; ++k) {
                double v = wl.data[(size_t)n * K + k];
                for (int j = 0; j < J; ++j) v += (double)wt_.data[(size_t)n * J + j] * (double)wh.data[(size_t)j * K + k];
                full[(size_t)n * K + k] = (float)v;
            }
        }
        std::vector<float> packed(2 * (size_t)N * K);
        pack_gemm_weight(full.data(), N, K, K, N, K, packed.data());
        reinject_.Wp = upload(packed);
        reinject_.bias = upload(bias);
        reinject_.N = N; reinject_.n_valid = N; reinject_.K1 = K; reinject_.K2 = 0;
        if (N % 128 == 0 && K % 64 == 0 && K <= 512 && wino_bf16x3()) {
            std::vector<float> x3((size_t)3 * N * K / 2);                     // uint16 planes
            pack_gemm_weight_bf16x3(full.data(), N, K, reinterpret_cast<uint16_t*>(x3.data()));
            reinject_.Wx3 = upload(x3);
            if (pipe_f16x2()) {
                std::vector<float> h2((size_t)N * K), osc(N);
                pack_gemm_weight_f16x2(full.data(), N, K, reinterpret_cast<uint16_t*>(h2.data()), osc.data());
                reinject_.W16 = upload(h2);
                reinject_.osc16 = upload(osc);
            }
        }
    }
    {
        const HostTensor& w = T("classifier.2.weight");
        const HostTensor& bb = T("classifier.2.bias");
        cls_w_ = upload(std::vector<float>(w.data, w.data + w.numel));
        cls_b_ = upload(std::vector<float>(bb.data, bb.data + bb.numel));
    }
    tensors_.clear();   // host pointers are not retained past construction
    pipe_built_ = pipe_f16x2() ? 2 : (wino_bf16x3() ? 1 : 0);
    pipe_ = pipe_built_;
    // the range guard's flag: host memory mapped into the device's address space -- the kernels store to it (rarely: only beyond fp16's range), the host
    // reads a plain word after whatever synchronisation its results needed anyway
    if (hipHostMalloc(reinterpret_cast<void**>(&range_flag_), 64, hipHostMallocMapped) != hipSuccess) throw std::runtime_error("hipHostMalloc(range flag) failed");
    *range_flag_ = 0;

    // ---- workspace: every intermediate gets its own slab (288 GB of HBM: no aliasing games).
    // Size it with a dry run of the launch schedule at max_crops.
    dry_run_ = true;
    ws_floats_ = (size_t)-1 / sizeof(float) / 2;
    ws_used_ = 0;
    (void)alloc((size_t)max_crops_ * CROP * CROP * IN_C);
    (void)alloc((size_t)max_crops_ * NUM_KP * HEAT * HEAT);
    stem_slab_ = alloc((size_t)max_crops_ * 128 * 128 * 64);
    stem_mid1_slab_ = alloc((size_t)max_crops_ * 128 * 128 * 64);
    ws_mark_ = ws_used_;
    // (every pipe this network can run on: the forms pick different kernels -- per-layer launches with their intermediate tensors where another form
    //  takes a one-launch block -- and a network that falls back to bf16x3 must find its workspace large enough)
    size_t need = 0;
    for (int p = pipe_built_; p >= (pipe_built_ == 0 ? 0 : 1); --p) {
        pipe_ = p;
        if (backbone(nullptr, IN_C, nullptr, max_crops_, nullptr) != SUO_OK) throw std::runtime_error("dry run failed");
        need = std::max(need, ws_used_);
    }
    pipe_ = pipe_built_;
    ws_floats_ = need;
    dry_run_ = false;
    if (hipMalloc(&ws_, ws_floats_ * sizeof(float)) != hipSuccess) throw std::runtime_error("hipMalloc(workspace) failed");
    if (hipMalloc(&d_mean_logit_, (size_t)max_crops_ * NUM_KP * sizeof(float)) != hipSuccess)
        throw std::runtime_error("hipMalloc failed");
    if (hipStreamCreateWithFlags(&own_stream_, hipStreamNonBlocking) != hipSuccess) throw std::runtime_error("hipStreamCreate failed");
    for (int i = 0; i < kNumSide; ++i) {
        if (hipStreamCreateWithFlags(&side_[i], hipStreamNonBlocking) != hipSuccess) throw std::runtime_error("hipStreamCreate failed");
    }
    for (int i = 0; i < kNumEvents; ++i)
        if (hipEventCreateWithFlags(&ev_[i], hipEventDisableTiming) != hipSuccess) throw std::runtime_error("hipEventCreate failed");
}

Net::~Net() {
    for (auto& kv : graphs_) { for (int i = 0; i < 2; ++i) if (kv.second.exec[i]) (void)hipGraphExecDestroy(kv.second.exec[i]); (void)hipGraphDestroy(kv.second.graph); }
    for (float* p : owned_) (void)hipFree(p);
    if (ws_) (void)hipFree(ws_);
    if (d_mean_logit_) (void)hipFree(d_mean_logit_);
    if (range_flag_) (void)hipHostFree(range_flag_);
    for (int i = 0; i < kNumSide; ++i) if (side_[i]) (void)hipStreamDestroy(side_[i]);
    if (own_stream_) (void)hipStreamDestroy(own_stream_);
    for (int i = 0; i < kNumEvents; ++i) if (ev_[i]) (void)hipEventDestroy(ev_[i]);
}

float* Net::alloc(size_t floats) {
    floats = (floats + 63) & ~(size_t)63;
    if (ws_used_ + floats > ws_floats_) throw std::runtime_error("workspace exhausted");
    float* p = ws_ + ws_used_;
    ws_used_ += floats;
    return p;
}

#define SUO_TRY(x) do { int _r = (x); if (_r != SUO_OK) return _r; } while (0)
#define SUO_LAUNCH(x) do { if (acct_on_) ++acct_launches_; if (!dry_run_) { int _r = (x); if (_r != SUO_OK) return _r; } } while (0)
#define SUO_HIP_LIVE(x) do { if (!dry_run_) SUO_HIP_CHECK(x); } while (0)

// ---- algorithmic (compulsory) HBM bytes of a launch: every operand read once, every result written once, the weights once.  Summed per kind over the launch
// schedule by Net::schedule_bytes (a dry run): what bench.py's `roofline_all.whole_call` divides by the step time.  wb = bytes per weight element in the form the
// launch reads (4: fp32 or two fp16 planes; 6: three bf16 planes).
enum { ACCT_STAGE = 0, ACCT_CONV3 = 1, ACCT_GEMM = 2, ACCT_BLOCK = 3, ACCT_ELTWISE = 4, ACCT_DECODE = 5, ACCT_KINDS = 6 };
static double gemm_bytes(const GemmArgs& g, double wb) {
    const double M = g.M, nv = g.nchw_hw > 0 ? g.n_valid : g.N;
    double b = 4.0 * M * (g.K1 + g.K2) + (g.R ? 4.0 * M * g.N : 0.0) + wb * (double)g.N * (g.K1 + g.K2);
    if (g.out) b += 4.0 * M * nv;
    if (g.pool_out) b += 4.0 * (M / 4) * g.N;
    return b;
}
static double conv_bytes(const ConvArgs& c, double wb, int taps, bool fused) {
    const double pin = (double)c.L * c.H * c.W, pout = (double)c.L * c.OH * c.OW;
    double b = 4.0 * pin * c.C + wb * (double)c.N * c.C * taps;
    if (!fused) return b + 4.0 * pout * c.N;
    b += 4.0 * pout * c.N2 * 2 + wb * (double)c.N2 * c.N;                   // skip read, out2 written, conv3's weights
    if (c.up) b += 4.0 * (pout / 4) * c.N2;
    if (c.n_out) b += 4.0 * pout * 128 + wb * 128.0 * 256.0;               // the next block's conv1 written, its weights
    return b;
}
static double block_bytes(const ResBlockArgs& a, double wb) {
    const double px = (double)a.L * a.H * a.W;
    return 4.0 * px * 256 * (a.pool_in ? 4 : 1) + 4.0 * px * 256 + (a.up ? 4.0 * (px / 4) * 256 : 0.0) + wb * (256.0 * 128 + 128.0 * 128 * 9 + 128.0 * 256);
}

// Residual.forward: three fused launches
// true when residual() ends in the fused Winograd tail (the only epilogue that can add an up-sampled tensor)
// Launch-size thresholds of the split-operand kernels.  The Winograd kernels and the bf16x3 GEMM were introduced for batched calls and measured against the fp32-pipe
// kernels' smaller tiles: from 256 tiles of 8 x 16 pixels / 32768 rows up.  The fp16 forms changed the balance for calls of FEW crops (SLAM passes run 2-7): a fused fp16
// tail takes ~50 us whatever the crop count below 8 (one workgroup per CU) where direct 3x3 + conv3 take 84-106; measured per network call of L crops (bench.py --only cnn
// --objects L --frames-per-step 1, ms, thresholds 256 / 32768 -> 32 / 4096): L = 1 1.544 -> 1.531, 2 1.631 -> 1.567, 3 1.886 -> 1.678, 4 1.921 -> 1.726, 5 2.155 -> 1.759,
// 6 2.218 -> 1.800, 7 2.435 -> 1.832, 8 and up unchanged.
long Net::wino_min_tiles() const {
    static const long env = (long)SUO_TUNE("SUO_WINO_FUSE_TILES", -1);      // (0: never fuse)
    return env >= 0 ? env : (pipe_ == 2 ? 32 : 256);
}
long Net::x3_min_rows() const {
    static const long env = (long)SUO_TUNE("SUO_GEMM_X3_MIN_ROWS", -1);
    return env >= 0 ? env : (pipe_ == 2 ? 4096 : 32768);
}

bool Net::residual_tail_is_fused(const ResidualW& r, int L, int H, int W) const {
    const long fuse_tiles = wino_min_tiles();
    ConvArgs c2 = {};
    c2.L = L; c2.H = H; c2.W = W; c2.C = r.c2.C; c2.OH = H; c2.OW = W; c2.N = r.c2.N;
    const long tiles = (long)((W + 15) / 16) * ((H + 7) / 8) * L;
    return r.c2.Wq && conv3x3_wino_pays(c2, pipe_ == 2 ? 32 : -1) && fuse_tiles > 0 && tiles >= fuse_tiles && !r.has_skip_conv && r.c3.N == 256 && r.c3.n_valid == 256 &&
           r.c3.K1 == 128 && r.cin == 256;
}

// A 256 -> 256 block on a small map in ONE launch (csrc/res_small_x3.hip / csrc/res_small.hip) instead of three: the call shape of the
// reference (one frame = 8 crops per call, lib/object_slam.py:1099).  Measured per block at 8 crops (tools/bench_res_block.py, us; per-layer
// launches -> one launch on the bf16 pipe): 32x32 47 -> 29.6, 16x16 26.6 -> 25.0; at 8x8 and 4x4 the three per-layer launches (13.5) stay
// faster -- a workgroup streams all 0.85 / 1.28 MB of the block's weights through ONE CU whatever its tile, 20-24 us at 35 bytes per clock.
//   SUO_RES_FUSED = 0: never; 1: the fp32-pipe kernel (bit-identical to the per-layer launches); 2 (default with SUO_WINO_BF16X3): the bf16x3 kernel
//   SUO_RES_FUSED_MAX_TILES: largest launch (4 x 8 pixel tiles) that takes it; beyond that the Winograd kernels' larger tiles win
int Net::residual_in_one_launch(const ResidualW& r, int L, int H, int W) const {
    static const int mode = (int)SUO_TUNE("SUO_RES_FUSED", 2);
    static const long max_tiles = (long)SUO_TUNE("SUO_RES_FUSED_MAX_TILES", 768);
    static const int min_side = (int)SUO_TUNE("SUO_RES_FUSED_MIN_SIDE", 16);
    if (mode <= 0 || !r.rb_w[0] || H > 32 || W > 32) return 0;
    const long t32 = (long)L * ((H + 3) / 4) * ((W + 7) / 8);
    // Half a round to three rounds of 4 x 8 pixel tiles (one workgroup per CU): the bf16x3 kernel, whatever the map (one frame at 32x32; batched frames at 8x8
    // and 4x4, where it replaces three launches of 17-47 us by one or two rounds of 30).  Fewer tiles than CUs on a map of >= 16 pixels a
    // side: the fp32 kernel's 4 x 4 tiles (16x16 at 8 crops: 128 workgroups x 23.4 us against 64 x 28).  Everything else -- few tiles on
    // 8x8 / 4x4 maps (the three per-layer launches spread over all CUs, 13.5 us), many tiles (the Winograd kernels) -- stays per-layer.
    // (129 ... 255 tiles, e.g. the 5 crops of a SLAM pass at 32x32: one partial round of the bf16x3 kernel, 30 us, against 4 x 4 tiles
    //  that no longer fit one per CU, ~40)
    static const long x3_from_env = (long)SUO_TUNE("SUO_RES_FUSED_X3_FROM", 129);
    // the fp16 form of the 4 x 8-tile kernel streams a third less weight per workgroup (csrc/res_small_x3.hip, NP = 2): 16.8 us at 64 tiles (16x16, 8 crops) where the
    // fp32 kernel's 128 tiles of 4 x 4 take 23.4 and the bf16x3 form 23.1 -- it takes over from 33 tiles (tools/bench_res_block.py; 8x8 at 8 crops = 16 tiles: 16.6
    // against 13.4 for the three per-layer launches, which stay)
    static const long f16_from = (long)SUO_TUNE("SUO_RES_FUSED_F16_FROM", 33);
    const long x3_from = (pipe_ == 2 && r.rbh_w[0] && mode >= 2) ? std::min(f16_from, x3_from_env) : x3_from_env;
    if (mode >= 2 && r.rbx_w[0] && t32 >= x3_from && t32 <= max_tiles) return 2;
    if (H >= min_side && W >= min_side && (t32 < x3_from || (mode == 1 && t32 <= max_tiles))) return 1;
    return 0;
}

int Net::residual_one_launch(const ResidualW& r, const float* x, float* out, int L, int H, int W, hipStream_t s, const float* up, bool pool_in) {
    const int kind = residual_in_one_launch(r, L, H, W);
    ResBlockArgs a = {};
    a.x = x; a.L = L; a.H = H; a.W = W; a.pool_in = pool_in ? 1 : 0; a.pro_scale = r.pro_scale; a.pro_shift = r.pro_shift;
    a.b1 = r.c1.bias; a.b2 = r.c2.bias; a.b3 = r.c3.bias; a.up = up; a.out = out;
    if (kind == 2 && pipe_ == 2 && r.rbh_w[0]) {                      // two fp16 planes: a third less weight traffic per workgroup, half the MFMAs
        a.W1 = r.rbh_w[0]; a.W2 = r.rbh_w[1]; a.W3 = r.rbh_w[2];
        a.osc1 = r.rbh_osc[0]; a.osc2 = r.rbh_osc[1]; a.osc3 = r.rbh_osc[2]; a.range_flag = range_flag_;
        acct(ACCT_BLOCK, block_bytes(a, 4));
        SUO_LAUNCH(launch_res_block_f16x2(a, s));
    } else if (kind == 2) {
        a.W1 = r.rbx_w[0]; a.W2 = r.rbx_w[1]; a.W3 = r.rbx_w[2];
        acct(ACCT_BLOCK, block_bytes(a, 6));
        SUO_LAUNCH(launch_res_block_x3(a, s));
    } else {
        a.W1 = r.rb_w[0]; a.W2 = r.rb_w[1]; a.W3 = r.rb_w[2];
        acct(ACCT_BLOCK, block_bytes(a, 4));
        SUO_LAUNCH(launch_res_block(a, s));
    }
    return SUO_OK;
}

int Net::maxpool(const float* in, float* out, int L, int H, int W, int C, hipStream_t s) {
    acct(ACCT_ELTWISE, 4.0 * L * H * W * C * 1.25);
    SUO_LAUNCH(launch_maxpool2(in, out, L, H, W, C, s));
    return SUO_OK;
}

// A 1x1 convolution whose result is also wanted max-pooled (nn.MaxPool2d(2, 2)): pooled in the GEMM's epilogue when the launch
// would use the persistent 128x128 kernel anyway (csrc/gemm_persist.hip: POOL), else GEMM + max-pool kernel.  g.out may be nullptr
// when only the pooled tensor is wanted.
int Net::gemm_maybe_pooled(GemmArgs& g, int L, int H, int W, float* pool_out, hipStream_t s, const GemmW* gw) {
    const float* Wx3 = gw ? gw->Wx3 : nullptr;
    static const int fuse_pool = (int)suo::env_switch("SUO_FUSE_POOL", 1);                    // 0: A/B
    // large launches with a bf16x3 form of the weights: on the bf16 pipe (csrc/gemm_bf16x3.hip)
    const long x3_min_rows = this->x3_min_rows();
    if (Wx3 && g.M >= x3_min_rows) {
        GemmArgs gx = g;
        gx.pool_out = pool_out; gx.pool_H = H; gx.pool_W = W;                  // the pool in the epilogue (maps of 64-column multiples), `out` optional
        const bool f16 = pipe_ == 2 && gw->W16;                              // the two-term fp16 form of the same kernel (csrc/f16x2.h)
        if (f16) { gx.oscale = gw->osc16; gx.range_flag = range_flag_; }
        auto launch = [&](const GemmArgs& a) { return f16 ? launch_gemm_f16x2_args(a, reinterpret_cast<const uint16_t*>(gw->W16), s) : launch_gemm_bf16x3_args(a, reinterpret_cast<const uint16_t*>(Wx3), s); };
        if (gemm_bf16x3_takes(gx)) { acct(ACCT_GEMM, gemm_bytes(gx, f16 ? 4 : 6)); SUO_LAUNCH(launch(gx)); return SUO_OK; }
        gx.pool_out = nullptr;                                                // else the pool as its own launch
        if (g.out && gemm_bf16x3_takes(gx)) {
            acct(ACCT_GEMM, gemm_bytes(gx, f16 ? 4 : 6));
            SUO_LAUNCH(launch(gx));
            if (pool_out) SUO_TRY(maxpool(g.out, pool_out, L, H, W, g.N, s));
            return SUO_OK;
        }
    }
    if (!pool_out) { acct(ACCT_GEMM, gemm_bytes(g, 4)); SUO_LAUNCH(launch_gemm1x1(g, s)); return SUO_OK; }
    GemmArgs gp = g;
    gp.pool_out = pool_out; gp.pool_H = H; gp.pool_W = W;
    const long tiles128 = (long)(g.M / 128) * (g.N / 128);
    if (fuse_pool && g.M > 4096 && tiles128 >= 512 && gemm1x1_can_pool(gp)) { acct(ACCT_GEMM, gemm_bytes(gp, 4)); SUO_LAUNCH(launch_gemm1x1(gp, s)); return SUO_OK; }
    if (!g.out) g.out = alloc((size_t)g.M * g.ldo);
    acct(ACCT_GEMM, gemm_bytes(g, 4));
    SUO_LAUNCH(launch_gemm1x1(g, s));
    return maxpool(g.out, pool_out, L, H, W, g.N, s);
}

// The next block's conv1 can ride on this block's fused fp16 tail when the separate launch would have been the fp16 GEMM on the same operands
// (256 -> 128 with a BatchNorm prologue, >= SUO_GEMM_X3_MIN_ROWS pixels, not a one-launch block): then the two are bit-identical (tests/test_gpu_f16x2.py)
bool Net::next_conv1_fusable(const ResidualW& next, int L, int H, int W) const {
    static const int on = (int)SUO_TUNE("SUO_FUSE_NEXT_CONV1", 1);                // 0: A/B
    const long x3_min_rows = this->x3_min_rows();
    return on && pipe_ == 2 && next.cin == 256 && next.c1.W16 && next.c1.osc16 && next.c1.N == 128 && next.c1.n_valid == 128 && next.c1.K1 == 256 && next.c1.K2 == 0 &&
           (long)L * H * W >= x3_min_rows && !residual_in_one_launch(next, L, H, W) &&
           !conv3x3_wino_f16x2_w8((long)((W + 15) / 16) * ((H + 7) / 8) * L);                     // (the eight-wave form of small launches does not carry it)
}

int Net::residual(const ResidualW& r, const float* x, float* out, int L, int H, int W, hipStream_t s, const float* up, float* pool_out, const ResidualW* next) {
    const int M = L * H * W;
    if (residual_in_one_launch(r, L, H, W)) {
        if (!out) out = alloc((size_t)M * 256);
        SUO_TRY(residual_one_launch(r, x, out, L, H, W, s, up, false));
        if (pool_out) SUO_TRY(maxpool(out, pool_out, L, H, W, 256, s));
        return SUO_OK;
    }
    float* mid1 = nullptr;
    for (size_t i = 0; i < pre_.size(); ++i)
        if (pre_[i].x == x && pre_[i].r == &r) {              // conv1 came with the producer block's tail (NEXT)
            mid1 = pre_[i].mid1;
            pre_.erase(pre_.begin() + i);
            break;
        }
    if (!mid1) {
        mid1 = alloc((size_t)M * r.c1.N);
        GemmArgs g1 = {};
        g1.A1 = x; g1.lda1 = r.cin; g1.K1 = r.c1.K1; g1.pro_scale = r.pro_scale; g1.pro_shift = r.pro_shift;
        g1.Wp = r.c1.Wp; g1.bias = r.c1.bias; g1.out = mid1; g1.ldo = r.c1.N; g1.M = M; g1.N = r.c1.N; g1.n_valid = r.c1.n_valid; g1.relu = 1;
        // large launches: on the bf16 pipe with 3-way split operands (464 vs 595 us at 256 crops / 64 x 64; below ~256 tiles the fp32 kernels' smaller tiles win)
        SUO_TRY(gemm_maybe_pooled(g1, L, H, W, nullptr, s, &r.c1));
    }
    float* mid2 = alloc((size_t)M * r.c2.N);
    ConvArgs c2 = {};
    c2.in = mid1; c2.L = L; c2.H = H; c2.W = W; c2.C = r.c2.C; c2.Wp = r.c2.Wp; c2.bias = r.c2.bias;
    c2.out = mid2; c2.OH = H; c2.OW = W; c2.N = r.c2.N; c2.relu = 1;
    const bool wino = r.c2.Wq && conv3x3_wino_pays(c2, pipe_ == 2 ? 32 : -1);       // 2.25x fewer MFMA MACs (csrc/conv_wino.hip)
    if (!wino && !r.has_skip_conv && r.c3.N == 256 && r.c3.n_valid == 256 && r.c3.K1 == 128 && r.cin == 256 && conv3x3_fusable(c2)) {
        // conv2 -> conv3 + skip in one launch: the 128-channel tensor between them never leaves the CU (csrc/conv.hip: FUSE)
        if (!out) out = alloc((size_t)M * 256);
        c2.W3p = r.c3.Wp; c2.bias3 = r.c3.bias; c2.R = x; c2.out2 = out; c2.N2 = 256;
        acct(ACCT_CONV3, conv_bytes(c2, 4, 9, true));
        SUO_LAUNCH(launch_conv3x3_fused(c2, s));
        if (pool_out) SUO_TRY(maxpool(out, pool_out, L, H, W, 256, s));
        return SUO_OK;
    }
    if (wino) {
        c2.Wp = r.c2.Wq;
        const long fuse_tiles = wino_min_tiles();
        const long tiles = (long)((W + 15) / 16) * ((H + 7) / 8) * L;
        if (fuse_tiles > 0 && tiles >= fuse_tiles && !r.has_skip_conv && r.c3.N == 256 && r.c3.n_valid == 256 && r.c3.K1 == 128 && r.cin == 256) {
            // conv2 -> conv3 + skip in one launch (933 vs 713 + 346 us at 64x64 / 128 crops, 257 vs 195 + 91 at 32x32)
            if (!out) out = alloc((size_t)M * 256);
            c2.W3p = r.c3.Wp; c2.bias3 = r.c3.bias; c2.R = x; c2.out2 = out; c2.N2 = 256; c2.up = up;
            if (pipe_ == 2 && r.c2.Wq16 && r.c3x16) {        // both products as two fp16 terms (csrc/f16x2.h)
                c2.Wp = r.c2.Wq16; c2.W3p = r.c3x16; c2.oscale = r.c2.osc16; c2.oscale3 = r.c3osc16; c2.range_flag = range_flag_;
                // the next block's conv1 on the tile while it is in the CU: its 256-channel input is written once and not re-read by a GEMM launch.
                // (Not with an up-sampled addend: that variant has no registers left -- measured no gain, tools/bench_f16x2.py.)
                if (next && !up && !pool_out && next_conv1_fusable(*next, L, H, W)) {
                    float* nm = alloc((size_t)M * 128);
                    c2.n_scale = next->pro_scale; c2.n_shift = next->pro_shift; c2.n_W1 = next->c1.W16; c2.n_osc1 = next->c1.osc16; c2.n_b1 = next->c1.bias; c2.n_out = nm;
                    pre_.push_back({out, next, nm});
                }
                acct(ACCT_CONV3, conv_bytes(c2, 4, 16, true));
                SUO_LAUNCH(launch_conv3x3_wino_f16x2_fused(c2, s));
            } else if (r.c2.Wq3 && r.c3x) {                   // both products on the bf16 pipe, 3-way split operands
                c2.Wp = r.c2.Wq3; c2.W3p = r.c3x; c2.w3_bf16x3 = 1;
                acct(ACCT_CONV3, conv_bytes(c2, 6, 16, true));
                SUO_LAUNCH(launch_conv3x3_wino_x3_fused(c2, s));
            } else {
                acct(ACCT_CONV3, conv_bytes(c2, 4, 16, true));
                SUO_LAUNCH(launch_conv3x3_wino_fused(c2, s));
            }
            if (pool_out) SUO_TRY(maxpool(out, pool_out, L, H, W, 256, s));
            return SUO_OK;
        }
        if (pipe_ == 2 && r.c2.Wq16) { c2.Wp = r.c2.Wq16; c2.oscale = r.c2.osc16; c2.range_flag = range_flag_; acct(ACCT_CONV3, conv_bytes(c2, 4, 16, false)); SUO_LAUNCH(launch_conv3x3_wino_f16x2(c2, s)); }
        else if (r.c2.Wq3) { c2.Wp = r.c2.Wq3; acct(ACCT_CONV3, conv_bytes(c2, 6, 16, false)); SUO_LAUNCH(launch_conv3x3_wino_x3(c2, s)); }
        else { acct(ACCT_CONV3, conv_bytes(c2, 4, 16, false)); SUO_LAUNCH(launch_conv3x3_wino(c2, s)); }
    } else {
        acct(ACCT_CONV3, conv_bytes(c2, 4, 9, false));
        SUO_LAUNCH(launch_conv3x3(c2, s));
    }
    if (up) { suo_set_error("residual: an up-sampled addend needs the fused Winograd tail"); return SUO_ERR_ARG; }
    GemmArgs g3 = {};
    g3.A1 = mid2; g3.lda1 = r.c2.N; g3.K1 = r.c3.K1;
    if (r.has_skip_conv) { g3.A2 = x; g3.lda2 = r.cin; g3.K2 = r.c3.K2; }
    else { g3.R = x; g3.ldr = r.cin; }
    g3.Wp = r.c3.Wp; g3.bias = r.c3.bias; g3.out = out; g3.ldo = r.cout; g3.M = M; g3.N = r.c3.N; g3.n_valid = r.c3.n_valid;
    return gemm_maybe_pooled(g3, L, H, W, pool_out, s, &r.c3);
}

// Hourglass.forward (hg.py:37-58).  The up1 branch is independent of the low branch until the
// final add: it can run on a side stream (fork/join with events) so the small, latency-bound low
// levels overlap with the large up1 kernels -- see n_side below for when that pays.
int Net::hourglass(const HourglassW& h, const float* x, float* out, int L, int H, int W, hipStream_t s, int depth_idx, const float* x_pooled) {
    const int C = 256;
    const size_t n_hi = (size_t)L * H * W * C, n_lo = n_hi / 4;
    // Side streams for the up1 branch are OFF by default (SUO_NET_SIDE_STREAMS=1|2 turns them on).  Measured on MI355X with the
    // harness's two network calls in flight: one frame (8 crops) per call 351 frames/s with two side streams, 415 with one, 426
    // with none; 32 frames per call 738 / 745 / 749.  The second call in flight already fills the gaps the fork was meant to fill,
    // and every extra stream competes for the 4 hardware queues with the other network and the geometry stream.  With a single
    // call in flight the fork is worth about 1 % (385 vs 381 frames/s at 8 crops).
    static const int n_side = std::max(0, std::min(kNumSide, (int)env_switch("SUO_NET_SIDE_STREAMS", 0)));
    // ... except on the SMALL maps of a call of few crops (the one-frame call: 16x16 and 8x8 at 8 crops): there every kernel is a handful of workgroups and a
    // dependent launch costs its latency, not its work -- the up1 blocks (16.8 / 2 x 13.4 us) run beside the low branch instead of in front of it.
    // SUO_NET_FORK_SMALL_PIXELS: largest L * H * W that forks (2048 = 16x16 at 8 crops); default 0 = never: MEASURED SLOWER -- one frame per call 2.010 ms with
    // the fork against 1.897 without (same box, tools/time_frame_chain.py): the captured graph's extra branch costs more than the two or three launches it hides.
    static const long fork_small = (long)SUO_TUNE("SUO_NET_FORK_SMALL_PIXELS", 0);
    const bool small_fork = !env_set("SUO_SERIAL") && fork_small > 0 && (long)L * H * W <= fork_small && H >= 8;
    const bool serial = (env_set("SUO_SERIAL") || n_side == 0) && !small_fork;     // one stream, kernels back to back
    hipStream_t side = serial ? s : side_[depth_idx % (n_side > 0 ? n_side : kNumSide)];
    hipEvent_t ev_fork = ev_[(ev_next_++) % (kNumEvents - 1)], ev_join = ev_[(ev_next_++) % (kNumEvents - 1)];      // (the last event is follow_null_stream's)
    float* up_a = alloc(n_hi);
    float* up_b = alloc(n_hi);
    SUO_HIP_LIVE(hipEventRecord(ev_fork, s));
    SUO_HIP_LIVE(hipStreamWaitEvent(side, ev_fork, 0));
    // "up1 + up2(low3)" (hg.py:56-58): when the last up1 block ends in the fused Winograd tail, that tail adds the up-sampled low
    // branch itself and writes `out` -- no up-sample kernel, no extra pass over the high-resolution tensor.  The block then has to
    // wait for the low branch; its predecessor still runs beside it on the side stream.
    static const int fuse_up = (int)suo::env_switch("SUO_FUSE_UPSAMPLE", 1);              // 0: A/B
    const bool up_in_tail = fuse_up && (residual_tail_is_fused(h.up1[1], L, H, W) || residual_in_one_launch(h.up1[1], L, H, W));
    SUO_TRY(residual(h.up1[0], x, up_a, L, H, W, side, nullptr, nullptr, &h.up1[1]));
    if (!up_in_tail) SUO_TRY(residual(h.up1[1], up_a, up_b, L, H, W, side));
    SUO_HIP_LIVE(hipEventRecord(ev_join, side));

    const float* pooled = x_pooled;                           // (the caller's producer kernel may have pooled x already)
    // max_pool2d(x) (hg.py:41) has ONE reader, the first low block: when that block runs in one launch it takes the pool while staging x
    static const int pool_in_block = (int)SUO_TUNE("SUO_RES_POOL_IN", 1);           // 0: A/B
    const bool pool_by_block = !pooled && pool_in_block && residual_in_one_launch(h.low1[0], L, H / 2, W / 2) && (H % 2 == 0) && (W % 2 == 0);
    if (!pooled && !pool_by_block) {
        float* p = alloc(n_lo);
        SUO_TRY(maxpool(x, p, L, H, W, C, s));
        pooled = p;
    }
    float* lo_a = alloc(n_lo);
    float* lo_b = alloc(n_lo);
    if (pool_by_block) SUO_TRY(residual_one_launch(h.low1[0], x, lo_a, L, H / 2, W / 2, s, nullptr, true));
    else SUO_TRY(residual(h.low1[0], pooled, lo_a, L, H / 2, W / 2, s, nullptr, nullptr, &h.low1[1]));
    // (lo_b has two readers -- the inner hourglass's up1[0] and, pooled, its low1[0]: the first one's conv1 rides along)
    SUO_TRY(residual(h.low1[1], lo_a, lo_b, L, H / 2, W / 2, s, nullptr, nullptr, h.n > 1 ? &h.inner->up1[0] : &h.low2[0]));
    float* low2 = alloc(n_lo);
    if (h.n > 1) {
        SUO_TRY(hourglass(*h.inner, lo_b, low2, L, H / 2, W / 2, s, depth_idx + 1));
    } else {
        float* t = alloc(n_lo);
        SUO_TRY(residual(h.low2[0], lo_b, t, L, H / 2, W / 2, s, nullptr, nullptr, &h.low2[1]));
        SUO_TRY(residual(h.low2[1], t, low2, L, H / 2, W / 2, s, nullptr, nullptr, &h.low3[0]));
    }
    float* l3a = alloc(n_lo);
    float* l3b = alloc(n_lo);
    SUO_TRY(residual(h.low3[0], low2, l3a, L, H / 2, W / 2, s, nullptr, nullptr, &h.low3[1]));
    SUO_TRY(residual(h.low3[1], l3a, l3b, L, H / 2, W / 2, s));
    SUO_HIP_LIVE(hipStreamWaitEvent(s, ev_join, 0));
    if (up_in_tail) SUO_TRY(residual(h.up1[1], up_a, out, L, H, W, s, l3b));
    else { acct(ACCT_ELTWISE, 4.0 * L * H * W * C * 2.25); SUO_LAUNCH(launch_upsample2_add(up_b, l3b, out, L, H, W, C, s)); }
    return SUO_OK;
}

// HourglassNet.forward (hg.py:95-119) from the staged NHWC input to NCHW logits
int Net::backbone(const float* in0, int in_c, float* logits, int L, hipStream_t s, bool stem_done) {
    ws_used_ = ws_mark_;
    ev_next_ = 0;
    pre_.clear();
    float* stem = stem_slab_;
    if (stem_done && stem_computes_r1_conv1()) pre_.push_back({stem, &r1_, stem_mid1_slab_});      // (the fused stem launch leaves r1's conv1 there)
    if (!stem_done) {                                          // (the fused stem of the prior-less pass has filled the slab already: csrc/stem_x3.hip)
        ConvArgs c = {};
        const ConvW& sw = in_c == IMG_C ? stem_img_ : stem_;
        c.in = in0; c.L = L; c.H = CROP; c.W = CROP; c.C = in_c; c.Wp = sw.Wp; c.bias = sw.bias;
        c.out = stem; c.OH = 128; c.OW = 128; c.N = 64; c.relu = 1;
        acct(ACCT_STAGE, conv_bytes(c, 4, 49, false));
        SUO_LAUNCH(launch_conv7x7s2(c, s));
    }
    // pool(r1(x)): the full-resolution r1 output has no other reader, so only its pooled form is written (csrc/gemm_persist.hip: POOL)
    float* p1 = alloc((size_t)L * 64 * 64 * 128);
    SUO_TRY(residual(r1_, stem, nullptr, L, 128, 128, s, nullptr, p1));
    float* r4o = alloc((size_t)L * 64 * 64 * 128);
    SUO_TRY(residual(r4_, p1, r4o, L, 64, 64, s));
    float* x = alloc((size_t)L * 64 * 64 * 256);
    float* xp = alloc((size_t)L * 32 * 32 * 256);             // max_pool2d(x): the first thing each Hourglass computes from x (hg.py:41)
    SUO_TRY(residual(r5_, r4o, x, L, 64, 64, s, nullptr, xp));
    const int M = L * 64 * 64;
    for (int i = 0; i < 2; ++i) {
        float* hg = alloc((size_t)M * 256);
        SUO_TRY(hourglass(hg_[i], x, hg, L, 64, 64, s, 0, xp));
        float* ra = alloc((size_t)M * 256);
        float* rb = alloc((size_t)M * 256);
        SUO_TRY(residual(post_[i][0], hg, ra, L, 64, 64, s, nullptr, nullptr, &post_[i][1]));
        SUO_TRY(residual(post_[i][1], ra, rb, L, 64, 64, s));
        // the last stack's lin -> head pair: `ll` has one reader, so it never leaves the CU (one launch, 1.2 GB of traffic instead of 3.3 at 256 crops)
        static const int chain_head = (int)SUO_TUNE("SUO_CHAIN_HEAD", 1);            // 0: A/B
        const long chain_min_rows = x3_min_rows();
        if (i == 1 && chain_head && pipe_ == 2 && lin_[i].W16 && head_[i].W16 && lin_[i].N == 256 && lin_[i].K1 == 256 && M >= chain_min_rows &&
            gemm_chain_head_takes(M, 256, NUM_KP, HEAT * HEAT)) {
            acct(ACCT_GEMM, 4.0 * M * 256 + 4.0 * M * NUM_KP + 4.0 * (256.0 * 256 + 64.0 * 256));
            SUO_LAUNCH(launch_gemm_chain_head(rb, 256, M, reinterpret_cast<const uint16_t*>(lin_[i].W16), lin_[i].osc16, lin_[i].bias, reinterpret_cast<const uint16_t*>(head_[i].W16),
                                              head_[i].osc16, head_[i].bias, logits, NUM_KP, HEAT * HEAT, range_flag_, s));
            continue;
        }
        float* ll = alloc((size_t)M * 256);
        GemmArgs gl = {};
        gl.A1 = rb; gl.lda1 = 256; gl.K1 = 256; gl.Wp = lin_[i].Wp; gl.bias = lin_[i].bias; gl.out = ll; gl.ldo = 256;
        gl.M = M; gl.N = 256; gl.n_valid = 256; gl.relu = 1;
        SUO_TRY(gemm_maybe_pooled(gl, L, 64, 64, nullptr, s, &lin_[i]));
        GemmArgs gh = {};
        gh.A1 = ll; gh.lda1 = 256; gh.K1 = 256; gh.Wp = head_[i].Wp; gh.bias = head_[i].bias; gh.M = M; gh.N = 64;
        if (i == 0) {
            // x <- x + ll_(ll) + tmpOut_(tmpOut(ll)) as one folded GEMM + residual (see the constructor)
            float* xn = alloc((size_t)M * 256);
            GemmArgs gr = {};
            gr.A1 = ll; gr.lda1 = 256; gr.K1 = 256;
            gr.Wp = reinject_.Wp; gr.bias = reinject_.bias; gr.R = x; gr.ldr = 256; gr.out = xn; gr.ldo = 256;
            gr.M = M; gr.N = 256; gr.n_valid = 256;
            xp = alloc((size_t)L * 32 * 32 * 256);
            SUO_TRY(gemm_maybe_pooled(gr, L, 64, 64, xp, s, &reinject_));
            x = xn;
        } else {
            gh.out = logits; gh.n_valid = NUM_KP; gh.nchw_hw = HEAT * HEAT;
            acct(ACCT_GEMM, gemm_bytes(gh, 4));
            SUO_LAUNCH(launch_gemm1x1(gh, s));
        }
    }
    return SUO_OK;
}

int Net::ensure_graph(float* in0, int in_c, float* logits, int L, hipStream_t s, hipGraphExec_t* exec, bool stem_done) {
    const int key = (L * 4 + (in_c == IMG_C ? 1 : 0) + (stem_done ? 2 : 0)) * 4 + pipe_;      // one captured graph per (crop count, staging layout, with / without the stem, pipe)
    auto it = graphs_.find(key);
    if (it == graphs_.end()) {
        GraphEntry ge;
        SUO_HIP_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
        int r;
        try {
            r = backbone(in0, in_c, logits, L, s, stem_done);
        } catch (...) {                                   // (an exception inside the schedule must not leave the stream capturing)
            hipGraph_t dead = nullptr;
            (void)hipStreamEndCapture(s, &dead);
            if (dead) (void)hipGraphDestroy(dead);
            throw;
        }
        hipError_t e = hipStreamEndCapture(s, &ge.graph);
        if (r != SUO_OK) return r;
        SUO_HIP_CHECK(e);
        for (int i = 0; i < 2; ++i) SUO_HIP_CHECK(hipGraphInstantiate(&ge.exec[i], ge.graph, nullptr, nullptr, 0));
        it = graphs_.emplace(key, ge).first;
    }
    *exec = it->second.exec[it->second.next];
    it->second.next ^= 1;
    return SUO_OK;
}

int Net::run_backbone(float* in0, int in_c, float* logits, int L, hipStream_t s, bool stem_done) {
    if (use_graph_) {
        hipGraphExec_t exec = nullptr;
        SUO_TRY(ensure_graph(in0, in_c, logits, L, s, &exec, stem_done));
        SUO_HIP_CHECK(hipGraphLaunch(exec, s));
        return SUO_OK;
    }
    return backbone(in0, in_c, logits, L, s, stem_done);
}

// A NULL-stream call runs on an internal NON-BLOCKING stream, which the legacy NULL stream does not order: whatever the caller enqueued there before the call
// (the frame's upload kernel, torch ops that produced the boxes) must have run before this call's first kernel reads it.
int Net::follow_null_stream() {
    hipEvent_t ev = ev_[kNumEvents - 1];
    SUO_HIP_CHECK(hipEventRecord(ev, nullptr));
    SUO_HIP_CHECK(hipStreamWaitEvent(own_stream_, ev, 0));
    return SUO_OK;
}

int Net::set_pipe(int p) {
    if (p < 0 || p > pipe_built_ || (p == 0 && pipe_built_ != 0)) {
        suo_set_error("suo_net_set_pipe: pipe %d not available (this network was built for pipe %d; the fp32 pipe needs SUO_WINO_BF16X3=0 at creation)", p, pipe_built_);
        return SUO_ERR_ARG;
    }
    pipe_ = p;
    return SUO_OK;
}

// The contract of the fp16 form (csrc/f16x2.h): a forward whose activations left fp16's range has INVALID outputs.  Whoever synchronised on them asks here
// before using them; on 1 the network has already been moved to the bf16 form (which has fp32's range) and the caller re-issues the call.  The blocking
// entries (stream == NULL) do this themselves.
int Net::range_exceeded() {
    const unsigned f = __atomic_exchange_n(range_flag_, 0u, __ATOMIC_RELAXED);
    if (!f) return 0;
    if (pipe_ == 2) {
        pipe_ = 1;
        fprintf(stderr, "libsuo_hip: an activation left the fp16 range (|x| >= %g); this network now runs the three-term bf16 form -- re-issue the call\n", (double)(S2_LIMIT / S2_XSCALE));
    }
    return 1;
}

// The launch schedule of ONE call of L crops cut from n_frames frames of H x W (with_priors: the 48-channel staging + full stem) walked as a dry run (nothing is
// launched, nothing allocated): algorithmic HBM bytes per kind of launch -- out[0..5] = staging / stem, 3x3 (+ fused tails), 1x1 GEMMs, one-launch blocks,
// pool / up-sample, decode + classifier -- and the number of launches.
int Net::schedule_bytes(int L, int n_frames, int H, int W, int with_priors, double* out, int* n_launches) {
    if (L <= 0 || L > max_crops_ || !out) { suo_set_error("suo_net_schedule_bytes: L=%d outside [1,%d] or null output", L, max_crops_); return SUO_ERR_ARG; }
    const bool was_dry = dry_run_;
    dry_run_ = true; acct_on_ = true; acct_launches_ = 0;
    for (int k = 0; k < ACCT_KINDS; ++k) acct_[k] = 0.0;
    int rc = SUO_OK;
    try {
        const double frame_bytes = (double)n_frames * H * W * 3;
        const int in_c = with_priors ? IN_C : IMG_C;
        const bool fstem = !with_priors && fused_stem();
        if (fstem) {
            acct(ACCT_STAGE, frame_bytes + 4.0 * L * 128 * 128 * 64 * (stem_computes_r1_conv1() ? 2 : 1) + 4.0 * 64 * 147);
            ++acct_launches_;
        } else {
            acct(ACCT_STAGE, frame_bytes + 4.0 * L * CROP * CROP * in_c + (with_priors ? 8.0 * L * NUM_KP : 0.0));
            ++acct_launches_;
        }
        rc = backbone(nullptr, in_c, nullptr, L, nullptr, fstem);
        acct(ACCT_DECODE, 4.0 * L * NUM_KP * (HEAT * HEAT + 2 + 4 + 1) + 4.0 * L * NUM_KP * 3 + 4.0 * NUM_KP * (NUM_KP + 1));
        acct_launches_ += 2;
    } catch (const std::exception& e) {
        suo_set_error("suo_net_schedule_bytes: %s", e.what());
        rc = SUO_ERR_ARG;
    }
    dry_run_ = was_dry; acct_on_ = false;
    for (int k = 0; k < ACCT_KINDS; ++k) out[k] = acct_[k];
    if (n_launches) *n_launches = acct_launches_;
    return rc;
}

// SUO_STEM_X3=0: the prior-less pass stages the crop (roi_align_concat_kernel) and runs the stem on the fp32 pipe inside the backbone, as rounds 1-3 did
// the fused stem launch of the fp16 pipe also computes r1's conv1 on its tile (csrc/stem_x3.hip: NEXT); a function of the network's state only -- suo_net_prepare captures the
// backbone without launching the stem
bool Net::stem_computes_r1_conv1() const {
    static const int on = (int)SUO_TUNE("SUO_STEM_NEXT", 1);              // 0: A/B
    return on && fused_stem() && pipe_ == 2 && stem_h2_w_ && r1_.c1.W16 && r1_.c1.osc16 && r1_.cin == 64 && r1_.c1.N == 64 && r1_.c1.K1 == 64 && r1_.c1.K2 == 0;
}
bool Net::fused_stem() const {
    static const int on = (int)suo::env_switch("SUO_STEM_X3", 1);
    return on != 0 && stem_x3_w_ != nullptr;
}

// Capture the backbone graph for L crops ahead of time (nothing runs): a stream of frames with a varying number of
// detections then never pays a capture inside a timed / latency-critical call.
int Net::prepare(int L, int with_priors, hipStream_t s) {
    if (L <= 0 || L > max_crops_) { suo_set_error("suo_net_prepare: L=%d outside [1,%d]", L, max_crops_); return SUO_ERR_ARG; }
    if (!use_graph_) return SUO_OK;
    const bool own = (s == nullptr);
    if (own) s = own_stream_;
    try {
        ws_used_ = 0;                                             // the same persistent slabs as forward()
        float* in0 = alloc((size_t)max_crops_ * CROP * CROP * IN_C);
        float* logits = alloc((size_t)max_crops_ * NUM_KP * HEAT * HEAT);
        stem_slab_ = alloc((size_t)max_crops_ * 128 * 128 * 64);          // the stem's output: persistent, the fused stem writes it OUTSIDE the captured graph
        stem_mid1_slab_ = alloc((size_t)max_crops_ * 128 * 128 * 64);     // ... and r1's conv1 of it, when the stem launch computes that too
        ws_mark_ = ws_used_;
        hipGraphExec_t exec = nullptr;
        SUO_TRY(ensure_graph(in0, with_priors ? IN_C : IMG_C, logits, L, s, &exec, !with_priors && fused_stem()));
    } catch (const std::exception& e) {
        suo_set_error("suo_net_prepare: %s", e.what());
        return SUO_ERR_ARG;
    }
    return SUO_OK;
}

// Backbone only, from an already staged NHWC [L,256,256,48] input (test / profiling entry).
int Net::forward_staged(const float* in0_user, int L, float* logits_out, hipStream_t s) {
    if (L <= 0 || L > max_crops_) { suo_set_error("suo_net_backbone: L=%d outside [1,%d]", L, max_crops_); return SUO_ERR_ARG; }
    const bool own = (s == nullptr);
    if (own) { s = own_stream_; SUO_TRY(follow_null_stream()); }
    try {
        ws_used_ = 0;
        float* in0 = alloc((size_t)max_crops_ * CROP * CROP * IN_C);
        float* logits = alloc((size_t)max_crops_ * NUM_KP * HEAT * HEAT);
        stem_slab_ = alloc((size_t)max_crops_ * 128 * 128 * 64);          // the stem's output: persistent, the fused stem writes it OUTSIDE the captured graph
        stem_mid1_slab_ = alloc((size_t)max_crops_ * 128 * 128 * 64);     // ... and r1's conv1 of it, when the stem launch computes that too
        ws_mark_ = ws_used_;
        if (in0_user)
            SUO_HIP_CHECK(hipMemcpyAsync(in0, in0_user, (size_t)L * CROP * CROP * IN_C * sizeof(float), hipMemcpyDeviceToDevice, s));
        SUO_TRY(run_backbone(in0, IN_C, logits, L, s));
        if (logits_out)
            SUO_HIP_CHECK(hipMemcpyAsync(logits_out, logits, (size_t)L * NUM_KP * HEAT * HEAT * sizeof(float), hipMemcpyDeviceToDevice, s));
    } catch (const std::exception& e) {
        suo_set_error("suo_net_backbone: %s", e.what());
        return SUO_ERR_ARG;
    }
    if (own) {
        SUO_HIP_CHECK(hipStreamSynchronize(s));
        if (range_exceeded()) return forward_staged(in0_user, L, logits_out, nullptr);      // (now on the bf16 form: cannot recurse twice)
    }
    return SUO_OK;
}

int Net::forward(const void* img, int fmt, int H, int W, const float* boxes, const int* box_img, int L, const float* priors,
                 const float* prior_uv, const uint8_t* prior_mask, float* uv, float* cov,
                 float* kp_prob, float* kp_logit, float* logits_out, hipStream_t s) {
    if (L <= 0 || L > max_crops_) { suo_set_error("suo_net_forward: L=%d outside [1,%d]", L, max_crops_); return SUO_ERR_ARG; }
    const bool own = (s == nullptr);
    if (own) { s = own_stream_; SUO_TRY(follow_null_stream()); }   // the legacy NULL stream cannot be captured: run on an internal stream and block
    try {
        // persistent slabs at the bottom of the workspace: staged input + logits
        ws_used_ = 0;
        float* in0 = alloc((size_t)max_crops_ * CROP * CROP * IN_C);
        float* logits = alloc((size_t)max_crops_ * NUM_KP * HEAT * HEAT);
        stem_slab_ = alloc((size_t)max_crops_ * 128 * 128 * 64);          // the stem's output: persistent, the fused stem writes it OUTSIDE the captured graph
        stem_mid1_slab_ = alloc((size_t)max_crops_ * 128 * 128 * 64);     // ... and r1's conv1 of it, when the stem launch computes that too
        ws_mark_ = ws_used_;
        const int in_c = (priors || prior_uv) ? IN_C : IMG_C;     // the slab is sized for IN_C; the prior-less layout uses a sixth of it
        if (in_c == IMG_C && fused_stem()) {
            // prior-less pass: RoIAlign + stem in one launch on the bf16 pipe (csrc/stem_x3.hip), ahead of the captured backbone (the frame and
            // the boxes are the caller's buffers: their addresses change from call to call, a captured launch could not take them)
            if (stem_computes_r1_conv1()) {
                // r1's conv1 on the tile while the stem has it (the stem's output is read by r1's skip convolution only)
                const StemNext nx = {r1_.pro_scale, r1_.pro_shift, reinterpret_cast<const uint16_t*>(r1_.c1.W16), r1_.c1.osc16, r1_.c1.bias, stem_mid1_slab_};
                SUO_LAUNCH(launch_stem_x3(img, fmt, H, W, boxes, box_img, L, reinterpret_cast<const uint16_t*>(stem_h2_w_), stem_x3_bias_, stem_slab_, s, stem_h2_osc_, range_flag_, &nx));
            } else if (pipe_ == 2 && stem_h2_w_)
                SUO_LAUNCH(launch_stem_x3(img, fmt, H, W, boxes, box_img, L, reinterpret_cast<const uint16_t*>(stem_h2_w_), stem_x3_bias_, stem_slab_, s, stem_h2_osc_, range_flag_));
            else
            SUO_LAUNCH(launch_stem_x3(img, fmt, H, W, boxes, box_img, L, reinterpret_cast<const uint16_t*>(stem_x3_w_), stem_x3_bias_, stem_slab_, s));
            SUO_TRY(run_backbone(in0, in_c, logits, L, s, true));
        } else {
            SUO_LAUNCH(launch_roi_align_concat(img, fmt, H, W, boxes, box_img, L, in_c, priors, prior_uv, prior_mask, in0, s));
            SUO_TRY(run_backbone(in0, in_c, logits, L, s, false));
        }
        SUO_LAUNCH(launch_decode(logits, L, uv, cov, d_mean_logit_, nullptr, nullptr, s));
        SUO_LAUNCH(launch_classifier(d_mean_logit_, cls_w_, cls_b_, L, kp_logit, kp_prob, s));
        if (logits_out)
            SUO_HIP_CHECK(hipMemcpyAsync(logits_out, logits, (size_t)L * NUM_KP * HEAT * HEAT * sizeof(float), hipMemcpyDeviceToDevice, s));
    } catch (const std::exception& e) {
        suo_set_error("suo_net_forward: %s", e.what());
        return SUO_ERR_ARG;
    }
    if (own) {
        SUO_HIP_CHECK(hipStreamSynchronize(s));
        if (range_exceeded()) return forward(img, fmt, H, W, boxes, box_img, L, priors, prior_uv, prior_mask, uv, cov, kp_prob, kp_logit, logits_out, nullptr);
    }
    return SUO_OK;
}

}  // namespace suo
