// Where does a workgroup of the dominant 3x3 convolution spend its time?  Builds csrc/conv.hip with -DSUO_CONV_PROFILE
// (thread 0 of every workgroup records the wall clock at: entry, first chunk staged, after each of the 4 channel
// chunks, exit, plus its XCC / HW_ID) and prints the phase statistics and the timeline of one compute unit.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DSUO_CONV_PROFILE -I suo_slam_amd/csrc -I include \
//       tools/micro/conv_prof.hip -o /tmp/conv_prof && /tmp/conv_prof [crops]
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <map>
#include <vector>

#include "../../suo_slam_amd/csrc/conv.hip"

void suo_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
namespace suo {
int launch_conv3x3_small(const ConvArgs&, hipStream_t) { return SUO_ERR_ARG; }
int launch_gemm_small(const GemmArgs&, hipStream_t) { return SUO_ERR_ARG; }
int launch_gemm_persist(const GemmArgs&, int, hipStream_t) { return SUO_ERR_ARG; }
}  // namespace suo

int main(int argc, char** argv) {
    const int L = argc > 1 ? atoi(argv[1]) : 128, H = 64, W = 64, C = 128, N = 128;
    const size_t n_in = (size_t)L * H * W * C, n_w = (size_t)N * C * 9;
    std::vector<float> h(n_w);
    for (size_t i = 0; i < n_w; ++i) h[i] = (float)((i * 2654435761u) >> 20 & 1023) / 4096.f - 0.125f;
    float *in, *out, *wp, *bias;
    hipMalloc(&in, n_in * 4); hipMalloc(&out, n_in * 4); hipMalloc(&wp, n_w * 4); hipMalloc(&bias, N * 4);
    hipMemset(in, 0, n_in * 4); hipMemset(bias, 0, N * 4);
    hipMemcpy(wp, h.data(), n_w * 4, hipMemcpyHostToDevice);
    suo::ConvArgs a{};
    a.in = in; a.L = L; a.H = H; a.W = W; a.C = C; a.Wp = wp; a.bias = bias; a.out = out; a.OH = H; a.OW = W; a.N = N; a.relu = 1;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) suo::launch_conv3x3(a, 0);
    hipEventRecord(e0, 0);
    if (suo::launch_conv3x3(a, 0) != SUO_OK) return 1;
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const int nwg = L * (H / 8) * (W / 16);
    std::vector<long long> p((size_t)16384 * 8);
    hipMemcpyFromSymbol(p.data(), HIP_SYMBOL(suo::g_conv_prof), p.size() * 8);
    const int n = std::min(nwg, 16384);
    long long t0 = p[0], t1 = p[6];
    for (int b = 0; b < n; ++b) { t0 = std::min(t0, p[b * 8]); t1 = std::max(t1, p[b * 8 + 6]); }
    const double tick_us = 0.01;     // s_memrealtime: 100 MHz
    printf("%d crops, %d workgroups: event time %.1f us, first entry -> last exit %.1f us = %.1f TFLOP/s\n", L, nwg, ms * 1e3,
           (t1 - t0) * tick_us, 2.0 * L * H * W * C * N * 9 / ((t1 - t0) * tick_us * 1e-6) / 1e12);
    const char* name[6] = {"prologue (first chunk staged)", "chunk 0", "chunk 1", "chunk 2", "chunk 3", "epilogue"};
    for (int ph = 0; ph < 6; ++ph) {
        std::vector<double> d(n);
        for (int b = 0; b < n; ++b) d[b] = (p[b * 8 + ph + 1] - p[b * 8 + ph]) * tick_us;
        std::sort(d.begin(), d.end());
        double s = 0;
        for (double v : d) s += v;
        printf("  %-30s mean %7.2f us   p10 %7.2f   median %7.2f   p90 %7.2f\n", name[ph], s / n, d[n / 10], d[n / 2], d[n * 9 / 10]);
    }
    {
        std::vector<double> d(n);
        double s = 0;
        for (int b = 0; b < n; ++b) { d[b] = (p[b * 8 + 6] - p[b * 8]) * tick_us; s += d[b]; }
        std::sort(d.begin(), d.end());
        printf("  %-30s mean %7.2f us   p10 %7.2f   median %7.2f   p90 %7.2f   (ideal alone at 155 TFLOP/s/256 CUs: %.1f us)\n", "workgroup total", s / n,
               d[n / 10], d[n / 2], d[n * 9 / 10], 2.0 * 128 * 128 * 1152 / (155e12 / 256) * 1e6);
    }
    {
        std::vector<long long> ck((size_t)16384 * 8);
        hipMemcpyFromSymbol(ck.data(), HIP_SYMBOL(suo::g_conv_prof_clk), ck.size() * 8);
        double cyc = 0, us = 0;
        for (int b = 0; b < n; ++b) { cyc += (double)(ck[b * 8 + 5] - ck[b * 8 + 1]); us += (p[b * 8 + 5] - p[b * 8 + 1]) * tick_us; }
        printf("  chunk loop: %.0f s_memtime ticks per workgroup in %.2f us = %.1f MHz; %.2f ticks per MFMA of a wave (2304 per tile)\n", cyc / n, us / n,
               cyc / us, cyc / n / 2304.0);
    }
    // residency: workgroups per (xcc, cu-ish key), busy time and the gaps between consecutive workgroups of one slot
    std::map<long long, std::vector<int>> by_cu;
    for (int b = 0; b < n; ++b) by_cu[(p[b * 8 + 7] >> 32 << 16) | ((p[b * 8 + 7] >> 8) & 0xffff)].push_back(b);
    printf("  %zu distinct (XCC, SE/SH/CU) keys, %.1f workgroups each\n", by_cu.size(), (double)n / by_cu.size());
    int shown = 0;
    for (auto& kv : by_cu) {
        if (shown++ >= 2) break;
        auto v = kv.second;
        std::sort(v.begin(), v.end(), [&](int x, int y) { return p[x * 8] < p[y * 8]; });
        printf("  timeline of key %llx (us since first entry):\n", kv.first);
        for (int b : v)
            printf("    wg %5d  wave-slot %2lld  start %8.2f  staged %8.2f  chunks %8.2f %8.2f %8.2f %8.2f  end %8.2f\n", b, p[b * 8 + 7] & 0xf,
                   (p[b * 8] - t0) * tick_us, (p[b * 8 + 1] - t0) * tick_us, (p[b * 8 + 2] - t0) * tick_us, (p[b * 8 + 3] - t0) * tick_us,
                   (p[b * 8 + 4] - t0) * tick_us, (p[b * 8 + 5] - t0) * tick_us, (p[b * 8 + 6] - t0) * tick_us);
    }
    return 0;
}
