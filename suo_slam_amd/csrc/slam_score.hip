// SLAM-mode hypothesis scoring: the chi-square inlier counts behind __estimate_camera_pose
// (/root/reference/lib/object_slam.py:975-1072: every PnP-derived camera hypothesis against every object of the view) and __maybe_reinit_objects
// (:595-697: each object's PnP pose and map pose against its detections of the last 15 views).  Per (pose, detection) pair the reference does
//     p = pts R^T + t;  uvw = p K^T;  r = uv - uvw.xy / uvw.z  (z > 0 only);  chi2 = r^T Sigma^-1 r  (Sigma's diagonal clamped at 1e-4, :669,:1054;
//     or |r|^2 / manual_kp_std^2 without covariances);  count(chi2 <= 5.991)
// as numpy calls per pair -- O(objects^2 + 15 objects) pairs per view.  Here the detections live in a device-resident store (a row per detection,
// written once when the detection is made), a call ships 112 bytes per pair (pose, keypoint selection, store slot) and gets the counts back:
// one wave per pair, one keypoint per lane, fp64, the expressions in the order the host restatement evaluates them
// (tests/host_scoring.py: _chi2_inliers_many; no contraction: the library is built with -ffp-contract=off).  Latency work, not bandwidth:
// 64-240 pairs x 41 keypoints; what it removes is ~0.25 ms of numpy per call from a 6.4 ms view.
#include <math.h>
#include <string.h>

#include <mutex>

#include "../../include/suo_hip.h"
#include "suo_internal.h"

#define SS_TRY(x) do { int _r = (x); if (_r != SUO_OK) return _r; } while (0)

namespace suo {

constexpr int SS_KP = 41;                                      // keypoints per detection row (NUM_KP)
constexpr int SS_PTS = 0, SS_UV = SS_KP * 3, SS_COV = SS_UV + SS_KP * 2, SS_K = SS_COV + SS_KP * 4, SS_N = SS_K + 9, SS_HAS_COV = SS_N + 1;
constexpr int SS_ROW = SS_HAS_COV + 1;                         // 380 doubles
constexpr int SS_PAIR = 14;                                    // doubles per pair record: T[3][4], selection bits (uint64), slot (int64)
static_assert(SS_ROW == SUO_SLAM_ROW && SS_PAIR == SUO_SLAM_PAIR, "include/suo_hip.h");

__global__ __launch_bounds__(256) void slam_score_kernel(const double* __restrict__ store, const double* __restrict__ pairs, int n_pairs,
                                                         double chi2_max, double kp_std2, int* __restrict__ counts) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= n_pairs) return;
    const double* pr = pairs + (size_t)b * SS_PAIR;
    const unsigned long long sel = __double_as_longlong(pr[12]);
    const long long slot = __double_as_longlong(pr[13]);
    const double* row = store + (size_t)slot * SS_ROW;
    bool in = false, bad = false;
    if (lane < SS_KP && ((sel >> lane) & 1ull)) {
        const double x = row[SS_PTS + lane * 3], y = row[SS_PTS + lane * 3 + 1], z = row[SS_PTS + lane * 3 + 2];
        double p[3], w[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) p[j] = ((x * pr[j * 4] + y * pr[j * 4 + 1]) + z * pr[j * 4 + 2]) + pr[j * 4 + 3];
#pragma unroll
        for (int j = 0; j < 3; ++j) w[j] = (p[0] * row[SS_K + j * 3] + p[1] * row[SS_K + j * 3 + 1]) + p[2] * row[SS_K + j * 3 + 2];
        if (w[2] > 0.0) {
            const double rx = row[SS_UV + lane * 2] - w[0] / w[2], ry = row[SS_UV + lane * 2 + 1] - w[1] / w[2];
            double chi2;
            if (row[SS_HAS_COV] != 0.0) {
                const double* c = row + SS_COV + lane * 4;
                const double a = fmax(c[0], 1e-4), d = fmax(c[3], 1e-4), bb = c[1], cc = c[2];
                chi2 = ((d * rx * rx - (bb + cc) * rx * ry) + a * ry * ry) / (a * d - bb * cc);
            } else {
                chi2 = (rx * rx + ry * ry) / kp_std2;
            }
            bad = chi2 != chi2;
            in = chi2 <= chi2_max;
        }
    }
    const unsigned long long m = __ballot(in), mb = __ballot(bad);
    if (lane == 0) {
        counts[b] = __popcll(m);
        if (mb) atomicOr(&counts[n_pairs], 1);                 // "NaN in information matrix" (the host asserts on it)
    }
}

struct SlamStore {
    double* rows = nullptr; int capacity = 0;
    int written = 0;                                            // slots [0, written) have been uploaded; a pair may only name those (the rest of the buffer is uninitialised)
    char* dev = nullptr; char* host = nullptr; size_t cap = 0;   // per-call scratch (device, pinned host), grow-only
    hipStream_t stream = nullptr;
    std::mutex mu;
};

static int ss_scratch(SlamStore* s, size_t bytes) {
    if (bytes <= s->cap) return SUO_OK;
    size_t c = s->cap ? s->cap : 65536;
    while (c < bytes) c *= 2;
    if (s->dev) SUO_HIP_CHECK(hipFree(s->dev));
    if (s->host) SUO_HIP_CHECK(hipHostFree(s->host));
    s->dev = s->host = nullptr; s->cap = 0;
    SUO_HIP_CHECK(hipMalloc((void**)&s->dev, c));
    SUO_HIP_CHECK(hipHostMalloc((void**)&s->host, c, hipHostMallocDefault));
    s->cap = c;
    return SUO_OK;
}

static int ss_reserve(SlamStore* s, int slots) {
    if (slots <= s->capacity) return SUO_OK;
    int c = s->capacity ? s->capacity : 256;
    while (c < slots) c *= 2;
    double* n = nullptr;
    SUO_HIP_CHECK(hipMalloc((void**)&n, (size_t)c * SS_ROW * sizeof(double)));
    if (s->rows) {
        SUO_HIP_CHECK(hipMemcpyAsync(n, s->rows, (size_t)s->capacity * SS_ROW * sizeof(double), hipMemcpyDeviceToDevice, s->stream));
        SUO_HIP_CHECK(hipStreamSynchronize(s->stream));
        SUO_HIP_CHECK(hipFree(s->rows));
    }
    s->rows = n; s->capacity = c;
    return SUO_OK;
}

}  // namespace suo

using suo::SlamStore;

extern "C" {

void* suo_slam_store_create(int capacity) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) { suo_set_error("suo_slam_store_create: no GPU"); return nullptr; }
    SlamStore* s = new SlamStore();
    if (hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess || suo::ss_reserve(s, capacity > 0 ? capacity : 256) != SUO_OK) {
        suo_set_error("suo_slam_store_create: allocation failed");
        delete s;
        return nullptr;
    }
    return s;
}

void suo_slam_store_destroy(void* h) {
    SlamStore* s = (SlamStore*)h;
    if (!s) return;
    if (s->rows) (void)hipFree(s->rows);
    if (s->dev) (void)hipFree(s->dev);
    if (s->host) (void)hipHostFree(s->host);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
}

int suo_slam_store_put(void* h, int first_slot, int n, const double* rows_host) {
    SlamStore* s = (SlamStore*)h;
    if (!s || first_slot < 0 || n < 0 || (n && !rows_host)) { suo_set_error("suo_slam_store_put: bad arguments"); return SUO_ERR_ARG; }
    if (n == 0) return SUO_OK;
    std::lock_guard<std::mutex> lk(s->mu);
    if (first_slot > s->written) { suo_set_error("suo_slam_store_put: slots %d..%d would leave a hole after %d written slots", first_slot, first_slot + n - 1, s->written); return SUO_ERR_ARG; }
    const size_t bytes = (size_t)n * suo::SS_ROW * sizeof(double);
    SS_TRY(suo::ss_reserve(s, first_slot + n));
    SS_TRY(suo::ss_scratch(s, bytes));
    memcpy(s->host, rows_host, bytes);
    SUO_HIP_CHECK(hipMemcpyAsync(s->rows + (size_t)first_slot * suo::SS_ROW, s->host, bytes, hipMemcpyHostToDevice, s->stream));
    SUO_HIP_CHECK(hipStreamSynchronize(s->stream));            // (the pinned scratch is the next call's too)
    if (first_slot + n > s->written) s->written = first_slot + n;
    return SUO_OK;
}

int suo_slam_score(void* h, int n_pairs, const double* pairs_host, double chi2_max, double kp_std2, int32_t* counts_host) {
    SlamStore* s = (SlamStore*)h;
    if (!s || n_pairs < 0 || (n_pairs && (!pairs_host || !counts_host))) { suo_set_error("suo_slam_score: bad arguments"); return SUO_ERR_ARG; }
    if (n_pairs == 0) return SUO_OK;
    std::lock_guard<std::mutex> lk(s->mu);
    for (int i = 0; i < n_pairs; ++i) {
        long long slot;
        memcpy(&slot, pairs_host + (size_t)i * suo::SS_PAIR + 13, 8);
        if (slot < 0 || slot >= s->written) { suo_set_error("suo_slam_score: pair %d names slot %lld, %d slots written", i, slot, s->written); return SUO_ERR_ARG; }
    }
    const size_t pb = (size_t)n_pairs * suo::SS_PAIR * sizeof(double), cb = (size_t)(n_pairs + 1) * sizeof(int32_t);
    const size_t co = (pb + 255) / 256 * 256;
    SS_TRY(suo::ss_scratch(s, co + cb));
    memcpy(s->host, pairs_host, pb);
    memset(s->host + co, 0, cb);
    SUO_HIP_CHECK(hipMemcpyAsync(s->dev, s->host, co + cb, hipMemcpyHostToDevice, s->stream));      // pairs + zeroed counts / NaN flag in one copy
    hipLaunchKernelGGL(suo::slam_score_kernel, dim3((n_pairs + 3) / 4), dim3(256), 0, s->stream, s->rows, (const double*)s->dev, n_pairs, chi2_max,
                       kp_std2, (int*)(s->dev + co));
    SUO_HIP_CHECK(hipGetLastError());
    SUO_HIP_CHECK(hipMemcpyAsync(s->host + co, s->dev + co, cb, hipMemcpyDeviceToHost, s->stream));
    SUO_HIP_CHECK(hipStreamSynchronize(s->stream));
    memcpy(counts_host, s->host + co, cb);
    return SUO_OK;
}

}  // extern "C"
