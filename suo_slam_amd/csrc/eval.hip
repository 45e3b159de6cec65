// ADD / ADD-S pose errors of the evaluation meter.
//
// Replaces the distance computations of EvalMeter.update (/root/reference/lib/utils/eval_meter.py:126-155):
//   pred = pts R_pred^T + t_pred,  gt = pts R_gt^T + t_gt                       (utils.transform_pts, utils.py:454-460)
//   ADD   = mean_i |gt_i - pred_i|                                              (__dists_add,     eval_meter.py:233-235)
//   ADD-S = mean_i min_j |gt_i - pred_j|                                        (__dists_add_sym, eval_meter.py:241-242)
// in fp32 like the reference (it casts poses to float32, eval_meter.py:133-134).  The reference materialises the
// [P,P,3] difference tensor (12 P^2 bytes) -- here the P^2 pairs never leave registers: a thread owns one gt point,
// the pred points of the block's slice are transformed once into LDS and read back as wave-uniform (broadcast)
// ds_read_b128, and the running minimum of the SQUARED distance is kept (sqrt is monotone, so min and sqrt commute
// exactly).  VALU-bound: 7 fp32 ops per pair; this is byte/ALU work, not a GEMM (|g|^2+|p|^2-2g.p would cancel).
// One evaluate.py update carries a single pose, so the pred points are additionally split over blockIdx.y and merged
// with an unsigned atomicMin on the float bits (order-preserving for non-negative floats) to fill 256 CUs.
#include <math.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "../../include/suo_hip.h"
#include "suo_internal.h"

namespace suo {

constexpr int EV_BLOCK = 256;     // threads per block
constexpr int EV_G = 4;           // gt points per thread (pair kernel)
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int EV_TILE = 1024;     // pred points staged in LDS per pass

struct MeshDb {
    int n_models = 0;
    std::vector<int> off;         // [n_models + 1] point offsets
    int max_pts = 0;
    float* pts_dev = nullptr;     // [off.back()][3]
    int* off_dev = nullptr;
    // per-call scratch, grow-only
    char* scratch_dev = nullptr; char* scratch_host = nullptr; size_t scratch_cap = 0;
    hipStream_t stream = nullptr;
    std::mutex mu;
};

struct EvalArgs {
    const float* pts; const int* off; const int* model;     // model[n]
    const float* Tp; const float* Tg;                       // [n][12] row-major 3x4
    unsigned* mind2;                                        // [n][stride] float bits of min squared distance
    float* out;                                             // [n][2] = ADD, ADD-S (means)
    int stride, splits;
};

__device__ __forceinline__ void xform(const float* T, float x, float y, float z, float& ox, float& oy, float& oz) {
    ox = ((x * T[0] + y * T[1]) + z * T[2]) + T[3];
    oy = ((x * T[4] + y * T[5]) + z * T[6]) + T[7];
    oz = ((x * T[8] + y * T[9]) + z * T[10]) + T[11];
}

__global__ __launch_bounds__(EV_BLOCK) void eval_init_kernel(EvalArgs a, int total) {
    const int i = blockIdx.x * EV_BLOCK + threadIdx.x;
    if (i < total) a.mind2[i] = 0x7f800000u;     // +inf
}

// Register blocking: a thread owns EV_G = 4 gt points as two float2 lanes, so one broadcast ds_read_b128 of a pred
// point feeds 4 pairs (the LDS pipe, shared by the CU's 4 SIMDs, would otherwise bound the loop) and the arithmetic is
// packed fp32 (v_pk_add / v_pk_mul / v_pk_fma: 3 + 1 + 2 packed ops + 2 v_min per 2 pairs = 4 VALU slots per pair).
__global__ __launch_bounds__(EV_BLOCK) void eval_pairs_kernel(EvalArgs a) {
#pragma clang fp contract(fast)
    __shared__ float4 tile[EV_TILE];
    const int z = blockIdx.z, m = a.model[z];
    const int p_begin = a.off[m], P = a.off[m + 1] - p_begin;
    const int g0 = blockIdx.x * (EV_BLOCK * EV_G);
    if (g0 >= P) return;
    int chunk = (P + a.splits - 1) / a.splits;
    chunk = (chunk + EV_BLOCK - 1) / EV_BLOCK * EV_BLOCK;
    const int j0 = blockIdx.y * chunk, j1 = min(P, j0 + chunk);
    if (j0 >= P) return;
    const float* pts = a.pts + (size_t)p_begin * 3;
    float Tp[12], Tg[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) { Tp[k] = a.Tp[z * 12 + k]; Tg[k] = a.Tg[z * 12 + k]; }
    f32x2 gx[EV_G / 2], gy[EV_G / 2], gz[EV_G / 2], best[EV_G / 2];
#pragma unroll
    for (int k = 0; k < EV_G; ++k) {
        const int g = g0 + threadIdx.x + k * EV_BLOCK;
        const int gi = g < P ? g : P - 1;
        float x, y, zz;
        xform(Tg, pts[gi * 3], pts[gi * 3 + 1], pts[gi * 3 + 2], x, y, zz);
        gx[k >> 1][k & 1] = x; gy[k >> 1][k & 1] = y; gz[k >> 1][k & 1] = zz;
        best[k >> 1][k & 1] = INFINITY;
    }
    for (int j = j0; j < j1; j += EV_TILE) {
        const int cnt = min(EV_TILE, j1 - j);
        __syncthreads();
        for (int t = threadIdx.x; t < EV_TILE; t += EV_BLOCK) {
            float4 v = make_float4(1e18f, 1e18f, 1e18f, 0.f);     // padding: never the minimum, never overflows to NaN
            if (t < cnt) xform(Tp, pts[(j + t) * 3], pts[(j + t) * 3 + 1], pts[(j + t) * 3 + 2], v.x, v.y, v.z);
            tile[t] = v;
        }
        __syncthreads();
        const int c4 = (cnt + 3) & ~3;
#pragma unroll 4
        for (int t = 0; t < c4; ++t) {
            const float4 p = tile[t];
#pragma unroll
            for (int k = 0; k < EV_G / 2; ++k) {
                const f32x2 dx = gx[k] - p.x, dy = gy[k] - p.y, dz = gz[k] - p.z;
                const f32x2 d2 = dz * dz + (dy * dy + dx * dx);
                best[k] = __builtin_elementwise_min(best[k], d2);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < EV_G; ++k) {
        const int g = g0 + threadIdx.x + k * EV_BLOCK;
        if (g < P) atomicMin(&a.mind2[(size_t)z * a.stride + g], __float_as_uint(best[k >> 1][k & 1]));
    }
}

// one block per pose: ADD recomputed directly, ADD-S from the merged minima; sums in double (the reference's
// fp32 mean differs from this by its own rounding only)
__global__ __launch_bounds__(EV_BLOCK) void eval_finalize_kernel(EvalArgs a) {
    __shared__ double red[2][EV_BLOCK / 64];
    const int z = blockIdx.x, m = a.model[z];
    const int p_begin = a.off[m], P = a.off[m + 1] - p_begin;
    const float* pts = a.pts + (size_t)p_begin * 3;
    float Tp[12], Tg[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) { Tp[k] = a.Tp[z * 12 + k]; Tg[k] = a.Tg[z * 12 + k]; }
    double s_add = 0.0, s_adds = 0.0;
    for (int g = threadIdx.x; g < P; g += EV_BLOCK) {
        float gx, gy, gz, px, py, pz;
        xform(Tg, pts[g * 3], pts[g * 3 + 1], pts[g * 3 + 2], gx, gy, gz);
        xform(Tp, pts[g * 3], pts[g * 3 + 1], pts[g * 3 + 2], px, py, pz);
        const float dx = gx - px, dy = gy - py, dz = gz - pz;
        s_add += (double)sqrtf((dx * dx + dy * dy) + dz * dz);
        s_adds += (double)sqrtf(__uint_as_float(a.mind2[(size_t)z * a.stride + g]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s_add += __shfl_down(s_add, o); s_adds += __shfl_down(s_adds, o); }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { red[0][w] = s_add; red[1][w] = s_adds; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double A = 0, S = 0;
        for (int k = 0; k < EV_BLOCK / 64; ++k) { A += red[0][k]; S += red[1][k]; }
        a.out[z * 2] = (float)(A / P);
        a.out[z * 2 + 1] = (float)(S / P);
    }
}

static int ensure_scratch(MeshDb* db, size_t bytes) {
    if (bytes <= db->scratch_cap) return SUO_OK;
    size_t ncap = (std::max(bytes, db->scratch_cap * 2) + 4095) & ~(size_t)4095;
    if (db->scratch_dev) (void)hipFree(db->scratch_dev);
    if (db->scratch_host) (void)hipHostFree(db->scratch_host);
    db->scratch_dev = nullptr; db->scratch_host = nullptr; db->scratch_cap = 0;
    SUO_HIP_CHECK(hipMalloc((void**)&db->scratch_dev, ncap));
    SUO_HIP_CHECK(hipHostMalloc((void**)&db->scratch_host, ncap, hipHostMallocDefault));
    db->scratch_cap = ncap;
    return SUO_OK;
}

}  // namespace suo

using namespace suo;

extern "C" int suo_mesh_db_create(int n_models, const int* n_pts, const float* pts, void** out) {
    if (n_models <= 0 || !n_pts || !pts || !out) { suo_set_error("suo_mesh_db_create: bad argument"); return SUO_ERR_ARG; }
    MeshDb* db = new MeshDb();
    db->n_models = n_models;
    db->off.assign(n_models + 1, 0);
    for (int i = 0; i < n_models; ++i) {
        if (n_pts[i] <= 0) { delete db; suo_set_error("suo_mesh_db_create: model %d has %d points", i, n_pts[i]); return SUO_ERR_ARG; }
        db->off[i + 1] = db->off[i] + n_pts[i];
        db->max_pts = std::max(db->max_pts, n_pts[i]);
    }
    auto fail = [&](hipError_t e, const char* what) {
        suo_set_error("suo_mesh_db_create: %s -> %s", what, hipGetErrorString(e));
        suo_mesh_db_destroy(db);
        return SUO_ERR_HIP;
    };
    hipError_t e;
    if ((e = hipStreamCreateWithFlags(&db->stream, hipStreamNonBlocking)) != hipSuccess) return fail(e, "hipStreamCreate");
    const size_t pbytes = (size_t)db->off.back() * 3 * sizeof(float);
    if ((e = hipMalloc((void**)&db->pts_dev, pbytes)) != hipSuccess) return fail(e, "hipMalloc(points)");
    if ((e = hipMalloc((void**)&db->off_dev, db->off.size() * sizeof(int))) != hipSuccess) return fail(e, "hipMalloc(offsets)");
    if ((e = hipMemcpy(db->pts_dev, pts, pbytes, hipMemcpyHostToDevice)) != hipSuccess) return fail(e, "hipMemcpy(points)");
    if ((e = hipMemcpy(db->off_dev, db->off.data(), db->off.size() * sizeof(int), hipMemcpyHostToDevice)) != hipSuccess) return fail(e, "hipMemcpy(offsets)");
    *out = db;
    return SUO_OK;
}

extern "C" void suo_mesh_db_destroy(void* h) {
    MeshDb* db = (MeshDb*)h;
    if (!db) return;
    if (db->pts_dev) (void)hipFree(db->pts_dev);
    if (db->off_dev) (void)hipFree(db->off_dev);
    if (db->scratch_dev) (void)hipFree(db->scratch_dev);
    if (db->scratch_host) (void)hipHostFree(db->scratch_host);
    if (db->stream) (void)hipStreamDestroy(db->stream);
    delete db;
}

extern "C" int suo_pose_errors(void* h, int n, const int* model_index, const float* T_pred, const float* T_gt, float* add, float* adds) {
    MeshDb* db = (MeshDb*)h;
    if (!db || n < 0 || (n > 0 && (!model_index || !T_pred || !T_gt || !add || !adds))) { suo_set_error("suo_pose_errors: bad argument"); return SUO_ERR_ARG; }
    if (n == 0) return SUO_OK;
    int pmax = 0;
    for (int i = 0; i < n; ++i) {
        if (model_index[i] < 0 || model_index[i] >= db->n_models) { suo_set_error("suo_pose_errors: model_index[%d]=%d out of range", i, model_index[i]); return SUO_ERR_ARG; }
        pmax = std::max(pmax, db->off[model_index[i] + 1] - db->off[model_index[i]]);
    }
    std::lock_guard<std::mutex> lk(db->mu);
    const int stride = (pmax + 63) & ~63;
    // staged block: model[n] | Tp[n][12] | Tg[n][12]   then device-only: mind2[n][stride] | out[n][2]
    const size_t o_model = 0, o_tp = (n * sizeof(int) + 15) & ~(size_t)15, o_tg = o_tp + (size_t)n * 48, staged = o_tg + (size_t)n * 48;
    const size_t o_out = staged, o_min = (o_out + (size_t)n * 8 + 255) & ~(size_t)255, total = o_min + (size_t)n * stride * 4;
    int rc = ensure_scratch(db, total);
    if (rc) return rc;
    memcpy(db->scratch_host + o_model, model_index, n * sizeof(int));
    memcpy(db->scratch_host + o_tp, T_pred, (size_t)n * 48);
    memcpy(db->scratch_host + o_tg, T_gt, (size_t)n * 48);
    SUO_HIP_CHECK(hipMemcpyAsync(db->scratch_dev, db->scratch_host, staged, hipMemcpyHostToDevice, db->stream));
    EvalArgs a;
    a.pts = db->pts_dev; a.off = db->off_dev;
    a.model = (const int*)(db->scratch_dev + o_model);
    a.Tp = (const float*)(db->scratch_dev + o_tp);
    a.Tg = (const float*)(db->scratch_dev + o_tg);
    a.out = (float*)(db->scratch_dev + o_out);
    a.mind2 = (unsigned*)(db->scratch_dev + o_min);
    a.stride = stride;
    const int gt_tiles = (pmax + EV_BLOCK * EV_G - 1) / (EV_BLOCK * EV_G);
    // >= ~2048 blocks where the problem allows it (256 CUs x 8 resident blocks), never finer than one LDS tile
    int splits = (2048 + gt_tiles * n - 1) / (gt_tiles * n);
    splits = std::max(1, std::min(splits, (pmax + EV_TILE - 1) / EV_TILE));
    a.splits = splits;
    const int total_min = n * stride;
    hipLaunchKernelGGL(eval_init_kernel, dim3((total_min + EV_BLOCK - 1) / EV_BLOCK), dim3(EV_BLOCK), 0, db->stream, a, total_min);
    hipLaunchKernelGGL(eval_pairs_kernel, dim3(gt_tiles, splits, n), dim3(EV_BLOCK), 0, db->stream, a);
    hipLaunchKernelGGL(eval_finalize_kernel, dim3(n), dim3(EV_BLOCK), 0, db->stream, a);
    SUO_HIP_CHECK(hipGetLastError());
    SUO_HIP_CHECK(hipMemcpyAsync(db->scratch_host + o_out, db->scratch_dev + o_out, (size_t)n * 8, hipMemcpyDeviceToHost, db->stream));
    SUO_HIP_CHECK(hipStreamSynchronize(db->stream));
    const float* o = (const float*)(db->scratch_host + o_out);
    for (int i = 0; i < n; ++i) { add[i] = o[2 * i]; adds[i] = o[2 * i + 1]; }
    return SUO_OK;
}
