// Point-to-plane ICP pieces around the k = 1 correspondence search (the GUI advertises ICP at
// app/UI/main_window.py:58 and 180-194 but the reference has no solver behind it; the arithmetic follows
// Open3D's TransformationEstimationPointToPlane::ComputeTransformation):
//   pc_icp_point_to_plane_terms  the 6x6 normal equations of one linearised step, reduced on the device
//   pc_transform                 p <- R p + t over a cloud, in place
// Sums have a fixed shape (each thread a fixed stride of the correspondences, block partials added by one
// wave in a fixed order), so a given input always gives the same bits.
#include "pc_common.h"

namespace {

constexpr int ICP_TERMS = 29;   // 21 (upper triangle of J^T J) + 6 (J^T r) + sum of d2 + count
constexpr int ICP_BLOCKS = 512;

template <typename T>
__global__ __launch_bounds__(256) void k_icp_terms(const double *__restrict__ src, int64_t m, const T *__restrict__ tgt,
                                                    const double *__restrict__ tn, const int32_t *__restrict__ corr,
                                                    const void *__restrict__ d2v, int d2_f64, double max_d2,
                                                    double *__restrict__ partial) {
    double acc[ICP_TERMS];
#pragma unroll
    for (int i = 0; i < ICP_TERMS; i++) acc[i] = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (int64_t)gridDim.x * 256) {
        const int32_t j = corr[i];
        if (j < 0) continue;
        const double d2 = d2_f64 ? reinterpret_cast<const double *>(d2v)[i] : (double)reinterpret_cast<const float *>(d2v)[i];
        if (!(d2 < max_d2)) continue;
        const double s[3] = {src[3 * i], src[3 * i + 1], src[3 * i + 2]};
        const double t[3] = {(double)tgt[3 * (int64_t)j], (double)tgt[3 * (int64_t)j + 1], (double)tgt[3 * (int64_t)j + 2]};
        const double n[3] = {tn[3 * (int64_t)j], tn[3 * (int64_t)j + 1], tn[3 * (int64_t)j + 2]};
        const double r = ((s[0] - t[0]) * n[0] + (s[1] - t[1]) * n[1]) + (s[2] - t[2]) * n[2];
        const double J[6] = {s[1] * n[2] - s[2] * n[1], s[2] * n[0] - s[0] * n[2], s[0] * n[1] - s[1] * n[0], n[0], n[1], n[2]};
        int q = 0;
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int b = a; b < 6; b++) acc[q++] += J[a] * J[b];
#pragma unroll
        for (int a = 0; a < 6; a++) acc[21 + a] += J[a] * r;
        acc[27] += d2;
        acc[28] += 1.0;
    }
    __shared__ double red[4][ICP_TERMS];
#pragma unroll
    for (int i = 0; i < ICP_TERMS; i++) {
        double v = acc[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < ICP_TERMS)
        partial[(int64_t)blockIdx.x * ICP_TERMS + threadIdx.x] =
            ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

// one wave per term: lane j adds the block partials j, j + 64, ... in that order, then a fixed shuffle tree
__global__ __launch_bounds__(64) void k_icp_final(const double *__restrict__ partial, int nblocks, double *__restrict__ out) {
    const int term = blockIdx.x;
    double v = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 64) v += partial[(int64_t)b * ICP_TERMS + term];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if (threadIdx.x == 0) out[term] = v;
}

struct Rigid {
    double m[12];  // rows of [R | t]
};

__global__ __launch_bounds__(256) void k_transform(double *__restrict__ xyz, int64_t n, Rigid T) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    // (the order of Eigen's 4x4 times (x, y, z, 1) product, no fused multiply-add)
    xyz[3 * i] = ((T.m[0] * x + T.m[1] * y) + T.m[2] * z) + T.m[3];
    xyz[3 * i + 1] = ((T.m[4] * x + T.m[5] * y) + T.m[6] * z) + T.m[7];
    xyz[3 * i + 2] = ((T.m[8] * x + T.m[9] * y) + T.m[10] * z) + T.m[11];
}

}  // namespace

extern "C" {

int pc_icp_point_to_plane_terms(pc_ctx *ctx, const double *src_dev, int64_t m, const void *tgt_xyz_dev, int tgt_dtype,
                                const double *tgt_normals_dev, const int32_t *corr_idx_dev, const void *corr_d2_dev,
                                int d2_dtype, double max_d2, double out_terms[29], void *stream) {
    PcLock lock(ctx, stream);  // calls on one context take turns (its block pool and read-back buffer are shared)
    if (!ctx || !out_terms || m < 0 || (tgt_dtype != PC_F32 && tgt_dtype != PC_F64) ||
        (d2_dtype != PC_F32 && d2_dtype != PC_F64) ||
        (m > 0 && (!src_dev || !tgt_xyz_dev || !tgt_normals_dev || !corr_idx_dev || !corr_d2_dev)))
        return pc_fail(ctx, PC_BAD_ARG, "pc_icp_point_to_plane_terms");
    PC_HIP(hipSetDevice(ctx->device));
    for (int i = 0; i < ICP_TERMS; i++) out_terms[i] = 0.0;
    if (m == 0) return PC_OK;
    hipStream_t st = (hipStream_t)stream;
    const int nb = (int)std::min<int64_t>(ICP_BLOCKS, (m + 255) / 256);
    Scratch partial(ctx, sizeof(double) * ICP_TERMS * (size_t)(nb + 1));
    if (!partial.ok()) return pc_fail(ctx, PC_NOMEM, "pc_icp_point_to_plane_terms scratch");
    double *fin = partial.as<double>() + (size_t)nb * ICP_TERMS;
    if (tgt_dtype == PC_F32)
        hipLaunchKernelGGL(k_icp_terms<float>, dim3(nb), dim3(256), 0, st, src_dev, m, (const float *)tgt_xyz_dev,
                           tgt_normals_dev, corr_idx_dev, corr_d2_dev, d2_dtype == PC_F64, max_d2, partial.as<double>());
    else
        hipLaunchKernelGGL(k_icp_terms<double>, dim3(nb), dim3(256), 0, st, src_dev, m, (const double *)tgt_xyz_dev,
                           tgt_normals_dev, corr_idx_dev, corr_d2_dev, d2_dtype == PC_F64, max_d2, partial.as<double>());
    hipLaunchKernelGGL(k_icp_final, dim3(ICP_TERMS), dim3(64), 0, st, partial.as<double>(), nb, fin);
    PC_HIP(hipGetLastError());
    double *host = reinterpret_cast<double *>(ctx->pinned);
    PC_HIP(hipMemcpyAsync(host, fin, sizeof(double) * ICP_TERMS, hipMemcpyDeviceToHost, st));
    PC_HIP(hipStreamSynchronize(st));
    for (int i = 0; i < ICP_TERMS; i++) out_terms[i] = host[i];
    return PC_OK;
}

int pc_transform(pc_ctx *ctx, double *xyz_dev, int64_t n, const double T[16], void *stream) {
    PcLock lock(ctx, stream);  // calls on one context take turns (its block pool and read-back buffer are shared)
    if (!ctx || !T || n < 0 || (n > 0 && !xyz_dev)) return pc_fail(ctx, PC_BAD_ARG, "pc_transform");
    PC_HIP(hipSetDevice(ctx->device));
    if (n == 0) return PC_OK;
    Rigid R;
    for (int i = 0; i < 12; i++) R.m[i] = T[i];
    hipLaunchKernelGGL(k_transform, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, xyz_dev, n, R);
    PC_HIP(hipGetLastError());
    return PC_OK;
}

}  // extern "C"
