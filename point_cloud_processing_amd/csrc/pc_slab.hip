// Slabs: one rank of a multi-GPU job indexes only ITS part of the cloud.
//
// With a replicated index every rank sorts all n points, which bounds the speed-up of a job whose search is
// already sharded.  A slab is the set of points whose z lies in [own_lo - halo, own_hi + halo): the rank
// indexes that subset alone (pc_grid_build on the gathered rows), answers the queries with z in
// [own_lo, own_hi) and checks every answer against the two open faces of the slab: a result whose k-th
// distance is below the distance to the nearest open face cannot have missed a point outside the subset
// (such a point lies beyond the face).  Results are exact, not approximate: the caller repeats the queries
// that fail the check on a wider slab or on a full index.  z is the slowest digit of the cell keys, so the
// queries of a z interval are one range of sorted positions (pc_grid_axis_range) for the *_compact searches.
// Subset rows keep the cloud's order, so "lower index first" between equal distances picks the same point.
#include <algorithm>

#include "pc_knn_dev.h"

namespace {

constexpr int HIST_MAX_BINS = 4096;

// axis 0..2: one histogram; axis < 0: one per axis (3 * nbins counters), from one pass over the cloud
template <typename T>
__global__ __launch_bounds__(256) void k_axis_hist(const T *__restrict__ xyz, int64_t n, int axis, double lo0, double lo1,
                                                   double lo2, double iw0, double iw1, double iw2, int nbins,
                                                   uint32_t *__restrict__ hist) {
    __shared__ uint32_t h[3 * HIST_MAX_BINS];
    const int na = axis < 0 ? 3 : 1;
    for (int b = threadIdx.x; b < na * nbins; b += 256) h[b] = 0u;
    __syncthreads();
    const double lo[3] = {lo0, lo1, lo2}, iw[3] = {iw0, iw1, iw2};
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        for (int a = 0; a < na; a++) {
            const int ax = axis < 0 ? a : axis;
            const double t = floor(((double)xyz[3 * i + ax] - lo[ax]) * iw[ax]);
            const int b = !(t > 0.0) ? 0 : (t >= (double)nbins ? nbins - 1 : (int)t);
            atomicAdd(&h[a * nbins + b], 1u);
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < na * nbins; b += 256)
        if (h[b]) atomicAdd(&hist[b], h[b]);
}

template <typename T>
__global__ __launch_bounds__(256) void k_slab_flags(const T *__restrict__ xyz, int64_t n, int axis, double lo, double hi,
                                                    uint8_t *__restrict__ flags) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double v = (double)xyz[3 * i + axis];
    flags[i] = (v >= lo && v < hi) ? 1 : 0;
}

// first sorted positions of z layers la and lb + 1, where la / lb are the layers of lo / hi
__global__ void k_axis_range(CellIndex ci, GridParams gp, double lo, double hi, int64_t *__restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    auto layer = [&](double z) {
        const double t = floor((z - gp.o[2]) * gp.s);  // (the expression behind the cell keys)
        return !(t > 0.0) ? 0 : (t >= (double)gp.dim[2] ? gp.dim[2] - 1 : (int)t);
    };
    const int la = layer(lo), lb = layer(hi);
    const int64_t per_layer = (int64_t)gp.dim[1] * gp.dim[0];
    out[0] = (int64_t)cell_begin(ci, (int64_t)la * per_layer);
    out[1] = lb + 1 >= gp.dim[2] ? (int64_t)ci.n : (int64_t)cell_begin(ci, (int64_t)(lb + 1) * per_layer);
}

// neighbour indices of the subset -> indices of the whole cloud, element by element (coalesced; the rows of
// queries outside the rank's interval are translated too, which harms nothing)
__global__ __launch_bounds__(256) void k_slab_translate(int32_t *__restrict__ idx, int64_t count,
                                                        const int32_t *__restrict__ gidx) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= count) return;
    const int32_t j = idx[e];
    if (j >= 0) idx[e] = gidx[j];
}

template <typename REC>
__global__ __launch_bounds__(256) void k_relabel(REC *__restrict__ rec, int32_t *__restrict__ perm, int64_t n,
                                                 const int32_t *__restrict__ labels) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int32_t g = labels[perm[i]];
    perm[i] = g;
    rec[i].idx = g;
}

template <typename T, typename TD>
__global__ __launch_bounds__(256) void k_slab_finish(const T *__restrict__ sub_xyz, const int32_t *__restrict__ gidx,
                                                     int64_t m_rows, int k, double radius2,
                                                     const TD *__restrict__ d2, const int32_t *__restrict__ cnt,
                                                     int32_t *__restrict__ owned, double own_lo, double own_hi,
                                                     double open_lo, double open_hi, const double *__restrict__ avg_rows,
                                                     double *__restrict__ avg_full, uint8_t *__restrict__ unc_flags,
                                                     unsigned long long *__restrict__ counts) {
    // (grid-stride rows, counts summed per thread, then per block: two atomics per block -- one per wave on the
    // same two words took longer than the rest of the kernel)
    uint32_t n_own = 0, n_unc = 0;
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < m_rows; r += (int64_t)gridDim.x * 256) {
        const int32_t local = owned[r];  // (row of sub_xyz; with gidx == NULL also the index in the whole cloud)
        const double z = (double)sub_xyz[3 * (size_t)local + 2];
        if (!(z >= own_lo && z < own_hi)) {
            owned[r] = -1;
            if (unc_flags) unc_flags[r] = 0;
            continue;
        }
        const int32_t g = gidx ? gidx[local] : local;
        owned[r] = g;
        n_own++;
        const int c = cnt ? cnt[r] : k;
        // no point outside the subset is closer than the nearer open face
        const double margin = fmin(z - open_lo, open_hi - z);  // (+inf where both faces are the cloud's own ends)
        const double lim = margin < INFINITY ? margin * margin * (1.0 - 1e-9) : INFINITY;
        bool uncertain;
        if (c >= k) {
            // (fp32 rows carry the fp64 distance rounded to nearest: widened by what that can hide)
            const double dk = (double)d2[(size_t)r * k + (k - 1)] * (sizeof(TD) == 4 ? 1.0 + 1e-6 : 1.0);
            uncertain = !(dk <= lim);
        } else {
            // fewer than k found: final only if nothing can hide beyond the faces within the search radius
            uncertain = !(lim == INFINITY || (radius2 >= 0.0 && radius2 <= lim));
        }
        n_unc += uncertain ? 1u : 0u;
        if (unc_flags) unc_flags[r] = uncertain ? 1 : 0;
        if (avg_full && avg_rows) avg_full[g] = avg_rows[r];
    }
    __shared__ uint32_t part[2];
    if (threadIdx.x < 2) part[threadIdx.x] = 0u;
    __syncthreads();
    if (n_own) atomicAdd(&part[0], n_own);
    if (n_unc) atomicAdd(&part[1], n_unc);
    __syncthreads();
    if (threadIdx.x < 2 && part[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)part[threadIdx.x]);
}

// The ranks' answers to the same queries merged: rank w's row of query q holds its cnt nearest points in ascending
// (d2, index) order; the union's k nearest in the same order are the exact answer when the ranks' subsets cover the
// cloud.  A point indexed by two ranks (halos overlap) appears twice with the same key: taken once.
__global__ __launch_bounds__(64) void k_merge_rows(int world, int64_t nq, int k, const int32_t *__restrict__ idx_all,
                                                   const double *__restrict__ d2_all, const int32_t *__restrict__ cnt_all,
                                                   int32_t *__restrict__ out_idx, double *__restrict__ out_d2,
                                                   int32_t *__restrict__ out_cnt) {
    const int64_t q = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (q >= nq) return;
    int cur[PC_MAX_K];  // (one cursor per rank: world <= PC_MAX_K)
    for (int w = 0; w < world; w++) cur[w] = 0;
    int got = 0;
    int32_t last = -1;
    while (got < k) {
        int best = -1;
        double bd = INFINITY;
        int32_t bi = 0x7fffffff;
        for (int w = 0; w < world; w++) {
            if (cur[w] >= cnt_all[(int64_t)w * nq + q]) continue;
            const int64_t at = ((int64_t)w * nq + q) * k + cur[w];
            const double d = d2_all[at];
            const int32_t i = idx_all[at];
            if (best < 0 || d < bd || (d == bd && i < bi)) {
                best = w;
                bd = d;
                bi = i;
            }
        }
        if (best < 0) break;
        cur[best]++;
        if (got > 0 && bi == last) continue;  // the same point from another rank's halo
        out_idx[q * k + got] = bi;
        out_d2[q * k + got] = bd;
        last = bi;
        got++;
    }
    for (int s = got; s < k; s++) {
        out_idx[q * k + s] = -1;
        out_d2[q * k + s] = INFINITY;
    }
    out_cnt[q] = got;
}

// mean neighbour distance and normal of listed queries from their finished rows (the searches' own consumers,
// fed from the rows instead of a candidate block)
template <typename T>
__global__ __launch_bounds__(64) void k_rows_consumers(ScanArgs A, const T *__restrict__ xyz, const int32_t *__restrict__ qidx,
                                                       int64_t nq, const int32_t *__restrict__ idx,
                                                       const int32_t *__restrict__ cnt) {
    const int64_t q = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (q >= nq) return;
    const int32_t me = qidx[q];
    const double qx = (double)xyz[3 * (size_t)me], qy = (double)xyz[3 * (size_t)me + 1], qz = (double)xyz[3 * (size_t)me + 2];
    const int k = A.k;
    emit_result<M_ALL>(A, q, cnt[q], [&](int s, double &d, double &x, double &y, double &z, int32_t &id) {
        id = idx[q * k + s];
        x = (double)xyz[3 * (size_t)id];
        y = (double)xyz[3 * (size_t)id + 1];
        z = (double)xyz[3 * (size_t)id + 2];
        const double dx = qx - x, dy = qy - y, dz = qz - z;
        d = dx * dx;
        d += dy * dy;
        d += dz * dz;
    });
}

}  // namespace

extern "C" int pc_axis_histogram(pc_ctx *ctx, const void *xyz_dev, int dtype, int64_t n, int axis, const double *lo,
                                 const double *hi, int nbins, uint32_t *out_counts, void *stream) {
    PcLock lock(ctx, stream);
    if (!ctx || n < 0 || (n > 0 && !xyz_dev) || axis > 2 || nbins < 1 || nbins > HIST_MAX_BINS || !out_counts || !lo || !hi ||
        (dtype != PC_F32 && dtype != PC_F64))
        return pc_fail(ctx, PC_BAD_ARG, "pc_axis_histogram");
    const int na = axis < 0 ? 3 : 1;
    double l[3] = {0, 0, 0}, iw[3] = {0, 0, 0};
    for (int a = 0; a < 3; a++) {
        if (axis >= 0 && a != axis) continue;
        const int j = axis < 0 ? a : 0;  // (one axis: lo[0] / hi[0]; all axes: lo[a] / hi[a])
        if (!(hi[j] >= lo[j])) return pc_fail(ctx, PC_BAD_ARG, "pc_axis_histogram: hi < lo");
        l[a] = lo[j];
        iw[a] = hi[j] > lo[j] ? (double)nbins / (hi[j] - lo[j]) : 0.0;
    }
    PC_HIP(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    for (int b = 0; b < na * nbins; b++) out_counts[b] = 0u;
    if (n == 0) return PC_OK;
    const size_t bytes = sizeof(uint32_t) * (size_t)(na * nbins);
    Scratch h(ctx, bytes);
    if (!h.ok()) return pc_fail(ctx, PC_NOMEM, "histogram");
    PC_HIP(hipMemsetAsync(h.p, 0, bytes, st));
    const unsigned nb = (unsigned)std::min<int64_t>((n + 255) / 256, 2048);
    if (dtype == PC_F32)
        hipLaunchKernelGGL(k_axis_hist<float>, dim3(nb), dim3(256), 0, st, (const float *)xyz_dev, n, axis, l[0], l[1], l[2],
                           iw[0], iw[1], iw[2], nbins, h.as<uint32_t>());
    else
        hipLaunchKernelGGL(k_axis_hist<double>, dim3(nb), dim3(256), 0, st, (const double *)xyz_dev, n, axis, l[0], l[1], l[2],
                           iw[0], iw[1], iw[2], nbins, h.as<uint32_t>());
    PC_HIP(hipGetLastError());
    PC_HIP(hipMemcpyAsync(out_counts, h.p, bytes, hipMemcpyDeviceToHost, st));
    PC_HIP(hipStreamSynchronize(st));
    return PC_OK;
}

extern "C" int pc_select_slab(pc_ctx *ctx, const void *xyz_dev, int dtype, int64_t n, int axis, double lo, double hi,
                              int32_t *out_index_dev, int64_t *m, void *stream) {
    PcLock lock(ctx, stream);
    if (!ctx || n < 0 || n > INT32_MAX || (n > 0 && (!xyz_dev || !out_index_dev)) || axis < 0 || axis > 2 || !m ||
        (dtype != PC_F32 && dtype != PC_F64))
        return pc_fail(ctx, PC_BAD_ARG, "pc_select_slab");
    PC_HIP(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    *m = 0;
    if (n == 0) return PC_OK;
    Scratch flags(ctx, (size_t)n);
    if (!flags.ok()) return pc_fail(ctx, PC_NOMEM, "slab flags");
    const unsigned nb = (unsigned)((n + 255) / 256);
    if (dtype == PC_F32)
        hipLaunchKernelGGL(k_slab_flags<float>, dim3(nb), dim3(256), 0, st, (const float *)xyz_dev, n, axis, lo, hi,
                           flags.as<uint8_t>());
    else
        hipLaunchKernelGGL(k_slab_flags<double>, dim3(nb), dim3(256), 0, st, (const double *)xyz_dev, n, axis, lo, hi,
                           flags.as<uint8_t>());
    PC_HIP(hipGetLastError());
    return pc_compact_flags(ctx, flags.as<uint8_t>(), n, out_index_dev, m, st);  // (ascending: the cloud's order is kept)
}

extern "C" int pc_grid_axis_range(pc_ctx *ctx, const pc_grid *grid, double lo, double hi, int64_t *s_begin, int64_t *s_end,
                                  void *stream) {
    PcLock lock(ctx, stream);
    if (!ctx || !grid || !s_begin || !s_end || !(hi >= lo)) return pc_fail(ctx, PC_BAD_ARG, "pc_grid_axis_range");
    PC_HIP(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    *s_begin = *s_end = 0;
    if (grid->n == 0) return PC_OK;
    Scratch out(ctx, 2 * sizeof(int64_t));
    if (!out.ok()) return pc_fail(ctx, PC_NOMEM, "range scratch");
    hipLaunchKernelGGL(k_axis_range, dim3(1), dim3(64), 0, st, grid->index(), grid->gp, lo, hi, out.as<int64_t>());
    PC_HIP(hipGetLastError());
    int64_t *hp = reinterpret_cast<int64_t *>(ctx->pinned);
    PC_HIP(hipMemcpyAsync(hp, out.p, 2 * sizeof(int64_t), hipMemcpyDeviceToHost, st));
    PC_HIP(hipStreamSynchronize(st));
    *s_begin = hp[0];
    *s_end = hp[1];
    return PC_OK;
}

extern "C" int pc_slab_finish(pc_ctx *ctx, const void *sub_xyz_dev, int dtype, const int32_t *sub_index_dev, int64_t m_rows,
                              int k, double radius, int32_t *idx_dev, const void *d2_dev, int d2_dtype,
                              const int32_t *cnt_dev, int32_t *owned_dev, double own_lo, double own_hi, double open_lo,
                              double open_hi, const double *avg_rows_dev, double *avg_full_dev,
                              int32_t *out_uncertain_rows_dev, int64_t *n_own, int64_t *n_uncertain, void *stream) {
    PcLock lock(ctx, stream);
    if (!ctx || m_rows < 0 || k < 1 || k > PC_MAX_K || !n_own || !n_uncertain ||
        (m_rows > 0 && (!sub_xyz_dev || !d2_dev || !owned_dev)) || (dtype != PC_F32 && dtype != PC_F64) ||
        (d2_dtype != PC_F32 && d2_dtype != PC_F64))
        return pc_fail(ctx, PC_BAD_ARG, "pc_slab_finish");
    PC_HIP(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    *n_own = *n_uncertain = 0;
    if (m_rows == 0) return PC_OK;
    Scratch counts(ctx, 2 * sizeof(unsigned long long)), uflags(ctx, out_uncertain_rows_dev ? (size_t)m_rows : 0);
    if (!counts.ok() || !uflags.ok()) return pc_fail(ctx, PC_NOMEM, "slab counts");
    uint8_t *uf = out_uncertain_rows_dev ? uflags.as<uint8_t>() : nullptr;
    PC_HIP(hipMemsetAsync(counts.p, 0, 2 * sizeof(unsigned long long), st));
    const unsigned nb = (unsigned)std::min<int64_t>((m_rows + 255) / 256, 2048);
    const double r2 = radius >= 0 ? radius * radius : -1.0;
    unsigned long long *c = counts.as<unsigned long long>();
    if (idx_dev && sub_index_dev)
        hipLaunchKernelGGL(k_slab_translate, dim3((unsigned)((m_rows * k + 255) / 256)), dim3(256), 0, st, idx_dev, m_rows * k,
                           sub_index_dev);
#define PC_SLAB_FINISH(T, TD)                                                                                             \
    hipLaunchKernelGGL((k_slab_finish<T, TD>), dim3(nb), dim3(256), 0, st, (const T *)sub_xyz_dev, sub_index_dev, m_rows, k, \
                       r2, (const TD *)d2_dev, cnt_dev, owned_dev, own_lo, own_hi, open_lo, open_hi, avg_rows_dev, \
                       avg_full_dev, uf, c)
    if (dtype == PC_F32 && d2_dtype == PC_F32)
        PC_SLAB_FINISH(float, float);
    else if (dtype == PC_F32)
        PC_SLAB_FINISH(float, double);
    else if (d2_dtype == PC_F32)
        PC_SLAB_FINISH(double, float);
    else
        PC_SLAB_FINISH(double, double);
#undef PC_SLAB_FINISH
    PC_HIP(hipGetLastError());
    unsigned long long *hp = reinterpret_cast<unsigned long long *>(ctx->pinned);
    PC_HIP(hipMemcpyAsync(hp, counts.p, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    PC_HIP(hipStreamSynchronize(st));
    *n_own = (int64_t)hp[0];
    *n_uncertain = (int64_t)hp[1];
    if (uf && *n_uncertain > 0) {  // the rows to repeat, ascending
        int64_t listed = 0;
        PC_TRY(pc_compact_flags(ctx, uf, m_rows, out_uncertain_rows_dev, &listed, st));
    }
    return PC_OK;
}

extern "C" int pc_grid_relabel(pc_ctx *ctx, pc_grid *grid, const int32_t *labels_dev, int64_t index_space, void *stream) {
    PcLock lock(ctx, stream);
    if (!ctx || !grid || (grid->n > 0 && !labels_dev) || index_space < grid->n || index_space > INT32_MAX || grid->finer ||
        grid->index_space > 0)
        return pc_fail(ctx, PC_BAD_ARG, "pc_grid_relabel");
    PC_HIP(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    for (pc_grid *l = grid; l; l = l->coarse) {
        l->index_space = index_space;
        if (l->n == 0) continue;
        const unsigned nb = (unsigned)((l->n + 255) / 256);
        if (l->dtype == PC_F32)
            hipLaunchKernelGGL(k_relabel<Rec32>, dim3(nb), dim3(256), 0, st, (Rec32 *)l->rec, l->perm, l->n, labels_dev);
        else
            hipLaunchKernelGGL(k_relabel<Rec64>, dim3(nb), dim3(256), 0, st, (Rec64 *)l->rec, l->perm, l->n, labels_dev);
    }
    PC_HIP(hipGetLastError());
    return PC_OK;
}

extern "C" int pc_merge_rows(pc_ctx *ctx, int world, int64_t nq, int k, const int32_t *idx_all_dev, const double *d2_all_dev,
                             const int32_t *cnt_all_dev, int32_t *out_idx_dev, double *out_d2_dev, int32_t *out_cnt_dev,
                             void *stream) {
    PcLock lock(ctx, stream);
    if (!ctx || world < 1 || world > PC_MAX_K || nq < 0 || k < 1 || k > PC_MAX_K ||
        (nq > 0 && (!idx_all_dev || !d2_all_dev || !cnt_all_dev || !out_idx_dev || !out_d2_dev || !out_cnt_dev)))
        return pc_fail(ctx, PC_BAD_ARG, "pc_merge_rows");
    PC_HIP(hipSetDevice(ctx->device));
    if (nq == 0) return PC_OK;
    hipLaunchKernelGGL(k_merge_rows, dim3((unsigned)((nq + 63) / 64)), dim3(64), 0, (hipStream_t)stream, world, nq, k, idx_all_dev,
                       d2_all_dev, cnt_all_dev, out_idx_dev, out_d2_dev, out_cnt_dev);
    PC_HIP(hipGetLastError());
    return PC_OK;
}

extern "C" int pc_rows_consumers(pc_ctx *ctx, const void *xyz_dev, int dtype, const int32_t *query_index_dev, int64_t nq, int k,
                                 const int32_t *idx_dev, const int32_t *cnt_dev, double *out_avg_dev, double *out_normals_dev,
                                 void *stream) {
    PcLock lock(ctx, stream);
    if (!ctx || nq < 0 || k < 1 || k > PC_MAX_K || (dtype != PC_F32 && dtype != PC_F64) ||
        (nq > 0 && (!xyz_dev || !query_index_dev || !idx_dev || !cnt_dev)))
        return pc_fail(ctx, PC_BAD_ARG, "pc_rows_consumers");
    PC_HIP(hipSetDevice(ctx->device));
    if (nq == 0 || (!out_avg_dev && !out_normals_dev)) return PC_OK;
    ScanArgs A{};
    A.k = k;
    A.out_f64 = out_avg_dev;
    A.out_nrm = out_normals_dev;
    A.compact_base = -1;
    const unsigned nb = (unsigned)((nq + 63) / 64);
    if (dtype == PC_F32)
        hipLaunchKernelGGL(k_rows_consumers<float>, dim3(nb), dim3(64), 0, (hipStream_t)stream, A, (const float *)xyz_dev,
                           query_index_dev, nq, idx_dev, cnt_dev);
    else
        hipLaunchKernelGGL(k_rows_consumers<double>, dim3(nb), dim3(64), 0, (hipStream_t)stream, A, (const double *)xyz_dev,
                           query_index_dev, nq, idx_dev, cnt_dev);
    PC_HIP(hipGetLastError());
    return PC_OK;
}
