// Exact k-nearest-neighbour scan over the uniform grid and the consumers fused onto it:
// statistical-outlier mean distance, PCA normals, nearest-neighbour distance, radius count.
//
// One query per lane.  A query walks the shells of cells around its own cell (rows of cells
// are contiguous point ranges because x is the fastest key digit), measures every candidate
// in fp64 in the reference's operation order, and stops once the k-th best squared distance
// is certified by the distance to the nearest face of the scanned block.  The per-lane
// neighbour list lives in LDS as list[slot][lane] (conflict-free when lanes touch the same
// slot).  All kernels are HBM/LDS/VALU work; nothing here is a dense contraction.
#include <math.h>

#include <algorithm>

#include "pc_common.h"

namespace {

enum Mode { M_KNN = 0, M_SOR = 1, M_NORMALS = 2, M_NNDIST = 3, M_COUNT = 4 };

struct ScanArgs {
    GridParams gp;
    const void *rec;
    const uint32_t *cell_start;
    const void *queries;  // NULL: self-query over sorted positions
    int64_t s_begin;      // first sorted position (self) / first query row (external)
    int64_t nq;
    int k;
    double radius2;  // < 0: none
    // outputs
    int32_t *out_idx;
    void *out_d2;
    int d2_f64;
    int32_t *out_cnt;
    double *out_f64;        // SOR avg / NN distance / normals
    const double *prior;    // normals only
};

template <typename REC>
struct RecTraits;
template <>
struct RecTraits<Rec32> {
    __device__ static inline void load(const Rec32 *p, double &x, double &y, double &z, int32_t &idx) {
        const float4 v = *reinterpret_cast<const float4 *>(p);
        x = (double)v.x;
        y = (double)v.y;
        z = (double)v.z;
        idx = __float_as_int(v.w);
    }
};
template <>
struct RecTraits<Rec64> {
    __device__ static inline void load(const Rec64 *p, double &x, double &y, double &z, int32_t &idx) {
        const double2 a = *reinterpret_cast<const double2 *>(p);
        const double2 b = *(reinterpret_cast<const double2 *>(p) + 1);
        x = a.x;
        y = a.y;
        z = b.x;
        idx = (int32_t)__double_as_longlong(b.y);
    }
};

// ---- closed-form symmetric 3x3 eigen-solver (smallest-eigenvalue eigenvector) ---------------------
// Same operation order as Open3D's FastEigen3x3 (after Eberly's RobustEigenSymmetric3x3).
__device__ inline void cross3(const double a[3], const double b[3], double o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ inline double dot3(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

__device__ inline void eigvec0(const double A[6], double ev, double o[3]) {
    double r0[3] = {A[0] - ev, A[1], A[2]};
    double r1[3] = {A[1], A[3] - ev, A[4]};
    double r2[3] = {A[2], A[4], A[5] - ev};
    double c01[3], c02[3], c12[3];
    cross3(r0, r1, c01);
    cross3(r0, r2, c02);
    cross3(r1, r2, c12);
    double d0 = dot3(c01, c01), d1 = dot3(c02, c02), d2 = dot3(c12, c12);
    double dmax = d0;
    int imax = 0;
    if (d1 > dmax) {
        dmax = d1;
        imax = 1;
    }
    if (d2 > dmax) imax = 2;
    double s = sqrt(imax == 0 ? d0 : (imax == 1 ? d1 : d2));
#pragma unroll
    for (int a = 0; a < 3; a++) o[a] = (imax == 0 ? c01[a] : (imax == 1 ? c02[a] : c12[a])) / s;
}

__device__ inline void eigvec1(const double A[6], const double e0[3], double ev, double o[3]) {
    double U[3], V[3];
    if (fabs(e0[0]) > fabs(e0[1])) {
        double il = 1 / sqrt(e0[0] * e0[0] + e0[2] * e0[2]);
        U[0] = -e0[2] * il;
        U[1] = 0;
        U[2] = e0[0] * il;
    } else {
        double il = 1 / sqrt(e0[1] * e0[1] + e0[2] * e0[2]);
        U[0] = 0;
        U[1] = e0[2] * il;
        U[2] = -e0[1] * il;
    }
    cross3(e0, U, V);
    double AU[3] = {A[0] * U[0] + A[1] * U[1] + A[2] * U[2], A[1] * U[0] + A[3] * U[1] + A[4] * U[2],
                    A[2] * U[0] + A[4] * U[1] + A[5] * U[2]};
    double AV[3] = {A[0] * V[0] + A[1] * V[1] + A[2] * V[2], A[1] * V[0] + A[3] * V[1] + A[4] * V[2],
                    A[2] * V[0] + A[4] * V[1] + A[5] * V[2]};
    double m00 = U[0] * AU[0] + U[1] * AU[1] + U[2] * AU[2] - ev;
    double m01 = U[0] * AV[0] + U[1] * AV[1] + U[2] * AV[2];
    double m11 = V[0] * AV[0] + V[1] * AV[1] + V[2] * AV[2] - ev;
    double a00 = fabs(m00), a01 = fabs(m01), a11 = fabs(m11);
    if (a00 >= a11) {
        double mx = a00 > a01 ? a00 : a01;
        if (mx > 0) {
            if (a00 >= a01) {
                m01 /= m00;
                m00 = 1 / sqrt(1 + m01 * m01);
                m01 *= m00;
            } else {
                m00 /= m01;
                m01 = 1 / sqrt(1 + m00 * m00);
                m00 *= m01;
            }
#pragma unroll
            for (int a = 0; a < 3; a++) o[a] = m01 * U[a] - m00 * V[a];
        } else {
#pragma unroll
            for (int a = 0; a < 3; a++) o[a] = U[a];
        }
    } else {
        double mx = a11 > a01 ? a11 : a01;
        if (mx > 0) {
            if (a11 >= a01) {
                m01 /= m11;
                m11 = 1 / sqrt(1 + m01 * m01);
                m01 *= m11;
            } else {
                m11 /= m01;
                m01 = 1 / sqrt(1 + m11 * m11);
                m11 *= m01;
            }
#pragma unroll
            for (int a = 0; a < 3; a++) o[a] = m11 * U[a] - m01 * V[a];
        } else {
#pragma unroll
            for (int a = 0; a < 3; a++) o[a] = U[a];
        }
    }
}

// cov = {c00,c01,c02,c11,c12,c22}
__device__ inline void fast_eigen3x3(const double cov[6], double out[3]) {
    double A[6];
    double mc = cov[0];
#pragma unroll
    for (int i = 1; i < 6; i++)
        if (cov[i] > mc) mc = cov[i];
    if (mc == 0) {
        out[0] = out[1] = out[2] = 0;
        return;
    }
#pragma unroll
    for (int i = 0; i < 6; i++) A[i] = cov[i] / mc;
    double norm = A[1] * A[1] + A[2] * A[2] + A[4] * A[4];
    if (norm > 0) {
        double q = (A[0] + A[3] + A[5]) / 3;
        double b00 = A[0] - q, b11 = A[3] - q, b22 = A[5] - q;
        double p = sqrt((b00 * b00 + b11 * b11 + b22 * b22 + norm * 2) / 6);
        double c00 = b11 * b22 - A[4] * A[4];
        double c01 = A[1] * b22 - A[4] * A[2];
        double c02 = A[1] * A[4] - b11 * A[2];
        double det = (b00 * c00 - A[1] * c01 + A[2] * c02) / (p * p * p);
        double hd = det * 0.5;
        hd = fmin(fmax(hd, -1.0), 1.0);
        double angle = acos(hd) / 3.0;
        const double two_thirds_pi = 2.09439510239319549;
        double beta2 = cos(angle) * 2;
        double beta0 = cos(angle + two_thirds_pi) * 2;
        double beta1 = -(beta0 + beta2);
        double e0 = q + p * beta0, e1 = q + p * beta1, e2 = q + p * beta2;
        double v0[3], v1[3], v2[3];
        if (hd >= 0) {
            eigvec0(A, e2, v2);
            if (e2 < e0 && e2 < e1) {
                out[0] = v2[0], out[1] = v2[1], out[2] = v2[2];
                return;
            }
            eigvec1(A, v2, e1, v1);
            if (e1 < e0 && e1 < e2) {
                out[0] = v1[0], out[1] = v1[1], out[2] = v1[2];
                return;
            }
            cross3(v1, v2, out);
        } else {
            eigvec0(A, e0, v0);
            if (e0 < e1 && e0 < e2) {
                out[0] = v0[0], out[1] = v0[1], out[2] = v0[2];
                return;
            }
            eigvec1(A, v0, e1, v1);
            if (e1 < e0 && e1 < e2) {
                out[0] = v1[0], out[1] = v1[1], out[2] = v1[2];
                return;
            }
            cross3(v0, v1, out);
        }
    } else {
        double a0 = A[0] * mc, a1 = A[3] * mc, a2 = A[5] * mc;
        out[0] = out[1] = out[2] = 0;
        if (a0 < a1 && a0 < a2)
            out[0] = 1;
        else if (a1 < a0 && a1 < a2)
            out[1] = 1;
        else
            out[2] = 1;
    }
}

// ---- the scan -----------------------------------------------------------------------------------
template <typename REC, typename TQ, int MODE>
__global__ __launch_bounds__(PC_WAVE) void k_scan(ScanArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    const int k = A.k;
    double *Ld = reinterpret_cast<double *>(smem);                                  // [k][64]
    int32_t *Lp = reinterpret_cast<int32_t *>(smem + sizeof(double) * PC_WAVE * k);  // [k][64]

    const int64_t qi = (int64_t)blockIdx.x * PC_WAVE + lane;
    if (qi >= A.nq) return;  // no barriers below: lanes are independent
    const REC *rec = reinterpret_cast<const REC *>(A.rec);
    const GridParams &gp = A.gp;

    double qx, qy, qz;
    int64_t out_row;
    if (A.queries == nullptr) {
        int32_t id;
        RecTraits<REC>::load(rec + (A.s_begin + qi), qx, qy, qz, id);
        out_row = id;
    } else {
        const TQ *q = reinterpret_cast<const TQ *>(A.queries) + 3 * (A.s_begin + qi);
        qx = (double)q[0];
        qy = (double)q[1];
        qz = (double)q[2];
        out_row = A.s_begin + qi;
    }
    const double qv[3] = {qx, qy, qz};
    int c[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        double t = floor((qv[a] - gp.o[a]) * gp.s);
        c[a] = !(t > 0.0) ? 0 : (t >= (double)gp.dim[a] ? gp.dim[a] - 1 : (int)t);
    }
    const double eps = 1e-7 * gp.cell;
    const double r2lim = A.radius2;
    int cnt = 0;
    double worst = INFINITY;  // k-th best d2 once the list is full
    int32_t worst_idx = 0x7fffffff;

    for (int R = 0;; R++) {
        int lo[3], hi[3];
#pragma unroll
        for (int a = 0; a < 3; a++) {
            lo[a] = c[a] - R < 0 ? 0 : c[a] - R;
            hi[a] = c[a] + R >= gp.dim[a] ? gp.dim[a] - 1 : c[a] + R;
        }
        for (int z = lo[2]; z <= hi[2]; z++) {
            for (int y = lo[1]; y <= hi[1]; y++) {
                const int64_t row = ((int64_t)z * gp.dim[1] + y) * gp.dim[0];
                const bool full = (z - c[2] == R) || (c[2] - z == R) || (y - c[1] == R) || (c[1] - y == R);
                const int nseg = full ? 1 : (R == 0 ? 1 : 2);
                for (int s = 0; s < nseg; s++) {
                    int x0, x1;
                    if (full) {
                        x0 = lo[0];
                        x1 = hi[0];
                    } else {
                        x0 = x1 = (s == 0) ? c[0] - R : c[0] + R;
                        if (x0 < 0 || x0 >= gp.dim[0]) continue;
                    }
                    const uint32_t jb = A.cell_start[row + x0], je = A.cell_start[row + x1 + 1];
                    for (uint32_t j = jb; j < je; j++) {
                        double px, py, pz;
                        int32_t pid;
                        RecTraits<REC>::load(rec + j, px, py, pz, pid);
                        const double dx = qx - px, dy = qy - py, dz = qz - pz;
                        double d = dx * dx;
                        d += dy * dy;
                        d += dz * dz;
                        if (r2lim >= 0.0 && !(d < r2lim)) continue;
                        if (MODE == M_COUNT) {
                            cnt++;
                            continue;
                        }
                        if (cnt == k && !(d < worst || (d == worst && pid < worst_idx))) continue;
                        // sorted insertion, ties -> lower original index first
                        int p = cnt == k ? k - 1 : cnt;
                        while (p > 0) {
                            const double dp = Ld[(p - 1) * PC_WAVE + lane];
                            if (dp < d) break;
                            const int32_t pp = Lp[(p - 1) * PC_WAVE + lane];
                            if (dp == d) {
                                double tx, ty, tz;
                                int32_t tid;
                                RecTraits<REC>::load(rec + pp, tx, ty, tz, tid);
                                if (tid < pid) break;
                            }
                            Ld[p * PC_WAVE + lane] = dp;
                            Lp[p * PC_WAVE + lane] = pp;
                            p--;
                        }
                        Ld[p * PC_WAVE + lane] = d;
                        Lp[p * PC_WAVE + lane] = (int32_t)j;
                        if (cnt < k) cnt++;
                        if (cnt == k) {
                            worst = Ld[(k - 1) * PC_WAVE + lane];
                            double tx, ty, tz;
                            RecTraits<REC>::load(rec + Lp[(k - 1) * PC_WAVE + lane], tx, ty, tz, worst_idx);
                        }
                    }
                }
            }
        }
        bool whole = true;
        double dmin = INFINITY;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            if (lo[a] > 0) {
                whole = false;
                dmin = fmin(dmin, qv[a] - (gp.o[a] + (double)lo[a] * gp.cell));
            }
            if (hi[a] < gp.dim[a] - 1) {
                whole = false;
                dmin = fmin(dmin, (gp.o[a] + (double)(hi[a] + 1) * gp.cell) - qv[a]);
            }
        }
        if (whole) break;
        const double b = dmin - eps;
        if (b > 0) {
            const double b2 = b * b;
            if (r2lim >= 0.0 && b2 >= r2lim) break;
            if (MODE != M_COUNT && cnt == k && worst <= b2) break;
        }
    }

    if (MODE == M_COUNT) {
        A.out_cnt[out_row] = cnt;
    } else if (MODE == M_KNN) {
        for (int s = 0; s < k; s++) {
            int32_t id = -1;
            double d = INFINITY;
            if (s < cnt) {
                double tx, ty, tz;
                RecTraits<REC>::load(rec + Lp[s * PC_WAVE + lane], tx, ty, tz, id);
                d = Ld[s * PC_WAVE + lane];
            }
            if (A.out_idx) A.out_idx[out_row * k + s] = id;
            if (A.out_d2) {
                if (A.d2_f64)
                    reinterpret_cast<double *>(A.out_d2)[out_row * k + s] = d;
                else
                    reinterpret_cast<float *>(A.out_d2)[out_row * k + s] = (float)d;
            }
        }
        if (A.out_cnt) A.out_cnt[out_row] = cnt;
    } else if (MODE == M_SOR) {
        double m = -1.0;
        if (cnt > 0) {
            double s = 0.0;
            for (int j = 0; j < cnt; j++) s += sqrt(Ld[j * PC_WAVE + lane]);
            m = s / (double)cnt;
        }
        A.out_f64[out_row] = m;
    } else if (MODE == M_NNDIST) {
        A.out_f64[out_row] = cnt <= 1 ? 0.0 : sqrt(Ld[1 * PC_WAVE + lane]);
    } else if (MODE == M_NORMALS) {
        double nv[3];
        if (cnt >= 3) {
            double cu[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            for (int j = 0; j < cnt; j++) {
                double px, py, pz;
                int32_t pid;
                RecTraits<REC>::load(rec + Lp[j * PC_WAVE + lane], px, py, pz, pid);
                cu[0] += px;
                cu[1] += py;
                cu[2] += pz;
                cu[3] += px * px;
                cu[4] += px * py;
                cu[5] += px * pz;
                cu[6] += py * py;
                cu[7] += py * pz;
                cu[8] += pz * pz;
            }
#pragma unroll
            for (int i = 0; i < 9; i++) cu[i] /= (double)cnt;
            double cov[6];
            cov[0] = cu[3] - cu[0] * cu[0];
            cov[3] = cu[6] - cu[1] * cu[1];
            cov[5] = cu[8] - cu[2] * cu[2];
            cov[1] = cu[4] - cu[0] * cu[1];
            cov[2] = cu[5] - cu[0] * cu[2];
            cov[4] = cu[7] - cu[1] * cu[2];
            fast_eigen3x3(cov, nv);
        } else {
            nv[0] = 0, nv[1] = 0, nv[2] = 1;  // identity covariance -> smallest-diagonal rule -> +z
        }
        const double nn = sqrt(nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2]);
        if (nn == 0.0) {
            if (A.prior) {
                nv[0] = A.prior[3 * out_row], nv[1] = A.prior[3 * out_row + 1], nv[2] = A.prior[3 * out_row + 2];
            } else {
                nv[0] = 0, nv[1] = 0, nv[2] = 1;
            }
        }
        if (A.prior &&
            nv[0] * A.prior[3 * out_row] + nv[1] * A.prior[3 * out_row + 1] + nv[2] * A.prior[3 * out_row + 2] < 0.0) {
            nv[0] *= -1.0, nv[1] *= -1.0, nv[2] *= -1.0;
        }
        A.out_f64[3 * out_row] = nv[0];
        A.out_f64[3 * out_row + 1] = nv[1];
        A.out_f64[3 * out_row + 2] = nv[2];
    }
}

template <int MODE>
int launch_scan(pc_ctx *ctx, const pc_grid *g, int qdtype, const ScanArgs &A, hipStream_t st) {
    if (A.nq <= 0) return PC_OK;
    const unsigned nb = (unsigned)((A.nq + PC_WAVE - 1) / PC_WAVE);
    const size_t lds = MODE == M_COUNT ? 0 : (sizeof(double) + sizeof(int32_t)) * PC_WAVE * (size_t)A.k;
    const bool ext = A.queries != nullptr;
    if (g->dtype == PC_F32) {
        if (ext && qdtype == PC_F64)
            hipLaunchKernelGGL((k_scan<Rec32, double, MODE>), dim3(nb), dim3(PC_WAVE), lds, st, A);
        else
            hipLaunchKernelGGL((k_scan<Rec32, float, MODE>), dim3(nb), dim3(PC_WAVE), lds, st, A);
    } else {
        if (ext && qdtype == PC_F32)
            hipLaunchKernelGGL((k_scan<Rec64, float, MODE>), dim3(nb), dim3(PC_WAVE), lds, st, A);
        else
            hipLaunchKernelGGL((k_scan<Rec64, double, MODE>), dim3(nb), dim3(PC_WAVE), lds, st, A);
    }
    PC_HIP(hipGetLastError());
    return PC_OK;
}

ScanArgs base_args(const pc_grid *g) {
    ScanArgs A{};
    A.gp = g->gp;
    A.rec = g->rec;
    A.cell_start = g->cell_start;
    A.radius2 = -1.0;
    return A;
}

// ---- small reductions for the statistical filter -------------------------------------------------
// Two-level, fixed-shape tree: the result does not depend on scheduling.
constexpr int RED_BLOCKS = 512;

template <int WHAT>  // 0: sum of avg>0 and count of avg>=0 ; 1: sum over avg>0 of (avg-mean)^2
__global__ __launch_bounds__(256) void k_sor_reduce(const double *__restrict__ avg, int64_t n, double mean,
                                                     double *__restrict__ part) {
    double s = 0.0, c = 0.0;
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t b = (int64_t)blockIdx.x * per, e = b + per < n ? b + per : n;
    for (int64_t i = b + threadIdx.x; i < e; i += 256) {
        const double v = avg[i];
        if (WHAT == 0) {
            if (v > 0) s += v;
            if (v >= 0) c += 1.0;
        } else {
            if (v > 0) s += (v - mean) * (v - mean);
        }
    }
    __shared__ double red[2][256];
    red[0][threadIdx.x] = s;
    red[1][threadIdx.x] = c;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            red[0][threadIdx.x] += red[0][threadIdx.x + o];
            red[1][threadIdx.x] += red[1][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = red[0][0];
        part[2 * blockIdx.x + 1] = red[1][0];
    }
}

__global__ void k_sor_reduce_final(const double *__restrict__ part, int nb, double *__restrict__ out) {
    if (threadIdx.x != 0) return;
    double s = 0.0, c = 0.0;
    for (int b = 0; b < nb; b++) {
        s += part[2 * b];
        c += part[2 * b + 1];
    }
    out[0] = s;
    out[1] = c;
}

__global__ __launch_bounds__(256) void k_sor_flags(const double *__restrict__ avg, int64_t n, double thr,
                                                    uint8_t *__restrict__ flags) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const double v = avg[i];
        flags[i] = (v > 0 && v < thr) ? 1 : 0;
    }
}

int sor_reduce(pc_ctx *ctx, int what, const double *avg, int64_t n, double mean, double out[2], hipStream_t st) {
    out[0] = out[1] = 0.0;
    if (n <= 0) return PC_OK;
    Scratch part(ctx, sizeof(double) * 2 * (RED_BLOCKS + 1));
    if (!part.ok()) return pc_fail(ctx, PC_NOMEM, "reduction scratch");
    double *p = part.as<double>();
    if (what == 0)
        hipLaunchKernelGGL(k_sor_reduce<0>, dim3(RED_BLOCKS), dim3(256), 0, st, avg, n, mean, p);
    else
        hipLaunchKernelGGL(k_sor_reduce<1>, dim3(RED_BLOCKS), dim3(256), 0, st, avg, n, mean, p);
    hipLaunchKernelGGL(k_sor_reduce_final, dim3(1), dim3(64), 0, st, p, RED_BLOCKS, p + 2 * RED_BLOCKS);
    PC_HIP(hipGetLastError());
    PC_HIP(hipMemcpyAsync(out, p + 2 * RED_BLOCKS, 2 * sizeof(double), hipMemcpyDeviceToHost, st));
    PC_HIP(hipStreamSynchronize(st));
    return PC_OK;
}

bool range_ok(const pc_grid *g, int64_t b, int64_t e) { return b >= 0 && e >= b && e <= g->n; }

}  // namespace

extern "C" {

int pc_knn(pc_ctx *ctx, const pc_grid *grid, const void *queries_dev, int qdtype, int64_t nq, int k, double radius,
           int32_t *out_idx_dev, void *out_d2_dev, int d2_dtype, int32_t *out_cnt_dev, void *stream) {
    if (!ctx || !grid || k < 1 || k > PC_MAX_K || nq < 0) return pc_fail(ctx, PC_BAD_ARG, "pc_knn");
    if (!queries_dev) return pc_knn_range(ctx, grid, 0, grid->n, k, radius, out_idx_dev, out_d2_dev, d2_dtype,
                                          out_cnt_dev, stream);
    PC_HIP(hipSetDevice(ctx->device));
    ScanArgs A = base_args(grid);
    A.queries = queries_dev;
    A.s_begin = 0;
    A.nq = nq;
    A.k = k;
    A.radius2 = radius >= 0 ? radius * radius : -1.0;
    A.out_idx = out_idx_dev;
    A.out_d2 = out_d2_dev;
    A.d2_f64 = d2_dtype == PC_F64;
    A.out_cnt = out_cnt_dev;
    return launch_scan<M_KNN>(ctx, grid, qdtype, A, (hipStream_t)stream);
}

int pc_knn_range(pc_ctx *ctx, const pc_grid *grid, int64_t s_begin, int64_t s_end, int k, double radius,
                 int32_t *out_idx_dev, void *out_d2_dev, int d2_dtype, int32_t *out_cnt_dev, void *stream) {
    if (!ctx || !grid || k < 1 || k > PC_MAX_K || !range_ok(grid, s_begin, s_end))
        return pc_fail(ctx, PC_BAD_ARG, "pc_knn_range");
    PC_HIP(hipSetDevice(ctx->device));
    ScanArgs A = base_args(grid);
    A.s_begin = s_begin;
    A.nq = s_end - s_begin;
    A.k = k;
    A.radius2 = radius >= 0 ? radius * radius : -1.0;
    A.out_idx = out_idx_dev;
    A.out_d2 = out_d2_dev;
    A.d2_f64 = d2_dtype == PC_F64;
    A.out_cnt = out_cnt_dev;
    return launch_scan<M_KNN>(ctx, grid, grid->dtype, A, (hipStream_t)stream);
}

int pc_sor_avg(pc_ctx *ctx, const pc_grid *grid, int64_t s_begin, int64_t s_end, int k, double *out_avg_dev,
               void *stream) {
    if (!ctx || !grid || k < 1 || k > PC_MAX_K || !range_ok(grid, s_begin, s_end) || (s_end > s_begin && !out_avg_dev))
        return pc_fail(ctx, PC_BAD_ARG, "pc_sor_avg");
    PC_HIP(hipSetDevice(ctx->device));
    ScanArgs A = base_args(grid);
    A.s_begin = s_begin;
    A.nq = s_end - s_begin;
    A.k = k;
    A.out_f64 = out_avg_dev;
    return launch_scan<M_SOR>(ctx, grid, grid->dtype, A, (hipStream_t)stream);
}

int pc_sor_sum(pc_ctx *ctx, const double *avg_dev, int64_t n, double *sum_pos, int64_t *n_valid, void *stream) {
    if (!ctx || n < 0 || (n > 0 && !avg_dev) || !sum_pos || !n_valid) return pc_fail(ctx, PC_BAD_ARG, "pc_sor_sum");
    PC_HIP(hipSetDevice(ctx->device));
    double o[2];
    PC_TRY(sor_reduce(ctx, 0, avg_dev, n, 0.0, o, (hipStream_t)stream));
    *sum_pos = o[0];
    *n_valid = (int64_t)o[1];
    return PC_OK;
}

int pc_sor_sqsum(pc_ctx *ctx, const double *avg_dev, int64_t n, double mean, double *sq, void *stream) {
    if (!ctx || n < 0 || (n > 0 && !avg_dev) || !sq) return pc_fail(ctx, PC_BAD_ARG, "pc_sor_sqsum");
    PC_HIP(hipSetDevice(ctx->device));
    double o[2];
    PC_TRY(sor_reduce(ctx, 1, avg_dev, n, mean, o, (hipStream_t)stream));
    *sq = o[0];
    return PC_OK;
}

int pc_sor_select(pc_ctx *ctx, const double *avg_dev, int64_t n, double threshold, uint8_t *out_mask_dev,
                  int32_t *out_kept_dev, int64_t *n_kept, void *stream) {
    if (!ctx || n < 0 || (n > 0 && !avg_dev) || !n_kept) return pc_fail(ctx, PC_BAD_ARG, "pc_sor_select");
    PC_HIP(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    *n_kept = 0;
    if (n == 0) return PC_OK;
    Scratch tmp(ctx, (size_t)n);
    if (!tmp.ok()) return pc_fail(ctx, PC_NOMEM, "mask scratch");
    uint8_t *flags = out_mask_dev ? out_mask_dev : tmp.as<uint8_t>();
    hipLaunchKernelGGL(k_sor_flags, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, avg_dev, n, threshold, flags);
    PC_HIP(hipGetLastError());
    return pc_compact_flags(ctx, flags, n, out_kept_dev, n_kept, st);
}

int pc_normals(pc_ctx *ctx, const pc_grid *grid, int64_t s_begin, int64_t s_end, int k, double radius,
               const double *prior_dev, double *out_normals_dev, void *stream) {
    if (!ctx || !grid || k < 1 || k > PC_MAX_K || !range_ok(grid, s_begin, s_end) ||
        (s_end > s_begin && !out_normals_dev))
        return pc_fail(ctx, PC_BAD_ARG, "pc_normals");
    PC_HIP(hipSetDevice(ctx->device));
    ScanArgs A = base_args(grid);
    A.s_begin = s_begin;
    A.nq = s_end - s_begin;
    A.k = k;
    A.radius2 = radius > 0 ? radius * radius : -1.0;
    A.out_f64 = out_normals_dev;
    A.prior = prior_dev;
    return launch_scan<M_NORMALS>(ctx, grid, grid->dtype, A, (hipStream_t)stream);
}

int pc_radius_count(pc_ctx *ctx, const pc_grid *grid, const void *queries_dev, int qdtype, int64_t nq,
                    int64_t s_begin, int64_t s_end, double radius, int32_t *out_cnt_dev, void *stream) {
    if (!ctx || !grid || !(radius >= 0) || !out_cnt_dev) return pc_fail(ctx, PC_BAD_ARG, "pc_radius_count");
    PC_HIP(hipSetDevice(ctx->device));
    ScanArgs A = base_args(grid);
    A.k = 1;
    A.radius2 = radius * radius;
    A.out_cnt = out_cnt_dev;
    if (queries_dev) {
        if (nq < 0) return pc_fail(ctx, PC_BAD_ARG, "pc_radius_count");
        A.queries = queries_dev;
        A.nq = nq;
        return launch_scan<M_COUNT>(ctx, grid, qdtype, A, (hipStream_t)stream);
    }
    if (!range_ok(grid, s_begin, s_end)) return pc_fail(ctx, PC_BAD_ARG, "pc_radius_count");
    A.s_begin = s_begin;
    A.nq = s_end - s_begin;
    return launch_scan<M_COUNT>(ctx, grid, grid->dtype, A, (hipStream_t)stream);
}

int pc_nn_distance(pc_ctx *ctx, const pc_grid *grid, int64_t s_begin, int64_t s_end, double *out_dev, void *stream) {
    if (!ctx || !grid || !range_ok(grid, s_begin, s_end) || (s_end > s_begin && !out_dev))
        return pc_fail(ctx, PC_BAD_ARG, "pc_nn_distance");
    PC_HIP(hipSetDevice(ctx->device));
    ScanArgs A = base_args(grid);
    A.s_begin = s_begin;
    A.nq = s_end - s_begin;
    A.k = 2;
    A.out_f64 = out_dev;
    return launch_scan<M_NNDIST>(ctx, grid, grid->dtype, A, (hipStream_t)stream);
}

}  // extern "C"
