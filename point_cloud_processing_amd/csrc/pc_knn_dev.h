// Device-side helpers shared by the neighbour-search kernels: record loads, the closed-form
// symmetric 3x3 eigen-solver and the per-query consumers (mean distance, normal, ...).
#pragma once
#include <math.h>

#include "pc_common.h"

// first sorted position whose cell key is >= key (key may be ncell: one past the last cell)
__device__ inline uint32_t cell_begin(const CellIndex &ci, int64_t key) {
    if (ci.table) return ci.table[key];
    uint32_t lo = 0, hi = ci.n;
    if (ci.row_start) {
        const uint32_t row = (uint32_t)(key / ci.nx);
        lo = ci.row_start[row];
        hi = ci.row_start[row + 1];
    }
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if ((int64_t)ci.keys[mid] < key)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

// cell_begin of two cells of one (y, z) row at once (key0 <= key1; key1 may be the first cell of the next
// row).  The sparse form is searched eight ways at a time, both searches in step: a look-up is a chain of
// dependent loads from a large array, each a likely TLB miss for a wave on its own, so the chain is kept
// short -- seven keys per search and round trip narrow a row of 4096 records in four trips, not twelve.
__device__ inline void cell_range(const CellIndex &ci, int64_t key0, int64_t key1, uint32_t &b, uint32_t &e) {
    if (ci.table) {
        b = ci.table[key0];
        e = ci.table[key1];
        return;
    }
    uint32_t lo[2] = {0, 0}, hi[2] = {ci.n, ci.n};
    if (ci.row_start) {
        const uint32_t row = (uint32_t)(key0 / ci.nx);
        lo[0] = lo[1] = ci.row_start[row];
        hi[0] = hi[1] = ci.row_start[row + 1];
    }
    const int64_t key[2] = {key0, key1};
    const uint32_t last = ci.n - 1u;  // (the loop runs only where ci.n > 0)
    while (lo[0] < hi[0] || lo[1] < hi[1]) {
        uint32_t kv[2][7], step[2];
        // all fourteen loads first, none of them under a branch (addresses past a range are clamped and
        // their values not used): they are in flight together and waited for once
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const uint32_t len = hi[q] - lo[q];
            // len <= 7: the entries themselves; else seven pivots an eighth of the range apart
            step[q] = len <= 7u ? 1u : len >> 3;
#pragma unroll
            for (int i = 0; i < 7; i++) {
                const uint32_t at = lo[q] + (len <= 7u ? (uint32_t)i : (uint32_t)(i + 1) * step[q] - 1u);
                kv[q][i] = ci.keys[min(at, last)];
            }
        }
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const uint32_t len = hi[q] - lo[q];
            int c = 0;  // loaded entries below the key: a prefix, the keys being sorted
#pragma unroll
            for (int i = 0; i < 7; i++) c += ((int64_t)kv[q][i] < key[q] && (len > 7u || (uint32_t)i < len)) ? 1 : 0;
            // pivot i sits at lo + (i + 1) * step - 1; the answer lies behind the last pivot below the key
            // and not behind the first one at or above it
            const uint32_t base = lo[q];
            const uint32_t hi_few = base + (uint32_t)c, lo_few = hi_few;
            const uint32_t hi_many = c < 7 ? base + (uint32_t)(c + 1) * step[q] - 1u : hi[q];
            const uint32_t lo_many = c > 0 ? base + (uint32_t)c * step[q] : base;
            hi[q] = len == 0 ? hi[q] : (len <= 7u ? hi_few : hi_many);
            lo[q] = len == 0 ? lo[q] : (len <= 7u ? lo_few : lo_many);
        }
    }
    b = lo[0];
    e = lo[1];
}

enum Mode { M_KNN = 0, M_SOR = 1, M_NORMALS = 2, M_NNDIST = 3, M_COUNT = 4, M_ALL = 5 };

// one resolution level of the spatial index
struct Level {
    GridParams gp;
    const void *rec;
    CellIndex ci;
};

struct ScanArgs {
    GridParams gp;
    const void *rec;
    CellIndex ci;
    const void *queries;  // NULL: self-query over sorted positions
    // External queries binned into the grid: records {x, y, z, query row} sorted by the grid's cell
    // key, and those sorted keys.  NULL: the queries are the grid's own points.  Where set, "sorted
    // position" of a query means its position in qrec, and per-query arrays are indexed by query row.
    const void *qrec;
    CellIndex qci;
    const void *qraw;     // the external queries as the caller passed them (nq x 3 of the grid's dtype)
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
    double *out_nrm;        // normals of the fused mode (M_ALL), where out_f64 is the SOR average
    const double *prior;    // normals only
    // compact outputs (query shards): row of the outputs that original index i is written to, or NULL: row i.
    // (prior normals are not supported together with it: they are indexed by the original index.)
    const int32_t *out_slot;
    // ... or, where the kernel knows the query's sorted position at level 0 (single-level indexes): row =
    // position - compact_base; < 0: off
    int64_t compact_base;
    // optional explicit list of sorted positions to answer (the exact re-run of queries that
    // the tile kernel could not certify); *qcount entries of qlist
    const int32_t *qlist;
    const float *qub;       // per listed query: an upper bound of its k-th squared distance (or +inf)
    const uint32_t *qcount;
    // tile kernel only
    int tile[3];            // tile edge in cells
    int ntile[3];           // tiles per axis
    int64_t s_end;          // self-query range end (sorted position)
    int cand_cap;           // candidates the LDS block holds
    const uint32_t *tile_list;  // occupied tiles (sparse index), or NULL: blockIdx + tile_base is the tile id
    int64_t tile_base;          // first tile of the launch (query shards launch the tiles of their own z layers only)
    // multi-level passes: per-point flags and bounds, indexed by ORIGINAL point index
    const uint8_t *qflag_in;    // NULL: every point of the range is a query; else only flagged points
    uint8_t *qflag_out;         // where uncertified queries are flagged for the next (coarser) pass;
                                // NULL on the last pass, which appends to flag_list instead
    float *qub_arr;             // upper bound of the k-th squared distance found so far
    // A flag is 0 (answered) or 1 + the first level that is to try the query again.  A pass answers the
    // flagged queries whose flag is at most flag_max.
    int flag_max;
    int flag_top;               // the flag of the coarsest level, where passes may postpone a query by its flag; 0: they may not
    int32_t *flag_list;     // queries handed to the exact re-run (one wave per query)
    float *flag_ub;
    uint32_t *flag_count;
    int32_t *flag2_list;    // queries the wave-per-query re-run hands on to the per-lane scan
    uint32_t *flag2_count;
    uint32_t flag_cap;
    int halves;             // the records of a cell are ordered low x-half first (fp32 grids of evenly filled clouds)
    int count_crowded;      // this pass counts the queries amid more candidates than a tile stages (flag_count[26])
    // wave-per-query re-run: the index levels, finest first
    Level lv[MAX_LEVELS + MAX_FINER];  // (the finer grids made for crowded queries come first)
    int nlv;
    // ring of the first block of a listed query that comes without a bound: 2 where a tile pass of the same
    // (only) level has looked at the query's 3 x 3 x 3 cells already and found them wanting, else 1
    int first_ring;
    // tile kernel: the threshold of the filter starts at this multiple of the certification bound (squared distances)
    double seed_scale;
};

// The flag of an open query amid more candidates than any level's pass stages (only where passes may postpone
// queries, ScanArgs::flag_top > 0): coarser cells hold more candidates still, so no later level lists it; it waits
// for the passes over ALL open queries (finer grids, rings of the finest level, one wave per query).
constexpr uint8_t FLAG_CROWDED = 200;

// an open query (flag set) that the pass of flag_max answers
__host__ __device__ inline bool flag_live(uint8_t flag, int flag_max) { return flag != 0 && (int)flag <= flag_max; }

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


// sums of x, y, z, xx, xy, xz, yy, yz, zz over the neighbours, nearest first (Open3D's ComputeCovariance)
__device__ inline void cumulant_add(double (&cu)[9], double px, double py, double pz) {
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

// the normal of one query from the cumulants of its cnt neighbours (cu is used up)
template <int MODE>
__device__ inline void emit_normal(const ScanArgs &A, int64_t out_row, int cnt, double (&cu)[9]) {
    {
        double *out_n = MODE == M_ALL ? A.out_nrm : A.out_f64;
        double nv[3];
        if (cnt >= 3) {
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
        out_n[3 * out_row] = nv[0];
        out_n[3 * out_row + 1] = nv[1];
        out_n[3 * out_row + 2] = nv[2];
    }
}


// Row of the outputs for the query with original index (or query row) `id` at sorted position `spos`.
__device__ inline int64_t out_row_of(const ScanArgs &A, int64_t id, int64_t spos) {
    if (A.out_slot) return A.out_slot[id];
    return A.compact_base >= 0 ? spos - A.compact_base : id;
}

// ---- per-query consumers -------------------------------------------------------------------------
// NB(s, d2, x, y, z, idx) yields the s-th nearest neighbour (ascending (d2, idx)) of this query.
// ONLY_NORMALS: the caller has written the rows / mean distance itself.
template <int MODE, bool ONLY_NORMALS = false, typename NB>
__device__ inline void emit_result(const ScanArgs &A, int64_t out_row, int cnt, NB nb) {
    const int k = A.k;
    const bool rows = !ONLY_NORMALS && (MODE == M_KNN || MODE == M_ALL) && (A.out_idx || A.out_d2 || A.out_cnt);
    const bool mean = !ONLY_NORMALS && (MODE == M_SOR || (MODE == M_ALL && A.out_f64));
    const bool normal = MODE == M_NORMALS || (MODE == M_ALL && A.out_nrm);
    // One walk over the neighbours feeds every consumer, and four neighbours are fetched before any of them is used:
    // a neighbour may be a record in global memory (fp64 grids), and a walk per consumer with one record in
    // flight waited for 3 k round trips per query (66 M voxels, k = 32, rows + mean distance + normals: 20 of
    // the first tile pass's 58 ms).  Every sum still takes its terms nearest first, one by one.
    constexpr int AT_ONCE = 4;
    double sum = 0.0;
    double cu[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int upto = rows ? k : ((mean || (normal && cnt >= 3)) ? cnt : 0);
    for (int s0 = 0; s0 < upto; s0 += AT_ONCE) {
        double d[AT_ONCE], x[AT_ONCE], y[AT_ONCE], z[AT_ONCE];
        int32_t id[AT_ONCE];
#pragma unroll
        for (int v = 0; v < AT_ONCE; v++) {
            id[v] = -1;
            d[v] = INFINITY;
            x[v] = y[v] = z[v] = 0.0;
            if (s0 + v < cnt) nb(s0 + v, d[v], x[v], y[v], z[v], id[v]);
        }
        // rows of whole 16-byte groups leave as vector stores (a lane's row is its own: word by word every store
        // instruction touches 64 cache lines for 4 bytes each)
        static_assert(AT_ONCE == 4, "the vector stores below write four neighbours");
        const bool vec = rows && (k & 3) == 0 && s0 + AT_ONCE <= k &&
                         ((reinterpret_cast<uintptr_t>(A.out_idx) | reinterpret_cast<uintptr_t>(A.out_d2)) & 15) == 0;
        if (vec) {
            if (A.out_idx) *reinterpret_cast<int4 *>(A.out_idx + out_row * k + s0) = make_int4(id[0], id[1], id[2], id[3]);
            if (A.out_d2) {
                if (A.d2_f64) {
                    double *o = reinterpret_cast<double *>(A.out_d2) + out_row * k + s0;
                    *reinterpret_cast<double2 *>(o) = make_double2(d[0], d[1]);
                    *reinterpret_cast<double2 *>(o + 2) = make_double2(d[2], d[3]);
                } else {
                    *reinterpret_cast<float4 *>(reinterpret_cast<float *>(A.out_d2) + out_row * k + s0) =
                        make_float4((float)d[0], (float)d[1], (float)d[2], (float)d[3]);
                }
            }
        }
#pragma unroll
        for (int v = 0; v < AT_ONCE; v++) {
            const int s = s0 + v;
            if (rows && !vec && s < k) {
                if (A.out_idx) A.out_idx[out_row * k + s] = id[v];
                if (A.out_d2) {
                    if (A.d2_f64)
                        reinterpret_cast<double *>(A.out_d2)[out_row * k + s] = d[v];
                    else
                        reinterpret_cast<float *>(A.out_d2)[out_row * k + s] = (float)d[v];
                }
            }
            if (s < cnt) {
                if (mean) sum += sqrt(d[v]);
                if (normal && cnt >= 3) cumulant_add(cu, x[v], y[v], z[v]);
            }
        }
    }
    if (rows && A.out_cnt) A.out_cnt[out_row] = cnt;
    if (mean) A.out_f64[out_row] = cnt > 0 ? sum / (double)cnt : -1.0;
    if (!ONLY_NORMALS && MODE == M_NNDIST) {
        double d = 0.0, x, y, z;
        int32_t id;
        if (cnt > 1) {
            nb(1, d, x, y, z, id);
            d = sqrt(d);
        }
        A.out_f64[out_row] = d;
    }
    if (normal) emit_normal<MODE>(A, out_row, cnt, cu);
}

// pc_knn_tile.hip: the tile-cooperative fast path (self-queries, k <= 50).
// Returns PC_OK after enqueueing; *handled = false when the configuration is not supported.
int pc_launch_tile_scan(pc_ctx *ctx, const pc_grid *g, int mode, ScanArgs &A, hipStream_t st, bool *handled);
// pc_knn_tile_*.hip: the tile kernel for one record type, with / without sub-tile splitting
void pc_tile_launch_f32(int ks, int mode, unsigned nblocks, int nw, size_t lds, hipStream_t st, const ScanArgs &A);
void pc_tile_launch_f32_split(int ks, int mode, unsigned nblocks, int nw, size_t lds, hipStream_t st, const ScanArgs &A);
void pc_tile_launch_f64(int ks, int mode, unsigned nblocks, int nw, size_t lds, hipStream_t st, const ScanArgs &A);
void pc_tile_launch_f64_split(int ks, int mode, unsigned nblocks, int nw, size_t lds, hipStream_t st, const ScanArgs &A);

// pc_knn_batch.hip: a level pass over the list of still-open queries, 64 to a wave
int pc_batch_open_list(pc_ctx *ctx, int dtype, const void *qsrc, int64_t n, const uint8_t *qflag, int flag_max, int32_t *list_out,
                       int64_t *count, int64_t *open_all, uint32_t *counter, hipStream_t st);
int pc_launch_batch_scan(pc_ctx *ctx, int dtype, int ks, int mode, const ScanArgs &A, const int32_t *list, int64_t count,
                         hipStream_t st);

// pc_knn_ring.hip: the open queries a ladder leaves, lane per query on the finest level with rings of several cells
int pc_launch_ring_scan(pc_ctx *ctx, int dtype, int ks, int mode, const ScanArgs &A, const int32_t *list, int64_t count,
                        int64_t stride, uint32_t *answered, hipStream_t st);

int pc_ring_eligible(pc_ctx *ctx, int dtype, const ScanArgs &A, const int32_t *list, int64_t count, uint32_t *out_dev, hipStream_t st);

// pc_knn_wave.hip: one wave per listed query (A.qlist / A.qub / A.qcount).
int pc_launch_wave_query(pc_ctx *ctx, const pc_grid *g, int mode, const ScanArgs &A, hipStream_t st);
