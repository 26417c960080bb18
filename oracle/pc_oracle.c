/*
 * CPU oracle for the neighbourhood stage (TEST INFRASTRUCTURE, not product code).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library.  The product path (point_cloud_processing_amd) never links, imports or calls it.
 *
 * PARITY UNPINNED: the arithmetic of this path lives in the third-party wheel
 * open3d==0.19.0 (/root/reference/requirements.txt:39), which is not vendored in the
 * reference tree and is not installable here; the reference holds no tests, fixtures or
 * golden vectors.  Every routine below is a restatement, from the published Open3D 0.19.0
 * sources, of the library routine that a reference call site invokes; each one names the
 * call site (reference file:line) and the upstream routine it restates.  The restatement is
 * cross-checked in tests/ against scipy.spatial.cKDTree, numpy.linalg.eigh and closed-form
 * known answers.
 *
 * All arithmetic is IEEE fp64 in the operation order of the upstream routine; the file is
 * compiled with -ffp-contract=off so no multiply-add is fused.
 */
#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

enum { ORC_OK = 0, ORC_BAD_ARG = -1, ORC_NOMEM = -2, ORC_OVERFLOW = -4 };

/* ------------------------------------------------------------------------------------------
 * Bounding box, translate, scale.
 * Call sites: app/Core/pointcloud_preprocess.py:45-53, :104-105.
 * Upstream: PointCloud::GetMinBound/GetMaxBound (component-wise min/max),
 * AxisAlignedBoundingBox::GetCenter = (min+max)*0.5, GetMaxExtent = (max-min).maxCoeff(),
 * TranslatePoints (p += t), ScalePoints (p = (p - c) * s + c).
 * ---------------------------------------------------------------------------------------- */
ORC_API int orc_bbox(const double *p, int64_t n, double mn[3], double mx[3]) {
    if (n <= 0) {
        for (int a = 0; a < 3; a++) mn[a] = mx[a] = 0.0;
        return ORC_OK;
    }
    for (int a = 0; a < 3; a++) mn[a] = mx[a] = p[a];
    for (int64_t i = 1; i < n; i++)
        for (int a = 0; a < 3; a++) {
            double v = p[3 * i + a];
            if (v < mn[a]) mn[a] = v;
            if (v > mx[a]) mx[a] = v;
        }
    return ORC_OK;
}

/* normalize_point_cloud, app/Core/pointcloud_preprocess.py:42-55:
 * center=(min+max)*0.5, scale=max extent, p += (-center), then (scale>0) p = (p-0)*(1/scale)+0. */
ORC_API int orc_normalize(double *p, int64_t n, double center[3], double *scale) {
    double mn[3], mx[3];
    orc_bbox(p, n, mn, mx);
    double ext = -DBL_MAX;
    for (int a = 0; a < 3; a++) {
        center[a] = (mn[a] + mx[a]) * 0.5;
        double e = mx[a] - mn[a];
        if (e > ext) ext = e;
    }
    if (n <= 0) ext = 0.0;
    *scale = ext;
    for (int64_t i = 0; i < n; i++)
        for (int a = 0; a < 3; a++) p[3 * i + a] += -center[a];
    if (ext > 0) {
        double s = 1.0 / ext;
        for (int64_t i = 0; i < n; i++)
            for (int a = 0; a < 3; a++) p[3 * i + a] = (p[3 * i + a] - 0.0) * s + 0.0;
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------
 * Voxel down-sampling.
 * Call sites: app/Core/pointcloud_preprocess.py:96, app/Infrastructure/filtering_adapter.py:88.
 * Upstream: PointCloud::VoxelDownSample.  vmin = min_bound - 0.5*vs; per point
 * ref = (p - vmin) / vs (true divide), index = int(floor(ref)); per-voxel sums are taken in
 * point-index order; outputs are sum / count (normals are NOT re-normalised).
 * Upstream emits voxels in std::unordered_map iteration order (unspecified); this oracle and
 * the device path both emit ascending linear key (iz*ny + iy)*nx + ix.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    uint64_t key;
    int64_t idx;
} vox_item;

static int vox_cmp(const void *a, const void *b) {
    const vox_item *x = (const vox_item *)a, *y = (const vox_item *)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);
}

/* Voxel index triple of every point (out_ijk: n x 3 int32) plus grid dims. */
ORC_API int orc_voxel_index(const double *p, int64_t n, double vs, int32_t *out_ijk, int32_t dims[3],
                            double vmin_out[3]) {
    if (!(vs > 0.0)) return ORC_BAD_ARG;
    double mn[3], mx[3];
    orc_bbox(p, n, mn, mx);
    double vmin[3], vmax[3], ext = -DBL_MAX;
    for (int a = 0; a < 3; a++) {
        vmin[a] = mn[a] - vs * 0.5;
        vmax[a] = mx[a] + vs * 0.5;
        if (vmax[a] - vmin[a] > ext) ext = vmax[a] - vmin[a];
        if (vmin_out) vmin_out[a] = vmin[a];
    }
    if (vs * (double)INT_MAX < ext) return ORC_OVERFLOW; /* "voxel_size is too small" */
    for (int a = 0; a < 3; a++) dims[a] = n > 0 ? (int32_t)floor((mx[a] - vmin[a]) / vs) + 1 : 0;
    for (int64_t i = 0; i < n; i++)
        for (int a = 0; a < 3; a++) out_ijk[3 * i + a] = (int32_t)floor((p[3 * i + a] - vmin[a]) / vs);
    return ORC_OK;
}

/* Returns the number of voxels M (>=0) or a negative error.  out_* must hold n rows.
 * nrm / col / out_n / out_c / out_cnt / out_ijk may be NULL. */
ORC_API int64_t orc_voxel_downsample(const double *p, const double *nrm, const double *col, int64_t n, double vs,
                                     double *out_p, double *out_n, double *out_c, int32_t *out_cnt,
                                     int32_t *out_ijk) {
    if (!(vs > 0.0)) return ORC_BAD_ARG;
    if (n == 0) return 0;
    int32_t *ijk = (int32_t *)malloc(sizeof(int32_t) * 3 * (size_t)n);
    vox_item *it = (vox_item *)malloc(sizeof(vox_item) * (size_t)n);
    if (!ijk || !it) {
        free(ijk);
        free(it);
        return ORC_NOMEM;
    }
    int32_t dims[3];
    int rc = orc_voxel_index(p, n, vs, ijk, dims, NULL);
    if (rc) {
        free(ijk);
        free(it);
        return rc;
    }
    for (int64_t i = 0; i < n; i++) {
        it[i].key = ((uint64_t)ijk[3 * i + 2] * (uint64_t)dims[1] + (uint64_t)ijk[3 * i + 1]) * (uint64_t)dims[0] +
                    (uint64_t)ijk[3 * i + 0];
        it[i].idx = i;
    }
    qsort(it, (size_t)n, sizeof(vox_item), vox_cmp);
    int64_t m = 0;
    for (int64_t s = 0; s < n;) {
        int64_t e = s;
        double sp[3] = {0, 0, 0}, sn[3] = {0, 0, 0}, sc[3] = {0, 0, 0};
        while (e < n && it[e].key == it[s].key) {
            int64_t i = it[e].idx;
            for (int a = 0; a < 3; a++) sp[a] += p[3 * i + a];
            if (nrm && !isnan(nrm[3 * i]) && !isnan(nrm[3 * i + 1]) && !isnan(nrm[3 * i + 2]))
                for (int a = 0; a < 3; a++) sn[a] += nrm[3 * i + a];
            if (col)
                for (int a = 0; a < 3; a++) sc[a] += col[3 * i + a];
            e++;
        }
        double cnt = (double)(e - s);
        for (int a = 0; a < 3; a++) {
            out_p[3 * m + a] = sp[a] / cnt;
            if (nrm && out_n) out_n[3 * m + a] = sn[a] / cnt;
            if (col && out_c) out_c[3 * m + a] = sc[a] / cnt;
        }
        if (out_cnt) out_cnt[m] = (int32_t)(e - s);
        if (out_ijk)
            for (int a = 0; a < 3; a++) out_ijk[3 * m + a] = ijk[3 * it[s].idx + a];
        m++;
        s = e;
    }
    free(ijk);
    free(it);
    return m;
}

/* ------------------------------------------------------------------------------------------
 * Exact k-nearest neighbours.
 * Call sites reach it through estimate_normals (pointcloud_preprocess.py:109-114),
 * remove_statistical_outlier (filtering_adapter.py:91), remove_radius_outlier (:94) and
 * compute_nearest_neighbor_distance (filtering_adapter.py:49, mesh_reconstruction.py:112).
 * Upstream: KDTreeFlann::SearchKNN / SearchHybrid / SearchRadius over nanoflann with
 * L2_Adaptor<double>: d2 = ((0 + dx*dx) + dy*dy) + dz*dz, dx = q - p, ascending d2,
 * min(k, N) results, the query point itself included when it belongs to the cloud.
 * Exact ties are unspecified upstream (tree visiting order); the canonical rule here and on
 * the device is: lower point index first.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    double d;
    int32_t i;
} nb_t;

static inline int nb_less(double d, int32_t i, double d2, int32_t i2) { return d < d2 || (d == d2 && i < i2); }

static inline void nb_insert(nb_t *L, int *cnt, int k, double d, int32_t i) {
    int n = *cnt;
    if (n == k) {
        if (!nb_less(d, i, L[k - 1].d, L[k - 1].i)) return;
        n = k - 1;
    }
    int p = n;
    while (p > 0 && nb_less(d, i, L[p - 1].d, L[p - 1].i)) {
        L[p] = L[p - 1];
        p--;
    }
    L[p].d = d;
    L[p].i = i;
    *cnt = n + 1;
}

static inline double dist2(const double *q, const double *p) {
    double dx = q[0] - p[0], dy = q[1] - p[1], dz = q[2] - p[2];
    double r = dx * dx;
    r += dy * dy;
    r += dz * dz;
    return r;
}

/* radius2 < 0: plain kNN.  radius2 >= 0: hybrid search (kNN, then keep the prefix with
 * d2 < radius2), which equals "k smallest among the points with d2 < radius2". */
ORC_API int orc_knn_brute(const double *pts, int64_t n, const double *q, int64_t nq, int k, double radius2,
                          int32_t *out_idx, double *out_d2, int32_t *out_cnt) {
    if (k < 1 || n < 0 || nq < 0) return ORC_BAD_ARG;
    int err = 0;
#pragma omp parallel
    {
        nb_t *L = (nb_t *)malloc(sizeof(nb_t) * (size_t)k);
        if (!L) {
#pragma omp atomic write
            err = 1;
        }
#pragma omp for schedule(dynamic, 64)
        for (int64_t qi = 0; qi < nq; qi++) {
            if (!L) continue;
            int cnt = 0;
            for (int64_t j = 0; j < n; j++) {
                double d = dist2(q + 3 * qi, pts + 3 * j);
                if (radius2 >= 0 && !(d < radius2)) continue;
                nb_insert(L, &cnt, k, d, (int32_t)j);
            }
            for (int s = 0; s < k; s++) {
                if (out_idx) out_idx[qi * k + s] = s < cnt ? L[s].i : -1;
                if (out_d2) out_d2[qi * k + s] = s < cnt ? L[s].d : INFINITY;
            }
            if (out_cnt) out_cnt[qi] = cnt;
        }
        free(L);
    }
    return err ? ORC_NOMEM : ORC_OK;
}

/* Uniform grid over the cloud's bounding box: counting sort of the points by cell,
 * x fastest, so one (y,z) row of cells is one contiguous point range. */
typedef struct {
    int64_t n;
    int32_t dim[3];
    double gmin[3];
    double cell, inv;
    int64_t *start; /* ncell + 1 */
    double *sp;     /* sorted points, 3n */
    int32_t *perm;  /* sorted position -> original index */
} orc_grid;

static inline int cell_of(const orc_grid *g, double v, int a) {
    double t = floor((v - g->gmin[a]) * g->inv);
    if (!(t > 0)) return 0;
    if (t >= (double)g->dim[a]) return g->dim[a] - 1;
    return (int)t;
}

ORC_API void orc_grid_destroy(void *h) {
    orc_grid *g = (orc_grid *)h;
    if (!g) return;
    free(g->start);
    free(g->sp);
    free(g->perm);
    free(g);
}

/* cell <= 0: choose the edge so that an average cell holds about `target` points. */
ORC_API void *orc_grid_create(const double *pts, int64_t n, double cell, double target) {
    orc_grid *g = (orc_grid *)calloc(1, sizeof(orc_grid));
    if (!g) return NULL;
    g->n = n;
    double mn[3], mx[3];
    orc_bbox(pts, n, mn, mx);
    double ext[3], emax = 0;
    for (int a = 0; a < 3; a++) {
        ext[a] = mx[a] - mn[a];
        if (ext[a] > emax) emax = ext[a];
        g->gmin[a] = mn[a];
    }
    if (!(cell > 0)) {
        if (!(target > 0)) target = 4.0;
        /* volume of the box, flat axes replaced by a thin slab, per target points */
        double vol = 1.0;
        for (int a = 0; a < 3; a++) vol *= (ext[a] > emax * 1e-3 ? ext[a] : emax * 1e-3);
        cell = n > 0 && vol > 0 ? cbrt(vol * target / (double)n) : 1.0;
        if (!(cell > 0)) cell = 1.0;
    }
    for (;;) { /* cap the table at ~4n + 4096 cells */
        double tot = 1.0;
        for (int a = 0; a < 3; a++) {
            double d = floor(ext[a] / cell) + 1.0;
            if (d > 2.0e9) d = 2.0e9;
            g->dim[a] = (int32_t)d;
            tot *= d;
        }
        if (tot <= 4.0 * (double)n + 4096.0) break;
        cell *= 1.3;
    }
    g->cell = cell;
    g->inv = 1.0 / cell;
    int64_t nc = (int64_t)g->dim[0] * g->dim[1] * g->dim[2];
    g->start = (int64_t *)calloc((size_t)nc + 1, sizeof(int64_t));
    g->sp = (double *)malloc(sizeof(double) * 3 * (size_t)(n > 0 ? n : 1));
    g->perm = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    int64_t *cid = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    if (!g->start || !g->sp || !g->perm || !cid) {
        free(cid);
        orc_grid_destroy(g);
        return NULL;
    }
    for (int64_t i = 0; i < n; i++) {
        int cx = cell_of(g, pts[3 * i], 0), cy = cell_of(g, pts[3 * i + 1], 1), cz = cell_of(g, pts[3 * i + 2], 2);
        cid[i] = ((int64_t)cz * g->dim[1] + cy) * g->dim[0] + cx;
        g->start[cid[i] + 1]++;
    }
    for (int64_t c = 0; c < nc; c++) g->start[c + 1] += g->start[c];
    int64_t *fill = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nc > 0 ? nc : 1));
    if (!fill) {
        free(cid);
        orc_grid_destroy(g);
        return NULL;
    }
    memcpy(fill, g->start, sizeof(int64_t) * (size_t)nc);
    for (int64_t i = 0; i < n; i++) { /* stable: ascending original index inside a cell */
        int64_t pos = fill[cid[i]]++;
        g->sp[3 * pos] = pts[3 * i];
        g->sp[3 * pos + 1] = pts[3 * i + 1];
        g->sp[3 * pos + 2] = pts[3 * i + 2];
        g->perm[pos] = (int32_t)i;
    }
    free(fill);
    free(cid);
    return g;
}

/* One query: ring expansion until the k-th best distance is certified by the distance to
 * the nearest face of the scanned block (or the block covers the grid).  count_only != 0
 * counts the points with d2 < radius2 instead of keeping a list. */
static int grid_query(const orc_grid *g, const double *q, int k, double radius2, nb_t *L, int count_only,
                      int64_t *count_out) {
    int c[3];
    for (int a = 0; a < 3; a++) c[a] = cell_of(g, q[a], a);
    const double eps = 1e-7 * g->cell;
    int cnt = 0;
    int64_t rc = 0;
    for (int R = 0;; R++) {
        int lo[3], hi[3];
        for (int a = 0; a < 3; a++) {
            lo[a] = c[a] - R < 0 ? 0 : c[a] - R;
            hi[a] = c[a] + R >= g->dim[a] ? g->dim[a] - 1 : c[a] + R;
        }
        for (int z = lo[2]; z <= hi[2]; z++)
            for (int y = lo[1]; y <= hi[1]; y++) {
                int64_t row = ((int64_t)z * g->dim[1] + y) * g->dim[0];
                int full = (z - c[2] == R) || (c[2] - z == R) || (y - c[1] == R) || (c[1] - y == R);
                int nseg = full ? 1 : (R == 0 ? 1 : 2);
                for (int s = 0; s < nseg; s++) {
                    int x0, x1;
                    if (full) {
                        x0 = lo[0];
                        x1 = hi[0];
                    } else {
                        x0 = x1 = s == 0 ? c[0] - R : c[0] + R;
                        if (x0 < 0 || x0 >= g->dim[0]) continue;
                    }
                    for (int64_t j = g->start[row + x0]; j < g->start[row + x1 + 1]; j++) {
                        double d = dist2(q, g->sp + 3 * j);
                        if (radius2 >= 0 && !(d < radius2)) continue;
                        if (count_only)
                            rc++;
                        else
                            nb_insert(L, &cnt, k, d, g->perm[j]);
                    }
                }
            }
        int whole = 1;
        double dmin = INFINITY;
        for (int a = 0; a < 3; a++) {
            if (lo[a] > 0) {
                whole = 0;
                double f = q[a] - (g->gmin[a] + (double)lo[a] * g->cell);
                if (f < dmin) dmin = f;
            }
            if (hi[a] < g->dim[a] - 1) {
                whole = 0;
                double f = (g->gmin[a] + (double)(hi[a] + 1) * g->cell) - q[a];
                if (f < dmin) dmin = f;
            }
        }
        if (whole) break;
        double b = dmin - eps;
        if (b > 0) {
            double b2 = b * b;
            if (radius2 >= 0 && b2 >= radius2) break;
            if (!count_only && cnt == k && L[k - 1].d <= b2) break;
        }
    }
    if (count_out) *count_out = rc;
    return cnt;
}

ORC_API int orc_grid_knn(const void *h, const double *q, int64_t nq, int k, double radius2, int32_t *out_idx,
                         double *out_d2, int32_t *out_cnt) {
    const orc_grid *g = (const orc_grid *)h;
    if (!g || k < 1) return ORC_BAD_ARG;
    int err = 0;
#pragma omp parallel
    {
        nb_t *L = (nb_t *)malloc(sizeof(nb_t) * (size_t)k);
        if (!L) {
#pragma omp atomic write
            err = 1;
        }
#pragma omp for schedule(dynamic, 1024)
        for (int64_t qi = 0; qi < nq; qi++) {
            if (!L) continue;
            int cnt = g->n > 0 ? grid_query(g, q + 3 * qi, k, radius2, L, 0, NULL) : 0;
            for (int s = 0; s < k; s++) {
                if (out_idx) out_idx[qi * k + s] = s < cnt ? L[s].i : -1;
                if (out_d2) out_d2[qi * k + s] = s < cnt ? L[s].d : INFINITY;
            }
            if (out_cnt) out_cnt[qi] = cnt;
        }
        free(L);
    }
    return err ? ORC_NOMEM : ORC_OK;
}

/* Self-query of every grid point, visiting the points in cell order (cache-friendly: this is
 * the variant the CPU baseline times); row i of the outputs belongs to original point i. */
ORC_API int orc_grid_knn_self(const void *h, int64_t s_begin, int64_t s_end, int k, double radius2,
                              int32_t *out_idx, double *out_d2, int32_t *out_cnt) {
    const orc_grid *g = (const orc_grid *)h;
    if (!g || k < 1 || s_begin < 0 || s_end > g->n || s_end < s_begin) return ORC_BAD_ARG;
    int err = 0;
#pragma omp parallel
    {
        nb_t *L = (nb_t *)malloc(sizeof(nb_t) * (size_t)k);
        if (!L) {
#pragma omp atomic write
            err = 1;
        }
#pragma omp for schedule(dynamic, 2048)
        for (int64_t s = s_begin; s < s_end; s++) {
            if (!L) continue;
            int cnt = grid_query(g, g->sp + 3 * s, k, radius2, L, 0, NULL);
            int64_t qi = g->perm[s];
            for (int t = 0; t < k; t++) {
                if (out_idx) out_idx[qi * k + t] = t < cnt ? L[t].i : -1;
                if (out_d2) out_d2[qi * k + t] = t < cnt ? L[t].d : INFINITY;
            }
            if (out_cnt) out_cnt[qi] = cnt;
        }
        free(L);
    }
    return err ? ORC_NOMEM : ORC_OK;
}

/* remove_radius_outlier's neighbour count (filtering_adapter.py:94; upstream
 * PointCloud::RemoveRadiusOutliers over nanoflann radiusSearch): points with d2 < r*r,
 * strict, the point itself included. */
ORC_API int orc_grid_radius_count(const void *h, const double *q, int64_t nq, double radius, int32_t *out_cnt) {
    const orc_grid *g = (const orc_grid *)h;
    if (!g || !(radius >= 0)) return ORC_BAD_ARG;
    double r2 = radius * radius;
#pragma omp parallel for schedule(dynamic, 1024)
    for (int64_t qi = 0; qi < nq; qi++) {
        int64_t c = 0;
        if (g->n > 0) grid_query(g, q + 3 * qi, 1, r2, NULL, 1, &c);
        out_cnt[qi] = (int32_t)c;
    }
    return ORC_OK;
}

/* compute_nearest_neighbor_distance (filtering_adapter.py:49, mesh_reconstruction.py:112;
 * upstream PointCloud::ComputeNearestNeighborDistance): kNN(k=2), sqrt(d2[1]), 0 if <2 found. */
ORC_API int orc_nn_distance(const double *pts, int64_t n, double *out) {
    void *g = orc_grid_create(pts, n, 0.0, 2.0);
    if (!g) return ORC_NOMEM;
#pragma omp parallel for schedule(dynamic, 1024)
    for (int64_t i = 0; i < n; i++) {
        nb_t L[2];
        int cnt = grid_query((orc_grid *)g, pts + 3 * i, 2, -1.0, L, 0, NULL);
        out[i] = cnt <= 1 ? 0.0 : sqrt(L[1].d);
    }
    orc_grid_destroy(g);
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------
 * Statistical outlier removal.
 * Call site: app/Infrastructure/filtering_adapter.py:91.
 * Upstream: PointCloud::RemoveStatisticalOutliers.  avg_i = sum_j sqrt(d2_ij) / found (the
 * zero self distance counts in the divisor), -1 if nothing found; mean over avg_i > 0 divided
 * by the number of points WITH results; Bessel std; keep avg_i > 0 && avg_i < mean + r*std.
 * stats = {mean, std, threshold, valid}.
 * ---------------------------------------------------------------------------------------- */
ORC_API int orc_sor_avg(const double *pts, int64_t n, int k, double *avg) {
    if (k < 1) return ORC_BAD_ARG;
    if (n == 0) return ORC_OK;
    void *g = orc_grid_create(pts, n, 0.0, k < 8 ? 2.0 : 4.0);
    if (!g) return ORC_NOMEM;
    int err = 0;
#pragma omp parallel
    {
        nb_t *L = (nb_t *)malloc(sizeof(nb_t) * (size_t)k);
        if (!L) {
#pragma omp atomic write
            err = 1;
        }
#pragma omp for schedule(dynamic, 1024)
        for (int64_t i = 0; i < n; i++) {
            if (!L) continue;
            int cnt = grid_query((orc_grid *)g, pts + 3 * i, k, -1.0, L, 0, NULL);
            double m = -1.0;
            if (cnt > 0) {
                double s = 0.0;
                for (int j = 0; j < cnt; j++) s += sqrt(L[j].d);
                m = s / (double)cnt;
            }
            avg[i] = m;
        }
        free(L);
    }
    orc_grid_destroy(g);
    return err ? ORC_NOMEM : ORC_OK;
}

ORC_API int orc_sor_threshold(const double *avg, int64_t n, double std_ratio, double stats[4], uint8_t *mask) {
    if (!(std_ratio > 0)) return ORC_BAD_ARG;
    int64_t valid = 0;
    for (int64_t i = 0; i < n; i++)
        if (avg[i] >= 0.0) valid++; /* points with results (avg = -1 marks none) */
    stats[3] = (double)valid;
    if (valid == 0) {
        stats[0] = stats[1] = stats[2] = 0.0;
        if (mask) memset(mask, 0, (size_t)n);
        return ORC_OK;
    }
    double mean = 0.0;
    for (int64_t i = 0; i < n; i++)
        if (avg[i] > 0) mean = mean + avg[i];
    mean /= (double)valid;
    double sq = 0.0;
    for (int64_t i = 0; i < n; i++) sq = sq + (avg[i] > 0 ? (avg[i] - mean) * (avg[i] - mean) : 0.0);
    double sd = sqrt(sq / (double)(valid - 1));
    double thr = mean + std_ratio * sd;
    stats[0] = mean;
    stats[1] = sd;
    stats[2] = thr;
    if (mask)
        for (int64_t i = 0; i < n; i++) mask[i] = (avg[i] > 0 && avg[i] < thr) ? 1 : 0;
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------
 * PCA normals.
 * Call site: app/Core/pointcloud_preprocess.py:109-114 (hybrid radius + max_nn = 30).
 * Upstream: PointCloud::EstimateNormals(param, fast_normal_computation = true) =
 * EstimatePerPointCovariances + utility::ComputeCovariance (one-pass cumulants) +
 * FastEigen3x3 (closed-form robust symmetric 3x3 eigen-solver after Eberly).
 * ---------------------------------------------------------------------------------------- */
static void cross3(const double a[3], const double b[3], double o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
static double dot3(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

/* A = {a00,a01,a02,a11,a12,a22} */
static void eigvec0(const double A[6], double ev, double o[3]) {
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
    const double *c = imax == 0 ? c01 : (imax == 1 ? c02 : c12);
    double s = sqrt(imax == 0 ? d0 : (imax == 1 ? d1 : d2));
    for (int a = 0; a < 3; a++) o[a] = c[a] / s;
}

static void eigvec1(const double A[6], const double e0[3], double ev, double o[3]) {
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
            for (int a = 0; a < 3; a++) o[a] = m01 * U[a] - m00 * V[a];
        } else
            for (int a = 0; a < 3; a++) o[a] = U[a];
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
            for (int a = 0; a < 3; a++) o[a] = m11 * U[a] - m01 * V[a];
        } else
            for (int a = 0; a < 3; a++) o[a] = U[a];
    }
}

/* cov = {c00,c01,c02,c11,c12,c22}; out = eigenvector of the smallest eigenvalue
 * (zero vector when every coefficient's maximum is 0). */
ORC_API void orc_fast_eigen3x3(const double cov[6], double out[3]) {
    double A[6];
    double mc = cov[0];
    for (int i = 1; i < 6; i++)
        if (cov[i] > mc) mc = cov[i];
    if (mc == 0) {
        out[0] = out[1] = out[2] = 0;
        return;
    }
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
                memcpy(out, v2, sizeof v2);
                return;
            }
            eigvec1(A, v2, e1, v1);
            if (e1 < e0 && e1 < e2) {
                memcpy(out, v1, sizeof v1);
                return;
            }
            cross3(v1, v2, out);
        } else {
            eigvec0(A, e0, v0);
            if (e0 < e1 && e0 < e2) {
                memcpy(out, v0, sizeof v0);
                return;
            }
            eigvec1(A, v0, e1, v1);
            if (e1 < e0 && e1 < e2) {
                memcpy(out, v1, sizeof v1);
                return;
            }
            cross3(v0, v1, out);
        }
    } else {
        /* diagonal matrix: axis of the smallest diagonal entry (A *= max_coeff restores cov) */
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

/* Covariance of the listed points, upstream utility::ComputeCovariance: cumulants of
 * x,y,z,xx,xy,xz,yy,yz,zz divided by the count, then C = E[pp^T] - E[p]E[p]^T. */
static void covariance_of(const double *pts, const nb_t *L, int cnt, double cov[6]) {
    double c[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int j = 0; j < cnt; j++) {
        const double *p = pts + 3 * (int64_t)L[j].i;
        c[0] += p[0];
        c[1] += p[1];
        c[2] += p[2];
        c[3] += p[0] * p[0];
        c[4] += p[0] * p[1];
        c[5] += p[0] * p[2];
        c[6] += p[1] * p[1];
        c[7] += p[1] * p[2];
        c[8] += p[2] * p[2];
    }
    for (int i = 0; i < 9; i++) c[i] /= (double)cnt;
    cov[0] = c[3] - c[0] * c[0];
    cov[3] = c[6] - c[1] * c[1];
    cov[5] = c[8] - c[2] * c[2];
    cov[1] = c[4] - c[0] * c[1];
    cov[2] = c[5] - c[0] * c[2];
    cov[4] = c[7] - c[1] * c[2];
}

/* radius <= 0: KDTreeSearchParamKNN(k); radius > 0: KDTreeSearchParamHybrid(radius, k).
 * prior: existing normals (n x 3) or NULL; upstream flips the new normal to agree with it. */
ORC_API int orc_normals(const double *pts, int64_t n, int k, double radius, const double *prior, double *out) {
    if (k < 1) return ORC_BAD_ARG;
    if (n == 0) return ORC_OK;
    void *g = orc_grid_create(pts, n, 0.0, k < 8 ? 2.0 : 4.0);
    if (!g) return ORC_NOMEM;
    double r2 = radius > 0 ? radius * radius : -1.0;
    int err = 0;
#pragma omp parallel
    {
        nb_t *L = (nb_t *)malloc(sizeof(nb_t) * (size_t)k);
        if (!L) {
#pragma omp atomic write
            err = 1;
        }
#pragma omp for schedule(dynamic, 1024)
        for (int64_t i = 0; i < n; i++) {
            if (!L) continue;
            int cnt = grid_query((orc_grid *)g, pts + 3 * i, k, r2, L, 0, NULL);
            double nv[3];
            if (cnt >= 3) {
                double cov[6];
                covariance_of(pts, L, cnt, cov);
                orc_fast_eigen3x3(cov, nv);
            } else {
                double I[6] = {1, 0, 0, 1, 0, 1};
                orc_fast_eigen3x3(I, nv);
            }
            double nn = sqrt(nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2]);
            if (nn == 0.0) {
                if (prior) {
                    nv[0] = prior[3 * i];
                    nv[1] = prior[3 * i + 1];
                    nv[2] = prior[3 * i + 2];
                } else {
                    nv[0] = 0;
                    nv[1] = 0;
                    nv[2] = 1;
                }
            }
            if (prior && nv[0] * prior[3 * i] + nv[1] * prior[3 * i + 1] + nv[2] * prior[3 * i + 2] < 0.0) {
                nv[0] *= -1.0;
                nv[1] *= -1.0;
                nv[2] *= -1.0;
            }
            out[3 * i] = nv[0];
            out[3 * i + 1] = nv[1];
            out[3 * i + 2] = nv[2];
        }
        free(L);
    }
    orc_grid_destroy(g);
    return err ? ORC_NOMEM : ORC_OK;
}

ORC_API int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
