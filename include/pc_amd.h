/*
 * pc_amd.h - C ABI of the MI355X (gfx950) neighbourhood stage.
 *
 * The reference has no FFI of its own: its hot path is eight method calls on
 * open3d.geometry.PointCloud objects made from Python (reference files cited per entry
 * point below).  This header is what a binding for that path binds instead of the Open3D
 * wheel: plain pointers and sizes, no C++ or torch types.
 *
 * Conventions
 *  - Every pointer named *_dev is DEVICE memory owned by the caller; every other pointer is
 *    host memory.  Point arrays are N x 3 row-major.  dtype is PC_F32 or PC_F64.
 *  - Every function returns 0 on success or a negative pc_status; pc_last_error() gives the
 *    message.  Nothing throws or aborts across the boundary.
 *  - Every function selects its context's device itself (the HIP current device is
 *    thread-local and the app calls from worker QThreads, reference
 *    app/Controllers/app_controller.py:428-441, 528-543), enqueues on `stream`
 *    (a hipStream_t passed as void*, NULL = default stream) and, where it returns host
 *    values, synchronises that stream before returning.
 *  - Threads: a context may be shared; its entry points take turns under a mutex held for the whole call (the
 *    app's filter and reconstruction workers overlap, app/Controllers/app_controller.py:428-441, 528-543).
 *    pc_last_error() returns the message of the calling thread's own last failure.
 *  - Streams: pooled scratch goes back to the context when a call returns, ordered behind the work on that
 *    call's stream; a call that arrives on another stream first waits for the previous call's stream, so a
 *    stream handed to one call must stay alive until the next call on that context has begun.
 *  - Neighbour order everywhere: ascending squared distance computed in fp64 as
 *    ((dx*dx)+dy*dy)+dz*dz with dx = q - p; exact ties -> lower point index first.
 */
#ifndef PC_AMD_H
#define PC_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pc_ctx pc_ctx;
typedef struct pc_grid pc_grid;

enum pc_dtype { PC_F32 = 0, PC_F64 = 1 };

enum pc_status {
    PC_OK = 0,
    PC_BAD_ARG = -1,   /* illegal parameter (Open3D raises RuntimeError for the same input) */
    PC_NOMEM = -2,     /* device or host allocation failed */
    PC_HIP_ERROR = -3, /* a HIP runtime call failed */
    PC_OVERFLOW = -4,  /* voxel_size too small for the cloud's extent */
    PC_NO_DEVICE = -5
};

/* Largest k accepted by the neighbour searches. */
#define PC_MAX_K 64

/* ---- context ----------------------------------------------------------------------------
 * pc_create also reads the library's tuning and diagnosis switches from the environment (PC_AMD_DEBUG and the
 * PC_AMD_* knobs named in csrc/pc_common.h, struct PcKnobs), once, into the context: later changes of the
 * environment do not reach a context that exists already, and no other entry point looks at the environment. */
int pc_create(int device, pc_ctx **out);
void pc_destroy(pc_ctx *ctx);
const char *pc_last_error(const pc_ctx *ctx);
const char *pc_status_string(int status);
int pc_abi_version(void);

/* Host helper of the upload path (no device work): dst[i] = (float)src[i] for count values; returns 1 when
 * every value is exactly representable as fp32 (Open3D holds fp64, but clouds loaded from LAS / E57 files were
 * fp32 once, reference app/Infrastructure/pointcloud_io.py:65-72: those can live in HBM as fp32), else 0. */
int pc_host_to_f32_exact(const double *src, int64_t count, float *dst);

/* ---- bounding box / rigid edits ------------------------------------------------------------
 * Replaces PointCloud.get_axis_aligned_bounding_box / translate / scale as used by
 * reference app/Core/pointcloud_preprocess.py:45-53 and :104-105. */
int pc_bbox(pc_ctx *ctx, const void *xyz_dev, int dtype, int64_t n, double out_min[3], double out_max[3],
            void *stream);
/* PointCloud.get_center(): the mean of the points (what translate(t, relative=False) subtracts).  Summed as a
 * fixed two-level tree, so the last bits can differ from Open3D's sequential sum. */
int pc_center(pc_ctx *ctx, const void *xyz_dev, int dtype, int64_t n, double out_center[3], void *stream);
/* In place on an fp64 cloud: if t != NULL p += t; then if has_scale p = (p - c) * s + c. */
int pc_affine(pc_ctx *ctx, double *xyz_dev, int64_t n, const double *t, int has_scale, double s, const double *c,
              void *stream);

/* ---- voxel down-sampling -------------------------------------------------------------------
 * Replaces PointCloud.voxel_down_sample (reference app/Core/pointcloud_preprocess.py:96,
 * app/Infrastructure/filtering_adapter.py:88): voxel index = floor((p - (min - vs/2)) / vs)
 * in fp64, output = per-voxel mean of points / normals / colors, summed in point-index
 * order.  Output order: ascending (iz, iy, ix).  Output buffers hold n rows; *out_m receives
 * the number of voxels.  normals / colors / out_count / out_ijk may be NULL. */
int pc_voxel_downsample(pc_ctx *ctx, const void *xyz_dev, int dtype, const double *normals_dev,
                        const double *colors_dev, int64_t n, double voxel_size, double *out_xyz_dev,
                        double *out_normals_dev, double *out_colors_dev, int32_t *out_count_dev,
                        int32_t *out_ijk_dev, int64_t *out_m, void *stream);
/* No reference counterpart (the reference has no distribution): the voxels of the layers layer_lo <= iz < layer_hi of
 * the voxel grid that pc_voxel_downsample lays over a cloud whose bounding box is box_min / box_max -- the box of the
 * WHOLE cloud, of which xyz_dev may hold all points or any part that contains the layers' points.  Same voxel
 * indices, centroids and order as the corresponding run of pc_voxel_downsample's output; only the points inside the
 * layers are sorted.  A rank of a multi-GPU job voxelises its own z interval (+ halo layers) with it: the voxel
 * order is (iz, iy, ix), so a rank's voxels are one contiguous run of the whole cloud's. */
int pc_voxel_downsample_layers(pc_ctx *ctx, const void *xyz_dev, int dtype, const double *normals_dev,
                               const double *colors_dev, int64_t n, double voxel_size, const double box_min[3],
                               const double box_max[3], int64_t layer_lo, int64_t layer_hi, double *out_xyz_dev,
                               double *out_normals_dev, double *out_colors_dev, int32_t *out_count_dev,
                               int32_t *out_ijk_dev, int64_t *out_m, void *stream);
/* Voxel index triple of every input point (n x 3 int32) and the voxel grid dimensions. */
int pc_voxel_assign(pc_ctx *ctx, const void *xyz_dev, int dtype, int64_t n, double voxel_size,
                    int32_t *out_ijk_dev, int32_t out_dims[3], void *stream);

/* ---- spatial index -------------------------------------------------------------------------
 * Replaces the KD-tree Open3D builds inside estimate_normals / remove_statistical_outlier /
 * remove_radius_outlier / compute_nearest_neighbor_distance (KDTreeFlann::SetGeometry):
 * a uniform grid over the bounding box, points radix-sorted by cell (x fastest), with a
 * cell-start table.  cell_size <= 0 chooses the edge from the point count and k_hint. */
int pc_grid_build(pc_ctx *ctx, const void *xyz_dev, int dtype, int64_t n, double cell_size, int k_hint,
                  pc_grid **out, void *stream);
void pc_grid_destroy(pc_ctx *ctx, pc_grid *grid);
/* info[0..2] = dims, info[3] = cell edge, info[4] = n, info[5] = device bytes held. */
int pc_grid_info(const pc_grid *grid, double info[6]);
/* Sorted-position -> original-index map of the grid (n x int32, device memory owned by the
 * grid).  Sorted position s is the unit the *_range entry points shard over. */
const int32_t *pc_grid_perm_dev(const pc_grid *grid);
/* Copies the original indices of sorted positions [s_begin, s_end) into out_dev. */
int pc_grid_perm_copy(pc_ctx *ctx, const pc_grid *grid, int64_t s_begin, int64_t s_end, int32_t *out_dev,
                      void *stream);

/* ---- exact kNN -----------------------------------------------------------------------------
 * Replaces KDTreeFlann::SearchKNN / SearchHybrid as reached from the calls listed above.
 * queries_dev == NULL: every point of the grid queries itself (the point is its own first
 * neighbour); results are written at the point's ORIGINAL index.  radius < 0: plain kNN;
 * radius >= 0: hybrid search (only neighbours with d2 < radius^2).  Rows are padded with
 * idx = -1, d2 = +inf.  out_d2_dev has element type d2_dtype.  Any out_* may be NULL. */
int pc_knn(pc_ctx *ctx, const pc_grid *grid, const void *queries_dev, int qdtype, int64_t nq, int k, double radius,
           int32_t *out_idx_dev, void *out_d2_dev, int d2_dtype, int32_t *out_cnt_dev, void *stream);
/* Self-query restricted to sorted positions [s_begin, s_end) (multi-GPU query sharding). */
int pc_knn_range(pc_ctx *ctx, const pc_grid *grid, int64_t s_begin, int64_t s_end, int k, double radius,
                 int32_t *out_idx_dev, void *out_d2_dev, int d2_dtype, int32_t *out_cnt_dev, void *stream);

/* ---- consumers fused onto the kNN result -------------------------------------------------------
 * pc_sor_avg: per-point mean neighbour distance of PointCloud.remove_statistical_outlier
 * (reference app/Infrastructure/filtering_adapter.py:91): kNN with k = nb_neighbors
 * including the point itself, avg = sum(sqrt(d2)) / found, written at the original index. */
int pc_sor_avg(pc_ctx *ctx, const pc_grid *grid, int64_t s_begin, int64_t s_end, int k, double *out_avg_dev,
               void *stream);
/* One neighbour search, every consumer: kNN rows (out_idx / out_d2 / out_cnt), the SOR mean
 * distance (out_avg) and the PCA normal (out_normals) of sorted positions [s_begin, s_end),
 * each written at the original index; any output may be NULL.  Equals pc_knn_range +
 * pc_sor_avg + pc_normals with the same k and radius, for a third of the search work. */
int pc_knn_fused(pc_ctx *ctx, const pc_grid *grid, int64_t s_begin, int64_t s_end, int k, double radius,
                 int32_t *out_idx_dev, void *out_d2_dev, int d2_dtype, int32_t *out_cnt_dev, double *out_avg_dev,
                 double *out_normals_dev, const double *prior_dev, void *stream);
/* pc_knn_fused for one rank of a multi-GPU job (queries sharded by sorted position over a replicated index): the
 * same search for sorted positions [s_begin, s_end), but every output has s_end - s_begin rows, row s - s_begin for
 * sorted position s, instead of n rows at the original index (100 M points x k = 32 would be 38 GB of rows per
 * rank).  out_owned_dev (may be NULL) receives the original index of every row.  No prior normals. */
int pc_knn_fused_compact(pc_ctx *ctx, const pc_grid *grid, int64_t s_begin, int64_t s_end, int k, double radius,
                         int32_t *out_idx_dev, void *out_d2_dev, int d2_dtype, int32_t *out_cnt_dev, double *out_avg_dev,
                         double *out_normals_dev, int32_t *out_owned_dev, void *stream);

/* The sums behind the statistical filter's mean and standard deviation as a tree that is the same for any
 * number of ranks: one partial per chunk of PC_SOR_CHUNK consecutive sorted positions, then one fixed pass over
 * the partials in chunk order.  pc_sor_chunk_sums writes the partials (2 doubles per chunk: what = 0: sum of
 * avg > 0 and number of avg >= 0; what = 1: sum over avg > 0 of (avg - mean)^2, and 0) of the chunks that make up
 * [s_begin, s_end), which must begin at a chunk boundary and end at one or at n.  avg_dev: compact != 0: rows
 * s - s_begin (pc_knn_fused_compact); else rows at the original index (pc_sor_avg).  partials_dev holds
 * 2 * pc_sor_chunks(n) doubles; ranks of a multi-GPU job zero it, fill their own chunks and add the arrays
 * elementwise (x + 0 is exact), then every rank calls pc_sor_chunk_finish on the same array: out[0], out[1]. */
#define PC_SOR_CHUNK 4096
int64_t pc_sor_chunks(int64_t n);
int pc_sor_chunk_sums(pc_ctx *ctx, const pc_grid *grid, const double *avg_dev, int compact, int64_t s_begin, int64_t s_end,
                      int what, double mean, double *partials_dev, void *stream);
int pc_sor_chunk_finish(pc_ctx *ctx, const double *partials_dev, int64_t nchunks, double out[2], void *stream);
/* The outcome of the statistical filter as a bit mask over original indices (bit i set = point i is REMOVED: a filter
 * removes a few per cent of the points, so marking those costs a tenth of marking the kept; pc_mask_bytes(n) bytes, a
 * multiple of 4, 4-byte aligned, zeroed by the caller): pc_sor_mask_scatter sets the bits of those of a rank's m compact
 * rows that fail avg[i] > 0 && avg[i] < threshold (bit owned[i]; rows with owned[i] < 0 are skipped); the ranks' masks
 * are added elementwise (their bits are disjoint: the sum is the union); pc_mask_select: ascending indices below n of
 * the CLEAR bits, i.e. the kept points. */
int64_t pc_mask_bytes(int64_t n);
int pc_sor_mask_scatter(pc_ctx *ctx, const double *avg_dev, const int32_t *owned_dev, int64_t m, double threshold,
                        uint8_t *mask_dev, void *stream);
int pc_mask_select(pc_ctx *ctx, const uint8_t *mask_dev, int64_t n, int32_t *out_kept_dev, int64_t *n_kept, void *stream);

/* The statistical filter's sums without any summation order (reference: RemoveStatisticalOutliers behind
 * app/Infrastructure/filtering_adapter.py:91 sums sequentially in index order; no parallel sum can follow that, so
 * the library defines its own, exact one).  Every term -- avg (what = 0) or (avg - mean)^2 (what = 1) of the rows with
 * avg > 0, and with owned >= 0 where owned_dev is given -- becomes a 256-bit fixed-point number of 2^-128 resolution,
 * cut into eight 32-bit digits that are ADDED as integers into words_dev[0..7] (digit j weighs 2^(32 j - 128));
 * words_dev[8] += rows with avg >= 0 (what = 0); words_dev[9] += terms of 2^96 or more (not representable: take
 * pc_sor_sum / pc_sor_sqsum then).  The caller zeroes the PC_SOR_EXACT_WORDS words before a sum begins; ranks of a
 * multi-GPU job add their words elementwise (all-reduce of 80 bytes).  The value is sum_j words[j] * 2^(32 j - 128),
 * the same bits for any number of ranks and any cut of the cloud. */
#define PC_SOR_EXACT_WORDS 10
int pc_sor_exact_sums(pc_ctx *ctx, const double *avg_dev, const int32_t *owned_dev, int64_t m, int what, double mean,
                      int64_t *words_dev, void *stream);

/* Deterministic partial sums over avg[0..n): *sum_pos = sum of avg > 0, *n_valid = #(avg >= 0). */
int pc_sor_sum(pc_ctx *ctx, const double *avg_dev, int64_t n, double *sum_pos, int64_t *n_valid, void *stream);
/* *sq = sum over avg > 0 of (avg - mean)^2. */
int pc_sor_sqsum(pc_ctx *ctx, const double *avg_dev, int64_t n, double mean, double *sq, void *stream);
/* mask[i] = avg > 0 && avg < threshold; kept indices ascending; *n_kept = their number.
 * out_mask_dev / out_kept_dev may be NULL. */
int pc_sor_select(pc_ctx *ctx, const double *avg_dev, int64_t n, double threshold, uint8_t *out_mask_dev,
                  int32_t *out_kept_dev, int64_t *n_kept, void *stream);

/* pc_sor_select for callers that cannot allocate (tight serving loops, harnesses that forbid hipMalloc).
 * No counterpart in the reference.  Same arguments and results as pc_sor_select, but the flag bytes and block
 * counts live in a static device array of PC_SOR_SELECT_STATIC_MAX_N + 8 KiB bytes that every GPU the library
 * runs on holds until the process ends: the call performs no hipMalloc / hipFree.  Like pc_sor_select it
 * synchronises the stream to return *n_kept, so it is not for use inside a HIP graph capture.
 * n > PC_SOR_SELECT_STATIC_MAX_N is rejected with PC_BAD_ARG.  All calls on one GPU share that array
 * (whatever their context) and take turns under a per-device mutex. */
#define PC_SOR_SELECT_STATIC_MAX_N (4 * 1024 * 1024)
int pc_sor_select_static(pc_ctx *ctx, const double *avg_dev, int64_t n, double threshold, uint8_t *out_mask_dev,
                         int32_t *out_kept_dev, int64_t *n_kept, void *stream);

/* PCA normals of PointCloud.estimate_normals(fast_normal_computation=True)
 * (reference app/Core/pointcloud_preprocess.py:109-114): neighbours = kNN (radius < 0) or
 * hybrid search; fewer than 3 -> (0,0,1); else one-pass covariance and the closed-form
 * symmetric 3x3 eigen-solver; prior_dev (n x 3, may be NULL) flips the sign to agree. */
int pc_normals(pc_ctx *ctx, const pc_grid *grid, int64_t s_begin, int64_t s_end, int k, double radius,
               const double *prior_dev, double *out_normals_dev, void *stream);

/* Neighbour count of PointCloud.remove_radius_outlier (filtering_adapter.py:94):
 * points with d2 < radius^2 (strict, self included).  queries_dev == NULL: self-query over
 * sorted positions [s_begin, s_end), written at the original index. */
int pc_radius_count(pc_ctx *ctx, const pc_grid *grid, const void *queries_dev, int qdtype, int64_t nq,
                    int64_t s_begin, int64_t s_end, double radius, int32_t *out_cnt_dev, void *stream);

/* The selection step of PointCloud.remove_radius_outlier (filtering_adapter.py:94): mask[i] = cnt[i] > nb_points
 * over the counts of pc_radius_count (or the capped counts of pc_knn); kept indices ascending; *n_kept = their
 * number.  out_mask_dev / out_kept_dev may be NULL. */
int pc_count_select(pc_ctx *ctx, const int32_t *cnt_dev, int64_t n, int32_t nb_points, uint8_t *out_mask_dev,
                    int32_t *out_kept_dev, int64_t *n_kept, void *stream);

/* PointCloud.compute_nearest_neighbor_distance (filtering_adapter.py:49,
 * app/Core/mesh_reconstruction.py:112): sqrt of the second-smallest d2, 0 if none. */
int pc_nn_distance(pc_ctx *ctx, const pc_grid *grid, int64_t s_begin, int64_t s_end, double *out_dev, void *stream);

/* Row gather dst[i] = src[idx[i]] for m rows of 3 doubles (PointCloud.select_by_index). */
int pc_gather_rows(pc_ctx *ctx, const double *src_dev, const int32_t *idx_dev, int64_t m, double *dst_dev,
                   void *stream);

/* The inverse: dst[idx[i]] = src[i] (a rank's compact rows put back at their original indices; idx distinct). */
int pc_scatter_rows(pc_ctx *ctx, const double *src_dev, const int32_t *idx_dev, int64_t m, double *dst_dev,
                    void *stream);
/* The same for rows of 3 floats (clouds kept as fp32 because every coordinate is fp32-representable). */
int pc_gather_rows_f32(pc_ctx *ctx, const float *src_dev, const int32_t *idx_dev, int64_t m, float *dst_dev,
                       void *stream);

/* ---- slabs: one rank of a multi-GPU job indexes only its part of the cloud --------------------------------
 * The reference has no distribution of any kind; the calls of this section serve the query sharding of
 * remove_statistical_outlier / estimate_normals (filtering_adapter.py:91, pointcloud_preprocess.py:109-114)
 * where sorting all n points on every rank would bound the speed-up.  A rank takes the points whose coordinate
 * along `axis` lies in [own_lo - halo, own_hi + halo) (pc_select_slab + pc_gather_rows*), indexes that subset
 * (pc_grid_build), searches the sorted positions of its own interval (pc_grid_axis_range + pc_knn_fused_compact)
 * and lets pc_slab_finish translate the rows back to indices of the whole cloud and count the answers that the
 * subset cannot vouch for (k-th distance beyond the nearer open face of the slab): those are repeated on a
 * wider slab or a full index, so results stay exact. */
/* Counts of the n points per bin of [lo, hi] along axis (nbins <= 4096 equal bins, ends clamped; lo / hi point to one
 * value each), or, with axis < 0, along all three axes at once (lo / hi point to three values, out_counts holds
 * 3 * nbins counters, axis-major), into host memory: what the cuts between the ranks' intervals and the extent of the
 * cloud's bulk are read from.  Synchronises the stream. */
int pc_axis_histogram(pc_ctx *ctx, const void *xyz_dev, int dtype, int64_t n, int axis, const double *lo, const double *hi,
                      int nbins, uint32_t *out_counts, void *stream);
/* Ascending indices of the points with lo <= coordinate < hi along axis (+-inf allowed); out_index_dev holds up to n
 * entries, *m receives their number.  Synchronises the stream. */
int pc_select_slab(pc_ctx *ctx, const void *xyz_dev, int dtype, int64_t n, int axis, double lo, double hi,
                   int32_t *out_index_dev, int64_t *m, void *stream);
/* The range of sorted positions made of the whole z layers of cells that hold every point with lo <= z <= hi (z is
 * the slowest digit of the cell keys: a z interval is one range).  Synchronises the stream. */
int pc_grid_axis_range(pc_ctx *ctx, const pc_grid *grid, double lo, double hi, int64_t *s_begin, int64_t *s_end,
                       void *stream);
/* Renames the points of an index built over a subset of a cloud: point i of the subset becomes labels[i], its
 * index in the whole cloud (labels ascending, below index_space = the whole cloud's size).  From then on every
 * "original index" the searches of this grid report -- neighbour indices, owned rows -- is an index of the whole
 * cloud, and outputs written "at the original index" need index_space rows.  Once per grid, before its first
 * search.  labels_dev is read by kernels on `stream` and must stay alive until they have run. */
int pc_grid_relabel(pc_ctx *ctx, pc_grid *grid, const int32_t *labels_dev, int64_t index_space, void *stream);
/* After pc_knn_fused_compact over a slab subset (sub_xyz: the subset's rows, sub_index: their indices in the whole
 * cloud; m_rows result rows; after pc_grid_relabel: sub_xyz = the whole cloud and sub_index = NULL, the rows carry
 * indices of the whole cloud already): rows whose query has own_lo <= z < own_hi get owned = index in the whole cloud and
 * neighbour indices translated likewise, all others owned = -1 (rows to ignore).  open_lo / open_hi: the subset's
 * z faces beyond which the cloud has more points (-inf / +inf where it has none).  avg_full (may be NULL, else n
 * doubles): avg_rows of the own rows written at their index in the whole cloud.  *n_own: own rows; *n_uncertain:
 * own rows whose k-th squared distance is not below the squared distance to the nearer open face (or that found
 * fewer than k although points may hide beyond a face within the radius); out_uncertain_rows (may be NULL, else m_rows
 * entries): those rows, ascending.  cnt_dev may be NULL (every row has k).
 * Synchronises the stream. */
int pc_slab_finish(pc_ctx *ctx, const void *sub_xyz_dev, int dtype, const int32_t *sub_index_dev, int64_t m_rows, int k,
                   double radius, int32_t *idx_dev, const void *d2_dev, int d2_dtype, const int32_t *cnt_dev,
                   int32_t *owned_dev, double own_lo, double own_hi, double open_lo, double open_hi,
                   const double *avg_rows_dev, double *avg_full_dev, int32_t *out_uncertain_rows_dev, int64_t *n_own,
                   int64_t *n_uncertain, void *stream);
/* The repeat of the answers no slab could vouch for: every rank answers ALL of them against its own slab (pc_knn with
 * the queries as an array), the rows travel, and pc_merge_rows puts them together: idx_all / d2_all hold world x nq x k
 * entries (rank-major; each row ascending in (d2, index), cnt_all its valid entries: world x nq), the outputs the nq
 * merged rows -- the k nearest of the union in the same order, a point that two ranks index (halos overlap) taken once.
 * Exact because the ranks' subsets cover the cloud: a true neighbour is among the k nearest of any subset holding it. */
int pc_merge_rows(pc_ctx *ctx, int world, int64_t nq, int k, const int32_t *idx_all_dev, const double *d2_all_dev,
                  const int32_t *cnt_all_dev, int32_t *out_idx_dev, double *out_d2_dev, int32_t *out_cnt_dev, void *stream);
/* The searches' consumers for finished rows: out_avg[q] = mean neighbour distance (remove_statistical_outlier,
 * filtering_adapter.py:91) and out_normals[q] = PCA normal (estimate_normals, pointcloud_preprocess.py:109-114) of query
 * point query_index[q] of the cloud xyz from its row idx[q] (cnt[q] valid entries, indices into xyz), computed as the
 * fused search computes them.  Either output may be NULL. */
int pc_rows_consumers(pc_ctx *ctx, const void *xyz_dev, int dtype, const int32_t *query_index_dev, int64_t nq, int k,
                      const int32_t *idx_dev, const int32_t *cnt_dev, double *out_avg_dev, double *out_normals_dev,
                      void *stream);

/* PointCloud.orient_normals_consistent_tangent_plane(k) (app/Core/pointcloud_preprocess.py:115): turns the
 * n normals (n x 3 fp64, in place) so that they agree along the minimum spanning tree of the neighbour graph
 * given by knn_idx (n x k rows of pc_knn over the same cloud; knn_cnt = valid entries per row, or NULL: all),
 * edge weight 1 - |n_i . n_j|; every connected piece is turned so that its highest point (xyz, n x 3 of
 * dtype) looks towards +z.  *out_pieces = number of connected pieces.  Needs n * k < 2^32. */
int pc_orient_normals(pc_ctx *ctx, const void *xyz_dev, int dtype, double *normals_dev, const int32_t *knn_idx_dev,
                      const int32_t *knn_cnt_dev, int64_t n, int k, int64_t *out_pieces, void *stream);

/* One linearised step of point-to-plane ICP.  The reference only advertises ICP (app/UI/main_window.py:58,
 * 180-194 lead to a stub); the arithmetic is Open3D's TransformationEstimationPointToPlane: for every source
 * point i with a correspondence j = corr_idx[i] >= 0 and corr_d2[i] < max_d2: r = (s_i - t_j) . n_j,
 * J = [s_i x n_j, n_j].  out_terms (host) = the 21 upper-triangle entries of sum J J^T (row by row), the 6 of
 * sum J r, sum of corr_d2 and the number of correspondences used.  src: m x 3 fp64 (already transformed);
 * corr_idx / corr_d2: the rows of pc_knn(k = 1) for src against the target.  Synchronises the stream. */
int pc_icp_point_to_plane_terms(pc_ctx *ctx, const double *src_dev, int64_t m, const void *tgt_xyz_dev, int tgt_dtype,
                                const double *tgt_normals_dev, const int32_t *corr_idx_dev, const void *corr_d2_dev,
                                int d2_dtype, double max_d2, double out_terms[29], void *stream);

/* PointCloud.transform: p <- R p + t over n points (fp64, in place); T = 4 x 4 row-major, last row ignored. */
int pc_transform(pc_ctx *ctx, double *xyz_dev, int64_t n, const double T[16], void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PC_AMD_H */
