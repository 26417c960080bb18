// Shared host-side declarations of the gfx950 neighbourhood-stage library.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <string>
#include <vector>

#include "pc_amd.h"

constexpr int EAGER_LEVELS = 4;  // levels of a multi-level index built with it; the rest are made when a search gets that far
constexpr int MAX_LEVELS = 8;  // resolution levels of one spatial index (pc_grid::coarse chain)
constexpr int MAX_FINER = 3;   // grids of half the cell edge each, made for the queries too crowded for level 0 (pc_grid::finer chain)

#define PC_WAVE 64  // gfx950 wavefront width (hard-coded: not a host compile-time constant)

// Tuning and diagnosis switches of the environment (PC_AMD_*), read once, when a context is created, and kept
// with it: no call on the hot path looks at the environment.
struct PcKnobs {
    bool debug = false;            // PC_AMD_DEBUG: per-pass log on stderr
    double cell_scale = 0.0;       // PC_AMD_CELL_SCALE: scales the automatic cell edge (0: off)
    bool no_halves = false;        // PC_AMD_NO_HALVES
    int eager_levels = 0;          // PC_AMD_EAGER_LEVELS: levels built with the index (0: EAGER_LEVELS, or 1 from lazy_min points on)
    int64_t lazy_min = 1ll << 20;  // PC_AMD_LAZY_MIN: clouds of this size and more build the finest level only; searches make the levels worth a pass
    int lazy_div = 0;              // PC_AMD_LAZY_DIV: a further level is made while more than n / this many queries are open (0: default)
    int max_levels = 0;            // PC_AMD_MAX_LEVELS: levels of a multi-level index at most (0: MAX_LEVELS)
    double seed_scale = 0.0;       // PC_AMD_SEED_SCALE: the tile kernel's starting threshold over the certification bound (0: default)
    int uneven_cap = 0;            // PC_AMD_UNEVEN_CAP: candidates an uneven level's tile stages (0: default)
    int tile_waves = 0;            // PC_AMD_TILE_WAVES (0: default)
    int tile[3] = {0, 0, 0};       // PC_AMD_TILE "tx,ty,tz" (0: automatic)
    int uneven_tile0 = 0, uneven_tile1 = 0;  // PC_AMD_UNEVEN_TILE0 / _TILE1 (0: default)
    int thin_tile = 0;             // PC_AMD_THIN_TILE: tile edge (cells) of a coarser level that answers millions of queries (0: default)
    int uneven_waves = 0;          // PC_AMD_UNEVEN_WAVES (0: default)
    bool exact_scan_only = false;  // PC_AMD_EXACT_SCAN_ONLY: one-query-per-lane fp64 scan only (A/B switch)
    bool no_batch = false;         // PC_AMD_NO_BATCH
    int batch_ks = 34;             // PC_AMD_BATCH_KS
    int64_t batch_ratio = 12;      // PC_AMD_BATCH_RATIO
    bool no_hints = false;         // PC_AMD_NO_HINTS
    bool no_mixed = false;         // PC_AMD_NO_MIXED
    int64_t mixed_min = 1ll << 20; // PC_AMD_MIXED_MIN
    uint32_t full_tile = 32;       // PC_AMD_FULL_TILE
    bool no_finer = false;         // PC_AMD_NO_FINER
    bool no_postpone = false;      // PC_AMD_NO_POSTPONE: sparse tiles' queries are not postponed tile by tile before the first list pass (A/B switch)
    double dense_factor = 16.0;    // PC_AMD_DENSE_FACTOR: a grid keeps a table of cell offsets while it has at most this many cells per point
    bool no_probe = false;         // PC_AMD_NO_PROBE: every cell edge tried is sorted, not counted first (A/B switch)
    bool no_mark_tiles = false;    // PC_AMD_NO_MARK_TILES: the occupied tiles of a sparse grid are always found by sorting tile ids (A/B switch)
    bool no_ring = false;          // PC_AMD_NO_RING: no lane-per-query ring pass before the wave-per-query kernel (A/B switch)
    int64_t ring_min = 65536;      // PC_AMD_RING_MIN: open queries from which that pass is worth its list
    bool no_bucket_build = false;  // PC_AMD_NO_BUCKET_BUILD: always the general index build (A/B switch)
};
PcKnobs pc_read_knobs();

struct pc_ctx {
    int device = 0;
    // grow-only cache of device blocks so that per-frame calls do not hipMalloc
    struct Block {
        void *p;
        size_t bytes;
        bool used;
    };
    std::vector<Block> pool;
    size_t pool_bytes = 0;
    void *pinned = nullptr;  // 4 KiB of page-locked host memory for small device-to-host readbacks
    // Entry points of the C ABI hold this for their whole call: the app's filter and reconstruction workers
    // (two threads, app/Controllers/app_controller.py:428-441 and 528-543) may call into one context at once.
    std::recursive_mutex mu;
    // The stream of the previous call: pooled blocks go back to the pool when a call returns, on the
    // understanding that the next user is ordered behind it on the same stream.  A call that arrives on
    // another stream waits for the previous one first.
    hipStream_t last_stream = nullptr;
    bool has_last_stream = false;
    PcKnobs knobs;  // the environment's switches as pc_create found them

    void *alloc(size_t bytes);
    void release(void *p);
    void trim();
};

int pc_fail(pc_ctx *ctx, int status, const char *what, hipError_t e = hipSuccess);



struct PcLock {
    pc_ctx *c;
    explicit PcLock(pc_ctx *ctx) : c(ctx) {
        if (c) c->mu.lock();
    }
    PcLock(pc_ctx *ctx, void *stream) : c(ctx) {
        if (!c) return;
        c->mu.lock();
        const hipStream_t st = (hipStream_t)stream;
        if (c->has_last_stream && c->last_stream != st) {
            (void)hipSetDevice(c->device);
            (void)hipStreamSynchronize(c->last_stream);
        }
        c->last_stream = st;
        c->has_last_stream = true;
    }
    ~PcLock() {
        if (c) c->mu.unlock();
    }
    PcLock(const PcLock &) = delete;
    PcLock &operator=(const PcLock &) = delete;
};

#define PC_HIP(call)                                                            \
    do {                                                                        \
        hipError_t e__ = (call);                                                \
        if (e__ != hipSuccess) return pc_fail(ctx, PC_HIP_ERROR, #call, e__);   \
    } while (0)

#define PC_TRY(call)               \
    do {                           \
        int rc__ = (call);         \
        if (rc__ != PC_OK) return rc__; \
    } while (0)

// RAII handle on a pooled device block
struct Scratch {
    pc_ctx *ctx;
    void *p = nullptr;
    Scratch(pc_ctx *c, size_t bytes) : ctx(c) { p = c->alloc(bytes ? bytes : 16); }
    ~Scratch() {
        if (p) ctx->release(p);
    }
    Scratch(const Scratch &) = delete;
    Scratch &operator=(const Scratch &) = delete;
    template <typename T>
    T *as() const {
        return reinterpret_cast<T *>(p);
    }
    bool ok() const { return p != nullptr; }
};

// Points sorted by cell, one record per point: coordinates + original index.
struct __attribute__((aligned(16))) Rec32 {
    float x, y, z;
    int32_t idx;
};
struct __attribute__((aligned(32))) Rec64 {
    double x, y, z;
    int64_t idx;
};

struct GridParams {
    double o[3];   // origin (grid min, or voxel min bound)
    double s;      // kNN grid: 1/cell ; voxel grid: voxel size
    double cell;   // cell edge
    int32_t dim[3];
};

// 1 for a point in the upper x-half of its cell c (cell index along x, clamped like the cell keys).  The grid
// build orders the records of a cell by this bit and the tile kernel splits its row segments by it: both must
// evaluate exactly this expression on the same fp64 value.
__host__ __device__ inline uint32_t x_half_of(double x, const GridParams &gp, int c) {
    return ((x - gp.o[0]) * gp.s - (double)c) >= 0.5 ? 1u : 0u;
}

// Cell key -> first sorted position of that cell.  Dense form: a table of ncell + 1 offsets.
// Sparse form (huge, mostly empty grids): no table; the sorted per-point cell keys are searched,
// inside the key's (y, z) row when a row directory exists (a handful of steps instead of log2 n).
struct CellIndex {
    const uint32_t *table;      // ncell + 1 entries, or NULL
    const uint32_t *keys;       // n sorted cell keys (sparse form)
    uint32_t n;
    const uint32_t *row_start;  // sparse form: first sorted position of every (y, z) row, + 2 end entries; or NULL
    uint32_t nx;                // cells per row
};

struct pc_grid {
    int64_t n = 0;
    int dtype = PC_F32;
    GridParams gp{};
    double bb_min[3] = {0, 0, 0}, bb_max[3] = {0, 0, 0};  // the cloud's bounding box
    void *rec = nullptr;          // Rec32[n] or Rec64[n]
    uint32_t *cell_start = nullptr;  // ncell + 1 (dense form), or NULL
    uint32_t *keys = nullptr;        // n sorted cell keys (sparse form), or NULL
    uint32_t *row_start = nullptr;   // sparse form: row directory (rows + 2 entries), or NULL
    CellIndex index() const { return CellIndex{cell_start, keys, (uint32_t)n, row_start, (uint32_t)gp.dim[0]}; }
    int32_t *perm = nullptr;      // sorted position -> original index
    int64_t ncell = 0;
    size_t bytes = 0;
    mutable int64_t occupied = 0;     // number of non-empty cells (0: not measured)
    mutable double seen = 0.0;        // points in the cell of a random point (point-weighted occupancy)
    // levels built from another level leave the two sums behind those numbers on the device and are read
    // on first need (pc_grid_stats): one host round trip less per level
    mutable unsigned long long *stats_dev = nullptr;
    mutable bool stats_pending = false;
    // tile shape of the cooperative kNN kernel, chosen on first use (0: not chosen yet)
    mutable int tile[3] = {0, 0, 0};
    mutable int tile_cap = 0;
    mutable int tile_waves = 1;  // waves per workgroup of that kernel
    mutable uint32_t *tile_list = nullptr;  // sparse form: linear ids of the occupied tiles
    mutable int64_t tile_count = 0;
    // the listed tiles that hold enough points to fill a good part of a wave (made on first need;
    // full_count < 0: not made yet) and the points they hold
    mutable uint32_t *full_tiles = nullptr;
    mutable int64_t full_count = -1;
    mutable int64_t full_points = 0;
    mutable std::vector<uint32_t> full_host;  // host copy of full_tiles (ascending): sub-ranges for query shards
    // the same cloud on a 2x coarser grid (and that one's own `coarse`): built for clouds that
    // fill their box unevenly, used by the exact re-run of queries whose surroundings are sparse
    mutable pc_grid *coarse = nullptr;  // (the first EAGER_LEVELS levels are built with the index, coarser ones on first need)
    // ... and on a 2x finer grid (and that one's own `finer`): made on first need, for the points
    // whose surroundings are too crowded for every level above; NULL until then
    mutable pc_grid *finer = nullptr;
    mutable bool finer_tried = false;  // a finer grid was asked for and cannot exist (cell-key range)
    int level = 0;        // 0: finest
    int k_hint = 0;       // the neighbour count the index was built for (levels made from another level: that level's)
    // records of one cell are ordered low x-half first (x offset inside the cell < half an edge), then by
    // original index: the tile kernel of evenly filled fp32 grids walks 2.5 cells per row instead of 3
    bool halves = false;
    bool uneven = false;  // part of a multi-level index (set on every level)
    // Original indices (Rec::idx, perm) are below this; 0: below n.  pc_grid_relabel renames the points of an index
    // built over a subset of a cloud to their indices in the whole cloud, which are below the whole cloud's size:
    // arrays indexed by original index (per-point flags and bounds of the level passes, output slots) are sized by it.
    int64_t index_space = 0;
};
inline int64_t pc_index_space(const pc_grid *g) { return g->index_space > 0 ? g->index_space : g->n; }

// pc_grid.hip
// fills g->occupied / g->seen if they are still on the device (synchronises then)
int pc_grid_stats(pc_ctx *ctx, const pc_grid *g, hipStream_t st);
// the same cloud on cells of half the edge (cached in g->finer); *out = NULL if the key range forbids it
int pc_grid_finer(pc_ctx *ctx, const pc_grid *g, hipStream_t st, const pc_grid **out);
// the next coarser level of a multi-level index (cells of twice the edge; kept in g->coarse); *out = NULL where there
// is none to make (MAX_LEVELS reached, or the grid is down to four cells an axis)
int pc_grid_coarser(pc_ctx *ctx, const pc_grid *g, hipStream_t st, const pc_grid **out);
// level `level` of a multi-level index (cells 2^level times the finest's), made from the records of `last`, the coarsest
// level there is, and hung behind it: levels in between are left out; *out = NULL where there is none to make
int pc_grid_level(pc_ctx *ctx, const pc_grid *last, int level, hipStream_t st, const pc_grid **out);
// (qrows_out, if given: the row directory of the sorted query keys -- first sorted position of every (y, z) row of g's cells,
// dim[1] * dim[2] + 2 entries -- so that a look-up among the queries stays inside one row)
int pc_bin_queries(pc_ctx *ctx, const pc_grid *g, const void *q_dev, int64_t nq, void *qrec_out, uint32_t *qkeys_out,
                   uint32_t *qrows_out, hipStream_t st);

// pc_knn_tile.hip
// The tile of the cooperative kNN kernel for a cloud that fills its box with `occ` points per cell on a grid of `dim`
// cells: the fullest shape whose expected population (+ 1.5 sigma) fits the workgroup's lanes; *waves = waves per
// workgroup, *pop = expected queries per tile.  Pure arithmetic: the grid build uses it to pick, among cell edges
// within a few per cent of each other, the one whose tiles divide the grid best.
void pc_dense_tile_shape(const pc_ctx *ctx, double occ, const int32_t dim[3], int tile[3], int *waves, double *pop);

// pc_build.hip
// records, index map and cell table of an evenly filled fp32 cloud on a dense grid (see the file's head)
int pc_build_bucketed(pc_ctx *ctx, const float *xyz, int64_t n, const GridParams &gp, int64_t ncell, int halfbit,
                      Rec32 *rec_out, int32_t *perm_out, uint32_t *cell_start_out, unsigned long long *stats_dev,
                      hipStream_t st, int *taken);

// pc_sort.hip
// device address of this GPU's copy of the static scratch array behind pc_sor_select_static (once per device)
int pc_static_scratch_init(int device);
int pc_radix_sort_pairs(pc_ctx *ctx, uint64_t *keys, uint32_t *vals, uint64_t *keys_tmp, uint32_t *vals_tmp,
                        int64_t n, int key_bits, hipStream_t st, bool *result_in_tmp);
int pc_radix_sort_pairs32(pc_ctx *ctx, uint32_t *keys, uint32_t *vals, uint32_t *keys_tmp, uint32_t *vals_tmp,
                          int64_t n, int key_bits, hipStream_t st, bool *result_in_tmp);
int pc_compact_flags(pc_ctx *ctx, const uint8_t *flags, int64_t n, int32_t *out_idx, int64_t *count, hipStream_t st);
int pc_exclusive_scan_u32(pc_ctx *ctx, uint32_t *data, int64_t n, hipStream_t st);
// flags (0/1 bytes) -> ascending indices of the set flags; *count on host (synchronises)
int pc_compact_flags(pc_ctx *ctx, const uint8_t *flags, int64_t n, int32_t *out_idx, int64_t *count, hipStream_t st);
