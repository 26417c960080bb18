// Context, bounding box, rigid edits, voxel down-sampling and the uniform-grid spatial index.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <thread>

#include "pc_common.h"

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
// The message of the last failed call, kept per calling thread: two worker threads may share one context, and a
// thread asks for the text of ITS failure (pc_last_error hands out a pointer into this thread's own string).
static thread_local std::string tl_err;

int pc_fail(pc_ctx *ctx, int status, const char *what, hipError_t e) {
    tl_err = std::string(pc_status_string(status)) + ": " + what;
    if (e != hipSuccess) tl_err += std::string(" (") + hipGetErrorString(e) + ")";
    (void)ctx;
    return status;
}

void *pc_ctx::alloc(size_t bytes) {
    bytes = (bytes + 255) & ~size_t(255);
    int best = -1;
    for (size_t i = 0; i < pool.size(); i++)
        if (!pool[i].used && pool[i].bytes >= bytes && pool[i].bytes <= 2 * bytes + (1u << 20))
            if (best < 0 || pool[i].bytes < pool[best].bytes) best = (int)i;
    if (best >= 0) {
        pool[best].used = true;
        return pool[best].p;
    }
    void *p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) {
        trim();  // drop cached free blocks and retry once
        if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
    }
    pool.push_back({p, bytes, true});
    pool_bytes += bytes;
    return p;
}

void pc_ctx::release(void *p) {
    for (auto &b : pool)
        if (b.p == p) {
            b.used = false;
            return;
        }
}

void pc_ctx::trim() {
    std::vector<Block> keep;
    for (auto &b : pool) {
        if (b.used)
            keep.push_back(b);
        else {
            (void)hipFree(b.p);
            pool_bytes -= b.bytes;
        }
    }
    pool.swap(keep);
}

PcKnobs pc_read_knobs() {
    {
        PcKnobs k;
        auto on = [](const char *name) { return getenv(name) != nullptr; };
        auto num = [](const char *name, long long dflt) {
            const char *e = getenv(name);
            return e ? atoll(e) : dflt;
        };
        k.debug = on("PC_AMD_DEBUG");
        if (const char *e = getenv("PC_AMD_CELL_SCALE")) k.cell_scale = atof(e);
        k.no_halves = on("PC_AMD_NO_HALVES");
        if (const char *e = getenv("PC_AMD_SEED_SCALE")) k.seed_scale = atof(e);
        if (const char *e = getenv("PC_AMD_MAX_LEVELS")) k.max_levels = atoi(e);
        if (const char *e = getenv("PC_AMD_LAZY_DIV")) k.lazy_div = atoi(e);
        if (const char *e = getenv("PC_AMD_EAGER_LEVELS")) k.eager_levels = atoi(e);
        k.uneven_cap = (int)num("PC_AMD_UNEVEN_CAP", 0);
        k.tile_waves = (int)num("PC_AMD_TILE_WAVES", 0);
        if (const char *e = getenv("PC_AMD_TILE")) {
            int v[3];
            if (sscanf(e, "%d,%d,%d", &v[0], &v[1], &v[2]) == 3)
                for (int a = 0; a < 3; a++) k.tile[a] = v[a];
        }
        k.uneven_tile0 = (int)num("PC_AMD_UNEVEN_TILE0", 0);
        k.uneven_tile1 = (int)num("PC_AMD_UNEVEN_TILE1", 0);
        k.uneven_waves = (int)num("PC_AMD_UNEVEN_WAVES", 0);
        k.exact_scan_only = on("PC_AMD_EXACT_SCAN_ONLY");
        k.no_batch = on("PC_AMD_NO_BATCH");
        k.batch_ks = (int)num("PC_AMD_BATCH_KS", 34);
        k.batch_ratio = num("PC_AMD_BATCH_RATIO", 12);
        k.no_hints = on("PC_AMD_NO_HINTS");
        k.no_mixed = on("PC_AMD_NO_MIXED");
        k.mixed_min = num("PC_AMD_MIXED_MIN", 1ll << 20);
        k.full_tile = (uint32_t)num("PC_AMD_FULL_TILE", 32);
        k.no_finer = on("PC_AMD_NO_FINER");
        k.no_bucket_build = on("PC_AMD_NO_BUCKET_BUILD");
        k.no_ring = on("PC_AMD_NO_RING");
        k.thin_tile = (int)num("PC_AMD_THIN_TILE", 0);
        k.no_probe = on("PC_AMD_NO_PROBE");
        k.no_mark_tiles = on("PC_AMD_NO_MARK_TILES");
        if (const char *e = getenv("PC_AMD_DENSE_FACTOR")) k.dense_factor = atof(e);
        k.no_postpone = on("PC_AMD_NO_POSTPONE");
        k.lazy_min = num("PC_AMD_LAZY_MIN", 1ll << 20);
        k.ring_min = num("PC_AMD_RING_MIN", 65536);
        return k;
    }
}

extern "C" {

int pc_abi_version(void) { return 1; }

const char *pc_status_string(int status) {
    switch (status) {
        case PC_OK: return "ok";
        case PC_BAD_ARG: return "bad argument";
        case PC_NOMEM: return "out of memory";
        case PC_HIP_ERROR: return "HIP error";
        case PC_OVERFLOW: return "voxel_size is too small";
        case PC_NO_DEVICE: return "no such device";
        default: return "unknown status";
    }
}

int pc_create(int device, pc_ctx **out) {
    if (!out) return PC_BAD_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return PC_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return PC_NO_DEVICE;
    pc_ctx *c = new (std::nothrow) pc_ctx();
    if (!c) return PC_NOMEM;
    c->device = device;
    if (hipHostMalloc(&c->pinned, 4096, hipHostMallocDefault) != hipSuccess) {
        delete c;
        return PC_NOMEM;
    }
    if (pc_static_scratch_init(device) != PC_OK) {
        (void)hipHostFree(c->pinned);
        delete c;
        return PC_HIP_ERROR;
    }
    c->knobs = pc_read_knobs();  // the environment is read here, once per context
    *out = c;
    return PC_OK;
}

void pc_destroy(pc_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    for (auto &b : ctx->pool) (void)hipFree(b.p);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    delete ctx;
}

const char *pc_last_error(const pc_ctx *ctx) {
    (void)ctx;
    return tl_err.empty() ? "no error" : tl_err.c_str();
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
namespace {

template <typename T>
__global__ __launch_bounds__(256) void k_bbox(const T *__restrict__ xyz, int64_t n, double *__restrict__ partial) {
    double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
#pragma unroll
        for (int a = 0; a < 3; a++) {
            double v = (double)xyz[3 * i + a];
            mn[a] = fmin(mn[a], v);
            mx[a] = fmax(mx[a], v);
        }
    }
    __shared__ double red[4][6];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int a = 0; a < 3; a++) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mn[a] = fmin(mn[a], __shfl_down(mn[a], o, 64));
            mx[a] = fmax(mx[a], __shfl_down(mx[a], o, 64));
        }
        if (lane == 0) {
            red[wave][a] = mn[a];
            red[wave][3 + a] = mx[a];
        }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        double v = red[0][threadIdx.x];
        for (int w = 1; w < 4; w++) v = threadIdx.x < 3 ? fmin(v, red[w][threadIdx.x]) : fmax(v, red[w][threadIdx.x]);
        partial[(size_t)blockIdx.x * 6 + threadIdx.x] = v;
    }
}

__global__ __launch_bounds__(64) void k_bbox_final(const double *__restrict__ partial, int nb, double *__restrict__ out) {
    const int lane = threadIdx.x;
    double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int b = lane; b < nb; b += 64) {
#pragma unroll
        for (int a = 0; a < 3; a++) {
            mn[a] = fmin(mn[a], partial[(size_t)b * 6 + a]);
            mx[a] = fmax(mx[a], partial[(size_t)b * 6 + 3 + a]);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; a++) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mn[a] = fmin(mn[a], __shfl_down(mn[a], o, 64));
            mx[a] = fmax(mx[a], __shfl_down(mx[a], o, 64));
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int a = 0; a < 3; a++) {
            out[a] = mn[a];
            out[3 + a] = mx[a];
        }
    }
}

// Per-block partial sums of the coordinates (fixed block count and fixed trees: the same bits on every run).
template <typename T>
__global__ __launch_bounds__(256) void k_coord_sums(const T *__restrict__ xyz, int64_t n, double *__restrict__ partial) {
    double s[3] = {0.0, 0.0, 0.0};
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t b = (int64_t)blockIdx.x * per, e = b + per < n ? b + per : n;
    for (int64_t i = b + threadIdx.x; i < e; i += 256) {
#pragma unroll
        for (int a = 0; a < 3; a++) s[a] += (double)xyz[3 * i + a];
    }
    __shared__ double red[3][256];
#pragma unroll
    for (int a = 0; a < 3; a++) red[a][threadIdx.x] = s[a];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
#pragma unroll
            for (int a = 0; a < 3; a++) red[a][threadIdx.x] += red[a][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x < 3) partial[3 * blockIdx.x + threadIdx.x] = red[threadIdx.x][0];
}

__global__ __launch_bounds__(64) void k_coord_sums_final(const double *__restrict__ partial, int nb, double *__restrict__ out) {
    double s[3] = {0.0, 0.0, 0.0};
    for (int b = threadIdx.x; b < nb; b += 64) {
#pragma unroll
        for (int a = 0; a < 3; a++) s[a] += partial[3 * b + a];
    }
#pragma unroll
    for (int a = 0; a < 3; a++) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s[a] += __shfl_down(s[a], o, 64);
    }
    if (threadIdx.x == 0) {
        out[0] = s[0];
        out[1] = s[1];
        out[2] = s[2];
    }
}

struct Affine {
    double t[3], c[3], s;
    int has_t, has_s;
};

__global__ __launch_bounds__(256) void k_affine(double *__restrict__ xyz, int64_t n3, Affine af) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n3) return;
    int a = (int)(i % 3);
    double v = xyz[i];
    if (af.has_t) v += af.t[a];
    if (af.has_s) v = (v - af.c[a]) * af.s + af.c[a];
    xyz[i] = v;
}

// voxel grid: index = floor((p - o) / s), true fp64 divide (Open3D VoxelDownSample order)
// kNN grid:   index = clamp(floor((p - o) * s)), bucketing only
template <typename T, bool VOXEL>
__global__ __launch_bounds__(256) void k_cell_keys(const T *__restrict__ xyz, int64_t n, GridParams gp,
                                                    uint64_t *__restrict__ keys, uint32_t *__restrict__ vals,
                                                    int32_t *__restrict__ ijk_out) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int32_t c[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        double v = (double)xyz[3 * i + a];
        if (VOXEL) {
            c[a] = (int32_t)floor((v - gp.o[a]) / gp.s);
        } else {
            double t = floor((v - gp.o[a]) * gp.s);
            c[a] = !(t > 0.0) ? 0 : (t >= (double)gp.dim[a] ? gp.dim[a] - 1 : (int32_t)t);
        }
    }
    if (keys) {
        keys[i] = ((uint64_t)c[2] * (uint64_t)gp.dim[1] + (uint64_t)c[1]) * (uint64_t)gp.dim[0] + (uint64_t)c[0];
        vals[i] = (uint32_t)i;
    }
    if (ijk_out) {
        ijk_out[3 * i] = c[0];
        ijk_out[3 * i + 1] = c[1];
        ijk_out[3 * i + 2] = c[2];
    }
}

// flags[i] = 1 where point i lies in a voxel layer iz with lo <= iz < hi (the voxel index expression of k_cell_keys)
template <typename T>
__global__ __launch_bounds__(256) void k_layer_flags(const T *__restrict__ xyz, int64_t n, GridParams gp, int64_t lo, int64_t hi,
                                                      uint8_t *__restrict__ flags) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t iz = (int64_t)floor(((double)xyz[3 * i + 2] - gp.o[2]) / gp.s);
    flags[i] = iz >= lo && iz < hi ? 1 : 0;
}

// voxel keys of the selected points only: keys[j] of point sel[j], vals[j] = sel[j]
template <typename T>
__global__ __launch_bounds__(256) void k_cell_keys_of(const T *__restrict__ xyz, const int32_t *__restrict__ sel, int64_t m,
                                                       GridParams gp, uint64_t *__restrict__ keys, uint32_t *__restrict__ vals) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    const int64_t i = sel[j];
    uint64_t c[3];
#pragma unroll
    for (int a = 0; a < 3; a++) c[a] = (uint64_t)(int32_t)floor(((double)xyz[3 * i + a] - gp.o[a]) / gp.s);
    keys[j] = (c[2] * (uint64_t)gp.dim[1] + c[1]) * (uint64_t)gp.dim[0] + c[0];
    vals[j] = (uint32_t)i;
}

// kNN grid keys as u32 (the dense table has fewer than 2^32 cells)
// (halfbit: one more key bit below the cell, set for points in the upper x-half of their cell -- the
// expression x_half_of() in pc_common.h, which the tile kernel repeats)
template <typename T>
__global__ __launch_bounds__(256) void k_cell_keys32(const T *__restrict__ xyz, int64_t n, GridParams gp,
                                                      uint32_t *__restrict__ keys, uint32_t *__restrict__ vals,
                                                      int halfbit = 0) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t c[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const double t = floor(((double)xyz[3 * i + a] - gp.o[a]) * gp.s);
        c[a] = !(t > 0.0) ? 0u : (t >= (double)gp.dim[a] ? (uint32_t)(gp.dim[a] - 1) : (uint32_t)t);
    }
    uint32_t key = (c[2] * (uint32_t)gp.dim[1] + c[1]) * (uint32_t)gp.dim[0] + c[0];
    if (halfbit) key = (key << 1) | x_half_of((double)xyz[3 * i], gp, (int)c[0]);
    keys[i] = key;
    vals[i] = (uint32_t)i;
}

// counts[cell] += 1 for every key, sorted or not (halfbit: the key's lowest bit is not part of the cell)
__global__ __launch_bounds__(256) void k_cell_count(const uint32_t *__restrict__ keys, int64_t n, uint32_t *__restrict__ counts, int halfbit) {
    // One atomic per run of equal cells among a wave's neighbouring lanes: the points come in some spatial order
    // (voxel centroids by (iz, iy, ix)), and a coarse trial edge sends hundreds of neighbours to the same counter
    // (a LiDAR frame's three trials: 22 -> 3.5 us each on average).
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & (PC_WAVE - 1);
    const uint32_t c = i < n ? keys[i] >> halfbit : 0xFFFFFFFFu;  // (no cell has that number: the lanes behind the end form runs of their own)
    const uint32_t before = __shfl_up(c, 1);
    const bool head = lane == 0 || before != c;
    const unsigned long long heads = __ballot(head);
    if (head && i < n) {
        const unsigned long long later = lane == PC_WAVE - 1 ? 0ull : heads >> (lane + 1);
        atomicAdd(counts + c, later ? (uint32_t)__ffsll((long long)later) : (uint32_t)(PC_WAVE - lane));
    }
}

// Per-cell counts from the SORTED keys, no searching: the last element of a run adds its end
// position, the first subtracts its start position (u32 wrap-around cancels); two contention-free
// atomics per occupied cell.  cnt must be zero on entry.
__global__ __launch_bounds__(256) void k_cell_runs(const uint32_t *__restrict__ keys, int64_t n,
                                                    uint32_t *__restrict__ cnt, int shift = 0) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t key = keys[i] >> shift;  // (the cell, without the x-half bit)
    const bool head = i == 0 || (keys[i - 1] >> shift) != key;
    const bool tail = i + 1 == n || (keys[i + 1] >> shift) != key;
    if (head && tail)
        cnt[key] = 1u;
    else if (tail)
        atomicAdd(&cnt[key], (uint32_t)(i + 1));
    else if (head)
        atomicSub(&cnt[key], (uint32_t)i);
}

// The same for (y, z) rows of cells: cnt[row] = points of the row (row = key / nx).
__global__ __launch_bounds__(256) void k_row_runs(const uint32_t *__restrict__ keys, int64_t n, uint32_t nx,
                                                   uint32_t *__restrict__ cnt) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t row = keys[i] / nx;
    const bool head = i == 0 || keys[i - 1] / nx != row;
    const bool tail = i + 1 == n || keys[i + 1] / nx != row;
    if (head && tail)
        cnt[row] = 1u;
    else if (tail)
        atomicAdd(&cnt[row], (uint32_t)(i + 1));
    else if (head)
        atomicSub(&cnt[row], (uint32_t)i);
}

// Occupancy statistics from the per-cell counts: stats[0] += sum of count^2 (n times the occupancy
// a random point sees), stats[1] += number of occupied cells.
__global__ __launch_bounds__(256) void k_count_stats(const uint32_t *__restrict__ cnt, int64_t ncell,
                                                      unsigned long long *__restrict__ stats) {
    unsigned long long sq = 0, runs = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < ncell; i += (int64_t)gridDim.x * 256) {
        const unsigned long long c = cnt[i];
        sq += c * c;
        runs += c ? 1 : 0;
    }
    __shared__ unsigned long long red[2][4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        sq += __shfl_down(sq, o, 64);
        runs += __shfl_down(runs, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = sq;
        red[1][threadIdx.x >> 6] = runs;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&stats[0], red[0][0] + red[0][1] + red[0][2] + red[0][3]);
        atomicAdd(&stats[1], red[1][0] + red[1][1] + red[1][2] + red[1][3]);
    }
}

// The same statistics straight from the sorted keys (grids too empty for a per-cell table): the
// first element of every run finds the run's end (a short linear probe, then a binary search).
__global__ __launch_bounds__(256) void k_run_stats(const uint32_t *__restrict__ keys, int64_t n,
                                                    unsigned long long *__restrict__ stats) {
    unsigned long long sq = 0, runs = 0;
    // grid-stride: a few thousand atomics in total instead of one pair per wave
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const uint32_t key = keys[i];
        if (i == 0 || keys[i - 1] != key) {
            int64_t e = i + 1;
            int probe = 0;
            while (e < n && probe < 16 && keys[e] == key) {
                e++;
                probe++;
            }
            if (e < n && keys[e] == key) {
                int64_t lo = e, hi = n;
                while (lo < hi) {
                    const int64_t mid = (lo + hi) >> 1;
                    if (keys[mid] <= key)
                        lo = mid + 1;
                    else
                        hi = mid;
                }
                e = lo;
            }
            sq += (unsigned long long)(e - i) * (unsigned long long)(e - i);
            runs += 1;
        }
    }
    __shared__ unsigned long long red[2][4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        sq += __shfl_down(sq, o, 64);
        runs += __shfl_down(runs, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = sq;
        red[1][threadIdx.x >> 6] = runs;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&stats[0], red[0][0] + red[0][1] + red[0][2] + red[0][3]);
        atomicAdd(&stats[1], red[1][0] + red[1][1] + red[1][2] + red[1][3]);
    }
}

template <typename T, typename REC>
__global__ __launch_bounds__(256) void k_gather_records(const T *__restrict__ xyz, const uint32_t *__restrict__ vals,
                                                         int64_t n, REC *__restrict__ rec,
                                                         int32_t *__restrict__ perm) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t j = vals[i];
    REC r;
    r.x = xyz[3 * (size_t)j];
    r.y = xyz[3 * (size_t)j + 1];
    r.z = xyz[3 * (size_t)j + 2];
    r.idx = (decltype(r.idx))j;
    rec[i] = r;
    if (perm) perm[i] = (int32_t)j;
}

// The same two steps from another level's records instead of the caller's array: a level of twice (or half)
// the cell edge visits that level's records nearly in order, where the caller's array is visited at random.
template <typename REC>
__global__ __launch_bounds__(256) void k_cell_keys32_rec(const REC *__restrict__ src, int64_t n, GridParams gp,
                                                          uint32_t *__restrict__ keys, uint32_t *__restrict__ vals) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const REC r = src[i];
    const double p[3] = {(double)r.x, (double)r.y, (double)r.z};
    uint32_t c[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const double t = floor((p[a] - gp.o[a]) * gp.s);
        c[a] = !(t > 0.0) ? 0u : (t >= (double)gp.dim[a] ? (uint32_t)(gp.dim[a] - 1) : (uint32_t)t);
    }
    keys[i] = (c[2] * (uint32_t)gp.dim[1] + c[1]) * (uint32_t)gp.dim[0] + c[0];
    vals[i] = (uint32_t)i;
}

template <typename REC>
__global__ __launch_bounds__(256) void k_regather_records(const REC *__restrict__ src, const uint32_t *__restrict__ vals,
                                                           int64_t n, REC *__restrict__ rec, int32_t *__restrict__ perm) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const REC r = src[vals[i]];
    rec[i] = r;
    if (perm) perm[i] = (int32_t)r.idx;
}

__global__ __launch_bounds__(256) void k_segment_heads(const uint64_t *__restrict__ keys, int64_t n,
                                                        uint8_t *__restrict__ flags) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) flags[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;
}

// one thread per voxel, sequential sums in point-index order (the sort is stable)
template <typename T>
__global__ __launch_bounds__(256) void k_voxel_reduce(const T *__restrict__ xyz, const double *__restrict__ nrm,
                                                       const double *__restrict__ col,
                                                       const uint64_t *__restrict__ keys,
                                                       const uint32_t *__restrict__ vals,
                                                       const int32_t *__restrict__ heads, int64_t m, int64_t n,
                                                       GridParams gp, double *__restrict__ out_xyz,
                                                       double *__restrict__ out_nrm, double *__restrict__ out_col,
                                                       int32_t *__restrict__ out_cnt, int32_t *__restrict__ out_ijk) {
    int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= m) return;
    int64_t s = heads[v], e = v + 1 < m ? heads[v + 1] : n;
    double sp[3] = {0, 0, 0}, sn[3] = {0, 0, 0}, sc[3] = {0, 0, 0};
    for (int64_t t = s; t < e; t++) {
        size_t i = vals[t];
#pragma unroll
        for (int a = 0; a < 3; a++) sp[a] += (double)xyz[3 * i + a];
        if (nrm) {
            double n0 = nrm[3 * i], n1 = nrm[3 * i + 1], n2 = nrm[3 * i + 2];
            if (!isnan(n0) && !isnan(n1) && !isnan(n2)) {
                sn[0] += n0;
                sn[1] += n1;
                sn[2] += n2;
            }
        }
        if (col) {
#pragma unroll
            for (int a = 0; a < 3; a++) sc[a] += col[3 * i + a];
        }
    }
    double cnt = (double)(e - s);
#pragma unroll
    for (int a = 0; a < 3; a++) {
        out_xyz[3 * v + a] = sp[a] / cnt;
        if (nrm && out_nrm) out_nrm[3 * v + a] = sn[a] / cnt;
        if (col && out_col) out_col[3 * v + a] = sc[a] / cnt;
    }
    if (out_cnt) out_cnt[v] = (int32_t)(e - s);
    if (out_ijk) {
        uint64_t key = keys[s];
        out_ijk[3 * v] = (int32_t)(key % (uint64_t)gp.dim[0]);
        out_ijk[3 * v + 1] = (int32_t)((key / (uint64_t)gp.dim[0]) % (uint64_t)gp.dim[1]);
        out_ijk[3 * v + 2] = (int32_t)(key / ((uint64_t)gp.dim[0] * (uint64_t)gp.dim[1]));
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_gather_rows(const T *__restrict__ src, const int32_t *__restrict__ idx,
                                                      int64_t m3, T *__restrict__ dst) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= m3) return;
    dst[i] = src[3 * (size_t)idx[i / 3] + (i % 3)];
}

__global__ __launch_bounds__(256) void k_scatter_rows(const double *__restrict__ src, const int32_t *__restrict__ idx,
                                                       int64_t m3, double *__restrict__ dst) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= m3) return;
    dst[3 * (size_t)idx[i / 3] + (i % 3)] = src[i];
}

inline unsigned blocks_for(int64_t n, int per) { return (unsigned)((n + per - 1) / per); }

int bits_for(uint64_t v) {
    int b = 0;
    while (v) {
        b++;
        v >>= 1;
    }
    return b > 0 ? b : 1;
}

int bbox_impl(pc_ctx *ctx, const void *xyz, int dtype, int64_t n, double mn[3], double mx[3], hipStream_t st) {
    if (n <= 0) {
        for (int a = 0; a < 3; a++) mn[a] = mx[a] = 0.0;
        return PC_OK;
    }
    int nb = (int)std::min<int64_t>(1024, (n + 255) / 256);
    Scratch part(ctx, sizeof(double) * 6 * (size_t)(nb + 1));
    if (!part.ok()) return pc_fail(ctx, PC_NOMEM, "bbox scratch");
    double *p = part.as<double>();
    if (dtype == PC_F32)
        hipLaunchKernelGGL(k_bbox<float>, dim3(nb), dim3(256), 0, st, (const float *)xyz, n, p);
    else
        hipLaunchKernelGGL(k_bbox<double>, dim3(nb), dim3(256), 0, st, (const double *)xyz, n, p);
    hipLaunchKernelGGL(k_bbox_final, dim3(1), dim3(64), 0, st, p, nb, p + (size_t)nb * 6);
    PC_HIP(hipGetLastError());
    double *h = reinterpret_cast<double *>(ctx->pinned);
    PC_HIP(hipMemcpyAsync(h, p + (size_t)nb * 6, 6 * sizeof(double), hipMemcpyDeviceToHost, st));
    PC_HIP(hipStreamSynchronize(st));
    for (int a = 0; a < 3; a++) {
        mn[a] = h[a];
        mx[a] = h[3 + a];
    }
    return PC_OK;
}

// voxel grid parameters exactly as Open3D derives them
int voxel_params(pc_ctx *ctx, const double mn[3], const double mx[3], double vs, GridParams *gp) {
    double ext = -INFINITY;
    for (int a = 0; a < 3; a++) {
        gp->o[a] = mn[a] - vs * 0.5;
        double vmax = mx[a] + vs * 0.5;
        ext = std::max(ext, vmax - gp->o[a]);
    }
    if (vs * (double)INT32_MAX < ext) return pc_fail(ctx, PC_OVERFLOW, "voxel_size is too small.");
    gp->s = vs;
    gp->cell = vs;
    for (int a = 0; a < 3; a++) gp->dim[a] = (int32_t)floor((mx[a] - gp->o[a]) / vs) + 1;
    // linear key must fit 63 bits
    long double tot = (long double)gp->dim[0] * (long double)gp->dim[1] * (long double)gp->dim[2];
    if (tot >= 9.0e18L) return pc_fail(ctx, PC_OVERFLOW, "voxel grid has more than 2^63 cells");
    return PC_OK;
}

}  // namespace

extern "C" {

int pc_bbox(pc_ctx *ctx, const void *xyz_dev, int dtype, int64_t n, double out_min[3], double out_max[3],
            void *stream) {
    PcLock lock(ctx, stream);  // calls on one context take turns (its block pool and read-back buffer are shared)
    if (!ctx || !out_min || !out_max || n < 0 || (n > 0 && !xyz_dev) || (dtype != PC_F32 && dtype != PC_F64))
        return pc_fail(ctx, PC_BAD_ARG, "pc_bbox");
    PC_HIP(hipSetDevice(ctx->device));
    return bbox_impl(ctx, xyz_dev, dtype, n, out_min, out_max, (hipStream_t)stream);
}

int pc_host_to_f32_exact(const double *src, int64_t count, float *dst) {
    // host helper of the upload path: dst = (float)src, returns 1 when every value survived the narrowing
    // unchanged (the cloud can then live in HBM as fp32 without losing anything), 0 otherwise
    if (count <= 0) return 1;
    if (!src || !dst) return 0;
    const int nt = (int)std::min<int64_t>(8, std::max<int64_t>(1, count / (1 << 18)));
    std::vector<int> exact((size_t)nt, 1);
    auto work = [&](int t) {
        const int64_t b = count * t / nt, e = count * (t + 1) / nt;
        int ok = 1;
        for (int64_t i = b; i < e; i++) {
            const float f = (float)src[i];
            dst[i] = f;
            ok &= ((double)f == src[i]) ? 1 : 0;  // (NaN compares unequal: such clouds stay fp64)
        }
        exact[(size_t)t] = ok;
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nt; t++) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
    int ok = 1;
    for (int v : exact) ok &= v;
    return ok;
}

int pc_center(pc_ctx *ctx, const void *xyz_dev, int dtype, int64_t n, double out_center[3], void *stream) {
    PcLock lock(ctx, stream);
    if (!ctx || !out_center || n < 0 || (n > 0 && !xyz_dev) || (dtype != PC_F32 && dtype != PC_F64))
        return pc_fail(ctx, PC_BAD_ARG, "pc_center");
    PC_HIP(hipSetDevice(ctx->device));
    out_center[0] = out_center[1] = out_center[2] = 0.0;
    if (n == 0) return PC_OK;
    hipStream_t st = (hipStream_t)stream;
    const int nb = 512;
    Scratch part(ctx, sizeof(double) * 3 * (size_t)(nb + 1));
    if (!part.ok()) return pc_fail(ctx, PC_NOMEM, "centre scratch");
    double *p = part.as<double>();
    if (dtype == PC_F32)
        hipLaunchKernelGGL(k_coord_sums<float>, dim3(nb), dim3(256), 0, st, (const float *)xyz_dev, n, p);
    else
        hipLaunchKernelGGL(k_coord_sums<double>, dim3(nb), dim3(256), 0, st, (const double *)xyz_dev, n, p);
    hipLaunchKernelGGL(k_coord_sums_final, dim3(1), dim3(64), 0, st, p, nb, p + 3 * (size_t)nb);
    PC_HIP(hipGetLastError());
    double *h = reinterpret_cast<double *>(ctx->pinned);
    PC_HIP(hipMemcpyAsync(h, p + 3 * (size_t)nb, 3 * sizeof(double), hipMemcpyDeviceToHost, st));
    PC_HIP(hipStreamSynchronize(st));
    for (int a = 0; a < 3; a++) out_center[a] = h[a] / (double)n;
    return PC_OK;
}

int pc_affine(pc_ctx *ctx, double *xyz_dev, int64_t n, const double *t, int has_scale, double s, const double *c,
              void *stream) {
    PcLock lock(ctx, stream);  // calls on one context take turns (its block pool and read-back buffer are shared)
    if (!ctx || n < 0 || (n > 0 && !xyz_dev) || (has_scale && !c)) return pc_fail(ctx, PC_BAD_ARG, "pc_affine");
    PC_HIP(hipSetDevice(ctx->device));
    if (n == 0) return PC_OK;
    Affine af{};
    af.has_t = t != nullptr;
    af.has_s = has_scale != 0;
    af.s = s;
    for (int a = 0; a < 3; a++) {
        af.t[a] = t ? t[a] : 0.0;
        af.c[a] = c ? c[a] : 0.0;
    }
    hipLaunchKernelGGL(k_affine, dim3(blocks_for(3 * n, 256)), dim3(256), 0, (hipStream_t)stream, xyz_dev, 3 * n, af);
    PC_HIP(hipGetLastError());
    return PC_OK;
}

int pc_voxel_assign(pc_ctx *ctx, const void *xyz_dev, int dtype, int64_t n, double voxel_size, int32_t *out_ijk_dev,
                    int32_t out_dims[3], void *stream) {
    PcLock lock(ctx, stream);  // calls on one context take turns (its block pool and read-back buffer are shared)
    if (!ctx || n < 0 || (n > 0 && (!xyz_dev || !out_ijk_dev)) || (dtype != PC_F32 && dtype != PC_F64))
        return pc_fail(ctx, PC_BAD_ARG, "pc_voxel_assign");
    if (!(voxel_size > 0.0)) return pc_fail(ctx, PC_BAD_ARG, "voxel_size <= 0.");
    PC_HIP(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) {
        if (out_dims) out_dims[0] = out_dims[1] = out_dims[2] = 0;
        return PC_OK;
    }
    double mn[3], mx[3];
    PC_TRY(bbox_impl(ctx, xyz_dev, dtype, n, mn, mx, st));
    for (int a = 0; a < 3; a++)
        if (!std::isfinite(mn[a]) || !std::isfinite(mx[a]))
            return pc_fail(ctx, PC_BAD_ARG, "non-finite coordinates: the cloud has no finite bounding box");
    GridParams gp{};
    PC_TRY(voxel_params(ctx, mn, mx, voxel_size, &gp));
    if (out_dims)
        for (int a = 0; a < 3; a++) out_dims[a] = gp.dim[a];
    if (dtype == PC_F32)
        hipLaunchKernelGGL((k_cell_keys<float, true>), dim3(blocks_for(n, 256)), dim3(256), 0, st,
                           (const float *)xyz_dev, n, gp, (uint64_t *)nullptr, (uint32_t *)nullptr, out_ijk_dev);
    else
        hipLaunchKernelGGL((k_cell_keys<double, true>), dim3(blocks_for(n, 256)), dim3(256), 0, st,
                           (const double *)xyz_dev, n, gp, (uint64_t *)nullptr, (uint32_t *)nullptr, out_ijk_dev);
    PC_HIP(hipGetLastError());
    PC_HIP(hipStreamSynchronize(st));
    return PC_OK;
}

static int voxel_downsample_impl(pc_ctx *ctx, const void *xyz_dev, int dtype, const double *normals_dev,
                                 const double *colors_dev, int64_t n, double voxel_size, const double *box_min,
                                 const double *box_max, int64_t layer_lo, int64_t layer_hi, double *out_xyz_dev,
                                 double *out_normals_dev, double *out_colors_dev, int32_t *out_count_dev,
                                 int32_t *out_ijk_dev, int64_t *out_m, void *stream) {
    if (!(voxel_size > 0.0)) return pc_fail(ctx, PC_BAD_ARG, "voxel_size <= 0.");
    PC_HIP(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    *out_m = 0;
    if (n == 0) return PC_OK;
    double mn[3], mx[3];
    if (box_min && box_max) {
        for (int a = 0; a < 3; a++) {
            mn[a] = box_min[a];
            mx[a] = box_max[a];
        }
    } else {
        PC_TRY(bbox_impl(ctx, xyz_dev, dtype, n, mn, mx, st));
    }
    for (int a = 0; a < 3; a++)
        if (!std::isfinite(mn[a]) || !std::isfinite(mx[a]) || mx[a] < mn[a])
            return pc_fail(ctx, PC_BAD_ARG, "non-finite coordinates: the cloud has no finite bounding box");
    GridParams gp{};
    PC_TRY(voxel_params(ctx, mn, mx, voxel_size, &gp));

    // a range of voxel layers: only the points inside it are keyed and sorted
    const bool ranged = box_min && box_max;
    int64_t m_in = n;
    Scratch sel(ctx, ranged ? sizeof(int32_t) * (size_t)n : 0);
    if (ranged) {
        Scratch lf(ctx, (size_t)n);
        if (!sel.ok() || !lf.ok()) return pc_fail(ctx, PC_NOMEM, "voxel scratch");
        if (dtype == PC_F32)
            hipLaunchKernelGGL(k_layer_flags<float>, dim3(blocks_for(n, 256)), dim3(256), 0, st, (const float *)xyz_dev, n, gp,
                               layer_lo, layer_hi, lf.as<uint8_t>());
        else
            hipLaunchKernelGGL(k_layer_flags<double>, dim3(blocks_for(n, 256)), dim3(256), 0, st, (const double *)xyz_dev, n, gp,
                               layer_lo, layer_hi, lf.as<uint8_t>());
        PC_TRY(pc_compact_flags(ctx, lf.as<uint8_t>(), n, sel.as<int32_t>(), &m_in, st));  // (synchronises: lf goes back)
        if (m_in == 0) return PC_OK;
    }
    Scratch keys(ctx, sizeof(uint64_t) * (size_t)m_in), keys2(ctx, sizeof(uint64_t) * (size_t)m_in);
    Scratch vals(ctx, sizeof(uint32_t) * (size_t)m_in), vals2(ctx, sizeof(uint32_t) * (size_t)m_in);
    Scratch flags(ctx, (size_t)m_in), heads(ctx, sizeof(int32_t) * (size_t)m_in);
    if (!keys.ok() || !keys2.ok() || !vals.ok() || !vals2.ok() || !flags.ok() || !heads.ok())
        return pc_fail(ctx, PC_NOMEM, "voxel scratch");
    if (ranged && dtype == PC_F32)
        hipLaunchKernelGGL(k_cell_keys_of<float>, dim3(blocks_for(m_in, 256)), dim3(256), 0, st, (const float *)xyz_dev,
                           sel.as<int32_t>(), m_in, gp, keys.as<uint64_t>(), vals.as<uint32_t>());
    else if (ranged)
        hipLaunchKernelGGL(k_cell_keys_of<double>, dim3(blocks_for(m_in, 256)), dim3(256), 0, st, (const double *)xyz_dev,
                           sel.as<int32_t>(), m_in, gp, keys.as<uint64_t>(), vals.as<uint32_t>());
    else if (dtype == PC_F32)
        hipLaunchKernelGGL((k_cell_keys<float, true>), dim3(blocks_for(n, 256)), dim3(256), 0, st,
                           (const float *)xyz_dev, n, gp, keys.as<uint64_t>(), vals.as<uint32_t>(), (int32_t *)nullptr);
    else
        hipLaunchKernelGGL((k_cell_keys<double, true>), dim3(blocks_for(n, 256)), dim3(256), 0, st,
                           (const double *)xyz_dev, n, gp, keys.as<uint64_t>(), vals.as<uint32_t>(), (int32_t *)nullptr);
    uint64_t ncell = (uint64_t)gp.dim[0] * (uint64_t)gp.dim[1] * (uint64_t)gp.dim[2];
    bool in_tmp = false;
    PC_TRY(pc_radix_sort_pairs(ctx, keys.as<uint64_t>(), vals.as<uint32_t>(), keys2.as<uint64_t>(),
                               vals2.as<uint32_t>(), m_in, bits_for(ncell - 1), st, &in_tmp));
    const uint64_t *sk = in_tmp ? keys2.as<uint64_t>() : keys.as<uint64_t>();
    const uint32_t *sv = in_tmp ? vals2.as<uint32_t>() : vals.as<uint32_t>();
    hipLaunchKernelGGL(k_segment_heads, dim3(blocks_for(m_in, 256)), dim3(256), 0, st, sk, m_in, flags.as<uint8_t>());
    int64_t m = 0;
    PC_TRY(pc_compact_flags(ctx, flags.as<uint8_t>(), m_in, heads.as<int32_t>(), &m, st));
    if (dtype == PC_F32)
        hipLaunchKernelGGL(k_voxel_reduce<float>, dim3(blocks_for(m, 256)), dim3(256), 0, st, (const float *)xyz_dev,
                           normals_dev, colors_dev, sk, sv, heads.as<int32_t>(), m, m_in, gp, out_xyz_dev,
                           out_normals_dev, out_colors_dev, out_count_dev, out_ijk_dev);
    else
        hipLaunchKernelGGL(k_voxel_reduce<double>, dim3(blocks_for(m, 256)), dim3(256), 0, st,
                           (const double *)xyz_dev, normals_dev, colors_dev, sk, sv, heads.as<int32_t>(), m, m_in, gp,
                           out_xyz_dev, out_normals_dev, out_colors_dev, out_count_dev, out_ijk_dev);
    PC_HIP(hipGetLastError());
    PC_HIP(hipStreamSynchronize(st));  // scratch is released on return
    *out_m = m;
    return PC_OK;
}

int pc_voxel_downsample(pc_ctx *ctx, const void *xyz_dev, int dtype, const double *normals_dev,
                        const double *colors_dev, int64_t n, double voxel_size, double *out_xyz_dev,
                        double *out_normals_dev, double *out_colors_dev, int32_t *out_count_dev,
                        int32_t *out_ijk_dev, int64_t *out_m, void *stream) {
    PcLock lock(ctx, stream);  // calls on one context take turns (its block pool and read-back buffer are shared)
    if (!ctx || !out_m || n < 0 || n >= INT32_MAX || (n > 0 && (!xyz_dev || !out_xyz_dev)) ||
        (dtype != PC_F32 && dtype != PC_F64))
        return pc_fail(ctx, PC_BAD_ARG, "pc_voxel_downsample");
    return voxel_downsample_impl(ctx, xyz_dev, dtype, normals_dev, colors_dev, n, voxel_size, nullptr, nullptr, 0, 0,
                                 out_xyz_dev, out_normals_dev, out_colors_dev, out_count_dev, out_ijk_dev, out_m, stream);
}

int pc_voxel_downsample_layers(pc_ctx *ctx, const void *xyz_dev, int dtype, const double *normals_dev,
                               const double *colors_dev, int64_t n, double voxel_size, const double box_min[3],
                               const double box_max[3], int64_t layer_lo, int64_t layer_hi, double *out_xyz_dev,
                               double *out_normals_dev, double *out_colors_dev, int32_t *out_count_dev,
                               int32_t *out_ijk_dev, int64_t *out_m, void *stream) {
    PcLock lock(ctx, stream);
    if (!ctx || !out_m || !box_min || !box_max || n < 0 || n >= INT32_MAX || (n > 0 && (!xyz_dev || !out_xyz_dev)) ||
        (dtype != PC_F32 && dtype != PC_F64))
        return pc_fail(ctx, PC_BAD_ARG, "pc_voxel_downsample_layers");
    return voxel_downsample_impl(ctx, xyz_dev, dtype, normals_dev, colors_dev, n, voxel_size, box_min, box_max, layer_lo,
                                 layer_hi, out_xyz_dev, out_normals_dev, out_colors_dev, out_count_dev, out_ijk_dev, out_m,
                                 stream);
}

// The index of a cloud given as an array (parent == NULL), or of the cloud another level of the same index
// holds (parent: its records are the source, xyz_dev is not read; ties inside a cell then follow the parent's
// order instead of the original index, which no search result depends on).
static int grid_build_impl(pc_ctx *ctx, const void *xyz_dev, int dtype, int64_t n, double cell_size, int k_hint,
                           pc_grid **out, void *stream, const pc_grid *parent) {
    hipStream_t st = (hipStream_t)stream;
    *out = nullptr;
    pc_grid *g = new (std::nothrow) pc_grid();
    if (!g) return pc_fail(ctx, PC_NOMEM, "grid");
    g->n = n;
    g->dtype = dtype;
    if (parent) g->index_space = parent->index_space;
    g->k_hint = parent ? parent->k_hint : k_hint;
    double mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
    int rc = PC_OK;
    if (parent) {
        for (int a = 0; a < 3; a++) {
            mn[a] = parent->bb_min[a];
            mx[a] = parent->bb_max[a];
        }
    } else {
        rc = bbox_impl(ctx, xyz_dev, dtype, n, mn, mx, st);
    }
    if (rc) {
        delete g;
        return rc;
    }
    // NaN coordinates never reach the box (fmin / fmax skip them: an all-NaN axis leaves +inf / -inf) and
    // infinite ones make it infinite: neither can be put into cells
    for (int a = 0; a < 3 && n > 0; a++)
        if (!std::isfinite(mn[a]) || !std::isfinite(mx[a]) || mx[a] < mn[a]) {
            delete g;
            return pc_fail(ctx, PC_BAD_ARG, "non-finite coordinates: the cloud has no finite bounding box");
        }
    for (int a = 0; a < 3; a++) {
        g->bb_min[a] = mn[a];
        g->bb_max[a] = mx[a];
    }
    double ext[3], emax = 0;
    for (int a = 0; a < 3; a++) {
        ext[a] = mx[a] - mn[a];
        emax = std::max(emax, ext[a]);
        g->gp.o[a] = mn[a];
    }
    const bool automatic = !(cell_size > 0);
    double cell = cell_size, target = 1.0;
    if (automatic) {
        // cell edge ~ the 99.5 % quantile of the k-th neighbour distance in a uniform cloud
        // (the ball then holds k + 2.6 sqrt(k) + 1 points): one ring of cells around a query's
        // cell certifies almost every query
        const double kk = (double)std::max(1, k_hint);
        // (x 1.16 in volume = x 1.05 in edge: measured optimum of the tile kernel at k = 2 ... 32)
        target = std::max(1.0, 1.16 * 0.2387 * (kk + 2.6 * sqrt(kk) + 1.0));
        double vol = 1.0;
        for (int a = 0; a < 3; a++) vol *= std::max(ext[a], emax * 1e-3);
        cell = (n > 0 && vol > 0) ? cbrt(vol * target / (double)n) : 1.0;
        if (!(cell > 0) || !std::isfinite(cell)) cell = 1.0;
        {   // tuning knob: scales the automatic edge
            const double f = ctx->knobs.cell_scale;
            if (f > 0.1 && f < 10.0) cell *= f;
        }
    }
    // u32 cell keys: below 2^32 cells.  A dense offset table is kept while it stays within
    // 16 n + 4096 cells; emptier grids use the sparse index (binary search over the sorted keys).
    auto fit_dims = [&](double c, int32_t dim[3]) {
        for (int it = 0;; it++) {
            double tot = 1.0;
            for (int a = 0; a < 3; a++) {
                double d = std::min(floor(ext[a] / c) + 1.0, 2.0e9);
                if (!(d >= 1.0)) d = 1.0;  // (never with a finite box; keeps the cast defined)
                dim[a] = (int32_t)d;
                tot *= d;
            }
            if (tot <= 4.0e9 || it >= 4096) return c;  // (1.1^4096 overflows any finite extent long before)
            c *= 1.1;
        }
    };
    cell = fit_dims(cell, g->gp.dim);
    if (automatic && !parent && n >= 65536 && !(ctx->knobs.cell_scale > 0)) {
        // Among the edges within 2.5 % of that one, take the one whose tiles waste the fewest lanes: the tile kernel of
        // a cloud that fills its box runs one workgroup per tile, a tile at the far end of an axis is cut off where
        // the grid ends, and a slightly different occupancy may admit a fuller tile shape.  Score = queries per lane
        // of all workgroups launched.  (10 M uniform points, k = 16: 110^3 cells in tiles of 5x3x2 score 0.873,
        // 112^3 cells in tiles of 4x4x2 0.890 -- kNN launch 1.84 -> 1.81 ms; 111^3 and 114^3 score 0.82 and took
        // 1.87 and 1.93 ms.  Uneven clouds settle their cells further down; nothing here depends on the choice but speed.)
        double best_score = -1.0, best_c = cell;
        for (int step = 0; step <= 20; step++) {
            const double f = 1.0 + 0.0025 * (double)((step + 1) / 2) * ((step & 1) ? -1.0 : 1.0);  // 1, 0.9975, 1.0025, ...
            int32_t dim[3];
            const double c = fit_dims(cell * f, dim);
            // (points per cell where the cloud is: the last layer of cells of an axis is only partly inside the box)
            double box = 1.0;
            for (int a = 0; a < 3; a++) box *= std::max(ext[a], c);
            int tile[3], waves = 1;
            double pop = 0.0;
            pc_dense_tile_shape(ctx, (double)n * c * c * c / box, dim, tile, &waves, &pop);
            double ntiles = 1.0;
            for (int a = 0; a < 3; a++) ntiles *= (double)((dim[a] + tile[a] - 1) / tile[a]);
            // (a tile with more queries than lanes costs one of its waves a second batch: counted as a wave more,
            // weighted with the chance of it under a normal approximation of the tile's population)
            const double over = pop > 0 ? 0.5 * erfc(((double)(PC_WAVE * waves) - pop) / sqrt(2.0 * pop)) : 0.0;
            const double score = (double)n / (ntiles * PC_WAVE * ((double)waves + over));
            if (score > best_score * 1.003) {  // (the nearer edge unless another is clearly better)
                best_score = score;
                best_c = c;
            }
        }
        cell = fit_dims(best_c, g->gp.dim);
    }

    size_t rec_bytes = (dtype == PC_F32 ? sizeof(Rec32) : sizeof(Rec64)) * (size_t)std::max<int64_t>(n, 1);
    size_t perm_bytes = sizeof(int32_t) * (size_t)std::max<int64_t>(n, 1);
    size_t cs_bytes = 0;
    rc = PC_OK;
    do {
        Scratch keys(ctx, sizeof(uint32_t) * (size_t)std::max<int64_t>(n, 1)),
            keys2(ctx, sizeof(uint32_t) * (size_t)std::max<int64_t>(n, 1));
        Scratch vals(ctx, sizeof(uint32_t) * (size_t)std::max<int64_t>(n, 1)),
            vals2(ctx, sizeof(uint32_t) * (size_t)std::max<int64_t>(n, 1));
        Scratch stats(ctx, 512);
        if (!keys.ok() || !keys2.ok() || !vals.ok() || !vals2.ok() || !stats.ok()) {
            rc = pc_fail(ctx, PC_NOMEM, "grid scratch");
            break;
        }
        bool in_tmp = false, dense = true, built_direct = false;
        // (A table of up to 16 cells per point: looking a cell up in it is two loads where the sorted keys of a table-less
        // grid are searched -- LiDAR frame p50 1.71 -> 1.64 ms, ICP over a 2 M-point torus 1.68 -> 1.48 ms per iteration at
        // 16 instead of 8; 64 bytes per point at most.)
        const double dense_factor = ctx->knobs.dense_factor;
        // Key + sort; in automatic mode, shrink the cells while a random point shares its cell with
        // far more points than intended (surfaces and clusters fill few cells of the box densely).
        // Where smaller cells separate nothing -- as many occupied cells as before: the crowd in a cell is copies
        // of one position (merged scans, coordinates quantised at a far origin) -- the previous, coarser edge is
        // taken back: ever smaller cells would only put the next distinct position dozens of empty cells away
        // (90 000 points on 1 100 positions: queries between the positions took seconds each in the ring scan).
        int64_t prev_occupied = -1;
        double prev_cell = cell;
        int32_t prev_dim[3] = {g->gp.dim[0], g->gp.dim[1], g->gp.dim[2]};
        bool settle = false;
        for (int attempt = 0; n > 0; attempt++) {
            g->gp.cell = cell;
            g->gp.s = 1.0 / cell;
            g->ncell = (int64_t)g->gp.dim[0] * g->gp.dim[1] * g->gp.dim[2];
            // evenly filled fp32 clouds (decided later; the bit is harmless otherwise): x-halves inside the cells
            dense = (double)g->ncell <= dense_factor * (double)n + 4096.0;
            // (not for the nearly empty cells of k <= 2: nothing to gain from halving one or two points)
            const int halfbit = (!parent && automatic && dense && dtype == PC_F32 && g->ncell < (1ll << 31) && k_hint > 2 &&
                                 !ctx->knobs.no_halves) ? 1 : 0;
            g->halves = halfbit != 0;
            unsigned long long *hs = reinterpret_cast<unsigned long long *>(ctx->pinned);
            built_direct = false;
            if (!parent && dtype == PC_F32 && dense && !ctx->knobs.no_bucket_build) {
                // an evenly filled fp32 cloud: records, index map and cell table in two moves of the records
                // (pc_build.hip); a cloud that crowds one bucket or one cell takes the general route below
                if (g->cell_start) ctx->release(g->cell_start);
                cs_bytes = sizeof(uint32_t) * (size_t)(g->ncell + 1);
                g->cell_start = (uint32_t *)ctx->alloc(cs_bytes);
                if (!g->rec) g->rec = ctx->alloc(rec_bytes);
                if (!g->perm) g->perm = (int32_t *)ctx->alloc(perm_bytes);
                if (!g->cell_start || !g->rec || !g->perm) {
                    rc = pc_fail(ctx, PC_NOMEM, "grid buffers");
                    break;
                }
                int taken = 0;
                rc = pc_build_bucketed(ctx, (const float *)xyz_dev, n, g->gp, g->ncell, halfbit, (Rec32 *)g->rec, g->perm,
                                       g->cell_start, stats.as<unsigned long long>(), st, &taken);
                if (rc) break;
                if (taken) {
                    if (hipMemcpyAsync(hs, stats.p, 3 * 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
                        hipStreamSynchronize(st) != hipSuccess) {
                        rc = pc_fail(ctx, PC_HIP_ERROR, "grid occupancy statistics");
                        break;
                    }
                    built_direct = hs[2] == 0;
                    if (ctx->knobs.debug)
                        fprintf(stderr, "[pc_amd]   bucketed build: %s\n", built_direct ? "taken" : "a bucket or cell overflowed, general build");
                }
            }
            // A cloud given as an array whose cells are still to be settled is COUNTED into the cell table first (one
            // atomic per point) and sorted only once the cell edge is accepted: an edge that is given up never pays
            // for a sort (a LiDAR frame tries three edges).
            bool probe = false;
            auto sort_keys = [&]() {
                return pc_radix_sort_pairs32(ctx, keys.as<uint32_t>(), vals.as<uint32_t>(), keys2.as<uint32_t>(), vals2.as<uint32_t>(), n,
                                             bits_for((uint64_t)g->ncell - 1) + halfbit, st, &in_tmp);
            };
            if (!built_direct) {
                if (parent && dtype == PC_F32)
                    hipLaunchKernelGGL(k_cell_keys32_rec<Rec32>, dim3(blocks_for(n, 256)), dim3(256), 0, st,
                                       (const Rec32 *)parent->rec, n, g->gp, keys.as<uint32_t>(), vals.as<uint32_t>());
                else if (parent)
                    hipLaunchKernelGGL(k_cell_keys32_rec<Rec64>, dim3(blocks_for(n, 256)), dim3(256), 0, st,
                                       (const Rec64 *)parent->rec, n, g->gp, keys.as<uint32_t>(), vals.as<uint32_t>());
                else if (dtype == PC_F32)
                    hipLaunchKernelGGL(k_cell_keys32<float>, dim3(blocks_for(n, 256)), dim3(256), 0, st,
                                       (const float *)xyz_dev, n, g->gp, keys.as<uint32_t>(), vals.as<uint32_t>(), halfbit);
                else
                    hipLaunchKernelGGL(k_cell_keys32<double>, dim3(blocks_for(n, 256)), dim3(256), 0, st,
                                       (const double *)xyz_dev, n, g->gp, keys.as<uint32_t>(), vals.as<uint32_t>(), 0);
                // per-cell counts (the offset table before its scan) while the table stays within 16 n + 4096
                // cells; emptier grids keep the sorted keys instead and are searched
                dense = (double)g->ncell <= dense_factor * (double)n + 4096.0;
                // (small clouds only: their sorts are launch-bound; 66 M points on a first edge of 270 points per cell took
                // 2 ms longer to count by atomics than to sort)
                probe = automatic && !parent && dense && n < (1ll << 20) && !ctx->knobs.no_probe;
                if (!probe) {
                    rc = sort_keys();
                    if (rc) break;
                }
                const uint32_t *sorted_keys = in_tmp ? keys2.as<uint32_t>() : keys.as<uint32_t>();
                if (g->cell_start) {
                    ctx->release(g->cell_start);
                    g->cell_start = nullptr;
                }
                if (dense) {
                    cs_bytes = sizeof(uint32_t) * (size_t)(g->ncell + 1);
                    g->cell_start = (uint32_t *)ctx->alloc(cs_bytes);
                    if (!g->cell_start) {
                        rc = pc_fail(ctx, PC_NOMEM, "cell table");
                        break;
                    }
                    if (hipMemsetAsync(g->cell_start, 0, cs_bytes, st) != hipSuccess) {
                        rc = pc_fail(ctx, PC_HIP_ERROR, "memset cell table");
                        break;
                    }
                    if (probe)
                        hipLaunchKernelGGL(k_cell_count, dim3(blocks_for(n, 256)), dim3(256), 0, st, keys.as<uint32_t>(), n, g->cell_start,
                                           halfbit);
                    else
                        hipLaunchKernelGGL(k_cell_runs, dim3(blocks_for(n, 256)), dim3(256), 0, st, sorted_keys, n, g->cell_start,
                                           halfbit);
                }
                // (a level made from another level has its cell edge given: nothing here waits for the numbers)
                const bool defer = parent != nullptr && !automatic;
                unsigned long long *sdev = stats.as<unsigned long long>();
                if (defer) {
                    g->stats_dev = (unsigned long long *)ctx->alloc(256);
                    if (!g->stats_dev) {
                        rc = pc_fail(ctx, PC_NOMEM, "grid statistics");
                        break;
                    }
                    sdev = g->stats_dev;
                }
                if (hipMemsetAsync(sdev, 0, 2 * 8, st) != hipSuccess) break;
                if (dense)
                    hipLaunchKernelGGL(k_count_stats, dim3(std::min(blocks_for(g->ncell, 256), 512u)), dim3(256), 0, st,
                                       g->cell_start, g->ncell, sdev);
                else
                    hipLaunchKernelGGL(k_run_stats, dim3(std::min(blocks_for(n, 256), 1024u)), dim3(256), 0, st, sorted_keys,
                                       n, sdev);
                if (defer) {
                    g->stats_pending = true;
                    break;
                }
                if (hipMemcpyAsync(hs, stats.p, 2 * 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
                    hipStreamSynchronize(st) != hipSuccess) {
                    rc = pc_fail(ctx, PC_HIP_ERROR, "grid occupancy statistics");
                    break;
                }
            }  // (general route)
            double seen = (double)hs[0] / (double)n;  // occupancy a random point sees (Poisson: mean + 1)
            g->occupied = (int64_t)hs[1];
            g->seen = seen;
            if (ctx->knobs.debug && automatic)
                fprintf(stderr, "[pc_amd]   grid attempt %d: cell %.4g cells %lld occupied %lld seen %.1f (target %.1f)\n", attempt,
                        cell, (long long)g->ncell, (long long)g->occupied, seen, target);
            bool accept = !automatic || attempt >= 8 || settle || seen <= 2.0 * (target + 1.0);
            if (!accept && attempt > 0 && g->occupied <= prev_occupied + prev_occupied / 50) {
                cell = prev_cell;
                for (int a = 0; a < 3; a++) g->gp.dim[a] = prev_dim[a];
                settle = true;
                continue;
            }
            if (!accept) {
                prev_occupied = g->occupied;
                prev_cell = cell;
                for (int a = 0; a < 3; a++) prev_dim[a] = g->gp.dim[a];
                int32_t dim2[3];
                const double want = cell * std::max(0.3, sqrt(1.15 * (target + 1.0) / seen));
                const double got = fit_dims(want, dim2);
                if (got > 0.93 * cell) {
                    accept = true;  // the key range binds: no finer grid available
                } else {
                    cell = got;
                    for (int a = 0; a < 3; a++) g->gp.dim[a] = dim2[a];
                }
            }
            if (accept) {
                if (probe) rc = sort_keys();  // (the edge stays: now the points are put in order)
                break;
            }
        }
        if (rc) break;
        g->gp.cell = cell;
        g->gp.s = 1.0 / cell;
        g->ncell = (int64_t)g->gp.dim[0] * g->gp.dim[1] * g->gp.dim[2];
        if (n == 0) dense = true;
        if (!dense) cs_bytes = sizeof(uint32_t) * (size_t)std::max<int64_t>(n, 1);
        if (!g->rec) g->rec = ctx->alloc(rec_bytes);
        if (!g->perm) g->perm = (int32_t *)ctx->alloc(perm_bytes);
        if (n == 0) {
            cs_bytes = sizeof(uint32_t) * (size_t)(g->ncell + 1);
            g->cell_start = (uint32_t *)ctx->alloc(cs_bytes);
        } else if (!dense) {
            g->keys = (uint32_t *)ctx->alloc(cs_bytes);
        }
        g->bytes = rec_bytes + cs_bytes + perm_bytes;
        if (!g->rec || !g->perm || (dense ? !g->cell_start : !g->keys)) {
            rc = pc_fail(ctx, PC_NOMEM, "grid buffers");
            break;
        }
        if (n == 0) {
            if (hipMemsetAsync(g->cell_start, 0, cs_bytes, st) != hipSuccess)
                rc = pc_fail(ctx, PC_HIP_ERROR, "memset cell_start");
            break;
        }
        if (built_direct) {
            // (everything is in place; the caller's array may change once this returns: the last read-back waited)
            break;
        }
        const uint32_t *sorted_keys = in_tmp ? keys2.as<uint32_t>() : keys.as<uint32_t>();
        if (dense) {
            // cell_start = exclusive scan of the per-cell counts
            rc = pc_exclusive_scan_u32(ctx, g->cell_start, g->ncell + 1, st);
            if (rc) break;
        } else {
            if (hipMemcpyAsync(g->keys, sorted_keys, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToDevice, st) !=
                hipSuccess) {
                rc = pc_fail(ctx, PC_HIP_ERROR, "copy sorted keys");
                break;
            }
            // row directory: searches for a cell then stay inside its (y, z) row
            const int64_t nrows = (int64_t)g->gp.dim[1] * g->gp.dim[2];
            if (nrows + 2 <= 16 * n + 65536) {
                const size_t rb = sizeof(uint32_t) * (size_t)(nrows + 2);
                g->row_start = (uint32_t *)ctx->alloc(rb);
                if (!g->row_start || hipMemsetAsync(g->row_start, 0, rb, st) != hipSuccess) {
                    rc = pc_fail(ctx, PC_NOMEM, "row directory");
                    break;
                }
                hipLaunchKernelGGL(k_row_runs, dim3(blocks_for(n, 256)), dim3(256), 0, st, sorted_keys, n,
                                   (uint32_t)g->gp.dim[0], g->row_start);
                rc = pc_exclusive_scan_u32(ctx, g->row_start, nrows + 2, st);
                if (rc) break;
                g->bytes += rb;
            }
        }
        const uint32_t *sv = in_tmp ? vals2.as<uint32_t>() : vals.as<uint32_t>();
        if (parent && dtype == PC_F32)
            hipLaunchKernelGGL(k_regather_records<Rec32>, dim3(blocks_for(n, 256)), dim3(256), 0, st,
                               (const Rec32 *)parent->rec, sv, n, (Rec32 *)g->rec, g->perm);
        else if (parent)
            hipLaunchKernelGGL(k_regather_records<Rec64>, dim3(blocks_for(n, 256)), dim3(256), 0, st,
                               (const Rec64 *)parent->rec, sv, n, (Rec64 *)g->rec, g->perm);
        else if (dtype == PC_F32)
            hipLaunchKernelGGL((k_gather_records<float, Rec32>), dim3(blocks_for(n, 256)), dim3(256), 0, st,
                               (const float *)xyz_dev, sv, n, (Rec32 *)g->rec, g->perm);
        else
            hipLaunchKernelGGL((k_gather_records<double, Rec64>), dim3(blocks_for(n, 256)), dim3(256), 0, st,
                               (const double *)xyz_dev, sv, n, (Rec64 *)g->rec, g->perm);
        // (the caller's array may change once this returns; a level made from another level reads only
        // buffers the index owns and need not wait)
        if (hipGetLastError() != hipSuccess || (!parent && hipStreamSynchronize(st) != hipSuccess))
            rc = pc_fail(ctx, PC_HIP_ERROR, "grid build kernels");
    } while (0);
    if (rc) {
        pc_grid_destroy(ctx, g);
        return rc;
    }
    // Unevenly filled box (surfaces, clusters, clutter): add coarser levels, twice the edge
    // each, so that a query in a sparse region finds k neighbours within one ring of SOME level.
    // (a cloud that fills its box evenly leaves exp(-mean occupancy) of the cells empty by chance)
    // (and a grid with more cells than points is mostly empty box whatever the count says)
    const double mean_occ = (double)n / (double)std::max<int64_t>(1, g->ncell);
    const double chance_filled = 1.0 - exp(-mean_occ);
    if (automatic && n > 4096 && g->occupied > 0 &&
        (mean_occ < 1.0 || (double)g->occupied < 0.85 * chance_filled * (double)g->ncell)) {
        pc_grid *fine = g;
        g->uneven = true;
        // (The first few levels now; a search that gets further makes the next one itself, pc_grid_coarser.  Most
        // clouds never look at levels 5 to 8, and sorting all points four more times was a fifth to a third of
        // their build: 2 M sphere 1.37 -> 1.0 ms, LiDAR frame 0.95 -> 0.73 ms, 13 M voxels of planes + clutter
        // 7.6 -> 5.4 ms, with the searches as fast or faster.)
        // (Big clouds get the finest level only: their searches look at what the first pass leaves open and make
        // just the levels worth a pass -- 66 M voxels of planes + clutter visit levels 0 and 2 of four.)
        const int eager = ctx->knobs.eager_levels >= 1 ? std::min(ctx->knobs.eager_levels, MAX_LEVELS)
                                                        : (n >= ctx->knobs.lazy_min ? 1 : EAGER_LEVELS);
        const int max_lv = ctx->knobs.max_levels > 0 ? std::min(ctx->knobs.max_levels, MAX_LEVELS) : eager;
        for (int lv = 1; lv < max_lv; lv++) {
            if (std::max(fine->gp.dim[0], std::max(fine->gp.dim[1], fine->gp.dim[2])) <= 4) break;
            pc_grid *c = nullptr;
            rc = grid_build_impl(ctx, nullptr, dtype, n, fine->gp.cell * 2.0, k_hint, &c, stream, fine);
            if (rc) {
                pc_grid_destroy(ctx, g);
                return rc;
            }
            fine->coarse = c;
            c->uneven = true;
            c->level = lv;
            g->bytes += c->bytes;
            fine = c;
        }
    }
    *out = g;
    return PC_OK;
}

int pc_grid_build(pc_ctx *ctx, const void *xyz_dev, int dtype, int64_t n, double cell_size, int k_hint, pc_grid **out,
                  void *stream) {
    PcLock lock(ctx, stream);  // calls on one context take turns (its block pool and read-back buffer are shared)
    if (!ctx || !out || n < 0 || n >= INT32_MAX || (n > 0 && !xyz_dev) || (dtype != PC_F32 && dtype != PC_F64))
        return pc_fail(ctx, PC_BAD_ARG, "pc_grid_build");
    PC_HIP(hipSetDevice(ctx->device));
    return grid_build_impl(ctx, xyz_dev, dtype, n, cell_size, k_hint, out, stream, nullptr);
}

}  // extern "C"

int pc_grid_stats(pc_ctx *ctx, const pc_grid *g, hipStream_t st) {
    if (!g->stats_pending) return PC_OK;
    unsigned long long *hs = reinterpret_cast<unsigned long long *>(ctx->pinned);
    PC_HIP(hipMemcpyAsync(hs, g->stats_dev, 2 * 8, hipMemcpyDeviceToHost, st));
    PC_HIP(hipStreamSynchronize(st));
    g->seen = g->n > 0 ? (double)hs[0] / (double)g->n : 0.0;
    g->occupied = (int64_t)hs[1];
    g->stats_pending = false;
    return PC_OK;
}

int pc_grid_coarser(pc_ctx *ctx, const pc_grid *g, hipStream_t st, const pc_grid **out) {
    *out = g->coarse;
    if (g->coarse || !g->uneven || g->n <= 0 || g->level < 0 || g->level + 1 >= MAX_LEVELS ||
        (ctx->knobs.max_levels > 0 && g->level + 1 >= ctx->knobs.max_levels) ||
        std::max(g->gp.dim[0], std::max(g->gp.dim[1], g->gp.dim[2])) <= 4)
        return PC_OK;
    pc_grid *c = nullptr;
    PC_TRY(grid_build_impl(ctx, nullptr, g->dtype, g->n, g->gp.cell * 2.0, 0, &c, (void *)st, g));
    c->uneven = true;
    c->level = g->level + 1;
    g->coarse = c;
    *out = c;
    return PC_OK;
}

int pc_grid_level(pc_ctx *ctx, const pc_grid *last, int level, hipStream_t st, const pc_grid **out) {
    *out = nullptr;
    if (last->coarse || !last->uneven || last->n <= 0 || last->level < 0 || level <= last->level || level >= MAX_LEVELS ||
        (ctx->knobs.max_levels > 0 && level >= ctx->knobs.max_levels))
        return PC_OK;
    // (like pc_grid_coarser level by level: none beyond a level that is down to four cells an axis)
    int d = std::max(last->gp.dim[0], std::max(last->gp.dim[1], last->gp.dim[2]));
    for (int v = last->level; v < level - 1; v++) d = d / 2 + 1;
    if (d <= 4) return PC_OK;
    pc_grid *c = nullptr;
    PC_TRY(grid_build_impl(ctx, nullptr, last->dtype, last->n, ldexp(last->gp.cell, level - last->level), 0, &c, (void *)st, last));
    c->uneven = true;
    c->level = level;
    last->coarse = c;
    *out = c;
    return PC_OK;
}

int pc_grid_finer(pc_ctx *ctx, const pc_grid *g, hipStream_t st, const pc_grid **out) {
    *out = g->finer;
    if (g->finer || g->finer_tried || g->n <= 0) return PC_OK;
    g->finer_tried = true;
    pc_grid *f = nullptr;
    PC_TRY(grid_build_impl(ctx, nullptr, g->dtype, g->n, 0.5 * g->gp.cell, 0, &f, (void *)st, g));
    if (f->gp.cell > 0.93 * g->gp.cell) {  // the cell-key range binds: no finer grid
        pc_grid_destroy(ctx, f);
        return PC_OK;
    }
    f->uneven = true;
    f->level = g->level - 1;
    g->finer = f;
    *out = f;
    return PC_OK;
}

// External queries binned into a grid: records {x, y, z, query row} in the order of the grid's cell
// keys (ties by query row) and the sorted keys themselves.  The queries have the grid's dtype.
int pc_bin_queries(pc_ctx *ctx, const pc_grid *g, const void *q_dev, int64_t nq, void *qrec_out, uint32_t *qkeys_out,
                   uint32_t *qrows_out, hipStream_t st) {
    const size_t nb = sizeof(uint32_t) * (size_t)std::max<int64_t>(nq, 1);
    Scratch keys(ctx, nb), keys2(ctx, nb), vals(ctx, nb), vals2(ctx, nb);
    if (!keys.ok() || !keys2.ok() || !vals.ok() || !vals2.ok()) return pc_fail(ctx, PC_NOMEM, "query binning scratch");
    if (nq <= 0) return PC_OK;
    if (g->dtype == PC_F32)
        hipLaunchKernelGGL(k_cell_keys32<float>, dim3(blocks_for(nq, 256)), dim3(256), 0, st, (const float *)q_dev, nq,
                           g->gp, keys.as<uint32_t>(), vals.as<uint32_t>());
    else
        hipLaunchKernelGGL(k_cell_keys32<double>, dim3(blocks_for(nq, 256)), dim3(256), 0, st, (const double *)q_dev, nq,
                           g->gp, keys.as<uint32_t>(), vals.as<uint32_t>());
    bool in_tmp = false;
    PC_TRY(pc_radix_sort_pairs32(ctx, keys.as<uint32_t>(), vals.as<uint32_t>(), keys2.as<uint32_t>(), vals2.as<uint32_t>(),
                                 nq, bits_for((uint64_t)g->ncell - 1), st, &in_tmp));
    const uint32_t *sk = in_tmp ? keys2.as<uint32_t>() : keys.as<uint32_t>();
    const uint32_t *sv = in_tmp ? vals2.as<uint32_t>() : vals.as<uint32_t>();
    if (g->dtype == PC_F32)
        hipLaunchKernelGGL((k_gather_records<float, Rec32>), dim3(blocks_for(nq, 256)), dim3(256), 0, st,
                           (const float *)q_dev, sv, nq, (Rec32 *)qrec_out, (int32_t *)nullptr);
    else
        hipLaunchKernelGGL((k_gather_records<double, Rec64>), dim3(blocks_for(nq, 256)), dim3(256), 0, st,
                           (const double *)q_dev, sv, nq, (Rec64 *)qrec_out, (int32_t *)nullptr);
    PC_HIP(hipGetLastError());
    PC_HIP(hipMemcpyAsync(qkeys_out, sk, sizeof(uint32_t) * (size_t)nq, hipMemcpyDeviceToDevice, st));
    if (qrows_out) {
        const int64_t nrows = (int64_t)g->gp.dim[1] * g->gp.dim[2];
        PC_HIP(hipMemsetAsync(qrows_out, 0, sizeof(uint32_t) * (size_t)(nrows + 2), st));
        hipLaunchKernelGGL(k_row_runs, dim3(blocks_for(nq, 256)), dim3(256), 0, st, sk, nq, (uint32_t)g->gp.dim[0], qrows_out);
        PC_HIP(hipGetLastError());
        PC_TRY(pc_exclusive_scan_u32(ctx, qrows_out, nrows + 2, st));
    }
    return PC_OK;
}

extern "C" {

void pc_grid_destroy(pc_ctx *ctx, pc_grid *g) {
    PcLock lock(ctx);  // calls on one context take turns (its block pool and read-back buffer are shared)
    if (!g) return;
    if (g->coarse) pc_grid_destroy(ctx, g->coarse);
    if (g->finer) pc_grid_destroy(ctx, g->finer);
    if (ctx) {
        if (g->rec) ctx->release(g->rec);
        if (g->cell_start) ctx->release(g->cell_start);
        if (g->keys) ctx->release(g->keys);
        if (g->row_start) ctx->release(g->row_start);
        if (g->tile_list) ctx->release(g->tile_list);
        if (g->full_tiles) ctx->release(g->full_tiles);
        if (g->stats_dev) ctx->release(g->stats_dev);
        if (g->perm) ctx->release(g->perm);
    }
    delete g;
}

int pc_grid_info(const pc_grid *g, double info[6]) {
    if (!g || !info) return PC_BAD_ARG;
    info[0] = g->gp.dim[0];
    info[1] = g->gp.dim[1];
    info[2] = g->gp.dim[2];
    info[3] = g->gp.cell;
    info[4] = (double)g->n;
    info[5] = (double)g->bytes;
    return PC_OK;
}

const int32_t *pc_grid_perm_dev(const pc_grid *g) { return g ? g->perm : nullptr; }

int pc_grid_perm_copy(pc_ctx *ctx, const pc_grid *g, int64_t s_begin, int64_t s_end, int32_t *out_dev, void *stream) {
    PcLock lock(ctx, stream);  // calls on one context take turns (its block pool and read-back buffer are shared)
    if (!ctx || !g || s_begin < 0 || s_end < s_begin || s_end > g->n || (s_end > s_begin && !out_dev))
        return pc_fail(ctx, PC_BAD_ARG, "pc_grid_perm_copy");
    PC_HIP(hipSetDevice(ctx->device));
    if (s_end > s_begin)
        PC_HIP(hipMemcpyAsync(out_dev, g->perm + s_begin, sizeof(int32_t) * (size_t)(s_end - s_begin),
                              hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return PC_OK;
}

int pc_gather_rows(pc_ctx *ctx, const double *src_dev, const int32_t *idx_dev, int64_t m, double *dst_dev,
                   void *stream) {
    PcLock lock(ctx, stream);  // calls on one context take turns (its block pool and read-back buffer are shared)
    if (!ctx || m < 0 || (m > 0 && (!src_dev || !idx_dev || !dst_dev))) return pc_fail(ctx, PC_BAD_ARG, "pc_gather_rows");
    PC_HIP(hipSetDevice(ctx->device));
    if (m == 0) return PC_OK;
    hipLaunchKernelGGL(k_gather_rows<double>, dim3(blocks_for(3 * m, 256)), dim3(256), 0, (hipStream_t)stream, src_dev,
                       idx_dev, 3 * m, dst_dev);
    PC_HIP(hipGetLastError());
    return PC_OK;
}

int pc_scatter_rows(pc_ctx *ctx, const double *src_dev, const int32_t *idx_dev, int64_t m, double *dst_dev,
                    void *stream) {
    PcLock lock(ctx, stream);
    if (!ctx || m < 0 || (m > 0 && (!src_dev || !idx_dev || !dst_dev))) return pc_fail(ctx, PC_BAD_ARG, "pc_scatter_rows");
    PC_HIP(hipSetDevice(ctx->device));
    if (m == 0) return PC_OK;
    hipLaunchKernelGGL(k_scatter_rows, dim3(blocks_for(3 * m, 256)), dim3(256), 0, (hipStream_t)stream, src_dev,
                       idx_dev, 3 * m, dst_dev);
    PC_HIP(hipGetLastError());
    return PC_OK;
}

int pc_gather_rows_f32(pc_ctx *ctx, const float *src_dev, const int32_t *idx_dev, int64_t m, float *dst_dev,
                       void *stream) {
    PcLock lock(ctx, stream);
    if (!ctx || m < 0 || (m > 0 && (!src_dev || !idx_dev || !dst_dev))) return pc_fail(ctx, PC_BAD_ARG, "pc_gather_rows_f32");
    PC_HIP(hipSetDevice(ctx->device));
    if (m == 0) return PC_OK;
    hipLaunchKernelGGL(k_gather_rows<float>, dim3(blocks_for(3 * m, 256)), dim3(256), 0, (hipStream_t)stream, src_dev,
                       idx_dev, 3 * m, dst_dev);
    PC_HIP(hipGetLastError());
    return PC_OK;
}

}  // extern "C"
