// Tile-cooperative exact kNN (k <= 50; fp32 and fp64 grids; the grid's own points or external
// queries binned into the grid).  This file: the host side (tile shapes, occupied-tile lists, the
// passes over the index levels); the kernel lives in pc_knn_tile_kernel.h.
//
// A workgroup of 1-4 waves owns one tile of cells.  The tile's candidate block (the tile plus a
// one-cell halo; its (y,z) rows are contiguous point ranges) is staged once into LDS; each lane
// then answers one query of the tile, walking its own 3x3x3 cells in lock-step with its wave.
// Selection is two-stage:
//   1. fp32 filter: key = (fp32 d2 bits with the low ORD_BITS cleared) | candidate ordinal.
//      Keys are unique u32.  A sorted list of KS = k + 2 keys per lane lives in registers
//      (L[s] = med3(L[s-1], key, L[s]), one v_med3_u32 per slot), but only keys below the list's
//      last entry can change it: those are parked in a lane-private LDS column and inserted in
//      batches.
//   2. exact finish: the KS survivors are re-measured in fp64 in the reference's operation
//      order, ranked by (d2, original index), and the result is accepted only if at least k of
//      them lie strictly below a bound that no unlisted candidate and no unscanned point can
//      beat.  Queries that cannot be certified (sparse surroundings, heavy ties, over-full
//      tiles) are flagged for the next coarser index level, then for the wave-per-query kernel.
// The consumers (mean distance, normal, ...) run on the certified list while the candidates
// are still in LDS.  HBM traffic per query is therefore the algorithmic minimum plus the halo
// re-reads served by L2.
#include <stdio.h>

#include <algorithm>
#include <vector>

#include "pc_knn_dev.h"

#include "pc_knn_tile_kernel.h"

namespace {

// ---- occupied tiles of a sparse grid -------------------------------------------------------------
__global__ __launch_bounds__(256) void k_key_heads(const uint32_t *__restrict__ keys, int64_t n,
                                                    uint8_t *__restrict__ flags) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) flags[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_tile_ids(const uint32_t *__restrict__ keys, const int32_t *__restrict__ heads,
                                                   int64_t m, int dx, int dy, int tx, int ty, int tz, int ntx, int nty,
                                                   uint32_t *__restrict__ out, uint32_t *__restrict__ vals) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    uint32_t key = keys[heads[i]];
    const uint32_t x = key % (uint32_t)dx;
    key /= (uint32_t)dx;
    const uint32_t y = key % (uint32_t)dy, z = key / (uint32_t)dy;
    out[i] = ((z / tz) * (uint32_t)nty + (y / ty)) * (uint32_t)ntx + (x / tx);
    vals[i] = (uint32_t)i;
}

__global__ __launch_bounds__(256) void k_take(const uint32_t *__restrict__ src, const int32_t *__restrict__ idx,
                                               int64_t m, uint32_t *__restrict__ dst) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < m) dst[i] = src[idx[i]];
}

// flags[tile of the key's cell] = 1 for every key (plain stores of the same value: no order matters)
__global__ __launch_bounds__(256) void k_mark_tiles(const uint32_t *__restrict__ keys, int64_t n, int dx, int dy, int tx, int ty,
                                                     int tz, int ntx, int nty, uint8_t *__restrict__ flags) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t key = keys[i];
    if (i > 0 && keys[i - 1] == key) return;  // (sorted keys: one store per occupied cell)
    const uint32_t x = key % (uint32_t)dx;
    key /= (uint32_t)dx;
    const uint32_t y = key % (uint32_t)dy, z = key / (uint32_t)dy;
    flags[((z / tz) * (uint32_t)nty + (y / ty)) * (uint32_t)ntx + (x / tx)] = 1;
}

// most tiles a grid may have for its occupied ones to be found by marking a byte per tile
constexpr uint64_t MARK_TILES_MAX = 64ull << 20;

// unique tile ids of the occupied cells, ascending (sparse grids only)
// Linear ids of the tiles (of g's tile shape) that hold at least one of n sorted cell keys.
int tile_list_from_keys(pc_ctx *ctx, const pc_grid *g, const uint32_t *keys, int64_t n, hipStream_t st, uint32_t **out,
                        int64_t *count) {
    *out = nullptr;
    *count = 0;
    const unsigned nb = (unsigned)((n + 255) / 256);
    {
        // Few enough tiles for a byte each: mark the occupied ones and compact the marks -- ascending tile ids without
        // sorting anything (five launches and one read-back where the sort below takes some twenty and three; one LiDAR
        // frame in fourteen has such a finest level, 60 k voxels in a few million tiles).
        const uint64_t mx = (uint64_t)((g->gp.dim[0] + g->tile[0] - 1) / g->tile[0]), my = (uint64_t)((g->gp.dim[1] + g->tile[1] - 1) / g->tile[1]),
                       mz = (uint64_t)((g->gp.dim[2] + g->tile[2] - 1) / g->tile[2]);
        const uint64_t mt = mx * my * mz;
        if (n > 0 && mt <= MARK_TILES_MAX && !ctx->knobs.no_mark_tiles) {
            Scratch marks(ctx, (size_t)mt);
            if (!marks.ok()) return pc_fail(ctx, PC_NOMEM, "tile marks");
            uint32_t *list = (uint32_t *)ctx->alloc(sizeof(uint32_t) * (size_t)std::max<int64_t>(std::min<int64_t>(n, (int64_t)mt), 1));
            if (!list) return pc_fail(ctx, PC_NOMEM, "tile list");
            PC_HIP(hipMemsetAsync(marks.p, 0, (size_t)mt, st));
            hipLaunchKernelGGL(k_mark_tiles, dim3(nb), dim3(256), 0, st, keys, n, g->gp.dim[0], g->gp.dim[1], g->tile[0], g->tile[1],
                               g->tile[2], (int)mx, (int)my, marks.as<uint8_t>());
            PC_HIP(hipGetLastError());
            int64_t nocc = 0;
            const int rc = pc_compact_flags(ctx, marks.as<uint8_t>(), (int64_t)mt, reinterpret_cast<int32_t *>(list), &nocc, st);
            if (rc != PC_OK) {
                ctx->release(list);
                return rc;
            }
            *out = list;
            *count = nocc;
            return PC_OK;
        }
    }
    Scratch flags(ctx, (size_t)n), heads(ctx, sizeof(int32_t) * (size_t)n);
    if (!flags.ok() || !heads.ok()) return pc_fail(ctx, PC_NOMEM, "tile list");
    hipLaunchKernelGGL(k_key_heads, dim3(nb), dim3(256), 0, st, keys, n, flags.as<uint8_t>());
    int64_t u = 0;
    PC_TRY(pc_compact_flags(ctx, flags.as<uint8_t>(), n, heads.as<int32_t>(), &u, st));
    Scratch tid(ctx, sizeof(uint32_t) * (size_t)u), tid2(ctx, sizeof(uint32_t) * (size_t)u);
    Scratch v(ctx, sizeof(uint32_t) * (size_t)u), v2(ctx, sizeof(uint32_t) * (size_t)u);
    if (!tid.ok() || !tid2.ok() || !v.ok() || !v2.ok()) return pc_fail(ctx, PC_NOMEM, "tile list");
    const int ntx = (g->gp.dim[0] + g->tile[0] - 1) / g->tile[0], nty = (g->gp.dim[1] + g->tile[1] - 1) / g->tile[1],
              ntz = (g->gp.dim[2] + g->tile[2] - 1) / g->tile[2];
    const unsigned ub = (unsigned)((u + 255) / 256);
    hipLaunchKernelGGL(k_tile_ids, dim3(ub), dim3(256), 0, st, keys, heads.as<int32_t>(), u, g->gp.dim[0], g->gp.dim[1],
                       g->tile[0], g->tile[1], g->tile[2], ntx, nty, tid.as<uint32_t>(), v.as<uint32_t>());
    uint64_t nt = (uint64_t)ntx * nty * ntz;
    int bits = 1;
    while (bits < 32 && (nt >> bits)) bits++;
    bool in_tmp = false;
    PC_TRY(pc_radix_sort_pairs32(ctx, tid.as<uint32_t>(), v.as<uint32_t>(), tid2.as<uint32_t>(), v2.as<uint32_t>(), u,
                                 bits, st, &in_tmp));
    const uint32_t *sorted = in_tmp ? tid2.as<uint32_t>() : tid.as<uint32_t>();
    hipLaunchKernelGGL(k_key_heads, dim3(ub), dim3(256), 0, st, sorted, u, flags.as<uint8_t>());
    int64_t nocc = 0;
    PC_TRY(pc_compact_flags(ctx, flags.as<uint8_t>(), u, heads.as<int32_t>(), &nocc, st));
    uint32_t *list = (uint32_t *)ctx->alloc(sizeof(uint32_t) * (size_t)std::max<int64_t>(nocc, 1));
    if (!list) return pc_fail(ctx, PC_NOMEM, "tile list");
    hipLaunchKernelGGL(k_take, dim3((unsigned)((nocc + 255) / 256)), dim3(256), 0, st, sorted, heads.as<int32_t>(), nocc,
                       list);
    PC_HIP(hipGetLastError());
    PC_HIP(hipStreamSynchronize(st));
    *out = list;
    *count = nocc;
    return PC_OK;
}

int build_tile_list(pc_ctx *ctx, const pc_grid *g, hipStream_t st) {
    return tile_list_from_keys(ctx, g, g->keys, g->n, st, &g->tile_list, &g->tile_count);
}

__global__ __launch_bounds__(256) void k_tile_nonempty(GridParams gp, CellIndex ci, int tx, int ty, int tz, int ntx,
                                                        int nty, int64_t ntile, uint8_t *__restrict__ flags) {
    int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= ntile) return;
    const int x0 = (int)(t % ntx) * tx, y0 = (int)((t / ntx) % nty) * ty, z0 = (int)(t / ((int64_t)ntx * nty)) * tz;
    const int x1 = min(x0 + tx, gp.dim[0]), y1 = min(y0 + ty, gp.dim[1]), z1 = min(z0 + tz, gp.dim[2]);
    uint32_t pop = 0;
    for (int z = z0; z < z1; z++)
        for (int y = y0; y < y1; y++) {
            const int64_t row = ((int64_t)z * gp.dim[1] + y) * gp.dim[0];
            pop += cell_begin(ci, row + x1) - cell_begin(ci, row + x0);
        }
    flags[t] = pop ? 1 : 0;
}

// ids of the non-empty tiles of a dense-table grid whose box is mostly empty
int build_tile_list_dense(pc_ctx *ctx, const pc_grid *g, hipStream_t st) {
    const int ntx = (g->gp.dim[0] + g->tile[0] - 1) / g->tile[0], nty = (g->gp.dim[1] + g->tile[1] - 1) / g->tile[1],
              ntz = (g->gp.dim[2] + g->tile[2] - 1) / g->tile[2];
    const int64_t nt = (int64_t)ntx * nty * ntz;
    if (nt > 0x7fffffff) return PC_OK;
    Scratch flags(ctx, (size_t)nt);
    g->tile_list = (uint32_t *)ctx->alloc(sizeof(uint32_t) * (size_t)std::max<int64_t>(std::min<int64_t>(nt, g->n), 1));
    if (!flags.ok() || !g->tile_list) return pc_fail(ctx, PC_NOMEM, "tile list");
    hipLaunchKernelGGL(k_tile_nonempty, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, st, g->gp, g->index(), g->tile[0],
                       g->tile[1], g->tile[2], ntx, nty, nt, flags.as<uint8_t>());
    int64_t nocc = 0;
    PC_TRY(pc_compact_flags(ctx, flags.as<uint8_t>(), nt, reinterpret_cast<int32_t *>(g->tile_list), &nocc, st));
    g->tile_count = nocc;
    return PC_OK;
}

// candidates an uneven level's tile stages (tuning knob: PC_AMD_UNEVEN_CAP)
// (An index built for few neighbours has small cells and its blocks hold a few hundred points: the smaller block lets
// more workgroups share a CU's LDS -- 2 M points, build hint k: sphere k = 1 / 4 / 8 1.50 / 1.35 / 1.41 -> 1.10 / 1.04 /
// 1.16 ms, torus 1.48 / 1.16 / 1.13 -> 1.17 / 0.97 / 0.97 ms, ICP iteration over the torus 1.32 -> 1.11 ms; planes
// and clusters 0 to 8 % faster.  From k = 16 on the full block stays: 768 lost 40 % on 4 M voxels of planes + clutter.)
int uneven_cap(const pc_ctx *ctx, const pc_grid *g) {
    if (const int e = ctx->knobs.uneven_cap) return std::min(CAND_CAP_MAX, std::max(256, e / 64 * 64));
    const int k = g ? g->k_hint : 0;
    return k >= 1 && k <= 2 ? 384 : (k >= 1 && k <= 8 ? 512 : UNEVEN_CAP);
}

// flags[i] = listed tile i holds at least `thresh` points; *points += the points of those tiles
__global__ __launch_bounds__(256) void k_tile_full(GridParams gp, CellIndex ci, int tx, int ty, int tz, int ntx, int nty,
                                                    const uint32_t *__restrict__ tiles, int64_t m, uint32_t thresh,
                                                    uint8_t *__restrict__ flags, unsigned long long *__restrict__ points) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t pop = 0;
    if (i < m) {
        const int64_t t = tiles[i];
        const int x0 = (int)(t % ntx) * tx, y0 = (int)((t / ntx) % nty) * ty, z0 = (int)(t / ((int64_t)ntx * nty)) * tz;
        const int x1 = min(x0 + tx, gp.dim[0]), y1 = min(y0 + ty, gp.dim[1]), z1 = min(z0 + tz, gp.dim[2]);
        for (int z = z0; z < z1; z++)
            for (int y = y0; y < y1; y++) {
                const int64_t row = ((int64_t)z * gp.dim[1] + y) * gp.dim[0];
                pop += cell_begin(ci, row + x1) - cell_begin(ci, row + x0);
            }
        if (pop < thresh) pop = 0;
        flags[i] = pop ? 1 : 0;
    }
    unsigned long long v = pop;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(points, v);
}

// g->full_tiles: the listed tiles with at least `thresh` points (once per grid)
int build_full_tiles(pc_ctx *ctx, const pc_grid *g, uint32_t thresh, hipStream_t st) {
    if (g->full_count >= 0) return PC_OK;
    g->full_count = 0;
    g->full_points = 0;
    const int64_t m = g->tile_count;
    if (!g->tile_list || m <= 0) return PC_OK;
    Scratch flags(ctx, (size_t)m), pick(ctx, sizeof(int32_t) * (size_t)m), pts(ctx, 256);
    if (!flags.ok() || !pick.ok() || !pts.ok()) return pc_fail(ctx, PC_NOMEM, "full tiles");
    PC_HIP(hipMemsetAsync(pts.p, 0, sizeof(unsigned long long), st));
    const int ntx = (g->gp.dim[0] + g->tile[0] - 1) / g->tile[0], nty = (g->gp.dim[1] + g->tile[1] - 1) / g->tile[1];
    hipLaunchKernelGGL(k_tile_full, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, g->gp, g->index(), g->tile[0],
                       g->tile[1], g->tile[2], ntx, nty, g->tile_list, m, thresh, flags.as<uint8_t>(),
                       pts.as<unsigned long long>());
    int64_t nfull = 0;
    PC_TRY(pc_compact_flags(ctx, flags.as<uint8_t>(), m, pick.as<int32_t>(), &nfull, st));
    unsigned long long *h = reinterpret_cast<unsigned long long *>(ctx->pinned);
    PC_HIP(hipMemcpyAsync(h, pts.p, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    PC_HIP(hipStreamSynchronize(st));
    g->full_points = (int64_t)h[0];
    if (nfull > 0) {
        g->full_tiles = (uint32_t *)ctx->alloc(sizeof(uint32_t) * (size_t)nfull);
        if (!g->full_tiles) return pc_fail(ctx, PC_NOMEM, "full tiles");
        hipLaunchKernelGGL(k_take, dim3((unsigned)((nfull + 255) / 256)), dim3(256), 0, st, g->tile_list, pick.as<int32_t>(),
                           nfull, g->full_tiles);
        PC_HIP(hipGetLastError());
        g->full_host.resize((size_t)nfull);
        PC_HIP(hipMemcpyAsync(g->full_host.data(), g->full_tiles, sizeof(uint32_t) * (size_t)nfull, hipMemcpyDeviceToHost, st));
        PC_HIP(hipStreamSynchronize(st));  // (pick goes back to the pool)
    }
    g->full_count = nfull;
    return PC_OK;
}

// every query of the range open, nothing known about its neighbours yet
template <typename REC>
__global__ __launch_bounds__(256) void k_open_range(const REC *__restrict__ rec, int64_t s_begin, int64_t s_end,
                                                     uint8_t *__restrict__ qflag, float *__restrict__ qub) {
    const int64_t s = s_begin + (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= s_end) return;
    double x, y, z;
    int32_t id;
    RecTraits<REC>::load(rec + s, x, y, z, id);
    qflag[id] = 1;
    qub[id] = INFINITY;
}

// The open queries of the listed tiles whose block -- the tile and a cell around it, a superset of every member's
// 27 cells -- holds fewer than 2 k points are postponed like the list kernel would postpone them one by one (too few
// candidates to find k of them within reach; the flag names the first level whose cells, eight times the volume each,
// promise 4 k), without a look at the queries themselves: one thread per tile, sixteen row look-ups, a flag per point.
// (66 M voxels of planes + clutter: 29.6 M of the 30.8 M queries the first list pass looked up, 21 ms for 1.2 M answers.)
template <typename REC>
__global__ __launch_bounds__(256) void k_tile_postpone(GridParams gp, CellIndex ci, const REC *__restrict__ rec, int tx, int ty,
                                                        int tz, int ntx, int nty, const uint32_t *__restrict__ tiles, int64_t m,
                                                        int k, int flag_max, int flag_top, int64_t s_begin, int64_t s_end,
                                                        uint8_t *__restrict__ qflag) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const int64_t t = tiles[i];
    const int x0 = (int)(t % ntx) * tx, y0 = (int)((t / ntx) % nty) * ty, z0 = (int)(t / ((int64_t)ntx * nty)) * tz;
    const int x1 = min(x0 + tx, gp.dim[0]), y1 = min(y0 + ty, gp.dim[1]), z1 = min(z0 + tz, gp.dim[2]);
    const int bx0 = max(x0 - 1, 0), bx1 = min(x1 + 1, gp.dim[0]);
    uint32_t pop = 0;
    for (int z = max(z0 - 1, 0); z < min(z1 + 1, gp.dim[2]); z++)
        for (int y = max(y0 - 1, 0); y < min(y1 + 1, gp.dim[1]); y++) {
            const int64_t row = ((int64_t)z * gp.dim[1] + y) * gp.dim[0];
            uint32_t b, e;
            cell_range(ci, row + bx0, row + bx1, b, e);
            pop += e - b;
        }
    if (pop >= 2u * (uint32_t)k) return;
    // (a member's 27 cells hold about 27 / 64 of the block's points: the growth is reckoned from that)
    uint32_t pred = max((pop * 27u + 63u) / 64u, 1u) * 8u;
    int f = flag_max + 1;
    while (pred < 4u * (uint32_t)k && f < flag_top) {
        pred *= 8u;
        f++;
    }
    for (int z = z0; z < z1; z++)
        for (int y = y0; y < y1; y++) {
            const int64_t row = ((int64_t)z * gp.dim[1] + y) * gp.dim[0];
            uint32_t b, e;
            cell_range(ci, row + x0, row + x1, b, e);
            for (uint32_t j = max((int64_t)b, s_begin); j < min((int64_t)e, s_end); j++) {
                double px, py, pz;
                int32_t id;
                RecTraits<REC>::load(rec + j, px, py, pz, id);
                if (qflag[id] == 1) qflag[id] = (uint8_t)f;
            }
        }
}

}  // namespace

void pc_dense_tile_shape(const pc_ctx *ctx, double occ, const int32_t dim[3], int best[3], int *waves_out, double *pop_out) {
    int waves = MAX_TILE_WAVES;  // measured at 10 M, k = 16: 4 waves 2.78 ms, 2 waves 3.27 ms (LDS-bound occupancy)
    if (const int e = ctx->knobs.tile_waves) waves = std::min(MAX_TILE_WAVES, std::max(1, e));
    best[0] = best[1] = best[2] = 1;
    double best_pop = -1.0, best_block = 0.0;
    for (int tz = 1; tz <= 6; tz++)
        for (int ty = 1; ty <= 6; ty++)
            for (int tx = 1; tx + 2 <= MAX_NX; tx++) {
                if ((ty + 2) * (tz + 2) > MAX_ROWS) continue;
                if (tx > std::max(1, (int)dim[0]) || ty > std::max(1, (int)dim[1]) || tz > std::max(1, (int)dim[2])) continue;
                const double pop = occ * tx * ty * tz, block = (double)(tx + 2) * (ty + 2) * (tz + 2);
                const double cand = occ * block;
                if (pop + 1.5 * sqrt(pop) > (double)(PC_WAVE * waves) && !(tx == 1 && ty == 1 && tz == 1)) continue;
                if (cand + 6.0 * sqrt(cand) + 8.0 > (double)CAND_CAP_MAX && !(tx == 1 && ty == 1 && tz == 1)) continue;
                if (pop > best_pop * 1.0001 || (pop > best_pop * 0.9999 && block < best_block)) {
                    best_pop = pop;
                    best_block = block;
                    best[0] = tx;
                    best[1] = ty;
                    best[2] = tz;
                }
            }
    {   // tuning knob: "tx,ty,tz"
        const int *v = ctx->knobs.tile;
        if (v[0] >= 1 && v[0] + 2 <= MAX_NX && v[1] >= 1 && v[2] >= 1 && (v[1] + 2) * (v[2] + 2) <= MAX_ROWS) {
            for (int a = 0; a < 3; a++) best[a] = std::min(v[a], std::max(1, (int)dim[a]));
            best_pop = occ * best[0] * best[1] * best[2];
        }
    }
    // no more waves than the expected population needs
    waves = std::min(waves, std::max(1, (int)ceil((best_pop + 1.5 * sqrt(std::max(best_pop, 0.0))) / PC_WAVE)));
    if (waves_out) *waves_out = waves;
    if (pop_out) *pop_out = best_pop;
}

namespace {

// chooses g->tile / g->tile_cap (once per grid)
// listed: how many queries the level's pass is to answer, where the caller knows (coarser levels of an
// uneven index, whose tiles are settled when the level is first searched); 0: unknown.
int choose_tile(pc_ctx *ctx, const pc_grid *g, hipStream_t st, int64_t listed = 0) {
    if (g->tile[0] > 0) return PC_OK;
    double occ = (double)g->n / (double)std::max<int64_t>(1, g->ncell);
    const bool dense = !g->uneven;
    if (dense) {
        // points per cell where the cloud is (the last layer of cells of an axis is only partly inside the box): what
        // a tile's population follows -- n / ncell runs a few per cent low and lets tiles overflow their lanes
        double box = 1.0;
        for (int a = 0; a < 3; a++) box *= std::max(g->bb_max[a] - g->bb_min[a], g->gp.cell);
        occ = std::max(occ, (double)g->n * g->gp.cell * g->gp.cell * g->gp.cell / box);
    }
    int best[3] = {1, 1, 1};
    double expect;
    int waves = 1;
    if (dense) {
        // The cloud fills its box.  A workgroup of `waves` waves shares one staged block, so a
        // bigger tile pays less halo per query and less LDS per wave.  Take the fullest shape whose
        // expected population (plus 1.5 sigma) still fits the workgroup's lanes; among equals the one
        // with the smallest block.
        pc_dense_tile_shape(ctx, occ, g->gp.dim, best, &waves, nullptr);
        expect = occ * (best[0] + 2) * (best[1] + 2) * (best[2] + 2);
        expect += 6.0 * sqrt(expect) + 8.0;  // Poisson mean + 6 sigma
    } else {
        // Unevenly filled box: every level of the index serves the points whose surroundings hold a
        // handful of points per cell AT THAT LEVEL (denser ones were answered by a finer level,
        // sparser ones wait for a coarser one).  Start from a 2x2x2 tile (a single cell on the coarser
        // levels, whose open points sit in fuller cells) and grow it while the occupied tiles hold
        // too few points to fill a wave (thin surfaces, scan lines); the shape is settled below.
        int base = g->level > 0 ? 1 : 2;
        if (const int e = g->level > 0 ? ctx->knobs.uneven_tile1 : ctx->knobs.uneven_tile0) base = std::max(1, e);
        for (int a = 0; a < 3; a++) best[a] = std::min(std::min(base, a == 0 ? MAX_NX - 2 : 6), std::max(1, (int)g->gp.dim[a]));
        expect = uneven_cap(ctx, g);
        waves = 1;  // (passes over few tiles take two: uneven_waves below)
    }
    int cap = ((int)expect + 63) / 64 * 64;
    g->tile_cap = std::min(CAND_CAP_MAX, std::max(256, cap));
    for (int a = 0; a < 3; a++) g->tile[a] = best[a];
    g->tile_waves = waves;
    for (int grow = 0; g->n > 0 && (g->keys || g->uneven); grow++) {
        if (g->keys)
            PC_TRY(build_tile_list(ctx, g, st));
        else
            PC_TRY(build_tile_list_dense(ctx, g, st));
        if (!g->uneven || !g->tile_list || grow >= 3) break;
        // grow only where the occupied cells are evenly filled (a surface of even density): the mean
        // says nothing about a cloud whose points sit mostly in a few full tiles among many empty ones
        PC_TRY(pc_grid_stats(ctx, g, st));
        // A coarser level that is to answer millions of queries, eight or more in every occupied cell (the thin
        // part of a big cloud: clutter between surfaces): a tile of one cell would give a wave a dozen
        // queries, eight cells fill it; their block of 64 cells needs the larger capacity.  (66 M voxels of
        // planes + clutter, 26 M queries answered at level 2: 116 -> 52 ms for that pass.  Small clouds and
        // surfaces lose with the bigger tile, hence the thresholds.)
        if (grow == 0 && g->level > 0 && g->tile[0] == 1 && listed >= (1ll << 20) && listed >= 8 * std::max<int64_t>(1, g->occupied) &&
            !ctx->knobs.uneven_tile1) {
            ctx->release(g->tile_list);
            g->tile_list = nullptr;
            g->tile_count = 0;
            // (3x3x3 since over-full tiles split twice, down to single cells: the same cloud's level 2, 48 -> 35 ms)
            for (int a = 0; a < 3; a++) g->tile[a] = std::min(ctx->knobs.thin_tile > 0 ? ctx->knobs.thin_tile : 3, std::max(1, (int)g->gp.dim[a]));
            g->tile_cap = CAND_CAP_MAX;
            grow = 2;  // (one more turn of the loop to list the tiles, then done)
            continue;
        }
        const double pop = (double)g->n / (double)std::max<int64_t>(1, g->tile_count);
        const double cell_mean = (double)g->n / (double)std::max<int64_t>(1, g->occupied);
        if (pop >= 16.0 || g->occupied <= 0 || g->seen > 2.0 * (cell_mean + 1.0)) break;
        int t2[3];
        bool changed = false;
        for (int a = 0; a < 3; a++) {
            t2[a] = std::min(g->tile[a] + 1, std::max(1, (int)g->gp.dim[a]));
            if (a == 0) t2[a] = std::min(t2[a], MAX_NX - 2);
            changed |= t2[a] != g->tile[a];
        }
        if (!changed || (t2[1] + 2) * (t2[2] + 2) > MAX_ROWS) break;
        ctx->release(g->tile_list);
        g->tile_list = nullptr;
        g->tile_count = 0;
        for (int a = 0; a < 3; a++) g->tile[a] = t2[a];
    }
    return PC_OK;
}

// Waves per tile of an uneven level's pass.  With few tiles the fullest ones (hundreds of queries next to
// a LiDAR sensor) set the length of the pass and a second wave halves it (config 3 frame: 2.21 -> 1.94 ms);
// with many tiles the GPU is full either way and the second wave only costs LDS (66 M voxels of planes +
// clutter: 304 -> 326 ms).  Four waves measured slower everywhere.
int uneven_waves(const pc_ctx *ctx, int64_t tiles) {
    if (const int e = ctx->knobs.uneven_waves) return std::min(MAX_TILE_WAVES, std::max(1, e));
    return tiles < 8192 ? 2 : 1;
}

// After the multi-level passes: the points still flagged, as (sorted position at level 0, bound).
template <typename REC>
__global__ __launch_bounds__(256) void k_collect_flagged(const REC *__restrict__ rec, int64_t s_begin, int64_t s_end,
                                                          const uint8_t *__restrict__ qflag,
                                                          const float *__restrict__ qub, int32_t *__restrict__ list,
                                                          float *__restrict__ ub, uint32_t *__restrict__ count,
                                                          uint32_t cap) {
    const int64_t s = s_begin + (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= s_end) return;
    double x, y, z;
    int32_t id;
    RecTraits<REC>::load(rec + s, x, y, z, id);
    if (qflag[id]) {
        const uint32_t p = atomicAdd(count, 1u);
        if (p < cap) {
            list[p] = (int32_t)s;
            ub[p] = qub[id];
        }
    }
}

// The same from the flags' side, for searches that hold a map from original index to output row (the compact
// outputs of a query range: row = sorted position - s_begin): the flags are read in order and only the open
// points look anything up, where the walk over the records above reads a flag at a random place for every record.
// out[0] / out[1] = first position of an ascending list whose entry is >= lo / >= hi (one lane each)
__global__ void k_list_bounds(const uint32_t *__restrict__ list, uint32_t n, uint32_t lo, uint32_t hi, uint32_t *__restrict__ out) {
    if (threadIdx.x >= 2) return;
    const uint32_t key = threadIdx.x == 0 ? lo : hi;
    uint32_t a = 0, b = n;
    while (a < b) {
        const uint32_t mid = (a + b) >> 1;
        if (list[mid] < key)
            a = mid + 1;
        else
            b = mid;
    }
    out[threadIdx.x] = a;
}

// counts[f] = points whose flag is f (f < 15), counts[15] = points with any larger flag (FLAG_CROWDED)
__global__ __launch_bounds__(256) void k_flag_hist(const uint8_t *__restrict__ qflag, int64_t npt, uint32_t *__restrict__ counts) {
    __shared__ uint32_t h[16];
    if (threadIdx.x < 16) h[threadIdx.x] = 0;
    __syncthreads();
    uint32_t mine[16];
#pragma unroll
    for (int b = 0; b < 16; b++) mine[b] = 0;
    const int64_t base = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 16;
    if (base + 16 <= npt && (reinterpret_cast<uintptr_t>(qflag) & 15) == 0) {
        const uint4 v = *reinterpret_cast<const uint4 *>(qflag + base);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const uint32_t f = (w[i >> 2] >> (8 * (i & 3))) & 255u;
#pragma unroll
            for (int b = 1; b < 16; b++) mine[b] += (b < 15 ? f == (uint32_t)b : f >= 15u) ? 1u : 0u;
        }
    } else {
        for (int i = 0; i < 16 && base + i < npt; i++) {
            const uint32_t f = qflag[base + i];
#pragma unroll
            for (int b = 1; b < 16; b++) mine[b] += (b < 15 ? f == (uint32_t)b : f >= 15u) ? 1u : 0u;
        }
    }
#pragma unroll
    for (int b = 1; b < 16; b++) {
        uint32_t v = mine[b];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(&h[b], v);
    }
    __syncthreads();
    if (threadIdx.x < 16 && threadIdx.x > 0 && h[threadIdx.x]) atomicAdd(counts + threadIdx.x, h[threadIdx.x]);
}

constexpr int COLLECT_PER_THREAD = 16;
__global__ __launch_bounds__(256) void k_collect_flagged_by_id(const uint8_t *__restrict__ qflag, const float *__restrict__ qub,
                                                                const int32_t *__restrict__ slot, int64_t npt, int64_t s_begin,
                                                                int32_t *__restrict__ list, float *__restrict__ ub,
                                                                uint32_t *__restrict__ count, uint32_t cap) {
    // 4096 points per workgroup, ONE atomic on the list's counter per workgroup that has open points at all (one per
    // open point, or per wave, and the counter's word is the bottleneck: 9 ms for 2.2 M open points)
    __shared__ uint32_t wsum[4], base;
    const int64_t i0 = (int64_t)blockIdx.x * (256 * COLLECT_PER_THREAD) + threadIdx.x;
    uint32_t open = 0;  // bit u: the thread's u-th point is open
#pragma unroll
    for (int u = 0; u < COLLECT_PER_THREAD; u++) {
        const int64_t i = i0 + (int64_t)u * 256;
        open |= (i < npt && qflag[i]) ? 1u << u : 0u;
    }
    const uint32_t mine = (uint32_t)__popc(open);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t x = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t tot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        base = tot ? atomicAdd(count, tot) : 0u;
    }
    __syncthreads();
    uint32_t p = base + x - mine;
    for (int w = 0; w < wave; w++) p += wsum[w];
#pragma unroll
    for (int u = 0; u < COLLECT_PER_THREAD; u++) {
        if (open >> u & 1u) {
            const int64_t i = i0 + (int64_t)u * 256;
            if (p < cap) {
                list[p] = (int32_t)(s_begin + slot[i]);
                ub[p] = qub[i];
            }
            p++;
        }
    }
}

}  // namespace

int pc_launch_tile_scan(pc_ctx *ctx, const pc_grid *g, int mode, ScanArgs &A, hipStream_t st, bool *handled) {
    *handled = false;
    if (A.queries != nullptr || A.k > 50 || mode == M_COUNT || A.nq <= 0) return PC_OK;
    if (ctx->knobs.exact_scan_only) return PC_OK;  // A/B switch: one-query-per-lane fp64 scan only
    // the levels of the index, finest first; every level gets a tile pass, a coarser one answering
    // only what the finer ones flagged
    const pc_grid *lv[MAX_LEVELS];
    int nlv = 0;
    for (const pc_grid *l = g; l && nlv < MAX_LEVELS; l = l->coarse) lv[nlv++] = l;
    for (int i = 0; i < nlv; i++) {
        if (i > 0 && lv[i]->uneven && lv[i]->tile[0] <= 0) continue;  // (settled when the level is first searched, below)
        PC_TRY(choose_tile(ctx, lv[i], st));
        int64_t nt = 1;
        for (int a = 0; a < 3; a++) nt *= (lv[i]->gp.dim[a] + lv[i]->tile[a] - 1) / lv[i]->tile[a];
        if (lv[i]->tile_list) nt = lv[i]->tile_count;
        if (nt > 0x7fffffff) return PC_OK;
    }
    A.s_end = A.s_begin + A.nq;
    A.flag_cap = (uint32_t)A.nq;
    A.seed_scale = ctx->knobs.seed_scale >= 1.001 ? ctx->knobs.seed_scale : 1.001;
    // [counts (256 B)] [list 1] [upper bounds 1] [list 2] [per-point bounds] [per-point flags]; the
    // block returns to the pool when this function ends, which is safe because everything that
    // touches it is ordered on `st`
    const size_t npt = A.qrec ? (size_t)A.nq : (size_t)pc_index_space(g);  // per-point arrays are indexed by original index
    // (the per-point arrays of a ladder of levels: also where a big cloud comes with its finest level only and the
    // search makes the others)
    const bool ladder = nlv > 1 || (g->uneven && g->n >= ctx->knobs.lazy_min && ctx->knobs.eager_levels == 0);
    Scratch flags(ctx, 3 * sizeof(int32_t) * (size_t)A.nq + 256 + (ladder ? 9 * npt + 128 : 0));
    if (!flags.ok()) return pc_fail(ctx, PC_NOMEM, "re-run lists");
    A.flag_count = flags.as<uint32_t>();
    A.flag2_count = A.flag_count + 16;
    A.flag_list = reinterpret_cast<int32_t *>(flags.as<char>() + 256);
    A.flag_ub = reinterpret_cast<float *>(A.flag_list + A.nq);
    A.flag2_list = A.flag_list + 2 * A.nq;
    PC_HIP(hipMemsetAsync(A.flag_count, 0, 256, st));
    uint8_t *qflag = nullptr;
    int32_t *open_list = nullptr;
    if (ladder) {
        A.qub_arr = reinterpret_cast<float *>(A.flag2_list + A.nq);
        open_list = reinterpret_cast<int32_t *>(A.qub_arr + npt);
        qflag = reinterpret_cast<uint8_t *>(open_list + npt);
        PC_HIP(hipMemsetAsync(qflag, 0, npt, st));
    }
    // coarser levels: a tile pass per occupied tile, or the open queries dealt 64 to a wave
    const PcKnobs &knobs = ctx->knobs;
    const bool batch_passes = !knobs.no_batch;
    int64_t open_prev = 0;  // queries still open after the previous pass
    const int64_t batch_min = 65536;
    int list_level = -1, binned_level = 0;  // the level open_list / qrec_l were made for
    int64_t list_count = 0;
    const int batch_max_ks = knobs.batch_ks;
    const int64_t batch_ratio = knobs.batch_ratio;
    const int ks = A.k <= 2 ? 4 : (A.k <= 8 ? 10 : (A.k <= 16 ? 18 : (A.k <= 20 ? 22 : (A.k <= 32 ? 34 : 52))));
    // external queries arrive binned into level 0; a coarser level bins them again into its own cells
    const void *qrec0 = A.qrec;
    const CellIndex qci0 = A.qci;
    const size_t qrec_bytes = (g->dtype == PC_F32 ? sizeof(Rec32) : sizeof(Rec64)) * (size_t)A.nq;
    Scratch qrec_l(ctx, A.qrec && ladder ? qrec_bytes : 0), qkeys_l(ctx, A.qrec && ladder ? sizeof(uint32_t) * (size_t)A.nq : 0);
    if (!qrec_l.ok() || !qkeys_l.ok()) return pc_fail(ctx, PC_NOMEM, "query records");
    // Postponed queries (self-queries of k-nearest searches on multi-level indexes): a list pass that finds
    // far too few candidates around a query names, in the query's flag, the first level worth trying again,
    // and no pass before that level lists the query.  (66 M voxels of planes + clutter: the 29 M clutter
    // points, answered at level 2, no longer ride through a pass at level 1 that can only look up their cells.)
    const bool hinted = qflag != nullptr && !A.qrec && A.radius2 < 0.0 && batch_passes && ks <= batch_max_ks &&
                        !knobs.no_hints;
    // (levels a search may still make, pc_grid_coarser, count: a postponed query may name one of them)
    int nlv_possible = nlv;
    if (g->uneven)
        for (int d = std::max(lv[nlv - 1]->gp.dim[0], std::max(lv[nlv - 1]->gp.dim[1], lv[nlv - 1]->gp.dim[2]));
             d > 4 && nlv_possible < MAX_LEVELS && !(knobs.max_levels > 0 && nlv_possible >= knobs.max_levels); d = d / 2 + 1)
            nlv_possible++;
    A.flag_top = hinted ? nlv_possible : 0;
    // (a big cloud comes with its finest level only; the levels worth a pass are made as the flags ask for them)
    const bool lazy = hinted && nlv == 1 && g->uneven && g->n >= knobs.lazy_min && knobs.eager_levels == 0;
    uint32_t *open_counter = A.flag_count + 29;  // (24..28: counters of instrumented builds)
    int64_t open_all = 0;
    for (int i = 0; i < nlv; i++) {
        const pc_grid *l = lv[i];
        if (A.qrec && i > 0) {
            if (binned_level != i) PC_TRY(pc_bin_queries(ctx, l, A.qraw, A.nq, qrec_l.p, qkeys_l.as<uint32_t>(), nullptr, st));
            binned_level = i;
            A.qrec = qrec_l.p;
            A.qci = CellIndex{nullptr, qkeys_l.as<uint32_t>(), (uint32_t)A.nq, nullptr, 0u};
        }
        A.gp = l->gp;
        A.rec = l->rec;
        A.ci = l->index();
        if (l->tile[0] <= 0) PC_TRY(choose_tile(ctx, l, st, hinted && list_level == i ? list_count : 0));
        int64_t nt = 1;
        for (int a = 0; a < 3; a++) {
            A.tile[a] = l->tile[a];
            A.ntile[a] = (l->gp.dim[a] + l->tile[a] - 1) / l->tile[a];
            nt *= A.ntile[a];
        }
        A.tile_list = l->tile_list;
        if (l->tile_list) nt = l->tile_count;
        // A level whose (open) queries are spread thinly over its tiles -- fewer than a dozen per occupied
        // tile; external queries on coarser levels always -- is answered from the list of open queries,
        // 64 to a wave, instead of a workgroup per tile (measured on surfaces, clusters and planes + clutter:
        // thresholds from 8 to "always" within 10 % of each other, 3 clearly slower).  Below 64 Ki queries the
        // list would not fill the GPU's wave slots and the tile pass, one workgroup per tile, wins -- unless the
        // tiles outnumber the queries, most of them holding none.
        const int64_t listed = hinted && list_level == i ? list_count : open_prev;  // queries this pass answers
        const bool batch = batch_passes && ks <= batch_max_ks && qflag != nullptr &&
                           (i > 0 ? (A.qrec || ((listed >= batch_min || listed < nt) && listed < batch_ratio * std::max<int64_t>(nt, 1)))
                                  : (!A.qrec && l->uneven && A.nq >= batch_min && A.nq < batch_ratio * std::max<int64_t>(nt, 1)));
        if (A.qrec && l->tile_list && !batch) {
            // the grid's list names the tiles that hold POINTS; external queries need the tiles that hold
            // queries (the block returns to the pool right away: everything using it is ordered on st)
            uint32_t *ql = nullptr;
            PC_TRY(tile_list_from_keys(ctx, l, A.qci.keys, (int64_t)A.qci.n, st, &ql, &nt));
            A.tile_list = ql;
            ctx->release(ql);
        }
        A.cand_cap = l->tile_cap;
        A.halves = l->halves && !l->uneven && g->dtype == PC_F32 ? 1 : 0;
        A.qflag_in = i == 0 ? nullptr : qflag;
        A.qflag_out = qflag;  // NULL with a single level: uncertified queries go straight to the list
        A.flag_max = hinted ? l->level + 1 : 255;
        A.count_crowded = i == 0;
        // First pass of an uneven cloud that the list kernel would take: where the cloud is dense (surfaces
        // in a sea of clutter) whole tiles are full of queries that share one candidate block, which is the
        // tile kernel's case (fp32 filter on an LDS block against fp64 distances from global memory).  The
        // full tiles go first, as a pass over "open" queries; the list kernel then takes what is left.
        bool mixed = false;
        // (Measured: 66 M voxels of planes + clutter, k = 32: 348 -> 304 ms; thresholds of 16 to 64 points per
        // full tile alike.  Small clouds would pay more for the two extra read-backs than they gain.)
        if (batch && i == 0 && !A.qrec && l->tile_list && !knobs.no_mixed &&
            l->n >= knobs.mixed_min) {
            const uint32_t thresh = knobs.full_tile;
            PC_TRY(build_full_tiles(ctx, l, thresh, st));
            mixed = l->full_count > 0 && l->full_points >= l->n / 4;
        }
        if (mixed) {
            const unsigned nbq = (unsigned)((A.nq + 255) / 256);
            if (g->dtype == PC_F32)
                hipLaunchKernelGGL(k_open_range<Rec32>, dim3(nbq), dim3(256), 0, st, (const Rec32 *)l->rec, A.s_begin, A.s_end,
                                   qflag, A.qub_arr);
            else
                hipLaunchKernelGGL(k_open_range<Rec64>, dim3(nbq), dim3(256), 0, st, (const Rec64 *)l->rec, A.s_begin, A.s_end,
                                   qflag, A.qub_arr);
            A.qflag_in = qflag;
            A.count_crowded = 0;  // (the list pass below sees every query the tile pass leaves open)
            // a share of the sorted positions lies in a slab of z cells: only the full tiles of that slab
            int64_t f0 = 0, f1 = l->full_count;
            if (A.s_begin > 0 || A.s_end < l->n) {
                const size_t rb = g->dtype == PC_F32 ? sizeof(Rec32) : sizeof(Rec64);
                char *hz = reinterpret_cast<char *>(ctx->pinned);
                PC_HIP(hipMemcpyAsync(hz, (const char *)l->rec + rb * (size_t)A.s_begin, rb, hipMemcpyDeviceToHost, st));
                PC_HIP(hipMemcpyAsync(hz + 64, (const char *)l->rec + rb * (size_t)(A.s_end - 1), rb, hipMemcpyDeviceToHost, st));
                PC_HIP(hipStreamSynchronize(st));
                int64_t tzr[2];
                for (int e = 0; e < 2; e++) {
                    const double z = g->dtype == PC_F32 ? (double)reinterpret_cast<const Rec32 *>(hz + 64 * e)->z
                                                        : reinterpret_cast<const Rec64 *>(hz + 64 * e)->z;
                    const double t = floor((z - l->gp.o[2]) * l->gp.s);
                    const int cz = !(t > 0.0) ? 0 : (t >= (double)l->gp.dim[2] ? l->gp.dim[2] - 1 : (int)t);
                    tzr[e] = cz / l->tile[2];
                }
                const uint64_t per_slab = (uint64_t)A.ntile[0] * (uint64_t)A.ntile[1];
                const uint64_t lo_id = (uint64_t)tzr[0] * per_slab, hi_id = (uint64_t)(tzr[1] + 1) * per_slab;
                f0 = std::lower_bound(l->full_host.begin(), l->full_host.end(), lo_id,
                                      [](uint32_t a, uint64_t b) { return (uint64_t)a < b; }) - l->full_host.begin();
                f1 = std::lower_bound(l->full_host.begin(), l->full_host.end(), hi_id,
                                      [](uint32_t a, uint64_t b) { return (uint64_t)a < b; }) - l->full_host.begin();
            }
            A.tile_list = l->full_tiles + f0;
            // (Full tiles are where the cloud is dense.  Where they hold a hundred points and more, their blocks
            // of 64 cells often exceed the usual capacity and the tile falls apart into its eight parts: the
            // largest block keeps them whole -- 66 M voxels of planes + clutter, 137 points per full tile: 72 ->
            // 60 ms for this pass.  Thinner ones only lose LDS occupancy to it: 4 M points, 79 per full tile,
            // 16.5 -> 17.5 ms for the whole search.)
            if (!knobs.uneven_cap && l->full_points >= 100 * l->full_count) A.cand_cap = CAND_CAP_MAX;
            const int nw = uneven_waves(ctx, f1 - f0);
            const size_t lds = sizeof(float) * 4 * ((size_t)A.cand_cap + 2) + sizeof(uint32_t) * (3 * MAX_ROWS + 4) +
                               (A.halves ? 2 : 1) * sizeof(uint16_t) * MAX_ROWS * (MAX_NX + 1) +
                               nw * std::max(sizeof(uint32_t) * PC_WAVE * (SEL_SLOTS + 1), sizeof(uint16_t) * PC_WAVE * (size_t)A.k);
            if (knobs.debug)
                fprintf(stderr, "[pc_amd]   full-tile pass 0: %lld of %lld tiles, %lld of %lld points\n", (long long)l->full_count,
                        (long long)l->tile_count, (long long)l->full_points, (long long)l->n);
            if (f1 <= f0) {
            } else if (g->dtype == PC_F32)
                pc_tile_launch_f32_split(ks, mode, (unsigned)(f1 - f0), nw, lds, st, A);
            else
                pc_tile_launch_f64_split(ks, mode, (unsigned)(f1 - f0), nw, lds, st, A);
            PC_HIP(hipGetLastError());
            A.tile_list = l->tile_list;
            A.cand_cap = l->tile_cap;
            A.count_crowded = 1;
            if (hinted && A.flag_top > A.flag_max && !knobs.no_postpone) {
                const unsigned nbt = (unsigned)((l->tile_count + 255) / 256);
                if (g->dtype == PC_F32)
                    hipLaunchKernelGGL(k_tile_postpone<Rec32>, dim3(nbt), dim3(256), 0, st, l->gp, l->index(), (const Rec32 *)l->rec, l->tile[0],
                                       l->tile[1], l->tile[2], A.ntile[0], A.ntile[1], l->tile_list, l->tile_count, A.k, A.flag_max,
                                       A.flag_top, A.s_begin, A.s_end, qflag);
                else
                    hipLaunchKernelGGL(k_tile_postpone<Rec64>, dim3(nbt), dim3(256), 0, st, l->gp, l->index(), (const Rec64 *)l->rec, l->tile[0],
                                       l->tile[1], l->tile[2], A.ntile[0], A.ntile[1], l->tile_list, l->tile_count, A.k, A.flag_max,
                                       A.flag_top, A.s_begin, A.s_end, qflag);
                PC_HIP(hipGetLastError());
            }
            int64_t nopen = 0;
            PC_TRY(pc_batch_open_list(ctx, g->dtype, l->rec, l->n, qflag, A.flag_max, open_list, &nopen, &open_all, open_counter, st));
            if (knobs.debug) fprintf(stderr, "[pc_amd]   batch pass 0: %lld open queries\n", (long long)nopen);
            PC_TRY(pc_launch_batch_scan(ctx, g->dtype, ks, mode, A, open_list, nopen, st));
        } else if (batch) {
            int64_t nopen = A.nq;
            if (i > 0 && list_level == i)
                nopen = list_count;  // made when the previous pass was counted
            else if (i > 0)
                PC_TRY(pc_batch_open_list(ctx, g->dtype, A.qrec ? A.qrec : l->rec, A.qrec ? A.nq : l->n, qflag, A.flag_max, open_list,
                                          &nopen, &open_all, open_counter, st));
            if (knobs.debug) fprintf(stderr, "[pc_amd]   batch pass %d: %lld open queries\n", i, (long long)nopen);
            PC_TRY(pc_launch_batch_scan(ctx, g->dtype, ks, mode, A, i > 0 ? open_list : nullptr, nopen, st));
        } else if (hinted && i > 0 && list_level == i && list_count == 0) {
            // (no open query starts at or before this level)
        } else {
            if (nt <= 0) continue;
            A.tile_base = 0;
            if (i > 0 && !A.qrec && A.tile_list && (A.s_begin > 0 || A.s_end < g->n) && A.s_end > A.s_begin) {
                // A query shard on a coarser level: the listed tiles are in ascending order of their ids, z slowest, so
                // the tiles that can hold the shard's queries are one run of the list -- found from the z cells of the
                // shard's first and last record on the finest level (a layer to spare either side; a tile outside the
                // run holds no flagged query, but finding that out means reading the flags of all its points: a slab
                // rank of eight on 66 M voxels, level 2: 8.9 ms with a fifth of the tiles in the halo).
                const size_t rsz = g->dtype == PC_F32 ? sizeof(Rec32) : sizeof(Rec64), zoff = g->dtype == PC_F32 ? 8 : 16;
                char *hp = reinterpret_cast<char *>(ctx->pinned);
                const char *rb = reinterpret_cast<const char *>(g->rec);
                PC_HIP(hipMemcpyAsync(hp, rb + rsz * (size_t)A.s_begin + zoff, 8, hipMemcpyDeviceToHost, st));
                PC_HIP(hipMemcpyAsync(hp + 8, rb + rsz * (size_t)(A.s_end - 1) + zoff, 8, hipMemcpyDeviceToHost, st));
                PC_HIP(hipStreamSynchronize(st));
                double zr[2];
                for (int e = 0; e < 2; e++) {
                    const double z = g->dtype == PC_F32 ? (double)*reinterpret_cast<const float *>(hp + 8 * e)
                                                        : *reinterpret_cast<const double *>(hp + 8 * e);
                    // (the record's own cell layer on the finest level, its lower / upper face)
                    const double t = floor((z - g->gp.o[2]) * g->gp.s);
                    zr[e] = g->gp.o[2] + (t + (e == 0 ? -1.0 : 2.0)) * g->gp.cell;
                }
                int tzr[2];
                for (int e = 0; e < 2; e++) {
                    const double t = floor((zr[e] - l->gp.o[2]) * l->gp.s);
                    const int cz = !(t > 0.0) ? 0 : (t >= (double)l->gp.dim[2] ? l->gp.dim[2] - 1 : (int)t);
                    tzr[e] = cz / l->tile[2];
                }
                const uint64_t per_slab = (uint64_t)A.ntile[0] * (uint64_t)A.ntile[1];
                const uint32_t lo_id = (uint32_t)std::min<uint64_t>((uint64_t)tzr[0] * per_slab, 0xffffffffull);
                const uint32_t hi_id = (uint32_t)std::min<uint64_t>((uint64_t)(tzr[1] + 1) * per_slab, 0xffffffffull);
                uint32_t *pos = A.flag_count + 56;  // (two words of the counter block)
                hipLaunchKernelGGL(k_list_bounds, dim3(1), dim3(64), 0, st, l->tile_list, (uint32_t)l->tile_count, lo_id, hi_id, pos);
                uint32_t *hb = reinterpret_cast<uint32_t *>(ctx->pinned);
                PC_HIP(hipMemcpyAsync(hb, pos, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
                PC_HIP(hipStreamSynchronize(st));
                if (hb[1] >= hb[0] && hb[1] <= (uint32_t)l->tile_count) {
                    A.tile_list = l->tile_list + hb[0];
                    nt = (int64_t)hb[1] - (int64_t)hb[0];
                }
                if (nt <= 0) continue;
            }
            if (i == 0 && !A.qrec && !A.tile_list && (A.s_begin > 0 || A.s_end < l->n) && A.s_end > A.s_begin) {
                // A query shard: its sorted positions are whole z layers of cells (more or less), i.e. one run of
                // tile ids -- only those are launched.  (Launching all tiles and letting the others leave at once
                // costs more than their launch: consecutive tiles share an XCD, so a shard's tiles would all sit on
                // one or two of the eight XCDs while the rest of the GPU watches.  10 M points, one slice of eight:
                // 1.62 ms against 1.77 ms for the whole cloud.)
                const size_t rsz = g->dtype == PC_F32 ? sizeof(Rec32) : sizeof(Rec64), zoff = g->dtype == PC_F32 ? 8 : 16;
                char *hp = reinterpret_cast<char *>(ctx->pinned);
                const char *rb = reinterpret_cast<const char *>(l->rec);
                PC_HIP(hipMemcpyAsync(hp, rb + rsz * (size_t)A.s_begin + zoff, 8, hipMemcpyDeviceToHost, st));
                PC_HIP(hipMemcpyAsync(hp + 8, rb + rsz * (size_t)(A.s_end - 1) + zoff, 8, hipMemcpyDeviceToHost, st));
                PC_HIP(hipStreamSynchronize(st));
                int tz[2];
                for (int e = 0; e < 2; e++) {
                    const double z = g->dtype == PC_F32 ? (double)*reinterpret_cast<const float *>(hp + 8 * e)
                                                        : *reinterpret_cast<const double *>(hp + 8 * e);
                    const double t = floor((z - l->gp.o[2]) * l->gp.s);  // (the expression behind the cell keys)
                    const int c = !(t > 0.0) ? 0 : (t >= (double)l->gp.dim[2] ? l->gp.dim[2] - 1 : (int)t);
                    tz[e] = c / l->tile[2];
                }
                if (tz[1] >= tz[0]) {
                    A.tile_base = (int64_t)tz[0] * A.ntile[0] * A.ntile[1];
                    nt = (int64_t)(tz[1] - tz[0] + 1) * A.ntile[0] * A.ntile[1];
                }
            }
            const int nw = l->uneven ? uneven_waves(ctx, nt) : std::max(1, l->tile_waves);
            if (knobs.debug)
                fprintf(stderr, "[pc_amd]   tile pass %d: tile %dx%dx%d cap %d waves %d tiles %lld\n", i, l->tile[0], l->tile[1],
                        l->tile[2], l->tile_cap, nw, (long long)nt);
            const size_t lds = sizeof(float) * 4 * ((size_t)A.cand_cap + 2) + sizeof(uint32_t) * (3 * MAX_ROWS + 4) +
                               (A.halves ? 2 : 1) * sizeof(uint16_t) * MAX_ROWS * (MAX_NX + 1) +
                               nw * std::max(sizeof(uint32_t) * PC_WAVE * (SEL_SLOTS + 1), sizeof(uint16_t) * PC_WAVE * (size_t)A.k);
            if (g->dtype == PC_F32) {
                if (l->uneven)
                    pc_tile_launch_f32_split(ks, mode, (unsigned)nt, nw, lds, st, A);
                else
                    pc_tile_launch_f32(ks, mode, (unsigned)nt, nw, lds, st, A);
            } else {
                if (l->uneven)
                    pc_tile_launch_f64_split(ks, mode, (unsigned)nt, nw, lds, st, A);
                else
                    pc_tile_launch_f64(ks, mode, (unsigned)nt, nw, lds, st, A);
            }
            PC_HIP(hipGetLastError());
        }
        if (qflag) {
            // how many points are still open decides whether another (coarser) pass is worth it
            // (counted by listing the open queries in the NEXT level's order where that level may want the
            // list: one read-back per level either way)
            int64_t open = 0;
            bool counted = false;
            if (i + 1 == nlv && nlv < MAX_LEVELS && g->uneven && lazy) {
                // The last level there is, of a big cloud that was given its finest level only: the flags say which
                // level is worth making next.  A postponed query names the level that is to list it; a level whose
                // list would hold under a sixteenth of the queries still waiting for some level is left out (its
                // queries ride on to the next one), and queries amid more candidates than a pass stages wait for no
                // level at all.  (66 M voxels of planes + clutter: levels 1 and 3 would answer 0.15 M and 0.04 M
                // queries of 29.6 M and 2.2 M open ones -- 8 ms of passes and 9 ms of sorting the cloud twice.)
                uint32_t *hc = A.flag_count + 32;  // (words 32..47 of the counter block)
                PC_HIP(hipMemsetAsync(hc, 0, 16 * sizeof(uint32_t), st));
                hipLaunchKernelGGL(k_flag_hist, dim3((unsigned)((npt + 4095) / 4096)), dim3(256), 0, st, qflag, (int64_t)npt, hc);
                uint32_t *hh = reinterpret_cast<uint32_t *>(ctx->pinned);
                PC_HIP(hipMemcpyAsync(hh, hc, 16 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
                PC_HIP(hipStreamSynchronize(st));
                int64_t waiting = 0;
                for (int f = 1; f < 15; f++) waiting += hh[f];
                open = waiting + hh[15];
                counted = true;
                // (making a level sorts the whole cloud: not for fewer than a 16th of it -- 2 M points in Gaussian
                // clusters, build + search: 7.8 / 6.6 / 6.6 / 8.8 ms at an 8th / 16th / 32nd / 64th; spheres and planes alike)
                if (waiting >= std::max<int64_t>(std::max<int64_t>(2048, A.nq / 200), l->n / (knobs.lazy_div > 0 ? knobs.lazy_div : 16))) {
                    int v = l->level + 1;
                    for (; v < MAX_LEVELS; v++) {
                        int64_t listed = 0;
                        for (int f = 1; f <= std::min(v + 1, 14); f++) listed += hh[f];
                        if (listed >= std::max<int64_t>(2048, waiting / 16)) break;
                    }
                    const pc_grid *more = nullptr;
                    if (v < MAX_LEVELS) PC_TRY(pc_grid_level(ctx, lv[i], v, st, &more));
                    if (more) lv[nlv++] = more;
                    if (knobs.debug)
                        fprintf(stderr, "[pc_amd]   %lld waiting for a level, %u crowded: level %d %s\n", (long long)waiting, hh[15], v,
                                more ? "made" : "not made");
                }
            } else if (i + 1 == nlv && nlv < MAX_LEVELS && g->uneven) {
                // the last level there is: the next one is made only if enough queries are still open for the
                // ladder to go on (one more count read back, at this level only)
                PC_TRY(pc_compact_flags(ctx, qflag, (int64_t)npt, nullptr, &open, st));
                counted = true;
                // (Making a level sorts all n points, some 0.35 ns each, and the wave-per-query kernel answers an open
                // query in some 20 ns; besides, what is still open after four levels is often open for want of FINER
                // cells, which no coarser level helps: 2 M points in Gaussian clusters leave 161 k open after level 3
                // and 136 k after level 7.  A further level is made while an eighth of the points is open -- measured
                // at an eighth / 16th / 32nd: clusters 4.4 / 5.6 / 5.8 ms, LiDAR frame 1.13 / 1.08 / 1.46 ms, 4 M points
                // of planes + clutter 14.2 / 15.3 / 15.4 ms (28.9 ms with four levels and no more).)
                if (open >= std::max<int64_t>(std::max<int64_t>(2048, A.nq / 200),
                                              nlv >= EAGER_LEVELS ? l->n / (knobs.lazy_div > 0 ? knobs.lazy_div : 8) : 0)) {
                    const pc_grid *more = nullptr;
                    PC_TRY(pc_grid_coarser(ctx, lv[i], st, &more));
                    if (more) lv[nlv++] = more;
                }
            }
            // (external queries are sorted into the next level's cells before they can be listed: not before a count has
            // shown that the ladder goes on -- an ICP iteration whose queries are all answered at the finest level binned
            // them a second time for nothing, a sixth of the iteration)
            if (qrec0 && !counted && i + 1 < nlv) {
                PC_TRY(pc_compact_flags(ctx, qflag, (int64_t)npt, nullptr, &open, st));
                counted = true;
                if (open < std::max<int64_t>(2048, A.nq / 200)) {
                    open_prev = open;
                    break;
                }
            }
            if (batch_passes && ks <= batch_max_ks && i + 1 < nlv) {
                const pc_grid *nl = lv[i + 1];
                if (qrec0) {
                    PC_TRY(pc_bin_queries(ctx, nl, A.qraw, A.nq, qrec_l.p, qkeys_l.as<uint32_t>(), nullptr, st));
                    binned_level = i + 1;
                }
                PC_TRY(pc_batch_open_list(ctx, g->dtype, qrec0 ? qrec_l.p : nl->rec, qrec0 ? A.nq : nl->n, qflag,
                                          hinted ? nl->level + 1 : 255, open_list, &list_count, &open, open_counter, st));
                list_level = i + 1;  // (listed: the queries the next level is to try; open: those and the ones that wait for a later one)
            } else if (!counted) {
                PC_TRY(pc_compact_flags(ctx, qflag, (int64_t)npt, nullptr, &open, st));
            }
            if (knobs.debug)
                fprintf(stderr, "[pc_amd]   level %d: cell %.4g dims %dx%dx%d tile %dx%dx%d cap %d tiles %lld sparse %d -> open %lld\n",
                        i, l->gp.cell, l->gp.dim[0], l->gp.dim[1], l->gp.dim[2], l->tile[0], l->tile[1], l->tile[2],
                        l->tile_cap, (long long)nt, l->keys != nullptr, (long long)open);
            // (No level is skipped after a pass that closed next to nothing: the next one may be the only
            // level whose cells suit the open points -- coarser ones hold more candidates than a tile stages.
            // Measured on 66 M voxels of planes + clutter: skipping sent 30 M queries to the wave-per-query
            // kernel, 658 ms against 352 ms.)
            open_prev = open;
            if (open < std::max<int64_t>(2048, A.nq / 200)) break;
        }
    }
    // Points still open after the ladder are either isolated (nothing helps but the wave-per-query
    // kernel) or sit amid more candidates than any level above could stage: for those, up to three grids
    // of half the cell edge each, made on first need and kept with the index.  A pass costs little for
    // the isolated ones (their cells are nearly empty); going finer stops when it closes under 5 %.
    uint32_t crowded = 0;
    if (qflag && batch_passes && !knobs.no_finer && open_prev >= std::max<int64_t>(2048, A.nq / 200)) {
        uint32_t *h = reinterpret_cast<uint32_t *>(ctx->pinned);
        PC_HIP(hipMemcpyAsync(h, A.flag_count + 26, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        PC_HIP(hipStreamSynchronize(st));
        crowded = h[0];  // queries the FIRST pass found amid more candidates than a tile stages
        if (knobs.debug) fprintf(stderr, "[pc_amd]   %lld open after the ladder, %u crowded in the first pass\n", (long long)open_prev, crowded);
    }
    if (crowded >= std::max<int64_t>(2048, A.nq / 100)) {
        const pc_grid *fine = g;
        for (int f = 0; f < MAX_FINER; f++) {
            const pc_grid *nf = nullptr;
            PC_TRY(pc_grid_finer(ctx, fine, st, &nf));
            if (!nf) break;
            fine = nf;
            if (qrec0) {
                PC_TRY(pc_bin_queries(ctx, fine, A.qraw, A.nq, qrec_l.p, qkeys_l.as<uint32_t>(), nullptr, st));
                A.qrec = qrec_l.p;
                A.qci = CellIndex{nullptr, qkeys_l.as<uint32_t>(), (uint32_t)A.nq, nullptr, 0u};
            }
            A.gp = fine->gp;
            A.rec = fine->rec;
            A.ci = fine->index();
            A.cand_cap = uneven_cap(ctx, g);
            A.qflag_in = qflag;
            A.qflag_out = qflag;
            A.flag_max = 255;
            A.flag_top = 0;
            int64_t nopen = 0, open = 0;
            PC_TRY(pc_batch_open_list(ctx, g->dtype, qrec0 ? qrec_l.p : fine->rec, qrec0 ? A.nq : fine->n, qflag, 255, open_list, &nopen,
                                      &open_all, open_counter, st));
            PC_TRY(pc_launch_batch_scan(ctx, g->dtype, ks, mode, A, open_list, nopen, st));
            PC_TRY(pc_compact_flags(ctx, qflag, (int64_t)npt, nullptr, &open, st));
            if (knobs.debug)
                fprintf(stderr, "[pc_amd]   finer level: cell %.4g, %lld open queries -> %lld\n", fine->gp.cell, (long long)nopen,
                        (long long)open);
            const bool progress = nopen - open >= nopen / 20;
            open_prev = open;
            if (!progress || open < std::max<int64_t>(2048, A.nq / 200)) break;
        }
    }
    // What the ladder leaves open in numbers is next to something dense: too few candidates on the fine levels,
    // too many for a tile on the coarse ones.  Those go, lane per query, through rings of the finest level cut to
    // the ball of their bound (pc_knn_ring.hip) before a whole wave is spent on each of the rest.  (66 M voxels of
    // planes + clutter, 2.19 M open: 44.6 ms of wave-per-query work.)
    if (qflag && !qrec0 && A.radius2 < 0.0 && ks <= 34 && batch_passes && !knobs.no_ring && open_prev >= knobs.ring_min) {
        A.gp = g->gp;
        A.rec = g->rec;
        A.ci = g->index();
        A.qrec = nullptr;
        A.qflag_in = qflag;
        A.qflag_out = qflag;
        int64_t nopen = 0, all = 0;
        PC_TRY(pc_batch_open_list(ctx, g->dtype, g->rec, g->n, qflag, 255, open_list, &nopen, &all, open_counter, st));
        // Whether the pass pays depends on how many of the queries its rings reach, which no count of open queries
        // tells.  A query whose bound needs more cells than the rings have is lost to the pass for sure (it leaves at
        // once, but its wave's other lanes walk on and the whole list was made for nothing): the pass runs where those
        // are under half of the list.  (4 M voxels of planes + clutter: 72 % of 204 k queries, the pass answered a tenth
        // and cost 2 ms of a 19 ms search; 66 M voxels: none of 2.3 M, 85 % of them without any bound, 99 % answered.)
        uint32_t *answered = A.flag_count + 30, *elig = A.flag_count + 58;
        uint32_t *h = reinterpret_cast<uint32_t *>(ctx->pinned);
        PC_TRY(pc_ring_eligible(ctx, g->dtype, A, open_list, nopen, elig, st));
        PC_HIP(hipMemcpyAsync(h, elig, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        PC_HIP(hipStreamSynchronize(st));
        const int64_t fits = h[0], boundless = h[1];
        if (knobs.debug)
            fprintf(stderr, "[pc_amd]   ring pass: of %lld open queries %lld with a bound that fits, %lld without a bound\n", (long long)nopen,
                    (long long)fits, (long long)boundless);
        if (2 * (fits + boundless) >= nopen || knobs.ring_min <= 1)
            PC_TRY(pc_launch_ring_scan(ctx, g->dtype, ks, mode, A, open_list, nopen, 1, answered, st));
        if (knobs.debug) {
            PC_HIP(hipMemcpyAsync(h, answered, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            PC_HIP(hipStreamSynchronize(st));
            fprintf(stderr, "[pc_amd]   ring pass: %lld open queries, %u answered\n", (long long)nopen, h[0]);
            PC_HIP(hipMemcpyAsync(h, A.flag_count + 48, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            PC_HIP(hipStreamSynchronize(st));
            if (h[7])
                fprintf(stderr, "[pc_amd]   ring pass per wave: %.1f iterations, %.1f insertion rounds; per query: %.1f picks, %.1f rows, %.1f candidates\n",
                        (double)h[0] / h[7], (double)h[4] / h[7], (double)h[1] / nopen, (double)h[2] / nopen, (double)h[3] / nopen);
        }
    }
    if (qflag) {
        // back to level 0: the re-run lists hold level-0 sorted positions
        A.gp = g->gp;
        A.rec = g->rec;
        A.ci = g->index();
        A.qrec = qrec0;
        A.qci = qci0;
        const unsigned nb = (unsigned)((A.nq + 255) / 256);
        const void *qsrc = A.qrec ? A.qrec : g->rec;  // (external queries: positions in the level-0 bins)
        if (!A.qrec && A.out_slot && A.compact_base < 0)
            hipLaunchKernelGGL(k_collect_flagged_by_id, dim3((unsigned)((npt + 256 * COLLECT_PER_THREAD - 1) / (256 * COLLECT_PER_THREAD))), dim3(256), 0, st, qflag, A.qub_arr,
                               A.out_slot, (int64_t)npt, A.s_begin, A.flag_list, A.flag_ub, A.flag_count, A.flag_cap);
        else if (g->dtype == PC_F32)
            hipLaunchKernelGGL(k_collect_flagged<Rec32>, dim3(nb), dim3(256), 0, st, (const Rec32 *)qsrc, A.s_begin,
                               A.s_end, qflag, A.qub_arr, A.flag_list, A.flag_ub, A.flag_count, A.flag_cap);
        else
            hipLaunchKernelGGL(k_collect_flagged<Rec64>, dim3(nb), dim3(256), 0, st, (const Rec64 *)qsrc, A.s_begin,
                               A.s_end, qflag, A.qub_arr, A.flag_list, A.flag_ub, A.flag_count, A.flag_cap);
        PC_HIP(hipGetLastError());
    }
    // second stage: one wave per uncertified query (device-side count, no host round trip) ...
    A.qlist = A.flag_list;
    A.qub = A.flag_ub;
    A.qcount = A.flag_count;
    // (an index of one level: whoever is listed without a bound has had its 27 cells searched in vain by the tile
    // pass -- all but the few queries of over-full tiles, which lose nothing by the wider first block)
    A.first_ring = nlv == 1 && g->finer == nullptr && !g->uneven ? 2 : 1;
    PC_TRY(pc_launch_wave_query(ctx, g, mode, A, st));
    // ... and whatever that stage could not hold is left for the per-lane scan of the caller
    A.qlist = A.flag2_list;
    A.qub = nullptr;
    A.qcount = A.flag2_count;
    *handled = true;
    return PC_OK;
}
