// Tile-cooperative exact kNN (self-queries, k <= 32; fp32 and fp64 grids).
//
// One wave owns one tile of cells.  The tile's candidate block (the tile plus a one-cell halo;
// its (y,z) rows are contiguous point ranges) is staged once into LDS; each lane then answers
// one query of the tile and every lane walks the SAME candidate sequence, so a candidate is one
// broadcast ds_read_b128.  Selection is two-stage:
//   1. fp32 filter: key = (fp32 d2 bits with the low ORD_BITS cleared) | candidate ordinal.
//      Keys are unique u32, so a sorted list of KS keys per lane is maintained in registers
//      with one v_med3_u32 per slot per candidate:  L[s] = med3(L[s-1], key, L[s]).
//   2. exact finish: the KS survivors are re-measured in fp64 in the reference's operation
//      order, ranked by (d2, original index), and the result is accepted only if at least k of
//      them lie strictly below a bound that no unlisted candidate and no unscanned point can
//      beat.  Queries that cannot be certified (sparse surroundings, heavy ties, over-full
//      tiles) are appended to a list that the one-query-per-lane fp64 scan answers afterwards.
// The consumers (mean distance, normal, ...) run on the certified list while the candidates
// are still in LDS.  HBM traffic per query is therefore the algorithmic minimum plus the halo
// re-reads served by L2.
#include <stdio.h>

#include <vector>

#include "pc_knn_dev.h"

namespace {

constexpr int ORD_BITS = 11;
constexpr int CAND_CAP_MAX = 1 << ORD_BITS;  // most candidates one tile may stage (32 KiB of LDS)
constexpr uint32_t ORD_MASK = CAND_CAP_MAX - 1;
constexpr uint32_t KEY_EMPTY = 0xFFFFFFFFu;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) float lds_f32;    // 32-bit LDS addresses: no 64-bit pointer math
typedef __attribute__((address_space(3))) uint32_t lds_u32;
constexpr int MAX_ROWS = 64;  // (tile_y + 2) * (tile_z + 2) must fit one wave
constexpr int MAX_NX = 7;     // tile_x + 2 cells per block row at most
constexpr int MAX_TILE_WAVES = 4;  // waves of one workgroup, all working on the same staged block
constexpr int UNEVEN_CAP = 1024;   // candidate capacity of the tiles of multi-level indexes
constexpr int SEL_SLOTS = 16; // parked keys per lane between two sorted insertions

// LDS reads whose pairing and issue point the compiler must not rearrange.  The values are valid only
// after lds_wait(), which is tied to them through its operands.
__device__ inline void lds_read_xyz_pair(uint32_t addr, f32x2 &x, f32x2 &y, f32x2 &z) {
    asm volatile(
        "ds_read2_b32 %0, %3 offset1:3\n\t"
        "ds_read2_b32 %1, %3 offset0:1 offset1:4\n\t"
        "ds_read2_b32 %2, %3 offset0:2 offset1:5"
        : "=&v"(x), "=&v"(y), "=&v"(z)
        : "v"(addr));
}
__device__ inline void lds_wait(f32x2 &x, f32x2 &y, f32x2 &z) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x), "+v"(y), "+v"(z));
}
__device__ inline uint32_t lds_read_u32(uint32_t addr) {
    uint32_t v;
    asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
__device__ inline void lds_write_u32(uint32_t addr, uint32_t v) { asm volatile("ds_write_b32 %0, %1" : : "v"(addr), "v"(v)); }
__device__ inline void lds_wait(uint32_t &v) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v)); }

template <int KS>
__device__ inline void list_insert(uint32_t (&L)[KS], uint32_t key) {
#pragma unroll
    for (int s = KS - 1; s > 0; s--) asm("v_med3_u32 %0, %1, %2, %0" : "+v"(L[s]) : "v"(L[s - 1]), "v"(key));
    asm("v_min_u32 %0, %0, %1" : "+v"(L[0]) : "v"(key));
}

// How the two record types appear in the LDS candidate block.
//  fp32 grids: the record itself {x, y, z, original index}: fp32 differences of the stored values
//              carry only a relative rounding error.
//  fp64 grids: {x, y, z relative to the block origin, rounded to fp32; sorted position}: the filter
//              works on rounded coordinates, so its keys carry an ABSOLUTE error of at most
//              ~2^-21 * block extent on the distance, which the certification bound subtracts; the
//              exact finish re-reads the fp64 record by its sorted position.
template <typename REC>
struct TileRec;
template <>
struct TileRec<Rec32> {
    static constexpr bool kRelative = false;
    __device__ static inline float4 stage(const Rec32 *rec, uint32_t src, const double *) {
        return *reinterpret_cast<const float4 *>(rec + src);
    }
    __device__ static inline void exact(const Rec32 *, const float4 &c, double &x, double &y, double &z, int32_t &id) {
        x = (double)c.x;
        y = (double)c.y;
        z = (double)c.z;
        id = __float_as_int(c.w);
    }
};
template <>
struct TileRec<Rec64> {
    static constexpr bool kRelative = true;
    __device__ static inline float4 stage(const Rec64 *rec, uint32_t src, const double *org) {
        double x, y, z;
        int32_t id;
        RecTraits<Rec64>::load(rec + src, x, y, z, id);
        return make_float4((float)(x - org[0]), (float)(y - org[1]), (float)(z - org[2]), __uint_as_float(src));
    }
    __device__ static inline void exact(const Rec64 *rec, const float4 &c, double &x, double &y, double &z,
                                        int32_t &id) {
        RecTraits<Rec64>::load(rec + __float_as_uint(c.w), x, y, z, id);
    }
};

// One tile (first cell c0, extent ext cells) answered by the whole workgroup.  Returns 1, having done
// nothing, if the tile's block holds more candidates than LDS and may_split allows the caller to come
// back with smaller tiles; 0 otherwise.  Every return is uniform over the workgroup.
template <typename REC, int KS, int MODE>
__device__ __forceinline__ int tile_body(const ScanArgs &A, char *smem, const int (&c0)[3], const int (&ext)[3],
                                         const bool may_split) {
    __syncthreads();  // a previous call of this workgroup is done with LDS
    // candidate block: xyz packed at a 12-byte stride (+ two pad slots) and the 4th word apart.  The
    // pair (j, j+1) of one coordinate is then ONE ds_read2_b32 at immediate offsets (0,3), (1,4),
    // (2,5) from a single address and lands as a register pair for packed fp32 math.
    const int cstride = A.cand_cap + 2;
    float *cxyz = reinterpret_cast<float *>(smem), *cws = cxyz + 3 * cstride;
    uint32_t *row_src = reinterpret_cast<uint32_t *>(cws + cstride);                               // [64] begin
    uint32_t *row_off = row_src + MAX_ROWS;                                                        // [65] offsets
    uint16_t *cell_off = reinterpret_cast<uint16_t *>(row_off + MAX_ROWS + 4);                     // [64][MAX_NX+1]
    // one area of lane-private columns: parked keys during the filter, result order afterwards
    const int tid = threadIdx.x, lane = tid & (PC_WAVE - 1), wid = tid >> 6, nthr = blockDim.x;
    // (one such area per wave)
    uint32_t *selbuf = reinterpret_cast<uint32_t *>(cell_off + MAX_ROWS * (MAX_NX + 1)) +
                       wid * max(PC_WAVE * (SEL_SLOTS + 1), (PC_WAVE / 2) * A.k);           // [SEL_SLOTS + 1][64]
    uint16_t *ordbuf = reinterpret_cast<uint16_t *>(selbuf);                                       // [k][64]
    const GridParams &gp = A.gp;
    const REC *rec = reinterpret_cast<const REC *>(A.rec);
    // where the queries come from: the grid's own records, or external queries binned into the grid
    const REC *qsrc = A.qrec ? reinterpret_cast<const REC *>(A.qrec) : rec;
    const CellIndex &qix = A.qrec ? A.qci : A.ci;
    const int k = A.k;

    int c1[3], lo[3], hi[3];                                              // last cell; block bounds
#pragma unroll
    for (int a = 0; a < 3; a++) {
        c1[a] = min(c0[a] + ext[a], gp.dim[a]) - 1;
        lo[a] = max(c0[a] - 1, 0);
        hi[a] = min(c1[a] + 1, gp.dim[a] - 1);
    }
    const int nry = hi[1] - lo[1] + 1, nrz = hi[2] - lo[2] + 1;
    const int nrows = nry * nrz;  // <= MAX_ROWS by construction of the tile shape
    // block origin and extent (world units): the frame of the fp32 filter of fp64 grids
    const double org[3] = {gp.o[0] + (double)lo[0] * gp.cell, gp.o[1] + (double)lo[1] * gp.cell,
                           gp.o[2] + (double)lo[2] * gp.cell};
    const double bext = gp.cell * (double)max(max(hi[0] - lo[0], hi[1] - lo[1]), hi[2] - lo[2]) + gp.cell;

    // ---- row table: lane r describes block row r -------------------------------------------------
    uint32_t row_begin = 0, row_cnt = 0, q_begin = 0, q_cnt = 0;
    if (lane < nrows) {
        const int y = lo[1] + lane % nry, z = lo[2] + lane / nry;
        const int64_t row = ((int64_t)z * gp.dim[1] + y) * gp.dim[0];
        row_begin = cell_begin(A.ci, row + lo[0]);
        row_cnt = cell_begin(A.ci, row + hi[0] + 1) - row_begin;
        if (y >= c0[1] && y <= c1[1] && z >= c0[2] && z <= c1[2]) {  // a row of the tile itself
            q_begin = cell_begin(qix, row + c0[0]);
            q_cnt = cell_begin(qix, row + c1[0] + 1) - q_begin;
            if (!A.qflag_in && !A.qrec) {  // first pass: restrict to the caller's sorted-position range
                const int64_t b = max((int64_t)q_begin, A.s_begin), e = min((int64_t)q_begin + q_cnt, A.s_end);
                q_begin = (uint32_t)b;
                q_cnt = e > b ? (uint32_t)(e - b) : 0u;
            }
        }
    }
    // inclusive prefix sums over the rows (wave scan)
    uint32_t cand_off = row_cnt, q_off = q_cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t a = __shfl_up(cand_off, o, 64), b = __shfl_up(q_off, o, 64);
        if (lane >= o) {
            cand_off += a;
            q_off += b;
        }
    }
    const uint32_t ncand = __builtin_amdgcn_readfirstlane(__shfl(cand_off, 63, 64));
    const uint32_t nquery = __builtin_amdgcn_readfirstlane(__shfl(q_off, 63, 64));
    if (nquery == 0) return 0;  // uniform
    cand_off -= row_cnt;      // exclusive
    q_off -= q_cnt;
    const bool too_many = ncand > (uint32_t)A.cand_cap;
    const bool splittable = may_split && (ext[0] > 1 || ext[1] > 1 || ext[2] > 1);
    if (A.qflag_in && too_many && !splittable) return 0;  // flagged points of an over-full block stay flagged
    if (A.qflag_in) {
        // later passes answer only the points an earlier pass flagged: leave if the tile has none
        bool any = false;
        for (uint32_t qbase = 0; qbase < nquery; qbase += PC_WAVE) {
            const uint32_t qo = qbase + lane;
            uint32_t sp = 0;
            bool have = false;
            for (int r = 0; r < nrows; r++) {
                const uint32_t ob = __shfl(q_off, r, 64), oc = __shfl(q_cnt, r, 64), gb = __shfl(q_begin, r, 64);
                if (qo < nquery && qo >= ob && qo < ob + oc) {
                    sp = gb + (qo - ob);
                    have = true;
                }
            }
            if (have) {
                double x, y, z;
                int32_t id;
                RecTraits<REC>::load(qsrc + sp, x, y, z, id);
                any = any || A.qflag_in[id] != 0;
            }
        }
        if (!__any(any)) return 0;
    }
    if (too_many && splittable) return 1;

    // over-full blocks do not fit LDS; blocks with fewer than k points cannot certify anything unless
    // they are the whole grid: both go straight to the re-run list
    const bool whole_grid = lo[0] == 0 && lo[1] == 0 && lo[2] == 0 && hi[0] == gp.dim[0] - 1 &&
                            hi[1] == gp.dim[1] - 1 && hi[2] == gp.dim[2] - 1;
    const bool overfull = ncand > (uint32_t)A.cand_cap || (ncand < (uint32_t)k && !whole_grid && A.radius2 < 0.0);

    // ---- stage the candidate block into LDS ------------------------------------------------------
    // flattened over the block's candidates: lane i copies candidates i, i+64, ...; the row of a
    // candidate follows from the row offsets, and four 16-byte loads are in flight per lane
    row_src[lane] = row_begin;
    row_off[lane] = cand_off;
    if (lane == 0) row_off[MAX_ROWS] = ncand;
    __syncthreads();
    if (!overfull) {
        int r = 0;
        for (uint32_t base = 0; base < ncand; base += 4 * nthr) {
            uint32_t src[4];
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const uint32_t i = base + b * nthr + tid;
                src[b] = 0;
                if (i < ncand) {
                    while (r + 1 < nrows && i >= row_off[r + 1]) r++;
                    src[b] = row_src[r] + (i - row_off[r]);
                }
            }
            float4 v[4];
#pragma unroll
            for (int b = 0; b < 4; b++) v[b] = TileRec<REC>::stage(rec, src[b], org);
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const uint32_t i = base + b * nthr + tid;
                if (i < ncand) {
                    cxyz[3 * i] = v[b].x;
                    cxyz[3 * i + 1] = v[b].y;
                    cxyz[3 * i + 2] = v[b].z;
                    cws[i] = v[b].w;
                }
            }
        }
    }
    // block-relative offset of every cell boundary of every block row
    const int nx = hi[0] - lo[0] + 1;
    for (int e = tid; e < nrows * (nx + 1); e += nthr) {
        const int r = e / (nx + 1), i = e - r * (nx + 1);
        const int y = lo[1] + r % nry, z = lo[2] + r / nry;
        const int64_t row = ((int64_t)z * gp.dim[1] + y) * gp.dim[0];
        cell_off[r * (MAX_NX + 1) + i] = (uint16_t)(row_off[r] + (cell_begin(A.ci, row + lo[0] + i) - row_src[r]));
    }
    __syncthreads();

    const double eps = 1e-7 * gp.cell;
    const double r2lim = A.radius2;

    for (uint32_t qbase = wid * PC_WAVE; qbase < nquery; qbase += nthr) {  // the waves share the tile's queries
        const uint32_t qo = qbase + lane;
        const bool active = qo < nquery;
        // which tile row holds query ordinal qo
        uint32_t spos = 0;
        for (int r = 0; r < nrows; r++) {  // shuffles stay outside divergent control flow
            const uint32_t ob = __shfl(q_off, r, 64), oc = __shfl(q_cnt, r, 64), gb = __shfl(q_begin, r, 64);
            if (active && qo >= ob && qo < ob + oc) spos = gb + (qo - ob);
        }
        double qxd = 0.0, qyd = 0.0, qzd = 0.0;
        int32_t qid = 0;
        bool live = active;  // a query this pass has to answer
        if (active) {
            RecTraits<REC>::load(qsrc + spos, qxd, qyd, qzd, qid);
            if (A.qflag_in) live = A.qflag_in[qid] != 0;
        }
        if (!__any(live)) continue;
        if (overfull) {
            if (live) {
                if (A.qflag_out) {
                    A.qflag_out[qid] = 1;
                    if (!A.qflag_in) A.qub_arr[qid] = INFINITY;
                } else {
                    const uint32_t p = atomicAdd(A.flag_count, 1u);
                    if (p < A.flag_cap) {
                        A.flag_list[p] = (int32_t)spos;
                        A.flag_ub[p] = A.qflag_in ? A.qub_arr[qid] : INFINITY;
                    }
                }
            }
            continue;
        }
        // filter coordinates: the stored fp32 values, or fp32 offsets from the block origin
        const float qx = TileRec<REC>::kRelative ? (float)(qxd - org[0]) : (float)qxd;
        const float qy = TileRec<REC>::kRelative ? (float)(qyd - org[1]) : (float)qyd;
        const float qz = TileRec<REC>::kRelative ? (float)(qzd - org[2]) : (float)qzd;

        uint32_t L[KS];
#pragma unroll
        for (int s = 0; s < KS; s++) L[s] = KEY_EMPTY;

        // ---- stage 1: fp32 filter over the lane's own 3x3x3 cells --------------------------------
        // Lanes of one cell walk the same LDS range (broadcast reads); the nine (dy, dz) row
        // segments are visited in lock-step, each for the longest segment any lane has.
        // A key can enter the list only if it is below the list's current last entry.  The lane
        // keeps a copy `thr` of that entry and parks the keys below it in its own column of selbuf;
        // the (expensive) sorted insertion runs over the parked keys only, when some lane's column
        // is nearly full and once more at the end.  `thr` lags behind the list, which only lets a
        // few keys too many through: the final list is the same KS smallest keys either way.
        uint32_t thr = KEY_EMPTY;
        lds_u32 *const col = (lds_u32 *)selbuf + lane;  // slot s of this lane: col[s * 64]
        const uint32_t col_a = (uint32_t)(uintptr_t)col;                   // LDS byte address of the column
        uint32_t wp = col_a - 4u * PC_WAVE;                                // last written slot (byte address)
        auto drain = [&]() {
            const uint32_t have = (wp + 4u * PC_WAVE - col_a) >> 8;
            uint32_t ra = (uint32_t)(uintptr_t)col;
            uint32_t nxt = lds_read_u32(ra);
            uint32_t s = 0;
            do {  // at least one round: a round of empty keys changes nothing
                lds_wait(nxt);
                const uint32_t key = s < have ? nxt : KEY_EMPTY;
                ra += 4 * PC_WAVE;
                nxt = lds_read_u32(ra);  // one slot ahead (the area has a spare row)
                list_insert<KS>(L, key);
                s++;
            } while (__any(s < have));
            lds_wait(nxt);
            thr = L[KS - 1];
            wp = col_a - 4u * PC_WAVE;
        };
        int qc[3];
        {
            const double qf[3] = {qxd, qyd, qzd};
#pragma unroll
            for (int a = 0; a < 3; a++) {
                const double tt = floor((qf[a] - gp.o[a]) * gp.s);
                qc[a] = !(tt > 0.0) ? 0 : (tt >= (double)gp.dim[a] ? gp.dim[a] - 1 : (int)tt);
            }
        }
        const int i0 = max(qc[0] - 1, 0) - lo[0], i1 = min(qc[0] + 1, gp.dim[0] - 1) - lo[0] + 1;
        // the nine (start, length) pairs and the wave-wide longest length of each, up front:
        // the nine max-reductions overlap instead of serialising in front of every segment.
        // Order: the query's own row, the four rows sharing a face with it, the four diagonal ones,
        // so that the threshold tightens early.
        constexpr int SEG_DY[9] = {0, -1, 1, 0, 0, -1, 1, -1, 1};
        constexpr int SEG_DZ[9] = {0, 0, 0, -1, 1, -1, -1, 1, 1};
        uint32_t seg_pack[9];  // start | len << 16
        int seg_max[9];
#pragma unroll
        for (int seg = 0; seg < 9; seg++) {
            const int y = qc[1] + SEG_DY[seg], z = qc[2] + SEG_DZ[seg];
            uint32_t st = 0, ln = 0;
            if (live && y >= lo[1] && y <= hi[1] && z >= lo[2] && z <= hi[2]) {
                const int r = (y - lo[1]) + nry * (z - lo[2]);
                st = cell_off[r * (MAX_NX + 1) + i0];
                ln = cell_off[r * (MAX_NX + 1) + i1] - st;
            }
            seg_pack[seg] = st | (ln << 16);
            seg_max[seg] = (int)ln;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
            for (int seg = 0; seg < 9; seg++) seg_max[seg] = max(seg_max[seg], __shfl_xor(seg_max[seg], o, 64));
        }
#pragma unroll
        for (int seg = 0; seg < 9; seg++) seg_max[seg] = __builtin_amdgcn_readfirstlane(seg_max[seg]);
        const f32x2 qx2 = {qx, qx}, qy2 = {qy, qy}, qz2 = {qz, qz};
        const uint32_t cand_a = (uint32_t)(uintptr_t)(const lds_f32 *)cxyz;  // LDS byte address
        const uint32_t pa_last = cand_a + 12u * (ncand - 1);
        const uint32_t wp_full = col_a + 4u * (SEL_SLOTS - 2) * PC_WAVE;  // (wp names the last written slot)
#pragma unroll
        for (int seg = 0; seg < 9; seg++) {
            const uint32_t start = seg_pack[seg] & 0xffffu;
            const uint32_t end = start + (seg_pack[seg] >> 16);
            const int tmax = seg_max[seg];
            // two candidates per step: one ds_read2_b32 per coordinate, packed fp32 math
            // (v_pk_add/mul/fma_f32); the next pair is requested as soon as this one is consumed
            uint32_t pa = min(cand_a + 12u * start, pa_last);
            f32x2 nx, ny, nz;
            lds_read_xyz_pair(pa, nx, ny, nz);
            uint32_t j0 = start, j1 = start + 1;
#pragma nounroll
            for (int t = 0; t < tmax; t += 2) {
                lds_wait(nx, ny, nz);
                const f32x2 dx = qx2 - nx, dy = qy2 - ny, dz = qz2 - nz;
                pa = min(pa + 24u, pa_last);
                lds_read_xyz_pair(pa, nx, ny, nz);
                f32x2 d = dx * dx;
                d = __builtin_elementwise_fma(dy, dy, d);
                d = __builtin_elementwise_fma(dz, dz, d);
                const uint32_t k0 = (__float_as_uint(d.x) & ~ORD_MASK) | j0;
                const uint32_t k1 = (__float_as_uint(d.y) & ~ORD_MASK) | j1;
                if (j0 < end && k0 < thr) {
                    wp += 4u * PC_WAVE;
                    lds_write_u32(wp, k0);
                }
                if (j1 < end && k1 < thr) {
                    wp += 4u * PC_WAVE;
                    lds_write_u32(wp, k1);
                }
                j0 += 2;
                j1 += 2;
                if (__any(wp >= wp_full)) drain();
            }
            lds_wait(nx, ny, nz);  // retire the last (unused) request
        }
        drain();
        if (!live) continue;

        // ---- stage 2: exact finish --------------------------------------------------------------------
        const double qv[3] = {qxd, qyd, qzd};
        bool whole = true;
        double dmin = INFINITY;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const int nlo = max(qc[a] - 1, 0), nhi = min(qc[a] + 1, gp.dim[a] - 1);
            if (nlo > 0) {
                whole = false;
                dmin = fmin(dmin, qv[a] - (gp.o[a] + (double)nlo * gp.cell));
            }
            if (nhi < gp.dim[a] - 1) {
                whole = false;
                dmin = fmin(dmin, (gp.o[a] + (double)(nhi + 1) * gp.cell) - qv[a]);
            }
        }
        double bound = INFINITY;  // no unscanned point is closer than sqrt(bound)
        if (!whole) {
            const double b = dmin - eps;
            bound = b > 0 ? b * b : 0.0;
        }
        // no unlisted candidate of the scanned cells is closer than sqrt(T * (1 - 2^-20))
        if (L[KS - 1] != KEY_EMPTY) {
            const double T = (double)__uint_as_float(L[KS - 1] & ~ORD_MASK);
            if (TileRec<REC>::kRelative) {
                // keys were measured on coordinates rounded in the block frame
                const double rt = sqrt(T) * (1.0 - 9.5367431640625e-07) - 4.76837158203125e-07 * bext;
                bound = fmin(bound, rt > 0.0 ? rt * rt : 0.0);
            } else {
                bound = fmin(bound, T * (1.0 - 9.5367431640625e-07));
            }
        }

        double dd[KS];
        uint32_t od[KS];
        int nsafe = 0, cnt = 0, nvalid = 0;
        double dmax = 0.0;  // farthest listed candidate: bounds the k-th distance when k are listed
#pragma unroll
        for (int s = 0; s < KS; s++) {
            dd[s] = INFINITY;
            od[s] = L[s] & ORD_MASK;
            if (L[s] != KEY_EMPTY) {
                double cx, cy, cz;
                int32_t cid;
                TileRec<REC>::exact(rec, make_float4(cxyz[3 * od[s]], cxyz[3 * od[s] + 1], cxyz[3 * od[s] + 2], cws[od[s]]), cx, cy, cz, cid);
                const double dx = qxd - cx, dy = qyd - cy, dz = qzd - cz;
                double d = dx * dx;
                d += dy * dy;
                d += dz * dz;
                nvalid++;
                dmax = fmax(dmax, d);
                if (d < bound) {
                    nsafe++;
                    if (!(r2lim >= 0.0 && !(d < r2lim))) {
                        dd[s] = d;
                        cnt++;
                    }
                }
            }
        }
        const bool ok = nsafe >= k || bound == INFINITY || (r2lim >= 0.0 && bound >= r2lim);
        if (!ok) {
            float ubf = nvalid >= k ? (float)(dmax * (1.0 + 1e-6)) : INFINITY;
            if (A.qflag_in) ubf = fminf(ubf, A.qub_arr[qid]);
            if (A.qflag_out) {
                A.qflag_out[qid] = 1;
                A.qub_arr[qid] = ubf;
            } else {
                const uint32_t p = atomicAdd(A.flag_count, 1u);
                if (p < A.flag_cap) {
                    A.flag_list[p] = (int32_t)spos;
                    A.flag_ub[p] = ubf;
                }
            }
            continue;
        }
        if (A.qflag_in && A.qflag_out) A.qflag_out[qid] = 0;  // answered: later passes skip it
        // The list is already ordered by the truncated fp32 key, so the exact (d2, index) order
        // differs from it by a few adjacent transpositions at most: bubble passes until no lane of
        // the wave swaps (excluded entries carry d2 = +inf and sink to the end).  Original indices
        // matter only between equal distances and are fetched then (keeps 18 registers free).
        auto orig_index = [&](uint32_t o) {
            double x, y, z;
            int32_t nid;
            TileRec<REC>::exact(rec, make_float4(0.f, 0.f, 0.f, cws[o]), x, y, z, nid);
            return nid;
        };
#pragma nounroll
        for (int pass = 0; pass < KS; pass++) {
            bool swapped = false;
#pragma unroll
            for (int s = 0; s + 1 < KS; s++) {
                const double da = dd[s], db = dd[s + 1];
                const uint32_t oa = od[s], ob = od[s + 1];
                const bool gt = da > db;
                dd[s] = fmin(da, db);
                dd[s + 1] = fmax(da, db);
                od[s] = gt ? ob : oa;
                od[s + 1] = gt ? oa : ob;
                swapped |= gt;
            }
            if (!__any(swapped)) break;
        }
        // equal distances (rare): order those neighbours by original index
        bool ties = false;
#pragma unroll
        for (int s = 0; s + 1 < KS; s++) ties |= dd[s] == dd[s + 1] && dd[s] != INFINITY;
        if (__any(ties)) {
#pragma nounroll
            for (int pass = 0; pass < KS; pass++) {
                bool swapped = false;
#pragma unroll
                for (int s = 0; s + 1 < KS; s++) {
                    const uint32_t oa = od[s], ob = od[s + 1];
                    bool gt = false;
                    if (dd[s] == dd[s + 1] && dd[s] != INFINITY) gt = orig_index(oa) > orig_index(ob);
                    od[s] = gt ? ob : oa;
                    od[s + 1] = gt ? oa : ob;
                    swapped |= gt;
                }
                if (!__any(swapped)) break;
            }
        }
        cnt = min(cnt, k);
        if (MODE == M_KNN) {
            // rows straight from the sorted registers (no second look at the candidates)
            const int64_t row = (int64_t)qid * k;
#pragma unroll
            for (int s = 0; s < KS; s++) {
                if (s < k) {  // wave-uniform
                    const bool have = s < cnt;
                    if (A.out_idx) A.out_idx[row + s] = have ? orig_index(od[s]) : -1;
                    if (A.out_d2) {
                        const double d = have ? dd[s] : INFINITY;
                        if (A.d2_f64)
                            reinterpret_cast<double *>(A.out_d2)[row + s] = d;
                        else
                            reinterpret_cast<float *>(A.out_d2)[row + s] = (float)d;
                    }
                }
            }
            if (A.out_cnt) A.out_cnt[qid] = cnt;
            continue;
        }
#pragma unroll
        for (int s = 0; s < KS; s++)
            if (s < k) ordbuf[s * PC_WAVE + lane] = (uint16_t)od[s];
        emit_result<MODE>(A, (int64_t)qid, cnt,
                          [&](int s, double &d, double &x, double &y, double &z, int32_t &nid) {
                              const uint32_t o = ordbuf[s * PC_WAVE + lane];
                              TileRec<REC>::exact(rec, make_float4(cxyz[3 * o], cxyz[3 * o + 1], cxyz[3 * o + 2], cws[o]), x, y, z, nid);
                              const double dx = qxd - x, dy = qyd - y, dz = qzd - z;
                              d = dx * dx;
                              d += dy * dy;
                              d += dz * dz;
                          });
    }
    return 0;
}

template <typename REC, int KS, int MODE, bool SPLIT>
__global__ __launch_bounds__(PC_WAVE * MAX_TILE_WAVES) void k_tile_scan(ScanArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // tile coordinates (x fastest so that neighbouring blocks share L2 lines)
    // Workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share an L2).  Remap so
    // that each XCD walks one contiguous eighth of the tile sequence: x/y-neighbouring tiles, whose
    // halos overlap, then hit the same L2 (placement is a speed matter only, never correctness).
    uint32_t bid = blockIdx.x;
    {
        const uint32_t nb = gridDim.x, q = nb >> 3, r = nb & 7u, xcd = bid & 7u;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    int64_t t = A.tile_list ? (int64_t)A.tile_list[bid] : (int64_t)bid;
    const int tx = (int)(t % A.ntile[0]);
    t /= A.ntile[0];
    const int ty = (int)(t % A.ntile[1]);
    const int tz = (int)(t / A.ntile[1]);
    const int c0[3] = {tx * A.tile[0], ty * A.tile[1], tz * A.tile[2]};  // first cell of the tile
    const int ext[3] = {A.tile[0], A.tile[1], A.tile[2]};
    if (!SPLIT) {  // (clouds that fill their box: the tile shape was sized for the LDS capacity)
        tile_body<REC, KS, MODE>(A, smem, c0, ext, false);
        return;
    }
    // The whole tile first; if its block is too full for LDS, its (up to) eight half-size parts
    // one after the other: big tiles fill the waves where the cloud is thin without giving up
    // the dense parts.
    const int half[3] = {(ext[0] + 1) / 2, (ext[1] + 1) / 2, (ext[2] + 1) / 2};
    int nsub = 0;
    for (int sub = -1; sub < nsub; sub++) {
        int sc0[3], sext[3];
        bool empty = false;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const int hi_part = sub < 0 ? 0 : (sub >> a) & 1;
            sc0[a] = c0[a] + hi_part * half[a];
            sext[a] = sub < 0 ? ext[a] : (hi_part ? ext[a] - half[a] : half[a]);
            empty |= sext[a] <= 0 || sc0[a] >= A.gp.dim[a];
        }
        if (empty) continue;
        const int again = tile_body<REC, KS, MODE>(A, smem, sc0, sext, sub < 0);
        if (sub < 0) {
            if (!again) break;
            nsub = 8;
        }
    }
}

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

// unique tile ids of the occupied cells, ascending (sparse grids only)
// Linear ids of the tiles (of g's tile shape) that hold at least one of n sorted cell keys.
int tile_list_from_keys(pc_ctx *ctx, const pc_grid *g, const uint32_t *keys, int64_t n, hipStream_t st, uint32_t **out,
                        int64_t *count) {
    *out = nullptr;
    *count = 0;
    const unsigned nb = (unsigned)((n + 255) / 256);
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

// chooses g->tile / g->tile_cap (once per grid)
int choose_tile(pc_ctx *ctx, const pc_grid *g, hipStream_t st) {
    if (g->tile[0] > 0) return PC_OK;
    const double occ = (double)g->n / (double)std::max<int64_t>(1, g->ncell);
    const bool dense = !g->uneven;
    int best[3] = {1, 1, 1};
    double expect;
    int waves = 1;
    if (dense) {
        // The cloud fills its box.  A workgroup of `waves` waves shares one staged block, so a
        // bigger tile pays less halo per query and less LDS per wave.  Take the fullest shape whose
        // expected population (plus 1.5 sigma) still fits the workgroup's lanes; among equals the one
        // with the smallest block.
        waves = MAX_TILE_WAVES;  // measured at 10 M, k = 16: 4 waves 2.78 ms, 2 waves 3.27 ms (LDS-bound occupancy)
        if (const char *e = getenv("PC_AMD_TILE_WAVES")) waves = std::min(MAX_TILE_WAVES, std::max(1, atoi(e)));
        double best_pop = -1.0, best_block = 0.0;
        for (int tz = 1; tz <= 6; tz++)
            for (int ty = 1; ty <= 6; ty++)
                for (int tx = 1; tx + 2 <= MAX_NX; tx++) {
                    if ((ty + 2) * (tz + 2) > MAX_ROWS) continue;
                    if (tx > std::max(1, (int)g->gp.dim[0]) || ty > std::max(1, (int)g->gp.dim[1]) ||
                        tz > std::max(1, (int)g->gp.dim[2]))
                        continue;
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
        // no more waves than the expected population needs
        waves = std::min(waves, std::max(1, (int)ceil((best_pop + 1.5 * sqrt(std::max(best_pop, 0.0))) / PC_WAVE)));
        expect = occ * (best[0] + 2) * (best[1] + 2) * (best[2] + 2);
        expect += 6.0 * sqrt(expect) + 8.0;  // Poisson mean + 6 sigma
    } else {
        // Unevenly filled box: every level of the index serves the points whose surroundings hold a
        // handful of points per cell AT THAT LEVEL (denser ones were answered by a finer level,
        // sparser ones wait for a coarser one).  Start from a 2x2x2 tile (a single cell on the coarser
        // levels, whose open points sit in fuller cells) and grow it while the occupied tiles hold
        // too few points to fill a wave (thin surfaces, scan lines); the shape is settled below.
        int base = g->level > 0 ? 1 : 2;
        if (const char *e = getenv(g->level > 0 ? "PC_AMD_UNEVEN_TILE1" : "PC_AMD_UNEVEN_TILE0")) base = std::max(1, atoi(e));
        for (int a = 0; a < 3; a++) best[a] = std::min(std::min(base, a == 0 ? MAX_NX - 2 : 6), std::max(1, (int)g->gp.dim[a]));
        expect = UNEVEN_CAP;
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

template <typename REC, int KS, bool SPLIT>
void launch_tile(int mode, unsigned nblocks, int nw, size_t lds, hipStream_t st, const ScanArgs &A) {
    const dim3 grid(nblocks), block(PC_WAVE * nw);
    switch (mode) {
        case M_KNN: hipLaunchKernelGGL((k_tile_scan<REC, KS, M_KNN, SPLIT>), grid, block, lds, st, A); break;
        case M_SOR: hipLaunchKernelGGL((k_tile_scan<REC, KS, M_SOR, SPLIT>), grid, block, lds, st, A); break;
        case M_NORMALS: hipLaunchKernelGGL((k_tile_scan<REC, KS, M_NORMALS, SPLIT>), grid, block, lds, st, A); break;
        case M_NNDIST: hipLaunchKernelGGL((k_tile_scan<REC, KS, M_NNDIST, SPLIT>), grid, block, lds, st, A); break;
        case M_ALL: hipLaunchKernelGGL((k_tile_scan<REC, KS, M_ALL, SPLIT>), grid, block, lds, st, A); break;
        default: break;
    }
}

template <typename REC, bool SPLIT>
void launch_tile_ks(int ks, int mode, unsigned nblocks, int nw, size_t lds, hipStream_t st, const ScanArgs &A) {
    switch (ks) {
        case 4: launch_tile<REC, 4, SPLIT>(mode, nblocks, nw, lds, st, A); break;
        case 10: launch_tile<REC, 10, SPLIT>(mode, nblocks, nw, lds, st, A); break;
        case 18: launch_tile<REC, 18, SPLIT>(mode, nblocks, nw, lds, st, A); break;
        case 22: launch_tile<REC, 22, SPLIT>(mode, nblocks, nw, lds, st, A); break;
        default: launch_tile<REC, 34, SPLIT>(mode, nblocks, nw, lds, st, A); break;
    }
}

}  // namespace

int pc_launch_tile_scan(pc_ctx *ctx, const pc_grid *g, int mode, ScanArgs &A, hipStream_t st, bool *handled) {
    *handled = false;
    if (A.queries != nullptr || A.k > 32 || mode == M_COUNT || A.nq <= 0) return PC_OK;
    if (getenv("PC_AMD_EXACT_SCAN_ONLY")) return PC_OK;  // A/B switch: one-query-per-lane fp64 scan only
    // the levels of the index, finest first; every level gets a tile pass, a coarser one answering
    // only what the finer ones flagged
    const pc_grid *lv[MAX_LEVELS];
    int nlv = 0;
    for (const pc_grid *l = g; l && nlv < MAX_LEVELS; l = l->coarse) lv[nlv++] = l;
    for (int i = 0; i < nlv; i++) {
        PC_TRY(choose_tile(ctx, lv[i], st));
        int64_t nt = 1;
        for (int a = 0; a < 3; a++) nt *= (lv[i]->gp.dim[a] + lv[i]->tile[a] - 1) / lv[i]->tile[a];
        if (lv[i]->tile_list) nt = lv[i]->tile_count;
        if (nt > 0x7fffffff) return PC_OK;
    }
    A.s_end = A.s_begin + A.nq;
    A.flag_cap = (uint32_t)A.nq;
    // [counts (256 B)] [list 1] [upper bounds 1] [list 2] [per-point bounds] [per-point flags]; the
    // block returns to the pool when this function ends, which is safe because everything that
    // touches it is ordered on `st`
    const size_t npt = A.qrec ? (size_t)A.nq : (size_t)g->n;
    Scratch flags(ctx, 3 * sizeof(int32_t) * (size_t)A.nq + 256 + (nlv > 1 ? 5 * npt + 64 : 0));
    if (!flags.ok()) return pc_fail(ctx, PC_NOMEM, "re-run lists");
    A.flag_count = flags.as<uint32_t>();
    A.flag2_count = A.flag_count + 16;
    A.flag_list = reinterpret_cast<int32_t *>(flags.as<char>() + 256);
    A.flag_ub = reinterpret_cast<float *>(A.flag_list + A.nq);
    A.flag2_list = A.flag_list + 2 * A.nq;
    PC_HIP(hipMemsetAsync(A.flag_count, 0, 256, st));
    uint8_t *qflag = nullptr;
    if (nlv > 1) {
        A.qub_arr = reinterpret_cast<float *>(A.flag2_list + A.nq);
        qflag = reinterpret_cast<uint8_t *>(A.qub_arr + npt);
        PC_HIP(hipMemsetAsync(qflag, 0, npt, st));
    }
    const int ks = A.k <= 2 ? 4 : (A.k <= 8 ? 10 : (A.k <= 16 ? 18 : (A.k <= 20 ? 22 : 34)));
    // external queries arrive binned into level 0; a coarser level bins them again into its own cells
    const void *qrec0 = A.qrec;
    const CellIndex qci0 = A.qci;
    const size_t qrec_bytes = (g->dtype == PC_F32 ? sizeof(Rec32) : sizeof(Rec64)) * (size_t)A.nq;
    Scratch qrec_l(ctx, A.qrec && nlv > 1 ? qrec_bytes : 0), qkeys_l(ctx, A.qrec && nlv > 1 ? sizeof(uint32_t) * (size_t)A.nq : 0);
    if (!qrec_l.ok() || !qkeys_l.ok()) return pc_fail(ctx, PC_NOMEM, "query records");
    for (int i = 0; i < nlv; i++) {
        const pc_grid *l = lv[i];
        if (A.qrec && i > 0) {
            PC_TRY(pc_bin_queries(ctx, l, A.qraw, A.nq, qrec_l.p, qkeys_l.as<uint32_t>(), st));
            A.qrec = qrec_l.p;
            A.qci = CellIndex{nullptr, qkeys_l.as<uint32_t>(), (uint32_t)A.nq, nullptr, 0u};
        }
        A.gp = l->gp;
        A.rec = l->rec;
        A.ci = l->index();
        int64_t nt = 1;
        for (int a = 0; a < 3; a++) {
            A.tile[a] = l->tile[a];
            A.ntile[a] = (l->gp.dim[a] + l->tile[a] - 1) / l->tile[a];
            nt *= A.ntile[a];
        }
        A.tile_list = l->tile_list;
        if (l->tile_list) nt = l->tile_count;
        if (A.qrec && l->tile_list) {
            // the grid's list names the tiles that hold POINTS; external queries need the tiles that hold
            // queries (the block returns to the pool right away: everything using it is ordered on st)
            uint32_t *ql = nullptr;
            PC_TRY(tile_list_from_keys(ctx, l, A.qci.keys, (int64_t)A.qci.n, st, &ql, &nt));
            A.tile_list = ql;
            ctx->release(ql);
        }
        A.cand_cap = l->tile_cap;
        A.qflag_in = i == 0 ? nullptr : qflag;
        A.qflag_out = qflag;  // NULL with a single level: uncertified queries go straight to the list
        if (nt <= 0) continue;
        const int nw = std::max(1, l->tile_waves);
        if (getenv("PC_AMD_DEBUG"))
            fprintf(stderr, "[pc_amd]   tile pass %d: tile %dx%dx%d cap %d waves %d tiles %lld\n", i, l->tile[0], l->tile[1],
                    l->tile[2], l->tile_cap, nw, (long long)nt);
        const size_t lds = sizeof(float) * 4 * ((size_t)A.cand_cap + 2) + sizeof(uint32_t) * (2 * MAX_ROWS + 4) +
                           sizeof(uint16_t) * MAX_ROWS * (MAX_NX + 1) +
                           nw * std::max(sizeof(uint32_t) * PC_WAVE * (SEL_SLOTS + 1), sizeof(uint16_t) * PC_WAVE * (size_t)A.k);
        if (g->dtype == PC_F32) {
            if (l->uneven)
                launch_tile_ks<Rec32, true>(ks, mode, (unsigned)nt, nw, lds, st, A);
            else
                launch_tile_ks<Rec32, false>(ks, mode, (unsigned)nt, nw, lds, st, A);
        } else {
            if (l->uneven)
                launch_tile_ks<Rec64, true>(ks, mode, (unsigned)nt, nw, lds, st, A);
            else
                launch_tile_ks<Rec64, false>(ks, mode, (unsigned)nt, nw, lds, st, A);
        }
        PC_HIP(hipGetLastError());
        if (qflag) {
            // how many points are still open decides whether another (coarser) pass is worth it
            int64_t open = 0;
            PC_TRY(pc_compact_flags(ctx, qflag, (int64_t)npt, nullptr, &open, st));
            if (getenv("PC_AMD_DEBUG"))
                fprintf(stderr, "[pc_amd]   level %d: cell %.4g dims %dx%dx%d tile %dx%dx%d cap %d tiles %lld sparse %d -> open %lld\n",
                        i, l->gp.cell, l->gp.dim[0], l->gp.dim[1], l->gp.dim[2], l->tile[0], l->tile[1], l->tile[2],
                        l->tile_cap, (long long)nt, l->keys != nullptr, (long long)open);
            if (open < std::max<int64_t>(2048, A.nq / 200)) break;
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
        if (g->dtype == PC_F32)
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
    PC_TRY(pc_launch_wave_query(ctx, g, mode, A, st));
    // ... and whatever that stage could not hold is left for the per-lane scan of the caller
    A.qlist = A.flag2_list;
    A.qub = nullptr;
    A.qcount = A.flag2_count;
    *handled = true;
    return PC_OK;
}
