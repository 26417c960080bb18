// Device side of the tile-cooperative exact kNN (see pc_knn_tile.hip for the overview): constants,
// the kernel and its launch switch.  Included by the host file and by one small translation unit per
// (record type, split) pair so that the instantiations compile in parallel.
#pragma once
#include "pc_knn_dev.h"

namespace {

constexpr int ORD_BITS = 11;
constexpr int CAND_CAP_MAX = 1 << ORD_BITS;  // most candidates one tile may stage (32 KiB of LDS)
constexpr uint32_t ORD_MASK = CAND_CAP_MAX - 1;
constexpr uint32_t KEY_EMPTY = 0xFFFFFFFFu;
constexpr uint32_t KEY_INF = 0x7F800000u;  // key of an infinite distance: such keys (sentinels, overflowed sums) never enter a list
constexpr uint32_t DRAIN_ROUNDS = 8;      // parked keys one lane inserts per drain while the scan is running
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) float lds_f32;    // 32-bit LDS addresses: no 64-bit pointer math
typedef __attribute__((address_space(3))) uint32_t lds_u32;
constexpr int MAX_ROWS = 64;  // (tile_y + 2) * (tile_z + 2) must fit one wave
constexpr int MAX_NX = 7;     // tile_x + 2 cells per block row at most (MAX_NX + 1 is a power of two: table indexing)
constexpr int MAX_TILE_WAVES = 4;  // waves of one workgroup, all working on the same staged block
constexpr int UNEVEN_CAP = 1024;   // candidate capacity of the tiles of multi-level indexes
constexpr int SEL_SLOTS = 16; // parked keys per lane between two sorted insertions

// LDS reads by byte address.  Plain loads: the compiler places the waits, and nothing it moves or copies can
// hold a value that has not arrived yet.  (A read issued in one asm statement and waited for in another looks
// finished to the compiler in between; a register copy put there once returned the previous kernel's data.)
typedef __attribute__((address_space(3))) uint16_t lds_u16;
__device__ inline uint32_t lds_read_u32(uint32_t addr) { return *(const lds_u32 *)(uintptr_t)addr; }
__device__ inline uint32_t lds_read_u16(uint32_t addr) { return *(const lds_u16 *)(uintptr_t)addr; }

// Maximum over the wave (uniform result): butterfly inside each row of 16 lanes with DPP
// (pairs, quads, half rows, rows: no LDS traffic), then the four row results through SGPRs.
__device__ inline int wave_max_i32(int v) {
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xF, 0xF, false));  // row_half_mirror
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xF, 0xF, false));  // row_mirror
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

// The filter's inner loop over one row segment, for the whole wave: lanes with j0 < end measure the candidate
// pair (j0, j0 + 1) against their query in packed fp32, build the keys and push them on their stack of
// parked keys (always written to the slot above the stack, which then grows by one slot or not: no
// branch), until every lane is through its segment or some lane's stack has reached wp_full.  nx/ny/nz
// hold the pair at `pa` on entry and on exit; inside, the request for the next pair goes out as soon as this
// one is consumed, and the block leaves with nothing in flight, so that every register it hands back holds
// what the compiler thinks it holds.
// Written as one block because the compiler's version of the same loop spends a third of its instructions
// on exec-mask bookkeeping.  Waits that gfx950 wants between dependent instructions are met by the order:
// a packed-fp32 result is read two instructions later at the earliest (an LDS read or an independent add sits
// in between), a compare's mask three later.
__device__ inline void scan_pairs(uint32_t &j0, const uint32_t end, uint32_t &pa, uint32_t &wp, f32x2 &nx, f32x2 &ny,
                                  f32x2 &nz, const f32x2 qx, const f32x2 qy, const f32x2 qz, const uint32_t thr,
                                  const uint32_t wp_full) {
#if defined(__HIP_DEVICE_COMPILE__)  // (gfx950 text: nothing for the host pass to parse)
    const uint32_t keep = ~ORD_MASK, inc = 4u * PC_WAVE;
    uint64_t saved, m0;
    asm volatile(
        "s_mov_b64 %[saved], exec\n\t"
        "v_cmpx_lt_u32 %[j0], %[end]\n\t"
        "s_cbranch_execz 2f\n"
        "1:\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_pk_add_f32 v[120:121], %[qx], %[nx] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 v[122:123], %[qy], %[ny] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 v[124:125], %[qz], %[nz] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_add_u32 %[pa], 24, %[pa]\n\t"
        "v_pk_mul_f32 v[120:121], v[120:121], v[120:121]\n\t"
        "ds_read2_b32 %[nx], %[pa] offset1:3\n\t"
        "ds_read2_b32 %[ny], %[pa] offset0:1 offset1:4\n\t"
        "v_pk_fma_f32 v[120:121], v[122:123], v[122:123], v[120:121]\n\t"
        "ds_read2_b32 %[nz], %[pa] offset0:2 offset1:5\n\t"
        "v_pk_fma_f32 v[120:121], v[124:125], v[124:125], v[120:121]\n\t"
        "v_add_u32 v127, 1, %[j0]\n\t"
        "v_and_or_b32 v126, v120, %[keep], %[j0]\n\t"
        "v_and_or_b32 v127, v121, %[keep], v127\n\t"
        "v_cmp_lt_u32 %[m0], v126, %[thr]\n\t"
        "ds_write_b32 %[wp], v126 offset:256\n\t"
        "v_cmp_lt_u32 vcc, v127, %[thr]\n\t"
        "v_add_u32 %[j0], 2, %[j0]\n\t"
        "v_cndmask_b32 v126, 0, %[inc], %[m0]\n\t"
        "v_add_u32 %[wp], %[wp], v126\n\t"
        "v_cndmask_b32 v126, 0, %[inc], vcc\n\t"
        "ds_write_b32 %[wp], v127 offset:256\n\t"
        "v_add_u32 %[wp], %[wp], v126\n\t"
        "v_cmp_ge_u32 vcc, %[wp], %[full]\n\t"
        "s_cbranch_vccnz 2f\n\t"
        "v_cmpx_lt_u32 %[j0], %[end]\n\t"
        "s_cbranch_execnz 1b\n"
        "2:\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_mov_b64 exec, %[saved]"
        : [j0] "+v"(j0), [pa] "+v"(pa), [wp] "+v"(wp), [nx] "+v"(nx), [ny] "+v"(ny), [nz] "+v"(nz), [saved] "=&s"(saved),
          [m0] "=&s"(m0)
        : [end] "v"(end), [qx] "v"(qx), [qy] "v"(qy), [qz] "v"(qz), [thr] "v"(thr), [full] "v"(wp_full), [keep] "s"(keep),
          [inc] "v"(inc)
        : "vcc", "scc", "memory", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127");
#endif
}

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
template <typename REC, int KS, int MODE, bool UNEVEN>
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
    uint32_t *row_key = row_off + MAX_ROWS + 4;                                                    // [64] key of the row's cell 0
    uint16_t *cell_off = reinterpret_cast<uint16_t *>(row_key + MAX_ROWS);                         // [64][MAX_NX+1]
    // one area of lane-private columns: parked keys during the filter, result order afterwards
    const int tid = threadIdx.x, lane = tid & (PC_WAVE - 1), wid = tid >> 6, nthr = blockDim.x;
    // (one such area per wave)
    uint16_t *cell_mid = cell_off + MAX_ROWS * (MAX_NX + 1);                                       // [64][MAX_NX+1]
    // (the table of mid points exists only where it is used: the launch sizes LDS the same way)
    uint32_t *selbuf = reinterpret_cast<uint32_t *>(cell_mid + (A.halves ? MAX_ROWS * (MAX_NX + 1) : 0)) +
                       wid * max(PC_WAVE * (SEL_SLOTS + 1), (PC_WAVE / 2) * A.k);           // [SEL_SLOTS + 1][64]
    uint16_t *ordbuf = reinterpret_cast<uint16_t *>(selbuf);                                       // [k][64]
    const GridParams &gp = A.gp;
    const REC *rec = reinterpret_cast<const REC *>(A.rec);
    // where the queries come from: the grid's own records, or external queries binned into the grid
    const REC *qsrc = A.qrec ? reinterpret_cast<const REC *>(A.qrec) : rec;
    const CellIndex &qix = A.qrec ? A.qci : A.ci;
    const int k = A.k;
    // Records of a cell ordered low x-half first: a query in the low half of its cell leaves out the high half
    // of the cell after it, one in the high half the low half of the cell before it -- 2.5 cells per row
    // instead of 3, and the scanned region still reaches a full cell edge beyond the query on either side.
    const bool halves = !UNEVEN && !TileRec<REC>::kRelative && A.halves;

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
    uint32_t my_row_key = 0;
    if (lane < nrows) {
        // (lane / nry without an integer division: exact for these small numbers)
        const int rz0 = (int)(((float)lane + 0.5f) * (1.0f / (float)nry));
        const int y = lo[1] + lane - rz0 * nry, z = lo[2] + rz0;
        const int64_t row = (int64_t)(((uint32_t)z * (uint32_t)gp.dim[1] + (uint32_t)y) * (uint32_t)gp.dim[0]);  // < 2^32 cells
        my_row_key = (uint32_t)row;
        // (a searched index looks both ends of a range up together, several pivots per round trip)
        if (A.ci.table) {
            row_begin = cell_begin(A.ci, row + lo[0]);
            row_cnt = cell_begin(A.ci, row + hi[0] + 1) - row_begin;
        } else {
            uint32_t e;
            cell_range(A.ci, row + lo[0], row + hi[0] + 1, row_begin, e);
            row_cnt = e - row_begin;
        }
        if (y >= c0[1] && y <= c1[1] && z >= c0[2] && z <= c1[2]) {  // a row of the tile itself
            if (qix.table) {
                q_begin = cell_begin(qix, row + c0[0]);
                q_cnt = cell_begin(qix, row + c1[0] + 1) - q_begin;
            } else {
                uint32_t e;
                cell_range(qix, row + c0[0], row + c1[0] + 1, q_begin, e);
                q_cnt = e - q_begin;
            }
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
    // flagged points of an over-full block stay flagged (where passes may postpone queries, they are re-flagged as
    // crowded below, so that no coarser level lists them again)
    if (A.qflag_in && too_many && !splittable && A.flag_top == 0) return 0;
    if (too_many && splittable) return 1;                 // (before looking for flagged points: a coarse tile holds millions)
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
                any = any || flag_live(A.qflag_in[id], A.flag_max);
            }
            if (__any(any)) break;  // (one is enough: the batches below look at their own flags)
        }
        if (!__any(any)) return 0;
    }

    // over-full blocks do not fit LDS; blocks with fewer than k points cannot certify anything unless
    // they are the whole grid: both go straight to the re-run list
    const bool whole_grid = lo[0] == 0 && lo[1] == 0 && lo[2] == 0 && hi[0] == gp.dim[0] - 1 &&
                            hi[1] == gp.dim[1] - 1 && hi[2] == gp.dim[2] - 1;
    const bool overfull = ncand > (uint32_t)A.cand_cap || (ncand < (uint32_t)k && !whole_grid && A.radius2 < 0.0);

    // ---- stage the candidate block into LDS ------------------------------------------------------
    // flattened over the block's candidates: lane i copies candidates i, i+64, ...; the row of a
    // candidate follows from the row offsets, and four 16-byte loads are in flight per lane
#if defined(ABL_STAGE) && ABL_STAGE == 1
    if (ncand != 0xdeadbeefu) return 0;
#endif
    row_src[lane] = row_begin;
    row_off[lane] = cand_off;
    row_key[lane] = my_row_key;
    if (lane == 0) row_off[MAX_ROWS] = ncand;
    __syncthreads();
    if (!overfull && !UNEVEN) {
        // Tiles of an evenly filled grid: a block row holds a few dozen candidates, so a wave copies whole rows, one
        // row per step with lane = candidate of the row, the rows' parameters read from lane r of the row table
        // (uniform), up to eight rows' loads in flight.  No per-candidate search for the row: the flattened loop
        // below spends 28 vector instructions per candidate on that, this one 3 (569 -> about 190 per wave of the
        // 10 M cloud's kernel).  The rare row of more than 64 candidates finishes in a loop of its own.
        const int nwaves = nthr >> 6;
        constexpr int ROWS_AT_ONCE = 8;
        for (int r0 = wid; r0 < nrows; r0 += ROWS_AT_ONCE * nwaves) {  // uniform
            float4 v[ROWS_AT_ONCE];
            uint32_t dst[ROWS_AT_ONCE];
#pragma unroll
            for (int b = 0; b < ROWS_AT_ONCE; b++) {
                const int r = min(r0 + b * nwaves, nrows - 1);
                const uint32_t src0 = __builtin_amdgcn_readlane(row_begin, r), cnt_r = __builtin_amdgcn_readlane(row_cnt, r),
                               dst0 = __builtin_amdgcn_readlane(cand_off, r);
                const bool have = r0 + b * nwaves < nrows && (uint32_t)lane < cnt_r;
                dst[b] = have ? dst0 + lane : 0xffffffffu;
                v[b] = TileRec<REC>::stage(rec, have ? src0 + lane : 0u, org);  // (record 0 exists: the tile has queries)
            }
#pragma unroll
            for (int b = 0; b < ROWS_AT_ONCE; b++) {
                if (dst[b] != 0xffffffffu) {
                    cxyz[3 * dst[b]] = v[b].x;
                    cxyz[3 * dst[b] + 1] = v[b].y;
                    cxyz[3 * dst[b] + 2] = v[b].z;
                    cws[dst[b]] = v[b].w;
                }
            }
        }
        for (int r = wid; r < nrows; r += nwaves) {  // what a long row has beyond its first 64 candidates
            const uint32_t src0 = __builtin_amdgcn_readlane(row_begin, r), cnt_r = __builtin_amdgcn_readlane(row_cnt, r),
                           dst0 = __builtin_amdgcn_readlane(cand_off, r);
            for (uint32_t o = PC_WAVE + lane; o < cnt_r; o += PC_WAVE) {
                const float4 w = TileRec<REC>::stage(rec, src0 + o, org);
                cxyz[3 * (dst0 + o)] = w.x;
                cxyz[3 * (dst0 + o) + 1] = w.y;
                cxyz[3 * (dst0 + o) + 2] = w.z;
                cws[dst0 + o] = w.w;
            }
        }
    } else if (!overfull) {
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
#if defined(ABL_STAGE) && ABL_STAGE == 2
    if (ncand != 0xdeadbeefu) return 0;
#endif
    // Two sentinel candidates behind the block, infinitely far away: the filter reads candidates in pairs and may
    // read one past the end of a row segment (see stage 1).
    if (!overfull && tid < 6) cxyz[3 * ncand + tid] = 1e30f;
    // block-relative offset of every cell boundary of every block row
    const int nx = hi[0] - lo[0] + 1;
    for (int e = tid; e < nrows * (MAX_NX + 1); e += nthr) {  // (entry e of the table: row e / 8, boundary e % 8)
        const int r = e >> 3, i = e & MAX_NX;
        if (i <= nx)
            cell_off[e] = (uint16_t)(row_off[r] + (cell_begin(A.ci, (int64_t)row_key[r] + lo[0] + i) - row_src[r]));
    }
    __syncthreads();
#if defined(ABL_STAGE) && ABL_STAGE == 3
    if (ncand != 0xdeadbeefu) return 0;
#endif
    if (halves) {
        // first high-half record of every block cell (the records of a cell are ordered by that bit)
        for (int e = tid; e < nrows * (MAX_NX + 1); e += nthr) {
            const int r = e >> 3, i = e & MAX_NX;
            if (i >= nx) continue;
            int a = cell_off[r * (MAX_NX + 1) + i], b = cell_off[r * (MAX_NX + 1) + i + 1];
            while (a < b) {
                const int m = (a + b) >> 1;
                if (x_half_of((double)cxyz[3 * m], gp, lo[0] + i))
                    b = m;
                else
                    a = m + 1;
            }
            cell_mid[r * (MAX_NX + 1) + i] = (uint16_t)a;
        }
        __syncthreads();
    }

    const double eps = 1e-7 * gp.cell;
    const double r2lim = A.radius2;
#ifdef ABL_NOQUERY
    if (ncand != 0xdeadbeefu) return 0;
#endif

    for (uint32_t qbase = wid * PC_WAVE; qbase < nquery; qbase += nthr) {  // the waves share the tile's queries
        const uint32_t qo = qbase + lane;
        const bool active = qo < nquery;
        // which row of the tile holds query ordinal qo (the rows of the halo hold none); that row is also
        // the query's cell along y and z
        uint32_t spos = 0;
        int ry = 0, rz = 0;  // the query's row, in block coordinates
        for (int z = c0[2] - lo[2]; z <= c1[2] - lo[2]; z++)
            for (int y = c0[1] - lo[1]; y <= c1[1] - lo[1]; y++) {  // (uniform: lane reads outside divergent control flow)
                const int r = y + nry * z;
                const uint32_t ob = __builtin_amdgcn_readlane(q_off, r), oc = __builtin_amdgcn_readlane(q_cnt, r),
                               gb = __builtin_amdgcn_readlane(q_begin, r);
                const bool here = qo - ob < oc;  // (unsigned: ob <= qo < ob + oc)
                spos = here ? gb + (qo - ob) : spos;
                ry = here ? y : ry;
                rz = here ? z : rz;
            }
        double qxd = 0.0, qyd = 0.0, qzd = 0.0;
        int32_t qid = 0;
        bool live = active;  // a query this pass has to answer
        if (active) {
            RecTraits<REC>::load(qsrc + spos, qxd, qyd, qzd, qid);
            if (A.qflag_in) live = flag_live(A.qflag_in[qid], A.flag_max);
        }
        if (!__any(live)) continue;
        if (overfull) {
            if (too_many && A.count_crowded) {  // crowded queries counted (finer grids worth making?)
                const unsigned long long m = __ballot(live);
                if (lane == 0 && m) atomicAdd(A.flag_count + 26, (uint32_t)__popcll(m));
            }
            if (live) {
                if (A.qflag_out) {
                    A.qflag_out[qid] = too_many && A.flag_top > 0 ? FLAG_CROWDED : (uint8_t)1;
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
        // segments are visited one after the other, each until the wave's longest segment is done
        // (lanes with shorter ones sit out the rest of that loop).
        // A key can enter the list only if it is below the list's current last entry.  The lane
        // keeps a copy `thr` of that entry and parks the keys below it on a stack in its own column of
        // selbuf; the (expensive) sorted insertion runs over parked keys only.  When some lane's stack is
        // nearly full every lane inserts its newest DRAIN_ROUNDS keys and keeps the rest parked: the wave
        // then pays for about the mean stack, not for the fullest one (rounds per 64 queries 89 -> 69 in
        // a model of the uniform 10 M cloud); what is left is inserted at the end.  `thr` lags behind
        // the list, which only lets a few keys too many through: the final list is the same KS
        // smallest keys either way.
        uint32_t thr0 = KEY_INF;  // no key at or above it is ever wanted (set below, once the query's cells are known)
        uint32_t thr = KEY_INF;
#ifdef PC_COUNT
        uint32_t n_steps = 0, n_rounds = 0, n_drains = 0, n_lock = 0;
#endif
        lds_u32 *const col = (lds_u32 *)selbuf + lane;  // slot s of this lane: col[s * 64]
        const uint32_t col_a = (uint32_t)(uintptr_t)col;                   // LDS byte address of the column
        uint32_t wp = col_a - 4u * PC_WAVE;                                // last written slot (byte address)
        auto drain = [&](const uint32_t rounds) {  // (rounds: the same for all lanes)
            // eight slots at a time, top down: the eight reads go out together (slots below the column's first
            // are read too -- the tables in front of the columns, at least 4 KB of them -- and not used)
            uint32_t ra = wp;
#pragma nounroll
            for (uint32_t r0 = 0; r0 < rounds; r0 += DRAIN_ROUNDS) {
                uint32_t key[DRAIN_ROUNDS];
#pragma unroll
                for (uint32_t i = 0; i < DRAIN_ROUNDS; i++) key[i] = lds_read_u32(ra - i * 4u * PC_WAVE);
#pragma unroll
                for (uint32_t i = 0; i < DRAIN_ROUNDS; i++) {
                    if (r0 + i < rounds) {  // uniform
                        list_insert<KS>(L, ra - i * 4u * PC_WAVE >= col_a ? key[i] : KEY_EMPTY);
#ifdef PC_COUNT
                        n_rounds++;
#endif
                    }
                }
                ra -= min(rounds - r0, DRAIN_ROUNDS) * 4u * PC_WAVE;
            }
#ifdef PC_COUNT
            n_drains++;
#endif
            wp = ra >= col_a ? ra : col_a - 4u * PC_WAVE;
            thr = min(L[KS - 1], thr0);
        };
        int qc[3];
        {   // the cell along x by the expression that made the cell keys; along y and z it is the row's
            const double tt = floor((qxd - gp.o[0]) * gp.s);
            qc[0] = !(tt > 0.0) ? 0 : (tt >= (double)gp.dim[0] ? gp.dim[0] - 1 : (int)tt);
            qc[1] = lo[1] + ry;
            qc[2] = lo[2] + rz;
        }
        const int i0 = max(qc[0] - 1, 0) - lo[0], i1 = min(qc[0] + 1, gp.dim[0] - 1) - lo[0] + 1;
        // which half-cell the walk leaves out: +1 the high half of the cell after the query's, -1 the low
        // half of the cell before it, 0 nothing (no such cell, or no halves in this grid)
        int trim = 0;
        if (halves) {
            const uint32_t hq = x_half_of(qxd, gp, qc[0]);
            if (!hq && qc[0] + 1 <= gp.dim[0] - 1) trim = 1;
            if (hq && qc[0] - 1 >= 0) trim = -1;
        }
        // distance to the nearest face of the scanned cells, measured like the cell keys: in offsets from the
        // grid origin (a face position added to a far origin would round at the origin's magnitude)
        const double qv[3] = {qxd - gp.o[0], qyd - gp.o[1], qzd - gp.o[2]};
        bool whole = true;
        double dmin = INFINITY;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const int nlo = max(qc[a] - 1, 0), nhi = min(qc[a] + 1, gp.dim[a] - 1);
            if (nlo > 0) {
                whole = false;
                dmin = fmin(dmin, qv[a] - (double)nlo * gp.cell);
            }
            if (nhi < gp.dim[a] - 1) {
                whole = false;
                dmin = fmin(dmin, (double)(nhi + 1) * gp.cell - qv[a]);
            }
            if (a == 0 && trim != 0) {  // the half-cell left out begins in the middle of its cell
                whole = false;
                const double mid = ((double)(qc[0] + trim) + 0.5) * gp.cell;
                dmin = fmin(dmin, trim > 0 ? mid - qv[0] : qv[0] - mid);
            }
        }
        double bound = INFINITY;  // no unscanned point is closer than sqrt(bound)
        if (!whole) {
            const double b = dmin - eps;
            bound = b > 0 ? b * b : 0.0;
        }
        // A candidate at or beyond that bound (or beyond the search radius) can never be one of the certified
        // neighbours: the threshold starts there instead of at infinity, and a quarter fewer keys are parked
        // (rounds of insertion per 64 queries 69 -> 54 in a model of the uniform 10 M cloud).  The margin of
        // 0.1 % covers the truncation of the keys (2^-12), the fp32 measurement and, for fp64 grids, the rounded
        // frame of the filter: what stays unlisted for this reason alone is farther than `bound`.
        {
            double seed = fmin(bound, r2lim >= 0.0 ? r2lim : INFINITY);
            if (seed < INFINITY) {
                if (TileRec<REC>::kRelative) {
                    const double rs = sqrt(seed) + 4.76837158203125e-07 * bext;
                    seed = rs * rs;
                }
                thr0 = min(KEY_INF, __float_as_uint((float)(seed * A.seed_scale)) & ~ORD_MASK);
            }
            thr = thr0;
        }
        // the nine (start, length) pairs up front: two table entries each, all reads in flight at once.
        // Order: the query's own row, the four rows sharing a face with it, the four diagonal ones,
        // so that the threshold tightens early.
        constexpr int SEG_DY[9] = {0, -1, 1, 0, 0, -1, 1, -1, 1};
        constexpr int SEG_DZ[9] = {0, 0, 0, -1, 1, -1, -1, 1, 1};
        uint32_t seg_pack[9];  // start | len << 16
        {
            constexpr uint32_t ROW_BYTES = 2u * (MAX_NX + 1), MID_BYTES = 2u * MAX_ROWS * (MAX_NX + 1);
            const uint32_t tb = (uint32_t)(uintptr_t)(const lds_u16 *)cell_off + ROW_BYTES * (uint32_t)(ry + nry * rz);
            const uint32_t a_st = tb + 2u * (uint32_t)i0 + (trim < 0 ? MID_BYTES : 0u);
            const uint32_t a_en = tb + 2u * (uint32_t)(trim > 0 ? i1 - 1 : i1) + (trim > 0 ? MID_BYTES : 0u);
            const bool ym = ry > 0, yp = ry < nry - 1, zm = rz > 0, zp = rz < nrz - 1;  // the neighbouring rows exist
            uint32_t st[9], en[9];
#pragma unroll
            for (int seg = 0; seg < 9; seg++) {
                const uint32_t off = ROW_BYTES * (uint32_t)(SEG_DY[seg] + nry * SEG_DZ[seg]);  // (wraps: fine)
                const bool ok = live && (SEG_DY[seg] < 0 ? ym : (SEG_DY[seg] > 0 ? yp : true)) &&
                                (SEG_DZ[seg] < 0 ? zm : (SEG_DZ[seg] > 0 ? zp : true));
                // (rows that do not exist read the query's own row instead: any valid address)
                st[seg] = lds_read_u16(ok ? a_st + off : a_st);
                en[seg] = lds_read_u16(ok ? a_en + off : a_st);
            }
#pragma unroll
            for (int seg = 0; seg < 9; seg++) seg_pack[seg] = (st[seg] & 0xffffu) | ((en[seg] - st[seg]) << 16);
        }
        // A query with fewer than 2 k candidates in its 27 cells will not find k of them inside the reach of
        // those cells (a ball inscribed in the block holds a fifth of an evenly filled block's points, under
        // half of a surface's): it is handed on without costing the wave any steps.  (Not when the 27 cells
        // are the whole grid or a radius limits the search: few neighbours are the answer then.  Levels of
        // uneven clouds only: an evenly filled grid has no such queries to speak of.)
        if (UNEVEN) {
            uint32_t total = 0;
            bool whole_grid = true;
#pragma unroll
            for (int seg = 0; seg < 9; seg++) total += seg_pack[seg] >> 16;
#pragma unroll
            for (int a = 0; a < 3; a++) whole_grid = whole_grid && qc[a] <= 1 && qc[a] + 1 >= gp.dim[a] - 1;
            if (total < 2u * (uint32_t)k && !whole_grid && r2lim < 0.0) {
#pragma unroll
                for (int seg = 0; seg < 9; seg++) seg_pack[seg] &= 0xffffu;
            }
        }
#ifdef ABL_NOSCAN
        if (seg_pack[0] != 0xdeadbeefu && L[3] != 7u) continue;
#endif
        // The wave leaves a segment when its longest lane is through: within the four face rows, and within the
        // four diagonal rows, every lane walks its segments longest first (the length is the high half of the
        // packed word), so that long meets long (steps per 64 queries 115 -> 108 in the model of the 10 M cloud).
        if (!UNEVEN) {
            auto ce = [](uint32_t &a, uint32_t &b) {
                const uint32_t hi = max(a, b), lo = min(a, b);
                a = hi;
                b = lo;
            };
#pragma unroll
            for (int g = 1; g < 9; g += 4) {
                ce(seg_pack[g], seg_pack[g + 1]);
                ce(seg_pack[g + 2], seg_pack[g + 3]);
                ce(seg_pack[g], seg_pack[g + 2]);
                ce(seg_pack[g + 1], seg_pack[g + 3]);
                ce(seg_pack[g + 1], seg_pack[g + 2]);
            }
        }
        const f32x2 qx2 = {qx, qx}, qy2 = {qy, qy}, qz2 = {qz, qz};
        const uint32_t cand_a = (uint32_t)(uintptr_t)(const lds_f32 *)cxyz;  // LDS byte address
        const uint32_t wp_full = col_a + 4u * (SEL_SLOTS - 2) * PC_WAVE;  // (wp names the last written slot)
#pragma unroll
        for (int seg = 0; seg < 9; seg++) {
            // Two candidates per step: one ds_read2_b32 per coordinate, packed fp32 math
            // (v_pk_add/mul/fma_f32); the next pair is requested as soon as this one is consumed.  A
            // segment of odd length takes the candidate behind it along: the next record of the block (a
            // point outside this query's cells, or the first point of the next row, which the query may
            // meet again in its own walk -- the exact finish drops such a double) or the sentinel behind
            // the block.  Extra candidates are real points with true distances: the result is exact
            // either way, and leaving out the test of the second candidate saves two instructions a step.
            uint32_t j0 = seg_pack[seg] & 0xffffu;
            const uint32_t end = j0 + (seg_pack[seg] >> 16);
            uint32_t pa = cand_a + 12u * j0;
            // (the first pair by plain loads; the candidate block is followed by two sentinels, so a pair
            // behind the last candidate is readable)
            const lds_f32 *const cp = (const lds_f32 *)(uintptr_t)pa;
            f32x2 nx = {cp[0], cp[3]}, ny = {cp[1], cp[4]}, nz = {cp[2], cp[5]};
            for (;;) {
                scan_pairs(j0, end, pa, wp, nx, ny, nz, qx2, qy2, qz2, thr, wp_full);
                // (all lanes are back together here: the drain serves the whole wave, also the lanes whose
                // segment had ended before)
                if (__any(wp >= wp_full)) drain(DRAIN_ROUNDS);
                if (!__any(j0 < end)) break;
            }
#ifdef PC_COUNT
            n_lock += (uint32_t)wave_max_i32((int)n_steps);
            n_steps = 0;
#endif
        }
        drain((uint32_t)wave_max_i32((int)((wp + 4u * PC_WAVE - col_a) >> 8)));  // everything: the fullest stack's height
#ifdef PC_COUNT
        {   // per 64-query batch: the longest lane's steps, rounds, drains
            const int ms = (int)n_lock, mr = wave_max_i32((int)n_rounds), md = wave_max_i32((int)n_drains);
            if (lane == 0) {
                atomicAdd(A.flag_count + 24, (uint32_t)ms);
                atomicAdd(A.flag_count + 25, (uint32_t)mr);
                atomicAdd(A.flag_count + 27, (uint32_t)md);
                atomicAdd(A.flag_count + 28, 1u);
            }
        }
#endif
#ifdef ABL_NOFINISH
        if (L[0] != 7u) { if (L[KS-1] == 12345u) A.out_cnt[qid] = 1; continue; }
#endif
        if (!live) continue;

        // ---- stage 2: exact finish --------------------------------------------------------------------
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

        // The KS listed candidates re-measured in fp64, without a branch: every slot is read and measured (an
        // empty slot or the second copy of a double reads candidate 0 and is left out afterwards), so that all
        // the LDS reads are in flight together.
        const double r2eff = r2lim >= 0.0 ? r2lim : INFINITY;
        double dd[KS];
        uint32_t od[KS];
        // (the long lists of a kernel that runs one wave per SIMD anyway keep the original indices they have just
        // read with the records: rows and ties need no second look at global memory)
        constexpr bool KEEP_ID = KS > 22 && KS <= 34 && (MODE == M_KNN || MODE == M_ALL);
        int32_t ci[KEEP_ID ? KS : 1];
        int nsafe = 0, cnt = 0, nvalid = 0;
        uint32_t last = 0;  // the largest listed key (the list's registers are free once this loop has read them)
#pragma unroll
        for (int s = 0; s < KS; s++) {
            last = L[s] != KEY_EMPTY ? L[s] : last;
            // (a key listed twice is the candidate a segment of odd length took along, met again in the walk)
            const bool valid = L[s] != KEY_EMPTY && (s == 0 || L[s] != L[s - 1]);
            od[s] = valid ? (L[s] & ORD_MASK) : 0u;
            double cx, cy, cz;
            int32_t cid;
            // (fp64 grids fetch the record by its sorted position: position 0 for a slot that is left out --
            // candidate 0 of a block without candidates was never staged; fp32 candidates carry their original
            // index in that word, read only where it is kept)
            TileRec<REC>::exact(rec, make_float4(cxyz[3 * od[s]], cxyz[3 * od[s] + 1], cxyz[3 * od[s] + 2],
                                                 (TileRec<REC>::kRelative || KEEP_ID) && valid ? cws[od[s]] : 0.f), cx, cy, cz, cid);
            const double dx = qxd - cx, dy = qyd - cy, dz = qzd - cz;
            double d = dx * dx;
            d += dy * dy;
            d += dz * dz;
            const bool safe = valid && d < bound, in = safe && d < r2eff;
            nvalid += valid ? 1 : 0;
            nsafe += safe ? 1 : 0;
            cnt += in ? 1 : 0;
            dd[s] = in ? d : INFINITY;
            if (KEEP_ID) ci[s] = cid;
        }
        const bool ok = nsafe >= k || bound == INFINITY || (r2lim >= 0.0 && bound >= r2lim);
        if (!ok) {
            // an upper bound of the k-th distance when k candidates are listed: the largest listed key, with
            // what truncation (2^-12) and the fp32 measurement (the rounded frame of fp64 grids) can hide
            float ubf = INFINITY;
            if (nvalid >= k) {
                const double th = (double)__uint_as_float(last & ~ORD_MASK) * (1.0 + 2.44140625e-4);
                if (TileRec<REC>::kRelative) {
                    const double rt = sqrt(th) * (1.0 + 1e-6) + 4.76837158203125e-07 * bext;
                    ubf = (float)(rt * rt * (1.0 + 1e-6));
                } else {
                    ubf = (float)(th * (1.0 + 2e-6));
                }
            }
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
            // (a look at the order costs a seventh of a pass: most waves need one pass, then one look)
            bool unordered = false;
#pragma unroll
            for (int s = 0; s + 1 < KS; s++) unordered |= dd[s] > dd[s + 1];
            if (!__any(unordered)) break;
#pragma unroll
            for (int s = 0; s + 1 < KS; s++) {
                const double da = dd[s], db = dd[s + 1];
                const uint32_t oa = od[s], ob = od[s + 1];
                const bool gt = da > db;
                dd[s] = fmin(da, db);
                dd[s + 1] = fmax(da, db);
                od[s] = gt ? ob : oa;
                od[s + 1] = gt ? oa : ob;
                if (KEEP_ID) {
                    const int32_t ia = ci[s], ib = ci[s + 1];
                    ci[s] = gt ? ib : ia;
                    ci[s + 1] = gt ? ia : ib;
                }
            }
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
                    if (dd[s] == dd[s + 1] && dd[s] != INFINITY) gt = KEEP_ID ? ci[s] > ci[s + 1] : orig_index(oa) > orig_index(ob);
                    od[s] = gt ? ob : oa;
                    od[s + 1] = gt ? oa : ob;
                    if (KEEP_ID) {
                        const int32_t ia = ci[s], ib = ci[s + 1];
                        ci[s] = gt ? ib : ia;
                        ci[s + 1] = gt ? ia : ib;
                    }
                    swapped |= gt;
                }
                if (!__any(swapped)) break;
            }
        }
        cnt = min(cnt, k);
#ifdef ABL_NOEMIT
        if (dd[0] != 12345.0) { if (od[KS - 1] == 54321u) A.out_cnt[qid] = 1; continue; }
#endif
        // Rows and mean distance straight from the sorted registers (no second look at the candidates).  Lists up
        // to 34 entries afford it: their kernels run one wave per SIMD whatever they keep alive, and the generic
        // consumer's walk, four records in flight at a time with nothing else to run, cost 66 M voxels with k = 32
        // 5 of a tile pass's 54 ms.  Longer lists go through the generic consumer.
        constexpr bool DIRECT = KS <= 34;
        auto index_at = [&](int s) { return KEEP_ID ? ci[KEEP_ID ? s : 0] : orig_index(od[s]); };
        const int64_t orow = out_row_of(A, qid, spos);  // row of the outputs
        if (DIRECT && (MODE == M_KNN || MODE == M_ALL) && (A.out_idx || A.out_d2 || A.out_cnt)) {
            const int64_t row = orow * k;
            if ((k & 3) == 0) {  // (uniform) rows of 16-byte groups: vector stores
#pragma unroll
                for (int s = 0; s + 3 < KS; s += 4) {
                    if (s < k) {
                        if (A.out_idx) {
                            int4 v;
                            v.x = s < cnt ? index_at(s) : -1;
                            v.y = s + 1 < cnt ? index_at(s + 1) : -1;
                            v.z = s + 2 < cnt ? index_at(s + 2) : -1;
                            v.w = s + 3 < cnt ? index_at(s + 3) : -1;
                            *reinterpret_cast<int4 *>(A.out_idx + row + s) = v;
                        }
                        if (A.out_d2) {
                            const double d0 = s < cnt ? dd[s] : INFINITY, d1 = s + 1 < cnt ? dd[s + 1] : INFINITY,
                                         d2v = s + 2 < cnt ? dd[s + 2] : INFINITY, d3 = s + 3 < cnt ? dd[s + 3] : INFINITY;
                            if (A.d2_f64) {
                                double *o = reinterpret_cast<double *>(A.out_d2) + row + s;
                                *reinterpret_cast<double2 *>(o) = make_double2(d0, d1);
                                *reinterpret_cast<double2 *>(o + 2) = make_double2(d2v, d3);
                            } else {
                                *reinterpret_cast<float4 *>(reinterpret_cast<float *>(A.out_d2) + row + s) =
                                    make_float4((float)d0, (float)d1, (float)d2v, (float)d3);
                            }
                        }
                    }
                }
            } else {
#pragma unroll
                for (int s = 0; s < KS; s++) {
                    if (s < k) {  // wave-uniform
                        const bool have = s < cnt;
                        if (A.out_idx) A.out_idx[row + s] = have ? index_at(s) : -1;
                        if (A.out_d2) {
                            const double d = have ? dd[s] : INFINITY;
                            if (A.d2_f64)
                                reinterpret_cast<double *>(A.out_d2)[row + s] = d;
                            else
                                reinterpret_cast<float *>(A.out_d2)[row + s] = (float)d;
                        }
                    }
                }
            }
            if (A.out_cnt) A.out_cnt[orow] = cnt;
        }
        if (DIRECT && MODE == M_KNN) continue;
        if (DIRECT && (MODE == M_SOR || (MODE == M_ALL && A.out_f64))) {
            // mean neighbour distance, summed nearest first like the generic consumer
            double sum = 0.0;
#pragma unroll
            for (int s = 0; s < KS; s++)
                if (s < cnt) sum += sqrt(dd[s]);
            A.out_f64[orow] = cnt > 0 ? sum / (double)cnt : -1.0;
            if (MODE == M_SOR) continue;
        }
        if (MODE == M_NNDIST) {
            A.out_f64[orow] = cnt > 1 ? sqrt(dd[1]) : 0.0;
            continue;
        }
        if (DIRECT && (MODE == M_NORMALS || MODE == M_ALL)) {
            // normals of the short lists: the cumulants straight from the sorted registers (same order of
            // additions as the generic consumer), all candidate reads in flight at once
            if (MODE == M_NORMALS || A.out_nrm) {
                double cu[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
                if (cnt >= 3) {
#pragma unroll
                    for (int s = 0; s < KS; s++) {
                        if (s < cnt) {
                            double x, y, z;
                            int32_t nid;
                            TileRec<REC>::exact(rec, make_float4(cxyz[3 * od[s]], cxyz[3 * od[s] + 1], cxyz[3 * od[s] + 2], cws[od[s]]), x, y, z, nid);
                            cumulant_add(cu, x, y, z);
                        }
                    }
                }
                emit_normal<MODE>(A, orow, cnt, cu);
            }
            continue;
        }
#pragma unroll
        for (int s = 0; s < KS; s++)
            if (s < k) ordbuf[s * PC_WAVE + lane] = (uint16_t)od[s];
        emit_result<MODE, DIRECT>(A, orow, cnt,
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

// Four waves per SIMD for the short lists whatever the consumer (plain rows fit anyway): the eigen-solver of the
// normals peaks above 128 registers once per query, far from the filter loop that sets the pace, and may spill
// there (fused pipeline on 10 M points: 4.28 -> 3.97 ms).
template <typename REC, int KS, int MODE, bool SPLIT>
__global__ __launch_bounds__(PC_WAVE * MAX_TILE_WAVES) __attribute__((amdgpu_waves_per_eu(KS <= 18 && MODE != M_KNN ? 4 : 1)))
void k_tile_scan(ScanArgs A) {
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
    // (tile ids are below 2^31 -- the launch checks it --: 32-bit divisions, a third of the 64-bit ones' instructions)
    const uint32_t t = A.tile_list ? A.tile_list[bid] : bid + (uint32_t)A.tile_base;
    const uint32_t n0 = (uint32_t)A.ntile[0], n1 = (uint32_t)A.ntile[1];
    const uint32_t t1 = t / n0;
    const int tx = (int)(t - t1 * n0);
    const int tz = (int)(t1 / n1);
    const int ty = (int)(t1 - (uint32_t)tz * n1);
    const int c0[3] = {tx * A.tile[0], ty * A.tile[1], tz * A.tile[2]};  // first cell of the tile
    const int ext[3] = {A.tile[0], A.tile[1], A.tile[2]};
    if (!SPLIT) {  // (clouds that fill their box: the tile shape was sized for the LDS capacity)
        tile_body<REC, KS, MODE, false>(A, smem, c0, ext, false);
        return;
    }
    // The whole tile first; if its block is too full for LDS, its (up to) eight half-size parts one after the other, and
    // a part that is too full still, its own eight parts: big tiles fill the waves where the cloud is thin without
    // giving up the dense parts (a tile of 3x3x3 cells comes down to single cells next to a wall).  One call site of
    // the tile's body: s1 / s2 name the part (of the tile) and the part of the part being worked on, -1: the whole.
    int s1 = -1, s2 = -1;
    for (;;) {
        int bc0[3] = {c0[0], c0[1], c0[2]}, bext[3] = {ext[0], ext[1], ext[2]};
        bool empty = false;
#pragma unroll
        for (int lvl = 0; lvl < 2; lvl++) {
            const int sub = lvl == 0 ? s1 : s2;
            if (sub < 0) continue;
#pragma unroll
            for (int a = 0; a < 3; a++) {
                const int half = (bext[a] + 1) / 2, hi_part = (sub >> a) & 1;
                bc0[a] += hi_part * half;
                bext[a] = hi_part ? bext[a] - half : half;
                empty |= bext[a] <= 0 || bc0[a] >= A.gp.dim[a];
            }
        }
        int again = 0;
        if (!empty) again = tile_body<REC, KS, MODE, true>(A, smem, bc0, bext, s2 < 0);
        if (s1 < 0) {
            if (!again) break;
            s1 = 0;
        } else if (s2 < 0) {
            if (again)
                s2 = 0;
            else
                s1++;
        } else if (++s2 == 8) {
            s2 = -1;
            s1++;
        }
        if (s1 == 8) break;
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
        case 34: launch_tile<REC, 34, SPLIT>(mode, nblocks, nw, lds, st, A); break;
        default: launch_tile<REC, 52, SPLIT>(mode, nblocks, nw, lds, st, A); break;
    }
}

}  // namespace
