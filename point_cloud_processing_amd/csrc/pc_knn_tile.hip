// Tile-cooperative exact kNN for fp32 grids (self-queries, k <= 32).
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
#include "pc_knn_dev.h"

namespace {

constexpr int ORD_BITS = 10;
constexpr int CAND_CAP = 1 << ORD_BITS;  // candidates one tile may stage (16 KiB of LDS)
constexpr uint32_t ORD_MASK = CAND_CAP - 1;
constexpr uint32_t KEY_EMPTY = 0xFFFFFFFFu;
constexpr int MAX_ROWS = 64;  // (tile_y + 2) * (tile_z + 2) must fit one wave

__device__ inline uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

template <int KS>
__device__ inline void list_insert(uint32_t (&L)[KS], uint32_t key) {
#pragma unroll
    for (int s = KS - 1; s > 0; s--) L[s] = umed3(L[s - 1], key, L[s]);
    L[0] = min(L[0], key);
}

template <int KS, int MODE>
__global__ __launch_bounds__(PC_WAVE) void k_tile_scan(ScanArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4 *cand = reinterpret_cast<float4 *>(smem);                                          // [CAND_CAP]
    uint16_t *ordbuf = reinterpret_cast<uint16_t *>(smem + sizeof(float4) * CAND_CAP);        // [k][64]
    const int lane = threadIdx.x;
    const GridParams &gp = A.gp;
    const Rec32 *rec = reinterpret_cast<const Rec32 *>(A.rec);
    const int k = A.k;

    // tile coordinates (x fastest so that neighbouring blocks share L2 lines)
    int t = blockIdx.x;
    const int tx = t % A.ntile[0];
    t /= A.ntile[0];
    const int ty = t % A.ntile[1];
    const int tz = t / A.ntile[1];
    const int c0[3] = {tx * A.tile[0], ty * A.tile[1], tz * A.tile[2]};  // first cell of the tile
    int c1[3], lo[3], hi[3];                                              // last cell; block bounds
#pragma unroll
    for (int a = 0; a < 3; a++) {
        c1[a] = min(c0[a] + A.tile[a], gp.dim[a]) - 1;
        lo[a] = max(c0[a] - 1, 0);
        hi[a] = min(c1[a] + 1, gp.dim[a] - 1);
    }
    const int nry = hi[1] - lo[1] + 1, nrz = hi[2] - lo[2] + 1;
    const int nrows = nry * nrz;  // <= MAX_ROWS by construction of the tile shape

    // ---- row table: lane r describes block row r -------------------------------------------------
    uint32_t row_begin = 0, row_cnt = 0, q_begin = 0, q_cnt = 0;
    if (lane < nrows) {
        const int y = lo[1] + lane % nry, z = lo[2] + lane / nry;
        const int64_t row = ((int64_t)z * gp.dim[1] + y) * gp.dim[0];
        row_begin = A.cell_start[row + lo[0]];
        row_cnt = A.cell_start[row + hi[0] + 1] - row_begin;
        if (y >= c0[1] && y <= c1[1] && z >= c0[2] && z <= c1[2]) {  // a row of the tile itself
            q_begin = A.cell_start[row + c0[0]];
            q_cnt = A.cell_start[row + c1[0] + 1] - q_begin;
            // restrict to the caller's sorted-position range
            const int64_t b = max((int64_t)q_begin, A.s_begin), e = min((int64_t)q_begin + q_cnt, A.s_end);
            q_begin = (uint32_t)b;
            q_cnt = e > b ? (uint32_t)(e - b) : 0u;
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
    const uint32_t ncand = __shfl(cand_off, 63, 64), nquery = __shfl(q_off, 63, 64);
    if (nquery == 0) return;  // wave-uniform
    cand_off -= row_cnt;      // exclusive
    q_off -= q_cnt;

    const bool overfull = ncand > (uint32_t)CAND_CAP;

    // ---- stage the candidate block into LDS (row by row, coalesced 16-byte records) ---------------
    if (!overfull) {
        for (int r = 0; r < nrows; r++) {
            const uint32_t rb = __shfl(row_begin, r, 64), rc = __shfl(row_cnt, r, 64), ro = __shfl(cand_off, r, 64);
            for (uint32_t i = lane; i < rc; i += PC_WAVE)
                cand[ro + i] = *reinterpret_cast<const float4 *>(rec + rb + i);
        }
    }
    __syncthreads();

    // certification geometry shared by all lanes: faces of the block that are not grid faces
    const double eps = 1e-7 * gp.cell;
    const double r2lim = A.radius2;

    for (uint32_t qbase = 0; qbase < nquery; qbase += PC_WAVE) {
        const uint32_t qo = qbase + lane;
        const bool active = qo < nquery;
        // which tile row holds query ordinal qo
        uint32_t spos = 0;
        for (int r = 0; r < nrows; r++) {  // shuffles stay outside divergent control flow
            const uint32_t ob = __shfl(q_off, r, 64), oc = __shfl(q_cnt, r, 64), gb = __shfl(q_begin, r, 64);
            if (active && qo >= ob && qo < ob + oc) spos = gb + (qo - ob);
        }
        if (overfull) {
            if (active) {
                const uint32_t p = atomicAdd(A.flag_count, 1u);
                if (p < A.flag_cap) A.flag_list[p] = (int32_t)spos;
            }
            continue;
        }
        float4 q4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (active) q4 = *reinterpret_cast<const float4 *>(rec + spos);
        const float qx = q4.x, qy = q4.y, qz = q4.z;

        uint32_t L[KS];
#pragma unroll
        for (int s = 0; s < KS; s++) L[s] = KEY_EMPTY;

        // ---- stage 1: fp32 filter over the shared candidate sequence ------------------------------
        uint32_t j = 0;
        for (; j + 4 <= ncand; j += 4) {
            const float4 a0 = cand[j], a1 = cand[j + 1], a2 = cand[j + 2], a3 = cand[j + 3];
#define PC_KEY(a, jj)                                                                                   \
    ((__float_as_uint(fmaf(qz - a.z, qz - a.z, fmaf(qy - a.y, qy - a.y, (qx - a.x) * (qx - a.x)))) & ~ORD_MASK) | \
     (jj))
            const uint32_t k0 = PC_KEY(a0, j), k1 = PC_KEY(a1, j + 1), k2 = PC_KEY(a2, j + 2), k3 = PC_KEY(a3, j + 3);
            list_insert<KS>(L, k0);
            list_insert<KS>(L, k1);
            list_insert<KS>(L, k2);
            list_insert<KS>(L, k3);
        }
        for (; j < ncand; j++) {
            const float4 a0 = cand[j];
            list_insert<KS>(L, PC_KEY(a0, j));
        }
#undef PC_KEY
        if (!active) continue;

        // ---- stage 2: exact finish --------------------------------------------------------------------
        const double qxd = (double)qx, qyd = (double)qy, qzd = (double)qz;
        const double qv[3] = {qxd, qyd, qzd};
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
        double bound = INFINITY;  // no unscanned point is closer than sqrt(bound)
        if (!whole) {
            const double b = dmin - eps;
            bound = b > 0 ? b * b : 0.0;
        }
        // no unlisted candidate of the block is closer than sqrt(T * (1 - 2^-20))
        if (L[KS - 1] != KEY_EMPTY)
            bound = fmin(bound, (double)__uint_as_float(L[KS - 1] & ~ORD_MASK) * (1.0 - 9.5367431640625e-07));

        double dd[KS];
        int32_t id[KS];
        int nsafe = 0;
#pragma unroll
        for (int s = 0; s < KS; s++) {
            dd[s] = INFINITY;
            id[s] = 0x7fffffff;
            if (L[s] != KEY_EMPTY) {
                const float4 c = cand[L[s] & ORD_MASK];
                const double dx = qxd - (double)c.x, dy = qyd - (double)c.y, dz = qzd - (double)c.z;
                double d = dx * dx;
                d += dy * dy;
                d += dz * dz;
                if (d < bound) {
                    nsafe++;
                    if (!(r2lim >= 0.0 && !(d < r2lim))) {
                        dd[s] = d;
                        id[s] = __float_as_int(c.w);
                    }
                }
            }
        }
        const bool ok = nsafe >= k || bound == INFINITY || (r2lim >= 0.0 && bound >= r2lim);
        if (!ok) {
            const uint32_t p = atomicAdd(A.flag_count, 1u);
            if (p < A.flag_cap) A.flag_list[p] = (int32_t)spos;
            continue;
        }
        // rank the accepted entries by (d2, original index); rank r < k goes to ordbuf[r][lane]
        int cnt = 0;
#pragma unroll
        for (int s = 0; s < KS; s++) {
            if (id[s] != 0x7fffffff) {
                int rank = 0;
#pragma unroll
                for (int u = 0; u < KS; u++)
                    rank += (dd[u] < dd[s] || (dd[u] == dd[s] && id[u] < id[s])) ? 1 : 0;
                if (rank < k) ordbuf[rank * PC_WAVE + lane] = (uint16_t)(L[s] & ORD_MASK);
                cnt++;
            }
        }
        cnt = min(cnt, k);
        emit_result<MODE>(A, (int64_t)__float_as_int(q4.w), cnt,
                          [&](int s, double &d, double &x, double &y, double &z, int32_t &nid) {
                              const float4 c = cand[ordbuf[s * PC_WAVE + lane]];
                              x = (double)c.x;
                              y = (double)c.y;
                              z = (double)c.z;
                              const double dx = qxd - x, dy = qyd - y, dz = qzd - z;
                              d = dx * dx;
                              d += dy * dy;
                              d += dz * dz;
                              nid = __float_as_int(c.w);
                          });
    }
}

template <int KS>
void launch_tile(int mode, unsigned nblocks, size_t lds, hipStream_t st, const ScanArgs &A) {
    switch (mode) {
        case M_KNN: hipLaunchKernelGGL((k_tile_scan<KS, M_KNN>), dim3(nblocks), dim3(PC_WAVE), lds, st, A); break;
        case M_SOR: hipLaunchKernelGGL((k_tile_scan<KS, M_SOR>), dim3(nblocks), dim3(PC_WAVE), lds, st, A); break;
        case M_NORMALS:
            hipLaunchKernelGGL((k_tile_scan<KS, M_NORMALS>), dim3(nblocks), dim3(PC_WAVE), lds, st, A);
            break;
        case M_NNDIST:
            hipLaunchKernelGGL((k_tile_scan<KS, M_NNDIST>), dim3(nblocks), dim3(PC_WAVE), lds, st, A);
            break;
        default: break;
    }
}

}  // namespace

int pc_launch_tile_scan(pc_ctx *ctx, const pc_grid *g, int mode, ScanArgs &A, hipStream_t st, bool *handled) {
    *handled = false;
    if (g->dtype != PC_F32 || A.queries != nullptr || A.k > 32 || mode == M_COUNT || A.nq <= 0) return PC_OK;
    if (getenv("PC_AMD_EXACT_SCAN_ONLY")) return PC_OK;  // A/B switch: one-query-per-lane fp64 scan only
    // tile shape: grow z, y, x in turn while the expected tile population stays within one wave
    const double occ = (double)g->n / (double)std::max<int64_t>(1, g->ncell);
    int tile[3] = {1, 1, 1};
    const int order[3] = {2, 1, 0};
    for (bool grew = true; grew;) {
        grew = false;
        for (int oi = 0; oi < 3; oi++) {
            const int a = order[oi];
            int t2[3] = {tile[0], tile[1], tile[2]};
            t2[a]++;
            if (t2[a] > g->gp.dim[a]) continue;
            if ((t2[1] + 2) * (t2[2] + 2) > MAX_ROWS) continue;
            if ((double)t2[0] * t2[1] * t2[2] * occ > 54.0) continue;
            tile[a] = t2[a];
            grew = true;
        }
    }
    int64_t nt = 1;
    for (int a = 0; a < 3; a++) {
        A.tile[a] = tile[a];
        A.ntile[a] = (g->gp.dim[a] + tile[a] - 1) / tile[a];
        nt *= A.ntile[a];
    }
    if (nt > 0x7fffffff) return PC_OK;
    A.s_end = A.s_begin + A.nq;
    A.flag_cap = (uint32_t)A.nq;
    Scratch flags(ctx, sizeof(int32_t) * (size_t)A.nq + 256);
    if (!flags.ok()) return pc_fail(ctx, PC_NOMEM, "re-run list");
    A.flag_count = flags.as<uint32_t>();
    A.flag_list = reinterpret_cast<int32_t *>(flags.as<char>() + 256);
    PC_HIP(hipMemsetAsync(A.flag_count, 0, 256, st));
    const size_t lds = sizeof(float4) * CAND_CAP + sizeof(uint16_t) * PC_WAVE * (size_t)A.k;
    const int ks = A.k <= 2 ? 4 : (A.k <= 8 ? 10 : (A.k <= 16 ? 18 : (A.k <= 20 ? 22 : 34)));
    switch (ks) {
        case 4: launch_tile<4>(mode, (unsigned)nt, lds, st, A); break;
        case 10: launch_tile<10>(mode, (unsigned)nt, lds, st, A); break;
        case 18: launch_tile<18>(mode, (unsigned)nt, lds, st, A); break;
        case 22: launch_tile<22>(mode, (unsigned)nt, lds, st, A); break;
        default: launch_tile<34>(mode, (unsigned)nt, lds, st, A); break;
    }
    PC_HIP(hipGetLastError());
    // the exact re-run answers whatever the tile kernel appended (device-side count, no host sync)
    A.qlist = A.flag_list;
    A.qcount = A.flag_count;
    *handled = true;
    return PC_OK;
}
