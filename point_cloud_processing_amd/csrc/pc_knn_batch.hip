// Exact kNN for a LIST of open queries of one index level: the coarser passes of a multi-level
// index, where only a few points per tile are still unanswered and one workgroup per tile would run
// nearly empty waves.  The open queries, in the level's cell order, are dealt 64 to a wave; lane =
// query.  Nothing is staged: a lane reads the records of its own 3x3x3 cells from global memory (L2
// serves the neighbours' re-reads), in lock-step with its wave over the nine (y, z) row segments.
// Every squared distance is computed in fp64 in the reference's operation order; its fp32 rounding
// (monotone) with the low bits replaced by the candidate's ordinal is a unique u32 key.  A sorted
// register list of KS = k + 2 keys (one v_med3_u32 per slot) takes only keys below its last entry,
// which are parked in a lane-private LDS column and inserted in batches -- the tile kernel's
// selection, fed from global memory.  The KS survivors are re-measured, ordered by (d2, original
// index) and certified against the distance to the unscanned cells and to the unlisted candidates
// exactly like in the tile kernel.  Uncertified queries get (or keep) their flag for the next level.
#include <algorithm>

#include "pc_knn_dev.h"

namespace {

constexpr uint32_t KEY_NONE = 0xFFFFFFFFu;
constexpr int ORD_BITS = 11;   // a lane's candidates are numbered below 2^11 (crowded queries are left out)
constexpr uint32_t ORD_MASK = (1u << ORD_BITS) - 1;
constexpr int PARK_SLOTS = 16; // parked keys per lane between two sorted insertions
constexpr int AHEAD = 4;       // candidate records in flight per lane

template <int KS>
__device__ inline void key_insert(uint32_t (&L)[KS], uint32_t key) {
#pragma unroll
    for (int s = KS - 1; s > 0; s--) asm("v_med3_u32 %0, %1, %2, %0" : "+v"(L[s]) : "v"(L[s - 1]), "v"(key));
    asm("v_min_u32 %0, %0, %1" : "+v"(L[0]) : "v"(key));
}

__device__ inline int wave_max(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return __builtin_amdgcn_readfirstlane(v);
}

template <typename REC, int KS, int MODE>
__global__ __launch_bounds__(PC_WAVE) void k_batch_scan(ScanArgs A, const int32_t *__restrict__ list, uint32_t count) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // lane-private columns: parked keys during the filter, then the ordered result (same memory)
    uint32_t *park = reinterpret_cast<uint32_t *>(smem);             // [PARK_SLOTS + 2][64]
    double *bd = reinterpret_cast<double *>(smem);                   // [KS][64] exact squared distances
    uint32_t *bp = reinterpret_cast<uint32_t *>(bd + KS * PC_WAVE);  // [KS][64] sorted positions
    const int lane = threadIdx.x;
    const int k = A.k;
    const GridParams &gp = A.gp;
    const REC *rec = reinterpret_cast<const REC *>(A.rec);
    const REC *qsrc = A.qrec ? reinterpret_cast<const REC *>(A.qrec) : rec;
    const uint32_t qi = blockIdx.x * PC_WAVE + lane;
    const bool active = qi < count;
    double qx = 0.0, qy = 0.0, qz = 0.0;
    int32_t qid = 0;
    // (no list: every sorted position of the caller's range is a query -- the first pass of a thin cloud)
    if (active) RecTraits<REC>::load(qsrc + (list ? (int64_t)list[qi] : A.s_begin + (int64_t)qi), qx, qy, qz, qid);
    const double qv[3] = {qx, qy, qz};
    int c[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const double t = floor((qv[a] - gp.o[a]) * gp.s);
        c[a] = !(t > 0.0) ? 0 : (t >= (double)gp.dim[a] ? gp.dim[a] - 1 : (int)t);
    }
    const int x0 = max(c[0] - 1, 0), x1 = min(c[0] + 1, gp.dim[0] - 1);
    // the nine row segments of this lane (own row first) and the wave-wide longest of each
    constexpr int SEG_DY[9] = {0, -1, 1, 0, 0, -1, 1, -1, 1};
    constexpr int SEG_DZ[9] = {0, 0, 0, -1, 1, -1, -1, 1, 1};
    uint32_t jb[9], jn[9];
    int tmax[9];
#pragma unroll
    for (int seg = 0; seg < 9; seg++) {
        const int y = c[1] + SEG_DY[seg], z = c[2] + SEG_DZ[seg];
        jb[seg] = 0;
        jn[seg] = 0;
        if (active && y >= 0 && y < gp.dim[1] && z >= 0 && z < gp.dim[2]) {
            const int64_t row = ((int64_t)z * gp.dim[1] + y) * gp.dim[0];
            uint32_t je;
            cell_range(A.ci, row + x0, row + x1 + 1, jb[seg], je);
            jn[seg] = je - jb[seg];
        }
    }
    // A query amid more candidates than a tile pass would stage is not worth a lane here either: it
    // keeps its flag (coarser cells are fuller still, so it ends with the wave-per-query kernel).
    uint32_t total = 0;
    uint8_t wait_flag = 1;  // the flag a query left open gets: 1 + the first level that lists it again
#pragma unroll
    for (int seg = 0; seg < 9; seg++) total += jn[seg];
    const bool crowded = total > (uint32_t)min(A.cand_cap, (int)ORD_MASK);
    if (crowded) {
#pragma unroll
        for (int seg = 0; seg < 9; seg++) jn[seg] = 0;
    }
    // ... and one with fewer than 2 k candidates in its 27 cells will not find k of them inside the reach of
    // those cells (a ball inscribed in the block holds a fifth of an evenly filled block's points, under
    // half of a surface's): it waits for coarser cells without costing the wave any steps.  (Not when the
    // 27 cells are the whole grid or a radius limits the search: few neighbours are the answer then.)
    {
        bool whole_grid = true;
#pragma unroll
        for (int a = 0; a < 3; a++) whole_grid = whole_grid && c[a] <= 1 && c[a] + 1 >= gp.dim[a] - 1;
        if (total < 2u * (uint32_t)k && !whole_grid && A.radius2 < 0.0) {
#pragma unroll
            for (int seg = 0; seg < 9; seg++) jn[seg] = 0;
            // The 27 cells of the next level hold about eight times as many where the cloud fills space and four
            // times as many on a surface.  Filling space, they must hold some 4 k before k of them lie within
            // reach (the ball is a sixth of the block, not a third as the disc on a surface): the query skips
            // the levels where eight-fold growth stays below 4 k -- which on a surface is four-fold growth
            // below 2 k.  (Where a denser part of the cloud comes into reach sooner, a level is missed; the
            // passes after it, or the wave-per-query kernel, still answer exactly.)
            if (A.flag_top > 0) {
                uint32_t pred = max(total, 1u) * 8u;
                int f = A.flag_max + 1;
                while (pred < 4u * (uint32_t)k && f < A.flag_top) {
                    pred *= 8u;
                    f++;
                }
                wait_flag = (uint8_t)f;
            }
        }
    }
    if (A.count_crowded) {  // how many queries are crowded decides whether finer grids are worth making
        const unsigned long long m = __ballot(crowded);
        if (lane == 0 && m) atomicAdd(A.flag_count + 26, (uint32_t)__popcll(m));
    }
#pragma unroll
    for (int seg = 0; seg < 9; seg++) tmax[seg] = wave_max((int)jn[seg]);

    auto dist2 = [&](uint32_t j, int32_t &pid) {
        double px, py, pz;
        RecTraits<REC>::load(rec + j, px, py, pz, pid);
        const double dx = qx - px, dy = qy - py, dz = qz - pz;
        double d = dx * dx;
        d += dy * dy;
        d += dz * dz;
        return d;
    };

    // ---- filter: one pass, keys below the list's last entry parked and inserted in batches -------------------
    // key = (exact fp64 distance rounded to fp32, low ORD bits cleared) | ordinal of the candidate among
    // this lane's candidates: unique, and monotone in the distance up to the cleared bits
    uint32_t base[9];
    {
        uint32_t acc = 0;
#pragma unroll
        for (int seg = 0; seg < 9; seg++) {
            base[seg] = acc;
            acc += jn[seg];
        }
    }
    uint32_t L[KS];
#pragma unroll
    for (int s = 0; s < KS; s++) L[s] = KEY_NONE;
    uint32_t thr = KEY_NONE;
    int parked = 0;
    auto drain = [&]() {
        int s = 0;
        do {
            const uint32_t key = s < parked ? park[s * PC_WAVE + lane] : KEY_NONE;
            key_insert<KS>(L, key);
            s++;
        } while (__any(s < parked));
        thr = L[KS - 1];
        parked = 0;
    };
    // AHEAD candidates per turn: their records are requested back to back, with no branch around the loads (a
    // lane past the end of its segment re-reads the segment's last record -- record 0 if it has none -- and
    // drops it), so that the wave waits for memory once per turn, not once per candidate
#pragma unroll
    for (int seg = 0; seg < 9; seg++) {
        const uint32_t n = jn[seg];
        const uint32_t jlast = n ? jb[seg] + n - 1u : 0u;
#pragma nounroll
        for (int t = 0; t < tmax[seg]; t += AHEAD) {
            double d[AHEAD];
#pragma unroll
            for (int u = 0; u < AHEAD; u++) {
                int32_t pid;
                d[u] = dist2(n ? min(jb[seg] + (uint32_t)(t + u), jlast) : 0u, pid);
            }
#pragma unroll
            for (int u = 0; u < AHEAD; u++) {
                if ((uint32_t)(t + u) < n) {
                    const uint32_t key = (__float_as_uint((float)d[u]) & ~ORD_MASK) | (base[seg] + (uint32_t)(t + u));
                    if (key < thr) {
                        park[parked * PC_WAVE + lane] = key;
                        parked++;
                    }
                }
            }
            if (__any(parked > PARK_SLOTS - AHEAD)) drain();
        }
    }
    drain();
    if (!active) return;
    if (crowded) {
        if (!A.qflag_in) A.qub_arr[qid] = INFINITY;  // first pass: flags and bounds are not initialised yet
        if (A.flag_top > 0)
            A.qflag_out[qid] = FLAG_CROWDED;  // (coarser cells are fuller still: no later level lists it)
        else if (!A.qflag_in)
            A.qflag_out[qid] = 1;
        return;
    }

    // ---- exact finish on the KS survivors -----------------------------------------------------------------
    // no unscanned point is closer than sqrt(bound) ...
    bool whole = true;
    double dmin = INFINITY;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const int nlo = max(c[a] - 1, 0), nhi = min(c[a] + 1, gp.dim[a] - 1);
        if (nlo > 0) {
            whole = false;
            dmin = fmin(dmin, (qv[a] - gp.o[a]) - (double)nlo * gp.cell);  // (in offsets from the origin, like the cell keys)
        }
        if (nhi < gp.dim[a] - 1) {
            whole = false;
            dmin = fmin(dmin, (double)(nhi + 1) * gp.cell - (qv[a] - gp.o[a]));
        }
    }
    double bound = INFINITY;
    if (!whole) {
        const double b = dmin - 1e-7 * gp.cell;
        bound = b > 0 ? b * b : 0.0;
    }
    // ... and no unlisted candidate of the scanned cells is closer than sqrt(T * (1 - 2^-20))
    if (L[KS - 1] != KEY_NONE) bound = fmin(bound, (double)__uint_as_float(L[KS - 1] & ~ORD_MASK) * (1.0 - 9.5367431640625e-07));
    const double r2lim = A.radius2;
    double dd[KS];
    uint32_t jj[KS];
    int nsafe = 0, cnt = 0, nvalid = 0;
    double dmax = 0.0;
    // (the survivors' records likewise: AHEAD loads in flight, empty slots read record 0 and drop it)
#pragma unroll
    for (int s = 0; s < KS; s++) {
        dd[s] = INFINITY;
        jj[s] = 0;
    }
#pragma unroll
    for (int s0 = 0; s0 < KS; s0 += AHEAD) {
        double dv[AHEAD];
        uint32_t jv[AHEAD];
#pragma unroll
        for (int u = 0; u < AHEAD; u++) {
            jv[u] = 0;
            if (s0 + u < KS) {
                const uint32_t o = L[s0 + u] & ORD_MASK;
                uint32_t j = 0;
#pragma unroll
                for (int seg = 0; seg < 9; seg++)
                    if (o >= base[seg]) j = jb[seg] + (o - base[seg]);
                jv[u] = L[s0 + u] != KEY_NONE ? j : 0u;
            }
        }
#pragma unroll
        for (int u = 0; u < AHEAD; u++) {
            int32_t pid;
            dv[u] = dist2(jv[u], pid);
        }
#pragma unroll
        for (int u = 0; u < AHEAD; u++) {
            if (s0 + u < KS && L[s0 + u] != KEY_NONE) {
                const int s = s0 + u;
                const double d = dv[u];
                jj[s] = jv[u];
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
    }
    const bool ok = nsafe >= k || bound == INFINITY || (r2lim >= 0.0 && bound >= r2lim);
    if (!ok) {
        const float ubf = nvalid >= k ? (float)(dmax * (1.0 + 1e-6)) : INFINITY;
        A.qub_arr[qid] = A.qflag_in ? fminf(A.qub_arr[qid], ubf) : ubf;
        A.qflag_out[qid] = wait_flag;
        return;  // flagged for the next level
    }
    // exact order: distance (bubble passes of min/max exchanges; the list is nearly sorted), then the rare
    // equal distances by original index
    auto orig_index = [&](uint32_t j) {
        double x, y, z;
        int32_t id;
        RecTraits<REC>::load(rec + j, x, y, z, id);
        return id;
    };
#pragma nounroll
    for (int pass = 0; pass < KS; pass++) {
        bool swapped = false;
#pragma unroll
        for (int s = 0; s + 1 < KS; s++) {
            const double da = dd[s], db = dd[s + 1];
            const uint32_t ja = jj[s], jb2 = jj[s + 1];
            const bool gt = da > db;
            dd[s] = fmin(da, db);
            dd[s + 1] = fmax(da, db);
            jj[s] = gt ? jb2 : ja;
            jj[s + 1] = gt ? ja : jb2;
            swapped |= gt;
        }
        if (!__any(swapped)) break;
    }
    bool ties = false;
#pragma unroll
    for (int s = 0; s + 1 < KS; s++) ties |= dd[s] == dd[s + 1] && dd[s] != INFINITY;
    if (__any(ties)) {
#pragma nounroll
        for (int pass = 0; pass < KS; pass++) {
            bool swapped = false;
#pragma unroll
            for (int s = 0; s + 1 < KS; s++) {
                const uint32_t ja = jj[s], jb2 = jj[s + 1];
                bool gt = false;
                if (dd[s] == dd[s + 1] && dd[s] != INFINITY) gt = orig_index(ja) > orig_index(jb2);
                jj[s] = gt ? jb2 : ja;
                jj[s + 1] = gt ? ja : jb2;
                swapped |= gt;
            }
            if (!__any(swapped)) break;
        }
    }
#pragma unroll
    for (int s = 0; s < KS; s++) {
        bd[s * PC_WAVE + lane] = dd[s];
        bp[s * PC_WAVE + lane] = jj[s];
    }
    A.qflag_out[qid] = 0;
    cnt = min(cnt, k);
    emit_result<MODE>(A, A.out_slot ? (int64_t)A.out_slot[qid] : (int64_t)qid, cnt, [&](int s, double &d, double &x, double &y, double &z, int32_t &id) {
        d = bd[s * PC_WAVE + lane];
        RecTraits<REC>::load(rec + bp[s * PC_WAVE + lane], x, y, z, id);
    });
}

template <typename REC, int KS>
void launch_mode(int mode, unsigned nb, size_t lds, hipStream_t st, const ScanArgs &A, const int32_t *list, uint32_t count) {
    switch (mode) {
        case M_KNN: hipLaunchKernelGGL((k_batch_scan<REC, KS, M_KNN>), dim3(nb), dim3(PC_WAVE), lds, st, A, list, count); break;
        case M_SOR: hipLaunchKernelGGL((k_batch_scan<REC, KS, M_SOR>), dim3(nb), dim3(PC_WAVE), lds, st, A, list, count); break;
        case M_NORMALS: hipLaunchKernelGGL((k_batch_scan<REC, KS, M_NORMALS>), dim3(nb), dim3(PC_WAVE), lds, st, A, list, count); break;
        case M_NNDIST: hipLaunchKernelGGL((k_batch_scan<REC, KS, M_NNDIST>), dim3(nb), dim3(PC_WAVE), lds, st, A, list, count); break;
        case M_ALL: hipLaunchKernelGGL((k_batch_scan<REC, KS, M_ALL>), dim3(nb), dim3(PC_WAVE), lds, st, A, list, count); break;
        default: break;
    }
}

template <typename REC>
void launch_ks(int ks, int mode, unsigned nb, hipStream_t st, const ScanArgs &A, const int32_t *list, uint32_t count) {
    const size_t lds = std::max((sizeof(double) + sizeof(uint32_t)) * PC_WAVE * (size_t)ks, sizeof(uint32_t) * PC_WAVE * (size_t)(PARK_SLOTS + 2));
    switch (ks) {
        case 4: launch_mode<REC, 4>(mode, nb, lds, st, A, list, count); break;
        case 10: launch_mode<REC, 10>(mode, nb, lds, st, A, list, count); break;
        case 18: launch_mode<REC, 18>(mode, nb, lds, st, A, list, count); break;
        case 22: launch_mode<REC, 22>(mode, nb, lds, st, A, list, count); break;
        default: launch_mode<REC, 34>(mode, nb, lds, st, A, list, count); break;
    }
}

// flags[s] = 1 where the record at sorted position s (of this level) belongs to an open query that the pass
// of flag_max answers
// (*open_all counts every open query, also those that wait for a later pass; a thread takes OPEN_PER_THREAD
// records a block-width apart so that the counter sees one atomic per wave and 1024 records)
constexpr int OPEN_PER_THREAD = 16;
template <typename REC>
__global__ __launch_bounds__(256) void k_open_flags(const REC *__restrict__ rec, int64_t n, const uint8_t *__restrict__ qflag,
                                                     int flag_max, uint8_t *__restrict__ out, uint32_t *__restrict__ open_all) {
    const int64_t base = (int64_t)blockIdx.x * (256 * OPEN_PER_THREAD) + threadIdx.x;
    uint32_t mine = 0;
#pragma unroll 4
    for (int u = 0; u < OPEN_PER_THREAD; u++) {
        const int64_t s = base + (int64_t)u * 256;
        if (s < n) {
            double x, y, z;
            int32_t id;
            RecTraits<REC>::load(rec + s, x, y, z, id);
            const uint8_t f = qflag[id];
            out[s] = flag_live(f, flag_max) ? 1 : 0;
            mine += f != 0 ? 1u : 0u;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
    if (mine && (threadIdx.x & (PC_WAVE - 1)) == 0) atomicAdd(open_all, mine);
}

}  // namespace

// The open queries (flag set and at most flag_max) of the level whose records are `qsrc` (n of them, the
// level's own points or external queries binned into it), as ascending sorted positions in list_out;
// *count of them and *open_all, the number of flags set at all, on the host (synchronises).  counter: one
// device word of the caller's.
int pc_batch_open_list(pc_ctx *ctx, int dtype, const void *qsrc, int64_t n, const uint8_t *qflag, int flag_max, int32_t *list_out,
                       int64_t *count, int64_t *open_all, uint32_t *counter, hipStream_t st) {
    *count = 0;
    *open_all = 0;
    if (n <= 0) return PC_OK;
    PC_HIP(hipMemsetAsync(counter, 0, sizeof(uint32_t), st));
    Scratch f(ctx, (size_t)n);
    if (!f.ok()) return pc_fail(ctx, PC_NOMEM, "open-query flags");
    const unsigned nb = (unsigned)((n + 256 * OPEN_PER_THREAD - 1) / (256 * OPEN_PER_THREAD));
    if (dtype == PC_F32)
        hipLaunchKernelGGL(k_open_flags<Rec32>, dim3(nb), dim3(256), 0, st, (const Rec32 *)qsrc, n, qflag, flag_max, f.as<uint8_t>(), counter);
    else
        hipLaunchKernelGGL(k_open_flags<Rec64>, dim3(nb), dim3(256), 0, st, (const Rec64 *)qsrc, n, qflag, flag_max, f.as<uint8_t>(), counter);
    PC_HIP(hipGetLastError());
    uint32_t *h = reinterpret_cast<uint32_t *>(ctx->pinned) + 32;  // (pc_compact_flags reads its count back through the first words)
    PC_HIP(hipMemcpyAsync(h, counter, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    PC_TRY(pc_compact_flags(ctx, f.as<uint8_t>(), n, list_out, count, st));  // (synchronises)
    *open_all = h[0];
    return PC_OK;
}

// One level pass over the listed open queries (A describes the level; A.qflag_out / A.qub_arr are set).
int pc_launch_batch_scan(pc_ctx *ctx, int dtype, int ks, int mode, const ScanArgs &A, const int32_t *list, int64_t count,
                         hipStream_t st) {
    if (count <= 0) return PC_OK;
    const unsigned nb = (unsigned)((count + PC_WAVE - 1) / PC_WAVE);
    if (dtype == PC_F32)
        launch_ks<Rec32>(ks, mode, nb, st, A, list, (uint32_t)count);
    else
        launch_ks<Rec64>(ks, mode, nb, st, A, list, (uint32_t)count);
    PC_HIP(hipGetLastError());
    return PC_OK;
}

