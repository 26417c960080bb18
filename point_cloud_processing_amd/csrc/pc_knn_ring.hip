// Exact kNN for the queries a multi-level ladder leaves open: points next to something dense -- too few
// candidates in the 27 cells of a fine level, more than a tile stages on the coarse ones (clutter a metre off a
// scanned wall).  Their k neighbours lie in a small ball that reaches into the dense part; what they need is the
// FINEST level, a ring of several cells and everything outside the ball cut away before it is read.
// Lane = query (the open queries in the finest level's cell order, 64 to a wave), so a wave's lanes walk the
// same rows: the rows of a ring of RING_MAX cells around the query's cell, where rows farther than the
// bound of the k-th distance are skipped and the others cut to the chord of the ball; every candidate is measured
// in fp64 from its record in global memory, its distance rounded to fp32 is the key, and keys below the bound
// and below the list's last entry are parked in a lane-private LDS column with the candidate's sorted position,
// then inserted into a sorted register list of KS = k + 2 (key, position) pairs.  The rows are walked in shells,
// nearest first; as soon as k candidates are listed their k-th distance is the bound, and a query is through
// when the shells walked hold the ball of its bound.  The survivors are
// re-measured, ordered by (d2, original index) and certified against the first unlisted key exactly like in
// the other kernels; a query that needs more than RING_MAX cells keeps its flag (and its better bound) for the
// wave-per-query kernel.
#include <algorithm>

#include "pc_knn_dev.h"

namespace {

constexpr uint32_t KEY_NONE = 0xFFFFFFFFu;
constexpr uint32_t KEY_BIAS = 0x00100000u;  // added to a key (the bits of a positive float) so that (key, position) read as fp64 is a normal number
constexpr uint32_t KEY_INF = 0x7FF00000u;   // the high half of +inf: an empty slot, and above every biased key
constexpr int PARK_SLOTS = 16;  // parked (key, position) pairs per lane between two sorted insertions
constexpr int AHEAD = 8;        // candidate records in flight per lane
constexpr int ROWS_AHEAD = 4;   // row look-ups in flight per lane
constexpr int LOOSE_ROWS = 4;   // ... while no bound cuts anything
constexpr int RING_MAX = 6;     // widest ring of cells a lane walks (169 rows) when it comes with a bound
constexpr int RING_FREE = 4;    // ... and when it has to find its bound on the way (its first rows are walked over this many cells either side)

template <typename REC, int KS, int MODE>
__global__ __launch_bounds__(PC_WAVE) __attribute__((amdgpu_waves_per_eu(2))) void k_ring_scan(ScanArgs A, const int32_t *__restrict__ list, uint32_t count,
                                                        uint32_t stride, uint32_t *__restrict__ answered) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t *park_k = reinterpret_cast<uint32_t *>(smem);           // [PARK_SLOTS][64]
    uint32_t *park_j = park_k + PARK_SLOTS * PC_WAVE;                // [PARK_SLOTS][64]
    double *bd = reinterpret_cast<double *>(smem);                   // [KS][64] exact squared distances (after the passes)
    uint32_t *bp = reinterpret_cast<uint32_t *>(bd + KS * PC_WAVE);  // [KS][64] sorted positions
    const int lane = threadIdx.x;
    const int k = A.k;
    const GridParams &gp = A.gp;
    const REC *rec = reinterpret_cast<const REC *>(A.rec);
    const uint32_t qi = blockIdx.x * PC_WAVE + lane;
    bool active = qi < count;  // (the launch takes every stride-th entry of the list)
    double qx = 0.0, qy = 0.0, qz = 0.0;
    int32_t qid = 0;
    if (active) RecTraits<REC>::load(rec + list[(uint64_t)qi * stride], qx, qy, qz, qid);
    if (active && A.qflag_out[qid] == 0) active = false;  // (answered by an earlier launch over a sample of the list)
    const double qv[3] = {qx, qy, qz};
    double u[3];
    int c[3];
    double edge = 0.5;  // distance to the nearest face of the query's own cell, in cells
#pragma unroll
    for (int a = 0; a < 3; a++) {
        u[a] = (qv[a] - gp.o[a]) * gp.s;
        const double t = floor(u[a]);
        c[a] = !(t > 0.0) ? 0 : (t >= (double)gp.dim[a] ? gp.dim[a] - 1 : (int)t);
        edge = fmin(edge, fmin(u[a] - (double)c[a], (double)(c[a] + 1) - u[a]));
    }
    edge = fmax(edge, 0.0);
    float e_lo[3], e_hi[3];  // the query's distance from the low / high face of its cell, in cells (never above the truth)
#pragma unroll
    for (int a = 0; a < 3; a++) {
        e_lo[a] = fmaxf((float)(u[a] - (double)c[a]) - 1e-6f, 0.f);
        e_hi[a] = fmaxf((float)((double)(c[a] + 1) - u[a]) - 1e-6f, 0.f);
    }
    // the ring whose block certainly holds the ball of squared radius b around the query (RING_MAX + 1: too wide)
    auto ring_for = [&](double b) {
        const double need = sqrt(b * (1.0 + 1e-6)) * gp.s - edge + 1e-9;
        return need > (double)RING_MAX ? RING_MAX + 1 : max(1, (int)ceil(need));
    };
    double ub = active ? (double)A.qub_arr[qid] : 0.0;  // upper bound of the k-th squared distance, or +inf
    const double ub_in = ub;
    // (A lane without a bound walks its first rows uncut, RING_FREE cells either side: its ball must fit that.  66 M voxels:
    // 6 cells for every lane took the pass from 25 to 30 ms; a slab of an eighth of them, whose list pass hands over
    // 0.46 M queries with bounds that need five cells, from 0.24 M to all of them answered here.)
    const int reach = ub < INFINITY ? RING_MAX : RING_FREE;
    bool live = active;  // still walking
    bool done = false;   // every point within `ub` was looked at
    // the list: (key + KEY_BIAS, sorted position) as the two halves of a positive, normal fp64 number, whose order is
    // the order of the pairs -- one v_min_f64 and one v_max_f64 per slot carry a pair to its place; empty slots are +inf
    double LD[KS];
#pragma unroll
    for (int s = 0; s < KS; s++) LD[s] = INFINITY;

    auto dist2_of = [&](const REC *from, uint32_t j) {
        double px, py, pz;
        int32_t pid;
        RecTraits<REC>::load(from + j, px, py, pz, pid);
        const double dx = qx - px, dy = qy - py, dz = qz - pz;
        double d = dx * dx;
        d += dy * dy;
        d += dz * dz;
        return d;
    };
    auto dist2 = [&](uint32_t j) { return dist2_of(rec, j); };

    // keys at or below the bound take part (fp32 rounding is monotone: d2 <= ub gives key <= the bound rounded up)
    uint32_t thr = !active ? 0u : (ub < INFINITY ? __float_as_uint(__double2float_ru(ub)) + 1u + KEY_BIAS : KEY_INF);
    uint32_t kth = KEY_INF;  // the k-th listed key
    uint32_t st_iter = 0, st_pick = 0, st_rows = 0, st_cand = 0, st_drain = 0, st_samp = 0, st_sbatch = 0;
    (void)st_samp, (void)st_sbatch, (void)st_iter, (void)st_pick, (void)st_rows, (void)st_cand, (void)st_drain;
    int parked = 0;
    bool tighter = false;  // a parked key lies below the k-th listed one: inserting it lowers the bound
    // parked pairs into the list; with k of them listed, their k-th distance is a bound (the key is the distance
    // rounded to nearest), and nothing beyond it needs to be parked again
    auto drain = [&]() {
        int s = 0;
        do {
            const bool have = s < parked;
            const uint32_t key = have ? park_k[s * PC_WAVE + lane] : KEY_INF;
            const uint32_t j = have ? park_j[s * PC_WAVE + lane] : 0u;
            st_drain++;
            double x = __hiloint2double((int)key, (int)j);
#pragma unroll
            for (int t = 0; t < KS; t++) {
                double lo;
                asm("v_min_f64 %0, %1, %2" : "=v"(lo) : "v"(LD[t]), "v"(x));
                asm("v_max_f64 %0, %1, %2" : "=v"(x) : "v"(LD[t]), "v"(x));
                LD[t] = lo;
            }
            s++;
        } while (__any(s < parked));
        parked = 0;
        tighter = false;
#pragma unroll
        for (int t = 0; t < KS; t++) kth = t == k - 1 ? (uint32_t)__double2hiint(LD[t]) : kth;
        if (kth != KEY_INF) {
            ub = fmin(ub, (double)__uint_as_float(kth - KEY_BIAS) * (1.0 + 1e-6));
            thr = min(thr, __float_as_uint(__double2float_ru(ub)) + 1u + KEY_BIAS);
        }
        thr = min(thr, (uint32_t)__double2hiint(LD[KS - 1]));
    };
    // Shells of rows around the query's own row, nearest first: shell s holds the rows with max(|dy|, |dz|) = s.
    // Once a bound is known a row beyond it costs nothing, the others are cut to the chord of the ball; before
    // that a row is walked over the widest ring's cells.  A lane is through when the shells walked hold the ball
    // of its bound.  Every lane walks at its own pace: it takes the next rows of its shell (up to ROWS_AHEAD that
    // are worth a look-up, their look-ups in flight together) when it has used up the candidates of the last
    // ones, so that a lane that meets a wall of a thousand candidates holds nobody up row by row: the wave runs
    // for its slowest lane's total, not for the sum of every row's slowest lane.
    int shell = 0, idx = 0;  // the next row of this lane: position idx on the perimeter of `shell`
    uint32_t sb[ROWS_AHEAD], sn[ROWS_AHEAD];  // the rows being used up: first sorted position and length
#pragma unroll
    for (int w = 0; w < ROWS_AHEAD; w++) {
        sb[w] = 0;
        sn[w] = 0;
    }
    uint32_t pos = 0, tot = 0;  // candidates of those rows used / in all
    while (__any(live)) {
        st_iter++;
        const bool need = live && pos >= tot;
        // The look-up of new rows costs the wave three times a round of candidates whoever takes part: it waits until
        // half of the walking lanes are out of candidates, or nobody has any left (66 M voxels, 2.3 M queries: 25.2 ->
        // 19.8 ms; a quarter / 40 % / 60 % / two thirds / all of the lanes: 22.0 / 20.7 / 19.7 / 19.8 / 28.4 ms).
        const int n_need = __popcll(__ballot(need)), n_live = __popcll(__ballot(live));
        const bool pick = 2 * n_need >= n_live || !__any(live && pos < tot);
        if (pick && __any(need && tighter)) drain();  // (rows are cut by the freshest bound)
        if (pick && need) {
            pos = 0;
            tot = 0;
            if (idx == 0 && shell > 0) {
                // shell - 1 is complete
                if (shell - 1 >= 1 && ub < INFINITY && ring_for(ub) <= shell - 1) {
                    done = true;
                    live = false;
                } else if (shell > reach) {
                    live = false;
                }
            }
            if (live) {
                st_pick++;
                const int nrows = shell == 0 ? 1 : 8 * shell, sd = 2 * shell;
                uint32_t key0[ROWS_AHEAD];  // (cell keys are below 2^32)
                int span[ROWS_AHEAD];
                int got = 0;
#pragma unroll
                for (int w = 0; w < ROWS_AHEAD; w++) {
                    key0[w] = 0;
                    span[w] = 0;
                }
                const bool bounded = ub < INFINITY;
                const bool tight = bounded && ring_for(ub) <= reach;
                // What is cut away is decided in fp32, in cells, on the generous side: the squared radius a ten-thousandth
                // wider, the chord a thousandth of a cell longer (the coordinates inside the ring are small numbers).
                const float ubc = bounded ? (float)(ub * gp.s * gp.s) * 1.0001f + 1e-4f : 0.f;
                const int xlo = max(c[0] - reach, 0), xhi = min(c[0] + reach, gp.dim[0] - 1);
                while (idx < nrows) {
                    if (got >= (tight ? ROWS_AHEAD : LOOSE_ROWS)) break;
                    int dy = 0, dz = 0;
                    if (shell > 0) {
                        const int q = (idx >= sd ? 1 : 0) + (idx >= 2 * sd ? 1 : 0) + (idx >= 3 * sd ? 1 : 0), t = idx - q * sd;
                        dy = q == 0 ? -shell + t : (q == 1 ? shell : (q == 2 ? shell - t : -shell));
                        dz = q == 0 ? -shell : (q == 1 ? -shell + t : (q == 2 ? shell : shell - t));
                    }
                    idx++;
                    st_rows++;
                    const int y = c[1] + dy, z = c[2] + dz;
                    if (y < 0 || y >= gp.dim[1] || z < 0 || z >= gp.dim[2]) continue;
                    int xa = xlo, xb = xhi;
                    if (bounded) {
                        // the row's distance from the query in y and z, then the chord of the ball along x
                        const float fy = dy == 0 ? 0.f : (float)(abs(dy) - 1) + (dy > 0 ? e_hi[1] : e_lo[1]);
                        const float fz = dz == 0 ? 0.f : (float)(abs(dz) - 1) + (dz > 0 ? e_hi[2] : e_lo[2]);
                        const float w2 = ubc - fy * fy - fz * fz;
                        if (w2 < 0.f) continue;
                        const float hw = __builtin_amdgcn_sqrtf(w2) + 1e-3f;
                        xa = max(xa, c[0] + (int)floorf(e_lo[0] - hw));
                        xb = min(xb, c[0] + (int)floorf(e_lo[0] + hw));
                    }
                    if (xa > xb) continue;
                    const uint32_t kk = (uint32_t)(((int64_t)z * gp.dim[1] + y) * gp.dim[0] + xa);
#pragma unroll
                    for (int w = 0; w < ROWS_AHEAD; w++) {
                        key0[w] = w == got ? kk : key0[w];
                        span[w] = w == got ? xb + 1 - xa : span[w];
                    }
                    got++;
                }
                if (idx >= nrows) {
                    shell++;
                    idx = 0;
                }
#pragma unroll
                for (int w = 0; w < ROWS_AHEAD; w++) {
                    sb[w] = 0;
                    sn[w] = 0;
                    if (w < got) {
                        uint32_t je;
                        cell_range(A.ci, (int64_t)key0[w], (int64_t)key0[w] + span[w], sb[w], je);
                        sn[w] = je - sb[w];
                    }
                    tot += sn[w];
                }
            }
        }
        if (__any(live && pos < tot)) {
            static_assert(ROWS_AHEAD == 4, "the position -> row mapping below is written out for four rows");
            const uint32_t c1 = sn[0], c2 = c1 + sn[1], c3 = c2 + sn[2];
            double d[AHEAD];
            uint32_t jv[AHEAD];
            // (no branch around the loads: a lane without that many candidates left reads record 0 and drops it)
#pragma unroll
            for (int v = 0; v < AHEAD; v++) {
                const uint32_t p = pos + (uint32_t)v;
                const uint32_t j = p < c1 ? sb[0] + p : (p < c2 ? sb[1] + (p - c1) : (p < c3 ? sb[2] + (p - c2) : sb[3] + (p - c3)));
                jv[v] = live && p < tot ? j : 0u;
                d[v] = dist2(jv[v]);
            }
#pragma unroll
            for (int v = 0; v < AHEAD; v++) {
                if (live && pos + (uint32_t)v < tot) {
                    st_cand++;
                    const uint32_t key = __float_as_uint((float)d[v]) + KEY_BIAS;
                    if (key < thr) {
                        park_k[parked * PC_WAVE + lane] = key;
                        park_j[parked * PC_WAVE + lane] = jv[v];
                        parked++;
                        tighter = tighter || key < kth;
                    }
                }
            }
            pos = min(pos + (uint32_t)AHEAD, tot);
            if (__any(parked > PARK_SLOTS - AHEAD)) drain();
        }
    }
    drain();
    uint32_t L[KS], J[KS];
#pragma unroll
    for (int s = 0; s < KS; s++) {
        const uint32_t hi = (uint32_t)__double2hiint(LD[s]);
        L[s] = hi == KEY_INF ? KEY_NONE : hi - KEY_BIAS;
        J[s] = hi == KEY_INF ? 0u : (uint32_t)__double2loint(LD[s]);
    }
#ifdef RING_STATS
    {
        auto wsum = [&](uint32_t v) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            return v;
        };
        const uint32_t a = wsum(st_pick), b = wsum(st_rows), cnd = wsum(st_cand), sm = wsum(st_samp);
        if (lane == 0) {
            atomicAdd(A.flag_count + 48, st_iter);    // outer iterations (per wave)
            atomicAdd(A.flag_count + 49, a);          // picks (lanes)
            atomicAdd(A.flag_count + 50, b);          // rows looked at (lanes)
            atomicAdd(A.flag_count + 51, cnd);        // candidates measured (lanes)
            atomicAdd(A.flag_count + 52, st_drain);   // insertion rounds (per wave)
            atomicAdd(A.flag_count + 53, sm);         // samples (lanes)
            atomicAdd(A.flag_count + 54, st_sbatch);  // sample batches (per wave)
            atomicAdd(A.flag_count + 55, 1u);
        }
    }
#endif
    if (!active) return;
    // ---- exact finish on the listed pairs ----------------------------------------------------------------
    // no unlisted candidate is closer than sqrt(T * (1 - 2^-20)), T the last key of a full list
    const double bound = L[KS - 1] != KEY_NONE ? (double)__uint_as_float(L[KS - 1]) * (1.0 - 9.5367431640625e-07) : INFINITY;
    double dd[KS];
    int nsafe = 0, nvalid = 0;
    double dmax = 0.0;
    if (done) {
#pragma unroll
        for (int s0 = 0; s0 < KS; s0 += AHEAD) {
            double dv[AHEAD];
#pragma unroll
            for (int v = 0; v < AHEAD; v++) dv[v] = s0 + v < KS ? dist2(J[s0 + v]) : 0.0;  // (empty slots read record 0)
#pragma unroll
            for (int v = 0; v < AHEAD; v++) {
                if (s0 + v < KS) {
                    const int s = s0 + v;
                    const bool valid = L[s] != KEY_NONE;
                    const bool safe = valid && dv[v] < bound;
                    nvalid += valid ? 1 : 0;
                    nsafe += safe ? 1 : 0;
                    dmax = valid ? fmax(dmax, dv[v]) : dmax;
                    dd[s] = safe ? dv[v] : INFINITY;
                }
            }
        }
    }
    const bool ok = done && nsafe >= k;
    if (answered) {  // (one atomic per wave)
        const unsigned long long m = __ballot(ok);
        if (ok && __ffsll((long long)m) - 1 == lane) atomicAdd(answered, (uint32_t)__popcll(m));
    }
    if (!ok) {
        // left for the wave-per-query kernel, with the best bound known
        double b = ub;
        if (done && nvalid >= k) b = fmin(b, dmax * (1.0 + 1e-6));
        if (b < ub_in) A.qub_arr[qid] = __double2float_ru(b);
        return;
    }
    const int cnt = k;
    auto orig_index = [&](uint32_t j) {
        double x, y, z;
        int32_t id;
        RecTraits<REC>::load(rec + j, x, y, z, id);
        return id;
    };
    // (lanes finish alone here: no wave-wide votes below)
    for (int pass = 0; pass < KS; pass++) {
        bool swapped = false;
#pragma unroll
        for (int s = 0; s + 1 < KS; s++) {
            const double da = dd[s], db = dd[s + 1];
            const uint32_t ja = J[s], jb2 = J[s + 1];
            const bool gt = da > db;
            dd[s] = fmin(da, db);
            dd[s + 1] = fmax(da, db);
            J[s] = gt ? jb2 : ja;
            J[s + 1] = gt ? ja : jb2;
            swapped |= gt;
        }
        if (!swapped) break;
    }
    bool ties = false;
#pragma unroll
    for (int s = 0; s + 1 < KS; s++) ties |= dd[s] == dd[s + 1] && dd[s] != INFINITY;
    if (ties) {
        for (int pass = 0; pass < KS; pass++) {
            bool swapped = false;
#pragma unroll
            for (int s = 0; s + 1 < KS; s++) {
                const uint32_t ja = J[s], jb2 = J[s + 1];
                bool gt = false;
                if (dd[s] == dd[s + 1] && dd[s] != INFINITY) gt = orig_index(ja) > orig_index(jb2);
                J[s] = gt ? jb2 : ja;
                J[s + 1] = gt ? ja : jb2;
                swapped |= gt;
            }
            if (!swapped) break;
        }
    }
#pragma unroll
    for (int s = 0; s < KS; s++) {
        bd[s * PC_WAVE + lane] = dd[s];
        bp[s * PC_WAVE + lane] = J[s];
    }
    A.qflag_out[qid] = 0;
    emit_result<MODE>(A, A.out_slot ? (int64_t)A.out_slot[qid] : (int64_t)qid, cnt, [&](int s, double &d, double &x, double &y, double &z, int32_t &id) {
        d = bd[s * PC_WAVE + lane];
        RecTraits<REC>::load(rec + bp[s * PC_WAVE + lane], x, y, z, id);
    });
}

// out[0] += listed queries whose bound fits the rings (the ball of the bound inside RING_MAX cells), out[1] += those that
// come without a bound
template <typename REC>
__global__ __launch_bounds__(256) void k_ring_eligible(ScanArgs A, const int32_t *__restrict__ list, uint32_t count, uint32_t *__restrict__ out) {
    const uint32_t qi = blockIdx.x * 256 + threadIdx.x;
    bool fits = false, boundless = false;
    if (qi < count) {
        double x, y, z;
        int32_t qid;
        RecTraits<REC>::load(reinterpret_cast<const REC *>(A.rec) + list[qi], x, y, z, qid);
        const float ub = A.qub_arr[qid];
        boundless = !(ub < INFINITY);
        fits = !boundless && sqrt((double)ub) * A.gp.s <= (double)RING_MAX;
    }
    const unsigned long long f = __ballot(fits), b = __ballot(boundless);
    if ((threadIdx.x & 63) == 0) {
        if (f) atomicAdd(out, (uint32_t)__popcll(f));
        if (b) atomicAdd(out + 1, (uint32_t)__popcll(b));
    }
}

template <typename REC, int KS>
void launch_mode(int mode, unsigned nb, size_t lds, hipStream_t st, const ScanArgs &A, const int32_t *list, uint32_t count, uint32_t stride, uint32_t *answered) {
    switch (mode) {
        case M_KNN: hipLaunchKernelGGL((k_ring_scan<REC, KS, M_KNN>), dim3(nb), dim3(PC_WAVE), lds, st, A, list, count, stride, answered); break;
        case M_SOR: hipLaunchKernelGGL((k_ring_scan<REC, KS, M_SOR>), dim3(nb), dim3(PC_WAVE), lds, st, A, list, count, stride, answered); break;
        case M_NORMALS: hipLaunchKernelGGL((k_ring_scan<REC, KS, M_NORMALS>), dim3(nb), dim3(PC_WAVE), lds, st, A, list, count, stride, answered); break;
        case M_NNDIST: hipLaunchKernelGGL((k_ring_scan<REC, KS, M_NNDIST>), dim3(nb), dim3(PC_WAVE), lds, st, A, list, count, stride, answered); break;
        case M_ALL: hipLaunchKernelGGL((k_ring_scan<REC, KS, M_ALL>), dim3(nb), dim3(PC_WAVE), lds, st, A, list, count, stride, answered); break;
        default: break;
    }
}

template <typename REC>
void launch_ks(int ks, int mode, unsigned nb, hipStream_t st, const ScanArgs &A, const int32_t *list, uint32_t count, uint32_t stride, uint32_t *answered) {
    const size_t lds = std::max((sizeof(double) + sizeof(uint32_t)) * PC_WAVE * (size_t)ks, 2 * sizeof(uint32_t) * PC_WAVE * (size_t)PARK_SLOTS);
    switch (ks) {
        case 4: launch_mode<REC, 4>(mode, nb, lds, st, A, list, count, stride, answered); break;
        case 10: launch_mode<REC, 10>(mode, nb, lds, st, A, list, count, stride, answered); break;
        case 18: launch_mode<REC, 18>(mode, nb, lds, st, A, list, count, stride, answered); break;
        case 22: launch_mode<REC, 22>(mode, nb, lds, st, A, list, count, stride, answered); break;
        default: launch_mode<REC, 34>(mode, nb, lds, st, A, list, count, stride, answered); break;
    }
}

}  // namespace

// One pass over `count` of the listed open queries, every stride-th entry of the list (ascending sorted positions of
// the level A describes -- the finest; A.qflag_out / A.qub_arr are the per-point flags and bounds; entries whose flag
// is clear already are left alone).  answered: a device word that counts the queries closed, or NULL.
int pc_launch_ring_scan(pc_ctx *ctx, int dtype, int ks, int mode, const ScanArgs &A, const int32_t *list, int64_t count,
                        int64_t stride, uint32_t *answered, hipStream_t st) {
    if (count <= 0) return PC_OK;
    if (ks > 34 || !A.qflag_out || !A.qub_arr) return pc_fail(ctx, PC_BAD_ARG, "ring scan");
    const unsigned nb = (unsigned)((count + PC_WAVE - 1) / PC_WAVE);
    if (dtype == PC_F32)
        launch_ks<Rec32>(ks, mode, nb, st, A, list, (uint32_t)count, (uint32_t)stride, answered);
    else
        launch_ks<Rec64>(ks, mode, nb, st, A, list, (uint32_t)count, (uint32_t)stride, answered);
    PC_HIP(hipGetLastError());
    return PC_OK;
}

// How many of the listed queries come with a bound that the rings hold (out_dev[0]) and how many without one (out_dev[1]).
int pc_ring_eligible(pc_ctx *ctx, int dtype, const ScanArgs &A, const int32_t *list, int64_t count, uint32_t *out_dev, hipStream_t st) {
    if (count <= 0) return PC_OK;
    const unsigned nb = (unsigned)((count + 255) / 256);
    if (dtype == PC_F32)
        hipLaunchKernelGGL(k_ring_eligible<Rec32>, dim3(nb), dim3(256), 0, st, A, list, (uint32_t)count, out_dev);
    else
        hipLaunchKernelGGL(k_ring_eligible<Rec64>, dim3(nb), dim3(256), 0, st, A, list, (uint32_t)count, out_dev);
    PC_HIP(hipGetLastError());
    return PC_OK;
}
