// One wave per query: the exact fp64 re-run of the queries the tile kernels could not certify.
// The wave looks at a block of cells around the query -- a ring of R cells at one level of the index --
// cuts away the rows (and row ends) that lie farther than a known upper bound of the k-th distance,
// strides its 64 lanes over the candidates that are left, keeps those within the bound, compacts them
// into LDS with ballot prefix sums, ranks them by (d2, original index) and certifies the result against
// the block faces.  As soon as a bound is known the next block is the smallest one, on the finest level,
// that certainly holds everything within the bound: that block certifies.  Until then the blocks grow
// level by level.  Queries whose survivor set does not fit LDS go to a second list that the
// one-query-per-lane scan answers.
#include "pc_knn_dev.h"

namespace {

constexpr int SCAP = 512;  // survivors one query may hold in LDS

constexpr int ROW_CACHE = 256;  // rows of a block kept in LDS (begin and running count): rings up to 7 cells

constexpr int JUMP_RING = 4;  // a known bound picks the finest level on which a ring of at most this many cells reaches it

constexpr int AHEAD = 4;  // candidate records in flight per lane

constexpr uint32_t PROBE_MIN = 2048, PROBE_LEN = 512;  // blocks with more candidates are probed first, with so many

template <typename REC, int MODE>
__global__ __launch_bounds__(PC_WAVE) __attribute__((amdgpu_waves_per_eu(MODE == M_ALL ? 3 : 4))) void k_wave_query(ScanArgs A) {
    __shared__ double sv_d[SCAP];
    __shared__ uint32_t sv_pos[SCAP];
    __shared__ int32_t sv_id[SCAP];
    __shared__ double res_d[PC_MAX_K];
    __shared__ uint32_t res_pos[PC_MAX_K];
    __shared__ double res_x[PC_MAX_K], res_y[PC_MAX_K], res_z[PC_MAX_K];
    __shared__ int32_t res_id[PC_MAX_K];
    __shared__ uint32_t row_b[ROW_CACHE], row_off[ROW_CACHE + 1];  // first record of each block row, candidates before it
    const int lane = threadIdx.x;
    const REC *rec0 = reinterpret_cast<const REC *>(A.rec);
    const int k = A.k;
    const uint32_t nq = min(*A.qcount, A.flag_cap);
    const double r2lim = A.radius2;

    for (uint32_t qi = blockIdx.x; qi < nq; qi += gridDim.x) {  // wave-uniform
        const int64_t spos = A.qlist[qi];
        double ub = (double)A.qub[qi];
        double qx, qy, qz;
        int32_t qid;
        RecTraits<REC>::load((A.qrec ? reinterpret_cast<const REC *>(A.qrec) : rec0) + spos, qx, qy, qz, qid);
        const double qv[3] = {qx, qy, qz};
        int cnt = 0, lvl = 0, R = ub < INFINITY || r2lim >= 0.0 ? 1 : max(A.first_ring, 1);
        bool spilled = false, jumped = false;
        // The finest level and ring whose block holds everything within sqrt(bound2) of the query: the ring
        // reaches R cells beyond the query's own cell on every side, i.e. at least R * cell.
        auto jump = [&](double bound2) -> bool {
            const double need = sqrt(bound2) * (1.0 + 1e-6);
            for (int l = 0; l < A.nlv; l++) {
                const double r = ceil(need / A.lv[l].gp.cell);
                if (r <= (double)JUMP_RING) {
                    lvl = l;
                    R = max(1, (int)r);
                    return true;
                }
            }
            return false;
        };
        {
            const double pb0 = r2lim >= 0.0 ? fmin(ub, r2lim) : ub;
            if (pb0 < INFINITY) jumped = jump(pb0);
        }
        for (;;) {
            lvl = __builtin_amdgcn_readfirstlane(lvl);
            R = __builtin_amdgcn_readfirstlane(R);
            // ---- the block: ring R around the query's cell at level lvl -----------------------------
            const GridParams gp = A.lv[lvl].gp;
            const REC *rec = reinterpret_cast<const REC *>(A.lv[lvl].rec);
            const CellIndex ci = A.lv[lvl].ci;
            const bool last_level = lvl == A.nlv - 1;
            int lo[3], hi[3], cc[3];
            bool whole = true;
            double dmin = INFINITY;
#pragma unroll
            for (int a = 0; a < 3; a++) {
                const double t = floor((qv[a] - gp.o[a]) * gp.s);
                const int c = __builtin_amdgcn_readfirstlane(
                    !(t > 0.0) ? 0 : (t >= (double)gp.dim[a] ? gp.dim[a] - 1 : (int)t));
                cc[a] = c;
                lo[a] = max(c - R, 0);
                hi[a] = min(c + R, gp.dim[a] - 1);
                if (lo[a] > 0) {
                    whole = false;
                    dmin = fmin(dmin, (qv[a] - gp.o[a]) - (double)lo[a] * gp.cell);  // (in offsets from the origin, like the cell keys)
                }
                if (hi[a] < gp.dim[a] - 1) {
                    whole = false;
                    dmin = fmin(dmin, (double)(hi[a] + 1) * gp.cell - (qv[a] - gp.o[a]));
                }
            }
            double b2 = INFINITY;  // no point outside the block is closer than sqrt(b2)
            if (!whole) {
                const double b = dmin - 1e-7 * gp.cell;
                b2 = b > 0 ? b * b : 0.0;
            }
            const bool can_end_short = whole || (r2lim >= 0.0 && b2 >= r2lim);
            const int nry = hi[1] - lo[1] + 1, nrows = nry * (hi[2] - lo[2] + 1);
            const bool flat = nrows <= ROW_CACHE;
            const int rc = (cc[1] - lo[1]) + nry * (cc[2] - lo[2]);  // the query's own row goes first: its points bound the rest
            double pb = r2lim >= 0.0 ? fmin(ub, r2lim) : ub;   // nothing farther than this (squared) is ever kept
            const double slack = 1e-6 * gp.cell;
            // A row of the block in scan order: its first record and its length, after cutting away what the
            // bound excludes -- the whole row when its (y, z) slab is farther than the bound, else the cells
            // on either end that are.
            auto row_range = [&](int ri, uint32_t &b, uint32_t &len) {
                const int r = ri + rc < nrows ? ri + rc : ri + rc - nrows;
                const int ry = lo[1] + r % nry, rz = lo[2] + r / nry;
                int x0 = lo[0], x1 = hi[0];
                b = 0;
                len = 0;
                if (pb < INFINITY) {
                    double rem = pb;
                    const int rcell[3] = {0, ry, rz};
#pragma unroll
                    for (int a = 1; a < 3; a++) {
                        const double q = qv[a] - gp.o[a];
                        const int c = rcell[a];
                        const double g = c > cc[a] ? (double)c * gp.cell - q : (c < cc[a] ? q - (double)(c + 1) * gp.cell : 0.0);
                        const double gs = fmax(g - slack, 0.0);
                        rem -= gs * gs;
                    }
                    if (rem < 0.0) return;
                    const double xr = sqrt(rem) + slack, q = qv[0] - gp.o[0];
                    const double f0 = floor((q - xr) * gp.s), f1 = floor((q + xr) * gp.s);
                    if (f0 > (double)x1 || f1 < (double)x0) return;
                    if (f0 > (double)x0) x0 = (int)f0;
                    if (f1 < (double)x1) x1 = (int)f1;
                }
                const int64_t row = ((int64_t)rz * gp.dim[1] + ry) * gp.dim[0];
                uint32_t e;
                cell_range(ci, row + x0, row + x1 + 1, b, e);
                len = e - b;
            };
            // Rows [first, first + ROW_CACHE) of the block under the bound pb, lanes over rows (the index searches of
            // all rows overlap here): their begins and running counts go to LDS when `keep`; returns their
            // population.  A block of at most ROW_CACHE rows is one such chunk.
            int crows = 0;  // rows of the chunk in LDS
            auto build_rows = [&](int first, bool keep) -> uint32_t {
                const int last = min(first + ROW_CACHE, nrows);
                uint32_t total = 0;
                for (int base = first; base < last; base += PC_WAVE) {
                    const int ri = base + lane;
                    uint32_t b = 0, len = 0;
                    if (ri < last) row_range(ri, b, len);
                    uint32_t inc = len;
#pragma unroll
                    for (int o = 1; o < PC_WAVE; o <<= 1) {
                        const uint32_t t = __shfl_up(inc, o, 64);
                        if (lane >= o) inc += t;
                    }
                    if (keep && ri < last) {
                        row_b[ri - first] = b;
                        row_off[ri - first] = total + inc - len;
                    }
                    total += __shfl(inc, PC_WAVE - 1, 64);
                }
                total = __builtin_amdgcn_readfirstlane(total);
                if (keep) {
                    crows = last - first;
                    if (lane == 0) row_off[crows] = total;
                    __syncthreads();
                }
                return total;
            };
            uint32_t total = 0;
            if (flat) {
                total = build_rows(0, true);
            } else {
                for (int first = 0; first < nrows; first += ROW_CACHE) total += build_rows(first, false);
            }

            bool certified = false;
            if (total >= (uint32_t)k || can_end_short) {
                // ---- survivors: candidates not farther than the bound, compacted into LDS ---------------
                // Thinning inside LDS: the k-th smallest of ANY 64 held survivors bounds the true k-th
                // distance, so everything farther than it can go (and later candidates are tested
                // against that bound).  Repeated while it still removes something.
                auto thin = [&](uint32_t mm) -> uint32_t {
                    while (mm > 2 * PC_WAVE) {
                        const double di = sv_d[lane];
                        const int32_t ii = sv_id[lane];
                        int rank = 0;
                        for (int j = 0; j < PC_WAVE; j++) {
                            const double dj = __shfl(di, j, 64);
                            const int32_t ij = __shfl(ii, j, 64);
                            rank += (dj < di || (dj == di && ij < ii)) ? 1 : 0;
                        }
                        const unsigned long long sel = __ballot(rank == k - 1);
                        const double cut = __shfl(di, __ffsll((long long)sel) - 1, 64);  // k <= 64: one lane has it
                        uint32_t kept = 0;
                        for (uint32_t base = 0; base < mm; base += PC_WAVE) {
                            const uint32_t i = base + lane;
                            const bool have = i < mm;
                            const double d = have ? sv_d[i] : 0.0;
                            const uint32_t p = have ? sv_pos[i] : 0u;
                            const int32_t id = have ? sv_id[i] : 0;
                            const bool keep = have && d <= cut;
                            const unsigned long long bal = __ballot(keep);
                            const uint32_t at = kept + __popcll(bal & ((1ull << lane) - 1ull));
                            __syncthreads();  // every lane has read its entry before any slot is rewritten
                            if (keep) {
                                sv_d[at] = d;
                                sv_pos[at] = p;
                                sv_id[at] = id;
                            }
                            kept += __popcll(bal);
                        }
                        __syncthreads();
                        ub = fmin(ub, cut);  // a bound of the true k-th distance, for later blocks too
                        if (kept == mm) break;  // ties only: nothing left to remove
                        mm = kept;
                    }
                    return mm;
                };
                uint32_t m = 0;
                bool overflow = false;  // more equal-distance survivors than LDS holds
                // one batch of up to 64 loaded candidates: test against the bound, append the survivors
                auto take = [&](bool have, uint32_t j, double px, double py, double pz, int32_t pid) {
                    if (m > SCAP - PC_WAVE) {  // wave-uniform: make room before the next 64 candidates
                        __syncthreads();
                        m = thin(m);
                        if (m > SCAP - PC_WAVE) {
                            overflow = true;
                            return;
                        }
                    }
                    const double dx = qx - px, dy = qy - py, dz = qz - pz;
                    double d = dx * dx;
                    d += dy * dy;
                    d += dz * dz;
                    const bool keep = have && d <= ub && !(r2lim >= 0.0 && !(d < r2lim));
                    const unsigned long long bal = __ballot(keep);
                    const uint32_t at = m + __popcll(bal & ((1ull << lane) - 1ull));
                    if (keep) {
                        sv_d[at] = d;
                        sv_pos[at] = j;
                        sv_id[at] = pid;
                    }
                    m += __popcll(bal);
                };
                // all rows as one run of `total` candidates, the first `limit` of them: every lane finds the row of
                // its candidate by the running counts (one row for the whole batch where rows are long), AHEAD
                // records are in flight per lane
                auto scan_flat = [&](uint32_t limit) {
                    int cur = 0;  // the row of the batch's first candidate (uniform)
                    for (uint32_t t0 = 0; t0 < limit && !overflow; t0 += AHEAD * PC_WAVE) {
                        uint32_t jj[AHEAD];
                        double px[AHEAD], py[AHEAD], pz[AHEAD];
                        int32_t pid[AHEAD];
                        // where the AHEAD batches' records are ...
#pragma unroll
                        for (int u = 0; u < AHEAD; u++) {
                            const uint32_t tf = t0 + u * PC_WAVE, t = tf + lane;
                            if (tf < limit)  // uniform
                                while (cur + 1 < crows && row_off[cur + 1] <= tf) cur++;
                            const bool one_row = cur + 1 >= crows || row_off[cur + 1] > tf + (PC_WAVE - 1);
                            int at = cur;
                            if (!one_row) {  // uniform
                                at = 0;  // the last row that starts at or before candidate t (rows of no length start where the next does)
#pragma unroll
                                for (int step = ROW_CACHE / 2; step > 0; step >>= 1)
                                    at += (at + step < crows && row_off[min(at + step, ROW_CACHE)] <= t) ? step : 0;
                            }
                            jj[u] = t < limit ? row_b[at] + (t - row_off[at]) : 0u;
                        }
                        // ... then the loads, back to back with no branch between or around them (lanes past the
                        // end read record 0 and drop it): they are in flight together and waited for once
#pragma unroll
                        for (int u = 0; u < AHEAD; u++) RecTraits<REC>::load(rec + jj[u], px[u], py[u], pz[u], pid[u]);
#pragma unroll
                        for (int u = 0; u < AHEAD; u++) {
                            if (t0 + u * PC_WAVE >= limit || overflow) break;
                            take(t0 + u * PC_WAVE + lane < limit, jj[u], px[u], py[u], pz[u], pid[u]);
                        }
                    }
                };
                // A big block under a loose (or no) bound: the first few hundred candidates -- the query's own
                // row first -- give a bound, the rows are cut again under it, and so on while that still removes
                // a quarter of the block.  (A stray point next to a cluster: 185 000 candidates in its first
                // block with k points, 2.7 ms for one wave to look at them all.)
                bool moved = false;  // the probe's bound asked for a block on a finer level
                for (int probe = 0; probe < 6 && flat && total > PROBE_MIN && !can_end_short; probe++) {
                    const double ub0 = ub;
                    scan_flat(PROBE_LEN);
                    __syncthreads();
                    if (!overflow) thin(m);
                    m = 0;
                    overflow = false;
                    if (!(ub < ub0)) break;  // nothing learned
                    pb = r2lim >= 0.0 ? fmin(ub, r2lim) : ub;
                    if (!jumped) {
                        // first bound of this query: cells of this level may be far wider than the bound (a stray
                        // point whose first block with k points spans a whole cluster) -- then the block that
                        // certifies lies on a finer level, where cutting rows by the bound means something
                        const int l0 = lvl, r0 = R;
                        if (jump(pb) && lvl < l0) {
                            jumped = true;
                            moved = true;
                            break;
                        }
                        lvl = l0;
                        R = r0;
                    }
                    const uint32_t before = total;
                    total = build_rows(0, true);
                    if (4ull * total > 3ull * before) break;
                }
                if (moved) continue;
                if (flat) {
                    scan_flat(total);
                } else {
                    // more rows than LDS keeps (a large ring on the coarsest level): a chunk of rows at a time
                    for (int first = 0; first < nrows && !overflow; first += ROW_CACHE) {
                        __syncthreads();  // the previous chunk's rows are done with
                        const uint32_t ct = build_rows(first, true);
                        scan_flat(ct);
                    }
                }
                __syncthreads();
                if (overflow) {
                    spilled = true;
                    break;
                }
                const uint32_t mm = thin(m);
                // rank the held survivors by (d2, original index)
                for (uint32_t base = 0; base < mm; base += PC_WAVE) {
                    const uint32_t i = base + lane;
                    const bool have = i < mm;
                    const double di = have ? sv_d[i] : INFINITY;
                    const int32_t ii = have ? sv_id[i] : 0x7fffffff;
                    uint32_t rank = 0;
                    for (uint32_t j = 0; j < mm; j++) {
                        const double dj = sv_d[j];
                        const int32_t ij = sv_id[j];
                        rank += (dj < di || (dj == di && ij < ii)) ? 1u : 0u;
                    }
                    if (have && rank < (uint32_t)k) {
                        res_d[rank] = di;
                        res_pos[rank] = sv_pos[i];
                    }
                }
                __syncthreads();
                cnt = (int)min(mm, (uint32_t)k);
                const double kth = cnt == k ? res_d[k - 1] : INFINITY;
                __syncthreads();
                certified = can_end_short || (cnt == k && kth <= b2);
                if (certified) {
                    // results refer to this level's records: fetched by the lanes side by side, consumed by one
                    if (lane < cnt) {
                        double x, y, z;
                        int32_t id;
                        RecTraits<REC>::load(rec + res_pos[lane], x, y, z, id);
                        res_x[lane] = x;
                        res_y[lane] = y;
                        res_z[lane] = z;
                        res_id[lane] = id;
                    }
                    __syncthreads();
                    if (lane == 0) {
                        emit_result<MODE>(A, out_row_of(A, qid, spos), cnt,
                                          [&](int s, double &d, double &x, double &y, double &z, int32_t &id) {
                                              d = res_d[s];
                                              x = res_x[s];
                                              y = res_y[s];
                                              z = res_z[s];
                                              id = res_id[s];
                                          });
                    }
                    break;
                }
                if (cnt == k) ub = fmin(ub, kth);
            }
            // The next block.  With a bound: the block that certainly holds everything within it (it certifies;
            // should rounding at a cell face ever deny that, the ring grows from there).  Without one: ring 1
            // at each level in turn (block edges 3, 6, 12, ... fine cells), then rings half as wide again each
            // time at the coarsest level.
            const double pbn = r2lim >= 0.0 ? fmin(ub, r2lim) : ub;
            if (jumped) {
                R++;
            } else if (pbn < INFINITY && jump(pbn)) {
                jumped = true;
            } else if (!last_level) {
                lvl++;
                R = 1;
            } else {
                R += max(1, R / 2);  // (no bound yet: half as far again; the first bound cuts the block back)
            }
        }
        if (spilled && lane == 0) {
            const uint32_t p = atomicAdd(A.flag2_count, 1u);
            if (p < A.flag_cap) A.flag2_list[p] = (int32_t)spos;
        }
        __syncthreads();
    }
}

template <typename REC>
void launch_wave(int mode, unsigned nb, hipStream_t st, const ScanArgs &A) {
    switch (mode) {
        case M_KNN: hipLaunchKernelGGL((k_wave_query<REC, M_KNN>), dim3(nb), dim3(PC_WAVE), 0, st, A); break;
        case M_SOR: hipLaunchKernelGGL((k_wave_query<REC, M_SOR>), dim3(nb), dim3(PC_WAVE), 0, st, A); break;
        case M_NORMALS: hipLaunchKernelGGL((k_wave_query<REC, M_NORMALS>), dim3(nb), dim3(PC_WAVE), 0, st, A); break;
        case M_NNDIST: hipLaunchKernelGGL((k_wave_query<REC, M_NNDIST>), dim3(nb), dim3(PC_WAVE), 0, st, A); break;
        case M_ALL: hipLaunchKernelGGL((k_wave_query<REC, M_ALL>), dim3(nb), dim3(PC_WAVE), 0, st, A); break;
        default: break;
    }
}

}  // namespace

// Answers A.qlist / A.qub / A.qcount; queries it cannot hold go to A.flag2_list / A.flag2_count.
int pc_launch_wave_query(pc_ctx *ctx, const pc_grid *g, int mode, const ScanArgs &A0, hipStream_t st) {
    ScanArgs A = A0;
    A.nlv = 0;
    // the finer grids that exist already, finest first: a query in a crowded corner of the cloud finds k
    // neighbours within a few of their cells, where level 0 would have it scan tens of thousands of points
    {
        const pc_grid *fine[MAX_FINER];
        int nf = 0;
        for (const pc_grid *f = g->finer; f && nf < MAX_FINER; f = f->finer) fine[nf++] = f;
        while (nf > 0) {
            const pc_grid *l = fine[--nf];
            A.lv[A.nlv].gp = l->gp;
            A.lv[A.nlv].rec = l->rec;
            A.lv[A.nlv].ci = l->index();
            A.nlv++;
        }
    }
    for (const pc_grid *l = g; l && A.nlv < MAX_LEVELS + MAX_FINER; l = l->coarse) {
        A.lv[A.nlv].gp = l->gp;
        A.lv[A.nlv].rec = l->rec;
        A.lv[A.nlv].ci = l->index();
        A.nlv++;
    }
    const unsigned nb = 8192;  // persistent waves; each strides over the list
    if (g->dtype == PC_F32)
        launch_wave<Rec32>(mode, nb, st, A);
    else
        launch_wave<Rec64>(mode, nb, st, A);
    PC_HIP(hipGetLastError());
    return PC_OK;
}
