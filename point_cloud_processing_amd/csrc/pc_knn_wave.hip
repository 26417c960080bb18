// One wave per query: the exact fp64 re-run of the few queries the tile kernel could not
// certify.  The 64 lanes stride over the candidates of the ring block around the query's cell,
// keep those not farther than a known upper bound of the k-th distance, compact them into LDS
// with ballot prefix sums, rank them by (d2, original index) and certify the result against
// the block faces; an uncertified query grows its ring.  Queries whose survivor set does not
// fit LDS go to a second list that the one-query-per-lane scan answers.
#include "pc_knn_dev.h"

namespace {

constexpr int SCAP = 512;  // survivors one query may hold in LDS

constexpr int ROW_CACHE = 256;

template <typename REC, int MODE>
__global__ __launch_bounds__(PC_WAVE) void k_wave_query(ScanArgs A) {
    __shared__ double sv_d[SCAP];
    __shared__ uint32_t sv_pos[SCAP];
    __shared__ int32_t sv_id[SCAP];
    __shared__ double res_d[PC_MAX_K];
    __shared__ uint32_t res_pos[PC_MAX_K];
    __shared__ uint32_t row_b[ROW_CACHE], row_e[ROW_CACHE];  // record range of each block row
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
        int cnt = 0, lvl = 0, R = 1;
        bool spilled = false;
        for (;;) {
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
            // population of the block: lanes over rows
            uint32_t total = 0;
            for (int r = lane; r < nrows; r += PC_WAVE) {
                const int64_t row = ((int64_t)(lo[2] + r / nry) * gp.dim[1] + (lo[1] + r % nry)) * gp.dim[0];
                const uint32_t b = cell_begin(ci, row + lo[0]), e = cell_begin(ci, row + hi[0] + 1);
                total += e - b;
                if (r < ROW_CACHE) {  // kept for the scan below: the searches of all rows overlap here
                    row_b[r] = b;
                    row_e[r] = e;
                }
            }
            __syncthreads();
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o, 64);
            total = __builtin_amdgcn_readfirstlane(total);

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
                        ub = fmin(ub, cut);  // a bound of the true k-th distance, for later rings too
                        if (kept == mm) break;  // ties only: nothing left to remove
                        mm = kept;
                    }
                    return mm;
                };
                uint32_t m = 0;
                bool overflow = false;  // more equal-distance survivors than LDS holds
                // the query's own row first: its nearest points give a tight bound at the first thinning
                const int rc = (cc[1] - lo[1]) + nry * (cc[2] - lo[2]);
                for (int ri = 0; ri < nrows && !overflow; ri++) {
                    const int r = ri + rc < nrows ? ri + rc : ri + rc - nrows;
                    const int64_t row = ((int64_t)(lo[2] + r / nry) * gp.dim[1] + (lo[1] + r % nry)) * gp.dim[0];
                    uint32_t jb, je;
                    if (r < ROW_CACHE) {
                        jb = row_b[r];
                        je = row_e[r];
                    } else {
                        // sparse index: lanes 0 and 1 search, the wave shares the two results
                        uint32_t jv = 0;
                        if (lane < 2) jv = cell_begin(ci, lane == 0 ? row + lo[0] : row + hi[0] + 1);
                        jb = __builtin_amdgcn_readfirstlane(__shfl(jv, 0, 64));
                        je = __builtin_amdgcn_readfirstlane(__shfl(jv, 1, 64));
                    }
                    for (uint32_t j0 = jb; j0 < je; j0 += PC_WAVE) {
                        if (m > SCAP - PC_WAVE) {  // wave-uniform: make room before the next 64 candidates
                            __syncthreads();
                            m = thin(m);
                            if (m > SCAP - PC_WAVE) {
                                overflow = true;
                                break;
                            }
                        }
                        const uint32_t j = j0 + lane;
                        bool keep = false;
                        double d = 0.0;
                        int32_t pid = 0;
                        if (j < je) {
                            double px, py, pz;
                            RecTraits<REC>::load(rec + j, px, py, pz, pid);
                            const double dx = qx - px, dy = qy - py, dz = qz - pz;
                            d = dx * dx;
                            d += dy * dy;
                            d += dz * dz;
                            keep = d <= ub && !(r2lim >= 0.0 && !(d < r2lim));
                        }
                        const unsigned long long bal = __ballot(keep);
                        const uint32_t at = m + __popcll(bal & ((1ull << lane) - 1ull));
                        if (keep) {
                            sv_d[at] = d;
                            sv_pos[at] = j;
                            sv_id[at] = pid;
                        }
                        m += __popcll(bal);
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
                    // results refer to this level's records
                    if (lane == 0) {
                        emit_result<MODE>(A, out_row_of(A, qid, spos), cnt,
                                          [&](int s, double &d, double &x, double &y, double &z, int32_t &id) {
                                              d = res_d[s];
                                              RecTraits<REC>::load(rec + res_pos[s], x, y, z, id);
                                          });
                    }
                    break;
                }
                if (cnt == k) ub = fmin(ub, kth);
            }
            // widen: ring 1 at each level in turn (block edges 3, 6, 12, ... fine cells), then ever
            // larger rings at the coarsest level
            if (!last_level) {
                lvl++;
                R = 1;
            } else {
                R++;
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
    for (const pc_grid *l = g; l && A.nlv < MAX_LEVELS; l = l->coarse) {
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
