// One wave per query: the exact fp64 re-run of the few queries the tile kernel could not
// certify.  The 64 lanes stride over the candidates of the ring block around the query's cell,
// keep those not farther than a known upper bound of the k-th distance, compact them into LDS
// with ballot prefix sums, rank them by (d2, original index) and certify the result against
// the block faces; an uncertified query grows its ring.  Queries whose survivor set does not
// fit LDS go to a second list that the one-query-per-lane scan answers.
#include "pc_knn_dev.h"

namespace {

constexpr int SCAP = 512;  // survivors one query may hold in LDS

template <typename REC, int MODE>
__global__ __launch_bounds__(PC_WAVE) void k_wave_query(ScanArgs A) {
    __shared__ double sv_d[SCAP];
    __shared__ uint32_t sv_pos[SCAP];
    __shared__ int32_t sv_id[SCAP];
    __shared__ double res_d[PC_MAX_K];
    __shared__ uint32_t res_pos[PC_MAX_K];
    const int lane = threadIdx.x;
    const REC *rec = reinterpret_cast<const REC *>(A.rec);
    const GridParams &gp = A.gp;
    const int k = A.k;
    const uint32_t nq = min(*A.qcount, A.flag_cap);
    const double eps = 1e-7 * gp.cell;
    const double r2lim = A.radius2;

    for (uint32_t qi = blockIdx.x; qi < nq; qi += gridDim.x) {  // wave-uniform
        const int64_t spos = A.qlist[qi];
        double ub = (double)A.qub[qi];
        double qx, qy, qz;
        int32_t qid;
        RecTraits<REC>::load(rec + spos, qx, qy, qz, qid);
        const double qv[3] = {qx, qy, qz};
        int c[3];
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const double t = floor((qv[a] - gp.o[a]) * gp.s);
            c[a] = __builtin_amdgcn_readfirstlane(!(t > 0.0) ? 0 : (t >= (double)gp.dim[a] ? gp.dim[a] - 1 : (int)t));
        }
        int cnt = 0;
        bool spilled = false;
        // ring 0 here means "the query's own row of three cells" and only serves to find an
        // upper bound when none came with the query
        for (int R = (ub < INFINITY) ? 1 : 0;; R++) {
            int lo[3], hi[3];
#pragma unroll
            for (int a = 0; a < 3; a++) {
                const int Ra = (R == 0 && a > 0) ? 0 : (R == 0 ? 1 : R);
                lo[a] = max(c[a] - Ra, 0);
                hi[a] = min(c[a] + Ra, gp.dim[a] - 1);
            }
            uint32_t m = 0;  // survivors so far (wave-uniform)
            for (int z = lo[2]; z <= hi[2]; z++)
                for (int y = lo[1]; y <= hi[1]; y++) {
                    const int64_t row = ((int64_t)z * gp.dim[1] + y) * gp.dim[0];
                    const uint32_t jb = A.cell_start[row + lo[0]], je = A.cell_start[row + hi[0] + 1];
                    for (uint32_t j0 = jb; j0 < je; j0 += PC_WAVE) {
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
                        if (keep && at < SCAP) {
                            sv_d[at] = d;
                            sv_pos[at] = j;
                            sv_id[at] = pid;
                        }
                        m += __popcll(bal);
                    }
                }
            __syncthreads();
            if (m > SCAP) {
                // too many survivors for LDS: tighten the bound from the ones held, or give up
                // after one attempt (heavy ties: every survivor at the same distance)
                spilled = true;
            }
            const uint32_t mm = min(m, (uint32_t)SCAP);
            // rank the held survivors by (d2, original index)
            double kth = INFINITY;
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
            if (cnt == k) kth = res_d[k - 1];
            if (spilled) {
                if (cnt == k && kth < ub) {
                    // the k-th of the held survivors bounds the true k-th: retry this ring with it
                    ub = kth;
                    spilled = false;
                    R--;
                    __syncthreads();
                    continue;
                }
                break;  // cannot shrink: hand over to the per-lane scan
            }
            if (R == 0) {
                if (cnt == k) ub = kth;
                __syncthreads();
                continue;  // now scan ring 1 with (or still without) a bound
            }
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
            if (whole) break;
            const double b = dmin - eps;
            if (b > 0) {
                const double b2 = b * b;
                if (r2lim >= 0.0 && b2 >= r2lim) break;
                if (cnt == k && kth <= b2) break;
            }
            if (cnt == k) ub = fmin(ub, kth);
            __syncthreads();
        }
        if (spilled) {
            if (lane == 0) {
                const uint32_t p = atomicAdd(A.flag2_count, 1u);
                if (p < A.flag_cap) A.flag2_list[p] = (int32_t)spos;
            }
        } else if (lane == 0) {
            emit_result<MODE>(A, (int64_t)qid, cnt, [&](int s, double &d, double &x, double &y, double &z, int32_t &id) {
                d = res_d[s];
                RecTraits<REC>::load(rec + res_pos[s], x, y, z, id);
            });
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
int pc_launch_wave_query(pc_ctx *ctx, const pc_grid *g, int mode, const ScanArgs &A, hipStream_t st) {
    const unsigned nb = 8192;  // persistent waves; each strides over the list
    if (g->dtype == PC_F32)
        launch_wave<Rec32>(mode, nb, st, A);
    else
        launch_wave<Rec64>(mode, nb, st, A);
    PC_HIP(hipGetLastError());
    return PC_OK;
}
