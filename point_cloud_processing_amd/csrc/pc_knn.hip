// Exact k-nearest-neighbour scan over the uniform grid and the consumers fused onto it:
// statistical-outlier mean distance, PCA normals, nearest-neighbour distance, radius count.
//
// One query per lane.  A query walks the shells of cells around its own cell (rows of cells
// are contiguous point ranges because x is the fastest key digit), measures every candidate
// in fp64 in the reference's operation order, and stops once the k-th best squared distance
// is certified by the distance to the nearest face of the scanned block.  The per-lane
// neighbour list lives in LDS as list[slot][lane] (conflict-free when lanes touch the same
// slot).  All kernels are HBM/LDS/VALU work; nothing here is a dense contraction.
#include <math.h>

#include <stdio.h>

#include <algorithm>

#include "pc_knn_dev.h"

namespace {

// ---- the scan -----------------------------------------------------------------------------------
template <typename REC, typename TQ, int MODE>
__global__ __launch_bounds__(PC_WAVE) void k_scan(ScanArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    const int k = A.k;
    double *Ld = reinterpret_cast<double *>(smem);                                  // [k][64]
    int32_t *Lp = reinterpret_cast<int32_t *>(smem + sizeof(double) * PC_WAVE * k);  // [k][64]
    const REC *rec = reinterpret_cast<const REC *>(A.rec);
    const GridParams &gp = A.gp;
    const int64_t nq = A.qlist ? (int64_t)min(*A.qcount, A.flag_cap) : A.nq;

    // lanes are independent (no barriers): a block walks its queries in a grid-stride loop
    for (int64_t qi = (int64_t)blockIdx.x * PC_WAVE + lane; qi < nq; qi += (int64_t)gridDim.x * PC_WAVE) {
        double qx, qy, qz;
        int64_t out_row;
        if (A.queries == nullptr) {
            int32_t id;
            const int64_t s = A.qlist ? (int64_t)A.qlist[qi] : A.s_begin + qi;
            RecTraits<REC>::load((A.qrec ? reinterpret_cast<const REC *>(A.qrec) : rec) + s, qx, qy, qz, id);
            out_row = out_row_of(A, id, s);
        } else {
            const TQ *q = reinterpret_cast<const TQ *>(A.queries) + 3 * (A.s_begin + qi);
            qx = (double)q[0];
            qy = (double)q[1];
            qz = (double)q[2];
            out_row = A.s_begin + qi;
        }
        const double qv[3] = {qx, qy, qz};
        int c[3];
#pragma unroll
        for (int a = 0; a < 3; a++) {
            double t = floor((qv[a] - gp.o[a]) * gp.s);
            c[a] = !(t > 0.0) ? 0 : (t >= (double)gp.dim[a] ? gp.dim[a] - 1 : (int)t);
        }
        const double eps = 1e-7 * gp.cell;
        const double r2lim = A.radius2;
        int cnt = 0;
        double worst = INFINITY;  // k-th best d2 once the list is full
        int32_t worst_idx = 0x7fffffff;

        for (int R = 0;; R++) {
            int lo[3], hi[3];
#pragma unroll
            for (int a = 0; a < 3; a++) {
                lo[a] = c[a] - R < 0 ? 0 : c[a] - R;
                hi[a] = c[a] + R >= gp.dim[a] ? gp.dim[a] - 1 : c[a] + R;
            }
            for (int z = lo[2]; z <= hi[2]; z++) {
                for (int y = lo[1]; y <= hi[1]; y++) {
                    const int64_t row = ((int64_t)z * gp.dim[1] + y) * gp.dim[0];
                    const bool full = (z - c[2] == R) || (c[2] - z == R) || (y - c[1] == R) || (c[1] - y == R);
                    const int nseg = full ? 1 : (R == 0 ? 1 : 2);
                    for (int s = 0; s < nseg; s++) {
                        int x0, x1;
                        if (full) {
                            x0 = lo[0];
                            x1 = hi[0];
                        } else {
                            x0 = x1 = (s == 0) ? c[0] - R : c[0] + R;
                            if (x0 < 0 || x0 >= gp.dim[0]) continue;
                        }
                        const uint32_t jb = cell_begin(A.ci, row + x0), je = cell_begin(A.ci, row + x1 + 1);
                        for (uint32_t j = jb; j < je; j++) {
                            double px, py, pz;
                            int32_t pid;
                            RecTraits<REC>::load(rec + j, px, py, pz, pid);
                            const double dx = qx - px, dy = qy - py, dz = qz - pz;
                            double d = dx * dx;
                            d += dy * dy;
                            d += dz * dz;
                            if (r2lim >= 0.0 && !(d < r2lim)) continue;
                            if (MODE == M_COUNT) {
                                cnt++;
                                continue;
                            }
                            if (cnt == k && !(d < worst || (d == worst && pid < worst_idx))) continue;
                            // sorted insertion, ties -> lower original index first
                            int p = cnt == k ? k - 1 : cnt;
                            while (p > 0) {
                                const double dp = Ld[(p - 1) * PC_WAVE + lane];
                                if (dp < d) break;
                                const int32_t pp = Lp[(p - 1) * PC_WAVE + lane];
                                if (dp == d) {
                                    double tx, ty, tz;
                                    int32_t tid;
                                    RecTraits<REC>::load(rec + pp, tx, ty, tz, tid);
                                    if (tid < pid) break;
                                }
                                Ld[p * PC_WAVE + lane] = dp;
                                Lp[p * PC_WAVE + lane] = pp;
                                p--;
                            }
                            Ld[p * PC_WAVE + lane] = d;
                            Lp[p * PC_WAVE + lane] = (int32_t)j;
                            if (cnt < k) cnt++;
                            if (cnt == k) {
                                worst = Ld[(k - 1) * PC_WAVE + lane];
                                double tx, ty, tz;
                                RecTraits<REC>::load(rec + Lp[(k - 1) * PC_WAVE + lane], tx, ty, tz, worst_idx);
                            }
                        }
                    }
                }
            }
            bool whole = true;
            double dmin = INFINITY;
#pragma unroll
            for (int a = 0; a < 3; a++) {
                if (lo[a] > 0) {
                    whole = false;
                    dmin = fmin(dmin, (qv[a] - gp.o[a]) - (double)lo[a] * gp.cell);  // (in offsets from the origin, like the cell keys)
                }
                if (hi[a] < gp.dim[a] - 1) {
                    whole = false;
                    dmin = fmin(dmin, (double)(hi[a] + 1) * gp.cell - (qv[a] - gp.o[a]));
                }
            }
            if (whole) break;
            const double b = dmin - eps;
            if (b > 0) {
                const double b2 = b * b;
                if (r2lim >= 0.0 && b2 >= r2lim) break;
                if (MODE != M_COUNT && cnt == k && worst <= b2) break;
            }
        }

        if (MODE == M_COUNT) {
            A.out_cnt[out_row] = cnt;
        } else {
            emit_result<MODE>(A, out_row, cnt,
                              [&](int s, double &d, double &x, double &y, double &z, int32_t &id) {
                                  d = Ld[s * PC_WAVE + lane];
                                  RecTraits<REC>::load(rec + Lp[s * PC_WAVE + lane], x, y, z, id);
                              });
        }
    }
}

template <int MODE>
int launch_scan(pc_ctx *ctx, const pc_grid *g, int qdtype, const ScanArgs &A0, hipStream_t st) {
    if (A0.nq <= 0) return PC_OK;
    ScanArgs A = A0;
    // fast path first; it leaves a device-side list of the queries it could not certify and the
    // exact scan below then runs over that list only (persistent blocks, no host round trip)
    bool tiled = false;
    PC_TRY(pc_launch_tile_scan(ctx, g, MODE, A, st, &tiled));
    const unsigned nb = tiled ? 512u : (unsigned)std::min<int64_t>((A.nq + PC_WAVE - 1) / PC_WAVE, 0x7fffffff);
    const size_t lds = MODE == M_COUNT ? 0 : (sizeof(double) + sizeof(int32_t)) * PC_WAVE * (size_t)A.k;
    const bool ext = A.queries != nullptr;
    if (g->dtype == PC_F32) {
        if (ext && qdtype == PC_F64)
            hipLaunchKernelGGL((k_scan<Rec32, double, MODE>), dim3(nb), dim3(PC_WAVE), lds, st, A);
        else
            hipLaunchKernelGGL((k_scan<Rec32, float, MODE>), dim3(nb), dim3(PC_WAVE), lds, st, A);
    } else {
        if (ext && qdtype == PC_F32)
            hipLaunchKernelGGL((k_scan<Rec64, float, MODE>), dim3(nb), dim3(PC_WAVE), lds, st, A);
        else
            hipLaunchKernelGGL((k_scan<Rec64, double, MODE>), dim3(nb), dim3(PC_WAVE), lds, st, A);
    }
    PC_HIP(hipGetLastError());
    if (tiled && ctx->knobs.debug) {  // development aid: how many queries each re-run stage received
        uint32_t c[32];
        PC_HIP(hipStreamSynchronize(st));
        PC_HIP(hipMemcpy(c, A.flag_count, sizeof(c), hipMemcpyDeviceToHost));
        fprintf(stderr, "[pc_amd] n=%lld k=%d tile=%dx%dx%d cap=%d tiles=%lld: wave-per-query %u, per-lane %u\n",
                (long long)A.nq, A.k, A.tile[0], A.tile[1], A.tile[2], A.cand_cap,
                (long long)(g->tile_list ? g->tile_count : (int64_t)A.ntile[0] * A.ntile[1] * A.ntile[2]), c[0], c[16]);
        if (c[28])  // only counted by builds with -DPC_COUNT
            fprintf(stderr, "[pc_amd]   per 64-query batch (%u batches): filter steps %.1f, insertion rounds %.1f, drains %.2f\n", c[28],
                    (double)c[24] / c[28], (double)c[25] / c[28], (double)c[27] / c[28]);
    }
    return PC_OK;
}

ScanArgs base_args(const pc_grid *g) {
    ScanArgs A{};
    A.compact_base = -1;
    A.flag_max = 255;
    A.flag_top = 0;
    A.gp = g->gp;
    A.rec = g->rec;
    A.ci = g->index();
    A.radius2 = -1.0;
    return A;
}

// ---- small reductions for the statistical filter -------------------------------------------------
// Two-level, fixed-shape tree: the result does not depend on scheduling.
constexpr int RED_BLOCKS = 512;

template <int WHAT>  // 0: sum of avg>0 and count of avg>=0 ; 1: sum over avg>0 of (avg-mean)^2
__global__ __launch_bounds__(256) void k_sor_reduce(const double *__restrict__ avg, int64_t n, double mean,
                                                     double *__restrict__ part) {
    double s = 0.0, c = 0.0;
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t b = (int64_t)blockIdx.x * per, e = b + per < n ? b + per : n;
    for (int64_t i = b + threadIdx.x; i < e; i += 256) {
        const double v = avg[i];
        if (WHAT == 0) {
            if (v > 0) s += v;
            if (v >= 0) c += 1.0;
        } else {
            if (v > 0) s += (v - mean) * (v - mean);
        }
    }
    __shared__ double red[2][256];
    red[0][threadIdx.x] = s;
    red[1][threadIdx.x] = c;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            red[0][threadIdx.x] += red[0][threadIdx.x + o];
            red[1][threadIdx.x] += red[1][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = red[0][0];
        part[2 * blockIdx.x + 1] = red[1][0];
    }
}

// one wave: lane j adds the partials j, j + 64, ... in that order, then a fixed shuffle tree
__global__ __launch_bounds__(64) void k_sor_reduce_final(const double *__restrict__ part, int nb, double *__restrict__ out) {
    double s = 0.0, c = 0.0;
    for (int b = threadIdx.x; b < nb; b += 64) {
        s += part[2 * b];
        c += part[2 * b + 1];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_down(s, o, 64);
        c += __shfl_down(c, o, 64);
    }
    if (threadIdx.x == 0) {
        out[0] = s;
        out[1] = c;
    }
}

__global__ __launch_bounds__(256) void k_sor_flags(const double *__restrict__ avg, int64_t n, double thr,
                                                    uint8_t *__restrict__ flags) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const double v = avg[i];
        flags[i] = (v > 0 && v < thr) ? 1 : 0;
    }
}

__global__ __launch_bounds__(256) void k_count_flags(const int32_t *__restrict__ cnt, int64_t n, int32_t nb,
                                                      uint8_t *__restrict__ flags) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) flags[i] = cnt[i] > nb ? 1 : 0;
}

// ---- the statistical filter's sums as a tree that does not depend on how the cloud is shared out -------
// One partial per chunk of PC_SOR_CHUNK consecutive SORTED positions (a block per chunk, fixed tree inside),
// then one wave over the partials in chunk order.  A rank of a multi-GPU job fills the partials of its own
// chunks; elementwise addition with the other ranks' zeros is exact, so every rank finishes the same array.
template <int WHAT>
__global__ __launch_bounds__(256) void k_sor_chunk_sums(const double *__restrict__ avg, const int32_t *__restrict__ perm,
                                                         int64_t avg_base, int64_t s_end, int64_t chunk0, double mean,
                                                         double *__restrict__ part) {
    const int64_t c = chunk0 + blockIdx.x;
    const int64_t b = c * PC_SOR_CHUNK, e = b + PC_SOR_CHUNK < s_end ? b + PC_SOR_CHUNK : s_end;
    double s = 0.0, cn = 0.0;
    for (int64_t i = b + threadIdx.x; i < e; i += 256) {
        // the value of sorted position i: compact rows (position - avg_base) or rows at the original index
        const double v = perm ? avg[perm[i]] : avg[i - avg_base];
        if (WHAT == 0) {
            if (v > 0) s += v;
            if (v >= 0) cn += 1.0;
        } else {
            if (v > 0) s += (v - mean) * (v - mean);
        }
    }
    __shared__ double red[2][256];
    red[0][threadIdx.x] = s;
    red[1][threadIdx.x] = cn;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            red[0][threadIdx.x] += red[0][threadIdx.x + o];
            red[1][threadIdx.x] += red[1][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        part[2 * c] = red[0][0];
        part[2 * c + 1] = red[1][0];
    }
}

// ---- the same sums, exact: no tree at all --------------------------------------------------------------
// Every term is turned into a 256-bit fixed-point number (2^-128 resolution: every bit of a double of 2^-76 or more;
// smaller terms lose the bits below 2^-128, each on its own, whatever its neighbours), cut into eight 32-bit digits,
// and the digits are added as integers -- per thread, per wave, then atomically into eight 64-bit words (room for
// 2^31 terms).  Integer addition does not care about order or grouping: one GPU, eight GPUs with any cut of the
// cloud, and a host restatement arrive at the same words, hence at the same mean and standard deviation, bit for bit.
// words[8] counts the rows with avg >= 0 (what = 0), words[9] the terms too large for the format (>= 2^96: the
// caller then takes the floating-point tree instead).
// v as floor(v * 2^128): first 32-bit digit `d` (0..5) and the three digits from there on; false (nothing set) for a
// value of 2^96 and more or no number at all
__device__ inline bool exact_digits(double v, int &d, uint32_t &d0, uint32_t &d1, uint32_t &d2) {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
    const int be = (int)(bits >> 52) & 0x7ff;
    unsigned long long m = (bits & ((1ull << 52) - 1)) | (be ? (1ull << 52) : 0ull);
    int sh = (be ? be : 1) - 1075 + 128;  // v = m * 2^(sh - 128)
    d = 0;
    d0 = d1 = d2 = 0;
    if (sh > 96 + 75) return false;  // v >= 2^96, or not a number
    if (sh < 0) {
        m = sh > -64 ? m >> (-sh) : 0ull;
        sh = 0;
    }
    d = sh >> 5;
    const int r = sh & 31;
    const unsigned long long lo = m << r;                      // low 64 bits of m << r
    const unsigned long long hi = r > 11 ? m >> (64 - r) : 0;  // the bits above them (m < 2^53)
    d0 = (uint32_t)lo;
    d1 = (uint32_t)(lo >> 32);
    d2 = (uint32_t)hi;
    return true;
}

constexpr int EXACT_BLOCKS = 1024, EXACT_AHEAD = 4;
template <int WHAT>
__global__ __launch_bounds__(256) void k_sor_exact(const double *__restrict__ avg, const int32_t *__restrict__ owned, int64_t m,
                                                    double mean, unsigned long long *__restrict__ partial /*[blocks][10]*/) {
    unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, cnt = 0, big = 0;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x; i0 < m; i0 += EXACT_AHEAD * stride) {
        double v[EXACT_AHEAD];
        int32_t o[EXACT_AHEAD];
#pragma unroll
        for (int u = 0; u < EXACT_AHEAD; u++) {  // all loads of the turn in flight together
            const int64_t i = i0 + u * stride;
            v[u] = i < m ? avg[i] : -1.0;
            o[u] = owned && i < m ? owned[i] : 0;
        }
#pragma unroll
        for (int u = 0; u < EXACT_AHEAD; u++) {
            const bool mine = i0 + u * stride < m && o[u] >= 0;  // (a slab's rows of a neighbouring rank carry -1)
            double term = v[u];
            if (WHAT == 0) {
                cnt += mine && v[u] >= 0 ? 1 : 0;
            } else {
                const double dv = v[u] - mean;
                term = dv * dv;
            }
            int d;
            uint32_t d0, d1, d2;
            const bool want = mine && v[u] > 0;
            const bool fits = exact_digits(want ? term : 0.0, d, d0, d1, d2);
            big += want && !fits ? 1 : 0;
            // The terms of one cloud lie within a few binades: as a rule every lane of the wave starts at the same
            // digit, and three plain additions do (digits of lanes without a term are zero and fit anywhere).
            const unsigned long long have = __ballot(want && fits);
            if (have == 0) continue;  // uniform
            const int dw = __builtin_amdgcn_readlane(d, (int)__builtin_ctzll(have));
            if (__ballot(want && fits && d != dw) == 0) {
                switch (dw) {  // uniform
                    case 0: acc[0] += d0; acc[1] += d1; acc[2] += d2; break;
                    case 1: acc[1] += d0; acc[2] += d1; acc[3] += d2; break;
                    case 2: acc[2] += d0; acc[3] += d1; acc[4] += d2; break;
                    case 3: acc[3] += d0; acc[4] += d1; acc[5] += d2; break;
                    case 4: acc[4] += d0; acc[5] += d1; acc[6] += d2; break;
                    default: acc[5] += d0; acc[6] += d1; acc[7] += d2; break;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; j++)
                    acc[j] += (unsigned long long)(j == d ? d0 : 0u) + (j == d + 1 ? d1 : 0u) + (j == d + 2 ? d2 : 0u);
            }
        }
    }
    __shared__ unsigned long long red[4][PC_SOR_EXACT_WORDS];
#pragma unroll
    for (int j = 0; j < 8; j++) {
#pragma unroll
        for (int o2 = 32; o2 > 0; o2 >>= 1) acc[j] += __shfl_down(acc[j], o2, 64);
    }
#pragma unroll
    for (int o2 = 32; o2 > 0; o2 >>= 1) {
        cnt += __shfl_down(cnt, o2, 64);
        big += __shfl_down(big, o2, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        const int w = threadIdx.x >> 6;
#pragma unroll
        for (int j = 0; j < 8; j++) red[w][j] = acc[j];
        red[w][8] = cnt;
        red[w][9] = big;
    }
    __syncthreads();
    if (threadIdx.x < PC_SOR_EXACT_WORDS)
        partial[(size_t)blockIdx.x * PC_SOR_EXACT_WORDS + threadIdx.x] =
            red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// words[j] += sum over the blocks' partial words (integers: any order); wave j of the workgroup sums word j
__global__ __launch_bounds__(64 * PC_SOR_EXACT_WORDS) void k_sor_exact_finish(const unsigned long long *__restrict__ partial,
                                                                               int nblocks, unsigned long long *__restrict__ words) {
    const int j = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned long long s = 0;
#pragma unroll 8
    for (int b = lane; b < nblocks; b += 64) s += partial[(size_t)b * PC_SOR_EXACT_WORDS + j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if (lane == 0) words[j] += s;
}

__global__ __launch_bounds__(256) void k_fill_slots(const int32_t *__restrict__ perm, int64_t s_begin, int64_t s_end,
                                                     int32_t *__restrict__ slot) {
    const int64_t s = s_begin + (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s < s_end) slot[perm[s]] = (int32_t)(s - s_begin);
}

// kept bit of original index owned[i], set with an atomic OR: the whole bit mask (n / 8 bytes) stays in L2, where
// n scattered byte stores would each rewrite a sector in HBM
__global__ __launch_bounds__(256) void k_mask_scatter(const double *__restrict__ avg, const int32_t *__restrict__ owned,
                                                       int64_t m, double thr, uint32_t *__restrict__ mask) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < m) {
        // (the removed points are marked, not the kept ones: a filter removes a few per cent, and every mark is an
        // atomic on a random word -- 10 M points: 0.37 ms marking the kept, 0.02 ms marking the removed)
        const double v = avg[i];
        const int32_t o = owned[i];
        if (!(v > 0 && v < thr) && o >= 0) atomicOr(&mask[(uint32_t)o >> 5], 1u << ((uint32_t)o & 31u));
    }
}

__global__ __launch_bounds__(256) void k_bits_to_flags(const uint32_t *__restrict__ mask, int64_t n, uint8_t *__restrict__ flags) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) flags[i] = ((mask[i >> 5] >> (i & 31)) & 1u) ^ 1u;  // (a set bit: removed)
}

int sor_reduce(pc_ctx *ctx, int what, const double *avg, int64_t n, double mean, double out[2], hipStream_t st) {
    out[0] = out[1] = 0.0;
    if (n <= 0) return PC_OK;
    Scratch part(ctx, sizeof(double) * 2 * (RED_BLOCKS + 1));
    if (!part.ok()) return pc_fail(ctx, PC_NOMEM, "reduction scratch");
    double *p = part.as<double>();
    if (what == 0)
        hipLaunchKernelGGL(k_sor_reduce<0>, dim3(RED_BLOCKS), dim3(256), 0, st, avg, n, mean, p);
    else
        hipLaunchKernelGGL(k_sor_reduce<1>, dim3(RED_BLOCKS), dim3(256), 0, st, avg, n, mean, p);
    hipLaunchKernelGGL(k_sor_reduce_final, dim3(1), dim3(64), 0, st, p, RED_BLOCKS, p + 2 * RED_BLOCKS);
    PC_HIP(hipGetLastError());
    double *hp = reinterpret_cast<double *>(ctx->pinned);
    PC_HIP(hipMemcpyAsync(hp, p + 2 * RED_BLOCKS, 2 * sizeof(double), hipMemcpyDeviceToHost, st));
    PC_HIP(hipStreamSynchronize(st));
    out[0] = hp[0];
    out[1] = hp[1];
    return PC_OK;
}

bool range_ok(const pc_grid *g, int64_t b, int64_t e) { return b >= 0 && e >= b && e <= g->n; }

}  // namespace

extern "C" {

int pc_knn(pc_ctx *ctx, const pc_grid *grid, const void *queries_dev, int qdtype, int64_t nq, int k, double radius,
           int32_t *out_idx_dev, void *out_d2_dev, int d2_dtype, int32_t *out_cnt_dev, void *stream) {
    PcLock lock(ctx, stream);  // calls on one context take turns (its block pool and read-back buffer are shared)
    if (!ctx || !grid || k < 1 || k > PC_MAX_K || nq < 0) return pc_fail(ctx, PC_BAD_ARG, "pc_knn");
    if (!queries_dev) return pc_knn_range(ctx, grid, 0, grid->n, k, radius, out_idx_dev, out_d2_dev, d2_dtype,
                                          out_cnt_dev, stream);
    PC_HIP(hipSetDevice(ctx->device));
    ScanArgs A = base_args(grid);
    A.s_begin = 0;
    A.nq = nq;
    A.k = k;
    A.radius2 = radius >= 0 ? radius * radius : -1.0;
    A.out_idx = out_idx_dev;
    A.out_d2 = out_d2_dev;
    A.d2_f64 = d2_dtype == PC_F64;
    A.out_cnt = out_cnt_dev;
    if (qdtype == grid->dtype && k <= 32 && nq >= 4096 && nq < INT32_MAX) {
        // many queries of the grid's own type: bin them into the grid (cell key, radix sort, records)
        // and let the tile kernel answer them like the grid's own points
        const size_t rb = grid->dtype == PC_F32 ? sizeof(Rec32) : sizeof(Rec64);
        // (The tile kernel finds the queries of a tile's rows of cells among the sorted query keys: with a directory of
        // the rows a search runs over the few dozen queries of one row instead of all of them -- some twenty dependent
        // loads less per tile; 2 M queries against 2 M points, k = 1: tile pass 610 -> 518 us.)
        const int64_t qnrows = (int64_t)grid->gp.dim[1] * grid->gp.dim[2];
        const bool with_rows = qnrows + 2 <= 16 * nq + 65536;
        Scratch qrec(ctx, rb * (size_t)nq), qkeys(ctx, sizeof(uint32_t) * (size_t)nq),
            qrows(ctx, with_rows ? sizeof(uint32_t) * (size_t)(qnrows + 2) : 0);
        if (!qrec.ok() || !qkeys.ok() || !qrows.ok()) return pc_fail(ctx, PC_NOMEM, "query records");
        PC_TRY(pc_bin_queries(ctx, grid, queries_dev, nq, qrec.p, qkeys.as<uint32_t>(), with_rows ? qrows.as<uint32_t>() : nullptr,
                              (hipStream_t)stream));
        A.qrec = qrec.p;
        A.qraw = queries_dev;
        A.qci = CellIndex{nullptr, qkeys.as<uint32_t>(), (uint32_t)nq, with_rows ? qrows.as<uint32_t>() : nullptr,
                          (uint32_t)grid->gp.dim[0]};
        return launch_scan<M_KNN>(ctx, grid, qdtype, A, (hipStream_t)stream);
    }
    A.queries = queries_dev;
    return launch_scan<M_KNN>(ctx, grid, qdtype, A, (hipStream_t)stream);
}

int pc_knn_range(pc_ctx *ctx, const pc_grid *grid, int64_t s_begin, int64_t s_end, int k, double radius,
                 int32_t *out_idx_dev, void *out_d2_dev, int d2_dtype, int32_t *out_cnt_dev, void *stream) {
    PcLock lock(ctx, stream);  // calls on one context take turns (its block pool and read-back buffer are shared)
    if (!ctx || !grid || k < 1 || k > PC_MAX_K || !range_ok(grid, s_begin, s_end))
        return pc_fail(ctx, PC_BAD_ARG, "pc_knn_range");
    PC_HIP(hipSetDevice(ctx->device));
    ScanArgs A = base_args(grid);
    A.s_begin = s_begin;
    A.nq = s_end - s_begin;
    A.k = k;
    A.radius2 = radius >= 0 ? radius * radius : -1.0;
    A.out_idx = out_idx_dev;
    A.out_d2 = out_d2_dev;
    A.d2_f64 = d2_dtype == PC_F64;
    A.out_cnt = out_cnt_dev;
    return launch_scan<M_KNN>(ctx, grid, grid->dtype, A, (hipStream_t)stream);
}

int pc_sor_avg(pc_ctx *ctx, const pc_grid *grid, int64_t s_begin, int64_t s_end, int k, double *out_avg_dev,
               void *stream) {
    PcLock lock(ctx, stream);  // calls on one context take turns (its block pool and read-back buffer are shared)
    if (!ctx || !grid || k < 1 || k > PC_MAX_K || !range_ok(grid, s_begin, s_end) || (s_end > s_begin && !out_avg_dev))
        return pc_fail(ctx, PC_BAD_ARG, "pc_sor_avg");
    PC_HIP(hipSetDevice(ctx->device));
    ScanArgs A = base_args(grid);
    A.s_begin = s_begin;
    A.nq = s_end - s_begin;
    A.k = k;
    A.out_f64 = out_avg_dev;
    return launch_scan<M_SOR>(ctx, grid, grid->dtype, A, (hipStream_t)stream);
}

int pc_knn_fused(pc_ctx *ctx, const pc_grid *grid, int64_t s_begin, int64_t s_end, int k, double radius,
                 int32_t *out_idx_dev, void *out_d2_dev, int d2_dtype, int32_t *out_cnt_dev, double *out_avg_dev,
                 double *out_normals_dev, const double *prior_dev, void *stream) {
    PcLock lock(ctx, stream);  // calls on one context take turns (its block pool and read-back buffer are shared)
    if (!ctx || !grid || k < 1 || k > PC_MAX_K || !range_ok(grid, s_begin, s_end))
        return pc_fail(ctx, PC_BAD_ARG, "pc_knn_fused");
    PC_HIP(hipSetDevice(ctx->device));
    ScanArgs A = base_args(grid);
    A.s_begin = s_begin;
    A.nq = s_end - s_begin;
    A.k = k;
    A.radius2 = radius >= 0 ? radius * radius : -1.0;
    A.out_idx = out_idx_dev;
    A.out_d2 = out_d2_dev;
    A.d2_f64 = d2_dtype == PC_F64;
    A.out_cnt = out_cnt_dev;
    A.out_f64 = out_avg_dev;
    A.out_nrm = out_normals_dev;
    A.prior = prior_dev;
    return launch_scan<M_ALL>(ctx, grid, grid->dtype, A, (hipStream_t)stream);
}

int pc_sor_sum(pc_ctx *ctx, const double *avg_dev, int64_t n, double *sum_pos, int64_t *n_valid, void *stream) {
    PcLock lock(ctx, stream);  // calls on one context take turns (its block pool and read-back buffer are shared)
    if (!ctx || n < 0 || (n > 0 && !avg_dev) || !sum_pos || !n_valid) return pc_fail(ctx, PC_BAD_ARG, "pc_sor_sum");
    PC_HIP(hipSetDevice(ctx->device));
    double o[2];
    PC_TRY(sor_reduce(ctx, 0, avg_dev, n, 0.0, o, (hipStream_t)stream));
    *sum_pos = o[0];
    *n_valid = (int64_t)o[1];
    return PC_OK;
}

int pc_sor_sqsum(pc_ctx *ctx, const double *avg_dev, int64_t n, double mean, double *sq, void *stream) {
    PcLock lock(ctx, stream);  // calls on one context take turns (its block pool and read-back buffer are shared)
    if (!ctx || n < 0 || (n > 0 && !avg_dev) || !sq) return pc_fail(ctx, PC_BAD_ARG, "pc_sor_sqsum");
    PC_HIP(hipSetDevice(ctx->device));
    double o[2];
    PC_TRY(sor_reduce(ctx, 1, avg_dev, n, mean, o, (hipStream_t)stream));
    *sq = o[0];
    return PC_OK;
}

int pc_sor_exact_sums(pc_ctx *ctx, const double *avg_dev, const int32_t *owned_dev, int64_t m, int what, double mean,
                      int64_t *words_dev, void *stream) {
    PcLock lock(ctx, stream);
    if (!ctx || m < 0 || (m > 0 && !avg_dev) || !words_dev || (what != 0 && what != 1) || ((uintptr_t)words_dev & 7u))
        return pc_fail(ctx, PC_BAD_ARG, "pc_sor_exact_sums");
    PC_HIP(hipSetDevice(ctx->device));
    if (m == 0) return PC_OK;
    const int blocks = (int)std::min<int64_t>(EXACT_BLOCKS, (m + 255) / 256);
    Scratch part(ctx, sizeof(unsigned long long) * PC_SOR_EXACT_WORDS * (size_t)blocks);
    if (!part.ok()) return pc_fail(ctx, PC_NOMEM, "exact sums scratch");
    unsigned long long *w = reinterpret_cast<unsigned long long *>(words_dev), *p = part.as<unsigned long long>();
    if (what == 0)
        hipLaunchKernelGGL(k_sor_exact<0>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, avg_dev, owned_dev, m, mean, p);
    else
        hipLaunchKernelGGL(k_sor_exact<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, avg_dev, owned_dev, m, mean, p);
    hipLaunchKernelGGL(k_sor_exact_finish, dim3(1), dim3(64 * PC_SOR_EXACT_WORDS), 0, (hipStream_t)stream, p, blocks, w);
    PC_HIP(hipGetLastError());
    return PC_OK;
}

int pc_sor_select(pc_ctx *ctx, const double *avg_dev, int64_t n, double threshold, uint8_t *out_mask_dev,
                  int32_t *out_kept_dev, int64_t *n_kept, void *stream) {
    PcLock lock(ctx, stream);  // calls on one context take turns (its block pool and read-back buffer are shared)
    if (!ctx || n < 0 || (n > 0 && !avg_dev) || !n_kept) return pc_fail(ctx, PC_BAD_ARG, "pc_sor_select");
    PC_HIP(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    *n_kept = 0;
    if (n == 0) return PC_OK;
    Scratch tmp(ctx, (size_t)n);
    if (!tmp.ok()) return pc_fail(ctx, PC_NOMEM, "mask scratch");
    uint8_t *flags = out_mask_dev ? out_mask_dev : tmp.as<uint8_t>();
    hipLaunchKernelGGL(k_sor_flags, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, avg_dev, n, threshold, flags);
    PC_HIP(hipGetLastError());
    return pc_compact_flags(ctx, flags, n, out_kept_dev, n_kept, st);
}

int pc_knn_fused_compact(pc_ctx *ctx, const pc_grid *grid, int64_t s_begin, int64_t s_end, int k, double radius,
                         int32_t *out_idx_dev, void *out_d2_dev, int d2_dtype, int32_t *out_cnt_dev, double *out_avg_dev,
                         double *out_normals_dev, int32_t *out_owned_dev, void *stream) {
    PcLock lock(ctx, stream);
    if (!ctx || !grid || k < 1 || k > PC_MAX_K || !range_ok(grid, s_begin, s_end))
        return pc_fail(ctx, PC_BAD_ARG, "pc_knn_fused_compact");
    PC_HIP(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    const int64_t m = s_end - s_begin;
    if (m == 0) return PC_OK;
    // Output row of a query: its sorted position minus s_begin.  The kernels of a single-level index know that
    // position; the coarser levels of a multi-level index (uneven clouds) sort the points differently and look the row
    // up by original index (map filled for the queries of the range; other entries are never read).
    const bool by_position = grid->coarse == nullptr && !grid->uneven;
    Scratch slot(ctx, by_position ? 0 : sizeof(int32_t) * (size_t)pc_index_space(grid));
    if (!slot.ok()) return pc_fail(ctx, PC_NOMEM, "output rows");
    if (!by_position)
        hipLaunchKernelGGL(k_fill_slots, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, grid->perm, s_begin, s_end,
                           slot.as<int32_t>());
    if (out_owned_dev)
        PC_HIP(hipMemcpyAsync(out_owned_dev, grid->perm + s_begin, sizeof(int32_t) * (size_t)m, hipMemcpyDeviceToDevice, st));
    ScanArgs A = base_args(grid);
    A.s_begin = s_begin;
    A.nq = m;
    A.k = k;
    A.radius2 = radius >= 0 ? radius * radius : -1.0;
    A.out_idx = out_idx_dev;
    A.out_d2 = out_d2_dev;
    A.d2_f64 = d2_dtype == PC_F64;
    A.out_cnt = out_cnt_dev;
    A.out_f64 = out_avg_dev;
    A.out_nrm = out_normals_dev;
    A.out_slot = by_position ? nullptr : slot.as<int32_t>();
    A.compact_base = by_position ? s_begin : -1;
    if (!out_avg_dev && !out_normals_dev) return launch_scan<M_KNN>(ctx, grid, grid->dtype, A, st);  // rows only
    return launch_scan<M_ALL>(ctx, grid, grid->dtype, A, st);
}

int64_t pc_sor_chunks(int64_t n) { return n > 0 ? (n + PC_SOR_CHUNK - 1) / PC_SOR_CHUNK : 0; }

int pc_sor_chunk_sums(pc_ctx *ctx, const pc_grid *grid, const double *avg_dev, int compact, int64_t s_begin, int64_t s_end,
                      int what, double mean, double *partials_dev, void *stream) {
    PcLock lock(ctx, stream);
    if (ctx && grid && range_ok(grid, s_begin, s_end) && s_end == s_begin) return PC_OK;  // (a rank without a chunk of its own)
    if (!ctx || !grid || !range_ok(grid, s_begin, s_end) || (s_end > s_begin && (!avg_dev || !partials_dev)) ||
        s_begin % PC_SOR_CHUNK != 0 || (s_end % PC_SOR_CHUNK != 0 && s_end != grid->n) || (what != 0 && what != 1))
        return pc_fail(ctx, PC_BAD_ARG, "pc_sor_chunk_sums: the range must consist of whole chunks");
    PC_HIP(hipSetDevice(ctx->device));
    if (s_end == s_begin) return PC_OK;
    const int64_t c0 = s_begin / PC_SOR_CHUNK, c1 = (s_end + PC_SOR_CHUNK - 1) / PC_SOR_CHUNK;
    const int32_t *perm = compact ? nullptr : grid->perm;
    if (what == 0)
        hipLaunchKernelGGL(k_sor_chunk_sums<0>, dim3((unsigned)(c1 - c0)), dim3(256), 0, (hipStream_t)stream, avg_dev, perm,
                           s_begin, s_end, c0, mean, partials_dev);
    else
        hipLaunchKernelGGL(k_sor_chunk_sums<1>, dim3((unsigned)(c1 - c0)), dim3(256), 0, (hipStream_t)stream, avg_dev, perm,
                           s_begin, s_end, c0, mean, partials_dev);
    PC_HIP(hipGetLastError());
    return PC_OK;
}

int pc_sor_chunk_finish(pc_ctx *ctx, const double *partials_dev, int64_t nchunks, double out[2], void *stream) {
    PcLock lock(ctx, stream);
    if (!ctx || nchunks < 0 || nchunks > INT32_MAX || (nchunks > 0 && !partials_dev) || !out)
        return pc_fail(ctx, PC_BAD_ARG, "pc_sor_chunk_finish");
    PC_HIP(hipSetDevice(ctx->device));
    out[0] = out[1] = 0.0;
    if (nchunks == 0) return PC_OK;
    hipStream_t st = (hipStream_t)stream;
    Scratch res(ctx, 256);
    if (!res.ok()) return pc_fail(ctx, PC_NOMEM, "reduction scratch");
    hipLaunchKernelGGL(k_sor_reduce_final, dim3(1), dim3(64), 0, st, partials_dev, (int)nchunks, res.as<double>());
    PC_HIP(hipGetLastError());
    double *hp = reinterpret_cast<double *>(ctx->pinned);
    PC_HIP(hipMemcpyAsync(hp, res.p, 2 * sizeof(double), hipMemcpyDeviceToHost, st));
    PC_HIP(hipStreamSynchronize(st));
    out[0] = hp[0];
    out[1] = hp[1];
    return PC_OK;
}

int64_t pc_mask_bytes(int64_t n) { return n > 0 ? ((n + 31) / 32) * 4 : 0; }

int pc_sor_mask_scatter(pc_ctx *ctx, const double *avg_dev, const int32_t *owned_dev, int64_t m, double threshold,
                        uint8_t *mask_dev, void *stream) {
    PcLock lock(ctx, stream);
    if (!ctx || m < 0 || (m > 0 && (!avg_dev || !owned_dev || !mask_dev)) || ((uintptr_t)mask_dev & 3u))
        return pc_fail(ctx, PC_BAD_ARG, "pc_sor_mask_scatter");
    PC_HIP(hipSetDevice(ctx->device));
    if (m == 0) return PC_OK;
    hipLaunchKernelGGL(k_mask_scatter, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, (hipStream_t)stream, avg_dev,
                       owned_dev, m, threshold, reinterpret_cast<uint32_t *>(mask_dev));
    PC_HIP(hipGetLastError());
    return PC_OK;
}

int pc_mask_select(pc_ctx *ctx, const uint8_t *mask_dev, int64_t n, int32_t *out_kept_dev, int64_t *n_kept, void *stream) {
    PcLock lock(ctx, stream);
    if (!ctx || n < 0 || (n > 0 && !mask_dev) || !n_kept || ((uintptr_t)mask_dev & 3u)) return pc_fail(ctx, PC_BAD_ARG, "pc_mask_select");
    PC_HIP(hipSetDevice(ctx->device));
    *n_kept = 0;
    if (n == 0) return PC_OK;
    Scratch flags(ctx, (size_t)n);
    if (!flags.ok()) return pc_fail(ctx, PC_NOMEM, "mask scratch");
    hipLaunchKernelGGL(k_bits_to_flags, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const uint32_t *>(mask_dev), n, flags.as<uint8_t>());
    PC_HIP(hipGetLastError());
    return pc_compact_flags(ctx, flags.as<uint8_t>(), n, out_kept_dev, n_kept, (hipStream_t)stream);
}

int pc_count_select(pc_ctx *ctx, const int32_t *cnt_dev, int64_t n, int32_t nb_points, uint8_t *out_mask_dev,
                    int32_t *out_kept_dev, int64_t *n_kept, void *stream) {
    PcLock lock(ctx, stream);
    if (!ctx || n < 0 || (n > 0 && !cnt_dev) || !n_kept) return pc_fail(ctx, PC_BAD_ARG, "pc_count_select");
    PC_HIP(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    *n_kept = 0;
    if (n == 0) return PC_OK;
    Scratch tmp(ctx, (size_t)n);
    if (!tmp.ok()) return pc_fail(ctx, PC_NOMEM, "mask scratch");
    uint8_t *flags = out_mask_dev ? out_mask_dev : tmp.as<uint8_t>();
    hipLaunchKernelGGL(k_count_flags, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, cnt_dev, n, nb_points, flags);
    PC_HIP(hipGetLastError());
    return pc_compact_flags(ctx, flags, n, out_kept_dev, n_kept, st);
}

int pc_normals(pc_ctx *ctx, const pc_grid *grid, int64_t s_begin, int64_t s_end, int k, double radius,
               const double *prior_dev, double *out_normals_dev, void *stream) {
    PcLock lock(ctx, stream);  // calls on one context take turns (its block pool and read-back buffer are shared)
    if (!ctx || !grid || k < 1 || k > PC_MAX_K || !range_ok(grid, s_begin, s_end) ||
        (s_end > s_begin && !out_normals_dev))
        return pc_fail(ctx, PC_BAD_ARG, "pc_normals");
    PC_HIP(hipSetDevice(ctx->device));
    ScanArgs A = base_args(grid);
    A.s_begin = s_begin;
    A.nq = s_end - s_begin;
    A.k = k;
    A.radius2 = radius > 0 ? radius * radius : -1.0;
    A.out_f64 = out_normals_dev;
    A.prior = prior_dev;
    return launch_scan<M_NORMALS>(ctx, grid, grid->dtype, A, (hipStream_t)stream);
}

int pc_radius_count(pc_ctx *ctx, const pc_grid *grid, const void *queries_dev, int qdtype, int64_t nq,
                    int64_t s_begin, int64_t s_end, double radius, int32_t *out_cnt_dev, void *stream) {
    PcLock lock(ctx, stream);  // calls on one context take turns (its block pool and read-back buffer are shared)
    if (!ctx || !grid || !(radius >= 0) || !out_cnt_dev) return pc_fail(ctx, PC_BAD_ARG, "pc_radius_count");
    PC_HIP(hipSetDevice(ctx->device));
    ScanArgs A = base_args(grid);
    A.k = 1;
    A.radius2 = radius * radius;
    A.out_cnt = out_cnt_dev;
    if (queries_dev) {
        if (nq < 0) return pc_fail(ctx, PC_BAD_ARG, "pc_radius_count");
        A.queries = queries_dev;
        A.nq = nq;
        return launch_scan<M_COUNT>(ctx, grid, qdtype, A, (hipStream_t)stream);
    }
    if (!range_ok(grid, s_begin, s_end)) return pc_fail(ctx, PC_BAD_ARG, "pc_radius_count");
    A.s_begin = s_begin;
    A.nq = s_end - s_begin;
    return launch_scan<M_COUNT>(ctx, grid, grid->dtype, A, (hipStream_t)stream);
}

int pc_nn_distance(pc_ctx *ctx, const pc_grid *grid, int64_t s_begin, int64_t s_end, double *out_dev, void *stream) {
    PcLock lock(ctx, stream);  // calls on one context take turns (its block pool and read-back buffer are shared)
    if (!ctx || !grid || !range_ok(grid, s_begin, s_end) || (s_end > s_begin && !out_dev))
        return pc_fail(ctx, PC_BAD_ARG, "pc_nn_distance");
    PC_HIP(hipSetDevice(ctx->device));
    ScanArgs A = base_args(grid);
    A.s_begin = s_begin;
    A.nq = s_end - s_begin;
    A.k = 2;
    A.out_f64 = out_dev;
    return launch_scan<M_NNDIST>(ctx, grid, grid->dtype, A, (hipStream_t)stream);
}

}  // extern "C"
