// The index build of an evenly filled fp32 cloud in two moves of the 16-byte records instead of three passes
// over (key, index) pairs and a random gather: (1) a stable partition of the points into buckets of a few
// thousand records by the high part of the cell key -- tiles of 8192 points, ordered by bucket inside LDS and
// written as runs of whole records --, (2) one workgroup per bucket that puts the bucket's records in final order
// through LDS and writes the bucket's part of the cell table and of the occupancy statistics.  Every global
// access is a coalesced run: 16-byte accesses scattered one by one cost 2.4 x their bytes on this memory system
// (measured, reads and writes alike), which is what made the gather the longest kernel of the old build.
// All of it is HBM-bound integer work; there is nothing for the matrix cores here.
#include "pc_common.h"

namespace {

// ---- cell keys ------------------------------------------------------------------------------------------
// The key of k_cell_keys32 (pc_grid.hip): cell index per axis = clamp(floor((p - o) * s)) in fp64, x fastest,
// one more bit below the cell for the upper x-half (x_half_of).
__device__ inline uint32_t cell_key_exact(float x, float y, float z, const GridParams &gp, int halfbit) {
    const double p[3] = {(double)x, (double)y, (double)z};
    uint32_t c[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const double t = floor((p[a] - gp.o[a]) * gp.s);
        c[a] = !(t > 0.0) ? 0u : (t >= (double)gp.dim[a] ? (uint32_t)(gp.dim[a] - 1) : (uint32_t)t);
    }
    uint32_t key = (c[2] * (uint32_t)gp.dim[1] + c[1]) * (uint32_t)gp.dim[0] + c[0];
    if (halfbit) key = (key << 1) | x_half_of(p[0], gp, (int)c[0]);
    return key;
}

// The same key from fp32 arithmetic wherever that provably gives the cell (and x-half) of the fp64 expression:
// with o an fp32 value, t = fl32(fl32(x - o) * fl32(s)) differs from (x - o) * s by less than 2^-22 of it (three
// roundings of 2^-24 each), so the two floors agree unless t lies within that margin of an integer -- or of the
// half mark, for the x-half bit; lanes that do take the fp64 expression (about one in ten thousand).
struct KeyF32 {
    float o[3];
    float s;
    int ok;  // 0: the origin is no fp32 value, every lane takes the fp64 expression
};
__device__ inline uint32_t cell_key(float x, float y, float z, const GridParams &gp, const KeyF32 &kf, int halfbit) {
    const float p[3] = {x, y, z};
    uint32_t c[3];
    bool safe = kf.ok != 0;
    float fx = 0.f;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float t = (p[a] - kf.o[a]) * kf.s;
        const float fl = floorf(t);
        const float f = t - fl;                   // (exact: both lie in one binade or fl is 0)
        const float eps = (fl + 2.0f) * 2.4e-7f;  // > 2^-22 (fl + 1)
        safe = safe && f > eps && f < 1.0f - eps && fl >= 0.f && fl < 1.0e9f;
        const int ci = (int)fl;
        c[a] = ci >= gp.dim[a] ? (uint32_t)(gp.dim[a] - 1) : (uint32_t)ci;
        if (a == 0) {
            fx = f;
            if (halfbit) safe = safe && fabsf(f - 0.5f) > eps && ci < gp.dim[0];
        }
    }
    if (!safe) return cell_key_exact(x, y, z, gp, halfbit);
    uint32_t key = (c[2] * (uint32_t)gp.dim[1] + c[1]) * (uint32_t)gp.dim[0] + c[0];
    if (halfbit) key = (key << 1) | (fx >= 0.5f ? 1u : 0u);
    return key;
}

// key / width for a bucket width that is no power of two: float estimate, corrected by the integer remainder
struct BucketDiv {
    uint32_t width;
    float inv;
};
__device__ inline uint32_t bucket_of(uint32_t key, const BucketDiv &bd) {
    uint32_t b = (uint32_t)((float)key * bd.inv);
    const int32_t r = (int32_t)(key - b * bd.width);
    b += r < 0 ? -1 : (r >= (int32_t)bd.width ? 1 : 0);
    return b;
}

constexpr int PT = 512;              // threads of the partition kernels
constexpr int PROUNDS = 16;          // points per thread and tile
constexpr int PTILE = PT * PROUNDS;  // 8192 points per workgroup
constexpr int PWAVES = PT / 64;
constexpr int MAXB = 2048;           // buckets at most (the partition's per-wave tables: PWAVES * MAXB words of LDS)

// four consecutive points as three 16-byte loads (16-byte aligned array, all four points inside it)
struct Points4 {
    float x[4], y[4], z[4];
};
__device__ inline Points4 load4(const float *__restrict__ xyz, int64_t group) {
    const float4 *p = reinterpret_cast<const float4 *>(xyz) + 3 * group;
    const float4 a = p[0], b = p[1], c = p[2];
    Points4 r;
    r.x[0] = a.x; r.y[0] = a.y; r.z[0] = a.z;
    r.x[1] = a.w; r.y[1] = b.x; r.z[1] = b.y;
    r.x[2] = b.z; r.y[2] = b.w; r.z[2] = c.x;
    r.x[3] = c.y; r.y[3] = c.z; r.z[3] = c.w;
    return r;
}

// ---- 1: per-tile bucket histogram, tile-major ------------------------------------------------------------
__global__ __launch_bounds__(PT) void k_bucket_hist(const float *__restrict__ xyz, int64_t n, GridParams gp, KeyF32 kf,
                                                    int halfbit, BucketDiv bd, int nb, int wide,
                                                    uint32_t *__restrict__ hist /*[tiles][nb]*/) {
    __shared__ uint32_t h[MAXB];
    for (int b = threadIdx.x; b < nb; b += PT) h[b] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * PTILE;
#pragma unroll 2
    for (int r = 0; r < PROUNDS / 4; r++) {
        const int64_t i = base + 4 * (r * PT + threadIdx.x);  // first of the thread's four points
        if (wide && i + 4 <= n) {
            const Points4 q = load4(xyz, i >> 2);
#pragma unroll
            for (int j = 0; j < 4; j++) atomicAdd(&h[bucket_of(cell_key(q.x[j], q.y[j], q.z[j], gp, kf, halfbit), bd)], 1u);
        } else {
            for (int64_t t = i; t < n && t < i + 4; t++)
                atomicAdd(&h[bucket_of(cell_key(xyz[3 * t], xyz[3 * t + 1], xyz[3 * t + 2], gp, kf, halfbit), bd)], 1u);
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < nb; b += PT) hist[(size_t)blockIdx.x * nb + b] = h[b];
}

// ---- 2: column scan of hist[tiles][nb]: every (bucket, tile) pair's first slot, buckets in order, tiles in
// order inside a bucket (what makes the partition stable) ------------------------------------------------------
constexpr int CG = 32;  // tiles per group
__global__ __launch_bounds__(256) void k_col_group_sums(const uint32_t *__restrict__ hist, int tiles, int nb,
                                                        uint32_t *__restrict__ gsum) {
    const int d = blockIdx.x * 256 + threadIdx.x, g = blockIdx.y;
    if (d >= nb) return;
    const int b0 = g * CG, b1 = min(b0 + CG, tiles);
    uint32_t s = 0;
    for (int b = b0; b < b1; b++) s += hist[(size_t)b * nb + d];
    gsum[(size_t)g * nb + d] = s;
}
// per bucket: exclusive running sum over the groups (relative to the bucket's start) and the bucket's total
__global__ __launch_bounds__(256) void k_col_scan_groups(uint32_t *__restrict__ gsum, int ngroups, int nb,
                                                         uint32_t *__restrict__ total) {
    const int d = blockIdx.x * 256 + threadIdx.x;
    if (d >= nb) return;
    uint32_t run = 0;
    for (int g = 0; g < ngroups; g++) {
        const uint32_t t = gsum[(size_t)g * nb + d];
        gsum[(size_t)g * nb + d] = run;
        run += t;
    }
    total[d] = run;
}
// one workgroup: bucket_start = exclusive scan of the bucket totals (nb <= MAXB), bucket_start[nb] = n
__global__ __launch_bounds__(1024) void k_bucket_starts(const uint32_t *__restrict__ total, int nb,
                                                        uint32_t *__restrict__ bucket_start) {
    __shared__ uint32_t part[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int PER = MAXB / 1024;
    uint32_t c[PER], tot = 0;
#pragma unroll
    for (int j = 0; j < PER; j++) {
        const int d = threadIdx.x * PER + j;
        c[j] = d < nb ? total[d] : 0u;
        tot += c[j];
    }
    uint32_t x = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    if (lane == 63) part[wave] = x;
    __syncthreads();
    uint32_t off = x - tot;
    for (int w = 0; w < wave; w++) off += part[w];
#pragma unroll
    for (int j = 0; j < PER; j++) {
        const int d = threadIdx.x * PER + j;
        if (d <= nb) bucket_start[d] = off;
        off += c[j];
    }
}
__global__ __launch_bounds__(256) void k_col_apply(uint32_t *__restrict__ hist, int tiles, int nb,
                                                   const uint32_t *__restrict__ gsum,
                                                   const uint32_t *__restrict__ bucket_start) {
    const int d = blockIdx.x * 256 + threadIdx.x, g = blockIdx.y;
    if (d >= nb) return;
    const int b0 = g * CG, b1 = min(b0 + CG, tiles);
    uint32_t run = gsum[(size_t)g * nb + d] + bucket_start[d];
    for (int b = b0; b < b1; b++) {
        const uint32_t t = hist[(size_t)b * nb + d];
        hist[(size_t)b * nb + d] = run;
        run += t;
    }
}

// ---- 3: stable partition of a tile of 8192 points into the buckets -----------------------------------------
// Wave w owns the contiguous eighth [w * 1024, (w + 1) * 1024) of the tile and ranks it in a table of its own:
// word = (lane tag << 16) | running count of the bucket.  Lanes of one round that meet in a bucket take their
// turns by an LDS minimum on the tag (lowest lane first), so the ranks follow the input order whatever the
// hardware's timing.  The tile is then ordered by bucket inside LDS, half a tile at a time, and leaves as runs of
// whole records.  Workgroup ids are dealt so that one XCD (its own L2) takes consecutive tiles: the runs of
// neighbouring tiles are neighbours in memory.
__global__ __launch_bounds__(PT) void k_bucket_scatter(const float *__restrict__ xyz, int64_t n, GridParams gp, KeyF32 kf,
                                                       int halfbit, BucketDiv bd, int nb, int wide,
                                                       const uint32_t *__restrict__ off /*[tiles][nb]*/,
                                                       Rec32 *__restrict__ out, int tiles) {
    constexpr int HALF = PTILE / 2;
    __shared__ __attribute__((aligned(16))) uint32_t stage[HALF * 4];  // 64 KB; the ranking tables live here first
    __shared__ uint32_t gbase[MAXB];
    __shared__ uint16_t dstart[MAXB];
    __shared__ uint32_t part[PWAVES];
    uint32_t *tab = stage;  // [PWAVES][nb]
    const int per = (tiles + 7) / 8;
    const int tile = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per || tile >= tiles) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t tbase = (int64_t)tile * PTILE;
    const int64_t wbase = tbase + (int64_t)wave * (PTILE / PWAVES);
    float px[PROUNDS], py[PROUNDS], pz[PROUNDS];
    uint32_t dr[PROUNDS];  // bucket << 16 | rank inside the wave's part, later bucket << 16 | slot in the tile
    if (wide && wbase + PTILE / PWAVES <= n) {
        // the wave's 1024 points as 16-byte loads, turned into one point per lane and round through LDS (the
        // wave's own 6 KB of the staging area, half of its points at a time)
        float *mine = reinterpret_cast<float *>(stage) + wave * (512 * 3);
        const float4 *src = reinterpret_cast<const float4 *>(xyz + 3 * wbase);
#pragma unroll
        for (int hh = 0; hh < 2; hh++) {
            float4 v[6];
#pragma unroll
            for (int t = 0; t < 6; t++) v[t] = src[hh * 384 + t * 64 + lane];
#pragma unroll
            for (int t = 0; t < 6; t++) reinterpret_cast<float4 *>(mine)[t * 64 + lane] = v[t];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 8; r++) {
                px[hh * 8 + r] = mine[3 * (r * 64 + lane)];
                py[hh * 8 + r] = mine[3 * (r * 64 + lane) + 1];
                pz[hh * 8 + r] = mine[3 * (r * 64 + lane) + 2];
            }
            __builtin_amdgcn_wave_barrier();
        }
    } else {
#pragma unroll
        for (int r = 0; r < PROUNDS; r++) {
            const int64_t i = wbase + r * 64 + lane;
            const bool v = i < n;
            px[r] = v ? xyz[3 * i] : 0.f;
            py[r] = v ? xyz[3 * i + 1] : 0.f;
            pz[r] = v ? xyz[3 * i + 2] : 0.f;
        }
    }
    __syncthreads();
    for (int b = tid; b < PWAVES * nb; b += PT) tab[b] = 0xffff0000u;
    __syncthreads();
    volatile uint32_t *T = tab + wave * nb;
#pragma unroll
    for (int r = 0; r < PROUNDS; r++) {
        const bool valid = wbase + r * 64 + lane < n;
        const uint32_t d = valid ? bucket_of(cell_key(px[r], py[r], pz[r], gp, kf, halfbit), bd) : 0u;
        bool pending = valid;
        const uint32_t seen = valid ? (T[d] & 0xffffu) : 0u;
        uint32_t it = 0, rank = 0;
        while (__any(pending)) {  // (as many turns as the most crowded bucket of the round has lanes: 64 at most)
            if (pending) atomicMin((uint32_t *)&T[d], ((uint32_t)lane << 16) | (seen + it));
            if (pending) {
                const uint32_t w = T[d];
                if ((w >> 16) == (uint32_t)lane) {
                    rank = seen + it;
                    T[d] = 0xffff0000u | (seen + it + 1);
                    pending = false;
                }
            }
            it++;
        }
        dr[r] = (d << 16) | rank;
    }
    __syncthreads();
    // per bucket: the waves' counts -> start of every wave's run inside the bucket's run; then the buckets' runs
    {
        constexpr int PER = MAXB / PT;  // 4
        uint32_t c[PER], tot = 0;
#pragma unroll
        for (int j = 0; j < PER; j++) {
            const int b = tid * PER + j;
            uint32_t run = 0;
            if (b < nb) {
#pragma unroll
                for (int w = 0; w < PWAVES; w++) {
                    const uint32_t cw = tab[w * nb + b] & 0xffffu;
                    tab[w * nb + b] = run;
                    run += cw;
                }
            }
            c[j] = run;
            tot += run;
        }
        uint32_t x = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(x, o, 64);
            if (lane >= o) x += y;
        }
        if (lane == 63) part[wave] = x;
        __syncthreads();
        uint32_t o2 = x - tot;
        for (int w = 0; w < wave; w++) o2 += part[w];
#pragma unroll
        for (int j = 0; j < PER; j++) {
            const int b = tid * PER + j;
            if (b < nb) {
                dstart[b] = (uint16_t)o2;
                gbase[b] = off[(size_t)tile * nb + b];
            }
            o2 += c[j];
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < PROUNDS; r++) {
        const uint32_t d = dr[r] >> 16;
        dr[r] = (d << 16) | ((uint32_t)dstart[d] + tab[wave * nb + d] + (dr[r] & 0xffffu));
    }
    __syncthreads();  // the tables are dead: their memory stages the records now
    const int64_t left = n - tbase;
    const int count = left < PTILE ? (int)left : PTILE;
#pragma unroll 1
    for (int h = 0; h < 2; h++) {
#pragma unroll
        for (int r = 0; r < PROUNDS; r++) {
            const int local = wave * (PTILE / PWAVES) + r * 64 + lane;
            const uint32_t slot = dr[r] & 0xffffu;
            if (local < count && (int)(slot / HALF) == h) {
                uint4 v;
                v.x = __float_as_uint(px[r]);
                v.y = __float_as_uint(py[r]);
                v.z = __float_as_uint(pz[r]);
                v.w = ((dr[r] >> 16) << 13) | (uint32_t)local;  // bucket (11 bits) and place in the tile (13 bits)
                *reinterpret_cast<uint4 *>(&stage[(slot - h * HALF) * 4]) = v;
            }
        }
        __syncthreads();
        for (int j = tid; j < HALF; j += PT) {
            const int slot = h * HALF + j;
            if (slot < count) {
                uint4 v = *reinterpret_cast<const uint4 *>(&stage[j * 4]);
                const uint32_t b = v.w >> 13;
                v.w = (uint32_t)(tbase + (v.w & 8191u));  // the original index
                reinterpret_cast<uint4 *>(out)[gbase[b] + (uint32_t)(slot - dstart[b])] = v;
            }
        }
        __syncthreads();
    }
}

// ---- 4: one bucket in final order, by one workgroup --------------------------------------------------------
// The bucket's records are read once, into registers (8 per thread at most).  Counting sort by the key's
// remainder with LDS atomics (arrival order inside a group = the records of half a cell), then every record
// counts the records of its group that came before it in the input -- which is ascending original index: the
// partition before was stable.  The records leave through an LDS staging area, 2048 at a time, as coalesced
// stores.  The workgroup also writes the first sorted position of each of its cells and adds its share of the
// occupancy statistics.
constexpr int ST = 1024;         // threads
constexpr int SREC = 8;          // records per thread at most
constexpr int SCAP = ST * SREC;  // records a bucket may hold
constexpr int SSTAGE = 2048;     // records staged at a time
constexpr int SUBKEYS = 2048;    // key remainders per bucket at most
constexpr int GROUP_MAX = 256;   // a group larger than this sends the build to the general path
__global__ __launch_bounds__(ST) void k_bucket_sort(const Rec32 *__restrict__ in, const uint32_t *__restrict__ bucket_start,
                                                    GridParams gp, KeyF32 kf, int halfbit, uint32_t width, int64_t ncell,
                                                    Rec32 *__restrict__ out, int32_t *__restrict__ perm,
                                                    uint32_t *__restrict__ cell_start,
                                                    unsigned long long *__restrict__ stats /*sum len^2, occupied, overflows*/) {
    __shared__ __attribute__((aligned(16))) uint32_t stage[SSTAGE * 4];  // 32 KB; `order` lives here first
    __shared__ uint32_t start[SUBKEYS];
    __shared__ uint32_t part[ST / 64];
    __shared__ unsigned long long red[2][ST / 64];
    uint16_t *order = reinterpret_cast<uint16_t *>(stage);  // [SCAP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t bs = bucket_start[blockIdx.x], be = bucket_start[blockIdx.x + 1];
    const uint32_t cnt = be - bs;
    if (cnt > (uint32_t)SCAP) {
        if (tid == 0) atomicAdd(&stats[2], 1ull);
        return;
    }
    for (int s = tid; s < SUBKEYS; s += ST) start[s] = 0;
    uint4 rc[SREC];
    uint32_t sk[SREC];  // the record's key remainder, later its slot
#pragma unroll
    for (int j = 0; j < SREC; j++) {
        const uint32_t e = j * ST + tid;
        rc[j] = e < cnt ? reinterpret_cast<const uint4 *>(in)[bs + e] : make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < SREC; j++) {
        sk[j] = cell_key(__uint_as_float(rc[j].x), __uint_as_float(rc[j].y), __uint_as_float(rc[j].z), gp, kf, halfbit) -
                blockIdx.x * width;
        if ((uint32_t)(j * ST + tid) < cnt) atomicAdd(&start[sk[j]], 1u);
    }
    __syncthreads();
    // exclusive scan of the counts; cell table and occupancy statistics on the way (two remainders per thread:
    // one whole cell where the cells are halved)
    {
        constexpr int PER = SUBKEYS / ST;  // 2
        uint32_t c[PER], tot = 0;
#pragma unroll
        for (int j = 0; j < PER; j++) {
            c[j] = start[tid * PER + j];
            tot += c[j];
        }
        uint32_t x = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(x, o, 64);
            if (lane >= o) x += y;
        }
        if (lane == 63) part[wave] = x;
        __syncthreads();
        uint32_t off = x - tot;
        for (int w = 0; w < wave; w++) off += part[w];
        unsigned long long sq = 0, occ = 0;
        bool big = false;
#pragma unroll
        for (int j = 0; j < PER; j++) {
            const int s = tid * PER + j;
            start[s] = off;
            big = big || c[j] > (uint32_t)GROUP_MAX;
            if (!halfbit || !(j & 1)) {
                const int64_t cell = ((int64_t)blockIdx.x * width + s) >> halfbit;
                if (s < (int)width && cell <= ncell) cell_start[cell] = bs + off;
                const unsigned long long len = halfbit ? c[j] + c[j + 1 < PER ? j + 1 : j] : c[j];
                sq += len * len;
                occ += len ? 1 : 0;
            }
            off += c[j];
        }
        if (__any(big) && lane == 0) atomicAdd(&stats[2], 1ull);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            sq += __shfl_down(sq, o, 64);
            occ += __shfl_down(occ, o, 64);
        }
        if (lane == 0) {
            red[0][wave] = sq;
            red[1][wave] = occ;
        }
    }
    __syncthreads();
    if (tid == 0) {
        unsigned long long a = 0, b = 0;
        for (int w = 0; w < ST / 64; w++) {
            a += red[0][w];
            b += red[1][w];
        }
        atomicAdd(&stats[0], a);  // (integer sums: the same whatever the order of the workgroups)
        atomicAdd(&stats[1], b);
    }
    // every record draws a slot inside its group (start[] ends up holding the groups' ends)
#pragma unroll
    for (int j = 0; j < SREC; j++) {
        const uint32_t e = j * ST + tid;
        if (e < cnt) order[atomicAdd(&start[sk[j]], 1u)] = (uint16_t)e;
    }
    __syncthreads();
    // a record's final slot: its group's first slot + the number of records of the group that came before it in
    // the input (a few records per group; the arrival order inside a group is whatever the atomics made it)
#pragma unroll
    for (int j = 0; j < SREC; j++) {
        const uint32_t e = j * ST + tid;
        if (e < cnt) {
            const uint32_t g0 = sk[j] ? start[sk[j] - 1] : 0u, g1 = start[sk[j]];
            uint32_t rank = 0;
            uint16_t v[8];
#pragma unroll
            for (int a = 0; a < 8; a++) v[a] = order[min(g0 + a, g1 - 1)];  // eight reads in flight
#pragma unroll
            for (int a = 0; a < 8; a++) rank += (g0 + a < g1 && v[a] < e) ? 1u : 0u;
            for (uint32_t a = g0 + 8; a < g1; a++) rank += order[a] < e ? 1u : 0u;
            sk[j] = g0 + rank;
        }
    }
    __syncthreads();  // `order` is dead: its memory stages the records now
    for (uint32_t q0 = 0; q0 < cnt; q0 += SSTAGE) {
#pragma unroll
        for (int j = 0; j < SREC; j++)
            if ((uint32_t)(j * ST + tid) < cnt && sk[j] - q0 < (uint32_t)SSTAGE)
                *reinterpret_cast<uint4 *>(&stage[(sk[j] - q0) * 4]) = rc[j];
        __syncthreads();
#pragma unroll
        for (int t = 0; t < SSTAGE / ST; t++) {
            const uint32_t s = q0 + t * ST + tid;
            if (s < cnt) {
                const uint4 v = *reinterpret_cast<const uint4 *>(&stage[(t * ST + tid) * 4]);
                reinterpret_cast<uint4 *>(out)[bs + s] = v;
                perm[bs + s] = (int32_t)v.w;
            }
        }
        __syncthreads();
    }
}

}  // namespace

// Records, index map and cell table of an fp32 cloud on the dense grid `gp` (cell keys as k_cell_keys32 makes
// them), or a decision that this cloud is not for this route: *taken = 0 and nothing is launched when the
// grid needs more than MAXB buckets, and stats_dev[2] != 0 afterwards when a bucket or a group turned out
// larger than a workgroup handles (the outputs are then incomplete: the caller runs the general build).
// stats_dev: three words, zeroed here; [0] += sum over the cells of count^2, [1] += occupied cells.
int pc_build_bucketed(pc_ctx *ctx, const float *xyz, int64_t n, const GridParams &gp, int64_t ncell, int halfbit,
                      Rec32 *rec_out, int32_t *perm_out, uint32_t *cell_start_out, unsigned long long *stats_dev,
                      hipStream_t st, int *taken) {
    *taken = 0;
    if (n < PTILE || ncell <= 0 || ncell >= (1ll << 30)) return PC_OK;
    const uint64_t keys = (uint64_t)(ncell + 1) << halfbit;  // key range, with the cell table's end entry
    // bucket width for about 5400 records a bucket (a workgroup holds 8192: room for a cloud half again as dense
    // in places), a whole number of cells
    const double per_key = (double)n / (double)((uint64_t)ncell << halfbit);
    double w = floor(5400.0 / per_key);
    uint32_t width = (uint32_t)std::min<double>(SUBKEYS, std::max(2.0, w)) & ~1u;
    const uint64_t nb64 = (keys + width - 1) / width;
    if (nb64 > (uint64_t)MAXB) return PC_OK;
    const int nb = (int)nb64;
    const int tiles = (int)((n + PTILE - 1) / PTILE), ngroups = (tiles + CG - 1) / CG;
    Scratch mid(ctx, sizeof(Rec32) * (size_t)n), hist(ctx, sizeof(uint32_t) * (size_t)tiles * nb),
        gsum(ctx, sizeof(uint32_t) * (size_t)ngroups * nb), total(ctx, sizeof(uint32_t) * MAXB),
        bstart(ctx, sizeof(uint32_t) * (MAXB + 1));
    if (!mid.ok() || !hist.ok() || !gsum.ok() || !total.ok() || !bstart.ok()) return pc_fail(ctx, PC_NOMEM, "index build scratch");
    KeyF32 kf{};
    kf.ok = 1;
    for (int a = 0; a < 3; a++) {
        kf.o[a] = (float)gp.o[a];
        if ((double)kf.o[a] != gp.o[a]) kf.ok = 0;
    }
    kf.s = (float)gp.s;
    if (!(gp.s > 0) || !std::isfinite(kf.s)) kf.ok = 0;
    const BucketDiv bd{width, 1.0f / (float)width};
    const int wide = ((uintptr_t)xyz & 15) == 0 ? 1 : 0;
    PC_HIP(hipMemsetAsync(stats_dev, 0, 3 * sizeof(unsigned long long), st));
    hipLaunchKernelGGL(k_bucket_hist, dim3(tiles), dim3(PT), 0, st, xyz, n, gp, kf, halfbit, bd, nb, wide, hist.as<uint32_t>());
    hipLaunchKernelGGL(k_col_group_sums, dim3((nb + 255) / 256, ngroups), dim3(256), 0, st, hist.as<uint32_t>(), tiles, nb,
                       gsum.as<uint32_t>());
    hipLaunchKernelGGL(k_col_scan_groups, dim3((nb + 255) / 256), dim3(256), 0, st, gsum.as<uint32_t>(), ngroups, nb,
                       total.as<uint32_t>());
    hipLaunchKernelGGL(k_bucket_starts, dim3(1), dim3(1024), 0, st, total.as<uint32_t>(), nb, bstart.as<uint32_t>());
    hipLaunchKernelGGL(k_col_apply, dim3((nb + 255) / 256, ngroups), dim3(256), 0, st, hist.as<uint32_t>(), tiles, nb,
                       gsum.as<uint32_t>(), bstart.as<uint32_t>());
    hipLaunchKernelGGL(k_bucket_scatter, dim3(((tiles + 7) / 8) * 8), dim3(PT), 0, st, xyz, n, gp, kf, halfbit, bd, nb, wide,
                       hist.as<uint32_t>(), mid.as<Rec32>(), tiles);
    hipLaunchKernelGGL(k_bucket_sort, dim3(nb), dim3(ST), 0, st, mid.as<Rec32>(), bstart.as<uint32_t>(), gp, kf, halfbit, width,
                       ncell, rec_out, perm_out, cell_start_out, stats_dev);
    PC_HIP(hipGetLastError());
    *taken = 1;
    return PC_OK;
}
