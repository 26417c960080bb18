// Stable LSD radix sort of (u64 key, u32 value) pairs, multi-block exclusive scan and ordered
// stream compaction for gfx950.  All three are bandwidth-bound integer passes; no matrix work.
#include "pc_common.h"

namespace {

constexpr int SORT_THREADS = 256;
constexpr int SORT_ROUNDS = 16;
constexpr int SORT_TILE = SORT_THREADS * SORT_ROUNDS;  // keys per block and pass
constexpr int RADIX = 256;

// ---- pass 1: per-block digit histogram ----------------------------------------------------------
template <typename K>
__global__ __launch_bounds__(SORT_THREADS) void k_radix_hist(const K *__restrict__ keys, int64_t n, int shift,
                                                              uint32_t *__restrict__ block_hist, int nblocks) {
    __shared__ uint32_t hist[RADIX];
    hist[threadIdx.x] = 0;
    __syncthreads();
    int64_t base = (int64_t)blockIdx.x * SORT_TILE;
#pragma unroll 4
    for (int r = 0; r < SORT_ROUNDS; r++) {
        int64_t i = base + (int64_t)r * SORT_THREADS + threadIdx.x;
        if (i < n) atomicAdd(&hist[(keys[i] >> shift) & (RADIX - 1)], 1u);
    }
    __syncthreads();
    block_hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = hist[threadIdx.x];  // digit-major
}

// ---- pass 3: stable scatter ---------------------------------------------------------------------
// The block first orders its tile by digit inside LDS, then writes it out: items of one digit
// leave as one contiguous run (coalesced stores) instead of one 4-byte store per item.
// Stable order inside the tile = (wave, round, lane) = ascending input position: wave w owns the
// contiguous quarter [w * 1024, (w + 1) * 1024) of the tile and ranks it on its own, so the ranking
// loop needs no block-wide barrier.
template <typename K>
__global__ __launch_bounds__(SORT_THREADS) void k_radix_scatter(const K *__restrict__ keys,
                                                                 const uint32_t *__restrict__ vals, int64_t n,
                                                                 int shift, const uint32_t *__restrict__ block_off,
                                                                 int nblocks, K *__restrict__ keys_out,
                                                                 uint32_t *__restrict__ vals_out) {
    constexpr int WAVES = SORT_THREADS / 64;
    __shared__ K keys_s[SORT_TILE];
    __shared__ uint32_t vals_s[SORT_TILE];
    __shared__ uint32_t wcnt[WAVES][RADIX];  // per wave: running digit counts, then offset of the wave's run
    __shared__ uint32_t dstart[RADIX];       // first tile-local slot of each digit
    __shared__ uint32_t gbase[RADIX];        // global slot of the block's first item of each digit
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int w = 0; w < WAVES; w++) wcnt[w][tid] = 0;
    gbase[tid] = block_off[(size_t)tid * nblocks + blockIdx.x];
    const int64_t base = (int64_t)blockIdx.x * SORT_TILE;
    const int64_t wbase = base + (int64_t)wave * (SORT_TILE / WAVES);
    K key[SORT_ROUNDS];
    uint32_t val[SORT_ROUNDS];
#pragma unroll
    for (int r = 0; r < SORT_ROUNDS; r++) {  // all loads in flight before the first use
        const int64_t i = wbase + r * 64 + lane;
        key[r] = i < n ? keys[i] : (K)0;
        val[r] = i < n ? vals[i] : 0u;
    }
    __syncthreads();
    uint32_t rank[SORT_ROUNDS];  // rank among the wave's items of the same digit
#pragma unroll
    for (int r = 0; r < SORT_ROUNDS; r++) {
        const bool valid = wbase + r * 64 + lane < n;
        const uint32_t d = (uint32_t)(key[r] >> shift) & (RADIX - 1);
        unsigned long long m = __ballot(valid);  // lanes of this wave holding the same digit
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const unsigned long long bal = __ballot((d >> b) & 1);
            m &= ((d >> b) & 1) ? bal : ~bal;
        }
        const uint32_t before = __popcll(m & ((1ull << lane) - 1ull));
        const uint32_t seen = valid ? wcnt[wave][d] : 0u;
        rank[r] = seen + before;
        // (wave-private row: the read above and this write are ordered by the wave's lock-step)
        if (valid && before == 0) wcnt[wave][d] = seen + __popcll(m);
    }
    __syncthreads();
    // digit totals -> tile-local start of each digit; per-wave counts -> offsets inside the digit's run
    {
        uint32_t tot = 0;
#pragma unroll
        for (int w = 0; w < WAVES; w++) {
            const uint32_t c = wcnt[w][tid];
            wcnt[w][tid] = tot;
            tot += c;
        }
        // exclusive scan of tot over the 256 digits (one per thread)
        uint32_t x = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(x, o, 64);
            if (lane >= o) x += y;
        }
        __shared__ uint32_t wsum[WAVES];
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        uint32_t off = 0;
        for (int w = 0; w < wave; w++) off += wsum[w];
        dstart[tid] = off + x - tot;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SORT_ROUNDS; r++) {
        if (wbase + r * 64 + lane < n) {
            const uint32_t d = (uint32_t)(key[r] >> shift) & (RADIX - 1);
            const uint32_t slot = dstart[d] + wcnt[wave][d] + rank[r];
            keys_s[slot] = key[r];
            vals_s[slot] = val[r];
        }
    }
    __syncthreads();
    const int64_t left = n - base;
    const int count = left < SORT_TILE ? (int)left : SORT_TILE;
#pragma unroll 4
    for (int j = tid; j < count; j += SORT_THREADS) {
        const K k = keys_s[j];
        const uint32_t d = (uint32_t)(k >> shift) & (RADIX - 1);
        const uint32_t pos = gbase[d] + ((uint32_t)j - dstart[d]);
        keys_out[pos] = k;
        vals_out[pos] = vals_s[j];
    }
}

// ---- multi-block exclusive scan (u32) ----------------------------------------------------------
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 16;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

__device__ inline uint32_t block_exclusive_scan(uint32_t v, uint32_t *total, uint32_t *lds /*[4]*/) {
    // inclusive wave scan with shuffles
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    if (lane == 63) lds[wave] = x;
    __syncthreads();
    uint32_t woff = 0, tot = 0;
    for (int w = 0; w < SCAN_THREADS / 64; w++) {
        uint32_t c = lds[w];
        if (w < wave) woff += c;
        tot += c;
    }
    __syncthreads();
    if (total) *total = tot;
    return woff + x - v;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_block_sums(const uint32_t *__restrict__ data, int64_t n,
                                                                   uint32_t *__restrict__ sums) {
    __shared__ uint32_t lds[4];
    int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++)
        if (base + j < n) s += data[base + j];
    uint32_t tot;
    block_exclusive_scan(s, &tot, lds);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

// one block scans up to a few hundred thousand block sums in place
__global__ __launch_bounds__(1024) void k_scan_single(uint32_t *__restrict__ data, int64_t n) {
    __shared__ uint32_t part[16];
    int64_t per = (n + 1023) / 1024;
    int64_t b = (int64_t)threadIdx.x * per, e = b + per < n ? b + per : n;
    uint32_t s = 0;
    for (int64_t i = b; i < e; i++) s += data[i];
    // exclusive scan of the 1024 partial sums: inclusive wave scans, then the 16 wave totals
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t x = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    if (lane == 63) part[wave] = x;
    __syncthreads();
    uint32_t acc = x - s;
    for (int w = 0; w < wave; w++) acc += part[w];
    for (int64_t i = b; i < e; i++) {
        uint32_t t = data[i];
        data[i] = acc;
        acc += t;
    }
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_apply(uint32_t *__restrict__ data, int64_t n,
                                                              const uint32_t *__restrict__ sums) {
    __shared__ uint32_t lds[4];
    int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++) {
        v[j] = base + j < n ? data[base + j] : 0;
        s += v[j];
    }
    uint32_t off = block_exclusive_scan(s, nullptr, lds) + (sums ? sums[blockIdx.x] : 0u);
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++) {
        if (base + j < n) data[base + j] = off;
        off += v[j];
    }
}

// ---- ordered compaction ----------------------------------------------------------------------
constexpr int CMP_THREADS = 256;
constexpr int CMP_ITEMS = 8;
constexpr int CMP_TILE = CMP_THREADS * CMP_ITEMS;

__global__ __launch_bounds__(CMP_THREADS) void k_compact_count(const uint8_t *__restrict__ flags, int64_t n,
                                                                uint32_t *__restrict__ counts) {
    __shared__ uint32_t lds[4];
    int64_t base = (int64_t)blockIdx.x * CMP_TILE + (int64_t)threadIdx.x * CMP_ITEMS;
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < CMP_ITEMS; j++)
        if (base + j < n) s += flags[base + j] ? 1u : 0u;
    uint32_t tot;
    block_exclusive_scan(s, &tot, lds);
    if (threadIdx.x == 0) counts[blockIdx.x] = tot;
    // (the entry behind the block counts becomes the total in the scan that follows: it starts at zero)
    if (blockIdx.x == 0 && threadIdx.x == 0) counts[gridDim.x] = 0u;
}

__global__ __launch_bounds__(CMP_THREADS) void k_compact_write(const uint8_t *__restrict__ flags, int64_t n,
                                                                const uint32_t *__restrict__ offsets,
                                                                int32_t *__restrict__ out) {
    __shared__ uint32_t lds[4];
    int64_t base = (int64_t)blockIdx.x * CMP_TILE + (int64_t)threadIdx.x * CMP_ITEMS;
    uint32_t s = 0;
    uint8_t f[CMP_ITEMS];
#pragma unroll
    for (int j = 0; j < CMP_ITEMS; j++) {
        f[j] = base + j < n ? flags[base + j] : 0;
        s += f[j] ? 1u : 0u;
    }
    uint32_t off = block_exclusive_scan(s, nullptr, lds) + offsets[blockIdx.x];
#pragma unroll
    for (int j = 0; j < CMP_ITEMS; j++)
        if (f[j]) out[off++] = (int32_t)(base + j);
}

}  // namespace

int pc_exclusive_scan_u32(pc_ctx *ctx, uint32_t *data, int64_t n, hipStream_t st) {
    if (n <= 0) return PC_OK;
    int64_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
    if (nb == 1) {  // one block's worth: one kernel
        hipLaunchKernelGGL(k_scan_apply, dim3(1), dim3(SCAN_THREADS), 0, st, data, n, (const uint32_t *)nullptr);
        PC_HIP(hipGetLastError());
        return PC_OK;
    }
    Scratch sums(ctx, sizeof(uint32_t) * (size_t)nb);
    if (!sums.ok()) return pc_fail(ctx, PC_NOMEM, "scan scratch");
    hipLaunchKernelGGL(k_scan_block_sums, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, st, data, n,
                       sums.as<uint32_t>());
    hipLaunchKernelGGL(k_scan_single, dim3(1), dim3(1024), 0, st, sums.as<uint32_t>(), nb);
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, st, data, n, sums.as<uint32_t>());
    PC_HIP(hipGetLastError());
    return PC_OK;
}

template <typename K>
static int radix_sort_impl(pc_ctx *ctx, K *keys, uint32_t *vals, K *keys_tmp, uint32_t *vals_tmp, int64_t n,
                           int key_bits, hipStream_t st, bool *result_in_tmp) {
    *result_in_tmp = false;
    if (n <= 1) return PC_OK;
    if (key_bits < 1) key_bits = 1;
    int passes = (key_bits + 7) / 8;
    int nblocks = (int)((n + SORT_TILE - 1) / SORT_TILE);
    Scratch hist(ctx, sizeof(uint32_t) * (size_t)RADIX * (size_t)nblocks);
    if (!hist.ok()) return pc_fail(ctx, PC_NOMEM, "radix histogram");
    K *kin = keys, *kout = keys_tmp;
    uint32_t *vin = vals, *vout = vals_tmp;
    for (int p = 0; p < passes; p++) {
        int shift = 8 * p;
        hipLaunchKernelGGL(k_radix_hist<K>, dim3(nblocks), dim3(SORT_THREADS), 0, st, kin, n, shift,
                           hist.as<uint32_t>(), nblocks);
        PC_TRY(pc_exclusive_scan_u32(ctx, hist.as<uint32_t>(), (int64_t)RADIX * nblocks, st));
        hipLaunchKernelGGL(k_radix_scatter<K>, dim3(nblocks), dim3(SORT_THREADS), 0, st, kin, vin, n, shift,
                           hist.as<uint32_t>(), nblocks, kout, vout);
        K *tk = kin;
        kin = kout;
        kout = tk;
        uint32_t *tv = vin;
        vin = vout;
        vout = tv;
    }
    PC_HIP(hipGetLastError());
    *result_in_tmp = (kin != keys);
    return PC_OK;
}

int pc_radix_sort_pairs(pc_ctx *ctx, uint64_t *keys, uint32_t *vals, uint64_t *keys_tmp, uint32_t *vals_tmp,
                        int64_t n, int key_bits, hipStream_t st, bool *result_in_tmp) {
    return radix_sort_impl<uint64_t>(ctx, keys, vals, keys_tmp, vals_tmp, n, key_bits, st, result_in_tmp);
}

int pc_radix_sort_pairs32(pc_ctx *ctx, uint32_t *keys, uint32_t *vals, uint32_t *keys_tmp, uint32_t *vals_tmp,
                          int64_t n, int key_bits, hipStream_t st, bool *result_in_tmp) {
    return radix_sort_impl<uint32_t>(ctx, keys, vals, keys_tmp, vals_tmp, n, key_bits, st, result_in_tmp);
}

int pc_compact_flags(pc_ctx *ctx, const uint8_t *flags, int64_t n, int32_t *out_idx, int64_t *count,
                     hipStream_t st) {
    *count = 0;
    if (n <= 0) return PC_OK;
    int64_t nb = (n + CMP_TILE - 1) / CMP_TILE;
    Scratch counts(ctx, sizeof(uint32_t) * (size_t)(nb + 1));
    if (!counts.ok()) return pc_fail(ctx, PC_NOMEM, "compaction scratch");
    uint32_t *c = counts.as<uint32_t>();
    hipLaunchKernelGGL(k_compact_count, dim3((unsigned)nb), dim3(CMP_THREADS), 0, st, flags, n, c);  // (also zeroes c[nb])
    PC_TRY(pc_exclusive_scan_u32(ctx, c, nb + 1, st));  // c[nb] becomes the total
    if (out_idx) hipLaunchKernelGGL(k_compact_write, dim3((unsigned)nb), dim3(CMP_THREADS), 0, st, flags, n, c, out_idx);
    PC_HIP(hipGetLastError());
    uint32_t *total = reinterpret_cast<uint32_t *>(ctx->pinned);
    PC_HIP(hipMemcpyAsync(total, c + nb, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    PC_HIP(hipStreamSynchronize(st));
    *count = (int64_t)*total;
    return PC_OK;
}

// ---- pc_sor_select_static: the ordered select of the statistical filter without any allocation ---------
// Same kernels as pc_sor_select / pc_compact_flags; the flag bytes and the per-block counts live in a
// file-scope device array instead of pooled blocks.  Every GPU holds its own copy of the array (at its own
// address, looked up once per device in pc_static_scratch_init) until the process ends.  Calls on one GPU share
// it, whatever their context: they take turns under a per-device mutex held until the call's stream has drained.
namespace {

constexpr int64_t STATIC_MAX_N = PC_SOR_SELECT_STATIC_MAX_N;
constexpr size_t STATIC_COUNT_BLOCKS = (size_t)(STATIC_MAX_N / CMP_TILE) + 1;         // + the total
constexpr size_t STATIC_COUNT_BYTES = (STATIC_COUNT_BLOCKS * sizeof(uint32_t) + 255) & ~size_t(255);
constexpr size_t STATIC_BYTES = (size_t)STATIC_MAX_N + STATIC_COUNT_BYTES;  // flag bytes, then the counts
static_assert(STATIC_COUNT_BLOCKS <= (size_t)SCAN_TILE, "the block counts must fit one scan block");

static __device__ __align__(256) unsigned char scratch[STATIC_BYTES];

constexpr int STATIC_MAX_DEVICES = 64;
void *g_scratch_ptr[STATIC_MAX_DEVICES];  // device address of `scratch` on each GPU (NULL: not looked up)
std::mutex g_scratch_mu;
std::mutex g_scratch_busy[STATIC_MAX_DEVICES];  // held from a call's first write to the array until its stream has drained

__global__ __launch_bounds__(256) void k_threshold_flags(const double *__restrict__ avg, int64_t n, double thr,
                                                          uint8_t *__restrict__ flags) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const double v = avg[i];
        flags[i] = (v > 0 && v < thr) ? 1 : 0;
    }
}

}  // namespace

// Called by pc_create with `device` current: the address of this GPU's copy of the scratch array.
int pc_static_scratch_init(int device) {
    if (device < 0 || device >= STATIC_MAX_DEVICES) return PC_OK;  // (pc_sor_select_static reports it)
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (g_scratch_ptr[device]) return PC_OK;
    void *p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)) != hipSuccess || !p) return PC_HIP_ERROR;
    g_scratch_ptr[device] = p;
    return PC_OK;
}

extern "C" int pc_sor_select_static(pc_ctx *ctx, const double *avg_dev, int64_t n, double threshold,
                                    uint8_t *out_mask_dev, int32_t *out_kept_dev, int64_t *n_kept, void *stream) {
    PcLock lock(ctx, stream);
    if (!ctx || n < 0 || (n > 0 && !avg_dev) || !n_kept) return pc_fail(ctx, PC_BAD_ARG, "pc_sor_select_static");
    if (n > STATIC_MAX_N)
        return pc_fail(ctx, PC_BAD_ARG, "pc_sor_select_static: more than PC_SOR_SELECT_STATIC_MAX_N values");
    PC_HIP(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    *n_kept = 0;
    if (n == 0) return PC_OK;
    unsigned char *base = nullptr;
    if (ctx->device >= 0 && ctx->device < STATIC_MAX_DEVICES) {
        std::lock_guard<std::mutex> g(g_scratch_mu);
        base = reinterpret_cast<unsigned char *>(g_scratch_ptr[ctx->device]);
    }
    if (!base) return pc_fail(ctx, PC_HIP_ERROR, "pc_sor_select_static: no static scratch on this device");
    std::lock_guard<std::mutex> busy(g_scratch_busy[ctx->device]);  // (released after the synchronise below)
    uint8_t *flags = out_mask_dev ? out_mask_dev : base;
    uint32_t *c = reinterpret_cast<uint32_t *>(base + STATIC_MAX_N);
    const int64_t nb = (n + CMP_TILE - 1) / CMP_TILE;
    // the slot behind the block counts becomes the total and must start at zero (the array keeps the
    // previous call's contents)
    PC_HIP(hipMemsetAsync(c + nb, 0, sizeof(uint32_t), st));
    hipLaunchKernelGGL(k_threshold_flags, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, avg_dev, n, threshold,
                       flags);
    hipLaunchKernelGGL(k_compact_count, dim3((unsigned)nb), dim3(CMP_THREADS), 0, st, flags, n, c);
    hipLaunchKernelGGL(k_scan_apply, dim3(1), dim3(SCAN_THREADS), 0, st, c, nb + 1, (const uint32_t *)nullptr);
    if (out_kept_dev)
        hipLaunchKernelGGL(k_compact_write, dim3((unsigned)nb), dim3(CMP_THREADS), 0, st, flags, n, c, out_kept_dev);
    PC_HIP(hipGetLastError());
    uint32_t *total = reinterpret_cast<uint32_t *>(ctx->pinned);  // page-locked since pc_create
    PC_HIP(hipMemcpyAsync(total, c + nb, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    PC_HIP(hipStreamSynchronize(st));
    *n_kept = (int64_t)*total;
    return PC_OK;
}
