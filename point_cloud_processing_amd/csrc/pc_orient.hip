// Consistent normal signs by propagation along the minimum spanning tree of the k-nearest-neighbour
// graph (the job of PointCloud.orient_normals_consistent_tangent_plane, app/Core/pointcloud_preprocess.py:115;
// Hoppe et al. 1992), entirely on the device.
//
// Graph: one undirected edge per neighbour slot (i, s) -> j = idx[i, s], weight 1 - |n_i . n_j| rounded to
// fp32; edges are totally ordered by the 64-bit key (weight bits << 32 | slot number), so the spanning
// tree is unique.  The tree is grown Boruvka-style: every round each piece picks its smallest edge
// to another piece (atomicMin on the key, proposed to the pieces of BOTH end points), pieces hook onto
// the piece at the other end (the lower id of a mutual pair stays a root) and all points are
// re-pointed to their new root.  Pieces at least halve every round: <= log2(n) rounds, one scalar
// read-back each.
//
// Signs travel with the union-find structure instead of a tree walk afterwards: link[i] holds
// (root << 1 | flip) where flip says whether n_i must be reversed relative to its root's normal so
// that every tree edge on the path joins agreeing normals.  Hooking piece c onto piece t through the
// edge (i, j) sets flip(c -> t) = flip(i) ^ flip(j) ^ [n_i . n_j < 0]; re-pointing composes flips by
// XOR.  A 32-bit word is read and written whole, so pointer jumping in place always sees a true
// (pointer, flip) pair.  In the end each connected piece is turned so that its highest point (lowest
// index among equals) looks towards +z.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "pc_common.h"

namespace {

constexpr uint32_t W_DEAD = 0xFFFFFFFFu;
constexpr unsigned long long KEY_NONE = ~0ull;

inline unsigned blocks_for(int64_t n, int per) { return (unsigned)std::max<int64_t>(1, (n + per - 1) / per); }

__device__ inline double dot_n(const double *__restrict__ nrm, int64_t i, int64_t j) {
    return (nrm[3 * i] * nrm[3 * j] + nrm[3 * i + 1] * nrm[3 * j + 1]) + nrm[3 * i + 2] * nrm[3 * j + 2];
}

// weight bits of every neighbour slot (W_DEAD: no edge), link[i] = own root
__global__ __launch_bounds__(256) void k_orient_init(const double *__restrict__ nrm, const int32_t *__restrict__ idx,
                                                      const int32_t *__restrict__ cnt, int64_t n, int k,
                                                      uint32_t *__restrict__ wbits, uint32_t *__restrict__ link) {
    const int64_t slot = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (slot >= n * k) return;
    const int64_t i = slot / k;
    const int s = (int)(slot - i * k);
    if (s == 0) link[i] = (uint32_t)i << 1;
    uint32_t w = W_DEAD;
    const int32_t j = idx[slot];
    if ((!cnt || s < cnt[i]) && j >= 0 && j < n && j != i) {
        const float f = (float)(1.0 - fabs(dot_n(nrm, i, j)));
        w = f > 0.f ? __float_as_uint(f) : 0u;  // (non-negative floats order like their bit patterns)
    }
    wbits[slot] = w;
}

// every live slot offers its edge to the pieces at both ends; slots inside one piece die for good
__global__ __launch_bounds__(256) void k_orient_propose(const int32_t *__restrict__ idx, int64_t nslot, int k,
                                                         uint32_t *__restrict__ wbits, const uint32_t *__restrict__ link,
                                                         unsigned long long *__restrict__ best) {
    const int64_t slot = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (slot >= nslot) return;
    const uint32_t w = wbits[slot];
    if (w == W_DEAD) return;
    const int64_t i = slot / k;
    const uint32_t ci = link[i] >> 1, cj = link[idx[slot]] >> 1;
    if (ci == cj) {
        wbits[slot] = W_DEAD;
        return;
    }
    const unsigned long long key = ((unsigned long long)w << 32) | (unsigned long long)slot;
    // (the plain read first: most offers lose against what is already there)
    if (key < best[ci]) atomicMin(&best[ci], key);
    if (key < best[cj]) atomicMin(&best[cj], key);
}

// roots with an outgoing edge hook onto the piece at its other end
__global__ __launch_bounds__(256) void k_orient_hook(const double *__restrict__ nrm, const int32_t *__restrict__ idx,
                                                      int64_t n, int k, const uint32_t *__restrict__ link,
                                                      const unsigned long long *__restrict__ best,
                                                      uint32_t *__restrict__ hook, uint32_t *__restrict__ nhook) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    bool hooked = false;
    if (c < n) {
        uint32_t h = (uint32_t)c << 1;  // stays as it is
        const uint32_t l = link[c];
        const unsigned long long key = best[c];
        if ((l >> 1) == (uint32_t)c && key != KEY_NONE) {
            const int64_t slot = (int64_t)(key & 0xFFFFFFFFull);
            const int64_t i = slot / k, j = idx[slot];
            const uint32_t li = link[i], lj = link[j];
            const uint32_t t = (li >> 1) == (uint32_t)c ? (lj >> 1) : (li >> 1);
            if (!(best[t] == key && (uint32_t)c < t)) {  // of a mutual pair the lower id stays the root
                const uint32_t flip = (li ^ lj ^ (dot_n(nrm, i, j) < 0.0 ? 1u : 0u)) & 1u;
                h = (t << 1) | flip;
                hooked = true;
            }
        }
        hook[c] = h;
    }
    const unsigned long long m = __ballot(hooked);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(nhook, (uint32_t)__popcll(m));
}

// one round of pointer jumping over the hooks (in place: every word read is a true (pointer, flip) pair)
__global__ __launch_bounds__(256) void k_orient_jump(uint32_t *__restrict__ hook, int64_t n) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= n) return;
    const uint32_t h = hook[c], p = h >> 1;
    if (p == (uint32_t)c) return;
    const uint32_t hp = hook[p];
    if ((hp >> 1) != p) hook[c] = (hp & ~1u) | ((h ^ hp) & 1u);
}

// every point to its new root (whatever is left of the hook chains is walked here)
__global__ __launch_bounds__(256) void k_orient_flatten(uint32_t *__restrict__ link, const uint32_t *__restrict__ hook,
                                                         int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t l = link[i];
    uint32_t r = l >> 1, flip = l & 1u;
    for (int64_t guard = 0; guard < n; guard++) {  // (a forest: the walk ends at a root)
        const uint32_t h = hook[r];
        if ((h >> 1) == r) break;
        flip ^= h & 1u;
        r = h >> 1;
    }
    link[i] = (r << 1) | flip;
}

__device__ inline unsigned long long ordered_bits(double z) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(z);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

template <typename T>
__global__ __launch_bounds__(256) void k_orient_top(const T *__restrict__ xyz, const uint32_t *__restrict__ link, int64_t n,
                                                     unsigned long long *__restrict__ ztop) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned long long zb = ordered_bits((double)xyz[3 * i + 2] + 0.0);  // (-0 counts as +0)
    const uint32_t r = link[i] >> 1;
    if (zb > ztop[r]) atomicMax(&ztop[r], zb);
}

template <typename T>
__global__ __launch_bounds__(256) void k_orient_seed(const T *__restrict__ xyz, const uint32_t *__restrict__ link, int64_t n,
                                                      const unsigned long long *__restrict__ ztop,
                                                      uint32_t *__restrict__ seed) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t r = link[i] >> 1;
    if (ordered_bits((double)xyz[3 * i + 2] + 0.0) == ztop[r] && (uint32_t)i < seed[r]) atomicMin(&seed[r], (uint32_t)i);
}

// per root: whether the whole piece turns round so that its seed looks towards +z (kept in the root's
// seed slot, bit 31; the seed's normal is read here and changed only by the next kernel)
__global__ __launch_bounds__(256) void k_orient_rootflip(const double *__restrict__ nrm, const uint32_t *__restrict__ link,
                                                          uint32_t *__restrict__ seed, int64_t n,
                                                          uint32_t *__restrict__ npieces) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    bool root = false;
    if (r < n && (link[r] >> 1) == (uint32_t)r) {
        root = true;
        const uint32_t s = seed[r];
        const uint32_t turn = (link[s] & 1u) ^ (nrm[3 * (int64_t)s + 2] < 0.0 ? 1u : 0u);
        seed[r] = turn;
    }
    const unsigned long long m = __ballot(root);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(npieces, (uint32_t)__popcll(m));
}

__global__ __launch_bounds__(256) void k_orient_apply(double *__restrict__ nrm, const uint32_t *__restrict__ link,
                                                       const uint32_t *__restrict__ turn, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t l = link[i];
    if ((l ^ turn[l >> 1]) & 1u) {
        nrm[3 * i] = -nrm[3 * i];
        nrm[3 * i + 1] = -nrm[3 * i + 1];
        nrm[3 * i + 2] = -nrm[3 * i + 2];
    }
}

}  // namespace

extern "C" int pc_orient_normals(pc_ctx *ctx, const void *xyz_dev, int dtype, double *normals_dev,
                                 const int32_t *knn_idx_dev, const int32_t *knn_cnt_dev, int64_t n, int k,
                                 int64_t *out_pieces, void *stream) {
    PcLock lock(ctx, stream);  // calls on one context take turns (its block pool and read-back buffer are shared)
    if (!ctx || n < 0 || k < 1 || (dtype != PC_F32 && dtype != PC_F64) ||
        (n > 0 && (!xyz_dev || !normals_dev || !knn_idx_dev)))
        return pc_fail(ctx, PC_BAD_ARG, "pc_orient_normals");
    // point ids share a word with the flip bit, slot numbers fill the low half of the edge key
    if (n >= (1ll << 31) || n * (int64_t)k >= (1ll << 32)) return pc_fail(ctx, PC_OVERFLOW, "pc_orient_normals: n * k >= 2^32");
    PC_HIP(hipSetDevice(ctx->device));
    if (out_pieces) *out_pieces = 0;
    if (n == 0) return PC_OK;
    hipStream_t st = (hipStream_t)stream;
    const int64_t nslot = n * k;
    Scratch wbits(ctx, sizeof(uint32_t) * (size_t)nslot), link(ctx, sizeof(uint32_t) * (size_t)n),
        hook(ctx, sizeof(uint32_t) * (size_t)n), best(ctx, sizeof(unsigned long long) * (size_t)n), count(ctx, 256);
    if (!wbits.ok() || !link.ok() || !hook.ok() || !best.ok() || !count.ok())
        return pc_fail(ctx, PC_NOMEM, "pc_orient_normals scratch");
    uint32_t *const cnt_dev = count.as<uint32_t>();
    uint32_t *const host = reinterpret_cast<uint32_t *>(ctx->pinned);
    const unsigned nb = blocks_for(n, 256), nbs = blocks_for(nslot, 256);
    hipLaunchKernelGGL(k_orient_init, dim3(nbs), dim3(256), 0, st, normals_dev, knn_idx_dev, knn_cnt_dev, n, k,
                       wbits.as<uint32_t>(), link.as<uint32_t>());
    PC_HIP(hipGetLastError());
    const bool debug = ctx->knobs.debug;
    for (int round = 0; round < 64; round++) {  // pieces at least halve per round: 31 would do
        PC_HIP(hipMemsetAsync(best.p, 0xFF, sizeof(unsigned long long) * (size_t)n, st));
        PC_HIP(hipMemsetAsync(cnt_dev, 0, sizeof(uint32_t), st));
        hipLaunchKernelGGL(k_orient_propose, dim3(nbs), dim3(256), 0, st, knn_idx_dev, nslot, k, wbits.as<uint32_t>(),
                           link.as<uint32_t>(), best.as<unsigned long long>());
        hipLaunchKernelGGL(k_orient_hook, dim3(nb), dim3(256), 0, st, normals_dev, knn_idx_dev, n, k, link.as<uint32_t>(),
                           best.as<unsigned long long>(), hook.as<uint32_t>(), cnt_dev);
        for (int j = 0; j < 5; j++) hipLaunchKernelGGL(k_orient_jump, dim3(nb), dim3(256), 0, st, hook.as<uint32_t>(), n);
        hipLaunchKernelGGL(k_orient_flatten, dim3(nb), dim3(256), 0, st, link.as<uint32_t>(), hook.as<uint32_t>(), n);
        PC_HIP(hipGetLastError());
        PC_HIP(hipMemcpyAsync(host, cnt_dev, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        PC_HIP(hipStreamSynchronize(st));
        if (debug) fprintf(stderr, "[pc_amd]   orient round %d: %u pieces hooked\n", round, host[0]);
        if (host[0] == 0) break;
    }
    // seeds: highest point of every piece, lowest index among equals (best is free again: z keys)
    unsigned long long *ztop = best.as<unsigned long long>();
    uint32_t *seed = hook.as<uint32_t>();
    PC_HIP(hipMemsetAsync(ztop, 0, sizeof(unsigned long long) * (size_t)n, st));
    PC_HIP(hipMemsetAsync(seed, 0xFF, sizeof(uint32_t) * (size_t)n, st));
    PC_HIP(hipMemsetAsync(cnt_dev, 0, sizeof(uint32_t), st));
    if (dtype == PC_F32) {
        hipLaunchKernelGGL(k_orient_top<float>, dim3(nb), dim3(256), 0, st, (const float *)xyz_dev, link.as<uint32_t>(), n, ztop);
        hipLaunchKernelGGL(k_orient_seed<float>, dim3(nb), dim3(256), 0, st, (const float *)xyz_dev, link.as<uint32_t>(), n, ztop, seed);
    } else {
        hipLaunchKernelGGL(k_orient_top<double>, dim3(nb), dim3(256), 0, st, (const double *)xyz_dev, link.as<uint32_t>(), n, ztop);
        hipLaunchKernelGGL(k_orient_seed<double>, dim3(nb), dim3(256), 0, st, (const double *)xyz_dev, link.as<uint32_t>(), n, ztop, seed);
    }
    hipLaunchKernelGGL(k_orient_rootflip, dim3(nb), dim3(256), 0, st, normals_dev, link.as<uint32_t>(), seed, n, cnt_dev);
    hipLaunchKernelGGL(k_orient_apply, dim3(nb), dim3(256), 0, st, normals_dev, link.as<uint32_t>(), seed, n);
    PC_HIP(hipGetLastError());
    PC_HIP(hipMemcpyAsync(host, cnt_dev, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    PC_HIP(hipStreamSynchronize(st));  // (also: the scratch blocks go back to the pool below)
    if (out_pieces) *out_pieces = (int64_t)host[0];
    return PC_OK;
}
