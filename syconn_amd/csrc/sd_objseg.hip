// First stage of probability map -> object segmentation on the device:
// /root/reference/syconn/extraction/object_extraction_steps.py:204-366 (_object_segmentation_thread).  Plain branches:
//     :316-317  tmp_data = np.array(tmp_data > threshold, dtype=np.uint8)
//     :354-356  mop_data = apply_morphological_operations(tmp_data.copy(), morph_ops, mop_kwargs=dict(structure=struct))
//               this_labels_data, max_label = scipy.ndimage.label(mop_data)
//     :357-358  this_labels_data, max_label = scipy.ndimage.label(tmp_data)
// with the morphology semantics of /root/reference/syconn/proc/image.py:357-438 (_multi_mop_findobjects) on a binary
// volume: every operation acts on the bounding box of the foreground; closing / dilation pad that box by `iterations`
// zeros per side (less than the reach of the 5x5x3 element), clip their dilations to the padded window and erode with
// "outside the window = background"; opening erodes and dilates inside the box.  oracle/objseg_ref.py spells this out and
// is pinned to the reference's own functions (tests/golden/g9_objseg.npz).
// Watershed branch (:319-352, what the default config selects for mi / sj / vc): see sd_object_segmentation_watershed below.
//
// Data layout: the binary volume lives BIT-PACKED along z (the fastest axis): one uint32 = 32 consecutive z voxels of a
// (x,y) row of the volume padded by P = largest `iterations` per side.  A morphology step is then a handful of word loads,
// funnel shifts and AND / OR per 32 voxels (an HBM/L2 stream of 1/8 byte per voxel) instead of 15 byte loads per voxel.
// Connected components: labels start as the linear index of the voxel's z-RUN start (read off the bit mask, no atomics), so
// only run heads and run/run adjacencies along y and x touch the union-find (atomicMin links, larger root under smaller: a
// component's root is its first voxel in raster order); roots are ranked by an exclusive scan -> ids 1..N in
// scipy.ndimage.label's order, bit-exact.
#include "../../include/syconn_dense.h"
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <cmath>
#include <cstdlib>

extern int sd_fail_msg(int code, const char* msg);

namespace {

constexpr int MAX_OFFS = 128;
struct Offs { int n; signed char dx[MAX_OFFS], dy[MAX_OFFS], dz[MAX_OFFS]; };

// volume extents, pad, padded extents (PZW = words per padded z-row)
struct Dom { int X, Y, Z, P; int PX, PY, PZ, PZW; };
__device__ __forceinline__ size_t widx(const Dom& d, int x, int y, int zw) { return ((size_t)x * d.PY + y) * d.PZW + zw; }
// linear index -> (fastest, middle, slowest) coordinate with 32-bit arithmetic (volumes here have < 2^31 voxels: labels are
// int32; three 64-bit divisions per voxel made these HBM-bound passes instruction-bound)
__device__ __forceinline__ void dec3(size_t i, int n0, int n1, int& c0, int& c1, int& c2) {
    const unsigned u = (unsigned)i, r = u / (unsigned)n0, q = r / (unsigned)n1;
    c0 = (int)(u - r * (unsigned)n0); c1 = (int)(r - q * (unsigned)n1); c2 = (int)q;
}

// bbox[0..2] = min (padded coords), bbox[3..5] = max + 1; empty foreground: min > max
__global__ __launch_bounds__(256) void k_bbox_init(int* bbox) {
    if (threadIdx.x < 3) bbox[threadIdx.x] = 0x7fffffff;
    else if (threadIdx.x < 6) bbox[threadIdx.x] = 0;
}

// one thread = one mask word: bit b = (prob[x][y][32 zw + b - P] >= cut), zero in the padding
__global__ __launch_bounds__(256) void k_threshold_bits(const uint8_t* prob, int cut, Dom d, uint32_t* A) {
    const size_t total = (size_t)d.PX * d.PY * d.PZW;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        int zw, y, x;
        dec3(i, d.PZW, d.PY, zw, y, x);
        y -= d.P; x -= d.P;
        uint32_t w = 0;
        if ((unsigned)x < (unsigned)d.X && (unsigned)y < (unsigned)d.Y) {
            const uint8_t* row = prob + ((size_t)x * d.Y + y) * d.Z;
            const int z0 = zw * 32 - d.P;
#pragma unroll 8
            for (int b = 0; b < 32; ++b) {
                const int z = z0 + b;
                if ((unsigned)z < (unsigned)d.Z && (int)row[z] >= cut) w |= 1u << b;
            }
        }
        A[i] = w;
    }
}

__global__ __launch_bounds__(256) void k_bbox_bits(const uint32_t* A, Dom d, int* bbox) {
    const size_t total = (size_t)d.PX * d.PY * d.PZW;
    int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const uint32_t w = A[i];
        if (w) {
            int zw, yy, xx;
            dec3(i, d.PZW, d.PY, zw, yy, xx);
            const int c[3] = {xx, yy, zw * 32 + __builtin_ctz(w)};
            const int czh = zw * 32 + 32 - __builtin_clz(w);
            lo[0] = min(lo[0], c[0]); lo[1] = min(lo[1], c[1]); lo[2] = min(lo[2], c[2]);
            hi[0] = max(hi[0], c[0] + 1); hi[1] = max(hi[1], c[1] + 1); hi[2] = max(hi[2], czh);
        }
    }
    __shared__ int red[4][6];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        for (int m = 32; m >= 1; m >>= 1) {
            lo[a] = min(lo[a], __shfl_xor(lo[a], m, 64));
            hi[a] = max(hi[a], __shfl_xor(hi[a], m, 64));
        }
        if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][a] = lo[a]; red[threadIdx.x >> 6][3 + a] = hi[a]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {          // one atomic per workgroup and bound (the six addresses are shared by the whole grid)
        const int a = threadIdx.x;
        int v = red[0][a];
        for (int w = 1; w < 4; ++w) v = a < 3 ? min(v, red[w][a]) : max(v, red[w][a]);
        if (a < 3) { if (v != 0x7fffffff) atomicMin(&bbox[a], v); }
        else if (v != 0) atomicMax(&bbox[a], v);
    }
}

// bits [lo, hi) of the padded z axis that fall into word zw
__device__ __forceinline__ uint32_t zmask(int zw, int lo, int hi) {
    const int a = max(lo - zw * 32, 0), b = min(hi - zw * 32, 32);
    if (b <= a) return 0u;
    const uint32_t upto_b = b >= 32 ? 0xffffffffu : ((1u << b) - 1u);
    return upto_b & ~((1u << a) - 1u);
}

// One erosion (dilate == 0) or dilation (dilate == 1) step inside the window W = bbox grown by `wpad` per side.  Outside W
// the result is 0; reads outside W (or outside the buffer) count as background.  crop != 0: additionally zero everything
// outside the bbox itself (the `res[n:-n, ...]` crop after a closing / dilation).  One thread = one word = 32 voxels.
__global__ __launch_bounds__(256) void k_morph_bits(const uint32_t* src, uint32_t* dst, Dom d, const int* bbox, int wpad,
                                                    int dilate, int crop, const Offs o) {
    const size_t total = (size_t)d.PX * d.PY * d.PZW;
    const int wl[3] = {bbox[0] - wpad, bbox[1] - wpad, bbox[2] - wpad};
    const int wh[3] = {bbox[3] + wpad, bbox[4] + wpad, bbox[5] + wpad};
    const int cl[3] = {crop ? bbox[0] : wl[0], crop ? bbox[1] : wl[1], crop ? bbox[2] : wl[2]};
    const int ch[3] = {crop ? bbox[3] : wh[0], crop ? bbox[4] : wh[1], crop ? bbox[5] : wh[2]};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        int zw, y, x;
        dec3(i, d.PZW, d.PY, zw, y, x);
        uint32_t r = 0;
        const uint32_t keep = (x >= cl[0] && x < ch[0] && y >= cl[1] && y < ch[1]) ? zmask(zw, cl[2], ch[2]) : 0u;
        if (keep) {
            r = dilate ? 0u : 0xffffffffu;
            for (int k = 0; k < o.n; ++k) {
                // erosion: result bit z = AND over S of in[v + s]; dilation: OR over S of in[v - s]
                const int sx = dilate ? -o.dx[k] : o.dx[k], sy = dilate ? -o.dy[k] : o.dy[k], sz = dilate ? -o.dz[k] : o.dz[k];
                const int ux = x + sx, uy = y + sy;
                uint32_t s = 0;
                if (ux >= wl[0] && ux < wh[0] && uy >= wl[1] && uy < wh[1] && (unsigned)ux < (unsigned)d.PX &&
                    (unsigned)uy < (unsigned)d.PY) {
                    const uint32_t* row = src + widx(d, ux, uy, 0);
                    // word of bits (32 zw + sz .. 32 zw + sz + 31), window-masked at the source
                    const uint32_t c0 = row[zw] & zmask(zw, wl[2], wh[2]);
                    if (sz == 0) s = c0;
                    else if (sz > 0) {
                        const uint32_t c1 = (zw + 1 < d.PZW) ? (row[zw + 1] & zmask(zw + 1, wl[2], wh[2])) : 0u;
                        s = (c0 >> sz) | (c1 << (32 - sz));
                    } else {
                        const uint32_t c1 = (zw > 0) ? (row[zw - 1] & zmask(zw - 1, wl[2], wh[2])) : 0u;
                        s = (c0 << (-sz)) | (c1 >> (32 + sz));
                    }
                }
                r = dilate ? (r | s) : (r & s);
            }
            r &= keep;
        }
        dst[i] = r;
    }
}

// ---- connected components ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int cc_find(const int* L, int a) {
    int p = __hip_atomic_load(&L[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (p != a) { a = p; p = __hip_atomic_load(&L[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    return a;
}
__device__ __forceinline__ void cc_union(int* L, int a, int b) {
    while (true) {
        a = cc_find(L, a);
        b = cc_find(L, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }        // link the larger root under the smaller one
        const int old = atomicMin(&L[a], b);
        if (old == a) return;
        a = old;                                             // somebody else re-linked a meanwhile: retry from there
    }
}

// foreground bit of volume voxel (x,y,z) (unpadded coordinates inside the volume)
__device__ __forceinline__ bool fg(const uint32_t* A, const Dom& d, int x, int y, int z) {
    const int pz = z + d.P;
    return (A[widx(d, x + d.P, y + d.P, pz >> 5)] >> (pz & 31)) & 1u;
}

// The labelling works on z-RUNS, not voxels: the union-find lives at the first voxel of every run (its HEAD, read off the mask
// words: a set bit whose lower neighbour is clear), all passes up to the last one are one thread per mask WORD and touch the 4-byte
// arrays only at heads (a 512^3 volume with 8 % foreground has 0.7 M runs but 10 M foreground voxels); only the final relabel is a
// pass over the voxels.
// padded z of the first voxel of the run that contains bit b of word zw of `row` (padding bits are 0; without padding the scan
// stops at word 0)
__device__ __forceinline__ int run_start_pz(const uint32_t* row, int zw, int b) {
    uint32_t zeros = ~row[zw] & (b ? ((1u << b) - 1u) : 0u);
    while (!zeros && zw > 0) { --zw; zeros = ~row[zw]; }
    return zeros ? (zw * 32 + 32 - __builtin_clz(zeros)) : 0;
}
// voxels of the run that starts at padded z `pz`
__device__ __forceinline__ int run_length(const uint32_t* row, int nwords, int pz) {
    int zw = pz >> 5, avail = 32 - (pz & 31), len = 0;
    uint32_t w = row[zw] >> (pz & 31);
    while (true) {
        const uint32_t inv = ~w;
        const int n = min(inv ? __builtin_ctz(inv) : 32, avail);
        len += n;
        if (n < avail || ++zw >= nwords) break;
        w = row[zw]; avail = 32;
    }
    return len;
}
// heads of the runs that start inside word zw (valid z range only)
__device__ __forceinline__ uint32_t run_heads(const uint32_t* row, int zw, int vlo, int vhi) {
    const uint32_t w = row[zw] & zmask(zw, vlo, vhi);
    const uint32_t below = zw > 0 ? ((row[zw - 1] & zmask(zw - 1, vlo, vhi)) >> 31) : 0u;
    return w & ~((w << 1) | below);
}
// L[head] = head for every run
__global__ __launch_bounds__(256) void k_cc_init_heads(const uint32_t* A, Dom d, int* L) {
    const size_t total = (size_t)d.X * d.Y * d.PZW;
    const int vlo = d.P, vhi = d.P + d.Z;
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (size_t)gridDim.x * 256) {
        int zw, y, x;
        dec3(t, d.PZW, d.Y, zw, y, x);
        if (!zmask(zw, vlo, vhi)) continue;
        uint32_t h = run_heads(A + widx(d, x + d.P, y + d.P, 0), zw, vlo, vhi);
        const int ibase = (int)(((size_t)x * d.Y + y) * d.Z) + zw * 32 - d.P;      // linear voxel index of bit 0 of this word
        while (h) { const int bit = __builtin_ctz(h); h &= h - 1; L[ibase + bit] = ibase + bit; }
    }
}
// optional byte mask of the labelled volume
__global__ __launch_bounds__(256) void k_mask_bytes(const uint32_t* A, Dom d, uint8_t* mask_out) {
    const size_t total = (size_t)d.X * d.Y * d.Z;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        int z, y, x;
        dec3(i, d.Z, d.Y, z, y, x);
        mask_out[i] = fg(A, d, x, y, z) ? 1 : 0;
    }
}
// unions across y and x, one thread per MASK WORD (32 voxels of a z-row): where this row and a neighbouring row are both
// foreground they touch along a z-interval; one union per interval suffices (each side of it lies inside one z-run), issued
// between the heads of the two runs at the interval's first voxel = the set bits of
// adj & ~(adj << 1 | carry from the word below)  -- a few bit operations per 32 voxels.
__global__ __launch_bounds__(256) void k_cc_merge_runs(const uint32_t* A, Dom d, int* L) {
    const size_t total = (size_t)d.X * d.Y * d.PZW;
    const int vlo = d.P, vhi = d.P + d.Z;
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (size_t)gridDim.x * 256) {
        int zw, y, x;
        dec3(t, d.PZW, d.Y, zw, y, x);
        const uint32_t vm = zmask(zw, vlo, vhi);
        if (!vm) continue;
        const uint32_t* row = A + widx(d, x + d.P, y + d.P, 0);
        const uint32_t w = row[zw] & vm;
        if (!w) continue;
        const uint32_t wlow = zw > 0 ? (row[zw - 1] & zmask(zw - 1, vlo, vhi)) : 0u;
        const int rbase = (int)(((size_t)x * d.Y + y) * d.Z) - d.P;                 // voxel index of padded z = 0 of this row
#pragma unroll
        for (int dir = 0; dir < 2; ++dir) {
            if (dir == 0 ? y == 0 : x == 0) continue;
            const uint32_t* nrow = dir == 0 ? A + widx(d, x + d.P, y - 1 + d.P, 0) : A + widx(d, x - 1 + d.P, y + d.P, 0);
            const uint32_t adj = w & nrow[zw];
            if (!adj) continue;
            const uint32_t carry = zw > 0 ? ((wlow & nrow[zw - 1]) >> 31) : 0u;
            uint32_t starts = adj & ~((adj << 1) | carry);
            const int noff = dir == 0 ? d.Z : d.Z * d.Y;
            while (starts) {
                const int bit = __builtin_ctz(starts);
                starts &= starts - 1;
                const int ha = max(run_start_pz(row, zw, bit), d.P), hb = max(run_start_pz(nrow, zw, bit), d.P);
                cc_union(L, rbase + ha, rbase - noff + hb);
            }
        }
    }
}
constexpr int SCAN_PER_THREAD = 8, SCAN_BLOCK = 256 * SCAN_PER_THREAD;      // mask words per thread / per workgroup of the ranking passes
// path compression of the run heads (L[head] = root) fused with the count of roots per block of SCAN_BLOCK mask words (a root is
// a head with L[head] == head: fixed once the merges are done, so it can be counted while other blocks still compress)
__global__ __launch_bounds__(256) void k_cc_compress_count(const uint32_t* A, Dom d, size_t nwords, int* L, int* blockcnt) {
    __shared__ int red[4];
    const int vlo = d.P, vhi = d.P + d.Z;
    const size_t base = (size_t)blockIdx.x * SCAN_BLOCK + threadIdx.x;
    int c = 0;
#pragma unroll
    for (int k = 0; k < SCAN_PER_THREAD; ++k) {
        const size_t t = base + (size_t)k * 256;           // consecutive threads = consecutive words
        if (t >= nwords) continue;
        int zw, y, x;
        dec3(t, d.PZW, d.Y, zw, y, x);
        if (!zmask(zw, vlo, vhi)) continue;
        uint32_t h = run_heads(A + widx(d, x + d.P, y + d.P, 0), zw, vlo, vhi);
        const int ibase = (int)(((size_t)x * d.Y + y) * d.Z) + zw * 32 - d.P;
        while (h) {
            const int i = ibase + __builtin_ctz(h);
            h &= h - 1;
            const int r = cc_find(L, i);
            L[i] = r;
            c += r == i ? 1 : 0;
        }
    }
    for (int m = 32; m >= 1; m >>= 1) c += __shfl_xor(c, m, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) blockcnt[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// exclusive scan of blockcnt[0..n) in place, total -> *max_label (one workgroup): each of the 16 waves scans a contiguous
// sixteenth of the array 64 entries at a time (coalesced loads, wave scan by shuffles, running carry), the wave totals are
// scanned in LDS and added in a second coalesced sweep (the round-per-1024-entries Hillis-Steele form took 0.11 ms for the 65 k
// blocks of a 512^3 volume: 64 rounds x 20 barriers)
__global__ __launch_bounds__(1024) void k_cc_scan_blocks(int* blockcnt, int n, int* max_label) {
    __shared__ int wtot[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int per = ((n + 15) / 16 + 63) / 64 * 64;            // entries per wave, a multiple of 64
    const int lo = min(n, wave * per), hi = min(n, lo + per);
    int carry = 0;
    for (int i0 = lo; i0 < hi; i0 += 64) {
        const int i = i0 + lane;
        const int v = i < hi ? blockcnt[i] : 0;
        int incl = v;
        for (int s = 1; s < 64; s <<= 1) { const int t = __shfl_up(incl, s, 64); if (lane >= s) incl += t; }
        if (i < hi) blockcnt[i] = carry + incl - v;
        carry += __shfl(incl, 63, 64);
    }
    if (lane == 0) wtot[wave] = carry;
    __syncthreads();
    int off = 0;
    for (int w = 0; w < wave; ++w) off += wtot[w];
    if (off)
        for (int i = lo + lane; i < hi; i += 64) blockcnt[i] += off;
    if (threadIdx.x == 1023) *max_label = off + carry;
}
// rank[root] = 1 + number of roots with a smaller raster index (words in raster order, bits ascending = voxels in raster order)
__global__ __launch_bounds__(256) void k_cc_rank(const uint32_t* A, Dom d, size_t nwords, const int* L, const int* blockcnt, int* rank) {
    __shared__ int wsum[4];
    const int vlo = d.P, vhi = d.P + d.Z;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t base = (size_t)blockIdx.x * SCAN_BLOCK + (size_t)threadIdx.x * SCAN_PER_THREAD;      // consecutive words per thread
    uint32_t roots[SCAN_PER_THREAD];
    int ib[SCAN_PER_THREAD];
    int c = 0;
#pragma unroll
    for (int k = 0; k < SCAN_PER_THREAD; ++k) {
        roots[k] = 0u; ib[k] = 0;
        if (base + k < nwords) {
            int zw, y, x;
            dec3(base + k, d.PZW, d.Y, zw, y, x);
            if (zmask(zw, vlo, vhi)) {
                uint32_t h = run_heads(A + widx(d, x + d.P, y + d.P, 0), zw, vlo, vhi);
                ib[k] = (int)(((size_t)x * d.Y + y) * d.Z) + zw * 32 - d.P;
                while (h) {
                    const int bit = __builtin_ctz(h);
                    h &= h - 1;
                    if (L[ib[k] + bit] == ib[k] + bit) roots[k] |= 1u << bit;
                }
            }
        }
        c += __builtin_popcount(roots[k]);
    }
    int incl = c;                                             // inclusive scan over the wave's lanes
    for (int s = 1; s < 64; s <<= 1) { const int t = __shfl_up(incl, s, 64); if (lane >= s) incl += t; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int off = blockcnt[blockIdx.x] + incl - c;
    for (int w = 0; w < wave; ++w) off += wsum[w];
#pragma unroll
    for (int k = 0; k < SCAN_PER_THREAD; ++k) {
        uint32_t r = roots[k];
        while (r) { const int bit = __builtin_ctz(r); r &= r - 1; rank[ib[k] + bit] = ++off; }
    }
}
// cnt[key] += 1 for every lane with key > 0, aggregated over runs of equal keys in consecutive lanes (voxels of one object sit
// next to each other: one atomic per run instead of one per voxel; all 64 lanes must call)
__device__ __forceinline__ void count_runs(int key, int* cnt) {
    const int lane = threadIdx.x & 63;
    const int prev = __shfl_up(key, 1, 64);
    const bool head = lane == 0 || key != prev;
    const unsigned long long heads = __ballot(head);
    if (head && key > 0) {
        const unsigned long long above = lane == 63 ? 0ull : (heads >> (lane + 1));
        const int len = above ? (__builtin_ctzll(above) + 1) : (64 - lane);
        atomicAdd(&cnt[key], len);
    }
}
// slot of key k (>= 1; 0 = empty) in a small open-addressing table in LDS, inserting it if absent; -1 if the probe sequence is full
template <int SLOTS>
__device__ __forceinline__ int lds_slot(int* skey, int k) {
    int slot = (int)(((unsigned)k * 2654435761u) >> 16) & (SLOTS - 1);
    for (int tries = 0; tries < 8; ++tries, slot = (slot + 1) & (SLOTS - 1)) {
        const int old = atomicCAS(&skey[slot], 0, k);
        if (old == 0 || old == k) return slot;
    }
    return -1;
}
// rank[head] = final label of every run (roots have theirs); cnt != nullptr: cnt[label] += voxels of the run -- aggregated per
// workgroup in a small LDS table first (the runs of one component sit next to each other: 0.7 M global atomics onto a few
// thousand addresses took 0.35 ms per 512^3)
__global__ __launch_bounds__(256) void k_cc_head_labels(const uint32_t* A, Dom d, const int* L, int* rank, int* cnt) {
    constexpr int SLOTS = 512;
    __shared__ int skey[SLOTS], sval[SLOTS];
    if (cnt) {
        for (int i = threadIdx.x; i < SLOTS; i += 256) { skey[i] = 0; sval[i] = 0; }
        __syncthreads();
    }
    const size_t total = (size_t)d.X * d.Y * d.PZW;
    const int vlo = d.P, vhi = d.P + d.Z;
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (size_t)gridDim.x * 256) {
        int zw, y, x;
        dec3(t, d.PZW, d.Y, zw, y, x);
        if (!zmask(zw, vlo, vhi)) continue;
        const uint32_t* row = A + widx(d, x + d.P, y + d.P, 0);
        uint32_t h = run_heads(row, zw, vlo, vhi);
        const int ibase = (int)(((size_t)x * d.Y + y) * d.Z) + zw * 32 - d.P;
        while (h) {
            const int bit = __builtin_ctz(h);
            h &= h - 1;
            const int i = ibase + bit, r = L[i];
            const int lab = rank[r];                          // a root's entry: written by k_cc_rank only
            if (r != i) rank[i] = lab;
            if (cnt) {
                const int len = run_length(row, d.PZW, zw * 32 + bit);
                int slot = (int)(((unsigned)lab * 2654435761u) >> 23), tries = 0;      // labels are >= 1: key 0 = empty
                for (; tries < 8; ++tries, slot = (slot + 1) & (SLOTS - 1)) {
                    const int k = atomicCAS(&skey[slot], 0, lab);
                    if (k == 0 || k == lab) { atomicAdd(&sval[slot], len); break; }
                }
                if (tries == 8) atomicAdd(&cnt[lab], len);
            }
        }
    }
    if (cnt) {
        __syncthreads();
        for (int i = threadIdx.x; i < SLOTS; i += 256)
            if (skey[i]) atomicAdd(&cnt[skey[i]], sval[i]);
    }
}
// the label volume: zero-filled by a memset, then every run writes its label over its voxels (one thread per mask word, the
// runs that START in it)
__global__ __launch_bounds__(256) void k_cc_fill_runs(const uint32_t* A, Dom d, int* L, const int* rank) {
    const size_t total = (size_t)d.X * d.Y * d.PZW;
    const int vlo = d.P, vhi = d.P + d.Z;
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (size_t)gridDim.x * 256) {
        int zw, y, x;
        dec3(t, d.PZW, d.Y, zw, y, x);
        if (!zmask(zw, vlo, vhi)) continue;
        const uint32_t* row = A + widx(d, x + d.P, y + d.P, 0);
        uint32_t h = run_heads(row, zw, vlo, vhi);
        const int ibase = (int)(((size_t)x * d.Y + y) * d.Z) + zw * 32 - d.P;
        while (h) {
            const int bit = __builtin_ctz(h);
            h &= h - 1;
            const int i = ibase + bit, lab = rank[i], len = run_length(row, d.PZW, zw * 32 + bit);
            for (int k = 0; k < len; ++k) L[i + k] = lab;
        }
    }
}


// ---- watershed branch (object_extraction_steps.py:319-352) ---------------------------------------------------------------
// generic single-workgroup exclusive scan of v[0..n) in place, total -> *total_out
__global__ __launch_bounds__(1024) void k_scan_excl(int* v, const int* n_ptr, int n_add, int* total_out) {
    __shared__ int part[1024];
    __shared__ int carry;
    const int n = *n_ptr + n_add;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + threadIdx.x;
        const int x = i < n ? v[i] : 0;
        part[threadIdx.x] = x;
        __syncthreads();
        for (int s = 1; s < 1024; s <<= 1) {
            const int t = threadIdx.x >= s ? part[threadIdx.x - s] : 0;
            __syncthreads();
            part[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < n) v[i] = carry + part[threadIdx.x] - x;
        __syncthreads();
        if (threadIdx.x == 1023) carry += part[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0 && total_out) *total_out = carry;
}
// p[0 .. *n_ptr + extra) = v: the per-id tables are sized for the worst case (half the voxels) but used up to the id count, which
// only the device knows
__global__ __launch_bounds__(256) void k_fill_ids(int* p, const int* n_ptr, int extra, int v) {
    const size_t n = (size_t)*n_ptr + extra;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = v;
}
__global__ __launch_bounds__(256) void k_fill_int(int* p, size_t n, int v) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = v;
}
// min_seed_vx filter, step 1: del[i] = 1 where the seed with id i (1..N) has fewer than min_size voxels (:325-329)
__global__ __launch_bounds__(256) void k_seed_flags(const int* cnt, const int* N, int min_size, int* del) {
    const int n = *N;
    for (int i = blockIdx.x * 256 + threadIdx.x; i <= n; i += gridDim.x * 256) del[i] = (i >= 1 && cnt[i] < min_size) ? 1 : 0;
}
// step 2: rd[] = exclusive scan of del[] (rank among the deleted ids); the ids 1..N are dense, so the rank of a kept id among
// the kept ones is (i - 1) - rd[i].  D[j] = j-th smallest deleted id, K[j] = j-th smallest kept id.
__global__ __launch_bounds__(256) void k_seed_lists(const int* cnt, const int* rd, const int* N, int min_size, int* D, int* K) {
    const int n = *N;
    for (int i = 1 + blockIdx.x * 256 + threadIdx.x; i <= n; i += gridDim.x * 256) {
        if (cnt[i] < min_size) D[rd[i]] = i;
        else K[(i - 1) - rd[i]] = i;
    }
}
// step 3 (:333-344): the j-th smallest deleted id is handed to the j-th LARGEST kept id as long as it is smaller than that id
// (the reference walks both sorted lists and stops at the first pair that fails): J = length of that prefix
__global__ __launch_bounds__(256) void k_seed_prefix(const int* D, const int* K, const int* N, const int* nd_ptr, int* J) {
    const int nd = *nd_ptr, m = *N - nd, lim = min(nd, m);
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicMin(J, lim);
    for (int j = blockIdx.x * 256 + threadIdx.x; j < lim; j += gridDim.x * 256)
        if (!(D[j] < K[m - 1 - j])) atomicMin(J, j);
}
// step 4: map[i] = 0 (deleted) | D[j] (the j-th largest kept id, j < J) | i
__global__ __launch_bounds__(256) void k_seed_map(const int* cnt, const int* rd, const int* D, const int* N, const int* nd_ptr,
                                                  const int* J, int min_size, int* map) {
    const int n = *N, m = n - *nd_ptr, jj = *J;
    for (int i = blockIdx.x * 256 + threadIdx.x; i <= n; i += gridDim.x * 256) {
        int v = i;
        if (i >= 1) {
            if (cnt[i] < min_size) v = 0;
            else { const int j = m - 1 - ((i - 1) - rd[i]); if (j < jj) v = D[j]; }
        }
        map[i] = v;
    }
}
__global__ __launch_bounds__(256) void k_apply_map(const uint32_t* S, Dom d, int* L, const int* map) {      // relabel_vol (block_processing_C.pyx:161-169)
    // one thread per word of S (the bits the seeds were labelled from): a z-run carries ONE id -- rewrite the runs whose id changes
    const size_t total = (size_t)d.X * d.Y * d.PZW;
    const int vlo = d.P, vhi = d.P + d.Z;
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (size_t)gridDim.x * 256) {
        int zw, y, x;
        dec3(t, d.PZW, d.Y, zw, y, x);
        if (!zmask(zw, vlo, vhi)) continue;
        const uint32_t* row = S + widx(d, x + d.P, y + d.P, 0);
        uint32_t h = run_heads(row, zw, vlo, vhi);
        const int ibase = (int)(((size_t)x * d.Y + y) * d.Z) + zw * 32 - d.P;
        while (h) {
            const int bit = __builtin_ctz(h);
            h &= h - 1;
            const int i = ibase + bit, l = L[i], tl = l > 0 ? map[l] : l;
            if (tl != l) {
                const int len = run_length(row, d.PZW, zw * 32 + bit);
                for (int k = 0; k < len; ++k) L[i + k] = tl;
            }
        }
    }
}
// ... and S follows: bits of deleted seeds are cleared (S == {id > 0} afterwards).  Every thread reads and writes its own word only:
// any voxel of a run tells whether the run was deleted.
__global__ __launch_bounds__(256) void k_seed_bits_sync(uint32_t* S, Dom d, const int* L) {
    const size_t total = (size_t)d.X * d.Y * d.PZW;
    const int vlo = d.P, vhi = d.P + d.Z;
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (size_t)gridDim.x * 256) {
        int zw, y, x;
        dec3(t, d.PZW, d.Y, zw, y, x);
        const uint32_t vm = zmask(zw, vlo, vhi);
        if (!vm) continue;
        uint32_t* const wp = S + widx(d, x + d.P, y + d.P, zw);
        const uint32_t w = *wp & vm;
        if (!w) continue;
        const int ibase = (int)(((size_t)x * d.Y + y) * d.Z) + zw * 32 - d.P;
        uint32_t first = w & ~(w << 1), keep = *wp;      // first bit of every stretch of ones inside this word
        while (first) {
            const int bit = __builtin_ctz(first);
            first &= first - 1;
            if (L[ibase + bit] == 0) {
                const uint32_t above = ~(w >> bit);                                   // zeros of the stretch's continuation
                const int n = above ? __builtin_ctz(above) : 32 - bit;
                keep &= ~((n >= 32 ? 0xffffffffu : ((1u << n) - 1u)) << bit);
            }
        }
        if (keep != *wp) *wp = keep;
    }
}

// Exact anisotropic Euclidean distance transform of the foreground to the nearest background voxel INSIDE the array (vigra
// distanceTransform(background=False, pixel_pitch): the array border is not background), as SQUARED distances in int32
// (pitches are the integer voxel sizes, :349-350).  Separable: per z-row the distance to the nearest zero of the row, then the
// lower envelope of parabolas along y and along x, searched outwards from the voxel itself -- a candidate at offset k cannot
// beat the current best once (pitch * k)^2 >= best, so the search stops after about distance / pitch steps.
constexpr int EDT_INF = 0x3f000000;
__global__ __launch_bounds__(256) void k_edt_z(const uint32_t* A, Dom d, int pz, int* g, const int* comp, const int* mn, const int* mx) {
    // g is zero (memset, like the second buffer of the axis passes); one thread per mask word, the z-runs that start in it: inside
    // a run [s, e) the nearest background voxel of the row is s - 1 or e -- if that lies inside the volume (the array border
    // is not background, and bits outside the volume's own z range, the morphology padding, do not count).  comp != nullptr: only
    // the runs of components with several markers (the others are never flooded; see k_edt_axis)
    const size_t nwords = (size_t)d.X * d.Y * d.PZW;
    const int vlo = d.P, vhi = d.P + d.Z;
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < nwords; t += (size_t)gridDim.x * 256) {
        int zw, y, x;
        dec3(t, d.PZW, d.Y, zw, y, x);
        if (!zmask(zw, vlo, vhi)) continue;
        const uint32_t* row = A + widx(d, x + d.P, y + d.P, 0);
        uint32_t h = run_heads(row, zw, vlo, vhi);
        const int ibase = (int)(((size_t)x * d.Y + y) * d.Z) + zw * 32 - d.P;
        while (h) {
            const int bit = __builtin_ctz(h);
            h &= h - 1;
            if (comp) { const int c = comp[ibase + bit]; if (!(mx[c] > mn[c])) continue; }
            const int s0 = zw * 32 + bit - d.P, len = run_length(row, d.PZW, zw * 32 + bit), e0 = s0 + len;      // run = z in [s0, e0)
            const bool below = s0 > 0, above = e0 < d.Z;
            for (int k = 0; k < len; ++k) {
                long dist = -1;
                if (below) dist = k + 1;
                if (above) dist = dist < 0 ? (long)(len - k) : min(dist, (long)(len - k));
                int best = EDT_INF;
                if (dist >= 0) { const long tt = (long)pz * dist; best = (int)min(tt * tt, (long)EDT_INF); }
                g[ibase + bit + k] = best;
            }
        }
    }
}
// one axis pass: out[i] = min over k of in[i + k * stride] + (pitch * k)^2 along an axis of extent n (position c), foreground
// voxels only (read off the mask bits: a wave of background voxels costs two word loads)
// comp != nullptr: only voxels of components with several markers.  The others may hold ANY value >= 0 in `in`: a voxel u of
// another component on v's line has a background voxel w between itself and v (else the two were connected), and w's candidate
// (pitch k_w)^2 beats (pitch k_u)^2 + in[u] whatever in[u] >= 0 is.
__global__ __launch_bounds__(256) void k_edt_axis(const uint32_t* A, const int* in, int* out, Dom d, int axis, int pitch, const int* comp,
                                                  const int* mn, const int* mx) {
    const size_t total = (size_t)d.X * d.Y * d.Z;
    const size_t stride = axis == 1 ? (size_t)d.Z : (size_t)d.Z * d.Y;
    const int n = axis == 1 ? d.Y : d.X;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        int z, y, x;
        dec3(i, d.Z, d.Y, z, y, x);
        if (!fg(A, d, x, y, z)) continue;
        if (comp) { const int cc = comp[i]; if (!(mx[cc] > mn[cc])) continue; }
        int best = in[i];
        const int c = axis == 1 ? y : x;
        // (a candidate at offset k cannot win once (pitch k)^2 >= best; four offsets per round with their loads in flight
        // together -- the chain of dependent loads was what bounded this pass; evaluating a few candidates too many is harmless)
        for (int k0 = 1; k0 < n; k0 += 4) {
            const long p0 = (long)pitch * k0;
            if (p0 * p0 >= (long)best || (c - k0 < 0 && c + k0 >= n)) break;
            int lo[4], hi[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = k0 + u;
                lo[u] = c - k >= 0 ? in[i - (size_t)k * stride] : EDT_INF;
                hi[u] = c + k < n ? in[i + (size_t)k * stride] : EDT_INF;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long pk = (long)pitch * (k0 + u), q = pk * pk;
                best = (int)min((long)best, min((long)lo[u], (long)hi[u]) + q);
            }
        }
        out[i] = best;
    }
}

// per mask component: smallest and largest marker id found inside it (mn = INT_MAX, mx = 0: none)
__global__ __launch_bounds__(256) void k_comp_markers(const uint32_t* S, Dom d, const int* comp, const int* mk, size_t total, int* mn, int* mx,
                                                      int* max_label) {
    // smallest / largest marker per component and the largest marker inside the mask (= largest label of the flood's result),
    // aggregated per workgroup in LDS: the runs of one component are neighbours, global atomics on its two words serialise
    constexpr int SLOTS = 512;
    __shared__ int skey[SLOTS], smin[SLOTS], smax[SLOTS], stop;
    for (int i = threadIdx.x; i < SLOTS; i += 256) { skey[i] = 0; smin[i] = 0x7fffffff; smax[i] = 0; }
    if (threadIdx.x == 0) stop = 0;
    __syncthreads();
    int top = 0;
    auto see = [&](int m, int c) {
        if (c <= 0) return;
        top = max(top, m);
        const int slot = lds_slot<SLOTS>(skey, c);
        if (slot >= 0) { atomicMin(&smin[slot], m); atomicMax(&smax[slot], m); }
        else { atomicMin(&mn[c], m); atomicMax(&mx[c], m); }
    };
    if (S) {      // S == {marker > 0}, one marker and one mask component per z-run of S: one thread per word, the runs that start in it
        const size_t nwords = (size_t)d.X * d.Y * d.PZW;
        const int vlo = d.P, vhi = d.P + d.Z;
        for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < nwords; t += (size_t)gridDim.x * 256) {
            int zw, y, x;
            dec3(t, d.PZW, d.Y, zw, y, x);
            if (!zmask(zw, vlo, vhi)) continue;
            uint32_t h = run_heads(S + widx(d, x + d.P, y + d.P, 0), zw, vlo, vhi);
            const int ibase = (int)(((size_t)x * d.Y + y) * d.Z) + zw * 32 - d.P;
            while (h) {
                const int i = ibase + __builtin_ctz(h);
                h &= h - 1;
                const int m = mk[i];
                if (m > 0) see(m, comp[i]);
            }
        }
    } else {
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
            const int m = mk[i];
            if (m > 0) see(m, comp[i]);
        }
    }
    if (top > 0) atomicMax(&stop, top);
    __syncthreads();
    for (int i = threadIdx.x; i < SLOTS; i += 256)
        if (skey[i]) { atomicMin(&mn[skey[i]], smin[i]); atomicMax(&mx[skey[i]], smax[i]); }
    if (threadIdx.x == 0 && stop > 0) atomicMax(max_label, stop);
}
// pool capacity of a component = its voxel count (counted by the labelling pass) if it holds several markers, else 0
__global__ __launch_bounds__(256) void k_comp_keep_multi(const int* NC, const int* mn, const int* mx, int* sz) {
    const int nc = *NC;
    for (int c = blockIdx.x * 256 + threadIdx.x; c <= nc; c += gridDim.x * 256)
        if (!(mx[c] > mn[c])) sz[c] = 0;
}
// out = the flood's start state: background 0; a component without markers 0; with ONE marker that marker everywhere (the
// flood cannot leave the mask component and nothing competes); with several markers the markers themselves, which are also
// appended to the component's heap array (unordered: k_ws_flood heapifies)
constexpr unsigned WS_KMAX = 0x7fffffffu;
__global__ __launch_bounds__(256) void k_ws_init_seq(const int* comp, const int* mk, const int* g, size_t total, const int* mn,
                                                 const int* mx, const int* off, int* hcnt, unsigned long long* hkey, int* hidx,
                                                 int* out) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = comp[i];
        int o = 0;
        if (c > 0 && mx[c] > 0) {
            if (mx[c] == mn[c]) o = mx[c];
            else {
                o = mk[i];
                if (o > 0) {
                    const int slot = off[c] + atomicAdd(&hcnt[c], 1);
                    hkey[slot] = (unsigned long long)(WS_KMAX - (unsigned)g[i]) << 32;      // age 0
                    hidx[slot] = (int)i;
                }
            }
        }
        out[i] = o;
    }
}
// Priority flood of one multi-marker component per workgroup: skimage.segmentation.watershed(-distance, markers, mask) restated
// (watershed_raveled, connectivity 1, no compactness): pop the element with the smallest (value, age); every unlabelled mask
// neighbour -- visited in the order -x, -y, -z, +z, +y, +x of the reference's (x,y,z) arrays -- takes the popped element's label
// AT PUSH TIME and enters the heap with the next age.  value = -distance: compared through the exact squared distance.  All
// markers enter with age 0; equal (value, age) -- possible only among marker voxels -- are ordered by raster index here
// (skimage leaves that order to its heap's internals).  Ages are counted per component: comparisons only ever happen between
// elements of one component, whose relative push order is the same as under a global counter.
__device__ __forceinline__ bool ws_less(unsigned long long ka, int ia, unsigned long long kb, int ib) { return ka < kb || (ka == kb && ia < ib); }
// One WORKGROUP (one wave, lane 0 works) per multi-marker component: the algorithm is a chain of dependent heap accesses, so what
// matters is their latency -- the first WS_LDS_CAP heap entries (the heap holds the flood's current FRONT, a few thousand voxels
// for organelle-sized components) live in LDS (~64 cycles per access instead of ~1000 for global memory), entries beyond that
// spill to the component's arena slice.  Components run in parallel (two workgroups per CU).
constexpr int WS_LDS_CAP = 4608;             // 4608 * 12 bytes = 54 KiB
__global__ __launch_bounds__(64) void k_ws_flood_seq(const int* comp, const int* g, Dom d, const int* NC, const int* mn, const int* mx,
                                                 const int* off, const int* hcnt, unsigned long long* hkey, int* hidx, int* out) {
    __shared__ unsigned long long lk[WS_LDS_CAP];
    __shared__ int li[WS_LDS_CAP];
    if (threadIdx.x != 0) return;
    const int nc = *NC;
    const int sY = d.Z, sX = d.Z * d.Y;
    for (int c = 1 + blockIdx.x; c <= nc; c += gridDim.x) {
        if (!(mx[c] > mn[c])) continue;
        unsigned long long* const GK = hkey + off[c];
        int* const GI = hidx + off[c];
        int n = hcnt[c];
        auto getk = [&](int i) -> unsigned long long { return i < WS_LDS_CAP ? lk[i] : GK[i]; };
        auto geti = [&](int i) -> int { return i < WS_LDS_CAP ? li[i] : GI[i]; };
        auto put = [&](int i, unsigned long long k, int ix) { if (i < WS_LDS_CAP) { lk[i] = k; li[i] = ix; } else { GK[i] = k; GI[i] = ix; } };
        for (int i = 0; i < n && i < WS_LDS_CAP; ++i) { lk[i] = GK[i]; li[i] = GI[i]; }      // markers (unordered) -> LDS part
        auto sift_down = [&](int i) {
            const unsigned long long k = getk(i); const int ix = geti(i);
            while (true) {
                int ch = 2 * i + 1;
                if (ch >= n) break;
                unsigned long long kc = getk(ch); int ic = geti(ch);
                if (ch + 1 < n) {
                    const unsigned long long k2 = getk(ch + 1); const int i2 = geti(ch + 1);
                    if (ws_less(k2, i2, kc, ic)) { ++ch; kc = k2; ic = i2; }
                }
                if (!ws_less(kc, ic, k, ix)) break;
                put(i, kc, ic); i = ch;
            }
            put(i, k, ix);
        };
        for (int i = n / 2 - 1; i >= 0; --i) sift_down(i);
        unsigned age = 0;
        while (n > 0) {
            const int idx = geti(0);
            const int lab = out[idx];
            --n;
            if (n > 0) { put(0, getk(n), geti(n)); sift_down(0); }
            int z, y, x;
            dec3((size_t)idx, d.Z, d.Y, z, y, x);
            const int nb[6] = {x > 0 ? idx - sX : -1, y > 0 ? idx - sY : -1, z > 0 ? idx - 1 : -1,
                               z + 1 < d.Z ? idx + 1 : -1, y + 1 < d.Y ? idx + sY : -1, x + 1 < d.X ? idx + sX : -1};
            // the 18 loads of the six neighbours are independent: issued together, one memory latency
            int cq[6], oq[6], gq[6];
#pragma unroll
            for (int e = 0; e < 6; ++e) {
                const int q = nb[e] < 0 ? idx : nb[e];
                cq[e] = comp[q]; oq[e] = out[q]; gq[e] = g[q];
            }
#pragma unroll
            for (int e = 0; e < 6; ++e) {
                const int q = nb[e];
                if (q < 0 || cq[e] != c || oq[e] != 0) continue;
                out[q] = lab;
                const unsigned long long k = ((unsigned long long)(WS_KMAX - (unsigned)gq[e]) << 32) | (unsigned long long)(++age);
                int i = n++;                              // sift up
                while (i > 0) {
                    const int pa = (i - 1) >> 1;
                    const unsigned long long kp = getk(pa); const int ip = geti(pa);
                    if (!ws_less(k, q, kp, ip)) break;
                    put(i, kp, ip); i = pa;
                }
                put(i, k, q);
            }
        }
    }
}
// ---- level-synchronous form of the same flood (the product path; k_ws_flood_seq above is kept as its cross-check) ----------------
// With level = squared distance (higher = popped first) the sequential pop sequence is: take the FIFO of the highest non-empty
// level W, generation by generation (a generation = the FIFO's content when its first element is popped; what a generation
// pushes at level W is the next one).  A popped element e_i ("block" i) labels its unlabelled neighbours; neighbours ABOVE W
// start a cascade that floods the whole connected set of unlabelled voxels above W it touches -- and labels that set's
// unlabelled rim -- before e_{i+1} is popped.  So, per generation, all blocks can run at once:
//   * an unlabelled voxel at or below W next to a generation element or next to a cascade region goes to the claimer with the
//     smallest block index (atomicMin on a per-voxel claim word); a cascade region belongs to the smallest block touching it
//     (min-propagation inside the region until nothing changes);
//   * what the generation pushes is ordered by block index; the order INSIDE a block is free: its elements carry one label and
//     stay contiguous in every FIFO, so by induction no comparison between two different labels ever depends on it.
// FIFO order is carried by a 32-bit stamp T: marker voxels T = raster index (all markers precede every pushed element, ties among
// them by raster index as in k_ws_flood_seq), pushed elements T = 2^31 + (elements popped before this generation) + block index.
// Only marker voxels with an unlabelled neighbour are queued (popping any other marker does nothing).
// tools/experiments/ws_levelsync_proto.py checks this formulation (with a shuffled in-block order) against the sequential
// restatement on tie-heavy and cascade-heavy landscapes; tests compare the kernel with oracle/objseg_ref.py and with k_ws_flood_seq.
//
// One workgroup per multi-marker mask component.  Queued elements wait in two unordered bags per component -- NEAR (level >=
// lo) and FAR (below), so that the per-level pass only touches the levels about to be popped --, a generation lives sorted by
// (T, voxel) in LDS (global memory beyond WSP_CAP elements).
constexpr int WSP_THREADS = 1024;      // 256: 13.8 / 30.7 ms, 512: 12.1 / 21.8, 1024: 11.5 / 20.3 ms (organelle-like / giant components, whole branch, 512^3)
constexpr int WSP_CAP = 8192;
constexpr unsigned WSP_FREE = 0xffffffffu;
constexpr int WS_OPEN = -1;            // label volume during the flood: voxel of a multi-marker component, not labelled yet
struct WsPool {
    int* bl[2]; unsigned* bt[2]; int* bv[2];      // bags: level, stamp, voxel (NEAR from the left end of a component's slice, FAR from the right)
    unsigned long long* ga[2];                    // generations beyond the LDS capacity
    int* cl; int* wl;                             // voxels claimed in this generation at or below W; cascade voxels (BFS queue)
    unsigned* claim;                              // per voxel: smallest claiming block of the generation that labels it
};
__device__ __forceinline__ int ld_agent(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned ld_agent(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// slot = (*counter)++ for the lanes with pred, one atomic per wave; every lane of the wave must call
__device__ __forceinline__ int wave_slot(int* counter, bool pred) {
    const unsigned long long m = __ballot(pred);
    if (!m) return -1;
    const int lane = threadIdx.x & 63, leader = __builtin_ctzll(m);
    int b = 0;
    if (lane == leader) b = atomicAdd(counter, __builtin_popcountll(m));
    b = __shfl(b, leader, 64);
    return pred ? b + __builtin_popcountll(m & ((1ull << lane) - 1ull)) : -1;
}
__device__ __forceinline__ int bcast(const int* p) { __syncthreads(); const int v = *p; __syncthreads(); return v; }
// ascending bitonic network for any n (positions >= n count as +infinity and never move); keys are unique.  A step is a set
// of disjoint pairs: thread t takes the pairs t, t + NT, ... (four at a time: the loads of a batch are independent)
template <int WSP_NT, class Get, class Put>
__device__ __forceinline__ void ws_sort(int n, Get get, Put put) {
    if (n < 2) return;
    unsigned np2 = 2;
    while (np2 < (unsigned)n) np2 <<= 1;
    const unsigned half = np2 >> 1;
    for (unsigned k = 2; k <= np2; k <<= 1) {
        bool flip = true;
        for (unsigned j = k >> 1; j > 0; j >>= 1) {
            const unsigned lj = 31u - (unsigned)__builtin_clz(j);
            for (unsigned t0 = threadIdx.x; t0 < half; t0 += 4 * WSP_NT) {
                unsigned ii[4], pp[4];
                unsigned long long a[4], b[4];
                bool ok[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const unsigned t = t0 + u * WSP_NT;
                    const unsigned i = ((t >> lj) << (lj + 1)) | (t & (j - 1));
                    const unsigned pr = flip ? (i ^ (k - 1)) : (i + j);
                    ii[u] = i; pp[u] = pr; ok[u] = t < half && pr < (unsigned)n;
                    if (ok[u]) { a[u] = get(i); b[u] = get(pr); }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (ok[u] && a[u] > b[u]) { put(ii[u], b[u]); put(pp[u], a[u]); }
            }
            flip = false;
            __syncthreads();
        }
    }
}
// start state as k_ws_init_seq; the marker voxels that can push anything enter the component's bag (left-aligned, buffer 0)
constexpr int WI_WORDS = 4;      // mask words per thread of k_ws_init (a workgroup owns 1024 consecutive words)
__global__ __launch_bounds__(256) void k_ws_init(const uint32_t* M, const uint32_t* S, const int* comp, const int* mk, const int* g, Dom d, size_t total,
                                                 const int* mn, const int* mx, const int* off, int* hcnt, WsPool P, int* out) {
    // `out` is zero (memset); one thread per mask word, the z-runs that start in it: a run lies in one component.  The queued
    // markers of a workgroup are collected in LDS and get their bag slots with ONE global atomic per component and workgroup
    // (one per marker serialised on the component's counter: 1.3 M atomics onto a few hundred addresses per 512^3).
    constexpr int SLOTS = 256, LIST = 3072;
    __shared__ int skey[SLOTS], scnt[SLOTS], sbase[SLOTS];
    __shared__ int lslot[LIST], lg[LIST], li[LIST];
    __shared__ int nlist;
    for (int i = threadIdx.x; i < SLOTS; i += 256) { skey[i] = 0; scnt[i] = 0; }
    if (threadIdx.x == 0) nlist = 0;
    __syncthreads();
    const int sY = d.Z, sX = d.Z * d.Y;
    const size_t nwords = (size_t)d.X * d.Y * d.PZW;
    const int vlo = d.P, vhi = d.P + d.Z;
#pragma unroll 1
    for (int kw = 0; kw < WI_WORDS; ++kw) {
        const size_t t = (size_t)blockIdx.x * (256 * WI_WORDS) + (size_t)kw * 256 + threadIdx.x;
        if (t >= nwords) continue;
        int zw, y, x;
        dec3(t, d.PZW, d.Y, zw, y, x);
        if (!zmask(zw, vlo, vhi)) continue;
        const uint32_t* row = M + widx(d, x + d.P, y + d.P, 0);
        uint32_t h = run_heads(row, zw, vlo, vhi);
        const int ibase = (int)(((size_t)x * d.Y + y) * d.Z) + zw * 32 - d.P;
        auto queue = [&](int c, int idx) {
            const int slot = lds_slot<SLOTS>(skey, c);
            const int pos = slot >= 0 ? atomicAdd(&nlist, 1) : LIST;
            if (pos < LIST) { atomicAdd(&scnt[slot], 1); lslot[pos] = slot; lg[pos] = g[idx]; li[pos] = idx; }
            else {
                const int gs = off[c] + atomicAdd(&hcnt[c], 1);
                P.bl[0][gs] = g[idx]; P.bt[0][gs] = (unsigned)idx; P.bv[0][gs] = idx;
            }
        };
        while (h) {
            const int bit = __builtin_ctz(h);
            h &= h - 1;
            const int i0 = ibase + bit, c = comp[i0];
            if (c <= 0 || mx[c] <= 0) continue;
            const int len = run_length(row, d.PZW, zw * 32 + bit), z0 = zw * 32 + bit - d.P;
            if (mx[c] == mn[c]) {
                const int o = mx[c];
                for (int k = 0; k < len; ++k) out[i0 + k] = o;
                continue;
            }
            for (int k = 0; k < len; ++k) {      // several markers: the markers themselves, WS_OPEN for the voxels the flood will label
                const int o = mk[i0 + k];
                out[i0 + k] = o > 0 ? o : WS_OPEN;
                if (o <= 0) P.claim[i0 + k] = WSP_FREE;      // (only these voxels are ever claimed)
            }
            if (S) continue;                     // (the queue: word-wise below)
            for (int k = 0; k < len; ++k) {      // arbitrary marker volumes (sd_marker_flood): neighbour tests on the label arrays
                const int idx = i0 + k, z = z0 + k;
                if (mk[idx] <= 0) continue;
                const int nb[6] = {x > 0 ? idx - sX : -1, y > 0 ? idx - sY : -1, z > 0 ? idx - 1 : -1,
                                   z + 1 < d.Z ? idx + 1 : -1, y + 1 < d.Y ? idx + sY : -1, x + 1 < d.X ? idx + sX : -1};
                bool open = false;
#pragma unroll
                for (int e = 0; e < 6; ++e) open |= nb[e] >= 0 && comp[nb[e]] == c && mk[nb[e]] <= 0;
                if (open) queue(c, idx);
            }
        }
        if (S) {
            // queued = marker voxels (S == {marker > 0} exactly) with a neighbour inside the mask (hence inside their component)
            // that has no marker: U = M & ~S shifted along z and read from the four neighbouring rows -- bit operations per word
            const uint32_t vm = zmask(zw, vlo, vhi);
            const uint32_t* srow = S + widx(d, x + d.P, y + d.P, 0);
            const uint32_t sw = srow[zw] & vm;
            if (sw) {
                auto U = [&](const uint32_t* mr, const uint32_t* sr, int w) -> uint32_t {
                    return (w >= 0 && w < d.PZW) ? (mr[w] & ~sr[w] & zmask(w, vlo, vhi)) : 0u;
                };
                const uint32_t u0 = U(row, srow, zw);
                uint32_t nbr = (u0 << 1) | (U(row, srow, zw - 1) >> 31) | (u0 >> 1) | (U(row, srow, zw + 1) << 31);
                if (x > 0) nbr |= U(M + widx(d, x - 1 + d.P, y + d.P, 0), S + widx(d, x - 1 + d.P, y + d.P, 0), zw);
                if (x + 1 < d.X) nbr |= U(M + widx(d, x + 1 + d.P, y + d.P, 0), S + widx(d, x + 1 + d.P, y + d.P, 0), zw);
                if (y > 0) nbr |= U(M + widx(d, x + d.P, y - 1 + d.P, 0), S + widx(d, x + d.P, y - 1 + d.P, 0), zw);
                if (y + 1 < d.Y) nbr |= U(M + widx(d, x + d.P, y + 1 + d.P, 0), S + widx(d, x + d.P, y + 1 + d.P, 0), zw);
                uint32_t open = sw & nbr;
                while (open) {
                    const int b = __builtin_ctz(open);
                    open &= open - 1;
                    const int idx = ibase + b, c = comp[idx];
                    if (c > 0 && mx[c] > mn[c]) queue(c, idx);
                }
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SLOTS; i += 256)
        if (skey[i]) { sbase[i] = off[skey[i]] + atomicAdd(&hcnt[skey[i]], scnt[i]); scnt[i] = 0; }
    __syncthreads();
    const int n = min(nlist, LIST);
    for (int i = threadIdx.x; i < n; i += 256) {
        const int slot = lslot[i], gs = sbase[slot] + atomicAdd(&scnt[slot], 1);
        P.bl[0][gs] = lg[i]; P.bt[0][gs] = (unsigned)li[i]; P.bv[0][gs] = li[i];
    }
}
#ifdef SD_WS_TIMING
__device__ unsigned long long g_ws_dbg[2048][16];
#define WS_T(slot) do { if (threadIdx.x == 0) { const unsigned long long _n = wall_clock64(); g_ws_dbg[blockIdx.x][slot] += _n - t_last; t_last = _n; } } while (0)
#define WS_C(slot, v) do { if (threadIdx.x == 0) g_ws_dbg[blockIdx.x][slot] += (v); } while (0)
#else
#define WS_T(slot) do {} while (0)
#define WS_C(slot, v) do {} while (0)
#endif
template <int WSP_NT>
__global__ __launch_bounds__(WSP_NT) void k_ws_flood(const int* __restrict__ comp, const int* __restrict__ g, Dom d, const int* NC,
                                                     const int* mn, const int* mx, const int* off, const int* off_total,
                                                     const int* hcnt, WsPool P, int* out) {
    __shared__ unsigned long long sA[WSP_CAP];
    __shared__ int s_nA2, s_nCL, s_tail, s_dirty, s_near, s_far, s_nextW, s_max;
    const int tid = threadIdx.x, nc = *NC;
    const int sY = d.Z, sX = d.Z * d.Y;
#ifdef SD_WS_TIMING
    unsigned long long t_last = wall_clock64();
#endif
    for (int c = 1 + blockIdx.x; c <= nc; c += gridDim.x) {
        if (!(mx[c] > mn[c])) continue;
        const int base = off[c];
        const int cap = (c < nc ? off[c + 1] : *off_total) - base;
        int nNear = 0, nFar = hcnt[c], nearBuf = 0, farBuf = 0, lo = 0, delta = 0, W = 0, nA = 0;
        bool farLeft = true;
        unsigned tbase = 0x80000000u;
        __syncthreads();
        if (tid == 0) { s_near = 0; s_far = nFar; s_nextW = 0; }
        __syncthreads();
        auto nbrs = [&](int idx, int (&nb)[6]) {
            int z, y, x;
            dec3((size_t)idx, d.Z, d.Y, z, y, x);
            nb[0] = x > 0 ? idx - sX : -1; nb[1] = y > 0 ? idx - sY : -1; nb[2] = z > 0 ? idx - 1 : -1;
            nb[3] = z + 1 < d.Z ? idx + 1 : -1; nb[4] = y + 1 < d.Y ? idx + sY : -1; nb[5] = x + 1 < d.X ? idx + sX : -1;
        };
        // a generation is collected in global memory (any order) and then sorted: in LDS if it fits, in place otherwise
        unsigned long long* const GA = P.ga[0] + base;
        unsigned long long* const GB = P.ga[1] + base;      // an over-long generation is read from here while GA collects the next
        auto sortA = [&](int n) {
            if (n <= WSP_CAP) {
                for (int i = tid; i < n; i += WSP_NT) sA[i] = GA[i];
                __syncthreads();
                ws_sort<WSP_NT>(n, [&](unsigned i) { return sA[i]; }, [&](unsigned i, unsigned long long v) { sA[i] = v; });
            } else {
                ws_sort<WSP_NT>(n, [&](unsigned i) { return GA[i]; }, [&](unsigned i, unsigned long long v) { GA[i] = v; });
                for (int i = tid; i < n; i += WSP_NT) GB[i] = GA[i];
                __syncthreads();
            }
        };
        while (true) {
            if (nNear == 0) {
                if (nFar == 0) break;
                // ---- window: the highest FAR level and everything within delta of it become NEAR
                if (tid == 0) { s_max = 0; s_near = 0; s_far = 0; }
                __syncthreads();
                int m = 0;
                for (int k = tid; k < nFar; k += WSP_NT) m = max(m, P.bl[farBuf][farLeft ? base + k : base + cap - 1 - k]);
                for (int sft = 32; sft >= 1; sft >>= 1) m = max(m, __shfl_xor(m, sft, 64));
                if ((tid & 63) == 0) atomicMax(&s_max, m);
                const int wmax = bcast(&s_max);
                if (delta == 0) delta = max(1, wmax >> 6);
                lo = wmax - delta;
                const int db = farBuf ^ 1;
                for (int k0 = 0; k0 < nFar; k0 += WSP_NT) {
                    const int k = k0 + tid;
                    const bool valid = k < nFar;
                    const int src = farLeft ? base + k : base + cap - 1 - k;
                    const int l = valid ? P.bl[farBuf][src] : 0;
                    const unsigned t = valid ? P.bt[farBuf][src] : 0u;
                    const int v = valid ? P.bv[farBuf][src] : 0;
                    const bool near = valid && l >= lo;
                    int sl = wave_slot(&s_near, near);
                    if (near) { P.bl[db][base + sl] = l; P.bt[db][base + sl] = t; P.bv[db][base + sl] = v; }
                    sl = wave_slot(&s_far, valid && !near);
                    if (valid && !near) { const int dst = base + cap - 1 - sl; P.bl[db][dst] = l; P.bt[db][dst] = t; P.bv[db][dst] = v; }
                }
                __syncthreads();
                nNear = s_near; nFar = s_far; nearBuf = farBuf = db; farLeft = false; W = wmax;
                __syncthreads();
                WS_T(0); WS_C(6, 1);
            }
            // ---- level W: its elements leave NEAR and form the first generation
            if (tid == 0) { s_nA2 = 0; s_near = 0; s_nextW = 0; }
            __syncthreads();
            {
                const int nb2 = nearBuf ^ 1;
                int m = 0;
                for (int k0 = 0; k0 < nNear; k0 += WSP_NT) {
                    const int k = k0 + tid;
                    const bool valid = k < nNear;
                    const int l = valid ? P.bl[nearBuf][base + k] : 0;
                    const unsigned t = valid ? P.bt[nearBuf][base + k] : 0u;
                    const int v = valid ? P.bv[nearBuf][base + k] : 0;
                    const bool isA = valid && l == W;
                    int sl = wave_slot(&s_nA2, isA);
                    if (isA) GA[sl] = ((unsigned long long)t << 32) | (unsigned)v;
                    const bool keep = valid && !isA;
                    sl = wave_slot(&s_near, keep);
                    if (keep) { P.bl[nb2][base + sl] = l; P.bt[nb2][base + sl] = t; P.bv[nb2][base + sl] = v; m = max(m, l); }
                }
                for (int sft = 32; sft >= 1; sft >>= 1) m = max(m, __shfl_xor(m, sft, 64));
                if ((tid & 63) == 0 && m > 0) atomicMax(&s_nextW, m);
                __syncthreads();
                nA = s_nA2; nearBuf = nb2;
                __syncthreads();
            }
            sortA(nA);
            WS_T(1); WS_C(7, 1);
            // ---- generations of level W
            while (nA > 0) {
                const bool ldsA = nA <= WSP_CAP;
                auto voxA = [&](unsigned i) -> int { return (int)(unsigned)(ldsA ? sA[i] : GB[i]); };
                if (tid == 0) { s_nA2 = 0; s_nCL = 0; s_tail = 0; s_dirty = 0; }
                __syncthreads();
                // claims of the generation's elements (WS_OPEN marks the voxels the flood may still label: no mask / component test)
                for (int i0 = 0; i0 < nA; i0 += WSP_NT) {
                    const int i = i0 + tid;
                    const bool valid = i < nA;
                    int nb[6];
                    nbrs(valid ? voxA(i) : 0, nb);
                    int oq[6];
#pragma unroll
                    for (int e = 0; e < 6; ++e) oq[e] = (valid && nb[e] >= 0) ? ld_agent(&out[nb[e]]) : 0;
                    unsigned oldv[6];
#ifdef SD_WS_TIMING
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); WS_T(12);
#endif
#pragma unroll
                    for (int e = 0; e < 6; ++e)        // all claims in flight before the first result is needed
                        oldv[e] = oq[e] == WS_OPEN ? atomicMin(&P.claim[nb[e]], (unsigned)i) : 0u;
#ifdef SD_WS_TIMING
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); WS_T(13);
#endif
#pragma unroll
                    for (int e = 0; e < 6; ++e) {
                        const bool first = oq[e] == WS_OPEN && oldv[e] == WSP_FREE;
                        const int sl = wave_slot(&s_nCL, first);
                        if (first) P.cl[base + sl] = nb[e];
                    }
                    WS_T(14);
                }
                // the claimed voxels above W are cascade seeds: they move to the queue wl
                const int nCL0 = bcast(&s_nCL);
                for (int k0 = 0; k0 < nCL0; k0 += WSP_NT) {
                    const int k = k0 + tid;
                    const bool valid = k < nCL0;
                    const int q = valid ? P.cl[base + k] : 0;
                    const bool up = valid && g[q] > W;
                    const int sl = wave_slot(&s_tail, up);
                    if (up) { P.wl[base + sl] = q; P.cl[base + k] = -1; }
                }
                int tail = bcast(&s_tail);
                WS_T(2); WS_C(8, 1); WS_C(9, nA);
                if (tail > 0) {
                    WS_C(10, 1);
                    // cascade regions: breadth-first growth, the owner (smallest block) travels along
                    int head = 0;
                    while (head < tail) {
                        for (int k0 = head; k0 < tail; k0 += WSP_NT) {
                            const int k = k0 + tid;
                            const bool valid = k < tail;
                            const int r = valid ? P.wl[base + k] : 0;
                            const unsigned o = valid ? ld_agent(&P.claim[r]) : 0u;
                            int nb[6];
                            nbrs(r, nb);
                            int oq[6], gq[6];
#pragma unroll
                            for (int e = 0; e < 6; ++e) {
                                const int q = (valid && nb[e] >= 0) ? nb[e] : 0;
                                oq[e] = ld_agent(&out[q]); gq[e] = g[q];
                            }
#pragma unroll
                            for (int e = 0; e < 6; ++e) {
                                const int q = nb[e];
                                const bool cand = valid && q >= 0 && oq[e] == WS_OPEN && gq[e] > W;
                                const unsigned old = cand ? atomicMin(&P.claim[q], o) : 0u;
                                const bool first = cand && old == WSP_FREE;
                                if (cand && !first && old > o) s_dirty = 1;
                                const int sl = wave_slot(&s_tail, first);
                                if (first) P.wl[base + sl] = q;
                            }
                        }
                        head = tail;
                        tail = bcast(&s_tail);
                    }
                    // two blocks met inside one region: relax until every voxel of it carries the smaller one
                    while (bcast(&s_dirty)) {
                        if (tid == 0) s_dirty = 0;
                        __syncthreads();
                        for (int k = tid; k < tail; k += WSP_NT) {
                            const int r = P.wl[base + k];
                            const unsigned o = ld_agent(&P.claim[r]);
                            int nb[6];
                            nbrs(r, nb);
#pragma unroll
                            for (int e = 0; e < 6; ++e) {
                                const int q = nb[e];
                                if (q >= 0 && ld_agent(&out[q]) == WS_OPEN && g[q] > W && atomicMin(&P.claim[q], o) > o) s_dirty = 1;
                            }
                        }
                    }
                    // the regions' unlabelled rim at or below W
                    for (int k0 = 0; k0 < tail; k0 += WSP_NT) {
                        const int k = k0 + tid;
                        const bool valid = k < tail;
                        const int r = valid ? P.wl[base + k] : 0;
                        const unsigned o = valid ? ld_agent(&P.claim[r]) : 0u;
                        int nb[6];
                        nbrs(r, nb);
#pragma unroll
                        for (int e = 0; e < 6; ++e) {
                            const int q = nb[e];
                            const bool cand = valid && q >= 0 && ld_agent(&out[q]) == WS_OPEN && g[q] <= W;
                            const unsigned old = cand ? atomicMin(&P.claim[q], o) : 0u;
                            const bool first = cand && old == WSP_FREE;
                            const int sl = wave_slot(&s_nCL, first);
                            if (first) P.cl[base + sl] = q;
                        }
                    }
                }
                const int nCL = bcast(&s_nCL);
                WS_T(3); WS_C(11, nCL);
                // labels; pushes: level W -> next generation, lower levels -> the bags
                for (int k = tid; k < tail; k += WSP_NT) {
                    const int r = P.wl[base + k];
                    out[r] = ld_agent(&out[voxA(ld_agent(&P.claim[r]))]);
                }
                for (int k0 = 0; k0 < nCL; k0 += WSP_NT) {
                    const int k = k0 + tid;
                    const int qq = k < nCL ? P.cl[base + k] : -1;
                    const bool valid = qq >= 0;
                    const int q = valid ? qq : 0;
                    const unsigned w = valid ? ld_agent(&P.claim[q]) : 0u;
                    const int lq = valid ? g[q] : 0;
                    if (valid) out[q] = ld_agent(&out[voxA(w)]);
                    const unsigned t = tbase + w;
                    const bool toA = valid && lq == W, toN = valid && lq < W && lq >= lo, toF = valid && lq < lo;
                    int sl = wave_slot(&s_nA2, toA);
                    if (toA) GA[sl] = ((unsigned long long)t << 32) | (unsigned)q;
                    sl = wave_slot(&s_near, toN);
                    if (toN) { P.bl[nearBuf][base + sl] = lq; P.bt[nearBuf][base + sl] = t; P.bv[nearBuf][base + sl] = q; atomicMax(&s_nextW, lq); }
                    sl = wave_slot(&s_far, toF);
                    if (toF) { const int dst = base + cap - 1 - sl; P.bl[farBuf][dst] = lq; P.bt[farBuf][dst] = t; P.bv[farBuf][dst] = q; }
                }
                tbase += (unsigned)nA;
                nA = bcast(&s_nA2);
                WS_T(4);
                sortA(nA);
                WS_T(5);
            }
            __syncthreads();
            nNear = s_near; nFar = s_far; W = s_nextW;
            __syncthreads();
        }
    }
}
// squared distances -> float32 distances (what vigra returns), optional output
__global__ __launch_bounds__(256) void k_sqrt_out(const int* g, size_t total, float* out) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) out[i] = sqrtf((float)g[i]);
}

inline int grid_for(size_t n, int cap = 8192) { size_t g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > (size_t)cap ? (size_t)cap : g)); }
inline size_t rup256(size_t v) { return (v + 255) & ~(size_t)255; }

struct WsLayout { size_t a, b, rank, blockcnt, bbox, total; };
Dom make_dom(int X, int Y, int Z, int P) {
    Dom d{X, Y, Z, P, X + 2 * P, Y + 2 * P, Z + 2 * P, 0};
    d.PZW = (d.PZ + 31) / 32;
    return d;
}
WsLayout ws_layout(int X, int Y, int Z, int P) {
    WsLayout w{};
    const Dom d = make_dom(X, Y, Z, P);
    const size_t pwords = (size_t)d.PX * d.PY * d.PZW, nvox = (size_t)X * Y * Z;
    // (run_cc scans blocks of mask WORDS, X * Y * PZW of them: more than voxels for thin volumes with a large pad)
    const size_t nblk = (std::max(nvox, (size_t)X * Y * d.PZW) + SCAN_BLOCK - 1) / SCAN_BLOCK;
    size_t cur = 0;
    w.a = cur; cur += rup256(pwords * 4);
    w.b = cur; cur += rup256(pwords * 4);
    w.rank = cur; cur += rup256(nvox * 4);
    w.blockcnt = cur; cur += rup256((nblk + 1) * 4);
    w.bbox = cur; cur += 256;
    w.total = cur;
    return w;
}

}  // namespace

extern "C" {

size_t sd_objseg_workspace_bytes(int X, int Y, int Z, int max_iterations) {
    if (X <= 0 || Y <= 0 || Z <= 0 || max_iterations < 0) return 0;
    return ws_layout(X, Y, Z, max_iterations).total;
}

}  // extern "C"

namespace {
// structuring element -> offset list; 0 on success
int make_offsets(const uint8_t* struct_host, int sx, int sy, int sz, Offs& o) {
    if (!struct_host || sx < 1 || sy < 1 || sz < 1 || !(sx & 1) || !(sy & 1) || !(sz & 1) || sx > 15 || sy > 15 || sz > 15)
        return sd_fail_msg(SD_ERR_INVALID, "sd_object_segmentation: structuring element must have odd extents <= 15");
    for (int x = 0; x < sx; ++x)
        for (int y = 0; y < sy; ++y)
            for (int z = 0; z < sz; ++z)
                if (struct_host[((size_t)x * sy + y) * sz + z]) {
                    if (o.n == MAX_OFFS) return sd_fail_msg(SD_ERR_INVALID, "sd_object_segmentation: structuring element too large");
                    o.dx[o.n] = (signed char)(x - sx / 2); o.dy[o.n] = (signed char)(y - sy / 2); o.dz[o.n] = (signed char)(z - sz / 2);
                    ++o.n;
                }
    if (o.n == 0) return sd_fail_msg(SD_ERR_INVALID, "sd_object_segmentation: empty structuring element");
    return SD_OK;
}
int check_ops(const int32_t* ops, const int32_t* iterations, int n_ops, bool allow_erosion, int& P) {
    for (int i = 0; i < n_ops; ++i) {
        if (ops[i] == SD_MOP_EROSION && !allow_erosion)
            return sd_fail_msg(SD_ERR_INVALID, "sd_object_segmentation: binary_erosion selects the reference's watershed branch "
                                               "(object_extraction_steps.py:319-352): use sd_object_segmentation_watershed");
        if (ops[i] != SD_MOP_OPENING && ops[i] != SD_MOP_CLOSING && ops[i] != SD_MOP_DILATION && ops[i] != SD_MOP_EROSION)
            return sd_fail_msg(SD_ERR_INVALID, "sd_object_segmentation: unknown morphological operation");
        if (iterations[i] < 1 || iterations[i] > 64) return sd_fail_msg(SD_ERR_INVALID, "sd_object_segmentation: iterations out of range");
        if (ops[i] == SD_MOP_CLOSING || ops[i] == SD_MOP_DILATION) P = std::max(P, iterations[i]);
    }
    return SD_OK;
}
// the reference's operation list on the bit-packed volume A (B = scratch); the result is in A afterwards.
// erosion / opening: erosions inside the bounding box (outside = background), opening then dilates clipped to the box;
// closing / dilation: window = box + `iterations`, dilations clipped to it, closing then erodes, result cropped to the box
void run_morph(hipStream_t s, uint32_t*& A, uint32_t*& B, const Dom& d, int* bbox, const int32_t* ops, const int32_t* iterations,
               int n_ops, const Offs& o) {
    const size_t pwords = (size_t)d.PX * d.PY * d.PZW;
    for (int i = 0; i < n_ops; ++i) {
        const int n = iterations[i];
        hipLaunchKernelGGL(k_bbox_init, dim3(1), dim3(256), 0, s, bbox);
        hipLaunchKernelGGL(k_bbox_bits, dim3(grid_for(pwords, 512)), dim3(256), 0, s, A, d, bbox);
        const bool shrink_first = ops[i] == SD_MOP_OPENING || ops[i] == SD_MOP_EROSION;
        const int wpad = shrink_first ? 0 : n;
        const int nfirst = n, nsecond = (ops[i] == SD_MOP_DILATION || ops[i] == SD_MOP_EROSION) ? 0 : n;
        const int first_dilate = shrink_first ? 0 : 1;
        for (int k = 0; k < nfirst + nsecond; ++k) {
            const int dil = k < nfirst ? first_dilate : 1 - first_dilate;
            const int crop = (k == nfirst + nsecond - 1) ? 1 : 0;
            hipLaunchKernelGGL(k_morph_bits, dim3(grid_for(pwords)), dim3(256), 0, s, A, B, d, bbox, wpad, dil, crop, o);
            std::swap(A, B);
        }
    }
}
// scipy.ndimage.label of the bit-packed volume A into L (ids 1..N in raster order of the first voxel), N -> *max_label_dev
void run_cc(hipStream_t s, const uint32_t* A, const Dom& d, int* L, int* rank, int* blockcnt, int* max_label_dev, uint8_t* mask_out,
            int* cnt = nullptr) {      // cnt: voxels per label, entries 0 .. N + 1 (zeroed here, once N is known)
    const size_t nvox = (size_t)d.X * d.Y * d.Z, nwords = (size_t)d.X * d.Y * d.PZW;
    hipLaunchKernelGGL(k_cc_init_heads, dim3(grid_for(nwords)), dim3(256), 0, s, A, d, L);
    if (mask_out) hipLaunchKernelGGL(k_mask_bytes, dim3(grid_for(nvox)), dim3(256), 0, s, A, d, mask_out);
    hipLaunchKernelGGL(k_cc_merge_runs, dim3(grid_for(nwords)), dim3(256), 0, s, A, d, L);
    const int nblk = (int)((nwords + SCAN_BLOCK - 1) / SCAN_BLOCK);
    hipLaunchKernelGGL(k_cc_compress_count, dim3(nblk), dim3(256), 0, s, A, d, nwords, L, blockcnt);
    hipLaunchKernelGGL(k_cc_scan_blocks, dim3(1), dim3(1024), 0, s, blockcnt, nblk, max_label_dev);
    hipLaunchKernelGGL(k_cc_rank, dim3(nblk), dim3(256), 0, s, A, d, nwords, L, blockcnt, rank);
    if (cnt) hipLaunchKernelGGL(k_fill_ids, dim3(1024), dim3(256), 0, s, cnt, max_label_dev, 2, 0);
    hipLaunchKernelGGL(k_cc_head_labels, dim3(grid_for(nwords)), dim3(256), 0, s, A, d, L, rank, cnt);
    (void)hipMemsetAsync(L, 0, nvox * sizeof(int), s);      // (after the head passes: they use L as the union-find array)
    hipLaunchKernelGGL(k_cc_fill_runs, dim3(grid_for(nwords)), dim3(256), 0, s, A, d, L, rank);
}
int cut_of(double threshold) {
    // (uint8 p > t) <=> p >= floor(t) + 1; threshold 0 means "already a 0/1 mask" (object_extraction_steps.py:316): cut 1
    const double c = std::floor(threshold) + 1.0;
    return threshold == 0.0 ? 1 : (c < 0.0 ? 0 : (c > 256.0 ? 256 : (int)c));
}

// workspace of the watershed branch: the plain layout + a copy of the mask bits, three more int32 volumes (markers, mask
// components, squared distances), per-id tables (a volume of n voxels has at most n/2 + 1 six-connected components) and the heap arena
struct WsLayout2 { WsLayout w; size_t mbits, mk, comp, g, tab[9], bag[6], ga[2], cl, wl, scal, total; size_t T; };
WsLayout2 ws_layout2(int X, int Y, int Z, int P) {
    WsLayout2 l{};
    l.w = ws_layout(X, Y, Z, P);
    const Dom d = make_dom(X, Y, Z, P);
    const size_t pwords = (size_t)d.PX * d.PY * d.PZW, nvox = (size_t)X * Y * Z;
    l.T = nvox / 2 + 1026;
    size_t cur = l.w.total;
    l.mbits = cur; cur += rup256(pwords * 4);
    l.mk = cur; cur += rup256(nvox * 4);
    l.comp = cur; cur += rup256(nvox * 4);
    l.g = cur; cur += rup256(nvox * 4);
    for (int i = 0; i < 9; ++i) { l.tab[i] = cur; cur += rup256(l.T * 4); }
    for (int i = 0; i < 6; ++i) { l.bag[i] = cur; cur += rup256(nvox * 4); }      // the flood's pool: 48 bytes per voxel of a
    for (int i = 0; i < 2; ++i) { l.ga[i] = cur; cur += rup256(nvox * 8); }       // multi-marker component, sized for the worst case
    l.cl = cur; cur += rup256(nvox * 4);
    l.wl = cur; cur += rup256(nvox * 4);
    l.scal = cur; cur += 256;
    l.total = cur;
    return l;
}

// pointers into the watershed workspace
struct WsBufs { int *rank, *blockcnt, *comp, *mn, *mx, *off, *hcnt, *scal; size_t T; WsPool pool; };
WsBufs ws_bufs(char* wb, const WsLayout2& l) {
    WsBufs b{};
    b.rank = reinterpret_cast<int*>(wb + l.w.rank);
    b.blockcnt = reinterpret_cast<int*>(wb + l.w.blockcnt);
    b.comp = reinterpret_cast<int*>(wb + l.comp);
    b.mn = reinterpret_cast<int*>(wb + l.tab[5]); b.mx = reinterpret_cast<int*>(wb + l.tab[6]);
    b.off = reinterpret_cast<int*>(wb + l.tab[7]); b.hcnt = reinterpret_cast<int*>(wb + l.tab[8]);
    b.scal = reinterpret_cast<int*>(wb + l.scal);      // [0] N seeds, [1] NC mask components, [2] deleted seeds, [3] J, [4] pool total
    b.T = l.T;
    for (int i = 0; i < 2; ++i) {
        b.pool.bl[i] = reinterpret_cast<int*>(wb + l.bag[3 * i]);
        b.pool.bt[i] = reinterpret_cast<unsigned*>(wb + l.bag[3 * i + 1]);
        b.pool.bv[i] = reinterpret_cast<int*>(wb + l.bag[3 * i + 2]);
        b.pool.ga[i] = reinterpret_cast<unsigned long long*>(wb + l.ga[i]);
    }
    b.pool.cl = reinterpret_cast<int*>(wb + l.cl);
    b.pool.wl = reinterpret_cast<int*>(wb + l.wl);
    b.pool.claim = reinterpret_cast<unsigned*>(b.rank);      // free once the distance transform and the labelling are done
    return b;
}
// skimage.segmentation.watershed(-distance, markers, mask) (:351) given the mask bits M, the marker volume mk and the squared
// distances g: mask components, then the flood of every component that holds several markers
// ... part 1: the mask components (ids in B.comp, voxel counts, marker ranges, pool slices) and the result's largest label
void flood_components(hipStream_t s, const uint32_t* M, const uint32_t* seed_bits, const Dom& d, const int* mk, const WsBufs& B,
                      int32_t* max_label_dev, uint8_t* mask_out_dev) {      // seed_bits: optional, == {mk > 0}
    const size_t nvox = (size_t)d.X * d.Y * d.Z;
    const int gt = grid_for(B.T);
    int *rank = B.rank, *blockcnt = B.blockcnt, *comp = B.comp, *mn = B.mn, *mx = B.mx, *off = B.off, *hcnt = B.hcnt, *scal = B.scal;
    run_cc(s, M, d, comp, rank, blockcnt, scal + 1, mask_out_dev, off);      // off[c] = voxels of component c (entries 0 .. NC + 1)
    hipLaunchKernelGGL(k_fill_ids, dim3(1024), dim3(256), 0, s, mn, scal + 1, 2, 0x7fffffff);
    hipLaunchKernelGGL(k_fill_ids, dim3(1024), dim3(256), 0, s, mx, scal + 1, 2, 0);
    hipLaunchKernelGGL(k_fill_ids, dim3(1024), dim3(256), 0, s, hcnt, scal + 1, 2, 0);
    hipLaunchKernelGGL(k_fill_int, dim3(1), dim3(256), 0, s, max_label_dev, (size_t)1, 0);
    hipLaunchKernelGGL(k_comp_markers, dim3(grid_for(seed_bits ? (size_t)d.X * d.Y * d.PZW : nvox)), dim3(256), 0, s, seed_bits, d, comp, mk, nvox, mn, mx, max_label_dev);
    hipLaunchKernelGGL(k_comp_keep_multi, dim3(gt), dim3(256), 0, s, scal + 1, mn, mx, off);
    hipLaunchKernelGGL(k_scan_excl, dim3(1), dim3(1024), 0, s, off, scal + 1, 1, scal + 4);
}
// ... part 2: start state + queued markers, then the flood of every multi-marker component (level-synchronous, one workgroup per
// component; SD_WS_SEQUENTIAL=1 selects the sequential restatement it is cross-checked with); g = squared distances
void flood_run(hipStream_t s, const uint32_t* M, const uint32_t* seed_bits, const Dom& d, const int* mk, const int* g, const WsBufs& B,
               int32_t* labels_dev) {
    const size_t nvox = (size_t)d.X * d.Y * d.Z;
    int *comp = B.comp, *mn = B.mn, *mx = B.mx, *off = B.off, *hcnt = B.hcnt, *scal = B.scal;
    const WsPool& pool = B.pool;
    const bool sequential = getenv("SD_WS_SEQUENTIAL") != nullptr;
    if (sequential) {
        hipLaunchKernelGGL(k_ws_init_seq, dim3(grid_for(nvox)), dim3(256), 0, s, comp, mk, g, nvox, mn, mx, off, hcnt, pool.ga[0], pool.cl, labels_dev);
        hipLaunchKernelGGL(k_ws_flood_seq, dim3(4096), dim3(64), 0, s, comp, g, d, scal + 1, mn, mx, off, hcnt, pool.ga[0], pool.cl, labels_dev);
    } else {
        (void)hipMemsetAsync(labels_dev, 0, nvox * sizeof(int), s);
        hipLaunchKernelGGL(k_ws_init, dim3((unsigned)(((size_t)d.X * d.Y * d.PZW + 256 * WI_WORDS - 1) / (256 * WI_WORDS))), dim3(256), 0, s, M, seed_bits, comp, mk, g, d, nvox, mn, mx, off, hcnt, pool, labels_dev);
        hipLaunchKernelGGL(k_ws_flood<WSP_THREADS>, dim3(2048), dim3(WSP_THREADS), 0, s, comp, g, d, scal + 1, mn, mx, off, scal + 4, hcnt, pool, labels_dev);
    }
}
}  // namespace

// ---- Gaussian pre-smoothing (sd_gaussian_threshold) ------------------------------------------------------------------------------
constexpr int GAUSS_MAX_R = 64;
struct GaussTaps { int r; double w[2 * GAUSS_MAX_R + 1]; };
// one axis of the separable filter: out[v] = float(sum_k w[k] * in[reflect(i + k - r)]), i = index of v along the axis
template <typename TI>
__global__ __launch_bounds__(256) void k_gauss_axis(const TI* __restrict__ in, float* __restrict__ out, size_t nvox, int n, long stride,
                                                    const GaussTaps t) {
    const int period = n > 1 ? 2 * (n - 1) : 1;
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < nvox; v += (size_t)gridDim.x * 256) {
        const int i = (int)((v / (size_t)stride) % (size_t)n);
        const TI* const line = in + (v - (size_t)i * stride);
        double acc = 0.0;
        for (int k = 0; k <= 2 * t.r; ++k) {
            int j = i + k - t.r;
            if (j < 0 || j >= n) {                     // mirror without repeating the edge sample, as often as the window needs
                j %= period;
                if (j < 0) j += period;
                if (j >= n) j = period - j;
            }
            acc += t.w[k] * (double)line[(size_t)j * stride];
        }
        out[v] = (float)acc;
    }
}
// mask = smoothed > threshold (threshold 0: any non-zero value); `sm` == nullptr: no axis was smoothed, the uint8 values themselves
__global__ __launch_bounds__(256) void k_gauss_threshold(const uint8_t* __restrict__ prob, const float* __restrict__ sm, size_t nvox,
                                                         float thr, uint8_t* __restrict__ mask, float* __restrict__ smoothed_out) {
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < nvox; v += (size_t)gridDim.x * 256) {
        const float x = sm ? sm[v] : (float)prob[v];
        mask[v] = x > thr ? 1 : 0;
        if (smoothed_out) smoothed_out[v] = x;
    }
}

extern "C" size_t sd_gauss_workspace_bytes(int X, int Y, int Z) {
    if (X <= 0 || Y <= 0 || Z <= 0) return 0;
    return 2 * (((size_t)X * Y * Z + 63) / 64) * 64 * sizeof(float);
}

extern "C" {

int sd_object_segmentation(const uint8_t* prob_dev, int X, int Y, int Z, double threshold, const int32_t* ops,
                           const int32_t* iterations, int n_ops, const uint8_t* struct_host, int sx, int sy, int sz,
                           int32_t* labels_dev, int32_t* max_label_dev, uint8_t* mask_out_dev, void* ws, size_t ws_bytes,
                           void* stream) {
    if (!prob_dev || !labels_dev || !max_label_dev || !ws || X <= 0 || Y <= 0 || Z <= 0 || n_ops < 0 || (n_ops && (!ops || !iterations)))
        return sd_fail_msg(SD_ERR_INVALID, "sd_object_segmentation: bad argument");
    if ((size_t)X * Y * Z >= (1ull << 31)) return sd_fail_msg(SD_ERR_INVALID, "sd_object_segmentation: volume must have < 2^31 voxels");
    if (threshold != threshold) return sd_fail_msg(SD_ERR_INVALID, "sd_object_segmentation: NaN threshold");
    int P = 0;
    int rc = check_ops(ops, iterations, n_ops, false, P);
    if (rc != SD_OK) return rc;
    Offs o{};
    if (n_ops && (rc = make_offsets(struct_host, sx, sy, sz, o)) != SD_OK) return rc;
    const WsLayout w = ws_layout(X, Y, Z, P);
    if (ws_bytes < w.total) return sd_fail_msg(SD_ERR_NOMEM, "sd_object_segmentation: workspace too small");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    char* const wb = reinterpret_cast<char*>(ws);
    uint32_t* A = reinterpret_cast<uint32_t*>(wb + w.a);
    uint32_t* B = reinterpret_cast<uint32_t*>(wb + w.b);
    const Dom d = make_dom(X, Y, Z, P);
    const size_t pwords = (size_t)d.PX * d.PY * d.PZW;
    hipLaunchKernelGGL(k_threshold_bits, dim3(grid_for(pwords)), dim3(256), 0, s, prob_dev, cut_of(threshold), d, A);
    run_morph(s, A, B, d, reinterpret_cast<int*>(wb + w.bbox), ops, iterations, n_ops, o);
    run_cc(s, A, d, labels_dev, reinterpret_cast<int*>(wb + w.rank), reinterpret_cast<int*>(wb + w.blockcnt), max_label_dev, mask_out_dev);
    return hipGetLastError() == hipSuccess ? SD_OK : sd_fail_msg(SD_ERR_HIP, "sd_object_segmentation: launch failed");
}

size_t sd_objseg_watershed_workspace_bytes(int X, int Y, int Z, int max_iterations) {
    if (X <= 0 || Y <= 0 || Z <= 0 || max_iterations < 0) return 0;
    return ws_layout2(X, Y, Z, max_iterations).total;
}

int sd_object_segmentation_watershed(const uint8_t* prob_dev, int X, int Y, int Z, double threshold, const int32_t* ops,
                                     const int32_t* iterations, int n_ops, const int32_t* seed_ops,
                                     const int32_t* seed_iterations, int n_seed_ops, const uint8_t* struct_host, int sx, int sy,
                                     int sz, int min_seed_vx, const int32_t* pixel_pitch_xyz, int32_t* labels_dev,
                                     int32_t* max_label_dev, int32_t* markers_out_dev, float* distance_out_dev,
                                     uint8_t* mask_out_dev, void* ws, size_t ws_bytes, void* stream) {
    if (!prob_dev || !labels_dev || !max_label_dev || !ws || X <= 0 || Y <= 0 || Z <= 0 || n_ops < 0 || n_seed_ops <= 0 ||
        (n_ops && (!ops || !iterations)) || !seed_ops || !seed_iterations || !pixel_pitch_xyz)
        return sd_fail_msg(SD_ERR_INVALID, "sd_object_segmentation_watershed: bad argument");
    if ((size_t)X * Y * Z >= (1ull << 31)) return sd_fail_msg(SD_ERR_INVALID, "sd_object_segmentation_watershed: volume must have < 2^31 voxels");
    if (threshold != threshold) return sd_fail_msg(SD_ERR_INVALID, "sd_object_segmentation_watershed: NaN threshold");
    if (seed_ops[0] != SD_MOP_EROSION) return sd_fail_msg(SD_ERR_INVALID, "sd_object_segmentation_watershed: the seed operations start with the first binary_erosion");
    for (int a = 0; a < 3; ++a)      // squared distances are int32: (pitch * extent)^2 must stay below EDT_INF / 3
        if (pixel_pitch_xyz[a] < 1 || (double)pixel_pitch_xyz[a] * (a == 0 ? X : a == 1 ? Y : Z) > 18000.0)
            return sd_fail_msg(SD_ERR_INVALID, "sd_object_segmentation_watershed: pixel pitch x extent out of range");
    int P = 0;
    int rc = check_ops(ops, iterations, n_ops, false, P);
    if (rc == SD_OK) rc = check_ops(seed_ops, seed_iterations, n_seed_ops, true, P);
    if (rc != SD_OK) return rc;
    Offs o{};
    if ((rc = make_offsets(struct_host, sx, sy, sz, o)) != SD_OK) return rc;
    const WsLayout2 l = ws_layout2(X, Y, Z, P);
    if (ws_bytes < l.total) return sd_fail_msg(SD_ERR_NOMEM, "sd_object_segmentation_watershed: workspace too small");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    char* const wb = reinterpret_cast<char*>(ws);
    uint32_t* A = reinterpret_cast<uint32_t*>(wb + l.w.a);
    uint32_t* B = reinterpret_cast<uint32_t*>(wb + l.w.b);
    uint32_t* M = reinterpret_cast<uint32_t*>(wb + l.mbits);
    const WsBufs B2 = ws_bufs(wb, l);
    int *rank = B2.rank, *blockcnt = B2.blockcnt, *scal = B2.scal;
    int* bbox = reinterpret_cast<int*>(wb + l.w.bbox);
    int* mk = reinterpret_cast<int*>(wb + l.mk);
    int* g = reinterpret_cast<int*>(wb + l.g);
    int* tab[5];
    for (int i = 0; i < 5; ++i) tab[i] = reinterpret_cast<int*>(wb + l.tab[i]);
    int *cnt = tab[0], *rd = tab[1], *D = tab[2], *K = tab[3], *map = tab[4];
    const Dom d = make_dom(X, Y, Z, P);
    const size_t pwords = (size_t)d.PX * d.PY * d.PZW, nvox = (size_t)X * Y * Z;
    const int gt = grid_for(l.T);

    // tmp_data: threshold + the operations before the first erosion (:316-322)
    hipLaunchKernelGGL(k_threshold_bits, dim3(grid_for(pwords)), dim3(256), 0, s, prob_dev, cut_of(threshold), d, A);
    run_morph(s, A, B, d, bbox, ops, iterations, n_ops, o);
    if (hipMemcpyAsync(M, A, pwords * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) return sd_fail_msg(SD_ERR_HIP, "sd_object_segmentation_watershed: copy failed");
    // markers: the erosions (and whatever follows them), scipy.ndimage.label (:323-327)
    run_morph(s, A, B, d, bbox, seed_ops, seed_iterations, n_seed_ops, o);
    run_cc(s, A, d, mk, rank, blockcnt, scal + 0, nullptr, min_seed_vx > 1 ? cnt : nullptr);      // cnt[id] = voxels of seed id
    if (min_seed_vx > 1) {      // :330-347: drop seeds smaller than min_seed_vx, fill the holes in the id space from the top
        hipLaunchKernelGGL(k_fill_int, dim3(1), dim3(256), 0, s, scal + 3, (size_t)1, 0x7fffffff);
        hipLaunchKernelGGL(k_seed_flags, dim3(gt), dim3(256), 0, s, cnt, scal + 0, min_seed_vx, rd);
        hipLaunchKernelGGL(k_scan_excl, dim3(1), dim3(1024), 0, s, rd, scal + 0, 1, scal + 2);
        hipLaunchKernelGGL(k_seed_lists, dim3(gt), dim3(256), 0, s, cnt, rd, scal + 0, min_seed_vx, D, K);
        hipLaunchKernelGGL(k_seed_prefix, dim3(gt), dim3(256), 0, s, D, K, scal + 0, scal + 2, scal + 3);
        hipLaunchKernelGGL(k_seed_map, dim3(gt), dim3(256), 0, s, cnt, rd, D, scal + 0, scal + 2, scal + 3, min_seed_vx, map);
        hipLaunchKernelGGL(k_apply_map, dim3(grid_for(pwords)), dim3(256), 0, s, A, d, mk, map);
        hipLaunchKernelGGL(k_seed_bits_sync, dim3(grid_for(pwords)), dim3(256), 0, s, A, d, mk);
    }
    if (markers_out_dev && hipMemcpyAsync(markers_out_dev, mk, nvox * 4, hipMemcpyDeviceToDevice, s) != hipSuccess)
        return sd_fail_msg(SD_ERR_HIP, "sd_object_segmentation_watershed: copy failed");
    // the connected components of tmp_data (the flood never leaves one), then its distance transform (:349-350) -- in the components
    // that hold several markers only, unless the caller wants the distances: nothing else is ever flooded
    flood_components(s, M, A, d, mk, B2, max_label_dev, mask_out_dev);
    const int* const ecomp = distance_out_dev ? nullptr : B2.comp;
    (void)hipMemsetAsync(g, 0, nvox * sizeof(int), s);          // the passes write foreground voxels only: both buffers are 0 elsewhere
    (void)hipMemsetAsync(rank, 0, nvox * sizeof(int), s);
    hipLaunchKernelGGL(k_edt_z, dim3(grid_for(pwords)), dim3(256), 0, s, M, d, (int)pixel_pitch_xyz[2], g, ecomp, B2.mn, B2.mx);
    hipLaunchKernelGGL(k_edt_axis, dim3(grid_for(nvox)), dim3(256), 0, s, M, g, rank, d, 1, (int)pixel_pitch_xyz[1], ecomp, B2.mn, B2.mx);
    hipLaunchKernelGGL(k_edt_axis, dim3(grid_for(nvox)), dim3(256), 0, s, M, rank, g, d, 2, (int)pixel_pitch_xyz[0], ecomp, B2.mn, B2.mx);
    if (distance_out_dev) hipLaunchKernelGGL(k_sqrt_out, dim3(grid_for(nvox)), dim3(256), 0, s, g, nvox, distance_out_dev);
    flood_run(s, M, A, d, mk, g, B2, labels_dev);
    return hipGetLastError() == hipSuccess ? SD_OK : sd_fail_msg(SD_ERR_HIP, "sd_object_segmentation_watershed: launch failed");
}

int sd_marker_flood(const int32_t* d2_dev, const int32_t* markers_dev, const uint8_t* mask_dev, int X, int Y, int Z,
                    int32_t* labels_dev, int32_t* max_label_dev, void* ws, size_t ws_bytes, void* stream) {
    if (!d2_dev || !markers_dev || !mask_dev || !labels_dev || !max_label_dev || !ws || X <= 0 || Y <= 0 || Z <= 0)
        return sd_fail_msg(SD_ERR_INVALID, "sd_marker_flood: bad argument");
    if ((size_t)X * Y * Z >= (1ull << 31)) return sd_fail_msg(SD_ERR_INVALID, "sd_marker_flood: volume must have < 2^31 voxels");
    const WsLayout2 l = ws_layout2(X, Y, Z, 0);
    if (ws_bytes < l.total) return sd_fail_msg(SD_ERR_NOMEM, "sd_marker_flood: workspace too small");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    char* const wb = reinterpret_cast<char*>(ws);
    const Dom d = make_dom(X, Y, Z, 0);
    uint32_t* M = reinterpret_cast<uint32_t*>(wb + l.mbits);
    hipLaunchKernelGGL(k_threshold_bits, dim3(grid_for((size_t)d.PX * d.PY * d.PZW)), dim3(256), 0, s, mask_dev, 1, d, M);
    const WsBufs B = ws_bufs(wb, l);
    flood_components(s, M, nullptr, d, markers_dev, B, max_label_dev, nullptr);
    flood_run(s, M, nullptr, d, markers_dev, d2_dev, B, labels_dev);
    return hipGetLastError() == hipSuccess ? SD_OK : sd_fail_msg(SD_ERR_HIP, "sd_marker_flood: launch failed");
}

// ---- Gaussian pre-smoothing of a probability map + threshold (object_extraction_steps.py:296-297, 316-317) -------------------
// vigra.gaussianSmoothing restated from its published algorithm (vigra is absent from the reference tree and this image: parity
// UNPINNED): separable, axes in memory order (x, then y, then z), per axis a window of radius int(3 sigma + 0.5) (at least 1)
// sampled from exp(-t^2 / (2 sigma^2)) and normalised to sum 1, reflective border without repeating the edge
// (BORDER_TREATMENT_REFLECT), sums in double (the promote type of float data and double taps), every pass stored as float32.
int sd_gaussian_threshold(const uint8_t* prob_dev, int X, int Y, int Z, const double* sigma_xyz, double threshold,
                          uint8_t* mask_dev, float* smoothed_dev, void* ws, size_t ws_bytes, void* stream) {
    if (!prob_dev || !sigma_xyz || !mask_dev || !ws || X <= 0 || Y <= 0 || Z <= 0)
        return sd_fail_msg(SD_ERR_INVALID, "sd_gaussian_threshold: bad argument");
    const size_t nvox = (size_t)X * Y * Z;
    if (nvox >= (1ull << 31)) return sd_fail_msg(SD_ERR_INVALID, "sd_gaussian_threshold: volume must have < 2^31 voxels");
    if (ws_bytes < sd_gauss_workspace_bytes(X, Y, Z)) return sd_fail_msg(SD_ERR_NOMEM, "sd_gaussian_threshold: workspace too small");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    float* buf[2] = {reinterpret_cast<float*>(ws), reinterpret_cast<float*>(ws) + ((nvox + 63) / 64) * 64};
    const int n[3] = {X, Y, Z};
    const long stride[3] = {(long)Y * Z, (long)Z, 1};
    const float* cur = nullptr;      // nullptr: still the uint8 input
    int nb = 0;
    for (int a = 0; a < 3; ++a) {
        const double sg = sigma_xyz[a];
        if (!(sg >= 0.0) || !std::isfinite(sg)) return sd_fail_msg(SD_ERR_INVALID, "sd_gaussian_threshold: sigma must be >= 0");
        if (sg == 0.0) continue;                                   // this axis is not smoothed
        GaussTaps t;
        t.r = std::max(1, (int)(3.0 * sg + 0.5));
        if (t.r > GAUSS_MAX_R) return sd_fail_msg(SD_ERR_INVALID, "sd_gaussian_threshold: sigma too large (window radius > 64)");
        double sum = 0.0;
        for (int k = -t.r; k <= t.r; ++k) sum += (t.w[k + t.r] = std::exp(-0.5 * (double)k * k / (sg * sg)));
        for (int k = 0; k <= 2 * t.r; ++k) t.w[k] /= sum;
        float* const dst = buf[nb];
        if (cur) hipLaunchKernelGGL(k_gauss_axis<float>, dim3(grid_for(nvox)), dim3(256), 0, s, cur, dst, nvox, n[a], stride[a], t);
        else hipLaunchKernelGGL(k_gauss_axis<uint8_t>, dim3(grid_for(nvox)), dim3(256), 0, s, prob_dev, dst, nvox, n[a], stride[a], t);
        cur = dst;
        nb ^= 1;
    }
    hipLaunchKernelGGL(k_gauss_threshold, dim3(grid_for(nvox)), dim3(256), 0, s, prob_dev, cur, nvox, (float)threshold, mask_dev, smoothed_dev);
    return hipGetLastError() == hipSuccess ? SD_OK : sd_fail_msg(SD_ERR_HIP, "sd_gaussian_threshold: launch failed");
}

#ifdef SD_WS_TIMING
int sd_debug_ws_timing(unsigned long long* host_out, int reset) {      // [2048][16]
    hipDeviceSynchronize();
    if (host_out) hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_ws_dbg), sizeof(unsigned long long) * 2048 * 16);
    if (reset) { static unsigned long long z[2048 * 16]; hipMemcpyToSymbol(HIP_SYMBOL(g_ws_dbg), z, sizeof(z)); }
    return 0;
}
#endif

}  // extern "C"
