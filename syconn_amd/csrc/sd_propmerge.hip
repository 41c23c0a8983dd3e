// Dataset-wide merge of per-chunk label statistics on the device (the chunk driver around the natives):
// what /root/reference/syconn/proc/sd_proc.py does with Python dictionaries per chunk (_map_subcell_extract_props_thread :617-678:
// the "purely inside this chunk and below min_obj_vx" filter; merge_prop_dicts :1248-1273; merge_map_dicts :1300-1322) is done here
// on RECORD ARRAYS that never leave HBM until the dataset is finished:
//
//   per chunk   sd_chunkprops_append / sd_chunkpairs_append read the hash tables sd_segstats_scan filled, apply the filter from the
//               table's own bounding boxes (an id lies on one of the six faces of the chunk  <=>  its box touches that face), shift
//               coordinates by the chunk's origin and append records at a device-side cursor -- no host synchronisation per chunk;
//   per dataset sd_propmerge_objects / sd_propmerge_pairs: stable LSD radix sort by id (rocPRIM), head flags, scan, one thread per
//               segment: voxel counts add up, the representative coordinate is the LAST chunk's (dict.update order), the bounding
//               boxes stay one per chunk, listed in chunk order (the reference appends them to a list per id).
//
// Records of one chunk are appended in table order (arbitrary) but chunks are appended in stream order and an id occurs at most once
// per chunk, so a STABLE sort by id alone yields chunk order inside every segment.
#include "../../include/syconn_dense.h"
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <stdint.h>
#include <algorithm>

extern int sd_fail_msg(int code, const char* msg);      // sd_api.hip: sets sd_last_error()

namespace {

typedef unsigned long long u64;
typedef unsigned int u32;

// view of an object table as sd_segstats.hip lays it out: keys | first | size | bbmin[3][cap] | bbmax[3][cap]
struct TabView {
    const u64* keys; const u64* first; const u64* size; const int* bbmin; const int* bbmax; u64 cap;
};
inline TabView tab_view(const void* base, u64 cap) {
    TabView t;
    t.keys = reinterpret_cast<const u64*>(base);
    t.first = t.keys + cap;
    t.size = t.first + cap;
    t.bbmin = reinterpret_cast<const int*>(t.size + cap);
    t.bbmax = t.bbmin + 3 * cap;
    t.cap = cap;
    return t;
}

// "purely inside this chunk and smaller than the threshold" (sd_proc.py:640-650 / :657-670); min_vx <= 1 disables the filter
__device__ __forceinline__ bool dropped(const TabView& t, u64 slot, u64 min_vx, int X, int Y, int Z) {
    if (min_vx <= 1 || t.size[slot] >= min_vx) return false;
    const bool on_face = t.bbmin[slot] == 0 || t.bbmin[t.cap + slot] == 0 || t.bbmin[2 * t.cap + slot] == 0 ||
                         t.bbmax[slot] == X || t.bbmax[t.cap + slot] == Y || t.bbmax[2 * t.cap + slot] == Z;
    return !on_face;
}

__global__ __launch_bounds__(256) void k_chunkprops_append(TabView t, int X, int Y, int Z, int ox, int oy, int oz, u64 min_vx,
                                                           u64* ids, int* rc, int* bb, u64* sizes, u64 max_rec, u64* cursor) {
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < t.cap; i += (u64)gridDim.x * 256) {
        const u64 k = t.keys[i];
        if (k == 0ull || t.size[i] == 0ull) continue;          // size 0: the id was only inserted for the overlap table
        if (dropped(t, i, min_vx, X, Y, Z)) continue;
        const u64 o = atomicAdd(cursor, 1ull);
        if (o >= max_rec) continue;                             // the caller sees cursor > max_rec and repeats with more room
        ids[o] = k; sizes[o] = t.size[i];
        const u64 f = t.first[i];                               // raster index in the (X,Y,Z) chunk, z fastest
        const u64 fxy = f / (u64)Z;
        rc[3 * o + 0] = (int)(fxy / (u64)Y) + ox; rc[3 * o + 1] = (int)(fxy % (u64)Y) + oy; rc[3 * o + 2] = (int)(f % (u64)Z) + oz;
        const int off[3] = {ox, oy, oz};
#pragma unroll
        for (int a = 0; a < 3; ++a) { bb[6 * o + a] = t.bbmin[a * t.cap + i] + off[a]; bb[6 * o + 3 + a] = t.bbmax[a * t.cap + i] + off[a]; }
    }
}

__global__ __launch_bounds__(256) void k_chunkpairs_append(const u64* pkeys, const u64* pcnt, u64 pcap, TabView sub, const u64* cell_keys,
                                                           int X, int Y, int Z, u64 min_vx, u64* out_sub, u64* out_cell, u64* out_cnt,
                                                           u64 max_rec, u64* cursor) {
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < pcap; i += (u64)gridDim.x * 256) {
        const u64 k = pkeys[i];
        if (k == 0ull) continue;
        const u64 ss = (k - 1) >> 32, cs = (k - 1) & 0xffffffffull;
        if (dropped(sub, ss, min_vx, X, Y, Z)) continue;       // a dropped organelle object leaves the overlap table as well (:668-669)
        const u64 o = atomicAdd(cursor, 1ull);
        if (o >= max_rec) continue;
        out_sub[o] = sub.keys[ss]; out_cell[o] = cell_keys[cs]; out_cnt[o] = pcnt[i];
    }
}

__global__ __launch_bounds__(256) void k_iota(u32* p, u64 n) {
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) p[i] = (u32)i;
}
__global__ __launch_bounds__(256) void k_gather64(const u64* src, const u32* idx, u64* dst, u64 n) {
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) dst[i] = src[idx[i]];
}
// head[i] = 1 where a new key starts in the sorted order
__global__ __launch_bounds__(256) void k_heads(const u64* ka, const u64* kb /* may be nullptr */, u32* head, u64 n) {
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256)
        head[i] = (i == 0 || ka[i] != ka[i - 1] || (kb && kb[i] != kb[i - 1])) ? 1u : 0u;
}

// one thread per segment of equal ids: total voxel count, representative coordinate of the last chunk; every thread also moves its own
// record's box into sorted order
__global__ __launch_bounds__(256) void k_reduce_objects(const u64* skey, const u32* perm, const u32* head, const u32* seg, const u64* sizes,
                                                        const int* rc, const int* bb, u64 n, u64* uniq, u64* tot, int* last_rc,
                                                        u32* seg_begin, int* bb_sorted, u64* n_unique) {
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) {
        const u32 src = perm[i];
#pragma unroll
        for (int a = 0; a < 6; ++a) bb_sorted[6 * i + a] = bb[6 * (u64)src + a];
        if (i == n - 1) *n_unique = (u64)seg[i];
        if (!head[i]) continue;
        const u64 k = skey[i];
        u64 sum = 0, j = i;
        for (; j < n && skey[j] == k; ++j) sum += sizes[perm[j]];
        const u32 s = seg[i] - 1u, last = perm[j - 1];
        uniq[s] = k; tot[s] = sum; seg_begin[s] = (u32)i;
        last_rc[3 * s] = rc[3 * (u64)last]; last_rc[3 * s + 1] = rc[3 * (u64)last + 1]; last_rc[3 * s + 2] = rc[3 * (u64)last + 2];
    }
}

__global__ __launch_bounds__(256) void k_reduce_pairs(const u64* ssub, const u64* scell, const u32* perm, const u32* head, const u32* seg,
                                                      const u64* counts, u64 n, u64* out_sub, u64* out_cell, u64* out_cnt, u64* n_unique) {
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) {
        if (i == n - 1) *n_unique = (u64)seg[i];
        if (!head[i]) continue;
        const u64 a = ssub[i], b = scell[i];
        u64 sum = 0;
        for (u64 j = i; j < n && ssub[j] == a && scell[j] == b; ++j) sum += counts[perm[j]];
        const u32 s = seg[i] - 1u;
        out_sub[s] = a; out_cell[s] = b; out_cnt[s] = sum;
    }
}

inline int grid_for(u64 n, int cap = 4096) { u64 g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > (u64)cap ? (u64)cap : g)); }
inline bool pow2(u64 v) { return v && !(v & (v - 1)); }
inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

// rocPRIM scratch for n records (radix sort of u64 keys with u32 values, inclusive scan of u32)
size_t prim_bytes(size_t n) {
    size_t a = 0, b = 0;
    u64* k = nullptr; u32* v = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, a, k, k, v, v, n, 0, 64, (hipStream_t)0);
    (void)rocprim::inclusive_scan(nullptr, b, v, v, n, rocprim::plus<u32>(), (hipStream_t)0);
    return up256(std::max(a, b));
}

// carve the caller's scratch: [u64 n | u64 n | u64 n | u32 n x 4 | rocPRIM]
struct Scratch { u64 *k0, *k1, *k2; u32 *i0, *i1, *head, *seg; void* prim; size_t prim_n; };
size_t scratch_bytes(size_t n) { return 3 * up256(n * 8) + 4 * up256(n * 4) + prim_bytes(n); }
Scratch carve(void* base, size_t n) {
    Scratch s; char* p = reinterpret_cast<char*>(base);
    s.k0 = reinterpret_cast<u64*>(p); p += up256(n * 8);
    s.k1 = reinterpret_cast<u64*>(p); p += up256(n * 8);
    s.k2 = reinterpret_cast<u64*>(p); p += up256(n * 8);
    s.i0 = reinterpret_cast<u32*>(p); p += up256(n * 4);
    s.i1 = reinterpret_cast<u32*>(p); p += up256(n * 4);
    s.head = reinterpret_cast<u32*>(p); p += up256(n * 4);
    s.seg = reinterpret_cast<u32*>(p); p += up256(n * 4);
    s.prim = p; s.prim_n = prim_bytes(n);
    return s;
}

}  // namespace

extern "C" {

int sd_chunkprops_append(const void* table, size_t cap_obj, int X, int Y, int Z, int ox, int oy, int oz, uint64_t min_obj_vx,
                         uint64_t* ids_dev, int32_t* rc_dev, int32_t* bbox_dev, uint64_t* sizes_dev, size_t max_records,
                         uint64_t* cursor_dev, void* stream) {
    if (!table || !pow2(cap_obj) || X <= 0 || Y <= 0 || Z <= 0 || !ids_dev || !rc_dev || !bbox_dev || !sizes_dev || !cursor_dev)
        return sd_fail_msg(SD_ERR_INVALID, "sd_chunkprops_append: bad argument");
    hipLaunchKernelGGL(k_chunkprops_append, dim3(grid_for(cap_obj)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       tab_view(table, cap_obj), X, Y, Z, ox, oy, oz, (u64)min_obj_vx, reinterpret_cast<u64*>(ids_dev), rc_dev, bbox_dev,
                       reinterpret_cast<u64*>(sizes_dev), (u64)max_records, reinterpret_cast<u64*>(cursor_dev));
    return hipGetLastError() == hipSuccess ? SD_OK : sd_fail_msg(SD_ERR_HIP, "sd_chunkprops_append: launch failed");
}

int sd_chunkpairs_append(const void* pair_table, size_t cap_pair, const void* sub_table, const void* cell_table, size_t cap_obj,
                         int X, int Y, int Z, uint64_t min_obj_vx, uint64_t* sub_ids_dev, uint64_t* cell_ids_dev, uint64_t* counts_dev,
                         size_t max_records, uint64_t* cursor_dev, void* stream) {
    if (!pair_table || !pow2(cap_pair) || !sub_table || !cell_table || !pow2(cap_obj) || X <= 0 || Y <= 0 || Z <= 0 || !sub_ids_dev ||
        !cell_ids_dev || !counts_dev || !cursor_dev)
        return sd_fail_msg(SD_ERR_INVALID, "sd_chunkpairs_append: bad argument");
    const u64* pk = reinterpret_cast<const u64*>(pair_table);
    hipLaunchKernelGGL(k_chunkpairs_append, dim3(grid_for(cap_pair)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), pk,
                       pk + cap_pair, (u64)cap_pair, tab_view(sub_table, cap_obj), reinterpret_cast<const u64*>(cell_table), X, Y, Z,
                       (u64)min_obj_vx, reinterpret_cast<u64*>(sub_ids_dev), reinterpret_cast<u64*>(cell_ids_dev),
                       reinterpret_cast<u64*>(counts_dev), (u64)max_records, reinterpret_cast<u64*>(cursor_dev));
    return hipGetLastError() == hipSuccess ? SD_OK : sd_fail_msg(SD_ERR_HIP, "sd_chunkpairs_append: launch failed");
}

size_t sd_propmerge_temp_bytes(size_t n_records) { return scratch_bytes(n_records ? n_records : 1); }

int sd_propmerge_objects(const uint64_t* ids_dev, const uint64_t* sizes_dev, const int32_t* rc_dev, const int32_t* bbox_dev, size_t n,
                         uint64_t* uniq_ids_dev, uint64_t* tot_sizes_dev, int32_t* last_rc_dev, uint32_t* seg_begin_dev,
                         int32_t* bbox_sorted_dev, uint64_t* n_unique_dev, void* temp_dev, size_t temp_bytes, void* stream) {
    if (!n_unique_dev) return sd_fail_msg(SD_ERR_INVALID, "sd_propmerge_objects: null count");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (hipMemsetAsync(n_unique_dev, 0, sizeof(uint64_t), s) != hipSuccess) return sd_fail_msg(SD_ERR_HIP, "memset failed");
    if (n == 0) return SD_OK;
    if (n >= (1ull << 32)) return sd_fail_msg(SD_ERR_INVALID, "sd_propmerge_objects: < 2^32 records per call");
    if (!ids_dev || !sizes_dev || !rc_dev || !bbox_dev || !uniq_ids_dev || !tot_sizes_dev || !last_rc_dev || !seg_begin_dev ||
        !bbox_sorted_dev || !temp_dev || temp_bytes < scratch_bytes(n))
        return sd_fail_msg(SD_ERR_INVALID, "sd_propmerge_objects: bad argument / scratch smaller than sd_propmerge_temp_bytes(n)");
    Scratch w = carve(temp_dev, n);
    const int g = grid_for(n);
    hipLaunchKernelGGL(k_iota, dim3(g), dim3(256), 0, s, w.i0, (u64)n);
    size_t pb = w.prim_n;
    if (rocprim::radix_sort_pairs(w.prim, pb, reinterpret_cast<const u64*>(ids_dev), w.k0, w.i0, w.i1, n, 0, 64, s) != hipSuccess)
        return sd_fail_msg(SD_ERR_HIP, "sd_propmerge_objects: radix sort failed");
    hipLaunchKernelGGL(k_heads, dim3(g), dim3(256), 0, s, w.k0, (const u64*)nullptr, w.head, (u64)n);
    pb = w.prim_n;
    if (rocprim::inclusive_scan(w.prim, pb, w.head, w.seg, n, rocprim::plus<u32>(), s) != hipSuccess)
        return sd_fail_msg(SD_ERR_HIP, "sd_propmerge_objects: scan failed");
    hipLaunchKernelGGL(k_reduce_objects, dim3(g), dim3(256), 0, s, w.k0, w.i1, w.head, w.seg, reinterpret_cast<const u64*>(sizes_dev),
                       rc_dev, bbox_dev, (u64)n, reinterpret_cast<u64*>(uniq_ids_dev), reinterpret_cast<u64*>(tot_sizes_dev), last_rc_dev,
                       seg_begin_dev, bbox_sorted_dev, reinterpret_cast<u64*>(n_unique_dev));
    return hipGetLastError() == hipSuccess ? SD_OK : sd_fail_msg(SD_ERR_HIP, "sd_propmerge_objects: launch failed");
}

int sd_propmerge_pairs(const uint64_t* sub_ids_dev, const uint64_t* cell_ids_dev, const uint64_t* counts_dev, size_t n,
                       uint64_t* out_sub_dev, uint64_t* out_cell_dev, uint64_t* out_counts_dev, uint64_t* n_unique_dev, void* temp_dev,
                       size_t temp_bytes, void* stream) {
    if (!n_unique_dev) return sd_fail_msg(SD_ERR_INVALID, "sd_propmerge_pairs: null count");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (hipMemsetAsync(n_unique_dev, 0, sizeof(uint64_t), s) != hipSuccess) return sd_fail_msg(SD_ERR_HIP, "memset failed");
    if (n == 0) return SD_OK;
    if (n >= (1ull << 32)) return sd_fail_msg(SD_ERR_INVALID, "sd_propmerge_pairs: < 2^32 records per call");
    if (!sub_ids_dev || !cell_ids_dev || !counts_dev || !out_sub_dev || !out_cell_dev || !out_counts_dev || !temp_dev ||
        temp_bytes < scratch_bytes(n))
        return sd_fail_msg(SD_ERR_INVALID, "sd_propmerge_pairs: bad argument / scratch smaller than sd_propmerge_temp_bytes(n)");
    Scratch w = carve(temp_dev, n);
    const int g = grid_for(n);
    const u64* sub = reinterpret_cast<const u64*>(sub_ids_dev);
    const u64* cell = reinterpret_cast<const u64*>(cell_ids_dev);
    // lexicographic (subcell id, cell id): stable sort by the minor key first, then by the major key
    hipLaunchKernelGGL(k_iota, dim3(g), dim3(256), 0, s, w.i0, (u64)n);
    size_t pb = w.prim_n;
    if (rocprim::radix_sort_pairs(w.prim, pb, cell, w.k0, w.i0, w.i1, n, 0, 64, s) != hipSuccess)
        return sd_fail_msg(SD_ERR_HIP, "sd_propmerge_pairs: radix sort failed");
    hipLaunchKernelGGL(k_gather64, dim3(g), dim3(256), 0, s, sub, w.i1, w.k1, (u64)n);
    pb = w.prim_n;
    if (rocprim::radix_sort_pairs(w.prim, pb, w.k1, w.k2, w.i1, w.i0, n, 0, 64, s) != hipSuccess)
        return sd_fail_msg(SD_ERR_HIP, "sd_propmerge_pairs: radix sort failed");
    hipLaunchKernelGGL(k_gather64, dim3(g), dim3(256), 0, s, cell, w.i0, w.k0, (u64)n);       // k2 = sorted subcell ids, k0 = their cell ids
    hipLaunchKernelGGL(k_heads, dim3(g), dim3(256), 0, s, w.k2, w.k0, w.head, (u64)n);
    pb = w.prim_n;
    if (rocprim::inclusive_scan(w.prim, pb, w.head, w.seg, n, rocprim::plus<u32>(), s) != hipSuccess)
        return sd_fail_msg(SD_ERR_HIP, "sd_propmerge_pairs: scan failed");
    hipLaunchKernelGGL(k_reduce_pairs, dim3(g), dim3(256), 0, s, w.k2, w.k0, w.i0, w.head, w.seg, reinterpret_cast<const u64*>(counts_dev),
                       (u64)n, reinterpret_cast<u64*>(out_sub_dev), reinterpret_cast<u64*>(out_cell_dev),
                       reinterpret_cast<u64*>(out_counts_dev), reinterpret_cast<u64*>(n_unique_dev));
    return hipGetLastError() == hipSuccess ? SD_OK : sd_fail_msg(SD_ERR_HIP, "sd_propmerge_pairs: launch failed");
}

}  // extern "C"
