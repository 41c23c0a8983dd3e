// Label-volume statistics on the device: the gfx950 counterpart of the reference's only
// native code, /root/reference/syconn/extraction/find_object_properties_C.pyx (find_object_properties :24-49,
// map_subcell_C :72-109, map_subcell_extract_props :112-192).
//
// The reference scans the volume once on one core and keeps per-id properties in std::unordered_map.  Here the scan is one
// HBM-bound streaming pass (8 B per voxel and volume, read exactly once) and the maps are open-addressing hash tables in
// caller-owned device memory, updated with order-independent atomics (min / max / add), so the result is deterministic.
// Label volumes are spatially coherent, so the atomics are issued per RUN of equal labels along the fastest axis (found
// inside a wavefront with one ballot), not per voxel: a wave that sees one object in its 64 voxels issues one update.
#include "../../include/syconn_dense.h"
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <cstdlib>
#include <string>

extern int sd_fail_msg(int code, const char* msg);      // sd_api.hip: sets sd_last_error()

namespace {

typedef unsigned long long u64;
constexpr u64 EMPTY = 0ull;                               // label 0 is background and never inserted

// Object table of `cap` slots (power of two), structure of arrays in one buffer:
//   keys u64[cap] | first u64[cap] (smallest raster index) | size u64[cap] | bbmin i32[3][cap] | bbmax i32[3][cap]
struct ObjTable {
    u64* keys; u64* first; u64* size; int* bbmin; int* bbmax; u64 cap;
};
__host__ __device__ inline ObjTable obj_table(void* base, u64 cap) {
    ObjTable t;
    char* p = reinterpret_cast<char*>(base);
    t.keys = reinterpret_cast<u64*>(p);
    t.first = t.keys + cap;
    t.size = t.first + cap;
    t.bbmin = reinterpret_cast<int*>(t.size + cap);
    t.bbmax = t.bbmin + 3 * cap;
    t.cap = cap;
    return t;
}
constexpr size_t OBJ_SLOT_BYTES = 3 * 8 + 6 * 4;
// Pair table: keys u64[cap] ((subcell slot << 32 | cell slot) + 1) | count u64[cap]
constexpr size_t PAIR_SLOT_BYTES = 16;

__device__ __forceinline__ u64 mix64(u64 k) {             // murmur3 finaliser
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return k;
}

// slot of `k` in an open-addressing table (linear probing), inserting it if absent; -1 when the table is full
__device__ __forceinline__ long find_or_insert(u64* keys, u64 cap, u64 k) {
    const u64 mask = cap - 1;
    u64 h = mix64(k) & mask;
    for (u64 probe = 0; probe < cap; ++probe, h = (h + 1) & mask) {
        u64 cur = __hip_atomic_load(&keys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == k) return (long)h;
        if (cur == EMPTY) {
            const u64 old = atomicCAS(&keys[h], EMPTY, k);
            if (old == EMPTY || old == k) return (long)h;
        }
    }
    return -1;
}

__global__ __launch_bounds__(256) void k_obj_init(ObjTable t) {
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < t.cap; i += (u64)gridDim.x * 256) {
        t.keys[i] = EMPTY; t.first[i] = ~0ull; t.size[i] = 0;
#pragma unroll
        for (int a = 0; a < 3; ++a) { t.bbmin[a * t.cap + i] = 0x7fffffff; t.bbmax[a * t.cap + i] = 0; }
    }
}
__global__ __launch_bounds__(256) void k_zero64(u64* p, u64 n) {
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) p[i] = 0;
}

template <typename X> struct SegTag { using type = X; };
constexpr int MAX_SUB = 8;
struct ScanParams {
    const void* cell;                  // (X,Y,Z) labels, z fastest; may be nullptr (then only subcell properties)
    const void* sub[MAX_SUB];
    int n_sub;
    int X, Y, Z;
    ObjTable cell_t;
    ObjTable sub_t[MAX_SUB];
    u64* pair_keys[MAX_SUB]; u64* pair_cnt[MAX_SUB]; u64 pair_cap;
    int want_props;                    // 0: overlap counts only (map_subcell_C)
    int* status;                       // [0] object table overflow, [1] pair table overflow
};

template <typename L>
__device__ __forceinline__ u64 load_label(const void* vol, u64 i) { return (u64)reinterpret_cast<const L*>(vol)[i]; }

// run structure of a wave: `head` lanes start a run of equal values inside one z-row; returns the run length for head lanes
__device__ __forceinline__ int run_length(bool head, int lane, int nvalid) {
    const u64 m = __ballot(head);
    const u64 later = (lane == 63) ? 0ull : (m >> (lane + 1));
    int next = later ? (lane + 1 + __builtin_ctzll(later)) : 64;
    if (next > nvalid) next = nvalid;
    return next - lane;
}

__device__ __forceinline__ void obj_update(const ObjTable& t, u64 key, u64 lin, int x, int y, int z, int len, int* status) {
    const long s = find_or_insert(t.keys, t.cap, key);
    if (s < 0) { atomicExch(&status[0], 1); return; }
    atomicMin(&t.first[s], lin);
    atomicAdd(&t.size[s], (u64)len);
    atomicMin(&t.bbmin[0 * t.cap + s], x); atomicMin(&t.bbmin[1 * t.cap + s], y); atomicMin(&t.bbmin[2 * t.cap + s], z);
    atomicMax(&t.bbmax[0 * t.cap + s], x + 1); atomicMax(&t.bbmax[1 * t.cap + s], y + 1);
    atomicMax(&t.bbmax[2 * t.cap + s], z + len);
}

// Workgroup-local aggregation: a small open-addressing table per volume in LDS collects the updates of the workgroup's
// contiguous range of the volume (label volumes are spatially coherent: the same few ids come back row after row), and
// only its occupied slots are merged into the global table when the workgroup is done -- global atomics drop from one
// set per run to one set per (workgroup, id).  A full LDS table simply sends further new ids to the global table directly.
constexpr int LDS_SLOTS = 512;                 // shared by the 1 + n_sub volumes of a scan (power-of-two share each)
struct LTab { u64* keys; u64* first; unsigned* size; int* bb; int cap; };

__device__ __forceinline__ int lds_find_or_insert(u64* keys, int cap, u64 k) {
    const int mask = cap - 1;
    int h = (int)(mix64(k) & (u64)mask);
    for (int probe = 0; probe < 24 && probe < cap; ++probe, h = (h + 1) & mask) {
        const u64 cur = keys[h];
        if (cur == k) return h;
        if (cur == EMPTY) {
            const u64 old = atomicCAS(&keys[h], EMPTY, k);
            if (old == EMPTY || old == k) return h;
        }
    }
    return -1;
}

// LDS table of (subcell LDS slot, cell LDS slot) pairs -> overlap count of this workgroup's range (32-bit keys: both ids are
// named by their slots in the workgroup's LDS object tables)
constexpr int LDS_PSLOTS = 1024;
__device__ __forceinline__ int lds_pair_slot(unsigned* keys, int cap, unsigned k) {
    const int mask = cap - 1;
    int h = (int)((k * 2654435761u) >> 7) & mask;
    for (int probe = 0; probe < 16 && probe < cap; ++probe, h = (h + 1) & mask) {
        const unsigned cur = keys[h];
        if (cur == k) return h;
        if (cur == 0u) {
            const unsigned old = atomicCAS(&keys[h], 0u, k);
            if (old == 0u || old == k) return h;
        }
    }
    return -1;
}

__device__ __forceinline__ void run_update(const LTab& lt, const ObjTable& gt, u64 key, u64 lin, int x, int y, int z, int len,
                                           int* status) {
    if (lt.cap) {
        const int s = lds_find_or_insert(lt.keys, lt.cap, key);
        if (s >= 0) {
            atomicMin(&lt.first[s], lin);
            atomicAdd(&lt.size[s], (unsigned)len);
            atomicMin(&lt.bb[0 * lt.cap + s], x); atomicMin(&lt.bb[1 * lt.cap + s], y); atomicMin(&lt.bb[2 * lt.cap + s], z);
            atomicMax(&lt.bb[3 * lt.cap + s], x + 1); atomicMax(&lt.bb[4 * lt.cap + s], y + 1);
            atomicMax(&lt.bb[5 * lt.cap + s], z + len);
            return;
        }
    }
    obj_update(gt, key, lin, x, y, z, len, status);
}

__device__ __forceinline__ void lds_flush(const LTab& lt, const ObjTable& gt, int* status) {
    for (int s = threadIdx.x; s < lt.cap; s += 256) {
        const u64 k = lt.keys[s];
        if (k == EMPTY) continue;
        const long g = find_or_insert(gt.keys, gt.cap, k);
        if (g < 0) { atomicExch(&status[0], 1); continue; }
        atomicMin(&gt.first[g], lt.first[s]);
        atomicAdd(&gt.size[g], (u64)lt.size[s]);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            atomicMin(&gt.bbmin[a * gt.cap + g], lt.bb[a * lt.cap + s]);
            atomicMax(&gt.bbmax[a * gt.cap + g], lt.bb[(3 + a) * lt.cap + s]);
        }
    }
}

// One wave = 64 consecutive voxels of the flattened volume; a workgroup owns a contiguous range of such wave-chunks.  Per
// volume a lane is a run head iff it is the wave's first lane, the first voxel of a z-row, or its label differs from the
// previous voxel's.
// V4 (rows of a multiple of 4 voxels, 16-byte aligned volumes): a LANE owns 4 consecutive voxels (one or two 16-byte loads per
// volume), a wave 256.  Run heads inside a lane are four compares, run lengths come from one ballot (lanes that hold a head), the
// lane's own 4-bit head mask and the head mask of the next lane that holds a head; the voxel coordinates (two integer divisions)
// are computed once per lane and iteration for all volumes.  The first form of this pass (one voxel per lane) spent ~300
// instructions per 64 voxels and volume -- it was instruction-bound at 0.9 TB/s of label reads; runs, atomics and results are the
// same in both forms (sums / minima / maxima do not depend on where a wave cuts a run).
template <typename L, bool V4, bool HC = false, int NS = -1>
__global__ __launch_bounds__(256) void k_segstats_scan(const ScanParams p, const int lcap) {
    __shared__ u64 l_keys[LDS_SLOTS];
    __shared__ u64 l_first[LDS_SLOTS];
    __shared__ unsigned l_size[LDS_SLOTS];
    __shared__ int l_bb[6 * LDS_SLOTS];
    __shared__ unsigned l_pkey[LDS_PSLOTS];      // overlap pairs of this workgroup (see lds_pair_slot), split over the subcell volumes
    __shared__ unsigned l_pcnt[LDS_PSLOTS];
    const int lane = threadIdx.x & 63;
    const int nvol = (p.cell ? 1 : 0) + p.n_sub;
    int pcap = 0;                                 // pair slots per subcell volume (power of two)
    if (p.cell && p.n_sub > 0 && p.want_props && lcap) { pcap = LDS_PSLOTS; while (pcap * p.n_sub > LDS_PSLOTS) pcap >>= 1; }
    for (int i = threadIdx.x; i < LDS_SLOTS; i += 256) {
        l_keys[i] = EMPTY; l_first[i] = ~0ull; l_size[i] = 0;
    }
    for (int i = threadIdx.x; i < LDS_PSLOTS; i += 256) { l_pkey[i] = 0u; l_pcnt[i] = 0u; }
    for (int i = threadIdx.x; i < 6 * LDS_SLOTS; i += 256) {
        // table t occupies slots [t*lcap, (t+1)*lcap): its bbox block is [6*t*lcap, 6*(t+1)*lcap), min rows first
        const int t = lcap ? i / (6 * lcap) : 0, r = lcap ? (i - t * 6 * lcap) / lcap : 0;
        l_bb[i] = r < 3 ? 0x7fffffff : 0;
    }
    __syncthreads();
    auto ltab = [&](int t) {
        LTab lt;
        lt.cap = (p.want_props && lcap && t < nvol) ? lcap : 0;
        lt.keys = l_keys + t * lcap; lt.first = l_first + t * lcap; lt.size = l_size + t * lcap; lt.bb = l_bb + 6 * t * lcap;
        return lt;
    };
    const LTab cell_lt = ltab(0);
    const u64 nvox = (u64)p.X * p.Y * p.Z;
    constexpr int VPW = V4 ? 256 : 64;            // voxels per wave and iteration
    const u64 nwaves = (nvox + VPW - 1) / VPW;
    const u64 per_wg = (nwaves + gridDim.x - 1) / gridDim.x;
    const u64 w_end = min(nwaves, (u64)(blockIdx.x + 1) * per_wg);
    if constexpr (V4) {
        // pair / object update of the heads at in-lane position J of one volume (hm: this lane's head mask, bit j = voxel j starts a run)
        auto run_len = [&](int j, unsigned hm, bool has_next, int nl, unsigned nhm) -> int {
            const unsigned rest = hm >> (j + 1);
            if (rest) return __builtin_ctz(rest) + 1;                                   // the next head is in this lane
            if (!has_next) return (4 - j) + 4 * (63 - lane);                            // ... or nowhere in this wave's 256 voxels
            return (4 - j) + 4 * (nl - lane - 1) + __builtin_ctz(nhm);                  // ... or in lane nl
        };
        // next lane above this one that holds a head (and that lane's head mask), from the ballot of head-holding lanes
        auto next_head = [&](unsigned hm, bool& has_next, int& nl, unsigned& nhm) {
            const u64 any = __ballot(hm != 0u);
            const u64 later = (lane == 63) ? 0ull : (any >> (lane + 1));
            has_next = later != 0ull;
            nl = has_next ? lane + 1 + __builtin_ctzll(later) : lane;
            nhm = (unsigned)__shfl((int)hm, nl, 64);
        };
        auto load4 = [&](const void* vol, u64 first, u64 (&v)[4]) {
            if constexpr (sizeof(L) == 8) {
                typedef __attribute__((ext_vector_type(2))) u64 u64x2;
                const u64x2 a = reinterpret_cast<const u64x2*>(vol)[first / 2], b = reinterpret_cast<const u64x2*>(vol)[first / 2 + 1];
                v[0] = a[0]; v[1] = a[1]; v[2] = b[0]; v[3] = b[1];
            } else {
                typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
                const u32x4 a = reinterpret_cast<const u32x4*>(vol)[first / 4];
                v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
            }
        };
        // head mask of a lane's four labels: bit 0 needs the previous lane's last label (a row start and the wave's first lane always
        // start a run); lanes beyond the volume hold zeros and a head at bit 0, which ends the last real run
        auto heads = [&](const u64 (&v)[4], bool valid, bool row_start) -> unsigned {
            const u64 prev = __shfl_up(v[3], 1, 64);
            unsigned hm = (!valid || row_start || v[0] != prev) ? 1u : 0u;
            hm |= (v[1] != v[0]) ? 2u : 0u; hm |= (v[2] != v[1]) ? 4u : 0u; hm |= (v[3] != v[2]) ? 8u : 0u;
            return valid ? hm : 1u;
        };
        // one subcell volume of this lane's four voxels: its own properties and its overlap with the cell labels
        auto do_sub = [&](int ii, const u64 (&sk)[4], const u64 (&ck)[4], unsigned chm, u64 first, bool valid, bool row_start, int x, int y, int z) {
            const unsigned shm = heads(sk, valid, row_start);
            if (p.want_props) {
                bool hn; int nl; unsigned nhm;
                next_head(shm, hn, nl, nhm);
                const LTab slt = ltab((p.cell ? 1 : 0) + ii);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (!__any((shm >> j) & 1u)) continue;
                    if (((shm >> j) & 1u) && sk[j] != 0)
                        run_update(slt, p.sub_t[ii], sk[j], first + j, x, y, z + j, run_len(j, shm, hn, nl, nhm), p.status);
                }
            }
            if (p.cell) {
                // overlap counts: runs of a constant (subcell id, cell id) pair start wherever either volume starts a run
                const unsigned phm = chm | shm;
                bool hn; int nl; unsigned nhm;
                next_head(phm, hn, nl, nhm);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool mine = ((phm >> j) & 1u) && sk[j] != 0 && ck[j] != 0;
                    if (!__any(mine)) continue;
                    if (!mine) continue;
                    const int plen = run_len(j, phm, hn, nl, nhm);
                    if (pcap) {
                        const LTab slt = ltab(1 + ii);
                        const int ls = lds_find_or_insert(slt.keys, slt.cap, sk[j]), lc = lds_find_or_insert(cell_lt.keys, cell_lt.cap, ck[j]);
                        if (ls >= 0 && lc >= 0) {
                            const int ps = lds_pair_slot(l_pkey + ii * pcap, pcap, (((unsigned)ls << 16) | (unsigned)lc) + 1u);
                            if (ps >= 0) { atomicAdd(&l_pcnt[ii * pcap + ps], (unsigned)plen); continue; }
                        }
                    }
                    const long ss = find_or_insert(p.sub_t[ii].keys, p.sub_t[ii].cap, sk[j]);
                    const long cs = find_or_insert(p.cell_t.keys, p.cell_t.cap, ck[j]);
                    if (ss < 0 || cs < 0) { atomicExch(&p.status[0], 1); continue; }
                    const long ps = find_or_insert(p.pair_keys[ii], p.pair_cap, (((u64)ss << 32) | (u64)cs) + 1);
                    if (ps < 0) { atomicExch(&p.status[1], 1); continue; }
                    atomicAdd(&p.pair_cnt[ii][ps], (u64)plen);
                }
            }
        };
        // PF (NS >= 0: the launch has exactly NS subcell volumes and, with HC, a cell volume -- at most four volumes): the labels of
        // the NEXT iteration are requested before this one is worked on, all volumes at once -- 8 KiB per wave in flight instead of
        // 2 KiB (SQ counters of the sequential form: waves wait 60 % of their cycles, 0.31 busy)
        constexpr bool PF = NS >= 0;
        constexpr int C0 = HC ? 1 : 0, NT = PF ? (C0 + NS > 0 ? C0 + NS : 1) : 1;
        u64 cur[NT][4], nxt[NT][4];
        auto fetch_all = [&](u64 w, u64 (&buf)[NT][4]) {
            const u64 f = w * 256 + 4 * (u64)lane;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                buf[t][0] = buf[t][1] = buf[t][2] = buf[t][3] = 0;
                if (f < nvox) load4(t < C0 ? p.cell : p.sub[t - C0], f, buf[t]);
            }
        };
        const u64 w_first = (u64)blockIdx.x * per_wg + (threadIdx.x >> 6);
        if constexpr (PF) { if (w_first < w_end) fetch_all(w_first, cur); }
        for (u64 w = w_first; w < w_end; w += 4) {
            if constexpr (PF) { if (w + 4 < w_end) fetch_all(w + 4, nxt); }
            const u64 first = w * 256 + 4 * (u64)lane;            // this lane's first voxel (a multiple of 4: never straddles a z-row)
            const bool valid = first < nvox;
            const u64 li = valid ? first : (nvox - 4);
            int z, y, x;
            if (nvox < (1ull << 32)) {
                const unsigned u = (unsigned)li, r = u / (unsigned)p.Z, q = r / (unsigned)p.Y;
                z = (int)(u - r * (unsigned)p.Z); y = (int)(r - q * (unsigned)p.Y); x = (int)q;
            } else {
                z = (int)(li % p.Z); y = (int)((li / p.Z) % p.Y); x = (int)(li / ((u64)p.Z * p.Y));
            }
            const bool row_start = (lane == 0) || (z == 0);
            u64 ck[4] = {0, 0, 0, 0};
            unsigned chm = valid ? 0u : 1u;
            if (p.cell) {
                if constexpr (PF && HC) { ck[0] = cur[0][0]; ck[1] = cur[0][1]; ck[2] = cur[0][2]; ck[3] = cur[0][3]; }
                else if (valid) load4(p.cell, first, ck);
                chm = heads(ck, valid, row_start);
                if (p.want_props) {
                    bool hn; int nl; unsigned nhm;
                    next_head(chm, hn, nl, nhm);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (!__any((chm >> j) & 1u)) continue;
                        if (((chm >> j) & 1u) && ck[j] != 0)
                            run_update(cell_lt, p.cell_t, ck[j], first + j, x, y, z + j, run_len(j, chm, hn, nl, nhm), p.status);
                    }
                }
            }
            if constexpr (PF) {
#pragma unroll
                for (int ii = 0; ii < NS; ++ii) do_sub(ii, cur[C0 + ii], ck, chm, first, valid, row_start, x, y, z);
#pragma unroll
                for (int t = 0; t < NT; ++t) { cur[t][0] = nxt[t][0]; cur[t][1] = nxt[t][1]; cur[t][2] = nxt[t][2]; cur[t][3] = nxt[t][3]; }
            } else {
                for (int ii = 0; ii < p.n_sub; ++ii) {
                    u64 sk[4] = {0, 0, 0, 0};
                    if (valid) load4(p.sub[ii], first, sk);
                    do_sub(ii, sk, ck, chm, first, valid, row_start, x, y, z);
                }
            }
        }
    } else
    for (u64 w = (u64)blockIdx.x * per_wg + (threadIdx.x >> 6); w < w_end; w += 4) {
        const u64 base = w * 64, lin = base + lane;
        const int nvalid = (int)((nvox - base) < 64 ? (nvox - base) : 64);
        const bool valid = lane < nvalid;
        const u64 li = valid ? lin : (nvox - 1);
        // (32-bit decode where the volume allows it: three 64-bit divisions per voxel made this pass instruction-bound)
        int z, y, x;
        if (nvox < (1ull << 32)) {
            const unsigned u = (unsigned)li, r = u / (unsigned)p.Z, q = r / (unsigned)p.Y;
            z = (int)(u - r * (unsigned)p.Z); y = (int)(r - q * (unsigned)p.Y); x = (int)q;
        } else {
            z = (int)(li % p.Z); y = (int)((li / p.Z) % p.Y); x = (int)(li / ((u64)p.Z * p.Y));
        }
        const bool row_start = (lane == 0) || (z == 0);
        u64 ck = 0;
        bool chead = false;
        if (p.cell) {
            ck = valid ? load_label<L>(p.cell, lin) : 0;
            const u64 prev = __shfl_up(ck, 1, 64);
            chead = valid && (row_start || ck != prev);
            if (p.want_props) {
                const int len = run_length(chead, lane, nvalid);
                if (chead && ck != 0) run_update(cell_lt, p.cell_t, ck, lin, x, y, z, len, p.status);
            }
        }
        for (int ii = 0; ii < p.n_sub; ++ii) {
            const u64 sk = valid ? load_label<L>(p.sub[ii], lin) : 0;
            const u64 prev = __shfl_up(sk, 1, 64);
            const bool shead = valid && (row_start || sk != prev);
            if (p.want_props) {
                const int len = run_length(shead, lane, nvalid);
                if (shead && sk != 0) run_update(ltab((p.cell ? 1 : 0) + ii), p.sub_t[ii], sk, lin, x, y, z, len, p.status);
            }
            if (p.cell) {
                // overlap counts: runs of a constant (subcell id, cell id) pair; both ids get a slot in their GLOBAL tables
                // here (their properties may still sit in the LDS tables), the two slots name the pair
                const bool phead = chead || shead;
                const int plen = run_length(phead, lane, nvalid);
                if (phead && sk != 0 && ck != 0) {
                    // workgroup-local first: the pair is named by the two ids' slots in the LDS object tables and counted in
                    // LDS; the three global hash lookups + the global atomic happen once per (workgroup, pair) at the end
                    if (pcap) {
                        const LTab slt = ltab(1 + ii);
                        const int ls = lds_find_or_insert(slt.keys, slt.cap, sk), lc = lds_find_or_insert(cell_lt.keys, cell_lt.cap, ck);
                        if (ls >= 0 && lc >= 0) {
                            const int ps = lds_pair_slot(l_pkey + ii * pcap, pcap, (((unsigned)ls << 16) | (unsigned)lc) + 1u);
                            if (ps >= 0) { atomicAdd(&l_pcnt[ii * pcap + ps], (unsigned)plen); continue; }
                        }
                    }
                    const long ss = find_or_insert(p.sub_t[ii].keys, p.sub_t[ii].cap, sk);
                    const long cs = find_or_insert(p.cell_t.keys, p.cell_t.cap, ck);
                    if (ss < 0 || cs < 0) { atomicExch(&p.status[0], 1); continue; }
                    const long ps = find_or_insert(p.pair_keys[ii], p.pair_cap, (((u64)ss << 32) | (u64)cs) + 1);
                    if (ps < 0) { atomicExch(&p.status[1], 1); continue; }
                    atomicAdd(&p.pair_cnt[ii][ps], (u64)plen);
                }
            }
        }
    }
    __syncthreads();
    if (pcap) {
        for (int ii = 0; ii < p.n_sub; ++ii) {
            const LTab slt = ltab(1 + ii);
            for (int sidx = threadIdx.x; sidx < pcap; sidx += 256) {
                const unsigned k = l_pkey[ii * pcap + sidx];
                if (k == 0u) continue;
                const u64 sk = slt.keys[(k - 1u) >> 16], ck = cell_lt.keys[(k - 1u) & 0xffffu];
                const long ss = find_or_insert(p.sub_t[ii].keys, p.sub_t[ii].cap, sk);
                const long cs = find_or_insert(p.cell_t.keys, p.cell_t.cap, ck);
                if (ss < 0 || cs < 0) { atomicExch(&p.status[0], 1); continue; }
                const long ps = find_or_insert(p.pair_keys[ii], p.pair_cap, (((u64)ss << 32) | (u64)cs) + 1);
                if (ps < 0) { atomicExch(&p.status[1], 1); continue; }
                atomicAdd(&p.pair_cnt[ii][ps], (u64)l_pcnt[ii * pcap + sidx]);
            }
        }
    }
    if (p.want_props && lcap) {
        if (p.cell) lds_flush(cell_lt, p.cell_t, p.status);
        for (int ii = 0; ii < p.n_sub; ++ii) lds_flush(ltab((p.cell ? 1 : 0) + ii), p.sub_t[ii], p.status);
    }
}

__global__ __launch_bounds__(256) void k_obj_compact(ObjTable t, u64* ids, u64* first, u64* size, int* bb, u64* count, u64 max_out) {
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < t.cap; i += (u64)gridDim.x * 256) {
        const u64 k = t.keys[i];
        if (k == EMPTY || t.size[i] == 0) continue;       // size 0: id only inserted for the overlap table
        const u64 o = atomicAdd(count, 1ull);
        if (o >= max_out) continue;
        ids[o] = k; first[o] = t.first[i]; size[o] = t.size[i];
#pragma unroll
        for (int a = 0; a < 3; ++a) { bb[o * 6 + a] = t.bbmin[a * t.cap + i]; bb[o * 6 + 3 + a] = t.bbmax[a * t.cap + i]; }
    }
}

__global__ __launch_bounds__(256) void k_pair_compact(const u64* pkeys, const u64* pcnt, u64 pcap, const u64* sub_keys,
                                                      const u64* cell_keys, u64* out_sub, u64* out_cell, u64* out_cnt,
                                                      u64* count, u64 max_out) {
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < pcap; i += (u64)gridDim.x * 256) {
        const u64 k = pkeys[i];
        if (k == EMPTY) continue;
        const u64 o = atomicAdd(count, 1ull);
        if (o >= max_out) continue;
        out_sub[o] = sub_keys[(k - 1) >> 32];
        out_cell[o] = cell_keys[(k - 1) & 0xffffffffull];
        out_cnt[o] = pcnt[i];
    }
}

inline bool pow2(u64 v) { return v && !(v & (v - 1)); }
// ---- globally unique object ids across chunks (object_extraction_steps.py:369-443 make_unique_labels, :658-736 apply_merge_list)
// per-chunk int32 component labels -> uint64 ids shifted by the chunk's offset (background stays 0)
__global__ __launch_bounds__(256) void k_labels_offset(const int32_t* __restrict__ lab, u64 n, u64 offset, u64* __restrict__ out) {
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) {
        const int32_t v = lab[i];
        out[i] = v > 0 ? (u64)v + offset : 0ull;
    }
}
// dst (nx,ny,nz contiguous, z fastest) = lut[src[x0 + i, y0 + j, z0 + k]] (lut == nullptr: the ids themselves); ids beyond the table
// raise the flag and pass through
__global__ __launch_bounds__(256) void k_labels_box_lut(const u64* __restrict__ src, int Y, int Z, int x0, int y0, int z0, int nx, int ny,
                                                        int nz, const u64* __restrict__ lut, u64 lut_len, u64* __restrict__ dst,
                                                        int32_t* status) {
    const u64 n = (u64)nx * ny * nz;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) {
        const unsigned k = (unsigned)(i % (unsigned)nz), ij = (unsigned)(i / (unsigned)nz);
        const unsigned j = ij % (unsigned)ny, ii = ij / (unsigned)ny;
        u64 v = src[((size_t)(x0 + ii) * Y + (y0 + j)) * Z + (z0 + k)];
        if (lut) {
            if (v < lut_len) v = lut[v];
            else if (status) *status = 1;
        }
        dst[i] = v;
    }
}

inline int grid_for(u64 n, int cap = 4096) { u64 g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > (u64)cap ? (u64)cap : g)); }

}  // namespace

extern "C" {

size_t sd_objtable_bytes(size_t capacity) { return capacity * OBJ_SLOT_BYTES; }
size_t sd_pairtable_bytes(size_t capacity) { return capacity * PAIR_SLOT_BYTES; }

int sd_segstats_scan(const void* cell_dev, const void* const* sub_devs, int n_sub, int dtype, int X, int Y, int Z,
                     void* cell_table, void* const* sub_tables, size_t cap_obj, void* const* pair_tables, size_t cap_pair,
                     int want_props, int32_t* status_dev, void* stream) {
    if ((!cell_dev && n_sub <= 0) || n_sub < 0 || n_sub > MAX_SUB || X <= 0 || Y <= 0 || Z <= 0 || !status_dev)
        return sd_fail_msg(SD_ERR_INVALID, "sd_segstats_scan: bad argument");
    if (dtype != SD_U32 && dtype != SD_U64) return sd_fail_msg(SD_ERR_INVALID, "sd_segstats_scan: dtype must be SD_U32 or SD_U64");
    if (!pow2(cap_obj) || cap_obj > (1ull << 31) || (n_sub > 0 && cell_dev && (!pow2(cap_pair) || !pair_tables)))
        return sd_fail_msg(SD_ERR_INVALID, "sd_segstats_scan: capacities must be powers of two (objects <= 2^31)");
    if ((cell_dev && !cell_table) || (n_sub > 0 && (!sub_devs || !sub_tables)))
        return sd_fail_msg(SD_ERR_INVALID, "sd_segstats_scan: null table");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    ScanParams p{};
    p.cell = cell_dev; p.n_sub = n_sub; p.X = X; p.Y = Y; p.Z = Z; p.want_props = want_props; p.status = status_dev;
    p.pair_cap = cap_pair;
    if (hipMemsetAsync(status_dev, 0, 2 * sizeof(int32_t), s) != hipSuccess) return sd_fail_msg(SD_ERR_HIP, "memset failed");
    if (cell_dev) {
        p.cell_t = obj_table(cell_table, cap_obj);
        hipLaunchKernelGGL(k_obj_init, dim3(grid_for(cap_obj)), dim3(256), 0, s, p.cell_t);
    }
    for (int i = 0; i < n_sub; ++i) {
        if (!sub_devs[i] || !sub_tables[i]) return sd_fail_msg(SD_ERR_INVALID, "sd_segstats_scan: null subcell volume / table");
        p.sub[i] = sub_devs[i];
        p.sub_t[i] = obj_table(sub_tables[i], cap_obj);
        hipLaunchKernelGGL(k_obj_init, dim3(grid_for(cap_obj)), dim3(256), 0, s, p.sub_t[i]);
        if (cell_dev) {
            if (!pair_tables[i]) return sd_fail_msg(SD_ERR_INVALID, "sd_segstats_scan: null pair table");
            p.pair_keys[i] = reinterpret_cast<u64*>(pair_tables[i]);
            p.pair_cnt[i] = p.pair_keys[i] + cap_pair;
            hipLaunchKernelGGL(k_zero64, dim3(grid_for(2 * cap_pair)), dim3(256), 0, s, p.pair_keys[i], (u64)2 * cap_pair);
        }
    }
    // four voxels per lane where every lane's 16 / 32 bytes are aligned and lie inside one z-row
    bool v4 = Z % 4 == 0 && getenv("SD_SEGSTATS_V1") == nullptr;
    const uintptr_t amask = 15;
    if (cell_dev && (reinterpret_cast<uintptr_t>(cell_dev) & amask)) v4 = false;
    for (int i = 0; i < n_sub; ++i) if (reinterpret_cast<uintptr_t>(sub_devs[i]) & amask) v4 = false;
    const u64 nwaves = ((u64)X * Y * Z + (v4 ? 255 : 63)) / (v4 ? 256 : 64);
    // each workgroup owns a contiguous range of >= 64 (16) wave-chunks (4096 voxels = a few z-rows); LDS slots per volume
    const int grid = (int)std::max<u64>(1, std::min<u64>((nwaves + (v4 ? 15 : 63)) / (v4 ? 16 : 64), 256 * 8));
    const int nvol = (cell_dev ? 1 : 0) + n_sub;
    int lcap = 32;
    while (lcap * 2 * nvol <= LDS_SLOTS) lcap *= 2;
    if (lcap * nvol > LDS_SLOTS) lcap = 0;            // too many volumes for the shared LDS table: global updates only
    static const bool no_lds = getenv("SD_SEGSTATS_NO_LDS") != nullptr;      // debugging aid / A-B
    if (no_lds) lcap = 0;
    // kernel form: one voxel per lane; four voxels per lane; ... with all volumes of the next iteration prefetched (<= 4 volumes)
    static const bool no_pf = getenv("SD_SEGSTATS_NO_PREFETCH") != nullptr;      // A/B
    auto launch = [&](auto tag) {
        using LT = typename decltype(tag)::type;
        void (*k)(const ScanParams, const int) = k_segstats_scan<LT, false>;
        if (v4) {
            k = k_segstats_scan<LT, true>;
            const int nt = nvol;
            if (!no_pf && nt <= 4) {
                if (cell_dev) k = n_sub == 0 ? k_segstats_scan<LT, true, true, 0> : n_sub == 1 ? k_segstats_scan<LT, true, true, 1>
                                : n_sub == 2 ? k_segstats_scan<LT, true, true, 2> : k_segstats_scan<LT, true, true, 3>;
                else k = n_sub == 1 ? k_segstats_scan<LT, true, false, 1> : n_sub == 2 ? k_segstats_scan<LT, true, false, 2>
                         : n_sub == 3 ? k_segstats_scan<LT, true, false, 3> : k_segstats_scan<LT, true, false, 4>;
            }
        }
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, s, p, lcap);
    };
    if (dtype == SD_U64) launch(SegTag<uint64_t>{}); else launch(SegTag<uint32_t>{});
    return hipGetLastError() == hipSuccess ? SD_OK : sd_fail_msg(SD_ERR_HIP, "sd_segstats_scan: launch failed");
}

int sd_labels_make_unique(const int32_t* labels_dev, size_t n, uint64_t offset, uint64_t* out_dev, void* stream) {
    if (!labels_dev || !out_dev) return sd_fail_msg(SD_ERR_INVALID, "sd_labels_make_unique: null argument");
    if (n == 0) return SD_OK;
    hipLaunchKernelGGL(k_labels_offset, dim3(grid_for(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), labels_dev, (u64)n,
                       (u64)offset, reinterpret_cast<u64*>(out_dev));
    return hipGetLastError() == hipSuccess ? SD_OK : sd_fail_msg(SD_ERR_HIP, "sd_labels_make_unique: launch failed");
}

int sd_labels_box_lut(const uint64_t* src_dev, int X, int Y, int Z, int x0, int y0, int z0, int nx, int ny, int nz,
                      const uint64_t* lut_dev, size_t lut_len, uint64_t* dst_dev, int32_t* status_dev, void* stream) {
    if (!src_dev || !dst_dev || X <= 0 || Y <= 0 || Z <= 0 || x0 < 0 || y0 < 0 || z0 < 0 || nx < 0 || ny < 0 || nz < 0 ||
        (long)x0 + nx > X || (long)y0 + ny > Y || (long)z0 + nz > Z)
        return sd_fail_msg(SD_ERR_INVALID, "sd_labels_box_lut: box outside the volume");
    const u64 n = (u64)nx * ny * nz;
    if (n == 0) return SD_OK;
    if (n >= (1ull << 32)) return sd_fail_msg(SD_ERR_INVALID, "sd_labels_box_lut: box must have < 2^32 voxels");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (status_dev && hipMemsetAsync(status_dev, 0, sizeof(int32_t), s) != hipSuccess) return sd_fail_msg(SD_ERR_HIP, "memset failed");
    hipLaunchKernelGGL(k_labels_box_lut, dim3(grid_for(n)), dim3(256), 0, s, reinterpret_cast<const u64*>(src_dev), Y, Z, x0, y0, z0, nx, ny,
                       nz, reinterpret_cast<const u64*>(lut_dev), (u64)lut_len, reinterpret_cast<u64*>(dst_dev), status_dev);
    return hipGetLastError() == hipSuccess ? SD_OK : sd_fail_msg(SD_ERR_HIP, "sd_labels_box_lut: launch failed");
}

int sd_segstats_compact_objects(const void* table, size_t cap_obj, uint64_t* ids_dev, uint64_t* first_dev, uint64_t* size_dev,
                                int32_t* bbox_dev, size_t max_out, uint64_t* count_dev, void* stream) {
    if (!table || !pow2(cap_obj) || !ids_dev || !first_dev || !size_dev || !bbox_dev || !count_dev)
        return sd_fail_msg(SD_ERR_INVALID, "sd_segstats_compact_objects: bad argument");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (hipMemsetAsync(count_dev, 0, sizeof(uint64_t), s) != hipSuccess) return sd_fail_msg(SD_ERR_HIP, "memset failed");
    hipLaunchKernelGGL(k_obj_compact, dim3(grid_for(cap_obj)), dim3(256), 0, s, obj_table(const_cast<void*>(table), cap_obj),
                       reinterpret_cast<u64*>(ids_dev), reinterpret_cast<u64*>(first_dev), reinterpret_cast<u64*>(size_dev),
                       bbox_dev, reinterpret_cast<u64*>(count_dev), (u64)max_out);
    return hipGetLastError() == hipSuccess ? SD_OK : sd_fail_msg(SD_ERR_HIP, "sd_segstats_compact_objects: launch failed");
}

int sd_segstats_compact_pairs(const void* pair_table, size_t cap_pair, const void* sub_table, const void* cell_table,
                              size_t cap_obj, uint64_t* sub_ids_dev, uint64_t* cell_ids_dev, uint64_t* counts_dev,
                              size_t max_out, uint64_t* count_dev, void* stream) {
    if (!pair_table || !pow2(cap_pair) || !sub_table || !cell_table || !pow2(cap_obj) || !sub_ids_dev || !cell_ids_dev ||
        !counts_dev || !count_dev)
        return sd_fail_msg(SD_ERR_INVALID, "sd_segstats_compact_pairs: bad argument");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (hipMemsetAsync(count_dev, 0, sizeof(uint64_t), s) != hipSuccess) return sd_fail_msg(SD_ERR_HIP, "memset failed");
    const u64* pk = reinterpret_cast<const u64*>(pair_table);
    hipLaunchKernelGGL(k_pair_compact, dim3(grid_for(cap_pair)), dim3(256), 0, s, pk, pk + cap_pair, (u64)cap_pair,
                       reinterpret_cast<const u64*>(sub_table), reinterpret_cast<const u64*>(cell_table),
                       reinterpret_cast<u64*>(sub_ids_dev), reinterpret_cast<u64*>(cell_ids_dev),
                       reinterpret_cast<u64*>(counts_dev), reinterpret_cast<u64*>(count_dev), (u64)max_out);
    return hipGetLastError() == hipSuccess ? SD_OK : sd_fail_msg(SD_ERR_HIP, "sd_segstats_compact_pairs: launch failed");
}

}  // extern "C"
