// First stage of probability map -> object segmentation on the device, non-watershed
// branches of /root/reference/syconn/extraction/object_extraction_steps.py:204-366 (_object_segmentation_thread):
//     :316-317  tmp_data = np.array(tmp_data > threshold, dtype=np.uint8)
//     :354-356  mop_data = apply_morphological_operations(tmp_data.copy(), morph_ops, mop_kwargs=dict(structure=struct))
//               this_labels_data, max_label = scipy.ndimage.label(mop_data)
//     :357-358  this_labels_data, max_label = scipy.ndimage.label(tmp_data)
// with the morphology semantics of /root/reference/syconn/proc/image.py:357-438 (_multi_mop_findobjects) on a binary
// volume: every operation acts on the bounding box of the foreground; closing / dilation pad that box by `iterations`
// zeros per side (less than the reach of the 5x5x3 element), clip their dilations to the padded window and erode with
// "outside the window = background"; opening erodes and dilates inside the box.  oracle/objseg_ref.py spells this out and
// is pinned to the reference's own functions (tests/golden/g9_objseg.npz).
//
// All passes are HBM-bound byte / int32 streams over the (x,y,z; z fastest) volume.  Connected components: 6-neighbour
// union-find on linear voxel indices (atomicMin links, roots = smallest index of a component = its first voxel in raster
// order), then the roots are ranked by an exclusive scan -> ids 1..N in scipy.ndimage.label's order, bit-exact.
#include "../../include/syconn_dense.h"
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <cmath>

extern int sd_fail_msg(int code, const char* msg);

namespace {

constexpr int MAX_OFFS = 128;
struct Offs { int n; signed char dx[MAX_OFFS], dy[MAX_OFFS], dz[MAX_OFFS]; };

struct Dom { int X, Y, Z, P; int PX, PY, PZ; };      // volume extents, pad, padded extents
__device__ __forceinline__ size_t pidx(const Dom& d, int x, int y, int z) { return ((size_t)x * d.PY + y) * d.PZ + z; }

// bbox[0..2] = min (padded coords), bbox[3..5] = max + 1; empty foreground: min > max
__global__ __launch_bounds__(256) void k_bbox_init(int* bbox) {
    if (threadIdx.x < 3) bbox[threadIdx.x] = 0x7fffffff;
    else if (threadIdx.x < 6) bbox[threadIdx.x] = 0;
}

__global__ __launch_bounds__(256) void k_threshold_pad(const uint8_t* prob, int cut, Dom d, uint8_t* A) {
    const size_t total = (size_t)d.PX * d.PY * d.PZ;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int z = (int)(i % d.PZ) - d.P, y = (int)((i / d.PZ) % d.PY) - d.P, x = (int)(i / ((size_t)d.PZ * d.PY)) - d.P;
        uint8_t v = 0;
        if ((unsigned)x < (unsigned)d.X && (unsigned)y < (unsigned)d.Y && (unsigned)z < (unsigned)d.Z)
            v = (int)prob[((size_t)x * d.Y + y) * d.Z + z] >= cut ? 1 : 0;
        A[i] = v;
    }
}

__global__ __launch_bounds__(256) void k_bbox(const uint8_t* A, Dom d, int* bbox) {
    const size_t total = (size_t)d.PX * d.PY * d.PZ;
    int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        if (A[i] == 1) {
            const int c[3] = {(int)(i / ((size_t)d.PZ * d.PY)), (int)((i / d.PZ) % d.PY), (int)(i % d.PZ)};
#pragma unroll
            for (int a = 0; a < 3; ++a) { lo[a] = min(lo[a], c[a]); hi[a] = max(hi[a], c[a] + 1); }
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        for (int m = 32; m >= 1; m >>= 1) {
            lo[a] = min(lo[a], __shfl_xor(lo[a], m, 64));
            hi[a] = max(hi[a], __shfl_xor(hi[a], m, 64));
        }
        if ((threadIdx.x & 63) == 0) {
            if (lo[a] != 0x7fffffff) atomicMin(&bbox[a], lo[a]);
            if (hi[a] != 0) atomicMax(&bbox[3 + a], hi[a]);
        }
    }
}

// One erosion (dilate == 0) or dilation (dilate == 1) step inside the window W = bbox grown by `wpad` per side (W lies
// inside the padded buffer because wpad <= P).  Outside W the result is 0; reads outside W count as background.
// crop != 0: additionally zero everything outside the bbox itself (the `res[n:-n, ...]` crop after a closing / dilation).
__global__ __launch_bounds__(256) void k_morph_step(const uint8_t* src, uint8_t* dst, Dom d, const int* bbox, int wpad,
                                                    int dilate, int crop, const Offs o) {
    const size_t total = (size_t)d.PX * d.PY * d.PZ;
    const int wl[3] = {bbox[0] - wpad, bbox[1] - wpad, bbox[2] - wpad};
    const int wh[3] = {bbox[3] + wpad, bbox[4] + wpad, bbox[5] + wpad};
    const int cl[3] = {crop ? bbox[0] : wl[0], crop ? bbox[1] : wl[1], crop ? bbox[2] : wl[2]};
    const int ch[3] = {crop ? bbox[3] : wh[0], crop ? bbox[4] : wh[1], crop ? bbox[5] : wh[2]};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int z = (int)(i % d.PZ), y = (int)((i / d.PZ) % d.PY), x = (int)(i / ((size_t)d.PZ * d.PY));
        uint8_t r = 0;
        if (x >= cl[0] && x < ch[0] && y >= cl[1] && y < ch[1] && z >= cl[2] && z < ch[2]) {
            r = dilate ? 0 : 1;
            for (int k = 0; k < o.n; ++k) {
                // erosion: all of v + S set; dilation: any of v - S set
                const int ux = dilate ? x - o.dx[k] : x + o.dx[k], uy = dilate ? y - o.dy[k] : y + o.dy[k],
                          uz = dilate ? z - o.dz[k] : z + o.dz[k];
                uint8_t s = 0;
                if (ux >= wl[0] && ux < wh[0] && uy >= wl[1] && uy < wh[1] && uz >= wl[2] && uz < wh[2])
                    s = src[pidx(d, ux, uy, uz)];
                if (dilate) { if (s) { r = 1; break; } }
                else if (!s) { r = 0; break; }
            }
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

__global__ __launch_bounds__(256) void k_cc_init(const uint8_t* A, Dom d, int* L, uint8_t* mask_out) {
    const size_t total = (size_t)d.X * d.Y * d.Z;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int z = (int)(i % d.Z), y = (int)((i / d.Z) % d.Y), x = (int)(i / ((size_t)d.Z * d.Y));
        const uint8_t v = A[pidx(d, x + d.P, y + d.P, z + d.P)];
        L[i] = v ? (int)i : -1;
        if (mask_out) mask_out[i] = v;
    }
}
__global__ __launch_bounds__(256) void k_cc_merge(Dom d, int* L) {
    const size_t total = (size_t)d.X * d.Y * d.Z;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        if (L[i] < 0) continue;
        const int z = (int)(i % d.Z), y = (int)((i / d.Z) % d.Y), x = (int)(i / ((size_t)d.Z * d.Y));
        if (z > 0 && L[i - 1] >= 0) cc_union(L, (int)i, (int)i - 1);
        if (y > 0 && L[i - d.Z] >= 0) cc_union(L, (int)i, (int)(i - d.Z));
        if (x > 0 && L[i - (size_t)d.Z * d.Y] >= 0) cc_union(L, (int)i, (int)(i - (size_t)d.Z * d.Y));
    }
}
__global__ __launch_bounds__(256) void k_cc_compress(size_t total, int* L) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256)
        if (L[i] >= 0) L[i] = cc_find(L, (int)i);
}

constexpr int SCAN_PER_THREAD = 8, SCAN_BLOCK = 256 * SCAN_PER_THREAD;
// roots per block of SCAN_BLOCK voxels
__global__ __launch_bounds__(256) void k_cc_count(size_t total, const int* L, int* blockcnt) {
    __shared__ int red[4];
    const size_t base = (size_t)blockIdx.x * SCAN_BLOCK + (size_t)threadIdx.x * SCAN_PER_THREAD;
    int c = 0;
#pragma unroll
    for (int k = 0; k < SCAN_PER_THREAD; ++k) c += (base + k < total && L[base + k] == (int)(base + k)) ? 1 : 0;
    for (int m = 32; m >= 1; m >>= 1) c += __shfl_xor(c, m, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) blockcnt[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// exclusive scan of blockcnt[0..n) in place, total -> *max_label (one workgroup)
__global__ __launch_bounds__(1024) void k_cc_scan_blocks(int* blockcnt, int n, int* max_label) {
    __shared__ int part[1024];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = i < n ? blockcnt[i] : 0;
        part[threadIdx.x] = v;
        __syncthreads();
        for (int s = 1; s < 1024; s <<= 1) {                  // Hillis-Steele inclusive scan
            const int t = threadIdx.x >= s ? part[threadIdx.x - s] : 0;
            __syncthreads();
            part[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < n) blockcnt[i] = carry + part[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += part[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) *max_label = carry;
}
// rank[root] = 1 + number of roots with a smaller raster index
__global__ __launch_bounds__(256) void k_cc_rank(size_t total, const int* L, const int* blockcnt, int* rank) {
    __shared__ int wsum[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t base = (size_t)blockIdx.x * SCAN_BLOCK + (size_t)threadIdx.x * SCAN_PER_THREAD;
    bool root[SCAN_PER_THREAD];
    int c = 0;
#pragma unroll
    for (int k = 0; k < SCAN_PER_THREAD; ++k) { root[k] = base + k < total && L[base + k] == (int)(base + k); c += root[k]; }
    int incl = c;                                             // inclusive scan over the wave's lanes
    for (int s = 1; s < 64; s <<= 1) { const int t = __shfl_up(incl, s, 64); if (lane >= s) incl += t; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int off = blockcnt[blockIdx.x] + incl - c;
    for (int w = 0; w < wave; ++w) off += wsum[w];
#pragma unroll
    for (int k = 0; k < SCAN_PER_THREAD; ++k)
        if (root[k]) rank[base + k] = ++off;
}
__global__ __launch_bounds__(256) void k_cc_relabel(size_t total, int* L, const int* rank) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int r = L[i];
        L[i] = r < 0 ? 0 : rank[r];      // rank[] is a separate buffer: roots read here are never overwritten
    }
}

inline int grid_for(size_t n, int cap = 8192) { size_t g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > (size_t)cap ? (size_t)cap : g)); }
inline size_t rup256(size_t v) { return (v + 255) & ~(size_t)255; }

struct WsLayout { size_t a, b, rank, blockcnt, bbox, total; };
WsLayout ws_layout(int X, int Y, int Z, int P) {
    WsLayout w{};
    const size_t pvox = (size_t)(X + 2 * P) * (Y + 2 * P) * (Z + 2 * P), nvox = (size_t)X * Y * Z;
    const size_t nblk = (nvox + SCAN_BLOCK - 1) / SCAN_BLOCK;
    size_t cur = 0;
    w.a = cur; cur += rup256(pvox);
    w.b = cur; cur += rup256(pvox);
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

int sd_object_segmentation(const uint8_t* prob_dev, int X, int Y, int Z, double threshold, const int32_t* ops,
                           const int32_t* iterations, int n_ops, const uint8_t* struct_host, int sx, int sy, int sz,
                           int32_t* labels_dev, int32_t* max_label_dev, uint8_t* mask_out_dev, void* ws, size_t ws_bytes,
                           void* stream) {
    if (!prob_dev || !labels_dev || !max_label_dev || !ws || X <= 0 || Y <= 0 || Z <= 0 || n_ops < 0 || (n_ops && (!ops || !iterations)))
        return sd_fail_msg(SD_ERR_INVALID, "sd_object_segmentation: bad argument");
    if ((size_t)X * Y * Z >= (1ull << 31)) return sd_fail_msg(SD_ERR_INVALID, "sd_object_segmentation: volume must have < 2^31 voxels");
    if (threshold != threshold) return sd_fail_msg(SD_ERR_INVALID, "sd_object_segmentation: NaN threshold");
    int P = 0;
    for (int i = 0; i < n_ops; ++i) {
        if (ops[i] == SD_MOP_EROSION)
            return sd_fail_msg(SD_ERR_INVALID, "sd_object_segmentation: binary_erosion selects the reference's watershed branch "
                                               "(object_extraction_steps.py:319-352), which is not implemented");
        if (ops[i] != SD_MOP_OPENING && ops[i] != SD_MOP_CLOSING && ops[i] != SD_MOP_DILATION)
            return sd_fail_msg(SD_ERR_INVALID, "sd_object_segmentation: unknown morphological operation");
        if (iterations[i] < 1 || iterations[i] > 64) return sd_fail_msg(SD_ERR_INVALID, "sd_object_segmentation: iterations out of range");
        if (ops[i] != SD_MOP_OPENING) P = std::max(P, iterations[i]);
    }
    Offs o{};
    if (n_ops) {
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
    }
    const WsLayout w = ws_layout(X, Y, Z, P);
    if (ws_bytes < w.total) return sd_fail_msg(SD_ERR_NOMEM, "sd_object_segmentation: workspace too small");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    char* const wb = reinterpret_cast<char*>(ws);
    uint8_t* A = reinterpret_cast<uint8_t*>(wb + w.a);
    uint8_t* B = reinterpret_cast<uint8_t*>(wb + w.b);
    int* rank = reinterpret_cast<int*>(wb + w.rank);
    int* blockcnt = reinterpret_cast<int*>(wb + w.blockcnt);
    int* bbox = reinterpret_cast<int*>(wb + w.bbox);
    Dom d{X, Y, Z, P, X + 2 * P, Y + 2 * P, Z + 2 * P};
    const size_t pvox = (size_t)d.PX * d.PY * d.PZ, nvox = (size_t)X * Y * Z;
    // (uint8 p > t) <=> p >= floor(t) + 1; threshold 0 means "already a 0/1 mask" (object_extraction_steps.py:316): cut 1
    const double c = std::floor(threshold) + 1.0;
    const int cut = threshold == 0.0 ? 1 : (c < 0.0 ? 0 : (c > 256.0 ? 256 : (int)c));
    hipLaunchKernelGGL(k_threshold_pad, dim3(grid_for(pvox)), dim3(256), 0, s, prob_dev, cut, d, A);
    for (int i = 0; i < n_ops; ++i) {
        const int n = iterations[i];
        hipLaunchKernelGGL(k_bbox_init, dim3(1), dim3(256), 0, s, bbox);
        hipLaunchKernelGGL(k_bbox, dim3(grid_for(pvox, 2048)), dim3(256), 0, s, A, d, bbox);
        const int wpad = ops[i] == SD_MOP_OPENING ? 0 : n;
        const int nfirst = n, nsecond = ops[i] == SD_MOP_DILATION ? 0 : n;
        const int first_dilate = ops[i] == SD_MOP_OPENING ? 0 : 1;
        for (int k = 0; k < nfirst + nsecond; ++k) {
            const int dil = k < nfirst ? first_dilate : 1 - first_dilate;
            const int crop = (k == nfirst + nsecond - 1) ? 1 : 0;
            hipLaunchKernelGGL(k_morph_step, dim3(grid_for(pvox)), dim3(256), 0, s, A, B, d, bbox, wpad, dil, crop, o);
            std::swap(A, B);
        }
    }
    int* L = labels_dev;
    hipLaunchKernelGGL(k_cc_init, dim3(grid_for(nvox)), dim3(256), 0, s, A, d, L, mask_out_dev);
    hipLaunchKernelGGL(k_cc_merge, dim3(grid_for(nvox)), dim3(256), 0, s, d, L);
    hipLaunchKernelGGL(k_cc_compress, dim3(grid_for(nvox)), dim3(256), 0, s, nvox, L);
    const int nblk = (int)((nvox + SCAN_BLOCK - 1) / SCAN_BLOCK);
    hipLaunchKernelGGL(k_cc_count, dim3(nblk), dim3(256), 0, s, nvox, L, blockcnt);
    hipLaunchKernelGGL(k_cc_scan_blocks, dim3(1), dim3(1024), 0, s, blockcnt, nblk, max_label_dev);
    hipLaunchKernelGGL(k_cc_rank, dim3(nblk), dim3(256), 0, s, nvox, L, blockcnt, rank);
    hipLaunchKernelGGL(k_cc_relabel, dim3(grid_for(nvox)), dim3(256), 0, s, nvox, L, rank);
    return hipGetLastError() == hipSuccess ? SD_OK : sd_fail_msg(SD_ERR_HIP, "sd_object_segmentation: launch failed");
}

}  // extern "C"
