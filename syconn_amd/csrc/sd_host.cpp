// Host-side helpers of the chunk pipeline (no GPU code): multi-threaded strided box copies between a (z,y,x) uint8
// volume in host memory and the dense pinned staging buffers of syconn_amd/parallel.py.  The reference cuts chunks with
// numpy slicing on one core (kd.load_raw / array slicing at /root/reference/syconn/handler/prediction.py:806-812); with
// eight GPUs behind one root process that copy is what bounds the node, so it is spread over host threads here.
#include "../../include/syconn_dense.h"
#include <algorithm>
#include <climits>
#include <cstring>
#include <map>
#include <thread>
#include <vector>

namespace {
template <typename F>
void parallel_rows(int64_t nrows, int64_t bytes_per_row, int n_threads, F&& f) {
    int nt = std::max(1, n_threads);
    const int64_t work = nrows * bytes_per_row;
    nt = (int)std::min<int64_t>(nt, std::max<int64_t>(1, work / (1 << 20)));      // at least ~1 MiB per thread
    if (nt <= 1) { f(0, nrows); return; }
    std::vector<std::thread> th;
    const int64_t per = (nrows + nt - 1) / nt;
    for (int t = 0; t < nt; ++t) {
        const int64_t a = t * per, b = std::min(nrows, a + per);
        if (a >= b) break;
        th.emplace_back([=, &f] { f(a, b); });
    }
    for (auto& x : th) x.join();
}

// ---- window clipping (sd_plan_clip_window) --------------------------------------------------------------------------------
inline int ksize(const sd_op_desc& d, int axis) { return axis == 0 ? d.kz : axis == 1 ? d.ky : d.kx; }
inline int64_t floordiv(int64_t a, int64_t b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }
inline int64_t ceildiv(int64_t a, int64_t b) { return -floordiv(-a, b); }

// Far side: smallest extent e (a multiple of `mult`, at most `full`) at which no output below `need` reads any buffer beyond
// its extent -- then the far border of every layer ('same' padding, partial ceil-mode pooling windows, the up-convolution
// crop) lies outside the cone of every wanted output.
int64_t clip_extent(const sd_op_desc* ops, int n, int axis, int64_t need, int64_t full, int64_t mult) {
    if (need >= full) return full;
    std::map<int, int64_t> reads;                  // buffer -> number of leading indices the wanted outputs read
    for (int i = n - 1; i >= 0; --i) {
        const sd_op_desc& d = ops[i];
        const int k = ksize(d, axis);
        int64_t r;
        if (d.kind == SD_OP_FINAL) r = need;
        else {
            auto it = reads.find(d.dst);
            if (it == reads.end()) continue;
            r = it->second;
        }
        if (d.kind == SD_OP_CONV) r += k / 2;
        else if (d.kind == SD_OP_POOL) r *= k;
        else if (d.kind == SD_OP_UPCONV) r = ceildiv(r, k);
        const int srcs[2] = {d.src0, d.kind == SD_OP_CONV ? d.src1 : -1};
        for (int s : srcs)
            if (s >= 0) { auto it = reads.find(s); if (it == reads.end() || it->second < r) reads[s] = r; }
    }
    auto fits = [&](int64_t e) {
        std::map<int, int64_t> ext;
        ext[0] = e;
        for (int i = 0; i < n; ++i) {
            const sd_op_desc& d = ops[i];
            if (d.kind == SD_OP_FINAL) continue;
            const int k = ksize(d, axis);
            int64_t a = ext[d.src0];
            if (d.kind == SD_OP_CONV && d.src1 >= 0) a = std::min(a, ext[d.src1]);       // autocrop: the larger operand loses its far end
            else if (d.kind == SD_OP_POOL) a = ceildiv(a, k);                              // ceil_mode
            else if (d.kind == SD_OP_UPCONV) a *= k;
            ext[d.dst] = a;
        }
        for (auto& br : reads)
            if (ext[br.first] < br.second) return false;
        return true;
    };
    auto it0 = reads.find(0);
    int64_t e = ceildiv(std::max(need, it0 == reads.end() ? need : it0->second), mult) * mult;
    while (e < full && !fits(e)) e += mult;
    return std::min(e, full);
}
}  // namespace

extern "C" {

int sd_host_box_copy(const uint8_t* src, int64_t src_stride_z, int64_t src_stride_y, uint8_t* dst, int64_t dst_stride_z,
                     int64_t dst_stride_y, int64_t nz, int64_t ny, int64_t nx, int n_threads) {
    if (!src || !dst || nz < 0 || ny < 0 || nx < 0) return SD_ERR_INVALID;
    if (nz == 0 || ny == 0 || nx == 0) return SD_OK;
    parallel_rows(nz * ny, nx, n_threads, [&](int64_t a, int64_t b) {
        for (int64_t r = a; r < b; ++r) {
            const int64_t z = r / ny, y = r - z * ny;
            std::memcpy(dst + z * dst_stride_z + y * dst_stride_y, src + z * src_stride_z + y * src_stride_y, (size_t)nx);
        }
    });
    return SD_OK;
}

int sd_host_zero(uint8_t* dst, int64_t nbytes, int n_threads) {
    if (!dst || nbytes < 0) return SD_ERR_INVALID;
    const int64_t row = 1 << 16, nrows = (nbytes + row - 1) / row;
    parallel_rows(nrows, row, n_threads, [&](int64_t a, int64_t b) {
        const int64_t lo = a * row, hi = std::min(nbytes, b * row);
        if (hi > lo) std::memset(dst + lo, 0, (size_t)(hi - lo));
    });
    return SD_OK;
}

int sd_plan_clip_window(const sd_op_desc* ops, int n_ops, int axis, int lo, int hi, int full, int multiple, int* start,
                        int* extent) {
    if (!ops || n_ops < 1 || axis < 0 || axis > 2 || !start || !extent || multiple < 1 || lo < 0 || hi < lo || hi > full)
        return SD_ERR_INVALID;
    *start = 0;
    *extent = full;
    for (int i = 0; i < n_ops; ++i) {
        const sd_op_desc& d = ops[i];
        if (d.kind == SD_OP_GROUPNORM) return SD_OK;                 // statistics over the whole window
        if (d.kind < SD_OP_CONV || d.kind > SD_OP_FINAL || d.src0 < 0 || d.dst < 0 || ksize(d, axis) < 1) return SD_ERR_INVALID;
    }
    if (ops[n_ops - 1].kind != SD_OP_FINAL) return SD_ERR_INVALID;
    int64_t s = 0;
    if (lo > 0) {
        // Near side: the window may start later by a multiple of the total pooling stride, as long as no wanted output
        // reads any buffer below its new first index.
        std::map<int, int64_t> scale, reads;           // buffer -> voxels of the input per element / lowest index read
        scale[0] = 1;
        for (int i = 0; i < n_ops; ++i) {
            const sd_op_desc& d = ops[i];
            if (d.kind == SD_OP_FINAL) continue;
            const int64_t sc = scale[d.src0];
            const int k = ksize(d, axis);
            scale[d.dst] = d.kind == SD_OP_POOL ? sc * k : d.kind == SD_OP_UPCONV ? std::max<int64_t>(1, sc / k) : sc;
        }
        for (int i = n_ops - 1; i >= 0; --i) {
            const sd_op_desc& d = ops[i];
            const int k = ksize(d, axis);
            int64_t r;
            if (d.kind == SD_OP_FINAL) r = lo;
            else {
                auto it = reads.find(d.dst);
                if (it == reads.end()) continue;
                r = it->second;
            }
            if (d.kind == SD_OP_CONV) r -= k / 2;
            else if (d.kind == SD_OP_POOL) r *= k;
            else if (d.kind == SD_OP_UPCONV) r = floordiv(r, k);
            const int srcs[2] = {d.src0, d.kind == SD_OP_CONV ? d.src1 : -1};
            for (int b : srcs)
                if (b >= 0) { auto it = reads.find(b); if (it == reads.end() || it->second > r) reads[b] = r; }
        }
        int64_t stride = 1, lowest = LLONG_MAX;
        for (auto& bs : scale) stride = std::max(stride, bs.second);
        int64_t g = stride, m = multiple;
        while (m) { const int64_t t = g % m; g = m; m = t; }
        stride = stride / g * multiple;
        for (auto& br : reads) lowest = std::min(lowest, br.second * scale[br.first]);
        if (lowest != LLONG_MAX) s = std::max<int64_t>(0, floordiv(lowest, stride) * stride);
    }
    *start = (int)s;
    *extent = (int)clip_extent(ops, n_ops, axis, hi - s, full - s, multiple);
    return SD_OK;
}

}  // extern "C"
