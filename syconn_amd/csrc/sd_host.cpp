// Host-side helpers of the chunk pipeline (no GPU code): multi-threaded strided box copies between a (z,y,x) uint8
// volume in host memory and the dense pinned staging buffers of syconn_amd/parallel.py.  The reference cuts chunks with
// numpy slicing on one core (kd.load_raw / array slicing at /root/reference/syconn/handler/prediction.py:806-812); with
// eight GPUs behind one root process that copy is what bounds the node, so it is spread over host threads here.
#include "../../include/syconn_dense.h"
#include <algorithm>
#include <cstring>
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

}  // extern "C"
