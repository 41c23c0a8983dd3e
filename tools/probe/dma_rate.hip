// dev probe: sustained global_load_lds (LDS-DMA, 16 B per lane = 1 KiB per wave instruction) rate per CU for the
// address patterns of the conv kernel:
//   0: 1 KiB contiguous per instruction, source small and shared by all CUs (weights, L2-resident)
//   1: halo gather, voxel-major layout: lane pair -> 32-byte record, records 128 B apart (C = 64), rows of 18 voxels
//   2: halo gather, channel-blocked layout: records 32 B apart within a row of 18 voxels (576 B contiguous)
//   3: like 1 with C = 32 (records 64 B apart)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__device__ __forceinline__ void glds16(const void* g, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int MODE>
__global__ void __launch_bounds__(512) k_dma(const char* src, size_t span, long long* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // the block's region: WG b works in a window of the source
    const size_t wg_off = MODE == 0 ? 0 : ((size_t)blockIdx.x * 1315423 * 128) % span;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int instr = (it * 4 + j) * 8 + wave;           // instruction number within the WG stream
            const char* g;
            if (MODE == 0) {
                g = src + ((size_t)(instr % 216) * 1024) + lane * 16;                   // 216 KiB weight set
            } else {
                const int hv = (instr % 34) * 32 + (lane >> 1);                          // halo voxel 0..1087
                const int chunk = (instr / 34) & 3;
                const int hx = hv % 18, hy = (hv / 18) % 10, hz = hv / 180;
                const size_t vox = ((size_t)(hz * 64 + hy) * 64 + hx);                  // H = W = 64 volume
                if (MODE == 1) g = src + wg_off + vox * 128 + chunk * 32 + (lane & 1) * 16;
                else if (MODE == 3) g = src + wg_off + vox * 64 + (chunk & 1) * 32 + (lane & 1) * 16;
                else g = src + wg_off + ((size_t)chunk * (span / 4)) / 4 + vox * 32 + (lane & 1) * 16;
            }
            glds16(g, lds + ((instr & 31) * 1024));
        }
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    long long t1 = __builtin_readcyclecounter();
    if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE> void run(const char* what, int nwg) {
    const size_t span = (size_t)1 << 30;
    char* src; long long* d;
    if (hipMalloc(&src, span + (64 << 20)) != hipSuccess) { printf("malloc failed\n"); return; } (void)hipMemset(src, 1, span + (64 << 20)); (void)hipMalloc(&d, nwg * 64);
    const int iters = 400;
    (void)hipFuncSetAttribute((const void*)k_dma<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 32768);
    for (int rep = 0; rep < 2; ++rep) k_dma<MODE><<<nwg, 512, 32768>>>(src, span, d, iters);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return; }
    std::vector<long long> h(nwg * 8); (void)hipMemcpy(h.data(), d, nwg * 64, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    double c = (double)h[h.size() / 2];
    double per_instr = c / (iters * 4.0 * 8.0);          // WG-level: 8 waves issue concurrently
    printf("%-58s WGs %3d: %.1f cycles per 1-KiB DMA instruction per CU  (%.1f B/clk/CU)\n", what, nwg, per_instr, 1024.0 / per_instr);
    (void)hipFree(src); (void)hipFree(d);
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    run<0>("contiguous 1 KiB, shared 216 KiB source (weights)", 1);
    run<0>("contiguous 1 KiB, shared 216 KiB source (weights)", 256);
    run<1>("halo gather, voxel-major C=64 (32 B of each 128-B record)", 1);
    run<1>("halo gather, voxel-major C=64 (32 B of each 128-B record)", 256);
    run<3>("halo gather, voxel-major C=32 (32 B of each 64-B record)", 256);
    run<2>("halo gather, channel-blocked (rows of 576 B contiguous)", 1);
    run<2>("halo gather, channel-blocked (rows of 576 B contiguous)", 256);
    return 0;
}
