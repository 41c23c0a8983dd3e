// dev probe: sustained MFMA issue rate and shader clock under a chip-wide dense bf16 MFMA load.
// build: hipcc -O3 --offload-arch=gfx950 tools/probe/mfma_clock.hip -o /tmp/mfma_clock ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(8))) __bf16 v8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ void __launch_bounds__(512) k_mfma(long long* out, int iters, float* sink) {
    v8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x & 7); b[i] = (__bf16)(float)(i); }
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
    }
    long long t1 = __builtin_readcyclecounter();
    long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    if (s == 12345.678f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) {
        size_t w = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
        out[2 * w] = t1 - t0; out[2 * w + 1] = r1 - r0;
    }
}

int main() {
    const int iters = 20000;
    for (int waves : {4, 8}) {
        for (int nwg : {1, 256, 1024}) {
            long long* d; float* sink;
            size_t nw = (size_t)nwg * waves;
            hipMalloc(&d, nw * 16); hipMalloc(&sink, 4);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            k_mfma<<<nwg, waves * 64>>>(d, iters, sink);
            hipEventRecord(e0);
            k_mfma<<<nwg, waves * 64>>>(d, iters, sink);
            hipEventRecord(e1); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<long long> h(nw * 2); hipMemcpy(h.data(), d, nw * 16, hipMemcpyDeviceToHost);
            std::vector<double> cyc, rt;
            for (size_t i = 0; i < nw; ++i) { cyc.push_back((double)h[2 * i]); rt.push_back((double)h[2 * i + 1]); }
            std::sort(cyc.begin(), cyc.end()); std::sort(rt.begin(), rt.end());
            double c = cyc[nw / 2], r = rt[nw / 2];
            double flops = (double)nw * iters * 4 * 32768.0;
            printf("waves/WG %d  WGs %4d: kernel %.3f ms  %.1f TFLOP/s | per wave: s_memtime %.0f (%.2f per MFMA), "
                   "s_memrealtime %.0f ticks (100 MHz -> %.1f us) => s_memtime rate %.3f GHz\n",
                   waves, nwg, ms, flops / ms / 1e9, c, c / (iters * 4.0), r, r / 100.0, c / (r * 10.0));
            hipFree(d); hipFree(sink);
        }
    }
    return 0;
}
