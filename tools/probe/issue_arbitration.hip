// dev probe: how do the two waves of a SIMD share the issue port?  Waves 0-3 of a workgroup (one per SIMD) run an MFMA program for a
// fixed number of iterations, their SIMD partners (waves 4-7) run a VALU program until the first group is done and count iterations.
// Reported in shader cycles (s_memtime), so DVFS does not matter.
// build: hipcc -O3 --offload-arch=gfx950 tools/probe/issue_arbitration.hip -o /tmp/arb ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 v8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// MA: 0 = waves 0-3 idle, 1 = 4 independent MFMAs per iteration, 2 = 4 dependent MFMAs, 3 = 2 chains of 2
// VB: VALU instructions per iteration of waves 4-7 (0 = idle);  PR: 0 none, 1 = s_setprio 3 for waves 4-7, 2 = for waves 0-3
template <int MA, int VB, int PR>
__global__ void __launch_bounds__(512) k(long long* out, float* sink, int iters) {
    __shared__ volatile int done;
    v8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x & 7); b[i] = (__bf16)(float)(i); }
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    unsigned v[16];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * (i + 1);
    const int wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) done = 0;
    __syncthreads();
    if (PR == 1 && wave >= 4) asm volatile("s_setprio 3");
    if (PR == 2 && wave < 4) asm volatile("s_setprio 3");
    long long t0 = __builtin_readcyclecounter(), n = 0;
    if (wave < 4) {
        if (MA) {
            for (int it = 0; it < iters; ++it) {
                if (MA == 1) {
                    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
                    c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
                } else if (MA == 2) {
                    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
                    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
                    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
                    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
                } else {
                    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
                    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
                }
            }
            n = iters;
        }
        if (threadIdx.x == 0) done = 1;
    } else if (VB) {
        if (MA == 0) {
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int j = 0; j < VB; ++j) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[j & 15]) : "v"(v[(j + 5) & 15]));
            }
            n = iters;
        } else {
            // (a fixed count that takes about as long alone as the partner's MFMA loop: 128 cycles per MFMA iteration, ~6 per VALU)
            const int nb = iters * 128 / (6 * VB);
            for (int it = 0; it < nb; ++it) {
#pragma unroll
                for (int j = 0; j < VB; ++j) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[j & 15]) : "v"(v[(j + 5) & 15]));
            }
            n = nb;
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i] + (float)v[i];
    if (s == 12345.678f) sink[0] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { out[2 * wave] = t1 - t0; out[2 * wave + 1] = n; }
}

template <int MA, int VB, int PR>
static void run(const char* what, int nwg) {
    const int iters = 4000;
    long long* d; float* sink;
    (void)hipMalloc(&d, 16 * 8); (void)hipMalloc(&sink, 4);
    k<MA, VB, PR><<<nwg, 512>>>(d, sink, iters);
    k<MA, VB, PR><<<nwg, 512>>>(d, sink, iters);
    (void)hipDeviceSynchronize();
    long long h[16];
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const double ca = (double)h[0] / (h[1] ? h[1] : 1), cb = (double)h[8] / (h[9] ? h[9] : 1);
    printf("%-58s WGs %3d | MFMA wave: %7.1f cyc / iteration (4 MFMA) | VALU wave: %7.1f cyc / iteration (%2d VALU) = %5.2f cyc / VALU", what, nwg,
           MA ? ca : 0.0, VB ? cb : 0.0, VB, VB ? cb / VB : 0.0);
    if (MA && VB) printf(" | wave durations %lld / %lld", h[0], h[8]);
    printf("\n");
    (void)hipFree(d); (void)hipFree(sink);
}

int main() {
    for (int nwg : {1, 256}) {
        run<1, 0, 0>("independent MFMAs alone", nwg);
        run<2, 0, 0>("dependent MFMAs alone", nwg);
        run<3, 0, 0>("two chains alone", nwg);
        run<0, 16, 0>("VALU alone (16 / iteration)", nwg);
        run<1, 16, 0>("independent MFMAs | partner VALU", nwg);
        run<2, 16, 0>("dependent MFMAs | partner VALU", nwg);
        run<3, 16, 0>("two chains | partner VALU", nwg);
        run<1, 16, 1>("independent MFMAs | partner VALU at priority 3", nwg);
        run<2, 16, 1>("dependent MFMAs | partner VALU at priority 3", nwg);
        run<1, 16, 2>("independent MFMAs at priority 3 | partner VALU", nwg);
        run<2, 16, 2>("dependent MFMAs at priority 3 | partner VALU", nwg);
    }
    return 0;
}
