// dev probe: do VALU instructions of one wave issue under the MFMAs of another wave of the same SIMD (and of the same wave)?
// build: hipcc -O3 --offload-arch=gfx950 tools/probe/mfma_valu_overlap.hip -o /tmp/mvo ; run on the GPU box.
// 8 waves per workgroup (2 per SIMD), one workgroup per CU, 256 workgroups.  Modes:
//   0: every wave: MFMA only (4 independent accumulators)          1: every wave: VALU only (NV v_pk_max per iteration)
//   2: waves 0-3 MFMA only, waves 4-7 VALU only (SIMD partners)     3: every wave: 4 MFMA + NV independent VALU per iteration
//   4: every wave: 4 MFMA + NV ds_read_b128 per iteration           5: as 3 with dependent MFMAs (ONE accumulator)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 v8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u4;

template <int MODE, int NV>
__global__ void __launch_bounds__(512) k(float* sink, int iters) {
    __shared__ u4 lds[1024];
    v8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x & 7); b[i] = (__bf16)(float)(i); }
    lds[threadIdx.x] = u4{threadIdx.x, 1u, 2u, 3u}; lds[threadIdx.x + 512] = u4{5u, 6u, 7u, threadIdx.x};
    __syncthreads();
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    unsigned v[16];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * (i + 1);
    const int wave = threadIdx.x >> 6;
    const bool do_m = MODE == 0 || MODE == 3 || MODE == 4 || MODE == 5 || (MODE == 2 && wave < 4);
    const bool do_v = MODE == 1 || MODE == 3 || MODE == 5 || (MODE == 2 && wave >= 4);
    u4 acc4 = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        if (do_m) {
            if (MODE == 5) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
            } else {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
            }
        }
        if (do_v) {
#pragma unroll
            for (int j = 0; j < NV; ++j) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(v[j & 15]) : "v"(v[(j + 5) & 15]));
        }
        if (MODE == 4) {
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                u4 r;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"((unsigned)(threadIdx.x & 63) * 16), "n"((j & 15) * 1024));
                asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
                acc4.x ^= 1u;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i] + (float)v[i];
    if (s == 12345.678f) sink[0] = s + acc4.x;
}

template <int MODE, int NV>
static void run(const char* what, int iters) {
    float* sink; (void)hipMalloc(&sink, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE, NV><<<256, 512>>>(sink, iters);
    (void)hipEventRecord(e0);
    k<MODE, NV><<<256, 512>>>(sink, iters);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("mode %d NV %2d  %-62s %8.3f ms  = %7.1f ns per iteration\n", MODE, NV, what, ms, ms * 1e6 / iters);
    (void)hipFree(sink);
}

int main() {
    const int it = 20000;
    run<0, 0>("all waves: 4 MFMA / iteration (2 waves per SIMD)", it);
    run<1, 8>("all waves: 8 VALU / iteration", it);
    run<1, 16>("all waves: 16 VALU / iteration", it);
    run<1, 32>("all waves: 32 VALU / iteration", it);
    run<2, 8>("waves 0-3: 4 MFMA, waves 4-7 (SIMD partners): 8 VALU", it);
    run<2, 16>("waves 0-3: 4 MFMA, waves 4-7: 16 VALU", it);
    run<2, 32>("waves 0-3: 4 MFMA, waves 4-7: 32 VALU", it);
    run<3, 4>("every wave: 4 MFMA + 4 VALU", it);
    run<3, 8>("every wave: 4 MFMA + 8 VALU", it);
    run<3, 16>("every wave: 4 MFMA + 16 VALU", it);
    run<3, 32>("every wave: 4 MFMA + 32 VALU", it);
    run<5, 8>("every wave: 4 dependent MFMA + 8 VALU", it);
    run<5, 16>("every wave: 4 dependent MFMA + 16 VALU", it);
    run<4, 4>("every wave: 4 MFMA + 4 ds_read_b128", it);
    run<4, 8>("every wave: 4 MFMA + 8 ds_read_b128", it);
    run<4, 16>("every wave: 4 MFMA + 16 ds_read_b128", it);
    return 0;
}
