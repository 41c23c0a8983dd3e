// dev probe: how long does `s_waitcnt vmcnt(0)` take after a workgroup's output-tile store burst (64 KiB per WG),
// as a function of how many WGs burst at the same time and of the store pattern?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(8))) __bf16 v8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u4;

// mode 0: each lane stores 16 B into a different 128-B voxel record (8 instr fill the records) -- the conv epilogue pattern
// mode 1: each instr stores 1 KiB contiguous (lane-linear)
template <int MODE>
__global__ void __launch_bounds__(512) k_probe(char* out, long long* stamps, int rounds, int mfma_iters, int stagger) {
    v8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x & 7); b[i] = (__bf16)(float)(i); }
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long long wsum = 0, wmax = 0;
    int it0 = mfma_iters;
    if (stagger) it0 += (blockIdx.x % 8) * (mfma_iters / 8);
    for (int r = 0; r < rounds; ++r) {
        const int n = r == 0 ? it0 : mfma_iters;
        for (int i = 0; i < n; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
        }
        // this wave's 8 KiB of the WG's 64 KiB tile
        char* base = out + ((size_t)(r * gridDim.x + blockIdx.x) * 8 + wave) * 8192;
        u4 v = {__builtin_bit_cast(unsigned, c0[0]), __builtin_bit_cast(unsigned, c1[1]), __builtin_bit_cast(unsigned, c2[2]), (unsigned)r};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            char* p = MODE == 0 ? base + lane * 128 + k * 16 : base + k * 1024 + lane * 16;
            *reinterpret_cast<u4*>(p) = v;
        }
        long long t0 = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        long long t1 = __builtin_readcyclecounter();
        wsum += t1 - t0; wmax = max(wmax, t1 - t0);
    }
    if (lane == 0) { stamps[(blockIdx.x * 8 + wave) * 2] = wsum / rounds; stamps[(blockIdx.x * 8 + wave) * 2 + 1] = wmax; }
    if (c0[3] + c1[3] + c2[3] + c3[3] == 12345.f) out[0] = 1;
}

template <int MODE> void run(int nwg, int stagger, const char* what, int rounds = 8) {
    const int iters = 300;     // 1200 MFMAs/wave = 2 waves/SIMD -> ~77K cycles between bursts
    char* out; long long* st;
    (void)hipMalloc(&out, (size_t)rounds * nwg * 65536); (void)hipMalloc(&st, nwg * 8 * 16);
    for (int rep = 0; rep < 3; ++rep) k_probe<MODE><<<nwg, 512>>>(out, st, rounds, iters, stagger);
    (void)hipDeviceSynchronize();
    std::vector<long long> h(nwg * 16); (void)hipMemcpy(h.data(), st, nwg * 8 * 16, hipMemcpyDeviceToHost);
    std::vector<double> m, x;
    for (int i = 0; i < nwg * 8; ++i) { m.push_back((double)h[2 * i]); x.push_back((double)h[2 * i + 1]); }
    std::sort(m.begin(), m.end()); std::sort(x.begin(), x.end());
    printf("%-44s rounds %3d WGs %4d: vmcnt(0) after the burst: mean-per-wave median %6.0f p90 %6.0f cycles, worst round median %6.0f\n",
           what, rounds, nwg, m[m.size() / 2], m[m.size() * 9 / 10], x[x.size() / 2]);
    (void)hipFree(out); (void)hipFree(st);
}

int main() {
    run<0>(1, 0, "16 B per lane into 128-B records, 1 WG");
    run<1>(1, 0, "1 KiB contiguous per instr, 1 WG");
    run<0>(32, 0, "16 B/lane, 32 WGs in phase");
    run<0>(256, 0, "16 B/lane, 256 WGs in phase");
    run<1>(256, 0, "1 KiB contiguous, 256 WGs in phase");
    run<0>(256, 1, "16 B/lane, 256 WGs, 8 staggered phases");
    run<1>(256, 1, "1 KiB contiguous, 256 WGs, 8 staggered phases");
    run<0>(256, 0, "16 B/lane, 256 WGs in phase, 1 GiB footprint", 64);
    run<1>(256, 0, "1 KiB contiguous, 256 WGs in phase, 1 GiB footprint", 64);
    run<1>(256, 0, "1 KiB contiguous, 256 WGs in phase, 2 GiB footprint", 128);
    return 0;
}
