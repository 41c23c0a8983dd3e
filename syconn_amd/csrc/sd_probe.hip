// Box calibration for the benchmark line: what a chip-wide dense bf16 MFMA loop SUSTAINS on this part and at what shader clock.
// The nominal 2.5 PFLOP/s of gfx950 is a 2.4 GHz figure; under matrix load the part is power-limited and
// different boxes settle at different clocks, so bench.py prints this figure beside the roofline fraction of the same run.
// No memory traffic, no LDS: 4 independent accumulators per wave, v_mfma_f32_32x32x16_bf16 back to back, on constant or on
// pseudo-random operands.
#include "../../include/syconn_dense.h"
#include <hip/hip_runtime.h>
#include <algorithm>
#include <vector>

extern int sd_fail_msg(int code, const char* msg);

namespace {
typedef __attribute__((ext_vector_type(8))) __bf16 v8bf;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// RANDOM: four A and four B fragments of pseudo-random bf16 values (magnitude 0.06 ... 0.5, random sign -- what activations and weights
// look like) take turns, so that the operand buses toggle from one MFMA to the next as they do in a convolution; the constant
// small integers of the other form toggle nothing and draw visibly less power (the part then holds a higher clock).
template <bool RANDOM>
__global__ void __launch_bounds__(512) k_probe_mfma(unsigned long long* out, int iters, float* sink) {
    v8bf a[4], b[4];
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 8; ++i) {
            if (RANDOM) {
                unsigned h = (threadIdx.x * 8u + i) * 2654435761u + j * 40503u + blockIdx.x * 97u;
                h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
                const unsigned short ba = (unsigned short)(0x3d80u + (h & 0x17fu)) | (unsigned short)((h >> 3) & 0x8000u);
                const unsigned short bb = (unsigned short)(0x3d80u + ((h >> 9) & 0x17fu)) | (unsigned short)((h >> 5) & 0x8000u);
                a[j][i] = __builtin_bit_cast(__bf16, ba); b[j][i] = __builtin_bit_cast(__bf16, bb);
            } else {
                a[j][i] = (__bf16)(float)(threadIdx.x & 7); b[j][i] = (__bf16)(float)(i);
            }
        }
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[2], c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[3], b[3], c3, 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    if (s == 12345.678f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
        out[2 * w] = t1 - t0; out[2 * w + 1] = r1 - r0;
    }
}
}  // namespace

extern "C" int sd_probe_mfma_rate(int n_workgroups, int waves_per_workgroup, int iters, double min_seconds, int random_operands,
                                  double* tflops_out, double* shader_ghz_out, void* stream) {
    if (n_workgroups <= 0 || waves_per_workgroup <= 0 || waves_per_workgroup > 8 || iters <= 0 || !tflops_out || !shader_ghz_out)
        return sd_fail_msg(SD_ERR_INVALID, "sd_probe_mfma_rate: bad argument");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const size_t nw = (size_t)n_workgroups * waves_per_workgroup;
    unsigned long long* d = nullptr; float* sink = nullptr;
    if (hipMalloc(&d, nw * 16) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return sd_fail_msg(SD_ERR_NOMEM, "sd_probe_mfma_rate: hipMalloc");
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    // launches back to back until `min_seconds` have passed (the clock needs tens of milliseconds to settle under load); the LAST
    // launch is the one reported
    float ms = 0.f, total_ms = 0.f;
    int rc = SD_OK;
    do {
        (void)hipEventRecord(e0, s);
        if (random_operands) hipLaunchKernelGGL(k_probe_mfma<true>, dim3(n_workgroups), dim3(waves_per_workgroup * 64), 0, s, d, iters, sink);
        else hipLaunchKernelGGL(k_probe_mfma<false>, dim3(n_workgroups), dim3(waves_per_workgroup * 64), 0, s, d, iters, sink);
        (void)hipEventRecord(e1, s);
        if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) { rc = SD_ERR_HIP; break; }
        total_ms += ms;
    } while (total_ms < min_seconds * 1e3);
    if (rc == SD_OK) {
        std::vector<unsigned long long> h(nw * 2);
        if (hipMemcpy(h.data(), d, nw * 16, hipMemcpyDeviceToHost) != hipSuccess) rc = SD_ERR_HIP;
        else {
            std::vector<double> ghz(nw);
            for (size_t i = 0; i < nw; ++i) ghz[i] = h[2 * i + 1] ? (double)h[2 * i] / ((double)h[2 * i + 1] * 10.0) : 0.0;   // 100 MHz ticks
            std::nth_element(ghz.begin(), ghz.begin() + nw / 2, ghz.end());
            *shader_ghz_out = ghz[nw / 2];
            *tflops_out = (double)nw * iters * 4.0 * 32768.0 / (ms * 1e-3) / 1e12;
        }
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(d); (void)hipFree(sink);
    return rc == SD_OK ? SD_OK : sd_fail_msg(rc, "sd_probe_mfma_rate: HIP error");
}
