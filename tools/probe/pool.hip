#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__device__ __forceinline__ float max_xor1(float m) {
    return fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0xB1, 0xF, 0xF, true)));
}
__device__ __forceinline__ float max_xor16(float m) {
#ifdef USE_BUILTIN
    const unsigned u = __builtin_bit_cast(unsigned, m);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);   // hipcc drops the fmaxf below (!)
    return fmaxf(__builtin_bit_cast(float, r[0]), __builtin_bit_cast(float, r[1]));
#else
    unsigned a = __builtin_bit_cast(unsigned, m), b = a;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return fmaxf(__builtin_bit_cast(float, a), __builtin_bit_cast(float, b));
#endif
}
__global__ void k(const float* in, float* out) {
    const unsigned l = threadIdx.x;
    float m = in[l];
    out[l] = max_xor1(m);
    out[64 + l] = max_xor16(m);
    out[128 + l] = max_xor16(max_xor1(m));
}
int main() {
    float h[64], *di, *dout, o[192];
    for (int i = 0; i < 64; ++i) h[i] = (float)((i * 37) % 64) - 20.f;
    hipMalloc(&di, 256); hipMalloc(&dout, 768);
    hipMemcpy(di, h, 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(di, dout);
    hipMemcpy(o, dout, 768, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        float e1 = fmaxf(h[l], h[l ^ 1]), e16 = fmaxf(h[l], h[l ^ 16]);
        float e = fmaxf(fmaxf(h[l], h[l ^ 1]), fmaxf(h[l ^ 16], h[l ^ 17]));
        if (o[l] != e1 || o[64 + l] != e16 || o[128 + l] != e) { ++bad; printf("lane %d: %g/%g %g/%g %g/%g\n", l, o[l], e1, o[64+l], e16, o[128+l], e); }
    }
    printf("bad lanes: %d\n", bad);
}
