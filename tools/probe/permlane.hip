// probe: exact lane semantics of v_permlane16_swap / v_permlane32_swap / DPP quad_perm on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    const unsigned l = threadIdx.x;
    unsigned a = l, b = 100 + l;
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[l] = r[0]; out[64 + l] = r[1];
    auto r2 = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[128 + l] = r2[0]; out[192 + l] = r2[1];
    out[256 + l] = __builtin_amdgcn_update_dpp(0, (int)l, 0xB1, 0xF, 0xF, true);
    auto r3 = __builtin_amdgcn_permlane16_swap(a, a, false, false);
    out[320 + l] = r3[0]; out[384 + l] = r3[1];
}
int main() {
    unsigned* d; hipMalloc(&d, 448 * 4);
    k<<<1, 64>>>(d);
    unsigned h[448]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char* names[] = {"p16 r0", "p16 r1", "p32 r0", "p32 r1", "dpp", "p16(a,a) r0", "p16(a,a) r1"};
    for (int s = 0; s < 7; ++s) { printf("%-12s", names[s]); for (int i = 0; i < 64; ++i) printf(" %u", h[s * 64 + i]); printf("\n"); }
    return 0;
}
