// Device-side helpers shared by the kernel translation units (sd_kernels.hip, sd_dec0.hip): storage types, the MFMA
// wrappers, packed 16-bit max, explicit LDS reads, cross-lane swaps, LDS-DMA.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <utility>

typedef __bf16 bf16_t;
typedef _Float16 f16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <typename T> struct Act;
template <> struct Act<bf16_t> {
    using v8 = bf16x8;
    using v4 = bf16x4;
    static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ unsigned pack2(float a, float b) {     // RNE, one v_cvt_pk_bf16_f32
        typedef __attribute__((ext_vector_type(2))) float f2;
        typedef __attribute__((ext_vector_type(2))) __bf16 t2;
        return __builtin_bit_cast(unsigned, __builtin_convertvector(f2{a, b}, t2));
    }
};
template <> struct Act<f16_t> {
    using v8 = f16x8;
    using v4 = f16x4;
    static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ unsigned pack2(float a, float b) {     // RNE, one v_cvt_pk_f16_f32
        typedef __attribute__((ext_vector_type(2))) float f2;
        typedef __attribute__((ext_vector_type(2))) _Float16 t2;
        return __builtin_bit_cast(unsigned, __builtin_convertvector(f2{a, b}, t2));
    }
};
// split-fp16 plan: two fp32 values -> packed fp16 hi pair and packed fp16 lo pair, v = hi + lo up to 2^-24 |v| (lo normal) /
// 2^-25 absolute (lo subnormal)
__device__ __forceinline__ void split_pk(float a, float b, unsigned& hi, unsigned& lo) {
    typedef __attribute__((ext_vector_type(2))) _Float16 h2;
    hi = Act<f16_t>::pack2(a, b);
    const h2 h = __builtin_bit_cast(h2, hi);
    lo = Act<f16_t>::pack2(a - (float)h[0], b - (float)h[1]);
}
// ... and back: the exact fp32 value of a (hi, lo) pair of packed fp16 pairs
__device__ __forceinline__ void join_pk(unsigned hi, unsigned lo, float& a, float& b) {
    typedef __attribute__((ext_vector_type(2))) _Float16 h2;
    const h2 h = __builtin_bit_cast(h2, hi), l = __builtin_bit_cast(h2, lo);
    a = (float)h[0] + (float)l[0];
    b = (float)h[1] + (float)l[1];
}
// max of two packed pairs of NON-NEGATIVE-or-any 16-bit floats against each other as signed 16-bit integers: for
// sign-magnitude floats this is the float max whenever at most one operand is negative (ReLU: max(x, +0) is exact
// for every x incl. -0; pooling: all operands are >= 0 after the ReLU).  One v_pk_max_i16 for two channels.
__device__ __forceinline__ unsigned pk_max16(unsigned a, unsigned b) {
    typedef __attribute__((ext_vector_type(2))) short s2;
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s2, a), __builtin_bit_cast(s2, b)));
}

// Planar first convolution (1 -> 32 channels) on UINT8 input as two bf16 MFMAs (weight layout: sd_api.hip, `w3_off`).  The input
// patch holds every voxel as the dword (bf16(v), bf16(v)) -- a uint8 is exact in bf16 --, r[0..4] = this lane's five patch reads
// (lanes 0-31: taps 0-4; lanes 32-63: taps 5-8 and the constant dword (1.0, 1.0), which meets the bias parts).  Products
// v * weight part are exact in fp32, the accumulation is the matrix core's fp32: conv(float32(v) / 255) to fp32 rounding.
constexpr unsigned SD_BF16_ONE_PAIR = 0x3f803f80u;
__device__ __forceinline__ unsigned u8_bf16_pair(unsigned v) {      // v in 0..255 -> (bf16(v), bf16(v))
    const unsigned b = __builtin_bit_cast(unsigned, (float)v);     // (low 16 bits of the float are zero: 8 significant bits)
    return b | (b >> 16);
}
__device__ __forceinline__ f32x16 first_u8_mfma(bf16x8 w0, bf16x8 w1, const unsigned (&r)[5]) {
    typedef __attribute__((ext_vector_type(4))) unsigned u4;
    const unsigned m = 0xffffu;
    const u4 b0 = {r[0], r[1], r[2], r[3]};
    const u4 b1 = {r[4], (r[0] & m) | (r[1] & ~m), (r[2] & m) | (r[3] & ~m), (r[4] & m) | (SD_BF16_ONE_PAIR & ~m)};
    f32x16 a;
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = 0.f;
    a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, __builtin_bit_cast(bf16x8, b0), a, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, __builtin_bit_cast(bf16x8, b1), a, 0, 0, 0);
}

// Hardware places workgroup b on XCD b % 8 (observed; used for L2 locality only).  Map it to a logical block id
// such that each XCD owns a contiguous run of logical ids (neighbouring blocks share halo voxels in that L2).
// Bijective for any grid size.
// ---- explicit LDS reads (see the tap loop of k_conv_mfma) --------------------------------------------------------
template <int OFF, typename V>
__device__ __forceinline__ void ds_read16(V& r, uint32_t addr) {     // 16 bytes per lane; completion via lgkmcnt
    static_assert(sizeof(V) == 16 && OFF >= 0 && OFF < 65536, "ds_read_b128 offset");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
}
template <typename V> __device__ __forceinline__ void tie(V& r) { asm volatile("" : "+v"(r)); }
__device__ __forceinline__ uint32_t lds_addr(const void* p) {        // LDS byte offset of a pointer into shared memory
    return (uint32_t)(uintptr_t)p;
}
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

__device__ __forceinline__ int xcd_remap(int b, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = b & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}

// Cross-lane helpers (gfx950): v_permlane32_swap exchanges the upper half-wave of `a` with the lower half-wave
// of `b`; v_permlane16_swap exchanges odd 16-lane rows of `a` with even rows of `b`.  Inline asm on purpose:
// with the builtin hipcc (ROCm 7.2) folds away arithmetic that combines the two results when both inputs hold the
// same value (tools/probe/pool.hip shows the dropped v_max).  `s_nop 1` = the 2 wait states a VALU write of an
// operand needs before v_permlane*_swap reads it (nothing pads hazards inside an asm statement); the trailing
// one keeps a dependent VALU read of the results out of the swap's shadow.
__device__ __forceinline__ void swap32(unsigned& a, unsigned& b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void swap16(unsigned& a, unsigned& b) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void swap32x4(unsigned& a0, unsigned& b0, unsigned& a1, unsigned& b1, unsigned& a2, unsigned& b2,
                                         unsigned& a3, unsigned& b3) {     // four independent swaps, one hazard pad
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\t"
                 "v_permlane32_swap_b32 %4, %5\n\tv_permlane32_swap_b32 %6, %7\n\ts_nop 1"
                 : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3));
}
// Pooling of RAW (signed, not yet normalised) packed 16-bit floats behind a deferred GroupNorm: y = relu(a x + b) is monotone in
// x -- rising for a >= 0, falling for a < 0 (the sign of a is the sign of gamma) --, so max over the window of y is y at the
// window's largest (smallest) x.  A packed pair becomes a pair of SIGNED 16-bit keys (pk_max16 is v_pk_max_i16) whose order is
// the float order (sign-magnitude -> two's complement: negative floats flip their magnitude bits), complemented where the minimum
// is wanted; PK_KEY_LOWEST is below every key (voxels beyond the volume).  The same function maps back.
constexpr unsigned PK_KEY_LOWEST = 0x80008000u;
__device__ __forceinline__ unsigned pk_order_key(unsigned u, unsigned dir) {
    const unsigned neg = ((u >> 15) & 0x00010001u) * 0xffffu;
    return u ^ (neg & 0x7fff7fffu) ^ dir;
}
__device__ __forceinline__ unsigned pk_order_unkey(unsigned k, unsigned dir) {
    const unsigned v = k ^ dir;
    const unsigned neg = ((v >> 15) & 0x00010001u) * 0xffffu;
    return v ^ (neg & 0x7fff7fffu);
}
// packed-pair max over the 2x2 (y,x) pooling window: lane^1 by DPP, lane^16 by one batched v_permlane16_swap of
// the eight registers of an accumulator tile; result valid in all four lanes
__device__ __forceinline__ void pool_xy_pk8(unsigned (&m)[8]) {
#pragma unroll
    for (int k = 0; k < 8; ++k)
        m[k] = pk_max16(m[k], (unsigned)__builtin_amdgcn_update_dpp(0, (int)m[k], 0xB1, 0xF, 0xF, true));
    unsigned b[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) b[k] = m[k];
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %8\n\tv_permlane16_swap_b32 %1, %9\n\t"
                 "v_permlane16_swap_b32 %2, %10\n\tv_permlane16_swap_b32 %3, %11\n\t"
                 "v_permlane16_swap_b32 %4, %12\n\tv_permlane16_swap_b32 %5, %13\n\t"
                 "v_permlane16_swap_b32 %6, %14\n\tv_permlane16_swap_b32 %7, %15\n\ts_nop 1"
                 : "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]), "+v"(m[4]), "+v"(m[5]), "+v"(m[6]), "+v"(m[7]),
                   "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]));
#pragma unroll
    for (int k = 0; k < 8; ++k) m[k] = pk_max16(m[k], b[k]);
}
// the same window maximum for packed fp16 pairs compared AS FLOATS (one v_pk_max_f16 per step; raw pooling of the fp16 plans)
__device__ __forceinline__ unsigned pk_fmax_f16(unsigned a, unsigned b) {
    unsigned r;
    asm volatile("v_pk_max_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ void pool_xy_pk8_f16(unsigned (&m)[8]) {
#pragma unroll
    for (int k = 0; k < 8; ++k)
        m[k] = pk_fmax_f16(m[k], (unsigned)__builtin_amdgcn_update_dpp(0, (int)m[k], 0xB1, 0xF, 0xF, true));
    unsigned b[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) b[k] = m[k];
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %8\n\tv_permlane16_swap_b32 %1, %9\n\t"
                 "v_permlane16_swap_b32 %2, %10\n\tv_permlane16_swap_b32 %3, %11\n\t"
                 "v_permlane16_swap_b32 %4, %12\n\tv_permlane16_swap_b32 %5, %13\n\t"
                 "v_permlane16_swap_b32 %6, %14\n\tv_permlane16_swap_b32 %7, %15\n\ts_nop 1"
                 : "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]), "+v"(m[4]), "+v"(m[5]), "+v"(m[6]), "+v"(m[7]),
                   "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]));
#pragma unroll
    for (int k = 0; k < 8; ++k) m[k] = pk_fmax_f16(m[k], b[k]);
}
__device__ __forceinline__ float max_xor1(float m) {       // max with lane^1 (DPP quad_perm [1,0,3,2])
    return fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0xB1, 0xF,
                                                                            0xF, true)));
}
__device__ __forceinline__ float max_xor16(float m) {      // max with lane^16, result in both lanes
    unsigned a = __builtin_bit_cast(unsigned, m), b = a;
    swap16(a, b);                                           // a = rows [0,0,2,2], b = rows [1,1,3,3]
    return fmaxf(__builtin_bit_cast(float, a), __builtin_bit_cast(float, b));
}

// 16 bytes per lane straight from global memory into LDS (lane i lands at lds_wave_base + 16*i)
__device__ __forceinline__ void glds16(const void* g, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// fp16 range guard.  fp16 storage overflows to +-inf above 65504 (bf16 and fp32 share fp32's exponent range and cannot).  An inf
// does NOT reliably reach the logits: the next convolution turns it into NaNs, and the packed integer ReLU maps a NaN with the
// sign bit set to 0 -- a whole tensor can come out as finite zeros (observed: tests/test_gpu_f32.py::test_fp16_range_guard).  So
// the guard sits where overflow is BORN: every epilogue that rounds fp32 accumulators to the storage type keeps a per-lane running
// maximum of |value| (one v_max3_f32 per two values) and raises a device flag when it reaches 65520 (the smallest magnitude
// that rounds to inf); the final-layer epilogues additionally test the softmax denominator / the logits.  The host reads
// the flag after the forward pass (sd_model_overflow).  Everything here compiles to nothing for bf16: the headline path pays nothing.
template <typename T> struct StoreGuard {
    unsigned m = 0u;       // running packed unsigned 16-bit maximum of the magnitudes stored so far
    // pk: two values already rounded to the storage type, sign bits clear (behind a ReLU)
    __device__ __forceinline__ void see(unsigned pk) {
        if constexpr (std::is_same<T, f16_t>::value) {
            typedef __attribute__((ext_vector_type(2))) unsigned short us2;
            m = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(us2, m), __builtin_bit_cast(us2, pk)));
        }
    }
    __device__ __forceinline__ void see_signed(unsigned pk) { see(pk & 0x7fff7fffu); }       // raw values of either sign
    __device__ __forceinline__ void flush(int* flag) const {      // inf / NaN = exponent field all ones = magnitude >= 0x7c00
        if constexpr (std::is_same<T, f16_t>::value) { if ((m & 0xffffu) >= 0x7c00u || (m >> 16) >= 0x7c00u) *flag = 1; }
    }
};
template <typename T>
__device__ __forceinline__ void range_guard(float v, int* flag) {
    if constexpr (std::is_same<T, f16_t>::value) {
        if (!(fabsf(v) < 3.0e38f)) *flag = 1;
    }
}
// One voxel's class logits -> what the output kind asks for (softmax with full-precision expf and a true division, floor(255 p),
// label rule) -> planar output.  The finishing step of the split-fp16 plan's final layer, fused (k_conv_mfma MODE 3) or not.
struct FinalOut { void* out; size_t out_tstride; int cout, kind; long nvox; int* ovf; };
template <typename T, typename LAB>
__device__ __forceinline__ void final_finish_exact(float (&l)[8], const FinalOut& f, const LAB& lab, int tile, size_t v) {
    float mx = -INFINITY;
#pragma unroll
    for (int co = 0; co < 8; ++co)
        if (co < f.cout) mx = fmaxf(mx, l[co]);
    if (f.kind != 0 /* SD_OUT_LOGITS_F32 */) {
        float sum = 0.f;
#pragma unroll
        for (int co = 0; co < 8; ++co) {
            l[co] = co < f.cout ? expf(l[co] - mx) : 0.f;
            sum += l[co];
        }
        if (!(fabsf(sum) < 3.0e38f)) *f.ovf = 1;
#pragma unroll
        for (int co = 0; co < 8; ++co) l[co] = l[co] / sum;
    } else {
        float c = 0.f;
#pragma unroll
        for (int co = 0; co < 8; ++co)
            if (co < f.cout) c = fmaf(l[co], 0.f, c);
        if (!(fabsf(c) < 3.0e38f)) *f.ovf = 1;
    }
    if (f.kind == 3 /* SD_OUT_LABELS_U8 */) {
        uint8_t lb = 0;
        for (int k = 0; k < lab.n; ++k) {
            const int id = lab.ids[k];
            float pv = 0.f;
#pragma unroll
            for (int co = 0; co < 8; ++co) pv = (co == id) ? l[co] : pv;
            if ((int)(uint8_t)(pv * 255.f) >= lab.cuts[k]) lb = (uint8_t)id;
        }
        (reinterpret_cast<uint8_t*>(f.out) + (size_t)tile * f.out_tstride)[v] = lb;
    } else if (f.kind == 2 /* SD_OUT_PROBS_U8 */) {
        uint8_t* out = reinterpret_cast<uint8_t*>(f.out) + (size_t)tile * f.out_tstride;
#pragma unroll
        for (int co = 0; co < 8; ++co)
            if (co < f.cout) out[(size_t)co * f.nvox + v] = (uint8_t)(l[co] * 255.f);
    } else {
        float* out = reinterpret_cast<float*>(reinterpret_cast<char*>(f.out) + (size_t)tile * f.out_tstride);
#pragma unroll
        for (int co = 0; co < 8; ++co)
            if (co < f.cout) out[(size_t)co * f.nvox + v] = l[co];
    }
}
template <typename T>
__device__ __forceinline__ float logit_probe(const float (&l)[8], int ncls) {     // 0 iff all existing logits are finite
    float c = 0.f;
    if constexpr (std::is_same<T, f16_t>::value) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k < ncls) c = fmaf(l[k], 0.f, c);
    }
    return c;
}
