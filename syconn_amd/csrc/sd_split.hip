// Split-fp16 plan (act_dtype SD_F16X2): the reference-precision plan on the matrix cores.
//
// The reference computes the U-Net in fp32 (/root/reference/syconn/handler/prediction.py:777-779: Predictor is built without
// `float16`).  fp32 has no matrix-core advantage on gfx950 (157 TFLOP/s on both pipes), fp16 has 2.5 PFLOP/s but only 11
// mantissa bits.  Here every activation and every weight is the sum of TWO fp16 numbers, x = hi + lo with hi = fp16(x),
// lo = fp16(x - hi) (22-23 mantissa bits), and a product sum is three fp16 MFMA passes with fp32 accumulation:
//     W.X  ~=  Wlo.Xhi + Whi.Xhi + Whi.Xlo          (the dropped Wlo.Xlo is 2^-22 of a term)
// A tensor of C (padded) channels is stored as 2 * C / 16 chunk planes [C/16 hi planes | C/16 lo planes] in the channel-blocked
// layout of the other plans, so the convolution is k_conv_mfma (sd_conv_mfma.h, MODE 3) walking the 3n VIRTUAL chunks
// [hi | hi | lo] of an n-chunk input against weight groups packed [lo | hi | hi].  Weights and bias carry a power of two per
// layer (2^k: largest |w| near 2^14, so every lo part is a normal fp16 number); the epilogues multiply by 2^-k (exact).
// Fused into that kernel's epilogue: MaxPool3d (on the fp32 values), the final 1x1x1 + softmax / labels (three MFMA products of
// scaled hi / lo weight fragments), GroupNorm statistics; MODE 4 computes the first convolution inside its consumer.
// This file: the launch rules of these forms and the HBM-bound helper passes on split tensors for what cannot be fused (pooling,
// final layer, GroupNorm statistics / apply [+ pooling], buffer read-back).  First convolution and up-convolutions are the SPLIT
// forms of the kernels in sd_kernels.hip.
#include "sd_conv_mfma.h"

namespace {

using T = f16_t;
using v8 = f16x8;

template <int KZ, int NT>
int launch_conv_split_knt(const ConvParams& p, int NB, hipStream_t s) {
    const long vox = (long)p.D * p.H * p.W * p.batch;
    const int nstages = (p.nchunk0 + p.nchunk1) * KZ;
    // the form rules of launch_conv_knt (sd_kernels.hip), incl. round 5's: 512-voxel workgroups from 256 of them on (deep layers stream
    // their weights once per workgroup), planar 96-column layers always as 4-wave workgroups
    const char* const big_env = getenv("SD_BIG_MIN");      // (A/B switches, read per launch)
    const long big_min = big_env ? atol(big_env) : 256;
    const bool big = (vox / 512) * NB >= big_min && !(KZ == 1 && NT == 3 && !getenv("SD_PLANAR_NT3_BIG"));
    const bool ff = p.final_wfrag != nullptr;
    static const size_t wres8 = (size_t)(getenv("SD_SPLIT_WRES8_KB") ? atoi(getenv("SD_SPLIT_WRES8_KB")) : 96) * 1024;
    if (big) {
        if (conv_lds_bytes<KZ, NT, 8, 2, 2>(nstages, ff) <= wres8) return launch_conv_k<T, KZ, NT, 8, 2, 2, 3>(p, NB, s);
        if constexpr (KZ == 3 && NT == 2) {
            const bool z_ok = getenv("SD_MT4_D_RULE") ? (p.D % 8 == 0 || p.D >= 96) : ((p.D + 7) / 8 * 8) * 100 <= ((p.D + 3) / 4 * 4) * 105;
            if (!ff && (vox / 1024) * NB >= 256 && z_ok) return launch_conv_k<T, KZ, NT, 8, 0, 4, 3>(p, NB, s);
        }
        if constexpr (KZ == 1 && NT == 2) {      // (the planar 4-tile form of launch_conv_knt)
            if (!getenv("SD_NO_PLANAR4") && !ff && !p.gn_sums && (p.H % 32 == 0 || p.H >= 128 || !getenv("SD_PLANAR4_H_RULE")) && (vox / 512) * NB >= 1024)
                return launch_conv_k<T, KZ, NT, 4, 0, 4, 3>(p, NB, s);
        }
        return launch_conv_k<T, KZ, NT, 8, 0, 2, 3>(p, NB, s);
    }
    if (conv_lds_bytes<KZ, NT, 4, 2, 2>(nstages, ff) <= 80 * 1024) return launch_conv_k<T, KZ, NT, 4, 2, 2, 3>(p, NB, s);
    if constexpr (KZ == 3 && NT == 3) {      // (the 4-wave streamed form of this shape has no room for the fused GroupNorm statistics)
        if (p.gn_sums) return launch_conv_k<T, KZ, NT, 8, 0, 2, 3>(p, NB, s);
    }
    return launch_conv_k<T, KZ, NT, 4, 0, 2, 3>(p, NB, s);
}

__device__ __forceinline__ void decode_zyx(unsigned v, unsigned W, unsigned H, int& x, int& y, int& z) {
    const unsigned r = v / W, zz = r / H;
    x = (int)(v - r * W); y = (int)(r - zz * H); z = (int)zz;
}
// 8 channels of one voxel: exact fp32 values of a (hi, lo) piece pair, and the split of 8 fp32 values
__device__ __forceinline__ void join8(v8 h, v8 l, float (&f)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = (float)h[e] + (float)l[e];
}
__device__ __forceinline__ void split8(const float (&f)[8], v8& h, v8& l) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { h[e] = (T)f[e]; l[e] = (T)(f[e] - (float)h[e]); }
}

// MaxPool3d k=(kz,2,2), ceil_mode=True on a split tensor: the maximum of the exact values, re-split (exactly the pair it came from).
__global__ __launch_bounds__(256) void k_maxpool_split(const PoolParams p) {
    const int ng = p.C / 8;
    const long total = (long)p.Do * p.Ho * p.Wo * ng;
    const T* const src = reinterpret_cast<const T*>(reinterpret_cast<const char*>(p.src) + blockIdx.z * p.tstride);
    T* const dst = reinterpret_cast<T*>(reinterpret_cast<char*>(p.dst) + blockIdx.z * p.tstride);
    const size_t slo = (size_t)(p.C >> 4) * p.Ps * SD_CHUNK, dlo = (size_t)(p.C >> 4) * p.Pd * SD_CHUNK;
    const long npv = (long)p.Do * p.Ho * p.Wo;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const unsigned pv = (unsigned)(idx >> 1), ch = pv / (unsigned)npv;  // (chunk, pooled voxel, half)
        const long v = pv - ch * (unsigned)npv;
        const int cg = (int)ch * 2 + (int)(idx & 1);
        int xo, yo, zo;
        decode_zyx((unsigned)v, (unsigned)p.Wo, (unsigned)p.Ho, xo, yo, zo);
        float mx[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) mx[e] = -INFINITY;
        for (int dz = 0; dz < p.kz; ++dz) {
            const int z = zo * p.kz + dz;
            if (z >= p.D) continue;
#pragma unroll
            for (int dy = 0; dy < 2; ++dy) {
                const int y = yo * 2 + dy;
                if (y >= p.H) continue;
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const int x = xo * 2 + dx;
                    if (x >= p.W) continue;
                    const T* const q = src + ((size_t)(cg >> 1) * p.Ps + ((size_t)z * p.H + y) * p.W + x) * SD_CHUNK + (cg & 1) * 8;
                    float f[8];
                    join8(*reinterpret_cast<const v8*>(q), *reinterpret_cast<const v8*>(q + slo), f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) mx[e] = fmaxf(mx[e], f[e]);
                }
            }
        }
        v8 h, l;
        split8(mx, h, l);
        T* const o = dst + ((size_t)(cg >> 1) * p.Pd + v) * SD_CHUNK + (cg & 1) * 8;
        *reinterpret_cast<v8*>(o) = h;
        *reinterpret_cast<v8*>(o + dlo) = l;
    }
}

// Final 1x1x1 convolution to <= 8 classes in fp32 FMA arithmetic on the exact values of the split tensor, softmax with full-
// precision expf and a true division (like the fp32 plan's k32_final), uint8 / label epilogues of the other plans.
template <int NCO>      // classes computed: 2, 4 or 8 >= p.cout (a 4-class net does not pay for 8 accumulator chains)
__global__ __launch_bounds__(256) void k_final_split(const FinalParams p) {
    const T* const src = reinterpret_cast<const T*>(reinterpret_cast<const char*>(p.src) + blockIdx.z * p.tstride);
    const size_t slo = (size_t)(p.Cs >> 4) * p.nvox * SD_CHUNK;
    const float* __restrict__ w = p.w;
    const float* const gss = p.gn_scale_shift
        ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.gn_scale_shift) + blockIdx.z * p.tstride) : nullptr;
    for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < p.nvox; v += (long)gridDim.x * 256) {
        float acc[NCO];
#pragma unroll
        for (int co = 0; co < NCO; ++co) acc[co] = 0.f;
        const int nc8 = p.Cs / 8;
        for (int g0 = 0; g0 < nc8; g0 += 4) {
            v8 xh[4], xl[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c8 = g0 + k;
                if (c8 < nc8) {
                    const T* const q = src + ((size_t)(c8 >> 1) * p.nvox + v) * SD_CHUNK + (c8 & 1) * 8;
                    xh[k] = *reinterpret_cast<const v8*>(q);
                    xl[k] = *reinterpret_cast<const v8*>(q + slo);
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c8 = g0 + k;
                if (c8 >= nc8) break;
                float f[8];
                join8(xh[k], xl[k], f);
                if (gss) {      // GroupNorm apply (+ ReLU) of a raw input tensor, fp32
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        f[e] = fmaf(f[e], gss[c8 * 8 + e], gss[p.Cs + c8 * 8 + e]);
                        if (p.gn_relu) f[e] = fmaxf(f[e], 0.f);
                    }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e)
#pragma unroll
                    for (int co = 0; co < NCO; ++co) acc[co] = fmaf(f[e], w[co * p.Cs + c8 * 8 + e], acc[co]);
            }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int co = 0; co < NCO; ++co)
            if (co < p.cout) { acc[co] += p.bias[co]; mx = fmaxf(mx, acc[co]); }
        if (p.out_kind != SD_OUT_LOGITS_F32) {
            float sum = 0.f;
#pragma unroll
            for (int co = 0; co < NCO; ++co)
                if (co < p.cout) { acc[co] = expf(acc[co] - mx); sum += acc[co]; }
            range_guard<T>(sum, p.ovf);
#pragma unroll
            for (int co = 0; co < NCO; ++co) acc[co] = acc[co] / sum;
        } else {
            float l8[8];
#pragma unroll
            for (int co = 0; co < 8; ++co) l8[co] = co < NCO ? acc[co < NCO ? co : 0] : 0.f;
            range_guard<T>(logit_probe<T>(l8, p.cout), p.ovf);
        }
        if (p.out_kind == SD_OUT_LABELS_U8) {
            uint8_t lab = 0;
            for (int k = 0; k < p.lab.n; ++k) {
                const int id = p.lab.ids[k];
                float pv = 0.f;
#pragma unroll
                for (int co = 0; co < NCO; ++co) pv = (co == id) ? acc[co] : pv;
                if ((int)(uint8_t)(pv * 255.f) >= p.lab.cuts[k]) lab = (uint8_t)id;
            }
            (reinterpret_cast<uint8_t*>(p.out) + blockIdx.z * p.out_tstride)[v] = lab;
        } else if (p.out_kind == SD_OUT_PROBS_U8) {
            uint8_t* out = reinterpret_cast<uint8_t*>(p.out) + blockIdx.z * p.out_tstride;
#pragma unroll
            for (int co = 0; co < NCO; ++co)
                if (co < p.cout) out[(size_t)co * p.nvox + v] = (uint8_t)(acc[co] * 255.f);
        } else {
            float* out = reinterpret_cast<float*>(reinterpret_cast<char*>(p.out) + blockIdx.z * p.out_tstride);
#pragma unroll
            for (int co = 0; co < NCO; ++co)
                if (co < p.cout) out[(size_t)co * p.nvox + v] = acc[co];
        }
    }
}

// GroupNorm statistics of a split tensor (one 16-channel chunk per blockIdx.y; the layout of k_gn_stats): fp32 partials of the
// exact values per thread (<= a few dozen each), double from there on.
__global__ __launch_bounds__(256) void k_gn_stats_split(const GnParams p) {
    __shared__ float red[256][33];
    const int tid = threadIdx.x, chunk = blockIdx.y;
    const long nvox = (long)p.D * p.H * p.W;
    const T* const buf = reinterpret_cast<const T*>(reinterpret_cast<const char*>(p.buf) + blockIdx.z * p.tstride) +
                         (size_t)chunk * p.P * SD_CHUNK;
    const size_t blo = (size_t)(p.C >> 4) * p.P * SD_CHUNK;
    double* const sums = reinterpret_cast<double*>(reinterpret_cast<char*>(p.sums) + blockIdx.z * p.tstride);
    float s[16], ss[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) { s[e] = 0.f; ss[e] = 0.f; }
    for (long v = (long)blockIdx.x * 256 + tid; v < nvox; v += (long)gridDim.x * 256) {
        int x, y, z;
        decode_zyx((unsigned)v, (unsigned)p.W, (unsigned)p.H, x, y, z);
        const T* q = buf + (((size_t)z * p.Hs + y) * p.Ws + x) * SD_CHUNK;
        float f[8], g[8];
        join8(*reinterpret_cast<const v8*>(q), *reinterpret_cast<const v8*>(q + blo), f);
        join8(*reinterpret_cast<const v8*>(q + 8), *reinterpret_cast<const v8*>(q + blo + 8), g);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            s[e] += f[e]; ss[e] = fmaf(f[e], f[e], ss[e]);
            s[8 + e] += g[e]; ss[8 + e] = fmaf(g[e], g[e], ss[8 + e]);
        }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) { red[tid][e] = s[e]; red[tid][16 + e] = ss[e]; }
    __syncthreads();
    __shared__ double part[8][32];
    {
        const int v = tid & 31, gseg = tid >> 5;
        double t = 0.0;
#pragma unroll 8
        for (int k = 0; k < 32; ++k) t += (double)red[gseg * 32 + k][v];
        part[gseg][v] = t;
    }
    __syncthreads();
    if (tid < 32) {
        double t = 0.0;
#pragma unroll
        for (int gseg = 0; gseg < 8; ++gseg) t += part[gseg][tid];
        atomicAdd(&sums[(tid >> 4) * p.C + chunk * SD_CHUNK + (tid & 15)], t);
    }
}

// y = relu(x * scale + shift) in fp32 on the exact values, re-split, in place
__global__ __launch_bounds__(256) void k_gn_apply_split(const GnParams p) {
    const int ng = p.C / 8;
    const long total = (long)p.D * p.H * p.W * ng;
    T* const buf = reinterpret_cast<T*>(reinterpret_cast<char*>(p.buf) + blockIdx.z * p.tstride);
    const size_t blo = (size_t)(p.C >> 4) * p.P * SD_CHUNK;
    const float* const scale_shift = reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.scale_shift) + blockIdx.z * p.tstride);
    const long nvx = (long)p.D * p.H * p.W;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const unsigned pv = (unsigned)(idx >> 1), ch = pv / (unsigned)nvx;
        const long v = pv - ch * (unsigned)nvx;
        const int cg = (int)ch * 2 + (int)(idx & 1);
        int x, y, z;
        decode_zyx((unsigned)v, (unsigned)p.W, (unsigned)p.H, x, y, z);
        T* const ptr = buf + ((size_t)(cg >> 1) * p.P + ((size_t)z * p.Hs + y) * p.Ws + x) * SD_CHUNK + (cg & 1) * 8;
        float f[8];
        join8(*reinterpret_cast<const v8*>(ptr), *reinterpret_cast<const v8*>(ptr + blo), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            f[e] = fmaf(f[e], scale_shift[cg * 8 + e], scale_shift[p.C + cg * 8 + e]);
            if (p.relu) f[e] = fmaxf(f[e], 0.f);
        }
        v8 h, l;
        split8(f, h, l);
        *reinterpret_cast<v8*>(ptr) = h;
        *reinterpret_cast<v8*>(ptr + blo) = l;
    }
}

// GroupNorm apply + ReLU with the following MaxPool3d(ceil_mode) fused: one thread per POOLED voxel and 8-channel group normalises
// the (pkz,2,2) window in place (exact values, fp32, re-split) and writes the window maximum -- the separate pooling pass would
// read the whole normalised tensor again
__global__ __launch_bounds__(256) void k_gn_apply_pool_split(const GnParams p) {
    const int ng = p.C / 8;
    const long total = (long)p.pD * p.pH * p.pW * ng;
    T* const buf = reinterpret_cast<T*>(reinterpret_cast<char*>(p.buf) + blockIdx.z * p.tstride);
    T* const pdst = reinterpret_cast<T*>(reinterpret_cast<char*>(p.pool_dst) + blockIdx.z * p.tstride);
    const size_t blo = (size_t)(p.C >> 4) * p.P * SD_CHUNK;
    const long npv = (long)p.pD * p.pH * p.pW;
    const size_t plo = (size_t)(p.C >> 4) * npv * SD_CHUNK;
    const float* const scale_shift = reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.scale_shift) + blockIdx.z * p.tstride);
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const unsigned pv = (unsigned)(idx >> 1), ch = pv / (unsigned)npv;  // (chunk, pooled voxel, half)
        const long v = pv - ch * (unsigned)npv;
        const int cg = (int)ch * 2 + (int)(idx & 1);
        int xo, yo, zo;
        decode_zyx((unsigned)v, (unsigned)p.pW, (unsigned)p.pH, xo, yo, zo);
        float sc[8], sh[8], mx[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { sc[e] = scale_shift[cg * 8 + e]; sh[e] = scale_shift[p.C + cg * 8 + e]; mx[e] = -INFINITY; }
        for (int dz = 0; dz < p.pkz; ++dz) {
            const int z = zo * p.pkz + dz;
            if (z >= p.D) continue;
#pragma unroll
            for (int dy = 0; dy < 2; ++dy) {
                const int y = yo * 2 + dy;
                if (y >= p.H) continue;
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const int x = xo * 2 + dx;
                    if (x >= p.W) continue;
                    T* const q = buf + ((size_t)(cg >> 1) * p.P + ((size_t)z * p.Hs + y) * p.Ws + x) * SD_CHUNK + (cg & 1) * 8;
                    float f[8];
                    join8(*reinterpret_cast<const v8*>(q), *reinterpret_cast<const v8*>(q + blo), f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        f[e] = fmaf(f[e], sc[e], sh[e]);
                        if (p.relu) f[e] = fmaxf(f[e], 0.f);
                        mx[e] = fmaxf(mx[e], f[e]);
                    }
                    v8 h, l;
                    split8(f, h, l);
                    *reinterpret_cast<v8*>(q) = h;
                    *reinterpret_cast<v8*>(q + blo) = l;
                }
            }
        }
        v8 h, l;
        split8(mx, h, l);
        T* const o = pdst + ((size_t)(cg >> 1) * npv + v) * SD_CHUNK + (cg & 1) * 8;
        *reinterpret_cast<v8*>(o) = h;
        *reinterpret_cast<v8*>(o + plo) = l;
    }
}

__global__ __launch_bounds__(256) void k_read_buffer_split(const T* buf, int C, int Cs, long nvox, float* out) {
    const long total = nvox * C;
    const size_t blo = (size_t)(Cs >> 4) * nvox * SD_CHUNK;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i / nvox);
        const long v = i - (long)c * nvox;
        const size_t o = ((size_t)(c >> 4) * nvox + v) * SD_CHUNK + (c & 15);
        out[i] = (float)buf[o] + (float)buf[o + blo];
    }
}

}  // namespace

// the split plan's first convolution (1 -> 32 channels, planar) can run inside its consumer (k_conv_mfma MODE 4): one 16-channel-
// pair input (six virtual chunks), NT = 1, enough blocks for persistent 512-voxel workgroups, LDS for 4 plane slots + resident weights
bool conv_can_fuse_first_split(int KZ, int NT, int NB, long vox, int nstages, bool fused_final) {
    if (KZ != 1 || NT != 1 || NB != 1 || nstages != 6) return false;
    if ((vox / 512) * NB < 512) return false;
    return conv_lds_bytes<1, 1, 8, 2, 4>(nstages, fused_final) + 2 * 36 * 20 * 4 + 352 * 4 <= (size_t)SD_LDS_BYTES;
}

int launch_conv_split(const ConvParams& p, int KZ, int NT, int NB, hipStream_t s) {
    if (p.pool_dir || p.gn0 || p.gn1) return SD_ERR_INVALID;     // fused: pooling, final layer, first convolution, GroupNorm statistics
    if (p.gn_sums && NT < 2) return SD_ERR_INVALID;              // (the one-tile forms carry no statistics code)
    if (p.first_in) {
        if (!conv_can_fuse_first_split(KZ, NT, NB, (long)p.D * p.H * p.W * p.batch, (p.nchunk0 + p.nchunk1) * KZ, p.final_wfrag != nullptr))
            return SD_ERR_INVALID;
        return launch_conv_k<T, 1, 1, 8, 4, 2, 4>(p, NB, s);
    }
    if (p.final_wfrag && KZ != 1) return SD_ERR_INVALID;                                   // ... the latter behind planar layers
    if (KZ == 3 && NT == 3) return launch_conv_split_knt<3, 3>(p, NB, s);
    if (KZ == 1 && NT == 3) return launch_conv_split_knt<1, 3>(p, NB, s);
    if (KZ == 3 && NT == 2) return launch_conv_split_knt<3, 2>(p, NB, s);
    if (KZ == 3 && NT == 1) return launch_conv_split_knt<3, 1>(p, NB, s);
    if (KZ == 1 && NT == 2) return launch_conv_split_knt<1, 2>(p, NB, s);
    if (KZ == 1 && NT == 1) return launch_conv_split_knt<1, 1>(p, NB, s);
    return SD_ERR_INVALID;
}

int launch_pool_split(const PoolParams& p, hipStream_t s) {
    const long total = (long)p.Do * p.Ho * p.Wo * (p.C / 8);
    if (total >= (1l << 32)) return SD_ERR_INVALID;       // (32-bit element decode in the kernel)
    hipLaunchKernelGGL(k_maxpool_split, dim3(grid_for(total), 1, p.batch), dim3(256), 0, s, p);
    return SD_LAUNCH_CHECK();
}

int launch_final_split(const FinalParams& p, hipStream_t s) {
    if (p.cout <= 2) hipLaunchKernelGGL(k_final_split<2>, dim3(grid_for(p.nvox), 1, p.batch), dim3(256), 0, s, p);
    else if (p.cout <= 4) hipLaunchKernelGGL(k_final_split<4>, dim3(grid_for(p.nvox), 1, p.batch), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(k_final_split<8>, dim3(grid_for(p.nvox), 1, p.batch), dim3(256), 0, s, p);
    return SD_LAUNCH_CHECK();
}

int launch_groupnorm_split(const GnParams& p, hipStream_t s) {
    const int ng = p.C / 8;
    const long nvox = (long)p.D * p.H * p.W;
    if (nvox * ng >= (1l << 32) || (p.pool_dst && (p.skip_apply || p.no_inplace))) return SD_ERR_INVALID;
    if (!p.skip_stats)      // (else: accumulated by the producing convolution's epilogue)
        hipLaunchKernelGGL(k_gn_stats_split, dim3(grid_for(nvox, 256 * 16, 1024), p.C / SD_CHUNK, p.batch), dim3(256), 0, s, p);
    const int rc = launch_gn_finalize(p, s);
    if (rc != SD_OK) return rc;
    if (p.pool_dst)         // apply + the pooling behind it in one pass
        hipLaunchKernelGGL(k_gn_apply_pool_split, dim3(grid_for((long)p.pD * p.pH * p.pW * ng), 1, p.batch), dim3(256), 0, s, p);
    else if (!p.skip_apply)      // (deferred: the only reader, the final layer, applies scale / shift itself)
        hipLaunchKernelGGL(k_gn_apply_split, dim3(grid_for(nvox * ng), 1, p.batch), dim3(256), 0, s, p);
    return SD_LAUNCH_CHECK();
}

int launch_read_buffer_split(const void* buf, int C, int Cs, long nvox, float* out, hipStream_t s) {
    hipLaunchKernelGGL(k_read_buffer_split, dim3(grid_for(nvox * C)), dim3(256), 0, s, reinterpret_cast<const T*>(buf), C, Cs, nvox, out);
    return SD_LAUNCH_CHECK();
}
