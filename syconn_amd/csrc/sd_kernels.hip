// gfx950 (MI355X / CDNA4) kernels of the dense 3D-CNN prediction path.  Written for 64-wide wavefronts and the
// 32x32 MFMA shapes; there is no other code path.
//
// Layout recap (private to the library): activations are voxel-major / channel-minor, 16-channel "chunks" are the
// MFMA k-step.  Every conv is computed TRANSPOSED on the matrix core: the weight fragment is the A operand
// (rows = output channels) and the activation fragment the B operand (columns = voxels), so that in the f32
// accumulator a lane owns ONE voxel (column = lane&31) and 4 runs of 4 consecutive channels -- which are 8-byte
// contiguous pieces of the channels-last output row.
#include "sd_internal.h"
#include "../../include/syconn_dense.h"

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
};
template <> struct Act<f16_t> {
    using v8 = f16x8;
    using v4 = f16x4;
    static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};

// Hardware places workgroup b on XCD b % 8 (observed; used for L2 locality only).  Map it to a logical block id
// such that each XCD owns a contiguous run of logical ids (neighbouring blocks share halo voxels in that L2).
// Bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int b, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = b & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}

// Store one 32(channel) x 32(voxel) accumulator tile: lane owns voxel column lane&31 and channel rows
// (r&3) + 8*(r>>2) + 4*(lane>>5).
template <typename T>
__device__ __forceinline__ void store_acc_tile(const f32x16& acc, T* vox, bool valid, int nbase,
                                               const float* __restrict__ bias, int relu, int Cd) {
    using v4 = typename Act<T>::v4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int n = nbase + 8 * q;
        if (valid && n < Cd) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(bias + n);
            v4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = acc[4 * q + e] + b[e];
                if (relu) v = fmaxf(v, 0.f);
                o[e] = (T)v;
            }
            *reinterpret_cast<v4*>(vox + n) = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// K2/K3: 3x3x3 / 1x3x3 'same' convolution, C_in >= 16, as a tap-looped implicit GEMM.
//   per workgroup: 256 output voxels x (NT*32) output channels; the input halo block of one 16-channel chunk is
//   staged in LDS (48-byte voxel pitch -> conflict-free 16-byte fragment reads) and re-used by all 9 / 27 taps;
//   the weight fragments of one (chunk, kz) group (9 taps) are double-buffered in LDS.
template <typename T, int KZ, int NT>
__global__ __launch_bounds__(256, 2) void k_conv_mfma(const ConvParams p) {
    using v8 = typename Act<T>::v8;
    constexpr int BZ = sd_bz(KZ), BY = sd_by(KZ), BX = SD_BX;
    constexpr int PZ = KZ / 2;
    constexpr int HZ = BZ + KZ - 1, HY = BY + 2, HX = BX + 2;
    constexpr int NH = HZ * HY * HX;
    constexpr int VSTR = 48;
    constexpr int A_BYTES = NH * VSTR;
    constexpr int B_BYTES = 9 * NT * 1024;
    constexpr int NIT_A = (NH * 2 + 255) / 256;
    constexpr int NIT_B = (B_BYTES / 16 + 255) / 256;
    __shared__ __attribute__((aligned(16))) char smem[A_BYTES + 2 * B_BYTES];
    char* const ldsA = smem;
    char* const ldsB = smem + A_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nsb = p.nbx * p.nby * p.nbz;
    const int lb = xcd_remap(blockIdx.x, nsb);
    const int bx = lb % p.nbx, by = (lb / p.nbx) % p.nby, bz = lb / (p.nbx * p.nby);
    const int x0 = bx * BX, y0 = by * BY, z0 = bz * BZ;
    const int nb = blockIdx.y;
    const int nchunks = p.nchunk0 + p.nchunk1;
    const int ngroups = nchunks * KZ;

    // per-lane fragment read offsets of the wave's two voxel tiles
    int xoff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int t = wave * 2 + i;
        const int tz = (KZ == 3) ? (t >> 2) : 0;
        const int ty0 = (KZ == 3) ? ((t & 3) * 2) : (t * 2);
        const int vy = ty0 + ((lane & 31) >> 4), vx = lane & 15;
        xoff[i] = ((tz * HY + vy) * HX + vx) * VSTR + (lane >> 5) * 16;
    }

    f32x16 acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const char* const wbase = reinterpret_cast<const char*>(p.wpack) + (size_t)nb * ngroups * B_BYTES;

    // prologue: weight group 0 -> ldsB[0]
    {
#pragma unroll
        for (int it = 0; it < NIT_B; ++it) {
            const int o = (tid + it * 256) * 16;
            if (o < B_BYTES) *reinterpret_cast<v8*>(ldsB + o) = *reinterpret_cast<const v8*>(wbase + o);
        }
    }

    for (int c = 0; c < nchunks; ++c) {
        const char* sbase;
        int Cs, Hs, Ws, cc;
        if (c < p.nchunk0) { sbase = (const char*)p.src0; Cs = p.C0; Hs = p.H0; Ws = p.W0; cc = c; }
        else { sbase = (const char*)p.src1; Cs = p.C1; Hs = p.H1; Ws = p.W1; cc = c - p.nchunk0; }

        // global -> registers (issued before the barrier so that the latency overlaps the previous chunk's tail)
        v8 areg[NIT_A];
#pragma unroll
        for (int it = 0; it < NIT_A; ++it) {
            const int idx = tid + it * 256;
            const int v = idx >> 1, half = idx & 1;
            const int hx = v % HX, hy = (v / HX) % HY, hz = v / (HX * HY);
            const int z = z0 + hz - PZ, y = y0 + hy - 1, x = x0 + hx - 1;
            const bool ok = (idx < NH * 2) && (unsigned)z < (unsigned)p.D && (unsigned)y < (unsigned)p.H &&
                            (unsigned)x < (unsigned)p.W;
            v8 val = {};
            if (ok) {
                const size_t e = ((size_t)(z * Hs + y) * Ws + x) * Cs + cc * SD_CHUNK + half * 8;
                val = *reinterpret_cast<const v8*>(sbase + e * sizeof(T));
            }
            areg[it] = val;
        }
        __syncthreads();  // S1: every wave is done reading the previous chunk's halo block
#pragma unroll
        for (int it = 0; it < NIT_A; ++it) {
            const int idx = tid + it * 256;
            if (idx < NH * 2) *reinterpret_cast<v8*>(ldsA + (idx >> 1) * VSTR + (idx & 1) * 16) = areg[it];
        }

#pragma unroll 1
        for (int kz = 0; kz < KZ; ++kz) {
            const int g = c * KZ + kz;
            __syncthreads();  // S2: halo block + weight group g visible; group g-1's buffer is free
            // prefetch weight group g+1 into registers, written to LDS after this group's MFMAs
            v8 breg[NIT_B];
            const bool more = (g + 1 < ngroups);
            if (more) {
                const char* wsrc = wbase + (size_t)(g + 1) * B_BYTES;
#pragma unroll
                for (int it = 0; it < NIT_B; ++it) {
                    const int o = (tid + it * 256) * 16;
                    if (o < B_BYTES) breg[it] = *reinterpret_cast<const v8*>(wsrc + o);
                }
            }
            const char* const bcur = ldsB + (g & 1) * B_BYTES + lane * 16;
            const char* const acur = ldsA + kz * (HY * HX * VSTR);
#pragma unroll
            for (int t9 = 0; t9 < 9; ++t9) {
                const int tapoff = ((t9 / 3) * HX + (t9 % 3)) * VSTR;
                v8 xf[2], wf[NT];
#pragma unroll
                for (int i = 0; i < 2; ++i) xf[i] = *reinterpret_cast<const v8*>(acur + xoff[i] + tapoff);
#pragma unroll
                for (int j = 0; j < NT; ++j) wf[j] = *reinterpret_cast<const v8*>(bcur + (t9 * NT + j) * 1024);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[i][j] = Act<T>::mfma(wf[j], xf[i], acc[i][j]);
            }
            if (more) {
                char* bnext = ldsB + ((g + 1) & 1) * B_BYTES;
#pragma unroll
                for (int it = 0; it < NIT_B; ++it) {
                    const int o = (tid + it * 256) * 16;
                    if (o < B_BYTES) *reinterpret_cast<v8*>(bnext + o) = breg[it];
                }
            }
        }
    }

    // epilogue: + bias, ReLU, convert, 8-byte stores
    T* const dst = reinterpret_cast<T*>(p.dst);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int t = wave * 2 + i;
        const int tz = (KZ == 3) ? (t >> 2) : 0;
        const int ty0 = (KZ == 3) ? ((t & 3) * 2) : (t * 2);
        const int vz = z0 + tz, vy = y0 + ty0 + ((lane & 31) >> 4), vx = x0 + (lane & 15);
        const bool valid = vz < p.D && vy < p.H && vx < p.W;
        T* vox = dst + ((size_t)(vz * p.H + vy) * p.W + vx) * p.Cd;
#pragma unroll
        for (int j = 0; j < NT; ++j)
            store_acc_tile<T>(acc[i][j], vox, valid, (nb * NT + j) * 32 + 4 * (lane >> 5), p.bias, p.relu, p.Cd);
    }
}

// ---------------------------------------------------------------------------------------------------------
// K0+K2/K3 for the first layer (C_in = 1): uint8 -> float(v)/255 normalisation fused into the halo load, the
// 9 / 27 taps are the k dimension of exact-f32 32x32x2 MFMAs (bitwise an fmaf chain), weights stay in registers.
template <typename T, int KZ, typename IN>
__global__ __launch_bounds__(256) void k_conv_first(const FirstParams p) {
    constexpr int BZ = sd_bz(KZ), BY = sd_by(KZ), BX = SD_BX;
    constexpr int PZ = KZ / 2;
    constexpr int HZ = BZ + KZ - 1, HY = BY + 2, HX = BX + 2;
    constexpr int NH = HZ * HY * HX;
    constexpr int NTAP = KZ * 9;
    constexpr int NSTEP = (NTAP + 1) / 2;
    __shared__ float patch[NH];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nsb = p.nbx * p.nby * p.nbz;
    const int lb = xcd_remap(blockIdx.x, nsb);
    const int bx = lb % p.nbx, by = (lb / p.nbx) % p.nby, bz = lb / (p.nbx * p.nby);
    const int x0 = bx * BX, y0 = by * BY, z0 = bz * BZ;
    const IN* const in = reinterpret_cast<const IN*>(p.in);

    for (int i = tid; i < NH; i += 256) {
        const int hx = i % HX, hy = (i / HX) % HY, hz = i / (HX * HY);
        const int z = z0 + hz - PZ, y = y0 + hy - 1, x = x0 + hx - 1;
        float v = 0.f;
        if ((unsigned)z < (unsigned)p.D && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W) {
            const IN raw = in[((size_t)z * p.H + y) * p.W + x];
            if constexpr (sizeof(IN) == 1) v = p.lut[raw]; else v = raw;
        }
        patch[i] = v;
    }
    __syncthreads();

    // tap offsets of this lane's k index (k = 2*step + (lane>>5)); taps beyond NTAP have zero weights
    int toff[NSTEP];
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
        int tap = 2 * s + (lane >> 5);
        if (tap >= NTAP) tap = 0;
        const int kz = tap / 9, ky = (tap % 9) / 3, kx = tap % 3;
        toff[s] = (kz * HY + ky) * HX + kx;
    }
    T* const dst = reinterpret_cast<T*>(p.dst);
    const int ntiles = (p.Cd + 31) / 32;
    for (int nt = 0; nt < ntiles; ++nt) {
        float wf[NSTEP];
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) wf[s] = p.wpack[(nt * NSTEP + s) * 64 + lane];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int t = wave * 2 + i;
            const int tz = (KZ == 3) ? (t >> 2) : 0;
            const int ty0 = (KZ == 3) ? ((t & 3) * 2) : (t * 2);
            const int ly = ty0 + ((lane & 31) >> 4), lx = lane & 15;
            const int base = (tz * HY + ly) * HX + lx;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int s = 0; s < NSTEP; ++s)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[s], patch[base + toff[s]], acc, 0, 0, 0);
            const int vz = z0 + tz, vy = y0 + ly, vx = x0 + lx;
            const bool valid = vz < p.D && vy < p.H && vx < p.W;
            T* vox = dst + ((size_t)(vz * p.H + vy) * p.W + vx) * p.Cd;
            store_acc_tile<T>(acc, vox, valid, nt * 32 + 4 * (lane >> 5), p.bias, p.relu, p.Cd);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// K5: ConvTranspose3d k = s = (kz,2,2): every output voxel takes exactly one tap, so the layer is ONE dense
// GEMM  [voxels x C_in] x [C_in x (taps*C_out)]  with a scatter epilogue.  No halo -> operands straight from
// global memory (each voxel's chunk is 32 contiguous bytes; successive chunks hit the same lines in L1/L2).
template <typename T>
__global__ __launch_bounds__(256) void k_upconv_mfma(const UpconvParams p) {
    using v8 = typename Act<T>::v8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long M = (long)p.D * p.H * p.W;
    const int nb = blockIdx.y;
    const T* const src = reinterpret_cast<const T*>(p.src);
    const T* const wp = reinterpret_cast<const T*>(p.wpack) + (size_t)nb * p.nchunk * (2 * 64 * 8);

    long m[2];
    bool mv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        m[i] = (long)blockIdx.x * 256 + wave * 64 + i * 32 + (lane & 31);
        mv[i] = m[i] < M;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll 2
    for (int c = 0; c < p.nchunk; ++c) {
        v8 xf[2], wf[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            v8 val = {};
            if (mv[i]) val = *reinterpret_cast<const v8*>(src + m[i] * p.Cs + c * SD_CHUNK + (lane >> 5) * 8);
            xf[i] = val;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) wf[j] = *reinterpret_cast<const v8*>(wp + ((size_t)(c * 2 + j) * 64 + lane) * 8);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = Act<T>::mfma(wf[j], xf[i], acc[i][j]);
    }

    using v4 = typename Act<T>::v4;
    T* const dst = reinterpret_cast<T*>(p.dst);
    const int H2 = 2 * p.H, W2 = 2 * p.W;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (!mv[i]) continue;
        const int x = (int)(m[i] % p.W), y = (int)((m[i] / p.W) % p.H), z = (int)(m[i] / ((long)p.W * p.H));
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int nbase = (nb * 2 + j) * 32 + 4 * (lane >> 5);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = nbase + 8 * q;
                if (n < p.ntot) {
                    const int tap = n / p.Cd, co = n - tap * p.Cd;
                    const int a = (p.kz == 2) ? (tap >> 2) : 0, b = (tap >> 1) & 1, cx = tap & 1;
                    const f32x4 bs = *reinterpret_cast<const f32x4*>(p.bias + n);
                    v4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc[i][j][4 * q + e] + bs[e];
                        if (p.relu) v = fmaxf(v, 0.f);
                        o[e] = (T)v;
                    }
                    const size_t vo = ((size_t)(z * p.kz + a) * H2 + (2 * y + b)) * W2 + (2 * x + cx);
                    *reinterpret_cast<v4*>(dst + vo * p.Cd + co) = o;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// K4: MaxPool3d k=(kz,2,2), ceil_mode=True.  One thread per (output voxel, 8-channel group), 16-byte accesses.
template <typename T>
__global__ __launch_bounds__(256) void k_maxpool(const PoolParams p) {
    using v8 = typename Act<T>::v8;
    const int ng = p.C / 8;
    const long total = (long)p.Do * p.Ho * p.Wo * ng;
    const T* const src = reinterpret_cast<const T*>(p.src);
    T* const dst = reinterpret_cast<T*>(p.dst);
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int cg = (int)(idx % ng);
        const long v = idx / ng;
        const int xo = (int)(v % p.Wo), yo = (int)((v / p.Wo) % p.Ho), zo = (int)(v / ((long)p.Wo * p.Ho));
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
                    const v8 val = *reinterpret_cast<const v8*>(src + (((size_t)z * p.H + y) * p.W + x) * p.C + cg * 8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) mx[e] = fmaxf(mx[e], (float)val[e]);
                }
            }
        }
        v8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (T)mx[e];
        *reinterpret_cast<v8*>(dst + (size_t)v * p.C + cg * 8) = o;
    }
}

// ---------------------------------------------------------------------------------------------------------
// K8+K9+K10: final 1x1x1 conv to <= 8 classes, optional softmax over classes, optional floor(255*p) -> uint8.
// One thread per voxel; weights are wave-uniform (scalar loads).  Output planar (cout, nvox) -> coalesced stores.
template <typename T>
__global__ __launch_bounds__(256) void k_final(const FinalParams p) {
    using v8 = typename Act<T>::v8;
    const T* const src = reinterpret_cast<const T*>(p.src);
    const float* __restrict__ w = p.w;
    for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < p.nvox; v += (long)gridDim.x * 256) {
        float acc[8];
#pragma unroll
        for (int co = 0; co < 8; ++co) acc[co] = 0.f;
        for (int c8 = 0; c8 < p.Cs / 8; ++c8) {
            const v8 xv = *reinterpret_cast<const v8*>(src + (size_t)v * p.Cs + c8 * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xf = (float)xv[e];
#pragma unroll
                for (int co = 0; co < 8; ++co) acc[co] = fmaf(xf, w[co * p.Cs + c8 * 8 + e], acc[co]);
            }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int co = 0; co < 8; ++co) {
            if (co < p.cout) { acc[co] += p.bias[co]; mx = fmaxf(mx, acc[co]); }
        }
        if (p.out_kind != SD_OUT_LOGITS_F32) {
            float sum = 0.f;
#pragma unroll
            for (int co = 0; co < 8; ++co)
                if (co < p.cout) { acc[co] = expf(acc[co] - mx); sum += acc[co]; }
#pragma unroll
            for (int co = 0; co < 8; ++co) acc[co] = acc[co] / sum;
        }
        if (p.out_kind == SD_OUT_PROBS_U8) {
            uint8_t* out = reinterpret_cast<uint8_t*>(p.out);
#pragma unroll
            for (int co = 0; co < 8; ++co)
                if (co < p.cout) out[(size_t)co * p.nvox + v] = (uint8_t)(acc[co] * 255.f);
        } else {
            float* out = reinterpret_cast<float*>(p.out);
#pragma unroll
            for (int co = 0; co < 8; ++co)
                if (co < p.cout) out[(size_t)co * p.nvox + v] = acc[co];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// K7: GroupNorm (runtime statistics, not foldable).  Phase 1: per-channel sum / sum of squares (fp32 partials per
// thread, double atomics per workgroup).  Phase 2: one small block turns them into per-channel scale / shift.
// Phase 3: y = relu(x*scale + shift) in place.
template <typename T>
__global__ __launch_bounds__(192) void k_gn_stats(const GnParams p) {
    using v8 = typename Act<T>::v8;
    __shared__ float red[192][17];
    const int ng = p.C / 8;
    const int tid = threadIdx.x;
    const int cg = tid % ng, vl = tid / ng, vper = 192 / ng;
    const long nvox = (long)p.D * p.H * p.W;
    const T* const buf = reinterpret_cast<const T*>(p.buf);
    float s[8], ss[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] = 0.f; ss[e] = 0.f; }
    for (long v = (long)blockIdx.x * vper + vl; v < nvox; v += (long)gridDim.x * vper) {
        const int x = (int)(v % p.W), y = (int)((v / p.W) % p.H), z = (int)(v / ((long)p.W * p.H));
        const v8 val = *reinterpret_cast<const v8*>(buf + (((size_t)z * p.Hs + y) * p.Ws + x) * p.C + cg * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float f = (float)val[e]; s[e] += f; ss[e] = fmaf(f, f, ss[e]); }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[tid][e] = s[e]; red[tid][8 + e] = ss[e]; }
    __syncthreads();
    for (int k0 = tid; k0 < ng * 16; k0 += 192) {
        const int g = k0 / 16, e = k0 % 16;
        double t = 0.0;
        for (int k = 0; k < vper; ++k) t += (double)red[k * ng + g][e];
        const int ch = g * 8 + (e & 7);
        atomicAdd(&p.sums[(e >> 3) * p.C + ch], t);
    }
}

__global__ void k_gn_finalize(const GnParams p) {
    const int cpg = p.cout / p.groups;
    const double n = (double)p.D * p.H * p.W * cpg;
    for (int c = threadIdx.x; c < p.C; c += blockDim.x) {
        float sc = 0.f, sh = 0.f;
        if (c < p.cout) {
            const int g = c / cpg;
            double s = 0.0, ss = 0.0;
            for (int k = g * cpg; k < (g + 1) * cpg; ++k) { s += p.sums[k]; ss += p.sums[p.C + k]; }
            const double mean = s / n;
            double var = ss / n - mean * mean;
            if (var < 0.0) var = 0.0;
            const double rstd = 1.0 / sqrt(var + (double)p.eps);
            sc = (float)(rstd * (double)p.gamma[c]);
            sh = (float)((double)p.beta[c] - mean * rstd * (double)p.gamma[c]);
        }
        p.scale_shift[c] = sc;
        p.scale_shift[p.C + c] = sh;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_gn_apply(const GnParams p) {
    using v8 = typename Act<T>::v8;
    const int ng = p.C / 8;
    const long total = (long)p.D * p.H * p.W * ng;
    T* const buf = reinterpret_cast<T*>(p.buf);
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int cg = (int)(idx % ng);
        const long v = idx / ng;
        const int x = (int)(v % p.W), y = (int)((v / p.W) % p.H), z = (int)(v / ((long)p.W * p.H));
        T* ptr = buf + (((size_t)z * p.Hs + y) * p.Ws + x) * p.C + cg * 8;
        v8 val = *reinterpret_cast<const v8*>(ptr);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float f = fmaf((float)val[e], p.scale_shift[cg * 8 + e], p.scale_shift[p.C + cg * 8 + e]);
            if (p.relu) f = fmaxf(f, 0.f);
            val[e] = (T)f;
        }
        *reinterpret_cast<v8*>(ptr) = val;
    }
}

// ---------------------------------------------------------------------------------------------------------
// K1 / K12 / K11 and test support: plain element-wise kernels, x fastest -> coalesced.
template <typename E>
__global__ __launch_bounds__(256) void k_tile_gather(const E* vol, int VD, int VH, int VW, int oz, int oy, int ox,
                                                     E* tile, int TD, int TH, int TW) {
    const long total = (long)TD * TH * TW;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int x = (int)(i % TW), y = (int)((i / TW) % TH), z = (int)(i / ((long)TW * TH));
        const int vz = oz + z, vy = oy + y, vx = ox + x;
        E v = 0;
        if ((unsigned)vz < (unsigned)VD && (unsigned)vy < (unsigned)VH && (unsigned)vx < (unsigned)VW)
            v = vol[((size_t)vz * VH + vy) * VW + vx];
        tile[i] = v;
    }
}

template <typename E>
__global__ __launch_bounds__(256) void k_tile_scatter(const E* tile, int C, int TD, int TH, int TW, int cz, int cy,
                                                      int cx, int KD, int KH, int KW, E* vol, int VD, int VH, int VW,
                                                      int oz, int oy, int ox) {
    const long per = (long)KD * KH * KW;
    const long total = per * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i / per);
        const long r = i - (long)c * per;
        const int x = (int)(r % KW), y = (int)((r / KW) % KH), z = (int)(r / ((long)KW * KH));
        const int vz = oz + z, vy = oy + y, vx = ox + x;
        if (vz < VD && vy < VH && vx < VW)
            vol[(((size_t)c * VD + vz) * VH + vy) * VW + vx] =
                tile[(((size_t)c * TD + cz + z) * TH + cy + y) * TW + cx + x];
    }
}

// `cuts[i]` = smallest integer strictly greater than threshold i, so that (prob > t) <=> (prob >= cut) exactly.
template <typename O>
__global__ __launch_bounds__(256) void k_labels(const uint8_t* probs, size_t nvox, const LabelArgs a, O* out) {
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < nvox; v += (size_t)gridDim.x * 256) {
        O lab = 0;
        for (int i = 0; i < a.n; ++i) {
            const int id = a.ids[i];
            if ((int)probs[(size_t)id * nvox + v] >= a.cuts[i]) lab = (O)id;
        }
        out[v] = lab;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_read_buffer(const T* buf, int C, int Cs, long nvox, float* out) {
    const long total = nvox * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i / nvox);
        const long v = i - (long)c * nvox;
        out[i] = (float)buf[(size_t)v * Cs + c];
    }
}

// =========================================================================================================
// launchers
static inline int grid_for(long total, int per_block = 256, int cap = 256 * 16) {
    long g = (total + per_block - 1) / per_block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}
#define SD_LAUNCH_CHECK() (hipGetLastError() == hipSuccess ? SD_OK : SD_ERR_HIP)

template <typename T>
static int launch_conv_t(const ConvParams& p, int KZ, int NT, int NB, hipStream_t s) {
    dim3 grid(p.nbx * p.nby * p.nbz, NB), block(256);
    if (KZ == 3 && NT == 2) hipLaunchKernelGGL((k_conv_mfma<T, 3, 2>), grid, block, 0, s, p);
    else if (KZ == 3 && NT == 1) hipLaunchKernelGGL((k_conv_mfma<T, 3, 1>), grid, block, 0, s, p);
    else if (KZ == 1 && NT == 2) hipLaunchKernelGGL((k_conv_mfma<T, 1, 2>), grid, block, 0, s, p);
    else if (KZ == 1 && NT == 1) hipLaunchKernelGGL((k_conv_mfma<T, 1, 1>), grid, block, 0, s, p);
    else return SD_ERR_INVALID;
    return SD_LAUNCH_CHECK();
}
int launch_conv(const ConvParams& p, int act_dtype, int KZ, int NT, int NB, hipStream_t s) {
    return act_dtype == SD_BF16 ? launch_conv_t<bf16_t>(p, KZ, NT, NB, s) : launch_conv_t<f16_t>(p, KZ, NT, NB, s);
}

template <typename T, typename IN>
static int launch_first_t(const FirstParams& p, int KZ, hipStream_t s) {
    dim3 grid(p.nbx * p.nby * p.nbz), block(256);
    if (KZ == 3) hipLaunchKernelGGL((k_conv_first<T, 3, IN>), grid, block, 0, s, p);
    else if (KZ == 1) hipLaunchKernelGGL((k_conv_first<T, 1, IN>), grid, block, 0, s, p);
    else return SD_ERR_INVALID;
    return SD_LAUNCH_CHECK();
}
int launch_first(const FirstParams& p, int act_dtype, int in_dtype, int KZ, hipStream_t s) {
    if (act_dtype == SD_BF16)
        return in_dtype == SD_U8 ? launch_first_t<bf16_t, uint8_t>(p, KZ, s) : launch_first_t<bf16_t, float>(p, KZ, s);
    return in_dtype == SD_U8 ? launch_first_t<f16_t, uint8_t>(p, KZ, s) : launch_first_t<f16_t, float>(p, KZ, s);
}

int launch_upconv(const UpconvParams& p, int act_dtype, int NB, hipStream_t s) {
    const long M = (long)p.D * p.H * p.W;
    dim3 grid((unsigned)((M + 255) / 256), NB), block(256);
    if (act_dtype == SD_BF16) hipLaunchKernelGGL((k_upconv_mfma<bf16_t>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((k_upconv_mfma<f16_t>), grid, block, 0, s, p);
    return SD_LAUNCH_CHECK();
}

int launch_pool(const PoolParams& p, int act_dtype, hipStream_t s) {
    const long total = (long)p.Do * p.Ho * p.Wo * (p.C / 8);
    dim3 grid(grid_for(total)), block(256);
    if (act_dtype == SD_BF16) hipLaunchKernelGGL((k_maxpool<bf16_t>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((k_maxpool<f16_t>), grid, block, 0, s, p);
    return SD_LAUNCH_CHECK();
}

int launch_final(const FinalParams& p, int act_dtype, hipStream_t s) {
    dim3 grid(grid_for(p.nvox)), block(256);
    if (act_dtype == SD_BF16) hipLaunchKernelGGL((k_final<bf16_t>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((k_final<f16_t>), grid, block, 0, s, p);
    return SD_LAUNCH_CHECK();
}

int launch_groupnorm(const GnParams& p, int act_dtype, hipStream_t s) {
    const int ng = p.C / 8;
    if (192 % ng != 0) return SD_ERR_INVALID;
    if (hipMemsetAsync(p.sums, 0, sizeof(double) * 2 * p.C, s) != hipSuccess) return SD_ERR_HIP;
    const long nvox = (long)p.D * p.H * p.W;
    const int vper = 192 / ng;
    dim3 g1(grid_for(nvox, vper * 8, 2048)), b1(192);
    dim3 g3(grid_for(nvox * ng)), b3(256);
    if (act_dtype == SD_BF16) {
        hipLaunchKernelGGL((k_gn_stats<bf16_t>), g1, b1, 0, s, p);
        hipLaunchKernelGGL(k_gn_finalize, dim3(1), dim3(256), 0, s, p);
        hipLaunchKernelGGL((k_gn_apply<bf16_t>), g3, b3, 0, s, p);
    } else {
        hipLaunchKernelGGL((k_gn_stats<f16_t>), g1, b1, 0, s, p);
        hipLaunchKernelGGL(k_gn_finalize, dim3(1), dim3(256), 0, s, p);
        hipLaunchKernelGGL((k_gn_apply<f16_t>), g3, b3, 0, s, p);
    }
    return SD_LAUNCH_CHECK();
}

int launch_tile_gather(const void* vol, int esize, int VD, int VH, int VW, int oz, int oy, int ox, void* tile, int TD,
                       int TH, int TW, hipStream_t s) {
    dim3 grid(grid_for((long)TD * TH * TW)), block(256);
    if (esize == 1)
        hipLaunchKernelGGL((k_tile_gather<uint8_t>), grid, block, 0, s, (const uint8_t*)vol, VD, VH, VW, oz, oy, ox,
                           (uint8_t*)tile, TD, TH, TW);
    else
        hipLaunchKernelGGL((k_tile_gather<float>), grid, block, 0, s, (const float*)vol, VD, VH, VW, oz, oy, ox,
                           (float*)tile, TD, TH, TW);
    return SD_LAUNCH_CHECK();
}

int launch_tile_scatter(const void* tile, int esize, int C, int TD, int TH, int TW, int cz, int cy, int cx, int KD,
                        int KH, int KW, void* vol, int VD, int VH, int VW, int oz, int oy, int ox, hipStream_t s) {
    dim3 grid(grid_for((long)C * KD * KH * KW)), block(256);
    if (esize == 1)
        hipLaunchKernelGGL((k_tile_scatter<uint8_t>), grid, block, 0, s, (const uint8_t*)tile, C, TD, TH, TW, cz, cy,
                           cx, KD, KH, KW, (uint8_t*)vol, VD, VH, VW, oz, oy, ox);
    else
        hipLaunchKernelGGL((k_tile_scatter<float>), grid, block, 0, s, (const float*)tile, C, TD, TH, TW, cz, cy, cx,
                           KD, KH, KW, (float*)vol, VD, VH, VW, oz, oy, ox);
    return SD_LAUNCH_CHECK();
}

int launch_labels(const uint8_t* probs, size_t nvox, const LabelArgs& a, void* out, int out_u64, hipStream_t s) {
    dim3 grid(grid_for((long)nvox)), block(256);
    if (out_u64) hipLaunchKernelGGL((k_labels<uint64_t>), grid, block, 0, s, probs, nvox, a, (uint64_t*)out);
    else hipLaunchKernelGGL((k_labels<uint8_t>), grid, block, 0, s, probs, nvox, a, (uint8_t*)out);
    return SD_LAUNCH_CHECK();
}

int launch_read_buffer(const void* buf, int act_dtype, int C, int Cs, int D, int H, int W, float* out,
                       hipStream_t s) {
    const long nvox = (long)D * H * W;
    dim3 grid(grid_for(nvox * C)), block(256);
    if (act_dtype == SD_BF16)
        hipLaunchKernelGGL((k_read_buffer<bf16_t>), grid, block, 0, s, (const bf16_t*)buf, C, Cs, nvox, out);
    else
        hipLaunchKernelGGL((k_read_buffer<f16_t>), grid, block, 0, s, (const f16_t*)buf, C, Cs, nvox, out);
    return SD_LAUNCH_CHECK();
}
