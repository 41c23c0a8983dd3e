// gfx950 (MI355X / CDNA4) kernels of the dense 3D-CNN prediction path.  Written for 64-wide wavefronts and the
// 32x32 MFMA shapes; there is no other code path.
//
// Layout recap (private to the library, sd_internal.h): activations are CHANNEL-BLOCKED [C/16][z][y][x][16]; a
// 16-channel "chunk" is one MFMA k-step and every chunk of an x-row is one contiguous run.  Every conv is computed
// TRANSPOSED on the matrix core: the weight fragment is the A operand (rows = output channels) and the activation
// fragment the B operand (columns = voxels), so that in the f32 accumulator a lane owns ONE voxel (column =
// lane&31) and 4 runs of 4 consecutive channels; the lane pair (l, l^32) trades quads so that each lane stores the
// 32 contiguous bytes of one (chunk, voxel) record.
#include "sd_internal.h"
#include "../../include/syconn_dense.h"
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <string>
#include <type_traits>
#include <utility>

#include "sd_device.h"

#include "sd_conv_mfma.h"


// ---------------------------------------------------------------------------------------------------------
// K0+K2/K3 for the first layer (C_in = 1): uint8 -> float(v)/255 normalisation fused into the halo load, the
// 9 / 27 taps are the k dimension of exact-f32 32x32x2 MFMAs (bitwise an fmaf chain), weights stay in registers.
// SPLIT (split-fp16 plan, sd_conv_mfma.h MODE 3): the fp32 result is stored as hi / lo fp16 planes (lo = Cd / 16 chunk planes further).
template <typename T, int KZ, typename IN, bool SPLIT = false>
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
    const IN* const in = reinterpret_cast<const IN*>(reinterpret_cast<const char*>(p.in) + blockIdx.z * p.in_tstride);

    for (int i = tid; i < NH; i += 256) {
        const int hx = i % HX, hy = (i / HX) % HY, hz = i / (HX * HY);
        const int z = z0 + hz - PZ, y = y0 + hy - 1, x = x0 + hx - 1;
        float v = 0.f;
        if ((unsigned)z < (unsigned)p.D && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W) {
            const IN raw = in[((size_t)z * p.H + y) * p.W + x];
            // float32(v) / 255 with IEEE division == numpy's raw.astype(np.float32) / 255. (prediction.py:808); the
            // uint8-vs-float32 input test checks the bit-equality on the device
            if constexpr (sizeof(IN) == 1) v = (float)raw / 255.0f; else v = raw;
        }
        patch[i] = v;
    }
    T* const dst = reinterpret_cast<T*>(reinterpret_cast<char*>(p.dst) + blockIdx.z * p.tstride);
    StoreGuard<T> sguard;
    const int ntiles = (p.Cd + 31) / 32;
    const int half = lane >> 5;
    const size_t P = (size_t)p.D * p.H * p.W;
    // uint8 input of a planar first convolution in the bf16 / fp16 plans: two bf16 MFMAs on the exact uint8 values with the weights / 255
    // split three ways (sd_device.h first_u8_mfma; the same arithmetic as the form fused into the next convolution, k_conv_mfma MODE 5)
    if constexpr (KZ == 1 && sizeof(IN) == 1 && !SPLIT) {
        if (p.wpack3) {
            __shared__ unsigned upatch[NH + 1];
            for (int i = tid; i < NH + 1; i += 256) {
                const int hx = i % HX, hy = i / HX;
                const int y = y0 + hy - 1, x = x0 + hx - 1;
                unsigned v = 0u;
                if (i < NH && (unsigned)z0 < (unsigned)p.D && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W)
                    v = u8_bf16_pair(in[((size_t)z0 * p.H + y) * p.W + x]);
                upatch[i] = i < NH ? v : SD_BF16_ONE_PAIR;
            }
            __syncthreads();
            for (int nt = 0; nt < ntiles; ++nt) {
                const bf16x8 w0 = reinterpret_cast<const bf16x8*>(p.wpack3)[(nt * 2 + 0) * 64 + lane];
                const bf16x8 w1 = reinterpret_cast<const bf16x8*>(p.wpack3)[(nt * 2 + 1) * 64 + lane];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int t = wave * 2 + i;
                    const int ly = t * 2 + ((lane & 31) >> 4), lx = lane & 15;
                    const int base = ly * HX + lx;
                    unsigned r[5];
#pragma unroll
                    for (int a = 0; a < 5; ++a) {
                        const int tap = half * 5 + a;
                        r[a] = upatch[tap < 9 ? base + (tap / 3) * HX + (tap % 3) : NH];
                    }
                    const f32x16 acc = first_u8_mfma(w0, w1, r);
                    const int vy = y0 + ly, vx = x0 + lx;
                    const bool valid = z0 < p.D && vy < p.H && vx < p.W;
                    unsigned pk[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        pk[k] = Act<T>::pack2(acc[2 * k], acc[2 * k + 1]);
                        if (p.relu) pk[k] = pk_max16(pk[k], 0u);
                        sguard.see_signed(pk[k]);
                    }
                    store_tile_rows_pk<T>(pk, dst, P, (size_t)(z0 * p.H + vy) * p.W + vx, valid, nt * 32, half, p.Cd);
                }
            }
            sguard.flush(p.ovf);
            return;
        }
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
    for (int nt = 0; nt < ntiles; ++nt) {
        float wf[NSTEP];
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) wf[s] = p.wpack[(nt * NSTEP + s) * 64 + lane];
        // (the folded bias rides in the last, otherwise empty k slot of the chain -- weight = bias, activation = 1.0: the last fma of
        // the chain, i.e. added AFTER the taps like the oracle does, without 16 adds per tile)
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
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[s], (s == NSTEP - 1 && half) ? 1.0f : patch[base + toff[s]], acc, 0, 0, 0);
            const int vz = z0 + tz, vy = y0 + ly, vx = x0 + lx;
            const bool valid = vz < p.D && vy < p.H && vx < p.W;
            if constexpr (SPLIT) {
                unsigned ph[8], pl[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    float a = acc[2 * k], b = acc[2 * k + 1];
                    if (p.relu) { a = fmaxf(a, 0.f); b = fmaxf(b, 0.f); }
                    split_pk(a, b, ph[k], pl[k]);
                    sguard.see_signed(ph[k]);
                }
                const size_t vo = (size_t)(vz * p.H + vy) * p.W + vx;
                store_tile_rows_pk<T>(ph, dst, P, vo, valid, nt * 32, half, p.Cd);
                store_tile_rows_pk<T>(pl, dst + (size_t)(p.Cd >> 4) * P * SD_CHUNK, P, vo, valid, nt * 32, half, p.Cd);
                continue;
            }
            // packed epilogue: one convert and one integer max per pair (relu(round(x)) == round(relu(x)))
            unsigned pk[8];
#pragma unroll
            for (int k = 0; k < 8; ++k)
                pk[k] = Act<T>::pack2(acc[2 * k], acc[2 * k + 1]);
            if (p.relu) {
#pragma unroll
                for (int k = 0; k < 8; ++k) pk[k] = pk_max16(pk[k], 0u);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) sguard.see_signed(pk[k]);
            store_tile_rows_pk<T>(pk, dst, P, (size_t)(vz * p.H + vy) * p.W + vx, valid, nt * 32, half, p.Cd);
        }
    }
    sguard.flush(p.ovf);
}

// ---------------------------------------------------------------------------------------------------------
// K5: ConvTranspose3d k = s = (kz,2,2): every output voxel takes exactly one tap, so the layer is ONE dense
// GEMM  [voxels x C_in] x [C_in x (taps*C_out)]  with a scatter epilogue.  No halo -> operands straight from
// global memory (each voxel's chunk is 32 contiguous bytes; successive chunks hit the same lines in L1/L2).
// SPLIT (split-fp16 plan): p.nchunk counts the 3n virtual chunks [hi | hi | lo] of the n-chunk input stored as planes [hi | lo],
// the weights are packed in that order (lo parts, hi parts, hi parts; times 2^k, undone by p.oscale), outputs are split again.
// voxel m of the (D,H,W) box an up-convolution launch works on -> its index inside a chunk plane of the source tensor (whose
// y / x extents are Hs, Ws: the box is the whole tensor unless sd_model_set_roi asked for a part of the output)
__device__ __forceinline__ size_t upconv_src_index(const UpconvParams& p, long m) {
    if (p.Hs == p.H && p.Ws == p.W) return (size_t)m;
    const unsigned mu = (unsigned)m, xy = mu % (unsigned)(p.W * p.H), z = mu / (unsigned)(p.W * p.H);
    const unsigned y = xy / (unsigned)p.W, x = xy - y * (unsigned)p.W;
    return ((size_t)z * p.Hs + y) * p.Ws + x;
}

template <typename T, bool GN, bool SPLIT = false>
__global__ __launch_bounds__(256) void k_upconv_mfma(const UpconvParams p) {
    using v8 = typename Act<T>::v8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long M = (long)p.D * p.H * p.W;
    const int nb = blockIdx.y;
    const T* const src = reinterpret_cast<const T*>(reinterpret_cast<const char*>(p.src) + blockIdx.z * p.tstride);
    const T* const wp = reinterpret_cast<const T*>(p.wpack) + (size_t)nb * p.nchunk * (2 * 64 * 8);

    long m[2];
    bool mv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        m[i] = (long)blockIdx.x * 256 + wave * 64 + i * 32 + (lane & 31);
        mv[i] = m[i] < M;
    }
    // accumulators start from the folded bias (padded with zeros to NB*64 entries): the summation order of every plan that
    // can serve an up-convolution (this kernel, k_upconv_rows, k_dec0) is bias first, then the chunks in order
    f32x16 acc[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 bs = *reinterpret_cast<const f32x4*>(p.bias + (nb * 2 + j) * 32 + 8 * q + 4 * (lane >> 5));
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[0][j][4 * q + e] = acc[1][j][4 * q + e] = bs[e];
        }

    __shared__ __attribute__((aligned(16))) float gtab[GN ? 2 * 1024 : 4];    // deferred GroupNorm scale / shift of this tile (Cs <= 1024)
    const float* gss = nullptr;
    if constexpr (GN) {
        const float* const g = reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.gn) + blockIdx.z * p.tstride);
        for (int i = tid; i < 2 * p.Cs; i += 256) gtab[i] = g[i];
        __syncthreads();
        gss = gtab;
    }
    for (int c = 0; c < p.nchunk; ++c) {
        v8 xf[2], wf[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            v8 val = {};
            if (mv[i]) {
                const int sc = (SPLIT && c >= p.nchunk / 3) ? c - p.nchunk / 3 : c;
                val = *reinterpret_cast<const v8*>(src + ((size_t)sc * p.Ps + upconv_src_index(p, m[i])) * SD_CHUNK + (lane >> 5) * 8);
                if constexpr (GN) val = gn_apply8<T>(val, gss + c * SD_CHUNK + (lane >> 5) * 8, gss + p.Cs + c * SD_CHUNK + (lane >> 5) * 8, p.gn_relu);
            }
            xf[i] = val;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) wf[j] = *reinterpret_cast<const v8*>(wp + ((size_t)(c * 2 + j) * 64 + lane) * 8);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = Act<T>::mfma(wf[j], xf[i], acc[i][j]);
    }

    T* const dst = reinterpret_cast<T*>(reinterpret_cast<char*>(p.dst) + blockIdx.z * p.tstride);
    StoreGuard<T> sguard;
    const int H2 = p.Hd, W2 = p.Wd;      // y / x extents of the dst tensor (2H, 2W unless the launch computes a sub-box)
    // The lane pair (l, l^32) holds channels 8q + 0..3 / 8q + 4..7 of the same voxel: trading quad q0 of the upper lane against
    // quad q1 of the lower one gives every lane 8 consecutive channels = ONE 16-byte store per quad pair instead of two
    // 8-byte ones (this generic form is bound by its scattered stores).  Swaps are executed by all lanes, stores are predicated.
    const int half = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const unsigned mu = mv[i] ? (unsigned)m[i] : 0u;      // (< 2^31 input voxels: checked by the launcher)
        const unsigned xy = mu % (unsigned)(p.W * p.H), z = mu / (unsigned)(p.W * p.H), y = xy / (unsigned)p.W, x = xy - y * (unsigned)p.W;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int nbase = (nb * 2 + j) * 32;
#pragma unroll
            for (int qp = 0; qp < 2; ++qp) {
                unsigned d[2][2], dl[2][2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int q = 2 * qp + h;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = acc[i][j][4 * q + e];
                        if constexpr (SPLIT) v[e] *= p.oscale;
                        if (p.relu) v[e] = fmaxf(v[e], 0.f);
                    }
                    if constexpr (SPLIT) {
                        split_pk(v[0], v[1], d[h][0], dl[h][0]);
                        split_pk(v[2], v[3], d[h][1], dl[h][1]);
                    } else {
                        d[h][0] = Act<T>::pack2(v[0], v[1]);
                        d[h][1] = Act<T>::pack2(v[2], v[3]);
                    }
                    sguard.see_signed(d[h][0]); sguard.see_signed(d[h][1]);
                }
                swap32(d[0][0], d[1][0]);
                swap32(d[0][1], d[1][1]);
                if constexpr (SPLIT) { swap32(dl[0][0], dl[1][0]); swap32(dl[0][1], dl[1][1]); }
                const int n8 = nbase + 8 * (2 * qp + half);       // first of this lane's 8 consecutive channels
                if (mv[i] && n8 < p.ntot) {
                    const int tap = n8 / p.Cd, co = n8 - tap * p.Cd;
                    const int a = (p.kz == 2) ? (tap >> 2) : 0, b = (tap >> 1) & 1, cx = tap & 1;
                    const size_t vo = ((size_t)(z * p.kz + a) * H2 + (2 * y + b)) * W2 + (2 * x + cx);
                    typedef __attribute__((ext_vector_type(4))) unsigned u4;
                    *reinterpret_cast<u4*>(dst + ((size_t)(co >> 4) * p.Pd + vo) * SD_CHUNK + (co & 15)) = u4{d[0][0], d[0][1], d[1][0], d[1][1]};
                    if constexpr (SPLIT)
                        *reinterpret_cast<u4*>(dst + ((size_t)((co + p.Cd) >> 4) * p.Pd + vo) * SD_CHUNK + (co & 15)) = u4{dl[0][0], dl[0][1], dl[1][0], dl[1][1]};
                }
            }
        }
    }
    sguard.flush(p.ovf);
}

// K5, row-coalescing form for the HBM-bound up-convolutions (compile-time NCH input chunks, NTAB = C_out/16 MFMA
// tiles per output row pair): one wave = 32 input voxels.  The activation fragments of all input chunks stay in
// registers (HBM read once); for each output row pair (a,b) the wave computes the 2*C_out contiguous channels of the
// two x-taps (= the two adjacent output voxels of every input voxel), rounds them, transposes them through a
// wave-private LDS tile (XOR-swizzled 16-byte pieces) and writes them out as fully coalesced 16-byte pieces: every
// input voxel yields one contiguous run of 2*C_out*sizeof(T) bytes per (a,b).
// WL: the weight fragments of the workgroup's taps live in LDS (loaded once) and the workgroup is persistent over
// voxel groups; blockIdx.y selects the (z-tap, y-tap) pair, so a workgroup needs NTAB*NCH KiB of weights.  Without
// WL every wave re-reads all its weights from L2 for each 32 voxels, which bounds the 128 -> 64 channel up-convolution.
// SPLIT (split-fp16 plan): NCH counts the REAL input chunks n; the 2n planes [hi | lo] of the input stay in registers and meet the
// 3n weight groups [lo | hi | hi] (virtual chunks [hi | hi | lo]); every output row is written twice, hi planes then lo planes.
// G > 1 (needs WL): the output channels are dealt to G workgroups per (z-tap, y-tap) pair, CD / G channels (a multiple of 32) of BOTH
// x-taps each -- blockIdx.y = pair * G + group.  A workgroup then keeps NTAB / G column tiles of weights in LDS: the split plan's
// 192 -> 96 channel shape has 36 weight groups per tile, 216 KiB for all six tiles but 72 KiB for two; with its weights left in L2
// every wave fetched 864 KiB of fragments per 32 input voxels (500 us per 128^3 tile, 9 % of semseg_axon's split plan).  The
// price is that the input planes are read by G times as many workgroups.
template <typename T, int NCH, int NTAB, bool WL, bool GN, bool SPLIT = false, int G = 1>
__global__ __launch_bounds__(256) void k_upconv_rows(const UpconvParams p) {
    constexpr int NX = SPLIT ? 2 * NCH : NCH;      // activation planes held in registers
    constexpr int NV = SPLIT ? 3 * NCH : NCH;      // weight groups per column tile
    static_assert(!(SPLIT && GN), "split plan: no deferred GroupNorm");
    static_assert(G == 1 || (WL && NTAB % (2 * G) == 0), "channel groups: LDS weights, whole tiles of each x-tap per group");
    using v8 = typename Act<T>::v8;
    using v4 = typename Act<T>::v4;
    typedef __attribute__((ext_vector_type(4))) unsigned u4;
    constexpr int CDR = NTAB * 16;           // channels of the output tensor
    constexpr int NTL = NTAB / G;            // column tiles of this workgroup
    constexpr int TPT = NTL / 2;             // ... per x-tap (G > 1)
    constexpr int CD = NTL * 16;             // channels this workgroup computes
    constexpr int ROW = 4 * CD;              // bytes per input voxel per (a,b): 2 taps * CD * 2 B
    constexpr int PPV = ROW / 16;            // 16-byte pieces per voxel
    constexpr int SWM = (PPV % 8 == 0) ? 7 : (PPV % 4 == 0) ? 3 : (PPV % 2 == 0) ? 1 : 0;   // XOR mask must divide the row
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, vl = lane & 31;
    char* const tile = smem + wave * (32 * ROW);
    const long M = (long)p.D * p.H * p.W;
    const T* const src = reinterpret_cast<const T*>(reinterpret_cast<const char*>(p.src) + blockIdx.z * p.tstride);
    const T* const wp = reinterpret_cast<const T*>(p.wpack);
    char* const dst = reinterpret_cast<char*>(p.dst) + blockIdx.z * p.tstride;
    const int H2 = p.Hd, W2 = p.Wd;      // y / x extents of the dst tensor (2H, 2W unless the launch computes a sub-box)
    StoreGuard<T> sguard;
    __shared__ __attribute__((aligned(16))) float gtab[GN ? 2 * NCH * SD_CHUNK : 4];     // deferred GroupNorm scale / shift of this tile
    const float* gss = nullptr;
    if constexpr (GN) {
        const float* const g = reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.gn) + blockIdx.z * p.tstride);
        for (int i = tid; i < 2 * NCH * SD_CHUNK; i += 256) gtab[i] = g[i];
        __syncthreads();
        gss = gtab;
    }

    // weight fragments of one (z-tap, y-tap) pair: the NTAB column tiles [ab*NTAB, ab*NTAB + NTAB) live in blocks of two
    // tiles (NCH * 2 KiB each); an odd NTAB straddles one block more (the launcher sizes the LDS for NTAB/2 + 2 blocks)
    const char* const wlds = smem + 4 * 32 * ROW;
    const int a_wg = WL ? (int)blockIdx.y / G : 0;
    const int grp = G > 1 ? (int)blockIdx.y % G : 0;
    // tile of the (pair, x-tap, channel) numbering n = tap * CDR + co that local tile jl of this workgroup is
    auto tile_of = [&](int ab, int jl) { return G == 1 ? ab * NTAB + jl : ab * NTAB + (jl / TPT) * (NTAB / 2) + grp * TPT + jl % TPT; };
    if constexpr (WL && G > 1) {
        // [local tile][weight group][64 lanes] x 16 bytes; in the packed image tile t is half (t & 1) of block t >> 1, [block][group][half]
        for (int i = tid; i < NTL * NV * 64; i += 256) {
            const int piece = i >> 6, jl = piece / NV, c = piece - jl * NV, tl = tile_of(a_wg, jl);
            *reinterpret_cast<u4*>(smem + 4 * 32 * ROW + (size_t)i * 16) =
                *reinterpret_cast<const u4*>(reinterpret_cast<const char*>(wp) + ((((size_t)(tl >> 1) * NV + c) * 2 + (tl & 1)) * 64 + (i & 63)) * 16);
        }
        __syncthreads();
    } else if constexpr (WL) {
        const int first_blk = (a_wg * NTAB) >> 1, last_blk = (a_wg * NTAB + NTAB - 1) >> 1;
        const int nbytes = (last_blk - first_blk + 1) * NV * 2048;
        const char* const wsrc = reinterpret_cast<const char*>(wp) + (size_t)first_blk * NV * 2048;
        for (int o = tid * 16; o < nbytes; o += 256 * 16)
            *reinterpret_cast<u4*>(smem + 4 * 32 * ROW + o) = *reinterpret_cast<const u4*>(wsrc + o);
        __syncthreads();
    }
    for (long m0 = ((long)blockIdx.x * 4 + wave) * 32; m0 < M; m0 += WL ? (long)gridDim.x * 128 : M) {
    const long m = m0 + vl;
    const bool mv = m < M;
    const size_t sidx = mv ? upconv_src_index(p, m) : 0;
    v8 xf[NX];
#pragma unroll
    for (int c = 0; c < NX; ++c) {
        v8 val = {};
        if (mv) val = *reinterpret_cast<const v8*>(src + ((size_t)c * p.Ps + sidx) * SD_CHUNK + half * 8);
        xf[c] = val;
    }
    if constexpr (GN) {
        // one chunk after the other (compiler barrier in between): with all table reads hoisted to the top the kernel
        // needs 18 more registers and loses a wave per SIMD, which costs this HBM-bound kernel more than the LDS latency
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (mv) xf[c] = gn_apply8<T>(xf[c], gss + c * SD_CHUNK + half * 8, gss + p.Cs + c * SD_CHUNK + half * 8, p.gn_relu);
            asm volatile("" ::: "memory");
        }
    }
    const int ab0 = WL ? a_wg : 0, nab = WL ? ab0 + 1 : p.kz * 2;
#pragma unroll 1
    for (int ab = ab0; ab < nab; ++ab) {
        // the accumulators start from the folded bias (the summation order of k_conv_mfma and of the fused level-0 decoder:
        // bias, then the input chunks in order -- the three plans must agree bit for bit)
        f32x16 acc[NTL];
#pragma unroll
        for (int j = 0; j < NTL; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + 32 * tile_of(ab, j) + 8 * q + 4 * half);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[j][4 * q + e] = b[e];
            }
#pragma unroll
        for (int c = 0; c < NV; ++c) {
#pragma unroll
            for (int j = 0; j < NTL; ++j) {
                const int tl = ab * NTAB + j;             // global 32-column tile (n = tap*CD + co ordering)
                v8 wf;
                if constexpr (WL && G > 1) {
                    wf = *reinterpret_cast<const v8*>(wlds + (((j * NV + c) * 64 + lane) * 16));
                } else if constexpr (WL) {
                    const int tloc = ab * NTAB + j - 2 * ((ab0 * NTAB) >> 1);
                    wf = *reinterpret_cast<const v8*>(wlds + ((((tloc >> 1) * NV + c) * 2 + (tloc & 1)) * 64 + lane) * 16);
                } else {
                    wf = *reinterpret_cast<const v8*>(wp + ((((size_t)(tl >> 1) * NV + c) * 2 + (tl & 1)) * 64 + lane) * 8);
                }
                acc[j] = Act<T>::mfma(wf, xf[(SPLIT && c >= NCH) ? c - NCH : c], acc[j]);
            }
        }
#pragma unroll
        for (int pass = 0; pass < (SPLIT ? 2 : 1); ++pass) {      // (split plan: hi planes, then lo planes, through the same LDS tile)
        // accumulators -> rounded rows in the wave's LDS tile: voxel vl, columns 32j + 8q + 4*half + e
#pragma unroll
        for (int j = 0; j < NTL; ++j) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                v4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[j][4 * q + e];
                    if constexpr (SPLIT) v *= p.oscale;
                    if (p.relu) v = fmaxf(v, 0.f);
                    if constexpr (SPLIT) {
                        const T h = (T)v;
                        o[e] = pass == 0 ? h : (T)(v - (float)h);
                    } else {
                        o[e] = (T)v;
                    }
                }
                if (pass == 0) {
                    typedef __attribute__((ext_vector_type(2))) unsigned u2g;
                    const u2g ob = __builtin_bit_cast(u2g, o);
                    sguard.see_signed(ob.x); sguard.see_signed(ob.y);
                }
                const int pc = (4 * j + q) ^ (vl & SWM);
                *reinterpret_cast<v4*>(tile + vl * ROW + pc * 16 + half * 8) = o;
            }
        }
        // LDS -> global, 16 bytes per lane.  Channel-blocked destination: per 16-channel chunk an input voxel yields 64
        // contiguous bytes (two x-taps x 16 channels) and consecutive input voxels of a row follow each other, so
        // the pieces are walked chunk-major: consecutive lanes = (voxel, tap, half) of ONE chunk = contiguous memory
        const int a = (p.kz == 2) ? (ab >> 1) : 0, b = ab & 1;
        // a lane handles only TWO input voxels in this loop (v = lane / 4 in even iterations, 16 + lane / 4 in odd ones):
        // their output positions are decoded once (32-bit divisions) instead of three 64-bit divisions per piece
        size_t ovv[2];
        bool okv[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const long mm = m0 + (lane >> 2) + 16 * e;
            okv[e] = mm < M;
            const unsigned mu = (unsigned)(okv[e] ? mm : 0), xy = mu % (unsigned)(p.W * p.H), z = mu / (unsigned)(p.W * p.H);
            const unsigned y = xy / (unsigned)p.W, x = xy - y * (unsigned)p.W;
            ovv[e] = ((size_t)(z * p.kz + a) * H2 + (2 * y + b)) * W2 + 2 * x;
        }
#pragma unroll
        for (int it = 0; it < (32 * PPV) / 64; ++it) {
            const int u = it * 64 + lane;
            const int chunk = u >> 7, r = u & 127;            // 128 pieces per chunk: 32 voxels x 2 taps x 2 halves
            const int v = r >> 2, tap = (r >> 1) & 1, hf = r & 1;
            const int piece = tap * (CD / 8) + chunk * 2 + hf;
            if (okv[it & 1]) {
                const size_t ov = ovv[it & 1] + tap;
                const u4 val = *reinterpret_cast<const u4*>(tile + v * ROW + ((piece ^ (v & SWM)) * 16));
                *reinterpret_cast<u4*>(dst + (((size_t)(grp * (CD / 16) + chunk + pass * (CDR / 16)) * p.Pd + ov) * SD_CHUNK + hf * 8) * sizeof(T)) = val;
            }
        }
        }
    }
    }
    sguard.flush(p.ovf);
}

// ---------------------------------------------------------------------------------------------------------
// element index -> coordinates with 32-bit arithmetic (the launchers reject tensors with >= 2^32 elements for these passes:
// three 64-bit divisions per 16-byte access made the HBM-bound pooling / GroupNorm passes VALU-bound)
__device__ __forceinline__ void decode_zyx(unsigned v, unsigned W, unsigned H, int& x, int& y, int& z) {
    const unsigned r = v / W, zz = r / H;
    x = (int)(v - r * W); y = (int)(r - zz * H); z = (int)zz;
}
// K4: MaxPool3d k=(kz,2,2), ceil_mode=True.  One thread per (output voxel, 8-channel group), 16-byte accesses.
template <typename T>
__global__ __launch_bounds__(256) void k_maxpool(const PoolParams p) {
    using v8 = typename Act<T>::v8;
    const int ng = p.C / 8;
    const long total = (long)p.Do * p.Ho * p.Wo * ng;
    const T* const src = reinterpret_cast<const T*>(reinterpret_cast<const char*>(p.src) + blockIdx.z * p.tstride);
    T* const dst = reinterpret_cast<T*>(reinterpret_cast<char*>(p.dst) + blockIdx.z * p.tstride);
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
                    const v8 val = *reinterpret_cast<const v8*>(src + ((size_t)(cg >> 1) * p.Ps + ((size_t)z * p.H + y) * p.W + x) * SD_CHUNK + (cg & 1) * 8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) mx[e] = fmaxf(mx[e], (float)val[e]);
                }
            }
        }
        v8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (T)mx[e];
        *reinterpret_cast<v8*>(dst + ((size_t)(cg >> 1) * p.Pd + v) * SD_CHUNK + (cg & 1) * 8) = o;
    }
}

// ---------------------------------------------------------------------------------------------------------
// K8+K9+K10: final 1x1x1 conv to <= 8 classes, optional softmax over classes, optional floor(255*p) -> uint8.
// One thread per voxel; weights are wave-uniform (scalar loads).  Output planar (cout, nvox) -> coalesced stores.
template <typename T>
__global__ __launch_bounds__(256) void k_final(const FinalParams p) {
    using v8 = typename Act<T>::v8;
    const T* const src = reinterpret_cast<const T*>(reinterpret_cast<const char*>(p.src) + blockIdx.z * p.tstride);
    const float* __restrict__ w = p.w;
    for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < p.nvox; v += (long)gridDim.x * 256) {
        float acc[8];
#pragma unroll
        for (int co = 0; co < 8; ++co) acc[co] = 0.f;
        const float* const gss = p.gn_scale_shift
            ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.gn_scale_shift) + blockIdx.z * p.tstride) : nullptr;
        // all 16-byte pieces of the voxel are requested before the first one is used (the layer is a pure stream: what
        // bounds it is bytes in flight per thread), in groups of up to 8 pieces = 64 channels
        const int nc8 = p.Cs / 8;
        for (int g0 = 0; g0 < nc8; g0 += 8) {
            v8 xr[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int c8 = g0 + k;
                if (c8 < nc8)
                    xr[k] = *reinterpret_cast<const v8*>(src + ((size_t)(c8 >> 1) * p.nvox + v) * SD_CHUNK + (c8 & 1) * 8);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int c8 = g0 + k;
                if (c8 >= nc8) break;
                if (gss) xr[k] = gn_apply8<T>(xr[k], gss + c8 * 8, gss + p.Cs + c8 * 8, p.gn_relu);   // deferred GroupNorm apply
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float xf = (float)xr[k][e];
#pragma unroll
                    for (int co = 0; co < 8; ++co) acc[co] = fmaf(xf, w[co * p.Cs + c8 * 8 + e], acc[co]);
                }
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
                if (co < p.cout) { acc[co] = __expf(acc[co] - mx); sum += acc[co]; }
            const float inv = 1.0f / sum;     // same exp / reciprocal form as the fused epilogue of k_conv_mfma
            range_guard<T>(sum, p.ovf);
#pragma unroll
            for (int co = 0; co < 8; ++co) acc[co] *= inv;
        } else {
            range_guard<T>(logit_probe<T>(acc, p.cout), p.ovf);
        }
        if (p.out_kind == SD_OUT_LABELS_U8) {
            uint8_t lab = 0;
            for (int k = 0; k < p.lab.n; ++k) {
                const int id = p.lab.ids[k];
                float pv = 0.f;
#pragma unroll
                for (int co = 0; co < 8; ++co) pv = (co == id) ? acc[co] : pv;
                if ((int)(uint8_t)(pv * 255.f) >= p.lab.cuts[k]) lab = (uint8_t)id;
            }
            (reinterpret_cast<uint8_t*>(p.out) + blockIdx.z * p.out_tstride)[v] = lab;
        } else if (p.out_kind == SD_OUT_PROBS_U8) {
            uint8_t* out = reinterpret_cast<uint8_t*>(p.out) + blockIdx.z * p.out_tstride;
#pragma unroll
            for (int co = 0; co < 8; ++co)
                if (co < p.cout) out[(size_t)co * p.nvox + v] = (uint8_t)(acc[co] * 255.f);
        } else {
            float* out = reinterpret_cast<float*>(reinterpret_cast<char*>(p.out) + blockIdx.z * p.out_tstride);
#pragma unroll
            for (int co = 0; co < 8; ++co)
                if (co < p.cout) out[(size_t)co * p.nvox + v] = acc[co];
        }
    }
}

// The same layer on the matrix core (the form used when the final layer cannot ride in a convolution's epilogue: GroupNorm
// networks).  logits[class][voxel] = W[class][channel] . act[channel][voxel] with the fp32 weights as hi + lo parts of
// the activation dtype (two MFMAs per 16-channel chunk, fp32-accurate products) and the activation fragment read
// straight from the channel-blocked tensor (a lane's 16 bytes = 8 channels of its voxel: one coalesced 1 KiB piece per
// wave and chunk).  A wave handles two 32-voxel tiles; the epilogue is the one of the fused final layer in k_conv_mfma.
// The scalar version above spends ~500 VALU operations per voxel (8 classes x Cs FMAs); this one ~1 MFMA per 5 voxels.
template <typename T>
__global__ __launch_bounds__(256) void k_final_mfma(const FinalParams p) {
    using v8 = typename Act<T>::v8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, vl = lane & 31;
    const int nch = p.Cs / SD_CHUNK;
    char* const wl = smem;                                               // [nch][2][64][16 B]
    float* const gl = reinterpret_cast<float*>(smem + (size_t)nch * 2048);   // scale[Cs], shift[Cs] of this tile
    float* const bl = gl + 2 * p.Cs;                                     // 8 class biases
    const T* const src = reinterpret_cast<const T*>(reinterpret_cast<const char*>(p.src) + blockIdx.z * p.tstride);
    for (int i = tid; i < nch * 128; i += 256)
        *reinterpret_cast<v8*>(wl + (size_t)i * 16) = *reinterpret_cast<const v8*>(reinterpret_cast<const T*>(p.wfrag) + (size_t)i * 8);
    if (p.gn_scale_shift) {
        const float* const gss = reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.gn_scale_shift) + blockIdx.z * p.tstride);
        for (int i = tid; i < 2 * p.Cs; i += 256) gl[i] = gss[i];
    }
    if (tid < 8) bl[tid] = tid < p.cout ? p.bias[tid] : 0.f;
    __syncthreads();
    for (long v0 = ((long)blockIdx.x * 4 + wave) * 64; v0 < p.nvox; v0 += (long)gridDim.x * 256) {
        f32x16 lg[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) lg[i][r] = 0.f;
        const long vv[2] = {v0 + vl, v0 + 32 + vl};
        for (int c = 0; c < nch; ++c) {
            v8 x[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                v8 val = {};
                if (vv[i] < p.nvox) val = *reinterpret_cast<const v8*>(src + ((size_t)c * p.nvox + vv[i]) * SD_CHUNK + half * 8);
                x[i] = val;
            }
            if (p.gn_scale_shift) {       // deferred GroupNorm apply + ReLU
                const float* const sc = gl + c * SD_CHUNK + half * 8;
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    if (vv[i] < p.nvox) x[i] = gn_apply8<T>(x[i], sc, sc + p.Cs, p.gn_relu);
            }
            const v8 w0 = *reinterpret_cast<const v8*>(wl + ((size_t)(c * 2 + 0) * 64 + lane) * 16);
            const v8 w1 = *reinterpret_cast<const v8*>(wl + ((size_t)(c * 2 + 1) * 64 + lane) * 16);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                lg[i] = Act<T>::mfma(w0, x[i], lg[i]);
                lg[i] = Act<T>::mfma(w1, x[i], lg[i]);
            }
        }
        // rows = classes: lower lanes hold classes 0-3 of their voxel in registers 0-3, upper lanes classes 4-7; one
        // half-wave swap per register gives the lower lane all 8 logits of tile 0's voxel, the upper lane tile 1's
        float l[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float bmine = bl[4 * half + e];
            unsigned a = __builtin_bit_cast(unsigned, lg[0][e] + bmine);
            unsigned b2 = __builtin_bit_cast(unsigned, lg[1][e] + bmine);
            swap32(a, b2);
            l[e] = __builtin_bit_cast(float, a);
            l[4 + e] = __builtin_bit_cast(float, b2);
        }
        float mx = -INFINITY;
#pragma unroll
        for (int co = 0; co < 8; ++co)
            if (co < p.cout) mx = fmaxf(mx, l[co]);
        float guard = 1.f;
        if (p.out_kind != SD_OUT_LOGITS_F32) {
            float sum = 0.f;
#pragma unroll
            for (int co = 0; co < 8; ++co) {
                l[co] = co < p.cout ? __expf(l[co] - mx) : 0.f;
                sum += l[co];
            }
            const float inv = 1.0f / sum;
            guard = sum;
#pragma unroll
            for (int co = 0; co < 8; ++co) l[co] *= inv;
        } else {
            guard = logit_probe<T>(l, p.cout);
        }
        const long v = vv[half];
        if (v < p.nvox) {
            range_guard<T>(guard, p.ovf);
            if (p.out_kind == SD_OUT_LABELS_U8) {
                uint8_t lab = 0;
                for (int k = 0; k < p.lab.n; ++k) {
                    const int id = p.lab.ids[k];
                    float pv = 0.f;
#pragma unroll
                    for (int co = 0; co < 8; ++co) pv = (co == id) ? l[co] : pv;
                    if ((int)(uint8_t)(pv * 255.f) >= p.lab.cuts[k]) lab = (uint8_t)id;
                }
                (reinterpret_cast<uint8_t*>(p.out) + blockIdx.z * p.out_tstride)[v] = lab;
            } else if (p.out_kind == SD_OUT_PROBS_U8) {
                uint8_t* out = reinterpret_cast<uint8_t*>(p.out) + blockIdx.z * p.out_tstride;
#pragma unroll
                for (int co = 0; co < 8; ++co)
                    if (co < p.cout) out[(size_t)co * p.nvox + v] = (uint8_t)(l[co] * 255.f);
            } else {
                float* out = reinterpret_cast<float*>(reinterpret_cast<char*>(p.out) + blockIdx.z * p.out_tstride);
#pragma unroll
                for (int co = 0; co < 8; ++co)
                    if (co < p.cout) out[(size_t)co * p.nvox + v] = l[co];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// K7: GroupNorm (runtime statistics, not foldable).  Phase 1: per-channel sum / sum of squares (fp32 partials per
// thread, double atomics per workgroup).  Phase 2: one small block turns them into per-channel scale / shift.
// Phase 3: y = relu(x*scale + shift) in place.
template <typename T>
__global__ __launch_bounds__(256) void k_gn_stats(const GnParams p) {
    // one 16-channel chunk plane per blockIdx.y: a thread reads the 32 contiguous bytes of a voxel, consecutive threads
    // consecutive voxels; fp32 partials per thread, fixed-order block reduction, one double atomic per channel
    using v8 = typename Act<T>::v8;
    __shared__ float red[256][33];
    const int tid = threadIdx.x, chunk = blockIdx.y;
    const long nvox = (long)p.D * p.H * p.W;
    const T* const buf = reinterpret_cast<const T*>(reinterpret_cast<const char*>(p.buf) + blockIdx.z * p.tstride) +
                         (size_t)chunk * p.P * SD_CHUNK;
    double* const sums = reinterpret_cast<double*>(reinterpret_cast<char*>(p.sums) + blockIdx.z * p.tstride);
    float s[16], ss[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) { s[e] = 0.f; ss[e] = 0.f; }
    for (long v = (long)blockIdx.x * 256 + tid; v < nvox; v += (long)gridDim.x * 256) {
        int x, y, z;
        decode_zyx((unsigned)v, (unsigned)p.W, (unsigned)p.H, x, y, z);
        const T* q = buf + (((size_t)z * p.Hs + y) * p.Ws + x) * SD_CHUNK;
        const v8 lo = *reinterpret_cast<const v8*>(q), hi = *reinterpret_cast<const v8*>(q + 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float f = (float)lo[e], g = (float)hi[e];
            s[e] += f; ss[e] = fmaf(f, f, ss[e]);
            s[8 + e] += g; ss[8 + e] = fmaf(g, g, ss[8 + e]);
        }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) { red[tid][e] = s[e]; red[tid][16 + e] = ss[e]; }
    __syncthreads();
    // fixed-order two-stage reduction over the 256 threads: thread (segment g, value v) sums 32 rows, then 32 threads sum the 8
    // segments (a single stage had 32 threads walk 256 rows each: as long as the streaming loop of a block itself)
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

// GroupNorm apply + ReLU with the following MaxPool3d(ceil_mode) fused: one thread per POOLED voxel and 8-channel
// group normalises the (pkz,2,2) window in place and writes the window maximum -- the separate pooling pass would
// read the whole normalised tensor again.
template <typename T>
__global__ __launch_bounds__(256) void k_gn_apply_pool(const GnParams p) {
    using v8 = typename Act<T>::v8;
    const int ng = p.C / 8;
    const long total = (long)p.pD * p.pH * p.pW * ng;
    T* const buf = reinterpret_cast<T*>(reinterpret_cast<char*>(p.buf) + blockIdx.z * p.tstride);
    T* const pdst = reinterpret_cast<T*>(reinterpret_cast<char*>(p.pool_dst) + blockIdx.z * p.tstride);
    const float* const scale_shift = reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.scale_shift) + blockIdx.z * p.tstride);
    const long npv = (long)p.pD * p.pH * p.pW;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const unsigned pv = (unsigned)(idx >> 1), ch = pv / (unsigned)npv;  // (chunk, pooled voxel, half)
        const long v = pv - ch * (unsigned)npv;
        const int cg = (int)ch * 2 + (int)(idx & 1);
        int xo, yo, zo;
        decode_zyx((unsigned)v, (unsigned)p.pW, (unsigned)p.pH, xo, yo, zo);
        const float* const sc = scale_shift + cg * 8;
        const float* const sh = scale_shift + p.C + cg * 8;
        float mx[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) mx[e] = -INFINITY;
        // all window loads are issued before the first value is used (clamped addresses, masked use): the pass is a pure stream
        // and what bounds it is bytes in flight per thread
        T* ptrs[8];
        bool okv[8];
        v8 vals[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int dz = k >> 2, dy = (k >> 1) & 1, dx = k & 1;
            const int z = zo * p.pkz + dz, y = yo * 2 + dy, x = xo * 2 + dx;
            okv[k] = dz < p.pkz && z < p.D && y < p.H && x < p.W;
            const int zc = okv[k] ? z : zo * p.pkz, yc = okv[k] ? y : yo * 2, xc = okv[k] ? x : xo * 2;
            ptrs[k] = buf + ((size_t)(cg >> 1) * p.P + ((size_t)zc * p.Hs + yc) * p.Ws + xc) * SD_CHUNK + (cg & 1) * 8;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k < 4 || p.pkz == 2) vals[k] = *reinterpret_cast<const v8*>(ptrs[k]);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (!(k < 4 || p.pkz == 2) || !okv[k]) continue;
            // (one helper for every place a GroupNorm is applied: all plans round identically)
            const v8 val = gn_apply8<T>(vals[k], sc, sh, p.relu);
#pragma unroll
            for (int e = 0; e < 8; ++e) mx[e] = fmaxf(mx[e], (float)val[e]);   // max of the ROUNDED values
            if (!p.no_inplace) *reinterpret_cast<v8*>(ptrs[k]) = val;
        }
        v8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (T)mx[e];
        *reinterpret_cast<v8*>(pdst + ((size_t)(cg >> 1) * ((size_t)p.pD * p.pH * p.pW) + v) * SD_CHUNK + (cg & 1) * 8) = o;
    }
}

__global__ void k_gn_finalize(const GnParams p) {
    double* const sums = reinterpret_cast<double*>(reinterpret_cast<char*>(p.sums) + blockIdx.z * p.tstride);
    float* const scale_shift = reinterpret_cast<float*>(reinterpret_cast<char*>(p.scale_shift) + blockIdx.z * p.tstride);
    const int cpg = p.cout / p.groups;
    const double n = (double)p.D * p.H * p.W * cpg;
    // one load per channel and statistic into LDS (the per-thread loops over a group's channels in global memory were a chain of
    // up to 192 dependent loads: 9 us per launch, 22 launches per forward of the 5-block net), the scratch is zeroed for the next
    // GroupNorm of this forward pass on the way (sd_forward_batch zeroes it once at the start)
    __shared__ double ls[2 * 1024];      // plans refuse GroupNorm over more channels (MODEL_FAIL "too many channels": C <= 2560 > 1024 handled below)
    __shared__ double gmean[256], grstd[256];
    const bool fits = p.C <= 1024 && p.groups <= 256;
    if (fits) {
        for (int c = threadIdx.x; c < 2 * p.C; c += blockDim.x) { ls[c] = sums[c]; sums[c] = 0.0; }
        __syncthreads();
        for (int g = threadIdx.x; g < p.groups; g += blockDim.x) {
            double s = 0.0, ss = 0.0;
            for (int k = g * cpg; k < (g + 1) * cpg; ++k) { s += ls[k]; ss += ls[p.C + k]; }      // same order as before
            const double mean = s / n;
            double var = ss / n - mean * mean;
            if (var < 0.0) var = 0.0;
            gmean[g] = mean;
            grstd[g] = 1.0 / sqrt(var + (double)p.eps);
        }
        __syncthreads();
        for (int c = threadIdx.x; c < p.C; c += blockDim.x) {
            float sc = 0.f, sh = 0.f;
            if (c < p.cout) {
                const int g = c / cpg;
                const double mean = gmean[g], rstd = grstd[g];
                sc = (float)(rstd * (double)p.gamma[c]);
                sh = (float)((double)p.beta[c] - mean * rstd * (double)p.gamma[c]);
            }
            scale_shift[c] = sc;
            scale_shift[p.C + c] = sh;
        }
        return;
    }
    for (int c = threadIdx.x; c < p.C; c += blockDim.x) {
        float sc = 0.f, sh = 0.f;
        if (c < p.cout) {
            const int g = c / cpg;
            double s = 0.0, ss = 0.0;
            for (int k = g * cpg; k < (g + 1) * cpg; ++k) { s += sums[k]; ss += sums[p.C + k]; }
            const double mean = s / n;
            double var = ss / n - mean * mean;
            if (var < 0.0) var = 0.0;
            const double rstd = 1.0 / sqrt(var + (double)p.eps);
            sc = (float)(rstd * (double)p.gamma[c]);
            sh = (float)((double)p.beta[c] - mean * rstd * (double)p.gamma[c]);
        }
        scale_shift[c] = sc;
        scale_shift[p.C + c] = sh;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * p.C; c += blockDim.x) sums[c] = 0.0;
}

// zero the first `nbytes` (multiple of 16) of every tile's workspace: GroupNorm statistics scratch at the start of a forward pass
__global__ void k_zero_scratch(char* ws, size_t tstride, int nbytes) {
    typedef __attribute__((ext_vector_type(4))) unsigned u4;
    u4* const q = reinterpret_cast<u4*>(ws + blockIdx.z * tstride);
    for (int i = threadIdx.x; i < nbytes / 16; i += blockDim.x) q[i] = u4{0u, 0u, 0u, 0u};
}

template <typename T>
__global__ __launch_bounds__(256) void k_gn_apply(const GnParams p) {
    using v8 = typename Act<T>::v8;
    const int ng = p.C / 8;
    const long total = (long)p.D * p.H * p.W * ng;
    T* const buf = reinterpret_cast<T*>(reinterpret_cast<char*>(p.buf) + blockIdx.z * p.tstride);
    const float* const scale_shift = reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.scale_shift) + blockIdx.z * p.tstride);
    const long nvx = (long)p.D * p.H * p.W;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        // (chunk, voxel, half): consecutive threads touch consecutive 16-byte pieces of one chunk plane
        const unsigned pv = (unsigned)(idx >> 1), ch = pv / (unsigned)nvx;
        const long v = pv - ch * (unsigned)nvx;
        const int cg = (int)ch * 2 + (int)(idx & 1);
        int x, y, z;
        decode_zyx((unsigned)v, (unsigned)p.W, (unsigned)p.H, x, y, z);
        T* ptr = buf + ((size_t)(cg >> 1) * p.P + ((size_t)z * p.Hs + y) * p.Ws + x) * SD_CHUNK + (cg & 1) * 8;
        *reinterpret_cast<v8*>(ptr) = gn_apply8<T>(*reinterpret_cast<const v8*>(ptr), scale_shift + cg * 8,
                                                     scale_shift + p.C + cg * 8, p.relu);
    }
}

// ---------------------------------------------------------------------------------------------------------
// K1 / K12 / K11 and test support: plain element-wise kernels, x fastest -> coalesced.
template <typename E>
__global__ __launch_bounds__(256) void k_tile_gather(const E* vol, int VD, int VH, int VW, int oz, int oy, int ox,
                                                     E* tile, int TD, int TH, int TW) {
    const long total = (long)TD * TH * TW;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int x, y, z;
        decode_zyx((unsigned)i, (unsigned)TW, (unsigned)TH, x, y, z);      // (tiles have < 2^32 voxels: checked by the launcher)
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
        const int c = (int)((unsigned)i / (unsigned)per);
        const long r = i - (long)c * per;
        int x, y, z;
        decode_zyx((unsigned)r, (unsigned)KW, (unsigned)KH, x, y, z);
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
        out[i] = (float)buf[((size_t)(c >> 4) * nvox + v) * SD_CHUNK + (c & 15)];
    }
}

// =========================================================================================================
// launchers

bool conv_can_fuse_first(int KZ, int NT, int NB, long vox, int nstages, bool fused_final) {
    if (KZ != 1 || NT > 2 || (nstages != 2 && !(nstages == 3 && NT == 2))) return false;
    if ((vox / 512) * NB < 512) return false;                                   // the `big` rule of launch_conv_knt
    if (nstages == 3)      // 48 filters: three resident halo slots; the resident-weight form of this layer holds one workgroup per CU already
        return getenv("SD_NO_FIRST_FUSE48") == nullptr &&
               conv_lds_bytes<1, 2, 8, 2, 3>(nstages, fused_final) + 2 * 36 * 20 * 4 + 16 + 2 * 2048 <= (size_t)SD_LDS_BYTES;
    const size_t lds = NT == 1 ? conv_lds_bytes<1, 1, 8, 2, 2>(nstages, fused_final) : conv_lds_bytes<1, 2, 8, 2, 2>(nstages, fused_final);
    return lds + 2 * 36 * 20 * 4 + 16 + 2048 <= 96 * 1024;      // (the larger of the two first-conv weight images: MODE 5)
}

template <typename T, int KZ, int NT>
static int launch_conv_knt(const ConvParams& p, int NB, hipStream_t s) {
    const long vox = (long)p.D * p.H * p.W * (p.batch_total > 0 ? p.batch_total : p.batch);      // all tiles of a batched launch set
    const int nstages = (p.nchunk0 + p.nchunk1) * KZ;
    // 512-voxel workgroups when they still give every CU work, else 256-voxel workgroups
    // (round 5: 256 instead of 512 workgroups.  The deep layers of small launch sets -- mivcsj's 768-channel layers at 32 x 8 x 8, every
    // net's level 2 / 3 on a single tile -- stream their whole weight tensor from L2 once per workgroup: 512-voxel workgroups halve that
    // traffic at the same eight waves per CU.  mivcsj 384 -> 768 / 768 -> 768: 53.5 / 104.1 -> 42.1 / 81.7 us per tile at 8 tiles; one
    // 128^3 tile: semseg_spine 0.81 -> 0.79 ms, mivcsj 3.37 -> 3.16 ms; 8-tile sets of the 32-filter nets: no layer changes form.)
    const char* const big_env = getenv("SD_BIG_MIN");      // (A/B switch, read per launch)
    const long big_min = big_env ? atol(big_env) : 256;
    // (planar layers with 96-column workgroups -- levels 2+ of the 48-filter family -- run faster as 4-wave workgroups at every size
    // measured: mivcsj's GroupNorm layers at 32 x 16 x 16 20.5 / 31.0 / 58.1 / 30.7 -> 15.1 / 17.3 / 33.4 / 17.6 us per tile, semseg_axon's at
    // 64 x 32 x 32 18.7 / 35.2 -> 17.4 / 32.3; SD_PLANAR_NT3_BIG=1 restores the 8-wave forms)
    const bool big = (vox / 512) * NB >= big_min && !(KZ == 1 && NT == 3 && !getenv("SD_PLANAR_NT3_BIG"));
    // Resident weights + persistent blocks where the whole layer's weights fit beside a 2-slot halo ring and two
    // workgroups still share a CU (level-0 layers); else streamed weights, one block per workgroup.  Deeper rings
    // (NSLOT 4/6, one workgroup per CU) were measured SLOWER on the level-0 layers (1.32 vs 1.23 ms per tile): those
    // layers are bound by per-wave instruction latency, not by bytes in flight, so resident waves win over ring depth.
    // (NT = 1 layers with 4 voxel tiles per wave -- 8 waves x 1024 voxels or 4 waves x 512 voxels -- measured SLOWER than the
    // 2-tile form, op19 54 -> 59 us per tile: these layers live on resident waves per CU, not on LDS reads per MFMA)
    if constexpr (KZ == 1 && NT <= 2) {
        if (p.first_in) {
            if (!conv_can_fuse_first(KZ, NT, NB, vox, nstages, p.final_wfrag != nullptr)) return SD_ERR_INVALID;
            if (p.first_w3 && !p.first_in_f32) {      // uint8 input: the first convolution on the bf16 pipe (MODE 5)
                if constexpr (NT == 2) {
                    if (nstages == 3) return launch_conv_k<T, KZ, NT, 8, 3, 2, 5>(p, NB, s);
                }
                return launch_conv_k<T, KZ, NT, 8, 2, 2, 5>(p, NB, s);
            }
            if constexpr (NT == 2) {
                if (nstages == 3) return launch_conv_k<T, KZ, NT, 8, 3, 2, 1>(p, NB, s);
            }
            return launch_conv_k<T, KZ, NT, 8, 2, 2, 1>(p, NB, s);
        }
    } else if (p.first_in) {
        return SD_ERR_INVALID;
    }
    // (one 16-wave workgroup per CU -- 1x64x16 blocks, 3-slot halo ring = a whole block of prefetch -- instead of two 8-wave
    // workgroups with one chunk of prefetch each: op19 52.7 -> 53.1 us.  Cycle stamps of that layer, per 512-voxel block of
    // 10.9 k cycles: DMA issue 2 x 0.86 k (24 halo gathers per chunk serialise in the CU's address path at ~36 cycles each),
    // tap loops 2 x 0.9 k, DMA wait + barrier ~2 k, fused final / softmax / store epilogue 4.5 k -- no single bottleneck.)
    if (p.gn0 || p.gn1) {      // deferred GroupNorm apply: same shape rules, MODE 2 kernels (LDS incl. the scale / shift table)
        const size_t gl = (size_t)p.batch * conv_gn_lds_per_tile(p.nchunk0 + p.nchunk1);
        if (big) {
            // (full-resolution layers: resident weights as far as the LDS reaches, like the plain form below -- mivcsj's level-0 convs
            // 194.7 / 273.8 / 163.9 -> 186.9 / 263.4 / 155.2 us per tile.  SD_WRES_GN_CAP_KB: A/B switch, read per launch)
            const char* const gcap_env = getenv("SD_WRES_GN_CAP_KB");
            const size_t gn_cap = (size_t)(gcap_env ? atoi(gcap_env) : ((long)p.D * p.H * p.W >= (1l << 20) ? 158 : 80)) * 1024;
            if (conv_lds_bytes<KZ, NT, 8, 2, 2>(nstages, p.final_wfrag != nullptr) + gl <= gn_cap) return launch_conv_k<T, KZ, NT, 8, 2, 2, 2>(p, NB, s);
            return launch_conv_k<T, KZ, NT, 8, 0, 2, 2>(p, NB, s);
        }
        if (conv_lds_bytes<KZ, NT, 4, 2, 2>(nstages, p.final_wfrag != nullptr) + gl <= 80 * 1024) return launch_conv_k<T, KZ, NT, 4, 2, 2, 2>(p, NB, s);
        return launch_conv_k<T, KZ, NT, 4, 0, 2, 2>(p, NB, s);
    }
    if (big) {
        // (full-resolution layers, round 5: resident weights up to 150 KiB -- one workgroup per CU either way from 80 KiB on.  The 48-filter
        // family's level-0 decoder convs: 96 -> 48 187 -> 180 us, 48 -> 48 + final 141 -> 130 us per tile; the 32-filter nets' level-2 layers,
        // which would fit as well, are no faster that way (9.7 -> 10.0 us) and keep the 96 KiB rule.  SD_WRES_CAP_KB: A/B, read per launch)
        const char* const cap_env = getenv("SD_WRES_CAP_KB");
        const size_t wres_cap = (size_t)(cap_env ? atoi(cap_env) : ((long)p.D * p.H * p.W >= (1l << 20) ? 150 : 96)) * 1024;
        if (conv_lds_bytes<KZ, NT, 8, 2, 2>(nstages, p.final_wfrag != nullptr) <= wres_cap) return launch_conv_k<T, KZ, NT, 8, 2>(p, NB, s);
        if constexpr (KZ == 3 && NT == 2) {
            // 8x8x16 blocks, 4 z-stacked voxel tiles per wave: 0.75 LDS fragment reads per MFMA instead of 1.0, half as many
            // stage barriers and block boundaries per MFMA (32->64 channels 52.6 -> 48.7 us per tile, 64->64 88 -> 85 us,
            // 128->64 164 -> 162 us).  Taller blocks waste more on a ragged z extent, hence the rule on D.
            const bool mt2 = getenv("SD_MT2") != nullptr;      // A/B switch (read per launch): the 4x8x16 / 2-tile form everywhere
            // (the z extent: 8-plane blocks may cover at most 5 % more planes than 4-plane blocks would -- 89 planes: 96 vs 92, the
            // reference tile's level 3, 517 -> 506 us per tile; rounds 2-4 asked for D % 8 == 0 || D >= 96: SD_MT4_D_RULE)
            const bool z_ok = getenv("SD_MT4_D_RULE") ? (p.D % 8 == 0 || p.D >= 96) : ((p.D + 7) / 8 * 8) * 100 <= ((p.D + 3) / 4 * 4) * 105;
            if (!mt2 && !p.final_wfrag && (vox / 1024) * NB >= 256 && z_ok)
                return launch_conv_k<T, KZ, NT, 8, 0, 4>(p, NB, s);
        }
        if constexpr (KZ == 1 && NT == 2) {
            // planar 64 / 128-filter layers with streamed weights (levels 2+ of the 32-filter family): 4 y-stacked voxel tiles per
            // wave, 4-wave workgroups of 1 x 32 x 16 voxels, two per CU -- 0.75 instead of 1.0 LDS fragment reads per MFMA (at 1.0
            // the four SIMDs of a CU ask for 128 B per cycle: all the LDS has) and a barrier per 72 instead of 36 MFMAs
            const bool no_p4 = getenv("SD_NO_PLANAR4") != nullptr;      // A/B switch (read per launch)
            // (rounds 2-4 asked for H % 32 == 0 || H >= 128; the 8-wave form this falls back to has 32-row blocks as well, so a ragged
            // height costs both the same: at 89 x 61 x 83 -- the reference tile's level 2 -- the four layers 631 -> 600 us per tile.
            // SD_PLANAR4_H_RULE restores the rule)
            if (!no_p4 && !p.final_wfrag && (p.H % 32 == 0 || p.H >= 128 || !getenv("SD_PLANAR4_H_RULE")) && (vox / 512) * NB >= 1024)
                return launch_conv_k<T, KZ, NT, 4, 0, 4>(p, NB, s);
        }
        if constexpr (KZ == 1 && NT == 3) {
            // (measured: 48-filter family -6 % on these layers; the NT = 2 layers are LDS-bandwidth bound -- 1.0 fragment reads per
            // MFMA -- and do not move)
            // planar layers with streamed weights: 3-deep rings for halo chunks and weight groups, DMA groups issued under the
            // SIMD partner's MFMAs (see RING in k_conv_mfma)
            const bool no_ring = getenv("SD_NO_RING") != nullptr;       // A/B switch (read per launch)
            if (!no_ring && nstages >= 3 && conv_lds_bytes<KZ, NT, 8, 2, -3>(nstages, p.final_wfrag != nullptr) <= (size_t)SD_LDS_BYTES)
                return launch_conv_k<T, KZ, NT, 8, -3>(p, NB, s);
        }
        return launch_conv_k<T, KZ, NT, 8, 0>(p, NB, s);
    }
    if (conv_lds_bytes<KZ, NT, 4, 2, 2>(nstages, p.final_wfrag != nullptr) <= 80 * 1024) return launch_conv_k<T, KZ, NT, 4, 2>(p, NB, s);
    return launch_conv_k<T, KZ, NT, 4, 0>(p, NB, s);
}
template <typename T>
static int launch_conv2_t(const ConvParams& p, int KZ, int NT, int NB, hipStream_t s) {
    if (KZ == 3 && NT == 3) return launch_conv_knt<T, 3, 3>(p, NB, s);
    if (KZ == 1 && NT == 3) return launch_conv_knt<T, 1, 3>(p, NB, s);
    if (KZ == 3 && NT == 2) return launch_conv_knt<T, 3, 2>(p, NB, s);
    if (KZ == 3 && NT == 1) return launch_conv_knt<T, 3, 1>(p, NB, s);
    if (KZ == 1 && NT == 2) return launch_conv_knt<T, 1, 2>(p, NB, s);
    if (KZ == 1 && NT == 1) return launch_conv_knt<T, 1, 1>(p, NB, s);
    return SD_ERR_INVALID;
}
int launch_conv(const ConvParams& p, int act_dtype, int KZ, int NT, int NB, hipStream_t s) {
    if (act_dtype == SD_F16X2) return launch_conv_split(p, KZ, NT, NB, s);
    return act_dtype == SD_BF16 ? launch_conv2_t<bf16_t>(p, KZ, NT, NB, s) : launch_conv2_t<f16_t>(p, KZ, NT, NB, s);
}

template <typename T, typename IN>
static int launch_first_t(const FirstParams& p, int KZ, hipStream_t s) {
    SD_NOTE_KERNEL(sizeof(IN) == 1 ? (p.wpack3 && KZ == 1 ? "k_conv_first<uint8 input, bf16 MFMA on the exact values>" : "k_conv_first<uint8 input, f32 MFMA chain>")
                                   : "k_conv_first<float32 input, f32 MFMA chain>");
    dim3 grid(p.nbx * p.nby * p.nbz, 1, p.batch), block(256);
    if (KZ == 3) hipLaunchKernelGGL((k_conv_first<T, 3, IN>), grid, block, 0, s, p);
    else if (KZ == 1) hipLaunchKernelGGL((k_conv_first<T, 1, IN>), grid, block, 0, s, p);
    else return SD_ERR_INVALID;
    return SD_LAUNCH_CHECK();
}
int launch_first(const FirstParams& p, int act_dtype, int in_dtype, int KZ, hipStream_t s) {
    if (act_dtype == SD_F16X2) {
        SD_NOTE_KERNEL("k_conv_first<split-fp16, f32 MFMA chain>");
        dim3 grid(p.nbx * p.nby * p.nbz, 1, p.batch), block(256);
        if (KZ == 3 && in_dtype == SD_U8) hipLaunchKernelGGL((k_conv_first<f16_t, 3, uint8_t, true>), grid, block, 0, s, p);
        else if (KZ == 3) hipLaunchKernelGGL((k_conv_first<f16_t, 3, float, true>), grid, block, 0, s, p);
        else if (KZ == 1 && in_dtype == SD_U8) hipLaunchKernelGGL((k_conv_first<f16_t, 1, uint8_t, true>), grid, block, 0, s, p);
        else if (KZ == 1) hipLaunchKernelGGL((k_conv_first<f16_t, 1, float, true>), grid, block, 0, s, p);
        else return SD_ERR_INVALID;
        return SD_LAUNCH_CHECK();
    }
    if (act_dtype == SD_BF16)
        return in_dtype == SD_U8 ? launch_first_t<bf16_t, uint8_t>(p, KZ, s) : launch_first_t<bf16_t, float>(p, KZ, s);
    return in_dtype == SD_U8 ? launch_first_t<f16_t, uint8_t>(p, KZ, s) : launch_first_t<f16_t, float>(p, KZ, s);
}

template <typename T, int NCH, int NTAB>
static int launch_upconv_rows(const UpconvParams& p, hipStream_t s) {
    { static const std::string name = "k_upconv_rows<NCH=" + std::to_string(NCH) + ",NTAB=" + std::to_string(NTAB) + ">"; SD_NOTE_KERNEL(name.c_str()); }
    const long M = (long)p.D * p.H * p.W;
    if (M >= (1l << 31)) return SD_ERR_INVALID;        // the kernel decodes input voxel indices with 32-bit arithmetic
    dim3 grid((unsigned)((M + 127) / 128), 1, p.batch), block(256);
    if (p.gn) hipLaunchKernelGGL((k_upconv_rows<T, NCH, NTAB, false, true>), grid, block, 4 * 32 * 64 * NTAB, s, p);
    else hipLaunchKernelGGL((k_upconv_rows<T, NCH, NTAB, false, false>), grid, block, 4 * 32 * 64 * NTAB, s, p);
    return SD_LAUNCH_CHECK();
}
template <typename T, int NCH, int NTAB>
static int launch_upconv_rows_wl(const UpconvParams& p, hipStream_t s) {      // LDS-resident weights, persistent
    { static const std::string name = "k_upconv_rows<NCH=" + std::to_string(NCH) + ",NTAB=" + std::to_string(NTAB) + ",LDS weights>"; SD_NOTE_KERNEL(name.c_str()); }
    const long M = (long)p.D * p.H * p.W;
    if (M >= (1l << 31)) return SD_ERR_INVALID;
    const size_t lds = 4 * 32 * 64 * NTAB + (size_t)(NTAB / 2 + 2 * (NTAB & 1)) * NCH * 2048;
    auto kern = p.gn ? k_upconv_rows<T, NCH, NTAB, true, true> : k_upconv_rows<T, NCH, NTAB, true, false>;
    {
        static std::mutex mu;
        static LaunchCache cache[2][SD_MAX_DEVICES];          // [plain / deferred-GroupNorm kernel][device]
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= SD_MAX_DEVICES) return SD_ERR_HIP;
        std::lock_guard<std::mutex> lock(mu);
        LaunchCache& c = cache[p.gn ? 1 : 0][dev];
        if (lds > c.attr_set) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
                hipSuccess) return SD_ERR_HIP;
            c.attr_set = lds;
        }
    }
    const int per_cu = std::max(1, (int)(SD_LDS_BYTES / lds));
    const long want = std::max(1L, (long)SD_NUM_CU * per_cu * 2 / (2 * p.kz * p.batch));     // ~2 rounds of workgroups
    dim3 grid((unsigned)std::min((M + 127) / 128, want), 2 * p.kz, p.batch), block(256);
    hipLaunchKernelGGL(kern, grid, block, lds, s, p);
    return SD_LAUNCH_CHECK();
}
// LDS-resident weights with G channel groups per tap pair (k_upconv_rows<G>): NTAB / G column tiles of weights per workgroup
template <typename T, int NCH, int NTAB, int G>
static int launch_upconv_rows_wl_g(const UpconvParams& p, hipStream_t s) {
    { static const std::string name = "k_upconv_rows<NCH=" + std::to_string(NCH) + ",NTAB=" + std::to_string(NTAB) + ",LDS weights,G=" + std::to_string(G) + ">"; SD_NOTE_KERNEL(name.c_str()); }
    const long M = (long)p.D * p.H * p.W;
    if (M >= (1l << 31) || p.gn) return SD_ERR_INVALID;
    constexpr int NTL = NTAB / G;
    const size_t lds = (size_t)4 * 32 * (4 * NTL * 16) + (size_t)NTL * NCH * 1024;
    auto kern = k_upconv_rows<T, NCH, NTAB, true, false, false, G>;
    if (lds > (size_t)SD_LDS_BYTES) return SD_ERR_INVALID;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return SD_ERR_HIP;
    const int per_cu = std::max(1, (int)(SD_LDS_BYTES / lds));
    const long want = std::max(1L, (long)SD_NUM_CU * per_cu * 2 / (2 * p.kz * G * p.batch));
    dim3 grid((unsigned)std::min((M + 127) / 128, want), 2 * p.kz * G, p.batch), block(256);
    hipLaunchKernelGGL(kern, grid, block, lds, s, p);
    return SD_LAUNCH_CHECK();
}
// true when launch_upconv runs this shape with the row-coalescing kernel (activations of all chunks in registers): the form
// into which a deferred GroupNorm apply folds for less than the separate apply pass costs (the generic MFMA kernel pays
// more for it than the pass it replaces: 192 -> 96 channels 80 -> 108 us vs a 10 us pass)
bool upconv_rows_kernel(int nchunk, int Cd) {
    static const bool no_wl = getenv("SD_NO_UPCONV_WL") != nullptr;
    return (nchunk == 4 && Cd == 32) || (nchunk == 3 && Cd == 32) || (nchunk == 2 && Cd == 16) || (nchunk == 8 && Cd == 64 && !no_wl) ||
           (nchunk == 6 && Cd == 48) || (nchunk == 12 && Cd == 96 && !no_wl);
}

template <typename T>
static int launch_upconv_t(const UpconvParams& p, int NB, hipStream_t s) {
    SD_NOTE_KERNEL("k_upconv_mfma");      // (the row-kernel launchers below overwrite it)
    // the store-bound full-resolution shapes get the row-coalescing kernel (64 -> 32 channels: 77 -> 52 us at 128^3);
    // at 128 -> 64 channels it needs its weights in LDS (plain rows kernel 85 us, k_upconv_mfma 63 us, LDS weights with
    // one tap pair per workgroup and two workgroups per CU 51 us; 48 -> 33 us per tile at 8 tiles per launch)
    static const bool no_wl = getenv("SD_NO_UPCONV_WL") != nullptr;
    if (p.nchunk == 4 && p.Cd == 32) return (no_wl || getenv("SD_UPCONV32_NO_WL")) ? launch_upconv_rows<T, 4, 2>(p, s) : launch_upconv_rows_wl<T, 4, 2>(p, s);
    if (p.nchunk == 3 && p.Cd == 32) return launch_upconv_rows<T, 3, 2>(p, s);
    if (p.nchunk == 2 && p.Cd == 16) return launch_upconv_rows<T, 2, 1>(p, s);
    if (p.nchunk == 8 && p.Cd == 64 && !no_wl) return launch_upconv_rows_wl<T, 8, 4>(p, s);
    // (192 -> 96 channels: k_upconv_mfma 76 us, LDS-weight rows kernel 83 us per tile in the channel-blocked layout)
    if (p.nchunk == 6 && p.Cd == 48) return no_wl ? launch_upconv_rows<T, 6, 3>(p, s) : launch_upconv_rows_wl<T, 6, 3>(p, s);
    // 192 -> 96 channels: the generic kernel re-reads the input for each of its 12 column blocks (PMC: 1.0 GB read for 0.2 GB
    // algorithmic per 8 tiles) and writes partial lines; rows kernel with LDS-resident weights 70 -> 60 us per tile
    if (p.nchunk == 12 && p.Cd == 96 && !no_wl) return launch_upconv_rows_wl<T, 12, 6>(p, s);
    // 256 -> 128 channels: rows kernel, weights in LDS, two channel groups per tap pair (64 KiB of weights each).  The generic kernel
    // slows down on shapes that are not powers of two: 89 x 31 x 42 -> 89 x 62 x 84 (the reference tile) 129.9 -> 77.6 us per tile, at
    // 64 x 16 x 16 -> 64 x 32 x 32 12.4 -> 11.5 us (SD_UPCONV128_MFMA: the generic kernel, A/B)
    // (a workgroup of these forms loads 48-64 KiB of weights: they pay from ~6 tiles of 128^3 per launch set on -- 256 -> 128 on ONE such
    // tile 21.0 -> 33.2 us, on two 16.1 -> 18.9 per tile: small launch sets keep the generic kernel)
    const bool many = (long)p.D * p.H * p.W * p.batch >= 98304;
    if (p.nchunk == 16 && p.Cd == 128 && !p.gn && !no_wl && many && !getenv("SD_UPCONV128_MFMA")) return launch_upconv_rows_wl_g<T, 16, 8, 2>(p, s);
    // 384 -> 192 channels (48-filter BatchNorm nets): six channel groups of 32 (48 KiB of weights each): 26.8 -> 17.3 us per 128^3 tile
    // (three groups of 64: 22.3; SD_UPCONV192_MFMA: the generic kernel, A/B)
    if (p.nchunk == 24 && p.Cd == 192 && !p.gn && !no_wl && many && !getenv("SD_UPCONV192_MFMA")) return launch_upconv_rows_wl_g<T, 24, 12, 6>(p, s);
    const long M = (long)p.D * p.H * p.W;
    if (M >= (1l << 31)) return SD_ERR_INVALID;        // (32-bit voxel decode in the kernel)
    dim3 grid((unsigned)((M + 255) / 256), NB, p.batch), block(256);
    if (p.gn) hipLaunchKernelGGL((k_upconv_mfma<T, true>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((k_upconv_mfma<T, false>), grid, block, 0, s, p);
    return SD_LAUNCH_CHECK();
}
int launch_upconv(const UpconvParams& p, int act_dtype, int NB, hipStream_t s) {
    if (act_dtype == SD_F16X2) {      // split-fp16 plan: row kernel for the full-resolution shapes, else the generic kernel (3n virtual chunks)
        SD_NOTE_KERNEL(p.nchunk == 12 || p.nchunk == 24 || p.nchunk == 18 || p.nchunk == 36 || p.nchunk == 48 ? "k_upconv_rows<split-fp16> / k_upconv_mfma<split-fp16>" : "k_upconv_mfma<split-fp16>");
        const long M = (long)p.D * p.H * p.W;
        if (M >= (1l << 31) || p.gn) return SD_ERR_INVALID;
        static const bool no_rows = getenv("SD_SPLIT_NO_ROWS") != nullptr;      // A/B switch
        if (!no_rows && p.nchunk == 12 && p.Cd == 32 && getenv("SD_SPLIT_ROWS32_NO_WL")) {       // 64 -> 32 channels (level 0), weights from L2
            dim3 grid((unsigned)((M + 127) / 128), 1, p.batch), block(256);
            hipLaunchKernelGGL((k_upconv_rows<f16_t, 4, 2, false, false, true>), grid, block, 4 * 32 * 64 * 2, s, p);
            return SD_LAUNCH_CHECK();
        }
        // the other row-kernel shapes: weights of one (z-tap, y-tap) pair in LDS, persistent (the forms of launch_upconv_rows_wl)
        auto rows_wl = [&](auto kern, int NTAB, int NV) -> int {
            const size_t lds = (size_t)4 * 32 * 64 * NTAB + (size_t)(NTAB / 2 + 2 * (NTAB & 1)) * NV * 2048;
            if (lds > (size_t)SD_LDS_BYTES) return SD_ERR_INVALID;
            // (one attribute call per launch: the shapes are few and the call is cheap next to a launch of this size)
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                return SD_ERR_HIP;
            const int per_cu = std::max(1, (int)(SD_LDS_BYTES / lds));
            const long want = std::max(1L, (long)SD_NUM_CU * per_cu * 2 / (2 * p.kz * p.batch));     // ~2 rounds of workgroups
            dim3 grid((unsigned)std::min((M + 127) / 128, want), 2 * p.kz, p.batch), block(256);
            hipLaunchKernelGGL(kern, grid, block, lds, s, p);
            return SD_LAUNCH_CHECK();
        };
        // 64 -> 32 channels (level 0) with the weights in LDS as well: without them every wave fetched 24 KiB of weight fragments from L2
        // per 32 input voxels and y-tap, three times the 8 KiB it writes (SD_SPLIT_ROWS32_NO_WL: the round-4 form, A/B)
        if (!no_rows && p.nchunk == 12 && p.Cd == 32 && !getenv("SD_SPLIT_ROWS32_NO_WL"))
            return rows_wl(k_upconv_rows<f16_t, 4, 2, true, false, true>, 2, 12);
        if (!no_rows && p.nchunk == 48 && p.Cd == 128 && M * p.batch >= 98304 && !getenv("SD_SPLIT_UPCONV128_MFMA")) {      // 256 -> 128: four channel groups of 32 (96 KiB of weights each)
            auto kern = k_upconv_rows<f16_t, 16, 8, true, false, true, 4>;
            const size_t lds = (size_t)4 * 32 * 128 + (size_t)2 * 48 * 1024;
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                return SD_ERR_HIP;
            const long want = std::max(1L, (long)SD_NUM_CU * 2 / (2 * p.kz * 4 * p.batch));
            dim3 grid((unsigned)std::min((M + 127) / 128, want), 2 * p.kz * 4, p.batch), block(256);
            hipLaunchKernelGGL(kern, grid, block, lds, s, p);
            return SD_LAUNCH_CHECK();
        }
        if (!no_rows && p.nchunk == 24 && p.Cd == 64) return rows_wl(k_upconv_rows<f16_t, 8, 4, true, false, true>, 4, 24);     // 128 -> 64
        if (!no_rows && p.nchunk == 18 && p.Cd == 48) return rows_wl(k_upconv_rows<f16_t, 6, 3, true, false, true>, 3, 18);     // 96 -> 48
        // 192 -> 96 channels: the 36 weight groups of all six column tiles do not fit the LDS; three workgroups per tap pair with the
        // two tiles of 32 channels each (k_upconv_rows<G = 3>: 72 KiB of weights + 16 KiB of transpose tiles)
        if (!no_rows && p.nchunk == 36 && p.Cd == 96 && !getenv("SD_SPLIT_NO_ROWS96") && !getenv("SD_SPLIT_ROWS96_L2")) {
            auto kern = k_upconv_rows<f16_t, 12, 6, true, false, true, 3>;
            const size_t lds = (size_t)4 * 32 * 128 + (size_t)2 * 36 * 1024;
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                return SD_ERR_HIP;
            const int per_cu = std::max(1, (int)(SD_LDS_BYTES / lds));
            const long want = std::max(1L, (long)SD_NUM_CU * per_cu * 2 / (2 * p.kz * 3 * p.batch));
            dim3 grid((unsigned)std::min((M + 127) / 128, want), 2 * p.kz * 3, p.batch), block(256);
            hipLaunchKernelGGL(kern, grid, block, lds, s, p);
            return SD_LAUNCH_CHECK();
        }
        // (SD_SPLIT_ROWS96_L2: the round-4 form of that shape, weights from L2 -- A/B)
        if (!no_rows && p.nchunk == 36 && p.Cd == 96 && !getenv("SD_SPLIT_NO_ROWS96")) {
            auto kern = k_upconv_rows<f16_t, 12, 6, false, false, true>;
            const size_t lds = (size_t)4 * 32 * 64 * 6;
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                return SD_ERR_HIP;
            dim3 grid((unsigned)((M + 127) / 128), 1, p.batch), block(256);
            hipLaunchKernelGGL(kern, grid, block, lds, s, p);
            return SD_LAUNCH_CHECK();
        }
        dim3 grid((unsigned)((M + 255) / 256), NB, p.batch), block(256);
        hipLaunchKernelGGL((k_upconv_mfma<f16_t, false, true>), grid, block, 0, s, p);
        return SD_LAUNCH_CHECK();
    }
    if (!getenv("SD_UPCONV_OLD") || p.gn)
        return act_dtype == SD_BF16 ? launch_upconv_t<bf16_t>(p, NB, s) : launch_upconv_t<f16_t>(p, NB, s);
    const long M = (long)p.D * p.H * p.W;
    dim3 grid((unsigned)((M + 255) / 256), NB, p.batch), block(256);
    if (act_dtype == SD_BF16) hipLaunchKernelGGL((k_upconv_mfma<bf16_t, false>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((k_upconv_mfma<f16_t, false>), grid, block, 0, s, p);
    return SD_LAUNCH_CHECK();
}

int launch_pool(const PoolParams& p, int act_dtype, hipStream_t s) {
    SD_NOTE_KERNEL(act_dtype == SD_F16X2 ? "k_maxpool_split" : "k_maxpool");
    if (act_dtype == SD_F16X2) return launch_pool_split(p, s);
    const long total = (long)p.Do * p.Ho * p.Wo * (p.C / 8);
    if (total >= (1l << 32)) return SD_ERR_INVALID;       // (32-bit element decode in the kernel)
    dim3 grid(grid_for(total), 1, p.batch), block(256);
    if (act_dtype == SD_BF16) hipLaunchKernelGGL((k_maxpool<bf16_t>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((k_maxpool<f16_t>), grid, block, 0, s, p);
    return SD_LAUNCH_CHECK();
}

int launch_final(const FinalParams& p, int act_dtype, hipStream_t s) {
    SD_NOTE_KERNEL(act_dtype == SD_F16X2 ? "k_final_split" : "k_final_mfma / k_final");
    if (act_dtype == SD_F16X2) return launch_final_split(p, s);
    static const bool scalar_final = getenv("SD_FINAL_SCALAR") != nullptr;     // debugging aid: the FMA-chain version
    const size_t lds = (size_t)(p.Cs / SD_CHUNK) * 2048 + (size_t)(2 * p.Cs + 8) * 4;
    if (p.wfrag && !scalar_final && lds <= 64 * 1024) {
        dim3 grid(grid_for(p.nvox, 256, 256 * 8), 1, p.batch), block(256);
        if (act_dtype == SD_BF16) hipLaunchKernelGGL((k_final_mfma<bf16_t>), grid, block, lds, s, p);
        else hipLaunchKernelGGL((k_final_mfma<f16_t>), grid, block, lds, s, p);
        return SD_LAUNCH_CHECK();
    }
    dim3 grid(grid_for(p.nvox), 1, p.batch), block(256);
    if (act_dtype == SD_BF16) hipLaunchKernelGGL((k_final<bf16_t>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((k_final<f16_t>), grid, block, 0, s, p);
    return SD_LAUNCH_CHECK();
}

int launch_gn_finalize(const GnParams& p, hipStream_t s) {
    hipLaunchKernelGGL(k_gn_finalize, dim3(1, 1, p.batch), dim3(256), 0, s, p);
    return SD_LAUNCH_CHECK();
}

int launch_groupnorm(const GnParams& p, int act_dtype, hipStream_t s) {
    SD_NOTE_KERNEL(p.skip_stats ? (p.skip_apply ? "k_gn_finalize" : "k_gn_finalize + k_gn_apply") : (p.skip_apply ? "k_gn_stats + k_gn_finalize" : "k_gn_stats + k_gn_finalize + k_gn_apply"));
    if (act_dtype == SD_F16X2) return launch_groupnorm_split(p, s);
    const int ng = p.C / 8;
    // (the statistics scratch is zero here: zeroed at the start of the forward pass and again by every k_gn_finalize)
    const long nvox = (long)p.D * p.H * p.W;
    if (nvox * ng >= (1l << 32)) return SD_ERR_INVALID;   // (32-bit element decode in the kernels)
    dim3 g1(grid_for(nvox, 256 * 16, 1024), p.C / SD_CHUNK, p.batch), b1(256);
    dim3 g3(grid_for(nvox * ng), 1, p.batch), b3(256);
    dim3 gp(grid_for((long)std::max(p.pD, 1) * std::max(p.pH, 1) * std::max(p.pW, 1) * ng), 1, p.batch);
    if (act_dtype == SD_BF16) {
        if (!p.skip_stats) hipLaunchKernelGGL((k_gn_stats<bf16_t>), g1, b1, 0, s, p);
        hipLaunchKernelGGL(k_gn_finalize, dim3(1, 1, p.batch), dim3(256), 0, s, p);
        if (p.skip_apply) {}
        else if (p.pool_dst) hipLaunchKernelGGL((k_gn_apply_pool<bf16_t>), gp, b3, 0, s, p);
        else hipLaunchKernelGGL((k_gn_apply<bf16_t>), g3, b3, 0, s, p);
    } else {
        if (!p.skip_stats) hipLaunchKernelGGL((k_gn_stats<f16_t>), g1, b1, 0, s, p);
        hipLaunchKernelGGL(k_gn_finalize, dim3(1, 1, p.batch), dim3(256), 0, s, p);
        if (p.skip_apply) {}
        else if (p.pool_dst) hipLaunchKernelGGL((k_gn_apply_pool<f16_t>), gp, b3, 0, s, p);
        else hipLaunchKernelGGL((k_gn_apply<f16_t>), g3, b3, 0, s, p);
    }
    return SD_LAUNCH_CHECK();
}

int launch_zero_scratch(void* ws, size_t tstride, int nbytes, int batch, hipStream_t s) {
    hipLaunchKernelGGL(k_zero_scratch, dim3(1, 1, batch), dim3(256), 0, s, reinterpret_cast<char*>(ws), tstride, nbytes);
    return SD_LAUNCH_CHECK();
}

int launch_tile_gather(const void* vol, int esize, int VD, int VH, int VW, int oz, int oy, int ox, void* tile, int TD,
                       int TH, int TW, hipStream_t s) {
    if ((long)TD * TH * TW >= (1l << 32)) return SD_ERR_INVALID;           // (32-bit element decode in the kernel)
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
    if ((long)C * KD * KH * KW >= (1l << 32)) return SD_ERR_INVALID;      // (32-bit element decode in the kernel)
    dim3 grid(grid_for((long)C * KD * KH * KW)), block(256);
    if (esize == 1)
        hipLaunchKernelGGL((k_tile_scatter<uint8_t>), grid, block, 0, s, (const uint8_t*)tile, C, TD, TH, TW, cz, cy,
                           cx, KD, KH, KW, (uint8_t*)vol, VD, VH, VW, oz, oy, ox);
    else
        hipLaunchKernelGGL((k_tile_scatter<float>), grid, block, 0, s, (const float*)tile, C, TD, TH, TW, cz, cy, cx,
                           KD, KH, KW, (float*)vol, VD, VH, VW, oz, oy, ox);
    return SD_LAUNCH_CHECK();
}

// Order-0 down-sampling by 2 (one mag-pyramid level of KnossosDataset.save_raw/save_seg, fast_resampling=True):
// dst[z,y,x] = src[2z,2y,2x].  HBM-bound strided pick; one thread per output voxel, x fastest.
template <typename E>
__global__ __launch_bounds__(256) void k_downsample2(const E* src, int H, int W, E* dst, int Do, int Ho, int Wo) {
    const long total = (long)Do * Ho * Wo;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int x, y, z;
        decode_zyx((unsigned)i, (unsigned)Wo, (unsigned)Ho, x, y, z);       // (< 2^32 output voxels: checked by the launcher)
        dst[i] = src[((size_t)(2 * z) * H + 2 * y) * W + 2 * x];
    }
}
int launch_downsample2(const void* src, int esize, int D, int H, int W, void* dst, hipStream_t s) {
    const int Do = (D + 1) / 2, Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    if ((long)Do * Ho * Wo >= (1l << 32)) return SD_ERR_INVALID;
    dim3 grid(grid_for((long)Do * Ho * Wo)), block(256);
    if (esize == 1) hipLaunchKernelGGL((k_downsample2<uint8_t>), grid, block, 0, s, (const uint8_t*)src, H, W, (uint8_t*)dst, Do, Ho, Wo);
    else if (esize == 8) hipLaunchKernelGGL((k_downsample2<uint64_t>), grid, block, 0, s, (const uint64_t*)src, H, W, (uint64_t*)dst, Do, Ho, Wo);
    else return SD_ERR_INVALID;
    return SD_LAUNCH_CHECK();
}

// Box majority vote (map_myelin2coords, /root/reference/syconn/reps/super_segmentation_helper.py:550-615): for every
// box origin (z,y,x; may lie partly outside the volume = zeros) count the voxels >= cut inside an (ez,ey,ex) box and
// emit (double)count / n_box > thresh_majority.  One wave per box; lanes stride over the box voxels, x fastest.
__global__ __launch_bounds__(256) void k_box_majority(const uint8_t* vol, int D, int H, int W, const int32_t* origins,
                                                      long n, int ez, int ey, int ex, int cut, double thresh_majority,
                                                      uint8_t* out) {
    const int lane = threadIdx.x & 63;
    const long box = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (box >= n) return;
    const int oz = origins[3 * box], oy = origins[3 * box + 1], ox = origins[3 * box + 2];
    const int nvox = ez * ey * ex;
    int cnt = 0;
    for (int i = lane; i < nvox; i += 64) {
        const int x = ox + i % ex, y = oy + (i / ex) % ey, z = oz + i / (ex * ey);
        if ((unsigned)z < (unsigned)D && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W)
            cnt += (int)vol[((size_t)z * H + y) * W + x] >= cut;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) cnt += __shfl_xor(cnt, m, 64);
    if (lane == 0) out[box] = ((double)cnt / (double)nvox > thresh_majority) ? 1 : 0;
}
int launch_box_majority(const uint8_t* vol, int D, int H, int W, const int32_t* origins, long n, int ez, int ey, int ex,
                        int cut, double thresh_majority, uint8_t* out, hipStream_t s) {
    if (n <= 0) return SD_OK;
    dim3 grid((unsigned)((n + 3) / 4)), block(256);
    hipLaunchKernelGGL(k_box_majority, grid, block, 0, s, vol, D, H, W, origins, n, ez, ey, ex, cut, thresh_majority, out);
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
    if (act_dtype == SD_F16X2) return launch_read_buffer_split(buf, C, Cs, nvox, out, s);
    if (act_dtype == SD_BF16)
        hipLaunchKernelGGL((k_read_buffer<bf16_t>), grid, block, 0, s, (const bf16_t*)buf, C, Cs, nvox, out);
    else
        hipLaunchKernelGGL((k_read_buffer<f16_t>), grid, block, 0, s, (const f16_t*)buf, C, Cs, nvox, out);
    return SD_LAUNCH_CHECK();
}
