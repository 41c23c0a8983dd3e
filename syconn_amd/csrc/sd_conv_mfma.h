// k_conv_mfma (the tap-looped implicit-GEMM convolution of the bf16 / fp16 plans) and its launcher, shared by the translation
// units that instantiate it: sd_kernels.hip (plain / fused-first / deferred-GroupNorm forms) and sd_split.hip (split-fp16 form).
#pragma once
#include "sd_internal.h"
#include "../../include/syconn_dense.h"
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <string>
#include <type_traits>
#include <utility>

#include "sd_device.h"


// Store the 32 channels x 32 voxels of one accumulator tile whose values are already packed as 4 x (4 channels):
// o[q] = channels cbase + 4*(lane>>5) + 8q + 0..3 of voxel (lane&31).  The lane pair (l, l^32) first trades quads
// (0 <-> 1 and 2 <-> 3) so that the LOWER lane owns channels 0-7 and 16-23 and the UPPER lane channels 8-15 and 24-31 of
// their voxel: the first store then writes the complete 32-byte records of 16-channel chunk cbase/16 -- lower lanes the
// first 16 bytes, upper lanes the second -- i.e. two fully covered 512-byte row runs per instruction, the second store
// the same for the next chunk.  (Before: each lane owned one whole 32-byte record and wrote it as two 16-byte pieces, so
// every store instruction half-filled 64 sectors; the epilogue of a block is bound by the CU's address path.)
// Must be called by all 64 lanes (stores are predicated, swaps are not).
template <typename T>
__device__ __forceinline__ void store_tile_rows(typename Act<T>::v4 (&o)[4], T* base, size_t P, size_t vidx, bool valid,
                                                int cbase, int half, int Cd) {
    typedef __attribute__((ext_vector_type(2))) unsigned u2;
    typedef __attribute__((ext_vector_type(4))) unsigned u4;
    u2 a0 = __builtin_bit_cast(u2, o[0]), a1 = __builtin_bit_cast(u2, o[1]);
    u2 a2 = __builtin_bit_cast(u2, o[2]), a3 = __builtin_bit_cast(u2, o[3]);
    unsigned x;
    x = a0.x; { unsigned y = a1.x; swap32(x, y); a0.x = x; a1.x = y; }
    x = a0.y; { unsigned y = a1.y; swap32(x, y); a0.y = x; a1.y = y; }
    x = a2.x; { unsigned y = a3.x; swap32(x, y); a2.x = x; a3.x = y; }
    x = a2.y; { unsigned y = a3.y; swap32(x, y); a2.y = x; a3.y = y; }
    if (valid) {
        T* const q = base + ((size_t)(cbase >> 4) * P + vidx) * SD_CHUNK + half * 8;
        if (cbase < Cd) { u4 v = {a0.x, a0.y, a1.x, a1.y}; *reinterpret_cast<u4*>(q) = v; }
        if (cbase + 16 < Cd) { u4 v = {a2.x, a2.y, a3.x, a3.y}; *reinterpret_cast<u4*>(q + P * SD_CHUNK) = v; }
    }
}

// Same, from the packed form pk[2q + h] = channels (cbase + 8q + 4*(lane>>5) + 2h, +1) of voxel (lane&31).
template <typename T>
__device__ __forceinline__ void store_tile_rows_pk(const unsigned (&pk)[8], T* base, size_t P, size_t vidx, bool valid,
                                                   int cbase, int half, int Cd) {
    typedef __attribute__((ext_vector_type(4))) unsigned u4;
    unsigned a0x = pk[0], a0y = pk[1], a1x = pk[2], a1y = pk[3], a2x = pk[4], a2y = pk[5], a3x = pk[6], a3y = pk[7];
    swap32x4(a0x, a1x, a0y, a1y, a2x, a3x, a2y, a3y);
    if (valid) {
        T* const q = base + ((size_t)(cbase >> 4) * P + vidx) * SD_CHUNK + half * 8;
        if (cbase < Cd) { u4 v = {a0x, a0y, a1x, a1y}; *reinterpret_cast<u4*>(q) = v; }
        if (cbase + 16 < Cd) { u4 v = {a2x, a2y, a3x, a3y}; *reinterpret_cast<u4*>(q + P * SD_CHUNK) = v; }
    }
}

// Same with the address arithmetic hoisted by the caller: q = this lane's 16-byte half of its voxel's record in the first
// chunk, cstride = elements between the chunk planes, c0 / c1 = chunk exists (wave-uniform).
template <typename T>
__device__ __forceinline__ void store_tile_rows_pk_at(const unsigned (&pk)[8], T* q, size_t cstride, bool valid, bool c0, bool c1) {
    typedef __attribute__((ext_vector_type(4))) unsigned u4;
    unsigned a0x = pk[0], a0y = pk[1], a1x = pk[2], a1y = pk[3], a2x = pk[4], a2y = pk[5], a3x = pk[6], a3y = pk[7];
    swap32x4(a0x, a1x, a0y, a1y, a2x, a3x, a2y, a3y);
    if (valid) {
        if (c0) { u4 v = {a0x, a0y, a1x, a1y}; *reinterpret_cast<u4*>(q) = v; }
        if (c1) { u4 v = {a2x, a2y, a3x, a3y}; *reinterpret_cast<u4*>(q + cstride) = v; }
    }
}

// + bias, ReLU, round to the storage type and store one accumulator tile (lane owns voxel column lane&31 and
// channel rows (r&3) + 8*(r>>2) + 4*(lane>>5)).
template <typename T>
__device__ __forceinline__ void store_acc_tile(const f32x16& acc, T* base, size_t P, size_t vidx, bool valid, int cbase, int half,
                                               const float* __restrict__ bias, int relu, int Cd) {
    using v4 = typename Act<T>::v4;
    v4 o[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int n = cbase + 4 * half + 8 * q;
        f32x4 b = {0.f, 0.f, 0.f, 0.f};
        if (n < Cd) b = *reinterpret_cast<const f32x4*>(bias + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = acc[4 * q + e] + b[e];
            if (relu) v = fmaxf(v, 0.f);
            o[q][e] = (T)v;
        }
    }
    store_tile_rows<T>(o, base, P, vidx, valid, cbase, half, Cd);
}

// Deferred GroupNorm apply of 8 channels of one voxel: relu(round_T(x*scale + shift)) == round_T(relu(x*scale + shift)), the
// arithmetic of k_gn_apply.  Written as "fma in fp32 from a 16-bit source, round once, then packed max with +0" so that
// the compiler can use v_fma_mix{lo,hi}_f16 (one instruction per element) and v_pk_max (one per pair).
template <typename T>
__device__ __forceinline__ typename Act<T>::v8 gn_apply8(typename Act<T>::v8 v, const float* sc, const float* sh, int relu) {
    typedef __attribute__((ext_vector_type(4))) unsigned u4;
    const f32x4 s0 = *reinterpret_cast<const f32x4*>(sc), s1 = *reinterpret_cast<const f32x4*>(sc + 4);
    const f32x4 t0 = *reinterpret_cast<const f32x4*>(sh), t1 = *reinterpret_cast<const f32x4*>(sh + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        v[e] = (T)fmaf((float)v[e], s0[e], t0[e]);
        v[4 + e] = (T)fmaf((float)v[4 + e], s1[e], t1[e]);
    }
    if (relu) {
        u4 u = __builtin_bit_cast(u4, v);
        u.x = pk_max16(u.x, 0u); u.y = pk_max16(u.y, 0u); u.z = pk_max16(u.z, 0u); u.w = pk_max16(u.w, 0u);
        v = __builtin_bit_cast(typename Act<T>::v8, u);
    }
    return v;
}

// ---------------------------------------------------------------------------------------------------------
// K2/K3: 3x3x3 / 1x3x3 'same' convolution, C_in >= 16, as a tap-looped implicit GEMM on the matrix cores,
// software-pipelined with LDS-DMA (global_load_lds, 16 B per lane).  Per workgroup: WAVES*64 output voxels x
// (NT*32) output channels; K = taps x C_in is walked in stages of (one 16-channel chunk) x (one kz plane):
//   * the halo block of chunk c+1 and the weight group of stage s+1 are DMA'd straight into LDS while stage s
//     computes -- no VGPR staging, ONE barrier per stage (stage = one 16-channel chunk x one kz plane = 9 taps);
//   * halo voxels are unpadded 32-byte records (DMA needs a lane-linear image); the two 16-byte halves of a
//     record are swapped on odd halo rows, which makes every ds_read_b128 of a 2x16-voxel fragment conflict-free
//     (checked exhaustively against the gfx950 lane groups); the swap is applied on the DMA source side;
//   * WAVES = 8: 512 output voxels per workgroup (3x3x3: 4x8x16, 1x3x3: 1x32x16) -> half the weight traffic and
//     less halo per voxel; WAVES = 4: 256 voxels for layers with few voxels.
template <int KZ, int WAVES, int MT> struct ConvGeo {      // MT = voxel tiles (2 y-rows x 16 x) per wave
    static constexpr int BZ = KZ == 3 ? (WAVES / 4) * MT : 1;      // 4 waves cover the 8 y-rows of one z-pair group
    static constexpr int BY = KZ == 3 ? 8 : WAVES * 2 * MT;
    static constexpr int BX = 16;
};


// Persistent form (NSLOT > 0): gridDim.x workgroups walk the output blocks round by round (block of round r =
// r*gridDim.x + XCD-contiguous remap of blockIdx.x) and ALL weight groups of the layer stay resident in LDS
// (level-0 layers: 18-72 KiB, loaded once per workgroup).  The halo chunks then form one continuous stream across
// blocks that is DMA'd through a ring of NSLOT LDS slots, NSLOT-1 chunks ahead of the MFMAs: these layers have few
// FLOPs per byte, so what bounds them is bytes in flight per CU (HBM latency x bandwidth ~ 50 KiB/CU), not a
// one-stage double buffer.  Every wave issues exactly AJ DMA instructions per chunk (padding ones go to a dummy
// slot) so that the stage-end wait is the compile-time counted `s_waitcnt vmcnt((NSLOT-2)*AJ)`; nothing inside the
// loop issues an ordinary VGPR load (bias and final-layer weights are preloaded), which would drain the ring.
// NSLOT == 0: weight groups are streamed (double-buffered); the workgroups are persistent as well, and the first
// weight group and halo chunk of a workgroup's next block are requested during the last stage of the current one.
// MODE 0: plain; 1 (FF): the first convolution is computed inside (see below); 2 (GN): one or both inputs are RAW tensors
// whose GroupNorm apply (+ReLU) was deferred to this consumer: every lane rewrites the 16-byte halo pieces it DMA'd into
// LDS as round_T(relu(x*scale + shift)) right after its own vmcnt wait and before the stage barrier (same arithmetic and
// rounding point as k_gn_apply, so results are bit-identical to the separate apply pass), with the per-(tile, channel)
// scale / shift of all tiles of the launch resident in LDS.  The normalised tensor is never written or re-read.
// halo DMA instructions with index < a_instr among pieces j0 ... j1-1 of wave w (piece j of wave w = instruction w + j * waves)
constexpr int dma_count(int w, int waves, int a_instr, int j0, int j1) {
    int n = 0;
    for (int j = j0; j < j1; ++j) n += (w + j * waves < a_instr) ? 1 : 0;
    return n;
}
// 3x3x3 stage with z-neighbour tiles (MT = 4; see the tap loop): LDS reads of a stage in program order -- W(0), X(kx 0, planes 0..3), then per
// step u = 3 kx + kz: W(u + 1), and per tile row i: [wait] MFMAs, refill (plane 4 after (kz 0, i 0), plane 5 after (kz 1, i 0), the next
// kx's plane i after (kz 2, i)).  LDS returns in order: the wait in front of row (u, i) may leave as many reads in flight as were issued
// after the youngest one it needs (X(kx, plane i + kz); for i = 0 also the last fragment of W(u)).
constexpr int zroll_wait(int nt, int u_want, int i_want) {
    int seq = 0;
    int wseq[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};       // sequence number of the LAST fragment of W(u)
    int xseq[3][6] = {{0}};
    seq += nt; wseq[0] = seq - 1;
    for (int m = 0; m < 4; ++m) xseq[0][m] = seq++;
    for (int u = 0; u < 9; ++u) {
        const int kx = u / 3, kz = u % 3;
        if (u + 1 < 9) { seq += nt; wseq[u + 1] = seq - 1; }
        for (int i = 0; i < 4; ++i) {
            if (u == u_want && i == i_want) {
                int need = xseq[kx][i + kz];
                if (i == 0 && wseq[u] > need) need = wseq[u];
                return seq - 1 - need;
            }
            if (kz == 0 && i == 0) xseq[kx][4] = seq++;
            else if (kz == 1 && i == 0) xseq[kx][5] = seq++;
            else if (kz == 2 && kx < 2) xseq[kx + 1][i] = seq++;
        }
    }
    return 0;
}
template <typename T, int KZ, int NT, int WAVES, int NSLOT, int MT, int MODE>
__global__ __launch_bounds__(WAVES * 64, WAVES == 12 ? 3 : 2) void k_conv_mfma(const ConvParams p) {
    // MODE 5 (FU8) = MODE 1 for UINT8 network input: the first convolution runs as two bf16 MFMAs per 32 halo voxels on the exact uint8
    // values (weights / 255 and bias split three ways into bf16 parts: sd_device.h first_u8_mfma) instead of five 64-cycle
    // v_mfma_f32_32x32x2_f32 -- a fifth of the matrix-pipe time; fp32-level result, not the bit pattern of the float32(v) / 255 chain.
    constexpr bool FF = MODE == 1 || MODE == 4 || MODE == 5, GN = MODE == 2, FU8 = MODE == 5;
    constexpr bool ZROLL = KZ == 3 && NT != 3;      // 3x3x3 stages by ky, taps (kx, kz) inside: see the stage loop
    // shader clock seen by this launch (bench.py: `roofline.clock_ghz_timed_region`): the first workgroup stamps the shader cycle counter
    // and the constant 100 MHz counter on entry and on exit (persistent workgroups live as long as the launch)
    if (p.clk && blockIdx.x == 0 && threadIdx.x == 0) { p.clk[0] = __builtin_readcyclecounter(); p.clk[1] = __builtin_amdgcn_s_memrealtime(); }
    // MODE 3 (SP, split-fp16 plan = act_dtype SD_F16X2): every tensor is stored as TWO fp16 planes per channel, x = hi + lo (hi =
    // fp16(x), lo = fp16(x - hi): 22 mantissa bits), as 2n chunks [n hi chunks | n lo chunks]; the weights are split the same
    // way (times a power of two per layer that keeps the lo parts normal; undone by p.oscale).  A product is computed as
    // Wlo.Xhi + Whi.Xhi + Whi.Xlo with fp32 accumulation: the chunk list of an n-chunk input is the 3n VIRTUAL chunks
    // [hi | hi | lo] of the same tensor (p.nchunk0 / p.nchunk1 count virtual chunks, the host packs the weight groups in that
    // order), so the stage loop is the one of the plain form run over three times the chunks.  The epilogue splits the fp32
    // results into hi / lo planes again (and can pool them: fused MaxPool3d; or feed a fused final 1x1x1 layer).  No GroupNorm statistics here.
    // MODE 4 = SP + FF: the split plan's first convolution (1 -> 32 channels) computed inside its consumer: its four halo planes
    // [hi0, hi1, lo0, lo1] are produced straight into FOUR resident LDS slots and the six virtual chunks [hi0, hi1, hi0, hi1, lo0,
    // lo1] of the block run on them -- the level-0 tensor behind the first convolution (268 MB per 128^3 tile as hi / lo planes)
    // is neither written nor read, and this convolution issues no halo DMA at all.
    constexpr bool SP = MODE == 3 || MODE == 4;
    static_assert(!SP || std::is_same<T, f16_t>::value, "split plan: fp16 planes");
    // register diet for the forms with >= 96 accumulator registers: nothing that can be recomputed per chunk stays live
    // across the block loop (DMA source addresses, halo-piece decode table)
    constexpr bool LEAN = MT == 4 || NT == 3;
    // SPREAD (asymmetric halo DMA): cycle stamps show that the two waves of a SIMD do not interleave their tap loops -- one
    // runs its 72 MFMAs at full rate (2.4 k cycles) while the other waits, then they swap -- and that the halo burst of a
    // kz = 0 stage (8 gathers per wave, ~3 k cycles with both waves of the SIMD issuing at the same time) delays BOTH of
    // them.  So the burst is moved into the time a wave would wait anyway: waves 4-7 issue theirs BEFORE their tap loop
    // (their SIMD partners 0-3 run MFMAs meanwhile), waves 0-3 AFTER their tap loop (their partners compute then).  Piece
    // addresses advance by additions (no divisions) and use 24-bit multiplies.
#ifndef SD_LATE_W
#define SD_LATE_W 1
#endif
#ifdef SD_NO_SPREAD
    constexpr bool SPREAD = false;
#else
    // (only the 4-tile form: with 2 voxel tiles per wave one wave cannot saturate the matrix pipe from its one-tap-ahead LDS
    // prefetch, the two waves of a SIMD really interleave, and the same schedule measured 2.4 % SLOWER on the 48-filter family)
    constexpr bool SPREAD = MT == 4 && KZ == 3 && WAVES == 8 && NSLOT == 0 && (MODE == 0 || MODE == 3);
#endif
    using v8 = typename Act<T>::v8;
    using G = ConvGeo<KZ, WAVES, MT>;
    constexpr bool WRES = NSLOT > 0;
    // RING (NSLOT < 0, planar layers with streamed weights): halo chunks AND weight groups go through rings of -NSLOT slots,
    // two stages ahead of the MFMAs.  A stage of these layers is short (36 MFMAs per wave) and its DMA group (3 halo gathers
    // + 3 weight pieces per wave, ~1.9 k cycles of the CU's address path with all 8 waves issuing at once) sat in front of
    // every tap loop.  Two stages of slack allow the asymmetric placement of the 3x3x3 form: waves 4-7 issue their group
    // before their tap loop, waves 0-3 after theirs, each under the MFMAs of its SIMD partner; the counted stage-end wait
    // leaves exactly the group just issued in flight.
    constexpr bool RING = NSLOT < 0;
    static_assert(!RING || (KZ == 1 && MODE == 0 && MT == 2 && NSLOT == -3), "ring form: planar, plain, 2 voxel tiles per wave");
    constexpr int NA = WRES ? NSLOT : RING ? -NSLOT : 2;   // halo slots
    constexpr int NW = RING ? -NSLOT : 2;                  // weight slots (streamed weights)
    constexpr int BZ = G::BZ, BY = G::BY, BX = G::BX;
    constexpr int PZ = KZ / 2;
    constexpr int HZ = BZ + KZ - 1, HY = BY + 2, HX = BX + 2;
    constexpr int NH = HZ * HY * HX;
    constexpr int A_INSTR = (NH * 2 + 63) / 64;          // 1 KiB DMA instructions per halo block
    constexpr int A_BYTES = A_INSTR * 1024;
    constexpr int B_INSTR = 9 * NT;
    constexpr int B_BYTES = B_INSTR * 1024;
    constexpr int SLICE = HY * HX * 32;
    constexpr int AJ = (A_INSTR + WAVES - 1) / WAVES;    // halo DMA instructions per wave per chunk
    constexpr int WJ = (B_INSTR + WAVES - 1) / WAVES;    // weight DMA instructions per wave and stage
    // DMA instructions allowed in flight at a stage end (RING: the group of AJ + WJ just issued)
    constexpr int WAITN = WRES ? (NA - 2) * AJ : RING ? AJ + WJ : 0;
    static_assert(WAITN < 64, "vmcnt immediate");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nchunks = p.nchunk0 + p.nchunk1;
    const int nstages = nchunks * KZ;
    char* const ldsA = smem;
    char* const ldsB = smem + NA * A_BYTES;
    // per-workgroup constants kept in LDS instead of registers (they would be live across the whole stage loop):
    // the folded bias of the NT*32 output channels (+ 8 class biases) and, with a fused final layer, its weight
    // fragments (NT*4 KiB, one 16-byte entry per lane and k-step)
    float* const wl = reinterpret_cast<float*>(ldsB + (WRES ? nstages : NW) * B_BYTES);
    char* const fwl = reinterpret_cast<char*>(wl) + SD_CONV_PARAM_BYTES;
    char* const ldsDummy = fwl + (p.final_wfrag ? NT * 4096 : 0);
    float* const fpatch = reinterpret_cast<float*>(ldsDummy + 1024);      // FF: normalised input patch (HY+2) x (HX+2)
    float* const gnl = fpatch;     // GN: [tile of the launch][chunk][16 scale, 16 shift] (FF and GN never coexist)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nsb = p.nbx * p.nby * p.nbz;
    const int nb = blockIdx.y;
    const int gsz = gridDim.x;

    // wave -> MT voxel tiles of (2 y-rows x 16 x); z-neighbours (3D) / y-neighbours (planar) share a wave
    int tzs[MT], tys[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        if (KZ == 3) { tzs[i] = MT * (wave >> 2) + i; tys[i] = 2 * (wave & 3); }
        else { tzs[i] = 0; tys[i] = 2 * MT * wave + 2 * i; }
    }
    const int dy = (lane & 31) >> 4, dxl = lane & 15, half = lane >> 5;
    int xoffE[MT], xoffO[MT];   // fragment read offsets for even / odd ky (row-parity swizzle of the 16-byte halves)
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int hv = (tzs[i] * HY + tys[i] + dy) * HX + dxl;
        xoffE[i] = hv * 32 + ((half ^ (dy & 1)) << 4);
        xoffO[i] = hv * 32 + ((half ^ (dy & 1) ^ 1) << 4);
    }

    // ordinary (VGPR-destination) global loads happen only here, before the first DMA is issued
    for (int idx = tid; idx < NT * 32; idx += WAVES * 64) {
        const int n = nb * NT * 32 + idx;
        wl[idx] = n < p.Cd ? p.bias[n] : 0.f;
    }
    if constexpr (GN) {
        // scale / shift of every (tile, input channel): source tables are [2*C] floats per tile (scale then shift)
        const int per_tile = nchunks * 32;
        for (int i = tid; i < p.batch * per_tile; i += WAVES * 64) {
            const int t = i / per_tile, r = i - t * per_tile, c = r >> 5, e = r & 31, ch = e & 15, is_shift = e >> 4;
            const bool s0 = c < p.nchunk0;
            const float* const base = s0 ? p.gn0 : p.gn1;
            const int C = s0 ? p.C0 : p.C1, cc = s0 ? c : c - p.nchunk0;
            float v = is_shift ? 0.f : 1.f;
            if (base) v = reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + (size_t)t * p.tstride)[is_shift * C + cc * SD_CHUNK + ch];
            gnl[i] = v;
        }
    }
    if (p.final_wfrag) {
        if (tid < 8) wl[NT * 32 + tid] = tid < p.final_cout ? p.final_b[tid] : 0.f;
        const T* const fwp = reinterpret_cast<const T*>(p.final_wfrag);
        for (int k = wave; k < NT * 4; k += WAVES)
            *reinterpret_cast<v8*>(fwl + (k * 64 + lane) * 16) = *reinterpret_cast<const v8*>(fwp + ((size_t)k * 64 + lane) * 8);
    }
    __syncthreads();

    const char* const wbase = reinterpret_cast<const char*>(p.wpack) + (size_t)nb * nstages * B_BYTES;

    auto dma_weights = [&](int s, int slot) {
        const char* src = wbase + (size_t)s * B_BYTES + lane * 16;
        char* dst = ldsB + slot * B_BYTES;
#pragma unroll
        for (int j = 0; j < (B_INSTR + WAVES - 1) / WAVES; ++j) {
            const int k = wave + j * WAVES;
            if (k < B_INSTR) glds16(src + k * 1024, dst + k * 1024);
        }
    };
    // halo voxel handled by this lane in its j-th DMA instruction of a chunk, packed hz<<20 | hy<<10 | hx<<1 | half
    // (the 16-byte halves of a 32-byte record are swapped on odd halo rows); -1 = beyond the halo block
    auto hpack_of = [&](int j) -> int {
        int idx = (wave + j * WAVES) * 64 + lane;
        if constexpr (LEAN && !GN && !(NT == 3 && MT == 2)) asm volatile("" : "+v"(idx));     // recomputed at every use (see dma_halo): no AJ live registers
        const int hv = idx >> 1;
        const int hx = hv % HX, hy = (hv / HX) % HY, hz = hv / (HX * HY);
        return (idx < NH * 2) ? ((hz << 20) | (hy << 10) | (hx << 1) | ((idx & 1) ^ (hy & 1))) : -1;
    };
    // (deferred GroupNorm apply: the decode table stays in registers -- it is needed twice per chunk, for the DMA and for the in-LDS
    // rewrite, and the ~30 VALU operations of a re-decode per piece are what the rewrite is made of)
    // POFF (every form outside the register diet, plain or split): a lane's halo-piece offsets inside a chunk plane are the same
    // for every chunk of a source tensor, so they are computed once per (block, source) into AJ registers (0xffffffff = beyond
    // the volume -> the zero page); a chunk's DMA is then one 64-bit add and one select per piece instead of decode, three bounds
    // tests, two quarter-rate multiplies and the add -- and the decode table plus what the compiler hoisted around it (12-60
    // registers, e.g. 213 -> 166 for the streamed 3x3x3 NT = 2 form, 171 -> 120 for its NT = 1 sibling) is gone.  Same addresses,
    // bit-identical results; time: -0.5 % on semseg_spine bf16 / semseg_axon, +-0 on the split plan (A/B of two libraries on one
    // box) -- the stage loop of these forms has ~58 VALU per 18 MFMAs, their epilogues 1000-1600 per block: that is where their
    // issue slots go.
#ifdef SD_NO_POFF
    constexpr bool POFF = false;
#else
    constexpr bool POFF = !LEAN && !GN;
#endif
    constexpr bool HPACK_REGS = !POFF && (!LEAN || GN || (NT == 3 && MT == 2));
    int hpack[HPACK_REGS ? AJ : 1];
    if constexpr (HPACK_REGS) {
#pragma unroll
        for (int j = 0; j < AJ; ++j) hpack[j] = hpack_of(j);
    }
    unsigned poff[POFF ? AJ : 1];
    auto halo_offsets = [&](int bz, int by, int bx, int Hs, int Ws) {
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            int idx = (wave + j * WAVES) * 64 + lane;
            asm volatile("" : "+v"(idx));      // decoded here, once per block: nothing of it stays live across the block loop
            const int hv = idx >> 1;
            const int hx = hv % HX, hy = (hv / HX) % HY, hz = hv / (HX * HY);
            const int z = bz - PZ + hz, y = by - 1 + hy, x = bx - 1 + hx;
            const bool ok = idx < NH * 2 && (unsigned)z < (unsigned)p.D && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
            // (< 2^32 bytes per chunk plane: sd_forward's size limit)
            const unsigned o = ((((unsigned)z * (unsigned)Hs + (unsigned)y) * (unsigned)Ws + (unsigned)x) * SD_CHUNK +
                                (unsigned)(((idx & 1) ^ (hy & 1)) * 8)) * (unsigned)sizeof(T);
            poff[j] = ok ? o : 0xffffffffu;
        }
    };
    // logical block of (round, this workgroup); -1 when the round has no block for it
    const int nsb_all = nsb * p.batch;      // the tiles of a batched launch are simply more blocks
#ifdef SD_XCD_ROUNDS
    auto block_of = [&](int round) -> int {
        const int base = round * gsz;
        const int n = min(gsz, nsb_all - base);
        return ((int)blockIdx.x < n) ? base + xcd_remap(blockIdx.x, n) : -1;
    };
#else
    // Every XCD (workgroup b runs on XCD b % 8) owns ONE contiguous range of the block list and walks it round by
    // round: consecutive rounds of an XCD are neighbouring rows / z-slabs, whose shared halo planes are then still in
    // that XCD's L2 (with the blocks of a round spread over the whole list they were fetched from HBM again).
    const int xk = blockIdx.x & 7, xj = blockIdx.x >> 3;
    const int xw = gsz / 8 + (xk < gsz % 8);                             // workgroups of this XCD
    const int xsize = nsb_all / 8 + (xk < nsb_all % 8);                  // blocks of this XCD
    const int xstart = xk * (nsb_all / 8) + min(xk, nsb_all % 8);
    auto block_of = [&](int round) -> int {
        const int idx = round * xw + xj;
        return idx < xsize ? xstart + idx : -1;
    };
#endif
    auto coords = [&](int lb, int& z0, int& y0, int& x0, int& tile) {
        tile = lb / nsb;
        lb -= tile * nsb;
        // 3x3x3: z fastest -- the blocks an XCD works on at the same time are then z-neighbours of one (y,x) column,
        // whose shared halo planes (2 of 6, the largest overlap of a 4x8x16 block) hit in L2 (-12 % HBM reads);
        // 1x3x3 blocks share nothing along z: x fastest, then y
        if (KZ == 3 && p.block_order == 1) {
            // brick order: y in slabs of 4 block rows, x in strips of 2 block columns, inside a (slab, strip) column z
            // slowest.  32 consecutive list entries -- what the 32 workgroups of an XCD work on at the same time -- are then
            // a compact 4 x 4 x 2 brick of blocks (16 x 32 x 32 voxels: each halo plane is shared with a block that is in
            // the same L2 at the same time in all three directions, not only along z), and an XCD's next round is the
            // brick above it.  Edge slabs / strips are simply narrower: a bijection for any grid.
            constexpr int BYB = 4, BXB = 2;
            const int slab_full = BYB * p.nbx * p.nbz;
            const int sl = lb / slab_full;
            int rem = lb - sl * slab_full;
            const int h = min(BYB, p.nby - sl * BYB);
            const int strip_full = BXB * h * p.nbz;
            const int st = rem / strip_full;
            rem -= st * strip_full;
            const int w = min(BXB, p.nbx - st * BXB);
            const int zi = rem / (h * w);
            rem -= zi * h * w;
            z0 = zi * BZ; y0 = (sl * BYB + rem / w) * BY; x0 = (st * BXB + rem % w) * BX;
        }
        else if (KZ == 3) { z0 = (lb % p.nbz) * BZ; x0 = ((lb / p.nbz) % p.nbx) * BX; y0 = (lb / (p.nbz * p.nbx)) * BY; }
        else { x0 = (lb % p.nbx) * BX; y0 = ((lb / p.nbx) % p.nby) * BY; z0 = (lb / (p.nbx * p.nby)) * BZ; }
    };
    // DMA chunk c of the block at (z0,y0,x0) into halo slot `slot`; real == false issues the same number of
    // instructions against the dummy slot (keeps the per-wave DMA count per chunk constant for the counted waits)
    auto dma_halo = [&](int c, int slot, int z0, int y0, int x0, int tile, bool real) {
        if constexpr (LEAN) {
            // keep the per-piece source addresses from being hoisted out of the block loop (up to 8 x 64-bit per wave): with
            // 96-128 accumulator registers there is no room for them; recomputing costs a few VALU ops per chunk
            z0 = __builtin_amdgcn_readfirstlane(z0); y0 = __builtin_amdgcn_readfirstlane(y0); x0 = __builtin_amdgcn_readfirstlane(x0);
            asm volatile("" : "+s"(z0), "+s"(y0), "+s"(x0));
        }
        const char* sbase;
        int Hs, Ws, cc;
        size_t Ps;
        if (c < p.nchunk0) { sbase = (const char*)p.src0; Ps = p.P0; Hs = p.H0; Ws = p.W0; cc = c; }
        else { sbase = (const char*)p.src1; Ps = p.P1; Hs = p.H1; Ws = p.W1; cc = c - p.nchunk0; }
        if constexpr (SP) {      // virtual chunks [hi | hi | lo] of an n-chunk tensor stored as [hi | lo]
            const int n = (c < p.nchunk0 ? p.nchunk0 : p.nchunk1) / 3;
            if (cc >= n) cc -= n;
        }
        sbase += (size_t)cc * Ps * (SD_CHUNK * sizeof(T)) + (size_t)tile * p.tstride;      // chunk plane of this tile
        char* dst = ldsA + slot * A_BYTES + wave * 1024;
        if constexpr (POFF) {
            // chunk 0 is the first DMA of a block, chunk nchunk0 the first of the second source (other extents behind an up-convolution)
            if (real && (c == 0 || (c == p.nchunk0 && (p.H1 != p.H0 || p.W1 != p.W0)))) halo_offsets(z0, y0, x0, Hs, Ws);
#pragma unroll
            for (int j = 0; j < AJ; ++j) {
                const bool inst = real && (wave + j * WAVES < A_INSTR);       // wave-uniform
                if (inst || NA > 2) {
                    const unsigned o = poff[j];
                    const char* src = reinterpret_cast<const char*>(p.zero);
                    if (inst && o != 0xffffffffu) src = sbase + o;
                    glds16(src, inst ? dst + j * (WAVES * 1024) : ldsDummy);
                }
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const bool inst = real && (wave + j * WAVES < A_INSTR);       // wave-uniform
            if (inst || NA > 2) {
                const int hp = HPACK_REGS ? hpack[HPACK_REGS ? j : 0] : hpack_of(j);
                const int z = z0 - PZ + (hp >> 20), y = y0 - 1 + ((hp >> 10) & 1023), x = x0 - 1 + ((hp >> 1) & 511);
                const bool ok = inst && hp >= 0 && (unsigned)z < (unsigned)p.D && (unsigned)y < (unsigned)p.H &&
                                (unsigned)x < (unsigned)p.W;
                const char* src = reinterpret_cast<const char*>(p.zero);
                if (ok) src = sbase + (((size_t)(z * Hs + y) * Ws + x) * SD_CHUNK + (hp & 1) * 8) * sizeof(T);
                glds16(src, inst ? dst + j * (WAVES * 1024) : ldsDummy);
            }
        }
    };
    // GN: what the last halo DMA of this wave was issued for (chunk, slot, block origin, tile); transformed in LDS once
    // it has landed
    int pd_c = -1, pd_slot = 0, pd_z = 0, pd_y = 0, pd_x = 0, pd_t = 0;
    auto gn_note = [&](int c, int slot, int bz, int by, int bx, int bt, bool real) {
        if constexpr (GN) { pd_c = real ? c : -1; pd_slot = slot; pd_z = bz; pd_y = by; pd_x = bx; pd_t = bt; }
    };
    auto gn_transform = [&]() {
        if constexpr (GN) {
            const int c = pd_c;
            pd_c = -1;
            if (c < 0) return;
            const bool s0 = c < p.nchunk0;
            if (!(s0 ? p.gn0 : p.gn1)) return;                     // this input is already normalised (wave-uniform)
            const int relu = s0 ? p.gn_relu0 : p.gn_relu1;
            const float* const tab = gnl + ((size_t)pd_t * nchunks + c) * 32;
            char* const base = ldsA + pd_slot * A_BYTES + wave * 1024 + lane * 16;
#pragma unroll
            for (int j = 0; j < AJ; ++j) {
                if (wave + j * WAVES >= A_INSTR) continue;         // wave-uniform
                const int hp = HPACK_REGS ? hpack[HPACK_REGS ? j : 0] : hpack_of(j);
                const int z = pd_z - PZ + (hp >> 20), y = pd_y - 1 + ((hp >> 10) & 1023), x = pd_x - 1 + ((hp >> 1) & 511);
                const bool ok = hp >= 0 && (unsigned)z < (unsigned)p.D && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
                if (ok) {                                          // out-of-volume pieces stay zero: the conv's zero padding
                    v8* const q = reinterpret_cast<v8*>(base + j * (WAVES * 1024));
                    const float* const sc = tab + (hp & 1) * 8;
                    *q = gn_apply8<T>(*q, sc, sc + 16, relu);
                }
            }
        }
    };
    // chunk number f of this workgroup's stream (f = round * nchunks + c) -> ring slot f % NA
    // The stream position is carried incrementally (chunk within block, block coordinates) so that the integer
    // divisions of block_of / coords run once per BLOCK, not once per chunk.
    int sf_c = 0, sf_round = 0, sf_slot = 0, sf_z = 0, sf_y = 0, sf_x = 0, sf_t = 0;
    bool sf_ok = false;
    auto stream_block = [&]() {
        const int lbf = block_of(sf_round);
        sf_ok = lbf >= 0;
        if (sf_ok) coords(lbf, sf_z, sf_y, sf_x, sf_t);
    };
    auto dma_stream_next = [&]() {
        if (sf_c == 0) stream_block();
        dma_halo(sf_c, sf_slot, sf_z, sf_y, sf_x, sf_t, sf_ok);
        gn_note(sf_c, sf_slot, sf_z, sf_y, sf_x, sf_t, sf_ok);
        if (++sf_c == nchunks) { sf_c = 0; ++sf_round; }
        if (++sf_slot == NA) sf_slot = 0;
    };

    // RING: weight group and halo chunk of the next stream position (= two stages ahead) into its ring slot; every wave
    // issues exactly WJ + AJ instructions (padding ones / positions behind the workgroup's last block go to the dummy slot)
    auto ring_issue = [&]() {
        if (sf_c == 0) stream_block();
        const char* src = wbase + (size_t)sf_c * B_BYTES + lane * 16;
        char* dst = ldsB + sf_slot * B_BYTES;
#pragma unroll
        for (int j = 0; j < WJ; ++j) {
            const int k = wave + j * WAVES;
            const bool inst = sf_ok && k < B_INSTR;                    // wave-uniform
            glds16(inst ? src + k * 1024 : reinterpret_cast<const char*>(p.zero), inst ? dst + k * 1024 : ldsDummy);
        }
        dma_halo(sf_c, sf_slot, sf_z, sf_y, sf_x, sf_t, sf_ok);
        if (++sf_c == nchunks) { sf_c = 0; ++sf_round; }
        if (++sf_slot == NA) sf_slot = 0;
    };

#ifdef SD_TIMING
    long long tstamp[8];
    long long sstamp[16];      // SD_STAGES: end of every stage of the probed block, [14] block start, [15] epilogue end
    int tcount = 0;
#define SD_T(i) do { if (tcount == SD_TB) tstamp[i] = __builtin_readcyclecounter(); } while (0)
#else
#define SD_T(i) do {} while (0)
#endif
#ifndef SD_TS
#define SD_TS 0
#endif
#ifndef SD_TB
#define SD_TB 2      // probed block of the workgroup (1 = first)
#endif
    int lb = block_of(0);
    if (lb < 0) return;
    int z0, y0, x0, tn;
    coords(lb, z0, y0, x0, tn);

    // FF: the normalised input patch of a block ((HY+2) x (HX+2) floats, two per thread) is fetched ONE BLOCK AHEAD into
    // registers and parked in the other half of a double-buffered LDS patch, so that no block starts by waiting for a
    // cold HBM load behind the previous block's output stores (vmcnt counts both, in order).
    constexpr int FPX = HX + 2, FPY = HY + 2, FNP = FPX * FPY, FPT = (FNP + WAVES * 64 - 1) / (WAVES * 64);
    float fpv[FF ? FPT : 1];
    auto patch_fetch = [&](int bz0, int by0, int bx0, int btile) {
#pragma unroll
        for (int k = 0; k < FPT; ++k) {
            const int i = tid + k * WAVES * 64;
            const int px = i % FPX, py = i / FPX;
            const int y = by0 - 2 + py, x = bx0 - 2 + px;
            float v = 0.f;
            if (i < FNP && (unsigned)bz0 < (unsigned)p.D && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W) {
                const size_t idx = ((size_t)bz0 * p.H + y) * p.W + x;
                const char* const in = reinterpret_cast<const char*>(p.first_in) + (size_t)btile * p.first_in_tstride;
                if constexpr (FU8) v = __builtin_bit_cast(float, u8_bf16_pair(reinterpret_cast<const uint8_t*>(in)[idx]));
                else if (p.first_in_f32) v = reinterpret_cast<const float*>(in)[idx];
                else v = (float)reinterpret_cast<const uint8_t*>(in)[idx] / 255.0f;
            }
            fpv[k] = v;
        }
    };
    auto patch_park = [&](int buf) {
#pragma unroll
        for (int k = 0; k < FPT; ++k) {
            const int i = tid + k * WAVES * 64;
            if (i < FNP) fpatch[buf * FNP + i] = fpv[k];
        }
    };
    // first-conv weight fragments and bias live in LDS as well: an ordinary global load inside the block loop would make
    // the wave wait (vmcnt, in order) for the previous block's output stores
    // NF1 = 32-channel tiles of the first convolution: 1 (32 filters; also the split plan), 2 (48 filters: three halo chunks, NA = 3)
    constexpr int NF1 = (FF && !SP && NA == 3) ? 2 : 1;
    float* const ffw = fpatch + 2 * FNP;                      // [NF1][5 k-steps][64 lanes] + [NF1 * 32] bias
    if constexpr (FF) {
        patch_fetch(z0, y0, x0, tn);
        patch_park(0);
        if constexpr (FU8) {      // [0] = the constant (1.0, 1.0) the bias parts meet; from +16 bytes: [NF1][2 MFMAs][64 lanes][8] bf16
            if (tid == 0) reinterpret_cast<unsigned*>(ffw)[0] = SD_BF16_ONE_PAIR;
            typedef __attribute__((ext_vector_type(4))) unsigned u4;
            for (int i = tid; i < NF1 * 2 * 64; i += WAVES * 64) reinterpret_cast<u4*>(ffw + 4)[i] = reinterpret_cast<const u4*>(p.first_w3)[i];
        } else
        for (int i = tid; i < NF1 * (5 * 64 + 32); i += WAVES * 64) ffw[i] = i < NF1 * 320 ? p.first_w[i] : p.first_bias[i - NF1 * 320];
    }
    if (WRES) {
        for (int s = 0; s < nstages; ++s) dma_weights(s, s);
        if (!FF) for (int f = 0; f < NA - 1; ++f) dma_stream_next();
    } else if (RING) {
        for (int f = 0; f < NA - 1; ++f) ring_issue();
    } else {
        dma_weights(0, 0);
        dma_halo(0, 0, z0, y0, x0, tn, true);
        gn_note(0, 0, z0, y0, x0, tn, true);
    }
    if constexpr (GN) {
        static_assert(!GN || WAITN == 0, "deferred GroupNorm apply: every halo DMA is awaited at its stage end");
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");     // also: the scale / shift table is in LDS
        gn_transform();
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(WAITN) : "memory");
    }

    StoreGuard<T> sguard;      // fp16 range guard over everything this lane rounds to the storage type (sd_device.h)
    // Static issue priority for the waves that run their tap loop FIRST in every stage (0 ... WAVES/2-1, also the older half of the
    // workgroup): -0.8 % on the six 3x3x3 launches of the headline net, A/B on one box; the other half at priority 1 instead:
    // +7 % (tools/experiments/round4_rejected.md).  Arithmetic untouched.
#ifndef SD_PRIO
#define SD_PRIO 2
#endif
    if constexpr (SPREAD && SD_PRIO != 0) {
        if ((SD_PRIO == 1) == (wave >= WAVES / 2)) asm volatile("s_setprio 1");
    }
    int gc = 0, gs = 0;   // chunk / stage counters across blocks (slot parity)
    int ph0 = 0;          // SPREAD: halo coordinates of this lane's first piece, packed z << 16 | y << 8 | x
    if constexpr (SPREAD) {
        const int hv0 = (wave * 64 + lane) >> 1;
        ph0 = ((hv0 / (HX * HY)) << 16) | (((hv0 / HX) % HY) << 8) | (hv0 % HX);
        static_assert(!SPREAD || (HX < 256 && HY < 256), "packed piece coordinates");
    }
    // MODE 5: lane constants of the FIT tiles of 32 halo voxels a wave converts per block -- patch byte offset of the voxel, its halo
    // coordinates hy << 16 | hx (-1: beyond the halo block), and this lane's byte offset inside a tile's 1 KiB of halo records
    constexpr int FIT = FU8 ? (NH + 32 * WAVES - 1) / (32 * WAVES) : 1;
    int fc_off[FIT], fc_yx[FIT], fc_wr = (lane & 31) * 32 + half * 8;
    if constexpr (FU8) {
#pragma unroll
        for (int it = 0; it < FIT; ++it) {
            const int hv = (wave + it * WAVES) * 32 + (lane & 31);
            const int hvc = hv < NH ? hv : NH - 1;
            const int hy = hvc / HX, hx = hvc % HX;
            fc_off[it] = (hy * (HX + 2) + hx) * 4;
            fc_yx[it] = hv < NH ? ((hy << 16) | hx) : -1;
            asm volatile("" : "+v"(fc_off[it]), "+v"(fc_yx[it]));      // (kept in registers, not re-derived per block)
        }
        asm volatile("" : "+v"(fc_wr));
    }
    for (int round = 0; lb >= 0; ++round) {
        const int nlb = block_of(round + 1);
        int nz0 = 0, ny0 = 0, nx0 = 0, ntn = 0;
        if (nlb >= 0) coords(nlb, nz0, ny0, nx0, ntn);

        if constexpr (FF) {
            // ---- fused FIRST convolution (1 -> 32 channels, 1x3x3, + BN + ReLU): the two 16-channel halo chunks of this
            // block are COMPUTED from the uint8 / float input tile instead of DMA'd from a materialised tensor (which
            // is never written).  Same arithmetic as k_conv_first: float32(v)/255 by IEEE division, the 9 taps as the
            // k dimension of exact-f32 32x32x2 MFMAs, bias after the chain, ReLU, rounding -- bit-identical values.
            static_assert(!FF || (KZ == 1 && WRES && (SP ? NA == 4 : (NA == 2 || NA == 3))), "fused first conv: planar, resident weights");
            constexpr int PXW = HX + 2, NSTEP1 = 5;
            const float* const fp = fpatch + (round & 1) * FNP;      // parked by the prologue / during the previous block
            if constexpr (FU8) {
                const unsigned* const up = reinterpret_cast<const unsigned*>(fp);
                const unsigned* const cpair = reinterpret_cast<const unsigned*>(ffw);
                const bf16x8* const w3l = reinterpret_cast<const bf16x8*>(ffw + 4);
                bf16x8 w30 = w3l[lane], w31 = w3l[64 + lane];
                int toff5[5];
#pragma unroll
                for (int a = 0; a < 5; ++a) {
                    const int tap = half * 5 + a;      // lanes 0-31: taps 0-4; lanes 32-63: taps 5-8, then the constant
                    toff5[a] = tap < 9 ? ((tap / 3) * PXW + (tap % 3)) * 4 : 0;      // (bytes)
                }
                const unsigned relu_floor = p.first_relu ? 0u : 0x80008000u;      // packed signed max against the most negative pair = identity
                const char* const upb = reinterpret_cast<const char*>(up);
#pragma unroll
                for (int it = 0; it < FIT; ++it) {
                    const int t = wave + it * WAVES;
                    if (t * 32 >= NH) break;                 // (wave-uniform)
                    const int yx = fc_yx[it];                // lane constants of this wave's it-th tile of halo voxels (below, before the block loop)
                    const bool inb = yx >= 0;
                    const int hy = yx >> 16, hx = yx & 0xffff;
                    const char* const ub = upb + fc_off[it];
                    unsigned r[5];
#pragma unroll
                    for (int a = 0; a < 4; ++a) r[a] = *reinterpret_cast<const unsigned*>(ub + toff5[a]);
                    r[4] = *reinterpret_cast<const unsigned*>(half ? reinterpret_cast<const char*>(cpair) : ub + toff5[4]);
                    const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
                    const bool invol = (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;     // z0 < D always
                    const bool border = __any(!invol);      // (wave-uniform: interior tiles skip the zero-padding selects)
                    const int hyb = (hy & 1) << 4;      // odd halo rows: the 16-byte halves of a record trade places
                    char* const wr0 = ldsA + t * 1024 + fc_wr + hyb;
                    char* const wr1 = ldsA + t * 1024 + fc_wr + (hyb ^ 16);
#pragma unroll
                    for (int ft = 0; ft < NF1; ++ft) {      // (48 filters: channels 0-31, then 32-47 + padding; fragments re-read per tile)
                        if constexpr (NF1 > 1) { w30 = w3l[(ft * 2 + 0) * 64 + lane]; w31 = w3l[(ft * 2 + 1) * 64 + lane]; }
                        const f32x16 a1 = first_u8_mfma(w30, w31, r);
                        unsigned o[8];
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            o[k] = pk_max16(Act<T>::pack2(a1[2 * k], a1[2 * k + 1]), relu_floor);
                            sguard.see_signed(o[k]);
                        }
                        if (border) {      // the second conv's zero padding (a real branch: the volatile statement keeps it from becoming eight selects)
                            asm volatile("; halo tile at the volume border");
                            if (!invol) {
#pragma unroll
                                for (int k = 0; k < 8; ++k) o[k] = 0u;
                            }
                        }
                        if (inb) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                typedef __attribute__((ext_vector_type(2))) unsigned u2;
                                if (NF1 > 1 && 2 * ft + (q >> 1) >= NA) continue;      // (the padding chunk of a 48-filter first conv)
                                const int so = ((gc + 2 * ft + (q >> 1)) % NA) * A_BYTES;
                                *reinterpret_cast<u2*>(((q & 1) ? wr1 : wr0) + so) = u2{o[2 * q], o[2 * q + 1]};
                            }
                        }
                    }
                }
            } else {
            float w1[NSTEP1];
            int toff1[NSTEP1];
#pragma unroll
            for (int st = 0; st < NSTEP1; ++st) {
                w1[st] = ffw[st * 64 + lane];
                int tap = 2 * st + half;
                if (tap >= 9) tap = 0;                       // (its weight is zero)
                toff1[st] = (tap / 3) * PXW + (tap % 3);
            }
            // (the folded bias is the weight of the chain's last, otherwise empty k slot and meets a 1.0 there: see k_conv_first)
            for (int t = wave; t * 32 < NH; t += WAVES) {
                const int hv = t * 32 + (lane & 31);
                const int hvc = hv < NH ? hv : NH - 1;
                const int hy = hvc / HX, hx = hvc % HX;
                const int base = hy * PXW + hx;
#pragma unroll
                for (int ft = 0; ft < NF1; ++ft) {      // (48 filters: channels 0-31, then 32-47 + padding; fragments re-read per tile)
                if constexpr (NF1 > 1) {
#pragma unroll
                    for (int st = 0; st < NSTEP1; ++st) w1[st] = ffw[(ft * NSTEP1 + st) * 64 + lane];
                }
                f32x16 a1;
#pragma unroll
                for (int r = 0; r < 16; ++r) a1[r] = 0.f;
#pragma unroll
                for (int st = 0; st < NSTEP1; ++st)
                    a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[st], (st == NSTEP1 - 1 && half) ? 1.0f : fp[base + toff1[st]], a1, 0, 0, 0);
                const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
                const bool invol = (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;     // z0 < D always
                if constexpr (SP) {
                    if (hv < NH) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {      // channels 8q + 4*half + 0..3: chunk q >> 1, 16-byte half q & 1 of the record
                            float v0 = a1[4 * q], v1 = a1[4 * q + 1], v2 = a1[4 * q + 2], v3 = a1[4 * q + 3];
                            if (p.first_relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
                            unsigned h01, l01, h23, l23;
                            split_pk(v0, v1, h01, l01);
                            split_pk(v2, v3, h23, l23);
                            sguard.see_signed(h01); sguard.see_signed(h23);
                            if (!invol) { h01 = 0u; l01 = 0u; h23 = 0u; l23 = 0u; }      // the second conv's zero padding
                            typedef __attribute__((ext_vector_type(2))) unsigned u2;
                            const int ro = hv * 32 + ((((q & 1) ^ (hy & 1))) << 4) + half * 8;
                            *reinterpret_cast<u2*>(ldsA + (q >> 1) * A_BYTES + ro) = u2{h01, h23};           // slots 0, 1: hi planes
                            *reinterpret_cast<u2*>(ldsA + (2 + (q >> 1)) * A_BYTES + ro) = u2{l01, l23};     // slots 2, 3: lo planes
                        }
                    }
                } else
                if (hv < NH) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        unsigned lo = Act<T>::pack2(a1[4 * q], a1[4 * q + 1]);
                        unsigned hi = Act<T>::pack2(a1[4 * q + 2], a1[4 * q + 3]);
                        if (p.first_relu) { lo = pk_max16(lo, 0u); hi = pk_max16(hi, 0u); }
                        sguard.see_signed(lo); sguard.see_signed(hi);
                        if (!invol) { lo = 0u; hi = 0u; }      // the second conv's zero padding
                        typedef __attribute__((ext_vector_type(2))) unsigned u2;
                        if (NF1 > 1 && 2 * ft + (q >> 1) >= NA) continue;      // (the padding chunk of a 48-filter first conv)
                        char* const slot = ldsA + ((gc + 2 * ft + (q >> 1)) % NA) * A_BYTES;
                        *reinterpret_cast<u2*>(slot + hv * 32 + ((((q & 1) ^ (hy & 1))) << 4) + half * 8) = u2{lo, hi};
                    }
                }
                }
            }
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (nlb >= 0) patch_fetch(nz0, ny0, nx0, ntn);       // lands during stage 0 (its end waits vmcnt(0))
        }
        f32x16 acc[MT][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = wl[j * 32 + 4 * half + 8 * (r >> 2) + (r & 3)];
#ifdef SD_TIMING
        ++tcount;
#endif
        SD_T(0);
#ifdef SD_RT
        const long long rt0 = __builtin_amdgcn_s_memrealtime();
#endif
        int s = 0;
        for (int c = 0; c < nchunks; ++c, ++gc) {
            // (split plan with the first convolution inside: the planes live in fixed slots [hi0, hi1, lo0, lo1])
            const char* const abuf = ldsA + ((SP && FF) ? (c < 4 ? (c & 1) : c - 2) : (gc % NA)) * A_BYTES;
            // SPREAD: the chunk fetched during this chunk's stages (the next chunk of this block, or the first chunk of the
            // workgroup's next block); everything a piece needs is pinned in scalar registers here -- a kernel-argument
            // s_load inside the tap loop would count on lgkmcnt and break the counted LDS waits
            bool pf_real = false;
            const char* pf_sbase = nullptr;
            const char* pf_zero = reinterpret_cast<const char*>(p.zero);
            char* pf_dst = nullptr;
            int pf_Hs = 0, pf_Ws = 0, pf_z = 0, pf_y = 0, pf_x = 0, pf_D = p.D, pf_H = p.H, pf_W = p.W;
            // (every wave runs this set-up in front of the chunk's first tap loop -- cycle stamps show that stage 600 cycles longer than the
            // other two.  Round 6 moved it behind the first tap loop for the waves that issue their pieces there: 2 % SLOWER on the 3x3x3
            // launches, tools/experiments/round6_notes.md: the lock-step start of a stage is worth more than the idle set-up costs.)
            auto pf_setup = [&]() {
                int pc = c + 1, pt = tn;
                pf_real = true; pf_z = z0; pf_y = y0; pf_x = x0;
                if (pc == nchunks) { pc = 0; pt = ntn; pf_z = nz0; pf_y = ny0; pf_x = nx0; pf_real = nlb >= 0; }
                size_t Ps;
                int pn;      // (SP) virtual chunks of the source tensor
                if (pc < p.nchunk0) { pf_sbase = (const char*)p.src0; Ps = p.P0; pf_Hs = p.H0; pf_Ws = p.W0; pn = p.nchunk0; }
                else { pf_sbase = (const char*)p.src1; Ps = p.P1; pf_Hs = p.H1; pf_Ws = p.W1; pc -= p.nchunk0; pn = p.nchunk1; }
                if constexpr (SP) { if (pc >= pn / 3) pc -= pn / 3; }      // [hi | hi | lo] -> planes [hi | lo]
                pf_sbase += (size_t)pc * Ps * (SD_CHUNK * sizeof(T)) + (size_t)pt * p.tstride;
                pf_z -= PZ; pf_y -= 1; pf_x -= 1;
                pf_dst = ldsA + ((gc + 1) & 1) * A_BYTES + wave * 1024;
                // (uniform by construction; readfirstlane makes that explicit for values that went through VALU divisions)
                auto rfl = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
                {
                    const uint64_t a = reinterpret_cast<uint64_t>(pf_sbase);
                    pf_sbase = reinterpret_cast<const char*>(((uint64_t)(unsigned)rfl((int)(a >> 32)) << 32) | (unsigned)rfl((int)a));
                }
                pf_Hs = rfl(pf_Hs); pf_Ws = rfl(pf_Ws); pf_z = rfl(pf_z); pf_y = rfl(pf_y); pf_x = rfl(pf_x);
                asm volatile("" : "+s"(pf_sbase), "+s"(pf_zero), "+s"(pf_Hs), "+s"(pf_Ws), "+s"(pf_z), "+s"(pf_y), "+s"(pf_x),
                             "+s"(pf_D), "+s"(pf_H), "+s"(pf_W));
            };
            (void)pf_setup;
            if constexpr (SPREAD) pf_setup();
            // halo coordinates of this lane's NEXT piece, advanced piece by piece (one piece = WAVES * 32 halo voxels further):
            // additions and two carries instead of the divisions of hpack_of (quarter-rate multiplies)
            constexpr int SV = WAVES * 32, SDZ = SV / (HY * HX), SDY = (SV % (HY * HX)) / HX, SDX = SV % HX;
            static_assert(SDY + 1 <= HY && SDX < HX, "piece stride");
            int ph_x = ph0 & 255, ph_y = (ph0 >> 8) & 255, ph_z = ph0 >> 16;     // piece 0 (decoded once per kernel)
            auto dma_piece = [&](int j) {                      // j-th halo piece of this wave (wave-uniform j, increasing)
                const int k = wave + j * WAVES;
                if (pf_real && j < AJ && k < A_INSTR) {
                    const int idx = k * 64 + lane;
                    const int z = pf_z + ph_z, y = pf_y + ph_y, x = pf_x + ph_x;
                    const bool ok = idx < NH * 2 && (unsigned)z < (unsigned)pf_D && (unsigned)y < (unsigned)pf_H &&
                                    (unsigned)x < (unsigned)pf_W;
                    const unsigned vox = ((unsigned)z * (unsigned)pf_Hs + (unsigned)y) * (unsigned)pf_Ws + (unsigned)x;   // < 2^32 voxels per tensor
                    const unsigned hf = ((unsigned)(lane ^ ph_y) & 1u) << 4;
                    const uint64_t off = ((uint64_t)(vox >> 27) << 32) | ((vox << 5) | hf);
                    const char* src = pf_zero;
                    if (ok) src = pf_sbase + off;
                    glds16(src, pf_dst + j * (WAVES * 1024));
                }
                ph_x += SDX;
                if (ph_x >= HX) { ph_x -= HX; ++ph_y; }
                ph_y += SDY;
                if (ph_y >= HY) { ph_y -= HY; ++ph_z; }
                ph_z += SDZ;
            };
            // Schedule of a wave's AJ pieces over the three stages of a chunk.  Waves WAVES/2 ... WAVES-1 ("early") issue EB pieces
            // BEFORE their tap loop in every stage; waves 0 ... WAVES/2-1 ("late") issue LB pieces AFTER their tap loop in the
            // kz = 0 / 1 stages (what is issued at the end of the kz = 2 stage would be awaited at once).  *_MIN: pieces the
            // highest wave of the group really issues in that stage (those with an instruction index < A_INSTR) -- the counted
            // stage-end wait may leave exactly that many in flight: they are younger than the weight group.
            constexpr int EB = (AJ + 2) / 3, LB = (AJ + 1) / 2;
            constexpr int E0 = dma_count(WAVES - 1, WAVES, A_INSTR, 0, EB), E1 = dma_count(WAVES - 1, WAVES, A_INSTR, EB, 2 * EB);
            constexpr int L0 = dma_count(WAVES / 2 - 1, WAVES, A_INSTR, 0, LB), L1 = dma_count(WAVES / 2 - 1, WAVES, A_INSTR, LB, 2 * LB);
            static_assert(!SPREAD || (3 * EB >= AJ && 2 * LB >= AJ), "halo piece schedule");
            auto halo_pieces = [&](int j0, int n) {
                for (int j = j0; j < j0 + n; ++j) dma_piece(j);
            };
#pragma unroll 1
            for (int kz = 0; kz < KZ; ++kz, ++s, ++gs) {
                if (s == SD_TS) SD_T(7);     // start of the probed stage
                if constexpr (RING) {
                    if (wave >= WAVES / 2) ring_issue();
                } else if (!WRES) {
                    // next weight group / next halo chunk into the other buffer; at the end of a block these are
                    // the first group and chunk of the workgroup's NEXT block, so its prologue hides behind the
                    // last stage and the epilogue of this one
                    // (SPREAD: the waves that run their tap loop first fetch their share of the weight group after it)
                    // (not in the last stage of a block: those waves go on to their epilogue, which must not wait for a DMA)
                    if (!(SPREAD && SD_LATE_W) || wave >= WAVES / 2 || s + 1 == nstages) {
                        if (s + 1 < nstages) dma_weights(s + 1, (gs + 1) & 1);
                        else if (nlb >= 0) dma_weights(0, (gs + 1) & 1);
                    }
                    if constexpr (SPREAD) {
                        if (wave >= WAVES / 2) halo_pieces(kz * EB, EB);
                    }
                    if (!SPREAD && kz == 0) {
                        if (c + 1 < nchunks) {
                            dma_halo(c + 1, (gc + 1) & 1, z0, y0, x0, tn, true);
                            gn_note(c + 1, (gc + 1) & 1, z0, y0, x0, tn, true);
                        } else if (nlb >= 0) {
                            dma_halo(0, (gc + 1) & 1, nz0, ny0, nx0, ntn, true);
                            gn_note(0, (gc + 1) & 1, nz0, ny0, nx0, ntn, true);
                        }
                    }
                } else if (kz == 0 && !FF) {
                    dma_stream_next();           // chunk gc + NA - 1 of this workgroup's stream
                }
                // Deferred GroupNorm apply, 3x3x3 layers: the chunk requested in the kz = 0 stage is only needed three stages later,
                // so its in-LDS rewrite does not have to sit between that stage's DMA wait and its barrier (where every wave of
                // the workgroup did it at the same time, ~2 k cycles per chunk with the matrix pipe idle).  It runs in the kz = 1
                // stage instead, ASYMMETRICALLY: waves 0 ... WAVES/2-1 rewrite their pieces BEFORE their tap loop, their SIMD
                // partners WAVES/2 ... AFTER theirs -- each group's VALU / LDS work runs under the other group's MFMAs.
                constexpr bool GN_ASYNC = GN && KZ == 3 && !WRES;
                if constexpr (GN_ASYNC) {
                    if (kz == 1 && wave < WAVES / 2) { gn_transform(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
                }
                if (s == SD_TS) SD_T(1);     // after the DMA issue of the probed stage
                const char* const bcur = ldsB + (WRES ? s : RING ? gs % NW : (gs & 1)) * B_BYTES + lane * 16;
                // ZROLL: the three stages of a chunk are the three KY rows of the stencil (the loop variable keeps its name); inside a stage
                // the taps run kx-major, kz-minor -- every output is summed in the order (chunk, ky, kx, kz) (the weight groups are packed in
                // that order, sd_api.hip).  A wave's voxel tiles are z-neighbours, so tile i at kz reads the halo plane
                // tile i + 1 reads at kz - 1: with kz innermost a fragment serves up to three MFMA rows (MT = 4 below).
                // (ZROLL = 3x3x3 layers with NT <= 2.  The NT = 3 forms -- the 48-filter family, two voxel tiles per wave, nothing to share --
                // keep kz stages and the order (chunk, kz, ky, kx): which order a LAYER is summed in depends on its channel count alone,
                // never on the launch, so a layer gives the same bits in every form that can run it.)
                const char* const acur = ZROLL ? abuf + kz * (HX * 32) : abuf + kz * SLICE;
                const bool odd_row = ZROLL && (kz & 1);      // (halo-row parity of the stage = which 16-byte half order the fragments have)
                // software-pipelined over the 9 taps: the fragments of tap t+1 are in flight while tap t's MFMAs run.
                // The LDS reads and their counted waits are inline asm: left to the compiler the reads are sunk next
                // to their use behind an lgkmcnt(0) (it prefers reusing the fragment registers), which idles the
                // matrix pipe for one LDS latency per tap.  LDS returns in order, so lgkmcnt(MT + NT) after issuing
                // tap t+1 means tap t has landed; `tie` makes the MFMAs depend on the post-wait values.
                if constexpr (MT == 4 && ZROLL) {
                    // 4 z-neighbour tiles per wave, planes P0 ... P5 of the halo at this stage's ky row: step u = (kx, kz) runs the MFMA rows
                    // (tile i) x (NT weight fragments of tap (kz, ky, kx)) on plane i + kz.  The four fragment registers rotate through the
                    // six planes of a kx column -- P4 replaces P0 right behind P0's only row, P5 replaces P1 behind P1's last row, and behind
                    // the kz = 2 rows the next column's P0 ... P3 come in: 6 x fragments + 3 NT weight fragments per 12 NT MFMAs (NT = 2:
                    // 0.5 LDS fragment reads per MFMA; the kz-stage form read 0.75 -- every halo plane three times per chunk).  Same
                    // registers as before (single-buffered x fragments refilled in place, double-buffered weight fragments), counted waits
                    // from the replayed issue order (zroll_wait).
                    v8 xq[4], wq[2][NT];
                    const uint32_t bA = lds_addr(bcur);
                    const uint32_t xb = lds_addr(acur) + (odd_row ? xoffO[0] : xoffE[0]);
                    auto load_p = [&](auto kxc, auto mc) {
                        constexpr int kx = decltype(kxc)::value, m = decltype(mc)::value;
                        ds_read16<kx * 32 + m * SLICE>(xq[(m + 2 * kx) & 3], xb);
                    };
                    auto load_w = [&](auto uc) {
                        constexpr int u = decltype(uc)::value;
                        static_for<NT>([&](auto jc) {
                            constexpr int j = decltype(jc)::value;
                            ds_read16<(u * NT + j) * 1024>(wq[u & 1][j], bA);
                        });
                    };
                    load_w(std::integral_constant<int, 0>{});
                    static_for<4>([&](auto mc) { load_p(std::integral_constant<int, 0>{}, mc); });
                    static_for<9>([&](auto uc) {
                        constexpr int u = decltype(uc)::value, kx = u / 3, kzz = u % 3, buf = u & 1;
                        if constexpr (u + 1 < 9) load_w(std::integral_constant<int, u + 1>{});
                        static_for<4>([&](auto ic) {
                            constexpr int i = decltype(ic)::value, slot = (i + kzz + 2 * kx) & 3;
                            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(zroll_wait(NT, u, i)));
                            tie(xq[slot]);
                            if constexpr (i == 0) {
#pragma unroll
                                for (int j = 0; j < NT; ++j) tie(wq[buf][j]);
                            }
#pragma unroll
                            for (int j = 0; j < NT; ++j) acc[i][j] = Act<T>::mfma(wq[buf][j], xq[slot], acc[i][j]);
                            if constexpr (kzz == 0 && i == 0) load_p(std::integral_constant<int, kx>{}, std::integral_constant<int, 4>{});
                            else if constexpr (kzz == 1 && i == 0) load_p(std::integral_constant<int, kx>{}, std::integral_constant<int, 5>{});
                            else if constexpr (kzz == 2 && kx < 2) load_p(std::integral_constant<int, kx + 1>{}, ic);
                        });
                    });
                } else if constexpr (MT == 4) {
                    // 4 voxel tiles per wave (KZ == 3: z-neighbours, so tile i's fragment sits i * SLICE bytes behind tile 0's:
                    // a ds_read offset immediate, ONE address register pair for all tiles).  Register budget: 128
                    // accumulator registers leave no room for double-buffered fragments, so the x fragments are single-
                    // buffered and REFILLED IN PLACE -- tile i's fragment of tap t+1 is requested right behind the two
                    // MFMAs that last read tap t's; only the weight fragments are double-buffered.  LDS returns in
                    // order: whenever a tile row starts, exactly 3 + NT younger reads are in flight (3 - i of this tap's x
                    // fragments, NT weight fragments and i x fragments of the next tap) -> one constant lgkmcnt.
                    // (planar layers, round 4: the four tiles are y-neighbours, two halo rows apart -- the same immediate-offset scheme)
                    constexpr int TSTEP = KZ == 3 ? SLICE : 2 * HX * 32;
                    v8 xq[4], wq[2][NT];
                    const uint32_t bA = lds_addr(bcur);
                    const uint32_t xE = lds_addr(acur) + xoffE[0], xO = lds_addr(acur) + xoffO[0];
                    auto load_x = [&](auto tc, auto ic) {
                        constexpr int t9 = decltype(tc)::value, i = decltype(ic)::value, ky = t9 / 3, kx = t9 % 3;
                        ds_read16<(ky * HX + kx) * 32 + i * TSTEP>(xq[i], (ky & 1) ? xO : xE);
                    };
                    auto load_w = [&](auto tc) {
                        constexpr int t9 = decltype(tc)::value;
                        static_for<NT>([&](auto jc) {
                            constexpr int j = decltype(jc)::value;
                            ds_read16<(t9 * NT + j) * 1024>(wq[t9 & 1][j], bA);
                        });
                    };
                    load_w(std::integral_constant<int, 0>{});
                    static_for<4>([&](auto ic) { load_x(std::integral_constant<int, 0>{}, ic); });
                    static_for<9>([&](auto tc) {
                        constexpr int t9 = decltype(tc)::value, buf = t9 & 1;
                        if constexpr (t9 + 1 < 9) load_w(std::integral_constant<int, t9 + 1>{});
                        static_for<4>([&](auto ic) {
                            constexpr int i = decltype(ic)::value;
                            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(t9 + 1 < 9 ? 3 + NT : 3 - i));
                            tie(xq[i]);
                            if constexpr (i == 0) {
#pragma unroll
                                for (int j = 0; j < NT; ++j) tie(wq[buf][j]);
                            }
#pragma unroll
                            for (int j = 0; j < NT; ++j) acc[i][j] = Act<T>::mfma(wq[buf][j], xq[i], acc[i][j]);
                            if constexpr (t9 + 1 < 9) load_x(std::integral_constant<int, t9 + 1>{}, ic);
                        });
                    });
                } else {
                v8 xf[2][MT], wf[2][NT];
                const uint32_t aE = lds_addr(acur), bA = lds_addr(bcur);
                uint32_t xaE[MT], xaO[MT];
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    xaE[i] = aE + xoffE[i]; xaO[i] = aE + xoffO[i];
                    if constexpr (ZROLL) { if (odd_row) { const uint32_t t = xaE[i]; xaE[i] = xaO[i]; xaO[i] = t; } }      // (the stage IS the ky row)
                }
                auto load_tap = [&](auto tc) {
                    // tap t9 = (ky, kx) of the stage's kz plane; ZROLL: step t9 = (kx, kz) of the stage's ky row
                    constexpr int t9 = decltype(tc)::value, ky = ZROLL ? 0 : t9 / 3, kx = ZROLL ? t9 / 3 : t9 % 3, buf = t9 & 1;
                    constexpr int tapoff = ZROLL ? kx * 32 + (t9 % 3) * SLICE : (ky * HX + kx) * 32;
#pragma unroll
                    for (int i = 0; i < MT; ++i) ds_read16<tapoff>(xf[buf][i], (ky & 1) ? xaO[i] : xaE[i]);
                    static_for<NT>([&](auto jc) {
                        constexpr int j = decltype(jc)::value;
                        ds_read16<(t9 * NT + j) * 1024>(wf[buf][j], bA);
                    });
                };
                load_tap(std::integral_constant<int, 0>{});
                static_for<9>([&](auto tc) {
                    constexpr int t9 = decltype(tc)::value, buf = t9 & 1;
                    if constexpr (t9 + 1 < 9) {
                        load_tap(std::integral_constant<int, t9 + 1>{});
                        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(MT + NT));
                    } else {
                        asm volatile("s_waitcnt lgkmcnt(0)");
                    }
#pragma unroll
                    for (int i = 0; i < MT; ++i) tie(xf[buf][i]);
#pragma unroll
                    for (int j = 0; j < NT; ++j) tie(wf[buf][j]);
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j)
                            acc[i][j] = Act<T>::mfma(wf[buf][j], xf[buf][i], acc[i][j]);
                });
                }
                if (s == SD_TS) SD_T(2);     // after the MFMAs of the probed stage
                // all of this wave's LDS reads done + the next chunk's DMA landed, then the workgroup barrier (raw
                // s_barrier: __syncthreads() would drain every DMA in flight with a vmcnt(0))
                if constexpr (GN_ASYNC) {
                    if (kz == 1 && wave >= WAVES / 2) gn_transform();
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                } else if constexpr (GN) {
                    // the chunk requested in this stage has landed: rewrite my pieces of it in place, then the barrier
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    gn_transform();
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                } else if constexpr (RING) {
                    if (wave < WAVES / 2) ring_issue();
                    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(WAITN) : "memory");
                } else if constexpr (SPREAD) {
                    const bool late = wave < WAVES / 2;
                    if (SD_LATE_W && late && s + 1 < nstages) dma_weights(s + 1, (gs + 1) & 1);
                    if (late && kz < 2) halo_pieces(kz * LB, LB);
                    // the weight group of the next stage was issued BEFORE this stage's pieces (vmcnt completes in order)
                    const int fly = !pf_real || kz == 2 ? 0 : late ? (kz == 0 ? L0 : L1) : (kz == 0 ? E0 : E1);
                    static_assert(!SPREAD || (E0 <= 4 && E1 <= 4 && L0 <= 4 && L1 <= 4), "counted waits below");
                    if (fly == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    else if (fly == 3) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    else if (fly == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    else if (fly == 1) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    else if (s + 1 == nstages && !p.gn_sums) {
                        // ASYMMETRIC EPILOGUE: the barrier that ends the last stage is taken INSIDE the epilogue.  Waves 0 ...
                        // WAVES/2-1 ran their tap loop first: instead of waiting for their SIMD partners they convert and store
                        // ALL their output tiles now, under the partners' MFMAs, and meet the barrier behind their epilogue.
                        // Waves WAVES/2 ... finish their tap loop later, store their FIRST channel group while the others are
                        // still in their epilogue, meet the barrier, and store the rest under the first-stage MFMAs of the
                        // others' next block.  No vmcnt wait sits between the stores and the barrier: everything this wave
                        // has DMA'd is awaited here, before the stores.
                        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    }
                    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(WAITN) : "memory");
                }
                if constexpr (FF) {
                    if (s == 0 && nlb >= 0) patch_park((round + 1) & 1);     // ordered by the barrier that ends stage 1
                }
                if (s == SD_TS) SD_T(3);     // after the barrier of the probed stage
#ifdef SD_STAGES
                if (tcount == SD_TB && s < 14) sstamp[s] = __builtin_readcyclecounter();
#endif
            }
        }

    SD_T(4);   // all stages done
    if constexpr (SP) {
        // ---- epilogue of the split-fp16 form: undo the weight scale, ReLU in fp32, hi = fp16(v), lo = fp16(v - hi); the hi tile
        // goes to chunk plane (channel / 16), the lo tile Cd / 16 planes further.  Fused MaxPool3d(ceil_mode): the window maximum
        // of the fp32 values ((kz,2,2) window = {z pair of tiles (3D)} x {lane^16 (y)} x {lane^1 (x)}; voxels beyond the volume
        // count as -inf), split like every other value -- exactly what the separate pooling pass computes from the stored pairs.
        T* const dsts = reinterpret_cast<T*>(reinterpret_cast<char*>(p.dst) + (size_t)tn * p.tstride);
        T* const dstl = dsts + (size_t)(p.Cd >> 4) * p.Pd * SD_CHUNK;
        T* const pdsts = reinterpret_cast<T*>(reinterpret_cast<char*>(p.pool_dst) + (size_t)tn * p.tstride);
        T* const pdstl = pdsts + (size_t)(p.Cd >> 4) * p.Pp * SD_CHUNK;
        const float osc = p.oscale;
        constexpr int NP = KZ == 3 ? 2 : 1;      // tiles per pooling window
        static_assert(MT % NP == 0, "z pairs of tiles");
        // fused GroupNorm statistics (the scheme of the other forms: the matrix core transposes a tile -- packed tile as A operand x
        // 0/1 selector -- so that a lane holds 16 voxels of ONE channel; here hi and lo tiles go through the same accumulator, which
        // then holds the exact fp32 values hi + lo)
        // (not in the 4-wave streamed NT = 3 form: 254 VGPRs without it, the launcher picks the 8-wave form for such a layer; not
        // in the NT = 1 forms either: their 32-channel level-0 layers live on two workgroups per CU = at most 128 VGPRs, which the
        // statistics code would break -- the plan does not fuse the statistics of a layer with one channel tile)
        constexpr bool GNS = NT >= 2 && !(KZ == 3 && NT == 3 && WAVES == 4 && !WRES);
        const bool gns = GNS && p.gn_sums != nullptr;
        float* const part = reinterpret_cast<float*>(ldsA + ((gc - 1) % NA) * A_BYTES);   // free: last chunk's halo slot
        v8 s1, s2;
        if (gns) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = 8 * (e >> 2) + 4 * half + (e & 3);
                s1[e] = (T)((c == (lane & 31)) ? 1.0f : 0.0f);
                s2[e] = (T)((c + 16 == (lane & 31)) ? 1.0f : 0.0f);
            }
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            float sj = 0.f, ssj = 0.f;
#pragma unroll
            for (int ip = 0; ip < MT; ip += NP) {
                bool vals[NP];
#pragma unroll
                for (int e2 = 0; e2 < NP; ++e2) {
                    const int i = ip + e2;
                    const int vz = z0 + tzs[i], vy = y0 + tys[i] + dy, vx = x0 + dxl;
                    const bool val = vz < p.D && vy < p.H && vx < p.W;
                    const size_t vo = (size_t)(vz * p.Hd + vy) * p.Wd + vx;
                    vals[e2] = val;
                    if (p.store_main) {
                        unsigned ph[8], pl[8];
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            float a = acc[i][j][2 * k] * osc, b = acc[i][j][2 * k + 1] * osc;
                            if (p.relu) { a = fmaxf(a, 0.f); b = fmaxf(b, 0.f); }
                            split_pk(a, b, ph[k], pl[k]);
                            sguard.see_signed(ph[k]);
                        }
                        store_tile_rows_pk<T>(ph, dsts, p.Pd, vo, val, (nb * NT + j) * 32, half, p.Cd);
                        store_tile_rows_pk<T>(pl, dstl, p.Pd, vo, val, (nb * NT + j) * 32, half, p.Cd);
                        if (gns) {
                            typedef __attribute__((ext_vector_type(4))) unsigned u4;
                            unsigned zr = 0u;
                            asm volatile("" : "+v"(zr));
                            u4 h0 = {ph[0], ph[1], ph[2], ph[3]}, h1 = {ph[4], ph[5], ph[6], ph[7]};
                            u4 l0 = {pl[0], pl[1], pl[2], pl[3]}, l1 = {pl[4], pl[5], pl[6], pl[7]};
                            if (!val) { h0 = u4{zr, zr, zr, zr}; h1 = h0; l0 = h0; l1 = h0; }
                            f32x16 d;
#pragma unroll
                            for (int r = 0; r < 16; ++r) d[r] = 0.f;
                            d = Act<T>::mfma(__builtin_bit_cast(v8, l0), s1, d);
                            d = Act<T>::mfma(__builtin_bit_cast(v8, l1), s2, d);
                            d = Act<T>::mfma(__builtin_bit_cast(v8, h0), s1, d);
                            d = Act<T>::mfma(__builtin_bit_cast(v8, h1), s2, d);
#pragma unroll
                            for (int r = 0; r < 16; ++r) { sj += d[r]; ssj = fmaf(d[r], d[r], ssj); }
                        }
                    }
                }
                if (p.pool_dst) {      // (values recomputed from the accumulators pair by pair: nothing extra stays live)
                    unsigned ph[8], pl[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        float a = -INFINITY, b = -INFINITY;
#pragma unroll
                        for (int e2 = 0; e2 < NP; ++e2) {
                            float ae = acc[ip + e2][j][2 * k] * osc, be = acc[ip + e2][j][2 * k + 1] * osc;
                            if (p.relu) { ae = fmaxf(ae, 0.f); be = fmaxf(be, 0.f); }
                            a = fmaxf(a, vals[e2] ? ae : -INFINITY);
                            b = fmaxf(b, vals[e2] ? be : -INFINITY);
                        }
                        a = max_xor16(max_xor1(a));
                        b = max_xor16(max_xor1(b));
                        split_pk(a, b, ph[k], pl[k]);      // (range guard: these are values the main store has already seen)
                    }
                    const int pz = (KZ == 3) ? (z0 + tzs[ip]) >> 1 : z0 + tzs[ip], py = (y0 + tys[ip]) >> 1, px = (x0 + dxl) >> 1;
                    const size_t po = (size_t)(pz * p.pH + py) * p.pW + px;
                    const bool writer = (dy == 0) && ((dxl & 1) == 0) && vals[0];
                    store_tile_rows_pk<T>(ph, pdsts, p.Pp, po, writer, (nb * NT + j) * 32, half, p.Cd);
                    store_tile_rows_pk<T>(pl, pdstl, p.Pp, po, writer, (nb * NT + j) * 32, half, p.Cd);
                }
            }
            if (gns) {
                float* const q = part + ((size_t)(wave * 2 + half) * (NT * 32) + j * 32 + (lane & 31)) * 2;
                q[0] = sj; q[1] = ssj;
            }
            if constexpr (SPREAD) {      // (asymmetric epilogue, see the stage loop)
                if (j == 0 && wave >= WAVES / 2 && !p.gn_sums) asm volatile("s_barrier" ::: "memory");
            }
        }
        if (gns) {      // waves and half-waves combined in a fixed order, one double atomic per channel and statistic per block
            __syncthreads();
            if (tid < NT * 64) {
                const int stat = tid / (NT * 32), cw = tid % (NT * 32);
                double t = 0.0;
                for (int w = 0; w < WAVES * 2; ++w) t += (double)part[((size_t)w * (NT * 32) + cw) * 2 + stat];
                const int ch = nb * NT * 32 + cw;
                if (ch < p.Cd)
                    atomicAdd(reinterpret_cast<double*>(reinterpret_cast<char*>(p.gn_sums) + (size_t)tn * p.tstride) +
                                  (size_t)stat * p.gn_C + ch, t);
            }
            __syncthreads();         // the slot is a DMA target again in the next block
        }
        // ---- fused conv_final (1x1x1) + softmax / uint8 / labels of the split plan: logits = W . (hi + lo) on the matrix core as
        // Wlo.Xhi + Whi.Xlo + Whi.Xhi with the fp32 final weights as scaled fp16 hi / lo fragments (the layout of the fp16 plan's
        // fused final layer: the B fragment of k-step s is the pair of packed quads (2s, 2s + 1) a lane holds); the output tensor
        // of this convolution is then neither written nor read back
        // (planar forms only -- every U-Net of the path ends in planar blocks -- so that the 3x3x3 forms do not carry its scalars)
        if constexpr (MT == 2 && KZ == 1) {
        if (p.final_wfrag) {
            const FinalOut fo{p.final_out, p.out_tstride, p.final_cout, p.final_kind, p.final_nvox, p.ovf};
            typedef __attribute__((ext_vector_type(4))) unsigned u4;
#pragma unroll
            for (int tp = 0; tp < MT; tp += 2) {
                f32x16 lgt[2];
                bool vv[2];
                size_t vo2[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int vz = z0 + tzs[tp + i], vy = y0 + tys[tp + i] + dy, vx = x0 + dxl;
                    vv[i] = vz < p.D && vy < p.H && vx < p.W;
                    vo2[i] = (size_t)(vz * p.Hd + vy) * p.Wd + vx;
#pragma unroll
                    for (int r = 0; r < 16; ++r) lgt[i][r] = 0.f;
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        unsigned ph[8], pl[8];
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            float a = acc[tp + i][j][2 * k] * osc, b = acc[tp + i][j][2 * k + 1] * osc;
                            if (p.relu) { a = fmaxf(a, 0.f); b = fmaxf(b, 0.f); }
                            split_pk(a, b, ph[k], pl[k]);
                            sguard.see_signed(ph[k]);
                        }
#pragma unroll
                        for (int s2 = 0; s2 < 2; ++s2) {
                            const v8 bh = __builtin_bit_cast(v8, u4{ph[4 * s2], ph[4 * s2 + 1], ph[4 * s2 + 2], ph[4 * s2 + 3]});
                            const v8 bl = __builtin_bit_cast(v8, u4{pl[4 * s2], pl[4 * s2 + 1], pl[4 * s2 + 2], pl[4 * s2 + 3]});
                            const v8 fw0 = *reinterpret_cast<const v8*>(fwl + (((j * 2 + s2) * 2 + 0) * 64 + lane) * 16);
                            const v8 fw1 = *reinterpret_cast<const v8*>(fwl + (((j * 2 + s2) * 2 + 1) * 64 + lane) * 16);
                            lgt[i] = Act<T>::mfma(fw1, bh, lgt[i]);
                            lgt[i] = Act<T>::mfma(fw0, bl, lgt[i]);
                            lgt[i] = Act<T>::mfma(fw0, bh, lgt[i]);
                        }
                    }
                }
                float l[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // (undo the weight scale and add the class bias by a compiler-visible VALU op in front of the asm swap)
                    const float bmine = wl[NT * 32 + 4 * half + e];
                    unsigned a = __builtin_bit_cast(unsigned, fmaf(lgt[0][e], p.final_oscale, bmine));
                    unsigned b2 = __builtin_bit_cast(unsigned, fmaf(lgt[1][e], p.final_oscale, bmine));
                    swap32(a, b2);      // lower lanes: all 8 logits of tile tp's voxel; upper lanes: tile tp + 1's
                    l[e] = __builtin_bit_cast(float, a);
                    l[4 + e] = __builtin_bit_cast(float, b2);
                }
                if (half ? vv[1] : vv[0]) final_finish_exact<T>(l, fo, p.lab, tn, half ? vo2[1] : vo2[0]);
            }
        }
        }
    } else if constexpr (MT == 4) {
        // ---- epilogue of the 4-tile form, one z-PAIR of tiles and one 32-channel group at a time (the packed values of
        // all 8 accumulator tiles at once would not fit beside the accumulators): pack + ReLU, GroupNorm partial sums, main
        // store, fused 2x2x2 pooling of the pair.
        typedef __attribute__((ext_vector_type(4))) unsigned u4;
        T* const dst4 = reinterpret_cast<T*>(reinterpret_cast<char*>(p.dst) + (size_t)tn * p.tstride);
        const int vy = y0 + tys[0] + dy, vx = x0 + dxl, vz0 = z0 + tzs[0];
        const bool vyx = vy < p.H && vx < p.W;
        // voxels between the wave's tiles: a z-plane (3x3x3: z-stacked tiles) or two rows (planar: y-stacked tiles)
        const size_t vo0 = (size_t)(vz0 * p.Hd + vy) * p.Wd + vx, vzs = KZ == 3 ? (size_t)p.Hd * p.Wd : (size_t)2 * p.Wd;
        float* const part = reinterpret_cast<float*>(ldsA + ((gc - 1) % NA) * A_BYTES);   // free: last chunk's halo slot
        const unsigned relu_floor = p.relu ? 0u : 0x80008000u;
        const unsigned guard_mask = 0x7fff7fffu;      // (range guard: magnitudes)
        v8 s1, s2;
        if (p.gn_sums) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = 8 * (e >> 2) + 4 * half + (e & 3);
                s1[e] = (T)((c == (lane & 31)) ? 1.0f : 0.0f);
                s2[e] = (T)((c + 16 == (lane & 31)) ? 1.0f : 0.0f);
            }
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            float sj = 0.f, ssj = 0.f;
            // this lane's half record of its voxel in z-plane vz0, first chunk of channel group j (tile z-planes and the
            // second chunk are wave-uniform offsets from it)
            T* const qj = dst4 + ((size_t)((nb * NT + j) * 2) * p.Pd + vo0) * SD_CHUNK + half * 8;
            const size_t cstr = p.Pd * SD_CHUNK;
            const bool cj0 = (nb * NT + j) * 32 < p.Cd, cj1 = (nb * NT + j) * 32 + 16 < p.Cd;
#pragma unroll
            for (int ip = 0; ip < 4; ip += 2) {
                unsigned pk2[2][8];
                bool val2[2];
#pragma unroll
                for (int e2 = 0; e2 < 2; ++e2) {
                    val2[e2] = KZ == 3 ? (vyx && (vz0 + ip + e2) < p.D) : (vx < p.W && vz0 < p.D && vy + 2 * (ip + e2) < p.H);
#pragma unroll
                    for (int k = 0; k < 8; ++k)      // ReLU = packed max against 0; without it against the most negative pair (identity)
                    {
                        pk2[e2][k] = pk_max16(Act<T>::pack2(acc[ip + e2][j][2 * k], acc[ip + e2][j][2 * k + 1]), relu_floor);
                        sguard.see(pk2[e2][k] & guard_mask);
                    }
                    if (p.gn_sums) {
                        u4 lo = {pk2[e2][0], pk2[e2][1], pk2[e2][2], pk2[e2][3]};
                        u4 hi = {pk2[e2][4], pk2[e2][5], pk2[e2][6], pk2[e2][7]};
                        unsigned zr = 0u;
                        asm volatile("" : "+v"(zr));      // (keeps the eight selects inside this branch: BatchNorm nets never take it)
                        if (!val2[e2]) { lo = u4{zr, zr, zr, zr}; hi = lo; }
                        f32x16 d;
#pragma unroll
                        for (int r = 0; r < 16; ++r) d[r] = 0.f;
                        d = Act<T>::mfma(__builtin_bit_cast(v8, lo), s1, d);
                        d = Act<T>::mfma(__builtin_bit_cast(v8, hi), s2, d);
#pragma unroll
                        for (int r = 0; r < 16; ++r) { sj += d[r]; ssj = fmaf(d[r], d[r], ssj); }
                    }
                    if (p.store_main)
                        store_tile_rows_pk_at<T>(pk2[e2], qj + (size_t)(ip + e2) * (vzs * SD_CHUNK), cstr, val2[e2], cj0, cj1);
                }
                if (p.pool_dst) {
                    T* const pdst = reinterpret_cast<T*>(reinterpret_cast<char*>(p.pool_dst) + (size_t)tn * p.tstride);
                    const bool writer = (dy == 0) && ((dxl & 1) == 0);
                    if constexpr (KZ == 3) {
                        unsigned m[8];
#pragma unroll
                        for (int k = 0; k < 8; ++k) m[k] = pk_max16(val2[0] ? pk2[0][k] : 0u, val2[1] ? pk2[1][k] : 0u);
                        pool_xy_pk8(m);
                        const int pz = (vz0 + ip) >> 1, py = (y0 + tys[0]) >> 1, px = (x0 + dxl) >> 1;
                        store_tile_rows_pk<T>(m, pdst, p.Pp, (size_t)(pz * p.pH + py) * p.pW + px, writer && val2[0],
                                              (nb * NT + j) * 32, half, p.Cd);
                    } else {
#pragma unroll
                        for (int e2 = 0; e2 < 2; ++e2) {      // planar: (1,2,2) windows, every tile pools alone
                            unsigned m[8];
#pragma unroll
                            for (int k = 0; k < 8; ++k) m[k] = val2[e2] ? pk2[e2][k] : 0u;
                            pool_xy_pk8(m);
                            const int py = (y0 + tys[ip + e2]) >> 1, px = (x0 + dxl) >> 1;
                            store_tile_rows_pk<T>(m, pdst, p.Pp, (size_t)(vz0 * p.pH + py) * p.pW + px, writer && val2[e2],
                                                  (nb * NT + j) * 32, half, p.Cd);
                        }
                    }
                }
            }
            if (p.gn_sums) {
                float* const q = part + ((size_t)(wave * 2 + half) * (NT * 32) + j * 32 + (lane & 31)) * 2;
                q[0] = sj; q[1] = ssj;
            }
            if constexpr (SPREAD) {      // (asymmetric epilogue, see the stage loop)
                if (j == 0 && wave >= WAVES / 2 && !p.gn_sums) asm volatile("s_barrier" ::: "memory");
            }
        }
        if (p.gn_sums) {
            __syncthreads();
            if (tid < NT * 64) {
                const int stat = tid / (NT * 32), cw = tid % (NT * 32);
                double t = 0.0;
                for (int w = 0; w < WAVES * 2; ++w) t += (double)part[((size_t)w * (NT * 32) + cw) * 2 + stat];
                const int ch = nb * NT * 32 + cw;
                if (ch < p.Cd)
                    atomicAdd(reinterpret_cast<double*>(reinterpret_cast<char*>(p.gn_sums) + (size_t)tn * p.tstride) +
                                  (size_t)stat * p.gn_C + ch, t);
            }
            __syncthreads();
        }
    } else {
    // ---- epilogue: + bias, ReLU, round to the storage type (kept in `acc` as the rounded value) ----------
    bool valid[MT];
    size_t voxoff[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int vz = z0 + tzs[i], vy = y0 + tys[i] + dy, vx = x0 + dxl;
        valid[i] = vz < p.D && vy < p.H && vx < p.W;
        voxoff[i] = (size_t)(vz * p.Hd + vy) * p.Wd + vx;        // voxel index inside the tile's tensor (Hd, Wd: its y / x extents)
    }
    T* const dst = reinterpret_cast<T*>(reinterpret_cast<char*>(p.dst) + (size_t)tn * p.tstride);
    // rounded outputs, packed two channels per register: pk[i][j][2q + h] = channels cbase + 8q + 4*half + 2h, +1.
    // (VALU work is 4 cycles per wave64 instruction and the whole workgroup sits in this epilogue at once, so it is
    // kept to one convert and one packed max per PAIR of values.)
    unsigned pk[MT][NT][8];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int k = 0; k < 8; ++k)     // bias is already inside (accumulator init)
                pk[i][j][k] = Act<T>::pack2(acc[i][j][2 * k], acc[i][j][2 * k + 1]);
    if (p.relu) {                           // relu(round(x)) == round(relu(x)); uniform branch
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int k = 0; k < 8; ++k) { pk[i][j][k] = pk_max16(pk[i][j][k], 0u); sguard.see(pk[i][j][k]); }
    } else if constexpr (std::is_same<T, f16_t>::value) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int k = 0; k < 8; ++k) sguard.see_signed(pk[i][j][k]);
    }
    // ---- fused GroupNorm statistics ---------------------------------------------------------------------------------
    // sum and sum of squares per output channel over the block's valid voxels, from the ROUNDED values (what the
    // separate statistics pass would read back).  The cross-lane reduction over voxels is done by the matrix core: with
    // the packed tile as A operand (rows = voxels, k = 16 channels) and a 0/1 selector as B, D[voxel][channel] is the
    // transposed tile -- lane = channel, 16 voxels in its registers, exact in fp32 -- so each lane just sums its
    // registers.  Waves and half-waves are combined in a fixed order through LDS (deterministic), then one double
    // atomicAdd per channel and statistic per block.
    if (p.gn_sums) {
        typedef __attribute__((ext_vector_type(4))) unsigned u4;
        float* const part = reinterpret_cast<float*>(ldsA + ((gc - 1) % NA) * A_BYTES);   // free: last chunk's halo slot
        const int n32 = lane & 31;
        v8 s1, s2;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = 8 * (e >> 2) + 4 * half + (e & 3);     // channel of k-slot (half, e) within a 16-channel group
            s1[e] = (T)((c == n32) ? 1.0f : 0.0f);
            s2[e] = (T)((c + 16 == n32) ? 1.0f : 0.0f);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            float sj = 0.f, ssj = 0.f;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                u4 lo = {pk[i][j][0], pk[i][j][1], pk[i][j][2], pk[i][j][3]};
                u4 hi = {pk[i][j][4], pk[i][j][5], pk[i][j][6], pk[i][j][7]};
                unsigned zr = 0u;
                asm volatile("" : "+v"(zr));      // (keeps the eight selects inside this branch: BatchNorm nets never take it)
                if (!valid[i]) { lo = u4{zr, zr, zr, zr}; hi = lo; }
                f32x16 d;
#pragma unroll
                for (int r = 0; r < 16; ++r) d[r] = 0.f;
                d = Act<T>::mfma(__builtin_bit_cast(v8, lo), s1, d);
                d = Act<T>::mfma(__builtin_bit_cast(v8, hi), s2, d);
#pragma unroll
                for (int r = 0; r < 16; ++r) { sj += d[r]; ssj = fmaf(d[r], d[r], ssj); }
            }
            float* const q = part + ((size_t)(wave * 2 + half) * (NT * 32) + j * 32 + n32) * 2;
            q[0] = sj; q[1] = ssj;
        }
        __syncthreads();
        if (tid < NT * 64) {
            const int stat = tid / (NT * 32), cw = tid % (NT * 32);
            double t = 0.0;
            for (int w = 0; w < WAVES * 2; ++w) t += (double)part[((size_t)w * (NT * 32) + cw) * 2 + stat];
            const int ch = nb * NT * 32 + cw;
            if (ch < p.Cd)
                atomicAdd(reinterpret_cast<double*>(reinterpret_cast<char*>(p.gn_sums) + (size_t)tn * p.tstride) +
                              (size_t)stat * p.gn_C + ch, t);
        }
        __syncthreads();         // the slot is a DMA target again in the next block
    }
#ifdef SD_T5_EARLY
    SD_T(5);
#endif
    // (the pooled tensor first: its window maxima read the packed values, the row stores below may then trade them in place)
    // ---- fused MaxPool(ceil_mode): (kz,2,2) window = {the wave's two tiles (3D)} x {lane^16 (y)} x {lane^1 (x)} ----
    // Only planned behind a ReLU (sd_api.hip): all values are >= 0, so the packed integer max is the float max and
    // voxels beyond the volume contribute 0.
    if (p.pool_dst) {
        T* const pdst = reinterpret_cast<T*>(reinterpret_cast<char*>(p.pool_dst) + (size_t)tn * p.tstride);
        const bool writer = (dy == 0) && ((dxl & 1) == 0);
        bool allv = true;
#pragma unroll
        for (int i = 0; i < MT; ++i) allv = allv && valid[i];
        const bool ragged = __any(!allv);      // wave-uniform
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            if (KZ == 3 && (i & 1)) continue;   // 3D: tiles (i, i+1) form ONE pooled tile (z pair)
            const int pz = (KZ == 3) ? (z0 + tzs[i]) >> 1 : z0, py = (y0 + tys[i]) >> 1, px = (x0 + dxl) >> 1;
            const size_t po = (size_t)(pz * p.pH + py) * p.pW + px;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                unsigned m[8];
                if (GN && p.pool_dir) {      // raw outputs, GroupNorm apply deferred to the readers: pool in the order of the floats
                    const unsigned* const dq = p.pool_dir + ((size_t)(nb * NT + j) * 2 + half) * 8;
                    unsigned dk[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) dk[k] = dq[k];
                    if constexpr (std::is_same<T, f16_t>::value) {      // fp16 has a packed float maximum: negate where the minimum is wanted
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            const unsigned flip = dk[k] & 0x80008000u;
                            m[k] = valid[i] ? (pk[i][j][k] ^ flip) : 0xfc00fc00u;      // (-inf, -inf)
                            if (KZ == 3) m[k] = pk_fmax_f16(m[k], valid[i | 1] ? (pk[i | 1][j][k] ^ flip) : 0xfc00fc00u);
                        }
                        pool_xy_pk8_f16(m);
#pragma unroll
                        for (int k = 0; k < 8; ++k) m[k] ^= dk[k] & 0x80008000u;
                    } else {
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        m[k] = valid[i] ? pk_order_key(pk[i][j][k], dk[k]) : PK_KEY_LOWEST;
                        if (KZ == 3) m[k] = pk_max16(m[k], valid[i | 1] ? pk_order_key(pk[i | 1][j][k], dk[k]) : PK_KEY_LOWEST);
                    }
                    pool_xy_pk8(m);
#pragma unroll
                    for (int k = 0; k < 8; ++k) m[k] = pk_order_unkey(m[k], dk[k]);
                    }
                } else if (ragged) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        m[k] = valid[i] ? pk[i][j][k] : 0u;
                        if (KZ == 3) m[k] = pk_max16(m[k], valid[i | 1] ? pk[i | 1][j][k] : 0u);
                    }
                    pool_xy_pk8(m);
                } else {      // every voxel of the wave's tiles lies inside the volume (all but the last blocks of a ragged extent)
#pragma unroll
                    for (int k = 0; k < 8; ++k) m[k] = KZ == 3 ? pk_max16(pk[i][j][k], pk[i | 1][j][k]) : pk[i][j][k];
                    pool_xy_pk8(m);
                }
                store_tile_rows_pk<T>(m, pdst, p.Pp, po, writer && valid[i], (nb * NT + j) * 32, half, p.Cd);
            }
        }
    }

    if (p.store_main) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int i = 0; i < MT; ++i)
                store_tile_rows_pk<T>(pk[i][j], dst, p.Pd, voxoff[i], valid[i], (nb * NT + j) * 32, half, p.Cd);
    }

#ifndef SD_T5_EARLY
    SD_T(5);   // main store done
#endif
    // ---- fused conv_final (1x1x1) + softmax + uint8, on the matrix core --------------------------------------------
    // logits[class][voxel] = W[class][channel] . act[channel][voxel]: the B fragment of k-step s is exactly the pair
    // of packed output quads (2s, 2s+1) this lane already holds (the k order of an MFMA is free, the weight fragment
    // is packed in the same channel order); the fp32 weights enter as bf16/f16 hi + lo parts (two MFMAs), which
    // keeps the product sum at fp32 accuracy.  Result rows = classes: lower lanes hold classes 0-3 of their voxel
    // in registers 0-3, upper lanes classes 4-7; one half-wave swap per register then gives the lower lane all 8
    // logits of tile tp's voxel and the upper lane those of tile tp+1's voxel.
    if (p.final_wfrag) {
        const long nvox = p.final_nvox;
#pragma unroll
        for (int tp = 0; tp < MT; tp += 2) {
            f32x16 lgt[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) lgt[i][r] = 0.f;
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        typedef __attribute__((ext_vector_type(4))) unsigned u4;
                        const unsigned* q4 = &pk[tp + i][j][4 * s2];          // quads 2*s2, 2*s2 + 1
                        const v8 bfrag = __builtin_bit_cast(v8, u4{q4[0], q4[1], q4[2], q4[3]});
                        const v8 fw0 = *reinterpret_cast<const v8*>(fwl + (((j * 2 + s2) * 2 + 0) * 64 + lane) * 16);
                        const v8 fw1 = *reinterpret_cast<const v8*>(fwl + (((j * 2 + s2) * 2 + 1) * 64 + lane) * 16);
                        lgt[i] = Act<T>::mfma(fw0, bfrag, lgt[i]);
                        lgt[i] = Act<T>::mfma(fw1, bfrag, lgt[i]);
                    }
            }
            float l[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // the class bias is added BEFORE the swap by a compiler-visible VALU op: an asm statement must not be
                // the first reader of an MFMA result (hipcc pads no MFMA -> VALU hazard for inline asm consumers)
                const float bmine = wl[NT * 32 + 4 * half + e];
                unsigned a = __builtin_bit_cast(unsigned, lgt[0][e] + bmine);
                unsigned b2 = __builtin_bit_cast(unsigned, lgt[1][e] + bmine);
                swap32(a, b2);      // lower: a = own tile-tp classes 0-3, b2 = tile-tp classes 4-7 (from the upper lane)
                l[e] = __builtin_bit_cast(float, a);
                l[4 + e] = __builtin_bit_cast(float, b2);
            }
            float mx = -INFINITY;
#pragma unroll
            for (int co = 0; co < 8; ++co)
                if (co < p.final_cout) mx = fmaxf(mx, l[co]);
            float guard = 1.f;
            if (p.final_kind != SD_OUT_LOGITS_F32) {
                float sum = 0.f;
#pragma unroll
                for (int co = 0; co < 8; ++co) {
                    l[co] = co < p.final_cout ? __expf(l[co] - mx) : 0.f;
                    sum += l[co];
                }
                const float inv = 1.0f / sum;
                guard = sum;
#pragma unroll
                for (int co = 0; co < 8; ++co) l[co] *= inv;
            } else {
                guard = logit_probe<T>(l, p.final_cout);
            }
            const bool vmine = half ? valid[tp + 1] : valid[tp];
            if (vmine) range_guard<T>(guard, p.ovf);
            const size_t v = half ? voxoff[tp + 1] : voxoff[tp];
            if (vmine) {
                if (p.final_kind == SD_OUT_LABELS_U8) {
                    uint8_t lab = 0;
                    for (int k = 0; k < p.lab.n; ++k) {
                        const int id = p.lab.ids[k];
                        float pv = 0.f;
#pragma unroll
                        for (int co = 0; co < 8; ++co) pv = (co == id) ? l[co] : pv;
                        if ((int)(uint8_t)(pv * 255.f) >= p.lab.cuts[k]) lab = (uint8_t)id;
                    }
                    (reinterpret_cast<uint8_t*>(p.final_out) + (size_t)tn * p.out_tstride)[v] = lab;
                } else if (p.final_kind == SD_OUT_PROBS_U8) {
                    uint8_t* out = reinterpret_cast<uint8_t*>(p.final_out) + (size_t)tn * p.out_tstride;
#pragma unroll
                    for (int co = 0; co < 8; ++co)
                        if (co < p.final_cout) out[(size_t)co * nvox + v] = (uint8_t)(l[co] * 255.f);
                } else {
                    float* out = reinterpret_cast<float*>(reinterpret_cast<char*>(p.final_out) + (size_t)tn * p.out_tstride);
#pragma unroll
                    for (int co = 0; co < 8; ++co)
                        if (co < p.final_cout) out[(size_t)co * nvox + v] = l[co];
                }
            }
        }
    }

    }
        if constexpr (SPREAD) {
            if (wave < WAVES / 2 && !p.gn_sums) asm volatile("s_barrier" ::: "memory");     // (asymmetric epilogue, see the stage loop)
        }
        SD_T(6);   // epilogue done
#ifdef SD_TIMING
        if (tcount == SD_TB && lane == 0 && p.dbg) {
            long long* o = p.dbg + ((size_t)blockIdx.x * WAVES + wave) * 8;
            for (int i = 0; i < 8; ++i) o[i] = tstamp[i];
#ifdef SD_STAGES
            sstamp[14] = tstamp[0]; sstamp[15] = tstamp[6];
            long long* o2 = p.dbg + (1 << 20) + ((size_t)blockIdx.x * WAVES + wave) * 16;
            for (int i = 0; i < 16; ++i) o2[i] = sstamp[i];
#endif
#ifdef SD_RT
            o[4] = rt0; o[5] = __builtin_amdgcn_s_memrealtime();          // (probe) absolute 100 MHz ticks
            o[6] = ((long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
#endif
        }
#endif
        lb = nlb; z0 = nz0; y0 = ny0; x0 = nx0; tn = ntn;
    }
    if (p.clk && blockIdx.x == 0 && threadIdx.x == 0) { p.clk[2] = __builtin_readcyclecounter(); p.clk[3] = __builtin_amdgcn_s_memrealtime(); }
    sguard.flush(p.ovf);
}

// =========================================================================================================
// launch support
static inline int grid_for(long total, int per_block = 256, int cap = 256 * 16) {
    long g = (total + per_block - 1) / per_block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}
#define SD_LAUNCH_CHECK() (hipGetLastError() == hipSuccess ? SD_OK : SD_ERR_HIP)

constexpr int SD_LDS_BYTES = 160 * 1024;
constexpr int SD_NUM_CU = 256;
constexpr int SD_MAX_DEVICES = 64;
struct LaunchCache { size_t attr_set = 0, occ_lds = 0; int occ = 1; };

template <int KZ, int NT, int WAVES, int MT, int NSLOT>
static size_t conv_lds_bytes(int nstages, bool fuse_final = false) {
    using G = ConvGeo<KZ, WAVES, MT>;
    constexpr int NH = (G::BZ + KZ - 1) * (G::BY + 2) * (G::BX + 2);
    constexpr int A_BYTES = (NH * 2 + 63) / 64 * 1024;
    return (size_t)(NSLOT > 0 ? NSLOT : NSLOT < 0 ? -NSLOT : 2) * A_BYTES + (size_t)(NSLOT > 0 ? nstages : NSLOT < 0 ? -NSLOT : 2) * 9 * NT * 1024 +
           SD_CONV_PARAM_BYTES + (fuse_final ? NT * 4096 : 0) + 1024;
}

template <typename T, int KZ, int NT, int WAVES, int NSLOT, int MT = 2, int MODE = 0>
static int launch_conv_k(ConvParams p, int NB, hipStream_t s) {
    using G = ConvGeo<KZ, WAVES, MT>;
    constexpr bool FF = MODE == 1 || MODE == 4 || MODE == 5;
    const size_t lds = conv_lds_bytes<KZ, NT, WAVES, MT, NSLOT>((p.nchunk0 + p.nchunk1) * KZ, p.final_wfrag != nullptr) +
                       (FF ? (size_t)2 * (G::BY + 4) * (G::BX + 4) * 4 + (MODE == 5 ? (size_t)16 + (NSLOT == 3 ? 2 : 1) * 2048
                                                                                      : (size_t)(MODE == 1 && NSLOT == 3 ? 2 : 1) * (5 * 64 + 32) * 4) : 0) +
                       (MODE == 2 ? (size_t)p.batch * conv_gn_lds_per_tile(p.nchunk0 + p.nchunk1) : 0);
    if (lds > (size_t)SD_LDS_BYTES) return SD_ERR_INVALID;
    p.nbx = (p.W + G::BX - 1) / G::BX; p.nby = (p.H + G::BY - 1) / G::BY; p.nbz = (p.D + G::BZ - 1) / G::BZ;
    if (const char* ns = getenv("SD_TIMING_NO_STORE")) {      // (timing probe: WRONG results; 1 = the 3x3x3 forms, 2 = every form)
        if (KZ == 3 || atoi(ns) >= 2) { p.store_main = 0; p.pool_dst = nullptr; }
    }
    static const int order = getenv("SD_BLOCK_ORDER") ? atoi(getenv("SD_BLOCK_ORDER")) : 1;
    p.block_order = order;
    // per-DEVICE cache of the dynamic-LDS attribute and the occupancy answer of this instantiation (a function attribute
    // set on one device does not carry over to a model created on another one in the same process); guarded, because
    // two models may launch their first forward from different threads
    auto kern = k_conv_mfma<T, KZ, NT, WAVES, NSLOT, MT, MODE>;
    static std::mutex mu;
    static LaunchCache cache[SD_MAX_DEVICES];
    int occ = 1;
    {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= SD_MAX_DEVICES) return SD_ERR_HIP;
        std::lock_guard<std::mutex> lock(mu);
        LaunchCache& c = cache[dev];
        if (lds > c.attr_set) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds) != hipSuccess) return SD_ERR_HIP;
            c.attr_set = lds;
        }
        if (lds != c.occ_lds) {   // resident workgroups per CU for this LDS footprint (registers + LDS)
            int n = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(kern), WAVES * 64, lds) !=
                hipSuccess) n = 1;
            c.occ = std::max(1, n);
            c.occ_lds = lds;
        }
        occ = c.occ;
    }
    const int nsb = p.nbx * p.nby * p.nbz * p.batch;
    const int wg_per_cu = std::min(occ, (int)(SD_LDS_BYTES / lds));
    const int cap = std::max(8, SD_NUM_CU * wg_per_cu / NB / 8 * 8);
    static const bool no_persist = getenv("SD_NO_PERSIST") != nullptr;   // debugging aid
    dim3 grid(!no_persist ? std::min(nsb, cap) : nsb, NB), block(WAVES * 64);
    {
        static const std::string name = std::string("k_conv_mfma<") + (std::is_same<T, bf16_t>::value ? "bf16" : "f16") + "," +
            std::to_string(KZ) + "x3x3,NT=" + std::to_string(NT) + ",WAVES=" + std::to_string(WAVES) + ",NSLOT=" + std::to_string(NSLOT) +
            ",MT=" + std::to_string(MT) + ",MODE=" + std::to_string(MODE) + ">";
        SD_NOTE_KERNEL(name.c_str());
    }
    hipLaunchKernelGGL(kern, grid, block, lds, s, p);
    return SD_LAUNCH_CHECK();
}
