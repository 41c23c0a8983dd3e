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
template <typename T, int KZ, int NT, int WAVES, int NSLOT, int MT, int MODE>
__global__ __launch_bounds__(WAVES * 64, WAVES == 12 ? 3 : 2) void k_conv_mfma(const ConvParams p) {
    constexpr bool FF = MODE == 1, GN = MODE == 2;
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
    constexpr bool SPREAD = MT == 4 && KZ == 3 && WAVES == 8 && NSLOT == 0 && MODE == 0;
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
    constexpr bool HPACK_REGS = !LEAN || GN || (NT == 3 && MT == 2);
    int hpack[HPACK_REGS ? AJ : 1];
    if constexpr (HPACK_REGS) {
#pragma unroll
        for (int j = 0; j < AJ; ++j) hpack[j] = hpack_of(j);
    }
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
        sbase += (size_t)cc * Ps * (SD_CHUNK * sizeof(T)) + (size_t)tile * p.tstride;      // chunk plane of this tile
        char* dst = ldsA + slot * A_BYTES + wave * 1024;
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
                if (p.first_in_f32) v = reinterpret_cast<const float*>(in)[idx];
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
    float* const ffw = fpatch + 2 * FNP;                      // [5 k-steps][64 lanes] + [32] bias
    if constexpr (FF) {
        patch_fetch(z0, y0, x0, tn);
        patch_park(0);
        for (int i = tid; i < 5 * 64 + 32; i += WAVES * 64) ffw[i] = i < 320 ? p.first_w[i] : p.first_bias[i - 320];
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
    int gc = 0, gs = 0;   // chunk / stage counters across blocks (slot parity)
    int ph0 = 0;          // SPREAD: halo coordinates of this lane's first piece, packed z << 16 | y << 8 | x
    if constexpr (SPREAD) {
        const int hv0 = (wave * 64 + lane) >> 1;
        ph0 = ((hv0 / (HX * HY)) << 16) | (((hv0 / HX) % HY) << 8) | (hv0 % HX);
        static_assert(!SPREAD || (HX < 256 && HY < 256), "packed piece coordinates");
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
            static_assert(!FF || (KZ == 1 && WRES && NA == 2), "fused first conv: planar, resident weights");
            constexpr int PXW = HX + 2, NSTEP1 = 5;
            const float* const fp = fpatch + (round & 1) * FNP;      // parked by the prologue / during the previous block
            float w1[NSTEP1];
            int toff1[NSTEP1];
#pragma unroll
            for (int st = 0; st < NSTEP1; ++st) {
                w1[st] = ffw[st * 64 + lane];
                int tap = 2 * st + half;
                if (tap >= 9) tap = 0;                       // (its weight is zero)
                toff1[st] = (tap / 3) * PXW + (tap % 3);
            }
            f32x4 b1[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) b1[q] = *reinterpret_cast<const f32x4*>(ffw + 320 + 8 * q + 4 * half);
            for (int t = wave; t * 32 < NH; t += WAVES) {
                const int hv = t * 32 + (lane & 31);
                const int hvc = hv < NH ? hv : NH - 1;
                const int hy = hvc / HX, hx = hvc % HX;
                const int base = hy * PXW + hx;
                f32x16 a1;
#pragma unroll
                for (int r = 0; r < 16; ++r) a1[r] = 0.f;
#pragma unroll
                for (int st = 0; st < NSTEP1; ++st)
                    a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[st], fp[base + toff1[st]], a1, 0, 0, 0);
                const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
                const bool invol = (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;     // z0 < D always
                if (hv < NH) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        unsigned lo = Act<T>::pack2(a1[4 * q] + b1[q][0], a1[4 * q + 1] + b1[q][1]);
                        unsigned hi = Act<T>::pack2(a1[4 * q + 2] + b1[q][2], a1[4 * q + 3] + b1[q][3]);
                        if (p.first_relu) { lo = pk_max16(lo, 0u); hi = pk_max16(hi, 0u); }
                        sguard.see_signed(lo); sguard.see_signed(hi);
                        if (!invol) { lo = 0u; hi = 0u; }      // the second conv's zero padding
                        typedef __attribute__((ext_vector_type(2))) unsigned u2;
                        char* const slot = ldsA + ((gc + (q >> 1)) % NA) * A_BYTES;
                        *reinterpret_cast<u2*>(slot + hv * 32 + ((((q & 1) ^ (hy & 1))) << 4) + half * 8) = u2{lo, hi};
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
            const char* const abuf = ldsA + (gc % NA) * A_BYTES;
            // SPREAD: the chunk fetched during this chunk's stages (the next chunk of this block, or the first chunk of the
            // workgroup's next block); everything a piece needs is pinned in scalar registers here -- a kernel-argument
            // s_load inside the tap loop would count on lgkmcnt and break the counted LDS waits
            bool pf_real = false;
            const char* pf_sbase = nullptr;
            const char* pf_zero = reinterpret_cast<const char*>(p.zero);
            char* pf_dst = nullptr;
            int pf_Hs = 0, pf_Ws = 0, pf_z = 0, pf_y = 0, pf_x = 0, pf_D = p.D, pf_H = p.H, pf_W = p.W;
            if constexpr (SPREAD) {
                int pc = c + 1, pt = tn;
                pf_real = true; pf_z = z0; pf_y = y0; pf_x = x0;
                if (pc == nchunks) { pc = 0; pt = ntn; pf_z = nz0; pf_y = ny0; pf_x = nx0; pf_real = nlb >= 0; }
                size_t Ps;
                if (pc < p.nchunk0) { pf_sbase = (const char*)p.src0; Ps = p.P0; pf_Hs = p.H0; pf_Ws = p.W0; }
                else { pf_sbase = (const char*)p.src1; Ps = p.P1; pf_Hs = p.H1; pf_Ws = p.W1; pc -= p.nchunk0; }
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
            }
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
                const char* const acur = abuf + kz * SLICE;
                // software-pipelined over the 9 taps: the fragments of tap t+1 are in flight while tap t's MFMAs run.
                // The LDS reads and their counted waits are inline asm: left to the compiler the reads are sunk next
                // to their use behind an lgkmcnt(0) (it prefers reusing the fragment registers), which idles the
                // matrix pipe for one LDS latency per tap.  LDS returns in order, so lgkmcnt(MT + NT) after issuing
                // tap t+1 means tap t has landed; `tie` makes the MFMAs depend on the post-wait values.
                if constexpr (MT == 4) {
                    // 4 voxel tiles per wave (KZ == 3: z-neighbours, so tile i's fragment sits i * SLICE bytes behind tile 0's:
                    // a ds_read offset immediate, ONE address register pair for all tiles).  Register budget: 128
                    // accumulator registers leave no room for double-buffered fragments, so the x fragments are single-
                    // buffered and REFILLED IN PLACE -- tile i's fragment of tap t+1 is requested right behind the two
                    // MFMAs that last read tap t's; only the weight fragments are double-buffered.  LDS returns in
                    // order: whenever a tile row starts, exactly 3 + NT younger reads are in flight (3 - i of this tap's x
                    // fragments, NT weight fragments and i x fragments of the next tap) -> one constant lgkmcnt.
                    static_assert(MT != 4 || KZ == 3, "MT = 4: z-stacked tiles");
                    v8 xq[4], wq[2][NT];
                    const uint32_t bA = lds_addr(bcur);
                    const uint32_t xE = lds_addr(acur) + xoffE[0], xO = lds_addr(acur) + xoffO[0];
                    auto load_x = [&](auto tc, auto ic) {
                        constexpr int t9 = decltype(tc)::value, i = decltype(ic)::value, ky = t9 / 3, kx = t9 % 3;
                        ds_read16<(ky * HX + kx) * 32 + i * SLICE>(xq[i], (ky & 1) ? xO : xE);
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
                for (int i = 0; i < MT; ++i) { xaE[i] = aE + xoffE[i]; xaO[i] = aE + xoffO[i]; }
                auto load_tap = [&](auto tc) {
                    constexpr int t9 = decltype(tc)::value, ky = t9 / 3, kx = t9 % 3, buf = t9 & 1;
                    constexpr int tapoff = (ky * HX + kx) * 32;
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
    if constexpr (MT == 4) {
        // ---- epilogue of the 4-tile form, one z-PAIR of tiles and one 32-channel group at a time (the packed values of
        // all 8 accumulator tiles at once would not fit beside the accumulators): pack + ReLU, GroupNorm partial sums, main
        // store, fused 2x2x2 pooling of the pair.
        typedef __attribute__((ext_vector_type(4))) unsigned u4;
        T* const dst4 = reinterpret_cast<T*>(reinterpret_cast<char*>(p.dst) + (size_t)tn * p.tstride);
        const int vy = y0 + tys[0] + dy, vx = x0 + dxl, vz0 = z0 + tzs[0];
        const bool vyx = vy < p.H && vx < p.W;
        const size_t vo0 = (size_t)(vz0 * p.H + vy) * p.W + vx, vzs = (size_t)p.H * p.W;
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
                    val2[e2] = vyx && (vz0 + ip + e2) < p.D;
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
                    unsigned m[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) m[k] = pk_max16(val2[0] ? pk2[0][k] : 0u, val2[1] ? pk2[1][k] : 0u);
                    pool_xy_pk8(m);
                    const int pz = (vz0 + ip) >> 1, py = (y0 + tys[0]) >> 1, px = (x0 + dxl) >> 1;
                    const bool writer = (dy == 0) && ((dxl & 1) == 0);
                    store_tile_rows_pk<T>(m, pdst, p.Pp, (size_t)(pz * p.pH + py) * p.pW + px, writer && val2[0],
                                          (nb * NT + j) * 32, half, p.Cd);
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
        voxoff[i] = (size_t)(vz * p.H + vy) * p.W + vx;          // voxel index inside the tile's tensor
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
                if (!valid[i]) { lo = u4{0u, 0u, 0u, 0u}; hi = lo; }
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
    // ---- fused MaxPool(ceil_mode): (kz,2,2) window = {the wave's two tiles (3D)} x {lane^16 (y)} x {lane^1 (x)} ----
    // Only planned behind a ReLU (sd_api.hip): all values are >= 0, so the packed integer max is the float max and
    // voxels beyond the volume contribute 0.
    if (p.pool_dst) {
        T* const pdst = reinterpret_cast<T*>(reinterpret_cast<char*>(p.pool_dst) + (size_t)tn * p.tstride);
        const bool writer = (dy == 0) && ((dxl & 1) == 0);
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
                } else {
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        m[k] = valid[i] ? pk[i][j][k] : 0u;
                        if (KZ == 3) m[k] = pk_max16(m[k], valid[i | 1] ? pk[i | 1][j][k] : 0u);
                    }
                    pool_xy_pk8(m);
                }
                store_tile_rows_pk<T>(m, pdst, p.Pp, po, writer && valid[i], (nb * NT + j) * 32, half, p.Cd);
            }
        }
    }

    // ---- fused conv_final (1x1x1) + softmax + uint8, on the matrix core --------------------------------------------
    // logits[class][voxel] = W[class][channel] . act[channel][voxel]: the B fragment of k-step s is exactly the pair
    // of packed output quads (2s, 2s+1) this lane already holds (the k order of an MFMA is free, the weight fragment
    // is packed in the same channel order); the fp32 weights enter as bf16/f16 hi + lo parts (two MFMAs), which
    // keeps the product sum at fp32 accuracy.  Result rows = classes: lower lanes hold classes 0-3 of their voxel
    // in registers 0-3, upper lanes classes 4-7; one half-wave swap per register then gives the lower lane all 8
    // logits of tile tp's voxel and the upper lane those of tile tp+1's voxel.
    if (p.final_wfrag) {
        const long nvox = (long)p.D * p.H * p.W;
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
    sguard.flush(p.ovf);
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
    T* const dst = reinterpret_cast<T*>(reinterpret_cast<char*>(p.dst) + blockIdx.z * p.tstride);
    StoreGuard<T> sguard;
    const int ntiles = (p.Cd + 31) / 32;
    const int half = lane >> 5;
    const size_t P = (size_t)p.D * p.H * p.W;
    for (int nt = 0; nt < ntiles; ++nt) {
        float wf[NSTEP];
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) wf[s] = p.wpack[(nt * NSTEP + s) * 64 + lane];
        f32x4 bq[4];                  // folded bias of this lane's 16 channels (added AFTER the tap chain, like the oracle)
#pragma unroll
        for (int q = 0; q < 4; ++q) bq[q] = *reinterpret_cast<const f32x4*>(p.bias + nt * 32 + 8 * q + 4 * half);
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
            // packed epilogue: one convert and one integer max per pair (relu(round(x)) == round(relu(x)))
            unsigned pk[8];
#pragma unroll
            for (int k = 0; k < 8; ++k)
                pk[k] = Act<T>::pack2(acc[2 * k] + bq[k >> 1][(2 * k) & 3], acc[2 * k + 1] + bq[k >> 1][(2 * k + 1) & 3]);
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
template <typename T, bool GN>
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
#pragma unroll 2
    for (int c = 0; c < p.nchunk; ++c) {
        v8 xf[2], wf[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            v8 val = {};
            if (mv[i]) {
                val = *reinterpret_cast<const v8*>(src + ((size_t)c * M + m[i]) * SD_CHUNK + (lane >> 5) * 8);
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
    const int H2 = 2 * p.H, W2 = 2 * p.W;
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
                unsigned d[2][2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int q = 2 * qp + h;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = acc[i][j][4 * q + e];
                        if (p.relu) v[e] = fmaxf(v[e], 0.f);
                    }
                    d[h][0] = Act<T>::pack2(v[0], v[1]);
                    d[h][1] = Act<T>::pack2(v[2], v[3]);
                    sguard.see_signed(d[h][0]); sguard.see_signed(d[h][1]);
                }
                swap32(d[0][0], d[1][0]);
                swap32(d[0][1], d[1][1]);
                const int n8 = nbase + 8 * (2 * qp + half);       // first of this lane's 8 consecutive channels
                if (mv[i] && n8 < p.ntot) {
                    const int tap = n8 / p.Cd, co = n8 - tap * p.Cd;
                    const int a = (p.kz == 2) ? (tap >> 2) : 0, b = (tap >> 1) & 1, cx = tap & 1;
                    const size_t vo = ((size_t)(z * p.kz + a) * H2 + (2 * y + b)) * W2 + (2 * x + cx);
                    typedef __attribute__((ext_vector_type(4))) unsigned u4;
                    *reinterpret_cast<u4*>(dst + ((size_t)(co >> 4) * p.Pd + vo) * SD_CHUNK + (co & 15)) = u4{d[0][0], d[0][1], d[1][0], d[1][1]};
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
template <typename T, int NCH, int NTAB, bool WL, bool GN>
__global__ __launch_bounds__(256) void k_upconv_rows(const UpconvParams p) {
    using v8 = typename Act<T>::v8;
    using v4 = typename Act<T>::v4;
    typedef __attribute__((ext_vector_type(4))) unsigned u4;
    constexpr int CD = NTAB * 16;            // channel stride of the output
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
    const int H2 = 2 * p.H, W2 = 2 * p.W;
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
    const int a_wg = WL ? (int)blockIdx.y : 0;
    if constexpr (WL) {
        const int first_blk = (a_wg * NTAB) >> 1, last_blk = (a_wg * NTAB + NTAB - 1) >> 1;
        const int nbytes = (last_blk - first_blk + 1) * NCH * 2048;
        const char* const wsrc = reinterpret_cast<const char*>(wp) + (size_t)first_blk * NCH * 2048;
        for (int o = tid * 16; o < nbytes; o += 256 * 16)
            *reinterpret_cast<u4*>(smem + 4 * 32 * ROW + o) = *reinterpret_cast<const u4*>(wsrc + o);
        __syncthreads();
    }
    for (long m0 = ((long)blockIdx.x * 4 + wave) * 32; m0 < M; m0 += WL ? (long)gridDim.x * 128 : M) {
    const long m = m0 + vl;
    const bool mv = m < M;
    v8 xf[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        v8 val = {};
        if (mv) val = *reinterpret_cast<const v8*>(src + ((size_t)c * M + m) * SD_CHUNK + half * 8);
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
        f32x16 acc[NTAB];
#pragma unroll
        for (int j = 0; j < NTAB; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + ab * 2 * CD + 32 * j + 8 * q + 4 * half);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[j][4 * q + e] = b[e];
            }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
#pragma unroll
            for (int j = 0; j < NTAB; ++j) {
                const int tl = ab * NTAB + j;             // global 32-column tile (n = tap*CD + co ordering)
                v8 wf;
                if constexpr (WL) {
                    const int tloc = ab * NTAB + j - 2 * ((ab0 * NTAB) >> 1);
                    wf = *reinterpret_cast<const v8*>(wlds + ((((tloc >> 1) * NCH + c) * 2 + (tloc & 1)) * 64 + lane) * 16);
                } else {
                    wf = *reinterpret_cast<const v8*>(wp + ((((size_t)(tl >> 1) * NCH + c) * 2 + (tl & 1)) * 64 + lane) * 8);
                }
                acc[j] = Act<T>::mfma(wf, xf[c], acc[j]);
            }
        }
        // accumulators -> rounded rows in the wave's LDS tile: voxel vl, columns 32j + 8q + 4*half + e
#pragma unroll
        for (int j = 0; j < NTAB; ++j) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                v4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[j][4 * q + e];
                    if (p.relu) v = fmaxf(v, 0.f);
                    o[e] = (T)v;
                }
                {
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
                *reinterpret_cast<u4*>(dst + (((size_t)chunk * p.Pd + ov) * SD_CHUNK + hf * 8) * sizeof(T)) = val;
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
#pragma unroll 2
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
    constexpr bool FF = MODE == 1;
    const size_t lds = conv_lds_bytes<KZ, NT, WAVES, MT, NSLOT>((p.nchunk0 + p.nchunk1) * KZ, p.final_wfrag != nullptr) +
                       (FF ? (size_t)2 * (G::BY + 4) * (G::BX + 4) * 4 + (5 * 64 + 32) * 4 : 0) +
                       (MODE == 2 ? (size_t)p.batch * conv_gn_lds_per_tile(p.nchunk0 + p.nchunk1) : 0);
    if (lds > (size_t)SD_LDS_BYTES) return SD_ERR_INVALID;
    p.nbx = (p.W + G::BX - 1) / G::BX; p.nby = (p.H + G::BY - 1) / G::BY; p.nbz = (p.D + G::BZ - 1) / G::BZ;
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
    hipLaunchKernelGGL(kern, grid, block, lds, s, p);
    return SD_LAUNCH_CHECK();
}

bool conv_can_fuse_first(int KZ, int NT, int NB, long vox, int nstages, bool fused_final) {
    if (KZ != 1 || NT > 2 || nstages != 2) return false;
    if ((vox / 512) * NB < 512) return false;                                   // the `big` rule of launch_conv_knt
    const size_t lds = NT == 1 ? conv_lds_bytes<1, 1, 8, 2, 2>(nstages, fused_final) : conv_lds_bytes<1, 2, 8, 2, 2>(nstages, fused_final);
    return lds + 2 * 36 * 20 * 4 + 352 * 4 <= 96 * 1024;
}

template <typename T, int KZ, int NT>
static int launch_conv_knt(const ConvParams& p, int NB, hipStream_t s) {
    const long vox = (long)p.D * p.H * p.W * (p.batch_total > 0 ? p.batch_total : p.batch);      // all tiles of a batched launch set
    const int nstages = (p.nchunk0 + p.nchunk1) * KZ;
    // 512-voxel workgroups when they still give every CU work, else 256-voxel workgroups
    const bool big = (vox / 512) * NB >= 512;
    // Resident weights + persistent blocks where the whole layer's weights fit beside a 2-slot halo ring and two
    // workgroups still share a CU (level-0 layers); else streamed weights, one block per workgroup.  Deeper rings
    // (NSLOT 4/6, one workgroup per CU) were measured SLOWER on the level-0 layers (1.32 vs 1.23 ms per tile): those
    // layers are bound by per-wave instruction latency, not by bytes in flight, so resident waves win over ring depth.
    // (NT = 1 layers with 4 voxel tiles per wave -- 8 waves x 1024 voxels or 4 waves x 512 voxels -- measured SLOWER than the
    // 2-tile form, op19 54 -> 59 us per tile: these layers live on resident waves per CU, not on LDS reads per MFMA)
    if constexpr (KZ == 1 && NT <= 2) {
        if (p.first_in) {
            if (!conv_can_fuse_first(KZ, NT, NB, vox, nstages, p.final_wfrag != nullptr)) return SD_ERR_INVALID;
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
            if (conv_lds_bytes<KZ, NT, 8, 2, 2>(nstages, p.final_wfrag != nullptr) + gl <= 80 * 1024) return launch_conv_k<T, KZ, NT, 8, 2, 2, 2>(p, NB, s);
            return launch_conv_k<T, KZ, NT, 8, 0, 2, 2>(p, NB, s);
        }
        if (conv_lds_bytes<KZ, NT, 4, 2, 2>(nstages, p.final_wfrag != nullptr) + gl <= 80 * 1024) return launch_conv_k<T, KZ, NT, 4, 2, 2, 2>(p, NB, s);
        return launch_conv_k<T, KZ, NT, 4, 0, 2, 2>(p, NB, s);
    }
    if (big) {
        if (conv_lds_bytes<KZ, NT, 8, 2, 2>(nstages, p.final_wfrag != nullptr) <= 96 * 1024) return launch_conv_k<T, KZ, NT, 8, 2>(p, NB, s);
        if constexpr (KZ == 3 && NT == 2) {
            // 8x8x16 blocks, 4 z-stacked voxel tiles per wave: 0.75 LDS fragment reads per MFMA instead of 1.0, half as many
            // stage barriers and block boundaries per MFMA (32->64 channels 52.6 -> 48.7 us per tile, 64->64 88 -> 85 us,
            // 128->64 164 -> 162 us).  Taller blocks waste more on a ragged z extent, hence the rule on D.
            const bool mt2 = getenv("SD_MT2") != nullptr;      // A/B switch (read per launch): the 4x8x16 / 2-tile form everywhere
            if (!mt2 && !p.final_wfrag && (vox / 1024) * NB >= 256 && (p.D % 8 == 0 || p.D >= 96))
                return launch_conv_k<T, KZ, NT, 8, 0, 4>(p, NB, s);
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
    return act_dtype == SD_BF16 ? launch_conv2_t<bf16_t>(p, KZ, NT, NB, s) : launch_conv2_t<f16_t>(p, KZ, NT, NB, s);
}

template <typename T, typename IN>
static int launch_first_t(const FirstParams& p, int KZ, hipStream_t s) {
    dim3 grid(p.nbx * p.nby * p.nbz, 1, p.batch), block(256);
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

template <typename T, int NCH, int NTAB>
static int launch_upconv_rows(const UpconvParams& p, hipStream_t s) {
    const long M = (long)p.D * p.H * p.W;
    if (M >= (1l << 31)) return SD_ERR_INVALID;        // the kernel decodes input voxel indices with 32-bit arithmetic
    dim3 grid((unsigned)((M + 127) / 128), 1, p.batch), block(256);
    if (p.gn) hipLaunchKernelGGL((k_upconv_rows<T, NCH, NTAB, false, true>), grid, block, 4 * 32 * 64 * NTAB, s, p);
    else hipLaunchKernelGGL((k_upconv_rows<T, NCH, NTAB, false, false>), grid, block, 4 * 32 * 64 * NTAB, s, p);
    return SD_LAUNCH_CHECK();
}
template <typename T, int NCH, int NTAB>
static int launch_upconv_rows_wl(const UpconvParams& p, hipStream_t s) {      // LDS-resident weights, persistent
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
    // the store-bound full-resolution shapes get the row-coalescing kernel (64 -> 32 channels: 77 -> 52 us at 128^3);
    // at 128 -> 64 channels it needs its weights in LDS (plain rows kernel 85 us, k_upconv_mfma 63 us, LDS weights with
    // one tap pair per workgroup and two workgroups per CU 51 us; 48 -> 33 us per tile at 8 tiles per launch)
    if (p.nchunk == 4 && p.Cd == 32) return launch_upconv_rows<T, 4, 2>(p, s);
    if (p.nchunk == 3 && p.Cd == 32) return launch_upconv_rows<T, 3, 2>(p, s);
    if (p.nchunk == 2 && p.Cd == 16) return launch_upconv_rows<T, 2, 1>(p, s);
    static const bool no_wl = getenv("SD_NO_UPCONV_WL") != nullptr;
    if (p.nchunk == 8 && p.Cd == 64 && !no_wl) return launch_upconv_rows_wl<T, 8, 4>(p, s);
    // (192 -> 96 channels: k_upconv_mfma 76 us, LDS-weight rows kernel 83 us per tile in the channel-blocked layout)
    if (p.nchunk == 6 && p.Cd == 48) return no_wl ? launch_upconv_rows<T, 6, 3>(p, s) : launch_upconv_rows_wl<T, 6, 3>(p, s);
    // 192 -> 96 channels: the generic kernel re-reads the input for each of its 12 column blocks (PMC: 1.0 GB read for 0.2 GB
    // algorithmic per 8 tiles) and writes partial lines; rows kernel with LDS-resident weights 70 -> 60 us per tile
    if (p.nchunk == 12 && p.Cd == 96 && !no_wl) return launch_upconv_rows_wl<T, 12, 6>(p, s);
    const long M = (long)p.D * p.H * p.W;
    if (M >= (1l << 31)) return SD_ERR_INVALID;        // (32-bit voxel decode in the kernel)
    dim3 grid((unsigned)((M + 255) / 256), NB, p.batch), block(256);
    if (p.gn) hipLaunchKernelGGL((k_upconv_mfma<T, true>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((k_upconv_mfma<T, false>), grid, block, 0, s, p);
    return SD_LAUNCH_CHECK();
}
int launch_upconv(const UpconvParams& p, int act_dtype, int NB, hipStream_t s) {
    if (!getenv("SD_UPCONV_OLD") || p.gn)
        return act_dtype == SD_BF16 ? launch_upconv_t<bf16_t>(p, NB, s) : launch_upconv_t<f16_t>(p, NB, s);
    const long M = (long)p.D * p.H * p.W;
    dim3 grid((unsigned)((M + 255) / 256), NB, p.batch), block(256);
    if (act_dtype == SD_BF16) hipLaunchKernelGGL((k_upconv_mfma<bf16_t, false>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((k_upconv_mfma<f16_t, false>), grid, block, 0, s, p);
    return SD_LAUNCH_CHECK();
}

int launch_pool(const PoolParams& p, int act_dtype, hipStream_t s) {
    const long total = (long)p.Do * p.Ho * p.Wo * (p.C / 8);
    if (total >= (1l << 32)) return SD_ERR_INVALID;       // (32-bit element decode in the kernel)
    dim3 grid(grid_for(total), 1, p.batch), block(256);
    if (act_dtype == SD_BF16) hipLaunchKernelGGL((k_maxpool<bf16_t>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((k_maxpool<f16_t>), grid, block, 0, s, p);
    return SD_LAUNCH_CHECK();
}

int launch_final(const FinalParams& p, int act_dtype, hipStream_t s) {
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

int launch_groupnorm(const GnParams& p, int act_dtype, hipStream_t s) {
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
    if (act_dtype == SD_BF16)
        hipLaunchKernelGGL((k_read_buffer<bf16_t>), grid, block, 0, s, (const bf16_t*)buf, C, Cs, nvox, out);
    else
        hipLaunchKernelGGL((k_read_buffer<f16_t>), grid, block, 0, s, (const f16_t*)buf, C, Cs, nvox, out);
    return SD_LAUNCH_CHECK();
}
