// Level-0 decoder of a planar-top U-Net as ONE streaming kernel (gfx950):
//
//     ConvTranspose(1x2x2, 64 -> 32)+BN+ReLU  ->  cat(up, skip)  ->  Conv 1x3x3 (64 -> 32)+BN+ReLU
//         ->  Conv 1x3x3 (32 -> 32)+BN+ReLU  ->  conv_final 1x1x1 + softmax + uint8 / label rule
//
// (the reference: elektronn3 UpConv at level 0 + conv_final, called from /root/reference/syconn/handler/prediction.py:777-868
// through Predictor.predict).  As separate launches these layers move 4 full-resolution 32/64-channel tensors through HBM
// (up-convolved tensor written + read, merge-conv output written + read); here only the skip tensor and the level-1 tensor
// are read and only the uint8 result is written.
//
// All three layers are planar, so nothing couples z-planes.  A workgroup owns (tile, x-strip of 64 columns, range of
// z-planes) and MARCHES down the rows: every tensor of the chain is a 1-D stream of "positions"
//     q = row * 68 + xx,   row = plane * HP + y,   xx = x - (x0 - 2)        (68 = 64 + 2 halo columns per conv and side)
// so that a conv tap (dy, dx) is the constant stream offset dy*68 + dx, a 32-position MFMA column tile is any 32
// consecutive positions (rows need not align with tiles; the columns that wrap around a row end are the halo columns,
// whose results are never used), and the only recomputation is the x halo (68/64).  Planes are separated by one or two
// zero rows (HP = 2*(H/2 + 1) >= H + 1), which are at the same time the conv zero padding in y.  Each stream lives in an
// LDS ring: U (up-convolved, produced by MFMA), S (skip, LDS-DMA), C1 (merge-conv output) -- 32-byte records per
// (chunk of 16 channels, position), the two 16-byte halves swapped where bit 3 of the ring index is set: every
// ds_read_b128 of 32 consecutive records is bank-conflict free for ANY start position.
//
// One step = 128 positions, one barrier.  Wave specialisation (8 waves, 2 per SIMD; the matrix pipe is per SIMD, so each
// SIMD gets one wave of each role and the same MFMA count):
//   waves 0-3 "merge": issue the LDS-DMA of the step (skip pieces 3 steps ahead, level-1 pieces 5 up-conv tiles ahead),
//        then the merge conv of one 32-position tile: 36 MFMAs, ALL 36 weight fragments resident in registers.
//   waves 4-7 "tail": one output parity of the up-convolution of a 32-position level-1 tile (4 MFMAs, 3 steps ahead),
//        then the second conv of one tile two steps behind the merge conv (18 MFMAs, weights in registers), the final
//        1x1x1 on the matrix core (hi + lo weight parts), softmax, uint8 / labels, global store.
// The only LDS fragment traffic is one activation fragment per MFMA; no weight fragment is ever re-read.
// Every output is summed in the order of the layer-wise kernels (bias, chunks in concat order, taps 0..8; the up-convolution
// kernels start from the bias too): bit-identical to the layer-by-layer plan.
#include "sd_internal.h"
#include "../../include/syconn_dense.h"
#include "sd_device.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <mutex>

namespace {

template <typename X> struct TypeTag { using type = X; };
constexpr int PW_WIDE = 68;     // positions per stream row: 64 columns + 2 halo columns per conv and side (template parameter PW of k_dec0;
constexpr int PW_NARROW = 60;   // 56 + 4 for widths that 64-column strips cover badly, e.g. 331 = 6 x 56 - 5 instead of 6 x 64 - 53)
constexpr int ST = 128;         // positions per step
constexpr int RU = 768;         // ring sizes in positions (multiples of 32; see the header of k_dec0 for the spans)
constexpr int RS = 640;
constexpr int RC = 512;
constexpr int QOFF = 7680;      // multiple of every ring size, added before a modulo so that logical positions < 0 work
// LDS map.  S ring at 0: fragment addresses wrap with min(a, a - ring bytes); U ring right behind it so that its chunk-0 plane
// ends at 64 KiB (wrap = bit 16); constants; C1 ring (power of two) aligned to its chunk-plane size (wrap = and/or);
// level-1 tile slots.  All chunk-1 planes are reached with a ds_read offset immediate (< 64 KiB).
constexpr int LDS_S = 0, LDS_U = LDS_S + 2 * RS * 32, LDS_CST = LDS_U + 2 * RU * 32, LDS_FW = LDS_CST + 512;
constexpr int LDS_C = LDS_CST + 8192, LDS_L = LDS_C + 2 * RC * 32, LDS_TOTAL = LDS_L + 4 * 4 * 1024;
static_assert(LDS_U + RU * 32 == 65536 && LDS_C % (RC * 32) == 0 && LDS_FW + 4096 <= LDS_C, "LDS map");
constexpr int PF = 6;           // activation fragments in flight per wave

// byte offset of (ring index, logical 16-byte half) inside a chunk plane of a ring
__device__ __forceinline__ uint32_t rec_off(uint32_t idx, uint32_t half) { return (idx << 5) | ((half ^ ((idx >> 3) & 1u)) << 4); }
__device__ __forceinline__ uint32_t wrap(uint32_t v, uint32_t ring) { return min(v, v - ring); }      // v in [0, 2*ring)

}  // namespace

// Spans (t = 128k = first position of the merge conv's step k):
//   merge conv(k)  writes C1 [t, t+128), reads U and S in [t-69, t+197)
//   second conv(k) computes [t-256, t-128), reads C1 in [t-325, t-59)                      -> C1 ring >= 453 (512)
//   skip DMA(k)    writes S [t+384, t+512), awaited at the end of step k+1                 -> S ring >= 581 (640)
//   up-conv(k)     processes level-1 positions [32(k+3), 32(k+3)+32) -> U rows up to position t+644; everything below
//                  136*floor(32(k+3)/34) >= t+248 is then complete, which covers t+128+197 of step k+1   -> U ring >= 713 (768)
//   level-1 DMA(k) fetches the tile of up-conv(k+2), awaited at the end of step k+1       -> 4 tile slots
// A wave can issue one instruction per 4 cycles, so the per-step instruction count of each role is what has to stay
// below the ~2000 cycles the 62 MFMAs of a SIMD take: positions are tracked per lane as (xx, y, plane) cursors that
// advance by additions and compares (no division in the loop), fragment addresses advance by a constant and a wrap, and the
// final layer / softmax / store runs once per TWO steps on both half-waves (lower lanes: even step's tile, upper: odd).
// KIND: an sd_out_kind (SD_OUT_LABELS_U8: label rule from the per-class table) or 4 = labels with the generic id list
// VIEW: the launch computes a sub-box of the tile (sd_model_set_roi): D, H, W, H1, W1 are the box, sH, sW, sH1, sW1 the y / x extents of
// the tensors it lies in (the bases are shifted by the host); zero padding at a box border that is not the tensor's only reaches the
// two outermost voxel shells, which the host keeps outside what it needs.
// PW: positions per stream row = strip width + 4 (68 in the text above; every span there scales with it: a tap reaches PW + 1 positions,
// a level-1 row of PW / 2 positions yields 2 * PW positions, up-conv tile u starts at or behind position 128u - 2 * PW).
template <typename T, int KIND, bool VIEW = false, int PW = PW_WIDE>
__global__ __launch_bounds__(512, 2) void k_dec0(const Dec0Params p) {
    static_assert(PW % 4 == 0 && PW <= PW_WIDE && PW >= 36, "ring sizes are laid out for rows of at most 68 positions");
    using v8 = typename Act<T>::v8;
    typedef __attribute__((ext_vector_type(4))) unsigned u4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const cst = reinterpret_cast<float*>(smem + LDS_CST);     // [0,32) up bias, [32,64) merge, [64,96) second, [96,104) classes
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    int b = blockIdx.x;
    const int zg = b % p.nzg; b /= p.nzg;
    const int strip = b % p.nstrip, tile = b / p.nstrip;
    const int z0 = zg * p.zsplit, nz = min(p.zsplit, p.D - z0);
    const int c0 = strip * (PW - 4);
    const int HP = p.HP, HP1 = p.HP1;
    const int PT = nz * HP * PW;
    const int nsteps = ((PT + ST - 1) / ST + 1) / 2 * 2 + 2;      // second conv runs in steps 2 .. nsteps-1: an even count
    const char* const skip = reinterpret_cast<const char*>(p.skip) + (size_t)tile * p.tstride;
    const char* const lvl1 = reinterpret_cast<const char*>(p.l1) + (size_t)tile * p.tstride;
    // Scalars used inside the step loops are pinned in SGPRs: left as kernel-argument reads the compiler re-loads them at
    // their uses (s_load + s_waitcnt lgkmcnt(0), ~200 cycles each) once it runs out of scalar registers.
    int gH = p.H, gW = p.W, gH1 = p.H1, gW1 = p.W1, gcout = p.final_cout;
    asm volatile("" : "+s"(gH), "+s"(gW), "+s"(gH1), "+s"(gW1), "+s"(gcout));
    int sH = VIEW ? p.sH : gH, sW = VIEW ? p.sW : gW, sH1 = VIEW ? p.sH1 : gH1, sW1 = VIEW ? p.sW1 : gW1;      // tensor extents (strides)
    if constexpr (VIEW) asm volatile("" : "+s"(sH), "+s"(sW), "+s"(sH1), "+s"(sW1));

    for (int i = tid; i < LDS_L / 16; i += 512) reinterpret_cast<u4*>(smem)[i] = u4{0u, 0u, 0u, 0u};     // rings (and constants)
    __syncthreads();
    if (tid < 32) { cst[tid] = p.bup[tid]; cst[32 + tid] = p.b1[tid]; cst[64 + tid] = p.b2[tid]; }
    if (tid < 8) {
        // classes beyond final_cout get a -inf logit (zero weights + this bias): they vanish from the maximum and the softmax sum
        // without any per-class select; label table: cut (never reached when the class is not listed) and (list position << 8 | class)
        cst[96 + tid] = tid < p.final_cout ? p.fb[tid] : -INFINITY;
        const unsigned e = tid < p.final_cout ? p.lab_cls[tid] : 0u;
        reinterpret_cast<unsigned*>(cst)[104 + tid] = e ? (e & 0xffffu) : 0x7fffffffu;
        reinterpret_cast<unsigned*>(cst)[112 + tid] = ((e >> 16) << 8) | (unsigned)tid;
    }
    for (int i = tid; i < 256; i += 512) reinterpret_cast<u4*>(smem + LDS_FW)[i] = reinterpret_cast<const u4*>(p.fw)[i];
    __syncthreads();

    // per-lane cursor of a stream position q = (plane*HP + y)*W' + xx (W' = 68 full-resolution / 34 level-1 positions per row);
    // positions before the stream start have plane < 0
    struct Cur { int xx, y, plane; };
    auto cur_at = [](int q, int roww, int rows) {
        int r = q >= 0 ? q / roww : -((-q + roww - 1) / roww);
        Cur c;
        c.xx = q - r * roww;
        c.plane = r >= 0 ? r / rows : -((-r + rows - 1) / rows);
        c.y = r - c.plane * rows;
        return c;
    };
    // advance by `dxx` columns + `dy` rows (0 <= dxx < roww, one wrap each: needs rows > dy + 1)
    auto cur_adv = [](Cur& c, int dxx, int dy, int roww, int rows) {
        c.xx += dxx; c.y += dy;
        if (c.xx >= roww) { c.xx -= roww; c.y += 1; }
        if (c.y >= rows) { c.y -= rows; c.plane += 1; }
    };
    auto bias_init = [&](int base) {
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 bq = *reinterpret_cast<const f32x4*>(cst + base + 8 * q + 4 * half);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[4 * q + e] = bq[e];
        }
        return acc;
    };
    // acc rounded + ReLU, zeroed where !ok, quads traded with lane^32; the lane's record (chunk = half, both 16-byte halves)
    // goes to LDS byte address `rec` (physical half order given by `sw` = 0 / 16)
    StoreGuard<T> sguard;      // fp16 range guard over every value this lane rounds to the storage type (sd_device.h)
    auto write_tile = [&](const f32x16& acc, bool ok, uint32_t rec, uint32_t sw) {
        unsigned pk[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            pk[k] = pk_max16(Act<T>::pack2(acc[2 * k], acc[2 * k + 1]), 0u);
            sguard.see(pk[k]);
            if (!ok) pk[k] = 0u;
        }
        swap32x4(pk[0], pk[4], pk[1], pk[5], pk[2], pk[6], pk[3], pk[7]);
        *reinterpret_cast<u4*>(smem + (rec + sw)) = u4{pk[0], pk[1], pk[4], pk[5]};
        *reinterpret_cast<u4*>(smem + (rec + (sw ^ 16u))) = u4{pk[2], pk[3], pk[6], pk[7]};
    };

#ifdef SD_DEC0_TIMING
    long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int tsk = -100;
#define D0_T(i) do { if (k == SD_DEC0_TIMING) ts[i] = __builtin_readcyclecounter(); } while (0)
#define D0_DUMP() do { if (p.dbg && lane == 0 && blockIdx.x < 64) for (int i = 0; i < 8; ++i) p.dbg[((size_t)blockIdx.x * 8 + wave) * 8 + i] = ts[i]; } while (0)
#else
#define D0_T(i) do {} while (0)
#define D0_DUMP() do {} while (0)
#endif
    // static issue priority of one role (A/B: SD_DEC0_PRIO = 1 tail waves, 2 merge waves, 3 tail waves at level 3)
    if (p.prio == 1 && wave >= 4) asm volatile("s_setprio 1");
    if (p.prio == 2 && wave < 4) asm volatile("s_setprio 1");
    if (p.prio == 3 && wave >= 4) asm volatile("s_setprio 3");
    if (wave < 4) {
        // ------------------------------------------------------------------------------------------- merge waves
        const int cw = wave;
        v8 w1[36];
#pragma unroll
        for (int i = 0; i < 36; ++i) w1[i] = reinterpret_cast<const v8*>(p.w1)[i * 64 + lane];
        const int dslot = lane & 1, dh = dslot ^ ((lane >> 4) & 1);      // DMA lane -> (record slot, logical half)
        // byte addresses of the 9 tap fragments (chunk 0) of this wave's tile: position 128k + 32cw - 69 + tap offset + lane.
        // Advancing a step adds 128 positions = 4096 bytes; neither that nor a ring wrap changes bit 3 of the ring index, so
        // the half swizzle stays as it is.
        uint32_t aU[9], aS[9];
#pragma unroll
        for (int t9 = 0; t9 < 9; ++t9) {
            const int off = (t9 / 3) * PW + (t9 % 3);
            aU[t9] = LDS_U + rec_off(wrap((uint32_t)(QOFF + 32 * cw - (PW + 1)) % RU + off + l31, RU), half);
            aS[t9] = LDS_S + rec_off(wrap((uint32_t)(QOFF + 32 * cw - (PW + 1)) % RS + off + l31, RS), half);
        }
        uint32_t dS = (uint32_t)(QOFF + ST * (-5 + 3) + 32 * cw) % RS;          // ring index of the skip piece (uniform)
        Cur cs = cur_at(ST * (-5 + 3) + 32 * cw + (lane >> 1), PW, HP);        // ... its position handled by this lane
        Cur cl = cur_at(32 * (-5 + 5) + (lane >> 1), PW / 2, HP1);             // level-1 position of this lane's DMA
        Cur cm = cur_at(32 * cw + l31, PW, HP);                                 // merge-conv output position (k = 0)
        uint32_t cC = LDS_C + half * (RC * 32) + (((uint32_t)(32 * cw + l31 + QOFF) & (RC - 1)) << 5);   // ... its C1 record
        const uint32_t cCsw = ((((uint32_t)(32 * cw + l31 + QOFF)) >> 3) & 1u) << 4;
        const size_t chunk_l1 = (size_t)cw * p.Pl1 * 32;
        const f32x16 bias_c1 = bias_init(32);
        for (int k = -5; k < nsteps; ++k) {
            D0_T(0);
            {   // skip pieces (tile cw of step k+3, both chunks)
                const int x = c0 - 2 + cs.xx;
                const char* s0 = reinterpret_cast<const char*>(p.zero);
                const char* s1 = s0;
                if ((unsigned)cs.plane < (unsigned)nz && cs.y < gH && (unsigned)x < (unsigned)gW) {
                    const unsigned row = __umul24(z0 + cs.plane, sH) + cs.y;
                    s0 = skip + (((size_t)__umul24(row, sW) + x) * 32 + dh * 16);
                    s1 = s0 + p.Ps * 32;
                }
                glds16(s0, smem + LDS_S + dS * 32);
                glds16(s1, smem + LDS_S + RS * 32 + dS * 32);
                dS += ST; if (dS >= RS) dS -= RS;
                cur_adv(cs, ST % PW, ST / PW, PW, HP);
            }
            {   // level-1 piece: chunk cw of up-conv tile u = k + 5
                const int u = k + 5;
                const int x1 = (c0 >> 1) - 1 + cl.xx;
                const char* s0 = reinterpret_cast<const char*>(p.zero);
                if ((unsigned)cl.plane < (unsigned)nz && cl.y < gH1 && (unsigned)x1 < (unsigned)gW1) {
                    const unsigned row = __umul24(z0 + cl.plane, sH1) + cl.y;
                    s0 = lvl1 + chunk_l1 + (((size_t)__umul24(row, sW1) + x1) * 32 + dh * 16);
                }
                glds16(s0, smem + LDS_L + ((u & 3) * 4 + cw) * 1024);
                cur_adv(cl, 32 % (PW / 2), 32 / (PW / 2), PW / 2, HP1);
            }
            D0_T(1);
#ifdef SD_PROBE_DEC0_NO_MERGE
            if (false) {
#else
            if (k >= 0) {
#endif
                f32x16 acc = bias_c1;
                // 36 fragments (chunks U0, U1, S0, S1 x 9 taps: the summation order of k_conv_mfma) through PF registers: fragment
                // i + PF is requested right behind the MFMA that read fragment i's register; LDS returns in order -> counted waits
                // (one per pair of MFMAs).  A tap's address is advanced to the next step (+128 positions, ring wrap) right behind
                // its last read, so that this VALU work issues in the shadow of the MFMAs.  (Two independent accumulator chains
                // were measured: no gain -- the partner wave fills the pipe -- and they cost 8 packed adds per tile.)
                v8 xq[PF];
                auto issue = [&](auto ic) {
                    constexpr int i = decltype(ic)::value, c = i / 9, t9 = i % 9;
                    if constexpr (c < 2) {
                        ds_read16<c * (RU * 32)>(xq[i % PF], aU[t9]);
                        if constexpr (c == 1) { const uint32_t a = aU[t9] + ST * 32; aU[t9] = a - (a >> 16) * (RU * 32); }
                    } else {
                        ds_read16<(c - 2) * (RS * 32)>(xq[i % PF], aS[t9]);
                        if constexpr (c == 3) { const uint32_t a = aS[t9] + ST * 32; aS[t9] = min(a, a - RS * 32); }
                    }
                };
                static_for<PF>([&](auto ic) { issue(ic); });
                static_for<36>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    if constexpr ((i & 1) == 0) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(i + PF <= 34 ? PF - 2 : (i + 2 <= 36 ? 34 - i : 0)));
                    tie(xq[i % PF]);
                    acc = Act<T>::mfma(w1[i], xq[i % PF], acc);
                    if constexpr (i + PF < 36) issue(std::integral_constant<int, i + PF>{});
                });
                D0_T(2);
                const bool ok = (unsigned)cm.plane < (unsigned)nz && cm.y < gH && (unsigned)(c0 - 2 + cm.xx) < (unsigned)gW;
                write_tile(acc, ok, cC, cCsw);
                cC = (cC & ~(uint32_t)(RC * 32 - 1)) | ((cC + ST * 32) & (RC * 32 - 1));
                cur_adv(cm, ST % PW, ST / PW, PW, HP);
            }
            D0_T(3);
            // everything this wave DMA'd in the PREVIOUS step has landed (3 instructions per step, in order)
            asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
            D0_T(4);
            asm volatile("s_barrier" ::: "memory");
            D0_T(5);
        }
        D0_DUMP();
        sguard.flush(p.ovf);
    } else {
        // -------------------------------------------------------------------------------------------- tail waves
        const int ow = wave - 4, py = ow >> 1, px = ow & 1;
        v8 w2[18], wu[4];
#pragma unroll
        for (int i = 0; i < 18; ++i) w2[i] = reinterpret_cast<const v8*>(p.w2)[i * 64 + lane];
#pragma unroll
        for (int c = 0; c < 4; ++c)      // up-conv fragments: [tap pair][chunk][tap & 1][64 lanes][8]
            wu[c] = reinterpret_cast<const v8*>(p.wup)[(((ow >> 1) * 4 + c) * 2 + (ow & 1)) * 64 + lane];
        uint32_t uB = (uint32_t)(QOFF + ST * (-5 + 3) - 2 * PW) % RU;      // ring index of position 128u - 2 PW, u = k + 3
        uint32_t aC[9];      // tap fragment addresses of the second conv's tile (chunk 0), for k = 2
#pragma unroll
        for (int t9 = 0; t9 < 9; ++t9) {
            const int off = (t9 / 3) * PW + (t9 % 3) - (PW + 1);
            aC[t9] = LDS_C + rec_off((uint32_t)(32 * ow + off + l31 + QOFF) & (RC - 1), half);
        }
        Cur cu = cur_at(32 * (-5 + 3) + l31, PW / 2, HP1);          // level-1 position of this lane in up-conv tile u
        Cur co = cur_at(ST * half + 32 * ow + l31, PW, HP);          // output position of this lane in the (even, odd) step pair
        v8 fwr[4];      // final layer: [k-step][hi / lo] weight fragments
#pragma unroll
        for (int i = 0; i < 4; ++i) fwr[i] = *reinterpret_cast<const v8*>(smem + LDS_FW + (i * 64 + lane) * 16);
        const f32x16 bias_up = bias_init(0), bias_c2 = bias_init(64);
        // class biases of this half-wave's four classes and the label tables: registers (an LDS read + wait each in the epilogue
        // otherwise -- four of them serial in front of the swaps)
        const f32x4 fbias = *reinterpret_cast<const f32x4*>(cst + 96 + 4 * half);
        const u4 cut0 = reinterpret_cast<const u4*>(cst + 104)[0], cut1 = reinterpret_cast<const u4*>(cst + 104)[1];
        const u4 key0 = reinterpret_cast<const u4*>(cst + 112)[0], key1 = reinterpret_cast<const u4*>(cst + 112)[1];      // (registers: an LDS read + wait per tile otherwise)
        unsigned pkA[8], pkB[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { pkA[i] = 0u; pkB[i] = 0u; }
        int pending = 0;      // 1: the final layer of a finished pair of tiles is due
        const long nvox = p.out_nvox;
        char* const outb = reinterpret_cast<char*>(p.final_out) + (size_t)tile * p.out_tstride;
        auto final_pair = [&]() {
            // ---- final 1x1x1 on the matrix core for the tiles of this step (pk) and the previous one (pkA): the B fragment
            // of k-step s is the pair of packed quads (see k_conv_mfma); then lower lanes take the even step's position,
            // upper lanes the odd step's: softmax / label rule / store once for both
            f32x16 lg[2];
#pragma unroll
            for (int r = 0; r < 16; ++r) { lg[0][r] = 0.f; lg[1][r] = 0.f; }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int hl = 0; hl < 2; ++hl) {
                    const v8 fw = fwr[s2 * 2 + hl];
                    lg[0] = Act<T>::mfma(fw, __builtin_bit_cast(v8, u4{pkA[4 * s2], pkA[4 * s2 + 1], pkA[4 * s2 + 2], pkA[4 * s2 + 3]}), lg[0]);
                    lg[1] = Act<T>::mfma(fw, __builtin_bit_cast(v8, u4{pkB[4 * s2], pkB[4 * s2 + 1], pkB[4 * s2 + 2], pkB[4 * s2 + 3]}), lg[1]);
                }
            float l[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float bmine = fbias[e];
                unsigned a = __builtin_bit_cast(unsigned, lg[0][e] + bmine);
                unsigned b2 = __builtin_bit_cast(unsigned, lg[1][e] + bmine);
                swap32(a, b2);      // lower: a = even tile classes 0-3, b2 = its classes 4-7; upper: the odd tile's
                l[e] = __builtin_bit_cast(float, a);
                l[4 + e] = __builtin_bit_cast(float, b2);
            }
#ifdef SD_DEC0_TIMING
            if (tsk == SD_DEC0_TIMING) ts[6] = __builtin_readcyclecounter();
#endif
            float guard = 1.f;
            if constexpr (KIND != SD_OUT_LOGITS_F32) {
                // classes beyond final_cout hold -inf (weights 0, bias -inf): they are simply skipped, two at a time
                float mx = fmaxf(l[0], l[1]);
                if (gcout > 2) mx = fmaxf(mx, fmaxf(l[2], l[3]));
                if (gcout > 4) mx = fmaxf(mx, fmaxf(l[4], l[5]));
                if (gcout > 6) mx = fmaxf(mx, fmaxf(l[6], l[7]));
                float sum = 0.f;
#pragma unroll
                for (int c2 = 0; c2 < 8; c2 += 2) {
                    if (c2 == 0 || gcout > c2) {
                        l[c2] = __expf(l[c2] - mx);
                        l[c2 + 1] = __expf(l[c2 + 1] - mx);      // (an absent odd class: exp(-inf) = 0)
                        sum += l[c2];
                        sum += l[c2 + 1];
                    } else {
                        l[c2] = 0.f; l[c2 + 1] = 0.f;
                    }
                }
                const float inv = 1.0f / sum;
                guard = sum;
#pragma unroll
                for (int c2 = 0; c2 < 8; c2 += 2)
                    if (c2 == 0 || gcout > c2) { l[c2] *= inv; l[c2 + 1] *= inv; }
            } else {
                guard = logit_probe<T>(l, gcout);
            }
#ifdef SD_DEC0_TIMING
            if (tsk == SD_DEC0_TIMING) ts[7] = __builtin_readcyclecounter();
#endif
            const int x = c0 - 2 + co.xx;
            if ((unsigned)co.plane < (unsigned)nz && co.y < gH && co.xx >= 2 && co.xx < PW - 2 && x < gW) {
                const size_t v = (size_t)(__umul24(__umul24(z0 + co.plane, sH) + co.y, sW) + x);      // < 2^31 (launch_dec0)
                range_guard<T>(guard, p.ovf);
                if constexpr (KIND >= SD_OUT_LABELS_U8) {
                    uint8_t lab = 0;
                    if constexpr (KIND == SD_OUT_LABELS_U8) {
                        // distinct ids: per class (list position + 1) << 16 | cut; the passing class latest in the list wins
                        unsigned best = 0u;
#pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            if ((c & ~1) != 0 && gcout <= (c & ~1)) continue;      // (uniform; classes that exist, in pairs)
                            const unsigned cut = c < 4 ? cut0[c & 3] : cut1[c & 3], key = c < 4 ? key0[c & 3] : key1[c & 3];
                            const unsigned q = (unsigned)(l[c] * 255.f);
                            best = max(best, q >= cut ? key : 0u);
                        }
                        lab = (uint8_t)(best & 0xffu);
                    } else
                    for (int i = 0; i < p.lab.n; ++i) {
                        const int id = p.lab.ids[i];
                        float pv = 0.f;
#pragma unroll
                        for (int c = 0; c < 8; ++c) pv = (c == id) ? l[c] : pv;
                        if ((int)(uint8_t)(pv * 255.f) >= p.lab.cuts[i]) lab = (uint8_t)id;
                    }
                    reinterpret_cast<uint8_t*>(outb)[v] = lab;
                } else if constexpr (KIND == SD_OUT_PROBS_U8) {
                    uint8_t* out = reinterpret_cast<uint8_t*>(outb);
#pragma unroll
                    for (int c = 0; c < 8; ++c)
                        if (c < gcout) out[(size_t)c * nvox + v] = (uint8_t)(l[c] * 255.f);
                } else {
                    float* out = reinterpret_cast<float*>(outb);
#pragma unroll
                    for (int c = 0; c < 8; ++c)
                        if (c < gcout) out[(size_t)c * nvox + v] = l[c];
                }
            }
            cur_adv(co, (2 * ST) % PW, (2 * ST) / PW, PW, HP);
        };
        for (int k = -5; k < nsteps; ++k) {
            const int u = k + 3;
            D0_T(0);
#ifdef SD_DEC0_TIMING
            tsk = k;
#endif
#ifdef SD_PROBE_DEC0_NO_FINAL      // (timing probes, tools/build_probe_libs.sh: WRONG results)
            pending = 0;
#endif
            if (pending) { final_pair(); pending = 0; }
#ifdef SD_DEC0_TIMING
            long long t_final = __builtin_readcyclecounter();
#endif
#ifdef SD_PROBE_DEC0_NO_UP
            if (false) {
#else
            if (u >= 0) {
#endif
                // ---- one output parity (py, px) of the up-convolution of level-1 tile u
                f32x16 acc = bias_up;
                const uint32_t lt = LDS_L + (u & 3) * 4096 + rec_off(l31, half);
                v8 xl[4];
                static_for<4>([&](auto cc) { ds_read16<decltype(cc)::value * 1024>(xl[decltype(cc)::value], lt); });
                static_for<4>([&](auto cc) {
                    constexpr int c = decltype(cc)::value;
                    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(3 - c));
                    tie(xl[c]);
                    acc = Act<T>::mfma(wu[c], xl[c], acc);
                });
                const int y = 2 * cu.y + py, xx = 2 * cu.xx + px;
                const bool ok = (unsigned)cu.plane < (unsigned)nz && y < gH && (unsigned)(c0 - 2 + xx) < (unsigned)gW;
                // position relative to 128u - 2 PW (>= 0, < 396 for every lane of the tile), then the ring index
                const int qrel = (cu.plane * HP + y) * PW + xx - (ST * u - 2 * PW);
                const uint32_t idx = wrap(uB + (uint32_t)qrel, RU);
                write_tile(acc, ok, LDS_U + half * (RU * 32) + (idx << 5), ((idx >> 3) & 1u) << 4);
            }
            cur_adv(cu, 32 % (PW / 2), 32 / (PW / 2), PW / 2, HP1);
            uB += ST; if (uB >= RU) uB -= RU;
            D0_T(1);
#ifdef SD_PROBE_DEC0_NO_SECOND
            if (false) {
#else
            if (k >= 2) {
#endif
                // ---- second conv of tile ow, two steps behind the merge conv
                f32x16 acc = bias_c2;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the up-conv's ring writes are out: the counted waits below start from zero
                v8 xq[PF];
                auto issue = [&](auto ic) {
                    constexpr int i = decltype(ic)::value, c = i / 9, t9 = i % 9;
                    ds_read16<c * (RC * 32)>(xq[i % PF], aC[t9]);
                    if constexpr (c == 1) aC[t9] = LDS_C | ((aC[t9] + ST * 32) & (RC * 32 - 1));     // next step's address
                };
                static_for<PF>([&](auto ic) { issue(ic); });
                static_for<18>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    if constexpr ((i & 1) == 0) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(i + PF <= 16 ? PF - 2 : (i + 2 <= 18 ? 16 - i : 0)));
                    tie(xq[i % PF]);
                    acc = Act<T>::mfma(w2[i], xq[i % PF], acc);
                    if constexpr (i + PF < 18) issue(std::integral_constant<int, i + PF>{});
                });
                D0_T(2);
                unsigned pk[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    pk[i] = pk_max16(Act<T>::pack2(acc[2 * i], acc[2 * i + 1]), 0u);
                    sguard.see(pk[i]);
                }
                if ((k & 1) == 0) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) pkA[i] = pk[i];
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) pkB[i] = pk[i];
                    pending = 1;      // final layer of the pair: at the start of the next steps, under the merge waves' MFMAs
                }
            }
            D0_T(3);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            D0_T(4);
            asm volatile("s_barrier" ::: "memory");
#ifdef SD_DEC0_TIMING
            if (k == SD_DEC0_TIMING) ts[5] = t_final;
#endif
        }
        if (pending) final_pair();
        D0_DUMP();
        sguard.flush(p.ovf);
    }
}

int launch_dec0(Dec0Params p, int act_dtype, hipStream_t s) {
    if (p.H <= 0 || p.W <= 0 || p.D <= 0 || p.batch <= 0) return SD_ERR_INVALID;
    p.HP = 2 * (p.H / 2 + 1);
    p.HP1 = p.HP / 2;
    p.magic_hp = (unsigned)(0x100000000ull / (unsigned)p.HP) + 1u;
    p.magic_hp1 = (unsigned)(0x100000000ull / (unsigned)p.HP1) + 1u;
    // strip width: 64 columns, or 56 where that covers the width with fewer stream positions (strips x (width + 4): 331 columns =
    // 6 x 68 or 6 x 60; 128 = 2 x 68 or 3 x 60).  Same arithmetic per voxel either way: bit-identical results.
    const char* const pw_env = getenv("SD_DEC0_PW");      // (A/B switch, read per launch: 68 or 60)
    const int force_pw = pw_env ? atoi(pw_env) : 0;
    const int nsw = (p.W + PW_WIDE - 5) / (PW_WIDE - 4), nsn = (p.W + PW_NARROW - 5) / (PW_NARROW - 4);
    const bool narrow = force_pw ? force_pw == PW_NARROW : nsn * PW_NARROW < nsw * PW_WIDE;
    const int pw = narrow ? PW_NARROW : PW_WIDE;
    p.nstrip = narrow ? nsn : nsw;
    p.lab_fast = 0;
    for (int c = 0; c < 8; ++c) p.lab_cls[c] = 0u;
    if (p.final_kind == SD_OUT_LABELS_U8) {
        p.lab_fast = 1;
        for (int i = 0; i < p.lab.n; ++i) {
            const int id = p.lab.ids[i];
            if (id < 0 || id >= 8 || p.lab_cls[id]) { p.lab_fast = 0; break; }      // repeated id: generic rule in the kernel
            p.lab_cls[id] = ((unsigned)(i + 1) << 16) | (unsigned)p.lab.cuts[i];
        }
    }
    // one workgroup per CU; each takes a contiguous range of z-planes of one (tile, strip)
    static int ncu = 0;
    static std::once_flag once;
    std::call_once(once, [] {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
        if (ncu <= 0) ncu = 256;
    });
    // Workgroups: one per (tile, strip, range of zsplit planes); only one fits a CU (LDS).  Pick the plane count per workgroup
    // that minimises the makespan rounds * (zsplit + prologue): many small workgroups balance a grid that does not divide the
    // CUs, few large ones save the per-workgroup prologue (LDS clear, weight fragments, pipeline fill ~ 1/8 plane).
    const int combos = p.batch * p.nstrip;
    double best = 1e30;
    for (int zs = 1; zs <= p.D; ++zs) {
        const long nwg = (long)combos * ((p.D + zs - 1) / zs);
        const double span = (double)((nwg + ncu - 1) / ncu) * (zs + 0.125);
        if (span < best * 0.999 || (span < best * 1.001 && zs > p.zsplit)) { best = std::min(best, span); p.zsplit = zs; }
    }
    p.nzg = (p.D + p.zsplit - 1) / p.zsplit;
    p.prio = getenv("SD_DEC0_PRIO") ? atoi(getenv("SD_DEC0_PRIO")) : 0;
    if ((long)p.zsplit * p.HP * pw > (1l << 30) || p.H < 8 || (long)p.D * p.H >= (1l << 24) || p.W >= (1 << 24))
        return SD_ERR_INVALID;      // 32-bit positions, 24-bit row arithmetic, cursor wraps once per advance (== dec0_shape_ok, sd_api.hip)
    if ((long)p.D * p.H * p.W >= (1l << 31)) return SD_ERR_INVALID;
    if (p.final_kind < 0 || p.final_kind > SD_OUT_LABELS_U8) return SD_ERR_INVALID;
    const int kind = p.final_kind == SD_OUT_LABELS_U8 && !p.lab_fast ? 4 : p.final_kind;
    // whole tile unless the host filled in the extents of the tensors a sub-box lies in
    if (p.sH == 0) { p.sH = p.H; p.sW = p.W; p.sH1 = p.H1; p.sW1 = p.W1; }
    if (p.Pl1 == 0) p.Pl1 = (size_t)p.D * p.H1 * p.W1;
    if (p.out_nvox == 0) p.out_nvox = (long)p.D * p.H * p.W;
    const bool vw = p.sH != p.H || p.sW != p.W || p.sH1 != p.H1 || p.sW1 != p.W1;
    void (*kern)(const Dec0Params) = nullptr;
    auto pick = [&](auto tc, auto vc, auto pc) -> void (*)(const Dec0Params) {
        using TT = typename decltype(tc)::type;
        constexpr bool V = decltype(vc)::value;
        constexpr int P = decltype(pc)::value;
        return kind == 0 ? k_dec0<TT, 0, V, P> : kind == 1 ? k_dec0<TT, 1, V, P> : kind == 2 ? k_dec0<TT, 2, V, P> : kind == 3 ? k_dec0<TT, 3, V, P>
                                                                                                                                  : k_dec0<TT, 4, V, P>;
    };
    auto pick_pw = [&](auto tc, auto vc) {
        return narrow ? pick(tc, vc, std::integral_constant<int, PW_NARROW>{}) : pick(tc, vc, std::integral_constant<int, PW_WIDE>{});
    };
    auto pick_view = [&](auto tc) { return vw ? pick_pw(tc, std::true_type{}) : pick_pw(tc, std::false_type{}); };
    kern = act_dtype == SD_BF16 ? pick_view(TypeTag<bf16_t>{}) : pick_view(TypeTag<f16_t>{});
    static std::mutex mu;
    static bool attr_done[40][64] = {};      // the attribute is per kernel and device
    {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return SD_ERR_HIP;
        std::lock_guard<std::mutex> g(mu);
        bool& done = attr_done[(act_dtype == SD_BF16 ? 0 : 5) + kind + (vw ? 10 : 0) + (narrow ? 20 : 0)][dev];
        if (!done) {
            const hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
            if (ea != hipSuccess) {
#ifdef SD_DEC0_TIMING
                fprintf(stderr, "k_dec0 attribute: %s\n", hipGetErrorString(ea));
#endif
                return SD_ERR_HIP;
            }
            done = true;
        }
    }
    {
        static const char* const names[2][2][5] = {
            {{"k_dec0<logits>", "k_dec0<probs f32>", "k_dec0<probs u8>", "k_dec0<labels>", "k_dec0<labels, generic id list>"},
             {"k_dec0<logits, sub-box>", "k_dec0<probs f32, sub-box>", "k_dec0<probs u8, sub-box>", "k_dec0<labels, sub-box>",
              "k_dec0<labels, generic id list, sub-box>"}},
            {{"k_dec0<logits, 56-column strips>", "k_dec0<probs f32, 56-column strips>", "k_dec0<probs u8, 56-column strips>",
              "k_dec0<labels, 56-column strips>", "k_dec0<labels, generic id list, 56-column strips>"},
             {"k_dec0<logits, sub-box, 56-column strips>", "k_dec0<probs f32, sub-box, 56-column strips>", "k_dec0<probs u8, sub-box, 56-column strips>",
              "k_dec0<labels, sub-box, 56-column strips>", "k_dec0<labels, generic id list, sub-box, 56-column strips>"}}};
        SD_NOTE_KERNEL(names[narrow ? 1 : 0][vw ? 1 : 0][kind]);
    }
    hipLaunchKernelGGL(kern, dim3(combos * p.nzg), dim3(512), LDS_TOTAL, s, p);
    const hipError_t e = hipGetLastError();
#ifdef SD_DEC0_TIMING
    if (e != hipSuccess) fprintf(stderr, "k_dec0 launch: %s\n", hipGetErrorString(e));
#endif
    return e == hipSuccess ? SD_OK : SD_ERR_HIP;
}
