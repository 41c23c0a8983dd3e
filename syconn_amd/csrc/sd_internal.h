// Internal interface between the host-side plan executor (sd_api.hip) and the gfx950 kernels (sd_kernels.hip).
// Nothing here is part of the C ABI (include/syconn_dense.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/syconn_dense.h"

// Activation tensors are CHANNEL-BLOCKED: a "chunk" is 16 consecutive channels = one MFMA k-step, and element
// (chunk k, z, y, x, c) sits at ((k*P + (z*H + y)*W + x)*16 + c) with P = D*H*W the voxels of the buffer's own
// extents ("plane").  Every 16-channel slab of an x-row is contiguous: a halo row of 18 voxels is one 576-byte run
// for the LDS-DMA gathers (voxel-major records cost a 128-byte line per 32 useful bytes).  C / Cd / Cs below are
// the padded channel COUNT (multiple of 16).
// Batched launches: every kernel can process `batch` independent tiles in ONE launch.  Tile t uses the workspace at
// byte offset t*tstride (all tiles share one workspace layout, so the same offset applies to every activation
// buffer and scratch pointer of a launch); the network input / final output of tile t sit t*in_tstride /
// t*out_tstride bytes after tile 0's.
constexpr int SD_CHUNK = 16;
constexpr int SD_CONV_PARAM_BYTES = 512;     // k_conv_mfma LDS constants: folded bias (<= 96 floats) + 8 class biases

// geometry of one workgroup of the first-layer convolution: 256 output voxels = 8 MFMA column tiles of
// (2 y-rows x 16 x); 3x3x3: 2x8x16 voxels, 1x3x3: 1x16x16 voxels.  (The generic conv picks its own, ConvGeo.)
constexpr int SD_BX = 16;
__host__ __device__ constexpr int sd_bz(int KZ) { return KZ == 3 ? 2 : 1; }
__host__ __device__ constexpr int sd_by(int KZ) { return KZ == 3 ? 8 : 16; }

// Every launcher notes the kernel symbol it picked (a string with static storage); sd_forward* keeps it per plan operation for
// sd_debug_op_kernel, so that a benchmark names the kernels that RAN instead of restating the selection rules.
extern thread_local const char* sd_tls_kernel;
#define SD_NOTE_KERNEL(str) (sd_tls_kernel = (str))

struct LabelArgs { int n; int ids[16]; int cuts[16]; };   // label rule: later entries override; p >= cut <=> p > t

struct ConvParams {
    const void* src0;  // first input (for a merged conv: the up-convolved tensor, cropped by reading fewer voxels)
    const void* src1;  // second input or nullptr
    int C0, H0, W0;    // padded channels and y/x extents (strides) of src0
    size_t P0, P1, Pd, Pp;  // voxels per chunk plane of src0 / src1 / dst / pool_dst
    int C1, H1, W1;
    int nchunk0, nchunk1;
    void* dst;
    int Cd;            // channel stride of dst; channels >= Cd are not stored
    int D, H, W;       // output extent = the box the launch computes (zero padding beyond it)
    int Hd, Wd;        // y / x extents of the dst tensor (= H, W unless the launch computes a sub-box of it: sd_model_set_roi)
    long final_nvox;   // voxels per class plane of final_out (= D*H*W unless a sub-box)
    const void* wpack; // [nb][chunk][kz][9][NT][64 lanes][8] of T
    const float* bias; // padded to NB*NT*32
    int relu;
    int nbx, nby, nbz; // workgroup grid over the output volume
    int block_order;   // 3x3x3: 0 = z-fastest block list, 1 = 4x4x2-brick order (L2 locality of the halo planes)
    const void* zero;  // >= 16 zero bytes (DMA source of out-of-volume halo voxels)
    int store_main;    // write dst (0 when only the fused final output is needed)
    void* pool_dst;    // fused MaxPool(ceil): pooled tensor (same channel stride), or nullptr
    const unsigned* pool_dir;  // with pool_dst behind a fused-statistics GroupNorm conv (raw outputs): per (32-column tile, lane half,
                               // register) a mask of 0xffff halves where the channel's gamma < 0 (pool the MINIMUM there), else nullptr
    int pH, pW;        // y/x extents of the pooled tensor
    const void* final_wfrag; // fused conv_final: hi/lo weight fragments [NT][2 k-steps][2][64 lanes][8] of T, or nullptr
    const float* final_b;
    int final_cout, final_kind;
    void* final_out;   // planar (cout, D*H*W)
    long long* dbg;         // SD_TIMING builds: per-wave cycle stamps
    unsigned long long* clk; // sd_profile_enable: workgroup 0 stamps {s_memtime, s_memrealtime} at entry and exit here (4 values), or nullptr
    int batch; size_t tstride, out_tstride;
    int batch_total;   // > 0: tiles of the whole launch SET when it is issued in tile groups (deferred GroupNorm): the kernel
                       // form (512- / 256-voxel workgroups, 4-tile form) must not depend on the size of the last group
    // fused GroupNorm statistics: per-channel sum / sum of squares of the stored outputs are added to
    // gn_sums[0..gn_C) / gn_sums[gn_C..2*gn_C) of the block's tile (doubles, zeroed by the host), or nullptr
    double* gn_sums; int gn_C;
    LabelArgs lab;     // final_kind == SD_OUT_LABELS_U8
    // fused FIRST convolution (1 -> 32 channels, 1x3x3): src0 is then NOT read, its two chunks are computed from the
    // network input inside the kernel (k_conv_mfma<..., FF = true>)
    const void* first_in; size_t first_in_tstride; int first_in_f32;
    const float* first_w;      // [5 k-steps][64 lanes] exact-f32 MFMA A fragments of the 32 output channels
    const float* first_bias;   // folded bias, 32 floats
    const void* first_w3;      // uint8 input on the bf16 pipe (MODE 5): [tile][2 MFMAs][64 lanes][8] bf16 parts of weights / 255 and bias, or nullptr
    int first_relu;
    // deferred GroupNorm apply (k_conv_mfma<..., MODE 2>): src0 / src1 hold RAW convolution outputs; gn0 / gn1 = their
    // per-tile [2*C] float tables (scale then shift; tile t at + t*tstride bytes) or nullptr when that input is final
    const float* gn0; const float* gn1; int gn_relu0, gn_relu1;
    int* ovf;          // fp16 range guard flag (sd_device.h: StoreGuard on every store, range_guard in the fused final layer)
    float oscale;      // split-fp16 plan (k_conv_mfma<..., MODE 3>): 2^-k, undoes the power-of-two scale of the packed weights / bias
    float final_oscale;   // ... and the same for the fused final layer's hi / lo weight fragments
};

struct FirstParams {
    const void* in;    // (D,H,W) planar uint8 or float
    int D, H, W;
    void* dst;
    int Cd;
    const float* wpack;  // [ntile][nstep][64 lanes] float
    const void* wpack3;  // planar first conv on uint8 input, bf16 / fp16 plans: [ntile][2 MFMAs][64 lanes][8] bf16 (sd_api.hip), or nullptr
    const float* bias;   // padded to ntile*32
    int relu;
    int nbx, nby, nbz;
    int batch; size_t tstride, in_tstride;
    int* ovf;          // fp16 range guard flag (sd_device.h)
};

struct UpconvParams {
    const void* src;   // (D,H,W,Cs)
    int Cs, nchunk;
    int D, H, W;       // the box of source voxels the launch works on
    size_t Ps; int Hs, Ws;   // voxels per chunk plane and y / x extents of the source tensor (D*H*W, H, W unless a sub-box)
    int Hd, Wd;        // y / x extents of the dst tensor (2H, 2W unless a sub-box)
    void* dst;         // (D*kz, 2H, 2W, Cd)
    size_t Pd;         // voxels per chunk plane of dst
    int Cd;
    int kz;            // 1 or 2
    const void* wpack; // [nb][chunk][NT=2][64][8]
    const float* bias; // per n = tap*Cd + co, padded to NB*64
    int relu;
    int ntot;          // ntaps*Cd
    int batch; size_t tstride;
    const float* gn; int gn_relu;   // deferred GroupNorm apply of `src` ([2*Cs] floats per tile) or nullptr
    int* ovf;          // fp16 range guard flag (sd_device.h)
    float oscale;      // split-fp16 plan: 2^-k of the packed weights / bias
};

struct PoolParams {
    const void* src; void* dst;
    int C;             // channel stride (same for src and dst)
    int D, H, W;       // src dims
    int Do, Ho, Wo;    // dst dims
    int kz;            // 1 or 2
    size_t Ps, Pd;     // voxels per chunk plane
    int batch; size_t tstride;
};

struct FinalParams {
    const void* src; int Cs; int cin;
    const float* w;    // [cout][Cs] float (zero padded)
    const float* bias; // [cout]
    int cout;
    void* out;         // planar (cout, nvox)
    int out_kind;
    long nvox;
    int batch; size_t tstride, out_tstride;
    LabelArgs lab;     // out_kind == SD_OUT_LABELS_U8
    // GroupNorm apply (+ReLU) of `src` fused into this pass: x -> round_T(relu(x*scale + shift)) per channel before the
    // 1x1x1 convolution (the normalised tensor is never written); scale_shift = [2*Cs] floats per tile or nullptr
    const float* gn_scale_shift; int gn_relu;
    const void* wfrag; // hi/lo MFMA A fragments of the weights in natural channel order: [Cs/16][2][64 lanes][8] of T
    int* ovf;          // fp16 range guard (sd_device.h: range_guard)
};

struct GnParams {
    void* buf; int C;          // channel stride
    int D, H, W;               // logical region
    int Hs, Ws;                // strides (buffer y/x extents)
    size_t P;                  // voxels per chunk plane of the buffer
    int groups, cout;          // real channel count
    float eps;
    const float* gamma; const float* beta;   // padded to C
    double* sums;              // [2*C] workspace (sum, sumsq)
    float* scale_shift;        // [2*C]
    int relu;
    int batch; size_t tstride;
    int skip_stats;            // the producing convolution already accumulated `sums` (ConvParams::gn_sums)
    void* pool_dst;            // fused MaxPool3d(ceil_mode) of the normalised tensor ((pkz,2,2) windows), or nullptr
    int pkz, pD, pH, pW;       // pooling kz (1 or 2) and pooled extents
    int skip_apply;            // statistics -> scale/shift only: the consumers apply them on the fly (deferred apply)
    int no_inplace;            // with pool_dst: write only the pooled normalised tensor, leave `buf` raw (deferred apply)
};

// Fused level-0 decoder (sd_dec0.hip): planar up-convolution 64 -> 32 + merge conv (32 + 32 -> 32) + conv 32 -> 32 + final
// 1x1x1 in one streaming kernel.  Weight fragment blobs are the ones the separate layers use.
struct Dec0Params {
    const void* l1;    // level-1 tensor, 4 chunks, extents (D, H1, W1)
    const void* skip;  // encoder skip tensor, 2 chunks, extents (D, H, W)
    size_t Ps;         // voxels per chunk plane of the skip tensor
    int D, H, W, H1, W1;   // the box the launch computes and its level-1 counterpart
    int sH, sW, sH1, sW1;  // y / x extents of the tensors the box lies in (0: the box is the whole tile; launch_dec0 fills them in)
    size_t Pl1;        // voxels per chunk plane of the level-1 tensor (0: D*H1*W1)
    long out_nvox;     // voxels per class plane of final_out (0: D*H*W)
    const void* wup; const float* bup;   // up-conv fragments [tap pair][chunk][tap & 1][64][8], folded bias (32 floats)
    const void* w1; const float* b1;     // merge conv fragments [chunk 0..3][tap][64][8], folded bias
    const void* w2; const float* b2;     // second conv fragments [chunk 0..1][tap][64][8], folded bias
    const void* fw; const float* fb;     // final layer: hi/lo fragments [2 k-steps][2][64][8], class bias
    int final_cout, final_kind;
    void* final_out;
    LabelArgs lab;
    int lab_fast; unsigned lab_cls[8];   // distinct label ids: per class (list position + 1) << 16 | cut, 0 = not listed (launch_dec0)
    const void* zero;  // >= 16 zero bytes
    int batch; size_t tstride, out_tstride;
    // filled in by launch_dec0
    int HP, HP1, nstrip, zsplit, nzg, prio;
    unsigned magic_hp, magic_hp1;
    long long* dbg;    // SD_DEC0_TIMING builds: per-wave cycle stamps of one step (else unused)
    int* ovf;          // fp16 range guard (sd_device.h: range_guard)
};
int launch_dec0(Dec0Params p, int act_dtype, hipStream_t s);

int launch_conv(const ConvParams& p, int act_dtype, int KZ, int NT, int NB, hipStream_t s);
// LDS bytes a deferred-GroupNorm convolution needs per tile of the launch for its scale / shift table
inline size_t conv_gn_lds_per_tile(int nchunks) { return (size_t)nchunks * 128; }
// true when launch_conv would run this planar layer with LDS-resident weights and 512-voxel workgroups -- the form
// that can compute a fused first convolution (ConvParams::first_in)
bool conv_can_fuse_first(int KZ, int NT, int NB, long vox_all_tiles, int nstages, bool fused_final);
int launch_first(const FirstParams& p, int act_dtype, int in_dtype, int KZ, hipStream_t s);
int launch_upconv(const UpconvParams& p, int act_dtype, int NB, hipStream_t s);
bool upconv_rows_kernel(int nchunk, int Cd);   // shapes served by k_upconv_rows (see sd_kernels.hip)
int launch_pool(const PoolParams& p, int act_dtype, hipStream_t s);
int launch_final(const FinalParams& p, int act_dtype, hipStream_t s);
int launch_groupnorm(const GnParams& p, int act_dtype, hipStream_t s);
// workspace scratch (first 64 KiB of a tile's workspace): [0, SD_GN_SCALE_OFF) GroupNorm sums (2 * C doubles, zero between ops),
// [SD_GN_SCALE_OFF, 64 KiB) scale / shift of a GroupNorm whose apply is not deferred (2 * C floats)
#define SD_GN_SCALE_OFF ((size_t)40960)
int launch_zero_scratch(void* ws, size_t tstride, int nbytes, int batch, hipStream_t s);
int launch_tile_gather(const void* vol, int esize, int VD, int VH, int VW, int oz, int oy, int ox, void* tile, int TD,
                       int TH, int TW, hipStream_t s);
int launch_tile_scatter(const void* tile, int esize, int C, int TD, int TH, int TW, int cz, int cy, int cx, int KD,
                        int KH, int KW, void* vol, int VD, int VH, int VW, int oz, int oy, int ox, hipStream_t s);
int launch_downsample2(const void* src, int esize, int D, int H, int W, void* dst, hipStream_t s);
int launch_box_majority(const uint8_t* vol, int D, int H, int W, const int32_t* origins, long n, int ez, int ey, int ex,
                        int cut, double thresh_majority, uint8_t* out, hipStream_t s);
int launch_labels(const uint8_t* probs, size_t nvox, const LabelArgs& a, void* out, int out_u64, hipStream_t s);
int launch_read_buffer(const void* buf, int act_dtype, int C, int Cs, int D, int H, int W, float* out, hipStream_t s);
// split-fp16 plan (sd_split.hip; act_dtype SD_F16X2): a tensor of C (padded) channels = 2 * C / 16 fp16 chunk planes [hi | lo]
int launch_conv_split(const ConvParams& p, int KZ, int NT, int NB, hipStream_t s);
bool conv_can_fuse_first_split(int KZ, int NT, int NB, long vox_all_tiles, int nstages, bool fused_final);
int launch_pool_split(const PoolParams& p, hipStream_t s);
int launch_final_split(const FinalParams& p, hipStream_t s);
int launch_groupnorm_split(const GnParams& p, hipStream_t s);
int launch_gn_finalize(const GnParams& p, hipStream_t s);      // statistics scratch -> per-channel scale / shift (sd_kernels.hip)
int launch_read_buffer_split(const void* buf, int C, int Cs, long nvox, float* out, hipStream_t s);

// ---- reference-precision mode (sd_f32.hip): act_dtype = SD_F32, planar fp32 activations, plain FMA kernels ----------------
struct sd_f32_model;
int f32_model_create(const sd_op_desc* ops, int n_ops, const float* W, size_t n_floats, sd_f32_model** out);
void f32_model_destroy(sd_f32_model* m);
int f32_final_cout(const sd_f32_model* m);
int f32_buf_channels(const sd_f32_model* m, int b);
size_t f32_workspace_bytes(sd_f32_model* m, int D, int H, int W);
int f32_forward(sd_f32_model* m, const void* in_dev, int in_dtype, int N, int D, int H, int W, void* out_dev, int out_kind,
                const LabelArgs* lab, void* ws, size_t ws_bytes, hipStream_t s, hipEvent_t* ev);
int f32_read_buffer(sd_f32_model* m, int buf, const void* ws, float* out, int32_t* dims4, hipStream_t s);
