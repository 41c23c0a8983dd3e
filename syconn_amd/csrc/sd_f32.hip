// Reference-precision mode of the library (act_dtype = SD_F32): the whole network in fp32 storage and fp32 FMA
// arithmetic -- what the reference computes (/root/reference/syconn/handler/prediction.py:777-779: Predictor is built
// without float16, so elektronn3 / torch run the U-Net in fp32).  It exists so that labels can be compared with the
// fp32 CPU oracle where only the summation order differs (relative logit error ~1e-6 instead of 1e-3 (fp16) / 7e-3
// (bf16)); it is NOT the fast path: plain vector FMAs (fp32 has no matrix-core advantage on gfx950: 157 TFLOP/s on both
// pipes), one launch per layer, no fusion.  Roughly 25x the time of the bf16 plan, still ~10^3 x the CPU oracle.
//
// Layout: activations planar (C, D, H, W) fp32 with the REAL channel count -- torch's own layout, so layer-wise buffers
// can be compared with the oracle's tensors directly.  Arithmetic follows torch's eval graph: conv (+ bias), then the
// BatchNorm affine x * alpha + beta' (alpha = gamma / sqrt(var + eps), beta' = beta - mean * alpha), then ReLU;
// GroupNorm as x * (rstd * gamma) + (beta - mean * rstd * gamma) with statistics accumulated in double.
#include "sd_internal.h"
#include "../../include/syconn_dense.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

int sd_fail_msg(int code, const char* msg);

namespace {

struct Dims3 { int d = 0, h = 0, w = 0; };
inline size_t rup_sz(size_t v, size_t m) { return (v + m - 1) / m * m; }

struct FOp {
    sd_op_desc d{};
    size_t w_off = 0;                       // float offsets into the device blob
    size_t bias_off = 0, alpha_off = 0, beta_off = 0;
    int CO = 16, ncob = 0;                  // output channels per workgroup / number of channel blocks
};

constexpr size_t F32_SCRATCH = 4096;        // per tile: GroupNorm sums, double[2 * groups]

// ------------------------------------------------------------------------------------------------------------------
struct Conv32 {
    const void* in0; const float* in1;
    int C0, C1;                 // channels of the two inputs (C1 = 0: single input)
    int H0, W0, H1, W1;         // y / x extents (strides) of the inputs
    size_t P0, P1;              // voxels per channel plane of the inputs
    float* dst; int Cout;
    int D, H, W;                // output extent (= the region of the inputs that is read: autocrop at the high end)
    const float* w;             // [ncob][Cin][KZ*9][CO]
    const float* bias; const float* alpha; const float* beta;   // padded to ncob*CO
    int relu;
    int nbx, nby, nbz;
    size_t tstride, in_tstride; // bytes between tiles: workspace / network input
    int in0_is_input;           // in0 is the network input (planar, one channel; uint8 is normalised as float(v)/255)
};

template <typename IN> __device__ __forceinline__ float ld_in(const IN* p);
template <> __device__ __forceinline__ float ld_in<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld_in<uint8_t>(const uint8_t* p) { return (float)(*p) / 255.f; }   // prediction.py:808

// Direct convolution, k = (KZ,3,3), 'same' zero padding.  Workgroup = TZ x TY x (4*TX) output voxels x CO output channels;
// a thread owns 4 consecutive x voxels x CO channels (4*CO fp32 accumulators).  Per input channel the halo patch of the
// block sits in LDS (double-buffered: the next channel's patch is fetched into registers while this one is consumed) and
// the KZ*9*CO weights of the (channel block, input channel) pair are wave-uniform -> scalar loads, FMAs with an SGPR operand.
template <int KZ, int TZ, int TY, int TX, int CO, typename IN>
__global__ __launch_bounds__(256) void k32_conv(const Conv32 p) {
    static_assert(TZ * TY * TX == 256, "one thread per (z, y, x-quad)");
    constexpr int PZ = TZ + KZ - 1, PY = TY + 2, PX = TX * 4 + 2, PXS = TX * 4 + 4;
    constexpr int PN = PZ * PY * PX, NL = (PN + 255) / 256;
    __shared__ __attribute__((aligned(16))) float patch[2][PZ * PY * PXS];
    const int tid = threadIdx.x;
    const int tx = tid % TX, ty = (tid / TX) % TY, tz = tid / (TX * TY);
    const int bx = blockIdx.x % p.nbx, by = (blockIdx.x / p.nbx) % p.nby, bz = blockIdx.x / (p.nbx * p.nby);
    const int x0 = bx * TX * 4, y0 = by * TY, z0 = bz * TZ;
    const int cob = blockIdx.y, Cin = p.C0 + p.C1;
    const char* const in0 = reinterpret_cast<const char*>(p.in0) + blockIdx.z * (p.in0_is_input ? p.in_tstride : p.tstride);
    const float* const in1 = p.in1 ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.in1) + blockIdx.z * p.tstride) : nullptr;

    float pre[NL];
    auto fetch = [&](int ci) {
        const bool s1 = ci >= p.C0;
        const int Hs = s1 ? p.H1 : p.H0, Ws = s1 ? p.W1 : p.W0;
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int i = tid + k * 256;
            const int px = i % PX, py = (i / PX) % PY, pz = i / (PX * PY);
            const int z = z0 + pz - (KZ == 3 ? 1 : 0), y = y0 + py - 1, x = x0 + px - 1;
            float v = 0.f;
            if (i < PN && z >= 0 && z < p.D && y >= 0 && y < p.H && x >= 0 && x < p.W) {
                const size_t o = ((size_t)z * Hs + y) * Ws + x;
                if (s1) v = in1[(size_t)(ci - p.C0) * p.P1 + o];
                else v = ld_in<IN>(reinterpret_cast<const IN*>(in0) + (size_t)ci * p.P0 + o);
            }
            pre[k] = v;
        }
    };
    auto park = [&](int buf) {
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int i = tid + k * 256;
            const int px = i % PX, py = (i / PX) % PY, pz = i / (PX * PY);
            if (i < PN) patch[buf][(pz * PY + py) * PXS + px] = pre[k];
        }
    };

    float acc[4][CO];
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int c = 0; c < CO; ++c) acc[v][c] = 0.f;

    fetch(0);
    park(0);
    __syncthreads();
    for (int ci = 0; ci < Cin; ++ci) {
        if (ci + 1 < Cin) fetch(ci + 1);
        const float* __restrict__ wci = p.w + ((size_t)cob * Cin + ci) * (KZ * 9 * CO);
        const float* const pc = patch[ci & 1];
#pragma unroll
        for (int kz = 0; kz < KZ; ++kz)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const float* const row = pc + ((tz + kz) * PY + (ty + ky)) * PXS + tx * 4;
                float in[6];
#pragma unroll
                for (int e = 0; e < 6; ++e) in[e] = row[e];
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int c = 0; c < CO; ++c) {
                        const float wv = wci[((kz * 3 + ky) * 3 + kx) * CO + c];
#pragma unroll
                        for (int v = 0; v < 4; ++v) acc[v][c] = fmaf(in[v + kx], wv, acc[v][c]);
                    }
            }
        if (ci + 1 < Cin) park((ci + 1) & 1);
        __syncthreads();
    }

    const int z = z0 + tz, y = y0 + ty, xb = x0 + tx * 4;
    if (z >= p.D || y >= p.H) return;
    float* const dst = reinterpret_cast<float*>(reinterpret_cast<char*>(p.dst) + blockIdx.z * p.tstride);
    const size_t P = (size_t)p.D * p.H * p.W, vo = ((size_t)z * p.H + y) * p.W + xb;
#pragma unroll
    for (int c = 0; c < CO; ++c) {
        const int co = cob * CO + c;
        if (co >= p.Cout) break;
        const float b = p.bias[co], al = p.alpha[co], be = p.beta[co];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            float r = (acc[v][c] + b) * al + be;
            if (p.relu) r = fmaxf(r, 0.f);
            if (xb + v < p.W) dst[(size_t)co * P + vo + v] = r;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
struct Up32 {
    const float* src; int Cin; int D, H, W;
    float* dst; int Cout; int kz;
    const float* w;          // [Cin][ncob][4 co][taps]
    const float* bias; const float* alpha; const float* beta;
    int relu; int ncob;
    size_t tstride;
};
// ConvTranspose3d k = s = (KZ2,2,2): one thread per INPUT voxel and block of 4 output channels; every output voxel takes
// exactly one tap.
template <int KZ2>
__global__ __launch_bounds__(256) void k32_upconv(const Up32 p) {
    constexpr int TAPS = KZ2 * 4;
    const long M = (long)p.D * p.H * p.W;
    const long m = (long)blockIdx.x * 256 + threadIdx.x;
    const int cob = blockIdx.y;
    const float* const src = reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.src) + blockIdx.z * p.tstride);
    float* const dst = reinterpret_cast<float*>(reinterpret_cast<char*>(p.dst) + blockIdx.z * p.tstride);
    float acc[4][TAPS];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int t = 0; t < TAPS; ++t) acc[c][t] = 0.f;
    const long mm = m < M ? m : M - 1;
    for (int ci = 0; ci < p.Cin; ++ci) {
        const float x = src[(size_t)ci * M + mm];
        const float* __restrict__ wci = p.w + ((size_t)ci * p.ncob + cob) * (4 * TAPS);
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int t = 0; t < TAPS; ++t) acc[c][t] = fmaf(x, wci[c * TAPS + t], acc[c][t]);
    }
    if (m >= M) return;
    const int xw = (int)(m % p.W), y = (int)((m / p.W) % p.H), z = (int)(m / ((long)p.W * p.H));
    const int H2 = 2 * p.H, W2 = 2 * p.W;
    const size_t Pd = (size_t)p.D * KZ2 * H2 * W2;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int co = cob * 4 + c;
        if (co >= p.Cout) break;
        const float b = p.bias[co], al = p.alpha[co], be = p.beta[co];
#pragma unroll
        for (int a = 0; a < KZ2; ++a)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                float r0 = (acc[c][(a * 2 + bb) * 2 + 0] + b) * al + be, r1 = (acc[c][(a * 2 + bb) * 2 + 1] + b) * al + be;
                if (p.relu) { r0 = fmaxf(r0, 0.f); r1 = fmaxf(r1, 0.f); }
                float2* const o = reinterpret_cast<float2*>(dst + (size_t)co * Pd + ((size_t)(z * KZ2 + a) * H2 + (2 * y + bb)) * W2 + 2 * xw);
                *o = float2{r0, r1};
            }
    }
}

// ------------------------------------------------------------------------------------------------------------------
struct Pool32 { const float* src; float* dst; int C, D, H, W, Do, Ho, Wo, kz; size_t tstride; };
__global__ __launch_bounds__(256) void k32_pool(const Pool32 p) {      // MaxPool3d k = (kz,2,2), ceil_mode=True
    const float* const src = reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.src) + blockIdx.z * p.tstride);
    float* const dst = reinterpret_cast<float*>(reinterpret_cast<char*>(p.dst) + blockIdx.z * p.tstride);
    const size_t total = (size_t)p.C * p.Do * p.Ho * p.Wo;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int xo = (int)(i % p.Wo), yo = (int)((i / p.Wo) % p.Ho);
        const size_t r = i / ((size_t)p.Wo * p.Ho);
        const int zo = (int)(r % p.Do), c = (int)(r / p.Do);
        float mx = -INFINITY;
        for (int dz = 0; dz < p.kz; ++dz)
            for (int dy = 0; dy < 2; ++dy)
                for (int dx = 0; dx < 2; ++dx) {
                    const int z = zo * p.kz + dz, y = yo * 2 + dy, x = xo * 2 + dx;
                    if (z < p.D && y < p.H && x < p.W) mx = fmaxf(mx, src[(((size_t)c * p.D + z) * p.H + y) * p.W + x]);
                }
        dst[i] = mx;
    }
}

// ------------------------------------------------------------------------------------------------------------------
struct Gn32 {
    float* buf; int C, groups;
    int D, H, W;             // region the statistics and the apply cover (autocrop of an up-convolution output)
    int Hs, Ws; size_t P;    // strides of the buffer
    const float* gamma; const float* beta; float eps; int relu;
    double* sums;            // per tile [groups][2], zeroed by the host
    size_t tstride;
};
__global__ __launch_bounds__(256) void k32_gn_stats(const Gn32 p) {
    const int g = blockIdx.y, cpg = p.C / p.groups;
    const float* const buf = reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.buf) + blockIdx.z * p.tstride);
    const size_t R = (size_t)p.D * p.H * p.W, total = R * cpg;
    double s = 0.0, ss = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int x = (int)(i % p.W), y = (int)((i / p.W) % p.H);
        const size_t r = i / ((size_t)p.W * p.H);
        const int z = (int)(r % p.D), c = g * cpg + (int)(r / p.D);
        const double v = buf[(size_t)c * p.P + ((size_t)z * p.Hs + y) * p.Ws + x];
        s += v; ss += v * v;
    }
    __shared__ double red[2][256];
    red[0][threadIdx.x] = s; red[1][threadIdx.x] = ss;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { red[0][threadIdx.x] += red[0][threadIdx.x + o]; red[1][threadIdx.x] += red[1][threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double* const sums = reinterpret_cast<double*>(reinterpret_cast<char*>(p.sums) + blockIdx.z * p.tstride);
        atomicAdd(&sums[2 * g], red[0][0]);
        atomicAdd(&sums[2 * g + 1], red[1][0]);
    }
}
__global__ __launch_bounds__(256) void k32_gn_apply(const Gn32 p) {
    float* const buf = reinterpret_cast<float*>(reinterpret_cast<char*>(p.buf) + blockIdx.z * p.tstride);
    const double* const sums = reinterpret_cast<const double*>(reinterpret_cast<const char*>(p.sums) + blockIdx.z * p.tstride);
    const int cpg = p.C / p.groups;
    const size_t R = (size_t)p.D * p.H * p.W, total = R * p.C;
    const double n = (double)R * cpg;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int x = (int)(i % p.W), y = (int)((i / p.W) % p.H);
        const size_t r = i / ((size_t)p.W * p.H);
        const int z = (int)(r % p.D), c = (int)(r / p.D), g = c / cpg;
        const double mean = sums[2 * g] / n;
        double var = sums[2 * g + 1] / n - mean * mean;
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)p.eps));
        const float scale = rstd * p.gamma[c], shift = p.beta[c] - scale * (float)mean;
        float* const e = buf + (size_t)c * p.P + ((size_t)z * p.Hs + y) * p.Ws + x;
        float v = *e * scale + shift;
        if (p.relu) v = fmaxf(v, 0.f);
        *e = v;
    }
}

// ------------------------------------------------------------------------------------------------------------------
struct Final32 {
    const float* src; int Cin; const float* w /* [Cin][8] */; const float* bias /* [8] */; int cout;
    void* out; int out_kind; size_t nvox; size_t tstride, out_tstride; LabelArgs lab;
};
__global__ __launch_bounds__(256) void k32_final(const Final32 p) {
    const float* const src = reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.src) + blockIdx.z * p.tstride);
    const float* __restrict__ w = p.w;
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < p.nvox; v += (size_t)gridDim.x * 256) {
        float acc[8];
#pragma unroll
        for (int co = 0; co < 8; ++co) acc[co] = 0.f;
        for (int ci = 0; ci < p.Cin; ++ci) {
            const float x = src[(size_t)ci * p.nvox + v];
#pragma unroll
            for (int co = 0; co < 8; ++co) acc[co] = fmaf(x, w[ci * 8 + co], acc[co]);
        }
        float mx = -INFINITY;
#pragma unroll
        for (int co = 0; co < 8; ++co)
            if (co < p.cout) { acc[co] += p.bias[co]; mx = fmaxf(mx, acc[co]); }
        if (p.out_kind != SD_OUT_LOGITS_F32) {       // softmax(1): exp(x - max) / sum, full-precision expf and division
            float sum = 0.f;
#pragma unroll
            for (int co = 0; co < 8; ++co)
                if (co < p.cout) { acc[co] = expf(acc[co] - mx); sum += acc[co]; }
#pragma unroll
            for (int co = 0; co < 8; ++co) acc[co] = acc[co] / sum;
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
            uint8_t* const out = reinterpret_cast<uint8_t*>(p.out) + blockIdx.z * p.out_tstride;
#pragma unroll
            for (int co = 0; co < 8; ++co)
                if (co < p.cout) out[(size_t)co * p.nvox + v] = (uint8_t)(acc[co] * 255.f);     // prediction.py:864-865
        } else {
            float* const out = reinterpret_cast<float*>(reinterpret_cast<char*>(p.out) + blockIdx.z * p.out_tstride);
#pragma unroll
            for (int co = 0; co < 8; ++co)
                if (co < p.cout) out[(size_t)co * p.nvox + v] = acc[co];
        }
    }
}

template <int KZ, typename IN>
int launch_conv32(const Conv32& q, int N, hipStream_t s) {
    Conv32 p = q;
    constexpr int CO = 16;
    const dim3 blk(256);
    const int ncob = (p.Cout + CO - 1) / CO;
    // block shape by the width of the layer: 64 / 32 / 16 columns (the deep levels of a tile are narrow)
    if (p.W > 40) {
        p.nbx = (p.W + 63) / 64; p.nby = (p.H + 15) / 16; p.nbz = p.D;
        k32_conv<KZ, 1, 16, 16, CO, IN><<<dim3(p.nbx * p.nby * p.nbz, ncob, N), blk, 0, s>>>(p);
    } else if (p.W > 20) {
        p.nbx = (p.W + 31) / 32; p.nby = (p.H + 15) / 16; p.nbz = (p.D + 1) / 2;
        k32_conv<KZ, 2, 16, 8, CO, IN><<<dim3(p.nbx * p.nby * p.nbz, ncob, N), blk, 0, s>>>(p);
    } else {
        p.nbx = (p.W + 15) / 16; p.nby = (p.H + 15) / 16; p.nbz = (p.D + 3) / 4;
        k32_conv<KZ, 4, 16, 4, CO, IN><<<dim3(p.nbx * p.nby * p.nbz, ncob, N), blk, 0, s>>>(p);
    }
    return hipGetLastError() == hipSuccess ? SD_OK : SD_ERR_HIP;
}

inline int grid_for(size_t total) { return (int)std::min<size_t>((total + 255) / 256, 1u << 16); }

}  // namespace

struct sd_f32_model {
    std::vector<FOp> ops;
    int nbuf = 0;
    std::vector<int> bufC;
    float* blob = nullptr;
    std::vector<Dims3> dims;
    std::vector<size_t> buf_off;
    int final_cout = 0;
};

namespace {

int infer32(const sd_f32_model* m, int D, int H, int W, std::vector<Dims3>& dims) {
    dims.assign(m->nbuf, Dims3{});
    dims[0] = {D, H, W};
    for (const FOp& op : m->ops) {
        const sd_op_desc& d = op.d;
        switch (d.kind) {
        case SD_OP_CONV: {
            const Dims3 a = dims[d.src0];
            Dims3 o = a;
            if (d.src1 >= 0) {
                o = dims[d.src1];
                if (a.d < o.d || a.h < o.h || a.w < o.w) return sd_fail_msg(SD_ERR_INVALID, "merge conv: src0 smaller than src1");
            }
            if (o.d <= 0) return sd_fail_msg(SD_ERR_INVALID, "conv input not produced yet");
            dims[d.dst] = o;
            break;
        }
        case SD_OP_POOL: {
            const Dims3 a = dims[d.src0];
            dims[d.dst] = {d.kz == 2 ? (a.d + 1) / 2 : a.d, (a.h + 1) / 2, (a.w + 1) / 2};
            break;
        }
        case SD_OP_UPCONV: {
            const Dims3 a = dims[d.src0];
            dims[d.dst] = {a.d * d.kz, a.h * 2, a.w * 2};
            break;
        }
        default: break;
        }
    }
    return SD_OK;
}

// one range per buffer from its writer to its last reader; disjoint lifetimes share memory (first fit)
size_t plan32(const sd_f32_model* m, const std::vector<Dims3>& dims, std::vector<size_t>& off) {
    const int nb = m->nbuf, nops = (int)m->ops.size();
    off.assign(nb, 0);
    std::vector<size_t> bytes(nb, 0);
    for (int b = 1; b < nb; ++b) bytes[b] = rup_sz((size_t)dims[b].d * dims[b].h * dims[b].w * m->bufC[b] * 4, 256);
    std::vector<int> first(nb, nops), last(nb, -1);
    for (int i = 0; i < nops; ++i) {
        const sd_op_desc& d = m->ops[i].d;
        auto rd = [&](int b) { if (b > 0) last[b] = std::max(last[b], i); };
        rd(d.src0); rd(d.src1);
        if (d.kind != SD_OP_FINAL && d.kind != SD_OP_GROUPNORM && d.dst > 0) { first[d.dst] = std::min(first[d.dst], i); last[d.dst] = std::max(last[d.dst], i); }
    }
    const bool no_reuse = getenv("SD_KEEP_ALL") || getenv("SD_NO_WS_REUSE");     // layer-wise debugging reads buffers back
    std::vector<int> order;
    for (int b = 1; b < nb; ++b)
        if (last[b] >= 0 && bytes[b]) order.push_back(b);
    std::sort(order.begin(), order.end(), [&](int a, int b) { return first[a] != first[b] ? first[a] < first[b] : a < b; });
    std::vector<int> placed;
    size_t top = F32_SCRATCH;
    for (int b : order) {
        std::vector<std::pair<size_t, size_t>> busy;
        for (int q : placed)
            if (no_reuse || (first[q] <= last[b] && first[b] <= last[q])) busy.push_back({off[q], off[q] + bytes[q]});
        std::sort(busy.begin(), busy.end());
        size_t cur = F32_SCRATCH;
        for (const auto& r : busy) {
            if (cur + bytes[b] <= r.first) break;
            cur = std::max(cur, r.second);
        }
        off[b] = cur;
        top = std::max(top, cur + bytes[b]);
        placed.push_back(b);
    }
    return top;
}

}  // namespace

void f32_model_destroy(sd_f32_model* m) {
    if (!m) return;
    if (m->blob) (void)hipFree(m->blob);
    delete m;
}

int f32_model_create(const sd_op_desc* ops, int n_ops, const float* W, size_t n_floats, sd_f32_model** out) {
    sd_f32_model* m = new sd_f32_model();
    auto chk = [&](int64_t off, size_t n) { return off >= 0 && (size_t)off + n <= n_floats; };
    int nbuf = 1;
    for (int i = 0; i < n_ops; ++i) nbuf = std::max(nbuf, std::max(ops[i].src0, std::max(ops[i].src1, ops[i].dst)) + 1);
    m->nbuf = nbuf;
    m->bufC.assign(nbuf, 0);
    m->bufC[0] = 1;
    std::vector<float> blob;
    auto alloc = [&](size_t n) { size_t o = rup_sz(blob.size(), 64); blob.resize(o + n, 0.f); return o; };
    const char* err = nullptr;
#define F32_FAIL(msg) do { err = (msg); goto done; } while (0)
    for (int i = 0; i < n_ops; ++i) {
        FOp op;
        op.d = ops[i];
        const sd_op_desc& d = op.d;
        // eval-mode BatchNorm as the affine torch applies behind the convolution
        auto affine = [&](int cout, int pad) -> bool {
            op.bias_off = alloc(pad); op.alpha_off = alloc(pad); op.beta_off = alloc(pad);
            if (!chk(d.b_off, cout)) return false;
            for (int c = 0; c < cout; ++c) { blob[op.bias_off + c] = W[d.b_off + c]; blob[op.alpha_off + c] = 1.f; }
            if (d.norm == 1) {
                if (!chk(d.gamma_off, cout) || !chk(d.beta_off, cout) || !chk(d.mean_off, cout) || !chk(d.var_off, cout)) return false;
                for (int c = 0; c < cout; ++c) {
                    const float al = W[d.gamma_off + c] / std::sqrt(W[d.var_off + c] + d.eps);
                    blob[op.alpha_off + c] = al;
                    blob[op.beta_off + c] = W[d.beta_off + c] - W[d.mean_off + c] * al;
                }
            }
            return true;
        };
        switch (d.kind) {
        case SD_OP_CONV: {
            if (d.ky != 3 || d.kx != 3 || (d.kz != 1 && d.kz != 3)) F32_FAIL("conv: only 3x3x3 and 1x3x3 kernels");
            if (d.src0 < 0 || d.dst <= 0 || d.cout <= 0) F32_FAIL("conv: bad buffer ids");
            if (d.src0 == 0 && (d.cin0 != 1 || d.src1 >= 0)) F32_FAIL("first conv must have exactly one input channel");
            if (d.src0 > 0 && m->bufC[d.src0] != d.cin0) F32_FAIL("conv: cin0 does not match producer of src0");
            if (d.src1 >= 0 && m->bufC[d.src1] != d.cin1) F32_FAIL("conv: cin1 does not match producer of src1");
            const int cin = d.cin0 + (d.src1 >= 0 ? d.cin1 : 0), taps = d.kz * 9;
            if (!chk(d.w_off, (size_t)d.cout * cin * taps)) F32_FAIL("conv: weight offsets");
            op.CO = 16; op.ncob = (d.cout + op.CO - 1) / op.CO;
            if (!affine(d.cout, op.ncob * op.CO)) F32_FAIL("conv: bias / norm offsets");
            op.w_off = alloc((size_t)op.ncob * cin * taps * op.CO);
            for (int co = 0; co < d.cout; ++co)
                for (int ci = 0; ci < cin; ++ci)
                    for (int t = 0; t < taps; ++t)
                        blob[op.w_off + (((size_t)(co / op.CO) * cin + ci) * taps + t) * op.CO + co % op.CO] =
                            W[d.w_off + ((size_t)co * cin + ci) * taps + t];
            m->bufC[d.dst] = d.cout;
            break;
        }
        case SD_OP_POOL:
            if (d.src0 <= 0 || d.dst <= 0 || (d.kz != 1 && d.kz != 2)) F32_FAIL("pool: bad arguments");
            m->bufC[d.dst] = m->bufC[d.src0];
            break;
        case SD_OP_UPCONV: {
            if (d.src0 <= 0 || d.dst <= 0 || (d.kz != 1 && d.kz != 2)) F32_FAIL("upconv: bad arguments");
            if (m->bufC[d.src0] != d.cin0) F32_FAIL("upconv: cin0 does not match producer");
            const int taps = d.kz * 4;
            if (!chk(d.w_off, (size_t)d.cin0 * d.cout * taps)) F32_FAIL("upconv: weight offsets");
            op.CO = 4; op.ncob = (d.cout + 3) / 4;
            if (!affine(d.cout, op.ncob * 4)) F32_FAIL("upconv: bias / norm offsets");
            op.w_off = alloc((size_t)d.cin0 * op.ncob * 4 * taps);
            for (int ci = 0; ci < d.cin0; ++ci)
                for (int co = 0; co < d.cout; ++co)
                    for (int t = 0; t < taps; ++t)
                        blob[op.w_off + (((size_t)ci * op.ncob + co / 4) * 4 + co % 4) * taps + t] = W[d.w_off + ((size_t)ci * d.cout + co) * taps + t];
            m->bufC[d.dst] = d.cout;
            break;
        }
        case SD_OP_GROUPNORM: {
            if (d.src0 <= 0 || d.groups <= 0 || d.groups > 128) F32_FAIL("groupnorm: bad arguments");
            const int C = m->bufC[d.src0];
            if (C % d.groups) F32_FAIL("groupnorm: channels not divisible by groups");
            if (!chk(d.gamma_off, C) || !chk(d.beta_off, C)) F32_FAIL("groupnorm: offsets");
            op.alpha_off = alloc(C); op.beta_off = alloc(C);
            for (int c = 0; c < C; ++c) { blob[op.alpha_off + c] = W[d.gamma_off + c]; blob[op.beta_off + c] = W[d.beta_off + c]; }
            break;
        }
        case SD_OP_FINAL: {
            if (d.src0 <= 0 || d.cout <= 0 || d.cout > 8) F32_FAIL("final conv: 1..8 output classes supported");
            if (m->bufC[d.src0] != d.cin0) F32_FAIL("final: cin0 does not match producer");
            if (!chk(d.w_off, (size_t)d.cout * d.cin0) || !chk(d.b_off, d.cout)) F32_FAIL("final: weight offsets");
            op.w_off = alloc((size_t)d.cin0 * 8); op.bias_off = alloc(8);
            for (int co = 0; co < d.cout; ++co) {
                for (int ci = 0; ci < d.cin0; ++ci) blob[op.w_off + (size_t)ci * 8 + co] = W[d.w_off + (size_t)co * d.cin0 + ci];
                blob[op.bias_off + co] = W[d.b_off + co];
            }
            m->final_cout = d.cout;
            break;
        }
        default: F32_FAIL("unknown op kind");
        }
        m->ops.push_back(op);
    }
    if (m->ops.empty() || m->ops.back().d.kind != SD_OP_FINAL) F32_FAIL("the plan must end with SD_OP_FINAL");
    {
        hipError_t e = hipMalloc((void**)&m->blob, blob.size() * 4 + 256);
        if (e == hipSuccess) e = hipMemcpy(m->blob, blob.data(), blob.size() * 4, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            f32_model_destroy(m);
            return sd_fail_msg(e == hipErrorOutOfMemory ? SD_ERR_NOMEM : SD_ERR_HIP, hipGetErrorString(e));
        }
    }
done:
#undef F32_FAIL
    if (err) {
        f32_model_destroy(m);
        return sd_fail_msg(SD_ERR_INVALID, err);
    }
    *out = m;
    return SD_OK;
}

int f32_final_cout(const sd_f32_model* m) { return m->final_cout; }
int f32_buf_channels(const sd_f32_model* m, int b) { return (b >= 0 && b < m->nbuf) ? m->bufC[b] : 0; }

size_t f32_workspace_bytes(sd_f32_model* m, int D, int H, int W) {
    std::vector<Dims3> dims;
    std::vector<size_t> off;
    if (infer32(m, D, H, W, dims) != SD_OK) return 0;
    return rup_sz(plan32(m, dims, off), 256);
}

int f32_forward(sd_f32_model* m, const void* in_dev, int in_dtype, int N, int D, int H, int W, void* out_dev, int out_kind,
                const LabelArgs* lab, void* ws, size_t ws_bytes, hipStream_t s, hipEvent_t* ev) {
    int rc = infer32(m, D, H, W, m->dims);
    if (rc != SD_OK) return rc;
    const size_t tstride = rup_sz(plan32(m, m->dims, m->buf_off), 256);
    if ((size_t)N * tstride > ws_bytes) return sd_fail_msg(SD_ERR_NOMEM, "workspace too small");
    const size_t in_tstride = (size_t)D * H * W * (in_dtype == SD_U8 ? 1 : 4);
    const size_t out_tstride = out_kind == SD_OUT_LABELS_U8 ? (size_t)D * H * W
                                                            : (size_t)m->final_cout * D * H * W * (out_kind == SD_OUT_PROBS_U8 ? 1 : 4);
    char* const wsb = reinterpret_cast<char*>(ws);
    auto bufp = [&](int b) { return reinterpret_cast<float*>(wsb + m->buf_off[b]); };
    auto fp = [&](size_t off) { return m->blob + off; };
    for (size_t i = 0; i < m->ops.size(); ++i) {
        const FOp& op = m->ops[i];
        const sd_op_desc& d = op.d;
        if (ev && hipEventRecord(ev[i], s) != hipSuccess) return sd_fail_msg(SD_ERR_HIP, "hipEventRecord");
        switch (d.kind) {
        case SD_OP_CONV: {
            const Dims3 o = m->dims[d.dst], a = m->dims[d.src0];
            Conv32 p{};
            p.in0 = d.src0 == 0 ? in_dev : (const void*)bufp(d.src0);
            p.in0_is_input = d.src0 == 0; p.C0 = d.cin0; p.H0 = a.h; p.W0 = a.w; p.P0 = (size_t)a.d * a.h * a.w;
            if (d.src1 >= 0) {
                const Dims3 b = m->dims[d.src1];
                p.in1 = bufp(d.src1); p.C1 = d.cin1; p.H1 = b.h; p.W1 = b.w; p.P1 = (size_t)b.d * b.h * b.w;
            }
            p.dst = bufp(d.dst); p.Cout = d.cout; p.D = o.d; p.H = o.h; p.W = o.w;
            p.w = fp(op.w_off); p.bias = fp(op.bias_off); p.alpha = fp(op.alpha_off); p.beta = fp(op.beta_off);
            p.relu = d.relu; p.tstride = tstride; p.in_tstride = in_tstride;
            if (d.src0 == 0 && in_dtype == SD_U8) rc = d.kz == 3 ? launch_conv32<3, uint8_t>(p, N, s) : launch_conv32<1, uint8_t>(p, N, s);
            else rc = d.kz == 3 ? launch_conv32<3, float>(p, N, s) : launch_conv32<1, float>(p, N, s);
            break;
        }
        case SD_OP_POOL: {
            const Dims3 a = m->dims[d.src0], o = m->dims[d.dst];
            Pool32 p{bufp(d.src0), bufp(d.dst), m->bufC[d.src0], a.d, a.h, a.w, o.d, o.h, o.w, d.kz, tstride};
            k32_pool<<<dim3(grid_for((size_t)p.C * o.d * o.h * o.w), 1, N), 256, 0, s>>>(p);
            break;
        }
        case SD_OP_UPCONV: {
            const Dims3 a = m->dims[d.src0];
            Up32 p{};
            p.src = bufp(d.src0); p.Cin = d.cin0; p.D = a.d; p.H = a.h; p.W = a.w; p.dst = bufp(d.dst); p.Cout = d.cout; p.kz = d.kz;
            p.w = fp(op.w_off); p.bias = fp(op.bias_off); p.alpha = fp(op.alpha_off); p.beta = fp(op.beta_off);
            p.relu = d.relu; p.ncob = op.ncob; p.tstride = tstride;
            const dim3 g((unsigned)(((size_t)a.d * a.h * a.w + 255) / 256), op.ncob, N);
            if (d.kz == 2) k32_upconv<2><<<g, 256, 0, s>>>(p);
            else k32_upconv<1><<<g, 256, 0, s>>>(p);
            break;
        }
        case SD_OP_GROUPNORM: {
            const Dims3 a = m->dims[d.src0];
            const Dims3 r = d.src1 >= 0 ? m->dims[d.src1] : a;
            Gn32 p{};
            p.buf = bufp(d.src0); p.C = m->bufC[d.src0]; p.groups = d.groups;
            p.D = r.d; p.H = r.h; p.W = r.w; p.Hs = a.h; p.Ws = a.w; p.P = (size_t)a.d * a.h * a.w;
            p.gamma = fp(op.alpha_off); p.beta = fp(op.beta_off); p.eps = d.eps; p.relu = d.relu;
            p.sums = reinterpret_cast<double*>(wsb); p.tstride = tstride;
            for (int t = 0; t < N; ++t)
                if (hipMemsetAsync(wsb + (size_t)t * tstride, 0, (size_t)2 * d.groups * sizeof(double), s) != hipSuccess)
                    return sd_fail_msg(SD_ERR_HIP, "hipMemsetAsync (GroupNorm sums)");
            const size_t per_group = (size_t)r.d * r.h * r.w * (p.C / d.groups);
            k32_gn_stats<<<dim3((unsigned)std::min<size_t>((per_group + 4095) / 4096, 256), d.groups, N), 256, 0, s>>>(p);
            k32_gn_apply<<<dim3(grid_for((size_t)r.d * r.h * r.w * p.C), 1, N), 256, 0, s>>>(p);
            break;
        }
        case SD_OP_FINAL: {
            const Dims3 a = m->dims[d.src0];
            if (a.d != D || a.h != H || a.w != W) return sd_fail_msg(SD_ERR_INVALID, "final layer shape != input shape");
            Final32 p{};
            p.src = bufp(d.src0); p.Cin = d.cin0; p.w = fp(op.w_off); p.bias = fp(op.bias_off); p.cout = d.cout;
            p.out = out_dev; p.out_kind = out_kind; p.nvox = (size_t)D * H * W; p.tstride = tstride; p.out_tstride = out_tstride;
            if (lab) p.lab = *lab;
            k32_final<<<dim3(grid_for(p.nvox), 1, N), 256, 0, s>>>(p);
            break;
        }
        default: return sd_fail_msg(SD_ERR_INVALID, "unknown op kind");
        }
        if (rc != SD_OK || hipGetLastError() != hipSuccess) return sd_fail_msg(SD_ERR_HIP, "fp32 layer launch failed");
    }
    if (ev && hipEventRecord(ev[m->ops.size()], s) != hipSuccess) return sd_fail_msg(SD_ERR_HIP, "hipEventRecord");
    return SD_OK;
}

int f32_read_buffer(sd_f32_model* m, int buf, const void* ws, float* out, int32_t* dims4, hipStream_t s) {
    if (buf <= 0 || buf >= m->nbuf || m->dims.empty()) return sd_fail_msg(SD_ERR_INVALID, "sd_debug_read_buffer: bad argument");
    const Dims3 a = m->dims[buf];
    if (dims4) { dims4[0] = m->bufC[buf]; dims4[1] = a.d; dims4[2] = a.h; dims4[3] = a.w; }
    if (!out) return SD_OK;
    const size_t n = (size_t)m->bufC[buf] * a.d * a.h * a.w * 4;
    if (hipMemcpyAsync(out, reinterpret_cast<const char*>(ws) + m->buf_off[buf], n, hipMemcpyDeviceToDevice, s) != hipSuccess)
        return sd_fail_msg(SD_ERR_HIP, "sd_debug_read_buffer: copy failed");
    return SD_OK;
}
