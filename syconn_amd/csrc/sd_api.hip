// Host side of libsyconn_dense_hip.so: the C ABI of include/syconn_dense.h.  Owns the network plan, folds
// BatchNorm, packs weights into MFMA fragment order, plans the workspace and enqueues the gfx950 kernels.
#include "sd_internal.h"
#include "../../include/syconn_dense.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <thread>
#include <atomic>

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
#define HIP_TRY(expr)                                                                                       \
    do {                                                                                                    \
        hipError_t _e = (expr);                                                                             \
        if (_e != hipSuccess)                                                                               \
            return fail(_e == hipErrorOutOfMemory ? SD_ERR_NOMEM : SD_ERR_HIP,                              \
                        std::string(#expr) + ": " + hipGetErrorString(_e));                                 \
    } while (0)

inline int rup(int v, int m) { return (v + m - 1) / m * m; }
inline size_t rup_sz(size_t v, size_t m) { return (v + m - 1) / m * m; }

// float -> bf16 (round to nearest even) / fp16 bits
inline uint16_t f2bf16(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
inline uint16_t f2f16(float f) {
    _Float16 h = (_Float16)f;
    uint16_t b;
    std::memcpy(&b, &h, 2);
    return b;
}
inline uint16_t cvt(float f, int act_dtype) { return act_dtype == SD_BF16 ? f2bf16(f) : f2f16(f); }

struct Dims { int d = 0, h = 0, w = 0; };

struct Op {
    sd_op_desc d{};
    // device pointers (into the model's weight blob)
    size_t wpack_off = 0, bias_off = 0, aux_off = 0;   // byte offsets
    size_t w3_off = 0;     // first conv, planar, bf16 / fp16 plans: three-way bf16 split of the weights / 255 for uint8 input (0: none)
    int NT = 1, NB = 1;
    bool first = false;    // conv reading the network input (cin = 1)
    int fuse_pool = -1;    // index of the MaxPool op computed in this conv's epilogue
    int fuse_final = -1;   // index of the final 1x1x1 op computed in this conv's epilogue
    int fuse_first = -1;   // conv: index of the first (cin = 1) convolution whose output (this conv's src0) it computes itself
    int fuse_gn = -1;      // conv: index of the GroupNorm op whose statistics this conv's epilogue accumulates
    int fuse_pool_raw = -1;   // conv with fuse_gn: index of the MaxPool op (behind that GroupNorm) whose RAW pooling this conv's epilogue does
    size_t pooldir_off = 0;   //   its per-channel direction masks (gamma < 0: minimum) in the blob
    bool stats_done = false;   // groupnorm: statistics come from the producing conv
    int gn_pool = -1;          // groupnorm: index of the MaxPool op fused into the apply pass
    bool pool_in_conv = false;   // groupnorm (deferred, with gn_pool): the producing conv pools the RAW tensor, its readers apply the table
    bool gn_defer = false;       // groupnorm: statistics -> scale / shift only; every consumer of the buffer applies them itself
    size_t fwfrag_off = 0;             // fused final 1x1x1: hi/lo MFMA weight fragments
    float oscale = 1.f;                // split-fp16 plan: 2^-k, undoes the power of two the packed weights / bias of this layer carry
    float final_oscale = 1.f;          //   ... and the one of the fused final layer's fragments
    bool skipped = false;  // op is executed inside its producer
    // up-convolution heading a fused level-0 decoder (sd_dec0.hip): indices of the merge conv and the second conv (whose
    // fuse_final names the final layer); those ops are `skipped` and their output buffers are never materialised
    int dec0_c1 = -1, dec0_c2 = -1;
    bool in_dec0 = false;
};

}  // namespace

thread_local const char* sd_tls_kernel = nullptr;      // (sd_internal.h)
int sd_fail_msg(int code, const char* msg) { return fail(code, msg); }   // for the other translation units

struct sd_model {
    int device = 0;
    int act_dtype = SD_BF16;
    std::vector<Op> ops;
    int nbuf = 0;
    std::vector<int> bufC;    // real channels per buffer id
    std::vector<int> bufCp;   // padded channel stride
    std::vector<const char*> op_kernel;   // last forward: kernel symbol each op's launch used (static strings; nullptr: no launch of its own)
    std::vector<int> op_exec;             // last forward: index of the op whose launch computed this op (itself, or the op it is fused into); -1: not run
    char* dev_blob = nullptr; // packed weights
    size_t blob_bytes = 0;
    void* dev_zero = nullptr;
    int* dev_ovf = nullptr;   // fp16 range-guard flag (set by the final-layer kernels, read and cleared by sd_model_overflow)
    // last forward
    std::vector<Dims> dims;
    std::vector<size_t> buf_off;
    int profile_slots = 0;             // 0 = off; else ring of event sets, one per sd_forward
    long n_forward = 0;
    int last_launches = 0;             // ops of the last forward that ran as their own launch
    std::vector<hipEvent_t> events;    // [slot][n_ops + 1]
    unsigned long long* dev_clk = nullptr;   // [slot][n_ops][4] clock stamps of the convolution launches (ConvParams::clk)
    std::vector<char> ev_recorded;     // [slot][n_ops + 1]: event was recorded by the forward that used the slot (ops that run inside
                                       // another op's launch record none: every record is a packet between two kernels)
    int final_cout = 0;
    bool keep_all = false;   // SD_KEEP_ALL=1: also store activations that only feed a fused consumer (tests)
    // sd_model_set_roi: the part of the network output the caller keeps (z,y,x; hi exclusive) -- decoder ops then compute only
    // what that box depends on (forward_impl)
    bool roi_set = false;
    int roi_lo[3] = {0, 0, 0}, roi_hi[3] = {0, 0, 0};
    bool ws_reuse = true;    // activation buffers with disjoint lifetimes share workspace memory
    // deferred GroupNorm apply: buffer b holds the RAW tensor, its consumers apply scale / shift (+ReLU) on the fly from a
    // per-tile [2*Cp] float table that lives in the workspace at gn_tab_off[b] (buf_gn[b] = index of the GroupNorm op)
    std::vector<int> buf_gn;
    std::vector<size_t> gn_tab_off;
    size_t ws_base = 65536;  // first byte of the activation buffers (behind the scratch and the tables)
    sd_f32_model* f32 = nullptr;   // act_dtype == SD_F32: the reference-precision plan (sd_f32.hip) serves every call
    bool split = false;            // act_dtype == SD_F16X2 (sd_split.hip): every buffer holds 2 * Cp / 16 fp16 chunk planes [hi | lo]
};

namespace {

// f(0) .. f(n - 1) on up to 48 host threads (independent items writing disjoint memory)
template <class F>
static void parallel_for(int n, F f) {
    const int nt = std::max(1, std::min({n, 48, (int)std::thread::hardware_concurrency()}));
    if (nt == 1) { for (int i = 0; i < n; ++i) f(i); return; }
    std::atomic<int> next{0};
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t)
        th.emplace_back([&] { for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) f(i); });
    for (auto& x : th) x.join();
}
constexpr size_t WS_SCRATCH = 65536;  // GroupNorm sums (double[2*C]) + scale/shift (float[2*C]) at workspace start

// shape inference for an input tile; fills dims[] per buffer.
int infer_shapes(const sd_model* m, int D, int H, int W, std::vector<Dims>& dims) {
    dims.assign(m->nbuf, Dims{});
    dims[0] = {D, H, W};
    for (const Op& op : m->ops) {
        const sd_op_desc& d = op.d;
        switch (d.kind) {
        case SD_OP_CONV: {
            const Dims a = dims[d.src0];
            Dims o = a;
            if (d.src1 >= 0) {
                o = dims[d.src1];
                if (a.d < o.d || a.h < o.h || a.w < o.w) return fail(SD_ERR_INVALID, "merge conv: src0 smaller than src1");
            }
            if (o.d <= 0) return fail(SD_ERR_INVALID, "conv input not produced yet");
            dims[d.dst] = o;
            break;
        }
        case SD_OP_POOL: {
            const Dims a = dims[d.src0];
            dims[d.dst] = {d.kz == 2 ? (a.d + 1) / 2 : a.d, (a.h + 1) / 2, (a.w + 1) / 2};
            break;
        }
        case SD_OP_UPCONV: {
            const Dims a = dims[d.src0];
            dims[d.dst] = {a.d * d.kz, a.h * 2, a.w * 2};
            break;
        }
        case SD_OP_GROUPNORM:
        case SD_OP_FINAL: break;
        default: return fail(SD_ERR_INVALID, "unknown op kind");
        }
    }
    return SD_OK;
}

// Workspace layout for one tile.  Every activation buffer lives from the op that writes it to the last op that reads
// it; buffers whose lifetimes do not overlap share memory (first-fit over the buffers in order of their first write).
// A U-Net keeps only the encoder skip tensors alive across the bottleneck, so a 178x243x331 reference-geometry tile needs
// about a third of the sum of all activations -- which is what lets several such tiles run in one launch set.
// `reuse == false` (SD_KEEP_ALL / SD_NO_FUSE / SD_NO_WS_REUSE: layer-wise debugging reads buffers back after the
// forward) gives every buffer its own range.
// shapes the fused level-0 decoder serves (sd_dec0.hip: cursor arithmetic) and is worth it for: it walks x-strips of 64
// columns at the cost of full strips, so a width that fills its last strip badly is faster layer by layer (measured: 66
// columns = 2 strips at 52 % -> 0.26 vs 0.21 ms; 331 columns = 6 strips at 86 % -> 1.24 vs 1.36 ms).  Others run the layers.
// (ONE predicate for plan_workspace, forward_impl and launch_dec0: the kernel's own limits -- 32-bit stream positions, 24-bit
// row arithmetic, H >= 8 for the cursor wraps -- are part of it, so a shape that passes here never fails in the launcher after
// the intermediate buffers have been dropped from the workspace)
bool dec0_shape_ok(const Dims& o) {
    // (round 5: strips of 56 columns where they cover the width with fewer stream positions -- launch_dec0 picks; the fill rule
    // counts positions: strips x (width + 4), 68 per 64-column strip)
    const long cost = std::min<long>((long)((o.w + 63) / 64) * 68, (long)((o.w + 55) / 56) * 60);
    const long hp = 2 * (o.h / 2 + 1);
    return o.h >= 8 && (long)o.d * o.h < (1l << 24) && o.w < (1 << 24) && (long)o.d * hp * 68 <= (1l << 30) &&
           (long)o.d * o.h * o.w < (1l << 31) && (long)o.w * 10 * 68 >= cost * 64 * 7;
}

struct Box { int lo[3], hi[3]; bool any; };      // z,y,x; hi exclusive; any == false: "the whole tensor" / "nothing yet" by context
// the sub-box the fused level-0 decoder computes for the model's output box of interest: two voxel shells wider in y / x (its two
// 3x3 convolutions zero-pad at the box border), starting on even y / x (the level-1 tensor is addressed at half resolution)
static Box dec0_view(const sd_model* m, const Dims& o) {
    Box v;
    const int n[3] = {o.d, o.h, o.w};
    v.any = false;
    for (int a = 0; a < 3; ++a) {
        const int r = a == 0 ? 0 : 2;
        v.lo[a] = std::max(0, m->roi_lo[a] - r);
        if (a > 0) v.lo[a] &= ~1;
        v.hi[a] = std::min(n[a], m->roi_hi[a] + r);
        v.any |= v.lo[a] > 0 || v.hi[a] < n[a];
    }
    return v;
}
// with an output box of interest the fused level-0 decoder runs on its sub-box if that is a shape it serves; else its layers
// run one by one on theirs
static bool use_dec0(const sd_model* m, const Dims& o) {
    if (!dec0_shape_ok(o)) return false;
    if (!m->roi_set) return true;
    static const bool no_view = getenv("SD_ROI_NO_DEC0") != nullptr;
    if (no_view) return false;
    for (int a = 0; a < 3; ++a)
        if (m->roi_lo[a] < 0 || m->roi_hi[a] > (a == 0 ? o.d : a == 1 ? o.h : o.w) || m->roi_lo[a] >= m->roi_hi[a]) return false;
    const Box v = dec0_view(m, o);
    return dec0_shape_ok(Dims{v.hi[0] - v.lo[0], v.hi[1] - v.lo[1], v.hi[2] - v.lo[2]});
}

size_t plan_workspace(const sd_model* m, const std::vector<Dims>& dims, std::vector<size_t>& off) {
    const int nb = m->nbuf;
    const bool dec0 = use_dec0(m, dims[0]);
    off.assign(nb, 0);
    const size_t WS_BASE = m->ws_base;      // statistics scratch + the scale / shift tables of deferred GroupNorm applies
    std::vector<size_t> bytes(nb, 0);
    for (int b = 1; b < nb; ++b) bytes[b] = rup_sz((size_t)dims[b].d * dims[b].h * dims[b].w * m->bufCp[b] * (m->split ? 4 : 2), 256);
    if (!m->ws_reuse) {
        size_t cur = WS_BASE;
        for (int b = 1; b < nb; ++b) { off[b] = cur; cur += bytes[b]; }
        return cur;
    }
    const int nops = (int)m->ops.size();
    std::vector<int> first(nb, nops), last(nb, -1);
    auto touch_w = [&](int b, int i) { if (b > 0) { first[b] = std::min(first[b], i); last[b] = std::max(last[b], i); } };
    auto touch_r = [&](int b, int i) { if (b > 0) last[b] = std::max(last[b], i); };
    for (int i = 0; i < nops; ++i) {
        const Op& op = m->ops[i];
        const sd_op_desc& d = op.d;
        if (op.in_dec0 && dec0) continue;               // runs inside the up-convolution's launch, writes no buffer
        if (op.dec0_c1 >= 0 && dec0) {                  // reads the level-1 tensor and the skip tensor, writes the network output
            touch_r(d.src0, i);
            touch_r(m->ops[op.dec0_c1].d.src1, i);
            continue;
        }
        touch_r(d.src0, i);
        touch_r(d.src1, i);
        if (d.kind != SD_OP_FINAL) touch_w(d.dst, i);
        // outputs of ops executed inside this op's launch are written NOW, not at their own position in the plan
        if (op.fuse_pool >= 0) touch_w(m->ops[op.fuse_pool].d.dst, i);
        if (op.gn_pool >= 0) touch_w(m->ops[op.gn_pool].d.dst, i);
        if (op.fuse_pool_raw >= 0) touch_w(m->ops[op.fuse_pool_raw].d.dst, i);
        // a first convolution computed inside its consumer may also run as its own launch (decided per launch)
    }
    std::vector<int> order;
    for (int b = 1; b < nb; ++b)
        if (last[b] >= 0 && bytes[b]) order.push_back(b);
    std::sort(order.begin(), order.end(), [&](int a, int b) { return first[a] != first[b] ? first[a] < first[b] : a < b; });
    std::vector<int> placed;
    size_t top = WS_BASE;
    for (int b : order) {
        // candidate offsets: scratch end and the end of every placed buffer that is alive at the same time
        std::vector<std::pair<size_t, size_t>> busy;      // [begin, end) ranges this buffer must avoid
        for (int q : placed)
            if (first[q] <= last[b] && first[b] <= last[q]) busy.push_back({off[q], off[q] + bytes[q]});
        std::sort(busy.begin(), busy.end());
        size_t cur = WS_BASE;
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

extern "C" {

const char* sd_last_error(void) { return g_err.c_str(); }
const char* sd_version(void) { return "syconn_dense_hip 0.1 (gfx950)"; }

int sd_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int sd_init(int device_ordinal) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(SD_ERR_NODEVICE, "no HIP device visible; libsyconn_dense_hip has no CPU fallback");
    if (device_ordinal < 0 || device_ordinal >= n) return fail(SD_ERR_INVALID, "device ordinal out of range");
    HIP_TRY(hipSetDevice(device_ordinal));
    return SD_OK;
}

int sd_model_num_ops(const sd_model* m) { return m ? (int)m->ops.size() : 0; }
int sd_debug_last_launch_count(const sd_model* m) { return m ? m->last_launches : 0; }
int sd_debug_op_kernel(const sd_model* m, int op, char* buf, int n) {
    if (!m) return -1;
    if (m->f32) {      // (the fp32 FMA plan keeps its own op list: one k32_* launch per op)
        if (buf && n > 0) snprintf(buf, (size_t)n, "k32_* (fp32 FMA plan)");
        return op;
    }
    if (op < 0 || op >= (int)m->ops.size()) return -1;
    const int e = op < (int)m->op_exec.size() ? m->op_exec[op] : -1;
    if (buf && n > 0) {
        const char* name = (e >= 0 && e < (int)m->op_kernel.size() && m->op_kernel[e]) ? m->op_kernel[e] : "";
        snprintf(buf, (size_t)n, "%s", name);
    }
    return e;
}

int sd_model_create(const sd_op_desc* ops, int n_ops, const float* W, size_t n_floats, int act_dtype,
                    sd_model** out) {
    if (!ops || n_ops <= 0 || !W || !out) return fail(SD_ERR_INVALID, "null argument");
    if (act_dtype != SD_BF16 && act_dtype != SD_F16 && act_dtype != SD_F32 && act_dtype != SD_F16X2)
        return fail(SD_ERR_INVALID, "act_dtype must be SD_BF16, SD_F16, SD_F16X2 or SD_F32");
    sd_model* m = new sd_model();
    m->act_dtype = act_dtype;
    const bool split = act_dtype == SD_F16X2;
    m->split = split;
    const int pack_dtype = split ? SD_F16 : act_dtype;      // element type of the packed MFMA weight fragments
    if (act_dtype == SD_F32) {       // reference-precision mode: its own plan, weights and kernels
        if (hipGetDevice(&m->device) != hipSuccess) { delete m; return fail(SD_ERR_NODEVICE, "no current HIP device"); }
        const int rc32 = f32_model_create(ops, n_ops, W, n_floats, &m->f32);
        if (rc32 != SD_OK) { delete m; return rc32; }
        for (int i = 0; i < n_ops; ++i) { Op op; op.d = ops[i]; m->ops.push_back(op); }
        m->final_cout = f32_final_cout(m->f32);
        *out = m;
        return SD_OK;
    }
    m->keep_all = getenv("SD_KEEP_ALL") != nullptr;
    m->ws_reuse = !m->keep_all && !getenv("SD_NO_FUSE") && !getenv("SD_NO_WS_REUSE");
    if (hipGetDevice(&m->device) != hipSuccess) { delete m; return fail(SD_ERR_NODEVICE, "no current HIP device"); }

    auto chk = [&](int64_t off, size_t n) { return off >= 0 && (size_t)off + n <= n_floats; };
    int nbuf = 1;
    for (int i = 0; i < n_ops; ++i) {
        nbuf = std::max(nbuf, std::max(ops[i].src0, std::max(ops[i].src1, ops[i].dst)) + 1);
    }
    m->nbuf = nbuf;
    m->bufC.assign(nbuf, 0);
    m->bufCp.assign(nbuf, 0);
    m->bufC[0] = 1;
    m->bufCp[0] = 1;

    std::vector<char> blob;  // host image of the packed weights
    auto blob_alloc = [&](size_t bytes) { size_t o = rup_sz(blob.size(), 256); blob.resize(o + bytes, 0); return o; };
    int rc = SD_OK;
    std::string err;
#define MODEL_FAIL(msg) do { err = (msg); rc = SD_ERR_INVALID; goto done; } while (0)

    for (int i = 0; i < n_ops; ++i) {
        Op op;
        op.d = ops[i];
        const sd_op_desc& d = op.d;
        // folded per-output-channel scale / shift of an eval-mode BatchNorm
        std::vector<float> sc, sh;
        auto fold_bn = [&](int cout) -> bool {
            sc.assign(cout, 1.f);
            sh.assign(cout, 0.f);
            if (d.norm == 1) {
                if (!chk(d.gamma_off, cout) || !chk(d.beta_off, cout) || !chk(d.mean_off, cout) || !chk(d.var_off, cout))
                    return false;
                for (int c = 0; c < cout; ++c) {
                    const float s = W[d.gamma_off + c] / std::sqrt(W[d.var_off + c] + d.eps);
                    sc[c] = s;
                    sh[c] = W[d.beta_off + c] - W[d.mean_off + c] * s;
                }
            }
            return true;
        };
        // split-fp16 plan: power of two 2^k that brings the largest folded weight of the layer into [2^14, 2^15) -- the lo parts
        // of all weights that matter are then normal fp16 numbers (hi + lo carries 22+ bits); bias and epilogue follow (exact)
        auto split_scale = [&](float maxabs) -> float {
            if (!split || !(maxabs > 0.f) || !std::isfinite(maxabs)) return 1.f;
            const int k = std::max(-60, std::min(60, 14 - std::ilogb(maxabs)));
            return std::ldexp(1.f, k);
        };
        // element of virtual chunk `r` (0 .. 3n-1) of an n-chunk split input: planes [hi | hi | lo] meet weights [lo | hi | hi]
        auto split_part = [&](float v, int r, int n) -> uint16_t {
            const uint16_t hi = f2f16(v);
            if (r >= n) return hi;
            _Float16 h; std::memcpy(&h, &hi, 2);
            return f2f16(v - (float)h);
        };
        switch (d.kind) {
        case SD_OP_CONV: {
            if (d.ky != 3 || d.kx != 3 || (d.kz != 1 && d.kz != 3)) MODEL_FAIL("conv: only 3x3x3 and 1x3x3 kernels");
            if (d.src0 < 0 || d.dst <= 0 || d.cout <= 0) MODEL_FAIL("conv: bad buffer ids");
            const int cin = d.cin0 + (d.src1 >= 0 ? d.cin1 : 0);
            const int taps = d.kz * 9;
            if (!chk(d.w_off, (size_t)d.cout * cin * taps) || !chk(d.b_off, d.cout)) MODEL_FAIL("conv: weight offsets");
            if (!fold_bn(d.cout)) MODEL_FAIL("conv: norm offsets");
            const int Cd = rup(d.cout, SD_CHUNK);
            m->bufC[d.dst] = d.cout;
            m->bufCp[d.dst] = Cd;
            const float* w = W + d.w_off;
            auto wat = [&](int co, int ci, int kz, int t9) {
                return w[((size_t)co * cin + ci) * taps + kz * 9 + t9] * sc[co];
            };
            if (d.src0 == 0) {
                if (d.cin0 != 1 || d.src1 >= 0) MODEL_FAIL("first conv must have exactly one input channel");
                op.first = true;
                const int ntile = (Cd + 31) / 32, nstep = (taps + 1) / 2;
                op.wpack_off = blob_alloc((size_t)ntile * nstep * 64 * 4);
                op.bias_off = blob_alloc((size_t)ntile * 32 * 4 + 16);
                float* wp = reinterpret_cast<float*>(blob.data() + op.wpack_off);
                float* bp = reinterpret_cast<float*>(blob.data() + op.bias_off);
                for (int nt = 0; nt < ntile; ++nt)
                    for (int s = 0; s < nstep; ++s)
                        for (int l = 0; l < 64; ++l) {
                            // (taps is odd: the last k slot of the MFMA chain is free and carries the folded BIAS -- the kernels feed it
                            // a 1.0, which makes the bias the last fma of the chain = the separate add it replaces, bit for bit)
                            const int n = nt * 32 + (l & 31), tap = 2 * s + (l >> 5);
                            wp[(nt * nstep + s) * 64 + l] = (n < d.cout && tap < taps) ? wat(n, 0, tap / 9, tap % 9)
                                                            : (n < d.cout && tap == taps) ? W[d.b_off + n] * sc[n] + sh[n] : 0.f;
                        }
                for (int n = 0; n < d.cout; ++n) bp[n] = W[d.b_off + n] * sc[n] + sh[n];
                if (d.kz == 1 && !split && act_dtype != SD_F32) {
                    // uint8 input on the bf16 matrix pipe (k_conv_first / k_conv_mfma MODE 5): a uint8 voxel v is EXACT in bf16, so
                    // conv(v / 255) = sum_t v_t * W'_t with W' = fp32(w / 255) carried as THREE bf16 parts (hi + mid + lo = all 24
                    // mantissa bits; every product v * part is exact in fp32) -- 9 taps x 3 parts + 3 parts of the folded bias
                    // (meeting a 1.0) are 30 of the 32 k slots of TWO v_mfma_f32_32x32x16_bf16 instead of five 64-cycle
                    // v_mfma_f32_32x32x2_f32.  Slot order (what the kernels' B operand looks like): lanes 0-31 hold taps 0-4,
                    // lanes 32-63 taps 5-8 and the constant; B dwords = [(t0,t0) (t1,t1) (t2,t2) (t3,t3)] [(t4,t4) (t0,t1) (t2,t3)
                    // (t4,1)] with a dword (a,b) = k slots (2i, 2i+1): a pair (t,t) meets (hi, mid), the mixed pairs meet lo parts.
                    op.w3_off = blob_alloc((size_t)ntile * 2 * 64 * 8 * 2);
                    uint16_t* w3 = reinterpret_cast<uint16_t*>(blob.data() + op.w3_off);
                    auto part3 = [&](float v, uint16_t (&o)[3]) {
                        float r = v;
                        for (int k = 0; k < 3; ++k) {
                            o[k] = f2bf16(r);
                            uint32_t u = (uint32_t)o[k] << 16; float h; std::memcpy(&h, &u, 4);
                            r -= h;      // exact
                        }
                    };
                    for (int nt = 0; nt < ntile; ++nt)
                        for (int l = 0; l < 64; ++l) {
                            const int n = nt * 32 + (l & 31), hf = l >> 5;
                            uint16_t tp[5][3] = {}, bpart[3] = {};
                            if (n < d.cout) {
                                for (int a = 0; a < 5; ++a) {
                                    const int tap = hf * 5 + a;
                                    if (tap < 9) part3((float)((double)wat(n, 0, 0, tap) / 255.0), tp[a]);
                                }
                                part3(W[d.b_off + n] * sc[n] + sh[n], bpart);
                            }
                            uint16_t* m0 = w3 + ((size_t)(nt * 2 + 0) * 64 + l) * 8;
                            uint16_t* m1 = w3 + ((size_t)(nt * 2 + 1) * 64 + l) * 8;
                            for (int a = 0; a < 4; ++a) { m0[2 * a] = tp[a][0]; m0[2 * a + 1] = tp[a][1]; }
                            if (hf == 0) {
                                m1[0] = tp[4][0]; m1[1] = tp[4][1];
                                m1[2] = tp[0][2]; m1[3] = tp[1][2]; m1[4] = tp[2][2]; m1[5] = tp[3][2]; m1[6] = tp[4][2]; m1[7] = 0;
                            } else {      // the fifth read of the upper lanes is the constant (1.0, 1.0), and so is their last dword
                                m1[0] = bpart[0]; m1[1] = bpart[1];
                                m1[2] = tp[0][2]; m1[3] = tp[1][2]; m1[4] = tp[2][2]; m1[5] = tp[3][2]; m1[6] = bpart[2]; m1[7] = 0;
                            }
                        }
                }
            } else {
                if (m->bufC[d.src0] != d.cin0) MODEL_FAIL("conv: cin0 does not match producer of src0");
                if (d.src1 >= 0 && m->bufC[d.src1] != d.cin1) MODEL_FAIL("conv: cin1 does not match producer of src1");
                const int nch0 = m->bufCp[d.src0] / SD_CHUNK, nch1 = d.src1 >= 0 ? m->bufCp[d.src1] / SD_CHUNK : 0;
                const int ntile = (d.cout + 31) / 32;
                // output channels per workgroup: 64 (NT=2); 96 (NT=3) where that divides the layer without padding tiles
                // (the 48-filter family: 96/192/384/768 channels) -- a 64-wide split would spend 25 % of the MFMAs of a
                // 96-channel layer on zero columns; NT=3 also needs fewer LDS fragment reads per MFMA (5 per 6)
                static const bool no_nt3 = getenv("SD_NO_NT3") != nullptr;
                op.NT = (ntile % 3 == 0 && !no_nt3) ? 3 : (ntile >= 2 ? 2 : 1);
                op.NB = (ntile + op.NT - 1) / op.NT;
                const int nchunks = (nch0 + nch1) * (split ? 3 : 1);      // (split-fp16 plan: virtual chunks)
                float wscale = 1.f;
                if (split) {
                    float mx = 0.f;
                    for (int co = 0; co < d.cout; ++co)
                        for (size_t k = 0; k < (size_t)cin * taps; ++k) mx = std::max(mx, std::fabs(w[(size_t)co * cin * taps + k] * sc[co]));
                    wscale = split_scale(mx);
                    op.oscale = 1.f / wscale;
                }
                const size_t gbytes = (size_t)9 * op.NT * 1024;
                op.wpack_off = blob_alloc((size_t)op.NB * nchunks * d.kz * gbytes);
                op.bias_off = blob_alloc((size_t)op.NB * op.NT * 32 * 4 + 16);
                uint16_t* wp = reinterpret_cast<uint16_t*>(blob.data() + op.wpack_off);
                float* bp = reinterpret_cast<float*>(blob.data() + op.bias_off);
                // (the packing of a 7.8 M-parameter net took 0.15 s on one core -- 15 % of a 1024 x 1024 x 256 file-system-mode job:
                // the (block, chunk) groups are independent)
                parallel_for(op.NB * nchunks, [&](int g0) {
                    const int nb = g0 / nchunks, c = g0 % nchunks;
                        // weight groups in the order the kernel's stages read them: [kz][9 taps (ky, kx)]; 3x3x3 layers with NT <= 2 (ZROLL in
                        // sd_conv_mfma.h): [stage ky][step (kx, kz)] -- those outputs are summed in the order (chunk, ky, kx, kz)
                        for (int kz = 0; kz < d.kz; ++kz)
                            for (int t9 = 0; t9 < 9; ++t9)
                                for (int j = 0; j < op.NT; ++j)
                                    for (int l = 0; l < 64; ++l)
                                        for (int e = 0; e < 8; ++e) {
                                            const int n = (nb * op.NT + j) * 32 + (l & 31);
                                            int ci;  // index into the concatenated weight input-channel axis
                                            const int m0 = split ? 3 : 1;
                                            const bool s0 = c < nch0 * m0;
                                            const int r = s0 ? c : c - nch0 * m0, nsrc = s0 ? nch0 : nch1;       // chunk within its source
                                            {
                                                const int k = (r % nsrc) * SD_CHUNK + (l >> 5) * 8 + e;
                                                ci = s0 ? (k < d.cin0 ? k : -1) : (k < d.cin1 ? d.cin0 + k : -1);
                                            }
                                            // (ZROLL layers: group index `kz` and in-group index `t9` name the tap (kz, ky, kx) = (t9 % 3, kz, t9 / 3))
                                            const float v = (n < d.cout && ci >= 0) ? (d.kz == 3 && op.NT != 3 ? wat(n, ci, t9 % 3, kz * 3 + t9 / 3) : wat(n, ci, kz, t9)) * wscale : 0.f;
                                            const size_t idx =
                                                ((((((size_t)nb * nchunks + c) * d.kz + kz) * 9 + t9) * op.NT + j) * 64 + l) * 8 + e;
                                            wp[idx] = split ? split_part(v, r, nsrc) : cvt(v, act_dtype);
                                        }
                });
                for (int n = 0; n < d.cout; ++n) bp[n] = (W[d.b_off + n] * sc[n] + sh[n]) * wscale;
            }
            break;
        }
        case SD_OP_POOL: {
            if (d.src0 <= 0 || d.dst <= 0 || (d.kz != 1 && d.kz != 2)) MODEL_FAIL("pool: bad arguments");
            m->bufC[d.dst] = m->bufC[d.src0];
            m->bufCp[d.dst] = m->bufCp[d.src0];
            break;
        }
        case SD_OP_UPCONV: {
            if (d.src0 <= 0 || d.dst <= 0 || (d.kz != 1 && d.kz != 2)) MODEL_FAIL("upconv: bad arguments");
            if (m->bufC[d.src0] != d.cin0) MODEL_FAIL("upconv: cin0 does not match producer");
            const int taps = d.kz * 4;
            if (!chk(d.w_off, (size_t)d.cin0 * d.cout * taps) || !chk(d.b_off, d.cout)) MODEL_FAIL("upconv: weight offsets");
            if (!fold_bn(d.cout)) MODEL_FAIL("upconv: norm offsets");
            const int Cd = rup(d.cout, SD_CHUNK);
            m->bufC[d.dst] = d.cout;
            m->bufCp[d.dst] = Cd;
            const int ntot = taps * Cd;
            op.NT = 2;
            op.NB = (ntot + 63) / 64;
            const int nreal = m->bufCp[d.src0] / SD_CHUNK;
            const int nchunk = nreal * (split ? 3 : 1);      // (split-fp16 plan: virtual chunks)
            float wscale = 1.f;
            if (split) {
                float mx = 0.f;
                for (int ci = 0; ci < d.cin0; ++ci)
                    for (int co = 0; co < d.cout; ++co)
                        for (int t = 0; t < taps; ++t) mx = std::max(mx, std::fabs(W[d.w_off + ((size_t)ci * d.cout + co) * taps + t] * sc[co]));
                wscale = split_scale(mx);
                op.oscale = 1.f / wscale;
            }
            op.wpack_off = blob_alloc((size_t)op.NB * nchunk * 2 * 64 * 8 * 2);
            op.bias_off = blob_alloc((size_t)op.NB * 64 * 4 + 16);
            uint16_t* wp = reinterpret_cast<uint16_t*>(blob.data() + op.wpack_off);
            float* bp = reinterpret_cast<float*>(blob.data() + op.bias_off);
            const float* w = W + d.w_off;
            for (int nb = 0; nb < op.NB; ++nb)
                for (int c = 0; c < nchunk; ++c)
                    for (int j = 0; j < 2; ++j)
                        for (int l = 0; l < 64; ++l)
                            for (int e = 0; e < 8; ++e) {
                                const int n = (nb * 2 + j) * 32 + (l & 31);
                                const int tap = n / Cd, co = n % Cd;
                                const int ci = (c % nreal) * SD_CHUNK + (l >> 5) * 8 + e;
                                float v = 0.f;
                                if (tap < taps && co < d.cout && ci < d.cin0)
                                    v = w[((size_t)ci * d.cout + co) * taps + tap] * sc[co] * wscale;
                                wp[((((size_t)nb * nchunk + c) * 2 + j) * 64 + l) * 8 + e] = split ? split_part(v, c, nreal) : cvt(v, act_dtype);
                            }
            for (int n = 0; n < ntot; ++n) {
                const int co = n % Cd;
                bp[n] = co < d.cout ? (W[d.b_off + co] * sc[co] + sh[co]) * wscale : 0.f;
            }
            break;
        }
        case SD_OP_GROUPNORM: {
            if (d.src0 <= 0 || d.groups <= 0) MODEL_FAIL("groupnorm: bad arguments");
            const int C = m->bufC[d.src0], Cp = m->bufCp[d.src0];
            if (C % d.groups) MODEL_FAIL("groupnorm: channels not divisible by groups");
            if ((size_t)Cp * 16 > SD_GN_SCALE_OFF || SD_GN_SCALE_OFF + (size_t)Cp * 8 > WS_SCRATCH) MODEL_FAIL("groupnorm: too many channels");
            if (!chk(d.gamma_off, C) || !chk(d.beta_off, C)) MODEL_FAIL("groupnorm: offsets");
            op.aux_off = blob_alloc((size_t)2 * Cp * 4);
            float* gp = reinterpret_cast<float*>(blob.data() + op.aux_off);
            for (int c = 0; c < C; ++c) { gp[c] = W[d.gamma_off + c]; gp[Cp + c] = W[d.beta_off + c]; }
            break;
        }
        case SD_OP_FINAL: {
            if (d.src0 <= 0 || d.cout <= 0 || d.cout > 8) MODEL_FAIL("final conv: 1..8 output classes supported");
            if (m->bufC[d.src0] != d.cin0) MODEL_FAIL("final: cin0 does not match producer");
            if (!chk(d.w_off, (size_t)d.cout * d.cin0) || !chk(d.b_off, d.cout)) MODEL_FAIL("final: weight offsets");
            const int Cs = m->bufCp[d.src0];
            op.wpack_off = blob_alloc((size_t)8 * Cs * 4);
            op.bias_off = blob_alloc(8 * 4);
            float* wp = reinterpret_cast<float*>(blob.data() + op.wpack_off);
            float* bp = reinterpret_cast<float*>(blob.data() + op.bias_off);
            for (int co = 0; co < d.cout; ++co) {
                for (int ci = 0; ci < d.cin0; ++ci) wp[co * Cs + ci] = W[d.w_off + (size_t)co * d.cin0 + ci];
                bp[co] = W[d.b_off + co];
            }
            // the same weights as hi + lo parts in the activation dtype, MFMA A-fragment order, natural channel order
            // (k_final_mfma): row = class (lane & 31), element jj of k-step c <-> channel 16c + 8*(lane>>5) + jj
            {
                const int nch = Cs / SD_CHUNK;
                op.aux_off = blob_alloc((size_t)nch * 2 * 64 * 8 * 2);
                uint16_t* fp = reinterpret_cast<uint16_t*>(blob.data() + op.aux_off);
                auto back = [&](uint16_t bits) -> float {
                    if (pack_dtype == SD_BF16) { uint32_t u = (uint32_t)bits << 16; float f; std::memcpy(&f, &u, 4); return f; }
                    _Float16 h; std::memcpy(&h, &bits, 2); return (float)h;
                };
                for (int c = 0; c < nch; ++c)
                    for (int l = 0; l < 64; ++l)
                        for (int jj = 0; jj < 8; ++jj) {
                            const int co = l & 31, ch = c * SD_CHUNK + 8 * (l >> 5) + jj;
                            const float wv = (co < d.cout && ch < d.cin0) ? W[d.w_off + (size_t)co * d.cin0 + ch] : 0.f;
                            const uint16_t hi = cvt(wv, pack_dtype);
                            fp[((size_t)(c * 2 + 0) * 64 + l) * 8 + jj] = hi;
                            fp[((size_t)(c * 2 + 1) * 64 + l) * 8 + jj] = cvt(wv - back(hi), pack_dtype);
                        }
            }
            m->final_cout = d.cout;
            break;
        }
        default: MODEL_FAIL("unknown op kind");
        }
        m->ops.push_back(op);
    }
    if (m->ops.empty() || m->ops.back().d.kind != SD_OP_FINAL) MODEL_FAIL("the plan must end with SD_OP_FINAL");
    m->buf_gn.assign(m->nbuf, -1);
    m->gn_tab_off.assign(m->nbuf, 0);
    m->ws_base = WS_SCRATCH;
    // epilogue fusions (SD_NO_FUSE=1 keeps every layer a separate launch, for layer-wise debugging; the split-fp16 plan runs
    // every op as its own launch)
    if (!getenv("SD_NO_FUSE") && split) {
        // split-fp16 plan: first conv (1 -> 32, 1x3x3) -> conv (1x3x3) of one input: the second conv computes the hi / lo halo
        // planes of the first conv's output itself (k_conv_mfma MODE 4; decided per launch)
        for (size_t i = 0; i + 1 < m->ops.size(); ++i) {
            Op& f = m->ops[i];
            Op& c = m->ops[i + 1];
            if (f.d.kind == SD_OP_CONV && f.first && f.d.kz == 1 && m->bufCp[f.d.dst] == 32 &&      // (17 ... 32 filters: padding channels have zero weights and bias)
                c.d.kind == SD_OP_CONV && !c.first && c.d.kz == 1 && c.d.src0 == f.d.dst && c.d.src1 < 0 &&
                !getenv("SD_NO_FIRST_FUSE")) {
                bool other_reader = false;
                for (size_t k = i + 2; k < m->ops.size(); ++k)
                    if (m->ops[k].d.src0 == f.d.dst || m->ops[k].d.src1 == f.d.dst) other_reader = true;
                if (!other_reader) c.fuse_first = (int)i;
            }
        }
        // GroupNorm (whole buffer) directly followed by the pooling of its output: one pass
        for (size_t i = 0; i + 1 < m->ops.size(); ++i) {
            Op& g = m->ops[i];
            Op& nx = m->ops[i + 1];
            if (g.d.kind == SD_OP_GROUPNORM && g.d.src1 < 0 && nx.d.kind == SD_OP_POOL && nx.d.src0 == g.d.src0 &&
                !getenv("SD_NO_GN_FUSE")) {
                g.gn_pool = (int)(i + 1);
                nx.skipped = true;
            }
        }
        // a GroupNorm whose only reader is the final layer: statistics -> scale / shift table only, k_final_split applies them (+ReLU)
        // in fp32 on the exact values it reads anyway -- the normalised tensor is neither written nor re-read
        if (!getenv("SD_NO_GN_DEFER") && !m->keep_all) {
            for (size_t i = 0; i < m->ops.size(); ++i) {
                Op& g = m->ops[i];
                if (g.d.kind != SD_OP_GROUPNORM) continue;
                const int b = g.d.src0;
                bool ok = true, any = false;
                for (size_t k = i + 1; k < m->ops.size(); ++k) {
                    const Op& r = m->ops[k];
                    if (r.d.src0 != b && r.d.src1 != b) continue;
                    any = true;
                    if (r.d.kind != SD_OP_FINAL) ok = false;
                }
                if (ok && any) { g.gn_defer = true; m->buf_gn[b] = (int)i; }
            }
            for (int b = 1; b < m->nbuf; ++b)
                if (m->buf_gn[b] >= 0) {
                    m->gn_tab_off[b] = m->ws_base;
                    m->ws_base += rup_sz((size_t)2 * m->bufCp[b] * 4, 256);
                }
        }
        // ... the pooling behind a convolution runs in that convolution's epilogue (on the fp32 values, any sign)
        for (size_t i = 0; i + 1 < m->ops.size(); ++i) {
            Op& c = m->ops[i];
            Op& nx = m->ops[i + 1];
            // GroupNorm right behind a conv over the conv's whole output: statistics in the conv epilogue
            if (c.d.kind == SD_OP_CONV && !c.first && c.NT >= 2 && nx.d.kind == SD_OP_GROUPNORM && nx.d.src0 == c.d.dst &&
                nx.d.src1 < 0 && !getenv("SD_NO_GN_FUSE")) {
                c.fuse_gn = (int)(i + 1);
                nx.stats_done = true;
                continue;
            }
            if (c.d.kind == SD_OP_CONV && !c.first && nx.d.kind == SD_OP_POOL && nx.d.src0 == c.d.dst &&
                (c.d.kz == 3) == (nx.d.kz == 2)) {
                c.fuse_pool = (int)(i + 1);
                nx.skipped = true;
            } else if (c.d.kind == SD_OP_CONV && !c.first && c.d.kz == 1 && nx.d.kind == SD_OP_FINAL && nx.d.src0 == c.d.dst &&
                       c.NB == 1 && i + 2 == m->ops.size() && !getenv("SD_SPLIT_NO_FINAL_FUSE")) {
                // ... and the final 1x1x1 behind the last convolution: fp32 weights as scaled fp16 hi + lo fragments in MFMA
                // A-fragment order (row = class (lane & 31), k-step s / element jj <-> channel j*32 + 8*(2s + (jj>>2)) + 4*(lane>>5) + (jj&3))
                c.fuse_final = (int)(i + 1);
                nx.skipped = true;
                const sd_op_desc& fd = nx.d;
                float mx = 0.f;
                for (size_t k = 0; k < (size_t)fd.cout * fd.cin0; ++k) mx = std::max(mx, std::fabs(W[fd.w_off + k]));
                float fscale = 1.f;
                if (mx > 0.f && std::isfinite(mx)) fscale = std::ldexp(1.f, std::max(-60, std::min(60, 14 - std::ilogb(mx))));
                c.final_oscale = 1.f / fscale;
                c.fwfrag_off = blob_alloc((size_t)c.NT * 2 * 2 * 64 * 8 * 2);
                uint16_t* fp = reinterpret_cast<uint16_t*>(blob.data() + c.fwfrag_off);
                for (int j = 0; j < c.NT; ++j)
                    for (int s2 = 0; s2 < 2; ++s2)
                        for (int l = 0; l < 64; ++l)
                            for (int jj = 0; jj < 8; ++jj) {
                                const int co = l & 31, ch = j * 32 + 8 * (2 * s2 + (jj >> 2)) + 4 * (l >> 5) + (jj & 3);
                                const float wv = (co < fd.cout && ch < fd.cin0) ? W[fd.w_off + (size_t)co * fd.cin0 + ch] * fscale : 0.f;
                                const uint16_t hi = f2f16(wv);
                                _Float16 h; std::memcpy(&h, &hi, 2);
                                fp[((size_t)((j * 2 + s2) * 2 + 0) * 64 + l) * 8 + jj] = hi;
                                fp[((size_t)((j * 2 + s2) * 2 + 1) * 64 + l) * 8 + jj] = f2f16(wv - (float)h);
                            }
            }
        }
    }
    if (!getenv("SD_NO_FUSE") && !split) {
        // first conv (1 -> 32, 1x3x3) -> conv (1x3x3) of one input: the second conv computes its halo of the first conv's
        // output on the fly (decided per launch: only the resident-weight form of the kernel can do it)
        for (size_t i = 0; i + 1 < m->ops.size(); ++i) {
            Op& f = m->ops[i];
            Op& c = m->ops[i + 1];
            if (f.d.kind == SD_OP_CONV && f.first && f.d.kz == 1 &&
                (m->bufCp[f.d.dst] == 32 || m->bufCp[f.d.dst] == 48) &&      // (17 ... 32 / 33 ... 48 filters: padding channels have zero weights and bias)
                c.d.kind == SD_OP_CONV && !c.first && c.d.kz == 1 && c.d.src0 == f.d.dst && c.d.src1 < 0 &&
                !getenv("SD_NO_FIRST_FUSE")) {
                bool other_reader = false;
                for (size_t k = i + 2; k < m->ops.size(); ++k)
                    if (m->ops[k].d.src0 == f.d.dst || m->ops[k].d.src1 == f.d.dst) other_reader = true;
                if (!other_reader) c.fuse_first = (int)i;
            }
        }
        // GroupNorm (whole buffer) directly followed by the pooling of its output: one pass
        for (size_t i = 0; i + 1 < m->ops.size(); ++i) {
            Op& g = m->ops[i];
            Op& nx = m->ops[i + 1];
            if (g.d.kind == SD_OP_GROUPNORM && g.d.src1 < 0 && nx.d.kind == SD_OP_POOL && nx.d.src0 == g.d.src0 &&
                !getenv("SD_NO_GN_FUSE")) {
                g.gn_pool = (int)(i + 1);
                nx.skipped = true;
            }
        }
        // Deferred GroupNorm apply: the op only turns the statistics into a scale / shift table; the tensor stays raw and
        // every reader normalises on the fly (convolutions in their LDS halo, up-convolutions and the final layer on their
        // register fragments, a pooling fused into the GroupNorm op writes the normalised pooled tensor only) -- the
        // normalised tensor is never written nor re-read.  Possible when every later reader of the buffer is such an op.
        if (!getenv("SD_NO_GN_FUSE") && !getenv("SD_NO_GN_DEFER") && !m->keep_all) {
            for (size_t i = 0; i < m->ops.size(); ++i) {
                Op& g = m->ops[i];
                if (g.d.kind != SD_OP_GROUPNORM) continue;
                const int b = g.d.src0;
                bool ok = true, any = false;
                for (size_t k = i + 1; k < m->ops.size() && ok; ++k) {
                    const Op& r = m->ops[k];
                    if (r.d.src0 != b && r.d.src1 != b) continue;
                    any = true;
                    if (r.d.kind == SD_OP_CONV) ok = !r.first && m->bufCp[b] <= 256;      // LDS table: 128 B per 16 channels and tile
                    else if (r.d.kind == SD_OP_UPCONV)      // only the row kernel absorbs the apply cheaply; else keep the pass
                        ok = upconv_rows_kernel(m->bufCp[b] / SD_CHUNK, rup(r.d.cout, SD_CHUNK));
                    else if (r.d.kind == SD_OP_FINAL) ok = true;
                    else if (r.d.kind == SD_OP_POOL) ok = (g.gn_pool == (int)k);
                    else if (r.d.kind == SD_OP_GROUPNORM && r.d.src0 != b) ok = !getenv("SD_GN_SKIP_NOT_DEFERRED");      // src1 of a GroupNorm only lends its extents (crop region)
                    else ok = false;                                                     // another GroupNorm etc.
                }
                if (ok && any) { g.gn_defer = true; m->buf_gn[b] = (int)i; }
            }
        }
        for (int b = 1; b < m->nbuf; ++b)
            if (m->buf_gn[b] >= 0) {
                m->gn_tab_off[b] = m->ws_base;
                m->ws_base += rup_sz((size_t)2 * m->bufCp[b] * 4, 256);
            }
        for (size_t i = 0; i + 1 < m->ops.size(); ++i) {
            Op& c = m->ops[i];
            Op& nx = m->ops[i + 1];
            if (c.d.kind != SD_OP_CONV || c.first) continue;
            // GroupNorm right behind a conv over the conv's whole output: statistics in the conv epilogue
            if (nx.d.kind == SD_OP_GROUPNORM && nx.d.src0 == c.d.dst && nx.d.src1 < 0 && !getenv("SD_NO_GN_FUSE")) {
                c.fuse_gn = (int)(i + 1);
                nx.stats_done = true;
                // ... and if that GroupNorm is deferred and feeds a pooling, this conv also pools its RAW output (per channel the
                // window's maximum or, for gamma < 0, minimum: relu(a x + b) is monotone in x) and the readers of the pooled tensor
                // apply the scale / shift table like the readers of the full-resolution one: the apply + pool pass disappears.
                // Needs the deferred-input kernel form (MODE 2 carries the epilogue) and deferral-capable readers.
                if (nx.gn_defer && nx.gn_pool == (int)i + 2 && !getenv("SD_NO_GN_POOL_RAW")) {
                    const Op& po = m->ops[i + 2];
                    const int pb = po.d.dst;
                    bool ok = (c.d.kz == 3) == (po.d.kz == 2) && m->bufCp[pb] <= 256 &&
                              (m->buf_gn[c.d.src0] >= 0 || (c.d.src1 >= 0 && m->buf_gn[c.d.src1] >= 0));
                    for (size_t k = i + 3; k < m->ops.size() && ok; ++k) {
                        const Op& r = m->ops[k];
                        if (r.d.src0 != pb && r.d.src1 != pb) continue;
                        if (r.d.kind == SD_OP_CONV) ok = !r.first;
                        else if (r.d.kind == SD_OP_UPCONV) ok = upconv_rows_kernel(m->bufCp[pb] / SD_CHUNK, rup(r.d.cout, SD_CHUNK));
                        else ok = r.d.kind == SD_OP_FINAL || (r.d.kind == SD_OP_GROUPNORM && r.d.src0 != pb);
                    }
                    if (ok) {
                        c.fuse_pool_raw = (int)i + 2;
                        nx.pool_in_conv = true;
                        m->buf_gn[pb] = (int)i + 1;
                        m->gn_tab_off[pb] = m->gn_tab_off[nx.d.src0];      // one table for the tensor and its pooled version
                        const int ntile = c.NT * c.NB;
                        c.pooldir_off = blob_alloc((size_t)ntile * 2 * 8 * 4);
                        uint32_t* dq = reinterpret_cast<uint32_t*>(blob.data() + c.pooldir_off);
                        // register r = 4 s + t of lane half h holds the channels 32 tile + 16 s + 8 (t >> 1) + 4 h + 2 (t & 1) + {0, 1}
                        for (int tile = 0; tile < ntile; ++tile)
                            for (int h = 0; h < 2; ++h)
                                for (int r = 0; r < 8; ++r) {
                                    uint32_t w = 0;
                                    for (int e = 0; e < 2; ++e) {
                                        const int ch = 32 * tile + 16 * (r >> 2) + 8 * ((r & 3) >> 1) + 4 * h + 2 * (r & 1) + e;
                                        if (ch < m->bufC[nx.d.src0] && W[nx.d.gamma_off + ch] < 0.f) w |= 0xffffu << (16 * e);
                                    }
                                    dq[(tile * 2 + h) * 8 + r] = w;
                                }
                    }
                }
                continue;
            }
            // (the fused pooling maximum works on the packed, rounded outputs as integers: exact behind a ReLU only)
            if (nx.d.kind == SD_OP_POOL && nx.d.src0 == c.d.dst && (c.d.kz == 3) == (nx.d.kz == 2) && c.d.relu) {
                c.fuse_pool = (int)(i + 1);
                nx.skipped = true;
            } else if (nx.d.kind == SD_OP_FINAL && nx.d.src0 == c.d.dst && c.NB == 1 && i + 2 == m->ops.size()) {
                c.fuse_final = (int)(i + 1);
                nx.skipped = true;
                // fp32 final weights as hi + lo parts in the activation dtype, in MFMA A-fragment order: row = class
                // (lane & 31), k-step s / element jj <-> channel j*32 + 8*(2s + (jj>>2)) + 4*(lane>>5) + (jj&3)
                const sd_op_desc& fd = nx.d;
                c.fwfrag_off = blob_alloc((size_t)c.NT * 2 * 2 * 64 * 8 * 2);
                uint16_t* fp = reinterpret_cast<uint16_t*>(blob.data() + c.fwfrag_off);
                auto back = [&](uint16_t bits) -> float {
                    if (act_dtype == SD_BF16) { uint32_t u = (uint32_t)bits << 16; float f; std::memcpy(&f, &u, 4); return f; }
                    _Float16 h; std::memcpy(&h, &bits, 2); return (float)h;
                };
                for (int j = 0; j < c.NT; ++j)
                    for (int s2 = 0; s2 < 2; ++s2)
                        for (int l = 0; l < 64; ++l)
                            for (int jj = 0; jj < 8; ++jj) {
                                const int co = l & 31, ch = j * 32 + 8 * (2 * s2 + (jj >> 2)) + 4 * (l >> 5) + (jj & 3);
                                const float wv = (co < fd.cout && ch < fd.cin0) ? W[fd.w_off + (size_t)co * fd.cin0 + ch] : 0.f;
                                const uint16_t hi = cvt(wv, act_dtype);
                                const uint16_t lo = cvt(wv - back(hi), act_dtype);
                                fp[((size_t)((j * 2 + s2) * 2 + 0) * 64 + l) * 8 + jj] = hi;
                                fp[((size_t)((j * 2 + s2) * 2 + 1) * 64 + l) * 8 + jj] = lo;
                            }
            }
        }
        // Level-0 decoder as one streaming launch: planar up-convolution 64 -> 32, merge conv (32 + 32 -> 32), conv 32 -> 32
        // with the fused final layer, all with folded BatchNorm + ReLU (no GroupNorm between them) and no other reader of
        // the two intermediate tensors.
        if (!m->keep_all && !getenv("SD_NO_DEC0")) {
            for (size_t i = 0; i + 3 < m->ops.size(); ++i) {
                Op& u = m->ops[i];
                Op& c1 = m->ops[i + 1];
                Op& c2 = m->ops[i + 2];
                if (u.d.kind != SD_OP_UPCONV || u.d.kz != 1 || !u.d.relu || m->bufCp[u.d.src0] != 64 || m->bufCp[u.d.dst] != 32) continue;
                if (c1.d.kind != SD_OP_CONV || c1.first || c1.d.kz != 1 || !c1.d.relu || c1.d.src0 != u.d.dst || c1.d.src1 < 0 ||
                    m->bufCp[c1.d.src1] != 32 || m->bufCp[c1.d.dst] != 32 || c1.NT != 1 || c1.NB != 1 || c1.fuse_gn >= 0 ||
                    c1.fuse_pool >= 0 || c1.fuse_final >= 0)
                    continue;
                if (c2.d.kind != SD_OP_CONV || c2.d.kz != 1 || !c2.d.relu || c2.d.src0 != c1.d.dst || c2.d.src1 >= 0 ||
                    m->bufCp[c2.d.dst] != 32 || c2.NT != 1 || c2.NB != 1 || c2.fuse_final != (int)i + 3)
                    continue;
                if (m->buf_gn[u.d.src0] >= 0 || m->buf_gn[c1.d.src1] >= 0) continue;
                bool other_reader = false;
                for (size_t k = 0; k < m->ops.size(); ++k) {
                    const sd_op_desc& r = m->ops[k].d;
                    if (k != i + 1 && (r.src0 == u.d.dst || r.src1 == u.d.dst)) other_reader = true;
                    if (k != i + 2 && (r.src0 == c1.d.dst || r.src1 == c1.d.dst)) other_reader = true;
                }
                if (other_reader) continue;
                u.dec0_c1 = (int)i + 1; u.dec0_c2 = (int)i + 2;
                c1.in_dec0 = c2.in_dec0 = true;      // (skipped per launch: dec0_shape_ok)
            }
        }
    }

    {
        m->blob_bytes = rup_sz(blob.size(), 256);
        hipError_t e = hipMalloc((void**)&m->dev_blob, m->blob_bytes + 1024 + 512);
        if (e == hipSuccess) e = hipMemset(m->dev_blob + m->blob_bytes + 1024, 0, 512);
        m->dev_zero = m->dev_blob + m->blob_bytes + 1024;
        m->dev_ovf = reinterpret_cast<int*>(m->dev_blob + m->blob_bytes + 1024 + 256);
        if (e == hipSuccess) e = hipMemcpy(m->dev_blob, blob.data(), blob.size(), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            err = std::string("weight upload: ") + hipGetErrorString(e);
            rc = e == hipErrorOutOfMemory ? SD_ERR_NOMEM : SD_ERR_HIP;
            goto done;
        }
    }
done:
#undef MODEL_FAIL
    if (rc != SD_OK) {
        sd_model_destroy(m);
        return fail(rc, err);
    }
    *out = m;
    return SD_OK;
}

void sd_model_destroy(sd_model* m) {
    if (!m) return;
    for (hipEvent_t e : m->events) (void)hipEventDestroy(e);
    if (m->dev_blob) (void)hipFree(m->dev_blob);
    if (m->dev_clk) (void)hipFree(m->dev_clk);
    f32_model_destroy(m->f32);
    delete m;
}

size_t sd_workspace_bytes(const sd_model* m, int D, int H, int W) {
    if (!m || D <= 0 || H <= 0 || W <= 0) { fail(SD_ERR_INVALID, "bad tile shape"); return 0; }
    if (m->f32) return f32_workspace_bytes(m->f32, D, H, W);
    std::vector<Dims> dims;
    std::vector<size_t> off;
    if (infer_shapes(m, D, H, W, dims) != SD_OK) return 0;
    return (plan_workspace(m, dims, off) + 255) & ~(size_t)255;
}

int sd_model_set_roi(sd_model* m, const int32_t* lo_zyx, const int32_t* hi_zyx) {
    if (!m || ((lo_zyx == nullptr) != (hi_zyx == nullptr))) return fail(SD_ERR_INVALID, "sd_model_set_roi: bad argument");
    m->roi_set = lo_zyx != nullptr;
    for (int a = 0; a < 3; ++a) { m->roi_lo[a] = lo_zyx ? lo_zyx[a] : 0; m->roi_hi[a] = hi_zyx ? hi_zyx[a] : 0; }
    if (m->roi_set)
        for (int a = 0; a < 3; ++a)
            if (m->roi_lo[a] < 0 || m->roi_hi[a] <= m->roi_lo[a]) { m->roi_set = false; return fail(SD_ERR_INVALID, "sd_model_set_roi: empty box"); }
    return SD_OK;
}

int sd_memcpy2d_async(void* dst, size_t dst_pitch, const void* src, size_t src_pitch, size_t width_bytes, size_t height, int kind,
                      void* stream) {
    if (!dst || !src || width_bytes > dst_pitch || width_bytes > src_pitch || kind < 0 || kind > 2)
        return fail(SD_ERR_INVALID, "sd_memcpy2d_async: bad argument");
    if (width_bytes == 0 || height == 0) return SD_OK;
    const hipMemcpyKind k = kind == 0 ? hipMemcpyHostToDevice : kind == 1 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    HIP_TRY(hipMemcpy2DAsync(dst, dst_pitch, src, src_pitch, width_bytes, height, k, reinterpret_cast<hipStream_t>(stream)));
    return SD_OK;
}

int sd_profile_enable(sd_model* m, int n_slots) {
    if (!m || n_slots < 0) return fail(SD_ERR_INVALID, "sd_profile_enable: bad argument");
    for (hipEvent_t e : m->events) (void)hipEventDestroy(e);
    m->events.clear();
    m->profile_slots = n_slots;
    m->n_forward = 0;
    m->ev_recorded.assign((size_t)std::max(n_slots, 0) * (m->ops.size() + 1), 0);
    if (m->dev_clk) { (void)hipFree(m->dev_clk); m->dev_clk = nullptr; }
    if (n_slots > 0) {
        m->events.resize((size_t)n_slots * (m->ops.size() + 1));
        for (auto& e : m->events) HIP_TRY(hipEventCreate(&e));
        const size_t nb = (size_t)n_slots * m->ops.size() * 4 * sizeof(unsigned long long);
        HIP_TRY(hipMalloc(&m->dev_clk, nb));
        HIP_TRY(hipMemset(m->dev_clk, 0, nb));
    }
    return SD_OK;
}

int sd_profile_read_clocks(sd_model* m, int slot, uint64_t* stamps, int n_ops) {
    if (!m || !stamps || m->profile_slots <= 0 || slot < 0 || slot >= m->profile_slots || !m->dev_clk)
        return fail(SD_ERR_INVALID, "sd_profile_read_clocks: profiling not enabled or bad slot");
    const int n = std::min<int>(n_ops, (int)m->ops.size());
    HIP_TRY(hipMemcpy(stamps, m->dev_clk + (size_t)slot * m->ops.size() * 4, (size_t)n * 4 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return SD_OK;
}

int sd_profile_read(sd_model* m, int slot, float* ms, int n_ops) {
    if (!m || !ms || m->profile_slots <= 0 || slot < 0 || slot >= m->profile_slots)
        return fail(SD_ERR_INVALID, "sd_profile_read: profiling not enabled or bad slot");
    const int n = std::min<int>(n_ops, (int)m->ops.size()), nall = (int)m->ops.size();
    const hipEvent_t* ev = m->events.data() + (size_t)slot * (m->ops.size() + 1);
    const char* rec = m->ev_recorded.data() + (size_t)slot * (m->ops.size() + 1);
    for (int i = 0; i < n; ++i) {
        ms[i] = 0.f;
        if (!rec[i]) continue;                       // ran inside another op's launch
        int j = i + 1;
        while (j < nall && !rec[j]) ++j;             // next op that launched (or the end-of-forward event)
        if (!rec[j]) return fail(SD_ERR_INVALID, "sd_profile_read: slot holds no complete forward");
        HIP_TRY(hipEventElapsedTime(&ms[i], ev[i], ev[j]));
    }
    return SD_OK;
}

int sd_model_overflow(sd_model* m, void* stream, int* flag_out) {
    if (!m || !flag_out) return fail(SD_ERR_INVALID, "null argument");
    *flag_out = 0;
    if (m->f32 || (m->act_dtype != SD_F16 && m->act_dtype != SD_F16X2) || !m->dev_ovf) return SD_OK;       // only fp16 storage can overflow
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    HIP_TRY(hipMemcpyAsync(flag_out, m->dev_ovf, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (*flag_out) HIP_TRY(hipMemsetAsync(m->dev_ovf, 0, sizeof(int), s));
    return SD_OK;
}

int sd_forward(sd_model* m, const void* in_dev, int in_dtype, int D, int H, int W, void* out_dev, int out_kind,
               void* ws, size_t ws_bytes, void* stream) {
    return sd_forward_batch(m, in_dev, in_dtype, 1, D, H, W, out_dev, out_kind, ws, ws_bytes, stream);
}

static bool make_label_args(int C, const int32_t* ids, const double* thresholds, int n_ids, LabelArgs& a) {
    if (!ids || !thresholds || n_ids <= 0 || n_ids > 16) return false;
    a.n = n_ids;
    for (int i = 0; i < n_ids; ++i) {
        if (ids[i] < 0 || ids[i] >= C) return false;
        a.ids[i] = ids[i];
        const double t = thresholds[i];
        if (t != t) return false;
        // (uint8 p > t) <=> p >= floor(t) + 1, exact for any real t
        const double c = std::floor(t) + 1.0;
        a.cuts[i] = c < 0.0 ? 0 : (c > 256.0 ? 256 : (int)c);
    }
    return true;
}

static int forward_impl(sd_model* m, const void* in_dev, int in_dtype, int N, int D, int H, int W, void* out_dev,
                        int out_kind, const LabelArgs* lab, void* ws, size_t ws_bytes, void* stream);

int sd_forward_batch(sd_model* m, const void* in_dev, int in_dtype, int N, int D, int H, int W, void* out_dev,
                     int out_kind, void* ws, size_t ws_bytes, void* stream) {
    if (out_kind < 0 || out_kind > 2) return fail(SD_ERR_INVALID, "bad out_kind");
    return forward_impl(m, in_dev, in_dtype, N, D, H, W, out_dev, out_kind, nullptr, ws, ws_bytes, stream);
}

int sd_forward_labels_batch(sd_model* m, const void* in_dev, int in_dtype, int N, int D, int H, int W,
                            const int32_t* ids, const double* thresholds, int n_ids, uint8_t* out_dev, void* ws,
                            size_t ws_bytes, void* stream) {
    if (!m) return fail(SD_ERR_INVALID, "null argument");
    LabelArgs a{};
    if (!make_label_args(m->final_cout, ids, thresholds, n_ids, a))
        return fail(SD_ERR_INVALID, "sd_forward_labels_batch: bad label arguments");
    return forward_impl(m, in_dev, in_dtype, N, D, H, W, out_dev, SD_OUT_LABELS_U8, &a, ws, ws_bytes, stream);
}

static int forward_impl(sd_model* m, const void* in_dev, int in_dtype, int N, int D, int H, int W, void* out_dev,
                        int out_kind, const LabelArgs* lab, void* ws, size_t ws_bytes, void* stream) {
    if (!m || !in_dev || !out_dev || !ws) return fail(SD_ERR_INVALID, "null argument");
    if (N <= 0 || N > 65535) return fail(SD_ERR_INVALID, "bad batch size");
    if (in_dtype != SD_U8 && in_dtype != SD_F32) return fail(SD_ERR_INVALID, "in_dtype must be SD_U8 or SD_F32");
    if (D <= 0 || H <= 0 || W <= 0) return fail(SD_ERR_INVALID, "bad tile shape");
    if ((long)D * H * W >= (1l << 31)) return fail(SD_ERR_INVALID, "tile has 2^31 or more voxels");
    if (m->f32) {
        hipEvent_t* ev32 = nullptr;
        if (m->profile_slots > 0) {
            ev32 = m->events.data() + (size_t)(m->n_forward % m->profile_slots) * (m->ops.size() + 1);
            char* rec = m->ev_recorded.data() + (size_t)(m->n_forward % m->profile_slots) * (m->ops.size() + 1);
            std::fill(rec, rec + m->ops.size() + 1, 1);      // (the fp32 plan launches every op)
        }
        ++m->n_forward;
        m->last_launches = (int)m->ops.size();
        return f32_forward(m->f32, in_dev, in_dtype, N, D, H, W, out_dev, out_kind, lab, ws, ws_bytes,
                           reinterpret_cast<hipStream_t>(stream), ev32);
    }
    int rc = infer_shapes(m, D, H, W, m->dims);
    if (rc != SD_OK) return rc;
    const size_t need = plan_workspace(m, m->dims, m->buf_off);
    // tile t of a batch uses [t*tstride, t*tstride + need) of the workspace, its input / output follow tile 0's
    const size_t tstride = (need + 255) & ~(size_t)255;
    if ((size_t)N * tstride > ws_bytes) return fail(SD_ERR_NOMEM, "workspace too small");
    const size_t in_tstride = (size_t)D * H * W * (in_dtype == SD_U8 ? 1 : 4);
    const size_t out_tstride = out_kind == SD_OUT_LABELS_U8 ? (size_t)D * H * W
                                                            : (size_t)m->final_cout * D * H * W * (out_kind == SD_OUT_PROBS_U8 ? 1 : 4);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    char* const wsb = reinterpret_cast<char*>(ws);
    const bool dec0 = use_dec0(m, m->dims[0]);
    auto bufp = [&](int b) -> void* { return wsb + m->buf_off[b]; };
    // ---- output box of interest (sd_model_set_roi): which sub-box every decoder op computes ------------------------------------
    // need[b] = the part of buffer b that must hold CORRECT values; an op that can work on a sub-box computes view = need[dst]
    // widened by its kernel radius (clipped to the tensor): the kernel zero-pads at the view's border, so the outermost shell of
    // the view is wrong where that border is not the tensor's -- it lies outside need[dst] by construction, and everything it
    // reads is real data (need[src] = view), so no uninitialised value is ever touched.  Ops that cannot (first conv, fused
    // pooling / statistics, separate pooling / final / GroupNorm launches) compute everything and need everything.
    const size_t nops = m->ops.size();
    std::vector<Box> view(nops, Box{{0, 0, 0}, {0, 0, 0}, false});          // any == false: the whole tensor
    bool has_gn = false;
    for (const Op& op : m->ops) has_gn |= op.d.kind == SD_OP_GROUPNORM;
    if (m->roi_set)
        for (int a = 0; a < 3; ++a) {
            const int n = a == 0 ? D : a == 1 ? H : W;
            if (m->roi_lo[a] < 0 || m->roi_hi[a] > n || m->roi_lo[a] >= m->roi_hi[a]) return fail(SD_ERR_INVALID, "sd_model_set_roi: box outside the tile");
        }
    if (m->roi_set && !has_gn && !m->keep_all) {
        std::vector<Box> need(m->nbuf, Box{{0, 0, 0}, {0, 0, 0}, false});   // any == false: nothing needed yet
        auto dims3 = [&](int b, int* n) { n[0] = m->dims[b].d; n[1] = m->dims[b].h; n[2] = m->dims[b].w; };
        auto add = [&](int b, const Box& x) {
            if (b <= 0) return;
            Box& t = need[b];
            for (int a = 0; a < 3; ++a) {
                t.lo[a] = t.any ? std::min(t.lo[a], x.lo[a]) : x.lo[a];
                t.hi[a] = t.any ? std::max(t.hi[a], x.hi[a]) : x.hi[a];
            }
            t.any = true;
        };
        auto add_full = [&](int b) { if (b > 0) { Box f; int n[3]; dims3(b, n); for (int a = 0; a < 3; ++a) { f.lo[a] = 0; f.hi[a] = n[a]; } f.any = true; add(b, f); } };
        for (size_t k = nops; k-- > 0;) {
            const Op& op = m->ops[k];
            const sd_op_desc& d = op.d;
            if (op.skipped || (op.in_dec0 && dec0)) continue;
            if (d.kind == SD_OP_UPCONV && op.dec0_c1 >= 0 && dec0) {      // the fused level-0 decoder on its sub-box
                const Box v = dec0_view(m, m->dims[m->ops[op.dec0_c1].d.src1]);
                if (v.any) {
                    view[k] = v;
                    add(m->ops[op.dec0_c1].d.src1, v);
                    Box l; l.any = true;
                    for (int a = 0; a < 3; ++a) { l.lo[a] = a == 0 ? v.lo[a] : v.lo[a] / 2; l.hi[a] = a == 0 ? v.hi[a] : (v.hi[a] + 1) / 2; }
                    add(d.src0, l);
                } else { add_full(d.src0); add_full(m->ops[op.dec0_c1].d.src1); }
                continue;
            }
            if (d.kind == SD_OP_CONV && !op.first) {
                Box c{{0, 0, 0}, {0, 0, 0}, false};
                if (op.fuse_final >= 0) { for (int a = 0; a < 3; ++a) { c.lo[a] = m->roi_lo[a]; c.hi[a] = m->roi_hi[a]; } c.any = true; }
                else c = need[d.dst];
                int n[3];
                dims3(d.dst, n);
                // fused MaxPool (1,2,2) / (2,2,2) behind a planar / 3x3x3 convolution: what the pooled tensor is needed for, in
                // this convolution's coordinates
                const int kzp = d.kz == 3 ? 2 : 1;
                if (op.fuse_pool >= 0 && need[m->ops[op.fuse_pool].d.dst].any) {
                    const Box& q = need[m->ops[op.fuse_pool].d.dst];
                    const int f[3] = {kzp, 2, 2};
                    Box up; up.any = true;
                    for (int a = 0; a < 3; ++a) { up.lo[a] = q.lo[a] * f[a]; up.hi[a] = std::min(n[a], q.hi[a] * f[a]); }
                    if (c.any) for (int a = 0; a < 3; ++a) { c.lo[a] = std::min(c.lo[a], up.lo[a]); c.hi[a] = std::max(c.hi[a], up.hi[a]); }
                    else c = up;
                }
                // in y / x only plain convolutions work on a sub-box; along z also those with fused pooling or the first
                // convolution inside (a z-range is a shift of every base pointer: planes are never addressed through the extent)
                static const bool no_zenc = getenv("SD_ROI_NO_ENCODER") != nullptr;
                const bool yx = op.fuse_pool < 0 && op.fuse_pool_raw < 0 && op.fuse_gn < 0 && op.fuse_first < 0;
                const bool zok = op.fuse_pool_raw < 0 && op.fuse_gn < 0 && (yx || !no_zenc);
                if (c.any && zok) {
                    const int r[3] = {d.kz / 2, 1, 1};
                    Box v; v.any = false;
                    for (int a = 0; a < 3; ++a) {
                        v.lo[a] = (a == 0 || yx) ? std::max(0, c.lo[a] - r[a]) : 0;
                        v.hi[a] = (a == 0 || yx) ? std::min(n[a], c.hi[a] + r[a]) : n[a];
                    }
                    if (op.fuse_pool >= 0 && kzp == 2) {      // whole pooling windows: an even start, an even end (or the tensor's own)
                        v.lo[0] &= ~1;
                        v.hi[0] = std::min(n[0], (v.hi[0] + 1) & ~1);
                    }
                    for (int a = 0; a < 3; ++a) v.any |= v.lo[a] > 0 || v.hi[a] < n[a];
                    if (v.any) { view[k] = v; add(d.src0, v); add(d.src1, v); continue; }
                }
                add_full(d.src0); add_full(d.src1);
            } else if (d.kind == SD_OP_UPCONV && need[d.dst].any) {
                const Box& c = need[d.dst];
                int n[3];
                dims3(d.src0, n);
                const int f[3] = {d.kz, 2, 2};
                Box v; v.any = false;
                for (int a = 0; a < 3; ++a) {
                    v.lo[a] = std::max(0, c.lo[a] / f[a]);
                    v.hi[a] = std::min(n[a], (c.hi[a] + f[a] - 1) / f[a]);
                    v.any |= v.lo[a] > 0 || v.hi[a] < n[a];
                }
                if (v.any) { view[k] = v; add(d.src0, v); } else add_full(d.src0);
            } else {
                add_full(d.src0); add_full(d.src1);
            }
        }
    }
    hipEvent_t* ev = nullptr;
    char* ev_rec = nullptr;
    if (m->profile_slots > 0) {
        ev = m->events.data() + (size_t)(m->n_forward % m->profile_slots) * (m->ops.size() + 1);
        ev_rec = m->ev_recorded.data() + (size_t)(m->n_forward % m->profile_slots) * (m->ops.size() + 1);
        std::fill(ev_rec, ev_rec + m->ops.size() + 1, 0);
    }
    ++m->n_forward;
    m->last_launches = 0;
    m->op_kernel.assign(m->ops.size(), nullptr);
    m->op_exec.assign(m->ops.size(), -1);
    const bool no_first_u8 = getenv("SD_NO_FIRST_U8") != nullptr;      // A/B switch (read per forward): uint8 input through the exact-f32 MFMA chain
    {
        // GroupNorm statistics scratch (sum and sum of squares per channel, doubles, at the start of every tile's workspace): zeroed
        // ONCE per forward pass; every k_gn_finalize leaves it zeroed for the next GroupNorm op
        int maxc = 0;
        for (const Op& op : m->ops)
            if (op.d.kind == SD_OP_GROUPNORM) maxc = std::max(maxc, m->bufCp[op.d.src0]);
        if ((size_t)2 * maxc * sizeof(double) > SD_GN_SCALE_OFF || SD_GN_SCALE_OFF + (size_t)2 * maxc * sizeof(float) > 65536)
            return fail(SD_ERR_INVALID, "GroupNorm: more channels than the statistics scratch holds");
        if (maxc > 0) {
            const int rc0 = launch_zero_scratch(wsb, tstride, 2 * maxc * (int)sizeof(double), N, s);
            if (rc0 != SD_OK) return fail(rc0, "zeroing the GroupNorm scratch failed");
        }
    }

    // voxels of all tiles that op k's launch computes (its sub-box, if it has one): what the launchers choose the kernel form by
    auto launch_vox = [&](size_t k) -> long {
        const Dims o = m->dims[m->ops[k].d.dst];
        const Box& v = view[k];
        return (v.any ? (long)(v.hi[0] - v.lo[0]) * (v.hi[1] - v.lo[1]) * (v.hi[2] - v.lo[2]) : (long)o.d * o.h * o.w) * N;
    };
    for (size_t i = 0; i < m->ops.size(); ++i) {
        const Op& op = m->ops[i];
        const sd_op_desc& d = op.d;
        if (op.skipped || (op.in_dec0 && dec0)) continue;
        if (op.first && d.kind == SD_OP_CONV && i + 1 < m->ops.size() && m->ops[i + 1].fuse_first == (int)i && !m->keep_all) {
            // (a first convolution that will run inside its consumer launches nothing: no event either; decided below per launch)
            const Op& c = m->ops[i + 1];
            const int nst1 = (m->bufCp[c.d.src0] / SD_CHUNK) * c.d.kz * (m->split ? 3 : 1);
            const bool fused1 = m->split
                ? conv_can_fuse_first_split(c.d.kz, c.NT, c.NB, launch_vox(i + 1), nst1, c.fuse_final >= 0)
                : conv_can_fuse_first(c.d.kz, c.NT, c.NB, launch_vox(i + 1), nst1, c.fuse_final >= 0);
            if (fused1) { ++m->last_launches; m->op_exec[i] = (int)i + 1; continue; }
        }
        if (ev) { HIP_TRY(hipEventRecord(ev[i], s)); ev_rec[i] = 1; }
        ++m->last_launches;
        sd_tls_kernel = nullptr;
        switch (d.kind) {
        case SD_OP_CONV: {
            const Dims o = m->dims[d.dst];
            const int BZ = sd_bz(d.kz), BY = sd_by(d.kz);
            bool first_fused_into_next = false;
            if (op.first && i + 1 < m->ops.size() && m->ops[i + 1].fuse_first == (int)i && !m->keep_all) {
                const Op& c = m->ops[i + 1];
                const int nst = (m->bufCp[c.d.src0] / SD_CHUNK) * c.d.kz * (m->split ? 3 : 1);
                first_fused_into_next = m->split
                    ? conv_can_fuse_first_split(c.d.kz, c.NT, c.NB, launch_vox(i + 1), nst, c.fuse_final >= 0)
                    : conv_can_fuse_first(c.d.kz, c.NT, c.NB, launch_vox(i + 1), nst, c.fuse_final >= 0);
            }
            if (first_fused_into_next) {
                // computed inside the next convolution
            } else if (op.first) {
                FirstParams p{};
                p.in = in_dev; p.D = o.d; p.H = o.h; p.W = o.w;
                p.dst = bufp(d.dst); p.Cd = m->bufCp[d.dst];
                p.wpack = reinterpret_cast<const float*>(m->dev_blob + op.wpack_off);
                p.wpack3 = (op.w3_off && !no_first_u8) ? m->dev_blob + op.w3_off : nullptr;
                p.bias = reinterpret_cast<const float*>(m->dev_blob + op.bias_off);
                p.relu = d.relu;
                p.batch = N; p.tstride = tstride; p.in_tstride = in_tstride; p.ovf = m->dev_ovf;
                p.nbx = (o.w + SD_BX - 1) / SD_BX; p.nby = (o.h + BY - 1) / BY; p.nbz = (o.d + BZ - 1) / BZ;
                rc = launch_first(p, m->act_dtype, in_dtype, d.kz, s);
            } else {
                ConvParams p{};
                const Dims a = m->dims[d.src0];
                p.src0 = bufp(d.src0); p.C0 = m->bufCp[d.src0]; p.H0 = a.h; p.W0 = a.w; p.P0 = (size_t)a.d * a.h * a.w;
                const int vmul = m->split ? 3 : 1;      // split-fp16 plan: virtual chunks [hi | hi | lo]
                p.nchunk0 = p.C0 / SD_CHUNK * vmul;
                p.oscale = op.oscale;
                if (d.src1 >= 0) {
                    const Dims b = m->dims[d.src1];
                    p.src1 = bufp(d.src1); p.C1 = m->bufCp[d.src1]; p.H1 = b.h; p.W1 = b.w; p.P1 = (size_t)b.d * b.h * b.w;
                    p.nchunk1 = p.C1 / SD_CHUNK * vmul;
                }
                p.dst = bufp(d.dst); p.Cd = m->bufCp[d.dst];
                p.D = o.d; p.H = o.h; p.W = o.w; p.Pd = (size_t)o.d * o.h * o.w;
                p.Hd = o.h; p.Wd = o.w; p.final_nvox = (long)o.d * o.h * o.w;
                p.wpack = m->dev_blob + op.wpack_off;
                p.bias = reinterpret_cast<const float*>(m->dev_blob + op.bias_off);
                p.relu = d.relu; p.zero = m->dev_zero;
                p.store_main = 1; p.ovf = m->dev_ovf;
                p.batch = N; p.tstride = tstride; p.out_tstride = out_tstride;
                if (ev && m->dev_clk) p.clk = m->dev_clk + ((size_t)((m->n_forward - 1) % m->profile_slots) * m->ops.size() + i) * 4;
#ifdef SD_TIMING
                p.dbg = (getenv("SD_TIMING_OP") && atoi(getenv("SD_TIMING_OP")) == (int)i) ? reinterpret_cast<long long*>(wsb + m->buf_off[1]) : nullptr;
#endif
                if (op.fuse_pool >= 0) {
                    const sd_op_desc& pd = m->ops[op.fuse_pool].d;
                    p.pool_dst = bufp(pd.dst); p.pH = m->dims[pd.dst].h; p.pW = m->dims[pd.dst].w;
                    p.Pp = (size_t)m->dims[pd.dst].d * p.pH * p.pW;
                }
                if (op.fuse_final >= 0) {
                    const Op& fo = m->ops[op.fuse_final];
                    if (o.d != D || o.h != H || o.w != W) return fail(SD_ERR_INVALID, "final layer shape != input shape");
                    p.final_wfrag = m->dev_blob + op.fwfrag_off;
                    p.final_b = reinterpret_cast<const float*>(m->dev_blob + fo.bias_off);
                    p.final_cout = fo.d.cout; p.final_kind = out_kind; p.final_out = out_dev;
                    p.final_oscale = op.final_oscale;
                    p.ovf = m->dev_ovf;
                    if (lab) p.lab = *lab;
                    p.store_main = m->keep_all ? 1 : 0;
                }
                if (op.fuse_first >= 0 && !m->keep_all &&
                    (m->split ? conv_can_fuse_first_split(d.kz, op.NT, op.NB, launch_vox(i), (p.nchunk0 + p.nchunk1) * d.kz,
                                                          op.fuse_final >= 0)
                              : conv_can_fuse_first(d.kz, op.NT, op.NB, launch_vox(i), (p.nchunk0 + p.nchunk1) * d.kz,
                                                    op.fuse_final >= 0))) {
                    const Op& fo = m->ops[op.fuse_first];
                    p.first_in = in_dev; p.first_in_tstride = in_tstride; p.first_in_f32 = in_dtype == SD_F32 ? 1 : 0;
                    p.first_w = reinterpret_cast<const float*>(m->dev_blob + fo.wpack_off);
                    p.first_bias = reinterpret_cast<const float*>(m->dev_blob + fo.bias_off);
                    p.first_w3 = (fo.w3_off && in_dtype == SD_U8 && !no_first_u8) ? m->dev_blob + fo.w3_off : nullptr;
                    p.first_relu = fo.d.relu;
                }
                if (op.fuse_gn >= 0) {      // (scratch is zero: see the start of the forward pass and k_gn_finalize)
                    p.gn_sums = reinterpret_cast<double*>(wsb); p.gn_C = p.Cd;
                }
                if (op.fuse_pool_raw >= 0) {
                    const sd_op_desc& pd = m->ops[op.fuse_pool_raw].d;
                    p.pool_dst = bufp(pd.dst); p.pH = m->dims[pd.dst].h; p.pW = m->dims[pd.dst].w;
                    p.Pp = (size_t)m->dims[pd.dst].d * p.pH * p.pW;
                    p.pool_dir = reinterpret_cast<const unsigned*>(m->dev_blob + op.pooldir_off);
                }
                p.nbx = (o.w + SD_BX - 1) / SD_BX; p.nby = (o.h + BY - 1) / BY; p.nbz = (o.d + BZ - 1) / BZ;
                if (view[i].any) {      // sub-box launch: same strides, shifted bases, the box as the extent
                    const Box& v = view[i];
                    const size_t esz = SD_CHUNK * 2;      // bytes per voxel of a chunk plane (bf16 / fp16 storage)
                    auto vox = [&](int hh, int ww) { return ((size_t)v.lo[0] * hh + v.lo[1]) * ww + v.lo[2]; };
                    p.src0 = reinterpret_cast<const char*>(p.src0) + vox(p.H0, p.W0) * esz;
                    if (p.src1) p.src1 = reinterpret_cast<const char*>(p.src1) + vox(p.H1, p.W1) * esz;
                    p.dst = reinterpret_cast<char*>(p.dst) + vox(o.h, o.w) * esz;
                    if (p.final_out)
                        p.final_out = reinterpret_cast<char*>(p.final_out) + vox(o.h, o.w) * (out_kind == SD_OUT_PROBS_U8 || out_kind == SD_OUT_LABELS_U8 ? 1 : 4);
                    if (p.pool_dst)      // (only z-ranges reach a convolution with fused pooling: lo[1] = lo[2] = 0)
                        p.pool_dst = reinterpret_cast<char*>(p.pool_dst) + (size_t)(v.lo[0] / (d.kz == 3 ? 2 : 1)) * p.pH * p.pW * esz;
                    if (p.first_in)
                        p.first_in = reinterpret_cast<const char*>(p.first_in) + (size_t)v.lo[0] * o.h * o.w * (in_dtype == SD_U8 ? 1 : 4);
                    p.D = v.hi[0] - v.lo[0]; p.H = v.hi[1] - v.lo[1]; p.W = v.hi[2] - v.lo[2];
                }
                auto gn_of = [&](int b, const float*& tab, int& relu) {
                    if (b > 0 && m->buf_gn[b] >= 0) {
                        tab = reinterpret_cast<const float*>(wsb + m->gn_tab_off[b]);
                        relu = m->ops[m->buf_gn[b]].d.relu;
                    }
                };
                gn_of(d.src0, p.gn0, p.gn_relu0);
                if (d.src1 >= 0) gn_of(d.src1, p.gn1, p.gn_relu1);
                if (p.gn0 || p.gn1) {
                    // the kernel keeps the scale / shift of all its tiles in LDS (128 B per 16 channels and tile): launch
                    // the tiles in groups that fit beside the halo / weight buffers (24 KiB are always free)
                    const size_t per_tile = conv_gn_lds_per_tile(p.nchunk0 + p.nchunk1);
                    const int group = (int)std::max<size_t>(1, std::min<size_t>((size_t)N, (24 * 1024) / per_tile));
                    for (int t0 = 0; t0 < N && rc == SD_OK; t0 += group) {
                        ConvParams q = p;
                        q.batch = std::min(group, N - t0);
                        q.batch_total = N;      // the kernel form is chosen for the whole launch set, not per tile group
                        auto adv = [&](const void* ptr) { return ptr ? reinterpret_cast<const char*>(ptr) + (size_t)t0 * tstride : nullptr; };
                        q.src0 = adv(p.src0); q.src1 = adv(p.src1);
                        q.dst = const_cast<char*>(reinterpret_cast<const char*>(adv(p.dst)));
                        q.pool_dst = p.pool_dst ? const_cast<char*>(reinterpret_cast<const char*>(adv(p.pool_dst))) : nullptr;
                        q.gn_sums = p.gn_sums ? reinterpret_cast<double*>(const_cast<char*>(reinterpret_cast<const char*>(adv(p.gn_sums)))) : nullptr;
                        q.gn0 = reinterpret_cast<const float*>(adv(p.gn0)); q.gn1 = reinterpret_cast<const float*>(adv(p.gn1));
                        q.final_out = p.final_out ? reinterpret_cast<char*>(p.final_out) + (size_t)t0 * out_tstride : nullptr;
                        rc = launch_conv(q, m->act_dtype, d.kz, op.NT, op.NB, s);
                    }
                } else {
                    rc = launch_conv(p, m->act_dtype, d.kz, op.NT, op.NB, s);
                }
            }
            break;
        }
        case SD_OP_POOL: {
            PoolParams p{};
            const Dims a = m->dims[d.src0], o = m->dims[d.dst];
            p.src = bufp(d.src0); p.dst = bufp(d.dst); p.C = m->bufCp[d.src0];
            p.D = a.d; p.H = a.h; p.W = a.w; p.Do = o.d; p.Ho = o.h; p.Wo = o.w; p.kz = d.kz;
            p.Ps = (size_t)a.d * a.h * a.w; p.Pd = (size_t)o.d * o.h * o.w;
            p.batch = N; p.tstride = tstride;
            rc = launch_pool(p, m->act_dtype, s);
            break;
        }
        case SD_OP_UPCONV: {
            if (op.dec0_c1 >= 0 && dec0) {
                const Op& c1 = m->ops[op.dec0_c1];
                const Op& c2 = m->ops[op.dec0_c2];
                const Op& fo = m->ops[c2.fuse_final];
                const Dims a = m->dims[d.src0], o = m->dims[c1.d.src1];
                if (o.d != D || o.h != H || o.w != W || a.d != D) return fail(SD_ERR_INVALID, "final layer shape != input shape");
                Dec0Params p{};
                p.l1 = bufp(d.src0); p.skip = bufp(c1.d.src1);
                p.D = o.d; p.H = o.h; p.W = o.w; p.H1 = a.h; p.W1 = a.w; p.Ps = (size_t)o.d * o.h * o.w;
                p.wup = m->dev_blob + op.wpack_off; p.bup = reinterpret_cast<const float*>(m->dev_blob + op.bias_off);
                p.w1 = m->dev_blob + c1.wpack_off; p.b1 = reinterpret_cast<const float*>(m->dev_blob + c1.bias_off);
                p.w2 = m->dev_blob + c2.wpack_off; p.b2 = reinterpret_cast<const float*>(m->dev_blob + c2.bias_off);
                p.fw = m->dev_blob + c2.fwfrag_off; p.fb = reinterpret_cast<const float*>(m->dev_blob + fo.bias_off);
                p.final_cout = fo.d.cout; p.final_kind = out_kind; p.final_out = out_dev;
                if (lab) p.lab = *lab;
                p.zero = m->dev_zero; p.ovf = m->dev_ovf;
                p.batch = N; p.tstride = tstride; p.out_tstride = out_tstride;
                if (view[i].any) {      // sub-box launch: the tensors' extents as strides, shifted bases, the box as the extent
                    const Box& v = view[i];
                    const size_t esz = SD_CHUNK * 2;
                    p.sH = o.h; p.sW = o.w; p.sH1 = a.h; p.sW1 = a.w;
                    p.Pl1 = (size_t)a.d * a.h * a.w; p.out_nvox = (long)o.d * o.h * o.w;
                    const size_t vo = ((size_t)v.lo[0] * o.h + v.lo[1]) * o.w + v.lo[2];
                    p.skip = reinterpret_cast<const char*>(p.skip) + vo * esz;
                    p.l1 = reinterpret_cast<const char*>(p.l1) + (((size_t)v.lo[0] * a.h + v.lo[1] / 2) * a.w + v.lo[2] / 2) * esz;
                    p.final_out = reinterpret_cast<char*>(p.final_out) + vo * (out_kind == SD_OUT_PROBS_U8 || out_kind == SD_OUT_LABELS_U8 ? 1 : 4);
                    p.D = v.hi[0] - v.lo[0]; p.H = v.hi[1] - v.lo[1]; p.W = v.hi[2] - v.lo[2];
                    p.H1 = (v.hi[1] + 1) / 2 - v.lo[1] / 2; p.W1 = (v.hi[2] + 1) / 2 - v.lo[2] / 2;
                }
                p.dbg = getenv("SD_DEC0_DBG") ? reinterpret_cast<long long*>(wsb) : nullptr;      // (timing builds) GroupNorm scratch
                rc = launch_dec0(p, m->act_dtype, s);
                break;
            }
            UpconvParams p{};
            const Dims a = m->dims[d.src0];
            p.src = bufp(d.src0); p.Cs = m->bufCp[d.src0]; p.nchunk = p.Cs / SD_CHUNK * (m->split ? 3 : 1);
            p.oscale = op.oscale;
            p.D = a.d; p.H = a.h; p.W = a.w;
            p.Ps = (size_t)a.d * a.h * a.w; p.Hs = a.h; p.Ws = a.w; p.Hd = m->dims[d.dst].h; p.Wd = m->dims[d.dst].w;
            p.dst = bufp(d.dst); p.Cd = m->bufCp[d.dst]; p.kz = d.kz;
            p.Pd = (size_t)m->dims[d.dst].d * m->dims[d.dst].h * m->dims[d.dst].w;
            if (view[i].any) {      // sub-box of the source voxels (and the 2x box of the output they produce)
                const Box& v = view[i];
                const size_t esz = SD_CHUNK * 2;
                p.src = reinterpret_cast<const char*>(p.src) + (((size_t)v.lo[0] * a.h + v.lo[1]) * a.w + v.lo[2]) * esz;
                p.dst = reinterpret_cast<char*>(p.dst) + (((size_t)v.lo[0] * d.kz * p.Hd + 2 * v.lo[1]) * p.Wd + 2 * v.lo[2]) * esz;
                p.D = v.hi[0] - v.lo[0]; p.H = v.hi[1] - v.lo[1]; p.W = v.hi[2] - v.lo[2];
            }
            p.wpack = m->dev_blob + op.wpack_off;
            p.bias = reinterpret_cast<const float*>(m->dev_blob + op.bias_off);
            p.relu = d.relu; p.ntot = d.kz * 4 * p.Cd; p.ovf = m->dev_ovf;
            p.batch = N; p.tstride = tstride;
            if (m->buf_gn[d.src0] >= 0) {
                p.gn = reinterpret_cast<const float*>(wsb + m->gn_tab_off[d.src0]);
                p.gn_relu = m->ops[m->buf_gn[d.src0]].d.relu;
            }
            rc = launch_upconv(p, m->act_dtype, op.NB, s);
            break;
        }
        case SD_OP_GROUPNORM: {
            GnParams p{};
            const Dims a = m->dims[d.src0];
            const Dims r = d.src1 >= 0 ? m->dims[d.src1] : a;
            p.buf = bufp(d.src0); p.C = m->bufCp[d.src0];
            p.D = r.d; p.H = r.h; p.W = r.w; p.Hs = a.h; p.Ws = a.w; p.P = (size_t)a.d * a.h * a.w;
            p.groups = d.groups; p.cout = m->bufC[d.src0]; p.eps = d.eps;
            p.gamma = reinterpret_cast<const float*>(m->dev_blob + op.aux_off);
            p.beta = p.gamma + p.C;
            p.sums = reinterpret_cast<double*>(wsb);
            // (not deferred: a fixed place behind the statistics scratch -- the scratch itself must stay zero between GroupNorm ops)
            p.scale_shift = reinterpret_cast<float*>(op.gn_defer ? wsb + m->gn_tab_off[d.src0] : wsb + SD_GN_SCALE_OFF);
            p.relu = d.relu;
            p.batch = N; p.tstride = tstride; p.skip_stats = op.stats_done ? 1 : 0;
            p.skip_apply = (op.gn_defer && (op.gn_pool < 0 || op.pool_in_conv)) ? 1 : 0;
            p.no_inplace = op.gn_defer ? 1 : 0;
            if (op.gn_pool >= 0 && !op.pool_in_conv) {
                const sd_op_desc& pd = m->ops[op.gn_pool].d;
                const Dims po = m->dims[pd.dst];
                p.pool_dst = bufp(pd.dst); p.pkz = pd.kz; p.pD = po.d; p.pH = po.h; p.pW = po.w;
            }
            rc = launch_groupnorm(p, m->act_dtype, s);
            break;
        }
        case SD_OP_FINAL: {
            FinalParams p{};
            const Dims a = m->dims[d.src0];
            p.src = bufp(d.src0); p.Cs = m->bufCp[d.src0]; p.cin = d.cin0;
            p.w = reinterpret_cast<const float*>(m->dev_blob + op.wpack_off);
            p.bias = reinterpret_cast<const float*>(m->dev_blob + op.bias_off);
            p.wfrag = m->dev_blob + op.aux_off;
            p.cout = d.cout; p.out = out_dev; p.out_kind = out_kind; p.ovf = m->dev_ovf;
            p.nvox = (long)a.d * a.h * a.w;
            p.batch = N; p.tstride = tstride; p.out_tstride = out_tstride;
            if (lab) p.lab = *lab;
            if (m->buf_gn[d.src0] >= 0) {     // deferred GroupNorm apply of the input
                p.gn_scale_shift = reinterpret_cast<const float*>(wsb + m->gn_tab_off[d.src0]);
                p.gn_relu = m->ops[m->buf_gn[d.src0]].d.relu;
            }
            if (a.d != D || a.h != H || a.w != W) return fail(SD_ERR_INVALID, "final layer shape != input shape");
            rc = launch_final(p, m->act_dtype, s);
            break;
        }
        default: rc = SD_ERR_INVALID;
        }
        if (rc != SD_OK) {
            char msg[128];
            snprintf(msg, sizeof msg, "launch of op %zu (kind %d) failed: %s", i, d.kind,
                     hipGetErrorString(hipGetLastError()));
            return fail(rc, msg);
        }
        // what ran: this op by the kernel its launcher noted, and inside that launch the ops fused into it
        m->op_kernel[i] = sd_tls_kernel;
        if (m->op_exec[i] < 0) m->op_exec[i] = (int)i;
        auto own = [&](int j) { if (j >= 0 && j < (int)m->ops.size()) m->op_exec[j] = (int)i; };
        own(op.fuse_pool); own(op.fuse_final); own(op.fuse_pool_raw); own(op.gn_pool);
        if (d.kind == SD_OP_UPCONV && op.dec0_c1 >= 0 && dec0) {
            own(op.dec0_c1); own(op.dec0_c2); own(m->ops[op.dec0_c2].fuse_final);
        }
    }
    if (ev) { HIP_TRY(hipEventRecord(ev[m->ops.size()], s)); ev_rec[m->ops.size()] = 1; }
    return SD_OK;
}

int sd_tile_gather(const void* vol, int dtype, int VD, int VH, int VW, int oz, int oy, int ox, void* tile, int TD, int TH,
                   int TW, void* stream) {
    if (!vol || !tile || (dtype != SD_U8 && dtype != SD_F32)) return fail(SD_ERR_INVALID, "sd_tile_gather: bad argument");
    int rc = launch_tile_gather(vol, dtype == SD_U8 ? 1 : 4, VD, VH, VW, oz, oy, ox, tile, TD, TH, TW,
                                reinterpret_cast<hipStream_t>(stream));
    return rc == SD_OK ? rc : fail(rc, "sd_tile_gather launch failed");
}

int sd_tile_scatter(const void* tile, int dtype, int C, int TD, int TH, int TW, int cz, int cy, int cx, int KD, int KH,
                    int KW, void* vol, int VD, int VH, int VW, int oz, int oy, int ox, void* stream) {
    if (!vol || !tile || (dtype != SD_U8 && dtype != SD_F32)) return fail(SD_ERR_INVALID, "sd_tile_scatter: bad argument");
    if (cz < 0 || cy < 0 || cx < 0 || cz + KD > TD || cy + KH > TH || cx + KW > TW || oz < 0 || oy < 0 || ox < 0)
        return fail(SD_ERR_INVALID, "sd_tile_scatter: crop outside tile");
    int rc = launch_tile_scatter(tile, dtype == SD_U8 ? 1 : 4, C, TD, TH, TW, cz, cy, cx, KD, KH, KW, vol, VD, VH, VW,
                                 oz, oy, ox, reinterpret_cast<hipStream_t>(stream));
    return rc == SD_OK ? rc : fail(rc, "sd_tile_scatter launch failed");
}

int sd_downsample2(const void* src, int dtype, int D, int H, int W, void* dst, void* stream) {
    if (!src || !dst || D <= 0 || H <= 0 || W <= 0 || (dtype != SD_U8 && dtype != SD_U64))
        return fail(SD_ERR_INVALID, "sd_downsample2: bad argument");
    int rc = launch_downsample2(src, dtype == SD_U8 ? 1 : 8, D, H, W, dst, reinterpret_cast<hipStream_t>(stream));
    return rc == SD_OK ? rc : fail(rc, "sd_downsample2 launch failed");
}

int sd_box_majority(const uint8_t* vol, int D, int H, int W, const int32_t* origins_zyx, size_t n, int ez, int ey, int ex,
                    double thresh_proba, double thresh_majority, uint8_t* out, void* stream) {
    if (!vol || (!origins_zyx && n) || (!out && n) || D <= 0 || H <= 0 || W <= 0 || ez <= 0 || ey <= 0 || ex <= 0)
        return fail(SD_ERR_INVALID, "sd_box_majority: bad argument");
    if (thresh_proba != thresh_proba) return fail(SD_ERR_INVALID, "sd_box_majority: NaN threshold");
    // (uint8 p > t) <=> p >= floor(t) + 1, exact for any real t
    const double c = std::floor(thresh_proba) + 1.0;
    const int cut = c < 0.0 ? 0 : (c > 256.0 ? 256 : (int)c);
    int rc = launch_box_majority(vol, D, H, W, origins_zyx, (long)n, ez, ey, ex, cut, thresh_majority, out,
                                 reinterpret_cast<hipStream_t>(stream));
    return rc == SD_OK ? rc : fail(rc, "sd_box_majority launch failed");
}

int sd_postproc_labels(const uint8_t* probs, int C, size_t nvox, const int32_t* ids, const double* thresholds, int n_ids,
                       void* out, int out_dtype, void* stream) {
    if (!probs || !ids || !thresholds || !out || n_ids <= 0 || n_ids > 16)
        return fail(SD_ERR_INVALID, "sd_postproc_labels: bad argument");
    if (out_dtype != SD_U8 && out_dtype != SD_U64) return fail(SD_ERR_INVALID, "sd_postproc_labels: out_dtype");
    LabelArgs a{};      // one threshold -> cut rule for the fused and the separate path (NaN rejected, cuts in [0, 256])
    if (!make_label_args(C, ids, thresholds, n_ids, a))
        return fail(SD_ERR_INVALID, "sd_postproc_labels: id out of range or NaN threshold");
    int rc = launch_labels(probs, nvox, a, out, out_dtype == SD_U64, reinterpret_cast<hipStream_t>(stream));
    return rc == SD_OK ? rc : fail(rc, "sd_postproc_labels launch failed");
}

int sd_debug_read_buffer(sd_model* m, int buf, const void* ws, float* out, int32_t* dims4, void* stream) {
    if (m && m->f32) return f32_read_buffer(m->f32, buf, ws, out, dims4, reinterpret_cast<hipStream_t>(stream));
    if (!m || buf <= 0 || buf >= m->nbuf || m->dims.empty()) return fail(SD_ERR_INVALID, "sd_debug_read_buffer: bad argument");
    const Dims a = m->dims[buf];
    if (dims4) { dims4[0] = m->bufC[buf]; dims4[1] = a.d; dims4[2] = a.h; dims4[3] = a.w; }
    if (!out) return SD_OK;
    int rc = launch_read_buffer(reinterpret_cast<const char*>(ws) + m->buf_off[buf], m->act_dtype, m->bufC[buf],
                                m->bufCp[buf], a.d, a.h, a.w, out, reinterpret_cast<hipStream_t>(stream));
    return rc == SD_OK ? rc : fail(rc, "sd_debug_read_buffer launch failed");
}

}  // extern "C"
