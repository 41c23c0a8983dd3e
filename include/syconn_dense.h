/*
 * syconn_dense.h -- C ABI of libsyconn_dense_hip.so: the MI355X (gfx950) compute layer under SyConn's chunked
 * dense 3D-CNN prediction path.
 *
 * The reference has NO FFI on this path: its boundary is the Python API of
 *   syconn/handler/prediction.py:594-868  (predict_dense_to_kd / dense_predictor / dense_predicton_helper)
 * which calls the third-party elektronn3.inference.Predictor (prediction.py:770-781, 863) which in turn runs a
 * TorchScript elektronn3 U-Net through torch/cuDNN.  This header is the C ABI the build adds UNDER that Python
 * API; every entry point cites the reference step it replaces.  INTEGRATION.md shows the
 * ctypes binding a SyConn maintainer would add.
 *
 * Conventions
 *   - return 0 (SD_OK) on success, negative on error; sd_last_error() gives the message (thread-local).
 *     SD_ERR_NOMEM is mapped by the Python layer to RuntimeError so that the reference's tile-halving retry
 *     loop (prediction.py:783-794) keeps working.
 *   - all device buffers are CALLER-OWNED (the host framework's allocator); the library owns only the packed
 *     weights inside an sd_model.  Pointers are plain device addresses, sizes are bytes / elements as stated.
 *   - every launch is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream); the caller
 *     synchronises.  A handle is bound to one device and is not thread-safe.
 *   - volumes are z,y,x (x fastest).  Network input is one channel, planar.  Network output is planar
 *     (C, D, H, W).  Activations inside the workspace are channel-blocked ([C/16][z][y][x][16]) in the model's
 *     activation dtype; that layout is private to the library.
 */
#ifndef SYCONN_DENSE_H
#define SYCONN_DENSE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SD_OK 0
#define SD_ERR_INVALID (-1) /* bad argument / unsupported layer combination */
#define SD_ERR_NOMEM (-2)   /* workspace too small or device allocation failed */
#define SD_ERR_HIP (-3)     /* HIP runtime error */
#define SD_ERR_NODEVICE (-4)

/* element types of caller-visible buffers */
enum sd_dtype { SD_U8 = 0, SD_F32 = 1, SD_BF16 = 2, SD_F16 = 3, SD_U64 = 4, SD_U32 = 5,
                SD_F16X2 = 6 /* only as act_dtype of sd_model_create: split-fp16 storage, see there */ };

/* what sd_forward writes: raw logits (model(inp)), softmax(1) (Predictor(apply_softmax=True), prediction.py:779),
 * or floor(255*softmax) as uint8 (dense_predicton_helper, prediction.py:864-865) */
enum sd_out_kind { SD_OUT_LOGITS_F32 = 0, SD_OUT_PROBS_F32 = 1, SD_OUT_PROBS_U8 = 2,
                   SD_OUT_LABELS_U8 = 3 /* only through sd_forward_labels_batch */ };

/* layer kinds of the network plan (the elektronn3 U-Net block structure) */
enum sd_op_kind {
    SD_OP_CONV = 1,      /* Conv3d k=(3,3,3)|(1,3,3), 'same' zero padding, +bias, +folded eval-BatchNorm, +ReLU;
                            src1 >= 0: input is cat((crop(src0), src1), channel) (UpConv merge, autocrop) */
    SD_OP_POOL = 2,      /* MaxPool3d k=(2,2,2)|(1,2,2), ceil_mode=True */
    SD_OP_UPCONV = 3,    /* ConvTranspose3d k=s=(2,2,2)|(1,2,2), +bias, +folded eval-BatchNorm, +ReLU */
    SD_OP_GROUPNORM = 4, /* in-place GroupNorm(groups) + ReLU on buffer src0, statistics over the region cropped to
                            the shape of buffer src1 (if >= 0) */
    SD_OP_FINAL = 5      /* Conv3d k=1 to `cout` classes (+ softmax / uint8 epilogue chosen at sd_forward) */
};

/* One layer.  Buffer id 0 is the network input (1 channel); every other id names an activation buffer inside the
 * workspace.  All *_off fields are FLOAT offsets into the weight blob given to sd_model_create, -1 if absent.
 * Weights are in PyTorch layout: Conv3d [cout][cin][kz][ky][kx], ConvTranspose3d [cin][cout][kz][ky][kx]. */
typedef struct sd_op_desc {
    int32_t kind;
    int32_t src0, src1, dst;
    int32_t cin0, cin1, cout;
    int32_t kz, ky, kx;
    int32_t relu;
    int32_t norm;   /* 0: none, 1: eval-mode BatchNorm folded into the layer (gamma/beta/mean/var offsets) */
    int32_t groups; /* SD_OP_GROUPNORM */
    float eps;
    int64_t w_off, b_off, gamma_off, beta_off, mean_off, var_off;
} sd_op_desc;

typedef struct sd_model sd_model;

/* Bind the calling thread to a device (replaces Predictor's device pick, row P1).  Returns SD_ERR_NODEVICE when no
 * gfx950 device is visible -- there is no CPU fallback in this library. */
int sd_init(int device_ordinal);

/* Number of visible HIP devices (0 if none). */
int sd_device_count(void);

/* Build a model: validate the plan, fold BatchNorm, convert + pack weights into MFMA fragment order and upload
 * them.  Replaces torch.jit.load(model.pts).to(device) (prediction.py:777, 1061-1062).
 * act_dtype: SD_BF16 or SD_F16 (storage type of activations / MFMA operands; accumulation is fp32); SD_F16X2 = the
 * REFERENCE-PRECISION plan ON THE MATRIX CORES: every activation and weight is kept as two fp16 numbers hi + lo (22 mantissa
 * bits; weights times a power of two per layer so that their lo parts stay normal) and every product is three fp16 MFMAs
 * Wlo.Xhi + Whi.Xhi + Whi.Xlo accumulated in fp32 -- fp32-level logits (what the reference computes, prediction.py:777-779)
 * at ~3x the cost of the SD_F16 plan; range-guarded like SD_F16 (sd_model_overflow); or SD_F32 = the
 * REFERENCE-PRECISION mode: fp32 storage and fp32 FMA arithmetic like the reference's own torch path (Predictor is built
 * without float16, prediction.py:777-779); planar activations, one plain launch per layer, ~25x slower than the bf16
 * plan -- for label-exactness checks against an fp32 implementation, not for throughput. */
int sd_model_create(const sd_op_desc* ops, int n_ops, const float* weights, size_t n_floats, int act_dtype,
                    sd_model** out);
void sd_model_destroy(sd_model* m);

/* Output box of interest for the following forward passes of this model (lo inclusive, hi exclusive, z,y,x in tile coordinates;
 * two NULLs: the whole tile again).  tiled_apply (elektronn3, row P3) keeps only the core of every tile and throws the overlap
 * rim away (prediction.py:777-779 overlap_shape; dense_predictor also crops the chunk's halo, :812): with a box set, the decoder
 * layers (up-convolutions, the convolutions behind them, the fused final layer) compute only the sub-boxes that box depends on
 * -- every value INSIDE the box is the one a whole-tile pass computes, bit for bit; what the output holds outside it is
 * unspecified.  The fused level-0 decoder kernel is replaced by its layers, so call this BEFORE sd_workspace_bytes.  Networks
 * with GroupNorm (statistics over whole tensors) and SD_F32 ignore the box. */
int sd_model_set_roi(sd_model* m, const int32_t* lo_zyx, const int32_t* hi_zyx);

/* Workspace bytes sd_forward needs for a (D,H,W) input tile (a multiple of 256); 0 on error.  sd_forward_batch needs
 * N times this value. */
size_t sd_workspace_bytes(const sd_model* m, int D, int H, int W);

/* One forward pass of the network on one tile = Predictor._predict (row P4: model(inp) [+ softmax(1)]).
 * in_dev: (D,H,W) planar, in_dtype SD_U8 (normalised as float32(v)/255, prediction.py:808) or SD_F32.
 * out_dev: (cout_final, D, H, W) planar, float32 or uint8 according to out_kind.
 * Size limits (SD_ERR_INVALID beyond them): a tile has fewer than 2^31 voxels and every activation tensor fewer than 2^32
 * 8-channel groups (the streaming passes decode element indices with 32-bit arithmetic); the reference's tiles are
 * 178 x 243 x 331 = 1.4e7 voxels. */
int sd_forward(sd_model* m, const void* in_dev, int in_dtype, int D, int H, int W, void* out_dev, int out_kind,
               void* workspace_dev, size_t ws_bytes, void* stream);

/* The same for N independent tiles of one shape in ONE set of launches = Predictor.predict's batch split
 * (`batch_size`, row P2): in_dev (N,D,H,W), out_dev (N,cout_final,D,H,W), both dense; workspace N *
 * sd_workspace_bytes(D,H,W).  Every kernel simply sees N times as many blocks, which fills the 256 CUs in the deep,
 * small layers of the network and amortises launch gaps; results are identical to N sd_forward calls. */
int sd_forward_batch(sd_model* m, const void* in_dev, int in_dtype, int N, int D, int H, int W, void* out_dev,
                     int out_kind, void* workspace_dev, size_t ws_bytes, void* stream);

/* fp16 range guard.  With act_dtype SD_F16 an activation above 65504 is stored as inf and reaches the final layer as inf / NaN
 * logits; the final-layer kernels raise a device flag when that happens (the reference computes in fp32 and cannot overflow
 * there, prediction.py:777-779).  Copies the flag to *flag_out (1 = some forward pass since the last call overflowed: its
 * results are invalid; rerun with SD_BF16 or SD_F32) and clears it.  SYNCHRONISES `stream` (the stream the forwards were
 * enqueued on).  Always 0 for SD_BF16 / SD_F32 models, whose storage types share fp32's exponent range. */
int sd_model_overflow(sd_model* m, void* stream, int* flag_out);

/* Forward pass with the label rule of dense_predictor (prediction.py:813-833, row A7) applied in the final layer's
 * epilogue: out_dev (N, D, H, W) uint8 = what sd_postproc_labels computes from the SD_OUT_PROBS_U8 result of
 * sd_forward_batch -- label = 0; for i in order: if floor(255*p[ids[i]]) > thresholds[i] then label = ids[i] -- without
 * writing and re-reading the probability maps.  ids in [0, cout_final), n_ids <= 16, thresholds in uint8 units. */
int sd_forward_labels_batch(sd_model* m, const void* in_dev, int in_dtype, int N, int D, int H, int W,
                            const int32_t* ids, const double* thresholds, int n_ids, uint8_t* out_dev,
                            void* workspace_dev, size_t ws_bytes, void* stream);

/* tiled_apply helpers (row P3).  Gather: copy the (TD,TH,TW) box starting at (oz,oy,ox) -- which may lie partly
 * outside the (VD,VH,VW) volume -- into a dense tile, zeros outside (zero-padded tile extraction).
 * dtype: SD_U8 or SD_F32 (copied verbatim). */
int sd_tile_gather(const void* vol_dev, int dtype, int VD, int VH, int VW, int oz, int oy, int ox, void* tile_dev,
                   int TD, int TH, int TW, void* stream);
/* Scatter: for each of C channels copy tile[c, cz:cz+KD, cy:cy+KH, cx:cx+KW] (tile is (C,TD,TH,TW)) into
 * vol[c, oz:oz+KD, oy:oy+KH, ox:ox+KW] (vol is (C,VD,VH,VW)); the crop-only stitching of tiled_apply and the
 * halo crop of prediction.py:812.  dtype: SD_U8 or SD_F32. */
int sd_tile_scatter(const void* tile_dev, int dtype, int C, int TD, int TH, int TW, int cz, int cy, int cx, int KD,
                    int KH, int KW, void* vol_dev, int VD, int VH, int VW, int oz, int oy, int ox, void* stream);

/* The label rule of prediction.py:813-833 for ONE multi-id target: out[v] = 0; for i in order:
 * if (probs[ids[i]][v] > thresholds[i]) out[v] = ids[i].  probs: (C, nvox) uint8; `ids` / `thresholds` are HOST
 * arrays (n_ids <= 16), thresholds already resolved by the caller (None -> 127.5, t<1 -> 255*t) and compared
 * exactly as numpy compares uint8 with a float64; out_dtype SD_U8 or SD_U64 (save_seg takes uint64). */
int sd_postproc_labels(const uint8_t* probs_dev, int C, size_t nvox, const int32_t* ids, const double* thresholds,
                       int n_ids, void* out_dev, int out_dtype, void* stream);

/* `height` pieces of `width_bytes` contiguous bytes each, `*_pitch` bytes apart (hipMemcpy2DAsync on `stream`): how a (z-range,
 * y-range, all x) strip of a (z,y,x) volume travels between a page-locked host volume and its device copy without touching the
 * rows around it (syconn_amd.parallel: per-strip result downloads).  kind: 0 host -> device, 1 device -> host, 2 device -> device. */
int sd_memcpy2d_async(void* dst, size_t dst_pitch, const void* src, size_t src_pitch, size_t width_bytes, size_t height, int kind,
                      void* stream);

/* Measurement / test support.  With n_slots > 0 every layer launch of an sd_forward is bracketed by HIP events
 * recorded on the launch stream; forward number k uses event set k % n_slots (counted from this call), so a timed
 * region of many forwards can be read back afterwards without synchronising inside it.  n_slots = 0 switches off.
 * sd_profile_read returns the per-layer milliseconds of one slot (the caller synchronised the stream before). */
int sd_profile_enable(sd_model* m, int n_slots);
int sd_profile_read(sd_model* m, int slot, float* ms_per_op, int n_ops);
/* With profiling enabled every convolution launch also stamps, from its first workgroup, the shader cycle counter and the constant
 * 100 MHz counter on entry and on exit: stamps[4 * op + {0,1,2,3}] = {cycles at entry, ticks at entry, cycles at exit, ticks at exit}
 * (zeros for ops without a convolution launch of their own) -- shader clock of the launch = (c1 - c0) / ((t1 - t0) * 10 ns).
 * The caller synchronised the stream before. */
int sd_profile_read_clocks(sd_model* m, int slot, uint64_t* stamps, int n_ops);
/* Box calibration: a chip-wide dense bf16 MFMA loop (n_workgroups x waves_per_workgroup waves, `iters` x 4 v_mfma_f32_32x32x16_bf16
 * each, no memory traffic) launched back to back for at least min_seconds; reports the LAST launch: sustained TFLOP/s and the median
 * shader clock of its waves.  random_operands != 0: four pseudo-random A / B fragment pairs take turns (operand buses toggle as in
 * a convolution; the part draws more power and holds a lower clock than on the constant operands of random_operands = 0).
 * Synchronises `stream`. */
int sd_probe_mfma_rate(int n_workgroups, int waves_per_workgroup, int iters, double min_seconds, int random_operands,
                       double* tflops_out, double* shader_ghz_out, void* stream);
/* Copy activation buffer `buf` of the last sd_forward out of the workspace as float32 planar (C, d, h, w);
 * dims are returned in dims4 = {C, d, h, w}.  out_dev may be NULL to query dims only. */
int sd_debug_read_buffer(sd_model* m, int buf, const void* workspace_dev, float* out_dev, int32_t* dims4,
                         void* stream);
int sd_model_num_ops(const sd_model* m);
/* Number of plan ops the last sd_forward* call executed as launches of their own (the others ran inside a fused launch:
 * pooling / final layer in a convolution's epilogue, first convolution inside the second, the level-0 decoder inside its
 * up-convolution's launch).  For tests that must know which plan a shape was served by. */
int sd_debug_last_launch_count(const sd_model* m);
/* Which kernel computed plan op `op` in the last sd_forward* call: returns the index of the op whose LAUNCH did it (`op` itself, or the
 * op it is fused into: a pooling / final layer in a convolution's epilogue, a first convolution inside the second, the members of the
 * level-0 decoder), -1 when it did not run; the kernel symbol of that launch ("k_conv_mfma<bf16,3x3x3,NT=2,WAVES=8,NSLOT=0,MT=4,MODE=0>")
 * is copied into buf (n bytes, NUL-terminated).  What bench.py names its roofline kernel by -- the launchers note what they picked. */
int sd_debug_op_kernel(const sd_model* m, int op, char* buf, int n);

const char* sd_last_error(void);
const char* sd_version(void);

/* ---- KNOSSOS overlay cubes (host side, no GPU needed) -------------------------------
 * Snappy raw-format codec for "*.seg.sz.zip": knossos_utils (third-party, called at
 * /root/reference/syconn/handler/prediction.py:700-702, 835-843) stores each 128^3 uint64 cube as
 * python-snappy compress(cube.tobytes()) inside a zip member.  Restates google/snappy format_description.txt (pinned
 * upstream: snappy 1.1.8 / python-snappy 0.6.0); any conforming stream decodes, the encoder's output is decodable by
 * any conforming decoder.  All return SD_OK or SD_ERR_INVALID (corrupt / truncated stream, capacity too small). */
size_t sd_snappy_max_compressed_length(size_t n);
int sd_snappy_compress(const void* src, size_t n, void* dst, size_t dst_capacity, size_t* dst_len);
int sd_snappy_uncompressed_length(const void* src, size_t n, size_t* result);
int sd_snappy_uncompress(const void* src, size_t n, void* dst, size_t dst_capacity, size_t* dst_len);

/* Order-0 down-sampling by 2 per axis on the device = one level of the mag pyramid KnossosDataset.save_raw /
 * save_seg(..., mags=[m, 2m, 4m], fast_resampling=True) writes (prediction.py:834-843): dst[z,y,x] = src[2z,2y,2x],
 * dst dims = ceil(src dims / 2).  dtype: SD_U8 or SD_U64. */
int sd_downsample2(const void* src_dev, int dtype, int D, int H, int W, void* dst_dev, void* stream);

/* ---- first consumer of the myelin probability map -----------------------------------
 * map_myelin2coords (/root/reference/syconn/reps/super_segmentation_helper.py:550-615): for each of n boxes with
 * origin origins_zyx[3i..3i+2] (voxels of `vol`, may lie partly outside = zeros, like kd.load_raw) and extent
 * (ez,ey,ex): out[i] = (count(vol > thresh_proba) / (ez*ey*ex) > thresh_majority), the division in float64 as in
 * the reference (:611-612).  vol: (D,H,W) uint8 on the device; origins int32 and out uint8 on the device. */
int sd_box_majority(const uint8_t* vol_dev, int D, int H, int W, const int32_t* origins_zyx_dev, size_t n, int ez, int ey,
                    int ex, double thresh_proba, double thresh_majority, uint8_t* out_dev, void* stream);

/* ---- label-volume statistics ------------------------------------------------------------
 * Device counterpart of the reference's Cython natives /root/reference/syconn/extraction/find_object_properties_C.pyx:
 * find_object_properties (:24-49), map_subcell_C (:72-109), map_subcell_extract_props (:112-192).  Volumes are
 * (X,Y,Z) with z fastest (the reference indexes chunk[x, y, z]), dtype SD_U32 or SD_U64, id 0 = background.
 * One streaming pass fills open-addressing hash tables in CALLER-OWNED device memory:
 *   object table  (sd_objtable_bytes(cap) bytes, cap a power of two <= 2^31): per non-zero id the smallest raster index
 *                 (= the reference's representative coordinate, the first voxel its scan meets), the voxel count and the
 *                 bounding box [min, max + 1);
 *   pair table    (sd_pairtable_bytes(cap) bytes): per (subcell id, cell id) the number of voxels where both are set.
 * sd_segstats_scan initialises the tables itself.  cell_dev may be NULL (properties of the `sub` volumes only); n_sub may
 * be 0 (= find_object_properties(cell)); want_props = 0 computes the overlap counts only (= map_subcell_C).
 * status_dev: int32[2], set to 1 when the object tables ([0]) / pair tables ([1]) were too small -- the caller retries
 * with a larger capacity (results of an overflowed pass are incomplete and must be discarded). */
size_t sd_objtable_bytes(size_t capacity);
size_t sd_pairtable_bytes(size_t capacity);
int sd_segstats_scan(const void* cell_dev, const void* const* sub_devs /* host array of device pointers */, int n_sub,
                     int dtype, int X, int Y, int Z, void* cell_table, void* const* sub_tables, size_t cap_obj,
                     void* const* pair_tables, size_t cap_pair, int want_props, int32_t* status_dev, void* stream);
/* Turn a filled object table into dense arrays (any order): ids / first raster index / voxel count (uint64 each) and
 * bbox int32[n][6] = (min x,y,z, max+1 x,y,z).  *count_dev = number of objects (may exceed max_out: then only max_out
 * records were written). */
int sd_segstats_compact_objects(const void* table, size_t cap_obj, uint64_t* ids_dev, uint64_t* first_dev,
                                uint64_t* size_dev, int32_t* bbox_dev, size_t max_out, uint64_t* count_dev, void* stream);
/* Dense (subcell id, cell id, overlap count) triples of one pair table; the ids are looked up in the two object tables
 * the same sd_segstats_scan call filled. */
int sd_segstats_compact_pairs(const void* pair_table, size_t cap_pair, const void* sub_table, const void* cell_table,
                              size_t cap_obj, uint64_t* sub_ids_dev, uint64_t* cell_ids_dev, uint64_t* counts_dev,
                              size_t max_out, uint64_t* count_dev, void* stream);

/* ---- dataset-wide merge of per-chunk statistics (the chunk driver around the natives) ----------------
 * Replaces, on record arrays in HBM, what /root/reference/syconn/proc/sd_proc.py does with Python dictionaries per chunk:
 * the filter of _map_subcell_extract_props_thread (:640-650, :657-670: an object that lies purely inside its chunk -- on none of the
 * six faces -- and has fewer than min_obj_vx voxels is dropped, for organelles from the overlap counts too), merge_prop_dicts
 * (:1248-1273) and merge_map_dicts (:1300-1322).
 * sd_chunkprops_append: reads an object table sd_segstats_scan filled for a chunk of extent (X,Y,Z) at origin (ox,oy,oz) and appends
 *   one record per surviving object at *cursor_dev (a device counter the caller zeroes once per dataset; NOT reset here): id, global
 *   representative coordinate int32[3], global box int32[6] = (min, max + 1), voxel count.  min_obj_vx <= 1: no filter.  Records
 *   beyond max_records are not written but still counted: the caller compares the final cursor with max_records.
 * sd_chunkpairs_append: the same for a pair table: (subcell id, cell id, overlap voxels), filter by the SUBCELL table's entry. */
int sd_chunkprops_append(const void* table, size_t cap_obj, int X, int Y, int Z, int ox, int oy, int oz, uint64_t min_obj_vx,
                         uint64_t* ids_dev, int32_t* rc_dev, int32_t* bbox_dev, uint64_t* sizes_dev, size_t max_records,
                         uint64_t* cursor_dev, void* stream);
int sd_chunkpairs_append(const void* pair_table, size_t cap_pair, const void* sub_table, const void* cell_table, size_t cap_obj,
                         int X, int Y, int Z, uint64_t min_obj_vx, uint64_t* sub_ids_dev, uint64_t* cell_ids_dev, uint64_t* counts_dev,
                         size_t max_records, uint64_t* cursor_dev, void* stream);
/* Merge n appended records (chunks appended in processing order): stable sort by id + one segment per id.
 *   uniq_ids / tot_sizes / last_rc[.][3]: per id (ascending) the summed voxel count and the representative coordinate of the LAST
 *   chunk holding it (dict.update order of merge_prop_dicts); seg_begin[u]: first position of id u in bbox_sorted int32[n][6], the
 *   per-chunk boxes in chunk order (the reference keeps a list of boxes per id); *n_unique_dev: number of ids.  Output arrays hold n
 *   entries each; temp_dev: sd_propmerge_temp_bytes(n) bytes.
 * sd_propmerge_pairs: unique (subcell id, cell id) ascending lexicographically with summed counts. */
size_t sd_propmerge_temp_bytes(size_t n_records);
int sd_propmerge_objects(const uint64_t* ids_dev, const uint64_t* sizes_dev, const int32_t* rc_dev, const int32_t* bbox_dev, size_t n,
                         uint64_t* uniq_ids_dev, uint64_t* tot_sizes_dev, int32_t* last_rc_dev, uint32_t* seg_begin_dev,
                         int32_t* bbox_sorted_dev, uint64_t* n_unique_dev, void* temp_dev, size_t temp_bytes, void* stream);
int sd_propmerge_pairs(const uint64_t* sub_ids_dev, const uint64_t* cell_ids_dev, const uint64_t* counts_dev, size_t n,
                       uint64_t* out_sub_dev, uint64_t* out_cell_dev, uint64_t* out_counts_dev, uint64_t* n_unique_dev, void* temp_dev,
                       size_t temp_bytes, void* stream);

/* ---- globally unique objects across chunks (the steps behind the per-chunk first stage) ----------------
 * make_unique_labels (/root/reference/syconn/extraction/object_extraction_steps.py:369-443: `matrix[matrix > 0] += offset` on the
 * chunk's component labels widened to uint64, offset = number of components in all earlier chunks,
 * object_extraction_wrapper.py:300-312): labels_dev int32[n] -> out_dev uint64[n]. */
int sd_labels_make_unique(const int32_t* labels_dev, size_t n, uint64_t offset, uint64_t* out_dev, void* stream);
/* Cut the box [x0, x0+nx) x [y0, y0+ny) x [z0, z0+nz) out of an (X,Y,Z) uint64 label volume (z fastest) into a contiguous
 * (nx,ny,nz) array, optionally through a look-up table: dst = lut[src] (lut_dev NULL: dst = src).  Serves
 *   * apply_merge_list (object_extraction_steps.py:717-731): crop the chunk's overlap margin and map every id through the
 *     merge list (`id_changer[this_cc]`), lut_len = max_label + 1;
 *   * the face slabs make_stitch_list compares (:560-575, cut_array_in_one_dim), whose co-occurring id pairs sd_segstats_scan counts.
 * status_dev (optional int32): set to 1 when an id >= lut_len was met (it passes through unmapped). */
int sd_labels_box_lut(const uint64_t* src_dev, int X, int Y, int Z, int x0, int y0, int z0, int nx, int ny, int nz,
                      const uint64_t* lut_dev, size_t lut_len, uint64_t* dst_dev, int32_t* status_dev, void* stream);

/* ---- probability map -> object segmentation, first stage ---------------------------------
 * Non-watershed branches of _object_segmentation_thread (/root/reference/syconn/extraction/object_extraction_steps.py:
 * 316-317 threshold, 354-358 morphology + scipy.ndimage.label) with the morphology semantics of
 * /root/reference/syconn/proc/image.py:357-438, 485-507 (_multi_mop_findobjects / apply_morphological_operations) on a
 * binary volume.  prob_dev: (X,Y,Z) uint8, z fastest (the reference's arrays are x,y,z here).
 *   threshold   uint8 scale, mask = prob > threshold (0: prob already is a 0/1 mask, object_extraction_steps.py:316);
 *   ops/iterations (HOST arrays, n_ops entries): the reference's operation list with runs of equal operations merged
 *               into `iterations` (image.py:510-519).  SD_MOP_EROSION selects the reference's watershed branch
 *               (:319-352) and is rejected here with SD_ERR_INVALID: use sd_object_segmentation_watershed below.
 *   struct_host (sx,sy,sz) uint8 HOST array, odd extents: the structuring element (get_aniso_struct, image.py:522-539);
 *   labels_dev  (X,Y,Z) int32: 6-connected components numbered 1..N in raster order of their first voxel, 0 = background
 *               -- identical to scipy.ndimage.label; *max_label_dev = N;
 *   mask_out_dev optional (X,Y,Z) uint8: the binary volume after the morphology.
 * Workspace: sd_objseg_workspace_bytes(X, Y, Z, largest `iterations` of any closing / dilation). */
enum sd_morph_op { SD_MOP_OPENING = 1, SD_MOP_CLOSING = 2, SD_MOP_DILATION = 3, SD_MOP_EROSION = 4 };
size_t sd_objseg_workspace_bytes(int X, int Y, int Z, int max_iterations);
int sd_object_segmentation(const uint8_t* prob_dev, int X, int Y, int Z, double threshold, const int32_t* ops,
                           const int32_t* iterations, int n_ops, const uint8_t* struct_host, int sx, int sy, int sz,
                           int32_t* labels_dev, int32_t* max_label_dev, uint8_t* mask_out_dev, void* workspace_dev,
                           size_t ws_bytes, void* stream);

/* The WATERSHED branch of the same function (object_extraction_steps.py:319-352) -- what SyConn's default config selects for
 * mi / sj / vc (config.yml:130-136: opening, closing, erosion(s)); taken when the operation list contains 'binary_erosion':
 *   ops / iterations          the operations BEFORE the first erosion -> tmp_data (:320-322);
 *   seed_ops / ...            the operations from the first erosion on (runs merged separately, image.py:510-519), applied to a
 *                             copy of tmp_data -> scipy.ndimage.label -> markers (:323-327);
 *   min_seed_vx               > 1: markers with fewer voxels are deleted and the freed ids handed to the largest surviving ids
 *                             (the reference's hole filling + relabel_vol, :330-347; block_processing_C.pyx:161-169);
 *   pixel_pitch_xyz           HOST int32[3] voxel size (scaling.astype(uint32)): the distance transform of tmp_data to its
 *                             background is exact Euclidean with that pitch (vigra distanceTransform(background=False), :349);
 *   labels_dev                skimage.segmentation.watershed(-distance, markers, mask=tmp_data) (:351): priority flood, 6-connected;
 *   markers_out_dev           optional (X,Y,Z) int32: the relabelled marker volume -- everything up to here is scipy / numpy in the
 *                             reference and reproduced bit for bit (pinned by tests/golden/g10_objseg_ws.npz);
 *   distance_out_dev          optional (X,Y,Z) float32: the distance transform.
 * vigra and skimage are absent from the reference tree and this image: distance transform and flood restate their published
 * algorithms (flood order: value, then age, marker voxels of equal value by raster index) and are parity-UNPINNED. */
size_t sd_objseg_watershed_workspace_bytes(int X, int Y, int Z, int max_iterations);
int sd_object_segmentation_watershed(const uint8_t* prob_dev, int X, int Y, int Z, double threshold, const int32_t* ops,
                                     const int32_t* iterations, int n_ops, const int32_t* seed_ops,
                                     const int32_t* seed_iterations, int n_seed_ops, const uint8_t* struct_host, int sx, int sy,
                                     int sz, int min_seed_vx, const int32_t* pixel_pitch_xyz, int32_t* labels_dev,
                                     int32_t* max_label_dev, int32_t* markers_out_dev, float* distance_out_dev,
                                     uint8_t* mask_out_dev, void* workspace_dev, size_t ws_bytes, void* stream);

/* The flood of that branch on its own: skimage.segmentation.watershed(-distance, markers, mask) (:351; watershed_raveled with
 * connectivity 1, no compactness, no watershed line) for distance^2 = d2_dev (int32 >= 0), an int32 marker volume (0 = none;
 * markers outside the mask are ignored) and a uint8 mask, all (X,Y,Z) with z fastest.  Pop order: higher d2 first, then
 * first-in first-out, marker voxels (all queued before anything else) among themselves by raster index; a voxel takes the label
 * of the popped neighbour that reaches it first.  Runs level-synchronously, one workgroup per mask component that holds several
 * markers (csrc/sd_objseg.hip::k_ws_flood); SD_WS_SEQUENTIAL=1 in the environment selects a sequential restatement (one lane per
 * component) that the tests cross-check it with.  Workspace: sd_objseg_watershed_workspace_bytes(X, Y, Z, 0). */
int sd_marker_flood(const int32_t* d2_dev, const int32_t* markers_dev, const uint8_t* mask_dev, int X, int Y, int Z,
                    int32_t* labels_dev, int32_t* max_label_dev, void* workspace_dev, size_t ws_bytes, void* stream);

/* Gaussian pre-smoothing of a probability map followed by the threshold (object_extraction_steps.py:296-297
 * `gaussianSmoothing(tmp_data, sigmas[...])`, vigra; :316-317 `tmp_data > thresholds[...]`) -- the optional first step of
 * _object_segmentation_thread (SyConn's own pipeline passes no sigmas).  prob_dev (X,Y,Z) uint8, z fastest; sigma_xyz HOST
 * double[3] per axis in voxels (0: axis not smoothed).  vigra's published algorithm restated (parity UNPINNED, vigra is absent
 * here): separable, axis order x, y, z, window radius int(3 sigma + 0.5) (>= 1, <= 64) of exp(-t^2 / 2 sigma^2) normalised to
 * sum 1, reflective border without repeating the edge, sums in double, every pass stored as float32.
 *   mask_dev     (X,Y,Z) uint8: 1 where smoothed > threshold (uint8 scale like the input), else 0 -- pass it to
 *                sd_object_segmentation* with threshold 0;
 *   smoothed_dev optional (X,Y,Z) float32: the smoothed map. */
size_t sd_gauss_workspace_bytes(int X, int Y, int Z);
int sd_gaussian_threshold(const uint8_t* prob_dev, int X, int Y, int Z, const double* sigma_xyz, double threshold,
                          uint8_t* mask_dev, float* smoothed_dev, void* workspace_dev, size_t ws_bytes, void* stream);

/* ---- host-side helpers of the chunk pipeline (no GPU) -----------------------------------------------------------------
 * Multi-threaded strided copy of an (nz, ny, nx)-byte box between two uint8 host arrays whose x-rows are contiguous
 * (strides in bytes), and a multi-threaded memset: what numpy slicing does on one core when the reference cuts a chunk
 * (+ halo) out of a volume and crops the result (/root/reference/syconn/handler/prediction.py:806-812). */
int sd_host_box_copy(const uint8_t* src, int64_t src_stride_z, int64_t src_stride_y, uint8_t* dst, int64_t dst_stride_z,
                     int64_t dst_stride_y, int64_t nz, int64_t ny, int64_t nx, int n_threads);
int sd_host_zero(uint8_t* dst, int64_t nbytes, int n_threads);

/* Window clipping for model tiles of which only a part is wanted (host arithmetic on the plan, no GPU).  The reference's
 * chunk grid overhangs the dataset (fit_box_size=True, /root/reference/syconn/handler/prediction.py:679-683), every chunk is
 * predicted with a halo ring that is cropped afterwards (:812), and tiled_apply (elektronn3, row P3) runs every tile on its
 * full window.  Along `axis` (0 = z, 1 = y, 2 = x) the outputs lo <= index < hi of a window of `full` voxels depend on a cone of
 * the input only; this returns the sub-window [*start, *start + *extent) on which those outputs have the SAME values as on the
 * whole window: the far border of every layer ('same' padding, partial ceil-mode pooling windows, the up-convolution crop)
 * stays outside every cone (backward pass: conv k reads k/2 further, pooling f reads f times as far, a transposed conv f
 * ceil(/ f)), and the near border moves only by multiples of the network's total pooling stride along the axis, never past
 * the lowest index a wanted output reads in any buffer.  *extent is a multiple of `multiple` (or what is left of `full`), *start a
 * multiple of lcm(stride, multiple).  Plans with SD_OP_GROUPNORM (statistics over the whole window) return (0, full). */
int sd_plan_clip_window(const sd_op_desc* ops, int n_ops, int axis, int lo, int hi, int full, int multiple, int* start,
                        int* extent);

#ifdef __cplusplus
}
#endif
#endif /* SYCONN_DENSE_H */
