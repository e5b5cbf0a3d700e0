/* libymk_hip.so - C ABI of the MI355X (gfx950) DocumentAnalyzer hot path.
 *
 * The reference (kotaro-kinoshita/yomitoku) is pure Python and has no FFI.  The seam this
 * ABI plugs into is the one its optional ONNX backend already uses: one call per network
 * forward with C-contiguous buffers in and out
 *   sess.run(["output"], {"input": ndarray})   text_detector.py:122-125,
 *                                              text_recognizer.py:248-251,
 *                                              layout_parser.py:252-258,
 *                                              table_structure_recognizer.py:262-268
 * and the `self.model(tensor)` call it stands in for (text_detector.py:127-129 etc.).
 * INTEGRATION.md shows the ctypes stub a yomitoku maintainer would add.
 *
 * Rules
 *   - plain pointers and sizes only; `stream` is a hipStream_t passed as void* (NULL = default);
 *   - pointers named *_dev are device (HBM) addresses owned by the caller; the library owns
 *     weights and workspace, and allocates nothing on the per-call path once a shape was seen;
 *   - every function returns 0 on success, non-zero on failure; ymk_last_error() returns the
 *     message of the calling thread's last failure;
 *   - a model handle may be used from one thread at a time; different handles are independent.
 */
#ifndef YMK_H
#define YMK_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ymk_model ymk_model;

int ymk_version(void);
const char* ymk_last_error(void);
/* number of visible HIP devices, or -1 */
int ymk_device_count(void);

/* ---- model lifecycle (replaces BaseModule.load_model, base.py:80-86) -------------------
 * kind: "dbnet" | "parseq" | "rtdetr".  Weights are handed over tensor by tensor under the
 * reference's state-dict names (host fp32, C-contiguous), scalar hyper-parameters by name,
 * then ymk_model_finalize() folds BatchNorm, repacks panels and uploads to `device`. */
ymk_model* ymk_model_create(const char* kind, int device);
void ymk_model_destroy(ymk_model* m);
/* One parameter may also be set on a finalized model: "conv_split" - how this model's convolutions / linear layers that fill
 * the chip multiply (ymk_conv_split.hip; grid-starved launches, attention and the fused greedy step are exact fp32 always):
 *   16  the default: fp32 operands scaled by powers of two and cut into two fp16 planes (11 + 11 significand bits), three
 *       v_mfma_f32_32x32x16_f16 per product tile, fp32 accumulation - products good to 2^-21, below the fp32 accumulation's
 *       own rounding for K >= 64 (measured: as far from the exact kernel as the exact kernel in another summation order)
 *   0   exact fp32 MFMA (v_mfma_f32_32x32x2_f32, an fmaf chain in k order)
 *   2 / 3   two / three bf16 planes, 3 / 6 MFMAs (2: products to 2^-15 only - evaluation)
 *   -1  follow the process-wide ymk_debug_option("conv_split") if set, else the default.
 * parseq only: "conv_split_encoder" - the same choice for the ViT blocks' linear layers alone (the decoder and the
 * vocabulary head then follow "conv_split"). */
int ymk_model_set_param(ymk_model* m, const char* key, double value);
int ymk_model_set_tensor(ymk_model* m, const char* name, const float* host_data, int ndim, const int64_t* dims);
int ymk_model_finalize(ymk_model* m);
/* Size the model's workspace once for the largest forward the caller will issue, so that forwards within the bound
 * never allocate or free device memory whatever their shape ("allocates nothing on the per-call path" also for calls
 * whose shape changes every time: the pages of a wave differ in size, the mini-batches of text_recognizer.py:158-203
 * differ in rows and width).  dbnet: n images of h x w (either orientation); rtdetr: n images of h x w; parseq: n
 * text lines over all groups of a call, each at most w pixels wide (h ignored). */
int ymk_model_reserve(ymk_model* m, int n, int h, int w, void* stream);
/* bytes of HBM held by the model's weights / workspace */
int64_t ymk_model_weight_bytes(const ymk_model* m);
int64_t ymk_model_workspace_bytes(const ymk_model* m);

/* ---- DBNet text detector (replaces DBNet.forward, models/dbnet_plus.py:243-246) --------
 * x_dev: fp32 [n][3][h][w] (h, w multiples of 32, as TextDetector.preprocess produces);
 * prob_dev: fp32 [n][1][h][w] = preds["binary"]. */
int ymk_dbnet_forward(ymk_model* m, const float* x_dev, int n, int h, int w, float* prob_dev, void* stream);

/* ---- PARSeq text recogniser (replaces PARSeq.forward, models/parseq.py:159-311) ----------
 * x_dev: fp32 [b][3][32][w] in [-1,1] (w a multiple of the patch width, <= 800: the dynamic-width
 * batches of TextRecognizer._collate, text_recognizer.py:146-156); logits_dev: fp32
 * [b][max_label_length+1][num_tokens-2] (always allocate the full 101 rows); *out_len receives the
 * number of valid rows per sample (101 when refine_iters >= 1, else the AR steps executed),
 * *ar_steps the greedy steps executed before every row held an <eos>.  The call blocks until the AR loop
 * has stopped (the host polls a mapped flag two steps behind the device - the reference's early-stop test,
 * models/parseq.py:245-250) and returns with the refinement pass still queued on the stream. */
int ymk_parseq_dims(ymk_model* m, int* num_steps, int* num_classes);
int ymk_parseq_forward(ymk_model* m, const float* x_dev, int b, int w, float* logits_dev, int* out_len, int* ar_steps,
                       void* stream);
/* The same forward over n_groups mini-batches in ONE call (TextRecognizer's per-page mini-batches, of one page or of
 * several: text_recognizer.py:158-203 runs them one after the other).  Group g is its own tensor x_dev[g]
 * ([b[g]][3][32][w[g]], padded to ITS widest crop exactly as _collate does - padding columns are ordinary ViT tokens, so
 * the groups are never re-padded to a common width); x_dev, b, w, out_len, ar_steps are HOST arrays of n_groups
 * entries.  Every layer runs once over the token rows of all groups laid end to end, attention reads per-sample
 * (row offset, length) tables, and one greedy loop serves all rows; logits_dev: [sum b][max_label_length+1][C] in
 * group order.  out_len[g] / ar_steps[g] are what group g's own ymk_parseq_forward call would have returned. */
int ymk_parseq_forward_groups(ymk_model* m, const float* const* x_dev, const int* b, const int* w, int n_groups,
                              float* logits_dev, int* out_len, int* ar_steps, void* stream);
/* what ParseqTokenizer.decode needs from softmax(logits) (parseq_tokenizer.py:79-87) without
 * materialising it: per row the arg-max class id and max probability. rows = b * out_len. */
int ymk_parseq_token_stats(const float* logits_dev, int rows, int num_classes, int* ids_dev, float* probs_dev,
                           void* stream);

/* ---- RT-DETRv2 layout parser / table-structure recogniser (replaces RTDETRv2.forward,
 * models/rtdetr.py:16-21).  x_dev: fp32 [b][3][640][640] in [0,1] (LayoutParser.preprocess,
 * layout_parser.py:195-199); logits_dev: fp32 [b][num_queries][num_classes] = pred_logits;
 * boxes_dev: fp32 [b][num_queries][4] = pred_boxes (cxcywh in [0,1]). */
int ymk_rtdetr_forward(ymk_model* m, const float* x_dev, int b, int h, int w, float* logits_dev, float* boxes_dev,
                       void* stream);

/* ---- fused pre-processing kernels (uint8 BGR page resident in HBM -> network input tensors) ----
 * ymk_det_preprocess: TextDetector.preprocess (text_detector.py:99-107; resize_shortest_edge's
 *   cv2.resize(INTER_AREA) + standardization_image, data/functions.py:196-247).  bgr_dev: uint8
 *   [h][w][3]; x_dev: fp32 [3][oh][ow] (oh, ow from resize_shortest_edge, computed by the caller).
 * ymk_pil_resize_to_chw: LayoutParser / TableStructureRecognizer.preprocess (layout_parser.py:195-199,
 *   table_structure_recognizer.py:169-186): BGR->RGB, PIL bilinear antialiased resize of the crop at
 *   (x0, y0) to oh x ow, ToTensor.  Coefficient tables follow Pillow's precompute_coeffs /
 *   normalize_coeffs_8bpc (bounds: [o][2] = first tap, count; coefs: [o][ksize] 22-bit ints).
 * ymk_crop_batch: ParseqDataset crops (data/dataset.py:105-124): perspective warp, optional 90 deg
 *   rotation, down-scale-only INTER_AREA to the 32 px canvas, ToTensor + Normalize(0.5, 0.5), -1
 *   padding to batch_w.  descs_dev: array of n records of ymk_crop_desc_size() bytes (layout in
 *   yomitoku_amd/csrc/ymk_image.hip: CropDesc); out_dev: fp32 [slots][3][out_h][batch_w]. */
int ymk_det_preprocess(const unsigned char* bgr_dev, int h, int w, int oh, int ow, float* x_dev, void* stream);
int ymk_pil_resize_to_chw(const unsigned char* page_dev, int page_w, int x0, int y0, const int* xbounds_dev,
                          const int* xcoef_dev, int ksize_x, const int* ybounds_dev, const int* ycoef_dev, int ksize_y,
                          int oh, int ow, float* x_dev, void* stream);
/* ymk_pil_resize_batch_to_chw: ymk_pil_resize_to_chw for n crops in one launch (TableStructureRecognizer.preprocess loops over
 *   the table boxes, table_structure_recognizer.py:169-186; here all crops of a forward share a launch).  blob_dev: int32
 *   words - n records of ymk_pil_batch_record_words() words {page address low, high; page width; x0; y0; ksize_x; ksize_y;
 *   word offsets into the blob of xbounds, xcoefs, ybounds, ycoefs; 0}, followed by the tables; x_dev: fp32 [n][3][oh][ow]. */
int ymk_pil_resize_batch_to_chw(const int* blob_dev, int n, int oh, int ow, float* x_dev, void* stream);
int ymk_pil_batch_record_words(void);
int ymk_crop_batch(const unsigned char* page_dev, int page_h, int page_w, const void* descs_dev, int n, int max_warp_w,
                   int max_warp_h, unsigned char* scratch_dev, float* out_dev, int batch_w, int out_h, void* stream);
/* ymk_crop_batch_levels: the same with `source_downscale` (data/dataset.py:26-41,64-79): descriptor.level picks the
 *   pyramid level a crop is cut from.  level_pages: HOST array of n_levels (<= 4) device pointers, level 0 = the page;
 *   a NULL entry (level not built because no quad uses it) aliases the page.
 * ymk_halve_u8c3: one pyramid step, cv2.resize(img, None, fx=0.5, fy=0.5, INTER_AREA) on uint8 [h][w][3];
 *   dst_h / dst_w = round-half-even(h / 2), (w / 2) as cv2 computes them. */
int ymk_crop_batch_levels(const unsigned char* const* level_pages, const int* level_h, const int* level_w, int n_levels,
                          const void* descs_dev, int n, int max_warp_w, int max_warp_h, unsigned char* scratch_dev,
                          float* out_dev, int batch_w, int out_h, void* stream);
int ymk_halve_u8c3(const unsigned char* src_dev, int h, int w, unsigned char* dst_dev, int dst_h, int dst_w, void* stream);
int ymk_crop_desc_size(void);

/* ---- DB post-processing on the host (replaces DBnetPostProcessor.boxes_from_bitmap,
 * postprocessor/dbnet_postporcessor.py:32-82: threshold, border following, min-area rectangles,
 * polygon-mean score, unclip, scaling to the original page).  prob_host: fp32 [h][w] HOST pointer
 * (the map after D2H); quads_out: int16 [capacity][4][2]; scores_out: double [capacity]. */
int ymk_db_postprocess(const float* prob_host, int h, int w, float thresh, float box_thresh, int min_size,
                       int max_candidates, float unclip_ratio, int dest_w, int dest_h, int16_t* quads_out,
                       double* scores_out, int capacity, int* count);

/* ---- table cell detector post-processing on the host (replaces find_holes_as_rects, table_cell_detector.py:116-143:
 * cv2.rectangle / morphologyEx(OPEN, close_ksize x close_ksize box, 3 iterations) / floodFill / findContours(EXTERNAL) /
 * boundingRect).  cell_boxes: int [n][4] (x1, y1, x2, y2) relative to the h x w table crop; rects_out: int [capacity][4],
 * each hole's bounding rectangle grown by `pad`, holes below min_area dropped; HOST pointers. */
int ymk_table_hole_rects(int h, int w, const int* cell_boxes, int n, int pad, int close_ksize, int min_area, int* rects_out,
                         int capacity, int* count);

/* ---- measurement aid for bench.py (not on the product path): between begin/end every launch of
 * the implicit-GEMM convolution kernel is bracketed by HIP events on its own stream; end returns
 * the summed kernel time, the algorithmic FLOPs (2*M*Cout*KH*KW*Cin, unpadded) and launch count.
 * ymk_prof_bytes: algorithmic HBM bytes of the launches since the last begin (input view + weights + output
 * [+ residual], each counted once) - what the PMC traffic of the same launches is compared with. */
int ymk_prof_begin(void);
/* Test / measurement knobs, process-wide (never touched by the product path; defaults in parentheses):
 *   "splitk_force" (-1)  >= 0: that split-K tile shape for every eligible launch      "no_splitk" (0)  1: conv_igemm only
 *   "conv_variant" (0)   alternative conv_igemm schedules for A/B runs (tools/conv_sweep.py)       "prof_dump" (0)  1: ymk_prof_end
 *   prints one line per launch      "parseq_unfused" (0)  1: per-op PARSeq decoder step at every width
 *   "conv_fast" (27)     bit 0: index shortcut of 1x1 / stride-1 layers, bit 1: residual rows fetched ahead, bit 3: K-tile rows of the
 *                        128 x 64 tile XOR-swizzled instead of padded (48 KB of LDS: three blocks per CU) and that tile for every
 *                        launch with K <= 512, bit 4: accumulators of ragged-Cout launches stored straight from registers;
 *                        bit 2: that direct epilogue for every plain store (A/B runs).  3 = the round-2 kernels.  Every setting
 *                        computes the same bits (tests/test_ops_gpu.py::test_conv_epilogue_variants_are_bit_identical).
 *   "dec_rows" (0)       samples per block of the fused greedy step: 0 = by row count, 1 / 2 / 4 forced (bit-identical results)
 *   "parseq_no_rowmax" (0)  1: the fused greedy loop writes every step's logits and arg-maxes them from memory (round-2 form);
 *                        0: the vocabulary head's epilogue reduces each 64-column tile to (max, column) and no AR logits exist
 *   "rowmax_tile" (0)    the kernel of that head where 128 x 128 tiles fill the chip: 0 = the A-stationary kernel, one column block
 *                        per block (K <= 192; else the 128 x 64 tile), 1 / 2 / 3 = 128 x 128 x 16 waves / 128 x 128 x 8 waves / 256 x 128,
 *                        4 = the A-stationary kernel with dealt column groups, 6 = 128 x 64 always (rounds 3-5) - the same pair
 *                        table, bit for bit, from each (tests/test_routes_gpu.py)
 * GELU: every kernel of the library (the exact-fp32 mode "conv_split" = 0 included, and the greedy decoder step) evaluates
 *   0.5 v (1 + erf(v / sqrt 2)) through one branch-free function (csrc/ymk_common.h gelu_f32: a rational erfc form on v_rcp_f32 /
 *   v_exp_f32) since round 5: absolute error < 5e-7 over the whole line (tests/test_ops_gpu.py), the size of the rounding of
 *   v x a libm-grade erff; the RELATIVE error of the negative tail (|GELU(v)| < 1e-6 beyond v = -5) is not bounded.  Results of
 *   "conv_split" = 0 are therefore not bit-comparable with rounds 1-4, whose kernels called erff.
 *   "conv_split" (-1)    process-wide operand precision of every model whose own "conv_split" parameter is unset, and of
 *                        ymk_op_conv2d: 0 = exact fp32 MFMA, 16 = two fp16 planes, 2 / 3 = bf16 planes (ymk_model_set_param above);
 *                        -1 = unset: models run their default (16), ymk_op_conv2d exact fp32.  Also set for a whole process by
 *                        the environment variable YMK_CONV_SPLIT (yomitoku_amd/_lib.py).
 *   "conv_split_tile" (0) tile shape of that path for A/B runs: 0 = by format, 1 = 128 x 64, 2 = 256 x 128 (16 waves),
 *                        3 = 128 x 128 (16 waves), 4 = 128 x 128 (8 waves), 11 = 256 x 256 (16 waves; two planes);
 *                        5-10 / 12-14 (bf16 only): 64-k stages, loads two stages ahead, stores threaded through the MFMAs
 *   "gemm_row_limit" (0) > 0: linear layers cut their rows into chunks of at most this many (rounded down to 1024s) - the chunking
 *                        that keeps an A operand view below the 4 GiB a buffer descriptor addresses, forced at test sizes
 *   "astat" (1)          1: chip-filling pointwise layers with K <= 192 and Cout > 64 run on the A-stationary kernel (K = 256 stays on
 *                        the register-staged kernel, where it measured ahead); 0: the round-4 routing (A/B runs).
 *                        "conv_split_tile" 30 forces that kernel for every launch it can run, K <= 256 (tests)
 *   "ar_publish" (1)     how a greedy step of the recogniser tells the host whether rows are still open: 1 = a one-thread launch
 *                        behind the greedy kernel writes the mapped host word (rows store a flag), 0 = the kernel's
 *                        last-arriving block does (rows count: rounds 1-5; A/B runs, tools/stress_call.py)
 *   "parseq_no_ln_fusion" (0)  1: the ViT blocks' LayerNorms as launches of their own (A/B runs, tests); 0: folded into the
 *                        operand load of the q|k|v and fc1 GEMMs wherever the A-stationary kernel takes them
 *   "act_planes" (1)     1: the tensor between a bottleneck's 1 x 1 reduction and its 3 x 3 convolution lives in HBM as the two fp16
 *                        planes the 3 x 3 multiplies with (written by the reduction's epilogue under the bound
 *                        max|x_in| x max row L1 norm + max|bias|, read by LDS-DMA without conversion) wherever both launches
 *                        fill the chip; 0: fp32 activations everywhere (A/B runs)
 *   "parseq_no_mlp_fusion" (0)  1: norm2 -> fc1 -> GELU -> fc2 of the ViT blocks as launches of their own through the [rows][4 D]
 *                        buffer (A/B runs, tests); 0: one launch with the hidden state on chip where the fused kernel runs it
 *   "amax_check" (0)     1: every fp16-split launch whose input came with a max|x| record from its producer ALSO measures the
 *                        input and compares (ymk_amax_check_counters) - the self-check of the record plumbing; 2: also names every
 *                        launch whose record lies BELOW the measured maximum on stderr (serialises the stream) */
int ymk_debug_option(const char* key, int value);
/* Tracing: with YMK_ROCTX=1 in the environment every forward (ymk_dbnet_forward, ymk_parseq_forward[_groups], ymk_rtdetr_forward),
 * ymk_parseq_token_stats, ymk_model_finalize and ymk_model_reserve runs inside a roctx range of its own name
 * (`rocprofv3 --marker-trace --kernel-trace`: which call a kernel belongs to).  The marker library is opened at run time
 * (librocprofiler-sdk-roctx.so, else libroctx64.so); without it, or without the variable, no range is pushed. */
/* Launch counters since the process started, for tests that must know a route was really taken: "astat_launches" (the
 * A-stationary short-K kernel), "ln_fused_launches" (those of them that carried a LayerNorm in their operand load),
 * "planes_written_launches" / "planes_read_launches" (convolutions whose output / input lives in HBM as fp16 planes),
 * "mlp_fused_launches" (ViT MLP halves run as one launch), "rowmax_wide_launches" (row-max heads on anything but the 128 x 64 tile).
 * And what a forward must NOT do (round 6): a ymk_*_forward whose workspace the caller sized first (ymk_model_reserve) never
 * allocates or frees device / pinned memory, never builds a weight copy and never waits for a stream - the split weight
 * copies and the max|x| words of the precision a model runs are built by ymk_model_finalize (and by ymk_model_set_param
 * when "conv_split" changes afterwards).  The fallbacks remain for callers of the bare ABI and are counted:
 * "allocs_in_forward" (hipMalloc / hipFree / hipHostMalloc / hipHostFree issued inside a forward), "arena_grows_in_forward"
 * (of those: the workspace grown because no reservation covered the shape), "lazy_panel_builds" (weight copies built on
 * first use: the process-wide precision switched after finalize), "syncs_in_forward" (stream waits those fallbacks made).
 * tests/test_serving_gpu.py holds all four at zero over both entry points of the analyzer. */
int ymk_stat(const char* key, int64_t* value);
/* out4 = {launches checked, records below the true max|x| (a bug), records more than 2^8 above it, largest record / truth
 * exponent distance} since the process started; synchronises the device. */
int ymk_amax_check_counters(int64_t* out4);
int ymk_prof_end(double* conv_ms, double* conv_flop, int64_t* conv_launches);
int ymk_prof_bytes(double* conv_bytes);
/* The launches of the span ymk_prof_end closed last, one row each in launch order: kernel time (ms), algorithmic FLOPs,
 * algorithmic HBM bytes (as above) and the MFMA products the kernel spends per fp32-grade product (0: exact fp32 MFMA,
 * 3: two fp16 or two bf16 planes, 6: three bf16 planes).  *count = how many there are; rows beyond `capacity` are not
 * written (capacity 0 with null arrays asks for the count).  What bench.py prices launch by launch against the closer
 * of the two roofs. */
int ymk_prof_launch_table(double* ms, double* flop, double* bytes, double* mfma_products, int64_t capacity, int64_t* count);

/* ---- single operators (exported for the parity tests; same kernels the models use) -----
 * NHWC fp32 tensors; weight in PyTorch OIHW order on the host. */
int ymk_op_conv2d(const float* x_dev, int n, int h, int w, int c, /* c % 4 == 0, or c == 4 with tap4 */
                  const float* w_host_oihw, int cout, int cin, int kh, int kw, const float* scale_host,
                  const float* bias_host, const float* res_dev, int stride, int pad, int dil, int act, int tap4,
                  float* y_dev, void* stream);
/* rows x d LayerNorm (eps as given); attention over contiguous [b][l][heads*hd] q/k/v with optional
 * boolean masks (non-zero = blocked): mask_qk [lq][lk], kpm [b][lk]; use_small selects the masked
 * small-query kernel even without masks (otherwise the fp32-MFMA flash kernel runs). */
/* A 1 x 1 layer with K = c <= 256 through the A-stationary fp16-split kernel (yomitoku_amd/csrc/ymk_conv_astat.hip; the models'
 * short-K pointwise layers run on it since round 5) at ANY row count, where the library's own routing
 * only takes chip-filling launches: y[m][cout] = act(scale * (x'[m][c] . w^T) + bias + res), x' = x - or, with ln_g_host /
 * ln_b_host ([c], host), LayerNorm(x; gamma, beta, ln_eps) folded into the operand load (c = 128 or 192; the input's
 * max|x| record is then the LayerNorm's static output bound, as in the models).  w_host_oc: [cout][c] on the host, x / res / y
 * on the device.  Same arithmetic as the register-staged fp16-split kernel (ymk_op_conv2d under "conv_split" 16 and
 * "conv_split_tile" 3): compared bit for bit in tests/test_conv_astat_gpu.py.  The launch is repeated `reps` times;
 * *kernel_ms = HIP-event time of the last one.  Refuses (error) what the kernel cannot run. */
int ymk_op_conv1x1_astat(const float* x_dev, int m, int c, const float* w_host_oc, int cout, const float* scale_host,
                         const float* bias_host, const float* res_dev, int act, const float* ln_g_host, const float* ln_b_host,
                         float ln_eps, float* y_dev, int reps, float* kernel_ms, void* stream);

/* The MLP half of a ViT block in one launch (yomitoku_amd/csrc/ymk_vit_mlp.hip; what the PARSeq encoder runs per block when a
 * forward fills the chip): y[m][d] = x + fc2(GELU(fc1(LayerNorm(x; gamma, beta, eps)))), fc1 [f][d] + bias [f], fc2 [d][f] + bias
 * [d] (host, nn.Linear layout), x / y on the device.  d = 192, f = 768, m >= 32 641 (256 blocks of 128 rows) - refused (error)
 * otherwise.  Repeated `reps` times IN PLACE on y (timing); *kernel_ms = HIP-event time of the last launch. */
int ymk_op_vit_mlp(const float* x_dev, int m, int d, int f, const float* ln_g_host, const float* ln_b_host, float ln_eps,
                   const float* w1_host_fd, const float* b1_host, const float* w2_host_df, const float* b2_host, float* y_dev, int reps,
                   float* kernel_ms, void* stream);

int ymk_op_layernorm(const float* x_dev, int rows, int d, const float* g_dev, const float* b_dev, float eps,
                     float* y_dev, void* stream);
int ymk_op_attention(const float* q_dev, const float* k_dev, const float* v_dev, float* o_dev, int b, int heads, int lq,
                     int lk, int hd, float scale, const unsigned char* mask_qk_dev, const unsigned char* kpm_dev,
                     int use_small, void* stream);
int ymk_op_maxpool3x3s2(const float* x_dev, int n, int h, int w, int c, float* y_dev, void* stream);
int ymk_op_upsample_bilinear(const float* x_dev, int n, int h, int w, int c, int oh, int ow, const float* add_dev,
                             float* y_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* YMK_H */
