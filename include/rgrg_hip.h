/* rgrg_hip.h - C ABI of librgrg_hip.so: the MI355X (gfx950) kernels behind the RGRG
 * inference hot path (ttanida/rgrg ReportGenerationModel.generate).
 *
 * The reference has no FFI of its own: its hot path sits behind a Python nn.Module
 * API and delegates the arithmetic to torch / torchvision /
 * transformers ops.  Each entry point below replaces the op (or op group) the
 * reference reaches at the cited file:line.  All pointers are DEVICE pointers
 * (fp32 unless stated), all tensors are dense row-major with the layout written
 * next to them, `stream` is a hipStream_t passed as void*, every function returns
 * 0 on success (RGRG_OK) or a negative RGRG_E* code and never allocates or
 * synchronises unless its comment says so.  Activations are NHWC ("channels
 * last"); weight repacking into the layouts named here is done once at load time
 * by the host side (rgrg_amd/engine.py: HipEngine.__init__ and its _pack_* helpers).
 */
#ifndef RGRG_HIP_H_
#define RGRG_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RGRG_OK 0
#define RGRG_EINVAL (-1) /* bad shape / argument */
#define RGRG_EHIP (-2)   /* a HIP call failed; see rgrg_last_error() */
#define RGRG_ESTATE (-3) /* handle used in the wrong state */

/* activation codes for the GEMM epilogue */
#define RGRG_ACT_NONE 0
#define RGRG_ACT_RELU 1
#define RGRG_ACT_GELU_NEW 2 /* 0.5x(1+tanh(sqrt(2/pi)(x+0.044715x^3))) - HF NewGELUActivation */

const char* rgrg_last_error(void);
int rgrg_abi_version(void);
/* gfx arch name of device `dev` into buf (e.g. "gfx950"); RGRG_EHIP if no device. */
int rgrg_device_arch(int dev, char* buf, int buflen);

/* ---------------------------------------------------------------------------------
 * Dense / implicit-GEMM fp32 contraction on the f32 MFMA (v_mfma_f32_32x32x2_f32).
 *   Y[m,n] = act( (sum_k A(m,k) * W[n,k]) * scale[n] + shift[n] + R[m,n] )
 * scale may be NULL (=1), shift may be NULL (=0; it carries either the bias or the
 * folded BatchNorm shift), R may be NULL.  K % 32 == 0.  `ws` is only needed when
 * splitk > 1 (splitk*M*N floats).
 *
 * rgrg_linear_f32: A is [M,K] row-major (lda = K), W is [N,K] (nn.Linear layout).
 *   replaces F.linear / torch.addmm at: box_head fc6/fc7, box_predictor
 *   (src/object_detector/custom_roi_heads.py:235-236), dim_reduction (:264),
 *   BinaryClassifierRegionSelection.classifier
 *   (src/binary_classifier/binary_classifier_region_selection.py:32), and - when
 *   more than 32 sequences decode together - Conv1DWithTrainedWeights.forward
 *   (src/language_model/language_model.py:25-29), GPT2MLP, uk/uv (:142-143),
 *   feature_space_transformation_nn (:284) and lm_head (:366).
 * rgrg_conv2d_nhwc_f32: A is the im2col view of X[B,H,W,Cin]; W is
 *   [Cout][KH][KW][Cin]; Y is [B,OH,OW,Cout]; Cin % 32 == 0.
 *   replaces nn.Conv2d + eval BatchNorm2d + ReLU (+ residual add) of the ResNet-50
 *   trunk (src/object_detector/object_detector.py:51-62,219) and the RPNHead convs
 *   (src/object_detector/custom_rpn.py:61).
 * --------------------------------------------------------------------------------- */
int rgrg_linear_f32(const float* A, const float* W, const float* scale, const float* shift, const float* R,
                    float* Y, int M, int N, int K, int ldy, int act, int splitk, float* ws, void* stream);
int rgrg_conv2d_nhwc_f32(const float* X, const float* W, const float* scale, const float* shift, const float* R,
                         float* Y, int B, int H, int Wd, int Cin, int Cout, int KH, int KW, int stride, int pad,
                         int act, int splitk, float* ws, void* stream);

/* ResNet stem: Conv2d(1,64,7,stride 2,pad 3,bias=False) + eval BN + ReLU
 * (object_detector.py:54), X [B,H,W] (1 channel), Wt [49][64] (tap-major), Y [B,H/2,W/2,64]. */
int rgrg_stem_conv7x7_f32(const float* X, const float* Wt, const float* scale, const float* shift, float* Y,
                          int B, int H, int Wd, void* stream);
/* nn.MaxPool2d(3, stride 2, pad 1) on NHWC (resnet50 child 3).  C % 4 == 0. */
int rgrg_maxpool3x3s2_nhwc_f32(const float* X, float* Y, int B, int H, int Wd, int C, void* stream);

/* ---------------------------------------------------------------------------------
 * RPN proposal generation for a single 16x16 feature level (custom_rpn.py:62-71 +
 * torchvision 0.13.1 AnchorGenerator / BoxCoder(1,1,1,1).decode / filter_proposals):
 * per image top-`pre_nms` of the A*H*W objectness logits (ties: lower index first),
 * sigmoid, decode, clip to [0,img_w]x[0,img_h], drop w/h < min_size, greedy NMS
 * (IoU > nms_thresh), first `post_nms` survivors in score order.
 *   head_out [B, HW, 5*A]: per location A objectness logits then A*4 deltas (a*4+c)
 *   anchors  [HW*A, 4] (x1,y1,x2,y2), index = loc*A + a
 *   proposals [B, post_nms, 4] (rows >= count are zero), counts int32 [B],
 *   offsets int32 [B+1] = exclusive prefix sum of counts.
 * pre_nms <= 1000, post_nms <= pre_nms, HW*A <= 65536.
 * --------------------------------------------------------------------------------- */
int rgrg_rpn_proposals_f32(const float* head_out, const float* anchors, float* proposals, int32_t* counts,
                           int32_t* offsets, int B, int HW, int A, int pre_nms, int post_nms, float nms_thresh,
                           float min_size, float img_w, float img_h, void* stream);

/* torchvision.ops.roi_align(aligned=False, sampling_ratio=2, output 8x8) fused with
 * AvgPool2d(8) (custom_roi_heads.py:232,253): feat NHWC [B,FH,FW,C] (C % 128 == 0,
 * FH*FW*128*4 bytes of LDS <= 128 KiB), RoIs = rows [offsets[b], offsets[b+1]) of
 * image b taken from proposals[b].  out [R, 64, C] (bin-major, channel-minor: the
 * fc6 weight is repacked to match), pooled [R, C].  R_total = offsets[B] (host copy). */
int rgrg_roi_align_avgpool_f32(const float* feat, const float* proposals, const int32_t* offsets, float* out,
                               float* pooled, int B, int FH, int FW, int C, int max_props, int R_total,
                               float spatial_scale, void* stream);
/* The same with the [R, 64, C] maps stored as 16-bit values (fp16 = 0: bf16, 1: float16; round to nearest even; pooled stays
 * f32 from unrounded values): what the box head consumes under torch.autocast (generate_reports_for_images.py:108) - fc6
 * then runs on rgrg_linear_bf16_f32 with half the A bytes. */
int rgrg_roi_align_avgpool_bf16maps(const float* feat, const float* proposals, const int32_t* offsets, uint16_t* out16,
                                    float* pooled, int B, int FH, int FW, int C, int max_props, int R_total,
                                    float spatial_scale, int fp16, void* stream);

/* CustomRoIHeads.get_top_region_features_detections_class_detected
 * (custom_roi_heads.py:63-208), eval: pred [R, ldp] holds 30 class logits then 120
 * box deltas per RoI; outputs class_detected uint8 [B,29], top_scores [B,29],
 * top_boxes [B,29,4], top_feats [B,29,C] (gathered rows of pooled). */
int rgrg_top1_per_class_f32(const float* pred, int ldp, const float* proposals, const int32_t* offsets,
                            const float* pooled, uint8_t* class_detected, float* top_scores, float* top_boxes,
                            float* top_feats, int B, int C, int max_props, float img_w, float img_h, void* stream);

/* Replaces nn.BCEWithLogitsLoss(pos_weight) on `logits[class_detected]` of the two region classifiers in
 * forward() (src/binary_classifier/binary_classifier_region_selection.py:22,40-44 with pos_weight 2.2;
 * binary_classifier_region_abnormal.py:29,43-47 with 6.0): mean over the rows with mask != 0 of
 * (1-y) x - (1 + (w-1) y) log_sigmoid(x).  logits f32 [n], mask/target u8 [n], loss: one f32 (nan when no row). */
int rgrg_bce_with_logits_masked_f32(const float* logits, const uint8_t* mask, const uint8_t* target, float pos_weight,
                                    int n, float* loss, void* stream);
/* Replaces the albumentations pipeline of get_image_tensor (src/full_model/generate_reports_for_images.py:129-147;
 * SURVEY 8(f) rank 4) for a decoded 8-bit gray image already in device memory: LongestMaxSize(512, cv2.INTER_AREA)
 * [the caller passes new_h/new_w = py3round(dim * 512 / max(h, w))] -> centred zero PadIfNeeded(512, 512) ->
 * Normalize(mean, std; max_pixel_value 255) -> dst f32 [512*512] (= the [1,1,512,512] tensor).  Shrinking uses
 * OpenCV's INTER_AREA tables / integer fast paths; an image smaller than 512 px is ENLARGED the way OpenCV does it
 * for INTER_AREA with an enlarged axis (its 8-bit fixed-point bilinear path with the area coordinate rule).
 * OpenCV / albumentations are third-party and absent here: arithmetic restated from their published sources,
 * parity unpinned (oracle/preprocess.py). */
int rgrg_preprocess_u8_f32(const uint8_t* src, int h, int w, int src_stride, int new_h, int new_w, float mean, float std,
                           float* dst, void* stream);
/* BinaryClassifierRegionSelection threshold + mask + row-major compaction
 * (binary_classifier_region_selection.py:53-61): selected = (logit > thr) & detected;
 * sel_rows int32 [n] lists the selected flat (image*29+region) indices in order,
 * *n_selected (device int32) their number; feats_out [S,D] = feats[sel_rows]. */
int rgrg_select_regions_f32(const float* logits, const uint8_t* class_detected, float thr, uint8_t* selected,
                            int32_t* sel_rows, int32_t* n_selected, int n, void* stream);
int rgrg_gather_rows_f32(const float* src, const int32_t* rows, float* dst, int n_rows, int D, void* stream);

/* ---------------------------------------------------------------------------------
 * Greedy decoder (LanguageModel.generate num_beams=1 -> greedy_search,
 * src/language_model/language_model.py:401-447,609-652, with forward :258-366 and
 * GPT2PseudoAttention :124-180).  The decoder object keeps pointers to the (caller
 * owned, device resident) weights, owns its KV cache / activations workspace and a
 * captured hipGraph of one decode step per sequence count.
 * --------------------------------------------------------------------------------- */
typedef struct rgrg_decoder_layer_weights {
    const float *ln1_g, *ln1_b, *ln2_g, *ln2_b;
    const float *c_attn_w, *c_attn_b;     /* [3072,1024] (transposed Conv1D), [3072] */
    const float *attn_proj_w, *attn_proj_b; /* [1024,1024], [1024] */
    const float *c_fc_w, *c_fc_b;         /* [4096,1024], [4096] */
    const float *mlp_proj_w, *mlp_proj_b; /* [1024,4096], [1024] */
} rgrg_decoder_layer_weights;

typedef struct rgrg_decoder_weights {
    int n_layer, d_model, n_head, vocab; /* 24, 1024, 16, 50257 */
    const float* wte;                    /* [vocab,1024]; also the tied lm_head */
    const float *lnf_g, *lnf_b;
    const float *fst0_w, *fst0_b, *fst2_w, *fst2_b; /* feature_space_transformation_nn */
    const float *ukv_w, *ukv_b; /* [n_layer*2*1024, 1024] rows = (layer, uk|uv, out), [n_layer*2*1024] */
    const rgrg_decoder_layer_weights* layers; /* host array of n_layer entries */
} rgrg_decoder_weights;

typedef struct rgrg_decoder rgrg_decoder;

/* Allocates device memory (weights repacked into MFMA-fragment tiles for the
 * <=32-sequence path - LayerNorm gains folded in -, workspace for max_seqs sequences x max_len
 * positions).  SYNCHRONISES THE DEVICE on entry and before returning: the weights are packed on the
 * decoder's own stream and this entry has no stream argument, so it first waits for whatever is still
 * producing them on the caller's streams.  Not on the hot loop.   max_seqs < 65536 (16-bit row tickets of the arg-max bookkeeping). */
int rgrg_decoder_create(const rgrg_decoder_weights* w, int max_seqs, int max_len, rgrg_decoder** out);
/* The same with the K/V cache in CALLER-owned memory: kv_cache = device buffer of rgrg_decoder_kv_cache_bytes(n_layer,
 * max_seqs, max_len) bytes, zero-filled by the caller, which must outlive the decoder; rgrg_decoder_destroy does not free it.
 * The host mirror allocates it as a torch tensor, so that the `presents` of LanguageModel.forward(use_cache=True)
 * (language_model.py:396-399) are ordinary views whose lifetime torch manages - they stay valid when the decoder is replaced. */
int rgrg_decoder_create_with_cache(const rgrg_decoder_weights* w, int max_seqs, int max_len, void* kv_cache, size_t kv_cache_bytes,
                                   rgrg_decoder** out);
/* Bytes of the cache [n_layer][K | V][max_seqs][16 heads][max_len + 1 slots][64] in fp32. */
size_t rgrg_decoder_kv_cache_bytes(int n_layer, int max_seqs, int max_len);
void rgrg_decoder_destroy(rgrg_decoder* d);
/* feats [S,1024] (selected region features); out_ids int64 [S,max_length] is filled
 * with the leading BOS, the generated ids, and PAD (50256) after a row finished;
 * *out_len (host) = L' = 1 + number of forward passes the reference would have run
 * (stops when every row has emitted EOS or at max_length).  max_length <= 0 means
 * "until all rows finished" bounded by the decoder's max_len.  Synchronises `stream`
 * before returning.  use_graph=0 launches the kernels eagerly (debug/profiling). */
int rgrg_decoder_generate(rgrg_decoder* d, const float* feats, int S, int max_length, int64_t* out_ids,
                          int out_ld, int* out_len, int use_graph, void* stream);
/* Beam search (LanguageModel.generate num_beams > 1 -> beam_search, language_model.py:450-475,
 * :529-607, with transformers 4.19.2 BeamSearchScorer semantics; 1 <= num_return_sequences <= num_beams).
 * The decoder must have been created with max_seqs >= S*num_beams.  Any num_beams, as in the reference (up to 16 the row / item
 * rankings keep per-thread candidate lists in registers; wider beams rank in 2*num_beams rounds over the row - same winners).
 * Device side: one decode step over the S*num_beams beam rows (KV cache never re-ordered:
 * per-slot ancestor table), per-row log-sum-exp + top-2*num_beams, per-item merge.  Host
 * side (one small D2H/H2D per step, like the reference's scorer): hypothesis bookkeeping.
 * out_ids int64 [S * num_return_sequences, out_ld] receives the best hypotheses of every item (best first),
 * *out_len their padded length. */
int rgrg_decoder_beam_search(rgrg_decoder* d, const float* feats, int S, int num_beams, int max_length,
                             int early_stopping, float length_penalty, int num_return_sequences, int64_t* out_ids,
                             int out_ld, int* out_len, void* stream);
/* Opt-in reduced precision for MANY sequences (BASELINE configs[2], batch 32): mode 1 (bfloat16) / 2 (float16) makes the
 * > 128-row paths run their projections on the 16-bit MFMA (16-bit weights and GEMM inputs, fp32 accumulate / LayerNorm /
 * softmax / residual) and keep the decode K/V cache in that type (what the reference's torch.autocast does to `present`);
 * 0 = fp32.  NOT bit-exact with the fp32 reference; the <= 128-sequence decode path (launch-latency bound) always stays
 * fp32.  May allocate and synchronise (a change of the 16-bit type re-converts the weight copies). */
int rgrg_decoder_set_precision(rgrg_decoder* d, int mode);
/* Replaces LanguageModel.forward(input_ids, attention_mask, image_hidden_states, return_loss, use_cache=False)
 * in eval mode (src/language_model/language_model.py:258-399): one teacher-forced pass over
 * T tokens per sequence - feature_space_transformation_nn, wte[ids] + wte[arange(T)] (:298-307), 24 blocks of
 * pseudo self-attention without cache (image key/value first, future columns -1e4, additive padding mask
 * (1 - [1|attention_mask]) * -10000, :84-160,:325-334), final LayerNorm, lm_head.
 *   feats [S,1024] f32, input_ids [S,T] int64 (every id in [0, vocab)), attention_mask [S,T] f32 or NULL (= ones),
 *   S <= max_seqs of the decoder, T <= 1023 (T + 1 keys <= GPT-2's 1024 positions; beyond 256 keys the attention
 *   recomputes its score tiles instead of holding them in registers).
 *   logits_out: NULL or f32 [S,T,vocab] (return_loss=False);
 *   loss_out:   NULL or one f32 = CrossEntropyLoss(ignore_index=-100) of logits[:, :-1] against input_ids[:, 1:]
 *               with the labels of attention_mask == 0 positions ignored (:368-396); nan when no label is scored.
 * Runs on the decoder's stream between two event edges with `stream`; does not synchronise the host.  Work space
 * for S*T token rows is grown on demand (first call of a larger size allocates). */
/* position_ids of the teacher-forced passes (src/language_model/language_model.py:293-307 embeds whatever it is given; the
 * default is arange(T)): int64 device array of S * T entries (per sentence) or T entries (one row, broadcast over the
 * sentences), consumed by the NEXT rgrg_decoder_lm_forward / rgrg_decoder_lm_loss_grad call; NULL restores the default.  Like the
 * reference's, they index the TOKEN table (wte[position_ids], :307) and are range-checked on the device with the token ids. */
int rgrg_decoder_set_lm_positions(rgrg_decoder* d, const int64_t* pos, int64_t n);
int rgrg_decoder_lm_forward(rgrg_decoder* d, const float* feats, const int64_t* input_ids, const float* attention_mask,
                            int S, int T, float* logits_out, float* loss_out, void* stream);
/* Replaces `language_model_loss.backward()` of the training loop (src/full_model/train_full_model.py:172-208) for the
 * decoder: the same teacher-forced pass as rgrg_decoder_lm_forward, keeping the activations, followed by the backward
 * pass.  Only what the reference trains in the language model receives a gradient: uk / uv of every
 * GPT2PseudoAttention (src/language_model/language_model.py:50-57) and feature_space_transformation_nn (:230-236);
 * every other GPT-2 tensor is frozen there (:207-213), so the 24 blocks only propagate activation gradients
 * (dX = dY W on transposed weight copies made at the first call).
 *   loss_scale    d(total_loss)/d(language_model_loss), e.g. the loss weight (and an AMP scale)
 *   dropout_p     0 = deterministic pass.  > 0: GPT-2's four dropout sites of train mode (embedding `drop` :311,
 *                 attn_dropout on the probabilities :116, resid_dropout :178, the MLP's dropout) with a counter-based
 *                 generator: mask = f(dropout_seed, layer*4 + site, element index) (Philox4x32-7: counter = index / 4,
 *                 word index % 4), recomputed by the backward pass; torch's generator stream cannot be reproduced, so
 *                 the masks differ from the reference's (rgrg_dropout_mask_f32 exports them for checks).
 *   dropout_seed  a fresh value per call
 *   loss_out      one f32, the unscaled loss
 *   grad_ukv_w    f32 [n_layer*2*1024, 1024]  rows = [uk_0; uv_0; uk_1; uv_1; ...]      grad_ukv_b  f32 [n_layer*2*1024]
 *   grad_fst0_w/b, grad_fst2_w/b   f32 [1024,1024] / [1024]  (Linear 0 and 2 of feature_space_transformation_nn)
 * Gradients are WRITTEN (not accumulated).  T <= 1023 (the reference's own limit).  Allocates work space on demand (~0.9 MB per token row). */
int rgrg_decoder_lm_loss_grad(rgrg_decoder* d, const float* feats, const int64_t* input_ids, const float* attention_mask,
                              int S, int T, float loss_scale, float dropout_p, uint64_t dropout_seed, float* loss_out,
                              float* grad_ukv_w, float* grad_ukv_b, float* grad_fst0_w, float* grad_fst0_b, float* grad_fst2_w,
                              float* grad_fst2_b, void* stream);
/* Measurement hook of bench.py's BASELINE configs[4] leg: the frozen-weight GEMMs of ONE training step of the 16-bit
 * (torch.autocast) flow - forward and activation-gradient GEMMs of the 24 blocks, lm_head forward / dgrad - launched back to
 * back on the decoder's stream between two HIP events, on the work space of the last rgrg_decoder_lm_loss_grad call of this
 * shape (overwritten: timing only).  flops_per_step = 2 M N K of those launches. */
int rgrg_decoder_time_train_gemms(rgrg_decoder* d, int S, int T, int iters, float* ms_per_step, double* flops_per_step,
                                  int* launches_per_step);
/* The dropout mask of one site as the training pass computes it: out[i] = 0 (dropped, probability p) or 1/(1-p), for the
 * flat element index i of the site's tensor ([S*T,1024] for sites 0/2/3, [S,16,T,T+1] for the attention probabilities);
 * stream_id = layer*4 + site; row_len = T + 1 for the attention probabilities (their generator index pads a row of keys to
 * a multiple of 4), 0 otherwise. */
int rgrg_dropout_mask_f32(uint64_t seed, uint32_t stream_id, float p, int64_t n, int row_len, float* out, void* stream);
/* Replaces the INCREMENTAL form of LanguageModel.forward - use_cache=True with or without past_key_values
 * (src/language_model/language_model.py:258-366, :396-399), the call the reference's own generate loop makes every step -
 * over this decoder's pre-allocated K/V cache instead of concatenated tensors.  past_len = number of tokens already in the
 * cache (0: feats [S,1024] must be given, the image key / value goes to slot 0, :135-157; > 0: feats must be NULL).  The T
 * tokens input_ids [S,T] (int64) go to cache slots past_len + 1 .. past_len + T; their embedding is wte[token] +
 * wte[position] with position = position_ids[s][j] (int64 [S,T] on the device, any values in [0, vocab): :293-307) or, when
 * position_ids is NULL, past_len + j (what prepare_inputs_for_generation passes, :498-520); each appends its key / value;
 * logits_out f32 [S,T,vocab] receives lm_logits of every fed position.  past_len + T <= the decoder's max_len.
 * attention_mask: NULL (all ones: generation) or f32 [S][past_len + T] over ALL token keys so far - the reference adds
 * (1 - mask) * -1e4 to the scores of a masked key for every query, the image key is never masked (:316-334).  Runs on the
 * decoder's stream between two event edges with `stream`. */
int rgrg_decoder_forward_cached(rgrg_decoder* d, const float* feats, const int64_t* input_ids, const int64_t* position_ids,
                                const float* attention_mask,
                                int S, int T, int past_len, float* logits_out, void* stream);
/* Device address and geometry of one cache plane (layer, kv = 0 key / 1 value): f32 [max_seqs][16][slots][64]; the host
 * wraps rows [:S], slots [:1 + tokens] as the `presents` views of forward(use_cache=True) - no copy. */
int rgrg_decoder_cache_plane(rgrg_decoder* d, int layer, int kv, void** ptr, int* max_seqs, int* slots, int* is_bf16);
/* Token ids of the two teacher-forced entries above are validated ON THE DEVICE (no host round trip per call): an id
 * outside [0, vocab) is clamped for every load (embedding row, cross-entropy label), the loss AND the gradients of that
 * pass come out as NaN, and the NEXT decoder call that finds the (asynchronously mirrored) error word set fails with
 * "index out of range in self" (torch.nn.Embedding's IndexError in the Python layer).  This entry synchronises the
 * decoder's stream and hands the pending error over (*pending = 1, cleared) - the host calls it before it destroys a
 * decoder so that the report is not lost with the object. */
int rgrg_decoder_take_id_error(rgrg_decoder* d, int* pending);
/* After an optimizer step changed the trainable decoder weights IN PLACE (the fst0/fst2/ukv pointers given to
 * rgrg_decoder_create): rebuild the kernel-side copies derived from them (packed skinny layouts, transposes). */
int rgrg_decoder_refresh_trainable(rgrg_decoder* d, void* stream);
/* Building blocks of the two region classifiers' backward pass (the autograd of
 * src/binary_classifier/binary_classifier_region_selection.py:32-44 and binary_classifier_region_abnormal.py:32-47
 * inside train_full_model.py:208 `backward()`); the GEMMs are rgrg_linear_f32 on transposed operands.
 *   rgrg_transpose_pad_f32: dst[c][r] = src[r][c], r < rows; 0 for rows <= r < rows_padded  (dst [cols, rows_padded])
 *   rgrg_colsum_f32:        out[c] = sum_r src[r][c]                                      (bias gradients)
 *   rgrg_relu_backward_f32: d[i] = 0 where h[i] <= 0                                      (h = post-ReLU activation)
 *   rgrg_bce_with_logits_masked_backward_f32: d(mean BCEWithLogits(pos_weight) over mask != 0)/d logits * scale,
 *                           written to dlogits[i*ld] (0 on unmasked rows) */
int rgrg_transpose_pad_f32(const float* src, float* dst, int rows, int cols, int rows_padded, void* stream);
int rgrg_colsum_f32(const float* src, float* out, int rows, int cols, void* stream);
int rgrg_relu_backward_f32(float* d, const float* h, int64_t n, void* stream);
int rgrg_bce_with_logits_masked_backward_f32(const float* logits, const uint8_t* mask, const uint8_t* target, float pos_weight,
                                             int n, float scale, float* dlogits, int ld, void* stream);
/* Replaces torch.optim.AdamW.step() for one parameter tensor (train_full_model.py:409, decoupled weight decay):
 *   p *= 1 - lr*wd; m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
 * with g = grad * grad_scale (1/AMP-scale, 1/accumulation steps).  All arrays f32 [n]; step t >= 1. */
int rgrg_adamw_step_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                        float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream);
/* The same update for n_items tensors in ceil(n_items / 64) launches (hyper-parameters and step shared): `items` = a HOST array
 * of records of five 64-bit words {param, grad, exp_avg, exp_avg_sq (device pointers to f32), element count}; the records
 * travel in the kernel arguments. */
int rgrg_adamw_multi_step_f32(const void* items, int n_items, float lr, float beta1, float beta2, float eps,
                              float weight_decay, int step, float grad_scale, void* stream);
/* The 16-bit entry points below (names say bf16 for history) take `fp16`: 0 = bfloat16, 1 = IEEE float16 - the dtype of the
 * caller's torch.autocast (the reference's scripts use float16: generate_reports_for_images.py:108, train_full_model.py:172).
 * Storage is uint16_t bits either way; the matrix core runs v_mfma_f32_32x32x16_bf16 / v_mfma_f32_32x32x16_f16 (same rate),
 * fp32 accumulate; conversions round to nearest even, fp16 overflows to inf beyond 65504 as torch's does.
 * fp32 -> 16 bit, and Y = act(r16(A) Wb^T + shift + R) (A fp32 [M,K], Wb 16-bit [N,K], K % 64 == 0). */
int rgrg_f32_to_bf16(const float* src, uint16_t* dst, int64_t n, int fp16, void* stream);
int rgrg_bf16_to_f32(const uint16_t* src, float* dst, int64_t n, int fp16, void* stream);   /* exact widening */
/* Replaces nn.Conv2d + eval BatchNorm2d (+ residual add) + ReLU of the ResNet-50 bottlenecks and the RPNHead convs
 * (src/object_detector/object_detector.py:51-62,219; custom_rpn.py:61) WHEN THE CALLER RUNS UNDER torch.autocast
 * (generate_reports_for_images.py:108: the reference's detector then computes in half precision): implicit GEMM on
 * v_mfma_f32_32x32x16_bf16, fp32 accumulation and epilogue.
 *   X16 [B,H,W,Cin] bf16 NHWC, Cin % 64 == 0, with 128 ZERO elements in front of it in memory (X16[-128..-1] == 0:
 *       the line padding taps read); Wb [Cout][KH][KW][Cin] bf16 with the BatchNorm scale folded in; shift f32 [Cout] or
 *       NULL; R16 bf16 [B,OH,OW,Cout] or NULL; the result goes to Y (f32) or Y16 (bf16), exactly one non-NULL. */
int rgrg_conv2d_nhwc_bf16(const uint16_t* X16, const uint16_t* Wb, const float* shift, const uint16_t* R16, float* Y,
                          uint16_t* Y16, int B, int H, int Wd, int Cin, int Cout, int KH, int KW, int stride, int pad,
                          int act, int fp16, void* stream);
int rgrg_linear_bf16w_f32(const float* A, const uint16_t* Wb, const float* shift, const float* R, float* Y, int M,
                          int N, int K, int ldy, int act, int fp16, void* stream);
/* The same product with BOTH operands already bf16 in device memory (A16 [M,K], Wb [N,K], K % 256 == 0): the
 * LDS-DMA kernel of the opt-in bf16 decode / box-head paths (operands go HBM -> LDS without touching registers, 4 LDS
 * stages in flight across the barriers).  fp32 accumulate; epilogue shift [N] / residual R fp32 [M,ldy] / activation
 * in fp32; the result is stored as fp32 (Y) or as bf16 (Y16, feeds the next GEMM) - exactly one of them non-NULL. */
int rgrg_linear_bf16_f32(const uint16_t* A16, const uint16_t* Wb, const float* shift, const float* R, float* Y,
                         uint16_t* Y16, int M, int N, int K, int ldy, int act, int fp16, void* stream);
/* Measurement hook (tools/gemm_bf16_bench.py): the LDS-DMA kernel with a forced configuration: tile = shape + 16 * stages
 * (shape 0 heuristic, 1 128x128, 2 64x64, 3 128x64, 4 64x128, 5 the 256x256 ping-pong kernel; stages 0 = 4, or 2 / 3 / 4 LDS
 * stages) and operand row
 * pitches lda / ldw in elements (0 = K). */
int rgrg_debug_linear_bf16_tile(const uint16_t* A16, const uint16_t* Wb, const float* shift, const float* R, float* Y,
                                int M, int N, int K, int ldy, int act, int tile, int lda, int ldw, int fp16, void* stream);
/* Test / measurement hook for the 16-bit training pass (torch.autocast around the reference's training step,
 * src/full_model/train_full_model.py:172-237): the same GEMM on a forced tile (as above; 5 = the 256 x 256 ping-pong kernel
 * of the large shapes) with the training epilogues - Y (f32) or Y16 (16 bit) = act(A16 Wb^T + shift + R); Ypre16 (optional):
 * the 16-bit value BEFORE the activation (c_fc keeps it for gelu_new'); G16 (optional, no R / act): the result is multiplied
 * by gelu_new'(G16[m, n]) - the activation gradient of GPT2MLP lands directly as d(c_fc output). */
int rgrg_debug_linear_bf16_train(const uint16_t* A16, const uint16_t* Wb, const float* shift, const float* R, float* Y,
                                 uint16_t* Y16, uint16_t* Ypre16, const uint16_t* G16, int M, int N, int K, int ldy, int act,
                                 int tile, int fp16, void* stream);
/* Test hook for the greedy lm_head of the many-sequence 16-bit decode step (src/language_model/language_model.py:420-428:
 * next_token_logits.argmax(-1)): the 256 x 256 GEMM leaves, per row and 256-column tile, the maximum of A16 Wb^T + shift and
 * its column (first maximum wins, like torch.argmax) in cand_val / cand_idx [M][ceil(N / 256)] instead of the logits. */
int rgrg_debug_linear_bf16_argmax(const uint16_t* A16, const uint16_t* Wb, const float* shift, int M, int N, int K,
                                  float* cand_val, int* cand_idx, int fp16, void* stream);
/* Test hooks for the LayerNorm folded around the 16-bit decode GEMMs (transformers GPT2Block: ln_1 -> c_attn, ln_2 -> c_fc;
 * src/language_model/language_model.py:338-366 runs them as separate modules).  rgrg_debug_ln_fold16: wb[n][k] =
 * round16(gain[k] w[n][k]), colsum[n] = sum_k wb[n][k] (of the ROUNDED values), shift[n] = bias[n] + sum_k beta[k] w[n][k].
 * rgrg_debug_linear_bf16_ln: the LDS-DMA GEMM as PRODUCER of the residual stream (Yb16 [M, ldy] = the fp32 result as 16 bit,
 * stats_out [M][16][2] = per-row (sum, sum of squares) of every 64-column block; N == 1024) or as CONSUMER (A16 = the raw
 * 16-bit rows, Wb = the scaled weights, ln_stats = those slots, ln_colsum; K == 1024, no residual):
 * Y = act(rstd_m (A16 Wb^T - mean_m colsum) + shift). */
int rgrg_debug_ln_fold16(const float* w, const float* gain, const float* beta, const float* bias, uint16_t* wb, float* colsum,
                         float* shift, int N, int K, int fp16, void* stream);
int rgrg_debug_linear_bf16_ln(const uint16_t* A16, const uint16_t* Wb, const float* shift, const float* R, float* Y,
                              uint16_t* Yb16, float* stats_out, const float* ln_stats, const float* ln_colsum, int M, int N,
                              int K, int ldy, int act, int fp16, void* stream);
/* The same GEMM variants (plus the plain one with a 16-bit output Y16; exactly one of Y / Y16) on the K-parity ping-pong kernel
 * (csrc/gemm_kp.inc, round 6) when kp != 0 - the kernel the decoder selects for the per-layer projections of the many-sequence
 * 16-bit decode step (GPT2Block's c_attn / c_proj / c_fc / mlp.c_proj, src/language_model/language_model.py:338-366): its tile
 * is a function of (N, K) only, so a row's result does not depend on M.  kp == 0: the LDS-DMA kernel's heuristic. */
int rgrg_debug_linear_bf16_ln_kp(const uint16_t* A16, const uint16_t* Wb, const float* shift, const float* R, float* Y,
                                 uint16_t* Y16, uint16_t* Yb16, float* stats_out, const float* ln_stats, const float* ln_colsum,
                                 int M, int N, int K, int ldy, int act, int fp16, int kp, void* stream);
/* ---- detector targets and losses: ObjectDetector.forward(images, targets), the detector half of
 * ReportGenerationModel.forward(images, image_targets, ...) (src/full_model/report_generation_model.py:55,91 ->
 * src/object_detector/object_detector.py:216-224 -> custom_rpn.py:74-83, custom_roi_heads.py:225-242, and underneath
 * torchvision 0.13.1 det_utils.Matcher / BoxCoder.encode / RegionProposalNetwork.compute_loss / fastrcnn_loss).
 * Matching, the BalancedPositiveNegativeSampler, add_gt_proposals, the gathers of the sampled rows and the losses are all
 * kernels; the caller allocates and passes pointers. */
/* Matcher: gt [B][G][4] (gt_count[b] valid rows), boxes [N][4] per image at boxes + b * box_image_stride floats
 * (stride 0: one shared set, the anchors), box_count[b] valid boxes (NULL: N).  matched [B][N] = index of the best gt
 * (first maximum), -1 below `low`, -2 between; allow_low_quality restores the arg-max of every box that ties a gt's
 * best IoU.  ws_best_per_gt: B * G ints of work space. */
int rgrg_box_match_f32(const float* gt, const int* gt_count, int G, const float* boxes, int64_t box_image_stride,
                       const int* box_count, int B, int N, float high, float low, int allow_low_quality, int* matched,
                       int* ws_best_per_gt, void* stream);
/* det_utils.BalancedPositiveNegativeSampler for B images at once (custom_rpn.py:79: batch 256, half positive;
 * custom_roi_heads.py:225: batch 512, a quarter positive).  The candidates' labels come from the match codes: matched
 * [B][n] (rgrg_box_match_f32), gt_labels [B][G] (NULL: RPN - matched >= 0 is a positive; else the matched box's class:
 * >= 1 positive, 0 negative, < 0 ignored), box_count[b] candidates per image (NULL: n).  Per image num_pos = min(#positive,
 * max_pos), num_neg = min(#negative, batch - num_pos); taken are the candidates with the SMALLEST keys of each class, lower
 * index first on ties: keys [B][n] fp32 when given (tests replay the reference's draws this way), else Philox4x32-10 words
 * of (seed, stage, image, index) - a uniformly random subset like torchvision's randperm()[:k].  ws_keys: B * n words of
 * work space.  mask [B][n]: 0 / 1 sampled positive / 2 sampled negative; list [B][batch]: the sampled indices ascending
 * (torch.where(pos | neg)), zero filled; count [B]. */
int rgrg_balanced_sample(const int* matched, const int64_t* gt_labels, int G, const int* box_count, const float* keys,
                         uint64_t seed, int stage, int B, int n, int batch, int max_pos, uint32_t* ws_keys, uint8_t* mask,
                         int* list, int* count, void* stream);
/* RegionProposalNetwork.compute_loss on the fused head output rpn_out [B * cells][ld] (anchors_per_cell objectness columns,
 * then 4 deltas per anchor) from the sampler's outputs: anchors [A][4] shared by the images, gt [B][G][4]; labels and
 * regression targets (BoxCoder.encode, weights 1) of the sampled anchors are derived in the kernel from mask / matched.
 * out2[0] = loss_objectness (BCE over the sampled anchors, mean), out2[1] = loss_rpn_box_reg (smooth-L1 beta 1/9 over the
 * sampled positives / #sampled); nan when nothing is sampled. */
int rgrg_rpn_loss_sampled_f32(const float* rpn_out, int ld, int anchors_per_cell, const int* matched, const float* gt, int G,
                              const float* anchors, const uint8_t* mask, const int* list, const int* count, int B, int A,
                              int batch, float* out2, void* stream);
/* RoIHeads.add_gt_proposals with static shapes: boxes [B][P + G][4] = props[b][:counts[b]], gt[b][:gt_count[b]], zeros;
 * box_count[b] = counts[b] + gt_count[b]. */
int rgrg_roi_add_gt_f32(const float* props, const int* counts, int P, const float* gt, const int* gt_count, int G, int B,
                        float* boxes, int* box_count, void* stream);
/* The sampled rows of RoIHeads.select_training_samples from (boxes, matched, list, count): props_s [B][K][4] zero padded,
 * offsets [B + 1], and in RoI order (row offsets[b] + k) labels_flat [B * K] and reg_targets [B * K][4] =
 * BoxCoder(wx, wy, ww, wh).encode(matched gt box, proposal); rows from offsets[B] on are left untouched. */
int rgrg_roi_gather_samples_f32(const float* boxes, const int* matched, const float* gt, const int64_t* gt_labels,
                                const int* gt_count, int G, const int* list, const int* count, int B, int N, int K, float wx,
                                float wy, float ww, float wh, float* props_s, int* offsets, int64_t* labels_flat,
                                float* reg_targets, void* stream);
/* fastrcnn_loss on pred [N][ld] = num_classes logits | num_classes x 4 deltas.  out2[0] = loss_classifier,
 * out2[1] = loss_box_reg. */
int rgrg_fastrcnn_loss_f32(const float* pred, int ld, int num_classes, const int64_t* labels, const float* reg_targets, int N,
                           float* out2, void* stream);

/* Debug/parity taps: logits of the LAST executed step [S, vocab] -> dst (device). */
int rgrg_decoder_copy_last_logits(rgrg_decoder* d, float* dst, int S, void* stream);
/* bench.py roofline, any sequence count / precision mode: `iters` replays of (a) the projection GEMM launches of one
 * decode step for S token rows as the step would launch them, (b) its 24 single-query attention launches at `nkeys`
 * keys per sequence, each family between one pair of HIP events on the decoder's stream.  Returns total ms of each,
 * the algorithmic flops and weight bytes of the GEMMs of ONE step, the K/V cache bytes ONE step reads at `nkeys`
 * keys (24 x 2 x S x 1024 x nkeys x element size), and the GEMM launches per step.  Call after a generate() so that
 * the decoder exists in the wanted precision mode.  one_range != 0: every launch covers all S rows (each kernel alone on the
 * GPU at the step's full size - what a serialising profiler sees of a 1-range step); 0: as the step launches them (the
 * many-sequence 16-bit step runs as concurrent row ranges on forked streams, whose launches overlap). */
int rgrg_decoder_time_step_parts(rgrg_decoder* d, int S, int nkeys, int iters, int one_range, float* ms_gemm, float* ms_attn,
                                 double* gemm_flops, double* gemm_weight_bytes, double* kv_bytes, int* gemm_launches);
/* Measurement hook: `iters` + 1 eagerly enqueued many-sequence decode steps at `nkeys` keys (state of the last generate() of S
 * sequences), the last one with an event behind every launch on the stream it was launched on.  recs [max_recs][3] = (first row
 * of the launch's row range, tag, ms since the step's first launch); tag = layer * 8 + {0 c_attn, 1 attention, 2 attn c_proj,
 * 3 c_fc, 4 mlp c_proj} (GPT2Block, src/language_model/language_model.py:338-366), 1000 embedding, 1001 ln_f, 1002 lm_head,
 * 1003 arg-max + bookkeeping (:629-650). */
int rgrg_decoder_trace_step(rgrg_decoder* d, int S, int nkeys, int iters, float* recs, int max_recs, int* n_out);
/* Measurement hook: `iters` x n_layer single-query attention launches (GPT2PseudoAttention, language_model.py:124-180) of the
 * many-sequence step at `nkeys` keys, asynchronously on `stream` (state of the last generate() of S sequences). */
int rgrg_decoder_attention_only(rgrg_decoder* d, int S, int nkeys, int iters, void* stream);
/* Token rows (sequences x beams) up to which a decode step of d runs the fused fragment-direct plan in its CURRENT precision mode
 * (128 in fp32; under autocast 64: <= 32 rows bit-exact fp32, 33-64 on 16-bit weights); more rows take the many-sequence path.
 * -1 for a null handle.  (No reference counterpart: the reference has one code path, HF GPT-2 modules under torch.autocast.) */
int rgrg_decoder_row_limit(rgrg_decoder* d);
/* Measurement hook (tools/attn_confine_probe.py): where the hardware placed each of n_wgs one-wave workgroups - out[2 i] = HW_ID
 * (wave / SIMD / CU / SH / SE fields), out[2 i + 1] = XCC_ID of workgroup i.  No reference counterpart. */
int rgrg_debug_hw_ids(unsigned* out, int n_wgs, void* stream);

/* Measurement helper (tools/microbench.py; not on the product path): host wall
 * microseconds per kernel of a dependent chain of n trivial kernels; mode 0 = eager on a
 * private stream, 1 = one hipGraph replay, 2 = eager on the null stream. */
int rgrg_debug_chain(int n, int mode, int blocks, float* us_per_kernel);
/* Measurement helper (tools/grid_barrier_bench.py): microseconds per grid-wide barrier among 256 resident workgroups of
 * 512 threads, with a payload hand-off of `payload_floats` per workgroup around it.  variant: 0 one atomic counter with
 * agent-scope fences, 1 without fences, 2 per-workgroup flags with fences, 3 flags with coherent payload accesses and no
 * fences, 4 two-level counters with fences, 5 the fences alone; round 4 (write-through sc1 payload stores, every workgroup
 * reads the slices of 64 producers): 6 XCD-hierarchical counters + one agent-scope acquire per workgroup, 7 the same with
 * sc1 consumer loads instead of the acquire, 8 one flat relaxed counter + sc1 loads; `variant | 0x100` = uneven load (stale-read
 * check).  payload_floats % 4 == 0.  stale_out[0] = hand-offs that read old data, stale_out[1] = 1 when a bounded spin gave up. */
int rgrg_debug_grid_barrier(int variant, int iters, int payload_floats, float* us_per_barrier, unsigned* stale_out);
/* Measurement helper: the XCD (0..7) every workgroup of `launches` back-to-back launches of `blocks` workgroups ran on,
 * out_host[launch * blocks + b] (host memory) - the placement the decode plan's L2 prefetch workgroups count on. */
int rgrg_debug_xcc_map(int blocks, int launches, int* out_host);

#ifdef __cplusplus
}
#endif
#endif /* RGRG_HIP_H_ */
