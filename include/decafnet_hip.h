/*
 * libdecafnet_hip.so -- C ABI of the MI355X (gfx950) DeCafNet grounding hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no torch types.  Every entry point
 * names the reference interface it replaces (paths relative to the reference repository).
 * All `const float*` / `uint8_t*` DATA pointers are DEVICE pointers (hipMalloc'ed or
 * torch-ROCm storage) unless a parameter says "host".  `stream` is a hipStream_t passed as
 * void* (NULL = the default stream).  Functions return 0 on success and -1 on failure;
 * dcf_last_error() then holds a message (thread local).
 *
 * There is NO CPU fallback: if the library, the device or a kernel is missing the call fails.
 */
#ifndef DECAFNET_HIP_H
#define DECAFNET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char* dcf_last_error(void);
int dcf_abi_version(void);

/* --------------------------------------------------------------------------------------------
 * Model: replaces PtTransformerEarlyFusionIterative (libs/modeling/model.py:397-565) as built by
 * create_model (libs/worker_v2.py:182-211).  Hyper-parameters mirror opt.model.* (libs/core/opt.py:75-130).
 * ------------------------------------------------------------------------------------------ */
typedef struct dcf_model dcf_model;

typedef struct dcf_config {
  int32_t D;              /* opt.model.vid_net.in_dim: expert / sidekick feature dim               */
  int32_t E;              /* opt.model.vid_net.embd_dim                                            */
  int32_t TE;             /* opt.model.text_net.embd_dim = fusion.text_dim                         */
  int32_t vid_heads;      /* opt.model.vid_net.n_heads                                             */
  int32_t fusion_heads;   /* opt.model.fusion.n_heads                                              */
  int32_t fusion_layers;  /* opt.model.fusion.n_layers                                             */
  int32_t n_embd_convs;   /* opt.model.vid_net.arch[0]                                             */
  int32_t n_stem;         /* opt.model.vid_net.arch[1]                                             */
  int32_t n_levels;       /* opt.model.vid_net.arch[2]  (= number of FPN levels = TCN depth)       */
  int32_t win;            /* opt.model.vid_net.mha_win_size (odd; 0 = global clip attention: a correctness-first fp32
                           * O(T^2) vector-ALU kernel, csrc/attn.hip k_global_attn -- no reference default uses it)  */
  int32_t head_layers;    /* opt.model.cls_head.n_layers (= reg_head.n_layers)                     */
  int32_t sn;             /* opt.model.sn     (clips per scoring block)                            */
  float sratio;           /* opt.model.sratio (fraction of blocks that keep expert features)       */
  int32_t msf;            /* opt.model.msf                                                         */
  int32_t norm;           /* opt.model.norm                                                        */
  int32_t use_abs_pe;     /* opt.model.vid_net.use_abs_pe                                          */
  int32_t max_batch;      /* queries processed together (>= 1); 0 = library default                */
  int32_t gemm_mode;      /* dense-conv arithmetic, all fp32-accurate: 0/16 = f16x3 (two fp16 planes per operand,
                           * 3 MFMA products, default), 6 = bf16x6 (three bf16 planes, 6 products),
                           * 1 = native fp32 MFMA                                                   */
  int32_t model_kind;     /* 0 = PtTransformerEarlyFusionIterative (libs/modeling/model.py:397),
                           * 1 = PtTransformer, late fusion (model.py:30),
                           * 2 = PtTransformerEarlyFusion (model.py:163): kind 0 without the refinement stage    */
  int32_t second_fusion;  /* model_kind 0: also fuse every pyramid level before the heads (model.py:443) */
  /* text_net = TextTransformer (libs/modeling/text_net.py:92-188); text_in = 0 builds the model without it */
  int32_t text_in;        /* opt.model.text_net.in_dim (token feature dim C_t)                     */
  int32_t text_layers;    /* opt.model.text_net.n_layers                                           */
  int32_t text_heads;     /* opt.model.text_net.n_heads                                            */
  int32_t text_abs_pe;    /* opt.model.text_net.use_abs_pe                                         */
  int32_t text_bkgd;      /* opt.model.text_net.use_bkgd_token                                     */
  /* ABI version 3 */
  int32_t scat;           /* opt.model.scat: the clip's raw sidekick score is one more vid_map input channel
                           * (model.py:413-414,550-551; PtTransformer model.py:46-47,128-129)      */
  int32_t sfonly;         /* opt.model.sfonly: with msf, vid_map sees the sidekick features only (model.py:546-547);
                           * D is then the sidekick feature dim = 2 * opt.model.vid_net.in_dim; ignored by model_kind 1 */
  int32_t text_kind;      /* 0 = TextTransformer (text_net.py:92-188), 1 = TextIdentity (text_net.py:22-89: optional embd_fc,
                           * optional AttNPool1D token when text_bkgd != 0; text_layers is ignored)  */
  /* ABI version 5 */
  int32_t xattn_affine;   /* opt.model.fusion.xattn_mode: 0 = 'adaln' (the decoder modulates LayerNorm(q), blocks.py:623-624,643),
                           * 1 = 'affine' (it modulates q itself: nn.Identity, blocks.py:625-626)   */
  /* ABI version 7 */
  int32_t vid_stride;     /* opt.model.vid_net.stride (video_net.py:39,59-74): a power of two; the first log2(stride) embedding
                           * convolutions are k5 / stride 2 / padding 2 and the pyramid starts at T / stride.  0 reads as 1 */
  int32_t pool_only;      /* opt.model.vid_net.pool_only (video_net.py:98-111): a branch layer is one depthwise k3 MaskedConv1D
                           * (vid_net.branch.{i}.conv.weight, stride 1 at level 0, 2 above) instead of a TransformerEncoder */
  /* ABI version 9 */
  int32_t attn_mode;      /* arithmetic of the attention products QK^T / PV on the matrix cores (text->clip cross attention, the
                           * fusion layers' attention half, the encoder's window attention): 0 = f16x3 (two fp16 planes per operand,
                           * three products: the error of an fp32 FMA chain; default), 1 = ONE fp16 product per multiply-add
                           * (11 significant bits per operand, fp32 accumulate -- what BASELINE configs[4] calls "bf16 MFMA
                           * attention", with fp16's three extra bits; opt-in, NOT fp32-accurate: bench.py --attn-mode f16 prints
                           * its delta against the fp32 oracle).  The projections around the products stay f16x3. */
} dcf_config;

int dcf_model_create(const dcf_config* cfg, dcf_model** out);
void dcf_model_destroy(dcf_model* m);

/* Bind one tensor of the reference state_dict by its NAME (the parameter ABI):
 * "vid_map.conv.weight", "fusion.layers.0.xattn.xattn.query.weight", ...  The memory is borrowed
 * (fp32, contiguous, device) and must stay alive until the model is destroyed or re-bound.
 * text_net.* tensors feed dcf_text_encode (ignored when cfg.text_in == 0).
 * Replaces nn.Module.load_state_dict(ckpt['model_ema']) (libs/worker_v2.py:806-812). */
int dcf_model_bind(dcf_model* m, const char* name, const float* data, const int64_t* shape, int32_t ndim);

/* Numerics status of the f16x3 GEMMs (blocking: synchronises `stream`).  Returns a bit set, or -1 on error:
 *   1 = some GEMM accumulator left the finite range since the last reset (an activation beyond the fp16 operand range
 *       |a| < 4094 of the f16x3 mode, or inf/NaN in the inputs): outputs since then are not trustworthy -- re-run with
 *       gemm_mode = 6;   2 = a weight did not fit (|w| >= 255.9) and the model fell back to bf16x6 at finalize;
 *   4 = the model runs bf16x6;  8 = the model runs the native fp32 MFMA path;
 *  16 = a LayerNorm that rides between two kernels as one-pass row statistics (sum, sum of squares) met a row whose mean dwarfs its
 *       spread (mean^2 > 64 (var + eps)): the variance of such a row loses more than ~1e-5 to cancellation -- call
 *       dcf_model_set_ln_carry(m, 0) and repeat the forward (ABI version 9).   reset != 0 clears the sources of 1 and 16. */
int dcf_numerics_status(dcf_model* m, int32_t reset, void* stream);
/* The same status word without blocking: enqueues a 4-byte device -> host copy of the sticky flag (bit 0 above) into
 * `host_dst` (pinned memory) on `stream`; the caller reads it once an event recorded after the call has completed.
 * Independently of either call, a forward that ends with the flag raised overwrites its logits with NaN, so that a plain
 * `model(...)` caller (the reference's Evaluator, libs/worker_v2.py:1007) cannot mistake them for valid scores.
 * (The word copied here is the raw device word: bit 0 = status 1, bit 1 = status 16.) */
int dcf_numerics_status_async(dcf_model* m, int32_t* host_dst, void* stream);
/* on = 0: every LayerNorm of the model runs as its own two-pass launch (blocks.py:125-131: mean, then the mean of squared deviations)
 * instead of riding between kernels as one-pass row statistics; on = 1 (default): carried where the kernels allow.  Drops captured
 * graphs.  What dcf_numerics_status bit 16 asks for.  ABI version 9. */
int dcf_model_set_ln_carry(dcf_model* m, int32_t on);

/* Absolute position encoding buffer `vid_net.pe` (non-persistent in the reference,
 * libs/modeling/video_net.py:75-78): token-major (T, E) fp32 already resampled for length T
 * (video_net.py:141-151).  Borrowed.  Needed only when use_abs_pe != 0. */
int dcf_model_set_pe(dcf_model* m, const float* pe_tokens, int64_t T);

/* Position encoding buffer `text_net.pe` (non-persistent, text_net.py:120-125): token-major (L, TE) fp32,
 * L >= the longest query (already resampled if longer than max_seq_len, text_net.py:172-177).  Borrowed. */
int dcf_model_set_text_pe(dcf_model* m, const float* pe_tokens, int64_t L);

/* Text encoder of one query: replaces model.encode_text(tokens, token_masks) = TextTransformer.forward
 * (libs/modeling/model.py:434-436, text_net.py:158-188; caller libs/worker_v2.py:953).
 *   tokens     : (C_t, Lq) fp32 channel-major = tensor[0] of the reference's (1, C_t, Lq) input
 *   token_mask : (Lq) bytes, 1 = valid token; NULL = all valid
 *   text_out   : (TE, Lk) fp32 channel-major, Lk = Lq + use_bkgd_token  (the layout dcf_forward_eval takes)
 *   mask_out   : (Lk) bytes = cat(mask[:1], mask) */
int dcf_text_encode(dcf_model* m, const float* tokens, const uint8_t* token_mask, int32_t Lq, float* text_out,
                    uint8_t* mask_out, void* stream);

/* Validate that every parameter is bound and repack convolution weights for the kernels. */
int dcf_model_finalize(dcf_model* m, void* stream);

/* Number of points per query, sum_l T / 2^l. */
int64_t dcf_points_per_query(const dcf_model* m, int64_t T);

/* Eval forward: replaces model(vid, shallow_vid, vid_masks, text, text_cls, text_masks, eval=True)
 * (libs/modeling/model.py:473-565; caller libs/worker_v2.py:1007).
 *   vid, shallow_vid : (D, T) fp32 channel-major = tensor[0] of the reference's (1, D, T) inputs
 *   vid_mask         : (T) bytes (torch.bool storage), 1 = valid clip
 *   text[q]          : HOST array of nq device pointers, each (TE, text_len[q]) fp32 = encode_text()[0][0]
 *   text_mask[q]     : HOST array of nq device pointers, each (text_len[q]) bytes; entries may be NULL (= all valid)
 *   text_len         : HOST array of nq ints
 *   text_cls         : (nq, D) fp32
 * Outputs (device, caller allocated), S = dcf_points_per_query(T), levels concatenated l = 0..L-1:
 *   logits_out (nq, S) raw fp32 logits;  offsets_out (nq, S, 2) fp32 >= 0;  masks_out (nq, S) bytes. */
int dcf_forward_eval(dcf_model* m, const float* vid, const float* shallow_vid, const uint8_t* vid_mask, int64_t T,
                     int32_t nq, const float* const* text, const uint8_t* const* text_mask, const int32_t* text_len,
                     const float* text_cls, float* logits_out, float* offsets_out, uint8_t* masks_out, void* stream);

/* Same forward on a WINDOW of a longer video (T-sharding across GPUs): the 0/1 clip gate
 * (nq, T) fp32 is supplied by the caller, who selected the top-k blocks on the all-gathered sidekick scores
 * of the whole video; the position encoding set by dcf_model_set_pe must be the window's slice of the
 * whole video's encoding.  No scoring, no text_cls. */
int dcf_forward_eval_gated(dcf_model* m, const float* vid, const float* shallow_vid, const uint8_t* vid_mask, int64_t T,
                           int32_t nq, const float* const* text, const uint8_t* const* text_mask, const int32_t* text_len,
                           const float* gate, float* logits_out, float* offsets_out, uint8_t* masks_out, void* stream);

/* Training-mode forward, forward values only (PtTransformerEarlyFusionIterative._drop_forward, libs/modeling/model.py:567-632,
 * with every dropout / drop-path probability 0): a batch of videos of one padded length, video v repeated for its
 * text_size[v] = nq_per_video[v] queries (`repeat_interleave`, model.py:579-582), i.e. the rows of the batch are the
 * (video, query) pairs in video order.  Arguments as dcf_forward_eval_videos; returns, per pair, what fuse_and_predict
 * returns (model.py:442-471): logits1_out (sum(nq), S) of the first cls_head, logits2_out (sum(nq), S), offsets_out
 * (sum(nq), S, 2), masks_out (sum(nq), S).  No backward pass.  ABI version 5. */
int dcf_forward_train_videos(dcf_model* m, int32_t nvid, const float* const* vid, const float* const* shallow_vid,
                             const uint8_t* const* vid_mask, int64_t T, const int32_t* nq_per_video, const float* const* text,
                             const uint8_t* const* text_mask, const int32_t* text_len, const float* const* text_cls,
                             float* logits1_out, float* logits2_out, float* offsets_out, uint8_t* masks_out, void* stream);

/* Point losses, forward values (libs/modeling/loss.py; used by Trainer.forward_backward, libs/worker_v2.py:441-461).
 *   dcf_sigmoid_focal_loss <- sigmoid_focal_loss(inputs, targets, alpha, gamma, smoothing, reduction)   (loss.py:5-57)
 *   dcf_ctr_iou_loss       <- ctr_giou_loss / ctr_diou_loss(input_offsets, target_offsets, reduction, eps) (loss.py:60-166)
 * n elements (rows of 2 offsets for the IoU losses) on the device; `select` (optional, n bytes) keeps the elements with a
 * non-zero byte -- the boolean-mask indexing `x[fpn_masks]` / `x[pos_masks]` of the caller without a compaction pass.
 * elem_out (optional, n floats): the per-element loss (reduction 'none'; unselected elements get 0); sum_out (optional,
 * 1 float) the sum over the selected elements and count_out (optional, 1 int32) their number ('sum' / 'mean'), added up
 * in a fixed order (deterministic).  kind: 0 = GIoU (reduces to IoU, loss.py:104), 1 = DIoU. */
int dcf_sigmoid_focal_loss(const float* inputs, const float* targets, const uint8_t* select, int64_t n, float alpha, float gamma,
                           int32_t smoothing, float* elem_out, float* sum_out, int32_t* count_out, void* stream);
int dcf_ctr_iou_loss(const float* input_offsets, const float* target_offsets, const uint8_t* select, int64_t n, int32_t kind,
                     float eps, float* elem_out, float* sum_out, int32_t* count_out, void* stream);

/* Throughput extension: several videos of the SAME padded length T in one forward (the reference evaluates one video per
 * call, model.py:496; videos no longer than opt.model.max_vid_len are all padded to that length, worker_v2.py:969-976).
 * Video v has nq_per_video[v] queries; text / text_mask / text_len and the outputs list the queries of all videos in
 * video order (sum(nq) entries, outputs (sum(nq), S) ...); text_cls[v] is (nq_per_video[v], D).  After vid_map every
 * kernel works on rows [query][t], so the result of each query is the one dcf_forward_eval gives for its video alone.
 * 1 <= nvid <= 16; with nvid > 1 max_batch must be <= 16.  ABI version 3. */
int dcf_forward_eval_videos(dcf_model* m, int32_t nvid, const float* const* vid, const float* const* shallow_vid,
                            const uint8_t* const* vid_mask, int64_t T, const int32_t* nq_per_video, const float* const* text,
                            const uint8_t* const* text_mask, const int32_t* text_len, const float* const* text_cls,
                            float* logits_out, float* offsets_out, uint8_t* masks_out, void* stream);

/* Debug taps for parity tests.  what 0 / 1 / 4 copy an intermediate of the LAST forward chunk into `dst` (device) now:
 *   0 = sidekick scores (nq, T); 1 = gate (B, T); 4 = pyramid features (B*S rows [level][b][t], E+32).
 * what 2 / 3 ARM a one-shot tap: the NEXT forward (run eagerly, no graph) copies 2 = the vid_map output / 3 = the fusion
 * output (B*T, E) token-major of its last query chunk into `dst` and disarms the tap; `dst` must stay alive until then and
 * hold max_floats >= B*T*E floats (checked by that forward). */
int dcf_debug_copy(dcf_model* m, int32_t what, float* dst, int64_t max_floats, void* stream);

/* One long video with few queries, T-sharded over ranks WITHOUT recomputing the top of the pyramid (extension; the reference's eval is
 * single-GPU, libs/worker_v2.py:922-924).  The pyramid is cut at level k: a rank runs levels 0..k on a NARROW window of Tn clips and levels
 * k+1..L-1 on a COARSE window of Tc level-k rows taken from the all-gathered level-k feature map; the refinement TCN (model.py:449-458)
 * couples the levels, hence a second exchange (the refined level-k map).  cvpr2025-decafnet_amd/dist.py `hybrid_plan` / `hybrid_forward`
 * derive the windows and run the collectives; windows are treated as sequences, their halos absorb the ends.  ABI version 8.
 *   phase 1: vid_map, early fusion, embedding, levels 0..k on the narrow window with the caller's gate (as dcf_forward_eval_gated; set the
 *            window's position-encoding slice with dcf_model_set_pe first) -> featk_out (nq, Tn >> k, E)
 *   phase 2: featk_c (nq, Tc, E) / maskk_c (Tc): the gathered level-k features and level-k validity on the coarse window; off_k = first
 *            level-k row of the narrow window inside the coarse one -> refk_out (nq, Tn >> k, 32): the refined map at level k
 *   phase 3: refk_c (nq, Tc, 32): the gathered refined map on the coarse window -> outputs of levels 0..k on the narrow window
 *            (logits_n (nq, Sn), offsets_n (nq, Sn, 2), masks_n (nq, Sn), Sn = sum_{l<=k} Tn >> l) and of levels k+1.. on the coarse one
 *            (.._c, Sc = sum_{j>=1} Tc >> j), levels concatenated as in dcf_forward_eval.
 * Any other forward on the model between the phases invalidates them (they share its workspace). */
int dcf_hybrid_phase1(dcf_model* m, int32_t k, const float* vid_w, const float* shallow_w, const uint8_t* mask_w, int64_t Tn, int64_t Tc,
                      int32_t nq, const float* const* text, const uint8_t* const* text_mask, const int32_t* text_len, const float* gate_w,
                      float* featk_out, void* stream);
int dcf_hybrid_phase2(dcf_model* m, const float* featk_c, const uint8_t* maskk_c, int64_t off_k, float* refk_out, void* stream);
int dcf_hybrid_phase3(dcf_model* m, const float* refk_c, float* logits_n, float* offsets_n, uint8_t* masks_n, float* logits_c,
                      float* offsets_c, uint8_t* masks_c, void* stream);

/* How the last dcf_forward_eval* call on this model was issued: 0 = eager kernel launches, 1 = replay of the captured HIP
 * graph, 2 = the call that captured the graph (and launched it).  A forward called on the NULL (legacy default) stream,
 * which cannot be captured, runs on an engine-owned stream ordered after / before the caller's by events.  ABI version 4. */
int dcf_graph_active(const dcf_model* m);
/* HIP-graph policy of the repeated forward: 0 = auto (replay for forwards of >= 65536 level-0 rows, i.e. batched queries /
 * videos; eager launches for the one-video-per-call pattern, which measures 5 % faster that way), 1 = always capture and
 * replay, 2 = never.  The environment variable DCF_NO_GRAPH=1 disables graphs whatever the mode.  ABI version 5. */
int dcf_model_set_graph_mode(dcf_model* m, int32_t mode);
/* Test / developer switch (process wide): override a built-in dispatch threshold so that the operator tests can send small
 * reference fixtures through the kernels the engine only picks for large grids.  Names: "dec_chain_min_rows" (level-0 rows from
 * which the attention half of a fusion layer runs as one kernel, csrc/dec_chain.hip), "enc_chain_min_rows" / "enc_attn_min_rows" (the same for
 * the two halves of the encoder layers' attention, csrc/enc_chain.hip), "fuse_scores" (0: the sidekick scores by their own kernels instead
 * of on the shallow vid_map GEMM), "tcn_frag" (0: every workgroup of a TCN layer builds its weight fragments itself instead of reading the
 * per-model image).  value < 0 restores the built-in value.  Not a reference interface; results do not depend on it beyond
 * rounding.  ABI version 6 (the last two names: 8). */
int dcf_debug_set_option(const char* name, int32_t value);
/* Measurement aid (not a reference interface): the matrix rate THIS device sustains.  Runs bare fp16 MFMA loops (shape 0:
 * v_mfma_f32_32x32x16_f16, shape 1: v_mfma_f32_16x16x32_f16; operands in registers, one wave per SIMD, one workgroup per CU,
 * mfmas_per_wave MFMAs per wave, a few launches back to back on the NULL stream, synchronous) and returns the CU count and the
 * nanoseconds one MFMA occupies a SIMD for.  Nominal: 32 (16) cycles at 2.4 GHz = 13.3 (6.7) ns; under load the part holds
 * 1.5 - 1.75 GHz, which bounds every kernel of the dense family below ~0.7 of the nominal peak the roofline prices against.
 * bench.py reports it as roofline.checks.mfma_sustained.  ABI version 10. */
int dcf_calib_mfma_rate(int32_t shape, int32_t mfmas_per_wave, int32_t* n_cus, float* ns_per_mfma);

/* --------------------------------------------------------------------------------------------
 * Proposal decoding: replaces Evaluator._collect_segments (libs/worker_v2.py:1131-1187).
 *   logits (nq, S), offsets (nq, S, 2), masks (nq, S) as produced by dcf_forward_eval.
 *   segs_out (nq, pre_nms_topk, 2), scores_out (nq, pre_nms_topk), counts_out (nq) int32 (device).
 * ------------------------------------------------------------------------------------------ */
int dcf_collect_segments(const float* logits, const float* offsets, const uint8_t* masks, int32_t nq, int64_t T,
                         int32_t n_levels, float pre_nms_thresh, int32_t pre_nms_topk, float seg_len_thresh,
                         float* segs_out, float* scores_out, int32_t* counts_out, void* stream);

/* The same with the optional external per-clip scores of _collect_segments (worker_v2.py:1137,1150-1156):
 * ext_scores (nq, T) fp32 device or NULL; they multiply sigmoid(logits) level by level after a k3/s2/p1 max-pool
 * per level.  ABI version 3. */
int dcf_collect_segments_ext(const float* logits, const float* offsets, const uint8_t* masks, const float* ext_scores,
                             int32_t nq, int64_t T, int32_t n_levels, float pre_nms_thresh, int32_t pre_nms_topk,
                             float seg_len_thresh, float* segs_out, float* scores_out, int32_t* counts_out, void* stream);

/* --------------------------------------------------------------------------------------------
 * NMS: replaces the extension module nms_1d_cpu_vg (libs/nms/src/nms_cpu.cpp:184-194).
 * Batched over nq independent problems laid out with a fixed `stride` (entries) per problem;
 * counts (device int32, may be NULL => every problem has n_max entries).  n_max <= 4096.
 *   dcf_nms_1d      <- nms_1d(segs, scores, iou_thresh)                       (nms_cpu.cpp:20-70)
 *   dcf_softnms_1d  <- softnms_1d(segs, scores, dets, iou_thresh, sigma, min_score, method) (:72-181)
 *   max_iters = 0 reproduces the reference (all picks); k > 0 stops after k picks (the only
 *   ones SoftNMSop keeps when max_num_segs = k, libs/nms/nms.py:54-59).
 * ------------------------------------------------------------------------------------------ */
int dcf_nms_1d(const float* segs, const float* scores, const int32_t* counts, int32_t nq, int32_t n_max,
               int32_t stride, float iou_thresh, int64_t* keep_out, int32_t* keep_counts_out, void* stream);
int dcf_softnms_1d(const float* segs, const float* scores, const int32_t* counts, int32_t nq, int32_t n_max,
                   int32_t stride, float iou_thresh, float sigma, float min_score, int32_t method,
                   int32_t max_iters, float* dets_out, int64_t* inds_out, int32_t* out_counts, void* stream);
/* segment_voting (libs/nms/nms.py:64-103). nms_segs rows have `nms_ld` floats (2, or 3 for dets). */
int dcf_segment_voting(const float* nms_segs, int32_t nms_ld, const int32_t* n1_counts, int32_t n1_max,
                       int32_t n1_stride, const float* all_segs, const float* all_scores,
                       const int32_t* n2_counts, int32_t n2_max, int32_t n2_stride, float iou_thresh,
                       int32_t nq, float* out, void* stream);

/* --------------------------------------------------------------------------------------------
 * Measurement aid: when enabled, every kernel launch of the library is bracketed by HIP events on
 * its own stream.  dcf_profile_report writes a JSON object {kernel: {count, ms, flops, bytes}} (flops
 * and bytes are the ALGORITHMIC work of the launches) into buf and returns the required size.
 * ------------------------------------------------------------------------------------------ */
int dcf_profile_enable(int32_t on);
int64_t dcf_profile_report(char* buf, int64_t cap);

/* --------------------------------------------------------------------------------------------
 * Single-operator entry points (used by the parity tests and micro-benchmarks).
 * ------------------------------------------------------------------------------------------ */
/* C[M][N] = act(A[M][K] * W[N][K]^T + bias): nn.Conv1d(k=1) on token-major activations.
 * act: 0 none, 1 exact GELU, 2 ReLU. */
int dcf_op_linear(const float* A, const float* W, const float* bias, float* C, int32_t M, int32_t N, int32_t K,
                  int32_t act, void* stream);
/* same through the bf16-split MFMA GEMM (gemm_bf16s.hip); nterms = 6 (fp32 accurate) or 3 */
int dcf_op_linear_split(const float* A, const float* W, const float* bias, float* C, int32_t M, int32_t N, int32_t K,
                        int32_t act, int32_t nterms, void* stream);
/* same with A given channel-major (K, M) -- the reference's (C, T) layout */
int dcf_op_linear_cm(const float* A_cm, const float* W, const float* bias, float* C, int32_t M, int32_t N, int32_t K,
                     void* stream);
/* Y = LayerNorm_channels(A W^T + bias) * ln_w + ln_b [ReLU], the normalisation fused into the GEMM epilogue (how the
 * head trunks run: MaskedConv1D -> LayerNorm -> ReLU, libs/modeling/head.py:53-64); C (optional, may be NULL) receives
 * the raw product.  Only shapes the engine fuses: N = 256, M >= 28672, K % 32 == 0. */
int dcf_op_linear_ln(const float* A, const float* W, const float* bias, const float* ln_w, const float* ln_b, float* C, float* Y,
                     int32_t M, int32_t N, int32_t K, int32_t relu, int32_t nterms, void* stream);

/* Two chained 1x1 convolutions with a channel LayerNorm between them, the LayerNorm carried as row statistics instead of a
 * pass over the rows (how attn.proj -> ln_ffn -> ffn.fc runs, libs/modeling/blocks.py:586-590):
 *   X = A W1^T + b1 (+ R)            written with (sum, sum of squares) of every row on the side
 *   Y = act(LayerNorm(X) W2^T + b2)  computed from the RAW X with ln_w folded into W2 and (mean, rstd) applied in the epilogue
 * act = erf GELU if gelu != 0.  Shapes the engine carries: N1 % 64 == 0, tile-kernel grids (M * N / 4096 > 256). */
int dcf_op_linear_ln_carry(const float* A, const float* W1, const float* b1, const float* R, const float* ln_w, const float* ln_b,
                           const float* W2, const float* b2, float* X, float* Y, int32_t M, int32_t N1, int32_t K1, int32_t N2,
                           int32_t gelu, int32_t nterms, void* stream);

/* The FFN half of a transformer block (libs/modeling/blocks.py:535-538 with the residual of :589-590):
 *   C = X + ls * ((GELU(LN(X) W1^T + b1) W2^T + b2) * mask),   W1 (4E, E), W2 (E, 4E), hidden width 4E
 * ln_w / ln_b NULL = no LayerNorm in front; ls NULL = 1; mask NULL = all rows valid; stats_out (optional, (M, E / 64, 2)):
 * (sum, sum of squares) of every row written to C.  f16x3 operand split.  chain = 0: two GEMMs with the hidden activations in
 * memory; chain = 1 (E = 256 only): one kernel, the hidden activations stay in registers (csrc/ffn_chain.hip: the default kernel;
 * chain = 2: its four-wave form, chain = 3: its eight-wave producer / consumer form -- bit-identical results).  C must not alias X.  With a LayerNorm
 * in front the one-kernel form carries it as one-pass row statistics; if a row turns out ill-conditioned for that (dcf_numerics_status
 * bit 16) the call repeats itself with the LayerNorm as its own two-pass launch, as the engine does after dcf_model_set_ln_carry(m, 0). */
int dcf_op_ffn(const float* X, const float* ln_w, const float* ln_b, const float* W1, const float* b1, const float* W2, const float* b2,
               const float* ls, const uint8_t* mask, float* C, float* stats_out, int32_t M, int32_t E, int32_t chain, void* stream);

/* same product on the bf16-split matrix-core path (how vid_map runs); needs M % 4 == 0, N % 128 == 0, K % 32 == 0 */
int dcf_op_linear_cm_split(const float* A_cm, const float* W, const float* bias, float* C, int32_t M, int32_t N, int32_t K,
                           int32_t nterms, void* stream);
/* A prediction head (libs/modeling/head.py:53-64 ClsHead, :95-103 RegHead) on B sequences of T token-major rows (B*T, C):
 *   x = relu(LayerNorm(MaskedConv1D_k3(x, mask)))  twice (W1 / ln1, W2 / ln2: PyTorch (C, C, 3) weights, no bias), then
 *   out (B*T, NO) = MaskedConv1D_k3(x, mask) with Wout (NO, C, 3) and bout; scale != 0: relu(scale * out) (RegHead's Scale + ReLU).
 * f16x3 operand split.  chain = 0: GEMM, LayerNorm and output-convolution launches with the trunk activations in memory;
 * chain = 1 (C = 256 / 288): one kernel, the trunk activations stay in registers (csrc/head_chain.hip). */
int dcf_op_head(const float* X, const uint8_t* mask, const float* W1, const float* ln1_w, const float* ln1_b, const float* W2,
                const float* ln2_w, const float* ln2_b, const float* Wout, const float* bout, float* out, int32_t B, int32_t T,
                int32_t C, int32_t NO, float scale, int32_t chain, void* stream);

/* MaskedConv1D(k=3, pad=1, no bias) on token-major (B*T, Cin) rows; W is the PyTorch (N, Cin, 3) weight. */
int dcf_op_conv3(const float* X, const uint8_t* mask, const float* W_ock, float* Y, int32_t B, int32_t T, int32_t Cin,
                 int32_t N, void* stream);
/* the same convolution on the split-operand matrix-core path the forward uses (nterms 16 = f16x3, 6 = bf16x6); Cin % 32 == 0.
 * ABI version 4. */
int dcf_op_conv3_split(const float* X, const uint8_t* mask, const float* W_ock, float* Y, int32_t B, int32_t T, int32_t Cin,
                       int32_t N, int32_t nterms, void* stream);
/* channel LayerNorm (libs/modeling/blocks.py:125-131) per row; w/b may be NULL. */
int dcf_op_layernorm(const float* X, const float* w, const float* b, float* Y, int32_t rows, int32_t C, int32_t relu,
                     void* stream);
/* cross-attention core (libs/modeling/blocks.py:374-389): Q (B*T, C), K/V (B*Lk, C), kvmask (B*Lk) -> O (B*T, C) */
int dcf_op_xattn(const float* Q, const float* K, const float* V, const uint8_t* kvmask, float* O, int32_t B, int32_t T,
                 int32_t Lk, int32_t C, int32_t heads, void* stream);
/* sliding-window attention core (blocks.py:204-325,357-373): Q/K/V (B*T, C), mask (B*T) -> O; window = 0: global attention over
 * every valid key of the sequence (blocks.py:339-356, :374-393; head dimension 32 or 64) */
int dcf_op_local_attn(const float* Q, const float* K, const float* V, const uint8_t* mask, float* O, int32_t B, int32_t T,
                      int32_t C, int32_t heads, int32_t window, void* stream);
/* sidekick scoring (model.py:500-505): shallow (D, T) channel-major, text_cls (nq, D) -> correl (nq, T) */
int dcf_op_sidekick(const float* shallow, const float* text_cls, float* correl, int32_t D, int32_t T, int32_t nq,
                    int32_t norm, void* stream);
/* block top-k gate (model.py:531-541) for nq queries: correl (nq, T), vid_mask (T) -> gate (nq, T) fp32 0/1 */
int dcf_op_gate(const float* correl, const uint8_t* vid_mask, float* gate, uint8_t* mask_out, int32_t T, int32_t nq,
                int32_t sn, double sratio, int32_t msf, void* stream);

/* --------------------------------------------------------------------------------------------
 * Composite blocks on a SCRATCH model (ABI version 4): dcf_model_create(cfg) + dcf_model_bind of the block's parameters
 * under `prefix` (reference state_dict names below it), never finalized.  They exist so that the reference's operator
 * fixtures (tests/golden/ops.npz) run through the very kernels and launch sequences of the forward.  Token-major rows.
 *   dcf_op_encoder : TransformerEncoder.forward (libs/modeling/blocks.py:578-591, ConvAttNLayer :462-473), stride 1 or 2;
 *                    X (B*T, E), mask (B*T) -> Y (B*T/stride, E), mask_out (B*T/stride).  Uses cfg E, vid_heads, win, gemm_mode.
 *   dcf_op_enc_pre : its front half alone -- ln_attn, the three depthwise k3 convolutions (blocks.py:63-106, groups = E),
 *                    their LayerNorms and, for stride 2, masked_max_pool1d (blocks.py:31-47) -> Qc, Kc, Vc, Skip (B*T/stride, E).
 *   dcf_op_decoder : TransformerDecoder.forward (blocks.py:632-650, ConvXAttNLayer :513-520, adaln) in place on X (B*T, E);
 *                    text[b] (TE, len_b) channel-major, text_mask[b] (len_b) or NULL.  Uses cfg E, TE, fusion_heads, gemm_mode.
 *   dcf_op_tcn     : TCN.forward (libs/modeling/tcn.py:66-84) on x (B*T, n_in) -> Y (B*T, 32).
 * ------------------------------------------------------------------------------------------ */
int dcf_op_encoder(dcf_model* m, const char* prefix, const float* X, const uint8_t* mask, int32_t B, int32_t T, int32_t stride,
                   float* Y, uint8_t* mask_out, void* stream);
int dcf_op_enc_pre(dcf_model* m, const char* prefix, const float* X, const uint8_t* mask, int32_t B, int32_t T, int32_t stride,
                   float* Qc, float* Kc, float* Vc, float* Skip, void* stream);
int dcf_op_decoder(dcf_model* m, const char* prefix, float* X, const uint8_t* mask, int32_t B, int32_t T,
                   const float* const* text, const uint8_t* const* text_mask, const int32_t* text_len, void* stream);
int dcf_op_tcn(dcf_model* m, const char* prefix, const float* x, const uint8_t* mask, int32_t B, int32_t T, int32_t n_in,
               int32_t n_layers, float* Y, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DECAFNET_HIP_H */
