/* hulc_hip.h — C ABI of libhulc_hip.so: the MI355X-native (gfx950) HULC / GCBC training step.
 *
 * The reference (lukashermann/hulc) is pure PyTorch-Lightning Python and has no FFI; this header is the boundary a
 * maintainer binds instead of the bodies of the hot-path functions listed below.  Plain pointers
 * and sizes only, no torch types.  All pointers are DEVICE pointers unless a parameter says "host".  Every function
 * returns 0 on success; on failure it returns non-zero and hulc_last_error() holds a message.  One context per
 * process / GPU, not thread-safe; all kernels are enqueued on the context's stream (hulc_set_stream) and the calls
 * are asynchronous w.r.t. the host except where noted.
 *
 * Reference interfaces replaced (file:line in /root/reference):
 *   hulc_forward_loss  <- Hulc.training_step per-modality body   hulc/models/hulc.py:433-469  (GCBC: gcbc.py:91-136)
 *                         = ConcatEncoders.forward                perceptual_encoders/concat_encoders.py:59-109
 *                         + Visual/LanguageGoalEncoder.forward    encoders/goal_encoders.py:31-36 / 64-69
 *                         + Hulc.lmp_train                        hulc/models/hulc.py:254-299
 *                         + LogisticDecoderRNN.loss               decoders/logistic_decoder_rnn.py:121-134
 *                         + Hulc.compute_kl_loss                  hulc/models/hulc.py:539-561
 *                         + Hulc.clip_auxiliary_loss              hulc/models/hulc.py:650-695
 *   hulc_forward_loss_pair <- the same for the 'vis' and the 'lang' modality of one step in a single pass (hulc.py:433-469 loops over them)
 *   hulc_backward      <- autograd backward of the above (Lightning: loss.backward())
 *   hulc_adam_step     <- torch.optim.Adam.step                   hulc/models/hulc.py:239-252, conf/model/optimizer/adam.yaml
 *   hulc_zero_grads    <- optimizer.zero_grad()
 *   hulc_scaler_*      <- torch.cuda.amp.GradScaler (Lightning NativeMixedPrecisionPlugin at precision: 16, conf/trainer/play_trainer.yaml:3)
 *   hulc_backward_allreduce / hulc_allreduce_grads <- DDPStrategy's gradient all-reduce   hulc/training.py:64-69  (RCCL, in the library)
 */
#ifndef HULC_HIP_H
#define HULC_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hulc_ctx hulc_ctx;

enum { HULC_KIND_HULC = 0, HULC_KIND_GCBC = 1, HULC_KIND_MCIL = 2, HULC_KIND_MCIL_GRU = 3 };   /* MCIL_GRU: conf/model/mcil.yaml with plan_recognition.rnn_type=nn.GRU */
enum { HULC_DTYPE_F32 = 0, HULC_DTYPE_BF16 = 1, HULC_DTYPE_F16 = 2 };   /* F16: IEEE half operands + dynamic loss scaling (hulc_scaler_*), the reference's `precision: 16` */

typedef struct hulc_config {
    int32_t kind;            /* HULC_KIND_*  (conf/model/hulc.yaml:16, gcbc.yaml:16, mcil.yaml:16) */
    int32_t dtype;           /* HULC_DTYPE_* : storage/MFMA operand type; accumulation is always fp32 */
    int32_t max_batch;       /* largest per-modality batch B the workspace is sized for */
    int32_t max_seq;         /* largest window length S (<= 64) */
    int32_t max_window;      /* rows of plan_recognition.position_embeddings (>= max_seq) */
    int32_t use_clip;        /* use_clip_auxiliary_loss (conf/model/hulc.yaml:28) */
    float kl_beta;           /* conf/loss/default.yaml:1 */
    float kl_balancing_mix;  /* conf/loss/default.yaml:3 */
    float dropout_p;         /* plan_recognition dropout_p (transformers.yaml:9); 0 = eval-mode parity */
    int32_t num_classes;     /* action_decoder.num_classes (hulc_default.yaml:11) */
    float gripper_alpha;     /* hulc_default.yaml:14 */
    float log_scale_min;     /* hulc_default.yaml:5 */
    uint64_t seed;           /* base seed of the counter RNG (dropout masks, plan sample) */
} hulc_config;

typedef struct hulc_batch {
    int32_t B, S;                /* windows, window length */
    int32_t is_lang;             /* 0: 'vis' modality (visual goal = emb[:, -1]); 1: 'lang' modality */
    const void* rgb_static;      /* (B,S,3,200,200) fp32 NCHW in [-1,1]   hulc.py:398   (or uint8 (B,S,200,200,3), see frames_u8) */
    const void* rgb_gripper;     /* (B,S,3,84,84)   fp32 NCHW                           (or uint8 (B,S,84,84,3)) */
    const float* actions;        /* (B,S,7) relative actions, last = +-1 gripper */
    const float* robot_obs;      /* (B,S,15) state_info.robot_obs (raw; euler angles in [3:6]) */
    const float* lang;           /* (B,384) language embedding, lang modality only */
    const int32_t* plan_idx;     /* optional (B,32) injected categorical sample (parity tests); NULL = sample on device */
    const int32_t* aux_rows;     /* HOST: indices b with use_for_aux_lang_loss[b] != 0 (lang modality, clip loss) */
    int32_t n_aux;
    uint64_t step;               /* optimizer step index, mixed into the dropout / sampling seeds */
    /* ---- uint8 ingest (zero-initialise for the reference's fp32 boundary) ------------------------------
     * frames_u8 != 0: rgb_static / rgb_gripper point to uint8 (B,S,H,W,C) frames as stored in the dataset; the dataloader transforms of
     * conf/datamodule/transforms/rand_shift.yaml — ScaleImageTensor (x/255), Normalize(0.5, 0.5) and RandomShiftsAug
     * (hulc/utils/transforms.py:8-29; pad 10 / 4) — are applied inside conv1's load path (bf16 mode) instead of on the CPU.
     * shift_*: (B*S,2) int32 (sx, sy) in [0, 2*pad] per frame, device memory; NULL = no augmentation (the validation transforms).
     * With shifts, pad_* must lie in [0, 16] (the replicate margin the kernels stage; a larger pad is an error); a shift outside [0, 2*pad] is clamped. */
    int32_t frames_u8;
    int32_t pad_static, pad_gripper;
    const int32_t* shift_static;
    const int32_t* shift_gripper;
    /* ---- HULC_KIND_MCIL (conf/model/mcil.yaml): optional (B,256) fp32 injected N(0,1) draw of pr_dist.rsample() (hulc.py:289),
     * device or host; NULL = drawn on device */
    const float* plan_eps;
    /* ---- absolute-action ingest (SURVEY.md §2 "next (f)" data-side row): actions_absolute != 0 -> `actions` holds ABSOLUTE tcp targets
     * (x,y,z, euler x,y,z, gripper) and the dataloader transform RelativeActions (hulc/utils/transforms.py:32-56) is applied on the
     * device against robot_obs[..., 0:6]: clip(pos - obs, +-max_rel_pos) / max_rel_pos, wrapped angle difference clipped to
     * +-max_rel_orn and scaled, gripper unchanged.  0 = the reference's boundary (relative actions already in `actions`). */
    int32_t actions_absolute;
    float max_rel_pos, max_rel_orn;
    /* ---- HBM-resident frame store (zero-initialise for a materialised batch) ----------------------------------
     * window_start != NULL (requires frames_u8): rgb_static / rgb_gripper do not hold this batch's (B,S,H,W,C) frames but a device-resident STORE
     * of `store_frames` uint8 (H,W,C) frames — whole episodes, uploaded once (CALVIN's ~2.4 M frames of both cameras are 340 GB as uint8: a split
     * per GPU of the node fits its 288 GB) — and window b is the S consecutive store frames [window_start[b], window_start[b] + S).  conv1's forward
     * and weight gradient gather their bands by index: no (B,S,H,W,C) tensor is materialised on either side of PCIe and nothing but B indices
     * (+ actions / robot_obs / shifts) crosses it per step.  This replaces the reference's host-side shared-memory frame cache
     * (README.md:85-86: ~20 minutes to fill; dataset/README.md:55-56) and its per-step uint8 -> fp32 -> H2D path.
     * window_start: (B) int64 on the DEVICE; a start outside [0, store_frames - S] is clamped by the kernels (never an out-of-bounds read).
     * shift_* stay (B*S,2) per batch frame.  The step is bit-identical to the same windows passed as a materialised uint8 batch. */
    const int64_t* window_start;
    int64_t store_frames;
} hulc_batch;

/* out_losses (device or host pointer, see `losses_on_host`): [total_mod, kl_scaled, action, clip] of this modality,
 * unweighted: total_mod = action + kl (hulc.py:297); clip is the raw contrastive loss (hulc.py:692). */
#define HULC_N_LOSSES 4

const char* hulc_last_error(void);
int hulc_ctx_create(const hulc_config* cfg, hulc_ctx** out);
int hulc_ctx_destroy(hulc_ctx* ctx);
int hulc_set_stream(hulc_ctx* ctx, void* hip_stream);
int64_t hulc_workspace_bytes(const hulc_ctx* ctx);

/* Borrow the flat fp32 parameter / gradient / Adam-moment buffers (numel elements each) and the table of tensors inside
 * them.  names are the reference state_dict keys; offsets are element offsets (multiples of 4).  Tensors may be packed
 * tightly or padded (hulc_amd.spec pads every tensor to 64 elements); the library never writes an element outside the listed tensors except
 * ONE padding element behind a perceptual_encoder.* tensor, if the layout has one, while a data-parallel all-reduce is in flight (the job-wide
 * skip vote of a failed persistent recurrence rides the gradients' own SUM; a layout without padding votes through a 4-byte all-reduce of its own). */
int hulc_bind_params(hulc_ctx* ctx, float* params, float* grads, float* adam_m, float* adam_v, int64_t numel,
                     int32_t n_tensors, const char* const* names, const int64_t* offsets, const int64_t* numels);
/* Refresh the compute-precision / packed / transposed weight copies after the fp32 parameters changed. */
int hulc_prepare_weights(hulc_ctx* ctx);
/* Post-condition (16-bit engines, option "lazy_zero_grads" = 1, the default): every gradient element is zero EXCEPT the large Linear weight
 * gradients whose writers can all store instead of accumulate (plan_proposal / goal-encoder MLP weights, plan_recognition.fc_state, the decoder's
 * weight_hh_l0 / weight_ih_l1 / weight_hh_l1): those keep the PREVIOUS step's values until their writer of the next backward stores, or until the
 * library itself is about to read them (a bucket's all-reduce, the end of hulc_backward / hulc_backward_part, an optimizer step), which zeroes
 * whatever is still stale.  After hulc_backward the buffer is exactly what a full memset + accumulate would have produced.  A caller that READS
 * or reduces the bound gradient buffer itself between hulc_zero_grads and the end of the next backward calls hulc_flush_grads first (or sets
 * hulc_set_option(ctx, "lazy_zero_grads", 0) once: hulc_zero_grads is then the plain memset). */
int hulc_zero_grads(hulc_ctx* ctx);
/* Zero every gradient tensor that is still marked stale (see above); a no-op when nothing is.  Enqueued on the context's stream. */
int hulc_flush_grads(hulc_ctx* ctx);

/* Forward + loss of ONE modality batch; keeps activations for hulc_backward.  loss_weight = 1/len(batch) (hulc.py:491),
 * clip_weight = clip_auxiliary_loss_beta (hulc.py:525) — used to scale the gradients in hulc_backward. */
int hulc_forward_loss(hulc_ctx* ctx, const hulc_batch* batch, float loss_weight, float clip_weight, float* out_losses,
                      int32_t losses_on_host);
/* Both modalities of one step (Hulc.training_step iterates 'vis' then 'lang', hulc.py:433-469) as ONE pass over vis->B + lang->B windows:
 * the perceptual encoders, plan networks and the action decoder are shared, only the goal encoder and the CLIP rows differ, so the
 * latency-bound part of the step runs once at twice the batch instead of twice.  Requires vis->B == lang->B, equal S and ingest options;
 * loss_weight is the per-modality weight (1/2).  out_losses[8] = [total_mod, kl_scaled, action, clip] of vis, then of lang.  The result
 * (losses and accumulated gradients) equals hulc_forward_loss + hulc_backward on vis followed by the same on lang, up to summation
 * order; hulc_backward / hulc_backward_part then run once for the pair. */
int hulc_forward_loss_pair(hulc_ctx* ctx, const hulc_batch* vis, const hulc_batch* lang, float loss_weight, float clip_weight, float* out_losses,
                           int32_t losses_on_host);
/* Backward of the last hulc_forward_loss; ACCUMULATES into the bound gradient buffer. */
int hulc_backward(hulc_ctx* ctx);
/* The same in two halves, so the host can start the gradient all-reduce of the 98 % of the parameters that are finished first:
 * part 0 = everything except the perceptual encoders (decoder, plan networks, goal encoders, CLIP head); part 1 = the encoders.
 * After part 0 the gradients of every non-encoder tensor (flat offsets >= the first `plan_proposal.*` tensor) are final. */
int hulc_backward_part(hulc_ctx* ctx, int32_t part);
/* Adam over the whole flat buffer; grad_scale (e.g. 1/world_size) is folded in. step counts from 1. */
int hulc_adam_step(hulc_ctx* ctx, float lr, float beta1, float beta2, float eps, int64_t step, float grad_scale);
/* The other optimizers the reference's conf tree ships (conf/model/optimizer/{adam,adamw,sgd}.yaml, instantiated by
 * hulc/models/hulc.py:239-240), same single pass over the flat buffers and the same fp16 loss-scaler handling as hulc_adam_step:
 *   HULC_OPT_ADAM   torch.optim.Adam   (weight_decay = L2: g += wd * p)         — hulc_adam_step == this with weight_decay 0
 *   HULC_OPT_ADAMW  torch.optim.AdamW  (decoupled: p *= 1 - lr * wd, then Adam)
 *   HULC_OPT_SGD    torch.optim.SGD    (g += wd * p; buf = momentum * buf + (1 - dampening) * g, buf = g on step 1; nesterov: g + momentum * buf);
 *                   the momentum buffer is the bound `adam_m` buffer, `adam_v` is untouched.
 * lr is a per-call argument: the warm-up schedules (conf/model/lr_scheduler/*.yaml, hulc.py:218-252) are host-side arithmetic. */
enum { HULC_OPT_ADAM = 0, HULC_OPT_ADAMW = 1, HULC_OPT_SGD = 2 };
typedef struct hulc_optim {
    int32_t kind;
    float lr, beta1, beta2, eps, weight_decay;
    float momentum, dampening;
    int32_t nesterov;
    int64_t step;                /* counts from 1 */
    float grad_scale;            /* e.g. 1 / world_size */
} hulc_optim;
int hulc_optimizer_step(hulc_ctx* ctx, const hulc_optim* opt);

/* ---- Data-parallel gradient all-reduce, owned by the library (replaces Lightning's DDPStrategy, hulc/training.py:64-69: mean of the
 * per-rank gradients; the 1/world factor is hulc_adam_step's grad_scale).  RCCL over xGMI on a private high-priority stream, ordered
 * against the context's stream by events only (no host synchronisation).  One process per GPU:
 *   rank 0: hulc_comm_unique_id(id) -> the host distributes the 128 bytes (any store / torch.distributed) -> every rank: hulc_comm_init.
 * hulc_backward_allreduce = hulc_backward of the step's LAST forward with the SUM all-reduce overlapped: the flat gradient buffer is
 *   reduced in module-group buckets in reverse-forward order — [action_decoder.* .. end of buffer] (61 MB fp32), plan_proposal.*,
 *   plan_recognition.*, visual_goal.* + language_goal.*, perceptual_encoder.* — each issued as soon as the backward stage that finalises
 *   it has been enqueued, so the wire is busy while the rest of the backward (the encoder convolutions last) still computes.  Whatever is
 *   enqueued on the context's stream afterwards (hulc_adam_step) waits for the last collective.
 * hulc_allreduce_grads = one all-reduce of the whole buffer after a finished hulc_backward (no overlap).
 * bucket_dtype: HULC_DTYPE_F32 (the reference's fp32 gradients) or the engine's 16-bit type (HULC_DTYPE_BF16 for a bf16 context,
 *   HULC_DTYPE_F16 for an fp16 one): the gradients cross the wire in 16 bits (half the bytes: xGMI rings are per-link bound), are summed
 *   by RCCL in that type and widened back.  hulc_comm_buckets returns the bucket ranges (element offsets) in issue order. */
int hulc_comm_unique_id(void* out_host /*128 bytes*/, int64_t cap);
/* Optional first phase of hulc_comm_init: resolves RCCL and creates the private stream — everything that can fail on one rank alone.  A
 * host that calls it on every rank and agrees on the result BEFORE any rank calls hulc_comm_init (which blocks in ncclCommInitRank until all
 * ranks arrive) cannot be deadlocked by a rank-local failure. */
int hulc_comm_prepare(hulc_ctx* ctx);
int hulc_comm_init(hulc_ctx* ctx, const void* unique_id_host, int32_t rank, int32_t world);
int hulc_comm_destroy(hulc_ctx* ctx);
int hulc_comm_buckets(hulc_ctx* ctx, int64_t* lo, int64_t* hi, int32_t cap);      /* returns the number of buckets, < 0 on error */
int hulc_comm_stats(hulc_ctx* ctx, int64_t* n_collectives, double* bytes_on_wire_per_rank);
/* Rank and size of the LIVE communicator as RCCL reports them (ncclCommUserRank / ncclCommCount) — not an echo of hulc_comm_init's arguments. */
int hulc_comm_size(hulc_ctx* ctx, int32_t* rank, int32_t* world);
/* With hulc_set_option(ctx, "comm_timing", 1): the LAST hulc_backward_allreduce's buckets as seen by events on the collectives' stream.
 * out[4 i .. 4 i + 3] = {start_us, end_us of bucket i's collective relative to the END of the backward on the context's stream (negative =
 * that much of it ran hidden under the backward), bytes on the wire per rank, bucket index}; *backward_us = the backward's own duration.
 * Synchronises both streams.  Returns the number of buckets written (<= cap_buckets), < 0 on error. */
int hulc_comm_timeline(hulc_ctx* ctx, double* out, int32_t cap_buckets, double* backward_us);
int hulc_allreduce_grads(hulc_ctx* ctx, int32_t bucket_dtype);
int hulc_backward_allreduce(hulc_ctx* ctx, int32_t bucket_dtype);

/* ---- fp16 mode (HULC_DTYPE_F16): dynamic loss scaling = torch.cuda.amp.GradScaler, which Lightning's native-AMP plugin drives for the
 * reference's `precision: 16` (conf/trainer/play_trainer.yaml:3; fp16 autocast for matmuls/convs, fp32 for softmax / LayerNorm / losses
 * and the fp32 island of decoders/utils/gripper_control.py:17,40 — the same split this library uses).  The scaler state lives on the
 * device: the loss kernels multiply the loss gradient by `scale`, hulc_adam_step checks the (all-reduced) gradient buffer for
 * non-finite values, divides the scale out, SKIPS the parameter update on inf/nan (its bias corrections count taken steps only, the
 * `step` argument is then ignored) and updates the scale: *= backoff_factor on a skipped step, *= growth_factor after
 * growth_interval consecutive good steps (torch/amp/grad_scaler.py).  The bound gradient buffer holds SCALED gradients.
 * An F16 context starts with GradScaler's defaults (65536, 2, 0.5, 2000); init_scale <= 0 turns scaling off; F32 / BF16 contexts
 * start with it off.  hulc_scaler_get synchronises (checkpointing: Lightning stores the scaler state next to the optimizer's). */
int hulc_scaler_enable(hulc_ctx* ctx, float init_scale, float growth_factor, float backoff_factor, int32_t growth_interval);
int hulc_scaler_get(hulc_ctx* ctx, float* scale, int32_t* growth_tracker, int64_t* skipped_steps, int32_t* last_found_inf,
                    int64_t* taken_steps /* each optional */);
/* taken_steps = optimizer steps actually taken (not skipped): Adam's bias corrections run on this device-side count in fp16 mode, exactly as
 * torch's optimizer `step` state only advances when GradScaler.step() lets optimizer.step() run.  It is part of a checkpoint: resuming with
 * warm moments and a zero count would shrink the first thousands of updates.  taken_steps < 0 leaves the count unchanged. */
int hulc_scaler_set(hulc_ctx* ctx, float scale, int32_t growth_tracker, int64_t taken_steps);

/* ---- Validation forward: one modality of Hulc.validation_step (hulc/models/hulc.py:770-797) = lmp_val
 * (:301-388): plan proposal and plan recognition each sample a plan (distributions.py:37-41), the decoder is run with both
 * (LogisticDecoderRNN.loss_and_act, logistic_decoder_rnn.py:85-100): NLL loss, a sampled action (_sample :234-258) mapped back to
 * the world frame (tcp_to_world_frame, gripper_control.py:39-63), its mean absolute error and the gripper success rate.
 * Eval-mode: dropout off.  Forward only — it invalidates the state hulc_backward needs.
 * Every pointer of hulc_val_noise is optional (device or host memory); NULL = draw on the device with the counter RNG. */
/* HULC_KIND_MCIL: the plans are continuous — plan_idx_pp / plan_idx_pr (here, in hulc_validate's plan_idx_*_out and in
 * hulc_rollout_plan's plan_inject / plan_out) then point to (B,256) fp32 sampled plans instead of (B,32) int32 category indices,
 * u_mix_* is (B,S,7,10) and u_act_* (B,S,7) (7 mixture dimensions, no gripper head). */
typedef struct hulc_val_noise {
    const int32_t* plan_idx_pp;  /* (B,32) categorical sample of the plan-proposal distribution */
    const int32_t* plan_idx_pr;  /* (B,32) ... of the plan-recognition distribution */
    const float* u_mix_pp;       /* (B,S,6,10) torch.rand draw selecting the mixture component (Gumbel argmax), pp pass */
    const float* u_act_pp;       /* (B,S,6)    torch.rand draw of the logistic inversion sampling, pp pass */
    const float* u_mix_pr;
    const float* u_act_pr;
} hulc_val_noise;
/* out_host[18] = {action_loss_pp, action_loss_pr, kl_loss (beta-scaled, hulc.py:539-561), gripper_sr_pp, gripper_sr_pr,
 *                 mae_pp[6], mae_pr[6], val_pred_clip_loss}  (per-dimension means over B and S: everything validation_step logs,
 *                 :798-833; the CLIP loss (:804-808) only for a lang batch with use_clip and batch->n_aux > 0, else 0).
 * plan_idx_*_out (B,32) int32 and pred_*_out (B,S,7) world-frame sampled actions: optional, device or host. */
#define HULC_N_VAL 18
int hulc_validate(hulc_ctx* ctx, const hulc_batch* batch, const hulc_val_noise* noise, float* out_host, int32_t* plan_idx_pp_out,
                  int32_t* plan_idx_pr_out, float* pred_pp_out, float* pred_pr_out);

/* ---- CLIP ground-truth validation metric (Hulc.on_validation_epoch_start, hulc/models/hulc.py:967-974, and the device part of
 * Hulc._clip_groundtruth_loss, :1024-1029).  hulc_clip_gt_encode runs language_goal (goal_encoders.py:64-69) and proj_vis_lang.mlp_lang
 * (proj_vis_lang.py) on m instruction embeddings (m,384) fp32, host or device, and keeps the (m,32) projections in `slot`
 * (HULC_GT_TRAIN = encoded_lang_train, HULC_GT_VAL = encoded_lang_val) until the slot is encoded again; call it at the start of every
 * validation epoch (the weights have moved).  hulc_clip_gt_scores returns logits_per_image (n,m) row-major on the host:
 * exp(logit_scale) * <image projection of masked row i of the LAST hulc_validate (lang batch, use_for_aux rows in aux_rows order),
 * projection j of the slot>, both L2-normalised.  It fails when that validate had no masked rows (the reference returns early, :988-989)
 * or the context has no CLIP head.  The per-row min-max normalisation and the task bookkeeping (:1031-1043) are host logic
 * (hulc_amd/hulc.py Hulc.clip_groundtruth).  scores_host == NULL with cap_floats == 0 is a shape query: *n_out / *m_out are filled and 0 is returned (they are also filled when the buffer is too small). */
#define HULC_GT_TRAIN 0
#define HULC_GT_VAL 1
int hulc_clip_gt_encode(hulc_ctx* ctx, const float* lang_emb, int32_t m, int32_t slot);
int hulc_clip_gt_scores(hulc_ctx* ctx, int32_t slot, float* scores_host, int64_t cap_floats, int32_t* n_out, int32_t* m_out);

/* ---- Rollout (Hulc.reset / step, hulc/models/hulc.py:843-957; stateful LogisticDecoderRNN.act, logistic_decoder_rnn.py:102-116).
 * hulc_rollout_plan  = get_pp_plan_vision (:905-927: obs and goal frame encoded as one 2-frame window) or get_pp_plan_lang
 *                      (:929-948): latent goal + a plan sampled from the plan proposal; clears the decoder's hidden state.
 * hulc_rollout_act   = predict_with_plan (:881-903): encode the current frame, one recurrent step, sample, tcp -> world.
 * The replan_freq counter lives in the host wrapper (hulc_amd.hulc.Hulc.step).  B = 1.
 * HULC_KIND_GCBC (GCBC.reset / step, hulc/models/gcbc.py:281-320): hulc_rollout_plan only encodes the latent goal (once per rollout,
 * plan_idx_* ignored), hulc_rollout_act runs the decoder without a plan. */
typedef struct hulc_rollout_obs {
    const float* rgb_static;     /* (1,1,3,200,200) device */
    const float* rgb_gripper;    /* (1,1,3,84,84)   device */
    const float* robot_obs_raw;  /* (15) device or host: raw proprioception, euler angles in [3:6] */
} hulc_rollout_obs;
int hulc_rollout_reset(hulc_ctx* ctx);
int hulc_rollout_plan(hulc_ctx* ctx, const hulc_rollout_obs* obs, const float* goal_rgb_static, const float* goal_rgb_gripper,
                      const float* goal_lang /* (384) device; exactly one of goal images / goal_lang */,
                      const int32_t* plan_idx_inject /* (32) or NULL */, int32_t* plan_idx_out /* (32) host or device, optional */);
int hulc_rollout_act(hulc_ctx* ctx, const hulc_rollout_obs* obs, const float* u_mix /* (6,10) or NULL */,
                     const float* u_act /* (6) or NULL */, float* action_out_host /* (7) */);
/* The reference's three public inference methods take and return the plan and the latent goal as VALUES (hulc.py:881-948; called by name
 * from hulc/evaluation/rollouts_interactive.py:158-164): hulc_rollout_get_goal reads the latent goal the last hulc_rollout_plan encoded
 * ((32) fp32, the second return value of get_pp_plan_vision / get_pp_plan_lang); hulc_rollout_set_state installs a caller-held plan
 * ((32) int32 category indices, (256) fp32 for HULC_KIND_MCIL, ignored / NULL for HULC_KIND_GCBC) and latent goal ((32) fp32) for the
 * following hulc_rollout_act calls = the `latent_goal, sampled_plan` arguments of predict_with_plan (:881-903).  Neither touches the
 * decoder's hidden state.  Pointers may be host or device memory. */
int hulc_rollout_get_goal(hulc_ctx* ctx, float* latent_goal_out /* (32) */);
int hulc_rollout_set_state(hulc_ctx* ctx, const void* plan, const float* latent_goal /* (32) */);

/* ---- MiniLM sentence encoder: the reference's SBert (hulc/models/encoders/language_network.py:8-17) =
 * sentence_transformers "all-MiniLM-L6-v2" (conf/model/sbert.yaml:2) = BERT encoder + masked mean pooling + L2 normalisation.
 * Forward only, fp32.  Weights: one flat fp32 device buffer + Hugging Face BertModel state_dict names (e.g.
 * "encoder.layer.0.attention.self.query.weight") with element offsets, like hulc_bind_params.  Token ids come from the host-side
 * WordPiece tokenizer (hulc_amd/sbert.py).  Defaults of all-MiniLM-L6-v2: layers 6, hidden 384, heads 12, intermediate 1536,
 * vocab 30522, max_position 512, ln_eps 1e-12, normalize 1. */
typedef struct hulc_sbert_config {
    int32_t layers, hidden, heads, intermediate, vocab, max_position;
    int32_t max_sentences, max_tokens;     /* workspace: sentences per call, tokens per sentence (<= 128) */
    int32_t normalize;                     /* 1: L2-normalise the pooled embedding */
    float ln_eps;
} hulc_sbert_config;
typedef struct hulc_sbert hulc_sbert;
int hulc_sbert_create(const hulc_sbert_config* cfg, hulc_sbert** out);
int hulc_sbert_destroy(hulc_sbert* ctx);
int hulc_sbert_set_stream(hulc_sbert* ctx, void* hip_stream);
int hulc_sbert_bind(hulc_sbert* ctx, const float* flat_params, int64_t numel, int32_t n_tensors, const char* const* names,
                    const int64_t* offsets, const int64_t* numels);
/* token_ids / attention_mask: (B,L) int32, device or host; out: (B,hidden) fp32, device or host.  Synchronises the stream. */
int hulc_sbert_encode(hulc_sbert* ctx, const int32_t* token_ids, const int32_t* attention_mask, int32_t B, int32_t L, float* out);

/* Runtime knobs: kl_beta (Hulc.set_kl_beta, hulc/models/hulc.py:563-565; called by hulc/utils/kl_callbacks.py:19-22) and the
 * transformer dropout probability (module.train()/eval(): 0 in eval mode). Take effect from the next hulc_forward_loss. */
int hulc_set_kl_beta(hulc_ctx* ctx, float kl_beta);
int hulc_set_dropout(hulc_ctx* ctx, float p);

/* Runtime options of a context.  "persistent_rnn" (default 1): the 2048-wide recurrences (action decoder nn.RNN,
 * hulc/models/decoders/utils/rnn.py:5-14; mcil's nn.RNN plan encoder, plan_recognition_net.py:12-42) run as ONE persistent launch per layer
 * and direction in the 16-bit engines (csrc/rnn_persist.h) — it needs every CU of the GPU; 0 = one launch per time step (choose this when
 * several processes share one GPU).  "fused_transformer" (default 1): one launch per plan-recognition encoder layer in the forward of the
 * 16-bit engines (csrc/tr_fused.h; plan_recognition_net.py:98-117; windows of up to 64 rows), 0 = the unfused kernels.  "timer_event_fence"
 * (default 0): 1 = the class timers below use default (system-fenced) HIP events instead of timing-only ones.  Returns non-zero for an unknown name. */
int hulc_set_option(hulc_ctx* ctx, const char* name, int64_t value);
/* Reads an option back.  Besides the settable names: "persistent_rnn" reports the EFFECTIVE state (0 once a persistent launch failed its census or
 * timed out — the context then runs one launch per time step; a failed launch never reaches the weights: its optimizer step skips itself on the
 * device, and calls that end in a synchronisation are run again), "persistent_rnn_fallbacks" counts such events.  More settable names:
 * "persist_under_comm" (default 0: recurrences that follow an issued bucket of hulc_backward_allreduce run one launch per step, because RCCL's
 * kernels hold CUs), "comm_timing" (hulc_comm_timeline), "debug_poison_partials" (tests). */
int hulc_get_option(hulc_ctx* ctx, const char* name, int64_t* value);

/* HIP-event timers around the launches of each kernel class, recorded on the context's stream (bench.py's roofline leg).
 * hulc_timers_read synchronises and writes a JSON object {"class": {"bound","launches","ms","flops","bytes"}} (algorithmic
 * FLOPs / bytes summed over the timed launches). */
int hulc_timers_enable(hulc_ctx* ctx, int32_t on, const char* only_class /* NULL or "" = every class */);
int hulc_timers_read(hulc_ctx* ctx, char* json_out, int64_t cap, int32_t reset);

/* Inspection (tests): copy a named internal tensor to HOST fp32. Returns element count in *n (cap = capacity). */
int hulc_get_tensor(hulc_ctx* ctx, const char* name, float* host_out, int64_t cap, int64_t* n);
int hulc_get_plan_idx(hulc_ctx* ctx, int32_t* host_out, int64_t cap);

/* Per-kernel test entry points (device pointers; dtype selects fp32 / bf16 / fp16 storage of A, B — HULC_DTYPE_F16 for hulc_k_gemm_nt only,
 * the other half-precision entries run the bf16 build of the kernels; the fp16 build of every kernel is covered by the fp16 step tests). */
int hulc_k_gemm_nt(int32_t dtype, const void* A, const void* B, float* C, int32_t M, int32_t N, int32_t K, int64_t lda,
                   int64_t ldb, int64_t ldc, const float* bias, int32_t relu, void* hip_stream);
/* conv weight-gradient kernel alone (bf16 NHWC activations): which = 2 (4x4 s2, 32->64) or 3 (3x3 s1, 64->64); square frames of
 * side IH; out = fp32 [64][KH*KW*CI] in packed (kh,kw,ci) order, overwritten. Synchronises. */
int hulc_k_conv_wgrad(int32_t which, const void* X, const void* dY, float* out, int32_t Nf, int32_t IH, void* hip_stream);
/* conv1's weight / bias gradient from uint8 (Nf,IH,IH,3) frames + RandomShiftsAug shifts alone (bf16 engine's kernels; form 0: raw rows through LDS, 1: conversion from the prefetch registers) */
/* host arithmetic only: groups [first, first + count) of a row's IW / 4 four-pixel groups that no RandomShiftsAug shift |dx| <= pad can push against a row end
 * (at most `cap` of them): the uint8 conv1 kernels convert these without clamps; tests check the property the kernels rely on */
int hulc_k_conv1_interior_groups(int32_t IW, int32_t pad, int32_t cap, int32_t* first, int32_t* count);
int hulc_k_conv1_wgrad_u8(const void* X, const int32_t* shifts, int32_t pad, const void* dY, float* dw_out, float* db_out, int32_t Nf, int32_t IH, int32_t form, int32_t fold, void* hip_stream);
/* raw-tile conv kernels alone (bf16 NHWC, square frames): mode 0 fwd 3x3/s1 64->64, 1 fwd 4x4/s2 32->64, 2 dgrad of (0), 3 dgrad of
 * (1). img side IMH, out side OUTH; w = packed weights as produced by hulc_prepare_weights (fwd [co][(kh,kw,ci)], dgrad per-parity
 * [class][ci][(a,b,co)]); bias fp32 / mask bf16 optional. Asynchronous on hip_stream.
 * Modes 4..6: conv1 forward (fp32 NCHW / uint8 NHWC / uint8 + shifts in `mask`).  Modes 7..9 = the forms the engine runs: 7 = mode 1
 * that also emits the ReLU bitmask of its output into `mask` (uint32 [Nf][OUTH][OUTW][2]); 8 / 9 = modes 2 / 3 with `mask` = ReLU
 * bitmask words (2 / 1 per output pixel).  relu: bit 0 = ReLU, bit 5 (32) = dynamic work claiming, bits 1..4 = bench ablations. */
int hulc_k_conv_tile(int32_t mode, const void* img, const void* w, const float* bias, const void* mask, void* out, int32_t Nf, int32_t IMH,
                     int32_t OUTH, int32_t relu, void* hip_stream);
/* skinny GEMM kernel alone (bf16 in / bf16 out), variant = waves*10 + row-tiles-per-workgroup; 300 = the production router; 400 = TWO
 * independent problems in one launch (second one at A + M*K, W + N*K, out + M*N; 32 < M <= 64, K = 2048); 500 = W repacked into the
 * fragment order first (every 16-row x 32-column block = 1 KB in MFMA lane order) and read that way by the K-chunked kernel (K = n x 2048,
 * synchronous); 510 = only the repack: `out` receives the N x K fragment-ordered copy.  Asynchronous unless noted. */
int hulc_k_skinny(const void* A, const void* W, void* out, int32_t M, int32_t N, int32_t K, int32_t variant, void* hip_stream);
/* one whole recurrence X[q0 + s dq] = f(X[q0 + (s-1) dq] W^T, res, mask), s = 1..S-1, as ONE persistent launch (bf16; csrc/rnn_persist.h).
 * X, res, mask: [S][B][2048]; W [2048][2048]; mask == NULL: act 1 ReLU / 2 tanh of (product + res); mask given: (product + res) * (mask > 0)
 * (act 1) or * (1 - mask^2) (act 2).  flags: hulc_k_rnn_persist_flag_words() zero-initialised uint32 (reusable with launch_index = 1, 2, ...),
 * err: one zero-initialised uint32 (non-zero after a failed launch).  Needs all CUs of the GPU; asynchronous. */
int hulc_k_rnn_persist(void* X, const void* W, const void* res, const void* mask, int32_t B, int32_t S, int32_t q0, int32_t dq, int32_t act,
                       uint32_t* flags, uint32_t* err, uint32_t launch_index, void* hip_stream);
int32_t hulc_k_rnn_persist_flag_words(void);
int hulc_k_trread_probe(const int32_t* elem_index_per_lane /*64*/, uint16_t* out /*64x4*/, void* hip_stream);
int hulc_k_cast(int32_t dtype, const float* src, void* dst, int64_t n, void* hip_stream);
/* the transformer's attention kernels alone (8 heads of 16, fp32 storage so that the check against a float64 softmax is tight):
 * qkv [B*S][384]; forward (dao == NULL): P [B][8][S][S] (post-softmax, pre-dropout) and ao [B*S][128] are written; backward (dao [B*S][128]
 * given): reads qkv, P, dao and writes dqkv [B*S][384].  variant 0 = one lane per query row (any S <= 64), 1 = the S <= 32 kernels the
 * engine runs (two lanes per query row), 2 = its kernels for 32 < S <= 64 (four waves per head, any S <= 64).  drop_p / seed: the attention-probability dropout (mask = hash(seed, element index)). */
int hulc_k_attention(int32_t variant, const float* qkv, float* P, float* ao, const float* dao, float* dqkv, int32_t B, int32_t S,
                     float drop_p, uint64_t seed, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif
