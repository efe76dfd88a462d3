/*
 * include/pnvo.h — C ABI of the MI355X-native PointNav-VO visual-odometry hot path (libpnvo.so).
 *
 * The reference (Xiaoming-Zhao/PointNav-VO) is 100 % Python and has no FFI; the "plugin API" this path sits
 * behind is the model registry + nn.Module protocol.  Each entry point below names the
 * reference interface it replaces (paths relative to /root/reference); INTEGRATION.md shows the ctypes stub
 * and the registry hook a maintainer would add on the reference side.
 *
 * Conventions: plain C types only (no torch / HIP types in signatures; a stream is passed as void* holding a
 * hipStream_t).  Every function returns 0 (PNVO_OK) or a negative error code and never throws; the message of
 * the last error on a handle is available from pnvo_last_error().  All tensor arguments of the compute entry
 * points are DEVICE pointers owned by the caller, NHWC, float32 unless stated; calls are asynchronous on the
 * given stream and never wait for the whole forward (the one partial wait — for the stem kernel, so that inputs
 * outside the fused stems' contract are handled inside the call — is described at pnvo_check_inputs and can be
 * switched off).  A handle is not thread-safe: one handle per
 * (process, stream), one process per GPU (the reference's own model: launch.py:11-12).
 */
#ifndef PNVO_H_
#define PNVO_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PNVO_OK 0
#define PNVO_ERR_ARG (-1)      /* bad argument / unsupported configuration */
#define PNVO_ERR_HIP (-2)      /* a HIP runtime call failed (message has hipGetErrorString) */
#define PNVO_ERR_STATE (-3)    /* call order violated (e.g. forward before load_weights) */
#define PNVO_ERR_WEIGHTS (-4)  /* state_dict table does not match the configured architecture */
#define PNVO_ERR_INPUT (-5)    /* observation data violates the reference's own contract (see pnvo_check_inputs) */

typedef struct pnvo_model_s *pnvo_handle;

/*
 * Architecture of one VO model = the constructor kwargs of the registered reference classes
 * (pointnav_vo/vo/models/vo_cnn.py:183-198, called at pointnav_vo/rl/common/base_trainer_with_vo.py:68-80).
 * n_* are PAIR channel counts: 6 / 2 / 2*discretized_depth_channels / 2, or 0 when the modality is not in
 * observation_space.  ngroups is baseplanes/2 (vo_cnn.py:206); backbone is resnet18 (resnet.py:226-229).
 */
typedef struct {
  int32_t width, height;      /* observation_size = (W, H) */
  int32_t n_rgb, n_depth, n_dd, n_tdv;
  int32_t baseplanes;         /* resnet_baseplanes (32; 64 for the *_wider variants, vo_cnn.py:325) */
  int32_t hidden;             /* hidden_size */
  int32_t out_dim;            /* output_dim (3: dx, dz, dyaw) */
  int32_t normalize;          /* normalize_visual_inputs -> RunningMeanAndVar present */
  int32_t act_embed;          /* 1 for vo_cnn_act_embed variants (vo_cnn_act_embed.py:17-75) */
  int32_t n_acts;             /* embedding rows - 1 (N_ACTS = 4) */
  int32_t flat_size;          /* after_compression_flat_size (2048) */
  int32_t max_batch;          /* workspace sizing hint; the workspace grows on demand */
  int32_t backbone_depth;     /* 0 or 18: resnet18 (BasicBlock); 50 / 101: Bottleneck [3,4,6,3] / [3,4,23,3] (resnet.py:226-241) */
} pnvo_config;

/* One entry of the reference state_dict: name exactly as model.state_dict() spells it,
 * row-major data at blob[offset .. offset+prod(shape)), reference layout (OIHW convs, [N][K] linears). */
typedef struct {
  const char *name;
  uint64_t offset;            /* in floats */
  int32_t ndim;
  int64_t shape[4];
} pnvo_tensor_desc;

/* Replaces: vo_model_cls(**kwargs).to(device)  (base_trainer_with_vo.py:68-81). */
int pnvo_create(const pnvo_config *cfg, int device, pnvo_handle *out);

/* Replaces: model.load_state_dict(ckpt["model_state"])  (base_trainer_with_vo.py:92-99).
 * blob is HOST memory (copied and re-laid out for the kernels; the caller keeps ownership).  Every tensor the
 * architecture needs must be present with the reference's shape, else PNVO_ERR_WEIGHTS. */
int pnvo_load_weights(pnvo_handle h, const float *blob, size_t n_floats, const pnvo_tensor_desc *toc, int ntoc);

/*
 * Replaces: model.eval(); model(obs_pairs[, actions])  (base_trainer_with_vo.py:286-291;
 * VisualOdometryCNNBase.forward vo_cnn.py:229-233; ResNetEncoder.forward :110-179).
 *   rgb   [B,H,W,n_rgb]   values 0..255        (obs_pairs["rgb"])
 *   depth [B,H,W,n_depth] values 0..1          (obs_pairs["depth"])
 *   dd    [B,H,W,n_dd]    one-hot {0,1}        (obs_pairs["discretized_depth"])
 *   tdv   [B,H,W,n_tdv]   values 0..1          (obs_pairs["top_down_view"])
 *   actions [B] int64 (act_embed variants only, else NULL)
 *   out   [B,out_dim]
 * Pointers of absent modalities must be NULL.  Dropout is identity (eval); "rnd" mode is not provided.
 */
int pnvo_forward(pnvo_handle h, const float *rgb, const float *depth, const float *dd, const float *tdv,
                 const int64_t *actions, int B, float *out, void *stream);

/*
 * Per-handle options: kernel-family selection and behaviour switches.  The reference has no counterpart (torch dispatches its
 * kernels itself); these exist so that tests and benchmarks can A/B the kernel families and so that integrations configure
 * a handle in code, not through the process environment.  The PNVO_<KEY> environment variables of earlier rounds are only
 * the DEFAULTS, read once in pnvo_create; no entry point reads the environment afterwards.
 *   key               values (first = default)             meaning
 *   "stem"            auto | mx | dd | dense                 stem kernel: bf16 matrix cores with exact 3-piece weights | one-hot
 *                                                            table gather | all-fp32 MFMA (takes ANY float input)
 *   "conv"            auto | x3 | fp32 | generic             3x3 / strided convs: 3-piece bf16-pipe kernel for launches of >= 192
 *                                                            workgroups (auto) or always (x3) | fp32-MFMA LDS kernels | generic
 *   "x3_s2"           on | off                               stride-2 convs on the 3-piece kernel
 *   "tail", "pool"    fused | separate                       BasicBlock tails / the max-pool folded into neighbouring kernels
 *   "input_fallback"  on | off                               see pnvo_check_inputs
 *   "small_net"       on | off                               batches of <= small_max pairs (the navigation loop's call shape,
 *   "small_max"       4 (1..4)                               rl/ppo/ppo_trainer.py:836-841): everything behind the stem conv in ONE
 *   "small_coop"      1 | 0                                  persistent launch (smallnet.hip) — default models with BasicBlock
 *                                                            backbones, options conv/tail/pool at their defaults, no tap.
 *                                                            small_coop = 1: hipLaunchCooperativeKernel (all workgroups resident
 *                                                            by the runtime's guarantee); 0: plain launch, 17 us less per call,
 *                                                            for a process that has the GPU to itself (one workgroup per CU, 144
 *                                                            of 256: two such kernels started at the same instant from different
 *                                                            queues could each hold half the chip — the barrier's bounded spin
 *                                                            then leaves NaN poses and fails the handle's next call)
 *   "graph"           0 | 1                                  replay the forward from a captured hipGraph
 *   "wgrad_stem"      mx | fp32        "pool_bwd" fused | separate        "dgrad" phase | masked        (training step)
 *   "bf16_fuse"       on | off         "bf16_stem3" 0 | 1    "conv3_nt" 0 | 1   "stem_dbg" <int>            (experiments)
 * Unknown keys / values return PNVO_ERR_ARG with the accepted spellings in pnvo_last_error.
 */
int pnvo_set_option(pnvo_handle h, const char *key, const char *value);
/* Current value of an option as text ("dense (fallback)" for "stem" once the input fallback engaged). */
int pnvo_get_option(pnvo_handle h, const char *key, char *buf, size_t cap);

/*
 * The same forward from the SENSOR frames (SURVEY.md section 8(b) sketch): replaces the observation construction of
 * BaseRLTrainerWithVO._compute_local_delta_states_from_vo (base_trainer_with_vo.py:172-269) AND the model call (:286-291)
 * without materialising obs_pairs — pair concatenation, the uint8 -> float cast and _discretize_depth_func (:135-167) happen in
 * the stem's operand fetch (7.86 MB per pair of float32 observation tensors are neither written nor read).
 *   rgb_frames   device uint8   [B][2][H][W][3]  (prev frame, cur frame; NULL for models without rgb)
 *   depth_frames device float32 [B][2][H][W]     values 0..1 (feeds the depth and the discretised-depth modality)
 *   tdv          device float32 [B][H][W][2]     the two top-down views (pnvo_topdown_view with out_pstride 2), NULL without it
 *   err_flag     device int32, may be NULL: set to 1 when a depth lies outside [0, 1] (the reference asserts, :136-137)
 * Results are bit-identical to pnvo_build_obs_pairs + pnvo_forward.  Configurations the frame-reading stem does not cover
 * (option stem / pieces away from their defaults, an attached training step, models outside its K-slot layout) are served by
 * materialising the pairs into a workspace of the handle first: same results, the cost of the separate path.
 */
int pnvo_forward_raw(pnvo_handle h, const uint8_t *rgb_frames, const float *depth_frames, const float *tdv, const int64_t *actions,
                     int B, float *out, int32_t *err_flag, void *stream);
/* pnvo_forward_dual (bfloat16, two action models, the second on the swapped pair) from the sensor frames. */
int pnvo_forward_dual_raw(pnvo_handle ha, pnvo_handle hb, const uint8_t *rgb_frames, const float *depth_frames, const float *tdv, int B,
                          float *out_a, float *out_b, int32_t *err_flag, void *stream);

/*
 * GROUPED forward (round 6): the pairs of up to three SEPARATE-ACTION models in one launch chain — what the navigation loop's call
 * needs when BaseRLTrainerWithVO._compute_local_delta_states_from_vo (base_trainer_with_vo.py:277-294: vo_model[act]) is batched over
 * environments: 8-32 pairs per simulator step split over the "forward" / "left" / "right" models are three forwards of ~50 dependent
 * launches each, bound by launch latency.  The pairs are sorted by model: handles[k] serves counts[k] consecutive pairs of the frame
 * tensors (layout of pnvo_forward_raw), sum(counts) == B, n_models <= 3, a count may be 0.  Every kernel of the chain picks a pair's
 * weights, weight scales and GroupNorm affine parameters by its model; the hidden layer and head run per model on its rows.
 * Requirements (else PNVO_ERR_STATE, never a fallback): float32 inference handles of one architecture on one device, default
 * float16-piece kernels, no training step attached, no tap, not an act-embed model.  Per pair float32-grade equal (same
 * tolerances against the fp64 oracle) to the model's own pnvo_forward_raw, whose kernel choice depends on the batch size.
 * One model with a non-zero count: the call is pnvo_forward_raw on that handle.
 */
int pnvo_forward_grouped_raw(const pnvo_handle *handles, const int32_t *counts, int n_models, const uint8_t *rgb_frames,
                             const float *depth_frames, const float *tdv, int B, float *out, int32_t *err_flag, void *stream);
/* PNVO_OK when these (1..3, non-null) handles can share a grouped forward, else the error it would return with the reason in
 * pnvo_last_error(handles[0]): what a caller's dispatch between one grouped and several per-model forwards asks. */
int pnvo_grouped_supported(const pnvo_handle *handles, int n_models);

/*
 * Arithmetic of pnvo_forward for this handle: 0 = float32 (default: exact-float32 products on the matrix cores), 1 = bfloat16
 * (BASELINE config 3: bf16 operands and bf16 activations in HBM, float32 accumulation / GroupNorm statistics / Linear layers).
 * The reference has no such switch (it would be model.bfloat16(), which also rounds the normalisation and the head);
 * resnet18 BasicBlock models with 32 base planes only.  Takes effect at the next forward; no reload needed.
 */
int pnvo_set_precision(pnvo_handle h, int precision);

/*
 * Replaces: the geometric-invariance dual forward of the joint left/right training and evaluation
 *   pred_a = vo_model[act_a](batch_pairs);  pred_b = vo_model[act_b](swapped batch_pairs)
 * (pointnav_vo/vo/engine/vo_cnn_regression_geo_invariance_engine.py:569-602; the swapped (cur, prev) entries are built by
 * pointnav_vo/vo/dataset/regression_geo_invariance_iter_dataset.py:342-386).  Both models run in every launch; model b sees
 * each observation tensor with its [prev | cur] channel halves exchanged WITHOUT the caller materialising the swapped pair
 * (a permutation of b's stem weights), so the observation tensors are read once.  Both handles must have the same
 * architecture and precision 1 (bfloat16).  Tensor contract as pnvo_forward; out_a / out_b [B,out_dim].
 */
int pnvo_forward_dual(pnvo_handle ha, pnvo_handle hb, const float *rgb, const float *depth, const float *dd, const float *tdv,
                      int B, float *out_a, float *out_b, void *stream);

/* Replaces: BaseRLTrainerWithVO._discretize_depth_func (base_trainer_with_vo.py:135-167), batched and strided so
 * that it writes straight into obs_pairs["discretized_depth"]:
 *   for p in [0,n): d = depth[p*in_stride];  onehot[p*out_stride + i] = (e_i <= d < e_{i+1}) for i in [0,bins)
 * (last bin closed; e_i = float32(i/bins), :105-115).  err_flag (device int32, may be NULL) is set to 1 if any
 * value is outside [0,1] (the reference asserts, :136-137). */
int pnvo_discretize_depth(const float *depth, int64_t n, int64_t in_stride, int bins, float *onehot,
                          int64_t out_stride, int32_t *err_flag, void *stream);

/*
 * Replaces: NormalizedDepth2TopDownViewHabitatTorch.gen_top_down_view (pointnav_vo/utils/geometry_utils.py:
 * 516-556), batched over N frames with no host round trip (the reference syncs 4..1066 times per frame and
 * round-trips through cv2 on the host, :529-536,582-606).
 *   frame f, pixel p=(h*W+w): depth[f*in_fstride + p*in_pstride]  ->  out[f*out_fstride + p*out_pstride]
 *   consts[7] (HOST): kinv00, kinv02, min_x, x_den, depth_scale, z_den, min_depth — computed by the caller exactly
 *   as the reference does (torch.inverse(K), _get_x_range; see pointnav-vo_amd/trainer.py).
 *   work: device scratch of pnvo_topdown_workspace_bytes(N,H,W) bytes.
 */
size_t pnvo_topdown_workspace_bytes(int N, int H, int W);
/* Both top-down views of n_pairs (prev, cur) frame pairs in ONE pass of the three kernels (the boundary call's shape:
 * base_trainer_with_vo.py:239-249 builds them one after the other): depth_frames device float32 [n_pairs][2][H][W] ->
 * tdv_pairs [n_pairs][H][W][2] (channel 0 = prev frame).  work >= pnvo_topdown_workspace_bytes(2 * n_pairs, H, W).  Same values as
 * two pnvo_topdown_view calls with out_pstride 2. */
int pnvo_topdown_view_pairs(const float *depth_frames, int n_pairs, int H, int W, const float *consts, int rows_around_center,
                            float *tdv_pairs, void *work, void *stream);
int pnvo_topdown_view(const float *depth, int N, int H, int W, int64_t in_fstride, int64_t in_pstride,
                      const float *consts, int rows_around_center, float *out, int64_t out_fstride,
                      int64_t out_pstride, void *work, void *stream);

/*
 * Replaces: the observation construction of BaseRLTrainerWithVO._compute_local_delta_states_from_vo
 * (base_trainer_with_vo.py:172-269) for n (prev, cur) pairs at once — numpy -> tensor copies, pair concatenation,
 * _discretize_depth_func, the two top-down views — as one pairs kernel + the top-down kernels on the given stream:
 *   rgb_frames   device uint8   [n][2][H][W][3]  (prev frame, cur frame; NULL for models without rgb)
 *   depth_frames device float32 [n][2][H][W]
 *   -> rgb_pairs [n,H,W,6] (0..255 as float), depth_pairs [n,H,W,2], dd_pairs [n,H,W,2*bins] (bins > 0), tdv_pairs [n,H,W,2]
 *      (NULL: skip; tdv_consts as for pnvo_topdown_view, tdv_work >= pnvo_topdown_workspace_bytes(n,H,W))
 *   err_flag (device int32, may be NULL) is set when a depth lies outside [0,1] (the reference asserts, :136-137).
 */
int pnvo_build_obs_pairs(const uint8_t *rgb_frames, const float *depth_frames, int n, int H, int W, int bins,
                         const float *tdv_consts, int rows_around_center, void *tdv_work, float *rgb_pairs, float *depth_pairs,
                         float *dd_pairs, float *tdv_pairs, int32_t *err_flag, void *stream);

/* Frame ring of the batched boundary call.  Consecutive simulator steps of an environment share a frame — this step's prev_obs IS the
 * last step's cur_obs (rl/ppo/ppo_trainer.py:724-841 keeps `prev_obs = observations`) — so the caller uploads one frame per environment
 * and step; the other half of every pair, and its top-down view, come from a per-environment device slot, which then takes the new frame.
 *   up_rgb / up_depth / up_tdv   device: the m uploaded frames [m][H][W][3] uint8 / [m][H][W] float32 / their top-down views [m][H][W]
 *                                 (pnvo_topdown_view on up_depth); up_rgb / up_tdv NULL for models without the modality
 *   ring_rgb / ring_depth / ring_tdv   device: [slots][H][W][3] / [slots][H][W] / [slots][H][W], owned by the caller across calls
 *   idx   device int32 [3][n]: for pair i the index into up_* of its cur frame | of its prev frame, or -1: take the prev frame from
 *         ring slot idx[2][i] | the ring slot that records the cur frame, or -1: none (every slot at most once per call)
 *   -> rgb_frames [n][2][H][W][3], depth_frames [n][2][H][W], tdv_pairs [n][H][W][2]: the inputs of pnvo_forward_raw, bit-identical to
 *      staging both frames of every pair. */
int pnvo_ring_assemble(const uint8_t *up_rgb, const float *up_depth, const float *up_tdv, uint8_t *ring_rgb, float *ring_depth,
                       float *ring_tdv, const int32_t *idx, int n, int H, int W, uint8_t *rgb_frames, float *depth_frames, float *tdv_pairs,
                       void *stream);

/* Host helper of the same boundary: gathers n separately allocated host frames (each bytes_each long) into one (pinned)
 * staging buffer with `threads` copy threads — the per-frame numpy copies were what bounded the batched boundary call. */
int pnvo_stage_frames(const void *const *src, int n, size_t bytes_each, void *dst, int threads);
/* Two frame lists (the rgb frames and the depth frames of a chunk) in one call: one wake-up of the copy workers. */
int pnvo_stage_frames2(const void *const *src_a, size_t bytes_a, void *dst_a, const void *const *src_b, size_t bytes_b, void *dst_b,
                       int n, int threads);

/* ---- VO dataset input pipeline on the device: what StatePairRegressionDataset._process_data
 * does per sample on 20 CPU workers (pointnav_vo/vo/dataset/regression_geo_invariance_iter_dataset.py:205-454), per chunk
 * on the GPU.  Reading the HDF5 file stays with the caller (h5py). ---- */

/* The numpy twin of the top-down view, NormalizedDepth2TopDownViewHabitat.gen_top_down_view (geometry_utils.py:275-470),
 * which the dataset uses (:247-262): same steps as pnvo_topdown_view with the projection in float64.
 *   consts[7] (HOST doubles): inv(K)[0,0], inv(K)[0,2], min_x, x_range*(1+eps), max_depth-min_depth, (max_depth-min_depth)*(1+eps),
 *   min_depth — computed by the caller with numpy as the reference does (pointnav-vo_amd/dataset.py). */
int pnvo_topdown_view_f64(const float *depth, int N, int H, int W, int64_t in_fstride, int64_t in_pstride,
                          const double *consts, int rows_around_center, float *out, int64_t out_fstride,
                          int64_t out_pstride, void *work, void *stream);

/* float16 bit patterns -> float32 (the HDF5 depth arrays, generate_datasets.py:272,290). */
int pnvo_half_to_float(const uint16_t *src, int64_t n, float *dst, void *stream);

/* Batch assembly for M entries from the N samples of a chunk (device pointers):
 *   prev_rgb/cur_rgb [N,H*W*3] uint8, prev_depth/cur_depth [N,H*W] float16 bits, tdv_frames [2,N,H*W] float32 (top-down view of
 *   the prev frames then the cur frames; NULL -> zeros), src[m] = sample of entry m, swap[m] != 0 -> the entry is (cur, prev)
 *   (the geometric-inversion entry, :342-386).  Outputs (each may be NULL): rgb_pairs [M,H,W,6] (0..255), depth_pairs [M,H,W,2],
 *   dd_pairs [M,H,W,2*bins] one-hot with the caller's HOST edges[bins+1] (the dataset compares float16 depth with
 *   float16-rounded edges, regression_iter_dataset.py:32-69; NULL -> float32(i/bins)), tdv_pairs [M,H,W,2].
 *   err_flag is set when a depth is outside [0,1] (the reference asserts, :33-34). */
int pnvo_dataset_pairs(const uint8_t *prev_rgb, const uint8_t *cur_rgb, const uint16_t *prev_depth, const uint16_t *cur_depth,
                       const float *tdv_frames, const int32_t *src, const int32_t *swap, int N, int M, int H, int W, int bins,
                       const float *edges, float *rgb_pairs, float *depth_pairs, float *dd_pairs, float *tdv_pairs,
                       int32_t *err_flag, void *stream);

/* ---- training step (BASELINE config 4).  One optimisation step of one action model, replacing the body of the
 * reference's training iteration: zero_grad / forward in train mode / loss / backward / optimizer.step()
 * (pointnav_vo/vo/engine/vo_cnn_regression_geo_invariance_engine.py:855-901; loss vo_cnn_engine.py:135-198; Adam
 * :122-133).  Parameters and gradients are caller-owned DEVICE buffers, flat, in the order of `toc` (the reference's
 * state_dict parameter order), so a data-parallel job all-reduces ONE buffer (RCCL).  Dropout must be 0. ---- */

/* Bind the flat parameter / gradient buffers (n_floats each) and build the device-side re-packing maps.
 * toc: parameters only (no RunningMeanAndVar buffers).  Also refreshes the kernel operands from `params`. */
int pnvo_train_attach(pnvo_handle h, float *params, float *grads, size_t n_floats, const pnvo_tensor_desc *toc, int ntoc);

/* Re-pack the kernel operands from the flat parameters (call after every optimiser step; replaces nothing in the
 * reference — torch modules read their parameters in place). */
int pnvo_train_refresh(pnvo_handle h, void *stream);

/* model.train(); out = model(obs_pairs) with every activation kept for the backward pass.  run_mean / run_var: device
 * [C] RunningMeanAndVar buffers AFTER this step's update (running_mean_and_var.py:23-63; the caller performs the
 * update — and its all-reduces — with pnvo_input_moments). */
int pnvo_train_forward(pnvo_handle h, const float *rgb, const float *depth, const float *dd, const float *tdv, int B,
                       const float *run_mean, const float *run_var, float *out, void *stream);

/* act_embed variants (vo_cnn_act_embed.py:63-72): the actions [B] (DEVICE int64, values in [0, n_acts]) of the NEXT
 * pnvo_train_forward / pnvo_train_backward pair; the caller keeps the buffer alive until the backward has run. */
int pnvo_train_set_actions(pnvo_handle h, const int64_t *actions);

/* loss.backward() given dLoss/dOut [B,out_dim]: fills the whole gradient buffer (overwrites; no accumulation). */
int pnvo_train_backward(pnvo_handle h, const float *grad_out, void *stream);

/*
 * Gradient-ready hook for data-parallel training (replaces nothing in the reference, whose VO training is single-process;
 * it is what torch DDP's bucketed all-reduce would do for config 4's RCCL gradient exchange).  During pnvo_train_backward the
 * library calls fn(user, first, count, stream) on the host each time the gradients of the flat range [first, first + count)
 * are final — i.e. the launches producing them are enqueued on `stream` — latest layers first: [layer4 .. output head],
 * [layer2 .. layer3], [stem .. layer1] for the reference's parameter order (other orders: one range at the end).  The caller
 * records an event and starts the all-reduce of that range on its communication stream while the rest of the backward runs.
 * pnvo_train_grad_buckets returns the same ranges without running a backward (a rank that must join the collectives of a
 * step in which it had no data).  fn = NULL removes the hook.
 */
typedef void (*pnvo_grad_ready_fn)(void *user, uint64_t first, uint64_t count, void *stream);
int pnvo_train_set_grad_hook(pnvo_handle h, pnvo_grad_ready_fn fn, void *user);
int pnvo_train_grad_buckets(pnvo_handle h, uint64_t *first, uint64_t *count, int cap, int *n_out);

/* Per-channel moments of the assembled network input (reference channel order, rgb/255), the statistics behind
 * RunningMeanAndVar's train-mode update: out[c] = mean_{n,pixel} (x_c - center_c)^power, power in {1,2}; center may be
 * NULL (0).  out: device [C].  power 3: both moments about `center` in ONE pass over the observation tensors —
 * out[c] = mean (x_c - center_c), out[C + c] = mean (x_c - center_c)^2, out: device [2C]. */
int pnvo_input_moments(pnvo_handle h, const float *rgb, const float *depth, const float *dd, const float *tdv, int B,
                       const float *center, int power, float *out, void *stream);

/* RunningMeanAndVar.forward, training branch, for ONE process (running_mean_and_var.py:41-60; the multi-process branch
 * :27-38 needs its all-reduces between the steps and stays on the host): m12 = pnvo_input_moments(power 3, center = the
 * current running mean) of a batch of B pairs; mean / var [C] and count [1] are the module's float32 buffers, updated in place
 * (batch mean, variance about it, Chan's merge).  All pointers are device pointers; asynchronous on `stream`. */
int pnvo_rmv_merge(const float *m12, int C, int B, float *mean, float *var, float *count, void *stream);

/* loss = sum_d mean_i (target - pred)^2 (vo_cnn_engine.py:146-194, unit weights) into *loss (device scalar, may be
 * NULL) and dLoss/dPred into grad [B,D] (may be NULL). */
int pnvo_mse_loss(const float *pred, const float *target, int B, int D, float *loss, float *grad, void *stream);

/* The regression loss in its general form: loss = sum_e coef[e] (target[e] - pred[e])^2 over n = B*D elements, grad =
 * dLoss/dPred.  coef carries everything _compute_loss multiplies by (vo_cnn_engine.py:146-194: loss_weights["dx"|"dz"|"dyaw"],
 * dz_regress_masks) and the 1/len(subset) of the per-data-type means the geometric-invariance engine takes
 * (vo_cnn_regression_geo_invariance_engine.py:690-740); the host builds it (pointnav-vo_amd/train.py regression_coef). */
int pnvo_mse_loss_coef(const float *pred, const float *target, const float *coef, int n, float *loss, float *grad,
                       void *stream);

/* _compute_geo_invariance_inverse_loss (vo_cnn_regression_geo_invariance_engine.py:367-449).  deltas [n_entries,3] in the
 * alternating order the reference asserts (cur_rel_to_prev_0, prev_rel_to_cur_0, ...); actions [n_entries] int32 (the even
 * rows are read; rows equal to move_forward drop the dz constraint).  out4 (device, may be NULL) = {weight * loss,
 * abs_diff_rot, abs_diff_pos[0], abs_diff_pos[1]}; grad (may be NULL) = weight * dLoss/dDeltas [n_entries,3]. */
int pnvo_geo_inverse_loss(const float *deltas, const int32_t *actions, int n_entries, int move_forward, float weight,
                          float *out4, float *grad, void *stream);

/* nn.Dropout(p) of the two places the reference has one — before visual_fc's Linear and before output_head's Linear
 * (pointnav_vo/vo/models/vo_cnn.py:216-227) — for the train-mode forward/backward.  torch's RNG stream cannot be
 * reproduced; the mask is a counter-based hash of (seed, forward count, layer, element): keep with probability 1-p,
 * kept values scaled by 1/(1-p), the same mask in the backward pass.  p = 0 (default) disables it. */
int pnvo_train_set_dropout(pnvo_handle h, float p, uint64_t seed);

/* The scaled mask (0 or 1/(1-p)) the LAST pnvo_train_forward used: layer 0 -> [B, fh*fw, compression channels padded to a multiple of 32]
 * (the kernel's NHWC order of the flattened feature), layer 1 -> [B, hidden], layer 2 -> [B, 32] (the action-embedding
 * columns of act_embed variants).  For checkers. */
int pnvo_train_dropout_mask(pnvo_handle h, int layer, float *out, void *stream);

/* torch.optim.Adam(weight_decay=0, amsgrad=False) on flat device buffers; step counts from 1. */
int pnvo_adam_step(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, size_t n, float lr, float beta1,
                   float beta2, float eps, int step, void *stream);

/* Free everything owned by the handle. */
int pnvo_destroy(pnvo_handle h);

/* ---- the navigation policy's per-step forward: PointNavResNetPolicy.act
 * (pointnav_vo/rl/policies/policy.py:29-46, resnet_policy.py:26-58 and 177-282) for the depth-only configuration of
 * configs/rl/ddppo_pointnav.yaml (resnet18 backbone, 2-layer LSTM, normalize_visual_inputs False, no obs transform). ---- */
typedef struct {
  int32_t width, height;      /* depth frame W, H (341, 192); the encoder sees (W/2, H/2) after avg_pool2d(2) */
  int32_t baseplanes;         /* resnet_baseplanes (32) */
  int32_t hidden;             /* hidden_size (512) */
  int32_t n_actions;          /* action_space.n (4) */
  int32_t rnn_layers;         /* num_recurrent_layers (2, LSTM) */
  int32_t flat_size;          /* after_compression_flat_size (2048) */
} pnvo_policy_config;

typedef struct pnvo_policy_s *pnvo_policy_handle;

/* policy_cls(observation_space=..., action_space=..., ...).to(device) — ddppo_trainer.py:115-133. */
int pnvo_policy_create(const pnvo_policy_config *cfg, int device, pnvo_policy_handle *out);

/* actor_critic.load_state_dict(...) — ddppo_trainer.py:140-146; tensor names exactly as PointNavResNetPolicy.state_dict()
 * spells them ("net.visual_encoder.backbone.conv1.0.weight", "net.state_encoder.rnn.weight_ih_l0",
 * "action_distribution.linear.weight", "critic.fc.weight", ...).  blob is HOST memory. */
int pnvo_policy_load_weights(pnvo_policy_handle h, const float *blob, size_t n_floats, const pnvo_tensor_desc *toc,
                             int ntoc);

/* features, rnn_hidden_states = net(observations, rnn_hidden_states, prev_actions, masks); logits / value of the two
 * heads (policy.py:32-36).  All pointers are DEVICE memory:
 *   depth [B,H,W,1] in 0..1 (observations["depth"]);  goal [B,2] = pointgoal_with_gps_compass (rho, phi);
 *   prev_actions [B] int64;  masks [B] (0 at an episode start);  hidden_in / hidden_out [2*rnn_layers, B, hidden]
 *   (h of every layer, then c of every layer: rnn_state_encoder.py:47-61);  features [B,hidden], logits [B,n_actions],
 *   value [B] may each be NULL.  Sampling / argmax over the logits stays with the caller (policy.py:38-43). */
int pnvo_policy_act(pnvo_policy_handle h, const float *depth, const float *goal, const int64_t *prev_actions,
                    const float *masks, const float *hidden_in, int B, float *hidden_out, float *features, float *logits,
                    float *value, void *stream);

int pnvo_policy_destroy(pnvo_policy_handle h);

/* F.avg_pool2d(x, 2) of 1-channel NHWC frames [N,H,W,1] -> [N,H/2,W/2,2] with channel 1 = 0 (resnet_policy.py:168). */
int pnvo_avgpool2(const float *depth, int N, int H, int W, float *out, void *stream);

/* Message of the last failing call on this handle (or of the last failing handle-less call if h is NULL). */
const char *pnvo_last_error(pnvo_handle h);

/* One-line note of the last SUCCESSFUL call that changed the handle's behaviour (the dense-stem fallback of pnvo_check_inputs,
 * a refused cooperative launch): "" when there is none.  Kept apart from pnvo_last_error, which only ever holds failures. */
const char *pnvo_last_note(pnvo_handle h);

/* ---- introspection used by tests and bench.py (not part of the drop-in surface) ---- */

/* Copy an intermediate activation of the NEXT pnvo_forward into dst (device, capacity in floats).  Names:
 * "input", "stem_conv", "maxpool", "layer{1..4}.{0,1}", "compression", "hidden".  Channel-padded NHWC; the
 * actual shape is returned by pnvo_tap_shape.  Pass name=NULL to clear. */
int pnvo_set_tap(pnvo_handle h, const char *name, float *dst, size_t capacity);
int pnvo_tap_shape(pnvo_handle h, const char *name, int B, int64_t shape[4]);

/* Per-kernel timing with HIP events recorded on the launch stream.  mode 0 = off, 1 = on.
 * pnvo_timing_read synchronises the events, fills up to cap entries and resets the accumulators. */
typedef struct {
  char name[96];
  int64_t launches;
  double total_ms;
  double flops;               /* algorithmic FLOPs of those launches (2*MACs), 0 for non-GEMM kernels */
  double bytes;               /* algorithmic bytes of those launches (tensor reads + writes, once each) */
} pnvo_kernel_time;
/* The same forward stopped after visual_fc's Linear + ReLU: hidden_out [B, hidden] (device).  This is the visual
 * feature the navigation policy consumes (rl/policies/resnet_policy.py:243-250) — see pnvo_policy_* below. */
int pnvo_forward_features(pnvo_handle h, const float *rgb, const float *depth, const float *dd, const float *tdv,
                          const int64_t *actions, int B, float *hidden_out, void *stream);

/* The fused stems exploit the reference's own observation contract: rgb holds integers 0..255 (uint8 frames cast to float,
 * base_trainer_with_vo.py:196-207) and the discretised depth is one-hot per frame (what _discretize_depth_func produces and
 * asserts, :163).  The reference MODEL, however, accepts any float tensor (vo_cnn.py:110-176), so a drop-in must too:
 *   option input_fallback = on (default): pnvo_forward / pnvo_forward_features stay ASYNCHRONOUS — no host wait.  Behind the fused
 *     stem the call enqueues the float32 stem once more, PREDICATED ON THE DEVICE on the flag the fused stem raises when a value
 *     breaks the contract (stem_lds_kernel<.., PAIRED> + pool_keys_from_raw_kernel: both return at once while the flag is down,
 *     ~2 x 3 us of GPU time for contract inputs), so the forward that met such a value repairs itself and delivers the correct
 *     result.  The host reads the (host-mapped) flag at its next entry, without waiting: from then on the handle launches the
 *     float32 stem directly (pnvo_last_note holds a one-line note, pnvo_get_option(h, "stem") reports "dense (fallback)";
 *     pnvo_set_option(h, "stem", ...) lifts it — the one call that waits for the device, to lower the flag safely).  Works inside a
 *     hipGraph capture too (the repair is captured with the forward).  pnvo_train_forward keeps a host-side decision: its
 *     backward must know which stem ran, so it waits (hipEventSynchronize) for its stem kernel — the first of its launches —
 *     and re-runs on the dense stem when the flag is up; the same holds for models the float32 LDS stem does not serve
 *     (stem outputs other than 32 / 64 channels).
 *   option input_fallback = off: no repair launches and no wait; the stem raises the flag and pnvo_check_inputs (definitive after
 *     the caller synchronised the stream) as well as every later forward on the handle return PNVO_ERR_INPUT until the weights
 *     are re-loaded. */
int pnvo_check_inputs(pnvo_handle h);

/* Which kernel family a conv of the residual stages / the compression conv (state_dict prefix, e.g.
 * "visual_encoder.backbone.layer1.0.convs.0") runs on at batch B with the handle's current options — "x2" / "x3" (float32 results
 * from three float16 / six bf16 MFMA terms per product), "fp32-lds", "fp32-generic", "smallnet" (a phase of the persistent
 * small-batch kernel: option small_net, batches <= small_max) — and the matrix-core FLOPs one launch EXECUTES (tile and
 * channel padding and the six-term expansion included): what bench.py prices against the peak of that pipe. */
int pnvo_layer_kernel(pnvo_handle h, const char *name, int B, char *family, size_t cap, double *executed_flops);

int pnvo_timing_mode(pnvo_handle h, int mode);
int pnvo_timing_read(pnvo_handle h, pnvo_kernel_time *entries, int cap, int *n_out);

/* Host-only helper (no GPU needed): pack an OIHW conv weight into the kernel's MFMA operand order.
 * out must hold pnvo_packed_conv_floats(cout, cin, kh, kw) floats. */
size_t pnvo_packed_conv_floats(int cout, int cin, int kh, int kw);
int pnvo_pack_conv_weight(const float *oihw, int cout, int cin, int kh, int kw, float *out);

const char *pnvo_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PNVO_H_ */
