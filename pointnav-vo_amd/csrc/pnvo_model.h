// pnvo_model.h — host-side model state shared by pnvo_api.hip (inference) and pnvo_train_api.hip (training step).
#pragma once
#include <hip/hip_runtime.h>

#include <map>
#include <functional>
#include <string>
#include <vector>

#include "../../include/pnvo.h"
#include "pnvo_internal.h"

static inline int rup(int x, int m) { return (x + m - 1) / m * m; }
static inline int halve(int x) { return (x - 1) / 2 + 1; }

struct Layer {
  std::string name, gn;     // state_dict prefixes (conv weight, following GroupNorm)
  int cin = 0, cinp = 0, cout = 0, coutp = 0, k = 1, kw = 1, stride = 1, pad = 0;
  int hin = 0, win = 0, hout = 0, wout = 0, groups = 1;
  float *wpk = nullptr, *gamma = nullptr, *beta = nullptr;   // device
  std::vector<float> host_w;  // OIHW copy of the loaded weight (source of the bf16 packing, pnvo_bf16.hip)
  // float32-on-bf16-pipe path (conv_x3.hip): three-piece packed weights, built on first use after a (re)load
  unsigned short *wpk_x3 = nullptr;
  unsigned long long x3_gen = 0;
  // two-piece float16 form of the same kernel (inference): packed weights, the inverse of their power-of-two scale, and an
  // upper bound of |input activation| from the producers' GroupNorm parameters (float16 pieces need it below 65504)
  unsigned short *wpk_x2 = nullptr;
  unsigned long long x2_gen = 0;
  float x2_oscale = 1.f;
  float in_bound = 3.0e38f;
};

// Per-handle options (pnvo_set_option; defaults from the PNVO_* environment, read ONCE in pnvo_create).
struct PnvoOptions {
  int stem = 0;        // 0 auto (bf16-matrix-core stem when the model's modalities fit it, else one-hot-aware, else dense), 1 mx, 2 dd, 3 dense
  int conv = 0;        // 0 auto (conv_x3 for launches of >= 192 workgroups, fp32-MFMA kernels below), 1 x3 at any size, 2 fp32, 3 generic
  int stem_form = 0;   // float16-piece stem: 0 auto (fast resident weights when every workgroup gets >= 4 tiles, else one tile per workgroup), 2 one tile per workgroup, 3 resident weights in the tile kernel's summation order (stem_rs_kernel), 4 fast (its own order)
  int pieces = 2;      // operand pieces of conv_x3 at inference: 2 float16 (three product terms) or 3 bf16 (six exact terms)
  int train_pieces = 2;  // the same choice for the TRAINING forward's convs (their backward-data convs keep three bf16 pieces:
                         //   gradients do not fit float16's range)
  int ds_side = 0;     // (opt-in) the 1x1 stride-2 downsample conv of a block (+ its GroupNorm finalisation) on a side stream next to the block's
                       // second 3x3 conv: -1 % at 256 pairs on a GPU of its own, but 10x slower when two processes share a GPU (the fork / join
                       // events across time-sliced queues) — off by default
  int x3_rows = 1;     // 32 -> 32 channel 3x3 stride-1 convs on the row-streaming kernel (conv_rows.hip) where it takes the launch
  int x3_persist = 1;  // shallow-stage 3x3 convs on the persistent form of conv_x3 (next tile's patch fetched during the K loop)
  int x3_strip = 1;    // 64- / 128-channel stride-1 convs on wide strip tiles with the N-tiles split over blockIdx.y (half the weight bytes per pixel)
  int pool_async = 0;  // pooled-key buffer of its own, re-initialised for the next forward on a side stream (see pool_keys).  OFF: measured
                       // slower at every batch (8 pairs 0.44 -> 0.50 ms, 256 pairs 2.37 -> 2.41): the two event hand-overs cost more than the 23 us fill
  int gn_defer = 0;    // pairs up to which multi-tile GroupNorm finalisations move into the consumer conv's prologue (0: never = default).
                       // Round 6: bit-identical, 4-10 launches fewer per forward — and slower (8 pairs 0.427 -> 0.448 ms, 32: 0.630 -> 0.663):
                       // a memory round trip + ~0.5 us of fp64 in EVERY consumer workgroup costs more than a 4 us launch
  int x3_fine = 1;     // small launches of the float16-piece convs take one N-tile per workgroup instead of falling back to the fp32-pipe kernels
  int x3_w8 = 1;       // 256-channel convs on 6 x 11 maps with a tile per CU or more: eight waves of (3,1) tiles per workgroup (two per SIMD) instead of four of (3,2)
  int x3_ksplit = 1;   // fine-plan conv tiles of three / four M-tiles behind >= 128 input channels: the four waves split the K walk, partial sums meet in LDS
  int fc_rows = 48;    // up to this many samples the hidden layer and the head run as fc_rows.hip's two launches (every model of a grouped forward in one)
  int head_fuse = 1;   // the output head (Linear hidden -> out_dim) is computed by the hidden layer's split-K reduction launch (one launch less)
  int ds_fuse = 1;     // the 1x1 stride-2 downsample conv rides on its block's first 3x3 conv (bit-identical raw output, one launch less, block input read once)
  int gn_fuse = 2;     // conv_x3 launches finalise their GroupNorm themselves (bit-identical, one launch less): 2 = launches with one tile per
                       // sample (the default, `on`); 1 = `last`: also launches with several tiles per sample, by the sample's last workgroup to
                       // arrive (round 6: measured SLOWER at every batch — every workgroup waits for its stores and an atomic round trip
                       // before it leaves — kept as an option and a test of the protocol); 0 = never
  int x3_s2 = 1;       // stride-2 convs on conv_x3
  int tail = 1;        // BasicBlock tails fused into the next conv's stager (0: residual_kernel)
  int pool = 1;        // max-pool fused into the stem's epilogue (0: gn_relu_maxpool_kernel)
  int conv3_nt = 0;    // 1: one N-tile per item in conv3_lds
  int graph = 0;       // forward replayed from a captured hipGraph
  int stem_dbg = 0, stem_dbg_pad = 0;   // developer instrumentation of the stem kernels
  int wgrad_stem = 0;  // 0 bf16 matrix cores, 1 fp32
  int wgrad3 = 1;      // weight gradient of the 3x3 stride-1 convs: 1 bf16 matrix cores (wgrad_x3.hip), 0 fp32 kernels
  int pool_bwd = 1;    // max-pool backward fused into the stem's GroupNorm backward
  int dgrad = 1;       // stride-2 backward-data as four parity-phase convs (0: masked taps)
  int bf16_fuse = 1;   // bf16 path: block tails fused
  int bf16_stem3 = 0;  // bf16 path: exact three-piece stem (experiment)
  int input_fallback = 1;   // contract-breaking input (fractional rgb, soft depth codes): re-run on the dense stem and stay on it
  int small_net = 1;   // batches of <= small_max pairs: everything behind the stem conv in ONE persistent launch (smallnet.hip)
  int small_max = 3;   // largest batch the persistent kernel takes (1..4; round 6: the per-layer launches with their fine plans win from 4 pairs on — 0.370 against 0.400 ms)
  int small_prof = 0;  // developer instrumentation: per-phase times of the persistent kernel on stderr
  int small_coop = 1;  // cooperative launch (hipLaunchCooperativeKernel): every workgroup resident by the runtime's guarantee, also next to other
                       // processes' kernels (+17 us); 0: plain launch of <= 144 workgroups (one per CU) for a process that owns the GPU
};

struct TimingRec {
  hipEvent_t a, b;
  int entry;
};

struct pnvo_model_s {
  pnvo_config cfg;
  int device = 0;
  std::string err;
  std::string note;           // pnvo_last_note: what a SUCCESSFUL call changed on the handle (never an error)
  bool loaded = false;

  int C = 0, CP = 0;                 // input channels, padded to 8
  int Hs = 0, Ws = 0, Hp = 0, Wp = 0, fh = 0, fw = 0, comp_c = 0, comp_cp = 0;
  std::vector<Layer> convs;          // stem, residual stages in execution order, compression
  Layer fc, head;
  float *fc_bias = nullptr, *head_bias = nullptr;   // device; fc_bias has 1 or n_acts+1 rows
  // grouped forward (pnvo_forward_grouped_raw), set on the LEADER handle for the duration of the call: handles of the action models
  // in sample order (grp[0] = this handle), cumulative sample ends; kernels pick a sample's operands by its model
  int grp_n = 0;
  struct pnvo_model_s *grp[3] = {nullptr, nullptr, nullptr};
  int grp_end[3] = {0, 0, 0};
  float *fc_rows_w = nullptr;                // device [hidden][fh * fw * comp_cp]: the hidden layer's weight rows in the activation's order (fc_rows.hip)
  float *head_w_plain = nullptr;             // device [out_dim][hidden]: the head's weight as loaded (the head riding on the hidden layer's split-K reduction)
  const float *head_ride_w = nullptr;        // ... the weight it reads: head_w_plain, or the flat parameter buffer of an attached training step
  float *head_ride_out = nullptr;            // set around the hidden layer's launch by the forward: where the riding head writes [B][out_dim]
  bool head_rode = false;                    //   ... and whether it did (else the forward launches the head)
  std::vector<float> mean, stdev;    // host copies for the assemble kernel arguments (reference channel order)
  // fused stem: K-order of the stem = observation tensors concatenated (rgb | depth | dd | tdv), 2-channel pieces
  std::vector<int> stem_ref_of_new;  // new channel -> reference channel (vo_cnn.py:169-174 order), -1 = pad
  std::vector<int> stem_tensor_of_new, stem_ch_of_new;
  float *stem_sc = nullptr, *stem_sh = nullptr, *zero_page = nullptr;   // device: whitening table in the new order
  float *kpart = nullptr;             // split-K partials of the linear layers (conv_mfma.hip conv_ksplit)
  size_t kpart_floats = 0;
  float *stem_wpk16 = nullptr;       // stem weights packed for the LDS-staged 16x16x4 kernel
  int CPL = 0;                       // stem channels per pixel in LDS (C rounded up to 16)
  // one-hot-aware stem (stem_dd.hip): dense channels + indicator on the matrix cores, depth bins as a table gather
  bool dd_ok = false;
  int dd_bins = 0;
  std::vector<int> dd_dense_tensor, dd_dense_ch;   // dense channel d -> (observation tensor, channel), -1 = indicator/pad
  float *dd_wpk = nullptr, *dd_table = nullptr, *dd_sc = nullptr, *dd_sh = nullptr;
  int *dd_flag = nullptr;            // host-mapped copy of dd_flag_dev (published by the kernel behind the stem): what the HOST reads
  int *dd_flag_dev = nullptr;        // device memory: raised by a fused stem whose stager met a value outside the observation contract;
                                     // read by the predicated repair launches (a host-mapped flag costs every wave a PCIe round trip)
  unsigned long long *dd_prof = nullptr;   // PNVO_STEM_DBG=9: {staging, K loop, epilogue} cycles, tiles

  // stem on the bf16 matrix cores (stem_mx.hip): exact three-piece bf16 weights -> float32 results (inference default)
  bool mx_ok = false;
  unsigned short *mx_wpk3 = nullptr;         // device: three-piece packing (float32 results)
  unsigned short *mx_wpk2 = nullptr;         // device: two float16 pieces (inference default) and the inverse of their scale
  float mx_oscale = 1.f;
  float *mx_scale2_dev = nullptr;            // training: {scale, 1/scale} of mx_wpk2 as the device-side re-pack chose it
  bool mx_wpk2_dev = false;                  // mx_wpk2 currently holds the device-side re-pack (scale in mx_scale2_dev), not the host's
  std::vector<float> mx_wk, mx_wk_swapped;   // host [cout][32 slots][49]: whitening-folded weights, as is / for the
                                             //   (cur, prev) channel-swapped pair (geometric-invariance dual forward)
  int mx_xslot[4] = {-1, -1, -1, -1};        // K-slots of the float-modality channels
  std::vector<int> mx_slot_ref, mx_slot_new; // K-slot -> reference channel / position in the stem's tensor-major order (-1: none)
  float *mx_pages = nullptr;                 // device: 64 zeros (out-of-image reads)
  unsigned long long *mx_prof = nullptr;     // PNVO_STEM_DBG=9: per-wave phase cycle sums of stem_mx / stem_ps
  bool mx_prof_rs = false;                   //   ... the last stem launch was the resident-weight form
  int num_cus = 256;                         // compute units of the device (grid of the persistent kernels)
  bool in_train_forward = false;
  bool train_mx = false;                     // the attached training step rebuilds the mx stem operands every step
  PnvoOptions opt;
  bool dense_sticky = false;                 // an input outside the mx/dd stems' contract was met: this handle stays on the dense stem
  int fallback_count = 0;                    // forwards re-run on the dense stem
  hipStream_t side_stream = nullptr;         // option ds_side: forked behind a block's first conv, joined before the block tail's consumer
  hipEvent_t side_fork = nullptr, side_join = nullptr;
  float *stats_side = nullptr;               // the side stream's GroupNorm partials (the main stream's conv writes m->stats meanwhile)
  size_t stats_side_floats = 0, stats_floats = 0;
  // option pool_async: the pooled stem keys live in a buffer of their own, and their re-initialisation for the NEXT forward (a 135 MB
  // fill at 256 pairs, 23 us on the critical path before the stem) runs on a stream of its own behind this forward's consumer of the
  // keys, next to the MFMA-bound deep stages; the next forward's stem waits for it through an event (no host wait)
  float *pool_keys = nullptr;                // [cap][Hp][Wp][baseplanes] int32 keys
  hipStream_t keys_stream = nullptr;
  hipEvent_t keys_free_ev = nullptr, keys_ready_ev = nullptr;
  bool keys_primed = false;                  // the fill for the next forward is enqueued (keys_ready_ev recorded behind it)
  int keys_primed_B = 0;                     //   ... for this many pairs
  // Deferred GroupNorm finalisation (option gn_defer, small launches): a conv_x3 producer whose consumer is a conv_x3_kernel launch
  // leaves its partial sums un-finalised; the consumer builds its sample's scale / shift table in its prologue (ConvX3Args::fin_in /
  // fin_res).  gn_pend[k]: what is pending behind the scale / shift pair ssA (0), ssB (1), ssD (2); defer_main / defer_ride: set by the
  // forward around a producer's pnvo_run_conv call.
  struct GnPend {
    const float *stats = nullptr;
    int slots = 0, cpg = 0;
    size_t layer = 0;                         // index into convs (gamma / beta; the same layer of the other models of a grouped forward)
    bool valid = false;
  } gn_pend[3];
  bool defer_main = false, defer_ride = false;
  float *statsB = nullptr;                   // partial sums of the convs that write ssB (the second conv of a block): a producer's sums
                                             // must outlive the next launch, which writes its own
  float *stats_ds = nullptr;                 // GroupNorm partials of a downsample conv riding on its block's first conv (stats_floats)
  float *gn_ctr = nullptr;                   // [cap][16] unsigned arrival counters of the in-kernel GroupNorm finalisation (zero between launches)
  hipEvent_t stem_ev = nullptr;              // recorded behind a contract-checking stem launch (pnvo_mark_stem)
  bool stem_ev_pending = false;
  // pnvo_forward_raw / pnvo_forward_dual_raw: sensor frames of the call in flight (the stem's RAW stager reads them)
  const unsigned char *raw_rgb = nullptr;
  const float *raw_depth = nullptr;
  int *raw_err = nullptr;
  float *rawws[3] = {nullptr, nullptr, nullptr};   // rgb / depth / dd pair tensors of the materialising fallback of the raw entry
  int rawws_cap = 0;
  int precision = 0;                         // pnvo_set_precision: 0 float32 (default), 1 bfloat16 (BASELINE config 3)
  unsigned long long load_gen = 0;           // bumped by pnvo_load_weights (operands derived lazily are rebuilt)
  unsigned long long weights_gen_at_load = 0;  // weights_gen as pnvo_load_weights left it
  unsigned long long uid = 0;                // process-unique handle id (cache keys that must not alias a re-used address)
  unsigned long long weights_gen = 0;        // bumped by pnvo_load_weights AND by every pnvo_train_refresh (conv_x3 operands)
  void *bf = nullptr;                        // Bf16State (pnvo_bf16.hip)             // set by pnvo_train_forward: its stem operands are rebuilt on the device

  int cap = 0;                       // batch the workspace is sized for
  float *xin = nullptr, *stem_raw = nullptr, *bufY[2] = {nullptr, nullptr};
  float *rawA = nullptr, *rawB = nullptr, *rawD = nullptr, *rawC = nullptr, *comp_raw = nullptr, *hid = nullptr,
        *stats = nullptr;
  bool bottleneck = false;           // resnet50 / resnet101 backbone
  std::vector<int> nblocks;          // residual blocks per stage
  float *ssA[2] = {nullptr, nullptr}, *ssB[2] = {nullptr, nullptr}, *ssD[2] = {nullptr, nullptr},
        *ssC[2] = {nullptr, nullptr};
  float *tapbuf = nullptr;
  size_t tapbuf_floats = 0;

  std::string tap_name;
  float *tap_dst = nullptr;
  size_t tap_cap = 0;

  bool features_only = false;        // pnvo_forward_features: stop after the hidden layer
  void *train = nullptr;             // TrainState (pnvo_train_api.hip), present after pnvo_train_attach
  void *small = nullptr;             // SmallNet (smallnet.hip): the persistent small-batch kernel's operands, built on first use
  int stem_slots_out = 0;            // statistics slots per sample the last stem launch wrote into `stats` ([B][slots][CP][2])
  bool stem_skip_finalize = false;   // the consumer of this forward's stem reduces those slots itself (smallnet.hip)

  // Opt-in (PNVO_GRAPH=1): the whole forward (~60 launches) captured once per (batch, tensor addresses, kernel
  // selection) into a hipGraph and replayed (see pnvo_forward for the measurement that keeps it off by default).
  struct GraphEntry {
    const void *key[8];
    int B;
    hipGraph_t graph;
    hipGraphExec_t exec;
    unsigned long long stamp;
  };
  std::vector<GraphEntry> graphs;
  std::vector<GraphEntry> seen;      // call shapes met once (captured when they come back: no capture for one-off calls)
  float *out_ws = nullptr;           // [cap, out_dim]: the graph's output (copied to the caller's tensor after replay)
  hipStream_t cap_stream = nullptr;
  int graph_mode = -1;               // -1: read PNVO_GRAPH on first use; 0 off; 1 on
  unsigned long long graph_clock = 0;

  int timing = 0;
  std::vector<pnvo_kernel_time> tentries;
  std::map<std::string, int> tindex;
  std::vector<TimingRec> trecs;
  std::vector<hipEvent_t> evpool;
};


// The tail of the previous BasicBlock (resnet.py:47-55) handed to the next block's first conv instead of a pass of its own:
// that conv's input is relu(x*in_scale+in_shift + r), r = res or res*res_scale+res_shift, and it also writes it to `out`
// (the block output: the next skip branch / downsample conv read it).
// res == nullptr: the conv's input x holds the pooled stem keys of stem_mx.hip (POOL); the conv decodes them, applies
// |in_scale|, in_shift and ReLU, and writes the pooled activations to `out`.
struct BlockTail {
  const float *res, *res_scale, *res_shift;
  float *out;
};

// The 1x1 stride-2 downsample conv of a stride-2 BasicBlock (resnet.py:192-195) riding on the launch of the block's first 3x3 conv
// (conv_x3_kernel<.., DSF>): raw output -> y, GroupNorm scale / shift -> ss[0] / ss[1] (finalised with the conv's own).
struct DsRide {
  const Layer *cd;
  float *y;
  float *const *ss;
  float *mu, *rstd;      // [B,groups] statistics of the downsample GroupNorm for a backward pass, or nullptr
};

// helpers implemented in pnvo_api.hip
int pnvo_run_conv(pnvo_handle m, const Layer &l, int B, const float *x, const float *in_scale, const float *in_shift,
                  float *y, int y_cstride, float *ss[2], const float *bias, const int64_t *bias_row, int relu_out,
                  hipStream_t s, const float *const *src, float *mu_out, float *rstd_out, const BlockTail *tail = nullptr,
                  const DsRide *ride = nullptr);
bool pnvo_conv_takes_ds(pnvo_handle m, const Layer &c1, const Layer &cd, int B);   // would pnvo_run_conv(c1) carry cd as a DsRide?
bool pnvo_conv_on_x3(pnvo_handle m, const Layer &l, int B);                        // would pnvo_run_conv(l) use conv_x3.hip?
bool pnvo_conv_takes_tail(pnvo_handle m, const Layer &l, int B);   // would pnvo_run_conv(l) accept a BlockTail (conv_x3 path)?
void pnvo_pack_conv_weight_cinp(const float *oihw, int cout, int cin, int cinp, int kh, int kw, std::vector<float> &out);
int pnvo_run_stem(pnvo_handle m, int B, const float *const *src, float *y, float *ss[2], float *mu_out, float *rstd_out,
                  hipStream_t s, int *pool_keys = nullptr);
bool pnvo_stem_on_mx(pnvo_handle m);
void pnvo_stem_raw_args(pnvo_handle m, pnvo::StemMXArgs &a);   // fills the RAW-stager fields of a stem launch when m->raw_depth is set
int pnvo_mark_stem(pnvo_handle m, hipStream_t s);
int pnvo_input_fallback(pnvo_handle m, hipStream_t s, bool *rerun);   // after the forward is enqueued: wait for the stem, re-run on the dense stem?
void pnvo_train_free(pnvo_handle m);   // pnvo_train_api.hip
const float *pnvo_train_weight_ptr(pnvo_handle m, const std::string &name);   // pnvo_train_api.hip: device pointer or nullptr
void pnvo_chain_in_bounds(pnvo_handle h, const std::function<float(const Layer &)> &gn_bound);   // pnvo_api.hip
const float *pnvo_train_x2_scale(pnvo_handle m, const std::string &name);     // device {scale, 1/scale} of that conv weight's float16 pieces, or nullptr
void pnvo_bf16_free(pnvo_handle m);    // pnvo_bf16.hip
bool pnvo_small_usable(pnvo_handle m, int B);   // smallnet.hip: does this call shape take the persistent kernel?
int pnvo_small_forward(pnvo_handle m, int B, const int64_t *actions, float *out, hipStream_t s);
void pnvo_small_free(pnvo_handle m);
int pnvo_forward_bf16(pnvo_handle *hs, int nm, const float *rgb, const float *depth, const float *dd, const float *tdv,
                      const int64_t *actions, int B, float *const *outs, hipStream_t s);
int pnvo_fail(pnvo_handle h, int code, const std::string &msg);
void pnvo_free_dev(float *&p);
int pnvo_ensure_workspace(pnvo_handle m, int B);
void pnvo_drop_graphs(pnvo_handle m);   // forget the captured forward graphs (their kernel arguments went stale)

#define HIPCHK(h, expr)                                                                                   \
  do {                                                                                                    \
    hipError_t e__ = (expr);                                                                              \
    if (e__ != hipSuccess)                                                                                \
      return pnvo_fail(h, PNVO_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));             \
  } while (0)

// scoped HIP-event timing of one launch (no-op unless pnvo_timing_mode(h, 1))
struct PnvoTimed {
  pnvo_model_s *m;
  hipStream_t s;
  int rec = -1;
  PnvoTimed(pnvo_model_s *m_, hipStream_t s_, const std::string &name, double flops, double bytes);
  ~PnvoTimed();
};
