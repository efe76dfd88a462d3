// smallnet.hip — the whole network behind the stem conv in ONE persistent launch, for the batch sizes of the navigation loop.
//
// The reference calls the VO model once per environment step with a batch of ONE pair (rl/ppo/ppo_trainer.py:836-841,
// challenge_2020/challenge2020_agent.py:311-394).  At that size the ~58 launches that follow the stem (sixteen 3x3 convs,
// three 1x1 downsample convs, twenty GroupNorm finalisations, eight residual passes, max-pool, compression, two Linear
// layers — resnet.py:29-55,153-212, vo_cnn.py:70-176) are a chain of dependent 5–25 µs kernels, each spread over 1–33
// workgroups: 0.42 ms of latency for 1.2 GFLOP.  Here every layer is a PHASE of one kernel: 144 workgroups of 8 waves (one per
// CU) meet at a grid barrier between phases, every phase spreads its layer over 96–144 of them, and what the small passes did
// rides on the phases themselves:
//   * GroupNorm finalisation: every producer tile writes {sum, sum of squares} per group; every consumer workgroup reduces the
//     slots of its sample in double precision (fixed order) into scale / shift tables in LDS;
//   * GroupNorm + ReLU, the BasicBlock tail relu(GN2(conv2) + skip) and the stem's GroupNorm + ReLU + max-pool are the input
//     transform of the consuming phase (the tile that owns a pixel also writes the block output the next skip branch reads);
//   * the 1x1 stride-2 downsample conv shares the staged patch of the block's first 3x3 conv (its centre tap).
// Convs are implicit GEMMs on v_mfma_f32_16x16x4_f32 (float32 in, float32 accumulate): a tile is TH x TW output pixels
// (16 * MB) x 16 output channels; the MB M-blocks and an 8/MB-way split of K = 9 * Cin go over the 8 waves, the K-split partials
// meet in LDS in wave order.  A fragments come from the LDS patch; a wave's B fragments of the whole K walk sit in registers,
// fetched in 16-byte loads from a host-side packing in walk order.  Linear layers: 4 outputs per workgroup, K over 2 waves each.
//
// Grid barrier: one device-scope counter, in two halves — ARRIVE right behind a phase's stores, then the next phase's tile,
// weights and addresses are prepared, then WAIT.  Tensors that cross workgroups are moved with agent-scope (sc1) loads / stores, so
// the barrier carries no cache write-back or invalidate.  The launch is cooperative by default; a bounded spin turns an
// impossible wait into NaN results and an error on the handle, never a hung device.  DESIGN.md section 4 has the measurements.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "pnvo_model.h"

namespace pnvo {

typedef float sn_f32x4 __attribute__((ext_vector_type(4)));

constexpr int SN_THREADS = 512, SN_WAVES = 8, SN_TAB = 1024, SN_MAXB = 8;   // (MAXB 8, TAB 1024: the policy's quarter-size frames, see pnvo_small_usable)
constexpr int sn_mblocks(int c4) { return c4 <= 8 ? 4 : c4 <= 16 ? 2 : 1; }   // M-blocks (16 output pixels) of a conv tile, by input channels / 4
constexpr int SN_KP = SN_WAVES / 4, SN_WPRE = 6;                  // linear layers: K parts per output, weight vectors fetched ahead
constexpr int SN_RED_FLOATS = 2 * SN_WAVES * 256;                 // K-split partials: two convs x 8 waves x (64 lanes x 4)
constexpr int SN_MAXPH = 40;                                      // phases (BasicBlock nets up to resnet34: 36)
constexpr int SN_DESC_FLOATS = SN_MAXPH * 72 + 64;                // the phase table (copied to LDS at kernel start) + this workgroup's first tiles
constexpr int SN_FIXED_FLOATS = 4 * SN_TAB + SN_RED_FLOATS + SN_DESC_FLOATS;   // tables + partials + phase table, then the patch / activation vector
constexpr unsigned SN_SPIN_LIMIT = 1u << 19;

struct SnGN {              // GroupNorm of a producer phase: partial statistics + affine parameters
  const float *part;       // layout 0: [B][G][slots][2];  layout 1 (the stem kernels): [B][slots][CP][2] per channel
  const float *gamma, *beta;
  int G, cpg, lgcpg, C, CP, slots, layout;   // cpg channels per group on the padded channel axis (a power of two); C real channels
  float inv_cnt;           // 1 / (pixels x real channels per group)
};

struct SnPhase {            // 70 dwords; GroupNorms first (their fields are read with a run-time base), then pointers, then integers
  SnGN gin, gres;
  const int *tiles;        // [ntiles] n << 24 | tile row << 16 | tile column << 8 | cout tile
  const float *in, *res;
  float *blk_out;
  const float *w, *w_ds, *bias;
  float *out, *part, *out_ds, *part_ds;
  int kind;                // 0 conv3x3 (+ optional 1x1 stride-2 conv on the centre tap), 1 GN+ReLU+max-pool, 2 linear
  int type;                // conv: stride * 8 + log2(cinp / 4) — selects the compiled variant
  int cinp, coutp, cout, Hin, Win, Ho, Wo;
  int tiles_y, tiles_x, NT, ntiles;
  int in_mode;             // 0 final activations; 1 relu(x*sc+sh); 2 relu(x*sc+sh + r), r = res; 3: r = res*sc2+sh2.  2/3: owner writes blk_out
  int out_cpg, out_G, out_slots, relu_out, K, use_row;
};

static_assert(sizeof(SnPhase) <= 72 * 4, "SnPhase outgrew its LDS slot");

struct SnArgs {
  const SnPhase *ph;
  int nph, B;
  unsigned *bar;
  unsigned bar_base;
  int *err;
  const int64_t *bias_row;   // [B] or nullptr (act-embed variants: row of the first Linear layer's bias table)
  float *final_out;          // the caller's [B, out_dim]
  int final_n;               // B * out_dim (a barrier that times out leaves NaNs there: never a plausible wrong pose)
  int dbg;                   // small_prof = 2: barriers only; 3: barriers without the cache write-back / invalidate (timing experiments)
  unsigned long long *prof;  // option small_prof: per phase {block 0 start, block 0 end, max phase time, max barrier wait} (10 ns ticks)
};

namespace {


__device__ __forceinline__ int sn_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <typename T>
__device__ __forceinline__ T *sn_uni(T *q) {            // a pointer read from the LDS phase table -> scalar registers
  const unsigned long long v = reinterpret_cast<unsigned long long>(q);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<T *>(((unsigned long long)hi << 32) | lo);
}

// The thread index inside phase code: wave index (a scalar, set once at kernel start) * 64 + the lane id from mbcnt.  Nothing
// derived from the threadIdx register has to live across the phases (it would be spilled to scratch memory and re-loaded — one
// memory round trip — at every phase entry).
__device__ __forceinline__ int sn_lane() {   // (volatile: computed where it is used, never hoisted out of the phase loop and spilled)
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}
#define SN_TID (sn_wave * 64 + sn_lane())

// A phase's descriptor in registers: lane k of d0 holds dword k of the LDS copy (d1: dwords 64..71).  One LDS read per phase; a
// field is a v_readlane with a constant (or, for the two GroupNorm blocks, scalar) lane index — no per-field LDS round trips.
struct SnRegs {
  int d0, d1;
};
__device__ __forceinline__ SnRegs sn_regs(const SnPhase *lp, int lane) {
  const int *q = reinterpret_cast<const int *>(lp);
  SnRegs r;
  r.d0 = q[lane];
  r.d1 = q[64 + (lane & 7)];
  return r;
}
template <int K>
__device__ __forceinline__ int sn_geti(const SnRegs &r) {
  return K < 64 ? __builtin_amdgcn_readlane(r.d0, K < 64 ? K : 0) : __builtin_amdgcn_readlane(r.d1, K < 64 ? 0 : K - 64);
}
template <int K, typename T>
__device__ __forceinline__ T *sn_getp(const SnRegs &r) {
  const unsigned lo = (unsigned)sn_geti<K>(r), hi = (unsigned)sn_geti<K + 1>(r);
  return reinterpret_cast<T *>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ int sn_geti_at(const SnRegs &r, int k) { return __builtin_amdgcn_readlane(r.d0, k); }   // k < 64, uniform
template <typename T>
__device__ __forceinline__ T *sn_getp_at(const SnRegs &r, int k) {
  const unsigned lo = (unsigned)sn_geti_at(r, k), hi = (unsigned)sn_geti_at(r, k + 1);
  return reinterpret_cast<T *>(((unsigned long long)hi << 32) | lo);
}
#define SN_I(r, f) sn_geti<(int)(offsetof(SnPhase, f) / 4)>(r)
#define SN_P(T, r, f) sn_getp<(int)(offsetof(SnPhase, f) / 4), T>(r)
// dword offsets inside an SnGN block (base = 0 for gin, 14 for gres)
constexpr int GN_PART = 0, GN_GAMMA = 2, GN_BETA = 4, GN_G = 6, GN_LGCPG = 8, GN_C = 9, GN_CP = 10, GN_SLOTS = 11, GN_LAYOUT = 12,
              GN_INV = 13, GN_DWORDS = 14;
static_assert(sizeof(SnGN) == 4 * GN_DWORDS && offsetof(SnGN, inv_cnt) == 4 * GN_INV && offsetof(SnGN, G) == 4 * GN_G, "SnGN layout");

// Data that crosses workgroups inside the launch (conv outputs, GroupNorm partials, block outputs, the hidden vector) is written
// and read with AGENT-scope accesses (sc1: write-through / coherent across the per-XCD L2s), so the grid barrier needs no L2
// write-back or invalidate — and weights, the instruction stream and the phase table stay cached across phases.
typedef __amdgpu_buffer_rsrc_t sn_rsrc_t;
constexpr int SN_SC1 = 16;
__device__ __forceinline__ sn_rsrc_t sn_rsrc(const void *q) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(q), 0, 0xffffffffu, 0x00020000);
}
__device__ __forceinline__ sn_f32x4 sn_ld4(sn_rsrc_t r, unsigned fidx) {
  return __builtin_bit_cast(sn_f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, fidx * 4u, 0, SN_SC1));
}
__device__ __forceinline__ float2 sn_ld2(sn_rsrc_t r, unsigned fidx) {
  return __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(r, fidx * 4u, 0, SN_SC1));
}
__device__ __forceinline__ void sn_st4(sn_rsrc_t r, unsigned fidx, sn_f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, v), r, fidx * 4u, 0, SN_SC1);
}
__device__ __forceinline__ void sn_st2(sn_rsrc_t r, unsigned fidx, float2 v) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((__vector_size__(2 * sizeof(unsigned)))) unsigned, v), r, fidx * 4u, 0, SN_SC1);
}
__device__ __forceinline__ void sn_st1(sn_rsrc_t r, unsigned fidx, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, fidx * 4u, 0, SN_SC1);
}

// Grid barrier in two halves.  ARRIVE as soon as a phase's last store is issued: every wave waits for its own stores, the
// workgroup meets, thread 0 bumps the device-scope counter.  Then the workgroup prepares the
// next phase (tile, weight fragments, addresses: nothing that depends on other workgroups) and only then WAITS: thread 0 polls
// the counter, the workgroup meets again.  A bounded spin turns an impossible wait into an error flag.
__device__ __forceinline__ void sn_grid_arrive(unsigned *ctr, int tid) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool sn_grid_wait(unsigned *ctr, unsigned target, int *err, int *flag, int tid) {
  if (tid == 0) {
    unsigned spins = 0;
    int bad = 0;
    while ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > SN_SPIN_LIMIT) {
        bad = 1;
        break;
      }
    }
    if (bad) *err = 1;
    *flag = bad;
  }
  __syncthreads();
  return *flag == 0;
}

// sum over the 16 lanes of a row (every lane of the row ends up with it): two quad permutes, half-row mirror, row mirror
__device__ __forceinline__ double sn_row_sum(double v) {
  union {
    double d;
    int i[2];
  } x, y;
#define SN_DPP_STEP(ctrl)                                                  \
  x.d = v;                                                                 \
  y.i[0] = __builtin_amdgcn_update_dpp(0, x.i[0], ctrl, 0xf, 0xf, false);  \
  y.i[1] = __builtin_amdgcn_update_dpp(0, x.i[1], ctrl, 0xf, 0xf, false);  \
  v += y.d;
  SN_DPP_STEP(0xB1)    // quad_perm [1,0,3,2]
  SN_DPP_STEP(0x4E)    // quad_perm [2,3,0,1]
  SN_DPP_STEP(0x141)   // row_half_mirror
  SN_DPP_STEP(0x140)   // row_mirror
#undef SN_DPP_STEP
  return v;
}

// One group per 16-lane row: NJ slot pairs per lane, all loads in flight together with gamma / beta.
template <int NJ>
__device__ __forceinline__ void sn_gn_rows(const SnRegs &r, int gb, int gi, int n, bool act, int l16, float *sc, float *sh) {
  const int g_slots = sn_geti_at(r, gb + GN_SLOTS), g_lgcpg = sn_geti_at(r, gb + GN_LGCPG), g_layout = sn_geti_at(r, gb + GN_LAYOUT),
            g_CP = sn_geti_at(r, gb + GN_CP), g_C = sn_geti_at(r, gb + GN_C), g_G = sn_geti_at(r, gb + GN_G), g_cpg = 1 << g_lgcpg;
  const float *g_gamma = sn_getp_at<const float>(r, gb + GN_GAMMA), *g_beta = sn_getp_at<const float>(r, gb + GN_BETA);
  const float g_inv = __builtin_bit_cast(float, sn_geti_at(r, gb + GN_INV));
  const int nk = g_layout == 0 ? g_slots : g_slots << g_lgcpg;
  const float2 *pp = reinterpret_cast<const float2 *>(sn_getp_at<const float>(r, gb + GN_PART)) + (g_layout == 0 ? 0u : (unsigned)(n * g_slots * g_CP));
  const sn_rsrc_t rp = sn_rsrc(pp);          // layout 0: partials written by other workgroups of this launch
  const unsigned row0 = g_layout == 0 ? (unsigned)((n * g_G + gi) * g_slots) : 0u;
  float2 v[NJ];
  float gm[2] = {0.f, 0.f}, bt[2] = {0.f, 0.f};
  // (branch-free: every lane loads a valid address and discards what it does not own — a predicated load would be waited for
  //  before the next one is issued)
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int kk = j * 16 + l16;
    const int kc = kk < nk ? kk : nk - 1;
    unsigned idx = row0 + (unsigned)kc;
    if (g_layout != 0) {
      const int slot = kc >> g_lgcpg;
      idx = (unsigned)(slot * g_CP + (gi << g_lgcpg) + (kc - (slot << g_lgcpg)));
    }
    v[j] = sn_ld2(rp, idx * 2u);
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int c = l16 + 16 * h, ch = (gi << g_lgcpg) + c;
    const int cc = ch < g_C ? ch : g_C - 1;
    gm[h] = g_gamma[cc];
    bt[h] = g_beta[cc];
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j)
    if (!act || j * 16 + l16 >= nk) v[j].x = v[j].y = 0.f;
  double s1 = 0.0, s2 = 0.0;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    s1 += (double)v[j].x;
    s2 += (double)v[j].y;
  }
  if (nk > NJ * 16) {                                  // (more slots than this variant keeps in flight)
    for (int kk = NJ * 16 + l16; kk < nk; kk += 16) {
      unsigned idx = row0 + (unsigned)kk;
      if (g_layout != 0) {
        const int slot = kk >> g_lgcpg;
        idx = (unsigned)(slot * g_CP + (gi << g_lgcpg) + (kk - (slot << g_lgcpg)));
      }
      if (act) {
        const float2 u = sn_ld2(rp, idx * 2u);
        s1 += (double)u.x;
        s2 += (double)u.y;
      }
    }
  }
  s1 = sn_row_sum(s1);
  s2 = sn_row_sum(s2);
  const double mu = s1 * (double)g_inv;
  double var = s2 * (double)g_inv - mu * mu;
  if (var < 0.0) var = 0.0;
  const float rstd = 1.0f / sqrtf((float)(var + 1e-5));
  const float muf = (float)mu;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int c = l16 + 16 * h, ch = (gi << g_lgcpg) + c;
    if (act && c < g_cpg) {
      const float sv = rstd * gm[h];
      sc[ch] = ch < g_C ? sv : 0.f;
      sh[ch] = ch < g_C ? __builtin_fmaf(-muf, sv, bt[h]) : 0.f;
    }
  }
  for (int c = l16 + 32; c < g_cpg; c += 16) {       // groups of more than 32 channels (GroupNorm(1, C) of a 64-channel compression)
    const int ch = (gi << g_lgcpg) + c;
    if (act) {
      const float sv = ch < g_C ? rstd * g_gamma[ch] : 0.f;
      sc[ch] = sv;
      sh[ch] = ch < g_C ? __builtin_fmaf(-muf, sv, g_beta[ch]) : 0.f;
    }
  }
}

// scale / shift of every channel of sample n into LDS tables for one or two GroupNorms (the input's, the skip branch's):
// one group per 16-lane row, 32 groups per pass of the workgroup, double-precision sums in a fixed order.
// (Groups of one table fill whole waves: G is 1 or a multiple of 4.)
__device__ __forceinline__ void sn_gn_tables(const SnRegs &r, float *sca, float *sha, bool use_b, float *scb, float *shb, int n, int tid) {
  const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15;
  const int Ga = sn_geti_at(r, GN_G), Gb = use_b ? sn_geti_at(r, GN_DWORDS + GN_G) : 0;
  const int total = Ga + Gb;
  for (int base = 0; base < total; base += 4 * SN_WAVES) {
    const int it0 = base + w * 4;
    if (it0 >= total) break;
    const bool sel = it0 >= Ga;
    const int gb = sel ? GN_DWORDS : 0;
    const int Gg = sel ? Gb : Ga;
    int gi = it0 - (sel ? Ga : 0) + (lane >> 4);
    const bool act = gi < Gg;
    gi = act ? gi : Gg - 1;                        // (inactive rows compute a valid group and drop it)
    float *sc = sel ? scb : sca, *sh = sel ? shb : sha;
    const int slots = sn_geti_at(r, gb + GN_SLOTS);
    const int nk = sn_geti_at(r, gb + GN_LAYOUT) == 0 ? slots : slots << sn_geti_at(r, gb + GN_LGCPG);
    if (nk <= 32)
      sn_gn_rows<2>(r, gb, gi, n, act, l16, sc, sh);
    else
      sn_gn_rows<17>(r, gb, gi, n, act, l16, sc, sh);
  }
}

#define SN_STAMP(k) do { if (pf != nullptr && tid == 0 && blockIdx.x == 0) pf[k] = wall_clock64(); } while (0)

// One 3x3 conv layer (pad 1) as a phase.  ST stride, LGC4 = log2(input channels / 4).
//   K split: KS = min(8, C4/4) waves per M-block, MB = 8 / KS M-blocks (16 output pixels each) per tile, SPT = C4 / KS K-steps per tap
//   and wave (4; 8 at 256 input channels).  Tile = TH x TW output pixels x 16 output channels; patch PH x PW x (CIN + 4) floats in LDS.
// Everything that does not depend on the previous phase — the tile, the B fragments of the whole K walk, the addresses and
// bounds of this thread's patch items — is computed BEFORE the grid barrier; behind it: loads, GroupNorm tables, patch, MFMAs.
template <int ST, int LGC4>
__device__ __forceinline__ bool sn_conv(const SnArgs &a, const SnPhase &p_lds, int pi, int mytile, float *lds, int *flag, unsigned long long *pf, int sn_wave) {
  constexpr int C4 = 1 << LGC4, CIN = 4 * C4, MB = sn_mblocks(C4), KS = SN_WAVES / MB, SPT = C4 / KS;
  constexpr int TH = MB == 4 ? 8 : 4, TW = MB == 1 ? 4 : 8, LGTW = MB == 1 ? 2 : 3;
  constexpr int PH = (TH - 1) * ST + 3, PW = (TW - 1) * ST + 3, CS = CIN + 4, ITEMS = PH * PW * C4;
  constexpr int NIT = (ITEMS + SN_THREADS - 1) / SN_THREADS, NSTEP = 9 * SPT;
  float *sc_in = lds, *sh_in = lds + SN_TAB, *sc_res = lds + 2 * SN_TAB, *sh_res = lds + 3 * SN_TAB;
  float *red = lds + 4 * SN_TAB;
  float *patch = lds + SN_FIXED_FLOATS;
  int tid = SN_TID;
  asm volatile("" : "+v"(tid));             // (per-thread constants of a phase stay inside the phase: no hoisting out of the phase loop)
  const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const SnRegs r = sn_regs(&p_lds, lane);   // (read here, inside the variant: held across the variant switch it is spilled to scratch)
  const int mb = w % MB, ks = w / MB;
  // the phase's parameters: lanes of the descriptor registers -> scalar registers
  const int Hin = SN_I(r, Hin), Win = SN_I(r, Win), Ho = SN_I(r, Ho), Wo = SN_I(r, Wo), coutp = SN_I(r, coutp), ntiles = SN_I(r, ntiles);
  const int mode = SN_I(r, in_mode), tiles_x = SN_I(r, tiles_x), out_cpg = SN_I(r, out_cpg), out_G = SN_I(r, out_G), out_slots = SN_I(r, out_slots);
  const float *p_in = SN_P(const float, r, in), *p_res = SN_P(const float, r, res), *p_w = SN_P(const float, r, w), *p_wds = SN_P(const float, r, w_ds);
  float *p_blk = SN_P(float, r, blk_out), *p_out = SN_P(float, r, out), *p_part = SN_P(float, r, part), *p_outds = SN_P(float, r, out_ds),
        *p_partds = SN_P(float, r, part_ds);
  const int *p_tiles = SN_P(const int, r, tiles);
  const sn_rsrc_t r_in = sn_rsrc(p_in), r_res = sn_rsrc(p_res), r_blk = sn_rsrc(p_blk);
  const bool has_ds = ST == 2 && p_wds != nullptr;

  int n = 0, nt = 0, txi = 0, tyi = 0;
  float b[NSTEP], bd[SPT];
  unsigned gofs[NIT];
  unsigned inimg = 0, own = 0;
  auto before = [&](int t) {                // tile decode, B fragments, item addresses: nothing here reads the previous phase's output
    const int tw = t == (int)blockIdx.x ? mytile : p_tiles[t];
    nt = tw & 255;
    txi = (tw >> 8) & 255;
    tyi = (tw >> 16) & 255;
    n = tw >> 24;
    // (weights are packed per wave in the order of its K walk, four steps per lane and load: sn_pack_conv)
    const sn_f32x4 *wp = reinterpret_cast<const sn_f32x4 *>(p_w) + ((size_t)(nt * KS + ks) * (NSTEP / 4)) * 64 + lane;
#pragma unroll
    for (int j = 0; j < NSTEP / 4; ++j) {
      const sn_f32x4 q = wp[j * 64];
      b[4 * j] = q[0];
      b[4 * j + 1] = q[1];
      b[4 * j + 2] = q[2];
      b[4 * j + 3] = q[3];
    }
    if (ST == 2) {
      if (has_ds) {
        const sn_f32x4 *wd = reinterpret_cast<const sn_f32x4 *>(p_wds) + ((size_t)(nt * KS + ks) * (SPT / 4)) * 64 + lane;
#pragma unroll
        for (int j = 0; j < SPT / 4; ++j) {
          const sn_f32x4 q = wd[j * 64];
          bd[4 * j] = q[0];
          bd[4 * j + 1] = q[1];
          bd[4 * j + 2] = q[2];
          bd[4 * j + 3] = q[3];
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    SN_STAMP(7);
    const int iy0 = tyi * (TH * ST) - 1, ix0 = txi * (TW * ST) - 1;
    inimg = own = 0;
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
      const int it = tid + j * SN_THREADS;
      const int c4 = it & (C4 - 1), pp = it >> LGC4;
      const int py = pp / PW, px = pp - py * PW;
      const int iy = iy0 + py, ix = ix0 + px;
      gofs[j] = 0;
      if ((j + 1 < NIT || it < ITEMS) && (unsigned)iy < (unsigned)Hin && (unsigned)ix < (unsigned)Win) {
        inimg |= 1u << j;
        gofs[j] = (unsigned)(((n * Hin + iy) * Win + ix) * CIN + c4 * 4);
        if (nt == 0 && py >= 1 && py <= ST * TH && px >= 1 && px <= ST * TW) own |= 1u << j;
      }
    }
  };
  const bool active = (int)blockIdx.x < ntiles;
  if (active) before(blockIdx.x);
  int aofs;
  {
    const int pl = mb * 16 + (lane & 15);
    const int ty = pl >> LGTW, tx = pl & (TW - 1);
    aofs = ((ty * ST) * PW + tx * ST) * CS + (lane >> 4) + ks * 4;
  }
  SN_STAMP(5);
  if (pi > 0 && !sn_grid_wait(a.bar, a.bar_base + (unsigned)pi * gridDim.x, a.err, flag, tid)) return false;
  unsigned long long t0 = 0;
  if (a.prof != nullptr && tid == 0) {
    t0 = wall_clock64();
    if (blockIdx.x == 0) a.prof[pi * 4 + 0] = t0;
  }

  int cur_n = -1;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    if (t != (int)blockIdx.x) {
      __syncthreads();                       // the previous tile's patch / partial reads are done
      before(t);
    }
    // ---- (1) this thread's patch items: issue the loads (nothing else is scheduled in front of them)
    __builtin_amdgcn_sched_barrier(0);
    sn_f32x4 va[NIT], vr[NIT];
#pragma unroll
    for (int j = 0; j < NIT; ++j) {          // (unconditional: out-of-image items read offset 0 and are zeroed below)
      va[j] = sn_ld4(r_in, gofs[j]);
      vr[j] = sn_f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (mode >= 2) {
#pragma unroll
      for (int j = 0; j < NIT; ++j) vr[j] = sn_ld4(r_res, gofs[j]);
    }
    __builtin_amdgcn_sched_barrier(0);
    SN_STAMP(0);
    // ---- (2) GroupNorm tables of this sample (their slot loads fly together with the loads above)
    if (n != cur_n) {
      if (mode >= 1) {
        sn_gn_tables(r, sc_in, sh_in, mode == 3, sc_res, sh_res, n, tid);
        __syncthreads();
      }
      cur_n = n;
    }
    SN_STAMP(1);
    // ---- (3) transform, patch into LDS (zero padding applies to the TRANSFORMED activation), block output by its owner
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
      const int it = tid + j * SN_THREADS;
      if (j + 1 < NIT || it < ITEMS) {
        const int c4 = it & (C4 - 1), pp = it >> LGC4;
        sn_f32x4 v = va[j];
        if (!((inimg >> j) & 1u)) v = sn_f32x4{0.f, 0.f, 0.f, 0.f};
        if (mode >= 1 && ((inimg >> j) & 1u)) {
          const sn_f32x4 sa = *reinterpret_cast<const sn_f32x4 *>(sc_in + c4 * 4);
          const sn_f32x4 sb = *reinterpret_cast<const sn_f32x4 *>(sh_in + c4 * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(v[e], sa[e], sb[e]);
          if (mode >= 2) {
            sn_f32x4 rr = vr[j];
            if (mode == 3) {
              const sn_f32x4 a2 = *reinterpret_cast<const sn_f32x4 *>(sc_res + c4 * 4);
              const sn_f32x4 b2 = *reinterpret_cast<const sn_f32x4 *>(sh_res + c4 * 4);
#pragma unroll
              for (int e = 0; e < 4; ++e) rr[e] = __builtin_fmaf(rr[e], a2[e], b2[e]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += rr[e];
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
          if (mode >= 2 && p_blk != nullptr && ((own >> j) & 1u)) sn_st4(r_blk, gofs[j], v);
        }
        *reinterpret_cast<sn_f32x4 *>(patch + pp * CS + c4 * 4) = v;
      }
    }
    __syncthreads();
    SN_STAMP(2);
    // ---- (4) K loop: this wave's M-block, its SPT K-steps of every tap
    sn_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f}, accd = {0.f, 0.f, 0.f, 0.f};
    {
      const float *ap = patch + aofs;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
        for (int i = 0; i < SPT; ++i) {
          const float av = ap[((tap / 3) * PW + tap % 3) * CS + i * KS * 4];
          if ((tap * SPT + i) & 1)
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[tap * SPT + i], acc1, 0, 0, 0);
          else
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[tap * SPT + i], acc0, 0, 0, 0);
        }
      }
      if (ST == 2) {
        if (has_ds) {                        // 1x1 stride-2 conv of the skip branch: the centre tap of the same patch
#pragma unroll
          for (int i = 0; i < SPT; ++i) accd = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[(PW + 1) * CS + i * KS * 4], bd[i], accd, 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) acc0[e] += acc1[e];
    *reinterpret_cast<sn_f32x4 *>(red + (w * 64 + lane) * 4) = acc0;
    if (has_ds) *reinterpret_cast<sn_f32x4 *>(red + SN_WAVES * 256 + (w * 64 + lane) * 4) = accd;
    SN_STAMP(3);
    __syncthreads();
    SN_STAMP(4);
    // ---- (5) epilogue: wave (conv, M-block) sums the K-split partials in wave order, stores, writes the group partials
    const int nconv = has_ds ? 2 : 1;
    for (int slot_w = w; slot_w < MB * nconv; slot_w += SN_WAVES) {
      const int which = slot_w / MB, mbe = slot_w % MB;
      sn_f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < KS; ++k) {
        const sn_f32x4 u = *reinterpret_cast<const sn_f32x4 *>(red + which * (SN_WAVES * 256) + ((k * MB + mbe) * 64 + lane) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += u[e];
      }
      const sn_rsrc_t r_out = sn_rsrc(which ? p_outds : p_out), r_part = sn_rsrc(which ? p_partds : p_part);
      const int col = lane & 15, rg = lane >> 4;
      const int oy0 = tyi * TH, ox0 = txi * TW;
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int pl = mbe * 16 + rg * 4 + e;
        const int oy = oy0 + (pl >> LGTW), ox = ox0 + (pl & (TW - 1));
        if (oy < Ho && ox < Wo) {
          sn_st1(r_out, (unsigned)(((n * Ho + oy) * Wo + ox) * coutp + nt * 16 + col), v[e]);
          s1 += v[e];
          s2 = __builtin_fmaf(v[e], v[e], s2);
        }
      }
      s1 += __shfl_xor(s1, 16);
      s2 += __shfl_xor(s2, 16);
      s1 += __shfl_xor(s1, 32);
      s2 += __shfl_xor(s2, 32);
      const int cpg = out_cpg, tile_m = tyi * tiles_x + txi;
      const int span = cpg < 16 ? cpg : 16;
      for (int o = 1; o < span; o <<= 1) {
        s1 += __shfl_xor(s1, o);
        s2 += __shfl_xor(s2, o);
      }
      if (rg == 0 && (col & (span - 1)) == 0) {
        int g, slot;
        if (cpg < 16) {
          g = (nt * 16 + col) / cpg;
          slot = tile_m * MB + mbe;
        } else {
          const int ntpg = cpg >> 4;
          g = nt / ntpg;
          slot = (tile_m * MB + mbe) * ntpg + (nt - g * ntpg);
        }
        float2 pr;
        pr.x = s1;
        pr.y = s2;
        sn_st2(r_part, (unsigned)((n * out_G + g) * out_slots + slot) * 2u, pr);
      }
    }
  }
  if (a.prof != nullptr) {
    __syncthreads();
    if (tid == 0) {
      const unsigned long long t1 = wall_clock64();
      if (blockIdx.x == 0) a.prof[pi * 4 + 1] = t1;
      atomicMax(&a.prof[pi * 4 + 2], t1 - t0);
    }
  }
  return true;
}

// GroupNorm + ReLU + MaxPool2d(3, 2, 1) of the stem output (resnet.py:165-168); the affine transform goes BEFORE the max
__device__ void sn_pool_phase(const SnPhase &p, int B, float *lds, int sn_wave) {
  float *sc = lds, *sh = lds + SN_TAB;
  int tid = SN_TID;
  asm volatile("" : "+v"(tid));
  const SnRegs r = sn_regs(&p, tid & 63);
  for (int n = 0; n < B; ++n) sn_gn_tables(r, sc + n * p.cinp, sh + n * p.cinp, false, nullptr, nullptr, n, tid);
  __syncthreads();
  const int Q = p.cinp >> 2;
  const int total = B * p.Ho * p.Wo * Q;
  const sn_rsrc_t r_out = sn_rsrc(sn_uni(p.out));
  for (int g = blockIdx.x * SN_THREADS + tid; g < total; g += gridDim.x * SN_THREADS) {
    const int q = g % Q;
    int r = g / Q;
    const int wo = r % p.Wo;
    r /= p.Wo;
    const int ho = r % p.Ho;
    const int n = r / p.Ho;
    const sn_f32x4 a = *reinterpret_cast<const sn_f32x4 *>(sc + n * p.cinp + 4 * q);
    const sn_f32x4 b = *reinterpret_cast<const sn_f32x4 *>(sh + n * p.cinp + 4 * q);
    sn_f32x4 v[9];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int hi2 = 2 * ho - 1 + kh, wi = 2 * wo - 1 + kw;
        // out-of-image taps read the window's centre again (always inside): max is idempotent
        const bool ok = (unsigned)hi2 < (unsigned)p.Hin && (unsigned)wi < (unsigned)p.Win;
        const int hh = ok ? hi2 : 2 * ho, ww = ok ? wi : 2 * wo;
        v[kh * 3 + kw] = *reinterpret_cast<const sn_f32x4 *>(p.in + (unsigned)(((n * p.Hin + hh) * p.Win + ww) * p.cinp + 4 * q));
      }
    sn_f32x4 m = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], __builtin_fmaf(v[k][e], a[e], b[e]));
    sn_st4(r_out, (unsigned)g * 4u, m);
  }
}

// Linear layer: y[b][o] = act(bias[row(b)][o] + sum_k W[o][k] x[b][k]), x = the input transform of [B][K] activations.
// Tile = 4 outputs; wave (output ol = w & 3, K part kq = w >> 2); its first SN_WPRE weight vectors per lane are fetched before
// the grid barrier.
__device__ __forceinline__ bool sn_linear(const SnArgs &a, const SnPhase &p, int pi, float *lds, int *flag, int sn_wave) {
  float *sc = lds, *sh = lds + SN_TAB;
  float *red = lds + 4 * SN_TAB;
  float *xs = lds + SN_FIXED_FLOATS;
  const int B = a.B, K = p.K, K4 = K >> 2;
  int tid = SN_TID;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Kq4 = (K4 + SN_KP - 1) / SN_KP;
  const int ol = w & 3, kq = w >> 2;
  const int lo = kq * Kq4, hi = lo + Kq4 < K4 ? lo + Kq4 : K4;
  sn_f32x4 wpre[SN_WPRE];
  {
    const int o = blockIdx.x * 4 + ol;
#pragma unroll
    for (int j = 0; j < SN_WPRE; ++j) {
      const int idx = lo + lane + 64 * j;
      const int oc = o < p.cout ? o : 0, ic = idx < hi ? idx : lo;
      wpre[j] = reinterpret_cast<const sn_f32x4 *>(p.w + (size_t)oc * K)[ic];
    }
  }
  if (pi > 0 && !sn_grid_wait(a.bar, a.bar_base + (unsigned)pi * gridDim.x, a.err, flag, tid)) return false;
  unsigned long long t0 = 0;
  if (a.prof != nullptr && tid == 0) {
    t0 = wall_clock64();
    if (blockIdx.x == 0) a.prof[pi * 4 + 0] = t0;
  }
  const sn_rsrc_t r_lin = sn_rsrc(sn_uni(p.in));
  if ((int)blockIdx.x < p.ntiles) {
    if (p.in_mode == 1) {
      const SnRegs r = sn_regs(&p, tid & 63);
      for (int n = 0; n < B; ++n) sn_gn_tables(r, sc + n * p.cinp, sh + n * p.cinp, false, nullptr, nullptr, n, tid);
      __syncthreads();
    }
    {
      const int C4 = p.cinp >> 2;
      for (int it = tid; it < B * K4; it += SN_THREADS) {
        sn_f32x4 v = sn_ld4(r_lin, (unsigned)it * 4u);
        if (p.in_mode == 1) {
          const int n = it / K4, c4 = (it - n * K4) & (C4 - 1);
          const sn_f32x4 s = *reinterpret_cast<const sn_f32x4 *>(sc + n * p.cinp + c4 * 4);
          const sn_f32x4 h = *reinterpret_cast<const sn_f32x4 *>(sh + n * p.cinp + c4 * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(__builtin_fmaf(v[e], s[e], h[e]), 0.f);
        }
        reinterpret_cast<sn_f32x4 *>(xs)[it] = v;
      }
    }
    __syncthreads();
    for (int t = blockIdx.x; t < p.ntiles; t += gridDim.x) {
      const int o = t * 4 + ol;
      float acc[SN_MAXB];
#pragma unroll
      for (int b = 0; b < SN_MAXB; ++b) acc[b] = 0.f;
      if (o < p.cout) {
        const sn_f32x4 *wrow = reinterpret_cast<const sn_f32x4 *>(p.w + (size_t)o * K);
        const bool pre = t == (int)blockIdx.x;
#pragma unroll
        for (int j = 0; j < SN_WPRE; ++j) {
          const int idx = lo + lane + 64 * j;
          if (idx < hi) {
            const sn_f32x4 wv = pre ? wpre[j] : wrow[idx];
#pragma unroll
            for (int b = 0; b < SN_MAXB; ++b)
              if (b < B) {
                const sn_f32x4 xv = reinterpret_cast<const sn_f32x4 *>(xs)[b * K4 + idx];
                acc[b] = __builtin_fmaf(wv[0], xv[0], acc[b]);
                acc[b] = __builtin_fmaf(wv[1], xv[1], acc[b]);
                acc[b] = __builtin_fmaf(wv[2], xv[2], acc[b]);
                acc[b] = __builtin_fmaf(wv[3], xv[3], acc[b]);
              }
          }
        }
        for (int idx = lo + lane + 64 * SN_WPRE; idx < hi; idx += 64) {
          const sn_f32x4 wv = wrow[idx];
#pragma unroll
          for (int b = 0; b < SN_MAXB; ++b)
            if (b < B) {
              const sn_f32x4 xv = reinterpret_cast<const sn_f32x4 *>(xs)[b * K4 + idx];
              acc[b] = __builtin_fmaf(wv[0], xv[0], acc[b]);
              acc[b] = __builtin_fmaf(wv[1], xv[1], acc[b]);
              acc[b] = __builtin_fmaf(wv[2], xv[2], acc[b]);
              acc[b] = __builtin_fmaf(wv[3], xv[3], acc[b]);
            }
        }
      }
#pragma unroll
      for (int b = 0; b < SN_MAXB; ++b) {
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) acc[b] += __shfl_xor(acc[b], s);
      }
      __syncthreads();
      if (lane == 0) {
#pragma unroll
        for (int b = 0; b < SN_MAXB; ++b) red[w * SN_MAXB + b] = acc[b];
      }
      __syncthreads();
      if (tid < 4 * B) {
        const int ol2 = tid & 3, b = tid >> 2, o2 = t * 4 + ol2;
        if (o2 < p.cout) {
          float v = 0.f;
#pragma unroll
          for (int k = 0; k < SN_KP; ++k) v += red[(k * 4 + ol2) * SN_MAXB + b];
          if (p.bias != nullptr) v += p.bias[(size_t)(a.bias_row != nullptr && p.use_row ? a.bias_row[b] : 0) * p.cout + o2];
          if (p.relu_out) v = fmaxf(v, 0.f);
          if (p.out != nullptr)
            sn_st1(sn_rsrc(p.out), (unsigned)(b * p.coutp + o2), v);   // the hidden vector: read by the head phase of other workgroups
          else
            a.final_out[(size_t)b * p.coutp + o2] = v;
        }
      }
    }
  }
  if (a.prof != nullptr) {
    __syncthreads();
    if (tid == 0) {
      const unsigned long long t1 = wall_clock64();
      if (blockIdx.x == 0) a.prof[pi * 4 + 1] = t1;
      atomicMax(&a.prof[pi * 4 + 2], t1 - t0);
    }
  }
  return true;
}

__global__ __launch_bounds__(SN_THREADS) void smallnet_kernel(SnArgs a) {
  extern __shared__ __align__(16) float sn_lds[];
  __shared__ int flag;
  // the phase table and this workgroup's first tile of every phase: global -> LDS once (nothing behind a grid barrier waits for
  // a descriptor or a tile list to arrive from memory)
  const int sn_wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  SnPhase *phl = reinterpret_cast<SnPhase *>(sn_lds + 4 * SN_TAB + SN_RED_FLOATS);
  int *mytiles = reinterpret_cast<int *>(sn_lds + 4 * SN_TAB + SN_RED_FLOATS + SN_MAXPH * 72);
  {
    const int nd = a.nph * (int)(sizeof(SnPhase) / 4);
    const int *src = reinterpret_cast<const int *>(a.ph);
    int *dst = reinterpret_cast<int *>(phl);
    for (int i = threadIdx.x; i < nd; i += SN_THREADS) dst[i] = src[i];
    __syncthreads();
    if ((int)threadIdx.x < a.nph) {
      const SnPhase &q = phl[threadIdx.x];
      mytiles[threadIdx.x] = (q.kind == 0 && (int)blockIdx.x < q.ntiles) ? q.tiles[blockIdx.x] : 0;
    }
    __syncthreads();
  }
  if (a.dbg >= 2) {                          // timing experiment: the grid barriers alone
    for (int pi = 0; pi < a.nph; ++pi) {
      if (pi > 0) {
        __syncthreads();
        if (threadIdx.x == 0) {
          if (a.dbg == 2) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
          __hip_atomic_fetch_add(a.bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          while ((int)(__hip_atomic_load(a.bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - (a.bar_base + (unsigned)pi * gridDim.x)) < 0)
            __builtin_amdgcn_s_sleep(1);
          if (a.dbg == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
      }
      if (threadIdx.x == 0 && blockIdx.x == 0) a.prof[pi * 4 + 0] = a.prof[pi * 4 + 1] = wall_clock64();
    }
    return;
  }
  for (int pi = 0; pi < a.nph; ++pi) {
    const SnPhase &p = phl[pi];
    const int mytile = sn_uni(mytiles[pi]);
    unsigned long long *pf = a.prof != nullptr ? a.prof + 256 + pi * 8 : nullptr;
    bool ok = true;
    const int kind = sn_uni(p.kind);
    if (kind == 0) {
      switch (sn_uni(p.type)) {
        case 8 + 3: ok = sn_conv<1, 3>(a, p, pi, mytile, sn_lds, &flag, pf, sn_wave); break;
        case 8 + 4: ok = sn_conv<1, 4>(a, p, pi, mytile, sn_lds, &flag, pf, sn_wave); break;
        case 8 + 5: ok = sn_conv<1, 5>(a, p, pi, mytile, sn_lds, &flag, pf, sn_wave); break;
        case 8 + 6: ok = sn_conv<1, 6>(a, p, pi, mytile, sn_lds, &flag, pf, sn_wave); break;
        case 16 + 3: ok = sn_conv<2, 3>(a, p, pi, mytile, sn_lds, &flag, pf, sn_wave); break;
        case 16 + 4: ok = sn_conv<2, 4>(a, p, pi, mytile, sn_lds, &flag, pf, sn_wave); break;
        case 16 + 5: ok = sn_conv<2, 5>(a, p, pi, mytile, sn_lds, &flag, pf, sn_wave); break;
        default: ok = sn_conv<2, 6>(a, p, pi, mytile, sn_lds, &flag, pf, sn_wave); break;
      }
    } else if (kind == 1) {
      if (a.prof != nullptr && SN_TID == 0 && blockIdx.x == 0) a.prof[pi * 4 + 0] = wall_clock64();
      sn_pool_phase(p, a.B, sn_lds, sn_wave);
      if (a.prof != nullptr && SN_TID == 0 && blockIdx.x == 0) a.prof[pi * 4 + 1] = wall_clock64();
    } else {
      ok = sn_linear(a, p, pi, sn_lds, &flag, sn_wave);
    }
    if (!ok) {
      for (int i = SN_TID; i < a.final_n; i += SN_THREADS) a.final_out[i] = __builtin_nanf("");
      return;
    }
    if (pi + 1 < a.nph) {
      sn_grid_arrive(a.bar, SN_TID);
      if (a.prof != nullptr && SN_TID == 0 && blockIdx.x == 0) a.prof[256 + (pi + 1) * 8 + 6] = wall_clock64();
    }
  }
}

// Weights of a conv phase in the order sn_conv's waves walk K: [cout tile nt][wave ks][step group j/4][lane][4], step
// j = tap * SPT + i of wave ks being K-step (tap, cq = ks + i * KS); lane (col = lane & 15, kk = lane >> 4) holds
// W[nt*16 + col][cq*4 + kk][tap].  k = 1 (the downsample conv): one tap.
void sn_pack_conv(const float *oihw, int cout, int cin, int cinp, int k, std::vector<float> &out) {
  const int T = k * k, C4 = cinp / 4, NT = (cout + 15) / 16;
  const int KS = SN_WAVES / sn_mblocks(C4), SPT = C4 / KS, NSTEP = T * SPT;
  out.assign((size_t)NT * KS * NSTEP * 64, 0.f);
  for (int nt = 0; nt < NT; ++nt)
    for (int ks = 0; ks < KS; ++ks)
      for (int j = 0; j < NSTEP; ++j)
        for (int lane = 0; lane < 64; ++lane) {
          const int tap = j / SPT, cq = ks + (j % SPT) * KS;
          const int co = nt * 16 + (lane & 15), ci = cq * 4 + (lane >> 4);
          if (co < cout && ci < cin)
            out[((((size_t)(nt * KS + ks) * (NSTEP / 4) + j / 4) * 64) + lane) * 4 + j % 4] = oihw[((size_t)co * cin + ci) * T + tap];
        }
}

int ilog2(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

}  // namespace

// -------------------------------------------------------------------------------------------------------------------------
struct SmallNet {
  unsigned long long load_gen = ~0ull;
  std::vector<float *> w;            // per conv of m->convs (index 0 unused), fragment-packed
  float *w_fc = nullptr, *w_head = nullptr;
  float *part[4] = {nullptr, nullptr, nullptr, nullptr};   // conv1 / conv2 / downsample / compression partial statistics
  size_t part_floats[4] = {0, 0, 0, 0};
  SnPhase *ph_dev = nullptr;
  int *tiles_dev = nullptr;
  size_t tiles_cap = 0;
  std::vector<SnPhase> ph;
  int B = -1, grid = 0, stem_slots = -1;
  bool features = false;             // built for pnvo_forward_features (no head phase)
  const void *ws_key[4] = {nullptr, nullptr, nullptr, nullptr};
  size_t lds_bytes = 0;
  unsigned *bar = nullptr;
  unsigned bar_base = 0;
  int *err = nullptr;                // host-mapped
  bool unsupported = false, failed = false;
  int cus = 0;
  bool attr_set = false;
  unsigned long long *prof = nullptr;   // device
};

}  // namespace pnvo

using namespace pnvo;

void pnvo_small_free(pnvo_handle m) {
  SmallNet *sn = static_cast<SmallNet *>(m->small);
  if (!sn) return;
  for (float *&q : sn->w) pnvo_free_dev(q);
  pnvo_free_dev(sn->w_fc);
  pnvo_free_dev(sn->w_head);
  for (float *&q : sn->part) pnvo_free_dev(q);
  if (sn->ph_dev) (void)hipFree(sn->ph_dev);
  if (sn->tiles_dev) (void)hipFree(sn->tiles_dev);
  if (sn->bar) (void)hipFree(sn->bar);
  if (sn->err) (void)hipHostFree(sn->err);
  if (sn->prof) (void)hipFree(sn->prof);
  delete sn;
  m->small = nullptr;
}

// Is this call shape one the persistent kernel takes?  (BasicBlock backbones whose channel counts are multiples of 32, batch
// <= small_max, no taps / per-launch timing / training forward; otherwise the per-layer launches run.)
bool pnvo_small_usable(pnvo_handle m, int B) {
  // small_max is quoted for the VO model's 48 x 86 pooled map (4 pairs: where the per-layer launches catch up); a model on smaller
  // frames — the navigation policy's encoder works on 96 x 170 depth, a 24 x 43 pooled map — has a quarter of the work per sample,
  // and the same amount of work is that many more samples (up to SN_MAXB)
  const long px = (long)m->Hp * m->Wp;
  const long bmax = std::min<long>(SN_MAXB, std::max<long>(m->opt.small_max, px > 0 ? (long)m->opt.small_max * (48 * 86) / px : 0));
  if (!m->opt.small_net || B < 1 || B > bmax) return false;
  if (m->bottleneck || m->tap_dst != nullptr || m->train != nullptr || m->graph_mode > 0) return false;
  if (m->precision != 0) return false;
  if (m->opt.conv != 0 || !m->opt.tail || !m->opt.pool || m->opt.conv3_nt) return false;   // an explicit kernel selection is honoured
  SmallNet *sn = static_cast<SmallNet *>(m->small);
  if (sn && (sn->unsupported || sn->failed)) return false;
  const pnvo_config &c = m->cfg;
  if (c.baseplanes % 32 != 0 || c.hidden % 4 != 0) return false;
  for (size_t k = 1; k < m->convs.size(); ++k)
    if (m->convs[k].cinp > 256 || (m->convs[k].cout != m->convs[k].coutp && k + 1 != m->convs.size())) return false;
  for (int st = 0; st < 4; ++st)
    if (m->nblocks[st] < 2) return false;
  if (B * std::max(m->convs[0].coutp, m->comp_cp) > SN_TAB) return false;
  if (m->comp_cp != 32 && m->comp_cp != 64 && m->comp_cp != 128) return false;
  // LDS: the largest patch of a conv phase, or the activation vectors of the Linear layers for the whole batch (large frames)
  size_t floats = (size_t)B * std::max((size_t)m->fh * m->fw * m->comp_cp, (size_t)c.hidden);
  for (size_t k = 1; k < m->convs.size(); ++k) {
    const Layer &l = m->convs[k];
    if (l.k != 3) continue;
    const int mb = sn_mblocks(l.cinp / 4), th = mb == 4 ? 8 : 4, tw = mb == 1 ? 4 : 8;
    floats = std::max(floats, (size_t)((th - 1) * l.stride + 3) * ((tw - 1) * l.stride + 3) * (l.cinp + 4));
  }
  if (((size_t)SN_FIXED_FLOATS + floats) * sizeof(float) > (size_t)156 * 1024) return false;
  if (1 + (int)m->convs.size() - 1 + 2 > SN_MAXPH) return false;
  return true;
}

namespace {

int sn_upload(pnvo_handle m, float *&dst, const std::vector<float> &v) {
  pnvo_free_dev(dst);
  HIPCHK(m, hipMalloc((void **)&dst, v.size() * sizeof(float)));
  HIPCHK(m, hipMemcpy(dst, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
  return PNVO_OK;
}

int sn_pack_weights(pnvo_handle m, SmallNet *sn) {
  int rc;
  sn->w.resize(m->convs.size(), nullptr);
  std::vector<float> pk;
  for (size_t k = 1; k < m->convs.size(); ++k) {
    const Layer &l = m->convs[k];
    sn_pack_conv(l.host_w.data(), l.cout, l.cin, l.cinp, l.k, pk);
    if ((rc = sn_upload(m, sn->w[k], pk)) != PNVO_OK) return rc;
  }
  {
    // Linear(flat -> hidden): column c*P + p of the reference's NCHW flatten becomes K index p*comp_cp + c of the NHWC map
    const int P = m->fh * m->fw, CP = m->comp_cp, hid = m->cfg.hidden, flat = m->comp_c * P;
    pk.assign((size_t)hid * P * CP, 0.f);
    for (int o = 0; o < hid; ++o)
      for (int c = 0; c < m->comp_c; ++c)
        for (int p = 0; p < P; ++p) pk[((size_t)o * P + p) * CP + c] = m->fc.host_w[(size_t)o * flat + (size_t)c * P + p];
    if ((rc = sn_upload(m, sn->w_fc, pk)) != PNVO_OK) return rc;
    if ((rc = sn_upload(m, sn->w_head, m->head.host_w)) != PNVO_OK) return rc;
  }
  sn->load_gen = m->load_gen;
  return PNVO_OK;
}

// Tile geometry of a conv phase (fixed by the input channel count: see sn_conv) and its tile list.
struct SnGeom {
  int MB, TH, TW, PH, PW, CS;
};
SnGeom sn_geom(const Layer &l) {
  SnGeom g;
  g.MB = sn_mblocks(l.cinp / 4);
  g.TH = g.MB == 4 ? 8 : 4;
  g.TW = g.MB == 1 ? 4 : 8;
  g.PH = (g.TH - 1) * l.stride + 3;
  g.PW = (g.TW - 1) * l.stride + 3;
  g.CS = l.cinp + 4;
  return g;
}

SnGN sn_gn(const Layer &l, const float *part, int slots) {
  SnGN g;
  std::memset(&g, 0, sizeof(g));
  g.part = part;
  g.gamma = l.gamma;
  g.beta = l.beta;
  g.G = l.groups;
  g.cpg = l.coutp / l.groups;
  g.lgcpg = ilog2(g.cpg);
  g.C = l.cout;
  g.CP = l.coutp;
  g.slots = slots;
  g.layout = 0;
  g.inv_cnt = (float)(1.0 / ((double)l.hout * l.wout * ((double)l.cout / l.groups)));
  return g;
}

int sn_build(pnvo_handle m, SmallNet *sn, int B) {
  const pnvo_config &c = m->cfg;
  std::vector<SnPhase> ph;
  size_t need[4] = {0, 0, 0, 0};
  size_t patch_floats = 0;
  std::vector<int> tile_words;               // all phases' tile lists, uploaded as one buffer
  auto conv_phase = [&](const Layer &l, int which_part) {
    SnPhase p;
    std::memset(&p, 0, sizeof(p));
    p.kind = 0;
    p.type = l.stride * 8 + ilog2(l.cinp / 4);
    p.cinp = l.cinp;
    p.coutp = l.coutp;
    p.cout = l.cout;
    p.Hin = l.hin;
    p.Win = l.win;
    p.Ho = l.hout;
    p.Wo = l.wout;
    const SnGeom g = sn_geom(l);
    p.tiles_y = (l.hout + g.TH - 1) / g.TH;
    p.tiles_x = (l.wout + g.TW - 1) / g.TW;
    p.NT = (l.cout + 15) / 16;
    p.ntiles = B * p.tiles_y * p.tiles_x * p.NT;
    p.tiles = reinterpret_cast<const int *>(tile_words.size() + 1);   // (offset + 1 until the buffer exists: fixed up below)
    for (int n = 0; n < B; ++n)
      for (int ty = 0; ty < p.tiles_y; ++ty)
        for (int tx = 0; tx < p.tiles_x; ++tx)
          for (int nt = 0; nt < p.NT; ++nt) tile_words.push_back(n << 24 | ty << 16 | tx << 8 | nt);
    p.out_G = l.groups;
    p.out_cpg = l.coutp / l.groups;
    p.out_slots = p.tiles_y * p.tiles_x * g.MB * std::max(1, p.out_cpg / 16);
    need[which_part] = std::max(need[which_part], (size_t)B * p.out_G * p.out_slots * 2);
    patch_floats = std::max(patch_floats, (size_t)g.PH * g.PW * g.CS);
    return p;
  };
  // ---- phase 0: GroupNorm + ReLU + max-pool of the stem output
  const Layer &stem = m->convs[0];
  {
    SnPhase p;
    std::memset(&p, 0, sizeof(p));
    p.kind = 1;
    p.cinp = stem.coutp;
    p.Hin = m->Hs;
    p.Win = m->Ws;
    p.Ho = m->Hp;
    p.Wo = m->Wp;
    p.in = m->stem_raw;
    p.out = m->bufY[0];
    p.gin = sn_gn(stem, m->stats, m->stem_slots_out);
    p.gin.layout = 1;
    p.ntiles = sn->cus;
    ph.push_back(p);
  }
  // ---- residual stages
  size_t li = 1;
  float *X = m->bufY[0], *Xn = m->bufY[1];
  bool have_tail = false;
  SnGN tail_gin, tail_gres;
  const float *tail_in = nullptr, *tail_res = nullptr;
  std::memset(&tail_gin, 0, sizeof(tail_gin));
  std::memset(&tail_gres, 0, sizeof(tail_gres));
  auto apply_input = [&](SnPhase &p) {       // the phase's input is the running block output: final (first block) or a pending tail
    if (have_tail) {
      p.in_mode = tail_gres.G > 0 ? 3 : 2;
      p.in = tail_in;
      p.gin = tail_gin;
      p.res = tail_res;
      p.gres = tail_gres;
      p.blk_out = Xn;
    } else {
      p.in_mode = 0;
      p.in = X;
    }
  };
  for (int stage = 1; stage <= 4; ++stage)
    for (int bi = 0; bi < m->nblocks[stage - 1]; ++bi) {
      const Layer &c1 = m->convs[li];
      const Layer &c2 = m->convs[li + 1];
      const bool ds = li + 2 < m->convs.size() && m->convs[li + 2].name.find("downsample") != std::string::npos;
      SnPhase p1 = conv_phase(c1, 0);
      apply_input(p1);
      if (have_tail) {
        if (ds && tail_gres.G > 0) return -1;      // would read and write the downsample buffers in one phase
        std::swap(X, Xn);                          // X now names the buffer this phase's owners write
        have_tail = false;
      }
      p1.w = sn->w[li];
      p1.out = m->rawA;
      p1.part = sn->part[0];
      if (ds) {
        const Layer &cd = m->convs[li + 2];
        if (cd.k != 1 || cd.stride != c1.stride || cd.cinp != c1.cinp || cd.coutp != c1.coutp || c1.stride != 2) return -1;
        p1.w_ds = sn->w[li + 2];
        p1.out_ds = m->rawD;
        p1.part_ds = sn->part[2];
        need[2] = std::max(need[2], (size_t)B * p1.out_G * p1.out_slots * 2);
      }
      ph.push_back(p1);
      SnPhase p2 = conv_phase(c2, 1);
      p2.in_mode = 1;
      p2.in = m->rawA;
      p2.gin = sn_gn(c1, sn->part[0], p1.out_slots);
      p2.w = sn->w[li + 1];
      p2.out = m->rawB;
      p2.part = sn->part[1];
      ph.push_back(p2);
      // the block's tail rides on the next phase
      have_tail = true;
      tail_in = m->rawB;
      tail_gin = sn_gn(c2, sn->part[1], p2.out_slots);
      if (ds) {
        tail_res = m->rawD;
        tail_gres = sn_gn(m->convs[li + 2], sn->part[2], p1.out_slots);
      } else {
        tail_res = X;
        std::memset(&tail_gres, 0, sizeof(tail_gres));
      }
      li += ds ? 3 : 2;
    }
  // ---- compression conv (GroupNorm(1, C) follows)
  const Layer &comp = m->convs[li];
  {
    SnPhase p = conv_phase(comp, 3);
    apply_input(p);
    p.blk_out = nullptr;
    p.w = sn->w[li];
    p.out = m->comp_raw;
    p.part = sn->part[3];
    ph.push_back(p);
  }
  const SnPhase &pc = ph.back();
  // ---- Linear + ReLU, output head
  {
    SnPhase p;
    std::memset(&p, 0, sizeof(p));
    p.kind = 2;
    p.cinp = m->comp_cp;
    p.K = m->fh * m->fw * m->comp_cp;
    p.cout = c.hidden;
    p.coutp = c.hidden;
    p.ntiles = (c.hidden + 3) / 4;
    p.in_mode = 1;
    p.in = m->comp_raw;
    p.gin = sn_gn(comp, sn->part[3], pc.out_slots);
    p.w = sn->w_fc;
    p.bias = m->fc_bias;
    p.relu_out = 1;
    p.use_row = 1;
    p.out = m->features_only ? nullptr : m->hid;   // pnvo_forward_features: the hidden vector IS the result (caller's tensor)
    ph.push_back(p);
    patch_floats = std::max(patch_floats, (size_t)B * p.K);
    SnPhase h;
    std::memset(&h, 0, sizeof(h));
    h.kind = 2;
    h.cinp = c.hidden;
    h.K = c.hidden;
    h.cout = c.out_dim;
    h.coutp = c.out_dim;
    h.ntiles = (c.out_dim + 3) / 4;
    h.in_mode = 0;
    h.in = m->hid;
    h.w = sn->w_head;
    h.bias = m->head_bias;
    h.relu_out = 0;
    h.out = nullptr;                               // the caller's tensor (SnArgs::final_out)
    if (!m->features_only) {
      ph.push_back(h);
      patch_floats = std::max(patch_floats, (size_t)B * h.K);
    }
  }
  // ---- buffers
  for (int k = 0; k < 4; ++k)
    if (need[k] > sn->part_floats[k]) {
      pnvo_free_dev(sn->part[k]);
      sn->part_floats[k] = 0;
      HIPCHK(m, hipMalloc((void **)&sn->part[k], need[k] * sizeof(float)));
      sn->part_floats[k] = need[k];
      return sn_build(m, sn, B);                   // pointers changed: lay the phases out again
    }
  sn->lds_bytes = ((size_t)SN_FIXED_FLOATS + patch_floats) * sizeof(float);
  if (sn->lds_bytes > (size_t)156 * 1024) return -1;
  int grid = 1;
  for (const SnPhase &p : ph)
    if (p.kind != 1) grid = std::max(grid, std::min(p.ntiles, sn->cus));
  sn->grid = grid;
  for (SnPhase &p : ph)
    if (p.kind == 1) p.ntiles = grid;
  if (tile_words.size() > sn->tiles_cap) {
    if (sn->tiles_dev) (void)hipFree(sn->tiles_dev);
    sn->tiles_dev = nullptr;
    sn->tiles_cap = 0;
    HIPCHK(m, hipMalloc((void **)&sn->tiles_dev, tile_words.size() * sizeof(int)));
    sn->tiles_cap = tile_words.size();
  }
  HIPCHK(m, hipMemcpy(sn->tiles_dev, tile_words.data(), tile_words.size() * sizeof(int), hipMemcpyHostToDevice));
  for (SnPhase &p : ph)
    if (p.kind == 0) p.tiles = sn->tiles_dev + (reinterpret_cast<size_t>(p.tiles) - 1);
  if (!sn->ph_dev) HIPCHK(m, hipMalloc((void **)&sn->ph_dev, 64 * sizeof(SnPhase)));
  if (ph.size() > (size_t)SN_MAXPH) return -1;
  HIPCHK(m, hipMemcpy(sn->ph_dev, ph.data(), ph.size() * sizeof(SnPhase), hipMemcpyHostToDevice));
  sn->ph = ph;
  sn->B = B;
  sn->features = m->features_only;
  sn->stem_slots = m->stem_slots_out;
  sn->ws_key[0] = m->stem_raw;
  sn->ws_key[1] = m->stats;
  sn->ws_key[2] = m->rawA;
  sn->ws_key[3] = m->fc_bias;
  return PNVO_OK;
}

}  // namespace

// Everything behind the stem conv (whose raw output is in m->stem_raw, its per-tile statistics in m->stats) for B pairs.
int pnvo_small_forward(pnvo_handle m, int B, const int64_t *actions, float *out, hipStream_t s) {
  SmallNet *sn = static_cast<SmallNet *>(m->small);
  if (!sn) {
    sn = new SmallNet();
    m->small = sn;
    hipDeviceProp_t prop;
    HIPCHK(m, hipGetDeviceProperties(&prop, m->device));
    sn->cus = prop.multiProcessorCount;
    HIPCHK(m, hipMalloc((void **)&sn->bar, 256));
    HIPCHK(m, hipMemset(sn->bar, 0, 256));
    HIPCHK(m, hipHostMalloc((void **)&sn->err, sizeof(int), hipHostMallocMapped | hipHostMallocCoherent));
    *(volatile int *)sn->err = 0;
  }
  if (*(volatile int *)sn->err != 0) {
    sn->failed = true;
    return pnvo_fail(m, PNVO_ERR_STATE, "an earlier small-batch forward of this handle did not complete (grid barrier timed out); "
                                        "set option small_net=off");
  }
  int rc;
  if (sn->load_gen != m->load_gen) {
    if ((rc = sn_pack_weights(m, sn)) != PNVO_OK) return rc;
    sn->B = -1;
  }
  if (sn->B != B || sn->features != m->features_only || sn->stem_slots != m->stem_slots_out || sn->ws_key[0] != m->stem_raw || sn->ws_key[1] != m->stats ||
      sn->ws_key[2] != m->rawA || sn->ws_key[3] != m->fc_bias) {
    HIPCHK(m, hipStreamSynchronize(s));            // the table of an in-flight launch is about to be rewritten
    rc = sn_build(m, sn, B);
    if (rc == -1) {
      sn->unsupported = true;
      return pnvo_fail(m, PNVO_ERR_STATE, "this model does not fit the small-batch kernel");
    }
    if (rc != PNVO_OK) return rc;
  }
  if (!sn->attr_set) {
    HIPCHK(m, hipFuncSetAttribute(reinterpret_cast<const void *>(smallnet_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
    sn->attr_set = true;
  }
  SnArgs a;
  a.ph = sn->ph_dev;
  a.nph = (int)sn->ph.size();
  a.B = B;
  a.bar = sn->bar;
  a.bar_base = sn->bar_base;
  a.err = sn->err;
  a.bias_row = actions;
  a.final_out = out;
  a.final_n = B * (m->features_only ? m->cfg.hidden : m->cfg.out_dim);
  a.prof = nullptr;
  a.dbg = m->opt.small_prof;
  if (m->opt.small_prof) {
    if (!sn->prof) HIPCHK(m, hipMalloc((void **)&sn->prof, 64 * 12 * sizeof(unsigned long long)));
    HIPCHK(m, hipMemsetAsync(sn->prof, 0, 64 * 12 * sizeof(unsigned long long), s));
    a.prof = sn->prof;
  }
  sn->bar_base += (unsigned)(a.nph - 1) * (unsigned)sn->grid;
  {
    PnvoTimed t(m, s, "smallnet", 0.0, 0.0);
    bool launched = false;
    if (m->opt.small_coop) {
      void *args[1] = {&a};
      const hipError_t ce = hipLaunchCooperativeKernel(reinterpret_cast<const void *>(smallnet_kernel), dim3((unsigned)sn->grid),
                                                       dim3(SN_THREADS), args, (unsigned)sn->lds_bytes, s);
      launched = ce == hipSuccess;
      if (!launched) {                             // a runtime without cooperative launches for this shape: plain launch from now on
        (void)hipGetLastError();
        m->opt.small_coop = 0;
        m->note = std::string("note: hipLaunchCooperativeKernel refused the small-batch kernel (") + hipGetErrorString(ce) +
                 "); this handle uses the plain launch (option small_coop = 0)";
      }
    }
    if (!launched) {
      hipLaunchKernelGGL(smallnet_kernel, dim3((unsigned)sn->grid), dim3(SN_THREADS), sn->lds_bytes, s, a);
      HIPCHK(m, hipGetLastError());
    }
  }
  if (a.prof != nullptr) {
    std::vector<unsigned long long> hp(64 * 12);
    HIPCHK(m, hipMemcpyAsync(hp.data(), sn->prof, hp.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    HIPCHK(m, hipStreamSynchronize(s));
    const unsigned long long z = hp[0];
    std::fprintf(stderr, "[pnvo] smallnet B=%d grid=%d lds=%zu: phase kind tiles | block0 start end (us) | max phase, max barrier wait (us)\n", B,
                 sn->grid, sn->lds_bytes);
    for (int pi = 0; pi < a.nph; ++pi)
    {
      std::fprintf(stderr, "[pnvo] smallnet %2d k%d t%2d %4d | %7.2f %7.2f | %6.2f %6.2f", pi, sn->ph[pi].kind, sn->ph[pi].type, sn->ph[pi].ntiles,
                   (hp[pi * 4] - z) * 0.01, (hp[pi * 4 + 1] - z) * 0.01, hp[pi * 4 + 2] * 0.01, hp[pi * 4 + 3] * 0.01);
      if (sn->ph[pi].kind == 0) {
        std::fprintf(stderr, " | arrive %5.2f weights %5.2f items %5.2f wait %5.2f", pi > 0 ? (hp[256 + pi * 8 + 6] - hp[(pi - 1) * 4 + 1]) * 0.01 : 0.0,
                     (hp[256 + pi * 8 + 7] - hp[256 + pi * 8 + 6]) * 0.01, (hp[256 + pi * 8 + 5] - hp[256 + pi * 8 + 7]) * 0.01,
                     (hp[pi * 4] - hp[256 + pi * 8 + 5]) * 0.01);
        std::fprintf(stderr, " | loads issued %5.2f tables %5.2f patch %5.2f K loop %5.2f sync %5.2f epilogue %5.2f", (hp[256 + pi * 8] - hp[pi * 4]) * 0.01,
                     (hp[256 + pi * 8 + 1] - hp[256 + pi * 8]) * 0.01, (hp[256 + pi * 8 + 2] - hp[256 + pi * 8 + 1]) * 0.01,
                     (hp[256 + pi * 8 + 3] - hp[256 + pi * 8 + 2]) * 0.01, (hp[256 + pi * 8 + 4] - hp[256 + pi * 8 + 3]) * 0.01,
                     (hp[pi * 4 + 1] - hp[256 + pi * 8 + 4]) * 0.01);
      }
      std::fprintf(stderr, "\n");
    }
  }
  return PNVO_OK;
}

