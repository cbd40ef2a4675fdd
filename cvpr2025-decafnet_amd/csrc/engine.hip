// Runtime of the grounding forward: owns the parameter table (bound by reference state_dict
// name), repacked convolution weights, the HBM workspace arena and the launch sequence of
// PtTransformerEarlyFusionIterative._drop_forward_eval (libs/modeling/model.py:480-565).
//
// HBM layout of one batched forward (B queries of one video, T0 padded clips, S = sum_l T0/2^l):
//   P1, P2      [T0][E]          query-independent halves of vid_map (W[:, :D].vid, W[:, D:].shallow)
//   X, R0..R6   [B*T0][E]        token-major activations (level l uses the first B*T_l rows)
//   H2          [B*T0][2E]       xattn projection (AdaLN scale | shift)
//   HID         [B*T0][4E]       FFN hidden
//   F           [B*S][E+32]      feature pyramid, rows ordered [level][query][t]; the last 32
//                                columns receive the refined logits (model.py:462-467)
//   HA, HB      [B*S][E+32]      head trunk ping-pong
//   mask_all, nbr_all [B*S]      per-row validity / k3-neighbour flags for every level
// Nothing here synchronises the host: vid_len, the gate and every mask stay on the device.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <string>
#include <atomic>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "../../include/decafnet_hip.h"
#include "attn.h"
#include "common.h"
#include "dec_chain.h"
#include "enc_chain.h"
#include "ffn_chain.h"
#include "gemm.h"
#include "head_chain.h"
#include "heads.h"
#include "postproc.h"
#include "rowops.h"
#include "score.h"

namespace dcf {

static thread_local std::string g_err;
void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
}

// ---- developer / test options (dcf_debug_set_option): named integers that override a built-in threshold, e.g. the row count from
// which a chain kernel replaces its launches, so that the operator tests can send small fixtures through the large-grid kernels
static std::unordered_map<std::string, int>& debug_options() {
  static std::unordered_map<std::string, int> o;
  return o;
}
static std::mutex& debug_options_mutex() {
  static std::mutex mu;
  return mu;
}
// bumped by every dcf_debug_set_option: a model whose captured graphs were recorded under another epoch drops them (the options
// choose kernels, a replay would keep running the old choice)
static std::atomic<int> g_option_epoch{0};
static int debug_option(const char* name, int dflt) {
  std::lock_guard<std::mutex> lock(debug_options_mutex());
  auto& o = debug_options();
  auto it = o.find(name);
  return it == o.end() ? dflt : it->second;
}

// ---- per-launch profiler (dcf_profile_*) --------------------------------------------------
struct ProfRec { std::string name; hipEvent_t a, b; double flops, bytes; };
static bool g_prof_on = false;
static std::vector<ProfRec> g_recs;

ProfScope::ProfScope(const char* name, hipStream_t s, double flops, double bytes) : idx(-1), st(s) {
  if (!g_prof_on) return;
  ProfRec r{name, nullptr, nullptr, flops, bytes};
  if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
  (void)hipEventRecord(r.a, st);
  g_recs.push_back(r);
  idx = (int)g_recs.size() - 1;
}
ProfScope::~ProfScope() {
  if (idx >= 0) (void)hipEventRecord(g_recs[idx].b, st);
}

// ---- tiny utility kernels ------------------------------------------------------------------
// dst[perm(i0,i1,i2)] = src[i0][i1][i2];  p0..p2 give the destination axis order
__global__ void k_permute3(const float* __restrict__ src, float* __restrict__ dst, int d0, int d1, int d2, int p0, int p1,
                           int p2) {
  const int n = d0 * d1 * d2;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int idx[3];
  idx[0] = i / (d1 * d2);
  idx[1] = (i / d2) % d1;
  idx[2] = i % d2;
  const int dims[3] = {d0, d1, d2};
  const int perm[3] = {p0, p1, p2};
  const int o = (idx[perm[0]] * dims[perm[1]] + idx[perm[1]]) * dims[perm[2]] + idx[perm[2]];
  dst[o] = src[i];
}

// masks_out[b][off_l + t] = mask_all[start_l + b*T_l + t].  Last kernel of a forward: when the sticky numerics word of the
// f16x3 GEMMs is raised -- bit 0: an operand left the fp16 range somewhere upstream, and a ReLU / max may have swallowed the
// NaN since; bit 1: a LayerNorm carried as one-pass row statistics met a row whose mean dwarfs its spread (common.h
// LN_ILL_RATIO), its rstd is off by an unbounded amount -- the logits of the forward are overwritten with NaN, so that a
// caller who never asks dcf_numerics_status -- the reference's Evaluator -- sees invalid scores instead of plausible wrong ones.
__global__ void k_masks_out(const uint8_t* __restrict__ mask_all, uint8_t* __restrict__ out, const LevelTable* lt,
                            const unsigned* __restrict__ status, float* __restrict__ logits) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  const int total = lt->start[lt->n_levels];
  if (r >= total) return;
  int l = 0;
  while (l + 1 < lt->n_levels && r >= lt->start[l + 1]) ++l;
  const int rel = r - lt->start[l];
  const int b = rel / lt->T[l], t = rel - b * lt->T[l];
  const int64_t o = (int64_t)b * lt->S + lt->off[l] + t;
  out[o] = mask_all[r];
  if (status && (status[0] & 3u)) logits[o] = __uint_as_float(0x7fc00000u);
}

// out[b][off_l + t] = rows[start_l + b*T_l + t]: a per-point value of the pyramid from level-major to query-major order
__global__ void k_points_out(const float* __restrict__ rows, float* __restrict__ out, const LevelTable* lt) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  const int total = lt->start[lt->n_levels];
  if (r >= total) return;
  int l = 0;
  while (l + 1 < lt->n_levels && r >= lt->start[l + 1]) ++l;
  const int rel = r - lt->start[l];
  const int b = rel / lt->T[l], t = rel - b * lt->T[l];
  out[(int64_t)b * lt->S + lt->off[l] + t] = rows[r];
}

// gate[b][t] = override[q0 + b][t]; mask = vid_mask (msf) or vid_mask & gate (model.py:544-545)
__global__ void k_apply_gate(const float* __restrict__ gate_in, const uint8_t* __restrict__ vid_mask, float* __restrict__ gate,
                             uint8_t* __restrict__ mask_out, int T, int rows, int msf) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const int t = r % T;
  const float g = gate_in[r];
  const bool m = vid_mask[t] != 0;
  gate[r] = g;
  mask_out[r] = msf ? m : (m && g != 0.f);
}

// LayerNorm folded into the 1x1 convolution that consumes it (GemmArgs::stats_in): Wf[n][k] = W[n][k] g[k],
// s[n] = sum_k Wf[n][k], c[n] = bias[n] + sum_k beta[k] W[n][k].  One wave per output channel.
__global__ __launch_bounds__(64) void k_fold_ln(const float* __restrict__ W, const float* __restrict__ bias, const float* __restrict__ g,
                                                const float* __restrict__ beta, float* __restrict__ Wf, float* __restrict__ s,
                                                float* __restrict__ c, int K) {
  const int n = blockIdx.x, lane = threadIdx.x;
  float a1 = 0.f, a2 = 0.f;
  for (int k = lane; k < K; k += 64) {
    const float w = W[(int64_t)n * K + k];
    const float wf = w * g[k];
    Wf[(int64_t)n * K + k] = wf;
    a1 += wf;
    a2 += beta[k] * w;
  }
  a1 = wave_sum(a1);
  a2 = wave_sum(a2);
  if (lane == 0) { s[n] = a1; c[n] = (bias ? bias[n] : 0.f) + a2; }
}

struct Bound {
  const float* p = nullptr;
  std::vector<int64_t> shape;
  int64_t numel() const {
    int64_t n = 1;
    for (auto s : shape) n *= s;
    return n;
  }
};

struct EncW {   // one TransformerEncoder of vid_net
  const float *ln_attn_w, *ln_attn_b, *dw_q, *dw_k, *dw_v, *qn_w, *qn_b, *kn_w, *kn_b, *vn_w, *vn_b;
  const float *wq, *bq, *wk, *bk, *wv, *bv, *wp, *bp, *ls_attn;
  const float *ln_ffn_w, *ln_ffn_b, *fc_w, *fc_b, *pj_w, *pj_b, *ls_ffn;
  const float *fc_wf, *fc_s, *fc_c;          // ffn.fc with ln_ffn folded in (k_fold_ln); nullptr where not built
  // enc_chain.hip: query / key / value with q / k / v_norm folded in, as chain images + the fold's s[n], c[n]; nullptr where not built
  const unsigned short* qkv_chain[3];
  const float *qkv_s[3], *qkv_c[3];
  const unsigned short* wp_chain;            // chain image of attn.proj (enc_chain.hip k_enc_attn) or nullptr
};
struct DecW {   // one TransformerDecoder of the fusion
  const float *ln_q_w, *ln_q_b, *ln_kv_w, *ln_kv_b, *dw, *qn_w, *qn_b;
  const float *wq, *bq, *wk, *bk, *wv, *bv, *wp, *bp;
  const float *wp_il, *bp_il;                // xattn.proj with its output rows in blocks of (32 scale rows, 32 shift rows of the same channels)
  const float *ln_ffn_w, *ln_ffn_b, *fc_w, *fc_b, *pj_w, *pj_b, *ls_ffn;
  const float *fc_wf, *fc_s, *fc_c;          // ffn.fc with ln_ffn folded in
  const unsigned short *wq_chain, *wp_chain; // chain images of xattn.query / the interleaved xattn.proj (dec_chain.hip) or nullptr
};
struct TextEncW {   // one TransformerEncoder of text_net (stride 0: no depthwise convs, global attention)
  const float *ln_attn_w, *ln_attn_b, *wq, *bq, *wk, *bk, *wv, *bv, *wp, *bp, *ls_attn;
  const float *ln_ffn_w, *ln_ffn_b, *fc_w, *fc_b, *pj_w, *pj_b, *ls_ffn;
};
struct HeadW {
  std::vector<const float*> conv;            // packed [N][3][Cin]
  std::vector<const float*> ln_w, ln_b;
  const float* out_w;                        // packed [NO][3][Cin]
  const float* out_b;
  const unsigned short* chain[2] = {nullptr, nullptr};   // chain images of the two trunk convolutions (head_chain.hip) or nullptr
};

// opt.model.vid_net.stride (video_net.py:39): the embedding convolutions divide the sequence by it; 0 (older callers) reads as 1
static inline int vid_stride_of(const dcf_config& c) { return c.vid_stride > 1 ? c.vid_stride : 1; }

struct Plan {    // geometry for one (T0, B, levels)
  int T0 = 0, B = 0, L = 0;
  LevelTable lt{};
  LevelTable* d_lt = nullptr;
};

}  // namespace dcf

using namespace dcf;

namespace dcf { struct HybridState; }
using dcf::HybridState;

struct dcf_model {
  dcf_config cfg{};
  std::unordered_map<std::string, Bound> bound;
  std::vector<float*> owned;                 // packed weights
  std::unordered_map<const float*, const unsigned short*> wsplit;   // fp32 weight -> [3][N][K] bf16 planes
  std::unordered_map<const float*, int64_t> wsplit_ldw;             // row pitch of the fp32 weight the planes were made from
  std::unordered_map<const float*, int> wsplit_terms;               // mode the image of a weight was made for (16 / 6)
  int gemm_terms = 16;                       // 16: f16x3 split MFMA GEMM (default); 6: bf16x6; 0: native fp32 MFMA
  bool force_x6 = false;                     // a weight did not fit the scaled fp16 range: the model runs bf16x6
  bool no_ln_carry = false;                  // dcf_model_set_ln_carry(m, 0): every LayerNorm as its own two-pass launch
  int option_epoch = 0;                      // g_option_epoch the captured graphs were recorded under
  unsigned* status = nullptr;                // device words: [0] sticky numerics flag of the f16x3 GEMMs, [1] weight range flag
  bool finalized = false;
  const float* pe = nullptr;
  int64_t pe_T = 0;

  // resolved weights
  const float *vid_map_w = nullptr, *vid_map_b = nullptr;
  // column blocks of the (E, Din) vid_map weight: expert half, sidekick half, the scat column (model.py:543-551)
  const float *vid_w1 = nullptr, *vid_w2 = nullptr, *vid_w3 = nullptr;
  int64_t vid_ldw = 0;
  std::vector<DecW> dec;
  const float *fus_out_w = nullptr, *fus_out_b = nullptr;
  const float *embd_fc_w = nullptr, *embd_fc_b = nullptr;
  const float *embd_fc_wf = nullptr, *embd_fc_s = nullptr, *embd_fc_c = nullptr;   // vid_net.embd_fc with fusion.ln_out folded in
  std::vector<const float*> embd_conv, embd_ln_w, embd_ln_b;
  std::vector<EncW> stem, branch;
  std::vector<const float*> pool_w;          // vid_net.pool_only: depthwise k3 weight [3][E] of every branch layer (video_net.py:107-109)
  HeadW cls1, cls2, reg;
  std::vector<float> reg_scales;             // host copy of reg_head.scales.{l}.scale
  const float *tcn_in_w = nullptr, *tcn_in_b = nullptr, *tcn_out_w = nullptr, *tcn_out_b = nullptr;
  std::vector<const float*> tcn_wd, tcn_bd, tcn_wp, tcn_bp, tcn_lnw, tcn_lnb;
  std::vector<const unsigned short*> tcn_frag;   // f16x3: MFMA fragment image of every TCN layer (launch_tcn_frag_image)
  // text_net (TextTransformer, text_net.py:92-188); empty when cfg.text_layers == 0
  const float *text_embd_w = nullptr, *text_embd_b = nullptr, *text_bkgd = nullptr;
  TextEncW text_pool{};                      // TextIdentity: attn_pool.attn.{query,key,value,proj} (text_net.py:50-53)
  std::vector<TextEncW> text_enc;
  const float* text_pe = nullptr;            // (text_pe_L, TE) token-major, borrowed
  int64_t text_pe_L = 0;
  char* text_ws = nullptr;
  size_t text_ws_bytes = 0;

  // workspace
  char* arena = nullptr;
  size_t arena_bytes = 0;
  std::vector<Plan> plans;
  // HIP graph of the last repeated forward (same pointers and sizes): one graph launch replaces ~135 kernel launches,
  // so a busy host cannot starve the GPU.  Captured on the second identical call, dropped whenever anything it bakes
  // in changes (weights, position encoding, workspace).
  std::vector<uint64_t> last_key, graph_key;
  hipGraph_t graph = nullptr;
  hipGraphExec_t graph_exec = nullptr;
  bool capturing = false;
  int graph_mode = 0;                        // dcf_model_set_graph_mode: 0 auto (by size), 1 always, 2 never
  std::vector<uint64_t> nocapture_key;       // argument set whose capture failed: run it eagerly, do not retry every call
  int last_launch = 0;                       // how the last forward was issued: 0 eager, 1 graph replay, 2 graph capture + launch
  // The legacy default stream (NULL: what torch's default stream is) cannot be captured.  A forward called on it hops to
  // this engine-owned non-blocking stream, ordered after / before the caller's stream by two events, so that the
  // reference's calling pattern (one stream, one video per call) replays a graph too.
  hipStream_t own = nullptr;
  hipEvent_t ev_in = nullptr, ev_out = nullptr;
  // last-forward bookkeeping for dcf_debug_copy
  struct {
    float *correl = nullptr, *gate = nullptr, *vidmap = nullptr, *fused = nullptr, *F = nullptr;
    int nq = 0, T0 = 0, B = 0, S = 0;
  } dbg;
  struct HybridState* hyb = nullptr;         // one long video sharded at pyramid level k (dcf_hybrid_phase1 / 2 / 3)
  int hyb_levels = 0;                        // > 0 while phase 1 runs: the forward builds levels 0 .. hyb_levels - 1 and stops behind the encoder
  size_t hyb_extra = 0;                      // bytes of workspace behind the forward's own buffers (the coarse pyramid)
  char* hyb_extra_ptr = nullptr;
  float* hyb_feat_out = nullptr;
  float* dbg_vidmap = nullptr;
  float* dbg_fused = nullptr;
  int64_t dbg_cap = 0;                        // capacity (floats) of the armed tap destinations
  bool keep_debug = false;
};

namespace dcf {

static void free_hybrid(dcf_model* m);

static void drop_graph(dcf_model* m, bool keep_last_key = false) {
  if (m->graph_exec) (void)hipGraphExecDestroy(m->graph_exec);
  if (m->graph) (void)hipGraphDestroy(m->graph);
  m->graph_exec = nullptr;
  m->graph = nullptr;
  m->graph_key.clear();
  if (!keep_last_key) m->last_key.clear();
}

static int free_model(dcf_model* m) {
  drop_graph(m);
  if (m->status) (void)hipFree(m->status);
  m->status = nullptr;
  for (float* p : m->owned) (void)hipFree(p);
  m->owned.clear();
  m->wsplit.clear();
  m->wsplit_ldw.clear();
  m->wsplit_terms.clear();
  for (auto& pl : m->plans) if (pl.d_lt) (void)hipFree(pl.d_lt);
  m->plans.clear();
  free_hybrid(m);                             // (the level-cut state of dcf_hybrid_phase1 / 2 / 3)
  if (m->arena) (void)hipFree(m->arena);
  if (m->text_ws) (void)hipFree(m->text_ws);
  if (m->ev_in) (void)hipEventDestroy(m->ev_in);
  if (m->ev_out) (void)hipEventDestroy(m->ev_out);
  if (m->own) (void)hipStreamDestroy(m->own);
  return 0;
}

static int get(dcf_model* m, const std::string& name, std::initializer_list<int64_t> shape, const float** out) {
  auto it = m->bound.find(name);
  DCF_CHECK(it != m->bound.end(), "parameter '%s' is not bound", name.c_str());
  const Bound& b = it->second;
  int64_t want = 1;
  for (auto s : shape) want *= s;
  DCF_CHECK(b.numel() == want, "parameter '%s' has %lld elements, expected %lld", name.c_str(), (long long)b.numel(),
            (long long)want);
  *out = b.p;
  return 0;
}

// repack a 3-d tensor [d0][d1][d2] with destination axis order (p0,p1,p2); result owned by the model
static int pack3(dcf_model* m, const float* src, int d0, int d1, int d2, int p0, int p1, int p2, hipStream_t st,
                 const float** out) {
  float* dst = nullptr;
  const size_t n = (size_t)d0 * d1 * d2;
  DCF_HIP(hipMalloc(&dst, n * sizeof(float)));
  m->owned.push_back(dst);
  hipLaunchKernelGGL(k_permute3, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, dst, d0, d1, d2, p0, p1, p2);
  DCF_HIP(hipGetLastError());
  *out = dst;
  return 0;
}

// bf16 planes of a GEMM weight [N][K] (row pitch K); owned by the model
static int split_weight(dcf_model* m, const float* W, int N, int K, hipStream_t st, int64_t ldw = 0, int terms = 0) {
  if (m->gemm_terms == 0 || m->wsplit.count(W)) return 0;
  if (!terms) terms = m->gemm_terms;
  unsigned short* planes = nullptr;
  DCF_HIP(hipMalloc(&planes, (size_t)3 * N * K * sizeof(unsigned short)));
  m->owned.push_back(reinterpret_cast<float*>(planes));
  if (launch_split_planes(W, planes, N, K, ldw ? ldw : K, st, terms, m->status ? m->status + 1 : nullptr)) return -1;
  m->wsplit[W] = planes;
  m->wsplit_ldw[W] = ldw ? ldw : K;
  m->wsplit_terms[W] = terms;
  return 0;
}
#define SPLIT(W, N, K) do { if (split_weight(m, (W), (N), (K), st)) return -1; } while (0)

// the (N, K) weight W / bias of a 1x1 convolution behind LayerNorm(g, beta): folded copies owned by the model (+ weight image)
static int fold_ln(dcf_model* m, const float* W, const float* bias, const float* g, const float* beta, int N, int K, hipStream_t st,
                   const float** wf, const float** s_out, const float** c_out) {
  float* buf = nullptr;
  DCF_HIP(hipMalloc(&buf, ((size_t)N * K + 2 * (size_t)N) * sizeof(float)));
  m->owned.push_back(buf);
  float* sv = buf + (size_t)N * K;
  hipLaunchKernelGGL(k_fold_ln, dim3(N), dim3(64), 0, st, W, bias, g, beta, buf, sv, sv + N, K);
  DCF_HIP(hipGetLastError());
  *wf = buf; *s_out = sv; *c_out = sv + N;
  return split_weight(m, buf, N, K, st);
}

#define GET(name, shape, dst) do { if (get(m, (name), shape, &(dst))) return -1; } while (0)
#define SH(...) std::initializer_list<int64_t>{__VA_ARGS__}

static int resolve_encoder(dcf_model* m, const std::string& p, int E, hipStream_t st, EncW& w) {
  const float* t;
  GET(p + ".ln_attn.weight", SH(E), w.ln_attn_w); GET(p + ".ln_attn.bias", SH(E), w.ln_attn_b);
  GET(p + ".attn.q_conv.conv.weight", SH(E, 3), t); if (pack3(m, t, 1, E, 3, 0, 2, 1, st, &w.dw_q)) return -1;
  GET(p + ".attn.k_conv.conv.weight", SH(E, 3), t); if (pack3(m, t, 1, E, 3, 0, 2, 1, st, &w.dw_k)) return -1;
  GET(p + ".attn.v_conv.conv.weight", SH(E, 3), t); if (pack3(m, t, 1, E, 3, 0, 2, 1, st, &w.dw_v)) return -1;
  GET(p + ".attn.q_norm.weight", SH(E), w.qn_w); GET(p + ".attn.q_norm.bias", SH(E), w.qn_b);
  GET(p + ".attn.k_norm.weight", SH(E), w.kn_w); GET(p + ".attn.k_norm.bias", SH(E), w.kn_b);
  GET(p + ".attn.v_norm.weight", SH(E), w.vn_w); GET(p + ".attn.v_norm.bias", SH(E), w.vn_b);
  GET(p + ".attn.attn.query.weight", SH(E, E), w.wq); GET(p + ".attn.attn.query.bias", SH(E), w.bq);
  GET(p + ".attn.attn.key.weight", SH(E, E), w.wk); GET(p + ".attn.attn.key.bias", SH(E), w.bk);
  GET(p + ".attn.attn.value.weight", SH(E, E), w.wv); GET(p + ".attn.attn.value.bias", SH(E), w.bv);
  GET(p + ".attn.attn.proj.weight", SH(E, E), w.wp); GET(p + ".attn.attn.proj.bias", SH(E), w.bp);
  GET(p + ".drop_path_attn.scale", SH(E), w.ls_attn);
  GET(p + ".ln_ffn.weight", SH(E), w.ln_ffn_w); GET(p + ".ln_ffn.bias", SH(E), w.ln_ffn_b);
  GET(p + ".ffn.fc.weight", SH(4 * E, E), w.fc_w); GET(p + ".ffn.fc.bias", SH(4 * E), w.fc_b);
  GET(p + ".ffn.proj.weight", SH(E, 4 * E), w.pj_w); GET(p + ".ffn.proj.bias", SH(E), w.pj_b);
  GET(p + ".drop_path_ffn.scale", SH(E), w.ls_ffn);
  SPLIT(w.wq, E, E); SPLIT(w.wk, E, E); SPLIT(w.wv, E, E); SPLIT(w.wp, E, E);
  SPLIT(w.fc_w, 4 * E, E); SPLIT(w.pj_w, E, 4 * E);
  w.fc_wf = w.fc_s = w.fc_c = nullptr;
  if (m->gemm_terms != 0 && E % 64 == 0 && fold_ln(m, w.fc_w, w.fc_b, w.ln_ffn_w, w.ln_ffn_b, 4 * E, E, st, &w.fc_wf, &w.fc_s, &w.fc_c)) return -1;
  for (int i = 0; i < 3; ++i) { w.qkv_chain[i] = nullptr; w.qkv_s[i] = w.qkv_c[i] = nullptr; }
  w.wp_chain = nullptr;
  if (m->gemm_terms == GEMM_F16X3 && enc_chain_supports(E, m->cfg.vid_heads, m->cfg.win > 0 ? m->cfg.win : 99)) {
    {
      unsigned short* img = nullptr;
      DCF_HIP(hipMalloc(&img, chain1_image_halfs(E, E) * sizeof(unsigned short)));
      m->owned.push_back(reinterpret_cast<float*>(img));
      if (launch_split_chain1(w.wp, img, E, E, st, nullptr)) return -1;          // (range: the same weights passed split_weight above)
      w.wp_chain = img;
    }
    const float* W3[3] = {w.wq, w.wk, w.wv};
    const float* B3[3] = {w.bq, w.bk, w.bv};
    const float* G3[3] = {w.qn_w, w.kn_w, w.vn_w};
    const float* H3[3] = {w.qn_b, w.kn_b, w.vn_b};
    for (int i = 0; i < 3; ++i) {
      const float* wf;
      if (fold_ln(m, W3[i], B3[i], G3[i], H3[i], E, E, st, &wf, &w.qkv_s[i], &w.qkv_c[i])) return -1;
      unsigned short* img = nullptr;
      DCF_HIP(hipMalloc(&img, chain1_image_halfs(E, E) * sizeof(unsigned short)));
      m->owned.push_back(reinterpret_cast<float*>(img));
      if (launch_split_chain1(wf, img, E, E, st, m->status ? m->status + 1 : nullptr)) return -1;     // (the gain widens the weight's range)
      w.qkv_chain[i] = img;
    }
  }
  return 0;
}

// TransformerDecoder parameters (blocks.py:594-630) under prefix p
static int resolve_decoder(dcf_model* m, const std::string& p, int E, int TE, hipStream_t st, DecW& w) {
  const float* t;
  GET(p + ".ln_xattn_q.weight", SH(E), w.ln_q_w); GET(p + ".ln_xattn_q.bias", SH(E), w.ln_q_b);
  GET(p + ".ln_xattn_kv.weight", SH(TE), w.ln_kv_w); GET(p + ".ln_xattn_kv.bias", SH(TE), w.ln_kv_b);
  GET(p + ".xattn.q_conv.conv.weight", SH(E, 3), t); if (pack3(m, t, 1, E, 3, 0, 2, 1, st, &w.dw)) return -1;
  GET(p + ".xattn.q_norm.weight", SH(E), w.qn_w); GET(p + ".xattn.q_norm.bias", SH(E), w.qn_b);
  GET(p + ".xattn.xattn.query.weight", SH(E, E), w.wq); GET(p + ".xattn.xattn.query.bias", SH(E), w.bq);
  GET(p + ".xattn.xattn.key.weight", SH(E, TE), w.wk); GET(p + ".xattn.xattn.key.bias", SH(E), w.bk);
  GET(p + ".xattn.xattn.value.weight", SH(E, TE), w.wv); GET(p + ".xattn.xattn.value.bias", SH(E), w.bv);
  GET(p + ".xattn.xattn.proj.weight", SH(2 * E, E), w.wp); GET(p + ".xattn.xattn.proj.bias", SH(2 * E), w.bp);
  GET(p + ".ln_ffn.weight", SH(E), w.ln_ffn_w); GET(p + ".ln_ffn.bias", SH(E), w.ln_ffn_b);
  GET(p + ".ffn.fc.weight", SH(4 * E, E), w.fc_w); GET(p + ".ffn.fc.bias", SH(4 * E), w.fc_b);
  GET(p + ".ffn.proj.weight", SH(E, 4 * E), w.pj_w); GET(p + ".ffn.proj.bias", SH(E), w.pj_b);
  GET(p + ".drop_path_ffn.scale", SH(E), w.ls_ffn);
  SPLIT(w.wq, E, E); SPLIT(w.wk, E, TE); SPLIT(w.wv, E, TE); SPLIT(w.wp, 2 * E, E);
  SPLIT(w.fc_w, 4 * E, E); SPLIT(w.pj_w, E, 4 * E);
  w.fc_wf = w.fc_s = w.fc_c = nullptr;
  if (m->gemm_terms != 0 && E % 128 == 0 && fold_ln(m, w.fc_w, w.fc_b, w.ln_ffn_w, w.ln_ffn_b, 4 * E, E, st, &w.fc_wf, &w.fc_s, &w.fc_c)) return -1;
  // the same projection for the GEMM that applies the modulation in its epilogue (G_ADALN): rows (2, E / 32, 32) -> (E / 32, 2, 32)
  w.wp_il = w.bp_il = nullptr;
  if (E % 32 == 0) {
    if (pack3(m, w.wp, 2, E / 32, 32 * E, 1, 0, 2, st, &w.wp_il)) return -1;
    if (pack3(m, w.bp, 2, E / 32, 32, 1, 0, 2, st, &w.bp_il)) return -1;
    SPLIT(w.wp_il, 2 * E, E);
  }
  // the attention half of the layer as one kernel (dec_chain.hip): chain-order fragment images of the two projections
  w.wq_chain = w.wp_chain = nullptr;
  if (m->gemm_terms == GEMM_F16X3 && w.wp_il && dec_chain_supports(E, m->cfg.fusion_heads, 1)) {
    unsigned short *iq = nullptr, *ip = nullptr;
    DCF_HIP(hipMalloc(&iq, chain1_image_halfs(E, E) * sizeof(unsigned short)));
    m->owned.push_back(reinterpret_cast<float*>(iq));
    DCF_HIP(hipMalloc(&ip, chain1_image_halfs(2 * E, E) * sizeof(unsigned short)));
    m->owned.push_back(reinterpret_cast<float*>(ip));
    if (launch_split_chain1(w.wq, iq, E, E, st, nullptr)) return -1;          // (range: the same weights passed split_weight above)
    if (launch_split_chain1(w.wp_il, ip, 2 * E, E, st, nullptr)) return -1;
    w.wq_chain = iq; w.wp_chain = ip;
  }
  return 0;
}

// TCN parameters (tcn.py:40-64) under prefix p: in (32, n_in, 1) -> [n_in][32]; dilated (32,32,3) -> [3][ci][co];
// 1x1 (32,32,1) -> [ci][co]
static int resolve_tcn(dcf_model* m, const std::string& pre, int n_in, int n_layers, hipStream_t st) {
  const float* t;
  m->tcn_wd.clear(); m->tcn_bd.clear(); m->tcn_wp.clear(); m->tcn_bp.clear(); m->tcn_lnw.clear(); m->tcn_lnb.clear();
  GET(pre + ".conv_1x1.weight", SH(TCN_HID, n_in), t); if (pack3(m, t, 1, TCN_HID, n_in, 0, 2, 1, st, &m->tcn_in_w)) return -1;
  GET(pre + ".conv_1x1.bias", SH(TCN_HID), m->tcn_in_b);
  for (int i = 0; i < n_layers; ++i) {
    const std::string p = pre + ".layers." + std::to_string(i);
    const float* pk;
    GET(p + ".conv_dilated.weight", SH(TCN_HID, TCN_HID, 3), t);
    if (pack3(m, t, TCN_HID, TCN_HID, 3, 2, 1, 0, st, &pk)) return -1;
    m->tcn_wd.push_back(pk);
    GET(p + ".conv_dilated.bias", SH(TCN_HID), t); m->tcn_bd.push_back(t);
    GET(p + ".conv_1x1.weight", SH(TCN_HID, TCN_HID), t);
    if (pack3(m, t, 1, TCN_HID, TCN_HID, 0, 2, 1, st, &pk)) return -1;
    m->tcn_wp.push_back(pk);
    if (m->gemm_terms == GEMM_F16X3) {                   // the layers run in f16x3 too: same weight range, same fallback
      if (launch_f16_weight_range(m->tcn_wd.back(), 3 * TCN_HID * TCN_HID, m->status ? m->status + 1 : nullptr, st)) return -1;
      if (launch_f16_weight_range(m->tcn_wp.back(), TCN_HID * TCN_HID, m->status ? m->status + 1 : nullptr, st)) return -1;
    }
    GET(p + ".conv_1x1.bias", SH(TCN_HID), t); m->tcn_bp.push_back(t);
    GET(p + ".norm.weight", SH(TCN_HID), t); m->tcn_lnw.push_back(t);
    GET(p + ".norm.bias", SH(TCN_HID), t); m->tcn_lnb.push_back(t);
  }
  GET(pre + ".conv_out.weight", SH(TCN_HID, TCN_HID), t);
  if (pack3(m, t, 1, TCN_HID, TCN_HID, 0, 2, 1, st, &m->tcn_out_w)) return -1;
  if (m->gemm_terms == GEMM_F16X3 && launch_f16_weight_range(m->tcn_out_w, TCN_HID * TCN_HID, m->status ? m->status + 1 : nullptr, st)) return -1;
  GET(pre + ".conv_out.bias", SH(TCN_HID), m->tcn_out_b);
  m->tcn_frag.clear();
  if (m->gemm_terms == GEMM_F16X3) {                     // the layers' weight fragments once per model, not once per workgroup
    for (int i = 0; i < n_layers; ++i) {
      unsigned short* img = nullptr;
      DCF_HIP(hipMalloc(&img, (size_t)TCN_FRAG_HALFS * sizeof(unsigned short)));
      m->owned.push_back(reinterpret_cast<float*>(img));
      if (launch_tcn_frag_image(m->tcn_wd[i], m->tcn_wp[i], i + 1 == n_layers ? m->tcn_out_w : nullptr, img, st)) return -1;
      m->tcn_frag.push_back(img);
    }
  }
  return 0;
}

// dense-conv arithmetic of the model (dcf_config.gemm_mode) and its status words
static int init_gemm_mode(dcf_model* m, hipStream_t st) {
  const int gm = m->cfg.gemm_mode;
  DCF_CHECK(gm == 0 || gm == 1 || gm == 6 || gm == 16, "gemm_mode %d: use 0 / 16 (f16x3), 6 (bf16x6) or 1 (fp32); the bf16x3 mode was replaced by f16x3", gm);
  m->gemm_terms = gm == 1 ? 0 : (gm == 6 ? GEMM_BF16X6 : GEMM_F16X3);
  if (m->force_x6 && m->gemm_terms == GEMM_F16X3) m->gemm_terms = GEMM_BF16X6;
  if (!m->status) DCF_HIP(hipMalloc(&m->status, 2 * sizeof(unsigned)));
  DCF_HIP(hipMemsetAsync(m->status, 0, 2 * sizeof(unsigned), st));
  return 0;
}

static int resolve_head(dcf_model* m, const std::string& p, const std::string& out_name, int C, int NO, int layers,
                        hipStream_t st, HeadW& h) {
  const float* t;
  for (int i = 0; i < layers; ++i) {
    const std::string s = std::to_string(i);
    GET(p + ".convs." + s + ".conv.weight", SH(C, C, 3), t);
    const float* pk;
    if (pack3(m, t, C, C, 3, 0, 2, 1, st, &pk)) return -1;      // (N, Cin, 3) -> [N][3][Cin]
    SPLIT(pk, C, 3 * C);
    h.conv.push_back(pk);
    const float *lw, *lb;
    GET(p + ".norms." + s + ".weight", SH(C), lw); GET(p + ".norms." + s + ".bias", SH(C), lb);
    h.ln_w.push_back(lw); h.ln_b.push_back(lb);
  }
  GET(p + "." + out_name + ".conv.weight", SH(NO, C, 3), t);
  if (pack3(m, t, NO, C, 3, 0, 2, 1, st, &h.out_w)) return -1;
  GET(p + "." + out_name + ".conv.bias", SH(NO), h.out_b);
  h.chain[0] = h.chain[1] = nullptr;
  if (m->gemm_terms == GEMM_F16X3 && layers == 2 && head_chain_supports(C, NO)) {      // the whole head as one kernel (head_chain.hip)
    for (int i = 0; i < 2; ++i) {
      unsigned short* img = nullptr;
      DCF_HIP(hipMalloc(&img, head_chain_image_halfs(C) * sizeof(unsigned short)));
      m->owned.push_back(reinterpret_cast<float*>(img));
      if (launch_split_chain3(h.conv[i], img, C, st, nullptr)) return -1;   // (range: the same weights passed split_weight above)
      h.chain[i] = img;
    }
  }
  return 0;
}

// model.py:411-414 / :543-551: what vid_map (PtTransformer: vid_net.embd_fc) sees.  sfonly only exists in the iterative
// model and only on the msf branch (`elif`, model.py:546); its input is then the sidekick features alone.
static inline bool vidmap_sfonly(const dcf_config& c) { return c.model_kind == 0 && c.msf && c.sfonly; }
static inline int vidmap_in_dim(const dcf_config& c) {
  return ((c.msf && !vidmap_sfonly(c)) ? 2 * c.D : c.D) + (c.scat ? 1 : 0);
}

static int finalize(dcf_model* m, hipStream_t st) {
  const dcf_config& c = m->cfg;
  const int E = c.E, D = c.D, TE = c.TE, L = c.n_levels;
  const bool sfonly = vidmap_sfonly(c);
  const int Din = vidmap_in_dim(c);
  for (float* p : m->owned) (void)hipFree(p);
  m->owned.clear();
  m->wsplit.clear();
  m->wsplit_ldw.clear();
  m->wsplit_terms.clear();
  if (init_gemm_mode(m, st)) return -1;
  m->dec.clear(); m->stem.clear(); m->branch.clear();
  m->embd_conv.clear(); m->embd_ln_w.clear(); m->embd_ln_b.clear();
  m->cls1 = HeadW(); m->cls2 = HeadW(); m->reg = HeadW();
  m->tcn_wd.clear(); m->tcn_bd.clear(); m->tcn_wp.clear(); m->tcn_bp.clear(); m->tcn_lnw.clear(); m->tcn_lnb.clear();
  const float* t;

  m->text_enc.clear();
  m->text_embd_w = m->text_embd_b = m->text_bkgd = nullptr;
  m->text_pool = TextEncW();
  if (c.text_kind == 1) {
    // TextIdentity (text_net.py:22-89): optional 1x1 embedding, optional AttNPool1D token (use_bkgd_token)
    DCF_CHECK(c.text_in > 0 && c.text_heads >= 1 && TE % c.text_heads == 0, "text_net (identity): in_dim=%d heads=%d do not fit TE=%d", c.text_in, c.text_heads, TE);
    if (m->bound.count("text_net.embd_fc.conv.weight")) {
      GET("text_net.embd_fc.conv.weight", SH(TE, c.text_in), m->text_embd_w); GET("text_net.embd_fc.conv.bias", SH(TE), m->text_embd_b);
    } else {
      DCF_CHECK(c.text_in == TE, "text_net (identity) without embd_fc needs in_dim == embd_dim (%d vs %d)", c.text_in, TE);
    }
    if (c.text_bkgd) {
      TextEncW& w = m->text_pool;
      const std::string p = "text_net.attn_pool.attn";
      GET(p + ".query.weight", SH(TE, TE), w.wq); GET(p + ".query.bias", SH(TE), w.bq);
      GET(p + ".key.weight", SH(TE, TE), w.wk); GET(p + ".key.bias", SH(TE), w.bk);
      GET(p + ".value.weight", SH(TE, TE), w.wv); GET(p + ".value.bias", SH(TE), w.bv);
      GET(p + ".proj.weight", SH(TE, TE), w.wp); GET(p + ".proj.bias", SH(TE), w.bp);
      SPLIT(w.wq, TE, TE); SPLIT(w.wk, TE, TE); SPLIT(w.wv, TE, TE); SPLIT(w.wp, TE, TE);
    }
  } else
  if (c.text_layers > 0 || c.text_in > 0) {
    DCF_CHECK(c.text_in > 0 && c.text_layers >= 0 && c.text_heads >= 1 && TE % c.text_heads == 0,
              "text_net: in_dim=%d layers=%d heads=%d do not fit TE=%d", c.text_in, c.text_layers, c.text_heads, TE);
    GET("text_net.embd_fc.conv.weight", SH(TE, c.text_in), m->text_embd_w); GET("text_net.embd_fc.conv.bias", SH(TE), m->text_embd_b);
    if (c.text_bkgd) GET("text_net.bkgd_token", SH(TE), m->text_bkgd);
    for (int i = 0; i < c.text_layers; ++i) {
      const std::string p = "text_net.transformer." + std::to_string(i);
      TextEncW w{};
      GET(p + ".ln_attn.weight", SH(TE), w.ln_attn_w); GET(p + ".ln_attn.bias", SH(TE), w.ln_attn_b);
      GET(p + ".attn.attn.query.weight", SH(TE, TE), w.wq); GET(p + ".attn.attn.query.bias", SH(TE), w.bq);
      GET(p + ".attn.attn.key.weight", SH(TE, TE), w.wk); GET(p + ".attn.attn.key.bias", SH(TE), w.bk);
      GET(p + ".attn.attn.value.weight", SH(TE, TE), w.wv); GET(p + ".attn.attn.value.bias", SH(TE), w.bv);
      GET(p + ".attn.attn.proj.weight", SH(TE, TE), w.wp); GET(p + ".attn.attn.proj.bias", SH(TE), w.bp);
      GET(p + ".drop_path_attn.scale", SH(TE), w.ls_attn);
      GET(p + ".ln_ffn.weight", SH(TE), w.ln_ffn_w); GET(p + ".ln_ffn.bias", SH(TE), w.ln_ffn_b);
      GET(p + ".ffn.fc.weight", SH(4 * TE, TE), w.fc_w); GET(p + ".ffn.fc.bias", SH(4 * TE), w.fc_b);
      GET(p + ".ffn.proj.weight", SH(TE, 4 * TE), w.pj_w); GET(p + ".ffn.proj.bias", SH(TE), w.pj_b);
      GET(p + ".drop_path_ffn.scale", SH(TE), w.ls_ffn);
      SPLIT(w.wq, TE, TE); SPLIT(w.wk, TE, TE); SPLIT(w.wv, TE, TE); SPLIT(w.wp, TE, TE);
      SPLIT(w.fc_w, 4 * TE, TE); SPLIT(w.pj_w, TE, 4 * TE);
      m->text_enc.push_back(w);
    }
  }

  if (c.model_kind == 1) {   // PtTransformer: vid_net.embd_fc takes the (2)D-wide gated input itself (model.py:43-48)
    GET("vid_net.embd_fc.conv.weight", SH(E, Din), m->vid_map_w); GET("vid_net.embd_fc.conv.bias", SH(E), m->vid_map_b);
  } else {
    GET("vid_map.conv.weight", SH(E, Din), m->vid_map_w); GET("vid_map.conv.bias", SH(E), m->vid_map_b);
  }
  // the deep / shallow column halves of the (E, [2]D[+1]) weight are separate GEMM operands with row pitch Din
  m->vid_w1 = (c.msf && sfonly) ? nullptr : m->vid_map_w;
  m->vid_w2 = c.msf ? (sfonly ? m->vid_map_w : m->vid_map_w + D) : nullptr;
  m->vid_w3 = nullptr;
  m->vid_ldw = Din;
  if (c.scat) {
    // the extra score column makes the row pitch odd: keep aligned copies of the column blocks (pitch D) and of the column
    float* blk[3] = {nullptr, nullptr, nullptr};
    const float* src[3] = {m->vid_w1, m->vid_w2, m->vid_map_w + (Din - 1)};
    const int wid[3] = {D, D, 1};
    for (int i = 0; i < 3; ++i) {
      if (!src[i]) continue;
      DCF_HIP(hipMalloc(&blk[i], (size_t)E * wid[i] * sizeof(float)));
      m->owned.push_back(blk[i]);
      DCF_HIP(hipMemcpy2DAsync(blk[i], (size_t)wid[i] * 4, src[i], (size_t)Din * 4, (size_t)wid[i] * 4, E, hipMemcpyDeviceToDevice, st));
    }
    m->vid_w1 = blk[0]; m->vid_w2 = blk[1]; m->vid_w3 = blk[2];
    m->vid_ldw = D;
  }
  // these two GEMMs read the raw feature files, whose range the model does not control; everything downstream is
  // bounded by LayerNorms.  In f16x3 mode they run without the activation pre-scale (|x| < 65504 instead of 4094; an
  // absolute representation floor of 2^-25 on the features).
  if (D % 32 == 0 && E % 32 == 0) {
    if (m->vid_w1 && split_weight(m, m->vid_w1, E, D, st, m->vid_ldw)) return -1;
    if (m->vid_w2 && split_weight(m, m->vid_w2, E, D, st, m->vid_ldw)) return -1;
  }
  for (int i = 0; i < c.fusion_layers; ++i) {
    DecW w{};
    if (resolve_decoder(m, "fusion.layers." + std::to_string(i), E, TE, st, w)) return -1;
    m->dec.push_back(w);
  }
  GET("fusion.ln_out.weight", SH(E), m->fus_out_w); GET("fusion.ln_out.bias", SH(E), m->fus_out_b);
  if (c.model_kind != 1) {
    GET("vid_net.embd_fc.conv.weight", SH(E, E), m->embd_fc_w); GET("vid_net.embd_fc.conv.bias", SH(E), m->embd_fc_b);
    SPLIT(m->embd_fc_w, E, E);
    m->embd_fc_wf = m->embd_fc_s = m->embd_fc_c = nullptr;
    if (m->gemm_terms != 0 && E % 64 == 0 && c.fusion_layers > 0 &&
        fold_ln(m, m->embd_fc_w, m->embd_fc_b, m->fus_out_w, m->fus_out_b, E, E, st, &m->embd_fc_wf, &m->embd_fc_s, &m->embd_fc_c)) return -1;
  }
  for (int i = 0, sv = vid_stride_of(c); i < c.n_embd_convs; ++i, sv = std::max(sv / 2, 1)) {
    const std::string s = std::to_string(i);
    const int taps = sv > 1 ? 5 : 3;                 // vid_net.stride > 1: k5 / stride 2 / padding 2 (video_net.py:62-70)
    GET("vid_net.embd_convs." + s + ".conv.weight", SH(E, E, taps), t);
    const float* pk;
    if (pack3(m, t, E, E, taps, 0, 2, 1, st, &pk)) return -1;
    SPLIT(pk, E, taps * E);
    m->embd_conv.push_back(pk);
    const float *lw, *lb;
    GET("vid_net.embd_norms." + s + ".weight", SH(E), lw); GET("vid_net.embd_norms." + s + ".bias", SH(E), lb);
    m->embd_ln_w.push_back(lw); m->embd_ln_b.push_back(lb);
  }
  for (int i = 0; i < c.n_stem; ++i) {
    EncW w{};
    if (resolve_encoder(m, "vid_net.stem." + std::to_string(i), E, st, w)) return -1;
    m->stem.push_back(w);
  }
  m->pool_w.clear();
  for (int i = 0; i < L; ++i) {
    if (c.pool_only) {
      const float* pk;
      GET("vid_net.branch." + std::to_string(i) + ".conv.weight", SH(E, 3), t);
      if (pack3(m, t, 1, E, 3, 0, 2, 1, st, &pk)) return -1;
      m->pool_w.push_back(pk);
      continue;
    }
    EncW w{};
    if (resolve_encoder(m, "vid_net.branch." + std::to_string(i), E, st, w)) return -1;
    m->branch.push_back(w);
  }
  if (resolve_head(m, "cls_head", "cls_head", E, 1, c.head_layers, st, m->cls1)) return -1;
  const int EH = c.model_kind == 0 ? E + TCN_HID : E;      // only the iterative model concatenates the refined logits (model.py:426-428)
  if (c.model_kind == 0 && resolve_head(m, "cls_head2", "cls_head", EH, 1, c.head_layers, st, m->cls2)) return -1;
  if (resolve_head(m, "reg_head", "reg_head", EH, 2, c.head_layers, st, m->reg)) return -1;
  m->reg_scales.assign(L, 1.f);
  for (int l = 0; l < L; ++l) {
    GET("reg_head.scales." + std::to_string(l) + ".scale", SH(1), t);
    DCF_HIP(hipMemcpyAsync(&m->reg_scales[l], t, sizeof(float), hipMemcpyDeviceToHost, st));
  }
  if (c.model_kind == 0 && resolve_tcn(m, "refine", L, L, st)) return -1;
  DCF_HIP(hipStreamSynchronize(st));
  for (auto& pl : m->plans) if (pl.d_lt) (void)hipFree(pl.d_lt);
  m->plans.clear();
  free_hybrid(m);                             // reg scales live in the level tables
  drop_graph(m);
  if (m->gemm_terms == GEMM_F16X3) {
    // did every weight fit the scaled fp16 range (|w| < 255.9)?  If not, rebuild the images for bf16x6.
    unsigned flags[2] = {0u, 0u};
    DCF_HIP(hipMemcpyAsync(flags, m->status, sizeof(flags), hipMemcpyDeviceToHost, st));
    DCF_HIP(hipStreamSynchronize(st));
    if (flags[1]) {
      m->force_x6 = true;
      return finalize(m, st);
    }
  }
  m->finalized = true;
  return 0;
}

// ---- workspace ------------------------------------------------------------------------------
struct Arena {
  char* base;
  size_t off = 0, cap;
  bool dry;
  template <typename T>
  T* take(size_t n) {
    off = (off + 255) & ~(size_t)255;
    T* p = dry ? nullptr : reinterpret_cast<T*>(base + off);
    off += n * sizeof(T);
    return p;
  }
};

struct Buffers {
  float *P1, *P2, *tn, *partial, *correl, *gate;
  uint8_t *mask_all, *nbr_all, *kvmask, *maskv;
  uint8_t* tile_flags;                        // [B][(T0 + 63) / 64] 64-clip row tiles a query's gate keeps (GateArgs::tile_flags)
  uint8_t *mask_pre, *nbr_pre;                // vid_net.stride > 1: masks / neighbour flags of the input-resolution levels T0, T0/2, .. T0/stride
  float* col5;                                // vid_net.stride > 1: [B*T0/2][5E] rows of a k5 / stride-2 embedding convolution
  float *X, *R[7], *H2, *HID, *F, *HA, *HB, *HC, *HD, *logits1, *tcnA, *tcnB, *kvn, *Kt, *Vt;
  unsigned short* kvimg;                      // [B] K / V^T fragment images of the projected text (dec_chain.hip)
  float* kmadd;                               // [B][64] additive key mask
  float* stats;                               // [rows][E / 64] (sum, sum of squares): row statistics carried between GEMMs
  float* hstats[2];                           // the same for the k3 trunks (heads over the whole pyramid, embedding convolutions)
};

static void carve(Arena& a, const dcf_config& c, int T0, int B, int nq, int S, int Lk, int nvid, Buffers& b) {
  const size_t rows0 = (size_t)B * T0, rowsAll = (size_t)B * S;
  const size_t rowsF = std::max(rows0, (c.model_kind == 1 || c.second_fusion) ? rowsAll : (size_t)0);   // rows the fusion stack sees
  const bool strided = vid_stride_of(c) > 1;
  const int E = c.E, EH = c.E + TCN_HID;
  b.P1 = a.take<float>((size_t)nvid * T0 * E);
  b.P2 = a.take<float>((size_t)nvid * T0 * E);
  b.maskv = a.take<uint8_t>(nvid > 1 ? (size_t)nvid * T0 : 0);     // the videos' masks side by side (several videos only)
  b.tn = a.take<float>((size_t)nq * c.D);
  b.partial = a.take<float>((size_t)SCORE_SLICES * (nq + nvid) * T0);
  b.correl = a.take<float>((size_t)nq * T0);
  b.gate = a.take<float>(rows0);
  b.tile_flags = a.take<uint8_t>((size_t)B * ((T0 + 63) / 64));
  b.mask_all = a.take<uint8_t>(rowsAll);
  b.nbr_all = a.take<uint8_t>(rowsAll);
  b.mask_pre = a.take<uint8_t>(strided ? 2 * rows0 : 0);
  b.nbr_pre = a.take<uint8_t>(strided ? 2 * rows0 : 0);
  b.col5 = a.take<float>(strided ? rows0 / 2 * 5 * E : 0);
  b.kvmask = a.take<uint8_t>((size_t)B * Lk);
  b.X = a.take<float>(rows0 * E);
  for (int i = 0; i < 7; ++i) b.R[i] = a.take<float>((i < 3 ? rowsF : rows0) * E);
  b.stats = a.take<float>(rowsF * (size_t)((E + 63) / 64) * 2);
  for (int i = 0; i < 2; ++i) b.hstats[i] = a.take<float>(rowsAll * (size_t)((EH + 63) / 64) * 2);
  b.H2 = a.take<float>(rowsF * 2 * E);
  b.HID = a.take<float>(rowsF * 4 * E);
  b.F = a.take<float>(rowsAll * EH);
  b.HA = a.take<float>(rowsAll * EH);
  b.HB = a.take<float>(rowsAll * EH);
  b.HC = a.take<float>(rowsAll * EH);      // second trunk pair: two heads of equal shape run in lockstep (run_head_pair)
  b.HD = a.take<float>(rowsAll * EH);
  b.logits1 = a.take<float>(rowsAll);
  b.tcnA = a.take<float>(rows0 * TCN_HID);
  b.tcnB = a.take<float>(rows0 * TCN_HID);
  b.kvn = a.take<float>((size_t)B * Lk * c.TE);
  b.Kt = a.take<float>((size_t)B * Lk * E);
  b.Vt = a.take<float>((size_t)B * Lk * E);
  b.kvimg = a.take<unsigned short>((size_t)B * kv_image_halfs(2));
  b.kmadd = a.take<float>((size_t)B * 64);
}

static int get_plan(dcf_model* m, int T0, int B, int L, hipStream_t st, Plan** out) {
  for (auto& p : m->plans)
    if (p.T0 == T0 && p.B == B && p.L == L) { *out = &p; return 0; }
  Plan p;
  p.T0 = T0; p.B = B; p.L = L;
  LevelTable& lt = p.lt;
  lt.n_levels = L; lt.B = B;
  int acc = 0;
  for (int l = 0; l < lt.n_levels; ++l) {
    lt.T[l] = T0 >> l;
    lt.off[l] = acc;
    lt.start[l] = B * acc;
    lt.scale[l] = m->reg_scales[l];
    acc += lt.T[l];
  }
  lt.S = acc;
  lt.start[lt.n_levels] = B * acc;
  DCF_HIP(hipMalloc(&p.d_lt, sizeof(LevelTable)));
  DCF_HIP(hipMemcpyAsync(p.d_lt, &p.lt, sizeof(LevelTable), hipMemcpyHostToDevice, st));
  DCF_HIP(hipStreamSynchronize(st));          // p.lt is copied below; keep the source alive until done
  m->plans.push_back(p);
  *out = &m->plans.back();
  return 0;
}

static GemmArgs gemm(const float* A, int64_t lda, const float* W, const float* bias, float* C, int64_t ldc, int M, int N,
                     int K) {
  GemmArgs g{};
  g.A = A; g.lda = lda; g.W = W; g.ldw = 0; g.bias = bias; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
  return g;
}

#define TRY(x) do { if ((x) != 0) return -1; } while (0)

// dense GEMM dispatch: bf16-split MFMA when the weight has split planes, fp32 MFMA otherwise
static int run_gemm(dcf_model* m, GemmArgs* g, int count, GemmAMode mode, hipStream_t st) {
  bool split = m->gemm_terms != 0;
  int terms = 0;
  for (int i = 0; i < count && split; ++i) {
    auto it = m->wsplit.find(g[i].W);
    if (it == m->wsplit.end() || (g[i].ldw ? g[i].ldw : g[i].K) != m->wsplit_ldw[g[i].W]) { split = false; break; }
    const int t = m->wsplit_terms[g[i].W];
    if (terms && t != terms) { split = false; break; }
    terms = t;
    g[i].Ws = it->second;
    g[i].status = m->status;
  }
  if (split && mode == A_CHANMAJOR && (g[0].N % 128 != 0 || g[0].M % 4 != 0)) split = false;
  for (int i = 0; i < count; ++i) DCF_CHECK(split || !(g[i].flags & G_ADALN), "internal: G_ADALN needs the split-operand GEMM");
  for (int i = 0; i < count; ++i) DCF_CHECK(split || !g[i].score_out, "internal: scores on the side need the split-operand GEMM");
  return split ? launch_gemm_split(g, count, mode, terms, st) : launch_gemm(g, count, mode, st);
}

// FFN (blocks.py:535-538): fc with the erf GELU in its epilogue, then proj (`go`: residual / LayerScale / mask epilogue)
constexpr int STATS_W = 64;
// stats != nullptr: X holds the RAW rows, fc_w / fc_b are the LayerNorm-folded weight and bias and ln_s its row sums; the row
// statistics come from the GEMM that produced X (GemmArgs::stats_in)
// E = 256 in the f16x3 mode, from FFN_CHAIN_MIN_ROWS rows on: fc, GELU and proj as ONE kernel whose hidden activations stay in
// registers (ffn_chain.hip; 4 KiB per row neither written nor read back).  Below that the 128-row tiles leave CUs idle and
// the GEMM pair on 64-row tiles is faster (16 384 rows = 128 tiles, half the chip: 78 - 86 us against 63 + 40 for the pair;
// 8 192 rows: the same 80 us against 31 + 29).
constexpr int FFN_CHAIN_MIN_ROWS = 16384;
static bool g_no_ffn_chain();
// (the row / width part of can_chain_ffn: what a producer needs to know to hand over row statistics)
static bool can_chain_ffn_rows(dcf_model* m, int rows, int E) {
  return !g_no_ffn_chain() && E == 256 && rows >= FFN_CHAIN_MIN_ROWS && m->gemm_terms == GEMM_F16X3;
}
static bool g_no_ffn_chain() {
  static const bool off = getenv("DCF_NO_FFN_CHAIN") != nullptr;     // developer switch: always the GEMM pair
  return off;
}
static bool can_chain_ffn(dcf_model* m, const float* fc_w, const GemmArgs& go, int rows, int E) {
  if (g_no_ffn_chain() || E != 256 || rows < FFN_CHAIN_MIN_ROWS || m->gemm_terms != GEMM_F16X3) return false;
  if (!m->wsplit.count(fc_w) || !m->wsplit.count(go.W) || m->wsplit_terms[fc_w] != GEMM_F16X3 || m->wsplit_terms[go.W] != GEMM_F16X3) return false;
  if (m->wsplit_ldw[fc_w] != E || m->wsplit_ldw[go.W] != 4 * E) return false;
  if (go.ln_w || !(go.flags & G_RES) || (go.flags & ~(G_RES | G_OUT_MASK)) || !go.R || !go.bias || go.a_scale > 0.f) return false;
  return true;
}
static int run_ffn_chain(dcf_model* m, const float* X, const float* fc_w, const float* fc_b, const GemmArgs& go, int rows, int E,
                         hipStream_t st, const float* stats, const float* ln_s) {
  FfnChainArgs a{};
  a.X = X; a.ldx = E; a.W1s = m->wsplit[fc_w]; a.b1 = fc_b; a.ln_s = ln_s; a.stats = stats; a.stats_slots = E / 64;
  a.W2s = m->wsplit[go.W]; a.b2 = go.bias; a.ls = go.ls; a.R = go.R; a.ldr = go.ldr;
  a.rowmask = (go.flags & G_OUT_MASK) ? go.rowmask : nullptr; a.C = go.C; a.ldc = go.ldc;
  a.stats_out = go.stats_out; a.stats_w = go.stats_w; a.status = m->status; a.M = rows;
  ProfScope prof("gemm_f16x3<ffn_chain>", st, 2.0 * rows * E * 4.0 * E * 2.0, (double)rows * E * 4.0 * 3.0);
  return launch_ffn_chain(a, st);
}

static int run_ffn(dcf_model* m, const float* X, const float* fc_w, const float* fc_b, GemmArgs go, float* HID, int rows, int E,
                   hipStream_t st, const float* stats = nullptr, const float* ln_s = nullptr) {
  if (can_chain_ffn(m, fc_w, go, rows, E)) return run_ffn_chain(m, X, fc_w, fc_b, go, rows, E, st, stats, ln_s);
  GemmArgs gf = gemm(X, E, fc_w, fc_b, HID, 4 * E, rows, 4 * E, E);
  gf.flags = G_GELU;
  if (stats) { gf.stats_in = stats; gf.ln_s = ln_s; gf.stats_slots = E / STATS_W; gf.stats_w = STATS_W; }
  TRY(run_gemm(m, &gf, 1, A_ROWS, st));
  return run_gemm(m, &go, 1, A_ROWS, st);
}

// can the LayerNorm between a producer GEMM (rows x n_prod, K = k_prod) and the ffn.fc that consumes it ride as row statistics?
static bool g_no_carry_env() {
  static const bool off = getenv("DCF_NO_LN_CARRY") != nullptr;      // developer switch: standalone LayerNorm launches instead
  return off;
}
// ... or the model was told so (dcf_model_set_ln_carry: a row's mean dwarfed its spread, common.h LN_ILL_RATIO)
#define g_no_carry() (g_no_carry_env() || m->no_ln_carry)
// (each GEMM is asked about with the arithmetic of ITS weight image: one of the two may have fallen back to bf16x6)
static bool can_carry_ln(dcf_model* m, const float* prod_w, const float* fc_wf, int rows, int n_prod, int k_prod, int E) {
  if (g_no_carry() || !fc_wf || !prod_w || m->gemm_terms == 0 || !m->wsplit.count(fc_wf) || !m->wsplit.count(prod_w)) return false;
  return gemm_can_carry_stats(rows, n_prod, k_prod, 1, m->wsplit_terms[prod_w]) &&
         gemm_can_carry_stats(rows, 4 * E, E, 1, m->wsplit_terms[fc_wf]);
}

// can this GEMM carry its LayerNorm in the epilogue?  (bf16-split path with planes for W, tile spanning the row)
// ... and is its K loop long enough to pay for the heavier epilogue (two workgroup-wide reductions on a 64-row tile)?  At
// K = 256 the fused kernel takes 117 us for 81920 rows against 55 + 39 us for the 128x256 kernel + the LayerNorm kernel (65
// against 28 + 20 us at 40960 rows); at K = 768 (embedding convolutions) 232 against ~280 us, at K = 1024 a tie (81920 rows,
// round 2: the unfused kernel was the 64x256 tile too).
static bool can_fuse_ln(dcf_model* m, const float* W, int M, int N, int K, GemmAMode mode) {
  // ... and only below 64 K rows: from there on the unfused GEMM runs on the 128x256 tile, which beats the 64x256 tile the
  // fused epilogue needs by more than the LayerNorm pass costs.  Measured at 8 videos per forward (profiles/r03_notes.md):
  // 131072x256x1024 fused 327 us against 228 + 49 us; the k3 convolutions 261120x256x768 366 against 285 + 53 us and
  // 131072x256x768 238 against 146 + 53 us (the 128x256 k3 tile reaches 350 - 360 TFLOP/s, 0.43 of the f16x3 peak).
  if (M >= 65536) return false;
  return m->gemm_terms != 0 && m->wsplit.count(W) && m->wsplit_ldw[W] == K && K >= 512 && gemm_can_fuse_ln(M, N, K, mode);
}

// conv -> LayerNorm -> ReLU -> conv (head / embedding trunks): can the LayerNorm + ReLU ride in the second convolution's A
// staging, fed by row statistics from the first one's epilogue?  (both on a tile kernel that has the two instantiations)
static bool can_norm_a(dcf_model* m, const float* W1, const float* W2, int M, int C, int* stats_w) {
  if (g_no_carry() || m->gemm_terms == 0 || !m->wsplit.count(W1) || !m->wsplit.count(W2)) return false;
  const int t1 = m->wsplit_terms[W1], t2 = m->wsplit_terms[W2];
  return t1 == t2 && gemm_can_norm_a(M, C, 3 * C, t1, stats_w) && !can_fuse_ln(m, W1, M, C, 3 * C, A_ROWS_TAP3);
}
static void norm_a(GemmArgs& g, const float* stats, int C, int stats_w, const float* ln_g, const float* ln_b) {
  g.a_stats = stats; g.a_stats_slots = C / stats_w; g.a_ln_g = ln_g; g.a_ln_b = ln_b;
}

// From 32 768 rows on, f16x3, E = 256, 4 heads, window <= 9, stride 1: ln_attn, the depthwise convolutions, q / k / v_norm and the three
// projections of an encoder layer as ONE kernel (enc_chain.hip k_enc_qkv) instead of k_enc_pre + the grouped GEMM.
static int enc_chain_min_rows() {
  const int o = debug_option("enc_chain_min_rows", -1);            // dcf_debug_set_option (tests), then the developer switch
  if (o >= 0) return o;
  static const int v = getenv("DCF_ENC_CHAIN_MIN_ROWS") ? atoi(getenv("DCF_ENC_CHAIN_MIN_ROWS")) : 32768;
  return v;
}
static bool can_chain_enc(dcf_model* m, const EncW& w, int rows, int stride, int64_t ldx) {
  static const bool off = getenv("DCF_NO_ENC_CHAIN") != nullptr;    // developer switch: the separate launches
  const dcf_config& c = m->cfg;
  // (m->no_ln_carry: the kernel folds q / k / v_norm with one-pass statistics of the convolution outputs)
  return !off && !m->no_ln_carry && m->gemm_terms == GEMM_F16X3 && w.qkv_chain[0] && w.qkv_chain[1] && w.qkv_chain[2] && stride == 1 &&
         enc_chain_supports(c.E, c.vid_heads, c.win > 0 ? c.win : 99) && rows >= enc_chain_min_rows() && ldx % 4 == 0;
}

// the sidekick scores as a by-product of the shallow vid_map GEMM (dcf_debug_set_option("fuse_scores", 0) = the scoring kernels)
static bool fuse_scores_on() {
  static const bool off = getenv("DCF_NO_FUSE_SCORES") != nullptr;
  return !off && debug_option("fuse_scores", 1) != 0;
}
static int enc_attn_min_rows() {
  const int o = debug_option("enc_attn_min_rows", -1);
  if (o >= 0) return o;
  static const int v = getenv("DCF_ENC_ATTN_MIN_ROWS") ? atoi(getenv("DCF_ENC_ATTN_MIN_ROWS")) : 32768;
  return v;
}
static bool can_chain_enc_attn(dcf_model* m, const EncW& w, int rows, int64_t ldr) {
  static const bool off = getenv("DCF_NO_ENC_ATTN") != nullptr;     // developer switch: k_local_attn + the projection GEMM
  const dcf_config& c = m->cfg;
  return !off && m->gemm_terms == GEMM_F16X3 && w.wp_chain && c.win > 0 && (c.win & 1) && enc_chain_supports(c.E, c.vid_heads, c.win) &&
         rows >= enc_attn_min_rows() && ldr % 4 == 0;
}

// TransformerEncoder (vid_net) at one level.  Xin: [B*T_in][ldx]; output rows [B*T_out] at Xout (ld ldo).
static int run_encoder(dcf_model* m, const EncW& w, Buffers& b, const float* Xin, int64_t ldx, const uint8_t* mask_in,
                       const uint8_t* mask_out, int B, int T_in, int stride, float* Xout, int64_t ldo, hipStream_t st) {
  const dcf_config& c = m->cfg;
  const int E = c.E, To = T_in / stride, rows = B * To;
  if (can_chain_enc(m, w, rows, stride, ldx)) {
    EncQkvArgs ea{};
    ea.X = Xin; ea.ldx = ldx; ea.mask_in = mask_in; ea.ln_w = w.ln_attn_w; ea.ln_b = w.ln_attn_b;
    ea.dw[0] = w.dw_q; ea.dw[1] = w.dw_k; ea.dw[2] = w.dw_v;
    for (int i = 0; i < 3; ++i) { ea.W[i] = w.qkv_chain[i]; ea.fs[i] = w.qkv_s[i]; ea.fc[i] = w.qkv_c[i]; ea.out[i] = b.R[4 + i]; }
    ea.B = B; ea.T_in = T_in; ea.status = m->status;
    ProfScope prof("gemm_f16x3<enc_qkv>", st, 2.0 * rows * E * 3.0 * E, (double)rows * E * 4.0 * 4.0);
    TRY(launch_enc_qkv(ea, st));
  } else {
  EncPreArgs ep{};
  ep.X = Xin; ep.ldx = ldx; ep.mask_in = mask_in; ep.ln_w = w.ln_attn_w; ep.ln_b = w.ln_attn_b;
  ep.dw_q = w.dw_q; ep.dw_k = w.dw_k; ep.dw_v = w.dw_v;
  ep.qn_w = w.qn_w; ep.qn_b = w.qn_b; ep.kn_w = w.kn_w; ep.kn_b = w.kn_b; ep.vn_w = w.vn_w; ep.vn_b = w.vn_b;
  ep.Qc = b.R[0]; ep.Kc = b.R[1]; ep.Vc = b.R[2]; ep.Skip = stride == 2 ? b.R[3] : nullptr;
  ep.B = B; ep.T_in = T_in; ep.C = E;
  TRY(launch_enc_pre(ep, stride, st));
  GemmArgs g3[3] = {gemm(b.R[0], E, w.wq, w.bq, b.R[4], E, rows, E, E), gemm(b.R[1], E, w.wk, w.bk, b.R[5], E, rows, E, E),
                    gemm(b.R[2], E, w.wv, w.bv, b.R[6], E, rows, E, E)};
  TRY(run_gemm(m, g3, 3, A_ROWS, st));
  }
  // the window attention, attn.proj and the residual as one kernel (enc_chain.hip k_enc_attn) where the FFN takes x' with row statistics
  if (can_chain_enc_attn(m, w, rows, stride == 2 ? (int64_t)E : ldx)) {
    const bool carry = w.fc_wf && m->wsplit.count(w.fc_wf) && !g_no_carry() &&
                       (can_chain_ffn_rows(m, rows, E) || gemm_can_carry_stats(rows, 4 * E, E, 1, m->wsplit_terms[w.fc_wf]));
    EncAttnArgs aa{};
    aa.Q = b.R[4]; aa.K = b.R[5]; aa.V = b.R[6]; aa.mask = mask_out; aa.Wp = w.wp_chain; aa.bp = w.bp; aa.ls = w.ls_attn;
    if (stride == 2) { aa.R = b.R[3]; aa.ldr = E; } else { aa.R = Xin; aa.ldr = ldx; }
    aa.Y = b.R[1]; aa.ldy = E; aa.stats_out = carry ? b.stats : nullptr; aa.stats_w = STATS_W; aa.B = B; aa.T = To; aa.win = c.win; aa.status = m->status; aa.attn_single = c.attn_mode == 1;
    {
      ProfScope prof("gemm_f16x3<enc_attn>", st, 2.0 * rows * E * E + 4.0 * rows * E * c.win, (double)rows * E * 4.0 * 5.0);
      TRY(launch_enc_attn(aa, st));
    }
    GemmArgs go = gemm(b.HID, 4 * E, w.pj_w, w.pj_b, Xout, ldo, rows, E, 4 * E);
    go.flags = G_RES | G_OUT_MASK; go.rowmask = mask_out; go.ls = w.ls_ffn; go.R = b.R[1]; go.ldr = E;
    if (carry) return run_ffn(m, b.R[1], w.fc_wf, w.fc_c, go, b.HID, rows, E, st, b.stats, w.fc_s);
    LnArgs ln{}; ln.X = b.R[1]; ln.ldx = E; ln.Y = b.R[2]; ln.ldy = E; ln.w = w.ln_ffn_w; ln.b = w.ln_ffn_b; ln.rows = rows; ln.C = E;
    TRY(launch_ln(ln, st));
    return run_ffn(m, b.R[2], w.fc_w, w.fc_b, go, b.HID, rows, E, st);
  }
  if (c.win > 0) {
    LocalAttnArgs la{b.R[4], b.R[5], b.R[6], mask_out, b.R[0], B, To, E, c.vid_heads, c.win};
    TRY(launch_local_attn(la, st));
  } else {                                                         // mha_win_size = 0: global self-attention (blocks.py:339-343)
    GlobalAttnArgs ga{b.R[4], b.R[5], b.R[6], mask_out, b.R[0], B, To, E, c.vid_heads};
    TRY(launch_global_attn(ga, st));
  }
  // x' = skip * mask + ls_attn * (proj(ctx) + b)                       (blocks.py:586)
  GemmArgs gp = gemm(b.R[0], E, w.wp, w.bp, b.R[1], E, rows, E, E);
  gp.flags = G_RES | G_RES_MASK; gp.rowmask = mask_out; gp.ls = w.ls_attn;
  if (stride == 2) { gp.R = b.R[3]; gp.ldr = E; } else { gp.R = Xin; gp.ldr = ldx; }
  // out = x' + ls_ffn * ((ffn) * mask)                                  (blocks.py:589-590)
  GemmArgs go = gemm(b.HID, 4 * E, w.pj_w, w.pj_b, Xout, ldo, rows, E, 4 * E);
  go.flags = G_RES | G_OUT_MASK; go.rowmask = mask_out; go.ls = w.ls_ffn; go.R = b.R[1]; go.ldr = E;
  if (can_fuse_ln(m, w.wp, rows, E, E, A_ROWS)) {                 // ln_ffn(x') rides in the epilogue
    gp.ln_w = w.ln_ffn_w; gp.ln_b = w.ln_ffn_b; gp.Y = b.R[2]; gp.ldy = E;
    TRY(run_gemm(m, &gp, 1, A_ROWS, st));
  } else if (can_carry_ln(m, w.wp, w.fc_wf, rows, E, E, E)) {
    // ... or as row statistics: the projection writes (sum, sum of squares) of every x' row, ffn.fc runs on the raw x' with
    // ln_ffn folded into its weights and applies (mean, rstd) in its epilogue -- ln_ffn(x') is neither written nor read
    gp.stats_out = b.stats; gp.stats_w = STATS_W;
    TRY(run_gemm(m, &gp, 1, A_ROWS, st));
    return run_ffn(m, b.R[1], w.fc_wf, w.fc_c, go, b.HID, rows, E, st, b.stats, w.fc_s);
  } else {
    TRY(run_gemm(m, &gp, 1, A_ROWS, st));
    LnArgs ln{}; ln.X = b.R[1]; ln.ldx = E; ln.Y = b.R[2]; ln.ldy = E; ln.w = w.ln_ffn_w; ln.b = w.ln_ffn_b; ln.rows = rows; ln.C = E;
    TRY(launch_ln(ln, st));
  }
  TRY(run_ffn(m, b.R[2], w.fc_w, w.fc_b, go, b.HID, rows, E, st));
  return 0;
}

// From 30 000 pyramid rows on (one video of T = 16 384 has 32 640), in the f16x3 mode: trunk and output convolution of a head as
// ONE kernel, the trunk activations in registers (head_chain.hip).  One video per forward: 1.69 - 1.72 against 1.73 - 1.76 ms with
// the GEMM launches (268 tiles of 122 rows: one round of workgroups and a sliver); two videos per forward: +6 %.
static int head_chain_min_rows() {
  static const int v = getenv("DCF_HEAD_CHAIN_MIN_ROWS") ? atoi(getenv("DCF_HEAD_CHAIN_MIN_ROWS")) : 30000;     // developer switch
  return v;
}
static bool g_no_head_chain() {
  static const bool off = getenv("DCF_NO_HEAD_CHAIN") != nullptr;     // developer switch: GEMM + LayerNorm + output-convolution launches
  return off;
}
static bool can_chain_head(dcf_model* m, const HeadW& h, int rows, int Cin, int NO) {
  if (g_no_head_chain() || rows < head_chain_min_rows() || m->gemm_terms != GEMM_F16X3 || !h.chain[0] || !h.chain[1]) return false;
  if (h.conv.size() != 2 || !head_chain_supports(Cin, NO)) return false;
  for (int i = 0; i < 2; ++i)
    if (!m->wsplit.count(h.conv[i]) || m->wsplit_terms[h.conv[i]] != GEMM_F16X3) return false;
  return true;
}
static HeadChainArgs head_chain_args(dcf_model* m, const HeadW& h, Buffers& b, const Plan& pl, int NO, int mode, int query_major, float* out) {
  HeadChainArgs a{};
  a.X = b.F; a.ldx = m->cfg.E + TCN_HID; a.nbr = b.nbr_all; a.W1c = h.chain[0]; a.W2c = h.chain[1];
  a.ln1_w = h.ln_w[0]; a.ln1_b = h.ln_b[0]; a.ln2_w = h.ln_w[1]; a.ln2_b = h.ln_b[1];
  a.Wout = h.out_w; a.bout = h.out_b; a.lt = pl.d_lt; a.out = out; a.rows = pl.B * pl.lt.S; a.NO = NO; a.mode = mode;
  a.query_major = query_major; a.status = m->status;
  return a;
}
// algorithmic work of the launches a head kernel replaces: two k3 convolutions (C x 3C) and the output convolution; bytes: the
// input rows once, the outputs once
static int run_head_chain(dcf_model* m, const HeadChainArgs* a, int count, int Cin, hipStream_t st) {
  double flops = 0., bytes = 0.;
  for (int i = 0; i < count; ++i) {
    flops += 2.0 * a[i].rows * Cin * 3.0 * Cin * 2.0 + 2.0 * a[i].rows * 3.0 * Cin * a[i].NO;
    bytes += (double)a[i].rows * (Cin + a[i].NO) * 4.0;
  }
  ProfScope prof("gemm_f16x3<head_chain>", st, flops, bytes);
  return launch_head_chain(a, count, Cin, st);
}

// one head trunk (n x [k3 conv, LN, ReLU]) + output conv over the whole pyramid
// rows [row0, row0 + rows) of the pyramid (a whole pyramid, or one level)
static int run_head(dcf_model* m, const HeadW& h, Buffers& b, const Plan& pl, int Cin, int NO, int mode, int query_major,
                    float* out, hipStream_t st, int row0 = 0, int rows = -1) {
  if (row0 == 0 && rows < 0 && can_chain_head(m, h, pl.B * pl.lt.S, Cin, NO)) {
    const HeadChainArgs a = head_chain_args(m, h, b, pl, NO, mode, query_major, out);
    return run_head_chain(m, &a, 1, Cin, st);
  }
  const int rowsAll = rows >= 0 ? rows : pl.B * pl.lt.S;
  const int ldf = m->cfg.E + TCN_HID;
  const float* in = b.F + (int64_t)row0 * ldf;
  int64_t ldin = ldf;
  float* HA = b.HA + (int64_t)row0 * Cin;
  float* HB = b.HB + (int64_t)row0 * Cin;
  const uint8_t* nbr = b.nbr_all + row0;
  const float *last_ln_w = nullptr, *last_ln_b = nullptr;
  float* hst = b.hstats[0] + (int64_t)row0 * ((Cin + 63) / 64) * 2;
  int pending = -1, pend_w = 0;                                  // layer whose LayerNorm + ReLU the next convolution applies on load
  for (size_t i = 0; i < h.conv.size(); ++i) {
    GemmArgs g = gemm(in, ldin, h.conv[i], nullptr, HA, Cin, rowsAll, Cin, 3 * Cin);
    g.cin = Cin; g.nbr = nbr;
    float* outp = (in == HB) ? HA : HB;                          // ping-pong between the two trunk buffers
    if (pending >= 0) { norm_a(g, hst, Cin, pend_w, h.ln_w[pending], h.ln_b[pending]); pending = -1; }
    int sw = 0;
    if (can_fuse_ln(m, h.conv[i], rowsAll, Cin, 3 * Cin, A_ROWS_TAP3) && !g.a_stats) {
      g.C = nullptr; g.ln_w = h.ln_w[i]; g.ln_b = h.ln_b[i]; g.Y = outp; g.ldy = Cin; g.ln_relu = 1;
      TRY(run_gemm(m, &g, 1, A_ROWS_TAP3, st));
    } else if (i + 1 < h.conv.size() && !g.a_stats && can_norm_a(m, h.conv[i], h.conv[i + 1], rowsAll, Cin, &sw)) {
      g.C = outp; g.stats_out = hst; g.stats_w = sw;             // raw output + row statistics: the next convolution normalises it
      TRY(run_gemm(m, &g, 1, A_ROWS_TAP3, st));
      pending = (int)i; pend_w = sw;
    } else {
      g.C = outp;                                                // raw conv output, normalised in place ...
      TRY(run_gemm(m, &g, 1, A_ROWS_TAP3, st));
      if (i + 1 == h.conv.size()) {                              // ... or, for the last layer, by the output convolution on load
        last_ln_w = h.ln_w[i]; last_ln_b = h.ln_b[i];
      } else {
        LnArgs ln{}; ln.X = outp; ln.ldx = Cin; ln.Y = outp; ln.ldy = Cin; ln.w = h.ln_w[i]; ln.b = h.ln_b[i];
        ln.rows = rowsAll; ln.C = Cin; ln.relu = 1;
        TRY(launch_ln(ln, st));
      }
    }
    in = outp; ldin = Cin;
  }
  ConvOutArgs co{};
  co.ln_w = last_ln_w; co.ln_b = last_ln_b;
  co.X = in; co.ldx = ldin; co.nbr = nbr; co.W = h.out_w; co.bias = h.out_b; co.lt = pl.d_lt;
  co.out = query_major ? out : out + row0;
  co.rows = rowsAll; co.C = Cin; co.NO = NO; co.row0 = row0; co.mode = mode; co.query_major = query_major;
  TRY(launch_conv_out(co, st));
  return 0;
}

// Two head trunks of the same shape on the same pyramid (cls_head2 and reg_head): each pair of k3 convolutions is one
// grid of twice the workgroups (blockIdx.z picks the operand set), so the kernel runs two rounds of workgroups whose
// prologues and epilogues overlap instead of two single-round launches.
static int run_head_pair(dcf_model* m, const HeadW& h1, const HeadW& h2, Buffers& b, const Plan& pl, int Cin, int NO1, int mode1,
                         float* out1, int NO2, int mode2, float* out2, hipStream_t st) {
  const int rowsAll = pl.B * pl.lt.S;
  if (can_chain_head(m, h1, rowsAll, Cin, NO1) && can_chain_head(m, h2, rowsAll, Cin, NO2)) {
    const HeadChainArgs a[2] = {head_chain_args(m, h1, b, pl, NO1, mode1, 1, out1), head_chain_args(m, h2, b, pl, NO2, mode2, 1, out2)};
    return run_head_chain(m, a, 2, Cin, st);      // one grid for the two heads
  }
  bool pair = h1.conv.size() == h2.conv.size() && !h1.conv.empty() && m->gemm_terms != 0;
  for (size_t i = 0; pair && i < h1.conv.size(); ++i)
    pair = !can_fuse_ln(m, h1.conv[i], rowsAll, Cin, 3 * Cin, A_ROWS_TAP3) && m->wsplit.count(h1.conv[i]) && m->wsplit.count(h2.conv[i]);
  if (!pair) {
    TRY(run_head(m, h1, b, pl, Cin, NO1, mode1, 1, out1, st));
    return run_head(m, h2, b, pl, Cin, NO2, mode2, 1, out2, st);
  }
  const int ldf = m->cfg.E + TCN_HID;
  const float* in[2] = {b.F, b.F};
  int64_t ldin = ldf;
  float* buf[2][2] = {{b.HA, b.HB}, {b.HC, b.HD}};
  const HeadW* hs[2] = {&h1, &h2};
  int pending = -1, pend_w = 0;
  for (size_t i = 0; i < h1.conv.size(); ++i) {
    float* outp[2] = {buf[0][i & 1], buf[1][i & 1]};
    GemmArgs g[2];
    for (int k = 0; k < 2; ++k) {
      g[k] = gemm(in[k], ldin, hs[k]->conv[i], nullptr, outp[k], Cin, rowsAll, Cin, 3 * Cin);
      g[k].cin = Cin; g[k].nbr = b.nbr_all;
      if (pending >= 0) norm_a(g[k], b.hstats[k], Cin, pend_w, hs[k]->ln_w[pending], hs[k]->ln_b[pending]);
    }
    const bool was_pending = pending >= 0;
    pending = -1;
    int sw = 0;
    // the LayerNorm + ReLU between two layers rides in the next layer's A staging when both run on a kernel that can
    const bool carry = !was_pending && i + 1 < h1.conv.size() && can_norm_a(m, h1.conv[i], h1.conv[i + 1], rowsAll, Cin, &sw) &&
                       can_norm_a(m, h2.conv[i], h2.conv[i + 1], rowsAll, Cin, &sw);
    if (carry)
      for (int k = 0; k < 2; ++k) { g[k].stats_out = b.hstats[k]; g[k].stats_w = sw; }
    TRY(run_gemm(m, g, 2, A_ROWS_TAP3, st));
    if (carry) { pending = (int)i; pend_w = sw; }
    for (int k = 0; k < 2; ++k) {
      if (i + 1 < h1.conv.size() && !carry) {                    // the last layer is normalised by the output convolution on load
        LnArgs ln{}; ln.X = outp[k]; ln.ldx = Cin; ln.Y = outp[k]; ln.ldy = Cin; ln.w = hs[k]->ln_w[i]; ln.b = hs[k]->ln_b[i];
        ln.rows = rowsAll; ln.C = Cin; ln.relu = 1;
        TRY(launch_ln(ln, st));
      }
      in[k] = outp[k];
    }
    ldin = Cin;
  }
  const int NOs[2] = {NO1, NO2}, modes[2] = {mode1, mode2};
  float* outs[2] = {out1, out2};
  for (int k = 0; k < 2; ++k) {
    ConvOutArgs co{};
    co.X = in[k]; co.ldx = ldin; co.nbr = b.nbr_all; co.W = hs[k]->out_w; co.bias = hs[k]->out_b; co.lt = pl.d_lt;
    co.out = outs[k]; co.rows = rowsAll; co.C = Cin; co.NO = NOs[k]; co.row0 = 0; co.mode = modes[k]; co.query_major = 1;
    co.ln_w = hs[k]->ln_w.back(); co.ln_b = hs[k]->ln_b.back();
    TRY(launch_conv_out(co, st));
  }
  return 0;
}

// From 16 384 level-0 rows on (one video of T = 16 384), in the f16x3 mode, E = 256, 4 heads, <= 64 text tokens: the attention half
// of a fusion layer as ONE kernel (dec_chain.hip).
static int dec_chain_min_rows() {
  const int o = debug_option("dec_chain_min_rows", -1);            // dcf_debug_set_option (tests), then the developer switch
  if (o >= 0) return o;
  static const int v = getenv("DCF_DEC_CHAIN_MIN_ROWS") ? atoi(getenv("DCF_DEC_CHAIN_MIN_ROWS")) : 16384;   // (one video per call: 1.70 against 1.74 ms)
  return v;
}
static bool can_chain_dec(dcf_model* m, const DecW& w, const LevelTable* lt, int rows, int64_t ldx, int Lk) {
  static const bool off = getenv("DCF_NO_DEC_CHAIN") != nullptr;    // developer switch: the separate launches
  const dcf_config& c = m->cfg;
  return !off && !lt && m->gemm_terms == GEMM_F16X3 && w.wq_chain && w.wp_chain && dec_chain_supports(c.E, c.fusion_heads, Lk) &&
         c.TE % 32 == 0 && rows >= dec_chain_min_rows() && ldx % 4 == 0;
}

// XAttNFusion._forward (fusion.py:56-66): n x TransformerDecoder (blocks.py:632-650) + ln_out.
// X [rows][ldx] is updated in place; the final ln_out goes to out [rows][ld_out].  Either one level of B sequences
// of T rows (lt == nullptr) or the whole pyramid (lt != nullptr: rows ordered [level][b][t], neighbour flags `nbr`
// delimit the sequences for the depthwise conv, the attention core is launched per level).
// carry_out != nullptr: the caller can take fusion.ln_out as row statistics (b.stats of the raw stream left in X) instead of
// the normalised rows in `out`; *carry_out says which of the two happened.
static int run_fusion(dcf_model* m, Buffers& b, float* X, int64_t ldx, int B, int T, const LevelTable* lt, const uint8_t* mask,
                      const uint8_t* nbr, const TextMeta* dm, int Lk, float* out, int64_t ld_out, hipStream_t st,
                      bool* carry_out = nullptr) {
  if (carry_out) *carry_out = false;
  const dcf_config& c = m->cfg;
  const int E = c.E;
  const int rows = lt ? lt->start[lt->n_levels] : B * T;
  for (size_t li = 0; li < m->dec.size(); ++li) {
    const DecW& w = m->dec[li];
    bool carry = false;
    const bool chain = can_chain_dec(m, w, lt, rows, ldx, Lk);
    TextLnArgs tl{*dm, b.kvn, b.kvmask, w.ln_kv_w, w.ln_kv_b, Lk, c.TE};
    TRY(launch_text_ln(tl, B, st));
    GemmArgs gkv[2] = {gemm(b.kvn, c.TE, w.wk, w.bk, b.Kt, E, B * Lk, E, c.TE), gemm(b.kvn, c.TE, w.wv, w.bv, b.Vt, E, B * Lk, E, c.TE)};
    TRY(run_gemm(m, gkv, 2, A_ROWS, st));
    if (chain) {
      // q3 = adaln(q) * scale + shift straight from the raw stream: ln_xattn_q, the depthwise convolution, q_norm, the query
      // projection, the cross attention and the modulating projection in one kernel; ln_ffn(q3) as row statistics where the FFN
      // can take them, by the LayerNorm kernel otherwise
      const int lk2 = Lk <= 32 ? 1 : 2;
      TRY(launch_kv_image(b.Kt, b.Vt, b.kvmask, B, Lk, lk2, b.kvimg, b.kmadd, st));
      carry = !g_no_carry() && w.fc_wf && m->wsplit.count(w.fc_wf) &&
              (can_chain_ffn_rows(m, rows, E) || gemm_can_carry_stats(rows, 4 * E, E, 1, m->wsplit_terms[w.fc_wf]));
      DecChainArgs da{};
      da.X = X; da.ldx = ldx; da.mask = mask; da.ln_q_w = w.ln_q_w; da.ln_q_b = w.ln_q_b; da.dw = w.dw; da.qn_w = w.qn_w; da.qn_b = w.qn_b;
      da.Wq = w.wq_chain; da.bq = w.bq; da.KV = b.kvimg; da.kmask = b.kmadd; da.Wp = w.wp_chain; da.bp = w.bp_il;
      da.Q3 = b.R[2]; da.ldq = E; da.stats_out = carry ? b.stats : nullptr; da.stats_w = STATS_W;
      da.B = B; da.T = T; da.affine = c.xattn_affine; da.lk2 = lk2; da.status = m->status; da.attn_single = c.attn_mode == 1;
      {
        ProfScope prof("gemm_f16x3<dec_chain>", st, 2.0 * rows * E * 3.0 * E + 4.0 * rows * E * Lk, (double)rows * E * 4.0 * 2.0);
        TRY(launch_dec_chain(da, st));
      }
      if (!carry) {
        LnArgs ln{}; ln.X = b.R[2]; ln.ldx = E; ln.Y = b.R[0]; ln.ldy = E; ln.w = w.ln_ffn_w; ln.b = w.ln_ffn_b; ln.rows = rows; ln.C = E;
        TRY(launch_ln(ln, st));
      }
    } else {
    DecPreArgs dp{X, ldx, mask, w.ln_q_w, w.ln_q_b, w.dw, w.qn_w, w.qn_b, b.R[0], b.R[1], lt ? 1 : B, lt ? rows : T, E};
    dp.nbr = lt ? nbr : nullptr;
    dp.affine = c.xattn_affine;
    TRY(launch_dec_pre(dp, st));
    GemmArgs gq = gemm(b.R[0], E, w.wq, w.bq, b.R[2], E, rows, E, E);
    TRY(run_gemm(m, &gq, 1, A_ROWS, st));
    if (lt) {
      for (int l = 0; l < lt->n_levels; ++l) {
        XAttnArgs xa{b.R[2] + (int64_t)lt->start[l] * E, b.Kt, b.Vt, b.kvmask, b.R[0] + (int64_t)lt->start[l] * E, B, lt->T[l], Lk, E, c.fusion_heads, m->status, c.attn_mode == 1};
        TRY(launch_xattn(xa, st));
      }
    } else {
      XAttnArgs xa{b.R[2], b.Kt, b.Vt, b.kvmask, b.R[0], B, T, Lk, E, c.fusion_heads, m->status, c.attn_mode == 1};
      TRY(launch_xattn(xa, st));
    }
    if (m->gemm_terms != 0 && w.wp_il && m->wsplit.count(w.wp_il) && gemm_can_fuse_adaln(rows, 2 * E, E)) {
      // q3 = Xa * scale + shift in the epilogue of the projection (blocks.py:643-646): the (rows, 2E) scale / shift tensor is
      // never written; Xn = ln_ffn(q3) by the LayerNorm kernel
      GemmArgs gh = gemm(b.R[0], E, w.wp_il, w.bp_il, b.R[2], E, rows, 2 * E, E);
      gh.flags = G_ADALN; gh.R = b.R[1]; gh.ldr = E;
      carry = can_carry_ln(m, w.wp_il, w.fc_wf, rows, 2 * E, E, E);            // ln_ffn(q3) as row statistics (see run_encoder)
      if (carry) { gh.stats_out = b.stats; gh.stats_w = STATS_W; }
      TRY(run_gemm(m, &gh, 1, A_ROWS, st));
      if (!carry) {
        LnArgs ln{}; ln.X = b.R[2]; ln.ldx = E; ln.Y = b.R[0]; ln.ldy = E; ln.w = w.ln_ffn_w; ln.b = w.ln_ffn_b; ln.rows = rows; ln.C = E;
        TRY(launch_ln(ln, st));
      }
    } else {
      GemmArgs gh = gemm(b.R[0], E, w.wp, w.bp, b.H2, 2 * E, rows, 2 * E, E);
      TRY(run_gemm(m, &gh, 1, A_ROWS, st));
      TRY(launch_dec_mid(b.R[1], b.H2, w.ln_ffn_w, w.ln_ffn_b, b.R[2], b.R[0], rows, E, st));
    }
    }   // (!chain)
    GemmArgs go = gemm(b.HID, 4 * E, w.pj_w, w.pj_b, X, ldx, rows, E, 4 * E);
    go.flags = G_RES | G_OUT_MASK; go.rowmask = mask; go.ls = w.ls_ffn; go.R = b.R[2]; go.ldr = E;
    const float* fc_in = carry ? b.R[2] : b.R[0];
    const float *fc_w = carry ? w.fc_wf : w.fc_w, *fc_b = carry ? w.fc_c : w.fc_b;
    if (li + 1 == m->dec.size() && m->fus_out_w && can_fuse_ln(m, w.pj_w, rows, E, 4 * E, A_ROWS)) {
      // last layer: only ln_out(x) is consumed afterwards (fusion.py:64-66), the raw stream is not written
      GemmArgs gf = gemm(fc_in, E, fc_w, fc_b, b.HID, 4 * E, rows, 4 * E, E);
      gf.flags = G_GELU;
      if (carry) { gf.stats_in = b.stats; gf.ln_s = w.fc_s; gf.stats_slots = E / STATS_W; gf.stats_w = STATS_W; }
      TRY(run_gemm(m, &gf, 1, A_ROWS, st));
      go.C = nullptr; go.ln_w = m->fus_out_w; go.ln_b = m->fus_out_b; go.Y = out; go.ldy = ld_out;
      TRY(run_gemm(m, &go, 1, A_ROWS, st));
      return 0;
    }
    if (li + 1 == m->dec.size() && carry_out && m->fus_out_w && ldx == E &&
        m->embd_fc_wf && m->wsplit.count(m->embd_fc_wf) && !g_no_carry() &&
        m->wsplit.count(w.pj_w) && gemm_can_carry_stats(rows, E, 4 * E, 1, m->wsplit_terms[w.pj_w]) &&
        gemm_can_carry_stats(rows, E, E, 1, m->wsplit_terms[m->embd_fc_wf])) {
      // last layer: ffn.proj leaves the raw stream in X together with its row statistics, vid_net.embd_fc (ln_out folded into its
      // weights) applies them: ln_out(x) is neither written nor read (fusion.py:64-66 -> video_net.py:131)
      go.stats_out = b.stats; go.stats_w = STATS_W;
      // (the fc half reads b.stats before the proj half overwrites it: stream order in the GEMM pair; in the one-kernel form a
      // wave reads the statistics of its own rows at its start and writes them at its end)
      TRY(run_ffn(m, fc_in, fc_w, fc_b, go, b.HID, rows, E, st, carry ? b.stats : nullptr, w.fc_s));
      *carry_out = true;
      return 0;
    }
    TRY(run_ffn(m, fc_in, fc_w, fc_b, go, b.HID, rows, E, st, carry ? b.stats : nullptr, w.fc_s));
  }
  if (!m->fus_out_w) return 0;          // dcf_op_decoder: the bare layer stack, result left in X
  LnArgs ln{}; ln.X = X; ln.ldx = ldx; ln.Y = out; ln.ldy = ld_out; ln.w = m->fus_out_w; ln.b = m->fus_out_b; ln.rows = rows; ln.C = E;
  TRY(launch_ln(ln, st));
  return 0;
}

// The videos of one forward: all padded to the same T, video v with nq[v] queries; the queries of all videos are one flat
// list (text / outputs in video order).  One video is the reference's call (model.py:496 asserts bs == 1); several are
// the throughput extension dcf_forward_eval_videos: after vid_map every kernel works on rows [query][t] and does not
// care which video a query belongs to.
constexpr int DCF_MAX_VIDEOS = 16;
struct VideoSet {
  int nvid = 0;
  const float* vid[DCF_MAX_VIDEOS];
  const float* shallow[DCF_MAX_VIDEOS];
  const uint8_t* mask[DCF_MAX_VIDEOS];
  const float* text_cls[DCF_MAX_VIDEOS];      // (nq[v], D); with gate_override: the gate (nq, T)
  int nq[DCF_MAX_VIDEOS];
  float* logits1_out = nullptr;               // optional (nq, S): the logits of the first cls_head (fpn_logits1, model.py:445,471)
};

struct MaskPtrs { const uint8_t* p[DCF_MAX_VIDEOS]; };
__global__ void k_gather_masks(MaskPtrs mp, uint8_t* __restrict__ dst, int T0) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < T0) dst[(size_t)blockIdx.y * T0 + t] = mp.p[blockIdx.y][t];
}


// =============================================================================================
// One long video cut at pyramid level k (dist.py hybrid_forward, SURVEY 8e: the NQ = 1 corner of T-sharding).
// A rank holds TWO ordinary power-of-two pyramids: the NARROW one, levels 0 .. k on a window of Tn clips (everything a forward
// does in front of level k + 1), and the COARSE one, levels k .. L - 1 on a window of Tc level-k rows whose level 0 is the
// all-gathered level-k feature map.  Three phases with one exchange between each two:
//   phase 1  narrow window: vid_map, early fusion, embedding, levels 0 .. k           -> level-k features (narrow window)
//   phase 2  coarse window: levels k + 1 .. L - 1; cls_head on both pyramids; the refinement TCN on the narrow window's clips over the
//            stacked logits of ALL levels (model.py:449-458); pooled down to level k    -> refined level-k map (narrow window)
//   phase 3  refined map pooled down the coarse pyramid; cls_head2 / reg_head on both    -> outputs of levels <= k (narrow) and > k (coarse)
// Windows are treated as sequences (zero padding at their ends); the halos of dist.hybrid_plan absorb that.
// =============================================================================================
struct HybridState {
  bool valid = false;
  int k = 0, Tn = 0, Tc = 0, B = 0, Lk = 0;
  Buffers bn{};
  Plan pn{}, pc{}, pch{};                     // narrow pyramid; coarse pyramid (levels k .. L-1); its levels k+1 .. (what the heads see)
  float *Fc = nullptr, *logits1c = nullptr, *stacked = nullptr;
  uint8_t *maskc = nullptr, *nbrc = nullptr;
};

static void free_hybrid(dcf_model* m) {
  if (!m->hyb) return;
  for (Plan* p : {&m->hyb->pn, &m->hyb->pc, &m->hyb->pch}) if (p->d_lt) (void)hipFree(p->d_lt);
  delete m->hyb;
  m->hyb = nullptr;
}

static int make_plan(dcf_model* m, Plan& p, const int* Tl, int n, int B, const float* scales, hipStream_t st) {
  p.T0 = Tl[0]; p.B = B; p.L = n;
  LevelTable& lt = p.lt;
  lt = LevelTable{};
  lt.n_levels = n; lt.B = B;
  int acc = 0;
  for (int l = 0; l < n; ++l) {
    lt.T[l] = Tl[l]; lt.off[l] = acc; lt.start[l] = B * acc; lt.scale[l] = scales ? scales[l] : 1.f;
    acc += Tl[l];
  }
  lt.S = acc; lt.start[n] = B * acc;
  if (!p.d_lt) DCF_HIP(hipMalloc(&p.d_lt, sizeof(LevelTable)));
  DCF_HIP(hipMemcpyAsync(p.d_lt, &p.lt, sizeof(LevelTable), hipMemcpyHostToDevice, st));
  DCF_HIP(hipStreamSynchronize(st));
  (void)m;
  return 0;
}

// size of the coarse pyramid's own buffers behind the forward's workspace (phase 1 reserves them: no reallocation between phases)
static size_t hybrid_extra_bytes(const dcf_config& c, int B, int Tc, int LC, int Tn, int L) {
  size_t rows = 0;
  for (int j = 0; j < LC; ++j) rows += (size_t)B * (Tc >> j);
  const size_t EH = c.E + TCN_HID;
  return rows * EH * 4 + 2 * (rows + 256) + rows * 4 + (size_t)B * Tn * L * 4 + 8 * 256;
}

static int hybrid_take(dcf_model* m, const Buffers& b, const Plan& pl, int B, int Lk, hipStream_t st) {
  HybridState& h = *m->hyb;
  const dcf_config& c = m->cfg;
  const int E = c.E, ldf = E + TCN_HID, k = h.k;
  h.bn = b; h.B = B; h.Lk = Lk;
  h.pn.lt = pl.lt; h.pn.T0 = pl.T0; h.pn.B = pl.B; h.pn.L = pl.L;
  if (!h.pn.d_lt) DCF_HIP(hipMalloc(&h.pn.d_lt, sizeof(LevelTable)));
  DCF_HIP(hipMemcpyAsync(h.pn.d_lt, pl.d_lt, sizeof(LevelTable), hipMemcpyDeviceToDevice, st));
  // the coarse pyramid's buffers
  const int LC = c.n_levels - k;
  Arena a{m->hyb_extra_ptr, 0, m->hyb_extra, false};
  size_t rows = 0;
  for (int j = 0; j < LC; ++j) rows += (size_t)B * (h.Tc >> j);
  h.Fc = a.take<float>(rows * ldf);
  h.maskc = a.take<uint8_t>(rows);
  h.nbrc = a.take<uint8_t>(rows);
  h.logits1c = a.take<float>(rows);
  h.stacked = a.take<float>((size_t)B * h.Tn * c.n_levels);
  DCF_CHECK(a.off <= m->hyb_extra, "internal: hybrid workspace");
  // level-k features of the narrow window -> caller (B, Tn >> k, E)
  DCF_HIP(hipMemcpy2DAsync(m->hyb_feat_out, (size_t)E * 4, b.F + (int64_t)pl.lt.start[k] * ldf, (size_t)ldf * 4, (size_t)E * 4,
                           (size_t)B * pl.lt.T[k], hipMemcpyDeviceToDevice, st));
  h.valid = true;
  return 0;
}

// u[b][t][l] = logits1 of level l at the narrow window's clip t (nearest: index t >> l), times the clip's mask for l > 0 (model.py:449-455);
// levels > k come from the coarse pyramid: its level j = l - k at index (((t >> k) + off_k) >> j)
__global__ void k_hybrid_stack(const float* __restrict__ l1n, const LevelTable* __restrict__ ltn, const float* __restrict__ l1c,
                               const LevelTable* __restrict__ ltc, const uint8_t* __restrict__ mask0, float* __restrict__ out,
                               int B, int Tn, int k, int L, int off_k) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= B * Tn) return;
  const int b = r / Tn, t = r - b * Tn;
  const float m0 = mask0[r] ? 1.f : 0.f;
  for (int l = 0; l < L; ++l) {
    float u;
    if (l <= k) u = l1n[ltn->start[l] + b * ltn->T[l] + (t >> l)];
    else {
      const int j = l - k - 1;                                    // level of the heads' coarse table (levels k + 1 ..)
      int i = ((t >> k) + off_k) >> (j + 1);
      i = i < 0 ? 0 : (i < ltc->T[j] ? i : ltc->T[j] - 1);
      u = l1c[ltc->start[j] + b * ltc->T[j] + i];
    }
    out[(int64_t)r * L + l] = l > 0 ? u * m0 : u;
  }
}

static int tcn_stack_env() {
  static const int v = getenv("DCF_TCN_STACK") ? atoi(getenv("DCF_TCN_STACK")) : -1;     // developer switch: leading TCN layers per launch
  return v;
}
static RefineArgs refine_args(dcf_model* m) {
  RefineArgs ra{};
  ra.w_in = m->tcn_in_w; ra.b_in = m->tcn_in_b;
  ra.host_w_dil = m->tcn_wd.data(); ra.host_b_dil = m->tcn_bd.data(); ra.host_w_pw = m->tcn_wp.data();
  ra.host_b_pw = m->tcn_bp.data(); ra.host_ln_w = m->tcn_lnw.data(); ra.host_ln_b = m->tcn_lnb.data();
  ra.w_out = m->tcn_out_w; ra.b_out = m->tcn_out_b;
  ra.host_frag = (!m->tcn_frag.empty() && debug_option("tcn_frag", 1) != 0) ? m->tcn_frag.data() : nullptr;
  ra.stack_layers = debug_option("tcn_stack", tcn_stack_env());          // (dcf_debug_set_option: 0 = layer by layer)
  ra.f16 = m->gemm_terms == GEMM_F16X3; ra.status = m->status;
  return ra;
}

// Buffers of the coarse pyramid's HEAD levels (k + 1 ..): the narrow pyramid's scratch with F / masks / logits re-based
static Buffers hybrid_coarse_heads(const HybridState& h, int ldf) {
  Buffers bc = h.bn;
  const int s1 = h.pc.lt.start[1];
  bc.F = h.Fc + (int64_t)s1 * ldf; bc.mask_all = h.maskc + s1; bc.nbr_all = h.nbrc + s1; bc.logits1 = h.logits1c;
  return bc;
}

static int hybrid_phase2(dcf_model* m, const float* featk_c, const uint8_t* maskk_c, int off_k, float* refk_out, hipStream_t st) {
  HybridState& h = *m->hyb;
  const dcf_config& c = m->cfg;
  const int E = c.E, ldf = E + TCN_HID, k = h.k, L = c.n_levels, LC = L - k, B = h.B;
  // ---- the coarse pyramid: level 0 = the gathered level-k features, its masks from the level-k validity of the window
  DCF_HIP(hipMemcpy2DAsync(h.Fc, (size_t)ldf * 4, featk_c, (size_t)E * 4, (size_t)E * 4, (size_t)B * h.Tc, hipMemcpyDeviceToDevice, st));
  for (int b = 0; b < B; ++b) DCF_HIP(hipMemcpyAsync(h.maskc + (size_t)b * h.Tc, maskk_c, (size_t)h.Tc, hipMemcpyDeviceToDevice, st));
  TRY(launch_pyramid_masks(h.maskc, h.nbrc, B, h.Tc, LC, h.pc.lt.start[LC], st));
  const LevelTable& lc = h.pc.lt;
  for (int j = 1; j < LC; ++j) {
    const float* xin = h.Fc + (int64_t)lc.start[j - 1] * ldf;
    float* xo = h.Fc + (int64_t)lc.start[j] * ldf;
    if (c.pool_only) TRY(launch_dwconv3(xin, ldf, h.maskc + lc.start[j - 1], m->pool_w[k + j], xo, ldf, B, lc.T[j - 1], 2, E, st));
    else TRY(run_encoder(m, m->branch[k + j], h.bn, xin, ldf, h.maskc + lc.start[j - 1], h.maskc + lc.start[j], B, lc.T[j - 1], 2, xo, ldf, st));
  }
  // ---- cls_head on both pyramids (level-major rows)
  TRY(run_head(m, m->cls1, h.bn, h.pn, E, 1, 0, 0, h.bn.logits1, st));
  Buffers bc = hybrid_coarse_heads(h, ldf);
  if (LC > 1) TRY(run_head(m, m->cls1, bc, h.pch, E, 1, 0, 0, h.logits1c, st));
  // ---- the refinement TCN on the narrow window's clips over the stacked logits of all levels
  const int rows0 = B * h.Tn;
  hipLaunchKernelGGL(k_hybrid_stack, dim3((rows0 + 255) / 256), dim3(256), 0, st, (const float*)h.bn.logits1, (const LevelTable*)h.pn.d_lt,
                     (const float*)h.logits1c, (const LevelTable*)h.pch.d_lt, (const uint8_t*)h.bn.mask_all, h.stacked, B, h.Tn, k, L, off_k);
  DCF_HIP(hipGetLastError());
  RefineArgs ra = refine_args(m);
  ra.stacked = h.stacked; ra.mask_all = h.bn.mask_all;
  ra.bufA = h.bn.tcnA; ra.bufB = h.bn.tcnB; ra.F = h.bn.F; ra.ldf = ldf; ra.E = E;
  ra.B = B; ra.T0 = h.Tn; ra.n_levels = L; ra.n_layers = L;
  TRY(launch_refine(ra, h.pn.lt, st));
  const LevelTable& ln = h.pn.lt;
  for (int l = 1; l <= k; ++l)
    TRY(launch_refine_pool(h.bn.F, ldf, E, h.bn.mask_all + ln.start[l - 1], ln.start[l - 1], ln.start[l], B, ln.T[l - 1], st));
  DCF_HIP(hipMemcpy2DAsync(refk_out, (size_t)TCN_HID * 4, h.bn.F + (int64_t)ln.start[k] * ldf + E, (size_t)ldf * 4, (size_t)TCN_HID * 4,
                           (size_t)B * ln.T[k], hipMemcpyDeviceToDevice, st));
  return 0;
}

static int hybrid_phase3(dcf_model* m, const float* refk_c, float* logits_n, float* offsets_n, uint8_t* masks_n, float* logits_c,
                         float* offsets_c, uint8_t* masks_c, hipStream_t st) {
  HybridState& h = *m->hyb;
  const dcf_config& c = m->cfg;
  const int E = c.E, ldf = E + TCN_HID, LC = c.n_levels - h.k, B = h.B;
  const LevelTable& lc = h.pc.lt;
  DCF_HIP(hipMemcpy2DAsync(h.Fc + E, (size_t)ldf * 4, refk_c, (size_t)TCN_HID * 4, (size_t)TCN_HID * 4, (size_t)B * h.Tc, hipMemcpyDeviceToDevice, st));
  for (int j = 1; j < LC; ++j)
    TRY(launch_refine_pool(h.Fc, ldf, E, h.maskc + lc.start[j - 1], lc.start[j - 1], lc.start[j], B, lc.T[j - 1], st));
  TRY(run_head_pair(m, m->cls2, m->reg, h.bn, h.pn, E + TCN_HID, 1, 0, logits_n, 2, 1, offsets_n, st));
  const int rows_n = h.pn.lt.start[h.pn.lt.n_levels];
  hipLaunchKernelGGL(k_masks_out, dim3((rows_n + 255) / 256), dim3(256), 0, st, (const uint8_t*)h.bn.mask_all, masks_n,
                     (const LevelTable*)h.pn.d_lt, (const unsigned*)m->status, logits_n);
  DCF_HIP(hipGetLastError());
  if (LC > 1) {
    Buffers bc = hybrid_coarse_heads(h, ldf);
    TRY(run_head_pair(m, m->cls2, m->reg, bc, h.pch, E + TCN_HID, 1, 0, logits_c, 2, 1, offsets_c, st));
    const int rows_c = h.pch.lt.start[h.pch.lt.n_levels];
    hipLaunchKernelGGL(k_masks_out, dim3((rows_c + 255) / 256), dim3(256), 0, st, (const uint8_t*)bc.mask_all, masks_c,
                       (const LevelTable*)h.pch.d_lt, (const unsigned*)m->status, logits_c);
    DCF_HIP(hipGetLastError());
  }
  return 0;
}

static int forward(dcf_model* m, const VideoSet& vs, int T0, int nq,
                   const float* const* text, const uint8_t* const* text_mask, const int32_t* text_len,
                   const float* gate_override, float* logits_out, float* offsets_out,
                   uint8_t* masks_out, hipStream_t st) {
  const dcf_config& c = m->cfg;
  const int E = c.E, D = c.D;
  const int L = m->hyb_levels > 0 ? m->hyb_levels : c.n_levels;      // dcf_hybrid_phase1: the pyramid up to the split level only
  const int nvid = vs.nvid;
  if (m->hyb) m->hyb->valid = false;              // the phases of a level-cut forward live in this workspace
  DCF_CHECK(m->finalized, "dcf_forward_eval: model not finalized");
  DCF_CHECK(T0 > 0 && nq > 0 && nvid >= 1 && nvid <= DCF_MAX_VIDEOS, "dcf_forward_eval: empty input");
  DCF_CHECK(!(gate_override && nvid != 1), "the externally gated forward takes one video");
  int video_of[DCF_MAX_VIDEOS * 64];           // flat query -> video
  {
    int tot = 0;
    for (int v = 0; v < nvid; ++v) {
      DCF_CHECK(vs.nq[v] >= 1 && tot + vs.nq[v] <= DCF_MAX_VIDEOS * 64, "dcf_forward_eval: bad query count of video %d", v);
      for (int i = 0; i < vs.nq[v]; ++i) video_of[tot++] = v;
    }
    DCF_CHECK(tot == nq, "dcf_forward_eval: query counts do not add up");
  }
  const uint8_t* vid_mask = vs.mask[0];
  // vid_net.stride = sv > 1 (video_net.py:59-74, worker_v2.py:285-286): vid_map and the early fusion run on the T0 input clips, the first
  // log2(sv) embedding convolutions halve the sequence, the pyramid starts at Tp = T0 / sv
  const int sv = vid_stride_of(c);
  int npre = 0;
  while ((1 << npre) < sv) ++npre;
  DCF_CHECK(T0 % (sv << (L - 1)) == 0, "T=%d must be a multiple of vid_net.stride * 2^(levels-1)=%d", T0, sv << (L - 1));
  const int Tp = T0 / sv;
  const int half = c.win / 2;
  DCF_CHECK(half == 0 || c.pool_only || (Tp >> (L - 1)) % half == 0, "T=%d: coarsest level must be a multiple of win//2=%d (blocks.py:216)", T0, half);
  if (c.use_abs_pe) DCF_CHECK(m->pe && m->pe_T == Tp, "position encoding for T=%d not set (dcf_model_set_pe)", Tp);
  DCF_CHECK(!(gate_override && sv > 1), "the externally gated (T-sharded) forward takes vid_net.stride = 1");
  const int Bmax = std::min(nq, c.max_batch > 0 ? c.max_batch : 8);
  DCF_CHECK(Bmax <= DCF_MAX_BATCH, "max_batch %d > %d", Bmax, DCF_MAX_BATCH);
  DCF_CHECK(nvid == 1 || Bmax <= 16, "several videos per forward need max_batch <= 16 (got %d)", Bmax);
  int Lk = 1;
  for (int q = 0; q < nq; ++q) {
    DCF_CHECK(text_len[q] >= 1 && text[q], "text %d is empty", q);
    Lk = std::max(Lk, (int)text_len[q]);
  }
  int S = 0;
  for (int l = 0; l < L; ++l) S += Tp >> l;

  // ---- workspace
  Buffers b{};
  {
    Arena dry{nullptr, 0, 0, true};
    carve(dry, c, T0, Bmax, nq, S, Lk, nvid, b);
    (void)dry.take<char>(m->hyb_extra);
    if (dry.off > m->arena_bytes) {
      DCF_CHECK(!m->capturing, "internal: workspace growth during graph capture");
      drop_graph(m, true);                       // the eager call that grows the workspace still counts as the first sighting
      DCF_HIP(hipStreamSynchronize(st));
      if (m->arena) DCF_HIP(hipFree(m->arena));
      m->arena = nullptr; m->arena_bytes = 0;
      DCF_HIP(hipMalloc(&m->arena, dry.off));
      m->arena_bytes = dry.off;
    }
    Arena real{m->arena, 0, m->arena_bytes, false};
    carve(real, c, T0, Bmax, nq, S, Lk, nvid, b);
    m->hyb_extra_ptr = real.take<char>(m->hyb_extra);
  }
  // ---- per video: sidekick scores and the query-independent halves of vid_map
  DCF_CHECK(!(gate_override && c.scat), "opt.model.scat needs the sidekick scores: the externally gated (T-sharded) forward does not take them");
  // The sidekick scores ride on the shallow half of vid_map where they can: that GEMM streams exactly the (D, T) matrix the scores
  // are a reduction of, so the scoring pass's read of it (4 KB per clip) disappears; otherwise three launches of their own.
  bool scores_on_gemm = !gate_override && fuse_scores_on() && m->vid_w2 && m->gemm_terms != 0 && m->wsplit.count(m->vid_w2) &&
                        m->wsplit_ldw[m->vid_w2] == m->vid_ldw && E % 128 == 0 && T0 % 4 == 0;
  for (int v = 0; v < nvid; ++v) scores_on_gemm = scores_on_gemm && vs.nq[v] <= GEMM_SCORE_MAXQ;
  {
    // ... and the text vectors of a video's queries have to fit beside the GEMM's tile in the default 64 KiB of LDS (the tile and its
    // statistics block take ~22 KiB): 4 queries x D = 2048 do, D = 4096 does not -- the scoring kernels serve those
    int maxq = 0;
    for (int v = 0; v < nvid; ++v) maxq = std::max(maxq, vs.nq[v]);
    scores_on_gemm = scores_on_gemm && (size_t)maxq * D * sizeof(float) <= 40960;
  }
  int q_of[DCF_MAX_VIDEOS];                          // first query row of video v in tn / correl
  for (int v = 0, q_off = 0; v < nvid; q_off += vs.nq[v], ++v) q_of[v] = q_off;
  if (!gate_override) {                              // the scores of every video's queries
    DCF_CHECK(nvid <= SCORE_MAXVID, "%d videos per forward > %d", nvid, SCORE_MAXVID);
    ScoreArgs sa{};
    for (int v = 0; v < nvid; ++v) {
      sa.shallow[v] = vs.shallow[v]; sa.text_cls[v] = vs.text_cls[v]; sa.nq[v] = vs.nq[v]; sa.qoff[v] = q_of[v];
    }
    sa.nvid = nvid; sa.tn = b.tn; sa.partial = b.partial; sa.correl = b.correl; sa.D = D; sa.T = T0; sa.NQ = nq; sa.norm = c.norm;
    if (scores_on_gemm) TRY(launch_text_cls_norm(sa, st));
    else TRY(launch_sidekick(sa, st));
  }
  // Gate first, expert product second (model.py:531-543): `vid * all_weight` is zero outside the top-k blocks, so the rows of
  // W1 . vid under a closed gate are never used -- with every query of a video in THIS chunk (nq <= max_batch) the gate of all of them
  // is known before the expert GEMMs start, and a row tile that none of the video's queries keeps is skipped (GemmArgs::tile_skip).
  // One query per video: the gate keeps int(0.3 n) of n blocks of `sn` clips, ~half of the 64-row tiles touch none of them;
  // q queries: ~0.5^q of the tiles (independent gates).  Several chunks of queries / the externally gated forward: every tile.
  const int nflags = (T0 + 63) / 64;
  static const bool no_gate_skip = getenv("DCF_NO_GATE_SKIP") != nullptr;        // developer switch: every row tile of the expert product
  const bool gate_first = !gate_override && nq <= Bmax && m->vid_w1 && m->gemm_terms != 0 && !no_gate_skip && debug_option("gate_skip", 1) != 0;
  unsigned long long vmap0 = 0;
  for (int i = 0; i < nq && i < 16; ++i) vmap0 |= (unsigned long long)video_of[i] << (4 * i);
  {
    // the deep and shallow halves of vid_map of every video have the same shape: three of them share a grid (blockIdx.z
    // selects the operand set), so five videos are four full launches instead of five two-thirds-full ones
    GemmArgs g[3];
    int ng = 0;
    auto flush = [&]() -> int {
      if (ng == 0) return 0;
      for (int i = 0; i < ng; ++i) { g[i].ldw = m->vid_ldw; g[i].a_scale = 1.f; }
      const int rc = run_gemm(m, g, ng, A_CHANMAJOR, st);
      ng = 0;
      return rc;
    };
    auto shallow_half = [&](int v) -> int {
      g[ng] = gemm(vs.shallow[v], T0, m->vid_w2, nullptr, b.P2 + (size_t)v * T0 * E, E, T0, E, D);
      if (scores_on_gemm) {
        g[ng].score_tn = b.tn + (size_t)q_of[v] * D; g[ng].score_out = b.correl + (size_t)q_of[v] * T0;
        g[ng].score_nq = vs.nq[v]; g[ng].score_norm = c.norm;
      }
      if (++ng == 3) return flush();
      return 0;
    };
    auto expert_half = [&](int v, bool skip) -> int {
      g[ng] = gemm(vs.vid[v], T0, m->vid_w1, nullptr, b.P1 + (size_t)v * T0 * E, E, T0, E, D);
      if (skip) { g[ng].tile_skip = b.tile_flags + (size_t)q_of[v] * nflags; g[ng].skip_nq = vs.nq[v]; g[ng].skip_stride = nflags; }
      if (++ng == 3) return flush();
      return 0;
    };
    auto gather_masks = [&]() -> int {
      if (nvid > 1) {                               // the videos' masks side by side: one launch (it was one copy node per video)
        MaskPtrs mp{};
        for (int v = 0; v < nvid; ++v) mp.p[v] = vs.mask[v];
        hipLaunchKernelGGL(k_gather_masks, dim3((unsigned)((T0 + 255) / 256), (unsigned)nvid), dim3(256), 0, st, mp, b.maskv, T0);
        DCF_HIP(hipGetLastError());
      }
      return 0;
    };
    if (gate_first) {
      if (m->vid_w2) for (int v = 0; v < nvid; ++v) TRY(shallow_half(v));
      TRY(flush());
      TRY(gather_masks());
      GateArgs ga{b.correl, nvid > 1 ? b.maskv : vid_mask, b.gate, sv > 1 ? b.mask_pre : b.mask_all, T0, nq, 0, c.sn, c.msf, (double)c.sratio, vmap0,
                  b.tile_flags, nflags};
      TRY(launch_gate(ga, st));
      // the expert halves of ALL videos in one grid where the operands allow it: the ~half of the row tiles that survive the gates
      // then fill the chip once (three launches of three videos each took as long as without the gate: a 64 x 256 tile is bound
      // by the latency of its 32 K steps, not by how many tiles run beside it)
      auto wit = m->wsplit.find(m->vid_w1);
      const bool one_grid = nvid > 1 && nvid <= GEMM_ZMAX && E % 256 == 0 && T0 % 4 == 0 && wit != m->wsplit.end() && m->wsplit_ldw[m->vid_w1] == m->vid_ldw;
      if (one_grid) {
        GemmArgs base = gemm(vs.vid[0], T0, m->vid_w1, nullptr, b.P1, E, T0, E, D);
        base.ldw = m->vid_ldw; base.a_scale = 1.f; base.Ws = wit->second; base.status = m->status; base.skip_stride = nflags;
        const float* zA[GEMM_ZMAX]; float* zC[GEMM_ZMAX]; const uint8_t* zs[GEMM_ZMAX]; int zq[GEMM_ZMAX];
        for (int v = 0; v < nvid; ++v) { zA[v] = vs.vid[v]; zC[v] = b.P1 + (size_t)v * T0 * E; zs[v] = b.tile_flags + (size_t)q_of[v] * nflags; zq[v] = vs.nq[v]; }
        // the profile prices this launch at the clips the gate keeps at least -- int(sratio n) of n blocks -- not at the full product:
        // the tiles it skips are work the reference does (on zeros) and this kernel does not
        TRY(launch_gemm_split_z(base, nvid, zA, zC, zs, zq, m->wsplit_terms[m->vid_w1], st, std::min(1.0, std::max(0.0, (double)c.sratio))));
      } else {
        for (int v = 0; v < nvid; ++v) TRY(expert_half(v, true));
        TRY(flush());
      }
    } else {
      for (int v = 0; v < nvid; ++v) {
        if (m->vid_w1) TRY(expert_half(v, false));
        if (m->vid_w2) TRY(shallow_half(v));
      }
      TRY(flush());
      TRY(gather_masks());
    }
  }
  if (nvid > 1) vid_mask = b.maskv;

  for (int q0 = 0; q0 < nq; q0 += Bmax) {
    const int B = std::min(Bmax, nq - q0);
    Plan* pl;
    TRY(get_plan(m, Tp, B, L, st, &pl));
    const LevelTable& lt = pl->lt;
    const int rows0 = B * T0, rowsP = B * Tp, rowsAll = B * S;
    uint8_t* mask_in = sv > 1 ? b.mask_pre : b.mask_all;      // validity of the T0 input clips (gate stage)
    unsigned long long vmap = 0;           // video of batch element i in nibble i (B <= 16 with several videos, <= 16 videos)
    for (int i = 0; i < B && i < 16; ++i) vmap |= (unsigned long long)video_of[q0 + i] << (4 * i);

    // ---- gate + masks for every level
    if (gate_override) {
      // T-sharded videos: the gate was selected globally (all-gathered scores) by the caller
      hipLaunchKernelGGL(k_apply_gate, dim3((rows0 + 255) / 256), dim3(256), 0, st, gate_override + (int64_t)q0 * T0, vid_mask,
                         b.gate, mask_in, T0, rows0, c.msf);
      DCF_HIP(hipGetLastError());
    } else if (!gate_first) {                      // (gate_first: selected in front of the expert products, above)
      GateArgs ga{b.correl, vid_mask, b.gate, mask_in, T0, B, q0, c.sn, c.msf, (double)c.sratio, vmap, nullptr, 0};
      TRY(launch_gate(ga, st));
    }
    int pre_off = 0;                              // first row of the last input-resolution level (= pyramid level 0) in mask_pre
    if (sv > 1) {
      int rows_pre = 0;
      for (int j = 0; j <= npre; ++j) { if (j == npre) pre_off = rows_pre; rows_pre += B * (T0 >> j); }
      TRY(launch_pyramid_masks(b.mask_pre, b.nbr_pre, B, T0, npre + 1, rows_pre, st));       // mask_j[i] = mask_0[i << j] (blocks.py:101-105)
      DCF_HIP(hipMemcpyAsync(b.mask_all, b.mask_pre + pre_off, (size_t)rowsP, hipMemcpyDeviceToDevice, st));
    }
    TRY(launch_pyramid_masks(b.mask_all, b.nbr_all, B, Tp, L, rowsAll, st));
    const uint8_t* mask0 = mask_in;               // input clips: vid_map, early fusion, embd_fc
    const uint8_t* maskP = b.mask_all;            // pyramid level 0

    // ---- vid_map (model.py:543-555)
    TRY(launch_vidmap_combine(m->vid_w1 ? b.P1 : nullptr, m->vid_w2 ? b.P2 : nullptr, m->vid_map_b, b.gate, mask0,
                              m->vid_w3, m->vid_w3 ? b.correl + (int64_t)q0 * T0 : nullptr, b.X, T0, rows0, E, vmap, st));
    if (m->keep_debug) DCF_CHECK((int64_t)rows0 * E <= m->dbg_cap, "dcf_debug_copy: armed destination holds %lld floats, the tap needs %lld", (long long)m->dbg_cap, (long long)rows0 * E);
    if (m->keep_debug && m->dbg_vidmap) DCF_HIP(hipMemcpyAsync(m->dbg_vidmap, b.X, (size_t)rows0 * E * 4, hipMemcpyDeviceToDevice, st));

    // ---- text side: pointers of this chunk
    // the kernels take the pointers by value (kernel argument): a captured graph bakes them in, which is what its key
    // (every text pointer and length) promises
    TextMeta tm{};
    for (int i = 0; i < B; ++i) {
      tm.text[i] = text[q0 + i];
      tm.text_mask[i] = text_mask ? text_mask[q0 + i] : nullptr;
      tm.len[i] = text_len[q0 + i];
    }
    const TextMeta* dm = &tm;

    // ---- early fusion: XAttNFusion on the level-0 sequence (fusion.py:56-66)
    bool fused_as_stats = false;                  // fusion.ln_out carried into vid_net.embd_fc as row statistics
    if (c.model_kind != 1) {
      const bool tap = m->keep_debug && m->dbg_fused;    // the `fused` debug tap wants the normalised rows themselves
      TRY(run_fusion(m, b, b.X, E, B, T0, nullptr, mask0, nullptr, dm, Lk, b.R[0], E, st, tap ? nullptr : &fused_as_stats));
      if (tap) DCF_HIP(hipMemcpyAsync(m->dbg_fused, b.R[0], (size_t)rows0 * E * 4, hipMemcpyDeviceToDevice, st));
    }

    // ---- vid_net: VideoTransformer.forward (video_net.py:123-164)
    {
      if (c.model_kind != 1 && fused_as_stats) {
        // embd_fc(ln_out(x) * mask) from the raw x in b.X: padded rows come out as finite garbage instead of the bias, and every
        // consumer masks its input rows (blocks.py:98-99).  Not in place: column tiles of other workgroups still read b.X.
        GemmArgs ge = gemm(b.X, E, m->embd_fc_wf, m->embd_fc_c, b.R[3], E, rows0, E, E);
        ge.flags = G_AMASK; ge.rowmask = mask0;
        ge.stats_in = b.stats; ge.ln_s = m->embd_fc_s; ge.stats_slots = E / STATS_W; ge.stats_w = STATS_W;
        TRY(run_gemm(m, &ge, 1, A_ROWS, st));
        std::swap(b.X, b.R[3]);
      } else if (c.model_kind != 1) {
        GemmArgs ge = gemm(b.R[0], E, m->embd_fc_w, m->embd_fc_b, b.X, E, rows0, E, E);
        ge.flags = G_AMASK; ge.rowmask = mask0;
        TRY(run_gemm(m, &ge, 1, A_ROWS, st));
      }   // late fusion (PtTransformer): b.X already IS embd_fc([gate*vid ; shallow] * mask), model.py:132-140
      if (sv > 1) {
        // vid_net.stride > 1 (video_net.py:62-73): while the sequence is longer than the pyramid's, a convolution is k5 / stride 2 /
        // padding 2 on the masked input -- its five taps gathered into rows [B T/2][5E] and one plain GEMM --, then k3 as usual; every
        // one followed by LayerNorm + ReLU, the last one by + pe * mask (video_net.py:136-152)
        int Tc = T0, off = 0;
        for (int i = 0; i < c.n_embd_convs; ++i) {
          const bool with_pe = c.use_abs_pe && i == c.n_embd_convs - 1;
          if (Tc > Tp) {
            TRY(launch_im2col5s2(b.X, E, b.mask_pre + off, b.col5, B, Tc, E, st));
            GemmArgs g = gemm(b.col5, 5 * E, m->embd_conv[i], nullptr, b.R[0], E, B * (Tc / 2), E, 5 * E);
            TRY(run_gemm(m, &g, 1, A_ROWS, st));
            off += B * Tc;
            Tc /= 2;
          } else {
            GemmArgs g = gemm(b.X, E, m->embd_conv[i], nullptr, b.R[0], E, B * Tc, E, 3 * E);
            g.cin = E; g.nbr = b.nbr_pre + off;
            TRY(run_gemm(m, &g, 1, A_ROWS_TAP3, st));
          }
          LnArgs ln{}; ln.X = b.R[0]; ln.ldx = E; ln.Y = b.X; ln.ldy = E; ln.w = m->embd_ln_w[i]; ln.b = m->embd_ln_b[i];
          ln.rows = B * Tc; ln.C = E; ln.relu = 1;
          if (with_pe) { ln.pe = m->pe; ln.mask = b.mask_pre + off; ln.T = Tc; }
          TRY(launch_ln(ln, st));
        }
        DCF_CHECK(Tc == Tp, "vid_net.arch[0]=%d embedding convolutions cannot divide the sequence by vid_net.stride=%d (video_net.py:53)", c.n_embd_convs, sv);
      }
      int epend = -1, epend_w = 0;                  // embedding layer whose LayerNorm + ReLU the next convolution applies on load
      for (int i = 0; i < c.n_embd_convs && sv == 1; ++i) {
        GemmArgs g = gemm(b.X, E, m->embd_conv[i], nullptr, b.R[0], E, rows0, E, 3 * E);
        g.cin = E; g.nbr = b.nbr_all;
        const bool with_pe = c.use_abs_pe && i == c.n_embd_convs - 1;
        int sw = 0;
        if (epend >= 0) {
          // A = the RAW output of the previous convolution (in R[0]); this one writes to R[3] (a level-0 sized scratch that no
          // fusion pass uses) and the LayerNorm reads it from there.  The buffers are NOT swapped: R[0..2] hold rowsF rows (the
          // whole pyramid with late / second fusion), R[3..6] only the level-0 rows.
          g.A = b.R[0]; g.C = b.R[3];
          norm_a(g, b.hstats[0], E, epend_w, m->embd_ln_w[epend], m->embd_ln_b[epend]);
          epend = -1;
          TRY(run_gemm(m, &g, 1, A_ROWS_TAP3, st));
          LnArgs ln{}; ln.X = b.R[3]; ln.ldx = E; ln.Y = b.X; ln.ldy = E; ln.w = m->embd_ln_w[i]; ln.b = m->embd_ln_b[i];
          ln.rows = rows0; ln.C = E; ln.relu = 1;
          if (with_pe) { ln.pe = m->pe; ln.mask = mask0; ln.T = T0; }
          TRY(launch_ln(ln, st));
        } else if (i + 1 < c.n_embd_convs && i + 2 == c.n_embd_convs && can_norm_a(m, m->embd_conv[i], m->embd_conv[i + 1], rows0, E, &sw)) {
          g.stats_out = b.hstats[0]; g.stats_w = sw;                 // raw output (R[0]) + row statistics; no LayerNorm launch
          TRY(run_gemm(m, &g, 1, A_ROWS_TAP3, st));
          epend = i; epend_w = sw;
        } else
        if (can_fuse_ln(m, m->embd_conv[i], rows0, E, 3 * E, A_ROWS_TAP3)) {
          // conv -> LN -> ReLU (+ pe * mask) in one kernel; the result goes to R[3] (same size as X) because the k3
          // taps of other workgroups still read X, then X and R[3] swap roles
          g.C = nullptr; g.ln_w = m->embd_ln_w[i]; g.ln_b = m->embd_ln_b[i]; g.Y = b.R[3]; g.ldy = E; g.ln_relu = 1;
          if (with_pe) { g.ln_pe = m->pe; g.ln_mask = mask0; g.ln_T = T0; }
          TRY(run_gemm(m, &g, 1, A_ROWS_TAP3, st));
          std::swap(b.X, b.R[3]);
        } else {
          TRY(run_gemm(m, &g, 1, A_ROWS_TAP3, st));
          LnArgs ln{}; ln.X = b.R[0]; ln.ldx = E; ln.Y = b.X; ln.ldy = E; ln.w = m->embd_ln_w[i]; ln.b = m->embd_ln_b[i];
          ln.rows = rows0; ln.C = E; ln.relu = 1;
          if (with_pe) { ln.pe = m->pe; ln.mask = mask0; ln.T = T0; }
          TRY(launch_ln(ln, st));
        }
      }
      if (c.use_abs_pe && c.n_embd_convs == 0) {
        LnArgs ln{}; ln.X = b.X; ln.ldx = E; ln.Y = b.X; ln.ldy = E; ln.rows = rows0; ln.C = E; ln.skip_ln = 1;
        ln.pe = m->pe; ln.mask = mask0; ln.T = T0;
        TRY(launch_ln(ln, st));
      }
      for (size_t i = 0; i < m->stem.size(); ++i) {
        // stem layers work in place at level 0: out -> R[3] is free for stride 1, then copy back via swap of roles
        TRY(run_encoder(m, m->stem[i], b, b.X, E, maskP, maskP, B, Tp, 1, b.R[3], E, st));
        DCF_HIP(hipMemcpyAsync(b.X, b.R[3], (size_t)rowsP * E * 4, hipMemcpyDeviceToDevice, st));
      }
      const int ldf = E + TCN_HID;
      const float* xin = b.X;
      int64_t ldx = E;
      for (int l = 0; l < L; ++l) {
        const int stride = l > 0 ? 2 : 1;
        const uint8_t* mi = b.mask_all + lt.start[l > 0 ? l - 1 : 0];
        const uint8_t* mo = b.mask_all + lt.start[l];
        float* xo = b.F + (int64_t)lt.start[l] * ldf;
        if (c.pool_only) TRY(launch_dwconv3(xin, ldx, mi, m->pool_w[l], xo, ldf, B, l > 0 ? lt.T[l - 1] : Tp, stride, E, st));   // video_net.py:107-109
        else TRY(run_encoder(m, m->branch[l], b, xin, ldx, mi, mo, B, l > 0 ? lt.T[l - 1] : Tp, stride, xo, ldf, st));
        xin = xo; ldx = ldf;
      }
    }

    if (m->hyb_levels > 0) {                      // dcf_hybrid_phase1: the pyramid stands up to level k; hand it over and stop
      TRY(hybrid_take(m, b, *pl, B, Lk, st));
      continue;
    }

    // ---- second / late fusion over the whole pyramid (model.py:443-444, :66-67; fusion.py:68-78), in place on F
    if (c.model_kind == 1 || c.second_fusion)
      TRY(run_fusion(m, b, b.F, E + TCN_HID, B, Tp, &lt, b.mask_all, b.nbr_all, dm, Lk, b.F, E + TCN_HID, st));

    if (c.model_kind != 0) {
      // ---- PtTransformer / PtTransformerEarlyFusion.fuse_and_predict (model.py:65-69, :204-209): cls_head / reg_head on the pyramid
      TRY(run_head_pair(m, m->cls1, m->reg, b, *pl, E, 1, 0, logits_out + (int64_t)q0 * S, 2, 1, offsets_out + (int64_t)q0 * S * 2, st));
    } else {
    // ---- heads: fuse_and_predict (model.py:442-471)
    TRY(run_head(m, m->cls1, b, *pl, E, 1, 0, 0, b.logits1, st));
    if (vs.logits1_out) {
      hipLaunchKernelGGL(k_points_out, dim3((rowsAll + 255) / 256), dim3(256), 0, st, (const float*)b.logits1,
                         vs.logits1_out + (int64_t)q0 * S, (const LevelTable*)pl->d_lt);
      DCF_HIP(hipGetLastError());
    }
    {
      RefineArgs ra{};
      ra.logits1 = b.logits1; ra.lt = pl->d_lt; ra.mask_all = b.mask_all;
      ra.w_in = m->tcn_in_w; ra.b_in = m->tcn_in_b;
      ra.host_w_dil = m->tcn_wd.data(); ra.host_b_dil = m->tcn_bd.data(); ra.host_w_pw = m->tcn_wp.data();
      ra.host_b_pw = m->tcn_bp.data(); ra.host_ln_w = m->tcn_lnw.data(); ra.host_ln_b = m->tcn_lnb.data();
      ra.w_out = m->tcn_out_w; ra.b_out = m->tcn_out_b;
      ra.host_frag = (!m->tcn_frag.empty() && debug_option("tcn_frag", 1) != 0) ? m->tcn_frag.data() : nullptr;
      ra.stack_layers = debug_option("tcn_stack", tcn_stack_env());          // (dcf_debug_set_option: 0 = layer by layer)
      ra.bufA = b.tcnA; ra.bufB = b.tcnB; ra.F = b.F; ra.ldf = E + TCN_HID; ra.E = E;
      ra.B = B; ra.T0 = Tp; ra.n_levels = L; ra.n_layers = L;
      ra.f16 = m->gemm_terms == GEMM_F16X3; ra.status = m->status;
      TRY(launch_refine(ra, lt, st));
    }
    TRY(run_head_pair(m, m->cls2, m->reg, b, *pl, E + TCN_HID, 1, 0, logits_out + (int64_t)q0 * S, 2, 1,
                      offsets_out + (int64_t)q0 * S * 2, st));
    }
    hipLaunchKernelGGL(k_masks_out, dim3((rowsAll + 255) / 256), dim3(256), 0, st, (const uint8_t*)b.mask_all,
                       masks_out + (int64_t)q0 * S, (const LevelTable*)pl->d_lt, (const unsigned*)m->status,
                       logits_out + (int64_t)q0 * S);
    DCF_HIP(hipGetLastError());

    m->dbg.correl = b.correl; m->dbg.gate = b.gate; m->dbg.F = b.F;
    m->dbg.nq = nq; m->dbg.T0 = T0; m->dbg.B = B; m->dbg.S = S;
  }
  m->keep_debug = false;                      // the taps of dcf_debug_copy(2 / 3) fill once
  m->dbg_vidmap = m->dbg_fused = nullptr;
  return 0;
}

// TextTransformer.forward (text_net.py:158-188) for one query: embd_fc 1x1 on x * mask, + pe * mask, background
// token, n x TransformerEncoder(stride 0) (blocks.py:578-591 with global MaskedMHA, blocks.py:374-393).
// tokens (C_t, Lq) channel-major -> text_out (TE, Lk) channel-major (the layout dcf_forward_eval takes), Lk = Lq + bkgd.
static int text_encode(dcf_model* m, const float* tokens, const uint8_t* token_mask, int Lq, float* text_out, uint8_t* mask_out,
                       hipStream_t st) {
  const dcf_config& c = m->cfg;
  DCF_CHECK(m->finalized, "dcf_text_encode: model not finalized");
  DCF_CHECK(m->text_embd_w || c.text_kind == 1, "dcf_text_encode: the model was created without a text encoder (text_in / text_layers = 0)");
  DCF_CHECK(Lq >= 1 && tokens && text_out && mask_out, "dcf_text_encode: bad arguments");
  const int TE = c.TE, Lk = Lq + (c.text_bkgd ? 1 : 0);
  if (c.text_abs_pe) DCF_CHECK(m->text_pe && m->text_pe_L >= Lq, "text position encoding for %d tokens not set (dcf_model_set_text_pe)", Lq);
  // workspace: X, X2, R0, Q, K, V [Lk][TE]; HID [Lk][4 TE]
  const size_t rowB = ((size_t)Lk * TE * sizeof(float) + 255) & ~(size_t)255;
  const size_t need = 10 * rowB;
  if (need > m->text_ws_bytes) {
    DCF_HIP(hipStreamSynchronize(st));
    if (m->text_ws) DCF_HIP(hipFree(m->text_ws));
    m->text_ws = nullptr; m->text_ws_bytes = 0;
    DCF_HIP(hipMalloc(&m->text_ws, need));
    m->text_ws_bytes = need;
  }
  auto buf = [&](int i) { return reinterpret_cast<float*>(m->text_ws + (size_t)i * rowB); };
  float *X = buf(0), *X2 = buf(1), *R0 = buf(2), *Q = buf(3), *K = buf(4), *V = buf(5), *HID = buf(6);   // HID spans 4 slots

  TextEmbedArgs te{tokens, token_mask, m->text_embd_w, m->text_embd_b, c.text_abs_pe ? m->text_pe : nullptr, m->text_bkgd, X, mask_out,
                   c.text_in, Lq, TE, (c.text_kind == 1 && c.text_bkgd) ? 1 : 0};
  TRY(launch_text_embed(te, st));
  if (c.text_kind == 1) {
    if (c.text_bkgd) {
      // AttNPool1D (blocks.py:396-411): h = [masked mean ; x], pooled token = MaskedMHA(h, kv_mask)[..., :1], out = [pool ; x]
      const TextEncW& w = m->text_pool;
      GemmArgs g3[3] = {gemm(X, TE, w.wq, w.bq, Q, TE, Lk, TE, TE), gemm(X, TE, w.wk, w.bk, K, TE, Lk, TE, TE),
                        gemm(X, TE, w.wv, w.bv, V, TE, Lk, TE, TE)};
      TRY(run_gemm(m, g3, 3, A_ROWS, st));
      XAttnArgs xa{Q, K, V, mask_out, R0, 1, Lk, Lk, TE, c.text_heads, m->status};
      TRY(launch_xattn(xa, st));
      GemmArgs gp = gemm(R0, TE, w.wp, w.bp, X, TE, 1, TE, TE);                       // only the pooled row is kept
      TRY(run_gemm(m, &gp, 1, A_ROWS, st));
    }
    TRY(launch_rows_to_chanmajor(X, text_out, Lk, TE, st));
    return 0;
  }
  for (const TextEncW& w : m->text_enc) {
    TRY(launch_mask_rows(X, mask_out, Lk, TE, st));                                   // x = x * mask   (blocks.py:581)
    LnArgs ln{}; ln.X = X; ln.ldx = TE; ln.Y = R0; ln.ldy = TE; ln.w = w.ln_attn_w; ln.b = w.ln_attn_b; ln.rows = Lk; ln.C = TE;
    TRY(launch_ln(ln, st));
    GemmArgs g3[3] = {gemm(R0, TE, w.wq, w.bq, Q, TE, Lk, TE, TE), gemm(R0, TE, w.wk, w.bk, K, TE, Lk, TE, TE),
                      gemm(R0, TE, w.wv, w.bv, V, TE, Lk, TE, TE)};
    TRY(run_gemm(m, g3, 3, A_ROWS, st));
    XAttnArgs xa{Q, K, V, mask_out, R0, 1, Lk, Lk, TE, c.text_heads, m->status};                   // softmax over the valid tokens
    TRY(launch_xattn(xa, st));
    GemmArgs gp = gemm(R0, TE, w.wp, w.bp, X2, TE, Lk, TE, TE);                        // x = skip * mask + ls * proj(ctx)
    gp.flags = G_RES; gp.R = X; gp.ldr = TE; gp.ls = w.ls_attn;
    TRY(run_gemm(m, &gp, 1, A_ROWS, st));
    LnArgs l2{}; l2.X = X2; l2.ldx = TE; l2.Y = R0; l2.ldy = TE; l2.w = w.ln_ffn_w; l2.b = w.ln_ffn_b; l2.rows = Lk; l2.C = TE;
    TRY(launch_ln(l2, st));
    GemmArgs gf = gemm(R0, TE, w.fc_w, w.fc_b, HID, 4 * TE, Lk, 4 * TE, TE);
    gf.flags = G_GELU;
    TRY(run_gemm(m, &gf, 1, A_ROWS, st));
    GemmArgs go = gemm(HID, 4 * TE, w.pj_w, w.pj_b, X, TE, Lk, TE, 4 * TE);             // x += ls * (ffn * mask)
    go.flags = G_RES | G_OUT_MASK; go.rowmask = mask_out; go.R = X2; go.ldr = TE; go.ls = w.ls_ffn;
    TRY(run_gemm(m, &go, 1, A_ROWS, st));
  }
  TRY(launch_rows_to_chanmajor(X, text_out, Lk, TE, st));
  return 0;
}

}  // namespace dcf

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

const char* dcf_last_error(void) { return dcf::g_err.c_str(); }
int dcf_abi_version(void) { return 10; }

int dcf_model_create(const dcf_config* cfg, dcf_model** out) {
  DCF_CHECK(cfg && out, "dcf_model_create: null argument");
  DCF_CHECK(cfg->E > 0 && cfg->E % 32 == 0 && cfg->E <= 992, "E=%d must be a positive multiple of 32 (<= 992)", cfg->E);
  DCF_CHECK(cfg->attn_mode == 0 || cfg->attn_mode == 1, "attn_mode=%d: 0 (f16x3) or 1 (one fp16 product)", cfg->attn_mode);
  DCF_CHECK(cfg->D > 0 && cfg->D % 32 == 0, "D=%d must be a positive multiple of 32", cfg->D);
  DCF_CHECK(cfg->TE > 0 && cfg->TE % 32 == 0, "TE=%d must be a positive multiple of 32", cfg->TE);
  DCF_CHECK(cfg->n_levels >= 1 && cfg->n_levels <= DCF_MAX_LEVELS, "n_levels=%d out of range", cfg->n_levels);
  DCF_CHECK(cfg->win == 0 || (cfg->win > 0 && (cfg->win & 1)), "mha_win_size=%d must be odd, or 0 for global self-attention over the clips", cfg->win);
  DCF_CHECK(cfg->fusion_layers >= 0 && cfg->head_layers >= 0 && cfg->n_embd_convs >= 0 && cfg->n_stem >= 0, "negative layer count");
  DCF_CHECK(cfg->sn >= 1, "sn must be >= 1");
  DCF_CHECK(cfg->model_kind >= 0 && cfg->model_kind <= 2, "model_kind must be 0 (iterative early fusion), 1 (late fusion) or 2 (early fusion)");
  {
    const int sv = cfg->vid_stride > 1 ? cfg->vid_stride : 1;
    int lg = 0;
    while ((1 << lg) < sv) ++lg;
    DCF_CHECK((sv & (sv - 1)) == 0 && cfg->n_embd_convs >= lg, "vid_net.stride=%d must be a power of two with arch[0]=%d >= log2(stride) (video_net.py:52-53)",
              sv, cfg->n_embd_convs);
  }
  int ndev = 0;
  DCF_HIP(hipGetDeviceCount(&ndev));
  DCF_CHECK(ndev > 0, "no HIP device");
  dcf_model* m = new dcf_model();
  m->cfg = *cfg;
  *out = m;
  return 0;
}

void dcf_model_destroy(dcf_model* m) {
  if (!m) return;
  dcf::free_model(m);
  delete m;
}

int dcf_model_bind(dcf_model* m, const char* name, const float* data, const int64_t* shape, int32_t ndim) {
  DCF_CHECK(m && name && data, "dcf_model_bind: null argument");
  dcf::Bound b;
  b.p = data;
  for (int i = 0; i < ndim; ++i) b.shape.push_back(shape[i]);
  m->bound[name] = b;
  m->finalized = false;
  return 0;
}

namespace dcf {
// eager on the first call with a given argument set, capture + replay from the second identical call on
static int forward_graph_on(dcf_model* m, const VideoSet& vs, int T0, int nq,
                            const float* const* text, const uint8_t* const* text_mask, const int32_t* text_len,
                            const float* gate, float* lo, float* oo, uint8_t* mo, hipStream_t st) {
  std::vector<uint64_t> key = {(uint64_t)vs.nvid, (uint64_t)T0, (uint64_t)nq, (uint64_t)vs.logits1_out,
                               (uint64_t)gate, (uint64_t)lo, (uint64_t)oo, (uint64_t)mo, (uint64_t)st, (uint64_t)m->pe, (uint64_t)m->pe_T};
  for (int v = 0; v < vs.nvid; ++v) {
    key.push_back((uint64_t)vs.vid[v]); key.push_back((uint64_t)vs.shallow[v]); key.push_back((uint64_t)vs.mask[v]);
    key.push_back((uint64_t)vs.text_cls[v]); key.push_back((uint64_t)vs.nq[v]);
  }
  for (int q = 0; q < nq; ++q) {
    key.push_back((uint64_t)text[q]);
    key.push_back(text_mask ? (uint64_t)text_mask[q] : 0);
    key.push_back((uint64_t)text_len[q]);
  }
  m->last_launch = 0;
  if (m->graph_exec && key == m->graph_key) {
    DCF_HIP(hipGraphLaunch(m->graph_exec, st));
    m->last_launch = 1;
    return 0;
  }
  if (key != m->last_key || key == m->nocapture_key) {   // first sighting (allocates workspace / plans), or known not to capture
    m->last_key = key;
    return forward(m, vs, T0, nq, text, text_mask, text_len, gate, lo, oo, mo, st);
  }
  // second identical call: capture
  drop_graph(m);
  m->last_key = key;
  if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) != hipSuccess) {
    (void)hipGetLastError();
    m->nocapture_key = key;
    return forward(m, vs, T0, nq, text, text_mask, text_len, gate, lo, oo, mo, st);
  }
  m->capturing = true;
  const int rc = forward(m, vs, T0, nq, text, text_mask, text_len, gate, lo, oo, mo, st);
  m->capturing = false;
  hipGraph_t g = nullptr;
  const hipError_t ec = hipStreamEndCapture(st, &g);
  if (rc != 0 || ec != hipSuccess || !g) {           // capture failed: nothing ran; fall back to an eager forward
    (void)hipGetLastError();
    if (g) (void)hipGraphDestroy(g);
    m->nocapture_key = key;
    const std::string err = g_err;
    const int rc2 = forward(m, vs, T0, nq, text, text_mask, text_len, gate, lo, oo, mo, st);
    if (rc2 != 0 && !err.empty()) g_err = err;
    return rc2;
  }
  hipGraphExec_t ge = nullptr;
  if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) {
    (void)hipGetLastError();
    (void)hipGraphDestroy(g);
    m->nocapture_key = key;
    return forward(m, vs, T0, nq, text, text_mask, text_len, gate, lo, oo, mo, st);
  }
  m->graph = g;
  m->graph_exec = ge;
  m->graph_key = key;
  DCF_HIP(hipGraphLaunch(ge, st));
  m->last_launch = 2;
  return 0;
}

static int forward_maybe_graph(dcf_model* m, const VideoSet& vs, int T0, int nq,
                               const float* const* text, const uint8_t* const* text_mask, const int32_t* text_len,
                               const float* gate, float* lo, float* oo, uint8_t* mo, hipStream_t st) {
  static const bool no_graph = getenv("DCF_NO_GRAPH") != nullptr;
  if (m->hyb) m->hyb->valid = false;              // whatever way this forward is issued (a graph replay does not pass through forward())
  if (m->option_epoch != g_option_epoch.load()) { drop_graph(m); m->option_epoch = g_option_epoch.load(); }
  // Auto: replay a graph only for forwards of >= 64 K level-0 rows.  Measured on MI355X (profiles/r03_notes.md): the batched
  // forward runs at the same speed either way (25.36 vs 25.35 ms per 24-video step) and the graph shields it from host
  // jitter; ONE video per call (~100 launches of 5 - 40 us) is 5 % faster launched eagerly (1.75 vs 1.84 ms: a graph node
  // costs ~0.9 us more than an in-order launch, and the host needs ~0.5 ms to issue the forward the GPU takes 1.75 ms for).
  const bool want = m->graph_mode == 1 || (m->graph_mode == 0 && (long long)nq * T0 >= 65536);
  const bool eligible = want && !no_graph && !g_prof_on && !m->keep_debug && nq > 0;
  if (!eligible) {
    m->last_launch = 0;
    return forward(m, vs, T0, nq, text, text_mask, text_len, gate, lo, oo, mo, st);
  }
  if (st != nullptr) return forward_graph_on(m, vs, T0, nq, text, text_mask, text_len, gate, lo, oo, mo, st);
  // legacy default stream: hop to the engine's own stream (see dcf_model::own)
  if (!m->own) {
    DCF_HIP(hipStreamCreateWithFlags(&m->own, hipStreamNonBlocking));
    DCF_HIP(hipEventCreateWithFlags(&m->ev_in, hipEventDisableTiming));
    DCF_HIP(hipEventCreateWithFlags(&m->ev_out, hipEventDisableTiming));
  }
  DCF_HIP(hipEventRecord(m->ev_in, st));
  DCF_HIP(hipStreamWaitEvent(m->own, m->ev_in, 0));
  const int rc = forward_graph_on(m, vs, T0, nq, text, text_mask, text_len, gate, lo, oo, mo, m->own);
  DCF_HIP(hipEventRecord(m->ev_out, m->own));
  DCF_HIP(hipStreamWaitEvent(st, m->ev_out, 0));
  return rc;
}
}  // namespace dcf

int dcf_model_set_pe(dcf_model* m, const float* pe_tokens, int64_t T) {
  DCF_CHECK(m, "dcf_model_set_pe: null model");
  m->pe = pe_tokens;
  m->pe_T = T;
  return 0;
}

int dcf_model_set_text_pe(dcf_model* m, const float* pe_tokens, int64_t L) {
  DCF_CHECK(m, "dcf_model_set_text_pe: null model");
  m->text_pe = pe_tokens;
  m->text_pe_L = L;
  return 0;
}

int dcf_text_encode(dcf_model* m, const float* tokens, const uint8_t* token_mask, int32_t Lq, float* text_out, uint8_t* mask_out,
                    void* stream) {
  DCF_CHECK(m, "dcf_text_encode: null model");
  return dcf::text_encode(m, tokens, token_mask, Lq, text_out, mask_out, (hipStream_t)stream);
}

int dcf_model_finalize(dcf_model* m, void* stream) {
  DCF_CHECK(m, "dcf_model_finalize: null model");
  return dcf::finalize(m, (hipStream_t)stream);
}

int dcf_numerics_status(dcf_model* m, int32_t reset, void* stream) {
  DCF_CHECK(m, "dcf_numerics_status: null model");
  int out = m->gemm_terms == dcf::GEMM_BF16X6 ? 4 : (m->gemm_terms == 0 ? 8 : 0);
  if (m->force_x6) out |= 2;
  if (m->status) {
    unsigned flag = 0u;
    hipStream_t st = (hipStream_t)stream;
    DCF_HIP(hipMemcpyAsync(&flag, m->status, sizeof(flag), hipMemcpyDeviceToHost, st));
    if (reset) DCF_HIP(hipMemsetAsync(m->status, 0, sizeof(unsigned), st));
    DCF_HIP(hipStreamSynchronize(st));
    if (flag & ~2u) out |= 1;
    if (flag & 2u) out |= 16;                  // one-pass LayerNorm statistics met an ill-conditioned row (common.h LN_ILL_RATIO)
  }
  return out;
}

int dcf_model_set_ln_carry(dcf_model* m, int32_t on) {
  DCF_CHECK(m, "dcf_model_set_ln_carry: null model");
  const bool off = on == 0;
  if (off != m->no_ln_carry) dcf::drop_graph(m);
  m->no_ln_carry = off;
  return 0;
}

int dcf_numerics_status_async(dcf_model* m, int32_t* host_dst, void* stream) {
  DCF_CHECK(m && host_dst, "dcf_numerics_status_async: null argument");
  if (!m->status) { *host_dst = 0; return 0; }
  DCF_HIP(hipMemcpyAsync(host_dst, m->status, sizeof(unsigned), hipMemcpyDeviceToHost, (hipStream_t)stream));
  return 0;
}

int64_t dcf_points_per_query(const dcf_model* m, int64_t T) {
  int64_t s = 0;
  const int64_t Tp = T / dcf::vid_stride_of(m->cfg);     // the pyramid starts behind the strided embedding convolutions
  for (int l = 0; l < m->cfg.n_levels; ++l) s += Tp >> l;
  return s;
}

int dcf_forward_eval(dcf_model* m, const float* vid, const float* shallow_vid, const uint8_t* vid_mask, int64_t T,
                     int32_t nq, const float* const* text, const uint8_t* const* text_mask, const int32_t* text_len,
                     const float* text_cls, float* logits_out, float* offsets_out, uint8_t* masks_out, void* stream) {
  DCF_CHECK(m && vid && shallow_vid && vid_mask && text && text_len && text_cls && logits_out && offsets_out && masks_out,
            "dcf_forward_eval: null argument");
  DCF_CHECK(T < (1ll << 24), "T too large");
  dcf::VideoSet vs;
  vs.nvid = 1; vs.vid[0] = vid; vs.shallow[0] = shallow_vid; vs.mask[0] = vid_mask; vs.text_cls[0] = text_cls; vs.nq[0] = nq;
  return dcf::forward_maybe_graph(m, vs, (int)T, nq, text, text_mask, text_len, nullptr,
                                  logits_out, offsets_out, masks_out, (hipStream_t)stream);
}

int dcf_forward_eval_videos(dcf_model* m, int32_t nvid, const float* const* vid, const float* const* shallow_vid,
                            const uint8_t* const* vid_mask, int64_t T, const int32_t* nq_per_video, const float* const* text,
                            const uint8_t* const* text_mask, const int32_t* text_len, const float* const* text_cls,
                            float* logits_out, float* offsets_out, uint8_t* masks_out, void* stream) {
  DCF_CHECK(m && vid && shallow_vid && vid_mask && nq_per_video && text && text_len && text_cls && logits_out && offsets_out && masks_out,
            "dcf_forward_eval_videos: null argument");
  DCF_CHECK(nvid >= 1 && nvid <= dcf::DCF_MAX_VIDEOS, "dcf_forward_eval_videos: 1 .. %d videos per call", dcf::DCF_MAX_VIDEOS);
  DCF_CHECK(T < (1ll << 24), "T too large");
  dcf::VideoSet vs;
  vs.nvid = nvid;
  int nq = 0;
  for (int v = 0; v < nvid; ++v) {
    DCF_CHECK(vid[v] && shallow_vid[v] && vid_mask[v] && text_cls[v] && nq_per_video[v] >= 1, "dcf_forward_eval_videos: video %d is incomplete", v);
    vs.vid[v] = vid[v]; vs.shallow[v] = shallow_vid[v]; vs.mask[v] = vid_mask[v]; vs.text_cls[v] = text_cls[v]; vs.nq[v] = nq_per_video[v];
    nq += nq_per_video[v];
  }
  return dcf::forward_maybe_graph(m, vs, (int)T, nq, text, text_mask, text_len, nullptr, logits_out, offsets_out, masks_out,
                                  (hipStream_t)stream);
}

int dcf_forward_train_videos(dcf_model* m, int32_t nvid, const float* const* vid, const float* const* shallow_vid,
                             const uint8_t* const* vid_mask, int64_t T, const int32_t* nq_per_video, const float* const* text,
                             const uint8_t* const* text_mask, const int32_t* text_len, const float* const* text_cls,
                             float* logits1_out, float* logits2_out, float* offsets_out, uint8_t* masks_out, void* stream) {
  DCF_CHECK(m && vid && shallow_vid && vid_mask && nq_per_video && text && text_len && text_cls && logits1_out && logits2_out && offsets_out && masks_out,
            "dcf_forward_train_videos: null argument");
  DCF_CHECK(m->cfg.model_kind == 0, "dcf_forward_train_videos: the iterative early-fusion model only (model.py:567-632)");
  DCF_CHECK(nvid >= 1 && nvid <= dcf::DCF_MAX_VIDEOS, "dcf_forward_train_videos: 1 .. %d videos per call", dcf::DCF_MAX_VIDEOS);
  DCF_CHECK(T < (1ll << 24), "T too large");
  dcf::VideoSet vs;
  vs.nvid = nvid;
  vs.logits1_out = logits1_out;
  int nq = 0;
  for (int v = 0; v < nvid; ++v) {
    DCF_CHECK(vid[v] && shallow_vid[v] && vid_mask[v] && text_cls[v] && nq_per_video[v] >= 1, "dcf_forward_train_videos: video %d is incomplete", v);
    vs.vid[v] = vid[v]; vs.shallow[v] = shallow_vid[v]; vs.mask[v] = vid_mask[v]; vs.text_cls[v] = text_cls[v]; vs.nq[v] = nq_per_video[v];
    nq += nq_per_video[v];
  }
  return dcf::forward_maybe_graph(m, vs, (int)T, nq, text, text_mask, text_len, nullptr, logits2_out, offsets_out, masks_out,
                                  (hipStream_t)stream);
}

int dcf_forward_eval_gated(dcf_model* m, const float* vid, const float* shallow_vid, const uint8_t* vid_mask, int64_t T,
                           int32_t nq, const float* const* text, const uint8_t* const* text_mask, const int32_t* text_len,
                           const float* gate, float* logits_out, float* offsets_out, uint8_t* masks_out, void* stream) {
  DCF_CHECK(m && vid && shallow_vid && vid_mask && text && text_len && gate && logits_out && offsets_out && masks_out,
            "dcf_forward_eval_gated: null argument");
  DCF_CHECK(T < (1ll << 24), "T too large");
  dcf::VideoSet vs;
  vs.nvid = 1; vs.vid[0] = vid; vs.shallow[0] = shallow_vid; vs.mask[0] = vid_mask; vs.text_cls[0] = nullptr; vs.nq[0] = nq;
  return dcf::forward_maybe_graph(m, vs, (int)T, nq, text, text_mask, text_len, gate,
                                  logits_out, offsets_out, masks_out, (hipStream_t)stream);
}

// ---- one long video cut at pyramid level k (see HybridState): three phases, an exchange between each two (dist.py hybrid_forward)
int dcf_hybrid_phase1(dcf_model* m, int32_t k, const float* vid_w, const float* shallow_w, const uint8_t* mask_w, int64_t Tn, int64_t Tc,
                      int32_t nq, const float* const* text, const uint8_t* const* text_mask, const int32_t* text_len, const float* gate_w,
                      float* featk_out, void* stream) {
  DCF_CHECK(m && vid_w && shallow_w && mask_w && text && text_len && gate_w && featk_out, "dcf_hybrid_phase1: null argument");
  const dcf_config& c = m->cfg;
  DCF_CHECK(m->finalized, "dcf_hybrid_phase1: model not finalized");
  DCF_CHECK(c.model_kind == 0 && !c.second_fusion && c.msf && !c.scat && dcf::vid_stride_of(c) == 1,
            "dcf_hybrid_phase1: the iterative early-fusion model with msf, without second_fusion / scat / vid_net.stride > 1");
  const int L = c.n_levels, LC = L - k, half = c.win / 2 > 0 ? c.win / 2 : 1;
  DCF_CHECK(k >= 0 && k < L, "dcf_hybrid_phase1: split level %d outside 0 .. %d", k, L - 1);
  DCF_CHECK(Tn > 0 && Tn < (1ll << 24) && Tn % ((int64_t)half << k) == 0, "dcf_hybrid_phase1: the narrow window (%lld clips) must be a multiple of %d", (long long)Tn, half << k);
  DCF_CHECK(Tc > 0 && Tc < (1ll << 24) && Tc % ((int64_t)half << (LC - 1)) == 0, "dcf_hybrid_phase1: the coarse window (%lld level-%d rows) must be a multiple of %d", (long long)Tc, k, half << (LC - 1));
  const int Bmax = c.max_batch > 0 ? c.max_batch : 8;
  DCF_CHECK(nq >= 1 && nq <= Bmax, "dcf_hybrid_phase1: 1 .. max_batch = %d queries per call", Bmax);
  DCF_CHECK(Tc <= Tn, "dcf_hybrid_phase1: the coarse window (%lld level-%d rows) must not exceed the narrow one (%lld clips): the coarse levels run in its scratch", (long long)Tc, k, (long long)Tn);
  hipStream_t st = (hipStream_t)stream;
  if (!m->hyb) m->hyb = new dcf::HybridState();
  dcf::HybridState& h = *m->hyb;
  h.valid = false; h.k = k; h.Tn = (int)Tn; h.Tc = (int)Tc;
  int Tl[DCF_MAX_LEVELS];
  for (int j = 0; j < LC; ++j) Tl[j] = (int)(Tc >> j);
  // (the level tables of the coarse pyramid are rebuilt -- a synchronising upload -- only when its geometry changes)
  const bool same = h.pc.d_lt && h.pc.L == LC && h.pc.T0 == (int)Tc && h.pc.B == nq && (LC == 1 || (h.pch.d_lt && h.pch.L == LC - 1));
  if (!same) {
    if (dcf::make_plan(m, h.pc, Tl, LC, nq, m->reg_scales.data() + k, st)) return -1;
    if (LC > 1 && dcf::make_plan(m, h.pch, Tl + 1, LC - 1, nq, m->reg_scales.data() + k + 1, st)) return -1;
  }
  dcf::VideoSet vs;
  vs.nvid = 1; vs.vid[0] = vid_w; vs.shallow[0] = shallow_w; vs.mask[0] = mask_w; vs.text_cls[0] = nullptr; vs.nq[0] = nq;
  m->hyb_levels = k + 1;
  m->hyb_extra = dcf::hybrid_extra_bytes(c, nq, (int)Tc, LC, (int)Tn, L);
  m->hyb_feat_out = featk_out;
  const int rc = dcf::forward(m, vs, (int)Tn, nq, text, text_mask, text_len, gate_w, nullptr, nullptr, nullptr, st);
  m->hyb_levels = 0;
  m->hyb_extra = 0;
  m->hyb_feat_out = nullptr;
  if (rc == 0) DCF_CHECK(h.valid, "internal: hybrid phase 1 did not reach the hand-over");
  return rc;
}

int dcf_hybrid_phase2(dcf_model* m, const float* featk_c, const uint8_t* maskk_c, int64_t off_k, float* refk_out, void* stream) {
  DCF_CHECK(m && featk_c && maskk_c && refk_out, "dcf_hybrid_phase2: null argument");
  DCF_CHECK(m->hyb && m->hyb->valid, "dcf_hybrid_phase2: no phase 1 on this model (or another forward ran since)");
  DCF_CHECK(off_k >= 0 && off_k + (m->hyb->Tn >> m->hyb->k) <= m->hyb->Tc, "dcf_hybrid_phase2: the narrow window must lie inside the coarse one");
  return dcf::hybrid_phase2(m, featk_c, maskk_c, (int)off_k, refk_out, (hipStream_t)stream);
}

int dcf_hybrid_phase3(dcf_model* m, const float* refk_c, float* logits_n, float* offsets_n, uint8_t* masks_n, float* logits_c,
                      float* offsets_c, uint8_t* masks_c, void* stream) {
  DCF_CHECK(m && refk_c && logits_n && offsets_n && masks_n, "dcf_hybrid_phase3: null argument");
  DCF_CHECK(m->hyb && m->hyb->valid, "dcf_hybrid_phase3: no phase 1 / 2 on this model (or another forward ran since)");
  DCF_CHECK(m->cfg.n_levels - m->hyb->k <= 1 || (logits_c && offsets_c && masks_c), "dcf_hybrid_phase3: null coarse outputs");
  const int rc = dcf::hybrid_phase3(m, refk_c, logits_n, offsets_n, masks_n, logits_c, offsets_c, masks_c, (hipStream_t)stream);
  m->hyb->valid = false;
  return rc;
}

int dcf_graph_active(const dcf_model* m) { return m ? m->last_launch : 0; }

int dcf_debug_set_option(const char* name, int32_t value) {
  DCF_CHECK(name && *name, "dcf_debug_set_option: empty name");
  static const char* known[] = {"dec_chain_min_rows", "enc_chain_min_rows", "enc_attn_min_rows", "fuse_scores", "tcn_frag", "gate_skip", "tcn_stack"};
  bool ok = false;
  for (const char* k : known) ok = ok || strcmp(k, name) == 0;
  DCF_CHECK(ok, "dcf_debug_set_option: unknown option '%s'", name);
  {
    std::lock_guard<std::mutex> lock(dcf::debug_options_mutex());
    if (value < 0) dcf::debug_options().erase(name);           // back to the built-in value
    else dcf::debug_options()[name] = value;
  }
  dcf::g_option_epoch.fetch_add(1);                         // every model drops its captured graphs at its next forward
  return 0;
}

int dcf_model_set_graph_mode(dcf_model* m, int32_t mode) {
  DCF_CHECK(m && mode >= 0 && mode <= 2, "dcf_model_set_graph_mode: mode must be 0 (auto), 1 (always) or 2 (never)");
  if (mode != m->graph_mode) dcf::drop_graph(m);
  m->graph_mode = mode;
  return 0;
}

int dcf_debug_copy(dcf_model* m, int32_t what, float* dst, int64_t max_floats, void* stream) {
  DCF_CHECK(m && dst, "dcf_debug_copy: null argument");
  hipStream_t st = (hipStream_t)stream;
  const float* src = nullptr;
  int64_t n = 0;
  const int E = m->cfg.E;
  switch (what) {
    case 0: src = m->dbg.correl; n = (int64_t)m->dbg.nq * m->dbg.T0; break;
    case 1: src = m->dbg.gate; n = (int64_t)m->dbg.B * m->dbg.T0; break;
    case 4: src = m->dbg.F; n = (int64_t)m->dbg.B * m->dbg.S * (E + dcf::TCN_HID); break;
    case 2: case 3: {
      // these buffers are overwritten during the forward: arm the tap, the NEXT forward fills dst (rows0 * E floats of its
      // last query chunk) and disarms it again
      m->keep_debug = true;
      m->dbg_cap = max_floats;
      if (what == 2) m->dbg_vidmap = dst; else m->dbg_fused = dst;
      return 0;
    }
    default: DCF_CHECK(false, "dcf_debug_copy: unknown selector %d", what);
  }
  DCF_CHECK(src, "dcf_debug_copy: no forward has run yet");
  DCF_CHECK(n <= max_floats, "dcf_debug_copy: destination too small (%lld > %lld)", (long long)n, (long long)max_floats);
  DCF_HIP(hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, st));
  return 0;
}

// ---- profiling ---------------------------------------------------------------------------------
int dcf_profile_enable(int32_t on) {
  for (auto& r : dcf::g_recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  dcf::g_recs.clear();
  dcf::g_prof_on = on != 0;
  return 0;
}

int64_t dcf_profile_report(char* buf, int64_t cap) {
  struct Agg { long count = 0; double ms = 0, flops = 0, bytes = 0; };
  std::vector<std::pair<std::string, Agg>> aggs;
  for (auto& r : dcf::g_recs) {
    if (hipEventSynchronize(r.b) != hipSuccess) continue;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
    Agg* a = nullptr;
    for (auto& kv : aggs) if (kv.first == r.name) a = &kv.second;
    if (!a) { aggs.emplace_back(r.name, Agg()); a = &aggs.back().second; }
    a->count++; a->ms += ms; a->flops += r.flops; a->bytes += r.bytes;
  }
  std::string out = "{";
  for (size_t i = 0; i < aggs.size(); ++i) {
    char line[512];
    snprintf(line, sizeof(line), "%s\"%s\": {\"count\": %ld, \"ms\": %.6f, \"flops\": %.6e, \"bytes\": %.6e}", i ? ", " : "",
             aggs[i].first.c_str(), aggs[i].second.count, aggs[i].second.ms, aggs[i].second.flops, aggs[i].second.bytes);
    out += line;
  }
  out += "}";
  if (buf && cap > 0) {
    size_t n = std::min((size_t)cap - 1, out.size());
    memcpy(buf, out.data(), n);
    buf[n] = 0;
  }
  return (int64_t)out.size() + 1;
}

// ---- post-processing -------------------------------------------------------------------------
int dcf_collect_segments(const float* logits, const float* offsets, const uint8_t* masks, int32_t nq, int64_t T,
                         int32_t n_levels, float pre_nms_thresh, int32_t pre_nms_topk, float seg_len_thresh,
                         float* segs_out, float* scores_out, int32_t* counts_out, void* stream) {
  return dcf_collect_segments_ext(logits, offsets, masks, nullptr, nq, T, n_levels, pre_nms_thresh, pre_nms_topk, seg_len_thresh,
                                  segs_out, scores_out, counts_out, stream);
}

int dcf_collect_segments_ext(const float* logits, const float* offsets, const uint8_t* masks, const float* ext_scores,
                             int32_t nq, int64_t T, int32_t n_levels, float pre_nms_thresh, int32_t pre_nms_topk,
                             float seg_len_thresh, float* segs_out, float* scores_out, int32_t* counts_out, void* stream) {
  DCF_CHECK(logits && offsets && masks && segs_out && scores_out && counts_out, "dcf_collect_segments: null argument");
  DCF_CHECK(n_levels >= 1 && n_levels <= 16, "dcf_collect_segments: n_levels out of range");
  dcf::CollectArgs a{};
  a.logits = logits; a.offsets = offsets; a.masks = masks;
  a.ext = ext_scores; a.T = (int)T;
  int acc = 0;
  for (int l = 0; l < n_levels; ++l) { a.off[l] = acc; acc += (int)(T >> l); }
  a.off[n_levels] = acc;
  a.S = acc; a.n_levels = n_levels;
  a.pre_nms_thresh = pre_nms_thresh; a.seg_len_thresh = seg_len_thresh; a.pre_nms_topk = pre_nms_topk;
  a.segs = segs_out; a.scores = scores_out; a.counts = counts_out;
  uint32_t* keys = nullptr;
  if (dcf::collect_needs_scratch(acc)) DCF_HIP(hipMallocAsync((void**)&keys, (size_t)nq * acc * sizeof(uint32_t), (hipStream_t)stream));
  a.keys = keys;
  int rc = dcf::launch_collect(a, nq, (hipStream_t)stream);
  if (keys) DCF_HIP(hipFreeAsync(keys, (hipStream_t)stream));
  return rc;
}

int dcf_nms_1d(const float* segs, const float* scores, const int32_t* counts, int32_t nq, int32_t n_max,
               int32_t stride, float iou_thresh, int64_t* keep_out, int32_t* keep_counts_out, void* stream) {
  DCF_CHECK(keep_out && keep_counts_out && (n_max == 0 || (segs && scores)), "dcf_nms_1d: null argument");
  dcf::NmsArgs a{segs, scores, counts, n_max, stride, iou_thresh, (long long*)keep_out, keep_counts_out};
  return dcf::launch_nms(a, nq, (hipStream_t)stream);
}

int dcf_softnms_1d(const float* segs, const float* scores, const int32_t* counts, int32_t nq, int32_t n_max,
                   int32_t stride, float iou_thresh, float sigma, float min_score, int32_t method,
                   int32_t max_iters, float* dets_out, int64_t* inds_out, int32_t* out_counts, void* stream) {
  DCF_CHECK(dets_out && inds_out && out_counts && (n_max == 0 || (segs && scores)), "dcf_softnms_1d: null argument");
  dcf::SoftNmsArgs a{segs, scores, counts, n_max, stride, iou_thresh, sigma, min_score, method, max_iters, dets_out,
                     (long long*)inds_out, out_counts};
  return dcf::launch_softnms(a, nq, (hipStream_t)stream);
}

int dcf_segment_voting(const float* nms_segs, int32_t nms_ld, const int32_t* n1_counts, int32_t n1_max,
                       int32_t n1_stride, const float* all_segs, const float* all_scores,
                       const int32_t* n2_counts, int32_t n2_max, int32_t n2_stride, float iou_thresh,
                       int32_t nq, float* out, void* stream) {
  DCF_CHECK(nms_segs && all_segs && all_scores && out, "dcf_segment_voting: null argument");
  dcf::VotingArgs a{nms_segs, nms_ld, n1_counts, n1_max, n1_stride, all_segs, all_scores, n2_counts, n2_max, n2_stride,
                    iou_thresh, out};
  return dcf::launch_voting(a, nq, (hipStream_t)stream);
}

// ---- single operators ---------------------------------------------------------------------------
int dcf_op_linear(const float* A, const float* W, const float* bias, float* C, int32_t M, int32_t N, int32_t K,
                  int32_t act, void* stream) {
  dcf::GemmArgs g = dcf::gemm(A, K, W, bias, C, N, M, N, K);
  g.flags = act == 1 ? dcf::G_GELU : act == 2 ? dcf::G_RELU : 0;
  return dcf::launch_gemm(&g, 1, dcf::A_ROWS, (hipStream_t)stream);
}

int dcf_op_linear_split(const float* A, const float* W, const float* bias, float* C, int32_t M, int32_t N, int32_t K,
                        int32_t act, int32_t nterms, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  unsigned short* planes = nullptr;
  DCF_HIP(hipMallocAsync((void**)&planes, (size_t)3 * N * K * sizeof(unsigned short), st));
  int rc = dcf::launch_split_planes(W, planes, N, K, K, st, nterms);
  if (rc == 0) {
    dcf::GemmArgs g = dcf::gemm(A, K, W, bias, C, N, M, N, K);
    g.Ws = planes;
    g.flags = act == 1 ? dcf::G_GELU : act == 2 ? dcf::G_RELU : 0;
    rc = dcf::launch_gemm_split(&g, 1, dcf::A_ROWS, nterms, st);
  }
  DCF_HIP(hipFreeAsync(planes, st));
  return rc;
}

int dcf_op_linear_cm(const float* A_cm, const float* W, const float* bias, float* C, int32_t M, int32_t N, int32_t K,
                     void* stream) {
  dcf::GemmArgs g = dcf::gemm(A_cm, M, W, bias, C, N, M, N, K);
  return dcf::launch_gemm(&g, 1, dcf::A_CHANMAJOR, (hipStream_t)stream);
}

int dcf_op_linear_ln(const float* A, const float* W, const float* bias, const float* ln_w, const float* ln_b, float* C, float* Y,
                     int32_t M, int32_t N, int32_t K, int32_t relu, int32_t nterms, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  DCF_CHECK(dcf::gemm_can_fuse_ln(M, N, K, dcf::A_ROWS), "dcf_op_linear_ln: %dx%dx%d cannot carry a fused LayerNorm (N = 256, M >= 28672)", M, N, K);
  unsigned short* planes = nullptr;
  DCF_HIP(hipMallocAsync((void**)&planes, (size_t)3 * N * K * sizeof(unsigned short), st));
  int rc = dcf::launch_split_planes(W, planes, N, K, K, st, nterms);
  if (rc == 0) {
    dcf::GemmArgs g = dcf::gemm(A, K, W, bias, C, N, M, N, K);
    g.Ws = planes;
    g.ln_w = ln_w; g.ln_b = ln_b; g.Y = Y; g.ldy = N; g.ln_relu = relu;
    rc = dcf::launch_gemm_split(&g, 1, dcf::A_ROWS, nterms, st);
  }
  DCF_HIP(hipFreeAsync(planes, st));
  return rc;
}

int dcf_op_linear_ln_carry(const float* A, const float* W1, const float* b1, const float* R, const float* ln_w, const float* ln_b,
                           const float* W2, const float* b2, float* X, float* Y, int32_t M, int32_t N1, int32_t K1, int32_t N2,
                           int32_t gelu, int32_t nterms, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  DCF_CHECK(A && W1 && ln_w && ln_b && W2 && X && Y, "dcf_op_linear_ln_carry: null argument");
  DCF_CHECK(N1 % 64 == 0 && dcf::gemm_can_carry_stats(M, N1, K1, 1, nterms) && dcf::gemm_can_carry_stats(M, N2, N1, 1, nterms),
            "dcf_op_linear_ln_carry: %dx%dx%d -> %d runs on the k-sliced kernel (no row statistics there)", M, N1, K1, N2);
  unsigned short *p1 = nullptr, *p2 = nullptr;
  float *wf = nullptr, *stats = nullptr;
  DCF_HIP(hipMallocAsync((void**)&p1, (size_t)3 * N1 * K1 * sizeof(unsigned short), st));
  DCF_HIP(hipMallocAsync((void**)&p2, (size_t)3 * N2 * N1 * sizeof(unsigned short), st));
  DCF_HIP(hipMallocAsync((void**)&wf, ((size_t)N2 * N1 + 2 * (size_t)N2) * sizeof(float), st));
  DCF_HIP(hipMallocAsync((void**)&stats, (size_t)M * (N1 / 64) * 2 * sizeof(float), st));
  float* sv = wf + (size_t)N2 * N1;
  hipLaunchKernelGGL(dcf::k_fold_ln, dim3(N2), dim3(64), 0, st, W2, b2, ln_w, ln_b, wf, sv, sv + N2, N1);
  int rc = dcf::launch_split_planes(W1, p1, N1, K1, K1, st, nterms);
  if (rc == 0) rc = dcf::launch_split_planes(wf, p2, N2, N1, N1, st, nterms);
  if (rc == 0) {
    dcf::GemmArgs g = dcf::gemm(A, K1, W1, b1, X, N1, M, N1, K1);
    g.Ws = p1; g.stats_out = stats; g.stats_w = 64;
    if (R) { g.flags = dcf::G_RES; g.R = R; g.ldr = N1; }
    rc = dcf::launch_gemm_split(&g, 1, dcf::A_ROWS, nterms, st);
  }
  if (rc == 0) {
    dcf::GemmArgs g = dcf::gemm(X, N1, wf, sv + N2, Y, N2, M, N2, N1);
    g.Ws = p2; g.flags = gelu ? dcf::G_GELU : 0;
    g.stats_in = stats; g.ln_s = sv; g.stats_slots = N1 / 64; g.stats_w = 64;
    rc = dcf::launch_gemm_split(&g, 1, dcf::A_ROWS, nterms, st);
  }
  DCF_HIP(hipFreeAsync(p1, st)); DCF_HIP(hipFreeAsync(p2, st)); DCF_HIP(hipFreeAsync(wf, st)); DCF_HIP(hipFreeAsync(stats, st));
  return rc;
}

int dcf_op_ffn(const float* X, const float* ln_w, const float* ln_b, const float* W1, const float* b1, const float* W2, const float* b2,
               const float* ls, const uint8_t* mask, float* C, float* stats_out, int32_t M, int32_t E, int32_t chain, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  DCF_CHECK(X && W1 && b1 && W2 && b2 && C && M > 0 && E % 64 == 0, "dcf_op_ffn: bad argument");
  DCF_CHECK(!chain || E == 256, "dcf_op_ffn: the one-kernel form exists for E = 256 only");
  DCF_CHECK(chain >= 0 && chain <= 3, "dcf_op_ffn: chain = %d (0 .. 3)", chain);
  // (the one-kernel form reads a row twice -- as X and, a tile later, as the residual -- and the retry below re-reads X after C is written)
  DCF_CHECK(C != X, "dcf_op_ffn: C must not alias X");
  DCF_CHECK(!stats_out || chain || dcf::gemm_can_carry_stats(M, E, 4 * E, 1, dcf::GEMM_F16X3), "dcf_op_ffn: %d rows run on a kernel without row statistics", M);
  const int H = 4 * E, nterms = dcf::GEMM_F16X3;
  unsigned short *p1 = nullptr, *p2 = nullptr;
  float *wf = nullptr, *stats = nullptr, *xn = nullptr, *hid = nullptr;
  DCF_HIP(hipMallocAsync((void**)&p1, (size_t)3 * H * E * sizeof(unsigned short), st));
  DCF_HIP(hipMallocAsync((void**)&p2, (size_t)3 * H * E * sizeof(unsigned short), st));
  const float *fc_w = W1, *fc_b = b1, *fc_s = nullptr, *fc_in = X;
  int rc = 0;
  if (ln_w && chain) {           // the LayerNorm rides as row statistics, its gain folded into the fc weight (GemmArgs::stats_in)
    DCF_HIP(hipMallocAsync((void**)&wf, ((size_t)H * E + 2 * (size_t)H) * sizeof(float), st));
    DCF_HIP(hipMallocAsync((void**)&stats, (size_t)M * (E / 64) * 2 * sizeof(float), st));
    float* sv = wf + (size_t)H * E;
    hipLaunchKernelGGL(dcf::k_fold_ln, dim3(H), dim3(64), 0, st, W1, b1, ln_w, ln_b, wf, sv, sv + H, E);
    rc = dcf::launch_row_stats(X, E, stats, M, E, 64, st);
    fc_w = wf; fc_s = sv; fc_b = sv + H;
  } else if (ln_w) {
    DCF_HIP(hipMallocAsync((void**)&xn, (size_t)M * E * sizeof(float), st));
    dcf::LnArgs ln{}; ln.X = X; ln.ldx = E; ln.Y = xn; ln.ldy = E; ln.w = ln_w; ln.b = ln_b; ln.rows = M; ln.C = E;
    rc = dcf::launch_ln(ln, st);
    fc_in = xn;
  }
  if (rc == 0) rc = dcf::launch_split_planes(fc_w, p1, H, E, E, st, nterms);
  if (rc == 0) rc = dcf::launch_split_planes(W2, p2, E, H, H, st, nterms);
  if (rc == 0 && chain) {
    dcf::FfnChainArgs a{};
    a.X = X; a.ldx = E; a.W1s = p1; a.b1 = fc_b; a.ln_s = fc_s; a.stats = stats; a.stats_slots = E / 64; a.W2s = p2; a.b2 = b2; a.ls = ls;
    a.R = X; a.ldr = E; a.rowmask = mask; a.C = C; a.ldc = E; a.stats_out = stats_out; a.stats_w = 64; a.M = M;
    a.variant = chain == 1 ? 0 : chain - 1;        // chain 2: the four-wave kernel, 3: the eight-wave kernel
    // the sticky numerics word of this call (bit 1: common.h LN_ILL_RATIO): only a folded LayerNorm can raise it, and only then
    // does the call pay for the word and the wait; freed on every path below
    unsigned* word = nullptr;
    unsigned flag = 0u;
    if (ln_w) {
      if (hipMallocAsync((void**)&word, sizeof(unsigned), st) != hipSuccess) { word = nullptr; rc = -1; dcf::set_error("dcf_op_ffn: hipMallocAsync failed"); }
      if (rc == 0 && hipMemsetAsync(word, 0, sizeof(unsigned), st) != hipSuccess) { rc = -1; dcf::set_error("dcf_op_ffn: hipMemsetAsync failed"); }
    }
    a.status = word;
    if (rc == 0) rc = dcf::launch_ffn_chain(a, st);
    if (rc == 0 && word) {
      if (hipMemcpyAsync(&flag, word, sizeof(flag), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
        rc = -1; dcf::set_error("dcf_op_ffn: reading the numerics word failed");
      }
    }
    if (word) (void)hipFreeAsync(word, st);
    if (rc == 0 && ln_w && (flag & 2u)) {
      // a row's mean dwarfs its spread: the folded one-pass statistics are not trustworthy for it -- what the engine does after
      // dcf_model_set_ln_carry(m, 0): the two-pass LayerNorm as its own launch, the same kernel on its output
      DCF_HIP(hipMallocAsync((void**)&xn, (size_t)M * E * sizeof(float), st));
      dcf::LnArgs ln{}; ln.X = X; ln.ldx = E; ln.Y = xn; ln.ldy = E; ln.w = ln_w; ln.b = ln_b; ln.rows = M; ln.C = E;
      rc = dcf::launch_ln(ln, st);
      if (rc == 0) rc = dcf::launch_split_planes(W1, p1, H, E, E, st, nterms);
      a.X = xn; a.b1 = b1; a.ln_s = nullptr; a.stats = nullptr; a.status = nullptr;
      if (rc == 0) rc = dcf::launch_ffn_chain(a, st);
    }
  } else if (rc == 0) {
    DCF_HIP(hipMallocAsync((void**)&hid, (size_t)M * H * sizeof(float), st));
    dcf::GemmArgs gf = dcf::gemm(fc_in, E, fc_w, fc_b, hid, H, M, H, E);
    gf.Ws = p1; gf.flags = dcf::G_GELU;
    rc = dcf::launch_gemm_split(&gf, 1, dcf::A_ROWS, nterms, st);
    if (rc == 0) {
      dcf::GemmArgs go = dcf::gemm(hid, H, W2, b2, C, E, M, E, H);
      go.Ws = p2; go.flags = dcf::G_RES | (mask ? dcf::G_OUT_MASK : 0); go.rowmask = mask; go.ls = ls; go.R = X; go.ldr = E;
      if (stats_out) { go.stats_out = stats_out; go.stats_w = 64; }
      rc = dcf::launch_gemm_split(&go, 1, dcf::A_ROWS, nterms, st);
    }
  }
  DCF_HIP(hipFreeAsync(p1, st)); DCF_HIP(hipFreeAsync(p2, st));
  if (wf) DCF_HIP(hipFreeAsync(wf, st));
  if (stats) DCF_HIP(hipFreeAsync(stats, st));
  if (xn) DCF_HIP(hipFreeAsync(xn, st));
  if (hid) DCF_HIP(hipFreeAsync(hid, st));
  return rc;
}

int dcf_op_linear_cm_split(const float* A_cm, const float* W, const float* bias, float* C, int32_t M, int32_t N, int32_t K,
                           int32_t nterms, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  unsigned short* planes = nullptr;
  DCF_HIP(hipMallocAsync((void**)&planes, (size_t)3 * N * K * sizeof(unsigned short), st));
  int rc = dcf::launch_split_planes(W, planes, N, K, K, st, nterms);
  if (rc == 0) {
    dcf::GemmArgs g = dcf::gemm(A_cm, M, W, bias, C, N, M, N, K);
    g.Ws = planes;
    rc = dcf::launch_gemm_split(&g, 1, dcf::A_CHANMAJOR, nterms, st);
  }
  DCF_HIP(hipFreeAsync(planes, st));
  return rc;
}

int dcf_op_conv3(const float* X, const uint8_t* mask, const float* W_ock, float* Y, int32_t B, int32_t T, int32_t Cin,
                 int32_t N, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int rows = B * T;
  float* wp = nullptr;
  uint8_t* nbr = nullptr;
  DCF_HIP(hipMallocAsync((void**)&wp, (size_t)N * Cin * 3 * sizeof(float), st));
  DCF_HIP(hipMallocAsync((void**)&nbr, (size_t)rows, st));
  const int n = N * Cin * 3;
  hipLaunchKernelGGL(dcf::k_permute3, dim3((n + 255) / 256), dim3(256), 0, st, W_ock, wp, N, Cin, 3, 0, 2, 1);
  int rc = dcf::launch_rowflags(mask, nbr, T, rows, st);
  if (rc == 0) {
    dcf::GemmArgs g = dcf::gemm(X, Cin, wp, nullptr, Y, N, rows, N, 3 * Cin);
    g.cin = Cin; g.nbr = nbr;
    rc = dcf::launch_gemm(&g, 1, dcf::A_ROWS_TAP3, st);
  }
  DCF_HIP(hipFreeAsync(wp, st));
  DCF_HIP(hipFreeAsync(nbr, st));
  return rc;
}

int dcf_op_conv3_split(const float* X, const uint8_t* mask, const float* W_ock, float* Y, int32_t B, int32_t T, int32_t Cin,
                       int32_t N, int32_t nterms, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int rows = B * T;
  float* wp = nullptr;
  uint8_t* nbr = nullptr;
  unsigned short* planes = nullptr;
  DCF_HIP(hipMallocAsync((void**)&wp, (size_t)N * Cin * 3 * sizeof(float), st));
  DCF_HIP(hipMallocAsync((void**)&nbr, (size_t)rows, st));
  DCF_HIP(hipMallocAsync((void**)&planes, (size_t)3 * N * Cin * 3 * sizeof(unsigned short), st));
  const int n = N * Cin * 3;
  hipLaunchKernelGGL(dcf::k_permute3, dim3((n + 255) / 256), dim3(256), 0, st, W_ock, wp, N, Cin, 3, 0, 2, 1);
  int rc = dcf::launch_rowflags(mask, nbr, T, rows, st);
  if (rc == 0) rc = dcf::launch_split_planes(wp, planes, N, 3 * Cin, 3 * Cin, st, nterms);
  if (rc == 0) {
    dcf::GemmArgs g = dcf::gemm(X, Cin, wp, nullptr, Y, N, rows, N, 3 * Cin);
    g.cin = Cin; g.nbr = nbr; g.Ws = planes;
    rc = dcf::launch_gemm_split(&g, 1, dcf::A_ROWS_TAP3, nterms, st);
  }
  DCF_HIP(hipFreeAsync(wp, st));
  DCF_HIP(hipFreeAsync(nbr, st));
  DCF_HIP(hipFreeAsync(planes, st));
  return rc;
}

int dcf_op_head(const float* X, const uint8_t* mask, const float* W1, const float* ln1_w, const float* ln1_b, const float* W2,
                const float* ln2_w, const float* ln2_b, const float* Wout, const float* bout, float* out, int32_t B, int32_t T,
                int32_t C, int32_t NO, float scale, int32_t chain, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  DCF_CHECK(X && mask && W1 && ln1_w && ln1_b && W2 && ln2_w && ln2_b && Wout && bout && out && B > 0 && T > 0, "dcf_op_head: null argument");
  DCF_CHECK((NO == 1 || NO == 2) && C % 32 == 0, "dcf_op_head: NO = %d, C = %d", NO, C);
  DCF_CHECK(!chain || dcf::head_chain_supports(C, NO), "dcf_op_head: the one-kernel form exists for C = 256 / 288 only");
  const int rows = B * T, nterms = dcf::GEMM_F16X3;
  float *wp[2] = {nullptr, nullptr}, *wo = nullptr, *ha = nullptr, *hb = nullptr;
  uint8_t* nbr = nullptr;
  unsigned short* img[2] = {nullptr, nullptr};
  dcf::LevelTable lt{}, *d_lt = nullptr;
  lt.n_levels = 1; lt.B = B; lt.S = T; lt.T[0] = T; lt.start[0] = 0; lt.start[1] = rows; lt.off[0] = 0; lt.scale[0] = scale;
  DCF_HIP(hipMallocAsync((void**)&d_lt, sizeof(lt), st));
  DCF_HIP(hipMemcpyAsync(d_lt, &lt, sizeof(lt), hipMemcpyHostToDevice, st));
  DCF_HIP(hipStreamSynchronize(st));                        // (lt is a stack object)
  DCF_HIP(hipMallocAsync((void**)&nbr, (size_t)rows, st));
  DCF_HIP(hipMallocAsync((void**)&wo, (size_t)NO * C * 3 * sizeof(float), st));
  const size_t img_halfs = chain ? dcf::head_chain_image_halfs(C) : (size_t)3 * C * C * 3;
  const float* Ws[2] = {W1, W2};
  int rc = dcf::launch_rowflags(mask, nbr, T, rows, st);
  for (int i = 0; i < 2 && rc == 0; ++i) {
    DCF_HIP(hipMallocAsync((void**)&wp[i], (size_t)C * C * 3 * sizeof(float), st));
    DCF_HIP(hipMallocAsync((void**)&img[i], img_halfs * sizeof(unsigned short), st));
    const int n = C * C * 3;
    hipLaunchKernelGGL(dcf::k_permute3, dim3((n + 255) / 256), dim3(256), 0, st, Ws[i], wp[i], C, C, 3, 0, 2, 1);
    rc = chain ? dcf::launch_split_chain3(wp[i], img[i], C, st) : dcf::launch_split_planes(wp[i], img[i], C, 3 * C, 3 * C, st, nterms);
  }
  {
    const int n = NO * C * 3;
    hipLaunchKernelGGL(dcf::k_permute3, dim3((n + 255) / 256), dim3(256), 0, st, Wout, wo, NO, C, 3, 0, 2, 1);
  }
  const int mode = scale != 0.f ? 1 : 0;                    // scale = 0: raw logits (ClsHead); otherwise relu(scale * y) (RegHead)
  if (rc == 0 && chain) {
    dcf::HeadChainArgs a{};
    a.X = X; a.ldx = C; a.nbr = nbr; a.W1c = img[0]; a.W2c = img[1]; a.ln1_w = ln1_w; a.ln1_b = ln1_b; a.ln2_w = ln2_w; a.ln2_b = ln2_b;
    a.Wout = wo; a.bout = bout; a.lt = d_lt; a.out = out; a.rows = rows; a.NO = NO; a.mode = mode; a.query_major = 0;
    rc = dcf::launch_head_chain(&a, 1, C, st);
  } else if (rc == 0) {
    DCF_HIP(hipMallocAsync((void**)&ha, (size_t)rows * C * sizeof(float), st));
    DCF_HIP(hipMallocAsync((void**)&hb, (size_t)rows * C * sizeof(float), st));
    const float* in = X;
    float* bufs[2] = {ha, hb};
    const float *lw[2] = {ln1_w, ln2_w}, *lb[2] = {ln1_b, ln2_b};
    for (int i = 0; i < 2 && rc == 0; ++i) {
      dcf::GemmArgs g = dcf::gemm(in, C, wp[i], nullptr, bufs[i], C, rows, C, 3 * C);
      g.cin = C; g.nbr = nbr; g.Ws = img[i];
      rc = dcf::launch_gemm_split(&g, 1, dcf::A_ROWS_TAP3, nterms, st);
      if (rc == 0) {
        dcf::LnArgs ln{}; ln.X = bufs[i]; ln.ldx = C; ln.Y = bufs[i]; ln.ldy = C; ln.w = lw[i]; ln.b = lb[i]; ln.rows = rows; ln.C = C; ln.relu = 1;
        rc = dcf::launch_ln(ln, st);
      }
      in = bufs[i];
    }
    if (rc == 0) {
      dcf::ConvOutArgs co{};
      co.X = in; co.ldx = C; co.nbr = nbr; co.W = wo; co.bias = bout; co.lt = d_lt; co.out = out; co.rows = rows; co.C = C; co.NO = NO;
      co.row0 = 0; co.mode = mode; co.query_major = 0;
      rc = dcf::launch_conv_out(co, st);
    }
  }
  for (int i = 0; i < 2; ++i) { if (wp[i]) DCF_HIP(hipFreeAsync(wp[i], st)); if (img[i]) DCF_HIP(hipFreeAsync(img[i], st)); }
  if (ha) DCF_HIP(hipFreeAsync(ha, st));
  if (hb) DCF_HIP(hipFreeAsync(hb, st));
  DCF_HIP(hipFreeAsync(wo, st)); DCF_HIP(hipFreeAsync(nbr, st)); DCF_HIP(hipFreeAsync(d_lt, st));
  return rc;
}

int dcf_op_layernorm(const float* X, const float* w, const float* b, float* Y, int32_t rows, int32_t C, int32_t relu,
                     void* stream) {
  dcf::LnArgs a{};
  a.X = X; a.ldx = C; a.Y = Y; a.ldy = C; a.w = w; a.b = b; a.rows = rows; a.C = C; a.relu = relu;
  return dcf::launch_ln(a, (hipStream_t)stream);
}

int dcf_op_xattn(const float* Q, const float* K, const float* V, const uint8_t* kvmask, float* O, int32_t B, int32_t T,
                 int32_t Lk, int32_t C, int32_t heads, void* stream) {
  dcf::XAttnArgs a{Q, K, V, kvmask, O, B, T, Lk, C, heads};
  return dcf::launch_xattn(a, (hipStream_t)stream);
}

int dcf_op_local_attn(const float* Q, const float* K, const float* V, const uint8_t* mask, float* O, int32_t B, int32_t T,
                      int32_t C, int32_t heads, int32_t window, void* stream) {
  if (window == 0) {
    dcf::GlobalAttnArgs g{Q, K, V, mask, O, B, T, C, heads};
    return dcf::launch_global_attn(g, (hipStream_t)stream);
  }
  dcf::LocalAttnArgs a{Q, K, V, mask, O, B, T, C, heads, window};
  return dcf::launch_local_attn(a, (hipStream_t)stream);
}

int dcf_op_sidekick(const float* shallow, const float* text_cls, float* correl, int32_t D, int32_t T, int32_t nq,
                    int32_t norm, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  float *tn = nullptr, *partial = nullptr;
  DCF_HIP(hipMallocAsync((void**)&tn, (size_t)nq * D * sizeof(float), st));
  DCF_HIP(hipMallocAsync((void**)&partial, (size_t)dcf::SCORE_SLICES * (nq + 1) * T * sizeof(float), st));
  int rc = dcf::launch_sidekick(dcf::score_args(shallow, text_cls, tn, partial, correl, D, T, nq, norm), st);
  DCF_HIP(hipFreeAsync(tn, st));
  DCF_HIP(hipFreeAsync(partial, st));
  return rc;
}

int dcf_op_gate(const float* correl, const uint8_t* vid_mask, float* gate, uint8_t* mask_out, int32_t T, int32_t nq,
                int32_t sn, double sratio, int32_t msf, void* stream) {
  dcf::GateArgs a{correl, vid_mask, gate, mask_out, T, nq, 0, sn, msf, sratio};
  return dcf::launch_gate(a, (hipStream_t)stream);
}

// ---- composite blocks on a scratch model (parity tests against the reference's operator fixtures) ----------------
namespace dcf {
// a scratch model holds only the bound parameters of one block: never finalized, its packed images are dropped after every call
static int scratch_begin(dcf_model* m, const char* what, hipStream_t st) {
  DCF_CHECK(m && !m->finalized, "%s: needs a scratch model (dcf_model_create + dcf_model_bind of the block's parameters, not finalized)", what);
  return init_gemm_mode(m, st);
}
static int scratch_end(dcf_model* m, hipStream_t st, int rc) {
  (void)hipStreamSynchronize(st);
  for (float* p : m->owned) (void)hipFree(p);
  m->owned.clear();
  m->wsplit.clear(); m->wsplit_ldw.clear(); m->wsplit_terms.clear();
  m->dec.clear();
  m->fus_out_w = m->fus_out_b = nullptr;
  return rc;
}
struct ScratchArena {
  char* base = nullptr;
  ~ScratchArena() { if (base) (void)hipFree(base); }
};
}  // namespace dcf

int dcf_op_encoder(dcf_model* m, const char* prefix, const float* X, const uint8_t* mask, int32_t B, int32_t T, int32_t stride,
                   float* Y, uint8_t* mask_out, void* stream) {
  using namespace dcf;
  hipStream_t st = (hipStream_t)stream;
  DCF_CHECK(prefix && X && mask && Y && mask_out && B >= 1 && T >= 1 && (stride == 1 || stride == 2) && T % stride == 0, "dcf_op_encoder: bad arguments");
  if (scratch_begin(m, "dcf_op_encoder", st)) return -1;
  const dcf_config& c = m->cfg;
  const int half = c.win / 2;
  DCF_CHECK(half == 0 || (T / stride) % half == 0, "dcf_op_encoder: T / stride = %d must be a multiple of win//2 = %d (blocks.py:216)", T / stride, half);
  EncW w{};
  int rc = resolve_encoder(m, prefix, c.E, st, w);
  ScratchArena sa;
  if (rc == 0) {
    Buffers b{};
    Arena dry{nullptr, 0, 0, true};
    carve(dry, c, T, B, B, T, 1, 1, b);
    if (hipMalloc(&sa.base, dry.off) != hipSuccess) { set_error("dcf_op_encoder: out of memory"); rc = -1; }
    if (rc == 0) {
      Arena real{sa.base, 0, dry.off, false};
      carve(real, c, T, B, B, T, 1, 1, b);
      if (stride == 2) rc = launch_mask_down(mask, mask_out, B * T / 2, st);
      else if (hipMemcpyAsync(mask_out, mask, (size_t)B * T, hipMemcpyDeviceToDevice, st) != hipSuccess) rc = -1;
      if (rc == 0) rc = run_encoder(m, w, b, X, c.E, mask, mask_out, B, T, stride, Y, c.E, st);
    }
  }
  return scratch_end(m, st, rc);
}

int dcf_op_enc_pre(dcf_model* m, const char* prefix, const float* X, const uint8_t* mask, int32_t B, int32_t T, int32_t stride,
                   float* Qc, float* Kc, float* Vc, float* Skip, void* stream) {
  using namespace dcf;
  hipStream_t st = (hipStream_t)stream;
  DCF_CHECK(prefix && X && mask && Qc && Kc && Vc && B >= 1 && T >= 1 && (stride == 1 || (stride == 2 && Skip)) && T % stride == 0, "dcf_op_enc_pre: bad arguments");
  if (scratch_begin(m, "dcf_op_enc_pre", st)) return -1;
  EncW w{};
  int rc = resolve_encoder(m, prefix, m->cfg.E, st, w);
  if (rc == 0) {
    EncPreArgs ep{};
    ep.X = X; ep.ldx = m->cfg.E; ep.mask_in = mask; ep.ln_w = w.ln_attn_w; ep.ln_b = w.ln_attn_b;
    ep.dw_q = w.dw_q; ep.dw_k = w.dw_k; ep.dw_v = w.dw_v;
    ep.qn_w = w.qn_w; ep.qn_b = w.qn_b; ep.kn_w = w.kn_w; ep.kn_b = w.kn_b; ep.vn_w = w.vn_w; ep.vn_b = w.vn_b;
    ep.Qc = Qc; ep.Kc = Kc; ep.Vc = Vc; ep.Skip = stride == 2 ? Skip : nullptr;
    ep.B = B; ep.T_in = T; ep.C = m->cfg.E;
    rc = launch_enc_pre(ep, stride, st);
  }
  return scratch_end(m, st, rc);
}

int dcf_op_decoder(dcf_model* m, const char* prefix, float* X, const uint8_t* mask, int32_t B, int32_t T,
                   const float* const* text, const uint8_t* const* text_mask, const int32_t* text_len, void* stream) {
  using namespace dcf;
  hipStream_t st = (hipStream_t)stream;
  DCF_CHECK(prefix && X && mask && text && text_len && B >= 1 && B <= DCF_MAX_BATCH && T >= 1, "dcf_op_decoder: bad arguments");
  if (scratch_begin(m, "dcf_op_decoder", st)) return -1;
  const dcf_config& c = m->cfg;
  DecW w{};
  int rc = resolve_decoder(m, prefix, c.E, c.TE, st, w);
  ScratchArena sa;
  if (rc == 0) {
    m->dec.assign(1, w);
    m->fus_out_w = m->fus_out_b = nullptr;
    int Lk = 1;
    TextMeta tm{};
    for (int i = 0; i < B; ++i) {
      tm.text[i] = text[i]; tm.text_mask[i] = text_mask ? text_mask[i] : nullptr; tm.len[i] = text_len[i];
      Lk = std::max(Lk, (int)text_len[i]);
    }
    Buffers b{};
    Arena dry{nullptr, 0, 0, true};
    carve(dry, c, T, B, B, T, Lk, 1, b);
    if (hipMalloc(&sa.base, dry.off) != hipSuccess) { set_error("dcf_op_decoder: out of memory"); rc = -1; }
    if (rc == 0) {
      Arena real{sa.base, 0, dry.off, false};
      carve(real, c, T, B, B, T, Lk, 1, b);
      rc = run_fusion(m, b, X, c.E, B, T, nullptr, mask, nullptr, &tm, Lk, X, c.E, st);
    }
  }
  return scratch_end(m, st, rc);
}

int dcf_op_tcn(dcf_model* m, const char* prefix, const float* x, const uint8_t* mask, int32_t B, int32_t T, int32_t n_in,
               int32_t n_layers, float* Y, void* stream) {
  using namespace dcf;
  hipStream_t st = (hipStream_t)stream;
  DCF_CHECK(prefix && x && mask && Y && B >= 1 && T >= 1 && n_in >= 1 && n_in <= DCF_MAX_LEVELS && n_layers >= 0, "dcf_op_tcn: bad arguments");
  if (scratch_begin(m, "dcf_op_tcn", st)) return -1;
  int rc = resolve_tcn(m, prefix, n_in, n_layers, st);
  float* buf = nullptr;
  if (rc == 0 && hipMalloc(&buf, (size_t)2 * B * T * TCN_HID * sizeof(float)) != hipSuccess) { set_error("dcf_op_tcn: out of memory"); rc = -1; }
  if (rc == 0) {
    LevelTable lt{};
    lt.n_levels = n_in; lt.B = B; lt.T[0] = T; lt.S = T;
    RefineArgs ra{};
    ra.stacked = x; ra.mask_all = mask;
    ra.w_in = m->tcn_in_w; ra.b_in = m->tcn_in_b;
    ra.host_w_dil = m->tcn_wd.data(); ra.host_b_dil = m->tcn_bd.data(); ra.host_w_pw = m->tcn_wp.data();
    ra.host_b_pw = m->tcn_bp.data(); ra.host_ln_w = m->tcn_lnw.data(); ra.host_ln_b = m->tcn_lnb.data();
    ra.w_out = m->tcn_out_w; ra.b_out = m->tcn_out_b;
    ra.host_frag = (!m->tcn_frag.empty() && dcf::debug_option("tcn_frag", 1) != 0) ? m->tcn_frag.data() : nullptr;
    ra.stack_layers = dcf::debug_option("tcn_stack", tcn_stack_env());          // (dcf_debug_set_option: 0 = layer by layer)
    ra.bufA = buf; ra.bufB = buf + (size_t)B * T * TCN_HID; ra.F = Y; ra.ldf = TCN_HID; ra.E = 0;
    ra.B = B; ra.T0 = T; ra.n_levels = n_in; ra.n_layers = n_layers;
    ra.f16 = m->gemm_terms == GEMM_F16X3; ra.status = m->status;
    rc = launch_refine(ra, lt, st);
  }
  rc = scratch_end(m, st, rc);
  if (buf) (void)hipFree(buf);
  return rc;
}

}  // extern "C"
