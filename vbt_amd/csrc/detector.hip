// EfficientDet-Lite int8 detector on gfx950 (MI355X): execution plan (planner + autotuner), launches and the C ABI.
// The single-op kernels are in op_kernels.h (included below); the fused kernel families live in their own translation units
// (k_*.hip, reached through launchers.h).
//
// Replaces the TFLite interpreter invoke at reference odt.py:58-66 (signature_fn(images=...)).
// Arithmetic contract = the kernels tflite-runtime 2.14 executes on x86-64 (XNNPACK delegate by default, TFLite builtin
// kernels for what it does not take; table in vbt_amd/quant.py), bit-exact with oracle/detector.c:
//   conv:  acc(int32) = sum (x_q - z_x) * w_q + bias_q ;  q = clamp(rne(float(acc) * M[c]) + z_y)   (XNNPACK qs8-qc8w, fp32)
//   add :  q = clamp(((bias + a*a_mult + b*b_mult) >> shift) + z_y), binary, integer                 (XNNPACK qs8-vadd-minmax)
//   post:  LOGISTIC / DEQUANTIZE tables, centre-size decode in double rounded to float once per quantity, greedy NMS in
//          (score desc, anchor asc) order                                                            (detection_postprocess.cc)
// Design (MI355X-first):
//   * activations int8 NHWC, batch-major [B][H][W][C]; pointwise convs run on the double-rate int8 MFMA
//     (v_mfma_i32_16x16x64_i8; the stem conv and the fused tiles' expand / project stages on v_mfma_i32_16x16x32_i8) with the
//     WEIGHTS as the A operand so every lane ends up holding 16 consecutive output channels of one pixel -> one 16-byte
//     coalesced store per lane;
//   * stand-alone depthwise convs: row / column walkers (v_cvt_f32_ubyteN + v_fma_f32, exact: |sum| < 2^24, 4 channels per
//     lane on contiguous NHWC channel vectors) or LDS tiles on the matrix pipe (diagonal-embedded weights);
//   * zero points are folded into the bias on the host at load time; padding uses the zero point;
//   * decode + NMS: one workgroup per frame, 256-bin score histogram -> bitonic sort of the top
//     candidates in LDS -> greedy suppression by one wavefront with ballot/shuffle.
#include <algorithm>
#include <cmath>
#include <chrono>
#include <functional>
#include <map>
#include <set>
#include <type_traits>
#include <tuple>

#include "launchers.h"   // dev_common.h + the argument structs / tile constants of every kernel family + the launchers

namespace vbt {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

bool lds_opt_in(const void* fn, LdsOptIn* state) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { set_error("lds_opt_in: no current HIP device"); return false; }
  if (state->dev[dev] > 0) return true;
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    set_error("device %d refuses more than 64 KB of dynamic LDS for a kernel that needs it: %s", dev, hipGetErrorString(e));
    return false;
  }
  state->dev[dev] = 1;
  return true;
}

#include "op_kernels.h"   // the single-op kernels (pw_a / pw_b / pw_c / pw_d, stem, depthwise, add, pool, resize, decode + NMS, frame resize)

// ------------------------------------------------------------------------------------------
// host: model, plan, launches
// ------------------------------------------------------------------------------------------
enum Family { F_STEM = 0, F_PW, F_DW, F_ADD, F_MAXPOOL, F_RESIZE, F_POST, F_MBCONV, F_SEPCONV, F_NODE, F_MULTI, F_STEMBLK, F_EXPDW, F_BAND, F_COUNT };
static const char* kFamilyName[F_COUNT] = {"stem_conv_mfma_i8", "pw_conv_mfma_i8", "dw_conv_f32acc", "add_requant",
                                           "maxpool3x3s2", "resize_nn", "decode_nms", "fused_mbconv", "fused_sepconv", "fused_bifpn_node", "fused_heads_multi",
                                           "fused_stem_block", "fused_expand_dw", "fused_sepconv_band"};

struct Step {
  int op;       // index into ops
  int family;
  // conv
  long* wp = nullptr;      // packed MFMA weights, 16x16x32 layout (device): stem kernel and the expand stage of the fused kernels
  v4i* wp64 = nullptr;     // pointwise convs: 16x16x64 layout (pack_weights64)
  int KS64 = 0;            // K-steps of 64
  int res_op = -1;         // F_PW: the residual ADD evaluated in the epilogue (op = that ADD, p_op = the conv)
  long* wdm = nullptr;     // depthwise: matrix-pipe (diagonal-embedded) weights
  int* bdm = nullptr;      // depthwise: bias folded for raw int8 inputs, padded to 64
  float* mdm = nullptr;    // depthwise: multipliers padded to 64
  float* wf = nullptr;     // depthwise weights as float [k*k][C] (device)
  int* bias = nullptr;     // folded bias (device, padded)
  float* mult = nullptr;   // multipliers (device, padded)
  int KS = 0, NB = 0;
  AddQ addq = {0, 0, 0, 0, 0, 0, 0};   // F_ADD: XNNPACK qs8-vadd parameters, derived from the tensor scales
  double alg_bytes_per_frame = 0, weight_bytes = 0, macs_per_frame = 0;
  // fused block (F_MBCONV / F_SEPCONV): constituent op indices (-1 = absent) and kernel arguments
  int e_op = -1, d_op = -1, p_op = -1, a_op = -1;
  int sum_op = -1;          // F_NODE: the n-ary ADD feeding the depthwise
  int src_tensor[3] = {-1, -1, -1};
  FusedArgs fa;
  int nbp = 0, lds_bytes = 0;
  int variant = -1;  // kernel variant chosen by the autotuner (-1 = heuristic default)
  double tuned_ms = 0;
  // F_MULTI: independent fused problems launched as one grid
  std::vector<Step> members;
  FusedArgs* d_multi = nullptr;
  // F_STEMBLK: stem -> depthwise -> project in one kernel (op = project op, e_op = stem op)
  StemBlockArgs sb;
  // F_BAND: SeparableConv / BiFPN node on row bands (band_block.h); members non-empty: several problems in one grid
  BandArgs bd_args;
  BandArgs* d_band = nullptr;   // device copy of the problem list (pointers are those of the whole batch)
  BandArgs ba;                  // single problem: passed by value
  int band_tiles = 0;           // workgroups per image of this problem
  // F_EXPDW: expand + depthwise on whole images, expanded channels split over workgroups (expdw_block.h; op = depthwise op)
  ExpDwArgs xd;
  ExpDw2Args xd2;              // the same step on the second form of the kernel (expdw2_block.h); variant 100 + cpw runs it
  bool xd2_ok = false;
  int xd2_lds = 0, xd2_gpw = 0, xd2_gpw16 = 0;   // input pixel groups per wave on 8 / 16 waves (0: that wave count is not available)
  // F_MBCONV on a low-resolution map: per-chunk weight records of the whole-image kernel (data == nullptr: not built)
  ImageBundle ib = {nullptr, 0, 0, 0, 0, 0, 0, 0};
};

// A group of consecutive graph ops with alternative realisations (all bit-identical); the planner keeps
// the fastest one measured on this device at this batch size.
struct Alt {
  std::vector<Step> steps;
  std::vector<int> hidden;  // tensors that never reach HBM under this alternative
  double ms = 0;
};
struct Group {
  std::vector<Alt> alts;
  int chosen = 0;
};

}  // namespace vbt

using namespace vbt;

struct vbt_model {
  Header hdr;
  std::vector<TensorRec> tensors;
  std::vector<OpRec> ops;
  std::vector<uint8_t> blob;
  int device = 0, max_batch = 0;
  std::vector<int8_t*> tptr;   // device pointer of each tensor ([max_batch][h][w][c])
  std::vector<size_t> telems;  // per-frame elements
  int8_t* arena = nullptr;
  uint8_t* frames_stage = nullptr;  // device staging for host frames
  float* out_boxes = nullptr;       // device staging for host outputs: ONE block boxes | scores | classes | counts ...
  float* out_scores = nullptr;
  float* out_classes = nullptr;
  int* out_counts = nullptr;
  unsigned char* out_host = nullptr;   // ... and its pinned host mirror: vbt_detect's results come back with one copy
  size_t out_bytes = 0;
  float* d_anchors = nullptr;
  unsigned char* d_luts = nullptr;   // post-process tables (see PostArgs)
  std::vector<float> post_tables_host;   // scores indexed by rank byte + 128
  std::vector<Step> steps;      // execution list (after fusion + autotuning)
  std::vector<Group> groups;
  std::vector<Step> op_steps;   // one per graph op (weights live here)
  std::vector<char> materialized;  // per tensor: written to HBM by the execution list
  int flags = 0;
  int n_sub = 1;                         // sub-batches run concurrently on side streams
  hipStream_t sub_streams[4] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_fork = nullptr, ev_join[4] = {nullptr, nullptr, nullptr, nullptr};
  // hipGraph replay of the forward for launch-bound (small) batches: one executable graph per (B, buffers)
  struct GraphKey {
    const void* frames; void* boxes; void* scores; void* classes; void* counts; int B;
    bool operator<(const GraphKey& o) const {
      return std::tie(frames, boxes, scores, classes, counts, B) < std::tie(o.frames, o.boxes, o.scores, o.classes, o.counts, o.B);
    }
  };
  std::map<GraphKey, hipGraphExec_t> graphs;
  hipStream_t cap_stream = nullptr;
  int graph_max_batch = 0;  // 0 = graphs off
  bool ran_eager = false;   // one forward has been enqueued outside a stream capture (per-device LDS opt-ins, lazy uploads)
  std::vector<void*> owned;  // device allocations to free
  // Parameter pool: weights, biases, multipliers and argument tables are sub-allocated from a few large device chunks and
  // mirrored on the host; flush_uploads() brings a chunk up to date with ONE copy (a model used to issue ~1 800 small blocking
  // hipMemcpy calls at creation).  Off under VBT_DEBUG_FENCE, where every buffer ends at its own allocation boundary.
  struct PoolChunk { char* dev; std::vector<char> host; size_t used, flushed; };
  std::vector<PoolChunk> pool;
  bool pool_dirty = false;
  int last_B = 0;
};

namespace vbt {

// Debug "electric fence" (VBT_DEBUG_FENCE=1): every device buffer is placed so that it ENDS at the end of its own
// 2 MiB-granular allocation; a kernel reading past the documented slack then touches unmapped memory and faults
// instead of silently reading a neighbour.  Used once per model family by tests/tools, never in production.
static bool fence_on() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("VBT_DEBUG_FENCE"); v = (e && e[0] == '1') ? 1 : 0; }
  return v == 1;
}
static hipError_t fenced_malloc(vbt_model* m, void** out, size_t bytes) {
  if (!fence_on()) {
    hipError_t e = hipMalloc(out, bytes);
    if (e == hipSuccess) m->owned.push_back(*out);
    return e;
  }
  const size_t G = 2u << 20;
  size_t total = (bytes + G - 1) / G * G;
  char* base = nullptr;
  hipError_t e = hipMalloc((void**)&base, total);
  if (e != hipSuccess) return e;
  m->owned.push_back(base);
  *out = base + ((total - bytes) & ~(size_t)255);   // keep 256-B alignment; the buffer ends <= 255 B before the fence
  return hipSuccess;
}

constexpr size_t POOL_CHUNK = 32u << 20;
static int pool_alloc(vbt_model* m, size_t bytes, void** dev, char** host) {
  bytes = (bytes + 255) & ~(size_t)255;   // 256-byte alignment, like hipMalloc
  if (m->pool.empty() || m->pool.back().used + bytes > m->pool.back().host.size()) {
    vbt_model::PoolChunk c;
    c.dev = nullptr; c.used = 0; c.flushed = 0;
    const size_t cap = std::max(POOL_CHUNK, bytes);
    VBT_HIP_CHECK(hipMalloc((void**)&c.dev, cap));
    m->owned.push_back(c.dev);
    c.host.assign(cap, 0);
    m->pool.push_back(std::move(c));
  }
  vbt_model::PoolChunk& c = m->pool.back();
  *dev = c.dev + c.used;
  *host = c.host.data() + c.used;
  c.used += bytes;
  m->pool_dirty = true;
  return VBT_OK;
}
// Everything uploaded since the last flush reaches the device: one copy per chunk that grew.  Called before any kernel of
// the model can run (launch_step).
static int flush_uploads(vbt_model* m) {
  if (!m->pool_dirty) return VBT_OK;
  for (auto& c : m->pool)
    if (c.used > c.flushed) {
      VBT_HIP_CHECK(hipMemcpy(c.dev + c.flushed, c.host.data() + c.flushed, c.used - c.flushed, hipMemcpyHostToDevice));
      c.flushed = c.used;
    }
  m->pool_dirty = false;
  return VBT_OK;
}

template <typename T>
static int upload(vbt_model* m, const std::vector<T>& h, T** d) {
  size_t bytes = std::max<size_t>(h.size() * sizeof(T), 16);
  if (fence_on()) {
    VBT_HIP_CHECK(fenced_malloc(m, (void**)d, bytes + 64));
    if (!h.empty()) VBT_HIP_CHECK(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return VBT_OK;
  }
  char* host = nullptr;
  int rc = pool_alloc(m, bytes + 64, (void**)d, &host);   // (+64: kernels read K-padding bytes past a weight row's end)
  if (rc) return rc;
  if (!h.empty()) memcpy(host, h.data(), h.size() * sizeof(T));
  return VBT_OK;
}

// Packed MFMA A-operand layout: [(nb*KS + ks)*4 + t][lane][8 bytes]; lane (i = lane&15, g = lane>>4) holds
// W[cout = 64nb + 16(i>>2) + 4t + (i&3)][k = 32ks + 8g + j].  kmap translates packed k -> source k (or -1).
static void pack_weights(const int8_t* w, int N, int K, int KS, int NB, const std::vector<int>* kmap, std::vector<long>& out) {
  out.assign((size_t)NB * KS * 4 * 64, 0);
  int8_t* o = (int8_t*)out.data();
  for (int nb = 0; nb < NB; nb++)
    for (int ks = 0; ks < KS; ks++)
      for (int t = 0; t < 4; t++)
        for (int lane = 0; lane < 64; lane++) {
          int i = lane & 15, g = lane >> 4;
          int co = 64 * nb + 16 * (i >> 2) + 4 * t + (i & 3);
          for (int j = 0; j < 8; j++) {
            int kp = 32 * ks + 8 * g + j;
            int k = kmap ? (kp < (int)kmap->size() ? (*kmap)[kp] : -1) : (kp < K ? kp : -1);
            int8_t v = (co < N && k >= 0) ? w[(size_t)co * K + k] : 0;
            o[((((size_t)(nb * KS + ks) * 4 + t) * 64 + lane) * 8) + j] = v;
          }
        }
}


// 16x16x64 layout: [(nb*KS + ks)*4 + t][lane][16 bytes]; lane (i = lane&15, g = lane>>4) holds
// W[cout = 64nb + 16(i>>2) + 4t + (i&3)][k = 64ks + 16g + j], zero beyond N / K.
static void pack_weights64(const int8_t* w, int N, int K, int KS, int NB, std::vector<v4i>& out, const std::vector<int>* kmap = nullptr) {
  out.assign((size_t)NB * KS * 4 * 64, (v4i){0, 0, 0, 0});
  int8_t* o = (int8_t*)out.data();
  for (int nb = 0; nb < NB; nb++)
    for (int ks = 0; ks < KS; ks++)
      for (int t = 0; t < 4; t++)
        for (int lane = 0; lane < 64; lane++) {
          const int i = lane & 15, g = lane >> 4, co = 64 * nb + 16 * (i >> 2) + 4 * t + (i & 3);
          for (int j = 0; j < 16; j++) {
            const int kp = 64 * ks + 16 * g + j;
            const int k = kmap ? (kp < (int)kmap->size() ? (*kmap)[kp] : -1) : (kp < K ? kp : -1);
            o[((((size_t)(nb * KS + ks) * 4 + t) * 64 + lane) * 16) + j] = (co < N && k >= 0) ? w[(size_t)co * K + k] : 0;
          }
        }
}

// Rq::kb (dev_common.h): the int32 accumulator of this conv, bias included, stays inside (-2^22, 2^22) for EVERY input.  Whatever way a
// kernel folds the zero point into its bias, the value it requantises is sum_k (x_k - z_x) w_k + b with |x_k - z_x| <= 255, so
// 255 * sum_k |w_k| + |b| bounds it per output channel.  VBT_NO_KBIAS: never (the kernels then convert with v_cvt_f32_i32).
static int conv_kb(const vbt_model* m, const OpRec& op) {
  const bool off = getenv("VBT_NO_KBIAS") != nullptr;   // (read per model: tests build both flavours in one process)
  if (off || (op.type != OP_STEM && op.type != OP_PW && op.type != OP_DW)) return 0;
  const TensorRec& tin = m->tensors[op.inputs[0]];
  const TensorRec& tout = m->tensors[op.output];
  const int8_t* w = (const int8_t*)(m->blob.data() + op.w_off);
  const int32_t* bq = (const int32_t*)(m->blob.data() + op.b_off);
  const bool dw = op.type == OP_DW;
  const int N = tout.c, K = dw ? op.k * op.k : op.k * op.k * tin.c;
  for (int co = 0; co < N; co++) {
    long long sa = 0;
    for (int k = 0; k < K; k++) sa += std::abs((int)(dw ? w[(size_t)k * N + co] : w[(size_t)co * K + k]));
    if (255 * sa + std::llabs((long long)bq[co]) >= (1ll << 22) - 1) return 0;
  }
  return 1;
}

// ---- fusion pass: MBConv (pw+relu6 -> dw -> pw [-> add]) and SeparableConv (dw -> pw) -> fused_block_kernel ----
static void choose_tile(int OH, int OW, int KK, int S, bool expand, int* TXo, int* TYo, int slots = 64) {
  double best = 1e300;
  for (int TX = 1; TX <= std::min(OW, 64); TX++) {
    int TXp = (TX + 3) & ~3;
    int TY = std::min(OH, slots / TXp);
    if (TY < 1) continue;
    int tiles = ((OW + TX - 1) / TX) * ((OH + TY - 1) / TY);
    int NPh = ((TXp - 1) * S + KK) * ((TY - 1) * S + KK);
    // halo pixels cost expand work + LDS loads; every tile also pays the 64-slot depthwise/project work
    double cost = tiles * ((expand ? 1.0 : 0.35) * NPh + (double)slots);
    if (cost < best) { best = cost; *TXo = TX; *TYo = TY; }
  }
}

// sources of a BiFPN node's sum.  pre_add >= 0: the sum is two chained binary ADDs (3-input sums of a TFLite graph):
// sources 0,1 are the inputs of ops[pre_add], source 2 the other input of the final ADD, chain = 1|2 the position of the
// partial sum among the final ADD's inputs (+1).
struct NodeSrc { int n; int tensor[3]; int mode[3]; int rs_op[3]; int pre_add = -1; int chain = 0; };

static int make_fused(vbt_model* m, int e_op, int d_op, int p_op, int a_op, Step* out, int sum_op = -1, const NodeSrc* ns = nullptr) {
  const OpRec& dop = m->ops[d_op];
  const OpRec& pop = m->ops[p_op];
  const bool expand = e_op >= 0;
  const int in_t = expand ? m->ops[e_op].inputs[0] : dop.inputs[0];
  const TensorRec& tin = m->tensors[in_t];
  const TensorRec& tdin = m->tensors[dop.inputs[0]];
  const TensorRec& tdout = m->tensors[dop.output];
  const TensorRec& tout = m->tensors[pop.output];
  const int Ce = tdin.c, Cp = (Ce + 63) / 64 * 64, kk = dop.k * dop.k;
  Step s;
  s.family = expand ? F_MBCONV : (sum_op >= 0 ? F_NODE : F_SEPCONV);
  s.sum_op = sum_op;
  s.op = a_op >= 0 ? a_op : p_op;  // the op whose output this step writes
  s.e_op = e_op; s.d_op = d_op; s.p_op = p_op; s.a_op = a_op;
  FusedArgs& a = s.fa;
  memset(&a, 0, sizeof(a));
  a.H = tin.h; a.W = tin.w; a.Cin = tin.c; a.OH = tout.h; a.OW = tout.w; a.Cout = tout.c;
  a.pad_t = dop.pad_t; a.pad_l = dop.pad_l;
  choose_tile(tout.h, tout.w, dop.k, dop.stride, expand, &a.TX, &a.TY);
  a.tiles_x = (tout.w + a.TX - 1) / a.TX;
  a.tiles_y = (tout.h + a.TY - 1) / a.TY;
  a.nchunks = Cp / 64;
  a.zx = tin.zero_point;
  const Step& ps = m->op_steps[p_op];
  if (expand) {
    const Step& es = m->op_steps[e_op];
    const OpRec& eop = m->ops[e_op];
    a.we = es.wp; a.be = es.bias; a.me = es.mult; a.KSe = es.KS;
    a.ze = tdin.zero_point; a.loe = eop.act_min; a.hie = eop.act_max;
    a.rqe = make_rq(a.ze, a.loe, a.hie, conv_kb(m, eop));
    // input tile rows hold the real channels (8-byte granules) + 8 bytes of bank spread, not the K padding of the expand
    // (KS * 32): the B-operand reads of the padded K steps run into the next pixel's bytes, which meet zero weights (the
    // last pixel's run into the E tile).  b1: 40 -> 24 bytes, b4 / b5: 72 -> 48 - LDS per workgroup sets the occupancy here.
    a.T0S = ((tin.c + 7) & ~7) + 8;
  } else {
    a.T0S = Cp + 16;
  }
  // depthwise parameters padded to Cp channels
  {
    const int8_t* w = (const int8_t*)(m->blob.data() + dop.w_off);
    const int32_t* bq = (const int32_t*)(m->blob.data() + dop.b_off);
    const float* mu = (const float*)(m->blob.data() + dop.m_off);
    std::vector<float> wf((size_t)kk * Cp, 0.0f), mult(Cp, 0.0f);
    std::vector<int> bias(Cp, 0);
    for (int c = 0; c < Ce; c++) {
      long sw = 0;
      for (int t = 0; t < kk; t++) { wf[(size_t)t * Cp + c] = (float)w[(size_t)t * Ce + c]; sw += w[(size_t)t * Ce + c]; }
      bias[c] = (int)((long)bq[c] - (long)(128 + tdin.zero_point) * sw);
      mult[c] = mu[c];
    }
    float* dwf; int* dbias; float* dmult;
    int rc;
    if ((rc = upload(m, wf, &dwf)) || (rc = upload(m, bias, &dbias)) || (rc = upload(m, mult, &dmult))) return rc;
    a.wd = dwf; a.bd = dbias; a.md = dmult;
    // matrix-pipe form: [chunk][cg][m][lane][8]: lane (i = lane&15 -> channel 64*chunk + 16*cg + i, g): k = 8g + j ->
    // tap 2m + (g>>1), channel-in-group 8(g&1) + j; non-zero only on the diagonal
    const int KT = (kk + 1) / 2;
    std::vector<long> wdm((size_t)(Cp / 64) * 4 * KT * 64, 0);
    std::vector<int> biasm(Cp, 0);
    int8_t* wb = (int8_t*)wdm.data();
    for (int ch = 0; ch < Cp / 64; ch++)
      for (int cg = 0; cg < 4; cg++)
        for (int mi = 0; mi < KT; mi++)
          for (int lane = 0; lane < 64; lane++) {
            int i = lane & 15, g = lane >> 4;
            int c = 64 * ch + 16 * cg + i;
            int tap = 2 * mi + (g >> 1);
            for (int j = 0; j < 8; j++) {
              int cp = 8 * (g & 1) + j;
              int8_t v = (tap < kk && cp == i && c < Ce) ? w[(size_t)tap * Ce + c] : 0;
              wb[((((size_t)(ch * 4 + cg) * KT + mi) * 64 + lane) * 8) + j] = v;
            }
          }
    for (int c = 0; c < Ce; c++) {
      long sw = 0;
      for (int t = 0; t < kk; t++) sw += w[(size_t)t * Ce + c];
      biasm[c] = (int)((long)bq[c] - (long)tdin.zero_point * sw);
    }
    long* dwdm; int* dbm;
    if ((rc = upload(m, wdm, &dwdm)) || (rc = upload(m, biasm, &dbm))) return rc;
    a.wdm = dwdm; a.bdm = dbm;
    if (expand && (dop.k == 3 || dop.k == 5)) {   // 16x16x64 form (FusedArgs::wd64)
      const int K64 = dop.k == 3 ? 3 : 7, k = dop.k;
      std::vector<v4i> w64((size_t)(Cp / 16) * K64 * 64, (v4i){0, 0, 0, 0});
      int8_t* o = (int8_t*)w64.data();
      for (int q = 0; q < Cp / 16; q++)
        for (int mi = 0; mi < K64; mi++)
          for (int lane = 0; lane < 64; lane++) {
            const int i = lane & 15, g = lane >> 4, c = 16 * q + i;
            int tap = -1;
            if (k == 3) tap = g < 3 ? mi * 3 + g : -1;
            else if (mi < 5) tap = mi * 5 + g;
            else if (mi == 5) tap = g * 5 + 4;
            else tap = g == 0 ? 24 : -1;
            if (tap >= 0 && c < Ce) o[(((size_t)q * K64 + mi) * 64 + lane) * 16 + i] = w[(size_t)tap * Ce + c];
          }
      v4i* d64;
      if ((rc = upload(m, w64, &d64))) return rc;
      a.wd64 = d64;
      std::vector<long> w64c((size_t)(Cp / 16) * 64, 0);      // one byte per operand: the diagonal byte of each lane
      int8_t* oc = (int8_t*)w64c.data();
      for (int q = 0; q < Cp / 16; q++)
        for (int mi = 0; mi < K64; mi++)
          for (int lane = 0; lane < 64; lane++)
            oc[((size_t)q * 64 + lane) * 8 + mi] = o[(((size_t)q * K64 + mi) * 64 + lane) * 16 + (lane & 15)];
      long* d64c;
      if ((rc = upload(m, w64c, &d64c))) return rc;
      a.wd64c = d64c;
    }
    a.zd = tdout.zero_point; a.lod = dop.act_min; a.hid = dop.act_max;
    a.rqd = make_rq(a.zd, a.lod, a.hid, conv_kb(m, dop));
  }
  // project weights re-packed with K padded to Cp
  {
    const int8_t* w = (const int8_t*)(m->blob.data() + pop.w_off);
    std::vector<v4i> wp;
    // the kernel instantiation covers nbp = {1,2,3,5} blocks and reads the weights of all of them: allocate (zero
    // rows) up to nbp, otherwise a 4-block layer (e.g. Cout = 208 in Lite2) reads past the packed array
    const int nbp_alloc = ps.NB <= 3 ? ps.NB : 5;
    pack_weights64(w, tout.c, Ce, Cp / 64, nbp_alloc, wp);
    v4i* dwp;
    int rc;
    if ((rc = upload(m, wp, &dwp))) return rc;
    a.wp = dwp; a.bp = ps.bias; a.mp = ps.mult; a.KSp = Cp / 64;
    a.zo = tout.zero_point; a.lop = pop.act_min; a.hip = pop.act_max;
    a.rqp = make_rq(a.zo, a.lop, a.hip);
  }
  if (a_op >= 0) {   // fuse_plan guarantees inputs = (project output, block input)
    a.has_res = 1;
    a.resq = m->op_steps[a_op].addq;
  }
  if (sum_op >= 0) {
    a.n_src = ns->n;
    for (int j = 0; j < ns->n; j++) {
      const TensorRec& ts = m->tensors[ns->tensor[j]];
      s.src_tensor[j] = ns->tensor[j];
      a.sh[j] = ts.h; a.sw[j] = ts.w; a.smode[j] = ns->mode[j];
      if (ns->mode[j] == 2) { a.spt[j] = m->ops[ns->rs_op[j]].pad_t; a.spl[j] = m->ops[ns->rs_op[j]].pad_l; }
    }
    a.sumq = m->op_steps[sum_op].addq;   // resize / max-pool outputs keep their input's quantisation, so the parameters hold for the absorbed sources
    a.chain = 0;
    if (ns->pre_add >= 0) {   // two chained binary ADDs
      a.chain = ns->chain;
      a.preq = m->op_steps[ns->pre_add].addq;
    }
  }
  s.nbp = ps.NB <= 3 ? ps.NB : 5;
  const int TXp = (a.TX + 3) & ~3;
  const int NPh = ((TXp - 1) * dop.stride + dop.k) * ((a.TY - 1) * dop.stride + dop.k);
  s.lds_bytes = ((NPh * a.T0S + 15) & ~15) + (expand ? NPh * FB_EST : 0) + 64 * FB_DST;
  if (!expand && a.nchunks == 1 && s.nbp <= 2) s.lds_bytes += s.nbp * (4096 + 512);   // projection weights + bias / multipliers staged in LDS
  // accounting = compulsory traffic of the constituent graph ops
  std::vector<int> parts{e_op, d_op, p_op, a_op, sum_op};
  if (ns) for (int j = 0; j < ns->n; j++) parts.push_back(ns->rs_op[j]);
  if (ns) parts.push_back(ns->pre_add);
  for (int oi : parts)
    if (oi >= 0) {
      s.alg_bytes_per_frame += m->op_steps[oi].alg_bytes_per_frame;
      s.weight_bytes += m->op_steps[oi].weight_bytes;
      s.macs_per_frame += m->op_steps[oi].macs_per_frame;
    }
  if (expand && Ce % 48 == 0 && Ce % 64 != 0) {
    // 48-channel chunking of the same block (fused_block.h, template NT = 3): no padded channels
    const OpRec& eop = m->ops[e_op];
    const int8_t* we = (const int8_t*)(m->blob.data() + eop.w_off);
    const int8_t* wd = (const int8_t*)(m->blob.data() + dop.w_off);
    const int8_t* wpj = (const int8_t*)(m->blob.data() + pop.w_off);
    const int K = tin.c, KSe = a.KSe, nch3 = Ce / 48, KT = (kk + 1) / 2;
    std::vector<long> we3((size_t)nch3 * KSe * 3 * 64, 0), wdm3((size_t)nch3 * 3 * KT * 64, 0);
    std::vector<v4i> wp3;
    int8_t* o = (int8_t*)we3.data();
    for (int c = 0; c < nch3; c++)
      for (int ks = 0; ks < KSe; ks++)
        for (int t = 0; t < 3; t++)
          for (int lane = 0; lane < 64; lane++) {
            const int i = lane & 15, kg = lane >> 4, co = 48 * c + 12 * (i >> 2) + 4 * t + (i & 3);
            for (int j = 0; j < 8; j++) {
              const int k = 32 * ks + 8 * kg + j;
              o[((((size_t)(c * KSe + ks) * 3 + t) * 64 + lane) * 8) + j] = k < K ? we[(size_t)co * K + k] : 0;
            }
          }
    int8_t* od = (int8_t*)wdm3.data();
    for (int c = 0; c < nch3; c++)
      for (int cg = 0; cg < 3; cg++)
        for (int mi = 0; mi < KT; mi++)
          for (int lane = 0; lane < 64; lane++) {
            const int i = lane & 15, g = lane >> 4, ch = 48 * c + 16 * cg + i, tap = 2 * mi + (g >> 1);
            for (int j = 0; j < 8; j++)
              od[((((size_t)(c * 3 + cg) * KT + mi) * 64 + lane) * 8) + j] = (tap < kk && 8 * (g & 1) + j == i) ? wd[(size_t)tap * Ce + ch] : 0;
          }
    std::vector<int> kmap((size_t)nch3 * 64, -1);
    for (int c = 0; c < nch3; c++)
      for (int q = 0; q < 48; q++) kmap[(size_t)c * 64 + q] = 48 * c + q;
    pack_weights64(wpj, tout.c, Ce, nch3, ps.NB <= 3 ? ps.NB : 5, wp3, &kmap);
    long *d1, *d2;
    v4i* d3;
    int rc;
    if ((rc = upload(m, we3, &d1)) || (rc = upload(m, wdm3, &d2)) || (rc = upload(m, wp3, &d3))) return rc;
    a.we3 = d1; a.wdm3 = d2; a.wp3 = d3; a.nch3 = nch3; a.KSp3 = nch3;
  }
  if (expand && tin.h * tin.w <= 400 && tout.h * tout.w <= 400) {
    // whole-image kernel (image_block.h): one contiguous record per 64-channel chunk
    const OpRec& eop = m->ops[e_op];
    const int8_t* we = (const int8_t*)(m->blob.data() + eop.w_off);
    const int32_t* bqe = (const int32_t*)(m->blob.data() + eop.b_off);
    const float* mue = (const float*)(m->blob.data() + eop.m_off);
    const int8_t* wd = (const int8_t*)(m->blob.data() + dop.w_off);
    const int32_t* bqd = (const int32_t*)(m->blob.data() + dop.b_off);
    const float* mud = (const float*)(m->blob.data() + dop.m_off);
    const int8_t* wpj = (const int8_t*)(m->blob.data() + pop.w_off);
    const int KSe = a.KSe, NB = (tout.c + 63) / 64, nch = Cp / 64;
    std::vector<long> pe, pp;
    pack_weights(we, Ce, tin.c, KSe, nch, nullptr, pe);
    pack_weights(wpj, tout.c, Ce, Cp / 32, NB, nullptr, pp);
    ImageBundle ib;
    ib.o_be = KSe * 2048; ib.o_me = ib.o_be + 256; ib.o_dw = ib.o_me + 256;
    ib.o_bd = ib.o_dw + ((kk * 64 + 15) & ~15); ib.o_md = ib.o_bd + 256; ib.o_wp = ib.o_md + 256;
    ib.bytes = ib.o_wp + NB * 4096;
    std::vector<unsigned char> rec((size_t)nch * ib.bytes, 0);
    for (int c = 0; c < nch; c++) {
      unsigned char* R = rec.data() + (size_t)c * ib.bytes;
      memcpy(R, pe.data() + (size_t)c * KSe * 4 * 64, (size_t)KSe * 2048);
      int* be = (int*)(R + ib.o_be); float* me = (float*)(R + ib.o_me);
      int* bd = (int*)(R + ib.o_bd); float* md = (float*)(R + ib.o_md);
      for (int i = 0; i < 64; i++) {
        const int ch = 64 * c + i;
        if (ch >= Ce) continue;
        long swe = 0, swd = 0;
        for (int k = 0; k < tin.c; k++) swe += we[(size_t)ch * tin.c + k];
        for (int t = 0; t < kk; t++) { swd += wd[(size_t)t * Ce + ch]; R[ib.o_dw + t * 64 + i] = (unsigned char)wd[(size_t)t * Ce + ch]; }
        be[i] = (int)((long)bqe[ch] - (long)tin.zero_point * swe);
        me[i] = mue[ch];
        bd[i] = (int)((long)bqd[ch] - (long)tdin.zero_point * swd);
        md[i] = mud[ch];
      }
      for (int nb = 0; nb < NB; nb++)
        for (int k2 = 0; k2 < 2; k2++)
          memcpy(R + ib.o_wp + (size_t)((nb * 2 + k2) * 4) * 512, pp.data() + ((size_t)(nb * (Cp / 32) + 2 * c + k2) * 4) * 64, 4 * 512);
    }
    unsigned char* drec;
    int rc;
    if ((rc = upload(m, rec, &drec))) return rc;
    ib.data = drec;
    s.ib = ib;
  }
  *out = s;
  return VBT_OK;
}

// the alternative of a group that runs it as ONE tile-kernel launch (fused_block.h), or nullptr
static const Alt* tile_alt(const Group& g, int family) {
  for (int i = (int)g.alts.size() - 1; i >= 0; i--)
    if (g.alts[i].steps.size() == 1 && g.alts[i].steps[0].family == family) return &g.alts[i];
  return nullptr;
}
static const Alt* band_alt(const Group& g) {
  for (int i = (int)g.alts.size() - 1; i >= 0; i--)
    if (g.alts[i].steps.size() == 1 && g.alts[i].steps[0].family == F_BAND && g.alts[i].steps[0].members.empty()) return &g.alts[i];
  return nullptr;
}

// ---- SeparableConv / BiFPN node on row bands (band_block.h) ----
static bool band_ok(const vbt_model* m, int d_op, int p_op) {
  if (m->flags & VBT_MODEL_NO_BAND) return false;
  if (const char* ns = getenv("VBT_SUBSTREAMS")) if (atoi(ns) > 1) return false;   // the problem list holds whole-batch pointers
  const OpRec& d = m->ops[d_op];
  const OpRec& p = m->ops[p_op];
  const TensorRec& ti = m->tensors[d.inputs[0]];
  const TensorRec& to = m->tensors[p.output];
  return d.k == 3 && d.stride == 1 && d.pad_t == 1 && d.pad_l == 1 && ti.c % 8 == 0 && ti.c >= 16 && ti.c <= 128 && to.c <= 128 && to.h == ti.h &&
         to.w == ti.w && ti.w <= 160;
}
static int band_lds(const BandArgs& a) {
  const int NT = (a.Cout + 15) / 16;
  return (a.rows + 2) * (a.W + 2) * a.CS + (((a.rows * a.W + 15) >> 4) << 4) * a.CS + NT * a.KS * 1024 + BD_WP_TAIL;
}
static int make_band(vbt_model* m, int d_op, int p_op, int sum_op, const NodeSrc* ns, Step* out) {
  const OpRec& dop = m->ops[d_op];
  const OpRec& pop = m->ops[p_op];
  const TensorRec& tin = m->tensors[dop.inputs[0]];
  const TensorRec& td = m->tensors[dop.output];
  const TensorRec& to = m->tensors[pop.output];
  Step s;
  s.family = F_BAND;
  s.op = p_op;
  s.d_op = d_op;
  s.p_op = p_op;
  s.sum_op = sum_op;
  BandArgs& a = s.bd_args;
  memset(&a, 0, sizeof(a));
  a.H = tin.h; a.W = tin.w; a.Cout = to.c;
  const int C = tin.c;
  a.C = C;
  a.CS = ((C + 15) / 16 | 1) * 16;
  a.NCG = (C + 15) / 16;
  a.KS = (C + 63) / 64;
  const int NT = (to.c + 15) / 16;
  // pixels per band: 320 fills a workgroup's sixteen waves with units (a batch of 64 brings enough bands to fill the GPU); a small
  // batch leaves most CUs idle, so there a map is cut into more, shorter bands (latency of one band ~ its pixel groups per wave)
  static const int band_px_env = getenv("VBT_BAND_PX") ? atoi(getenv("VBT_BAND_PX")) : 0;
  const int band_px = band_px_env > 0 ? band_px_env : (m->max_batch <= 8 ? 64 : 320);   // (batch 1 / 8, four forwards in flight: 64 px +3-5 % over 320, tools/band_px_sweep.sh)
  const int nb = std::max(1, (tin.h * tin.w + band_px - 1) / band_px);
  a.rows = (tin.h + nb - 1) / nb;
  a.nbands = (tin.h + a.rows - 1) / a.rows;
  a.zx4 = (unsigned)(tin.zero_point & 255) * 0x01010101u;
  const int8_t* wd = (const int8_t*)(m->blob.data() + dop.w_off);
  const int32_t* bqd = (const int32_t*)(m->blob.data() + dop.b_off);
  const float* mud = (const float*)(m->blob.data() + dop.m_off);
  const int8_t* wpj = (const int8_t*)(m->blob.data() + pop.w_off);
  std::vector<v4i> pd((size_t)a.NCG * 3 * 64, (v4i){0, 0, 0, 0}), pp((size_t)NT * a.KS * 64, (v4i){0, 0, 0, 0});
  int8_t* od = (int8_t*)pd.data();
  for (int cg = 0; cg < a.NCG; cg++)
    for (int mi = 0; mi < 3; mi++)
      for (int lane = 0; lane < 64; lane++) {
        const int i = lane & 15, g = lane >> 4, ch = 16 * cg + i, tap = 4 * mi + g;
        for (int j = 0; j < 16; j++) od[(((size_t)(cg * 3 + mi) * 64 + lane) * 16) + j] = (tap < 9 && j == i && ch < C) ? wd[(size_t)tap * C + ch] : 0;
      }
  int8_t* op_ = (int8_t*)pp.data();
  for (int t = 0; t < NT; t++)
    for (int ks = 0; ks < a.KS; ks++)
      for (int lane = 0; lane < 64; lane++) {
        const int i = lane & 15, g = lane >> 4, co = 16 * t + i;
        for (int j = 0; j < 16; j++) {
          const int k = 64 * ks + 16 * g + j;
          op_[((((size_t)t * a.KS + ks) * 64 + lane) * 16) + j] = (co < to.c && k < C) ? wpj[(size_t)co * C + k] : 0;
        }
      }
  std::vector<int> bd(a.NCG * 16, 0);
  std::vector<float> md(a.NCG * 16, 0.0f);
  for (int ch = 0; ch < C; ch++) {
    long sw = 0;
    for (int t = 0; t < 9; t++) sw += wd[(size_t)t * C + ch];
    bd[ch] = (int)((long)bqd[ch] - (long)tin.zero_point * sw);
    md[ch] = mud[ch];
  }
  v4i *dpd, *dpp;
  int* dbd;
  float* dmd;
  int rc;
  if ((rc = upload(m, pd, &dpd)) || (rc = upload(m, pp, &dpp)) || (rc = upload(m, bd, &dbd)) || (rc = upload(m, md, &dmd))) return rc;
  a.wd = dpd; a.wp = dpp; a.bd = dbd; a.md = dmd;
  a.bp = m->op_steps[p_op].bias;   // folded with the depthwise output's zero point, padded to 64
  a.mp = m->op_steps[p_op].mult;
  a.rqd = make_rq(td.zero_point, dop.act_min, dop.act_max, conv_kb(m, dop));
  a.rqp = make_rq(to.zero_point, pop.act_min, pop.act_max, conv_kb(m, pop));
  a.x = m->tptr[dop.inputs[0]];
  a.out = m->tptr[pop.output];
  std::vector<int> parts{d_op, p_op, sum_op};
  if (sum_op >= 0) {
    a.n_src = ns->n;
    a.x = nullptr;
    for (int j = 0; j < ns->n; j++) {
      const TensorRec& ts = m->tensors[ns->tensor[j]];
      a.src[j] = m->tptr[ns->tensor[j]];
      a.sh[j] = ts.h; a.sw[j] = ts.w; a.smode[j] = ns->mode[j];
      if (ns->mode[j] == 2) { a.spt[j] = m->ops[ns->rs_op[j]].pad_t; a.spl[j] = m->ops[ns->rs_op[j]].pad_l; }
      parts.push_back(ns->rs_op[j]);
    }
    a.sumq = m->op_steps[sum_op].addq;
    a.chain = 0;
    if (ns->pre_add >= 0) { a.chain = ns->chain; a.preq = m->op_steps[ns->pre_add].addq; parts.push_back(ns->pre_add); }
  }
  s.band_tiles = a.nbands;
  s.lds_bytes = band_lds(a);
  std::vector<BandArgs> one{a};
  if ((rc = upload(m, one, &s.d_band))) return rc;
  s.ba = a;
  for (int oi : parts)
    if (oi >= 0) {
      s.alg_bytes_per_frame += m->op_steps[oi].alg_bytes_per_frame;
      s.weight_bytes += m->op_steps[oi].weight_bytes;
      s.macs_per_frame += m->op_steps[oi].macs_per_frame;
    }
  *out = s;
  return VBT_OK;
}

// The box / class heads run the same SeparableConv chain on 5 pyramid levels (x 2 heads): layer j of every
// chain is independent of layer j of the others, so all of them go out as ONE grid (fused_block_multi_kernel).
static int batch_heads(vbt_model* m) {
  struct Info { int gi, chain, depth; };
  std::vector<Info> heads;
  std::map<int, std::pair<int, int>> by_tensor;  // output tensor -> (chain, depth)
  int nchains = 0, first = -1;
  for (int gi = 0; gi < (int)m->groups.size(); gi++) {
    const Group& g = m->groups[gi];
    const Alt* fap = tile_alt(g, F_SEPCONV);
    if (!fap) continue;
    const Step& st = fap->steps[0];
    if (m->ops[st.d_op].level < 0) continue;
    int tin = m->ops[st.d_op].inputs[0], tout = m->ops[st.p_op].output;
    auto it = by_tensor.find(tin);
    int chain = it == by_tensor.end() ? nchains++ : it->second.first;
    int depth = it == by_tensor.end() ? 0 : it->second.second + 1;
    by_tensor[tout] = {chain, depth};
    heads.push_back({gi, chain, depth});
    if (first < 0) first = gi;
  }
  if (heads.size() < 2) return VBT_OK;
  int maxd = 0;
  for (auto& h : heads) maxd = std::max(maxd, h.depth);
  // Depth d may only be merged if every deeper layer is merged too: an unmerged successor sits right behind its
  // own predecessor in list order, i.e. ahead of the merged launch that would produce its input.
  auto members_of = [&](int d) {
    std::vector<int> mem;
    for (auto& h : heads)
      if (h.depth == d) mem.push_back(h.gi);
    return mem;
  };
  auto mergeable = [&](const std::vector<int>& mem) {
    if (mem.size() < 2 || mem.size() > 12) return false;
    const Step& s0 = tile_alt(m->groups[mem[0]], F_SEPCONV)->steps[0];
    if (!((s0.nbp == 1 || s0.nbp == 2) && m->ops[s0.d_op].k == 3 && m->ops[s0.d_op].stride == 1)) return false;  // the instantiations built below
    for (int gi : mem) {
      const Step& st = tile_alt(m->groups[gi], F_SEPCONV)->steps[0];
      if (st.nbp != s0.nbp || m->ops[st.d_op].k != 3 || m->ops[st.d_op].stride != 1) return false;
    }
    return true;
  };
  int d0 = maxd + 1;
  while (d0 > 0 && mergeable(members_of(d0 - 1))) d0--;
  if (d0 > maxd) return VBT_OK;
  std::map<int, Group> at;  // position (index of the last member) -> merged group
  std::vector<char> consumed(m->groups.size(), 0);
  for (int d = d0; d <= maxd; d++) {
    std::vector<int> mem = members_of(d);
    const Step& s0 = tile_alt(m->groups[mem[0]], F_SEPCONV)->steps[0];
    Group g;
    Alt unf, each, multi, bandm;
    Step ms, bs;
    bs.family = F_BAND;
    bs.op = s0.op;
    bs.d_op = s0.d_op;
    std::vector<BandArgs> bargs;
    bool all_band = true;
    ms.family = F_MULTI;
    ms.op = s0.op;
    ms.d_op = s0.d_op;
    ms.nbp = s0.nbp;
    std::vector<FusedArgs> hargs;
    int last = 0;
    for (int gi : mem) {
      const Group& src = m->groups[gi];
      for (const Step& st : src.alts[0].steps) unf.steps.push_back(st);
      const Alt* ta = tile_alt(src, F_SEPCONV);
      const Step& fs = ta->steps[0];
      each.steps.push_back(fs);
      for (int t : ta->hidden) { each.hidden.push_back(t); multi.hidden.push_back(t); bandm.hidden.push_back(t); }
      const Alt* ba = band_alt(src);
      if (ba && ba->steps[0].bd_args.n_src == 0) {   // (the multi-problem band kernels are built without the node-sum path)
        Step b1 = ba->steps[0];
        {   // the head grid runs 8-wave workgroups on shorter bands (band_block.h)
          BandArgs& ha = b1.bd_args;
          const int nbh = std::max(1, (ha.H * ha.W + BD_HEAD_MAXPX - 1) / BD_HEAD_MAXPX);
          ha.rows = (ha.H + nbh - 1) / nbh;
          ha.nbands = (ha.H + ha.rows - 1) / ha.rows;
          b1.band_tiles = ha.nbands;
          b1.lds_bytes = band_lds(ha);
          b1.ba = ha;
        }
        bs.members.push_back(b1);
        bargs.push_back(b1.bd_args);
        bs.lds_bytes = std::max(bs.lds_bytes, b1.lds_bytes);
        bs.alg_bytes_per_frame += b1.alg_bytes_per_frame;
        bs.weight_bytes += b1.weight_bytes;
        bs.macs_per_frame += b1.macs_per_frame;
      } else {
        all_band = false;
      }
      ms.members.push_back(fs);
      ms.lds_bytes = std::max(ms.lds_bytes, fs.lds_bytes);
      ms.alg_bytes_per_frame += fs.alg_bytes_per_frame;
      ms.weight_bytes += fs.weight_bytes;
      ms.macs_per_frame += fs.macs_per_frame;
      FusedArgs a = fs.fa;
      a.x = m->tptr[m->ops[fs.d_op].inputs[0]];
      a.out = m->tptr[m->ops[fs.op].output];
      hargs.push_back(a);
      consumed[gi] = 1;
      last = std::max(last, gi);
    }
    int rc = upload(m, hargs, &ms.d_multi);
    if (rc) return rc;
    multi.steps.push_back(ms);
    g.alts.push_back(unf);
    g.alts.push_back(each);
    g.alts.push_back(multi);
    g.chosen = 2;
    if (all_band && bargs.size() <= 12) {   // the same layer of every chain on row bands, one grid
      int rcb = upload(m, bargs, &bs.d_band);
      if (rcb) return rcb;
      bandm.steps.push_back(bs);
      g.alts.push_back(bandm);
      g.chosen = 3;
    }
    at[last] = g;
  }
  std::vector<Group> out;
  for (int gi = 0; gi < (int)m->groups.size(); gi++) {
    if (!consumed[gi]) out.push_back(m->groups[gi]);
    auto it = at.find(gi);
    if (it != at.end()) out.push_back(it->second);
  }
  m->groups.swap(out);
  return VBT_OK;
}

// ---- network entry: STEM(3x3/2, 3 -> 32) -> DW(3x3/1) -> PW(32 -> <=16) as one kernel (stem_block.h) ----
static bool stem_block_ok(const vbt_model* m, int si, const std::vector<int>& consumers) {
  const int no = (int)m->ops.size();
  if (si + 2 >= no) return false;
  const OpRec& st = m->ops[si];
  const OpRec& d = m->ops[si + 1];
  const OpRec& p = m->ops[si + 2];
  if (st.type != OP_STEM || d.type != OP_DW || p.type != OP_PW) return false;
  const TensorRec& ti = m->tensors[st.inputs[0]];
  const TensorRec& ts = m->tensors[st.output];
  const TensorRec& td = m->tensors[d.output];
  const TensorRec& to = m->tensors[p.output];
  return st.k == 3 && st.stride == 2 && ti.c == 3 && ts.c == 32 && ti.w % 4 == 0 && d.inputs[0] == st.output && consumers[st.output] == 1 &&
         d.k == 3 && d.stride == 1 && d.pad_t == 1 && d.pad_l == 1 && td.c == 32 && p.inputs[0] == d.output && consumers[d.output] == 1 &&
         to.c <= 16 && to.c % 4 == 0;
}

static int make_stem_block(vbt_model* m, int si, Step* out) {
  const OpRec& st = m->ops[si];
  const OpRec& d = m->ops[si + 1];
  const OpRec& p = m->ops[si + 2];
  const TensorRec& ti = m->tensors[st.inputs[0]];
  const TensorRec& ts = m->tensors[st.output];
  const TensorRec& td = m->tensors[d.output];
  const TensorRec& to = m->tensors[p.output];
  Step s;
  s.op = si + 2;
  s.family = F_STEMBLK;
  s.e_op = si;
  s.d_op = si + 1;
  s.p_op = si + 2;
  StemBlockArgs& a = s.sb;
  a.frames = nullptr; a.out = nullptr;
  a.H = ti.h; a.W = ti.w; a.SH = ts.h; a.SW = ts.w; a.Cout = to.c;
  a.spad_t = st.pad_t; a.spad_l = st.pad_l;
  a.tiles_x = (ts.w + 15) / 16; a.tiles_y = (ts.h + 15) / 16;
  a.in_pad4 = (unsigned)((ti.zero_point + 128) & 255) * 0x01010101u;
  a.zs4 = (unsigned)(ts.zero_point & 255) * 0x01010101u;
  a.rqs = make_rq(ts.zero_point, st.act_min, st.act_max, conv_kb(m, st));
  a.rqd = make_rq(td.zero_point, d.act_min, d.act_max, conv_kb(m, d));
  a.rqp = make_rq(to.zero_point, p.act_min, p.act_max, conv_kb(m, p));
  int rc;
  {  // stem: K index 8kg + j -> kernel row kg, byte j of its 9 (kg < 3); (row j, byte 8) for kg == 3, j < 3
    const int8_t* w = (const int8_t*)(m->blob.data() + st.w_off);
    const int32_t* bq = (const int32_t*)(m->blob.data() + st.b_off);
    const float* mu = (const float*)(m->blob.data() + st.m_off);
    std::vector<long> ws(2 * 64, 0);
    int8_t* o = (int8_t*)ws.data();
    for (int t = 0; t < 2; t++)
      for (int lane = 0; lane < 64; lane++) {
        const int i = lane & 15, kg = lane >> 4, co = 8 * (i >> 2) + 4 * t + (i & 3);
        for (int j = 0; j < 8; j++) {
          const int f = kg < 3 ? kg * 9 + j : (j < 3 ? j * 9 + 8 : -1);
          o[((size_t)t * 64 + lane) * 8 + j] = f >= 0 ? w[(size_t)co * 27 + f] : 0;
        }
      }
    std::vector<int> bs(32);
    std::vector<float> ms(mu, mu + 32);
    for (int c = 0; c < 32; c++) {
      long sw = 0;
      for (int k = 0; k < 27; k++) sw += w[(size_t)c * 27 + k];
      bs[c] = (int)((long)bq[c] - (long)ti.zero_point * sw);
    }
    long* dws; int* dbs; float* dms;
    if ((rc = upload(m, ws, &dws)) || (rc = upload(m, bs, &dbs)) || (rc = upload(m, ms, &dms))) return rc;
    a.ws = dws; a.bs = dbs; a.ms = dms;
  }
  const Step& ds = m->op_steps[si + 1];  // matrix-pipe depthwise bias / multipliers of the dw op
  a.bdm = ds.bdm; a.mdm = ds.mdm;
  {  // depthwise weights in the 16x16x64 form: [cg][m][lane] x 16 B, tap (row m, column g), diagonal byte i
    const int8_t* w = (const int8_t*)(m->blob.data() + d.w_off);
    std::vector<v4i> w64(2 * 3 * 64, (v4i){0, 0, 0, 0});
    int8_t* o = (int8_t*)w64.data();
    for (int cg = 0; cg < 2; cg++)
      for (int mi = 0; mi < 3; mi++)
        for (int lane = 0; lane < 64; lane++) {
          const int i = lane & 15, g = lane >> 4;
          if (g < 3) o[(((size_t)cg * 3 + mi) * 64 + lane) * 16 + i] = w[(size_t)(mi * 3 + g) * 32 + 16 * cg + i];
        }
    v4i* d64;
    if ((rc = upload(m, w64, &d64))) return rc;
    a.wd64 = d64;
  }
  {  // project: row i = cout i, K = 32
    const int8_t* w = (const int8_t*)(m->blob.data() + p.w_off);
    const int32_t* bq = (const int32_t*)(m->blob.data() + p.b_off);
    const float* mu = (const float*)(m->blob.data() + p.m_off);
    std::vector<long> wp(64, 0);
    int8_t* o = (int8_t*)wp.data();
    std::vector<int> bp(16, 0);
    std::vector<float> mp(16, 0.0f);
    for (int lane = 0; lane < 64; lane++) {
      const int i = lane & 15, kg = lane >> 4;
      for (int j = 0; j < 8; j++) o[(size_t)lane * 8 + j] = i < to.c ? w[(size_t)i * 32 + 8 * kg + j] : 0;
    }
    for (int c = 0; c < to.c; c++) {
      long sw = 0;
      for (int k = 0; k < 32; k++) sw += w[(size_t)c * 32 + k];
      bp[c] = (int)((long)bq[c] - (long)td.zero_point * sw);
      mp[c] = mu[c];
    }
    long* dwp; int* dbp; float* dmp;
    if ((rc = upload(m, wp, &dwp)) || (rc = upload(m, bp, &dbp)) || (rc = upload(m, mp, &dmp))) return rc;
    a.wp = dwp; a.bp = dbp; a.mp = dmp;
  }
  for (int k = 0; k < 3; k++) {
    const Step& os = m->op_steps[si + k];
    s.alg_bytes_per_frame += os.alg_bytes_per_frame;
    s.weight_bytes += os.weight_bytes;
    s.macs_per_frame += os.macs_per_frame;
  }
  *out = s;
  return VBT_OK;
}

// ---- expand + depthwise on whole images (expdw_block.h) ----
// LDS of the whole-image / row-band expand + depthwise kernel for `nbands` bands: T0 (input rows of the tallest band) | E | D
struct ExpDwGeom { int nbands, brows, t0_bytes, e_bytes, d_bytes; };
static ExpDwGeom expdw_geom(int H, int W, int OH, int OW, int k, int stride, int pad_t, int pad_l, int T0S, int nbands) {
  ExpDwGeom g;
  g.nbands = nbands;
  g.brows = (OH + nbands - 1) / nbands;
  g.nbands = (OH + g.brows - 1) / g.brows;
  const int PW = std::max((OW - 1) * stride + k, pad_l + W);
  int in_rows = 0;
  for (int b = 0; b < g.nbands; b++) {
    const int oy0 = b * g.brows, oy1 = std::min(oy0 + g.brows, OH);
    const int PHb = (oy1 - oy0 - 1) * stride + k;
    const int lo = std::max(oy0 * stride - pad_t, 0), hi = std::min(oy0 * stride - pad_t + PHb, H);
    in_rows = std::max(in_rows, hi - lo);
  }
  const int PHmax = (g.brows - 1) * stride + k;
  g.t0_bytes = (in_rows * W * T0S + 15) & ~15;
  g.e_bytes = PHmax * PW * XD_EST;
  g.d_bytes = ((g.brows * OW + 15) / 16) * 16 * XD_EST;
  return g;
}
// whole image when the map has at most 400 pixels and fits (every block of Lite0: the round-2 plan is unchanged); otherwise the
// fewest bands whose workgroup stays below VBT_XD_BAND_LDS bytes (default 96 KB: one and a half workgroups' worth of a CU)
static ExpDwGeom expdw_choose(int H, int W, int OH, int OW, int k, int stride, int pad_t, int pad_l, int T0S) {
  static const int budget = getenv("VBT_XD_BAND_LDS") ? atoi(getenv("VBT_XD_BAND_LDS")) : 96 * 1024;
  ExpDwGeom g = expdw_geom(H, W, OH, OW, k, stride, pad_t, pad_l, T0S, 1);
  if (H * W <= 400 && OH * OW <= 400 && g.t0_bytes + g.e_bytes + g.d_bytes <= 160 * 1024) return g;
  for (int nb = 2; nb <= OH; nb++) {
    g = expdw_geom(H, W, OH, OW, k, stride, pad_t, pad_l, T0S, nb);
    if (g.t0_bytes + g.e_bytes + g.d_bytes <= budget) return g;
  }
  return g;
}
static bool expdw_ok(const vbt_model* m, int e_op, int d_op) {
  const OpRec& e = m->ops[e_op];
  const OpRec& d = m->ops[d_op];
  const TensorRec& tin = m->tensors[e.inputs[0]];
  const TensorRec& tout = m->tensors[d.output];
  const int KS64 = (tin.c + 63) / 64;
  const bool shape = (d.k == 3 && d.stride == 1) || (d.k == 5 && (d.stride == 1 || d.stride == 2));
  if (!(shape && tin.h * tin.w <= 1024 && tin.c % 8 == 0 && tout.c % 16 == 0 && KS64 >= 2 && KS64 <= 4)) return false;
  const ExpDwGeom g = expdw_choose(tin.h, tin.w, tout.h, tout.w, d.k, d.stride, d.pad_t, d.pad_l, ((tin.c + 15) / 16 | 1) * 16);
  return g.t0_bytes + g.e_bytes + g.d_bytes <= 160 * 1024;
}
static int make_expdw2(vbt_model* m, int e_op, int d_op, Step* s, const std::vector<v4i>& pe, const std::vector<int>& be, const std::vector<float>& me,
                       const std::vector<int>& bd, const std::vector<float>& md);
static int make_expdw(vbt_model* m, int e_op, int d_op, Step* out) {
  const OpRec& eop = m->ops[e_op];
  const OpRec& dop = m->ops[d_op];
  const TensorRec& tin = m->tensors[eop.inputs[0]];
  const TensorRec& te = m->tensors[eop.output];
  const TensorRec& td = m->tensors[dop.output];
  Step s;
  s.family = F_EXPDW;
  s.op = d_op;
  s.e_op = e_op;
  s.d_op = d_op;
  ExpDwArgs& a = s.xd;
  memset(&a, 0, sizeof(a));
  a.H = tin.h; a.W = tin.w; a.Cin = tin.c; a.OH = td.h; a.OW = td.w; a.Ce = te.c;
  a.pad_t = dop.pad_t; a.pad_l = dop.pad_l;
  a.PH = std::max((a.OH - 1) * dop.stride + dop.k, a.pad_t + a.H);
  a.PW = std::max((a.OW - 1) * dop.stride + dop.k, a.pad_l + a.W);
  const int KS64 = (tin.c + 63) / 64, K = tin.c, Ce = te.c, nch = (Ce + 63) / 64, kk = dop.k * dop.k, KT = (kk + 3) / 4;
  // input rows hold the real channels (16-byte granules), not the K padding: the B-operand reads of the padded K run into the
  // next pixel's bytes and meet zero weights (the last pixel's into the E tile).  An odd number of 16-byte granules per row
  // makes the 16-pixel b128 reads bank-conflict-free (80, 112, 208 bytes for 80, 112, 192 channels; was 160 / 224): the
  // 20x20 blocks free 32 KB of LDS per CU for the other forwards in flight.
  a.T0S = ((tin.c + 15) / 16 | 1) * 16;
  a.nchunks = nch;
  a.cpw = 1;
  const ExpDwGeom geo = expdw_choose(a.H, a.W, a.OH, a.OW, dop.k, dop.stride, a.pad_t, a.pad_l, a.T0S);
  a.nbands = geo.nbands; a.brows = geo.brows; a.t0_bytes = geo.t0_bytes; a.e_bytes = geo.e_bytes;
  const int8_t* we = (const int8_t*)(m->blob.data() + eop.w_off);
  const int32_t* bqe = (const int32_t*)(m->blob.data() + eop.b_off);
  const float* mue = (const float*)(m->blob.data() + eop.m_off);
  const int8_t* wd = (const int8_t*)(m->blob.data() + dop.w_off);
  const int32_t* bqd = (const int32_t*)(m->blob.data() + dop.b_off);
  const float* mud = (const float*)(m->blob.data() + dop.m_off);
  std::vector<v4i> pe((size_t)nch * KS64 * 4 * 64, (v4i){0, 0, 0, 0});
  std::vector<int> be(nch * 64, 0), bd(nch * 64, 0);
  std::vector<float> me(nch * 64, 0.0f), md(nch * 64, 0.0f);
  int8_t* o = (int8_t*)pe.data();
  for (int c = 0; c < nch; c++)
    for (int ks = 0; ks < KS64; ks++)
      for (int t = 0; t < 4; t++)
        for (int lane = 0; lane < 64; lane++) {
          const int i = lane & 15, g = lane >> 4, ch = 64 * c + 16 * t + i;
          for (int j = 0; j < 16; j++) {
            const int k = 64 * ks + 16 * g + j;
            o[((((size_t)(c * KS64 + ks) * 4 + t) * 64 + lane) * 16) + j] = (ch < Ce && k < K) ? we[(size_t)ch * K + k] : 0;
          }
        }
  std::vector<long> pdc((size_t)nch * 4 * 64, 0);
  int8_t* od = (int8_t*)pdc.data();
  for (int c = 0; c < nch; c++)
    for (int cg = 0; cg < 4; cg++)
      for (int mi = 0; mi < KT; mi++)
        for (int lane = 0; lane < 64; lane++) {
          const int i = lane & 15, g = lane >> 4, ch = 64 * c + 16 * cg + i, tap = 4 * mi + g;
          od[(((size_t)(c * 4 + cg)) * 64 + lane) * 8 + mi] = (tap < kk && ch < Ce) ? wd[(size_t)tap * Ce + ch] : 0;
        }
  for (int ch = 0; ch < Ce; ch++) {
    long swe = 0, swd = 0;
    for (int k = 0; k < K; k++) swe += we[(size_t)ch * K + k];
    for (int t = 0; t < kk; t++) swd += wd[(size_t)t * Ce + ch];
    be[ch] = (int)((long)bqe[ch] - (long)tin.zero_point * swe);
    me[ch] = mue[ch];
    bd[ch] = (int)((long)bqd[ch] - (long)te.zero_point * swd);
    md[ch] = mud[ch];
  }
  v4i* dpe;
  long* dpd;
  int *dbe, *dbd;
  float *dme, *dmd;
  int rc;
  if ((rc = upload(m, pe, &dpe)) || (rc = upload(m, pdc, &dpd)) || (rc = upload(m, be, &dbe)) || (rc = upload(m, bd, &dbd)) ||
      (rc = upload(m, me, &dme)) || (rc = upload(m, md, &dmd)))
    return rc;
  a.we = dpe; a.wdc = dpd; a.be = dbe; a.bd = dbd; a.me = dme; a.md = dmd;
  a.rqe = make_rq(te.zero_point, eop.act_min, eop.act_max, conv_kb(m, eop));
  a.rqd = make_rq(td.zero_point, dop.act_min, dop.act_max, conv_kb(m, dop));
  a.zeb = (unsigned)(te.zero_point & 255) * 0x01010101u;
  s.lds_bytes = geo.t0_bytes + geo.e_bytes + geo.d_bytes;
  for (int oi : {e_op, d_op}) {
    s.alg_bytes_per_frame += m->op_steps[oi].alg_bytes_per_frame;
    s.weight_bytes += m->op_steps[oi].weight_bytes;
    s.macs_per_frame += m->op_steps[oi].macs_per_frame;
  }
  if ((rc = make_expdw2(m, e_op, d_op, &s, pe, be, me, bd, md))) return rc;
  *out = s;
  return VBT_OK;
}

// ---- second form of the expand + depthwise kernel (expdw2_block.h): stride 1, Cin % 16 == 0 ----
// LDS cycles of the depthwise operand reads (ds_read_b128: four groups of 16 lanes, one cycle per group when its 16-byte pieces
// fall on distinct quarters of the 64 banks; equal addresses broadcast) summed over the positions of a band, for a row stride EYS
static long xd2_read_cycles(int PR, int XB, int EYS, int RM) {   // PR position rows, RM rows of the expanded image between them
  static const int grp[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                 {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                                 {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59},
                                 {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
  const int NPOS = PR * XB;
  long cycles = 0;
  for (int pg = 0; pg * 16 < NPOS; pg++)
    for (int k = 0; k < 4; k++) {
      std::vector<int> seen[16];
      int worst = 1;
      for (int j = 0; j < 16; j++) {
        const int lane = grp[k][j], r = lane & 15, g = lane >> 4, n = std::min(pg * 16 + r, NPOS - 1);
        const int addr = (n / XB) * RM * EYS + (n % XB) * 16 + (g >> 1) * EYS + 16 * (g & 1);
        std::vector<int>& v = seen[(addr >> 4) & 15];
        if (std::find(v.begin(), v.end(), addr) == v.end()) v.push_back(addr);
        worst = std::max(worst, (int)v.size());
      }
      cycles += worst;
    }
  return cycles;
}
struct ExpDw2Geom { bool ok; int nbands, brows, XB, EQS, EYS, e_bytes, PS, pe_off, pd_off, lds, gpw, gpw16; };
static ExpDw2Geom expdw2_geom(int H, int W, int OH, int OW, int k, int stride, int pad_t, int KS64, int nbands) {
  ExpDw2Geom g{};
  const int DY = stride, DX = 4 / stride;   // output pixels of a depthwise position: 1 x 4 (stride 1), 2 x 2 (stride 2)
  g.brows = (OH + nbands - 1) / nbands;
  if (DY == 2) g.brows = (g.brows + 1) & ~1;   // whole row pairs per band
  g.nbands = (OH + g.brows - 1) / g.brows;
  g.XB = (OW + DX - 1) / DX;
  const int KT2 = (stride * (DY - 1) + k + 1) / 2, PW = (OW - 1) * stride + k;
  g.EQS = (4 * PW + 15) & ~15;
  int in_rows = 0;
  for (int b = 0; b < g.nbands; b++) {
    const int oy0 = b * g.brows, oy1 = std::min(oy0 + g.brows, OH);
    const int lo = std::max(oy0 * stride - pad_t, 0), hi = std::min(oy0 * stride - pad_t + (oy1 - oy0 - 1) * stride + k, H);
    in_rows = std::max(in_rows, hi - lo);
  }
  const int PR = (g.brows + DY - 1) / DY;   // position rows of the tallest band
  long best = -1;
  for (int pad = 0; pad < 256; pad += 16) {
    const long cyc = xd2_read_cycles(PR, g.XB, 16 * g.EQS + pad, stride * DY);
    if (best < 0 || cyc < best) { best = cyc; g.EYS = 16 * g.EQS + pad; }
  }
  const int PHe = stride * DY * (PR - 1) + 2 * KT2;   // the last MFMA of a position may read a row past the kernel: zero weights, but the row must exist
  g.e_bytes = (PHe * g.EYS + 32 + 15) & ~15;
  int ps4 = g.brows * OW;
  while ((ps4 & 31) != 2) ps4++;
  g.PS = 4 * ps4;
  g.pe_off = g.e_bytes + 16 * g.PS;
  g.pd_off = g.pe_off + 16 * (KS64 * 256 + 32);
  g.lds = g.pd_off + 16 * (KT2 * 256 + 32);
  const int npgi = (in_rows * W + 15) / 16, need = (npgi + 3) / 4;
  g.gpw = need <= 2 ? 2 : need <= 4 ? 4 : need <= 7 ? 7 : 0;
  const int need16 = (npgi + 7) / 8;
  g.gpw16 = need16 <= 1 ? 1 : need16 <= 2 ? 2 : need16 <= 4 ? 4 : 0;
  if (g.gpw16 * KS64 > 8) g.gpw16 = 0;    // (the input operands a wave keeps: 4 registers each; beyond these the kernels spill)
  if (g.gpw * KS64 > 21) g.gpw = 0;
  if (stride == 2) g.gpw = 0;             // stride 2 exists on 16 waves only
  g.ok = (g.gpw > 0 || g.gpw16 > 0) && g.lds <= 100 * 1024 && PR * g.XB <= 16 * XD2_NPG && g.e_bytes < 65536 && 16 * g.PS < 65535;   // (16-bit LDS offsets in the kernel)
  return g;
}
static ExpDw2Geom expdw2_choose(int H, int W, int OH, int OW, int k, int stride, int pad_t, int KS64) {
  ExpDw2Geom g{};
  for (int nb = 1; nb <= OH; nb++) {
    g = expdw2_geom(H, W, OH, OW, k, stride, pad_t, KS64, nb);
    if (g.ok) return g;
  }
  g.ok = false;
  return g;
}
// fills s->xd2 from the finished first-form arguments s->xd and the host images of its expand weights / biases / multipliers
static int make_expdw2(vbt_model* m, int e_op, int d_op, Step* s, const std::vector<v4i>& pe, const std::vector<int>& be, const std::vector<float>& me,
                       const std::vector<int>& bd, const std::vector<float>& md) {
  static const bool enabled = !(getenv("VBT_XD_V2") && atoi(getenv("VBT_XD_V2")) == 0);
  const OpRec& dop = m->ops[d_op];
  const ExpDwArgs& a1 = s->xd;
  s->xd2_ok = false;
  const int KS64 = (a1.Cin + 63) / 64;
  if (!enabled || (dop.stride != 1 && dop.stride != 2) || (dop.k != 3 && dop.k != 5) || a1.Cin % 8 != 0 || KS64 < 2 || KS64 > 4) return VBT_OK;
  const ExpDw2Geom geo = expdw2_choose(a1.H, a1.W, a1.OH, a1.OW, dop.k, dop.stride, a1.pad_t, KS64);
  if (!geo.ok) return VBT_OK;
  ExpDw2Args& a = s->xd2;
  memset(&a, 0, sizeof(a));
  a.H = a1.H; a.W = a1.W; a.Cin = a1.Cin; a.OH = a1.OH; a.OW = a1.OW; a.Ce = a1.Ce;
  a.pad_t = a1.pad_t; a.pad_l = a1.pad_l;
  a.nchunks = a1.nchunks; a.cpw = 1; a.nbands = geo.nbands; a.brows = geo.brows;
  a.XB = geo.XB; a.EQS = geo.EQS; a.EYS = geo.EYS; a.e_bytes = geo.e_bytes; a.PS = geo.PS; a.pe_off = geo.pe_off; a.pd_off = geo.pd_off;
  a.rqe = a1.rqe; a.zeb = a1.zeb; a.rqd = a1.rqd;
  const TensorRec& te = m->tensors[m->ops[e_op].output];
  const int Ce = te.c, nch = a.nchunks, kk = dop.k, S = dop.stride, KT2 = (S * (S - 1) + kk + 1) / 2;
  const int NE = KS64 * 256 + 32, ND = KT2 * 256 + 32;
  const int8_t* wd = (const int8_t*)(m->blob.data() + dop.w_off);
  std::vector<v4i> ppe((size_t)nch * NE), ppd((size_t)nch * ND);
  for (int c = 0; c < nch; c++) {
    v4i* e = &ppe[(size_t)c * NE];
    memcpy(e, &pe[(size_t)c * KS64 * 256], sizeof(v4i) * KS64 * 256);
    memcpy(e + KS64 * 256, &be[c * 64], 256);
    memcpy(e + KS64 * 256 + 16, &me[c * 64], 256);
    v4i* d = &ppd[(size_t)c * ND];
    unsigned* tab = (unsigned*)d;
    for (int q = 0; q < 16; q++)
      for (int mi = 0; mi < KT2; mi++)
        for (int lane = 0; lane < 64; lane++) {
          const int i = lane & 15, g = lane >> 4, qq = i >> 2, cc = i & 3, ch = 64 * c + 4 * q + cc;
          const int dy = S == 2 ? qq >> 1 : 0, dx = S == 2 ? qq & 1 : qq;   // the output pixel of the position this operand row computes
          const int ty = 2 * mi + (g >> 1) - S * dy;
          unsigned w4 = 0;
          for (int j = 0; j < 4; j++) {
            const int tx = 4 * (g & 1) + j - S * dx;
            if (ty >= 0 && ty < kk && tx >= 0 && tx < kk && ch < Ce) w4 |= (unsigned)(uint8_t)wd[(size_t)(ty * kk + tx) * Ce + ch] << (8 * j);
          }
          tab[(q * KT2 + mi) * 64 + lane] = w4;
        }
    memcpy(d + KT2 * 256, &bd[c * 64], 256);
    memcpy(d + KT2 * 256 + 16, &md[c * 64], 256);
  }
  v4i *dpe, *dpd;
  int rc;
  if ((rc = upload(m, ppe, &dpe)) || (rc = upload(m, ppd, &dpd))) return rc;
  a.pe = dpe; a.pd = dpd;
  s->xd2_ok = true;
  s->xd2_lds = geo.lds;
  s->xd2_gpw = geo.gpw;
  s->xd2_gpw16 = geo.gpw16;
  return VBT_OK;
}

// the projection conv `p_op` with the block's residual ADD `a_op` = ADD(conv output, skip) evaluated in its epilogue
static Step pw_with_residual(const vbt_model* m, int p_op, int a_op) {
  Step s = m->op_steps[p_op];
  s.op = a_op;
  s.p_op = p_op;
  s.res_op = a_op;
  s.addq = m->op_steps[a_op].addq;
  s.alg_bytes_per_frame += m->op_steps[a_op].alg_bytes_per_frame;
  return s;
}

static int fuse_plan(vbt_model* m) {
  const int no = (int)m->ops.size();
  std::vector<int> consumers(m->tensors.size(), 0);
  for (const OpRec& op : m->ops)
    for (int i = 0; i < op.n_inputs; i++) consumers[op.inputs[i]]++;
  const bool fuse_mb = !(m->flags & 1) && !(m->flags & 2);
  const bool fuse_sep = !(m->flags & 1) && !(m->flags & 4);
  auto dw_ok = [&](const OpRec& d) { return (d.k == 3 || d.k == 5) && (d.stride == 1 || d.stride == 2); };
  auto sep_ok = [&](int di) {
    if (di + 1 >= no) return false;
    const OpRec& d = m->ops[di];
    const OpRec& p = m->ops[di + 1];
    return d.type == OP_DW && dw_ok(d) && p.type == OP_PW && p.inputs[0] == d.output && consumers[d.output] == 1 &&
           m->tensors[d.inputs[0]].c % 8 == 0 && (m->tensors[p.output].c + 63) / 64 <= 5;
  };
  // BiFPN nodes: ADD(2|3 inputs) -> DW -> PW where some ADD inputs come from a RESIZE / MAXPOOL used only here
  const bool fuse_node = fuse_sep && !(m->flags & VBT_MODEL_NO_NODE_FUSION);
  std::vector<int> producer(m->tensors.size(), -1);
  for (int i = 0; i < no; i++) producer[m->ops[i].output] = i;
  std::vector<char> absorbed(no, 0);
  std::vector<NodeSrc> node_of(no);
  std::vector<char> is_node(no, 0);
  if (fuse_node)
    for (int i = 0; i < no; i++) {
      const OpRec& ad = m->ops[i];
      if (ad.type != OP_ADD || ad.n_inputs != 2 || !sep_ok(i + 1) || m->ops[i + 1].inputs[0] != ad.output ||
          consumers[ad.output] != 1 || m->tensors[ad.output].c % 4 != 0)
        continue;
      NodeSrc ns;
      auto absorb = [&](int j, int t) {   // source j = tensor t, read through the resize / max pool that produced it when possible
        const int pj = producer[t];
        ns.tensor[j] = t; ns.mode[j] = 0; ns.rs_op[j] = -1;
        const bool same_q = pj >= 0 && m->tensors[m->ops[pj].inputs[0]].scale == m->tensors[t].scale && m->tensors[m->ops[pj].inputs[0]].zero_point == m->tensors[t].zero_point;
        if (pj >= 0 && same_q && consumers[t] == 1 && (m->ops[pj].type == OP_RESIZE_NN || (m->ops[pj].type == OP_MAXPOOL && m->ops[pj].k == 3 && m->ops[pj].stride == 2))) {
          ns.tensor[j] = m->ops[pj].inputs[0];
          ns.mode[j] = m->ops[pj].type == OP_RESIZE_NN ? 1 : 2;
          ns.rs_op[j] = pj;
          absorbed[pj] = 1;
        }
      };
      int pre = -1, pos = 0;
      if (ad.n_inputs == 2 && i >= 1)
        for (int j = 0; j < 2; j++) {
          const int pj = producer[ad.inputs[j]];
          if (pj == i - 1 && m->ops[pj].type == OP_ADD && m->ops[pj].n_inputs == 2 && consumers[ad.inputs[j]] == 1) { pre = pj; pos = j; }
        }
      if (pre >= 0) {
        ns.n = 3;
        ns.pre_add = pre;
        ns.chain = pos + 1;
        absorb(0, m->ops[pre].inputs[0]);
        absorb(1, m->ops[pre].inputs[1]);
        absorb(2, ad.inputs[1 - pos]);
        absorbed[pre] = 1;
      } else {
        ns.n = ad.n_inputs;
        for (int j = 0; j < ad.n_inputs; j++) absorb(j, ad.inputs[j]);
      }
      node_of[i] = ns;
      is_node[i] = 1;
    }
  for (int i = 0; i < no;) {
    const OpRec& op = m->ops[i];
    Group g;
    int span = 1;
    if (absorbed[i]) { i++; continue; }  // emitted with its node
    if (is_node[i]) {
      const NodeSrc& ns = node_of[i];
      Alt unf, a1, a2;
      for (int j = 0; j < ns.n; j++)
        if (ns.rs_op[j] >= 0) { unf.steps.push_back(m->op_steps[ns.rs_op[j]]); a1.steps.push_back(m->op_steps[ns.rs_op[j]]); }
      if (ns.pre_add >= 0) { unf.steps.push_back(m->op_steps[ns.pre_add]); a1.steps.push_back(m->op_steps[ns.pre_add]); }
      unf.steps.push_back(m->op_steps[i]);
      unf.steps.push_back(m->op_steps[i + 1]);
      unf.steps.push_back(m->op_steps[i + 2]);
      g.alts.push_back(unf);
      a1.steps.push_back(m->op_steps[i]);
      Step s1, s2;
      int rc = make_fused(m, -1, i + 1, i + 2, -1, &s1);
      if (rc) return rc;
      if (s1.lds_bytes <= 64 * 1024) {
        a1.steps.push_back(s1);
        a1.hidden.push_back(m->ops[i + 1].output);
        g.alts.push_back(a1);
      }
      rc = make_fused(m, -1, i + 1, i + 2, -1, &s2, i, &ns);
      if (rc) return rc;
      if (s2.lds_bytes <= 64 * 1024) {
        a2.steps.push_back(s2);
        for (int j = 0; j < ns.n; j++)
          if (ns.rs_op[j] >= 0) a2.hidden.push_back(m->ops[ns.rs_op[j]].output);
        if (ns.pre_add >= 0) a2.hidden.push_back(m->ops[ns.pre_add].output);
        a2.hidden.push_back(op.output);
        a2.hidden.push_back(m->ops[i + 1].output);
        g.alts.push_back(a2);
        if (band_ok(m, i + 1, i + 2)) {   // the same node on row bands (band_block.h)
          Alt a3;
          Step s3;
          rc = make_band(m, i + 1, i + 2, i, &ns, &s3);
          if (rc) return rc;
          if (s3.lds_bytes <= 160 * 1024) {
            a3.steps.push_back(s3);
            a3.hidden = a2.hidden;
            g.alts.push_back(a3);
          }
        }
      }
      g.chosen = (int)g.alts.size() - 1;
      m->groups.push_back(g);
      i += 3;
      continue;
    }
    if (fuse_sep && !(m->flags & VBT_MODEL_NO_STEM_FUSION) && stem_block_ok(m, i, consumers)) {
      Alt unf, a1, a2;
      for (int k = 0; k < 3; k++) unf.steps.push_back(m->op_steps[i + k]);
      g.alts.push_back(unf);
      a1.steps.push_back(m->op_steps[i]);
      Step s1, s2;
      int rc = make_fused(m, -1, i + 1, i + 2, -1, &s1);
      if (rc) return rc;
      if (s1.lds_bytes <= 64 * 1024) {
        a1.steps.push_back(s1);
        a1.hidden.push_back(m->ops[i + 1].output);
        g.alts.push_back(a1);
      }
      rc = make_stem_block(m, i, &s2);
      if (rc) return rc;
      a2.steps.push_back(s2);
      a2.hidden.push_back(op.output);
      a2.hidden.push_back(m->ops[i + 1].output);
      g.alts.push_back(a2);
      g.chosen = (int)g.alts.size() - 1;
      m->groups.push_back(g);
      i += 3;
      continue;
    }
    bool mb = op.type == OP_PW && i + 2 < no && m->ops[i + 1].inputs[0] == op.output && consumers[op.output] == 1 && sep_ok(i + 1) &&
              m->tensors[op.inputs[0]].c % 8 == 0;
    if (mb && (fuse_mb || fuse_sep)) {
      const OpRec& d = m->ops[i + 1];
      const OpRec& p = m->ops[i + 2];
      int a_op = -1;
      if (i + 3 < no) {
        const OpRec& ad = m->ops[i + 3];
        if (ad.type == OP_ADD && ad.n_inputs == 2 && ad.inputs[0] == p.output && ad.inputs[1] == op.inputs[0] &&
            consumers[p.output] == 1 && d.stride == 1 && m->tensors[op.inputs[0]].c == m->tensors[p.output].c)
          a_op = i + 3;
      }
      span = a_op >= 0 ? 4 : 3;
      Alt unf;
      for (int k = 0; k < span; k++) unf.steps.push_back(m->op_steps[i + k]);
      g.alts.push_back(unf);
      if (a_op >= 0 && fuse_sep && m->tensors[p.output].c % 8 == 0) {   // one kernel per conv, the residual ADD in the projection's epilogue
        Alt ar;
        ar.steps.push_back(m->op_steps[i]);
        ar.steps.push_back(m->op_steps[i + 1]);
        ar.steps.push_back(pw_with_residual(m, i + 2, a_op));
        ar.hidden.push_back(p.output);
        g.alts.push_back(ar);
      }
      if (fuse_sep) {  // expand as its own kernel, dw+project(+add) fused
        Alt a2;
        a2.steps.push_back(m->op_steps[i]);
        Step s;
        int rc = make_fused(m, -1, i + 1, i + 2, -1, &s);  // residual stays a separate ADD (its skip input is not in T0)
        if (rc) return rc;
        if (s.lds_bytes <= 64 * 1024) {
          a2.steps.push_back(s);
          a2.hidden.push_back(d.output);
          if (a_op >= 0) a2.steps.push_back(m->op_steps[a_op]);
          g.alts.push_back(a2);
        }
      }
      if (fuse_mb) {
        Alt a3;
        Step s;
        int rc = make_fused(m, i, i + 1, i + 2, a_op, &s);
        if (rc) return rc;
        if (s.lds_bytes <= 64 * 1024) {
          a3.steps.push_back(s);
          a3.hidden.push_back(op.output);
          a3.hidden.push_back(d.output);
          if (a_op >= 0) a3.hidden.push_back(p.output);
          g.alts.push_back(a3);
        }
      }
      if (fuse_mb && !(m->flags & VBT_MODEL_NO_EXPDW) && expdw_ok(m, i, i + 1)) {
        // low-resolution blocks: expand + depthwise on whole images (channel-split grid), projection as a pointwise GEMM with
        // the residual in its epilogue
        Alt ax;
        Step sx;
        int rc = make_expdw(m, i, i + 1, &sx);
        if (rc) return rc;
        if (sx.lds_bytes <= 160 * 1024) {
          ax.steps.push_back(sx);
          ax.hidden.push_back(op.output);
          if (a_op >= 0 && m->tensors[p.output].c % 8 == 0) {
            ax.steps.push_back(pw_with_residual(m, i + 2, a_op));
            ax.hidden.push_back(p.output);
          } else {
            ax.steps.push_back(m->op_steps[i + 2]);
            if (a_op >= 0) ax.steps.push_back(m->op_steps[a_op]);
          }
          g.alts.push_back(ax);
        }
      }
    } else if (fuse_sep && sep_ok(i)) {
      span = 2;
      Alt unf;
      unf.steps.push_back(m->op_steps[i]);
      unf.steps.push_back(m->op_steps[i + 1]);
      g.alts.push_back(unf);
      Alt a2;
      Step s;
      int rc = make_fused(m, -1, i, i + 1, -1, &s);
      if (rc) return rc;
      if (s.lds_bytes <= 64 * 1024) {
        a2.steps.push_back(s);
        a2.hidden.push_back(op.output);
        g.alts.push_back(a2);
        if (band_ok(m, i, i + 1)) {
          Alt a3;
          Step s3;
          rc = make_band(m, i, i + 1, -1, nullptr, &s3);
          if (rc) return rc;
          if (s3.lds_bytes <= 160 * 1024) {
            a3.steps.push_back(s3);
            a3.hidden.push_back(op.output);
            g.alts.push_back(a3);
          }
        }
      }
    } else {
      Alt unf;
      unf.steps.push_back(m->op_steps[i]);
      g.alts.push_back(unf);
    }
    g.chosen = (int)g.alts.size() - 1;  // without autotuning: the most fused alternative
    m->groups.push_back(g);
    i += span;
  }
  if (fuse_sep && !(m->flags & VBT_MODEL_NO_HEAD_BATCHING)) {
    int rc = batch_heads(m);
    if (rc) return rc;
  }
  return VBT_OK;
}

// Pointwise convs that read nothing but tensors already there when the first of them runs are merged into one launch
// (pw_multi_kernel): in an EfficientDet graph the P6 conv and the five lateral convs of the first BiFPN cell all read backbone
// outputs.  The anchor is the stand-alone pointwise conv with the most followers; a conv followed by the two 3x3/2 max pools
// that make P6 and P7 takes them along.  Every merged tensor is still written, by the same arithmetic.
static bool pwm_mergeable(const vbt_model* m, const Step& s) {
  if (s.family != F_PW || s.res_op >= 0 || !s.members.empty() || !s.wp64) return false;
  const OpRec& op = m->ops[s.op];
  return op.type == OP_PW && m->tensors[op.output].c % 4 == 0 && s.KS64 >= 1;
}
static void merge_side_convs(vbt_model* m) {
  static const bool off = getenv("VBT_NO_PW_MERGE") != nullptr;
  if (off || (m->flags & (VBT_MODEL_NO_FUSION | VBT_MODEL_NO_PW_MERGE))) return;
  // sub-batch streams: the merged launch's problem list holds whole-batch pointers (the guard band_ok() has; n_sub is final by now)
  if (m->n_sub > 1) return;
  const int ns = (int)m->steps.size(), no = (int)m->ops.size();
  // which step runs which graph op: the ops a step names, then (absorbed resamples / partial sums) the step of their consumer
  std::vector<int> step_of(no, -1);
  std::function<void(const Step&, int)> claim = [&](const Step& s, int i) {
    for (int o : {s.op, s.e_op, s.d_op, s.p_op, s.a_op, s.res_op, s.sum_op})
      if (o >= 0 && o < no) step_of[o] = i;
    for (const Step& ms : s.members) claim(ms, i);
  };
  for (int i = 0; i < ns; i++) claim(m->steps[i], i);
  std::vector<std::vector<int>> readers(m->tensors.size());
  std::vector<int> producer(m->tensors.size(), -1);
  for (int o = 0; o < no; o++) {
    producer[m->ops[o].output] = o;
    for (int k = 0; k < m->ops[o].n_inputs; k++) readers[m->ops[o].inputs[k]].push_back(o);
  }
  for (int o = no - 1; o >= 0; o--)
    if (step_of[o] < 0) {
      int best = ns;
      for (int r : readers[m->ops[o].output]) if (step_of[r] >= 0) best = std::min(best, step_of[r]);
      step_of[o] = best < ns ? best : -1;
    }
  for (int o = 0; o < no; o++) if (step_of[o] < 0) return;    // an op nobody runs: leave the plan alone
  auto made_at = [&](int tensor) { return producer[tensor] >= 0 ? step_of[producer[tensor]] : -1; };
  auto first_read = [&](int tensor, int except_op = -1) {
    int f = ns;
    for (int r : readers[tensor]) if (r != except_op) f = std::min(f, step_of[r]);
    return f;
  };
  // the P6 / P7 chain: a conv whose output feeds a 3x3/2 max pool that feeds another one (steps of their own)
  auto pools_of = [&](int conv_step, int* k1, int* k2) {
    const int t0 = m->ops[m->steps[conv_step].op].output;
    for (int r1 : readers[t0]) {
      const OpRec& p1 = m->ops[r1];
      if (p1.type != OP_MAXPOOL || p1.k != 3 || p1.stride != 2 || m->steps[step_of[r1]].family != F_MAXPOOL) continue;
      for (int r2 : readers[p1.output]) {
        const OpRec& p2 = m->ops[r2];
        if (p2.type != OP_MAXPOOL || p2.k != 3 || p2.stride != 2 || m->steps[step_of[r2]].family != F_MAXPOOL) continue;
        const TensorRec &to = m->tensors[t0], &t1 = m->tensors[p1.output];
        if (to.c % 16 != 0 || (size_t)((to.h * to.w * to.c + 15) & ~15) + (size_t)t1.h * t1.w * t1.c > 64 * 1024) continue;
        *k1 = step_of[r1]; *k2 = step_of[r2];
        return true;
      }
    }
    return false;
  };
  // every stand-alone conv as the anchor (the slot the merged launch takes): member j fits when its input exists before the anchor
  // and nobody reads its output before the anchor has run
  int best = -1, best_chain = -1, bk1 = -1, bk2 = -1;
  std::vector<int> best_set;
  for (int a = 0; a < ns; a++) {
    if (!pwm_mergeable(m, m->steps[a])) continue;
    std::vector<int> set;
    int chain = -1, k1 = -1, k2 = -1;
    for (int j = 0; j < ns && (int)set.size() < PWM_MAX; j++) {
      if (!pwm_mergeable(m, m->steps[j])) continue;
      const OpRec& cj = m->ops[m->steps[j].op];
      if (j > a && made_at(cj.inputs[0]) >= a) continue;            // hoisted to the anchor: its input must exist by then
      int c1, c2;
      if (chain < 0 && pools_of(j, &c1, &c2)) {
        // the pools come along: the conv's other readers and the pools' readers must all run after the anchor
        const OpRec &p1 = m->ops[m->steps[c1].op], &p2 = m->ops[m->steps[c2].op];
        if (first_read(cj.output, m->steps[c1].op) > a && first_read(p1.output, m->steps[c2].op) > a && first_read(p2.output) > a) {
          chain = j; k1 = c1; k2 = c2;
          set.push_back(j);
          continue;
        }
      }
      if (first_read(cj.output) > a) set.push_back(j);               // (sunk or hoisted: nobody reads its output before the anchor has run)
    }
    // no member may read another member's output (they run side by side)
    for (bool again = true; again;) {
      again = false;
      for (size_t q = 0; q < set.size(); q++) {
        const int src = made_at(m->ops[m->steps[set[q]].op].inputs[0]);
        if (src >= 0 && std::find(set.begin(), set.end(), src) != set.end()) {
          if (set[q] == chain) chain = -1;
          set.erase(set.begin() + q);
          again = true;
          break;
        }
      }
    }
    if (std::find(set.begin(), set.end(), a) == set.end()) continue;
    const int score = (int)set.size() + (chain >= 0 ? 2 : 0);
    const int best_score = (int)best_set.size() + (best_chain >= 0 ? 2 : 0);
    if (set.size() >= 2 && score >= best_score) { best = a; best_set = set; best_chain = chain; bk1 = k1; bk2 = k2; }
  }
  if (best < 0) return;
  if (best_chain >= 0) {   // the chain problem goes first (pw_multi_kernel: problem 0)
    best_set.erase(std::find(best_set.begin(), best_set.end(), best_chain));
    best_set.insert(best_set.begin(), best_chain);
  }
  Step merged = m->steps[best_set[0]];
  merged.members.clear();
  merged.alg_bytes_per_frame = merged.weight_bytes = merged.macs_per_frame = 0;
  merged.nbp = 0;   // (1: problem 0 carries its two pools)
  std::vector<char> drop(ns, 0);
  for (int j : best_set) {
    merged.members.push_back(m->steps[j]);
    merged.alg_bytes_per_frame += m->steps[j].alg_bytes_per_frame;
    merged.weight_bytes += m->steps[j].weight_bytes;
    merged.macs_per_frame += m->steps[j].macs_per_frame;
    drop[j] = 1;
  }
  if (best_chain >= 0) {
    const TensorRec &to = m->tensors[m->ops[m->steps[best_chain].op].output], &t1 = m->tensors[m->ops[m->steps[bk1].op].output];
    merged.nbp = 1;
    merged.lds_bytes = (int)(((to.h * to.w * to.c + 15) & ~15) + t1.h * t1.w * t1.c);
    for (int k : {bk1, bk2}) {
      merged.members.push_back(m->steps[k]);
      merged.alg_bytes_per_frame += m->steps[k].alg_bytes_per_frame;
      drop[k] = 1;
    }
  }
  std::vector<Step> out;
  for (int i = 0; i < ns; i++) {
    if (i == best) out.push_back(merged);
    else if (!drop[i]) out.push_back(m->steps[i]);
  }
  m->steps.swap(out);
}

static void finalize_plan(vbt_model* m) {
  m->steps.clear();
  m->materialized.assign(m->tensors.size(), 1);
  for (const Group& g : m->groups) {
    const Alt& a = g.alts[g.chosen];
    for (const Step& s : a.steps) m->steps.push_back(s);
    for (int t : a.hidden) m->materialized[t] = 0;
  }
  merge_side_convs(m);
}

static int build_plan(vbt_model* m) {
  const int no = (int)m->ops.size();
  for (int oi = 0; oi < no; oi++) {
    const OpRec& op = m->ops[oi];
    const TensorRec& to = m->tensors[op.output];
    Step s;
    s.op = oi;
    double in_el = 0;
    for (int i = 0; i < op.n_inputs; i++) {
      const TensorRec& ti = m->tensors[op.inputs[i]];
      in_el += (double)ti.h * ti.w * ti.c;
    }
    double out_el = (double)to.h * to.w * to.c;
    s.alg_bytes_per_frame = in_el + out_el;
    if (op.type == OP_STEM || op.type == OP_PW || op.type == OP_DW) {
      const TensorRec& ti = m->tensors[op.inputs[0]];
      const int8_t* w = (const int8_t*)(m->blob.data() + op.w_off);
      const int32_t* bq = (const int32_t*)(m->blob.data() + op.b_off);
      const float* mu = (const float*)(m->blob.data() + op.m_off);
      const int N = to.c;
      const int zx = ti.zero_point;
      if (op.type == OP_DW) {
        s.family = F_DW;
        const int C = N, kk = op.k * op.k;
        if (C % 4 != 0 || !((op.k == 3 || op.k == 5) && (op.stride == 1 || op.stride == 2))) {
          set_error("unsupported depthwise conv: C=%d k=%d s=%d", C, op.k, op.stride);
          return VBT_ERR_ARG;
        }
        std::vector<float> wf((size_t)kk * C);
        std::vector<int> bias(C);
        std::vector<float> mult(mu, mu + C);
        for (int c = 0; c < C; c++) {
          long sw = 0;
          for (int t = 0; t < kk; t++) { wf[(size_t)t * C + c] = (float)w[(size_t)t * C + c]; sw += w[(size_t)t * C + c]; }
          bias[c] = (int)((long)bq[c] - (long)(128 + zx) * sw);  // acc uses u = x_q + 128, pad u = 128 + z_x
        }
        int rc;
        if ((rc = upload(m, wf, &s.wf)) || (rc = upload(m, bias, &s.bias)) || (rc = upload(m, mult, &s.mult))) return rc;
        {  // matrix-pipe form (see make_fused)
          const int Cp = (C + 63) / 64 * 64, KT = (kk + 1) / 2;
          std::vector<long> wdm((size_t)(Cp / 64) * 4 * KT * 64, 0);
          std::vector<int> biasm(Cp, 0);
          std::vector<float> multm(Cp, 0.0f);
          int8_t* wb = (int8_t*)wdm.data();
          for (int ch = 0; ch < Cp / 64; ch++)
            for (int cg = 0; cg < 4; cg++)
              for (int mi = 0; mi < KT; mi++)
                for (int lane = 0; lane < 64; lane++) {
                  int i = lane & 15, g = lane >> 4, c = 64 * ch + 16 * cg + i, tap = 2 * mi + (g >> 1);
                  for (int j = 0; j < 8; j++) {
                    int cp = 8 * (g & 1) + j;
                    wb[((((size_t)(ch * 4 + cg) * KT + mi) * 64 + lane) * 8) + j] = (tap < kk && cp == i && c < C) ? w[(size_t)tap * C + c] : 0;
                  }
                }
          for (int c = 0; c < C; c++) {
            long sw = 0;
            for (int t = 0; t < kk; t++) sw += w[(size_t)t * C + c];
            biasm[c] = (int)((long)bq[c] - (long)zx * sw);
            multm[c] = mu[c];
          }
          if ((rc = upload(m, wdm, &s.wdm)) || (rc = upload(m, biasm, &s.bdm)) || (rc = upload(m, multm, &s.mdm))) return rc;
        }
        s.weight_bytes = (double)kk * C;
        s.macs_per_frame = out_el * kk;
      } else {
        const bool stem = op.type == OP_STEM;
        s.family = stem ? F_STEM : F_PW;
        const int K = stem ? op.k * op.k * ti.c : ti.c;
        if (stem && (op.k != 3 || ti.c != 3 || op.stride != 2)) { set_error("unsupported stem conv"); return VBT_ERR_ARG; }
        if (!stem && (K % 8) != 0) { set_error("pointwise conv needs Cin %% 8 == 0 (got %d)", K); return VBT_ERR_ARG; }
        s.KS = (K + 31) / 32;
        s.NB = (N + 63) / 64;
        std::vector<int> kmap;
        if (stem) {
          kmap.assign(32, -1);
          for (int g = 0; g < 3; g++)
            for (int j = 0; j < 8; j++) kmap[8 * g + j] = (g * 3 + j / 3) * 3 + j % 3;  // (ky=g, kx=j/3, c=j%3)
          for (int j = 0; j < 3; j++) kmap[24 + j] = (j * 3 + 2) * 3 + 2;                // (ky=j, kx=2, c=2)
        }
        std::vector<long> wp;
        pack_weights(w, N, K, s.KS, s.NB, stem ? &kmap : nullptr, wp);
        std::vector<int> bias((size_t)s.NB * 64, 0);
        std::vector<float> mult((size_t)s.NB * 64, 0.0f);
        for (int c = 0; c < N; c++) {
          long sw = 0;
          for (int k = 0; k < K; k++) sw += w[(size_t)c * K + k];
          bias[c] = (int)((long)bq[c] - (long)zx * sw);  // acc = sum x_q*w ; (x_q - z_x) folded here
          mult[c] = mu[c];
        }
        int rc;
        if ((rc = upload(m, wp, &s.wp)) || (rc = upload(m, bias, &s.bias)) || (rc = upload(m, mult, &s.mult))) return rc;
        if (!stem) {
          s.KS64 = (K + 63) / 64;
          std::vector<v4i> wp64;
          pack_weights64(w, N, K, s.KS64, s.NB, wp64);
          if ((rc = upload(m, wp64, &s.wp64))) return rc;
        }
        s.weight_bytes = (double)N * K;
        s.macs_per_frame = out_el * K;
      }
    } else if (op.type == OP_ADD) {
      s.family = F_ADD;
      if (((long)to.h * to.w * to.c) % 4 != 0 || op.n_inputs != 2) { set_error("unsupported add (binary int8 ADD on a multiple of 4 elements expected)"); return VBT_ERR_ARG; }
      const TensorRec& ta = m->tensors[op.inputs[0]];
      const TensorRec& tb = m->tensors[op.inputs[1]];
      AddParams ap;
      if (!xnn_add_params(ta.scale, tb.scale, to.scale, ta.zero_point, tb.zero_point, &ap)) {
        set_error("op %d: ADD input/output scale ratio outside [2^-10, 2^8) (XNNPACK refuses it too)", oi);
        return VBT_ERR_ARG;
      }
      if (ap.bias != op.add_q[0] || ap.am != op.add_q[1] || ap.bm != op.add_q[2] || ap.shift != op.add_q[3]) {
        set_error("op %d: ADD parameters stored in the container (%d,%d,%d,%d) differ from those derived from the tensor scales (%d,%d,%d,%d)",
                  oi, op.add_q[0], op.add_q[1], op.add_q[2], op.add_q[3], ap.bias, ap.am, ap.bm, ap.shift);
        return VBT_ERR_ARG;
      }
      s.addq = make_addq(ap, to.zero_point, op.act_min, op.act_max);
      // survey accounting: a 3-input BiFPN sum is one add; the partial sum between its two binary ADDs is not traffic
      auto sole_add_consumer = [&](int t) {
        int n = 0, add = 0;
        for (const OpRec& o2 : m->ops)
          for (int i = 0; i < o2.n_inputs; i++)
            if (o2.inputs[i] == t) { n++; add += o2.type == OP_ADD; }
        return n == 1 && add == 1;
      };
      if (sole_add_consumer(op.output)) s.alg_bytes_per_frame -= out_el;
      for (int i = 0; i < 2; i++) {
        bool from_add = false;
        for (int o2 = 0; o2 < oi; o2++) from_add |= m->ops[o2].type == OP_ADD && m->ops[o2].output == op.inputs[i];
        if (from_add && sole_add_consumer(op.inputs[i])) s.alg_bytes_per_frame -= (double)m->tensors[op.inputs[i]].h * m->tensors[op.inputs[i]].w * m->tensors[op.inputs[i]].c;
      }
    } else if (op.type == OP_MAXPOOL) {
      s.family = F_MAXPOOL;
      if (to.c % 4 != 0 || op.k != 3 || op.stride != 2) { set_error("unsupported maxpool"); return VBT_ERR_ARG; }
    } else if (op.type == OP_RESIZE_NN) {
      s.family = F_RESIZE;
      if (to.c % 4 != 0) { set_error("unsupported resize"); return VBT_ERR_ARG; }
      // TFLite's kernel computes src = min(floor(dst * (float)in / out), in - 1) in float32; the kernels here use the
      // integer form floor(dst * in / out): refuse a geometry on which the two differ
      const TensorRec& ti = m->tensors[op.inputs[0]];
      for (int ax = 0; ax < 2; ax++) {
        const int in = ax ? ti.w : ti.h, out = ax ? to.w : to.h;
        const float scale = (float)in / (float)out;
        for (int d = 0; d < out; d++)
          if (std::min((int)floorf((float)d * scale), in - 1) != (d * in) / out) {
            set_error("op %d: nearest-neighbour resize %d -> %d is not mapped", oi, in, out);
            return VBT_ERR_ARG;
          }
      }
    } else if (op.type == OP_POSTPROCESS) {
      s.family = F_POST;
      if (op.n_inputs != 10 || m->hdr.max_detections != VBT_MAX_DETECTIONS || m->hdr.num_anchors > 65535) {
        set_error("unsupported postprocess configuration");
        return VBT_ERR_ARG;
      }
      s.alg_bytes_per_frame = in_el + m->hdr.max_detections * 24.0;
      s.weight_bytes = (double)m->hdr.num_anchors * 16;
    } else {
      set_error("unknown op type %d", op.type);
      return VBT_ERR_ARG;
    }
    m->op_steps.push_back(s);
  }
  return fuse_plan(m);
}

template <int KS>
static void launch_pw_a(int MS, dim3 grid, hipStream_t st, const int8_t* x, const v4i* wp, Epi e, ResArgs ra, int8_t* out, long M, int K,
                        int N, int NB, int nb_per_y) {
  if (MS == 2) pw_a_kernel<KS, 2><<<grid, 256, 0, st>>>(x, wp, e, ra, out, M, K, N, NB, nb_per_y);
  else pw_a_kernel<KS, 1><<<grid, 256, 0, st>>>(x, wp, e, ra, out, M, K, N, NB, nb_per_y);
}

static bool ppw2_fits(const vbt_model* m, const Step& st) {
  const OpRec& dop = m->ops[st.d_op];
  const int TX = 16, TY = 8;   // the 128-pixel kernels are DW64 (fused_block.h): 16 x 8 tiles only
  if (st.fa.OW < TX || st.fa.OH < TY || !st.fa.wd64) return false;
  const int TXp = TX;
  const int NPh = ((TXp - 1) * dop.stride + dop.k) * ((TY - 1) * dop.stride + dop.k);
  return ((NPh * st.fa.T0S + 15) & ~15) + ((NPh * FB_EST + 15) & ~15) + 128 * FB_DST <= 64 * 1024;
}

// Launches one plan step for frames [boff, boff + B) of the batch (every tensor is batch-major).
// Whole-image MBConv kernel (image_block.h): eligibility and launch geometry.
struct ImageGeom { int PW, PH, NB, maxu, lds; bool ok; };
static ImageGeom image_geom(const vbt_model* m, const Step& s) {
  ImageGeom g{0, 0, 0, 0, 0, false};
  if (s.family != F_MBCONV) return g;
  const FusedArgs& a = s.fa;
  const OpRec& dop = m->ops[s.d_op];
  const int HW = a.H * a.W, OHW = a.OH * a.OW;
  if (HW > 400 || OHW > 400) return g;
  g.PH = std::max((a.OH - 1) * dop.stride + dop.k, a.pad_t + a.H);
  g.PW = std::max((a.OW - 1) * dop.stride + dop.k, a.pad_l + a.W);
  g.NB = (a.Cout + 63) / 64;
  const int NPGo = (OHW + 15) / 16;
  const int units = (NPGo * g.NB + IB_WAVES - 1) / IB_WAVES;
  g.maxu = units <= 2 ? 2 : units <= 3 ? 3 : 4;
  g.lds = ((HW * a.T0S + 15) & ~15) + g.PH * g.PW * FB_EST + NPGo * 16 * FB_DST + s.ib.bytes;
  g.ok = s.ib.data != nullptr && units <= 4 && g.lds <= 160 * 1024 && s.ib.bytes <= 16 * IB_NPF * IB_THREADS;
  return g;
}
static int launch_step(vbt_model* m, const Step& s, int B, hipStream_t st, const uint8_t* frames, float* boxes, float* scores,
                       float* classes, int* counts, int boff = 0) {
  if (m->pool_dirty) { const int rc = flush_uploads(m); if (rc) return rc; }
  const OpRec& op = m->ops[s.op];
  const TensorRec& to = m->tensors[op.output];
  auto TP = [&](int t) { return m->tptr[t] + (size_t)boff * m->telems[t]; };
  frames += (size_t)boff * m->hdr.image_size * m->hdr.image_size * 3;
  boxes += (size_t)boff * m->hdr.max_detections * 4;
  scores += (size_t)boff * m->hdr.max_detections;
  classes += (size_t)boff * m->hdr.max_detections;
  counts += boff;
  int8_t* out = TP(op.output);
  Epi e{s.bias, s.mult, to.zero_point, op.act_min, op.act_max, make_rq(to.zero_point, op.act_min, op.act_max)};   // (F_PW rebuilds it from its conv op)
  switch (s.family) {
    case F_STEM: {
      const TensorRec& ti = m->tensors[op.inputs[0]];
      long M = (long)B * to.h * to.w;
      dim3 grid((unsigned)((M + 63) / 64));
      stem_kernel<<<grid, 256, 0, st>>>(frames, s.wp, e, out, M, ti.h, ti.w, to.h, to.w, to.c, op.pad_t, op.pad_l, ti.zero_point);
      break;
    }
    case F_PW: {
      if (!s.members.empty()) {   // several convs in one launch (merge_side_convs)
        if (boff != 0) { set_error("merged pointwise convs: sub-batch streams are not supported (VBT_SUBSTREAMS)"); return VBT_ERR_ARG; }
        PwMulti pm;
        memset(&pm, 0, sizeof(pm));
        const int nconv = (int)s.members.size() - (s.nbp ? 2 : 0);
        int acc = 0;
        for (int i = 0; i < nconv; i++) {
          const Step& ms = s.members[i];
          const OpRec& pop = m->ops[ms.op];
          const TensorRec& ti = m->tensors[pop.inputs[0]];
          const TensorRec& tpo = m->tensors[pop.output];
          PwProb& q = pm.p[i];
          q.x = TP(pop.inputs[0]); q.wp = ms.wp64; q.bias = ms.bias; q.mult = ms.mult; q.out = TP(pop.output);
          q.rq = make_rq(tpo.zero_point, pop.act_min, pop.act_max);
          q.K = ti.c; q.KS = ms.KS64; q.N = tpo.c; q.NB = ms.NB;
          pm.start[i] = acc;
          if (i == 0 && s.nbp) {
            const OpRec &p1 = m->ops[s.members[nconv].op], &p2 = m->ops[s.members[nconv + 1].op];
            const TensorRec &t1 = m->tensors[p1.output], &t2 = m->tensors[p2.output];
            q.M = tpo.h * tpo.w;
            q.pool1 = TP(p1.output); q.pool2 = TP(p2.output);
            q.H = tpo.h; q.W = tpo.w; q.H1 = t1.h; q.W1 = t1.w; q.pt1 = p1.pad_t; q.pl1 = p1.pad_l;
            q.H2 = t2.h; q.W2 = t2.w; q.pt2 = p2.pad_t; q.pl2 = p2.pad_l;
            acc += B;
          } else {
            q.M = B * tpo.h * tpo.w;
            acc += ((q.M + 63) / 64) * q.NB;
          }
        }
        pm.n = nconv;
        pm.start[nconv] = acc;
        for (int i = nconv + 1; i <= PWM_MAX; i++) pm.start[i] = acc;
        pw_multi_kernel<<<dim3((unsigned)acc), 256, s.nbp ? s.lds_bytes : 0, st>>>(pm);
        break;
      }
      // op = the op whose output is written: the conv itself, or the residual ADD evaluated in its epilogue (res_op)
      const OpRec& pop = m->ops[s.res_op >= 0 ? s.p_op : s.op];
      const TensorRec& ti = m->tensors[pop.inputs[0]];
      const TensorRec& tpo = m->tensors[pop.output];
      const int8_t* x = TP(pop.inputs[0]);
      e = Epi{s.bias, s.mult, tpo.zero_point, pop.act_min, pop.act_max, make_rq(tpo.zero_point, pop.act_min, pop.act_max)};
      ResArgs ra{nullptr, s.addq};
      if (s.res_op >= 0) ra.res = TP(m->ops[s.res_op].inputs[1]);   // ADD(conv output, skip): planner guarantees the order
      long M = (long)B * tpo.h * tpo.w;
      int K = ti.c, N = tpo.c;
      // VBT_PW_VARIANT (tests): the kernel variant of every pointwise conv with K > 256, whatever the plan says
      static const int pw_force = getenv("VBT_PW_VARIANT") ? atoi(getenv("VBT_PW_VARIANT")) : -100;
      const int pw_variant = (pw_force != -100 && s.KS64 > 4) ? pw_force : s.variant;
      if (s.KS64 <= 4) {
        int MS = s.variant >= 0 ? (s.variant & 1) + 1 : (M >= 32768 ? 2 : 1);
        long waves = (M + 16 * MS - 1) / (16 * MS);
        unsigned gx = (unsigned)((waves + 3) / 4);
        int ysplit = 1;
        while (gx * ysplit < 1024 && ysplit < s.NB) ysplit++;
        int nb_per_y = (s.NB + ysplit - 1) / ysplit;
        ysplit = (s.NB + nb_per_y - 1) / nb_per_y;
        dim3 grid(gx, ysplit);
        switch (s.KS64) {
          case 1: launch_pw_a<1>(MS, grid, st, x, s.wp64, e, ra, out, M, K, N, s.NB, nb_per_y); break;
          case 2: launch_pw_a<2>(MS, grid, st, x, s.wp64, e, ra, out, M, K, N, s.NB, nb_per_y); break;
          case 3: launch_pw_a<3>(MS, grid, st, x, s.wp64, e, ra, out, M, K, N, s.NB, nb_per_y); break;
          default: launch_pw_a<4>(MS, grid, st, x, s.wp64, e, ra, out, M, K, N, s.NB, nb_per_y); break;
        }
      } else if (pw_variant == 3 || pw_variant == 4) {  // weights shared through LDS, 64 (variant 3) or 128 (variant 4) pixels per workgroup
        const int ms = pw_variant - 2;
        dim3 grid((unsigned)((M + 64 * ms - 1) / (64 * ms)), (unsigned)s.NB);
        if (ms == 2) pw_d_kernel<2><<<grid, 256, 0, st>>>(x, s.wp64, e, ra, out, M, K, s.KS64, N, s.NB);
        else pw_d_kernel<1><<<grid, 256, 0, st>>>(x, s.wp64, e, ra, out, M, K, s.KS64, N, s.NB);
      } else if (pw_variant == 5 || pw_variant == 6) {  // the block's whole weight panel in LDS, a K loop without barriers (pw_e_kernel), 64 / 128 pixels per workgroup
        const int ms = pw_variant - 4;
        const int lds = s.KS64 * 4096;
        if (lds > 160 * 1024) { set_error("pw_conv: a weight panel of %d bytes does not fit the LDS", lds); return VBT_ERR_ARG; }
        dim3 grid((unsigned)((M + 64 * ms - 1) / (64 * ms)), (unsigned)s.NB);
        if (ms == 2) {
          if (lds > 64 * 1024) VBT_LDS_OPT_IN(pw_e_kernel<2, 4>);
          pw_e_kernel<2, 4><<<grid, 256, lds, st>>>(x, s.wp64, e, ra, out, M, K, s.KS64, N, s.NB);
        } else {
          if (lds > 64 * 1024) VBT_LDS_OPT_IN(pw_e_kernel<1, 4>);
          pw_e_kernel<1, 4><<<grid, 256, lds, st>>>(x, s.wp64, e, ra, out, M, K, s.KS64, N, s.NB);
        }
      } else if (pw_variant == 2) {  // split-K over the 4 waves of a workgroup
        // one 64-channel block per workgroup (16 KB of LDS for the cross-wave reduction, twice the workgroups) rather than two
        // (32 KB): +1.4 % end to end with three forwards in flight - the workgroups of one launch then fit the CUs in one round
        static const int nbt_max = getenv("VBT_PWC_NBT") ? atoi(getenv("VBT_PWC_NBT")) : 1;
        static const int ms_env = getenv("VBT_PWC_MS") ? atoi(getenv("VBT_PWC_MS")) : 2;
        int nbt = std::min(s.NB, nbt_max);
        // two pixel groups per workgroup (half the weight bytes through L1) as long as the grid still has two workgroups per CU
        const int ms = (ms_env >= 2 && nbt == 1 && ((M + 31) / 32) * s.NB >= 512) ? 2 : 1;
        dim3 grid((unsigned)((M + 16 * ms - 1) / (16 * ms)), (unsigned)((s.NB + nbt - 1) / nbt));
        if (ms == 2) pw_c_kernel<1, 2><<<grid, 256, 0, st>>>(x, s.wp64, e, ra, out, M, K, s.KS64, N, s.NB);
        else if (nbt == 1) pw_c_kernel<1, 1><<<grid, 256, 0, st>>>(x, s.wp64, e, ra, out, M, K, s.KS64, N, s.NB);
        else pw_c_kernel<2, 1><<<grid, 256, 0, st>>>(x, s.wp64, e, ra, out, M, K, s.KS64, N, s.NB);
      } else {
        long waves = (M + 15) / 16;
        unsigned gx = (unsigned)((waves + 3) / 4);
        int nbt = std::min(s.NB, 4);   // (1 or 2 blocks per wave measured no better: 93.0 / 92.7 k vs 93.0 k frames/s)
        if (gx < 512 && nbt > 2) nbt = 2;  // more workgroups for the low-resolution layers
        if (gx < 128) nbt = 1;
        dim3 grid(gx, (s.NB + nbt - 1) / nbt);
        switch (nbt) {
          case 1: pw_b_kernel<1><<<grid, 256, 0, st>>>(x, s.wp64, e, ra, out, M, K, s.KS64, N, s.NB); break;
          case 2: pw_b_kernel<2><<<grid, 256, 0, st>>>(x, s.wp64, e, ra, out, M, K, s.KS64, N, s.NB); break;
          case 3: pw_b_kernel<3><<<grid, 256, 0, st>>>(x, s.wp64, e, ra, out, M, K, s.KS64, N, s.NB); break;
          default: pw_b_kernel<4><<<grid, 256, 0, st>>>(x, s.wp64, e, ra, out, M, K, s.KS64, N, s.NB); break;
        }
      }
      break;
    }
    case F_DW: {
      const TensorRec& ti = m->tensors[op.inputs[0]];
      const int8_t* x = TP(op.inputs[0]);
      int C = to.c;
      unsigned pb = (unsigned)((128 + ti.zero_point) & 255);
      unsigned pad4 = pb | (pb << 8) | (pb << 16) | (pb << 24);
      if (s.variant == 100 || s.variant == 101) {  // LDS-tiled, chunk-parallel (101: depthwise on the matrix pipe)
        DwTileArgs a;
        a.x = x; a.out = out; a.wf = s.wf; a.bias = s.bias; a.mult = s.mult;
        a.wdm = s.wdm; a.bdm = s.bdm; a.mdm = s.mdm;
        a.H = ti.h; a.W = ti.w; a.C = C; a.OH = to.h; a.OW = to.w; a.pad_t = op.pad_t; a.pad_l = op.pad_l;
        choose_tile(to.h, to.w, op.k, op.stride, false, &a.TX, &a.TY);
        a.tiles_x = (to.w + a.TX - 1) / a.TX;
        a.tiles_y = (to.h + a.TY - 1) / a.TY;
        a.zx = ti.zero_point;
        a.rq = e.rq;
        const int TXp = (a.TX + 3) & ~3;
        const int NPh = ((TXp - 1) * op.stride + op.k) * ((a.TY - 1) * op.stride + op.k);
        dim3 grid((unsigned)((long)B * a.tiles_x * a.tiles_y), (unsigned)((C + 63) / 64));
        const int lds = NPh * 80;
        launch_dw_tile(a, op.k, op.stride, s.variant == 101, grid, lds, st);
      } else if (s.variant == 0) {  // one output row x 4 columns per lane
        long total = (long)B * to.h * ((to.w + 3) / 4) * (C / 4);
        dim3 grid((unsigned)((total + 255) / 256));
#define DW_LAUNCH(KK, S) dw_kernel<KK, S><<<grid, 256, 0, st>>>(x, s.wf, e, out, total, ti.h, ti.w, C, to.h, to.w, op.pad_t, op.pad_l, pad4)
        if (op.k == 3 && op.stride == 1) DW_LAUNCH(3, 1);
        else if (op.k == 3 && op.stride == 2) DW_LAUNCH(3, 2);
        else if (op.k == 5 && op.stride == 1) DW_LAUNCH(5, 1);
        else DW_LAUNCH(5, 2);
#undef DW_LAUNCH
      } else {               // column walker, `rows` output rows per lane
        const int XR = (to.w + 3) / 4;
        long per_seg = (long)B * XR * (C / 4);
        int rows = s.variant > 0 ? s.variant : (int)std::min<long>(std::max<long>(to.h * per_seg / 400000, 1), 16);
        rows = std::min(rows, to.h);
        int nseg = (to.h + rows - 1) / rows;
        long total = per_seg * nseg;
        dim3 grid((unsigned)((total + 255) / 256));
#define DW_LAUNCH(KK, S) dw_col_kernel<KK, S><<<grid, 256, 0, st>>>(x, s.wf, e, out, total, ti.h, ti.w, C, to.h, to.w, op.pad_t, op.pad_l, pad4, rows, nseg)
        if (op.k == 3 && op.stride == 1) DW_LAUNCH(3, 1);
        else if (op.k == 3 && op.stride == 2) DW_LAUNCH(3, 2);
        else if (op.k == 5 && op.stride == 1) DW_LAUNCH(5, 1);
        else DW_LAUNCH(5, 2);
#undef DW_LAUNCH
      }
      break;
    }
    case F_ADD: {
      long n4 = (long)B * to.h * to.w * to.c / 4;
      add_kernel<<<dim3((unsigned)((n4 + 1023) / 1024)), 256, 0, st>>>(TP(op.inputs[0]), TP(op.inputs[1]), s.addq, out, n4);
      break;
    }
    case F_MAXPOOL: {
      const TensorRec& ti = m->tensors[op.inputs[0]];
      long total = (long)B * to.h * to.w * (to.c / 4);
      maxpool_kernel<<<dim3((unsigned)((total + 255) / 256)), 256, 0, st>>>(TP(op.inputs[0]), out, total, ti.h, ti.w, ti.c,
                                                                             to.h, to.w, op.pad_t, op.pad_l);
      break;
    }
    case F_RESIZE: {
      const TensorRec& ti = m->tensors[op.inputs[0]];
      long total = (long)B * to.h * to.w * (to.c / 4);
      resize_kernel<<<dim3((unsigned)((total + 255) / 256)), 256, 0, st>>>(TP(op.inputs[0]), out, total, ti.h, ti.w, ti.c,
                                                                            to.h, to.w);
      break;
    }
    case F_MULTI: {
      if (boff != 0) {  // side-stream sub-batches: pointers differ, launch the members one by one
        for (const Step& ms : s.members) {
          Step t = ms;
          t.variant = s.variant;
          int rc = launch_step(m, t, B, st, frames - (size_t)boff * m->hdr.image_size * m->hdr.image_size * 3, boxes - (size_t)boff * m->hdr.max_detections * 4,
                               scores - (size_t)boff * m->hdr.max_detections, classes - (size_t)boff * m->hdr.max_detections, counts - boff, boff);
          if (rc) return rc;
        }
        break;
      }
      MultiTiles mt;
      mt.n = (int)s.members.size();
      int acc = 0;
      for (int i = 0; i < mt.n; i++) {
        mt.start[i] = acc;
        acc += B * s.members[i].fa.tiles_x * s.members[i].fa.tiles_y;
      }
      mt.start[mt.n] = acc;
      const OpRec& dop = m->ops[s.d_op];
      dim3 grid((unsigned)acc);
      const bool mdw = s.variant != 0;
      { const int rc = launch_fused_multi(s.d_multi, mt, dop.k, dop.stride, s.nbp, mdw, s.lds_bytes, (unsigned)acc, st); if (rc) return rc; }
      break;
    }
    case F_MBCONV:
    case F_NODE:
    case F_SEPCONV: {
      FusedArgs a = s.fa;
      a.x = s.e_op >= 0 ? TP(m->ops[s.e_op].inputs[0]) : TP(m->ops[s.d_op].inputs[0]);
      for (int j = 0; j < 3; j++) a.src[j] = s.src_tensor[j] >= 0 ? TP(s.src_tensor[j]) : nullptr;
      a.out = out;
      const OpRec& dop = m->ops[s.d_op];
      dim3 grid((unsigned)((long)B * a.tiles_x * a.tiles_y));
      const bool ex = s.family == F_MBCONV;
      // (the tile option of the variant is applied below, before the launch macros use `grid` / `lds_bytes`)
      // variant bit 0: depthwise on the matrix pipe (default) / VALU; bits 1..: 0 = heuristic tile, 1 = half-height tile
      int var = s.variant < 0 ? (((m->flags & VBT_MODEL_IMAGE_BLOCKS) && image_geom(m, s).ok) ? 5 : ((m->flags & VBT_MODEL_CHUNK48) ? 9 : 1)) : s.variant;
      if (s.variant < 0 && (m->flags & VBT_MODEL_TILE128) && s.family == F_MBCONV && s.nbp <= 2 && (a.KSe == 1 || a.KSe == 2) && ppw2_fits(m, s)) var |= 16;
      const bool mdw = var & 1;
      const bool nt3 = (var & 8) && mdw && a.nch3 > 0 && s.nbp <= 2 && a.KSe >= 1 && a.KSe <= 4;  // 48-channel chunks
      if (var & 4) {  // one workgroup per image (low-resolution blocks)
        const ImageGeom ig = image_geom(m, s);
        if (!ig.ok) { set_error("fused_mbconv: whole-image variant not applicable"); return VBT_ERR_ARG; }
        launch_mbconv_image(a, s.ib, dop.k, dop.stride, ig.maxu, ig.PW, ig.PH, ig.NB, ig.lds, B, st);
        break;
      }
      int lds_bytes = s.lds_bytes;
      if (((var >> 1) & 1) == 1 && a.TY >= 2) {
        a.TY = (a.TY + 1) / 2;
        a.tiles_y = (a.OH + a.TY - 1) / a.TY;
        const int TXp_ = (a.TX + 3) & ~3;
        const int NPh_ = ((TXp_ - 1) * dop.stride + dop.k) * ((a.TY - 1) * dop.stride + dop.k);
        lds_bytes = ((NPh_ * a.T0S + 15) & ~15) + (s.family == F_MBCONV ? NPh_ * FB_EST : 0) + 64 * FB_DST;
        if (s.family != F_MBCONV && a.nchunks == 1 && s.nbp <= 2) lds_bytes += s.nbp * (4096 + 512);
        grid = dim3((unsigned)((long)B * a.tiles_x * a.tiles_y));
      }
      if (nt3) {  // 72-byte E rows (fused_block.h)
        const int TXp_ = (a.TX + 3) & ~3;
        const int NPh_ = ((TXp_ - 1) * dop.stride + dop.k) * ((a.TY - 1) * dop.stride + dop.k);
        lds_bytes = ((NPh_ * a.T0S + 15) & ~15) + ((NPh_ * 72 + 15) & ~15) + 64 * FB_DST;
      }
      // variant bit 4: 128-pixel tiles (PPW = 2), matrix-pipe depthwise, register-resident expand weights (K <= 64), <= 128 output channels
      const bool ppw2 = (var & 16) && ex && mdw && (a.KSe == 1 || a.KSe == 2) && s.nbp <= 2 && !((var >> 1) & 1);
      if (ppw2) {
        a.TX = 16; a.TY = 8;
        a.tiles_x = (a.OW + a.TX - 1) / a.TX;
        a.tiles_y = (a.OH + a.TY - 1) / a.TY;
        const int TXp_ = (a.TX + 3) & ~3;
        const int NPh_ = ((TXp_ - 1) * dop.stride + dop.k) * ((a.TY - 1) * dop.stride + dop.k);
        lds_bytes = ((NPh_ * a.T0S + 15) & ~15) + ((NPh_ * (nt3 ? 48 : FB_EST) + 15) & ~15) + 128 * FB_DST;
        grid = dim3((unsigned)((long)B * a.tiles_x * a.tiles_y));
        if (a.OW < 16 || a.OH < 8 || !a.wd64) { set_error("fused_mbconv: the 128-pixel variant needs maps of at least 16 x 8"); return VBT_ERR_ARG; }
        if (lds_bytes > 64 * 1024) { set_error("fused_mbconv: 128-pixel tile needs %d bytes of LDS", lds_bytes); return VBT_ERR_ARG; }
        return launch_fused_block(a, FusedLaunch{dop.k, dop.stride, s.nbp, true, true, nt3, true, true, lds_bytes, grid.x}, st);
        break;
      }
      // 64-pixel tiles of exactly 8 x 8 outputs, K <= 64, <= 128 output channels: depthwise on the 16x16x64 MFMA (DW64)
      const bool dw64 = ex && mdw && a.wd64 && a.TX == 8 && a.TY == 8 && (a.KSe == 1 || a.KSe == 2) && s.nbp <= 2 && !getenv("VBT_NO_DW64");
      if (dw64 && nt3) {   // E rows are 48 bytes in the 48-channel DW64 kernels
        const int NPh_ = ((8 - 1) * dop.stride + dop.k) * ((8 - 1) * dop.stride + dop.k);
        lds_bytes = ((NPh_ * a.T0S + 15) & ~15) + ((NPh_ * 48 + 15) & ~15) + 64 * FB_DST;
      }
      { const int rc = launch_fused_block(a, FusedLaunch{dop.k, dop.stride, s.nbp, ex, mdw, nt3, false, dw64, lds_bytes, grid.x}, st); if (rc) return rc; }
      break;
    }
    case F_BAND: {
      if (boff != 0) { set_error("fused_sepconv_band: sub-batch streams are not supported (VBT_SUBSTREAMS)"); return VBT_ERR_ARG; }
      MultiTiles mt;
      int acc = 0;
      if (s.members.empty()) {
        launch_band_one(s.ba, (unsigned)(B * s.band_tiles), s.lds_bytes, st);
        break;
      } else {
        mt.n = (int)s.members.size();
        for (int i = 0; i < mt.n; i++) {
          mt.start[i] = acc;
          acc += B * s.members[i].band_tiles;
        }
      }
      mt.start[mt.n] = acc;
      launch_band_multi(s.d_band, mt, s.members[0].bd_args.C, (unsigned)acc, s.lds_bytes, st);
      break;
    }
    case F_EXPDW: {
      const OpRec& eop = m->ops[s.e_op];
      const OpRec& dop = m->ops[s.d_op];
      // variant: chunks per workgroup on the first form of the kernel, 100 + chunks per workgroup on the second with 8 waves, 200 + chunks
      // with 16 waves; -1: heuristic default
      // VBT_XD_VARIANT (tests): that variant for every step that supports it, whatever the plan says
      static const int xd_force = getenv("VBT_XD_VARIANT") ? atoi(getenv("VBT_XD_VARIANT")) : -100;
      int variant = s.variant;
      if (xd_force != -100 && (xd_force < 100 || (s.xd2_ok && (xd_force < 200 ? s.xd2_gpw > 0 : s.xd2_gpw16 > 0)))) variant = std::min(xd_force, xd_force / 100 * 100 + s.xd.nchunks);
      if (s.xd2_ok && (variant >= 100 || variant < 0)) {
        if (variant >= 200 && s.xd2_gpw16 == 0) { set_error("expand + depthwise: plan asks for the 16-wave form on a step that does not support it"); return VBT_ERR_ARG; }
        if (variant >= 100 && variant < 200 && s.xd2_gpw == 0) { set_error("expand + depthwise: plan asks for the 8-wave form on a step that does not support it"); return VBT_ERR_ARG; }
        const int nw = (variant >= 200 || (variant < 0 && s.xd2_gpw16 > 0)) ? 16 : 8;
        ExpDw2Args a = s.xd2;
        a.x = TP(eop.inputs[0]);
        a.out = out;
        a.cpw = variant >= 100 ? variant % 100 : std::max(1, (a.nchunks * a.nbands * B + 1023) / 1024);   // default: about four workgroups per CU
        a.cpw = std::min(a.cpw, a.nchunks);
        const int ngroups = (a.nchunks + a.cpw - 1) / a.cpw;
        const int rc = launch_expdw2(a, dop.k, dop.stride, (a.Cin + 63) / 64, nw, nw == 16 ? s.xd2_gpw16 : s.xd2_gpw, (unsigned)(B * ngroups * a.nbands), s.xd2_lds, st);
        if (rc) return rc;
        break;
      }
      if (variant >= 100) { set_error("expand + depthwise: plan asks for the second kernel form on a step that does not support it"); return VBT_ERR_ARG; }
      ExpDwArgs a = s.xd;
      a.x = TP(eop.inputs[0]);
      a.out = out;
      a.cpw = variant > 0 ? variant : std::max(1, (a.nchunks * a.nbands * B + 511) / 512);   // default: about two workgroups per CU
      const int ngroups = (a.nchunks + a.cpw - 1) / a.cpw;
      const int KS64 = (a.Cin + 63) / 64;
      dim3 grid((unsigned)(B * ngroups * a.nbands));
      launch_expdw(a, dop.k, dop.stride, KS64, grid.x, s.lds_bytes, st);
      break;
    }
    case F_STEMBLK: {
      StemBlockArgs a = s.sb;
      a.frames = frames;
      a.out = out;
      launch_stem_block(a, a.rqs.full && a.rqd.full && a.rqp.full, (unsigned)((long)B * a.tiles_x * a.tiles_y), st);
      break;
    }
    case F_POST: {
      PostArgs p;
      int base = 0;
      p.C = m->hdr.num_classes > 0 ? m->hdr.num_classes : 1;
      for (int l = 0; l < 5; l++) {
        const TensorRec& tc = m->tensors[op.inputs[l]];
        p.cls[l] = TP(op.inputs[l]);
        p.box[l] = TP(op.inputs[5 + l]);
        p.base[l] = base;
        base += tc.h * tc.w * tc.c / p.C;
      }
      p.base[5] = base;
      p.anchors = m->d_anchors;
      p.tables = m->d_luts;
      p.A = m->hdr.num_anchors;
      p.max_det = m->hdr.max_detections;
      p.iou_thr = m->hdr.nms_iou_threshold;
      int qmin = 128;   // in rank bytes
      for (int q = 127; q >= -128; q--)
        if (m->post_tables_host[q + 128] >= m->hdr.nms_score_threshold) qmin = q; else break;
      p.qmin = qmin;
      const int lds = post_lds_bytes(p.A);
      if (lds > 64 * 1024) VBT_LDS_OPT_IN(postprocess_kernel);
      if (lds > 160 * 1024 || p.A > 65535) { set_error("decode + NMS: %d anchors do not fit the kernel (LDS %d bytes, 16-bit anchor index)", p.A, lds); return VBT_ERR_CAPACITY; }
      postprocess_kernel<<<dim3((unsigned)B), POST_THREADS, lds, st>>>(p, boxes, scores, classes, counts);
      break;
    }
  }
  return VBT_OK;
}

// VBT_AUTOTUNE_CONCURRENCY=n (default 1): time each candidate with n copies in flight on n streams (same buffers, same
// results) and rank by time per copy, i.e. by throughput under contention - what a pipelined caller (Pipeline depth n)
// experiences - instead of by isolated latency.
static double time_step(vbt_model* m, const Step& s, int B, int reps) {
  static int nconc = -1;
  static hipStream_t cs[4] = {nullptr, nullptr, nullptr, nullptr};
  if (nconc < 0) {
    const char* e = getenv("VBT_AUTOTUNE_CONCURRENCY");
    nconc = e ? std::max(1, std::min(4, atoi(e))) : 1;
    if (nconc > 1)
      for (int i = 0; i < nconc; i++) (void)hipStreamCreateWithFlags(&cs[i], hipStreamNonBlocking);
  }
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return 1e30;
  float ms = 1e30f;
  if (nconc <= 1) {
    launch_step(m, s, B, nullptr, m->frames_stage, m->out_boxes, m->out_scores, m->out_classes, m->out_counts);
    (void)hipEventRecord(e0, nullptr);
    for (int r = 0; r < reps; r++)
      launch_step(m, s, B, nullptr, m->frames_stage, m->out_boxes, m->out_scores, m->out_classes, m->out_counts);
    (void)hipEventRecord(e1, nullptr);
    if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) ms = 1e30f;
    ms /= reps;
  } else {
    (void)hipDeviceSynchronize();
    for (int i = 0; i < nconc; i++)
      launch_step(m, s, B, cs[i], m->frames_stage, m->out_boxes, m->out_scores, m->out_classes, m->out_counts);
    (void)hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; r++)
      for (int i = 0; i < nconc; i++)
        launch_step(m, s, B, cs[i], m->frames_stage, m->out_boxes, m->out_scores, m->out_classes, m->out_counts);
    (void)hipDeviceSynchronize();
    ms = (float)(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / (reps * nconc));
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return ms;
}

// Plan-time autotuning: every alternative computes bit-identical tensors, so only speed is at stake.
// Kernel variants the planner offers for a step: what the autotuner times, and all a plan file may select (anything else in a file
// refuses the file: load_plan).  -1 = the launcher's own default.
static std::vector<int> candidate_variants(const vbt_model* m, const Step& st) {
  std::vector<int> cand{-1};
  const OpRec& op = m->ops[st.op];
  if (st.family == F_DW) {
    cand = {0};
    for (int r : {1, 2, 4, 8, 16})
      if (r <= m->tensors[op.output].h) cand.push_back(r);
    if (m->tensors[op.output].c % 8 == 0) { cand.push_back(100); cand.push_back(101); }
  } else if (st.family == F_PW && st.KS64 <= 4) {
    cand = {0, 1};
  } else if (st.family == F_PW) {
    cand = {-1, 2, 3, 4, 5, 6};
  } else if (st.family == F_MBCONV || st.family == F_SEPCONV || st.family == F_NODE) {
    cand = {0, 1, 3};   // VALU dw, matrix-pipe dw, matrix-pipe dw + half-height tile
    if (image_geom(m, st).ok) cand.push_back(5);  // one workgroup per image
    if (st.family == F_MBCONV && st.fa.nch3 > 0 && st.nbp <= 2 && st.fa.KSe >= 1 && st.fa.KSe <= 4) { cand.push_back(9); cand.push_back(11); }  // 48-channel chunks
    if (st.family == F_MBCONV && st.nbp <= 2 && (st.fa.KSe == 1 || st.fa.KSe == 2) && ppw2_fits(m, st)) {   // 128-pixel tiles
      cand.push_back(17);
      if (st.fa.nch3 > 0) cand.push_back(25);
    }
  } else if (st.family == F_MULTI) {
    cand = {0, 1};
  } else if (st.family == F_EXPDW) {
    cand.clear();
    for (int cpw : {1, 2, 3, 4, 6})
      if (cpw <= st.xd.nchunks) cand.push_back(cpw);
    if (st.xd2_ok)
      for (int cpw : {1, 2, 3, 4, 6})
        if (cpw <= st.xd.nchunks) {
          if (st.xd2_gpw > 0) cand.push_back(100 + cpw);
          if (st.xd2_gpw16 > 0) cand.push_back(200 + cpw);
        }
  }
  return cand;
}

static void autotune(vbt_model* m) {
  const int B = (m->max_batch + m->n_sub - 1) / m->n_sub, reps = 4;  // the batch one stream actually sees
  for (Group& g : m->groups) {
    bool single = g.alts.size() == 1 && g.alts[0].steps.size() == 1;
    if (single) {
      int f = g.alts[0].steps[0].family;
      if (f != F_DW && f != F_PW) continue;  // nothing to choose
    }
    for (Alt& a : g.alts) {
      a.ms = 0;
      for (Step& st : a.steps) {
        const std::vector<int> cand = candidate_variants(m, st);
        double best = 1e30;
        int bestv = -1;
        for (int v : cand) {
          Step t = st;
          t.variant = v;
          double ms = time_step(m, t, B, reps);
          if (getenv("VBT_AUTOTUNE_VERBOSE") && cand.size() > 1 && atoi(getenv("VBT_AUTOTUNE_VERBOSE")) > 1)
            fprintf(stderr, "[autotune]   op %d %s v%d %.1fus\n", st.op, kFamilyName[st.family], v, ms * 1e3);
          if (ms < best) { best = ms; bestv = v; }
        }
        st.variant = bestv;
        st.macs_per_frame = st.macs_per_frame;  // (unchanged)
        st.tuned_ms = best;
        a.ms += best;
      }
    }
    int bi = 0;
    for (size_t i = 1; i < g.alts.size(); i++)
      if (g.alts[i].ms < g.alts[bi].ms) bi = (int)i;
    if (const char* pe = getenv("VBT_PREFER_IMAGE")) {  // experiment: whole-image kernel on maps of at most this many pixels
      for (size_t i = 0; i < g.alts.size(); i++)
        if (g.alts[i].steps.size() == 1 && g.alts[i].steps[0].family == F_MBCONV && image_geom(m, g.alts[i].steps[0]).ok &&
            g.alts[i].steps[0].fa.H * g.alts[i].steps[0].fa.W <= atoi(pe)) {
          bi = (int)i;
          g.alts[i].steps[0].variant = 5;
        }
    }
    g.chosen = bi;
    if (getenv("VBT_AUTOTUNE_VERBOSE")) {
      const Step& f = g.alts[0].steps[0];
      const TensorRec& to = m->tensors[m->ops[g.alts[0].steps.back().op].output];
      fprintf(stderr, "[autotune] op %3d.. out %3dx%3dx%4d :", f.op, to.h, to.w, to.c);
      for (size_t i = 0; i < g.alts.size(); i++) {
        fprintf(stderr, " alt%zu%s %.1fus(", i, (int)i == bi ? "*" : "", g.alts[i].ms * 1e3);
        for (const Step& st : g.alts[i].steps) fprintf(stderr, "%s:v%d=%.1f ", kFamilyName[st.family], st.variant, st.tuned_ms * 1e3);
        fprintf(stderr, ")");
      }
      fprintf(stderr, "\n");
    }
  }
  (void)hipDeviceSynchronize();
}

// Plan cache.  Format 2 (written): "VBTPLAN2 <ngroups>" then per group "<chosen alternative> <nsteps> <family>:<variant> ..." - the
// kernel family of every step of the chosen alternative by NAME, so that a file tuned for another build of the planner (an
// alternative added, removed or re-ordered: the bare indices of format 1 would still load and silently select other kernels) is
// refused and the plan re-tuned.  Format 1 ("<ngroups>" then "<chosen> <nsteps> <variant>...") is still read - the group and step
// counts are all it can be checked against - and re-written in format 2 when VBT_PLAN_CONVERT is set.
static bool load_plan(vbt_model* m, const char* path) {
  // the shape a file may select from: this library's groups, their alternatives, the family of every step and the variants the planner
  // offers for it (container_parse.h: parse_plan_file refuses everything else, and the model is tuned afresh)
  PlanShape shape;
  for (const Group& g : m->groups) {
    std::vector<std::vector<PlanStepShape>> alts;
    for (const Alt& a : g.alts) {
      std::vector<PlanStepShape> steps;
      for (const Step& st : a.steps) {
        PlanStepShape ps;
        ps.family = kFamilyName[st.family];
        ps.variants = candidate_variants(m, st);
        // the fused tile kernels read their variant as a set of flags (matrix-pipe depthwise, half-height tile, 48-channel chunks,
        // 128-pixel tiles ...), each guarded by its own fit test in launch_step: plans searched under load (tools/tune_under_load.py)
        // hold combinations the isolated autotuner does not time
        if (st.family == F_MBCONV || st.family == F_SEPCONV || st.family == F_NODE)
          for (int v = 0; v < 32; v++)
            if (std::find(ps.variants.begin(), ps.variants.end(), v) == ps.variants.end()) ps.variants.push_back(v);
        if (std::find(ps.variants.begin(), ps.variants.end(), -1) == ps.variants.end()) ps.variants.push_back(-1);
        if (std::find(ps.variants.begin(), ps.variants.end(), st.variant) == ps.variants.end()) ps.variants.push_back(st.variant);   // the heuristic plan's own choice
        steps.push_back(ps);
      }
      alts.push_back(steps);
    }
    shape.groups.push_back(alts);
  }
  std::vector<PlanChoice> sel;
  std::string note;
  if (!parse_plan_file(path, shape, &sel, &note)) {
    if (note != "no such file") fprintf(stderr, "[vbt] plan %s: %s - plan refused, re-tuning\n", path, note.c_str());
    return false;
  }
  for (size_t gi = 0; gi < m->groups.size(); gi++) {
    m->groups[gi].chosen = sel[gi].chosen;
    Alt& a = m->groups[gi].alts[(size_t)sel[gi].chosen];
    for (size_t i = 0; i < a.steps.size(); i++) a.steps[i].variant = sel[gi].variants[i];
  }
  return true;
}
static void save_plan(const vbt_model* m, const char* path) {
  FILE* f = fopen(path, "w");
  if (!f) return;
  fprintf(f, "VBTPLAN2 %d\n", (int)m->groups.size());
  for (const Group& g : m->groups) {
    const Alt& a = g.alts[g.chosen];
    fprintf(f, "%d %d", g.chosen, (int)a.steps.size());
    for (const Step& st : a.steps) fprintf(f, " %s:%d", kFamilyName[st.family], st.variant);
    fprintf(f, "\n");
  }
  fclose(f);
}

static int enqueue_forward(vbt_model* m, const uint8_t* frames_dev, int B, hipStream_t st, float* boxes, float* scores,
                           float* classes, int* counts, hipEvent_t* evs) {
  const int nsub = (!evs && m->n_sub > 1 && B >= 2 * m->n_sub) ? m->n_sub : 1;
  if (nsub == 1) {
    int i = 0;
    for (const Step& s : m->steps) {
      if (evs) (void)hipEventRecord(evs[i], st);
      int rc = launch_step(m, s, B, st, frames_dev, boxes, scores, classes, counts);
      if (rc) return rc;
      i++;
    }
    if (evs) (void)hipEventRecord(evs[i], st);
  } else {
    // Independent sub-batches on side streams: the many small, latency-bound kernels of one sub-batch overlap
    // with the other's.  Fork from / join into the caller's stream with events.
    (void)hipEventRecord(m->ev_fork, st);
    const int per = (B + nsub - 1) / nsub;
    for (int k = 0; k < nsub; k++) {
      const int b0 = k * per, bk = std::min(per, B - b0);
      if (bk <= 0) break;
      hipStream_t ss = m->sub_streams[k];
      (void)hipStreamWaitEvent(ss, m->ev_fork, 0);
      for (const Step& s : m->steps) {
        int rc = launch_step(m, s, bk, ss, frames_dev, boxes, scores, classes, counts, b0);
        if (rc) return rc;
      }
      (void)hipEventRecord(m->ev_join[k], ss);
      (void)hipStreamWaitEvent(st, m->ev_join[k], 0);
    }
  }
  VBT_HIP_CHECK(hipGetLastError());
  m->last_B = B;
  return VBT_OK;
}

// Forward = eager launches, or (small batches: the 120-odd launches are host-bound) replay of a captured hipGraph.
static int forward(vbt_model* m, const uint8_t* frames_dev, int B, hipStream_t st, float* boxes, float* scores, float* classes,
                   int* counts) {
  RoctxRange range("vbt:detect");
  if (m->pool_dirty) { const int rc = flush_uploads(m); if (rc) return rc; }   // (never inside a stream capture)
  if (B > m->graph_max_batch || !m->cap_stream) return enqueue_forward(m, frames_dev, B, st, boxes, scores, classes, counts, nullptr);
  vbt_model::GraphKey key{frames_dev, boxes, scores, classes, counts, B};
  auto it = m->graphs.find(key);
  if (it == m->graphs.end()) {
    if (m->graphs.size() >= 256) {  // bounded cache
      for (auto& kv : m->graphs) (void)hipGraphExecDestroy(kv.second);
      m->graphs.clear();
    }
    hipGraph_t g = nullptr;
    hipGraphExec_t ge = nullptr;
    if (!m->ran_eager) {
      // The first forward of a model never runs inside a capture: the per-device LDS opt-ins of its kernels (hipFuncSetAttribute) and any
      // failure they report happen here, eagerly, on the caller's stream (same buffers, same results as the replay that follows).
      const int rc0 = enqueue_forward(m, frames_dev, B, st, boxes, scores, classes, counts, nullptr);
      if (rc0) return rc0;
      m->ran_eager = true;
    }
    VBT_HIP_CHECK(hipStreamBeginCapture(m->cap_stream, hipStreamCaptureModeThreadLocal));
    int rc = enqueue_forward(m, frames_dev, B, m->cap_stream, boxes, scores, classes, counts, nullptr);
    hipError_t e = hipStreamEndCapture(m->cap_stream, &g);
    if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
    if (e != hipSuccess) { set_error("hipStreamEndCapture failed: %s", hipGetErrorString(e)); return VBT_ERR_HIP; }
    e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) { set_error("hipGraphInstantiate failed: %s", hipGetErrorString(e)); return VBT_ERR_HIP; }
    it = m->graphs.emplace(key, ge).first;
  }
  VBT_HIP_CHECK(hipGraphLaunch(it->second, st));
  m->last_B = B;
  return VBT_OK;
}

}  // namespace vbt

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
extern "C" {

const char* vbt_last_error(void) { return vbt::g_err; }

int vbt_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int vbt_model_create(const char* path, int device, int max_batch, vbt_model** out) {
  const char* nf = getenv("VBT_FUSION_FLAGS");  // bit0: no fusion, bit1: no MBConv fusion, bit2: no SeparableConv fusion
  return vbt_model_create_ex(path, device, max_batch, nf ? atoi(nf) : VBT_MODEL_DEFAULT_FLAGS, out);
}

int vbt_model_tensor_materialized(const vbt_model* m, int id) {
  if (!m || id < 0 || id >= (int)m->tensors.size()) { set_error("bad tensor id"); return VBT_ERR_ARG; }
  return m->materialized[id] ? 1 : 0;
}

int vbt_model_num_launches(const vbt_model* m) { return m ? (int)m->steps.size() : VBT_ERR_ARG; }

int vbt_model_create_ex(const char* path, int device, int max_batch, int flags, vbt_model** out) {
  if (!path || !out || max_batch < 1) { set_error("vbt_model_create: bad argument"); return VBT_ERR_ARG; }
  *out = nullptr;
  vbt_model* m = new vbt_model();
  {
    // reader + structural validation (container_parse.h): every index the planner and the kernels follow is in range before they see it
    ContainerData cd;
    std::string why;
    if (!read_container(path, &cd, &why)) { delete m; set_error("%s", why.c_str()); return VBT_ERR_IO; }
    m->hdr = cd.hdr;
    m->tensors.swap(cd.tensors);
    m->ops.swap(cd.ops);
    m->blob.swap(cd.blob);
  }
  m->device = device;
  m->max_batch = max_batch;
  m->flags = flags;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
    delete m;
    set_error("vbt_model_create: HIP device %d not available (%d visible) - the HIP path has no CPU fallback", device, ndev);
    return VBT_ERR_HIP;
  }
  int rc = VBT_OK;
  auto fail = [&](int code) { vbt_model_destroy(m); return code; };
  if (hipSetDevice(device) != hipSuccess) { set_error("hipSetDevice(%d) failed", device); return fail(VBT_ERR_HIP); }
  // activation arena: every graph tensor keeps its own [max_batch][h][w][c] int8 buffer
  size_t total = 0;
  m->telems.resize(m->tensors.size());
  std::vector<size_t> off(m->tensors.size());
  for (size_t i = 0; i < m->tensors.size(); i++) {
    const TensorRec& t = m->tensors[i];
    m->telems[i] = (size_t)t.h * t.w * t.c;
    off[i] = total;
    size_t bytes = (int)i == m->hdr.input_tensor ? 0 : m->telems[i] * max_batch;
    total += (bytes + 255) / 256 * 256 + 256;
  }
  if (fenced_malloc(m, (void**)&m->arena, total + 4096) != hipSuccess) { set_error("hipMalloc(%zu) for activations failed", total); return fail(VBT_ERR_HIP); }
  (void)hipMemset(m->arena, 0, total + 4096);
  m->tptr.resize(m->tensors.size());
  for (size_t i = 0; i < m->tensors.size(); i++) m->tptr[i] = m->arena + off[i];
  size_t fbytes = (size_t)max_batch * m->hdr.image_size * m->hdr.image_size * 3;
  const int md = m->hdr.max_detections;
  m->out_bytes = (size_t)max_batch * (md * 24 + 4);
  if (fenced_malloc(m, (void**)&m->frames_stage, fbytes + 64) != hipSuccess || hipMalloc((void**)&m->out_boxes, m->out_bytes) != hipSuccess ||
      hipHostMalloc((void**)&m->out_host, m->out_bytes, hipHostMallocDefault) != hipSuccess) {
    set_error("hipMalloc for staging buffers failed");
    return fail(VBT_ERR_HIP);
  }
  m->out_scores = m->out_boxes + (size_t)max_batch * md * 4;
  m->out_classes = m->out_scores + (size_t)max_batch * md;
  m->out_counts = (int*)(m->out_classes + (size_t)max_batch * md);
  if ((rc = build_plan(m)) != VBT_OK) return fail(rc);
  for (const OpRec& op : m->ops)
    if (op.type == OP_POSTPROCESS) {
      std::vector<float> an((const float*)(m->blob.data() + op.aux_off), (const float*)(m->blob.data() + op.aux_off) + (size_t)m->hdr.num_anchors * 4);
      if ((size_t)op.aux2_off + VBT_POST_TABLE_BYTES > m->blob.size()) { set_error("post-process tables truncated"); return fail(VBT_ERR_IO); }
      std::vector<unsigned char> lut(m->blob.data() + op.aux2_off, m->blob.data() + op.aux2_off + VBT_POST_TABLE_BYTES);
      const float* sv = (const float*)(lut.data() + 6144);
      if (sv[0] != sv[1] || sv[2] != sv[3]) { set_error("post-process: y_scale != x_scale or h_scale != w_scale"); return fail(VBT_ERR_ARG); }
      {
        // The stored tables are checked, not trusted: every entry is derived again from the quantisation of the class and
        // box tensors (XNNPACK's x8 LOGISTIC table in float32 with glibc expf; DEQUANTIZE as one float32 product; the
        // decode's divisions and exp() in double, detection_postprocess.cc) and a container that differs is refused.
        const int nl = op.n_inputs / 2;
        const TensorRec& tc = m->tensors[op.inputs[0]];
        const TensorRec& tb = m->tensors[op.inputs[nl]];
        for (int l = 1; l < nl; l++) {   // CONCATENATION: one quantisation for all of its inputs
          const TensorRec &c2 = m->tensors[op.inputs[l]], &b2 = m->tensors[op.inputs[nl + l]];
          if (c2.scale != tc.scale || c2.zero_point != tc.zero_point || b2.scale != tb.scale || b2.zero_point != tb.zero_point) {
            set_error("post-process: head outputs of level %d are quantised differently from level 0", l);
            return fail(VBT_ERR_ARG);
          }
        }
        const float* st_score = (const float*)lut.data();
        const float* st_box = st_score + 256;
        const double* st_dq = (const double*)(lut.data() + 2048);
        const double* st_ex = st_dq + 256;
        for (int q = -128; q < 128; q++) {
          const float x = tc.scale * (float)(q - tc.zero_point);
          float y = 256.0f / (1.0f + expf(-x));
          y = y < 0.0f ? 0.0f : (y > 255.0f ? 255.0f : y);
          const float want_score = (1.0f / 256.0f) * (float)lrintf(y);
          const float want_box = tb.scale * (float)(q - tb.zero_point);
          const double want_dq = (double)want_box / (double)sv[0];
          const double want_ex = exp((double)want_box / (double)sv[2]);
          if (st_score[q + 128] != want_score || st_box[q + 128] != want_box || st_dq[q + 128] != want_dq || st_ex[q + 128] != want_ex) {
            set_error("post-process: stored table entry %d differs from the one derived from the tensor scales", q);
            return fail(VBT_ERR_ARG);
          }
        }
      }
      // device tables: scores re-indexed by rank byte, decode tables, class byte -> rank byte
      std::vector<unsigned char> dev(1024 + 2048 + 2048 + 256, 0);
      const float* score = (const float*)lut.data();
      float* score_by_rank = (float*)dev.data();
      signed char* rank = (signed char*)(dev.data() + 5120);
      for (int q = 1; q < 256; q++)
        if (score[q] < score[q - 1]) { set_error("post-process: score table is not monotone"); return fail(VBT_ERR_ARG); }
      int r = 127;   // highest class byte gets rank 127; a strictly lower score steps the rank down
      for (int q = 255; q >= 0; q--) {
        if (q < 255 && score[q] != score[q + 1]) r--;
        rank[q] = (signed char)r;
        score_by_rank[r + 128] = score[q];
      }
      for (int i = -128; i < r; i++) score_by_rank[i + 128] = -1.0f;   // unused rank bytes: below every threshold
      memcpy(dev.data() + 1024, lut.data() + 2048, 4096);
      if ((rc = upload(m, an, &m->d_anchors)) || (rc = upload(m, dev, &m->d_luts))) return fail(rc);
      m->post_tables_host.assign(score_by_rank, score_by_rank + 256);
    }
  {
    const char* ns = getenv("VBT_SUBSTREAMS");
    int want = ns ? atoi(ns) : 1;  // side streams measured no gain on MI355X at B = 64 (the GPU is busy, not starved)
    if (m->flags & VBT_MODEL_SINGLE_STREAM) want = 1;
    m->n_sub = std::max(1, std::min(want, 4));
    if (max_batch < 2 * m->n_sub) m->n_sub = 1;
    if (m->n_sub > 1) {
      bool ok = hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming) == hipSuccess;
      for (int k = 0; k < m->n_sub && ok; k++)
        ok = hipStreamCreateWithFlags(&m->sub_streams[k], hipStreamNonBlocking) == hipSuccess &&
             hipEventCreateWithFlags(&m->ev_join[k], hipEventDisableTiming) == hipSuccess;
      if (!ok) { set_error("cannot create side streams"); return fail(VBT_ERR_HIP); }
    }
  }
  {
    const char* gm = getenv("VBT_GRAPH_MAX_BATCH");
    m->graph_max_batch = (m->flags & VBT_MODEL_NO_GRAPH) ? 0 : (gm ? atoi(gm) : 8);
    if (m->graph_max_batch > 0 && hipStreamCreateWithFlags(&m->cap_stream, hipStreamNonBlocking) != hipSuccess) m->cap_stream = nullptr;
  }
  if (!(m->flags & VBT_MODEL_NO_AUTOTUNE)) {
    // VBT_PLAN_FILE: reuse a previously tuned plan (keeps profiled and un-profiled runs on the same kernels)
    const char* pf = getenv("VBT_PLAN_FILE");
    char path[1024];
    if (pf) snprintf(path, sizeof(path), "%s.b%d.f%d", pf, max_batch, m->flags);
    if (!pf || !load_plan(m, path)) {
      autotune(m);
      if (pf) save_plan(m, path);
    } else if (getenv("VBT_PLAN_CONVERT")) {
      save_plan(m, path);       // a format-1 file comes back in format 2 (same choices, kernel families by name)
    }
  }
  finalize_plan(m);
  if ((rc = flush_uploads(m)) != VBT_OK) return fail(rc);
  *out = m;
  return VBT_OK;
}

void vbt_model_destroy(vbt_model* m) {
  if (!m) return;
  for (int k = 0; k < 4; k++) {
    if (m->sub_streams[k]) (void)hipStreamDestroy(m->sub_streams[k]);
    if (m->ev_join[k]) (void)hipEventDestroy(m->ev_join[k]);
  }
  if (m->ev_fork) (void)hipEventDestroy(m->ev_fork);
  for (auto& kv : m->graphs) (void)hipGraphExecDestroy(kv.second);
  if (m->cap_stream) (void)hipStreamDestroy(m->cap_stream);
  for (void* p : m->owned) (void)hipFree(p);
  (void)hipFree(m->out_boxes);  // (one block: boxes | scores | classes | counts); arena and frames_stage are in `owned`
  if (m->out_host) (void)hipHostFree(m->out_host);
  delete m;
}

int vbt_model_input_shape(const vbt_model* m, int shape[4]) {
  if (!m || !shape) { set_error("bad argument"); return VBT_ERR_ARG; }
  shape[0] = m->max_batch; shape[1] = m->hdr.image_size; shape[2] = m->hdr.image_size; shape[3] = 3;
  return VBT_OK;
}
int vbt_model_num_tensors(const vbt_model* m) { return m ? (int)m->tensors.size() : VBT_ERR_ARG; }
int vbt_model_num_ops(const vbt_model* m) { return m ? (int)m->ops.size() : VBT_ERR_ARG; }
int vbt_model_tensor_shape(const vbt_model* m, int id, int shape[3]) {
  if (!m || !shape || id < 0 || id >= (int)m->tensors.size()) { set_error("bad tensor id"); return VBT_ERR_ARG; }
  shape[0] = m->tensors[id].h; shape[1] = m->tensors[id].w; shape[2] = m->tensors[id].c;
  return VBT_OK;
}

int vbt_detect_async(vbt_model* m, const uint8_t* frames_dev, int B, void* stream, float* boxes, float* scores, float* classes,
                     int32_t* counts) {
  if (!m || !frames_dev || !boxes || !scores || !classes || !counts) { set_error("vbt_detect_async: NULL argument"); return VBT_ERR_ARG; }
  if (B < 1 || B > m->max_batch) { set_error("vbt_detect: batch %d outside 1..%d", B, m->max_batch); return VBT_ERR_CAPACITY; }
  return forward(m, frames_dev, B, (hipStream_t)stream, boxes, scores, classes, counts);
}

int vbt_detect(vbt_model* m, const uint8_t* frames, int B, int frames_on_device, void* stream, float* boxes, float* scores,
               float* classes, int32_t* counts, int outputs_on_device) {
  if (!m || !frames || !boxes || !scores || !classes || !counts) { set_error("vbt_detect: NULL argument"); return VBT_ERR_ARG; }
  if (B < 1 || B > m->max_batch) { set_error("vbt_detect: batch %d outside 1..%d", B, m->max_batch); return VBT_ERR_CAPACITY; }
  hipStream_t st = (hipStream_t)stream;
  VBT_HIP_CHECK(hipSetDevice(m->device));
  const uint8_t* fd = frames;
  size_t fbytes = (size_t)B * m->hdr.image_size * m->hdr.image_size * 3;
  if (!frames_on_device) {
    VBT_HIP_CHECK(hipMemcpyAsync(m->frames_stage, frames, fbytes, hipMemcpyHostToDevice, st));
    fd = m->frames_stage;
  }
  float *db = boxes, *ds = scores, *dc = classes;
  int* dn = counts;
  if (!outputs_on_device) { db = m->out_boxes; ds = m->out_scores; dc = m->out_classes; dn = m->out_counts; }
  int rc = forward(m, fd, B, st, db, ds, dc, dn);
  if (rc) return rc;
  if (!outputs_on_device) {
    // one copy of the staging block into its pinned mirror, one synchronisation of this stream, then the caller's arrays are
    // filled on the host (four copies into pageable memory used to be four staged transfers)
    const int md = m->hdr.max_detections;
    const size_t mb = (size_t)m->max_batch;
    const unsigned char* d = (const unsigned char*)m->out_boxes;
    unsigned char* h = m->out_host;
    if (B == m->max_batch) {
      VBT_HIP_CHECK(hipMemcpyAsync(h, d, m->out_bytes, hipMemcpyDeviceToHost, st));
    } else {   // only the B-frame prefix of each of the four tensors (an interpreter created for 256 frames and called with 1 moved 615 KB)
      const size_t off[4] = {0, mb * md * 16, mb * md * 20, mb * md * 24}, len[4] = {(size_t)B * md * 16, (size_t)B * md * 4, (size_t)B * md * 4, (size_t)B * 4};
      for (int i = 0; i < 4; i++) VBT_HIP_CHECK(hipMemcpyAsync(h + off[i], d + off[i], len[i], hipMemcpyDeviceToHost, st));
    }
    VBT_HIP_CHECK(hipStreamSynchronize(st));
    memcpy(boxes, h, (size_t)B * md * 16);
    memcpy(scores, h + mb * md * 16, (size_t)B * md * 4);
    memcpy(classes, h + mb * md * 20, (size_t)B * md * 4);
    memcpy(counts, h + mb * md * 24, (size_t)B * 4);
  }
  return VBT_OK;
}

int vbt_model_read_tensor(vbt_model* m, int id, int B, int8_t* host_out) {
  if (!m || !host_out || id < 0 || id >= (int)m->tensors.size() || id == m->hdr.input_tensor) { set_error("bad tensor id"); return VBT_ERR_ARG; }
  if (B < 1 || B > m->max_batch) { set_error("bad batch"); return VBT_ERR_CAPACITY; }
  if (!m->materialized[id]) { set_error("tensor %d lives only in LDS (fused away); create the model with VBT_MODEL_NO_FUSION to read it", id); return VBT_ERR_STATE; }
  VBT_HIP_CHECK(hipDeviceSynchronize());
  VBT_HIP_CHECK(hipMemcpy(host_out, m->tptr[id], m->telems[id] * B, hipMemcpyDeviceToHost));
  return VBT_OK;
}

// A HIP stream gets its hardware queue at its FIRST command, round-robin over GPU_MAX_HW_QUEUES (rocprofv3 Queue_Id).  Streams
// drawn from a framework's pool may have been used before, so a pipeline's streams can land on one queue and serialise
// (measured: 89 k -> 58 k frames/s).  Streams created here run one empty launch at once: streams created back to back sit on
// consecutive queues.
__global__ void stream_touch_kernel() {}
int vbt_stream_create(int device, void** stream_out) {
  if (!stream_out) { set_error("vbt_stream_create: NULL argument"); return VBT_ERR_ARG; }
  VBT_HIP_CHECK(hipSetDevice(device));
  hipStream_t st = nullptr;
  VBT_HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  stream_touch_kernel<<<1, 64, 0, st>>>();
  hipError_t e = hipStreamSynchronize(st);
  if (e != hipSuccess) { (void)hipStreamDestroy(st); set_error("vbt_stream_create: %s", hipGetErrorString(e)); return VBT_ERR_HIP; }
  *stream_out = (void*)st;
  return VBT_OK;
}
// Do two streams share a hardware queue?  A single-wave kernel that spins for `us` microseconds on each: side by side they
// take `us`, on one in-order queue 2 x `us`.  (The queue of a stream cannot be queried; GPU otherwise idle when called.)
__global__ void stream_spin_kernel(long ticks) {
  const long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
int vbt_streams_share_queue(void* a, void* b, int us, int* shared) {
  if (!a || !b || !shared || us < 20 || us > 100000) { set_error("vbt_streams_share_queue: bad argument"); return VBT_ERR_ARG; }
  const long ticks = (long)us * 100;   // wall_clock64: 100 MHz
  VBT_HIP_CHECK(hipStreamSynchronize((hipStream_t)a));
  VBT_HIP_CHECK(hipStreamSynchronize((hipStream_t)b));
  auto t0 = std::chrono::steady_clock::now();
  stream_spin_kernel<<<1, 64, 0, (hipStream_t)a>>>(ticks);
  stream_spin_kernel<<<1, 64, 0, (hipStream_t)b>>>(ticks);
  VBT_HIP_CHECK(hipStreamSynchronize((hipStream_t)a));
  VBT_HIP_CHECK(hipStreamSynchronize((hipStream_t)b));
  const double el = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  *shared = el > 1.6 * us ? 1 : 0;
  return VBT_OK;
}
int vbt_stream_destroy(void* stream) {
  if (!stream) return VBT_OK;
  VBT_HIP_CHECK(hipStreamDestroy((hipStream_t)stream));
  return VBT_OK;
}

}  // extern "C"
namespace vbt {
int resize_frames_dev(const uint8_t* src_dev, int B, int H, int W, uint8_t* dst_dev, int h, int w, int swap_rb, int compact, hipStream_t st) {
  if (compact && H < 2) { set_error("resize: compact rows need a source of at least two rows"); return VBT_ERR_ARG; }
  const long total = (long)B * h * w;
  const float sy = (float)H / (float)h, sx = (float)W / (float)w;
  resize_bilinear_kernel<<<dim3((unsigned)((total + 255) / 256)), 256, 0, st>>>(src_dev, dst_dev, total, H, W, h, w, sy, sx, swap_rb, compact);
  VBT_HIP_CHECK(hipGetLastError());
  return VBT_OK;
}
}  // namespace vbt
extern "C" {

int vbt_resize_frames(const uint8_t* src, int B, int H, int W, int src_on_device, uint8_t* dst, int h, int w, int dst_on_device,
                      int swap_rb, int device, void* stream) {
  if (!src || !dst || B < 1 || H < 1 || W < 1 || h < 1 || w < 1) { set_error("vbt_resize_frames: bad argument"); return VBT_ERR_ARG; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
    set_error("vbt_resize_frames: HIP device %d not available (%d visible) - no CPU fallback", device, ndev);
    return VBT_ERR_HIP;
  }
  VBT_HIP_CHECK(hipSetDevice(device));
  hipStream_t st = (hipStream_t)stream;
  size_t sb = (size_t)B * H * W * 3, db = (size_t)B * h * w * 3;
  uint8_t *ds = nullptr, *dd = nullptr;
  const uint8_t* sp = src;
  uint8_t* dp = dst;
  if (!src_on_device) {
    VBT_HIP_CHECK(hipMalloc((void**)&ds, sb));
    VBT_HIP_CHECK(hipMemcpyAsync(ds, src, sb, hipMemcpyHostToDevice, st));
    sp = ds;
  }
  if (!dst_on_device) {
    VBT_HIP_CHECK(hipMalloc((void**)&dd, db));
    dp = dd;
  }
  hipError_t e = resize_frames_dev(sp, B, H, W, dp, h, w, swap_rb, 0, st) == VBT_OK ? hipSuccess : hipErrorLaunchFailure;
  if (e == hipSuccess && !dst_on_device) e = hipMemcpyAsync(dst, dd, db, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess && (ds || dd)) e = hipStreamSynchronize(st);
  if (ds) (void)hipFree(ds);
  if (dd) (void)hipFree(dd);
  if (e != hipSuccess) { set_error("vbt_resize_frames failed: %s", hipGetErrorString(e)); return VBT_ERR_HIP; }
  return VBT_OK;
}

#ifdef VBT_POST_PROF
int vbt_post_prof_read(unsigned long long* out16, int reset) {
  if (reset) { unsigned long long z[16] = {0}; VBT_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(vbt::g_post_prof), z, sizeof(z))); return VBT_OK; }
  VBT_HIP_CHECK(hipDeviceSynchronize());
  VBT_HIP_CHECK(hipMemcpyFromSymbol(out16, HIP_SYMBOL(vbt::g_post_prof), 16 * sizeof(unsigned long long)));
  return VBT_OK;
}
#endif

int vbt_model_kernel_stats(const vbt_model* m, int B, vbt_kernel_stat* out, int cap, int* n) {
  if (!m || !out || !n || cap < F_COUNT) { set_error("bad argument"); return VBT_ERR_ARG; }
  for (int i = 0; i < F_COUNT; i++) {
    memset(&out[i], 0, sizeof(out[i]));
    snprintf(out[i].name, sizeof(out[i].name), "%s", kFamilyName[i]);
  }
  for (const Step& s : m->steps) {
    out[s.family].launches++;
    out[s.family].algorithmic_bytes += s.alg_bytes_per_frame * B + s.weight_bytes;
    out[s.family].macs += s.macs_per_frame * B;
  }
  *n = F_COUNT;
  return VBT_OK;
}

int vbt_model_profile(vbt_model* m, const uint8_t* frames_dev, int B, int reps, void* stream, double* ms_out, int cap) {
  if (!m || !frames_dev || !ms_out || cap < F_COUNT || reps < 1) { set_error("bad argument"); return VBT_ERR_ARG; }
  if (B < 1 || B > m->max_batch) { set_error("bad batch"); return VBT_ERR_CAPACITY; }
  hipStream_t st = (hipStream_t)stream;
  const int ns = (int)m->steps.size();
  std::vector<hipEvent_t> evs(ns + 1);
  for (auto& e : evs) VBT_HIP_CHECK(hipEventCreate(&e));
  for (int i = 0; i < F_COUNT; i++) ms_out[i] = 0.0;
  int rc = VBT_OK;
  for (int r = 0; r < reps && rc == VBT_OK; r++) {
    rc = enqueue_forward(m, frames_dev, B, st, m->out_boxes, m->out_scores, m->out_classes, m->out_counts, evs.data());
    if (rc) break;
    if (hipStreamSynchronize(st) != hipSuccess) { set_error("stream sync failed"); rc = VBT_ERR_HIP; break; }
    for (int i = 0; i < ns; i++) {
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, evs[i], evs[i + 1]);
      ms_out[m->steps[i].family] += ms;
    }
  }
  for (auto& e : evs) (void)hipEventDestroy(e);
  for (int i = 0; i < F_COUNT; i++) ms_out[i] /= reps;
  return rc;
}

// One bracket per kernel family: all launches of family i of the plan back to back on `stream` (`reps` passes between ONE pair
// of HIP events), so ms_out[i] / launches is an average launch duration without the ~3 us a pair of events around every short
// launch adds - the figure rocprofv3 --kernel-trace reports for the same kernels (profiles/) to within the dispatch gap.
int vbt_model_profile_families(vbt_model* m, int B, int reps, void* stream, double* ms_out, int cap) {
  if (!m || !ms_out || cap < F_COUNT || reps < 1) { set_error("bad argument"); return VBT_ERR_ARG; }
  if (B < 1 || B > m->max_batch) { set_error("bad batch"); return VBT_ERR_CAPACITY; }
  hipStream_t st = (hipStream_t)stream;
  hipEvent_t e0, e1;
  VBT_HIP_CHECK(hipEventCreate(&e0));
  VBT_HIP_CHECK(hipEventCreate(&e1));
  int rc = VBT_OK;
  for (int f = 0; f < F_COUNT && rc == VBT_OK; f++) {
    ms_out[f] = 0.0;
    bool any = false;
    for (const Step& s : m->steps) any |= s.family == f;
    if (!any) continue;
    for (int pass = 0; pass < 2 && rc == VBT_OK; pass++) {   // pass 0: warm (code objects, caches)
      const int n = pass == 0 ? 1 : reps;
      (void)hipEventRecord(e0, st);
      for (int r = 0; r < n && rc == VBT_OK; r++)
        for (const Step& s : m->steps)
          if (s.family == f && rc == VBT_OK) rc = launch_step(m, s, B, st, m->frames_stage, m->out_boxes, m->out_scores, m->out_classes, m->out_counts);
      (void)hipEventRecord(e1, st);
      if (hipStreamSynchronize(st) != hipSuccess) { set_error("stream sync failed"); rc = VBT_ERR_HIP; break; }
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, e0, e1);
      if (pass == 1) ms_out[f] = (double)ms / reps;
    }
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return rc;
}

// Per-launch timing of the plan (one forward in flight, HIP events around every launch): step i of the execution list ->
// family name, index of the last graph op it covers, kernel variant and milliseconds (average over `reps`).
int vbt_model_profile_steps(vbt_model* m, const uint8_t* frames_dev, int B, int reps, void* stream, vbt_step_time* out, int cap, int* n) {
  if (!m || !frames_dev || !out || !n || reps < 1) { set_error("bad argument"); return VBT_ERR_ARG; }
  if (B < 1 || B > m->max_batch) { set_error("bad batch"); return VBT_ERR_CAPACITY; }
  const int ns = (int)m->steps.size();
  if (cap < ns) { set_error("%d plan steps, buffer holds %d", ns, cap); return VBT_ERR_CAPACITY; }
  hipStream_t st = (hipStream_t)stream;
  std::vector<hipEvent_t> evs(ns + 1);
  for (auto& e : evs) VBT_HIP_CHECK(hipEventCreate(&e));
  for (int i = 0; i < ns; i++) {
    const Step& s = m->steps[i];
    memset(&out[i], 0, sizeof(out[i]));
    snprintf(out[i].family, sizeof(out[i].family), "%s", kFamilyName[s.family]);
    out[i].op = s.op;
    out[i].first_op = s.e_op >= 0 ? s.e_op : (s.sum_op >= 0 ? s.sum_op : (s.d_op >= 0 ? s.d_op : s.op));
    out[i].variant = s.variant;
    out[i].algorithmic_bytes = s.alg_bytes_per_frame * B + s.weight_bytes;
    out[i].macs = s.macs_per_frame * B;
  }
  int rc = VBT_OK;
  for (int r = 0; r < reps && rc == VBT_OK; r++) {
    rc = enqueue_forward(m, frames_dev, B, st, m->out_boxes, m->out_scores, m->out_classes, m->out_counts, evs.data());
    if (rc) break;
    if (hipStreamSynchronize(st) != hipSuccess) { set_error("stream sync failed"); rc = VBT_ERR_HIP; break; }
    for (int i = 0; i < ns; i++) {
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, evs[i], evs[i + 1]);
      out[i].ms += ms / reps;
    }
  }
  for (auto& e : evs) (void)hipEventDestroy(e);
  *n = ns;
  return rc;
}

// Measurement: every plan step launched `reps` times back to back on ONE stream, then `reps` times on each of `nstreams`
// streams at once.  conc_ms[i] (time per launch with the streams racing) against single_ms[i] says how much of step i a
// second and third forward in flight can hide: equal -> the kernel saturates a resource, 1/nstreams -> pure latency.
int vbt_model_profile_overlap(vbt_model* m, int B, int reps, int nstreams, float* single_ms, float* conc_ms, int cap, int* n) {
  if (!m || !single_ms || !conc_ms || !n || reps < 1 || nstreams < 1 || nstreams > 8) { set_error("bad argument"); return VBT_ERR_ARG; }
  if (B < 1 || B > m->max_batch) { set_error("bad batch"); return VBT_ERR_CAPACITY; }
  const int ns = (int)m->steps.size();
  if (cap < ns) { set_error("%d plan steps, buffer holds %d", ns, cap); return VBT_ERR_CAPACITY; }
  std::vector<hipStream_t> ss(nstreams, nullptr);
  for (auto& st : ss) VBT_HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  int rc = VBT_OK;
  for (int i = 0; i < ns && rc == VBT_OK; i++) {
    const Step& s = m->steps[i];
    for (int pass = 0; pass < 2 && rc == VBT_OK; pass++) {
      const int k = pass == 0 ? 1 : nstreams;
      for (int j = 0; j < k && rc == VBT_OK; j++) rc = launch_step(m, s, B, ss[j], m->frames_stage, m->out_boxes, m->out_scores, m->out_classes, m->out_counts);
      if (rc) break;
      VBT_HIP_CHECK(hipDeviceSynchronize());
      auto t0 = std::chrono::steady_clock::now();
      for (int r = 0; r < reps; r++)
        for (int j = 0; j < k; j++) (void)launch_step(m, s, B, ss[j], m->frames_stage, m->out_boxes, m->out_scores, m->out_classes, m->out_counts);
      VBT_HIP_CHECK(hipDeviceSynchronize());
      const float ms = (float)(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / (reps * k));
      (pass == 0 ? single_ms : conc_ms)[i] = ms;
    }
  }
  for (auto& st : ss) (void)hipStreamDestroy(st);
  *n = ns;
  return rc;
}

}  // extern "C"
