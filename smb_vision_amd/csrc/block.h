// One pre-LN transformer block (LayerNorm -> QKV -> [RoPE] -> attention -> projection + residual -> LayerNorm -> MLP + residual),
// forward and backward, sequenced over the kernels of this library.  Shared by the MIM step (VideoMAELayer, spec
// src/models/videomae/modeling_videomae.py:405-431) and the V-JEPA step (VJEPA2Layer, src/models/vjepa/modeling_vjepa.py:426-478).
#pragma once
#include "smbv_common.h"
#include "smbv_kernels.h"
#include <cmath>
#include <vector>

namespace smbv {

struct BlockPtrs {  // fp32 parameters (or their gradients) of one block
  float *ln1_w, *ln1_b, *q_w, *k_w, *v_w, *q_b, *k_b /* may be null: VideoMAE has no key bias */, *v_b, *proj_w, *proj_b, *ln2_w, *ln2_b, *fc1_w,
      *fc1_b, *fc2_w, *fc2_b;
};

struct BlockDims {
  size_t rows = 0;  // B * S tokens
  int w = 0, ff = 0, B = 0, S = 0, H = 0;
  int Dh = 64;      // true head dim (8..64, multiple of 8).  The attention kernels have native builds for 64- and 32-wide heads; any other
                    // width runs in 64-wide slots, zero-padded in the QKV / projection weights (exact: padded q/k columns add 0 to every
                    // score, padded v columns give 0 outputs)
  float eps = 1e-12f;
  int rope_grid = 0;              // > 0: 3-axis RoPE on q and k with this grid size (tokens per row; tokens per frame = grid^2)
  const int* pos = nullptr;       // token position ids for RoPE (null = index within the sequence)
  const float* rope_table = nullptr;
  int rope_rows = 0;              // positions per axis the table covers (frames, rows and columns of the token grid all stay below it)
  int attn_fp8 = 0;               // bit 0: attention forward on the block-scaled fp8 MFMA (attention_fp8.hip); bit 1: also the backward's recomputed
                                  // S / dP products (query-owner blocks, attention_bwd.hip attn_dq_fp8_block); the accumulating products stay bf16
  int o_resid = 0;                // 1: the attention forward also saves the bf16 residual of its fp32 output (LayerWs::attn_o_lo) and the backward's
                                  // delta = rowsum(dO * O) is formed from O + O_lo (not with attn_fp8, whose forward keeps no fp32 output tile layout in common)
  int slot() const { return Dh == 32 ? 32 : 64; }  // columns a head occupies in the QKV / attention-output matrices
  int wa() const { return H * slot(); }
  bool padded() const { return Dh != slot(); }
  float scale() const { return 1.0f / std::sqrt((float)Dh); }
};

struct LayerWs {
  bf16_t *wqkv, *wqkv_t, *proj, *proj_t, *fc1, *fc1_t, *fc2, *fc2_t;  // bf16 operand copies of the weights
  float* bqkv;
  bf16_t *ln1, *qkv, *attn_o, *ln2, *pre, *act;                       // saved activations
  bf16_t* attn_o_lo = nullptr;                                        // bf16(O_fp32 - attn_o) when BlockDims::o_resid, else null
  float *mean1, *rstd1, *mean2, *rstd2, *lse, *x_mid;
};

struct BwdScratch {  // sized for the widest block of the model (d_act, d_qkv and the blocks' LayerNorm partials live in the WgradPipe's rings)
  bf16_t *d_ln, *d_o, *gath0, *gath1;
  float *delta, *gpad, *gpad_b;  // gpad / gpad_b: head-padded weight / bias gradients before un-padding (null when no block is padded)
  void* scratch;        // LN-bwd partials / colsum partials / gemm_tn slabs of the products the STEP issues on the caller's stream (head, embeddings)
  size_t scratch_bytes;
  void* f8ws = nullptr; // quantised operands of the fp8 attention forward / backward (null unless a block asks for it)
  size_t f8ws_bytes = 0;
  void* wscratch;       // gemm_tn slabs + column-sum partials of the weight-gradient products that run on the side stream
  size_t wscratch_bytes;
  void* attn_split = nullptr;   // partial accumulators of the attention backward's split sweeps (attention_bwd.hip SweepPart)
  size_t attn_split_bytes = 0;
};

struct ForkJoin {  // streams.hip
  hipStream_t side = nullptr, wside = nullptr;
  hipEvent_t fork = nullptr, join = nullptr, wfork = nullptr, wjoin = nullptr;
  hipEvent_t wdone[4] = {nullptr, nullptr, nullptr, nullptr};  // WgradPipe: "deferred launch #k has finished" (slot k & 3)
  bool ready = false;
  void* mu = nullptr;  // std::mutex*: a record + wait pair on one of the events above is made under it (two host threads on one device)
};
ForkJoin* fork_join_for_current_device();
// `to` waits for everything enqueued on `from` so far: hipEventRecord(ev, from) + hipStreamWaitEvent(to, ev) as ONE step under the device's mutex,
// so that another host thread's re-record of the same event cannot fall between the two
int stream_edge(ForkJoin* fj, hipEvent_t ev, hipStream_t from, hipStream_t to);
// The two halves on their own, each under the same mutex (the wdone events of a WgradPipe are recorded at the end of a deferred launch and waited for two
// blocks later: a second host thread stepping another model on the device may re-record the slot in between - the wait then covers a LATER point of the
// in-order side stream, which is conservative - but the runtime calls on one hipEvent_t never overlap)
int event_record_locked(ForkJoin* fj, hipEvent_t ev, hipStream_t on);
int stream_wait_locked(ForkJoin* fj, hipStream_t st, hipEvent_t ev);

// The library's per-device weight-gradient side stream (streams.hip).  begin(): work enqueued on `st` afterwards
// starts only after everything enqueued on the caller's stream so far; end(): the caller's stream waits for all side work.  With the debug
// variant bit 16384 (or if the stream cannot be created) `st` IS the caller's stream and both calls do nothing.
struct SideStream {
  hipStream_t st = nullptr;
  void* impl = nullptr;
  bool used = false;
  hipStream_t caller_ = nullptr;
  int begin(hipStream_t caller);
  int end(hipStream_t caller);
  ~SideStream() { if (used) end(caller_); }  // an error return between begin() and end() still joins the side stream back into the caller's
};

// The device's sticky status word (streams.hip): 4 bytes of host-mapped memory per device, written by kernels that detect a fault they cannot return
// (attention_bwd.hip pair_fault) and read - without synchronising - by every entry point: non-zero = the library refuses further work on that device
// with SMBV_ERR_STATE until smbv_device_status(1) clears it.
unsigned* device_status_word_dev();            // device-side address for kernel arguments (null if the word could not be created)
int device_status(bool clear);                 // SMBV_OK, or SMBV_ERR_STATE with the fault code in the error text
int pair_state_debug(int op, hipStream_t st);  // attention_bwd.hip
#define SMBV_CHECK_DEVICE_STATE() do { int _ds = smbv::device_status(false); if (_ds) return _ds; } while (0)

struct Carver {
  char* base;
  size_t off = 0;
  template <typename T>
  T* take(size_t n) {
    off = (off + 255) & ~(size_t)255;
    T* p = base ? (T*)(base + off) : nullptr;
    off += n * sizeof(T);
    return p;
  }
};

extern int g_variant;  // debug hook, capi.hip (smbv_debug_set_variant)
extern int g_ab;       // second A/B word: environment SMBV_DEBUG_AB (capi.hip)
struct PrepList {
  std::vector<PrepJob> jobs;
  int tiles = 0;
  void add(const float* src, void* dst, int R, int C, int lds, int ldd, int kind, void* dst2 = nullptr, int ldd2 = 0) {
    PrepJob j;
    j.src = src; j.dst = dst; j.dst2 = dst2; j.R = R; j.C = C; j.ldd = ldd; j.kind = kind; j.tile0 = tiles; j.lds = lds; j.ldd2 = ldd2;
    jobs.push_back(j);
    tiles += prep_job_tiles(j);
  }
  void cast(const float* src, bf16_t* dst, int R, int C) { both(src, dst, C, nullptr, 0, R, C); }
  void tcast(const float* src, bf16_t* dst, int R, int C, int ldd) { both(src, nullptr, 0, dst, ldd, R, C); }
  // fp32 master [R, C] -> bf16 copy dst [R, ldd] and / or transposed bf16 copy dstT [C, lddT], the master read ONCE (kind 4) when the shapes allow the
  // vector accesses (every shipped width does); otherwise the scalar 32 x 32 jobs, one per copy
  void both(const float* src, bf16_t* dst, int ldd, bf16_t* dstT, int lddT, int R, int C) {
    const bool vec = C % 4 == 0 && R % 4 == 0 && !((uintptr_t)src & 15) && (!dst || (ldd % 4 == 0 && !((uintptr_t)dst & 7))) &&
                     (!dstT || (lddT % 4 == 0 && !((uintptr_t)dstT & 7)));
    if (vec && !(g_variant & 1)) { add(src, dst, R, C, C, ldd, 4, dstT, lddT); return; }   // (debug bit 1: the round-4 scalar jobs, A/B)
    if (dst) add(src, dst, R, C, C, ldd, 0);
    if (dstT) add(src, dstT, R, C, C, lddT, 1);
  }
};

void carve_layer(Carver& c, LayerWs& L, const BlockDims& d, const LayerWs* share_acts = nullptr);
// appends this block's weight-preparation jobs; for padded heads also enqueues the zero fills of the padded buffers on `st`
int prep_layer(PrepList& pl, const BlockPtrs& p, LayerWs& L, const BlockDims& d, hipStream_t st);
struct BwdNeeds {  // running maxima over the blocks of a model (elements, except scratch / wscratch: bytes)
  size_t act = 0, we = 0, qkv = 0, gpad = 0, gpad_b = 0, delta = 0, scratch = 0, wscratch = 0, f8 = 0, attn_split = 0;
};
void bwd_scratch_requirements(const BlockDims& d, BwdNeeds& n);
void carve_bwd_scratch(Carver& cv, BwdScratch& sc, const BwdNeeds& n);  // everything of BwdScratch except gath0 / gath1

// Deferred weight gradients (round 4).  A block's four weight-gradient products (+ bias sums, + the LayerNorm parameter reductions) do not feed the
// rest of the backward: with a pipe, layer_bwd only collects them and, once the block's input gradient is done, issues them as ONE grouped launch
// (gemm_tn_grouped: every tile over all tokens, no slabs) on the library's side stream, where it runs beside the NEXT block's backward.  What that
// launch reads must outlive the block: the bf16 residual gradients (in / mid) come from a ring of six buffers (block k: in = gb[2k], mid = gb[2k+1],
// out = gb[2k+2], indices mod 6), d_act / d_qkv / the LayerNorm partials from rings of two; block k starts only after launch k-2 has finished
// (event ring), which frees exactly the ring entries it overwrites.  Blocks the grouped launch cannot take (head-padded widths, widths that are not
// multiples of 128, a ranged block at B > 1) run their products inline on the same side stream, as before; they use the
// same rings.  Under recompute the blocks of a stack alternate between two activation sets and the step waits for launch k - 2 before it re-runs block
// k's forward (mim_step.hip): the launches - and the gradients, bit for bit - are the resident step's.
struct WgradPipe {
  ForkJoin* fj = nullptr;        // null until begin(); stays null when the side stream is unavailable or switched off (variant bit 16384): all inline
  bool allow_defer = true;       // false: every block inline
  bool last_block = false;       // the step sets it around the LAST block of its backward: its grouped launch has nothing to run beside and is split (gemm.hip)
  int k = 0;                     // blocks issued through this pipe
  int waited = -1;               // the caller's stream has waited for deferred launches <= waited
  bool deferred[4] = {false, false, false, false};  // launch (k & 3) was deferred and its wdone event recorded
  bf16_t* gb[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  bf16_t* d_act[2] = {nullptr, nullptr};
  bf16_t* d_qkv[2] = {nullptr, nullptr};
  void* ln1_part[2] = {nullptr, nullptr};
  void* ln2_part[2] = {nullptr, nullptr};
  size_t part_bytes = 0;
  bf16_t* cur() const { return gb[(2 * k) % 6]; }   // the bf16 gradient w.r.t. the next block's OUTPUT is written here by whoever produces it
  int begin();                                      // looks up the device's side stream / events
  int wait_for(int launch, hipStream_t st);         // st waits for deferred launch #launch (no-op if it was inline or already waited for)
  int drain(hipStream_t st);                        // st waits for everything deferred so far (end of the backward)
  // DP: a gradient group produced INLINE on `st` is final here - recorded through the side stream when earlier groups are still pending there, so that
  // the events complete in their array order (dp.py waits for a bucket's LAST event only)
  int mark_inline(hipEvent_t ev, hipStream_t st);
};
void carve_wgrad_pipe(Carver& cv, WgradPipe& pipe, size_t gb_elems, const struct BwdNeeds& n);

// Sq > 0: only tokens qoff..qoff+Sq-1 of every sequence are needed downstream (see mim_step.hip); 0 = all tokens.
int layer_fwd(const BlockPtrs& p, LayerWs& L, const BlockDims& d, BwdScratch& sc, const float* x_in, float* x_out, bf16_t* x_out_bf16, int Sq, int qoff,
              hipStream_t st);
// Backward of one block through a WgradPipe (above): the bf16 gradient w.r.t. the block's output comes in at pipe.cur(), leaves at pipe.cur() (after the call: k advanced); g: fp32 gradient in / out, g2: the
// mid-block partner.  dp_event (may be null): recorded when this block's parameter gradients are final - on the side stream for a deferred block.
int layer_bwd(const BlockPtrs& p, const BlockPtrs& gr, LayerWs& L, const BlockDims& d, BwdScratch& sc, const float* x_in, float* g, float* g2, WgradPipe& pipe,
              int Sq, int qoff, hipEvent_t dp_event, hipStream_t st);

#define RC(x) do { int _rc = (x); if (_rc) return _rc; } while (0)
#define HIPRC(x) do { hipError_t _e = (x); if (_e != hipSuccess) { set_error("%s: %s", #x, hipGetErrorString(_e)); return (int)_e; } } while (0)

}  // namespace smbv
