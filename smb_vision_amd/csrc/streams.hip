// Host-side fork/join resources of the library: per device, two private non-blocking streams and their events, created on first use and
// kept for the life of the process (no device memory involved; declared in include/smbv.h's conventions).
//   side  : attention backward runs its dQ kernel here beside dK/dV (attention_bwd.hip)
//   wside : the block backward runs its weight-gradient products here beside the input-gradient ones (block.hip, SideStream)
// Stream order is preserved for the caller: everything forked is joined back into the caller's stream before the forking function returns
// (attention) / before the block's gradients are declared final (block backward).
#include "block.h"
#include <mutex>

namespace smbv {
extern int g_variant;

ForkJoin* fork_join_for_current_device() {
  static ForkJoin table[64];
  static std::mutex mu;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lk(mu);
  ForkJoin& f = table[dev];
  if (!f.ready) {
    // A/B (variant bit 30): the side streams at the LOWEST priority the device offers, so that the caller's stream - the critical path - wins the block
    // dispatcher whenever both have work ready (default: equal priorities)
    int lo = 0, hi = 0;
    if ((g_variant & (1 << 30)) && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && lo != hi) {
      if (hipStreamCreateWithPriority(&f.side, hipStreamNonBlocking, lo) != hipSuccess) return nullptr;
      if (hipStreamCreateWithPriority(&f.wside, hipStreamNonBlocking, lo) != hipSuccess) return nullptr;
    } else {
      if (hipStreamCreateWithFlags(&f.side, hipStreamNonBlocking) != hipSuccess) return nullptr;
      if (hipStreamCreateWithFlags(&f.wside, hipStreamNonBlocking) != hipSuccess) return nullptr;
    }
    hipEvent_t* ev[8] = {&f.fork, &f.join, &f.wfork, &f.wjoin, &f.wdone[0], &f.wdone[1], &f.wdone[2], &f.wdone[3]};
    for (auto e : ev)
      if (hipEventCreateWithFlags(e, hipEventDisableTiming) != hipSuccess) return nullptr;
    f.mu = new std::mutex();
    f.ready = true;
  }
  return &f;
}

struct StatusWord { unsigned* host = nullptr; unsigned* dev = nullptr; bool tried = false; };
static StatusWord* status_for_current_device() {
  static StatusWord table[64];
  static std::mutex mu;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lk(mu);
  StatusWord& w = table[dev];
  if (!w.tried) {
    w.tried = true;
    void* h = nullptr;
    void* d = nullptr;
    if (hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess && hipHostGetDevicePointer(&d, h, 0) == hipSuccess) {
      w.host = (unsigned*)h;
      w.dev = (unsigned*)d;
      *(volatile unsigned*)w.host = 0u;
    } else {
      (void)hipGetLastError();
    }
  }
  return &w;
}
unsigned* device_status_word_dev() {
  StatusWord* w = status_for_current_device();
  return w ? w->dev : nullptr;
}
int device_status(bool clear) {
  StatusWord* w = status_for_current_device();
  if (!w || !w->host) return SMBV_OK;
  const unsigned code = *(volatile unsigned*)w->host;
  if (clear) *(volatile unsigned*)w->host = 0u;
  if (!code) return SMBV_OK;
  const char* what = code == 0x5001u ? "attention backward: a scheduling word of another launch epoch (two launches shared one slot, or the slot was left dirty)"
                     : code == 0x5002u ? "attention backward: a block found both tile queues exhausted (a tile was skipped: gradients of that launch are incomplete)"
                     : code == 0x5003u ? "attention backward: the completion counter carried another launch's epoch"
                     : code == 0x5004u ? "split launch: a consumer half gave up waiting for its producer half's partial sums (that tile's gradients were not stored)"
                     : code == 0x5005u ? "split weight-gradient launch: a block drew a ticket outside its launch's run (two launches shared one slot, or the slot was left dirty)"
                                       : "unknown fault";
  set_error("device fault 0x%x recorded by an earlier kernel - %s; results since then are not to be trusted (smbv_debug_pair_state(0) restores the scheduling state, smbv_device_status(1) clears the word)", code, what);
  return SMBV_ERR_STATE;
}

int stream_edge(ForkJoin* fj, hipEvent_t ev, hipStream_t from, hipStream_t to) {
  std::lock_guard<std::mutex> lk(*(std::mutex*)fj->mu);
  HIPRC(hipEventRecord(ev, from));
  HIPRC(hipStreamWaitEvent(to, ev, 0));
  return SMBV_OK;
}

int event_record_locked(ForkJoin* fj, hipEvent_t ev, hipStream_t on) {
  std::lock_guard<std::mutex> lk(*(std::mutex*)fj->mu);
  HIPRC(hipEventRecord(ev, on));
  return SMBV_OK;
}
int stream_wait_locked(ForkJoin* fj, hipStream_t st, hipEvent_t ev) {
  std::lock_guard<std::mutex> lk(*(std::mutex*)fj->mu);
  HIPRC(hipStreamWaitEvent(st, ev, 0));
  return SMBV_OK;
}

int SideStream::begin(hipStream_t caller) {
  if (!impl) {
    ForkJoin* fj = (g_variant & 16384) ? nullptr : fork_join_for_current_device();
    impl = fj;
    st = fj ? fj->wside : caller;
  }
  ForkJoin* fj = (ForkJoin*)impl;
  if (!fj) { st = caller; return SMBV_OK; }
  RC(stream_edge(fj, fj->wfork, caller, fj->wside));
  used = true;
  caller_ = caller;
  return SMBV_OK;
}

int SideStream::end(hipStream_t caller) {
  ForkJoin* fj = (ForkJoin*)impl;
  if (!fj || !used) return SMBV_OK;
  used = false;
  RC(stream_edge(fj, fj->wjoin, fj->wside, caller));
  return SMBV_OK;
}

}  // namespace smbv
