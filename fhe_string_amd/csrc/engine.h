// engine.h -- device engine (Level 0/1 of the C ABI): HIP context, ciphertext arena,
// key storage, LUT registry and the batched launches.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <thread>
#include <vector>

#include "common.h"

struct fhes_program;

namespace fhes {
// An execution lane: a HIP stream with the per-launch scratch its launches use.  The context's own stream is
// the MAIN lane (asynchronous, what a single caller sees); a program run that finds the main lane occupied by
// another caller's work takes a SIDE lane -- own stream, own keyswitch scratch, own slot map -- so that the
// reference's concurrent callers (rayon workers, /root/reference/src/ciphertext/search.rs:34-40, mod.rs:292
// `T: Sync`) overlap on the device instead of queueing behind one mutex.
struct Lane {
  hipStream_t stream = nullptr;
  uint64_t *ks_out = nullptr;   // [nodes][ks_stride]
  int64_t ks_out_nodes = 0;
  int8_t *ks_dig = nullptr;     // digits of a wide launch, A-fragment order: [ceil(nodes / 128) * 8][ksteps][64][16]
  uint64_t *ks_body = nullptr;  // [nodes]
  unsigned char *gen_scratch = nullptr;  // generic blind rotate: accumulators and digit spectra of a launch
  size_t gen_scratch_bytes = 0;
  int32_t *slot_map = nullptr;  // slot map of the program run this lane carries
  size_t slot_map_cap = 0;
  bool is_main = false;
  bool busy = false;            // side lanes: taken by a running call (guarded by fhes_ctx::mu)
  // buffers this lane has outgrown: hipFree waits for the whole DEVICE, i.e. for every other lane's running work, so a
  // side lane parks them here until the context goes away (the main lane, a single caller's, frees at once)
  std::vector<void *> retired;
  void drop(void *p) { if (!p) return; if (is_main) (void)hipFree(p); else retired.push_back(p); }
};
struct RunCounters {            // what one call adds to fhes_stats (merged under fhes_ctx::mu)
  uint64_t pbs = 0, launches = 0, linear = 0;
};
}  // namespace fhes

struct fhes_ctx {
  fhes_params p;
  fhes::Dims dims;
  int device = 0;
  int n_cus = 256;
  uint64_t uid = 0;          // unique per context (program images are keyed by it, not by address)
  std::mutex attr_mu;                              // kernel function -> dynamic-LDS size its attribute was set to (engine.hip: ensure_dyn_lds)
  std::map<const void *, size_t> attr_lds;
  std::vector<double2 *> lab_bsk_e;   // lab (FHES_BSK_E_CANDIDATES): copies of the engine-layout key at other addresses
  unsigned long long *lat_stamps = nullptr;   // lab: phase stamps of the stamped latency build (variant 26)
  std::atomic<uint64_t> host_bytes_up{0}, host_bytes_down{0};   // ciphertext bytes moved over PCIe by arena upload / download
  bool wpp_auto = true;     // classic 2_2 sets, N = 2048: the wave-per-polynomial build wherever its rounds of 1024 are cheaper than the throughput build's of 512 (FHES_WPP_AUTO=0: never)
  int mb_split_max = 0;     // multi-bit sets: launches of at most this many bootstraps take the split form (bundles first, then the sequential kernel)
  bool mb_lat512 = false;   // multi-bit sets: one-per-CU launches take the 512-thread latency build (blind_rotate_mb_lat_kernel)
  int lat_max = 0;   // launches of at most this many bootstraps take the latency build of the classic blind rotate (0: never)
  hipStream_t stream = nullptr;   // the main lane's stream (== main.stream)
  // Locking: `mu` guards the host-side state (arena free list, LUT registry, lanes, stats).  A program run
  // holds it while it sets up and while it cleans up, NOT while it queues its launches; `arena_rw` is held
  // shared while launches are queued (they read `arena`) and exclusively by an arena growth, which replaces
  // the allocation.  Order: mu before arena_rw; nobody waits for mu while holding arena_rw.
  std::mutex mu;
  std::shared_mutex arena_rw;
  std::condition_variable lane_cv;
  fhes::Lane main;                                 // wraps `stream` and the scratch pointers below
  std::vector<std::unique_ptr<fhes::Lane>> side;   // created on demand, at most kMaxSideLanes
  static constexpr int kMaxSideLanes = 15;         // main + 15 = the 16 hardware queues the library asks for (engine.hip, runtime_defaults)
  // Work queued on the main stream by calls that have RETURNED (the main lane is asynchronous) and may still be
  // running: the arena rows it writes (results) and reads or re-uses (scratch, handed back "stream-ordered" at return),
  // with an event behind it.  A side lane waits for exactly those entries that overlap its own inputs or the rows it was
  // just given -- not for a call that merely shares the context (that would put the lanes behind each other).
  // `lo/hi[1]` (scratch) is handed back to the free list only when the event has completed: a range freed "stream-ordered"
  // at return would be given to another lane while the kernels of this call still use it.
  struct MainPending { int32_t lo[2], hi[2]; hipEvent_t ev; bool free_second; };
  std::vector<MainPending> main_pending;
  std::vector<hipEvent_t> pending_ev_free;
  bool main_untracked = false;                     // a Level-0 call queued work on the main lane: its rows are unknown
  int active_runs = 0;                             // program runs between set-up and clean-up (guarded by mu)
  bool main_held = false;                          // a call is queueing launches on the main lane
  std::thread::id main_user{};                     // the thread whose work the main stream last received

  // ciphertext arena: u64 [capacity][ct_stride]
  uint64_t *arena = nullptr;
  int64_t capacity = 0;
  bool arena_fixed = false;  // FHES_ARENA_SLOTS given: no growth
  int ct_stride = 0;
  std::map<int32_t, int32_t> free_ranges;  // first -> count (coalesced)
  int64_t in_use = 0;

  // keys
  bool keys_loaded = false;
  void *bsk_e_base = nullptr;  // allocation behind bsk_e when it was made by fhes_upload_keys (a broadcast allocates bsk_e itself)
  int mb_ilv = 0;  // multi-bit key in the interleaved layout (FHES_MB_INTERLEAVE=1 at key upload; round-4 A/B)
  uint64_t *ksk = nullptr;   // [big][ks_level][ks_stride]
  uint64_t *ksk_bias = nullptr;  // [ks_stride]: (B/2) * sum_{i,l} ksk[i][l][j]  (biased-digit keyswitch)
  int ks_stride = 0;
  int8_t *ksk_pk = nullptr;  // keyswitch key as signed byte planes in MFMA B-fragment order (keyswitch_mfma_kernel)
  int ks_n_wb = 0;           // blocks of 16 key words: ceil((n + 1) / 16)
  uint64_t *bsk_std = nullptr;  // standard-domain bootstrapping key as uploaded (what `ServerKey: Serialize` hands out again; round 5)
  double2 *bsk_f = nullptr;  // canonical [n_ggsw][2][2][M]   (multi-bit: [n/g][2^g][2][2][M])
  double2 *bsk_e = nullptr;  // engine layout for blind-rotate variants 2/3 and multi-bit: [n_ggsw][2][2][8][M/8]
  double2 *twist = nullptr, *tw = nullptr;
  double2 *twist_inv = nullptr;  // twist * (2/N): untwist factors with the inverse-FFT scale folded in
  double2 *twist_inv64 = nullptr;  // the same * 2^-64 (variant 5: torus_from_f64_pair takes y * 2^-64)
  double2 *root = nullptr;       // exp(i*pi*t/N), t < 2N (multi-bit: Fourier image of a monomial)
  double2 *psi = nullptr;        // [M] forward twiddles: psi[2^l + b] = block b of the stage with 2^l blocks
  double2 *odd_fix = nullptr;    // [M] odd slots: factor applied to the Fourier key (see dev::fwd_butterfly_last)
  double2 *cdpp = nullptr;       // [8][M/8] constants of the forward lane-pair stage, thread-major

  // LUT registry
  uint64_t *lut_polys = nullptr;  // [lut_capacity][N]
  int lut_capacity = 0;
  std::vector<std::vector<uint64_t>> lut_tables;

  // per-launch scratch: in the lanes (`main`, `side`)
  unsigned char *desc_buf = nullptr;  // staging for fhes_pbs_batch descriptors
  size_t desc_cap = 0;

  // stats / timing
  fhes_stats stats{};
  bool timing = false;
  struct Timed { hipEvent_t e0, e1, e2; };
  std::vector<Timed> pending;         // launches whose events are not yet read
  std::vector<hipEvent_t> ev_free;    // recycled events
  int br_variant = 0;
  int ks_tb = 0;        // developer knob FHES_KS_TB: 16 = 16 ciphertexts per keyswitch block (default 8)
  int ks_mfma_min = 1;  // launches of at least this many ciphertexts take the matrix-core keyswitch (FHES_KS_MFMA_MIN; 0 = never)
  int ks_min_cps = 8;   // smallest mask range per keyswitch split (developer knob FHES_KS_MIN_CPS)

  explicit fhes_ctx(const fhes_params &pp) : p(pp), dims(pp) {}
};

namespace fhes {

struct DevNodeArrays {  // device pointers
  const int32_t *term_ptr, *term_idx;
  const int64_t *term_coef;
  const uint64_t *consts;
  const int32_t *lut_ids, *out_idx, *slot_map;
  int32_t n_slots = 0;  // stride of slot_map per instance
};

int hip_fail(hipError_t e, const char *what);
#define FHES_HIP(call)                                        \
  do {                                                        \
    hipError_t _e = (call);                                   \
    if (_e != hipSuccess) return fhes::hip_fail(_e, #call);   \
  } while (0)

// launches keyswitch + blind rotate for nodes [first, first+count) of n_inst instances on `lane`
// (the caller owns the lane; counters are merged into the context's stats by merge_counters_locked)
int launch_pbs(fhes_ctx *ctx, Lane &lane, const DevNodeArrays &a, int32_t first, int32_t count, int32_t n_inst, RunCounters *rc);
int launch_linear(fhes_ctx *ctx, Lane &lane, const DevNodeArrays &a, int32_t first, int32_t count, int32_t n_inst, RunCounters *rc);
void merge_counters_locked(fhes_ctx *ctx, const RunCounters &rc);
// Lane hand-out.  acquire_main_locked waits (on lane_cv, releasing mu) until no call is queueing on the main
// lane and takes it; acquire_lane_locked gives the main lane to a caller who finds it free and idle (or whose
// own work is the last it received) and a side lane to everybody else.
void acquire_main_locked(fhes_ctx *ctx, std::unique_lock<std::mutex> &lk);
int acquire_lane_locked(fhes_ctx *ctx, std::unique_lock<std::mutex> &lk, Lane **out);
void release_lane_locked(fhes_ctx *ctx, Lane *lane);
int ensure_slot_map(Lane &lane, size_t entries);
// main-lane bookkeeping (see fhes_ctx::main_pending); ranges are [lo, hi) pairs, an empty pair is lo == hi
int main_pending_push_locked(fhes_ctx *ctx, int32_t lo0, int32_t hi0, int32_t lo1, int32_t hi1, bool free_second = false);
void main_pending_prune_locked(fhes_ctx *ctx);
struct MainLaneHold {  // Level-0 entry points: hold mu for the whole call AND the main lane
  fhes_ctx *ctx;
  std::unique_lock<std::mutex> lk;
  explicit MainLaneHold(fhes_ctx *c) : ctx(c), lk(c->mu) { acquire_main_locked(c, lk); }
  ~MainLaneHold() { release_lane_locked(ctx, &ctx->main); }
};
int32_t arena_alloc_locked(fhes_ctx *ctx, int32_t count);
int arena_free_locked(fhes_ctx *ctx, int32_t first, int32_t count);
int32_t lut_register_locked(fhes_ctx *ctx, const uint64_t *table, int32_t n);
int collect_timing_locked(fhes_ctx *ctx);  // synchronises the stream
bool ctx_is_live(fhes_ctx *ctx);

}  // namespace fhes
