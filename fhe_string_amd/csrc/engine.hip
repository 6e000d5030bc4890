// engine.hip -- Level 0 of the C ABI: HIP context, arena, keys, LUTs, batched launches.
// Reference boundary: the field `ServerKey.k` (/root/reference/src/server_key.rs:13); every
// `k.k.<op>_parallelized` call of the reference becomes a set of nodes
//     out = PBS_lut( sum coef*ct + const )
// executed here in large batches (one keyswitch launch + one blind-rotate launch).
#include "engine.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <set>

#include "kernels.hip.h"
#include "kernel_br_lat.hip.h"
#include "kernel_br_wpp.hip.h"

namespace fhes {

// HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); streams that share one run their kernels
// one after the other, so with the default at most ~2-4 of a context's lanes make progress at a time (measured:
// 4 concurrent finds 463 ms with 4 queues, 265 ms with 16; one find 221 ms).  The ROCm runtime reads the variable
// when it initialises, i.e. at the process's first HIP call: this constructor runs when the library is loaded
// and only supplies a default -- a value the host application has set wins, and an application that has touched
// HIP before it loads this library has to set it itself.
__attribute__((constructor)) static void runtime_defaults() { setenv("GPU_MAX_HW_QUEUES", "16", 0); }

// registry of live contexts: ciphertext objects may outlive their server key
static std::mutex g_live_mu;
static std::set<fhes_ctx *> g_live;
bool ctx_is_live(fhes_ctx *ctx) {
  std::lock_guard<std::mutex> lk(g_live_mu);
  return g_live.count(ctx) != 0;
}

int hip_fail(hipError_t e, const char *what) {
  return fail(FHES_ERR_NO_DEVICE, "HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
}

void host_fft_tables(int N, std::vector<double> &twist, std::vector<double> &tw) {
  const int M = N / 2;
  twist.resize((size_t)2 * M);
  tw.resize((size_t)M);  // M/2 complex
  const long double pi = 3.14159265358979323846264338327950288L;
  for (int j = 0; j < M; j++) {
    long double a = pi * (long double)j / (long double)N;
    twist[2 * j] = (double)cosl(a);
    twist[2 * j + 1] = (double)sinl(a);
  }
  // exactly symmetric under quarter turns: tw[t + M/4] = -i * tw[t] = (tw[t].im, -tw[t].re) bit for bit;
  // the kernels derive those entries from their partners instead of loading them.  Exact values at the trivial angles.
  for (int t = 0; t < M / 2; t++) {
    if (M >= 4 && t >= M / 4) {
      tw[2 * t] = tw[2 * (t - M / 4) + 1];
      tw[2 * t + 1] = -tw[2 * (t - M / 4)];
      continue;
    }
    long double a = -2.0L * pi * (long double)t / (long double)M;
    tw[2 * t] = (double)cosl(a);
    tw[2 * t + 1] = (double)sinl(a);
    if (t == 0) { tw[0] = 1.0; tw[1] = 0.0; }
  }
  twist[0] = 1.0; twist[1] = 0.0;
}

// Tables of the forward transform (same values as the oracle: cosl / sinl in long double, exact entries at
// the trivial angles).  root[t] = exp(i pi t / N), t < 2N; slot m of a spectrum is the evaluation at
// root[(1 - 4 bitrev(m)) mod 2N]; psi[2^l + b] = twiddle of block b of the stage with 2^l blocks
// = (root of the block's first slot)^half; odd_fix[odd m] = square of the last-stage twiddle of m's pair;
// cdpp[e * M/8 + q] = constant of the lane-pair stage for the point the thread q holds in register e.
void host_fwd_tables(int N, std::vector<double> &root, std::vector<double> &psi, std::vector<double> &odd_fix,
                     std::vector<double> &cdpp) {
  const int M = N / 2, TP = M / 8;
  int logm = 0;
  while ((1 << logm) < M) logm++;
  root.assign((size_t)4 * N, 0.0);
  const long double pi = 3.14159265358979323846264338327950288L;
  // root[t + N/2] = i * root[t] = (-root[t].im, root[t].re), bit for bit: only the first quadrant comes from cosl / sinl
  for (int t = 0; t < 2 * N; t++) {
    if (t >= N / 2) {
      root[2 * t] = -root[2 * (t - N / 2) + 1];
      root[2 * t + 1] = root[2 * (t - N / 2)];
      continue;
    }
    const long double a = pi * (long double)t / (long double)N;
    root[2 * t] = (double)cosl(a);
    root[2 * t + 1] = (double)sinl(a);
    if (t == 0) { root[0] = 1.0; root[1] = 0.0; }
  }
  auto slot_exp = [&](int m) {
    int f = 0;
    for (int b = 0; b < logm; b++) f |= ((m >> b) & 1) << (logm - 1 - b);
    return (long)((((1 - 4 * f) % (2 * N)) + 2 * N) % (2 * N));
  };
  psi.assign((size_t)2 * M, 0.0);
  odd_fix.assign((size_t)2 * M, 0.0);
  for (int l = 0; l < logm; l++) {
    const int nb = 1 << l, half = M >> (l + 1);
    for (int b = 0; b < nb; b++) {
      const long e = (slot_exp(b * 2 * half) * half) % (2 * N);
      psi[2 * (nb + b)] = root[2 * e];
      psi[2 * (nb + b) + 1] = root[2 * e + 1];
      if (half == 1) {
        const long e2 = (2 * e) % (2 * N);
        odd_fix[2 * (2 * b)] = 1.0;
        odd_fix[2 * (2 * b + 1)] = root[2 * e2];
        odd_fix[2 * (2 * b + 1) + 1] = root[2 * e2 + 1];
      }
    }
  }
  cdpp.assign((size_t)2 * M, 0.0);
  const int LO = logm - 3 * ((logm - 1) / 3);  // the last 3-bit pass owns index bits [LO, LO + 3); LO = 1
  for (int e = 0; e < 8; e++)
    for (int q = 0; q < TP; q++) {
      const int p = ((q >> LO) << (LO + 3)) | (e << LO) | (q & ((1 << LO) - 1));  // dev::fft_point<LOGM, LO, 3>
      const double tre = psi[2 * (M / 2 + p / 2)], tim = psi[2 * (M / 2 + p / 2) + 1];
      cdpp[2 * ((size_t)e * TP + q)] = (p & 1) ? -tre : tre;
      cdpp[2 * ((size_t)e * TP + q) + 1] = tim;
    }
}

// the shapes the specialised throughput kernels (variant 5, multi-bit) are written for
static bool fast_shape(const fhes_params &p) {
  return p.glwe_dim == 1 && p.pbs_level == 1 && (p.pbs_base_log == 23 || p.pbs_base_log == 18) && (p.poly_size == 2048 || p.poly_size == 256);
}

// What the device serves (round 3: every classic shape of `generate_keys_with_params`, lib.rs:50-62, within the
// limits below; anything else is refused HERE, loudly, never run on a wrong kernel):
//   * k = 1, L = 1, N = 2048 (256 for tests), base 2^23 / 2^18: the throughput kernels, classic and multi-bit;
//   * classic sets with k <= 3, L <= 3, N = 256 .. 8192: the generic kernel (blind_rotate_gen_kernel) -- one
//     workgroup of (k + 1) * N / 16 threads (<= 1024) with (k + 1) * N * 8 bytes of LDS (<= 128 KB).
static bool kernels_support(const fhes_params &p, std::string *why) {
  if (p.poly_size < 256 || p.poly_size > 8192) { *why = "HIP kernels are instantiated for poly_size 256 .. 8192"; return false; }
  if (p.ks_level > dev::KS_MAXLV || p.ks_base_log > 7) { *why = "keyswitch decomposition too large"; return false; }
  // keyswitch_kernel keeps the rounded prefix of a mask coefficient in 32 bits
  if (p.ks_level * p.ks_base_log > 31) { *why = "keyswitch decomposition wider than 31 bits (ks_level * ks_base_log)"; return false; }
  if (p.lwe_dim + 1 > 1024) { *why = "lwe_dim too large (at most 1023)"; return false; }
  if (p.grouping_factor > 1) {
    if (p.grouping_factor > 3) { *why = "HIP kernels support grouping factors 2 and 3"; return false; }
    if (!fast_shape(p)) { *why = "the multi-bit kernels are written for glwe_dim 1, pbs_level 1, pbs_base_log 23 / 18, poly_size 2048 (256)"; return false; }
    return true;
  }
  if (fast_shape(p)) return true;
  if (p.glwe_dim > 3 || p.pbs_level > 3) { *why = "the generic blind rotate serves glwe_dim <= 3 and pbs_level <= 3"; return false; }
  if ((p.glwe_dim + 1) * (p.poly_size / 16) > 1024 || (p.poly_size >= 4096 && p.glwe_dim > 1)) {
    *why = "the generic blind rotate needs (glwe_dim + 1) * poly_size / 16 <= 1024 threads, and glwe_dim 1 from poly_size 4096";
    return false;
  }
  if ((size_t)(p.glwe_dim + 1) * p.poly_size * 8 + (size_t)(p.lwe_dim + 4) * 4 > 160 * 1024) { *why = "the generic blind rotate needs (glwe_dim + 1) * poly_size * 8 bytes of LDS"; return false; }
  if (p.pbs_base_log * p.pbs_level > 62) { *why = "PBS decomposition wider than 62 bits"; return false; }
  return true;
}

// ---- arena (first-fit free list of handle ranges) -----------------------------------------
// Grows the arena to at least `min_capacity` slots (handles are slot numbers, so they stay valid):
// new allocation, device-to-device copy, old one released.  Called with the stream idle.
static int arena_grow_locked(fhes_ctx *ctx, int64_t min_capacity) {
  if (min_capacity <= ctx->capacity) return FHES_OK;
  if (ctx->arena_fixed) return fail(FHES_ERR_OOM, "ciphertext arena exhausted (%lld slots, fixed by FHES_ARENA_SLOTS)", (long long)ctx->capacity);
  int64_t cap = std::max<int64_t>(min_capacity + min_capacity / 8, ctx->capacity * 2);
  cap = std::min<int64_t>(cap, (int64_t)INT32_MAX - 1);
  const size_t row = (size_t)ctx->ct_stride * 8;
  size_t free_b = 0, total_b = 0;
  FHES_HIP(hipSetDevice(ctx->device));
  // no launch may be queued while the allocation is replaced (runs hold arena_rw shared while they queue), and
  // everything queued on any lane must have finished with the old one
  std::unique_lock<std::shared_mutex> grow_lk(ctx->arena_rw);
  FHES_HIP(hipDeviceSynchronize());
  FHES_HIP(hipMemGetInfo(&free_b, &total_b));
  const int64_t fit = (int64_t)((free_b - std::min<size_t>(free_b, (size_t)1 << 30)) / row);  // keep 1 GiB for keys / staging
  if (cap > fit) cap = fit;
  if (cap < min_capacity)
    return fail(FHES_ERR_OOM, "ciphertext arena cannot grow to %lld slots (%.1f GB): %.1f GB of device memory free",
                (long long)min_capacity, (double)min_capacity * row / 1e9, (double)free_b / 1e9);
  uint64_t *bigger = nullptr;
  FHES_HIP(hipMalloc(&bigger, (size_t)cap * row));
  hipError_t e = hipMemcpy(bigger, ctx->arena, (size_t)ctx->capacity * row, hipMemcpyDeviceToDevice);
  if (e != hipSuccess) { (void)hipFree(bigger); return hip_fail(e, "arena copy"); }
  FHES_HIP(hipFree(ctx->arena));
  ctx->arena = bigger;
  const int32_t old_cap = (int32_t)ctx->capacity;
  ctx->capacity = cap;
  // append [old_cap, cap) to the free list (merging with a free range that ends at old_cap)
  int32_t first = old_cap, count = (int32_t)(cap - old_cap);
  if (!ctx->free_ranges.empty()) {
    auto last = std::prev(ctx->free_ranges.end());
    if (last->first + last->second == old_cap) { first = last->first; count += last->second; ctx->free_ranges.erase(last); }
  }
  ctx->free_ranges[first] = count;
  return FHES_OK;
}

static int32_t arena_take(fhes_ctx *ctx, int32_t count) {
  for (auto it = ctx->free_ranges.begin(); it != ctx->free_ranges.end(); ++it) {
    if (it->second >= count) {
      const int32_t first = it->first, left = it->second - count;
      ctx->free_ranges.erase(it);
      if (left > 0) ctx->free_ranges[first + count] = left;
      ctx->in_use += count;
      return first;
    }
  }
  return -1;
}

// The arena starts at 2^16 slots (1.1 GB) and grows on demand (a program run asks for
// n_inst * fhes_program_scratch_cells at once): sized by what runs, not by an environment variable.
// FHES_ARENA_SLOTS, when set, pins the size instead (no growth).
int32_t arena_alloc_locked(fhes_ctx *ctx, int32_t count) {
  if (count <= 0) return fail(FHES_ERR_INVALID, "arena_alloc: count must be positive");
  int32_t h = arena_take(ctx, count);
  if (h >= 0) return h;
  if (!ctx->main_pending.empty()) {  // rows that returned main-lane calls still hold: reclaim what has finished ...
    main_pending_prune_locked(ctx);
    h = arena_take(ctx, count);
    if (h >= 0) return h;
    if (!ctx->main_pending.empty()) {  // ... and wait for the rest before the arena is grown for them
      FHES_HIP(hipStreamSynchronize(ctx->stream));
      main_pending_prune_locked(ctx);
      h = arena_take(ctx, count);
      if (h >= 0) return h;
    }
  }
  // the request must fit behind the last used slot at worst
  int64_t tail_free = 0;
  if (!ctx->free_ranges.empty()) {
    auto last = std::prev(ctx->free_ranges.end());
    if (last->first + last->second == ctx->capacity) tail_free = last->second;
  }
  const int rc = arena_grow_locked(ctx, ctx->capacity + (int64_t)count - tail_free);
  if (rc) return rc;
  h = arena_take(ctx, count);
  if (h < 0) return fail(FHES_ERR_OOM, "ciphertext arena exhausted (%lld slots)", (long long)ctx->capacity);
  return h;
}

int arena_free_locked(fhes_ctx *ctx, int32_t first, int32_t count) {
  if (first < 0 || count <= 0 || (int64_t)first + count > ctx->capacity)
    return fail(FHES_ERR_INVALID, "arena_free: bad range");
  auto next = ctx->free_ranges.lower_bound(first);
  if (next != ctx->free_ranges.end() && next->first < first + count)
    return fail(FHES_ERR_INVALID, "arena_free: range overlaps a free range");
  int32_t new_first = first, new_count = count;
  if (next != ctx->free_ranges.begin()) {
    auto prev = std::prev(next);
    if (prev->first + prev->second > first) return fail(FHES_ERR_INVALID, "arena_free: double free");
    if (prev->first + prev->second == first) {
      new_first = prev->first;
      new_count += prev->second;
      ctx->free_ranges.erase(prev);
    }
  }
  if (next != ctx->free_ranges.end() && next->first == first + count) {
    new_count += next->second;
    ctx->free_ranges.erase(next);
  }
  ctx->free_ranges[new_first] = new_count;
  ctx->in_use -= count;
  return FHES_OK;
}

// ---- LUT registry ------------------------------------------------------------------------
int32_t lut_register_locked(fhes_ctx *ctx, const uint64_t *table, int32_t n) {
  const Dims &d = ctx->dims;
  if (!table || n != d.space()) return fail(FHES_ERR_INVALID, "lut_register: need %d entries", d.space());
  std::vector<uint64_t> t(table, table + n);
  for (auto &v : t)
    if (v >= (uint64_t)d.space()) return fail(FHES_ERR_INVALID, "lut_register: value out of range");
  for (size_t i = 0; i < ctx->lut_tables.size(); i++)
    if (ctx->lut_tables[i] == t) return (int32_t)i;
  if ((int)ctx->lut_tables.size() >= ctx->lut_capacity) return fail(FHES_ERR_OOM, "LUT registry full");
  // accumulator body polynomial: box size N/space, shifted by half a box
  const int N = d.N, box = N / d.space();
  std::vector<uint64_t> poly(N);
  for (int j = 0; j < N; j++) {
    const int jj = j + box / 2;
    poly[j] = (jj < N) ? t[jj / box] * d.delta : (uint64_t)0 - t[0] * d.delta;
  }
  const int32_t id = (int32_t)ctx->lut_tables.size();
  hipError_t e = hipMemcpy(ctx->lut_polys + (size_t)id * N, poly.data(), sizeof(uint64_t) * N, hipMemcpyHostToDevice);
  if (e != hipSuccess) return hip_fail(e, "lut upload");
  ctx->lut_tables.push_back(std::move(t));
  return id;
}

// ---- launches ------------------------------------------------------------------------------
static dev::NodeDesc make_desc(const DevNodeArrays &a, int32_t first, int32_t count, int32_t v0, int32_t v_count) {
  dev::NodeDesc d;
  d.term_ptr = a.term_ptr; d.term_idx = a.term_idx; d.term_coef = a.term_coef;
  d.consts = a.consts; d.lut_ids = a.lut_ids; d.out_idx = a.out_idx; d.slot_map = a.slot_map;
  d.first = first; d.count = count; d.n_slots = a.n_slots; d.v0 = v0; d.v_count = v_count;
  return d;
}

// keyswitch scratch of a lane (its owner calls this; a lane's earlier launches have been queued on its own
// stream, and hipFree waits for the device)
static int ensure_ks_out(fhes_ctx *ctx, Lane &lane, int64_t nodes) {
  if (nodes <= lane.ks_out_nodes) return FHES_OK;
  lane.drop(lane.ks_out);
  lane.drop(lane.ks_dig);
  lane.drop(lane.ks_body);
  lane.ks_out = nullptr; lane.ks_dig = nullptr; lane.ks_body = nullptr;
  lane.ks_out_nodes = 0;
  // side lanes carry single operations or a handful of combined ones (levels of a few to a few thousand bootstraps);
  // growing means parking the old buffers (Lane::drop), so start where a combined config-2 find already fits
  const int64_t floor_nodes = lane.is_main ? 1024 : 8192;
  const int64_t cap = ((std::max<int64_t>(nodes, floor_nodes) + 127) / 128) * 128;  // whole 128-row tiles of the MFMA keyswitch
  FHES_HIP(hipMalloc(&lane.ks_out, (size_t)cap * ctx->ks_stride * sizeof(uint64_t)));
  FHES_HIP(hipMalloc(&lane.ks_dig, (size_t)cap * ctx->dims.big * ctx->dims.ks_level));
  FHES_HIP(hipMalloc(&lane.ks_body, (size_t)cap * sizeof(uint64_t)));
  lane.ks_out_nodes = cap;
  return FHES_OK;
}

int ensure_slot_map(Lane &lane, size_t entries) {
  if (entries <= lane.slot_map_cap) return FHES_OK;
  FHES_HIP(hipStreamSynchronize(lane.stream));
  lane.drop(lane.slot_map);
  lane.slot_map = nullptr;
  lane.slot_map_cap = 0;
  const size_t cap = lane.is_main ? entries : std::max<size_t>(entries * 2, (size_t)1 << 16);
  FHES_HIP(hipMalloc(&lane.slot_map, 4 * cap));
  lane.slot_map_cap = cap;
  return FHES_OK;
}

// ---- lanes ---------------------------------------------------------------------------------
void acquire_main_locked(fhes_ctx *ctx, std::unique_lock<std::mutex> &lk) {
  ctx->lane_cv.wait(lk, [&] { return !ctx->main_held; });
  ctx->main_held = true;
  ctx->main_user = std::this_thread::get_id();
}

int acquire_lane_locked(fhes_ctx *ctx, std::unique_lock<std::mutex> &lk, Lane **out) {
  const auto me = std::this_thread::get_id();
  for (;;) {
    // the main lane: nobody is queueing on it, and either this thread's own work is the last it received (a
    // single caller keeps its asynchronous pipeline) or the stream has drained
    if (!ctx->main_held && (ctx->main_user == me || ctx->main_user == std::thread::id() ||
                            hipStreamQuery(ctx->stream) == hipSuccess)) {
      (void)hipGetLastError();  // hipStreamQuery reports "not ready" as an error value
      ctx->main_held = true;
      ctx->main_user = me;
      *out = &ctx->main;
      return FHES_OK;
    }
    (void)hipGetLastError();
    for (auto &l : ctx->side)
      if (!l->busy) { l->busy = true; *out = l.get(); return FHES_OK; }
    if ((int)ctx->side.size() < fhes_ctx::kMaxSideLanes) {
      auto l = std::make_unique<Lane>();
      FHES_HIP(hipStreamCreateWithFlags(&l->stream, hipStreamNonBlocking));
      l->busy = true;
      *out = l.get();
      ctx->side.push_back(std::move(l));
      return FHES_OK;
    }
    ctx->lane_cv.wait(lk);  // every lane is taken: wait for one to come back
  }
}

void release_lane_locked(fhes_ctx *ctx, Lane *lane) {
  if (!lane) return;
  if (lane == &ctx->main) {
    if (ctx->main_held && ctx->main_untracked) {  // a Level-0 call: which rows it touches is not tracked -> all of them
      (void)main_pending_push_locked(ctx, 0, INT32_MAX, 0, 0);
      ctx->main_untracked = false;
    }
    ctx->main_held = false;
  } else {
    lane->busy = false;
  }
  ctx->lane_cv.notify_all();
}

void main_pending_prune_locked(fhes_ctx *ctx) {
  size_t w = 0;
  for (size_t i = 0; i < ctx->main_pending.size(); i++) {
    if (hipEventQuery(ctx->main_pending[i].ev) == hipSuccess) {
      const auto &mp = ctx->main_pending[i];
      if (mp.free_second && mp.hi[1] > mp.lo[1]) (void)arena_free_locked(ctx, mp.lo[1], mp.hi[1] - mp.lo[1]);  // the deferred scratch
      ctx->pending_ev_free.push_back(mp.ev);
    } else {
      ctx->main_pending[w++] = ctx->main_pending[i];
    }
  }
  (void)hipGetLastError();  // "not ready" is reported as an error value
  ctx->main_pending.resize(w);
}

int main_pending_push_locked(fhes_ctx *ctx, int32_t lo0, int32_t hi0, int32_t lo1, int32_t hi1, bool free_second) {
  main_pending_prune_locked(ctx);
  hipEvent_t ev;
  if (!ctx->pending_ev_free.empty()) { ev = ctx->pending_ev_free.back(); ctx->pending_ev_free.pop_back(); }
  else FHES_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  FHES_HIP(hipEventRecord(ev, ctx->stream));
  ctx->main_pending.push_back({{lo0, lo1}, {hi0, hi1}, ev, free_second});
  return FHES_OK;
}

void merge_counters_locked(fhes_ctx *ctx, const RunCounters &rc) {
  ctx->stats.pbs_count += rc.pbs;
  ctx->stats.pbs_launches += rc.launches;
  ctx->stats.linear_count += rc.linear;
}

// wide launches: digits -> i8 matrix product on the matrix cores (kernels.hip.h K1b); same integers, same bits
static int launch_keyswitch_mfma(fhes_ctx *ctx, Lane &lane, const dev::NodeDesc &d) {
  const Dims &dm = ctx->dims;
  const int ksteps = dm.big * dm.ks_level / 64;
  hipLaunchKernelGGL(dev::keyswitch_digits_kernel, dim3(d.v_count), dim3(128), 0, lane.stream, d, ctx->arena,
                     ctx->ct_stride, dm.big, dm.ks_level, dm.ks_base_log, ksteps, lane.ks_dig, lane.ks_body);
  FHES_HIP(hipGetLastError());
  const int row_blocks = (d.v_count + 127) / 128, col_blocks = (ctx->ks_n_wb + 1) / 2;
  // few row blocks: split the digit range so that the launch still covers the chip (u64 atomics: exact)
  int splits = 1;
  while (splits < 16 && row_blocks * col_blocks * splits < 512 && ksteps % (splits * 2) == 0) splits *= 2;
  if (splits > 1) FHES_HIP(hipMemsetAsync(lane.ks_out, 0, (size_t)d.v_count * ctx->ks_stride * sizeof(uint64_t), lane.stream));
  hipLaunchKernelGGL(dev::keyswitch_mfma_kernel, dim3(row_blocks, col_blocks, splits), dim3(256), 0, lane.stream,
                     lane.ks_dig, ctx->ksk_pk, lane.ks_body, lane.ks_out, ctx->ks_stride, ksteps, ctx->ks_n_wb, dm.n,
                     d.v_count, ksteps / splits, splits > 1 ? 1 : 0);
  FHES_HIP(hipGetLastError());
  return FHES_OK;
}

static int launch_keyswitch(fhes_ctx *ctx, Lane &lane, const dev::NodeDesc &d) {
  const Dims &dm = ctx->dims;
  if (ctx->ks_mfma_min > 0 && d.v_count >= ctx->ks_mfma_min && ctx->ksk_pk) return launch_keyswitch_mfma(ctx, lane, d);
  FHES_HIP(hipMemsetAsync(lane.ks_out, 0, (size_t)d.v_count * ctx->ks_stride * sizeof(uint64_t), lane.stream));
  // 8 ciphertexts per block; FHES_KS_TB=16 forces 16 (measured no faster: kernels.hip.h)
  const bool wide = ctx->ks_tb == dev::KS_TB_WIDE;
  const int tb = wide ? dev::KS_TB_WIDE : dev::KS_TB;
  const int tiles = (d.v_count + tb - 1) / tb;
  // wide launches: at most one split per KS_IC-coefficient chunk; small launches (the latency
  // path, a handful of tiles) split down to FHES_KS_MIN_CPS coefficients so that the 61 MB key is
  // streamed by all CUs
  const int min_cps = ctx->ks_min_cps;
  const int max_splits = dm.big / (tiles <= 8 ? min_cps : dev::KS_IC);
  int splits = 1;
  while (splits < max_splits && tiles * splits < 1024) splits *= 2;
  const int cps = dm.big / splits;
  dim3 grid(tiles, splits), block(256);
  const int jpt = ctx->ks_stride / 256;
#define KS_LAUNCH(J)                                                                                        \
  if (wide)                                                                                                 \
    hipLaunchKernelGGL((dev::keyswitch_kernel<J, dev::KS_TB_WIDE>), grid, block, 0, lane.stream, d, ctx->arena, \
                       ctx->ct_stride, dm.big, ctx->ksk, ctx->ksk_bias, dm.ks_level, dm.ks_base_log,       \
                       ctx->ks_stride, dm.n, cps, lane.ks_out);                                             \
  else                                                                                                      \
    hipLaunchKernelGGL((dev::keyswitch_kernel<J, dev::KS_TB>), grid, block, 0, lane.stream, d, ctx->arena, \
                       ctx->ct_stride, dm.big, ctx->ksk, ctx->ksk_bias, dm.ks_level, dm.ks_base_log,       \
                       ctx->ks_stride, dm.n, cps, lane.ks_out)
  switch (jpt) {
    case 1: KS_LAUNCH(1); break;
    case 2: KS_LAUNCH(2); break;
    case 3: KS_LAUNCH(3); break;
    case 4: KS_LAUNCH(4); break;
    default: return fail(FHES_ERR_UNSUPPORTED, "ks_stride %d", ctx->ks_stride);
  }
#undef KS_LAUNCH
  FHES_HIP(hipGetLastError());
  return FHES_OK;
}


// hipFuncAttributeMaxDynamicSharedMemorySize is per kernel FUNCTION (and device): remembered per context by the function's
// address and the size asked for, so that two instantiations never share a flag (the interleaved and plain
// multi-bit builds did) and a larger request for the same function is set again.
static int ensure_dyn_lds(fhes_ctx *ctx, const void *fn, size_t lds) {
  std::lock_guard<std::mutex> g(ctx->attr_mu);
  auto it = ctx->attr_lds.find(fn);
  if (it != ctx->attr_lds.end() && it->second >= lds) return FHES_OK;
  FHES_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  ctx->attr_lds[fn] = lds;
  return FHES_OK;
}

#ifdef FHES_LAB
template <int LOGN>
static int launch_br_v1(fhes_ctx *ctx, Lane &lane, const dev::NodeDesc &d) {
  constexpr int N = 1 << LOGN, M = N / 2;
  const size_t lds = (size_t)2 * N * 8 + (size_t)2 * M * 16 + (size_t)((ctx->dims.n + 1 + 3) & ~3) * 4;
  auto kern = dev::blind_rotate_v1_kernel<LOGN, 23>;
  if (int rc = ensure_dyn_lds(ctx, reinterpret_cast<const void *>(kern), lds)) return rc;
  hipLaunchKernelGGL(kern, dim3(d.v_count), dim3(M / 4), lds, lane.stream, d, ctx->arena,
                     ctx->ct_stride, lane.ks_out, ctx->ks_stride, ctx->dims.n, ctx->bsk_f,
                     ctx->lut_polys, ctx->twist, ctx->tw, ctx->psi);
  FHES_HIP(hipGetLastError());
  return FHES_OK;
}
#endif  // FHES_LAB

template <int LOGN, int G, bool PIPE, bool LAT = false, int BL = 23, bool ILV = false, bool SCAL = false>
static int launch_br_mb(fhes_ctx *ctx, Lane &lane, const dev::NodeDesc &d) {
  if (BL == 23 && ctx->dims.pbs_base_log == 18) return launch_br_mb<LOGN, G, PIPE, LAT, 18, ILV, SCAL>(ctx, lane, d);
  // launches of at most one workgroup per CU take the uniform monomial factors through the scalar cache (kernels.hip.h, SCAL)
  if (!SCAL && d.v_count <= ctx->n_cus && ctx->br_variant != 3) return launch_br_mb<LOGN, G, PIPE, LAT, BL, ILV, true>(ctx, lane, d);
  constexpr int N = 1 << LOGN, M = N / 2;
  const int n_deg = (ctx->dims.n / G) * ((1 << G) - 1);
  constexpr int SLOTS = (M >= 128 ? M / 128 : 1) * 142;
  const size_t lds = (size_t)2 * SLOTS * 16 + (size_t)M * 16 + (size_t)(M / 2) * 16 + (size_t)M * 16 + (size_t)((n_deg + 7) & ~7) * 2;
  auto kern = dev::blind_rotate_mb_kernel<LOGN, BL, G, PIPE, LAT, ILV, SCAL>;
  if (int rc = ensure_dyn_lds(ctx, reinterpret_cast<const void *>(kern), lds)) return rc;
#ifdef FHES_LAB
  static const int lab_root_mask = getenv("FHES_MB_ROOT_MASK") ? atoi(getenv("FHES_MB_ROOT_MASK")) : -1;
  static const int lab_group_mask = getenv("FHES_MB_GROUP_MASK") ? atoi(getenv("FHES_MB_GROUP_MASK")) : -1;
  hipLaunchKernelGGL(kern, dim3(d.v_count), dim3(M / 4), lds, lane.stream, d, ctx->arena,
                     ctx->ct_stride, lane.ks_out, ctx->ks_stride, ctx->dims.n, ctx->bsk_e,
                     ctx->lut_polys, ctx->twist_inv, ctx->tw, ctx->root, ctx->psi, ctx->cdpp, lab_root_mask, lab_group_mask);
#else
  hipLaunchKernelGGL(kern, dim3(d.v_count), dim3(M / 4), lds, lane.stream, d, ctx->arena,
                     ctx->ct_stride, lane.ks_out, ctx->ks_stride, ctx->dims.n, ctx->bsk_e,
                     ctx->lut_polys, ctx->twist_inv, ctx->tw, ctx->root, ctx->psi, ctx->cdpp);
#endif
  FHES_HIP(hipGetLastError());
  return FHES_OK;
}

// the 512-thread latency build of the multi-bit blind rotate (kernel_br_lat.hip.h); N = 2048
template <int G, bool ILV, int BL = 23>
static int launch_br_mb_lat(fhes_ctx *ctx, Lane &lane, const dev::NodeDesc &d) {
  if (BL == 23 && ctx->dims.pbs_base_log == 18) return launch_br_mb_lat<G, ILV, 18>(ctx, lane, d);
  const int n_deg = (ctx->dims.n / G) * ((1 << G) - 1);
  const size_t lds = dev::lat_mb_lds_bytes(n_deg);
  auto kern = dev::blind_rotate_mb_lat_kernel<BL, G, ILV>;
  if (int rc = ensure_dyn_lds(ctx, reinterpret_cast<const void *>(kern), lds)) return rc;
  hipLaunchKernelGGL(kern, dim3(d.v_count), dim3(dev::kLatThreads), lds, lane.stream, d, ctx->arena, ctx->ct_stride, lane.ks_out,
                     ctx->ks_stride, ctx->dims.n, ctx->bsk_e, ctx->lut_polys, ctx->twist_inv, ctx->tw, ctx->root, ctx->psi);
  FHES_HIP(hipGetLastError());
  return FHES_OK;
}

// the SPLIT multi-bit blind rotate (kernel_br_lat.hip.h): every bundle of every step on the whole device, then the sequential part
template <int G, bool ILV, int BL = 23>
static int launch_br_mb_split(fhes_ctx *ctx, Lane &lane, const dev::NodeDesc &d) {
  if (BL == 23 && ctx->dims.pbs_base_log == 18) return launch_br_mb_split<G, ILV, 18>(ctx, lane, d);
  constexpr int M = 1024;
  const int n_groups = ctx->dims.n / G;
  const size_t need = (size_t)d.v_count * n_groups * 4 * M * 16;
  if (lane.gen_scratch_bytes < need) {
    lane.drop(lane.gen_scratch);
    lane.gen_scratch = nullptr;
    lane.gen_scratch_bytes = 0;
    FHES_HIP(hipMalloc(&lane.gen_scratch, need));
    lane.gen_scratch_bytes = need;
  }
  double2 *bundles = reinterpret_cast<double2 *>(lane.gen_scratch);
  hipLaunchKernelGGL((dev::mb_bundle_kernel<G, ILV>), dim3(n_groups * 4), dim3(256), 0, lane.stream, d.v_count, lane.ks_out, ctx->ks_stride,
                     ctx->dims.n, ctx->bsk_e, ctx->root, bundles);
  FHES_HIP(hipGetLastError());
  auto kern = dev::blind_rotate_mb_seq_kernel<BL>;
  if (int rc = ensure_dyn_lds(ctx, reinterpret_cast<const void *>(kern), dev::kSeqLdsBytes)) return rc;
  hipLaunchKernelGGL(kern, dim3(d.v_count), dim3(dev::kLatThreads), dev::kSeqLdsBytes, lane.stream, d, ctx->arena, ctx->ct_stride, lane.ks_out,
                     ctx->ks_stride, ctx->dims.n, n_groups, bundles, ctx->lut_polys, ctx->twist_inv, ctx->tw, ctx->psi);
  FHES_HIP(hipGetLastError());
  return FHES_OK;
}

template <int LOGN, int PRIO = 0, int BL = 23>
static int launch_br_v5(fhes_ctx *ctx, Lane &lane, const dev::NodeDesc &d) {
  if (BL == 23 && ctx->dims.pbs_base_log == 18) return launch_br_v5<LOGN, PRIO, 18>(ctx, lane, d);
  constexpr int N = 1 << LOGN, M = N / 2;
  constexpr int SLOTS = M;
  const size_t lds = (size_t)2 * N * 8 + (size_t)2 * SLOTS * 16 + (size_t)2 * (M / 2) * 16;  // N = 2048: exactly 80 KB, two per CU
  auto kern = dev::blind_rotate_v5_kernel<LOGN, BL, PRIO>;
  if (int rc = ensure_dyn_lds(ctx, reinterpret_cast<const void *>(kern), lds)) return rc;
  hipLaunchKernelGGL(kern, dim3(d.v_count), dim3(M / 4), lds, lane.stream, d, ctx->arena,
                     ctx->ct_stride, lane.ks_out, ctx->ks_stride, ctx->dims.n, ctx->bsk_e,
                     ctx->lut_polys, ctx->twist_inv64, ctx->tw, ctx->psi, ctx->cdpp);
  FHES_HIP(hipGetLastError());
  return FHES_OK;
}

// the latency build of the classic blind rotate (kernel_br_lat.hip.h): ONE bootstrap on 512 threads, one workgroup per CU
template <int BL = 23, int OPT = 0>
static int launch_br_lat(fhes_ctx *ctx, Lane &lane, const dev::NodeDesc &d) {
  if (BL == 23 && (OPT & 3) == 0 && ctx->dims.pbs_base_log == 18) return launch_br_lat<18, OPT>(ctx, lane, d);
  auto kern = dev::blind_rotate_lat_kernel<BL, OPT>;
  if (int rc = ensure_dyn_lds(ctx, reinterpret_cast<const void *>(kern), dev::kLatLdsBytes)) return rc;
  if ((OPT >> 7) & 1) {   // lab: phase stamps of the first 256 workgroups, written to $FHES_LAT_STAMPS when the context goes away
    if (d.v_count > 256) return fail(FHES_ERR_INVALID, "the stamped latency build takes launches of at most 256 bootstraps");
    if (!ctx->lat_stamps) {
      FHES_HIP(hipMalloc(&ctx->lat_stamps, (size_t)256 * 8 * dev::kLatStamps * 8));
      FHES_HIP(hipMemset(ctx->lat_stamps, 0, (size_t)256 * 8 * dev::kLatStamps * 8));
    }
  }
  hipLaunchKernelGGL(kern, dim3(d.v_count), dim3(dev::kLatThreads), dev::kLatLdsBytes, lane.stream, d, ctx->arena,
                     ctx->ct_stride, lane.ks_out, ctx->ks_stride, ctx->dims.n, ctx->bsk_e, ctx->lut_polys, ctx->twist_inv64,
                     ctx->tw, ctx->psi, ctx->lat_stamps);
  FHES_HIP(hipGetLastError());
  return FHES_OK;
}

// the wave-per-polynomial build of the classic blind rotate (kernel_br_wpp.hip.h): a bootstrap on 128 threads, four per CU
template <int BL = 23, bool STAMP = false, int ABL = 0, int BOOTS = 1, bool TWL = false>
static int launch_br_wpp(fhes_ctx *ctx, Lane &lane, const dev::NodeDesc &d) {
  if (BL == 23 && !STAMP && ABL == 0 && !TWL && BOOTS <= 2 && ctx->dims.pbs_base_log == 18) return launch_br_wpp<18, false, 0, BOOTS>(ctx, lane, d);
  auto kern = dev::blind_rotate_wpp_kernel<BL, STAMP, ABL, BOOTS, TWL>;
  constexpr size_t lds = dev::wpp_lds_bytes(BOOTS, TWL);
  static_assert(lds <= 160 * 1024, "one workgroup's LDS");
  if (int rc = ensure_dyn_lds(ctx, reinterpret_cast<const void *>(kern), lds)) return rc;
  if (STAMP && !ctx->lat_stamps) {   // lab: stamps of the first 1024 workgroups, written to $FHES_LAT_STAMPS when the context goes away
    FHES_HIP(hipMalloc(&ctx->lat_stamps, (size_t)256 * 8 * dev::kLatStamps * 8));
    FHES_HIP(hipMemset(ctx->lat_stamps, 0, (size_t)256 * 8 * dev::kLatStamps * 8));
  }
  hipLaunchKernelGGL(kern, dim3((d.v_count + BOOTS - 1) / BOOTS), dim3(dev::kWppThreads * BOOTS), lds, lane.stream, d, ctx->arena,
                     ctx->ct_stride, lane.ks_out, ctx->ks_stride, ctx->dims.n, ctx->bsk_e, ctx->lut_polys, ctx->twist_inv64,
                     ctx->tw, ctx->psi, STAMP ? ctx->lat_stamps : nullptr, (int)d.v_count);
  FHES_HIP(hipGetLastError());
  return FHES_OK;
}

#ifdef FHES_LAB
// variant 8: three workgroups per CU (48 KB of LDS, <= 168 registers); N = 2048 only
template <int BL = 23, int ABL = 0>
static int launch_br_v8(fhes_ctx *ctx, Lane &lane, const dev::NodeDesc &d) {
  if (BL == 23 && ABL == 0 && ctx->dims.pbs_base_log == 18) return launch_br_v8<18>(ctx, lane, d);
  constexpr int N = 2048, M = N / 2;
  const size_t lds = (size_t)2 * M * 16 + (size_t)2 * (M / 2) * 16;  // 48 KB
  auto kern = dev::blind_rotate_v8_kernel<11, BL, ABL>;
  // the threads' own accumulator words wait in global memory between two updates: 32 KB per workgroup
  const size_t need = (size_t)d.v_count * 8 * 256 * 16;
  if (lane.gen_scratch_bytes < need) {
    lane.drop(lane.gen_scratch);
    lane.gen_scratch = nullptr;
    lane.gen_scratch_bytes = 0;
    FHES_HIP(hipMalloc(&lane.gen_scratch, need));
    lane.gen_scratch_bytes = need;
  }
  if (int rc = ensure_dyn_lds(ctx, reinterpret_cast<const void *>(kern), lds)) return rc;
  hipLaunchKernelGGL(kern, dim3(d.v_count), dim3(M / 4), lds, lane.stream, d, ctx->arena, ctx->ct_stride, lane.ks_out, ctx->ks_stride,
                     ctx->dims.n, ctx->bsk_e, ctx->lut_polys, ctx->twist_inv64, ctx->tw, ctx->psi, ctx->cdpp,
                     reinterpret_cast<ulonglong2 *>(lane.gen_scratch));
  FHES_HIP(hipGetLastError());
  return FHES_OK;
}
#endif  // FHES_LAB

// generic shapes (k <= 3, L <= 3, N = 256 .. 8192): blind_rotate_gen_kernel with its per-workgroup global scratch
static size_t gen_scratch_per_wg(const Dims &d) {
  return (size_t)(d.k + 1) * d.N * 8 + (size_t)(d.k + 1) * d.L * d.M * 16;
}
// LDS-resident form: accumulator + digit spectra next to the exchange buffers when they fit (kGenLdsBudget leaves room for
// the runtime's own use of the 160 KB); otherwise the global scratch
constexpr size_t kGenLdsBudget = 156 * 1024;
static size_t gen_lds_bytes(const Dims &d, bool resident) {
  const size_t base = (size_t)(d.k + 1) * d.M * 16 + (size_t)((d.n + 1 + 3) & ~3) * 4;
  return resident ? base + gen_scratch_per_wg(d) : base;
}
static bool gen_lds_resident(const Dims &d) { return gen_lds_bytes(d, true) <= kGenLdsBudget; }
template <int LOGN>
static int launch_br_gen_n(fhes_ctx *ctx, Lane &lane, const dev::NodeDesc &d) {
  constexpr int N = 1 << LOGN, M = N / 2;
  const Dims &dm = ctx->dims;
  const bool resident = gen_lds_resident(dm) && !getenv("FHES_GEN_GLOBAL_SCRATCH");  // developer knob: the global-scratch form
  const size_t per = gen_scratch_per_wg(dm), need = resident ? 0 : per * (size_t)d.v_count;
  if (lane.gen_scratch_bytes < need) {
    lane.drop(lane.gen_scratch);
    lane.gen_scratch = nullptr;
    lane.gen_scratch_bytes = 0;
    FHES_HIP(hipMalloc(&lane.gen_scratch, need));
    lane.gen_scratch_bytes = need;
  }
  const size_t lds = gen_lds_bytes(dm, resident);
  if (resident) {
    auto kern = dev::blind_rotate_gen_kernel<LOGN, true>;
    if (int rc = ensure_dyn_lds(ctx, reinterpret_cast<const void *>(kern), lds)) return rc;
    hipLaunchKernelGGL(kern, dim3(d.v_count), dim3((dm.k + 1) * (M / 8)), lds, lane.stream, d, ctx->arena, ctx->ct_stride,
                       lane.ks_out, ctx->ks_stride, dm.n, dm.k, dm.L, dm.pbs_base_log, ctx->bsk_f, ctx->lut_polys, ctx->twist,
                       ctx->tw, ctx->psi, (unsigned char *)nullptr, per);
  } else {
    auto kern = dev::blind_rotate_gen_kernel<LOGN, false>;
    if (int rc = ensure_dyn_lds(ctx, reinterpret_cast<const void *>(kern), lds)) return rc;
    hipLaunchKernelGGL(kern, dim3(d.v_count), dim3((dm.k + 1) * (M / 8)), lds, lane.stream, d, ctx->arena, ctx->ct_stride,
                       lane.ks_out, ctx->ks_stride, dm.n, dm.k, dm.L, dm.pbs_base_log, ctx->bsk_f, ctx->lut_polys, ctx->twist,
                       ctx->tw, ctx->psi, lane.gen_scratch, per);
  }
  FHES_HIP(hipGetLastError());
  return FHES_OK;
}
// N = 8192 (glwe_dim 1): one thread group per ciphertext, products accumulated in registers (blind_rotate_big_kernel);
// (a lab build keeps the generic kernel at this size behind FHES_GEN_BIG=0)
template <int LOGN, int LV>
static int launch_br_big_n(fhes_ctx *ctx, Lane &lane, const dev::NodeDesc &d) {
  constexpr int N = 1 << LOGN, M = N / 2;
  const Dims &dm = ctx->dims;
  const size_t per = (size_t)2 * N * 8, need = per * (size_t)d.v_count;
  if (lane.gen_scratch_bytes < need) {
    lane.drop(lane.gen_scratch);
    lane.gen_scratch = nullptr;
    lane.gen_scratch_bytes = 0;
    FHES_HIP(hipMalloc(&lane.gen_scratch, need));
    lane.gen_scratch_bytes = need;
  }
  const size_t lds = (size_t)2 * M * 16 + (size_t)2 * 512 * 16 + (size_t)((dm.n + 1 + 3) & ~3) * 4;
  auto kern = dev::blind_rotate_big_kernel<LOGN, LV>;
  if (int rc = ensure_dyn_lds(ctx, reinterpret_cast<const void *>(kern), lds)) return rc;
  hipLaunchKernelGGL(kern, dim3(d.v_count), dim3(M / 8), lds, lane.stream, d, ctx->arena, ctx->ct_stride, lane.ks_out, ctx->ks_stride,
                     dm.n, dm.pbs_base_log, ctx->bsk_f, ctx->lut_polys, ctx->twist, ctx->tw, ctx->psi, lane.gen_scratch, per);
  FHES_HIP(hipGetLastError());
  return FHES_OK;
}
static bool use_big_kernel(const Dims &dm) {
#ifdef FHES_LAB   // FHES_GEN_BIG=0: the generic kernel at N = 8192 (spills 48 / 114 registers: lab only since round 5)
  if (const char *e = getenv("FHES_GEN_BIG")) {
    if (atoi(e) == 0) return false;
  }
#endif
  return dm.k == 1 && dm.N == 8192;
}
static int launch_br_gen(fhes_ctx *ctx, Lane &lane, const dev::NodeDesc &d) {
  if (use_big_kernel(ctx->dims)) {
    switch (ctx->dims.L) {
      case 1: return launch_br_big_n<13, 1>(ctx, lane, d);
      case 2: return launch_br_big_n<13, 2>(ctx, lane, d);
      default: return launch_br_big_n<13, 3>(ctx, lane, d);
    }
  }
  switch (ctx->dims.N) {
    case 256: return launch_br_gen_n<8>(ctx, lane, d);
    case 512: return launch_br_gen_n<9>(ctx, lane, d);
    case 1024: return launch_br_gen_n<10>(ctx, lane, d);
    case 2048: return launch_br_gen_n<11>(ctx, lane, d);
    case 4096: return launch_br_gen_n<12>(ctx, lane, d);
#ifdef FHES_LAB
    case 8192: return launch_br_gen_n<13>(ctx, lane, d);
#endif
    default: return fail(FHES_ERR_UNSUPPORTED, "poly_size %d", ctx->dims.N);
  }
}

// Blind-rotate builds of the classic 2_2 sets.  The PRODUCT library holds what `fhes_set_kernel_variant(ctx, 0)` can choose
// (VERDICT r4 item 8):
//   0   auto: the latency build for launches of at most one workgroup per CU (ctx->lat_max), the throughput build above
//   2   the latency build (kernel_br_lat.hip.h: one bootstrap on 512 threads; N = 2048)
//   3   the throughput build at every launch size (blind_rotate_v5_kernel, no priority hints; 7 = its number of rounds 2-4)
//   17  the wave-per-polynomial build (kernel_br_wpp.hip.h: one wave per polynomial, two bootstraps per workgroup, two workgroups per CU;
//       N = 2048): 8-10 % above the throughput build per full round of 1024, behind it where its last round is a small remainder
//   12  the generic kernel on a 2_2 set (the second implementation the tests compare with)
// A lab build (-DFHES_LAB: FHES_LAB=1 python -m fhe_string_amd.build, scripts/) also carries the simple reference kernel (1),
// the two wave-priority policies of the throughput kernel (5, 6), variant 8 (three workgroups per CU, 28 % slower) and the
// timing-only ablations with WRONG results (9, 10, 13, 14; 20-26: switches of the latency build).  All correct ones: same bits.
static int launch_blind_rotate(fhes_ctx *ctx, Lane &lane, const dev::NodeDesc &d) {
  if (ctx->dims.g > 1) {
    // multi-bit parameter sets: one kernel in three builds -- latency (launches of at most one
    // workgroup per CU: everything in registers, bundle pipeline), throughput with the pipeline
    // (g = 3) and without (g = 2).  br_variant 2 forces the latency build, 3 the throughput build.
    const bool lat = ctx->br_variant == 2 || (ctx->br_variant == 0 && d.v_count <= ctx->n_cus);
    // the 512-thread latency build: the automatic choice for one-per-CU launches at N = 2048 (variant 2 forces it; variant 4 forces it
    // even when FHES_MB_LAT512=0 switched the automatic choice off)
    // variant 16 (and, once measured, the automatic choice for a handful of bootstraps): bundles first, then the sequential part
    if (ctx->dims.N == 2048 && d.v_count <= dev::kSplitMax && (ctx->br_variant == 16 || (ctx->br_variant == 0 && ctx->mb_split_max > 0 && d.v_count <= ctx->mb_split_max))) {
      if (ctx->dims.g == 3) return ctx->mb_ilv ? launch_br_mb_split<3, true>(ctx, lane, d) : launch_br_mb_split<3, false>(ctx, lane, d);
      return ctx->mb_ilv ? launch_br_mb_split<2, true>(ctx, lane, d) : launch_br_mb_split<2, false>(ctx, lane, d);
    }
    if (ctx->dims.N == 2048 && (ctx->br_variant == 4 || ((ctx->br_variant == 0 || ctx->br_variant == 2) && ctx->mb_lat512 && (ctx->br_variant == 2 || d.v_count <= ctx->lat_max)))) {
      // (g = 3 with the PLAIN key layout -- FHES_MB_INTERLEAVE=0, not the default -- stays on the 256-thread build: that
      // instantiation of the 512-thread kernel spills four registers)
      if (ctx->dims.g == 3 && ctx->mb_ilv) return launch_br_mb_lat<3, true>(ctx, lane, d);
      if (ctx->dims.g == 2) return ctx->mb_ilv ? launch_br_mb_lat<2, true>(ctx, lane, d) : launch_br_mb_lat<2, false>(ctx, lane, d);
    }
    if (ctx->dims.g == 3) {  // measured: the latency build gains nothing for g = 3 (3.21 vs 3.16 ms)
      if (ctx->mb_ilv) return ctx->dims.N == 2048 ? launch_br_mb<11, 3, true, false, 23, true>(ctx, lane, d) : launch_br_mb<8, 3, true, false, 23, true>(ctx, lane, d);
      return ctx->dims.N == 2048 ? launch_br_mb<11, 3, true>(ctx, lane, d) : launch_br_mb<8, 3, true>(ctx, lane, d);
    }
    if (ctx->mb_ilv) {
      if (ctx->dims.N == 2048) return lat ? launch_br_mb<11, 2, true, true, 23, true>(ctx, lane, d) : launch_br_mb<11, 2, false, false, 23, true>(ctx, lane, d);
      return lat ? launch_br_mb<8, 2, true, true, 23, true>(ctx, lane, d) : launch_br_mb<8, 2, false, false, 23, true>(ctx, lane, d);
    }
    if (ctx->dims.N == 2048) return lat ? launch_br_mb<11, 2, true, true>(ctx, lane, d) : launch_br_mb<11, 2, false>(ctx, lane, d);
    return lat ? launch_br_mb<8, 2, true, true>(ctx, lane, d) : launch_br_mb<8, 2, false>(ctx, lane, d);
  }
  if (!fast_shape(ctx->p) || ctx->br_variant == 12) return launch_br_gen(ctx, lane, d);
  const int variant = ctx->br_variant;
  const bool n2048 = ctx->dims.N == 2048;
  if (variant == 2 || (variant == 0 && n2048 && ctx->lat_max > 0 && d.v_count <= ctx->lat_max)) {
    if (!n2048) return fail(FHES_ERR_UNSUPPORTED, "the latency build of the blind rotate is written for N = 2048");
    return launch_br_lat<23, dev::kLatDefaultOpt>(ctx, lane, d);
  }
  if (variant == 17) {
    if (!n2048) return fail(FHES_ERR_UNSUPPORTED, "the wave-per-polynomial build of the blind rotate is written for N = 2048");
    return launch_br_wpp<23, false, 0, 2>(ctx, lane, d);
  }
  // auto: the wave-per-polynomial build (two bootstraps per workgroup, two workgroups per CU: rounds of 1024) is 8-10 % faster per full
  // round than the throughput build (rounds of 512), but a partial last round costs it more.  Both are step functions of the launch
  // size; the choice compares them (ms on the box of profiles/r05_probe33_wpp2_sizes.txt -- only the ORDER matters, and it held on every
  // box measured): 896-1024, 1536, 2048 and everything from 2560 on go to the wave-per-polynomial build, 257-768, 1152-1280 and 1792 stay
  if (variant == 0 && n2048 && ctx->wpp_auto) {
    const int v = d.v_count;
    const int r5 = v % 512, r10 = v % 1024;
    const double t_v5 = 0.45 + 4.22 * (v / 512) + (r5 == 0 ? 0.0 : (r5 <= 256 ? 2.45 : 4.3));
    const double t_wpp = 0.6 + 7.6 * (v / 1024) + (r10 == 0 ? 0.0 : (r10 <= 512 ? 4.0 : 7.9));
    if (t_wpp < t_v5) return launch_br_wpp<23, false, 0, 2>(ctx, lane, d);
  }
  if (variant == 0 || variant == 3 || variant == 7) return n2048 ? launch_br_v5<11, 2>(ctx, lane, d) : launch_br_v5<8, 2>(ctx, lane, d);
#ifdef FHES_LAB
  if (variant >= 27 && variant <= 32) {   // wave-per-polynomial build: phase stamps and timing-only ablations (WRONG results) of its one-bootstrap-per-workgroup form; 30: that form itself
    if (!n2048 || ctx->dims.pbs_base_log != 23) return fail(FHES_ERR_UNSUPPORTED, "the lab builds exist for the default set only");
    // FOUR bootstraps per workgroup (512 threads, one workgroup per CU): 31 with the untwist factors in LDS, 32 without -- written at the
    // end of round 5 without a device in reach: compiled (no spills), NEVER RUN; scripts/r06_probe1.sh checks the bits first
    if (variant == 31) return launch_br_wpp<23, false, 0, 4, true>(ctx, lane, d);
    if (variant == 32) return launch_br_wpp<23, false, 0, 4, false>(ctx, lane, d);
    if (variant == 30) return launch_br_wpp<23, false, 0, 1>(ctx, lane, d);   // ONE bootstrap per workgroup, four workgroups per CU (the first form: 126 k against 136 k)
    if (variant == 28) return launch_br_wpp<23, false, 1>(ctx, lane, d);   // without the per-lane table requests
    if (variant == 29) return launch_br_wpp<23, false, 2>(ctx, lane, d);   // without any global load inside the loop
    return launch_br_wpp<23, true>(ctx, lane, d);
  }
  if (variant >= 20 && variant <= 26) {   // switches of the latency build (scripts/r05_probe*.sh)
    if (!n2048 || ctx->dims.pbs_base_log != 23) return fail(FHES_ERR_UNSUPPORTED, "the lab builds exist for the default set only");
    using dev::lat_opt;
    switch (variant) {
      case 20: return launch_br_lat<23, lat_opt(0, 0, 0, 0)>(ctx, lane, d);
      case 21: return launch_br_lat<23, lat_opt(0, 1, 0, 0)>(ctx, lane, d);     // half barriers through LDS flags
      case 22: return launch_br_lat<23, lat_opt(0, 2, 0, 0)>(ctx, lane, d);     // WRONG results: no half barriers
      case 23: return launch_br_lat<23, lat_opt(0, 0, 0, 1, 0, 1) | (1 << 9) | (1 << 10)>(ctx, lane, d);   // + the in-wave part without LDS (layout v2)
      case 24: return launch_br_lat<23, lat_opt(0, 0, 0, 1, 0, 1) | (1 << 9)>(ctx, lane, d);   // interleaved stores + rotated-read addresses ahead of B6
      case 25: return launch_br_lat<23, lat_opt(0, 0, 0, 1, 0, 1)>(ctx, lane, d);   // stores pair by pair behind their butterflies
      default: return launch_br_lat<23, lat_opt(0, 0, 0, 0, 1)>(ctx, lane, d);  // phase stamps
    }
  }
  if (variant == 8) {
    if (!n2048) return fail(FHES_ERR_UNSUPPORTED, "blind-rotate variant 8 is written for N = 2048");
    return launch_br_v8<>(ctx, lane, d);
  }
  if (variant == 13 || variant == 14) {  // timing-only ablations of variant 8 (WRONG results)
    if (!n2048 || ctx->dims.pbs_base_log != 23) return fail(FHES_ERR_UNSUPPORTED, "the ablation builds exist for the default set only");
    return variant == 13 ? launch_br_v8<23, 1>(ctx, lane, d) : launch_br_v8<23, 2>(ctx, lane, d);
  }
  if (variant == 1) {
    if (ctx->dims.pbs_base_log != 23) return fail(FHES_ERR_UNSUPPORTED, "blind-rotate variant 1 is instantiated for pbs_base_log 23 only");
    return n2048 ? launch_br_v1<11>(ctx, lane, d) : launch_br_v1<8>(ctx, lane, d);
  }
  if (variant == 5) return n2048 ? launch_br_v5<11, 0>(ctx, lane, d) : launch_br_v5<8, 0>(ctx, lane, d);
  if (variant == 6) return n2048 ? launch_br_v5<11, 1>(ctx, lane, d) : launch_br_v5<8, 1>(ctx, lane, d);
  if ((variant == 9 || variant == 10) && n2048) return variant == 9 ? launch_br_v5<11, 4>(ctx, lane, d) : launch_br_v5<11, 5>(ctx, lane, d);
#endif
  return fail(FHES_ERR_UNSUPPORTED, "blind-rotate variant %d is not in this build (0, 2, 3, 7, 12, 17; a lab build has the others)", variant);
}

static int take_event(fhes_ctx *ctx, hipEvent_t *e) {
  if (!ctx->ev_free.empty()) {
    *e = ctx->ev_free.back();
    ctx->ev_free.pop_back();
    return FHES_OK;
  }
  FHES_HIP(hipEventCreate(e));
  return FHES_OK;
}

int collect_timing_locked(fhes_ctx *ctx) {
  if (ctx->pending.empty()) return FHES_OK;
  FHES_HIP(hipStreamSynchronize(ctx->stream));
  for (auto &t : ctx->pending) {
    float ks = 0, br = 0;
    FHES_HIP(hipEventElapsedTime(&ks, t.e0, t.e1));
    FHES_HIP(hipEventElapsedTime(&br, t.e1, t.e2));
    ctx->stats.last_ks_ms = ks; ctx->stats.last_br_ms = br;
    ctx->stats.total_ks_ms += ks; ctx->stats.total_br_ms += br;
    ctx->stats.timed_launches += 1;
    ctx->ev_free.push_back(t.e0); ctx->ev_free.push_back(t.e1); ctx->ev_free.push_back(t.e2);
  }
  ctx->pending.clear();
  return FHES_OK;
}

int launch_pbs(fhes_ctx *ctx, Lane &lane, const DevNodeArrays &a, int32_t first, int32_t count, int32_t n_inst, RunCounters *cnt) {
  if (!ctx->keys_loaded) return fail(FHES_ERR_STATE, "server keys not uploaded");
  if (count <= 0 || n_inst <= 0) return FHES_OK;
  const int64_t total = (int64_t)count * n_inst;
  if (total > INT32_MAX) return fail(FHES_ERR_INVALID, "batch too large");
  int32_t kMaxChunk = 16384;
  if (!fast_shape(ctx->p) || ctx->br_variant == 12)  // the generic kernel's global scratch: at most 1 GiB per launch
    kMaxChunk = (int32_t)std::max<size_t>(64, std::min<size_t>(16384, ((size_t)1 << 30) / gen_scratch_per_wg(ctx->dims)));
#ifdef FHES_LAB   // FHES_LAUNCH_CHUNK=n (64..16384, read at every call): shorter launches -- scripts/r06_probe3.sh asks whether the g = 3
                  // kernel's slow processes (L2 misses x 3.5) are workgroups drifting apart in LONG launches: a launch
                  // boundary every few rounds brings them back together.  Same kernels on sub-ranges: the results cannot change.
  if (const char *e = getenv("FHES_LAUNCH_CHUNK")) {
    const long v = strtol(e, nullptr, 10);
    if (v >= 64 && v <= 16384) kMaxChunk = std::min<int32_t>(kMaxChunk, (int32_t)v);
  }
#endif
  for (int32_t off = 0; off < (int32_t)total; off += kMaxChunk) {
    const int32_t c = std::min<int32_t>(kMaxChunk, (int32_t)total - off);
    int rc = ensure_ks_out(ctx, lane, c);
    if (rc) return rc;
    dev::NodeDesc d = make_desc(a, first, count, off, c);
    fhes_ctx::Timed t{};
    const bool timed = ctx->timing && lane.is_main;  // the event lists belong to the main lane's holder
    if (timed) {
      if ((rc = take_event(ctx, &t.e0)) || (rc = take_event(ctx, &t.e1)) || (rc = take_event(ctx, &t.e2))) return rc;
      FHES_HIP(hipEventRecord(t.e0, lane.stream));
    }
    rc = launch_keyswitch(ctx, lane, d);
    if (rc) return rc;
    if (timed) FHES_HIP(hipEventRecord(t.e1, lane.stream));
    rc = launch_blind_rotate(ctx, lane, d);
    if (rc) return rc;
    if (timed) {
      FHES_HIP(hipEventRecord(t.e2, lane.stream));
      ctx->pending.push_back(t);
    }
    cnt->pbs += (uint64_t)c;
    cnt->launches += 1;
  }
  return FHES_OK;
}

int launch_linear(fhes_ctx *ctx, Lane &lane, const DevNodeArrays &a, int32_t first, int32_t count, int32_t n_inst, RunCounters *cnt) {
  if (count <= 0 || n_inst <= 0) return FHES_OK;
  dev::NodeDesc d = make_desc(a, first, count, 0, count * n_inst);
  hipLaunchKernelGGL(dev::linear_kernel, dim3(count * n_inst), dim3(256), 0, lane.stream, d, ctx->arena,
                     ctx->ct_stride, ctx->dims.big);
  FHES_HIP(hipGetLastError());
  cnt->linear += (uint64_t)count * n_inst;
  return FHES_OK;
}

static int ensure_desc(fhes_ctx *ctx, size_t bytes) {
  if (bytes <= ctx->desc_cap) return FHES_OK;
  if (ctx->desc_buf) FHES_HIP(hipFree(ctx->desc_buf));
  ctx->desc_buf = nullptr;
  ctx->desc_cap = 0;
  const size_t cap = std::max<size_t>(bytes * 2, 1 << 20);
  FHES_HIP(hipMalloc(&ctx->desc_buf, cap));
  ctx->desc_cap = cap;
  return FHES_OK;
}

// copies host descriptor arrays into the ctx staging buffer and returns device views
static int stage_desc(fhes_ctx *ctx, int32_t count, const int32_t *term_ptr, const int32_t *term_idx,
                      const int64_t *term_coef, const uint64_t *consts, const int32_t *lut_ids,
                      const int32_t *out, DevNodeArrays *a) {
  if (count <= 0 || !term_ptr || !consts || !out) return fail(FHES_ERR_INVALID, "bad batch arguments");
  const int32_t nt = term_ptr[count];
  if (term_ptr[0] != 0 || nt < 0 || (nt > 0 && (!term_idx || !term_coef))) return fail(FHES_ERR_INVALID, "bad term arrays");
  for (int32_t t = 0; t < nt; t++)
    if (term_idx[t] < 0 || term_idx[t] >= ctx->capacity) return fail(FHES_ERR_INVALID, "term handle out of range");
  for (int32_t j = 0; j < count; j++) {
    if (out[j] < 0 || out[j] >= ctx->capacity) return fail(FHES_ERR_INVALID, "output handle out of range");
    if (term_ptr[j + 1] < term_ptr[j]) return fail(FHES_ERR_INVALID, "term_ptr not monotone");
    if (lut_ids && (lut_ids[j] < 0 || lut_ids[j] >= (int)ctx->lut_tables.size())) return fail(FHES_ERR_INVALID, "unknown lut id");
  }
  auto al = [](size_t v) { return (v + 15) & ~(size_t)15; };
  const size_t o_tp = 0, o_ti = al(o_tp + 4 * (size_t)(count + 1)), o_tc = al(o_ti + 4 * (size_t)std::max(nt, 1)),
               o_c = al(o_tc + 8 * (size_t)std::max(nt, 1)), o_l = al(o_c + 8 * (size_t)count),
               o_o = al(o_l + 4 * (size_t)count), total = al(o_o + 4 * (size_t)count);
  int rc = ensure_desc(ctx, total);
  if (rc) return rc;
  std::vector<unsigned char> h(total, 0);
  memcpy(&h[o_tp], term_ptr, 4 * (size_t)(count + 1));
  if (nt) { memcpy(&h[o_ti], term_idx, 4 * (size_t)nt); memcpy(&h[o_tc], term_coef, 8 * (size_t)nt); }
  memcpy(&h[o_c], consts, 8 * (size_t)count);
  if (lut_ids) memcpy(&h[o_l], lut_ids, 4 * (size_t)count);
  memcpy(&h[o_o], out, 4 * (size_t)count);
  FHES_HIP(hipStreamSynchronize(ctx->stream));  // staging buffer may still be in use
  FHES_HIP(hipMemcpy(ctx->desc_buf, h.data(), total, hipMemcpyHostToDevice));
  unsigned char *b = ctx->desc_buf;
  a->term_ptr = (const int32_t *)(b + o_tp); a->term_idx = (const int32_t *)(b + o_ti);
  a->term_coef = (const int64_t *)(b + o_tc); a->consts = (const uint64_t *)(b + o_c);
  a->lut_ids = (const int32_t *)(b + o_l); a->out_idx = (const int32_t *)(b + o_o);
  a->slot_map = nullptr;
  a->n_slots = 0;
  return FHES_OK;
}

}  // namespace fhes

using namespace fhes;

extern "C" {

fhes_ctx *fhes_ctx_create(const fhes_params *p, int32_t device) {
  std::string why;
  if (!p || !params_valid(*p, &why)) { fail(FHES_ERR_INVALID, "invalid parameters: %s", why.c_str()); return nullptr; }
  if (!kernels_support(*p, &why)) { fail(FHES_ERR_UNSUPPORTED, "%s", why.c_str()); return nullptr; }
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) {
    fail(FHES_ERR_NO_DEVICE, "no HIP device available (%s); this engine has no CPU fallback",
         e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    return nullptr;
  }
  if (device < 0 || device >= ndev) { fail(FHES_ERR_INVALID, "device %d out of range (%d devices)", device, ndev); return nullptr; }
  auto *ctx = new fhes_ctx(*p);
  ctx->device = device;
  const Dims &d = ctx->dims;
  auto bail = [&](hipError_t err, const char *what) -> fhes_ctx * {
    hip_fail(err, what);
    fhes_ctx_destroy(ctx);
    return nullptr;
  };
  if ((e = hipSetDevice(device)) != hipSuccess) return bail(e, "hipSetDevice");
  if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) return bail(e, "hipStreamCreate");
  ctx->main.stream = ctx->stream;
  ctx->main.is_main = true;

  {
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) return bail(e, "hipGetDeviceProperties");
    ctx->n_cus = prop.multiProcessorCount;
  }
  // launches of at most one workgroup per CU take the latency build (kernel_br_lat.hip.h); FHES_LAT_MAX=0 keeps the throughput
  // build for every size, another number moves the switch-over (developer knob, scripts/lat_probe.py)
  ctx->lat_max = ctx->n_cus;
  if (const char *e6 = getenv("FHES_LAT_MAX")) ctx->lat_max = atoi(e6) < 0 ? 0 : atoi(e6);
  // multi-bit sets at N = 2048: one-per-CU launches take the 512-thread latency build (g = 2: 2.05 against 2.52 ms per bootstrap,
  // g = 3: 2.24 against 2.34; FHES_MB_LAT512=0 keeps the 256-thread one, which also serves the N = 256 test sets)
  ctx->mb_lat512 = d.g > 1 && d.N == 2048;
  if (const char *e7 = getenv("FHES_MB_LAT512")) ctx->mb_lat512 = atoi(e7) != 0;
  // ... and launches of up to 128 (g = 2) / 192 (g = 3) bootstraps the SPLIT form: every bundle of every step first, on the whole
  // device, then the sequential kernel (0.75 ms per g = 3 bootstrap against 2.24 fused; break-even at ~160 / ~256 bootstraps:
  // profiles/r05_probe20_split_crossover.txt; 16-24 MB of scratch per bootstrap).  FHES_MB_SPLIT_MAX=0 switches it off.
  ctx->mb_split_max = (d.g > 1 && d.N == 2048) ? (d.g == 3 ? 192 : 128) : 0;
  if (const char *e8 = getenv("FHES_MB_SPLIT_MAX")) ctx->mb_split_max = atoi(e8) < 0 ? 0 : atoi(e8);
  if (const char *e9 = getenv("FHES_WPP_AUTO")) ctx->wpp_auto = atoi(e9) != 0;
  if (const char *e5 = getenv("FHES_KS_MFMA_MIN")) ctx->ks_mfma_min = atoi(e5);  // developer knob: 0 = VALU keyswitch only, 1 = always MFMA
  if (const char *e3 = getenv("FHES_KS_TB")) ctx->ks_tb = atoi(e3);  // developer knob: 8 / 16 force the tile height
  if (const char *e2 = getenv("FHES_KS_MIN_CPS")) {
    const int v = atoi(e2);
    if (v >= 1 && v <= dev::KS_IC && (v & (v - 1)) == 0) ctx->ks_min_cps = v;
  }
  ctx->ct_stride = d.big + 16;
  const char *env = getenv("FHES_ARENA_SLOTS");
  ctx->arena_fixed = env != nullptr;
  ctx->capacity = env ? atoll(env) : (int64_t)1 << 16;  // 2^16 slots * 16.5 KB = 1.1 GB; grows on demand
  if (ctx->capacity < 64) ctx->capacity = 64;
  if ((e = hipMalloc(&ctx->arena, (size_t)ctx->capacity * ctx->ct_stride * 8)) != hipSuccess) return bail(e, "hipMalloc(arena)");
  ctx->free_ranges[0] = (int32_t)ctx->capacity;

  ctx->ks_stride = ((d.n + 1 + 255) / 256) * 256;
  ctx->lut_capacity = 256;
  if ((e = hipMalloc(&ctx->lut_polys, (size_t)ctx->lut_capacity * d.N * 8)) != hipSuccess) return bail(e, "hipMalloc(luts)");

  std::vector<double> twist, tw;
  host_fft_tables(d.N, twist, tw);
  if ((e = hipMalloc(&ctx->twist, twist.size() * 8)) != hipSuccess) return bail(e, "hipMalloc(twist)");
  if ((e = hipMalloc(&ctx->tw, tw.size() * 8)) != hipSuccess) return bail(e, "hipMalloc(tw)");
  if ((e = hipMemcpy(ctx->twist, twist.data(), twist.size() * 8, hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "copy twist");
  if ((e = hipMemcpy(ctx->tw, tw.data(), tw.size() * 8, hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "copy tw");
  {
    std::vector<double> ti(twist);
    for (auto &v : ti) v *= 1.0 / (double)d.M;  // power of two: exact
    if ((e = hipMalloc(&ctx->twist_inv, ti.size() * 8)) != hipSuccess) return bail(e, "hipMalloc(twist_inv)");
    if ((e = hipMemcpy(ctx->twist_inv, ti.data(), ti.size() * 8, hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "copy twist_inv");
    for (auto &v : ti) v *= 0x1p-64;  // variant 5 rounds y * 2^-64 to the torus (torus_from_f64_pair): exact as well
    if ((e = hipMalloc(&ctx->twist_inv64, ti.size() * 8)) != hipSuccess) return bail(e, "hipMalloc(twist_inv64)");
    if ((e = hipMemcpy(ctx->twist_inv64, ti.data(), ti.size() * 8, hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "copy twist_inv64");
  }
  {  // forward-transform tables; root also gives the Fourier images of monomials (multi-bit)
    std::vector<double> root, psi, odd_fix, cdpp;
    host_fwd_tables(d.N, root, psi, odd_fix, cdpp);
    if ((e = hipMalloc(&ctx->root, root.size() * 8)) != hipSuccess) return bail(e, "hipMalloc(root)");
    if ((e = hipMemcpy(ctx->root, root.data(), root.size() * 8, hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "copy root");
    if ((e = hipMalloc(&ctx->psi, psi.size() * 8)) != hipSuccess) return bail(e, "hipMalloc(psi)");
    if ((e = hipMemcpy(ctx->psi, psi.data(), psi.size() * 8, hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "copy psi");
    if ((e = hipMalloc(&ctx->odd_fix, odd_fix.size() * 8)) != hipSuccess) return bail(e, "hipMalloc(odd_fix)");
    if ((e = hipMemcpy(ctx->odd_fix, odd_fix.data(), odd_fix.size() * 8, hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "copy odd_fix");
    if ((e = hipMalloc(&ctx->cdpp, cdpp.size() * 8)) != hipSuccess) return bail(e, "hipMalloc(cdpp)");
    if ((e = hipMemcpy(ctx->cdpp, cdpp.data(), cdpp.size() * 8, hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "copy cdpp");
  }
  {
    std::lock_guard<std::mutex> lk(g_live_mu);
    static uint64_t next_uid = 1;
    ctx->uid = next_uid++;
    g_live.insert(ctx);
  }
  return ctx;
}

void fhes_ctx_destroy(fhes_ctx *ctx) {
  if (!ctx) return;
  {
    std::lock_guard<std::mutex> lk(g_live_mu);
    g_live.erase(ctx);
  }
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  if (ctx->lat_stamps) {   // lab: the stamped latency build leaves its record behind
    if (const char *path = getenv("FHES_LAT_STAMPS")) {
      std::vector<unsigned long long> h((size_t)256 * 8 * dev::kLatStamps);
      if (hipMemcpy(h.data(), ctx->lat_stamps, h.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
        if (FILE *f = fopen(path, "wb")) { fwrite(h.data(), 8, h.size(), f); fclose(f); }
      }
    }
    (void)hipFree(ctx->lat_stamps);
  }
  (void)hipFree(ctx->arena); (void)hipFree(ctx->ksk); (void)hipFree(ctx->ksk_bias); (void)hipFree(ctx->bsk_f); (void)hipFree(ctx->bsk_std); (void)hipFree(ctx->bsk_e_base ? ctx->bsk_e_base : (void *)ctx->bsk_e);
  (void)hipFree(ctx->twist); (void)hipFree(ctx->tw); (void)hipFree(ctx->twist_inv); (void)hipFree(ctx->twist_inv64); (void)hipFree(ctx->root); (void)hipFree(ctx->psi); (void)hipFree(ctx->odd_fix); (void)hipFree(ctx->cdpp); (void)hipFree(ctx->lut_polys);
  (void)hipDeviceSynchronize();
  auto drop_lane = [](Lane &l) {
    (void)hipFree(l.ks_out); (void)hipFree(l.ks_dig); (void)hipFree(l.ks_body); (void)hipFree(l.slot_map); (void)hipFree(l.gen_scratch);
    for (void *p : l.retired) (void)hipFree(p);
  };
  drop_lane(ctx->main);
  for (auto &l : ctx->side) {
    drop_lane(*l);
    if (l->stream) (void)hipStreamDestroy(l->stream);
  }
  for (auto &mp : ctx->main_pending) (void)hipEventDestroy(mp.ev);
  for (auto &ev : ctx->pending_ev_free) (void)hipEventDestroy(ev);
  (void)hipFree(ctx->ksk_pk); (void)hipFree(ctx->desc_buf);
  for (auto &t : ctx->pending) { (void)hipEventDestroy(t.e0); (void)hipEventDestroy(t.e1); (void)hipEventDestroy(t.e2); }
  for (auto &ev : ctx->ev_free) (void)hipEventDestroy(ev);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

int fhes_upload_keys(fhes_ctx *ctx, const uint64_t *bsk, const uint64_t *ksk) {
  if (!ctx || !bsk || !ksk) return fail(FHES_ERR_INVALID, "null argument");
  MainLaneHold hold(ctx);
  FHES_HIP(hipSetDevice(ctx->device));
  FHES_HIP(hipDeviceSynchronize());  // no lane may still read the keys that are replaced
  const Dims &d = ctx->dims;
  // keyswitch key, rows padded to ks_stride words
  {
    const size_t rows = (size_t)d.big * d.ks_level;
    std::vector<uint64_t> padded(rows * ctx->ks_stride, 0);
    for (size_t r = 0; r < rows; r++)
      memcpy(&padded[r * ctx->ks_stride], ksk + r * (d.n + 1), sizeof(uint64_t) * (d.n + 1));
    if (!ctx->ksk) FHES_HIP(hipMalloc(&ctx->ksk, padded.size() * 8));
    FHES_HIP(hipMemcpy(ctx->ksk, padded.data(), padded.size() * 8, hipMemcpyHostToDevice));
    std::vector<uint64_t> bias(ctx->ks_stride, 0);
    for (size_t r = 0; r < rows; r++)
      for (int j = 0; j <= d.n; j++) bias[j] += padded[r * ctx->ks_stride + j];
    const uint64_t half_base = 1ull << (d.ks_base_log - 1);
    for (auto &v : bias) v *= half_base;
    if (!ctx->ksk_bias) FHES_HIP(hipMalloc(&ctx->ksk_bias, bias.size() * 8));
    FHES_HIP(hipMemcpy(ctx->ksk_bias, bias.data(), bias.size() * 8, hipMemcpyHostToDevice));
    // the same key as signed byte planes in MFMA B-fragment order (kernels.hip.h K1b):
    // kpk[k / 64][word / 16][byte plane][lane][16], k = l * big + i, lane = word % 16 + 16 * ((k % 64) / 16)
    if (((size_t)d.big * d.ks_level) % 64 == 0) {
      const int ksteps = d.big * d.ks_level / 64, n_wb = (d.n + 1 + 15) / 16;
      std::vector<int8_t> pk((size_t)ksteps * n_wb * 8 * 64 * 16, 0);
      for (int k = 0; k < ksteps * 64; k++) {
        const int l = k / d.big, i = k % d.big;
        const uint64_t *krow = &padded[((size_t)i * d.ks_level + l) * ctx->ks_stride];
        const int s = k >> 6, g = (k & 63) >> 4, j = k & 15;
        for (int word = 0; word <= d.n; word++) {
          uint64_t v = krow[word];
          const size_t base = ((((size_t)s * n_wb + (word >> 4)) * 8) * 64 + ((word & 15) + 16 * g)) * 16 + j;
          for (int t = 0; t < 8; t++) {
            const int8_t sb = (int8_t)(uint8_t)(v & 0xFF);
            v = (v >> 8) + (sb < 0 ? 1 : 0);
            pk[base + (size_t)t * 64 * 16] = sb;
          }
        }
      }
      if (!ctx->ksk_pk) FHES_HIP(hipMalloc(&ctx->ksk_pk, pk.size()));
      FHES_HIP(hipMemcpy(ctx->ksk_pk, pk.data(), pk.size(), hipMemcpyHostToDevice));
      ctx->ks_n_wb = n_wb;
    }
  }
  // bootstrapping key -> Fourier domain on the device
  {
    const size_t npoly = (size_t)d.n_ggsw * d.L * (d.k + 1) * (d.k + 1);
    // the standard-domain words stay on the device (48.6 MB of 288 GB for the default set): a server that RECEIVED its key can
    // hand it out again -- `#[derive(Serialize)] struct ServerKey`, server_key.rs:11 -- without a client key (fhes_download_keys)
    if (!ctx->bsk_std) FHES_HIP(hipMalloc(&ctx->bsk_std, npoly * d.N * 8));
    uint64_t *tmp = ctx->bsk_std;
    FHES_HIP(hipMemcpy(tmp, bsk, npoly * d.N * 8, hipMemcpyHostToDevice));
    if (!ctx->bsk_f) FHES_HIP(hipMalloc(&ctx->bsk_f, npoly * d.M * 16));
    const size_t lds = (size_t)d.M * 16;
#define FHES_B2F(LG) hipLaunchKernelGGL((dev::bsk_to_fourier_kernel<LG>), dim3(npoly), dim3(d.M / 8), lds, ctx->stream, tmp, ctx->bsk_f, ctx->psi, ctx->odd_fix)
    switch (d.N) {
      case 256: FHES_B2F(8); break;
      case 512: FHES_B2F(9); break;
      case 1024: FHES_B2F(10); break;
      case 2048: FHES_B2F(11); break;
      case 4096: FHES_B2F(12); break;
      default: FHES_B2F(13); break;
    }
#undef FHES_B2F
    FHES_HIP(hipGetLastError());
    if (fast_shape(ctx->p)) {  // the register-order layout of the throughput kernels
      if (!ctx->bsk_e) {
        size_t pad = 0;
#ifdef FHES_LAB   // round-4 experiment (does the multi-bit kernel's rate depend on where the key lies?): FHES_BSK_E_PAD bytes, at
                  // most 1 GiB, in front of the engine-layout key; the padded allocation is released through bsk_e_base
        if (const char *pad_s = getenv("FHES_BSK_E_PAD")) pad = std::min<size_t>((size_t)strtoull(pad_s, nullptr, 0) & ~(size_t)255, (size_t)1 << 30);
#endif
        FHES_HIP(hipMalloc(&ctx->bsk_e_base, npoly * d.M * 16 + pad));
        ctx->bsk_e = reinterpret_cast<double2 *>(reinterpret_cast<unsigned char *>(ctx->bsk_e_base) + pad);
#ifdef FHES_LAB
        if (getenv("FHES_DEBUG_ADDR")) fprintf(stderr, "[fhes] bsk_e at %p (%zu bytes, pad %zu)\n", (void *)ctx->bsk_e, npoly * d.M * 16, pad);
#endif
      }
      // multi-bit keys: the 2^g GGSWs of a group interleaved row by row (round 4: g = 3 runs 7-10 % faster from it in alternating
      // same-run launches, profiles/r04_ab_multibit_layout.txt).  Default: on for g = 3; FHES_MB_INTERLEAVE=0 / 1 forces it.
      {
        const char *e = getenv("FHES_MB_INTERLEAVE");
        ctx->mb_ilv = d.g > 1 ? (e ? (atoi(e) != 0) : (d.g == 3)) : 0;
      }
      const int nc_ilv = ctx->mb_ilv ? (1 << d.g) : 0;
      if (d.N == 2048)
        hipLaunchKernelGGL((dev::bsk_to_engine_layout_kernel<11>), dim3(npoly), dim3(d.M / 8), 0, ctx->stream, ctx->bsk_f, ctx->bsk_e, nc_ilv);
      else
        hipLaunchKernelGGL((dev::bsk_to_engine_layout_kernel<8>), dim3(npoly), dim3(d.M / 8), 0, ctx->stream, ctx->bsk_f, ctx->bsk_e, nc_ilv);
      FHES_HIP(hipGetLastError());
    }
    FHES_HIP(hipStreamSynchronize(ctx->stream));
#ifdef FHES_LAB   // lab: K copies of the engine-layout key in allocations of their own (does the rate follow the PLACEMENT of the key?
                  // fhes_set_kernel_variant(ctx, 100 + i) switches to copy i; scripts/r05_probe12.sh)
    if (const char *e = getenv("FHES_BSK_E_CANDIDATES")) {
      const int K = atoi(e);
      if (ctx->bsk_e && ctx->lab_bsk_e.empty() && K > 1 && K <= 16) {
        ctx->lab_bsk_e.push_back(ctx->bsk_e);
        for (int i = 1; i < K; i++) {
          double2 *q = nullptr;
          void *hole = nullptr;
          if (getenv("FHES_BSK_E_HOLES")) (void)hipMalloc(&hole, (size_t)(1 + i) * 3 * 1024 * 1024 + 4096 * i);   // perturb the allocator between candidates (leaked on purpose)
          FHES_HIP(hipMalloc(&q, npoly * d.M * 16));
          FHES_HIP(hipMemcpy(q, ctx->bsk_e, npoly * d.M * 16, hipMemcpyDeviceToDevice));
          ctx->lab_bsk_e.push_back(q);
        }
      }
    }
#endif
  }
  ctx->keys_loaded = true;
  return FHES_OK;
}

// The inverse of fhes_upload_keys: the standard-domain bootstrapping key exactly as it was uploaded (or received by a key
// broadcast) and the keyswitching key without its row padding -- the words a server-side `ServerKey::serialize` writes.
int fhes_download_keys(fhes_ctx *ctx, uint64_t *bsk, uint64_t *ksk) {
  if (!ctx || !bsk || !ksk) return fail(FHES_ERR_INVALID, "null argument");
  if (!ctx->keys_loaded || !ctx->bsk_std || !ctx->ksk) return fail(FHES_ERR_STATE, "no server keys on this context");
  MainLaneHold hold(ctx);
  FHES_HIP(hipSetDevice(ctx->device));
  const Dims &d = ctx->dims;
  const size_t npoly = (size_t)d.n_ggsw * d.L * (d.k + 1) * (d.k + 1);
  FHES_HIP(hipMemcpy(bsk, ctx->bsk_std, npoly * d.N * 8, hipMemcpyDeviceToHost));
  const size_t rows = (size_t)d.big * d.ks_level, w = (size_t)(d.n + 1) * 8;
  FHES_HIP(hipMemcpy2D(ksk, w, ctx->ksk, (size_t)ctx->ks_stride * 8, w, rows, hipMemcpyDeviceToHost));
  return FHES_OK;
}

int32_t fhes_lut_register(fhes_ctx *ctx, const uint64_t *table, int32_t n) {
  if (!ctx) return fail(FHES_ERR_INVALID, "null ctx");
  std::lock_guard<std::mutex> lk(ctx->mu);
  FHES_HIP(hipSetDevice(ctx->device));
  return lut_register_locked(ctx, table, n);
}

int32_t fhes_arena_alloc(fhes_ctx *ctx, int32_t count) {
  if (!ctx) return fail(FHES_ERR_INVALID, "null ctx");
  std::lock_guard<std::mutex> lk(ctx->mu);
  FHES_HIP(hipSetDevice(ctx->device));
  return arena_alloc_locked(ctx, count);
}

int fhes_arena_free(fhes_ctx *ctx, int32_t first, int32_t count) {
  if (!ctx) return fail(FHES_ERR_INVALID, "null ctx");
  if (!ctx_is_live(ctx)) return FHES_OK;  // the arena went away with its context
  std::lock_guard<std::mutex> lk(ctx->mu);
  FHES_HIP(hipSetDevice(ctx->device));
  FHES_HIP(hipStreamSynchronize(ctx->stream));  // slots may still be read by queued work
  return arena_free_locked(ctx, first, count);
}

int fhes_arena_upload(fhes_ctx *ctx, int32_t first, int32_t count, const uint64_t *host) {
  if (!ctx || !host || first < 0 || count <= 0 || (int64_t)first + count > ctx->capacity)
    return fail(FHES_ERR_INVALID, "arena_upload: bad range");
  std::lock_guard<std::mutex> lk(ctx->mu);
  FHES_HIP(hipSetDevice(ctx->device));
  const size_t w = (size_t)ctx->dims.ct_words() * 8;
  FHES_HIP(hipMemcpy2DAsync(ctx->arena + (size_t)first * ctx->ct_stride, (size_t)ctx->ct_stride * 8, host, w, w,
                            count, hipMemcpyHostToDevice, ctx->stream));
  FHES_HIP(hipStreamSynchronize(ctx->stream));
  ctx->host_bytes_up += (uint64_t)w * (uint64_t)count;
  return FHES_OK;
}

int fhes_arena_download(fhes_ctx *ctx, int32_t first, int32_t count, uint64_t *host) {
  if (ctx && !ctx_is_live(ctx)) return fail(FHES_ERR_STATE, "the server key owning this ciphertext was destroyed");
  if (!ctx || !host || first < 0 || count <= 0 || (int64_t)first + count > ctx->capacity)
    return fail(FHES_ERR_INVALID, "arena_download: bad range");
  std::lock_guard<std::mutex> lk(ctx->mu);
  FHES_HIP(hipSetDevice(ctx->device));
  const size_t w = (size_t)ctx->dims.ct_words() * 8;
  FHES_HIP(hipMemcpy2DAsync(host, w, ctx->arena + (size_t)first * ctx->ct_stride, (size_t)ctx->ct_stride * 8, w,
                            count, hipMemcpyDeviceToHost, ctx->stream));
  FHES_HIP(hipStreamSynchronize(ctx->stream));
  ctx->host_bytes_down += (uint64_t)w * (uint64_t)count;
  return FHES_OK;
}

int fhes_pbs_batch(fhes_ctx *ctx, int32_t count, const int32_t *term_ptr, const int32_t *term_idx,
                   const int64_t *term_coef, const uint64_t *consts, const int32_t *lut_ids,
                   const int32_t *out_handles) {
  if (!ctx || !lut_ids) return fail(FHES_ERR_INVALID, "null argument");
  MainLaneHold hold(ctx);
  FHES_HIP(hipSetDevice(ctx->device));
  DevNodeArrays a;
  int rc = stage_desc(ctx, count, term_ptr, term_idx, term_coef, consts, lut_ids, out_handles, &a);
  if (rc) return rc;
  RunCounters cnt;
  ctx->main_untracked = true;
  rc = launch_pbs(ctx, ctx->main, a, 0, count, 1, &cnt);
  merge_counters_locked(ctx, cnt);
  return rc;
}

int fhes_linear_batch(fhes_ctx *ctx, int32_t count, const int32_t *term_ptr, const int32_t *term_idx,
                      const int64_t *term_coef, const uint64_t *consts, const int32_t *out_handles) {
  if (!ctx) return fail(FHES_ERR_INVALID, "null argument");
  MainLaneHold hold(ctx);
  FHES_HIP(hipSetDevice(ctx->device));
  DevNodeArrays a;
  int rc = stage_desc(ctx, count, term_ptr, term_idx, term_coef, consts, nullptr, out_handles, &a);
  if (rc) return rc;
  RunCounters cnt;
  ctx->main_untracked = true;
  rc = launch_linear(ctx, ctx->main, a, 0, count, 1, &cnt);
  merge_counters_locked(ctx, cnt);
  return rc;
}

int fhes_sync(fhes_ctx *ctx) {
  if (!ctx) return fail(FHES_ERR_INVALID, "null ctx");
  FHES_HIP(hipSetDevice(ctx->device));
  FHES_HIP(hipStreamSynchronize(ctx->stream));
  return FHES_OK;
}

int fhes_get_stats(fhes_ctx *ctx, fhes_stats *out) {
  if (!ctx || !out) return fail(FHES_ERR_INVALID, "null argument");
  MainLaneHold hold(ctx);  // the timing event lists belong to whoever holds the main lane
  FHES_HIP(hipSetDevice(ctx->device));
  int rc = collect_timing_locked(ctx);
  if (rc) return rc;
  ctx->stats.arena_capacity = (uint64_t)ctx->capacity;
  ctx->stats.arena_in_use = (uint64_t)ctx->in_use;
  ctx->stats.host_bytes_up = ctx->host_bytes_up.load();
  ctx->stats.host_bytes_down = ctx->host_bytes_down.load();
  *out = ctx->stats;
  return FHES_OK;
}

int fhes_reset_stats(fhes_ctx *ctx) {
  if (!ctx) return fail(FHES_ERR_INVALID, "null ctx");
  MainLaneHold hold(ctx);
  FHES_HIP(hipSetDevice(ctx->device));
  int rc = collect_timing_locked(ctx);
  if (rc) return rc;
  ctx->stats = fhes_stats{};
  ctx->host_bytes_up = 0;
  ctx->host_bytes_down = 0;
  return FHES_OK;
}

int fhes_set_timing(fhes_ctx *ctx, int32_t enabled) {
  if (!ctx) return fail(FHES_ERR_INVALID, "null ctx");
  ctx->timing = enabled != 0;
  return FHES_OK;
}

int fhes_set_kernel_variant(fhes_ctx *ctx, int32_t v) {
  if (!ctx) return fail(FHES_ERR_INVALID, "null ctx");
  const bool product = v == 0 || v == 2 || v == 3 || v == 4 || v == 7 || v == 12 || v == 16 || v == 17;
#ifdef FHES_LAB
  if (v >= 100 && v < 100 + (int)ctx->lab_bsk_e.size()) {   // lab: switch to another copy of the engine-layout key
    (void)hipDeviceSynchronize();
    ctx->bsk_e = ctx->lab_bsk_e[(size_t)(v - 100)];
    fprintf(stderr, "[fhes lab] bsk_e copy %d at %p\n", v - 100, (void *)ctx->bsk_e);
    return FHES_OK;
  }
  const bool wrong = v == 9 || v == 10 || v == 13 || v == 14 || v == 22 || v == 28 || v == 29;   // timing-only ablations
  const bool lab = v == 1 || v == 5 || v == 6 || v == 8 || wrong || (v >= 20 && v <= 32);
  if (wrong) {   // a developer has to ask for wrong results by name
    const char *e = getenv("FHES_DEV_ABLATIONS");
    if (!e || std::string(e) != "1")
      return fail(FHES_ERR_UNSUPPORTED, "blind-rotate variant %d is a timing-only ablation with wrong results (set FHES_DEV_ABLATIONS=1)", v);
  }
#else
  const bool lab = false;
#endif
  if (!product && !lab)
    return fail(v < 0 || v > 26 ? FHES_ERR_INVALID : FHES_ERR_UNSUPPORTED,
                "blind-rotate variant %d is not in this build (0 = auto, 2 = latency build, 3 / 7 = throughput build, 17 = wave-per-polynomial build, 12 = generic kernel)", v);
  ctx->br_variant = v;
  return FHES_OK;
}

int fhes_debug_bsk_fourier(fhes_ctx *ctx, double *out) {
  if (!ctx || !out) return fail(FHES_ERR_INVALID, "null argument");
  if (!ctx->keys_loaded) return fail(FHES_ERR_STATE, "server keys not uploaded");
  FHES_HIP(hipSetDevice(ctx->device));
  FHES_HIP(hipMemcpy(out, ctx->bsk_f, (size_t)ctx->dims.n_ggsw * ctx->dims.L * (ctx->dims.k + 1) * (ctx->dims.k + 1) * ctx->dims.M * 16, hipMemcpyDeviceToHost));
  return FHES_OK;
}

int fhes_debug_fwd_tables(const fhes_params *p, double *psi, double *odd_fix) {
  if (!p || !psi || !odd_fix) return fail(FHES_ERR_INVALID, "null argument");
  std::vector<double> root, a, b, c;
  host_fwd_tables(p->poly_size, root, a, b, c);
  memcpy(psi, a.data(), a.size() * 8);
  memcpy(odd_fix, b.data(), b.size() * 8);
  return FHES_OK;
}

int fhes_debug_fft_tables(const fhes_params *p, double *twist, double *tw) {
  if (!p || !twist || !tw) return fail(FHES_ERR_INVALID, "null argument");
  std::vector<double> a, b;
  host_fft_tables(p->poly_size, a, b);
  memcpy(twist, a.data(), a.size() * 8);
  memcpy(tw, b.data(), b.size() * 8);
  return FHES_OK;
}

int fhes_set_keyswitch_mfma_min(fhes_ctx *ctx, int32_t min_count) {
  if (!ctx || min_count < 0) return fail(FHES_ERR_INVALID, "bad argument");
  std::lock_guard<std::mutex> lk(ctx->mu);
  ctx->ks_mfma_min = min_count;
  return FHES_OK;
}

static int debug_keyswitch_impl(fhes_ctx *ctx, int32_t count, const int32_t *in_handles, uint32_t *a_tilde, uint64_t *words);

int fhes_debug_keyswitch(fhes_ctx *ctx, int32_t count, const int32_t *in_handles, uint32_t *a_tilde) {
  if (!a_tilde) return fail(FHES_ERR_INVALID, "bad argument");
  return debug_keyswitch_impl(ctx, count, in_handles, a_tilde, nullptr);
}

int fhes_debug_keyswitch_words(fhes_ctx *ctx, int32_t count, const int32_t *in_handles, uint64_t *words) {
  if (!words) return fail(FHES_ERR_INVALID, "bad argument");
  return debug_keyswitch_impl(ctx, count, in_handles, nullptr, words);
}

static int debug_keyswitch_impl(fhes_ctx *ctx, int32_t count, const int32_t *in_handles, uint32_t *a_tilde, uint64_t *words) {
  if (!ctx || !in_handles || count <= 0) return fail(FHES_ERR_INVALID, "bad argument");
  MainLaneHold hold(ctx);
  FHES_HIP(hipSetDevice(ctx->device));
  if (!ctx->keys_loaded) return fail(FHES_ERR_STATE, "server keys not uploaded");
  std::vector<int32_t> tp(count + 1), ti(count), lut(count, 0), out(in_handles, in_handles + count);
  std::vector<int64_t> tc(count, 1);
  std::vector<uint64_t> cs(count, 0);
  for (int i = 0; i <= count; i++) tp[i] = i;
  for (int i = 0; i < count; i++) ti[i] = in_handles[i];
  DevNodeArrays a;
  int rc = stage_desc(ctx, count, tp.data(), ti.data(), tc.data(), cs.data(), nullptr, out.data(), &a);
  if (rc) return rc;
  rc = ensure_ks_out(ctx, ctx->main, count);
  if (rc) return rc;
  dev::NodeDesc d = make_desc(a, 0, count, 0, count);
  rc = launch_keyswitch(ctx, ctx->main, d);
  if (rc) return rc;
  std::vector<uint64_t> h((size_t)count * ctx->ks_stride);
  FHES_HIP(hipStreamSynchronize(ctx->stream));
  FHES_HIP(hipMemcpy(h.data(), ctx->main.ks_out, h.size() * 8, hipMemcpyDeviceToHost));
  int lg = 0;
  while ((1 << lg) < 2 * ctx->dims.N) lg++;
  if (words)
    for (int c = 0; c < count; c++)
      memcpy(words + (size_t)c * (ctx->dims.n + 1), &h[(size_t)c * ctx->ks_stride], sizeof(uint64_t) * (ctx->dims.n + 1));
  if (a_tilde)
  for (int c = 0; c < count; c++)
    for (int j = 0; j <= ctx->dims.n; j++)
      a_tilde[(size_t)c * (ctx->dims.n + 1) + j] =
          (uint32_t)((((h[(size_t)c * ctx->ks_stride + j] >> (64 - lg - 1)) + 1) >> 1) & (uint64_t)(2 * ctx->dims.N - 1));
  return FHES_OK;
}

}  // extern "C"
