// program.hip -- Level 1 of the C ABI: compiled circuits (host-side compile / export) and
// their execution on the device engine, one keyswitch + blind-rotate launch pair per level.
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <map>
#include <memory>
#include <thread>
#include <mutex>
#include <stdexcept>

#include "circuit/strings.h"
#include "engine.h"

using namespace fhes;

struct DeviceImage {
  int device = 0;
  unsigned char *buf = nullptr;  // all descriptor arrays (immutable after creation; the slot map of a run lives in its lane)
  DevNodeArrays arrays{};
};

struct fhes_program {
  circ::Program prog;
  // flattened descriptor arrays (slot numbering)
  std::vector<int32_t> term_ptr, term_idx, lut_ids, out_idx;
  std::vector<int64_t> term_coef;
  std::vector<uint64_t> consts;
  int32_t n_slots = 0, n_pbs_levels = 0;
  bool has_linear_level = false;
  // device-side placement of node results: node j lives in scratch cell scratch_idx[j] of its
  // instance (cells are re-used once every reader of the previous occupant ran in an earlier level)
  std::vector<int32_t> scratch_idx;
  int32_t n_scratch = 0;
  std::mutex mu;
  std::map<uint64_t, DeviceImage> images;  // keyed by fhes_ctx::uid
};

static void flatten(fhes_program *fp) {
  const circ::Program &p = fp->prog;
  const Dims d(p.params);
  fp->term_ptr.push_back(0);
  for (size_t j = 0; j < p.nodes.size(); j++) {
    const circ::Node &nd = p.nodes[j];
    for (const auto &t : nd.terms) {
      fp->term_idx.push_back(t.slot);
      fp->term_coef.push_back(t.coef);
    }
    fp->term_ptr.push_back((int32_t)fp->term_idx.size());
    fp->consts.push_back((uint64_t)nd.cst * d.delta);  // wraps mod 2^64 == mod 2*space*delta
    fp->lut_ids.push_back(nd.lut);
    fp->out_idx.push_back(p.n_inputs + (int32_t)j);
  }
  fp->n_slots = p.n_inputs + (int32_t)p.nodes.size();
  const int nl = (int)p.level_ptr.size() - 1;
  fp->has_linear_level = nl > 0 && !p.nodes.empty() && p.nodes.back().lut < 0;
  fp->n_pbs_levels = fp->has_linear_level ? nl - 1 : nl;

  // ---- liveness: last level that reads each node, then first-fit re-use of scratch cells ----
  const int32_t n_nodes = (int32_t)p.nodes.size();
  std::vector<int32_t> level_of(n_nodes, 0), last_use(n_nodes, -1);
  for (int lv = 0; lv < nl; lv++)
    for (int32_t j = p.level_ptr[lv]; j < p.level_ptr[lv + 1]; j++) level_of[j] = lv;
  for (int32_t j = 0; j < n_nodes; j++)
    for (const auto &t : p.nodes[j].terms)
      if (t.slot >= p.n_inputs) last_use[t.slot - p.n_inputs] = std::max(last_use[t.slot - p.n_inputs], level_of[j]);
  for (int32_t s : p.out_slots)
    if (s >= p.n_inputs) last_use[s - p.n_inputs] = INT32_MAX;  // results stay
  fp->scratch_idx.assign(n_nodes, 0);
  std::vector<int32_t> free_cells;
  std::vector<std::vector<int32_t>> expiring(nl + 1);  // expiring[lv]: cells whose occupant was last read in level lv
  int32_t next_cell = 0;
  for (int lv = 0; lv < nl; lv++) {
    // a cell read for the last time in level lv-1 (or earlier) may be written in level lv: the
    // reads of a level (keyswitch / linear kernel) are queued before the writes of the NEXT level
    if (lv > 0)
      for (int32_t c : expiring[lv - 1]) free_cells.push_back(c);
    for (int32_t j = p.level_ptr[lv]; j < p.level_ptr[lv + 1]; j++) {
      int32_t c;
      if (!free_cells.empty()) { c = free_cells.back(); free_cells.pop_back(); }
      else c = next_cell++;
      fp->scratch_idx[j] = c;
      const int32_t lu = std::max(last_use[j], lv);  // a node nobody reads expires with its own level
      if (lu != INT32_MAX) expiring[std::min(lu, nl)].push_back(c);
    }
  }
  fp->n_scratch = next_cell;
}

extern "C" {

fhes_program *fhes_program_compile(const fhes_params *p, const char *op, const int32_t *shape,
                                   int32_t n_shape, int32_t num_blocks_usize) {
  std::string why;
  if (!p || !op || (n_shape > 0 && !shape) || !params_valid(*p, &why)) {
    fail(FHES_ERR_INVALID, "program_compile: bad argument %s", why.c_str());
    return nullptr;
  }
  try {
    auto *fp = new fhes_program;
    fp->prog = circ::compile_op(*p, op, std::vector<int32_t>(shape, shape + n_shape), num_blocks_usize);
    flatten(fp);
    return fp;
  } catch (const std::exception &e) {
    fail(FHES_ERR_INVALID, "program_compile(%s): %s", op, e.what());
    return nullptr;
  }
}

void fhes_program_destroy(fhes_program *fp) {
  if (!fp) return;
  for (auto &kv : fp->images) {
    (void)hipSetDevice(kv.second.device);
    (void)hipFree(kv.second.buf);
  }
  delete fp;
}

int fhes_program_info(const fhes_program *fp, int64_t info[8]) {
  if (!fp || !info) return fail(FHES_ERR_INVALID, "null argument");
  info[0] = fp->prog.n_inputs;
  info[1] = (int64_t)fp->prog.out_slots.size();
  info[2] = (int64_t)fp->prog.nodes.size();
  info[3] = (int64_t)fp->prog.level_ptr.size() - 1;
  info[4] = (int64_t)fp->term_idx.size();
  info[5] = (int64_t)fp->prog.luts.size();
  info[6] = fp->n_slots;
  info[7] = fp->prog.n_pbs;
  return FHES_OK;
}

int32_t fhes_program_depth(const fhes_program *fp) {
  if (!fp) return fail(FHES_ERR_INVALID, "null argument");
  return fp->prog.depth;
}

int32_t fhes_program_scratch_cells(const fhes_program *fp) {
  if (!fp) return fail(FHES_ERR_INVALID, "null argument");
  return fp->n_scratch;
}

int32_t fhes_program_out_parts(const fhes_program *fp, int32_t *parts, int32_t max_parts) {
  if (!fp || (max_parts > 0 && !parts)) return fail(FHES_ERR_INVALID, "null argument");
  const int32_t n = (int32_t)fp->prog.out_parts.size();
  for (int32_t i = 0; i < n && i < max_parts; i++) parts[i] = fp->prog.out_parts[i];
  return n;
}

int fhes_program_out_meta(const fhes_program *fp, int32_t *degree, double *noise) {
  if (!fp || !degree || !noise) return fail(FHES_ERR_INVALID, "null argument");
  for (size_t i = 0; i < fp->prog.out_slots.size(); i++) {
    degree[i] = i < fp->prog.out_deg.size() ? (int32_t)fp->prog.out_deg[i] : -1;
    noise[i] = i < fp->prog.out_noise.size() ? fp->prog.out_noise[i] : -1.0;
  }
  return FHES_OK;
}

const char *fhes_op_signature(const char *op) { return op ? circ::op_signature(op) : nullptr; }

int fhes_program_export(const fhes_program *fp, int32_t *level_ptr, int32_t *term_ptr, int32_t *term_idx,
                        int64_t *term_coef, uint64_t *consts, int32_t *lut_ids, int32_t *out_idx,
                        uint64_t *lut_tables, int32_t *out_slots, uint64_t *out_trivial) {
  if (!fp || !level_ptr || !term_ptr || !consts || !lut_ids || !out_idx || !lut_tables || !out_slots)
    return fail(FHES_ERR_INVALID, "null argument");
  memcpy(level_ptr, fp->prog.level_ptr.data(), 4 * fp->prog.level_ptr.size());
  memcpy(term_ptr, fp->term_ptr.data(), 4 * fp->term_ptr.size());
  if (!fp->term_idx.empty()) {
    if (!term_idx || !term_coef) return fail(FHES_ERR_INVALID, "null term arrays");
    memcpy(term_idx, fp->term_idx.data(), 4 * fp->term_idx.size());
    memcpy(term_coef, fp->term_coef.data(), 8 * fp->term_coef.size());
  }
  memcpy(consts, fp->consts.data(), 8 * fp->consts.size());
  memcpy(lut_ids, fp->lut_ids.data(), 4 * fp->lut_ids.size());
  memcpy(out_idx, fp->out_idx.data(), 4 * fp->out_idx.size());
  const int space = 1 << (fp->prog.params.message_bits + fp->prog.params.carry_bits);  // entries per table (16 for the 2_2 sets)
  for (size_t i = 0; i < fp->prog.luts.size(); i++)
    for (int x = 0; x < space; x++) lut_tables[i * (size_t)space + x] = fp->prog.luts[i][x];
  for (size_t i = 0; i < fp->prog.out_slots.size(); i++) {
    out_slots[i] = fp->prog.out_slots[i];
    if (out_trivial) out_trivial[i] = 0;
  }
  return FHES_OK;
}

}  // extern "C"

// one run of `n_inst` instances on a lane of the context
static int run_impl(fhes_ctx *ctx, fhes_program *fp, int32_t n_inst, const int32_t *in_handles, int32_t *out_handles) {
  // Concurrency (round 3): the context mutex is held while the run is set up (arena ranges, program image, a
  // lane) and while it is cleaned up, NOT while its launches are queued; a caller that finds the main lane
  // carrying another caller's work gets a side lane (own stream and scratch) and returns when its results are
  // complete, so N threads calling into one context overlap on the device (engine.h, `Lane`).
  std::unique_lock<std::mutex> lk(ctx->mu);
  struct Active {  // calls inside a run: what tells a newcomer that it has company (the combiner below)
    fhes_ctx *c;
    explicit Active(fhes_ctx *cc) : c(cc) { c->active_runs++; }
    ~Active() { c->active_runs--; }
  } active(ctx);
  FHES_HIP(hipSetDevice(ctx->device));
  if (!ctx->keys_loaded) return fail(FHES_ERR_STATE, "server keys not uploaded");
  const circ::Program &p = fp->prog;
  const int32_t n_in = p.n_inputs, n_nodes = (int32_t)p.nodes.size(), n_out = (int32_t)p.out_slots.size();
  const int32_t n_slots = std::max(fp->n_slots, 1);
  if ((int64_t)n_inst * std::max(n_nodes, 1) > INT32_MAX / 2) return fail(FHES_ERR_INVALID, "program_run: batch too large");
  for (int64_t i = 0; i < (int64_t)n_in * n_inst; i++)
    if (in_handles[i] < 0 || in_handles[i] >= ctx->capacity) return fail(FHES_ERR_INVALID, "program_run: bad input handle");

  // ---- device image (once per ctx) ----
  std::unique_lock<std::mutex> plk(fp->mu);
  auto it = fp->images.find(ctx->uid);
  if (it == fp->images.end()) {
    DeviceImage img;
    img.device = ctx->device;
    std::vector<int32_t> luts(fp->lut_ids.size());
    std::vector<int32_t> local2ctx(p.luts.size());
    for (size_t i = 0; i < p.luts.size(); i++) {
      uint64_t t[circ::kMaxSpace];
      for (int x = 0; x < circ::kMaxSpace; x++) t[x] = p.luts[i][x];
      const int32_t id = lut_register_locked(ctx, t, ctx->dims.space());
      if (id < 0) return id;
      local2ctx[i] = id;
    }
    for (size_t j = 0; j < luts.size(); j++) luts[j] = fp->lut_ids[j] < 0 ? -1 : local2ctx[fp->lut_ids[j]];
    auto al = [](size_t v) { return (v + 15) & ~(size_t)15; };
    const size_t nt = std::max<size_t>(fp->term_idx.size(), 1), nn = std::max<size_t>(n_nodes, 1);
    const size_t o_tp = 0, o_ti = al(o_tp + 4 * (nn + 1)), o_tc = al(o_ti + 4 * nt), o_c = al(o_tc + 8 * nt),
                 o_l = al(o_c + 8 * nn), o_o = al(o_l + 4 * nn), total = al(o_o + 4 * nn);
    std::vector<unsigned char> h(total, 0);
    memcpy(&h[o_tp], fp->term_ptr.data(), 4 * fp->term_ptr.size());
    if (!fp->term_idx.empty()) {
      memcpy(&h[o_ti], fp->term_idx.data(), 4 * fp->term_idx.size());
      memcpy(&h[o_tc], fp->term_coef.data(), 8 * fp->term_coef.size());
    }
    if (n_nodes) {
      memcpy(&h[o_c], fp->consts.data(), 8 * (size_t)n_nodes);
      memcpy(&h[o_l], luts.data(), 4 * (size_t)n_nodes);
      memcpy(&h[o_o], fp->out_idx.data(), 4 * (size_t)n_nodes);
    }
    FHES_HIP(hipMalloc(&img.buf, total));
    FHES_HIP(hipMemcpy(img.buf, h.data(), total, hipMemcpyHostToDevice));
    img.arrays.term_ptr = (const int32_t *)(img.buf + o_tp);
    img.arrays.term_idx = (const int32_t *)(img.buf + o_ti);
    img.arrays.term_coef = (const int64_t *)(img.buf + o_tc);
    img.arrays.consts = (const uint64_t *)(img.buf + o_c);
    img.arrays.lut_ids = (const int32_t *)(img.buf + o_l);
    img.arrays.out_idx = (const int32_t *)(img.buf + o_o);
    img.arrays.n_slots = n_slots;
    it = fp->images.emplace(ctx->uid, img).first;
  }
  DevNodeArrays arrays = it->second.arrays;  // by value: the slot map is this run's
  plk.unlock();

  // ---- lane ----
  struct LaneGuard {  // destroyed AFTER the arena guard below (declared first): both run with ctx->mu held
    fhes_ctx *ctx;
    Lane *lane = nullptr;
    ~LaneGuard() { release_lane_locked(ctx, lane); }
  } lane_guard{ctx};
  {
    const int lrc = acquire_lane_locked(ctx, lk, &lane_guard.lane);
    if (lrc) return lrc;
  }
  Lane &lane = *lane_guard.lane;
  {
    const int mrc = ensure_slot_map(lane, (size_t)n_inst * n_slots);
    if (mrc) return mrc;
  }
  arrays.slot_map = lane.slot_map;

  // ---- slot maps for this run ----
  // Everything this call takes from the arena is handed back on EVERY failing exit (the outputs too:
  // the caller only owns them when the call succeeds); a program without outputs allocates none.
  struct ArenaGuard {
    fhes_ctx *ctx;
    int32_t out_base = -1, out_total = 0, scratch = -1, scratch_total = 0;
    bool keep_outputs = false;
    ~ArenaGuard() {
      if (scratch >= 0) arena_free_locked(ctx, scratch, scratch_total);  // stream-ordered: later work queues behind the reads
      if (out_base >= 0 && !keep_outputs) arena_free_locked(ctx, out_base, out_total);
    }
  } guard{ctx};
  const int32_t out_total = n_out * n_inst;
  int32_t out_base = 0;
  if (out_total > 0) {
    out_base = arena_alloc_locked(ctx, out_total);
    if (out_base < 0) return out_base;
    guard.out_base = out_base;
    guard.out_total = out_total;
  }
  int32_t scratch = -1;
  if (n_nodes > 0 && fp->n_scratch > 0) {
    scratch = arena_alloc_locked(ctx, fp->n_scratch * n_inst);
    if (scratch < 0) return scratch;
    guard.scratch = scratch;
    guard.scratch_total = fp->n_scratch * n_inst;
  }
  std::vector<int32_t> map((size_t)n_inst * n_slots);
  std::vector<std::pair<int32_t, int32_t>> copies;  // (dst handle, src handle index into map)
  std::vector<char> taken(n_slots);
  for (int32_t inst = 0; inst < n_inst; inst++) {
    int32_t *m = &map[(size_t)inst * n_slots];
    for (int32_t i = 0; i < n_in; i++) m[i] = in_handles[(size_t)inst * n_in + i];
    for (int32_t j = 0; j < n_nodes; j++) m[n_in + j] = scratch + inst * fp->n_scratch + fp->scratch_idx[j];
    std::fill(taken.begin(), taken.end(), 0);
    for (int32_t k = 0; k < n_out; k++) {
      const int32_t s = p.out_slots[k];
      const int32_t h = out_base + inst * n_out + k;
      out_handles[(size_t)inst * n_out + k] = h;
      if (s >= n_in && !taken[s]) {
        m[s] = h;  // the node writes its result straight into the output handle
        taken[s] = 1;
      } else {
        copies.push_back({h, inst * n_slots + s});
      }
    }
  }
  // ---- a side lane: which returned-but-unfinished main-stream work does this run depend on? ----
  std::vector<hipEvent_t> waits;
  if (!lane.is_main) {
    main_pending_prune_locked(ctx);
    for (const auto &mp : ctx->main_pending) {
      bool hit = false;
      for (int k = 0; k < 2 && !hit; k++) {
        if (mp.lo[k] >= mp.hi[k]) continue;
        auto overlaps = [&](int64_t lo, int64_t hi) { return lo < hi && lo < mp.hi[k] && mp.lo[k] < hi; };
        hit = overlaps(out_base, (int64_t)out_base + out_total) || (scratch >= 0 && overlaps(scratch, (int64_t)scratch + (int64_t)fp->n_scratch * n_inst));
        for (int64_t i = 0; i < (int64_t)n_in * n_inst && !hit; i++) hit = in_handles[i] >= mp.lo[k] && in_handles[i] < mp.hi[k];
      }
      if (hit) waits.push_back(mp.ev);
    }
  }

  // ---- queue the launches: without the context mutex, with the arena pinned (arena_rw shared) ----
  RunCounters cnt;
  int rc = FHES_OK;
  {
    std::shared_lock<std::shared_mutex> pin(ctx->arena_rw);
    lk.unlock();
    auto queue = [&]() -> int {
      // inputs that are results of a RETURNED main-lane call still running, rows that such a call handed back while
      // its kernels still use them: behind exactly those (fhes_ctx::main_pending)
      for (hipEvent_t ev : waits) FHES_HIP(hipStreamWaitEvent(lane.stream, ev, 0));
      FHES_HIP(hipMemcpyAsync(lane.slot_map, map.data(), 4 * map.size(), hipMemcpyHostToDevice, lane.stream));
      FHES_HIP(hipStreamSynchronize(lane.stream));  // `map` is a stack object
      int r = FHES_OK;
      // per-level trace, the engine's form of the reference's `log::trace!` per loop index (search.rs:37, mod.rs:342,
      // split.rs:277; enabled there by RUST_LOG=trace): FHES_LOG=trace prints one line per level as it is queued
      static const bool trace = [] { const char *e = getenv("FHES_LOG"); return e && std::string(e) == "trace"; }();
      for (int lv = 0; lv < fp->n_pbs_levels && r == FHES_OK; lv++) {
        if (trace)
          fprintf(stderr, "[fhes trace] %s: level %d/%d, %d bootstraps x %d instance(s), lane %s\n", p.name.c_str(), lv + 1,
                  fp->n_pbs_levels, p.level_ptr[lv + 1] - p.level_ptr[lv], n_inst, lane.is_main ? "main" : "side");
        r = launch_pbs(ctx, lane, arrays, p.level_ptr[lv], p.level_ptr[lv + 1] - p.level_ptr[lv], n_inst, &cnt);
      }
      if (r == FHES_OK && fp->has_linear_level) {
        const int lv = fp->n_pbs_levels;
        r = launch_linear(ctx, lane, arrays, p.level_ptr[lv], p.level_ptr[lv + 1] - p.level_ptr[lv], n_inst, &cnt);
      }
      if (r == FHES_OK) {
        const size_t bytes = (size_t)ctx->dims.ct_words() * 8;
        for (auto &c : copies)
          FHES_HIP(hipMemcpyAsync(ctx->arena + (size_t)c.first * ctx->ct_stride,
                                  ctx->arena + (size_t)map[c.second] * ctx->ct_stride, bytes,
                                  hipMemcpyDeviceToDevice, lane.stream));
      }
      return r;
    };
    rc = queue();
  }
  // a side lane's results must be complete before anybody can hold their handles: the main stream does not
  // know about this lane (making it wait would serialise the lanes behind each other)
  if (!lane.is_main) {
    hipError_t e = hipStreamSynchronize(lane.stream);
    if (e != hipSuccess && rc == FHES_OK) rc = hip_fail(e, "side lane");
  }
  lk.lock();  // the guards below free arena ranges and hand the lane back under the mutex
  if (lane.is_main) {
    // this call returns with its work queued: remember what it writes, and keep its scratch out of the free list until
    // the kernels that use it have finished (another lane would be handed the same rows otherwise)
    const bool defer = scratch >= 0 && rc == FHES_OK;
    if (main_pending_push_locked(ctx, out_total > 0 ? out_base : 0, out_total > 0 ? out_base + out_total : 0, scratch >= 0 ? scratch : 0,
                                 scratch >= 0 ? scratch + fp->n_scratch * n_inst : 0, defer) == FHES_OK && defer)
      guard.scratch = -1;  // freed by main_pending_prune_locked
  }
  merge_counters_locked(ctx, cnt);
  guard.keep_outputs = rc == FHES_OK;
  return rc;
}

extern "C" {

// Combining (round 3): single-instance runs of the SAME program that arrive while other runs are in flight are merged
// into ONE batched run -- the first such caller opens a short window (FHES_COMBINE_WINDOW_US, default 300), later ones
// join it, the opener runs them as `n` instances (every level one launch pair for all of them instead of `n` small ones)
// and hands each caller its own output handles.  A caller that finds the context idle runs at once, so a single-threaded
// application never waits.  Results are the same bits either way (instances are independent).  FHES_COMBINE=0 disables.
namespace {
struct Forming {
  std::vector<const int32_t *> ins;
  std::vector<int32_t *> outs;
  int state = 0;  // 0 open, 1 running, 2 done
  int rc = FHES_OK;
  std::string err;
  std::condition_variable cv;
};
std::mutex g_forming_mu;  // guards the table below (its entries are per context AND program)
std::map<std::pair<uint64_t, const fhes_program *>, std::shared_ptr<Forming>> g_forming;
constexpr size_t kMaxCombine = 64;
}  // namespace

int fhes_program_run_batch(fhes_ctx *ctx, const fhes_program *cfp, int32_t n_inst,
                           const int32_t *in_handles, int32_t *out_handles) {
  auto *fp = const_cast<fhes_program *>(cfp);
  if (!ctx || !fp || !out_handles || n_inst < 1 || (fp->prog.n_inputs > 0 && !in_handles))
    return fail(FHES_ERR_INVALID, "program_run: bad argument");
  if (memcmp(&ctx->p, &fp->prog.params, sizeof(fhes_params)) != 0)
    return fail(FHES_ERR_INVALID, "program_run: program was compiled for different parameters");
  static const bool combine = [] { const char *e = getenv("FHES_COMBINE"); return !(e && std::string(e) == "0"); }();
  static const int window_us = [] { const char *e = getenv("FHES_COMBINE_WINDOW_US"); const int v = e ? atoi(e) : 300; return v < 0 ? 0 : v; }();
  if (!combine || n_inst != 1) return run_impl(ctx, fp, n_inst, in_handles, out_handles);
  const auto key = std::make_pair(ctx->uid, (const fhes_program *)fp);
  std::shared_ptr<Forming> f;
  bool leader = false;
  {
    std::unique_lock<std::mutex> fl(g_forming_mu);
    auto it = g_forming.find(key);
    if (it != g_forming.end() && it->second->state == 0 && it->second->ins.size() < kMaxCombine) {
      f = it->second;  // join the open window
      f->ins.push_back(in_handles);
      f->outs.push_back(out_handles);
      f->cv.wait(fl, [&] { return f->state == 2; });
      if (f->rc != FHES_OK) return fail(f->rc, "%s", f->err.c_str());
      return FHES_OK;
    }
    bool company;
    {
      std::lock_guard<std::mutex> lk(ctx->mu);
      company = ctx->active_runs > 0;
    }
    if (company && it == g_forming.end()) {  // others are running: open a window for whoever comes next
      f = std::make_shared<Forming>();
      f->ins.push_back(in_handles);
      f->outs.push_back(out_handles);
      g_forming[key] = f;
      leader = true;
    }
  }
  if (!leader) return run_impl(ctx, fp, 1, in_handles, out_handles);
  if (window_us > 0) std::this_thread::sleep_for(std::chrono::microseconds(window_us));
  size_t n;
  {
    std::lock_guard<std::mutex> fl(g_forming_mu);
    f->state = 1;  // closed: later arrivals open their own
    g_forming.erase(key);
    n = f->ins.size();
  }
  int rc;
  if (n == 1) {
    rc = run_impl(ctx, fp, 1, in_handles, out_handles);
  } else {
    const size_t n_in = (size_t)fp->prog.n_inputs, n_out = fp->prog.out_slots.size();
    std::vector<int32_t> all_in(n * n_in), all_out(n * std::max<size_t>(n_out, 1));
    for (size_t i = 0; i < n; i++)
      if (n_in) memcpy(&all_in[i * n_in], f->ins[i], 4 * n_in);
    rc = run_impl(ctx, fp, (int32_t)n, all_in.data(), all_out.data());
    if (rc == FHES_OK)
      for (size_t i = 0; i < n; i++)
        if (n_out) memcpy(f->outs[i], &all_out[i * n_out], 4 * n_out);
  }
  {
    std::lock_guard<std::mutex> fl(g_forming_mu);
    f->rc = rc;
    if (rc != FHES_OK) f->err = fhes_last_error();
    f->state = 2;
  }
  f->cv.notify_all();
  return rc;
}

int fhes_program_run(fhes_ctx *ctx, const fhes_program *fp, const int32_t *in_handles, int32_t *out_handles) {
  return fhes_program_run_batch(ctx, fp, 1, in_handles, out_handles);
}

}  // extern "C"
