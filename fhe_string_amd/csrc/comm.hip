// comm.hip -- multi-GPU level of the C ABI: one process per GPU, position-sharded search.
// the positions of find_all (/root/reference/src/ciphertext/search.rs:33-41) are
// independent; the serial scans are not.  Rank g runs substr_eq for its block of positions, one
// all-gather of the per-position BooleanBlocks, replicated reduction.  Transports: RCCL on device
// pointers (resolved with dlopen: no link-time dependency) or caller-supplied host callbacks.
#include <dlfcn.h>
#include <sys/stat.h>
#include <rccl/rccl.h>  // types and prototypes only; the functions are looked up at run time

#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "engine.h"

using namespace fhes;

namespace {

struct RcclApi {
  void *lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  std::string error;
  std::string rccl_path, hip_path;  // the files the bound RCCL and ITS HIP runtime were loaded from
};

// The file a code address lives in (dladdr), as (device, inode) + path: "which HIP runtime is this?"
struct LibId {
  std::string path;
  dev_t dev = 0;
  ino_t ino = 0;
  bool ok = false;
  bool same(const LibId &o) const { return ok && o.ok && dev == o.dev && ino == o.ino; }
};
LibId lib_of(const void *addr) {
  LibId id;
  Dl_info info;
  if (!addr || !dladdr(addr, &info) || !info.dli_fname) return id;
  id.path = info.dli_fname;
  struct stat st;
  if (stat(info.dli_fname, &st) == 0) { id.dev = st.st_dev; id.ino = st.st_ino; id.ok = true; }
  return id;
}

// RCCL has to drive the SAME HIP runtime instance this library allocates its arena with.  A process can hold two of them --
// PyTorch wheels bundle libamdhip64 / librccl, and when this library is loaded BEFORE torch the loader keeps the system
// runtime for us and adds torch's next to it; `dlopen("librccl.so.1")` then hands back torch's RCCL (same soname), bound to a
// runtime that has never seen our context: ncclCommInitRank answers "unhandled cuda error / no ROCm-capable device"
// (gpurun_out/r03_flaky.txt).  So: take a candidate, ask IT for hipGetDeviceCount (dlsym on the handle walks its own
// dependencies), compare that runtime's file with the one our own calls resolve to, and fall back to the librccl that sits
// next to OUR runtime (loaded by absolute path: a second copy with its own bindings).
RcclApi api;   // bound once, by the first caller of rccl()

RcclApi *rccl() {
  static std::once_flag once;
  std::call_once(once, [] {
    const LibId ours = lib_of(reinterpret_cast<const void *>(&hipGetDeviceCount));
    std::vector<std::string> names = {"librccl.so.1", "librccl.so"};
    if (ours.ok) {
      const std::string dir = ours.path.substr(0, ours.path.find_last_of('/') + 1);
      names.push_back(dir + "librccl.so.1");
      names.push_back(dir + "librccl.so");
    }
    names.push_back("/opt/rocm/lib/librccl.so.1");
    // FHES_RCCL_LIB names the one library to try (a site with RCCL elsewhere; tests/test_abi.py points it at nothing)
    if (const char *only = getenv("FHES_RCCL_LIB")) names.assign(1, std::string(only));
    std::string seen, load_error;
    for (const std::string &name : names) {
      void *h = dlopen(name.c_str(), RTLD_NOW | RTLD_LOCAL);
      if (!h) {
        const char *e = dlerror();   // ONE call: glibc hands the message out once and clears it
        if (load_error.empty()) load_error = e ? e : (name + ": not found");
        continue;
      }
      const LibId theirs = lib_of(dlsym(h, "hipGetDeviceCount"));
      if (!ours.ok || !theirs.ok || ours.same(theirs)) {   // (a runtime we cannot identify is taken on trust, as before)
        api.lib = h;
        api.rccl_path = lib_of(dlsym(h, "ncclCommInitRank")).path;
        api.hip_path = theirs.path;
        break;
      }
      seen += (seen.empty() ? "" : "; ") + lib_of(dlsym(h, "ncclCommInitRank")).path + " -> " + theirs.path;
      dlclose(h);
    }
    if (!api.lib) {
      api.error = seen.empty() ? "cannot load librccl: " + (load_error.empty() ? std::string("not found") : load_error)
                               : "every librccl in reach is bound to another HIP runtime than this library (" + ours.path +
                                     "): " + seen + " -- two ROCm runtimes in one process (e.g. PyTorch's bundled one next to /opt/rocm); "
                                     "load this library after the framework that brings its own runtime, or make one runtime visible";
      return;
    }
#define FHES_SYM(field, sym)                                                        \
  api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.lib, sym));          \
  if (!api.field) { api.error = std::string("librccl lacks ") + sym; return; }
    FHES_SYM(GetUniqueId, "ncclGetUniqueId")
    FHES_SYM(CommInitRank, "ncclCommInitRank")
    FHES_SYM(CommDestroy, "ncclCommDestroy")
    FHES_SYM(AllGather, "ncclAllGather")
    FHES_SYM(Broadcast, "ncclBroadcast")
    FHES_SYM(AllReduce, "ncclAllReduce")
    FHES_SYM(GetErrorString, "ncclGetErrorString")
#undef FHES_SYM
  });
  return api.error.empty() ? &api : nullptr;
}

std::string rccl_error() {
  rccl();
  return api.error;
}

int rccl_fail(ncclResult_t r, const char *what) {
  RcclApi *a = rccl();
  return fail(FHES_ERR_NO_DEVICE, "RCCL error %d (%s) in %s [librccl %s on HIP runtime %s]", (int)r, a ? a->GetErrorString(r) : "?", what,
              a ? a->rccl_path.c_str() : "?", a ? a->hip_path.c_str() : "?");
}

}  // namespace

struct fhes_comm {
  fhes_ctx *ctx = nullptr;
  fhes_params params{};            // of ctx, or given with the executor hook
  bool has_exec = false;           // host-memory executor hook instead of a device context (fhes_comm_create_host_exec)
  fhes_executor exec{};
  int32_t rank = 0, world = 1;
  ncclComm_t nccl = nullptr;  // RCCL transport
  fhes_allgather_fn ag = nullptr;  // host transport
  fhes_bcast_fn bc = nullptr;
  void *user = nullptr;
  std::mutex mu;
  std::map<std::string, fhes_program *> progs;
};

#define FHES_NCCL(call)                                        \
  do {                                                         \
    ncclResult_t _r = (call);                                  \
    if (_r != ncclSuccess) return rccl_fail(_r, #call);        \
  } while (0)

// ---- transport primitives on device memory (count bytes; multiples of 8) -------------------------
static int dev_all_gather(fhes_comm *c, void *base, size_t bytes_per_rank) {
  fhes_ctx *ctx = c->ctx;
  unsigned char *b = static_cast<unsigned char *>(base);
  if (c->world == 1) return FHES_OK;
  if (c->nccl) {
    FHES_NCCL(rccl()->AllGather(b + (size_t)c->rank * bytes_per_rank, b, bytes_per_rank / 8, ncclUint64, c->nccl, ctx->stream));
    return FHES_OK;
  }
  std::vector<unsigned char> send(bytes_per_rank), recv(bytes_per_rank * (size_t)c->world);
  FHES_HIP(hipStreamSynchronize(ctx->stream));
  FHES_HIP(hipMemcpy(send.data(), b + (size_t)c->rank * bytes_per_rank, bytes_per_rank, hipMemcpyDeviceToHost));
  if (c->ag(c->user, send.data(), recv.data(), (int64_t)bytes_per_rank) != 0) return fail(FHES_ERR_STATE, "host all-gather callback failed");
  FHES_HIP(hipMemcpy(b, recv.data(), recv.size(), hipMemcpyHostToDevice));
  return FHES_OK;
}

static int dev_broadcast(fhes_comm *c, void *buf, size_t bytes, int root) {
  fhes_ctx *ctx = c->ctx;
  if (c->world == 1 || bytes == 0) return FHES_OK;
  if (c->nccl) {
    FHES_NCCL(rccl()->Broadcast(buf, buf, bytes / 8, ncclUint64, root, c->nccl, ctx->stream));
    return FHES_OK;
  }
  std::vector<unsigned char> h(bytes);
  FHES_HIP(hipStreamSynchronize(ctx->stream));
  if (c->rank == root) FHES_HIP(hipMemcpy(h.data(), buf, bytes, hipMemcpyDeviceToHost));
  if (c->bc(c->user, h.data(), (int64_t)bytes, root) != 0) return fail(FHES_ERR_STATE, "host broadcast callback failed");
  if (c->rank != root) FHES_HIP(hipMemcpy(buf, h.data(), bytes, hipMemcpyHostToDevice));
  return FHES_OK;
}

// in-place sum (wrapping u64 = LWE addition) of `bytes` bytes over all ranks
static int dev_all_reduce_sum(fhes_comm *c, void *buf, size_t bytes) {
  fhes_ctx *ctx = c->ctx;
  if (c->world == 1 || bytes == 0) return FHES_OK;
  if (c->nccl) {
    FHES_NCCL(rccl()->AllReduce(buf, buf, bytes / 8, ncclUint64, ncclSum, c->nccl, ctx->stream));
    return FHES_OK;
  }
  std::vector<uint64_t> send(bytes / 8), recv(bytes / 8 * (size_t)c->world);
  FHES_HIP(hipStreamSynchronize(ctx->stream));
  FHES_HIP(hipMemcpy(send.data(), buf, bytes, hipMemcpyDeviceToHost));
  if (c->ag(c->user, send.data(), recv.data(), (int64_t)bytes) != 0) return fail(FHES_ERR_STATE, "host all-gather callback failed");
  for (size_t w = 0; w < send.size(); w++) {
    uint64_t acc = 0;
    for (int r = 0; r < c->world; r++) acc += recv[(size_t)r * send.size() + w];
    send[w] = acc;
  }
  FHES_HIP(hipMemcpy(buf, send.data(), bytes, hipMemcpyHostToDevice));
  return FHES_OK;
}

// ---- the planner's view of "ciphertext rows and programs": the device context, or a caller's executor ----
// (fhes_comm_create_host_exec: the CPU-only tests run THIS planner with the oracle as the executor, over gloo)
namespace {
struct X {
  fhes_comm *c;
  int32_t alloc(int32_t n) { return c->has_exec ? c->exec.alloc(c->exec.user, n) : fhes_arena_alloc(c->ctx, n); }
  void release(int32_t first, int32_t n) {
    if (first < 0 || n <= 0) return;
    if (c->has_exec) (void)c->exec.release(c->exec.user, first, n);
    else (void)fhes_arena_free(c->ctx, first, n);
  }
  int run(const fhes_program *p, const int32_t *in, int32_t *out) {
    if (!c->has_exec) return fhes_program_run(c->ctx, p, in, out);
    const int rc = c->exec.run(c->exec.user, p, in, out);
    return rc == 0 ? FHES_OK : fail(FHES_ERR_STATE, "executor hook failed (%d)", rc);
  }
  size_t row_bytes() const { return c->has_exec ? (size_t)c->exec.row_words * 8 : (size_t)c->ctx->ct_stride * 8; }
  int copy_rows(int32_t dst, int32_t src, int32_t n) {
    if (n <= 0) return FHES_OK;
    if (c->has_exec) { memmove(c->exec.rows(c->exec.user, dst), c->exec.rows(c->exec.user, src), (size_t)n * row_bytes()); return FHES_OK; }
    fhes_ctx *ctx = c->ctx;
    std::lock_guard<std::mutex> lk(ctx->mu);
    FHES_HIP(hipSetDevice(ctx->device));
    FHES_HIP(hipMemcpyAsync(ctx->arena + (size_t)dst * ctx->ct_stride, ctx->arena + (size_t)src * ctx->ct_stride,
                            (size_t)n * row_bytes(), hipMemcpyDeviceToDevice, ctx->stream));
    return main_pending_push_locked(ctx, dst, dst + n, 0, 0);  // queued on the main stream: other lanes order behind it
  }
  int zero_rows(int32_t first, int32_t n) {  // all-zero words = the trivial encryption of 0
    if (n <= 0) return FHES_OK;
    if (c->has_exec) { memset(c->exec.rows(c->exec.user, first), 0, (size_t)n * row_bytes()); return FHES_OK; }
    fhes_ctx *ctx = c->ctx;
    std::lock_guard<std::mutex> lk(ctx->mu);
    FHES_HIP(hipSetDevice(ctx->device));
    FHES_HIP(hipMemsetAsync(ctx->arena + (size_t)first * ctx->ct_stride, 0, (size_t)n * row_bytes(), ctx->stream));
    return main_pending_push_locked(ctx, first, first + n, 0, 0);
  }
  int all_gather_rows(int32_t first, int32_t per_rank) {  // in place: rank r owns rows [first + r * per_rank, ...)
    if (c->world == 1 || per_rank <= 0) return FHES_OK;
    if (!c->has_exec) return fhes_comm_all_gather_cts(c, first, per_rank);
    const size_t b = (size_t)per_rank * row_bytes();
    std::vector<unsigned char> recv(b * (size_t)c->world);
    if (c->ag(c->user, c->exec.rows(c->exec.user, first + c->rank * per_rank), recv.data(), (int64_t)b) != 0)
      return fail(FHES_ERR_STATE, "host all-gather callback failed");
    memcpy(c->exec.rows(c->exec.user, first), recv.data(), recv.size());
    return FHES_OK;
  }
  int all_reduce_rows(int32_t first, int32_t n) {  // in place: sum over the ranks, word by word
    if (c->world == 1 || n <= 0) return FHES_OK;
    if (!c->has_exec) {
      fhes_ctx *ctx = c->ctx;
      std::lock_guard<std::mutex> lk(ctx->mu);
      FHES_HIP(hipSetDevice(ctx->device));
      const int rc = dev_all_reduce_sum(c, ctx->arena + (size_t)first * ctx->ct_stride, (size_t)n * row_bytes());
      return rc ? rc : main_pending_push_locked(ctx, first, first + n, 0, 0);
    }
    const size_t words = (size_t)n * row_bytes() / 8;
    uint64_t *rows = c->exec.rows(c->exec.user, first);
    std::vector<uint64_t> recv(words * (size_t)c->world);
    if (c->ag(c->user, rows, recv.data(), (int64_t)(words * 8)) != 0) return fail(FHES_ERR_STATE, "host all-gather callback failed");
    for (size_t w = 0; w < words; w++) {
      uint64_t acc = 0;
      for (int r = 0; r < c->world; r++) acc += recv[(size_t)r * words + w];
      rows[w] = acc;
    }
    return FHES_OK;
  }
};
}  // namespace

extern "C" {

int fhes_comm_unique_id(uint8_t id[FHES_COMM_ID_BYTES]) {
  static_assert(sizeof(ncclUniqueId) == FHES_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
  if (!id) return fail(FHES_ERR_INVALID, "null argument");
  RcclApi *a = rccl();
  if (!a) return fail(FHES_ERR_STATE, "RCCL unavailable: %s", rccl_error().c_str());
  ncclUniqueId u;
  FHES_NCCL(a->GetUniqueId(&u));
  memcpy(id, &u, sizeof(u));
  return FHES_OK;
}

fhes_comm *fhes_comm_create_rccl(fhes_ctx *ctx, int32_t rank, int32_t world, const uint8_t id[FHES_COMM_ID_BYTES]) {
  if (!ctx || !id || world < 1 || rank < 0 || rank >= world) { fail(FHES_ERR_INVALID, "comm_create_rccl: bad argument"); return nullptr; }
  RcclApi *a = rccl();
  if (!a) { fail(FHES_ERR_STATE, "RCCL unavailable: %s", rccl_error().c_str()); return nullptr; }
  if (hipSetDevice(ctx->device) != hipSuccess) { fail(FHES_ERR_NO_DEVICE, "hipSetDevice failed"); return nullptr; }
  ncclUniqueId u;
  memcpy(&u, id, sizeof(u));
  auto *c = new fhes_comm;
  c->ctx = ctx; c->rank = rank; c->world = world; c->params = ctx->p;
  ncclResult_t r = a->CommInitRank(&c->nccl, world, u, rank);
  if (r != ncclSuccess) { rccl_fail(r, "ncclCommInitRank"); delete c; return nullptr; }
  return c;
}

// Which librccl / HIP runtime the RCCL transport is bound to ("" before the first use or when unavailable).
const char *fhes_rccl_library_path(void) { return rccl() ? api.rccl_path.c_str() : ""; }
const char *fhes_rccl_hip_runtime_path(void) { return rccl() ? api.hip_path.c_str() : ""; }

// One all-gather, one all-reduce and one broadcast of a few words on memory of THIS library's runtime, through whatever
// transport the communicator has -- with RCCL also at world == 1, where the data-path functions return early: it is the
// check that the bound librccl accepts our device pointers (the two-runtimes failure shows here, not at the first wide run).
int fhes_comm_self_test(fhes_comm *c) {
  if (!c || !c->ctx) return fail(FHES_ERR_INVALID, "comm_self_test: needs a device communicator");
  fhes_ctx *ctx = c->ctx;
  if (hipSetDevice(ctx->device) != hipSuccess) return fail(FHES_ERR_NO_DEVICE, "hipSetDevice failed");
  const int W = c->world, per = 64;
  std::vector<uint64_t> h((size_t)W * per, 0);
  for (int q = 0; q < per; q++) h[(size_t)c->rank * per + q] = 0x1000ull * (uint64_t)(c->rank + 1) + (uint64_t)q;
  uint64_t *d = nullptr;
  FHES_HIP(hipMalloc(reinterpret_cast<void **>(&d), h.size() * 8));
  int rc = FHES_OK;
  auto finish = [&](int code) { (void)hipFree(d); return code; };
  if (hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice) != hipSuccess) return finish(fail(FHES_ERR_NO_DEVICE, "self test: upload failed"));
  if (c->nccl) {
    RcclApi *a = rccl();
    ncclResult_t r = a->AllGather(d + (size_t)c->rank * per, d, per, ncclUint64, c->nccl, ctx->stream);
    if (r == ncclSuccess) r = a->AllReduce(d, d, per, ncclUint64, ncclSum, c->nccl, ctx->stream);   // first block: sum over ranks of rank 0's words
    if (r == ncclSuccess) r = a->Broadcast(d + per * (size_t)(W - 1), d + per * (size_t)(W - 1), per, ncclUint64, W - 1, c->nccl, ctx->stream);
    if (r != ncclSuccess) return finish(rccl_fail(r, "self test collectives"));
  } else {
    rc = dev_all_gather(c, d, (size_t)per * 8);
    if (rc == FHES_OK) rc = dev_all_reduce_sum(c, d, (size_t)per * 8);
    if (rc == FHES_OK) rc = dev_broadcast(c, d + per * (size_t)(W - 1), (size_t)per * 8, W - 1);
    if (rc != FHES_OK) return finish(rc);
  }
  if (hipStreamSynchronize(ctx->stream) != hipSuccess) return finish(fail(FHES_ERR_NO_DEVICE, "self test: the collectives faulted on the stream"));
  if (hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return finish(fail(FHES_ERR_NO_DEVICE, "self test: download failed"));
  for (int r = 0; r < W; r++)
    for (int q = 0; q < per; q++) {
      const uint64_t want = r == 0 ? (uint64_t)W * (0x1000ull + (uint64_t)q) : 0x1000ull * (uint64_t)(r + 1) + (uint64_t)q;
      if (h[(size_t)r * per + q] != want)
        return finish(fail(FHES_ERR_STATE, "self test: word %d of rank %d's block is %llx, expected %llx", q, r, (unsigned long long)h[(size_t)r * per + q],
                           (unsigned long long)want));
    }
  return finish(FHES_OK);
}

fhes_comm *fhes_comm_create_host(fhes_ctx *ctx, int32_t rank, int32_t world, fhes_allgather_fn ag, fhes_bcast_fn bc, void *user) {
  if (world < 1 || rank < 0 || rank >= world || (world > 1 && (!ag || !bc))) { fail(FHES_ERR_INVALID, "comm_create_host: bad argument"); return nullptr; }
  auto *c = new fhes_comm;
  c->ctx = ctx; c->rank = rank; c->world = world; c->ag = ag; c->bc = bc; c->user = user;
  if (ctx) c->params = ctx->p;
  return c;
}

fhes_comm *fhes_comm_create_host_exec(const fhes_params *p, int32_t rank, int32_t world, fhes_allgather_fn ag, fhes_bcast_fn bc,
                                      void *user, const fhes_executor *exec) {
  std::string why;
  if (!p || !params_valid(*p, &why) || !exec || !exec->alloc || !exec->release || !exec->rows || !exec->run ||
      exec->row_words < fhes_ct_words(p) || world < 1 || rank < 0 || rank >= world || (world > 1 && (!ag || !bc))) {
    fail(FHES_ERR_INVALID, "comm_create_host_exec: bad argument %s", why.c_str());
    return nullptr;
  }
  auto *c = new fhes_comm;
  c->rank = rank; c->world = world; c->ag = ag; c->bc = bc; c->user = user;
  c->params = *p;
  c->has_exec = true;
  c->exec = *exec;
  return c;
}

void fhes_comm_destroy(fhes_comm *c) {
  if (!c) return;
  for (auto &kv : c->progs) fhes_program_destroy(kv.second);
  if (c->nccl && rccl()) (void)rccl()->CommDestroy(c->nccl);
  delete c;
}

int32_t fhes_comm_rank(const fhes_comm *c) { return c ? c->rank : -1; }
int32_t fhes_comm_world(const fhes_comm *c) { return c ? c->world : -1; }
int32_t fhes_comm_is_rccl(const fhes_comm *c) { return c && c->nccl ? 1 : 0; }

int fhes_shard_range(int64_t n, int32_t world, int32_t rank, int64_t *lo, int64_t *hi) {
  if (n < 0 || world < 1 || rank < 0 || rank >= world || !lo || !hi) return fail(FHES_ERR_INVALID, "shard_range: bad argument");
  const int64_t c = (n + world - 1) / world;
  *lo = std::min<int64_t>((int64_t)rank * c, n);
  *hi = std::min<int64_t>(*lo + c, n);
  return FHES_OK;
}

int fhes_comm_all_gather_host(fhes_comm *c, const void *send, void *recv, int64_t bytes_per_rank) {
  if (!c || !send || !recv || bytes_per_rank < 0) return fail(FHES_ERR_INVALID, "all_gather_host: bad argument");
  if (bytes_per_rank == 0) return FHES_OK;
  if (c->world == 1) { memcpy(recv, send, (size_t)bytes_per_rank); return FHES_OK; }
  if (!c->nccl) {
    if (c->ag(c->user, send, recv, bytes_per_rank) != 0) return fail(FHES_ERR_STATE, "host all-gather callback failed");
    return FHES_OK;
  }
  if (bytes_per_rank % 8) return fail(FHES_ERR_INVALID, "all_gather_host over RCCL: size must be a multiple of 8");
  std::lock_guard<std::mutex> lk(c->ctx->mu);
  FHES_HIP(hipSetDevice(c->ctx->device));
  unsigned char *d = nullptr;
  FHES_HIP(hipMalloc(&d, (size_t)bytes_per_rank * c->world));
  hipError_t e = hipMemcpy(d + (size_t)c->rank * bytes_per_rank, send, (size_t)bytes_per_rank, hipMemcpyHostToDevice);
  int rc = e == hipSuccess ? dev_all_gather(c, d, (size_t)bytes_per_rank) : hip_fail(e, "staging copy");
  if (rc == FHES_OK) {
    e = hipStreamSynchronize(c->ctx->stream);
    if (e == hipSuccess) e = hipMemcpy(recv, d, (size_t)bytes_per_rank * c->world, hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = hip_fail(e, "staging copy");
  }
  (void)hipFree(d);
  return rc;
}

int fhes_comm_all_gather_cts(fhes_comm *c, int32_t first, int32_t count) {
  if (!c || !c->ctx || first < 0 || count <= 0) return fail(FHES_ERR_INVALID, "all_gather_cts: bad argument");
  fhes_ctx *ctx = c->ctx;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if ((int64_t)first + (int64_t)count * c->world > ctx->capacity) return fail(FHES_ERR_INVALID, "all_gather_cts: range outside the arena");
  FHES_HIP(hipSetDevice(ctx->device));
  const int rc = dev_all_gather(c, ctx->arena + (size_t)first * ctx->ct_stride, (size_t)count * ctx->ct_stride * 8);
  return rc ? rc : main_pending_push_locked(ctx, first, first + count * c->world, 0, 0);
}

int fhes_comm_broadcast_cts(fhes_comm *c, int32_t first, int32_t count, int32_t root) {
  if (!c || !c->ctx || first < 0 || count <= 0 || root < 0 || root >= c->world) return fail(FHES_ERR_INVALID, "broadcast_cts: bad argument");
  fhes_ctx *ctx = c->ctx;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if ((int64_t)first + count > ctx->capacity) return fail(FHES_ERR_INVALID, "broadcast_cts: range outside the arena");
  FHES_HIP(hipSetDevice(ctx->device));
  const int rc = dev_broadcast(c, ctx->arena + (size_t)first * ctx->ct_stride, (size_t)count * ctx->ct_stride * 8, root);
  return rc ? rc : main_pending_push_locked(ctx, first, first + count, 0, 0);
}

int fhes_comm_broadcast_keys(fhes_comm *c, int32_t root) {
  if (!c || !c->ctx || root < 0 || root >= c->world) return fail(FHES_ERR_INVALID, "broadcast_keys: bad argument");
  fhes_ctx *ctx = c->ctx;
  MainLaneHold hold(ctx);
  FHES_HIP(hipSetDevice(ctx->device));
  FHES_HIP(hipDeviceSynchronize());  // no lane may still read the keys that are replaced
  if (c->rank == root && !ctx->keys_loaded) return fail(FHES_ERR_STATE, "broadcast_keys: the root has no keys");
  const Dims &d = ctx->dims;
  const size_t ksk_b = (size_t)d.big * d.ks_level * ctx->ks_stride * 8, bias_b = (size_t)ctx->ks_stride * 8;
  const size_t bsk_b = (size_t)d.n_ggsw * d.L * (d.k + 1) * (d.k + 1) * d.M * 16;
  if (!ctx->ksk) FHES_HIP(hipMalloc(&ctx->ksk, ksk_b));
  if (!ctx->ksk_bias) FHES_HIP(hipMalloc(&ctx->ksk_bias, bias_b));
  if (!ctx->bsk_f) FHES_HIP(hipMalloc(&ctx->bsk_f, bsk_b));
  const bool has_e = ctx->bsk_e != nullptr || (d.k == 1 && d.L == 1 && (d.N == 2048 || d.N == 256));  // throughput-kernel layout
  if (has_e && !ctx->bsk_e) FHES_HIP(hipMalloc(&ctx->bsk_e, bsk_b));
  // the packed i8 key of the matrix-core keyswitch (the default path) travels too: a rank that kept a stale
  // one would keyswitch with the OLD key, a rank without one would fall back to the VALU kernel
  const bool has_pk = ((size_t)d.big * d.ks_level) % 64 == 0;
  const int n_wb = (d.n + 1 + 15) / 16;
  const size_t pk_b = has_pk ? (size_t)(d.big * d.ks_level / 64) * n_wb * 8 * 64 * 16 : 0;
  if (has_pk && !ctx->ksk_pk) FHES_HIP(hipMalloc(&ctx->ksk_pk, pk_b));
  int rc = dev_broadcast(c, ctx->ksk, ksk_b, root);
  if (rc == FHES_OK && has_pk) rc = dev_broadcast(c, ctx->ksk_pk, pk_b, root);
  if (rc == FHES_OK && has_pk) ctx->ks_n_wb = n_wb;
  if (rc == FHES_OK) rc = dev_broadcast(c, ctx->ksk_bias, bias_b, root);
  if (rc == FHES_OK) rc = dev_broadcast(c, ctx->bsk_f, bsk_b, root);
  {   // the standard-domain words travel too (same size: N u64 per polynomial = M double2), so that every rank can serialise its key
    if (!ctx->bsk_std) FHES_HIP(hipMalloc(&ctx->bsk_std, bsk_b));
    if (rc == FHES_OK) rc = dev_broadcast(c, ctx->bsk_std, bsk_b, root);
  }
  if (rc == FHES_OK && has_e) rc = dev_broadcast(c, ctx->bsk_e, bsk_b, root);
  if (rc == FHES_OK && d.g > 1) {  // the layout of a multi-bit key (engine.h: mb_ilv) belongs to the key: it travels with it
    uint64_t *flag = nullptr;
    FHES_HIP(hipMalloc(reinterpret_cast<void **>(&flag), 8));
    uint64_t v = (uint64_t)ctx->mb_ilv;
    hipError_t he = hipMemcpy(flag, &v, 8, hipMemcpyHostToDevice);
    if (he == hipSuccess) {
      rc = dev_broadcast(c, flag, 8, root);
      if (rc == FHES_OK) he = hipStreamSynchronize(ctx->stream);
      if (rc == FHES_OK && he == hipSuccess) he = hipMemcpy(&v, flag, 8, hipMemcpyDeviceToHost);
    }
    (void)hipFree(flag);
    if (he != hipSuccess) return fail(FHES_ERR_NO_DEVICE, "broadcast_keys: %s", hipGetErrorString(he));
    if (rc == FHES_OK) ctx->mb_ilv = (int)v;
  }
  if (rc == FHES_OK) { FHES_HIP(hipStreamSynchronize(ctx->stream)); ctx->keys_loaded = true; }
  return rc;
}

static fhes_program *comm_program(fhes_comm *c, const char *op, const std::vector<int32_t> &shape, int32_t bu) {
  std::string key = op;
  for (int32_t v : shape) key += ":" + std::to_string(v);
  key += ":u" + std::to_string(bu);
  auto it = c->progs.find(key);
  if (it != c->progs.end()) return it->second;
  fhes_program *p = fhes_program_compile(&c->params, op, shape.data(), (int32_t)shape.size(), bu);
  if (p) c->progs[key] = p;
  return p;
}

// one status word per rank after the last stage of an operation: a rank that failed locally has still entered
// every collective (with zero rows), so nobody hangs, and EVERY rank learns that the result is void
static int commit(fhes_comm *c, int rc_local, const char *what) {
  if (c->world == 1) return rc_local;
  int64_t mine = rc_local;
  std::vector<int64_t> all((size_t)c->world, 0);
  const int rc = fhes_comm_all_gather_host(c, &mine, all.data(), 8);
  if (rc) return rc;
  for (int r = 0; r < c->world; r++)
    if (all[(size_t)r] != 0)
      return r == c->rank ? rc_local
                          : fail(FHES_ERR_STATE, "%s: rank %d failed with status %lld; the result is void on every rank", what, r, (long long)all[(size_t)r]);
  return FHES_OK;
}

int fhes_sharded_op(fhes_comm *c, const char *op_in, int32_t a_first, int32_t a_slots, int32_t p_first,
                    int32_t p_slots, int32_t bu, int32_t *out_handles, int32_t max_out, int32_t *n_out) {
  if (!c || (!c->ctx && !c->has_exec) || !op_in || !out_handles || !n_out || a_slots < 1 || p_slots < 1 || a_first < 0 || p_first < 0)
    return fail(FHES_ERR_INVALID, "sharded_op: bad argument");
  std::string o = op_in, suffix;
  {
    static const std::string ld = "_logdepth";  // opt-in parallel-prefix forms: passed on to every stage
    if (o.size() > ld.size() && o.compare(o.size() - ld.size(), ld.size(), ld) == 0) { suffix = ld; o.resize(o.size() - ld.size()); }
  }
  const bool is_split = o == "split" || o == "rsplit";
  const char *reduce = o == "contains" ? "any_of" : o == "find" ? "index_of" : o == "rfind_last" ? "rindex_of" : nullptr;
  if (!reduce && !is_split) return fail(FHES_ERR_INVALID, "sharded_op: op must be contains / find / rfind_last / split / rsplit");
  std::lock_guard<std::mutex> lk(c->mu);
  X x{c};
  const int msg_mod = 1 << c->params.message_bits;
  int bc = 0;
  for (int v = 1; v < 128; v *= msg_mod) bc++;  // blocks per char (lib.rs:60-61)
  const int64_t n_pos = a_slots - 1;               // start positions (search.rs:34)
  int64_t lo = 0, hi = 0;
  int rc = fhes_shard_range(n_pos, c->world, c->rank, &lo, &hi);
  if (rc) return rc;
  const int32_t chunk = (int32_t)((n_pos + c->world - 1) / c->world);
  const int32_t n_split = a_slots + 1;             // slices of split_opt: max_len(self) + 2 (split.rs:270)
  const int32_t n_final = is_split ? n_split * (1 + 2 * bu) : (o == "contains" ? 1 : 1 + bu);
  if (max_out < (is_split ? 3 : n_final)) return fail(FHES_ERR_INVALID, "sharded_op: need room for %d handles", is_split ? 3 : n_final);
  std::vector<int32_t> ab;                         // handles of the two operands, in program order
  for (int k = 0; k < a_slots * bc; k++) ab.push_back(a_first + k);
  for (int k = 0; k < p_slots * bc; k++) ab.push_back(p_first + k);
  // everything below runs on EVERY rank in the same order; a local failure is carried in `rc` past the collectives
  std::vector<std::pair<int32_t, int32_t>> owned;  // scratch ranges to hand back
  auto take = [&](int32_t n) { int32_t h = n > 0 ? x.alloc(n) : -1; if (h >= 0) owned.push_back({h, n}); else if (n > 0 && rc == FHES_OK) rc = h; return h; };
  auto drop_all = [&]() { for (auto &r : owned) x.release(r.first, r.second); owned.clear(); };
  auto run_prog = [&](const char *name, const std::vector<int32_t> &shape, const std::vector<int32_t> &in, std::vector<int32_t> &out) {
    if (rc != FHES_OK) return;
    fhes_program *pr = comm_program(c, (std::string(name) + suffix).c_str(), shape, bu);
    if (!pr) { rc = FHES_ERR_INVALID; return; }
    int64_t info[8];
    fhes_program_info(pr, info);
    out.assign((size_t)std::max<int64_t>(info[1], 1), -1);
    if ((int64_t)in.size() != info[0]) { rc = fail(FHES_ERR_STATE, "internal: %s takes %lld inputs, %zu given", name, (long long)info[0], in.size()); return; }
    rc = x.run(pr, in.data(), out.data());
    if (rc != FHES_OK) out.assign(out.size(), -1);
  };

  // ---- stage 1: found[i] = substr_eq(i, p) for this rank's block of start positions (search.rs:33-41) ----
  const bool partial_any = o == "contains" && c->world > 1 && c->world <= 15 && suffix.empty();
  int32_t G = -1;                                  // gathered found[]: world segments of `chunk` rows, found[i] = row G + i
  std::vector<int32_t> local;                      // this rank's found[lo, hi)
  if (hi > lo) run_prog("find_all_range", {a_slots, p_slots, (int32_t)lo, (int32_t)hi}, ab, local);
  if (rc == FHES_OK && hi > lo) owned.push_back({local[0], (int32_t)(hi - lo)});
  std::vector<int32_t> result;
  if (partial_any) {
    // contains (search.rs:17-20): `any` of the local block, then ONE all-reduce of that single
    // boolean -- LWE addition is wrapping u64 addition -- and a final lookup [sum > 0] (sum <= world <= 15 < 16)
    const int32_t R = take(1);
    if (R >= 0) {
      int zr = x.zero_rows(R, 1);             // a rank without positions contributes the trivial 0
      if (zr && rc == FHES_OK) rc = zr;
      if (hi > lo && rc == FHES_OK) {
        std::vector<int32_t> part;
        run_prog("any_of", {(int32_t)(hi - lo)}, local, part);
        if (rc == FHES_OK) { owned.push_back({part[0], 1}); rc = x.copy_rows(R, part[0], 1); }
      }
      int cr = x.all_reduce_rows(R, 1);
      if (cr && rc == FHES_OK) rc = cr;
      run_prog("sum_gt_zero", {c->world}, {R}, result);
    }
  } else {
    if (n_pos > 0) {
      G = take(chunk * c->world);
      if (G >= 0) {
        if (hi > lo && rc == FHES_OK) rc = x.copy_rows(G + (int32_t)lo, local[0], (int32_t)(hi - lo));
        int cr = x.all_gather_rows(G, chunk);
        if (cr && rc == FHES_OK) rc = cr;
      }
    }
    std::vector<int32_t> found((size_t)n_pos);
    for (int64_t i = 0; i < n_pos; i++) found[(size_t)i] = G + (int32_t)i;
    if (!is_split) {
      run_prog(reduce, {(int32_t)n_pos}, found, result);   // replicated: deterministic, identical on every rank
    } else if (!suffix.empty()) {
      // log-depth forms: everything behind find_all in one replicated program -- the barrel shifter that replaces the
      // per-index gathers is cheaper than the grid divided by the ranks and does not partition by index
      std::vector<int32_t> in2(ab);
      in2.insert(in2.end(), found.begin(), found.end());
      run_prog((o + ".rest").c_str(), {a_slots, p_slots}, in2, result);
    } else {
      // ---- split / rsplit (split.rs:238-333) ----
      // stage 2 (replicated): non-overlapping matches, len(p), len(self), pattern_empty
      std::vector<int32_t> in2(ab), head;
      in2.insert(in2.end(), found.begin(), found.end());
      run_prog(o == "split" ? "split_head" : "rsplit_head", {a_slots, p_slots}, in2, head);
      if (rc == FHES_OK) owned.push_back({head[0], n_split + 2 * bu + 1});
      // stage 3: is_some_i = element_at_bool(matches, i - len(p)) & ... for this rank's block of OUTPUT indices
      // (mod.rs:271-287 called per index at split.rs:283-285: the O(n^2) grid), one all-gather of the booleans
      int64_t lo3 = 0, hi3 = 0;
      (void)fhes_shard_range(n_split, c->world, c->rank, &lo3, &hi3);
      const int32_t chunk3 = (n_split + c->world - 1) / c->world;
      const int32_t G3 = take(chunk3 * c->world);
      if (G3 >= 0 && hi3 > lo3) {
        std::vector<int32_t> in3, part;
        if (rc == FHES_OK) for (int i = 0; i < n_split + 2 * bu; i++) in3.push_back(head[0] + i);   // matches, len(p), len(self)
        run_prog("split_is_some_range", {n_split, (int32_t)lo3, (int32_t)hi3}, in3, part);
        if (rc == FHES_OK) { owned.push_back({part[0], (int32_t)(hi3 - lo3)}); rc = x.copy_rows(G3 + (int32_t)lo3, part[0], (int32_t)(hi3 - lo3)); }
      }
      if (G3 >= 0) { int cr = x.all_gather_rows(G3, chunk3); if (cr && rc == FHES_OK) rc = cr; }
      // stage 4 (replicated): the carried next_match chain and the slice bounds
      std::vector<int32_t> in4;
      if (rc == FHES_OK) {
        for (int i = 0; i < n_split; i++) in4.push_back(head[0] + i);
        for (int i = 0; i < n_split; i++) in4.push_back(G3 + i);
        for (int i = 0; i < 2 * bu + 1; i++) in4.push_back(head[0] + n_split + i);
      }
      run_prog(o == "split" ? "split_tail" : "rsplit_tail", {n_split}, in4, result);
    }
  }
  drop_all();
  rc = commit(c, rc, op_in);
  if (rc) {
    if (!result.empty() && result[0] >= 0) x.release(result[0], n_final);
    return rc;
  }
  if (is_split) {  // one contiguous range: is_some[n], start[n][Bu], end[n][Bu]
    out_handles[0] = result[0];
    out_handles[1] = result[0] + n_split;
    out_handles[2] = result[0] + n_split * (1 + bu);
    *n_out = 3;
  } else {
    for (int k = 0; k < n_final; k++) out_handles[k] = result[0] + k;
    *n_out = n_final;
  }
  return FHES_OK;
}

// Every operation whose circuit starts with find_all(self, pattern) (search.rs:33-41) -- rfind, the split family, split_once,
// strip_suffix, replace / replacen -- shards that part the same way: found[] per rank + one all-gather, then the operation's
// "<op>.rest" program (circuit/strings.cpp: the same circuit with find_all handed in) replicated.  Operands are the operation's
// ciphertext operands in signature order (fhes_op_signature: S / U / C / B), as first handle + block count; the first two S
// operands are (self, pattern).  Results: one contiguous range, its parts as fhes_program_out_parts reports them.
int fhes_sharded_op_ex(fhes_comm *c, const char *op_in, const int32_t *operand_first, const int32_t *operand_blocks, int32_t n_operands,
                       const int64_t *iargs, int32_t n_iargs, int32_t bu, int32_t *out_first, int32_t *parts, int32_t max_parts, int32_t *n_parts) {
  if (!c || (!c->ctx && !c->has_exec) || !op_in || !operand_first || !operand_blocks || n_operands < 2 || !out_first || !parts || !n_parts || (n_iargs > 0 && !iargs))
    return fail(FHES_ERR_INVALID, "sharded_op_ex: bad argument");
  const char *sig = fhes_op_signature(op_in);
  if (!sig) return fail(FHES_ERR_INVALID, "sharded_op_ex: unknown operation '%s'", op_in);
  std::string o = op_in, suffix;
  {
    static const std::string ld = "_logdepth";
    if (o.size() > ld.size() && o.compare(o.size() - ld.size(), ld.size(), ld) == 0) { suffix = ld; o.resize(o.size() - ld.size()); }
  }
  static const char *kHasFindAll[] = {"rfind", "split", "rsplit", "split_inclusive", "split_terminator", "rsplit_terminator", "splitn", "rsplitn",
                                      "split_once", "rsplit_once", "strip_suffix", "replace", "replacen", "find", "contains"};
  bool known = false;
  for (const char *k : kHasFindAll) known = known || o == k;
  if (!known) return fail(FHES_ERR_INVALID, "sharded_op_ex: '%s' has no find_all(self, pattern) to shard", op_in);
  std::lock_guard<std::mutex> lk(c->mu);
  X x{c};
  const int msg_mod = 1 << c->params.message_bits;
  int bc = 0;
  for (int v = 1; v < 128; v *= msg_mod) bc++;
  // shape of the operation and the handles of (self, pattern)
  std::vector<int32_t> shape, all_in, str_first, str_slots;
  int ai = 0, ii = 0;
  for (const char *q = sig; *q; ++q) {
    if (*q == 'I') {
      if (ii >= n_iargs || iargs[ii] < 0 || iargs[ii] > INT32_MAX) return fail(FHES_ERR_INVALID, "sharded_op_ex: integer argument %d missing or out of range", ii);
      shape.push_back((int32_t)iargs[ii++]);
      continue;
    }
    if (ai >= n_operands) return fail(FHES_ERR_INVALID, "sharded_op_ex: '%s' (signature %s) needs more operands", op_in, sig);
    const int32_t first = operand_first[ai], blocks = operand_blocks[ai];
    const int32_t want = *q == 'U' ? bu : *q == 'C' ? bc : *q == 'B' ? 1 : -1;
    if (first < 0 || blocks < 1 || (want > 0 && blocks != want) || (*q == 'S' && blocks % bc) || (*q != 'S' && want < 0))
      return fail(FHES_ERR_INVALID, "sharded_op_ex: operand %d does not fit signature %s", ai, sig);
    if (*q == 'S') { shape.push_back(blocks / bc); str_first.push_back(first); str_slots.push_back(blocks / bc); }
    for (int k = 0; k < blocks; k++) all_in.push_back(first + k);
    ai++;
  }
  if (ai != n_operands || ii != n_iargs || str_first.size() < 2) return fail(FHES_ERR_INVALID, "sharded_op_ex: wrong operand count for '%s' (signature %s)", op_in, sig);
  const int32_t a_slots = str_slots[0], p_slots = str_slots[1];
  const int64_t n_pos = a_slots - 1;
  int64_t lo = 0, hi = 0;
  int rc = fhes_shard_range(n_pos, c->world, c->rank, &lo, &hi);
  if (rc) return rc;
  const int32_t chunk = (int32_t)((n_pos + c->world - 1) / c->world);
  std::vector<std::pair<int32_t, int32_t>> owned;
  auto run_prog = [&](const std::string &name, const std::vector<int32_t> &shp, const std::vector<int32_t> &in, std::vector<int32_t> &out, fhes_program **used) {
    if (rc != FHES_OK) return;
    fhes_program *pr = comm_program(c, name.c_str(), shp, bu);
    if (used) *used = pr;
    if (!pr) { rc = FHES_ERR_INVALID; return; }
    int64_t info[8];
    fhes_program_info(pr, info);
    out.assign((size_t)std::max<int64_t>(info[1], 1), -1);
    if ((int64_t)in.size() != info[0]) { rc = fail(FHES_ERR_STATE, "internal: %s takes %lld inputs, %zu given", name.c_str(), (long long)info[0], in.size()); return; }
    rc = x.run(pr, in.data(), out.data());
  };
  std::vector<int32_t> ab, local;
  for (int k = 0; k < a_slots * bc; k++) ab.push_back(str_first[0] + k);
  for (int k = 0; k < p_slots * bc; k++) ab.push_back(str_first[1] + k);
  if (hi > lo) run_prog("find_all_range", {a_slots, p_slots, (int32_t)lo, (int32_t)hi}, ab, local, nullptr);
  if (rc == FHES_OK && hi > lo) owned.push_back({local[0], (int32_t)(hi - lo)});
  int32_t G = -1;
  if (n_pos > 0) {
    G = x.alloc(chunk * c->world);
    if (G < 0) { if (rc == FHES_OK) rc = G; }
    else {
      owned.push_back({G, chunk * c->world});
      if (hi > lo && rc == FHES_OK) rc = x.copy_rows(G + (int32_t)lo, local[0], (int32_t)(hi - lo));
      const int cr = x.all_gather_rows(G, chunk);
      if (cr && rc == FHES_OK) rc = cr;
    }
  }
  std::vector<int32_t> in2(all_in), result;
  for (int64_t i = 0; i < n_pos; i++) in2.push_back(G + (int32_t)i);
  fhes_program *rest = nullptr;
  run_prog(o + ".rest" + suffix, shape, in2, result, &rest);
  for (auto &r : owned) x.release(r.first, r.second);
  int64_t info[8] = {0};
  if (rest) fhes_program_info(rest, info);
  rc = commit(c, rc, op_in);
  if (rc) {
    if (!result.empty() && result[0] >= 0) x.release(result[0], (int32_t)info[1]);
    return rc;
  }
  const int32_t np = fhes_program_out_parts(rest, parts, max_parts);
  if (np < 1 || np > max_parts) { x.release(result[0], (int32_t)info[1]); return fail(FHES_ERR_INVALID, "sharded_op_ex: need room for %d parts", np); }
  *n_parts = np;
  *out_first = result[0];
  return FHES_OK;
}

int fhes_sharded_search(fhes_comm *c, const char *op, int32_t a_first, int32_t a_slots, int32_t p_first,
                        int32_t p_slots, int32_t bu, int32_t *out_handles, int32_t max_out, int32_t *n_out) {
  const std::string o = op ? op : "";
  if (o.rfind("contains", 0) != 0 && o.rfind("find", 0) != 0 && o.rfind("rfind_last", 0) != 0)
    return fail(FHES_ERR_INVALID, "sharded_search: op must be contains / find / rfind_last");
  return fhes_sharded_op(c, op, a_first, a_slots, p_first, p_slots, bu, out_handles, max_out, n_out);
}

}  // extern "C"
