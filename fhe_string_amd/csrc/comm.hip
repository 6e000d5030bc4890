// comm.hip -- multi-GPU level of the C ABI: one process per GPU, position-sharded search.
// the positions of find_all (/root/reference/src/ciphertext/search.rs:33-41) are
// independent; the serial scans are not.  Rank g runs substr_eq for its block of positions, one
// all-gather of the per-position BooleanBlocks, replicated reduction.  Transports: RCCL on device
// pointers (resolved with dlopen: no link-time dependency) or caller-supplied host callbacks.
#include <dlfcn.h>
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
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  std::string error;
};

RcclApi *rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      api.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (api.lib) break;
    }
    if (!api.lib) { api.error = std::string("cannot load librccl: ") + dlerror(); return; }
#define FHES_SYM(field, sym)                                                        \
  api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.lib, sym));          \
  if (!api.field) { api.error = std::string("librccl lacks ") + sym; return; }
    FHES_SYM(GetUniqueId, "ncclGetUniqueId")
    FHES_SYM(CommInitRank, "ncclCommInitRank")
    FHES_SYM(CommDestroy, "ncclCommDestroy")
    FHES_SYM(AllGather, "ncclAllGather")
    FHES_SYM(Broadcast, "ncclBroadcast")
    FHES_SYM(GetErrorString, "ncclGetErrorString")
#undef FHES_SYM
  });
  return api.error.empty() ? &api : nullptr;
}

int rccl_fail(ncclResult_t r, const char *what) {
  RcclApi *a = rccl();
  return fail(FHES_ERR_NO_DEVICE, "RCCL error %d (%s) in %s", (int)r, a ? a->GetErrorString(r) : "?", what);
}

}  // namespace

struct fhes_comm {
  fhes_ctx *ctx = nullptr;
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

extern "C" {

int fhes_comm_unique_id(uint8_t id[FHES_COMM_ID_BYTES]) {
  static_assert(sizeof(ncclUniqueId) == FHES_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
  if (!id) return fail(FHES_ERR_INVALID, "null argument");
  RcclApi *a = rccl();
  if (!a) return fail(FHES_ERR_NO_DEVICE, "RCCL unavailable");
  ncclUniqueId u;
  FHES_NCCL(a->GetUniqueId(&u));
  memcpy(id, &u, sizeof(u));
  return FHES_OK;
}

fhes_comm *fhes_comm_create_rccl(fhes_ctx *ctx, int32_t rank, int32_t world, const uint8_t id[FHES_COMM_ID_BYTES]) {
  if (!ctx || !id || world < 1 || rank < 0 || rank >= world) { fail(FHES_ERR_INVALID, "comm_create_rccl: bad argument"); return nullptr; }
  RcclApi *a = rccl();
  if (!a) { fail(FHES_ERR_NO_DEVICE, "RCCL unavailable"); return nullptr; }
  if (hipSetDevice(ctx->device) != hipSuccess) { fail(FHES_ERR_NO_DEVICE, "hipSetDevice failed"); return nullptr; }
  ncclUniqueId u;
  memcpy(&u, id, sizeof(u));
  auto *c = new fhes_comm;
  c->ctx = ctx; c->rank = rank; c->world = world;
  ncclResult_t r = a->CommInitRank(&c->nccl, world, u, rank);
  if (r != ncclSuccess) { rccl_fail(r, "ncclCommInitRank"); delete c; return nullptr; }
  return c;
}

fhes_comm *fhes_comm_create_host(fhes_ctx *ctx, int32_t rank, int32_t world, fhes_allgather_fn ag, fhes_bcast_fn bc, void *user) {
  if (world < 1 || rank < 0 || rank >= world || (world > 1 && (!ag || !bc))) { fail(FHES_ERR_INVALID, "comm_create_host: bad argument"); return nullptr; }
  auto *c = new fhes_comm;
  c->ctx = ctx; c->rank = rank; c->world = world; c->ag = ag; c->bc = bc; c->user = user;
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
  return dev_all_gather(c, ctx->arena + (size_t)first * ctx->ct_stride, (size_t)count * ctx->ct_stride * 8);
}

int fhes_comm_broadcast_cts(fhes_comm *c, int32_t first, int32_t count, int32_t root) {
  if (!c || !c->ctx || first < 0 || count <= 0 || root < 0 || root >= c->world) return fail(FHES_ERR_INVALID, "broadcast_cts: bad argument");
  fhes_ctx *ctx = c->ctx;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if ((int64_t)first + count > ctx->capacity) return fail(FHES_ERR_INVALID, "broadcast_cts: range outside the arena");
  FHES_HIP(hipSetDevice(ctx->device));
  return dev_broadcast(c, ctx->arena + (size_t)first * ctx->ct_stride, (size_t)count * ctx->ct_stride * 8, root);
}

int fhes_comm_broadcast_keys(fhes_comm *c, int32_t root) {
  if (!c || !c->ctx || root < 0 || root >= c->world) return fail(FHES_ERR_INVALID, "broadcast_keys: bad argument");
  fhes_ctx *ctx = c->ctx;
  std::lock_guard<std::mutex> lk(ctx->mu);
  FHES_HIP(hipSetDevice(ctx->device));
  if (c->rank == root && !ctx->keys_loaded) return fail(FHES_ERR_STATE, "broadcast_keys: the root has no keys");
  const Dims &d = ctx->dims;
  const size_t ksk_b = (size_t)d.big * d.ks_level * ctx->ks_stride * 8, bias_b = (size_t)ctx->ks_stride * 8;
  const size_t bsk_b = (size_t)d.n_ggsw * 4 * d.M * 16;
  if (!ctx->ksk) FHES_HIP(hipMalloc(&ctx->ksk, ksk_b));
  if (!ctx->ksk_bias) FHES_HIP(hipMalloc(&ctx->ksk_bias, bias_b));
  if (!ctx->bsk_f) FHES_HIP(hipMalloc(&ctx->bsk_f, bsk_b));
  if (!ctx->bsk_e) FHES_HIP(hipMalloc(&ctx->bsk_e, bsk_b));
  int rc = dev_broadcast(c, ctx->ksk, ksk_b, root);
  if (rc == FHES_OK) rc = dev_broadcast(c, ctx->ksk_bias, bias_b, root);
  if (rc == FHES_OK) rc = dev_broadcast(c, ctx->bsk_f, bsk_b, root);
  if (rc == FHES_OK) rc = dev_broadcast(c, ctx->bsk_e, bsk_b, root);
  if (rc == FHES_OK) { FHES_HIP(hipStreamSynchronize(ctx->stream)); ctx->keys_loaded = true; }
  return rc;
}

static fhes_program *comm_program(fhes_comm *c, const char *op, const std::vector<int32_t> &shape, int32_t bu) {
  std::string key = op;
  for (int32_t v : shape) key += ":" + std::to_string(v);
  key += ":u" + std::to_string(bu);
  auto it = c->progs.find(key);
  if (it != c->progs.end()) return it->second;
  fhes_program *p = fhes_program_compile(&c->ctx->p, op, shape.data(), (int32_t)shape.size(), bu);
  if (p) c->progs[key] = p;
  return p;
}

int fhes_sharded_search(fhes_comm *c, const char *op, int32_t a_first, int32_t a_slots, int32_t p_first,
                        int32_t p_slots, int32_t bu, int32_t *out_handles, int32_t max_out, int32_t *n_out) {
  if (!c || !c->ctx || !op || !out_handles || !n_out || a_slots < 1 || p_slots < 1 || a_first < 0 || p_first < 0)
    return fail(FHES_ERR_INVALID, "sharded_search: bad argument");
  const std::string o = op;
  const char *reduce = o == "contains" ? "any_of" : o == "find" ? "index_of" : o == "rfind_last" ? "rindex_of" : nullptr;
  if (!reduce) return fail(FHES_ERR_INVALID, "sharded_search: op must be contains / find / rfind_last");
  std::lock_guard<std::mutex> lk(c->mu);
  fhes_ctx *ctx = c->ctx;
  const int msg_mod = 1 << ctx->p.message_bits;
  int bc = 0;
  for (int v = 1; v < 128; v *= msg_mod) bc++;  // blocks per char (lib.rs:60-61)
  const int64_t n_pos = a_slots - 1;               // start positions (search.rs:34)
  int64_t lo = 0, hi = 0;
  int rc = fhes_shard_range(n_pos, c->world, c->rank, &lo, &hi);
  if (rc) return rc;
  const int32_t chunk = (int32_t)((n_pos + c->world - 1) / c->world);
  const int32_t n_red_out = o == "contains" ? 1 : 1 + bu;
  if (max_out < n_red_out) return fail(FHES_ERR_INVALID, "sharded_search: need room for %d handles", n_red_out);

  // gathered found[]: world segments of `chunk` rows, so that found[i] is row G + i
  int32_t G = -1;
  if (n_pos > 0) {
    G = fhes_arena_alloc(ctx, chunk * c->world);
    if (G < 0) return G;
  }
  auto release = [&]() { if (G >= 0) fhes_arena_free(ctx, G, chunk * c->world); };
  if (hi > lo) {
    fhes_program *pr = comm_program(c, "find_all_range", {a_slots, p_slots, (int32_t)lo, (int32_t)hi}, bu);
    if (!pr) { release(); return FHES_ERR_INVALID; }
    std::vector<int32_t> in;
    for (int k = 0; k < a_slots * bc; k++) in.push_back(a_first + k);
    for (int k = 0; k < p_slots * bc; k++) in.push_back(p_first + k);
    std::vector<int32_t> out((size_t)(hi - lo));
    rc = fhes_program_run(ctx, pr, in.data(), out.data());
    if (rc) { release(); return rc; }
    {  // own segment into place (device-to-device, stream-ordered behind the program)
      std::lock_guard<std::mutex> lk2(ctx->mu);
      hipError_t e = hipSetDevice(ctx->device);
      if (e == hipSuccess)
        e = hipMemcpyAsync(ctx->arena + (size_t)(G + lo) * ctx->ct_stride, ctx->arena + (size_t)out[0] * ctx->ct_stride,
                           (size_t)(hi - lo) * ctx->ct_stride * 8, hipMemcpyDeviceToDevice, ctx->stream);
      if (e != hipSuccess) rc = hip_fail(e, "segment copy");
    }
    fhes_arena_free(ctx, out[0], (int32_t)(hi - lo));
    if (rc) { release(); return rc; }
  }
  if (n_pos > 0 && c->world > 1) {
    rc = fhes_comm_all_gather_cts(c, G, chunk);
    if (rc) { release(); return rc; }
  }
  fhes_program *red = comm_program(c, reduce, {(int32_t)n_pos}, bu);
  if (!red) { release(); return FHES_ERR_INVALID; }
  std::vector<int32_t> in((size_t)n_pos);
  for (int64_t i = 0; i < n_pos; i++) in[(size_t)i] = G + (int32_t)i;
  std::vector<int32_t> out((size_t)n_red_out);
  rc = fhes_program_run(ctx, red, in.data(), out.data());
  release();
  if (rc) return rc;
  for (int k = 0; k < n_red_out; k++) out_handles[k] = out[k];
  *n_out = n_red_out;
  return FHES_OK;
}

}  // extern "C"
