// Sanitizer harness only (tests/asan): the device entry points the host-side sources refer to, answered as a
// machine without a HIP device answers them.  This is NOT a CPU fallback of the product: nothing computes, every
// call fails with FHES_ERR_NO_DEVICE exactly like libfhe_string_amd.so does on a box without a GPU, so that
// csrc/client.cpp, csrc/objects.cpp and csrc/circuit/*.cpp link into an ASan + UBSan executable with g++.
#include "../../fhe_string_amd/csrc/common.h"

using fhes::fail;

extern "C" {
fhes_ctx *fhes_ctx_create(const fhes_params *, int32_t) { fail(FHES_ERR_NO_DEVICE, "sanitizer harness: no HIP device"); return nullptr; }
void fhes_ctx_destroy(fhes_ctx *) {}
int fhes_upload_keys(fhes_ctx *, const uint64_t *, const uint64_t *) { return fail(FHES_ERR_NO_DEVICE, "sanitizer harness: no HIP device"); }
int fhes_download_keys(fhes_ctx *, uint64_t *, uint64_t *) { return fail(FHES_ERR_NO_DEVICE, "sanitizer harness: no HIP device"); }
int32_t fhes_arena_alloc(fhes_ctx *, int32_t) { return fail(FHES_ERR_NO_DEVICE, "sanitizer harness: no HIP device"); }
int fhes_arena_free(fhes_ctx *, int32_t, int32_t) { return FHES_OK; }
int fhes_arena_upload(fhes_ctx *, int32_t, int32_t, const uint64_t *) { return fail(FHES_ERR_NO_DEVICE, "sanitizer harness: no HIP device"); }
int fhes_arena_download(fhes_ctx *, int32_t, int32_t, uint64_t *) { return fail(FHES_ERR_NO_DEVICE, "sanitizer harness: no HIP device"); }
int fhes_linear_batch(fhes_ctx *, int32_t, const int32_t *, const int32_t *, const int64_t *, const uint64_t *, const int32_t *) {
  return fail(FHES_ERR_NO_DEVICE, "sanitizer harness: no HIP device");
}
fhes_program *fhes_program_compile(const fhes_params *, const char *, const int32_t *, int32_t, int32_t) {
  fail(FHES_ERR_NO_DEVICE, "sanitizer harness: programs are compiled through circ::compile_op");
  return nullptr;
}
void fhes_program_destroy(fhes_program *) {}
int fhes_program_info(const fhes_program *, int64_t *) { return fail(FHES_ERR_NO_DEVICE, "sanitizer harness"); }
int32_t fhes_program_out_parts(const fhes_program *, int32_t *, int32_t) { return fail(FHES_ERR_NO_DEVICE, "sanitizer harness"); }
int fhes_program_out_meta(const fhes_program *, int32_t *, double *) { return fail(FHES_ERR_NO_DEVICE, "sanitizer harness"); }
int fhes_program_run(fhes_ctx *, const fhes_program *, const int32_t *, int32_t *) { return fail(FHES_ERR_NO_DEVICE, "sanitizer harness"); }
int32_t fhes_comm_world(const fhes_comm *) { return 1; }
int fhes_sharded_search(fhes_comm *, const char *, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t *, int32_t, int32_t *) {
  return fail(FHES_ERR_NO_DEVICE, "sanitizer harness");
}
int fhes_sharded_op(fhes_comm *, const char *, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t *, int32_t, int32_t *) {
  return fail(FHES_ERR_NO_DEVICE, "sanitizer harness");
}
int fhes_sharded_op_ex(fhes_comm *, const char *, const int32_t *, const int32_t *, int32_t, const int64_t *, int32_t, int32_t, int32_t *, int32_t *,
                       int32_t, int32_t *) {
  return fail(FHES_ERR_NO_DEVICE, "sanitizer harness");
}
}
namespace fhes {
bool ctx_is_live(fhes_ctx *) { return false; }
}

#include "../../fhe_string_amd/csrc/circuit/strings.h"
extern "C" const char *fhes_op_signature(const char *op) { return op ? fhes::circ::op_signature(op) : nullptr; }
