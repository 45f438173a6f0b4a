// seams.cpp -- the host-side seams of the C ABI that own no device state: the BufferAllocator-shaped pinned allocator
// with its MemoryLimit quota (supersonic/base/memory/memory.h:100-233,465-520) and the order-preserving STRING
// dictionary (StringPiece order: supersonic/base/infrastructure/types_infrastructure.h:238-246; the Arena deep-copy rule:
// supersonic/base/memory/arena.h, cursor/core/filter.cc:205-230).
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <limits>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/ssgpu.h"

struct ssgpu_allocator {
  bool pinned = false;       // hipHostMalloc (a device is bound) or plain aligned host memory (bind-only context)
  int64_t quota = -1;        // soft quota in bytes, -1 = unlimited
  int64_t used = 0;
  std::unordered_map<void*, size_t> live;
};

extern "C" {

void* ssgpu_ctx_stream(ssgpu_ctx* ctx);
int ssgpu_ctx_has_device(const ssgpu_ctx* ctx);

int ssgpu_allocator_create(ssgpu_ctx* ctx, int64_t quota_bytes, ssgpu_allocator** out) {
  if (!ctx || !out) return SSGPU_ERROR_INVALID_ARGUMENT_VALUE;
  ssgpu_allocator* a = new ssgpu_allocator;
  a->pinned = ssgpu_ctx_has_device(ctx) != 0;
  a->quota = quota_bytes < 0 ? -1 : quota_bytes;
  *out = a;
  return SSGPU_OK;
}

static void raw_free(ssgpu_allocator* a, void* p) { if (a->pinned) (void)hipHostFree(p); else free(p); }
static void* raw_alloc(ssgpu_allocator* a, size_t bytes) {
  void* p = nullptr;
  const size_t n = std::max<size_t>(bytes, 1);           // zero-size requests return non-NULL data (memory.h:112-117)
  if (a->pinned) { if (hipHostMalloc(&p, n, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; } }
  else if (posix_memalign(&p, 256, (n + 255) & ~size_t(255)) != 0) return nullptr;
  return p;
}

void ssgpu_allocator_destroy(ssgpu_allocator* a) {
  if (!a) return;
  for (auto& kv : a->live) raw_free(a, kv.first);
  delete a;
}

int64_t ssgpu_allocator_available(const ssgpu_allocator* a) {
  if (!a) return 0;
  return a->quota < 0 ? std::numeric_limits<int64_t>::max() : std::max<int64_t>(a->quota - a->used, 0);
}
int64_t ssgpu_allocator_allocated(const ssgpu_allocator* a) { return a ? a->used : 0; }

int ssgpu_allocator_allocate(ssgpu_allocator* a, size_t requested, size_t minimal, void** out, size_t* granted) {
  if (!a || !out || minimal > requested) return SSGPU_ERROR_INVALID_ARGUMENT_VALUE;
  *out = nullptr; if (granted) *granted = 0;
  size_t grant = requested;
  if (a->quota >= 0) {
    const int64_t avail = std::max<int64_t>(a->quota - a->used, 0);
    if ((int64_t)grant > avail) grant = (size_t)avail;          // best effort: as much as the quota leaves ...
    if (grant < minimal) return SSGPU_ERROR_MEMORY_EXCEEDED;     // ... but never less than `minimal`
  }
  void* p = raw_alloc(a, grant);
  if (!p) return SSGPU_ERROR_MEMORY_EXCEEDED;
  a->live[p] = grant; a->used += (int64_t)grant;
  *out = p; if (granted) *granted = grant;
  return SSGPU_OK;
}

int ssgpu_allocator_reallocate(ssgpu_allocator* a, void* p, size_t requested, size_t minimal, void** out, size_t* granted) {
  if (!a || !out) return SSGPU_ERROR_INVALID_ARGUMENT_VALUE;
  if (!p) return ssgpu_allocator_allocate(a, requested, minimal, out, granted);
  auto it = a->live.find(p);
  if (it == a->live.end() || minimal > requested) return SSGPU_ERROR_INVALID_ARGUMENT_VALUE;
  const size_t old = it->second;
  size_t grant = requested;
  if (a->quota >= 0) {
    // conservative like the reference's mediator (memory.h, memory_test.cc:183-198: "realloc might degenerate to
    // create-copy-free"): the whole new size has to fit NEXT TO the old buffer
    const int64_t avail = std::max<int64_t>(a->quota - a->used, 0);
    if ((int64_t)grant > avail) grant = (size_t)avail;
    if (grant < minimal) { *out = nullptr; return SSGPU_ERROR_MEMORY_EXCEEDED; }       // the old buffer stays valid
  }
  void* q = raw_alloc(a, grant);
  if (!q) { *out = nullptr; return SSGPU_ERROR_MEMORY_EXCEEDED; }
  memcpy(q, p, std::min(old, grant));
  raw_free(a, p);
  a->live.erase(it);
  a->live[q] = grant; a->used += (int64_t)grant - (int64_t)old;
  *out = q; if (granted) *granted = grant;
  return SSGPU_OK;
}

void ssgpu_allocator_free(ssgpu_allocator* a, void* p) {
  if (!a || !p) return;
  auto it = a->live.find(p);
  if (it == a->live.end()) return;
  a->used -= (int64_t)it->second;
  raw_free(a, p);
  a->live.erase(it);
}

}  // extern "C"

// ---- order-preserving dictionary ---------------------------------------------------------------------------------
struct ssgpu_dict {
  std::vector<std::string> values;                  // sorted: memcmp, then length == std::string's operator< on bytes
  std::unordered_map<std::string, int32_t> code;
};

extern "C" {

int ssgpu_dict_create(const char* const* strings, const int32_t* lengths, int64_t n, ssgpu_dict** out) {
  if (!out || n < 0 || (n > 0 && (!strings || !lengths))) return SSGPU_ERROR_INVALID_ARGUMENT_VALUE;
  ssgpu_dict* d = new ssgpu_dict;
  d->values.reserve((size_t)n);
  for (int64_t i = 0; i < n; ++i) if (strings[i] && lengths[i] >= 0) d->values.emplace_back(strings[i], (size_t)lengths[i]);
  // std::string compares as unsigned bytes, shorter-is-less on a common prefix: the reference's StringPiece order
  std::sort(d->values.begin(), d->values.end());
  d->values.erase(std::unique(d->values.begin(), d->values.end()), d->values.end());
  if (d->values.size() > (size_t)std::numeric_limits<int32_t>::max()) { delete d; return SSGPU_ERROR_MEMORY_EXCEEDED; }
  for (size_t i = 0; i < d->values.size(); ++i) d->code[d->values[i]] = (int32_t)i;
  *out = d;
  return SSGPU_OK;
}
void ssgpu_dict_destroy(ssgpu_dict* d) { delete d; }
int32_t ssgpu_dict_size(const ssgpu_dict* d) { return d ? (int32_t)d->values.size() : 0; }

int ssgpu_dict_encode(const ssgpu_dict* d, const char* const* strings, const int32_t* lengths, const uint8_t* is_null, int64_t n, int32_t* codes) {
  if (!d || !codes || n < 0 || (n > 0 && (!strings || !lengths))) return SSGPU_ERROR_INVALID_ARGUMENT_VALUE;
  int rc = SSGPU_OK;
  for (int64_t i = 0; i < n; ++i) {
    if (is_null && is_null[i]) { codes[i] = 0; continue; }
    auto it = d->code.find(std::string(strings[i] ? strings[i] : "", strings[i] ? (size_t)lengths[i] : 0));
    if (it == d->code.end()) { codes[i] = -1; rc = SSGPU_ERROR_INVALID_ARGUMENT_VALUE; } else codes[i] = it->second;
  }
  return rc;
}
int ssgpu_dict_decode(const ssgpu_dict* d, int32_t code, const char** bytes, int32_t* length) {
  if (!d || code < 0 || (size_t)code >= d->values.size() || !bytes || !length) return SSGPU_ERROR_INVALID_ARGUMENT_VALUE;
  *bytes = d->values[(size_t)code].data(); *length = (int32_t)d->values[(size_t)code].size();
  return SSGPU_OK;
}

}  // extern "C"
