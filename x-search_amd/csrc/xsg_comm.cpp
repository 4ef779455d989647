// xsg_comm.cpp -- the one exchange step of a multi-GPU search: RCCL over xGMI.
//
// The reference has no distributed backend (only N threads over chunks,
// include/xsearch/Searcher.h:141-145).  Chunk ranges shard across GPUs with no data-path
// collective; what the devices exchange is tiny and happens once per search:
//   - ncclAllReduce(sum) of the counter vector {matches, lines, newlines, bytes}  (xs::count / xs::count_lines)
//   - ncclAllGather of one uint64 per device: its newline total, from which every device derives the line-index
//     base of its chunk range                                                     (xs::line_indices, no metafile)
// Messages are 8-32 bytes: latency-bound, the xGMI link rate is irrelevant.
//
// librccl is bound at run time (dlopen), never at link time: a process that already carries an RCCL -- PyTorch-ROCm
// bundles its own next to its own HIP runtime -- must use THAT copy (two RCCLs / two HIP runtimes in one process do
// not see each other's devices), and hosts that only ever search on one GPU need no RCCL at all.  Without a usable
// librccl the create calls fail with XSG_ENOTSUP and callers fall back to summing on the host (documented in xsg.h).
#include <dlfcn.h>
#include <link.h>
#include <rccl/rccl.h>

#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "xsg_objects.h"

using namespace xsg;

namespace {

struct Rccl {
  void* h = nullptr;
  std::string path;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

std::mutex g_mu;
Rccl g_rccl;
bool g_tried = false;

int find_loaded_rccl(struct dl_phdr_info* info, size_t, void* data) {
  if (info->dlpi_name && strstr(info->dlpi_name, "librccl")) {
    *static_cast<std::string*>(data) = info->dlpi_name;
    return 1;
  }
  return 0;
}

int need_rccl() {
  std::lock_guard<std::mutex> g(g_mu);
  if (g_rccl.h) return XSG_OK;
  if (g_tried) return fail(XSG_ENOTSUP, "no usable librccl on this host");
  g_tried = true;
  std::string loaded;
  dl_iterate_phdr(find_loaded_rccl, &loaded);  // the copy this process already carries wins
  void* h = nullptr;
  if (const char* env = getenv("XSG_RCCL_LIB")) {
    h = dlopen(env, RTLD_NOW | RTLD_LOCAL);
    if (h) loaded = env;
  }
  if (!h && !loaded.empty()) h = dlopen(loaded.c_str(), RTLD_NOW | RTLD_LOCAL);
  if (!h) {
    static const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", nullptr};
    for (const char* const* n = names; *n && !h; ++n) {
      h = dlopen(*n, RTLD_NOW | RTLD_LOCAL);
      if (h) loaded = *n;
    }
  }
  if (!h) {
    g_tried = getenv("XSG_RCCL_LIB") == nullptr;  // with an explicit path the next call looks again (it may have been corrected)
    return fail(XSG_ENOTSUP, "librccl not found (%s)", dlerror());
  }
  Rccl r;
  r.h = h;
  r.path = loaded;
  r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
  r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
  r.CommInitAll = (decltype(r.CommInitAll))dlsym(h, "ncclCommInitAll");
  r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
  r.AllReduce = (decltype(r.AllReduce))dlsym(h, "ncclAllReduce");
  r.AllGather = (decltype(r.AllGather))dlsym(h, "ncclAllGather");
  r.GroupStart = (decltype(r.GroupStart))dlsym(h, "ncclGroupStart");
  r.GroupEnd = (decltype(r.GroupEnd))dlsym(h, "ncclGroupEnd");
  r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
  if (!r.GetUniqueId || !r.CommInitRank || !r.CommInitAll || !r.CommDestroy || !r.AllReduce || !r.AllGather ||
      !r.GroupStart || !r.GroupEnd || !r.GetErrorString) {
    dlclose(h);
    g_tried = false;  // a later call may find another copy (XSG_RCCL_LIB set in the meantime)
    return fail(XSG_ENOTSUP, "'%s' lacks the expected RCCL symbols", loaded.c_str());
  }
  g_rccl = r;
  return XSG_OK;
}

}  // namespace

#define NCCL_TRY(expr)                                                                                          \
  do {                                                                                                          \
    ncclResult_t _r = (expr);                                                                                   \
    if (_r != ncclSuccess)                                                                                      \
      return xsg::fail(XSG_EHIP, "%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(_r), __FILE__, __LINE__); \
  } while (0)

// One communicator handle = the local end(s) of one RCCL clique.
//   rank form  : one process per GPU; `comms` has one entry (this process's rank)
//   local form : one process drives n GPUs; `comms[i]` belongs to ctxs[i] (ncclCommInitAll)
struct xsg_comm {
  std::vector<ncclComm_t> comms;
  std::vector<xsg_ctx*> ctxs;
  std::vector<int> devices;          // the ctxs' devices, kept by value: xsg_comm_destroy must not look into a ctx that
                                     // the caller may already have destroyed
  std::vector<uint64_t*> d_scratch;  // per local device: 64 uint64 (gather buffers, staged totals)
  int nranks = 0, rank = 0;
  bool local = false;
};

extern "C" int xsg_comm_unique_id(void* id, size_t cap) {
  if (!id || cap < XSG_COMM_ID_BYTES) return fail(XSG_EINVAL, "id buffer must hold %d bytes", XSG_COMM_ID_BYTES);
  static_assert(XSG_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
  XSG_TRY(need_rccl());
  ncclUniqueId u;
  NCCL_TRY(g_rccl.GetUniqueId(&u));
  memcpy(id, &u, sizeof u);
  return XSG_OK;
}

extern "C" const char* xsg_comm_library(void) {
  if (need_rccl() != XSG_OK) return "";
  return g_rccl.path.c_str();
}

static int comm_scratch(xsg_comm* c) {
  c->devices.clear();
  for (xsg_ctx* x : c->ctxs) c->devices.push_back(x->device);
  for (xsg_ctx* x : c->ctxs) {
    HIP_TRY(hipSetDevice(x->device));
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, 8 * 64 * 2));
    c->d_scratch.push_back(static_cast<uint64_t*>(p));
  }
  return XSG_OK;
}

extern "C" void xsg_comm_destroy(xsg_comm* c) {
  if (!c) return;
  for (size_t i = 0; i < c->comms.size(); ++i) {
    if (i < c->devices.size()) (void)hipSetDevice(c->devices[i]);
    if (c->comms[i]) (void)g_rccl.CommDestroy(c->comms[i]);
  }
  for (size_t i = 0; i < c->d_scratch.size(); ++i) {
    if (i < c->devices.size()) (void)hipSetDevice(c->devices[i]);
    (void)hipFree(c->d_scratch[i]);
  }
  delete c;
}

extern "C" int xsg_comm_create_rank(xsg_ctx* ctx, int nranks, int rank, const void* id, xsg_comm** out) {
  if (!ctx || !out || !id) return fail(XSG_EINVAL, "null argument");
  *out = nullptr;
  if (nranks < 1 || rank < 0 || rank >= nranks) return fail(XSG_EINVAL, "bad rank %d of %d", rank, nranks);
  XSG_TRY(need_rccl());
  HIP_TRY(hipSetDevice(ctx->device));
  xsg_comm* c = new (std::nothrow) xsg_comm();
  if (!c) return fail(XSG_ENOMEM, "host allocation failed");
  c->nranks = nranks;
  c->rank = rank;
  c->ctxs.push_back(ctx);
  ncclUniqueId u;
  memcpy(&u, id, sizeof u);
  ncclComm_t cm = nullptr;
  ncclResult_t r = g_rccl.CommInitRank(&cm, nranks, u, rank);
  if (r != ncclSuccess) {
    delete c;
    return fail(XSG_EHIP, "ncclCommInitRank(rank %d of %d) failed: %s", rank, nranks, g_rccl.GetErrorString(r));
  }
  c->comms.push_back(cm);
  const int rc = comm_scratch(c);
  if (rc != XSG_OK) {
    xsg_comm_destroy(c);
    return rc;
  }
  *out = c;
  return XSG_OK;
}

extern "C" int xsg_comm_create_local(xsg_ctx* const* ctxs, int n, xsg_comm** out) {
  if (!ctxs || !out || n < 1) return fail(XSG_EINVAL, "bad argument");
  *out = nullptr;
  XSG_TRY(need_rccl());
  std::vector<int> devs;
  for (int i = 0; i < n; ++i) {
    if (!ctxs[i]) return fail(XSG_EINVAL, "ctx %d is null", i);
    for (int d : devs)
      if (d == ctxs[i]->device)  // RCCL refuses two ranks on one device; the caller sums those on the host
        return fail(XSG_ENOTSUP, "device %d is listed twice: one RCCL rank per GPU", d);
    devs.push_back(ctxs[i]->device);
  }
  xsg_comm* c = new (std::nothrow) xsg_comm();
  if (!c) return fail(XSG_ENOMEM, "host allocation failed");
  c->nranks = n;
  c->local = true;
  c->ctxs.assign(ctxs, ctxs + n);
  c->comms.assign((size_t)n, nullptr);
  ncclResult_t r = g_rccl.CommInitAll(c->comms.data(), n, devs.data());
  if (r != ncclSuccess) {
    c->comms.clear();
    delete c;
    return fail(XSG_EHIP, "ncclCommInitAll(%d devices) failed: %s", n, g_rccl.GetErrorString(r));
  }
  const int rc = comm_scratch(c);
  if (rc != XSG_OK) {
    xsg_comm_destroy(c);
    return rc;
  }
  *out = c;
  return XSG_OK;
}

extern "C" int xsg_comm_size(xsg_comm* c, int* nranks, int* rank) {
  if (!c) return fail(XSG_EINVAL, "comm is null");
  if (nranks) *nranks = c->nranks;
  if (rank) *rank = c->local ? -1 : c->rank;
  return XSG_OK;
}

// rank form: in-place sum of k uint64 device words across the ranks, enqueued on `stream` (NULL: the ctx's own)
extern "C" int xsg_reduce_counts_async(xsg_comm* c, uint64_t* d_counters, int k, void* stream) {
  if (!c || !d_counters || k < 1) return fail(XSG_EINVAL, "bad argument");
  if (c->local) return fail(XSG_ESTATE, "xsg_reduce_counts_async needs a communicator from xsg_comm_create_rank");
  HIP_TRY(hipSetDevice(c->ctxs[0]->device));
  hipStream_t st = stream ? static_cast<hipStream_t>(stream) : c->ctxs[0]->stream;
  NCCL_TRY(g_rccl.AllReduce(d_counters, d_counters, (size_t)k, ncclUint64, ncclSum, c->comms[0], st));
  return XSG_OK;
}

// Either form, blocking: totals[0..k) = sum over every rank/device of its k device words (left in place, summed).
//   rank form : d_counters[0] is this rank's vector
//   local form: d_counters[i] is the vector on ctxs[i]'s device
extern "C" int xsg_reduce_counts(xsg_comm* c, uint64_t* const* d_counters, int k, uint64_t* totals) {
  if (!c || !d_counters || !totals || k < 1 || k > 64) return fail(XSG_EINVAL, "bad argument");
  const size_t n = c->comms.size();
  NCCL_TRY(g_rccl.GroupStart());
  for (size_t i = 0; i < n; ++i) {
    if (!d_counters[i]) {
      (void)g_rccl.GroupEnd();
      return fail(XSG_EINVAL, "d_counters[%zu] is null", i);
    }
    (void)hipSetDevice(c->ctxs[i]->device);
    const ncclResult_t r = g_rccl.AllReduce(d_counters[i], d_counters[i], (size_t)k, ncclUint64, ncclSum, c->comms[i],
                                            c->ctxs[i]->stream);
    if (r != ncclSuccess) {
      (void)g_rccl.GroupEnd();
      return fail(XSG_EHIP, "ncclAllReduce failed: %s", g_rccl.GetErrorString(r));
    }
  }
  NCCL_TRY(g_rccl.GroupEnd());
  HIP_TRY(hipSetDevice(c->ctxs[0]->device));
  HIP_TRY(hipMemcpyAsync(totals, d_counters[0], 8 * (size_t)k, hipMemcpyDeviceToHost, c->ctxs[0]->stream));
  for (size_t i = 0; i < n; ++i) {
    HIP_TRY(hipSetDevice(c->ctxs[i]->device));
    HIP_TRY(hipStreamSynchronize(c->ctxs[i]->stream));
  }
  return XSG_OK;
}

// Line-index bases (xs::line_indices without a metafile): every rank/device contributes the number of
// '\n' in its chunk range; out receives all nranks values in rank order (the exclusive prefix is the caller's).
//   rank form : mine[0] is this rank's value;  local form: mine[i] belongs to ctxs[i]
extern "C" int xsg_allgather_u64(xsg_comm* c, const uint64_t* mine, uint64_t* out) {
  if (!c || !mine || !out) return fail(XSG_EINVAL, "null argument");
  if (c->nranks > 64) return fail(XSG_ENOTSUP, "more than 64 ranks");
  const size_t n = c->comms.size();
  for (size_t i = 0; i < n; ++i) {
    HIP_TRY(hipSetDevice(c->ctxs[i]->device));
    HIP_TRY(hipMemcpyAsync(c->d_scratch[i], mine + i, 8, hipMemcpyHostToDevice, c->ctxs[i]->stream));
  }
  NCCL_TRY(g_rccl.GroupStart());
  for (size_t i = 0; i < n; ++i) {
    (void)hipSetDevice(c->ctxs[i]->device);
    const ncclResult_t r =
        g_rccl.AllGather(c->d_scratch[i], c->d_scratch[i] + 64, 1, ncclUint64, c->comms[i], c->ctxs[i]->stream);
    if (r != ncclSuccess) {
      (void)g_rccl.GroupEnd();
      return fail(XSG_EHIP, "ncclAllGather failed: %s", g_rccl.GetErrorString(r));
    }
  }
  NCCL_TRY(g_rccl.GroupEnd());
  HIP_TRY(hipSetDevice(c->ctxs[0]->device));
  HIP_TRY(hipMemcpyAsync(out, c->d_scratch[0] + 64, 8 * (size_t)c->nranks, hipMemcpyDeviceToHost, c->ctxs[0]->stream));
  for (size_t i = 0; i < n; ++i) {
    HIP_TRY(hipSetDevice(c->ctxs[i]->device));
    HIP_TRY(hipStreamSynchronize(c->ctxs[i]->stream));
  }
  return XSG_OK;
}
