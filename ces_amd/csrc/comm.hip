// The exchange step of a sharded ensemble behind the C ABI (include/cesx.h, cesx_comm_* / cesx_allreduce_*): RCCL
// all-reduces of the packed fp64 moment buffer over xGMI, issued on the stream the caller names (the engine's own
// streams in ces_amd/dist.py: the head on the side stream in front of chol(C), the tail on the caller's stream).
// A C caller -- or the reference bound as in INTEGRATION.md -- runs a sharded ensemble with these entry points alone;
// nothing here needs torch.distributed.  (one ncclAllReduce(sum) of the packed buffer per step; the
// default splits it in head + tail of the same total payload)
//
// librccl is bound at run time (dlopen / dlsym): an engine that never shards does not load it, and a process that
// already holds a copy (PyTorch ships its own) keeps using that one -- RTLD_NOLOAD first.
#include "cesx_internal.h"
#include <dlfcn.h>
#include <cstdlib>
#include <cstring>

namespace {

struct Rccl {
    // (the subset of rccl.h this file needs; enum values and the 128-byte id are part of NCCL's stable ABI)
    typedef int result_t;
    typedef void* comm_t;
    struct unique_id { char internal[CESX_COMM_ID_BYTES]; };
    result_t (*GetUniqueId)(unique_id*) = nullptr;
    result_t (*CommInitRank)(comm_t*, int, unique_id, int) = nullptr;
    result_t (*CommDestroy)(comm_t) = nullptr;
    result_t (*CommSplit)(comm_t, int, int, comm_t*, void*) = nullptr;      // optional (NCCL >= 2.18): a second communicator without a second id
    result_t (*AllReduce)(const void*, void*, size_t, int, int, comm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(result_t) = nullptr;
    void* lib = nullptr;
    std::string err;
    bool ok = false;
};

void rccl_load(Rccl& r);

// (bound once per process, whichever thread asks first: a function-local static's initialiser runs under the
//  language's own lock)
Rccl& rccl() {
    static Rccl r = [] { Rccl q; rccl_load(q); return q; }();
    return r;
}

void rccl_load(Rccl& r) {
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* nm : names) { r.lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD); if (r.lib) break; }
    if (!r.lib)
        for (const char* nm : names) { r.lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL); if (r.lib) break; }
    if (!r.lib) { const char* de = dlerror(); r.err = std::string("librccl.so not found (dlopen") + (de ? std::string(": ") + de : std::string()) + ")"; return; }
    auto sym = [&](const char* nm) { void* p = dlsym(r.lib, nm); if (!p && r.err.empty()) r.err = std::string("librccl: missing symbol ") + nm; return p; };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    r.ok = r.err.empty();
    r.CommSplit = reinterpret_cast<decltype(r.CommSplit)>(dlsym(r.lib, "ncclCommSplit"));
}

constexpr int NCCL_DOUBLE = 8, NCCL_SUM = 0, NCCL_MAX = 2;      // ncclDataType_t / ncclRedOp_t (rccl.h)

int fail(cesx::Engine& e, const char* what, int res) {
    Rccl& r = rccl();
    e.err = std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(res) : "RCCL error");
    return CESX_ERCCL;
}

int all_reduce(cesx_handle h, double* buf, size_t count, int op, void* stream, const char* who) {
    if (!h) return CESX_EINVAL;
    cesx::Engine& e = *reinterpret_cast<cesx::Engine*>(h);
    if (!buf) { e.err = std::string(who) + ": null pointer"; return CESX_EINVAL; }
    if (!e.comm) { e.err = std::string(who) + ": cesx_comm_init has not been called"; return CESX_ESTATE; }
    if (count == 0) return CESX_OK;
    int prev = -1;
    if (hipGetDevice(&prev) == hipSuccess && prev != e.cfg.device) (void)hipSetDevice(e.cfg.device);
    // One communicator per stream: a collective issued on the engine's side stream (the head of the moment buffer, beside the
    // second Gram launch) goes through the side communicator, everything else through the main one.  RCCL orders the
    // collectives of ONE communicator -- issued from two streams they are chained by its own cross-stream events, the head
    // of step i + 1 behind the tail of step i -- and two ranks that interleave the two streams differently would wait for
    // each other; two communicators are independent of each other by construction.
    Rccl::comm_t c = ((hipStream_t)stream == e.side && e.comm_side) ? e.comm_side : e.comm;
    const int res = rccl().AllReduce(buf, buf, count, NCCL_DOUBLE, op, c, (hipStream_t)stream);
    if (prev >= 0 && prev != e.cfg.device) (void)hipSetDevice(prev);
    ++e.comm_calls;
    e.comm_doubles += count;
    return res == 0 ? CESX_OK : fail(e, who, res);
}

}  // namespace

extern "C" {

int cesx_comm_unique_id(void* id_out) {
    if (!id_out) return CESX_EINVAL;
    Rccl& r = rccl();
    // (no handle to hang the message on: cesx_last_error(NULL) returns it, as for cesx_create)
    if (!r.ok) { cesx::set_global_error("cesx_comm_unique_id: " + r.err); return CESX_ERCCL; }
    Rccl::unique_id id;
    const int res = r.GetUniqueId(&id);
    if (res != 0) {
        cesx::set_global_error(std::string("ncclGetUniqueId: ") + (r.GetErrorString ? r.GetErrorString(res) : "RCCL error"));
        return CESX_ERCCL;
    }
    std::memcpy(id_out, id.internal, CESX_COMM_ID_BYTES);
    return CESX_OK;
}

int cesx_comm_init(cesx_handle h, int nranks, int rank, const void* unique_id) {
    if (!h) return CESX_EINVAL;
    cesx::Engine& e = *reinterpret_cast<cesx::Engine*>(h);
    if (!unique_id || nranks < 1 || rank < 0 || rank >= nranks) { e.err = "cesx_comm_init: bad argument"; return CESX_EINVAL; }
    if (e.comm) { e.err = "cesx_comm_init: this handle already has a communicator (cesx_comm_destroy first)"; return CESX_ESTATE; }
    Rccl& r = rccl();
    if (!r.ok) { e.err = "cesx_comm_init: " + r.err; return CESX_ERCCL; }
    int prev = -1;
    if (hipGetDevice(&prev) == hipSuccess && prev != e.cfg.device) (void)hipSetDevice(e.cfg.device);
    Rccl::unique_id id;
    std::memcpy(id.internal, unique_id, CESX_COMM_ID_BYTES);
    Rccl::comm_t c = nullptr;
    const int res = r.CommInitRank(&c, nranks, id, rank);
    // the side stream's communicator: split off the first one (the same ranks, the same order; a collective call, every rank
    // makes it here).  CESX_COMM_SPLIT=0, or a library without ncclCommSplit: both streams share the one communicator
    Rccl::comm_t c2 = nullptr;
    int res2 = 0;
    const char* sv = std::getenv("CESX_COMM_SPLIT");
    if (res == 0 && r.CommSplit && !(sv && sv[0] == '0')) res2 = r.CommSplit(c, 0, rank, &c2, nullptr);
    if (prev >= 0 && prev != e.cfg.device) (void)hipSetDevice(prev);
    if (res != 0) return fail(e, "ncclCommInitRank", res);
    if (res2 != 0) c2 = nullptr;      // (ncclCommSplit is collective: it fails on every rank or on none -- all then share the one communicator; cesx_comm_count says so)
    e.comm = c; e.comm_side = c2; e.comm_nranks = nranks; e.comm_rank = rank;
    return CESX_OK;
}

int cesx_comm_destroy(cesx_handle h) {
    if (!h) return CESX_EINVAL;
    cesx::Engine& e = *reinterpret_cast<cesx::Engine*>(h);
    if (!e.comm) return CESX_OK;
    if (e.comm_side) (void)rccl().CommDestroy(e.comm_side);
    const int res = rccl().CommDestroy(e.comm);
    e.comm = nullptr; e.comm_side = nullptr; e.comm_nranks = 0; e.comm_rank = 0;
    return res == 0 ? CESX_OK : fail(e, "ncclCommDestroy", res);
}

int cesx_comm_nranks(cesx_handle h) { return h ? reinterpret_cast<cesx::Engine*>(h)->comm_nranks : 0; }

int cesx_comm_count(cesx_handle h) {
    if (!h) return 0;
    const cesx::Engine& e = *reinterpret_cast<cesx::Engine*>(h);
    return (e.comm ? 1 : 0) + (e.comm_side ? 1 : 0);
}

int cesx_comm_stats(cesx_handle h, unsigned long long* calls, unsigned long long* doubles) {
    if (!h) return CESX_EINVAL;
    cesx::Engine& e = *reinterpret_cast<cesx::Engine*>(h);
    if (calls) *calls = e.comm_calls;
    if (doubles) *doubles = e.comm_doubles;
    return CESX_OK;
}

int cesx_allreduce_head(cesx_handle h, double* mom, void* stream) {
    if (!h) return CESX_EINVAL;
    return all_reduce(h, mom, reinterpret_cast<cesx::Engine*>(h)->ml.uu_len(), NCCL_SUM, stream, "cesx_allreduce_head");
}

int cesx_allreduce_tail(cesx_handle h, double* mom, void* stream) {
    if (!h) return CESX_EINVAL;
    cesx::Engine& e = *reinterpret_cast<cesx::Engine*>(h);
    if (!mom) { e.err = "cesx_allreduce_tail: null pointer"; return CESX_EINVAL; }
    return all_reduce(h, mom + e.ml.uu_len(), e.ml.len() - e.ml.uu_len(), NCCL_SUM, stream, "cesx_allreduce_tail");
}

int cesx_allreduce_whole(cesx_handle h, double* mom, void* stream) {
    if (!h) return CESX_EINVAL;
    return all_reduce(h, mom, reinterpret_cast<cesx::Engine*>(h)->ml.len(), NCCL_SUM, stream, "cesx_allreduce_whole");
}

int cesx_allreduce_sum(cesx_handle h, double* buf, size_t count, void* stream) {
    return all_reduce(h, buf, count, NCCL_SUM, stream, "cesx_allreduce_sum");
}

int cesx_allreduce_max(cesx_handle h, double* buf, size_t count, void* stream) {
    return all_reduce(h, buf, count, NCCL_MAX, stream, "cesx_allreduce_max");
}

}  // extern "C"
