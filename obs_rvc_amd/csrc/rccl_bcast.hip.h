// rccl_bcast.hip.h -- the one exchange step of the multi-GPU mode (BASELINE configs[4]): the shared flat-L2
// retrieval index is broadcast from rank 0 to every rank's HBM over RCCL / xGMI once at load.  Streams shard with no per-chunk
// collective, so this is the only collective of the whole engine.  It sits behind the C ABI so that a Rust (or C) host needs no
// Python: librccl is resolved lazily with dlopen -- single-GPU users never load it, and inside a process that already carries a
// RCCL (PyTorch) the same library instance is reused through its soname.
//
// The reference has no counterpart (one RvcInfer per process, rvc/src/rvc.rs:133-134; index search is a TODO at rvc.rs:159).
#pragma once
#include <dlfcn.h>

namespace rvc {

struct RcclUid { char internal[128]; };                 // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128), passed by value
typedef void *RcclComm;
struct RcclApi {
    void *lib = nullptr;
    int (*GetUniqueId)(RcclUid *) = nullptr;
    int (*CommInitRank)(RcclComm *, int, RcclUid, int) = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, RcclComm, hipStream_t) = nullptr;
    int (*CommDestroy)(RcclComm) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string err;
};

static RcclApi &rccl_api()
{
    static RcclApi api;
    if (api.lib || !api.err.empty()) return api;
    const char *names[] = {getenv("RVC_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
        if (!n || !*n) continue;
        api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (api.lib) break;
    }
    if (!api.lib) { api.err = std::string("cannot load librccl: ") + (dlerror() ? dlerror() : "not found"); return api; }
    api.GetUniqueId = (int (*)(RcclUid *))dlsym(api.lib, "ncclGetUniqueId");
    api.CommInitRank = (int (*)(RcclComm *, int, RcclUid, int))dlsym(api.lib, "ncclCommInitRank");
    api.Broadcast = (int (*)(const void *, void *, size_t, int, int, RcclComm, hipStream_t))dlsym(api.lib, "ncclBroadcast");
    api.CommDestroy = (int (*)(RcclComm))dlsym(api.lib, "ncclCommDestroy");
    api.GetErrorString = (const char *(*)(int))dlsym(api.lib, "ncclGetErrorString");
    if (!api.GetUniqueId || !api.CommInitRank || !api.Broadcast || !api.CommDestroy) { api.err = "librccl lacks the ncclBroadcast entry points"; api.lib = nullptr; }
    return api;
}

#define RCCLCHK(api, expr)                                                                                                   \
    do {                                                                                                                     \
        int r_ = (expr);                                                                                                     \
        if (r_ != 0) throw std::runtime_error(std::string(#expr) + ": " + ((api).GetErrorString ? (api).GetErrorString(r_) : "rccl error")); \
    } while (0)

}  // namespace rvc

extern "C" {

rvc_status rvc_rccl_unique_id(void *id128)
{
    if (!id128) return RVC_SHAPE;
    RcclApi &api = rccl_api();
    if (!api.lib) { fprintf(stderr, "rvc_rccl_unique_id: %s\n", api.err.c_str()); return RVC_BACKEND; }
    RcclUid u;
    if (api.GetUniqueId(&u) != 0) return RVC_BACKEND;
    memcpy(id128, u.internal, sizeof u.internal);
    return RVC_OK;
}

rvc_status rvc_index_broadcast(rvc_engine *e, const void *unique_id128, int rank, int world, const float *vectors, size_t n, size_t dim)
{
    return guarded(e, [&]() {
        if (world < 1 || rank < 0 || rank >= world || !unique_id128) throw ShapeError("index broadcast: bad rank / world / unique id");
        if (rank == 0 && !vectors && !e->d_index) throw ShapeError("index broadcast: rank 0 has neither host vectors nor a loaded index");
        RcclApi &api = rccl_api();
        if (!api.lib) throw std::runtime_error(api.err);
        HIPCHK(hipDeviceSynchronize());
        RcclUid uid; memcpy(uid.internal, unique_id128, sizeof uid.internal);
        RcclComm comm = nullptr;
        RCCLCHK(api, api.CommInitRank(&comm, world, uid, rank));
        float *d_new = nullptr; unsigned long long *d_hdr = nullptr;
        try {
            // header first: the other ranks learn (n, dim) from rank 0
            unsigned long long hdr[2] = {0, 0};
            if (rank == 0) { hdr[0] = vectors ? n : e->index_n; hdr[1] = vectors ? dim : e->index_dim; }
            HIPCHK(hipMalloc(&d_hdr, sizeof hdr));
            HIPCHK(hipMemcpy(d_hdr, hdr, sizeof hdr, hipMemcpyHostToDevice));
            RCCLCHK(api, api.Broadcast(d_hdr, d_hdr, sizeof hdr, /*ncclUint8*/ 1, 0, comm, e->stream));
            HIPCHK(hipStreamSynchronize(e->stream));
            HIPCHK(hipMemcpy(hdr, d_hdr, sizeof hdr, hipMemcpyDeviceToHost));
            const size_t bn = (size_t)hdr[0], bd = (size_t)hdr[1];
            if (bn < KNN_K || bd < 1 || bn * bd > ((size_t)1 << 36)) throw ShapeError("index broadcast: implausible index size from rank 0");
            if (rank != 0 && n && dim && (n != bn || dim != bd)) throw ShapeError("index broadcast: this rank expected a different index shape than rank 0 sent");
            HIPCHK(hipMalloc(&d_new, bn * bd * sizeof(float)));
            if (rank == 0) {
                if (vectors) HIPCHK(hipMemcpy(d_new, vectors, bn * bd * sizeof(float), hipMemcpyHostToDevice));
                else HIPCHK(hipMemcpy(d_new, e->d_index, bn * bd * sizeof(float), hipMemcpyDeviceToDevice));
            }
            // one ncclBroadcast of the whole matrix: 307 MB for 100k x 768; over the xGMI mesh the root feeds its peers on distinct links
            RCCLCHK(api, api.Broadcast(d_new, d_new, bn * bd, /*ncclFloat32*/ 7, 0, comm, e->stream));
            HIPCHK(hipStreamSynchronize(e->stream));
            if (e->d_index && e->index_owned) (void)hipFree(e->d_index);
            e->d_index = d_new; d_new = nullptr; e->index_owned = true;
            e->index_n = bn; e->index_dim = bd;
            build_index_transpose(e);
            e->plans.clear(); e->last_plan = nullptr;
        } catch (...) {
            if (d_new) (void)hipFree(d_new);
            if (d_hdr) (void)hipFree(d_hdr);
            (void)api.CommDestroy(comm);
            throw;
        }
        (void)hipFree(d_hdr);
        RCCLCHK(api, api.CommDestroy(comm));
        return RVC_OK;
    });
}

}  // extern "C"
