// rccl_bcast.hip.h -- the one exchange step of the multi-GPU mode (BASELINE configs[4]): the shared flat-L2
// retrieval index is broadcast from rank 0 to every rank's HBM over RCCL / xGMI once at load.  Streams shard with no per-chunk
// collective, so this is the only collective of the whole engine.  It sits behind the C ABI so that a Rust (or C) host needs no
// Python: librccl is resolved lazily with dlopen -- single-GPU users never load it, and inside a process that already carries a
// RCCL (PyTorch) the same library instance is reused through its soname.
//
// The reference has no counterpart (one RvcInfer per process, rvc/src/rvc.rs:133-134; index search is a TODO at rvc.rs:159).
#pragma once
#include <dlfcn.h>
#include <chrono>
#include <mutex>

namespace rvc {

struct RcclUid { char internal[128]; };                 // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128), passed by value
typedef void *RcclComm;
struct RcclApi {
    void *lib = nullptr;
    int (*GetUniqueId)(RcclUid *) = nullptr;
    int (*CommInitRank)(RcclComm *, int, RcclUid, int) = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, RcclComm, hipStream_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, RcclComm, hipStream_t) = nullptr;
    int (*CommCount)(RcclComm, int *) = nullptr;
    int (*CommDestroy)(RcclComm) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string err;
};

// resolved once per process (std::call_once: a second engine on another thread never sees a half-filled table)
static RcclApi &rccl_api()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {getenv("RVC_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        void *lib = nullptr;
        for (const char *n : names) {
            if (!n || !*n) continue;
            lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib) { api.err = std::string("cannot load librccl: ") + (dlerror() ? dlerror() : "not found"); return; }
        api.GetUniqueId = (int (*)(RcclUid *))dlsym(lib, "ncclGetUniqueId");
        api.CommInitRank = (int (*)(RcclComm *, int, RcclUid, int))dlsym(lib, "ncclCommInitRank");
        api.Broadcast = (int (*)(const void *, void *, size_t, int, int, RcclComm, hipStream_t))dlsym(lib, "ncclBroadcast");
        api.AllReduce = (int (*)(const void *, void *, size_t, int, int, RcclComm, hipStream_t))dlsym(lib, "ncclAllReduce");
        api.CommCount = (int (*)(RcclComm, int *))dlsym(lib, "ncclCommCount");
        api.CommDestroy = (int (*)(RcclComm))dlsym(lib, "ncclCommDestroy");
        api.GetErrorString = (const char *(*)(int))dlsym(lib, "ncclGetErrorString");
        if (!api.GetUniqueId || !api.CommInitRank || !api.Broadcast || !api.AllReduce || !api.CommCount || !api.CommDestroy) { api.err = "librccl lacks the ncclBroadcast entry points"; return; }
        api.lib = lib;                  // published last
    });
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

// RVC_OK when librccl can be loaded in this process (no communicator is created): hosts call it on every rank and agree on the result
// BEFORE the first collective, so that a rank without the library does not leave the others waiting in ncclCommInitRank.
rvc_status rvc_rccl_available(void)
{
    RcclApi &api = rccl_api();
    if (!api.lib) { fprintf(stderr, "rvc_rccl_available: %s\n", api.err.c_str()); return RVC_BACKEND; }
    return RVC_OK;
}

rvc_status rvc_index_broadcast(rvc_engine *e, const void *unique_id128, int rank, int world, const float *vectors, size_t n, size_t dim)
{
    return guarded(e, [&]() {
        // Only what every rank can check about ITSELF and what a host agrees on beforehand (rvc_rccl_available) may throw before the
        // communicator exists.  Rank 0's own preconditions (vectors present, at least 4 of them) do NOT throw here: the other ranks are
        // already on their way into ncclCommInitRank and would wait for a peer that left.  A failing rank 0 sends an empty
        // header instead; every rank then rejects it in the agreement all-reduce and all of them return the error together.
        if (world < 1 || rank < 0 || rank >= world || !unique_id128) throw ShapeError("index broadcast: bad rank / world / unique id");
        std::string root_err;
        if (rank == 0 && !vectors && !e->d_index) root_err = "index broadcast: rank 0 has neither host vectors nor a loaded index";
        else if (rank == 0 && vectors && (n < KNN_K || dim < 1)) root_err = "index broadcast: index needs at least 4 vectors";
        RcclApi &api = rccl_api();
        if (!api.lib) throw std::runtime_error(api.err);
        HIPCHK(hipDeviceSynchronize());
        typedef std::chrono::steady_clock clk;
        const auto t0 = clk::now();
        RcclUid uid; memcpy(uid.internal, unique_id128, sizeof uid.internal);
        RcclComm comm = nullptr;
        RCCLCHK(api, api.CommInitRank(&comm, world, uid, rank));
        int ranks = 0;
        (void)api.CommCount(comm, &ranks);
        const auto t1 = clk::now();
        float *d_new = nullptr; unsigned long long *d_hdr = nullptr;
        std::string local_err;
        try {
            // header first: the other ranks learn (n, dim) from rank 0
            unsigned long long hdr[2] = {0, 0};
            if (rank == 0 && root_err.empty()) { hdr[0] = vectors ? n : e->index_n; hdr[1] = vectors ? dim : e->index_dim; }
            HIPCHK(hipMalloc(&d_hdr, 4 * sizeof(unsigned long long)));
            HIPCHK(hipMemcpy(d_hdr, hdr, sizeof hdr, hipMemcpyHostToDevice));
            RCCLCHK(api, api.Broadcast(d_hdr, d_hdr, sizeof hdr, /*ncclUint8*/ 1, 0, comm, e->stream));
            HIPCHK(hipStreamSynchronize(e->stream));
            HIPCHK(hipMemcpy(hdr, d_hdr, sizeof hdr, hipMemcpyDeviceToHost));
            const size_t bn = (size_t)hdr[0], bd = (size_t)hdr[1];
            // local verdict on the header (shape, memory), then ONE all-reduce of it: every rank leaves together or goes on together --
            // a rank that simply threw here would leave the others blocked in the payload broadcast
            int ok = 1;
            if (!root_err.empty()) { ok = 0; local_err = root_err; }
            else if (bn < KNN_K || bd < 1 || bn * bd > ((size_t)1 << 36)) { ok = 0; local_err = "index broadcast: rank 0 sent no usable index (its arguments were rejected there, or the size is implausible)"; }
            else if (rank != 0 && n && dim && (n != bn || dim != bd)) { ok = 0; local_err = "index broadcast: this rank expected a different index shape than rank 0 sent"; }
            else if (hipMalloc(&d_new, bn * bd * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); d_new = nullptr; ok = 0; local_err = "index broadcast: out of device memory for the index"; }
            int *d_ok = reinterpret_cast<int *>(d_hdr + 2);
            HIPCHK(hipMemcpy(d_ok, &ok, sizeof ok, hipMemcpyHostToDevice));
            RCCLCHK(api, api.AllReduce(d_ok, d_ok, 1, /*ncclInt32*/ 2, /*ncclMin*/ 3, comm, e->stream));
            HIPCHK(hipStreamSynchronize(e->stream));
            int all_ok = 0;
            HIPCHK(hipMemcpy(&all_ok, d_ok, sizeof all_ok, hipMemcpyDeviceToHost));
            if (!all_ok) throw ShapeError(local_err.empty() ? "index broadcast: another rank rejected the index" : local_err);
            if (rank == 0) {
                if (vectors) HIPCHK(hipMemcpy(d_new, vectors, bn * bd * sizeof(float), hipMemcpyHostToDevice));
                else HIPCHK(hipMemcpy(d_new, e->d_index, bn * bd * sizeof(float), hipMemcpyDeviceToDevice));
            }
            // one ncclBroadcast of the whole matrix: 307 MB for 100k x 768; over the xGMI mesh the root feeds its peers on distinct links
            RCCLCHK(api, api.Broadcast(d_new, d_new, bn * bd, /*ncclFloat32*/ 7, 0, comm, e->stream));
            HIPCHK(hipStreamSynchronize(e->stream));
            const auto t2 = clk::now();
            if (e->d_index && e->index_owned) (void)hipFree(e->d_index);
            e->d_index = d_new; d_new = nullptr; e->index_owned = true;
            e->index_n = bn; e->index_dim = bd;
            build_index_aux(e);                       // fragment-order copy + norms, on the device
            e->plans.clear(); e->last_plan = nullptr;
            const auto t3 = clk::now();
            auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
            e->bcast_ms[0] = ms(t0, t1); e->bcast_ms[1] = ms(t1, t2); e->bcast_ms[2] = ms(t2, t3); e->bcast_ranks = ranks;
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

// the last rvc_index_broadcast of this engine: ms[0] communicator set-up (ncclCommInitRank), ms[1] header + agreement + payload
// broadcast (incl. rank 0's upload), ms[2] device-side repack (fragment order + norms); *ranks = ncclCommCount of the communicator
rvc_status rvc_index_broadcast_info(rvc_engine *e, double ms[3], int *ranks)
{
    if (!e) return RVC_BACKEND;
    if (ms) for (int i = 0; i < 3; i++) ms[i] = e->bcast_ms[i];
    if (ranks) *ranks = e->bcast_ranks;
    return RVC_OK;
}

}  // extern "C"
