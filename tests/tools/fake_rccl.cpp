// fake_rccl.cpp -- TEST INFRASTRUCTURE, never a default: a stand-in for the six nccl* entry points rvc_index_broadcast resolves
// (obs_rvc_amd/csrc/rccl_bcast.hip.h), so that the multi-rank C-ABI path -- unique id hand-over, communicator set-up, header broadcast,
// agreement all-reduce, payload broadcast, the non-root receive side and the fail-together logic -- can run with world = 2 on a box
// that has ONE GPU (the code had never executed on a rank != 0 anywhere).  Selected only through RVC_RCCL_LIB=<this .so>
// by tests/test_gpu_multi.py; the product never names it.
//
// Transport: the ranks are separate processes on the same GPU; they meet in a sparse file under /dev/shm (or $TMPDIR) named by the
// unique id.  A collective = root's device buffer -> hipMemcpy D2H into the shared mapping -> process barrier -> the other ranks'
// hipMemcpy H2D.  Every call synchronises the stream it is given first (the real library is stream-ordered; the engine synchronises
// right behind each collective anyway).  Barriers time out (FAKE_RCCL_TIMEOUT_S, default 60 s) and return an error instead of hanging.
//
//   g++ -O2 -fPIC -shared -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include fake_rccl.cpp -o libfakerccl.so -L/opt/rocm/lib -lamdhip64
#include <hip/hip_runtime_api.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>

namespace {

struct Uid { char internal[128]; };
constexpr size_t kCap = (size_t)1 << 30;           // payload window (sparse: only touched pages exist)
constexpr unsigned kMagic = 0x46524343u;           // "FRCC"

struct Shared {
    std::atomic<unsigned> magic;                   // set by rank 0 when the file is ready
    std::atomic<unsigned> arrived;                 // barrier: arrivals of the current generation
    std::atomic<unsigned> generation;
    std::atomic<unsigned> joined, left;
    int slots[64];                                 // all-reduce operands
    char pad[4096 - 5 * sizeof(unsigned) - 64 * sizeof(int)];
    char payload[1];
};
struct Comm { Shared *sh; int rank, world; std::string path; };

double timeout_s() { const char *t = getenv("FAKE_RCCL_TIMEOUT_S"); return t ? atof(t) : 60.0; }

std::string path_of(const Uid &u)
{
    char tok[33]; memcpy(tok, u.internal, 32); tok[32] = 0;
    const char *dir = access("/dev/shm", W_OK) == 0 ? "/dev/shm" : (getenv("TMPDIR") ? getenv("TMPDIR") : "/tmp");
    return std::string(dir) + "/fakerccl_" + tok;
}

// sense-reversing barrier over the shared mapping; false = timed out (a peer died or never came)
bool barrier(Comm *c)
{
    Shared *s = c->sh;
    const unsigned gen = s->generation.load(std::memory_order_acquire);
    if (s->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (unsigned)c->world) {
        s->arrived.store(0, std::memory_order_relaxed);
        s->generation.store(gen + 1, std::memory_order_release);
        return true;
    }
    const auto t0 = std::chrono::steady_clock::now();
    while (s->generation.load(std::memory_order_acquire) == gen) {
        std::this_thread::sleep_for(std::chrono::microseconds(50));
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s()) return false;
    }
    return true;
}

size_t type_bytes(int dt) { switch (dt) { case 0: case 1: return 1; case 2: case 3: case 7: return 4; case 4: case 5: case 8: return 8; case 6: case 9: return 2; default: return 0; } }

}  // namespace

extern "C" {

int ncclGetUniqueId(Uid *out)
{
    if (!out) return 4;
    memset(out, 0, sizeof *out);
    unsigned long long r[2] = {(unsigned long long)getpid(), (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count()};
    FILE *f = fopen("/dev/urandom", "rb");
    if (f) { unsigned long long x[2]; if (fread(x, sizeof x, 1, f) == 1) { r[0] ^= x[0]; r[1] ^= x[1]; } fclose(f); }
    snprintf(out->internal, sizeof out->internal, "%016llx%016llx", r[0], r[1]);
    return 0;
}

int ncclCommInitRank(void **comm, int world, Uid uid, int rank)
{
    if (!comm || world < 1 || world > 64 || rank < 0 || rank >= world) return 4;       // ncclInvalidArgument
    const std::string path = path_of(uid);
    const size_t bytes = sizeof(Shared) + kCap;
    int fd = -1;
    const auto t0 = std::chrono::steady_clock::now();
    if (rank == 0) {
        fd = open(path.c_str(), O_RDWR | O_CREAT | O_EXCL, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) { if (fd >= 0) close(fd); return 2; }
    } else {
        while ((fd = open(path.c_str(), O_RDWR)) < 0) {
            std::this_thread::sleep_for(std::chrono::milliseconds(2));
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s()) return 2;
        }
        struct stat st;
        while (fstat(fd, &st) == 0 && (size_t)st.st_size < bytes) {
            std::this_thread::sleep_for(std::chrono::milliseconds(2));
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s()) { close(fd); return 2; }
        }
    }
    void *m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return 2;
    Comm *c = new Comm{(Shared *)m, rank, world, path};
    if (rank == 0) c->sh->magic.store(kMagic, std::memory_order_release);
    else while (c->sh->magic.load(std::memory_order_acquire) != kMagic) {
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s()) { munmap(m, bytes); delete c; return 2; }
    }
    c->sh->joined.fetch_add(1);
    if (!barrier(c)) { munmap(m, bytes); delete c; return 2; }        // like the real call: returns when every rank has joined
    *comm = c;
    return 0;
}

int ncclCommCount(void *comm, int *n) { if (!comm || !n) return 4; *n = ((Comm *)comm)->world; return 0; }

int ncclBroadcast(const void *send, void *recv, size_t count, int dtype, int root, void *comm, hipStream_t stream)
{
    Comm *c = (Comm *)comm;
    const size_t bytes = count * type_bytes(dtype);
    if (!c || !type_bytes(dtype) || bytes > kCap || root < 0 || root >= c->world) return 4;
    if (hipStreamSynchronize(stream) != hipSuccess) return 1;
    if (c->rank == root && hipMemcpy(c->sh->payload, send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    if (!barrier(c)) return 2;
    if (c->rank != root) { if (hipMemcpy(recv, c->sh->payload, bytes, hipMemcpyHostToDevice) != hipSuccess) return 1; }
    else if (recv != send && hipMemcpy(recv, send, bytes, hipMemcpyDeviceToDevice) != hipSuccess) return 1;
    if (!barrier(c)) return 2;            // the window may be rewritten only after everybody has read it
    return 0;
}

// the engine's one use: a 1-element int32 MIN (the ranks' verdict on the header); SUM / MAX come for free
int ncclAllReduce(const void *send, void *recv, size_t count, int dtype, int op, void *comm, hipStream_t stream)
{
    Comm *c = (Comm *)comm;
    if (!c || dtype != 2 || count != 1) return 4;
    if (hipStreamSynchronize(stream) != hipSuccess) return 1;
    int v = 0;
    if (hipMemcpy(&v, send, sizeof v, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    c->sh->slots[c->rank] = v;
    if (!barrier(c)) return 2;
    int r = c->sh->slots[0];
    for (int i = 1; i < c->world; i++) { const int x = c->sh->slots[i]; r = op == 0 ? r + x : (op == 2 ? (x > r ? x : r) : (x < r ? x : r)); }
    if (!barrier(c)) return 2;
    if (hipMemcpy(recv, &r, sizeof r, hipMemcpyHostToDevice) != hipSuccess) return 1;
    return 0;
}

int ncclCommDestroy(void *comm)
{
    Comm *c = (Comm *)comm;
    if (!c) return 4;
    const bool last = c->sh->left.fetch_add(1) + 1 == (unsigned)c->world;
    munmap(c->sh, sizeof(Shared) + kCap);
    if (last) unlink(c->path.c_str());
    delete c;
    return 0;
}

const char *ncclGetErrorString(int r)
{
    switch (r) { case 0: return "no error"; case 1: return "fake rccl: HIP error"; case 2: return "fake rccl: system error / peer timed out"; case 4: return "fake rccl: invalid argument"; default: return "fake rccl: error"; }
}

}  // extern "C"
