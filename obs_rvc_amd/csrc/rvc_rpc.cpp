// rvc-rpc -- stdio RPC server, wire-compatible with the reference's `rvc-rpc` binary
// (reference: rvc-rpc/src/main.rs:8-103; client side: obs-rvc/src/rvcadapter.rs:34-119).
//
//   rvc-rpc <"v1"|"v2"> <"rmvpe"> <model_path> <data_path>
// Request (little endian): u32 nbytes | nbytes of f32 PCM @16 kHz | u32 sample_frame_16k_size | i32 pitch_shift |
//                          u32 skip_head | u32 return_length
// Reply:                   u32 nbytes | f32 PCM at the model rate
// Like the reference, any failure terminates the process (the plugin then respawns it, obs-rvc/src/lib.rs:716-727);
// "Ready to receive input" goes to stderr (main.rs:62).  Built on the C ABI only.
#include "../../include/rvc_mi355x.h"

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static void die(rvc_engine *e, const char *what, int rc)
{
    fprintf(stderr, "rvc-rpc: %s failed (status %d): %s\n", what, rc, e ? rvc_last_error_message(e) : "");
    exit(101);   // Rust panics exit with 101
}

static bool read_exact(FILE *f, void *buf, size_t n) { return n == 0 || fread(buf, 1, n, f) == n; }

int main(int argc, char **argv)
{
    if (argc < 4) {   // main.rs:14-17: `args.len() < 4` prints the usage and returns (exit status 0)
        fprintf(stderr, "Usage: rvc-rpc <version> <f0_algorithm> <model> <data>\n");
        return 0;
    }
    if (argc < 5) {   // exactly three arguments pass that check and then main.rs:22 indexes args[4]: panic, exit status 101
        fprintf(stderr, "rvc-rpc: index out of bounds: the len is %d but the index is 4 (main.rs:22)\n", argc);
        return 101;
    }
    // RvcModelVersion::from(&str): "v1" -> V1, anything else -> V2 (enums.rs:66-74); PitchAlgorithm: always Rmvpe (enums.rs:126-133)
    const int version = strcmp(argv[1], "v1") == 0 ? RVC_VERSION_V1 : RVC_VERSION_V2;
    const char *model_path = argv[3], *data_path = argv[4];
    rvc_engine *e = nullptr;
    int rc = rvc_create(data_path, -1, &e);
    if (rc != RVC_OK) die(nullptr, "engine creation (no MI355X visible?)", rc);
    if ((rc = rvc_load_contentvec(e, version)) != RVC_OK) die(e, "loading contentvec model", rc);   // main.rs:35-40
    if ((rc = rvc_load_f0(e, RVC_PITCH_RMVPE)) != RVC_OK) die(e, "loading f0 model", rc);             // main.rs:42-47
    if ((rc = rvc_load_model(e, model_path)) != RVC_OK) die(e, "loading model", rc);                  // main.rs:49-54
    if (const char *s = getenv("RVC_NOISE_SEED")) rvc_set_noise_seed(e, (uint32_t)strtoul(s, nullptr, 10), 0);
    if (getenv("RVC_USE_GRAPH")) rvc_set_use_graph(e, 1);

    static char ibuf[1 << 20], obuf[1 << 20];   // 1 MiB buffers as in main.rs:59-60
    setvbuf(stdin, ibuf, _IOFBF, sizeof ibuf);
    setvbuf(stdout, obuf, _IOFBF, sizeof obuf);
    fprintf(stderr, "Ready to receive input\n");
    fflush(stderr);

    std::vector<float> in, out;
    for (;;) {
        uint32_t nbytes, frame, skip_head, return_length;
        int32_t pitch_shift;
        if (!read_exact(stdin, &nbytes, 4)) return 101;      // read_exact(..).unwrap() -> panic on EOF (main.rs:66)
        in.resize(nbytes / 4);
        std::vector<unsigned char> raw(nbytes);
        if (!read_exact(stdin, raw.data(), nbytes)) return 101;
        memcpy(in.data(), raw.data(), (size_t)(nbytes / 4) * 4);
        if (!read_exact(stdin, &frame, 4) || !read_exact(stdin, &pitch_shift, 4) || !read_exact(stdin, &skip_head, 4) ||
            !read_exact(stdin, &return_length, 4))
            return 101;
        size_t n_out = 0;
        // the reply holds return_length 10 ms hops at the model rate (<= 480 samples each at 48 kHz; 1024 leaves room for any
        // synthesizer hop).  A wire value no chunk of the plugin can produce (its slider tops out at a few hundred hops; infer itself
        // rejects skip_head + return_length > 2T + 1) must not size a multi-gigabyte buffer: the reference would panic inside infer
        // (rvc.rs:155 slice out of range) -> exit 101
        if (return_length > (1u << 16)) { fprintf(stderr, "rvc-rpc: return_length %u out of range\n", return_length); return 101; }
        out.resize((size_t)return_length * 1024 + 16);
        rc = rvc_infer(e, in.data(), in.size(), frame, 1, pitch_shift, skip_head, return_length, out.data(), out.size(), &n_out);
        if (rc != RVC_OK) die(e, "infer", rc);                                                         // main.rs:93 unwrap
        const uint32_t obytes = (uint32_t)(n_out * 4);
        fwrite(&obytes, 4, 1, stdout);
        fwrite(out.data(), 4, n_out, stdout);
        fflush(stdout);
    }
}
