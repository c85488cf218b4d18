/* Minimal C client of the C ABI (include/rvc_mi355x.h): what a cgo / Rust-FFI / C host does.
 *   gcc -std=c99 -I include examples/c_smoke.c -L obs_rvc_amd/csrc -lrvc_mi355x -Wl,-rpath,$PWD/obs_rvc_amd/csrc -lm -o c_smoke
 *   ./c_smoke <data_path> <model.rvcw>
 * Runs RvcInfer::new / load_* / infer (rvc/src/rvc.rs:30-220) on a synthetic 2.24 s ring and the native session
 * (process_one_frame, obs-rvc/src/lib.rs:659-795) on a 160 ms chunk; prints sizes and RMS values. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "rvc_mi355x.h"

static double rms(const float *x, size_t n) { double s = 0; for (size_t i = 0; i < n; i++) s += (double)x[i] * x[i]; return n ? sqrt(s / (double)n) : 0; }

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s <data_path> <model.rvcw>\n", argv[0]); return 2; }
    rvc_engine *e = NULL;
    if (rvc_create(argv[1], -1, &e) != RVC_OK) { fprintf(stderr, "rvc_create failed (no HIP device?)\n"); return 1; }
    if (rvc_load_contentvec(e, RVC_VERSION_V2) != RVC_OK || rvc_load_f0(e, RVC_PITCH_RMVPE) != RVC_OK || rvc_load_model(e, argv[2]) != RVC_OK) {
        fprintf(stderr, "load failed: %s\n", rvc_last_error_message(e)); return 1;
    }
    rvc_set_noise_seed(e, 1234, 0);
    const size_t L = 35840, frame16k = 2560;              /* BASELINE 160 ms geometry */
    float *in = (float *)malloc(L * sizeof(float));
    for (size_t i = 0; i < L; i++) in[i] = 0.1f * (float)sin(2.0 * 3.14159265358979 * 180.0 * (double)i / 16000.0) + 0.05f * (float)sin(2.0 * 3.14159265358979 * 360.0 * (double)i / 16000.0);
    size_t n_out = 0;
    float *out = (float *)malloc(64000 * sizeof(float));
    rvc_status rc = rvc_infer(e, in, L, frame16k, 1, 12, 200, 21, out, 64000, &n_out);
    if (rc != RVC_OK) { fprintf(stderr, "infer failed (%d): %s\n", (int)rc, rvc_last_error_message(e)); return 1; }
    printf("infer: %zu samples, rms %.6f, gpu %.3f ms\n", n_out, rms(out, n_out), (double)rvc_last_gpu_ms(e));
    /* output buffer too small: NdarrayShapeError, required size still reported */
    rc = rvc_infer(e, in, L, frame16k, 1, 12, 200, 21, out, 10, &n_out);
    printf("small buffer: status %d (expected %d), required %zu\n", (int)rc, (int)RVC_SHAPE, n_out);
    /* the plugin-side chain: host rate = 100 * (n_out / 21) so that no real resampling ratio is needed for any model rate */
    const size_t host_rate = 100 * (n_out / 21);
    rvc_session *s = NULL;
    rc = rvc_session_create(e, host_rate, 0.16, 0.07, 2.0, host_rate, 12, 0.75, 0, &s);
    if (rc != RVC_OK) { fprintf(stderr, "session failed (%d): %s\n", (int)rc, rvc_last_error_message(e)); return 1; }
    const size_t F = rvc_session_frame_size(s);
    float *chunk = (float *)malloc(F * sizeof(float)), *frame = (float *)malloc(F * sizeof(float));
    size_t off = 0; double last = 0;
    for (int c = 0; c < 20; c++) {
        for (size_t i = 0; i < F; i++) chunk[i] = 0.1f * (float)sin(2.0 * 3.14159265358979 * 180.0 * (double)(c * F + i) / (double)host_rate);
        rc = rvc_session_process(s, chunk, F, frame, F, &off);
        if (rc != RVC_OK && rc != RVC_PANIC) { fprintf(stderr, "session_process failed (%d): %s\n", (int)rc, rvc_last_error_message(e)); return 1; }
        last = rms(frame, F);
    }
    printf("session: frame %zu samples, last rms %.6f, last sola offset %zu\n", F, last, off);
    rvc_session_destroy(s);
    rvc_destroy(e);
    free(in); free(out); free(chunk); free(frame);
    printf("ok %s\n", rvc_version());
    return 0;
}
