// state.hip.h -- per-stream state, per-call parameters and the counter-based noise generator (kernels.hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rvc {

// ------------------------------------------------------------------------------------
// per-stream state + per-call parameters
// ------------------------------------------------------------------------------------
struct CallParams {
    float uppower;          // (round 2: one multiplier per call; now per stream, StreamState::uppower -- kept for layout)
    uint32_t seed;
    uint32_t chunk_base;    // chunk counter of stream 0 is chunk[b] (kept per stream on device)
    int pad_;
};
struct StreamState {        // one per stream
    float cache_pitchf[1024];   // rvc.rs:42
    uint32_t chunk;
    uint32_t stream_id;
    int status;             // 0 ok, else a set of ST_* bits (atomicOr by the kernels, read and cleared by the host: engine.hip check_status)
    float uppower;          // this stream's 2^(pitch_shift / 12), truncating division (rvc.rs:121): every stream is its own caller
};

// status bits of a stream (several kernels of one chunk may report; a plain store would lose the earlier report)
enum { ST_PANIC = 1,          // the reference would have panicked (rmvpe.rs:124 out-of-bounds gather)
       ST_HANDOFF = 2,        // the GRU recurrence's cross-workgroup hand-off timed out: the chunk is lost
       ST_KNN_TIMEOUT = 4 };  // the one-launch retrieval's selectors gave up waiting for a workgroup: the host recomputes the chunk's retrieval

// Philox4x32-10, the same counter layout as oracle/rvc_oracle.c (ora_philox_normal)
__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
__device__ __forceinline__ float u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }
__device__ __forceinline__ void philox_normal4(uint32_t seed, uint32_t stream, uint32_t chunk, uint32_t purpose, uint32_t blk, float z[4])
{
    uint32_t c[4] = {blk, 0u, chunk, purpose};
    philox4x32_10(c, seed, stream);
    float r0 = sqrtf(-2.0f * logf(u01(c[0]))), a0 = 6.28318530717958647692f * u01(c[1]);
    float r1 = sqrtf(-2.0f * logf(u01(c[2]))), a1 = 6.28318530717958647692f * u01(c[3]);
    z[0] = r0 * cosf(a0); z[1] = r0 * sinf(a0); z[2] = r1 * cosf(a1); z[3] = r1 * sinf(a1);
}

}  // namespace rvc
