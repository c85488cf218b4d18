// synth_front.h -- the synthesizer's front half (text encoder -> prior sample -> normalising flows, reference: the "phone / pitch /
// pitchf" -> latent part of the ONNX graph run at rvc/src/rvc.rs:193-214) as ONE persistent launch for one stream.
//
// At one stream these ~70 layers work on a 21-column window (return_length): each was a 5-7 us launch whose time is the launch
// skeleton, not the arithmetic.  synth_front_kernel keeps G workgroups resident and walks the layers as
// "steps": every step's output travels to the workgroups that consume it as 8-byte {tag, value} granules (agent-scope relaxed atomic
// stores / loads: the data is the flag, placement independent, cdna_hip_programming.md guideline 16 form R2; measured 0.5-0.6 us per
// hand-off against ~4 us per kernel boundary).  Arithmetic is the same fp32 MFMA (v_mfma_f32_16x16x4_f32) over the same
// fragment-major weight panels the implicit-GEMM launches use.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rvc {

struct StreamState;
struct CallParams;

struct SfW { const float *w, *b; int M, nchunks; };      // fragment-major panel [M/16][nchunks][64][4] + bias [M]

struct SfLayer { SfW qkv, o, ff1, ff2; const float *rel_k, *rel_v, *ln1_g, *ln1_b, *ln2_g, *ln2_b; };
struct SfFlow { SfW pre, post, in[4], rs[4]; int flipped; };

constexpr int SF_MAX_LAYERS = 8, SF_MAX_FLOWS = 6, SF_MAX_T = 48, SF_THREADS = 512;

struct SynFrontP {
    int T;                    // return_length: columns of every activation (<= SF_MAX_T)
    int C, H, F, I;           // phone dim, hidden, FFN filter, inter (latent) channels
    int heads, window, n_layers, n_flows, wn_layers, enc_k, wn_k;
    const float *phone; int phone_ld;      // [C][phone_ld] plain floats (gather_phone / retrieval blend output)
    const int *pitch;                      // [T] coarse pitch
    const float *pitch_emb;                // [256][H]
    SfW phone_w, proj;
    SfLayer layer[SF_MAX_LAYERS];
    SfFlow flow[SF_MAX_FLOWS];
    unsigned long long *gran;              // granule workspace (synth_front_ws_granules() entries, zero-initialised once)
    unsigned *epoch;                       // tag base of this launch; the chunk's last kernel advances it (no per-launch memset, replay-safe)
    float *z_out; int z_ld;                // latent [I][z_ld] as plain floats for the decoder's first convolution
    const StreamState *st; const CallParams *cp;
    int *status;                           // stream status word: 7 = a hand-off timed out
    unsigned long long *stamps;            // tuning aid: [steps + 2] device wall-clock stamps of workgroup 0 (or nullptr)
};

// whether the persistent kernel covers this configuration (else the engine keeps the per-layer launches)
bool synth_front_supported(const SynFrontP &p);
size_t synth_front_ws_granules(const SynFrontP &p);
size_t synth_front_lds_bytes(const SynFrontP &p);
int synth_front_steps(const SynFrontP &p);
int synth_front_grid(const SynFrontP &p);              // workgroups of the launch (all must be co-resident: <= 256, one per CU)
void launch_synth_front(const SynFrontP &p, hipStream_t s);

}  // namespace rvc
