/*
 * rvc_mi355x_debug.h -- TEST HOOKS of librvc_mi355x.so.  Not part of the drop-in boundary (include/rvc_mi355x.h is): these entry
 * points exist so that the parity tests under tests/ and the profiling tools under tests/tools/ can force a code path the planner
 * would not pick for the geometry at hand, or look inside a plan.  A host (the `rvc` crate shim, rvc-rpc) never calls them, and none
 * of them is reachable through the environment.  Apart from rvc_* of the two headers the library exports nothing
 * (obs_rvc_amd/csrc/exports.map).
 */
#ifndef RVC_MI355X_DEBUG_H
#define RVC_MI355X_DEBUG_H
#include "rvc_mi355x.h"
#ifdef __cplusplus
extern "C" {
#endif

/* set (value != NULL) or clear one of the named planner hooks (obs_rvc_amd/csrc/plan.hip, kTestHooks: every one of them selects between
 * kernels / tiles / plan structures that compute the SAME result up to fp32 summation order); 0 = done, -1 = not a hook.  Plans built
 * before a hook changed are dropped. */
int rvc_debug_option(const char *name, const char *value);
/* device timestamps at section boundaries of the last call (hook RVC_STAMPS): "name us" lines; returns the number of stamps */
int rvc_debug_stamps(rvc_engine *e, char *buf, size_t cap);
/* launches (ops) of the last call's plan */
int rvc_debug_last_plan(rvc_engine *e, int *n_ops);
/* one line per profiled launch of the last call: "<us> <gflop> <description>" */
int rvc_debug_profile_dump(rvc_engine *e, char *buf, size_t cap);
/* one Conv1d / one Conv2d 3x3 or ConvTranspose2d 3x3 stride 2 / the folded-LayerNorm launch pair on deterministic data through whatever kernel
 * the planner (or a hook) picks, against a double-precision host evaluation: largest |gpu - host| / rms(host); negative on failure */
double rvc_debug_conv_check(rvc_engine *e, int M, int Cin, int KW, int dil, int N, int streams, int pre_act);
double rvc_debug_conv2d_check(rvc_engine *e, int M, int Cin, int H, int W, int streams, int kind, int residual);
double rvc_debug_ln_fold_check(rvc_engine *e, int M, int K, int N, float offset);
/* the autotuner's decisions of this process, one line each ("<layer signature> -> [choice] <kernel description> | <us> (<candidates>)"); returns the number of
 * entries.  reset forgets them (the next plan build measures again). */
int rvc_debug_autotune_dump(char *buf, size_t cap);
void rvc_debug_autotune_reset(void);
/* weight slabs alive on a device: count and bytes (obs_rvc_amd/csrc/plan.hip, wmalloc) */
int rvc_debug_weight_slabs(int device, int *count, size_t *bytes);
/* the kernel the planner chose for the last rvc_debug_conv*_check launch ("reg", "g32", "c32s", ...) */
const char *rvc_debug_last_kernel(void);

#ifdef __cplusplus
}
#endif
#endif
