// version.cpp -- rvc_version(): the only unit that sees the hash of the library's sources, so a change anywhere relinks but does not
// recompile the other translation units.
#ifndef RVC_SRC_HASH
#define RVC_SRC_HASH "unhashed"
#endif
// "... rvc-mi355x-src:<sha256[:16] of the sources this binary was compiled from>" (obs_rvc_amd/_native.py source_hash / binary_hash)
extern "C" const char *rvc_version(void) { return "rvc-mi355x 0.3 (gfx950) rvc-mi355x-src:" RVC_SRC_HASH; }
