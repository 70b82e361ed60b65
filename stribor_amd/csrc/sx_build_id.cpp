// The library's build id: the first 16 hex digits of the sha256 over its sources (Makefile rule sx_build_id.inc).
extern "C" const char *sx_build_id(void) {
    return
#include "sx_build_id.inc"
        ;
}

// ---- experiment knobs of the library, in ONE place -----------------------------------------------------------------------------
// Every environment variable the library reads goes through sx_debug_knob (read once per process, by name); none is needed to use
// the library -- they serve the A/B timings of tools/ (knob_sweep.sh, sweep_affine.py, bench_slab.py):
//   SX_NO_PURE_MODE     fused kernel: split-coupling programs run the general kernel (MODE 0 / 2) instead of MODE 5 - 8
//   SX_STATIC_CHUNKS    fused kernel: static grid-stride chunks instead of the ticket hand-out
//   SX_BLOCKS_PER_CU    fused kernel: workgroups per CU of the launch grid
//   SX_AFFINE_VARIANT / SX_AFFINE_GRID   stand-alone affine kernel: code variant / grid size
//   SX_CUMSUM_NO_PIPE   Cumsum kernel without its software pipeline
//   SX_SLAB_SPW / SX_SLAB_NO_XCD         spline slab backward: slabs per workgroup, no XCD-aware workgroup map
#include <stdlib.h>
#include <string.h>
extern "C" int sx_debug_knob(const char *name, int dflt) {
    static const char *const known[] = {"SX_NO_PURE_MODE", "SX_STATIC_CHUNKS", "SX_BLOCKS_PER_CU", "SX_AFFINE_VARIANT", "SX_AFFINE_GRID",
                                        "SX_CUMSUM_NO_PIPE", "SX_SLAB_SPW", "SX_SLAB_NO_XCD"};
    bool ok = false;
    for (const char *k : known) ok = ok || strcmp(k, name) == 0;
    if (!ok) return dflt;                      // (an undeclared name is never read from the environment)
    const char *e = getenv(name);
    if (e == nullptr) return dflt;
    char *end = nullptr;
    const long v = strtol(e, &end, 10);
    return end == e ? 1 : (int)v;              // set without a number (or to a non-number) = 1; "0" = 0
}
