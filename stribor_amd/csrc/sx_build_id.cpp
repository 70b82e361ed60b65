// The library's build id: the first 16 hex digits of the sha256 over its sources (Makefile rule sx_build_id.inc).
extern "C" const char *sx_build_id(void) {
    return
#include "sx_build_id.inc"
        ;
}
