// Library identification.
#include "common.h"
extern "C" const char* mtd_version(void) { return "mtdgan_hip 0.1.0 (gfx950)"; }
