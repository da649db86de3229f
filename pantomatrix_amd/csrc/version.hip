#include "common.h"
extern "C" int emage_abi_version(void) { return 17; }
extern "C" const char* emage_target_arch(void) { return "gfx950"; }
