#include "common.h"
extern "C" int irr_abi_version(void) { return 2; }
