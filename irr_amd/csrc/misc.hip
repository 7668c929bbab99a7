#include "common.h"
extern "C" int irr_abi_version(void) { return 3; }   // 3: Adam scalars are doubles, irr_conv_pack_job_block0_offset
