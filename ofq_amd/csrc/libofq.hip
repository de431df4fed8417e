// Single translation unit of libofq_hip.so (gfx950 only).
#include "common.h"
extern "C" int ofq_abi_version(void) { return OFQ_ABI_VERSION; }
#include "statsq.hip"
#include "lsq.hip"
#include "softmax_lsq.hip"
#include "gemm_f32.hip"
#include "qgemm.hip"
#include "layernorm.hip"
#include "misc.hip"
#include "adamw.hip"
