// Single translation unit of libofq_hip.so (gfx950 only).
#include "common.h"
extern "C" int ofq_abi_version(void) { return OFQ_ABI_VERSION; }
#ifndef OFQ_SOURCE_HASH
#define OFQ_SOURCE_HASH "unhashed-build--"
#endif
// content hash of csrc/* + include/ofq_hip.h at build time (ofq_amd/build.py); the loader compares it with the sources
// it finds next to the library and refuses (or rebuilds) a stale .so
extern "C" const char* ofq_source_hash(void) { return "OFQ_SOURCE_HASH=" OFQ_SOURCE_HASH; }
#include "statsq.hip"
#include "lsq.hip"
#include "softmax_lsq.hip"
#include "gemm_f32.hip"
#include "qgemm_args.h"
#include "qgemm_i8.hip"
#include "qgemm_planes.hip"
#include "qgemm_tn.hip"
#include "qgemm_nn.hip"
#include "qgemm_codes.hip"
#include "qgemm_i8_bwd.hip"
#include "qgemm_nt_wide.hip"
#include "qgemm_nt_sk.hip"
#include "gemm_planes_f32.hip"
#include "qgemm_nt_lsq.hip"
#include "qattn_launch.hip"
#include "qattn_fused.hip"
#include "attn_f32.hip"
#include "layernorm.hip"
#include "misc.hip"
#include "adamw.hip"
#include "input_pipeline.hip"
