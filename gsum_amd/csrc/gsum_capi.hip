// libgsum_hip.so — host side of the C ABI declared in include/gsum_hip.h.
// One context = one GPU = two HIP streams (main + high-priority panel stream for look-ahead).
// the library is built with -fvisibility=hidden: what include/gsum_hip.h declares (and, in the lab build, gsum_hip_debug.h) is all it exports
#pragma GCC visibility push(default)
#include "gsum_hip.h"
#ifdef GSUM_LAB
#include "gsum_hip_debug.h"
#endif
#pragma GCC visibility pop
#include "gsum_kernels.hip.h"

// The host side lives in host/*.hip.h, included here in dependency order.  Every entry point is declared extern "C" in
// include/gsum_hip.h (lab build: gsum_hip_debug.h), so the definitions below have C linkage without a wrapping block.
#include "host/context.hip.h"
#include "host/gemm.hip.h"
#include "host/matrices.hip.h"
#include "host/potrf.hip.h"
#include "host/api_context.hip.h"
#include "host/api_operators.hip.h"
#include "host/api_fused.hip.h"
#include "host/wave.hip.h"
#include "host/api_lml.hip.h"
#include "host/api_multi.hip.h"
#include "host/api_grad.hip.h"
#include "host/api_measure.hip.h"
