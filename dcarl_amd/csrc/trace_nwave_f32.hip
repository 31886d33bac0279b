// fp32-storage instances of the multi-wave online kernel (one translation unit per storage type: they compile in parallel)
#include "trace_nwave_impl.h"
namespace dcarl {
template bool launch_trace_nwave<float>(const float*, const uint8_t*, const int64_t*, const int32_t*, const int32_t*, int, int, const DevParams&,
                                        float*, uint8_t*, int32_t*, double*, int32_t*, float*, int32_t*, hipStream_t, int, const TraceCarry&);
}
