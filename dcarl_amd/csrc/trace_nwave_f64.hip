// f64-storage instances of the multi-wave online kernel (what the drop-in scripts run on the bundled float64 tables)
#include "trace_nwave_impl.h"
namespace dcarl {
template bool launch_trace_nwave<double>(const double*, const uint8_t*, const int64_t*, const int32_t*, const int32_t*, int, int, const DevParams&,
                                         double*, uint8_t*, int32_t*, double*, int32_t*, float*, int32_t*, hipStream_t, int, const TraceCarry&);
}
