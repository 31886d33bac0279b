// trace_tab_kernel instances for f32 record storage (see trace_tab_impl.h)
#define DCARL_TAB_T float
#include "trace_tab_impl.h"
