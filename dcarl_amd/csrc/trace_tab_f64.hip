// trace_tab_kernel instances for f64 record storage (see trace_tab_impl.h)
#define DCARL_TAB_T double
#include "trace_tab_impl.h"
