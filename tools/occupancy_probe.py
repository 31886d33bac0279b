"""Report how many trace_kernel blocks are resident per CU (HIP occupancy API) for a given LDS size."""
import ctypes as C, sys
hip = C.CDLL("libamdhip64.so")
# use a raw hip module query through torch is awkward; approximate through device props
class Props(C.Structure):
    _fields_ = [("raw", C.c_char * 4096)]
import torch
p = torch.cuda.get_device_properties(0)
print("CUs", p.multi_processor_count, "shared mem per block", getattr(p, "shared_memory_per_block", None),
      "shared per multiprocessor", getattr(p, "shared_memory_per_multiprocessor", None),
      "max threads per multiprocessor", p.max_threads_per_multi_processor)
