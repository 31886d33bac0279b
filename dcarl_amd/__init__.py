"""dcarl_amd — MI355X-native implementation of DCARL's confidence-estimation hot path.

Importing the package does not touch the GPU; the first call that needs the HIP library loads it and raises
``DcarlError`` if it is missing or no gfx950 device is visible (there is no CPU fallback)."""
from ._lib import DcarlError, device_info, load as load_library, require_gpu
from .params import Params
from .records import RecordTable
from .estimator import BoundsResult, ConfidenceEstimator, TraceResult, TraceState, census_report
from . import carla_records, dist, episodes, frenet, layout, reference_api, rls, sampler, stream, workloads

__all__ = ["DcarlError", "Params", "RecordTable", "ConfidenceEstimator", "TraceResult", "TraceState", "BoundsResult", "dist",
           "carla_records", "episodes", "frenet", "layout", "reference_api", "rls", "sampler", "stream", "workloads", "device_info", "load_library", "require_gpu"]
