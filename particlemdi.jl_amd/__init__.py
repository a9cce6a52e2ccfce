"""MI355X-native conditional-SMC sweep for ParticleMDI (hot path of src/pmdi.jl).

The directory name `particlemdi.jl_amd` is not a Python identifier; load it
with `__graft_entry__.load_package()` which registers it as `particlemdi_jl_amd`.
"""
from ._lib import (  # noqa: F401
    ABI_VERSION, CATEGORICAL, EXPORTS, GAUSSIAN, KIND_BY_NAME, LIB_PATH, NEGBINOM,
    ClusterBatch, Comm, CsvWriter, Gibbs, format_float64, read_allocations, PmdiError, Sweeper, build, lib,
    STEP_ALIGN, STEP_BEGIN, STEP_FEATSEL, STEP_HYPERS, STEP_SWEEP,
)
