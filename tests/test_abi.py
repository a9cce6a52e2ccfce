"""The C-ABI library builds for gfx950, loads, and exports every symbol include/pmdi_hip.h
declares.  No compute calls here (no GPU in this container)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "pmdi_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pmdi_[A-Za-z_0-9]+)\s*\(", src)))


def test_header_symbols_exported(pkg):
    lib = pkg.lib()
    names = declared_functions()
    assert len(names) >= 17
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/pmdi_hip.h but not exported"
    assert set(pkg.EXPORTS) == set(names)
    assert lib.pmdi_abi_version() == pkg.ABI_VERSION == 2


def test_the_library_reads_the_environment_only_on_request():
    """no process-global state behind the ABI.  The kernel-selection knobs travel in pmdi_config.tuning; getenv appears in
    the product's sources in exactly one function, pmdi_tuning_from_env, which a caller has to call itself."""
    csrc = os.path.join(ROOT, "particlemdi.jl_amd", "csrc")
    hits = []
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".cpp", ".h")):
            txt = open(os.path.join(csrc, f), errors="ignore").read()
            hits += [(f, i + 1) for i, line in enumerate(txt.split("\n")) if re.search(r"\bgetenv\s*\(", line)]
    assert hits and all(f == "pmdi_api.cpp" for f, _ in hits), hits
    api = open(os.path.join(csrc, "pmdi_api.cpp")).read()
    body = api[api.index("void pmdi_tuning_from_env("):]
    body = body[:body.index("\n}\n") + 3]
    assert api.count("getenv") == body.count("getenv"), "getenv outside pmdi_tuning_from_env"


def test_code_object_is_gfx950(pkg):
    blob = open(pkg.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    assert b"pmdi_sweep_kernel" in blob


def test_oracle_is_not_linked_into_the_product(pkg):
    blob = open(pkg.LIB_PATH, "rb").read()
    assert b"pmdi_oracle" not in blob
    for root, _, files in os.walk(os.path.join(ROOT, "particlemdi.jl_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".jl")):
                txt = open(os.path.join(root, f), errors="ignore").read()
                assert not re.search(r"#include[^\n]*oracle|import[^\n]*oracle|from\s+oracle|libpmdi_oracle|ccall[^\n]*oracle", txt), \
                    f"{f} uses the oracle"


def test_no_gpu_means_loud_failure(pkg):
    import numpy as np
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; this checks the no-device error path")
    with pytest.raises(pkg.PmdiError) as e:
        pkg.Sweeper([np.zeros((10, 2))], ["gaussian"], 3, 4)
    assert e.value.code == -2          # PMDI_E_DEVICE: there is no CPU fallback


def test_argument_validation_happens_before_device_use(pkg):
    import numpy as np
    x = np.zeros((10, 2))
    for kwargs in (dict(N=1, P=4), dict(N=3, P=1), dict(N=11, P=4)):
        with pytest.raises(pkg.PmdiError) as e:
            pkg.Sweeper([x], ["gaussian"], kwargs["N"], kwargs["P"])
        assert e.value.code == -1      # the @asserts of src/pmdi.jl:50-55
