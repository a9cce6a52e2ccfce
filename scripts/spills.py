"""Register / spill report of every pmdi_sweep_kernel build (hipcc cross-compiles; no GPU needed)."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "particlemdi.jl_amd", "csrc", "pmdi_sweep.hip")
with tempfile.TemporaryDirectory() as tmp:
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "--cuda-device-only",
                        "-c", src, "-o", os.path.join(tmp, "x.o"), "-Rpass-analysis=kernel-resource-usage"] + sys.argv[1:], capture_output=True, text=True)
cur = None
rows = {}
for line in r.stderr.splitlines():
    m = re.search(r"Function Name: \S*pmdi_sweep_kernel(\w+?)EEEvPK9SweepArgs", line)
    if m:
        cur = m.group(1); rows[cur] = {}
    for key in ("VGPRs", "VGPRs Spill", "SGPRs Spill", "ScratchSize \\[bytes/lane\\]", "Occupancy \\[waves/SIMD\\]"):
        m = re.search(r"\s" + key + r": (\d+)", line)
        if m and cur:
            rows[cur][key.replace("\\", "")] = int(m.group(1))
for k, v in rows.items():
    print(k, v)
