"""Where do the register spills of the sweep kernel sit?  Compiles pmdi_sweep.hip with line tables (no GPU needed) and maps
every scratch_store / scratch_load of the chosen builds to its source line.

    python scripts/spill_sites.py [extra hipcc flags]

Round 2: in the 256-register build `<256,2,false>` 22 of the 36 scratch stores sit on make_ks() (pmdi_device.h) -- the 22
wave-uniform pointers of a (chain, dataset), computed at the top of EVERY step and parked in scratch: 22 stores x 256 lanes x
4 B = 22 KB per chain and step, about half of the spill write-back that dominates WRITE_SIZE (profiles/README.md).  With
the addresses rebuilt from the argument block where they are used (round 3 default) the kernel body has 5 stores / 6 loads.
(Its first version loaded the argument block through a generic pointer: 4 000 extra flat_load instructions instead of s_load --
caught by the instruction-class line below, fixed by going through the constant address space.)
"""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "particlemdi.jl_amd", "csrc", "pmdi_sweep.hip")
with tempfile.TemporaryDirectory() as tmp:
    out = os.path.join(tmp, "x.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "--cuda-device-only",
                           "-gline-tables-only", "-S", src, "-o", out] + sys.argv[1:], stderr=subprocess.DEVNULL)
    text = open(out).read().split("\n")
files, cur, loc = {}, None, None
hist = collections.defaultdict(collections.Counter)
for line in text:
    m = re.match(r"^(_Z\w+):", line)
    if m:
        cur, loc = m.group(1), None
        continue
    if line.startswith(".Lfunc_end"):
        cur = None
        continue
    m = re.match(r'\s+\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', line)
    if m:
        files[int(m.group(1))] = m.group(3) or m.group(2)
        continue
    m = re.match(r"\s+\.loc\s+(\d+)\s+(\d+)", line)
    if m:
        loc = (int(m.group(1)), int(m.group(2)))
        continue
    if cur and re.match(r"\s+scratch_(store|load)", line):
        f = os.path.basename(files.get(loc[0], "?")) if loc else "?"
        hist[cur][(f, loc[1] if loc else 0, "st" if "scratch_store" in line else "ld")] += 1
cls = collections.Counter()
for line in text:       # instruction classes of the whole translation unit: a variant that turns scalar loads into flat loads shows here
    m = re.match(r"\s+(flat|global|scratch|s_load|ds|buffer)_", line) or re.match(r"\s+(s_load)", line)
    if m:
        cls[m.group(1)] += 1
print("instruction classes (all builds of the file):", dict(cls))
for fn, h in hist.items():
    if not any(t in fn for t in ("sweep_kernelILi256ELi2ELb0", "sweep_kernelILi512ELi4ELb0", "sweep_resampleILi256", "sweep_resampleILi512")):
        continue
    st = sum(v for (f, l, k), v in h.items() if k == "st"); ld = sum(v for (f, l, k), v in h.items() if k == "ld")
    print(f"{fn}: {st} scratch stores, {ld} scratch loads")
    by_file = collections.Counter()
    for (f, l, k), v in h.items():
        by_file[(f, k)] += v
    for (f, k), v in sorted(by_file.items()):
        print(f"    {f:28s} {k} {v}")
    top = sorted(h.items(), key=lambda kv: -kv[1])[:8]
    print("    top sites:", ", ".join(f"{f}:{l} {k} x{v}" for (f, l, k), v in top))
