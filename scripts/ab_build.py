"""A/B tooling: build the CURRENT working tree into build_ab/lib_<NAME>.so, next to the in-tree library.

    python scripts/ab_build.py NAME [extra hipcc flags...]

Then, on the GPU box, run the variants back to back in one gpurun call (same box, same thermal state):

    for v in build_ab/lib_a.so build_ab/lib_b.so particlemdi.jl_amd/libpmdi_hip.so; do
        PMDI_NO_BUILD=1 PMDI_LIB_PATH=$v python bench.py --no-cpu --steps 8 --warmup 2 > gpurun_out/ab.json 2> gpurun_out/ab.err || exit 1
        python scripts/bline.py $v < gpurun_out/ab.json
    done

A variant must pass `pytest tests/test_gpu_sweep.py tests/test_gpu_soak.py -m gpu` with PMDI_LIB_PATH set before its bench number means
anything.  (Round 3: the lazy-address KS went through exactly that -- 36 tests equal, HL 459.9 -> 495.6 it/s -- and is the default now;
the other round-2 experiments were deleted unrun: the settled-chain kernel replaces what they patched.)

PMDI_NO_BUILD=1 keeps the box from rebuilding a variant from the (different) sources that travelled with it.  build_ab/ is
git-ignored but travels with gpurun.  (Round 2: column table 469.8 / +uniform log-weights 465.4 / +LDS mirrors 442.1 it/s.)
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name = sys.argv[1]
out = os.path.join(ROOT, "build_ab", f"lib_{name}.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
env = dict(os.environ, PMDI_LIB_PATH=out, PMDI_EXTRA_HIPCC_FLAGS=" ".join(sys.argv[2:]))
env.pop("PMDI_NO_BUILD", None)
subprocess.check_call([sys.executable, "-c", "import __graft_entry__ as G; pkg = G.load_package(); pkg.build(force=True)"], cwd=ROOT, env=env)
print(out)
