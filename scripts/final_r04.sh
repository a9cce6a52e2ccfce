# round-4 record of the final tree: GPU tests, the driver's command, then the rocprofv3 passes of the HL operating point
mkdir -p gpurun_out
timeout -k 10 420 python -m pytest tests -m gpu -x -q > gpurun_out/gputest.log 2>&1 || { tail -25 gpurun_out/gputest.log; exit 1; }
tail -2 gpurun_out/gputest.log
timeout -k 10 330 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver_flags.json 2> gpurun_out/bench_driver_flags.err || { tail -5 gpurun_out/bench_driver_flags.err; exit 1; }
python scripts/bench_brief.py gpurun_out/bench_driver_flags.json | head -1
cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 python3 $GRAFT_REPO_ROOT/scripts/profile_config.py HL r04 --steps 8 --warmup 3 > $GRAFT_REPO_ROOT/gpurun_out/profile_HL.log 2>&1 || { tail -20 $GRAFT_REPO_ROOT/gpurun_out/profile_HL.log; exit 1; }
tail -2 $GRAFT_REPO_ROOT/gpurun_out/profile_HL.log
