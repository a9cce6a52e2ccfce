# same box, back to back: parity of the ticket build first, then HL and cfg2 with and without the ticket
mkdir -p gpurun_out/ab
timeout -k 10 400 python -m pytest tests/test_gpu_sweep.py -m gpu -x -q -k "ticket or settled or hand_over or launch_groups or independent or T5 or mixed" > gpurun_out/ab/ticket_tests.log 2>&1 || { tail -20 gpurun_out/ab/ticket_tests.log; exit 1; }
tail -2 gpurun_out/ab/ticket_tests.log
for t in 2 0; do
  PMDI_TICKET=$t timeout -k 10 330 python bench.py --no-cpu --no-latency-form --steps 10 --warmup 5 > gpurun_out/ab/ticket$t.json 2> gpurun_out/ab/ticket$t.err || exit 1
  python scripts/bench_brief.py gpurun_out/ab/ticket$t.json | head -1
done
for t in 2 0; do
  PMDI_TICKET=$t timeout -k 10 200 python bench.py --workload cfg2 --no-cpu --steps 10 --warmup 3 > gpurun_out/ab/cfg2_ticket$t.json 2> gpurun_out/ab/cfg2_ticket$t.err || exit 1
  python scripts/bench_brief.py gpurun_out/ab/cfg2_ticket$t.json | head -1
done
