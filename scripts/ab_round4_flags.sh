mkdir -p gpurun_out/ab
for v in particlemdi.jl_amd/libpmdi_hip.so build_ab/lib_ilp.so build_ab/lib_trk.so build_ab/lib_mem.so; do
  n=$(basename $v .so)
  PMDI_NO_BUILD=1 PMDI_LIB_PATH=$v timeout -k 10 200 python bench.py --no-cpu --no-latency-form --steps 8 --warmup 2 --chains 1024 > gpurun_out/ab/$n.json 2> gpurun_out/ab/$n.err || exit 1
  python scripts/bench_brief.py gpurun_out/ab/$n.json | head -1
done
timeout -k 10 330 python bench.py --no-cpu --no-latency-form --steps 10 --warmup 5 --chains 4096 --pool-frac 0.35 > gpurun_out/ab/c4096.json 2> gpurun_out/ab/c4096.err || exit 1
python scripts/bench_brief.py gpurun_out/ab/c4096.json | head -1
