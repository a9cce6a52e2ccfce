"""Print the few numbers of a bench.py JSON line that a tuning session looks at (profiling aid)."""
import json
import sys

for path in sys.argv[1:]:
    try:
        d = json.loads(open(path).read().strip().splitlines()[-1])
    except Exception as e:           # noqa: BLE001
        print(path, "unreadable:", e)
        continue
    pc = d.get("parity_check_detail", {})
    print(f"{path}: {d['value']:.1f} it/s  sweep {d['sweep_kernel_ms']:.0f} ms  chains {d['config']['chains_per_gpu']}  busy {d.get('chain_slot_busy_frac')}  "
          f"frac {d['roofline']['frac']:.4f}  parity {d.get('parity_check')}  burnin {d.get('burnin_iters_per_sec')}")
    print("   kernels:", json.dumps(d.get("kernels")))
    print("   per chain:", json.dumps(d.get("per_chain_iters_per_sec", {}) and {k: v for k, v in d["per_chain_iters_per_sec"].items() if k != "note"}),
          " stats:", json.dumps(d.get("sweep_stats_last")))
    if pc.get("chains"):
        print("   parity chains:", [(c["which"], c["kernel"], c["allocations_equal"] and c["counters_equal"]) for c in pc["chains"]])
