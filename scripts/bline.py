"""Print the key numbers of a bench.py JSON line read from stdin (development aid)."""
import json, sys
d = json.loads(sys.stdin.read().strip().split("\n")[-1])
lf = d.get("latency_form") or {}
print(" ".join(sys.argv[1:]), round(d["value"], 1), "it/s; sweep", round(d["sweep_kernel_ms"], 1), "ms; p50/slowest chain",
      round(d["per_chain_iters_per_sec"]["p50"], 2), round(d["per_chain_iters_per_sec"]["slowest"], 2), "busy", d.get("chain_slot_busy_frac"),
      "burn-in", d.get("burnin_iters_per_sec") and round(d["burnin_iters_per_sec"], 1), "frac", round(d["roofline"]["frac"], 3),
      "latency form p50", lf.get("per_chain_iters_per_sec", {}).get("p50"))
