"""Timing of the co-clustering-count kernel on the cfg2 shape: S pooled samples, n = 10000."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
pkg = G.load_package()
from particlemdi_jl_amd import psm
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
K = int(sys.argv[3]) if len(sys.argv) > 3 else 1
g = torch.Generator(device="cuda"); g.manual_seed(1)
smp = torch.randint(0, 20, (S, K, n), dtype=torch.uint8, device="cuda", generator=g)
NL = int(sys.argv[4]) if len(sys.argv) > 4 else 20
out = psm.psm_counts_device(smp, 0, n, NL); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3):
    out = psm.psm_counts_device(smp, 0, n, NL)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 3
cmp_ = float(S) * K * n * n
print(f"psm counts S={S} K={K} n={n} n_labels={NL} ({'MFMA int8' if 1 <= NL <= 64 else 'byte compares'}): {ms:.2f} ms per call, {cmp_ / ms / 1e9:.2f} T label-compares/s, "
      f"HBM bytes (samples read once per tile row/col + counts written) ~{(K*n*n*4 + S*K*n*2*(n/64)) / 1e9:.2f} GB "
      f"-> {(K*n*n*4 + S*K*n*2*(n/64)) / ms / 1e6:.0f} GB/s")
t0 = time.perf_counter()
ref = torch.zeros((n, n), dtype=torch.int32, device="cuda")
for t in range(min(S, 50)):
    lab = smp[t, 0]
    ref += (lab[:, None] == lab[None, :])
torch.cuda.synchronize()
print(f"torch loop (the host mirror's formulation) on the same GPU: {(time.perf_counter() - t0) / min(S, 50) * S * 1e3:.0f} ms extrapolated to S samples")
if S <= 50:
    print("equal to the torch loop:", bool((out[0] == ref).all()))
