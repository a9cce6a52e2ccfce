"""Replays the case scripts/soak.py saved on a mismatch (gpurun_out/soak_fail.npz) with per-observation traces."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
pkg = G.load_package(); O = G.load_oracle()
z = np.load(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/soak_fail.npz")
kinds = [str(k) for k in z["kinds"]]; K = len(kinds)
data = [z[f"data{k}"] for k in range(K)]
N, P, Cn, q1, q2, block, n1, it, seed = (int(z[k]) for k in ("N", "P", "Cn", "q1", "q2", "block", "n1", "it", "seed"))
os.environ["PMDI_LIGHT_IDS"], os.environ["PMDI_VERY_HEAVY"], os.environ["PMDI_KSPLIT"] = (str(x) for x in z["env"])
for k, v in [a.split("=") for a in sys.argv[2:]]:
    os.environ[k] = v
flags = None if z["flags"].size == 0 else z["flags"]
print("case:", dict(N=N, P=P, Cn=Cn, q1=q1, q2=q2, block=block, n1=n1, it=it, seed=seed, kinds=kinds, n=data[0].shape[0], env=list(z["env"]), over=sys.argv[2:]))
sw = pkg.Sweeper(data, kinds, N, P, n_chains=Cn, seed=seed, q1_mode=q1, q2_mode=q2, block_threads=block)
print("split", sw.split, "T", sw.block_threads, "lds", sw.lds_bytes)
r = sw.sweep(it, z["s"], z["order"], n1, z["Pi"], z["Phi"], flags=flags, trace=True)
Dcum = np.cumsum([d.shape[1] for d in data])[:-1]
for c in range(Cn):
    o = O.Oracle(data, kinds, N, P, seed=seed + c, q1_mode=q1, q2_mode=q2).sweep(it, z["s"][c], z["order"][c], n1, z["Pi"][c], z["Phi"][c],
                                                                            flags=None if flags is None else np.split(flags[c], Dcum), trace=True)
    same = (r["s"][c] == o["s"]).all()
    bad = np.where(~np.isclose(r["trace"][c], o["trace"], rtol=1e-9, atol=1e-9).all(axis=1))[0]
    print(f"chain {c}: allocations equal {same}; p_star {int(r['p_star'][c])} vs {o['p_star']}; stats gpu {r['stats'][c]} cpu {o['stats']}")
    if bad.size:
        b = bad[0]
        print(f"   first diverging swept observation {b} of {len(o['trace'])}: gpu {r['trace'][c][b]} cpu {o['trace'][b]}")
        if b > 0:
            print(f"   the one before: gpu {r['trace'][c][b-1]} cpu {o['trace'][b-1]}")
