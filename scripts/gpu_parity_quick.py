"""Quick GPU-vs-oracle parity probe with step traces (development aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as G
pkg = G.load_package(); O = G.load_oracle()

def run(name, data, kinds, N, P, iters, seed, rho=0.25, T=0, q1=0, flags=None):
    rng = np.random.default_rng(seed)
    n = data[0].shape[0]; K = len(data)
    sw = pkg.Sweeper(data, kinds, N, P, n_chains=1, seed=seed, block_threads=T, q1_mode=q1)
    orc = O.Oracle(data, kinds, N, P, seed=seed, q1_mode=q1)
    s = rng.integers(1, N + 1, size=(n, K))
    n1 = int(np.floor(rho * n))
    npairs = max(1, K * (K - 1) // 2)
    ok = True
    for it in range(1, iters + 1):
        order = rng.permutation(n) + 1
        Pi = rng.gamma(1.0 / N, 1.0, size=(N, K)) + 1e-12; Pi /= Pi.sum(0)
        Phi = rng.gamma(1.0, 0.2, size=npairs)
        t0 = time.time()
        g = sw.sweep(it, s[None], order[None], n1, Pi[None], Phi[None], flags=None if flags is None else flags[None], trace=True)
        tg = time.time() - t0
        t0 = time.time()
        o = orc.sweep(it, s, order, n1, Pi, Phi, flags=None if flags is None else np.split(flags, np.cumsum([d.shape[1] for d in data])[:-1]), trace=True)
        to = time.time() - t0
        same = (g["s"][0] == o["s"]).all() and int(g["p_star"][0]) == o["p_star"]
        tr_g, tr_o = g["trace"][0], o["trace"]
        bad = np.where(~np.isclose(tr_g, tr_o, rtol=1e-9, atol=1e-9).all(axis=1))[0]
        lwok = np.allclose(g["logweight"][0], o["logweight"], rtol=1e-9, atol=1e-8)
        print(f"{name} it={it} same={same} lw={lwok} trace_first_bad={bad[:1]} gpu={tg*1e3:.1f}ms cpu={to*1e3:.1f}ms stats_g={g['stats'][0]} stats_o={ {k:v for k,v in o['stats'].items() if k!='seconds'} }")
        if len(bad):
            b = bad[0]; print("   gpu", tr_g[b], "\n   cpu", tr_o[b])
        if not same:
            d = np.argwhere(g["s"][0] != o["s"]); print("   first diff", d[:5].tolist(), "pstar", g["p_star"], o["p_star"])
            ok = False; break
        s = o["s"]
    if ok:
        eg, eo = sw.export_state(0), orc.export()
        for key in ("particle", "counts", "cluster_n", "max_id"):
            if not (eg[key] == eo[key]).all():
                print("   export mismatch", key); ok = False
    print(name, "PASS" if ok else "FAIL")
    return ok

rng = np.random.default_rng(0)
d3 = [np.vstack([rng.normal(2, 1, (50, 16)), rng.normal(-2, 1, (50, 16))]) for _ in range(3)]
allok = True
allok &= run("t5_P2", d3, ["gaussian"] * 3, 10, 2, 2, 1)
allok &= run("t5_P64", d3, ["gaussian"] * 3, 10, 64, 3, 2)
allok &= run("t5_P1024", d3, ["gaussian"] * 3, 10, 1024, 3, 3)
allok &= run("t5_P1024_T256", d3, ["gaussian"] * 3, 10, 1024, 2, 4, T=256)
n = 300; z = rng.integers(0, 3, n)
g = rng.normal(size=(n, 8)) + 2.5 * (z[:, None] - 1)
c = 1 + (rng.random((n, 6)) < (0.15 + 0.35 * z[:, None])).astype(np.int64) + (z[:, None] == 2) * rng.integers(0, 2, (n, 6))
nb = rng.geometric(0.2 + 0.25 * z[:, None], size=(n, 5)) - 1
allok &= run("mixed3", [g, c, nb], ["gaussian", "categorical", "negbinom"], 12, 256, 3, 5)
allok &= run("mixed3_q1", [g, c, nb], ["gaussian", "categorical", "negbinom"], 12, 256, 2, 6, q1=1)
fl = (rng.random(8 + 6 + 5) < 0.6).astype(np.uint8)
allok &= run("mixed3_flags", [g, c, nb], ["gaussian", "categorical", "negbinom"], 12, 200, 2, 7, flags=fl)
print("ALL", "PASS" if allok else "FAIL")
sys.exit(0 if allok else 1)
