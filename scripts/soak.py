"""Randomised parity soak: the HIP sweep (through the C ABI) against the oracle on many small random
configurations -- data types, N (up to 128), P, K, chains, quirk switches (Q1, Q2), flags, workgroup widths, launch-split thresholds, both forms of
the K > 1 sweep.
Usage: python scripts/soak.py [seconds] [seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
pkg = G.load_package(); O = G.load_oracle()


def run(budget, seed, max_cases=10**9, verbose=True):
    rng = np.random.default_rng(seed)
    t_end = time.time() + budget
    ncase = 0
    saved = {k: os.environ.get(k) for k in ("PMDI_LIGHT_IDS", "PMDI_VERY_HEAVY", "PMDI_KSPLIT", "PMDI_SETTLED", "PMDI_S2_CLS", "PMDI_STICKY")}
    try:
        while time.time() < t_end and ncase < max_cases:
            ncase += _one_case(rng)
            if verbose and ncase % 20 == 0:
                print(f"{ncase} configurations equal", flush=True)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return ncase


def _one_case(rng):
    K = int(rng.integers(1, 4)); n = int(rng.integers(30, 220)); N = int(rng.integers(2, 21))
    if rng.random() < 0.1:
        N = int(rng.integers(65, min(n, 128) + 1)) if n > 65 else N      # two 64-lane chunks per class
    P = int(rng.choice([2, 8, 33, 64, 200, 256, 512, 1024])); Cn = int(rng.integers(1, 5)); q1 = int(rng.integers(0, 2))
    q2 = int(rng.random() < 0.2)
    os.environ["PMDI_KSPLIT"] = str(int(rng.integers(0, 2)))          # both forms of the K > 1 sweep
    if os.environ.get("PMDI_SOAK_BIG"):       # fewer, larger cases: more particles, labels, datasets, observations
        K = int(rng.integers(1, 5)); n = int(rng.integers(150, 500)); N = int(rng.integers(10, 51)); P = int(rng.choice([512, 1024, 2048, 4096]))
    block = int(rng.choice([0, 0, 0, 128, 256, 512, 1024])); n1 = int(rng.integers(1, n + 1)); iters = int(rng.integers(1, 5))
    # the settled-chain kernel (where the shape has it): after a chain's first sweep or from the first sweep on (then the random start
    # hands every chain over to the general kernel's code mid-sweep), 16 or up to 32 particle classes in its tables, sticky or not
    os.environ["PMDI_SETTLED"] = str(int(rng.choice([1, 2, 2])))
    os.environ["PMDI_S2_CLS"] = str(int(rng.choice([16, 32])))
    os.environ["PMDI_STICKY"] = str(int(rng.choice([0, 3])))
    os.environ["PMDI_LIGHT_IDS"] = str(int(rng.choice([2, 10, 40, 400])))
    os.environ["PMDI_VERY_HEAVY"] = str(int(rng.choice([0, 1, 2, 128])))
    z = rng.integers(0, 3, n); sep = float(rng.choice([0.0, 1.0, 3.0]))
    data, kinds = [], []
    for k in range(K):
        kind = str(rng.choice(["gaussian", "categorical", "negbinom"])); D = int(rng.integers(1, 9))
        if kind == "gaussian": x = rng.normal(size=(n, D)) + sep * (z[:, None] - 1)
        elif kind == "categorical": x = 1 + rng.integers(0, 2, (n, D)) + (z[:, None] == 2) * rng.integers(0, 3, (n, D))
        else: x = rng.geometric(0.2 + 0.2 * z[:, None], size=(n, D)) - 1
        data.append(x); kinds.append(kind)
    sumD = sum(d.shape[1] for d in data)
    flags = (rng.random((Cn, sumD)) < 0.7).astype(np.uint8) if rng.random() < 0.3 else None
    seed = int(rng.integers(0, 2**31))
    desc = f"K={K} n={n} N={N} P={P} C={Cn} q1={q1} T={block} n1={n1} it={iters} kinds={kinds} sep={sep} light={os.environ['PMDI_LIGHT_IDS']} settled={os.environ['PMDI_SETTLED']} cls={os.environ['PMDI_S2_CLS']} sticky={os.environ['PMDI_STICKY']} vh={os.environ['PMDI_VERY_HEAVY']} flags={'y' if flags is not None else 'n'} q2={q2} split={os.environ['PMDI_KSPLIT']} seed={seed}"
    try:
        sw = pkg.Sweeper(data, kinds, N, P, n_chains=Cn, seed=seed, q1_mode=q1, q2_mode=q2, block_threads=block)
    except Exception as e:
        return 0
    orc = [O.Oracle(data, kinds, N, P, seed=seed + c, q1_mode=q1, q2_mode=q2) for c in range(Cn)]
    s = rng.integers(1, N + 1, size=(Cn, n, K))
    Dcum = np.cumsum([d.shape[1] for d in data])[:-1]
    for it in range(1, iters + 1):
        order = np.stack([rng.permutation(n) + 1 for _ in range(Cn)])
        hyp = []
        for _ in range(Cn):
            Pi = rng.gamma(1.0 / N, 1.0, size=(N, K)) + 1e-12; Pi /= Pi.sum(0)
            hyp.append((Pi, rng.gamma(1.0, 0.2, size=max(1, K * (K - 1) // 2))))
        r = sw.sweep(it, s, order, n1, np.stack([h[0] for h in hyp]), np.stack([h[1] for h in hyp]), flags=flags)
        for c in range(Cn):
            o = orc[c].sweep(it, s[c], order[c], n1, hyp[c][0], hyp[c][1], flags=None if flags is None else np.split(flags[c], Dcum))
            ok = (r["s"][c] == o["s"]).all() and int(r["p_star"][c]) == o["p_star"] and all(
                r["stats"][c][key] == o["stats"][key] for key in ("n_operations", "n_resamples", "n_clones", "max_id", "sum_classes")) \
                and np.allclose(r["logweight"][c], o["logweight"], rtol=1e-9, atol=1e-9)
            if not ok:
                os.makedirs("gpurun_out", exist_ok=True)
                np.savez("gpurun_out/soak_fail.npz", kinds=np.array(kinds), N=N, P=P, Cn=Cn, q1=q1, q2=q2, block=block, n1=n1, it=it, chain=c, seed=seed,
                         s=s, order=order, Pi=np.stack([h[0] for h in hyp]), Phi=np.stack([h[1] for h in hyp]),
                         flags=np.zeros(0) if flags is None else flags, env=np.array([os.environ["PMDI_LIGHT_IDS"], os.environ["PMDI_VERY_HEAVY"], os.environ["PMDI_KSPLIT"]]),
                         **{f"data{k}": d for k, d in enumerate(data)})
                raise AssertionError(f"MISMATCH: {desc} iteration {it} chain {c}")
        s = r["s"].copy()
    for o in orc: o.close()
    sw.close()
    return 1


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    n = run(budget, seed)
    print(f"soak done: {n} random configurations, all equal to the oracle")
