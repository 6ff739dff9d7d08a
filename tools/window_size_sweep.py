import json, os, subprocess, sys
sys.path.insert(0, '.')
def child():
    import numpy as np, torch
    from gecco_amd import _native as nat, synth
    wl = synth.workload("C2")
    model = nat.Model.from_tables(wl["w"], wl["trans"])
    dev = torch.device("cuda", 0)
    res = {}
    for n in (200_000, 1_000_000):
        rng = np.random.default_rng(n)
        cptr, gptr, attr = synth.synth_contigs(rng, [200] * (n // 200), model.num_attrs)
        gp, at = torch.from_numpy(gptr).to(dev), torch.from_numpy(attr).to(dev)
        p = torch.zeros(n, dtype=torch.float64, device=dev)
        for W in (5, 10, 32):
            plan = nat.Plan(model, cptr, W, 1, True, device=0)
            res[f"{n}/W{W}"] = round(plan.time_windowed(gp.data_ptr(), at.data_ptr(), p.data_ptr(), 1, 0, warmup=5, iters=50) * 1e3, 2)
    print("RES", json.dumps(res))
if len(sys.argv) > 1:
    child()
else:
    for t in ("1", "2", "1", "2"):
        out = subprocess.run([sys.executable, __file__, "child"], capture_output=True, text=True, env=dict(os.environ, GECCO_CRF_TILES_PER_WG=t))
        print("tiles", t, [l for l in out.stdout.splitlines() if l.startswith("RES")] or out.stderr[-300:])
