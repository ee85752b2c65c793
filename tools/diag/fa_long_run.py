"""GPU box: 4 000 FateAvatar steps (one and four frames per step) with the maintenance schedule compressed (densify every 700,
prune every 900, opacity reset every 1 500 steps) and keep_coherent on: the set grows 100 k -> ~124 k, stays stored sorted, the
graph is re-captured after every change of the set, nothing overflows, every parameter stays finite."""
import os, sys, time, json, importlib.util
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
spec = importlib.util.spec_from_file_location("ts", os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "tools", "train_synthetic.py"))
ts = importlib.util.module_from_spec(spec); spec.loader.exec_module(ts)
dev = torch.device("cuda:0")
for K in (1, 4):
    su = ts.fateavatar_setup(100_000, 512, dev, views=8, views_per_step=K, keep_coherent=True)
    st, cams, posed, gts, nf = su["st"], su["cams"], su["posed"], su["gts"], su["n_frames"]
    cfg = dict(densify_interval=700, increase_num=5000, prune_interval=900, opacity_reset_interval=1500, max_points_num=140_000)
    t0 = time.time(); did_all = {}; losses = []
    for it in range(1, 4001):
        if K == 1:
            f = it % nf; loss = st.step(cams[f], posed[f], gts[f])
        else:
            fs = [(it * K + k) % nf for k in range(K)]
            loss = st.step([cams[f] for f in fs], [posed[f] for f in fs], [gts[f] for f in fs])[0]
        did = st.maintain(it, cfg)
        for k, v in did.items(): did_all[k] = did_all.get(k, 0) + (v if isinstance(v, int) and not isinstance(v, bool) else 1)
        if it % 500 == 0: losses.append(round(float(loss), 6))
    torch.cuda.synchronize(); st.check()
    sorted_now = bool(torch.equal(st.coherent_order(), torch.arange(st.pc.P, device=dev)))
    print(json.dumps(dict(K=K, P=st.pc.P, did=did_all, overflows=st.overflows, losses=losses, stored_sorted=sorted_now,
                          finite=bool(torch.isfinite(st.pc.flat).all()), seconds=round(time.time() - t0, 1))), flush=True)
