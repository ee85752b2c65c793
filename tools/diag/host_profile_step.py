import os, sys, cProfile, pstats, importlib.util
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
spec = importlib.util.spec_from_file_location("ts", os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "tools", "train_synthetic.py"))
ts = importlib.util.module_from_spec(spec); spec.loader.exec_module(ts)
dev = torch.device("cuda:0")
su = ts.fateavatar_setup(100_000, 512, dev)
st, cams, posed, gts, nf = su["st"], su["cams"], su["posed"], su["gts"], su["n_frames"]
for it in range(20):
    st.step(cams[it % nf], posed[it % nf], gts[it % nf])
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for it in range(3000):
    f = it % nf
    st.step(cams[f], posed[f], gts[f])
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
