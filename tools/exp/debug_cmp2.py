import numpy as np, sys
sys.path.insert(0, '.')
from vic_amd.abi import C
names = {v: k for k, v in C.items() if k.startswith("SD_")}
inames = {v: k for k, v in C.items() if k.startswith("SI_")}
L = {v: np.load("gpurun_out/cmp%s.npz" % v) for v in ("", "_a", "_z", "_p")}
for v in ("_z", "_p", ""):
    for k in ("sd", "si", "fl"):
        print("a vs", v or "plain", k, "identical:", np.array_equal(L["_a"][k], L[v][k], equal_nan=True))
a, b = L["_a"], L[""]
for s in range(14):
    df = np.argwhere(~((a["sd"][s] == b["sd"][s]) | (np.isnan(a["sd"][s]) & np.isnan(b["sd"][s]))))
    di = np.argwhere(a["si"][s] != b["si"][s])
    if len(df) or len(di):
        print("step", s, "hrus", sorted(set(df[:, 1].tolist()) | set(di[:, 1].tolist())))
        for r, c in df[:60]:
            print("   ", names.get(int(r), int(r)), c, a["sd"][s][r, c], b["sd"][s][r, c])
        for r, c in di[:20]:
            print("   ", inames.get(int(r), int(r)), c, a["si"][s][r, c], b["si"][s][r, c])
        break
