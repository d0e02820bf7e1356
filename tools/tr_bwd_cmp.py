import sys, numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
for n in ("tr_bd1", "tr_bb1", "tr_bd0", "tr_bb0", "tr_dy1", "tr_dx"):
    x, y = a[n], b[n]
    d = np.abs(x - y)
    print(n, "rel max diff %.3e" % (d.max() / max(np.abs(x).max(), 1e-30)), "scale %.3e" % np.abs(x).max())
    if n in ("tr_bb1",):
        print("   per column block (q|k|v) x head:", np.round(d.reshape(-1, 3, 8, 16).max((0, 3)) / np.abs(x).max(), 3).tolist())
        print("   per (b,t):", np.round(d.max(-1) / np.abs(x).max(), 3).tolist())
