import sys, numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
bad = 0
for k in a.files:
    if not np.array_equal(a[k], b[k], equal_nan=True):
        d = np.argwhere(a[k] != b[k])
        print("DIFF", k, len(d), d[:5].tolist()); bad += 1
        if bad > 12: break
print("fields compared", len(a.files), "different", bad)
