"""Distribution of the per-wavefront phase stamps of the last launch (ALORE_NMPC_STAMPS_DUMP=<file>):
python tools/stamp_hist.py <file>"""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.int64).reshape(-1, 8)
names = ["load+lin", "backward", "forward", "kkt", "obj+store", "total", "prediction"]
print("waves", a.shape[0])
for i, n in enumerate(names):
    v = a[:, i]
    print(f"{n:12s} mean {v.mean():8.0f}  p50 {np.percentile(v,50):8.0f}  p90 {np.percentile(v,90):8.0f}  p99 {np.percentile(v,99):8.0f}  max {v.max():8.0f}")
tot = a[:, 5]
order = np.argsort(tot)[::-1][:8]
print("slowest waves (index: phases):")
for o in order:
    print(o, a[o, :7])
# position dependence: mean total by quarter of the grid
q = a.shape[0] // 4
print("mean total by grid quarter:", [int(tot[i*q:(i+1)*q].mean()) for i in range(4)])
print("mean load by grid quarter:", [int(a[i*q:(i+1)*q, 0].mean()) for i in range(4)])
