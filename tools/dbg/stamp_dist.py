"""Distribution of the per-wavefront phase stamps of one B=4096 launch (ALORE_NMPC_STAMPS=1 ALORE_NMPC_STAMPS_DUMP=file)."""
import sys, numpy as np
a = np.fromfile(sys.argv[1], dtype=np.int64).reshape(-1, 8)
names = ["load+lin", "backward", "forward", "kkt", "obj", "total", "prediction"]
for i, n in enumerate(names):
    v = a[:, i]
    print(f"{n:11s} mean {v.mean():8.0f}  p10 {np.percentile(v,10):8.0f}  p50 {np.percentile(v,50):8.0f}  p90 {np.percentile(v,90):8.0f}  p99 {np.percentile(v,99):8.0f}  max {v.max():8.0f}")
t = a[:, 5]
slow = np.argsort(t)[-8:]
print("slowest wavefronts (index, phases):")
for i in slow: print("  ", i, a[i, :7].tolist())
# correlation of slow totals with position in the grid
print("mean total by eighth of the grid:", [int(t[k::8].mean()) for k in range(8)], "(workgroup index mod 8 ~ XCD)")
print("mean total first/last quarter of the grid:", int(t[:len(t)//4].mean()), int(t[-len(t)//4:].mean()))
