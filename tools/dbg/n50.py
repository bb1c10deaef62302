import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from alore_legged_manipulator_amd.nmpc import BatchedNmpc
from alore_legged_manipulator_amd.scenarios import make_batch
for N,B in ((50,4096),(20,4096),(20,64),(20,1),(7,4096)):
    hb = make_batch(B, N)
    for L in (0, 16, 32, 64):
        try:
            e = BatchedNmpc(B, N, lanes_per_problem=L, slots=12)
        except Exception as ex:
            print(N, B, L, 'unsupported'); continue
        e.load(hb, slot=None)
        e.rti(1, slot=0); e.rti(1, slot=1); torch.cuda.synchronize()
        c0 = torch.cuda.Event(enable_timing=True); c1 = torch.cuda.Event(enable_timing=True)
        c0.record()
        for i in range(2, 12): e.rti(1, slot=i)
        c1.record(); torch.cuda.synchronize()
        ms = c0.elapsed_time(c1) / 10
        o = e.fetch(slot=5)
        print(f"N={N} B={B} L={L}->{e.launch_info()['lanes_per_problem']}: {ms*1e3:.1f} us/launch iters {np.bincount(o['n_iter'])} st {np.unique(o['status'])}")
        del e
