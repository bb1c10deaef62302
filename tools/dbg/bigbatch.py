import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from alore_legged_manipulator_amd.nmpc import BatchedNmpc
from alore_legged_manipulator_amd.scenarios import make_batch
N = 20
for B in (32768,):
    hb = make_batch(B, N)
    for L in (4, 8, 16, 32):
        for ws in (0, 8):
            try:
                e = BatchedNmpc(B, N, lanes_per_problem=L, slots=12, warm_start_steps=ws)
            except Exception as ex:
                print(B, L, 'unsupported', ex); continue
            e.load(hb, slot=None)
            e.rti(1, slot=0); e.rti(1, slot=1); torch.cuda.synchronize()
            c0 = torch.cuda.Event(enable_timing=True); c1 = torch.cuda.Event(enable_timing=True)
            c0.record()
            for i in range(2, 12): e.rti(1, slot=i)
            c1.record(); torch.cuda.synchronize()
            ms = c0.elapsed_time(c1) / 10
            print(f"B={B} L={L} ws={ws}: {ms*1e3:.1f} us/launch  {B/ms/1e3:.3e} solves/s  hbm {4192*B/ms/1e6/8000*100:.1f}%  info={e.launch_info()['lds_bytes_per_block']}")
            del e
