"""One-off: long device closed loop (trajectories end, robots reach their goals, commands go to zero)."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from alore_legged_manipulator_amd.nmpc import BatchedNmpc
from alore_legged_manipulator_amd.host import Polynome
from alore_legged_manipulator_amd.scenarios import arc_pose
B, N = 2048, 20
rng = np.random.default_rng(1)
vw = rng.uniform([0.4, -0.9], [1.6, 0.9], (B, 2))
T = np.array([1.0, 1.0, 1.0]); Tc = np.cumsum(T)
msgs = [Polynome(np.stack([w * Tc[:-1], v * Tc[:-1]], 1), T, [0, 0, w, v, 0, 0], [w * Tc[-1], v * Tc[-1], w, v, 0, 0], [0, 0, 0], [-0.3, 0.3, 0.1], 0.0) for v, w in vw]
e = BatchedNmpc(B, N, diagnostics=False)
W = np.tile(np.diag([10, 10, 0.5, 0.1, 0.1]).astype(np.float32), (B, N, 1, 1)); WN = np.tile(np.diag([10, 10, 0.5]).astype(np.float32), (B, 1, 1))
e.load({"W": W, "WN": WN})
e.refs_init(max_pieces=4, max_checkpoints=40); e.refs_set_polynomes(np.arange(B), msgs)
e.plant_init(); e.plant_set_state(rng.uniform(-0.05, 0.05, (B, 3)), np.tile([0.1, -0.3, 0.3], (B, 1)))
worst_status = 0
for blk in range(6):
    e.closed_loop_run(0.01 * (blk * 100 + 1), 0.01, 100)
    torch.cuda.synchronize()
    pose, vwp, goal = e.plant_get_state()
    st = e.t["status"].cpu().numpy(); worst_status = max(worst_status, int((st != 0).sum()))
    ref = np.stack([arc_pose(v, w, 0.1, min(0.01 * (blk + 1) * 100, 3.0)) for v, w in vw])
    err = np.hypot(pose[:, 0] - ref[:, 0], pose[:, 1] - ref[:, 1])
    print(f"t = {(blk+1):d}.00 s: finite {np.isfinite(pose).all()}  at_goal {int(goal.sum())}/{B}  unsolved {int((st!=0).sum())}  position error mean {err.mean():.3f} max {err.max():.3f} m  |v| max {np.abs(vwp[:,0]).max():.2f}")
