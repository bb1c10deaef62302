import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from alore_legged_manipulator_amd.backend import BatchedMSPlanner
from alore_legged_manipulator_amd.flat_traj import monte_carlo_goals
from oracle.backend_driver import BackendOracle, EsdfGrid
orc = BackendOracle(); grid = EsdfGrid.free(half=20.0)
fts = monte_carlo_goals(96, seed=20260206)
pl = BatchedMSPlanner(96, 16); pl.set_map(grid.dist, grid.x_lo, grid.y_lo, grid.res)
res = pl.minco_plan(fts)
dev = []
for b, ft in enumerate(fts):
    ref = orc.minco_plan(grid, ft)
    dev.append((abs(res["cost"][b] - ref["cost"]) / abs(ref["cost"]), b, res["cost"][b], ref["cost"], int(res["evals"][b]), ref.get("evals"), int(res["alm_rounds"][b]), ref.get("alm_rounds"), int(res["lbfgs_ret"][b]), ref.get("lbfgs_ret"), res["T"][b].sum(), ref["T"].sum()))
dev.sort(reverse=True)
for d in dev[:6]: print(d)
