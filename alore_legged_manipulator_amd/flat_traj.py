"""FlatTrajData -- the input format of the back_end optimiser -- for synthetic batches.

The reference produces this struct in its front end from a JPS grid path
(planning_ddr_opt/front_end/src/jps_planner/jps_planner.cpp:212-366, struct at
front_end/include/front_end/traj_representation.h:46-58).  The graph search itself is out of scope
(SURVEY.md section 2, row 7); what the batched optimiser needs is its OUTPUT for a given polyline, and
that part is restated here in NumPy for way-point paths (on a flat, obstacle-free map the pruned JPS path
is the straight segment start -> goal):

* ``sample_path``     getSampleTraj        :212-253  turn-in-place / straight-segment nodes (x, y, yaw, dyaw, ds)
* ``with_time``       getTrajsWithTime     :255-366  weighted arc length, trapezoidal timing, uniform-in-time
                                                     samples (yaw, s, t) and (x, y, yaw), boundary states
* ``evaluate_duration`` / ``evaluate_length``  :378-441  trapezoidal velocity profile

Host-side data preparation only (float64 NumPy); nothing here runs per optimiser iteration.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np


@dataclass
class FrontEndParams:
    """front_end/config/jps3ms.yaml, back_end/config/global_planning3ms.yaml, plan_manager/config/car3ms.yaml"""
    distance_weight: float = 1.40   # jps_distance_weight
    yaw_weight: float = 0.30        # jps_yaw_weight
    traj_cut_length: float = 600.0  # trajCutLength
    sample_time: float = 0.4        # timeResolution
    min_traj_num: int = 3           # mintrajNum
    max_vel: float = 3.0
    max_acc: float = 2.0


@dataclass
class FlatTraj:
    """One FlatTrajData.  M = len(traj_pts) + 1 pieces."""
    traj_pts: np.ndarray        # (M-1, 3)  yaw, s, t
    init_T: float
    positions: np.ndarray       # (M-1, 3)  x, y, yaw
    start_state: np.ndarray     # (2, 3)    [yaw | s] x [p v a]
    final_state: np.ndarray     # (2, 3)
    start_xytheta: np.ndarray   # (3,)
    final_xytheta: np.ndarray   # (3,)
    if_cut: bool = False
    meta: dict = field(default_factory=dict)

    @property
    def pieces(self) -> int:
        return len(self.traj_pts) + 1


def _normalize(ref: float, ang: float) -> float:
    while ref - ang > math.pi:
        ang += 2 * math.pi
    while ref - ang < -math.pi:
        ang -= 2 * math.pi
    return ang


def evaluate_duration(length, start_v, end_v, max_v, max_a):
    sv2, ev2, mv2 = start_v ** 2, end_v ** 2, max_v ** 2
    if start_v > max_v:
        sv2 = mv2
    if end_v > max_v:
        ev2 = mv2
    crit = (mv2 - sv2) / (2 * max_a) + (mv2 - ev2) / (2 * max_a)
    if length >= crit:
        return (max_v - start_v) / max_a + (max_v - end_v) / max_a + (length - crit) / max_v
    tmpv = math.sqrt(0.5 * (sv2 + ev2 + 2 * max_a * length))
    return (tmpv - start_v) / max_a + (tmpv - end_v) / max_a


def evaluate_length(t, length, start_v, end_v, max_v, max_a):
    sv2, ev2, mv2 = start_v ** 2, end_v ** 2, max_v ** 2
    if start_v > max_v:
        sv2 = mv2
    if end_v > max_v:
        ev2 = mv2
    crit = (mv2 - sv2) / (2 * max_a) + (mv2 - ev2) / (2 * max_a)
    if length >= crit:
        t1 = (max_v - start_v) / max_a
        t2 = t1 + (length - crit) / max_v
        if t <= t1:
            return start_v * t + 0.5 * max_a * t ** 2
        if t <= t2:
            return start_v * t1 + 0.5 * max_a * t1 ** 2 + (t - t1) * max_v
        return start_v * t1 + 0.5 * max_a * t1 ** 2 + (t2 - t1) * max_v + max_v * (t - t2) - 0.5 * max_a * (t - t2) ** 2
    tmpv = math.sqrt(0.5 * (sv2 + ev2 + 2 * max_a * length))
    tmpt = (tmpv - start_v) / max_a
    if t <= tmpt:
        return start_v * t + 0.5 * max_a * t ** 2
    return start_v * tmpt + 0.5 * max_a * tmpt ** 2 + tmpv * (t - tmpt) - 0.5 * max_a * (t - tmpt) ** 2


def sample_path(path_xy: np.ndarray, start_yaw: float, end_yaw: float) -> list:
    """Nodes (x, y, yaw, dyaw, ds): rotate towards each segment, drive it, finally rotate to end_yaw."""
    path_xy = np.asarray(path_xy, dtype=np.float64)
    nodes = [np.array([path_xy[0, 0], path_xy[0, 1], start_yaw, 0.0, 0.0])]
    th = math.atan2(path_xy[1, 1] - path_xy[0, 1], path_xy[1, 0] - path_xy[0, 0])
    th = _normalize(start_yaw, th)
    nodes.append(np.array([path_xy[0, 0], path_xy[0, 1], th, th - start_yaw, 0.0]))
    # the reference pushes the same heading a second time, computed from the reversed segment + pi
    th2 = math.atan2(path_xy[0, 1] - path_xy[1, 1], path_xy[0, 0] - path_xy[1, 0]) + math.pi
    th2 = _normalize(start_yaw, th2)
    nodes.append(np.array([path_xy[0, 0], path_xy[0, 1], th2, th2 - start_yaw, 0.0]))
    for i in range(1, len(path_xy) - 1):
        prev = nodes[-1]
        ds = math.hypot(path_xy[i, 0] - prev[0], path_xy[i, 1] - prev[1])
        nodes.append(np.array([path_xy[i, 0], path_xy[i, 1], prev[2], 0.0, ds]))
        th = math.atan2(path_xy[i + 1, 1] - path_xy[i, 1], path_xy[i + 1, 0] - path_xy[i, 0])
        th = _normalize(nodes[-1][2], th)
        nodes.append(np.array([path_xy[i, 0], path_xy[i, 1], th, th - nodes[-1][2], 0.0]))
    prev = nodes[-1]
    ds = math.hypot(path_xy[-1, 0] - prev[0], path_xy[-1, 1] - prev[1])
    nodes.append(np.array([path_xy[-1, 0], path_xy[-1, 1], prev[2], 0.0, ds]))
    th = _normalize(nodes[-1][2], end_yaw)
    nodes.append(np.array([path_xy[-1, 0], path_xy[-1, 1], th, th - nodes[-1][2], 0.0]))
    return nodes


def with_time(nodes: list, prm: FrontEndParams, start_vaj=(0.0, 0.0, 0.0), start_oaj=(0.0, 0.0, 0.0)) -> FlatTraj:
    cut = [nodes[0]]
    lengths, wlengths = [0.0], [0.0]
    all_len = all_w = 0.0
    if_cut = False
    cut_state = nodes[-1][:3].copy()
    for k in range(1, len(nodes)):
        nd = nodes[k]
        if all_len + abs(nd[4]) >= prm.traj_cut_length and nd[4] != 0:
            if_cut = True
            former = nodes[k - 1][:3]
            frac = (prm.traj_cut_length - all_len) / abs(nd[4])
            cut_state = former + (nd[:3] - former) * frac
            s5 = np.array([cut_state[0], cut_state[1], cut_state[2], frac * nd[3], prm.traj_cut_length - all_len])
            cut.append(s5)
            all_len += s5[4]
            lengths.append(all_len)
            all_w += prm.yaw_weight * abs(s5[3]) + prm.distance_weight * abs(s5[4])
            wlengths.append(all_w)
            break
        cut.append(nd)
        all_len += nd[4]
        lengths.append(all_len)
        all_w += prm.yaw_weight * abs(nd[3]) + prm.distance_weight * abs(nd[4])
        wlengths.append(all_w)
    total_t = evaluate_duration(all_w, start_vaj[0], 0.0, prm.max_vel, prm.max_acc)
    sample_t = total_t / max(int(total_t / prm.sample_time + 0.5), prm.min_traj_num)
    pts, poss = [], []
    idx = 1
    t = sample_t
    while t < total_t - 1e-3:
        arc = evaluate_length(t, all_w, start_vaj[0], 0.0, prm.max_vel, prm.max_acc)
        for k in range(idx, len(cut)):
            if wlengths[k] >= arc:
                idx = k
                l1 = wlengths[k] - arc
                l = wlengths[k] - wlengths[k - 1]
                s = lengths[k - 1] + (l - l1) / l * cut[k][4]
                yaw = cut[k - 1][2] + (l - l1) / l * cut[k][3]
                pts.append([yaw, s, t])
                poss.append([l1 / l * cut[k - 1][0] + (l - l1) / l * cut[k][0],
                             l1 / l * cut[k - 1][1] + (l - l1) / l * cut[k][1], yaw])
                break
        t += sample_t
    start_state = np.zeros((2, 3))
    final_state = np.zeros((2, 3))
    start_state[:, 0] = (cut[0][2], 0.0)
    start_state[0, 1:] = start_oaj[:2]
    start_state[1, 1:] = start_vaj[:2]
    final_state[:, 0] = (cut[-1][2], lengths[len(cut) - 1])
    return FlatTraj(traj_pts=np.array(pts, dtype=np.float64).reshape(-1, 3), init_T=float(sample_t),
                    positions=np.array(poss, dtype=np.float64).reshape(-1, 3), start_state=start_state,
                    final_state=final_state, start_xytheta=np.array(nodes[0][:3], dtype=np.float64),
                    final_xytheta=np.array(cut_state, dtype=np.float64), if_cut=if_cut,
                    meta={"total_time": total_t, "length": all_len})


def straight_goal(start_xytheta, goal_xytheta, prm: FrontEndParams | None = None) -> FlatTraj:
    """FlatTrajData for the two-point path start -> goal (what JPS + corner pruning give on a free map)."""
    prm = prm or FrontEndParams()
    path = np.array([[start_xytheta[0], start_xytheta[1]], [goal_xytheta[0], goal_xytheta[1]]], dtype=np.float64)
    return with_time(sample_path(path, float(start_xytheta[2]), float(goal_xytheta[2])), prm)


def waypoint_path(path_xy, start_yaw: float, end_yaw: float, prm: FrontEndParams | None = None) -> FlatTraj:
    prm = prm or FrontEndParams()
    return with_time(sample_path(np.asarray(path_xy, dtype=np.float64), start_yaw, end_yaw), prm)


def monte_carlo_goals(count: int, seed: int = 20260206, prm: FrontEndParams | None = None) -> list:
    """SURVEY.md section 8(d), config 5 inputs: start pose ~ U([-5, 5]^2 x [-pi, pi]), goal 3-8 m away at a
    random bearing with a random final heading."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(count):
        sx, sy = rng.uniform(-5, 5, 2)
        syaw = rng.uniform(-math.pi, math.pi)
        dist = rng.uniform(3.0, 8.0)
        bearing = rng.uniform(-math.pi, math.pi)
        gyaw = rng.uniform(-math.pi, math.pi)
        out.append(straight_goal((sx, sy, syaw), (sx + dist * math.cos(bearing), sy + dist * math.sin(bearing), gyaw), prm))
    return out
