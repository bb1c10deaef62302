import sys, math, numpy as np
sys.path.insert(0,'.')
from alore_legged_manipulator_amd.host import RefSampler, BatchedMpcController
from alore_legged_manipulator_amd.nmpc import BatchedNmpc
from tests.test_host_layer import arc_polynome, exact_pose
from tests.test_closed_loop import plant_step
N, dt = 20, 0.01
v, w, xv = 1.2, 0.5, 0.1; yr, yl = -0.3, 0.3
def run(mode):
    s = RefSampler(N, dt); s.traj(arc_polynome(v, w, 0.0, [1.0,1.0,1.0], xv=xv)); s.icr(yr,yl,xv)
    pose = np.array([0.1, -0.1, 0.2])
    if mode == 'engine':
        e = BatchedNmpc(1, N)
        e.load({'W': np.tile(np.diag([10,10,0.5,0.1,0.1]).astype(np.float32), (1,N,1,1)), 'WN': np.diag([10,10,0.5]).astype(np.float32)[None],
                'od': np.tile(np.float32([xv,yr,yl]), (1,N+1,1))})
    else:
        c = BatchedMpcController(1, N, dt, delay_num=0)
        c.robots[0].traj(arc_polynome(v, w, 0.0, [1.0,1.0,1.0], xv=xv)); c.robots[0].icr(yr,yl,xv)
    for step in range(150):
        now = 0.02 + step*dt
        if mode == 'engine':
            s.odom(*pose)
            rs, ri, _ = s.refs(now, True)
            d = {'y': np.concatenate([rs[:N], ri[:N]],1)[None], 'yN': rs[N][None], 'x0': pose[None]}
            if step == 0:
                d['x'] = np.tile(pose, (1,N+1,1)); d['u'] = np.zeros((1,N,2))
            e.load(d); e.rti(1)
            u = e.fetch(('u','status','n_iter'))
            u0 = u['u'][0,0]; st = (u['status'][0], u['n_iter'][0])
        else:
            c.robots[0].odom(*pose)
            cmd = c.tick(now); u0 = cmd[0]; st = c.prediction(0)[2]
        pose = plant_step(pose, u0[0], u0[1], (xv,yr,yl), dt)
        ref = exact_pose(v,w,0.0,(0,0),xv, now+dt)
        if step % 15 == 0: print(mode, step, st, 'err', np.hypot(pose[0]-ref[0], pose[1]-ref[1]), 'u0', u0)
run('engine'); run('controller')
