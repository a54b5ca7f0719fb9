"""In-kernel cycle stamps of k_solve_wave (needs a -DWV_STAMPS build loaded through CNFHIP_LIB): per step attempt, the
cycles wave 0 spends in the six stage evaluations + error estimate, in the meeting, and in the controller.

    CNFHIP_LIB=build_abl/lib_wvstamps.so python tools/wave_stamps.py [cfg ...]      (WV_BATCH=n: that batch instead of the config's)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import continuousnf.jl_amd as cnf
from continuousnf.jl_amd import configs

for i in ([int(a) for a in sys.argv[1:]] or [1, 2]):
    wl = configs.BASELINE[i]
    if os.environ.get("WV_BATCH"):
        import dataclasses
        wl = dataclasses.replace(wl, batch=int(os.environ["WV_BATCH"]))
    icnf = configs.build(wl, sol_kwargs=dict(adaptive=False, dt=(wl.tspan[1] - wl.tspan[0]) / 32))
    flat = configs.glorot_params(wl.dims, 1)
    xs_h, eps_h = configs.synthetic_inputs(wl, wl.batch, 1)
    xs, eps = torch.from_numpy(xs_h).cuda(), torch.from_numpy(eps_h).cuda()
    tr = icnf.set_step_trace(64)
    for _ in range(3):
        cnf.inference(icnf, cnf.TrainMode(), xs, flat, {}, eps=eps)
    torch.cuda.synchronize()
    t = tr.cpu().numpy()[:32]
    print(f"config {i} B={wl.batch} launches={icnf.last_stats['launches']}: stages {np.median(t[:, 0]):.0f}  meeting {np.median(t[:, 1]):.0f}  "
          f"controller {np.median(t[:, 2]):.0f} cycles per attempt (median of 32; min {t[:, 0].min():.0f} / {t[:, 1].min():.0f})")
    icnf.close()
