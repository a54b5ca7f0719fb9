"""In-kernel cycle stamps of k_solve_bcast (needs a -DBC_STAMPS build loaded through CNFHIP_LIB): cycles wave 1 of workgroup
3 spends per phase, summed over a fixed-step solve of 32 steps (193 evaluations).

    CNFHIP_LIB=build_abl/lib_bcstamps.so python tools/bcast_stamps.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import continuousnf.jl_amd as cnf
from continuousnf.jl_amd import configs

wl = configs.BASELINE[5]
for train in (True, False):
    icnf = configs.build(wl, sol_kwargs=dict(adaptive=False, dt=1.0 / 32))
    flat = configs.glorot_params(wl.dims, 1)
    xs_h, eps_h = configs.synthetic_inputs(wl, wl.batch, 1)
    xs, eps = torch.from_numpy(xs_h).cuda(), torch.from_numpy(eps_h).cuda()
    tr = icnf.set_step_trace(64)
    mode = cnf.TrainMode() if train else cnf.TestMode()
    for _ in range(3):
        cnf.inference(icnf, mode, xs, flat, {}, eps=eps if train else None)
    torch.cuda.synchronize()
    t = tr.cpu().numpy().reshape(-1)[160:170]
    nev = 193.0
    names = ["barriers", "P1 W1 (resident)", "P2 W2 (stream)", "P3 W2^T (stream) / C (resident)", "P4 W1^T (resident)", "elementwise", "sums", "meeting (per attempt x32)"]
    print(f"config 5 {'Train' if train else 'Test'}: launches {icnf.last_stats['launches']}; cycles per evaluation: " +
          ", ".join(f"{n} {t[i] / (32.0 if i == 7 else nev):.0f}" for i, n in enumerate(names)) + f"; total per evaluation {t[:7].sum() / nev:.0f}")
    icnf.close()
