"""Times a lock-step adaptive inference of BASELINE config 3 over a ONE-rank RCCL communicator (the reduction of the three
floats is on the stream) against the plain solves of the same columns.   python tools/lockstep_time.py"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import continuousnf.jl_amd as cnf
from continuousnf.jl_amd import configs
from continuousnf.jl_amd.parallel import RcclComm
wl = configs.BASELINE[3]
flat = torch.from_numpy(configs.glorot_params(wl.dims, 12345)).cuda()
xs_h, eps_h = configs.synthetic_inputs(wl, wl.batch, 1)
xs = torch.from_numpy(np.ascontiguousarray(xs_h.T)).cuda().t()
eps = torch.from_numpy(np.ascontiguousarray(eps_h.T)).cuda().t()
ic = configs.build(wl, sol_kwargs=configs.README_TOLERANCES)
def t(n=20):
    cnf.inference(ic, cnf.TrainMode(), xs, flat, {}, eps=eps); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): cnf.inference(ic, cnf.TrainMode(), xs, flat, {}, eps=eps)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, dict(ic.last_stats)
print("one-launch solve: %.3f ms %s" % t())
comm = RcclComm(1, 0, RcclComm.unique_id(), 0)
comm.lockstep(ic)
print("lock-step over RCCL (1 rank): %.3f ms %s" % t())
comm.lockstep(ic, enable=False)
comm.close()
