"""In-kernel phase stamps of the weight-gradient contraction (needs a -DWB_STAMPS build of cnf_grad.hip, loaded through
CNFHIP_LIB): cycles per 32-sample chunk that wave 0 of a full-width tile spends in each phase."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import continuousnf.jl_amd as cnf
from continuousnf.jl_amd import configs, _lib
wl = configs.BASELINE[3]
B = 8192
flat = torch.from_numpy(configs.glorot_params(wl.dims, 3, 0.05)).cuda()
xs_h, eps_h = configs.synthetic_inputs(wl, B, 3)
xs, eps = torch.from_numpy(xs_h).cuda(), torch.from_numpy(eps_h).cuda()
icnf = configs.build(wl, sol_kwargs=configs.README_TOLERANCES)
for _ in range(3):
    cnf.loss_and_grad(icnf, cnf.TrainMode(), xs, flat, {}, eps=eps)
torch.cuda.synchronize()
lib = _lib.lib()
out = (ctypes.c_ulonglong * 8)()
lib.cnf_debug_wgrad_stamps.restype = ctypes.c_int
assert lib.cnf_debug_wgrad_stamps(out) == 0
n = out[7]
names = ["wait for the chunk", "split + image stores", "barrier 1", "operand reads + MFMAs", "barrier 2", "loop", "issuing the requests"]
tot = sum(out[i] for i in range(7))
print("chunks", n, " shader cycles per chunk:")
for i, nm in enumerate(names):
    print(f"  {nm:36s} {out[i] / n:8.1f} cycles  {100.0 * out[i] / tot:5.1f} %")
print(f"  total {tot / n:.1f} cycles per chunk")
