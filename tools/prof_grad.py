"""Times loss_and_grad (row f3) against the forward loss on the BASELINE shapes.
    python tools/prof_grad.py [cfg B reps [jvp]] ...     default: a table over a few (cfg, B) pairs"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def run(i, B, reps=5, jvp=False):
    import torch
    import continuousnf.jl_amd as cnf
    from continuousnf.jl_amd import configs
    wl = configs.BASELINE[i]
    flat = torch.from_numpy(configs.glorot_params(wl.dims, i, 0.05)).cuda()
    xs_h, eps_h = configs.synthetic_inputs(wl, B, i)
    xs, eps = torch.from_numpy(xs_h).cuda(), torch.from_numpy(eps_h).cuda()
    icnf = configs.build(wl, jvp=jvp, sol_kwargs=configs.README_TOLERANCES)
    out = {}
    for name, fn in (("loss", lambda: cnf.loss(icnf, cnf.TrainMode(), xs, flat, {}, eps=eps)),
                     ("loss_and_grad", lambda: cnf.loss_and_grad(icnf, cnf.TrainMode(), xs, flat, {}, eps=eps))):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        out[name] = (time.perf_counter() - t0) / reps * 1e3
    st = icnf.last_stats
    print(f"cfg{i}{' (JVP compute mode)' if jvp else ''} B={B}: loss {out['loss']:.2f} ms, loss_and_grad {out['loss_and_grad']:.2f} ms "
          f"({out['loss_and_grad'] / out['loss']:.1f}x), steps {st['naccept']}+{st['nreject']}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 5, jvp=len(sys.argv) > 4 and sys.argv[4] == "jvp")
    else:
        for i, B in ((1, 32), (2, 32), (2, 4096), (3, 32), (3, 256), (3, 2048), (3, 4096), (3, 8192), (5, 32), (5, 256), (5, 2048)):
            run(i, B)
        for i, B in ((2, 32), (3, 32), (3, 2048), (3, 8192), (5, 32), (5, 2048)):
            run(i, B, jvp=True)
