import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import continuousnf.jl_amd as cnf
from continuousnf.jl_amd import mlj
nn = cnf.Chain(cnf.Dense(16, 48, "tanh"), cnf.Dense(48, 16, "tanh"))
icnf = cnf.construct(cnf.RNODE, nn, 8, 8, compute_mode=cnf.HIPVecJacMatrixMode(), tspan=(0.0, 13.0), steer_rate=0.1, lambda3=1e-2, rng=1)
r = np.random.default_rng(1).beta(2.0, 4.0, size=(8, 1024)).astype(np.float32)
model = mlj.ICNFModel(icnf, optimizers=(mlj.Adam(eta=1e-3),), n_epochs=5, batch_size=32)
mlj.fit(model, 0, r.T)
model = mlj.ICNFModel(icnf, optimizers=(mlj.Adam(eta=1e-3),), n_epochs=30, batch_size=32)
pr = cProfile.Profile(); pr.enable()
_, _, rep = mlj.fit(model, 0, r.T)
pr.disable()
print("ms per iteration", 1e3 * rep["stats"]["time"] / rep["stats"]["iterations"])
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
