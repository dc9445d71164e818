"""Where the time of one kagome_36-sized cluster goes, stage by stage (development aid):
cluster -> make_ising_model -> Hamiltonian plan -> greedy -> SA with the reference's defaults."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from annealing_sign_problem_amd import common, operators, synthetic  # noqa: E402
from annealing_sign_problem_amd import annealer as sa  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
op = operators.Operator.from_config(synthetic.kagome_lattice())
start = int("01" * 18, 2)
t = time.time()
cluster = synthetic.grow_cluster(op, start, size, seed=1)
print("grow_cluster(%d): %.3f s" % (cluster.shape[0], time.time() - t), flush=True)


class Timed:
    def __init__(self):
        self.t = {}

    def __call__(self, name, fn, *a, **kw):
        t0 = time.time()
        out = fn(*a, **kw)
        self.t[name] = self.t.get(name, 0.0) + time.time() - t0
        return out


for rep in range(2):  # second pass = warm
    T = Timed()
    log_psi = T("log_psi", synthetic.hashed_log_amplitudes, cluster)
    t0 = time.time()
    model = common.make_ising_model(cluster, op, log_psi=log_psi)
    T.t["make_ising_model"] = time.time() - t0
    h = model.ising_hamiltonian
    T("plan", h.plan)
    xg = T("greedy", common.solve_ising_model, model, mode="greedy")
    xs = T("sa(5120x64)", common.solve_ising_model, model, mode="sa")
    print("pass %d  K=%d nnz=%d:" % (rep, model.size, h.exchange.nnz),
          "  ".join("%s %.3f s" % kv for kv in T.t.items()), flush=True)
    # the fused device build on its own, then the reference's route stage by stage
    T = Timed()
    psi0 = np.ascontiguousarray(np.exp(log_psi).real)
    psi0 /= np.linalg.norm(psi0)
    T("fused ising (call)", op.device().ising, cluster, psi0)
    print("         fused ising: call %.4f s, device %.3f ms" % (T.t["fused ising (call)"],
                                                                 op.device().last_ms), flush=True)
    T("extend (call)", op.device().extend, cluster)
    print("         extend: call %.4f s, device %.3f ms" % (T.t["extend (call)"],
                                                            op.device().last_ms), flush=True)
    T = Timed()
    T("apply(device)", common._batched_apply, op, cluster)
    o, c, n = T("apply(host numpy)", lambda: (lambda r: (r[0][:, 0].copy(), r[1].real.copy(), r[2]))(op.batched_apply(cluster)))
    psi = np.exp(log_psi).real
    psi /= np.linalg.norm(psi)
    r = T("ising_elements(gpu+copies)", common.ising_elements, cluster, psi, o, c, n)
    import scipy.sparse

    def sym():
        m = scipy.sparse.csr_matrix((r[2], r[0], r[3]), shape=(cluster.shape[0],) * 2)
        m = 0.5 * (m + m.T)
        m.sort_indices()
        return m.tocoo()

    T("symmetrise(scipy)", sym)
    print("         make_ising_model split:", "  ".join("%s %.3f s" % kv for kv in T.t.items()),
          flush=True)
e_g = h.energy(xg)
e_s = h.energy(xs)
print("energies: greedy %.12g  sa %.12g" % (e_g, e_s))

# order-1 extension + sparsify of a smaller cluster (the sampled_components pipeline step)
small = synthetic.grow_cluster(op, start, max(size // 28, 50), seed=2)
log_fn = synthetic.hashed_log_amplitudes
for rep in range(2):
    T = Timed()
    m0 = T("make(order 0)", common.make_ising_model, small, op, log_psi_fn=log_fn)
    m1 = T("extension", common.make_hamiltonian_extension, m0, log_fn)
    m2 = T("sparsify", common.sparsify_using_global_cutoff, m1, 1e-4, small)
    T("plan", m2.ising_hamiltonian.plan)
    T("greedy", common.solve_ising_model, m2, mode="greedy", frozen_spins=small)
    print("order-1 pass %d  K0=%d -> %d -> %d (nnz %d):" % (rep, m0.size, m1.size, m2.size,
          m2.ising_hamiltonian.exchange.nnz), "  ".join("%s %.3f s" % kv for kv in T.t.items()),
          flush=True)
