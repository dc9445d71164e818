"""Which VISITING ORDER reproduces the reference's published success probabilities?  (CPU-only
analysis, no GPU: tools/order_probe.c is a plain Metropolis annealer with incremental local fields.)

    python tools/order_probe.py <model> <sweeps> <chains> <modes>     e.g.  sk_16_3 200 4096 0,2

modes: 0 typewriter (spins 0..n-1 every sweep), 1 random site selection, 2 a fresh random
permutation of the spins every sweep, 3 / 4 the colour classes of this package's plan in a fresh
random order every sweep (own order per chain / one order shared by all chains).  Same geometric
ladder between the same automatic beta estimates as annealer.anneal.  `<model>_odd` flips the
spin-inversion character of a symmetric model.  Results of round 2: profiles/r02_order_probe.txt —
typewriter order coincides with ASP-SA-1's colour order, a random permutation per sweep with the
published curves of the five symmetry-free models."""
import ctypes, json, os, subprocess, sys, time
import numpy as np, scipy.sparse
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from annealing_sign_problem_amd import operators, synthetic
from helpers import reference_route_ising
HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join('/tmp', 'asp_order_probe.so')
subprocess.check_call(['gcc', '-O3', '-march=native', '-fopenmp', '-shared', '-fPIC',
                       os.path.join(HERE, 'order_probe.c'), '-o', SO, '-lm'])
lib = ctypes.CDLL(SO)
name, sweeps, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
modes = [int(m) for m in sys.argv[4].split(',')]
cfg = synthetic.load_models()[name.replace("_odd", "")]
if name.endswith("_odd"):
    cfg = dict(cfg, basis=dict(cfg["basis"], spin_inversion=-1))
op = operators.Operator.from_config(cfg); op.basis.build()
e0, psi = op.ground_state()
amp = psi / np.linalg.norm(psi)
J = scipy.sparse.csr_matrix(reference_route_ising(op, op.basis.states, amp)); J.sort_indices()
n = J.shape[0]
A = J + J.T; A.setdiag(0); A.eliminate_zeros(); absA = abs(A)
beta0 = np.log(2) / (2 * np.asarray(absA.sum(axis=1)).ravel()).max()
beta1 = np.log(100) / (2 * absA.data.min())
betas = np.geomspace(beta0, beta1, sweeps)
exact = np.sign(psi)
pub = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'published_sa_curves.json')))['models'][name.replace("_odd", "")][str(sweeps)]
print("ground state energy", e0)
print(name, sweeps, "published", pub['acc_prob_mean'], "ASP-SA-1", pub.get('mi355x_acc_prob_mean'), flush=True)
indptr = J.indptr.astype(np.int64); indices = J.indices.astype(np.int32); data = J.data.astype(np.float64)
from annealing_sign_problem_amd import _lib
info = _lib.SaInfo(); colors = np.zeros(n, np.int32); pos = np.zeros(n, np.uint32)
_lib.check(_lib.load().asp_sa_layout_host(n, _lib.ptr(indptr), _lib.ptr(indices), _lib.ptr(data), _lib.ptr(np.zeros(n)), ctypes.byref(info), _lib.ptr(colors), _lib.ptr(pos)))
ncol = int(info.num_colors)
corder = np.argsort(colors, kind="stable").astype(np.int32)
cstart = np.concatenate([[0], np.cumsum(np.bincount(colors, minlength=ncol))]).astype(np.int64)
print("colours:", ncol, np.bincount(colors).tolist())
NAMES = ["typewriter", "random site", "random permutation", "random colour order (own per chain)", "random colour order (shared by all chains)"]
for mode in modes:
    shared = 1 if mode == 4 else 0
    cmode = 3 if mode >= 3 else mode
    spins = np.zeros((reps, n), np.int8); es = np.zeros(reps)
    t0 = time.time()
    lib.order_probe(ctypes.c_int64(n), indptr.ctypes, indices.ctypes, data.ctypes, ctypes.c_int(cmode), betas.ctypes,
                    ctypes.c_int(sweeps), ctypes.c_int(reps), ctypes.c_uint64(4242 + mode), spins.ctypes, es.ctypes,
                    ctypes.c_int(ncol), cstart.ctypes, corder.ctypes, ctypes.c_int(shared))
    acc = np.mean(spins == exact[None, :].astype(np.int8), axis=1); acc = np.maximum(acc, 1 - acc)
    p = float(np.mean(acc > 0.995))
    print("  mode %d (%s): P(acc>0.995) = %.4f +- %.4f  [%.0f s]" % (mode, NAMES[mode], p, np.sqrt(p*(1-p)/reps), time.time()-t0), flush=True)
