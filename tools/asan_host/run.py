"""Drives the host-only plan layout and greedy tree (built with ASan/UBSan by ../asan_host.sh)
over degenerate, sparse and dense instances."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from annealing_sign_problem_amd import synthetic, _lib
lib = ctypes.CDLL(sys.argv[1])
def ptr(a): return a.ctypes.data_as(ctypes.c_void_p)
rng=np.random.default_rng(0)
cases=[synthetic.planted_cluster(k,seed=k,mean_degree=d) [:2] for k,d in ((1,1.0),(2,1.0),(65,3.0),(1000,23.0),(5000,6.0),(20000,23.0))]
cases.append(synthetic.sk_cluster(700,degree=300,seed=3))
import scipy.sparse
cases.append((scipy.sparse.csr_matrix((5,5)),np.zeros(5)))
cases.append((scipy.sparse.identity(7,format='csr'),np.ones(7)))
# round 3: the symmetric-J short cut and the general merge on the same couplings (upper triangle),
# and a matrix that is symmetric but for one missing mirror element
J,h=synthetic.planted_cluster(3000,seed=9,mean_degree=12.0)[:2]
cases.append((scipy.sparse.triu(2.0*J,1).tocsr(),h))
lil=scipy.sparse.lil_matrix(J); i,j=(int(v[0]) for v in J.nonzero()); lil[j,i]=0.0
cases.append((scipy.sparse.csr_matrix(lil),h))
for J,h in cases:
    J=scipy.sparse.csr_matrix(J); J.sort_indices(); n=J.shape[0]
    ip=J.indptr.astype(np.int64); ix=J.indices.astype(np.int32); d=J.data.astype(np.float64); h=np.ascontiguousarray(h,dtype=np.float64)
    info=_lib.SaInfo(); col=np.zeros(max(n,1),np.int32); pos=np.zeros(max(n,1),np.uint32)
    rc=lib.asp_sa_layout_host(ctypes.c_uint64(n),ptr(ip),ptr(ix),ptr(d),ptr(h),ctypes.byref(info),ptr(col),ptr(pos))
    words=(n+63)//64; x=np.zeros(max(words,1),np.uint64)
    rc2=lib.asp_sa_greedy_tree_host(ctypes.c_uint64(n),ptr(ip),ptr(ix),ptr(d),ptr(h),ptr(x))
    print(n,"layout rc",rc,"colors",info.num_colors,"greedy rc",rc2)
print("done")
