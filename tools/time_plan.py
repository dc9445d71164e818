"""Host time of the plan layout (asp_sa_layout_host) with 1 and N host threads (development aid)."""
import ctypes
import os
import subprocess
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 1 and sys.argv[1] == "child":
    from annealing_sign_problem_amd import _lib, synthetic

    lib = _lib.load()
    for k, deg in ((100000, 23.0), (100000, 8.0)):
        J, h, _ = synthetic.planted_cluster(k, mean_degree=deg, seed=1)
        Jc = J.tocsr()
        ip, ix = Jc.indptr.astype(np.int64), Jc.indices.astype(np.int32)
        info = _lib.SaInfo()
        best = 1e9
        for _ in range(4):
            t = time.time()
            lib.asp_sa_layout_host(k, _lib.ptr(ip), _lib.ptr(ix), _lib.ptr(Jc.data), _lib.ptr(h),
                                   ctypes.byref(info), None, None)
            best = min(best, time.time() - t)
        print("  K=%d dbar=%.0f: layout %.1f ms" % (k, deg, best * 1e3), flush=True)
else:
    for threads in ("1", "4", "8", "16"):
        print("ASP_HOST_THREADS=%s" % threads, flush=True)
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, ASP_HOST_THREADS=threads))
