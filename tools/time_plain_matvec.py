"""Time of one matrix-free product y = Hx on sk_32_1's 601 080 390-state basis
(csrc/plain_basis.hip).  (Development aid; GPU.)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from annealing_sign_problem_amd import operators, sector_ed, synthetic  # noqa: E402

op = operators.Operator.from_config(synthetic.load_models()[sys.argv[1] if len(sys.argv) > 1 else "sk_32_1"])
matrix = sector_ed.PlainBasisMatrix(op, log=print)
x = torch.randn(matrix.n, dtype=torch.float64, device="cuda")
y = torch.empty_like(x)
matrix.matvec(x, out=y)
times = []
for _ in range(4):
    t0 = time.perf_counter()
    matrix.matvec(x, out=y)
    times.append(time.perf_counter() - t0)
bonds = len(op.bond_table()[0])
print("matvec: %s s; %.2e matrix elements/s (dimension x %d bonds / 2 off-diagonal)" % (
    " ".join("%.3f" % t for t in times), matrix.n * bonds / 2 / min(times), bonds), flush=True)
print("checksum %.12e" % float(torch.dot(x, y)))
