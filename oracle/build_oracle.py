"""Build recipe for the oracle (test infrastructure, never shipped).

Two artefacts, both git-ignored:

* ``oracle/liboracle.so``   — this directory's own C restatements
  (``build_matrix_oracle.c``, ``sa_oracle.c``, ``greedy_oracle.c``,
  ``operator_oracle.c``).
* ``oracle/_ref/libbuild_matrix_ref.so`` — the REFERENCE's coupling build,
  compiled from its sources where they lie (``/root/reference/cbits``); no
  reference source is copied into this repository.  Only built when the
  reference checkout is present (it is absent on the GPU box, which uses the
  file built here and shipped with the snapshot).

Flags: ``-std=c11`` puts gcc in ISO mode, i.e. ``-ffp-contract=off`` — the
semantics of the reference as its own cffi build compiles it (no ``-mfma`` on
x86-64 baseline ⇒ no fused multiply-add can be emitted).
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE_CBITS = "/root/reference/cbits"

ORACLE_SOURCES = ["build_matrix_oracle.c", "sa_oracle.c", "greedy_oracle.c", "operator_oracle.c"]
ORACLE_LIB = os.path.join(HERE, "liboracle.so")
REF_DIR = os.path.join(HERE, "_ref")
REF_LIB = os.path.join(REF_DIR, "libbuild_matrix_ref.so")

# x86-64-v3 (AVX2 + FMA): fma() in sa_oracle.c becomes one instruction; every
# server CPU this runs on has it.  -ffp-contract=off: no implicit contraction.
ORACLE_CFLAGS = [
    "-O2", "-std=c11", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
    "-mavx2", "-mfma", "-fopenmp", "-Wall", "-Wextra",
]
REF_CFLAGS = ["-O3", "-std=c11", "-fPIC", "-shared"]


def _stale(target: str, sources: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources + [os.path.abspath(__file__)])


def _run(cmd: list[str]) -> None:
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        sys.stderr.write(proc.stdout + proc.stderr)
        raise RuntimeError("command failed: " + " ".join(cmd))


def build_oracle(force: bool = False) -> str:
    sources = [os.path.join(HERE, s) for s in ORACLE_SOURCES if os.path.exists(os.path.join(HERE, s))]
    if force or _stale(ORACLE_LIB, sources):
        _run(["gcc", *ORACLE_CFLAGS, *sources, "-o", ORACLE_LIB, "-lm"])
    return ORACLE_LIB


def build_reference(force: bool = False) -> str | None:
    """Compile the reference's cbits/build_matrix.c in place -> oracle/_ref/."""
    src = os.path.join(REFERENCE_CBITS, "build_matrix.c")
    if not os.path.exists(src):
        return REF_LIB if os.path.exists(REF_LIB) else None
    os.makedirs(REF_DIR, exist_ok=True)
    if force or _stale(REF_LIB, [src, os.path.join(REFERENCE_CBITS, "build_matrix.h")]):
        _run(["gcc", *REF_CFLAGS, "-I", REFERENCE_CBITS, src, "-o", REF_LIB, "-lm"])
    return REF_LIB


if __name__ == "__main__":
    print(build_oracle(force="--force" in sys.argv))
    print(build_reference(force="--force" in sys.argv))
