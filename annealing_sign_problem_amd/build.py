"""In-tree build of libasp_hip.so (hipcc, gfx950 only).

``python -m annealing_sign_problem_amd.build`` or ``__graft_entry__.build()``.
The shared object lands next to this file so that it travels with the tree to
the GPU box; nothing is installed into site-packages.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(ROOT, "include")
# Development aid: ASP_LIB_TAG=x builds/loads libasp_hip_x.so with ASP_EXTRA_FLAGS appended
# (same-box A/B timing of kernel variants); unset in normal use.
_TAG = os.environ.get("ASP_LIB_TAG", "")
EXTRA_FLAGS = os.environ.get("ASP_EXTRA_FLAGS", "").split() if _TAG else []
LIB_NAME = "libasp_hip%s.so" % ("_" + _TAG if _TAG else "")
LIB_PATH = os.path.join(HERE, LIB_NAME)
STAMP_PATH = os.path.join(HERE, ".libasp_hip%s.stamp" % ("_" + _TAG if _TAG else ""))

SOURCES = [
    "asp_common.hip",
    "build_matrix.hip",
    "ising_elements.hip",
    "operator_apply.hip",
    "sector_basis.hip",
    "key_table.hip",
    "plain_basis.hip",
    "sparsify.hip",
    "sa_plan.cpp",
    "greedy.cpp",
    "sa_sweep.hip",
    "sa_shuffled.hip",
]

# -ffp-contract=off: the parity contract needs every multiply/add rounded on its
# own unless the source says fma().
HIPCC_FLAGS = [
    "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared",
    "-ffp-contract=off", "-fno-fast-math",
    "-Wall", "-Wextra", "-Wno-unused-parameter",
]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found; libasp_hip.so cannot be built")


def _header_deps() -> list[str]:
    deps = []
    for d in (CSRC, INCLUDE):
        for name in sorted(os.listdir(d)):
            if name.endswith((".h", ".hpp")):
                deps.append(os.path.join(d, name))
    return deps


def _included_headers(source: str) -> list[str]:
    """The repository headers `source` includes, transitively (quoted includes only)."""
    import re

    seen: dict[str, None] = {}
    todo = [source]
    while todo:
        path = todo.pop()
        with open(path, errors="replace") as f:
            names = re.findall(r'^\s*#\s*include\s*"([^"]+)"', f.read(), flags=re.M)
        for name in names:
            for d in (os.path.dirname(path), CSRC, INCLUDE):
                cand = os.path.join(d, name)
                if os.path.exists(cand):
                    if cand not in seen:
                        seen[cand] = None
                        todo.append(cand)
                    break
    return sorted(seen)


def _fingerprint(sources: list[str], headers: list[str] | None = None) -> str:
    h = hashlib.sha256()
    h.update(" ".join(HIPCC_FLAGS + EXTRA_FLAGS).encode())
    for path in list(sources) + (_header_deps() if headers is None else headers):
        # names relative to the repository: the same tree must give the same stamp wherever it
        # is mounted (the GPU box runs a copy under a scratch path)
        h.update(os.path.relpath(path, os.path.dirname(HERE)).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def fingerprint() -> str:
    """Fingerprint of the sources and flags libasp_hip.so is built from (what the stamp file next
    to the library holds).  bench.py ties committed profiler counters to it."""
    return _fingerprint([os.path.join(CSRC, s) for s in SOURCES])


# What the profiler counters of a sweep case depend on: the kernel's source file and the plan
# builder that lays out what it reads (with the headers they include).  profiles/sweep_counters.json
# ties every case to the fingerprint of its set, so that a change to an unrelated kernel file (the
# coupling build, say) does not void the counters of the sweep kernels, and a change to
# sa_shuffled.hip voids the shuffled cases only.
KERNEL_SOURCE_SETS = {
    "colour": ["sa_sweep.hip", "sa_plan.cpp"],
    "shuffled": ["sa_shuffled.hip", "sa_plan.cpp"],
}
SETS_STAMP_PATH = STAMP_PATH[:-len(".stamp")] + ".sets.stamp"


def source_set_fingerprints() -> dict:
    """{set name: fingerprint} of KERNEL_SOURCE_SETS for the sources as they are now."""
    out = {}
    for name, members in KERNEL_SOURCE_SETS.items():
        sources = [os.path.join(CSRC, s) for s in members]
        headers = sorted({h for s in sources for h in _included_headers(s)})
        out[name] = _fingerprint(sources, headers)
    return out


def built_source_set_fingerprints() -> dict | None:
    """The same, as recorded when the library on disk was built, or None."""
    import json

    try:
        with open(SETS_STAMP_PATH) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def built_fingerprint() -> str | None:
    """Fingerprint recorded when the library on disk was built, or None."""
    try:
        with open(STAMP_PATH) as f:
            return f.read().strip() or None
    except OSError:
        return None


def _compile_objects(sources: list[str], verbose: bool) -> list[str]:
    """One object per source, compiled in parallel and kept under build/obj keyed by the hash
    of (source, headers, flags): a change to one kernel file recompiles that file only."""
    from concurrent.futures import ThreadPoolExecutor

    obj_dir = os.path.join(ROOT, "build", "obj" + ("_" + _TAG if _TAG else ""))
    os.makedirs(obj_dir, exist_ok=True)
    compile_flags = [f for f in HIPCC_FLAGS if f != "-shared"] + EXTRA_FLAGS
    jobs = []
    for src in sources:
        key = _fingerprint([src], _included_headers(src))[:20]
        base = os.path.splitext(os.path.basename(src))[0]
        obj = os.path.join(obj_dir, "%s.%s.o" % (base, key))
        jobs.append((src, base, obj))

    def one(job):
        src, base, obj = job
        if os.path.exists(obj):
            return None
        for stale in os.listdir(obj_dir):
            if stale.startswith(base + ".") and stale.endswith(".o"):
                os.remove(os.path.join(obj_dir, stale))
        tmp = "%s.%d.tmp" % (obj, os.getpid())
        cmd = [hipcc(), *compile_flags, "-I", INCLUDE, "-I", CSRC, "-x", "hip", "-c", src, "-o", tmp]
        if verbose:
            print(" ".join(cmd))
        proc = subprocess.run(cmd, capture_output=True, text=True)
        if proc.returncode != 0:
            if os.path.exists(tmp):
                os.remove(tmp)
            return proc.stdout + proc.stderr
        if verbose and proc.stderr:
            sys.stderr.write(proc.stderr)
        os.replace(tmp, obj)
        return None

    workers = max(1, min(len(jobs), os.cpu_count() or 1, 8))
    with ThreadPoolExecutor(max_workers=workers) as pool:
        errors = [e for e in pool.map(one, jobs) if e]
    if errors:
        sys.stderr.write("\n".join(errors))
        raise RuntimeError("hipcc failed building " + LIB_NAME)
    return [obj for _, _, obj in jobs]


def build(force: bool = False, verbose: bool = False) -> str:
    sources = [os.path.join(CSRC, s) for s in SOURCES]
    missing = [s for s in sources if not os.path.exists(s)]
    if missing:
        raise RuntimeError("missing sources: " + ", ".join(missing))
    if _TAG and os.environ.get("ASP_NO_REBUILD") == "1" and os.path.exists(LIB_PATH):
        return LIB_PATH  # A/B against a variant built from an OLDER source tree
    fp = _fingerprint(sources)

    def current() -> bool:
        if not (os.path.exists(LIB_PATH) and os.path.exists(STAMP_PATH)):
            return False
        with open(STAMP_PATH) as f:
            if f.read().strip() != fp:
                return False
        if built_source_set_fingerprints() is None and os.access(HERE, os.W_OK):
            _write_sets_stamp()  # (a library built before the per-set stamps existed: same sources)
        return True

    if not force and current():
        return LIB_PATH
    if not os.access(HERE, os.W_OK):
        if os.path.exists(LIB_PATH):
            return LIB_PATH
        raise RuntimeError("cannot build: %s is not writable" % HERE)
    # one builder at a time (N ranks of a torch.distributed launch import this together); the
    # library is written under a temporary name and renamed, so nobody loads half a file
    import fcntl

    with open(LIB_PATH + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and current():  # somebody else built it while we waited
            return LIB_PATH
        if force:
            import shutil as _sh

            _sh.rmtree(os.path.join(ROOT, "build", "obj" + ("_" + _TAG if _TAG else "")),
                       ignore_errors=True)
        objects = _compile_objects(sources, verbose)
        tmp = "%s.%d.tmp" % (LIB_PATH, os.getpid())
        cmd = [hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared", *objects, "-o", tmp]
        if verbose:
            print(" ".join(cmd))
        proc = subprocess.run(cmd, capture_output=True, text=True)
        if proc.returncode != 0:
            sys.stderr.write(proc.stdout + proc.stderr)
            if os.path.exists(tmp):
                os.remove(tmp)
            raise RuntimeError("hipcc failed linking " + LIB_NAME)
        os.replace(tmp, LIB_PATH)
        with open(STAMP_PATH, "w") as f:
            f.write(fp)
        _write_sets_stamp()
    return LIB_PATH


def _write_sets_stamp() -> None:
    import json

    tmp = "%s.%d.tmp" % (SETS_STAMP_PATH, os.getpid())
    with open(tmp, "w") as f:
        json.dump(source_set_fingerprints(), f)
    os.replace(tmp, SETS_STAMP_PATH)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
