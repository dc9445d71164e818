"""HDF5 formats of the path's boundary (annealing_sign_problem/common.py:750-780):
the Ising-model dump and the SpinED ground-state layout, through the package's own reader/writer
(h5py is not in this image's main interpreter) — round trips, and cross-checks against the real
h5py/libhdf5 of a second interpreter where one exists."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import scipy.sparse

from annealing_sign_problem_amd import hdf5_lite

H5PY_PYTHON = "/opt/conda/bin/python3.9"


def _has_h5py_interpreter():
    if not os.path.exists(H5PY_PYTHON):
        return False
    return subprocess.run([H5PY_PYTHON, "-c", "import h5py"], capture_output=True).returncode == 0


needs_h5py = pytest.mark.skipif(not _has_h5py_interpreter(), reason="no interpreter with h5py")


def _example_tree(rng):
    return {
        "elements": rng.normal(size=37),
        "indices": rng.integers(-5, 900, size=37).astype(np.int32),
        "indptr": np.arange(12, dtype=np.int32),
        "energy": np.float64(-63.12622047596263),
        "signs": rng.integers(0, 2**63, size=3).astype(np.uint64) | np.uint64(1 << 63),
        "empty": np.zeros(0, dtype=np.float64),
        "matrix": rng.normal(size=(3, 5)).astype(np.float32),
        "hamiltonian": {"eigenvectors": rng.normal(size=(1, 20)), "eigenvalues": np.array([-2.5])},
        "basis": {"representatives": np.arange(20, dtype=np.uint64) * np.uint64(977),
                  "nested": {"deep": np.int16(-7)}},
    }


def _assert_same(a, b):
    assert sorted(a) == sorted(b)
    for key in a:
        if isinstance(a[key], dict):
            _assert_same(a[key], b[key])
        else:
            x, y = np.asarray(a[key]), np.asarray(b[key])
            assert x.dtype == y.dtype and x.shape == y.shape and x.tobytes() == y.tobytes(), key


def test_round_trip_of_every_kind_of_dataset(tmp_path):
    tree = _example_tree(np.random.default_rng(1))
    path = str(tmp_path / "a.h5")
    hdf5_lite.write(path, tree)
    _assert_same(hdf5_lite.read(path), tree)
    assert hdf5_lite.lookup(hdf5_lite.read(path), "/basis/nested/deep") == -7


def test_many_links_span_several_symbol_table_nodes(tmp_path):
    tree = {"d%03d" % i: np.full(i % 5, i, dtype=np.int64) for i in range(70)}
    path = str(tmp_path / "many.h5")
    hdf5_lite.write(path, tree)
    _assert_same(hdf5_lite.read(path), tree)


def test_rejects_what_is_not_hdf5(tmp_path):
    path = tmp_path / "junk.h5"
    path.write_bytes(b"not an hdf5 file at all")
    with pytest.raises(ValueError):
        hdf5_lite.read(str(path))


def test_ground_state_and_ising_dump_round_trip(tmp_path):
    """load_ground_state / dump_ising_model_to_hdf5 in the reference's layout (no GPU needed:
    the Hamiltonian object only stores the matrix until a plan is asked for)."""
    from annealing_sign_problem_amd import common
    from annealing_sign_problem_amd import annealer as sa

    rng = np.random.default_rng(2)
    psi = rng.normal(size=130)
    psi /= np.linalg.norm(psi)
    reps = np.sort(rng.choice(1 << 20, size=130, replace=False)).astype(np.uint64)
    gs_path = str(tmp_path / "gs.h5")
    common.save_ground_state(gs_path, psi, -12.5, reps)
    got_psi, got_e, got_reps = common.load_ground_state(gs_path)
    assert got_psi.tobytes() == psi.tobytes() and got_e == -12.5 and np.array_equal(got_reps, reps)

    j = scipy.sparse.random(130, 130, density=0.05, random_state=3, format="csr")
    j = (j + j.T).tocsr()

    class FakeOperator:
        def expectation(self, v):
            return complex(-3.25, 0.0)

    model = common.IsingModel(reps, FakeOperator(), sa.Hamiltonian(j, rng.normal(size=130)),
                              sa.signs_to_bits(np.sign(psi)))
    path = str(tmp_path / "ising.h5")
    common.dump_ising_model_to_hdf5(model, psi, path)
    raw = hdf5_lite.read(path)
    assert raw["elements"].dtype == np.float64 and raw["indices"].dtype == np.int32
    assert raw["indptr"].dtype == np.int32 and raw["signs"].dtype == np.uint64
    assert raw["energy"].shape == () and float(raw["energy"]) == -3.25
    ham, energy, signs = common.load_ising_model_from_hdf5(path)
    want = model.ising_hamiltonian.exchange
    assert (ham.exchange != want).nnz == 0 and ham.exchange.data.tobytes() == want.data.tobytes()
    assert ham.field.tobytes() == model.ising_hamiltonian.field.tobytes()
    assert energy == -3.25 and np.array_equal(signs, sa.signs_to_bits(np.sign(psi)))


@needs_h5py
def test_libhdf5_reads_what_this_package_writes(tmp_path):
    tree = _example_tree(np.random.default_rng(4))
    path = str(tmp_path / "ours.h5")
    hdf5_lite.write(path, tree)
    script = r"""
import h5py, json, sys, numpy as np
out = {}
def visit(name, obj):
    if isinstance(obj, h5py.Dataset):
        a = np.asarray(obj)
        out[name] = [str(a.dtype), list(a.shape), a.tobytes().hex()]
with h5py.File(sys.argv[1], "r") as f:
    f.visititems(visit)
print(json.dumps(out))
"""
    proc = subprocess.run([H5PY_PYTHON, "-c", script, path], capture_output=True, text=True)
    assert proc.returncode == 0, proc.stderr
    seen = json.loads(proc.stdout)

    def flatten(t, prefix=""):
        for k, v in t.items():
            if isinstance(v, dict):
                yield from flatten(v, prefix + k + "/")
            else:
                yield prefix + k, np.asarray(v)

    expected = dict(flatten(tree))
    assert sorted(seen) == sorted(expected)
    for name, a in expected.items():
        dtype, shape, payload = seen[name]
        assert np.dtype(dtype) == a.dtype and tuple(shape) == a.shape and payload == a.tobytes().hex(), name


@needs_h5py
@pytest.mark.parametrize("options", ["", "chunks=(1, 7)", "chunks=(1, 7), compression='gzip', shuffle=True",
                                     "chunks=(1, 16), compression='gzip'"])
def test_this_package_reads_what_libhdf5_writes(tmp_path, options):
    """Files as SpinED / h5py produce them: contiguous, chunked, and chunked with the deflate and
    shuffle filters."""
    path = str(tmp_path / "theirs.h5")
    script = r"""
import h5py, sys, numpy as np
rng = np.random.default_rng(9)
with h5py.File(sys.argv[1], "w") as f:
    g = f.create_group("hamiltonian")
    g.create_dataset("eigenvectors", data=rng.normal(size=(2, 50)) %s)
    g["eigenvalues"] = np.array([-3.5, -1.0])
    f.create_group("basis")["representatives"] = (np.arange(50, dtype=np.uint64) * np.uint64(12345678901))
    f["energy"] = -1.25
    f["indices"] = np.arange(9, dtype=np.int32) - 4
    f["big_endian"] = np.arange(5, dtype=">i4")
""" % ((", " + options) if options else "")
    proc = subprocess.run([H5PY_PYTHON, "-c", script, path], capture_output=True, text=True)
    assert proc.returncode == 0, proc.stderr
    tree = hdf5_lite.read(path)
    rng = np.random.default_rng(9)
    vectors = rng.normal(size=(2, 50))
    assert tree["hamiltonian"]["eigenvectors"].tobytes() == vectors.tobytes()
    assert np.array_equal(tree["hamiltonian"]["eigenvalues"], [-3.5, -1.0])
    assert np.array_equal(tree["basis"]["representatives"],
                          np.arange(50, dtype=np.uint64) * np.uint64(12345678901))
    assert float(tree["energy"]) == -1.25
    assert np.array_equal(tree["indices"], np.arange(9) - 4) and tree["indices"].dtype == np.int32
    assert np.array_equal(tree["big_endian"], np.arange(5))
    from annealing_sign_problem_amd import common

    psi, energy, reps = common.load_ground_state(path)
    assert psi.tobytes() == vectors[0].tobytes() and energy == -3.5 and reps.shape == (50,)
