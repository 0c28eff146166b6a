"""qsim -> amplitude network front end (cotengra_amd/circuits.py) and the
Sycamore m10 configuration (BASELINE.json configs[2]): network built from the
reference's qsim gate list, tree + slicing found by the reference's optimizer,
golden amplitude computed by the reference (tests/golden/gen/make_m10.py)."""
import json
import os

import numpy as np
import pytest

import cotengra_amd as ca
from cotengra_amd.circuits import circuit_to_network, gate_matrix, parse_qsim
from oracle import contract_ref as orc

HERE = os.path.dirname(os.path.abspath(__file__))
M10_TREE = os.path.join(HERE, "golden", "trees", "sycamore_m10.json")
M10_ARRAYS = os.path.join(HERE, "golden", "sycamore_m10_arrays.npz")
M10_EXPECTED = os.path.join(HERE, "golden", "sycamore_m10_expected.npz")

QSIM = """4
0 hz_1_2 0
0 x_1_2 1
0 y_1_2 2
0 x_1_2 3
1 rz 0 0.3
1 rz 1 -1.1
1 fs 0 1 1.5157741664069029 0.5567125777723744
2 y_1_2 0
2 hz_1_2 1
2 x_1_2 2
3 rz 1 2.0
3 rz 2 0.7
3 fs 1 2 1.2 -0.4
3 fs 0 3 0.9 0.2
4 x_1_2 0
4 y_1_2 3
"""


def statevector(n, gates):
    psi = np.zeros([2] * n, complex)
    psi[(0,) * n] = 1
    for name, qs, ps in gates:
        U = gate_matrix(name, ps)
        if len(qs) == 1:
            psi = np.moveaxis(np.tensordot(U, psi, axes=([1], [qs[0]])), 0, qs[0])
        else:
            psi = np.moveaxis(
                np.tensordot(U.reshape(2, 2, 2, 2), psi, axes=([2, 3], [qs[0], qs[1]])), [0, 1], list(qs))
    return psi


def test_gates_are_unitary():
    for name, ps in (("x_1_2", ()), ("y_1_2", ()), ("hz_1_2", ()), ("rz", (0.37,)), ("fs", (1.1, -0.6))):
        U = gate_matrix(name, ps)
        assert np.allclose(U @ U.conj().T, np.eye(len(U)))
    # sqrt gates square to X, Y, W up to a global phase
    X = np.array([[0, 1], [1, 0]]); Y = np.array([[0, -1j], [1j, 0]])
    for name, P in (("x_1_2", X), ("y_1_2", Y), ("hz_1_2", (X + Y) / np.sqrt(2))):
        U2 = gate_matrix(name) @ gate_matrix(name)
        ph = U2[0, 1] / P[0, 1]
        assert abs(abs(ph) - 1) < 1e-12 and np.allclose(U2, ph * P)


@pytest.mark.parametrize("simplify", [False, True])
def test_network_amplitudes_match_statevector(simplify):
    n, gates = parse_qsim(QSIM)
    assert n == 4 and len(gates) == 16
    psi = statevector(n, gates)
    assert abs(np.vdot(psi, psi) - 1) < 1e-12
    for bits in ("0000", "1010", "0111", "1111"):
        inputs, output, sd, arrays = circuit_to_network(n, gates, bits, simplify=simplify)
        if simplify:
            assert all(len(t) >= 3 for t in inputs) or len(inputs) == 1
        tree = ca.array_contract_tree(inputs, output, sd)
        amp = orc.contract(tree, arrays)
        assert abs(amp - psi[tuple(int(b) for b in bits)]) < 1e-13


def random_circuit(n, depth, seed):
    """Sycamore-style layers: a random single-qubit gate on every qubit, then
    fSim gates on a brick pattern of neighbouring pairs."""
    rng = np.random.default_rng(seed)
    gates = []
    for d in range(depth):
        for q in range(n):
            name = ("x_1_2", "y_1_2", "hz_1_2", "rz")[int(rng.integers(0, 4))]
            gates.append((name, (q,), (float(rng.normal()),) if name == "rz" else ()))
        for q in range(d % 2, n - 1, 2):
            gates.append(("fs", (q, q + 1), (float(rng.normal()), float(rng.normal()))))
    return gates


@pytest.mark.parametrize("bits", ["0??1", "????", "1?0?"])
@pytest.mark.parametrize("simplify", [False, True])
def test_open_qubits_give_amplitude_batches(bits, simplify):
    n, gates = parse_qsim(QSIM)
    psi = statevector(n, gates)
    inputs, output, sd, arrays = circuit_to_network(n, gates, bits, simplify=simplify)
    assert len(output) == bits.count("?")
    tree = ca.array_contract_tree(inputs, output, sd)
    amps = orc.contract(tree, arrays)
    sel = tuple(slice(None) if b == "?" else int(b) for b in bits)
    assert np.allclose(amps, psi[sel], atol=1e-13)


@pytest.mark.gpu
def test_batched_amplitudes_on_gpu_sliced():
    """12 qubits, 6 layers, 5 open qubits: a 32-amplitude batch.  The tree is
    sliced on one output (outer) and two inner indices, so the HIP path goes
    through chunk scatter + inner accumulation; checked against the dense
    state vector."""
    n = 12
    gates = random_circuit(n, 6, seed=7)
    psi = statevector(n, gates)
    bits = "0?1?0??10?01"
    inputs, output, sd, arrays = circuit_to_network(n, gates, bits, simplify=True, dtype="complex128")
    tree = ca.array_contract_tree(inputs, output, sd)
    sel = tuple(slice(None) if b == "?" else int(b) for b in bits)
    ref = psi[sel]
    assert np.allclose(orc.contract(tree, arrays), ref, atol=1e-12)
    big = max((p for p, _, _ in tree.traverse()), key=tree.get_size)
    inner = [ix for ix in tree.get_legs(big) if ix not in tree.output][:2]
    for ix in [tree.output[1]] + inner:
        tree.remove_ind_(ix)
    assert tree.nslices == 8
    for dtype, tol in (("complex128", 1e-11), ("complex64", 2e-5)):
        got = np.asarray(tree.contract([a.astype(dtype) for a in arrays]))
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() <= tol * np.abs(ref).max()


def m10():
    rec = ca.load_network(M10_TREE)
    tree = ca.tree_from_record(rec)
    z = np.load(M10_ARRAYS)
    arrays = [z[f"t{i}"] for i in range(tree.N)]
    return rec, tree, arrays, np.load(M10_EXPECTED)


needs_m10 = pytest.mark.skipif(not os.path.exists(M10_EXPECTED), reason="m10 fixture not generated")


@needs_m10
def test_m10_fixture_oracle_slices():
    rec, tree, arrays, exp = m10()
    assert tree.N == 170 and tree.nslices >= 64 and rec["stats"]["nslices"] == tree.nslices
    assert sorted(len(t) for t in tree.inputs)[0] >= 3
    got = orc.contract_slice(tree, arrays, 1)
    assert abs(got - exp["slice1"]) <= 1e-12 * abs(exp["slice1"])


@needs_m10
@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["complex128", "complex64"])
def test_m10_full_amplitude_on_gpu(dtype):
    """All slices of the m10 amplitude on the device vs the reference's full
    CPU contraction (the north-star gate: 1e-5 relative, both precisions)."""
    rec, tree, arrays, exp = m10()
    xs = [a.astype(dtype) for a in arrays]
    amp = complex(np.asarray(tree.contract(xs)))
    ref = complex(exp["amplitude"])
    assert abs(amp - ref) <= (1e-10 if dtype == "complex128" else 1e-5) * abs(ref)
    for key in exp.files:
        if key.startswith("slice"):
            i = int(key[5:])
            got = complex(np.asarray(tree.contract_slice(xs, i)))
            tol = 1e-10
            if dtype == "complex64":
                # north-star 1e-5, or 8x what numpy itself loses in single precision
                np64 = complex(orc.contract_slice(tree, xs, i))
                tol = max(1e-5, 8.0 * abs(np64 - complex(exp[key])) / abs(exp[key]))
            assert abs(got - complex(exp[key])) <= tol * abs(exp[key]), (abs(got - complex(exp[key])) / abs(exp[key]), tol)


M10_OPEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "trees", "sycamore_m10_open8.json")
M10_OPEN_ARRAYS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sycamore_m10_open8_arrays.npz")
M10_OPEN_EXPECTED = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sycamore_m10_open8_expected.npz")


def m10_open():
    tree = ca.tree_from_record(ca.load_network(M10_OPEN))
    z = np.load(M10_OPEN_ARRAYS)
    return tree, [z[f"t{i}"] for i in range(tree.N)], np.load(M10_OPEN_EXPECTED)


def test_m10_open_fixture_oracle_slice():
    """Sycamore m10 with 8 open output qubits (tests/golden/gen/make_m10_open.py): a slice of the
    256-amplitude batch by the oracle == the reference's; amplitude 0...0 of the batch == the
    committed single m10 amplitude (another network, another tree)."""
    tree, arrays, exp = m10_open()
    assert len(tree.output) == 8 and tree.nslices == 8
    got = np.asarray(orc.contract_slice(tree, arrays, 1))
    assert got.shape == (2,) * 8
    assert np.allclose(got, exp["slice1"], rtol=1e-11, atol=1e-14)
    single = np.load(M10_EXPECTED)["amplitude"]
    assert abs(exp["amplitudes"].reshape(-1)[0] - single) <= 1e-10 * abs(single)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["complex128", "complex64"])
def test_m10_batched_amplitudes_on_gpu(dtype):
    """All 256 amplitudes in one contraction on the device vs the reference's."""
    tree, arrays, exp = m10_open()
    xs = [a.astype(dtype) for a in arrays]
    got = np.asarray(tree.contract(xs))
    ref = exp["amplitudes"]
    assert got.shape == ref.shape
    scale = np.abs(ref).max()
    tol = 1e-10
    if dtype == "complex64":
        tol = max(1e-5, 8.0 * np.abs(np.asarray(orc.contract(tree, xs)) - ref).max() / scale)
    assert np.abs(got - ref).max() <= tol * scale
    s1 = np.asarray(tree.contract_slice(xs, 1))
    assert np.abs(s1 - exp["slice1"]).max() <= max(tol, 1e-10) * np.abs(exp["slice1"]).max() * 4
