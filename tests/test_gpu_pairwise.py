"""Kernel-level GPU parity: single pairwise contractions with shuffled index
layouts against numpy.einsum (complex128 accumulate), sized to hit every MFMA
tile configuration, split-K, the k-reduction kernel, batch (hyper) indices,
ragged tile edges and non-power-of-two extents."""
import numpy as np
import pytest

import cotengra_amd as ca
import golden_util as G
from cotengra_amd.interface import einsum

pytestmark = pytest.mark.gpu

CASES = [
    # eq, sizes
    ("abcd,cdef->abef", dict(a=16, b=32, c=8, d=4, e=8, f=8)),        # N=64 config
    ("abcd,cdef->abef", dict(a=64, b=32, c=4, d=4, e=4, f=4)),        # N=16 config
    ("abcd,cdef->feba", dict(a=64, b=32, c=4, d=4, e=8, f=4)),        # N=32, scattered output
    ("dacb,fdce->abef", dict(a=16, b=32, c=8, d=4, e=8, f=8)),        # permuted operands
    ("abk,kc->abc", dict(a=8, b=4, k=65536, c=32)),                   # split-K (32 x 65536 x 32)
    ("ak,kb->ab", dict(a=2, k=1 << 18, b=2)),                         # k-reduction kernel
    ("k,k->", dict(k=1 << 20)),                                       # dot product
    ("xab,xbc->xac", dict(x=5, a=96, b=24, c=40)),                    # batch + ragged
    ("axb,bxc->xca", dict(x=3, a=130, b=17, c=33)),                   # ragged everything
    ("abc,cd->abd", dict(a=81, b=27, c=9, d=27)),                     # powers of three
    ("ab,cd->abcd", dict(a=64, b=64, c=8, d=8)),                      # outer product
    ("ab,ab->ab", dict(a=512, b=300)),                                # Hadamard
    ("abc,bcd->ad", dict(a=4096, b=32, c=32, d=256)),                 # K=1024 GEMM
    ("abcdefgh,hgfeij->abcdij", dict(a=8, b=8, c=8, d=8, e=2, f=2, g=2, h=2, i=4, j=4)),
    # tall-skinny streaming kernel (R >= 8192, K <= 128, N <= 64)
    ("abck,kn->abcn", dict(a=32, b=32, c=16, k=16, n=16)),            # 15: contiguous k
    ("kabc,nk->cban", dict(a=32, b=32, c=16, k=32, n=32)),            # 16: k slowest, scattered out
    ("akbc,kn->abcn", dict(a=64, b=16, c=16, k=128, n=64)),           # 17: K=128, N=64 (large LDS)
    ("abkc,kn->abcn", dict(a=40, b=25, c=10, k=12, n=5)),             # 18: ragged R/K/N (general path)
    ("abcdefghijklmnop,dhlp->abcefgijkmno", {ix: 2 for ix in "abcdefghijklmnop"}),  # 19: bit-permuted
    ("abcdefghijklmnop,pdxhyl->xabcefygijkmno", {ix: 2 for ix in "abcdefghijklmnopxy"}),  # 20
    # short contractions in the streaming kernel (only the real K columns are gathered)
    ("abck,kn->abcn", dict(a=32, b=32, c=16, k=8, n=4)),              # 21: K=8
    ("akbc,kn->abcn", dict(a=32, b=32, c=16, k=4, n=64)),             # 22: K=4, N=64
    ("abcdefghijklmnop,dhpxy->abcefgijklmnoxy", {ix: 2 for ix in "abcdefghijklmnopxy"}),  # 23: K=8 bits
    ("abkc,kn->abcn", dict(a=32, b=32, c=16, k=12, n=16)),            # 24: K=12
    ("abck,kn->abcn", dict(a=32, b=32, c=16, k=2, n=8)),              # 25: K=2 streaming (pairs along k)
    ("akbc,kn->abcn", dict(a=32, b=3, c=16, k=3, n=8)),               # 26: K=3 streaming, ragged rows
    ("kabc,kn->abcn", dict(a=32, b=32, c=16, k=2, n=2)),              # 27: K=2, k slowest, N=2
    ("ak,kb->ab", dict(a=20, k=1 << 17, b=9)),                        # 28: k-streaming, ragged R / N
    ("ka,bk->ab", dict(a=32, k=1 << 16, b=16)),                       # 29: k-streaming, k slowest in A
    ("aklm,mlkb->ba", dict(a=8, k=64, l=64, m=32, b=32)),             # 30: k-streaming, 3 contracted indices
    # row-wise FMA kernel: tall steps, a handful of multiply-adds per row, odd extents / batch index
    ("abkc,kn->abcn", dict(a=27, b=32, c=27, k=4, n=4)),              # rows 3^6 * 2^5, 4 columns
    ("akbc,knm->abcnm", dict(a=16, b=8, c=81, k=6, n=3, m=3)),         # K = 6, N = 9 (12-column variant)
    ("abkc,knm->abcnm", dict(a=32, b=8, c=36, k=12, n=6, m=3)),       # K = 12, N = 18 (24-column variant)
    ("xabk,xkn->xabn", dict(x=3, a=128, b=64, k=8, n=3)),             # batch index, 8192 rows per entry
    ("abck,kn->abcn", dict(a=32, b=32, c=16, k=8, n=8)),              # powers of two, K, N <= 8
    ("abkc,kn->nabc", dict(a=27, b=32, c=27, k=4, n=4)),              # columns slowest in the output
    ("abkc,knm->abcnm", dict(a=27, b=32, c=12, k=9, n=5, m=6)),       # K = 9, N = 30 (32-column variant)
    # 32 columns that are the fastest index of the result: the LDS-staged store path at its
    # largest (66 KB of dynamic LDS: the kernel opts in); odd row count so that it IS this kernel
    ("abkc,knm->abcnm", dict(a=27, b=32, c=12, k=6, n=4, m=8)),       # K = 6, N = 32
    ("xabk,xkn->xabn", dict(x=3, a=96, b=96, k=8, n=32)),             # the same with a batch index
    # two to four outputs under a long contraction: one wave per k-chunk computes all of them
    # (pair_kred_multi_kernel, round 4) -- shared and per-output operands, odd K, k slowest
    ("ak,k->a", dict(a=3, k=99999)),                                  # the second operand is shared
    ("xk,xk->x", dict(x=3, k=70001)),                                 # both operands depend on the output
    ("ka,kb->ab", dict(a=2, k=65537, b=2)),                           # k slowest in both, 2 x 2 outputs
    ("akl,lk->a", dict(a=2, k=300, l=301)),                           # two contracted indices, transposed
]


@pytest.mark.parametrize("dtype", ["complex64", "complex128", "float32", "float64"])
@pytest.mark.parametrize("case", range(len(CASES)))
def test_pairwise(case, dtype):
    eq, sizes = CASES[case]
    if dtype in ("complex128", "float64", "float32") and case in (4, 12, 17):
        pytest.skip("large case only exercised on the complex64 MFMA path")
    (ta, tb), out = ca.eq_to_inputs_output(eq)
    rng = np.random.default_rng(case)
    def mk(t):
        shape = [sizes[i] for i in t]
        x = rng.normal(size=shape)
        if "complex" in dtype:
            x = x + 1j * rng.normal(size=shape)
        return x.astype(dtype)
    a, b = mk(ta), mk(tb)
    hi = "complex128" if "complex" in dtype else "float64"
    ref = np.einsum(eq, a.astype(hi), b.astype(hi), optimize=True)
    got = einsum(eq, a, b, optimize=[(0, 1)])
    assert np.shape(got) == np.shape(ref)
    scale = np.abs(ref).max()
    tol = 1e-12
    if dtype in ("complex64", "float32"):
        # numpy's own single-precision einsum of the same operands sets the scale of
        # what rounding may cost on this shape (long reductions: ~ sqrt(K) eps)
        tol = G.single_gate(ref, np.einsum(eq, a, b, optimize=True))
    assert np.abs(np.asarray(got) - ref).max() <= tol * scale, (np.abs(np.asarray(got) - ref).max() / scale, tol)


SKINNY = [
    # rows paired along the fastest index of A and C, K a power of two, N in {1, 2, 4}
    ("kabc,kn->abcn", dict(a=64, b=64, c=32, k=8, n=2)),
    ("akbc,nk->abcn", dict(a=64, b=64, c=32, k=4, n=4)),
    ("abkc,k->abc", dict(a=64, b=64, c=32, k=16)),
    ("kabc,kn->abcn", dict(a=128, b=64, c=32, k=2, n=1)),
    ("ajbklc,ljkn->abcn", dict(a=32, b=64, c=64, j=2, k=2, l=2, n=2)),   # three contracted bits
]


@pytest.mark.parametrize("case", range(len(SKINNY)))
def test_skinny_kernel(case):
    """K*N <= 16 with a huge row count: the FMA streaming kernel (no MFMA tile
    to fill), checked against numpy in complex128."""
    from cotengra_amd.contractor import HipContractor

    eq, sizes = SKINNY[case]
    (ta, tb), out = ca.eq_to_inputs_output(eq)
    rng = np.random.default_rng(100 + case)
    arrays = [
        (rng.normal(size=[sizes[i] for i in t]) + 1j * rng.normal(size=[sizes[i] for i in t])).astype("complex64")
        for t in (ta, tb)
    ]
    tree = ca.ContractionTree.from_path([ta, tb], out, sizes, path=[(0, 1)])
    fn = HipContractor(tree)
    st = fn.setup(*arrays)
    names = [n for n in st["exec"].step_kernels() if n.startswith("pair_")]
    assert names and all(n.startswith("pair_skinny_kernel") for n in names), names
    got = np.asarray(fn(*arrays))
    ref = np.einsum(eq, *[x.astype("complex128") for x in arrays], optimize=True)
    assert got.shape == ref.shape
    tol = G.single_gate(ref, np.einsum(eq, *arrays, optimize=True)) * np.abs(ref).max()
    assert np.abs(got - ref).max() <= tol
    # with strip_exponent the step scales by 1 / (facA * facB)
    m, e = fn(*arrays, strip_exponent=True)
    assert np.abs(np.asarray(m) * 10.0**e - ref).max() <= tol
    fn.close()


def test_tensordot_numpy_semantics():
    from cotengra_amd.interface import tensordot

    rng = np.random.default_rng(3)
    a = rng.normal(size=(4, 2, 3)) + 1j * rng.normal(size=(4, 2, 3))
    b = rng.normal(size=(3, 4, 5)) + 1j * rng.normal(size=(3, 4, 5))
    for axes in (((0, 2), (1, 0)), ((2,), (0,)), 0, ((-1,), (0,))):
        got = np.asarray(tensordot(a, b, axes))
        ref = np.tensordot(a, b, axes)
        assert got.shape == ref.shape and np.abs(got - ref).max() < 1e-12 * np.abs(ref).max()
    c = rng.normal(size=(3, 5, 4))
    assert np.allclose(np.asarray(tensordot(a.real, c, 1)), np.tensordot(a.real, c, 1))
    with pytest.raises(ValueError):
        tensordot(a, b, ((0,), (0,)))


def test_strip_exponent_toggle_on_live_executor():
    """The same contractor called without and then with strip_exponent (and
    back): the kernel hints of the MFMA steps must survive the switch."""
    from cotengra_amd.contractor import HipContractor

    eq, sizes = "abcd,cdef->abef", dict(a=16, b=32, c=8, d=4, e=8, f=8)
    (ta, tb), out = ca.eq_to_inputs_output(eq)
    rng = np.random.default_rng(7)
    arrays = [
        (rng.normal(size=[sizes[i] for i in t]) + 1j * rng.normal(size=[sizes[i] for i in t])).astype("complex64")
        for t in (ta, tb)
    ]
    ref = np.einsum(eq, *[x.astype("complex128") for x in arrays], optimize=True)
    fn = HipContractor(ca.ContractionTree.from_path([ta, tb], out, sizes, path=[(0, 1)]))
    tol = G.single_gate(ref, np.einsum(eq, *arrays, optimize=True)) * np.abs(ref).max()
    assert np.abs(np.asarray(fn(*arrays)) - ref).max() <= tol
    m, e = fn(*arrays, strip_exponent=True)
    assert np.abs(np.asarray(m) * 10.0**e - ref).max() <= tol
    assert np.abs(np.asarray(fn(*arrays)) - ref).max() <= tol
    fn.close()


@pytest.mark.parametrize("case", range(len(CASES) - 11, len(CASES) - 4))   # (the last four are k-reductions)
def test_rowwise_kernel_takes_these_steps(case):
    """The row-wise cases above are served by pair_rowwise_kernel (complex64): K, N <= 8,
    or 32-row groups that are not base + constant (an odd extent fastest among the
    rows) with K < 16, or a batch index."""
    import torch

    from cotengra_amd.contractor import HipContractor

    eq, sizes = CASES[case]
    (ta, tb), out = ca.eq_to_inputs_output(eq)
    tree = ca.ContractionTree.from_path([ta, tb], out, sizes, path=[(0, 1)])
    rng = np.random.default_rng(case)
    arrays = [(rng.normal(size=[sizes[i] for i in t]) + 1j * rng.normal(size=[sizes[i] for i in t])).astype("complex64")
              for t in (ta, tb)]
    fn = HipContractor(tree)
    st = fn.setup(*[torch.as_tensor(a, device="cuda") for a in arrays])
    names = st["exec"].step_kernels()
    fn.close()
    assert any(n.startswith("pair_rowwise_kernel") for n in names), names
