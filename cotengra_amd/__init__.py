"""cotengra_amd -- MI355X-native sliced contraction-tree executor.

Drop-in execution backend for the contraction path of jcmgray/cotengra
(``ContractionTree.contract`` / ``contract_slice`` / ``einsum_expression``):
Python host code compiles a tree into a static plan and drives hand-written
HIP (gfx950) kernels through the C ABI declared in ``include/ctg_hip.h``.
"""

from .tree import ContractionTree, SliceInfo, get_slice_strides  # noqa: F401
from .utils import (  # noqa: F401
    eq_to_inputs_output,
    inputs_output_to_eq,
    lattice_equation,
    load_network,
    make_arrays_from_inputs,
    tree_from_record,
)

__version__ = "0.1.0"
from . import interface, plan  # noqa: F401,E402
from .contractor import HipContractor, make_contractor  # noqa: F401,E402
from .interface import (  # noqa: F401,E402
    array_contract,
    array_contract_expression,
    array_contract_tree,
    einsum,
    einsum_expression,
    einsum_tree,
    greedy_path,
    tensordot,
)
