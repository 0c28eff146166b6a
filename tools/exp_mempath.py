#!/usr/bin/env python
"""One streaming-kernel launch (R x 32 x 32, contiguous rows) and one torch
elementwise 1R:1W launch of the same byte volume, for PMC comparison under
rocprofv3 (tools/exp_mempath.sh)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import cotengra_amd as ca  # noqa: E402
from cotengra_amd.contractor import HipContractor  # noqa: E402

R = 1 << 26
sizes = dict(a=R, k=32, b=32)
tree = ca.ContractionTree.from_path([("a", "k"), ("k", "b")], ("a", "b"), sizes, path=[(0, 1)])
A = torch.view_as_complex(torch.randn([R, 32, 2], device="cuda"))
B = torch.view_as_complex(torch.randn([32, 32, 2], device="cuda"))
fn = HipContractor(tree)
st = fn.setup(A, B)
for _ in range(2):
    st["exec"].run_slices(0, 1, 1)
st["exec"].sync()
x = torch.view_as_real(A).reshape(-1)
y = torch.empty_like(x)
for _ in range(2):
    torch.mul(x, 2.0, out=y)
torch.cuda.synchronize()
fn.close()
