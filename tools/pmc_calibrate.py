"""Known-byte-count launches for calibrating rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 in OUR access patterns
(MI355X_MICROARCH.md §HBM: FETCH_SIZE under-reports wide coalesced reads by 2x; other patterns are uncalibrated).

  bn_apply_kernel  : float4 streaming read of N*C*4 bytes + float4 streaming write of N*C*4 bytes (N = 2^21, C = 64 -> 512 MiB each,
                     larger than the 256 MiB Infinity Cache)
  gather_rows_kernel with a random permutation: row-granular (256 B) gather of the same volume + streaming write
Run under:  rocprofv3 --pmc FETCH_SIZE --kernel-trace ... / --pmc WRITE_SIZE ...
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from virconv_amd import ops  # noqa: E402

be = ops.get_backend()
n, c = 1 << 21, 64
x = torch.randn((n, c), device="cuda")
mean, var = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
g, b = torch.ones(c, device="cuda"), torch.zeros(c, device="cuda")
perm = torch.randperm(n, device="cuda")
for _ in range(3):
    y, _, _ = be.bn_forward(x, g, b, mean, var, False, 0.0, 1e-3, True)
    f, _ = be.gather_rows(x, None, perm)
torch.cuda.synchronize()
print("bytes each way:", n * c * 4)
