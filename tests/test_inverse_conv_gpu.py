"""SparseInverseConv3d / SparseInverseConv2d through the spconv facade on the HIP backend (VERDICT r3 "missing" #3).

Reference: `post_act_block(..., conv_type='inverseconv')` / `post_act_block2d` build `spconv.SparseInverseConv3d/2d(in, out, k,
indice_key=..., bias=False)` (pcdet/models/backbones_3d/spconv_backbone.py:95-97,119-121); unused by VirConv-L/T/S but part of
the operator API of SURVEY 8b.  Semantics (spconv): the inverse conv runs on the OUTPUT tensor of the strided conv that owns
`indice_key` and produces a tensor at that conv's INPUT coordinates, through the transposed pair table:
    out[i, :] = sum over (o, kappa) with  p_i = q_o * stride - pad + kappa * dil  of  y[o, :] @ W[:, kappa, :]^T,
i.e. a dense `conv_transpose` read back at the original active sites.  Checked three ways: (1) against an independent dense
float64 oracle (torch conv_transpose{2,3}d on the densified tensor), (2) against the sparse oracle backend running the same
facade modules on the CPU, forward AND backward (dX, dW), (3) bit-stability run to run.  Tolerance: fp32 features within
1e-4 * max(1, max|ref|) element-wise (north_star), indices bit-exact."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import dense_ref
from oracle.backend import OracleBackend
from virconv_amd import ops, spconv, synth

pytestmark = pytest.mark.gpu


def _unique2(seed, n, bs, shape):
    rng = np.random.default_rng(seed)
    idx = np.stack([rng.integers(0, bs, n), rng.integers(0, shape[0], n), rng.integers(0, shape[1], n)], 1).astype(np.int32)
    idx = np.unique(idx, axis=0)
    return idx[rng.permutation(idx.shape[0])]


@pytest.mark.parametrize("ndim,shape,cin,cmid,cout", [(3, (21, 64, 48), 16, 32, 16), (3, (20, 33, 47), 8, 16, 8),
                                                     (2, (160, 60), 16, 16, 32)])
def test_inverse_conv_facade_forward_backward(hip_backend, ndim, shape, cin, cmid, cout):
    bs = 2
    rng = np.random.default_rng(cin + cout + ndim)
    idx = synth.small_scene_indices(5, 6001, shape, bs) if ndim == 3 else _unique2(5, 4000, bs, shape)
    n = idx.shape[0]
    feats = rng.standard_normal((n, cin)).astype(np.float32)
    Down = spconv.SparseConv3d if ndim == 3 else spconv.SparseConv2d
    Up = spconv.SparseInverseConv3d if ndim == 3 else spconv.SparseInverseConv2d
    torch.manual_seed(3)
    down = Down(cin, cmid, 3, stride=2, padding=1, bias=False, indice_key="sp")
    up = Up(cmid, cout, 3, indice_key="sp", bias=False)
    g_out = rng.standard_normal((n, cout)).astype(np.float32)

    def run(device, backend, dtype):
        with ops.use_backend(backend):
            d, u = down.to(device=device, dtype=dtype), up.to(device=device, dtype=dtype)
            for m in (d, u):
                m.zero_grad(set_to_none=True)
            f = torch.from_numpy(feats).to(device=device, dtype=dtype).requires_grad_(True)
            x = spconv.SparseConvTensor(f, torch.from_numpy(idx).to(device), list(shape), bs)
            y = d(x)
            z = u(y)
            (z.features * torch.from_numpy(g_out).to(device=device, dtype=dtype)).sum().backward()
            res = (z.features.detach().cpu().double().numpy(), z.indices.cpu().numpy(), y.features.detach().cpu().double().numpy(),
                   y.indices.cpu().numpy(), f.grad.cpu().double().numpy(), d.weight.grad.cpu().double().numpy(),
                   u.weight.grad.cpu().double().numpy())
        return res

    hip = run("cuda", hip_backend, torch.float32)
    hip2 = run("cuda", hip_backend, torch.float32)
    for a, b in zip(hip, hip2):
        assert np.array_equal(a, b), "not bit-stable run to run"
    ref = run("cpu", OracleBackend(), torch.float64)
    down.float(), up.float()
    # indices: the inverse conv lands on the strided conv's INPUT coordinates, in their order
    assert np.array_equal(hip[1], idx) and np.array_equal(hip[3], ref[3])
    names = ("inverse-conv output", None, "strided-conv output", None, "dX", "dW(down)", "dW(up)")
    for a, b, name in zip(hip, ref, names):
        if name is None:
            continue
        tol = 1e-4 * max(1.0, float(np.abs(b).max()))
        assert float(np.abs(a - b).max()) <= tol, (name, float(np.abs(a - b).max()), tol)
    # independent dense oracle of the inverse conv: conv_transpose of the densified strided-conv output, read at the input sites
    yd = dense_ref.densify(torch.from_numpy(ref[2]), ref[3], [(s + 2 - 3) // 2 + 1 for s in shape], bs)
    w = up.weight.detach().double()                                   # (Cout, *k, Cin) -> conv_transpose wants (Cin, Cout, *k)
    wt = w.permute(ndim + 1, 0, *range(1, ndim + 1)).contiguous()
    opad = [s - ((((s + 2 - 3) // 2 + 1) - 1) * 2 - 2 + 3) for s in shape]
    dense = (F.conv_transpose3d if ndim == 3 else F.conv_transpose2d)(yd, wt, stride=2, padding=1, output_padding=opad)
    ii = torch.from_numpy(idx.astype(np.int64))
    picked = dense.movedim(1, -1)[tuple(ii[:, a] for a in range(ndim + 1))]
    tol = 1e-4 * max(1.0, float(picked.abs().max()))
    assert float((torch.from_numpy(hip[0]) - picked).abs().max()) <= tol
