"""RoI grid pooling on CPU: the oracle against hand-computable cases and a literal per-point loop, and the
host-side mirror (virconv_amd.voxel_pool) on the oracle backend against the fixture produced by the reference's own Python
(tests/golden/make_golden_pool.py)."""
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, fill_parameters
from oracle import pooling_ref
from oracle.backend import OracleBackend
from virconv_amd import ops, spconv, synth, voxel_pool

VOXEL_SIZE = [0.05, 0.05, 0.05]
PC_RANGE = [0.0, -40.0, -3.0, 70.4, 40.0, 1.0]
CFG = dict(query_ranges=[[2, 2, 2], [4, 4, 4]], radii=[0.4, 0.8], nsamples=[16, 16], mlps=[[32, 32, 32], [32, 32, 32]],
           pool_method="max_pool")


def literal_voxel_query(max_range, radius, nsample, xyz, new_xyz, new_coords, vol):
    """voxel_query_gpu.cu:10-90 transcribed as plain loops (one query at a time), then voxel_query_utils.py:39-44."""
    zr, yr, xr = max_range
    _, R1, R2, R3 = vol.shape
    M = new_coords.shape[0]
    idx = np.zeros((M, nsample), np.int32)
    r2 = np.float32(radius) * np.float32(radius)
    for m in range(M):
        b, cz, cy, cx = (int(v) for v in new_coords[m])
        cnt = 0
        for dz in range(-zr, zr + 1):
            z = cz + dz
            if z < 0 or z >= R1:
                continue
            for dy in range(-yr, yr + 1):
                y = cy + dy
                if y < 0 or y >= R2:
                    continue
                for dx in range(-xr, xr + 1):
                    x = cx + dx
                    if x < 0 or x >= R3:
                        continue
                    nb = vol[b, z, y, x]
                    if nb < 0:
                        continue
                    d = xyz[nb] - new_xyz[m]
                    d2 = np.float32(np.float32(d[0] * d[0]) + np.float32(d[1] * d[1])) + np.float32(d[2] * d[2])
                    if d2 > r2:
                        continue
                    if cnt < nsample:
                        if cnt == 0:
                            idx[m, :] = nb
                        idx[m, cnt] = nb
                        cnt += 1
        if cnt == 0:
            idx[m, 0] = -1
    empty = idx[:, 0] == -1
    idx[empty] = 0
    return idx, empty


def _scene(seed, n, shape=(11, 40, 36), bs=2, stride=4):
    idx = synth.small_scene_indices(seed, n, shape, bs)
    xyz = pooling_ref.voxel_centers(idx[:, 1:4], stride, VOXEL_SIZE, PC_RANGE)
    return idx, xyz, shape, bs


def _queries(rng, idx, xyz, m, shape, stride=4, spread=0.5):
    pick = rng.integers(0, idx.shape[0], m)
    q = (xyz[pick] + rng.uniform(-spread, spread, (m, 3))).astype(np.float32)
    vs = np.asarray(VOXEL_SIZE, np.float32) * stride
    c = np.floor((q - np.asarray(PC_RANGE[:3], np.float32)) / vs).astype(np.int32)           # [x, y, z]
    coords = np.concatenate([idx[pick][:, :1], c[:, [2, 1, 0]]], 1).astype(np.int32)        # [b, z, y, x]
    return q, coords


def test_voxel2pinds_and_centres_known_answers():
    idx = np.array([[0, 1, 2, 3], [1, 0, 0, 0], [0, 1, 2, 3]], np.int32)                      # duplicate coordinate
    vol = pooling_ref.voxel2pinds(idx, 2, (2, 3, 4))
    assert vol.shape == (2, 2, 3, 4) and vol[0, 1, 2, 3] == 2 and vol[1, 0, 0, 0] == 1 and (vol >= 0).sum() == 2
    c = pooling_ref.voxel_centers(np.array([[0, 0, 0], [1, 2, 3]]), 4, VOXEL_SIZE, PC_RANGE)
    np.testing.assert_allclose(c, [[0.1, -39.9, -2.9], [0.7, -39.5, -2.7]], rtol=0, atol=1e-5)


def test_voxel_query_known_answers():
    shape = (5, 5, 5)
    idx = np.array([[0, 2, 2, 1], [0, 2, 2, 2], [0, 2, 2, 3], [0, 4, 4, 4]], np.int32)       # three in a row + a far one
    xyz = pooling_ref.voxel_centers(idx[:, 1:4], 1, [1, 1, 1], [0, 0, 0])
    vol = pooling_ref.voxel2pinds(idx, 1, shape)
    q = np.array([[2.5, 2.5, 2.5]], np.float32)
    c = np.array([[0, 2, 2, 2]], np.int32)
    out, empty = pooling_ref.voxel_query([1, 1, 1], 1.01, 4, xyz, q, c, vol)
    assert not empty[0] and out.tolist() == [[0, 1, 2, 0]]          # scan order x ascending, unused slot = first hit
    out, empty = pooling_ref.voxel_query([1, 1, 1], 0.5, 4, xyz, q, c, vol)
    assert out.tolist() == [[1, 1, 1, 1]]                           # radius keeps only the centre voxel
    out, empty = pooling_ref.voxel_query([1, 1, 1], 1.01, 2, xyz, q, c, vol)
    assert out.tolist() == [[0, 1]]                                 # nsample cap keeps the first two in scan order
    out, empty = pooling_ref.voxel_query([0, 0, 0], 9.0, 3, xyz, np.array([[0.5, 0.5, 0.5]], np.float32),
                                         np.array([[0, 0, 0, 0]], np.int32), vol)
    assert empty[0] and out.tolist() == [[0, 0, 0]]                 # empty ball: mask + zero row
    out, empty = pooling_ref.voxel_query([2, 2, 2], 9.0, 3, xyz, np.array([[60.0, 0.5, 0.5]], np.float32),
                                         np.array([[0, 0, 0, 60]], np.int32), vol)
    assert empty[0]                                                 # query voxel outside the grid
    out, empty = pooling_ref.voxel_query([4, 4, 4], 9.0, 8, xyz, q, c, vol)
    assert out.tolist() == [[0, 1, 2, 3, 0, 0, 0, 0]]               # range clipped by the grid, z-major order


@pytest.mark.parametrize("rng_zyx,radius,nsample", [([2, 2, 2], 0.4, 16), ([4, 4, 4], 0.8, 16), ([1, 2, 3], 0.5, 3)])
def test_vectorised_oracle_equals_literal_loops(rng_zyx, radius, nsample):
    rng = np.random.default_rng(3)
    idx, xyz, shape, bs = _scene(4, 900)
    vol = pooling_ref.voxel2pinds(idx, bs, shape)
    q, coords = _queries(rng, idx, xyz, 120, shape)
    coords[:5, 3] += 40                                             # some queries outside the grid
    a, ea = pooling_ref.voxel_query(rng_zyx, radius, nsample, xyz, q, coords, vol)
    b, eb = literal_voxel_query(rng_zyx, radius, nsample, xyz, q, coords, vol)
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(ea, eb)
    assert 0 < ea.sum() < ea.size


def test_group_points_and_grad_match_torch_indexing():
    rng = np.random.default_rng(0)
    fbc, ibc = np.array([50, 70], np.int32), np.array([30, 20], np.int32)
    f = rng.standard_normal((120, 8)).astype(np.float32)
    idx = np.concatenate([rng.integers(0, 50, (30, 6)), rng.integers(0, 70, (20, 6))]).astype(np.int32)
    out = pooling_ref.group_points(f, fbc, idx, ibc)
    glob = idx.astype(np.int64) + np.repeat([0, 50], [30, 20])[:, None]
    ft = torch.from_numpy(f).requires_grad_(True)
    ref = ft[torch.from_numpy(glob)].permute(0, 2, 1)
    np.testing.assert_array_equal(out, ref.detach().numpy())
    g = rng.standard_normal(out.shape).astype(np.float32)
    ref.backward(torch.from_numpy(g))
    np.testing.assert_allclose(pooling_ref.group_points_grad(g, idx, ibc, fbc, 120), ft.grad.numpy(), rtol=1e-5, atol=1e-5)


def _fixture():
    return np.load(os.path.join(GOLDEN, "voxel_pool_ref.npz"))


def run_module_against_fixture(device, backend, tol):
    """Shared by the CPU (oracle backend) and GPU (HIP backend) tests."""
    g = _fixture()
    shape, bs, stride = (21, 100, 88), g["new_xyz"].shape[0], 4
    dev = torch.device(device)
    with ops.use_backend(backend):
        sp = spconv.SparseConvTensor(torch.from_numpy(g["features"]).to(dev), torch.from_numpy(g["indices"]).to(dev),
                                     list(shape), bs)
        xyz = voxel_pool.get_voxel_centers(sp.indices[:, 1:4], stride, VOXEL_SIZE, PC_RANGE)
        np.testing.assert_array_equal(xyz.cpu().numpy(), g["xyz"])
        cnt = torch.tensor([(g["indices"][:, 0] == b).sum() for b in range(bs)], dtype=torch.int32, device=dev)
        v2p = voxel_pool.generate_voxel2pinds(sp)
        new_xyz = torch.from_numpy(g["new_xyz"]).view(-1, 3).contiguous().to(dev)
        coords = torch.from_numpy(g["new_coords_bxyz"]).to(dev)
        new_cnt = torch.full((bs,), g["new_xyz"].shape[1], dtype=torch.int32, device=dev)
        grp = voxel_pool.VoxelQueryAndGrouping([2, 2, 2], 0.4, 16)
        gf, gx, empty = grp(coords[:, [0, 3, 2, 1]].contiguous(), xyz.contiguous(), cnt, new_xyz, new_cnt, sp.features, v2p)
        np.testing.assert_array_equal(empty.cpu().numpy(), g["q0_empty"])
        np.testing.assert_array_equal(gx.cpu().numpy(), g["q0_grouped_xyz"])      # same rows in the same order
        np.testing.assert_allclose(gf.cpu().numpy().sum(axis=1), g["q0_grouped_features_sum"], rtol=1e-5, atol=1e-5)
        for mode in ("train", "eval"):
            mod = voxel_pool.NeighborVoxelSAModuleMSG(**CFG)
            fill_parameters(mod, 7)
            mod = mod.to(dev).train(mode == "train")
            f = torch.from_numpy(g["features"]).to(dev).requires_grad_(True)
            y = mod(xyz=xyz.contiguous(), xyz_batch_cnt=cnt, new_xyz=new_xyz, new_xyz_batch_cnt=new_cnt, new_coords=coords,
                    features=f, voxel2point_indices=v2p)
            (y * torch.from_numpy(g["out_grad"]).to(dev)).sum().backward()

            def close(a, b, what):
                err = np.abs(a - b).max() / max(1.0, np.abs(b).max())
                assert err < tol, (mode, what, err)
            close(y.detach().cpu().numpy(), g[f"{mode}_out"], "out")
            close(f.grad.cpu().numpy(), g[f"{mode}_grad_features"], "grad_features")
            close(mod.mlps_in[0][0].weight.grad.cpu().numpy(), g[f"{mode}_grad_w_in0"], "grad_w_in0")
            close(mod.mlps_pos[1][0].weight.grad.cpu().numpy(), g[f"{mode}_grad_w_pos1"], "grad_w_pos1")


def test_host_mirror_reproduces_the_reference_module_on_the_oracle_backend():
    run_module_against_fixture("cpu", OracleBackend(), 1e-5)


def test_state_dict_keys_match_the_reference_module_layout():
    mod = voxel_pool.NeighborVoxelSAModuleMSG(**CFG)
    keys = set(mod.state_dict())
    for k in range(2):
        assert {f"mlps_in.{k}.0.weight", f"mlps_in.{k}.1.running_mean", f"mlps_pos.{k}.0.weight", f"mlps_pos.{k}.1.weight",
                f"mlps_out.{k}.0.weight", f"mlps_out.{k}.1.bias"} <= keys
    assert mod.mlps_pos[0][0].weight.shape == (32, 3, 1, 1)
