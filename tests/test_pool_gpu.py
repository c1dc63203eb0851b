"""RoI grid pooling operators on the GPU, through the C ABI: HIP vs the CPU oracle (indices bit-exact),
the host mirror vs the fixture made by the reference's own module, and KITTI-scale properties."""
import numpy as np
import pytest
import torch

from oracle import pooling_ref
from test_pool_cpu import PC_RANGE, VOXEL_SIZE, _queries, _scene, run_module_against_fixture
from virconv_amd import synth

pytestmark = pytest.mark.gpu


def _cuda(*arrs):
    return [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in arrs]


@pytest.mark.parametrize("rng_zyx,radius,nsample", [([2, 2, 2], 0.4, 16), ([4, 4, 4], 0.8, 16), ([1, 2, 3], 0.5, 3),
                                                   ([0, 0, 0], 0.2, 1), ([3, 3, 31], 3.0, 70), ([5, 6, 2], 1.0, 32)])
def test_voxel_query_bit_exact(hip_backend, rng_zyx, radius, nsample):
    rng = np.random.default_rng(11)
    idx, xyz, shape, bs = _scene(7, 4000, shape=(11, 80, 72))
    q, coords = _queries(rng, idx, xyz, 3000, shape, spread=0.7)
    coords[:40, 3] += 100      # outside the grid
    coords[40:60, 1] -= 20
    coords[60:70, 0] = 5       # invalid batch index: treated as an empty ball, never read
    vol = pooling_ref.voxel2pinds(idx, bs, shape)
    ok = (coords[:, 0] >= 0) & (coords[:, 0] < bs)
    ref_idx, ref_empty = np.zeros((coords.shape[0], nsample), np.int32), np.ones(coords.shape[0], bool)
    ref_idx[ok], ref_empty[ok] = pooling_ref.voxel_query(rng_zyx, radius, nsample, xyz, q[ok], coords[ok], vol)
    it, xt, qt, ct = _cuda(idx, xyz, q, coords)
    ws = hip_backend.voxel_index_build(it, bs, shape)
    out, empty = hip_backend.voxel_query(ws, idx.shape[0], bs, shape, xt, qt, ct, rng_zyx, radius, nsample)
    np.testing.assert_array_equal(empty.cpu().numpy(), ref_empty)
    np.testing.assert_array_equal(out.cpu().numpy(), ref_idx)
    assert 0 < ref_empty.sum() < ref_empty.size


def test_voxel_query_empty_inputs_and_argument_errors(hip_backend):
    idx, xyz, shape, bs = _scene(1, 100)
    it, xt = _cuda(idx, xyz)
    ws = hip_backend.voxel_index_build(it, bs, shape)
    out, empty = hip_backend.voxel_query(ws, idx.shape[0], bs, shape, xt, torch.zeros((0, 3)).cuda(),
                                         torch.zeros((0, 4), dtype=torch.int32).cuda(), [1, 1, 1], 1.0, 4)
    assert out.shape == (0, 4) and empty.shape == (0,)
    ws0 = hip_backend.voxel_index_build(torch.zeros((0, 4), dtype=torch.int32).cuda(), bs, shape)   # empty tensor
    q, c = _cuda(np.zeros((5, 3), np.float32), np.zeros((5, 4), np.int32))
    out, empty = hip_backend.voxel_query(ws0, 0, bs, shape, torch.zeros((0, 3)).cuda(), q, c, [1, 1, 1], 1.0, 4)
    assert bool(empty.all()) and int(out.abs().sum()) == 0
    with pytest.raises(RuntimeError):
        hip_backend.voxel_query(ws, idx.shape[0], bs, shape, xt, q, c, [1, 1, 32], 1.0, 4)           # x_range > 31


@pytest.mark.parametrize("c,nsample", [(3, 16), (32, 16), (64, 7), (100, 5)])
def test_group_points_exact_and_grad(hip_backend, c, nsample):
    rng = np.random.default_rng(c)
    fbc, ibc = np.array([500, 0, 700], np.int32), np.array([300, 10, 200], np.int32)
    ibc[1] = 0                                                          # a sample without queries and without voxels
    f = rng.standard_normal((1200, c)).astype(np.float32)
    idx = np.concatenate([rng.integers(0, 500, (300, nsample)), rng.integers(0, 700, (200, nsample))]).astype(np.int32)
    ft, it, fb, ib = _cuda(f, idx, fbc, ibc)
    out = hip_backend.group_points(ft, fb, it, ib)
    np.testing.assert_array_equal(out.cpu().numpy(), pooling_ref.group_points(f, fbc, idx, ibc))
    g = rng.standard_normal((500, c, nsample)).astype(np.float32)
    gf = hip_backend.group_points_grad(torch.from_numpy(g).cuda(), it, ib, fb, 1200).cpu().numpy()
    ref = pooling_ref.group_points_grad(g, idx, ibc, fbc, 1200)
    assert np.abs(gf - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max())


def test_host_mirror_reproduces_the_reference_module_on_hip(hip_backend):
    run_module_against_fixture("cuda", hip_backend, 1e-4)


def test_kitti_scale_query_properties(hip_backend):
    """x_conv3-scale tensor (stride 4, ~50 k voxels per frame, bs 2) and 2 x 128 x 216 grid points (ROI_GRID_POOL)."""
    rng = np.random.default_rng(0)
    shape, bs, stride = (21, 400, 352), 2, 4
    rows = []
    for b in range(bs):
        fr = synth.make_frame(b)
        p = np.concatenate([fr["points_lidar"], fr["points_virtual"]])[:, :3]
        c = np.floor((p - np.asarray(PC_RANGE[:3], np.float32)) / (np.asarray(VOXEL_SIZE, np.float32) * stride)).astype(np.int64)
        c = c[(c >= 0).all(1) & (c[:, 0] < shape[2]) & (c[:, 1] < shape[1]) & (c[:, 2] < shape[0])]
        lin = np.unique((c[:, 2] * shape[1] + c[:, 1]) * shape[2] + c[:, 0])
        z, r = np.divmod(lin, shape[1] * shape[2])
        y, x = np.divmod(r, shape[2])
        rows.append(np.stack([np.full_like(z, b), z, y, x], 1))
    idx = np.concatenate(rows).astype(np.int32)
    xyz = pooling_ref.voxel_centers(idx[:, 1:4], stride, VOXEL_SIZE, PC_RANGE)
    m_per = 128 * 216
    qs, cs = [], []
    for b in range(bs):
        sel = np.nonzero(idx[:, 0] == b)[0]
        q, c = _queries(rng, idx[sel], xyz[sel], m_per, shape, spread=1.5)
        qs.append(q)
        cs.append(c)
    q, coords = np.concatenate(qs), np.concatenate(cs)
    it, xt, qt, ct = _cuda(idx, xyz, q, coords)
    ws = hip_backend.voxel_index_build(it, bs, shape)
    for rng_zyx, radius in (([2, 2, 2], 0.4), ([4, 4, 4], 0.8)):
        out, empty = hip_backend.voxel_query(ws, idx.shape[0], bs, shape, xt, qt, ct, rng_zyx, radius, 16)
        out2, empty2 = hip_backend.voxel_query(ws, idx.shape[0], bs, shape, xt, qt, ct, rng_zyx, radius, 16)
        assert torch.equal(out, out2) and torch.equal(empty, empty2)              # deterministic
        ref_idx, ref_empty = pooling_ref.voxel_query(rng_zyx, radius, 16, xyz, q, coords, pooling_ref.voxel2pinds(idx, bs, shape))
        np.testing.assert_array_equal(empty.cpu().numpy(), ref_empty)
        np.testing.assert_array_equal(out.cpu().numpy(), ref_idx)
        o = out.cpu().numpy()[~ref_empty]
        d = np.linalg.norm(xyz[o] - q[~ref_empty][:, None, :], axis=2)
        assert d.max() <= radius * (1 + 1e-6)                                     # every returned voxel is inside the ball
        assert (idx[o][:, :, 0] == coords[~ref_empty][:, None, 0]).all()          # and in the query's sample
