"""CPU-only: the 3DGS .ply layout (scope row f2; strategies/base_gaussian_model.py:165-248)."""
import struct

import numpy as np
import torch

from clm_gs_amd import io_ply


def test_ply_layout_and_roundtrip(tmp_path):
    g = torch.Generator().manual_seed(0)
    n = 37
    xyz, shs = torch.randn(n, 3, generator=g), torch.randn(n, 48, generator=g)
    op, sc, rot = torch.randn(n, 1, generator=g), torch.randn(n, 3, generator=g), torch.randn(n, 4, generator=g)
    p = tmp_path / "point_cloud.ply"
    io_ply.save_ply(str(p), xyz, shs, op, sc, rot)
    raw = p.read_bytes()
    head, body = raw.split(b"end_header\n", 1)
    lines = head.decode().splitlines()
    assert lines[0] == "ply" and lines[1] == "format binary_little_endian 1.0" and lines[2] == f"element vertex {n}"
    props = [l.split()[2] for l in lines[3:]]
    # order of construct_list_of_attributes (base_gaussian_model.py:165-187)
    assert props[:6] == ["x", "y", "z", "nx", "ny", "nz"] and props[6:9] == ["f_dc_0", "f_dc_1", "f_dc_2"]
    assert props[9:54] == [f"f_rest_{i}" for i in range(45)]
    assert props[54:] == ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
    assert len(body) == n * 62 * 4
    row5 = struct.unpack("<62f", body[5 * 62 * 4: 6 * 62 * 4])
    sh = shs.reshape(n, 16, 3)
    assert np.allclose(row5[0:3], xyz[5].numpy()) and row5[3:6] == (0.0, 0.0, 0.0)
    assert np.allclose(row5[6:9], sh[5, 0].numpy())
    # f_rest is channel-major: f_rest_{c*15 + (k-1)} = coefficient k, channel c
    assert np.isclose(row5[9 + 1 * 15 + 3], sh[5, 4, 1].item())
    assert np.isclose(row5[9 + 2 * 15 + 14], sh[5, 15, 2].item())
    assert np.isclose(row5[54], op[5, 0].item()) and np.allclose(row5[58:62], rot[5].numpy())
    d = io_ply.load_ply(str(p))
    for k, t in (("xyz", xyz), ("shs48", shs), ("opacity", op), ("scaling", sc), ("rotation", rot)):
        assert torch.equal(d[k], t), k


def test_split_ply_naming_and_reassembly(tmp_path):
    """save_sub_plys writes <stem>_rk{i}_ws{n}.ply slices (scene/__init__.py:262-277 naming) whose
    concatenation is the single-file model."""
    from clm_gs_amd import utils
    from clm_gs_amd.strategies.base_gaussian_model import BaseGaussianModel

    utils.set_args(utils.default_args(bsz=4))
    g = torch.Generator().manual_seed(1)
    n = 23

    class _M(BaseGaussianModel):  # storage-only model on the host: enough for the file formats
        get_features = property(lambda self: self._shs.reshape(-1, 16, 3))

        def create_from_tensors(self, xyz, shs48, scaling, rotation, opacity, spatial_lr_scale=1.0):
            self._xyz, self._shs, self._scaling, self._rotation, self._opacity = xyz, shs48, scaling, rotation, opacity

        def _shs48_rows(self, mask):
            return self._shs

        all_parameters = training_setup = _append_rows = prune_points = reset_opacity = lambda self, *a: None

    m = _M(3)
    m.create_from_tensors(torch.randn(n, 3, generator=g), torch.randn(n, 48, generator=g),
                          torch.randn(n, 3, generator=g), torch.randn(n, 4, generator=g), torch.randn(n, 1, generator=g))
    files = m.save_sub_plys(str(tmp_path / "point_cloud.ply"), 3, 8)
    assert [f.rsplit("/", 1)[1] for f in files] == [f"point_cloud_rk{i}_ws3.ply" for i in range(3)]
    assert [io_ply.load_ply(f)["xyz"].shape[0] for f in files] == [8, 8, 7]
    m2 = _M(3)
    m2.load_sub_plys(str(tmp_path / "point_cloud.ply"), 3)
    for a, b in ((m._xyz, m2._xyz), (m._shs, m2._shs), (m._opacity, m2._opacity), (m._scaling, m2._scaling),
                 (m._rotation, m2._rotation)):
        assert torch.equal(a, b)
