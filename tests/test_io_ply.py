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
