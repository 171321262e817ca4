"""3DGS .ply exchange format without plyfile (scope row f2; reference:
strategies/base_gaussian_model.py:165-248 save_ply / load_raw_ply).

Binary little-endian, one float32 property per attribute, in the reference's order:
x y z nx ny nz f_dc_0..2 f_rest_0..44 opacity scale_0..2 rot_0..3.  f_rest is CHANNEL-major
(index = c * 15 + (k - 1), from .view(-1,15,3).transpose(1,2).flatten(1) at :214-223) while the
in-memory SH row is coefficient-major (index = k * 3 + c).
"""
import numpy as np
import torch


def attribute_names():
    names = ["x", "y", "z", "nx", "ny", "nz"]
    names += [f"f_dc_{i}" for i in range(3)]
    names += [f"f_rest_{i}" for i in range(45)]
    names += ["opacity"] + [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)]
    return names


def save_ply(path, xyz, shs48, opacity, scaling, rotation):
    """All inputs are the STORED parameters (logit opacity, log scale, raw quaternion)."""
    xyz, shs48, opacity, scaling, rotation = (
        t.detach().float().cpu().numpy() for t in (xyz, shs48, opacity, scaling, rotation))
    n = xyz.shape[0]
    sh = shs48.reshape(n, 16, 3)
    f_dc = sh[:, 0, :]                                     # [n,3]  (c)
    f_rest = sh[:, 1:, :].transpose(0, 2, 1).reshape(n, 45)  # [n, c*15 + (k-1)]
    cols = np.concatenate([xyz, np.zeros_like(xyz), f_dc, f_rest, opacity.reshape(n, 1),
                           scaling.reshape(n, 3), rotation.reshape(n, 4)], axis=1).astype("<f4")
    names = attribute_names()
    assert cols.shape[1] == len(names) == 62
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % n
    header += "".join("property float %s\n" % a for a in names) + "end_header\n"
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(np.ascontiguousarray(cols).tobytes())


def load_ply(path):
    """-> dict(xyz[n,3], shs48[n,48], opacity[n,1], scaling[n,3], rotation[n,4]) float32 tensors.
    Properties are located by NAME (as load_raw_ply does), so extra / reordered columns are fine."""
    with open(path, "rb") as f:
        assert f.readline().strip() == b"ply"
        fmt = f.readline().decode().split()
        assert fmt[1] == "binary_little_endian", fmt
        props, n = [], None
        while True:
            line = f.readline().decode().strip()
            if line == "end_header":
                break
            tok = line.split()
            if tok[0] == "element":
                assert tok[1] == "vertex"
                n = int(tok[2])
            elif tok[0] == "property":
                assert tok[1] in ("float", "float32"), line
                props.append(tok[2])
        data = np.frombuffer(f.read(n * len(props) * 4), dtype="<f4").reshape(n, len(props))
    col = {p: i for i, p in enumerate(props)}
    get = lambda names: np.stack([data[:, col[a]] for a in names], axis=1)
    n_rest = len([p for p in props if p.startswith("f_rest_")])
    assert n_rest == 45, "SH degree 3 expected (3*(3+1)^2 - 3 = 45 f_rest properties)"
    sh = np.zeros((n, 16, 3), np.float32)
    sh[:, 0, :] = get(["f_dc_0", "f_dc_1", "f_dc_2"])
    sh[:, 1:, :] = get([f"f_rest_{i}" for i in range(45)]).reshape(n, 3, 15).transpose(0, 2, 1)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    return dict(xyz=t(get(["x", "y", "z"])), shs48=t(sh.reshape(n, 48)), opacity=t(get(["opacity"])),
                scaling=t(get(["scale_0", "scale_1", "scale_2"])),
                rotation=t(get(["rot_0", "rot_1", "rot_2", "rot_3"])))
