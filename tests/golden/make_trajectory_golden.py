"""BUILD-CONTAINER ONLY.  Imports the reference's render_bigcity_images.py (never copied) under
tests/golden/ref_harness.py and records what its trajectory generator produces:

    python tests/golden/make_trajectory_golden.py  ->  tests/golden/trajectory_expected.json

  generate_convex_hull_trajectory_v2 (render_bigcity_images.py:149-268): the camera poses along the
  hard-coded BigCity hull for two (n_frames, height) settings -- world_view_transform (row-vector
  convention), camera centre, image_name, FoV -- with the fixed rotation main() uses (:934-936).
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as RH  # noqa: E402


def main():
    RH.install_stubs()
    for name in ("numba.cuda.cudadrv", "numba.cuda.cudadrv.runtime"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["numba.cuda"].__path__ = []
    sys.modules["numba.cuda.cudadrv.runtime"].Runtime = object
    import utils.general_utils as rutils
    rutils.set_log_file(RH.NullLog())
    args, _ = RH.reference_default_args()
    rutils.set_args(args)
    with RH.CudaToCpu():
        import render_bigcity_images as RB
        R_fixed = np.array([[1, 0, 0], [0, 1, 0], [0, 0, -1]])  # render_bigcity_images.py:934-936
        out = {"R_fixed": R_fixed.tolist(), "cases": []}
        for n_frames, hz, w, h in ((24, 30.0, 64, 48), (7, 12.5, 96, 64)):
            fovx, fovy = 1.1, 0.85
            cams = RB.generate_convex_hull_trajectory_v2(R_fixed=R_fixed, height_z=hz, n_frames=n_frames, FoVx=fovx,
                                                         FoVy=fovy, width=w, height=h)
            case = dict(n_frames=n_frames, height_z=hz, width=w, height=h, FoVx=fovx, FoVy=fovy, cameras=[])
            for c in cams:
                wvt = torch.as_tensor(c.world_view_transform).double()
                centre = torch.inverse(wvt.t())[:3, 3]
                case["cameras"].append(dict(image_name=c.image_name, uid=int(c.uid),
                                            world_view_transform=wvt.tolist(), centre=centre.tolist()))
            out["cases"].append(case)
    with open(os.path.join(HERE, "trajectory_expected.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("trajectory golden written:", [len(c["cameras"]) for c in out["cases"]])


if __name__ == "__main__":
    main()
