"""Camera objects exposing what the engines read (scene/cameras.py:39-126,
train.py:278-312): FoVx, FoVy, world_view_transform (row-vector convention,
i.e. the TRANSPOSE of the 4x4 world->camera matrix), K, camtoworlds,
original_image (uint8 [3,H,W] on the GPU), image_name, create_k_on_gpu()."""
import math

import numpy as np
import torch


class Camera:
    def __init__(self, uid, world_to_cam, FoVx, FoVy, width, height, image_u8=None,
                 image_name=None, device="cuda"):
        self.uid = uid
        self.FoVx, self.FoVy = float(FoVx), float(FoVy)
        self.image_width, self.image_height = int(width), int(height)
        self.image_name = image_name or f"cam_{uid:05d}"
        w2c = torch.as_tensor(world_to_cam, dtype=torch.float32)
        self.world_view_transform = w2c.t().contiguous().to(device)
        # planar and contiguous, as the reference keeps it (scene/cameras.py:74 "image.contiguous()"): the loss
        # kernels read it in place; a strided image would cost fused.camera_forward_finish a copy per camera
        self.original_image = image_u8.to(device).contiguous() if image_u8 is not None else None
        self.K = self.create_k_on_gpu(device)
        c2w = torch.inverse(w2c)
        self.camtoworlds = c2w[None].to(device)  # [1,4,4] as train.py:293-301
        # host copies of the per-camera constants the kernels take by value (fused._cam_host would
        # otherwise read them back from the device: three blocking copies per new camera)
        k_host = self.create_k_on_gpu("cpu")
        self._clmgs_host = (np.ascontiguousarray(w2c.numpy().astype(np.float32).reshape(16)),
                            np.ascontiguousarray(k_host.numpy().astype(np.float32).reshape(9)),
                            np.ascontiguousarray(c2w[:3, 3].numpy().astype(np.float32).reshape(3)))

    def create_k_on_gpu(self, device="cuda"):
        fx = self.image_width / (2 * math.tan(self.FoVx * 0.5))
        fy = self.image_height / (2 * math.tan(self.FoVy * 0.5))
        return torch.tensor([[fx, 0, self.image_width / 2.0], [0, fy, self.image_height / 2.0],
                             [0, 0, 1]], dtype=torch.float32, device=device)
