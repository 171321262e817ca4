"""Camera objects exposing what the engines read (scene/cameras.py:39-126,
train.py:278-312): FoVx, FoVy, world_view_transform (row-vector convention,
i.e. the TRANSPOSE of the 4x4 world->camera matrix), K, camtoworlds,
original_image (uint8 [3,H,W] on the GPU), image_name, create_k_on_gpu()."""
import math

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
        self.original_image = image_u8.to(device) if image_u8 is not None else None
        self.K = self.create_k_on_gpu(device)
        self.camtoworlds = torch.inverse(w2c)[None].to(device)  # [1,4,4] as train.py:293-301

    def create_k_on_gpu(self, device="cuda"):
        fx = self.image_width / (2 * math.tan(self.FoVx * 0.5))
        fy = self.image_height / (2 * math.tan(self.FoVy * 0.5))
        return torch.tensor([[fx, 0, self.image_width / 2.0], [0, fy, self.image_height / 2.0],
                             [0, 0, 1]], dtype=torch.float32, device=device)
