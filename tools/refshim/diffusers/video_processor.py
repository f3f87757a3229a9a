import numpy as np
import torch


class VideoProcessor:
    """Subset of diffusers.video_processor.VideoProcessor used by the reference pipeline.

    preprocess: tensor [3,H,W] / [1,3,H,W] or PIL image in [0,1] -> [1,3,H,W] in [-1,1] (no resize for tensors).
    postprocess_video: [B,C,F,H,W] in [-1,1] -> numpy [B,F,H,W,C] in [0,1]  ((x/2+0.5).clamp(0,1)).
    """

    def __init__(self, vae_scale_factor=8):
        self.vae_scale_factor = vae_scale_factor

    def preprocess(self, image, height=None, width=None):
        if isinstance(image, torch.Tensor):
            t = image if image.dim() == 4 else image.unsqueeze(0)
        else:
            if height is not None and image.size != (width, height):
                image = image.resize((width, height))
            t = torch.from_numpy(np.array(image).astype(np.float32) / 255.0).permute(2, 0, 1).unsqueeze(0)
        return 2.0 * t - 1.0

    def postprocess_video(self, video, output_type="np"):
        outs = []
        for b in range(video.shape[0]):
            v = video[b].permute(1, 0, 2, 3)
            v = (v / 2 + 0.5).clamp(0, 1)
            outs.append(v.cpu().permute(0, 2, 3, 1).float().numpy())
        return np.stack(outs)
