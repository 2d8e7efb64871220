"""Joint optimisation with the COCO settings.

Counterpart of /root/reference/recon/recon_fit_coco.py: `ReconFitterCoco` = `ReconFitterBehave` with
  * `scale_body_kpts` (:32-53): the human-object patch is moved to the mean crop centre of the training set (1008, 995)
    before the 2-D keypoints are mapped into the network input image;
  * `get_loss_weights` (:55-74): stronger pose / contact / keypoint regularisation.
The data loader (`init_dataloader` :20-30, cv2-based `TestData`) is not part of the device path."""
import torch

from .recon_fit_behave import ReconFitterBehave


class ReconFitterCoco(ReconFitterBehave):
    MEAN_CROP_CENTER = (1008.0, 995.0)     # recon_fit_coco.py:43

    def scale_body_kpts(self, kpts, resize_scale, crop_scale, old_crop_center):
        """kpts (B,25,3) in the original image -> coordinates in the network input image, with the crop centre moved to
        the training set's mean crop centre   [recon_fit_coco.py:32-53]"""
        B = old_crop_center.shape[0]
        crop_center = torch.tensor([list(self.MEAN_CROP_CENTER)] * B, device=kpts.device, dtype=kpts.dtype)
        pxy = kpts[:, :, :2] * resize_scale.unsqueeze(1).unsqueeze(1)
        pxy = pxy - old_crop_center.unsqueeze(1) + crop_center.unsqueeze(1)
        crop_org = crop_scale * self.camera.crop_size
        pxy = pxy - crop_center.unsqueeze(1) + crop_org.unsqueeze(1).unsqueeze(1) / 2
        pxy = pxy * self.net_in_size / crop_org.unsqueeze(1).unsqueeze(1)
        return torch.cat([pxy, kpts[:, :, 2:3]], -1)

    def get_loss_weights(self):
        """[recon_fit_coco.py:55-74]"""
        w = {"beta": 10.0 ** 0, "pose": 10.0 ** -5, "hand": 10.0 ** -5, "j2d": 0.8 ** 2, "object": 90.0 ** 2,
             "part": 0.05 ** 2, "contact": 150.0 ** 2, "scale": 2.0 ** 2, "df_h": 30.0 ** 2, "smplz": 30 ** 2,
             "pinit": 10 ** 2, "ocent": 30 ** 2, "mask": 0.3 ** 2, "collide": 15 ** 2, "trans": 10.0 ** 2}
        return {k: (lambda cst, it, c=c: c * cst / (1 + it)) for k, c in w.items()}
