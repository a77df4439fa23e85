"""In-HBM synthetic pair generator obeying the reference dataloader's OUTPUT CONTRACT.

The reference feeds HomographyModel from TF queue-runners over JPEG files (dataloader.py:76-235) made
offline by utils/gen_synthetic_data.py; neither COCO nor cv2 exists here, so benchmarks and the trainer
draw pairs with the same sampling law directly on the GPU:

  * patch origin  x0 ~ U{rho .. W-rho-P}, y0 ~ U{rho .. H-rho-P}; pts1 = TL,(x+P,y),(x+P,y+P),(x,y+P);
    gt ~ U{-rho..rho}^8                                              (gen_synthetic_data.py:42-53,118-121)
  * I' (p) = I(H_gt p)  -- produced with THIS library's DLT + warp kernels (f64 solve), which is the
    numpy_transformer(inv(H)) of gen_synthetic_data.py:62-64 without the uint8 round trip
  * images normalised by mean=(118.93,113.97,102.60), std=(69.85,68.81,72.45)        (dataloader.py:99-100)
  * I1/I2 = channel-mean of the normalised frame gathered at the patch (dataloader.py:210-227);
    patch_indices = (v+y0)*W + (u+x0), u fastest                                      (dataloader.py:203-207)

Returns the 9 tensors in the order HomographyModel takes them.
"""
import torch

from . import ops

MEAN_I = (118.93, 113.97, 102.60)
STD_I = (69.85, 68.81, 72.45)


def smooth_images(B, H, W, generator, device):
    """uniform[0,255] noise at 1/8 resolution, bilinearly upsampled, then (x - mean)/std: natural-ish
    spectra (local contrast ~0.3/px) so that photometric gradients are informative."""
    lo = torch.rand(B, 3, H // 8 + 2, W // 8 + 2, generator=generator, device=device) * 255.0
    img = torch.nn.functional.interpolate(lo, size=(H, W), mode='bilinear', align_corners=True)
    mean = torch.tensor(MEAN_I, device=device).view(1, 3, 1, 1)
    std = torch.tensor(STD_I, device=device).view(1, 3, 1, 1)
    return ((img - mean) / std).permute(0, 2, 3, 1).contiguous()           # NHWC


def multiscale_images(B, H, W, generator, device, octaves=(64, 32, 16, 8, 4)):
    """Sum of bilinearly upsampled uniform noise at 1/64 ... 1/4 resolution with amplitude ~ cell size^0.5: a 1/f-like
    spectrum, closer to the photographs the reference trains on (MS-COCO) than the single-octave `smooth` texture.
    The coarse octaves give the photometric loss a gradient that points the right way from tens of pixels out; with
    one 8-px octave alone the loss surface at a 26 px mean displacement is a field of unrelated local minima."""
    img = torch.zeros(B, 3, H, W, device=device)
    for s in octaves:
        lo = torch.rand(B, 3, H // s + 2, W // s + 2, generator=generator, device=device) - 0.5
        w = float(s) ** 0.5
        img = img + w * torch.nn.functional.interpolate(lo, size=(H, W), mode='bilinear', align_corners=True)
    img = img / img.std()                                              # unit variance, like the "(x - mean)/std" frames
    return img.permute(0, 2, 3, 1).contiguous()                       # NHWC


def make_batch(B, img_h=240, img_w=320, patch_size=128, rho=45, seed=0, device='cuda', kind='smooth'):
    """-> dict(I1, I2, I1_aug, I2_aug, I_aug, I_prime_aug, pts1, gt, patch_indices) on `device`."""
    dev = torch.device(device)
    g = torch.Generator(device=dev).manual_seed(seed)
    H, W, P = img_h, img_w, patch_size
    if W - 2 * rho - P < 0 or H - 2 * rho - P < 0:
        raise ValueError('patch + 2*rho does not fit the frame')
    if kind == 'smooth':
        I = smooth_images(B, H, W, g, dev)
    elif kind == 'multiscale':
        I = multiscale_images(B, H, W, g, dev)
    else:
        I = torch.randn(B, H, W, 3, generator=g, device=dev)
    x0 = torch.randint(rho, W - rho - P + 1, (B,), generator=g, device=dev)
    y0 = torch.randint(rho, H - rho - P + 1, (B,), generator=g, device=dev)
    pts1 = torch.stack([x0, y0, x0 + P, y0, x0 + P, y0 + P, x0, y0 + P], 1).float()
    gt = torch.randint(-rho, rho + 1, (B, 8), generator=g, device=dev).float()
    u = torch.arange(P, device=dev)
    patch_indices = ((u[None, :, None] + y0[:, None, None]) * W
                     + (u[None, None, :] + x0[:, None, None])).reshape(B, P * P).int().contiguous()
    with torch.no_grad():
        _, theta = ops.solve_dlt(pts1, gt, img_w=W, img_h=H, solve_f64=True)
        I_prime, _ = ops.transformer(I, theta, (H, W), with_condition=False)
        I1 = ops.gray_patch_gather(I, patch_indices, P)
        I2 = ops.gray_patch_gather(I_prime, patch_indices, P)
    # augment_list == ['normalize'] (the reference default): the *_aug streams equal the plain ones
    return dict(I1=I1, I2=I2, I1_aug=I1, I2_aug=I2, I_aug=I, I_prime_aug=I_prime, pts1=pts1, gt=gt,
                patch_indices=patch_indices)


def model_args(batch):
    """Positional tensors of HomographyModel(args, I1, I2, I1_aug, I2_aug, I_aug, I_prime_aug, h4p, gt, patch_indices)."""
    return (batch['I1'], batch['I2'], batch['I1_aug'], batch['I2_aug'], batch['I_aug'], batch['I_prime_aug'],
            batch['pts1'], batch['gt'], batch['patch_indices'])
