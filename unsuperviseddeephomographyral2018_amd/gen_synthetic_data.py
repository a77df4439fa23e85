"""Synthetic pair generator from a directory of photographs -- mirror of /root/reference/code/utils/gen_synthetic_data.py with
its own command line (:222-249), so that a user who HAS the raw images (the reference uses MS-COCO; none exist in this
repository) gets the dataset `homography_CNN_synthetic.py --data_path ...` trains from:

    python -m unsuperviseddeephomographyral2018_amd.gen_synthetic_data --mode train --num_data 100000 \
           --raw_data_path /data/coco/train2014/ --data_path /data/synthetic/45/
    python -m unsuperviseddeephomographyral2018_amd.gen_synthetic_data --mode test --num_data 100000 --test_num_data 5000 \
           --test_raw_data_path /data/coco/val2014/ --data_path /data/synthetic/45/

Per raw image (homographyGeneration, :12-137): resize to img_w x img_h; `img_per_real` times draw a patch origin
x ~ U{rho .. W-rho-P}, y ~ U{rho .. H-rho-P} and eight corner jitters ~ U{-rho .. rho} (:42-53), H = the 4-point homography
(cv2.getPerspectiveTransform, :56), I' = numpy_transformer(I, inv(H)) cast to uint8 (:62-64 -> numpy_spatial_transformer.py:
131,141); write I/<index>.jpg, I_prime/<index>.jpg and one row each of pts1 / gt / file names (:100-126).  Test mode numbers
its files from --num_data on and writes the test_* text files (:252-255).

Here the 4-point solve and the warp are THIS library's hot-path kernels (uh_dlt_forward in f64, uh_warp_forward without
`condition`), a batch of pairs per launch; decode, resize and JPEG encoding are PIL on host threads.  Differences from the
reference, all on the host side: cv2.resize (bilinear, no anti-aliasing) -> PIL BILINEAR with reducing_gap off; cv2.imwrite's
JPEG quality 95 is kept; Python's `random` -> a seeded torch generator (--seed); `--visual`, `--debug`, `--artifact_mode` are not
offered.  Without --raw_data_path the images are procedural (synthetic.multiscale_images), which is what
tools/train_from_disk.py uses.
"""
import argparse
import os
import shutil
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import ops, synthetic


def str2bool(s):
    return s.lower() == 'true'


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument('--mode', type=str, default='test', help='Train or test', choices=['train', 'test'])
    p.add_argument('--color', type=str2bool, default='true', help='Generate color or gray images')
    p.add_argument('--raw_data_path', type=str, default='', help='The raw data path (photographs); empty: procedural textures')
    p.add_argument('--test_raw_data_path', type=str, default='', help='The test raw data path')
    p.add_argument('--data_path', type=str, required=True, help='Where the dataset is written (I/, I_prime/, text files)')
    p.add_argument('--I_dir', type=str, default=None)
    p.add_argument('--I_prime_dir', type=str, default=None)
    p.add_argument('--pts1_file', type=str, default=None)
    p.add_argument('--test_pts1_file', type=str, default=None)
    p.add_argument('--num_data', type=int, default=100000, help='The data size for training')
    p.add_argument('--test_num_data', type=int, default=5000, help='The data size for test')
    p.add_argument('--gt_file', type=str, default=None)
    p.add_argument('--test_gt_file', type=str, default=None)
    p.add_argument('--filenames_file', type=str, default=None)
    p.add_argument('--test_filenames_file', type=str, default=None)
    p.add_argument('--img_w', type=int, default=320)
    p.add_argument('--img_h', type=int, default=240)
    p.add_argument('--rho', type=int, default=45)
    p.add_argument('--patch_size', type=int, default=128)
    p.add_argument('--img_per_real', type=int, default=2)
    p.add_argument('--resume', type=str, default='N', help='Y: append to existing data. N: delete old data, create new data')
    p.add_argument('--start_index', type=int, default=0, help='start_index of the new created sample')
    # --- additions ---
    p.add_argument('--seed', type=int, default=0)
    p.add_argument('--batch', type=int, default=128, help='pairs warped per launch')
    p.add_argument('--jpeg_quality', type=int, default=95, help='cv2.imwrite default')
    return p


def resolve_paths(args):
    """The reference's default file names under data_path (:209-218); test mode continues the numbering (:252-255)."""
    d = args.data_path
    args.I_dir = args.I_dir or os.path.join(d, 'I')
    args.I_prime_dir = args.I_prime_dir or os.path.join(d, 'I_prime')
    args.pts1_file = args.pts1_file or os.path.join(d, 'pts1.txt')
    args.gt_file = args.gt_file or os.path.join(d, 'gt.txt')
    args.filenames_file = args.filenames_file or os.path.join(d, 'train_synthetic.txt')
    args.test_pts1_file = args.test_pts1_file or os.path.join(d, 'test_pts1.txt')
    args.test_gt_file = args.test_gt_file or os.path.join(d, 'test_gt.txt')
    args.test_filenames_file = args.test_filenames_file or os.path.join(d, 'test_synthetic.txt')
    if args.mode == 'test':
        args.start_index = args.num_data
        args.num_data = args.test_num_data
        args.raw_data_path = args.test_raw_data_path
    return args


def sample_law(n, H, W, P, rho, generator, device):
    """pts1 [n,8] and gt [n,8] (f32): homographyGeneration :42-53, corner order TL, TR, BR, BL as the reference lists them."""
    if W - 2 * rho - P < 0 or H - 2 * rho - P < 0:
        raise ValueError('patch + 2*rho does not fit the frame')
    x0 = torch.randint(rho, W - rho - P + 1, (n,), generator=generator, device=device)
    y0 = torch.randint(rho, H - rho - P + 1, (n,), generator=generator, device=device)
    pts1 = torch.stack([x0, y0, x0 + P, y0, x0 + P, y0 + P, x0, y0 + P], 1).float()
    gt = torch.randint(-rho, rho + 1, (n, 8), generator=generator, device=device).float()
    return pts1, gt


def warp_pairs(I_u8, pts1, gt):
    """I' = numpy_transformer(I, inv(H)) cast to uint8 for a batch: I_u8 [n,H,W,3] uint8 on the HIP device -> uint8 [n,H,W,3].
    The sampling law is the library's (clip-then-weight bilinear, edge-clamped); the cast truncates like ndarray.astype(uint8)
    on values already inside [0, 255] (a convex combination of uint8 values)."""
    n, H, W, _ = I_u8.shape
    with torch.no_grad():
        _, theta = ops.solve_dlt(pts1, gt, img_w=W, img_h=H, solve_f64=True)
        out, _ = ops.transformer(I_u8.float().contiguous(), theta, (H, W), with_condition=False)
    return out.clamp_(0, 255).to(torch.uint8)


def _load(path, W, H):
    from PIL import Image
    try:
        with Image.open(path) as im:
            im = im.convert('RGB').resize((W, H), Image.BILINEAR, reducing_gap=None)
            return np.asarray(im, dtype=np.uint8)
    except Exception:                                           # noqa: BLE001 -- "Error with image": skipped, as the reference does (:24-26)
        return None


def generate(args, device=None):
    """-> number of pairs written."""
    from PIL import Image
    args = resolve_paths(args)
    dev = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
    test = args.mode == 'test'
    f_names, f_pts1, f_gt = ((args.test_filenames_file, args.test_pts1_file, args.test_gt_file) if test
                             else (args.filenames_file, args.pts1_file, args.gt_file))
    fresh = args.resume.lower() == 'n'
    if fresh:                                                    # dataCollection :141-163
        for f in (f_names, f_pts1, f_gt):
            if os.path.exists(f):
                os.remove(f)
        if not test and args.start_index == 0:
            shutil.rmtree(args.I_dir, ignore_errors=True)
            shutil.rmtree(args.I_prime_dir, ignore_errors=True)
    os.makedirs(args.I_dir, exist_ok=True)
    os.makedirs(args.I_prime_dir, exist_ok=True)
    H, W, P = args.img_h, args.img_w, args.patch_size
    gen = torch.Generator(device=dev).manual_seed(args.seed + (1 if test else 0))
    raw = sorted(n for n in os.listdir(args.raw_data_path) if not n.startswith('.')) if args.raw_data_path else None
    index, last = args.start_index, args.start_index + args.num_data
    written = 0
    with ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 4)) as ex, \
            open(f_names, 'a') as fn, open(f_pts1, 'ab') as fp, open(f_gt, 'ab') as fg:
        cursor = 0
        while index < last:
            want = min(args.batch, last - index)
            n_img = -(-want // args.img_per_real)
            if raw is not None:
                paths = raw[cursor:cursor + n_img]
                cursor += len(paths)
                if not paths:
                    break                                        # the raw images ran out before num_data (as in the reference)
                imgs = [a for a in ex.map(lambda n: _load(os.path.join(args.raw_data_path, n), W, H), paths) if a is not None]
                if not imgs:
                    continue
                base = torch.from_numpy(np.stack(imgs)).to(dev)
            else:
                tex = synthetic.multiscale_images(n_img, H, W, gen, dev)
                base = (tex * 50.0 + 128.0).clamp(0, 255).to(torch.uint8)
            I = base.repeat_interleave(args.img_per_real, dim=0)[:want]        # img_per_real pairs per real image, consecutively
            if not args.color and not test:                                    # gray training frames (:104-105)
                g = I.float().mean(3, keepdim=True).to(torch.uint8)
                I = g.expand(-1, -1, -1, 3).contiguous()
            pts1, gt = sample_law(len(I), H, W, P, args.rho, gen, dev)
            Ip = warp_pairs(I, pts1, gt)
            I_h, Ip_h, pts_h, gt_h = I.cpu().numpy(), Ip.cpu().numpy(), pts1.cpu().numpy(), gt.cpu().numpy()
            jobs = []
            for k in range(len(I_h)):
                name = '%d.jpg' % index
                jobs.append(ex.submit(Image.fromarray(I_h[k]).save, os.path.join(args.I_dir, name), quality=args.jpeg_quality))
                jobs.append(ex.submit(Image.fromarray(Ip_h[k]).save, os.path.join(args.I_prime_dir, name), quality=args.jpeg_quality))
                np.savetxt(fg, [gt_h[k]], delimiter=' ')
                np.savetxt(fp, [pts_h[k]], delimiter=' ')
                fn.write('%s %s\n' % (name, name))
                index += 1
                written += 1
            for j in jobs:
                j.result()
    return written


def main(argv=None):
    args = build_parser().parse_args(argv)
    if not torch.cuda.is_available():
        raise SystemExit('gen_synthetic_data needs an MI355X: the warp has no CPU fallback')
    n = generate(args)
    print('wrote %d pairs under %s (%s mode)' % (n, args.data_path, args.mode))


if __name__ == '__main__':
    main()
