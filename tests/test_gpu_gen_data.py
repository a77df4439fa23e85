"""GPU: gen_synthetic_data -- the reference's data generator (/root/reference/code/utils/gen_synthetic_data.py) on this library's
DLT + warp kernels: on-disk layout and text formats (:100-126, :209-218), the sampling law (:42-53), img_per_real consecutive
pairs per raw image, test mode continuing the numbering (:252-255), skipped unreadable files (:24-26) -- and I' against the
oracle's f64 warp of the same image by the same homography (the uint8 cast can land one level apart at a .999 / .001 edge)."""
import os

import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

from oracle import hotpath_numpy as O      # noqa: E402  (checker only)


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.skip('no HIP device')
    return torch.device('cuda:0')


def test_generator_from_a_directory_of_photographs(dev, tmp_path, uh_lib_path):
    from PIL import Image
    from unsuperviseddeephomographyral2018_amd import gen_synthetic_data as G
    raw = tmp_path / 'raw'; raw.mkdir()
    rs = np.random.RandomState(0)
    for k, (h, w) in enumerate([(300, 400), (480, 640), (240, 320), (200, 500), (333, 333)]):
        lo = rs.randint(0, 256, size=(h // 16 + 1, w // 16 + 1, 3)).astype(np.uint8)
        Image.fromarray(lo).resize((w, h), Image.BICUBIC).save(str(raw / ('img%d.jpg' % k)), quality=95)
    (raw / '.hidden').write_text('x')                           # removeHiddenfile
    (raw / 'broken.jpg').write_bytes(b'not a jpeg')             # "Error with image": skipped
    out = tmp_path / 'synthetic'
    H, W, P, rho = 120, 160, 64, 20
    common = ['--data_path', str(out), '--img_h', str(H), '--img_w', str(W), '--patch_size', str(P), '--rho', str(rho),
              '--raw_data_path', str(raw), '--test_raw_data_path', str(raw), '--batch', '4']
    n = G.generate(G.build_parser().parse_args(['--mode', 'train', '--num_data', '8'] + common), dev)
    assert n == 8
    nt = G.generate(G.build_parser().parse_args(['--mode', 'test', '--num_data', '8', '--test_num_data', '4'] + common), dev)
    assert nt == 4
    assert sorted(os.listdir(out / 'I'), key=lambda s: int(s[:-4])) == ['%d.jpg' % i for i in range(12)]
    assert sorted(os.listdir(out / 'I_prime')) == sorted(os.listdir(out / 'I'))
    names = open(out / 'train_synthetic.txt').read().split('\n')[:-1]
    assert names == ['%d.jpg %d.jpg' % (i, i) for i in range(8)]
    assert open(out / 'test_synthetic.txt').read().split('\n')[:-1] == ['%d.jpg %d.jpg' % (i, i) for i in range(8, 12)]
    pts1 = np.loadtxt(out / 'pts1.txt'); gt = np.loadtxt(out / 'gt.txt')
    assert pts1.shape == (8, 8) and gt.shape == (8, 8) and np.loadtxt(out / 'test_gt.txt').shape == (4, 8)
    x0, y0 = pts1[:, 0], pts1[:, 1]
    assert (x0 >= rho).all() and (x0 <= W - rho - P).all() and (y0 >= rho).all() and (y0 <= H - rho - P).all()
    assert (pts1 == np.stack([x0, y0, x0 + P, y0, x0 + P, y0 + P, x0, y0 + P], 1)).all()
    assert (np.abs(gt) <= rho).all() and (gt == np.round(gt)).all() and np.abs(gt).max() > rho // 2
    # img_per_real = 2: pairs 2j and 2j + 1 come from the same photograph, with different homographies
    for j in range(4):
        a = np.asarray(Image.open(out / 'I' / ('%d.jpg' % (2 * j))))
        b = np.asarray(Image.open(out / 'I' / ('%d.jpg' % (2 * j + 1))))
        assert a.shape == (H, W, 3) and np.array_equal(a, b) and not np.array_equal(gt[2 * j], gt[2 * j + 1])
    # the reference's own loader reads the result back (same files the trainer takes with --data_path)
    from unsuperviseddeephomographyral2018_amd import dataloader as dl
    prm = dl.dataloader_params(data_path=str(out) + '/', filenames_file=str(out / 'train_synthetic.txt'),
                               pts1_file=str(out / 'pts1.txt'), gt_file=str(out / 'gt.txt'), mode='train', batch_size=4,
                               img_h=H, img_w=W, patch_size=P, augment_list=['normalize'], do_augment=0.0)
    batch = next(dl.Dataloader(prm, shuffle=False, device=dev).stream())
    assert batch['I_aug'].shape == (4, H, W, 3) and torch.equal(batch['gt'].cpu(), torch.tensor(gt[:4], dtype=torch.float32))
    # resume = N starts over; resume = Y appends
    G.generate(G.build_parser().parse_args(['--mode', 'train', '--num_data', '2', '--resume', 'Y', '--start_index', '12'] + common), dev)
    assert np.loadtxt(out / 'pts1.txt').shape == (10, 8) and os.path.exists(out / 'I' / '13.jpg')


def test_warped_frame_equals_the_oracle_warp_up_to_the_uint8_cast(dev, uh_lib_path):
    from unsuperviseddeephomographyral2018_amd import gen_synthetic_data as G
    H, W, P, rho, n = 60, 80, 32, 10, 6
    g = torch.Generator(device=dev).manual_seed(3)
    I = torch.randint(0, 256, (n, H, W, 3), generator=g, device=dev, dtype=torch.uint8)
    pts1, gt = G.sample_law(n, H, W, P, rho, g, dev)
    got = G.warp_pairs(I, pts1, gt).cpu().numpy().astype(np.int32)
    Hm = O.solve_dlt_lapack64(pts1.cpu().numpy(), gt.cpu().numpy())
    theta = O.theta_from_H(Hm, W, H, np.float64)
    ref, _ = O.transformer(I.cpu().numpy().astype(np.float64), theta, (H, W), np.float64)
    want = np.clip(ref, 0, 255).astype(np.uint8).astype(np.int32)
    d = np.abs(got - want)
    assert d.max() <= 1 and (d > 0).mean() < 0.01, (d.max(), (d > 0).mean())
