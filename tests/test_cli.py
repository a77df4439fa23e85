"""The trainer's command line is the reference's ("same CLI flags", /root/reference/code/
homography_CNN_synthetic.py:49-85), checked against tests/golden/ref_cli_flags.json -- every parser.add_argument of the reference
and the README's own command lines, read with `ast` by tests/golden/make_golden.py (`cli`).  Also: --num_gpus is never silently
ignored (dist.resolve_num_gpus / dist.self_launch), and find_percentile equals the reference's function on
tests/golden/ref_find_percentile.npz (`percentile`: utils/utils.py:655-672 executed as-is)."""
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, 'golden')


@pytest.fixture(scope='module')
def ref_cli():
    with open(os.path.join(GOLD, 'ref_cli_flags.json')) as f:
        return json.load(f)


@pytest.fixture(scope='module')
def cnn():
    from unsuperviseddeephomographyral2018_amd import homography_CNN_synthetic as m
    return m


def _action(parser, flag):
    for a in parser._actions:
        if flag in a.option_strings:
            return a
    return None


def test_every_reference_flag_is_accepted_with_its_type(ref_cli, cnn):
    flags = ref_cli['flags']
    assert len(flags) == 31                                             # the reference's count (:50-85)
    parser = cnn.build_parser()
    sample = {'str': 'x', 'int': '3', 'float': '0.25', 'str2bool': 'True', None: 'a'}
    py = {'str': str, 'int': int, 'float': float, 'str2bool': bool, None: str}
    for f in flags:
        act = _action(parser, f['flag'])
        assert act is not None, 'reference flag %s (line %d) is not accepted' % (f['flag'], f['line'])
        assert act.nargs == f['nargs'], f['flag']
        if f['choices'] is not None:
            assert list(act.choices) == f['choices'], f['flag']
        # the flag parses a value of the reference's type to that Python type
        val = f['choices'][0] if f['choices'] else sample[f['type']]
        ns = parser.parse_args([f['flag'], val])
        got = getattr(ns, f['flag'].lstrip('-'))
        if f['nargs'] == '+':
            assert got == [val]
        else:
            assert type(got) is py[f['type']], (f['flag'], got)
        # literal defaults are the reference's -- except the two documented departures
        if f['default_is_literal'] and f['flag'] not in ('--num_gpus',):
            dflt = getattr(parser.parse_args([]), f['flag'].lstrip('-'))
            ref_default = f['default']
            if f['type'] == 'str2bool':
                ref_default = ref_default.lower() == 'true'
            assert dflt == ref_default, (f['flag'], dflt, ref_default)
    # --num_gpus: reference default 2 (towers in one process).  Here None = "the launcher decides"; documented in --help
    assert parser.parse_args([]).num_gpus is None
    assert 'launcher' in _action(parser, '--num_gpus').help


def test_readme_command_lines_parse_verbatim(ref_cli, cnn):
    cmds = ref_cli['readme_commands']
    assert [c['line'] for c in cmds] == [125, 170, 174, 180, 185, 302]
    parser = cnn.build_parser()
    for c in cmds:
        ns = parser.parse_args(c['argv'])
        assert ns.mode in ('train', 'test') and ns.loss_type in ('h_loss', 'l1_loss')
        assert ns.lr == float(c['argv'][c['argv'].index('--lr') + 1])
    ns = parser.parse_args(cmds[0]['argv'])                              # README.md:125 ... --visual True
    assert ns.visual is True
    notes = cnn.unsupported_flag_notes(ns, cmds[0]['argv'])
    assert len(notes) == 1 and '--visual' in notes[0]
    # flags nobody set produce no note (save_visual defaults to True in the reference)
    assert cnn.unsupported_flag_notes(parser.parse_args(cmds[3]['argv']), cmds[3]['argv']) == []
    argv = ['--mode', 'test', '--save_visual', 'True', '--I_dir', '/d/I/', '--I_prime_dir=/d/I_prime/']
    assert len(cnn.unsupported_flag_notes(parser.parse_args(argv), argv)) == 2      # --save_visual is honoured (test mode), not noted


def test_num_gpus_is_never_silently_ignored(monkeypatch):
    from unsuperviseddeephomographyral2018_amd import dist as D
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    assert D.resolve_num_gpus(None, 128) == ('run', 1)
    assert D.resolve_num_gpus(1, 128) == ('run', 1)
    assert D.resolve_num_gpus(8, 128) == ('launch', 8)
    with pytest.raises(ValueError, match='split'):
        D.resolve_num_gpus(3, 128)                                       # tf.split(batch, 3) fails in the reference too
    with pytest.raises(ValueError):
        D.resolve_num_gpus(0, 128)
    monkeypatch.setenv('WORLD_SIZE', '4')
    assert D.resolve_num_gpus(None, 128) == ('run', 4)
    assert D.resolve_num_gpus(4, 128) == ('run', 4)
    with pytest.raises(ValueError, match='contradicts the launcher'):
        D.resolve_num_gpus(8, 128)
    with pytest.raises(ValueError, match='split'):
        D.resolve_num_gpus(None, 130)


def test_trainer_launches_itself_for_num_gpus(monkeypatch, cnn, capsys):
    """`... homography_CNN_synthetic --num_gpus 2` with no launcher around it becomes `python -m torch.distributed.run --nnodes=1
    --nproc-per-node 2 --master-addr 127.0.0.1 --master-port <free> -m <this module> <same arguments>`; under a launcher it trains
    in this process (here: reaches train(), which refuses a GPU-less box) and a contradicting WORLD_SIZE is an error."""
    calls = []

    class Launched(Exception):
        pass

    def fake_execv(path, argv):
        calls.append((path, list(argv)))
        raise Launched()
    monkeypatch.setattr(os, 'execv', fake_execv)
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    argv = ['--mode', 'train', '--lr', '5e-4', '--loss_type', 'h_loss', '--visual', 'True', '--num_gpus', '2']
    with pytest.raises(Launched):
        cnn.main(argv)
    path, cmd = calls[0]
    assert path == sys.executable and cmd[:3] == [sys.executable, '-m', 'torch.distributed.run']
    assert '--nnodes=1' in cmd and cmd[cmd.index('--nproc-per-node') + 1] == '2'
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and int(cmd[cmd.index('--master-port') + 1]) > 0
    k = cmd.index(cnn.MODULE)
    assert cmd[k - 1] == '-m' and cmd[k + 1:] == argv
    assert '--visual True' in capsys.readouterr().err
    # under a launcher with the same world size: no re-launch, train() is entered
    calls.clear()
    monkeypatch.setenv('WORLD_SIZE', '2')
    entered = []
    monkeypatch.setattr(cnn, 'train', lambda a: entered.append(a.num_gpus))
    cnn.main(argv)
    assert entered == [2] and calls == []
    # ... with another world size: loud failure, never a silent one-GPU run
    monkeypatch.setenv('WORLD_SIZE', '8')
    with pytest.raises(SystemExit, match='contradicts the launcher'):
        cnn.main(argv)
    # no --num_gpus, no launcher: one rank, in this process
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    entered.clear()
    cnn.main(argv[:-2])
    assert entered == [None] and calls == []


def test_find_percentile_equals_the_reference_function(cnn):
    z = np.load(os.path.join(GOLD, 'ref_find_percentile.npz'))
    keys = sorted(k[2:] for k in z.files if k.startswith('x_'))
    assert 'ref' in keys and 'f32' in keys and len(keys) == 8
    for k in keys:
        x, y = z['x_' + k], z['y_' + k]
        got = cnn.find_percentile(x)
        assert got.shape == (3, 2)
        np.testing.assert_array_equal(np.asarray(got, np.float64), y, err_msg=k)       # NaN == NaN here (empty interval, n = 3)
    # the reference's own test vector (utils/utils.py:675), as the Python list it passes
    np.testing.assert_array_equal(np.asarray(cnn.find_percentile([10, 1.10, 2, 3, 4, 5, 6, 6, 7, 8, 9]), np.float64), z['y_ref'])
    # the test loop hands it a list of float32 scalars, like sess.run's h_loss values (:537)
    x32 = z['x_f32']
    np.testing.assert_array_equal(np.asarray(cnn.find_percentile([np.float32(v) for v in x32]), np.float64), z['y_f32'])


def test_gen_synthetic_data_paths_follow_the_reference_defaults():
    """gen_synthetic_data: default file names under data_path (utils/gen_synthetic_data.py:209-218) and the test-mode numbering
    (:252-255: start_index = num_data, num_data = test_num_data, raw images from test_raw_data_path)."""
    from unsuperviseddeephomographyral2018_amd import gen_synthetic_data as G
    p = G.build_parser()
    for f in ('--mode', '--color', '--raw_data_path', '--test_raw_data_path', '--data_path', '--I_dir', '--I_prime_dir', '--pts1_file',
              '--test_pts1_file', '--num_data', '--test_num_data', '--gt_file', '--test_gt_file', '--filenames_file',
              '--test_filenames_file', '--img_w', '--img_h', '--rho', '--patch_size', '--img_per_real', '--resume', '--start_index'):
        assert _action(p, f) is not None, f                                # the reference's flags (its --visual / --debug / --artifact_mode aside)
    d = p.parse_args(['--data_path', '/d/s'])
    assert (d.mode, d.num_data, d.test_num_data, d.img_w, d.img_h, d.rho, d.patch_size, d.img_per_real, d.resume) == \
        ('test', 100000, 5000, 320, 240, 45, 128, 2, 'N')                  # the reference's defaults (:190-199, :222-249)
    a = G.resolve_paths(p.parse_args(['--data_path', '/d/s', '--mode', 'train', '--raw_data_path', '/raw']))
    assert (a.I_dir, a.I_prime_dir) == ('/d/s/I', '/d/s/I_prime') and a.start_index == 0 and a.raw_data_path == '/raw'
    assert (a.pts1_file, a.gt_file, a.filenames_file) == ('/d/s/pts1.txt', '/d/s/gt.txt', '/d/s/train_synthetic.txt')
    t = G.resolve_paths(p.parse_args(['--data_path', '/d/s', '--num_data', '70', '--test_num_data', '9', '--test_raw_data_path', '/rt']))
    assert (t.start_index, t.num_data, t.raw_data_path) == (70, 9, '/rt')
    assert (t.test_pts1_file, t.test_gt_file, t.test_filenames_file) == ('/d/s/test_pts1.txt', '/d/s/test_gt.txt', '/d/s/test_synthetic.txt')


def test_save_correspondences_img_draws_the_report_image(tmp_path, cnn):
    """--save_visual (reference test loop :539-552 -> utils.save_correspondences_img / draw_matches, utils/utils.py:209-308), with
    PIL: two frames side by side, predicted quadrilateral on the second, ground-truth quadrilaterals, four coloured matches."""
    from PIL import Image
    h, w = 60, 80
    img1 = np.full((h, w, 3), 40, np.uint8); img2 = np.full((h, w, 3), 90, np.uint8)
    c1 = np.array([[20, 15], [50, 15], [50, 45], [20, 45]], np.float32)
    gt = np.array([[3, -2], [-4, 1], [2, 5], [-1, -3]], np.float32)
    pred = gt + 2.0
    path = cnn.save_correspondences_img(img1, img2, c1, c1 + gt, c1 + pred, str(tmp_path / 'report'), '0_l1_loss_loss_1.5.jpg')
    assert os.path.exists(path) and path.endswith('report/0_l1_loss_loss_1.5.jpg')
    out = np.asarray(Image.open(path)).astype(int)
    assert out.shape == (h, 2 * w, 3)
    assert abs(out[2, 2] - 40).max() <= 6 and abs(out[2, w + 2] - 90).max() <= 6            # the two frames, untouched corners
    # the ground-truth edge on the first frame (blue-ish (2, 10, 240), width 3) and the prediction on the second ((5, 225, 225))
    assert out[30, 20, 2] > 150 and out[30, 20, 0] < 90                                      # its left edge (no match line crosses it)
    x, y = int(w + (c1 + pred)[0, 0] + 10), int((c1 + pred)[0, 1])
    band = out[y - 3:y + 4, w + 25:w + 45].reshape(-1, 3)
    assert ((band[:, 1] > 150) & (band[:, 2] > 150) & (band[:, 0] < 110)).any()
    # denorm_img: the dataloader's constants, per channel and for gray
    z = np.zeros((2, 2, 3)); assert np.allclose(cnn.denorm_img(z)[0, 0], [118.93, 113.97, 102.60])
    assert np.allclose(cnn.denorm_img(np.ones((2, 2))), np.mean([118.93, 113.97, 102.60]) + np.mean([69.85, 68.81, 72.45]))
