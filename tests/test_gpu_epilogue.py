"""GPU: the conv epilogues (csrc/uh_epilogue.hip: they keep BITS for the backward, not the activation) through the C ABI, against
plain PyTorch f32 ops of the same function -- relu(y + b) and max_pool2d(relu(y + b), 2, 2) (homography_model.py:88-105) -- on
every element, bit for bit (these are selections and copies: no tolerance), including exact ties inside pooling windows (first
maximum wins, as max_pool2d), all-dead windows, NaN inputs, ragged sizes that end inside a 128-float4 chunk and every channel
count the entry points accept.  The bias gradient is a sum: compared with a tolerance and required to be run-to-run identical."""
import ctypes as C

import pytest

torch = pytest.importorskip('torch')
import torch.nn.functional as F          # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.skip('no HIP device')
    return torch.device('cuda:0')


@pytest.fixture(scope='module')
def lib(uh_lib_path):
    from unsuperviseddeephomographyral2018_amd import _lib
    return _lib.load()


def _p(t):
    return C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _tied(shape, gen, dev, levels=5, nan_frac=0.0):
    """values on a coarse grid (many exact ties and exact zeros after the bias), optionally a few NaNs"""
    t = torch.randint(-levels, levels + 1, shape, generator=gen, device=dev).float() * 0.5
    if nan_frac:
        t[torch.rand(shape, generator=gen, device=dev) < nan_frac] = float('nan')
    return t


@pytest.mark.parametrize('npix,Cc', [(1, 4), (31, 4), (33, 8), (1000, 64), (4097, 64), (777, 128), (129, 256), (65, 512),
                                     (37, 1024), (64 * 48 * 48, 64)])
def test_bias_relu_bits_forward_backward(lib, dev, npix, Cc):
    from unsuperviseddeephomographyral2018_amd import _lib
    g = torch.Generator(device=dev).manual_seed(npix * 7 + Cc)
    y0 = _tied((npix, Cc), g, dev)
    bias = _tied((Cc,), g, dev, levels=2)
    gy = torch.randn(npix, Cc, generator=g, device=dev)
    ref = F.relu(y0 + bias)
    y = y0.clone()
    nmask = lib.uh_relu_mask_bytes(npix, Cc)
    assert nmask == ((npix * Cc // 4 + 127) // 128) * 64
    mask = torch.full((nmask + 64,), 0xAB, dtype=torch.uint8, device=dev)         # + guard bytes
    _lib.check(lib.uh_bias_relu_forward(_p(y), _p(bias), _p(mask), npix, Cc, _stream()), 'fwd')
    assert torch.equal(y, ref)
    assert bool((mask[nmask:] == 0xAB).all())                                     # nothing written past the mask
    nws = lib.uh_bias_relu_backward_workspace_bytes(npix, Cc)
    outs = []
    for _ in range(2):
        ws = torch.empty(nws // 4, device=dev)
        gout = torch.full((npix, Cc), float('nan'), device=dev)
        db = torch.empty(Cc, device=dev)
        _lib.check(lib.uh_bias_relu_backward(_p(mask), _p(gy), _p(gout), _p(db), _p(ws), nws, npix, Cc, _stream()), 'bwd')
        outs.append((gout, db))
    want = torch.where(ref > 0, gy, torch.zeros_like(gy))
    assert torch.equal(outs[0][0], want)
    assert torch.equal(outs[0][1], outs[1][1])                                     # deterministic reduction
    dbw = want.double().sum(0)
    assert float((outs[0][1].double() - dbw).abs().max()) <= 1e-5 * max(1.0, float(want.abs().sum(0).max()))
    # forward only: no mask, same activation
    y2 = y0.clone()
    _lib.check(lib.uh_bias_relu_forward(_p(y2), _p(bias), None, npix, Cc, _stream()), 'fwd nomask')
    assert torch.equal(y2, ref)


@pytest.mark.parametrize('N,Hh,Ww,Cc,nan', [(1, 2, 2, 4, 0.0), (2, 6, 10, 8, 0.0), (3, 8, 8, 64, 0.0), (2, 14, 6, 128, 0.0),
                                            (1, 4, 4, 1024, 0.0), (2, 8, 8, 64, 0.02), (16, 64, 64, 64, 0.0)])
def test_bias_relu_pool_bits_forward_backward(lib, dev, N, Hh, Ww, Cc, nan):
    from unsuperviseddeephomographyral2018_amd import _lib
    g = torch.Generator(device=dev).manual_seed(N * 1000 + Hh * 10 + Cc)
    y0 = _tied((N, Hh, Ww, Cc), g, dev, levels=3, nan_frac=nan)                    # NHWC storage
    bias = _tied((Cc,), g, dev, levels=1)
    gp = torch.randn(N, Hh // 2, Ww // 2, Cc, generator=g, device=dev)
    # torch: NCHW-logical views of the NHWC storage.  NaN: fmaxf(NaN, 0) = 0 in the epilogue (documented: a NaN conv output is
    # dead), while F.relu propagates it -> compare against relu with NaN -> 0
    a = (y0 + bias).permute(0, 3, 1, 2)
    a = torch.where(torch.isnan(a), torch.zeros_like(a), a).clamp_min(0).detach().requires_grad_(True)
    pooled_ref = F.max_pool2d(a, 2, 2)
    pooled_ref.backward(gp.permute(0, 3, 1, 2))
    # max_pool2d routes to the first maximum of a window; relu's mask: windows whose maximum is 0 give no gradient
    g_ref = (a.grad * (a.detach() > 0)).permute(0, 2, 3, 1).contiguous()
    y = y0.clone()
    pooled = torch.empty(N, Hh // 2, Ww // 2, Cc, device=dev)
    nmask = lib.uh_pool_mask_bytes(N, Hh, Ww, Cc)
    assert nmask == N * (Hh // 2) * (Ww // 2) * (Cc // 4) * 2
    mask = torch.full((nmask + 64,), 0xAB, dtype=torch.uint8, device=dev)
    _lib.check(lib.uh_bias_relu_pool_forward(_p(y), _p(bias), _p(pooled), _p(mask), N, Hh, Ww, Cc, _stream()), 'pool fwd')
    assert torch.equal(pooled, pooled_ref.detach().permute(0, 2, 3, 1))
    same = (y == y0) | (torch.isnan(y) & torch.isnan(y0))
    assert bool(same.all())                                                        # the conv output is left untouched
    assert bool((mask[nmask:] == 0xAB).all())
    # forward only (no backward will follow): no mask, same pooled map
    pooled2 = torch.empty_like(pooled)
    _lib.check(lib.uh_bias_relu_pool_forward(_p(y), _p(bias), _p(pooled2), None, N, Hh, Ww, Cc, _stream()), 'pool fwd nomask')
    assert torch.equal(pooled2, pooled)
    nws = lib.uh_bias_relu_pool_backward_workspace_bytes(N, Hh, Ww, Cc)
    outs = []
    for _ in range(2):
        ws = torch.empty(nws // 4, device=dev)
        gout = torch.full((N, Hh, Ww, Cc), float('nan'), device=dev)
        db = torch.empty(Cc, device=dev)
        _lib.check(lib.uh_bias_relu_pool_backward(_p(mask), _p(gp), _p(gout), _p(db), _p(ws), nws, N, Hh, Ww, Cc, _stream()),
                   'pool bwd')
        outs.append((gout, db))
    assert torch.equal(outs[0][0], g_ref)
    assert torch.equal(outs[0][1], outs[1][1])
    dbw = g_ref.double().sum((0, 1, 2))
    assert float((outs[0][1].double() - dbw).abs().max()) <= 1e-5 * max(1.0, float(g_ref.abs().sum((0, 1, 2)).max()))


def test_epilogue_argument_errors(lib, dev):
    one = _p(torch.zeros(4096, device=dev))
    assert lib.uh_relu_mask_bytes(100, 6) == 0 and lib.uh_pool_mask_bytes(2, 7, 8, 64) == 0
    assert lib.uh_bias_relu_forward(one, None, one, 16, 64, None) == -1
    assert lib.uh_bias_relu_forward(one, one, one, 16, 96, None) == -3
    assert lib.uh_bias_relu_backward(one, one, one, one, None, 0, 16, 64, None) == -4
    assert lib.uh_bias_relu_pool_forward(one, one, one, one, 1, 3, 4, 64, None) == -2
    assert lib.uh_bias_relu_pool_backward(None, one, one, one, one, 1 << 20, 1, 4, 4, 64, None) == -1


def test_fuzz_epilogues_random_shapes(lib, dev):
    """40 random shapes per kernel pair (channel counts from the accepted set, pixel counts that end anywhere inside a chunk, odd
    batch sizes), random data with ties: forward bits and backward routing against torch on every element."""
    from unsuperviseddeephomographyral2018_amd import _lib
    rs = __import__('numpy').random.RandomState(11)
    chans = [4, 8, 16, 32, 64, 128, 256]
    for it in range(40):
        Cc = int(chans[rs.randint(len(chans))])
        g = torch.Generator(device=dev).manual_seed(1000 + it)
        # ---- bias + ReLU
        npix = int(rs.randint(1, 3000))
        y0 = _tied((npix, Cc), g, dev, levels=4); bias = _tied((Cc,), g, dev, levels=2)
        gy = torch.randn(npix, Cc, generator=g, device=dev)
        y = y0.clone(); mask = torch.empty(lib.uh_relu_mask_bytes(npix, Cc), dtype=torch.uint8, device=dev)
        _lib.check(lib.uh_bias_relu_forward(_p(y), _p(bias), _p(mask), npix, Cc, _stream()), 'fwd')
        ref = F.relu(y0 + bias)
        assert torch.equal(y, ref), (it, npix, Cc)
        nws = lib.uh_bias_relu_backward_workspace_bytes(npix, Cc)
        ws = torch.empty(nws // 4, device=dev); gout = torch.empty_like(gy); db = torch.empty(Cc, device=dev)
        _lib.check(lib.uh_bias_relu_backward(_p(mask), _p(gy), _p(gout), _p(db), _p(ws), nws, npix, Cc, _stream()), 'bwd')
        want = torch.where(ref > 0, gy, torch.zeros_like(gy))
        assert torch.equal(gout, want), (it, npix, Cc)
        assert float((db.double() - want.double().sum(0)).abs().max()) <= 1e-5 * max(1.0, float(want.abs().sum(0).max()))
        # ---- bias + ReLU + pool
        N, Hh, Ww = int(rs.randint(1, 6)), 2 * int(rs.randint(1, 12)), 2 * int(rs.randint(1, 12))
        y0 = _tied((N, Hh, Ww, Cc), g, dev, levels=3); gp = torch.randn(N, Hh // 2, Ww // 2, Cc, generator=g, device=dev)
        a = (y0 + bias).permute(0, 3, 1, 2).clamp_min(0).detach().requires_grad_(True)
        pr = F.max_pool2d(a, 2, 2); pr.backward(gp.permute(0, 3, 1, 2))
        g_ref = (a.grad * (a.detach() > 0)).permute(0, 2, 3, 1).contiguous()
        pooled = torch.empty(N, Hh // 2, Ww // 2, Cc, device=dev)
        mask = torch.empty(lib.uh_pool_mask_bytes(N, Hh, Ww, Cc), dtype=torch.uint8, device=dev)
        _lib.check(lib.uh_bias_relu_pool_forward(_p(y0), _p(bias), _p(pooled), _p(mask), N, Hh, Ww, Cc, _stream()), 'pool fwd')
        assert torch.equal(pooled, pr.detach().permute(0, 2, 3, 1)), (it, N, Hh, Ww, Cc)
        nws = lib.uh_bias_relu_pool_backward_workspace_bytes(N, Hh, Ww, Cc)
        ws = torch.empty(nws // 4, device=dev); gout = torch.empty(N, Hh, Ww, Cc, device=dev); db = torch.empty(Cc, device=dev)
        _lib.check(lib.uh_bias_relu_pool_backward(_p(mask), _p(gp), _p(gout), _p(db), _p(ws), nws, N, Hh, Ww, Cc, _stream()), 'pool bwd')
        assert torch.equal(gout, g_ref), (it, N, Hh, Ww, Cc)
