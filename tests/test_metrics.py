"""Quality metrics: the CPU restatement's own properties (no GPU), and the HIP kernels against it.
SSIM parity against scikit-image is UNPINNED (the package is absent; see oracle/metrics_oracle.py)."""
import numpy as np
import pytest
import torch

from oracle import metrics_oracle as MO
from tests.util import dense_like


def _pair(shape, seed):
    rng = np.random.default_rng(seed)
    a = rng.random(shape, dtype=np.float32)
    b = np.clip(a + 0.1 * rng.standard_normal(shape).astype(np.float32), 0, 1).astype(np.float32)
    return torch.from_numpy(a), torch.from_numpy(b)


def test_ssim_restatement_properties():
    a, b = _pair((2, 1, 40, 52), 1)
    assert MO.structural_similarity(a, a) == pytest.approx(1.0, abs=1e-12)
    s_ab, s_ba = MO.structural_similarity(a, b), MO.structural_similarity(b, a)
    assert s_ab == pytest.approx(s_ba, abs=1e-12) and 0.0 < s_ab < 1.0
    # constant images: means only -> (2 ux uy + C1) / (ux^2 + uy^2 + C1), variance terms cancel to C2 / C2
    x, y = torch.full((1, 1, 16, 16), 0.25), torch.full((1, 1, 16, 16), 0.75)
    c1 = (0.01 * 2) ** 2
    assert MO.structural_similarity(x, y) == pytest.approx((2 * 0.25 * 0.75 + c1) / (0.25 ** 2 + 0.75 ** 2 + c1), abs=1e-12)
    # direct 49-term window sums at one pixel against the filtered form
    im1, im2 = a[0, 0].numpy().astype(np.float64), b[0, 0].numpy().astype(np.float64)
    w1, w2 = im1[10:17, 20:27], im2[10:17, 20:27]
    ux, uy = w1.mean(), w2.mean()
    cn = 49 / 48
    vx, vy, vxy = cn * ((w1 * w1).mean() - ux * ux), cn * ((w2 * w2).mean() - uy * uy), cn * ((w1 * w2).mean() - ux * uy)
    c2 = (0.03 * 2) ** 2
    s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux * ux + uy * uy + c1) * (vx + vy + c2))
    from scipy.ndimage import uniform_filter
    U = lambda z: uniform_filter(z, size=7)
    vxm, vym, vxym = cn * (U(im1 * im1) - U(im1) ** 2), cn * (U(im2 * im2) - U(im2) ** 2), cn * (U(im1 * im2) - U(im1) * U(im2))
    S = ((2 * U(im1) * U(im2) + c1) * (2 * vxym + c2)) / ((U(im1) ** 2 + U(im2) ** 2 + c1) * (vxm + vym + c2))
    assert S[13, 23] == pytest.approx(s, rel=1e-10)


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(1, 1, 180, 240), (3, 1, 37, 53), (2, 3, 64, 48), (1, 1, 7, 9)])
def test_metrics_on_device_match_the_cpu_restatement(shape):
    from bde2vid_amd import metrics as M
    a, b = _pair(shape, 7 + shape[-1])
    ad, bd = a.cuda(), b.cuda()
    assert float(M.mse_loss(ad, bd)) == pytest.approx(float(MO.mse_loss(a, b)), rel=2e-6)
    assert M.structural_similarity(ad, bd) == pytest.approx(MO.structural_similarity(a, b), abs=1e-10)
    assert M.structural_similarity(ad, ad) == pytest.approx(1.0, abs=1e-12)
    per = M.mse_per_image(ad, bd).cpu().numpy()
    ref = ((a - b) ** 2).reshape(shape[0], -1).double().mean(dim=1).numpy()
    assert np.allclose(per, ref, rtol=2e-6)


@pytest.mark.gpu
def test_score_sequence_like_eval_model():
    from bde2vid_amd import metrics as M
    pairs = [_pair((1, 1, 60, 72), 100 + i) for i in range(5)]
    mean, detail = M.score_sequence([p[0].cuda() for p in pairs], [p[1] for p in pairs])
    assert len(detail['mse']) == len(detail['ssim']) == 5
    for i, (a, b) in enumerate(pairs):
        assert detail['mse'][i] == pytest.approx(float(MO.mse_loss(a, b)), rel=2e-6)
        assert detail['ssim'][i] == pytest.approx(MO.structural_similarity(a, b), abs=1e-10)
    assert mean['ssim'] == pytest.approx(sum(detail['ssim']) / 5)
    with pytest.raises(RuntimeError):
        M.structural_similarity(torch.zeros(1, 1, 5, 9).cuda(), torch.zeros(1, 1, 5, 9).cuda())
    with pytest.raises(NotImplementedError):
        M.perceptual_loss(None, None)
