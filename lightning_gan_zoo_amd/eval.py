"""Evaluation path of the reference's ``InceptionMetrics`` callback (core/callback_inception_metrics.py:136-246;
SURVEY.md 8-f3) minus the Inception network itself, whose weights are fetched from a URL
(core/submodules/gan_stability/metrics/inception.py:13) and cannot be obtained offline:

* ``SampleDump``   -- the fixed latent set drawn once from the HOST generator at construction (:166-168) and the
                      eval-mode generator sweep over it on the GPU (:183-203), producing the same uint8 HWC images
                      the callback writes to PNG files (including its clamp of the tanh output to [0, 1]);
* ``activation_statistics`` / ``frechet_distance`` -- FID from two activation sets (:223-231; the formula of
                      core/submodules/gan_stability/metrics/fid_score.py:25-80);
* ``polynomial_mmd_averages`` -- KID (:15-133, :233-234).

``evaluate`` strings them together around any ``feature_fn(images_uint8_nhwc) -> [n, d]`` feature extractor.
The arithmetic is pinned to the reference's own functions by tests/golden/eval_metrics.npz
(tests/golden/make_eval_golden.py).  Host-side numpy / scipy like the reference; only the generator runs on the GPU.
"""
import numpy as np
import torch


class SampleDump:
    def __init__(self, module, n_samples=5000, batch_size=16):
        # drawn from the host generator right after the module is built, like the callback's __init__ (:166-168)
        self.n_samples, self.batch_size = n_samples, batch_size
        self.z_samples = torch.split(module.noise_distn.sample((n_samples, module.cfg.model.noise_dim)), batch_size)

    @torch.no_grad()
    def images(self, module):
        """Yields uint8 arrays [b, H, W, 3]: generator in eval mode over the fixed latents (:186-198)."""
        was_training = module.training
        module.eval()
        try:
            for z in self.z_samples:
                samples = module.generator(z.to(module.device))
                if samples.shape[1] == 1:                     # greyscale -> RGB (:193-195)
                    samples = torch.cat(3 * [samples], dim=1)
                samples = torch.clamp(samples, 0, 1)          # sic: the tanh output is clamped, not de-normalised (:197)
                samples = samples.permute(0, 2, 3, 1).detach().cpu().numpy()
                yield (samples * 255).astype(int).astype(np.uint8)
        finally:
            module.train(was_training)


def make_grid(images, nrow=8, padding=2, normalize=False, pad_value=0.0):
    """torchvision.utils.make_grid for a [N, C, H, W] tensor (the subset the reference uses,
    core/lightning_module.py:68-69: ``normalize=True``, defaults otherwise): with ``normalize`` the WHOLE tensor is
    shifted / scaled to [0, 1] by its own min / max (``scale_each=False``, the 1e-5 guard of torchvision's
    ``norm_ip``); images are laid out ``nrow`` per row with ``padding`` pixels of ``pad_value`` around each;
    single-channel input is repeated to three channels."""
    t = images.detach()
    if t.dim() == 3:
        t = t.unsqueeze(0)
    if t.shape[1] == 1:
        t = torch.cat((t, t, t), 1)
    if normalize:
        t = t.clone()
        lo, hi = float(t.min()), float(t.max())
        t = t.clamp(min=lo, max=hi).sub(lo).div(max(hi - lo, 1e-5))
    n = t.shape[0]
    if n == 1:
        return t.squeeze(0)
    xmaps = min(nrow, n)
    ymaps = int(np.ceil(float(n) / xmaps))
    height, width = int(t.shape[2] + padding), int(t.shape[3] + padding)
    grid = t.new_full((t.shape[1], height * ymaps + padding, width * xmaps + padding), pad_value)
    k = 0
    for y in range(ymaps):
        for x in range(xmaps):
            if k >= n:
                break
            grid[:, y * height + padding:(y + 1) * height, x * width + padding:(x + 1) * width] = t[k]
            k += 1
    return grid


def activation_statistics(act):
    act = np.asarray(act)
    return np.mean(act, axis=0), np.cov(act, rowvar=False)


def frechet_distance(mu1, sigma1, mu2, sigma2, eps=1e-6):
    """d^2 = |mu1 - mu2|^2 + Tr(S1 + S2 - 2 (S1 S2)^(1/2)); returns nan when the matrix square root has a
    non-negligible imaginary diagonal (fid_score.py:25-80)."""
    from scipy import linalg
    mu1, mu2 = np.atleast_1d(mu1), np.atleast_1d(mu2)
    sigma1, sigma2 = np.atleast_2d(sigma1), np.atleast_2d(sigma2)
    if mu1.shape != mu2.shape or sigma1.shape != sigma2.shape:
        raise ValueError("the two statistics have different dimensions")
    delta = mu1 - mu2
    root, _ = linalg.sqrtm(sigma1.dot(sigma2), disp=False)
    if not np.isfinite(root).all():          # nearly singular product: regularise both covariances
        ridge = np.eye(sigma1.shape[0]) * eps
        root = linalg.sqrtm((sigma1 + ridge).dot(sigma2 + ridge))
    if np.iscomplexobj(root):
        if not np.allclose(np.diagonal(root).imag, 0, atol=1e-3):
            return float("nan")
        root = root.real
    return float(delta.dot(delta) + np.trace(sigma1) + np.trace(sigma2) - 2 * np.trace(root))


def _poly_kernel(x, y, degree, gamma, coef0):
    if gamma is None:
        gamma = 1.0 / x.shape[1]
    return (gamma * x.dot(y.T) + coef0) ** degree


def _sq(a):
    a = np.ravel(a)
    return a.dot(a)


def _mmd2_and_variance(k_xx, k_xy, k_yy, var_at_m=None, ret_var=True):
    """Unbiased MMD^2 estimate and its variance estimate from three m x m kernel matrices
    (callback_inception_metrics.py:57-133, the 'unbiased' estimator with explicit diagonals)."""
    m = k_xx.shape[0]
    if var_at_m is None:
        var_at_m = m
    dx, dy = np.diagonal(k_xx), np.diagonal(k_yy)
    rx = k_xx.sum(axis=1) - dx              # row sums without the diagonal
    ry = k_yy.sum(axis=1) - dy
    cxy0, cxy1 = k_xy.sum(axis=0), k_xy.sum(axis=1)
    sx, sy, sxy = rx.sum(), ry.sum(), cxy0.sum()
    mmd2 = (sx + sy) / (m * (m - 1)) - 2 * sxy / (m * m)
    if not ret_var:
        return mmd2
    sx2 = _sq(k_xx) - _sq(dx)
    sy2 = _sq(k_yy) - _sq(dy)
    sxy2 = _sq(k_xy)
    dxx = rx.dot(cxy1)
    dyy = ry.dot(cxy0)
    m1, m2 = m - 1, m - 2
    zeta1 = (1 / (m * m1 * m2) * (_sq(rx) - sx2 + _sq(ry) - sy2)
             - 1 / (m * m1) ** 2 * (sx ** 2 + sy ** 2)
             + 1 / (m * m * m1) * (_sq(cxy1) + _sq(cxy0) - 2 * sxy2)
             - 2 / m ** 4 * sxy ** 2
             - 2 / (m * m * m1) * (dxx + dyy)
             + 2 / (m ** 3 * m1) * (sx + sy) * sxy)
    zeta2 = (1 / (m * m1) * (sx2 + sy2)
             - 1 / (m * m1) ** 2 * (sx ** 2 + sy ** 2)
             + 2 / (m * m) * sxy2
             - 2 / m ** 4 * sxy ** 2
             - 4 / (m * m * m1) * (dxx + dyy)
             + 4 / (m ** 3 * m1) * (sx + sy) * sxy)
    var = 4 * (var_at_m - 2) / (var_at_m * (var_at_m - 1)) * zeta1 + 2 / (var_at_m * (var_at_m - 1)) * zeta2
    return mmd2, var


def polynomial_mmd(codes_g, codes_r, degree=3, gamma=None, coef0=1, var_at_m=None, ret_var=True):
    """k(x, y) = (gamma <x, y> + coef0)^degree, gamma = 1 / dim by default (:42-55)."""
    k_xx = _poly_kernel(codes_g, codes_g, degree, gamma, coef0)
    k_yy = _poly_kernel(codes_r, codes_r, degree, gamma, coef0)
    k_xy = _poly_kernel(codes_g, codes_r, degree, gamma, coef0)
    return _mmd2_and_variance(k_xx, k_xy, k_yy, var_at_m=var_at_m, ret_var=ret_var)


def polynomial_mmd_averages(codes_g, codes_r, n_subsets=50, subset_size=1000, ret_var=True, **kernel_args):
    """KID: MMD^2 over ``n_subsets`` random subsets drawn with numpy's GLOBAL generator, without replacement, the
    generated codes first (:19-40).  Note the callback passes (real_act, fake_act) in that order (:233)."""
    subset_size = min(len(codes_g), len(codes_r), subset_size)
    m = min(codes_g.shape[0], codes_r.shape[0])
    mmds = np.zeros(n_subsets)
    variances = np.zeros(n_subsets)
    for i in range(n_subsets):
        g = codes_g[np.random.choice(len(codes_g), subset_size, replace=False)]
        r = codes_r[np.random.choice(len(codes_r), subset_size, replace=False)]
        o = polynomial_mmd(g, r, **kernel_args, var_at_m=m, ret_var=ret_var)
        if ret_var:
            mmds[i], variances[i] = o
        else:
            mmds[i] = o
    return (mmds, variances) if ret_var else mmds


def evaluate(module, dump, feature_fn, real_act, n_subsets=100):
    """FID / KID of ``module.generator`` against real activations ``real_act`` (:205-238) with a pluggable
    feature extractor in place of InceptionV3's 2048-d pool features."""
    fake_act = np.concatenate([np.asarray(feature_fn(img)) for img in dump.images(module)], axis=0)
    real_mu, real_sigma = activation_statistics(real_act)
    fake_mu, fake_sigma = activation_statistics(fake_act)
    fid = frechet_distance(real_mu, real_sigma, fake_mu, fake_sigma)
    kid = polynomial_mmd_averages(np.asarray(real_act), fake_act, n_subsets=n_subsets)
    return {"fid": fid, "kid": float(kid[0].mean()), "kid_std": float(kid[0].std())}


def fid_from_weight_file(module, real_images_u8, weights=None, n_samples=5000, batch_size=16, n_subsets=100):
    """FID / KID end to end (reference core/callback_inception_metrics.py:183-246): the fixed latent set, the eval-mode
    generator sweep, InceptionV3 pool features on the HIP kernels with the reference's weight FILE, Frechet distance and
    KID.  ``weights``: path of pytorch-fid's ``pt_inception-2015-12-05-6726825d.pth`` (default: the ``GZ_FID_WEIGHTS``
    environment variable; the reference downloads it, metrics/inception.py:13 -- there is no network here); the file's
    sha256 prefix is verified.  ``real_images_u8``: uint8 [n, H, W, 3] real images, or None to compare the generated
    set with itself (a self-check: FID 0)."""
    import os
    from .inception import InceptionFeatures, load_fid_weights
    weights = weights or os.environ.get("GZ_FID_WEIGHTS")
    if not weights:
        raise RuntimeError("lightning_gan_zoo_amd.eval: FID needs the published Inception weight file; pass weights=<path> "
                           "or set GZ_FID_WEIGHTS (pt_inception-2015-12-05-6726825d.pth)")
    features = InceptionFeatures(load_fid_weights(weights, module.device), batch_size)
    dump = SampleDump(module, n_samples, batch_size)
    if real_images_u8 is None:
        real_act = np.concatenate([np.asarray(features(img)) for img in dump.images(module)], axis=0)
    else:
        real_act = features(real_images_u8)
    return evaluate(module, dump, features, real_act, n_subsets=n_subsets)
