"""Real-data input step (reference core/lightning_module.py:42-47,89-92): ImageFolder ordering,
Resize / ToTensor / Normalize semantics, batching without shuffling, device-side normalisation."""
import os

import numpy as np
import pytest
import torch


def make_folder(root):
    from PIL import Image
    rng = np.random.RandomState(3)
    spec = {"b_class": ["z.png", "a.png", "m.jpg"], "a_class": ["2.png", "10.png", "notes.txt"], "c_class/sub": ["k.png"]}
    for d, names in spec.items():
        os.makedirs(os.path.join(root, d), exist_ok=True)
        for n in names:
            path = os.path.join(root, d, n)
            if n.endswith(".txt"):
                open(path, "w").write("not an image")
                continue
            h, w = rng.randint(20, 50, size=2)
            Image.fromarray(rng.randint(0, 256, size=(h, w, 3), dtype=np.uint8)).save(path)


def test_image_folder_order_and_transform(tmp_path):
    from PIL import Image
    from lightning_gan_zoo_amd.run_network import ImageFolderImages, image_folder_samples
    root = str(tmp_path)
    make_folder(root)
    samples, classes = image_folder_samples(root)
    assert classes == ["a_class", "b_class", "c_class"]
    rel = [(os.path.relpath(p, root), c) for p, c in samples]
    assert rel == [("a_class/10.png", 0), ("a_class/2.png", 0), ("b_class/a.png", 1), ("b_class/m.jpg", 1),
                   ("b_class/z.png", 1), ("c_class/sub/k.png", 2)]
    from cpu_harness import HostNormalisedFolder
    data = ImageFolderImages(root, batch=4, img_size=16, channels=3, mean=0.5, std=0.5, device="cpu")
    with pytest.raises(RuntimeError, match="no CPU fallback"):       # the device half needs the GPU
        next(iter(data))
    it = iter(HostNormalisedFolder(data))
    (x0, l0), (x1, l1), (x2, l2) = next(it), next(it), next(it)
    assert x0.shape == (4, 3, 16, 16) and x1.shape == (2, 3, 16, 16)       # the incomplete last batch is kept
    assert l0.tolist() == [0, 0, 1, 1] and l1.tolist() == [1, 2] and l2.tolist() == l0.tolist()   # next epoch, same order
    img = Image.open(samples[2][0]).convert("RGB").resize((16, 16), Image.BILINEAR)
    ref = (np.asarray(img, dtype=np.float32).transpose(2, 0, 1) / 255.0 - 0.5) / 0.5
    assert np.abs(x0[2].numpy() - ref).max() < 1e-6 and float(x0.min()) >= -1.0 and float(x0.max()) <= 1.0
    grey = ImageFolderImages(root, batch=3, img_size=8, channels=1, mean=0.5, std=0.5, device="cpu")
    assert next(iter(HostNormalisedFolder(grey)))[0].shape == (3, 1, 8, 8)
    # two data-parallel ranks see disjoint halves in DistributedSampler(shuffle=False) order
    halves = [ImageFolderImages(root, 4, 16, 3, 0.5, 0.5, "cpu", rank=r, world=2) for r in range(2)]
    assert [h.order for h in halves] == [[0, 2, 4], [1, 3, 5]]


@pytest.mark.gpu
def test_device_side_normalisation_and_prefetch(tmp_path):
    from lightning_gan_zoo_amd import functional as F
    from lightning_gan_zoo_amd.run_network import ImageFolderImages
    u8 = torch.randint(0, 256, (5, 12, 20, 3), dtype=torch.uint8)
    ref = (u8.permute(0, 3, 1, 2).float() / 255 - 0.4) / 0.25
    out = F.normalize_u8_images(u8.cuda(), 0.4, 0.25)
    assert out.shape == (5, 3, 12, 20) and float((out.cpu() - ref).abs().max()) < 1e-5
    root = str(tmp_path)
    make_folder(root)
    from cpu_harness import HostNormalisedFolder
    cpu = iter(HostNormalisedFolder(ImageFolderImages(root, 4, 16, 3, 0.5, 0.5, "cpu")))
    gpu = iter(ImageFolderImages(root, 4, 16, 3, 0.5, 0.5, "cuda"))
    for _ in range(5):                      # more batches than pinned buffers: the ring is reused
        (xc, lc), (xg, lg) = next(cpu), next(gpu)
        assert torch.equal(lc, lg.cpu()) and float((xc - xg.cpu()).abs().max()) < 1e-5
