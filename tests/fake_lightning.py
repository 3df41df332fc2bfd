"""Test infrastructure: the hooks ``pl.Trainer.fit(model)`` of the Lightning generation the reference targets (1.1 /
1.2) calls on a LightningModule with two frequency-scheduled optimizers, in its order --
nothing else.  Used to exercise INTEGRATION.md's route 1 (the reference's own ``run_network.py:72`` with the hot-path
``_target_`` strings answered by this package) without pytorch_lightning in the image.

    configure_optimizers()                      -> ({optimizer, lr_scheduler, frequency}, ...)
    train_dataloader(), val_dataloader()
    sanity check: validation_step x <= 2, validation_epoch_end
    per epoch:  per batch (moved to the module's device): the optimizer whose turn it is by ``frequency``,
                toggle_optimizer, training_step(batch, batch_idx, optimizer_idx), backward, step, zero_grad, untoggle
                validation_step x n, validation_epoch_end(outputs); every lr_scheduler.step()
"""
import numpy as np
import torch


class RecordingExperiment:
    def __init__(self):
        self.images = []

    def add_image(self, tag, img, step):
        self.images.append((tag, tuple(img.shape), int(step)))


class _Logger:
    def __init__(self):
        self.experiment = RecordingExperiment()


class TinyImages(torch.utils.data.Dataset):
    """An ImageFolder-shaped dataset (``root``, ``transform``; items = (PIL image, class index)) that needs no files:
    what ``cfg.dataset.{train,val,test}`` points at in the drop-in tests."""

    def __init__(self, root, transform=None, n=10, size=40):
        self.root, self.transform, self.n, self.size = root, transform, int(n), int(size)

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        from PIL import Image
        rng = np.random.RandomState(1000 + i)
        img = Image.fromarray(rng.randint(0, 256, size=(self.size, self.size + 8, 3), dtype=np.uint8))
        if self.transform is not None:
            img = self.transform(img)
        return img, i % 3


class FakeTrainer:
    def __init__(self, max_epochs=1, num_sanity_val_steps=2, limit_val_batches=2):
        self.max_epochs, self.sanity, self.limit_val = max_epochs, num_sanity_val_steps, limit_val_batches
        self.calls = []            # (hook name, detail) in call order
        self.losses = []
        self.logger = _Logger()

    def _to_device(self, batch, device):
        return tuple(t.to(device) if torch.is_tensor(t) else t for t in batch)

    def _validate(self, model, loader, limit):
        outs = []
        for i, batch in enumerate(loader):
            if i >= limit:
                break
            self.calls.append(("validation_step", i))
            with torch.no_grad():
                outs.append(model.validation_step(self._to_device(batch, model.device), i))
        self.calls.append(("validation_epoch_end", len(outs)))
        model.validation_epoch_end(outs)

    def fit(self, model):
        model.logger = self.logger
        self.calls.append(("configure_optimizers", None))
        optim = model.configure_optimizers()
        freqs = [int(o["frequency"]) for o in optim]
        self.calls.append(("train_dataloader", None))
        train = model.train_dataloader()
        self.calls.append(("val_dataloader", None))
        val = model.val_dataloader()
        if self.sanity:
            self._validate(model, val, self.sanity)
        total = 0
        for epoch in range(self.max_epochs):
            model.current_epoch = epoch
            for batch_idx, batch in enumerate(train):
                batch = self._to_device(batch, model.device)
                place = total % sum(freqs)
                opt_idx = int(np.argmax(np.cumsum(freqs) > place))
                mine = {id(p) for g in optim[opt_idx]["optimizer"].param_groups for p in g["params"]}
                saved = [(p, p.requires_grad) for p in model.parameters()]
                for p, _ in saved:                      # toggle_optimizer: only this optimizer's parameters
                    p.requires_grad_(id(p) in mine)
                self.calls.append(("training_step", (batch_idx, opt_idx)))
                loss = model.training_step(batch, batch_idx, opt_idx)
                loss.backward()
                optim[opt_idx]["optimizer"].step()
                optim[opt_idx]["optimizer"].zero_grad()
                for p, rg in saved:                     # untoggle
                    p.requires_grad_(rg)
                self.losses.append(float(loss.detach()))
                total += 1
            self._validate(model, val, self.limit_val)
            for o in optim:
                if o.get("lr_scheduler") is not None:
                    o["lr_scheduler"].step()
        return self
