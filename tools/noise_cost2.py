"""Where does sample_noise spend its host time INSIDE the training loop (busy stream)?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from lightning_gan_zoo_amd import harness
dev = torch.device("cuda", 0)
torch.set_num_threads(8)
module, trainer = bench.build_trainer("dc_gan", 128, dev, 1, False, 64)
data = bench.synthetic_batch(128, dev, 0, 64)
acc = {}
def timed(label, fn):
    t0 = time.perf_counter(); r = fn(); acc[label] = acc.get(label, 0.0) + time.perf_counter() - t0; return r
st = harness._stager
def to_device(t, device):
    device = torch.device(device)
    key = (tuple(t.shape), t.dtype)
    slot = st.slots.get(key)
    if slot is None:
        slot = [[torch.empty(t.shape, dtype=t.dtype).pin_memory() for _ in range(st.depth)], [None] * st.depth, 0]
        st.slots[key] = slot
    bufs, events, cur = slot
    if events[cur] is not None:
        timed("event.synchronize", events[cur].synchronize)
    timed("pinned copy_", lambda: bufs[cur].copy_(t))
    out = timed("to(device, non_blocking)", lambda: bufs[cur].to(device, non_blocking=True))
    ev = timed("Event()", torch.cuda.Event)
    timed("ev.record", lambda: ev.record(torch.cuda.current_stream(device)))
    events[cur] = ev
    slot[2] = (cur + 1) % st.depth
    return out
st.to_device = to_device
orig = module.noise_distn.sample
module.noise_distn.sample = lambda shape: timed("Normal.sample", lambda: orig(shape))
for _ in range(6): trainer.step(data)
torch.cuda.synchronize(); acc.clear()
n = 20
for _ in range(2 * n): trainer.step(data)
torch.cuda.synchronize()
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("%-26s %.1f us per call" % (k, v / (2 * n) * 1e6))
