"""Host-side cost of the torch.distributed calls ddp.GradSync makes, single-rank RCCL: all_reduce(async), work.wait(),
event record / stream wait.  python tools/ddp_host_cost.py"""
import os, socket, time
import torch, torch.distributed as dist
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
flat = torch.zeros(12_000_000, device=dev)
def t(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    dt = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize(); return dt
works = []
print("all_reduce(async) 6.5 MB host us:", t(lambda: works.append(dist.all_reduce(flat[:1_600_000], async_op=True))))
print("work.wait() host us:", t(lambda: works.pop().wait(), 100))
works.clear()
print("all_reduce(async) 40 MB host us:", t(lambda: works.append(dist.all_reduce(flat[:10_000_000], async_op=True)), 50))
works.clear()
side = torch.cuda.Stream()
ev = torch.cuda.Event()
print("event record us:", t(lambda: ev.record()))
print("stream.wait_event us:", t(lambda: side.wait_event(ev)))
def under_side():
    ev.record()
    with torch.cuda.stream(side):
        side.wait_event(ev)
        works.append(dist.all_reduce(flat[:1_600_000], async_op=True))
print("record + side-stream all_reduce us:", t(under_side))
torch.cuda.synchronize()
# device time of a single-rank all_reduce
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for n in (1_600_000, 10_000_000):
    e0.record(); w = dist.all_reduce(flat[:n], async_op=True); w.wait(); e1.record(); torch.cuda.synchronize()
    print("device ms all_reduce+wait", n * 4 / 1e6, "MB:", e0.elapsed_time(e1))
# does the call block the host while the compute stream is busy?
a = torch.randn(8192, 8192, device=dev)
torch.cuda.synchronize()
for what in ("all_reduce(async)", "all_reduce + wait", "event record"):
    for _ in range(6):
        a @ a                                  # ~6 x 7 ms of queued GPU work
    t0 = time.perf_counter()
    if what == "event record":
        ev.record()
    else:
        w = dist.all_reduce(flat[:1_600_000], async_op=True)
        if "wait" in what:
            w.wait()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("busy stream: %s host us %.1f (drain afterwards %.1f ms)" % (what, (t1 - t0) * 1e6, (t2 - t1) * 1e3))
x = torch.zeros(1024, device=dev)
print("tiny add_ launch host us:", t(lambda: x.add_(1.0)))
dist.destroy_process_group()
