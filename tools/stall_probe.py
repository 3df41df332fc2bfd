import sys, os, time, gc, torch
sys.path.insert(0, os.getcwd())
import bench
torch.set_num_threads(min(8, torch.get_num_threads()))
dev = torch.device("cuda", 0)
mode = sys.argv[1] if len(sys.argv) > 1 else "gc"
if mode == "nogc":
    gc.disable()
m, t = bench.build_trainer("dc_gan", 128, dev, 1)
b = bench.synthetic_batch(128, dev, 0)
for _ in range(6): t.step(b)
torch.cuda.synchronize()
ts = []
gcs = []
if mode == "gc":
    gc.callbacks.append(lambda phase, info: gcs.append((phase, info.get("generation"), time.perf_counter())))
for c in range(150):
    t0 = time.perf_counter()
    t.step(b); t.step(b)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
big = [(i, round(x, 1)) for i, x in enumerate(ts) if x > 9]
print(mode, "cycles > 9 ms:", big, "median %.2f" % sorted(ts)[len(ts)//2])
if gcs:
    dur = {}
    for (ph, g, tt) in gcs:
        if ph == "start": st = tt
        else: dur.setdefault(g, []).append((tt - st) * 1e3)
    print({g: (len(v), round(max(v), 1)) for g, v in dur.items()})
