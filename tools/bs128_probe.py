import sys, os, time, torch
sys.path.insert(0, os.getcwd())
import bench
torch.set_num_threads(min(8, torch.get_num_threads()))
dev = torch.device("cuda", 0)
def run(bs, cycles, tag):
    m, t = bench.build_trainer("dc_gan", bs, dev, 1)
    b = bench.synthetic_batch(bs, dev, 0)
    for _ in range(6): t.step(b)
    torch.cuda.synchronize()
    ts = []
    for c in range(cycles):
        t0 = time.perf_counter()
        t.step(b); t.step(b)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    print(tag, "per-cycle ms (sync each):", " ".join("%.1f" % x for x in ts))
    t0 = time.perf_counter()
    for c in range(cycles):
        t.step(b); t.step(b)
    torch.cuda.synchronize()
    print(tag, "async avg ms:", (time.perf_counter() - t0) * 1e3 / cycles)
    del m, t
    torch.cuda.empty_cache()
if len(sys.argv) > 1:
    run(512, 20, "bs512")
run(128, 30, "bs128")
run(128, 30, "bs128 again")
