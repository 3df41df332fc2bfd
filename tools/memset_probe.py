"""Who issues the device memsets / device-to-device copies of a step?  torch.profiler (CPU + GPU activities) over one
optimizer cycle; every hipMemsetAsync / hipMemcpyAsync runtime call is attributed to the innermost framework operator
that encloses it on its host thread.     python tools/memset_probe.py hologan [img] [--sync]"""
import json
import os
import sys
import tempfile
from collections import Counter

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench      # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
expt = args[0] if args else "hologan"
img = int(args[1]) if len(args) > 1 else bench.NATIVE_IMG_SIZE.get(expt, 64)
sync = "--sync" in sys.argv
batch = {"dc_gan": 128, "hologan": 64, "wgan_gp": 256, "wgan": 512}[expt]
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)


def run():
    module, trainer = bench.build_trainer(expt, batch, dev, 1, force_sync=sync, img_size=img)
    b = bench.synthetic_batch(batch, dev, 0, img)
    n = len(trainer.order)
    for _ in range(2 * n):
        trainer.step(b)
    trainer.finish()
    torch.cuda.synchronize()
    acts = [torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA]
    with torch.profiler.profile(activities=acts) as prof:
        for _ in range(n):
            trainer.step(b)
        trainer.finish()
        torch.cuda.synchronize()
    path = os.path.join(tempfile.mkdtemp(), "trace.json")
    prof.export_chrome_trace(path)
    ev = json.load(open(path))["traceEvents"]
    ops = [e for e in ev if e.get("ph") == "X" and e.get("cat") in ("cpu_op", "user_annotation", "python_function")]
    rt = [e for e in ev if e.get("ph") == "X" and e.get("cat") in ("cuda_runtime", "cuda_driver")
          and ("emset" in e["name"] or "emcpy" in e["name"])]
    print("runtime calls:", Counter(e["name"] for e in rt).most_common())
    who = Counter()
    for r in rt:
        t0, tid = r["ts"], r["tid"]
        enclosing = [o for o in ops if o["tid"] == tid and o["ts"] <= t0 <= o["ts"] + o["dur"]]
        enclosing.sort(key=lambda o: o["dur"])
        chain = " <- ".join(o["name"] for o in enclosing[:3]) or "(no enclosing op)"
        who[(r["name"], chain)] += 1
    for (name, chain), c in who.most_common(40):
        print("%4d  %-22s %s" % (c, name, chain[:150]))
    gpu = Counter(e["name"][:60] for e in ev if e.get("ph") == "X" and e.get("cat") in ("gpu_memset", "gpu_memcpy"))
    print("gpu side:", gpu.most_common(10))


if sync:
    with bench.single_rank_rccl(dev):
        run()
else:
    run()
