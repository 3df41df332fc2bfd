"""CU-contention rehearsal on ONE GPU.

A bounded "channel" kernel (tools/contention_probe.hip -> tools/bin/libgz_probe.so, `make -C tools`) occupies R
workgroup slots of the chip from a side stream -- 256 threads and `lds` bytes of LDS each, streaming a slice of a
buffer -- while the DCGAN step runs on the main stream: what RCCL's channel kernels do underneath backward once the
gradient exchange overlaps it.  For every R the step is timed with the planner's CU budget at 256 (plans sized for an
exclusive chip) and at 256 - R (gz_set_cu_budget, what ddp.GradSync sets).

    python tools/contention_rehearsal.py [out.json]          # default gpurun_out/r04_contention.json

The channel kernel is bounded by an iteration count calibrated to a few seconds; the host only ever shortens its run
(a flag in pinned host memory)."""
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from lightning_gan_zoo_amd._lib import lib  # noqa: E402

PROBE = os.path.join(ROOT, "tools", "bin", "libgz_probe.so")
SLICE_BYTES = 3 << 19            # 1.5 MB per channel workgroup


class Channels:
    def __init__(self, device):
        self.dll = ctypes.CDLL(PROBE)
        self.dll.gz_probe_occupy.restype = ctypes.c_int
        self.dll.gz_probe_occupy.argtypes = [ctypes.c_void_p, ctypes.c_ulonglong, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        self.device = device
        self.side = torch.cuda.Stream(device=device)
        self.flag = torch.zeros(1, dtype=torch.int32).pin_memory()
        self.passes = torch.zeros(256, dtype=torch.int64, device=device)
        self.buf = torch.zeros(64 * SLICE_BYTES // 4, dtype=torch.float32, device=device)

    def launch(self, R, lds, iters, sleep):
        self.flag[0] = 0
        self.passes.zero_()
        torch.cuda.synchronize()
        rc = self.dll.gz_probe_occupy(ctypes.c_void_p(self.buf.data_ptr()), R * SLICE_BYTES, R, lds, iters, sleep,
                                      ctypes.c_void_p(self.flag.data_ptr()), ctypes.c_void_p(self.passes.data_ptr()),
                                      ctypes.c_void_p(self.side.cuda_stream))
        if rc != 0:
            raise RuntimeError("gz_probe_occupy failed: %d" % rc)

    def stop(self):
        self.flag[0] = 1
        self.side.synchronize()
        return int(self.passes.max().item())

    def calibrate(self, R, lds, sleep):
        """Passes per second of one channel workgroup (all R running, nothing else on the chip)."""
        self.launch(R, lds, 200, sleep)
        t0 = time.perf_counter()
        self.side.synchronize()
        dt = time.perf_counter() - t0
        return 200 / max(dt, 1e-4)


def time_steps(trainer, data, steps):
    per = len(trainer.order)
    torch.cuda.current_stream().synchronize()        # (not the device: the channel kernel runs on the side stream)
    t0 = time.perf_counter()
    for _ in range(steps * per):
        trainer.step(data)
    trainer.finish()
    torch.cuda.current_stream().synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r04_contention.json")
    dev = torch.device("cuda", 0)
    torch.set_num_threads(min(8, torch.get_num_threads()))
    ch = Channels(dev)
    rows = []
    for batch, steps in ((128, 40), (512, 16)):
        module, trainer = bench.build_trainer("dc_gan", batch, dev, 1)
        data = bench.synthetic_batch(batch, dev, 0)
        for _ in range(6):
            trainer.step(data)
        import gc
        gc.collect()
        gc.freeze()
        base = min(time_steps(trainer, data, steps) for _ in range(3))
        rows.append({"batch": batch, "R": 0, "lds": 0, "sleep": 0, "cu_budget": 256, "ms_per_step": round(base, 3),
                     "vs_alone": 1.0})
        print("bs %d alone: %.3f ms" % (batch, base), flush=True)
        for (R, lds, sleep) in ((8, 65536, 0), (16, 65536, 0), (32, 65536, 0), (64, 65536, 0), (32, 8192, 0),
                                (32, 65536, 40)):
            rate = ch.calibrate(R, lds, sleep)
            for budget in (256, 256 - R):
                lib.gz_set_cu_budget(budget)
                for _ in range(2):                       # plans (and workspace sizes) follow the budget: warm up
                    trainer.step(data)
                torch.cuda.synchronize()
                best, passes = None, 0
                for _ in range(2):
                    iters = int(rate * (steps * base * 1e-3 * 4 + 0.5))      # bounded: ~4x the expected run
                    ch.launch(R, lds, iters, sleep)
                    ms = time_steps(trainer, data, steps)
                    passes = ch.stop()
                    best = ms if best is None else min(best, ms)
                gbs = passes * SLICE_BYTES * 2 * R / (steps * best * 1e-3) / 1e9 if passes else 0.0
                rows.append({"batch": batch, "R": R, "lds": lds, "sleep": sleep, "cu_budget": budget,
                             "ms_per_step": round(best, 3), "vs_alone": round(best / base, 4),
                             "channel_traffic_GBps": round(gbs, 1), "channel_passes": passes})
                print("bs %d R %d lds %d sleep %d budget %d: %.3f ms (x%.3f), channels moved %.0f GB/s"
                      % (batch, R, lds, sleep, budget, best, best / base, gbs), flush=True)
            lib.gz_set_cu_budget(256)
        gc.unfreeze()
        del trainer, module, data
        torch.cuda.empty_cache()
    json.dump({"note": "dc_gan G+D pair on one MI355X with R channel workgroups (256 threads, `lds` bytes of LDS, "
                       "streaming 1.5 MB slices, `sleep` x 64 clocks between passes) running on a side stream; "
                       "cu_budget = what the convolution planner was told (gz_set_cu_budget)",
               "rows": rows}, open(out_path, "w"), indent=1)
    print("wrote", out_path)


if __name__ == "__main__":
    main()
