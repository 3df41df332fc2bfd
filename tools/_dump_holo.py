import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch, scenario
from test_oracle_golden import build_oracle_step, load_golden, set_alpha
from test_parity_gpu import build_product_step
for size in ("tiny", "full"):
    inputs, golden, cond = load_golden("hologan", size)
    step = build_product_step("hologan", size)
    out = scenario.run_scenario(step, inputs, "cuda", full=(size == "tiny"), set_alpha=set_alpha,
                                shadow=build_oracle_step("hologan", size))
    np.savez_compressed(f"gpurun_out/holo_{size}_hip.npz", **{k.replace("/", "|"): np.asarray(v) for k, v in out.items()})
