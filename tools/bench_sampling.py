"""BASELINE.json configs[4]: the autoregressive sampling loop -- 32 independent songs,
music_length 10 -> 10 phrase-encoder calls + 40 bar decodes (encoder + decoder + D2-fixed refiner) -- eager vs
HIP-graph replay.  Prints one JSON line (not the driver's bench contract; that is bench.py)."""
import json, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "musicgeneration_vae-torch_amd"))
import torch
import __graft_entry__ as ge
ge.build()
from graph.model import Model
from hipops import functional as HF
import maker_bar

songs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
length = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda", 0)
torch.manual_seed(0)
gen = Model(use_refiner=True).to(dev).eval()
HF.manual_seed(1)
maker_bar.sample(gen, 1, songs, dev)                    # warm-up (autotune, tables)
torch.cuda.synchronize()


def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best


HF.manual_seed(1)
ref = maker_bar.sample(gen, length, songs, dev)
t_eager = timed(lambda: maker_bar.sample(gen, length, songs, dev))
gs = maker_bar.GraphSampler(gen, songs, dev)
HF.manual_seed(1)
got = gs.sample(length)
same = float((got == ref).float().mean())
t_graph = timed(lambda: gs.sample(length))
bars = songs * length * 4
print(json.dumps({"workload": "sampling loop, %d songs x %d phrases x 4 bars, fp32, refiner (D2-fixed) on" % (songs, length),
                  "eager_s": t_eager, "graph_s": t_graph, "bars_per_s_eager": bars / t_eager, "bars_per_s_graph": bars / t_graph,
                  "ms_per_bar_call_eager": 1e3 * t_eager / (length * 4), "ms_per_bar_call_graph": 1e3 * t_graph / (length * 4),
                  "graph_equals_eager_fraction": same}))
