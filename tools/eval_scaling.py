"""Launch time of k_stage_eval (plain / with the step folded in), k_update, k_linesearch against the batch size:
    python tools/eval_scaling.py 131072 262144 524288"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dto_amd
from dto_amd import problems as P
from bench import make_guesses_device, event_time_ms
dev = torch.device("cuda", 0)
p = P.build_acrobot(T=1000, evaluate_hessian=True)
s = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True, name="acrobot")
nz = s.nlp.num_variables
st = torch.cuda.current_stream().cuda_stream
for B in [int(x) for x in sys.argv[1:]]:
    z0 = make_guesses_device(s, p, B, 1000, dev)
    s.begin_batch(z0.data_ptr(), B, nz, stream=st)
    s.iterate_batch(5, stream=st)
    torch.cuda.synchronize()
    mid = ["conv", "kkt_fwd", "kkt_bwd", "kkt_post", "linesearch", "ls_reduce"]
    out = dict(batch=B, hbm_free_gb=round(torch.cuda.mem_get_info(dev)[0] / 1e9, 1))
    t = {}
    s.launch_op("eval", stream=st)
    for r in range(4):
        for o in mid:
            dt = event_time_ms(lambda: s.launch_op(o, stream=st), 1)
            t.setdefault(o, []).append(dt)
        if r % 2 == 0:   # two-kernel way
            t.setdefault("update", []).append(event_time_ms(lambda: s.launch_op("update", stream=st), 1))
            t.setdefault("eval", []).append(event_time_ms(lambda: s.launch_op("eval", stream=st), 1))
        else:
            pass
    # fused passes: an even number
    for r in range(4):
        t.setdefault("update_eval", []).append(event_time_ms(lambda: s.launch_op("update_eval", stream=st), 1))
        for o in mid:
            s.launch_op(o, stream=st)
    s.launch_op("update", stream=st)
    torch.cuda.synchronize()
    out.update({k: round(float(np.mean(v)), 3) for k, v in t.items()})
    print(json.dumps(out), flush=True)
    s.release_state()
    del z0
    torch.cuda.empty_cache()
