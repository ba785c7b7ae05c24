"""Probe: latency of a chain of small dependent launches (128 x 256 leapfrog + callable pairs, one recorded sequence)
on a second stream while full-ensemble launches (32 768 x 256) saturate the GPU on the first one."""
import os, sys, time, json, torch
sys.path.insert(0, os.getcwd())
from blackjax_amd import _lib
dev = torch.device("cuda:0")
D = 256
def mk(n):
    return dict(n=n, q=torch.randn(n, D, device=dev), p=torch.randn(n, D, device=dev), g=torch.randn(n, D, device=dev),
                lp=torch.empty(n, device=dev))
imm = torch.ones(D, device=dev); iv = torch.ones(D, device=dev)
def pair(stream, b):
    _lib.call("bjx_leapfrog_diag", stream, b["n"], D, 2, 0.01, None, imm.data_ptr(), 0, b["q"].data_ptr(),
              b["p"].data_ptr(), b["g"].data_ptr(), b["q"].data_ptr(), b["p"].data_ptr())
    _lib.call("bjx_target_diag_gaussian", stream, b["n"], D, iv.data_ptr(), b["q"].data_ptr(), b["lp"].data_ptr(),
              b["g"].data_ptr())
out = {}
for prio in (0, -1):
    big, small = mk(32768), mk(128)
    sa = torch.cuda.Stream(); sb = torch.cuda.Stream(priority=prio)
    SEQ = 64
    with torch.cuda.stream(sb):
        for _ in range(8): pair(sb.cuda_stream, small)
    torch.cuda.synchronize()
    cg = torch.cuda.CUDAGraph()
    with torch.cuda.graph(cg, stream=sb):
        for _ in range(SEQ): pair(torch.cuda.current_stream().cuda_stream, small)
    # the big loop as a graph too (host enqueue rate must not bind)
    with torch.cuda.stream(sa):
        for _ in range(4): pair(sa.cuda_stream, big)
    torch.cuda.synchronize()
    cgb = torch.cuda.CUDAGraph()
    with torch.cuda.graph(cgb, stream=sa):
        for _ in range(64): pair(torch.cuda.current_stream().cuda_stream, big)
    torch.cuda.synchronize()
    def time_small(reps, load):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if load:
            with torch.cuda.stream(sa):
                b0.record()
                for _ in range(load): cgb.replay()
                b1.record()
            time.sleep(0.002)
        with torch.cuda.stream(sb):
            e0.record()
            for _ in range(reps): cg.replay()
            e1.record()
        torch.cuda.synchronize()
        r = {"small_us_per_pair": e0.elapsed_time(e1) * 1e3 / (reps * SEQ)}
        if load: r["big_us_per_pair"] = b0.elapsed_time(b1) * 1e3 / (load * 64)
        return r
    res = {"alone": time_small(40, 0), "alone2": time_small(40, 0)}
    # bulk alone
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(sa):
        e0.record()
        for _ in range(10): cgb.replay()
        e1.record()
    torch.cuda.synchronize()
    res["big_alone_us_per_pair"] = e0.elapsed_time(e1) * 1e3 / 640
    res["under_load"] = time_small(40, 12)   # 12 x 64 big pairs ~ 40-60 ms of load
    res["under_load2"] = time_small(80, 20)
    out[f"priority {prio}"] = res
print(json.dumps(out, indent=1))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/lane_latency_probe.json", "w"), indent=1)
