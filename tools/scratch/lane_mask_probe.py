"""Probe: as lane_latency_probe.py, with the two streams created by hipExtStreamCreateWithCUMask -- the small chain on a
few reserved CUs, the full-ensemble launches on the others."""
import os, sys, time, json, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from blackjax_amd import _lib
dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
hip = ctypes.CDLL("libamdhip64.so")
def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)()
    for b in bits: words[b // 32] |= (1 << (b % 32))
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)
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
for name, lane_bits in (("8 CUs, bits 0-7", list(range(8))), ("16 CUs, bits 0-15", list(range(16))),
                        ("8 CUs, bits 0,32,..", list(range(0, 256, 32))), ("no mask", None)):
    big, small = mk(32768), mk(128)
    if lane_bits is None:
        sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    else:
        sb = masked_stream(lane_bits)
        sa = masked_stream([b for b in range(256) if b not in lane_bits])
    SEQ = 64
    res = {}
    try:
        with torch.cuda.stream(sb):
            for _ in range(8): pair(sb.cuda_stream, small)
        with torch.cuda.stream(sa):
            for _ in range(4): pair(sa.cuda_stream, big)
        torch.cuda.synchronize()
        cg = torch.cuda.CUDAGraph()
        with torch.cuda.graph(cg, stream=sb):
            for _ in range(SEQ): pair(torch.cuda.current_stream().cuda_stream, small)
        cgb = torch.cuda.CUDAGraph()
        with torch.cuda.graph(cgb, stream=sa):
            for _ in range(64): pair(torch.cuda.current_stream().cuda_stream, big)
        torch.cuda.synchronize()
        def run(reps, load, graph_small=True, graph_big=True):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if load:
                with torch.cuda.stream(sa):
                    b0.record()
                    for _ in range(load):
                        if graph_big: cgb.replay()
                        else:
                            for _ in range(64): pair(sa.cuda_stream, big)
                    b1.record()
                time.sleep(0.002)
            with torch.cuda.stream(sb):
                e0.record()
                for _ in range(reps):
                    if graph_small: cg.replay()
                    else:
                        for _ in range(SEQ): pair(sb.cuda_stream, small)
                e1.record()
            torch.cuda.synchronize()
            r = {"small_us_per_pair": round(e0.elapsed_time(e1) * 1e3 / (reps * SEQ), 2)}
            if load: r["big_us_per_pair"] = round(b0.elapsed_time(b1) * 1e3 / (load * 64), 2)
            return r
        res["alone"] = run(40, 0)
        res["big alone"] = run(0.0001 and 1, 8)["big_us_per_pair"]
        res["under load (graphs)"] = run(60, 16)
        res["under load (big plain launches)"] = run(60, 16, True, False)
        res["under load (both plain)"] = run(20, 16, False, False)
    except Exception as e:
        res["error"] = repr(e)
    out[name] = res
    print(name, json.dumps(res), flush=True)
json.dump(out, open("gpurun_out/lane_mask_probe.json", "w"), indent=1)
