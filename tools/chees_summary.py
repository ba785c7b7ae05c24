import json, sys
d = json.load(sys.stdin)
print(round(d["value"] / 1e6, 1), "M/s; pooled ms/step", round(d["pooled_statistics_ms_per_step"], 3), "share", round(d["pooled_statistics_share_of_wall"], 3))
for k, v in d["kernels"].items():
    print(" ", k, round(v["avg_us"], 1), "us", round(v["GBps"]), "GB/s")
