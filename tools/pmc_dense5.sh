#!/bin/bash
# Round 5: what does the C5 dense leapfrog launch wait on?  Counters-only passes (separate --pmc
# runs, no tracing in them) + one kernel-trace pass for the durations, over tools/dense_pmc_workload.py: the fused
# launch k_dense_gemm_tn8<1, 2>, the plain GEMM on the same core loop <0, 0>, and the vendor library's fp32 GEMM.
# JSON -> stdout (copy into profiles/r05/dense_c5_pmc.json).  BJX_PMC_TAG names a variant build.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_dense5${BJX_PMC_TAG:+_$BJX_PMC_TAG}
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
W="python $R/tools/dense_pmc_workload.py"
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- $W > $OUT/kt.log 2>&1
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" \
           "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_LDS SQ_WAVES" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VALU" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- $W > $OUT/g$i.log 2>&1
done
cd $R
python - "$OUT" <<'PY'
import csv, glob, json, collections, re, sys
out = sys.argv[1]
def name_of(k):
    m = re.search(r'k_dense_gemm_(tn8|tnw)<\s*(\d)\s*,\s*(\d)\s*>', k)
    if m:
        base = {"12": "fused_%s<EPI_DRIFT,2>", "00": "plain_%s<EPI_STORE,0>"}.get(m.group(2) + m.group(3))
        return base % m.group(1) if base else None
    if "k_dense" in k or "bjx" in k:
        return None
    if re.search(r'Cijk|gemm|sgemm', k):
        return "vendor:" + k[:60]
    return None
res = collections.defaultdict(dict)
dur = collections.defaultdict(list)
for f in glob.glob(out + '/kt/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        n = name_of(r['Kernel_Name'])
        if n:
            dur[n].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
            res[n]["vgpr"] = r.get("VGPR_Count"); res[n]["accum_vgpr"] = r.get("Accum_VGPR_Count")
            res[n]["sgpr"] = r.get("SGPR_Count"); res[n]["lds_bytes"] = r.get("LDS_Block_Size")
            res[n]["workgroup"] = r.get("Workgroup_Size_X"); res[n]["grid"] = r.get("Grid_Size_X")
for n, d in dur.items():
    d = d[2:] if len(d) > 4 else d
    res[n]["launch_us_avg"] = sum(d) / len(d); res[n]["launch_us_min"] = min(d); res[n]["launches"] = len(d)
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/g*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        n = name_of(r['Kernel_Name'])
        if n:
            cnt[n][r['Counter_Name']].append(float(r['Counter_Value']))
            for k_csv, k_out in (("VGPR_Count", "vgpr"), ("Accum_VGPR_Count", "accum_vgpr"), ("LDS_Block_Size", "lds_bytes")):
                if r.get(k_csv) not in (None, ""):
                    res[n].setdefault(k_out, r[k_csv])
for n, d in cnt.items():
    c = {k: sum(v[1:]) / len(v[1:]) if len(v) > 1 else v[0] for k, v in d.items()}
    res[n]["counters_per_launch"] = c
    if "SQ_WAVE_CYCLES" in c and c.get("SQ_WAVE_CYCLES"):
        w = c["SQ_WAVE_CYCLES"]
        res[n]["derived"] = {
            "mfma_busy_cycles_per_simd": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024.0,
            "kernel_core_clock_cycles": c.get("GRBM_GUI_ACTIVE"),
            "mfma_pipe_busy_frac_of_kernel": (c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024.0) / c["GRBM_GUI_ACTIVE"] if c.get("GRBM_GUI_ACTIVE") else None,
            "core_clock_GHz": c["GRBM_GUI_ACTIVE"] / (res[n]["launch_us_avg"] * 1e3) if c.get("GRBM_GUI_ACTIVE") and res[n].get("launch_us_avg") else None,
            "wait_any_frac_of_wave_cycles": c.get("SQ_WAIT_ANY", 0) / w,
            "wait_inst_any_frac": c.get("SQ_WAIT_INST_ANY", 0) / w,
            "wait_inst_lds_frac": c.get("SQ_WAIT_INST_LDS", 0) / w,
            "active_inst_any_frac": c.get("SQ_ACTIVE_INST_ANY", 0) / w,
            "lds_bank_conflict_frac_of_lds_active": c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"] if c.get("SQ_LDS_IDX_ACTIVE") else None,
            "hbm_side_bytes": (2.0 * c.get("FETCH_SIZE", 0) + c.get("WRITE_SIZE", 0)) * 1024.0 if "FETCH_SIZE" in c else None,
            "l2_hit_rate": c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]) if c.get("TCC_HIT_sum") else None,
        }
print(json.dumps({"shape": "16384 x 512 x 512 fp32, v_mfma_f32_32x32x2_f32", "flop_per_launch": 2 * 16384 * 512 * 512,
                  "note": "SQ_* counters are sums over all SEs as rocprofv3 reports them; *_CYCLES of the SQ are in quad-cycles where the guide says so; "
                          "FETCH_SIZE doubled per the guide's gfx950 correction; separate --pmc passes, durations from a kernel-trace pass of the same workload",
                  "kernels": res}, indent=1))
PY
rm -rf $OUT/kt $OUT/g?/
