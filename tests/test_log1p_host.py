"""The engine's correctly-rounded -log1p (blackjax_amd/csrc/bjx_log1p.h: table-driven argument reduction, no division,
exactly rounded IEEE operations only) against the contract ``(float)(-log1p((double)t))`` (oracle/fp.py::log1p_cr), on
the host.  The same source runs on the device inside ErfInv32 (jax.random.normal, blackjax/util.py:88-91); the device
repeats the exhaustive comparison against its own library (tests/test_hmc_gpu.py::test_device_log1p_exhaustive).

EXHAUSTIVE: all 1 065 353 217 fp32 inputs of (-1, 0], about 15 s on 8 cores: 0 mismatches, 402 inputs deferred to the
slow table (every one of them in the table with the contract's value), max relative error of the fp64 value 2^-50.5."""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "blackjax_amd", "csrc", "bjx_log1p_table.h")


def test_log1p_matches_contract_for_every_fp32_input():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle", "c"), "check_log1p"], check=True,
                   capture_output=True)
    stride = os.environ.get("BJX_LOG1P_STRIDE", "1")
    out = subprocess.run([os.path.join(ROOT, "oracle", "c", "check_log1p"), stride], capture_output=True,
                         text=True)
    rep = json.loads(out.stdout)
    assert out.returncode == 0, rep
    assert rep["mismatch"] == 0 and rep["deferred_missing_from_table"] == 0
    assert rep["max_rel_err_log2"] < -49
    if stride == "1":
        assert rep["checked"] == 1065353217 and rep["deferred"] == rep["slow_table_entries"]
    assert rep["deferred"] < rep["checked"] * 1e-5


def test_committed_tables_are_what_the_generator_writes():
    """The argument-reduction table is regenerated (50-digit arithmetic) and compared with the committed header; the slow
    table is sorted by input bit pattern (the device bisects it) and holds negative inputs of (-1, 0] only."""
    text = open(HDR).read()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "c", "gen_log1p_table.py"), "--reduction-only"],
                         capture_output=True, text=True, check=True)
    regenerated = [tuple(ln.split()) for ln in out.stdout.strip().splitlines()]
    committed = re.findall(r"^\s+\{(-?0x[0-9a-fp+.-]+), (-?0x[0-9a-fp+.-]+)\},\s+// \d+$", text, re.M)
    assert len(regenerated) == 129 and committed == regenerated
    assert committed[74] == committed[75] == ("0x1.0000000000000p+0", "0x0.0p+0")  # r = m - 1 exactly next to m = 1
    slow = [(int(a, 16), int(b, 16)) for a, b in re.findall(r"\{0x([0-9A-F]{8})u, 0x([0-9A-F]{8})u\}", text)]
    n_decl = int(re.search(r"#define BJX_L1P_N_SLOW (\d+)", text).group(1))
    assert len(slow) == n_decl and n_decl > 0
    keys = [a for a, _ in slow]
    assert keys == sorted(set(keys)) and all(0x80000000 <= k < 0xBF800000 for k in keys)
