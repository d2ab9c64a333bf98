"""The committed measurement evidence is self-consistent (CPU): profiles/pmc_traffic.json is what
tools/pmc_traffic.py derives from the committed rocprofv3 --pmc csv files, bench.py reads it, and the
last bench line under profiles/ honours the bench contract (keys, roofline and cpu_baseline objects)."""
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")


def test_pmc_traffic_json_is_derived_from_the_committed_counter_files(tmp_path):
    want = json.load(open(os.path.join(PROF, "pmc_traffic.json")))
    tag = want.get("counter_files", "r01l")     # the committed csv pair the json was derived from
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        shutil.copy(os.path.join(PROF, "%s_pmc_%s.csv" % (tag, c)), tmp_path / ("pmc_%s.csv" % c))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"), str(tmp_path)],
                         capture_output=True, text=True, check=True).stdout
    got = json.loads(out)
    for k in want:                              # (keys the tool added later are absent from an older json)
        if k not in ("build", "counter_files"):
            assert got[k] == want[k], k
    conv = want["conv_family_per_step"]
    assert conv["launches"] >= 13 and 1e8 < conv["hbm_bytes"] < 5e9
    # the gather moves about its algorithmic bytes (no wasted re-reads) ...
    g = want["gather_n262144"]
    assert 0.8 < g["hbm_bytes"] / g["algorithmic_bytes"] < 1.2
    # ... and the gather from the cache-resident taps far fewer than its algorithmic reads
    t = want["gather_from_taps_n2048"]
    assert t["hbm_bytes"] < 0.5 * t["algorithmic_bytes"]


def test_last_bench_line_honours_the_contract():
    files = sorted(glob.glob(os.path.join(PROF, "r0??_bench.json")))
    assert files
    line = json.loads(open(files[-1]).read().strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["n_gpus"] == 1 and line["higher_is_better"] is True and line["vs_baseline"] is None
    assert "workload" in line["config"] and "model" not in line["config"]
    assert abs(line["value"] - 2048 / line["ms_per_step"] * 1e3) / line["value"] < 1e-6
    r = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    c = line["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["value"] < line["value"]
