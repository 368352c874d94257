"""GPU: bench.py keeps its output contract -- one JSON line with the driver's fields, the roofline of the seed-pass
kernel, the CPU baseline (checked equal to the GPU results) and the untimed A/B through the plain operators -- on a
small custom workload (the default one needs a 3 Gbp index)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--no-wide-table"], ["--no-canonical"]])
def test_bench_json_contract(extra):
    """the default configuration (canonical table, one seed pass for both strands) and the per-strand one over the direct table"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--ref-len", "4e6", "--reads", "40000", "--steps", "2",
                          "--warmup", "1", "--kmer", "15", "--cpu-sample", "5000", "--robust-copies", "40"] + extra, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "reads/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["config"]["workload"] == "custom" and d["config"]["reads_per_gpu"] == 40000
    assert abs(d["value"] - 40000 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]
    canon = extra != ["--no-canonical"]
    assert d["config"]["canonical_table"] is canon and d["config"]["wide_entries"] is (not extra)
    assert d["roofline"]["launches_per_step"] == (1 if canon else 2) and ("match_both" in d["stage_ms"]) == canon
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    # a roofline fraction is a fraction: the bytes are those the TIMED launch moves (sector-granular, counted by the kernel)
    assert 0.0 < r["frac"] <= 1.0 and 0.0 < r["reference_algorithm"]["frac"] <= 1.0
    assert r["useful_bytes_per_launch"] <= r["bytes_per_launch"] and 0.0 < r["useful_frac"] <= r["frac"]
    assert abs(r["achieved"] - r["bytes_per_launch"] / (r["ms_per_launch"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert r["gathered_sectors_per_launch"] > 0 and 0.5 < r["sectors_per_seed"] < 40.0       # about 1.07 on the 3 Gbp / k = 17 workload; rank steps dominate on this toy index
    assert r["traffic"] is None or "profiles/" in r["traffic_source"]          # counter traffic comes from a named profile, never from thin air
    assert "i16" in d["dtype"] and d["extend"]["gcups"] > 0 and d["extend"]["effective_gcups"] >= d["extend"]["gcups"] * 0.5
    # the DP kernel's own bound: instruction issue (a fraction of it, from the row loop's instruction count)
    ib = d["extend"]["dp_issue_bound"]
    assert ib["instructions_per_row"] > 300 and ib["cells_per_wave_row"] == 64 * 2 * 31 and 0.0 < ib["frac"] <= 1.0
    assert d["traceback"]["cigars_truncated"] == 0
    assert d["left_out_of_the_step"]["build"]["index_and_tables_s"] > 0
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["results_equal_gpu"] is True and c["cores"] >= 1
    assert d["plain_operators"]["results_equal"] is True
    assert d["traceback"]["scores_equal_scoring_pass"] is True
    assert d["aligned_fraction"] > 0.99 and d["correct_locus_fraction"] > 0.98
    assert c["cores"] <= c["logical_cpus"] and (c["kind"] != "reference" or c["port"]["results_equal_reference"] is True)
    # BASELINE configs 2, 4, 5 at kernel / composition level, in the same line
    cf = d["configs"]
    assert "error" not in cf, cf
    assert cf["sw_benchmark_100k"]["gcups"] > 0 and cf["sw_benchmark_100k"]["scores_equal_oracle_sample"] is True
    assert cf["fm_seeds_1M"]["queries_per_s"] > 0 and 0.0 < cf["fm_seeds_1M"]["no_table"]["alg_frac_of_hbm_peak"] <= 1.0
    assert cf["banded_local_6.25M"]["gcups"] > 0 and 0.0 < cf["banded_local_6.25M"]["dp_issue_frac"] <= 1.0
    assert cf["paired_end_1M"]["pairs_per_s"] > 0 and cf["paired_end_1M"]["concordant_fraction"] > 0.9
    nm = d["nvbowtie_mode"]
    assert "error" not in nm, nm
    assert nm["ms_per_step"] > 0 and nm["n_extensions"] >= 40000 * 0.9 and nm["aligned_fraction"] > 0.97 and nm["best_score_equals_default_pipeline"] > 0.97
    sw = d["strong_scaling_sweep_1gpu"]
    assert [e["reads"] for e in sw] == [5000, 10000, 20000, 40000] and all(e["ms_per_step"] > 0 for e in sw)
    # the robust-input step (repeat family + per-base qualities + ragged reads) over the canonical table, checked against the plain operators
    if canon:
        ov = d["overlapped_step"]
        assert "error" not in ov, ov
        assert ov["ms_per_step"] > 0 and ov["results_equal"] is True
        ch = d["cpp_host"]
        assert "error" not in ch, ch
        assert ch["ms_per_step"] > 0 and ch["results_equal_python_step"] is True
        rb = d["robust"]
        assert "error" not in rb, rb
        assert rb["ms_per_step"] > 0 and rb["plain_operators"]["results_equal"] is True and rb["aligned_fraction"] > 0.97
    else:
        assert "robust" not in d
