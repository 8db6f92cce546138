"""bench.py's ONE line (the driver's contract): strict JSON, shorter than 4 KB whatever the run measured, and the
`--gpus N` relaunch under torch.distributed.run -- on CPU through `--dry-run` (which measures nothing and says so)."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def _strict(text):
    def refuse(name):
        raise ValueError("non-strict JSON constant " + name)
    return json.loads(text, parse_constant=refuse)


def _stub():
    long = "x" * 5000
    roof = {k: 1.0 / 3.0 for k in bench.ROOF_KEYS}
    roof.update(kernel="k_ldl_front<" + long + ">", bound="mfma", unit="TFLOP/s", traffic=43010377, stage_ms_per_step={long[:50] + str(i): 0.1 for i in range(200)},
                timing=long)
    return {"metric": "IPM iters/sec (ADA' form+factor+solve)", "value": np.float64(2594.4962577111796), "unit": "IPM iters/s", "n_gpus": 1, "steps": 20,
            "warmup": 5, "ms_per_step": 0.38543127477169037, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": long,
            "config": {"workload": long, "parallelism": long, "more": long}, "roofline": roof,
            "cpu_baseline": {"value": 18.99, "unit": "IPM iters/s", "cores": 1, "kind": "reference", "sample": long, "stage_ms_per_unit": {"a": float("nan")}},
            "other_configs": [{"note": long, "v": float("inf")}] * 50, "phases_ms_per_step": {"solve": {"frac": float("nan")}}}


def test_line_is_short_strict_json_with_the_contract_fields():
    line = bench.compact_line(_stub(), {"speedup_vs_cpu_reference": float("nan"), "mex_inclusive_value": 293.8, "c": 1, "d": 2, "e": 3, "dropped": 4},
                              "profiles/bench_detail_x.json")
    assert "\n" not in line and len(line.encode()) < 4096
    d = _strict(line)
    for k in bench.TOP_KEYS + ("config", "roofline", "cpu_baseline", "detail"):
        assert k in d
    assert set(d["roofline"]) == set(bench.ROOF_KEYS) and set(d["cpu_baseline"]) == set(bench.BASE_KEYS)
    assert set(d["config"]) == {"workload", "parallelism"}
    assert d["value"] == 2594.5 and d["roofline"]["traffic"] == 43010377 and d["roofline"]["bound"] == "mfma"
    assert d["speedup_vs_cpu_reference"] is None and "dropped" not in d         # NaN -> null; five extras at most
    assert "other_configs" not in d and "phases_ms_per_step" not in d


def test_detail_file_is_strict_json(tmp_path, monkeypatch):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    rel = bench.write_detail(_stub(), "a tag/with:odd chars")
    assert rel == os.path.join("profiles", "bench_detail_a_tag_with_odd_chars.json")
    d = _strict(open(tmp_path / rel).read())
    assert len(d["other_configs"]) == 50 and d["other_configs"][0]["v"] is None


def _run(argv, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return _strict(lines[0])


def test_dry_run_single_process():
    d = _run(["--dry-run", "--steps", "3", "--warmup", "1"])
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["value"] is None and d["dry_run"] is True


def test_gpus_2_without_a_launcher_relaunches_under_torch_distributed_run():
    """`python bench.py --gpus 2` (no WORLD_SIZE around it): two ranks come up on 127.0.0.1, rank 0 alone prints, n_gpus = 2."""
    d = _run(["--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1"])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"].startswith("2 rank")
