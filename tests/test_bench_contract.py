"""bench.py's contract with the driver, the parts that can be checked without a GPU:
 * `--gpus N` without a launcher never prints a line for another number of GPUs: it re-launches itself under
   torch.distributed.run when N devices are visible and stops with a non-zero status when fewer are (round-5 review:
   a bare `python bench.py --gpus 8` used to measure one GPU and print "n_gpus": 1);
 * the transfer-inclusive figures sit where the driver's record keeps them: BENCH_rNN.json holds the
   first 20 scalar entries of `config` (observed in BENCH_r05.json: it ended at `mode`, in front of `value_pcie`).
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_n_without_a_launcher_and_without_n_devices_fails_loudly():
    import torch
    have = torch.cuda.device_count()
    if have >= 2:
        pytest.skip("%d GPUs visible: the request can be met" % have)
    env = {kk: v for kk, v in os.environ.items() if kk not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "RC_BENCH_SHARED_GPU")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")], "no bench line may be printed"
    msg = p.stderr.decode()
    assert "--gpus 2 needs 2 GPUs, %d visible" % have in msg, msg[-1000:]


def test_world_size_that_differs_from_gpus_is_refused():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode != 0 and b"WORLD_SIZE (1) != --gpus (8)" in p.stderr
    assert not [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]


def test_transfer_inclusive_figures_survive_the_drivers_truncation():
    sys.path.insert(0, ROOT)
    import bench as B
    # a config assembled in any order -- here: the order of the newest full line under profiles/ when there is one, else the key
    # list of the bench's own line with the transfer figures LAST (the round-5 order) -- must come out with them in front
    lines = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_bench_config2.json"))
    cfg = None
    for f in reversed(lines):
        try:
            cfg = json.loads(open(os.path.join(ROOT, "profiles", f)).readline())["config"]
            break
        except (ValueError, KeyError):
            continue
    assert cfg is not None, "profiles/ holds no full bench line of the headline preset"
    for kk in ("value_pcie", "value_pcie_frac_of_value", "value_pcie_at_cli_default_batch"):
        cfg.setdefault(kk, 1.0)
    scrambled = dict(reversed(list(cfg.items())))
    kept = B.driver_kept_config(B.order_config(scrambled))
    assert len(kept) == 20 and list(kept)[0] == "workload"
    assert list(kept)[1:4] == ["value_pcie", "value_pcie_frac_of_value", "value_pcie_at_cli_default_batch"]
    for kk in ("value_pcie_at_cli_default_batch_frac_of_value", "preset", "reads_per_gpu", "read_len", "k", "table_kmers", "parallelism"):
        assert kk in kept, kk
    # and the round-5 order really did lose them (the truncation is modelled correctly)
    r5 = json.load(open(os.path.join(ROOT, "BENCH_r05.json"))) if os.path.exists(os.path.join(ROOT, "BENCH_r05.json")) else None
    if r5:
        full = json.loads(open(os.path.join(ROOT, "profiles", "r5_bench_config2.json")).readline())["config"]
        assert list(B.driver_kept_config(full)) == list(r5["parsed"]["config"])
