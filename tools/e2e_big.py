#!/usr/bin/env python3
"""Files -> files at the north star's size (round 6): the `rcorrector` binary on up to 200 M synthetic reads (the headline
preset's generator: 150 bp pairs, k = 23), so that the run's fixed costs (exec + HIP, table, exit: ~0.7 s) amortise and the
host side's ceiling for the 8-GPU target is known before the node is.

Writes x_1.fq / x_2.fq arena by arena (12.5 M pairs each) under --dir, then times
    rcorrector -p x_1.fq x_2.fq -k 23 -od out            (one GPU: counts the k-mers itself, one pass over the files)
    RC_SHARED_GPU=1 rcorrector ... -gpus 8               (eight contexts on this box's one device: the -gpus N host path)
each `--repeat` times, and prints one JSON object.  Scales the input down to what the work directory's file system and the
host's memory hold (inputs + outputs + the text kept in memory by the one-pass run).  Dev / measurement tool; GPU box only.
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import synth_int  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=100_000_000)
    ap.add_argument("--dir", default="/tmp/rc_e2e_big")
    ap.add_argument("--repeat", type=int, default=2)
    ap.add_argument("--gpus", type=str, default="1,8", help="comma list of -gpus values (N > 1 runs with RC_SHARED_GPU=1)")
    a = ap.parse_args()
    L, k = 150, 23
    rec_bytes = 2 + 9 + 1 + L + 1 + 2 + L + 1
    shutil.rmtree(a.dir, ignore_errors=True)
    os.makedirs(a.dir, exist_ok=True)
    st = os.statvfs(a.dir)
    free = st.f_bavail * st.f_frsize
    mem = os.sysconf("SC_PHYS_PAGES") * os.sysconf("SC_PAGE_SIZE")
    reads = a.reads // 2 * 2
    # inputs + outputs on the file system (and, page cache to page cache, in memory), the text once more in the process
    while reads > 2_000_000 and (2.2 * reads * rec_bytes > 0.8 * free or 3.5 * reads * rec_bytes > 0.6 * mem):
        reads = reads // 2 // 2 * 2
    pairs = reads // 2
    dev = torch.device("cuda", 0)
    gen = synth_int.Synth(1002, L, 30000, 1500, 0.8, 0.005, True, device=dev)   # the headline preset's data (bench.py PRESETS[2])
    t0 = time.time()
    f1, f2 = open(os.path.join(a.dir, "x_1.fq"), "wb"), open(os.path.join(a.dir, "x_2.fq"), "wb")
    CH = 6_250_000
    for lo in range(0, pairs, CH):
        m = min(CH, pairs - lo)
        s0, q0 = gen.generate(lo, m)
        S, Q = s0.view(2 * m, L + 1)[:, :L].cpu().numpy(), q0.view(2 * m, L + 1)[:, :L].cpu().numpy()
        ids = np.char.zfill(np.arange(lo, lo + m).astype(str), 9)
        idb = np.frombuffer("".join(ids.tolist()).encode(), dtype=np.uint8).reshape(m, 9)
        for f, half in ((f1, 0), (f2, 1)):
            rec = np.empty((m, rec_bytes), dtype=np.uint8)
            rec[:, 0] = ord('@'); rec[:, 1] = ord('r'); c = 2
            rec[:, c:c + 9] = idb; c += 9
            rec[:, c] = 10; c += 1
            rec[:, c:c + L] = S[half * m:(half + 1) * m]; c += L
            rec[:, c] = 10; c += 1
            rec[:, c] = ord('+'); rec[:, c + 1] = 10; c += 2
            rec[:, c:c + L] = Q[half * m:(half + 1) * m]; c += L
            rec[:, c] = 10
            rec.tofile(f)
        del s0, q0
    f1.close(); f2.close()
    t_gen = time.time() - t0
    del gen
    torch.cuda.empty_cache()
    cli = os.path.join(ROOT, "rcorrector_amd", "rcorrector")
    out = {"reads": reads, "asked": a.reads, "read_len": L, "k": k, "input_bytes": 2 * pairs * rec_bytes, "generate_inputs_s": round(t_gen, 1),
           "work_dir_free_bytes": free, "host_memory_bytes": mem, "host_cores": os.cpu_count(), "runs": []}
    for g in [int(x) for x in a.gpus.split(",")]:
        for rep in range(a.repeat):
            od = os.path.join(a.dir, "out")
            shutil.rmtree(od, ignore_errors=True)
            os.sync()
            env = dict(os.environ, RC_TIMING="1")
            if g > 1:
                env["RC_SHARED_GPU"] = "1"
            t0 = time.time()
            env["RC_T0"] = repr(t0)
            p = subprocess.run([cli, "-p", "x_1.fq", "x_2.fq", "-k", str(k), "-od", od] + (["-gpus", str(g)] if g > 1 else []),
                               cwd=a.dir, env=env, stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, timeout=1800)
            wall = time.time() - t0
            err = p.stderr.decode()
            loop_s = startup_s = None
            for ln in err.splitlines():
                if ln.startswith("[rc timing]") and "correction loop" in ln:
                    startup_s = float(ln.split("bad quality) ")[1].split(" s")[0])
                    loop_s = float(ln.split("(read, correct, write) ")[1].split(" s")[0])
            ob = sum(os.path.getsize(os.path.join(od, f)) for f in os.listdir(od)) if os.path.isdir(od) else 0
            out["runs"].append({"gpus": g, "shared_gpu": g > 1, "rc": p.returncode, "process_wall_s": round(wall, 3),
                                "reads_per_s_whole_process": reads / wall, "loop_s": loop_s, "loop_reads_per_s": reads / loop_s if loop_s else None,
                                "startup_s": startup_s, "output_bytes": ob,
                                "rc_timing": [ln for ln in err.splitlines() if ln.startswith("[rc timing]") and not ln.startswith("[rc timing] +")][-6:]
                                if p.returncode == 0 else err[-600:]})
            shutil.rmtree(od, ignore_errors=True)
    shutil.rmtree(a.dir, ignore_errors=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
