#!/usr/bin/env python3
"""Register / spill figures of the k_correct instances as the compiler reports them (hipcc -Rpass-analysis=kernel-resource-usage,
cross-compiled: no GPU needed) -> profiles/r6_k3_resources.json, which bench.py attaches to config.k_correct."""
import json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "rcorrector_amd", "csrc")
out = {}
for f in ("rc_correct.hip", "rc_correct_k23.hip", "rc_correct_k25.hip", "rc_correct_k31.hip"):
    p = subprocess.run(["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-Wno-unused-function",
                        "-Rpass-analysis=kernel-resource-usage", "--cuda-device-only", "-c", os.path.join(SRC, f), "-o", "/dev/null"],
                       capture_output=True, text=True)
    cur = None
    for ln in p.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", ln)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            cur = name if name.startswith("void k_correct") else None
            if cur:
                out[cur] = {}
            continue
        m = re.search(r"remark:\s+(TotalSGPRs|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\d+)", ln)
        if m and cur:
            out[cur][m.group(1)] = int(m.group(2))
json.dump(out, open(os.path.join(ROOT, "profiles", "r6_k3_resources.json"), "w"), indent=1)
for k, v in out.items():
    print(k, v)
