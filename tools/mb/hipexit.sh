#!/bin/bash
# how long a HIP process takes to go away after _exit(), as its parent sees it: nothing allocated, 1 GB, 8 GB of device memory
for gb in 0 1 8; do
  python3 - "$gb" <<'PY'
import subprocess, sys, time, os
exe = os.path.join(os.path.dirname(os.path.abspath(sys.argv[0])) if False else "tools/mb", "hipexit")
for rep in range(3):
    t0 = time.time()
    p = subprocess.run([exe, sys.argv[1]], stdout=subprocess.PIPE)
    t1 = time.time()
    t_exit = float(p.stdout.decode().split()[0])
    print("hipMalloc %s GB: process %.3f s, of which %.3f s between _exit() and the parent's wait" % (sys.argv[1], t1 - t0, t1 - t_exit))
PY
done
