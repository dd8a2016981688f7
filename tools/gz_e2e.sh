#!/bin/bash
# dev: files-to-files on .gz inputs (8 M x 150 bp pairs): two passes vs one
cd "$(dirname "$0")/.."
OUT=gpurun_out/gz; mkdir -p $OUT; export TMPDIR=/tmp
python - > $OUT/gz.txt 2>&1 <<'P'
import os, sys, time, subprocess, hashlib, gzip
ROOT=os.getcwd(); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+"/tools")
import numpy as np, torch, bench
d="/tmp/rc_gz"; os.makedirs(d, exist_ok=True)
n, L = 8_000_000, 150
seq, qual = bench.synth_reads_gpu(77000, n, L, 30000, 1500, 0.8, 0.005, torch.device("cuda",0), paired=True)
S = seq.view(n, L+1)[:, :L].cpu().numpy(); Q = qual.view(n, L+1)[:, :L].cpu().numpy()
def write_fq(path, s, q):
    m=len(s); ids=np.char.zfill(np.arange(m).astype(str), 9)
    idb=np.frombuffer("".join(ids.tolist()).encode(), dtype=np.uint8).reshape(m, 9)
    rec=np.empty((m, 2+9+1+L+1+2+L+1), dtype=np.uint8)
    rec[:,0]=ord('@'); rec[:,1]=ord('r'); c=2
    rec[:,c:c+9]=idb; c+=9; rec[:,c]=10; c+=1
    rec[:,c:c+L]=s; c+=L; rec[:,c]=10; c+=1
    rec[:,c]=ord('+'); rec[:,c+1]=10; c+=2
    rec[:,c:c+L]=q; c+=L; rec[:,c]=10
    rec.tofile(path)
write_fq(d+"/x_1.fq", S[:n//2], Q[:n//2]); write_fq(d+"/x_2.fq", S[n//2:], Q[n//2:])
t0=time.time()
ps=[subprocess.Popen(["gzip","-1","-k","-f",d+"/x_%d.fq"%i]) for i in (1,2)]
[p.wait() for p in ps]
print("gzip -1: %.1f s; %.0f MB each -> %.0f MB" % (time.time()-t0, os.path.getsize(d+"/x_1.fq")/1e6, os.path.getsize(d+"/x_1.fq.gz")/1e6), flush=True)
cli=ROOT+"/rcorrector_amd/rcorrector"
def run(tag, inputs, env):
    od=d+"/out_"+tag; subprocess.run(["rm","-rf",od]); os.sync()
    t0=time.time()
    p=subprocess.run([cli]+inputs+["-k","23","-od",od], cwd=d, env=dict(os.environ, RC_TIMING="1", RC_T0=repr(t0), **env), stderr=subprocess.PIPE)
    dt=time.time()-t0
    outs=sorted(os.listdir(od))
    f=od+"/"+outs[0]
    data=gzip.open(f,"rb").read() if f.endswith(".gz") else open(f,"rb").read()
    print("%-22s wall %.2f s -> %.2f M reads/s; %s md5 %s" % (tag, dt, n/dt/1e6, outs, hashlib.md5(data).hexdigest()), flush=True)
    for ln in p.stderr.decode().splitlines():
        if "counting pass" in ln or "correction loop" in ln or "stage totals" in ln: print("    "+ln)
run("plain_one", ["-p","x_1.fq","x_2.fq"], {})
run("gz_two", ["-p","x_1.fq.gz","x_2.fq.gz"], {"RC_RESIDENT":"0"})
run("gz_one", ["-p","x_1.fq.gz","x_2.fq.gz"], {})
P
cat $OUT/gz.txt; rm -rf /tmp/rc_gz
