"""The reads the GPU path finishes before the general search (threshold kernel: clean on the real counts; k_single:
isolated substitutions, rc_single.h conditions (0)-(6)) are finished RIGHT: tests/k2s_model.py restates those conditions
over the oracle's primitives, and for every read the model accepts, the oracle's ErrorCorrection + GetKmerInformation
must give the model's answer -- return value, corrected bases, l / m / h.  (The kernels themselves meet the oracle in
the GPU suite; this pins the argument they rest on, on the CPU.)"""
import numpy as np
import pytest

import datasets
import k2s_model as M


def _reads(d, want, oracle):
    if d["mode"] == 1:
        seqs = d["seqs1"] + d["seqs2"]
        n1 = len(d["seqs1"])
        mate = lambda i: i + n1 if i < n1 else i - n1   # noqa: E731
        out = oracle.unpack_reads(want[4], oracle.pack_reads(d["seqs1"])[1]) + oracle.unpack_reads(want[5], oracle.pack_reads(d["seqs2"])[1])
    else:
        seqs = d["seqs1"]
        mate = (lambda i: i ^ 1) if d["mode"] == 2 else None
        out = oracle.unpack_reads(want[4], oracle.pack_reads(d["seqs1"])[1])
    return seqs, mate, out


@pytest.mark.parametrize("double", [False, True], ids=["as_built", "with_class_D"])
@pytest.mark.parametrize("name", ["se_k23", "pe_k23", "il_k23", "skew", "k11", "k32", "varlen", "polya_k23", "pe_151", "pe_160_k15", "se_151", "k31_mc8", "nrich", "pe_var", "k15"])
def test_model_of_the_early_finish_agrees_with_the_oracle(oracle, name, double):
    d = datasets.make(name)
    k, mfk = d["k"], d["mfk"]
    T = oracle.Table(k, len(d["keys"]))
    T.put_many(d["keys"], d["counts"])
    P = oracle.make_params(k, mfk, d["rate"], b"H")
    want = datasets.run_oracle(oracle, d)
    ret, l, m, h = want[:4]
    seqs, mate, out = _reads(d, want, oracle)
    strong, info = M.front_end(P, T, seqs, k)
    accepted = changed = 0
    for i, s in enumerate(seqs):
        pt = -1 if mate is None else int(min(strong[i], strong[mate(i)]))
        r = M.finished_early(P, T, s, k, mfk, int(strong[i]), int(info[i]), pt, allow_double=double)
        if r is None:
            continue
        accepted += 1
        changed += r[0] > 0
        assert r == (int(ret[i]), out[i], int(l[i]), int(m[i]), int(h[i])), "read %d of %s" % (i, name)
    if name not in ("k31_mc8", "nrich", "pe_var"):   # (5 % errors, N-rich, 3 % errors: few reads finish early there)
        assert accepted > 0.05 * len(seqs), (name, accepted)     # the model is not vacuous on any of these sets
    if name in ("skew", "k11", "pe_k23"):
        assert changed > 0.02 * len(seqs), (name, changed)
