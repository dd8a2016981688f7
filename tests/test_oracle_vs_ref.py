"""CPU (build container only): the oracle's CLI against the UNMODIFIED reference binary on the same
random inputs the GPU fuzz test uses -- this is what makes `oracle_cli` a valid stand-in for the
reference on the GPU box.  Skipped where oracle/_ref was not built (no /root/reference)."""
import os
import subprocess

import pytest

from test_gpu_fuzz import _random_case


@pytest.mark.parametrize("seed", list(range(900, 940)))
def test_oracle_cli_equals_reference_on_random_inputs(oracle, seed, tmp_path):
    if not os.path.exists(oracle.REF_BIN):
        pytest.skip("oracle/_ref not built here (no /root/reference)")
    d = str(tmp_path)
    args = _random_case(seed, d)
    outs = {}
    for name, binary, more in (("ref", oracle.REF_BIN, ["-t", "3"] if seed % 2 else []), ("cpu", oracle.CLI_BIN, ["-t", "2"])):
        od = os.path.join(d, name)
        os.makedirs(od)
        p = subprocess.run([binary] + args + ["-od", od] + more, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert p.returncode == 0, p.stderr.decode()
        outs[name] = (p.stderr, {f: open(os.path.join(od, f), "rb").read() for f in sorted(os.listdir(od))})
    assert outs["ref"][1].keys() == outs["cpu"][1].keys() and outs["ref"][1]
    for f in outs["ref"][1]:
        assert outs["ref"][1][f] == outs["cpu"][1][f], "%s differs (seed %d, args %s)" % (f, seed, args)
    assert outs["ref"][0] == outs["cpu"][0], "stderr differs (seed %d)" % seed


@pytest.mark.parametrize("seed", list(range(300, 330)))
def test_oracle_cli_equals_reference_on_io_quirks(oracle, seed, tmp_path):
    """Input quirks with defined behaviour in the reference (tests/io_quirks.py): the oracle's
    command-line front end must write the reference's bytes and stderr lines."""
    import io_quirks
    import subprocess
    if not os.path.exists(oracle.REF_BIN):
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    d = str(tmp_path)
    args = io_quirks.make_case(seed, d, modes=(0, 1, 2))
    res = []
    for name, binary in (("ref", oracle.REF_BIN), ("ora", oracle.CLI_BIN)):
        od = os.path.join(d, name)
        os.makedirs(od)
        p = subprocess.run([binary] + args + ["-od", od, "-verbose"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        res.append((p.returncode, p.stderr, {f: open(os.path.join(od, f), "rb").read() for f in sorted(os.listdir(od))}, p.stdout))
    assert res[0][0] == 0 and res[0] == res[1]


@pytest.mark.parametrize("seed", list(range(400, 412)))
def test_oracle_dump_loader_equals_reference_on_quirky_dumps(oracle, seed, tmp_path):
    """tests/io_quirks.py: make_quirky_dump -- 'Stored N kmers', the ERROR_RATE estimate and the
    corrected reads must be the reference's when the dump departs from the clean jellyfish layout."""
    import io_quirks
    import subprocess
    import synth
    if not os.path.exists(oracle.REF_BIN):
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    d = str(tmp_path)
    io_quirks.make_quirky_dump(seed, os.path.join(d, "d.jf"), mid_n=seed % 2 == 0)
    s1, q1, _, _, _ = synth.make_reads(seed, 30, 60, n_tx=2, l_tx=200, e=0.01)
    synth.write_fastq(os.path.join(d, "a.fq"), s1, q1)
    res = []
    for name, binary in (("ref", oracle.REF_BIN), ("ora", oracle.CLI_BIN)):
        od = os.path.join(d, name)
        os.makedirs(od)
        p = subprocess.run([binary, "-r", "a.fq", "-k", "23", "-c", "d.jf", "-od", od, "-wk", "0.5"], cwd=d,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        res.append((p.returncode, p.stderr, open(os.path.join(od, "a.cor.fq"), "rb").read()))
    assert res[0][0] == 0 and res[0] == res[1]
