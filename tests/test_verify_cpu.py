"""CPU: the `verify` accuracy scorer (rcorrector_amd/csrc/rc_verify.cpp) prints exactly what the
reference's scorer prints (verify.cpp:131-483) -- on raw simulated reads, on the same reads
after correction, on trimmed / indel variants (the longest-common-subsequence alignment path), for
every option.  Compared against the committed outputs of the reference binary
(tests/golden/verify/, written by tests/golden/make_golden.py) and, when oracle/_ref/verify_ref is
built, against that binary directly on further seeds."""
import os
import subprocess

import pytest

import datasets
import golden_util as gu
import synth

VERIFY = os.path.join(gu.ROOT, "rcorrector_amd", "verify")
REF_VERIFY = os.path.join(gu.ROOT, "oracle", "_ref", "verify_ref")
GOLD = os.path.join(gu.GOLDEN, "verify")
OPTION_SETS = {"plain": [], "v": ["-v"], "bv": ["-bv"], "exp": ["-exp"], "noindel": ["-noindel"], "all": ["-v", "-bv", "-exp", "-noindel"]}


@pytest.fixture(scope="module")
def verify_bin():
    subprocess.run(["make", "-C", os.path.join(gu.ROOT, "rcorrector_amd", "csrc"), "../verify"], check=True,
                   stdout=subprocess.DEVNULL)
    return VERIFY


def build_cases(d, oracle_cli, seed=11):
    """raw.fq (uncorrected), cor.fq (corrected by the CPU oracle CLI; the header line gains the
    ' l:.. m:.. h:.. cor' fields), indel.fq (trimmed / indel variants).  Deterministic."""
    heads, reads, quals = datasets.mason_style_reads(seed=seed)
    datasets.write_mason_fastq(os.path.join(d, "raw.fq"), heads, reads, quals)
    datasets.write_mason_fastq(os.path.join(d, "indel.fq"), *datasets.mason_indel_variants(heads, reads, quals))
    import numpy as np
    arr = np.frombuffer(b"".join(reads), dtype=np.uint8).reshape(len(reads), -1)
    keys, cnt = synth.count_kmers([arr], 23)
    synth.write_dump(os.path.join(d, "dump.jf"), keys, cnt, 23)
    subprocess.run([oracle_cli, "-r", "raw.fq", "-k", "23", "-c", "dump.jf", "-od", d], cwd=d, check=True,
                   stderr=subprocess.DEVNULL)
    os.replace(os.path.join(d, "raw.cor.fq"), os.path.join(d, "cor.fq"))
    return ["raw.fq", "cor.fq", "indel.fq"]


def run(binary, path, opts):
    p = subprocess.run([binary, path] + opts, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()
    return p.stdout


def test_verify_equals_committed_reference_outputs(verify_bin):
    files = sorted(f for f in os.listdir(GOLD) if f.endswith(".fq"))
    assert files
    for f in files:
        for name, opts in OPTION_SETS.items():
            want = open(os.path.join(GOLD, "%s.%s.txt" % (f[:-3], name)), "rb").read()
            assert run(verify_bin, os.path.join(GOLD, f), opts) == want, (f, name)


def test_golden_inputs_are_reproducible(oracle_cli, tmp_path):
    # the committed inputs are exactly what build_cases() makes (so the goldens can be regenerated)
    for f in build_cases(str(tmp_path), oracle_cli):
        assert open(tmp_path / f, "rb").read() == open(os.path.join(GOLD, f), "rb").read(), f


@pytest.mark.skipif(not os.path.exists(REF_VERIFY), reason="oracle/_ref/verify_ref not built (needs /root/reference)")
@pytest.mark.parametrize("seed", [3, 4, 5, 6])
def test_verify_equals_reference_binary_on_more_seeds(verify_bin, oracle_cli, seed, tmp_path):
    for f in build_cases(str(tmp_path), oracle_cli, seed=seed):
        for name, opts in OPTION_SETS.items():
            assert run(verify_bin, str(tmp_path / f), opts) == run(REF_VERIFY, str(tmp_path / f), opts), (f, name)


def test_corrected_reads_score_better_than_raw(verify_bin):
    """Sanity of the numbers themselves: the corrector turns most erroneous reads into true
    positives and introduces few false positives."""
    def field(out, section, key):
        sec = out.split(section.encode())[1]
        return int(sec.split(key.encode() + b": ")[1].split(b"\n")[0])
    raw = run(verify_bin, os.path.join(GOLD, "raw.fq"), [])
    cor = run(verify_bin, os.path.join(GOLD, "cor.fq"), [])
    assert field(raw, "Base level:", "TP") == 0 and field(raw, "Base level:", "FN") > 300
    assert field(cor, "Base level:", "TP") > 0.8 * field(raw, "Base level:", "FN")
    assert field(cor, "Base level:", "FP") < 0.05 * field(cor, "Base level:", "TP")
