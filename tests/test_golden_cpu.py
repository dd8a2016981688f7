"""CPU: the oracle (oracle_cli) must reproduce, byte for byte, what the unmodified reference
binary wrote for every golden fixture -- *.cor.fq files and the stderr parameter lines.  This is
what pins the oracle (and is re-checked against a fresh reference run where /root/reference
exists)."""
import hashlib
import os
import subprocess

import pytest

import golden_util as gu


@pytest.fixture(scope="module")
def oracle_cli(oracle):
    assert os.path.exists(oracle.CLI_BIN)
    return oracle.CLI_BIN


@pytest.mark.parametrize("name", gu.FIXTURES)
def test_oracle_reproduces_reference_outputs(oracle_cli, name, tmp_path):
    p = gu.run_fixture(oracle_cli, name, tmp_path, extra=["-t", "2"])
    gu.assert_same_as_reference(name, tmp_path, p.stderr)


def test_sample_pins_from_survey():
    # BASELINE.md §2 / SURVEY.md §8(c): md5 of the reference's output on its own Sample pair
    ref = os.path.join(gu.GOLDEN, "fx_sample", "ref")
    md5 = lambda f: hashlib.md5(open(os.path.join(ref, f), "rb").read()).hexdigest()
    assert md5("sample_read1.cor.fq") == "d67b929e0ce992bddd34525c78df8fe5"
    assert md5("sample_read2.cor.fq") == "40724361c10ba362ea15d1ae0b6b59dc"
    err = open(os.path.join(ref, "stderr.txt")).read()
    assert "Stored 253 kmers" in err and "Processed 70 reads" in err and "Corrected 21 bases" in err


def test_interleaved_equals_paired_golden():
    # -i and -p give the same records
    a = open(os.path.join(gu.GOLDEN, "fx_pe_k23", "ref", "reads_1.cor.fq"), "rb").read().split(b"\n")
    b = open(os.path.join(gu.GOLDEN, "fx_pe_k23", "ref", "reads_2.cor.fq"), "rb").read().split(b"\n")
    il = open(os.path.join(gu.GOLDEN, "fx_il_k23", "ref", "reads_il.cor.fq"), "rb").read().split(b"\n")
    recs = lambda x: [tuple(x[i:i + 4]) for i in range(0, len(x) - 1, 4)]
    ra, rb, ri = recs(a), recs(b), recs(il)
    assert ri[0::2] == ra and ri[1::2] == rb


@pytest.mark.parametrize("name", gu.FIXTURES)
def test_committed_goldens_are_what_the_reference_writes(oracle, name, tmp_path):
    if not os.path.exists(oracle.REF_BIN):
        pytest.skip("oracle/_ref not built here (no /root/reference): goldens were generated in the build container")
    p = gu.run_fixture(oracle.REF_BIN, name, tmp_path)
    gu.assert_same_as_reference(name, tmp_path, p.stderr)


def test_oracle_threads_and_stdout(oracle_cli, tmp_path):
    # -t 1 == -t 8, and -stdout prints the same records (pairs alternate, main.cpp:487-495)
    p1 = gu.run_fixture(oracle_cli, "fx_pe_k23", tmp_path, extra=["-t", "1"])
    gu.assert_same_as_reference("fx_pe_k23", tmp_path, p1.stderr)
    p = gu.run_fixture(oracle_cli, "fx_pe_k23", tmp_path / "x", extra=["-stdout", "-t", "4"])
    ref = os.path.join(gu.GOLDEN, "fx_pe_k23", "ref")
    a = open(os.path.join(ref, "reads_1.cor.fq"), "rb").read().split(b"\n")
    b = open(os.path.join(ref, "reads_2.cor.fq"), "rb").read().split(b"\n")
    want = []
    for i in range(0, len(a) - 1, 4):
        want += a[i:i + 4] + b[i:i + 4]
    assert p.stdout.split(b"\n")[:-1] == want


@pytest.mark.parametrize("name", gu.FIXTURES)
def test_oracle_verbose_transcript_matches_reference(oracle_cli, name, tmp_path):
    """-verbose exposes the intermediate state of every read (original counts, each
    (strongTrustThreshold, trustThreshold) iteration incl. the INT_MIN ones, the strong-trusted
    bitmap, post-correction counts); the oracle must print the reference's transcript byte for byte."""
    import gzip
    want = gzip.open(os.path.join(gu.GOLDEN, name, "verbose.txt.gz"), "rb").read()
    p = gu.run_fixture(oracle_cli, name, tmp_path, extra=["-verbose"])
    assert p.stdout == want
    assert want.count(b"strong trust threshold=") >= want.count(b"Before correction:")


def test_fasta_fixture_oracle_equals_reference_batch_path(oracle_cli, oracle, tmp_path):
    """FASTA input (Reads.h:108-162): pinned to the reference's -t > 1 path (its -t 1 loop crashes on
    FASTA, see make_golden.py); the oracle must write the same bytes, and so must a fresh reference run."""
    p = gu.run_fixture(oracle_cli, "fa_se_k23", tmp_path)
    gu.assert_same_as_reference("fa_se_k23", tmp_path, p.stderr)
    if os.path.exists(oracle.REF_BIN):
        p = gu.run_fixture(oracle.REF_BIN, "fa_se_k23", tmp_path / "r")
        gu.assert_same_as_reference("fa_se_k23", tmp_path / "r", p.stderr)
