"""GPU: the drop-in `rcorrector` binary (C++ host over the C ABI over the HIP kernels) must write
the same bytes as the unmodified reference for every golden fixture: *.cor.fq, stderr parameter
lines, -stdout, gz in/out, any batch size."""
import gzip
import os
import shutil

import pytest

import golden_util as gu

pytestmark = pytest.mark.gpu
CLI = os.path.join(gu.ROOT, "rcorrector_amd", "rcorrector")


@pytest.mark.parametrize("name", gu.FIXTURES)
def test_cli_reproduces_reference_outputs(name, tmp_path):
    p = gu.run_fixture(CLI, name, tmp_path)
    gu.assert_same_as_reference(name, tmp_path, p.stderr)


@pytest.mark.parametrize("name", gu.FIXTURES + ["fa_se_k23"])
@pytest.mark.parametrize("extra", [[], ["-batch", "50"]], ids=["one_batch", "batch50"])
def test_cli_packed_transport_reproduces_reference_outputs(name, extra, tmp_path):
    """`-packed` (or RC_TRANSPORT=packed): the batches cross PCIe through rc_submit_packed -- 2-bit bases, quality bits,
    the letters outside ACGT as a list down, ret / l / m / h and the substitutions as a fix list up, applied to the arenas in
    front of the formatter -- and every golden fixture, the FASTA one included (no quality array), comes out as the reference
    wrote it; a FASTQ batch with an empty quality line takes the byte path (fx_io_quirks)."""
    p = gu.run_fixture(CLI, name, tmp_path, extra=["-packed"] + extra)
    gu.assert_same_as_reference(name, tmp_path, p.stderr)


@pytest.mark.parametrize("lanes", ["1", "0"])
@pytest.mark.parametrize("name", ["fx_se_k23", "fx_pe_k23", "fx_il_k23", "fx_edge"])
def test_cli_small_batches_and_thread_flag(name, lanes, tmp_path, monkeypatch):
    # output must not depend on the batch size (the reference's is 512*T, main.cpp:441), nor on whether the batches in flight
    # run in slot lanes (a context per slot, kernels of consecutive batches side by side) or one after the other in one context
    monkeypatch.setenv("RC_SLOT_LANES", lanes)
    p = gu.run_fixture(CLI, name, tmp_path, extra=["-batch", "50", "-t", "8", "-inflight", "3"])
    gu.assert_same_as_reference(name, tmp_path, p.stderr)


@pytest.mark.parametrize("name", gu.FIXTURES)
def test_cli_verbose_transcript_equals_reference(name, tmp_path):
    """-verbose: the reference's per-read transcript (original counts, every (strong, trust)
    iteration, the trusted-base bitmap, post-correction counts; ErrorCorrection.cpp:686-689,759-770,
    856-857,1088-1094,1590-1597) byte for byte, and the output files unchanged by it."""
    p = gu.run_fixture(CLI, name, tmp_path, extra=["-verbose", "-batch", "64"])
    want = gzip.open(os.path.join(gu.GOLDEN, name, "verbose.txt.gz"), "rb").read()
    assert p.stdout == want
    gu.assert_same_as_reference(name, tmp_path, p.stderr)


def test_cli_verbose_iteration_capacity_is_checked(tmp_path):
    # one read of fx_pe_k23 lowers its thresholds once (2 iterations): a capacity of 1 must fail
    # loudly rather than print a truncated transcript, 2 is enough
    import subprocess
    d = os.path.join(gu.GOLDEN, "fx_pe_k23")
    args = open(os.path.join(d, "cmd.txt")).read().split()
    want = gzip.open(os.path.join(d, "verbose.txt.gz"), "rb").read()
    assert want.count(b"strong trust threshold=") == want.count(b"Before correction:") + 1
    for cap, ok in ((1, False), (2, True)):
        p = subprocess.run([CLI] + args + ["-od", str(tmp_path), "-verbose", "-verbose-iter", str(cap)], cwd=d,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        if ok:
            assert p.returncode == 0 and p.stdout == want
        else:
            assert p.returncode != 0 and b"-verbose-iter" in p.stderr


def test_cli_stdout_pairs_alternate(tmp_path):
    p = gu.run_fixture(CLI, "fx_pe_k23", tmp_path, extra=["-stdout"])
    ref = os.path.join(gu.GOLDEN, "fx_pe_k23", "ref")
    a = open(os.path.join(ref, "reads_1.cor.fq"), "rb").read().split(b"\n")
    b = open(os.path.join(ref, "reads_2.cor.fq"), "rb").read().split(b"\n")
    want = []
    for i in range(0, len(a) - 1, 4):
        want += a[i:i + 4] + b[i:i + 4]
    assert p.stdout.split(b"\n")[:-1] == want


def test_cli_gz_in_gz_out(tmp_path):
    src = os.path.join(gu.GOLDEN, "fx_se_k23")
    work = tmp_path / "in"
    work.mkdir()
    with open(os.path.join(src, "reads.fq"), "rb") as f, gzip.open(work / "reads.fq.gz", "wb") as g:
        shutil.copyfileobj(f, g)
    shutil.copy(os.path.join(src, "dump.jf"), work / "dump.jf")
    out = tmp_path / "out"
    p = gu.run_fixture(CLI, "fx_se_k23", out, args_override=["-r", str(work / "reads.fq.gz"), "-k", "23", "-c", str(work / "dump.jf")])
    got = gzip.open(out / "reads.cor.fq.gz", "rb").read()      # name rule Reads.h:39-75,140-147
    assert got == open(os.path.join(src, "ref", "reads.cor.fq"), "rb").read()
    assert p.stderr == open(os.path.join(src, "ref", "stderr.txt"), "rb").read()


def test_cli_multiple_files_in_one_run(tmp_path):
    # -r a -r b: batches never span files (Reads.h:332-358); same records as separate runs
    a = os.path.join(gu.GOLDEN, "fx_se_k23")
    p = gu.run_fixture(CLI, "fx_se_k23", tmp_path, args_override=["-r", "reads.fq", "-r", os.path.join(a, "reads.fq"), "-k", "23", "-c", "dump.jf", "-batch", "128"])
    want = open(os.path.join(a, "ref", "reads.cor.fq"), "rb").read()
    assert open(tmp_path / "reads.cor.fq", "rb").read() == want  # second -r truncates and rewrites the same name
    assert b"Processed 800 reads" in p.stderr


def test_cli_gz_paired_multi_member_output(tmp_path):
    """.gz in -> .gz out for a pair (Reads.h:140-147 naming); the output is written as parallel gzip
    members, which must decompress to exactly the reference's records."""
    src = os.path.join(gu.GOLDEN, "fx_pe_k23")
    work = tmp_path / "in"
    work.mkdir()
    for n in ("reads_1.fq", "reads_2.fq"):
        with open(os.path.join(src, n), "rb") as f, gzip.open(work / (n + ".gz"), "wb") as g:
            shutil.copyfileobj(f, g)
    out = tmp_path / "out"
    gu.run_fixture(CLI, "fx_pe_k23", out, args_override=["-p", str(work / "reads_1.fq.gz"), str(work / "reads_2.fq.gz"),
                                                          "-k", "23", "-c", os.path.join(src, "dump.jf"), "-batch", "100"])
    for n in ("reads_1", "reads_2"):
        got = gzip.open(out / (n + ".cor.fq.gz"), "rb").read()
        assert got == open(os.path.join(src, "ref", n + ".cor.fq"), "rb").read()
    import subprocess
    assert subprocess.run(["gzip", "-t", str(out / "reads_1.cor.fq.gz")]).returncode == 0


def test_cli_unpaired_files_are_refused(tmp_path):
    """main.cpp:462-467: a mate file with a different number of records ends the run with
    'ERROR: The files are not paired!' and exit status 1."""
    import subprocess
    src = os.path.join(gu.GOLDEN, "fx_pe_k23")
    work = tmp_path / "in"
    work.mkdir()
    shutil.copy(os.path.join(src, "reads_1.fq"), work / "reads_1.fq")
    lines = open(os.path.join(src, "reads_2.fq"), "rb").read().split(b"\n")
    open(work / "reads_2.fq", "wb").write(b"\n".join(lines[:-5]) + b"\n")     # one record fewer
    p = subprocess.run([CLI, "-p", str(work / "reads_1.fq"), str(work / "reads_2.fq"), "-k", "23", "-c", os.path.join(src, "dump.jf"),
                        "-od", str(tmp_path / "out")], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 1 and b"ERROR: The files are not paired!" in p.stderr


def test_cli_empty_and_one_base_reads(oracle, tmp_path):
    """Records whose sequence line is empty (or a single base) travel through packing, the kernels
    and the writer like any other read (the reference prints them as unfixable, l/m/h 0)."""
    import subprocess
    import numpy as np
    import synth
    d = str(tmp_path)
    s1, q1, _, _, _ = synth.make_reads(3, 200, 60, n_tx=3, l_tx=300, e=0.01)
    keys, cnt = synth.count_kmers([s1], 23)
    synth.write_dump(os.path.join(d, "d.jf"), keys, cnt, 23)
    with open(os.path.join(d, "a.fq"), "wb") as f:
        for i in range(len(s1)):
            r, q = s1[i].tobytes(), q1[i].tobytes()
            if i % 7 == 3:
                r = q = b""
            if i % 11 == 5:
                r, q = r[:1], q[:1]
            f.write(b"@r%d\n%s\n+\n%s\n" % (i, r, q))
    outs = []
    for name, binary, more in (("g", CLI, ["-batch", "32"]), ("c", oracle.CLI_BIN, [])):
        od = os.path.join(d, name)
        os.makedirs(od)
        p = subprocess.run([binary, "-r", "a.fq", "-k", "23", "-c", "d.jf", "-od", od, "-verbose"] + more, cwd=d,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert p.returncode == 0, p.stderr.decode()
        outs.append((open(os.path.join(od, "a.cor.fq"), "rb").read(), p.stderr, p.stdout))
    for j, what in enumerate(("output", "stderr", "verbose transcript")):
        if outs[0][j] != outs[1][j]:
            a, b = outs[0][j].split(b"\n"), outs[1][j].split(b"\n")
            k = next(i for i in range(min(len(a), len(b))) if a[i] != b[i]) if any(x != y for x, y in zip(a, b)) else min(len(a), len(b))
            raise AssertionError("%s differs at line %d: gpu %r vs cpu %r (context %r)" % (what, k, a[k:k + 2], b[k:k + 2], b[max(0, k - 3):k]))
    assert b"@r3 l:0 m:0 h:0 unfixable_error\n\n+\n\n" in outs[0][0]


@pytest.mark.parametrize("name", ["fx_pe_k23", "fx_se_k23", "fx_il_k23", "fx_k31_mc8"])
@pytest.mark.parametrize("with_dump", [True, False])
def test_cli_two_gpu_path_equals_one_gpu(name, with_dump, tmp_path):
    """`-gpus 2`: the dump is parsed (or the k-mers counted) ONCE, the bucket array is replicated device
    to device, the digests are compared, and batches are dealt to whichever context is free; the output
    must be what `-gpus 1` writes, byte for byte (the reference: one Store for all workers,
    main.cpp:294-308,451).  RC_SHARED_GPU=1 puts both "GPUs" on device 0 so that this runs on a one-GPU box."""
    d = os.path.join(gu.GOLDEN, name)
    args = open(os.path.join(d, "cmd.txt")).read().split()
    if not with_dump:
        i = args.index("-c")
        args = args[:i] + args[i + 2:]
    import subprocess
    outs = []
    for gpus in (1, 2):
        od = tmp_path / ("g%d" % gpus)
        p = subprocess.run([CLI] + args + ["-od", str(od), "-gpus", str(gpus), "-batch", "64", "-inflight", "1"], cwd=d,
                           env=dict(os.environ, RC_SHARED_GPU="1"), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert p.returncode == 0, p.stderr.decode()
        outs.append((p.stderr, {f: open(od / f, "rb").read() for f in sorted(os.listdir(od))}))
    assert outs[0][0] == outs[1][0]
    assert outs[0][1] == outs[1][1] and outs[0][1]
    if with_dump:
        gu.assert_same_as_reference(name, tmp_path / "g2", outs[1][0])


@pytest.mark.parametrize("staged", [False, True])
def test_cli_eight_gpu_path_two_batches_in_flight_each(staged, tmp_path):
    """`-gpus 8 -inflight 2` -- sixteen worker threads over eight contexts, the table replicated to seven of them with
    all copies in flight at once (rc_table_replicate_async), directly device to device or (RC_REPLICATE_STAGED: what
    GPUs without peer access get) through page-locked host memory -- gives the reference's bytes.  RC_SHARED_GPU=1:
    all eight "GPUs" are device 0."""
    import subprocess
    name = "fx_pe_k23"
    d = os.path.join(gu.GOLDEN, name)
    args = open(os.path.join(d, "cmd.txt")).read().split()
    od = tmp_path / "g8"
    env = dict(os.environ, RC_SHARED_GPU="1")
    if staged:
        env["RC_REPLICATE_STAGED"] = "1"
    p = subprocess.run([CLI] + args + ["-od", str(od), "-gpus", "8", "-batch", "50", "-inflight", "2"], cwd=d, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()
    gu.assert_same_as_reference(name, od, p.stderr)


def test_cli_mixed_plain_and_gz_pair(tmp_path):
    """`-p a.fq.gz b.fq`: each output takes its compression from its own input name (Reads::AddReadFile,
    Reads.h:140-147): a gzip stream for the first mates, plain text for the second -- and the other way round."""
    src = os.path.join(gu.GOLDEN, "fx_pe_k23")
    for gz_first in (True, False):
        work = tmp_path / ("in%d" % gz_first)
        work.mkdir()
        names = []
        for n, z in (("reads_1.fq", gz_first), ("reads_2.fq", not gz_first)):
            if z:
                with open(os.path.join(src, n), "rb") as f, gzip.open(work / (n + ".gz"), "wb") as g:
                    shutil.copyfileobj(f, g)
                names.append(str(work / (n + ".gz")))
            else:
                shutil.copy(os.path.join(src, n), work / n)
                names.append(str(work / n))
        out = tmp_path / ("out%d" % gz_first)
        gu.run_fixture(CLI, "fx_pe_k23", out, args_override=["-p", names[0], names[1], "-k", "23", "-c", os.path.join(src, "dump.jf"), "-batch", "100"])
        for n, z in (("reads_1", gz_first), ("reads_2", not gz_first)):
            want = open(os.path.join(src, "ref", n + ".cor.fq"), "rb").read()
            if z:
                assert gzip.open(out / (n + ".cor.fq.gz"), "rb").read() == want
            else:
                assert open(out / (n + ".cor.fq"), "rb").read() == want


def test_cli_refuses_a_fastq_fasta_pair(tmp_path):
    import subprocess
    src = os.path.join(gu.GOLDEN, "fx_pe_k23")
    fa = tmp_path / "m.fa"
    lines = open(os.path.join(src, "reads_2.fq"), "rb").read().split(b"\n")
    with open(fa, "wb") as f:
        for i in range(0, len(lines) - 1, 4):
            f.write(b">" + lines[i][1:] + b"\n" + lines[i + 1] + b"\n")
    p = subprocess.run([CLI, "-p", os.path.join(src, "reads_1.fq"), str(fa), "-k", "23", "-c", os.path.join(src, "dump.jf"), "-od", str(tmp_path)],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode != 0 and b"same format" in p.stderr


def test_cli_fasta_input(tmp_path):
    """FASTA in -> .cor.fa out (no quality information: the vetoes see qual[0] == 0, the marker of the
    reference's batch path, Reads.h:241): the bytes the reference writes at -t 2."""
    p = gu.run_fixture(CLI, "fa_se_k23", tmp_path, extra=["-batch", "96"])
    gu.assert_same_as_reference("fa_se_k23", tmp_path, p.stderr)
