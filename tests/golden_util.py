"""Shared helpers of the golden-fixture tests: run a stage-3 binary on a fixture exactly as
tests/golden/make_golden.py ran the reference, and diff every output byte."""
import glob
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
FIXTURES = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "fx_*")))


def run_fixture(binary, name, outdir, extra=(), args_override=None):
    d = os.path.join(GOLDEN, name)
    args = args_override if args_override is not None else open(os.path.join(d, "cmd.txt")).read().split()
    p = subprocess.run([binary] + list(args) + ["-od", str(outdir)] + list(extra), cwd=d,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()
    return p


def assert_same_as_reference(name, outdir, stderr_bytes, check_stderr=True):
    ref = os.path.join(GOLDEN, name, "ref")
    files = sorted(f for f in os.listdir(ref) if f != "stderr.txt" and not f.startswith("verbose"))
    assert files, "fixture %s has no reference outputs" % name
    for f in files:
        want = open(os.path.join(ref, f), "rb").read()
        got = open(os.path.join(str(outdir), f), "rb").read()
        assert got == want, "%s/%s differs from the reference's output" % (name, f)
    if check_stderr:
        want = open(os.path.join(ref, "stderr.txt"), "rb").read()
        assert stderr_bytes == want, "stderr of %s differs:\n%s\nvs\n%s" % (name, stderr_bytes.decode(), want.decode())
