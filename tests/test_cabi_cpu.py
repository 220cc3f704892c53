"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every
symbol include/shark_hip.h declares, refuses to compute without a GPU (no CPU
fallback), and the `shark` CLI keeps the reference's argument contract."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "shark_amd", "libsharkhip.so")
CLI = os.path.join(ROOT, "shark_amd", "bin", "shark")


@pytest.fixture(scope="module")
def built():
    if not (os.path.exists(LIB) and os.path.exists(CLI)):
        subprocess.run(["make", "-C", os.path.join(ROOT, "shark_amd", "csrc"), "-j4", "all"], check=True,
                       stdout=subprocess.DEVNULL)
    return True


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "shark_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(shk_[a-z_0-9]+)\s*\(", hdr)))


def test_header_symbols_are_exported(built):
    syms = _declared_symbols()
    assert len(syms) >= 19
    lib = C.CDLL(LIB)
    for s in syms:
        assert hasattr(lib, s), "libsharkhip.so does not export %s" % s
    from shark_amd import EXPORTS
    assert sorted(EXPORTS) == syms


def test_no_cpu_fallback_without_gpu(built):
    """without a HIP device shk_create fails with SHK_ERR_NO_DEVICE; nothing computes on the CPU"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from shark_amd import SharkHip, SharkHipError
    with pytest.raises(SharkHipError, match="no HIP device"):
        SharkHip(k=17, bf_bits=1 << 20)
    lib = C.CDLL(LIB)
    lib.shk_strerror.restype = C.c_char_p
    assert lib.shk_strerror(-7) == b"no HIP device"
    assert lib.shk_strerror(0) == b"ok"


def test_product_does_not_reference_the_oracle():
    """the oracle is test infrastructure: nothing under shark_amd/ may import, link or call it"""
    bad = []
    for d, _, files in os.walk(os.path.join(ROOT, "shark_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h", "Makefile")):
                txt = open(os.path.join(d, f), errors="ignore").read()
                if re.search(r"\boracle\b|liboracle|so_analyze|so_shark", txt):
                    bad.append(os.path.join(d, f))
    assert not bad, bad
    out = subprocess.run(["ldd", LIB], capture_output=True, text=True).stdout if os.path.exists(LIB) else ""
    assert "oracle" not in out


def _run_cli(args):
    return subprocess.run([CLI] + args, capture_output=True, text=True)


def test_cli_argument_contract(built, tmp_path):
    """argument_parser.hpp:84-174: exit codes and messages"""
    r = _run_cli(["-h"])
    assert r.returncode == 0 and r.stderr.startswith("Usage: shark -r <references> -1 <sample1>")
    r = _run_cli([])
    assert r.returncode == 1 and "shark : missing required files" in r.stderr
    r = _run_cli(["-r", "x.fa", "-1", "y.fq", "-k", "32"])
    assert r.returncode == 1 and "shark: k must be in the range [1, 31]." in r.stderr
    r = _run_cli(["-r", "x.fa", "-1", "y.fq", "-k", "0"])
    assert r.returncode == 1
    r = _run_cli(["-r", "x.fa", "-1", "y.fq", "-c", "1.5"])
    assert r.returncode == 1 and "shark: c must be in the range [0, 1]." in r.stderr
    r = _run_cli(["-r", "x.fa", "-1", "y.fq", "-q", "-3"])
    assert r.returncode == 1 and "shark: q must be a positive value." in r.stderr
    r = _run_cli(["-r", "x.fa", "-1", "y.fq", "-t", "0"])
    assert r.returncode == 1 and "shark: at least 1 thread is required." in r.stderr
    r = _run_cli(["--bogus"])
    assert r.returncode == 1 and "shark : unknown argument" in r.stderr


def test_bench_starts_its_own_ranks_and_fails_loudly_without_a_gpu():
    """`python3 bench.py --gpus 2` (no launcher): the parent starts two ranks before touching torch or a GPU and passes a
    rank's failure on -- here every rank fails because there is no GPU and no CPU path"""
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and not r.stdout.strip()
    assert r.stderr.count("no GPU (there is no CPU path)") >= 1 and "of 2 exited with code" in r.stderr
    env.update(WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=3" in r.stderr


def test_cli_reports_a_sample_that_cannot_be_opened(built, tmp_path):
    """a typo'd -1 / -2 path is an error message and exit code 1 at once, not a hang behind the index build"""
    fa = tmp_path / "g.fa"
    fa.write_text(">g\nACGTACGTACGTACGTACGTACGT\n")
    fq = tmp_path / "a.fq"
    fq.write_text("@r\nACGTACGTACGTACGTACGT\n+\nIIIIIIIIIIIIIIIIIIII\n")
    r = subprocess.run([CLI, "-r", str(fa), "-1", str(tmp_path / "nonexistent.fq")], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "cannot open the sample" in r.stderr
    r = subprocess.run([CLI, "-r", str(fa), "-1", str(fq), "-2", str(tmp_path / "nonexistent.fq")], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "cannot open the sample" in r.stderr


def test_cli_error_exits_are_exit_code_1_with_the_default_outputs(built, tmp_path):
    """every error return behind the point where the output writers exist -- with the DEFAULT -o (sharked_sample.1 is always opened,
    argument_parser.hpp:168-173, so a writer with live helper threads is in scope) -- is exit code 1 and a message, never
    std::terminate from a joinable thread's destructor (rc 134): a reference that cannot be opened, and, on a box without a GPU,
    the context that cannot be created"""
    fq = tmp_path / "a.fq"
    fq.write_text("@r\nACGTACGTACGTACGTACGT\n+\nIIIIIIIIIIIIIIIIIIII\n")
    r = subprocess.run([CLI, "-r", str(tmp_path / "nonexistent.fa"), "-1", str(fq), "-2", str(fq)], capture_output=True, text=True, timeout=120, cwd=tmp_path)
    assert r.returncode == 1, (r.returncode, r.stderr[-500:])
    assert "cannot open" in r.stderr and "terminate called" not in r.stderr
    import torch
    if torch.cuda.is_available():
        return
    fa = tmp_path / "g.fa"
    fa.write_text(">g\nACGTACGTACGTACGTACGTACGT\n")
    r = subprocess.run([CLI, "-r", str(fa), "-1", str(fq), "-2", str(fq)], capture_output=True, text=True, timeout=120, cwd=tmp_path)
    assert r.returncode == 1, (r.returncode, r.stderr[-500:])
    assert "cannot create a context on GPU 0" in r.stderr and "terminate called" not in r.stderr
    r = subprocess.run([CLI, "-r", str(fa), "-1", str(fq)], capture_output=True, text=True, timeout=120, cwd=tmp_path)
    assert r.returncode == 1 and "terminate called" not in r.stderr
