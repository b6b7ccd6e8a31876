"""Host algorithms of libgasm + the oracle under AddressSanitizer and UndefinedBehaviorSanitizer (CPU build only: g++
-fsanitize=address,undefined; the GPU pool has no sanitizer support).  tests/san_driver.cpp runs the greedy merge in both
forms, the shuffle, the signatures, Myers' edit distance and the sequence-file reader on randomised inputs and compares
with the oracle; any sanitizer report fails the run."""
import gzip
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_algorithms_under_asan_ubsan(tmp_path):
    csrc = os.path.join(ROOT, "genomeassembler_dev_amd", "csrc")
    exe = os.path.join(ROOT, "oracle", "_build", "san_driver")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
           "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + csrc, "-Wno-deprecated-declarations",
           os.path.join(ROOT, "tests", "san_driver.cpp"), os.path.join(csrc, "host_algos.cpp"), os.path.join(csrc, "seqio.cpp"),
           os.path.join(ROOT, "oracle", "gasm_oracle.cpp"), "-o", exe, "-lpthread", "-lz"]
    subprocess.run(cmd, check=True, cwd=ROOT)
    fq = tmp_path / "a.fastq"
    fq.write_text("@r1\nACGTAC\n+\nIIIIII\n@r2\nacgtn\n+\nIIIII\n@r3\nTTTT\n+\n@@@@")
    fa = tmp_path / "b.fa.gz"
    with gzip.open(fa, "wb") as f:
        f.write(b">c1\nACG\nTAC\n>c2\nGGGG\n\n>c3\n" + b"ACGT" * 5000 + b"\n")
    bad = tmp_path / "c.txt"
    bad.write_text("not a sequence file\n")
    # a gzip stream cut off in the middle: zlib hands back what it could inflate and then an error — the reader must not
    # return the shortened read set as if it were the file (GASM_ERR_INVALID = -1)
    import random
    rnd = random.Random(5)
    whole = gzip.compress("".join(f">r{i}\n{''.join(rnd.choice('ACGT') for _ in range(80))}\n" for i in range(4000)).encode())
    cut = tmp_path / "d.fa.gz"
    cut.write_bytes(whole[:len(whole) // 2])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe, str(fq), str(fa), str(bad), str(cut)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "checks ok" in r.stdout and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr
    assert "a.fastq: status 0, 2 reads kept, 1 dropped, 10 bases" in r.stdout
    assert "b.fa.gz: status 0, 3 reads kept, 0 dropped, 20010 bases" in r.stdout
    assert "c.txt: status -1" in r.stdout
    assert "d.fa.gz: status -1" in r.stdout, r.stdout
