"""The host-only translation units of libcrd (crd_host.cpp: geometry, slabs, halo plan, steady states, ICs; crd_io.cpp: ini
reader, text writer; crd_trace.cpp: the roctx binding, here with nobody listening) under AddressSanitizer + UndefinedBehaviorSanitizer, and under ThreadSanitizer (the writer's thread pool).  GPU code cannot be sanitised on this pool, host
code can: it is plain C++, so g++ builds it without the HIP toolchain."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, "crdmodel_amd", "csrc")


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
@pytest.mark.parametrize("sanitizers", ["address,undefined", "thread"])
def test_host_entry_points_under_sanitizers(tmp_path, sanitizers):
    exe = tmp_path / "host_sanitize"
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=" + sanitizers, "-fno-sanitize-recover=all",
           "-I", os.path.join(ROOT, "include"), "-I", CSRC,
           os.path.join(ROOT, "tests", "native", "host_sanitize.cpp"), os.path.join(CSRC, "crd_host.cpp"), os.path.join(CSRC, "crd_io.cpp"),
           os.path.join(CSRC, "crd_trace.cpp"), "-o", str(exe), "-lpthread", "-ldl"]
    build = subprocess.run(cmd, capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr and "cannot find" in build.stderr:
        pytest.skip("sanitizer runtimes not installed")
    assert build.returncode == 0, build.stderr[-4000:]
    work = tmp_path / "work"
    work.mkdir()
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", TSAN_OPTIONS="halt_on_error=1",
               CRD_WRITER_THREADS="4")  # the writer formats rows on a thread pool: make sure it has threads to race with
    run = subprocess.run([str(exe), str(work)], capture_output=True, text=True, env=env, timeout=240)
    if run.returncode != 0 and "unexpected memory mapping" in run.stderr:
        pytest.skip("ThreadSanitizer cannot map its shadow memory under this kernel's address-space layout")
    assert run.returncode == 0, (run.stdout[-2000:], run.stderr[-6000:])
    assert "host sanitize run ok" in run.stdout
