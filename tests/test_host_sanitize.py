"""Host scene model under AddressSanitizer + UndefinedBehaviorSanitizer.

Compiles tests/native/host_sanitize.cpp together with the C entry points of the
host mirror (path-tracer_amd/host/pt_host_capi.cpp) with -fsanitize=address,undefined
and runs it: scene builders, flattening, scene-file round trips, malformed files
and error paths.  Host code only; needs no GPU.
"""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "path-tracer_amd", "host")
LIB = os.path.join(ROOT, "path-tracer_amd", "lib")


def test_host_model_is_clean_under_sanitizers(tmp_path):
    exe = tmp_path / "host_sanitize"
    cmd = ["g++", "-std=c++17", "-g", "-O1", "-ffp-contract=off", "-fsanitize=address,undefined",
           "-fno-sanitize-recover=undefined", "-static-libasan", "-static-libubsan",
           os.path.join(ROOT, "tests", "native", "host_sanitize.cpp"), os.path.join(HOST, "pt_host_capi.cpp"),
           "-I" + os.path.join(ROOT, "include"), "-I" + HOST, "-L" + LIB, "-lpt_amd",
           "-Wl,-rpath," + LIB, "-o", str(exe)]
    build = subprocess.run(cmd, capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-4000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:verify_asan_link_order=0",
               UBSAN_OPTIONS="print_stacktrace=1")
    run = subprocess.run([str(exe), str(tmp_path)], capture_output=True, text=True, env=env, timeout=300)
    assert run.returncode == 0, (run.stdout + run.stderr)[-6000:]
    assert "host checks passed" in run.stdout
