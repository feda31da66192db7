"""SURVEY.md section 5, sanitizer row (CPU container only; nothing here touches a GPU, and gpurun refuses sanitizer runs there).

  * the C oracle under gcc's AddressSanitizer + UndefinedBehaviorSanitizer (`make -C oracle asan`): its own CPU tests re-run in a
    python started with LD_PRELOAD=libasan against that build;
  * the HOST side of libsilent_hip -- ~2 700 lines of validation, tile / region / tap tables, row programs, walk plans, weight
    streams, workspace layout and staging -- built without a GPU behind it (pysilent_amd/csrc/silent_host_shim.h: launches
    compiled out, device memory = host memory) under clang's ASan + UBSan, driven by tests/sanitizer_worker.py: fuzzed extents /
    crops / regions, bad arguments, and the exception barrier of the C ABI (an injected std::bad_alloc inside every entry point,
    and real allocation failures through the planners, must come back as SILENT_E_NOMEM).
"""
import os
import shutil
import subprocess
import sys

import pytest

from conftest import ROOT

SAN_ENV = {"ASAN_OPTIONS": "detect_leaks=0:alloc_dealloc_mismatch=0:abort_on_error=0",
           "UBSAN_OPTIONS": "print_stacktrace=1:halt_on_error=1"}


def _run(cmd, env, timeout):
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout, cwd=ROOT)
    return p.returncode, p.stdout


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_c_oracle_under_asan_ubsan():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"])
    rt_lib = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    if not os.path.isabs(rt_lib):
        pytest.skip("gcc has no shared asan runtime")
    env = dict(os.environ, LD_PRELOAD=rt_lib, SILENT_ORACLE_SO=os.path.join(ROOT, "oracle", "libsilent_oracle_asan.so"),
               OMP_NUM_THREADS="4", **SAN_ENV)
    rc, out = _run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", "tests/test_oracle.py", "-k",
                    "c_port or rgb_pass_frames or region_pool or max_pool_ignores"], env, 600)
    assert rc == 0 and "AddressSanitizer" not in out and "runtime error" not in out, out[-4000:]
    assert " passed" in out


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not available")
def test_host_side_of_the_library_under_asan_ubsan():
    sys.path.insert(0, os.path.join(ROOT, "pysilent_amd", "csrc"))
    import build as B
    lib = B.build_host_asan()
    rt_lib = B.asan_runtime()
    if rt_lib is None:
        pytest.skip("the ROCm LLVM has no shared asan runtime")
    env = dict(os.environ, LD_PRELOAD=rt_lib, SILENT_LIB_PATH=lib, **SAN_ENV)
    rc, out = _run([sys.executable, os.path.join(ROOT, "tests", "sanitizer_worker.py"), "0", "12"], env, 600)
    assert rc == 0 and "sanitizer worker ok" in out, out[-4000:]
    assert "AddressSanitizer" not in out and "runtime error" not in out, out[-4000:]


def test_the_product_library_has_no_host_only_symbols():
    """The fault injectors and the host-only shim exist in the sanitizer build only."""
    out = subprocess.check_output(["nm", "-D", "--defined-only", os.path.join(ROOT, "pysilent_amd", "lib", "libsilent_hip.so")], text=True)
    assert "silent_host_" not in out and "__hipRegisterFatBinary" not in out.replace("U __hip", "")
