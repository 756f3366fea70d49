"""The C oracle under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build only; GPU sanitizers are not
available on this pool).  Index arithmetic of every constraint group / COO segment is exercised for
k_trans from 1 to N+1 and both init modes."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_oracle_is_clean_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "sanitize_driver")
    src = os.path.join(ROOT, "oracle")
    subprocess.check_call(["gcc", "-O1", "-g", "-ffp-contract=off", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=undefined", "-o", exe, os.path.join(src, "sanitize_driver.c"),
                           os.path.join(src, "qln_oracle.c"), "-lm"])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    out = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "sanitize_driver ok" in out.stdout
