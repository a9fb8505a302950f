"""The pow rule LTP_POW_LIBM on the CPU: csrc/ltp_libm_pow.hpp — the header the device kernels include — compiled by plain g++ and
compared with the installed libm's pow bit for bit (tests/cpp/libm_pow_test.cc), and its tables against their generator."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_restated_glibc_pow_equals_the_installed_libm():
    """24.5 M inputs: the planner's exponents (0.5, 2, 3, 4, 6) on planner-like and on arbitrary x bit patterns (negative, subnormal,
    huge), arbitrary (x, y) incl. any y bit pattern, every pair of 34 edge values, results in the subnormal / overflow range.
    (profiles/r05_libm_pow_host_soak.json: 1.7e10 inputs, no mismatch.) Needs a host with FMA: glibc then runs the variant restated."""
    flags = open("/proc/cpuinfo").read()
    if not (" fma " in flags and " avx2 " in flags):
        pytest.skip("this host's glibc runs the non-FMA pow: the restated variant is the FMA one (ltp_host_libm_pow_rule tells a caller so)")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp"), "-s", "libm_pow_test"])
    p = subprocess.run([os.path.join(ROOT, "tests", "cpp", "libm_pow_test"), "2", "20251004"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["mismatches"] == 0 and line["inputs"] > 24_000_000
    assert 0.0 < line["libm_pow3_not_correctly_rounded_frac"] < 0.01      # why the rule exists: libm's cube is not always the rounded cube


def test_pow_tables_are_what_the_generator_computes():
    """ltp_libm_pow_tables.inc is generated (mpmath) from the rules glibc's sources state, not copied: regenerate and compare; here,
    where a glibc is installed, also compare every entry with the tables inside libm.so.6."""
    gen = os.path.join(ROOT, "tools", "gen_libm_pow_tables.py")
    subprocess.check_call([sys.executable, gen, "--check"])
    if os.path.exists("/lib/x86_64-linux-gnu/libm.so.6"):
        subprocess.check_call([sys.executable, gen, "--check", "--check-libm"])


def test_host_probe_names_the_rule_of_the_installed_libm():
    """ltp_host_libm_pow_rule — the installed pow(x, 3 | 4 | 6 | 1/2) against both rules, in the caller's process,
    no GPU. Here (glibc >= 2.28 on an FMA host) it must answer LTP_POW_LIBM with no mismatch, and see that the installed pow is NOT
    the correctly rounded one (about one power in a thousand); the answer must agree with the 24.5 M-input comparison above."""
    sys.path.insert(0, ROOT)
    from longtermplanner_amd import LongTermPlanner
    rule, n_libm, n_exact = LongTermPlanner.powRuleMatchingHostLibm()
    flags = open("/proc/cpuinfo").read()
    if " fma " in flags and " avx2 " in flags and os.path.exists("/lib/x86_64-linux-gnu/libm.so.6"):
        assert rule == "libm" and n_libm == 0, (rule, n_libm, n_exact)
        assert 50 < n_exact < 2000, n_exact          # 2^18 probes, ~1 in 1 200 not correctly rounded
    else:
        assert rule in ("libm", "exact", None) and (rule != "libm" or n_libm == 0) and (rule != "exact" or n_exact == 0)
    # fewer probes: same verdict, counts scale
    r2, l2, e2 = LongTermPlanner.powRuleMatchingHostLibm(1 << 14)
    assert r2 == rule or (rule is None and r2 == "exact") and l2 <= n_libm and e2 <= n_exact
