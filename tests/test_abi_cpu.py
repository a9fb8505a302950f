"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every declared symbol, and the
product fails loudly (never falls back to a CPU path) when no HIP device is present."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def abi():
    from longtermplanner_amd import _abi
    _abi.build()
    return _abi


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "ltp_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ltp_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(abi):
    names = _declared_symbols()
    assert len(names) >= 20
    lib = C.CDLL(abi.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ltp_hip.h but not exported"
    # and the Python binding binds exactly the declared set
    assert sorted(abi.exported_symbols()) == names


def test_row_stride(abi):
    lib = abi.lib()
    assert lib.ltp_row_stride(0) == 0 and lib.ltp_row_stride(-3) == 0
    assert lib.ltp_row_stride(1) == 32 and lib.ltp_row_stride(32) == 32 and lib.ltp_row_stride(33) == 64
    assert lib.ltp_row_stride(1718) == 1728


def test_sample_opts_are_validated_before_anything_runs(abi):
    """ltp_sample_batch_ex: the policy is a size-versioned struct of named fields; an unset size or a field out of
    range is LTP_ERR_INVALID_ARGUMENT before any device work (checked here without a device and without a handle)."""
    lib = abi.lib()
    O = abi.SampleOpts
    assert C.sizeof(O) == 28
    text = open(os.path.join(ROOT, "include", "ltp_hip.h")).read()
    fields = re.search(r"typedef struct \{([^}]*)\} ltp_sample_opts;", text, flags=re.S).group(1)
    assert re.findall(r"^\s*(?:unsigned|int) (\w+);", fields, flags=re.M) == [f[0] for f in O._fields_]

    def call(o):
        return lib.ltp_sample_batch_ex(None, 0, 0, None, None, None, None, 0, C.addressof(o) if o is not None else None, None)
    INVALID = 1
    assert call(O(0, 0, 0, 0, 0, 0, 0)) == INVALID                      # size not set
    for bad in (dict(format=2), dict(stores=-1), dict(sampler=5), dict(verdict=2), dict(interleave=70000), dict(dry_run=3)):
        assert call(O(**dict(dict(size=C.sizeof(O)), **bad))) == INVALID, bad
    # a well-formed struct gets past the field checks and is refused for the null handle / arguments instead (same code, later check)
    assert call(O(size=C.sizeof(O), sampler=2, verdict=1)) == INVALID and call(None) == INVALID


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_has_gpu(), reason="checks the no-device behaviour")
def test_no_device_fails_loudly(abi):
    import longtermplanner_amd as m
    with pytest.raises(m.LtpError) as e:
        m.LongTermPlanner(1, 0.001, [-1], [1], [1], [1], [1])
    assert e.value.code == 2  # LTP_ERR_NO_DEVICE


def test_product_never_touches_the_oracle():
    # the oracle is test infrastructure: nothing under the package, include/ or the C-ABI sources may reference it
    bad = []
    for base in ("longtermplanner_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".hpp", ".hip", ".cc", "Makefile")):
                    txt = open(os.path.join(dirpath, f), errors="ignore").read()
                    if re.search(r"\boracle\b|ltpo_|libltp_oracle", txt):
                        bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_cpp_dropin_builds_with_plain_gxx_and_fails_loudly_without_gpu(abi):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp"), "-s", "all"])
    exe = os.path.join(ROOT, "tests", "cpp", "dropin_tests")
    assert os.path.exists(exe)
    if not _has_gpu():
        r = subprocess.run([exe], capture_output=True, text=True)
        assert r.returncode == 2 and "no CPU fallback" in r.stdout


def test_user_side_consumer_builds_against_the_public_device_header_only(abi):
    """tests/cpp/example_consumer.hip — a user's own on-device consumer of the run tables — compiles with plain
    hipcc against include/ltp_run_tables.hpp and nothing else of this repository, and the build records whether it saw the
    reference (tests/cpp/build_stamp.txt, read by the GPU suite)."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp"), "-s", "all"])
    src = open(os.path.join(ROOT, "tests", "cpp", "example_consumer.hip")).read()
    includes = re.findall(r'#include\s+[<"]([^>"]+)[>"]', src)
    assert sorted(includes) == ["hip/hip_runtime.h", "ltp_run_tables.hpp"], includes
    hdr = open(os.path.join(ROOT, "include", "ltp_run_tables.hpp")).read()
    assert re.findall(r'#include\s+[<"]([^>"]+)[>"]', hdr) == ["hip/hip_runtime.h"], "the public device header must be self-contained"
    lib = C.CDLL(os.path.join(ROOT, "tests", "cpp", "libexample_consumer.so"))
    assert hasattr(lib, "example_peak_velocity") and hasattr(lib, "example_box_clearance")
    stamp = open(os.path.join(ROOT, "tests", "cpp", "build_stamp.txt")).read()
    assert ("reference_present=1" in stamp) == os.path.exists("/root/reference/tests/src/long_term_planner_tests.cc")
    if "reference_present=1" in stamp:
        assert os.path.exists(os.path.join(ROOT, "tests", "cpp", "reference_tests"))


def _build_cmake_consumer(tmp_path):
    import shutil
    if shutil.which("cmake") is None:
        pytest.skip("cmake not installed")
    build = tmp_path / "build"
    subprocess.check_call(["cmake", "-S", os.path.join(ROOT, "tests", "cmake_consumer"), "-B", str(build),
                           f"-Dlong_term_planner_DIR={os.path.join(ROOT, 'cmake')}"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["cmake", "--build", str(build)], stdout=subprocess.DEVNULL)
    return str(build / "consumer")


def test_cmake_package_resolves_like_the_reference(abi, tmp_path):
    """find_package(long_term_planner) + target long_term_planner::long_term_planner, version 1.0.0
    (/root/reference/CMakeLists.txt:17-57): an unchanged consumer project configures, builds and links against the
    drop-in; without a GPU the program reports the missing device instead of computing on the CPU."""
    exe = _build_cmake_consumer(tmp_path)
    p = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    import torch
    if torch.cuda.device_count() == 0:
        assert p.returncode == 3 and "no device" in p.stdout
    else:
        assert p.returncode == 0 and "planTrajectory: true" in p.stdout


def test_a_touched_header_makes_the_library_stale():
    """_abi.build() takes its inputs from the Makefile (`make print-deps`), so an edit of ANY header the kernels
    include — the public include/ltp_run_tables.hpp, which holds the sampler's arithmetic, among them — requests a rebuild."""
    import os
    from longtermplanner_amd import _abi
    deps = _abi.build_inputs()
    names = {os.path.relpath(d, ROOT) for d in deps}
    for must in ("include/ltp_run_tables.hpp", "include/ltp_hip.h", "longtermplanner_amd/csrc/ltp_libm_pow.hpp",
                 "longtermplanner_amd/csrc/ltp_libm_pow_tables.inc", "longtermplanner_amd/csrc/ltp_sampler.hip", "longtermplanner_amd/csrc/Makefile"):
        assert must in names, must
    csrc = os.path.join(ROOT, "longtermplanner_amd", "csrc")
    listed = {os.path.basename(d) for d in deps}
    assert all(f in listed for f in os.listdir(csrc) if f.endswith((".hip", ".hpp", ".inc"))), "a source file is missing from the Makefile's lists"
    _abi.build()
    assert not _abi.stale()
    hdr = os.path.join(ROOT, "include", "ltp_run_tables.hpp")
    st = os.stat(hdr)
    try:
        os.utime(hdr, (st.st_atime, os.path.getmtime(_abi.LIB_PATH) + 5))
        assert _abi.stale()
    finally:
        os.utime(hdr, (st.st_atime, st.st_mtime))
    assert not _abi.stale()


def test_eigen_typed_roots_overloads_compile_and_resolve():
    """The Eigen-typed overloads of include/long_term_planner/roots.h against tests/cpp/not_eigen (a container-only stand-in: NOT
    Eigen): syntax and overload resolution (static_asserts in tests/cpp/roots_eigen_signature_test.cc), no GPU needed. Without the
    stand-in on the include path the same header compiles with the std::vector signatures only."""
    cpp = os.path.join(ROOT, "tests", "cpp")
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-I" + os.path.join(cpp, "not_eigen"), "-I" + inc,
                           os.path.join(cpp, "roots_eigen_signature_test.cc")])
    p = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I" + inc, "-x", "c++", "-"], input='#include "long_term_planner/roots.h"\n'
                       'int main() { return (int)long_term_planner::roots<double>(std::vector<double>{1.0, -3.0, 2.0}).size(); }\n', text=True, capture_output=True)
    assert p.returncode == 0, p.stderr
